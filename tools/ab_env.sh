#!/bin/bash
# training-step A/B of environment switches on ONE GPU, two rounds:  tools/ab_env.sh "" "VAR=1" "VAR2=x" ...
for round in 1 2; do
for v in "$@"; do
  echo -n "[${v:-default}] "
  for b in 256 128; do env $v timeout -k 10 200 python bench.py --mode train --batch $b --steps 400 --warmup 40 --no-cpu --no-sustained 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['per_gpu_batch'], round(d['ms_per_step'],4), end='   ')"; done; echo
done; done
