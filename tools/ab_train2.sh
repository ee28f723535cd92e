#!/bin/bash
# training-step A/B of library builds on ONE GPU, two rounds:  tools/ab_train2.sh shipped name1 ...  (lib/libscs_<name>.so)
for round in 1 2; do
for lib in "$@"; do
  if [ "$lib" != "shipped" ]; then export SCS_LIB_PATH=$PWD/sub-cortical_segmentation_amd/lib/libscs_$lib.so; else unset SCS_LIB_PATH; fi
  echo -n "[$lib] "
  for b in 256 128; do timeout -k 10 200 python bench.py --mode train --batch $b --steps 400 --warmup 40 --no-cpu --no-sustained 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['per_gpu_batch'], round(d['ms_per_step'],4), end='   ')"; done; echo
done; done
