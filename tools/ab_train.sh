#!/bin/bash
# A/B of training-step variants selected by environment variables: tools/ab_train.sh "VAR=1" "VAR2=x" ...
for v in "" "$@"; do
  echo -n "[${v:-baseline}]  "; env $v timeout -k 10 200 python bench.py --mode train --steps 300 --warmup 30 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4), 'ms/step ', round(d['value']), 'patches/s')"
done
