#!/bin/bash
# conv5 block backward as one launch (conv5_bwd.h) against the generic data-gradient + weight-gradient kernels, same GPU, two rounds
for round in 1 2; do
for b in 256 128; do
for form in 0 1024; do
  t=$(SCS_C5_FUSED_MAX_BATCH=$form timeout -k 10 200 python bench.py --mode train --batch $b --steps 300 --warmup 30 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); st=d['stages']; k=[x for x in st if x.startswith('conv5 backward') or x=='conv5 dgrad'][0]; print(round(d['ms_per_step'],4), k[:14], 'stage', round(st[k]['ms']*1000,1), 'us', d['launches_per_step'], 'launches')")
  echo "[batch $b, fused bound $form] train $t"
done; done; done
