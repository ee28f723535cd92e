#!/bin/bash
# dense head as one launch (dense_chain.h) against one launch per layer, same process image, same GPU, two rounds
for round in 1 2; do
for b in 256 128; do
for form in 0 384; do
  t=$(SCS_DENSE_CHAIN_MAX_BATCH=$form timeout -k 10 200 python bench.py --mode train --batch $b --steps 300 --warmup 30 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4))")
  echo "[batch $b, chain bound $form] train $t ms"
done; done; done
