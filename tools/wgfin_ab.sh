#!/bin/bash
# reduction of the conv3..5 weight-gradient partials on the caller's stream behind the layer-1 launch (beside the conv2 weight gradient)
# against the side stream in front of that weight gradient (SCS_WGFIN_SIDE=1), same GPU, two rounds
for round in 1 2; do
for b in 256 128; do
for form in side main; do
  if [ $form = side ]; then export SCS_WGFIN_SIDE=1; else unset SCS_WGFIN_SIDE; fi
  t=$(timeout -k 10 200 python bench.py --mode train --batch $b --steps 300 --warmup 30 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4), d['launches_per_step'], 'launches')")
  echo "[batch $b, conv3..5 reduction on the $form stream] train $t"
done; done; done
