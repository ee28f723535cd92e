#!/bin/bash
# the tail of the training step: conv2 weight gradient + its reduction on the caller's stream behind the conv2 data gradient, layer-1 launch and
# conv3..5 reduction on the side stream (default) against the round-2 order (SCS_TAIL_SIDE=1: all but the layer-1 launch on the side stream),
# same GPU, three rounds
for round in 1 2 3; do
for b in 256 128; do
for form in side main; do
  if [ $form = side ]; then export SCS_TAIL_SIDE=1; else unset SCS_TAIL_SIDE; fi
  t=$(timeout -k 10 200 python bench.py --mode train --batch $b --steps 300 --warmup 30 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4), d['launches_per_step'], 'launches')")
  echo "[batch $b, conv2 weight gradient on the $form stream] train $t"
done; done; done
