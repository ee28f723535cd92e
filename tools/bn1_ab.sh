#!/bin/bash
# conv1's batch statistics finalized inside the prep + moments launch against bn1_final_kernel as a launch of its own, same GPU, two rounds
for round in 1 2; do
for b in 256 128; do
for form in 0 512; do
  t=$(SCS_BN1_FUSED_MAX_BATCH=$form timeout -k 10 200 python bench.py --mode train --batch $b --steps 300 --warmup 30 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); st=d['stages']; k=[x for x in st if x.startswith('prep')][0]; f=st.get('bn1 finalize',{'ms':0}); print(round(d['ms_per_step'],4), 'prep+moments', round(st[k]['ms']*1000,1), 'us, bn1 finalize', round(f['ms']*1000,1), 'us', d['launches_per_step'], 'launches')")
  echo "[batch $b, bn1 fused bound $form] train $t"
done; done; done
