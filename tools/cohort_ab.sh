#!/bin/bash
# whole-cohort inference with the towers evaluated per voxel ('patch') and shared between neighbours ('shared'), same GPU
for t in patch shared; do
  timeout -k 10 500 python bench.py --mode cohort --subjects ${SUBJECTS:-12} --towers $t --no-cpu 2>/dev/null | tail -1 > gpurun_out/cohort_$t.json
  python3 -c "
import json; d=json.load(open('gpurun_out/cohort_$t.json'))
print('[$t]', round(d['value']), 'subjects/h  wall %.3f s  gpu %.4f s  cpu %.3f s per subject' % (d['host']['wall_seconds_per_subject_per_rank'], d['host']['gpu_seconds_per_subject'], d['host']['cpu_seconds_per_subject']), d['config']['towers'])"
done
