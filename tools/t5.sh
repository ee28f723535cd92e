python -m pytest tests/test_gpu_train.py -x -q -k "conv5 or dense_head" > gpurun_out/r04_t5_tests.log 2>&1; tail -3 gpurun_out/r04_t5_tests.log
for tf in 0 1; do
echo "== SCS_TAIL_FUSED=$tf"
SCS_TAIL_FUSED=$tf SCS_CHAIN_STAMPS=1 python bench.py --mode train --batch 256 --steps 3 --warmup 2 --no-cpu --no-sustained 2>&1 | grep -A8 "dense chain" | tail -9
SCS_TAIL_FUSED=$tf python bench.py --mode train --batch 256 --steps 300 --warmup 30 --no-cpu --no-sustained 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['per_gpu_batch'], round(d['ms_per_step'],4), round(d['net_frac_of_f32_peak'],4), d['launches_per_step'])"
done
