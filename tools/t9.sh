python -m pytest tests/test_gpu_forward.py -x -q > gpurun_out/r04_t9_tests.log 2>&1; tail -3 gpurun_out/r04_t9_tests.log
for round in 1 2; do for ff in 0 1; do echo -n "[fwd_fused $ff] "; SCS_FWD_FUSED=$ff timeout -k 10 200 python bench.py --no-cpu --no-train-block --no-sustained --no-dice --steps 10 --warmup 2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['stages']
print(round(d['value']), round(d['roofline']['frac'],4), ' '.join('%s %.3f/%.0f' % (k.split('+')[0][:6], v['ms_per_step'], v['tflops'] or 0) for k,v in s.items()))"; done; done
