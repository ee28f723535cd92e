for s in "64,128,256,64" "32,64,128,32" "32,32,64,32" "16,32,64,16" "64,64,128,32" "128,128,256,64"; do
  echo -n "$s  "; SCS_WG_SLICES=$s timeout -k 10 200 python bench.py --mode train --steps 300 --warmup 30 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4), {k:round(v['ms'],3) for k,v in d['stages'].items() if 'wgrad' in k})"
done
