for lib in "" mt2w4 mt2w3 mt3w3 mt4w2; do
  if [ -n "$lib" ]; then export SCS_LIB_PATH=$PWD/sub-cortical_segmentation_amd/lib/libscs_$lib.so; else unset SCS_LIB_PATH; fi
  echo -n "[${lib:-default}] "
  timeout -k 10 200 python bench.py --no-cpu --no-train-block --steps 10 --warmup 2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); s=d['stages']
print(round(d['value']), ' '.join('%s %.3f/%.0f' % (k.split('+')[0][:6], v['ms_per_step'], v['tflops'] or 0) for k,v in s.items()))"
done
