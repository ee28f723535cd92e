#!/bin/bash
# A/B of library builds in ONE session on ONE GPU (devices differ by several per cent: never compare across gpurun calls):
#   tools/ab_libs.sh name1 name2 ...   (sub-cortical_segmentation_amd/lib/libscs_<name>.so; "" = the shipped libscs.so), two rounds
for round in 1 2; do
for lib in "$@"; do
  if [ "$lib" != "shipped" ]; then export SCS_LIB_PATH=$PWD/sub-cortical_segmentation_amd/lib/libscs_$lib.so; else unset SCS_LIB_PATH; fi
  f=$(timeout -k 10 200 python bench.py --no-cpu --no-train-block --steps 10 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['roofline']['frac'],4))")
  t=$(timeout -k 10 200 python bench.py --mode train --steps 300 --warmup 30 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4))")
  echo "[$lib] forward $f   train $t ms"
done; done
