#!/bin/bash
# forward throughput of library variants (lib/libscs_<name>.so; "shipped" = libscs.so) on ONE GPU, two rounds
for round in 1 2; do
for lib in "$@"; do
  if [ "$lib" != "shipped" ]; then export SCS_LIB_PATH=$PWD/sub-cortical_segmentation_amd/lib/libscs_$lib.so; else unset SCS_LIB_PATH; fi
  f=$(timeout -k 10 200 python bench.py --no-cpu --no-train-block --steps 10 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['roofline']['frac'],4))")
  echo "[$lib] forward $f"
done; done
