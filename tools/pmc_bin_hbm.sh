#!/bin/bash
# HBM traffic passes over a stand-alone binary (builder tool):  bash tools/pmc_bin_hbm.sh <name> <binary> [args...]  ->  gpurun_out/<name>_pmc_hbm.txt
# (FETCH_SIZE and WRITE_SIZE in separate passes, no trace domains beside them; tools/pmc_summary.py applies the gfx950 corrections)
set -o pipefail
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT/hb_csv
BIN=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $OUT/hb_$i -- $BIN "$@" > $OUT/hb_$i.log 2>&1 || { tail -5 $OUT/hb_$i.log; }
  for f in $(find $OUT/hb_$i -name "*counter_collection.csv"); do cp $f $OUT/hb_csv/$i.csv; done
  i=$((i+1))
done
python3 $ROOT/tools/pmc_summary.py "$OUT/hb_csv/*.csv" $OUT/${NAME}_pmc_hbm.json > $OUT/${NAME}_pmc_hbm.txt
rm -rf $OUT/hb_0 $OUT/hb_1 $OUT/hb_csv $OUT/hb_*.log
