#!/bin/bash
# SQ counter passes over a stand-alone binary (builder tool):  bash tools/pmc_bin.sh <name> <binary> [args...]  ->  gpurun_out/<name>_pmc_sq.txt
set -o pipefail
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT/sqb_csv
BIN=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_LDS"; do
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/sqb_$i -- $BIN "$@" > $OUT/sqb_$i.log 2>&1 || { tail -5 $OUT/sqb_$i.log; }
  for f in $(find $OUT/sqb_$i -name "*counter_collection.csv"); do cp $f $OUT/sqb_csv/$i.csv; done
  i=$((i+1))
done
python3 $ROOT/tools/pmc_summary.py "$OUT/sqb_csv/*.csv" $OUT/${NAME}_pmc_sq.json > $OUT/${NAME}_pmc_sq.txt
rm -rf $OUT/sqb_0 $OUT/sqb_1 $OUT/sqb_2 $OUT/sqb_3 $OUT/sqb_csv $OUT/sqb_*.log
