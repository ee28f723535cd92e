#!/bin/bash
# SQ counter passes over the forward bench (builder tool): vector / matrix instruction counts per inference kernel
#   bash tools/pmc_fwd_sq.sh   ->  gpurun_out/r04_pmc_sq_forward.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT/sqf_csv
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/sqf_$i -- python3 $ROOT/bench.py --no-cpu --no-sustained --no-dice --no-train-block --steps 2 --warmup 1 > $OUT/sqf_$i.log 2>&1 || { tail -5 $OUT/sqf_$i.log; }
  for f in $(find $OUT/sqf_$i -name "*counter_collection.csv"); do cp $f $OUT/sqf_csv/$i.csv; done
  i=$((i+1))
done
python3 $ROOT/tools/pmc_summary.py "$OUT/sqf_csv/*.csv" $OUT/r04_pmc_sq_forward.json > $OUT/r04_pmc_sq_forward.txt
rm -rf $OUT/sqf_0 $OUT/sqf_1 $OUT/sqf_2 $OUT/sqf_csv
