#!/bin/bash
# SQ counter passes over the training step (builder tool): where do the wave cycles of each kernel go?
#   bash tools/pmc_sq.sh   ->  gpurun_out/r04_pmc_sq_train.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT/sq_csv
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_LDS"; do
  echo "== $set"
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/sq_$i -- python3 $ROOT/bench.py --mode train --no-cpu --no-sustained --steps 20 --warmup 5 > $OUT/sq_$i.log 2>&1 || { tail -5 $OUT/sq_$i.log; }
  for f in $(find $OUT/sq_$i -name "*counter_collection.csv"); do cp $f $OUT/sq_csv/$i.csv; done
  i=$((i+1))
done
python3 $ROOT/tools/pmc_summary.py "$OUT/sq_csv/*.csv" $OUT/r04_pmc_sq_train.json > $OUT/r04_pmc_sq_train.txt
rm -rf $OUT/sq_0 $OUT/sq_1 $OUT/sq_2 $OUT/sq_3 $OUT/sq_csv
tail -5 $OUT/r04_pmc_sq_train.txt
