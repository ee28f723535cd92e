#!/bin/bash
# Everything profiles/ of a round is made from, on one MI355X:  bash tools/profile_round.sh r05 [all | bench | prof]
#   (`bench`: the bench lines; `prof`: the rocprofv3 passes - two gpurun calls where one would pass the call's time limit)
#   bench lines (default = forward + train_step block, train, volume, cohort in three forms), test_scan per stage, rocprofv3 kernel traces of the forward and of the
#   training run (per-kernel statistics, a one-step launch timeline), and the HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE in
#   separate runs, no trace domains beside them).  Results land in gpurun_out/; tools/pmc_traffic.py then writes
#   profiles/traffic.json against gpurun_out/csrc_sha16.txt.
set -o pipefail
R=${1:-r06}
PART=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python3 -c "import bench; print(bench.csrc_sha16())" > $OUT/csrc_sha16.txt
if [ "$PART" != "prof" ]; then
echo "== bench default"; timeout -k 10 400 python3 bench.py > $OUT/${R}_bench_default.json.log 2> $OUT/${R}_bench_default.err || exit 1
echo "== bench train";   timeout -k 10 300 python3 bench.py --mode train --steps 300 --warmup 30 > $OUT/${R}_bench_train.json.log 2>/dev/null || exit 1
echo "== bench train, option train_split"; timeout -k 10 300 python3 bench.py --mode train --dtype f16x2 --steps 300 --warmup 30 --no-cpu > $OUT/${R}_bench_train_split.json.log 2>/dev/null || exit 1
echo "== bench forward, split kernels as the line's value"; timeout -k 10 300 python3 bench.py --dtype f16x2 --no-cpu --no-train-block --no-dice > $OUT/${R}_bench_forward_split.json.log 2>/dev/null || exit 1
echo "== bench volume";  timeout -k 10 300 python3 bench.py --mode volume > $OUT/${R}_bench_volume.json.log 2>/dev/null || exit 1
echo "== bench cohort";  timeout -k 10 500 python3 bench.py --mode cohort --subjects 10 > $OUT/${R}_bench_cohort.json.log 2>/dev/null || exit 1
echo "== bench cohort, towers per voxel (the reference's procedure)"
timeout -k 10 500 python3 bench.py --mode cohort --subjects 10 --towers patch --no-cpu > $OUT/${R}_bench_cohort_patch_towers.json.log 2>/dev/null || exit 1
echo "== bench cohort, four processes on the GPU"
timeout -k 10 500 python3 bench.py --mode cohort --subjects 10 --procs-per-gpu 4 --no-cpu > $OUT/${R}_bench_cohort_4procs.json.log 2>/dev/null || exit 1
echo "== test_scan, one subject, per stage"
timeout -k 10 300 python3 tools/scan_cpu_profile.py 2>&1 | grep -v amdgpu.ids > $OUT/${R}_scan_cpu_profile.txt || exit 1
fi
[ "$PART" = "bench" ] && { echo "== done (bench part)"; exit 0; }
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace forward"
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/prof_fwd -- python3 $ROOT/bench.py --no-cpu --no-train-block --no-sustained --no-dice --steps 10 --warmup 2 > $OUT/prof_fwd.log 2>&1 || exit 1
python3 $ROOT/tools/rocpd_stats.py $(find $OUT/prof_fwd -name "*.db" | head -1) > $OUT/${R}_forward_kernel_stats.csv || exit 1
echo "== kernel trace train"
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/prof_trn -- python3 $ROOT/bench.py --mode train --no-cpu --no-sustained --steps 60 --warmup 10 > $OUT/prof_trn.log 2>&1 || exit 1
DB=$(find $OUT/prof_trn -name "*.db" | head -1)
python3 $ROOT/tools/rocpd_stats.py $DB > $OUT/${R}_train_kernel_stats.csv || exit 1
python3 $ROOT/tools/rocpd_timeline.py $DB > $OUT/${R}_train_timeline.txt || exit 1
echo "== kernel trace train, option train_split"
timeout -k 10 400 rocprofv3 --kernel-trace -d $OUT/prof_trs -- python3 $ROOT/bench.py --mode train --dtype f16x2 --no-cpu --no-sustained --steps 60 --warmup 10 > $OUT/prof_trs.log 2>&1 || exit 1
python3 $ROOT/tools/rocpd_stats.py $(find $OUT/prof_trs -name "*.db" | head -1) > $OUT/${R}_train_split_kernel_stats.csv || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c forward"
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_fwd_$c -- python3 $ROOT/bench.py --no-cpu --no-train-block --no-sustained --no-dice --steps 3 --warmup 1 > $OUT/pmc_fwd_$c.log 2>&1 || exit 1
  echo "== pmc $c train"
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_trn_$c -- python3 $ROOT/bench.py --mode train --no-cpu --no-sustained --steps 20 --warmup 5 > $OUT/pmc_trn_$c.log 2>&1 || exit 1
  echo "== pmc $c train, option train_split"
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_trs_$c -- python3 $ROOT/bench.py --mode train --dtype f16x2 --no-cpu --no-sustained --steps 20 --warmup 5 > $OUT/pmc_trs_$c.log 2>&1 || exit 1
done
mkdir -p $OUT/pmc_fwd_csv $OUT/pmc_trn_csv
i=0; for f in $(find $OUT/pmc_fwd_FETCH_SIZE $OUT/pmc_fwd_WRITE_SIZE -name "*counter_collection.csv"); do cp $f $OUT/pmc_fwd_csv/$i.csv; i=$((i+1)); done
i=0; for f in $(find $OUT/pmc_trn_FETCH_SIZE $OUT/pmc_trn_WRITE_SIZE $OUT/pmc_trs_FETCH_SIZE $OUT/pmc_trs_WRITE_SIZE -name "*counter_collection.csv"); do cp $f $OUT/pmc_trn_csv/$i.csv; i=$((i+1)); done
python3 $ROOT/tools/pmc_summary.py "$OUT/pmc_fwd_csv/*.csv" $OUT/${R}_pmc_forward.json > $OUT/${R}_pmc_forward.txt || exit 1
python3 $ROOT/tools/pmc_summary.py "$OUT/pmc_trn_csv/*.csv" $OUT/${R}_pmc_train.json > $OUT/${R}_pmc_train.txt || exit 1
rm -rf $OUT/prof_fwd $OUT/prof_trn $OUT/prof_trs $OUT/pmc_trs_FETCH_SIZE $OUT/pmc_trs_WRITE_SIZE $OUT/pmc_fwd_FETCH_SIZE $OUT/pmc_fwd_WRITE_SIZE $OUT/pmc_trn_FETCH_SIZE $OUT/pmc_trn_WRITE_SIZE $OUT/pmc_fwd_csv $OUT/pmc_trn_csv
echo "== done"; ls -la $OUT
