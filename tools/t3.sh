python -m pytest tests/test_gpu_train.py -x -q -k "fused_adam or ragged or determin or stream_order" > gpurun_out/r04_t3_tests.log 2>&1; tail -3 gpurun_out/r04_t3_tests.log
for b in 256 128; do python bench.py --mode train --batch $b --steps 300 --warmup 30 --no-cpu 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['per_gpu_batch'], round(d['ms_per_step'],4), round(d['net_frac_of_f32_peak'],4), d['launches_per_step'], d['sustained']['ratio_to_burst'])"; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_trn -- python3 $R/bench.py --mode train --no-cpu --no-sustained --steps 60 --warmup 10 > $R/gpurun_out/prof_trn.log 2>&1
DB=$(find $R/gpurun_out/prof_trn -name "*.db" | head -1)
python3 $R/tools/rocpd_timeline.py $DB > $R/gpurun_out/r04_t3_timeline.txt
rm -rf $R/gpurun_out/prof_trn
