"""bench.py's output contract on the MI355X: one JSON line with the keys the driver reads, a `roofline` for the dominant
kernel (event-timed) and a `cpu_baseline`, in the forward and in the training mode."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
        'dtype', 'data', 'config', 'roofline', 'cpu_baseline'}


def run_bench(args):
    # (the line under test is the DEFAULT one: whatever forces an option form on the suite from outside stays outside)
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'SCS_FWD_SPLIT', 'SCS_TRAIN_SPLIT', 'SCS_SPLIT_HEAD')}
    env['SCS_BENCH_CPU_BUDGET'] = '2'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.parametrize('mode', ['forward', 'train'])
def test_bench_line(mode):
    out = run_bench(['--mode', mode, '--steps', '3', '--warmup', '1'] + (['--batch', '8192', '--train-steps', '20'] if mode == 'forward' else []))
    assert KEYS <= set(out), KEYS - set(out)
    assert out['n_gpus'] == 1 and out['steps'] == 3 and out['warmup'] == 1 and out['unit'] == 'patches/s'
    assert out['dtype'] == 'f32' and out['data'] == 'synthetic' and out['vs_baseline'] is None and out['scaling'] == 'weak'
    assert out['value'] > 1e4 and abs(out['ms_per_step'] * out['value'] / 1e3 - out['config']['per_gpu_batch']) < 1e-6 * out['config']['per_gpu_batch'] + 1
    rf = out['roofline']
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and rf['peak'] == 157.3 and 0.05 < rf['frac'] < 1.0
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9 and rf['launch_ms'] > 0
    cb = out['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and cb['unit'] == 'patches/s' and cb['sample']
    if mode == 'train':
        assert out['launches_per_step'] <= 30 and 'conv2 dgrad + layer-1 sums' in out['stages']
    else:
        # the default (forward) line also carries the training step: BASELINE.json's metric is "fwd & train step"
        ts = out['train_step']
        assert out['ranks_seen'] == 1
        for key, bsz in (('batch256', 256), ('batch128', 128)):
            blk = ts[key]
            assert blk['per_gpu_batch'] == bsz and blk['steps'] == 20 and blk['unit'] == 'patches/s'
            assert abs(blk['ms_per_step'] * blk['value'] / 1e3 - bsz) < 1.0
            assert 0.02 < blk['frac_of_f32_peak'] < 1.0 and blk['launches_per_step'] <= 30
            assert blk['roofline']['bound'] == 'mfma' and blk['roofline']['launch_ms'] > 0 and blk['stages']


@pytest.mark.parametrize('procs', [1, 2])
def test_cohort_line(procs):
    """`bench.py --mode cohort`: test_cohort end to end on synthetic 256x256x192 subjects (BASELINE configs[4]); the line states the
    host budget, which form of the tower evaluation ran, and - with several processes sharing the GPU - how many ranks answered."""
    out = run_bench(['--mode', 'cohort', '--subjects', '2', '--no-cpu', '--procs-per-gpu', str(procs)])
    assert out['unit'] == 'subjects/h' and out['n_gpus'] == 1 and out['procs_per_gpu'] == procs and out['ranks_seen'] == procs
    assert out['value'] > 2000 and out['steps'] == 2 and out['dtype'] == 'f32' and out['vs_baseline'] is None
    assert out['config']['towers'] == 'auto (ran as: shared)'            # a dilated-mask ROI fills its box densely
    h = out['host']
    assert 0 < h['gpu_seconds_per_subject'] < 0.2 and 0 < h['cpu_seconds_per_subject'] < 3.0
    assert h['reader_threads_per_rank'] >= 1 and h['writer_threads_per_rank'] >= 1 and h['usable_cpus'] >= 1
