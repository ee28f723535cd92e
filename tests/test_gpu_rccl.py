"""De-risking the first multi-GPU run on a one-GPU box: RCCL itself on a one-rank process group, and
bench.py's own rank launcher (spawn_ranks) on real GPU work with the gloo stand-in backend."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'SCS_DIST_BACKEND')}
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    env.update(extra)
    return env


@pytest.mark.timeout(900)
def test_rccl_one_rank_group_is_the_identity_on_the_real_arenas():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'rccl_one_rank_worker.py')], env=_clean_env(), capture_output=True,
                       text=True, timeout=800)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert 'RCCL_ONE_RANK_OK' in r.stdout


def _bench(args, **env):
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=_clean_env(SCS_BENCH_CPU_BUDGET='2', **env),
                          capture_output=True, text=True, timeout=800)


@pytest.mark.timeout(1800)
@pytest.mark.parametrize('mode', ['train', 'forward'])
def test_bench_gpus_2_through_its_own_launcher(mode):
    """`bench.py --gpus 2` with no torchrun environment: spawn_ranks starts two ranks (both on the one GPU of this box, gloo in
    place of RCCL), ONE line comes back, it says two ranks took part, and the forward line carries the train_step block."""
    args = ['--gpus', '2', '--mode', mode, '--steps', '3', '--warmup', '1', '--train-steps', '10']
    if mode == 'forward':
        args += ['--batch', '8192']
    r = _bench(args, SCS_DIST_BACKEND='gloo')
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks_seen'] == 2 and out['dist_backend'] == 'gloo'
    assert out['value'] > 1e4 and 'cpu_baseline' not in out
    per = out['config']['per_gpu_batch']
    assert abs(out['ms_per_step'] * out['value'] / 1e3 - 2 * per) < 1e-6 * per + 1          # whole-job value = both ranks' patches
    if mode == 'forward':
        ts = out['train_step']
        for key in ('batch256', 'batch128'):
            assert ts[key]['value'] > 1e3 and ts[key]['launches_per_step'] <= 30 and 0.0 < ts[key]['roofline']['frac'] < 1.0
            assert abs(ts[key]['ms_per_step'] * ts[key]['value'] / 1e3 - 2 * ts[key]['per_gpu_batch']) < 1.0


def test_bench_gpus_2_without_the_override_refuses_on_one_gpu():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('this box has two GPUs: --gpus 2 is a real RCCL run here')
    r = _bench(['--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0
    assert 'only 1 HIP device' in (r.stdout + r.stderr)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
