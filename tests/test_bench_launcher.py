"""bench.py's own multi-rank launcher: `--gpus N` without a torchrun environment must start N ranks (never a silent
1-GPU run), and a world size that differs from --gpus must be an error.  Rehearsed on CPU with --dry-run (gloo)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, env_extra=None, drop=('WORLD_SIZE', 'RANK', 'LOCAL_RANK')):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.timeout(900)
def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    r = _run(['--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['dry_run'] is True and out['steps'] == 3 and out['warmup'] == 1
    assert out['value'] is None                                    # a rehearsal never reports a throughput


def test_world_size_mismatch_is_an_error():
    r = _run(['--gpus', '2', '--dry-run'], env_extra={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'}, drop=())
    assert r.returncode != 0
    assert 'refusing' in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
