"""bench.py --gpus N outside a launcher starts the N ranks itself (python -m torch.distributed.run as a child).  Driven
here without a GPU through --dry-run: the ranks meet over gloo, shard the config[3] / config[4] lists and run the
exchange steps (gatherv of a ragged match table to rank 0, all-gather of equal displacement blocks)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv):
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), capture_output=True, text=True, timeout=300, cwd=ROOT)


def test_gpus2_spawns_two_ranks_and_rank0_prints_one_line():
    r = _run('--gpus', '2', '--dry-run', '--stitch-sections', '1', '--align-sections', '3')
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['ranks'] == 2 and d['dry_run'] is True
    assert d['pair_shards_cover_the_list'] is True
    assert d['allgather_shape'] == [2, 3, 4, 2]
    assert d['exchange_backend'] == 'torch'


def test_more_ranks_than_gpus_is_an_error_not_a_silent_single_rank():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('this host has the GPUs')
    r = _run('--gpus', '2')
    assert r.returncode != 0
    assert 'GPU(s) are visible' in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
