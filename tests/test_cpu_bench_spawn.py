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
    # every sharded leg of the bench through the helpers the real run calls (run_steps_threaded, match_table,
    # gather_table_timed, allgather_timed): one gather per step in call order although the calls finish out of order on four
    # host threads, every row of every rank on the root, equal blocks all-gathered rank by rank, rates summed
    assert d['headline']['ok'] and d['headline']['gather_calls'] == 4 and d['headline']['rows_on_root'] == d['headline']['rows_sent_by_all_ranks'] > 0
    assert d['stitch_sections']['ok'] and d['align_sections']['ok'] and d['fem']['ok']


def test_three_ranks_and_one_rank_take_the_same_code():
    r = _run('--gpus', '3', '--dry-run', '--stitch-sections', '2', '--align-sections', '2')
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert d['ranks'] == 3 and d['pair_shards_cover_the_list'] and d['allgather_shape'] == [3, 2, 4, 2]
    assert d['headline']['ok'] and d['stitch_sections']['ok'] and d['align_sections']['ok'] and d['fem']['ok']
    r = _run('--dry-run')
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert d['ranks'] == 1 and d['exchange_backend'] == 'none' and d['headline']['ok'] and d['fem']['ok']


def test_a_failing_matcher_call_stops_the_scheduler_instead_of_hanging():
    """run_steps_threaded: an exception in one worker thread ends the run (the exchange thread must not wait for the result
    that never comes) and is re-raised"""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module('bench')
    import pytest
    seen = []

    def step(i):
        if i == 5:
            raise RuntimeError('call 5 failed')
        return 'LR', i, {'n': i}

    def exchange(batch):
        seen.append([i for i, _ in batch])
    with pytest.raises(RuntimeError, match='call 5 failed'):
        bench.run_steps_threaded(range(12), step, exchange, 4, 3, [None], lambda h: None, 0.0)
    assert all(b == sorted(b) for b in seen) and (not seen or seen[0] == [0, 1, 2, 3])
    # the sequential form and the threaded form hand the same batches over, in the same order
    a, b = [], []
    ok = lambda i: ('LR', i, {'n': i})
    bench.run_steps_threaded(range(3, 13), ok, lambda batch: a.append([i for i, _ in batch]), 4, 1, [None], lambda h: None, 0.0)
    bench.run_steps_threaded(range(3, 13), ok, lambda batch: b.append([i for i, _ in batch]), 4, 4, [None], lambda h: None, 0.0)
    assert a == b == [[3, 4, 5, 6], [7, 8, 9, 10], [11, 12]]


def test_more_ranks_than_gpus_is_an_error_not_a_silent_single_rank():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip('this host has the GPUs')
    r = _run('--gpus', '2')
    assert r.returncode != 0
    assert 'GPU(s) are visible' in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
