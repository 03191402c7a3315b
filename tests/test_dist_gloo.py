"""N > 1 host logic on CPU: world_size-2 gloo run of the sharding + match-table gather."""
import os
import socket

import numpy as np
import pytest

from feabas_amd import dist as fdist


def test_shard_range_partitions():
    for n in (0, 1, 7, 48640, 1024):
        for world in (1, 2, 3, 8):
            spans = [fdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    n_pairs = 11
    a, b = fdist.shard_range(n_pairs, rank, world)
    rng = np.random.default_rng(100 + rank)
    # ragged local tables: pair p has (p % 4) + 1 matches
    pid = np.concatenate([np.full((p % 4) + 1, p - a) for p in range(a, b)])
    xy0 = rng.standard_normal((pid.size, 2)); xy1 = xy0 + 0.25; w = rng.uniform(0.3, 1, pid.size)
    table = fdist.gather_match_table(pid, xy0, xy1, w, pair_offset=a)
    parts = fdist.allgather_ragged(np.arange(3 + rank, dtype=np.float64).reshape(-1, 1) + 10 * rank)
    np.savez(os.path.join(outdir, f'r{rank}.npz'), table=table, local=np.concatenate((pid[:, None] + a, xy0, xy1, w[:, None]), axis=1),
             p0=parts[0], p1=parts[1])
    dist.barrier()
    dist.destroy_process_group()


def test_gather_match_table_world2(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / 'r0.npz'); r1 = np.load(tmp_path / 'r1.npz')
    np.testing.assert_array_equal(r0['table'], r1['table'])                  # every rank holds the full table
    expect = np.concatenate((r0['local'], r1['local']), axis=0)
    np.testing.assert_allclose(r0['table'], expect)
    assert np.all(np.diff(r0['table'][:, 0]) >= 0)                            # ordered by global pair id
    assert sorted(set(r0['table'][:, 0].astype(int))) == list(range(11))
    np.testing.assert_array_equal(r0['p0'].ravel(), [0, 1, 2]); np.testing.assert_array_equal(r0['p1'].ravel(), [10, 11, 12, 13])
