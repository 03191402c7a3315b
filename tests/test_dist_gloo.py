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
    table = fdist.gather_match_table(pid, xy0, xy1, w, pair_offset=a, root=None)          # on every rank
    root_only = fdist.gather_match_table(pid, xy0, xy1, w, pair_offset=a, root=world - 1, dtype=np.float32)    # one gather to one rank, wire precision
    assert (root_only is not None) == (rank == world - 1)
    if root_only is not None:
        assert root_only.dtype == np.float32
        np.testing.assert_allclose(root_only, table, rtol=1e-6, atol=1e-6)
    parts = fdist.allgather_ragged(np.arange(3 + rank, dtype=np.float64).reshape(-1, 1) + 10 * rank)
    eq = fdist.exchange().allgather(np.full((2, 3), float(rank)))
    assert eq.shape == (world, 2, 3) and all(np.all(eq[r] == r) for r in range(world))
    empty = fdist.exchange().gatherv(np.zeros((0, 6)) if rank == 0 else np.ones((rank, 6)), root=0)      # an empty contribution
    if rank == 0:
        assert [p.shape[0] for p in empty] == list(range(world))
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


def _coupled_system(n_sections=4, nv=150, seed=0):
    """a chain of `n_sections` sections (2-D elastic-like SPD blocks of nv vertices = 2 nv DoF each) coupled to their
    neighbours by a few links, like an alignment window (aligner.py:510-535): returns scipy CSR A (SPD) and b"""
    from scipy import sparse
    rng = np.random.default_rng(seed)
    n = 2 * nv
    blocks = []
    for s in range(n_sections):
        L = sparse.diags([-1.0, 2.8, -1.0], [-2, 0, 2], shape=(n, n)) + sparse.diags([-0.3, -0.3], [-1, 1], shape=(n, n))
        blocks.append(L * rng.uniform(0.8, 1.2))
    A = sparse.block_diag(blocks, format='lil')
    N = n * n_sections
    for s in range(n_sections - 1):                                  # links: w (u_a - u_b)^2 on both coordinates of a vertex pair
        for _ in range(25):
            va, vb = rng.integers(0, nv, 2)
            w = rng.uniform(0.3, 1.0)
            for c in range(2):
                i, j = s * n + 2 * va + c, (s + 1) * n + 2 * vb + c
                A[i, i] += w; A[j, j] += w; A[i, j] -= w; A[j, i] -= w
    A = A.tocsr()
    b = rng.standard_normal(N)
    return A, b


def _pcg_worker(rank, world, port, outdir, use_gpu):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    A, b = _coupled_system()
    n_sec = 4
    sec0, sec1 = fdist.shard_range(n_sec, rank, world)               # whole sections per rank
    per = A.shape[0] // n_sec
    r0, r1 = sec0 * per, sec1 * per
    rows = A[r0:r1].tocsr()
    part = fdist.RowPartition(rows.indptr, rows.indices, r0)
    b_loc = b[r0:r1].copy()
    minv = 1.0 / A.diagonal()[r0:r1]
    if use_gpu:
        # the rows on the GPU, the vectors in the library's buffers; the ranks share the one GPU of the box, so there is no RCCL
        # communicator (solver_comm -> None) and halo + scalars travel over the gloo group: every kernel of the loop runs
        dr = fdist.DeviceRows(part, rows.indptr, rows.data)
        comm = fdist.solver_comm()
        assert comm is None or world == 1
        x, it, rel = fdist.pcg_row_partitioned_dev(part, dr, b_loc, minv, rtol=1e-10, maxiter=5000, comm=comm)
        dr.free()
    else:
        from scipy import sparse
        loc = sparse.csr_matrix((rows.data, part.local_cols, rows.indptr), shape=(r1 - r0, part.n_loc + part.n_halo))
        x, it, rel = fdist.pcg_row_partitioned(part, lambda ext: loc @ ext, b_loc, minv, rtol=1e-10, maxiter=5000)
    np.savez(os.path.join(outdir, f'x{rank}.npz'), x=np.asarray(x), it=it, rel=rel, r0=r0, halo=part.n_halo)
    dist.barrier()
    dist.destroy_process_group()


def _check_pcg(tmp_path, world):
    from scipy.sparse.linalg import spsolve
    A, b = _coupled_system()
    exact = spsolve(A.tocsc(), b)
    parts = [np.load(tmp_path / f'x{r}.npz') for r in range(world)]
    x = np.concatenate([p['x'] for p in parts])
    assert all(int(p['it']) == int(parts[0]['it']) for p in parts) and float(parts[0]['rel']) <= 1e-10
    if world > 1:
        assert all(int(p['halo']) > 0 for p in parts)
    np.testing.assert_allclose(x, exact, atol=1e-7 * np.abs(exact).max())


@pytest.mark.parametrize('world', [1, 2, 3])
def test_coupled_window_pcg_gloo(tmp_path, world):
    """the row-partitioned PCG of the coupled alignment window (halo exchange + one fused all-reduce per iteration) on
    CPU tensors over gloo, local products by scipy (injected: the product path applies the device SpMV): the assembled
    solution equals the direct solve of the global system"""
    import torch.multiprocessing as mp
    mp.spawn(_pcg_worker, args=(world, _free_port(), str(tmp_path), False), nprocs=world, join=True)
    _check_pcg(tmp_path, world)


@pytest.mark.gpu
def test_coupled_window_pcg_device_rows_world2(tmp_path):
    """the same with the local rows and every vector on the GPU (DeviceRows + the fb_cgcg_* kernels on fb_malloc buffers), two
    processes sharing the one GPU of the box, scalars and halos over gloo (RCCL refuses two ranks on one device)"""
    import torch.multiprocessing as mp
    mp.spawn(_pcg_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)
    _check_pcg(tmp_path, 2)


def _window(n_sections=5, seed=3):
    """an alignment window: a locked first section and four free ones (grid meshes of different sizes, smoothly displaced
    relative to each other), every section linked to the next by 300 matches"""
    from feabas_amd import mesh, optimizer
    from oracle import fem_ref
    rng = np.random.default_rng(seed)
    meshes = []
    for s in range(n_sections):
        nx, ny = 22 + 2 * (s % 3), 18 + (s % 2)
        v, t = fem_ref.grid_mesh(nx, ny, 10.0)
        L = 10.0 * nx
        d = np.stack((1.5 * np.sin(2 * np.pi * v[:, 1] / L + 0.7 * s), 1.2 * np.cos(2 * np.pi * v[:, 0] / L - 0.4 * s)), axis=-1)
        meshes.append(mesh.Mesh(v + d, t, uid=float(s), locked=(s == 0)))
    links = []
    for s in range(n_sections - 1):
        m0, m1 = meshes[s], meshes[s + 1]
        n = 300
        xy = np.stack((rng.uniform(5, 200, n), rng.uniform(5, 160, n)), axis=-1)
        rel = np.stack((1.5 * np.sin(xy[:, 1] / 40.0 + s), 1.0 * np.cos(xy[:, 0] / 50.0 - s)), axis=-1)       # what the next section must follow
        lk, _ = optimizer.Link.from_coordinates(m0, m1, xy, xy + rel + rng.normal(0, 0.05, xy.shape), weight=rng.uniform(0.3, 1, n).astype(np.float32))
        links.append(lk)
    return meshes, links


def _window_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from feabas_amd import optimizer
    meshes, links = _window()
    slm = optimizer.SLM(meshes, links, stiffness_lambda=1.0, crosslink_lambda=-1.0)
    cost = slm.optimize_linear(tol=1e-10, distributed=True)
    a, b = fdist.shard_range(4, rank, world)
    out = {f'v{m.uid:.0f}': m.vertices_w_offset(1) for m in meshes[1:][a:b]}
    np.savez(os.path.join(outdir, f'w{rank}.npz'), cost=np.array(cost), iters=slm.last_solve['iters'], halo=slm.last_solve['halo'], **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('world', [1, 2, 3])
def test_optimize_linear_distributed_equals_single_gpu(tmp_path, world):
    """SLM.optimize_linear(distributed=group): the coupled window (aligner.py:696-727) with its rows partitioned by section
    over 1 / 2 / 3 ranks (processes sharing the GPU, gloo between them): device assembly of every rank's rows, halo exchange,
    fused CG kernels -- the sections end where the single-GPU solve of the whole window puts them"""
    import torch.multiprocessing as mp
    from feabas_amd import optimizer
    meshes, links = _window()
    v_before = [m.vertices_w_offset(1).copy() for m in meshes]
    ref_cost = optimizer.SLM(meshes, links, stiffness_lambda=1.0, crosslink_lambda=-1.0).optimize_linear(tol=1e-10)
    motion = max(np.abs(m.vertices_w_offset(1) - v0).max() for m, v0 in zip(meshes, v_before))
    assert motion > 0.5
    mp.spawn(_window_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = {}
    for r in range(world):
        d = np.load(tmp_path / f'w{r}.npz')
        got.update({k: d[k] for k in d.files if k.startswith('v')})
        assert abs(d['cost'][0] - ref_cost[0]) <= 1e-9 * ref_cost[0]              # the same ||b||: same lambdas, same assembly
        assert d['cost'][1] <= 1e-10 * d['cost'][0] * 1.01
        assert (int(d['halo']) > 0) == (world > 1)
    assert sorted(got) == ['v1', 'v2', 'v3', 'v4']
    for m in meshes[1:]:
        np.testing.assert_allclose(got[f'v{m.uid:.0f}'], m.vertices_w_offset(1), atol=1e-6 * motion)


_GROUPS = [0, 0, 1, 1, 2, 3, 3, 4]


def _grouped_window(seed=5):
    """eight sections in five groups whose members share their degrees of freedom (optimizer.py:1378-1415): {0 locked, 1} -- held as
    a whole, section 1 is free but does not move --, {2, 3}, {4}, {5, 6}, {7}; a chain of links, plus one from section 2 to section 6"""
    from feabas_amd import mesh, optimizer
    from oracle import fem_ref
    rng = np.random.default_rng(seed)
    meshes = []
    for s, g in enumerate(_GROUPS):
        nx, ny = 20 + 2 * (g % 3), 17 + (g % 2)
        v, t = fem_ref.grid_mesh(nx, ny, 10.0)
        L = 10.0 * nx
        d = np.stack((1.5 * np.sin(2 * np.pi * v[:, 1] / L + 0.7 * g), 1.2 * np.cos(2 * np.pi * v[:, 0] / L - 0.4 * g)), axis=-1)
        meshes.append(mesh.Mesh(v + d, t, uid=float(s), locked=(s == 0)))
    links = []
    for a, b in [(s, s + 1) for s in range(len(_GROUPS) - 1)] + [(2, 6)]:
        n = 250
        xy = np.stack((rng.uniform(5, 180, n), rng.uniform(5, 150, n)), axis=-1)
        rel = np.stack((1.5 * np.sin(xy[:, 1] / 40.0 + a), 1.0 * np.cos(xy[:, 0] / 50.0 - b)), axis=-1)
        lk, _ = optimizer.Link.from_coordinates(meshes[a], meshes[b], xy, xy + rel + rng.normal(0, 0.05, xy.shape), weight=rng.uniform(0.3, 1, n).astype(np.float32))
        links.append(lk)
    return meshes, links


def _grouped_worker(rank, world, port, outdir, by_uid):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from feabas_amd import optimizer
    meshes, links = _grouped_window()
    slm = optimizer.SLM(meshes, links, stiffness_lambda=1.0, crosslink_lambda=-1.0)
    kw = {}
    if by_uid:                                      # ownership by uid: whole groups per rank, in an order that is not the list's
        split = {1: [[1, 2, 3, 4, 5, 6, 7]], 2: [[5, 6, 7, 1], [4, 2, 3]]}[world]      # (1: the free member of the held group, for the lambdas)
        kw['owned'] = [float(u) for u in split[rank]]
    cost = slm.optimize_linear(tol=1e-10, distributed=True, groupings=_GROUPS, **kw)
    np.savez(os.path.join(outdir, f'g{rank}.npz'), cost=np.array(cost), rows=slm.last_solve['rows'], **{f'v{m.uid:.0f}': m.vertices_w_offset(1) for m in meshes})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('world,by_uid', [(1, False), (2, False), (3, False), (2, True)])
def test_optimize_linear_distributed_with_groupings_equals_single_gpu(tmp_path, world, by_uid):
    """optimize_linear(distributed=..., groupings=...): the unit of ownership is the group (its members share the rows); a group with
    a locked member stays where it is, its free member still counts in the trace-relative lambdas (mesh-level sums, optimizer.py:1573-1590).
    Same lambdas, costs and node positions as the single-GPU grouped solve, which goldens G19 / G21 pin to the reference"""
    import torch.multiprocessing as mp
    from feabas_amd import optimizer
    meshes, links = _grouped_window()
    v_before = [m.vertices_w_offset(1).copy() for m in meshes]
    ref_cost = optimizer.SLM(meshes, links, stiffness_lambda=1.0, crosslink_lambda=-1.0).optimize_linear(tol=1e-10, groupings=_GROUPS)
    moved = [np.abs(m.vertices_w_offset(1) - v0).max() for m, v0 in zip(meshes, v_before)]
    motion = max(moved)
    assert motion > 0.5 and moved[0] == 0 and moved[1] == 0
    np.testing.assert_allclose(meshes[2].vertices_w_offset(1) - v_before[2], meshes[3].vertices_w_offset(1) - v_before[3], atol=1e-12)
    mp.spawn(_grouped_worker, args=(world, _free_port(), str(tmp_path), by_uid), nprocs=world, join=True)
    rows = 0
    for r in range(world):
        d = np.load(tmp_path / f'g{r}.npz')
        rows += int(d['rows'])
        assert abs(d['cost'][0] - ref_cost[0]) <= 1e-9 * ref_cost[0]
        assert d['cost'][1] <= 1e-10 * d['cost'][0] * 1.01
        for k, (m, v0) in enumerate(zip(meshes, v_before)):
            got = d[f'v{m.uid:.0f}']
            if np.abs(got - v0).max() == 0 and moved[k] > 0:
                continue                              # a section of another rank: not moved here
            np.testing.assert_allclose(got, m.vertices_w_offset(1), atol=1e-6 * motion)
    free_dofs = sum(2 * meshes[k].num_vertices for k in (2, 4, 5, 7))      # one set of rows per free group
    assert rows == free_dofs
    seen = set()
    for r in range(world):
        d = np.load(tmp_path / f'g{r}.npz')
        seen |= {k for k, v0 in enumerate(v_before) if np.abs(d[f'v{k}'] - v0).max() > 0}
    assert seen == {2, 3, 4, 5, 6, 7}


def test_pcg_host_path_restarts_after_a_dropped_step():
    """a dropped step (non-positive denominator) used to make every later step a dropped one.  Now the step
    after it restarts the recurrence: an indefinite system is reported after `check_every` dropped steps instead of running
    to maxiter, and the scalar recurrence recovers from a single dropped step"""
    g, a, b_, drop = fdist._cg_scalars([4.0, -1.0, 9.0], 0.0, 0.0, True)            # delta <= 0 at the start: dropped
    assert (a, b_, drop) == (0.0, 0.0, True)
    g, a, b_, drop = fdist._cg_scalars([3.0, 2.0, 5.0], g, a, False)                # next step: restart, alpha = gamma / delta
    assert (a, b_, drop) == (1.5, 0.0, False)

    class _P:                                                                        # a one-rank partition without a process group
        n_loc, n_halo, world, group = 4, 0, 1, None
        def exchange(self, u): return np.zeros(0)
    A = np.diag([1.0, -2.0, 3.0, -4.0])                                              # indefinite
    with pytest.raises(FloatingPointError):
        fdist.pcg_row_partitioned(_P(), lambda e: A @ e, np.array([0.0, 1.0, 0.0, 1.0]), np.ones(4), rtol=1e-12, maxiter=100000, check_every=8)
    S = np.diag([1.0, 2.0, 3.0, 4.0])
    x, it, rel = fdist.pcg_row_partitioned(_P(), lambda e: S @ e, np.ones(4), 1.0 / np.diag(S), rtol=1e-12, maxiter=50)
    np.testing.assert_allclose(x, 1.0 / np.diag(S), rtol=1e-10)
