"""The RCCL exchange steps behind the C ABI (fb_comm_* / fb_gatherv_dev / fb_allgather_dev / fb_allreduce_f64_dev) on the one
GPU of the test box: a communicator of one rank exercises library loading, communicator creation and every entry point
(counts, own-part copy of the gatherv, all-gather, all-reduce).  More ranks need more GPUs: RCCL refuses two ranks on one
device; the N-rank logic of the same class runs over gloo in tests/test_dist_gloo.py."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_exchange_one_rank(fb):
    import ctypes as C
    import torch.distributed as dist
    from feabas_amd import _lib, dist as fdist
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        ex = fdist.Exchange(backend='rccl')
        assert ex.backend == 'rccl' and ex.comm is not None
        rank, world = C.c_int(-1), C.c_int(-1)
        _lib.check(_lib.load().fb_comm_info(_lib.ctx(), ex.comm, C.byref(rank), C.byref(world)))
        assert (rank.value, world.value) == (0, 1)
        tab = np.arange(60, dtype=np.float32).reshape(10, 6)
        parts = ex.gatherv(tab, root=0)
        assert len(parts) == 1 and np.array_equal(parts[0], tab)
        assert ex.gatherv(np.zeros((0, 6), np.float32), root=0)[0].shape == (0, 6)
        x = np.random.default_rng(0).standard_normal((1000, 2))
        np.testing.assert_array_equal(ex.allgather(x), x[None])
        # all-reduce of the three scalars of the coupled-window PCG
        lib, ctx = _lib.load(), _lib.ctx()
        d = _lib.DeviceBuffer.from_array(np.array([1.5, -2.0, 3.25])); o = _lib.DeviceBuffer(24)
        _lib.check(lib.fb_allreduce_f64_dev(ctx, ex.comm, d.ptr, o.ptr, 3, 0))
        np.testing.assert_array_equal(o.to_array((3,), np.float64), [1.5, -2.0, 3.25])
        assert lib.fb_allreduce_f64_dev(ctx, ex.comm, d.ptr, o.ptr, 3, 7) == -1          # unknown reduction: FB_ERR_ARG
        d.free(); o.free()
        ex.close()
    finally:
        dist.destroy_process_group()


def _one_rank_group():
    import torch.distributed as dist
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=0, world_size=1)
    return dist


def test_sendrecv_and_the_fused_loop_with_a_self_halo(fb):
    """fb_sendrecv_dev (grouped ncclSend / ncclRecv) and fb_cgcg_solve_dev with a real exchange on ONE rank: the rank sends to
    and receives from itself.  The system is A = A_own + A_halo S with S picking the rows the "neighbour" would own, stored
    as [A_own | A_halo]; every iteration packs u[S] (fb_gather_f64_dev), moves it through RCCL into the halo part of the
    extended vector and all-reduces the three scalars.  The solution equals the direct solve of A."""
    import ctypes as C
    from scipy import sparse
    from scipy.sparse.linalg import spsolve
    from feabas_amd import _lib, dist as fdist
    dist = _one_rank_group()
    try:
        lib, ctx = _lib.load(), _lib.ctx()
        ex = fdist.Exchange(backend='rccl', ctx=ctx)
        # --- plain send / receive to self: two messages in one group
        a = np.arange(1000, dtype=np.float64); b = -np.arange(37, dtype=np.float64)
        da, db = _lib.DeviceBuffer.from_array(a), _lib.DeviceBuffer.from_array(b)
        ra, rb = _lib.DeviceBuffer(a.nbytes), _lib.DeviceBuffer(b.nbytes)
        peers = np.zeros(2, dtype=np.int32)
        sp = (C.c_void_p * 2)(da.ptr.value, db.ptr.value); rp = (C.c_void_p * 2)(ra.ptr.value, rb.ptr.value)
        nb = np.array([a.nbytes, b.nbytes], dtype=np.int64)
        _lib.check(lib.fb_sendrecv_dev(ctx, ex.comm, 2, _lib.ptr(peers), sp, _lib.ptr(nb), 2, _lib.ptr(peers), rp, _lib.ptr(nb)))
        np.testing.assert_array_equal(ra.to_array(a.shape, np.float64), a); np.testing.assert_array_equal(rb.to_array(b.shape, np.float64), b)
        for d in (da, db, ra, rb):
            d.free()
        # --- the fused loop with a self halo
        rng = np.random.default_rng(5)
        n, nh = 600, 80
        L = sparse.diags([-1.0, 2.9, -1.0], [-2, 0, 2], shape=(n, n)) + sparse.diags([-0.25, -0.25], [-1, 1], shape=(n, n))
        sel = np.sort(rng.choice(n, nh, replace=False)).astype(np.int32)                 # rows whose values arrive "from the neighbour"
        H = sparse.random(n, nh, density=0.02, random_state=7, format='csr') * 0.05
        S = sparse.csr_matrix((np.ones(nh), (np.arange(nh), sel)), shape=(nh, n))
        Hs = (H @ S); Hs = Hs + Hs.T                                                      # keep A symmetric: couple through S both ways
        A = (L + Hs).tocsr()
        # [A_own | A_halo]: the part of Hs that goes through the halo columns is H (the transposed half stays in own columns)
        own = (L + (H @ S).T).tocsr()
        ext = sparse.hstack((own, H)).tocsr(); ext.sort_indices()
        assert abs((own + H @ S) - A).max() < 1e-15

        class _Part:
            n_loc, n_halo, world = n, nh, 1
            local_cols = ext.indices.astype(np.int64)
            def lists(self):
                return (np.zeros(1, np.int32), np.array([0, nh], np.int64), sel, np.zeros(1, np.int32), np.array([0, nh], np.int64))
        part = _Part()
        rows = fdist.DeviceRows(part, ext.indptr, ext.data)
        bvec = rng.standard_normal(n)
        x, it, rel = fdist.pcg_row_partitioned_dev(part, rows, bvec, 1.0 / A.diagonal(), rtol=1e-11, maxiter=4000, comm=ex.comm)
        rows.free()
        exact = spsolve(A.tocsc(), bvec)
        assert rel <= 1e-11 and 5 < it < 4000
        np.testing.assert_allclose(x, exact, atol=1e-8 * np.abs(exact).max())
        assert ex.rccl_ranks() == 1
        ex.close()
    finally:
        dist.destroy_process_group()


def test_coupled_window_on_a_one_rank_communicator_vs_oracle(fb):
    """SLM.optimize_linear(distributed=...) of the 5-section alignment window (one locked, four free: aligner.py:696-727) with
    the whole fused loop of fb_cgcg_solve_dev on a 1-rank RCCL communicator (FEABAS_HIP_EXCHANGE=rccl: the scalars go through
    fb_allreduce_f64_dev every iteration) against the ORACLE's exact solve of the window, oracle/fem_ref.optimize_linear"""
    from feabas_amd import optimizer, dist as fdist
    from oracle import fem_ref
    from test_dist_gloo import _window
    dist = _one_rank_group()
    os.environ['FEABAS_HIP_EXCHANGE'] = 'rccl'
    try:
        meshes, links = _window()
        before = [m.vertices_w_offset(1).copy() for m in meshes]
        rms = [fem_ref.RefMesh(m.vertices(-1).copy(), m.triangles, uid=m.uid, locked=m.locked) for m in meshes]
        by = {m.uid: r for m, r in zip(meshes, rms)}
        rls = [fem_ref.RefLink(by[lk.meshes[0].uid], by[lk.meshes[1].uid], lk._tid0, lk._tid1, lk._B0, lk._B1, weight=lk._weight) for lk in links]
        ref_cost = fem_ref.optimize_linear(rms, rls, exact=True)
        slm = optimizer.SLM(meshes, links, stiffness_lambda=1.0, crosslink_lambda=-1.0)
        cost = slm.optimize_linear(tol=1e-10, distributed=True)
        assert slm.last_solve['exchange'] == 'rccl' and slm.last_solve['iters'] > 10
        assert abs(cost[0] - ref_cost[0]) <= 1e-6 * ref_cost[0]
        motion = max(np.abs(r.vertices_w_offset(fem_ref.GEAR_MOVING) - v0).max() for r, v0 in zip(rms, before))
        assert motion > 0.5
        for m, r in zip(meshes[1:], rms[1:]):
            np.testing.assert_allclose(m.vertices_w_offset(1), r.vertices_w_offset(fem_ref.GEAR_MOVING), atol=1e-6 * motion)
        fdist.release_exchanges()
    finally:
        del os.environ['FEABAS_HIP_EXCHANGE']
        dist.destroy_process_group()
