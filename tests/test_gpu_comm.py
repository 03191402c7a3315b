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
