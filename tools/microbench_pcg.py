"""Jacobi-PCG iteration rate by system size: fb_pcg_fixed_iters on grid-mesh stiffness systems (run once with
FEABAS_HIP_PCG_GRAPH_NB=0 for the launch-by-launch loop)."""
import ctypes as C
import os
import sys
import time

import numpy as np
from scipy import sparse

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
import fem_ref                                                # noqa: E402
from feabas_amd import _lib                                   # noqa: E402


def main():
    lib, ctx = _lib.load(), _lib.ctx()
    rng = np.random.default_rng(0)
    print('graph max nb', os.environ.get('FEABAS_HIP_PCG_GRAPH_NB', 'default'))
    for side in (16, 32, 64, 118, 250, 354):
        v, t = fem_ref.grid_mesh(side, side, 10.0)
        K, _ = fem_ref.mesh_stiffness(v, None, t)
        n = K.shape[0]
        A = sparse.csr_matrix(K + sparse.diags(rng.uniform(0.01, 0.1, n))); A.sort_indices()
        h = C.c_void_p()
        ip = A.indptr.astype(np.int64); ix = A.indices.astype(np.int32); va = A.data.astype(np.float64)
        _lib.check(lib.fb_csr_upload(ctx, n, _lib.ptr(ip), _lib.ptr(ix), _lib.ptr(va), 1, C.byref(h)))
        b = A.dot(rng.standard_normal(n)); rr = C.c_double()
        iters = 960
        _lib.check(lib.fb_pcg_fixed_iters(ctx, h, _lib.ptr(b), 64, C.byref(rr)))
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            _lib.check(lib.fb_pcg_fixed_iters(ctx, h, None, iters, C.byref(rr)))
            best = min(best, time.perf_counter() - t0)
        print(f'nb {n // 2:7d}: {best / iters * 1e6:6.2f} us / iteration   (relres {rr.value:.2e})')
        lib.fb_csr_destroy(ctx, h)


if __name__ == '__main__':
    main()
