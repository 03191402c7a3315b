"""the weakly pinned 1e6-DoF system of bench.py (hard_50_links) through the Jacobi-PCG and the multigrid-PCG"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
for nl in (50, 5000, 200000):
    slm = bench.build_fem_system(int(os.environ.get('GRID', 708)), nl, seed=2)
    slm._assemble(0, 1, 1)
    sl, cl = slm.relative_lambda_trace(1.0, -1.0)
    _lib.check(lib.fb_sys_form(ctx, slm._sys, sl, cl))
    n = 2 * slm._nv
    x = np.zeros(n); it = C.c_int(); rr = C.c_double()
    for pre in (1, 2, 3):
        for tol in (1e-4, 1e-7):
            x[:] = 0
            t0 = time.time()
            rc = lib.fb_sys_solve(ctx, slm._sys, _lib.ptr(x), 0, tol, 0.0, 400000, pre, C.byref(it), C.byref(rr))
            print(f'links {nl:6d} precond {pre} tol {tol:g}: {1e3*(time.time()-t0):8.1f} ms  iters {it.value:5d} relres {rr.value:.2e} rc {rc}')
    _lib.check(lib.fb_prof_reset(ctx)); _lib.check(lib.fb_prof_enable(ctx, 1))
    x[:] = 0
    lib.fb_sys_solve(ctx, slm._sys, _lib.ptr(x), 0, 1e-7, 0.0, 400000, 2, C.byref(it), C.byref(rr))
    _lib.check(lib.fb_prof_enable(ctx, 0))
    print({k: (v[0], round(v[1], 3)) for k, v in _lib.prof_snapshot().items()})
    del slm
