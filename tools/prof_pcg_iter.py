"""PCG iteration of the 1.0 M-DoF system of bench.py's fem record: wall per iteration over 400 fixed iterations, and the
event profile of the two kernels"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
slm = bench.build_fem_system(708, 200000)
slm._assemble(0, 1, 1)
sl, cl = slm.relative_lambda_trace(1.0, -1.0)
_lib.check(lib.fb_sys_form(ctx, slm._sys, sl, cl))
rr = C.c_double()
_lib.check(lib.fb_sys_solve_fixed(ctx, slm._sys, 64, C.byref(rr))); _lib.check(lib.fb_sync(ctx))
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    _lib.check(lib.fb_sys_solve_fixed(ctx, slm._sys, 400, C.byref(rr))); _lib.check(lib.fb_sync(ctx))
    best = min(best, time.perf_counter() - t0)
print(f'{best / 400 * 1e6:.2f} us per iteration ({400 / best:.0f} it/s), relres {rr.value:.2e}')
_lib.check(lib.fb_prof_reset(ctx)); _lib.check(lib.fb_prof_enable(ctx, 1))
_lib.check(lib.fb_sys_solve_fixed(ctx, slm._sys, 200, C.byref(rr)))
_lib.check(lib.fb_prof_enable(ctx, 0))
for k, v in _lib.prof_snapshot().items():
    print(f'   {k:24s} launches {v[0]:4d}  {1e3 * v[1] / max(v[0], 1):8.2f} us each')
