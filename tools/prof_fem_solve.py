"""where the time of fem.solve_to_1e4_s goes: the same call sequence as bench.py::bench_fem with the pieces timed apart"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
slm = bench.build_fem_system(708, 200000)
slm._assemble(0, 1, 1)
sl, cl = slm.relative_lambda_trace(1.0, -1.0)
_lib.check(lib.fb_sys_form(ctx, slm._sys, sl, cl))
nv = C.c_int64(); nnzb = C.c_int64(); nl = C.c_int64()
_lib.check(lib.fb_sys_info(ctx, slm._sys, C.byref(nv), C.byref(nnzb), C.byref(nl)))
rr = C.c_double(); it = C.c_int()
_lib.check(lib.fb_sys_solve_fixed(ctx, slm._sys, 20, C.byref(rr))); _lib.check(lib.fb_sync(ctx))
for label, mk in (('fresh np.zeros', lambda: np.zeros(2 * nv.value)), ('touched np.ones', lambda: np.ones(2 * nv.value)), ('fresh np.zeros again', lambda: np.zeros(2 * nv.value))):
    x = mk()
    for rep in range(3):
        t0 = time.time()
        _lib.check(lib.fb_sys_solve(ctx, slm._sys, _lib.ptr(x), 0, 1e-4, 0.0, -1, 1, C.byref(it), C.byref(rr)))
        print(f'{label} call {rep}: {1e3 * (time.time() - t0):8.3f} ms  iters {it.value} relres {rr.value:.2e}')
_lib.check(lib.fb_prof_reset(ctx)); _lib.check(lib.fb_prof_enable(ctx, 1))
x = np.ones(2 * nv.value)
t0 = time.time()
_lib.check(lib.fb_sys_solve(ctx, slm._sys, _lib.ptr(x), 0, 1e-4, 0.0, -1, 1, C.byref(it), C.byref(rr)))
print('with the event profile on:', 1e3 * (time.time() - t0), 'ms')
_lib.check(lib.fb_prof_enable(ctx, 0))
for k, v in _lib.prof_snapshot().items():
    print(f'   {k:24s} launches {v[0]:4d}  {v[1]:8.3f} ms')
