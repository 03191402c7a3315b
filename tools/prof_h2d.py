"""cost of host -> device copies of 4 MB arrays: fresh pageable numpy arrays, a reused pageable array, a pinned buffer"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
n = 500 * 500 * 2
d = _lib.DeviceBuffer(n * 8)
def t_copy(a):
    t0 = time.perf_counter()
    _lib.check(lib.fb_memcpy_h2d(ctx, d.ptr, _lib.ptr(a), a.nbytes)); _lib.check(lib.fb_sync(ctx))
    return 1e3 * (time.perf_counter() - t0)
base = np.random.default_rng(0).standard_normal(n)
print('fresh arrays     ', [round(t_copy(base + k), 3) for k in range(6)])
a = base.copy()
print('one reused array ', [round(t_copy(a), 3) for k in range(6)])
hp = C.c_void_p(); _lib.check(lib.fb_host_alloc(ctx, n * 8, C.byref(hp)))
pin = np.ctypeslib.as_array(C.cast(hp, C.POINTER(C.c_double)), shape=(n,))
def t_pin(a):
    t0 = time.perf_counter()
    pin[:] = a
    _lib.check(lib.fb_memcpy_h2d(ctx, d.ptr, hp, a.nbytes)); _lib.check(lib.fb_sync(ctx))
    return 1e3 * (time.perf_counter() - t0)
print('through pinned   ', [round(t_pin(base + k), 3) for k in range(6)])
out = np.empty(n)
def t_d2h(o):
    t0 = time.perf_counter()
    _lib.check(lib.fb_memcpy_d2h(ctx, _lib.ptr(o), d.ptr, o.nbytes)); _lib.check(lib.fb_sync(ctx))
    return 1e3 * (time.perf_counter() - t0)
print('d2h fresh        ', [round(t_d2h(np.empty(n)), 3) for k in range(6)])
print('d2h reused       ', [round(t_d2h(out), 3) for k in range(6)])
