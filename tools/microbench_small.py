"""one shape of the on-chip NCC class, for profiling"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
N, h, w = int(os.environ.get('NB', 7680)), int(os.environ.get('BH', 75)), int(os.environ.get('BW', 73))
rng = np.random.default_rng(0)
a = rng.standard_normal((512, h, w)).astype(np.float32)
a = np.tile(a, (-(-N // 512), 1, 1))[:N]
d0 = _lib.DeviceBuffer.from_array(a); d1 = _lib.DeviceBuffer.from_array(np.roll(a, (2, -3), (1, 2)))
dx = _lib.DeviceBuffer(N * 8); dy = _lib.DeviceBuffer(N * 8); cf = _lib.DeviceBuffer(N * 4)
for r in range(2):
    _lib.check(lib.fb_ncc_batch_dev(ctx, d0.ptr, d1.ptr, N, 1, h, w, h, w, 0, 1, 2, dx.ptr, dy.ptr, cf.ptr))
_lib.check(lib.fb_sync(ctx)); _lib.check(lib.fb_prof_reset(ctx)); _lib.check(lib.fb_prof_enable(ctx, 1))
for r in range(int(os.environ.get('REPS', 5))):
    _lib.check(lib.fb_ncc_batch_dev(ctx, d0.ptr, d1.ptr, N, 1, h, w, h, w, 0, 1, 2, dx.ptr, dy.ptr, cf.ptr))
_lib.check(lib.fb_sync(ctx)); _lib.check(lib.fb_prof_enable(ctx, 0))
for k, (n, ms, b) in _lib.prof_snapshot().items():
    print(f'{k:22s} {ms/n:8.3f} ms/launch  {1e3*ms/n/N:7.3f} us/block  {b/ms/1e6 if ms else 0:8.1f} GB/s')
