"""per-kernel durations of the coarse block round of the 4k configuration through fb_ncc_blocks_dev"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
P, H, W = int(os.environ.get('P', 64)), int(os.environ.get('H', 4096)), int(os.environ.get('W', 510))
rng = np.random.default_rng(0)
a = rng.standard_normal((P, H, W)).astype(np.float32)
d0 = _lib.DeviceBuffer.from_array(a); d1 = _lib.DeviceBuffer.from_array(np.roll(a, (2, -3), (1, 2)))
bh, bw = (H // 4, W) if H > W else (H, W // 4)
blk = []
for p in range(P):
    for k in range(4):
        x0, y0 = (0, k * bh) if H > W else (k * bw, 0)
        blk.append([p, x0, y0, bh, bw, x0, y0, bh, bw])
blk = np.asarray(blk, dtype=np.int32)
nb = blk.shape[0]
dblk = _lib.DeviceBuffer.from_array(blk)
dx = _lib.DeviceBuffer(nb * 8); dy = _lib.DeviceBuffer(nb * 8); cf = _lib.DeviceBuffer(nb * 4)
nfl = lambda v: lib.fb_next_fast_len(v)
Fh, Fw = nfl(2 * bh - 1), nfl(2 * bw - 1)
def run():
    _lib.check(lib.fb_ncc_blocks_dev(ctx, d0.ptr, d1.ptr, H, W, H, W, nb, dblk.ptr, bh, bw, Fh, Fw, 0, 2, dx.ptr, dy.ptr, cf.ptr))
for r in range(2): run()
_lib.check(lib.fb_sync(ctx)); _lib.check(lib.fb_prof_reset(ctx)); _lib.check(lib.fb_prof_enable(ctx, 1))
for r in range(int(os.environ.get('REPS', 5))): run()
_lib.check(lib.fb_sync(ctx)); _lib.check(lib.fb_prof_enable(ctx, 0))
print('blocks', nb, 'FFT', Fh, Fw, 'dx', dx.to_array((4,), np.float64), 'dy', dy.to_array((4,), np.float64))
for k, (n, ms, b) in _lib.prof_snapshot().items():
    print(f'{k:22s} launches {n:3d} {ms/n:8.3f} ms/launch  {1e3*ms/(int(os.environ.get("REPS", 5)))/nb:7.2f} us/block  {b/ms/1e6 if ms else 0:8.1f} GB/s')
