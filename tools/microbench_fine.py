"""the fine block round of the 4k configuration through fb_ncc_blocks_dev (crop mode out of the DoG strips in HBM, as the
matcher calls it): 385 blocks of 75 x 73 (LR strips 4096 x 510) or 73 x 75 (UD strips 510 x 4096) per pair, FFT 75 x 75"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
P = int(os.environ.get('P', 128))
for H, W in ((4096, 510), (510, 4096)):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((8, H, W)).astype(np.float32)
    a = np.tile(a, (-(-P // 8), 1, 1))[:P]
    d0 = _lib.DeviceBuffer.from_array(a); d1 = _lib.DeviceBuffer.from_array(np.roll(a, (2, -3), (1, 2)))
    ny, nx = -(-H // 75), -(-W // 75)
    bh, bw = -(-H // ny), -(-W // nx)
    ys = np.round(np.linspace(0, H - bh, ny)).astype(int); xs = np.round(np.linspace(0, W - bw, nx)).astype(int)
    blk = np.asarray([[p, x0, y0, bh, bw, x0 + 1, y0 - 2, bh, bw] for p in range(P) for y0 in ys for x0 in xs], dtype=np.int32)
    nb = blk.shape[0]
    dblk = _lib.DeviceBuffer.from_array(blk)
    dx = _lib.DeviceBuffer(nb * 8); dy = _lib.DeviceBuffer(nb * 8); cf = _lib.DeviceBuffer(nb * 4)
    Fh, Fw = lib.fb_next_fast_len(bh), lib.fb_next_fast_len(bw)
    def run():
        _lib.check(lib.fb_ncc_blocks_dev(ctx, d0.ptr, d1.ptr, H, W, H, W, nb, dblk.ptr, bh, bw, Fh, Fw, 1, 2, dx.ptr, dy.ptr, cf.ptr))
    for r in range(2): run()
    _lib.check(lib.fb_sync(ctx)); _lib.check(lib.fb_prof_reset(ctx)); _lib.check(lib.fb_prof_enable(ctx, 1))
    R = int(os.environ.get('REPS', 5))
    for r in range(R): run()
    _lib.check(lib.fb_sync(ctx)); _lib.check(lib.fb_prof_enable(ctx, 0))
    print(f'{H}x{W}: blocks {nb} of {bh}x{bw}, FFT {Fh}x{Fw}, dx {dx.to_array((3,), np.float64)} dy {dy.to_array((3,), np.float64)} conf {cf.to_array((3,), np.float32)}')
    for k, (n, ms, b) in _lib.prof_snapshot().items():
        print(f'  {k:20s} launches {n:3d} {ms/n:8.3f} ms/launch  {1e3*ms/R/nb:7.4f} us/block  {b/ms/1e6 if ms else 0:8.1f} GB/s')
    del d0, d1
