"""micro-benchmark of the alignment / README block classes of fb_ncc_batch_dev (compile-time mixed-radix streaming kernels,
fb_ncc_ct.hip); FEABAS_HIP_FFT_GENERIC=1 runs the same shapes on the run-time mixed-radix kernels (A/B)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feabas_amd import _lib
from feabas_amd.matcher import next_fast_len as nfl
lib, ctx = _lib.load(), _lib.ctx()
def run(N, h, w, pad, sub=1, reps=8):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((min(N, 256), h, w)).astype(np.float32)
    a = np.tile(a, (-(-N // a.shape[0]), 1, 1))[:N]
    d0 = _lib.DeviceBuffer.from_array(a); d1 = _lib.DeviceBuffer.from_array(np.roll(a, (2, -3), (1, 2)))
    dx = _lib.DeviceBuffer(N * 8); dy = _lib.DeviceBuffer(N * 8); cf = _lib.DeviceBuffer(N * 4)
    ms = C.c_float(); ts = []
    for r in range(reps + 1):
        _lib.check(lib.fb_timer_start(ctx))
        _lib.check(lib.fb_ncc_batch_dev(ctx, d0.ptr, d1.ptr, N, 1, h, w, h, w, pad, sub, 2, dx.ptr, dy.ptr, cf.ptr))
        _lib.check(lib.fb_timer_stop(ctx, C.byref(ms)))
        if r: ts.append(ms.value)
    for b in (d0, d1, dx, dy, cf): b.free()
    return float(np.median(ts))
for (h, w, N, pad) in ((280, 280, 1024, 1), (70, 70, 1024, 1), (70, 70, 8192, 1), (74, 72, 1024, 1), (67, 75, 1024, 1), (280, 280, 1024, 0), (400, 400, 512, 1), (140, 140, 2048, 1)):
    t = run(N, h, w, pad)
    fh, fw = (nfl(2 * h - 1), nfl(2 * w - 1)) if pad else (nfl(h), nfl(w))
    alg = N * (2 * h * w * 4 + 48 * fh * (fw // 2 + 1))
    print(f'{h}x{w} pad={pad} -> reference FFT {fh}x{fw} N={N:5d}: {t:8.3f} ms  {1e3*t/N:8.2f} us/block pair  algorithmic {alg/t/1e6:8.1f} GB/s')
