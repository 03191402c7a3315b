"""micro-benchmark of fb_ncc_batch_dev on device-resident stacks: per-launch time vs batch size"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
def run(N, h, w, pad, sub=1, reps=5):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((min(N, 512), h, w)).astype(np.float32)
    a = np.tile(a, (-(-N // a.shape[0]), 1, 1))[:N]
    d0 = _lib.DeviceBuffer.from_array(a); d1 = _lib.DeviceBuffer.from_array(np.roll(a, (2, -3), (1, 2)))
    dx = _lib.DeviceBuffer(N * 8); dy = _lib.DeviceBuffer(N * 8); cf = _lib.DeviceBuffer(N * 4)
    ms = C.c_float()
    best = 1e9
    for r in range(reps + 1):
        _lib.check(lib.fb_timer_start(ctx))
        _lib.check(lib.fb_ncc_batch_dev(ctx, d0.ptr, d1.ptr, N, 1, h, w, h, w, pad, sub, 2, dx.ptr, dy.ptr, cf.ptr))
        _lib.check(lib.fb_timer_stop(ctx, C.byref(ms)))
        if r: best = min(best, ms.value)
    for b in (d0, d1, dx, dy, cf): b.free()
    return best
for (h, w, pad) in ((75, 73, 0), (72, 72, 0), (64, 64, 0), (75, 73, 1)):
    for N in (1, 256, 768, 1536, 7680, 24640):
        t = run(N, h, w, pad)
        print(f'{h}x{w} pad={pad} N={N:6d}: {t:8.3f} ms  {1e3*t/N:8.2f} us/block  {N/t/1e3*1e3:10.0f} blocks/s')
print('--- streaming class')
for (h, w, N) in ((1024, 510, 128), (1024, 490, 64), (2048, 255, 32), (510, 1024, 128), (255, 2048, 32), (280, 280, 400)):
    t = run(N, h, w, 1, sub=0, reps=3)
    from feabas_amd.matcher import next_fast_len as nfl
    fh, fw = nfl(2 * h - 1), nfl(2 * w - 1)
    alg = N * (2 * h * w * 4 + 48 * fh * (fw // 2 + 1))
    print(f'{h}x{w} -> FFT {fh}x{fw} N={N:4d}: {t:8.3f} ms  {1e3*t/N:8.2f} us/pair  algorithmic {alg/t/1e6:8.1f} GB/s')
