"""micro-benchmark of fb_dog_dev on device-resident strips (prints a CRC of the output so that kernel variants can be
compared bit for bit)"""
import ctypes as C, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
rng = np.random.default_rng(0)
for (N, H, W, s) in ((128, 4096, 510, 2.5), (128, 510, 4096, 2.5), (256, 2048, 255, 1.25), (4, 3000, 500, 2.5), (3, 333, 217, 3.0)):
    a = rng.integers(0, 256, (N, H, W), dtype=np.uint8)
    d = _lib.DeviceBuffer.from_array(a); o = _lib.DeviceBuffer(N * H * W * 4)
    ms = C.c_float(); ts = []
    for r in range(21):
        _lib.check(lib.fb_timer_start(ctx))
        _lib.check(lib.fb_dog_dev(ctx, d.ptr, 0, N, H, W, s, None, 1, o.ptr))
        _lib.check(lib.fb_timer_stop(ctx, C.byref(ms)))
        if r: ts.append(ms.value)
    best = float(np.median(ts))
    crc = zlib.crc32(o.to_array((N, H, W), np.float32).tobytes())
    print(f'dog N={N} {H}x{W} sigma={s}: {best:7.3f} ms (median of 20, min {min(ts):.3f})  {N*H*W*5/best/1e6:8.1f} GB/s algorithmic  {1e3*best/N:7.2f} us/image  crc {crc:08x}')
    d.free(); o.free()
