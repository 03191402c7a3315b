import sys, time, numpy as np, ctypes as C
sys.path.insert(0, '.')
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import StripBatchMatcher, MatcherPool
lib, ctx = _lib.load(), _lib.ctx()
P, H, W = 64, 4096, 510
s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
_lib.check(lib.fb_synth_strips_dev(ctx, P, 0, H, W, 7, 20, 1, 0.4, s0.ptr, s1.ptr, sh.ptr))
h0 = s0.to_array((P, H, W), np.uint8); h1 = s1.to_array((P, H, W), np.uint8)
items = [(h0[k], h1[k]) for k in range(P)]
pin = _lib.PinnedBuffer(2 * P * H * W); dev = _lib.DeviceBuffer(2 * P * H * W)
m = StripBatchMatcher(P, H, W, residue_len=2.0, pool=MatcherPool())
def chunk():
    t0 = time.perf_counter()
    srcs = (C.c_void_p * (2 * P))(*([a.ctypes.data for a, _ in items] + [b.ctypes.data for _, b in items]))
    hs = np.full(2 * P, H, np.int32); ws = np.full(2 * P, W, np.int32); pt = np.full(2 * P, W, np.int64)
    _lib.check(lib.fb_host_pack2d(ctx, pin.ptr, 2 * P, H, W, srcs, _lib.ptr(hs), _lib.ptr(ws), _lib.ptr(pt), 2))
    t1 = time.perf_counter()
    _lib.check(lib.fb_memcpy_h2d(ctx, dev.ptr, pin.ptr, 2 * P * H * W))
    _lib.check(lib.fb_sync(ctx)); t2 = time.perf_counter()
    out = m.match(dev.ptr, dev.offset(P * H * W)); t3 = time.perf_counter()
    per = StripBatchMatcher.per_pair(out); t4 = time.perf_counter()
    return [1e3 * (b - a) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4))]
chunk()
ts = np.array([chunk() for _ in range(5)]).mean(axis=0)
print('pack %.1f ms, h2d %.1f ms (%.1f GB/s), match %.1f ms, per_pair %.1f ms' % (ts[0], ts[1], 2 * P * H * W / ts[1] / 1e6, ts[2], ts[3]))
