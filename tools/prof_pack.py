"""raw rates of the host side of the PCIe-inclusive path: fb_host_pack2d (pageable strips -> page-locked stack) and the H2D copy
of the stack, alone and from several host threads at once"""
import sys, time, os, threading, ctypes as C, numpy as np
sys.path.insert(0, '.')
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
H, W, n = 4096, 510, 64
rng = np.random.default_rng(0)
src = [rng.integers(0, 255, (H, W), dtype=np.uint8) for _ in range(2 * n)]
hs = np.full(2 * n, H, np.int32); ws = np.full(2 * n, W, np.int32); pitches = np.full(2 * n, W, np.int64)
srcs = (C.c_void_p * (2 * n))(*[a.ctypes.data for a in src])
nbytes = 2 * n * H * W

def run(NT, PT, reps=6, copy=True):
    ctxs = [ctx] + [_lib.new_context() for _ in range(NT - 1)]
    pins = [_lib.PinnedBuffer(nbytes) for _ in range(NT)]; devs = [_lib.DeviceBuffer(nbytes) for _ in range(NT)]
    tp = [0.0] * NT; tc = [0.0] * NT
    def w(t):
        _lib.use_context(ctxs[t])
        for r in range(reps):
            t0 = time.time()
            _lib.check(lib.fb_host_pack2d(_lib.ctx(), pins[t].ptr, 2 * n, H, W, srcs, _lib.ptr(hs), _lib.ptr(ws), _lib.ptr(pitches), PT))
            t1 = time.time()
            if copy:
                _lib.check(lib.fb_memcpy_h2d(_lib.ctx(), devs[t].ptr, pins[t].ptr, nbytes)); _lib.check(lib.fb_sync(_lib.ctx()))
            t2 = time.time()
            if r:
                tp[t] += t1 - t0; tc[t] += t2 - t1
        _lib.use_context(None)
    t0 = time.time()
    ths = [threading.Thread(target=w, args=(t,)) for t in range(NT)]
    [t.start() for t in ths]; [t.join() for t in ths]
    wall = time.time() - t0
    tot = NT * reps * nbytes
    print(f'threads {NT} x pack threads {PT}: wall {tot / wall / 1e9:6.1f} GB/s ({NT * reps * n / wall:7.0f} pairs/s)   pack {nbytes * (reps - 1) / max(tp) / 1e9:5.1f} GB/s per thread, h2d {nbytes * (reps - 1) / max(max(tc), 1e-9) / 1e9:5.1f} GB/s per thread')
    for p in pins: p.free()
    for d in devs: d.free()
    for h in ctxs[1:]: _lib.destroy_context(h)
print('cpu budget', _lib.cpu_budget())
for NT, PT in ((1, 1), (1, 2), (1, 4), (1, 8), (4, 2), (8, 1), (8, 2), (16, 1)):
    run(NT, PT)
