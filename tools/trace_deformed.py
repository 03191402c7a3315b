"""stage clock of fb_match_strips (FEABAS_HIP_MATCH_TRACE=1) for 128 LR pairs with the default 0.4 px warp and with the 2 px
warp of the `deformed` record (mesh1 relaxed into a non-rigid field between the spacings)"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import StripBatchMatcher, MatcherPool
lib, ctx = _lib.load(), _lib.ctx()
P, H, W = 128, 4096, 510
pool = MatcherPool()
for warp in (0.4, 2.0):
    s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
    _lib.check(lib.fb_synth_strips_dev(ctx, P, 0, H, W, 7, 20, 1, warp, s0.ptr, s1.ptr, sh.ptr))
    m = StripBatchMatcher(P, H, W, residue_len=2.0, pool=pool)
    for _ in range(4):
        r = m.match(s0.ptr, s1.ptr)
    t = time.perf_counter()
    for _ in range(8):
        r = m.match(s0.ptr, s1.ptr)
    print(f'warp {warp}: {1e3 * (time.perf_counter() - t) / 8:.2f} ms per call, deformed pairs {int(np.sum(r["deformed"]))}', flush=True)
    lib.fb_prof_reset(ctx); lib.fb_prof_enable(ctx, 1); m.match(s0.ptr, s1.ptr); lib.fb_prof_enable(ctx, 0)
    snap = _lib.prof_snapshot()
    print('   kernels', round(sum(v[1] for v in snap.values()), 2), 'ms:', {k: (v[0], round(v[1], 2)) for k, v in sorted(snap.items(), key=lambda kv: -kv[1][1])[:14]}, flush=True)
    print(f'--- warp {warp}', file=sys.stderr, flush=True)
    m.free()
    for b in (s0, s1, sh):
        b.free()
