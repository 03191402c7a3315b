"""Host / device time of one StripBatchMatcher.match call on the 4k configuration, rigid branch (0.4-px warp) against the
deformed branch (2-px warp): wall per call on one host thread, event time per kernel, cProfile of the host side."""
import sys, time, cProfile, pstats, numpy as np
sys.path.insert(0, '.')
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import StripBatchMatcher
lib, ctx = _lib.load(), _lib.ctx()
P, H, W = 64, 4096, 510
s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
m = StripBatchMatcher(P, H, W, residue_len=2.0)
for warp in (0.4, 2.0):
    _lib.check(lib.fb_synth_strips_dev(ctx, P, 0, H, W, 7, 20, 1, warp, s0.ptr, s1.ptr, sh.ptr))
    r = m.match(s0.ptr, s1.ptr)
    _lib.check(lib.fb_sync(ctx)); t = time.perf_counter()
    for _ in range(3):
        r = m.match(s0.ptr, s1.ptr)
    wall = (time.perf_counter() - t) / 3
    lib.fb_prof_enable(ctx, 1); lib.fb_prof_reset(ctx)
    r = m.match(s0.ptr, s1.ptr)
    snap = _lib.prof_snapshot(); lib.fb_prof_enable(ctx, 0)
    tiers = np.bincount(np.concatenate(list(m.last_tiers.values())), minlength=4) if m.last_tiers else None
    print(f'warp {warp}: wall {1e3 * wall:.1f} ms per call, deformed {int(r["deformed"].sum())}/{P}, tiers {tiers}, kernels {sum(v[1] for v in snap.values()):.2f} ms')
    print('   ', {k: round(v[1], 2) for k, v in sorted(snap.items(), key=lambda kv: -kv[1][1])})
    if warp > 1:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(3):
            m.match(s0.ptr, s1.ptr)
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(18)
m.free()                                                   # FEABAS_HIP_MATCH_TRACE=1: the stage clock of fb_match_strips prints here
