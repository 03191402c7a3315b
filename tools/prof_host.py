"""host-side profile of StripBatchMatcher.match (where the per-step Python time goes)"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import StripBatchMatcher
lib, ctx = _lib.load(), _lib.ctx()
P, H, W = 64, 4096, 510
s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
_lib.check(lib.fb_synth_strips_dev(ctx, P, 0, H, W, 2026, 20, 1, 0.4, s0.ptr, s1.ptr, sh.ptr))
m = StripBatchMatcher(P, H, W)
m.match(s0.ptr, s1.ptr); m.match(s0.ptr, s1.ptr)
t0 = time.time()
for _ in range(5): m.match(s0.ptr, s1.ptr)
print('ms per match() of %d pairs: %.2f' % (P, (time.time() - t0) / 5 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): m.match(s0.ptr, s1.ptr)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
