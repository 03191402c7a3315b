import sys, time, cProfile, pstats, numpy as np
sys.path.insert(0, '.')
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import StripBatchMatcher
lib, ctx = _lib.load(), _lib.ctx()
P, H, W = 64, 4096, 510
s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
_lib.check(lib.fb_synth_strips_dev(ctx, P, 0, H, W, 7, 20, 1, 0.4, s0.ptr, s1.ptr, sh.ptr))
m = StripBatchMatcher(P, H, W, residue_len=2.0)
m.match(s0.ptr, s1.ptr); m.match(s0.ptr, s1.ptr)
t = time.time()
for _ in range(5): r = m.match(s0.ptr, s1.ptr)
print('uniform call', 1e3 * (time.time() - t) / 5, 'ms')
pr = cProfile.Profile(); pr.enable()
for _ in range(5): m.match(s0.ptr, s1.ptr)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
t = time.time(); pp = StripBatchMatcher.per_pair(r); print('per_pair', 1e3 * (time.time() - t), 'ms')
