import sys, cProfile, pstats, numpy as np, ctypes as C
sys.path.insert(0, '.')
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import StripBatchMatcher
lib, ctx = _lib.load(), _lib.ctx()
P, H, W = 64, 4096, 510
s0 = _lib.DeviceBuffer(P*H*W); s1 = _lib.DeviceBuffer(P*H*W); sh = _lib.DeviceBuffer(P*8)
_lib.check(lib.fb_synth_strips_dev(ctx, P, 0, H, W, 7, 20, 1, 2.0, s0.ptr, s1.ptr, sh.ptr))
m = StripBatchMatcher(P, H, W, residue_len=2.0)
r = m.match(s0.ptr, s1.ptr)
pr = cProfile.Profile(); pr.enable()
for _ in range(3): r = m.match(s0.ptr, s1.ptr)
pr.disable()
print('deformed', r['deformed'].sum(), 'tiers', np.bincount(np.concatenate(list(m.last_tiers.values())), minlength=4))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
