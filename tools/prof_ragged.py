"""host profile of one ragged chunk (32 pairs of 32 shapes around 4096 x 510) through RaggedStripBatchMatcher"""
import sys, time, cProfile, pstats, numpy as np
sys.path.insert(0, '.')
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import RaggedStripBatchMatcher, StripBatchMatcher, MatcherPool
lib, ctx = _lib.load(), _lib.ctx()
P, H, W = 32, 4096, 510
s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
_lib.check(lib.fb_synth_strips_dev(ctx, P, 0, H, W, 7, 20, 1, 0.4, s0.ptr, s1.ptr, sh.ptr))
rng = np.random.default_rng(0)
shapes = []
while len(shapes) < P:
    shp = (H - int(rng.integers(0, 30)), W - int(rng.integers(0, 12)))
    if RaggedStripBatchMatcher.bucket_key(*shp) == RaggedStripBatchMatcher.bucket_key(H, W):
        shapes.append(shp)
pool = MatcherPool()
def run():
    m = RaggedStripBatchMatcher(shapes, residue_len=2.0, pool=pool)
    r = m.match(s0.ptr, s1.ptr)          # the synthetic strips are larger than the extents: their corners are matched
    m.free()
    return r
run()
t = time.time(); r = run(); print('ragged chunk', 1e3 * (time.time() - t), 'ms, valid', int(r['valid'].sum()))
mu = StripBatchMatcher(P, H, W, residue_len=2.0, pool=pool); mu.match(s0.ptr, s1.ptr)
t = time.time(); mu.match(s0.ptr, s1.ptr); print('uniform chunk', 1e3 * (time.time() - t), 'ms')
pr = cProfile.Profile(); pr.enable(); run(); run(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
# one matcher, several calls: the stage clock of fb_match_strips (FEABAS_HIP_MATCH_TRACE=1) and the kernels of one call
for cls, arg in ((RaggedStripBatchMatcher, (shapes,)), (StripBatchMatcher, (P, H, W))):
    m = cls(*arg, residue_len=2.0, pool=pool)
    for _ in range(8):
        m.match(s0.ptr, s1.ptr)
    lib.fb_prof_enable(ctx, 1); lib.fb_prof_reset(ctx)
    m.match(s0.ptr, s1.ptr)
    snap = _lib.prof_snapshot(); lib.fb_prof_enable(ctx, 0)
    print(cls.__name__, 'kernels', round(sum(v[1] for v in snap.values()), 2), 'ms', {k: (v[0], round(v[1], 2)) for k, v in sorted(snap.items(), key=lambda kv: -kv[1][1])})
    m.free()
