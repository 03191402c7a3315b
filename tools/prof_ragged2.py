"""where a ragged chunk spends its host time: matcher construction (fb_strip_matcher_create_ragged), the match call, the
release; FEABAS_HIP_MATCH_TRACE=1 adds the stage clock of fb_match_strips for the last call of each kind"""
import sys, time, os, numpy as np
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
acc = np.zeros(4)
for rep in range(12):
    t0 = time.perf_counter()
    m = RaggedStripBatchMatcher(shapes, residue_len=2.0, pool=pool)
    t1 = time.perf_counter()
    m._native_matcher()
    t2 = time.perf_counter()
    m.match(s0.ptr, s1.ptr)
    t3 = time.perf_counter()
    m.free()
    t4 = time.perf_counter()
    if rep >= 2:
        acc += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
print('ragged: python ctor %.2f ms, native create %.2f ms, match %.2f ms, free %.2f ms' % tuple(1e3 * acc / 10))
mu = StripBatchMatcher(P, H, W, residue_len=2.0, pool=pool)
for _ in range(3):
    mu.match(s0.ptr, s1.ptr)
t = time.perf_counter()
for _ in range(10):
    mu.match(s0.ptr, s1.ptr)
print('uniform: match %.2f ms' % (1e2 * (time.perf_counter() - t)))
if os.environ.get('FEABAS_HIP_MATCH_TRACE'):
    print('--- uniform', file=sys.stderr, flush=True)
    mu.match(s0.ptr, s1.ptr)
    print('--- ragged', file=sys.stderr, flush=True)
    m = RaggedStripBatchMatcher(shapes, residue_len=2.0, pool=pool); m.match(s0.ptr, s1.ptr); m.free()
    print('--- uniform (all calls)', file=sys.stderr, flush=True)
mu.free()
