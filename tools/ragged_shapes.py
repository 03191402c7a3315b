import sys, time, numpy as np
sys.path.insert(0, '.')
from feabas_amd import _lib, matcher
lib, ctx = _lib.load(), _lib.ctx()
P, H, W = 96, 4096, 510
s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
_lib.check(lib.fb_synth_strips_dev(ctx, P, 0, H, W, 7, 20, 1, 0.4, s0.ptr, s1.ptr, sh.ptr))
h0 = s0.to_array((P, H, W), np.uint8); h1 = s1.to_array((P, H, W), np.uint8)
rng = np.random.default_rng(0)
pairs = []
for k in range(P):
    dh, dw = int(rng.integers(0, 30)), int(rng.integers(0, 12))
    pairs.append((np.ascontiguousarray(h0[k, :H - dh, :W - dw]), np.ascontiguousarray(h1[k, :H - dh, :W - dw])))
cfg = dict(sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2)
matcher.stitching_matcher(pairs[0][0], pairs[0][1], **cfg)
t = time.time()
for a, b in pairs:
    r = matcher.stitching_matcher(a, b, **cfg)
dt = time.time() - t
print(f'per-pair surface, {P} distinct shapes: {P / dt:.1f} pairs/s ({1e3 * dt / P:.1f} ms per pair)')
t = time.time()
matcher.stitching_matcher_batch(pairs, batch=32, threads=4, **cfg)
t = time.time()
out = matcher.stitching_matcher_batch(pairs, batch=32, threads=4, **cfg)
dt = time.time() - t
print(f'batch surface, {P} distinct shapes: {P / dt:.1f} pairs/s; matched {sum(o[0] is not None for o in out)}')
ref = [matcher.stitching_matcher(a, b, **cfg) for a, b in pairs[:6]]
print('ragged batch == per pair:', all(np.allclose(o[0], r[0], atol=1e-5) and np.allclose(o[2], r[2], atol=1e-5) for o, r in zip(out, ref)))
same = [(h0[k], h1[k]) for k in range(P)]
matcher.stitching_matcher_batch(same, batch=P, threads=1, **cfg)
t = time.time(); out = matcher.stitching_matcher_batch(same, batch=P, threads=1, **cfg); dt = time.time() - t
print(f'batch surface, one shape: {P / dt:.1f} pairs/s')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for a, b in pairs:
    matcher.stitching_matcher(a, b, **cfg)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
