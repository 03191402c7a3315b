import sys, time, cProfile, pstats, numpy as np
sys.path.insert(0, '.')
from feabas_amd import _lib, matcher
lib, ctx = _lib.load(), _lib.ctx()
P, H, W = 4, 4096, 510
s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
_lib.check(lib.fb_synth_strips_dev(ctx, P, 0, H, W, 7, 20, 1, 0.4, s0.ptr, s1.ptr, sh.ptr))
h0 = s0.to_array((P, H, W), np.uint8); h1 = s1.to_array((P, H, W), np.uint8)
cfg = dict(sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2)
for k in range(P): matcher.stitching_matcher(h0[k], h1[k], **cfg)
t = time.time()
for r in range(10):
    for k in range(P): matcher.stitching_matcher(h0[k], h1[k], **cfg)
print('one pair at a time, one shape: %.2f ms per pair' % (1e3 * (time.time() - t) / (10 * P)))
pr = cProfile.Profile(); pr.enable()
for r in range(5):
    for k in range(P): matcher.stitching_matcher(h0[k], h1[k], **cfg)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
