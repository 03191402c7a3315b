"""cProfile of stitching_matcher_batch on ONE host thread (ragged or UNIFORM=1 list of 256 pairs): the Python held under the
interpreter lock per chunk shows as tottime outside the ctypes calls"""
import sys, time, os, cProfile, pstats, numpy as np
sys.path.insert(0, '.')
from feabas_amd import _lib, matcher as fmatcher
lib, ctx = _lib.load(), _lib.ctx()
N, H, W = 256, 4096, 510
nh = 128
s0 = _lib.DeviceBuffer(nh * H * W); s1 = _lib.DeviceBuffer(nh * H * W); sh = _lib.DeviceBuffer(nh * 8)
_lib.check(lib.fb_synth_strips_dev(ctx, nh, 0, H, W, 7, 20, 1, 0.4, s0.ptr, s1.ptr, sh.ptr))
h0 = s0.to_array((nh, H, W), np.uint8); h1 = s1.to_array((nh, H, W), np.uint8)
rng = np.random.default_rng(5)
pairs = []
for k in range(N):
    dh, dw = (0, 0) if os.environ.get('UNIFORM') else (int(rng.integers(0, 30)), int(rng.integers(0, 12)))
    pairs.append((h0[k % nh, :H - dh, :W - dw], h1[k % nh, :H - dh, :W - dw]))
cfg = dict(sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2.0)
fmatcher.stitching_matcher_batch(pairs, batch=32, threads=1, **cfg)
t = time.time(); fmatcher.stitching_matcher_batch(pairs, batch=32, threads=1, **cfg); print('one thread', round(time.time() - t, 3), 's')
pr = cProfile.Profile(); pr.enable()
fmatcher.stitching_matcher_batch(pairs, batch=32, threads=1, **cfg)
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
