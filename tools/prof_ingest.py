"""the ragged (UNIFORM=1: equal-shape) host-ingest list of bench.py, several passes: wall per pass (one-time costs show in the
first ones); NT host threads, IB pairs per chunk"""
import sys, time, os, numpy as np
sys.path.insert(0, '.')
from feabas_amd import _lib, matcher as fmatcher
lib, ctx = _lib.load(), _lib.ctx()
N, H, W = int(os.environ.get('NP', 1024)), 4096, 510
IB, NT = int(os.environ.get('IB', 64)), int(os.environ.get('NT', 8))
nh = 512
s0 = _lib.DeviceBuffer(nh * H * W); s1 = _lib.DeviceBuffer(nh * H * W); sh = _lib.DeviceBuffer(nh * 8)
_lib.check(lib.fb_synth_strips_dev(ctx, nh, 0, H, W, 7, 20, 1, 0.4, s0.ptr, s1.ptr, sh.ptr))
h0 = s0.to_array((nh, H, W), np.uint8); h1 = s1.to_array((nh, H, W), np.uint8)
rng = np.random.default_rng(5)
ragged = []
for k in range(N):
    dh, dw = (0, 0) if os.environ.get('UNIFORM') else (int(rng.integers(0, 30)), int(rng.integers(0, 12)))
    ragged.append((h0[k % nh, :H - dh, :W - dw], h1[k % nh, :H - dh, :W - dw]))
cfg = dict(sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2.0)
for ps in range(int(os.environ.get('PASSES', 4))):
    t = time.time()
    out = fmatcher.stitching_matcher_batch(ragged, batch=IB, threads=NT, **cfg)
    dt = time.time() - t
    print(f'pass {ps}: {dt:.3f} s, {N / dt:.0f} pairs/s, matched {sum(o[0] is not None for o in out)}', flush=True)
