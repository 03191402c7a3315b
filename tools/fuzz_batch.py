"""randomised check of stitching_matcher_batch (shape buckets, ragged batches, deformed meshes inside ragged batches) against
the per-pair surface on a list of pairs of random shapes"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import importlib.util
spec = importlib.util.spec_from_file_location('tp', 'tests/test_gpu_pipeline.py'); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
from feabas_amd import matcher
rng = np.random.default_rng(77)
pairs, meta = [], []
for t in range(44):
    base = [(1536, 120), (120, 1536), (1024, 256), (2400, 200)][t % 4]
    H = base[0] - int(rng.integers(0, 14)) if base[0] > base[1] else base[0] - int(rng.integers(0, 6))
    W = base[1] - int(rng.integers(0, 6)) if base[0] > base[1] else base[1] - int(rng.integers(0, 14))
    if t % 11 == 0:
        H, W = base                                         # a few pairs share a shape exactly
    warp = float(rng.choice([0.0, 0.3, 2.5]))
    shift = (int(rng.integers(-7, 8)), int(rng.integers(-7, 8)))
    pairs.append(tp._warped_pair(H, W, 900 + t, shift=shift, warp=warp)); meta.append((H, W, warp))
cfg = dict(sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2)
got = matcher.stitching_matcher_batch(pairs, batch=8, threads=3, **cfg)
bad = 0
for k, ((a, b), g) in enumerate(zip(pairs, got)):
    e = matcher.stitching_matcher(a, b, **cfg)
    if e[0] is None or g[0] is None:
        ok = (e[0] is None) == (g[0] is None)
    else:
        ok = g[0].shape == e[0].shape and np.abs(g[0] - e[0]).max() < 3e-4 and np.abs(g[1] - e[1]).max() < 3e-4 and np.abs(g[2] - e[2]).max() < 3e-4 \
            and abs(g[3] - e[3]) < 3e-3 * max(e[3], 1e-3)
    bad += not ok
    if not ok:
        print(k, meta[k], 'MISMATCH', None if g[0] is None else g[0].shape, None if e[0] is None else e[0].shape)
print(len(pairs), 'pairs,', len({p[0].shape for p in pairs}), 'shapes, mismatches', bad)
matcher.stitching_matcher_batch_release()
