"""randomised A/B of fb_match_strips against the numpy statement of the same sequence (StripBatchMatcher route='native' /
'host'): uniform batches of random shapes / options on device-synthesised strips with some hand-made hard pairs (no
texture, 2.5 px warp), a third of them with valid-pixel masks / photometric statistics, and ragged batches of random
extents inside one bucket.  Tables of pairs the entry finishes must be
bit-identical; flagged pairs and the strain agree to rounding."""
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import importlib.util
spec = importlib.util.spec_from_file_location('tp', 'tests/test_gpu_pipeline.py'); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
import feabas_amd
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import StripBatchMatcher, RaggedStripBatchMatcher
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0; nflag = 0; npair = 0; ndef = 0
t0 = time.time()


def compare(mn, mh, rn, rh, P, tag):
    global bad, nflag, npair, ndef
    ok = all(np.array_equal(rn[k], rh[k]) for k in ('tx', 'ty', 'conf0', 'valid', 'deformed'))
    gn = StripBatchMatcher.per_pair(rn); gh = StripBatchMatcher.per_pair(rh)
    for p in range(P):
        npair += 1; nflag += bool(mn.last_flags is not None and mn.last_flags[p]); ndef += bool(rn['deformed'][p] and not mn.last_flags[p])
        if not rh['valid'][p]:
            ok &= gn[p]['xy0'] is None
            continue
        if gn[p]['xy0'] is None or gn[p]['xy0'].shape != gh[p]['xy0'].shape:
            ok = False; continue
        # a pair the entry finishes is bit-identical unless its mesh1 was deformed in a batch whose solve the two routes composed
        # differently (the entry hands pairs back to a sub-batch; the PCG scalars of the block-diagonal system are shared)
        exact = mn.last_flags is not None and not mn.last_flags[p] and not (rh['deformed'][p] and mn.last_flags.any())
        for k in ('xy0', 'xy1', 'weight'):
            ok &= np.array_equal(gn[p][k], gh[p][k]) if exact else bool(np.abs(gn[p][k] - gh[p][k]).max() < 1e-6)
        ok &= abs(gn[p]['strain'] - gh[p]['strain']) <= 1e-6 * max(abs(gh[p]['strain']), 1e-4)
    if not ok:
        bad += 1
        print('MISMATCH', tag)


for c in range(ncase):
    long_side = int(rng.choice([640, 1024, 1536, 2048, 3000])) - int(rng.integers(0, 9))
    short = int(rng.choice([120, 200, 255, 256, 300, 510])) - int(rng.integers(0, 3))
    H, W = (long_side, short) if rng.random() < 0.5 else (short, long_side)
    if c % 6 == 5:
        H = W = int(rng.choice([255, 300, 510]))                   # corner overlaps: one spacing
    P = int(rng.integers(2, 9))
    cds = 0.5 if rng.random() < 0.7 else 1
    kw = dict(coarse_downsample=cds, residue_mode=str(rng.choice(['huber', 'threshold'])), residue_len=float(rng.choice([2.0, 3.0, 5.0])),
              min_num_blocks=int(rng.choice([2, 3])), conf_thresh=float(rng.choice([0.33, 0.5])))
    s0, s1, _ = tp._synth(feabas_amd, P, H, W, seed=int(rng.integers(1, 1 << 20)), max_shift=int(rng.integers(2, 20)), warp=float(rng.choice([0.0, 0.3, 0.8])))
    h0 = s0.to_array((P, H, W), np.uint8); h1 = s1.to_array((P, H, W), np.uint8)
    if rng.random() < 0.5:
        h1[int(rng.integers(0, P))] = rng.integers(0, 256, (H, W), dtype=np.uint8)
    if rng.random() < 0.5 and min(H, W) >= 100:
        k = int(rng.integers(0, P)); h0[k], h1[k] = tp._warped_pair(H, W, int(rng.integers(1, 999)), (3, -2), 2.5)
    d0 = _lib.DeviceBuffer.from_array(h0); d1 = _lib.DeviceBuffer.from_array(h1)
    mn = StripBatchMatcher(P, H, W, route='native', **kw); mh = StripBatchMatcher(P, H, W, route='host', **kw)
    extras = {}
    if rng.random() < 0.35:                                        # valid-pixel masks on some strips, photometric statistics
        mk0 = [None] * P; mk1 = [None] * P
        for ml in (mk0, mk1):
            for p in range(P):
                if rng.random() < 0.4:
                    mk = np.ones((H, W), dtype=np.uint8)
                    if rng.random() < 0.5: mk[:, :int(rng.integers(1, W // 3))] = 0
                    else: mk[int(rng.integers(H // 2, H - 1)):] = 0
                    ml[p] = mk
        extras = dict(masks0=mk0, masks1=mk1, compute_photometric=bool(rng.random() < 0.7))
    rn = mn.match(d0.ptr, d1.ptr, **extras); rh = mh.match(d0.ptr, d1.ptr, **extras)
    compare(mn, mh, rn, rh, P, ('uniform', H, W, P, kw, bool(extras)))
    if extras.get('compute_photometric'):
        for a, b in zip(rn['phtm'], rh['phtm']):
            if (a is None) != (b is None) or (a is not None and not np.allclose(a, b, rtol=1e-5, equal_nan=True)):
                bad += 1; print('PHOTOMETRIC MISMATCH', a, b)
    mn.free(); mh.free(); d0.free(); d1.free(); s0.free(); s1.free()

for c in range(max(2, ncase // 3)):
    base = [(1536, 120), (120, 1536), (1024, 256), (2400, 200)][c % 4]
    shapes = []
    key = RaggedStripBatchMatcher.bucket_key(*base)
    while len(shapes) < 6:
        shp = (base[0] - int(rng.integers(0, 16)), base[1] - int(rng.integers(0, 4))) if base[0] > base[1] else (base[0] - int(rng.integers(0, 4)), base[1] - int(rng.integers(0, 16)))
        if RaggedStripBatchMatcher.bucket_key(*shp) == key:
            shapes.append(shp)
    pairs = [tp._warped_pair(h, w, int(rng.integers(1, 9999)), shift=(int(rng.integers(-6, 7)), int(rng.integers(-6, 7))), warp=float(rng.choice([0.0, 0.3, 2.5]))) for h, w in shapes]
    P = len(shapes)
    Hm, Wm = max(h for h, _ in shapes), max(w for _, w in shapes)
    stage = np.full((2, P, Hm, Wm), 200, dtype=np.uint8)
    for k, (a, b) in enumerate(pairs):
        stage[0, k, :a.shape[0], :a.shape[1]] = a; stage[1, k, :b.shape[0], :b.shape[1]] = b
    dev = _lib.DeviceBuffer.from_array(stage)
    kw = dict(residue_len=2.0, coarse_downsample=0.5 if c % 2 == 0 else 1)
    mn = RaggedStripBatchMatcher(shapes, route='native', **kw); mh = RaggedStripBatchMatcher(shapes, route='host', **kw)
    rn = mn.match(dev.ptr, dev.offset(P * Hm * Wm)); rh = mh.match(dev.ptr, dev.offset(P * Hm * Wm))
    compare(mn, mh, rn, rh, P, ('ragged', shapes, kw))
    mn.free(); mh.free(); dev.free()
print(f'{ncase} uniform + {max(2, ncase // 3)} ragged batches, {npair} pairs ({nflag} handed back by the entry, {ndef} finished there with a deformed mesh), mismatching batches {bad}, {time.time() - t0:.0f} s')
