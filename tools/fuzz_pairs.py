"""randomised parity sweep: stitching_matcher against the oracle pipeline over random strip shapes (odd sizes included), warps, offsets,
coarse_downsample, residue length and residue mode.  `python tools/fuzz_pairs.py [big | scales]` (scales: coarse / fine downsample
factors away from the defaults, general-mesh route)"""
import sys, numpy as np, traceback
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import importlib.util
spec = importlib.util.spec_from_file_location('tp', 'tests/test_gpu_pipeline.py'); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
from feabas_amd import matcher
from oracle import pipeline_ref
big = len(sys.argv) > 1 and sys.argv[1] == 'big'          # long strips: two or three spacings, deformed meshes
scales = len(sys.argv) > 1 and sys.argv[1] == 'scales'
rng = np.random.default_rng(321 if big else 123)
bad = 0
for t in range(12 if big else 24):
    long_side = int(rng.integers(2400, 4200) if big else rng.integers(900, 1900)); short = int(rng.integers(70, 520) if big else rng.integers(100, 300))
    H, W = (long_side, short) if rng.random() < 0.5 else (short, long_side)
    warp = float(rng.choice([0.3, 1.5, 3.0, 4.0]) if big else rng.choice([0.0, 0.3, 1.5, 3.0]))
    shift = (int(rng.integers(-9, 10)), int(rng.integers(-9, 10)))
    cds = float(rng.choice([0.5, 1.0])); rl = float(rng.choice([2.0, 5.0])); mode = str(rng.choice(['huber', 'threshold']))
    fds = 1.0
    if scales:
        cds, fds = [(0.25, 0.5), (0.5, 0.5), (0.25, 1.0), (0.4, 0.8), (1 / 3, 1.0), (0.3, 0.6)][t % 6]
    s0, s1 = tp._warped_pair(H, W, 500 + t, shift=shift, warp=warp)
    cfg = dict(sigma=2.5, coarse_downsample=cds if cds != 1.0 else 1, fine_downsample=fds if fds != 1.0 else 1, conf_thresh=0.33, residue_len=rl, residue_mode=mode)
    try:
        got = matcher.stitching_matcher(s0, s1, **cfg)
        exp = pipeline_ref.match_pair(s0, s1, coarse_downsample=cds, fine_downsample=fds, residue_len=rl, residue_mode=mode)
        if got[0] is None or exp['xy0'] is None:
            ok = (got[0] is None) == (exp['xy0'] is None)
            msg = 'none'
        else:
            ok = got[0].shape == exp['xy0'].shape and np.abs(got[0] - exp['xy0']).max() < 5e-3 / fds and np.abs(got[1] - exp['xy1']).max() < 5e-3 / fds \
                and np.abs(got[2] - exp['weight']).max() < 5e-3 and abs(got[3] - exp['strain']) < 5e-3 * max(exp['strain'], 1e-3)
            msg = f"n={got[0].shape[0]} dxy={np.abs(got[0] - exp['xy0']).max():.1e} dw={np.abs(got[2] - exp['weight']).max():.1e} deformed={exp.get('deformed', False)}"
    except Exception as e:
        ok = False; msg = 'EXC ' + repr(e); traceback.print_exc()
    print(t, (H, W), 'warp', warp, 'shift', shift, 'cds', round(cds, 3), 'fds', fds, 'rl', rl, mode, 'OK' if ok else 'MISMATCH', msg, flush=True)
    bad += not ok
print('mismatches', bad)
