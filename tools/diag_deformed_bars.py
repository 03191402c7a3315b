"""distribution of |device - oracle| over the matches of the deformed-mesh test pairs, at two relaxation tolerances
(what the 3e-3 px bar of the deformed branch really holds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import StripBatchMatcher
from oracle import pipeline_ref
from test_gpu_pipeline import _warped_pair

cases = [((1536, 120), [(1, (4, -3), 3.0), (3, (1, 5), 2.0)]), ((120, 1536), [(1, (4, -3), 3.0), (3, (1, 5), 2.0)]),
         ((3600, 72), [(21, (2, -3), 2.5), (22, (-3, 4), 1.5)]), ((4096, 510), [(22, (-4, 9), 2.0)])]
if os.environ.get('QUICK'):
    cases = cases[:1]
for (H, W), prs in cases:
    pairs = [_warped_pair(H, W, s, sh, w) for s, sh, w in prs]
    s0 = np.stack([p[0] for p in pairs]); s1 = np.stack([p[1] for p in pairs])
    d0 = _lib.DeviceBuffer.from_array(s0); d1 = _lib.DeviceBuffer.from_array(s1)
    exps = [pipeline_ref.match_pair(s0[p], s1[p], residue_len=2.0) for p in range(len(pairs))]
    for tol in (1e-9, 1e-2, 2e-3):
        m = StripBatchMatcher(len(pairs), H, W, residue_len=2.0, relax_tol=tol)
        got = StripBatchMatcher.per_pair(m.match(d0.ptr, d1.ptr))
        for p, (g, e) in enumerate(zip(got, exps)):
            if g['xy0'].shape != e['xy0'].shape:
                print(H, W, p, tol, 'SHAPE', g['xy0'].shape, e['xy0'].shape); continue
            d = np.maximum(np.abs(g['xy0'] - e['xy0']).max(axis=1), np.abs(g['xy1'] - e['xy1']).max(axis=1))
            dw = np.abs(g['weight'] - e['weight'])
            ds = np.abs(g['strain'] - e['strain']) / max(abs(e['strain']), 1e-30)
            fld = np.abs(m.last_field[p] - e['mesh1_field']).max() if g['deformed'] else 0.0
            print(f'{H}x{W} pair {p} tol {tol:g}: n {d.size} xy max {d.max():.2e} p99 {np.percentile(d, 99):.2e} p90 {np.percentile(d, 90):.2e} med {np.median(d):.2e} '
                  f'n>1e-4 {(d > 1e-4).sum()} | w max {dw.max():.2e} n>1e-4 {(dw > 1e-4).sum()} | strain rel {ds:.2e} | field {fld:.2e} iters {getattr(m, "last_relax_iters", None)}')
        m.free()
    d0.free(); d1.free()
