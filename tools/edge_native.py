import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import importlib.util
spec = importlib.util.spec_from_file_location('tp', '/root/repo/tests/test_gpu_pipeline.py'); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
import feabas_amd
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import StripBatchMatcher
bad = 0
cases = [(1, 64, 64, {}), (1, 100, 2000, {}), (3, 2000, 100, {}), (2, 4096, 510, {}), (300, 300, 120, {}), (4, 1024, 256, dict(spacings=[20.0])),
         (4, 1024, 256, dict(spacings=[300.0, 150.0, 75.0, 40.0])), (5, 90, 90, dict(coarse_downsample=1)), (2, 257, 1023, dict(min_num_blocks=4)),
         (6, 512, 128, dict(residue_len=0)), (6, 512, 128, dict(compute_strain=False)), (3, 512, 128, dict(conf_thresh=0.9))]
for P, H, W, kw in cases:
    s0, s1, _ = tp._synth(feabas_amd, P, H, W, seed=3 + P + H, max_shift=min(10, min(H, W) // 8), warp=0.3)
    try:
        mn = StripBatchMatcher(P, H, W, route='native', **kw); mh = StripBatchMatcher(P, H, W, route='host', **kw)
        rn = mn.match(s0.ptr, s1.ptr); rh = mh.match(s0.ptr, s1.ptr)
        ok = all(np.array_equal(rn[k], rh[k]) for k in ('tx', 'ty', 'conf0', 'valid', 'deformed'))
        gn = StripBatchMatcher.per_pair(rn); gh = StripBatchMatcher.per_pair(rh)
        for p in range(P):
            if not rh['valid'][p]:
                ok &= gn[p]['xy0'] is None; continue
            ex = not mn.last_flags[p]
            for k in ('xy0', 'xy1', 'weight'):
                ok &= gn[p][k].shape == gh[p][k].shape and (np.array_equal(gn[p][k], gh[p][k]) if ex else np.abs(gn[p][k] - gh[p][k]).max() < 1e-6)
            ok &= abs(gn[p]['strain'] - gh[p]['strain']) <= 1e-6 * max(abs(gh[p]['strain']), 1e-4)
        print((P, H, W, kw), 'valid', int(rh['valid'].sum()), 'flags', int((mn.last_flags != 0).sum()), 'rows', rn['pair'].size, 'OK' if ok else 'MISMATCH')
        bad += not ok
        mn.free(); mh.free()
    except Exception as e:
        print((P, H, W, kw), 'raised', type(e).__name__, str(e)[:120])
    s0.free(); s1.free()
print('mismatches', bad)
