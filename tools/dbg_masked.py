import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if len(sys.argv) > 1:
    sys.path.insert(0, 'tests')
    import feabas_amd as fb
    from test_gpu_pipeline import _warped_pair
    H, W = 1024, 256
    s0, s1 = _warped_pair(H, W, 77, shift=(-5, 3), warp=0.3)
    mask0 = np.ones((H, W), dtype=bool); mask0[100:180, 30:120] = False; mask0[700:, :40] = False
    mask1 = np.ones((H, W), dtype=bool); mask1[400:520, 150:] = False
    s0 = s0.copy(); s0[~mask0] = 0
    s1 = s1.copy(); s1[~mask1] = 0
    xy0, xy1, wt, strain, phtm = fb.matcher.stitching_matcher(s0, s1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2, mask0=mask0, mask1=mask1, compute_photometric=True)
    np.save(sys.argv[1], np.concatenate((xy0, xy1, wt[:, None]), axis=1))
else:
    for tag, env in (('exact', {'FEABAS_HIP_FFT_EXACT': '1'}), ('promoted', {})):
        subprocess.run([sys.executable, __file__, f'/tmp/dbg_{tag}.npy'], env=dict(os.environ, **env), check=True)
    a, b = np.load('/tmp/dbg_exact.npy'), np.load('/tmp/dbg_promoted.npy')
    print(a.shape, b.shape)
    ka = {tuple(np.round(r[:2], 1)): r for r in a}; kb = {tuple(np.round(r[:2], 1)): r for r in b}
    for k in sorted(set(ka) ^ set(kb)):
        print('only in', 'exact' if k in ka else 'promoted', (ka.get(k) if k in ka else kb.get(k)))
    print('max weight diff on common', max(abs(ka[k][4] - kb[k][4]) for k in set(ka) & set(kb)))
