import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from feabas_amd import _lib
from feabas_amd.stitch_pipeline import StripBatchMatcher
lib, ctx = _lib.load(), _lib.ctx()
P, H, W = 256, 510, 510
s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
_lib.check(lib.fb_synth_strips_dev(ctx, P, 30000000, H, W, 2027, 20, 1, 0.4, s0.ptr, s1.ptr, sh.ptr))
shift = sh.to_array((P, 2), np.int32)
m = StripBatchMatcher(P, H, W, residue_len=2.0)
got = StripBatchMatcher.per_pair(m.match(s0.ptr, s1.ptr))
bad_pos = []; tot = 0; err = []
for p in range(P):
    g = got[p]
    d = g['xy1'] - g['xy0'] + shift[p]
    tot += d.shape[0]
    bad = np.abs(d).max(axis=1) >= 0.5
    err.append(np.abs(d).max(axis=1))
    for q in np.flatnonzero(bad):
        bad_pos.append((p, g['xy0'][q], d[q], g['weight'][q], shift[p]))
err = np.concatenate(err)
print('matches', tot, 'outside half px', len(bad_pos), 'fraction inside', 1 - len(bad_pos) / tot)
print('error quantiles 50/90/99/max', np.quantile(err, [0.5, 0.9, 0.99, 1.0]))
for b in bad_pos[:12]:
    print(b)
