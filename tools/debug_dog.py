"""where does fb_dog differ from the oracle? (rows / columns of the bad pixels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feabas_amd import common
from oracle import ncc_ref
rng = np.random.default_rng(3)
for (h, w, s) in ((4096, 510, 2.5), (2048, 255, 1.25), (97, 131, 2.5), (7, 300, 1.25), (510, 4096, 2.5), (333, 217, 3.0)):
    img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    got = common.masked_dog_filter(img, s)
    exp = ncc_ref.masked_dog_filter(img, s)
    err = np.abs(got - exp)
    bad = ~(err <= 1e-4 * np.abs(exp).max())
    ys, xs = np.nonzero(bad)
    print(h, w, s, 'bad', bad.sum(), 'max', np.nanmax(err), 'nan', np.isnan(got).sum())
    if bad.any():
        print('  rows', np.unique(ys)[:40], '... n', np.unique(ys).size)
        print('  cols', np.unique(xs)[:40], '... n', np.unique(xs).size)
        print('  rows mod 64', np.unique(ys % 64)[:64])
