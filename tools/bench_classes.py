"""kernel-only rates of the xcorr_fft shape classes (bench.bench_xcorr_classes) without the rest of bench.py; optional name filter"""
import json
import sys

import bench
from feabas_amd import _lib

lib, ctx = _lib.load(), _lib.ctx()
out = bench.bench_xcorr_classes(lib, ctx, _lib, only=sys.argv[1:] or None)
for k, v in out.items():
    if True:
        print(k, json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()}))
