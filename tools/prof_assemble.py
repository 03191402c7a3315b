"""host profile of SLM._assemble (numeric, pattern cached) on the 1.0 M-DoF system of bench.py's fem record"""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import bench
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
slm = bench.build_fem_system(708, 200000)
slm._assemble(0, 1, 1)
for _ in range(3):
    t = time.time(); slm._assemble(0, 1, 1); _lib.check(lib.fb_sync(ctx)); print('assemble', round(1e3 * (time.time() - t), 2), 'ms')
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    slm._assemble(0, 1, 1)
_lib.check(lib.fb_sync(ctx))
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
