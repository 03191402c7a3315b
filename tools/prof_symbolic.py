"""host profile of the FIRST SLM._assemble (symbolic phase included) on the 1.0 M-DoF system of bench.py's fem record"""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import bench
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
for rep in range(2):
    slm = bench.build_fem_system(708, 200000)
    pr = cProfile.Profile(); pr.enable()
    t = time.time(); slm._assemble(0, 1, 1); _lib.check(lib.fb_sync(ctx)); dt = time.time() - t
    pr.disable()
    print('first assemble', round(1e3 * dt, 2), 'ms')
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
