"""where the time of bench.py::bench_align (bboxes_mesh_renderer_matcher on 512 blocks of 280 x 280) goes: host profile + kernel profile"""
import cProfile, pstats, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from feabas_amd import _lib
lib, ctx = _lib.load(), _lib.ctx()
print({k: v for k, v in bench.bench_align(lib, ctx, _lib).items() if k in ('value', 'ms_per_call')})
_lib.check(lib.fb_prof_reset(ctx)); _lib.check(lib.fb_prof_enable(ctx, 1))
pr = cProfile.Profile(); pr.enable()
out = bench.bench_align(lib, ctx, _lib)
pr.disable()
_lib.check(lib.fb_prof_enable(ctx, 0))
print({k: v for k, v in out.items() if k in ('value', 'ms_per_call')})
for k, v in sorted(_lib.prof_snapshot().items(), key=lambda kv: -kv[1][1]):
    print(f'   {k:28s} launches {v[0]:4d}  {v[1]:8.3f} ms')
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
