"""Round 5's incident on the CPU box: does a pair of meshes that a diverged relaxation blew up (both meshes of a floating pair
scaled about their middle -- the linearised rotation of the null space, x += theta * (-y, x), IS a scaling by sqrt(1 + theta^2))
drive `distribute_matching_blocks` into a raster that eats the host?  Run under the RSS watchdog with a small limit, once with
the product's 1e8-cell guard removed (the state of the tree the boxes were lost on) and once with it.
usage: python tools/repro_r05_incident.py [scale=60] [limit_gb=6] [guard=0|1]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
limit = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
guard = int(sys.argv[3]) if len(sys.argv) > 3 else 0
os.environ['FEABAS_RSS_LIMIT_GB'] = str(limit)
from feabas_amd import _watchdog, matcher, mesh
from oracle import fem_ref
_watchdog.start()
_watchdog.start_backstop()

S, h = 8192.0, 100.0                       # the section pair of bench.py's section_matcher record
n = int(S / h) + 1
v, t = fem_ref.grid_mesh(n, n, h)
c = v.mean(axis=0)
m0 = mesh.Mesh((v - c) * scale + c, t, uid=0)
m1 = mesh.Mesh((v - c) * scale + c + 3.0, t, uid=1)
if not guard:                              # the tree of the incident had no guard: raster() went straight to np.meshgrid
    src = matcher._RegionPair.raster
    import inspect, textwrap
    code = textwrap.dedent(inspect.getsource(src)).replace('cells > 1e8', 'False')
    ns = {}
    exec(code, matcher.__dict__, ns)
    matcher._RegionPair.raster = ns['raster']
t0 = time.time()
print('scale %g: common region %.3g px wide, raster step %.3g -> %.3g cells; limit %.1f GB, guard %d'
      % (scale, S * scale, 100.0 / 4, (S * scale / 25.0) ** 2, limit, guard), flush=True)
try:
    b0, b1 = matcher.distribute_matching_blocks(m0, m1, 100.0, min_boundary_distance=20, shrink_factor=0.7)
    print('returned %d blocks after %.1f s, RSS %.1f GB' % (b0.shape[0], time.time() - t0, _watchdog.tree_rss_gb()))
except ValueError as e:
    print('refused after %.2f s: %s' % (time.time() - t0, e))
