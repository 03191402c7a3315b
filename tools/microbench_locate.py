"""Mesh.tri_finder on a raster of 468 x 468 points over a 13 k-triangle mesh (the section matcher's second-round region raster):
wall per call; FEABAS_HIP_LOCATE_PLAIN=1 takes the plain walk"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from scipy.spatial import Delaunay
from feabas_amd import _lib
from feabas_amd.mesh import Mesh
rng = np.random.default_rng(0)
S = 8192
g = np.arange(0, S, 100.0)
gx, gy = np.meshgrid(np.append(g, S - 1), np.append(g, S - 1))
v = np.stack((gx.ravel(), gy.ravel()), -1).astype(np.float64)
inner = (v[:, 0] > 0) & (v[:, 0] < S - 1) & (v[:, 1] > 0) & (v[:, 1] < S - 1)
v[inner] += rng.uniform(-0.3, 0.3, (int(inner.sum()), 2)) * 100
m = Mesh(v, Delaunay(v).simplices.astype(np.int32), uid=0)
_lib.ctx()
xs = np.arange(8.75, S, 17.5)
pts = np.stack(np.meshgrid(xs, xs), -1).reshape(-1, 2)
for _ in range(3):
    t0 = time.perf_counter(); tid = m.tri_finder(pts, gear=1); dt = time.perf_counter() - t0
    print(f'{pts.shape[0]} points, {m.num_triangles} triangles: {1e3 * dt:.2f} ms, inside {(tid >= 0).mean():.3f}')
lib, ctx = _lib.load(), _lib.ctx()
lib.fb_prof_reset(ctx); lib.fb_prof_enable(ctx, 1)
m.tri_finder(pts, gear=1)
lib.fb_prof_enable(ctx, 0)
print({k: (v_[0], round(v_[1], 3)) for k, v_ in _lib.prof_snapshot().items()})
