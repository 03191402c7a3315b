"""where the host time of one section of the config[4] workload goes (cProfile over a few sections)"""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feabas_amd import mesh, optimizer, constant as const, _lib
n = 500; h = 20.0
xs = h * np.arange(n); vx, vy = np.meshgrid(xs, xs); v = np.stack((vx.ravel(), vy.ravel()), axis=-1)
idx = np.arange(n * n).reshape(n, n)
a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
tri = np.concatenate((np.stack((a, b, d), -1), np.stack((a, d, c), -1))).astype(np.int32)
prev = mesh.Mesh(v.copy(), tri, uid=0, locked=True); cur = mesh.Mesh(v.copy(), tri, uid=1); nxt = mesh.Mesh(v.copy(), tri, uid=2, locked=True)
slm = optimizer.SLM([prev, cur, nxt], [], stiffness_lambda=1.0, crosslink_lambda=-1.0)
rng = np.random.default_rng(0)
nl = 50000
def one(k):
    for m_ in (prev, nxt):
        m_.unlock(); m_.set_vertices(v + rng.normal(0, 1.0, v.shape), const.MESH_GEAR_MOVING); m_.lock()
    cur.set_vertices(v.copy(), const.MESH_GEAR_MOVING); cur.set_offset(np.zeros((1, 2)), const.MESH_GEAR_MOVING)
    lk = [(np.sort(rng.integers(0, tri.shape[0], nl)), rng.dirichlet((1, 1, 1), nl), rng.uniform(0.3, 1.0, nl).astype(np.float32)) for _ in range(2)]
    slm.links = [optimizer.Link(m0, m1, tid, tid, B, B, weight=w) for (m0, m1), (tid, B, w) in zip(((prev, cur), (cur, nxt)), lk)]
    t0 = time.time(); slm.optimize_linear(tol=1e-4); return time.time() - t0
one(0); one(1)
pr = cProfile.Profile(); pr.enable()
ts = [one(k) for k in range(4)]
pr.disable()
print('optimize_linear per section', np.round(ts, 3), 'iters', slm.last_solve['iters'])
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
pstats.Stats(pr).print_callers('reduce')
pstats.Stats(pr).print_callers('_collections_abc')
