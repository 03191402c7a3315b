"""first and later sections of one config[4] unit on one thread: optimize_linear wall per section (FEABAS_HIP_FEM_TRACE=1
adds the steps of the symbolic phase), cProfile of a first section"""
import cProfile, pstats, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from feabas_amd import mesh, optimizer, constant as const, _lib
n = 500; h = 20.0
xs = h * np.arange(n); vx, vy = np.meshgrid(xs, xs); v = np.stack((vx.ravel(), vy.ravel()), axis=-1)
idx = np.arange(n * n).reshape(n, n)
a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
tri = np.concatenate((np.stack((a, b, d), -1), np.stack((a, d, c), -1))).astype(np.int32)
rng = np.random.default_rng(0)
nl = 50000
inputs = [(v + rng.normal(0, 1.0, v.shape), v + rng.normal(0, 1.0, v.shape),
           [(np.sort(rng.integers(0, tri.shape[0], nl)), rng.dirichlet((1, 1, 1), nl), rng.uniform(0.3, 1.0, nl).astype(np.float32)) for _ in range(2)]) for _ in range(4)]
zero = np.zeros((1, 2))


def unit():
    prev = mesh.Mesh(v.copy(), tri, uid=0, locked=True); cur = mesh.Mesh(v.copy(), tri, uid=1); nxt = mesh.Mesh(v.copy(), tri, uid=2, locked=True)
    return prev, cur, nxt, optimizer.SLM([prev, cur, nxt], [], stiffness_lambda=1.0, crosslink_lambda=-1.0)


def one(u, k):
    prev, cur, nxt, slm = u
    vp, vn, lk = inputs[k]
    t0 = time.perf_counter()
    for m_, vv in ((prev, vp), (nxt, vn)):
        m_.unlock(); m_.set_vertices(vv, const.MESH_GEAR_MOVING); m_.lock()
    cur.set_vertices(v.copy(), const.MESH_GEAR_MOVING); cur.set_offset(zero, const.MESH_GEAR_MOVING)
    slm.links = [optimizer.Link(m0, m1, tid, tid, B, B, weight=w) for (m0, m1), (tid, B, w) in zip(((prev, cur), (cur, nxt)), lk)]
    t1 = time.perf_counter()
    slm.optimize_linear(tol=1e-4)
    t2 = time.perf_counter()
    out = cur.vertices_w_offset(const.MESH_GEAR_MOVING) - v
    t3 = time.perf_counter()
    return [round(1e3 * x, 2) for x in (t1 - t0, t2 - t1, t3 - t2)]


for rep in range(3):
    u = unit()
    print('unit', rep, 'sections (set-up, optimize_linear, read-back) ms:', [one(u, k) for k in range(4)], flush=True)
u = unit()
pr = cProfile.Profile(); pr.enable(); one(u, 0); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(16)
