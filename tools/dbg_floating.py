import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import feabas_amd
from feabas_amd import _lib, mesh, optimizer
from feabas_amd.mesh import bsr_download
from oracle import fem_ref, region_ref
rng = np.random.default_rng(5)
v, t = fem_ref.grid_mesh(14, 11, 10.0)
m0 = mesh.Mesh(v, t, uid=0); m1 = mesh.Mesh(v + rng.normal(0, 0.4, v.shape), t, uid=1)
n = 120
tid = rng.integers(0, t.shape[0], n); B = rng.dirichlet((1, 1, 1), n)
w = rng.uniform(0.5, 1, n).astype(np.float32)
slm = optimizer.SLM([m0, m1], [optimizer.Link(m0, m1, tid, tid, B, B, weight=w)], stiffness_lambda=0.5)
v0a, v1a = m0.vertices_w_offset(1).copy(), m1.vertices_w_offset(1).copy()
cost = slm.optimize_linear(tol=1e-12)
d = np.concatenate(((m0.vertices_w_offset(1) - v0a).ravel(), (m1.vertices_w_offset(1) - v1a).ravel()))
A = bsr_download(slm._sys, 4, slm._nv, slm._nnzb)
b = np.empty(2 * slm._nv); _lib.check(_lib.load().fb_sys_get(_lib.ctx(), slm._sys, 5, _lib.ptr(b)))
print('iters', slm.last_solve, 'cost', cost)
x = region_ref._solve_jacobi_krylov_limit(A, b)
print('|Ad-b|/|b|', np.linalg.norm(A @ d - b) / np.linalg.norm(b), '|Ax-b|/|b|', np.linalg.norm(A @ x - b) / np.linalg.norm(b))
print('max|d - x|', np.abs(d - x).max(), 'max|x|', np.abs(x).max())
diff = (d - x).reshape(-1, 2)
print('diff mean', diff.mean(axis=0), 'std', diff.std(axis=0))
Md = A.diagonal()
e = np.zeros((2 * slm._nv, 2)); e[0::2, 0] = 1; e[1::2, 1] = 1
print('M-orth of d:', e.T @ (Md * d), ' of x:', e.T @ (Md * x), ' I-orth of d', e.T @ d)
# oracle system of the same pair
r0 = fem_ref.RefMesh(v, t, uid=0); r1 = fem_ref.RefMesh(v1a, t, uid=1)
lk = fem_ref.RefLink(r0, r1, tid, tid, B, B, weight=w)
Ao, bo, _ = fem_ref.linear_system([r0, r1], [lk], 0.5, -1.0)
Ao = 0.5 * (Ao + Ao.T)
print('A vs oracle A', abs(A - Ao).max() / abs(Ao).max(), 'b', np.abs(b - bo).max() / np.abs(bo).max())
