"""The device PCG on FLOATING systems, solver only (nothing is applied to a mesh): the three relaxations of the island pair of
tests/test_gpu_renderer.py::test_section_matcher_vs_oracle with both sections free (tests/golden/floating_island_pair_systems.npz,
dumped from the oracle's loop on the CPU: 4 null vectors -- two floating sub-systems --, soft rotations at 4e-9 of the
largest eigenvalue) and a floating 8192^2 section pair built here, at tolerances down to what doubles cannot give.
FEABAS_HIP_PCG_TRACE=1 prints the legs.  usage: python tools/probe_pcg_floating.py [best=1]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['FEABAS_HIP_PCG_BEST'] = sys.argv[1] if len(sys.argv) > 1 else '1'
os.environ.setdefault('FEABAS_HIP_PCG_TRACE', '1')
import numpy as np
from scipy import sparse
from feabas_amd import optimizer, _watchdog
from oracle import fem_ref
_watchdog.start(8.0)


def cases():
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'floating_island_pair_systems.npz'))
    for k in range(3):
        n = z['b%d' % k].size
        yield 'island pair, relaxation %d' % k, sparse.csr_matrix((z['data%d' % k], z['indices%d' % k], z['indptr%d' % k]), shape=(n, n)), z['b%d' % k]
    rng = np.random.default_rng(3)
    v, t = fem_ref.grid_mesh(83, 83, 100.0)
    m0 = fem_ref.RefMesh(v, t, uid=0); m1 = fem_ref.RefMesh(v + 0.0, t, uid=1)
    n = 4000
    tid = rng.integers(0, t.shape[0], n); B = rng.dirichlet((1, 1, 1), n)
    B1 = B + rng.normal(0, 0.02, B.shape); B1 /= B1.sum(axis=1, keepdims=True)
    w = rng.uniform(0.3, 1, n).astype(np.float32)
    A, b, _ = fem_ref.linear_system([m0, m1], [fem_ref.RefLink(m0, m1, tid, tid, B, B1, weight=w)], 0.5, -1.0, 0, 1, 1)
    yield 'floating 8192^2 pair, 4000 links', sparse.csr_matrix(A), np.asarray(b, dtype=np.float64)


for name, A, b in cases():
    bn = np.linalg.norm(b)
    xc, itc, relc = fem_ref.pcg(A, b, rtol=1e-13, maxiter=200000)
    print('== %s: n %d, CPU pcg to %.1e in %d iterations, |x|max %.4g' % (name, b.size, relc, itc, np.abs(xc).max()), flush=True)
    for tol in (1e-9, 1e-11, 1e-13, 1e-15):
        t0 = time.time()
        x = optimizer.solve(A, b, tol=tol, M='jacobi')
        rel = np.linalg.norm(A @ x - b) / bn
        print('   tol %.0e: true relres %.3e  |x|max %.4g  max|x - x_cpu| / |x_cpu|max %.3e   %.2f s'
              % (tol, rel, np.abs(x).max(), np.abs(x - xc).max() / np.abs(xc).max(), time.time() - t0), flush=True)
        assert np.all(np.isfinite(x))
