"""Generates tests/golden/floating_island_pair_systems.npz: the three linear systems the ORACLE's section-matcher loop
(oracle/region_ref.section_match: matcher.py:370-427, 430-778 restated) solves for the island pair of
tests/test_gpu_renderer.py::test_section_matcher_floating_pair_vs_oracle with BOTH sections free.  Not a reference fixture (the
reference is not involved): a CPU-made input for the device PCG on floating systems -- two link-connected floating sub-systems
(4 translations in the null space), A and b with the float32 noise of the reference's arithmetic that fem_ref restates.
usage (repo root, CPU, ~30 s): python tests/golden/make_floating_systems.py"""
import os
import sys

import numpy as np
from scipy import sparse
from scipy.ndimage import map_coordinates

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_gpu_renderer as T                              # noqa: E402  (_island_pair, _texture: host-only helpers)
from oracle import fem_ref, region_ref                     # noqa: E402

rng = np.random.default_rng(17)
(v0, t0, v1, t1), _, _ = T._island_pair(rng)
SH, SW = 600, 1080
base = T._texture(rng, SH, SW)
yy, xx = np.meshgrid(np.arange(SH, dtype=np.float64), np.arange(SW, dtype=np.float64), indexing='ij')
ux = 3.0 * np.sin(2 * np.pi * yy / 700.0 + 0.4) + 1.0 * (xx / SW) ** 2
uy = 2.5 * np.cos(2 * np.pi * xx / 900.0) - 1.0 * (xx / SW) * (yy / SH)
img1 = np.clip(np.rint(map_coordinates(base.astype(np.float32), [yy + uy, xx + ux], order=1, mode='nearest')), 0, 255).astype(np.uint8)
dump = []
solve = region_ref._solve_jacobi_krylov_limit


def hook(A, b):
    dump.append((sparse.csr_matrix(A), np.array(b, dtype=np.float64)))
    return solve(A, b)


region_ref._solve_jacobi_krylov_limit = hook
r0 = fem_ref.RefMesh(v0, t0, uid=0); r1 = fem_ref.RefMesh(v1, t1, uid=1)
region_ref.section_match(r0, r1, base, img1, compute_strain=True, batch_size=100, spacings=[150, 60], sigma=2.5, conf_thresh=0.3, residue_len=3.0,
                         min_boundary_distance=12, stiffness_lambda=0.5)
out = {}
for k, (A, b) in enumerate(dump):
    A.sort_indices()
    out['indptr%d' % k] = A.indptr.astype(np.int64); out['indices%d' % k] = A.indices.astype(np.int32); out['data%d' % k] = A.data; out['b%d' % k] = b
path = os.path.join(ROOT, 'tests', 'golden', 'floating_island_pair_systems.npz')
np.savez_compressed(path, **out)
print('wrote', path, [d[0].shape for d in dump])
