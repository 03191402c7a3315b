"""Parity at the sizes BASELINE.json names ("full-size parity holes"): the whole pair pipeline on 4096 x 510 /
510 x 4096 strips against the oracle pipeline, the 1 002 528-DoF system of config[2] through its true residual recomputed
on the host plus an exact-solve comparison at 354 x 354, and the fixed point of the Newton-Raphson driver."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from oracle import fem_ref, pipeline_ref

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_pipeline import _warped_pair          # noqa: E402  (the texture / warp generator of the pipeline tests)
from conftest import load_golden                    # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('H,W', [(4096, 510), (510, 4096)])
def test_pair_pipeline_at_the_4k_strip_size_vs_oracle(fb, H, W):
    """config[1] strip shape, both orientations: one rigid pair (odd offset: the rigid relaxation branch) and one pair
    with a 2 px warp (deformed-mesh branch) through StripBatchMatcher against pipeline_ref.match_pair: the block grid
    of the reference (4 coarse + 385 fine blocks), integer displacements bit-exact, sub-pixel coordinates, weights, strain"""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    pairs = [_warped_pair(H, W, 21, (7, -5), 0.0), _warped_pair(H, W, 22, (-4, 9), 2.0)]
    s0 = np.stack([p[0] for p in pairs]); s1 = np.stack([p[1] for p in pairs])
    d0 = _lib.DeviceBuffer.from_array(s0); d1 = _lib.DeviceBuffer.from_array(s1)
    m = StripBatchMatcher(2, H, W, residue_len=2.0)
    np.testing.assert_allclose(m.spacings, [1024.0, 75.0], rtol=1e-12)
    got = StripBatchMatcher.per_pair(m.match(d0.ptr, d1.ptr))
    ndef = 0
    for p in range(2):
        exp = pipeline_ref.match_pair(s0[p], s1[p], residue_len=2.0)
        g = got[p]
        assert (g['tx'], g['ty']) == (exp['tx'], exp['ty'])
        assert abs(g['conf0'] - exp['conf0']) < 1e-4
        assert g['deformed'] == bool(exp.get('deformed', False))
        ndef += int(g['deformed'])
        assert g['xy0'].shape == exp['xy0'].shape and g['xy0'].shape[0] > 300          # 385 fine blocks, most of them confident
        np.testing.assert_array_equal(np.round(g['xy1'] - g['xy0']), np.round(exp['xy1'] - exp['xy0']))
        # rigid and deformed pairs alike: 1e-4 px, 1e-4 on weights and strain (measured, tools/diag_deformed_bars.py: at most 5e-5 px /
        # 3e-5 / 4e-6 relative on every deformed test pair, whether the device relaxes to 1e-9 or to 1e-13)
        tol = 1e-4
        np.testing.assert_allclose(g['xy0'], exp['xy0'], atol=tol)
        np.testing.assert_allclose(g['xy1'], exp['xy1'], atol=tol)
        np.testing.assert_allclose(g['weight'], exp['weight'], atol=1e-4)
        np.testing.assert_allclose(g['strain'], exp['strain'], rtol=1e-4, atol=1e-8)
    assert ndef == 1
    m.free(); d0.free(); d1.free()


@pytest.mark.parametrize('H,W', [(4096, 510), (510, 4096)])
def test_pair_pipeline_with_the_coarse_level_at_full_resolution_vs_oracle(fb, H, W):
    """coarse_downsample = 1 on a 4k strip: the global translation is ONE correlation at FFT 8192 x 1024 (1024 x 8192) -- the
    8192-point forms of the power-of-two core (split columns / rows of 8192 points, fb_ncc_p2.inc), reached through the
    pipeline's own entry; everything after it as in the test above"""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    pairs = [_warped_pair(H, W, 31, (9, -6), 0.0)]
    s0 = np.stack([p[0] for p in pairs]); s1 = np.stack([p[1] for p in pairs])
    d0 = _lib.DeviceBuffer.from_array(s0); d1 = _lib.DeviceBuffer.from_array(s1)
    m = StripBatchMatcher(1, H, W, residue_len=2.0, coarse_downsample=1)
    g = StripBatchMatcher.per_pair(m.match(d0.ptr, d1.ptr))[0]
    exp = pipeline_ref.match_pair(pairs[0][0], pairs[0][1], residue_len=2.0, coarse_downsample=1)
    assert (g['tx'], g['ty']) == (exp['tx'], exp['ty']) and abs(exp['tx']) + abs(exp['ty']) > 10
    assert abs(g['conf0'] - exp['conf0']) < 1e-4
    assert g['xy0'].shape == exp['xy0'].shape and g['xy0'].shape[0] > 300
    np.testing.assert_allclose(g['xy0'], exp['xy0'], atol=1e-4)
    np.testing.assert_allclose(g['xy1'], exp['xy1'], atol=1e-4)
    np.testing.assert_allclose(g['weight'], exp['weight'], atol=1e-4)
    m.free(); d0.free(); d1.free()


def _grid_system(fb, grid, nlinks, seed=0):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.build_fem_system(grid, nlinks, seed=seed)


def test_million_dof_relaxation_true_residual_on_the_host(fb):
    """config[2]: 708 x 708 nodes = 1 002 528 DoF, 200 k links.  optimize_linear(tol=1e-4) moves the free mesh; A and b
    are downloaded (fb_sys_get) and ||A d - b|| / ||b|| is recomputed with scipy on the host from the field the MESH
    holds afterwards -- the whole chain assembly -> PCG -> Mesh.set_field at the full size."""
    from feabas_amd import _lib
    from feabas_amd.mesh import bsr_download
    slm = _grid_system(fb, 708, 200000)
    m1 = slm.meshes[1]
    v0 = m1.vertices_w_offset(1).copy()
    cost = slm.optimize_linear(tol=1e-4)
    assert slm._nv * 2 == 1002528
    assert cost[1] <= 1e-4 * cost[0]
    d = (m1.vertices_w_offset(1) - v0).ravel()
    A = bsr_download(slm._sys, 4, slm._nv, slm._nnzb)
    b = np.empty(2 * slm._nv)
    _lib.check(_lib.load().fb_sys_get(_lib.ctx(), slm._sys, 5, _lib.ptr(b)))
    assert abs(A - A.T).max() <= 1e-12 * abs(A).max()
    rel = np.linalg.norm(A @ d - b) / np.linalg.norm(b)
    assert rel <= 1e-4, rel
    assert abs(np.linalg.norm(b) - cost[0]) <= 1e-9 * cost[0]
    # the imposed field is smooth (5 sin, 4 cos): the relaxed mesh follows it
    assert 3.0 < np.abs(d).max() < 5.5


def test_quarter_million_dof_relaxation_vs_exact_oracle_solve(fb):
    """354 x 354 nodes (250 632 DoF), 50 k links: node displacements of SLM.optimize_linear against the oracle's
    assembly (pinned by G4-G8) solved with a sparse LU, 1e-4 of the largest displacement"""
    n, nl = 354, 50000
    slm = _grid_system(fb, n, nl)
    m0, m1 = slm.meshes
    lk = slm.links[0]
    v = m1.vertices(0).copy()
    r0 = fem_ref.RefMesh(m0.vertices(0), m0.triangles, uid=0, locked=True)
    r1 = fem_ref.RefMesh(v, m1.triangles, uid=1)
    rl = fem_ref.RefLink(r0, r1, lk._tid0, lk._tid1, lk._B0, lk._B1, weight=lk.weight(use_mask=False))
    ref_cost = fem_ref.optimize_linear([r0, r1], [rl], exact=True)
    cost = slm.optimize_linear(tol=1e-10)
    got = m1.vertices_w_offset(1) - v
    exp = r1.vertices_w_offset(1) - v
    scale = np.abs(exp).max()
    assert scale > 3.0
    assert np.abs(got - exp).max() <= 1e-4 * scale, np.abs(got - exp).max() / scale
    assert abs(cost[0] - ref_cost[0]) <= 1e-6 * ref_cost[0]


def test_newton_raphson_fixed_point_vs_oracle(fb):
    """optimize_Newton_Raphson on the mixed-material mesh of golden G12 (engineering + St-Venant-Kirchhoff +
    Neo-Hookean triangles, element maths pinned by G10 / G12) pulled by links to a displaced locked twin, against the
    oracle's exact Newton iteration: same fixed point to 1e-4 of the motion"""
    g = load_golden('g12_mixed_materials.npz')
    v, t = g['v'], g['t']
    rng = np.random.default_rng(3)
    disp = 0.6 * (g['vmov'] - v) + np.array([[0.4, -0.3]])
    n = 300
    tid = rng.integers(0, t.shape[0], n); B = rng.dirichlet((1, 1, 1), n)
    w = rng.uniform(0.4, 1.0, n).astype(np.float32)
    m0 = fb.mesh.Mesh(v + disp, t, uid=0, locked=True)
    m1 = fb.mesh.Mesh(v.copy(), t, stiffness_multiplier=g['mult'], tri_model=g['model'], tri_nu=g['nu'],
                      tri_matmult=g['matmult'].astype(np.float32), uid=1)
    slm = fb.optimizer.SLM([m0, m1], [fb.optimizer.Link(m0, m1, tid, tid, B, B, weight=w)], stiffness_lambda=1.0, crosslink_lambda=1.0)
    c0, c1 = slm.optimize_Newton_Raphson(max_newtonstep=8, tol=1e-9)
    r0 = fem_ref.RefMesh(v + disp, t, uid=0, locked=True)
    r1 = fem_ref.RefMesh(v.copy(), t, uid=1)
    rl = fem_ref.RefLink(r0, r1, tid, tid, B, B, weight=w)
    costs = fem_ref.newton_fixed_point(r0, r1, [rl], g['mult'], g['model'], g['nu'], g['matmult'].astype(np.float32))
    assert costs[-1] <= 1e-6 * costs[0] and len(costs) >= 3               # really non-linear: more than one Newton step
    assert abs(c0 - costs[0]) <= 1e-5 * costs[0]                           # same first out-of-balance force
    got = m1.vertices_w_offset(1) - v
    exp = r1.vertices_w_offset(1) - v
    scale = np.abs(exp).max()
    assert scale > 0.1
    assert np.abs(got - exp).max() <= 1e-4 * scale, np.abs(got - exp).max() / scale


def test_corner_pairs_batch_vs_oracle(fb):
    """config[3] corner overlaps (510 x 510, the diagonal neighbours of a 20 x 20 tile grid): a batch of 64 synthetic pairs from
    the generator bench.py uses (offsets in +-20 px, 0.4 px smooth warp) through StripBatchMatcher against
    pipeline_ref.match_pair pair by pair -- one spacing (75), one padded sub-pixel round: same matches, same weights -- and the
    fraction of matches within half a pixel of the generator's INTEGER offset is the same on both sides.  That fraction is
    what bench.py reports as stitch_sections.corner.matches_within_half_px_of_truth (0.979 against 0.99999 on the edge
    strips).  It is not a miss rate: the generator adds a smooth warp of 0.4 px amplitude that the integer "truth" ignores,
    the distances are 0.26 px in the median, 0.505 at the 99th percentile and never above 0.57 (tools/corner_probe.py on 256
    pairs, 12 544 matches) -- warp plus the +-0.5 clip of the sub-pixel fit (matcher.py:84-106).  On an edge strip the
    coarse round and the mesh relaxation absorb the warp's low-frequency part first; a corner has a single round."""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    lib, ctx = _lib.load(), _lib.ctx()
    P, H, W = 64, 510, 510
    s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
    _lib.check(lib.fb_synth_strips_dev(ctx, P, 30000000, H, W, 2027, 20, 1, 0.4, s0.ptr, s1.ptr, sh.ptr))
    h0 = s0.to_array((P, H, W), np.uint8); h1 = s1.to_array((P, H, W), np.uint8); shift = sh.to_array((P, 2), np.int32)
    m = StripBatchMatcher(P, H, W, residue_len=2.0)
    assert len(m.spacings) == 1 and abs(m.spacings[0] - 75.0) < 1e-9
    got = StripBatchMatcher.per_pair(m.match(s0.ptr, s1.ptr))
    n_got = n_exp = in_got = in_exp = 0
    for p in range(P):
        exp = pipeline_ref.match_pair(h0[p], h1[p], residue_len=2.0)
        g = got[p]
        assert (g['tx'], g['ty']) == (exp['tx'], exp['ty'])
        assert g['xy0'].shape == exp['xy0'].shape
        np.testing.assert_array_equal(np.round(g['xy1'] - g['xy0']), np.round(exp['xy1'] - exp['xy0']))
        tol = 1e-4
        np.testing.assert_allclose(g['xy0'], exp['xy0'], atol=tol)
        np.testing.assert_allclose(g['xy1'], exp['xy1'], atol=tol)
        np.testing.assert_allclose(g['weight'], exp['weight'], atol=1e-4)
        dg = g['xy1'] - g['xy0'] + shift[p]; de = exp['xy1'] - exp['xy0'] + shift[p]
        n_got += dg.shape[0]; n_exp += de.shape[0]
        in_got += int(np.sum(np.abs(dg).max(axis=1) < 0.5)); in_exp += int(np.sum(np.abs(de).max(axis=1) < 0.5))
    assert n_got == n_exp and n_got > 30 * P
    assert in_got == in_exp
    assert 0.9 < in_got / n_got <= 1.0
    worst = max(np.abs(got[p]['xy1'] - got[p]['xy0'] + shift[p]).max() for p in range(P))
    assert worst < 0.75                                  # warp amplitude + clip: nobody is a whole pixel off
    m.free(); s0.free(); s1.free(); sh.free()
