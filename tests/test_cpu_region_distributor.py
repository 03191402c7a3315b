"""Host geometry of the alignment-side matcher on triangles (no shapely in the image): the region-aware block distributor
(feabas/matcher.py:894-1043), sub-meshes by triangle connectivity and the dealing of initial matches to the part pairs
(optimizer.py:688-754, 1818-1858).  matplotlib's trifinder locates the points when there is no device context."""
import numpy as np
import pytest

from feabas_amd import constant as const
from feabas_amd import matcher, optimizer
from feabas_amd.mesh import Mesh


def _grid_mesh(x0, y0, nx, ny, h, uid, hole=None):
    xs = x0 + h * np.arange(nx); ys = y0 + h * np.arange(ny)
    vx, vy = np.meshgrid(xs, ys)
    v = np.stack((vx.ravel(), vy.ravel()), axis=-1)
    idx = np.arange(nx * ny).reshape(ny, nx)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tri = np.concatenate((np.stack((a, b, d), -1), np.stack((a, d, c), -1)))
    if hole is not None:
        ctr = v[tri].mean(axis=1)
        keep = ~((ctr[:, 0] > hole[0]) & (ctr[:, 0] < hole[2]) & (ctr[:, 1] > hole[1]) & (ctr[:, 1] < hole[3]))
        tri = tri[keep]
    return v, tri


def _two_islands(uid, shift=(0.0, 0.0)):
    """two rectangles 600 x 400 (one with a 200 x 120 hole) 150 px apart, as ONE mesh"""
    va, ta = _grid_mesh(0, 0, 16, 11, 40.0, uid, hole=(200, 120, 400, 240))
    vb, tb = _grid_mesh(750, 40, 13, 9, 40.0, uid)
    v = np.concatenate((va, vb)) + np.asarray(shift)
    t = np.concatenate((ta, tb + va.shape[0]))
    used = np.unique(t)
    remap = -np.ones(v.shape[0], dtype=np.int64); remap[used] = np.arange(used.size)
    return Mesh(v[used], remap[t], uid=uid)


def test_connected_triangles_boundary_and_submesh():
    m = _two_islands(3.0)
    n, lab = m.connected_triangles()
    assert n == 2
    parts = m.divide_disconnected_mesh()
    assert [p.uid for p in parts] == [3.05, 3.1] and sum(p.num_triangles for p in parts) == m.num_triangles
    assert all(p.connected_triangles()[0] == 1 for p in parts)
    # outline: the outer rectangle + the hole of island A, the rectangle of island B
    be = m.boundary_edges()
    length = np.sum(np.linalg.norm(m.vertices(const.MESH_GEAR_INITIAL)[be[:, 0]] - m.vertices(const.MESH_GEAR_INITIAL)[be[:, 1]], axis=1))
    assert abs(length - (2 * (600 + 400) + 2 * (200 + 120) + 2 * (480 + 320))) < 1e-9
    # sub-mesh keeps every gear and the per-triangle arrays
    m.set_vertices(m.vertices(const.MESH_GEAR_INITIAL) + 1.5, const.MESH_GEAR_MOVING)
    sub = m.submesh(lab == 1)
    assert sub.num_vertices == 13 * 9 and np.allclose(sub.vertices(const.MESH_GEAR_MOVING) - sub.vertices(const.MESH_GEAR_INITIAL), 1.5)
    soft = Mesh(m.vertices(const.MESH_GEAR_INITIAL), m.triangles, tri_model=np.zeros(m.num_triangles, np.int32),
                tri_matmult=np.where(lab == 1, 0.05, 1.0).astype(np.float32))
    assert np.array_equal(soft.triangle_mask_for_stiffness(stiffness_multiplier_threshold=0.1), lab == 0)


@pytest.mark.parametrize('mbd', [0.0, 20.0])
def test_cartesian_region_blocks_lie_inside_both_meshes(mbd):
    m0 = _two_islands(0.0)
    m1 = _two_islands(1.0, shift=(12.0, -9.0))
    sp = 50.0
    b0, b1 = matcher.distribute_matching_blocks(m0, m1, sp, min_boundary_distance=mbd, shrink_factor=(1.0, 0.7), zorder=True)
    c = 0.5 * (b0[:, :2] + b0[:, 2:])
    assert np.array_equal(c, 0.5 * (b1[:, :2] + b1[:, 2:])) and c.shape[0] > 100
    # equal triangle sizes: mesh0 takes the smaller factor (matcher.py:953-956)
    assert np.all(b0[:, 2] - b0[:, 0] == 2 * np.ceil(50 * 0.7 / 2)) and np.all(b1[:, 2] - b1[:, 0] == 2 * np.ceil(50 * 1.0 / 2))
    reg = matcher._RegionPair(m0, m1, const.MESH_GEAR_MOVING)
    assert reg.inside(c).all()
    assert np.all(reg.boundary_distance(c) >= mbd - 1e-9)
    # not in the hole, not in the gap between the islands
    assert not np.any((c[:, 0] > 212) & (c[:, 0] < 400) & (c[:, 1] > 120) & (c[:, 1] < 231))
    assert not np.any((c[:, 0] > 600) & (c[:, 0] < 762))
    # one lattice per island (step = spacing), every admissible lattice point taken
    for sel in (c[:, 0] < 650, c[:, 0] > 650):
        p = c[sel]
        fx = np.mod(p[:, 0] - p[0, 0], sp); fy = np.mod(p[:, 1] - p[0, 1], sp)
        assert np.all(np.minimum(fx, sp - fx) < 1e-9) and np.all(np.minimum(fy, sp - fy) < 1e-9)
        gx, gy = np.meshgrid(np.arange(p[:, 0].min(), p[:, 0].max() + 1, sp), np.arange(p[:, 1].min(), p[:, 1].max() + 1, sp))
        full = np.stack((gx.ravel(), gy.ravel()), axis=-1)
        ok = reg.select(full, erode=mbd)
        assert ok.sum() == p.shape[0]
    # coverage: about area / spacing^2 blocks
    area = (600 * 400 - 200 * 120) + 480 * 320
    assert 0.6 * area / sp ** 2 < c.shape[0] < 1.05 * area / sp ** 2
    with pytest.raises(NotImplementedError):
        matcher.distribute_matching_blocks(m0, m1, sp, dfunc='intersect_triangulation')


def test_no_overlap_gives_no_blocks():
    m0 = _two_islands(0.0)
    m1 = _two_islands(1.0, shift=(5000.0, 0.0))
    b0, b1 = matcher.distribute_matching_blocks(m0, m1, 50.0)
    assert b0.shape == (0, 4) and b1.shape == (0, 4)


def test_initial_matches_are_dealt_to_the_part_pairs():
    m0 = _two_islands(0.0)
    m1 = _two_islands(1.0, shift=(3.0, 2.0))
    rng = np.random.default_rng(0)
    pa = np.stack((rng.uniform(20, 180, 40), rng.uniform(20, 380, 40)), axis=-1)           # island A, left of the hole
    pb = np.stack((rng.uniform(770, 1200, 25), rng.uniform(60, 340, 25)), axis=-1)          # island B
    xy0 = np.concatenate((pa, pb)); xy1 = xy0 + np.array([3.0, 2.0])
    w = rng.uniform(0.5, 1, xy0.shape[0]).astype(np.float32)
    opt = optimizer.SLM([m0, m1], stiffness_lambda=0.5)
    assert opt.add_link_from_coordinates(m0.uid, m1.uid, xy0, xy1, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_INITIAL), weight=w)
    assert opt.divide_disconnected_submeshes(prune_links=True)
    assert len(opt.meshes) == 4 and len(opt.links) == 2
    sizes = sorted(lk.num_matches for lk in opt.links)
    assert sizes == [25, 40]
    for lk in opt.links:
        a, b = lk.meshes
        assert np.floor(a.uid) == 0 and np.floor(b.uid) == 1 and round((a.uid % 1) * 100) == round((b.uid % 1) * 100)     # A with A, B with B
        np.testing.assert_allclose(lk.xy1(gear=const.MESH_GEAR_INITIAL, use_mask=False) - lk.xy0(gear=const.MESH_GEAR_INITIAL, use_mask=False), [[3.0, 2.0]] * lk.num_matches, atol=1e-9)


def test_section_matcher_rejects_unknown_keywords():
    m0 = _two_islands(0.0); m1 = _two_islands(1.0)
    with pytest.raises(TypeError, match='not honoured'):
        matcher.section_matcher(m0, m1, None, None, no_such_option=1)


def _mesh_with_a_refinement_region(uid, shift=(0.0, 0.0), factor=0.25):
    v, t = _grid_mesh(0, 0, 26, 21, 40.0, uid)                               # 1000 x 800
    ctr = v[t].mean(axis=1)
    mids = np.where((ctr[:, 0] > 400) & (ctr[:, 0] < 720) & (ctr[:, 1] > 240) & (ctr[:, 1] < 560), 4, 0).astype(np.int32)
    return Mesh(v + np.asarray(shift), t, uid=uid, material_ids=mids, material_names={'default': 0, 'fold_refine': 4},
                material_area_constraints={'default': 1.0, 'fold_refine': factor})


def test_refinement_regions_get_their_own_lattice():
    """refine_mode of the block distributor (feabas/matcher.py:914-1016): the triangles of a material with area_constraint 0.25
    get a lattice of a quarter of the spacing and blocks of spacing x 0.25^0.5 (refine_box_exp 0.5), the rest of the common
    region the plain lattice WITHOUT what the finer level covered; mode 1 keeps the refinement blocks only, mode 0 ignores the
    materials; the finest level comes first in the list"""
    m0 = _mesh_with_a_refinement_region(0)
    m1 = _mesh_with_a_refinement_region(1, shift=(12.0, -8.0))
    m1.material_ids[:] = 0                                                    # only mesh 0 marks a refinement region
    spacing = 160.0
    in_ref = lambda c: (c[:, 0] > 400) & (c[:, 0] < 720) & (c[:, 1] > 240) & (c[:, 1] < 560)
    b0_none, _ = matcher.distribute_matching_blocks(m0, m1, spacing, refine_mode=0, zorder=False)
    b0_both, b1_both = matcher.distribute_matching_blocks(m0, m1, spacing, refine_mode=2, zorder=False)
    b0_only, _ = matcher.distribute_matching_blocks(m0, m1, spacing, refine_mode='refine_only', zorder=False)
    side = lambda b: b[:, 2] - b[:, 0]
    c_none, c_both, c_only = (0.5 * (b[:, :2] + b[:, 2:]) for b in (b0_none, b0_both, b0_only))
    # mode 0: one lattice of 160 px, blocks of 160 px, some of them inside the marked region
    assert np.all(side(b0_none) == 160) and in_ref(c_none).any()
    # refinement only: a 40-px lattice of 80-px blocks inside the marked region, nothing outside
    assert c_only.shape[0] >= 40 and in_ref(c_only).all() and np.all(side(b0_only) == 80)
    d = np.abs(c_only[:, None, :] - c_only[None, :, :]).sum(axis=-1)
    assert np.isclose(np.min(d[d > 0]), 40.0)
    # both: the refinement blocks first, then the coarse lattice with the marked region left out
    nf = c_only.shape[0]
    np.testing.assert_array_equal(b0_both[:nf], b0_only)
    assert np.all(side(b0_both[nf:]) == 160) and not in_ref(c_both[nf:]).any() and c_both.shape[0] > nf
    assert c_both[nf:].shape[0] < c_none.shape[0]
    assert np.all(side(b1_both[:nf]) == 80)
    # a mesh without named materials: refine_mode changes nothing
    p0 = Mesh(m0.vertices(const.MESH_GEAR_INITIAL), m0.triangles, uid=5)
    a, _ = matcher.distribute_matching_blocks(p0, m1, spacing, refine_mode=2, zorder=False)
    np.testing.assert_array_equal(a, b0_none)


@pytest.mark.parametrize('case', ['boundary', 'refine_both', 'refine_only'])
def test_distribute_matching_blocks_vs_oracle_on_the_host(case):
    """the comparison of tests/test_gpu_renderer.py::test_distribute_matching_blocks_vs_oracle without a device (points are then
    located by matplotlib's trifinder on both sides): the product's blocks equal oracle/region_ref.py's exactly once the oracle
    takes the product's lattice phase part by part"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_renderer import _island_pair
    from oracle import region_ref
    (v0, t0, v1, t1), (M0, M1), mats = _island_pair(np.random.default_rng(11))
    kw = dict(boundary=dict(refine_mode=0, min_boundary_distance=25), refine_both=dict(refine_mode=2, min_boundary_distance=15),
              refine_only=dict(refine_mode=1))[case]
    sp = 110.0
    g0, g1 = matcher.distribute_matching_blocks(M0, M1, sp, gear=const.MESH_GEAR_INITIAL, **kw)
    assert g0.shape[0] > 20
    e0, e1 = region_ref.distribute_matching_blocks(v0, t0, v1, t1, sp, materials=mats, anchor_blocks=g0, res=None, **kw)
    np.testing.assert_array_equal(g0, e0)
    np.testing.assert_array_equal(g1, e1)
