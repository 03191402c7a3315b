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
