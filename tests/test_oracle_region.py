"""the oracle's restatement of the region-aware block distributor (oracle/region_ref.py, feabas/matcher.py:894-1043) on cases
worked by hand from the reference's statements (shapely is not in the image: no golden vectors can be made for it)"""
import numpy as np
import pytest

from oracle import fem_ref, ncc_ref, region_ref


def _rect_mesh(x0, y0, x1, y1, h):
    nx, ny = int(round((x1 - x0) / h)) + 1, int(round((y1 - y0) / h)) + 1
    v, t = fem_ref.grid_mesh(nx, ny, h, origin=(x0, y0))
    return v, t


def test_lattice_of_two_overlapping_rectangles_by_hand():
    """region0 = [0, 400] x [0, 300], region1 = [100, 520] x [-50, 260]: reg_crx0 = [100, 400] x [0, 260].  Its representative
    point is the middle of the scan line through y = 130 (the middle of the bounds; the nearest vertex ordinates are 0 and
    260): (250, 130).  spacing 100 (matcher.py:1030-1033): rx_mn = 250 - ((250 - 100) // 100) * 100 = 150, ry_mn = 130 - 100 = 30;
    np.arange(150, 400, 100) x np.arange(30, 260, 100) -> 3 x 3 centres, all inside; blocks of half side ceil(100 / 2) = 50"""
    v0, t0 = _rect_mesh(0, 0, 400, 300, 20.0)
    v1, t1 = _rect_mesh(100, -50, 520, 260, 30.0)[0], _rect_mesh(100, -50, 520, 260, 30.0)[1]
    # (the second grid is 420 x 310: 15 x 11.33 cells of 30 -> the helper rounds; make it exact instead)
    v1, t1 = fem_ref.grid_mesh(15, 32, 10.0, origin=(100.0, -50.0)); v1 = v1 * np.array([[3.0, 1.0]]) - np.array([[200.0, 0.0]])
    assert v1[:, 0].min() == 100 and v1[:, 0].max() == 520 and v1[:, 1].min() == -50 and v1[:, 1].max() == 260
    b0, b1 = region_ref.distribute_matching_blocks(v0, t0, v1, t1, 100.0, res=0.5)
    cx, cy = np.meshgrid([150.0, 250.0, 350.0], [30.0, 130.0, 230.0])
    exp = np.stack((cx.ravel(), cy.ravel()), axis=-1)
    ctr = 0.5 * (b0[:, :2] + b0[:, 2:])
    assert sorted(map(tuple, ctr)) == sorted(map(tuple, exp))
    np.testing.assert_array_equal(b0[:, 2:] - b0[:, :2], 100.0)
    np.testing.assert_array_equal(b0, b1)
    # z-order of the rounded lattice indices (matcher.py:1005-1010)
    order = ncc_ref.z_order(np.round((ctr - ctr.min(axis=0)) / 100.0))
    np.testing.assert_array_equal(order, np.arange(9))
    # shrink_factor as a pair: the mesh with the larger mean triangle gets the larger factor (matcher.py:951-956); here mesh 1
    # (150 px^2 per triangle against 200): (min, max) = (0.5, 1) -> blocks of half side 25 in mesh 0, 50 in mesh 1
    c0, c1 = region_ref.distribute_matching_blocks(v0, t0, v1, t1, 100.0, res=0.5, shrink_factor=(1, 0.5))
    assert np.all(c0[:, 2:] - c0[:, :2] == 100.0) and np.all(c1[:, 2:] - c1[:, :2] == 50.0)        # mesh 0: 200 px^2 > mesh 1: 150 px^2


def test_min_boundary_distance_and_its_relaxation_by_hand():
    """the same overlap [100, 400] x [0, 260] (78 000 px^2).  min_boundary_distance 40: the eroded region [140, 360] x [40, 220]
    holds 39 600 px^2 >= one half -> kept; of the 9 centres only x in {150.. } >= 140: (150|250|350, 130) and (.., 30)? no: y = 30
    < 40 is out, y = 230 > 220 is out -> the middle row only.  min_boundary_distance 80: [180, 320] x [80, 180] = 14 000 px^2 <
    one half -> bound_coeff = 0.3 / (1 - 14000 / 78000) = 0.3656 -> distance 29.25: [129.25, 370.75] x [29.25, 230.75] = 48 660 >=
    one half -> centres with 129.25 <= x <= 370.75, 29.25 <= y <= 230.75: all nine but none lost in x; y = 30 and 230 stay"""
    v0, t0 = _rect_mesh(0, 0, 400, 300, 20.0)
    v1, t1 = fem_ref.grid_mesh(15, 32, 10.0, origin=(100.0, -50.0)); v1 = v1 * np.array([[3.0, 1.0]]) - np.array([[200.0, 0.0]])
    b0, _ = region_ref.distribute_matching_blocks(v0, t0, v1, t1, 100.0, res=0.5, min_boundary_distance=40)
    ctr = 0.5 * (b0[:, :2] + b0[:, 2:])
    # the lattice is anchored on the ERODED region: its bounds are [140, 360] x [40, 220], its representative point (250, 130):
    # rx_mn = 250 - ((250 - 140) // 100) * 100 = 150, ry_mn = 130 - ((130 - 40) // 100) * 100 = 130 -> x in {150, 250, 350}, y in {130}
    assert sorted(map(tuple, ctr)) == [(150.0, 130.0), (250.0, 130.0), (350.0, 130.0)]
    b0, _ = region_ref.distribute_matching_blocks(v0, t0, v1, t1, 100.0, res=0.5, min_boundary_distance=80)
    ctr = 0.5 * (b0[:, :2] + b0[:, 2:])
    # eroded by 29.25: bounds [129.25, 370.75] x [29.25, 230.75]: rx_mn = 250 - 100 = 150, ry_mn = 130 - 100 = 30 -> 3 x 3 again
    assert ctr.shape[0] == 9 and ctr[:, 1].min() == 30.0 and ctr[:, 1].max() == 230.0


def test_refinement_levels_by_hand():
    """a material 'wrinkle' (area_constraint 0.25) on the triangles of mesh 0 with centroid x < 200, one named 'refine_zone'
    (area_constraint 1, picked by its name) on those with x > 340.  refine_mode 2 on the overlap [100, 400] x [0, 260]:
    level 0.25 first -- lattice step 25 on [100, 200] x [0, 260], blocks of spacing * 0.25 * 0.25^(0.5 - 1) = 50 -> half side 25 --,
    then level 1.0: 'refine_zone' and reg_crx0 together (unary_union = the whole overlap) minus what level 0.25 covered:
    [200, 400] x [0, 260], step 100"""
    v0, t0 = _rect_mesh(0, 0, 400, 300, 20.0)
    v1, t1 = fem_ref.grid_mesh(15, 32, 10.0, origin=(100.0, -50.0)); v1 = v1 * np.array([[3.0, 1.0]]) - np.array([[200.0, 0.0]])
    c = v0[t0].mean(axis=1)
    ids = np.zeros(t0.shape[0], dtype=np.int32); ids[c[:, 0] < 200] = 4; ids[c[:, 0] > 340] = 9
    mats = ((ids, {'default': (0, 1.0), 'wrinkle': (4, 0.25), 'refine_zone': (9, 1.0)}), None)
    b0, b1 = region_ref.distribute_matching_blocks(v0, t0, v1, t1, 100.0, res=0.5, refine_mode=2, materials=mats)
    side = b0[:, 2] - b0[:, 0]
    fine, coarse = side == 50.0, side == 100.0
    assert fine.sum() + coarse.sum() == b0.shape[0] and np.all(np.flatnonzero(fine) < np.flatnonzero(coarse).min())       # finest level first
    cf = 0.5 * (b0[fine, :2] + b0[fine, 2:]); cc = 0.5 * (b0[coarse, :2] + b0[coarse, 2:])
    # fine level: region [100, 200] x [0, 260], representative point (150, 130): x in {100, 125, 150, 175, (200)}, y = 130 + 25 k in [0, 260]
    assert cf[:, 0].min() >= 100 and cf[:, 0].max() <= 200 and np.allclose((cf[:, 0] - 150.0) % 25.0, 0) and np.allclose((cf[:, 1] - 130.0) % 25.0, 0)
    assert 4 * 11 <= cf.shape[0] <= 5 * 11
    # coarse level: [200, 400] x [0, 260], representative point (300, 130): x in {200?, 300, 400?} ...: the lattice phase is 300 mod 100
    assert np.allclose((cc[:, 0] - 300.0) % 100.0, 0) and np.allclose((cc[:, 1] - 130.0) % 100.0, 0) and cc[:, 0].min() >= 200
    # refine_mode 1: the refinement regions only -- 'wrinkle' (0.25) and 'refine_zone' (factor 1: [340, 400] x [0, 260], step 100)
    r0, _ = region_ref.distribute_matching_blocks(v0, t0, v1, t1, 100.0, res=0.5, refine_mode=1, materials=mats)
    rs = r0[:, 2] - r0[:, 0]
    rc = 0.5 * (r0[rs == 100.0, :2] + r0[rs == 100.0, 2:])
    assert (rs == 50.0).sum() == fine.sum() and rc.shape[0] >= 2 and rc[:, 0].min() >= 340
    # refine_mode 0: no levels
    z0, _ = region_ref.distribute_matching_blocks(v0, t0, v1, t1, 100.0, res=0.5, refine_mode=0, materials=mats)
    assert np.all(z0[:, 2] - z0[:, 0] == 100.0) and z0.shape[0] == 9


def test_parts_get_their_own_lattice_and_anchors_can_be_handed_in():
    """two islands of mesh 0 over one large mesh 1: every connected part of the region gets a lattice through ITS representative
    point (matcher.py:1024-1035); with anchor_blocks the phase of a part's lattice is taken from a block that lies in it"""
    va, ta = _rect_mesh(0, 0, 200, 200, 20.0)
    vb, tb = _rect_mesh(310, 37, 530, 257, 20.0)
    v0 = np.concatenate((va, vb)); t0 = np.concatenate((ta, tb + va.shape[0]))
    v1, t1 = _rect_mesh(-40, -40, 600, 300, 40.0)
    b0, _ = region_ref.distribute_matching_blocks(v0, t0, v1, t1, 80.0, res=0.5)
    ctr = 0.5 * (b0[:, :2] + b0[:, 2:])
    left, right = ctr[ctr[:, 0] < 250], ctr[ctr[:, 0] > 250]
    assert np.allclose((left - np.array([100.0, 100.0])) % 80.0, 0) and np.allclose((right - np.array([420.0, 147.0])) % 80.0, 0)
    assert left.shape[0] == 9 and right.shape[0] == 9
    hint = np.array([[100.0 + 13 - 40, 100.0 - 7 - 40, 100.0 + 13 + 40, 100.0 - 7 + 40]])             # a block of the left island, phase (13, -7)
    a0, _ = region_ref.distribute_matching_blocks(v0, t0, v1, t1, 80.0, res=0.5, anchor_blocks=hint)
    ac = 0.5 * (a0[:, :2] + a0[:, 2:])
    assert np.allclose((ac[ac[:, 0] < 250] - np.array([113.0, 93.0])) % 80.0, 0) and np.allclose((ac[ac[:, 0] > 250] - np.array([420.0, 147.0])) % 80.0, 0)
    # no overlap at all
    e0, e1 = region_ref.distribute_matching_blocks(va, ta, vb, tb, 80.0)
    assert e0.shape == (0, 4) and e1.shape == (0, 4)


def test_krylov_limit_of_a_floating_pair():
    """two free meshes linked to each other: A is singular (common translations); the solve the section matcher's oracle uses
    returns the solution a Jacobi-preconditioned Krylov method reaches from zero -- checked against fem_ref.pcg itself"""
    rng = np.random.default_rng(5)
    v, t = fem_ref.grid_mesh(7, 6, 10.0)
    m0 = fem_ref.RefMesh(v, t, uid=0); m1 = fem_ref.RefMesh(v + rng.normal(0, 0.4, v.shape), t, uid=1)
    n = 40
    tid = rng.integers(0, t.shape[0], n); B = rng.dirichlet((1, 1, 1), n)
    lk = fem_ref.RefLink(m0, m1, tid, tid, B, B, weight=rng.uniform(0.5, 1, n).astype(np.float32))
    A, b, _ = fem_ref.linear_system([m0, m1], [lk], 0.5, -1.0)
    A = 0.5 * (A + A.T)
    x = region_ref._solve_jacobi_krylov_limit(A, b)
    assert np.linalg.norm(A @ x - b) <= 1e-9 * np.linalg.norm(b)
    xp, it, rel = fem_ref.pcg(A, b, rtol=1e-13, maxiter=5000, minv=1.0 / A.diagonal())
    assert rel < 1e-11
    np.testing.assert_allclose(x, xp, atol=1e-8 * np.abs(x).max())
    # and it is NOT the Euclidean minimum-norm solution (the diagonal of A is not constant)
    xm = np.linalg.lstsq(A.toarray(), b, rcond=1e-12)[0]
    assert np.abs(x - xm).max() > 1e-6 * np.abs(x).max()
