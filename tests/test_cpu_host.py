"""CPU-side checks of the product package: the C-ABI library loads and exports
every symbol include/feabas_hip.h declares (no compute calls), and the host
logic (bbox helpers, next_fast_len) matches the golden vectors."""
import ctypes
import os

import numpy as np
import pytest

from conftest import load_golden
import feabas_amd
from feabas_amd import _lib, common, matcher


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), n
    # and every prototype we bind is declared in the header
    assert set(_lib._PROTOS) <= set(names)
    # the product library exports the boundary and nothing else: the test hooks live in libfeabas_hip_test.so
    import subprocess
    exported = {ln.split()[-1] for ln in subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout.splitlines()
                if ln.split()[-1].startswith('fb_')}
    if not os.environ.get('FEABAS_HIP_LIB'):                     # (an A/B or sanitizer build may carry the hooks)
        assert exported == set(names), sorted(exported ^ set(names))
    tlib = _lib.load_test()
    hooks = [n for n in _lib.declared_symbols('feabas_hip_test.h') if n not in names]
    assert sorted(hooks) == sorted(_lib._TEST_PROTOS) and len(hooks) == 5
    for n in list(names) + hooks:
        assert hasattr(tlib, n), n


def test_bindings_agree_with_the_header():
    """every fb_* entry the python side calls has a ctypes prototype (an unprototyped call would pass pointers as C ints), and every
    prototype has as many arguments as the declaration in include/feabas_hip.h (test hooks: include/feabas_hip_test.h)"""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    protos = dict(_lib._PROTOS); protos.update(_lib._TEST_PROTOS)
    used, used_product = set(), set()
    for f in glob.glob(os.path.join(root, 'feabas_amd', '*.py')) + [os.path.join(root, 'bench.py'), os.path.join(root, '__graft_entry__.py')] \
            + glob.glob(os.path.join(root, 'tests', '*.py')) + glob.glob(os.path.join(root, 'tools', '*.py')):
        found = re.findall(r'\.(fb_[a-z0-9_]+)\b', open(f).read())
        used.update(found)
        if os.sep + 'feabas_amd' + os.sep in f or f.endswith(('bench.py', '__graft_entry__.py')):
            used_product.update(found)
    assert len(used) > 80 and used <= set(protos), sorted(used - set(protos))
    assert not (used_product & set(_lib._TEST_PROTOS)), 'the product calls a test hook'
    seen = 0
    for header in ('feabas_hip.h', 'feabas_hip_test.h'):
        hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(root, 'include', header)).read(), flags=re.S)
        for m in re.finditer(r'\b[A-Za-z_][A-Za-z0-9_]*\s*\*?\s+\*?(fb_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;', hdr, flags=re.S):
            name, args = m.group(1), m.group(2).strip()
            if name in protos:
                seen += 1
                assert len(protos[name][1]) == (0 if args in ('', 'void') else len(args.split(','))), name
    assert seen == len(protos)


def test_no_cpu_fallback_without_gpu():
    lib = _lib.load()
    if lib.fb_create(0):
        pytest.skip('a GPU is visible')
    with pytest.raises(RuntimeError):
        matcher.xcorr_fft(np.zeros((1, 8, 8), np.float32), np.zeros((1, 8, 8), np.float32))


def test_product_does_not_import_oracle():
    import os, re
    root = os.path.dirname(os.path.abspath(feabas_amd.__file__))
    for fn in os.listdir(root):
        if fn.endswith('.py'):
            src = open(os.path.join(root, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), fn


def test_next_fast_len():
    g = load_golden('g11_bbox.npz')
    got = np.array([matcher.next_fast_len(n) for n in range(1, 4200)])
    np.testing.assert_array_equal(got, g['nfl'])


def test_bbox_helpers_golden():
    g = load_golden('g11_bbox.npz')
    k = 0
    while f'div{k}_in' in g:
        p = g[f'div{k}_in']
        mnb = tuple(int(v) for v in p[6:])
        kw = dict(min_num_blocks=mnb[0] if len(mnb) == 1 else mnb, shrink_factor=p[5])
        if p[4] > 0:
            kw['block_size'] = p[4]
        res = np.stack(common.divide_bbox(tuple(p[:4]), **kw), axis=-1)
        np.testing.assert_array_equal(res, g[f'div{k}_out'])
        k += 1
    np.testing.assert_array_equal(common.z_order(g['z_in']), g['z_out'])
    np.testing.assert_allclose(common.bbox_centers(g['bb_in']), g['bb_centers'])
    np.testing.assert_allclose(common.bbox_sizes(g['bb_in']), g['bb_sizes'])

    class M:
        def __init__(self, bb): self.bb = bb
        def bbox(self, gear=None): return self.bb
    for k in range(2):
        p = g[f'dist{k}_in']
        r0, r1 = matcher.distributor_cartesian_bbox(M(p[:4]), M(p[4:8]), p[8], min_num_blocks=int(p[9]), zorder=True)
        np.testing.assert_array_equal(r0, g[f'dist{k}_bb0'])
        np.testing.assert_array_equal(r1, g[f'dist{k}_bb1'])


def test_mesh_gears_and_field_semantics():
    from feabas_amd.mesh import Mesh
    from feabas_amd import constant as const
    from oracle import fem_ref
    v, t = fem_ref.grid_mesh(5, 4, 10.0)
    m = Mesh(v, t, uid=3)
    r = fem_ref.RefMesh(v, t, uid=3)
    m.apply_translation((1.5, -2.0), const.MESH_GEAR_FIXED)
    r.apply_translation((1.5, -2.0), fem_ref.GEAR_FIXED)
    d = np.random.default_rng(0).standard_normal(v.shape)
    m.set_field(d, gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_MOVING))
    r.set_field(d, gear=(fem_ref.GEAR_MOVING, fem_ref.GEAR_MOVING))
    for gear in (-1, 0, 1, 2):
        np.testing.assert_allclose(m.vertices(gear), r.vertices(gear))
        np.testing.assert_allclose(m.offset(gear), r.offset(gear))
    m.anneal(gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_FIXED), mode=const.ANNEAL_COPY_EXACT)
    r.anneal_copy((fem_ref.GEAR_MOVING, fem_ref.GEAR_FIXED))
    np.testing.assert_allclose(m.vertices_w_offset(0), r.vertices_w_offset(0))
    tid = np.array([0, 5, 11])
    xy = r.bary2cart(tid, np.array([[0.2, 0.3, 0.5]] * 3), 1)
    tid2, B = m.cart2bary(xy, 1, tid=None)
    np.testing.assert_array_equal(tid2, tid)
    np.testing.assert_allclose(B, [[0.2, 0.3, 0.5]] * 3, atol=1e-12)


def test_translation_optimisers_vs_reference_golden():
    """SLM.optimize_translation_lsqr / optimize_translation_w_filtering (optimizer.py:974-1125) are host code in the reference and
    here (scipy lsqr on #tiles unknowns): golden G15 from the reference, no GPU involved"""
    from conftest import load_golden
    import feabas_amd
    from feabas_amd import constant as const
    from feabas_amd.mesh import Mesh
    from feabas_amd.optimizer import Link, SLM
    g = load_golden('g15_translation.npz')

    def system():
        ms = [Mesh(g['v'], g['t'], uid=k) for k in range(6)]
        ms[0].lock()
        links = []
        for k in range(7):
            a, b = g[f'l{k}_ab']
            links.append(Link(ms[a], ms[b], g[f'l{k}_tid0'], g[f'l{k}_tid1'], g[f'l{k}_B0'], g[f'l{k}_B1'], weight=g[f'l{k}_w']))
        return ms, links, SLM(ms, links=links)
    ms, links, slm = system()
    cost, residue = slm.optimize_translation_lsqr(tol=1e-12)
    np.testing.assert_allclose(cost, g['lsqr_cost'], rtol=1e-9)
    np.testing.assert_allclose(residue, g['lsqr_residue'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.stack([m.offset(const.MESH_GEAR_FIXED).ravel() for m in ms]), g['lsqr_offsets'], atol=1e-8)
    ms, links, slm = system()
    nd, cost2 = slm.optimize_translation_w_filtering(tol=1e-12, residue_threshold=1.0)
    assert nd == int(g['filt_disabled'])
    np.testing.assert_array_equal(np.array([lk._disabled for lk in links]), g['filt_link_disabled'])
    np.testing.assert_allclose(cost2[0], g['filt_cost'][0], rtol=1e-9)
    assert cost2[1] < 1e-9
    np.testing.assert_allclose(np.stack([m.offset(const.MESH_GEAR_FIXED).ravel() for m in ms]), g['filt_offsets'], atol=1e-8)


@pytest.mark.parametrize('name,mode', [('grigid', 0), ('gaffine', 1), ('crigid', 2), ('caffine', 3)])
def test_g16_anneal_modes(name, mode):
    """Mesh.anneal rigid / affine, whole mesh and per connected component (mesh.py:2421-2451), and the region
    relax_mesh_most_deformed frees (optimizer.py:2157-2188) -- host logic of the product against the reference"""
    from conftest import load_golden
    from feabas_amd import constant as const
    from feabas_amd.mesh import Mesh
    g = load_golden('g16_relax.npz')
    m = Mesh(g['an_v'], g['an_t'], moving_vertices=g['an_vmov'].copy(), moving_offset=np.array([[1.0, 2.0]]), uid=4)
    m.anneal(gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_FIXED), mode=mode)
    np.testing.assert_allclose(m.vertices(const.MESH_GEAR_FIXED), g[f'an_{name}_vfix'], atol=1e-9)
    np.testing.assert_allclose(m.offset(const.MESH_GEAR_FIXED), g[f'an_{name}_foff'], atol=1e-9)


def test_g16_masked_field_and_measures():
    from conftest import load_golden
    from feabas_amd import constant as const
    from feabas_amd.mesh import Mesh
    g = load_golden('g16_relax.npz')
    gear = (const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)
    m = Mesh(g['v'], g['t'], stiffness_multiplier=g['mult'], moving_vertices=g['vmov'].copy(), moving_offset=g['moff'].copy(), uid=3)
    np.testing.assert_allclose(m.triangle_area_deform(gear), g['area_deform'], rtol=1e-12)
    np.testing.assert_allclose(m.triangle_edge_deform(gear), g['edge_deform'], rtol=1e-12)
    np.testing.assert_allclose(Mesh.svds_to_deform(g['area_deform'].reshape(-1, 1)), g['svd_deform_area'], rtol=1e-12)
    np.testing.assert_allclose(m.effective_stiffness_multiplier(), g['eff_mult'], rtol=1e-7)
    vm = np.zeros(g['v'].shape[0], dtype=bool)
    vm[g['free_vtx']] = True
    d = np.arange(2 * vm.sum(), dtype=np.float64).reshape(-1, 2)
    m.apply_field(d, gear[1], vtx_mask=vm)                  # mesh.py:2393-2396: only the masked vertices move, the offset stays
    np.testing.assert_array_equal(m.vertices(gear[1])[vm], g['vmov'][vm] + d)
    np.testing.assert_array_equal(m.vertices(gear[1])[~vm], g['vmov'][~vm])
    np.testing.assert_array_equal(m.offset(gear[1]), g['moff'])


def _kinked_mesh():
    from oracle import pipeline_ref, fem_ref
    W, H = 120, 1536
    v, tri, xs, ys = pipeline_ref.cartesian_mesh(W, H, 40.0)
    rng = np.random.default_rng(0)
    U = np.stack((2.5 * np.sin(v[:, 1] / 300) + 0.8 * np.cos(v[:, 0] / 40), 1.5 * np.cos(v[:, 1] / 200)), -1)
    k = rng.integers(0, v.shape[0], 6)
    U[k] += rng.normal(0, 0.6, (6, 2))
    m1 = fem_ref.RefMesh(v, tri, uid=1)
    m1.set_field(U, gear=(fem_ref.GEAR_FIXED, fem_ref.GEAR_MOVING))
    return W, H, v, tri, xs, ys, m1, rng


def test_deformed_block_tiers_match_the_renderer_restatement():
    """feabas_amd/deformed.py (batched numpy over the blocks of a pair) against the statement-by-statement oracle of
    MeshRenderer.from_mesh / crop_field (renderer.py:90-109, 397-416, 499-511): tier of every block, the triangles that
    touch it, and the affine maps"""
    from feabas_amd import deformed
    from oracle import pipeline_ref, fem_ref, ncc_ref
    W, H, v, tri, xs, ys, m1, rng = _kinked_mesh()
    vm = m1.vertices_w_offset(fem_ref.GEAR_MOVING)
    bb0, bb1 = ncc_ref.distributor_cartesian_bbox((2.5, -2.5, W + 2.5, H - 2.5), (vm[:, 0].min(), vm[:, 1].min(), vm[:, 0].max(), vm[:, 1].max()),
                                                  40.0, min_num_blocks=2)
    img = rng.standard_normal((H, W)).astype(np.float32)
    seen = set()
    for tol in (0.1, 0.3, 1.0, 5.0):
        _, tiers = pipeline_ref.render_blocks_mesh1(m1, img, bb1, tol, return_tiers=True)
        tier, A, hits = deformed.block_affines(vm, v, tri, bb1, tol)
        np.testing.assert_array_equal(tier, tiers)
        seen |= set(tier.tolist())
        off = m1.offset(fem_ref.GEAR_MOVING).ravel()
        v0 = m1.vertices(fem_ref.GEAR_MOVING)
        for b in np.flatnonzero(tier == 2)[:25]:
            hit = pipeline_ref.tri_box_intersects(v0[tri], bb1[b] - np.tile(off, 2) - 0.5)
            np.testing.assert_array_equal(hit, hits[b])
            idx = np.unique(tri[hit])
            _, Ab = fem_ref.fit_affine(v[idx], v0[idx], return_rigid=True, svd_clip=None)
            pt = np.array([[bb1[b, 0], bb1[b, 1]], [bb1[b, 2], bb1[b, 3]]], dtype=float)
            np.testing.assert_allclose(pt @ A[b, :2, :2] + A[b, 2, :2], (pt - off) @ Ab[:2, :2] + Ab[2, :2], atol=1e-9)
    assert seen == {1, 2, 3}


def test_deformed_exact_field_and_point_location_match_matplotlib():
    """deformed.exact_field / deformed.locate against matplotlib.tri (what the reference uses: LinearTriInterpolator,
    renderer.py:116-117; trifinder, mesh.py:2113)"""
    import matplotlib.tri as mt
    from feabas_amd import deformed
    from oracle import fem_ref
    W, H, v, tri, xs, ys, m1, rng = _kinked_mesh()
    vm = m1.vertices_w_offset(fem_ref.GEAR_MOVING)
    T = mt.Triangulation(vm[:, 0], vm[:, 1], tri)
    ix, iy = mt.LinearTriInterpolator(T, v[:, 0]), mt.LinearTriInterpolator(T, v[:, 1])
    for x0, y0, h, w in ((40, 200, 38, 40), (-6, -5, 50, 45), (90, 1500, 44, 40)):       # inside, corner, far corner
        hit = deformed.tri_box_hits(vm[tri], np.array([[x0, y0, x0 + w, y0 + h]]) - 0.5)[0]
        mx, my, mk = deformed.exact_field(vm, v, tri, np.flatnonzero(hit), x0, y0, h, w)
        xx, yy = np.meshgrid(np.arange(x0, x0 + w, dtype=float), np.arange(y0, y0 + h, dtype=float))
        ex, ey = ix(xx, yy), iy(xx, yy)
        np.testing.assert_array_equal(mk, ~np.ma.getmaskarray(ex))
        np.testing.assert_allclose(mx[mk], ex.filled(0)[mk], atol=1e-9); np.testing.assert_allclose(my[mk], ey.filled(0)[mk], atol=1e-9)
    pts = np.stack((rng.uniform(vm[:, 0].min() - 2, vm[:, 0].max() + 2, 4000), rng.uniform(vm[:, 1].min() - 2, vm[:, 1].max() + 2, 4000)), -1)
    tid, B = deformed.locate(vm, tri, xs, ys, pts)
    np.testing.assert_array_equal(tid, T.get_trifinder()(pts[:, 0], pts[:, 1]))
    ok = tid >= 0
    assert 0 < (~ok).sum() < ok.sum()
    np.testing.assert_allclose(np.sum(vm[tri[tid[ok]]] * B[ok][:, :, None], axis=1), pts[ok], atol=1e-9)


def test_oracle_deformed_branch_recovers_a_smooth_warp():
    """the oracle's deformed branch on a pair warped by a few pixels: matches in the INITIAL gear follow the injected
    displacement (size-independent property: xy1 - xy0 = -(shift + warp) at the match)"""
    from scipy.ndimage import gaussian_filter, map_coordinates
    from oracle import pipeline_ref
    H, W = 1536, 120
    rng = np.random.default_rng(1)
    pad = 64
    tex = gaussian_filter(rng.standard_normal((H + 2 * pad, W + 2 * pad)), 1.6) + 1.8 * gaussian_filter(rng.standard_normal((H + 2 * pad, W + 2 * pad)), 5.0)
    tex = 128 + 45 * tex / tex.std()
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    wx = 3.0 * np.sin(2 * np.pi * yy / H * 1.5 + 0.3) * np.cos(np.pi * xx / H)
    wy = 3.0 * np.cos(2 * np.pi * xx / H * 1.2 + 0.7) * (0.5 + 0.5 * np.sin(2 * np.pi * yy / H))
    s0 = np.clip(np.round(map_coordinates(tex, [yy + pad, xx + pad], order=1)), 0, 255).astype(np.uint8)
    s1 = np.clip(np.round(map_coordinates(tex, [yy + pad - 3 + wy, xx + pad + 4 + wx], order=3)), 0, 255).astype(np.uint8)
    r = pipeline_ref.match_pair(s0, s1, residue_len=2.0)
    assert r['deformed'] and r['xy0'].shape[0] > 100
    # strip1(q) = tex(q + s + w(q)) and strip0(p) = tex(p): a match has p = q + s + w(q)
    q = r['xy1']
    wq = np.stack((map_coordinates(wx, [q[:, 1], q[:, 0]], order=1), map_coordinates(wy, [q[:, 1], q[:, 0]], order=1)), -1)
    err = (r['xy0'] - q) - (np.array([4.0, -3.0]) + wq)
    assert np.abs(err).max() < 1.0 and np.percentile(np.abs(err), 90) < 0.4 and np.abs(np.median(err, axis=0)).max() < 0.15
    assert np.ptp(r['mesh1_field'][:, 0]) > 2.0


def test_cxx_deformed_geometry_matches_numpy():
    """fb_deformed_block_affines / fb_deformed_locate (host C++ behind the C ABI, no device work, ctx = NULL) against the
    numpy statements of feabas_amd/deformed.py, which the tests above pin to the renderer restatement"""
    from feabas_amd import _lib, deformed
    from oracle import pipeline_ref, ncc_ref
    lib = _lib.load()
    W, H = 120, 1536
    v, tri, xs, ys = pipeline_ref.cartesian_mesh(W, H, 40.0)
    rng = np.random.default_rng(0)
    Q = 3
    vm = np.empty((Q,) + v.shape)
    for q in range(Q):
        U = np.stack((2.5 * np.sin(v[:, 1] / 300) + 0.8 * np.cos(v[:, 0] / 40), 1.5 * np.cos(v[:, 1] / 200)), -1) * (0.02 if q == 2 else 1)
        k = rng.integers(0, v.shape[0], 6)
        U[k] += rng.normal(0, 0.6 if q < 2 else 0.0, (6, 2))
        vm[q] = v + U + rng.normal(0, 1, 2)
    bbs = [ncc_ref.distributor_cartesian_bbox((2.5, -2.5, W + 2.5, H - 2.5), (vm[q, :, 0].min(), vm[q, :, 1].min(), vm[q, :, 0].max(), vm[q, :, 1].max()),
                                              40.0, min_num_blocks=2)[1] for q in range(Q)]
    nblk = min(b.shape[0] for b in bbs)
    bb = np.ascontiguousarray(np.stack([b[:nblk] for b in bbs]), dtype=np.int32)
    seen = set()
    for tol in (0.1, 0.3, 1.0, 5.0):
        tier = np.empty((Q, nblk), np.int32); A6 = np.empty((Q, nblk, 6)); lo = np.empty((Q, 2))
        assert lib.fb_deformed_block_affines(None, Q, xs.size, ys.size, _lib.ptr(xs), _lib.ptr(ys), 0, _lib.ptr(vm), nblk, _lib.ptr(bb), tol, None,
                                             _lib.ptr(tier), _lib.ptr(A6), _lib.ptr(lo)) == 0
        for q in range(Q):
            t2, A, _ = deformed.block_affines(vm[q], v, tri, bb[q], tol)
            np.testing.assert_array_equal(t2, tier[q])
            seen |= set(t2.tolist())
            a = t2 < 3
            if a.any():
                ref = np.stack((A[:, 0, 0], A[:, 1, 0], A[:, 2, 0], A[:, 0, 1], A[:, 1, 1], A[:, 2, 1]), -1)
                np.testing.assert_allclose(A6[q][a], ref[a], atol=1e-9)
                cx = np.stack((bb[q, a, 0], bb[q, a, 2] - 1), -1).astype(float); cy = np.stack((bb[q, a, 1], bb[q, a, 3] - 1), -1).astype(float)
                mx = cx[:, :, None] * ref[a, 0, None, None] + cy[:, None, :] * ref[a, 1, None, None] + ref[a, 2, None, None]
                assert abs(mx.min() - lo[q, 0]) < 1e-9
    assert seen == {1, 2, 3}
    K = 5000
    po = rng.integers(0, Q, K).astype(np.int32)
    pts = np.stack((rng.uniform(-5, W + 5, K), rng.uniform(-5, H + 5, K)), -1)
    tid = np.empty(K, np.int32); B = np.empty((K, 3))
    assert lib.fb_deformed_locate(None, Q, xs.size, ys.size, _lib.ptr(xs), _lib.ptr(ys), 0, _lib.ptr(vm), K, _lib.ptr(po), _lib.ptr(pts),
                                  _lib.ptr(tid), _lib.ptr(B)) == 0
    for q in range(Q):
        s_ = po == q
        t2, B2 = deformed.locate(vm[q], tri, xs, ys, pts[s_])
        np.testing.assert_array_equal(t2, tid[s_])
        ok = t2 >= 0
        np.testing.assert_allclose(B[s_][ok], B2[ok], atol=1e-12)
    assert 0 < (tid < 0).sum() < K // 2
    # exact field of a few blocks (inside, corners) of pair 1
    h, w = 38, 40
    org = np.array([[40, 200], [-6, -5], [90, 1500], [60, 700]], dtype=np.int32)
    po = np.ones(4, np.int32)
    mx = np.empty((4, h, w)); my = np.empty((4, h, w)); mk = np.empty((4, h, w), np.uint8)
    assert lib.fb_deformed_exact_field(None, Q, xs.size, ys.size, _lib.ptr(xs), _lib.ptr(ys), 0, _lib.ptr(vm), 4, _lib.ptr(po), _lib.ptr(org), h, w,
                                       _lib.ptr(mx), _lib.ptr(my), _lib.ptr(mk)) == 0
    for e in range(4):
        hit = deformed.tri_box_hits(vm[1][tri], np.array([[org[e, 0], org[e, 1], org[e, 0] + w, org[e, 1] + h]]) - 0.5)[0]
        a, b, c = deformed.exact_field(vm[1], v, tri, np.flatnonzero(hit), int(org[e, 0]), int(org[e, 1]), h, w)
        np.testing.assert_array_equal(c, mk[e].astype(bool))
        np.testing.assert_array_equal(a[c], mx[e][c]); np.testing.assert_array_equal(b[c], my[e][c])
    assert 0 < mk.mean() < 1


def test_find_overlaps_readme_grid():
    """stitcher.find_overlaps (stitcher.py:418-437) on the README's 3 x 2 grid: 4 left-right + 3 up-down + 4 diagonal
    corner overlaps, each as (later tile, earlier tile), z-ordered by overlap centre"""
    from feabas_amd import stitcher
    nom = np.array([[x, y] for y in (0, 2700) for x in (0, 3600, 7200)])
    bboxes = np.concatenate((nom, nom + np.array([4000, 3000])), axis=1)
    ov = stitcher.find_overlaps(bboxes, tile_size=(3000, 4000))
    assert ov.shape == (11, 2) and np.all(ov[:, 0] > ov[:, 1])
    assert {tuple(p) for p in ov.tolist()} == {(1, 0), (2, 1), (3, 0), (4, 0), (4, 1), (4, 3), (3, 1), (5, 1), (5, 2), (5, 4), (4, 2)}
    _, wd = stitcher.bbox_intersections(bboxes[ov[:, 0]], bboxes[ov[:, 1]])
    assert sorted(wd.tolist()) == [300] * 7 + [400] * 4
    assert stitcher.find_overlaps(bboxes[:1]).shape == (0, 2)
    m, s, p, err = stitcher.match_list_of_overlaps(np.empty((0, 2), int), [], bboxes)
    assert (m, s, p, err) == ({}, {}, {}, False)


def test_host_pack2d():
    """fb_host_pack2d: row-pitched views gathered into the corners of a staging stack"""
    from feabas_amd import _lib
    import ctypes as C
    rng = np.random.default_rng(2)
    big = rng.integers(0, 256, (5, 90, 120), dtype=np.uint8)
    views = [big[0, :80, :100], big[1, 3:90, 7:64], big[2], big[3, ::1, :1], big[4, :1, :]]
    H, W = 90, 120
    dst = np.full((len(views), H, W), 255, dtype=np.uint8)
    srcs = (C.c_void_p * len(views))(*[v.ctypes.data for v in views])
    hs = np.array([v.shape[0] for v in views], dtype=np.int32); ws = np.array([v.shape[1] for v in views], dtype=np.int32)
    pitches = np.array([v.strides[0] for v in views], dtype=np.int64)
    for threads in (1, 3):
        dst[...] = 255
        assert _lib.load().fb_host_pack2d(None, _lib.ptr(dst), len(views), H, W, srcs, _lib.ptr(hs), _lib.ptr(ws), _lib.ptr(pitches), threads) == 0
        for k, v in enumerate(views):
            np.testing.assert_array_equal(dst[k, :v.shape[0], :v.shape[1]], v)
            assert (dst[k, v.shape[0]:] == 255).all() and (dst[k, :, v.shape[1]:] == 255).all()
    assert _lib.load().fb_host_pack2d(None, _lib.ptr(dst), 1, 10, 10, srcs, _lib.ptr(hs), _lib.ptr(ws), _lib.ptr(pitches), 1) != 0     # does not fit


def test_match_list_of_overlaps_bookkeeping(monkeypatch):
    """stitcher.match_list_of_overlaps (stitcher.py:552-603) without a GPU: the strips handed to the matcher are the
    overlap enlarged by the margin and clipped to each tile, matches come back shifted into tile coordinates, thin
    overlaps are skipped, maskout_val becomes a dilated mask, index_mapper renames the keys"""
    from feabas_amd import stitcher
    TH, TW = 300, 400
    nom = np.array([[0, 0], [360, 4], [0, 270], [372, 268]])
    bboxes = np.concatenate((nom, nom + np.array([TW, TH])), axis=1)
    rng = np.random.default_rng(0)
    tiles = [rng.integers(1, 255, (TH, TW), dtype=np.uint8) for _ in range(4)]
    tiles[1][20:30, 5:15] = 255                                   # maskout_val pixels inside an overlap
    seen = []

    def fake_batch(pairs, batch=32, threads=2, **cfg):
        seen.append((pairs, cfg))
        out = []
        for img0, img1, mk0, mk1 in pairs:
            assert img0.shape == img1.shape
            xy = np.array([[1.0, 2.0], [3.0, 4.0]])
            out.append((xy, xy + 0.5, np.ones(2, np.float32), 0.01, None) if img0.shape[0] > 60 or img0.shape[1] > 60 else (None, None, 0.3, None, None))
        return out
    monkeypatch.setattr(stitcher, 'stitching_matcher_batch', fake_batch)
    overlaps = stitcher.find_overlaps(bboxes, tile_size=(TH, TW))
    matches, strains, phtm, err = stitcher.match_list_of_overlaps(overlaps, tiles, bboxes, min_overlap_width=25, margin=20, maskout_val=255,
                                                                  index_mapper={0: 'a', 1: 'b', 2: 'c', 3: 'd'}, matcher_config=dict(sigma=2.5))
    assert not err and not phtm and seen[0][1] == dict(sigma=2.5)
    pairs = seen[0][0]
    # the diagonal overlap of tiles 0 and 3 is 28 x 32 px: wider than min_overlap_width, so it is matched too
    assert len(pairs) == len(overlaps) == 6 or len(pairs) == len([1 for _ in overlaps])
    for (i, j), bbox_pair in zip(overlaps, pairs):
        ov, wd = stitcher.bbox_intersections(bboxes[i], bboxes[j])
        big = ov + np.array([-20, -20, 20, 20])
        b0 = stitcher.bbox_intersections(big, bboxes[i])[0]
        exp0 = tiles[i][b0[1] - bboxes[i][1]:b0[3] - bboxes[i][1], b0[0] - bboxes[i][0]:b0[2] - bboxes[i][0]]
        np.testing.assert_array_equal(bbox_pair[0], exp0)
        if i == 1 or j == 1:
            mk = bbox_pair[2] if i == 1 else bbox_pair[3]
            src = bbox_pair[0] if i == 1 else bbox_pair[1]
            if (src == 255).any():
                assert mk is not None and not mk[src == 255].any() and mk.mean() < 1 and (~mk).sum() > (src == 255).sum()    # dilated
    for (ki, kj), (xy0, xy1, wt) in matches.items():
        assert ki in 'abcd' and kj in 'abcd'
        i, j = 'abcd'.index(ki), 'abcd'.index(kj)
        ov, _ = stitcher.bbox_intersections(bboxes[i], bboxes[j])
        big = ov + np.array([-20, -20, 20, 20])
        off0 = stitcher.bbox_intersections(big, bboxes[i])[0][:2] - bboxes[i][:2]
        off1 = stitcher.bbox_intersections(big, bboxes[j])[0][:2] - bboxes[j][:2]
        np.testing.assert_allclose(xy0, np.array([[1.0, 2.0], [3.0, 4.0]]) + off0)
        np.testing.assert_allclose(xy1, np.array([[1.5, 2.5], [3.5, 4.5]]) + off1)
        assert strains[(ki, kj)] == 0.01
    assert 1 <= len(matches) <= len(pairs)
    # a margin <= 2 is a ratio of the overlap width (stitcher.py:556-559)
    seen.clear()
    stitcher.match_list_of_overlaps(overlaps[:1], tiles, bboxes, min_overlap_width=25, margin=0.5, matcher_config={})
    i, j = overlaps[0]
    ov, wd = stitcher.bbox_intersections(bboxes[i], bboxes[j])
    assert seen[0][0][0][0].shape[0] <= ov[3] - ov[1] + 2 * int(0.5 * wd) and min(seen[0][0][0][0].shape) >= wd


def test_renderer_pairwise_sat_matches_the_dense_form():
    """renderer._sat_hits (candidate pairs) = deformed.tri_box_hits (all pairs): the `intersects` query of renderer.py:405"""
    from feabas_amd import deformed, renderer
    rng = np.random.default_rng(31)
    tp = rng.uniform(0, 100, (40, 3, 2))
    boxes = np.sort(rng.uniform(-10, 110, (25, 2, 2)), axis=1).reshape(25, 4)[:, [0, 1, 2, 3]]
    boxes = np.stack((boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]), axis=-1)
    dense = deformed.tri_box_hits(tp, boxes)
    bi, ti = np.nonzero(np.ones_like(dense))
    got = renderer._sat_hits(tp[ti], boxes[bi]).reshape(dense.shape)
    np.testing.assert_array_equal(got, dense)
    assert 0.05 < dense.mean() < 0.95


def test_oracle_u8_remap_rule():
    """the restated cv2 uint8 bilinear rule (oracle/pipeline_ref.remap_origin): integer maps return the pixels, a half-pixel
    phase is the rounded mean, every value stays within 1/2 grey level of the float rule; zero outside the image"""
    from oracle import pipeline_ref
    rng = np.random.default_rng(32)
    img = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    yy, xx = np.meshgrid(np.arange(30.0), np.arange(35.0), indexing='ij')
    same = pipeline_ref.remap_origin(img, xx + 5, yy + 3, (2, 1))
    np.testing.assert_array_equal(same, img[3:33, 5:40].astype(np.float32))
    half = pipeline_ref.remap_origin(img, xx + 5.5, yy + 3, (2, 1))
    a, b = img[3:33, 5:40].astype(np.int64), img[3:33, 6:41].astype(np.int64)
    np.testing.assert_array_equal(half, ((a + b + 1) >> 1).astype(np.float32))
    mx, my = xx + 4.3 + 0.01 * yy, yy + 2.7 - 0.02 * xx
    u8 = pipeline_ref.remap_origin(img, mx, my, (0, 0))
    f32 = pipeline_ref.remap_origin(img.astype(np.float32), mx, my, (0, 0))
    assert np.abs(u8 - f32).max() <= 0.5 + 1e-4
    out = pipeline_ref.remap_origin(img, xx - 100, yy, (-104, -4))
    assert not out.any()
    shifted = pipeline_ref.remap_origin(img, xx + 5, yy + 3, (2, 1), img_origin=(-7, 4))      # image pixel (0, 0) at (-7, 4)
    np.testing.assert_array_equal(shifted[1:, :], img[0:29, 12:47].astype(np.float32))
    assert not shifted[0].any()


def test_link_terms_match_the_numpy_statement():
    """fb_link_terms (host C++): vertex ids, [B0 | -B1] and Link.dxy (optimizer.py:248-255) of the matches of one link,
    against the numpy statement Mesh.bary2cart makes -- bit for bit, negative triangle ids wrapping like numpy's"""
    from feabas_amd import _lib
    from oracle import fem_ref
    rng = np.random.default_rng(12)
    v, t = fem_ref.grid_mesh(31, 23, 7.5)
    t = np.ascontiguousarray(t, dtype=np.int32)
    K = 30000
    v0 = v + rng.normal(0, 0.4, v.shape); v1 = v + rng.normal(0, 0.4, v.shape)
    tid0 = rng.integers(0, t.shape[0], K).astype(np.int64); tid1 = rng.integers(0, t.shape[0], K).astype(np.int64)
    tid0[5] = -1; tid1[9] = -2
    B0 = rng.dirichlet((1, 1, 1), K); B1 = rng.dirichlet((1, 1, 1), K)
    off = np.array([0.37, -1.25])
    for voff0, voff1 in ((-1, 0), (400, -1), (0, v.shape[0])):
        nodes6 = np.empty((K, 6), dtype=np.int32); bary6 = np.empty((K, 6)); rxy = np.empty((K, 2))
        rc = _lib.load().fb_link_terms(None, K, _lib.ptr(t), t.shape[0], _lib.ptr(v0), _lib.ptr(tid0), _lib.ptr(B0), voff0, _lib.ptr(t), t.shape[0], _lib.ptr(v1),
                                       _lib.ptr(tid1), _lib.ptr(B1), voff1, float(off[0]), float(off[1]), _lib.ptr(nodes6), _lib.ptr(bary6), _lib.ptr(rxy))
        assert rc == 0
        e0 = np.full((K, 3), -1) if voff0 < 0 else t[tid0] + voff0
        e1 = np.full((K, 3), -1) if voff1 < 0 else t[tid1] + voff1
        np.testing.assert_array_equal(nodes6, np.concatenate((e0, e1), axis=1))
        np.testing.assert_array_equal(bary6, np.concatenate((B0, -B1), axis=1))

        def b2c(vv, tid, B):
            idx = t[tid]
            out = vv[idx[:, 0]] * B[:, 0:1]; out += vv[idx[:, 1]] * B[:, 1:2]; out += vv[idx[:, 2]] * B[:, 2:3]
            return out
        np.testing.assert_array_equal(rxy, (b2c(v1, tid1, B1) - b2c(v0, tid0, B0)) + off)
    bad = tid0.copy(); bad[3] = t.shape[0]
    assert _lib.load().fb_link_terms(None, K, _lib.ptr(t), t.shape[0], None, _lib.ptr(bad), None, 0, _lib.ptr(t), t.shape[0], None, _lib.ptr(tid1), None, -1,
                                     0.0, 0.0, _lib.ptr(nodes6), None, None) != 0


def test_round_stepper_of_the_block_matcher():
    """fb_schedule_* (the C-ABI stepper of the coarse-to-fine walk, matcher.py:567-716) on hand-worked walks: the jump to the
    smallest spacing that still holds 4 x the displacement (one place at a time with max_spacing_skip = 0, padding off when a
    place was reached by a jump of exactly one), the one-place step with padding when the displacement does not allow a jump,
    dwelling, the enlarged extra round, a fixed `pad`; and against the one-line rule of the strip oracle (pipeline_ref.py:441-442)"""
    from feabas_amd.matcher import _RoundPlan

    def walk(plan, dis):
        out = []
        for d in dis:
            r = plan.due()
            if r is None:
                break
            redo = plan.advance(d)
            out.append(r + (redo,))
        out.append(plan.due())
        plan.close()
        return out
    # the 4k strip: spacings [1024, 75]; a displacement of 12 px lets the walk go to 75 by a one-place jump: no padding there
    assert walk(_RoundPlan([75, 1024]), [12.0, 0.3]) == [(1024.0, False, True, False), (75.0, True, False, False), None]
    # ... of 30 px (4 x 30 > 75) it may not jump: it still moves on (allow_dwell = 0), but with padded blocks
    assert walk(_RoundPlan([1024, 75]), [30.0, 0.3]) == [(1024.0, False, True, False), (75.0, True, True, False), None]
    # three spacings, jump over one: max_spacing_skip = 0 clips it to one place; = 1 takes both and pads (the skipped range was never searched)
    assert walk(_RoundPlan([1000, 300, 75]), [2.0, 2.0, 2.0]) == [(1000.0, False, True, False), (300.0, False, False, False), (75.0, True, False, False), None]
    assert walk(_RoundPlan([1000, 300, 75], max_spacing_skip=1), [2.0, 2.0]) == [(1000.0, False, True, False), (75.0, True, True, False), None]
    # dwelling twice on a spacing whose displacement stays large
    w = walk(_RoundPlan([400, 100], allow_dwell=2), [200.0] * 6)
    assert [r[0] for r in w[:-1]] == [400.0, 400.0, 400.0, 100.0, 100.0, 100.0] and w[-1] is None
    # the displacement outruns the largest spacing: one extra round at ceil(4 d), repeated before anything is linked, then the list
    w = walk(_RoundPlan([400, 100], allow_enlarge=True), [300.2, 300.2, 1.0, 1.0])
    assert w == [(400.0, False, True, True), (1201.0, False, True, False), (400.0, False, True, False), (100.0, True, False, False), None]
    assert walk(_RoundPlan([400, 100], allow_enlarge=False), [300.0, 1.0]) == [(400.0, False, True, False), (100.0, True, True, False), None]
    # a fixed `pad` is never overruled
    assert [r[2] for r in walk(_RoundPlan([1024, 75], pad=False), [30.0, 0.3])[:-1]] == [False, False]
    # the strip oracle's rule for its two-spacing walk
    rng = np.random.default_rng(0)
    for _ in range(200):
        spacings = np.sort(rng.uniform(30, 1500, 2))[::-1]
        d = float(rng.uniform(0.05, 400))
        plan = _RoundPlan(spacings)
        plan.due(); plan.advance(d)
        nxt = plan.due(); plan.close()
        next_pos = np.searchsorted(-spacings, -4 * d) - 1
        assert nxt is not None and nxt[2] == ((min(next_pos, 1) > 1) if next_pos > 0 else True)


def test_signed_area_host_loop_equals_the_numpy_statement():
    """fb_signed_area (whole meshes) against cross(p1 - p0, p2 - p1) of common.py:672-676: the same bits, negative indices
    as numpy takes them, an index outside the vertex list refused"""
    rng = np.random.default_rng(8)
    V, T = 30000, 70000
    v = rng.standard_normal((V, 2)) * 300
    t = rng.integers(0, V, (T, 3)).astype(np.int32)
    t[11] = [-1, -V, 5]
    p = v[t]
    exp = common.cross2d(p[:, 1, :] - p[:, 0, :], p[:, 2, :] - p[:, 1, :])
    np.testing.assert_array_equal(common.signed_area(v, t), exp)
    np.testing.assert_array_equal(common.signed_area(v[:, ::-1][:, ::-1], t[::2]), exp[::2])        # non-contiguous views
    np.testing.assert_array_equal(common.signed_area(v, t[:100]), exp[:100])                         # small: the numpy statement itself
    t[20, 2] = V
    with pytest.raises(IndexError):
        common.signed_area(v, t)


def test_block_uncovered_threads_equal_one_thread():
    """fb_mesh_block_uncovered deals the blocks to host threads above 4 096 blocks: the same numbers block by block as
    the calls on the halves"""
    lib = _lib.load()
    rng = np.random.default_rng(2)
    n = 40
    xs = np.arange(n) * 10.0
    v = np.stack(np.meshgrid(xs, xs), -1).reshape(-1, 2) + rng.uniform(-2, 2, (n * n, 2))
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tri = np.ascontiguousarray(np.concatenate((np.stack((a, b, d), -1), np.stack((a, d, c), -1))), dtype=np.int32)
    NB, cap, h, w = 9000, 8, 12, 9
    org = np.ascontiguousarray(rng.uniform(-20, 10.0 * n, (NB, 2)))
    cand = np.ascontiguousarray(rng.integers(0, tri.shape[0], (NB, cap)), dtype=np.int32)
    cnt = np.ascontiguousarray(rng.integers(0, cap + 3, NB), dtype=np.int32)
    unc = np.empty(NB)
    assert lib.fb_mesh_block_uncovered(None, v.shape[0], _lib.ptr(v), _lib.ptr(tri), NB, _lib.ptr(org), h, w, cap, _lib.ptr(cand), _lib.ptr(cnt), _lib.ptr(unc)) == 0
    for lo, hi in ((0, 1500), (1500, 3000), (7000, 9000)):
        part = np.empty(hi - lo)
        o_, c_, k_ = np.ascontiguousarray(org[lo:hi]), np.ascontiguousarray(cand[lo:hi]), np.ascontiguousarray(cnt[lo:hi])
        assert lib.fb_mesh_block_uncovered(None, v.shape[0], _lib.ptr(v), _lib.ptr(tri), hi - lo, _lib.ptr(o_), h, w, cap, _lib.ptr(c_), _lib.ptr(k_), _lib.ptr(part)) == 0
        np.testing.assert_array_equal(part, unc[lo:hi])


def test_triangle_edge_deform_host_loop_equals_the_numpy_statement():
    """Mesh.triangle_edge_deform through fb_tri_edge_ratio (meshes of 1 024 triangles and more) against the numpy statement of
    mesh.py:1966-1976, bit for bit"""
    from scipy.spatial import Delaunay
    from feabas_amd.mesh import Mesh
    rng = np.random.default_rng(12)
    v = rng.uniform(0, 500, (1500, 2))
    t = Delaunay(v).simplices.astype(np.int32)
    assert t.shape[0] >= 1024
    m = Mesh(v, t, uid=0)
    m.set_vertices(v + rng.normal(0, 0.8, v.shape), 1)
    got = m.triangle_edge_deform(gear=(0, 1))
    v0, v1 = m.vertices(0), m.vertices(1)
    tr = np.roll(t, 1, axis=-1)
    d0 = np.sum((v0[t] - v0[tr]) ** 2, axis=-1); d1 = np.sum((v1[t] - v1[tr]) ** 2, axis=-1)
    np.testing.assert_array_equal(got, np.exp(np.max(np.abs(0.5 * np.log(d1 / d0)), axis=-1)))


def test_optimize_linear_does_not_apply_a_runaway_field():
    """SLM._solution_is_sane: a solved field that is not finite, or that moves nodes of a FLOATING sub-system (link-connected
    meshes without a locked member) by more than a thousand mesh extents (the null-space drift of a floating system pushed past
    what doubles can give), is not applied -- downstream steps size their buffers by where the meshes are; meshes linked to a
    locked one are pinned and not screened for size; a free mesh that is NOT linked to the locked one still floats"""
    from feabas_amd import mesh, optimizer
    from oracle import fem_ref
    rng = np.random.default_rng(0)
    v, t = fem_ref.grid_mesh(5, 4, 10.0)
    m0 = mesh.Mesh(v, t, uid=0); m1 = mesh.Mesh(v + 1.0, t, uid=1); m2 = mesh.Mesh(v + 2.0, t, uid=2)
    tid = rng.integers(0, t.shape[0], 6); B = rng.dirichlet((1, 1, 1), 6)
    slm = optimizer.SLM([m0, m1, m2], [optimizer.Link(m0, m1, tid, tid, B, B)])
    nv = m0.num_vertices
    n = 2 * 3 * nv
    slm.last_solve = {}
    assert slm._solution_is_sane(np.zeros(n)) and slm._solution_is_sane(np.full(n, 500.0)) and 'rejected' not in slm.last_solve
    assert not slm._solution_is_sane(np.full(n, 1e9)) and 'extent' in slm.last_solve['rejected']
    bad = np.zeros(n); bad[3] = np.nan
    assert not slm._solution_is_sane(bad) and slm.last_solve['rejected'] == 'not finite'
    m0.locked = True                                          # m1 is pinned through its link to m0; m2 floats on its own
    n = 2 * 2 * nv
    big1 = np.zeros(n); big1[:2 * nv] = 1e9
    big2 = np.zeros(n); big2[2 * nv:] = 1e9
    slm.last_solve = {}
    assert slm._solution_is_sane(big1) and 'rejected' not in slm.last_solve
    assert not slm._solution_is_sane(big2) and 'floating' in slm.last_solve['rejected']
    bad = np.zeros(n); bad[1] = np.inf
    assert not slm._solution_is_sane(bad)


def test_held_dof_selectors_fold_into_groups_like_the_reference():
    """optimize_linear(groupings=, remove_extra_dof= / remove_material_dof=): the selector over the free meshes' degrees of freedom
    in the layout of the grouped system, `edc = (T_m @ edc) > 0` (optimizer.py:1412-1413), against the oracle's T_m (pinned by
    golden G21 from the reference)"""
    from scipy import sparse
    from feabas_amd import mesh, optimizer
    from oracle import fem_ref
    g = load_golden('g21_grouped_dof.npz')
    ms = [mesh.Mesh(g[f'm{k}_v'], g[f'm{k}_t'], uid=k, soft_factor=float(g[f'm{k}_soft'])) for k in range(3)]
    rs = [fem_ref.RefMesh(g[f'm{k}_v'], g[f'm{k}_t'], uid=k) for k in range(3)]
    links, rlinks = [], []
    for k in range(2):
        a, b = g[f'l{k}_ab']
        args = (g[f'l{k}_tid0'], g[f'l{k}_tid1'], g[f'l{k}_B0'], g[f'l{k}_B1'])
        links.append(optimizer.Link(ms[a], ms[b], *args, weight=g[f'l{k}_w']))
        rlinks.append(fem_ref.RefLink(rs[a], rs[b], *args, weight=g[f'l{k}_w']))
    slm = optimizer.SLM(ms, links)

    def oracle_fold(edc, groupings, meshes):
        group_u, indx, group_nm, _ = np.unique(groupings, return_index=True, return_inverse=True, return_counts=True)
        vnum = np.array([meshes[k].num_vertices * 2 for k in indx])
        goff = np.concatenate(([0], np.cumsum(vnum)[:-1]))
        i0, i1, cur = [], [], 0
        for m, gio in zip(meshes, goff[group_nm]):
            sz = 2 * m.num_vertices
            i0.append(np.arange(cur, cur + sz)); i1.append(np.arange(gio, gio + sz)); cur += sz
        T = sparse.csr_matrix((np.ones(cur, dtype=np.float32), (np.concatenate(i1), np.concatenate(i0))), shape=(int(vnum.sum()), cur))
        return np.asarray(T @ edc.astype(np.float32)).ravel() > 0
    edc = fem_ref.extra_dof_selector(rs, rlinks)
    held = slm._extra_dof_mask(g['groupings'])
    np.testing.assert_array_equal(held, oracle_fold(edc, g['groupings'], rs))
    assert held.size == 2 * (ms[0].num_vertices + ms[1].num_vertices) and list(np.flatnonzero(~held)) == [0, 1, 2]
    np.testing.assert_array_equal(slm._extra_dof_mask(None), edc)
    # a hold on one member of a group is undone by a member that does not hold it: three equal meshes, 0 and 1 grouped
    eq = [mesh.Mesh(g['m1_v'] + k, g['m1_t'], uid=k) for k in range(3)]
    tid = np.arange(10); B = np.full((10, 3), 1 / 3)
    slm2 = optimizer.SLM(eq, [optimizer.Link(eq[0], eq[2], tid, tid, B, B), optimizer.Link(eq[1], eq[2], tid, tid, B, B)])
    assert slm2._extra_dof_mask(np.array([0, 0, 1])).all() and (~slm2._extra_dof_mask(np.array([0, 1, 1]))).sum() == 3
    # named materials: held vertices of a region, folded the same way
    ids = np.zeros(eq[0].num_triangles, dtype=np.int32); ids[:6] = 5
    eq[1].material_ids = ids; eq[1].material_names = {'default': 0, 'hold': 5}
    sel = slm2._material_dof_mask('hold', None)
    vheld = np.unique(eq[1].triangles[:6])
    assert (~sel).sum() == 2 * vheld.size
    grouped = slm2._material_dof_mask('hold', np.array([0, 1, 1]))          # mesh 2 (same group) does not hold them: free again
    assert grouped.all() and grouped.size == 2 * 2 * eq[0].num_vertices
    eq[2].material_ids = ids; eq[2].material_names = {'default': 0, 'hold': 5}
    grouped = slm2._material_dof_mask('hold', np.array([0, 1, 1]))
    assert (~grouped).sum() == 2 * vheld.size and not grouped[2 * eq[0].num_vertices + 2 * vheld[0]]


def test_round_stepper_against_the_walks_of_the_reference():
    """golden G22: the coarse-to-fine walk captured from the REFERENCE's own iterative_xcorr_matcher_w_mesh (matcher.py:567-716;
    its block matcher replaced by a script of displacements, everything else as it stands) against the library's stepper
    (fb_schedule_*, what every block matcher of this package walks with): the same rounds -- spacing, padding, the sub-pixel
    round -- for jumps, clipped and allowed skips, dwelling, the enlarged extra round, a fixed pad, unsorted spacing lists"""
    import json
    from feabas_amd.matcher import _RoundPlan
    g = load_golden('g22_schedule_walks.npz')
    scenarios = json.loads(bytes(g['scenarios']).decode())
    assert len(scenarios) >= 14
    for name, sc in scenarios.items():
        calls = g[name + '_calls']
        sp_list = np.sort(np.asarray(sc['spacings'], dtype=np.float64))[::-1]
        plan = _RoundPlan(sp_list, sc.get('allow_enlarge', False), sc.get('allow_dwell', 0), sc.get('max_spacing_skip', 0), sc.get('pad', None))
        got = []
        while True:
            r = plan.due()
            if r is None or len(got) > 4 * len(calls):
                break
            got.append(r)
            plan.advance(sc['dis'][min(len(got) - 1, len(sc['dis']) - 1)])
        plan.close()
        assert len(got) == calls.shape[0], (name, got, calls.tolist())
        for (sp, last, pad), (side, gpad, gsub, gtol, _) in zip(got, calls):
            assert bool(pad) == bool(gpad) and bool(last) == bool(gsub), (name, got, calls.tolist())
            # the reference's record carries the spacing through affine_approx_tol = 0.1 on the last spacing, max(1, 0.02 sp) before
            assert (gtol == 0.1 and sp == sp_list[-1]) if last else abs(max(1.0, 0.02 * sp) - gtol) < 1e-9, (name, sp, gtol)
            assert side <= np.ceil(sp) + 1e-9                       # (divide_bbox fits the blocks into the overlap: never larger than the spacing)


def test_python_sources_have_no_unbound_globals():
    """tools/check_names.py over the package, bench.py, the entry module, the oracle and the tests: a name that only a GPU run would
    reach (a side record of bench.py, an error path of the matcher) must not wait for the GPU box to turn out misspelt"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_names.py')], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:]


@pytest.mark.parametrize('case', ['ratio_margin', 'masked_mapped', 'pixel_margin'])
def test_g25_overlap_bookkeeping_vs_reference(monkeypatch, case):
    """stitcher.match_list_of_overlaps against the reference's Stitcher.subprocess_match_list_of_overlaps (stitcher.py:474-621,
    golden G25): around the same scripted matcher both sides crop the same strips for the same overlaps (margin as a ratio / in
    pixels, minimum width, boxes clipped to their tiles), derive the same masks from `maskout_val`, add the same offsets to
    the matches and file them under the same keys"""
    from feabas_amd import stitcher
    g = load_golden('g25_overlap_bookkeeping.npz')
    kw = {'ratio_margin': dict(margin=1.0, min_overlap_width=0),
          'masked_mapped': dict(margin=0.5, min_overlap_width=35, maskout_val=0, index_mapper=np.arange(6) + 100,
                                matcher_config=dict(compute_photometric=True, conf_thresh=0.4)),
          'pixel_margin': dict(margin=30, min_overlap_width=10, maskout_val=0)}[case]
    notes = []

    def scripted(img0, img1, mask0, mask1, cfg):
        """(tests/golden/make_golden.py::scripted_strip_result)"""
        k = len(notes)
        notes.append(np.array([img0.shape[0], img0.shape[1], img1.shape[0], img1.shape[1], int(img0.astype(np.int64).sum()), int(img1.astype(np.int64).sum()),
                               -1 if mask0 is None else int(np.count_nonzero(mask0)), -1 if mask1 is None else int(np.count_nonzero(mask1))], dtype=np.int64))
        if k % 4 == 3:
            return (None, None, cfg.get('conf_thresh', 0.3), None, None)
        h0, w0 = img0.shape[:2]; h1, w1 = img1.shape[:2]
        xy0 = np.array([[1.0, 2.0], [0.5 * w0, 0.5 * h0], [w0 - 3.0 + 0.25 * k, h0 - 2.0]])
        xy1 = np.array([[2.0, 1.0], [0.5 * w1 - 0.5, 0.5 * h1 + 0.125 * k], [w1 - 4.0, h1 - 1.0]])
        phtm = (10.0 + k, 20.0 + k, 3.0, 4.0) if cfg.get('compute_photometric', False) else None
        return (xy0, xy1, np.array([0.5, 0.75, 0.25 + 0.01 * k]), 0.01 * (k + 1), phtm)
    monkeypatch.setattr(stitcher, 'stitching_matcher_batch', lambda pairs, batch=32, threads=2, **cfg: [scripted(*pr, cfg) for pr in pairs])
    tiles = [g[f'tile{k}'] for k in range(6)]
    matches, strains, phtm, err = stitcher.match_list_of_overlaps(g['overlaps'], tiles, g['bboxes'], **kw)
    assert not err
    np.testing.assert_array_equal(np.stack(notes), g[f'{case}_calls'])
    keys = sorted(matches)
    np.testing.assert_array_equal(np.array(keys, dtype=np.int64).reshape(-1, 2), g[f'{case}_keys'])
    for j, key in enumerate(keys):
        np.testing.assert_array_equal(matches[key][0], g[f'{case}_m{j}_xy0'])
        np.testing.assert_array_equal(matches[key][1], g[f'{case}_m{j}_xy1'])
        np.testing.assert_array_equal(matches[key][2], g[f'{case}_m{j}_w'])
        assert strains[key] == float(g[f'{case}_m{j}_strain'])
        np.testing.assert_array_equal(np.array(phtm[key], dtype=np.float64) if key in phtm else np.empty(0), g[f'{case}_m{j}_phtm'])


def test_g26_slm_bookkeeping_vs_reference():
    """the host-side bookkeeping of SLM around the solves against the reference (golden G26, optimizer.py:688-754, 1678-1703,
    1758-1858): adjacency of the link graph (directional or not), connected subsystems, match residues, and a mesh that falls
    into two islands being replaced by its parts with the links dealt to the parts that hold their matches"""
    from feabas_amd import optimizer
    from feabas_amd.mesh import Mesh
    import feabas_amd.constant as const
    g = load_golden('g26_slm_bookkeeping.npz')
    m0 = Mesh(g['v0'], g['t0'], uid=0); m0.lock()
    m1 = Mesh(g['v1'], g['t1'], uid=1)
    m2 = Mesh(g['v2'], g['t2'], uid=2)
    m2.set_vertices(g['v2'] + g['d2'], const.MESH_GEAR_MOVING)
    opt = optimizer.SLM([m0, m1, m2])
    meshes = {0: m0, 1: m1, 2: m2}
    for name, (a, b) in {'01': (0, 1), '12': (1, 2), '02': (0, 2)}.items():
        opt.add_link(optimizer.Link(meshes[a], meshes[b], g[f'lk{name}_tid0'], g[f'lk{name}_tid1'], g[f'lk{name}_B0'], g[f'lk{name}_B1'], weight=g[f'lk{name}_w']))

    def check(tag):
        assert len(opt.links) == int(g[f'{tag}_nlinks'])
        np.testing.assert_allclose([m.uid for m in opt.meshes], g[f'{tag}_mesh_uids'], atol=1e-12)
        np.testing.assert_array_equal([m.num_triangles for m in opt.meshes], g[f'{tag}_mesh_ntri'])
        for k, lk in enumerate(opt.links):
            np.testing.assert_allclose(lk.uids, g[f'{tag}_l{k}_uids'], atol=1e-12)
            np.testing.assert_allclose(lk.xy0(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True), g[f'{tag}_l{k}_xy0'], atol=1e-9)
            np.testing.assert_allclose(lk.xy1(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True), g[f'{tag}_l{k}_xy1'], atol=1e-9)
            np.testing.assert_allclose(lk.weight(use_mask=False), g[f'{tag}_l{k}_w'], atol=1e-12)
        np.testing.assert_allclose(opt.linkage_adjacency().toarray(), g[f'{tag}_adj'], atol=1e-9)
        np.testing.assert_allclose(opt.linkage_adjacency(directional=True).toarray(), g[f'{tag}_adj_dir'], atol=1e-9)
        lab, ncomp = opt.connected_subsystems
        assert ncomp == int(g[f'{tag}_ncomp'])
        np.testing.assert_array_equal(lab, g[f'{tag}_labels'])
        for q in (0, 0.75, 1):
            np.testing.assert_allclose(opt.match_residues(gear=const.MESH_GEAR_MOVING, quantile=q), g[f'{tag}_res_q{q}'], atol=1e-9)
        np.testing.assert_allclose(opt.match_residues(gear=const.MESH_GEAR_INITIAL, quantile=0.5), g[f'{tag}_res_init'], atol=1e-9)
    check('whole')
    assert opt.divide_disconnected_submeshes() == bool(g['divided'])
    check('parts')


def test_g27_mesh_gears_vs_reference():
    """Mesh's gears against the reference (golden G27, mesh.py:1189-1330, 2232-2413): the walk of tests/golden/walks.py -- masked and
    unmasked translations, fields and affine maps between and inside gears, a gear that is not set falling back to the one below,
    a locked mesh ignoring everything -- leaves the same coordinates and offsets at every gear after every step; areas,
    deformation measures, bounds, connectivity by vertex and by edge, division into parts, sub-mesh"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from walks import g27_mesh_walk
    from feabas_amd.mesh import Mesh
    import feabas_amd.constant as const
    g = load_golden('g27_mesh_gears.npz')
    gears = dict(i=const.MESH_GEAR_INITIAL, f=const.MESH_GEAR_FIXED, m=const.MESH_GEAR_MOVING, s=const.MESH_GEAR_STAGING)
    steps = []

    def record(tag, m):
        steps.append(tag)
        for k, gear in gears.items():
            np.testing.assert_allclose(m.vertices_w_offset(gear), g[f'{tag}_{k}_vo'], atol=1e-10, err_msg=f'{tag} {k}')
            np.testing.assert_allclose(np.asarray(m.offset(gear), dtype=np.float64).reshape(1, 2), g[f'{tag}_{k}_off'], atol=1e-10, err_msg=f'{tag} {k} offset')
        np.testing.assert_allclose(m.estimate_translation(), g[f'{tag}_est'], atol=1e-10)
        np.testing.assert_allclose(m.estimate_translation(gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_STAGING)), g[f'{tag}_est_fs'], atol=1e-10)
    m, v, tri, mask = g27_mesh_walk(Mesh, const, record)
    assert len(steps) == 18
    np.testing.assert_array_equal(v, g['v']); np.testing.assert_array_equal(tri, g['tri'])
    for k, gear in gears.items():
        np.testing.assert_allclose(m.triangle_areas(gear=gear), g[f'areas_{k}'], rtol=1e-12)
        np.testing.assert_allclose(m.bbox(gear=gear), g[f'bbox_{k}'], atol=1e-10)
        np.testing.assert_allclose(m.bbox(gear=gear, offsetting=False), g[f'bbox_{k}_raw'], atol=1e-10)
    np.testing.assert_allclose(m.triangle_area_deform(gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_MOVING)), g['area_deform'], rtol=1e-12)
    np.testing.assert_allclose(m.triangle_edge_deform(gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_MOVING)), g['edge_deform'], rtol=1e-12)
    mb = Mesh(g['bow_v'], g['bow_t'], uid=7)

    def same_partition(a, b):
        a, b = np.asarray(a), np.asarray(b)
        return np.array_equal(a[:, None] == a[None, :], b[:, None] == b[None, :])
    nv, lv = mb.connected_vertices()
    nt, lt = mb.connected_triangles()
    assert nv == int(g['bow_nv']) and nt == int(g['bow_nt']) and same_partition(lv, g['bow_lv']) and same_partition(lt, g['bow_lt'])
    assert nv == 2 and nt == 3                               # the bow tie: one component by vertex, two by edge
    parts = mb.divide_disconnected_mesh()
    np.testing.assert_allclose([p.uid for p in parts], g['bow_part_uids'], atol=1e-12)
    for k, p in enumerate(parts):
        np.testing.assert_array_equal(p.vertices_w_offset(const.MESH_GEAR_INITIAL)[p.triangles], g[f'bow_part{k}_v'][g[f'bow_part{k}_t']])
    sub = m.submesh(g['sub_tmask'], uid=3.5)
    for k, gear in gears.items():
        np.testing.assert_allclose(sub.vertices_w_offset(gear)[sub.triangles], g[f'sub_{k}_vo'][g['sub_t']], atol=1e-10)


def test_cxx_general_mesh_block_affines_match_numpy():
    """fb_mesh_block_affines (host C++, no device work, ctx = NULL) -- the per-block affine fits and the tolerance test of
    MeshRenderer.crop_field for a GENERAL triangulation (renderer.py:397-416, 499-511) -- against the numpy statement
    deformed.block_affines (pinned to the renderer restatement by the tests above) on an irregular Delaunay mesh with a
    smooth + a few rough node displacements: tiers 2 / 3 and the affine maps of the tier-2 blocks"""
    from scipy.spatial import Delaunay
    from feabas_amd import _lib, deformed
    lib = _lib.load()
    rng = np.random.default_rng(11)
    gx, gy = np.meshgrid(np.arange(0, 900, 60.0), np.arange(0, 700, 60.0))
    v_img = np.stack((gx.ravel(), gy.ravel()), axis=-1) + rng.uniform(-18, 18, (gx.size, 2))
    tris = np.ascontiguousarray(Delaunay(v_img).simplices, dtype=np.int32)
    U = np.stack((4.0 * np.sin(v_img[:, 1] / 180.0) + 1.5 * np.cos(v_img[:, 0] / 75.0), 3.0 * np.cos(v_img[:, 1] / 140.0)), axis=-1)
    rough = rng.integers(0, v_img.shape[0], 12)
    U[rough] += rng.normal(0, 1.2, (12, 2))
    v_mov = np.ascontiguousarray(v_img + U + np.array([7.0, -4.0]))
    h, w = 70, 90
    org = np.ascontiguousarray(np.stack(np.meshgrid(np.arange(20, 760, 95.0), np.arange(15, 600, 80.0)), axis=-1).reshape(-1, 2))
    nb = org.shape[0]
    bboxes = np.concatenate((org, org + np.array([w, h])), axis=1)
    # candidates: every triangle whose bounding box touches the block's (a superset of the triangles that intersect it, like
    # the lists of fb_mesh_candidates_dev)
    p = v_mov[tris]
    tlo, thi = p.min(axis=1), p.max(axis=1)
    box = bboxes - 0.5
    touch = (tlo[None, :, 0] <= box[:, None, 2]) & (thi[None, :, 0] >= box[:, None, 0]) & (tlo[None, :, 1] <= box[:, None, 3]) & (thi[None, :, 1] >= box[:, None, 1])
    count = np.ascontiguousarray(touch.sum(axis=1), dtype=np.int32)
    cap = int(count.max())
    cand = np.full((nb, cap), -1, dtype=np.int32)
    for b in range(nb):
        cand[b, :count[b]] = np.flatnonzero(touch[b])
    seen = set()
    for tol in (0.05, 0.3, 1.0, 3.0):
        tier = np.full(nb, 3, dtype=np.int32); A6 = np.zeros((nb, 6))
        assert lib.fb_mesh_block_affines(None, v_mov.shape[0], _lib.ptr(v_mov), _lib.ptr(np.ascontiguousarray(v_img)), _lib.ptr(tris), nb, _lib.ptr(org), h, w, cap,
                                         _lib.ptr(cand), _lib.ptr(count), tol, _lib.ptr(tier), _lib.ptr(A6)) == 0
        # (a tolerance so loose that the ONE global fit passes is decided before the blocks are looked at: MeshRenderer._tiers)
        t2, A, _ = deformed.block_affines(v_mov, v_img, tris, bboxes, tol)
        if (t2 == 1).all():
            continue
        assert not (tier == -1).any()
        np.testing.assert_array_equal(tier, t2)
        seen |= set(tier.tolist())
        a = tier == 2
        ref = np.stack((A[:, 0, 0], A[:, 1, 0], A[:, 2, 0], A[:, 0, 1], A[:, 1, 1], A[:, 2, 1]), -1)
        np.testing.assert_allclose(A6[a], ref[a], atol=1e-9)
    assert seen == {2, 3}


def test_context_destroy_hooks_empty_the_matcher_caches(monkeypatch):
    """a context that goes away takes the per-pair matchers, the pool and the batch workers made under it along, freed while it is
    still current -- and only those (no GPU: stand-ins for the handles and the library)"""
    from feabas_amd import matcher as fm
    log = []

    class Thing:
        def __init__(self, name):
            self.name = name

        def free(self):
            log.append((self.name, _lib.ctx()))

    class FakeLib:
        def fb_destroy(self, h):
            log.append(('destroy', h))
    ha, hb, hw = object(), object(), object()
    monkeypatch.setattr(_lib, 'load', lambda: FakeLib())
    monkeypatch.setattr(_lib, '_ctx', hb)
    monkeypatch.setattr(fm, '_pair_matchers', {('k1', id(ha)): Thing('pair_a'), ('k2', id(hb)): Thing('pair_b')})
    monkeypatch.setattr(fm, '_pools', {id(ha): Thing('pool_a'), id(hb): Thing('pool_b')})
    monkeypatch.setattr(fm, '_batch_workers', {(id(ha), 'slots'): dict(io=[(Thing('pin_a'), Thing('dev_a'))]),
                                               (id(ha), 0): dict(ctx=ha, state=dict(matchers={('k', 32): Thing('matcher_a0')}, pool=Thing('wpool_a0'))),
                                               (id(ha), 1): dict(ctx=hw, state=dict(matchers={('k', 8): Thing('matcher_a1')})),
                                               (id(hb), 0): dict(ctx=hb, state=dict(matchers={('k', 32): Thing('matcher_b0')}))})
    _lib.use_context(None)
    _lib.destroy_context(ha)
    names = [n for n, _ in log]
    assert set(names) == {'pair_a', 'pool_a', 'pin_a', 'dev_a', 'matcher_a0', 'wpool_a0', 'matcher_a1', 'destroy'}
    assert [c for n, c in log if n in ('pair_a', 'pool_a', 'matcher_a0', 'wpool_a0')] == [ha] * 4       # freed with the dying context current
    assert [c for n, c in log if n == 'matcher_a1'] == [hw]                                               # a worker's matcher under the worker's context
    assert [h for n, h in log if n == 'destroy'] == [hw, ha]                                              # the worker's context first, then the owner's
    assert list(fm._pair_matchers) == [('k2', id(hb))] and list(fm._pools) == [id(hb)] and list(fm._batch_workers) == [(id(hb), 0)]
    assert getattr(_lib._tls, 'ctx', None) is None                                                        # the caller's context is current again


def _mg_level(rng, grids, jitter=0.2):
    """a level of the multigrid set-up as host arrays: several structured triangulated grids (one mesh each), block pattern =
    vertex adjacency through the triangles + the diagonal"""
    from scipy import sparse
    xy, comp, rows, cols = [], [], [], []
    off = 0
    for k, (nx, ny, step, ox, oy) in enumerate(grids):
        gx, gy = np.meshgrid(ox + step * np.arange(nx), oy + step * np.arange(ny))
        v = np.stack((gx.ravel(), gy.ravel()), axis=-1) + rng.uniform(-jitter, jitter, (nx * ny, 2)) * step
        idx = np.arange(nx * ny).reshape(ny, nx)
        a, b, c_, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
        tri = np.concatenate((np.stack((a, b, d), -1), np.stack((a, d, c_), -1)))
        for i in range(3):
            for j in range(3):
                rows.append(tri[:, i] + off); cols.append(tri[:, j] + off)
        xy.append(v); comp.append(np.full(nx * ny, k, dtype=np.int32)); off += nx * ny
    n = off
    A = sparse.csr_matrix((np.ones(sum(r.size for r in rows)), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    A.sum_duplicates(); A.sort_indices()
    return np.ascontiguousarray(np.concatenate(xy)), np.ascontiguousarray(np.concatenate(comp)), A


def _mg_coarsen(lib, xy, comp, A, bs=2, fine_scale=1.0):
    n = xy.shape[0]
    rowptr = np.ascontiguousarray(A.indptr, dtype=np.int32); col = np.ascontiguousarray(A.indices, dtype=np.int32)
    nc, maxc, cell, cnnz = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double(), ctypes.c_int64()
    agg = np.empty(n, np.int32); rel = np.empty((n, 2))
    args = (n, bs, _lib.ptr(xy), _lib.ptr(comp), _lib.ptr(rowptr), _lib.ptr(col), fine_scale, ctypes.byref(nc), ctypes.byref(cell), _lib.ptr(agg), _lib.ptr(rel))
    assert lib.fb_debug_mg_coarsen(*args, None, None, None, ctypes.byref(cnnz), None, 0, ctypes.byref(maxc)) == 0
    cxy = np.empty((nc.value, 2)); ccomp = np.empty(nc.value, np.int32); crow = np.empty(nc.value + 1, np.int32); ccol = np.empty(cnnz.value, np.int32)
    assert lib.fb_debug_mg_coarsen(*args, _lib.ptr(cxy), _lib.ptr(ccomp), _lib.ptr(crow), ctypes.byref(cnnz), _lib.ptr(ccol), ccol.size, ctypes.byref(maxc)) == 0
    return dict(nc=nc.value, cell=cell.value, agg=agg, rel=rel, cxy=cxy, ccomp=ccomp, crow=crow, ccol=ccol, maxc=maxc.value)


def check_mg_coarsening(xy, comp, A, r):
    """what a coarsening step of csrc/fb_mg.inc promises, whatever the number of host threads that produced it"""
    from scipy import sparse
    n, nc, agg = xy.shape[0], r['nc'], r['agg']
    assert 0 < nc <= n and agg.min() == 0 and agg.max() == nc - 1
    first = np.full(nc, n, dtype=np.int64)
    np.minimum.at(first, agg, np.arange(n))
    assert np.all(np.diff(first) > 0)                                   # aggregates are numbered in the order of their first node
    assert np.array_equal(r['ccomp'], comp[first])
    assert np.array_equal(comp, r['ccomp'][agg])                        # an aggregate never joins two meshes
    cnt = np.bincount(agg, minlength=nc)
    cen = np.stack((np.bincount(agg, xy[:, 0], nc), np.bincount(agg, xy[:, 1], nc)), -1) / cnt[:, None]
    np.testing.assert_allclose(r['cxy'], cen, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(r['rel'], (xy - cen[agg]) / r['cell'], rtol=1e-10, atol=1e-10)
    assert np.abs(r['rel']).max() <= 1.0 + 1e-9                         # the nodes of an aggregate lie in one cell
    P = sparse.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
    Cp = (P.T @ A @ P).tocsr(); Cp.sort_indices()
    np.testing.assert_array_equal(r['crow'], Cp.indptr)
    np.testing.assert_array_equal(r['ccol'], Cp.indices)                # the Galerkin pattern, rows sorted
    assert r['maxc'] == int(np.diff(Cp.indptr).max())


def test_cxx_multigrid_coarsening_host_half():
    """fb_debug_mg_coarsen (host only): one coarsening step of the multigrid set-up -- grid-cell aggregates per mesh, relative node
    positions, the Galerkin pattern -- on a small two-mesh level and on one large enough for its host loops to run on several
    threads (>= 131072 nodes / >= 2048 aggregates): invariants against numpy / scipy"""
    lib = _lib.load_test()
    rng = np.random.default_rng(3)
    xy, comp, A = _mg_level(rng, [(23, 17, 10.0, 0.0, 0.0), (12, 31, 7.0, 400.0, -50.0)])
    r = _mg_coarsen(lib, xy, comp, A)
    check_mg_coarsening(xy, comp, A, r)
    assert 4 <= r['nc'] < xy.shape[0] // 8
    xy, comp, A = _mg_level(rng, [(640, 500, 5.0, 0.0, 0.0), (150, 160, 5.0, 5000.0, 100.0)])
    r = _mg_coarsen(lib, xy, comp, A)
    check_mg_coarsening(xy, comp, A, r)
    assert r['nc'] >= 2048
    # a coarse level (3 degrees of freedom per node) coarsens like a fine one
    r3 = _mg_coarsen(lib, np.ascontiguousarray(r['cxy']), np.ascontiguousarray(r['ccomp']),
                     __import__('scipy.sparse', fromlist=['csr_matrix']).csr_matrix((np.ones(r['ccol'].size), r['ccol'], r['crow']), shape=(r['nc'], r['nc'])), bs=3, fine_scale=r['cell'])
    assert r3['nc'] < r['nc']


def test_cxx_strip_matcher_host_arithmetic_matches_python():
    """the host arithmetic of fb_match_strips (host-only hooks, no device): the rigid fits of every pair's matches against
    common.fit_affine (pinned to the reference's spatial.fit_affine by golden G13), the automatic spacings against
    matcher.auto_spacings (matcher.py:243-251) and the node grid against Mesh.from_bbox(cartesian=True)"""
    from feabas_amd.mesh import Mesh
    from feabas_amd.stitch_pipeline import grid_counts
    lib = _lib.load_test()
    rng = np.random.default_rng(21)
    P = 9
    pid, p0, p1, wt = [], [], [], []
    for p in range(P):
        n = [40, 3, 2, 0, 25, 60, 5, 12, 30][p]
        q = rng.uniform(0, 500, (n, 2)) * ([1, 1] if p != 6 else [1, 0])           # pair 6: collinear matches (rank deficient)
        th = rng.uniform(-0.05, 0.05)
        Rm = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
        t = q @ Rm * (1.0 + rng.uniform(-0.01, 0.01)) + rng.uniform(-20, 20, 2) + rng.normal(0, 0.3, (n, 2))
        if p == 7:
            t = t * np.array([-1.0, 1.0])                                            # a reflection: not a rigid motion
        pid.append(np.full(n, p, np.int32)); p1.append(q); p0.append(t); wt.append(rng.uniform(0.1, 1.0, n).astype(np.float32))
    pid, p0, p1, wt = np.concatenate(pid), np.ascontiguousarray(np.concatenate(p0)), np.ascontiguousarray(np.concatenate(p1)), np.concatenate(wt)
    R = np.empty((P, 3, 3)); bad = np.empty(P, np.uint8)
    assert lib.fb_debug_rigid_fits(P, pid.size, _lib.ptr(pid), _lib.ptr(p0), _lib.ptr(p1), _lib.ptr(wt), _lib.ptr(R), _lib.ptr(bad)) == 0
    assert bad.tolist() == [0, 0, 1, 0, 0, 0, 1, 1, 0]          # two matches, collinear matches, a reflection: the statement-by-statement route
    np.testing.assert_array_equal(R[3], np.eye(3))                 # a pair without matches is not visited
    for p in range(P):
        if bad[p] or p == 3:
            continue
        s_ = pid == p
        _, Rr = common.fit_affine(p0[s_], p1[s_], return_rigid=True, weight=wt[s_].astype(np.float64), svd_clip=(1, 1))
        np.testing.assert_allclose(R[p], Rr, atol=1e-9, rtol=1e-9)
    for H, W in [(4096, 510), (510, 4096), (1536, 120), (60, 60), (90, 700), (2048, 255), (75, 76), (301, 299)]:
        cnt = ctypes.c_int()
        out = np.empty(8)
        assert lib.fb_debug_auto_spacings(H, W, _lib.ptr(out), 8, ctypes.byref(cnt)) == 0
        np.testing.assert_allclose(out[:cnt.value], np.sort(matcher.auto_spacings((H, W), (H, W)))[::-1], rtol=1e-14)
        for ms in (25.0, 40.0, 75.0, float(out[cnt.value - 1])):
            for mnb in (1, 2, 3):
                nx, ny = ctypes.c_int(), ctypes.c_int()
                assert lib.fb_debug_grid_counts(H, W, ms, mnb, ctypes.byref(nx), ctypes.byref(ny)) == 0
                assert (nx.value, ny.value) == grid_counts(H, W, ms, mnb)
                m = Mesh.from_bbox((0, 0, W, H), cartesian=True, mesh_size=ms, min_num_blocks=mnb)
                assert (nx.value, ny.value) == (m.grid_xs.size, m.grid_ys.size)


class _FakeBuf:
    def __init__(self, nbytes=0):
        self.nbytes = int(nbytes); self.ptr = ctypes.c_void_p(1)

    def offset(self, n):
        return ctypes.c_void_p(1)

    def free(self):
        self.ptr = None


def _install_fake_device(monkeypatch, match_fn):
    """stand-ins for everything stitching_matcher_batch touches on the device: contexts, staging buffers, the packing / copy entries
    and the strip matcher (match_fn(call number, number of pairs) -> per-pair results or an exception)"""
    import threading
    from feabas_amd import matcher as fm, stitch_pipeline as sp
    calls = [0]
    lock = threading.Lock()

    class FakeLib:
        def __getattr__(self, name):
            if name.startswith('fb_'):
                return lambda *a: 0
            raise AttributeError(name)

    class FakeMatcher:
        def __init__(self, n, H, W, pool=None, **opts):
            self.n = n

        def match(self, d0, d1, masks0=None, masks1=None, compute_photometric=False):
            with lock:
                calls[0] += 1
                k = calls[0]
            return match_fn(k, self.n)

        def free(self):
            pass

        @staticmethod
        def per_pair(out):
            return out['per']

    class FakePool:
        def free(self):
            pass
    main = object()
    monkeypatch.setattr(_lib, 'load', lambda: FakeLib())
    monkeypatch.setattr(_lib, '_ctx', main)
    monkeypatch.setattr(_lib, 'ctx', lambda device=None: getattr(_lib._tls, 'ctx', None) or main)
    monkeypatch.setattr(_lib, 'new_context', lambda device=None: object())
    monkeypatch.setattr(_lib, 'destroy_context', lambda h: None)
    monkeypatch.setattr(_lib, 'PinnedBuffer', _FakeBuf)
    monkeypatch.setattr(_lib, 'DeviceBuffer', _FakeBuf)
    monkeypatch.setattr(sp, 'StripBatchMatcher', FakeMatcher)
    monkeypatch.setattr(sp, 'MatcherPool', FakePool)
    monkeypatch.setattr(fm, '_batch_workers', {})
    return calls


@pytest.mark.parametrize('threads', [1, 2, 3, 4, 6])
def test_stitching_matcher_batch_queues_without_a_gpu(monkeypatch, threads):
    """the host side of stitching_matcher_batch -- chunks, loader and matcher threads, staging slots, end marks -- on stand-ins for the
    device objects: every pair comes back in input order whatever the number of threads; a matcher call that raises (like a device
    out-of-memory in one chunk) surfaces in the caller instead of leaving the other threads waiting, and the next call
    works again.  (tests/test_gpu_pipeline.py holds the same scenario on the device.)"""
    from feabas_amd import matcher as fm

    def ok(k, n):
        return dict(per=[dict(xy0=np.full((2, 2), float(k)), xy1=np.zeros((2, 2)), weight=np.ones(2), strain=0.01, deferred=False) for _ in range(n)], phtm=None)
    calls = _install_fake_device(monkeypatch, ok)
    pairs = [(np.full((64, 32), k, np.uint8), np.full((64, 32), k + 1, np.uint8)) for k in range(23)]
    out = fm.stitching_matcher_batch(pairs, batch=4, threads=threads, sigma=2.5, coarse_downsample=0.5)
    assert len(out) == 23 and all(o[0] is not None and o[0].shape == (2, 2) for o in out) and calls[0] == 6

    def third_fails(k, n):
        if k == calls_before[0] + 3:
            raise RuntimeError('injected failure in one chunk')
        return ok(k, n)
    calls_before = [calls[0]]
    calls2 = _install_fake_device(monkeypatch, third_fails)
    calls_before[0] = calls2[0]
    import threading
    done = []

    def run():
        try:
            fm.stitching_matcher_batch(pairs, batch=2, threads=threads, sigma=2.5, coarse_downsample=0.5)
            done.append('returned')
        except RuntimeError as e:
            done.append(str(e))
    th = threading.Thread(target=run, daemon=True)
    th.start(); th.join(timeout=60)
    assert not th.is_alive(), 'stitching_matcher_batch hangs after a matcher failure'
    assert done == ['injected failure in one chunk']
    calls3 = _install_fake_device(monkeypatch, ok)
    out = fm.stitching_matcher_batch(pairs[:5], batch=2, threads=threads, sigma=2.5, coarse_downsample=0.5)
    assert len(out) == 5 and all(o[0] is not None for o in out) and calls3[0] == 3


def test_g28_affine_cascade_vs_reference():
    """SLM.optimize_affine_cascade (optimizer.py:1128-1189, host code in the reference and here) on the six-tile system of G15, every tile
    rotated / scaled / offset at the start gear: the order in which tiles are placed, each tile's rigid / clipped / affine fit onto its placed
    neighbours, the gears read and written, two locked tiles, a tile no link reaches -- golden G28, the walk of tests/golden/walks.py"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from walks import g28_cascade_walk
    from feabas_amd.mesh import Mesh
    from feabas_amd.optimizer import Link, SLM
    import feabas_amd.constant as const
    g = load_golden('g28_affine_cascade.npz')
    seen = []

    def record(tag, ms, modified):
        seen.append(tag)
        assert bool(modified) == bool(g[f'{tag}_modified'])
        for k, gear in (('f', const.MESH_GEAR_FIXED), ('m', const.MESH_GEAR_MOVING)):
            np.testing.assert_allclose(np.stack([m.vertices_w_offset(gear) for m in ms]), g[f'{tag}_{k}'], atol=1e-8, err_msg=f'{tag} {k}')
    g28_cascade_walk(Mesh, Link, SLM, const, dict(load_golden('g15_translation.npz')), record)
    assert len(seen) == 7


def test_g29_cartesian_node_grid_vs_reference():
    """Mesh.from_bbox(cartesian=True): the node grid (mesh.py:403-435: block counts with the aspect-ratio rule, linspace - 0.5) the
    reference hands to its mesher, for 110 boxes / mesh sizes / minimum block counts (golden G29); the library's grid_counts agrees.
    (The triangles are the mesher's -- `triangle`, absent -- so only the product's two-triangles-per-cell topology is checked.)"""
    from feabas_amd.mesh import Mesh
    import feabas_amd.constant as const
    g = load_golden('g29_cartesian_grid.npz')
    lib = _lib.load_test()
    for k, (x0, y0, x1, y1, ms, mnb) in enumerate(g['cases']):
        m = Mesh.from_bbox((x0, y0, x1, y1), cartesian=True, mesh_size=float(ms), min_num_blocks=int(mnb))
        want = g[f'c{k}_v']
        np.testing.assert_allclose(m.vertices_w_offset(const.MESH_GEAR_INITIAL), want, atol=1e-10, err_msg=str(k))
        nx, ny = m.grid_xs.size, m.grid_ys.size
        assert nx * ny == want.shape[0] and int(g[f'c{k}_nseg']) == (nx - 1) * ny + nx * (ny - 1)
        assert m.num_triangles == 2 * (nx - 1) * (ny - 1)
        if x0 == 0 and y0 == 0:
            cx, cy = ctypes.c_int(), ctypes.c_int()
            assert lib.fb_debug_grid_counts(int(y1), int(x1), float(ms), int(mnb), ctypes.byref(cx), ctypes.byref(cy)) == 0
            assert (cx.value, cy.value) == (nx, ny)


def test_g34_links_to_divided_meshes_vs_reference():
    """SLM.add_link_from_coordinates (optimizer.py:637-684) after a mesh fell into its connected parts: a link addressed to the parent uid
    is dealt to the parts that hold its matches (each match once with submesh_exclusive, else wherever it lands), a name that is loaded
    already is skipped unless check_duplicates is off, an unknown uid adds nothing, a part can be addressed itself -- golden G34"""
    from feabas_amd import optimizer
    from feabas_amd.mesh import Mesh
    import feabas_amd.constant as const
    g = load_golden('g34_links_to_divided_meshes.npz')
    g26 = load_golden('g26_slm_bookkeeping.npz')
    m0 = Mesh(g26['v0'], g26['t0'], uid=0); m0.lock()
    opt = optimizer.SLM([m0, Mesh(g26['v1'], g26['t1'], uid=1), Mesh(g26['v2'], g26['t2'], uid=2)])
    assert opt.divide_disconnected_submeshes() == bool(g['divided'])
    np.testing.assert_allclose([m.uid for m in opt.meshes], g['mesh_uids'], atol=1e-12)
    p = {k: (g[f'p{k}_xy0'], g[f'p{k}_xy1'], g[f'p{k}_w']) for k in ('02', '12')}
    res = [opt.add_link_from_coordinates(0, 2, *p['02'][:2], weight=p['02'][2], name='a'),
           opt.add_link_from_coordinates(1, 2, *p['12'][:2], weight=p['12'][2], name='b', submesh_exclusive=False),
           opt.add_link_from_coordinates(0, 2, *p['02'][:2], weight=p['02'][2], name='a'),
           opt.add_link_from_coordinates(0, 2, *p['02'][:2], weight=p['02'][2], name='a', check_duplicates=False),
           opt.add_link_from_coordinates(0, 7, *p['02'][:2], weight=p['02'][2]),
           opt.add_link_from_coordinates(2.1, 1, p['12'][1], p['12'][0], weight=p['12'][2])]
    assert res == g['added'].tolist() and len(opt.links) == int(g['nlinks'])
    for k, lk in enumerate(opt.links):
        np.testing.assert_allclose(lk.uids, g[f'l{k}_uids'], atol=1e-12)
        np.testing.assert_allclose(lk.xy0(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True), g[f'l{k}_xy0'], atol=1e-9)
        np.testing.assert_allclose(lk.xy1(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True), g[f'l{k}_xy1'], atol=1e-9)
        np.testing.assert_allclose(lk.weight(use_mask=False), g[f'l{k}_w'], atol=1e-7)


def test_g35_flag_outcasts_vs_reference():
    """SLM.flag_outcasts (optimizer.py:1604-1625: the meshes the aligner leaves out of a window, aligner.py:700) against golden G35: one
    subsystem, two subsystems with a locked tile, without any lock (the minority is cast out), with locks in two subsystems, and a second
    call on the flagged meshes (the reference then casts out everything that is not tied to a lock -- reproduced as it is)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from walks import g35_outcast_walk
    from feabas_amd.mesh import Mesh
    from feabas_amd.optimizer import Link, SLM
    g = load_golden('g35_outcasts.npz')
    seen = []

    def record(tag, flags, ms):
        seen.append(tag)
        np.testing.assert_array_equal(np.asarray(flags, dtype=bool), g[tag], err_msg=tag)
        np.testing.assert_array_equal([bool(getattr(m, 'is_outcast', False)) for m in ms], g[tag + '_kept'], err_msg=tag)
    g35_outcast_walk(Mesh, Link, SLM, dict(load_golden('g15_translation.npz')), record)
    assert len(seen) == 10


def test_region_raster_refuses_a_runaway_pair():
    """round 5's incident, host end (tools/repro_r05_incident.py shows the chain: a floating pair scaled x60 by a diverged
    relaxation -> 3.9e8 raster cells -> > 6 GB within seconds): _RegionPair.raster and the oracle's raster refuse more than 1e8
    cells with a message, before anything of that size is allocated"""
    import tracemalloc
    from feabas_amd import mesh
    from oracle import fem_ref, region_ref
    v, t = fem_ref.grid_mesh(21, 21, 100.0)
    c = v.mean(axis=0)
    big = (v - c) * 600.0 + c                                   # 1.2e6 px wide: 2.3e9 cells at the step of spacing / 4
    m0 = mesh.Mesh(big, t, uid=0); m1 = mesh.Mesh(big + 3.0, t, uid=1)
    tracemalloc.start()
    with pytest.raises(ValueError, match='not where images could be'):
        matcher.distribute_matching_blocks(m0, m1, 100.0, min_boundary_distance=20, shrink_factor=0.7)
    with pytest.raises(ValueError, match='common region of'):
        region_ref.distribute_matching_blocks(big, t, big + 3.0, t, 100.0)
    peak = tracemalloc.get_traced_memory()[1]
    tracemalloc.stop()
    assert peak < 64e6, peak
    nan = v.copy(); nan[5] = np.nan                               # a non-finite mesh has no common region with anything, or is refused
    try:
        e0, e1 = matcher.distribute_matching_blocks(mesh.Mesh(nan, t, uid=0), mesh.Mesh(v, t, uid=1), 100.0)
        assert e0.shape[0] == 0 and e1.shape[0] == 0
    except ValueError:
        pass
    # the same pair where images can be: served
    b0, b1 = matcher.distribute_matching_blocks(mesh.Mesh(v, t, uid=0), mesh.Mesh(v + 3.0, t, uid=1), 100.0)
    assert b0.shape[0] > 100 and b0.shape == b1.shape


@pytest.mark.parametrize('mode', ['thread', 'backstop'])
def test_rss_watchdog_ends_a_runaway_process(mode):
    """feabas_amd/_watchdog.py (every pytest run and bench.py start it): a process whose resident set passes the limit ends with
    exit code 3 (thread: with a stack dump) or is killed by the backstop child (works while a C call holds the GIL); a process
    under the limit is left alone and the backstop ends with it"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys, time, os\nsys.path.insert(0, %r)\nos.environ["FEABAS_RSS_LIMIT_GB"] = "0.5"\n'
            'from feabas_amd import _watchdog\n'
            '%s\nimport numpy as np\nkeep = []\n'
            'for i in range(int(sys.argv[1])):\n    keep.append(np.ones(1 << 24)); time.sleep(0.05)\n'
            'print("alive", flush=True)\n') % (root, '_watchdog.start()' if mode == 'thread' else 'p = _watchdog.start_backstop()')
    r = subprocess.run([sys.executable, '-c', code, '16'], capture_output=True, text=True, timeout=120)      # 2 GB in 128 MB pieces
    assert 'alive' not in r.stdout
    assert 'watchdog' in r.stderr and (r.returncode == 3 if mode == 'thread' else r.returncode == -9), (r.returncode, r.stderr[-300:])
    r = subprocess.run([sys.executable, '-c', code, '2'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and 'alive' in r.stdout and 'watchdog' not in r.stderr
