#!/usr/bin/env python3
"""Generate the golden fixtures tests/golden/*.npz from the REFERENCE itself.

Runs only in the build container (needs /root/reference; the GPU box never
sees it).  The reference is 100 % Python but imports C-backed packages that
are absent here (cv2, shapely, triangle, ...); they are replaced by MagicMock
modules -- every function captured below executes on
numpy/scipy alone.  Output = inputs + the reference's outputs, nothing else.

    cd /tmp && python /root/repo/tests/golden/make_golden.py
"""
import os
import sys
from unittest.mock import MagicMock

os.environ.setdefault('OMP_NUM_THREADS', '1')
for _name in ['cv2', 'h5py', 'shapely', 'shapely.geometry', 'shapely.ops', 'triangle', 'rtree',
              'rtree.index', 'pyamg', 'tensorstore', 'skimage', 'skimage.morphology', 'google',
              'google.cloud', 'google.cloud.storage', 'dask', 'dask.distributed', 'dask_jobqueue']:
    _m = MagicMock()
    _m.__path__ = []
    sys.modules[_name] = _m
sys.path.insert(0, '/root/reference')
os.chdir('/tmp')

import numpy as np                                   # noqa: E402
from scipy import sparse                             # noqa: E402
from scipy.ndimage import gaussian_filter            # noqa: E402
from scipy.spatial import Delaunay                   # noqa: E402

from feabas import matcher, common, optimizer, material   # noqa: E402
from feabas.mesh import Mesh                               # noqa: E402
import feabas.constant as const                            # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def texture(rng, h, w, sigma=1.5):
    """band-limited noise: smoothed white noise + a slow component."""
    a = gaussian_filter(rng.standard_normal((h, w)), sigma)
    b = gaussian_filter(rng.standard_normal((h, w)), 8 * sigma)
    t = a / a.std() + 0.7 * b / b.std()
    return t


def to_u8(t):
    return np.clip(128 + 40 * t / t.std(), 0, 255).astype(np.uint8)


def coo(M):
    M = sparse.coo_matrix(M)
    M.sum_duplicates()
    o = np.lexsort((M.col, M.row))
    return M.row[o].astype(np.int64), M.col[o].astype(np.int64), M.data[o]


# ----------------------------------------------------------------------- G1
def g1_xcorr():
    rng = np.random.default_rng(101)
    out = {}
    cases = {
        'A': ((75, 73), (75, 73), 6),
        'B': ((74, 72), (67, 75), 6),
        'C': ((128, 128), (128, 128), 4),
    }
    for name, (s0, s1, n) in cases.items():
        H = max(s0[0], s1[0]); W = max(s0[1], s1[1])
        big = common.masked_dog_filter(to_u8(texture(rng, 4 * H, 4 * W)), 2.5).astype(np.float32)
        # shifts: small, ~half block, and one exactly at the padded F/2 wrap
        shifts = [(0, 0), (3, -2), (-7, 5), (H // 2 - 1, -(W // 2 - 2)), (-(H // 2), W // 2), (11, 17)][:n]
        i0 = np.zeros((n,) + s0, np.float32)
        i1 = np.zeros((n,) + s1, np.float32)
        for k, (sy, sx) in enumerate(shifts):
            y0 = H + 5 * k; x0 = W + 3 * k
            i0[k] = big[y0:y0 + s0[0], x0:x0 + s0[1]]
            i1[k] = big[y0 + sy:y0 + sy + s1[0], x0 + sx:x0 + sx + s1[1]]
        i1 += 0.05 * i1.std() * rng.standard_normal(i1.shape).astype(np.float32)
        out[f'{name}_img0'] = i0
        out[f'{name}_img1'] = i1
        for pad in (True, False):
            for sub in (True, False):
                for cm in (0, 1, 2):
                    dx, dy, cf = matcher.xcorr_fft(i0, i1, conf_mode=cm, pad=pad, subpixel=sub)
                    key = f'{name}_p{int(pad)}_s{int(sub)}_c{cm}'
                    out[key + '_dx'] = np.asarray(dx, dtype=np.float64)
                    out[key + '_dy'] = np.asarray(dy, dtype=np.float64)
                    out[key + '_conf'] = np.asarray(cf)
    # 4-D (N,H,W,C) input
    i0 = np.stack([out['A_img0'][:3, :40, :36], out['A_img0'][:3, 20:60, 30:66]], axis=-1)
    i1 = np.stack([out['A_img1'][:3, :40, :36], out['A_img1'][:3, 20:60, 30:66]], axis=-1)
    out['D_img0'] = np.ascontiguousarray(i0)
    out['D_img1'] = np.ascontiguousarray(i1)
    for pad in (True, False):
        dx, dy, cf = matcher.xcorr_fft(i0, i1, conf_mode=2, pad=pad, subpixel=True)
        out[f'D_p{int(pad)}_s1_c2_dx'] = dx
        out[f'D_p{int(pad)}_s1_c2_dy'] = dy
        out[f'D_p{int(pad)}_s1_c2_conf'] = cf
    np.savez_compressed(os.path.join(OUT, 'g1_xcorr.npz'), **out)


# ----------------------------------------------------------------------- G2
def g2_dog():
    rng = np.random.default_rng(202)
    img = to_u8(texture(rng, 160, 128))
    blob = gaussian_filter(rng.standard_normal((160, 128)), 12) > 0.0
    blob[:, :6] = True
    out = {'img': img, 'mask': blob}
    for s in (1.25, 2.5, 3.5):
        out[f'dog_s{s}'] = common.masked_dog_filter(img, s)
    out['dog_masked_signed'] = common.masked_dog_filter(img, 2.5, mask=blob)
    out['dog_masked_unsigned'] = common.masked_dog_filter(img, 2.5, mask=blob, signed=False)
    stack = np.stack([to_u8(texture(rng, 64, 48)) for _ in range(3)])
    out['stack'] = stack
    out['dog_stack_s2.5'] = common.masked_dog_filter(stack, 2.5)
    fimg = texture(rng, 50, 70).astype(np.float32)
    out['fimg'] = fimg
    out['dog_fimg_s1.25'] = common.masked_dog_filter(fimg, 1.25)
    np.savez_compressed(os.path.join(OUT, 'g2_dog.npz'), **out)


# ----------------------------------------------------------------------- G3
def g3_global():
    rng = np.random.default_rng(303)
    big = to_u8(texture(rng, 520, 360))
    s0 = big[40:440, 30:290]
    s1 = big[40 - 7:440 - 7, 30 + 5:290 + 5]
    d0 = common.masked_dog_filter(s0, 1.25).astype(np.float32)
    d1 = common.masked_dog_filter(s1, 1.25).astype(np.float32)
    out = {'d0': d0, 'd1': d1}
    out['plain'] = np.array(matcher.global_translation_matcher(d0, d1, conf_thresh=0.3), dtype=np.float64)
    # force the 6-block fallback; a zero-variance block in img1 must be skipped
    e1 = d1.copy()
    e1[:134, :130] = 0
    out['e1'] = e1
    out['fallback'] = np.array(matcher.global_translation_matcher(d0, e1, conf_thresh=2.0), dtype=np.float64)
    # unequal strip sizes
    f1 = d1[:380, :250].copy()
    out['f1'] = f1
    out['unequal'] = np.array(matcher.global_translation_matcher(d0, f1, conf_thresh=2.0), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'g3_global.npz'), **out)


# ----------------------------------------------------------------------- FEM helpers
def grid(nx, ny, h, origin=(0.0, 0.0)):
    xs = origin[0] + h * np.arange(nx)
    ys = origin[1] + h * np.arange(ny)
    vx, vy = np.meshgrid(xs, ys)
    v = np.stack((vx.ravel(), vy.ravel()), axis=-1)
    idx = np.arange(nx * ny).reshape(ny, nx)
    a = idx[:-1, :-1].ravel(); b = idx[:-1, 1:].ravel(); c = idx[1:, :-1].ravel(); d = idx[1:, 1:].ravel()
    par = ((np.arange(nx - 1)[None, :] + np.arange(ny - 1)[:, None]) % 2).ravel().astype(bool)
    t0 = np.where(par[:, None], np.stack((a, b, c), -1), np.stack((a, b, d), -1))
    t1 = np.where(par[:, None], np.stack((b, d, c), -1), np.stack((a, d, c), -1))
    return v, np.concatenate((t0, t1), axis=0)


def mat_table(nu):
    d = dict(material.MATERIAL_DEFAULT)
    d['poisson_ratio'] = nu
    return material.MaterialTable(table={'default': d})


# ----------------------------------------------------------------------- G4 / G5
def g45_stiffness():
    rng = np.random.default_rng(404)
    out = {}
    vg, tg = grid(20, 15, 10.0)
    pts = rng.uniform(0, 200, size=(180, 2))
    td = Delaunay(pts).simplices.astype(np.int64)
    for name, (v, t) in {'grid': (vg, tg), 'rand': (pts, td)}.items():
        out[f'{name}_v'] = v
        out[f'{name}_t'] = t
        mult = rng.uniform(0.2, 2.0, size=t.shape[0]).astype(np.float32)
        out[f'{name}_mult'] = mult
        disp = np.stack((3 * np.sin(v[:, 1] / 40), 2 * np.cos(v[:, 0] / 55)), axis=-1) + 0.3 * rng.standard_normal(v.shape)
        out[f'{name}_vmov'] = v + disp
        for nu in (0.0, 0.3):
            mat = material.Material(**dict(material.MATERIAL_DEFAULT, poisson_ratio=nu))
            N = mat.sparse_engineering_shape_matrix(v[t], t, 2 * v.shape[0])
            r, c, d = coo(N)
            out[f'{name}_nu{nu}_N_r'] = r; out[f'{name}_nu{nu}_N_c'] = c; out[f'{name}_nu{nu}_N_d'] = d
            K = mat.engineering_stiffness_matrix_from_shape(N, multiplier=mult)
            r, c, d = coo(K)
            out[f'{name}_nu{nu}_K_r'] = r; out[f'{name}_nu{nu}_K_c'] = c; out[f'{name}_nu{nu}_K_d'] = d
            # G5: through Mesh.stiffness_matrix with a displaced MOVING gear
            m = Mesh(v.copy(), t.copy(), material_table=mat_table(nu), stiffness_multiplier=mult.copy(),
                     moving_vertices=v + disp, uid=7)
            Km, stress = m.stiffness_matrix(gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING))
            r, c, d = coo(Km)
            out[f'{name}_nu{nu}_Km_r'] = r; out[f'{name}_nu{nu}_Km_c'] = c; out[f'{name}_nu{nu}_Km_d'] = d
            out[f'{name}_nu{nu}_stress'] = stress
    np.savez_compressed(os.path.join(OUT, 'g45_stiffness.npz'), **out)


# ----------------------------------------------------------------------- G6-G9
def build_slm(rng, out=None):
    """3-mesh system: mesh0 locked, meshes 1 and 2 free, links 0-1, 1-2, 0-2."""
    v0, t0 = grid(12, 9, 10.0)
    v1, t1 = grid(10, 10, 11.0, origin=(3.0, -2.0))
    v2, t2 = grid(9, 12, 9.0, origin=(-4.0, 5.0))
    ms = [Mesh(v0, t0, uid=0, locked=True), Mesh(v1, t1, uid=1), Mesh(v2, t2, uid=2, soft_factor=0.5)]
    ms[1].apply_translation((1.5, -0.75), const.MESH_GEAR_FIXED)
    ms[2].apply_translation((-2.25, 1.0), const.MESH_GEAR_FIXED)
    links = []
    spec = []
    for (a, b, n) in ((0, 1, 60), (1, 2, 50), (0, 2, 40)):
        tid0 = rng.integers(0, ms[a].triangles.shape[0], size=n)
        tid1 = rng.integers(0, ms[b].triangles.shape[0], size=n)
        B0 = rng.dirichlet((1, 1, 1), size=n)
        B1 = rng.dirichlet((1, 1, 1), size=n)
        w = rng.uniform(0.3, 1.0, size=n).astype(np.float32)
        links.append(optimizer.Link(ms[a], ms[b], tid0, tid1, B0, B1, weight=w))
        spec.append((a, b, tid0, tid1, B0, B1, w))
    if out is not None:
        for k, m in enumerate(ms):
            out[f'm{k}_v'] = m.vertices(gear=const.MESH_GEAR_INITIAL)
            out[f'm{k}_t'] = m.triangles
            out[f'm{k}_off'] = m.offset(gear=const.MESH_GEAR_FIXED)
        for k, (a, b, tid0, tid1, B0, B1, w) in enumerate(spec):
            out[f'l{k}_ab'] = np.array([a, b])
            out[f'l{k}_tid0'] = tid0; out[f'l{k}_tid1'] = tid1
            out[f'l{k}_B0'] = B0; out[f'l{k}_B1'] = B1; out[f'l{k}_w'] = w
    return ms, links


def g6789_system():
    rng = np.random.default_rng(606)
    out = {}
    ms, links = build_slm(rng, out)
    slm = optimizer.SLM(ms, links=list(links), stiffness_lambda=1.0, crosslink_lambda=-1.0)
    S, _ = slm.crosslink_shape_matrix()
    r, c, d = coo(S); out['S_r'] = r; out['S_c'] = c; out['S_d'] = d
    K, stress = slm.stiffness_matrix(gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING))
    C, rhs = slm.crosslink_terms(start_gear=const.MESH_GEAR_MOVING, target_gear=const.MESH_GEAR_MOVING)
    r, c, d = coo(K); out['K_r'] = r; out['K_c'] = c; out['K_d'] = d
    r, c, d = coo(C); out['C_r'] = r; out['C_c'] = c; out['C_d'] = d
    out['stress'] = stress
    out['rhs'] = rhs
    ls, lc = slm.relative_lambda_trace(1.0, -1.0)
    out['lambdas'] = np.array([ls, lc], dtype=np.float64)
    A = ls * K + lc * C
    b = lc * rhs - ls * stress
    r, c, d = coo(A); out['A_r'] = r; out['A_c'] = c; out['A_d'] = d
    out['b'] = b
    # G7: solve() tight, deterministic
    x = optimizer.solve(A, b, 'minres', tol=1e-11, M='jacobi', tolerated_perturbation=None,
                        check_converge=True, chances=None, eval_step=10)
    out['x_solve'] = x
    from scipy.sparse.linalg import spsolve
    out['x_direct'] = spsolve(sparse.csc_matrix(0.5 * (A + A.T)), b)
    edc = np.ones(b.size, dtype=bool)
    edc[:3] = False
    edc[100:140:7] = False
    out['edc'] = edc
    out['x_edc'] = optimizer.solve(A, b, 'minres', tol=1e-11, M='jacobi', tolerated_perturbation=None,
                                   check_converge=True, chances=None, eval_step=10, extra_dof_constraint=edc)
    # G8: optimize_linear end-to-end
    cost = slm.optimize_linear(tol=1e-11, tolerated_perturbation=None,
                               callback_settings={'chances': None, 'eval_step': 10}, check_converge=True)
    out['cost'] = np.array(cost, dtype=np.float64)
    for k, m in enumerate(ms):
        out[f'm{k}_v_after'] = m.vertices(gear=const.MESH_GEAR_MOVING)
        out[f'm{k}_off_after'] = m.offset(gear=const.MESH_GEAR_MOVING)
    # G9: residue weights (after the solve)
    for k, lk in enumerate(links):
        out[f'l{k}_sample_err'] = np.asarray(lk.sample_err, dtype=np.float64)
        lk.set_huber_residue_filter(0.5)
        lk.adjust_weight_from_residue(gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_MOVING))
        out[f'l{k}_huber'] = lk._residue_weight.copy()
        lk.set_hard_residue_filter(0.8)
        lk.adjust_weight_from_residue(gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_MOVING))
        out[f'l{k}_thresh'] = np.asarray(lk._residue_weight).astype(np.float32)
        out[f'l{k}_dxy_after'] = lk.dxy(gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_MOVING), use_mask=False)
    np.savez_compressed(os.path.join(OUT, 'g6789_system.npz'), **out)


# ----------------------------------------------------------------------- G10
def g10_elements():
    rng = np.random.default_rng(1010)
    out = {}
    p = rng.uniform(0, 30, size=(50, 3, 2))
    e = p[:, 1] - p[:, 0]; f = p[:, 2] - p[:, 1]
    flip = (e[:, 0] * f[:, 1] - e[:, 1] * f[:, 0]) < 0
    p[flip] = p[flip][:, ::-1]
    uv = 0.1 * rng.standard_normal((50, 6)) * np.sqrt(np.abs(e[:, :1] * f[:, 1:] - e[:, 1:] * f[:, :1]))
    out['tripts'] = p
    out['uv'] = uv
    for model, tag in ((const.MATERIAL_MODEL_ENG, 'eng'), (const.MATERIAL_MODEL_SVK, 'svk'), (const.MATERIAL_MODEL_NHK, 'nhk')):
        for nu in ((0.0, 0.3) if model != const.MATERIAL_MODEL_NHK else (0.0,)):
            mat = material.Material(type=model, poisson_ratio=nu, name=f'g10_{tag}_{nu}', uid=50 + model)
            Ms = list(mat.shape_matrix_from_vertices(p))
            if model == const.MATERIAL_MODEL_ENG and nu == 0.0:
                out['B'] = Ms[0]; out['areas'] = Ms[1]
            K, P, mm = mat.element_stiffness_matrices_from_shape_matrices(Ms, uv=uv)
            out[f'{tag}_nu{nu}_K'] = K
            out[f'{tag}_nu{nu}_P'] = P
    np.savez_compressed(os.path.join(OUT, 'g10_elements.npz'), **out)


# ----------------------------------------------------------------------- G11
def g11_bbox():
    out = {}
    cases = [((0, 0, 510, 4096), 1024, 1, 1), ((0, 0, 510, 4096), 75, 2, 1), ((-3.5, 10.5, 500.5, 3010.5), 75, 2, 1),
             ((0, 0, 4000, 400), 74.3, 1, 0.7), ((0, 0, 260, 400), None, (3, 2), 1), ((5, 7, 133, 80), 25, 2, 1)]
    for k, (bb, bs, mnb, sf) in enumerate(cases):
        kw = dict(min_num_blocks=mnb, shrink_factor=sf)
        if bs is not None:
            kw['block_size'] = bs
        res = common.divide_bbox(bb, **kw)
        out[f'div{k}_in'] = np.array(list(bb) + [bs if bs is not None else -1, sf] + list(np.atleast_1d(mnb)), dtype=np.float64)
        out[f'div{k}_out'] = np.stack(res, axis=-1).astype(np.float64)
    rng = np.random.default_rng(1111)
    ij = rng.integers(0, 37, size=(200, 2)).astype(np.float64)
    out['z_in'] = ij
    out['z_out'] = common.z_order(ij)
    bbs = np.array([[0, 0, 75, 73], [10, -4, 85, 70], [3, 3, 4, 9]], dtype=np.float64)
    out['bb_in'] = bbs
    out['bb_centers'] = common.bbox_centers(bbs)
    out['bb_sizes'] = common.bbox_sizes(bbs)
    # distributor_cartesian_bbox arithmetic on plain bboxes (mesh.bbox is just min/max of vertices)
    class _M:
        def __init__(self, bb): self._bb = bb
        def bbox(self, gear=None): return self._bb
    for k, (b0, b1, sp, mnb) in enumerate([((-0.5 + 6, -0.5 - 4, 509.5 + 6, 4095.5 - 4), (-0.5, -0.5, 509.5, 4095.5), 1024, 1),
                                           ((-0.5 + 6, -0.5 - 4, 509.5 + 6, 4095.5 - 4), (-0.5, -0.5, 509.5, 4095.5), 75.0, 2)]):
        r0, r1 = matcher.distributor_cartesian_bbox(_M(b0), _M(b1), sp, min_num_blocks=mnb, zorder=True)
        out[f'dist{k}_in'] = np.array(list(b0) + list(b1) + [sp, mnb], dtype=np.float64)
        out[f'dist{k}_bb0'] = r0.astype(np.float64)
        out[f'dist{k}_bb1'] = r1.astype(np.float64)
    from scipy.fftpack import next_fast_len
    out['nfl'] = np.array([next_fast_len(n) for n in range(1, 4200)], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, 'g11_bbox.npz'), **out)


# ----------------------------------------------------------------------- G12
def g12_mixed_materials():
    """Mesh.stiffness_matrix on a mesh with linear ENG, Neo-Hookean and St-Venant-Kirchhoff regions
    (mesh.py:2992-3083): tangent stiffness + internal force at a displaced MOVING gear."""
    rng = np.random.default_rng(1212)
    v, t = grid(14, 10, 10.0)
    mids = np.zeros(t.shape[0], dtype=np.int16)
    ctr = v[t].mean(axis=1)
    mids[ctr[:, 0] > 90] = 5
    mids[(ctr[:, 0] <= 90) & (ctr[:, 1] > 60)] = 6
    tab = {'default': dict(material.MATERIAL_DEFAULT),
           'nhk': {'type': const.MATERIAL_MODEL_NHK, 'uid': 5, 'stiffness_multiplier': 0.7},
           'svk': {'type': const.MATERIAL_MODEL_SVK, 'uid': 6, 'poisson_ratio': 0.25, 'stiffness_multiplier': 1.3}}
    mt = material.MaterialTable(table=tab)
    disp = np.stack((2 * np.sin(v[:, 1] / 30), 1.5 * np.cos(v[:, 0] / 40)), -1)
    mult = rng.uniform(0.5, 1.5, t.shape[0]).astype(np.float32)
    m = Mesh(v.copy(), t.copy(), material_table=mt, material_ids=mids.copy(), stiffness_multiplier=mult.copy(),
             moving_vertices=v + disp, uid=3)
    K, stress = m.stiffness_matrix(gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING))
    out = {'v': v, 'vmov': v + disp, 't': m.triangles, 'mult': m._stiffness_multiplier, 'mids': m._material_ids.astype(np.int32)}
    # per-triangle material description in the (material-sorted) triangle order of the Mesh
    model = np.zeros(m.triangles.shape[0], dtype=np.int32); nu = np.zeros(m.triangles.shape[0]); mm = np.ones(m.triangles.shape[0])
    for name, uid in (('nhk', 5), ('svk', 6)):
        sel = m._material_ids == uid
        model[sel] = mt[name]._type; nu[sel] = mt[name]._poisson_ratio; mm[sel] = mt[name]._stiffness_multiplier
    out['model'] = model; out['nu'] = nu; out['matmult'] = mm
    r, c, d = coo(K)
    out['K_r'] = r; out['K_c'] = c; out['K_d'] = d
    out['stress'] = stress
    np.savez_compressed(os.path.join(OUT, 'g12_mixed_materials.npz'), **out)


# ----------------------------------------------------------------------- G13
def g13_strain():
    """spatial.fit_affine (spatial.py:21-73) and the matcher's strain estimate (matcher.py:752-777): rigid
    initialisation by optimize_affine_cascade, anneal, optimize_linear, sqrt(Es / Es0)."""
    from feabas import spatial
    rng = np.random.default_rng(1313)
    out = {}
    # fit_affine: generic weighted, collinear (rank 2), reflected (avoid_flip) point sets
    for k in range(4):
        n = 40
        p1 = rng.uniform(-200, 200, size=(n, 2))
        th = rng.uniform(-0.2, 0.2)
        M = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]]) @ np.diag(rng.uniform(0.9, 1.1, 2))
        if k == 2:
            p1[:, 1] = 0.5 * p1[:, 0] + 3.0                       # collinear
        if k == 3:
            M = M @ np.diag([1.0, -1.0])                          # reflection
        p0 = p1 @ M + rng.uniform(-30, 30, size=(1, 2)) + rng.normal(0, 0.5, size=(n, 2))
        w = rng.uniform(0.3, 1.0, size=n).astype(np.float32)
        A, R = spatial.fit_affine(p0, p1, return_rigid=True, weight=w, svd_clip=(1, 1), avoid_flip=True)
        out[f'fa{k}_p0'] = p0; out[f'fa{k}_p1'] = p1; out[f'fa{k}_w'] = w; out[f'fa{k}_A'] = A; out[f'fa{k}_R'] = R
    # strain chain on a cartesian pair
    nx, ny, h = 15, 5, 20.0
    xs = h * np.arange(nx) - 0.5; ys = h * np.arange(ny) - 0.5
    vx, vy = np.meshgrid(xs, ys)
    v = np.stack((vx.ravel(), vy.ravel()), axis=-1)
    idx = np.arange(nx * ny).reshape(ny, nx)
    a = idx[:-1, :-1].ravel(); b = idx[:-1, 1:].ravel(); c = idx[1:, :-1].ravel(); d = idx[1:, 1:].ravel()
    tri = np.stack((np.stack((a, b, d), -1), np.stack((a, d, c), -1)), axis=1).reshape(-1, 3)
    m0 = Mesh(v, tri, uid=0)
    m0.apply_translation((3.0, -2.0), const.MESH_GEAR_FIXED)
    m0.lock()
    m1 = Mesh(v, tri, uid=1)
    n = 60
    tid1 = rng.integers(0, tri.shape[0], size=n)
    B1 = rng.dirichlet((2, 2, 2), size=n)
    xy1 = m1.bary2cart(tid1, B1, const.MESH_GEAR_INITIAL, offsetting=True)
    th = 0.01
    xy0_fixed = (xy1 - xy1.mean(0)) @ np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]]) + xy1.mean(0) + np.array([3.4, -1.7]) \
        + 0.8 * np.stack((np.sin(xy1[:, 0] / 60.0), np.cos(xy1[:, 1] / 25.0)), -1)
    # mesh0 points: barycentrics in mesh0 of (xy0_fixed - translation), clipped into the mesh by construction of the field
    q0 = xy0_fixed - np.array([3.0, -2.0])
    i = np.clip(np.searchsorted(xs, q0[:, 0], side='right') - 1, 0, nx - 2)
    j = np.clip(np.searchsorted(ys, q0[:, 1], side='right') - 1, 0, ny - 2)
    uu = (q0[:, 0] - xs[i]) / h; ww = (q0[:, 1] - ys[j]) / h
    tid0 = 2 * (j * (nx - 1) + i) + (ww > uu)
    _, B0 = m0.cart2bary(xy0_fixed, const.MESH_GEAR_FIXED, tid=tid0)
    w = rng.uniform(0.35, 1.0, size=n).astype(np.float32)
    link = optimizer.Link(m0, m1, tid0, tid1, B0, B1, weight=w)
    opt = optimizer.SLM([m0, m1], stiffness_lambda=1.0, assert_dominance=False)
    opt.add_link(link)
    opt.optimize_affine_cascade(start_gear=const.MESH_GEAR_INITIAL, target_gear=const.MESH_GEAR_FIXED, svd_clip=(1, 1))
    out['st_v_fixed'] = m1.vertices(gear=const.MESH_GEAR_FIXED).copy()
    out['st_off_fixed'] = m1.offset(gear=const.MESH_GEAR_FIXED).copy()
    opt.anneal(gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING), mode=const.ANNEAL_COPY_EXACT)
    opt.optimize_linear(tol=1e-11, tolerated_perturbation=None, callback_settings={'chances': None, 'eval_step': 10}, check_converge=True)
    v0 = m1.vertices(gear=const.MESH_GEAR_FIXED)
    v1 = m1.vertices(gear=const.MESH_GEAR_MOVING)
    dv = v1 - v0
    v0 = v0 - np.mean(v0, axis=0, keepdims=True)
    dv = dv - np.mean(dv, axis=0, keepdims=True)
    St, _ = m1.stiffness_matrix()
    Es = max(0, St.dot(dv.ravel()).dot(dv.ravel()))
    Es0 = max(0, St.dot(v0.ravel()).dot(v0.ravel()))
    out.update(st_v=v, st_tri=tri, st_t0=np.array([3.0, -2.0]), st_tid0=tid0, st_B0=B0, st_tid1=tid1, st_B1=B1, st_w=w,
               st_Es=np.float64(Es), st_Es0=np.float64(Es0), st_strain=np.float64((Es / Es0) ** 0.5), st_v_moving=v1.copy())
    np.savez_compressed(os.path.join(OUT, 'g13_strain.npz'), **out)


# ----------------------------------------------------------------------- G14
def g14_groupings():
    """SLM.optimize_linear(groupings=...) (optimizer.py:1378-1415): meshes 1 and 2 form one group (shared DoFs),
    mesh 0 is locked, mesh 3 is on its own."""
    rng = np.random.default_rng(1414)
    out = {}
    v0, t0 = grid(10, 8, 10.0)
    va, ta = grid(9, 9, 11.0, origin=(2.0, -3.0))
    v3, t3 = grid(8, 10, 9.0, origin=(-5.0, 4.0))
    ms = [Mesh(v0, t0, uid=0, locked=True), Mesh(va, ta, uid=1), Mesh(va + np.array([0.4, -0.2]), ta, uid=2, soft_factor=0.7), Mesh(v3, t3, uid=3)]
    ms[1].apply_translation((1.0, 0.5), const.MESH_GEAR_FIXED)
    ms[3].apply_translation((-1.5, 2.0), const.MESH_GEAR_FIXED)
    links = []
    for k, (a, b, n) in enumerate(((0, 1, 40), (1, 3, 35), (2, 3, 30), (0, 2, 25))):
        tid0 = rng.integers(0, ms[a].triangles.shape[0], size=n)
        tid1 = rng.integers(0, ms[b].triangles.shape[0], size=n)
        B0 = rng.dirichlet((1, 1, 1), size=n); B1 = rng.dirichlet((1, 1, 1), size=n)
        w = rng.uniform(0.3, 1.0, size=n).astype(np.float32)
        links.append(optimizer.Link(ms[a], ms[b], tid0, tid1, B0, B1, weight=w))
        out[f'l{k}_ab'] = np.array([a, b]); out[f'l{k}_tid0'] = tid0; out[f'l{k}_tid1'] = tid1
        out[f'l{k}_B0'] = B0; out[f'l{k}_B1'] = B1; out[f'l{k}_w'] = w
    for k, m in enumerate(ms):
        out[f'm{k}_v'] = m.vertices(gear=const.MESH_GEAR_INITIAL); out[f'm{k}_t'] = m.triangles
        out[f'm{k}_off'] = m.offset(gear=const.MESH_GEAR_FIXED)
    groupings = np.array([0, 1, 1, 2])
    slm = optimizer.SLM(ms, links=links, stiffness_lambda=1.0, crosslink_lambda=-1.0)
    cost = slm.optimize_linear(tol=1e-11, groupings=groupings, tolerated_perturbation=None,
                               callback_settings={'chances': None, 'eval_step': 10}, check_converge=True)
    out['groupings'] = groupings
    out['cost'] = np.array(cost, dtype=np.float64)
    for k, m in enumerate(ms):
        out[f'm{k}_v_after'] = m.vertices(gear=const.MESH_GEAR_MOVING); out[f'm{k}_off_after'] = m.offset(gear=const.MESH_GEAR_MOVING)
    np.savez_compressed(os.path.join(OUT, 'g14_groupings.npz'), **out)


# ----------------------------------------------------------------------- G15
def g15_translation():
    """SLM.optimize_translation_lsqr / optimize_translation_w_filtering (optimizer.py:974-1125): a 2 x 3 arrangement of
    tiles with known offsets, one locked, one link carrying a gross error."""
    rng = np.random.default_rng(1515)
    out = {}
    v, t = grid(6, 5, 20.0)
    true_t = np.array([[0, 0], [3.0, -2.0], [-4.0, 1.5], [2.5, 2.5], [-1.0, -3.0], [5.0, 0.5]])
    ms = [Mesh(v, t, uid=k) for k in range(6)]
    ms[0].lock()
    pairs = [(0, 1), (1, 2), (0, 3), (3, 4), (4, 5), (1, 4), (2, 5)]
    links = []
    for k, (a, b) in enumerate(pairs):
        n = 12
        tid0 = rng.integers(0, t.shape[0], size=n); tid1 = rng.integers(0, t.shape[0], size=n)
        B0 = rng.dirichlet((2, 2, 2), size=n); B1 = rng.dirichlet((2, 2, 2), size=n)
        # consistent with the true offsets: p0 + t_a == p1 + t_b  (+ noise); build xy1 from xy0
        xy0 = ms[a].bary2cart(tid0, B0, const.MESH_GEAR_FIXED, offsetting=True)
        xy1 = xy0 + true_t[a] - true_t[b] + rng.normal(0, 0.05, size=(n, 2))
        if k == 5:
            xy1 = xy1 + np.array([9.0, -7.0])                       # a bad link
        # barycentrics of xy1 in mesh b (may fall outside the triangle: barycentric extrapolation is fine for this test)
        _, B1 = ms[b].cart2bary(xy1, const.MESH_GEAR_FIXED, tid=tid1)
        w = rng.uniform(0.4, 1.0, size=n).astype(np.float32)
        links.append(optimizer.Link(ms[a], ms[b], tid0, tid1, B0, B1, weight=w))
        out[f'l{k}_ab'] = np.array([a, b]); out[f'l{k}_tid0'] = tid0; out[f'l{k}_tid1'] = tid1
        out[f'l{k}_B0'] = B0; out[f'l{k}_B1'] = B1; out[f'l{k}_w'] = w
    out['v'] = v; out['t'] = t
    slm = optimizer.SLM(ms, links=links)
    cost, residue = slm.optimize_translation_lsqr(tol=1e-12)
    out['lsqr_cost'] = np.array(cost); out['lsqr_residue'] = residue
    out['lsqr_offsets'] = np.stack([m.offset(gear=const.MESH_GEAR_FIXED).ravel() for m in ms])
    # fresh system for the filtered variant
    ms2 = [Mesh(v, t, uid=k) for k in range(6)]
    ms2[0].lock()
    links2 = [optimizer.Link(ms2[a], ms2[b], out[f'l{k}_tid0'], out[f'l{k}_tid1'], out[f'l{k}_B0'], out[f'l{k}_B1'], weight=out[f'l{k}_w'])
              for k, (a, b) in enumerate(pairs)]
    slm2 = optimizer.SLM(ms2, links=links2)
    nd, cost2 = slm2.optimize_translation_w_filtering(tol=1e-12, residue_threshold=1.0)
    out['filt_disabled'] = np.array(nd); out['filt_cost'] = np.array(cost2)
    out['filt_offsets'] = np.stack([m.offset(gear=const.MESH_GEAR_FIXED).ravel() for m in ms2])
    out['filt_link_disabled'] = np.array([lk._disabled for lk in links2])
    np.savez_compressed(os.path.join(OUT, 'g15_translation.npz'), **out)


# ----------------------------------------------------------------------- G16
def g16_relax():
    """optimizer.relax_mesh / relax_mesh_most_deformed (optimizer.py:2110-2190), Mesh.stiffness_matrix_local_normalized
    (mesh.py:3086-3129), the rigid/affine anneal modes (mesh.py:2421-2456) and the per-triangle deformation measures
    (mesh.py:1966-1986, 3358-3365)."""
    from feabas import config
    rng = np.random.default_rng(1616)
    out = {}
    gear = (const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)
    tight = dict(tol=1e-11, tolerated_perturbation=None, callback_settings={'chances': None, 'eval_step': 10})
    v, t = grid(16, 12, 10.0)
    mult = rng.uniform(0.5, 2.0, size=t.shape[0]).astype(np.float32)
    mult[5] = 1e-5                                      # below max/1000: exercises the multiplier clip
    disp = np.stack((2 * np.sin(v[:, 1] / 40), 1.5 * np.cos(v[:, 0] / 55)), axis=-1)
    hot = np.array([16 * 5 + 7, 16 * 5 + 8, 16 * 6 + 7])
    disp[hot] += np.array([[14.0, 3.0], [-6.0, 9.0], [4.0, -8.0]])      # three flipped triangles
    out['v'] = v; out['t'] = t; out['mult'] = mult; out['vmov'] = v + disp
    out['moff'] = np.array([[3.0, -1.5]])

    def fresh():
        return Mesh(v.copy(), t.copy(), stiffness_multiplier=mult.copy(), moving_vertices=v + disp,
                    moving_offset=out['moff'].copy(), uid=3)

    m = fresh()
    out['area_deform'] = m.triangle_area_deform(gear=gear)
    out['edge_deform'] = m.triangle_edge_deform(gear=gear)
    out['eff_mult'] = m.effective_stiffness_multiplier(gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_MOVING))
    out['svd_deform_area'] = Mesh.svds_to_deform(out['area_deform'].reshape(-1, 1))
    tm = np.zeros(t.shape[0], dtype=bool)
    tm[rng.choice(t.shape[0], 60, replace=False)] = True
    tm[5] = True
    out['tmask'] = tm
    K, stress = m.stiffness_matrix_local_normalized(gear=gear, tri_mask=tm)
    r, c, d = coo(K)
    out['Kn_r'] = r; out['Kn_c'] = c; out['Kn_d'] = d; out['Kn_stress'] = stress
    # relax_mesh with a block of free triangles, converged
    ft = np.zeros(t.shape[0], dtype=bool)
    ctr = v[t].mean(axis=1)
    ft[(np.abs(ctr[:, 0] - 75) < 38) & (np.abs(ctr[:, 1] - 55) < 32)] = True
    out['free_tri'] = ft
    m = fresh()
    out['ft_modified'] = np.array(optimizer.relax_mesh(m, free_triangles=ft, gear=gear, **tight))
    out['ft_vmov'] = m.vertices(gear=gear[1]); out['ft_moff'] = m.offset(gear=gear[1])
    out['ft_vfix'] = m.vertices(gear=gear[0]); out['ft_foff'] = m.offset(gear=gear[0])
    # ... and with a list of free vertices
    fv = np.unique(t[np.isin(t, hot).any(axis=1)])
    out['free_vtx'] = fv
    m = fresh()
    out['fv_modified'] = np.array(optimizer.relax_mesh(m, free_vertices=fv, gear=gear, **tight))
    out['fv_vmov'] = m.vertices(gear=gear[1]); out['fv_moff'] = m.offset(gear=gear[1])
    # relax_mesh_most_deformed: which region it frees (the arguments it hands to relax_mesh) and its converged result
    captured = {}
    orig = optimizer.relax_mesh

    def spy(M, free_vertices=None, free_triangles=None, **kw):
        captured['fv'] = None if free_vertices is None else np.array(free_vertices)
        captured['ft'] = None if free_triangles is None else np.array(free_triangles)
        # relax_mesh_most_deformed has no way to pass solver settings on: with the defaults (tol 1e-7, the random-perturbation
        # exit of optimizer.solve) the end state is one of many; the capture runs the same call converged and deterministic
        return orig(M, free_vertices=free_vertices, free_triangles=free_triangles, **{**kw, **tight})

    optimizer.relax_mesh = spy
    try:
        m = fresh()
        out['md_flip_modified'] = np.array(optimizer.relax_mesh_most_deformed(m, gear=gear, deform_cutoff=-1))
        out['md_flip_free_vtx'] = captured['fv']
        out['md_flip_vmov'] = m.vertices(gear=gear[1]); out['md_flip_moff'] = m.offset(gear=gear[1])
        out['md_flip_area_deform'] = m.triangle_area_deform(gear=gear)
        for name, iqr in (('md_cut', 0), ('md_iqr', 1.5)):
            m = fresh()
            captured.clear()
            out[f'{name}_modified'] = np.array(optimizer.relax_mesh_most_deformed(m, gear=gear, deform_cutoff=config.MAXIMUM_DEFORM_ALLOWED, iqr=iqr))
            out[f'{name}_free_tri'] = captured['ft']
            out[f'{name}_vmov'] = m.vertices(gear=gear[1]); out[f'{name}_moff'] = m.offset(gear=gear[1])
        out['deform_cutoff'] = np.array(config.MAXIMUM_DEFORM_ALLOWED)
    finally:
        optimizer.relax_mesh = orig
    # anneal modes on a mesh with two connected components
    v2, t2 = grid(6, 5, 10.0, origin=(100.0, 20.0))
    vv = np.concatenate((v[:16 * 4], v2), axis=0)
    t_a = t[np.all(t < 16 * 4, axis=1)]
    tt = np.concatenate((t_a, t2 + 16 * 4), axis=0)
    th0, th1 = 0.05, -0.08
    R0 = np.array([[np.cos(th0), np.sin(th0)], [-np.sin(th0), np.cos(th0)]])
    R1 = np.array([[np.cos(th1), np.sin(th1)], [-np.sin(th1), np.cos(th1)]])
    vm = vv.copy()
    vm[:16 * 4] = vv[:16 * 4] @ R0 * 1.03 + np.array([4.0, -2.0])
    vm[16 * 4:] = vv[16 * 4:] @ R1 * 0.97 + np.array([-3.0, 5.0])
    vm += 0.2 * rng.standard_normal(vm.shape)
    out['an_v'] = vv; out['an_t'] = tt; out['an_vmov'] = vm
    for name, mode in (('grigid', const.ANNEAL_GLOBAL_RIGID), ('gaffine', const.ANNEAL_GLOBAL_AFFINE),
                       ('crigid', const.ANNEAL_CONNECTED_RIGID), ('caffine', const.ANNEAL_CONNECTED_AFFINE)):
        m = Mesh(vv.copy(), tt.copy(), moving_vertices=vm.copy(), moving_offset=np.array([[1.0, 2.0]]), uid=4)
        m.anneal(gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_FIXED), mode=mode)
        out[f'an_{name}_vfix'] = m.vertices(gear=const.MESH_GEAR_FIXED); out[f'an_{name}_foff'] = m.offset(gear=const.MESH_GEAR_FIXED)
    np.savez_compressed(os.path.join(OUT, 'g16_relax.npz'), **out)


# ----------------------------------------------------------------------- G17
def g17_newton():
    """SLM.optimize_Newton_Raphson / optimize_elastic (optimizer.py:1440-1555) on the mixed-material mesh of G12 (engineering +
    Neo-Hookean + St-Venant-Kirchhoff regions) pulled by links towards a displaced locked twin.  Deterministic settings: no
    random-perturbation exit, no `chances` exit, so every inner solve runs to its tolerance and the result is the fixed point
    of the iteration up to that tolerance."""
    rng = np.random.default_rng(1717)
    v, t = grid(14, 10, 10.0)
    mids = np.zeros(t.shape[0], dtype=np.int16)
    ctr = v[t].mean(axis=1)
    mids[ctr[:, 0] > 90] = 5
    mids[(ctr[:, 0] <= 90) & (ctr[:, 1] > 60)] = 6
    tab = {'default': dict(material.MATERIAL_DEFAULT),
           'nhk': {'type': const.MATERIAL_MODEL_NHK, 'uid': 5, 'stiffness_multiplier': 0.7},
           'svk': {'type': const.MATERIAL_MODEL_SVK, 'uid': 6, 'poisson_ratio': 0.25, 'stiffness_multiplier': 1.3}}
    mult = rng.uniform(0.5, 1.5, t.shape[0]).astype(np.float32)
    disp = 0.6 * np.stack((2 * np.sin(v[:, 1] / 30), 1.5 * np.cos(v[:, 0] / 40)), -1) + np.array([[0.4, -0.3]])
    n = 300
    out = {}

    def build():
        mt = material.MaterialTable(table=tab)
        m0 = Mesh(v + disp, t.copy(), uid=0, locked=True)
        m1 = Mesh(v.copy(), t.copy(), material_table=mt, material_ids=mids.copy(), stiffness_multiplier=mult.copy(), uid=1)
        return mt, m0, m1
    mt, m0, m1 = build()
    # the Mesh sorts its triangles by material: matches are drawn in that order so that tid means the same to everybody
    tt = m1.triangles
    tid = rng.integers(0, tt.shape[0], n); B = rng.dirichlet((1, 1, 1), n)
    w = rng.uniform(0.4, 1.0, n).astype(np.float32)
    model = np.zeros(tt.shape[0], dtype=np.int32); nu = np.zeros(tt.shape[0]); mm = np.ones(tt.shape[0])
    for name, uid in (('nhk', 5), ('svk', 6)):
        sel = m1._material_ids == uid
        model[sel] = mt[name]._type; nu[sel] = mt[name]._poisson_ratio; mm[sel] = mt[name]._stiffness_multiplier
    out.update(v=v, disp=disp, t0=m0.triangles, t1=tt, mult=m1._stiffness_multiplier, model=model, nu=nu, matmult=mm,
               tid=tid, B=B, w=w)
    # the locked twin keeps the input triangle order; a match (tid, B) on it is expressed through the free mesh's triangle:
    # same three vertices in the same order, so the locked side uses the triangle LIST of the free mesh
    det = dict(tolerated_perturbation=None, callback_settings={'chances': None, 'eval_step': 10})
    for case, call in (('nr', lambda slm: slm.optimize_Newton_Raphson(max_newtonstep=8, tol=1e-9, **det)),
                       ('nr3', lambda slm: slm.optimize_Newton_Raphson(max_newtonstep=3, tol=1e-6, **det)),
                       ('elastic', lambda slm: slm.optimize_elastic(max_newtonstep=6, tol=1e-8, **det)),
                       ('huber', lambda slm: slm.optimize_Newton_Raphson(max_newtonstep=6, tol=1e-8, residue_mode='huber', residue_len=0.2, **det))):
        mt, m0, m1 = build()
        m0 = Mesh(v + disp, tt.copy(), uid=0, locked=True)
        lk = optimizer.Link(m0, m1, tid, tid, B, B, weight=w)
        slm = optimizer.SLM([m0, m1], links=[lk], stiffness_lambda=1.0, crosslink_lambda=1.0)
        c0, c1 = call(slm)
        out[f'{case}_cost'] = np.array([c0, c1], dtype=np.float64)
        out[f'{case}_v_after'] = m1.vertices(gear=const.MESH_GEAR_MOVING)
        out[f'{case}_off_after'] = m1.offset(gear=const.MESH_GEAR_MOVING)
        out[f'{case}_residue_weight'] = np.asarray(lk._residue_weight, dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, 'g17_newton.npz'), **out)


# ----------------------------------------------------------------------- G18
def g18_locked_neighbours():
    """SLM.optimize_linear of ONE free section between two LOCKED neighbours (the independent-unit mode of the aligner's
    sliding window, aligner.py:696-727 with one free section): the unit of BASELINE.json config[4]."""
    rng = np.random.default_rng(1818)
    v, t = grid(40, 36, 12.0)
    L = 12.0 * 39

    def field(g):
        ph = 1.1 * g
        return np.stack((3 * np.sin(2 * np.pi * v[:, 1] / L + ph) + np.cos(4 * np.pi * v[:, 0] / L - ph),
                         2.5 * np.cos(2 * np.pi * v[:, 0] / L - ph) + np.sin(4 * np.pi * v[:, 1] / L + ph)), axis=-1)
    prev = Mesh(v + (field(0) - field(1)), t.copy(), uid=0, locked=True)
    cur = Mesh(v.copy(), t.copy(), uid=1)
    nxt = Mesh(v + (field(2) - field(1)), t.copy(), uid=2, locked=True)
    out = dict(v=v, t=t, v_prev=prev.vertices(gear=const.MESH_GEAR_INITIAL), v_next=nxt.vertices(gear=const.MESH_GEAR_INITIAL))
    links = []
    for k, (a, b) in enumerate(((prev, cur), (cur, nxt))):
        n = 900
        tid = np.sort(rng.integers(0, t.shape[0], n)); B = rng.dirichlet((1, 1, 1), n)
        w = rng.uniform(0.3, 1.0, n).astype(np.float32)
        links.append(optimizer.Link(a, b, tid, tid, B, B, weight=w))
        out[f'l{k}_tid'] = tid; out[f'l{k}_B'] = B; out[f'l{k}_w'] = w
    slm = optimizer.SLM([prev, cur, nxt], links=links, stiffness_lambda=1.0, crosslink_lambda=-1.0)
    cost = slm.optimize_linear(tol=1e-11, tolerated_perturbation=None, callback_settings={'chances': None, 'eval_step': 10}, check_converge=True)
    out['cost'] = np.array(cost, dtype=np.float64)
    out['v_after'] = cur.vertices(gear=const.MESH_GEAR_MOVING)
    out['off_after'] = cur.offset(gear=const.MESH_GEAR_MOVING)
    np.savez_compressed(os.path.join(OUT, 'g18_locked_neighbours.npz'), **out)


# ----------------------------------------------------------------------- G19
def g19_area_stretch():
    """Materials whose stiffness follows the area stretch of the triangle (stiffness_func): Mesh.stiffness_matrix through
    nonlinear_engineering_stiffness_matrix (mesh.py:2937-2971; the default "wrinkle" material of
    configs/default_material_table.yaml:46-56, material.asymmetrical_elasticity, material.py:546-551) and the f(J) factor of
    St-Venant-Kirchhoff / Neo-Hookean elements (material.py:242, 262, 307-308), on a mesh that is compressed in one half and
    stretched in the other; plus SLM.optimize_Newton_Raphson on it (a mesh with such a material is non-linear: K is
    re-assembled every step)."""
    rng = np.random.default_rng(1919)
    v, t = grid(14, 10, 10.0)
    ctr = v[t].mean(axis=1)
    wr = dict(strain=[0.0, 0.75, 1.0, 1.01], stiffness=[1.5, 1.0, 0.5, 1.0e-7])          # default_material_table.yaml:54-56
    tab_full = {'default': dict(material.MATERIAL_DEFAULT),
                'wrinkle': {'type': const.MATERIAL_MODEL_ENG, 'uid': 7, 'stiffness_multiplier': 0.4, 'poisson_ratio': 0.0,
                            'stiffness_func_factory': 'feabas.material.asymmetrical_elasticity', 'stiffness_func_params': wr},
                'fold': {'type': const.MATERIAL_MODEL_ENG, 'uid': 8, 'stiffness_multiplier': 0.9, 'poisson_ratio': 0.3,
                         'stiffness_func_factory': 'feabas.material.asymmetrical_elasticity',
                         'stiffness_func_params': dict(strain=[0.2, 0.9, 1.0, 1.3], stiffness=[3.0, 1.2, 1.0, 0.25])},
                'svkf': {'type': const.MATERIAL_MODEL_SVK, 'uid': 9, 'poisson_ratio': 0.25, 'stiffness_multiplier': 1.3,
                         'stiffness_func_factory': 'feabas.material.asymmetrical_elasticity',
                         'stiffness_func_params': dict(strain=[0.0, 0.5, 1.0, 1.2], stiffness=[2.0, 1.0, 0.8, 0.1])},
                'nhkf': {'type': const.MATERIAL_MODEL_NHK, 'uid': 10, 'stiffness_multiplier': 0.7,
                         'stiffness_func_factory': 'feabas.material.asymmetrical_elasticity', 'stiffness_func_params': wr}}
    # a smooth map: compressed on the left (area ratio down to ~0.7), stretched on the right (up to ~1.15), sheared a little
    L = 130.0
    sx = 1.0 + 0.16 * np.sin(np.pi * (v[:, 0] / L - 0.5)); sy = 1.0 - 0.06 * np.cos(2 * np.pi * v[:, 1] / 90.0)
    vmov = np.stack((np.cumsum(np.ones(1)) * 0 + (v[:, 0] - 65) * sx + 65 + 0.8 * np.sin(v[:, 1] / 25), (v[:, 1] - 45) * sy + 45 + 0.5 * np.cos(v[:, 0] / 35)), -1)
    mult = rng.uniform(0.5, 1.5, t.shape[0]).astype(np.float32)
    out = {'v': v, 'vmov': vmov, 'mult_in': mult}

    def describe(m, mt, names):
        """per-triangle material description in the (material-sorted) triangle order of the Mesh"""
        T = m.triangles.shape[0]
        model = np.zeros(T, dtype=np.int32); nu = np.zeros(T); mm = np.ones(T); func = np.full(T, -1, dtype=np.int32)
        tabs = []
        for name in names:
            mat = mt[name]
            sel = m._material_ids == mat.uid
            model[sel] = mat._type; nu[sel] = mat._poisson_ratio; mm[sel] = mat._stiffness_multiplier
            if mat._stiffness_func is not None:
                func[sel] = len(tabs)
                tabs.append((np.asarray(mat._stiffness_func_params['strain'], dtype=np.float64), np.asarray(mat._stiffness_func_params['stiffness'], dtype=np.float64)))
        return model, nu, mm, func, tabs

    cases = {'wr': ({'default': 0, 'wrinkle': 7}, lambda c: np.where((c[:, 0] > 30) & (c[:, 0] < 100) & (c[:, 1] > 20), 7, 0)),
             'all': ({'wrinkle': 7, 'fold': 8}, lambda c: np.where(c[:, 1] > 45, 8, 7)),                   # no linear triangle: baseline over all
             'mix': ({'default': 0, 'wrinkle': 7, 'fold': 8, 'svkf': 9, 'nhkf': 10},
                     lambda c: np.select([c[:, 0] < 30, c[:, 0] < 60, (c[:, 0] < 95) & (c[:, 1] > 50), c[:, 0] >= 95], [7, 9, 8, 10], 0))}
    for case, (names, region) in cases.items():
        tab = {k: dict(tab_full[k]) for k in names} if 'default' in names else {**{k: dict(tab_full[k]) for k in names}, 'default': dict(tab_full['default'])}
        mt = material.MaterialTable(table=tab)
        mids = region(ctr).astype(np.int16)
        m = Mesh(v.copy(), t.copy(), material_table=mt, material_ids=mids.copy(), stiffness_multiplier=mult.copy(), moving_vertices=vmov.copy(), uid=3)
        assert not m.is_linear
        K, stress = m.stiffness_matrix(gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING))
        model, nu, mm, func, tabs = describe(m, mt, [k for k in names])
        out[f'{case}_t'] = m.triangles; out[f'{case}_mult'] = m._stiffness_multiplier; out[f'{case}_model'] = model; out[f'{case}_nu'] = nu
        out[f'{case}_matmult'] = mm; out[f'{case}_func'] = func
        out[f'{case}_ntab'] = np.array(len(tabs))
        for k, (xs, ys) in enumerate(tabs):
            out[f'{case}_tab{k}_x'] = xs; out[f'{case}_tab{k}_y'] = ys
        r, c, d = coo(K)
        out[f'{case}_K_r'] = r; out[f'{case}_K_c'] = c; out[f'{case}_K_d'] = d
        out[f'{case}_stress'] = stress
        # the same with the shape matrices taken at the MOVING gear (what a Newton-Raphson step asks for): zero stress of the
        # engineering part, tangent of the others at zero displacement, area stretch still INITIAL -> MOVING
        K2, stress2 = m.stiffness_matrix(gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_MOVING))
        r, c, d = coo(K2)
        out[f'{case}_K2_r'] = r; out[f'{case}_K2_c'] = c; out[f'{case}_K2_d'] = d
        out[f'{case}_stress2'] = stress2
    # ---- Newton-Raphson: wrinkle / SVK regions on the left are pulled into compression (area stretch 0.8 .. 0.95 of the linear
    #      triangles': the branch the wrinkle material exists for), fold / NHK regions on the right are stretched a little.  (With
    #      triangles hopping across the cliff of the wrinkle table between 1.0 and 1.01 the reference's iteration -- K re-assembled
    #      every step, f(J) not differentiated -- oscillates and its end state is not a property of the system; kept out of the pin.)
    tab_nr = {k: dict(tab_full[k]) for k in ('default', 'wrinkle', 'fold', 'svkf')}
    tab_nr['nhkf'] = dict(tab_full['nhkf'], stiffness_func_params=dict(strain=[0.5, 0.9, 1.1, 1.5], stiffness=[1.6, 1.1, 0.9, 0.6]))
    region_nr = lambda c: np.select([c[:, 0] < 30, c[:, 0] < 55, (c[:, 0] >= 70) & (c[:, 0] < 100) & (c[:, 1] > 50), c[:, 0] >= 100], [7, 9, 8, 10], 0)
    names_nr = ['default', 'wrinkle', 'fold', 'svkf', 'nhkf']
    n = 300
    det = dict(tolerated_perturbation=None, callback_settings={'chances': None, 'eval_step': 10})
    ramp = np.clip((70.0 - v[:, 0]) / 25.0, 0.0, 1.0)                      # 1 on the left, 0 from x = 70 on
    ramp = ramp * ramp * (3 - 2 * ramp)
    pull = v + np.stack((0.11 * (60.0 - v[:, 0]) * ramp + 0.02 * np.maximum(v[:, 0] - 90.0, 0.0), 0.4 * np.sin(v[:, 0] / 40.0)), -1) + np.array([[0.3, -0.2]])

    def build():
        mt = material.MaterialTable(table={k: dict(tab_nr[k]) for k in names_nr})
        m1 = Mesh(v.copy(), t.copy(), material_table=mt, material_ids=region_nr(ctr).astype(np.int16), stiffness_multiplier=mult.copy(), uid=1)
        return mt, m1
    mt, m1 = build()
    tt = m1.triangles
    model, nu, mm, func, tabs = describe(m1, mt, names_nr)
    out.update(nr_t=tt, nr_pull=pull, nr_mult=m1._stiffness_multiplier, nr_model=model, nr_nu=nu, nr_matmult=mm, nr_func=func, nr_ntab=np.array(len(tabs)))
    for k, (xs, ys) in enumerate(tabs):
        out[f'nr_tab{k}_x'] = xs; out[f'nr_tab{k}_y'] = ys
    tid = rng.integers(0, tt.shape[0], n); B = rng.dirichlet((1, 1, 1), n)
    w = rng.uniform(0.4, 1.0, n).astype(np.float32)
    out.update(nr_tid=tid, nr_B=B, nr_w=w)
    for case, call in (('nr', lambda slm: slm.optimize_Newton_Raphson(max_newtonstep=12, tol=1e-9, **det)),
                       ('nr3', lambda slm: slm.optimize_Newton_Raphson(max_newtonstep=3, tol=1e-6, **det)),
                       ('elastic', lambda slm: slm.optimize_elastic(max_newtonstep=10, tol=1e-8, **det))):
        mt, m1 = build()
        m0 = Mesh(pull.copy(), tt.copy(), uid=0, locked=True)
        lk = optimizer.Link(m0, m1, tid, tid, B, B, weight=w)
        slm = optimizer.SLM([m0, m1], links=[lk], stiffness_lambda=1.0, crosslink_lambda=1.0)
        c0, c1 = call(slm)
        out[f'{case}_cost'] = np.array([c0, c1], dtype=np.float64)
        out[f'{case}_v_after'] = m1.vertices(gear=const.MESH_GEAR_MOVING)
        out[f'{case}_off_after'] = m1.offset(gear=const.MESH_GEAR_MOVING)
    np.savez_compressed(os.path.join(OUT, 'g19_area_stretch.npz'), **out)


# ----------------------------------------------------------------------- G20
def g20_xcorr_normalized():
    """xcorr_fft(normalize=True) (matcher.py:71-81, 119-122): the correlation surfaces divided by the clipped, max-normalised
    correlation of the masks -- with explicit masks (holes), with the default all-ones masks, padded and not, all three confidence
    modes.  No call site of the reference enables it; pinned for the completeness of row a1."""
    rng = np.random.default_rng(2020)
    out = {}
    H, W, n = 64, 60, 4
    big = common.masked_dog_filter(to_u8(texture(rng, 4 * H, 4 * W)), 2.0).astype(np.float32)
    shifts = [(0, 0), (4, -3), (-9, 6), (13, 11)]
    i0 = np.zeros((n, H, W), np.float32); i1 = np.zeros((n, H, W), np.float32)
    for k, (sy, sx) in enumerate(shifts):
        y0 = H + 7 * k; x0 = W + 5 * k
        i0[k] = big[y0:y0 + H, x0:x0 + W]
        i1[k] = big[y0 + sy:y0 + sy + H, x0 + sx:x0 + sx + W]
    i1 += 0.05 * i1.std() * rng.standard_normal(i1.shape).astype(np.float32)
    m0 = np.ones((H, W), np.float32); m0[:18, :25] = 0; m0[50:, 40:] = 0
    m1 = np.ones((H, W), np.float32); m1[20:44, 30:] = 0
    i0m = i0 * m0; i1m = i1 * m1
    out.update(img0=i0, img1=i1, mask0=m0, mask1=m1)
    for tag, (a, b, kw) in {'masks': (i0m, i1m, dict(mask0=m0, mask1=m1)), 'ones': (i0, i1, {})}.items():
        for pad in (True, False):
            for cm in (0, 1, 2):
                dx, dy, cf = matcher.xcorr_fft(a, b, conf_mode=cm, pad=pad, subpixel=True, normalize=True, **kw)
                key = f'{tag}_p{int(pad)}_c{cm}'
                out[key + '_dx'] = np.asarray(dx, dtype=np.float64); out[key + '_dy'] = np.asarray(dy, dtype=np.float64)
                out[key + '_conf'] = np.asarray(cf)
    np.savez_compressed(os.path.join(OUT, 'g20_xcorr_normalized.npz'), **out)


# ----------------------------------------------------------------------- G21
def g21_grouped_dof():
    """SLM.optimize_linear(groupings=..., remove_extra_dof=True) (optimizer.py:1360-1415: the selector of the held degrees of
    freedom is made mesh by mesh and folded into the groups, `edc = (T_m @ edc) > 0` -- a degree of freedom of a group is solved
    when ANY member has it solved): three free meshes, nothing locked; mesh 0 is alone in its group and is the first mesh of the
    floating system, so its first three degrees of freedom are held; meshes 1 and 2 share theirs."""
    rng = np.random.default_rng(2121)
    out = {}
    va, ta = grid(8, 7, 10.0)
    vb, tb = grid(9, 6, 11.0, origin=(3.0, -2.0))
    ms = [Mesh(vb, tb, uid=0), Mesh(va, ta, uid=1), Mesh(va + np.array([0.3, 0.2]), ta, uid=2, soft_factor=0.8)]
    ms[1].apply_translation((0.6, -0.4), const.MESH_GEAR_FIXED)
    ms[2].apply_translation((-1.0, 1.5), const.MESH_GEAR_FIXED)
    links = []
    for k, (a, b, n) in enumerate(((0, 1, 40), (0, 2, 35))):
        tid0 = rng.integers(0, ms[a].triangles.shape[0], size=n)
        tid1 = rng.integers(0, ms[b].triangles.shape[0], size=n)
        B0 = rng.dirichlet((1, 1, 1), size=n); B1 = rng.dirichlet((1, 1, 1), size=n)
        w = rng.uniform(0.3, 1.0, size=n).astype(np.float32)
        links.append(optimizer.Link(ms[a], ms[b], tid0, tid1, B0, B1, weight=w))
        out[f'l{k}_ab'] = np.array([a, b]); out[f'l{k}_tid0'] = tid0; out[f'l{k}_tid1'] = tid1
        out[f'l{k}_B0'] = B0; out[f'l{k}_B1'] = B1; out[f'l{k}_w'] = w
    for k, m in enumerate(ms):
        out[f'm{k}_v'] = m.vertices(gear=const.MESH_GEAR_INITIAL); out[f'm{k}_t'] = m.triangles
        out[f'm{k}_off'] = m.offset(gear=const.MESH_GEAR_FIXED)
        out[f'm{k}_soft'] = np.float64(m.soft_factor)
    groupings = np.array([0, 1, 1])
    slm = optimizer.SLM(ms, links=links, stiffness_lambda=1.0, crosslink_lambda=-1.0)
    # (allow_direct_solve reaches optimizer.solve through callback_settings: the reference factorises the reduced system itself
    # -- held at one corner the floating system is too soft for its MINRES legs to finish)
    cost = slm.optimize_linear(tol=1e-11, groupings=groupings, remove_extra_dof=True, tolerated_perturbation=None,
                               callback_settings={'chances': None, 'eval_step': 10, 'allow_direct_solve': True}, check_converge=True)
    out['groupings'] = groupings
    out['cost'] = np.array(cost, dtype=np.float64)
    for k, m in enumerate(ms):
        out[f'm{k}_v_after'] = m.vertices(gear=const.MESH_GEAR_MOVING); out[f'm{k}_off_after'] = m.offset(gear=const.MESH_GEAR_MOVING)
    np.savez_compressed(os.path.join(OUT, 'g21_grouped_dof.npz'), **out)


# ----------------------------------------------------------------------- G22
def g22_schedule_walks():
    """the coarse-to-fine walk of iterative_xcorr_matcher_w_mesh (matcher.py:567-716) captured from the REFERENCE's own loop:
    the block matcher is replaced by a script (every block reports one prescribed displacement, confidence 1), everything else
    -- distributor, links, relaxations, the spacing / padding decisions -- runs as it stands.  Per call of the block matcher the
    record is (block side = spacing, pad, subpixel, affine_approx_tol, number of blocks); scenarios vary the spacing list,
    allow_enlarge, allow_dwell, max_spacing_skip, a fixed pad, and the displacements the rounds report."""
    import json
    v0, t0 = grid(41, 41, 50.0)                             # 2000 x 2000 px
    scenarios = {
        'two_jump':         dict(spacings=[1024, 75], dis=[12.0, 0.3]),
        'two_nojump':       dict(spacings=[1024, 75], dis=[30.0, 0.3]),
        'three_clip':       dict(spacings=[1000, 300, 75], dis=[2.0, 2.0, 2.0]),
        'three_skip1':      dict(spacings=[1000, 300, 75], dis=[2.0, 2.0, 2.0], max_spacing_skip=1),
        'three_skip2_mid':  dict(spacings=[1000, 300, 75], dis=[30.0, 10.0, 2.0], max_spacing_skip=2),
        'dwell2':           dict(spacings=[400, 100], dis=[200.0] * 8, allow_dwell=2),
        'dwell1_then_jump': dict(spacings=[400, 100, 30], dis=[150.0, 20.0, 20.0, 1.0, 1.0], allow_dwell=1),
        'enlarge':          dict(spacings=[400, 100], dis=[300.2, 300.2, 1.0, 1.0], allow_enlarge=True),
        'enlarge_off':      dict(spacings=[400, 100], dis=[300.0, 1.0], allow_enlarge=False),
        'pad_false':        dict(spacings=[1024, 75], dis=[30.0, 0.3], pad=False),
        'pad_true':         dict(spacings=[1000, 300, 75], dis=[2.0, 2.0, 2.0], pad=True),
        'tiny_dis':         dict(spacings=[600, 150], dis=[0.05, 0.05]),
        'four':             dict(spacings=[1600, 400, 100, 25], dis=[90.0, 20.0, 5.0, 1.0, 1.0]),
        'unsorted':         dict(spacings=[75, 1024, 300], dis=[60.0, 15.0, 1.0]),
    }
    out = {}
    real = matcher.bboxes_mesh_renderer_matcher
    try:
        for name, sc in scenarios.items():
            m0 = Mesh(v0, t0, uid=0); m0.lock()
            m1 = Mesh(v0 + 0.0, t0, uid=1)
            dis = list(sc['dis'])
            calls = []

            def scripted(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **kw):
                k = min(len(calls), len(dis) - 1)
                c0 = common.bbox_centers(bboxes0)
                side = float(np.round(bboxes0[0, 2] - bboxes0[0, 0]))
                calls.append([side, bool(kw.get('pad')), bool(kw.get('subpixel')), float(kw.get('affine_approx_tol')), int(bboxes0.shape[0])])
                d = np.array([dis[k], 0.0])
                return c0 - 0.5 * d, c0 + 0.5 * d, np.ones(c0.shape[0], dtype=np.float32)
            matcher.bboxes_mesh_renderer_matcher = scripted
            kw = {k: v for k, v in sc.items() if k not in ('spacings', 'dis')}
            res = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, None, None, spacings=np.array(sc['spacings'], dtype=np.float64), compute_strain=False,
                                                         conf_thresh=0.3, residue_len=0, opt_tol=1e-10, **kw)
            out[name + '_calls'] = np.array(calls, dtype=np.float64)
            out[name + '_nmatch'] = np.int64(0 if res[0] is None else res[0].shape[0])
    finally:
        matcher.bboxes_mesh_renderer_matcher = real
    out['scenarios'] = np.frombuffer(json.dumps(scenarios).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, 'g22_schedule_walks.npz'), **out)


# ----------------------------------------------------------------------- G23
def scripted_block_matches(rnd, bboxes0, bboxes1, seed):
    """the script both G23 and its test hand to the loop in place of render + DoG + NCC: every block reports a smooth
    displacement that shrinks from round to round, a deterministic pseudo-random confidence in [0.15, 1) and, for a few blocks,
    an outlier (what the huber / threshold residue weights are there for); the block -> point rule is matcher.py:840-849"""
    b0 = np.asarray(bboxes0, dtype=np.float64); b1 = np.asarray(bboxes1, dtype=np.float64)
    c0 = 0.5 * (b0[:, :2] + b0[:, 2:]); c1 = 0.5 * (b1[:, :2] + b1[:, 2:])
    s0 = np.stack((b0[:, 3] - b0[:, 1], b0[:, 2] - b0[:, 0]), axis=-1); s1 = np.stack((b1[:, 3] - b1[:, 1], b1[:, 2] - b1[:, 0]), axis=-1)
    ratio = (s0 / (s0 + s1))[:, ::-1]
    amp = (6.0, 2.0, 0.6, 0.2)[min(rnd, 3)]
    dx = amp * np.sin(c0[:, 1] / 310.0 + 0.4 + rnd) + 0.3 * amp * (c0[:, 0] / 2000.0)
    dy = amp * np.cos(c0[:, 0] / 270.0 - rnd) - 0.2 * amp * (c0[:, 1] / 2000.0)
    h = np.abs(np.modf(np.sin(np.round(c0[:, 0]) * 12.9898 + np.round(c0[:, 1]) * 78.233 + 37.0 * rnd + seed) * 43758.5453)[0])
    conf = (0.15 + 0.85 * h).astype(np.float32)
    dxy = np.stack((dx, dy), axis=-1)
    dxy[h > 0.93] += np.array([8.0, -5.0])
    return c0 - dxy * ratio, c1 + dxy * (1 - ratio), conf


def g23_matcher_loop():
    """iterative_xcorr_matcher_w_mesh (matcher.py:430-778) END TO END with its block matcher scripted (scripted_block_matches):
    blocks on the bounds of the moving gears, confidence filter, links through the moving gears, relaxation, relax_first + huber
    / threshold residue weights and the second solve, the walk over the spacings, the final matches in the initial gear and the
    strain chain -- the composite the oracle restates in region_ref.section_match / pipeline_ref.match_pair.  The reference's
    solves are made deterministic and converged (tol 1e-11, no random-perturbation exit), as for G14 / G16."""
    real_match = matcher.bboxes_mesh_renderer_matcher
    real_ol = optimizer.SLM.optimize_linear

    def converged(self, **kw):
        kw['tol'] = 1e-11; kw['tolerated_perturbation'] = None; kw['callback_settings'] = {'chances': None, 'eval_step': 10}
        kw['check_converge'] = True
        return real_ol(self, **kw)
    rng = np.random.default_rng(2323)
    out = {}
    va, ta = grid(27, 20, 60.0)                             # 1560 x 1140 px, locked
    gx, gy = np.meshgrid(np.arange(0, 1561, 65.0), np.arange(0, 1141, 57.0))
    vb = np.stack((gx.ravel(), gy.ravel()), axis=-1)
    inner = (vb[:, 0] > 0) & (vb[:, 0] < gx.max()) & (vb[:, 1] > 0) & (vb[:, 1] < gy.max())
    vb[inner] += rng.uniform(-0.25, 0.25, (int(inner.sum()), 2)) * 57.0
    tb = Delaunay(vb).simplices.astype(np.int32)
    pb = vb[tb]
    area = (pb[:, 1, 0] - pb[:, 0, 0]) * (pb[:, 2, 1] - pb[:, 0, 1]) - (pb[:, 1, 1] - pb[:, 0, 1]) * (pb[:, 2, 0] - pb[:, 0, 0])
    tb[area < 0] = tb[area < 0][:, ::-1]
    out.update(v0=va, t0=ta, v1=vb, t1=tb)
    cases = {'huber': dict(spacings=[400.0, 100.0], residue_mode='huber', residue_len=3.0, seed=1.0, off0=(3.0, -2.0)),
             'threshold3': dict(spacings=[600.0, 150.0, 40.0], residue_mode='threshold', residue_len=4.0, seed=2.0, off0=(-1.5, 2.5)),
             'no_residue': dict(spacings=[300.0, 80.0], residue_mode='huber', residue_len=0.0, seed=3.0, off0=(0.0, 0.0))}
    try:
        matcher.bboxes_mesh_renderer_matcher = None        # (set per case below)
        optimizer.SLM.optimize_linear = converged
        for name, cs in cases.items():
            m0 = Mesh(va, ta, uid=0)
            m0.apply_translation(cs['off0'], const.MESH_GEAR_FIXED)
            m0.lock()
            m1 = Mesh(vb.copy(), tb, uid=1)
            rounds = []

            def scripted(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **kw):
                k = len(rounds)
                rounds.append(dict(bboxes0=np.array(bboxes0), bboxes1=np.array(bboxes1), pad=bool(kw.get('pad')), subpixel=bool(kw.get('subpixel')),
                                   field1=mesh1.vertices_w_offset(gear=const.MESH_GEAR_MOVING) - mesh1.vertices_w_offset(gear=const.MESH_GEAR_INITIAL)))
                return scripted_block_matches(k, bboxes0, bboxes1, cs['seed'])
            matcher.bboxes_mesh_renderer_matcher = scripted
            xy0, xy1, wt, strain = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, None, None, spacings=np.array(cs['spacings']), distributor='cartesian_bbox',
                                                                          conf_thresh=0.3, residue_len=cs['residue_len'], residue_mode=cs['residue_mode'],
                                                                          compute_strain=True, stiffness_lambda=0.5, min_num_blocks=2)
            out[f'{name}_nrounds'] = np.int64(len(rounds))
            for k, r in enumerate(rounds):
                for key in ('bboxes0', 'bboxes1', 'field1'):
                    out[f'{name}_r{k}_{key}'] = r[key]
                out[f'{name}_r{k}_flags'] = np.array([r['pad'], r['subpixel']])
            out[f'{name}_xy0'] = xy0; out[f'{name}_xy1'] = xy1; out[f'{name}_weight'] = np.asarray(wt); out[f'{name}_strain'] = np.float64(strain)
            out[f'{name}_field1_final'] = m1.vertices_w_offset(gear=const.MESH_GEAR_MOVING) - m1.vertices_w_offset(gear=const.MESH_GEAR_INITIAL)
            out[f'{name}_params'] = np.array([cs['residue_len'], cs['seed'], cs['off0'][0], cs['off0'][1], 1.0 if cs['residue_mode'] == 'threshold' else 0.0])
            out[f'{name}_spacings'] = np.array(cs['spacings'])
    finally:
        matcher.bboxes_mesh_renderer_matcher = real_match
        optimizer.SLM.optimize_linear = real_ol
    np.savez_compressed(os.path.join(OUT, 'g23_matcher_loop.npz'), **out)


# ----------------------------------------------------------------------- G24
def scripted_strip_blocks(case, rnd, bboxes0, bboxes1, H, W):
    """(dx, dy, conf) of every block of a round, in place of crop + xcorr_fft: 'rigid' -- the coarse round reports ONE integer
    vector (the relaxation is then a rigid translation), 'deformed' / 'three' -- the coarse rounds disagree along the strip (mesh1
    bends); the last round reports a small smooth field with a few outliers for the residue weights"""
    b0 = np.asarray(bboxes0, dtype=np.float64)
    c = 0.5 * (b0[:, :2] + b0[:, 2:])
    u, v = c[:, 0] / W, c[:, 1] / H
    h = np.abs(np.modf(np.sin(np.round(c[:, 0]) * 12.9898 + np.round(c[:, 1]) * 78.233 + 37.0 * rnd + len(case)) * 43758.5453)[0])
    nrounds_coarse = 2 if case == 'three' else 1
    if rnd < nrounds_coarse:
        if case == 'rigid':
            dx, dy = np.full(c.shape[0], 3.0), np.full(c.shape[0], -2.0)
        else:
            a = 1.0 / (1 + rnd)
            dx = a * (2.0 + 3.0 * v - 1.0 * u); dy = a * (-1.0 + 2.5 * np.sin(3.0 * v + rnd))
        conf = np.full(c.shape[0], 0.9, dtype=np.float32)
        if c.shape[0] > 2:
            conf[-1] = 0.2                                   # one block below the threshold
    else:
        dx = 0.3 * np.sin(7.0 * v + 2.0 * u) + 0.05 * (h - 0.5); dy = 0.25 * np.cos(5.0 * v) - 0.05 * (h - 0.5)
        conf = (0.2 + 0.8 * h).astype(np.float32)
        out = h > 0.9
        dx = dx + 6.0 * out; dy = dy - 4.0 * out
    return dx, dy, conf


def g24_strip_loop():
    """stitching_matcher's loop (matcher.py:353-364 -> 430-778) with its block matches scripted, on the cartesian mesh pair of a
    strip: mesh0 translated by the global translation and locked, mesh1 free.  Three cases: a rigid coarse round, a bending
    one, three spacings with two bending rounds.  The meshes are built from the grid of Mesh.from_bbox(cartesian=True)
    (mesh.py:406-433) with the triangles the oracle uses (the reference's come from `triangle`)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from oracle import pipeline_ref as _pr                  # (only its cartesian_mesh: vertices of mesh.py:406-433 + a triangulation)
    real_match = matcher.bboxes_mesh_renderer_matcher
    real_ol = optimizer.SLM.optimize_linear

    def converged(self, **kw):
        kw['tol'] = 1e-11; kw['tolerated_perturbation'] = None; kw['callback_settings'] = {'chances': None, 'eval_step': 10}
        kw['check_converge'] = True
        return real_ol(self, **kw)
    out = {}
    cases = {'rigid': dict(H=1024, W=256, spacings=[256.0, 64.0], t0=(5.0, -3.0), residue_len=2.0),
             'deformed': dict(H=1024, W=256, spacings=[256.0, 64.0], t0=(-4.0, 6.0), residue_len=2.0),
             'three': dict(H=2048, W=128, spacings=[512.0, 128.0, 32.0], t0=(2.0, 1.0), residue_len=3.0)}
    try:
        optimizer.SLM.optimize_linear = converged
        for name, cs in cases.items():
            H, W = cs['H'], cs['W']
            v, tri, xs, ys = _pr.cartesian_mesh(W, H, float(min(cs['spacings'])), min_num_blocks=2)
            m0 = Mesh(v, tri, uid=0)
            m0.apply_translation(cs['t0'], const.MESH_GEAR_FIXED)
            m0.lock()
            m1 = Mesh(v.copy(), tri, uid=1)
            rounds = []

            def scripted(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **kw):
                k = len(rounds)
                rounds.append(dict(bboxes0=np.array(bboxes0), bboxes1=np.array(bboxes1), pad=bool(kw.get('pad')), subpixel=bool(kw.get('subpixel')),
                                   field1=mesh1.vertices_w_offset(gear=const.MESH_GEAR_MOVING) - mesh1.vertices_w_offset(gear=const.MESH_GEAR_INITIAL)))
                dx, dy, conf = scripted_strip_blocks(name, k, bboxes0, bboxes1, H, W)
                c0 = common.bbox_centers(bboxes0); c1 = common.bbox_centers(bboxes1)
                s0 = common.bbox_sizes(bboxes0); s1 = common.bbox_sizes(bboxes1)
                ratio = (s0 / (s0 + s1))[:, ::-1]
                dxy = np.stack((dx, dy), axis=-1)
                return c0 - dxy * ratio, c1 + dxy * (1 - ratio), conf
            matcher.bboxes_mesh_renderer_matcher = scripted
            xy0, xy1, wt, strain = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, None, None, spacings=np.array(cs['spacings']), distributor='cartesian_bbox',
                                                                          conf_thresh=0.33, residue_len=cs['residue_len'], residue_mode='huber', opt_tol=None,
                                                                          compute_strain=True, min_num_blocks=2)
            out[f'{name}_nrounds'] = np.int64(len(rounds))
            for k, r in enumerate(rounds):
                for key in ('bboxes0', 'bboxes1', 'field1'):
                    out[f'{name}_r{k}_{key}'] = r[key]
                out[f'{name}_r{k}_flags'] = np.array([r['pad'], r['subpixel']])
            out[f'{name}_xy0'] = xy0; out[f'{name}_xy1'] = xy1; out[f'{name}_weight'] = np.asarray(wt); out[f'{name}_strain'] = np.float64(strain)
            out[f'{name}_params'] = np.array([H, W, cs['t0'][0], cs['t0'][1], cs['residue_len']], dtype=np.float64)
            out[f'{name}_spacings'] = np.array(cs['spacings'])
    finally:
        matcher.bboxes_mesh_renderer_matcher = real_match
        optimizer.SLM.optimize_linear = real_ol
    np.savez_compressed(os.path.join(OUT, 'g24_strip_loop.npz'), **out)


# ----------------------------------------------------------------------- G25
def scripted_strip_result(k, img0, img1, mask0, mask1, cfg):
    """what the scripted stitching_matcher hands back for its k-th call, and what it notes about the call"""
    note = np.array([img0.shape[0], img0.shape[1], img1.shape[0], img1.shape[1], int(img0.astype(np.int64).sum()), int(img1.astype(np.int64).sum()),
                     -1 if mask0 is None else int(np.count_nonzero(mask0)), -1 if mask1 is None else int(np.count_nonzero(mask1))], dtype=np.int64)
    if k % 4 == 3:
        return note, (None, None, cfg.get('conf_thresh', 0.3), None, None)
    h0, w0 = img0.shape[:2]; h1, w1 = img1.shape[:2]
    xy0 = np.array([[1.0, 2.0], [0.5 * w0, 0.5 * h0], [w0 - 3.0 + 0.25 * k, h0 - 2.0]])
    xy1 = np.array([[2.0, 1.0], [0.5 * w1 - 0.5, 0.5 * h1 + 0.125 * k], [w1 - 4.0, h1 - 1.0]])
    weight = np.array([0.5, 0.75, 0.25 + 0.01 * k])
    phtm = (10.0 + k, 20.0 + k, 3.0, 4.0) if cfg.get('compute_photometric', False) else None
    return note, (xy0, xy1, weight, 0.01 * (k + 1), phtm)


def g25_overlap_bookkeeping():
    """Stitcher.subprocess_match_list_of_overlaps (stitcher.py:474-621) around a scripted stitching_matcher: which strips it
    crops for every overlap (margins as a ratio and in pixels, the minimum width, boxes clipped to their tiles), the masks it
    derives from `maskout_val`, the offsets it adds to the matches and the keys it files them under (`index_mapper`).  The image
    loader is a stand-in that cuts the asked box out of arrays held in memory (dal.StaticImageLoader reads files through cv2)."""
    from feabas import stitcher as rstitcher
    rng = np.random.default_rng(25)
    th, tw = 300, 400
    bboxes = []
    for r in range(2):
        for c in range(3):
            x0 = c * (tw - 44) + int(rng.integers(-6, 7)); y0 = r * (th - 38) + int(rng.integers(-6, 7))
            bboxes.append((x0, y0, x0 + tw, y0 + th))
    bboxes = np.array(bboxes, dtype=np.int64)
    yy, xx = np.meshgrid(np.arange(-20, 2 * th + 20), np.arange(-20, 3 * tw + 20), indexing='ij')
    world = ((xx * 7 + yy * 13 + (xx * yy) // 31) % 251 + 1).astype(np.uint8)             # 1..251: no pixel is 0 by itself
    tiles = []
    for k, (x0, y0, x1, y1) in enumerate(bboxes):
        t = world[y0 + 20:y1 + 20, x0 + 20:x1 + 20].copy()
        if k in (1, 4):                                         # a blanked corner patch and a blanked band (the scanner's fill value)
            t[:40, -60:] = 0
            t[-25:, 100:180] = 0
        tiles.append(t)
    overlaps = np.array([(1, 0), (2, 1), (3, 0), (4, 1), (4, 3), (5, 2), (5, 4), (4, 0), (3, 1), (5, 1), (4, 2)], dtype=np.int64)

    class ArrayLoader:
        def __init__(self, imgpaths, bxs, root_dir=None, **cfg):
            self.imgrootdir = root_dir; self.imgrelpaths = list(imgpaths); self.cfg = cfg; self.bxs = np.asarray(bxs)
            self.filepaths_generator = list(imgpaths)

        def crop(self, bbox, idx, return_index=False, **kw):
            x0, y0, x1, y1 = (int(v) for v in bbox)
            bx0, by0 = self.bxs[idx][:2]
            assert x0 >= bx0 and y0 >= by0 and x1 <= self.bxs[idx][2] and y1 <= self.bxs[idx][3]
            return tiles[idx][y0 - by0:y1 - by0, x0 - bx0:x1 - bx0]

        def clear_cache(self):
            pass
    cases = {'ratio_margin': dict(margin=1.0, min_overlap_width=0),
             'masked_mapped': dict(margin=0.5, min_overlap_width=35, maskout_val=0, index_mapper=np.arange(6) + 100,
                                   matcher_config=dict(compute_photometric=True, conf_thresh=0.4)),
             'pixel_margin': dict(margin=30, min_overlap_width=10, maskout_val=0)}
    out = dict(bboxes=bboxes, overlaps=overlaps)
    for k, t in enumerate(tiles):
        out[f'tile{k}'] = t
    real_loader, real_matcher = rstitcher.StaticImageLoader, rstitcher.stitching_matcher
    try:
        rstitcher.StaticImageLoader = ArrayLoader
        for name, kw in cases.items():
            notes = []

            def scripted(img0, img1, mask0=None, mask1=None, **cfg):
                note, res = scripted_strip_result(len(notes), img0, img1, mask0, mask1, cfg)
                notes.append(note)
                return res
            rstitcher.stitching_matcher = scripted
            matches, strains, phtm, err = rstitcher.Stitcher.subprocess_match_list_of_overlaps(overlaps, [f't{k}.png' for k in range(6)], bboxes, **kw)
            assert not err
            keys = sorted(matches)
            out[f'{name}_calls'] = np.stack(notes)
            out[f'{name}_keys'] = np.array(keys, dtype=np.int64).reshape(-1, 2)
            for j, key in enumerate(keys):
                out[f'{name}_m{j}_xy0'], out[f'{name}_m{j}_xy1'], out[f'{name}_m{j}_w'] = matches[key]
                out[f'{name}_m{j}_strain'] = np.float64(strains[key])
                out[f'{name}_m{j}_phtm'] = np.array(phtm[key], dtype=np.float64) if key in phtm else np.empty(0)
    finally:
        rstitcher.StaticImageLoader, rstitcher.stitching_matcher = real_loader, real_matcher
    np.savez_compressed(os.path.join(OUT, 'g25_overlap_bookkeeping.npz'), **out)


# ----------------------------------------------------------------------- G26
def g26_inputs():
    """three meshes: a square (locked), an L-shaped one, and one that falls into two islands; links 0-1, 1-2, 0-2 with a few matches
    that miss (outside a mesh) and weights; everything at the INITIAL gear, mesh 2 displaced at MOVING"""
    rng = np.random.default_rng(26)

    def grid(x0, y0, nx, ny, step, drop=None):
        gx, gy = np.meshgrid(x0 + step * np.arange(nx + 1), y0 + step * np.arange(ny + 1))
        v = np.stack((gx.ravel(), gy.ravel()), axis=-1).astype(np.float64)
        tri = []
        for j in range(ny):
            for i in range(nx):
                if drop is not None and drop(i, j):
                    continue
                a = j * (nx + 1) + i; b = a + 1; c = a + nx + 1; d = c + 1
                tri += [(a, b, d), (a, d, c)]
        tri = np.array(tri, dtype=np.int32)
        used = np.unique(tri)
        remap = -np.ones(v.shape[0], dtype=np.int64); remap[used] = np.arange(used.size)
        return v[used], remap[tri].astype(np.int32)
    v0, t0 = grid(0, 0, 6, 6, 50.0)
    v1, t1 = grid(180, 20, 6, 6, 50.0, drop=lambda i, j: i >= 3 and j >= 3)
    v2, t2 = grid(40, 220, 8, 4, 50.0, drop=lambda i, j: i in (3, 4))            # two islands: columns 0-2 and 5-7
    pts = {}
    for name, (lo, hi) in {'01': ((190, 30), (295, 295)), '12': ((185, 225), (440, 330)), '02': ((45, 225), (295, 295))}.items():
        n = 40
        p = rng.uniform(lo, hi, (n, 2))
        pts[name] = (p, p + rng.normal(0, 1.5, (n, 2)), rng.uniform(0.2, 1.0, n))
    d2 = 3.0 * np.stack((np.sin(v2[:, 1] / 70.0), np.cos(v2[:, 0] / 90.0)), axis=-1)
    return (v0, t0), (v1, t1), (v2, t2), pts, d2


def g26_slm_bookkeeping():
    """the host-side bookkeeping of SLM around the solves (optimizer.py:637-754, 1678-1703, 1758-1858): links from coordinates
    (which matches survive the point location), linkage_adjacency, connected_subsystems, match_residues,
    divide_disconnected_submeshes with prune_links / distribute_link (the sub-mesh uids, which part gets which matches)"""
    (v0, t0), (v1, t1), (v2, t2), pts, d2 = g26_inputs()
    out = dict(v0=v0, t0=t0, v1=v1, t1=t1, v2=v2, t2=t2, d2=d2)
    m0 = Mesh(v0, t0, uid=0); m0.lock()
    m1 = Mesh(v1, t1, uid=1)
    m2 = Mesh(v2, t2, uid=2)
    m2.set_vertices(v2 + d2, const.MESH_GEAR_MOVING)
    opt = optimizer.SLM([m0, m1, m2])
    from matplotlib.tri import Triangulation
    meshes = {0: (m0, v0, t0), 1: (m1, v1, t1), 2: (m2, v2, t2)}

    def locate(k, p):
        # (the reference's own tri_finder needs shapely as soon as a mesh has two regions, like mesh 2: the links of the whole
        # meshes are made from triangle ids and barycentric coordinates located here and stored in the fixture)
        _, v, t = meshes[k]
        tid = np.asarray(Triangulation(v[:, 0], v[:, 1], t).get_trifinder()(p[:, 0], p[:, 1]))
        ok = tid >= 0
        tri = v[t[np.where(ok, tid, 0)]]
        T = np.stack((tri[:, 0] - tri[:, 2], tri[:, 1] - tri[:, 2]), axis=-1)
        l12 = np.linalg.solve(T, (p - tri[:, 2])[..., None])[..., 0]
        return tid, ok, np.concatenate((l12, 1 - l12.sum(axis=1, keepdims=True)), axis=1)
    for name, (a, b) in {'01': (0, 1), '12': (1, 2), '02': (0, 2)}.items():
        p0, p1, w = pts[name]
        tid0, ok0, B0 = locate(a, p0); tid1, ok1, B1 = locate(b, p1)
        ok = ok0 & ok1
        out[f'lk{name}_tid0'], out[f'lk{name}_tid1'], out[f'lk{name}_B0'], out[f'lk{name}_B1'], out[f'lk{name}_w'] = tid0[ok], tid1[ok], B0[ok], B1[ok], w[ok]
        opt.add_link(optimizer.Link(meshes[a][0], meshes[b][0], tid0[ok], tid1[ok], B0[ok], B1[ok], weight=w[ok]))

    def describe(tag):
        out[f'{tag}_nlinks'] = np.int64(len(opt.links))
        out[f'{tag}_mesh_uids'] = np.array([m.uid for m in opt.meshes], dtype=np.float64)
        out[f'{tag}_mesh_ntri'] = np.array([m.num_triangles for m in opt.meshes], dtype=np.int64)
        for k, lk in enumerate(opt.links):
            out[f'{tag}_l{k}_uids'] = np.array(lk.uids, dtype=np.float64)
            out[f'{tag}_l{k}_xy0'] = lk.xy0(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True)
            out[f'{tag}_l{k}_xy1'] = lk.xy1(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True)
            out[f'{tag}_l{k}_w'] = np.asarray(lk.weight(use_mask=False), dtype=np.float64)
        # (the reference caches the first matrix it makes, whatever `directional` was: the cache is emptied between the two calls)
        opt._linkage_adjacency = None; opt._connected_subsystems = None
        out[f'{tag}_adj_dir'] = opt.linkage_adjacency(directional=True).toarray()
        opt._linkage_adjacency = None
        out[f'{tag}_adj'] = opt.linkage_adjacency().toarray()
        lab, ncomp = opt.connected_subsystems
        out[f'{tag}_ncomp'] = np.int64(ncomp); out[f'{tag}_labels'] = np.asarray(lab, dtype=np.int64)
        for q in (0, 0.75, 1):
            out[f'{tag}_res_q{q}'] = opt.match_residues(gear=const.MESH_GEAR_MOVING, quantile=q)
        out[f'{tag}_res_init'] = opt.match_residues(gear=const.MESH_GEAR_INITIAL, quantile=0.5)
    describe('whole')
    out['divided'] = np.bool_(opt.divide_disconnected_submeshes())
    describe('parts')
    np.savez_compressed(os.path.join(OUT, 'g26_slm_bookkeeping.npz'), **out)


# ----------------------------------------------------------------------- G27
sys.path.insert(0, OUT)
from walks import g27_mesh_walk                       # noqa: E402  (the statements that drive the reference here and the product in its test)


def g27_mesh_gears():
    """Mesh's gears (mesh.py:1189-1330, 2232-2413): lazily shared vertex arrays, offsets that fall back from gear to gear, masked and
    unmasked translations / fields / affine maps, the locked mesh that ignores them; plus areas, deformation measures, bounds,
    connectivity (vertex- and edge-wise, on a mesh with a bow tie) and a sub-mesh"""
    out = {}
    gears = dict(i=const.MESH_GEAR_INITIAL, f=const.MESH_GEAR_FIXED, m=const.MESH_GEAR_MOVING, s=const.MESH_GEAR_STAGING)

    def record(tag, m):
        for g, gear in gears.items():
            out[f'{tag}_{g}_vo'] = np.array(m.vertices_w_offset(gear=gear))
            out[f'{tag}_{g}_off'] = np.array(m.offset(gear=gear), dtype=np.float64).reshape(1, 2)
        out[f'{tag}_est'] = np.asarray(m.estimate_translation(), dtype=np.float64)
        out[f'{tag}_est_fs'] = np.asarray(m.estimate_translation(gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_STAGING)), dtype=np.float64)
    m, v, tri, mask = g27_mesh_walk(Mesh, const, record)
    out['v'], out['tri'], out['mask'] = v, tri, mask
    for g, gear in gears.items():
        out[f'areas_{g}'] = m.triangle_areas(gear=gear)
        out[f'bbox_{g}'] = np.asarray(m.bbox(gear=gear)); out[f'bbox_{g}_raw'] = np.asarray(m.bbox(gear=gear, offsetting=False))
    out['area_deform'] = m.triangle_area_deform(gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_MOVING))
    out['edge_deform'] = m.triangle_edge_deform(gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_MOVING))
    # connectivity: two grids joined by ONE shared vertex (a bow tie), and a third apart
    vb = np.array([[0, 0], [10, 0], [0, 10], [10, 10], [20, 10], [10, 20], [20, 20], [40, 0], [50, 0], [40, 10]], dtype=np.float64)
    tb = np.array([[0, 1, 2], [1, 3, 2], [3, 4, 5], [4, 6, 5], [7, 8, 9]], dtype=np.int32)
    mb = Mesh(vb, tb, uid=7)
    nv, lv = mb.connected_vertices()
    nt, lt = mb.connected_triangles()
    out['bow_v'], out['bow_t'] = vb, tb
    out['bow_nv'], out['bow_lv'], out['bow_nt'], out['bow_lt'] = np.int64(nv), np.asarray(lv, dtype=np.int64), np.int64(nt), np.asarray(lt, dtype=np.int64)
    parts = mb.divide_disconnected_mesh()
    out['bow_part_uids'] = np.array([p.uid for p in parts], dtype=np.float64)
    for k, p in enumerate(parts):
        out[f'bow_part{k}_v'] = p.vertices_w_offset(gear=const.MESH_GEAR_INITIAL); out[f'bow_part{k}_t'] = np.asarray(p.triangles, dtype=np.int64)
    tmask = np.zeros(m.num_triangles, dtype=bool); tmask[::3] = True
    sub = m.submesh(tmask, uid=3.5)
    out['sub_tmask'] = tmask
    out['sub_t'] = np.asarray(sub.triangles, dtype=np.int64)
    for g, gear in gears.items():
        out[f'sub_{g}_vo'] = np.array(sub.vertices_w_offset(gear=gear))
    np.savez_compressed(os.path.join(OUT, 'g27_mesh_gears.npz'), **out)


# ----------------------------------------------------------------------- G28
def g28_affine_cascade():
    """SLM.optimize_affine_cascade (optimizer.py:1128-1189) on a multi-tile system: the order in which the free tiles are placed (most
    link weight to placed tiles first), the fit of each onto its placed neighbours through the links (rigid / clipped / affine), the
    gears it reads and writes, a tile no link reaches (tests/golden/walks.py::g28_cascade_walk drives reference and product alike)"""
    from walks import g28_cascade_walk
    g15 = dict(np.load(os.path.join(OUT, 'g15_translation.npz')))
    out = {}

    def record(tag, ms, modified):
        out[f'{tag}_modified'] = np.bool_(modified)
        for g_, gear in (('f', const.MESH_GEAR_FIXED), ('m', const.MESH_GEAR_MOVING)):
            out[f'{tag}_{g_}'] = np.stack([m.vertices_w_offset(gear=gear) for m in ms])
    g28_cascade_walk(Mesh, optimizer.Link, optimizer.SLM, const, g15, record)
    np.savez_compressed(os.path.join(OUT, 'g28_affine_cascade.npz'), **out)


# ----------------------------------------------------------------------- G29
def g29_cartesian_grid():
    """Mesh.from_bbox(cartesian=True) (mesh.py:403-435): the node grid it hands to the mesher (vertices and the grid segments; the
    triangles come from `triangle`, which is absent) for a sweep of boxes, mesh sizes and minimum block counts -- Mesh.from_PSLG is
    replaced by a recorder"""
    cases = []
    for (W, H) in ((510, 4096), (4096, 510), (120, 1536), (122, 1526), (255, 2048), (60, 60), (700, 90), (301, 299), (64, 1000)):
        for ms in (25.0, 40.0, 75.0, 200.0):
            for mnb in (1, 2, 3):
                cases.append((0, 0, W, H, ms, mnb))
    cases += [(-7.5, 3.25, 500.25, 911.0, 60.0, 2), (10, 20, 30, 40, 100.0, 1)]
    real = Mesh.from_PSLG
    got = {}

    def recorder(cls, vertices, segments, **kwargs):
        got['v'] = np.array(vertices); got['seg'] = np.array(segments); got['mesh_size'] = float(kwargs['mesh_size'])
        return None
    out = dict(cases=np.array(cases, dtype=np.float64))
    try:
        Mesh.from_PSLG = classmethod(recorder)
        for k, (x0, y0, x1, y1, ms, mnb) in enumerate(cases):
            Mesh.from_bbox((x0, y0, x1, y1), cartesian=True, mesh_size=ms, min_num_blocks=int(mnb))
            out[f'c{k}_v'] = got['v']; out[f'c{k}_nseg'] = np.int64(got['seg'].shape[0]); out[f'c{k}_ms'] = np.float64(got['mesh_size'])
    finally:
        Mesh.from_PSLG = real
    np.savez_compressed(os.path.join(OUT, 'g29_cartesian_grid.npz'), **out)


# ----------------------------------------------------------------------- G30
def g30_seeded_loop():
    """iterative_xcorr_matcher_w_mesh WITH initial matches (matcher.py:552-563: a link from the matches, optimize_affine_cascade at the
    FIXED gear, rigid anneal of the MOVING gear, one relaxation) and then the loop of G23: section 1 comes in its own frame (rotated by
    0.02 rad and shifted by (40, -25) px), the initial matches say so, the scripted block matches refine.  Meshes of G23.  Solves made
    converged and deterministic like there (the seed's `precondition='smoothed_aggregation'` asks for pyamg, absent: dropped -- the fixed
    point does not depend on the preconditioner)."""
    g23 = np.load(os.path.join(OUT, 'g23_matcher_loop.npz'))
    va, ta, vb, tb = g23['v0'], g23['t0'], g23['v1'], g23['t1']
    real_match = matcher.bboxes_mesh_renderer_matcher
    real_ol = optimizer.SLM.optimize_linear

    def converged(self, **kw):
        kw['tol'] = 1e-11; kw['tolerated_perturbation'] = None; kw['callback_settings'] = {'chances': None, 'eval_step': 10}
        kw['check_converge'] = True
        kw.pop('precondition', None)
        return real_ol(self, **kw)
    rng = np.random.default_rng(3030)
    th = 0.02
    Rm = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
    shift = np.array([40.0, -25.0])
    vb1 = vb @ Rm + shift                                  # section 1 in its own frame
    p = np.stack((rng.uniform(100, 1400, 60), rng.uniform(100, 1000, 60)), axis=-1)
    ixy0 = p + rng.normal(0, 0.8, p.shape)
    ixy1 = p @ Rm + shift
    iw = rng.uniform(0.3, 1.0, 60)
    out = dict(v1=vb1, ixy0=ixy0, ixy1=ixy1, iw=iw, spacings=np.array([400.0, 100.0]))
    try:
        optimizer.SLM.optimize_linear = converged
        m0 = Mesh(va, ta, uid=0)
        m0.lock()
        m1 = Mesh(vb1.copy(), tb, uid=1)
        rounds = []

        def scripted(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **kw):
            k = len(rounds)
            rounds.append(dict(bboxes0=np.array(bboxes0), bboxes1=np.array(bboxes1), pad=bool(kw.get('pad')), subpixel=bool(kw.get('subpixel')),
                               moving1=mesh1.vertices_w_offset(gear=const.MESH_GEAR_MOVING), fixed1=mesh1.vertices_w_offset(gear=const.MESH_GEAR_FIXED)))
            return scripted_block_matches(k, bboxes0, bboxes1, 5.0)
        matcher.bboxes_mesh_renderer_matcher = scripted
        xy0, xy1, wt, strain = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, None, None, spacings=np.array([400.0, 100.0]), distributor='cartesian_bbox',
                                                                      conf_thresh=0.3, residue_len=3.0, residue_mode='huber', compute_strain=False,
                                                                      stiffness_lambda=0.5, min_num_blocks=2, initial_matches=common.Match(ixy0, ixy1, iw))
        out['nrounds'] = np.int64(len(rounds))
        for k, r in enumerate(rounds):
            for key in ('bboxes0', 'bboxes1', 'moving1', 'fixed1'):
                out[f'r{k}_{key}'] = r[key]
            out[f'r{k}_flags'] = np.array([r['pad'], r['subpixel']])
        out['xy0'] = xy0; out['xy1'] = xy1; out['weight'] = np.asarray(wt)
        out['moving1_final'] = m1.vertices_w_offset(gear=const.MESH_GEAR_MOVING)
    finally:
        matcher.bboxes_mesh_renderer_matcher = real_match
        optimizer.SLM.optimize_linear = real_ol
    np.savez_compressed(os.path.join(OUT, 'g30_seeded_loop.npz'), **out)


# ----------------------------------------------------------------------- G31
G31_CASES = {
    'decay': dict(spacings=[600.0, 150.0, 40.0], residue_mode='huber', residue_len=3.0, link_weight_decay=0.5),
    'enlarge': dict(spacings=[30.0, 12.0], residue_mode='huber', residue_len=3.0, allow_enlarge=True),
    'skip_shrink': dict(spacings=[600.0, 300.0, 150.0, 75.0], residue_mode='threshold', residue_len=4.0, max_spacing_skip=1, shrink_factor=0.7, pad=False),
    'dwell': dict(spacings=[200.0, 60.0], residue_mode='huber', residue_len=2.0, allow_dwell=1, subpixel=True, min_num_blocks=3),
}


def g31_loop_options():
    """the loop of G23 under the keywords that change its course: link_weight_decay (the links of earlier rounds stay, decayed),
    allow_enlarge (a first round whose displacement outruns the largest spacing is repeated with larger blocks), max_spacing_skip with a
    shrink factor and fixed padding, allow_dwell with a fixed sub-pixel flag and three blocks minimum"""
    import json
    g23 = np.load(os.path.join(OUT, 'g23_matcher_loop.npz'))
    va, ta, vb, tb = g23['v0'], g23['t0'], g23['v1'], g23['t1']
    real_match = matcher.bboxes_mesh_renderer_matcher
    real_ol = optimizer.SLM.optimize_linear

    def converged(self, **kw):
        kw['tol'] = 1e-11; kw['tolerated_perturbation'] = None; kw['callback_settings'] = {'chances': None, 'eval_step': 10}
        kw['check_converge'] = True
        return real_ol(self, **kw)
    out = dict(cases=np.array(json.dumps(G31_CASES)))
    try:
        optimizer.SLM.optimize_linear = converged
        for name, cs in G31_CASES.items():
            m0 = Mesh(va, ta, uid=0)
            m0.apply_translation((2.0, -1.0), const.MESH_GEAR_FIXED)
            m0.lock()
            m1 = Mesh(vb.copy(), tb, uid=1)
            rounds = []

            def scripted(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **kw):
                k = len(rounds)
                rounds.append(dict(bboxes0=np.array(bboxes0), bboxes1=np.array(bboxes1), pad=bool(kw.get('pad')), subpixel=bool(kw.get('subpixel')),
                                   field1=mesh1.vertices_w_offset(gear=const.MESH_GEAR_MOVING) - mesh1.vertices_w_offset(gear=const.MESH_GEAR_INITIAL)))
                return scripted_block_matches(k, bboxes0, bboxes1, 7.0)
            matcher.bboxes_mesh_renderer_matcher = scripted
            kw = dict(cs); kw['spacings'] = np.array(cs['spacings'])
            xy0, xy1, wt, strain = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, None, None, distributor='cartesian_bbox', conf_thresh=0.3, compute_strain=False,
                                                                          stiffness_lambda=0.5, **kw)
            out[f'{name}_nrounds'] = np.int64(len(rounds))
            for k, r in enumerate(rounds):
                for key in ('bboxes0', 'bboxes1', 'field1'):
                    out[f'{name}_r{k}_{key}'] = r[key]
                out[f'{name}_r{k}_flags'] = np.array([r['pad'], r['subpixel']])
            out[f'{name}_xy0'] = xy0; out[f'{name}_xy1'] = xy1; out[f'{name}_weight'] = np.asarray(wt)
            out[f'{name}_field1_final'] = m1.vertices_w_offset(gear=const.MESH_GEAR_MOVING) - m1.vertices_w_offset(gear=const.MESH_GEAR_INITIAL)
    finally:
        matcher.bboxes_mesh_renderer_matcher = real_match
        optimizer.SLM.optimize_linear = real_ol
    np.savez_compressed(os.path.join(OUT, 'g31_loop_options.npz'), **out)


# ----------------------------------------------------------------------- G32
G32_CASES = {
    'online_huber': dict(online_anneal=True, max_newtonstep=4, tol=1e-8, residue_mode='huber', residue_len=2.0, stiffness_lambda=0.5),
    'online_threshold': dict(online_anneal=True, max_newtonstep=3, tol=1e-6, residue_mode='threshold', residue_len=3.0, stiffness_lambda=1.0, crosslink_lambda=2.0),
    'plain_steps': dict(_newton=True, max_newtonstep=3, tol=1e-7, stiffness_lambda=[2.0, 1.0, 0.5], crosslink_lambda=-1.0),
    'rigid_anneal': dict(_newton=True, max_newtonstep=3, tol=1e-7, anneal_mode=2, residue_mode='huber', residue_len=[4.0, 1.5], stiffness_lambda=0.5),
}


def g32_inputs():
    g23 = np.load(os.path.join(OUT, 'g23_matcher_loop.npz'))
    rng = np.random.default_rng(3232)
    p = np.stack((rng.uniform(60, 1500, 220), rng.uniform(60, 1080, 220)), axis=-1)
    u = np.stack((5.0 * np.sin(p[:, 1] / 300.0) + 2.0 * (p[:, 0] / 1500.0), 4.0 * np.cos(p[:, 0] / 260.0)), axis=-1)
    q = p + u + rng.normal(0, 0.2, p.shape)
    out_ = rng.random(220) > 0.93
    q[out_] += np.array([9.0, -6.0])
    return g23['v0'], g23['t0'], g23['v1'], g23['t1'], p, q, rng.uniform(0.3, 1.0, 220)


def g32_newton_driver():
    """SLM.optimize_Newton_Raphson / optimize_elastic(online_anneal=True) (optimizer.py:1440-1555) as a DRIVER, on linear meshes: the
    per-step ladders (tolerances, lambdas, residue lengths), the annealing of the resting shape from the STAGING gear with its
    relax_higly_deformed, the residue re-weighting between steps, the cost floor and the early last step -- inner solves converged"""
    import json
    va, ta, vb, tb, p, q, w = g32_inputs()
    real_ol = optimizer.SLM.optimize_linear

    def converged(self, **kw):
        kw['tol'] = 1e-12; kw['atol'] = 0.0; kw['maxiter'] = None
        kw['tolerated_perturbation'] = None; kw['callback_settings'] = {'chances': None, 'eval_step': 10}
        kw['check_converge'] = True
        return real_ol(self, **kw)
    out = dict(cases=np.array(json.dumps(G32_CASES)), xy0=p, xy1=q, w=w)
    gears = dict(i=const.MESH_GEAR_INITIAL, f=const.MESH_GEAR_FIXED, m=const.MESH_GEAR_MOVING, s=const.MESH_GEAR_STAGING)
    try:
        optimizer.SLM.optimize_linear = converged
        for name, kw in G32_CASES.items():
            m0 = Mesh(va, ta, uid=0)
            m0.apply_translation((2.0, -1.0), const.MESH_GEAR_FIXED)
            m0.lock()
            m1 = Mesh(vb.copy(), tb, uid=1)
            opt = optimizer.SLM([m0, m1], stiffness_lambda=0.7)
            opt.add_link_from_coordinates(0, 1, p, q, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_INITIAL), weight=w, check_duplicates=False)
            kw_ = dict(kw)
            cost = opt.optimize_Newton_Raphson(**kw_) if kw_.pop('_newton', False) else opt.optimize_elastic(**kw_)
            out[f'{name}_cost'] = np.array([np.nan if c is None else c for c in cost], dtype=np.float64)
            for g_, gear in gears.items():
                out[f'{name}_{g_}'] = m1.vertices_w_offset(gear=gear)
            lk = opt.links[0]
            out[f'{name}_lw'] = np.asarray(lk.weight(use_mask=False), dtype=np.float64)
    finally:
        optimizer.SLM.optimize_linear = real_ol
    np.savez_compressed(os.path.join(OUT, 'g32_newton_driver.npz'), **out)


# ----------------------------------------------------------------------- G33
def g33_render_weights():
    """render weights of materials in the matcher (material.py:27-30, mesh.py:1836-1859, 2168-2170, optimizer.py:59): mesh 1 of G23 with a
    band of a material that weighs 1e-3 in rendering (like the soft / wrinkle entries of the default material table).  (a) the masks and
    per-triangle weights; (b) Link.from_coordinates with its default threshold 0.1, with 0 and with 0.5e-3: which matches survive; (c) the
    loop of G23 with render_weight_threshold = 0.1: matches that land in the band are dropped round after round."""
    g23 = np.load(os.path.join(OUT, 'g23_matcher_loop.npz'))
    va, ta, vb, tb = g23['v0'], g23['t0'], g23['v1'], g23['t1']
    ctr = vb[tb].mean(axis=1)
    mids = np.where((ctr[:, 0] > 600) & (ctr[:, 0] < 900), 5, 0).astype(np.int16)
    tab = {'default': dict(material.MATERIAL_DEFAULT), 'soft_look': {'uid': 5, 'render_weight': 1.0e-3},
           'hidden': {'uid': 6, 'render': False}}
    mids[(ctr[:, 1] > 1000) & (ctr[:, 0] < 300)] = 6

    def mesh1():
        return Mesh(vb.copy(), tb.copy(), material_table=material.MaterialTable(table=tab), material_ids=mids.copy(), uid=1)
    m1 = mesh1()
    out = dict(t1=np.asarray(m1.triangles), mids=np.asarray(m1._material_ids, dtype=np.int32))
    out['weights'] = m1.weight_multiplier_for_render()
    for thr in (0.0, 0.1, 1.0e-3, 0.5e-3):
        # (a fresh mesh per threshold: the reference caches the first mask a mesh computes, whatever threshold later calls name)
        out[f'mask_{thr}'] = mesh1().triangle_mask_for_render(render_weight_threshold=thr)
    rng = np.random.default_rng(3333)
    p = np.stack((rng.uniform(30, 1500, 300), rng.uniform(30, 1100, 300)), axis=-1)
    q = p + rng.normal(0, 1.0, p.shape)
    w = rng.uniform(0.2, 1.0, 300)
    out.update(lp=p, lq=q, lw=w)
    m0 = Mesh(va, ta, uid=0)
    for tag, kw in (('default', {}), ('zero', dict(render_weight_threshold=0)), ('low', dict(render_weight_threshold=0.5e-3))):
        lk, mask = optimizer.Link.from_coordinates(m0, m1, p, q, weight=w, **kw)
        out[f'link_{tag}_mask'] = mask
        out[f'link_{tag}_xy1'] = lk.xy1(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True)
    real_match = matcher.bboxes_mesh_renderer_matcher
    real_ol = optimizer.SLM.optimize_linear

    def converged(self, **kw):
        kw['tol'] = 1e-11; kw['tolerated_perturbation'] = None; kw['callback_settings'] = {'chances': None, 'eval_step': 10}
        kw['check_converge'] = True
        return real_ol(self, **kw)
    try:
        optimizer.SLM.optimize_linear = converged
        m0 = Mesh(va, ta, uid=0)
        m0.apply_translation((2.0, -1.0), const.MESH_GEAR_FIXED)
        m0.lock()
        m1 = mesh1()
        rounds = []

        def scripted(mesh0, mesh1_, ld0, ld1, bboxes0, bboxes1, **kw):
            k = len(rounds)
            rounds.append(dict(bboxes0=np.array(bboxes0), rwt=float(kw.get('render_weight_threshold', -1)),
                               field1=mesh1_.vertices_w_offset(gear=const.MESH_GEAR_MOVING) - mesh1_.vertices_w_offset(gear=const.MESH_GEAR_INITIAL)))
            return scripted_block_matches(k, bboxes0, bboxes1, 9.0)
        matcher.bboxes_mesh_renderer_matcher = scripted
        xy0, xy1, wt, strain = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, None, None, spacings=np.array([400.0, 100.0]), distributor='cartesian_bbox',
                                                                      conf_thresh=0.3, residue_len=3.0, residue_mode='huber', compute_strain=False,
                                                                      stiffness_lambda=0.5, min_num_blocks=2, render_weight_threshold=0.1)
        out['nrounds'] = np.int64(len(rounds))
        for k, r in enumerate(rounds):
            out[f'r{k}_bboxes0'] = r['bboxes0']; out[f'r{k}_field1'] = r['field1']; out[f'r{k}_rwt'] = np.float64(r['rwt'])
        out['xy0'] = xy0; out['xy1'] = xy1; out['weight'] = np.asarray(wt)
    finally:
        matcher.bboxes_mesh_renderer_matcher = real_match
        optimizer.SLM.optimize_linear = real_ol
    np.savez_compressed(os.path.join(OUT, 'g33_render_weights.npz'), **out)


# ----------------------------------------------------------------------- G34
def g34_links_to_divided_meshes():
    """SLM.add_link_from_coordinates (optimizer.py:637-684) AFTER a mesh has been divided into its connected parts: a link addressed to the
    parent uid is dealt to the parts that hold its matches (every match once with submesh_exclusive, else wherever it lands), a link whose
    name is already loaded is skipped (check_duplicates), a uid the system does not know adds nothing.  Meshes and matches of G26."""
    (v0, t0), (v1, t1), (v2, t2), pts, d2 = g26_inputs()
    out = {}
    m0 = Mesh(v0, t0, uid=0); m0.lock()
    m1 = Mesh(v1, t1, uid=1)
    m2 = Mesh(v2, t2, uid=2)
    opt = optimizer.SLM([m0, m1, m2])
    out['divided'] = np.bool_(opt.divide_disconnected_submeshes())
    out['mesh_uids'] = np.array([m.uid for m in opt.meshes], dtype=np.float64)
    res = []
    res.append(opt.add_link_from_coordinates(0, 2, pts['02'][0], pts['02'][1], weight=pts['02'][2], name='a'))
    res.append(opt.add_link_from_coordinates(1, 2, pts['12'][0], pts['12'][1], weight=pts['12'][2], name='b', submesh_exclusive=False))
    res.append(opt.add_link_from_coordinates(0, 2, pts['02'][0], pts['02'][1], weight=pts['02'][2], name='a'))                    # loaded already
    res.append(opt.add_link_from_coordinates(0, 2, pts['02'][0], pts['02'][1], weight=pts['02'][2], name='a', check_duplicates=False))
    res.append(opt.add_link_from_coordinates(0, 7, pts['02'][0], pts['02'][1], weight=pts['02'][2]))                              # no such mesh
    res.append(opt.add_link_from_coordinates(2.1, 1, pts['12'][1], pts['12'][0], weight=pts['12'][2]))                            # a part addressed itself
    out['added'] = np.array(res)
    out['nlinks'] = np.int64(len(opt.links))
    for name in ('02', '12'):
        out[f'p{name}_xy0'], out[f'p{name}_xy1'], out[f'p{name}_w'] = pts[name]
    for k, lk in enumerate(opt.links):
        out[f'l{k}_uids'] = np.array(lk.uids, dtype=np.float64)
        out[f'l{k}_xy0'] = lk.xy0(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True)
        out[f'l{k}_xy1'] = lk.xy1(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True)
        out[f'l{k}_w'] = np.asarray(lk.weight(use_mask=False), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'g34_links_to_divided_meshes.npz'), **out)


# ----------------------------------------------------------------------- G35
def g35_outcasts():
    """SLM.flag_outcasts (optimizer.py:1604-1625): the meshes the aligner leaves out of a window (aligner.py:700)"""
    from walks import g35_outcast_walk
    g15 = dict(np.load(os.path.join(OUT, 'g15_translation.npz')))
    out = {}

    def record(tag, flags, ms):
        out[tag] = np.asarray(flags, dtype=bool)
        out[tag + '_kept'] = np.array([bool(m.is_outcast) for m in ms])
    g35_outcast_walk(Mesh, optimizer.Link, optimizer.SLM, g15, record)
    np.savez_compressed(os.path.join(OUT, 'g35_outcasts.npz'), **out)


# ----------------------------------------------------------------------- G36
G36_CASES = {
    'late_silence': dict(spacings=[400.0, 100.0, 40.0], silent_from=2),       # the last round finds nothing confident: the links of round 1 stand
    'early_silence': dict(spacings=[400.0, 100.0], silent_from=0),            # the first round finds nothing: no result
    'apart': dict(spacings=[400.0, 100.0], silent_from=99, shift1=(5000.0, 0.0)),   # the meshes do not overlap
    'tiny_motion': dict(spacings=[400.0, 100.0], silent_from=99, amp=0.004),  # displacements below 0.1 px: linked, never relaxed
}


def g36_loop_exits():
    """the ways out of the loop (matcher.py:592-598, 671-679, 719-724, 744-751): a round without a confident block after earlier rounds
    linked (break: the earlier links are the result) or before anything was linked (no result), meshes that do not overlap, and matches
    so small that no relaxation is run (max_dis <= 0.1)"""
    import json
    g23 = np.load(os.path.join(OUT, 'g23_matcher_loop.npz'))
    va, ta, vb, tb = g23['v0'], g23['t0'], g23['v1'], g23['t1']
    real_match = matcher.bboxes_mesh_renderer_matcher
    real_ol = optimizer.SLM.optimize_linear
    solves = []

    def converged(self, **kw):
        solves.append(1)
        kw['tol'] = 1e-11; kw['tolerated_perturbation'] = None; kw['callback_settings'] = {'chances': None, 'eval_step': 10}
        kw['check_converge'] = True
        return real_ol(self, **kw)
    out = dict(cases=np.array(json.dumps(G36_CASES)))
    try:
        optimizer.SLM.optimize_linear = converged
        for name, cs in G36_CASES.items():
            m0 = Mesh(va, ta, uid=0)
            m0.lock()
            m1 = Mesh(vb.copy() + np.array(cs.get('shift1', (0.0, 0.0))), tb, uid=1)
            rounds = []
            del solves[:]

            def scripted(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **kw):
                k = len(rounds)
                rounds.append(np.array(bboxes0))
                xy0, xy1, conf = scripted_block_matches(k, bboxes0, bboxes1, 11.0)
                if 'amp' in cs:
                    xy1 = xy0 + (xy1 - xy0) * (cs['amp'] / 6.0)
                if k >= cs['silent_from']:
                    conf = conf * 0.1
                return xy0, xy1, conf
            matcher.bboxes_mesh_renderer_matcher = scripted
            xy0, xy1, wt, strain = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, None, None, spacings=np.array(cs['spacings']), distributor='cartesian_bbox',
                                                                          conf_thresh=0.3, residue_len=3.0, residue_mode='huber', compute_strain=False,
                                                                          stiffness_lambda=0.5, min_num_blocks=2)
            out[f'{name}_nrounds'] = np.int64(len(rounds))
            out[f'{name}_nsolves'] = np.int64(len(solves))
            out[f'{name}_none'] = np.bool_(xy0 is None)
            if xy0 is not None:
                out[f'{name}_xy0'] = xy0; out[f'{name}_xy1'] = xy1; out[f'{name}_weight'] = np.asarray(wt)
            else:
                out[f'{name}_wt'] = np.float64(wt)
            out[f'{name}_strain'] = np.float64(strain)
            out[f'{name}_field1_final'] = m1.vertices_w_offset(gear=const.MESH_GEAR_MOVING) - m1.vertices_w_offset(gear=const.MESH_GEAR_INITIAL)
    finally:
        matcher.bboxes_mesh_renderer_matcher = real_match
        optimizer.SLM.optimize_linear = real_ol
    np.savez_compressed(os.path.join(OUT, 'g36_loop_exits.npz'), **out)


# ----------------------------------------------------------------------- G37
def g37_section_matcher_call():
    """section_matcher (matcher.py:370-396) down to its call of the loop: the defaults it fills in (sigma 2.5, batch_size 100, distributor
    'cartesian_region', link_weight_decay 0, render_weight_threshold 0.1, stiffness_lambda 0.5), what it passes through, and the sub-meshes
    it hands over after dropping the triangles of materials softer than stiffness_multiplier_threshold -- the loop replaced by a recorder"""
    import json
    g23 = np.load(os.path.join(OUT, 'g23_matcher_loop.npz'))
    vb, tb = g23['v1'], g23['t1']
    ctr = vb[tb].mean(axis=1)
    mids = np.where(ctr[:, 0] > 1200, 5, 0).astype(np.int16)
    tab = {'default': dict(material.MATERIAL_DEFAULT), 'jelly': {'uid': 5, 'stiffness_multiplier': 0.05}}
    real = matcher.iterative_xcorr_matcher_w_mesh
    seen = {}

    def recorder(mesh0, mesh1, ld0, ld1, **kw):
        seen['kw'] = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in kw.items()}
        seen['ntri'] = [int(mesh0.num_triangles), int(mesh1.num_triangles)]
        seen['loaders'] = [ld0, ld1]
        return np.zeros((1, 2)), np.ones((1, 2)), np.ones(1), 0.07
    out = {}
    try:
        matcher.iterative_xcorr_matcher_w_mesh = recorder
        for tag, kw in (('defaults', {}), ('given', dict(spacings=[200, 50], conf_thresh=0.4, sigma=3.5, stiffness_multiplier_threshold=0.0, compute_strain=True,
                                                          residue_len=-2, shrink_factor=0.7, distributor='cartesian_bbox', stiffness_lambda=0.25))):
            m0 = Mesh(g23['v0'], g23['t0'], uid=0)
            m1 = Mesh(vb.copy(), tb.copy(), material_table=material.MaterialTable(table=tab), material_ids=mids.copy(), uid=1)
            res = matcher.section_matcher(m0, m1, 'loader0', 'loader1', **kw)
            out[f'{tag}_kw'] = np.array(json.dumps(seen['kw'], sort_keys=True))
            out[f'{tag}_ntri'] = np.array(seen['ntri'])
            out[f'{tag}_strain'] = np.float64(res[3])
    finally:
        matcher.iterative_xcorr_matcher_w_mesh = real
    out['mids'] = np.asarray(Mesh(vb.copy(), tb.copy(), material_table=material.MaterialTable(table=tab), material_ids=mids.copy(), uid=1)._material_ids, dtype=np.int32)
    out['t1'] = np.asarray(Mesh(vb.copy(), tb.copy(), material_table=material.MaterialTable(table=tab), material_ids=mids.copy(), uid=1).triangles)
    np.savez_compressed(os.path.join(OUT, 'g37_section_matcher_call.npz'), **out)


if __name__ == '__main__':
    for fn in (g1_xcorr, g2_dog, g3_global, g45_stiffness, g6789_system, g10_elements, g11_bbox, g12_mixed_materials, g13_strain, g14_groupings, g15_translation, g16_relax, g17_newton, g18_locked_neighbours, g19_area_stretch, g20_xcorr_normalized, g21_grouped_dof, g22_schedule_walks, g23_matcher_loop, g24_strip_loop, g25_overlap_bookkeeping, g26_slm_bookkeeping, g27_mesh_gears, g28_affine_cascade, g29_cartesian_grid, g30_seeded_loop, g31_loop_options, g32_newton_driver, g33_render_weights, g34_links_to_divided_meshes, g35_outcasts, g36_loop_exits, g37_section_matcher_call):
        if len(sys.argv) > 1 and fn.__name__ not in sys.argv[1:]:
            continue
        fn()
        print('wrote', fn.__name__)
