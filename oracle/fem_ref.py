"""Oracle (CPU restatement) of the FEM relaxation path.  TEST INFRASTRUCTURE --
see ``oracle/__init__.py``; never imported by the product package.

Restates, with scipy.sparse like the reference, the element maths of
material.py, the assembly of mesh.py:2893-3083, the cross-link terms and the
linear system of optimizer.py:802-901,1257-1437,1573-1590, and the solver
contract of optimizer.py:1945-2080.  Mixed precision follows the reference:
vertices/shape matrix float64, D float32, cross-link matrix float32, stress
cast to float32, final system float64.
"""
import numpy as np
from scipy import sparse
from scipy.sparse import csgraph, linalg as spla

GEAR_INITIAL, GEAR_FIXED, GEAR_MOVING, GEAR_STAGING = -1, 0, 1, 2   # constant.py:6-10
MODEL_ENG, MODEL_SVK, MODEL_NHK = 0, 1, 2                            # constant.py:34-36


def cross2d(a, b):
    """common.py:895-896."""
    return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]


# ------------------------------------------------------------------ element maths
def eng_shape_matrix(tripts, T, num_dof):
    """material.py:134-159: N (3T x num_dof) CSR float64; rows exx, eyy, gxy."""
    tripts = np.asarray(tripts, dtype=np.float64).reshape(-1, 3, 2)
    T = np.asarray(T).reshape(-1, 3)
    e = np.roll(tripts, -1, axis=-2) - np.roll(tripts, 1, axis=-2)     # e_i = p_{i+1} - p_{i-1}
    nt = tripts.shape[0]
    a = np.abs(cross2d(e[:, 0, :], e[:, 1, :])).reshape(-1, 1, 1)       # 2*area
    e = e / (a ** 0.5)
    ty = e[:, :, 1]
    tx = -e[:, :, 0]
    cols = np.concatenate((2 * T, 2 * T + 1, 2 * T, 2 * T + 1), axis=-1)
    vals = np.concatenate((ty, tx, tx, ty), axis=-1)
    indptr = np.cumsum(np.insert(np.tile([3, 3, 6], nt), 0, 0))
    return sparse.csr_matrix((vals.ravel(), cols.ravel(), indptr), shape=(3 * nt, num_dof))


def eng_stiffness_from_shape(N, multiplier=None, nu=0.0, mat_multiplier=1.0, stretch_factor=None):
    """material.py:162-182: K = N^T D N, D float32 per triangle."""
    nt = N.shape[0] // 3
    if mat_multiplier == 0 or nt == 0:
        return None
    m = np.full(nt, mat_multiplier, dtype=np.float32)
    if multiplier is not None:
        m = multiplier * m
    if stretch_factor is not None:
        m = m * np.asarray(stretch_factor).ravel()
    d0 = (m.reshape(-1, 1) * np.array([1, 1, (1 - nu) / 2])).ravel()
    if nu == 0:
        D = sparse.diags(d0, dtype=np.float32)
    else:
        d1 = (m.reshape(-1, 1) * np.array([nu, 0, 0])).ravel()[:-1]
        D = sparse.diags([d1, d0, d1], [-1, 0, 1], dtype=np.float32)
    return N.T @ D @ N


def mesh_stiffness(v_shape, v_cur, triangles, tri_mult=None, nu=0.0, mat_multiplier=1.0):
    """mesh.py:3058-3083 for a mesh whose triangles all use one linear ENG
    material (the default table, default_material_table.yaml):
    K from the shape gear, stress = K (v_cur - v_shape) cast float32."""
    v_shape = np.asarray(v_shape, dtype=np.float64)
    T = np.asarray(triangles)
    ndof = 2 * v_shape.shape[0]
    N = eng_shape_matrix(v_shape[T], T, ndof)
    K = eng_stiffness_from_shape(N, multiplier=tri_mult, nu=nu, mat_multiplier=mat_multiplier)
    if v_cur is None or v_cur is v_shape:
        stress = np.zeros(ndof, dtype=np.float32)
    else:
        stress = K.dot((np.asarray(v_cur, dtype=np.float64) - v_shape).ravel()).astype(np.float32)
    return K, stress


def element_shape_B(tripts):
    """material.py:185-209: B (T x 4 x 6, float32) and a = 2*area (T x 1 x 1)."""
    tripts = np.asarray(tripts, dtype=np.float64).reshape(-1, 3, 2)
    e = np.roll(tripts, -1, axis=-2) - np.roll(tripts, 1, axis=-2)
    a = np.abs(cross2d(e[:, 0, :], e[:, 1, :])).reshape(-1, 1, 1)
    e = e / a
    B = np.zeros((tripts.shape[0], 4, 6), dtype=np.float32)
    t0 = e[:, :, 1]
    t1 = -e[:, :, 0]
    B[:, 0, 0::2] = t0
    B[:, 1, 0::2] = t1
    B[:, 2, 1::2] = t0
    B[:, 3, 1::2] = t1
    return B, a


def element_stiffness(B, areas, uv, model, nu=0.0):
    """material.py:212-309 (without stiffness_func): tangent K_e (T x 6 x 6)
    and internal force P_e (T x 6 x 1), float32 pipeline like the reference."""
    f32 = np.float32
    nt = B.shape[0]
    uv = np.asarray(uv).astype(f32).reshape(-1, 6, 1)
    D = np.eye(3, dtype=f32)
    D[[0, 1], [1, 0]] = nu
    D[-1, -1] = (1 - nu) / 2
    sel = np.array([[1, 0, 0, 0], [0, 0, 0, 1], [0, 1, 1, 0]], dtype=f32)
    if model == MODEL_ENG:
        Bn = sel @ B
        K = areas * (np.swapaxes(Bn, 1, 2) @ D @ Bn)
        P = K @ uv
    elif model == MODEL_SVK:
        Ft = (B @ uv).reshape(-1, 2, 2) + np.eye(2, dtype=f32)
        FtT = np.swapaxes(Ft, 1, 2)
        Et = 0.5 * (FtT @ Ft - np.eye(2, dtype=f32))
        E = sel @ Et.reshape(-1, 4, 1)
        Bc = np.array([[1, 0, 1, 0], [0, 1, 0, 1]], dtype=f32) @ B
        Fc = np.tile(FtT, (1, 1, 3))
        Bn = np.concatenate((Bc * Fc, np.sum(Bc * Fc[:, ::-1, :], axis=1, keepdims=True)), axis=1)
        S = D @ E
        Sg = np.zeros((nt, 4, 4), dtype=f32)
        Sg[:, 0, 0] = S[:, 0, 0]; Sg[:, 2, 2] = S[:, 0, 0]
        Sg[:, 1, 1] = S[:, 1, 0]; Sg[:, 3, 3] = S[:, 1, 0]
        Sg[:, 0, 1] = S[:, 2, 0]; Sg[:, 1, 0] = S[:, 2, 0]
        Sg[:, 2, 3] = S[:, 2, 0]; Sg[:, 3, 2] = S[:, 2, 0]
        P = areas * (np.swapaxes(Bn, 1, 2) @ S)
        K = areas * (np.swapaxes(Bn, 1, 2) @ D @ Bn + np.swapaxes(B, 1, 2) @ Sg @ B)
    elif model == MODEL_NHK:
        Ft = (B @ uv).reshape(-1, 2, 2) + np.eye(2, dtype=f32)
        J = np.linalg.det(Ft).reshape(-1, 1, 1)
        U = np.array([[0, 0, 0, 1], [0, 0, -1, 0], [0, -1, 0, 0], [1, 0, 0, 0]], dtype=f32)
        F = Ft.reshape(-1, 4, 1)
        Fu = U @ F
        I4 = np.eye(4, dtype=f32)
        P = 0.5 * areas * (np.swapaxes(B, 1, 2) @ (I4 - U / J) @ F)
        K = 0.5 * areas * (np.swapaxes(B, 1, 2) @ (I4 - U / J + (Fu @ np.swapaxes(Fu, 1, 2)) / (J ** 2)) @ B)
    else:
        raise NotImplementedError
    return K, P


# ------------------------------------------------------------------ mesh state
class RefMesh:
    """Minimal state of mesh.py:212-293,1221-1325 needed on the FEM path:
    four gears of vertices + offsets with the reference's aliasing rules."""

    def __init__(self, vertices, triangles, uid=0, locked=False, soft_factor=1.0,
                 stiffness_multiplier=None, nu=0.0):
        v = np.asarray(vertices, dtype=np.float64).reshape(-1, 2)
        self.triangles = np.asarray(triangles).reshape(-1, 3)
        self._v = {GEAR_INITIAL: v, GEAR_FIXED: v, GEAR_MOVING: None, GEAR_STAGING: None}
        z = np.zeros((1, 2))
        self._off = {GEAR_INITIAL: z, GEAR_FIXED: z, GEAR_MOVING: z, GEAR_STAGING: z}
        self.uid = float(uid)
        self.locked = locked
        self.soft_factor = soft_factor
        self.nu = nu
        if stiffness_multiplier is None:
            stiffness_multiplier = np.ones(self.triangles.shape[0], dtype=np.float32)
        self.stiffness_multiplier = stiffness_multiplier

    @property
    def num_vertices(self):
        return self._v[GEAR_INITIAL].shape[0]

    def vertices(self, gear):                               # mesh.py:1221-1234,1286-1320
        if gear == GEAR_MOVING and self._v[gear] is None:
            return self._v[GEAR_FIXED]
        if gear == GEAR_STAGING and self._v[gear] is None:
            return self.vertices(GEAR_MOVING)
        return self._v[gear]

    def offset(self, gear):                                 # mesh.py:1256-1265
        if self._v[gear] is None:
            if gear == GEAR_MOVING:
                return self._off[GEAR_FIXED]
            return self.offset(GEAR_MOVING)
        return self._off[gear]

    def vertices_w_offset(self, gear):
        return self.vertices(gear) + self.offset(gear)

    def apply_translation(self, dxy, gear):                 # mesh.py:2272-2286 (unmasked)
        dxy = np.asarray(dxy, dtype=np.float64).reshape(1, 2)
        if self.locked or not np.any(dxy):
            return
        v = self.vertices(gear)
        off = self.offset(gear)
        self._v[gear] = v
        self._off[gear] = off + dxy

    def set_field(self, dxy, gear=(GEAR_FIXED, GEAR_MOVING)):   # mesh.py:2400-2413 (unmasked)
        if self.locked:
            return
        v0 = self.vertices(gear[0])
        off0 = self.offset(gear[0])
        m = np.mean(dxy.reshape(-1, 2), axis=0, keepdims=True)
        self._v[gear[-1]] = v0 + (dxy - m)
        self._off[gear[-1]] = off0 + m

    def set_affine(self, A, gear=(GEAR_FIXED, GEAR_MOVING)):     # mesh.py:2342-2354 (unmasked, gear[0] != gear[-1])
        if self.locked:
            return
        v0 = self.vertices(gear[0])
        off0 = self.offset(gear[0])
        self._v[gear[-1]] = v0 @ A[:-1, :-1]
        self._off[gear[-1]] = off0 @ A[:-1, :-1] + A[-1, :-1]

    def anneal_copy(self, gear=(GEAR_MOVING, GEAR_FIXED)):   # mesh.py:2452-2456
        if self.locked:
            return
        off0 = self.offset(gear[0])
        v0 = self.vertices(gear[0])
        self._v[gear[1]] = v0
        self._off[gear[1]] = off0

    # ---- masked state changes, rigid / affine annealing, deformation measures (the relax_mesh path)
    def set_vertices(self, v, gear, vtx_mask=None):         # mesh.py:2232-2243
        if self.locked:
            return
        if self._v[gear] is None:
            self._off[gear] = self.offset(gear)
        if vtx_mask is None:
            self._v[gear] = v
        else:
            cur = self.vertices(gear).copy()
            cur[vtx_mask] = v
            self._v[gear] = cur

    def set_offset(self, offset, gear):                     # mesh.py:2246-2249
        if not self.locked:
            self._off[gear] = offset

    def apply_field(self, dxy, gear, vtx_mask=None):        # mesh.py:2381-2397
        if self.locked or not np.any(dxy):
            return
        v0 = self.vertices(gear)
        off0 = self.offset(gear)
        if vtx_mask is None:
            m = np.mean(dxy.reshape(-1, 2), axis=0, keepdims=True)
            self.set_vertices(v0 + (dxy - m), gear)
            self.set_offset(off0 + m, gear)
        else:
            self.set_vertices(v0[vtx_mask] + dxy, gear, vtx_mask=vtx_mask)
            self.set_offset(off0, gear)

    def apply_affine(self, A, gear, vtx_mask=None):         # mesh.py:2324-2339
        if self.locked or np.all(A == np.eye(3)):
            return
        v0 = self.vertices(gear)
        off0 = self.offset(gear)
        if vtx_mask is None:
            self.set_vertices(v0 @ A[:-1, :-1], gear)
            self.set_offset(off0 @ A[:-1, :-1] + A[-1, :-1], gear)
        else:
            v1 = v0[vtx_mask] @ A[:-1, :-1] + off0 @ A[:-1, :-1] + A[-1, :-1] - off0
            self.set_vertices(v1, gear, vtx_mask=vtx_mask)
            self.set_offset(off0, gear)

    def connected_vertices(self):                           # mesh.py:1762-1780, 1644-1656
        t = self.triangles
        e = np.concatenate((t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]), axis=0)
        n = self.num_vertices
        A = sparse.csr_matrix((np.ones(e.shape[0], dtype=bool), (e[:, 0], e[:, 1])), shape=(n, n))
        return csgraph.connected_components(A, directed=False, return_labels=True)

    def anneal(self, gear=(GEAR_MOVING, GEAR_FIXED), mode=2):   # mesh.py:2421-2458; modes constant.py:27-31
        if self.locked:
            return
        if mode in (0, 1):
            v0 = self.vertices_w_offset(gear[0])
            v1 = self.vertices_w_offset(gear[1])
            if mode == 0:
                _, R = fit_affine(v0, v1, return_rigid=True)
                self.apply_affine(R, gear[1])
            else:
                self.apply_affine(fit_affine(v0, v1, return_rigid=False), gear[1])
        elif mode in (2, 3):
            n_conn, v_conn = self.connected_vertices()
            self.anneal(gear=gear, mode=0)
            if n_conn == 1 and mode == 2:
                return
            v0 = self.vertices_w_offset(gear[0])
            v1 = self.vertices_w_offset(gear[1])
            for cid in range(n_conn):
                idx = v_conn == cid
                if mode == 2:
                    _, R = fit_affine(v0[idx], v1[idx], return_rigid=True)
                    self.apply_affine(R, gear[1], vtx_mask=idx)
                else:
                    self.apply_affine(fit_affine(v0[idx], v1[idx], return_rigid=False), gear[1], vtx_mask=idx)
        elif mode == 4:
            self.set_vertices(self.vertices(gear[0]), gear[1])
            self.set_offset(self.offset(gear[0]), gear[1])
        else:
            raise ValueError

    def triangle_area_deform(self, gear=(GEAR_INITIAL, GEAR_MOVING)):   # mesh.py:1979-1986
        def area(v):
            p = v[self.triangles]
            return cross2d(p[:, 1] - p[:, 0], p[:, 2] - p[:, 1])
        return area(self.vertices(gear[-1])) / area(self.vertices(gear[0]))

    def triangle_edge_deform(self, gear=(GEAR_INITIAL, GEAR_MOVING)):   # mesh.py:1966-1976
        v0 = self.vertices(gear[0])
        v1 = self.vertices(gear[-1])
        T = self.triangles
        Tr = np.roll(T, 1, axis=-1)
        d0 = np.sum((v0[T] - v0[Tr]) ** 2, axis=-1)
        d1 = np.sum((v1[T] - v1[Tr]) ** 2, axis=-1)
        return np.exp(np.max(np.abs(0.5 * np.log(d1 / d0)), axis=-1))

    def effective_stiffness_multiplier(self):               # mesh.py:1600-1621, one material without stiffness_func
        return self.stiffness_multiplier * np.ones_like(self.stiffness_multiplier)

    def stiffness_matrix_local_normalized(self, gear=(GEAR_FIXED, GEAR_MOVING), tri_mask=None, max_stiffness_ratio=1000):
        """mesh.py:3086-3129: K = N^T diag(m, m, m/2) N over the masked triangles (shape at gear[0], Poisson ratio 0,
        multipliers clipped at max/max_stiffness_ratio), stress = float32(K (v[gear1] - v[gear0]))."""
        tidx = np.arange(self.triangles.shape[0])
        if tri_mask is not None:
            tidx = tidx[tri_mask]
        if tidx.size == 0:
            return None, None
        v0 = self.vertices(gear[0])
        T = self.triangles[tidx]
        N = eng_shape_matrix(v0[T], T, 2 * self.num_vertices)
        mm = self.effective_stiffness_multiplier()[tidx]
        if max_stiffness_ratio is not None:
            mn = np.max(mm) / max_stiffness_ratio
            if mn == 0:
                mn = 1
            mm = mm.clip(mn, None)
        D = sparse.diags((mm.reshape(-1, 1) * np.array([1, 1, 0.5])).ravel(), dtype=np.float32)
        K = N.T @ D @ N
        dxy = self.vertices(gear[-1]) - v0
        return K, K.dot(dxy.ravel()).astype(np.float32)

    def triangle_areas(self, gear=GEAR_INITIAL):            # mesh.py:1753-1758, common.py:672-676
        p = self.vertices(gear)[self.triangles]
        return cross2d(p[:, 1] - p[:, 0], p[:, 2] - p[:, 1])

    def cart2bary(self, xy, gear, tid):                     # mesh.py:2191-2217 (tid given)
        xy = np.atleast_2d(xy) - self.offset(gear)
        p = self.vertices(gear)[self.triangles[tid]]
        v0 = xy - p[:, 0]; v1 = xy - p[:, 1]; v2 = xy - p[:, 2]
        a0 = cross2d(v1, v2); a1 = cross2d(v2, v0); a2 = cross2d(v0, v1)
        ac = a0 + a1 + a2
        return np.stack((a0 / ac, a1 / ac, a2 / ac), axis=-1)

    def bary2cart(self, tid, B, gear, offsetting=True):     # mesh.py:2220-2228
        v = self.vertices_w_offset(gear) if offsetting else self.vertices(gear)
        return np.sum(v[self.triangles[tid]] * B.reshape(-1, 3, 1), axis=-2)

    def stiffness_matrix(self, gear=(GEAR_FIXED, GEAR_MOVING)):
        v0 = self.vertices(gear[0])
        v1 = self.vertices(gear[-1])
        return mesh_stiffness(v0, None if v1 is v0 else v1, self.triangles,
                              tri_mult=self.stiffness_multiplier, nu=self.nu)


class RefLink:
    """optimizer.py:17-50: matches as (tid, barycentric) pairs on two meshes."""

    def __init__(self, mesh0, mesh1, tid0, tid1, B0, B1, weight=None, strain=0.05):
        self.meshes = [mesh0, mesh1]
        self.tid0 = np.asarray(tid0); self.tid1 = np.asarray(tid1)
        self.B0 = np.asarray(B0, dtype=np.float64); self.B1 = np.asarray(B1, dtype=np.float64)
        a0 = mesh0.triangle_areas(GEAR_INITIAL)[self.tid0]
        a1 = mesh1.triangle_areas(GEAR_INITIAL)[self.tid1]
        self.sample_err = 0.4387 * (np.minimum(a0, a1)) ** 0.5 * strain     # :26-30
        self.weight = ((self.tid0 >= 0) & (self.tid1 >= 0)).astype(np.float32)
        if weight is not None:
            self.weight = self.weight * weight
        self.residue_weight = np.ones_like(self.weight)

    def total_weight(self):                                 # :313-317
        return self.weight * self.residue_weight

    def dxy(self, gears):                                   # :226-255
        m0, m1 = self.meshes
        x0 = m0.bary2cart(self.tid0, self.B0, gears[0], offsetting=False)
        x1 = m1.bary2cart(self.tid1, self.B1, gears[1], offsetting=False)
        return (x1 - x0) + (m1.offset(gears[1]) - m0.offset(gears[0]))

    def residue_weights(self, gears, mode, length):         # :174-205
        d = self.dxy(gears)
        dis = np.sum(d ** 2, axis=-1) ** 0.5
        dis = ((dis ** 2 - self.sample_err ** 2).clip(0, None)) ** 0.5
        if mode == 'huber':
            w = length / np.maximum(dis, length)
        else:
            w = dis <= length
        return np.asarray(w).astype(np.float32)


# ------------------------------------------------------------------ system assembly
def index_offsets(meshes):
    """optimizer.py:960-970: DoF offset per mesh, -1 for locked."""
    off = []
    cur = 0
    for m in meshes:
        if m.locked:
            off.append(-1)
        else:
            off.append(cur)
            cur += 2 * m.num_vertices
    return off, cur


def system_stiffness(meshes, gear=(GEAR_FIXED, GEAR_MOVING)):
    """optimizer.py:802-829: block-diagonal K*soft_factor and stress."""
    Ks, Ss = [], []
    for m in meshes:
        if m.locked:
            continue
        K, s = m.stiffness_matrix(gear)
        Ks.append(K * m.soft_factor)
        Ss.append(s * m.soft_factor)
    return sparse.block_diag(Ks, format='csr'), np.concatenate(Ss, axis=None)


def crosslink_shape_matrix(meshes, links):
    """optimizer.py:873-901 + Link.shape_matrix_contrib (114-131): S float32."""
    offs, ndof = index_offsets(meshes)
    lut = {id(m): o for m, o in zip(meshes, offs)}
    data, idx, ptr = [], [], [np.array([0])]
    cur = 0
    for lk in links:
        m0, m1 = lk.meshes
        Bs, Is = [], []
        if not m0.locked:
            Bs.append(lk.B0)
            Is.append(2 * m0.triangles[lk.tid0] + lut[id(m0)])
        if not m1.locked:
            Bs.append(-lk.B1)
            Is.append(2 * m1.triangles[lk.tid1] + lut[id(m1)])
        if not Bs:
            continue
        B = np.concatenate(Bs, axis=-1)
        I = np.concatenate(Is, axis=-1)
        data.append(B.ravel()); idx.append(I.ravel())
        ptr.append((np.arange(B.shape[0]) + 1) * B.shape[1] + cur)
        cur += B.size
    data = np.concatenate(data); idx = np.concatenate(idx); ptr = np.concatenate(ptr)
    return sparse.csr_matrix((data, idx, ptr), shape=(ptr.size - 1, ndof), dtype=np.float32)


def crosslink_terms(meshes, links, start_gear=GEAR_MOVING, target_gear=GEAR_MOVING):
    """optimizer.py:832-870: C = Cx + Cy (float32), rhs float64."""
    S = crosslink_shape_matrix(meshes, links)
    rows = []
    wts = []
    for lk in links:
        if all(m.locked for m in lk.meshes):
            continue
        gears = [target_gear if m.locked else start_gear for m in lk.meshes]
        rows.append(lk.dxy(gears))
        wts.append(lk.total_weight())
    r = np.concatenate(rows, axis=0).astype(np.float64)
    w = np.concatenate(wts).astype(np.float32)
    D = sparse.diags(w, shape=(w.size, w.size))
    Cx = (S.T @ D @ S).T
    Cy = sparse.csr_matrix((Cx.data, Cx.indices + 1, np.insert(Cx.indptr[:-1], 0, 0)), shape=Cx.shape)
    C = Cx + Cy
    rhs = S.T.dot(w * r[:, 0])
    ry = S.T.dot(w * r[:, 1])
    rhs[1:] = rhs[1:] + ry[0:-1]
    return C, rhs


def relative_lambda_trace(K, C, stiffness_lambda, crosslink_lambda):
    """optimizer.py:1573-1590."""
    if stiffness_lambda < 0 or crosslink_lambda < 0:
        ratio = abs(stiffness_lambda / crosslink_lambda)
        tr = C.trace()
        if tr == 0:
            stiffness_lambda = 0
        else:
            dk = K.diagonal()
            dc = C.diagonal()
            stiffness_lambda = abs(ratio * tr / np.sum(dk[dc != 0]))
        crosslink_lambda = 1.0
    return stiffness_lambda, crosslink_lambda


def linear_system(meshes, links, stiffness_lambda=1.0, crosslink_lambda=-1.0,
                  shape_gear=GEAR_FIXED, start_gear=GEAR_MOVING, target_gear=GEAR_MOVING):
    """optimizer.py:1307-1310,1416-1418: A = ls*K + lc*C, b = lc*rhs - ls*stress."""
    K, stress = system_stiffness(meshes, gear=(shape_gear, start_gear))
    C, rhs = crosslink_terms(meshes, links, start_gear=start_gear, target_gear=target_gear)
    ls, lc = relative_lambda_trace(K, C, stiffness_lambda, crosslink_lambda)
    A = ls * K + lc * C
    b = lc * rhs - ls * stress
    return A, b, (K, stress, C, rhs, ls, lc)


# ------------------------------------------------------------------ solvers
def jacobi_diag(A):
    """optimizer.py:1962-1966: 1/clip(diag(A), min(1, max/1000))."""
    d = A.diagonal()
    if d.max() <= 0:
        return None
    return 1.0 / d.clip(min(1.0, d.max() / 1000), None)


def pcg(A, b, x0=None, rtol=1e-7, maxiter=10000, minv=None):
    """Jacobi-preconditioned CG to ||Ax-b|| <= rtol*||b|| (true residual
    re-evaluated at exit).  This is the algorithm the HIP solver implements;
    it reaches the same fixed point as the reference's restarted MINRES
    (SURVEY.md sec.7 'Hard parts')."""
    A = sparse.csr_matrix(A)
    n = b.size
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    if minv is None:
        minv = jacobi_diag(A)
        if minv is None:
            minv = np.ones(n)
    bn = np.linalg.norm(b)
    if bn == 0 or maxiter == 0:
        return np.zeros(n), 0, 0.0
    r = b - A.dot(x)
    z = minv * r
    p = z.copy()
    rz = r.dot(z)
    it = 0
    while it < maxiter:
        if np.linalg.norm(r) <= rtol * bn:
            break
        Ap = A.dot(p)
        alpha = rz / p.dot(Ap)
        x += alpha * p
        r -= alpha * Ap
        z = minv * r
        rz_new = r.dot(z)
        p = z + (rz_new / rz) * p
        rz = rz_new
        it += 1
    rel = np.linalg.norm(A.dot(x) - b) / bn
    return x, it, rel


def solve_direct(A, b):
    """Ground truth for parity: sparse LU of the symmetrised system."""
    A = sparse.csc_matrix(0.5 * (A + A.T))
    return spla.spsolve(A, b)


def solve_reference_style(A, b, tol=1e-7, atol=None, maxiter=None, eval_step=10, chances=None,
                          early_stop_thresh=None, x0=None):
    """optimizer.py:1945-2080 with solver='minres', M='jacobi',
    tolerated_perturbation=None (the unseeded-random early exit is disabled so
    that the result is deterministic), no edc, check_converge=True.
    Returns (x, number of minres iterations, number of outer restarts)."""
    A = sparse.csr_matrix(0.5 * (A + A.T))
    d = A.diagonal()
    M = sparse.diags(1 / d.clip(min(1.0, d.max() / 1000), None)) if d.max() > 0 else None
    bn = np.linalg.norm(b)
    if maxiter == 0 or bn == 0:
        return np.zeros_like(b), 0, 0
    if atol is not None:
        tol = max(tol, atol / bn)
    atol = tol * bn
    x = np.zeros_like(b) if x0 is None else x0
    tol0 = tol
    total_it = 0
    rounds = 0
    maxiter_t = maxiter

    class _Stop(Exception):
        pass

    while True:
        st = dict(count=0, min_cost=np.inf, sol=None, last_cost=np.inf, last_x=0, exit_count=0, code=0)

        def cb(xk):                                         # optimizer.py:1913-1942
            st['count'] += 1
            c = st['count']
            if (c % eval_step == 0) or (c < min(eval_step, 5)):
                cost = np.linalg.norm(A.dot(xk) - b)
                if cost < st['min_cost']:
                    st['min_cost'] = cost
                    st['sol'] = xk.copy()
                if cost < atol:
                    raise _Stop
                if (chances is not None) and (c >= eval_step):
                    if cost > st['last_cost']:
                        st['exit_count'] += 1
                    elif early_stop_thresh is not None:
                        if np.max(np.abs(xk - st['last_x'])) <= early_stop_thresh:
                            st['exit_count'] += 1
                        else:
                            st['exit_count'] = 0
                    else:
                        st['exit_count'] = 0
                    if st['exit_count'] > chances:
                        st['code'] = 2
                        raise _Stop
                    st['last_x'] = xk.copy()
                    st['last_cost'] = cost
        mi = 100 if maxiter_t is None else min(maxiter_t, 100)
        try:
            x, _ = spla.minres(A, b, x0=x, M=M, maxiter=mi, callback=cb, rtol=tol)
            cost0 = np.linalg.norm(A.dot(x) - b)
            if cost0 > st['min_cost']:
                x = st['sol']
            cost = min(cost0, st['min_cost'])
        except _Stop:
            x = st['sol']
            cost = st['min_cost']
        total_it += st['count']
        rounds += 1
        if cost <= atol:
            break
        if st['code'] != 0:
            break
        if maxiter_t is not None:
            maxiter_t -= st['count']
            if maxiter_t <= 0:
                break
        tol = max(tol0, 0.1 * atol / cost)
    return x, total_it, rounds


def apply_solution(meshes, dd, start_gear=GEAR_MOVING, target_gear=GEAR_MOVING):
    """optimizer.py:1421-1434 + mesh.py:2400-2413."""
    offs, _ = index_offsets(meshes)
    for m, o in zip(meshes, offs):
        if o < 0:
            continue
        d = dd[o:o + 2 * m.num_vertices].reshape(-1, 2)
        m.set_field(d, gear=(start_gear, target_gear))


def optimize_linear_grouped(meshes, links, groupings, stiffness_lambda=1.0, crosslink_lambda=-1.0,
                            shape_gear=GEAR_FIXED, start_gear=GEAR_MOVING, target_gear=GEAR_MOVING, return_system=False,
                            remove_extra_dof=False, dof_selector=None):
    """optimizer.py:1378-1437 with `groupings`: the members of a group share their degrees of freedom
    (T K T^T / mean(count) ...), a group with a locked member is locked; exact solve.  Returns (||b||, ||A d - b||).
    remove_extra_dof / dof_selector: the selector over the degrees of freedom of the free MESHES (optimizer.py:1320-1377) is
    folded into the groups, `edc = (T_m @ edc) > 0` (optimizer.py:1412-1413): a group's degree of freedom is solved when any
    member has it solved; held ones leave the system (optimizer.py:1976-1991) and stay zero -- the cost is that of the FULL system."""
    K, stress = system_stiffness(meshes, gear=(shape_gear, start_gear))
    C, rhs = crosslink_terms(meshes, links, start_gear=start_gear, target_gear=target_gear)
    # the lambdas come from the cached MESH-level terms (relative_lambda_trace reads self.stiffness_matrix() /
    # self.crosslink_terms(), optimizer.py:1573-1590), not from the grouped matrices
    ls, lc = relative_lambda_trace(K, C, stiffness_lambda, crosslink_lambda)
    lock = np.array([m.locked for m in meshes])
    groupings = np.asarray(groupings)
    group_u, indx, group_nm, g_cnt = np.unique(groupings, return_index=True, return_inverse=True, return_counts=True)
    glock = np.zeros(group_u.size, dtype=bool)
    np.logical_or.at(glock, group_nm, lock)
    vnum = np.array([meshes[k].num_vertices * 2 for k in indx]) * (~glock)
    acc = np.cumsum(vnum)
    gdof = int(acc[-1])
    goff = np.concatenate(([0], acc[:-1]))
    goff[glock] = -1
    expanded = goff[group_nm]
    cur = 0
    i0, i1 = [], []
    for m, gio in zip(meshes, expanded):
        if m.locked:
            continue
        sz = 2 * m.num_vertices
        if gio >= 0:
            i0.append(np.arange(cur, cur + sz)); i1.append(np.arange(gio, gio + sz))
        cur += sz
    i0 = np.concatenate(i0); i1 = np.concatenate(i1)
    T = sparse.csr_matrix((np.ones(i0.size, dtype=np.float32), (i1, i0)), shape=(gdof, K.shape[0]))
    mc = np.mean(g_cnt)
    K = T @ K @ T.transpose() / mc
    C = T @ C @ T.transpose() / mc
    stress = T @ stress / mc
    rhs = T @ rhs / mc
    A = ls * K + lc * C
    b = lc * rhs - ls * stress
    A = 0.5 * (A + A.T)
    edc = dof_selector if dof_selector is not None else (extra_dof_selector(meshes, links) if remove_extra_dof else None)
    if edc is not None:
        edc = np.asarray(T @ np.asarray(edc, dtype=np.float32)).ravel() > 0
    if edc is not None and not edc.all():
        dd = np.zeros_like(b)
        dd[edc] = solve_direct(sparse.csr_matrix(A)[edc][:, edc], b[edc])
    else:
        dd = solve_direct(A, b)
    cost = (float(np.linalg.norm(b)), float(np.linalg.norm(A.dot(dd) - b)))
    if cost[1] < cost[0]:
        for m, gio in zip(meshes, expanded):
            if m.locked or gio < 0:
                continue
            m.set_field(dd[gio:gio + 2 * m.num_vertices].reshape(-1, 2), gear=(start_gear, target_gear))
    if return_system:
        return cost, A, b, expanded
    return cost


def extra_dof_selector(meshes, links):
    """optimizer.py:1360-1377 (remove_extra_dof): in every link-connected subsystem without a locked mesh the first three
    degrees of freedom of its first mesh are held.  Returns the boolean selector over the degrees of freedom of the free
    meshes (True = solved) or None."""
    from scipy.sparse import csgraph
    idx = {id(m): k for k, m in enumerate(meshes)}
    n = len(meshes)
    adj = sparse.lil_matrix((n, n))
    for lk in links:
        a, b = idx[id(lk.meshes[0])], idx[id(lk.meshes[1])]
        adj[a, b] = 1; adj[b, a] = 1
    _, lbl = csgraph.connected_components(adj.tocsr(), directed=False, return_labels=True)
    lock = np.array([m.locked for m in meshes])
    rm = np.zeros(n, dtype=bool)
    for v in np.unique(lbl):
        k = np.nonzero(lbl == v)[0]
        if not np.any(lock[k]):
            rm[k[0]] = True
    if not np.any(rm):
        return None
    edc = []
    for flg, m in zip(rm, meshes):
        if m.locked:
            continue
        sel = np.ones(m.num_vertices * 2, dtype=bool)
        if flg:
            sel[:3] = False
        edc.append(sel)
    return np.concatenate(edc, axis=None)


def material_dof_selector(meshes, names, material_ids, named_uids):
    """optimizer.py:1320-1359 (remove_material_dof): the vertices of the triangles of the named materials are held; a name that
    carries '_freeborder' frees again every vertex that a triangle of another material uses (those are treated first).
    material_ids: per mesh the material id of every triangle (or None), named_uids: per mesh {name: uid}.  Returns the boolean
    selector over the degrees of freedom of the free meshes (True = solved)."""
    marker = '_freeborder'
    names = [names] if isinstance(names, str) else list(names)
    free_border = [s.replace(marker, '') for s in names if marker in s]
    fixed_border = [s for s in names if marker not in s]
    edc = []
    for m, mids, table in zip(meshes, material_ids, named_uids):
        if m.locked:
            continue
        T = np.asarray(m.triangles)
        sel = np.ones(m.num_vertices * 2, dtype=bool)
        for grp, border in ((free_border, True), (fixed_border, False)):
            for name in grp:
                tid = np.zeros(T.shape[0], dtype=bool)
                if mids is not None and name in table:
                    tid = tid | (np.asarray(mids) == table[name])
                vn = np.unique(T[tid])
                sel[2 * vn] = False; sel[2 * vn + 1] = False
                if border:
                    vp = np.unique(T[~tid])
                    sel[2 * vp] = True; sel[2 * vp + 1] = True
        edc.append(sel)
    return np.concatenate(edc, axis=None)


def optimize_linear(meshes, links, tol=1e-7, stiffness_lambda=1.0, crosslink_lambda=-1.0,
                    shape_gear=GEAR_FIXED, start_gear=GEAR_MOVING, target_gear=GEAR_MOVING, exact=True, remove_extra_dof=False,
                    dof_selector=None):
    """optimizer.py:1257-1437 (no groupings).  exact=True solves the system to its fixed point (the parity anchor);
    otherwise PCG to tol.  remove_extra_dof: the held degrees of freedom leave the system (optimizer.py:1976-1991: rows and
    columns are cut out, the solution is zero there)."""
    A, b, _ = linear_system(meshes, links, stiffness_lambda, crosslink_lambda,
                            shape_gear, start_gear, target_gear)
    A = 0.5 * (A + A.T)
    edc = dof_selector if dof_selector is not None else (extra_dof_selector(meshes, links) if remove_extra_dof else None)
    if edc is not None and not edc.all():
        Ar = sparse.csr_matrix(A)[edc][:, edc]
        dr = solve_direct(Ar, b[edc]) if exact else pcg(Ar, b[edc], rtol=tol)[0]
        dd = np.zeros_like(b)
        dd[edc] = dr
    elif exact:
        dd = solve_direct(A, b)
    else:
        dd, _, _ = pcg(A, b, rtol=tol)
    cost = (float(np.linalg.norm(b)), float(np.linalg.norm(A.dot(dd) - b)))
    if cost[1] < cost[0]:
        apply_solution(meshes, dd, start_gear, target_gear)
    return cost


def fit_affine(pts0, pts1, return_rigid=False, weight=None, svd_clip=(1, 1), avoid_flip=True):
    """spatial.py:21-73: pts0 ~ pts1 @ A (3x3, row-vector convention); with return_rigid also the transform whose
    2x2 part has its singular values clipped to svd_clip.  Restated statement by statement (including the scaled
    translation row of the least-squares solution that the reference carries into the result)."""
    pts0 = np.asarray(pts0, dtype=np.float64).reshape(-1, 2)
    pts1 = np.asarray(pts1, dtype=np.float64).reshape(-1, 2)
    mm0 = pts0.mean(axis=0)
    mm1 = pts1.mean(axis=0)
    pts0 = pts0 - mm0
    pts1 = pts1 - mm1
    std0 = np.sum(np.std(pts0, axis=0) ** 2) ** 0.5
    std1 = np.sum(np.std(pts1, axis=0) ** 2) ** 0.5
    std_scl = max(std0, std1)
    if std_scl < 1e-6:
        std_scl = 1
    p0 = np.insert(pts0 / std_scl, 2, 1, axis=-1)
    p1 = np.insert(pts1 / std_scl, 2, 1, axis=-1)
    if weight is not None:
        w = np.asarray(weight) ** 0.5
        p0 = p0 * w.reshape(-1, 1)
        p1 = p1 * w.reshape(-1, 1)
    res = np.linalg.lstsq(p1, p0, rcond=None)
    r1 = np.linalg.matrix_rank(p0)
    A = res[0]
    r = min(res[2], r1)
    if avoid_flip and np.linalg.det(A) < 0:
        r = 2
    if r == 1:
        A = np.eye(3)
    elif r == 2:
        q0 = np.concatenate((pts0, pts0[:, ::-1] * np.array([1, -1])), axis=0)
        q1 = np.concatenate((pts1, pts1[:, ::-1] * np.array([1, -1])), axis=0)
        A = np.linalg.lstsq(np.insert(q1 / std_scl, 2, 1, axis=-1), np.insert(q0 / std_scl, 2, 1, axis=-1), rcond=None)[0]
    R = A
    if return_rigid and svd_clip is not None:
        u, sv, vh = np.linalg.svd(A[:2, :2], compute_uv=True)
        sv = sv.clip(svd_clip[0], svd_clip[-1])
        R = A.copy()
        R[:2, :2] = u @ np.diag(sv) @ vh
        R[-1, :2] = R[-1, :2] + mm0 - mm1 @ R[:2, :2]
        R[:, -1] = np.array([0, 0, 1])
    A[-1, :2] = A[-1, :2] + mm0 - mm1 @ A[:2, :2]
    A[:, -1] = np.array([0, 0, 1])
    return (A, R) if return_rigid else A


def svds_to_deform(s):
    """mesh.py:3358-3365: (N, k) singular values -> deformation in [0, 1) (not flipped) or >= 1 (flipped)."""
    s = np.asarray(s, dtype=np.float64)
    d = np.where(s < 1, 1 - s, 1 - 1 / np.where(s == 0, 1, s))
    return np.max(d, axis=-1)


def relax_mesh(M, free_vertices=None, free_triangles=None, gear=(GEAR_FIXED, GEAR_MOVING)):
    """optimizer.py:2110-2154 with the inner ``solve`` replaced by the exact solution of the same system: free the
    given vertices (or those belonging only to free triangles), re-rest the mesh on its INITIAL shape rigidly aligned
    to the current one, relax the free vertices with the rest held."""
    locked = M.locked
    M.locked = False
    nv = M.num_vertices
    if free_vertices is not None:
        vindx = free_vertices
    elif free_triangles is not None:
        T = M.triangles[~free_triangles]
        vindx = ~np.isin(np.arange(nv), np.unique(T))
    else:
        return False
    vmask = np.zeros(nv, dtype=bool)
    vmask[vindx] = True
    if not np.any(vmask):
        return False
    tmask = np.any(vmask[M.triangles], axis=-1)
    vmask_pad = np.repeat(vmask, 2)
    fixed_vertices = M.vertices(gear[0])
    fixed_offset = M.offset(gear[0])
    M.anneal(gear=(GEAR_INITIAL, gear[0]), mode=4)
    M.anneal(gear=gear[::-1], mode=2)
    K, stress = M.stiffness_matrix_local_normalized(gear=gear, tri_mask=tmask)
    if K is None:
        return False
    A = sparse.csr_matrix(K)[vmask_pad][:, vmask_pad]
    b = -stress[vmask_pad].astype(np.float64)
    dd = solve_direct(A, b) if np.any(b) else np.zeros_like(b)
    modified = False
    if np.linalg.norm(A.dot(dd) - b) < np.linalg.norm(b) and np.any(dd != 0):
        modified = True
        M.apply_field(dd.reshape(-1, 2), gear[-1], vtx_mask=vmask)
    if gear[0] != gear[1]:
        M.set_vertices(fixed_vertices, gear[0])
        M.set_offset(fixed_offset, gear[0])
    M.locked = locked
    return modified


def most_deformed_region(M, gear=(GEAR_FIXED, GEAR_MOVING), deform_cutoff=0.35, iqr=0):
    """the selection of optimizer.py:2157-2188: (free_vertices, None) for the flip-only mode (deform_cutoff < 0),
    (None, free_triangles) otherwise, (None, None) if nothing exceeds the threshold."""
    sa = M.triangle_area_deform(gear=gear).reshape(-1, 1)
    if deform_cutoff < 0:
        tmask = svds_to_deform(sa) >= 1
        if not np.any(tmask):
            return None, None
        return np.unique(M.triangles[tmask]), None
    deform_thresh = 1 - 1 / (abs(deform_cutoff) + 1)
    sd = M.triangle_edge_deform(gear=gear).reshape(-1, 1)
    defm = np.maximum(svds_to_deform(sa), svds_to_deform(sd))
    m0 = M.effective_stiffness_multiplier()
    idx_m = m0 >= 0.5 * np.median(m0)
    defm = defm - np.median(defm[idx_m])
    thresh_t = max(deform_thresh, 0)
    if iqr > 0:
        qq = np.quantile(defm[idx_m], (0.25, 0.75))
        thresh_t = min(thresh_t, np.max(qq) + iqr * np.ptp(qq))
    tmask = defm > max(thresh_t, 1.0e-3)
    if not np.any(tmask):
        return None, None
    vid = np.unique(M.triangles[tmask])
    return None, np.all(np.isin(M.triangles, vid), axis=-1)


def relax_mesh_most_deformed(M, gear=(GEAR_FIXED, GEAR_MOVING), deform_cutoff=0.35, iqr=0):
    fv, ft = most_deformed_region(M, gear=gear, deform_cutoff=deform_cutoff, iqr=iqr)
    if fv is None and ft is None:
        return False
    return relax_mesh(M, free_vertices=fv, free_triangles=ft, gear=gear)


# ------------------------------------------------------------------ synthetic meshes
def grid_mesh(nx, ny, h=10.0, origin=(0.0, 0.0), diag='alt'):
    """Structured triangulation of an nx x ny node grid (not a reference
    function: `triangle` is absent; meshes are inputs)."""
    xs = origin[0] + h * np.arange(nx)
    ys = origin[1] + h * np.arange(ny)
    vx, vy = np.meshgrid(xs, ys)
    v = np.stack((vx.ravel(), vy.ravel()), axis=-1)
    idx = np.arange(nx * ny).reshape(ny, nx)
    a = idx[:-1, :-1].ravel(); b = idx[:-1, 1:].ravel()
    c = idx[1:, :-1].ravel(); d = idx[1:, 1:].ravel()
    if diag == 'alt':
        par = ((np.arange(nx - 1)[None, :] + np.arange(ny - 1)[:, None]) % 2).ravel().astype(bool)
    else:
        par = np.zeros(a.size, dtype=bool)
    t0 = np.where(par[:, None], np.stack((a, b, c), -1), np.stack((a, b, d), -1))
    t1 = np.where(par[:, None], np.stack((b, d, c), -1), np.stack((a, d, c), -1))
    tri = np.concatenate((t0, t1), axis=0)
    return v, tri


def stiffness_func_table(x, strain, stiffness):
    """material.asymmetrical_elasticity (material.py:546-551): scipy.interpolate.interp1d(strain, stiffness, kind='linear',
    bounds_error=False, fill_value=(stiffness[0], stiffness[-1])) restated: the segment is found with searchsorted (left) clipped
    to [1, n - 1], y = slope (x - x_lo) + y_lo; below / above the table the first / last stiffness."""
    xs = np.asarray(strain, dtype=np.float64); ys = np.asarray(stiffness, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    hi = np.clip(np.searchsorted(xs, x), 1, xs.size - 1)
    lo = hi - 1
    slope = (ys[hi] - ys[lo]) / (xs[hi] - xs[lo])
    y = slope * (x - xs[lo]) + ys[lo]
    y = np.where(x < xs[0], ys[0], y)
    y = np.where(x > xs[-1], ys[-1], y)
    return y


def area_stretch(v_init, v_cur, triangles, linear_mask):
    """mesh.py:2952-2963 (= 3030-3040): signed area of every triangle at the current gear over its INITIAL one, over the same
    ratio of the summed absolute areas of the linear triangles (of all triangles when there is none)."""
    T = np.asarray(triangles)
    def signed_area(v):                                                     # common.py:672-676
        p = np.asarray(v, dtype=np.float64)[T]
        return 0.5 * cross2d(p[:, 1, :] - p[:, 0, :], p[:, 2, :] - p[:, 0, :])
    a0, a1 = signed_area(v_init), signed_area(v_cur)
    if np.any(linear_mask):
        base = np.sum(np.abs(a1[linear_mask])) / np.sum(np.abs(a0[linear_mask]))
    else:
        base = np.sum(np.abs(a1)) / np.sum(np.abs(a0))
    return (a1 / a0) / base


def mesh_stiffness_mixed(v_shape, v_cur, triangles, tri_mult, model, nu, matmult, func=None, tabs=None, v_init=None):
    """mesh.py:3058-3083 for a mesh that mixes linear engineering triangles (model 0, N^T D N path,
    mesh.py:2914-2933) with St-Venant-Kirchhoff / Neo-Hookean triangles (models 1 / 2, element loop of
    mesh.py:2992-3054).  Per-triangle arrays: tri_mult (mesh multiplier, float32), model, nu, matmult
    (material multiplier).  func (per triangle, -1 = none) indexes tabs = [(strain, stiffness), ...]: materials whose
    stiffness is multiplied by a function of the triangle's area stretch INITIAL -> current (v_init; engineering triangles
    through nonlinear_engineering_stiffness_matrix, mesh.py:2937-2971, the others through the f(J) modifier of
    material.py:307-308).  Returns (K float64 CSR, stress float32)."""
    v_shape = np.asarray(v_shape, dtype=np.float64)
    v_cur = np.asarray(v_cur, dtype=np.float64)
    T = np.asarray(triangles)
    ndof = 2 * v_shape.shape[0]
    dxy = v_cur - v_shape
    K = sparse.csr_matrix((ndof, ndof), dtype=np.float64)
    if func is None:
        func = np.full(T.shape[0], -1, dtype=np.int32)
    func = np.asarray(func)
    fac = np.ones(T.shape[0], dtype=np.float64)
    if np.any(func >= 0):
        st = area_stretch(v_shape if v_init is None else v_init, v_cur, T, (model == MODEL_ENG) & (func < 0))
        for k, (xs, ys) in enumerate(tabs):
            fac[func == k] = stiffness_func_table(st[func == k], xs, ys)
    lin = model == MODEL_ENG
    for nu_v in np.unique(nu[lin]):
        for mm_v in np.unique(matmult[lin & (nu == nu_v)]):
            for has_f in (False, True):                                    # linear materials first (mesh.py:3060-3067)
                sel = np.flatnonzero(lin & (nu == nu_v) & (matmult == mm_v) & ((func >= 0) == has_f))
                if sel.size == 0:
                    continue
                N = eng_shape_matrix(v_shape[T[sel]], T[sel], ndof)
                K = K + eng_stiffness_from_shape(N, multiplier=tri_mult[sel], nu=float(nu_v), mat_multiplier=float(mm_v),
                                                 stretch_factor=fac[sel] if has_f else None)
    stress = K.dot(dxy.ravel()).astype(np.float32)                        # mesh.py:3068-3072
    Td = np.repeat(T * 2, 2, axis=-1)
    Td[:, 1::2] += 1
    V = np.zeros(ndof, dtype=np.float32)
    for md in (MODEL_SVK, MODEL_NHK):
        for nu_v in np.unique(nu[model == md]):
            for mm_v in np.unique(matmult[(model == md) & (nu == nu_v)]):
                sel = np.flatnonzero((model == md) & (nu == nu_v) & (matmult == mm_v))
                B, areas = element_shape_B(v_shape[T[sel]])
                uv = dxy[T[sel]].reshape(-1, 6)
                Ke, Pe = element_stiffness(B, areas, uv, md, nu=float(nu_v))
                mm = tri_mult[sel].reshape(-1, 1, 1) * (float(mm_v) * fac[sel].reshape(-1, 1, 1))       # mesh.py:3041, material.py:307-308
                i1 = np.tile(Td[sel].reshape(-1, 1, 6), (1, 6, 1))
                i2 = np.swapaxes(i1, 1, 2)
                K = K + sparse.csr_matrix(((Ke * mm).ravel(), (i1.ravel(), i2.ravel())), shape=(ndof, ndof))
                np.add.at(V, Td[sel].ravel(), (Pe * mm).ravel())
    return sparse.csr_matrix(K), stress + V


def newton_fixed_point(m_locked, m_free, links, tri_mult, model, nu, matmult, stiffness_lambda=1.0, crosslink_lambda=1.0,
                       max_steps=30, tol=1e-6, func=None, tabs=None):
    """Fixed point of SLM.optimize_Newton_Raphson (optimizer.py:1440-1544) for one free mesh with non-linear elements
    linked to a locked one, with fixed (positive) lambdas: every step re-assembles the tangent stiffness and the internal
    force at the current MOVING gear (mesh.py:2937-3083), solves the tangent system exactly (sparse LU instead of the
    reference's Krylov legs, whose exits depend on wall-clock settings) and applies the field like optimize_linear
    (optimizer.py:1433-1434).  The internal force is float32 like the reference's stress (mesh.py:3068-3072), which floors
    ||b|| at a few 1e-7 of its first value: `tol` sits above that floor.  The fixed point does not depend on the iteration path.  Returns the list of ||b|| per step."""
    from scipy.sparse.linalg import spsolve
    t = m_free.triangles
    costs = []
    for _ in range(max_steps):
        K, stress = mesh_stiffness_mixed(m_free.vertices(GEAR_FIXED), m_free.vertices(GEAR_MOVING), t, tri_mult, model, nu, matmult,
                                         func=func, tabs=tabs, v_init=m_free.vertices(GEAR_INITIAL))
        C, rhs = crosslink_terms([m_locked, m_free], links, start_gear=GEAR_MOVING, target_gear=GEAR_MOVING)
        A = (stiffness_lambda * m_free.soft_factor) * K + crosslink_lambda * C.astype(np.float64)
        b = crosslink_lambda * rhs - (stiffness_lambda * m_free.soft_factor) * stress.astype(np.float64)
        costs.append(float(np.linalg.norm(b)))
        if costs[-1] <= tol * costs[0]:
            break
        d = spsolve(sparse.csc_matrix(0.5 * (A + A.T)), b)
        m_free.set_field(d.reshape(-1, 2), gear=(GEAR_MOVING, GEAR_MOVING))
    return costs
