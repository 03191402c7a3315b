"""Host geometry for a tile pair whose mesh1 has been relaxed into a NON-rigid field between two spacings of
``matcher.iterative_xcorr_matcher_w_mesh`` (feabas/matcher.py:717-742).  The reference keeps all of this on the host
as well (shapely / matplotlib / numpy inside MeshRenderer and Mesh); here it is numpy over the blocks of one pair:

* ``block_affines``  -- the tier decision of ``MeshRenderer.crop_field`` (renderer.py:453-563) with the affine
  approximator of ``MeshRenderer.from_mesh`` (renderer.py:90-109): global affine, per-block affine
  (``bbox_affine_tform``, renderer.py:397-416), or the exact field;
* ``exact_field``    -- the piecewise-linear inverse map of ``field_w_weight`` (renderer.py:259-300) for one block;
* ``locate``         -- ``Mesh.tri_finder`` + ``cart2bary`` (mesh.py:2080-2217) on the deformed cartesian mesh.

The sampling itself (cv2.remap's bilinear rule) runs on the device: inside the NCC loaders for the affine tiers
(fb_ncc_blocks_affine_dev), through fb_remap_dev for the exact tier.
"""
import numpy as np

from .common import fit_affine


def affine_residue(v1, v0, A):
    """largest distance between v1 and v0 mapped through A (renderer.py:99-100, 414)"""
    return float(np.max(np.sum((v1 - v0 @ A[:2, :2] - A[-1, :2]) ** 2, axis=-1)) ** 0.5)


def fit_affine_rows(v1, v0, member):
    """``spatial.fit_affine(v1[idx], v0[idx])`` (spatial.py:21-73, unweighted) for B vertex subsets at once.
    member [B, V] bool.  The least squares of spatial.py:44 on centred points is block diagonal: the 2x2 part solves
    the 2x2 normal equations (the common scale cancels), the translation row is mm0 - mm1 @ A.  Subsets that are rank
    deficient or flipped (spatial.py:45-60) go through the statement-by-statement function.
    Returns A [B, 3, 3] with v1 ~ v0 @ A[:2, :2] + A[2, :2] and the largest residue per subset."""
    member = np.asarray(member, dtype=bool)
    B = member.shape[0]
    w = member.astype(np.float64)
    n = np.maximum(w.sum(axis=1), 1.0)
    m0 = (w @ v1) / n[:, None]                                  # mm0: mean of pts0 = v1
    m1 = (w @ v0) / n[:, None]                                  # mm1: mean of pts1 = v0
    c0 = (v1[None, :, :] - m0[:, None, :]) * w[:, :, None]
    c1 = (v0[None, :, :] - m1[:, None, :]) * w[:, :, None]
    G = np.einsum('bvi,bvj->bij', c1, c1)
    Hm = np.einsum('bvi,bvj->bij', c1, c0)
    detG = G[:, 0, 0] * G[:, 1, 1] - G[:, 0, 1] * G[:, 1, 0]
    scale = np.maximum(G[:, 0, 0], G[:, 1, 1])
    ok = (w.sum(axis=1) >= 3) & (detG > 1e-9 * scale * scale)
    A2 = np.tile(np.eye(2), (B, 1, 1))
    A2[ok] = np.linalg.solve(G[ok], Hm[ok])
    ok &= np.linalg.det(A2) > 0
    A = np.tile(np.eye(3), (B, 1, 1))
    A[:, :2, :2] = A2
    A[:, 2, :2] = m0 - np.einsum('bi,bij->bj', m1, A2)
    for b in np.flatnonzero(~ok):
        idx = np.flatnonzero(member[b])
        if idx.size:
            A[b] = fit_affine(v1[idx], v0[idx], return_rigid=True, svd_clip=None)[1]
    d = v1[None, :, :] - (np.einsum('vi,bij->bvj', v0, A[:, :2, :2]) + A[:, None, 2, :2])
    res = np.sqrt(np.max(np.where(member, np.sum(d * d, axis=-1), 0.0), axis=1))
    res[~member.any(axis=1)] = np.inf
    return A, res


def tri_box_hits(tp, boxes):
    """closed triangles tp [T, 3, 2] against closed boxes [B, 4] = (xmin, ymin, xmax, ymax): what
    ``STRtree.query(box, predicate='intersects')`` answers (renderer.py:405).  Separating-axis test on the two box
    axes and the three edge normals; touching counts.  Returns [B, T] bool."""
    tp = np.asarray(tp, dtype=np.float64)
    bx = np.asarray(boxes, dtype=np.float64)
    tx0, tx1 = tp[:, :, 0].min(axis=1), tp[:, :, 0].max(axis=1)
    ty0, ty1 = tp[:, :, 1].min(axis=1), tp[:, :, 1].max(axis=1)
    sep = (tx1[None, :] < bx[:, None, 0]) | (tx0[None, :] > bx[:, None, 2]) | (ty1[None, :] < bx[:, None, 1]) | (ty0[None, :] > bx[:, None, 3])
    cx = np.stack((bx[:, 0], bx[:, 2], bx[:, 2], bx[:, 0]), axis=-1)          # [B, 4] corner x
    cy = np.stack((bx[:, 1], bx[:, 1], bx[:, 3], bx[:, 3]), axis=-1)
    for k in range(3):
        e = tp[:, (k + 1) % 3] - tp[:, k]
        nx_, ny_ = -e[:, 1], e[:, 0]                                           # [T]
        pt = tp[:, :, 0] * nx_[:, None] + tp[:, :, 1] * ny_[:, None]           # [T, 3]
        pb = cx[:, None, :] * nx_[None, :, None] + cy[:, None, :] * ny_[None, :, None]    # [B, T, 4]
        sep |= (pt.max(axis=1)[None, :] < pb.min(axis=2)) | (pt.min(axis=1)[None, :] > pb.max(axis=2))
    return ~sep


def block_affines(vm, v_init, tris, bboxes, tol):
    """tier of every block of one pair and its affine map.
    vm: MOVING vertices of mesh1 (with offset), v_init: its INITIAL vertices, bboxes [B, 4] in the MOVING frame.
    Returns tier [B] (1 global affine, 2 block affine, 3 exact field), A [B, 3, 3] (image = moving @ A[:2, :2] + A[2, :2];
    undefined for tier 3) and hits [B, T] (None for tier 1): the triangles that touch each block."""
    bboxes = np.asarray(bboxes, dtype=np.float64)
    nb = bboxes.shape[0]
    tier = np.full(nb, 3, dtype=np.int32)
    A = np.tile(np.eye(3), (nb, 1, 1))
    if not tol > 0:
        return tier, A, tri_box_hits(vm[tris], bboxes - 0.5)
    A_g = fit_affine(v_init, vm)                                 # renderer.py:98: fit_affine(v1_a, v0_a)
    if affine_residue(v_init, vm, A_g) < tol:
        tier[:] = 1
        A[:] = A_g
        return tier, A, None
    hits = tri_box_hits(vm[tris], bboxes - 0.5)                  # renderer.py:405: box(*(bbox0 - 0.5))
    member = np.zeros((nb, vm.shape[0]), dtype=bool)
    bi, ti = np.nonzero(hits)
    for k in range(3):
        member[bi, tris[ti, k]] = True
    A_b, res = fit_affine_rows(v_init, vm, member)
    good = res < tol
    tier[good] = 2
    A[good] = A_b[good]
    return tier, A, hits


def exact_field(vm, v_init, tris, hit_tris, x0, y0, h, w):
    """field_w_weight (renderer.py:259-300) for one block: output pixel (x0 + i, y0 + j) is located in the MOVING
    triangles `hit_tris` and mapped to the image by linear interpolation of the INITIAL vertices
    (matplotlib.tri.LinearTriInterpolator in the reference).  Pixels outside every triangle are masked."""
    xs = np.linspace(x0, x0 + w, num=w, endpoint=False, dtype=float)
    ys = np.linspace(y0, y0 + h, num=h, endpoint=False, dtype=float)
    xx, yy = np.meshgrid(xs, ys)
    map_x = np.zeros((h, w)); map_y = np.zeros((h, w)); mask = np.zeros((h, w), dtype=bool)
    for t in hit_tris:
        p = vm[tris[t]]
        d0x, d0y = xx - p[0, 0], yy - p[0, 1]
        d1x, d1y = xx - p[1, 0], yy - p[1, 1]
        d2x, d2y = xx - p[2, 0], yy - p[2, 1]
        a0 = d1x * d2y - d1y * d2x; a1 = d2x * d0y - d2y * d0x; a2 = d0x * d1y - d0y * d1x
        tot = a0 + a1 + a2
        with np.errstate(divide='ignore', invalid='ignore'):
            b0, b1, b2 = a0 / tot, a1 / tot, a2 / tot
        inside = (b0 >= 0) & (b1 >= 0) & (b2 >= 0) & ~mask
        if not inside.any():
            continue
        q = v_init[tris[t]]
        map_x[inside] = (b0 * q[0, 0] + b1 * q[1, 0] + b2 * q[2, 0])[inside]
        map_y[inside] = (b0 * q[0, 1] + b1 * q[1, 1] + b2 * q[2, 1])[inside]
        mask |= inside
    return map_x, map_y, mask


def locate(vm, tris, xs, ys, pts, eps=1e-9):
    """triangle and barycentric coordinates of points in the DEFORMED cartesian mesh (vm = MOVING vertices with offset,
    cells of the nx x ny grid split as Mesh.from_bbox does): Mesh.tri_finder + cart2bary (mesh.py:2080-2217).
    The cell is guessed by pulling the point back with the displacement of its nearest grid node, then the triangles of
    the 3 x 3 cells around it are tested.  Returns tid [K] (-1 outside the mesh) and B [K, 3] (nan outside)."""
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    K = pts.shape[0]
    nx, ny = xs.size, ys.size
    gx, gy = np.meshgrid(xs, ys)
    U = vm.reshape(ny, nx, 2) - np.stack((gx, gy), axis=-1)
    q = pts - U.reshape(-1, 2).mean(axis=0)
    for _ in range(2):
        i = np.clip(np.rint((q[:, 0] - xs[0]) / (xs[-1] - xs[0]) * (nx - 1)).astype(np.int64), 0, nx - 1)
        j = np.clip(np.rint((q[:, 1] - ys[0]) / (ys[-1] - ys[0]) * (ny - 1)).astype(np.int64), 0, ny - 1)
        q = pts - U[j, i]
    ci = np.clip(np.searchsorted(xs, q[:, 0], side='right') - 1, 0, nx - 2)
    cj = np.clip(np.searchsorted(ys, q[:, 1], side='right') - 1, 0, ny - 2)
    tid = np.full(K, -1, dtype=np.int32)
    bary = np.full((K, 3), np.nan)
    best = np.full(K, -np.inf)
    for dj in (0, -1, 1):
        for di in (0, -1, 1):
            ii, jj = ci + di, cj + dj
            valid = (ii >= 0) & (ii < nx - 1) & (jj >= 0) & (jj < ny - 1)
            cell = np.where(valid, jj * (nx - 1) + ii, 0)
            for half in (0, 1):
                t = 2 * cell + half
                p = vm[tris[t]]                                            # [K, 3, 2]
                d0, d1, d2 = pts - p[:, 0], pts - p[:, 1], pts - p[:, 2]
                a0 = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]
                a1 = d2[:, 0] * d0[:, 1] - d2[:, 1] * d0[:, 0]
                a2 = d0[:, 0] * d1[:, 1] - d0[:, 1] * d1[:, 0]
                tot = a0 + a1 + a2
                with np.errstate(divide='ignore', invalid='ignore'):
                    b = np.stack((a0 / tot, a1 / tot, a2 / tot), axis=-1)
                score = np.where(valid, np.nan_to_num(b.min(axis=1), nan=-np.inf), -np.inf)
                take = (score > best) & (best < 0)                          # keep the first triangle that contains the point
                tid[take] = t[take]
                bary[take] = b[take]
                best[take] = score[take]
    out = best < -eps
    tid[out] = -1
    bary[out] = np.nan
    return tid, bary
