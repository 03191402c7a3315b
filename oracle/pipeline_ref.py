"""Oracle of the per-tile-pair matcher (NCC side of matcher.stitching_matcher,
feabas/matcher.py:224-367 + 430-751).  Two branches:
 * rigid: every mesh relaxation between rounds (matcher.py:725) is a rigid integer translation and every crop of
   MeshRenderer.crop_multiple is an integer translation of the DoG'd strip;
 * deformed: any other relaxation.  mesh1 keeps its relaxed MOVING gear, the next round's blocks come from the deformed
   bounding box, image-1 patches are rendered through the mesh with the renderer's three tiers (global affine,
   per-block affine, exact piecewise-linear field; renderer.py:47-166, 397-563, 601-648) and cv2.remap's bilinear rule
   (UNPINNED restatement, ncc_ref.remap_bilinear_cv), links are located in the deformed mesh (mesh.py:2080-2217).
Every relaxation is solved exactly on the cartesian mesh pair (the reference iterates to tol 0.01 / max(1, max_dis)).
TEST INFRASTRUCTURE (see oracle/__init__.py).  cv2 / triangle / shapely are absent: the x0.5 INTER_AREA downsample is
the unpinned restatement ncc_ref.area_downsample2, shapely's triangle / box `intersects` is a separating-axis test,
matplotlib.tri (present) provides the trifinder and the linear interpolator like the reference.
"""
import numpy as np

from . import ncc_ref, fem_ref

MAXIMUM_DEFORM_ALLOWED = 0.35     # feabas/config.py:33


def cartesian_mesh(W, H, mesh_size, min_num_blocks=2, max_aspect_ratio=2):
    """Mesh.from_bbox((0,0,W,H), cartesian=True) (mesh.py:403-435): node grid at pixel centres - 0.5.
    The reference lets `triangle` pick one diagonal per rectangle (implementation defined);
    here every cell (a b / c d) is split along a-d into (a,b,d), (a,d,c)."""
    nx = max(np.round(W / mesh_size), min_num_blocks)
    ny = max(np.round(H / mesh_size), min_num_blocks)
    dx, dy = W / nx, H / ny
    if dx > max_aspect_ratio * dy:
        dx = max_aspect_ratio * dy
    elif dy > max_aspect_ratio * dx:
        dy = max_aspect_ratio * dx
    nx = int(np.ceil(W / dx)) + 1
    ny = int(np.ceil(H / dy)) + 1
    xs = np.linspace(0, W, num=nx, endpoint=True) - 0.5
    ys = np.linspace(0, H, num=ny, endpoint=True) - 0.5
    vx, vy = np.meshgrid(xs, ys)
    v = np.stack((vx.ravel(), vy.ravel()), axis=-1)
    idx = np.arange(nx * ny).reshape(ny, nx)
    a = idx[:-1, :-1].ravel(); b = idx[:-1, 1:].ravel(); c = idx[1:, :-1].ravel(); d = idx[1:, 1:].ravel()
    tri = np.stack((np.stack((a, b, d), -1), np.stack((a, d, c), -1)), axis=1).reshape(-1, 3)
    return v, tri, xs, ys


def locate_cartesian(xs, ys, pts):
    """triangle id of points in the mesh of cartesian_mesh (vertex coordinates without offset)"""
    nx, ny = xs.size, ys.size
    i = np.clip(np.searchsorted(xs, pts[:, 0], side='right') - 1, 0, nx - 2)
    j = np.clip(np.searchsorted(ys, pts[:, 1], side='right') - 1, 0, ny - 2)
    u = (pts[:, 0] - xs[i]) / (xs[i + 1] - xs[i])
    w = (pts[:, 1] - ys[j]) / (ys[j + 1] - ys[j])
    return 2 * (j * (nx - 1) + i) + (w > u)


def relax_mesh1(W, H, mesh_size, t0, t1, xy0, xy1, weight, residue_len=0.0, min_num_blocks=2, return_mesh=False, residue_mode='huber'):
    """matcher.py:717-737 with an exact solve: mesh0 (locked, translated by t0), mesh1 (free, translated by t1),
    one link from the matched points (MOVING gear).  Returns the displacement of every mesh1 vertex and, with
    residue_len > 0, the huber residue weight of every match after the relaxation (optimizer.py:174-191)."""
    v, tri, xs, ys = cartesian_mesh(W, H, mesh_size, min_num_blocks=min_num_blocks)
    m0 = fem_ref.RefMesh(v, tri, uid=0)
    m0.apply_translation(t0, fem_ref.GEAR_FIXED)
    m0.locked = True
    m1 = fem_ref.RefMesh(v, tri, uid=1)
    m1.apply_translation(t1, fem_ref.GEAR_FIXED)
    tid0 = locate_cartesian(xs, ys, xy0 - m0.offset(fem_ref.GEAR_MOVING))
    tid1 = locate_cartesian(xs, ys, xy1 - m1.offset(fem_ref.GEAR_MOVING))
    B0 = m0.cart2bary(xy0, fem_ref.GEAR_MOVING, tid0)
    B1 = m1.cart2bary(xy1, fem_ref.GEAR_MOVING, tid1)
    link = fem_ref.RefLink(m0, m1, tid0, tid1, B0, B1, weight=weight)
    before = m1.vertices_w_offset(fem_ref.GEAR_MOVING).copy()
    fem_ref.optimize_linear([m0, m1], [link], exact=True)
    if residue_len > 0:
        # adjust_link_weight_by_residue(relax_first=True) (matcher.py:736, optimizer.py:763-779): a region of the free mesh
        # deformed beyond the (twice converted) cutoff is relaxed before the residues are taken
        cutoff = 1 - 1 / (MAXIMUM_DEFORM_ALLOWED + 1)
        fem_ref.relax_mesh_most_deformed(m1, gear=(fem_ref.GEAR_FIXED, fem_ref.GEAR_MOVING), deform_cutoff=cutoff)
    u = m1.vertices_w_offset(fem_ref.GEAR_MOVING) - before
    if return_mesh:
        return u, m0, m1, link
    if residue_len > 0:
        return u, link.residue_weights((fem_ref.GEAR_MOVING, fem_ref.GEAR_MOVING), residue_mode, residue_len)
    return u


def strain_estimate(W, H, mesh_size, t0, xy0, xy1, weight, min_num_blocks=2, stiffness_lambda=1.0):
    """matcher.py:752-777 with exact solves: fresh mesh pair (mesh0 locked at its translation t0), one link from the
    final matches (INITIAL gear), rigid initialisation of mesh1 (optimize_affine_cascade, optimizer.py:1128-1189,
    svd_clip (1, 1)), anneal, optimize_linear, strain = sqrt(Es / Es0) on the free mesh."""
    v, tri, xs, ys = cartesian_mesh(W, H, mesh_size, min_num_blocks=min_num_blocks)
    m0 = fem_ref.RefMesh(v, tri, uid=0)
    m0.apply_translation(t0, fem_ref.GEAR_FIXED)
    m0.locked = True
    m1 = fem_ref.RefMesh(v, tri, uid=1)
    tid0 = locate_cartesian(xs, ys, xy0)
    tid1 = locate_cartesian(xs, ys, xy1)
    B0 = m0.cart2bary(xy0, fem_ref.GEAR_INITIAL, tid0)
    B1 = m1.cart2bary(xy1, fem_ref.GEAR_INITIAL, tid1)
    link = fem_ref.RefLink(m0, m1, tid0, tid1, B0, B1, weight=weight)
    return strain_from_link(m0, m1, link, stiffness_lambda=stiffness_lambda)[0]


def strain_from_link(m0, m1, link, stiffness_lambda=1.0):
    """core of strain_estimate for a locked m0 (FIXED gear placed), an untouched m1 and their link.
    Returns (strain, Es, Es0, R)."""
    tid0, tid1, B0, B1 = link.tid0, link.tid1, link.B0, link.B1
    # cascade: mesh1 points at INITIAL onto mesh0 points at FIXED
    p1 = m1.bary2cart(tid1, B1, fem_ref.GEAR_INITIAL)
    p0 = m0.bary2cart(tid0, B0, fem_ref.GEAR_FIXED)
    _, R = fem_ref.fit_affine(p0, p1, return_rigid=True, weight=link.total_weight(), svd_clip=(1, 1), avoid_flip=True)
    m1.set_affine(R, gear=(fem_ref.GEAR_INITIAL, fem_ref.GEAR_FIXED))
    m1.anneal_copy(gear=(fem_ref.GEAR_FIXED, fem_ref.GEAR_MOVING))
    fem_ref.optimize_linear([m0, m1], [link], stiffness_lambda=stiffness_lambda, exact=True)
    v0 = m1.vertices(fem_ref.GEAR_FIXED)
    dv = m1.vertices(fem_ref.GEAR_MOVING) - v0
    v0 = v0 - np.mean(v0, axis=0, keepdims=True)
    dv = dv - np.mean(dv, axis=0, keepdims=True)
    St, _ = m1.stiffness_matrix()
    Es = max(0, St.dot(dv.ravel()).dot(dv.ravel()))
    Es0 = max(0, St.dot(v0.ravel()).dot(v0.ravel()))
    return (Es / Es0) ** 0.5, Es, Es0, R


def _crop(img, x0, y0, h, w):
    """h x w window at (x0, y0), zero outside the image (StreamLoader fillval=0 +
    cv2.remap BORDER_CONSTANT, common.py:329-330)."""
    out = np.zeros((h, w), dtype=img.dtype)
    H, W = img.shape
    ya, yb = max(y0, 0), min(y0 + h, H)
    xa, xb = max(x0, 0), min(x0 + w, W)
    if ya < yb and xa < xb:
        out[ya - y0:yb - y0, xa - x0:xb - x0] = img[ya:yb, xa:xb]
    return out


# ------------------------------------------------------------------ deformed mesh1: renderer tiers + link location
def tri_box_intersects(tp, box):
    """shapely `intersects` of closed triangles tp [T, 3, 2] with the closed box (xmin, ymin, xmax, ymax) -- what
    STRtree.query(box, predicate='intersects') answers at renderer.py:405.  Separating axes: the two box axes and the
    three edge normals; touching counts as intersecting."""
    tp = np.asarray(tp, dtype=np.float64)
    x0, y0, x1, y1 = (float(b) for b in box)
    sep = (tp[:, :, 0].max(axis=1) < x0) | (tp[:, :, 0].min(axis=1) > x1) | (tp[:, :, 1].max(axis=1) < y0) | (tp[:, :, 1].min(axis=1) > y1)
    corners = np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]])
    for k in range(3):
        e = tp[:, (k + 1) % 3] - tp[:, k]
        nrm = np.stack((-e[:, 1], e[:, 0]), axis=-1)
        pt = np.einsum('tvj,tj->tv', tp, nrm)
        pb = corners @ nrm.T                                    # [4, T]
        sep |= (pt.max(axis=1) < pb.min(axis=0)) | (pt.min(axis=1) > pb.max(axis=0))
    return ~sep


def _mpl_triangulation(m1):
    import matplotlib.tri
    v0 = m1.vertices(fem_ref.GEAR_MOVING)
    return matplotlib.tri.Triangulation(v0[:, 0], v0[:, 1], triangles=m1.triangles)


def locate_deformed(m1, xy):
    """Mesh.tri_finder (mesh.py:2080-2143) for one contiguous region without collisions: matplotlib's trifinder on the
    MOVING vertices (offset removed from the points); -1 outside the mesh."""
    pts = np.atleast_2d(xy) - m1.offset(fem_ref.GEAR_MOVING)
    return np.asarray(_mpl_triangulation(m1).get_trifinder()(pts[:, 0], pts[:, 1]))


def affine_residue(v1, v0, A):
    return np.max(np.sum((v1 - v0 @ A[:2, :2] - A[-1, :2]) ** 2, axis=-1)) ** 0.5


def _clip_convex(poly, clip):
    """Sutherland-Hodgman: convex polygon `poly` [n, 2] clipped by the convex polygon `clip` [m, 2] (either orientation)"""
    clip = np.asarray(clip, dtype=np.float64)
    if np.sum(clip[:, 0] * np.roll(clip[:, 1], -1) - np.roll(clip[:, 0], -1) * clip[:, 1]) < 0:
        clip = clip[::-1]
    out = [tuple(p) for p in np.asarray(poly, dtype=np.float64)]
    for k in range(clip.shape[0]):
        a, b = clip[k], clip[(k + 1) % clip.shape[0]]
        side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])       # >= 0: inside (counter-clockwise clip)
        inp, out = out, []
        for i in range(len(inp)):
            p, q = inp[i], inp[(i + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp >= 0) != (sq >= 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
        if not out:
            return np.zeros((0, 2))
    return np.array(out)


def _poly_area(p):
    if p.shape[0] < 3:
        return 0.0
    return 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1]))


def render_blocks_mesh1(m1, img, bboxes, tol, return_tiers=False, img_origin=(0, 0), return_mask=False, precise_mask=False):
    """MeshRenderer.from_mesh(mesh1, affine_approx_tol=tol) (renderer.py:47-166) + crop_multiple(bboxes, mode=RENDER_FULL,
    log_sigma=0, remap_interp=INTER_LINEAR) (renderer.py:601-648) for a mesh of one region without collisions over a
    StreamLoader of `img` (fillval 0).  Per block (crop_field, renderer.py:453-563): the global affine when its residue
    is below tol, else the affine fitted to the vertices of the triangles that touch the block when that is below tol,
    else the exact piecewise-linear field (masked outside the mesh).  All fields are rendered as ONE map
    (render_by_subregions, common.py:257-350: one remap origin for the whole stack).
    precise_mask (= log_sigma > 0, renderer.py:491-511): the mask of an affine block follows crop_field_affine
    (renderer.py:437-447) -- the affine image of the box (bbox0 - 0.5) is intersected with the mesh region in IMAGE space;
    when 1 px^2 or more of it lies outside, the mask keeps the pixels whose source point lies in the region.  The region is
    the union of the image-space triangles; the reference's shapely buffer(-0.5).simplify(0.5) of it is not restated
    (shapely is absent): tier numbers 11 / 12 mark such blocks."""
    import matplotlib.tri
    off = m1.offset(fem_ref.GEAR_MOVING).ravel()
    v0 = m1.vertices(fem_ref.GEAR_MOVING)
    v1 = m1.vertices_w_offset(fem_ref.GEAR_INITIAL)
    A_g = fem_ref.fit_affine(v1, v0)
    res_g = affine_residue(v1, v0, A_g)
    tri = None
    tri_img = None
    fx, fy, msk, tiers = [], [], [], []
    for bbox in np.asarray(bboxes):
        bbox0 = np.asarray(bbox, dtype=np.float64) - np.tile(off, 2)
        outwd = round(bbox0[2] - bbox0[0]); outht = round(bbox0[3] - bbox0[1])
        xs = np.linspace(bbox0[0], bbox0[2], num=outwd, endpoint=False, dtype=float)
        ys = np.linspace(bbox0[1], bbox0[3], num=outht, endpoint=False, dtype=float)
        xx, yy = np.meshgrid(xs, ys)
        A = None
        if tol > 0:
            if res_g < tol:
                A, tier = A_g, 1
            else:
                hit = tri_box_intersects(v0[m1.triangles], bbox0 - 0.5)
                if hit.any():
                    idx = np.unique(m1.triangles[hit])
                    _, A_b = fem_ref.fit_affine(v1[idx], v0[idx], return_rigid=True, svd_clip=None)
                    if affine_residue(v1[idx], v0[idx], A_b) < tol:
                        A, tier = A_b, 2
        if A is not None:
            x_f = xx * A[0, 0] + yy * A[1, 0] + A[2, 0]
            y_f = xx * A[0, 1] + yy * A[1, 1] + A[2, 1]
            mk = np.ones_like(x_f, dtype=bool)
            if precise_mask:
                b = bbox0 - 0.5
                box = np.array([[b[0], b[1]], [b[2], b[1]], [b[2], b[3]], [b[0], b[3]]])
                shpbox = box @ A[:2, :2] + A[2, :2]
                tp = v1[m1.triangles]
                near = (tp[:, :, 0].max(1) >= shpbox[:, 0].min()) & (tp[:, :, 0].min(1) <= shpbox[:, 0].max()) & \
                       (tp[:, :, 1].max(1) >= shpbox[:, 1].min()) & (tp[:, :, 1].min(1) <= shpbox[:, 1].max())
                inter = sum(_poly_area(_clip_convex(t3, shpbox)) for t3 in tp[near])
                if _poly_area(shpbox) - inter >= 1:
                    if tri_img is None:
                        tri_img = matplotlib.tri.Triangulation(v1[:, 0], v1[:, 1], triangles=m1.triangles).get_trifinder()
                    mk = np.asarray(tri_img(x_f, y_f)) >= 0
                    tier += 10
        else:
            tier = 3
            if tri is None:
                tri = _mpl_triangulation(m1)
                ix = matplotlib.tri.LinearTriInterpolator(tri, v1[:, 0])
                iy = matplotlib.tri.LinearTriInterpolator(tri, v1[:, 1])
            mx, my = ix(xx, yy), iy(xx, yy)
            bad = np.ma.getmaskarray(mx) | np.ma.getmaskarray(my)
            x_f = np.nan_to_num(np.ma.getdata(mx)); y_f = np.nan_to_num(np.ma.getdata(my))
            mk = ~bad
        fx.append(x_f); fy.append(y_f); msk.append(mk); tiers.append(tier)
    map_x = np.concatenate(fx, axis=0); map_y = np.concatenate(fy, axis=0); mask = np.concatenate(msk, axis=0)
    out = np.zeros(map_x.shape, dtype=np.float32)
    if mask.any():
        xmin = np.floor(map_x[mask].min()) - 4; ymin = np.floor(map_y[mask].min()) - 4
        val = remap_origin(img, map_x, map_y, (int(xmin), int(ymin)), img_origin)
        out[mask] = val[mask]
    out = out.reshape(len(fx), -1, map_x.shape[1])
    if return_mask:
        return out, mask.reshape(out.shape), np.array(tiers)
    return (out, np.array(tiers)) if return_tiers else out


def remap_origin(img, map_x, map_y, origin, img_origin=(0, 0)):
    """common.remap on the sub-image img_loader.crop((xmin, ymin, xmax, ymax)) (zero outside the image, whose pixel (0, 0)
    sits at img_origin) with maps made relative to its integer origin in float32 (common.py:329-330).  float32 images:
    ncc_ref.remap_bilinear_cv on the zero-extended image.  uint8 images: cv2's fixed-point bilinear path -- weights of
    the 1/32-px phases scaled by 2^15 as int16 (the unit weight saturates to 32767, the table's sum fix-up puts the missing
    1 on the last tap), (sum + 2^14) >> 15 (FixedPtCast); restated from the published OpenCV algorithm, cv2 itself is
    absent from the image (parity unpinned for this function)."""
    mxt = (map_x - origin[0]).astype(np.float32); myt = (map_y - origin[1]).astype(np.float32)
    H, W = img.shape
    sx = np.rint(mxt * np.float32(32)).astype(np.int64); sy = np.rint(myt * np.float32(32)).astype(np.int64)
    ix = (sx >> 5) + int(origin[0]) - int(img_origin[0]); iy = (sy >> 5) + int(origin[1]) - int(img_origin[1])
    a = sx & 31; b = sy & 31

    def tap(src, yy_, xx_, zero):
        inside = (yy_ >= 0) & (yy_ < H) & (xx_ >= 0) & (xx_ < W)
        return np.where(inside, src[np.clip(yy_, 0, H - 1), np.clip(xx_, 0, W - 1)], zero)
    if img.dtype == np.uint8:
        src = img.astype(np.int64)
        w00 = (32 - a) * (32 - b) * 32; w01 = a * (32 - b) * 32; w10 = (32 - a) * b * 32; w11 = a * b * 32
        unit = (a == 0) & (b == 0)
        w00 = np.where(unit, 32767, w00); w11 = np.where(unit, 1, w11)
        acc = tap(src, iy, ix, 0) * w00 + tap(src, iy, ix + 1, 0) * w01 + tap(src, iy + 1, ix, 0) * w10 + tap(src, iy + 1, ix + 1, 0) * w11
        return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.float32)
    ax = a.astype(np.float32) * np.float32(1 / 32); ay = b.astype(np.float32) * np.float32(1 / 32)
    one = np.float32(1)
    w00 = (one - ay) * (one - ax); w01 = (one - ay) * ax; w10 = ay * (one - ax); w11 = ay * ax
    img = np.asarray(img, dtype=np.float32)
    z = np.float32(0)
    return ((tap(img, iy, ix, z) * w00 + tap(img, iy, ix + 1, z) * w01) + tap(img, iy + 1, ix, z) * w10) + tap(img, iy + 1, ix + 1, z) * w11


def bboxes_mesh_renderer_matcher(mesh0, mesh1, img0, img1, bboxes0, bboxes1, sigma=0.0, conf_mode=ncc_ref.FFT_CONF_MIRROR,
                                 pad=True, subpixel=False, affine_approx_tol=0.0, img_origin0=(0, 0), img_origin1=(0, 0),
                                 return_stacks=False):
    """matcher.py:781-861 for blocks of one size, batch_size None, meshes of one region without collisions over
    StreamLoaders of img0 / img1 (fillval 0): both stacks rendered (render_blocks_mesh1 above), DoG with the stack masks
    when sigma > 0 (renderer.py:632-641), xcorr_fft, block displacement -> point pair (matcher.py:840-849)."""
    stacks, masks = [], []
    for m, img, bb, org in ((mesh0, img0, bboxes0, img_origin0), (mesh1, img1, bboxes1, img_origin1)):
        st, mk, _ = render_blocks_mesh1(m, img, bb, affine_approx_tol, img_origin=org, return_mask=True, precise_mask=sigma > 0)
        if sigma > 0:
            st = ncc_ref.masked_dog_filter(st, sigma, mask=mk)
        stacks.append(st); masks.append(mk)
    dx, dy, conf = ncc_ref.xcorr_fft(stacks[0], stacks[1], conf_mode=conf_mode, pad=pad, subpixel=subpixel)
    xy0, xy1 = ncc_ref.block_points(np.asarray(bboxes0), np.asarray(bboxes1), dx, dy)
    if return_stacks:
        return xy0, xy1, conf, stacks, masks
    return xy0, xy1, conf


def relax_deformed(m0, m1, xs, ys, xy0, xy1, weight, residue_len, resolve, residue_mode='huber'):
    """matcher.py:717-742 on a mesh pair that keeps its state: link from the matched points in the MOVING gears (mesh0 is
    the translated grid, mesh1 is located through its deformed triangles; points outside are dropped, optimizer.py:
    51-82), optimize_linear (exact), then -- residue_len > 0 -- relax_higly_deformed + huber residue weights
    (optimizer.py:763-790) and, when a weight changed and `resolve`, a second solve (matcher.py:737-741).
    Returns (link, kept rows)."""
    tid1 = locate_deformed(m1, xy1)
    ok = tid1 >= 0
    xy0, xy1, weight, tid1 = xy0[ok], xy1[ok], weight[ok], tid1[ok]
    if not ok.any():
        return None, ok
    tid0 = locate_cartesian(xs, ys, xy0 - m0.offset(fem_ref.GEAR_MOVING))
    B0 = m0.cart2bary(xy0, fem_ref.GEAR_MOVING, tid0)
    B1 = m1.cart2bary(xy1, fem_ref.GEAR_MOVING, tid1)
    link = fem_ref.RefLink(m0, m1, tid0, tid1, B0, B1, weight=weight)
    fem_ref.optimize_linear([m0, m1], [link], exact=True)
    if residue_len > 0:
        cutoff = 1 - 1 / (MAXIMUM_DEFORM_ALLOWED + 1)
        fem_ref.relax_mesh_most_deformed(m1, gear=(fem_ref.GEAR_FIXED, fem_ref.GEAR_MOVING), deform_cutoff=cutoff)
        rw = link.residue_weights((fem_ref.GEAR_MOVING, fem_ref.GEAR_MOVING), residue_mode, residue_len)
        if np.any(rw != link.residue_weight):
            link.residue_weight = rw
            if resolve:
                fem_ref.optimize_linear([m0, m1], [link], exact=True)
    return link, ok


def photometric(raw0, raw1, g0, g1, tx0, ty0, mask0_g=None, mask1_g=None):
    """matcher.py:279-314 with sigma > 0: mean grey level of the two raw coarse images and mean |DoG| of the filtered ones
    over the overlap of the translated bounding boxes (both masks applied)."""
    txx, tyy = int(tx0), int(ty0)
    bb0 = (txx, tyy, g0.shape[1] + txx, g0.shape[0] + tyy)
    bb1 = (0, 0, g1.shape[1], g1.shape[0])
    (xa, ya, xb, yb), _ = ncc_ref.intersect_bbox(bb0, bb1)
    xa, ya, xb, yb = int(xa), int(ya), int(xb), int(yb)
    i0 = (slice(ya - tyy, yb - tyy), slice(xa - txx, xb - txx))
    i1 = (slice(ya, yb), slice(xa, xb))
    m0 = np.ones((yb - ya, xb - xa), dtype=bool) if mask0_g is None else mask0_g[i0]
    m1 = np.ones((yb - ya, xb - xa), dtype=bool) if mask1_g is None else mask1_g[i1]
    mp = m0 & m1
    if np.sum(m0) <= 3:
        return None
    return (np.mean(raw0[i0][mp]), np.mean(raw1[i1][mp]), np.mean(np.abs(g0[i0][mp])), np.mean(np.abs(g1[i1][mp])))


def match_pair(strip0, strip1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, min_num_blocks=2,
               conf_mode=ncc_ref.FFT_CONF_MIRROR, residue_len=5.0, mask0=None, mask1=None, compute_photometric=False, spacings=None, residue_mode='huber',
               fine_downsample=1, block_script=None, script_start=None):
    """strip0/strip1: uint8 H x W overlap strips (mask0/mask1: bool, True = valid pixel).
    Returns dict(tx, ty, conf0, xy0, xy1, weight, needs_host, strain, phtm, ...).  tx / ty: the global translation in pixels
    of the FINE images (the strips themselves unless fine_downsample != 1); xy0 / xy1 in pixels of the strips.
    block_script (golden G24): a callable (round, bboxes0, bboxes1) -> (dx, dy, conf) that stands in for crop + NCC of every
    round, with script_start = (H, W, tx, ty) in place of the image stages in front of the loop (strip0 / strip1 are not looked
    at): the loop BETWEEN the block matches -- blocks, links, rigid / deformed relaxation, residue weights, the walk, final
    matches, strain -- then runs on the same inputs as the reference's own loop driven by the same script."""
    cd, fd = coarse_downsample, fine_downsample
    if block_script is not None:
        return _match_pair_loop(None, None, script_start[0], script_start[1], float(script_start[2]), float(script_start[3]), np.asarray(spacings, dtype=np.float64),
                                float(residue_len), dict(tx=float(script_start[2]), ty=float(script_start[3]), conf0=1.0, xy0=None, xy1=None, weight=None,
                                                         needs_host=False, strain=0.05, phtm=None),
                                conf_thresh, min_num_blocks, conf_mode, residue_mode, 1, block_script)

    def shrink(img, mk, f):
        if f == 1:
            return img, mk
        small = ncc_ref.area_downsample2(img) if f == 0.5 else ncc_ref.area_resize(img, f)
        # cv2.resize(mask, fx=f, INTER_NEAREST) (matcher.py:257-264): source pixel floor(dst / f) (UNPINNED)
        return small, (None if mk is None else ncc_ref.nearest_resize_mask(mk, f)[:small.shape[0], :small.shape[1]])
    g0, mask0_g = shrink(strip0, mask0, cd)
    g1, mask1_g = shrink(strip1, mask1, cd)
    raw0, raw1 = g0, g1
    g0 = ncc_ref.masked_dog_filter(g0, sigma * cd, mask=mask0_g)          # matcher.py:273-274
    g1 = ncc_ref.masked_dog_filter(g1, sigma * cd, mask=mask1_g)
    tx, ty, conf0 = ncc_ref.global_translation_matcher(g0, g1, conf_mode=conf_mode, conf_thresh=conf_thresh)
    res = dict(tx=tx * fd / cd, ty=ty * fd / cd, conf0=conf0, xy0=None, xy1=None, weight=None,                  # matcher.py:338-339
               needs_host=False, strain=0.05, phtm=None)
    if conf0 < conf_thresh:                                                  # matcher.py:277-278
        return res
    if compute_photometric:
        res['phtm'] = photometric(raw0, raw1, g0, g1, tx, ty, mask0_g, mask1_g)
    if fd == cd:                                                             # matcher.py:315-317
        f0, f1 = g0, g1
    else:
        b0, mask0_f = shrink(strip0, mask0, fd)                              # matcher.py:319-335
        b1, mask1_f = shrink(strip1, mask1, fd)
        f0 = ncc_ref.masked_dog_filter(b0, sigma * fd, mask=mask0_f)         # matcher.py:336-337
        f1 = ncc_ref.masked_dog_filter(b1, sigma * fd, mask=mask1_f)
    # spacings follow the shapes of the strips as given, then scale with the fine images; so does residue_len (matcher.py:243-251, 341, 352)
    spacings = (ncc_ref.auto_spacings(strip0.shape, strip1.shape) if spacings is None else np.asarray(spacings, dtype=np.float64)) * fd
    residue_len = residue_len * fd
    return _match_pair_loop(f0, f1, f0.shape[0], f0.shape[1], res['tx'], res['ty'], spacings, residue_len, res, conf_thresh, min_num_blocks, conf_mode,
                            residue_mode, fd, None)


def _match_pair_loop(f0, f1, H, W, tx, ty, spacings, residue_len, res, conf_thresh, min_num_blocks, conf_mode, residue_mode, fd, block_script):
    """iterative_xcorr_matcher_w_mesh as stitching_matcher calls it (matcher.py:353-364, 430-778) on the band-passed fine images"""
    spacings = np.sort(spacings)[::-1]
    bbox0 = (-0.5 + tx, -0.5 + ty, W - 0.5 + tx, H - 0.5 + ty)              # Mesh.from_bbox + apply_translation
    pad = True
    itx, ity = int(round(tx)), int(round(ty))
    t1 = np.zeros(2)                                                         # rigid motion mesh1 acquired so far
    last = None
    deformed = None                                                          # (m0, m1, xs, ys) once mesh1 is not a translated grid
    mesh_size = float(np.min(spacings))
    for rnd, sp in enumerate(spacings):
        is_last = rnd == spacings.size - 1
        mnb = min_num_blocks if is_last else 1
        if deformed is None:
            bbox1 = (-0.5 + t1[0], -0.5 + t1[1], W - 0.5 + t1[0], H - 0.5 + t1[1])
        else:
            vm = deformed[1].vertices_w_offset(fem_ref.GEAR_MOVING)          # Mesh.bbox(gear=MOVING), matcher.py:877
            bbox1 = (vm[:, 0].min(), vm[:, 1].min(), vm[:, 0].max(), vm[:, 1].max())
        i1x, i1y = int(round(t1[0])), int(round(t1[1]))
        bb0, bb1 = ncc_ref.distributor_cartesian_bbox(bbox0, bbox1, sp, min_num_blocks=mnb, zorder=True)
        if bb0 is None:
            if rnd == 0:
                return res
            break
        if block_script is not None:
            out = block_script(rnd, bb0, bb1)
            if len(out) == 5:                                # the script names the blocks of this round itself (G24 'rigid': see the test)
                bb0, bb1 = np.asarray(out[3]), np.asarray(out[4])
            dx, dy, cf = out[:3]
            res.setdefault('rounds', []).append(dict(bboxes0=bb0, bboxes1=bb1, pad=pad, subpixel=is_last,
                                                      field1=(np.tile(t1, (1, 1)) if deformed is None else
                                                              deformed[1].vertices_w_offset(fem_ref.GEAR_MOVING) - deformed[1].vertices_w_offset(fem_ref.GEAR_INITIAL))))
        else:
            h = int(bb0[0, 3] - bb0[0, 1]); w = int(bb0[0, 2] - bb0[0, 0])
            s0 = np.stack([_crop(f0, int(b[0]) - itx, int(b[1]) - ity, h, w) for b in bb0])
            if deformed is None:
                s1 = np.stack([_crop(f1, int(b[0]) - i1x, int(b[1]) - i1y, h, w) for b in bb1])
            else:
                tol = 0.1 if is_last else max(1, 0.02 * sp)                      # matcher.py:578-603 (affine_approximated_render)
                s1, tiers = render_blocks_mesh1(deformed[1], f1, bb1, tol, return_tiers=True)
                res.setdefault('tiers', []).append(tiers)
            dx, dy, cf = ncc_ref.xcorr_fft(s0, s1, conf_mode=conf_mode, pad=pad, subpixel=is_last)
        xy0, xy1 = ncc_ref.block_points(bb0, bb1, dx, dy)
        keep = cf > conf_thresh
        if not np.any(keep):
            if rnd == 0:
                return res
            break                                            # matcher.py:671-679: keep the links of the last good round
        xy0, xy1, wt = xy0[keep], xy1[keep], cf[keep]
        max_dis = np.max(np.sum((xy0 - xy1) ** 2, axis=-1)) ** 0.5
        if not is_last:
            next_pos = np.searchsorted(-spacings, -4 * max_dis) - 1           # matcher.py:689-716
            pad = (min(next_pos, rnd + 1) > rnd + 1) if next_pos > rnd else True
        if deformed is not None:
            m0, m1, xs, ys = deformed
            if max_dis > 0.1:
                link, ok = relax_deformed(m0, m1, xs, ys, xy0, xy1, wt, residue_len, resolve=not is_last, residue_mode=residue_mode)
            else:                                            # link only (matcher.py:717), no relaxation
                tid1 = locate_deformed(m1, xy1)
                ok = tid1 >= 0
                link = None
                if ok.any():
                    tid0 = locate_cartesian(xs, ys, xy0[ok] - m0.offset(fem_ref.GEAR_MOVING))
                    link = fem_ref.RefLink(m0, m1, tid0, tid1[ok], m0.cart2bary(xy0[ok], fem_ref.GEAR_MOVING, tid0),
                                           m1.cart2bary(xy1[ok], fem_ref.GEAR_MOVING, tid1[ok]), weight=wt[ok])
            if not is_last:                                  # the field the next round renders through
                res['mesh1_field'] = m1.vertices_w_offset(fem_ref.GEAR_MOVING) - m1.vertices_w_offset(fem_ref.GEAR_INITIAL)
            if link is None:                                 # matcher.py:719-723: no link could be made
                if rnd == 0:
                    return res
                break
            last = (m0.bary2cart(link.tid0, link.B0, fem_ref.GEAR_INITIAL) + np.array([tx, ty]),
                    m1.bary2cart(link.tid1, link.B1, fem_ref.GEAR_INITIAL), link.total_weight(), max_dis)
            continue
        last = (xy0, xy1 - t1, wt, max_dis)                   # INITIAL gear: barycentric coordinates are fixed at link creation
        if is_last and max_dis > 0.1 and residue_len > 0:
            # matcher.py:725-737: relax, then Link.weight = conf * huber residue weight (no second solve: sp_indx ran out)
            _, rw = relax_mesh1(W, H, mesh_size, (tx, ty), t1, xy0, xy1, wt, residue_len=residue_len,
                                min_num_blocks=min_num_blocks, residue_mode=residue_mode)
            last = (xy0, xy1 - t1, wt * rw, max_dis)
        if not is_last and max_dis > 0.1:
            # mesh relaxation (matcher.py:725-729), exact solve on the cartesian mesh pair
            u, m0, m1, link = relax_mesh1(W, H, mesh_size, (tx, ty), t1, xy0, xy1, wt, min_num_blocks=min_num_blocks, return_mesh=True)
            um = u.mean(axis=0)
            if np.abs(u - um).max() < 1e-6 and np.abs(um - np.round(um)).max() < 1e-6:
                t1 = t1 + np.round(um)                   # a rigid integer translation: crops stay exact
            else:
                # any other field: mesh1 keeps its MOVING gear.  Residue step of matcher.py:730-741 on that state
                if residue_len > 0:
                    cutoff = 1 - 1 / (MAXIMUM_DEFORM_ALLOWED + 1)
                    fem_ref.relax_mesh_most_deformed(m1, gear=(fem_ref.GEAR_FIXED, fem_ref.GEAR_MOVING), deform_cutoff=cutoff)
                    rw = link.residue_weights((fem_ref.GEAR_MOVING, fem_ref.GEAR_MOVING), residue_mode, residue_len)
                    if np.any(rw != link.residue_weight):
                        link.residue_weight = rw
                        fem_ref.optimize_linear([m0, m1], [link], exact=True)
                        last = (xy0, xy1 - t1, link.total_weight(), max_dis)
                _, _, xs, ys = cartesian_mesh(W, H, mesh_size, min_num_blocks=min_num_blocks)
                deformed = (m0, m1, xs, ys)
                res['deformed'] = True
                res['mesh1_field'] = m1.vertices_w_offset(fem_ref.GEAR_MOVING) - m1.vertices_w_offset(fem_ref.GEAR_INITIAL)
    if last is not None and residue_mode == 'threshold':
        keep_rows = np.asarray(last[2]) > 0                  # Link.mask (optimizer.py:399-402): use_mask=True at matcher.py:749-751
        last = (last[0][keep_rows], last[1][keep_rows], last[2][keep_rows], last[3])
        if not keep_rows.any():
            last = None
    if last is not None:
        res['xy0'] = last[0] - np.array([tx, ty])
        res['xy1'] = last[1]
        res['weight'] = last[2]
        res['max_dis'] = last[3]
        # matcher.py:752-777 (compute_strain defaults to True)
        res['strain'] = strain_estimate(W, H, float(np.min(spacings)), (tx, ty), res['xy0'], res['xy1'], res['weight'],
                                        min_num_blocks=min_num_blocks)
        if fd != 1:                                                          # matcher.py:365-367
            res['xy0'] = ncc_ref.scale_coordinates(res['xy0'], 1 / fd)
            res['xy1'] = ncc_ref.scale_coordinates(res['xy1'], 1 / fd)
    return res
