"""TEST INFRASTRUCTURE ONLY (tests/, smoke and bench's cpu_baseline leg may import this; the product never does).

Numpy restatement of the reference's region-aware block distributor and of the section matcher built on it:

* ``distribute_matching_blocks`` / ``_region2grid_cartesian``      feabas/matcher.py:894-1016, 1019-1043
* ``section_matcher`` -> ``iterative_xcorr_matcher_w_mesh``         feabas/matcher.py:370-427, 430-778
  (no initial matches, linear materials, distributor 'cartesian_region')

The reference builds the regions with shapely polygons (absent from the build image: PARITY UNPINNED for this module, like
every composite of pipeline_ref).  What shapely computes is restated on point predicates, which is exact for everything the
result depends on except two things, both said where they occur:

* ``Polygon.buffer(-d)`` rounds reflex corners with 8-segment quarter circles; the predicate here is the true Euclidean
  distance to the outline (the regions differ by slivers of at most 0.5 % of d next to such corners);
* the lattice of a connected part is anchored at GEOS's ``representative_point`` (InteriorPointArea: the middle of the widest
  stretch of the scan line halfway between the two vertex ordinates next to the middle of the bounds).  The vertices of a
  buffered / clipped polygon exist only inside GEOS; ``representative_point_raster`` evaluates the same rule on a fine raster
  of the predicate, and a caller that wants to compare lattices block for block hands the anchors in (``anchor_blocks``:
  any block of the other implementation fixes the lattice phase of the part it lies in).
"""
import numpy as np
from scipy import ndimage

from . import fem_ref, ncc_ref, pipeline_ref

GEAR_INITIAL, GEAR_FIXED, GEAR_MOVING = fem_ref.GEAR_INITIAL, fem_ref.GEAR_FIXED, fem_ref.GEAR_MOVING
DEFAULT_AVG_DEFORM = 0.05
MAXIMUM_DEFORM_ALLOWED = pipeline_ref.MAXIMUM_DEFORM_ALLOWED


# ------------------------------------------------------------------ regions as predicates
def boundary_segments(v, t):
    """outline of Mesh.shapely_regions (mesh.py: union of the triangles): the edges that belong to exactly one triangle"""
    e = np.sort(np.concatenate((t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]), axis=0), axis=1)
    u, cnt = np.unique(e, axis=0, return_counts=True)
    return v[u[cnt == 1]]                                                  # [E, 2, 2]


def _locate(v, t, pts):
    """triangle of every point (-1 outside), closed triangles: matplotlib's trapezoid map"""
    import matplotlib.tri
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    return np.asarray(matplotlib.tri.Triangulation(v[:, 0], v[:, 1], triangles=t).get_trifinder()(pts[:, 0], pts[:, 1]))


def _segment_distance(pts, segs):
    a, b = segs[:, 0], segs[:, 1]
    ab = b - a
    l2 = np.maximum(np.sum(ab * ab, axis=1), 1e-300)
    out = np.empty(pts.shape[0])
    step = max(1, int(2e6 // max(1, a.shape[0])))
    for s in range(0, pts.shape[0], step):
        p = pts[s:s + step]
        tt = np.clip(np.einsum('pej,ej->pe', p[:, None, :] - a[None], ab) / l2[None], 0.0, 1.0)
        d2 = np.sum((p[:, None, :] - (a[None] + tt[..., None] * ab[None])) ** 2, axis=2)
        out[s:s + step] = np.sqrt(d2.min(axis=1))
    return out


class CommonRegion:
    """reg_crx0 = region0.intersection(region1) (matcher.py:944-946) and what is cut out of it, as a predicate:
    ``contains(pts, erode, only, exclude)`` <=> pts in reg_crx0.buffer(-erode), in ANY of the `only` triangle sets (a refinement
    level's material regions, matcher.py:963-976) and in NONE of the `exclude` sets (``covered``, matcher.py:977-996).
    only / exclude: lists of (mesh index, boolean triangle mask)."""

    def __init__(self, v0, t0, v1, t1):
        self.vt = ((np.asarray(v0, dtype=np.float64), np.asarray(t0)), (np.asarray(v1, dtype=np.float64), np.asarray(t1)))
        self.segs = np.concatenate([boundary_segments(v, t) for v, t in self.vt], axis=0)
        lo = np.maximum(self.vt[0][0].min(axis=0), self.vt[1][0].min(axis=0))
        hi = np.minimum(self.vt[0][0].max(axis=0), self.vt[1][0].max(axis=0))
        self.bbox = np.concatenate((lo, hi))
        self.valid = bool(np.all(hi > lo))

    def contains(self, pts, erode=0.0, only=None, exclude=None):
        pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
        tids = [_locate(v, t, pts) for v, t in self.vt]
        ok = (tids[0] >= 0) & (tids[1] >= 0)
        for sets, want in ((only, True), (exclude, False)):
            if sets is None:
                continue
            hit = np.zeros(pts.shape[0], dtype=bool)
            for k, mask in sets:
                hit |= (tids[k] >= 0) & np.asarray(mask, dtype=bool)[np.maximum(tids[k], 0)]
            ok &= hit if want else ~hit
        if erode > 0 and ok.any():
            idx = np.flatnonzero(ok)
            ok[idx] = _segment_distance(pts[idx], self.segs) >= erode
        return ok

    def raster(self, res, **kw):
        x0, y0, x1, y1 = self.bbox
        cells = ((x1 - x0) / res) * ((y1 - y0) / res)
        if not np.isfinite(cells) or cells > 1e8:                              # (a runaway mesh must not take the host down)
            raise ValueError(f'common region of {x1 - x0:.3g} x {y1 - y0:.3g} px at raster step {res:.3g}')
        xs = np.arange(x0 + 0.5 * res, x1, res); ys = np.arange(y0 + 0.5 * res, y1, res)
        xx, yy = np.meshgrid(xs, ys)
        return xs, ys, self.contains(np.stack((xx.ravel(), yy.ravel()), axis=-1), **kw).reshape(yy.shape)


def representative_point_raster(xs, ys, part):
    """GEOS InteriorPointArea on a raster of one polygon: scan line through the middle of the bounds (GEOS moves it to the
    nearest ordinate halfway between two vertices; on a raster the row of the middle itself), the widest stretch inside,
    its middle."""
    rr, cc = np.nonzero(part)
    res = xs[1] - xs[0] if xs.size > 1 else 1.0
    y_mid = 0.5 * ((ys[rr.min()] - 0.5 * res) + (ys[rr.max()] + 0.5 * res))
    row = int(np.clip(np.round((y_mid - ys[0]) / res), rr.min(), rr.max()))
    run = np.flatnonzero(part[row])
    if run.size == 0:
        row = rr[np.argmax(np.bincount(rr)[rr])]
        run = np.flatnonzero(part[row])
    brk = np.flatnonzero(np.diff(run) > 1)
    starts = np.concatenate(([0], brk + 1)); ends = np.concatenate((brk, [run.size - 1]))
    w = int(np.argmax(ends - starts))
    return 0.5 * (xs[run[starts[w]]] + xs[run[ends[w]]]), (y_mid if abs(ys[row] - y_mid) <= 0.5 * res + 1e-9 else ys[row])


def region2grid_cartesian(region, spacing, res, anchor_points=None, **pred):
    """matcher.py:1019-1043.  One lattice per connected part (``region.geoms``): anchored at the part's representative point
    -- or, when ``anchor_points`` are given, at the first of them that lies in the part --, spanning the part's bounds, kept
    where it lies in the part (MultiPoint.intersection).  unary_union of the parts' points: sorted by (x, y), duplicates merged.
    The parts, their bounds and membership of a lattice point in a part come from a raster of step `res` (labels, 8-connected);
    membership in the region itself is tested exactly."""
    xs, ys, msk = region.raster(res, **pred)
    if not msk.any():
        return None
    lab, nlab = ndimage.label(msk, structure=np.ones((3, 3), dtype=bool))

    # a point of the region that no raster cell sees (a spur thinner than the raster) belongs to the part the spur hangs on:
    # the label of the nearest cell inside the region
    near = lab if msk.all() else lab[tuple(ndimage.distance_transform_edt(lab == 0, return_distances=False, return_indices=True))]

    def label_of(p):
        ci = np.clip(np.round((p[:, 0] - xs[0]) / res).astype(int), 0, xs.size - 1)
        ri = np.clip(np.round((p[:, 1] - ys[0]) / res).astype(int), 0, ys.size - 1)
        return near[ri, ci]
    if anchor_points is not None and len(anchor_points):
        anchor_points = np.asarray(anchor_points, dtype=np.float64).reshape(-1, 2)
        anchor_points = anchor_points[region.contains(anchor_points, **pred)]          # only points of THIS region can carry its phase
    a_lab = label_of(anchor_points) if anchor_points is not None and len(anchor_points) else None
    territory = ndimage.find_objects(near)
    cntrs = []
    for k in range(1, nlab + 1):
        part = lab == k
        rr, cc = np.nonzero(part)
        # reg.bounds (matcher.py:1027) only say how far the lattice reaches (its phase is the representative point's): the bounds
        # of the part's territory (all raster cells nearer to it than to another part) hold the part with every spur of it
        ty, tx = territory[k - 1]
        bx0, by0, bx1, by1 = (float(b) for b in region.bbox)               # (the last raster cell may end short of the bounds)
        rx_mn = bx0 if tx.start == 0 else xs[tx.start] - 0.5 * res
        rx_mx = bx1 if tx.stop == xs.size else xs[tx.stop - 1] + 0.5 * res
        ry_mn = by0 if ty.start == 0 else ys[ty.start] - 0.5 * res
        ry_mx = by1 if ty.stop == ys.size else ys[ty.stop - 1] + 0.5 * res
        if a_lab is not None and np.any(a_lab == k):
            rx, ry = anchor_points[np.flatnonzero(a_lab == k)[0]]
        else:
            rx, ry = representative_point_raster(xs, ys, part)
        gx0 = rx - ((rx - rx_mn) // spacing) * spacing                     # matcher.py:1030-1031
        gy0 = ry - ((ry - ry_mn) // spacing) * spacing
        gxx, gyy = np.meshgrid(np.arange(gx0, rx_mx, spacing), np.arange(gy0, ry_mx, spacing))
        rv = np.stack((gxx.ravel(), gyy.ravel()), axis=-1)
        if rv.shape[0] == 0:
            continue
        rv = rv[label_of(rv) == k]
        if rv.shape[0]:
            rv = rv[region.contains(rv, **pred)]
        if rv.shape[0]:
            cntrs.append(rv)
    if not cntrs:
        return None
    pts = np.concatenate(cntrs, axis=0)
    pts = pts[np.lexsort((pts[:, 1], pts[:, 0]))]
    if pts.shape[0] > 1:
        pts = pts[np.concatenate(([True], np.any(pts[1:] != pts[:-1], axis=1)))]
    return pts


def distribute_matching_blocks(v0, t0, v1, t1, spacing, refine_mode=2, shrink_factor=1, refine_box_exp=0.5, min_box_side=5,
                               max_box_side=np.inf, min_boundary_distance=0, zorder=True, materials=None, res=1.0,
                               anchor_blocks=None):
    """matcher.py:894-1016, distributor 'cartesian_region'.  v / t: vertices (in the gear the blocks are laid out in) and
    triangles of the two meshes, already restricted to what is rendered (render_weight_threshold, matcher.py:937-941).
    materials: per mesh ``(material_ids [T], {name: (uid, area_constraint)})`` or None (no refinement regions).
    res: raster step of the areas / connected parts (the polygons' exact areas only enter through the one-half test of the
    boundary-distance loop); None: a quarter of the lattice step of each level, at least 1 -- the raster the product documents
    for itself (a spur of the region thinner than the raster is attached to the nearest part it can see).  anchor_blocks: bboxes (of mesh 0) of another implementation whose lattice phase is taken over
    part by part (module docstring).  Returns (bboxes0, bboxes1)."""
    whole = CommonRegion(v0, t0, v1, t1)
    empty = (np.empty((0, 4)), np.empty((0, 4)))
    res_fixed = res
    if not whole.valid or not whole.raster(max(spacing / 4.0, 1.0) if res is None else res)[2].any():
        return empty
    if not hasattr(shrink_factor, '__len__'):
        shrink_factor = (shrink_factor, shrink_factor)
    else:                                                                  # matcher.py:951-956
        def mean_area(v, t):
            p = v[t]
            return np.abs(0.5 * ((p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0]))).sum() / t.shape[0]
        shrink_factor = (max(shrink_factor), min(shrink_factor)) if mean_area(v0, t0) > mean_area(v1, t1) else (min(shrink_factor), max(shrink_factor))
    regs = {}                                                              # area factor -> list of triangle sets (None: reg_crx0 itself)
    if refine_mode == 0 or refine_mode == 2:
        regs[1.0] = [None]
    if refine_mode != 0 and materials is not None:
        for k, mat in enumerate(materials):
            if mat is None:
                continue
            ids, table = mat
            for name, (uid, factor) in table.items():
                if ('refine' not in name) and (factor == 0 or factor >= 1):
                    continue
                mask = np.asarray(ids) == uid
                if mask.any():
                    regs.setdefault(float(factor), []).append((k, mask))
    anchors = None if anchor_blocks is None else 0.5 * (np.asarray(anchor_blocks)[:, :2] + np.asarray(anchor_blocks)[:, 2:])
    out0, out1 = [], []
    covered = []                                                           # triangle sets of the finer levels; None once reg_crx0 itself was a level
    for factor in sorted(regs):
        if covered is None:
            break                                                          # everything is covered: region_crx is empty from here on
        spc = spacing * factor
        shrnk0 = factor ** (refine_box_exp - 1)
        res = max(spc / 4.0, 1.0) if res_fixed is None else res_fixed
        sets = None if any(s is None for s in regs[factor]) else regs[factor]
        area_r = float(whole.raster(res, only=sets)[2].sum()) * res * res
        pred = dict(only=sets, exclude=covered or None, erode=0.0)
        if min_boundary_distance > 0 and area_r > 0:
            bound_coeff = 1.0
            while True:                                                    # matcher.py:985-994
                pred['erode'] = min_boundary_distance * shrnk0 * bound_coeff
                area_c = float(whole.raster(res, **pred)[2].sum()) * res * res
                if area_c >= 0.5 * area_r:
                    break
                bound_coeff *= 0.3 / (1 - area_c / area_r)
                if bound_coeff < 0.1:
                    pred['erode'] = 0.0
                    break
        covered = None if sets is None else covered + sets
        if area_r == 0:
            continue
        sides = (spc * shrnk0 * np.array(shrink_factor, dtype=np.float64)).clip(min_box_side, max_box_side)
        h0, h1 = np.ceil(sides[0] / 2), np.ceil(sides[1] / 2)
        level_anchors = None if anchors is None else anchors[np.asarray(anchor_blocks)[:, 2] - np.asarray(anchor_blocks)[:, 0] == 2 * h0]
        cntrs = region2grid_cartesian(whole, spc, res, anchor_points=level_anchors, **pred)
        if cntrs is None:
            continue
        b0 = np.concatenate((cntrs - h0, cntrs + h0), axis=-1)
        b1 = np.concatenate((cntrs - h1, cntrs + h1), axis=-1)
        if zorder:
            x_rnd = np.round((cntrs[:, 0] - cntrs[:, 0].min()) / spc)
            y_rnd = np.round((cntrs[:, 1] - cntrs[:, 1].min()) / spc)
            idx = ncc_ref.z_order(np.stack((x_rnd, y_rnd), axis=-1))
            b0, b1 = b0[idx], b1[idx]
        out0.append(b0); out1.append(b1)
    if not out0:
        return empty
    return np.concatenate(out0, axis=0), np.concatenate(out1, axis=0)


# ------------------------------------------------------------------ the pair of a section matcher
def _floating_translations(Ad, tol=1e-5):
    """the translations that lie in the null space of a 2-DoF-per-vertex system: for every connected component of the vertex
    graph of A and each axis, the indicator t with max |A t| <= tol x the component's largest diagonal entry (a component that a
    link ties to a locked mesh has row sums of the order of the link weights).  Returns a list of DoF index arrays.  The
    reference's matrices carry float32 noise: t^T A t is +-1e-10 of the largest eigenvalue, not zero, and the
    soft rotation of a floating pair sits only ~50 x above that -- an eigenvalue threshold cannot tell them apart, the
    structure can."""
    from scipy import sparse
    from scipy.sparse import csgraph
    n = Ad.shape[0]
    S = sparse.coo_matrix(Ad)
    keep = S.data != 0
    G = sparse.csr_matrix((np.ones(int(keep.sum())), (S.row[keep] // 2, S.col[keep] // 2)), shape=((n + 1) // 2, (n + 1) // 2))
    nc, lab = csgraph.connected_components(G, directed=False)
    d = np.abs(np.asarray(Ad.diagonal() if hasattr(Ad, 'diagonal') else np.diag(Ad)))
    groups = []
    for axis in (0, 1):
        t = np.zeros(n); t[axis::2] = 1.0
        y = np.abs(np.asarray(Ad @ t).ravel())
        for c in range(nc):
            v = np.flatnonzero(lab == c)
            dofs = np.concatenate((2 * v, 2 * v + 1)); dofs = dofs[dofs < n]
            if d[dofs].max() > 0 and y[dofs].max() <= tol * d[dofs].max():
                g = 2 * v + axis
                groups.append(g[g < n])
    return groups


def _solve_jacobi_krylov_limit(A, b):
    """what a Jacobi-preconditioned Krylov method started from zero converges to (the reference's restarted MINRES with
    M = diag(A)^-1, optimizer.py:1962-1971, and a Jacobi-PCG alike): the solution of A x = b that is M-orthogonal to the
    null space of A (M = the clipped diagonal of optimizer.py:1962-1966).  Two free meshes linked to each other have one
    (common rigid translations, one pair per link-connected set of mesh components): the solution is unique only up to it, and
    the Krylov iterates never leave M^-1 range(A).  The null vectors are taken from the structure (_floating_translations), the
    system is deflated by them exactly -- P A P x = P b with P the orthogonal projector off the translations --, solved densely
    (the pairs of the tests have ~2 k unknowns) and made M-orthogonal to them."""
    Ad = 0.5 * (A + A.T)
    groups = _floating_translations(Ad)
    Ad = Ad.toarray() if hasattr(Ad, 'toarray') else np.asarray(Ad)
    b = np.asarray(b, dtype=np.float64)
    if not groups:
        w, V = np.linalg.eigh(Ad)
        ok = w > 1e-11 * w.max()
        return V[:, ok] @ ((V[:, ok].T @ b) / w[ok])
    n = b.size
    Q = np.zeros((n, len(groups)))
    for k, g in enumerate(groups):
        Q[g, k] = 1.0 / np.sqrt(g.size)
    P = np.eye(n) - Q @ Q.T
    w, V = np.linalg.eigh(P @ Ad @ P)
    ok = w > 1e-11 * w.max()
    x = V[:, ok] @ ((V[:, ok].T @ (P @ b)) / w[ok])
    d = np.diag(Ad)
    Md = d.clip(min(1.0, d.max() / 1000), None) if d.max() > 0 else np.ones(n)
    for g in groups:
        x[g] -= np.sum(Md[g] * x[g]) / np.sum(Md[g])
    return x


def _optimize_linear(meshes, links, stiffness_lambda):
    """optimizer.py:1257-1437 to its fixed point (the Krylov limit above)"""
    A, b, _ = fem_ref.linear_system(meshes, links, stiffness_lambda, -1.0, GEAR_FIXED, GEAR_MOVING, GEAR_MOVING)
    dd = _solve_jacobi_krylov_limit(A, np.asarray(b, dtype=np.float64))
    fem_ref.apply_solution(meshes, dd, GEAR_MOVING, GEAR_MOVING)


def _link_from_coordinates(m0, m1, xy0, xy1, weight, gear):
    """Link.from_coordinates (optimizer.py:51-82): points located in both meshes at `gear`; pairs with a point outside are dropped"""
    def find(m, xy):
        v = m.vertices(gear)
        return _locate(v, m.triangles, np.asarray(xy) - m.offset(gear))
    tid0, tid1 = find(m0, xy0), find(m1, xy1)
    ok = (tid0 >= 0) & (tid1 >= 0)
    if not ok.any():
        return None
    B0 = m0.cart2bary(xy0[ok], gear, tid0[ok]); B1 = m1.cart2bary(xy1[ok], gear, tid1[ok])
    return fem_ref.RefLink(m0, m1, tid0[ok], tid1[ok], B0, B1, weight=np.asarray(weight)[ok])


def _batches(bboxes0, bboxes1, batch_size):
    """matcher.py:805-821: runs of one block size, cut into batches of about batch_size"""
    n = bboxes0.shape[0]
    sz0 = np.round(np.stack((bboxes0[:, 3] - bboxes0[:, 1], bboxes0[:, 2] - bboxes0[:, 0]), axis=-1))
    sz1 = np.round(np.stack((bboxes1[:, 3] - bboxes1[:, 1], bboxes1[:, 2] - bboxes1[:, 0]), axis=-1))
    chg = np.nonzero(np.any(np.diff(sz0, axis=0), axis=-1) | np.any(np.diff(sz1, axis=0), axis=-1))[0]
    edges = np.concatenate(([0], chg + 1, [n]), axis=None)
    if batch_size is None or batch_size >= n:
        return edges
    parts = []
    for a, b in zip(edges[:-1], edges[1:]):
        nb = max(1, int(np.ceil((b - a) / batch_size)))
        parts.append(np.linspace(a, b, num=nb + 1, endpoint=True))
    return np.unique(np.round(np.concatenate(parts, axis=-1)).astype(np.int32))


def section_match(m0, m1, img0, img1, spacings=(100,), sigma=2.5, batch_size=100, conf_thresh=0.3, residue_mode='huber', residue_len=0,
                  conf_mode=ncc_ref.FFT_CONF_MIRROR, min_boundary_distance=0, shrink_factor=1, refine_mode=2, stiffness_lambda=0.5,
                  compute_strain=False, materials=None, res=1.0, anchor_rounds=None, trace=None, distributor='cartesian_region',
                  min_num_blocks=2, block_matcher=None):
    """matcher.py:370-427 (no initial matches: straight into the loop) + 430-778 for a pair of fem_ref.RefMesh with linear
    materials (soft / unrendered triangles already removed by the caller) over two images whose pixel (0, 0) sits at the
    origin.  Every relaxation is solved to its fixed point.  anchor_rounds: per round the mesh-0 blocks of another
    implementation (lattice phase only, module docstring).  Returns (xy0, xy1, weight, strain).
    distributor 'cartesian_bbox': the blocks of matcher.py:865-891 instead (the loop of stitching_matcher).  block_matcher: a
    callable (round, m0, m1, bboxes0, bboxes1, pad, subpixel, tol) -> (xy0, xy1, conf) in place of render + DoG + NCC -- golden
    G23 drives the reference's own loop and this one with the same script, which pins everything BETWEEN the block matches
    (blocks on the moving bounds, links, relaxation, residue weights, the walk, final matches, strain) to the reference."""
    invalid = (None, None, 0, DEFAULT_AVG_DEFORM)
    spacings = np.sort(np.asarray(spacings, dtype=np.float64))[::-1]
    if compute_strain:
        ori = [fem_ref.RefMesh(m.vertices(GEAR_INITIAL), m.triangles, uid=m.uid, locked=m.locked, soft_factor=m.soft_factor) for m in (m0, m1)]
    for m in (m0, m1):
        m.anneal_copy(gear=(GEAR_MOVING, GEAR_FIXED))                       # matcher.py:565-566
    sp, sp_indx, initialized, pad, rnd = float(spacings[0]), 0, False, True, 0
    link = None
    while sp_indx < spacings.size:
        last = sp == spacings[-1]
        rfm = refine_mode if (last or refine_mode != 2) else 0             # matcher.py:572-590
        tol = 0.1 if last else max(1, 0.02 * sp)
        vm = [m.vertices_w_offset(GEAR_MOVING) for m in (m0, m1)]
        if distributor == 'cartesian_bbox':                                # matcher.py:865-891 on the bounds of the MOVING gears
            bounds = [np.concatenate((v.min(axis=0), v.max(axis=0))) for v in vm]
            bb0, bb1 = ncc_ref.distributor_cartesian_bbox(bounds[0], bounds[1], sp, min_num_blocks=min_num_blocks if last else 1,
                                                          shrink_factor=shrink_factor, zorder=True)
            if bb0 is None:
                return invalid
        else:
            bb0, bb1 = distribute_matching_blocks(vm[0], m0.triangles, vm[1], m1.triangles, sp, refine_mode=rfm, shrink_factor=shrink_factor,
                                                  min_boundary_distance=min_boundary_distance, zorder=True, materials=materials, res=res,
                                                  anchor_blocks=None if anchor_rounds is None or rnd >= len(anchor_rounds) else anchor_rounds[rnd])
        if bb0.shape[0] == 0:
            if not initialized:
                return invalid
            break
        if block_matcher is not None:
            xy0, xy1, conf = block_matcher(rnd, m0, m1, bb0, bb1, pad, bool(last), tol)
        else:
            edges = _batches(bb0, bb1, batch_size)
            parts = [pipeline_ref.bboxes_mesh_renderer_matcher(m0, m1, img0, img1, bb0[a:b], bb1[a:b], sigma=sigma, conf_mode=conf_mode, pad=pad,
                                                                subpixel=bool(last), affine_approx_tol=tol) for a, b in zip(edges[:-1], edges[1:])]
            xy0 = np.concatenate([p[0] for p in parts]); xy1 = np.concatenate([p[1] for p in parts]); conf = np.concatenate([p[2] for p in parts])
        if trace is not None:
            trace.append(dict(sp=sp, bboxes0=bb0, bboxes1=bb1, conf=conf, pad=pad))
        rnd += 1
        if np.all(conf <= conf_thresh):
            if not initialized:
                return invalid
            break
        keep = conf > conf_thresh
        xy0, xy1, wt = xy0[keep], xy1[keep], conf[keep]
        max_dis = np.max(np.sum((xy0 - xy1) ** 2, axis=-1)) ** 0.5
        next_pos = np.searchsorted(-spacings, -4 * max_dis) - 1            # matcher.py:689-716 (allow_enlarge False, allow_dwell 0, no skips)
        if next_pos > sp_indx:
            next_pos = min(next_pos, sp_indx + 1)
            pad = next_pos > sp_indx + 1
            sp_indx = next_pos
        else:
            pad = True
            sp_indx += 1
        new_link = _link_from_coordinates(m0, m1, xy0, xy1, wt, GEAR_MOVING)     # (link_weight_decay 0: the older links are gone)
        if new_link is None:
            if not initialized:
                return invalid
            break
        link = new_link
        if max_dis > 0.1:
            _optimize_linear([m0, m1], [link], stiffness_lambda)
            if residue_len > 0:
                cutoff = 1 - 1 / (MAXIMUM_DEFORM_ALLOWED + 1)
                for m in (m0, m1):                                         # relax_higly_deformed (optimizer.py:763-772)
                    if not m.locked:
                        fem_ref.relax_mesh_most_deformed(m, gear=(GEAR_FIXED, GEAR_MOVING), deform_cutoff=cutoff)
                rw = link.residue_weights((GEAR_MOVING, GEAR_MOVING), residue_mode, residue_len)
                if np.any(rw != link.residue_weight):
                    link.residue_weight = rw
                    if sp_indx < spacings.size:
                        _optimize_linear([m0, m1], [link], stiffness_lambda)
            if trace is not None:
                trace[-1]['field1'] = m1.vertices_w_offset(GEAR_MOVING) - m1.vertices_w_offset(GEAR_INITIAL)
        initialized = True
        if 0 <= sp_indx < spacings.size:
            sp = float(spacings[sp_indx])
    if link is None:
        return invalid
    w = link.total_weight()
    use = w > 0                                                            # Link.mask (optimizer.py:399-402)
    xy0 = m0.bary2cart(link.tid0[use], link.B0[use], GEAR_INITIAL)
    xy1 = m1.bary2cart(link.tid1[use], link.B1[use], GEAR_INITIAL)
    weight = w[use]
    strain = DEFAULT_AVG_DEFORM
    if compute_strain:                                                     # matcher.py:752-777
        o0, o1 = ori
        lk = _link_from_coordinates(o0, o1, xy0, xy1, weight, GEAR_INITIAL)
        one_locked = o0.locked or o1.locked
        # optimize_affine_cascade(INITIAL -> FIXED, svd_clip (1, 1)): the free mesh(es) brought onto the other rigidly
        p0 = o0.bary2cart(lk.tid0, lk.B0, GEAR_INITIAL); p1 = o1.bary2cart(lk.tid1, lk.B1, GEAR_INITIAL)
        if o0.locked or not o1.locked:
            _, R = fem_ref.fit_affine(p0, p1, return_rigid=True, weight=lk.total_weight(), svd_clip=(1, 1), avoid_flip=True)
            o1.set_affine(R, gear=(GEAR_INITIAL, GEAR_FIXED))
        else:
            _, R = fem_ref.fit_affine(p1, p0, return_rigid=True, weight=lk.total_weight(), svd_clip=(1, 1), avoid_flip=True)
            o0.set_affine(R, gear=(GEAR_INITIAL, GEAR_FIXED))
        for m in (o0, o1):
            m.anneal_copy(gear=(GEAR_FIXED, GEAR_MOVING))
        _optimize_linear([o0, o1], [lk], stiffness_lambda)
        soft_avg = np.mean([m.soft_factor for m in (o0, o1)])
        Es = Es0 = 0.0
        for m in (o0, o1):
            if (one_locked and not m.locked) or ((not one_locked) and m.soft_factor <= soft_avg):
                v0 = m.vertices(GEAR_FIXED)
                dv = m.vertices(GEAR_MOVING) - v0
                v0 = v0 - np.mean(v0, axis=0, keepdims=True); dv = dv - np.mean(dv, axis=0, keepdims=True)
                St, _ = m.stiffness_matrix()
                Es += max(0, St.dot(dv.ravel()).dot(dv.ravel())); Es0 += max(0, St.dot(v0.ravel()).dot(v0.ravel()))
        strain = (Es / Es0) ** 0.5
    return xy0, xy1, weight, strain
