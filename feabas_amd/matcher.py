"""Host mirror of feabas.matcher for the NCC path.

``xcorr_fft`` runs on the GPU (fb_ncc_batch); ``global_translation_matcher``
and the block distributor keep the reference's host-side control flow and call
the GPU kernels for the arithmetic.
"""
import ctypes as C

import os
import sys
import time

import numpy as np

from . import _lib, common
from . import constant as const


def next_fast_len(n):
    """scipy.fftpack.next_fast_len (feabas/matcher.py:6,59-62)."""
    return int(_lib.load().fb_next_fast_len(int(n)))


def xcorr_fft(img0, img1, conf_mode=const.FFT_CONF_MIRROR, **kwargs):
    """feabas/matcher.py:22-135.

    img0: N x H0 x W0 (x C), img1: N x H1 x W1 (x C).  Returns dx, dy (float64)
    and conf so that centre(img1) + (dx, dy) corresponds to centre(img0).
    kwargs: sigma, mask0, mask1 (DoG pre-filter, matcher.py:54-56), subpixel,
    pad, normalize (matcher.py:70-81, 119-122: the surfaces divided by the overlap
    of the masks at every lag -- no reference call site enables it; the rocFFT
    class of the library, ``fb_ncc_batch_normalized``).
    """
    sigma = kwargs.get('sigma', 0)
    mask0 = kwargs.get('mask0', None)
    mask1 = kwargs.get('mask1', None)
    normalize = bool(kwargs.get('normalize', False))
    subpixel = kwargs.get('subpixel', False)
    pad = kwargs.get('pad', True)
    img0 = np.asarray(img0)
    img1 = np.asarray(img1)
    if img0.ndim > 3:
        img0 = np.moveaxis(img0, -1, 1)
    if img1.ndim > 3:
        img1 = np.moveaxis(img1, -1, 1)
    if sigma > 0:
        img0 = common.masked_dog_filter(img0, sigma, mask=mask0)
        img1 = common.masked_dog_filter(img1, sigma, mask=mask1)
    a = np.ascontiguousarray(img0, dtype=np.float32)
    b = np.ascontiguousarray(img1, dtype=np.float32)
    n = a.shape[0]
    ch = a.shape[1] if a.ndim > 3 else 1
    if b.shape[0] != n or (b.shape[1] if b.ndim > 3 else 1) != ch:
        raise ValueError('img0 and img1 must agree in batch and channel counts')
    h0, w0 = a.shape[-2:]
    h1, w1 = b.shape[-2:]
    dx = np.empty(n, dtype=np.float64)
    dy = np.empty(n, dtype=np.float64)
    conf = np.empty(n, dtype=np.float32)
    if n > 0 and normalize:
        mks = []
        for mk, shp, name in ((mask0, (h0, w0), 'mask0'), (mask1, (h1, w1), 'mask1')):
            if mk is not None:
                mk = np.ascontiguousarray(mk, dtype=np.float32)
                if mk.shape != shp:
                    raise ValueError(f'xcorr_fft(normalize=True): {name} must have the shape of one image, {shp}')
            mks.append(mk)
        _lib.check(_lib.load().fb_ncc_batch_normalized(_lib.ctx(), _lib.ptr(a), _lib.ptr(b), n, ch, h0, w0, h1, w1,
                                                       None if mks[0] is None else _lib.ptr(mks[0]), None if mks[1] is None else _lib.ptr(mks[1]),
                                                       1 if pad else 0, 1 if subpixel else 0, int(conf_mode),
                                                       _lib.ptr(dx), _lib.ptr(dy), _lib.ptr(conf)))
    elif n > 0:
        _lib.check(_lib.load().fb_ncc_batch(_lib.ctx(), _lib.ptr(a), _lib.ptr(b), n, ch, h0, w0, h1, w1,
                                            1 if pad else 0, 1 if subpixel else 0, int(conf_mode),
                                            _lib.ptr(dx), _lib.ptr(dy), _lib.ptr(conf)))
    return dx, dy, conf


def _best_divide(shape, divide_factor):
    """grid (rows, cols) of ~divide_factor blocks with the most moderate aspect
    ratio (feabas/matcher.py:162-177)."""
    if hasattr(divide_factor, '__len__'):
        return tuple(divide_factor[:2])
    aspect = shape[0] / shape[1]
    best = np.inf
    grid = None
    for f in range(1, int(divide_factor ** 0.5) + 1):
        if divide_factor % f != 0:
            continue
        q = f ** 2 / divide_factor
        for cand, val in (((int(divide_factor / f), int(f)), abs(np.log(aspect * q))),
                          ((int(f), int(divide_factor / f)), abs(np.log(aspect / q)))):
            if val < best:
                best, grid = val, cand
    return grid


def _fit_span(lo, hi, full, limit):
    """grow [lo, hi) to `full` pixels and slide it inside [0, limit) (matcher.py:189-194)."""
    grow = int(np.ceil((full - (hi - lo)) / 2))
    span = np.array((lo - grow, hi + grow))
    return (span - min(span[0], 0) - max(span[1] - limit, 0)).clip(0, limit)


def global_translation_matcher(img0, img1, **kwargs):
    """feabas/matcher.py:138-221: whole-image translation, with a second shot on
    ~divide_factor sub-blocks when the confidence is low."""
    sigma = kwargs.get('sigma', 0.0)
    conf_mode = kwargs.get('conf_mode', const.FFT_CONF_MIRROR)
    conf_thresh = kwargs.get('conf_thresh', 0.3)
    divide_factor = kwargs.get('divide_factor', 6)
    if sigma > 0:
        img0 = common.masked_dog_filter(img0, sigma, mask=kwargs.get('mask0', None))
        img1 = common.masked_dog_filter(img1, sigma, mask=kwargs.get('mask1', None))
    ht0, wd0 = img0.shape[-2:]
    ht1, wd1 = img1.shape[-2:]
    tx, ty, conf = xcorr_fft(img0[None], img1[None], conf_mode=conf_mode, pad=True)
    tx, ty, conf = tx.item(), ty.item(), conf.item()
    tx += (wd1 - wd0) / 2
    ty += (ht1 - ht0) / 2
    if conf > conf_thresh:
        return tx, ty, conf
    grid = _best_divide(np.minimum((ht0, wd0), (ht1, wd1)), divide_factor)
    xa0, ya0, xb0, yb0 = common.divide_bbox((0, 0, wd0, ht0), min_num_blocks=grid)
    xa1, ya1, xb1, yb1 = common.divide_bbox((0, 0, wd1, ht1), min_num_blocks=grid)
    blocks0, blocks1, off_x, off_y = [], [], [], []
    for k in range(xa0.size):
        bw = max(xb0[k] - xa0[k], xb1[k] - xa1[k])
        bh = max(yb0[k] - ya0[k], yb1[k] - ya1[k])
        ys0 = _fit_span(ya0[k], yb0[k], bh, ht0)
        xs0 = _fit_span(xa0[k], xb0[k], bw, wd0)
        blk0 = img0[ys0[0]:ys0[1], xs0[0]:xs0[1]]
        if np.ptp(blk0) == 0:
            continue
        ys1 = _fit_span(ya1[k], yb1[k], bh, ht1)
        xs1 = _fit_span(xa1[k], xb1[k], bw, wd1)
        blk1 = img1[ys1[0]:ys1[1], xs1[0]:xs1[1]]
        if np.ptp(blk1) == 0:
            continue
        blocks0.append(blk0)
        blocks1.append(blk1)
        off_x.append((np.ptp(xs1) - np.ptp(xs0)) / 2 + xs1[0] - xs0[0])
        off_y.append((np.ptp(ys1) - np.ptp(ys0)) / 2 + ys1[0] - ys0[0])
    if not blocks0:
        return tx, ty, conf
    btx, bty, bconf = xcorr_fft(np.stack(blocks0, axis=0), np.stack(blocks1, axis=0), conf_mode=conf_mode, pad=True)
    btx = btx + np.array(off_x)
    bty = bty + np.array(off_y)
    k_best = int(np.argmax(bconf))
    if bconf[k_best] >= conf:
        tx, ty, conf = btx[k_best], bty[k_best], bconf[k_best]
    return tx, ty, conf


def distributor_cartesian_bbox(mesh0, mesh1, spacing, **kwargs):
    """feabas/matcher.py:865-891: z-ordered block grid over the intersection of
    the two mesh bounding boxes.  mesh0/mesh1 need a ``bbox(gear=)`` method."""
    gear = kwargs.get('gear', const.MESH_GEAR_MOVING)
    min_num_blocks = kwargs.get('min_num_blocks', 1)
    shrink_factor = kwargs.get('shrink_factor', 1)
    zorder = kwargs.get('zorder', False)
    if not hasattr(shrink_factor, '__len__'):
        shrink_factor = (shrink_factor, shrink_factor)
    bbox, valid = common.intersect_bbox(mesh0.bbox(gear=gear), mesh1.bbox(gear=gear))
    if not valid:
        return None, None
    bb0 = np.stack(common.divide_bbox(bbox, block_size=spacing, min_num_blocks=min_num_blocks,
                                      shrink_factor=shrink_factor[0]), axis=-1)
    bb1 = np.stack(common.divide_bbox(bbox, block_size=spacing, min_num_blocks=min_num_blocks,
                                      shrink_factor=shrink_factor[1]), axis=-1)
    if zorder:
        col = np.round((bb0[:, 0] - bb0[:, 0].min()) / spacing)
        row = np.round((bb0[:, 1] - bb0[:, 1].min()) / spacing)
        order = common.z_order(np.stack((col, row), axis=-1))
        bb0, bb1 = bb0[order], bb1[order]
    return bb0, bb1


class _RegionPair:
    """The region both meshes cover (shapely: region0.intersection(region1), matcher.py:944-946) as a point predicate on
    triangles: a point lies in it when it lies in a triangle of each mesh, and its distance to the region's outline is the
    smaller of its distances to the outlines (boundary edges) of the two meshes -- so ``buffer(-d)`` (matcher.py:987) is
    ``inside & (distance >= d)``.  Areas and connected parts are taken on a raster of the predicate (step `res`): shapely is
    not in the image, and nothing downstream needs the polygons themselves."""

    def __init__(self, mesh0, mesh1, gear, only=None, exclude=None):
        """only / exclude: lists of (mesh index, boolean triangle mask) -- the part of the common region that lies in ANY of the
        `only` triangle sets and in NONE of the `exclude` ones (the material regions of a refinement level and what the finer
        levels covered already, matcher.py:963-995); None = no restriction"""
        self.meshes = (mesh0, mesh1)
        self.gear = gear
        self.only, self.exclude = only, exclude
        self._rasters = {}
        self._located = {}
        self.segs = []
        for m in (mesh0, mesh1):
            v = m.vertices_w_offset(gear)
            self.segs.append(v[m.boundary_edges()])                          # [E, 2, 2]
        self.segs = np.concatenate(self.segs, axis=0)
        bb, self.valid = common.intersect_bbox(mesh0.bbox(gear=gear), mesh1.bbox(gear=gear))
        self.bbox = np.asarray(bb, dtype=np.float64)

    def _locate(self, pts):
        """triangle of every point in each of the two meshes (-1 outside; points outside mesh 0 are not looked up in mesh 1)"""
        ok = np.ones(pts.shape[0], dtype=bool)
        tids = []
        for m in self.meshes:
            if ok.all():                                   # (no gather / scatter of the point list while nothing has dropped out)
                tid = np.asarray(m.tri_finder(pts, gear=self.gear))
                ok = tid >= 0
            else:
                idx = np.flatnonzero(ok)
                tid = np.full(pts.shape[0], -1, dtype=np.int32)
                if idx.size:
                    tid[idx] = m.tri_finder(pts[idx], gear=self.gear)
                    ok[idx] = tid[idx] >= 0
            tids.append(tid)
        return tids

    def inside(self, pts, tids=None):
        pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
        if tids is None:
            tids = self._locate(pts)
        ok = (tids[0] >= 0) & (tids[1] >= 0)
        for sets, want in ((self.only, True), (self.exclude, False)):
            if sets is None or not ok.any():
                continue
            hit = np.zeros(pts.shape[0], dtype=bool)
            idx = np.flatnonzero(ok)
            for k, tri_mask in sets:
                hit[idx] |= np.asarray(tri_mask, dtype=bool)[tids[k][idx]]
            ok &= hit if want else ~hit
        return ok

    def boundary_distance(self, pts, cap=np.inf):
        """distance of every point to the nearest outline segment (values above `cap` need not be exact)"""
        pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
        out = np.full(pts.shape[0], np.inf)
        a, b = self.segs[:, 0], self.segs[:, 1]
        ab = b - a
        l2 = np.maximum(np.sum(ab * ab, axis=1), 1e-300)
        lo, hi = np.minimum(a, b) - cap, np.maximum(a, b) + cap
        step = max(1, int(4e6 // max(1, a.shape[0])))
        for s0 in range(0, pts.shape[0], step):
            p = pts[s0:s0 + step]
            near = np.all((p[:, None, :] >= lo[None]) & (p[:, None, :] <= hi[None]), axis=2) if np.isfinite(cap) else None
            t = np.clip(np.einsum('pej,ej->pe', p[:, None, :] - a[None], ab) / l2[None], 0.0, 1.0)
            d2 = np.sum((p[:, None, :] - (a[None] + t[..., None] * ab[None])) ** 2, axis=2)
            if near is not None:
                d2 = np.where(near, d2, np.inf)
            out[s0:s0 + step] = np.sqrt(d2.min(axis=1)) if d2.shape[1] else np.inf
        return out

    def near_boundary(self, pts, d):
        """which points lie closer than `d` to an outline segment.  Exact, and without the points x segments table of
        ``boundary_distance`` (12 s for the raster of an 8192^2 section pair): a k-d tree of the points hands every segment the
        points inside the circle around its middle that holds its d-neighbourhood; only those pairs are measured."""
        from scipy.spatial import cKDTree
        pts = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
        out = np.zeros(pts.shape[0], dtype=bool)
        if pts.shape[0] == 0 or self.segs.shape[0] == 0:
            return out
        a, b = self.segs[:, 0], self.segs[:, 1]
        ab = b - a
        l2 = np.sum(ab * ab, axis=1)
        hits = cKDTree(pts).query_ball_point(0.5 * (a + b), 0.5 * np.sqrt(l2) + d)
        cnt = np.fromiter((len(h) for h in hits), dtype=np.int64, count=len(hits))
        if not cnt.any():
            return out
        ip = np.concatenate([np.asarray(h, dtype=np.int64) for h in hits if len(h)])
        iseg = np.repeat(np.arange(a.shape[0]), cnt)
        p = pts[ip]
        t = np.clip(np.einsum('ij,ij->i', p - a[iseg], ab[iseg]) / np.maximum(l2[iseg], 1e-300), 0.0, 1.0)
        d2 = np.sum((p - (a[iseg] + t[:, None] * ab[iseg])) ** 2, axis=1)
        out[ip[d2 < d * d]] = True
        return out

    def near_boundary_raster(self, xs, ys, d):
        """``near_boundary`` for the points of a raster (xs x ys, evenly spaced) as a [len(ys), len(xs)] mask: every outline segment
        is measured against the raster cells of its own bounding box grown by d and against nothing else -- no tree, no point list
        (the raster of an 8192^2 section pair at step 17.5 has 219 k points, 330 outline segments and 17 k such cells)"""
        out = np.zeros((ys.size, xs.size), dtype=bool)
        if out.size == 0 or self.segs.shape[0] == 0:
            return out
        a, b = self.segs[:, 0], self.segs[:, 1]
        ab = b - a
        l2 = np.maximum(np.sum(ab * ab, axis=1), 1e-300)
        lo, hi = np.minimum(a, b) - d, np.maximum(a, b) + d
        rx = (xs[-1] - xs[0]) / (xs.size - 1) if xs.size > 1 else 1.0
        ry = (ys[-1] - ys[0]) / (ys.size - 1) if ys.size > 1 else 1.0
        # one cell of slack on both sides: the candidate box only has to CONTAIN the d-neighbourhood, the distance test is exact
        ix0 = np.clip(np.floor((lo[:, 0] - xs[0]) / rx).astype(np.int64) - 1, 0, xs.size); ix1 = np.clip(np.ceil((hi[:, 0] - xs[0]) / rx).astype(np.int64) + 2, 0, xs.size)
        iy0 = np.clip(np.floor((lo[:, 1] - ys[0]) / ry).astype(np.int64) - 1, 0, ys.size); iy1 = np.clip(np.ceil((hi[:, 1] - ys[0]) / ry).astype(np.int64) + 2, 0, ys.size)
        wx, wy = np.maximum(ix1 - ix0, 0), np.maximum(iy1 - iy0, 0)
        cnt = wx * wy
        total = int(cnt.sum())
        if total == 0:
            return out
        iseg = np.repeat(np.arange(a.shape[0]), cnt)
        local = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        ix = ix0[iseg] + local % wx[iseg]
        iy = iy0[iseg] + local // wx[iseg]
        p = np.stack((xs[ix], ys[iy]), axis=-1)
        t = np.clip(np.einsum('ij,ij->i', p - a[iseg], ab[iseg]) / l2[iseg], 0.0, 1.0)
        d2 = np.sum((p - (a[iseg] + t[:, None] * ab[iseg])) ** 2, axis=1)
        hit = d2 < d * d
        out[iy[hit], ix[hit]] = True
        return out

    def select(self, pts, erode=0.0, tids=None):
        ok = self.inside(pts, tids)
        if erode > 0 and ok.any():
            idx = np.flatnonzero(ok)
            ok[idx] = ~self.near_boundary(np.asarray(pts).reshape(-1, 2)[idx], erode)
        return ok

    def restricted(self, only=None, exclude=None):
        """the same pair of meshes (outlines shared) with triangle-set restrictions"""
        r = object.__new__(_RegionPair)
        r.meshes, r.gear, r.segs, r.bbox, r.valid = self.meshes, self.gear, self.segs, self.bbox, self.valid
        r.only, r.exclude, r._rasters = only, exclude, {}
        r._located = self._located                              # the point location of a raster does not depend on the restriction
        return r

    def raster(self, res, erode=0.0):
        key = (float(res), float(erode))
        if key not in self._rasters:
            x0, y0, x1, y1 = self.bbox
            # (a pair of meshes whose coordinates ran away -- a relaxation that went wrong upstream -- must fail here with a
            # message, not take the host down with a raster of 1e10 cells)
            cells = ((x1 - x0) / res) * ((y1 - y0) / res)
            if not np.isfinite(cells) or cells > 1e8:
                raise ValueError(f'distribute_matching_blocks: the common region spans {x1 - x0:.3g} x {y1 - y0:.3g} px at a raster step of {res:.3g}: '
                                 'the meshes are not where images could be (diverged relaxation?)')
            xs = np.arange(x0 + 0.5 * res, x1, res); ys = np.arange(y0 + 0.5 * res, y1, res)
            xx, yy = np.meshgrid(xs, ys)
            pts = np.stack((xx.ravel(), yy.ravel()), axis=-1)
            if float(res) not in self._located:               # located once per raster step, whatever is cut out of the region afterwards
                self._located[float(res)] = self._locate(pts)
            ok = self.inside(pts, self._located[float(res)]).reshape(yy.shape)
            if erode > 0 and ok.any():
                ok &= ~self.near_boundary_raster(xs, ys, erode)
            self._rasters[key] = (xs, ys, ok)
        return self._rasters[key]


def _region2grid_cartesian(region, spacing, erode=0.0, res=None, **kwargs):
    """matcher.py:1019-1043: a lattice of step `spacing` per connected part of the region, anchored at the part's
    representative point, kept where it falls inside the part.  `region` is a _RegionPair; the parts, their bounds and their
    representative points (GEOS takes the middle of the widest stretch of the scan line through the middle of the bounds)
    come from a raster of the region with step `res`; membership of the lattice points is tested exactly."""
    from scipy import ndimage
    res = max(spacing / 4.0, 1.0) if res is None else res
    xs, ys, msk = region.raster(res, erode)
    if not msk.any():
        return None
    lab, nlab = ndimage.label(msk, structure=np.ones((3, 3), dtype=bool))
    # every raster cell -> the label of the nearest cell inside the region: a lattice point in a spur of the region that is
    # thinner than the raster (no cell centre falls into it) belongs to the part the spur hangs on
    near = lab if msk.all() else lab[tuple(ndimage.distance_transform_edt(lab == 0, return_distances=False, return_indices=True))]
    territory = ndimage.find_objects(near)
    cntrs = []
    for k in range(1, nlab + 1):
        rr, cc = np.nonzero(lab == k)
        # The lattice PHASE comes from the representative point alone; the bounds of the part (matcher.py:1027) only say how
        # far the lattice reaches.  A spur of the part that the raster does not see can reach far beyond the cells it does
        # see, so the lattice is laid over the part's TERRITORY -- the bounds of all raster cells that are nearer to this part
        # than to any other -- and every point is kept by the two exact tests: it lies in the region, and the part nearest to
        # it is this one.
        ty, tx = territory[k - 1]
        bx0, by0, bx1, by1 = (float(b) for b in region.bbox)               # (the last raster cell may end short of the bounds)
        rx_mn = bx0 if tx.start == 0 else xs[tx.start] - 0.5 * res
        rx_mx = bx1 if tx.stop == xs.size else xs[tx.stop - 1] + 0.5 * res
        ry_mn = by0 if ty.start == 0 else ys[ty.start] - 0.5 * res
        ry_mx = by1 if ty.stop == ys.size else ys[ty.stop - 1] + 0.5 * res
        row = min(max(int(round((0.5 * (ys[rr.min()] + ys[rr.max()]) - ys[0]) / res)), rr.min()), rr.max())
        run = np.flatnonzero(lab[row] == k)
        if run.size == 0:                                          # the middle row misses the part (a ring, a C): take its fullest row
            row = rr[np.argmax(np.bincount(rr)[rr])]
            run = np.flatnonzero(lab[row] == k)
        brk = np.flatnonzero(np.diff(run) > 1)
        starts = np.concatenate(([0], brk + 1)); ends = np.concatenate((brk, [run.size - 1]))
        w = int(np.argmax(ends - starts))
        rx = 0.5 * (xs[run[starts[w]]] + xs[run[ends[w]]]); ry = ys[row]
        gx0 = rx - ((rx - rx_mn) // spacing) * spacing
        gy0 = ry - ((ry - ry_mn) // spacing) * spacing
        gxx, gyy = np.meshgrid(np.arange(gx0, rx_mx, spacing), np.arange(gy0, ry_mx, spacing))
        rv = np.stack((gxx.ravel(), gyy.ravel()), axis=-1)
        if rv.shape[0] == 0:
            continue
        ci = np.clip(np.round((rv[:, 0] - xs[0]) / res).astype(int), 0, xs.size - 1)
        ri = np.clip(np.round((rv[:, 1] - ys[0]) / res).astype(int), 0, ys.size - 1)
        rv = rv[near[ri, ci] == k]
        if rv.shape[0]:
            cntrs.append(rv[region.select(rv, erode)])
    cntrs = [c for c in cntrs if c.shape[0]]
    if not cntrs:
        return None
    # unary_union of the MultiPoints: duplicates merged, rows sorted by (x, y) -- np.unique(axis=0) without its structured-view sort
    pts = np.concatenate(cntrs, axis=0)
    pts = pts[np.lexsort((pts[:, 1], pts[:, 0]))]
    if pts.shape[0] > 1:
        pts = pts[np.concatenate(([True], np.any(pts[1:] != pts[:-1], axis=1)))]
    return pts


def distribute_matching_blocks(mesh0, mesh1, spacing, dfunc='cartesian_region', **kwargs):
    """feabas/matcher.py:894-1016: blocks on a lattice over the region BOTH meshes cover, at least `min_boundary_distance`
    away from its outline (relaxed until half of the region survives), z-ordered.  The reference builds the region with
    shapely polygons; here it is a point predicate on the triangles of the two meshes (_RegionPair) -- same lattice rule,
    same block sizes; the lattice anchor of a connected part is its representative point as GEOS defines it, located on a
    raster of the region (step spacing / 4), so anchors agree with shapely's up to that step.
    ``refine_mode`` (0 ignore, 1 refinement regions only, 2 both -- the default): the triangles of a material whose name holds
    'refine' or whose ``area_constraint`` lies in (0, 1) (Mesh.material_ids / material_names / material_area_constraints) get
    a lattice of step spacing x area_constraint and blocks of spacing x area_constraint^refine_box_exp, finest level first; a
    coarser level leaves out what a finer one covered."""
    refine_mode = kwargs.get('refine_mode', 2)
    refine_box_exp = kwargs.get('refine_box_exp', 0.5)
    gear = kwargs.get('gear', const.MESH_GEAR_MOVING)
    shrink_factor = kwargs.get('shrink_factor', 1)
    min_box_side = kwargs.get('min_box_side', 5)
    max_box_side = kwargs.get('max_box_side', np.inf)
    min_boundary_distance = kwargs.get('min_boundary_distance', 0)
    zorder = kwargs.get('zorder', True)
    render_weight_threshold = kwargs.get('render_weight_threshold', 0)
    if isinstance(dfunc, str):
        if dfunc.lower() == 'cartesian_region':
            dfunc = _region2grid_cartesian
        elif dfunc.lower() == 'intersect_triangulation':
            raise NotImplementedError("distributor 'intersect_triangulation' meshes the region with `triangle` (matcher.py:1046-1058), which is not in the image")
        else:
            raise ValueError(f'unsupported distributor type {dfunc}')
    if isinstance(refine_mode, str):
        refine_mode = 0 if refine_mode.lower() == 'none' else (1 if 'only' in refine_mode.lower() else 2)
    empty = (np.empty((0, 4)), np.empty((0, 4)))
    if render_weight_threshold > 0:
        mesh0 = mesh0.submesh(mesh0.triangle_mask_for_render(render_weight_threshold=render_weight_threshold))
        mesh1 = mesh1.submesh(mesh1.triangle_mask_for_render(render_weight_threshold=render_weight_threshold))
    whole = _RegionPair(mesh0, mesh1, gear)
    if not whole.valid:
        return empty
    res0 = max(spacing / 4.0, 1.0)
    if float(whole.raster(res0)[2].sum()) == 0:
        return empty
    if not hasattr(shrink_factor, '__len__'):
        shrink_factor = (shrink_factor, shrink_factor)
    else:
        a0 = np.sum(mesh0.triangle_areas(gear=gear)) / mesh0.num_triangles
        a1 = np.sum(mesh1.triangle_areas(gear=gear)) / mesh1.num_triangles
        shrink_factor = (max(shrink_factor), min(shrink_factor)) if a0 > a1 else (min(shrink_factor), max(shrink_factor))
    # levels: area factor -> triangle sets (None = the whole common region)
    levels = {}
    if refine_mode != 1:
        levels[1.0] = None
    if refine_mode != 0:
        for k, m in enumerate((mesh0, mesh1)):
            for name, uid in getattr(m, 'material_names', {}).items():
                factor = float(getattr(m, 'material_area_constraints', {}).get(name, 1.0))
                if ('refine' not in name) and (factor == 0 or factor >= 1):
                    continue
                tri_mask = m.material_ids == uid
                if tri_mask.any() and levels.get(factor, ()) is not None:
                    levels.setdefault(factor, []).append((k, tri_mask))
    out0, out1 = [], []
    covered = []                                   # triangle sets of the finer levels
    everything = False
    for factor in sorted(levels):
        if everything:
            break
        spc = spacing * factor
        box_scale = factor ** (refine_box_exp - 1)
        res = max(spc / 4.0, 1.0)
        sets = levels[factor]
        level = whole if sets is None else whole.restricted(only=sets)
        area_r = float(level.raster(res)[2].sum()) * res * res
        region = level if not covered else whole.restricted(only=sets, exclude=covered)
        if sets is None:
            everything = True
        else:
            covered = covered + sets
        if area_r == 0:
            continue
        erode = 0.0
        if min_boundary_distance > 0:
            bound_coeff = 1.0
            for _ in range(64):
                erode = min_boundary_distance * box_scale * bound_coeff
                area_c = float(region.raster(res, erode)[2].sum()) * res * res
                if area_c >= 0.5 * area_r:
                    break
                bound_coeff *= 0.3 / (1 - area_c / area_r)
                if bound_coeff < 1e-3:
                    erode = 0.0
                    break
        cntrs = dfunc(region, spc, erode=erode, res=res)
        if cntrs is None:
            continue
        sides = (spc * box_scale * np.array(shrink_factor, dtype=np.float64)).clip(min_box_side, max_box_side)
        h0, h1 = np.ceil(sides[0] / 2), np.ceil(sides[1] / 2)
        b0 = np.concatenate((cntrs - h0, cntrs + h0), axis=-1)
        b1 = np.concatenate((cntrs - h1, cntrs + h1), axis=-1)
        if zorder:
            x_rnd = np.round((cntrs[:, 0] - cntrs[:, 0].min()) / spc)
            y_rnd = np.round((cntrs[:, 1] - cntrs[:, 1].min()) / spc)
            idx = common.z_order(np.stack((x_rnd, y_rnd), axis=-1))
            b0, b1 = b0[idx], b1[idx]
        out0.append(b0); out1.append(b1)
    if not out0:
        return empty
    return np.concatenate(out0, axis=0), np.concatenate(out1, axis=0)


def block_displacements_to_points(bboxes0, bboxes1, dx, dy):
    """feabas/matcher.py:840-849: block displacement -> a pair of matched points."""
    ctr0 = common.bbox_centers(bboxes0)
    ctr1 = common.bbox_centers(bboxes1)
    sz0 = common.bbox_sizes(bboxes0).astype(np.float64)
    sz1 = common.bbox_sizes(bboxes1).astype(np.float64)
    ratio = (sz0 / (sz0 + sz1))[:, ::-1]
    dxy = np.stack((dx, dy), axis=-1)
    return ctr0 - dxy * ratio, ctr1 + dxy * (1 - ratio)


def bboxes_mesh_renderer_matcher(mesh0, mesh1, image_loader0, image_loader1, bboxes0, bboxes1, **kwargs):
    """feabas/matcher.py:781-861 for general triangulated meshes (one region, no collisions): both block stacks are
    rendered through the meshes from images resident in HBM (``renderer.MeshRenderer``), band-passed with their masks when
    ``sigma`` > 0 and cross-correlated, all on the device; only (dx, dy, conf) per block come back.
    image_loader0/1: a ``renderer.ResidentImage``, a 2-D array (pixel (0, 0) at the image-space origin) or a
    ``renderer.MeshRenderer`` built before (kept by the caller across spacings).  Returns (xy0, xy1, conf)."""
    from . import renderer as _rd
    batch_size = kwargs.get('batch_size', None)
    sigma = kwargs.get('sigma', 0.0)
    conf_mode = kwargs.get('conf_mode', const.FFT_CONF_MIRROR)
    pad = kwargs.get('pad', True)
    subpixel = kwargs.get('subpixel', False)
    tol = kwargs.get('affine_approx_tol', 0.0)
    mask_range = kwargs.get('mask_range', None)
    if kwargs.get('geodesic_mask', False):
        raise NotImplementedError('geodesic_mask=True is outside the device renderer (the default of the alignment configuration is False)')
    from .mesh import Mesh

    def as_mesh(m):                                                            # matcher.py:792-799: init dict or Mesh H5 file
        if isinstance(m, dict):
            m = dict(m)
            return Mesh(m.pop('vertices'), m.pop('triangles'), **m)
        return Mesh.from_h5(m) if isinstance(m, str) else m
    mesh0, mesh1 = as_mesh(mesh0), as_mesh(mesh1)
    bboxes0 = np.asarray(bboxes0).reshape(-1, 4)
    bboxes1 = np.asarray(bboxes1).reshape(-1, 4)
    num_blocks = bboxes0.shape[0]
    empty = (np.empty((0, 2)), np.empty((0, 2)), np.empty(0))
    if num_blocks == 0:
        return empty
    sz0 = np.round(common.bbox_sizes(bboxes0))
    sz1 = np.round(common.bbox_sizes(bboxes1))
    chg = np.flatnonzero(np.any(np.diff(sz0, axis=0), axis=-1) | np.any(np.diff(sz1, axis=0), axis=-1))
    edges = np.concatenate(([0], chg + 1, [num_blocks]))
    # batch_size is the reference's host-memory knob (matcher.py:811-817).  With merge_batches (default) the device path sizes
    # its batches for HBM instead -- at least 2^27 pixels per stack -- so that a round of small blocks is a few launches, not
    # hundreds; the batch only enters the result through the stack-wide np.ptp of the masked DoG (common.py:369), i.e. for
    # blocks that stick out of the mesh.  merge_batches=False reproduces the reference's batches exactly.
    merge = kwargs.get('merge_batches', True)
    if batch_size is not None and batch_size < num_blocks:
        parts = []
        for a, b in zip(edges[:-1], edges[1:]):
            bs = batch_size
            if merge:
                bs = max(bs, (1 << 27) // max(1, int(sz0[a, 0] * sz0[a, 1]), int(sz1[a, 0] * sz1[a, 1])))
            nbt = max(1, int(np.ceil((b - a) / bs)))
            parts.append(np.linspace(a, b, num=nbt + 1, endpoint=True))
        edges = np.unique(np.round(np.concatenate(parts)).astype(np.int32))
    own = []
    renders = []
    rwt = kwargs.get('render_weight_threshold', 0)
    for mesh, loader in ((mesh0, image_loader0), (mesh1, image_loader1)):
        if isinstance(loader, _rd.MeshRenderer):
            renders.append(loader)
        else:
            # renderer.py:59-61: only the triangles of materials that are rendered and weigh at least the threshold carry image
            shown = mesh.triangle_mask_for_render(render_weight_threshold=rwt)
            r = None if not shown.any() else _rd.MeshRenderer.from_mesh(mesh if shown.all() else mesh.submesh(shown), image_loader=loader, affine_approx_tol=tol)
            if r is None:
                for o in own:
                    o.free()
                return empty
            renders.append(r); own.append(r)
    lib, ctx = _lib.load(), _lib.ctx()
    xy0, xy1, conf = [], [], []
    try:
        for a, b in zip(edges[:-1], edges[1:]):
            bufs = []
            try:
                stacks, shapes, covered = [], [], True
                for r, bb in ((renders[0], bboxes0[a:b]), (renders[1], bboxes1[a:b])):
                    d_out, d_mask, shape, _ = r.render_stack_dev(bb, precise_mask=sigma > 0)
                    bufs += [d_out, d_mask]
                    if not d_mask.count_nonzero(int(np.prod(shape))):             # crop_multiple -> None: batch skipped, matcher.py:835-839
                        covered = False
                        break
                    if sigma > 0:
                        d_out = r.filter_stack_dev(d_out, d_mask, shape, sigma, mask_range=mask_range)
                        bufs.append(d_out)
                    stacks.append(d_out); shapes.append(shape)
                if not covered:
                    continue
                n = b - a
                d_res = _lib.DeviceBuffer(20 * n)
                bufs.append(d_res)
                _lib.check(lib.fb_ncc_batch_dev(ctx, stacks[0].ptr, stacks[1].ptr, n, 1, shapes[0][1], shapes[0][2], shapes[1][1], shapes[1][2],
                                                int(bool(pad)), int(bool(subpixel)), int(conf_mode), d_res.ptr, d_res.offset(8 * n), d_res.offset(16 * n)))
                raw = d_res.to_array((20 * n,), np.uint8)
                dx = raw[:8 * n].view(np.float64); dy = raw[8 * n:16 * n].view(np.float64); cf = raw[16 * n:].view(np.float32)
                p0, p1 = block_displacements_to_points(bboxes0[a:b], bboxes1[a:b], dx, dy)
                xy0.append(p0); xy1.append(p1); conf.append(cf.copy())
            finally:
                for d in bufs:
                    d.free()
    finally:
        for o in own:
            o.free()
    if not xy0:
        return empty
    return np.concatenate(xy0, axis=0), np.concatenate(xy1, axis=0), np.concatenate(conf, axis=0)


DEFAULT_AVG_DEFORM = 0.05                # feabas/config.py:32


class _RoundPlan:
    """Which spacing the next round of block matching runs at, and with which padding: the stepper of the C ABI
    (fb_schedule_*, csrc/fb_geom.hip -- feabas/matcher.py:567-716), shared with every other caller of the library."""

    def __init__(self, spacings, allow_enlarge=False, allow_dwell=0, max_spacing_skip=0, pad=None):
        sp = np.ascontiguousarray(spacings, dtype=np.float64).ravel()
        self._lib = _lib.load()
        self.count = int(sp.size)
        self._h = self._lib.fb_schedule_create(_lib.ptr(sp), sp.size, int(bool(allow_enlarge)), int(allow_dwell), int(max_spacing_skip),
                                               -1 if pad is None else int(bool(pad)))
        if not self._h:
            raise ValueError('no spacings')

    def due(self):
        """(spacing, is the smallest one, pad the blocks) of the round that is due; None when the walk is over"""
        sp, last, pad = C.c_double(), C.c_int(), C.c_int()
        if not self._lib.fb_schedule_round(self._h, C.byref(sp), C.byref(last), C.byref(pad)):
            return None
        return sp.value, bool(last.value), bool(pad.value)

    def advance(self, max_dis, multiplier=4.0):
        """report the largest displacement of the round; True = repeat it at the enlarged spacing before linking anything"""
        redo = C.c_int()
        _lib.check(self._lib.fb_schedule_advance(self._h, float(max_dis), float(multiplier), C.byref(redo)))
        return bool(redo.value)

    def close(self):
        if self._h:
            self._lib.fb_schedule_destroy(self._h)
            self._h = None


class _PairRelaxation:
    """The spring-linked pair of a block matcher: the two meshes, the matches of the current round as their links, and the
    relaxation that follows every round (feabas/matcher.py:551-566, 717-742)."""

    def __init__(self, mesh0, mesh1, stiffness_lambda, residue_mode, residue_len, link_weight_decay, render_weight_threshold=0):
        from . import optimizer
        self.meshes = (mesh0, mesh1)
        self.render_weight_threshold = render_weight_threshold     # matches in triangles of materials that weigh no more are dropped (matcher.py:557, 719)
        self.linear = mesh0.is_linear and mesh1.is_linear
        self.slm = optimizer.SLM([mesh0, mesh1], stiffness_lambda=stiffness_lambda)
        self.residue_mode, self.residue_len, self.decay = residue_mode, residue_len, link_weight_decay
        if residue_len > 0 and residue_mode not in ('huber', 'threshold'):
            raise ValueError(residue_mode)

    def _solve(self, tol, steps):
        if self.linear:
            self.slm.optimize_linear(tol=tol)
        else:
            self.slm.optimize_Newton_Raphson(max_newtonstep=steps, tol=tol)

    def seed(self, initial_matches):
        """matcher.py:555-566: a first alignment from matches the caller brings (affine cascade, rigid anneal, one relaxation);
        without them the current MOVING gear is the start"""
        m0, m1 = self.meshes
        if initial_matches is None:
            for m in self.meshes:
                m.anneal(gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_FIXED), mode=const.ANNEAL_COPY_EXACT)
            return
        xy0, xy1, weight = initial_matches[:3] if isinstance(initial_matches, (tuple, list)) else \
            (initial_matches.xy0, initial_matches.xy1, initial_matches.weight)
        self.slm.add_link_from_coordinates(m0.uid, m1.uid, xy0, xy1, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_INITIAL), weight=weight,
                                           render_weight_threshold=self.render_weight_threshold)
        self.slm.optimize_affine_cascade(start_gear=const.MESH_GEAR_FIXED, target_gear=const.MESH_GEAR_FIXED, svd_clip=None)
        self.slm.anneal(gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING), mode=const.ANNEAL_CONNECTED_RIGID)
        self._solve(1e-6 if self.linear else 1e-4, 5)

    def retire_links(self):
        """the matches of the previous rounds: dropped, or kept with decayed weights (link_weight_decay)"""
        if self.decay == 0:
            self.slm.clear_links()
        else:
            for lnk in self.slm.links:
                lnk._weight = lnk._weight * self.decay

    def link(self, xy0, xy1, weight):
        m0, m1 = self.meshes
        self.slm.add_link_from_coordinates(m0.uid, m1.uid, xy0, xy1, gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_MOVING), weight=weight,
                                           render_weight_threshold=self.render_weight_threshold)
        return len(self.slm.links)

    def relax(self, tol, more_rounds, snap_rigid=False):
        """relax; with a residue length the matches are re-weighted by what is left of them (huber / threshold) and, when that
        changed a weight and another round follows, relaxed again.  snap_rigid (the strip routes): a relaxation that moved a mesh by a
        whole-pixel translation to within 1e-6 -- every block of a round without sub-pixel peaks measured the same -- is taken as
        exactly that translation, like stitch_pipeline does: the next round's block grid rounds the bounding box of the moved mesh, and
        a translated image edge sits exactly on a rounding tie, which the noise of the solve would otherwise decide (it does in
        the reference; the two strip routes of this package then put the last row of blocks a pixel apart)"""
        before = [None if m.locked else (m.vertices(const.MESH_GEAR_MOVING).copy(), np.array(m.offset(const.MESH_GEAR_MOVING), dtype=np.float64))
                  for m in self.meshes] if snap_rigid else None
        self._relax(tol, more_rounds)
        if before is not None:
            for m, was in zip(self.meshes, before):
                if was is None:
                    continue
                d = m.vertices_w_offset(const.MESH_GEAR_MOVING) - (was[0] + was[1])
                whole = np.round(d.mean(axis=0))
                if np.abs(d - whole).max() < 1e-6:
                    m.set_vertices(was[0], const.MESH_GEAR_MOVING)          # the vertices as they were, the whole pixels in the offset
                    m.set_offset(was[1] + whole.reshape(np.shape(was[1])), const.MESH_GEAR_MOVING)

    def _relax(self, tol, more_rounds):
        self._solve(tol, 3)
        if self.residue_len > 0:
            if self.residue_mode == 'huber':
                self.slm.set_link_residue_huber(self.residue_len)
            else:
                self.slm.set_link_residue_threshold(self.residue_len)
            changed, _ = self.slm.adjust_link_weight_by_residue(relax_first=True)
            if changed and more_rounds:
                self._solve(tol, 3)

    @property
    def links(self):
        return self.slm.links


def _strain_of_matches(mesh0, mesh1, xy0, xy1, weight, stiffness_lambda, render_weight_threshold=0):
    """matcher.py:752-777: how much elastic energy the matches ask of the pair, as a fraction of the energy of the shape itself:
    the untouched meshes, brought together rigidly (affine cascade clipped to rotations) and relaxed once; strain = sqrt of
    dv^T K dv / v0^T K v0 summed over the free (or softer) mesh(es).  The energies are evaluated on the device."""
    from . import optimizer
    slm = optimizer.SLM([mesh0, mesh1], stiffness_lambda=stiffness_lambda)
    slm.add_link_from_coordinates(mesh0.uid, mesh1.uid, xy0, xy1, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_INITIAL), weight=weight,
                                  render_weight_threshold=render_weight_threshold)
    slm.optimize_affine_cascade(start_gear=const.MESH_GEAR_INITIAL, target_gear=const.MESH_GEAR_FIXED, svd_clip=(1, 1))
    slm.anneal(gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING), mode=const.ANNEAL_COPY_EXACT)
    if mesh0.is_linear and mesh1.is_linear:
        slm.optimize_linear(tol=1e-6)
    else:
        slm.optimize_Newton_Raphson(max_newtonstep=5, tol=1e-4)
    one_locked = mesh0.locked or mesh1.locked
    soft_avg = np.mean([m.soft_factor for m in slm.meshes])
    counted = [m for m in slm.meshes if ((not m.locked) if one_locked else (m.soft_factor <= soft_avg))]
    moved = shape = 0.0
    for m in counted:
        v0 = m.vertices(gear=const.MESH_GEAR_FIXED)
        dv = m.vertices(gear=const.MESH_GEAR_MOVING) - v0
        e_dv, e_v0 = m.stiffness_energy([dv - dv.mean(axis=0, keepdims=True), v0 - v0.mean(axis=0, keepdims=True)])
        moved += max(0.0, e_dv); shape += max(0.0, e_v0)
    return (moved / shape) ** 0.5


def iterative_xcorr_matcher_w_mesh(mesh0, mesh1, image_loader0, image_loader1, spacings, **kwargs):
    """feabas/matcher.py:430-778 for two general meshes over images resident in HBM: rounds of block NCC through the meshes
    (``bboxes_mesh_renderer_matcher``) alternate with the relaxation of the pair, coarse to fine.  Which spacing a round runs
    at comes from the library's round stepper (``_RoundPlan`` -> fb_schedule_*), the pair and its links live in
    ``_PairRelaxation``, the strain estimate in ``_strain_of_matches``; every arithmetic step (render, DoG, NCC, assembly,
    PCG, energies) runs on the device.  ``num_workers`` has nothing to distribute here (one device renders and correlates the
    whole round).  The batched strip form of the same loop is ``stitch_pipeline.StripBatchMatcher`` (one call = many pairs)."""
    from . import renderer as _rd
    get = kwargs.get
    conf_thresh = get('conf_thresh', 0.3)
    distributor = get('distributor', 'cartesian_bbox')
    residue_len = get('residue_len', 0)
    if residue_len < 0:                                 # in units of the section thickness (matcher.py:518-520)
        residue_len = max(1, abs(residue_len) * get('section_thickness', const.DEFAULT_THICKNESS) / mesh0.resolution)
    trace = get('trace', None)                          # a list: one record per round (debugging / tests)
    # the reference's 0.01 / max(1, max_dis) (matcher.py:685-688) is an iteration budget: at that tolerance the field in weakly
    # constrained corners depends on the Krylov path taken.  Like the strip pipeline (and the oracle) the solves here are
    # converged by default (relax_tol = 1e-9); relax_tol = None takes the reference's tolerance as it stands -- section_matcher's
    # default, where the two relaxations of a section pair are a third of the time at 1e-9 (2 x ~1 400 PCG iterations)
    relax_tol, opt_tol = get('relax_tol', 1e-9), get('opt_tol', None)
    relax_tol = np.inf if relax_tol is None else relax_tol
    affine_render = get('affine_approximated_render', True)
    snap_rigid = bool(get('snap_rigid', False))        # _PairRelaxation.relax; set by the strip routes of stitching_matcher
    failed = (None, None, 0, DEFAULT_AVG_DEFORM)
    spacings = np.array(spacings, dtype=np.float64).ravel()
    if np.any(spacings < 1):                            # fractions of the longer side of the overlap (matcher.py:541-551)
        bbox, valid = common.intersect_bbox(mesh0.bbox(gear=const.MESH_GEAR_MOVING), mesh1.bbox(gear=const.MESH_GEAR_MOVING))
        if not valid:
            return failed
        spacings[spacings < 1] *= max(bbox[2] - bbox[0], bbox[3] - bbox[1])
    untouched = (mesh0.copy(), mesh1.copy()) if get('compute_strain', True) else None
    rwt = get('render_weight_threshold', 0)
    pair = _PairRelaxation(mesh0, mesh1, get('stiffness_lambda', 1), get('residue_mode', 'huber'), residue_len, get('link_weight_decay', 0.0), rwt)
    pair.seed(get('initial_matches', None))
    plan = _RoundPlan(spacings, get('allow_enlarge', False), get('allow_dwell', 0), get('max_spacing_skip', 0), get('pad', None))
    images, borrowed = [], []
    for ld in (image_loader0, image_loader1):
        images.append(ld if isinstance(ld, _rd.ResidentImage) else _rd.ResidentImage(ld))
        if images[-1] is not ld:
            borrowed.append(images[-1])
    linked_once = False
    try:
        while True:
            rnd = plan.due()
            if rnd is None:
                break
            sp, last, pad = rnd
            if distributor == 'cartesian_bbox':
                boxes0, boxes1 = distributor_cartesian_bbox(mesh0, mesh1, sp, min_num_blocks=get('min_num_blocks', 2) if last else 1,
                                                            shrink_factor=get('shrink_factor', 1), zorder=True)
            else:
                # refinement regions: on the last spacing as asked; before it, 'both' (2) means none yet (matcher.py:572-590)
                rfm = get('refine_mode', 2)
                if isinstance(rfm, str):
                    rfm = 0 if rfm.lower() == 'none' else (1 if 'only' in rfm.lower() else 2)
                boxes0, boxes1 = distribute_matching_blocks(mesh0, mesh1, sp, dfunc=distributor, min_boundary_distance=get('min_boundary_distance', 0),
                                                            shrink_factor=get('shrink_factor', 1), zorder=True, refine_mode=rfm if (last or rfm != 2) else 0,
                                                            render_weight_threshold=get('render_weight_threshold', 0))
            if boxes0 is None:
                return failed
            if boxes0.shape[0] == 0:                    # the meshes do not overlap any more
                if not linked_once:
                    return failed
                break
            subpixel = get('subpixel', None)
            tol_render = ((0.1 if last else max(1, 0.02 * sp)) if affine_render else 0)
            xy0, xy1, conf = bboxes_mesh_renderer_matcher(mesh0, mesh1, images[0], images[1], boxes0, boxes1, batch_size=get('batch_size', None),
                                                          pad=pad, subpixel=bool(last) if subpixel is None else subpixel, affine_approx_tol=tol_render,
                                                          sigma=get('sigma', 0.0), conf_mode=get('conf_mode', const.FFT_CONF_MIRROR),
                                                          mask_range=get('mask_range', None), render_weight_threshold=rwt)
            good = conf > conf_thresh
            if not good.any():
                if not linked_once:
                    return failed
                break
            pair.retire_links()
            xy0, xy1, wt = xy0[good], xy1[good], conf[good]
            max_dis = float(np.sqrt(np.max(np.sum((xy0 - xy1) ** 2, axis=-1))))
            if trace is not None:
                trace.append(dict(sp=float(sp), blocks=int(conf.size), kept=int(good.sum()), max_dis=max_dis, pad=bool(pad),
                                  subpixel=bool(last) if subpixel is None else bool(subpixel), tol=float(tol_render), conf=conf.copy(), bboxes0=boxes0.copy(), bboxes1=boxes1.copy()))
            if plan.advance(max_dis):
                continue                                # the displacement outran the largest spacing: once more, with larger blocks
            if pair.link(xy0, xy1, wt) == 0:
                if not linked_once:
                    return failed
                break
            if max_dis > 0.1:
                pair.relax(min(relax_tol, 0.01 / max(1, max_dis)) if opt_tol is None else opt_tol, more_rounds=plan.due() is not None, snap_rigid=snap_rigid)
            if trace is not None:
                trace[-1]['field1'] = mesh1.vertices_w_offset(const.MESH_GEAR_MOVING) - mesh1.vertices_w_offset(const.MESH_GEAR_INITIAL)
                trace[-1]['solve'] = dict(getattr(pair.slm, 'last_solve', {}))
            linked_once = True
    finally:
        plan.close()
        for im in borrowed:
            im.free()
    if len(pair.links) == 0:
        return failed
    newest = pair.links[-1]
    xy0 = newest.xy0(gear=const.MESH_GEAR_INITIAL, use_mask=True, combine=True)
    xy1 = newest.xy1(gear=const.MESH_GEAR_INITIAL, use_mask=True, combine=True)
    weight = newest.weight(use_mask=True)
    strain = DEFAULT_AVG_DEFORM if untouched is None else _strain_of_matches(untouched[0], untouched[1], xy0, xy1, weight, get('stiffness_lambda', 1), rwt)
    return xy0, xy1, weight, strain


_SECTION_MATCHER_KW = {
    # honoured
    'initial_matches', 'spacings', 'sigma', 'batch_size', 'distributor', 'link_weight_decay', 'compute_strain', 'stiffness_multiplier_threshold',
    'render_weight_threshold', 'stiffness_lambda', 'conf_thresh', 'residue_mode', 'residue_len', 'opt_tol', 'min_num_blocks', 'shrink_factor',
    'allow_dwell', 'allow_enlarge', 'pad', 'subpixel', 'max_spacing_skip', 'affine_approximated_render', 'conf_mode', 'min_boundary_distance',
    'section_thickness', 'trace', 'relax_tol', 'merge_batches', 'mask_range', 'refine_mode',
    # accepted and without effect on the device path, each for a stated reason
    'num_workers',          # one device renders and correlates a whole round: nothing to distribute over a process pool
    'callback_settings',    # the PCG runs to its tolerance: no early-stop / timeout exits
    'check_duplicates',     # Link.from_coordinates option of the reference; duplicates do not occur in lattice matches
    'geodesic_mask',        # False only
}


def section_matcher(mesh0, mesh1, image_loader0, image_loader1, **kwargs):
    """feabas/matcher.py:370-427 with the reference's defaults (``distributor='cartesian_region'``, sigma 2.5, batch_size 100,
    stiffness_lambda 0.5, thresholds 0.1): triangles of soft materials are dropped first (``stiffness_multiplier_threshold``);
    two connected meshes (or no initial matches) go straight to ``iterative_xcorr_matcher_w_mesh``; otherwise the meshes are
    cut into their connected parts, the initial matches are dealt to the part pairs (``SLM.divide_disconnected_submeshes``)
    and every pair is matched on its own.  Image loaders: ``renderer.ResidentImage``, a 2-D array, or any object with the
    ``crop(bbox, **kw)`` method of the reference's loaders (``ResidentImage.from_loader``).
    A keyword this mirror does not know raises instead of being swallowed."""
    from . import optimizer
    from . import renderer as _rd
    kwargs = dict(kwargs)
    unknown = set(kwargs) - _SECTION_MATCHER_KW
    if unknown:
        raise TypeError(f'section_matcher: keyword(s) not honoured by the device path: {sorted(unknown)}')
    if kwargs.get('geodesic_mask', False):
        raise NotImplementedError('geodesic_mask=True is outside the device renderer')
    initial_matches = kwargs.pop('initial_matches', None)
    spacings = kwargs.pop('spacings', [100])
    kwargs.setdefault('sigma', 2.5)
    kwargs.setdefault('batch_size', 100)
    kwargs.setdefault('distributor', 'cartesian_region')
    kwargs.setdefault('link_weight_decay', 0.0)
    kwargs.setdefault('relax_tol', None)                 # the reference's stopping tolerance of the relaxations (matcher.py:685-688)
    compute_strain = kwargs.pop('compute_strain', False)
    stiffness_multiplier_threshold = kwargs.get('stiffness_multiplier_threshold', 0.1)
    kwargs.setdefault('render_weight_threshold', 0.1)
    stiffness_lambda = kwargs.setdefault('stiffness_lambda', 0.5)
    if stiffness_multiplier_threshold > 0:
        mesh0 = mesh0.submesh(mesh0.triangle_mask_for_stiffness(stiffness_multiplier_threshold=stiffness_multiplier_threshold))
        mesh1 = mesh1.submesh(mesh1.triangle_mask_for_stiffness(stiffness_multiplier_threshold=stiffness_multiplier_threshold))
    # loaders of the reference (dal.*Loader: crop(bbox)) become resident images once, over the area the meshes can reach
    images, own = [], []
    for m, ld in ((mesh0, image_loader0), (mesh1, image_loader1)):
        if isinstance(ld, (_rd.ResidentImage, _rd.MeshRenderer, np.ndarray)):
            images.append(ld)
        elif hasattr(ld, 'crop'):
            b = m.bbox(gear=const.MESH_GEAR_INITIAL)
            pad_px = int(np.ceil(4 * kwargs['sigma'])) + 8
            images.append(_rd.ResidentImage.from_loader(ld, (int(np.floor(b[0])) - pad_px, int(np.floor(b[1])) - pad_px,
                                                             int(np.ceil(b[2])) + pad_px, int(np.ceil(b[3])) + pad_px)))
            own.append(images[-1])
        else:
            images.append(ld)
    try:
        if (initial_matches is None) or (mesh0.connected_triangles()[0] == 1 and mesh1.connected_triangles()[0] == 1):
            return iterative_xcorr_matcher_w_mesh(mesh0, mesh1, images[0], images[1], spacings=spacings, initial_matches=initial_matches,
                                                  compute_strain=compute_strain, **kwargs)
        opt = optimizer.SLM([mesh0, mesh1], stiffness_lambda=stiffness_lambda)
        xy0, xy1, weight = initial_matches[:3] if isinstance(initial_matches, (tuple, list)) else \
            (initial_matches.xy0, initial_matches.xy1, initial_matches.weight)
        opt.add_link_from_coordinates(mesh0.uid, mesh1.uid, xy0, xy1, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_INITIAL), weight=weight)
        opt.divide_disconnected_submeshes(prune_links=True)
        xy0, xy1, weight = [], [], []
        strain = DEFAULT_AVG_DEFORM
        for lnk in opt.links:
            m0_t, m1_t = lnk.meshes
            ini = (lnk.xy0(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True),
                   lnk.xy1(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True), lnk.weight(use_mask=False))
            xy0_t, xy1_t, wt_t, strain = iterative_xcorr_matcher_w_mesh(m0_t.copy(), m1_t.copy(), images[0], images[1], spacings=spacings,
                                                                        compute_strain=compute_strain, initial_matches=ini, **kwargs)
            if xy0_t is not None:
                if (m0_t.uid - m1_t.uid) * (mesh0.uid - mesh1.uid) > 0:
                    xy0.append(xy0_t); xy1.append(xy1_t)
                else:
                    xy0.append(xy1_t); xy1.append(xy0_t)
                weight.append(wt_t)
        if len(xy0) == 0:
            return None, None, 0, DEFAULT_AVG_DEFORM
        return np.concatenate(xy0, axis=0), np.concatenate(xy1, axis=0), np.concatenate(weight, axis=0), strain
    finally:
        for im in own:
            im.free()


def section_matcher_batch(jobs, threads=4):
    """``section_matcher`` for a list of section pairs -- what the aligner's matching stage deals to its workers
    (aligner.py: match_main, one pair per job).  jobs: sequence of ``(mesh0, mesh1, image_loader0, image_loader1)`` or
    ``(mesh0, mesh1, image_loader0, image_loader1, kwargs)``; ``threads`` host threads with a context (HIP stream) each take the
    jobs in turn, so the kernels and copies of one pair run while another thread is between two entries of the library (27 of the
    ~55 ms of a pair are python, which holds the interpreter lock; the entries do not): 37-40 pairs/s on four threads against
    17-18 one after the other (8192^2 sections, bench.py).  Returns the 4-tuples of ``section_matcher`` in job order; the first
    exception of a job is raised after the other threads have finished their current pair."""
    import threading
    jobs = list(jobs)
    if not jobs:
        return []
    nthr = max(1, min(int(threads), len(jobs)))
    main_ctx = _lib.ctx()
    results = [None] * len(jobs)
    errors = []
    take = threading.Lock()
    nxt = [0]

    def worker(t):
        # the contexts of the extra threads are kept between calls like those of stitching_matcher_batch (their arenas and code
        # objects are warm the second time; stitching_matcher_batch_release frees them)
        if t == 0:
            h = main_ctx
        else:
            slot = _batch_workers.setdefault((id(main_ctx), ('S', t)), {})
            if 'ctx' not in slot:
                slot['ctx'] = _lib.new_context()
            h = slot['ctx']
        try:
            with _lib.using(h):
                while not errors:
                    with take:
                        k = nxt[0]; nxt[0] += 1
                    if k >= len(jobs):
                        break
                    job = jobs[k]
                    kw = dict(job[4]) if len(job) > 4 and job[4] else {}
                    results[k] = section_matcher(job[0], job[1], job[2], job[3], **kw)
        except Exception as e:                              # noqa: BLE001 -- re-raised in the calling thread
            errors.append(e)
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nthr)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    if errors:
        raise errors[0]
    return results


def auto_spacings(shape0, shape1):
    """feabas/matcher.py:243-251."""
    shp = np.minimum(shape0, shape1)
    s_max = max(shp) * 0.25
    s_min = max(min(75, min(shp) / 3), 25)
    if s_min > s_max:
        return np.array([s_min])
    count = max(1, round(np.log(s_max / s_min) / np.log(4)))
    return np.exp(np.linspace(np.log(s_min), np.log(s_max), num=count, endpoint=True))


_pair_matchers = {}
_PAIR_MATCHER_CACHE = 4
_batch_workers = {}          # (process context, worker index) -> {ctx, state}: resources stitching_matcher_batch keeps
_pools = {}                  # context -> MatcherPool of the per-pair surface


def _forget_context(h):
    """a context is about to be destroyed: the per-pair matchers and the pool made under it are freed while it is still alive,
    so that nothing in these caches points at a dead stream (a later context may get the same address)"""
    key = id(h)
    with _lib.using(h):
        for k in [k for k in _pair_matchers if k[-1] == key]:
            _pair_matchers.pop(k).free()
        pool = _pools.pop(key, None)
        if pool is not None:
            pool.free()
        stitching_matcher_batch_release(h)


_lib.on_context_destroy(_forget_context)


def stitching_matcher_batch_release(main=None):
    """free the contexts, staging buffers and matchers that stitching_matcher_batch keeps between calls (main: only those of the
    calls made under that context)"""
    for (owner, t), slot in list(_batch_workers.items()):
        if main is not None and owner != id(main):
            continue
        _batch_workers.pop((owner, t))
        if t == 'slots':                                      # the staging slots shared by loaders and matchers
            for pin_, dev_ in slot.get('io', ()):
                pin_.free(); dev_.free()
            continue
        with _lib.using(slot.get('ctx')):                     # (the caller's own context is current again afterwards)
            st = slot.get('state', {})
            for r in tuple(st.get('matchers', {}).values()) + tuple(st.get('io', ())):
                r.free()
            if 'pool' in st:
                st['pool'].free()
        if t != 0:
            _lib.destroy_context(slot.get('ctx'))


def _stitching_options(kwargs):
    """matcher_config of stitching_matcher (matcher.py:229-241) -> (StripBatchMatcher options, mask0, mask1,
    compute_photometric); options the device path does not cover raise NotImplementedError"""
    kw = dict(kwargs)
    sigma = kw.pop('sigma', 2.5)
    coarse_downsample = kw.pop('coarse_downsample', 1)
    fine_downsample = kw.pop('fine_downsample', 1)
    residue_mode = kw.pop('residue_mode', 'huber')
    opts = dict(sigma=sigma, coarse_downsample=coarse_downsample, conf_thresh=kw.pop('conf_thresh', 0.3),
                min_num_blocks=kw.pop('min_num_blocks', 2), conf_mode=kw.pop('conf_mode', const.FFT_CONF_MIRROR),
                residue_len=kw.pop('residue_len', 5), stiffness_lambda=kw.pop('stiffness_lambda', 1),
                compute_strain=kw.pop('compute_strain', True), residue_mode=residue_mode)
    mask0 = kw.pop('mask0', None)
    mask1 = kw.pop('mask1', None)
    spacings = kw.pop('spacings', None)
    if spacings is not None:
        spacings = np.asarray(spacings, dtype=np.float64).ravel()
        if spacings.size == 0:
            raise ValueError('stitching_matcher: empty spacings')
    opts['spacings'] = spacings
    opts['_relative_spacings'] = spacings is not None and bool(np.any(spacings < 1))
    compute_photometric = bool(kw.pop('compute_photometric', False))
    if compute_photometric and not sigma > 0:
        raise NotImplementedError('stitching_matcher(compute_photometric=True) needs sigma > 0 on the device path')
    kw.pop('opt_tol', None); kw.pop('pad', None)
    if kw:
        raise NotImplementedError(f'stitching_matcher: unsupported options {sorted(kw)}')
    if residue_mode not in ('huber', 'threshold'):
        raise ValueError("stitching_matcher: residue_mode must be 'huber' or 'threshold' (matcher.py:730-735)")
    if not (0 < fine_downsample <= 1 and 0 < coarse_downsample <= 1):
        raise NotImplementedError('stitching_matcher: coarse_downsample and fine_downsample are shrinking factors in (0, 1]')
    # fine_downsample != 1 or a coarse_downsample other than 1 / 0.5: the general-mesh route (the batched strip pipeline is
    # built for full-resolution fine images and the x0.5 coarse level of default_stitching_configs.yaml:18-19)
    opts['_fine_downsample'] = fine_downsample
    opts['_general_scales'] = fine_downsample != 1 or coarse_downsample not in (1, 0.5)
    return opts, mask0, mask1, compute_photometric


def _check_strips(img0, img1, coarse_downsample, contiguous=True):
    img0 = np.asarray(img0); img1 = np.asarray(img1)
    if contiguous or img0.ndim != 2 or img0.strides[-1] != img0.itemsize:      # rows must be contiguous; a row pitch is fine for the batch packer
        img0 = np.ascontiguousarray(img0)
    if contiguous or img1.ndim != 2 or img1.strides[-1] != img1.itemsize:
        img1 = np.ascontiguousarray(img1)
    if img0.ndim != 2 or img1.ndim != 2 or img0.dtype != np.uint8 or img1.dtype != np.uint8:
        raise NotImplementedError('stitching_matcher: the device path takes two 2-D uint8 strips')
    if min(img0.shape) < 4 or min(img1.shape) < 4:
        raise NotImplementedError('stitching_matcher: strips thinner than 4 pixels are not on the device path')
    return img0, img1


def _overlap_statistics(raw0, raw1, flt0, flt1, valid0, valid1, tx, ty, band_passed):
    """photometric record of a pair (matcher.py:279-314): over the window where image 0, shifted by the integer translation
    (tx, ty), overlaps image 1, and where both masks hold -- (mean grey 0, mean grey 1, mean |DoG| 0, mean |DoG| 1) when the images
    were band-passed, (mean 0, mean 1, std 0, std 1) of the unfiltered images otherwise; None when image 0's mask leaves three pixels
    or fewer inside the window"""
    window, _ = common.intersect_bbox((tx, ty, flt0.shape[1] + tx, flt0.shape[0] + ty), (0, 0, flt1.shape[1], flt1.shape[0]))
    x_lo, y_lo, x_hi, y_hi = (int(v) for v in window)
    in0 = (slice(y_lo - ty, y_hi - ty), slice(x_lo - tx, x_hi - tx))
    in1 = (slice(y_lo, y_hi), slice(x_lo, x_hi))
    everywhere = np.ones((y_hi - y_lo, x_hi - x_lo), dtype=bool)
    ok0 = everywhere if valid0 is None else valid0[in0]
    ok1 = everywhere if valid1 is None else valid1[in1]
    if np.sum(ok0) <= 3:
        return None
    both = ok0 & ok1
    if band_passed:
        return (np.mean(raw0[in0][both]), np.mean(raw1[in1][both]), np.mean(np.abs(flt0[in0][both])), np.mean(np.abs(flt1[in1][both])))
    return (np.mean(flt0[in0][both]), np.mean(flt1[in1][both]), np.std(flt0[in0][both]), np.std(flt1[in1][both]))


def _stitching_matcher_general(img0, img1, opts, mask0, mask1, compute_photometric):
    """The pairs the batched strip pipeline does not take -- strips of UNEQUAL shape (the reference works on whatever the two
    crops are, matcher.py:244), spacings relative to the overlap (< 1, matcher.py:343-350), shrinking factors other than the
    defaults -- through the same stages as feabas/matcher.py:224-367: area resize, DoG and the global translation on the
    device, the statistics of the overlap when asked for, two cartesian meshes of the fine images' sizes and the general-mesh
    loop ``iterative_xcorr_matcher_w_mesh`` (device renderer + NCC + SLM); coordinates come back in pixels of the strips."""
    from . import renderer as _rd
    from .mesh import Mesh
    sigma, cds = opts['sigma'], opts['coarse_downsample']
    fds = opts.get('_fine_downsample', 1)
    conf_thresh, conf_mode, mnb = opts['conf_thresh'], opts['conf_mode'], opts['min_num_blocks']
    spacings = opts['spacings']
    spacings = auto_spacings(img0.shape, img1.shape) if spacings is None else np.array(spacings, dtype=np.float64)

    def shrink(img, mk, f):
        """cv2.resize(INTER_AREA) of the image, cv2.resize(INTER_NEAREST) of its mask (matcher.py:254-266, 318-335)"""
        if f == 1:
            return img, (None if mk is None else np.asarray(mk, dtype=bool))
        small = common.area_downsample2(img) if f == 0.5 else common.area_resize(img, f)
        return small, (None if mk is None else common.nearest_resize_mask(mk, f)[:small.shape[0], :small.shape[1]])

    def dog(img, s_, mk):
        return common.masked_dog_filter(img, s_, mask=mk) if sigma > 0 else img.astype(np.float32)
    r0, mg0 = shrink(img0, mask0, cds)
    r1, mg1 = shrink(img1, mask1, cds)
    g0, g1 = dog(r0, sigma * cds, mg0), dog(r1, sigma * cds, mg1)
    tx0, ty0, conf0 = global_translation_matcher(g0, g1, conf_mode=conf_mode, conf_thresh=conf_thresh)
    if conf0 < conf_thresh:
        return None, None, conf_thresh, None, None
    phtm = _overlap_statistics(r0, r1, g0, g1, mg0, mg1, int(tx0), int(ty0), sigma > 0) if compute_photometric else None
    if fds == cds:                                                            # matcher.py:315-317
        f0, f1 = g0, g1
    else:                                                                     # matcher.py:318-337
        b0, mf0 = shrink(img0, mask0, fds)
        b1, mf1 = shrink(img1, mask1, fds)
        f0, f1 = dog(b0, sigma * fds, mf0), dog(b1, sigma * fds, mf1)
    tx0, ty0 = tx0 * fds / cds, ty0 * fds / cds                               # matcher.py:338-339
    residue_len = opts['residue_len'] * fds                                   # matcher.py:341
    if np.any(spacings < 1):                                                  # matcher.py:343-350
        bb, _ = common.intersect_bbox(np.array((0, 0, f0.shape[1], f0.shape[0])) + np.tile((tx0, ty0), 2), (0, 0, f1.shape[1], f1.shape[0]))
        spacings = spacings.copy()
        spacings[spacings < 1] *= max(bb[2] - bb[0], bb[3] - bb[1])
    spacings = spacings * fds                                                 # matcher.py:352
    min_spacing = float(np.min(spacings))
    mesh0 = Mesh.from_bbox((0, 0, f0.shape[1], f0.shape[0]), cartesian=True, mesh_size=min_spacing, min_num_blocks=mnb, uid=0)
    mesh1 = Mesh.from_bbox((0, 0, f1.shape[1], f1.shape[0]), cartesian=True, mesh_size=min_spacing, min_num_blocks=mnb, uid=1)
    mesh0.apply_translation((tx0, ty0), const.MESH_GEAR_FIXED)
    mesh0.lock()
    im0, im1 = _rd.ResidentImage(np.ascontiguousarray(f0, dtype=np.float32)), _rd.ResidentImage(np.ascontiguousarray(f1, dtype=np.float32))
    try:
        xy0, xy1, weight, strain = iterative_xcorr_matcher_w_mesh(mesh0, mesh1, im0, im1, spacings=spacings, distributor='cartesian_bbox',
                                                                  residue_len=residue_len, residue_mode=('threshold' if opts['residue_mode'] == 'threshold' else 'huber'),
                                                                  conf_thresh=conf_thresh, conf_mode=conf_mode, min_num_blocks=mnb,
                                                                  stiffness_lambda=opts['stiffness_lambda'], compute_strain=opts['compute_strain'], snap_rigid=True)
    finally:
        im0.free(); im1.free()
    if xy0 is None:
        return None, None, conf_thresh, None, None
    if fds != 1:                                                              # matcher.py:365-367
        xy0, xy1 = common.scale_coordinates(xy0, 1 / fds), common.scale_coordinates(xy1, 1 / fds)
    return xy0, xy1, weight, strain, phtm


def stitching_matcher(img0, img1, **kwargs):
    """feabas/matcher.py:224-367 for one pair of overlap strips: returns ``(xy0, xy1, weight, strain, phtm)`` in
    strip-local pixel coordinates, or ``(None, None, conf_thresh, None, None)`` when the strips do not match
    (matcher.py:278) -- no exception for "no match".

    Equal-shape 2-D uint8 strips with ``coarse_downsample`` in (1, 0.5) and ``fine_downsample = 1`` (the defaults of
    default_stitching_configs.yaml:18-19) run through the batch pipeline (``stitch_pipeline.StripBatchMatcher`` with a batch
    of one; callers with many pairs use ``stitching_matcher_batch``): automatic or explicit spacings in pixels, optional masks
    (mask0 / mask1, True = valid pixel) and photometric statistics, mesh relaxations between spacings of any shape (rigid or
    deformed mesh1), ``residue_mode`` 'huber' or 'threshold'.  Strips of unequal shape, spacings relative to the overlap
    (< 1) and any other shrinking factors 0 < ``coarse_downsample``, ``fine_downsample`` <= 1 take the general-mesh route
    (``_stitching_matcher_general``: area resize, DoG, global translation, renderer + NCC + SLM on the device, pair by pair).
    Enlarging factors raise NotImplementedError."""
    from .stitch_pipeline import StripBatchMatcher
    opts, mask0, mask1, compute_photometric = _stitching_options(kwargs)
    img0, img1 = _check_strips(img0, img1, opts['coarse_downsample'])
    relative = opts.pop('_relative_spacings')
    scales = opts.pop('_general_scales')
    if img0.shape != img1.shape or relative or scales:
        return _stitching_matcher_general(img0, img1, opts, mask0, mask1, compute_photometric)
    opts.pop('_fine_downsample')
    H, W = img0.shape
    spacings = opts['spacings']
    key = (H, W) + tuple(None if v is None else (tuple(v.tolist()) if isinstance(v, np.ndarray) else v) for v in opts.values()) + (id(_lib.ctx()),)
    m = _pair_matchers.get(key)
    if m is None:
        from .stitch_pipeline import MatcherPool
        pool = _pools.setdefault(id(_lib.ctx()), MatcherPool())
        m = StripBatchMatcher(1, H, W, pool=pool, **opts)
        # strip shapes vary from pair to pair (stitcher.py:561-571): keep the device buffers of a few recent shapes only
        while len(_pair_matchers) >= _PAIR_MATCHER_CACHE:
            _pair_matchers.pop(next(iter(_pair_matchers))).free()
        _pair_matchers[key] = m
    else:
        _pair_matchers[key] = _pair_matchers.pop(key)              # most recently used last
    d0 = _lib.DeviceBuffer.from_array(img0); d1 = _lib.DeviceBuffer.from_array(img1)
    try:
        for name, mk in (('mask0', mask0), ('mask1', mask1)):
            if mk is not None and np.asarray(mk).shape != (H, W):
                raise ValueError(f'stitching_matcher: {name} must have the shape of its strip')
        out = m.match(d0.ptr, d1.ptr, masks0=None if mask0 is None else [mask0], masks1=None if mask1 is None else [mask1],
                      compute_photometric=compute_photometric)
        res = StripBatchMatcher.per_pair(out)[0]
    finally:
        d0.free(); d1.free()
    if res['xy0'] is None:
        return None, None, opts['conf_thresh'], None, None
    return res['xy0'], res['xy1'], res['weight'], res['strain'], (out['phtm'][0] if compute_photometric else None)


def stitching_matcher_batch(pairs, batch=32, threads=2, **kwargs):
    """``stitching_matcher`` for a list of host-resident strip pairs -- what ``Stitcher.subprocess_match_list_of_overlaps``
    (stitcher.py:552-613) does one pair at a time.  pairs: sequence of ``(img0, img1)`` or ``(img0, img1, mask0, mask1)``;
    kwargs: the matcher_config of ``stitching_matcher`` (masks per pair, not as options).  Pairs are bucketed by strip
    shape, every bucket is cut into chunks of ``batch`` pairs and the chunks are dealt to ``threads`` host threads, each
    with its own context (HIP stream), page-locked staging buffer and ``StripBatchMatcher``, so that packing and the
    host-to-device copy of one chunk overlap the kernels of another.  Returns the 5-tuples in input order (the
    reference's no-match tuple where the strips do not match); a pair the device path cannot take raises like
    ``stitching_matcher``."""
    import threading
    from .stitch_pipeline import StripBatchMatcher
    for name in ('mask0', 'mask1'):
        if kwargs.get(name, None) is not None:
            raise ValueError('stitching_matcher_batch: masks are given per pair, as (img0, img1, mask0, mask1)')
    opts, _, _, compute_photometric = _stitching_options(kwargs)
    relative = opts.pop('_relative_spacings') or opts.pop('_general_scales')
    opts.pop('_general_scales', None)
    gen_opts = dict(opts)
    opts.pop('_fine_downsample')
    items = []
    general = {}
    for k, pr in enumerate(pairs):
        img0, img1 = _check_strips(pr[0], pr[1], opts['coarse_downsample'], contiguous=False)
        mk0, mk1 = (pr[2], pr[3]) if len(pr) > 2 else (None, None)
        for mk, im in ((mk0, img0), (mk1, img1)):
            if mk is not None and np.asarray(mk).shape != im.shape:
                raise ValueError('stitching_matcher_batch: a mask must have the shape of its strip')
        if img0.shape != img1.shape or relative:
            # two crops of different size (an overlap clipped by a tile border on one side only), or spacings relative to the
            # overlap: the general-mesh route, pair by pair
            general[k] = _stitching_matcher_general(np.ascontiguousarray(img0), np.ascontiguousarray(img1), gen_opts, mk0, mk1, compute_photometric)
            img0 = img1 = None
        items.append((img0, img1, mk0, mk1))
    if not items:
        return []
    if general:
        keep = [k for k in range(len(items)) if k not in general]
        rest = stitching_matcher_batch([pairs[k] for k in keep], batch=batch, threads=threads, **kwargs) if keep else []
        out = [None] * len(items)
        for k, r in zip(keep, rest):
            out[k] = r
        for k, r in general.items():
            out[k] = r
        return out
    # chunks: ('uniform', (H, W), indices) -- pairs of one strip shape, through StripBatchMatcher; ('ragged', key, indices) -- pairs of unequal shape that share the
    # mesh topology and the number of spacings, through RaggedStripBatchMatcher (strips differ in shape from pair to pair
    # in a real section, stitcher.py:561-571; a batch per shape would be a batch of one)
    from .stitch_pipeline import RaggedStripBatchMatcher, MatcherPool
    by_shape = {}
    for k, it in enumerate(items):
        by_shape.setdefault(it[0].shape, []).append(k)
    chunks, loose = [], []
    for shape, idx in by_shape.items():
        # (masked pairs and photometric statistics go through ragged chunks like any other pair since round 6: the ragged
        # matcher filters a masked strip inside its slot and takes the statistics over every pair's own overlap)
        full = len(idx) // batch * batch if len(idx) >= batch else 0
        for c in range(0, full, batch):
            chunks.append(('uniform', shape, idx[c:c + batch]))
        loose.extend(idx[full:])
    by_key = {}
    for k in loose:
        H, W = items[k][0].shape
        by_key.setdefault(RaggedStripBatchMatcher.bucket_key(H, W, opts['min_num_blocks'], opts['spacings']), []).append(k)
    for key, idx in by_key.items():
        for c in range(0, len(idx), batch):
            part = idx[c:c + batch]
            if len({items[k][0].shape for k in part}) == 1:
                chunks.append(('uniform', items[part[0]][0].shape, part))
            else:
                chunks.append(('ragged', key, part))
    # largest chunks first (the remainders of the shape / mesh-grid buckets end the list: a short drain, and no thread starts a
    # 32-pair chunk while the others are done); the order of the results does not depend on it.  (Round 6 also tried cutting the
    # first and last chunks into quarters against the fill and drain of the loader -> matcher pipeline: 0.82 instead of 0.86 of
    # the resident rate -- a small chunk costs more per pair than it saves in waiting.  And letting every matcher thread stage its own
    # first chunk beside the loaders, so that all matchers start after one staging time: eight threads packing at once take 12-18 ms
    # per chunk instead of 8, the matchers start at 15-25 ms instead of 7-17: 9.0 k against 9.3 k pairs/s.)
    chunks.sort(key=lambda ch: -len(ch[2]))
    results = [None] * len(items)
    errors = []
    deferred = []

    def slot_bytes(kind, idx):
        hm = max(items[k][0].shape[0] for k in idx); wm = max(items[k][0].shape[1] for k in idx)
        return 2 * len(idx) * hm * wm
    need = max((slot_bytes(kind, idx) for kind, _, idx in chunks), default=0)
    nthr = max(1, min(int(threads), len(chunks)))
    main_ctx = _lib.ctx()
    okey = tuple(None if v is None else (tuple(v.tolist()) if isinstance(v, np.ndarray) else v) for v in opts.values())
    # Two kinds of host threads share the chunks: LOADERS pack the strips of a chunk into a page-locked stack and copy it to the
    # device on a stream of their own, MATCHERS run the strip matcher on a stack that has landed.  A chunk waits in one of a few
    # staging slots in between, so the copies of the next chunks overlap the kernels of the current ones whatever the number of
    # chunks per thread (with every thread doing both in turn the device idled while all of them were copying: 2 chunks per
    # thread at 1024 pairs).  Below three threads each thread does both.
    n_load = 0 if nthr < 3 else max(1, (3 * nthr) // 8)
    if nthr >= 3 and os.environ.get('FEABAS_HIP_INGEST_LOADERS'):
        n_load = max(1, min(nthr - 1, int(os.environ['FEABAS_HIP_INGEST_LOADERS'])))
    n_match = nthr - n_load
    import queue
    shared = _batch_workers.setdefault((id(main_ctx), 'slots'), {})
    n_slots = (n_match + n_load + 1) if n_load else 0
    if n_load and (shared.get('bytes', 0) < need or len(shared.get('io', ())) < n_slots):
        for pin_, dev_ in shared.get('io', ()):
            pin_.free(); dev_.free()
        shared['io'] = [(_lib.PinnedBuffer(need), _lib.DeviceBuffer(need)) for _ in range(n_slots)]
        shared['bytes'] = need
    free_q, ready_q = queue.Queue(), queue.Queue()
    for sl in range(n_slots):
        free_q.put(sl)
    # FEABAS_HIP_INGEST_TRACE=1: busy time of every loader and matcher thread (stage / match calls) against the wall, on stderr
    trace = [] if os.environ.get('FEABAS_HIP_INGEST_TRACE') else None
    _time = time
    t_start = _time.perf_counter()

    def timed(kind, who, fn, *a):
        if trace is None:
            return fn(*a)
        t0 = _time.perf_counter()
        try:
            return fn(*a)
        finally:
            trace.append((kind, who, t0 - t_start, _time.perf_counter() - t_start))
    next_chunk = [0]
    take = threading.Lock()

    def stage(chunk, pin, dev):
        """both strips of every pair of the chunk into the page-locked stack [2][n][Hm][Wm] (C++ memcpy, no interpreter lock) and
        on to the device (returns when the copy has completed)"""
        _, _, idx = chunk
        n = len(idx)
        Hm = max(items[k][0].shape[0] for k in idx); Wm = max(items[k][0].shape[1] for k in idx)
        srcs = (C.c_void_p * (2 * n))(*([items[k][0].ctypes.data for k in idx] + [items[k][1].ctypes.data for k in idx]))
        hs = np.array([items[k][0].shape[0] for k in idx] * 2, dtype=np.int32)
        ws = np.array([items[k][0].shape[1] for k in idx] * 2, dtype=np.int32)
        pitches = np.array([items[k][0].strides[0] for k in idx] + [items[k][1].strides[0] for k in idx], dtype=np.int64)
        lib_ = _lib.load()
        _lib.check(lib_.fb_host_pack2d(_lib.ctx(), pin.ptr, 2 * n, Hm, Wm, srcs, _lib.ptr(hs), _lib.ptr(ws), _lib.ptr(pitches), 2))
        _lib.check(lib_.fb_memcpy_h2d(_lib.ctx(), dev.ptr, pin.ptr, 2 * n * Hm * Wm))
        _lib.check(lib_.fb_sync(_lib.ctx()))
        return n, Hm, Wm

    def loader(j):
        slot = _batch_workers.setdefault((id(main_ctx), ('L', j)), {})
        if 'ctx' not in slot:
            slot['ctx'] = _lib.new_context()
        _lib.use_context(slot['ctx'])
        try:
            while not errors:
                with take:
                    c = next_chunk[0]; next_chunk[0] += 1
                if c >= len(chunks):
                    break
                sl = free_q.get()
                pin, dev = shared['io'][sl]
                ready_q.put((c, sl) + timed('stage', j, stage, chunks[c], pin, dev))
        except Exception as e:                                # noqa: BLE001 -- re-raised in the calling thread
            errors.append(e)
        finally:
            ready_q.put(None)                                 # one end mark per loader
            _lib.use_context(None)

    def match_chunk(state, chunk, dev, n, Hm, Wm):
        kind, what, idx = chunk
        if kind == 'uniform':
            cache = state.setdefault('matchers', {})           # (shape, pairs, options) -> StripBatchMatcher, the two used last
            key = (what, n, okey)
            m = cache.pop(key, None)
            if m is None:
                while len(cache) >= 2:
                    cache.pop(next(iter(cache))).free()
                m = StripBatchMatcher(n, Hm, Wm, pool=state['pool'], **opts)
            cache[key] = m
        else:
            m = RaggedStripBatchMatcher([items[k][0].shape for k in idx], pool=state['pool'], **opts)
        mk0 = [items[k][2] for k in idx]
        mk1 = [items[k][3] for k in idx]
        has_mask = any(v is not None for v in mk0 + mk1)
        try:
            out = m.match(dev.ptr, dev.offset(n * Hm * Wm), masks0=mk0 if has_mask else None, masks1=mk1 if has_mask else None,
                          compute_photometric=compute_photometric)
        finally:
            if kind == 'ragged':
                m.free()
        per = StripBatchMatcher.per_pair(out)
        for j, k in enumerate(idx):
            r = per[j]
            if r.get('deferred'):
                deferred.append(k)
            elif r['xy0'] is None:
                results[k] = (None, None, opts['conf_thresh'], None, None)
            else:
                results[k] = (r['xy0'], r['xy1'], r['weight'], r['strain'], out['phtm'][j] if compute_photometric else None)

    ended = [0]

    def worker(t):
        # worker t keeps its context, buffer pool and the matcher of the last uniform shape between calls
        # (stitching_matcher_batch_release frees them)
        slot = _batch_workers.setdefault((id(main_ctx), t), {})
        if 'ctx' not in slot:
            slot['ctx'] = main_ctx if t == 0 else _lib.new_context()
        _lib.use_context(slot['ctx'])
        state = slot.setdefault('state', {})
        try:
            if 'pool' not in state:
                state['pool'] = MatcherPool()
            if n_load == 0:
                # one or two threads: each stages its own chunks
                if state.get('bytes', 0) < need:
                    for r in state.get('io', ()):
                        r.free()
                    state['io'] = (_lib.PinnedBuffer(need), _lib.DeviceBuffer(need))
                    state['bytes'] = need
                pin, dev = state['io']
                for chunk in chunks[t::nthr]:
                    n, Hm, Wm = stage(chunk, pin, dev)
                    match_chunk(state, chunk, dev, n, Hm, Wm)
                return
            while True:
                got = ready_q.get()
                if got is None:
                    with take:
                        ended[0] += 1
                        last = ended[0] >= n_load
                    if last:                                  # every loader is done: wake the other matchers and leave
                        for _ in range(n_match):
                            ready_q.put('stop')
                    continue
                if got == 'stop':
                    break
                c, sl, n, Hm, Wm = got
                try:
                    if not errors:
                        timed('match', t, match_chunk, state, chunks[c], shared['io'][sl][1], n, Hm, Wm)
                finally:
                    free_q.put(sl)
        except Exception as e:                                # noqa: BLE001 -- re-raised in the calling thread
            errors.append(e)
            if n_load:
                # let the loaders run out: this thread keeps taking what they post -- staged chunks (their slots go back, a
                # loader may be waiting for one), and END MARKS, which it counts like a healthy matcher does: the 'stop's of
                # the other matchers are posted by whoever takes the last mark, and a mark swallowed here would leave them
                # waiting in ready_q.get() for ever.  Its own 'stop' ends the loop.
                while True:
                    try:
                        got = ready_q.get(timeout=0.05)
                    except queue.Empty:
                        with take:
                            done = ended[0] >= n_load
                        if done and all(not th.is_alive() for th in lths):
                            break
                        continue
                    if got is None:
                        with take:
                            ended[0] += 1
                            last = ended[0] >= n_load
                        if last:
                            for _ in range(n_match):
                                ready_q.put('stop')
                    elif got == 'stop':
                        break
                    else:
                        free_q.put(got[1])
        finally:
            _lib.use_context(None)
    lths = [threading.Thread(target=loader, args=(j,)) for j in range(n_load)]
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(n_match)]
    for th in lths + ths:
        th.start()
    for th in lths + ths:
        th.join()
    if trace is not None:
        wall = _time.perf_counter() - t_start
        for kind in ('stage', 'match'):
            who = sorted({w for k_, w, _, _ in trace if k_ == kind})
            busy = [sum(b - a for k_, w, a, b in trace if k_ == kind and w == x) for x in who]
            calls = [sum(1 for k_, w, _, _ in trace if k_ == kind and w == x) for x in who]
            first = [min(a for k_, w, a, b in trace if k_ == kind and w == x) for x in who]
            last = [max(b for k_, w, a, b in trace if k_ == kind and w == x) for x in who]
            sys.stderr.write(f'stitching_matcher_batch {kind}: wall {1e3 * wall:.1f} ms, threads {len(who)}, calls {calls}, busy ms {[round(1e3 * v, 1) for v in busy]}, '
                             f'mean call {1e3 * sum(busy) / max(1, sum(calls)):.2f} ms, first start ms {[round(1e3 * v, 1) for v in first]}, '
                             f'last end ms {[round(1e3 * v, 1) for v in last]}\n')
    if errors:
        raise errors[0]
    for k in deferred:
        results[k] = stitching_matcher(items[k][0], items[k][1], **kwargs)
    return results
