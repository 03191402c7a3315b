"""Device-resident tile-pair matcher: the NCC side of ``matcher.stitching_matcher``
(feabas/matcher.py:224-367) for a BATCH of overlap-strip pairs that already sit in HBM.

Per pair, in the reference's order:
  1. x0.5 area downsample (matcher.py:255-256)            fb_area_downsample2_dev
  2. DoG at sigma*0.5 (273-274)                             fb_dog_dev
  3. global translation, padded FFT (275 -> 138-158)        fb_ncc_batch_dev
  4. DoG at full resolution (336-337)                       fb_dog_dev
  5. coarse-to-fine block matching over the auto spacings (243-251, 578-745): blocks from
     distributor_cartesian_bbox (865-891), crops by integer translation
     (MeshRenderer.crop_multiple for a translated, undeformed mesh), xcorr_fft with the
     reference's pad / subpixel schedule (579-603, 690-716), block -> point pairs (840-849)
                                                            fb_ncc_blocks_dev
  6. last-round mesh relaxation + huber residue weights (725-737) and the strain estimate (752-777)
     for the whole batch as one block-diagonal system             fb_sys_update_links / _form_groups / _solve_groups
The low-confidence second shot of global_translation_matcher (159-221): inside fb_match_strips; the numpy statement of
this module runs it per affected pair through the host mirror in matcher.py (block NCCs on the device).
Mesh relaxation between spacings (725-742): a uniform block displacement is applied as the rigid translation it relaxes
to (crops stay integer translations); any other field makes the pair DEFORMED: mesh1 keeps the relaxed displacement of
its nodes (fb_pairs_relax_bary: total displacement from the FIXED gear, huber re-weighting and second solve included),
the next round's blocks come from the deformed bounding box, image-1 windows are gathered through the renderer's tiers
(deformed.block_affines -> fb_ncc_blocks_affine_dev inside the NCC loaders; the exact piecewise-linear tier through
fb_remap_dev + fb_ncc_batch_dev), matches are located in the deformed triangles (deformed.locate) and reported in the
INITIAL gear through their barycentric coordinates (matcher.py:748-751).  SURVEY.md sec.8f rows 1-2.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from . import constant as const
from .matcher import auto_spacings, next_fast_len
from .mesh import Mesh
from . import deformed as dfm

DEFAULT_AVG_DEFORM = 0.05            # feabas/config.py:32


def _divide_bbox_batch(xmin, ymin, xmax, ymax, block_size, min_num_blocks):
    """feabas/common.py:380-409 for arrays of bounding boxes that share (Nx, Ny).
    Returns x0, y0 [P, Ny*Nx] int32 and the block width/height per pair."""
    wd = xmax - xmin
    ht = ymax - ymin
    nx = np.maximum(np.ceil(wd / block_size), min_num_blocks)
    ny = np.maximum(np.ceil(ht / block_size), min_num_blocks)
    dx = np.ceil(wd / nx).astype(np.int64)
    dy = np.ceil(ht / ny).astype(np.int64)
    return nx.astype(np.int64), ny.astype(np.int64), dx, dy


def _z_order_batch(ix, iy):
    """feabas/common.py:196-215 along the last axis (base 2, two dimensions)."""
    ix = ix - ix.min(axis=-1, keepdims=True)
    iy = iy - iy.min(axis=-1, keepdims=True)
    if ix.ndim == 2 and ix.shape[0] > 1 and np.array_equal(ix, np.broadcast_to(ix[:1], ix.shape)) and np.array_equal(iy, np.broadcast_to(iy[:1], iy.shape)):
        # the usual case: every pair of the group has the same block index grid
        return np.broadcast_to(_z_order_batch(ix[:1], iy[:1]), ix.shape)
    if ix.ndim == 2 and ix.shape[0] == 1:
        # one grid: the same few grids come back call after call (one per spacing and strip shape)
        key = (ix.shape[1], ix.tobytes(), iy.tobytes())
        hit = _ZORDER_CACHE.get(key)
        if hit is not None:
            return hit
        out = _z_order_rows(ix, iy)
        if len(_ZORDER_CACHE) < 256:
            _ZORDER_CACHE[key] = out
        return out
    return _z_order_rows(ix, iy)


_ZORDER_CACHE = {}


def _z_order_rows(ix, iy):
    sx = np.zeros_like(ix)
    sy = np.zeros_like(iy)
    level = 0
    while np.any(ix > 0) or np.any(iy > 0):
        sx = sx + (ix % 2) * (2 ** (2 * level))
        sy = sy + (iy % 2) * (2 ** (2 * level))
        ix = np.floor(ix / 2)
        iy = np.floor(iy / 2)
        level += 1
    return np.argsort(sx + 2 * sy, axis=-1, kind='stable')


_NFL = np.zeros(0, dtype=np.int64)


def _nfl_table(n):
    """next_fast_len(v) for v < n, shared by every matcher of the process (strip shapes vary from pair to pair)"""
    global _NFL
    if _NFL.size < n:
        grown = np.array([next_fast_len(v) for v in range(_NFL.size, max(n, 2 * _NFL.size, 8194))], dtype=np.int64)
        _NFL = np.concatenate((_NFL, grown))
    return _NFL


def grid_counts(H, W, mesh_size, min_num_blocks=2, max_aspect_ratio=2):
    """node counts (nx, ny) of Mesh.from_bbox((0, 0, W, H), cartesian=True) (mesh.py:403-435) without building the mesh"""
    wd, ht = float(W), float(H)
    nx = max(np.round(wd / mesh_size), min_num_blocks)
    ny = max(np.round(ht / mesh_size), min_num_blocks)
    dx, dy = wd / nx, ht / ny
    if dx > max_aspect_ratio * dy:
        dx = max_aspect_ratio * dy
    elif dy > max_aspect_ratio * dx:
        dy = max_aspect_ratio * dx
    return int(np.ceil(wd / dx)) + 1, int(np.ceil(ht / dy)) + 1


class MatcherPool:
    """Device buffers and relaxation systems that outlive a StripBatchMatcher.  Strip shapes vary from pair to pair in a
    real section (stitcher.py:561-571), so matchers come and go; their buffers are handed back here instead of to
    hipFree, and a relaxation system is kept per (pairs, grid) topology -- only its stiffness values are re-assembled
    for a new geometry.  One pool per context (host thread)."""

    def __init__(self):
        self.buffers = []
        self.systems = {}

    def take(self, nbytes):
        nbytes = int(nbytes)
        best = None
        for k, b in enumerate(self.buffers):
            if b.nbytes >= nbytes and b.nbytes <= 4 * max(nbytes, 4096) and (best is None or b.nbytes < self.buffers[best].nbytes):
                best = k
        if best is not None:
            return self.buffers.pop(best)
        return _lib.DeviceBuffer(nbytes)

    def give(self, buf):
        if buf is not None and buf.ptr is not None:
            self.buffers.append(buf)
            if len(self.buffers) > 24:                          # bound the pool: drop the smallest
                self.buffers.sort(key=lambda b: b.nbytes)
                self.buffers.pop(0).free()

    def free(self):
        for b in self.buffers:
            b.free()
        self.buffers = []
        for sysh in self.systems.values():
            _lib.load().fb_sys_destroy(_lib.ctx(), sysh)
        self.systems = {}


class _LazyBuffer:
    """device buffer of the host route, taken from the pool the first time that route touches it (a matcher that stays on
    fb_match_strips, which owns its own buffers, never allocates them)"""

    def __init__(self, name):
        self.key = '_buf_' + name

    def __get__(self, obj, cls=None):
        if obj is None:
            return self
        b = obj.__dict__.get(self.key)
        if b is None:
            nbytes = obj._buf_bytes.get(self.key)
            if nbytes is None:
                return None
            b = obj._pool.take(nbytes) if obj._pool is not None else _lib.DeviceBuffer(int(nbytes))
            obj.__dict__[self.key] = b
        return b

    def __set__(self, obj, val):
        obj.__dict__[self.key] = val


class StripBatchMatcher:
    d_small = _LazyBuffer('small')
    d_dogc = _LazyBuffer('dogc')
    d_dogf = _LazyBuffer('dogf')
    d_blk = _LazyBuffer('blk')
    d_out = _LazyBuffer('out')

    def __init__(self, P, H, W, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, min_num_blocks=2,
                 conf_mode=const.FFT_CONF_MIRROR, residue_len=5, stiffness_lambda=1.0, relax_tol=1e-9, compute_strain=True, spacings=None,
                 pool=None, residue_mode='huber', route=None):
        assert coarse_downsample in (0.5, 1)
        self._pool = pool
        # 'native': fb_match_strips (the whole sequence behind one C entry; pairs it hands back take the host route);
        # 'host': the numpy statement below for every pair.  Masks, photometric statistics and ragged batches are host only.
        self._route = route or os.environ.get('FEABAS_HIP_STRIP_ROUTE', 'native')
        if self._route not in ('native', 'host'):
            raise ValueError("route must be 'native' or 'host'")
        self._opts = dict(sigma=sigma, coarse_downsample=coarse_downsample, conf_thresh=conf_thresh, min_num_blocks=min_num_blocks, conf_mode=conf_mode,
                          residue_len=residue_len, stiffness_lambda=stiffness_lambda, relax_tol=relax_tol, compute_strain=compute_strain,
                          spacings=spacings, residue_mode=residue_mode)
        self._native = None
        self._general = {}
        self._gather = None
        self._prefer_host = False
        self.last_flags = None
        self.P, self.H, self.W = int(P), int(H), int(W)
        self.sigma = float(sigma)
        self.cds = coarse_downsample
        self.conf_thresh = float(conf_thresh)
        self.mnb = int(min_num_blocks)
        self.conf_mode = int(conf_mode)
        if spacings is None:
            self.spacings = np.sort(auto_spacings((H, W), (H, W)))[::-1]     # matcher.py:243-251, 567
        else:
            self.spacings = np.sort(np.asarray(spacings, dtype=np.float64).ravel())[::-1]
            if self.spacings.size == 0 or np.any(self.spacings < 1):
                raise NotImplementedError('spacings relative to the overlap (< 1, matcher.py:343-350) are not on the device path')
        self._nfl = _nfl_table(2 * max(H, W) + 2)
        # per-pair strip extents and spacing values ([P], [P, nsp]): constant here, set by RaggedStripBatchMatcher
        self._Hs = np.full(self.P, self.H, dtype=np.int64)
        self._Ws = np.full(self.P, self.W, dtype=np.int64)
        self._sp = np.tile(self.spacings, (self.P, 1))
        from .common import half_size
        hc, wc = (half_size(H), half_size(W)) if coarse_downsample == 0.5 else (H, W)      # cv2.resize(fx=0.5): cvRound(n / 2)
        self.hc, self.wc = hc, wc
        n = self.P
        self.max_blocks = n * 1024
        self._buf_bytes = {'_buf_small': 2 * n * hc * wc if coarse_downsample == 0.5 else None, '_buf_dogc': 2 * n * hc * wc * 4,
                           '_buf_dogf': 2 * n * H * W * 4, '_buf_blk': self.max_blocks * 9 * 4,
                           '_buf_out': self.max_blocks * 20}       # d_out per launch: [dx f64 N][dy f64 N][conf f32 N], one D2H copy
        self.residue_len = float(residue_len)                 # matcher.py:236 (fine_downsample = 1)
        if residue_mode not in ('huber', 'threshold'):
            raise ValueError("residue_mode must be 'huber' or 'threshold' (matcher.py:730-735)")
        self.residue_mode = 1 if residue_mode == 'threshold' else 0
        self.stiffness_lambda = float(stiffness_lambda)       # matcher.py:507
        self.relax_tol = float(relax_tol)
        self.compute_strain = bool(compute_strain)          # matcher.py:497
        self.last_strain_solve = None
        self._relax_sys = None
        self.last_relax = None
        self.last_tiers = {}
        self._ragged = False

    def free(self):
        pool = self._pool
        if self._relax_sys is not None:
            if pool is not None:
                pool.systems[self._sys_key] = self._relax_sys          # kept for the next matcher of this topology
            else:
                _lib.load().fb_sys_destroy(_lib.ctx(), self._relax_sys)
            self._relax_sys = None
        for key in ('_buf_small', '_buf_dogc', '_buf_dogf', '_buf_blk', '_buf_out'):
            b = self.__dict__.pop(key, None)
            if b is not None:
                pool.give(b) if pool is not None else b.free()
        if self._native is not None:
            _lib.load().fb_strip_matcher_destroy(_lib.ctx(), self._native)
            self._native = None
        for sub in self._general.values():
            sub.free()
        self._general = {}
        if self._gather is not None:
            for b in self._gather:
                pool.give(b) if pool is not None else b.free()
            self._gather = None
        for name in [k for k in vars(self) if k.startswith('_scr_')]:
            b = getattr(self, name)
            pool.give(b) if pool is not None else b.free()
            delattr(self, name)

    # ------------------------------------------------------------------ stages
    def _masked_dog(self, src_ptr, dtype, h, w, sigma, mask, dst_ptr):
        """common.masked_dog_filter(img, sigma, mask=mask) (common.py:353-377) of ONE resident image into its slot"""
        lib, ctx = _lib.load(), _lib.ctx()
        mk = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        assert mk.shape == (h, w)
        d_m = _lib.DeviceBuffer.from_array(mk)
        try:
            _lib.check(lib.fb_dog_dev(ctx, src_ptr, dtype, 1, h, w, sigma, d_m.ptr, 1, dst_ptr))
            _lib.check(lib.fb_sync(ctx))
        finally:
            d_m.free()

    def _global(self, strips0, strips1, masks=None, need_small=False):
        lib, ctx = _lib.load(), _lib.ctx()
        n, H, W, hc, wc = self.P, self.H, self.W, self.hc, self.wc
        if self.cds == 0.5 and masks is None and not need_small and int(4.0 * self.sigma * self.cds + 0.5) in (5, 6, 8, 10):
            # the x0.5 area downsample inside the loader of the coarse DoG: the coarse uint8 image (only the masked DoG and the
            # photometric statistics read it again) is never written
            _lib.check(lib.fb_dog_down2_pair_dev(ctx, strips0, strips1, n, H, W, self.sigma * self.cds, 1, self.d_dogc.ptr))
        elif self.cds == 0.5:
            _lib.check(lib.fb_area_downsample2_dev(ctx, strips0, n, H, W, self.d_small.ptr))
            _lib.check(lib.fb_area_downsample2_dev(ctx, strips1, n, H, W, self.d_small.offset(n * hc * wc)))
            _lib.check(lib.fb_dog_dev(ctx, self.d_small.ptr, 0, 2 * n, hc, wc, self.sigma * self.cds, None, 1, self.d_dogc.ptr))
        else:
            _lib.check(lib.fb_dog_pair_dev(ctx, strips0, strips1, 0, n, hc, wc, self.sigma, 1, self.d_dogc.ptr))
        if masks is not None:
            # masked pairs: the coarse DoG of their images again, with the halo suppression of common.py:368-374; the
            # coarse mask is cv2.resize(mask, fx=0.5, INTER_NEAREST) = every second pixel (matcher.py:257-264)
            for side, (strips, mlist) in enumerate(((strips0, masks[0]), (strips1, masks[1]))):
                for p, mk in enumerate(mlist):
                    if mk is None:
                        continue
                    mg = np.asarray(mk)[::2, ::2][:hc, :wc] if self.cds == 0.5 else np.asarray(mk)
                    src = self.d_small.offset((side * n + p) * hc * wc) if self.cds == 0.5 else C.c_void_p(strips + p * H * W)
                    self._masked_dog(src, 0, hc, wc, self.sigma * self.cds, mg, self.d_dogc.offset((side * n + p) * hc * wc * 4))
        _lib.check(lib.fb_ncc_batch_dev(ctx, self.d_dogc.ptr, self.d_dogc.offset(n * hc * wc * 4), n, 1, hc, wc, hc, wc,
                                        1, 0, self.conf_mode, self.d_out.ptr, self.d_out.offset(8 * n), self.d_out.offset(16 * n)))
        tx, ty, cf = (np.array(a) for a in self._fetch_out(n))    # equal strip sizes: (W1-W0)/2 = 0 (matcher.py:155-156)
        # low-confidence pairs get the second shot on ~6 sub-blocks (matcher.py:159-221).  Rare: the two coarse DoG
        # strips of such a pair are handed to the host mirror of global_translation_matcher, whose block NCCs run on
        # the device again.
        low = np.flatnonzero(~(cf > self.conf_thresh))
        if low.size:
            from .matcher import global_translation_matcher
            img_bytes = hc * wc * 4
            for p in low:
                g0 = np.empty((hc, wc), dtype=np.float32); g1 = np.empty((hc, wc), dtype=np.float32)
                _lib.check(lib.fb_memcpy_d2h(ctx, _lib.ptr(g0), self.d_dogc.offset(p * img_bytes), img_bytes))
                _lib.check(lib.fb_memcpy_d2h(ctx, _lib.ptr(g1), self.d_dogc.offset((n + p) * img_bytes), img_bytes))
                tx[p], ty[p], cf[p] = global_translation_matcher(g0, g1, conf_mode=self.conf_mode, conf_thresh=self.conf_thresh)
        return tx, ty, cf

    def _fetch_out(self, nb):
        """(dx, dy, conf) of the last launch that wrote nb results into d_out"""
        raw = self.d_out.to_array((20 * nb,), np.uint8)
        return raw[:8 * nb].view(np.float64), raw[8 * nb:16 * nb].view(np.float64), raw[16 * nb:].view(np.float32)

    def _fine_dog(self, strips0, strips1, masks=None):
        lib, ctx = _lib.load(), _lib.ctx()
        n, H, W = self.P, self.H, self.W
        if self.cds == 1:
            self.d_dogf_view = self.d_dogc        # matcher.py:315-317: same image when fine == coarse
            return
        _lib.check(lib.fb_dog_pair_dev(ctx, strips0, strips1, 0, n, H, W, self.sigma, 1, self.d_dogf.ptr))
        if masks is not None:
            for side, (strips, mlist) in enumerate(((strips0, masks[0]), (strips1, masks[1]))):
                for p, mk in enumerate(mlist):
                    if mk is not None:
                        self._masked_dog(C.c_void_p(strips + p * H * W), 0, H, W, self.sigma, mk, self.d_dogf.offset((side * n + p) * H * W * 4))
        self.d_dogf_view = self.d_dogf

    def _photometric(self, strips0, strips1, tx_c, ty_c, masks):
        """matcher.py:279-314 (sigma > 0) for every pair: mean grey level of the raw coarse images and mean |DoG| of the
        filtered ones over the overlap of the translated bounding boxes.  Host numpy on downloaded coarse images, like
        the reference; tx_c, ty_c: the global translation at the coarse scale."""
        n, H, W, hc, wc = self.P, self.H, self.W, self.hc, self.wc
        if self.cds == 0.5:
            raw = self.d_small.to_array((2 * n, hc, wc), np.uint8)
        else:
            raw = np.empty((2 * n, hc, wc), dtype=np.uint8)
            for side, strips in enumerate((strips0, strips1)):
                _lib.check(_lib.load().fb_memcpy_d2h(_lib.ctx(), _lib.ptr(raw[side * n:(side + 1) * n]), C.c_void_p(strips), n * hc * wc))
        dog = self.d_dogc.to_array((2 * n, hc, wc), np.float32)
        out = []
        for p in range(n):
            txx, tyy = int(tx_c[p]), int(ty_c[p])
            xa, ya = max(txx, 0), max(tyy, 0)
            xb, yb = min(wc + txx, wc), min(hc + tyy, hc)
            i0 = (slice(ya - tyy, yb - tyy), slice(xa - txx, xb - txx)); i1 = (slice(ya, yb), slice(xa, xb))
            shape = (max(yb - ya, 0), max(xb - xa, 0))

            def coarse(mk):
                if mk is None:
                    return None
                return np.asarray(mk, dtype=bool)[::2, ::2][:hc, :wc] if self.cds == 0.5 else np.asarray(mk, dtype=bool)
            mg0 = coarse(masks[0][p]) if masks is not None else None
            mg1 = coarse(masks[1][p]) if masks is not None else None
            m0 = np.ones(shape, dtype=bool) if mg0 is None else mg0[i0]
            m1 = np.ones(shape, dtype=bool) if mg1 is None else mg1[i1]
            mp = m0 & m1
            if np.sum(m0) <= 3:
                out.append(None)
                continue
            out.append((np.mean(raw[p][i0][mp]), np.mean(raw[n + p][i1][mp]), np.mean(np.abs(dog[p][i0][mp])), np.mean(np.abs(dog[n + p][i1][mp]))))
        return out

    def _blocks(self, tx, ty, t1, sel, spacing, mnb, bounds=None):
        """block descriptors for the pairs `sel` (all share Nx, Ny): returns (blk [Q, nblk, 9], bboxes [Q, nblk, 4]).
        bounds: (xmin, ymin, xmax, ymax) of the intersected mesh bounding boxes of `sel` when mesh1 is deformed."""
        H, W = self._Hs[sel], self._Ws[sel]
        spacing = np.broadcast_to(np.asarray(spacing, dtype=np.float64), (sel.size,))
        # mesh bounding boxes in the MOVING gear (Mesh.from_bbox: vertices at pixel centres - 0.5)
        if bounds is None:
            xmin = np.maximum(-0.5 + tx[sel], -0.5 + t1[sel, 0]); ymin = np.maximum(-0.5 + ty[sel], -0.5 + t1[sel, 1])
            xmax = np.minimum(W - 0.5 + tx[sel], W - 0.5 + t1[sel, 0]); ymax = np.minimum(H - 0.5 + ty[sel], H - 0.5 + t1[sel, 1])
        else:
            xmin, ymin, xmax, ymax = bounds
        nx, ny, dx, dy = _divide_bbox_batch(xmin, ymin, xmax, ymax, spacing, mnb)
        assert np.all(nx == nx[0]) and np.all(ny == ny[0])
        nxi, nyi = int(nx[0]), int(ny[0])
        xt = np.round(np.linspace(xmin, xmax - dx, num=nxi, endpoint=True, axis=-1)).astype(np.int32)    # [Q, nx]
        yt = np.round(np.linspace(ymin, ymax - dy, num=nyi, endpoint=True, axis=-1)).astype(np.int32)    # [Q, ny]
        x0 = np.broadcast_to(xt[:, None, :], (sel.size, nyi, nxi)).reshape(sel.size, -1)
        y0 = np.broadcast_to(yt[:, :, None], (sel.size, nyi, nxi)).reshape(sel.size, -1)
        order = _z_order_batch(np.round((x0 - x0.min(axis=-1, keepdims=True)) / spacing[:, None]),
                               np.round((y0 - y0.min(axis=-1, keepdims=True)) / spacing[:, None]))
        x0 = np.take_along_axis(x0, order, axis=-1)
        y0 = np.take_along_axis(y0, order, axis=-1)
        nblk = x0.shape[1]
        bb = np.stack((x0, y0, x0 + dx[:, None].astype(np.int32), y0 + dy[:, None].astype(np.int32)), axis=-1)
        blk = np.empty((sel.size, nblk, 9), dtype=np.int32)
        blk[:, :, 0] = sel[:, None]
        # image-0 window: output coordinate - mesh offset (renderer.crop_field: bbox - offset)
        blk[:, :, 1] = x0 - np.round(tx[sel]).astype(np.int32)[:, None]
        blk[:, :, 2] = y0 - np.round(ty[sel]).astype(np.int32)[:, None]
        blk[:, :, 3] = dy[:, None]; blk[:, :, 4] = dx[:, None]
        if t1 is None:                                       # deformed mesh1: the window comes from the affine / exact gather
            blk[:, :, 5] = 0; blk[:, :, 6] = 0
        else:
            blk[:, :, 5] = x0 - np.round(t1[sel, 0]).astype(np.int32)[:, None]
            blk[:, :, 6] = y0 - np.round(t1[sel, 1]).astype(np.int32)[:, None]
        blk[:, :, 7] = dy[:, None]; blk[:, :, 8] = dx[:, None]
        return blk, bb

    def _match_round(self, tx, ty, t1, active, spacing, mnb, pad_flags, subpixel):
        """one spacing round for the `active` pairs.  Returns a list of groups
        (pair ids [Q], bboxes [Q, nblk, 4], dx, dy, conf [Q, nblk])."""
        lib, ctx = _lib.load(), _lib.ctx()
        n, H, W = self.P, self.H, self.W                     # H, W: the slot size of the strip stacks
        Hs, Ws = self._Hs, self._Ws                          # per-pair strip extents
        spacing = np.broadcast_to(np.asarray(spacing, dtype=np.float64), (n,))
        # group by block grid, then by FFT shape (matcher.py:59-62 on the block size)
        xmin = np.maximum(-0.5 + tx, -0.5 + t1[:, 0]); xmax = np.minimum(Ws - 0.5 + tx, Ws - 0.5 + t1[:, 0])
        ymin = np.maximum(-0.5 + ty, -0.5 + t1[:, 1]); ymax = np.minimum(Hs - 0.5 + ty, Hs - 0.5 + t1[:, 1])
        nx, ny, dx, dy = _divide_bbox_batch(xmin, ymin, xmax, ymax, spacing, mnb)
        ok = (xmax > xmin) & (ymax > ymin)
        dx = np.where(ok, dx, 1); dy = np.where(ok, dy, 1)
        nfl_h = nfl_w = self._nfl
        fh = np.where(pad_flags, nfl_h[np.clip(2 * dy - 1, 0, None)], nfl_h[dy])
        fw = np.where(pad_flags, nfl_w[np.clip(2 * dx - 1, 0, None)], nfl_w[dx])
        key = ((nx * 4096 + ny) * 8192 + fh) * 8192 + fw
        key = np.where(active & ok, key, -1)
        groups = []
        dogf = self.d_dogf_view
        img1 = dogf.offset(n * H * W * 4)
        for kv in np.unique(key):
            if kv < 0:
                continue
            sel = np.flatnonzero(key == kv)
            gfh, gfw = int(fh[sel[0]]), int(fw[sel[0]])
            blk, bb = self._blocks(tx, ty, t1, sel, spacing[sel], mnb)
            nb = blk.shape[0] * blk.shape[1]
            assert nb <= self.max_blocks
            flat = np.ascontiguousarray(blk.reshape(-1, 9))
            _lib.check(lib.fb_memcpy_h2d(ctx, self.d_blk.ptr, _lib.ptr(flat), flat.nbytes))
            _lib.check(lib.fb_ncc_blocks_dev(ctx, dogf.ptr, img1, H, W, H, W, nb, self.d_blk.ptr, int(dy[sel].max()), int(dx[sel].max()), gfh, gfw,
                                             1 if subpixel else 0, self.conf_mode, self.d_out.ptr, self.d_out.offset(8 * nb), self.d_out.offset(16 * nb)))
            ddx, ddy, dcf = (a.reshape(sel.size, -1) for a in self._fetch_out(nb))
            groups.append((sel, bb, ddx, ddy, dcf))
        return groups

    # ------------------------------------------------------------------ deformed pairs
    def _scratch(self, name, nbytes):
        """grow-only device scratch of the rarely taken exact-field tier"""
        buf = getattr(self, '_scr_' + name, None)
        if buf is None or buf.nbytes < nbytes:
            if buf is not None:
                self._pool.give(buf) if self._pool is not None else buf.free()
            buf = self._pool.take(nbytes) if self._pool is not None else _lib.DeviceBuffer(int(nbytes))
            setattr(self, '_scr_' + name, buf)
        return buf

    def _match_round_deformed(self, tx, ty, U, pairs, spacing, mnb, pad_flags, subpixel, is_last):
        """one spacing round for pairs whose mesh1 is deformed (U [P, V, 2] = MOVING - INITIAL of its nodes): the block grid
        covers the intersection with the deformed bounding box (matcher.py:877), image-0 windows are integer crops
        (mesh0 is a translated grid), image-1 windows go through MeshRenderer.crop_multiple's tiers with
        affine_approx_tol = 0.1 in the last round, max(1, 0.02 spacing) before (matcher.py:578-603).  Same return as
        `_match_round`; `self.last_tiers[pair]` keeps the tier of every block."""
        lib, ctx = _lib.load(), _lib.ctx()
        n, H, W = self.P, self.H, self.W
        self._relax_system()
        m = self._mesh
        V = m.num_vertices
        v_all = self._v_init_u.reshape(n, V, 2)                              # INITIAL nodes of every pair's mesh1
        tris = m.triangles
        spacing = np.broadcast_to(np.asarray(spacing, dtype=np.float64), (n,))[pairs]
        tol_q = np.full(pairs.size, 0.1) if is_last else np.maximum(1.0, 0.02 * spacing)      # matcher.py:578-603
        Hs, Ws = self._Hs[pairs], self._Ws[pairs]
        vm = v_all[pairs] + U[pairs]                                         # [Q, V, 2]
        xmin = np.maximum(-0.5 + tx[pairs], vm[:, :, 0].min(axis=1)); ymin = np.maximum(-0.5 + ty[pairs], vm[:, :, 1].min(axis=1))
        xmax = np.minimum(Ws - 0.5 + tx[pairs], vm[:, :, 0].max(axis=1)); ymax = np.minimum(Hs - 0.5 + ty[pairs], vm[:, :, 1].max(axis=1))
        ok = (xmax > xmin) & (ymax > ymin)                                    # common.intersect_bbox validity
        nx, ny, dx, dy = _divide_bbox_batch(xmin, ymin, xmax, ymax, spacing, mnb)
        dx = np.where(ok, dx, 1); dy = np.where(ok, dy, 1)
        pf = pad_flags[pairs]
        nfl = self._nfl
        fh = np.where(pf, nfl[np.clip(2 * dy - 1, 0, nfl.size - 1)], nfl[np.clip(dy, 0, nfl.size - 1)])
        fw = np.where(pf, nfl[np.clip(2 * dx - 1, 0, nfl.size - 1)], nfl[np.clip(dx, 0, nfl.size - 1)])
        # one group per block grid, FFT shape, pad flag and block size (the windows of a group are rendered to one size)
        cols = np.stack((nx, ny, fh, fw, pf.astype(np.int64), dx, dy), axis=-1).astype(np.int64)
        cols[~ok] = -1
        dogf = self.d_dogf_view
        img1 = dogf.offset(n * H * W * 4)
        groups = []
        for kv in np.unique(cols, axis=0):
            if kv[0] < 0:
                continue
            gi = np.flatnonzero(np.all(cols == kv, axis=1))
            sel = pairs[gi]
            gfh, gfw, gpad = int(fh[gi[0]]), int(fw[gi[0]]), bool(pf[gi[0]])
            blk, bb = self._blocks(tx, ty, None, sel, spacing[gi], mnb, bounds=(xmin[gi], ymin[gi], xmax[gi], ymax[gi]))
            Q, nblk = blk.shape[:2]
            h, w = int(dy[gi[0]]), int(dx[gi[0]])
            aff = np.zeros((Q, nblk, 10))
            exact = []                                                        # (q, block ids, map_x, map_y, mask)
            # tiers + affine maps of all blocks of the group (C++, no interpreter lock; deformed.block_affines is the
            # same computation in numpy)
            vmg = np.ascontiguousarray(vm[gi])
            bbi = np.ascontiguousarray(bb, dtype=np.int32)
            tiers = np.empty((Q, nblk), dtype=np.int32); A6 = np.empty((Q, nblk, 6)); lo = np.empty((Q, 2))
            gxs = np.ascontiguousarray(self._gx[sel]); gys = np.ascontiguousarray(self._gy[sel])      # every pair's node grid
            tolg = np.ascontiguousarray(tol_q[gi])
            _lib.check(lib.fb_deformed_block_affines(ctx, Q, m.grid_xs.size, m.grid_ys.size, _lib.ptr(gxs), _lib.ptr(gys), 1,
                                                     _lib.ptr(vmg), nblk, _lib.ptr(bbi), 0.0, _lib.ptr(tolg), _lib.ptr(tiers), _lib.ptr(A6),
                                                     _lib.ptr(lo)))
            aff[:, :, 0] = bb[:, :, 0]; aff[:, :, 1] = bb[:, :, 1]
            aff[:, :, 2:8] = A6
            for q in range(Q):
                tier = tiers[q]
                v_init = v_all[sel[q]]
                if (tier < 0).any():                                          # degenerate / flipped fit: statement-by-statement route
                    tier, A, _ = dfm.block_affines(vmg[q], v_init, tris, bb[q], float(tolg[q]))
                    tiers[q] = tier
                    aff[q, :, 2] = A[:, 0, 0]; aff[q, :, 3] = A[:, 1, 0]; aff[q, :, 4] = A[:, 2, 0]
                    aff[q, :, 5] = A[:, 0, 1]; aff[q, :, 6] = A[:, 1, 1]; aff[q, :, 7] = A[:, 2, 1]
                    rows = np.flatnonzero(tier < 3)
                    if rows.size:
                        cx = np.stack((bb[q, rows, 0], bb[q, rows, 2] - 1), axis=-1).astype(np.float64)
                        cy = np.stack((bb[q, rows, 1], bb[q, rows, 3] - 1), axis=-1).astype(np.float64)
                        lo[q, 0] = (cx[:, :, None] * A[rows, None, None, 0, 0] + cy[:, None, :] * A[rows, None, None, 1, 0] + A[rows, None, None, 2, 0]).min()
                        lo[q, 1] = (cx[:, :, None] * A[rows, None, None, 0, 1] + cy[:, None, :] * A[rows, None, None, 1, 1] + A[rows, None, None, 2, 1]).min()
                    else:
                        lo[q] = np.inf
                self.last_tiers[int(sel[q])] = tier.copy()
                # one remap origin for the whole stack of a pair (render_by_subregions, common.py:316-321):
                # floor(min of the rendered maps) - 4; an affine map takes its extremes at the corner pixels (lo)
                lo_x, lo_y = lo[q]
                ex = np.flatnonzero(tier == 3)
                if ex.size:
                    # exact piecewise-linear field of these blocks (C++; deformed.exact_field states it in numpy)
                    po = np.zeros(ex.size, dtype=np.int32)
                    org = np.ascontiguousarray(bb[q, ex, :2], dtype=np.int32)
                    emx = np.empty((ex.size, h, w)); emy = np.empty((ex.size, h, w)); emk8 = np.empty((ex.size, h, w), dtype=np.uint8)
                    vq = np.ascontiguousarray(vmg[q][None])
                    gx1 = np.ascontiguousarray(gxs[q]); gy1 = np.ascontiguousarray(gys[q])
                    _lib.check(lib.fb_deformed_exact_field(ctx, 1, m.grid_xs.size, m.grid_ys.size, _lib.ptr(gx1), _lib.ptr(gy1), 0,
                                                           _lib.ptr(vq), ex.size, _lib.ptr(po), _lib.ptr(org), h, w, _lib.ptr(emx), _lib.ptr(emy),
                                                           _lib.ptr(emk8)))
                    emk = emk8.astype(bool)
                    if emk.any():
                        lo_x = min(lo_x, emx[emk].min()); lo_y = min(lo_y, emy[emk].min())
                    exact.append((q, ex, emx, emy, emk))
                if np.isfinite(lo_x):
                    aff[q, :, 8] = np.floor(lo_x) - 4; aff[q, :, 9] = np.floor(lo_y) - 4
            nb = Q * nblk
            assert nb <= self.max_blocks
            flat = np.ascontiguousarray(blk.reshape(-1, 9))
            affc = np.ascontiguousarray(aff.reshape(-1, 10))
            d_aff = self._scratch('aff', affc.nbytes)
            _lib.check(lib.fb_memcpy_h2d(ctx, self.d_blk.ptr, _lib.ptr(flat), flat.nbytes))
            _lib.check(lib.fb_memcpy_h2d(ctx, d_aff.ptr, _lib.ptr(affc), affc.nbytes))
            _lib.check(lib.fb_ncc_blocks_affine_dev(ctx, dogf.ptr, img1, H, W, H, W, nb, self.d_blk.ptr, d_aff.ptr, h, w, gfh, gfw,
                                                    1 if subpixel else 0, self.conf_mode, self.d_out.ptr, self.d_out.offset(8 * nb),
                                                    self.d_out.offset(16 * nb)))
            ddx, ddy, dcf = (np.array(a).reshape(Q, nblk) for a in self._fetch_out(nb))
            if exact:
                self._exact_blocks(exact, blk, aff, sel, h, w, gpad, subpixel, ddx, ddy, dcf)
            groups.append((sel, bb, ddx, ddy, dcf))
        return groups

    def _exact_blocks(self, exact, blk, aff, sel, h, w, pad, subpixel, ddx, ddy, dcf):
        """the exact-field tier (renderer.py:511-563) of a few blocks: both windows are materialised (fb_remap_dev; the
        image-0 window through an integer map) and correlated as a stack (fb_ncc_batch_dev)."""
        lib, ctx = _lib.load(), _lib.ctx()
        n, H, W = self.P, self.H, self.W
        dogf = self.d_dogf_view                                              # [2 n][H][W]: image 1 of pair p is image n + p
        where = [(q, b, k, rec) for rec in exact for q in (rec[0],) for k, b in enumerate(rec[1])]
        nb = len(where)
        N2 = 2 * nb
        jj, ii = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing='ij')
        ids = np.empty(N2, dtype=np.int32); org = np.empty((N2, 2), dtype=np.int32)
        mxa = np.empty((N2, h, w), dtype=np.float32); mya = np.empty((N2, h, w), dtype=np.float32); mka = np.ones((N2, h, w), dtype=np.uint8)
        for e, (q, b, k, rec) in enumerate(where):
            # first half of the stack: image-0 windows, integer crop at blk's (x0, y0); second half: image-1 windows
            # through the exact map, relative to the pair's remap origin
            ids[e] = int(sel[q]); org[e] = (int(blk[q, b, 1]), int(blk[q, b, 2])); mxa[e] = ii; mya[e] = jj
            ids[nb + e] = n + int(sel[q]); org[nb + e] = (int(aff[q, b, 8]), int(aff[q, b, 9]))
            mxa[nb + e] = rec[2][k] - aff[q, b, 8]; mya[nb + e] = rec[3][k] - aff[q, b, 9]; mka[nb + e] = rec[4][k]
        px = h * w
        bufs = {k: self._scratch(k, sz) for k, sz in (('ids', 4 * N2), ('org', 8 * N2), ('mx', 4 * N2 * px), ('my', 4 * N2 * px),
                                                      ('mk', N2 * px), ('st', 4 * N2 * px), ('res', 20 * nb))}
        for k, a_ in (('ids', ids), ('org', org), ('mx', mxa), ('my', mya), ('mk', mka)):
            _lib.check(lib.fb_memcpy_h2d(ctx, bufs[k].ptr, _lib.ptr(a_), a_.nbytes))
        _lib.check(lib.fb_remap_dev(ctx, dogf.ptr, H, W, N2, bufs['ids'].ptr, h, w, bufs['mx'].ptr, bufs['my'].ptr, bufs['mk'].ptr,
                                    bufs['org'].ptr, bufs['st'].ptr))
        r = bufs['res']
        _lib.check(lib.fb_ncc_batch_dev(ctx, bufs['st'].ptr, bufs['st'].offset(4 * nb * px), nb, 1, h, w, h, w, 1 if pad else 0,
                                        1 if subpixel else 0, self.conf_mode, r.ptr, r.offset(8 * nb), r.offset(16 * nb)))
        raw = r.to_array((20 * nb,), np.uint8)
        ex_dx, ex_dy, ex_cf = raw[:8 * nb].view(np.float64), raw[8 * nb:16 * nb].view(np.float64), raw[16 * nb:].view(np.float32)
        for e, (q, b, _, _) in enumerate(where):
            ddx[q, b], ddy[q, b], dcf[q, b] = ex_dx[e], ex_dy[e], ex_cf[e]

    def _relax_general(self, pid, xy0_mov, nodes3, B1, conf, resolve):
        """matcher.py:725-741 for matches given by their mesh1 triangle (nodes3: vertex ids inside the union mesh) and
        barycentric coordinates: optimize_linear as the TOTAL displacement of mesh1 from its FIXED gear
        (fb_pairs_relax_bary), relax_higly_deformed + huber residue weights, and -- `resolve`, the rounds before the last --
        a second solve for the pairs whose weights changed (matcher.py:737-741).
        Returns rw [K] and the node displacement x [P, V, 2] (rows of pairs without matches are zero)."""
        lib, ctx = _lib.load(), _lib.ctx()
        sysh = self._relax_system()
        m = self._mesh
        V, K = m.num_vertices, pid.size
        nodes3 = np.ascontiguousarray(nodes3, dtype=np.int32)
        B1 = np.ascontiguousarray(B1, dtype=np.float64)
        xy0c = np.ascontiguousarray(xy0_mov, dtype=np.float64)
        xy1_fixed = np.sum(self._v_init_u[nodes3] * B1[:, :, None], axis=1)      # mesh1 at its FIXED gear (= INITIAL, no offset)
        dxy0 = np.ascontiguousarray(xy1_fixed - xy0c)
        area = float(np.abs(m.triangle_areas(const.MESH_GEAR_INITIAL)[0]))
        sample_err = 0.4387 * area ** 0.5 * DEFAULT_AVG_DEFORM
        se_each = None
        if self._ragged:                                      # triangle areas differ from pair to pair
            se_each = np.ascontiguousarray(self._sample_err_each[np.asarray(pid)])
        tid_local = (self._tri_of_nodes(nodes3 - (np.asarray(pid)[:, None] * V))).astype(np.int32)
        pid32 = np.ascontiguousarray(pid, dtype=np.int32)
        zero_t = np.zeros((self.P, 2))

        def solve(w32):
            rw = np.empty(K, dtype=np.float32)
            x = np.empty(2 * self.P * V, dtype=np.float64)
            iters, relres = C.c_int(), C.c_double()
            _lib.check(lib.fb_pairs_relax_bary(ctx, sysh, self.P, K, _lib.ptr(nodes3), _lib.ptr(B1), _lib.ptr(dxy0), _lib.ptr(w32),
                                               self.residue_len if self.residue_len > 0 else 1.0, self.residue_mode, sample_err, _lib.ptr(se_each), self.stiffness_lambda,
                                               self.relax_tol, _lib.ptr(rw), _lib.ptr(x), C.byref(iters), C.byref(relres)))
            self.last_relax = dict(iters=iters.value, relres=relres.value, matches=int(K), relaxed_first=0)
            return rw, x.reshape(self.P, V, 2)

        w32 = np.ascontiguousarray(conf, dtype=np.float32)
        rw, x = solve(w32)
        self._links_rows = None
        if self.residue_len <= 0:
            return np.ones(K, dtype=np.float32), x
        self._relax_first(x, pid32, xy0c, None, zero_t, rw, sample_err if se_each is None else se_each, tid_B=(tid_local, B1))
        if resolve and np.any(rw != 1):
            changed = np.unique(pid32[rw != 1])
            _, x2 = solve(np.ascontiguousarray(w32 * rw))
            x[changed] = x2[changed]
        return rw, x

    def _tri_of_nodes(self, local3):
        """triangle id of a vertex triple given in the order of Mesh.triangles: cell (a b / c d) -> (a, b, d), (a, d, c)"""
        nx = self._mesh.grid_xs.size
        l3 = np.asarray(local3, dtype=np.int64)
        a = l3[:, 0]
        j, i = a // nx, a % nx
        return 2 * (j * (nx - 1) + i) + (l3[:, 1] != a + 1)

    # ------------------------------------------------------------------ last-round relaxation
    def _relax_system(self):
        """P copies of the cartesian mesh of matcher.py:354-356 as ONE block-diagonal GPU system.  mesh0 is
        locked (matcher.py:361), so a match couples only the three vertices of its mesh1 triangle: the symbolic
        pattern and the stiffness K never change between batches; only the links do."""
        if self._relax_sys is not None:
            return self._relax_sys
        lib, ctx = _lib.load(), _lib.ctx()
        m = Mesh.from_bbox((0, 0, self.W, self.H), cartesian=True, mesh_size=float(np.min(self.spacings)),
                           min_num_blocks=self.mnb, uid=1)
        self._mesh = m
        V, T, P = m.num_vertices, m.num_triangles, self.P
        self._sys_key = (P, m.grid_xs.size, m.grid_ys.size)
        sysh = self._pool.systems.pop(self._sys_key, None) if self._pool is not None else None
        if sysh is None:
            sysh = C.c_void_p()
            _lib.check(lib.fb_sys_create(ctx, P * V, C.byref(sysh)))
            # the P copies enter as ONE mesh (one assembly launch per stiffness state)
            tri_u = np.ascontiguousarray((m.triangles[None, :, :] + (np.arange(P) * V)[:, None, None]).reshape(-1, 3), dtype=np.int32)
            mid = C.c_int()
            _lib.check(lib.fb_sys_add_mesh(ctx, sysh, 0, _lib.ptr(tri_u), P * V, P * T, C.byref(mid)))
            _lib.check(lib.fb_sys_set_links(ctx, sysh, 0, None))
            nnzb = C.c_int64()
            _lib.check(lib.fb_sys_finalize(ctx, sysh, C.byref(nnzb)))
        # (a system of the same topology from the pool keeps its symbolic pattern; the stiffness is re-assembled below)
        self._mult_u = np.ascontiguousarray(np.tile(m.element_multiplier(), P), dtype=np.float32)
        self._v_init_u = np.ascontiguousarray(np.tile(m.vertices(const.MESH_GEAR_INITIAL), (P, 1)), dtype=np.float64)
        self._relax_sys = sysh
        self._k_state = None
        self._assemble_union(self._v_init_u, 'initial')       # translation invariant: shape = INITIAL vertices, no stress
        v0 = self._v_init_u.reshape(P, V, 2)
        v0 = np.ascontiguousarray((v0 - v0.mean(axis=1, keepdims=True)).reshape(-1, 2))
        es0 = np.empty(P)
        _lib.check(lib.fb_sys_group_energy(ctx, sysh, P, _lib.ptr(v0), _lib.ptr(es0)))
        self._es0 = float(es0[0])                             # v0^T K v0 of the centred mesh: the same for every copy and rotation
        self._gx = np.tile(m.grid_xs, (P, 1)); self._gy = np.tile(m.grid_ys, (P, 1))      # node coordinates of every pair's grid
        return sysh

    def _pair_mesh(self, p):
        """a fresh Mesh object of pair p's mesh1 (host statements that work pair by pair)"""
        m = self._mesh.copy(uid=1)
        m.grid_xs, m.grid_ys = self._mesh.grid_xs, self._mesh.grid_ys
        return m

    def _assemble_union(self, v_shape, state):
        """stiffness K of the P mesh copies at the given shapes ([P V, 2]); skipped when K already holds `state`"""
        if state is not None and state == self._k_state:
            return
        m = self._mesh
        _lib.check(_lib.load().fb_sys_assemble_mesh(_lib.ctx(), self._relax_sys, 0, _lib.ptr(v_shape), None, _lib.ptr(self._mult_u),
                                                    m.poisson_ratio, 1.0))
        self._k_state = state

    def _final_relax(self, pid, xy0, xy1, wt, t1):
        """matcher.py:725-737 for every pair of the batch at once (fb_pairs_relax): relax mesh1 against the last-round
        links (optimize_linear, to the fixed point), then huber residue weights (optimizer.py:174-191, 203-205).
        pid [K]; xy0/xy1 [K, 2] in the MOVING gear; wt [K] confidences; t1 [P, 2] mesh1 offsets.
        Returns the residue weight [K] float32 and the displacement of every mesh1 vertex [P, V, 2]."""
        lib, ctx = _lib.load(), _lib.ctx()
        sysh = self._relax_system()
        m = self._mesh
        V, K = m.num_vertices, pid.size
        pid32 = np.ascontiguousarray(pid, dtype=np.int32)
        xy0c = np.ascontiguousarray(xy0, dtype=np.float64)
        xy1i = np.ascontiguousarray(xy1 - t1[pid], dtype=np.float64)             # mesh1 coordinates without its offset
        t1c = np.ascontiguousarray(t1, dtype=np.float64)
        w32 = np.ascontiguousarray(wt, dtype=np.float32)
        rw = np.empty(K, dtype=np.float32)
        x = np.empty(2 * self.P * V, dtype=np.float64)
        iters, relres = C.c_int(), C.c_double()
        area = float(np.abs(m.triangle_areas(const.MESH_GEAR_INITIAL)[0]))
        sample_err = 0.4387 * area ** 0.5 * DEFAULT_AVG_DEFORM          # optimizer.py:26-30, equal triangles on both sides
        _lib.check(lib.fb_pairs_relax(ctx, sysh, self.P, m.grid_xs.size, m.grid_ys.size, _lib.ptr(m.grid_xs), _lib.ptr(m.grid_ys), K,
                                      _lib.ptr(pid32), _lib.ptr(xy0c), _lib.ptr(xy1i), _lib.ptr(t1c), _lib.ptr(w32), self.residue_len, self.residue_mode,
                                      sample_err, self.stiffness_lambda, self.relax_tol, _lib.ptr(rw), _lib.ptr(x),
                                      C.byref(iters), C.byref(relres)))
        self.last_relax = dict(iters=iters.value, relres=relres.value, matches=int(K), relaxed_first=0)
        self._links_rows = (K, pid32, xy1i)                   # what the links resident in the system were built from
        x = x.reshape(self.P, V, 2)
        self._relax_first(x, pid32, xy0c, xy1i, t1c, rw, sample_err)
        return rw, x

    def _relax_first(self, x, pid, xy0, xy1i, t1, rw, sample_err, tid_B=None):
        """adjust_link_weight_by_residue(relax_first=True) (matcher.py:736 -> optimizer.py:763-779): before the residues
        are taken, a region of mesh1 that the relaxation deformed beyond the cutoff is relaxed on its own
        (relax_mesh_most_deformed).  Screen: with d = the largest displacement difference along a grid edge relative to
        that edge, every triangle's area and edge deformation is below 2 d, and nothing is freed below
        1 - 1 / (1 + (1 - 1 / 1.35)) = 0.206 -- pairs with d <= 0.1 are done.  The others (gross mismatches only) take the
        reference's statements one pair at a time: optimizer.relax_mesh_most_deformed (device assembly + PCG), then the
        residue weights of that pair from the relaxed mesh."""
        m = self._mesh
        nx, ny = m.grid_xs.size, m.grid_ys.size
        g = x.reshape(self.P, ny, nx, 2)
        ex = np.sqrt(np.sum(np.diff(g, axis=2) ** 2, axis=-1)) / np.diff(self._gx, axis=1)[:, None, :]
        ey = np.sqrt(np.sum(np.diff(g, axis=1) ** 2, axis=-1)) / np.diff(self._gy, axis=1)[:, :, None]
        d = np.maximum(ex.max(axis=(1, 2)), ey.max(axis=(1, 2)))
        suspects = np.flatnonzero(d > 0.1)
        if suspects.size == 0:
            return
        from .optimizer import relax_mesh_most_deformed
        gear = (const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)
        cutoff = 1 - 1 / (const.MAXIMUM_DEFORM_ALLOWED + 1)      # SLM.relax_higly_deformed hands the converted value down
        se_each = np.broadcast_to(np.asarray(sample_err, dtype=np.float64), (pid.size,))
        for p in suspects:
            rows = np.flatnonzero(pid == p)
            if rows.size == 0:
                continue
            m1 = self._pair_mesh(p)
            v_init = m1.vertices(const.MESH_GEAR_INITIAL)
            m1.set_field(t1[p] + x[p], gear=gear)
            if not relax_mesh_most_deformed(m1, gear=gear, deform_cutoff=cutoff):
                continue
            self.last_relax['relaxed_first'] += 1
            if tid_B is None:
                pts = xy1i[rows]
                tid = m1.locate_cartesian(pts)
                _, B = m1.cart2bary(pts, const.MESH_GEAR_INITIAL, tid=tid)
            else:
                tid, B = tid_B[0][rows], tid_B[1][rows]
            dxy = m1.bary2cart(tid, B, const.MESH_GEAR_MOVING, offsetting=True) - xy0[rows]
            dis = np.sum(dxy ** 2, axis=-1) ** 0.5
            dis = ((dis ** 2 - se_each[rows] ** 2).clip(0, None)) ** 0.5              # optimizer.py:183-185
            rw[rows] = ((dis <= self.residue_len) if self.residue_mode == 1 else (self.residue_len / np.maximum(dis, self.residue_len))).astype(np.float32)
            x[p] = m1.vertices_w_offset(const.MESH_GEAR_MOVING) - (v_init + t1[p])

    def _rigid_fits(self, pid, p0, p1, wt):
        """spatial.fit_affine(p0, p1, return_rigid=True, weight, svd_clip=(1, 1)) for every pair at once.  One pass
        accumulates the raw (weighted and unweighted) moments of every pair; centring by the unweighted means and the
        common scale max(std0, std1) (spatial.py:24-33) are applied to the P small moment matrices, then the weighted
        least squares is the 3x3 normal-equation solve of each pair and the rigid part the polar rotation of its 2x2
        block.  Pairs that are rank deficient, reflected or have fewer than 3 matches go through the
        statement-by-statement host function.  The rows of a pair must be contiguous."""
        from .common import fit_affine
        P, K = self.P, pid.size
        starts = np.concatenate(([0], np.flatnonzero(np.diff(pid) != 0) + 1))
        present = pid[starts]
        w = np.asarray(wt, dtype=np.float64)
        x0, y0, x1, y1 = p0[:, 0], p0[:, 1], p1[:, 0], p1[:, 1]
        F = np.empty((K, 21))
        F[:, 0] = 1.0; F[:, 1] = x0; F[:, 2] = y0; F[:, 3] = x1; F[:, 4] = y1
        F[:, 5] = x0 * x0; F[:, 6] = y0 * y0; F[:, 7] = x1 * x1; F[:, 8] = y1 * y1
        F[:, 9] = w; F[:, 10] = w * x1; F[:, 11] = w * y1; F[:, 12] = w * x0; F[:, 13] = w * y0
        F[:, 14] = F[:, 10] * x1; F[:, 15] = F[:, 10] * y1; F[:, 16] = F[:, 11] * y1
        F[:, 17] = F[:, 10] * x0; F[:, 18] = F[:, 10] * y0; F[:, 19] = F[:, 11] * x0; F[:, 20] = F[:, 11] * y0
        S = np.zeros((P, 21))
        S[present] = np.add.reduceat(F, starts, axis=0)
        cnt = S[:, 0]
        n = np.maximum(cnt, 1.0)
        m0x, m0y, m1x, m1y = S[:, 1] / n, S[:, 2] / n, S[:, 3] / n, S[:, 4] / n
        var0 = (S[:, 5] / n - m0x ** 2) + (S[:, 6] / n - m0y ** 2)
        var1 = (S[:, 7] / n - m1x ** 2) + (S[:, 8] / n - m1y ** 2)
        scl = np.sqrt(np.maximum(np.maximum(var0, var1), 0.0))
        scl = np.where(scl < 1e-6, 1.0, scl)
        sw, sx1, sy1, sx0, sy0 = S[:, 9], S[:, 10], S[:, 11], S[:, 12], S[:, 13]
        cx1, cy1, cx0, cy0 = sx1 - m1x * sw, sy1 - m1y * sw, sx0 - m0x * sw, sy0 - m0y * sw     # sums of w (p - mean)
        G = np.empty((P, 3, 3)); Hm = np.empty((P, 3, 3))
        s2 = scl * scl
        G[:, 0, 0] = (S[:, 14] - 2 * m1x * sx1 + m1x ** 2 * sw) / s2
        G[:, 0, 1] = G[:, 1, 0] = (S[:, 15] - m1x * sy1 - m1y * sx1 + m1x * m1y * sw) / s2
        G[:, 1, 1] = (S[:, 16] - 2 * m1y * sy1 + m1y ** 2 * sw) / s2
        G[:, 0, 2] = G[:, 2, 0] = cx1 / scl
        G[:, 1, 2] = G[:, 2, 1] = cy1 / scl
        G[:, 2, 2] = sw
        Hm[:, 0, 0] = (S[:, 17] - m1x * sx0 - m0x * sx1 + m1x * m0x * sw) / s2
        Hm[:, 0, 1] = (S[:, 18] - m1x * sy0 - m0y * sx1 + m1x * m0y * sw) / s2
        Hm[:, 1, 0] = (S[:, 19] - m1y * sx0 - m0x * sy1 + m1y * m0x * sw) / s2
        Hm[:, 1, 1] = (S[:, 20] - m1y * sy0 - m0y * sy1 + m1y * m0y * sw) / s2
        Hm[:, 0, 2] = cx1 / scl; Hm[:, 1, 2] = cy1 / scl
        Hm[:, 2, 0] = cx0 / scl; Hm[:, 2, 1] = cy0 / scl; Hm[:, 2, 2] = sw
        ok = cnt >= 3
        ev = np.linalg.eigvalsh(np.where(ok[:, None, None], G, np.eye(3)))
        ok &= ev[:, 0] > 1e-9 * ev[:, 2]
        A = np.tile(np.eye(3), (P, 1, 1))
        A[ok] = np.linalg.solve(G[ok], Hm[ok])
        ok &= np.linalg.det(A) > 0
        u, sv, vh = np.linalg.svd(A[:, :2, :2])
        R = A.copy()
        R[:, :2, :2] = u @ vh                                  # singular values clipped to (1, 1)
        mm0 = np.stack((m0x, m0y), axis=-1); mm1 = np.stack((m1x, m1y), axis=-1)
        R[:, 2, :2] = A[:, 2, :2] + mm0 - np.einsum('pi,pij->pj', mm1, R[:, :2, :2])
        R[:, :, 2] = np.array([0.0, 0.0, 1.0])
        for p in np.flatnonzero(~ok & (cnt > 0)):
            s = pid == p
            R[p] = fit_affine(p0[s], p1[s], return_rigid=True, weight=wt[s], svd_clip=(1, 1), avoid_flip=True)[1]
        return R

    def _strain(self, pid, xy0, xy1, wt, txy, reuse_links=False):
        """matcher.py:752-777 for every pair of the batch: fresh mesh pair, rigid initialisation of mesh1 from the
        final matches (optimize_affine_cascade, svd_clip (1, 1); host, `_rigid_fits`), anneal, optimize_linear(tol=1e-6),
        strain = sqrt(Es / Es0) with the stiffness of mesh1 at its rigidly placed shape (fb_pairs_strain).
        pid/xy0/xy1/wt: the final match table (INITIAL gears), rows of a pair contiguous; txy [P, 2] translation of the
        locked mesh0.  Returns strain [P] (DEFAULT_AVG_DEFORM where a pair has no match).  reuse_links: the table is
        row for row the one the preceding `_final_relax` call was given."""
        lib, ctx = _lib.load(), _lib.ctx()
        sysh = self._relax_system()
        m = self._mesh
        P = self.P
        if pid.size == 0:
            return np.full(P, DEFAULT_AVG_DEFORM)
        nseg = 1 + int(np.count_nonzero(np.diff(pid) != 0))
        if nseg != int(np.count_nonzero(np.bincount(pid, minlength=P))):          # rows of a pair not contiguous
            o = np.argsort(pid, kind='stable')
            pid, xy0, xy1, wt = pid[o], xy0[o], xy1[o], wt[o]
            reuse_links = False
        p0 = np.ascontiguousarray(xy0 + txy[pid], dtype=np.float64)              # mesh0 points, FIXED gear
        xy1c = np.ascontiguousarray(xy1, dtype=np.float64)
        w32 = np.ascontiguousarray(wt, dtype=np.float32)
        R = np.ascontiguousarray(self._rigid_fits(pid, p0, xy1c, w32))
        pid32 = np.ascontiguousarray(pid, dtype=np.int32)
        strain = np.empty(P)
        iters, relres = C.c_int(), C.c_double()
        _lib.check(lib.fb_pairs_strain(ctx, sysh, P, m.grid_xs.size, m.grid_ys.size, _lib.ptr(m.grid_xs), _lib.ptr(m.grid_ys), pid.size,
                                       _lib.ptr(pid32), _lib.ptr(p0), _lib.ptr(xy1c), _lib.ptr(w32), _lib.ptr(R), self.stiffness_lambda,
                                       self._es0, 1 if reuse_links else 0, DEFAULT_AVG_DEFORM, _lib.ptr(strain), C.byref(iters), C.byref(relres)))
        self.last_strain_solve = dict(iters=iters.value, relres=relres.value, matches=int(pid.size))
        return strain

    # ------------------------------------------------------------------ driver
    def _rows_bary(self, pid, xy1_init, nodes3, B1):
        """mesh1 triangle (vertex ids inside the union mesh) and barycentric coordinates of every row; rows of pairs whose
        mesh1 is still a translated grid (nodes3 < 0) are located on the grid (cart2bary, mesh.py:2191-2217)."""
        self._relax_system()
        m = self._mesh
        V = m.num_vertices
        nodes3 = np.array(nodes3, dtype=np.int64); B1 = np.array(B1, dtype=np.float64)
        g = np.flatnonzero(nodes3[:, 0] < 0)
        if g.size:
            tid = m.locate_cartesian(xy1_init[g])
            _, Bg = m.cart2bary(xy1_init[g], const.MESH_GEAR_INITIAL, tid=tid)
            nodes3[g] = m.triangles[tid] + (np.asarray(pid)[g] * V)[:, None]
            B1[g] = Bg
        return nodes3, B1

    def match(self, strips0, strips1, masks0=None, masks1=None, compute_photometric=False):
        """stitching_matcher for the P resident pairs; see `_match_host` for the arguments and the result.  Batches go through
        fb_match_strips (one C entry for the whole sequence, masks and the deformed-mesh branch included); the pairs it hands back
        (flags != 0: relax_first, folded block of a deformed mesh, degenerate rigid fit) and photometric batches of unequal strips
        take the numpy statement of the same sequence."""
        if self._route == 'native' and (not self._ragged or not compute_photometric):
            if self._prefer_host:
                # the entry handed back most pairs of the last batch: this one takes the host statement directly, the next
                # one tries the entry again
                self._prefer_host = False
                return self._match_host(strips0, strips1, masks0, masks1, compute_photometric)
            return self._match_native(strips0, strips1, masks0, masks1, compute_photometric)
        return self._match_host(strips0, strips1, masks0, masks1, compute_photometric)

    def _native_matcher(self):
        if self._native is None:
            lib, ctx = _lib.load(), _lib.ctx()
            auto = self._ragged and self._opts.get('spacings') is None          # automatic spacings per strip shape
            sp = np.ascontiguousarray(self.spacings, dtype=np.float64)
            o = _lib.StripOpts(self.sigma, 1 if self.cds == 0.5 else 0, self.conf_thresh, self.mnb, self.conf_mode, self.residue_len,
                               self.residue_mode, self.stiffness_lambda, self.relax_tol, int(self.compute_strain), 0 if auto else int(sp.size),
                               None if auto else sp.ctypes.data)
            h = C.c_void_p()
            if self._ragged:
                shapes = np.ascontiguousarray(np.stack((self._Hs, self._Ws), axis=-1), dtype=np.int32)
                _lib.check(lib.fb_strip_matcher_create_ragged(ctx, self.P, self.H, self.W, _lib.ptr(shapes), C.byref(o), C.byref(h)))
            else:
                _lib.check(lib.fb_strip_matcher_create(ctx, self.P, self.H, self.W, C.byref(o), C.byref(h)))
            self._native = h
        return self._native

    @staticmethod
    def _effective_masks(masks0, masks1, n):
        """(masks0, masks1) as lists of uint8 arrays / None, or None when no mask has a zero (common.py:368: such a mask
        changes nothing)"""
        if masks0 is None and masks1 is None:
            return None
        out = []
        for ml in (masks0, masks1):
            ml = [None] * n if ml is None else list(ml)
            out.append([None if (mk is None or np.all(mk)) else np.ascontiguousarray(np.asarray(mk) != 0, dtype=np.uint8) for mk in ml])
        if all(mk is None for ml in out for mk in ml):
            return None
        return tuple(out)

    def _match_native(self, strips0, strips1, masks0=None, masks1=None, compute_photometric=False):
        lib, ctx = _lib.load(), _lib.ctx()
        n = self.P
        strips0 = strips0.value if hasattr(strips0, 'value') else strips0
        strips1 = strips1.value if hasattr(strips1, 'value') else strips1
        h = self._native_matcher()
        masks = self._effective_masks(masks0, masks1, n)
        arrs = [None, None]
        if masks is not None:
            for side in (0, 1):
                for p, mk in enumerate(masks[side]):
                    # (a ragged batch: the mask of pair p has the pair's own shape, one contiguous array)
                    want = (int(self._Hs[p]), int(self._Ws[p])) if self._ragged else (self.H, self.W)
                    if mk is not None and mk.shape != want:
                        raise ValueError(f'the mask of pair {p} must have the shape of its strip, {want[0]} x {want[1]}')
                arrs[side] = (C.c_void_p * n)(*[None if mk is None else mk.ctypes.data for mk in masks[side]])
        # always stated: the extras belong to the call that follows, and a call that never happened (an exception in between)
        # must not leave its mask pointers behind
        _lib.check(lib.fb_strip_matcher_set_extras(ctx, h, arrs[0], arrs[1], 1 if compute_photometric else 0))
        tx = np.empty(n); ty = np.empty(n); cf0 = np.empty(n, dtype=np.float32); strain = np.empty(n)
        valid = np.empty(n, dtype=np.uint8); flags = np.empty(n, dtype=np.uint8)
        nrows = C.c_int64()
        _lib.check(lib.fb_match_strips(ctx, h, C.c_void_p(strips0), C.c_void_p(strips1), _lib.ptr(tx), _lib.ptr(ty), _lib.ptr(cf0), _lib.ptr(valid),
                                       _lib.ptr(flags), _lib.ptr(strain), C.byref(nrows)))
        K = nrows.value
        pid = np.empty(K, dtype=np.int32); xy0 = np.empty((K, 2)); xy1 = np.empty((K, 2)); wt = np.empty(K, dtype=np.float32)
        _lib.check(lib.fb_match_strips_table(ctx, h, _lib.ptr(pid), _lib.ptr(xy0), _lib.ptr(xy1), _lib.ptr(wt)))
        it_r, it_s, rr_r, rr_s = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        _lib.check(lib.fb_strip_matcher_info(ctx, h, None, None, None, None, C.byref(it_r), C.byref(rr_r), C.byref(it_s), C.byref(rr_s)))
        self.last_relax = dict(iters=it_r.value, relres=rr_r.value, matches=int(K), relaxed_first=0)
        self.last_strain_solve = dict(iters=it_s.value, relres=rr_s.value, matches=int(K))
        self.last_flags = flags
        self.last_field = None
        self.last_tiers = {}
        zeros = np.zeros(n, dtype=bool)
        # pairs whose mesh1 was relaxed into a non-rigid field between two spacings (matcher.py:725-742): node field and tiers
        deformed = np.zeros(n, dtype=np.uint8); ntier = np.zeros(n, dtype=np.int32); V = C.c_int()
        _lib.check(lib.fb_match_strips_deformed(ctx, h, _lib.ptr(deformed), _lib.ptr(ntier), C.byref(V)))
        if deformed.any():
            field = np.empty((n, V.value, 2)); tiers = np.empty(int(ntier.sum()), dtype=np.int32)
            _lib.check(lib.fb_match_strips_field(ctx, h, _lib.ptr(field), _lib.ptr(tiers)))
            self.last_field = field
            at = np.concatenate(([0], np.cumsum(ntier)))
            self.last_tiers = {p: tiers[at[p]:at[p + 1]].copy() for p in range(n) if ntier[p]}
        phtm = None
        if compute_photometric:
            ph = np.empty((n, 4)); has = np.empty(n, dtype=np.uint8)
            _lib.check(lib.fb_match_strips_photometric(ctx, h, _lib.ptr(ph), _lib.ptr(has)))
            phtm = [tuple(ph[p]) if has[p] else None for p in range(n)]
        res = dict(tx=tx, ty=ty, conf0=cf0, valid=valid.astype(bool), needs_host=zeros, deformed=deformed.astype(bool), deferred=zeros.copy(),
                   pair=pid.astype(np.int64), xy0=xy0, xy1=xy1, weight=wt, strain=strain, phtm=phtm)
        fl = np.flatnonzero(flags)
        if fl.size:
            self._general_route(res, fl, strips0, strips1, masks, compute_photometric)
        return res

    def _general_route(self, res, fl, strips0, strips1, masks=None, compute_photometric=False):
        """the pairs `fl` of the batch through the host route; their results replace those of fb_match_strips in `res`"""
        lib, ctx = _lib.load(), _lib.ctx()
        n, H, W = self.P, self.H, self.W
        self._prefer_host = fl.size > n // 2
        m0, m1 = (None, None) if masks is None else masks
        if fl.size == n:
            res.update(self._match_host(strips0, strips1, m0, m1, compute_photometric))
            return
        sub = self._general.pop(fl.size, None) if not self._ragged else None      # a ragged sub-batch has its own shapes
        if sub is None:
            if len(self._general) >= 4:                          # sub-matchers are kept for the sizes that came last
                self._general.pop(next(iter(self._general))).free()
            sub = self._sub_matcher(fl)
        self._general[fl.size] = sub
        need = fl.size * H * W
        if self._gather is None or self._gather[0].nbytes < need:
            if self._gather is not None:
                for b in self._gather:
                    self._pool.give(b) if self._pool is not None else b.free()
            take = self._pool.take if self._pool is not None else _lib.DeviceBuffer
            self._gather = (take(need), take(need))
        # gather the strips of the flagged pairs, runs of consecutive pairs in one copy
        runs = np.split(fl, np.flatnonzero(np.diff(fl) != 1) + 1)
        at = 0
        for r in runs:
            for buf, src in zip(self._gather, (strips0, strips1)):
                _lib.check(lib.fb_memcpy_d2d(ctx, buf.offset(at * H * W), C.c_void_p(src + int(r[0]) * H * W), r.size * H * W))
            at += r.size
        g = sub._match_host(self._gather[0].ptr, self._gather[1].ptr, None if m0 is None else [m0[p] for p in fl], None if m1 is None else [m1[p] for p in fl],
                            compute_photometric)
        if compute_photometric:
            for k, p in enumerate(fl):
                res['phtm'][p] = g['phtm'][k]
        for k in ('tx', 'ty', 'conf0', 'valid', 'deformed', 'deferred', 'strain'):
            res[k][fl] = g[k]
        for k, src in (('pair', fl[g['pair']]), ('xy0', g['xy0']), ('xy1', g['xy1']), ('weight', g['weight'])):
            res[k] = np.concatenate((res[k], src), axis=0)
        if sub.last_field is not None:
            if self.last_field is None:
                self.last_field = np.zeros((n,) + sub.last_field.shape[1:])
            self.last_field[fl] = sub.last_field
        elif self.last_field is not None:
            self.last_field[fl] = 0.0
        for p in fl:
            self.last_tiers.pop(int(p), None)
        self.last_tiers.update({int(fl[k]): v for k, v in sub.last_tiers.items()})
        if sub.last_relax is not None and self.last_relax is not None:
            self.last_relax['relaxed_first'] += sub.last_relax.get('relaxed_first', 0)

    def _sub_matcher(self, fl):
        """a host-route matcher for the pairs `fl` of this batch (strips gathered into slots of this matcher's size)"""
        return StripBatchMatcher(fl.size, self.H, self.W, pool=self._pool, route='host', **self._opts)

    def _match_host(self, strips0, strips1, masks0=None, masks1=None, compute_photometric=False):
        """strips0/strips1: device pointers to uint8 [P][H][W].  masks0/masks1: optional lists of P host arrays (H x W,
        non-zero = valid pixel) or None entries -- the masked DoG of matcher.py:257-274, 336-337.  compute_photometric:
        result['phtm'] = per pair (av0, av1, std0, std1) of matcher.py:279-314.  Returns a dict of arrays:
        tx, ty, conf0, valid, needs_host, deformed [P]; the match table as flat arrays pair, xy0, xy1, weight
        (rows of one pair are contiguous, pairs in ascending order within a block-grid group).  `needs_host` is kept
        for callers of earlier versions and is always False: pairs whose mesh relaxation between spacings is not a rigid
        translation are `deformed` and stay on the device path."""
        n = self.P
        strips0 = strips0.value if hasattr(strips0, 'value') else strips0
        strips1 = strips1.value if hasattr(strips1, 'value') else strips1
        masks = None
        if masks0 is not None or masks1 is not None:
            masks = (list(masks0) if masks0 is not None else [None] * n, list(masks1) if masks1 is not None else [None] * n)
            # a mask without a zero changes nothing (common.py:368)
            masks = tuple([None if (mk is None or np.all(mk)) else mk for mk in ml] for ml in masks)
            if all(mk is None for ml in masks for mk in ml):
                masks = None
        tx, ty, cf0 = self._global(strips0, strips1, masks, need_small=compute_photometric)
        phtm = self._photometric(strips0, strips1, tx, ty, masks) if compute_photometric else None
        scale = 1.0 / self.cds
        tx = tx * scale; ty = ty * scale                     # matcher.py:338-339
        active = cf0 >= self.conf_thresh                     # matcher.py:277-278
        self._fine_dog(strips0, strips1, masks)
        spacings = self.spacings
        pad = np.ones(n, dtype=bool)
        has_last = np.zeros(n, dtype=bool)
        table = None
        last_links = None
        txy = np.stack((tx, ty), axis=-1)
        t1 = np.zeros((n, 2))                                # translation of mesh1 acquired by rigid relaxations
        U = None                                             # [n, V, 2] node displacement of a deformed mesh1 (MOVING - INITIAL)
        is_deformed = np.zeros(n, dtype=bool)
        deferred = np.zeros(n, dtype=bool)
        self.last_tiers = {}
        live = active.copy()                                 # pairs still iterating over the spacings
        for rnd in range(spacings.size):
            sp = spacings[rnd]
            is_last = rnd == spacings.size - 1
            mnb = self.mnb if is_last else 1
            rows = []
            to_relax = np.zeros(n, dtype=bool)               # pairs whose relaxation of this round is not a rigid translation
            groups = [(g, False) for g in self._match_round(tx, ty, t1, live & ~is_deformed, self._sp[:, rnd], mnb, pad, subpixel=is_last)]
            dp = np.flatnonzero(live & is_deformed)
            if dp.size:
                groups += [(g, True) for g in self._match_round_deformed(tx, ty, U, dp, self._sp[:, rnd], mnb, pad, is_last, is_last)]
                m = self._mesh
                V = m.num_vertices
                v_all = self._v_init_u.reshape(n, V, 2)
            for (sel, bb, ddx, ddy, dcf), is_def in groups:
                keep = dcf > self.conf_thresh                # matcher.py:671-683
                anyk = keep.any(axis=1)
                ctr = 0.5 * np.stack((bb[..., 0] + bb[..., 2], bb[..., 1] + bb[..., 3]), axis=-1) - 0.5      # bbox_centers
                dxy = np.stack((ddx, ddy), axis=-1)
                xy0 = ctr - dxy * 0.5                        # equal block sizes: ratio 0.5 (matcher.py:844-849)
                xy1 = ctr + dxy * 0.5
                dis2 = np.where(keep, np.sum((xy0 - xy1) ** 2, axis=-1), -1.0)
                max_dis = np.sqrt(np.maximum(dis2.max(axis=1), 0.0))
                nodes3 = np.full(keep.shape + (3,), -1, dtype=np.int64)
                B1 = np.full(keep.shape + (3,), np.nan)
                if is_def:
                    # Link.from_coordinates on the MOVING gear of the deformed mesh1 (matcher.py:717, optimizer.py:51-82):
                    # points outside the mesh are dropped; the INITIAL coordinates follow from the barycentric ones
                    xy1_init = np.zeros_like(xy1)
                    qq, kk = np.nonzero(keep)
                    if qq.size:
                        vmg = np.ascontiguousarray(v_all[sel] + U[sel])
                        pts = np.ascontiguousarray(xy1[qq, kk]); q32 = np.ascontiguousarray(qq, dtype=np.int32)
                        tid = np.empty(qq.size, dtype=np.int32); Bq = np.empty((qq.size, 3))
                        gxs = np.ascontiguousarray(self._gx[sel]); gys = np.ascontiguousarray(self._gy[sel])
                        _lib.check(_lib.load().fb_deformed_locate(_lib.ctx(), sel.size, m.grid_xs.size, m.grid_ys.size, _lib.ptr(gxs),
                                                                  _lib.ptr(gys), 1, _lib.ptr(vmg), qq.size, _lib.ptr(q32), _lib.ptr(pts),
                                                                  _lib.ptr(tid), _lib.ptr(Bq)))
                        inside = tid >= 0
                        keep[qq[~inside], kk[~inside]] = False
                        qq, kk, tid, Bq = qq[inside], kk[inside], tid[inside], Bq[inside]
                        nodes3[qq, kk] = m.triangles[tid] + (sel[qq] * V)[:, None]
                        B1[qq, kk] = Bq
                        xy1_init[qq, kk] = np.sum(v_all[sel[qq][:, None], m.triangles[tid]] * Bq[:, :, None], axis=1)
                    has_link = keep.any(axis=1)
                else:
                    xy1_init = xy1 - t1[sel][:, None, :]     # INITIAL gear of mesh1 at link creation (matcher.py:748-751)
                    has_link = anyk
                if rnd == 0:
                    active[sel[~has_link]] = False           # invalid_output (matcher.py:672-673, 719-721)
                live[sel[~has_link]] = False                 # ... or break with the links so far (674-675, 722-723)
                if not is_last:
                    # spacing schedule (matcher.py:689-716), max_spacing_skip = 0
                    next_pos = np.sum(self._sp[sel] > (4 * max_dis)[:, None], axis=1) - 1     # = searchsorted(-spacings, -4 max_dis) - 1
                    pad[sel] = np.where(next_pos > rnd, np.minimum(next_pos, rnd + 1) > rnd + 1, True)
                    # max_dis > 0.1: the reference relaxes mesh1 against the links (matcher.py:725-742).  When every
                    # kept block reports the same displacement the exact minimiser is the rigid translation
                    # u = xy0 - xy1 of mesh1 (zero elastic and zero link energy; checked against the FEM oracle in
                    # tests/test_gpu_pipeline.py); residues vanish, so the huber re-weighting changes nothing.
                    # Any other field is solved on the device below and makes the pair `deformed`.
                    move = has_link & (max_dis > 0.1)
                    if is_def:
                        to_relax[sel[move]] = True
                    elif move.any():
                        u = xy0 - xy1                                          # [Q, nblk, 2]
                        first = np.argmax(keep, axis=1)
                        u0 = u[np.arange(sel.size), first]                     # displacement of the first kept block
                        uniform = np.all(~keep[..., None] | (u == u0[:, None, :]), axis=(1, 2))
                        rigid = move & uniform
                        t1[sel[rigid]] += u0[rigid]
                        to_relax[sel[move & ~uniform]] = True
                pid = np.broadcast_to(sel[:, None], keep.shape)
                relax = np.broadcast_to((max_dis > 0.1)[:, None], keep.shape)
                rows.append((pid[keep], xy0[keep], xy1_init[keep], dcf[keep], xy1[keep], relax[keep], nodes3[keep], B1[keep]))
                has_last[sel[has_link]] = True
            if rows:
                prev = table
                table = tuple(np.concatenate([r[k] for r in rows], axis=0) for k in range(8))
                last_links = None
                pid_l, xy0_l, xy1i_l, wt_l, xy1_l, rl, nd_l, B1_l = table
                if to_relax.any():
                    # non-rigid relaxation between spacings (matcher.py:725-741): mesh1 of these pairs keeps the field
                    r = to_relax[pid_l]
                    nd_r, B1_r = self._rows_bary(pid_l[r], xy1i_l[r], nd_l[r], B1_l[r])
                    rw, x = self._relax_general(pid_l[r], xy0_l[r], nd_r, B1_r, wt_l[r], resolve=True)
                    if U is None:
                        U = np.zeros((n,) + x.shape[1:])
                    pr = np.flatnonzero(to_relax)
                    U[pr] = x[pr]
                    is_deformed[pr] = True
                    wt_new = wt_l.copy()
                    wt_new[r] = wt_l[r] * rw                                      # Link.weight (optimizer.py:313-317)
                    table = table[:3] + (wt_new,) + table[4:]
                if is_last and self.residue_len > 0 and rl.any():
                    # last round (matcher.py:725-737): relaxation + huber residue weights, pairs with max_dis > 0.1.  All rows
                    # of the round enter the block-diagonal system (a pair that needs no relaxation is solved and ignored), so
                    # that the strain estimate below can reuse the same links
                    if is_deformed.any() or self._ragged:
                        nd_a, B1_a = self._rows_bary(pid_l, xy1i_l, nd_l, B1_l)
                        rw, _ = self._relax_general(pid_l, xy0_l, nd_a, B1_a, wt_l, resolve=False)
                    else:
                        rw, _ = self._final_relax(pid_l, xy0_l, xy1_l, wt_l, t1)
                        last_links = True
                    wt_new = wt_l.copy()
                    wt_new[rl] = wt_l[rl] * rw[rl]                             # Link.weight (optimizer.py:313-317)
                    table = table[:3] + (wt_new,) + table[4:]
                table = table[:4]
                if prev is not None and prev[0].size:
                    # a pair without a confident block in this round keeps the links of its last good round
                    # (the reference breaks out of the loop before clear_links, matcher.py:671-679)
                    carry = ~np.isin(prev[0], table[0])
                    if carry.any():
                        table = tuple(np.concatenate((a, b[carry]), axis=0) for a, b in zip(table, prev))
                        last_links = None
        valid = active & has_last & ~deferred
        if table is None:
            table = (np.zeros(0, np.int64), np.zeros((0, 2)), np.zeros((0, 2)), np.zeros(0, np.float32))
        pid, xy0, xy1, wt = table
        ok = valid[pid]
        if self.residue_mode == 1:
            # threshold mode: matches cut by the residue filter are masked out of the link (Link.mask, optimizer.py:399-402)
            ok = ok & (wt > 0)
            valid = valid & (np.bincount(pid[ok], minlength=n) > 0)
        if not ok.all():
            pid, xy0, xy1, wt = pid[ok], xy0[ok], xy1[ok], wt[ok]
            last_links = None
        # output in the INITIAL gear: mesh0 points lose the translation (matcher.py:748-751)
        xy0 = xy0 - txy[pid]
        strain = self._strain(pid, xy0, xy1, wt, txy, reuse_links=bool(last_links)) if self.compute_strain else np.full(n, DEFAULT_AVG_DEFORM)
        self.last_field = U
        return dict(tx=tx, ty=ty, conf0=cf0, valid=valid, needs_host=np.zeros(n, dtype=bool), deformed=is_deformed, deferred=deferred, pair=pid,
                    xy0=xy0, xy1=xy1, weight=wt, strain=strain, phtm=phtm)

    @staticmethod
    def per_pair(res):
        """split the flat match table into the per-pair tuples stitching_matcher returns"""
        out = []
        n = res['tx'].size
        pid = np.asarray(res['pair'])
        order = np.argsort(pid, kind='stable')                # one sort instead of a mask per pair
        bounds = np.searchsorted(pid[order], np.arange(n + 1))
        xy0, xy1, wt = res['xy0'][order], res['xy1'][order], res['weight'][order]
        deferred = res.get('deferred', np.zeros(n, dtype=bool))
        for p in range(n):
            if res['valid'][p]:
                a, b = bounds[p], bounds[p + 1]
                out.append(dict(tx=res['tx'][p], ty=res['ty'][p], conf0=float(res['conf0'][p]), needs_host=False, deformed=bool(res['deformed'][p]),
                                deferred=False, xy0=xy0[a:b], xy1=xy1[a:b], weight=wt[a:b], strain=float(res['strain'][p])))
            else:
                out.append(dict(tx=res['tx'][p], ty=res['ty'][p], conf0=float(res['conf0'][p]), needs_host=False, deformed=False,
                                deferred=bool(deferred[p]), xy0=None, xy1=None, weight=None, strain=DEFAULT_AVG_DEFORM))
        return out


class RaggedStripBatchMatcher(StripBatchMatcher):
    """A batch of tile pairs whose strips differ in SIZE (the usual case in a real section: the overlap follows the stage
    jitter, stitcher.py:561-571) but share the mesh topology and the number of spacings -- `bucket_key`.  Strips sit
    in zero-padded slots of the largest size; every stage works on each pair's own extent: x0.5 downsample and DoG with
    per-image sizes (fb_area_downsample2_sizes_dev, fb_dog_sizes_dev), the whole-strip NCC through block descriptors
    grouped by FFT shape, per-pair block grids and spacing values, one mesh GEOMETRY per pair inside the shared
    block-diagonal system (per-pair node coordinates, Es0 and sample errors; fb_pairs_relax_bary / fb_pairs_strain_bary),
    and the deformed-mesh branch with per-pair node grids and tolerances (fb_deformed_*); photometric statistics on every
    pair's own extent (host numpy, like the reference's); masked pairs get their masked DoG image by image inside their slots."""

    @staticmethod
    def bucket_key(H, W, min_num_blocks=2, spacings=None):
        """pairs with equal keys can share a batch: number of spacings, node grid of Mesh.from_bbox (mesh.py:403-435)"""
        sp = np.sort(auto_spacings((H, W), (H, W)))[::-1] if spacings is None else np.sort(np.asarray(spacings, dtype=np.float64))[::-1]
        return (sp.size,) + grid_counts(H, W, float(np.min(sp)), min_num_blocks)

    def __init__(self, shapes, slot=None, **opts):
        shapes = np.asarray(shapes, dtype=np.int64).reshape(-1, 2)
        Hmax, Wmax = int(shapes[:, 0].max()), int(shapes[:, 1].max())
        if slot is not None:                                  # strips sit in slots larger than the largest of them
            assert slot[0] >= Hmax and slot[1] >= Wmax
            Hmax, Wmax = int(slot[0]), int(slot[1])
        spacings = opts.get('spacings', None)
        super().__init__(shapes.shape[0], Hmax, Wmax, **opts)
        self._ragged = True
        self._Hs, self._Ws = shapes[:, 0].copy(), shapes[:, 1].copy()
        if spacings is None:
            sp = [np.sort(auto_spacings((h, w), (h, w)))[::-1] for h, w in shapes]
            if len({a.size for a in sp}) != 1:
                raise ValueError('RaggedStripBatchMatcher: the pairs of a batch must have the same number of spacings')
            self._sp = np.stack(sp)
        self._nfl = _nfl_table(2 * max(Hmax, Wmax) + 2)
        from .common import half_size
        if self.cds == 0.5:
            self._hcs = np.array([half_size(int(h)) for h in self._Hs]); self._wcs = np.array([half_size(int(w)) for w in self._Ws])
        else:
            self._hcs, self._wcs = self._Hs, self._Ws
        n = self.P
        self._d_sizes = self._d_sizes_c = None               # device copies of the extents, made when the host route first needs them

    @property
    def d_sizes(self):
        if self._d_sizes is None:
            self._d_sizes = _lib.DeviceBuffer.from_array(np.ascontiguousarray(np.tile(np.stack((self._Hs, self._Ws), -1), (2, 1)), dtype=np.int32))
        return self._d_sizes

    @property
    def d_sizes_c(self):
        if self._d_sizes_c is None:
            self._d_sizes_c = _lib.DeviceBuffer.from_array(np.ascontiguousarray(np.tile(np.stack((self._hcs, self._wcs), -1), (2, 1)), dtype=np.int32))
        return self._d_sizes_c

    def _sub_matcher(self, fl):
        opts = dict(self._opts)
        return RaggedStripBatchMatcher(np.stack((self._Hs[fl], self._Ws[fl]), axis=-1), slot=(self.H, self.W), pool=self._pool, route='host', **opts)

    def free(self):
        for name in ('_d_sizes', '_d_sizes_c'):
            b = getattr(self, name, None)
            if b is not None:
                b.free()
                setattr(self, name, None)
        super().free()

    # ---- image stages on per-image extents
    def _masked_dog_in_slot(self, src_ptr, src_pitch, h, w, sigma, mask, dst_ptr, dst_pitch):
        """common.masked_dog_filter(img, sigma, mask=mask) of ONE uint8 image that occupies the top-left h x w pixels of a slot
        (row pitch src_pitch bytes) into the same corner of its float32 slot (row pitch dst_pitch bytes): the image is copied
        out dense, filtered like StripBatchMatcher._masked_dog does, and copied back"""
        lib, ctx = _lib.load(), _lib.ctx()
        mk = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        assert mk.shape == (h, w)
        d_m = _lib.DeviceBuffer.from_array(mk)
        d_in, d_out = _lib.DeviceBuffer(h * w), _lib.DeviceBuffer(4 * h * w)
        try:
            _lib.check(lib.fb_memcpy2d_d2d(ctx, d_in.ptr, w, src_ptr, int(src_pitch), w, h))
            _lib.check(lib.fb_dog_dev(ctx, d_in.ptr, 0, 1, h, w, sigma, d_m.ptr, 1, d_out.ptr))
            _lib.check(lib.fb_memcpy2d_d2d(ctx, dst_ptr, int(dst_pitch), d_out.ptr, 4 * w, 4 * w, h))
            _lib.check(lib.fb_sync(ctx))
        finally:
            d_m.free(); d_in.free(); d_out.free()

    def _coarse_mask(self, mk, p):
        """cv2.resize(mask, fx=0.5, INTER_NEAREST) of pair p's mask = every second pixel (matcher.py:257-264), on the pair's extent"""
        h_, w_ = int(self._hcs[p]), int(self._wcs[p])
        return np.asarray(mk)[::2, ::2][:h_, :w_] if self.cds == 0.5 else np.asarray(mk)[:h_, :w_]

    def _global(self, strips0, strips1, masks=None, need_small=False):
        lib, ctx = _lib.load(), _lib.ctx()
        n, H, W, hc, wc = self.P, self.H, self.W, self.hc, self.wc
        if self.cds == 0.5:
            _lib.check(lib.fb_area_downsample2_sizes_dev(ctx, strips0, n, H, W, self.d_sizes.ptr, self.d_small.ptr))
            _lib.check(lib.fb_area_downsample2_sizes_dev(ctx, strips1, n, H, W, self.d_sizes.ptr, self.d_small.offset(n * hc * wc)))
            _lib.check(lib.fb_dog_sizes_dev(ctx, self.d_small.ptr, 0, 2 * n, hc, wc, self.d_sizes_c.ptr, self.sigma * self.cds, 1, self.d_dogc.ptr))
        else:
            _lib.check(lib.fb_dog_sizes_dev(ctx, strips0, 0, n, hc, wc, self.d_sizes_c.ptr, self.sigma, 1, self.d_dogc.ptr))
            _lib.check(lib.fb_dog_sizes_dev(ctx, strips1, 0, n, hc, wc, self.d_sizes_c.ptr, self.sigma, 1, self.d_dogc.offset(n * hc * wc * 4)))
        if masks is not None:
            # masked pairs: the coarse DoG of their images again with the halo suppression of common.py:368-374, each on its own
            # extent of its slot (StripBatchMatcher._global does the same on whole slots)
            for side, (strips, mlist) in enumerate(((strips0, masks[0]), (strips1, masks[1]))):
                for p, mk in enumerate(mlist):
                    if mk is None:
                        continue
                    src = self.d_small.offset((side * n + p) * hc * wc) if self.cds == 0.5 else C.c_void_p(strips + p * H * W)
                    self._masked_dog_in_slot(src, wc, int(self._hcs[p]), int(self._wcs[p]), self.sigma * self.cds, self._coarse_mask(mk, p),
                                             self.d_dogc.offset((side * n + p) * hc * wc * 4), 4 * wc)
        # whole-strip NCC (matcher.py:153) of every pair's own extent: block descriptors, one launch per FFT shape
        fh = self._nfl[2 * self._hcs - 1]; fw = self._nfl[2 * self._wcs - 1]
        tx = np.zeros(n); ty = np.zeros(n); cf = np.zeros(n, dtype=np.float32)
        key = fh * 65536 + fw
        img1 = self.d_dogc.offset(n * hc * wc * 4)
        for kv in np.unique(key):
            sel = np.flatnonzero(key == kv)
            blk = np.zeros((sel.size, 9), dtype=np.int32)
            blk[:, 0] = sel
            blk[:, 3] = self._hcs[sel]; blk[:, 4] = self._wcs[sel]; blk[:, 7] = self._hcs[sel]; blk[:, 8] = self._wcs[sel]
            _lib.check(lib.fb_memcpy_h2d(ctx, self.d_blk.ptr, _lib.ptr(blk), blk.nbytes))
            nb = sel.size
            _lib.check(lib.fb_ncc_blocks_dev(ctx, self.d_dogc.ptr, img1, hc, wc, hc, wc, nb, self.d_blk.ptr, int(self._hcs[sel].max()), int(self._wcs[sel].max()),
                                             int(fh[sel[0]]), int(fw[sel[0]]), 0, self.conf_mode, self.d_out.ptr, self.d_out.offset(8 * nb),
                                             self.d_out.offset(16 * nb)))
            gx, gy, gc = self._fetch_out(nb)
            tx[sel] = gx; ty[sel] = gy; cf[sel] = gc
        low = np.flatnonzero(~(cf > self.conf_thresh))
        if low.size:                                          # second shot of global_translation_matcher (matcher.py:159-221), per pair
            from .matcher import global_translation_matcher
            full = self.d_dogc.to_array((2 * n, hc, wc), np.float32) if low.size > 2 else None
            for p in low:
                h_, w_ = int(self._hcs[p]), int(self._wcs[p])
                if full is not None:
                    g0, g1 = full[p, :h_, :w_], full[n + p, :h_, :w_]
                else:
                    g0 = self.d_dogc.to_array((2 * n, hc, wc), np.float32)[p, :h_, :w_]
                    g1 = self.d_dogc.to_array((2 * n, hc, wc), np.float32)[n + p, :h_, :w_]
                tx[p], ty[p], cf[p] = global_translation_matcher(np.ascontiguousarray(g0), np.ascontiguousarray(g1), conf_mode=self.conf_mode,
                                                                  conf_thresh=self.conf_thresh)
        return tx, ty, cf

    def _fine_dog(self, strips0, strips1, masks=None):
        lib, ctx = _lib.load(), _lib.ctx()
        n, H, W = self.P, self.H, self.W
        if self.cds == 1:
            self.d_dogf_view = self.d_dogc
            return
        _lib.check(lib.fb_dog_sizes_dev(ctx, strips0, 0, n, H, W, self.d_sizes.ptr, self.sigma, 1, self.d_dogf.ptr))
        _lib.check(lib.fb_dog_sizes_dev(ctx, strips1, 0, n, H, W, self.d_sizes.ptr, self.sigma, 1, self.d_dogf.offset(n * H * W * 4)))
        if masks is not None:
            for side, (strips, mlist) in enumerate(((strips0, masks[0]), (strips1, masks[1]))):
                for p, mk in enumerate(mlist):
                    if mk is not None:
                        h_, w_ = int(self._Hs[p]), int(self._Ws[p])
                        self._masked_dog_in_slot(C.c_void_p(strips + p * H * W), W, h_, w_, self.sigma, np.asarray(mk)[:h_, :w_],
                                                 self.d_dogf.offset((side * n + p) * H * W * 4), 4 * W)
        self.d_dogf_view = self.d_dogf

    def _photometric(self, strips0, strips1, tx_c, ty_c, masks):
        """matcher.py:279-314 on every pair's own extent of its slot (the statistics of StripBatchMatcher._photometric with
        per-pair image sizes and per-pair coarse masks)"""
        n, hc, wc = self.P, self.hc, self.wc
        if self.cds == 0.5:
            raw = self.d_small.to_array((2 * n, hc, wc), np.uint8)
        else:
            raw = np.empty((2 * n, hc, wc), dtype=np.uint8)
            for side, strips in enumerate((strips0, strips1)):
                _lib.check(_lib.load().fb_memcpy_d2h(_lib.ctx(), _lib.ptr(raw[side * n:(side + 1) * n]), C.c_void_p(strips), n * hc * wc))
        dog = self.d_dogc.to_array((2 * n, hc, wc), np.float32)
        out = []
        for p in range(n):
            h_, w_ = int(self._hcs[p]), int(self._wcs[p])
            txx, tyy = int(tx_c[p]), int(ty_c[p])
            xa, ya = max(txx, 0), max(tyy, 0)
            xb, yb = min(w_ + txx, w_), min(h_ + tyy, h_)
            if yb <= ya or xb <= xa:
                out.append(None)
                continue
            i0 = (slice(ya - tyy, yb - tyy), slice(xa - txx, xb - txx)); i1 = (slice(ya, yb), slice(xa, xb))
            mk0 = None if masks is None or masks[0][p] is None else np.asarray(self._coarse_mask(masks[0][p], p), dtype=bool)
            mk1 = None if masks is None or masks[1][p] is None else np.asarray(self._coarse_mask(masks[1][p], p), dtype=bool)
            m0 = np.ones((yb - ya, xb - xa), dtype=bool) if mk0 is None else mk0[i0]
            m1 = np.ones((yb - ya, xb - xa), dtype=bool) if mk1 is None else mk1[i1]
            if np.sum(m0) <= 3:                               # matcher.py:297
                out.append(None)
                continue
            mp = m0 & m1
            out.append((np.mean(raw[p][i0][mp]), np.mean(raw[n + p][i1][mp]), np.mean(np.abs(dog[p][i0][mp])), np.mean(np.abs(dog[n + p][i1][mp]))))
        return out

    # ---- one mesh geometry per pair inside the shared system
    def _relax_system(self):
        if self._relax_sys is not None:
            return self._relax_sys
        lib, ctx = _lib.load(), _lib.ctx()
        P = self.P
        # one template mesh for the topology; the node coordinates of every pair follow from its strip size
        m = Mesh.from_bbox((0, 0, int(self._Ws[0]), int(self._Hs[0])), cartesian=True, mesh_size=float(np.min(self._sp[0])),
                           min_num_blocks=self.mnb, uid=1)
        nx, ny = m.grid_xs.size, m.grid_ys.size
        for p in range(P):
            if grid_counts(int(self._Hs[p]), int(self._Ws[p]), float(np.min(self._sp[p])), self.mnb) != (nx, ny):
                raise ValueError('RaggedStripBatchMatcher: the pairs of a batch must share the node grid of their meshes (bucket_key)')
        self._mesh = m                                        # the topology (triangles, node grid sizes)
        self._gx = np.linspace(0.0, self._Ws.astype(np.float64), num=nx, endpoint=True, axis=-1) - 0.5      # mesh.py:430-431 per pair
        self._gy = np.linspace(0.0, self._Hs.astype(np.float64), num=ny, endpoint=True, axis=-1) - 0.5
        V, T = m.num_vertices, m.num_triangles
        self._sys_key = (P, m.grid_xs.size, m.grid_ys.size)
        sysh = self._pool.systems.pop(self._sys_key, None) if self._pool is not None else None
        if sysh is None:
            sysh = C.c_void_p()
            _lib.check(lib.fb_sys_create(ctx, P * V, C.byref(sysh)))
            tri_u = np.ascontiguousarray((m.triangles[None, :, :] + (np.arange(P) * V)[:, None, None]).reshape(-1, 3), dtype=np.int32)
            mid = C.c_int()
            _lib.check(lib.fb_sys_add_mesh(ctx, sysh, 0, _lib.ptr(tri_u), P * V, P * T, C.byref(mid)))
            _lib.check(lib.fb_sys_set_links(ctx, sysh, 0, None))
            nnzb = C.c_int64()
            _lib.check(lib.fb_sys_finalize(ctx, sysh, C.byref(nnzb)))
        self._mult_u = np.ascontiguousarray(np.tile(m.element_multiplier(), P), dtype=np.float32)
        vx = np.broadcast_to(self._gx[:, None, :], (P, ny, nx)); vy = np.broadcast_to(self._gy[:, :, None], (P, ny, nx))
        self._v_init_u = np.ascontiguousarray(np.stack((vx, vy), axis=-1).reshape(-1, 2), dtype=np.float64)
        self._relax_sys = sysh
        self._k_state = None
        self._assemble_union(self._v_init_u, 'initial')
        v0 = self._v_init_u.reshape(P, V, 2)
        v0 = np.ascontiguousarray((v0 - v0.mean(axis=1, keepdims=True)).reshape(-1, 2))
        es0 = np.empty(P)
        _lib.check(lib.fb_sys_group_energy(ctx, sysh, P, _lib.ptr(v0), _lib.ptr(es0)))
        self._es0_each = es0
        self._es0 = float(es0[0])
        # Mesh.triangle_areas is the cross product of two edges (common.py:672-676): (dx, 0) x (0, dy) for the triangles of a cell
        area = (self._gx[:, 1] - self._gx[:, 0]) * (self._gy[:, 1] - self._gy[:, 0])
        self._sample_err_each = 0.4387 * area ** 0.5 * DEFAULT_AVG_DEFORM          # optimizer.py:26-30, per pair
        return sysh

    def _pair_mesh(self, p):
        return Mesh.from_bbox((0, 0, int(self._Ws[p]), int(self._Hs[p])), cartesian=True, mesh_size=float(np.min(self._sp[p])),
                              min_num_blocks=self.mnb, uid=1)

    def _locate_grid(self, pid, pts):
        """triangle + barycentric coordinates of points given in the INITIAL gear of their pair's grid mesh (cart2bary on
        the right triangles of a cell; the statements of pairs_build_links in fb_fem.hip with per-pair node coordinates)"""
        self._relax_system()
        pid = np.asarray(pid)
        nx, ny = self._gx.shape[1], self._gy.shape[1]
        # the grids are uniform (Mesh.from_bbox: linspace(0, W, nx) - 0.5): the cell follows from one division per axis
        cw = ((self._gx[:, -1] - self._gx[:, 0]) / (nx - 1))[pid]; ch = ((self._gy[:, -1] - self._gy[:, 0]) / (ny - 1))[pid]
        fx = (pts[:, 0] - self._gx[pid, 0]) / cw; fy = (pts[:, 1] - self._gy[pid, 0]) / ch
        i = np.clip(np.floor(fx), 0, nx - 2).astype(np.int64); j = np.clip(np.floor(fy), 0, ny - 2).astype(np.int64)
        u = fx - i; w = fy - j
        up = w > u
        tid = 2 * (j * (nx - 1) + i) + up
        B = np.where(up[:, None], np.stack((1.0 - w, u, w - u), axis=-1), np.stack((1.0 - u, u - w, w), axis=-1))
        return tid, B

    def _rows_bary(self, pid, xy1_init, nodes3, B1):
        V = self._relax_system() and self._mesh.num_vertices
        tid, B = self._locate_grid(pid, np.asarray(xy1_init, dtype=np.float64))
        nodes = self._mesh.triangles[tid] + (np.asarray(pid) * V)[:, None]
        return nodes.astype(np.int64), B

    def _strain(self, pid, xy0, xy1, wt, txy, reuse_links=False):
        lib, ctx = _lib.load(), _lib.ctx()
        sysh = self._relax_system()
        P = self.P
        if pid.size == 0:
            return np.full(P, DEFAULT_AVG_DEFORM)
        nseg = 1 + int(np.count_nonzero(np.diff(pid) != 0))
        if nseg != int(np.count_nonzero(np.bincount(pid, minlength=P))):
            o = np.argsort(pid, kind='stable')
            pid, xy0, xy1, wt = pid[o], xy0[o], xy1[o], wt[o]
        p0 = np.ascontiguousarray(xy0 + txy[pid], dtype=np.float64)
        xy1c = np.ascontiguousarray(xy1, dtype=np.float64)
        w32 = np.ascontiguousarray(wt, dtype=np.float32)
        R = np.ascontiguousarray(self._rigid_fits(pid, p0, xy1c, w32))
        nodes, B = self._rows_bary(pid, xy1c, None, None)
        nodes = np.ascontiguousarray(nodes, dtype=np.int32); B = np.ascontiguousarray(B)
        pid32 = np.ascontiguousarray(pid, dtype=np.int32)
        strain = np.empty(P)
        iters, relres = C.c_int(), C.c_double()
        es0 = np.ascontiguousarray(self._es0_each)
        _lib.check(lib.fb_pairs_strain_bary(ctx, sysh, P, pid.size, _lib.ptr(pid32), _lib.ptr(nodes), _lib.ptr(B), _lib.ptr(p0), _lib.ptr(xy1c),
                                            _lib.ptr(w32), _lib.ptr(R), self.stiffness_lambda, _lib.ptr(es0), DEFAULT_AVG_DEFORM, _lib.ptr(strain),
                                            C.byref(iters), C.byref(relres)))
        self.last_strain_solve = dict(iters=iters.value, relres=relres.value, matches=int(pid.size))
        return strain
