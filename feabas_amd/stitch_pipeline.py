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
What is NOT here yet (DESIGN.md "scope of the pair pipeline"): the low-confidence fallback
of global_translation_matcher (159-221, host path exists in matcher.py); NON-RIGID mesh relaxation
between spacings (725-742; a uniform block displacement is applied as the rigid translation it
relaxes to) together with the bilinear patch gather it needs; the
final residue re-weighting and strain estimate.  Pairs that would take the non-rigid branch are
flagged in the result (``needs_host``).
"""
import ctypes as C

import numpy as np

from . import _lib
from . import constant as const
from .matcher import auto_spacings, next_fast_len
from .mesh import Mesh

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


class StripBatchMatcher:
    def __init__(self, P, H, W, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, min_num_blocks=2,
                 conf_mode=const.FFT_CONF_MIRROR, residue_len=5, stiffness_lambda=1.0, relax_tol=1e-9):
        assert coarse_downsample in (0.5, 1)
        assert H % 2 == 0 and W % 2 == 0 or coarse_downsample == 1
        self.P, self.H, self.W = int(P), int(H), int(W)
        self.sigma = float(sigma)
        self.cds = coarse_downsample
        self.conf_thresh = float(conf_thresh)
        self.mnb = int(min_num_blocks)
        self.conf_mode = int(conf_mode)
        self.spacings = np.sort(auto_spacings((H, W), (H, W)))[::-1]     # matcher.py:243-251, 567
        self._nfl = np.array([next_fast_len(v) for v in range(0, 2 * max(H, W) + 2)])
        hc, wc = (H // 2, W // 2) if coarse_downsample == 0.5 else (H, W)
        self.hc, self.wc = hc, wc
        n = self.P
        self.d_small = _lib.DeviceBuffer(2 * n * hc * wc) if coarse_downsample == 0.5 else None
        self.d_dogc = _lib.DeviceBuffer(2 * n * hc * wc * 4)
        self.d_dogf = _lib.DeviceBuffer(2 * n * H * W * 4)
        self.max_blocks = n * 1024
        self.d_blk = _lib.DeviceBuffer(self.max_blocks * 9 * 4)
        self.d_out = _lib.DeviceBuffer(self.max_blocks * 20)       # per launch: [dx f64 N][dy f64 N][conf f32 N], one D2H copy
        self.residue_len = float(residue_len)                 # matcher.py:236 (fine_downsample = 1)
        self.stiffness_lambda = float(stiffness_lambda)       # matcher.py:507
        self.relax_tol = float(relax_tol)
        self._relax_sys = None
        self.last_relax = None

    def free(self):
        if self._relax_sys is not None:
            _lib.load().fb_sys_destroy(_lib.ctx(), self._relax_sys)
            self._relax_sys = None
        for b in (self.d_small, self.d_dogc, self.d_dogf, self.d_blk, self.d_out):
            if b is not None:
                b.free()

    # ------------------------------------------------------------------ stages
    def _global(self, strips0, strips1):
        lib, ctx = _lib.load(), _lib.ctx()
        n, H, W, hc, wc = self.P, self.H, self.W, self.hc, self.wc
        if self.cds == 0.5:
            _lib.check(lib.fb_area_downsample2_dev(ctx, strips0, n, H, W, self.d_small.ptr))
            _lib.check(lib.fb_area_downsample2_dev(ctx, strips1, n, H, W, self.d_small.offset(n * hc * wc)))
            _lib.check(lib.fb_dog_dev(ctx, self.d_small.ptr, 0, 2 * n, hc, wc, self.sigma * self.cds, None, 1, self.d_dogc.ptr))
        else:
            _lib.check(lib.fb_dog_dev(ctx, strips0, 0, n, hc, wc, self.sigma, None, 1, self.d_dogc.ptr))
            _lib.check(lib.fb_dog_dev(ctx, strips1, 0, n, hc, wc, self.sigma, None, 1, self.d_dogc.offset(n * hc * wc * 4)))
        _lib.check(lib.fb_ncc_batch_dev(ctx, self.d_dogc.ptr, self.d_dogc.offset(n * hc * wc * 4), n, 1, hc, wc, hc, wc,
                                        1, 0, self.conf_mode, self.d_out.ptr, self.d_out.offset(8 * n), self.d_out.offset(16 * n)))
        return self._fetch_out(n)    # equal strip sizes: (W1-W0)/2 = 0 (matcher.py:155-156)

    def _fetch_out(self, nb):
        """(dx, dy, conf) of the last launch that wrote nb results into d_out"""
        raw = self.d_out.to_array((20 * nb,), np.uint8)
        return raw[:8 * nb].view(np.float64), raw[8 * nb:16 * nb].view(np.float64), raw[16 * nb:].view(np.float32)

    def _fine_dog(self, strips0, strips1):
        lib, ctx = _lib.load(), _lib.ctx()
        n, H, W = self.P, self.H, self.W
        if self.cds == 1:
            self.d_dogf_view = self.d_dogc        # matcher.py:315-317: same image when fine == coarse
            return
        _lib.check(lib.fb_dog_dev(ctx, strips0, 0, n, H, W, self.sigma, None, 1, self.d_dogf.ptr))
        _lib.check(lib.fb_dog_dev(ctx, strips1, 0, n, H, W, self.sigma, None, 1, self.d_dogf.offset(n * H * W * 4)))
        self.d_dogf_view = self.d_dogf

    def _blocks(self, tx, ty, t1, sel, spacing, mnb):
        """block descriptors for the pairs `sel` (all share Nx, Ny): returns (blk [Q, nblk, 9], bboxes [Q, nblk, 4])"""
        H, W = self.H, self.W
        # mesh bounding boxes in the MOVING gear (Mesh.from_bbox: vertices at pixel centres - 0.5)
        xmin = np.maximum(-0.5 + tx[sel], -0.5 + t1[sel, 0]); ymin = np.maximum(-0.5 + ty[sel], -0.5 + t1[sel, 1])
        xmax = np.minimum(W - 0.5 + tx[sel], W - 0.5 + t1[sel, 0]); ymax = np.minimum(H - 0.5 + ty[sel], H - 0.5 + t1[sel, 1])
        nx, ny, dx, dy = _divide_bbox_batch(xmin, ymin, xmax, ymax, spacing, mnb)
        assert np.all(nx == nx[0]) and np.all(ny == ny[0])
        nxi, nyi = int(nx[0]), int(ny[0])
        xt = np.round(np.linspace(xmin, xmax - dx, num=nxi, endpoint=True, axis=-1)).astype(np.int32)    # [Q, nx]
        yt = np.round(np.linspace(ymin, ymax - dy, num=nyi, endpoint=True, axis=-1)).astype(np.int32)    # [Q, ny]
        x0 = np.broadcast_to(xt[:, None, :], (sel.size, nyi, nxi)).reshape(sel.size, -1)
        y0 = np.broadcast_to(yt[:, :, None], (sel.size, nyi, nxi)).reshape(sel.size, -1)
        order = _z_order_batch(np.round((x0 - x0.min(axis=-1, keepdims=True)) / spacing),
                               np.round((y0 - y0.min(axis=-1, keepdims=True)) / spacing))
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
        blk[:, :, 5] = x0 - np.round(t1[sel, 0]).astype(np.int32)[:, None]
        blk[:, :, 6] = y0 - np.round(t1[sel, 1]).astype(np.int32)[:, None]
        blk[:, :, 7] = dy[:, None]; blk[:, :, 8] = dx[:, None]
        return blk, bb

    def _match_round(self, tx, ty, t1, active, spacing, mnb, pad_flags, subpixel):
        """one spacing round for the `active` pairs.  Returns a list of groups
        (pair ids [Q], bboxes [Q, nblk, 4], dx, dy, conf [Q, nblk])."""
        lib, ctx = _lib.load(), _lib.ctx()
        n, H, W = self.P, self.H, self.W
        # group by block grid, then by FFT shape (matcher.py:59-62 on the block size)
        xmin = np.maximum(-0.5 + tx, -0.5 + t1[:, 0]); xmax = np.minimum(W - 0.5 + tx, W - 0.5 + t1[:, 0])
        ymin = np.maximum(-0.5 + ty, -0.5 + t1[:, 1]); ymax = np.minimum(H - 0.5 + ty, H - 0.5 + t1[:, 1])
        nx, ny, dx, dy = _divide_bbox_batch(xmin, ymin, xmax, ymax, spacing, mnb)
        nfl_h = nfl_w = self._nfl
        fh = np.where(pad_flags, nfl_h[np.clip(2 * dy - 1, 0, None)], nfl_h[dy])
        fw = np.where(pad_flags, nfl_w[np.clip(2 * dx - 1, 0, None)], nfl_w[dx])
        key = ((nx * 4096 + ny) * 8192 + fh) * 8192 + fw
        key = np.where(active, key, -1)
        groups = []
        dogf = self.d_dogf_view
        img1 = dogf.offset(n * H * W * 4)
        for kv in np.unique(key):
            if kv < 0:
                continue
            sel = np.flatnonzero(key == kv)
            gfh, gfw = int(fh[sel[0]]), int(fw[sel[0]])
            blk, bb = self._blocks(tx, ty, t1, sel, spacing, mnb)
            nb = blk.shape[0] * blk.shape[1]
            assert nb <= self.max_blocks
            flat = np.ascontiguousarray(blk.reshape(-1, 9))
            _lib.check(lib.fb_memcpy_h2d(ctx, self.d_blk.ptr, _lib.ptr(flat), flat.nbytes))
            _lib.check(lib.fb_ncc_blocks_dev(ctx, dogf.ptr, img1, H, W, H, W, nb, self.d_blk.ptr, int(dy[sel].max()), int(dx[sel].max()), gfh, gfw,
                                             1 if subpixel else 0, self.conf_mode, self.d_out.ptr, self.d_out.offset(8 * nb), self.d_out.offset(16 * nb)))
            ddx, ddy, dcf = (a.reshape(sel.size, -1) for a in self._fetch_out(nb))
            groups.append((sel, bb, ddx, ddy, dcf))
        return groups

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
        V = m.num_vertices
        sysh = C.c_void_p()
        _lib.check(lib.fb_sys_create(ctx, self.P * V, C.byref(sysh)))
        v0 = np.ascontiguousarray(m.vertices(const.MESH_GEAR_INITIAL), dtype=np.float64)
        for p in range(self.P):
            mid = C.c_int()
            _lib.check(lib.fb_sys_add_mesh(ctx, sysh, p * V, _lib.ptr(m.triangles), V, m.num_triangles, C.byref(mid)))
        _lib.check(lib.fb_sys_set_links(ctx, sysh, 0, None))
        nnzb = C.c_int64()
        _lib.check(lib.fb_sys_finalize(ctx, sysh, C.byref(nnzb)))
        for p in range(self.P):
            m.assemble_into(sysh, p, v0, None, 1.0)           # translation invariant: shape = INITIAL vertices, no stress
        self._relax_sys = sysh
        return sysh

    def _final_relax(self, pid, xy0, xy1, wt, t1):
        """matcher.py:725-737 for every pair of the batch at once: relax mesh1 against the last-round links
        (optimize_linear, to the fixed point), then huber residue weights (optimizer.py:174-191, 203-205).
        pid [K] sorted or not; xy0/xy1 [K, 2] in the MOVING gear; wt [K] confidences; t1 [P, 2] mesh1 offsets.
        Returns the residue weight [K] float32, the displacement of the mesh1 end of each match and of every
        mesh1 vertex [P, V, 2]."""
        lib, ctx = _lib.load(), _lib.ctx()
        sysh = self._relax_system()
        m = self._mesh
        V = m.num_vertices
        K = pid.size
        q1 = xy1 - t1[pid]                                    # mesh1 coordinates without its offset
        tid1 = m.locate_cartesian(q1)
        tv = m.triangles[tid1]                                # [K, 3]
        pv = m.vertices(const.MESH_GEAR_INITIAL)[tv]          # [K, 3, 2]
        d0, d1, d2 = q1 - pv[:, 0], q1 - pv[:, 1], q1 - pv[:, 2]
        a0 = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]        # mesh.py:2191-2217
        a1 = d2[:, 0] * d0[:, 1] - d2[:, 1] * d0[:, 0]
        a2 = d0[:, 0] * d1[:, 1] - d0[:, 1] * d1[:, 0]
        tot = a0 + a1 + a2
        B1 = np.stack((a0 / tot, a1 / tot, a2 / tot), axis=-1)
        nodes6 = np.full((K, 6), -1, dtype=np.int32)
        nodes6[:, 3:] = tv + (pid * V)[:, None]
        bary6 = np.zeros((K, 6))
        bary6[:, 0] = 1.0                                     # locked side: not used by the assembly
        bary6[:, 3:] = -B1
        w32 = np.ascontiguousarray(wt, dtype=np.float32)
        dxy = np.ascontiguousarray(xy1 - xy0, dtype=np.float64)          # Link.dxy (optimizer.py:248-255)
        _lib.check(lib.fb_sys_update_links(ctx, sysh, K, _lib.ptr(nodes6)))
        _lib.check(lib.fb_sys_assemble_links(ctx, sysh, _lib.ptr(bary6), _lib.ptr(w32), _lib.ptr(dxy)))
        _lib.check(lib.fb_sys_form_groups(ctx, sysh, self.P, self.stiffness_lambda, -1.0, None))
        x = np.zeros(2 * self.P * V, dtype=np.float64)
        iters, relres = C.c_int(), C.c_double()
        _lib.check(lib.fb_sys_solve_groups(ctx, sysh, self.P, _lib.ptr(x), self.relax_tol, 0.0, 20 * V, 1, C.byref(iters), C.byref(relres)),
                   allow=(_lib.FB_ERR_NOCONV,))
        self.last_relax = dict(iters=iters.value, relres=relres.value, matches=int(K))
        x = x.reshape(-1, 2)
        u = np.sum(x[nodes6[:, 3:]] * B1[:, :, None], axis=1)            # displacement of the mesh1 end of each match
        res = dxy + u
        dis2 = np.sum(res ** 2, axis=-1)
        area = float(np.abs(m.triangle_areas(const.MESH_GEAR_INITIAL)[0]))
        sample_err = 0.4387 * area ** 0.5 * DEFAULT_AVG_DEFORM          # optimizer.py:26-30, equal triangles on both sides
        dis = np.sqrt(np.clip(dis2 - sample_err ** 2, 0, None))
        L = self.residue_len
        return (L / np.maximum(dis, L)).astype(np.float32), u, x.reshape(self.P, V, 2)

    # ------------------------------------------------------------------ driver
    def match(self, strips0, strips1):
        """strips0/strips1: device pointers to uint8 [P][H][W].  Returns a dict of arrays:
        tx, ty, conf0, valid, needs_host [P]; the match table as flat arrays pair, xy0, xy1, weight
        (rows of one pair are contiguous, pairs in ascending order within a block-grid group)."""
        n = self.P
        tx, ty, cf0 = self._global(strips0, strips1)
        scale = 1.0 / self.cds
        tx = tx * scale; ty = ty * scale                     # matcher.py:338-339
        active = cf0 >= self.conf_thresh                     # matcher.py:277-278
        self._fine_dog(strips0, strips1)
        spacings = self.spacings
        pad = np.ones(n, dtype=bool)
        needs_host = np.zeros(n, dtype=bool)
        has_last = np.zeros(n, dtype=bool)
        table = None
        txy = np.stack((tx, ty), axis=-1)
        t1 = np.zeros((n, 2))                                # translation of mesh1 acquired by rigid relaxations
        live = active.copy()                                 # pairs still iterating over the spacings
        for rnd in range(spacings.size):
            sp = spacings[rnd]
            is_last = rnd == spacings.size - 1
            mnb = self.mnb if is_last else 1
            rows = []
            for sel, bb, ddx, ddy, dcf in self._match_round(tx, ty, t1, live, sp, mnb, pad, subpixel=is_last):
                keep = dcf > self.conf_thresh                # matcher.py:671-683
                anyk = keep.any(axis=1)
                if rnd == 0:
                    active[sel[~anyk]] = False               # invalid_output (matcher.py:672-673)
                live[sel[~anyk]] = False                     # ... or break with the links so far (674-675)
                ctr = 0.5 * np.stack((bb[..., 0] + bb[..., 2], bb[..., 1] + bb[..., 3]), axis=-1) - 0.5      # bbox_centers
                dxy = np.stack((ddx, ddy), axis=-1)
                xy0 = ctr - dxy * 0.5                        # equal block sizes: ratio 0.5 (matcher.py:844-849)
                xy1 = ctr + dxy * 0.5
                xy1_init = xy1 - t1[sel][:, None, :]         # INITIAL gear of mesh1 at link creation (matcher.py:748-751)
                dis2 = np.where(keep, np.sum((xy0 - xy1) ** 2, axis=-1), -1.0)
                max_dis = np.sqrt(np.maximum(dis2.max(axis=1), 0.0))
                if not is_last:
                    # spacing schedule (matcher.py:689-716), max_spacing_skip = 0
                    next_pos = np.searchsorted(-spacings, -4 * max_dis) - 1
                    pad[sel] = np.where(next_pos > rnd, np.minimum(next_pos, rnd + 1) > rnd + 1, True)
                    # max_dis > 0.1: the reference relaxes mesh1 against the links (matcher.py:725-742).  When every
                    # kept block reports the same displacement the exact minimiser is the rigid translation
                    # u = xy0 - xy1 of mesh1 (zero elastic and zero link energy; checked against the FEM oracle in
                    # tests/test_gpu_pipeline.py); residues vanish, so the huber re-weighting changes nothing.
                    # Any other field needs the deformed-mesh crop: flagged.
                    move = anyk & (max_dis > 0.1)
                    if move.any():
                        u = xy0 - xy1                                          # [Q, nblk, 2]
                        first = np.argmax(keep, axis=1)
                        u0 = u[np.arange(sel.size), first]                     # displacement of the first kept block
                        uniform = np.all(~keep[..., None] | (u == u0[:, None, :]), axis=(1, 2))
                        rigid = move & uniform
                        t1[sel[rigid]] += u0[rigid]
                        needs_host[sel[move & ~uniform]] = True
                pid = np.broadcast_to(sel[:, None], keep.shape)
                relax = np.broadcast_to((max_dis > 0.1)[:, None], keep.shape)
                rows.append((pid[keep], xy0[keep], xy1_init[keep], dcf[keep], xy1[keep], relax[keep]))
                has_last[sel[anyk]] = True
            if rows:
                prev = table
                table = tuple(np.concatenate([r[k] for r in rows], axis=0) for k in range(6))
                if is_last and self.residue_len > 0:
                    # last round (matcher.py:725-737): relaxation + huber residue weights, pairs with max_dis > 0.1
                    pid_l, xy0_l, _, wt_l, xy1_l, rl = table
                    if rl.any():
                        rw, _, _ = self._final_relax(pid_l[rl], xy0_l[rl], xy1_l[rl], wt_l[rl], t1)
                        wt_new = wt_l.copy()
                        wt_new[rl] = wt_l[rl] * rw                                 # Link.weight (optimizer.py:313-317)
                        table = table[:3] + (wt_new,) + table[4:]
                table = table[:4]
                if prev is not None and prev[0].size:
                    # a pair without a confident block in this round keeps the links of its last good round
                    # (the reference breaks out of the loop before clear_links, matcher.py:671-679)
                    carry = ~np.isin(prev[0], table[0])
                    if carry.any():
                        table = tuple(np.concatenate((a, b[carry]), axis=0) for a, b in zip(table, prev))
        valid = active & has_last
        if table is None:
            table = (np.zeros(0, np.int64), np.zeros((0, 2)), np.zeros((0, 2)), np.zeros(0, np.float32))
        pid, xy0, xy1, wt = table
        ok = valid[pid]
        pid, xy0, xy1, wt = pid[ok], xy0[ok], xy1[ok], wt[ok]
        # output in the INITIAL gear: mesh0 points lose the translation (matcher.py:748-751)
        xy0 = xy0 - txy[pid]
        return dict(tx=tx, ty=ty, conf0=cf0, valid=valid, needs_host=needs_host, pair=pid, xy0=xy0, xy1=xy1, weight=wt)

    @staticmethod
    def per_pair(res):
        """split the flat match table into the per-pair tuples stitching_matcher returns"""
        out = []
        for p in range(res['tx'].size):
            m = res['pair'] == p
            if res['valid'][p]:
                out.append(dict(tx=res['tx'][p], ty=res['ty'][p], conf0=float(res['conf0'][p]), needs_host=bool(res['needs_host'][p]),
                                xy0=res['xy0'][m], xy1=res['xy1'][m], weight=res['weight'][m]))
            else:
                out.append(dict(tx=res['tx'][p], ty=res['ty'][p], conf0=float(res['conf0'][p]), needs_host=False,
                                xy0=None, xy1=None, weight=None))
        return out
