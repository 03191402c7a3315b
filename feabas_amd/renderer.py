"""Host mirror of the part of ``feabas.renderer.MeshRenderer`` the block matcher uses -- ``from_mesh`` (renderer.py:47-166)
and ``crop_multiple`` (601-648) -- for general triangulated meshes of one region without collisions.  The image stays
resident in HBM; locating pixels in triangles, the field, cv2.remap's bilinear sampling and the masked DoG of the stack are
device kernels (csrc/fb_render.hip, fb_dog.hip).  The host keeps what the reference keeps in shapely / numpy too: the
affine approximator (global fit, per-block fit over the triangles that touch the block) and its tolerance test.

Outside the scope here (the reference's optional modes): several regions / collision weights (MESH_TRIFINDER_INNERMOST),
geodesic masks, multichannel images, a renderer resolution different from the loader's.
"""
import numpy as np

from . import _lib
from . import constant as const
from .common import fit_affine
from .deformed import affine_residue


class ResidentImage:
    """a 2-D uint8 / float32 image in HBM whose pixel (0, 0) sits at ``origin`` = (x0, y0) of the image space the mesh
    maps into; zero outside (dal.StreamLoader fillval 0)."""

    def __init__(self, image, origin=(0, 0)):
        image = np.asarray(image)
        if image.ndim != 2:
            raise NotImplementedError('multichannel images are outside the device renderer')
        if image.dtype != np.uint8:
            image = image.astype(np.float32, copy=False)
        self.dtype = 0 if image.dtype == np.uint8 else 1                 # FB_U8 / FB_F32
        self.shape = image.shape
        self.origin = (int(origin[0]), int(origin[1]))
        self.buf = _lib.DeviceBuffer.from_array(image)

    @classmethod
    def from_loader(cls, loader, bbox, **kwargs):
        """the area `bbox` = (xmin, ymin, xmax, ymax) of a reference-style image loader (dal.StreamLoader /
        MosaicLoader / ...: ``crop(bbox, return_empty=, fillval=)``, dal.py:1045-1050) read ONCE and kept in HBM: what the
        reference's renderer asks its loader for block by block (renderer.py:601-631) is served from this copy.  Pixels of
        the area outside the loader's image hold the loader's fill value (its crop fills them); outside `bbox` reads 0."""
        bbox = [int(v) for v in bbox]
        img = loader.crop(bbox, return_empty=True, **kwargs)
        if img is None:
            img = np.zeros((bbox[3] - bbox[1], bbox[2] - bbox[0]), dtype=np.uint8)
        return cls(np.ascontiguousarray(img), origin=(bbox[0], bbox[1]))

    def free(self):
        self.buf.free()


def _sat_hits(tp, boxes):
    """closed triangles tp [K, 3, 2] against their closed boxes [K, 4] (pairwise): the ``intersects`` predicate of the
    STRtree query at renderer.py:405; touching counts (cf. deformed.tri_box_hits for the dense form)."""
    bx = np.asarray(boxes, dtype=np.float64)
    sep = (tp[:, :, 0].max(axis=1) < bx[:, 0]) | (tp[:, :, 0].min(axis=1) > bx[:, 2]) | \
          (tp[:, :, 1].max(axis=1) < bx[:, 1]) | (tp[:, :, 1].min(axis=1) > bx[:, 3])
    cx = np.stack((bx[:, 0], bx[:, 2], bx[:, 2], bx[:, 0]), axis=-1)
    cy = np.stack((bx[:, 1], bx[:, 1], bx[:, 3], bx[:, 3]), axis=-1)
    for k in range(3):
        e = tp[:, (k + 1) % 3] - tp[:, k]
        nx_, ny_ = -e[:, 1], e[:, 0]
        pt = tp[:, :, 0] * nx_[:, None] + tp[:, :, 1] * ny_[:, None]
        pb = cx * nx_[:, None] + cy * ny_[:, None]
        sep |= (pt.max(axis=1) < pb.min(axis=1)) | (pt.min(axis=1) > pb.max(axis=1))
    return ~sep


class MeshRenderer:
    def __init__(self, mesh, image, gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_INITIAL), affine_approx_tol=0.0):
        self.mesh = mesh
        self.image = image if isinstance(image, ResidentImage) else ResidentImage(image)
        self._own_image = not isinstance(image, ResidentImage)
        self.tol = float(affine_approx_tol)
        self._offset = np.asarray(mesh.offset(gear[0]), dtype=np.float64).ravel()          # renderer.py:83
        self.v0 = np.ascontiguousarray(mesh.vertices(gear[0]), dtype=np.float64)             # field domain, offset removed
        self.v1 = np.ascontiguousarray(mesh.vertices_w_offset(gear[-1]), dtype=np.float64)   # image space
        self.tris = np.ascontiguousarray(mesh.triangles, dtype=np.int32)
        self.d_v0 = _lib.DeviceBuffer.from_array(self.v0)
        self.d_v1 = _lib.DeviceBuffer.from_array(self.v1)
        self.d_tris = _lib.DeviceBuffer.from_array(self.tris)
        self._global = None
        if self.tol > 0:                                                                     # renderer.py:92-101
            vidx = np.unique(self.tris)
            A = fit_affine(self.v1[vidx], self.v0[vidx])
            self._global = (A, affine_residue(self.v1[vidx], self.v0[vidx], A))

    @classmethod
    def from_mesh(cls, srcmesh, gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_INITIAL), **kwargs):
        image = kwargs.get('image_loader', kwargs.get('image', None))
        if image is None:
            raise RuntimeError('Image loader not defined.')
        if srcmesh.num_triangles == 0:
            return None
        return cls(srcmesh, image, gear=gear, affine_approx_tol=kwargs.get('affine_approx_tol', 0.0))

    def free(self):
        for b in (self.d_v0, self.d_v1, self.d_tris):
            b.free()
        if self._own_image:
            self.image.free()

    # ------------------------------------------------------------------------------------------------------------------
    def _candidates(self, d_org, nb, h, w):
        lib, ctx = _lib.load(), _lib.ctx()
        cap = 64
        while True:
            d_cand = _lib.DeviceBuffer(4 * nb * cap)
            d_cnt = _lib.DeviceBuffer(4 * nb)
            _lib.check(lib.fb_mesh_candidates_dev(ctx, self.tris.shape[0], self.d_v0.ptr, self.d_tris.ptr, nb, d_org.ptr, h, w, cap,
                                                  d_cand.ptr, d_cnt.ptr))
            cnt = d_cnt.to_array((nb,), np.int32)
            if cnt.max(initial=0) <= cap:
                return d_cand, d_cnt, cnt, cap
            d_cand.free(); d_cnt.free()
            cap = int(2 ** np.ceil(np.log2(cnt.max())))

    def _precise(self, tier, d_org, nb, h, w, d_cand, d_cnt, cap):
        """precise_mask of crop_field_affine (renderer.py:437-447): an affine block of which 1 px^2 or more lies outside the
        mesh is masked pixel by pixel (tier + 10); the area is taken in MOVING coordinates from the candidate triangles
        (fb_mesh_block_uncovered_dev: the candidate lists never leave the device)"""
        sel = np.flatnonzero((tier == 1) | (tier == 2))
        if sel.size:
            d_unc = _lib.DeviceBuffer(8 * nb)
            try:
                _lib.check(_lib.load().fb_mesh_block_uncovered_dev(_lib.ctx(), self.d_v0.ptr, self.d_tris.ptr, nb, d_org.ptr, h, w, cap, d_cand.ptr, d_cnt.ptr,
                                                                   d_unc.ptr))
                unc = d_unc.to_array((nb,), np.float64)
            finally:
                d_unc.free()
            out = sel[unc[sel] >= 1.0]
            tier[out] += 10
        return tier

    def _tiers(self, org, d_org, h, w, d_cand, d_cnt, cnt, cap, precise=False):
        """crop_field's choice per block (renderer.py:453-511): 1 global affine, 2 block affine, 3 exact field; + 10 when the
        precise mask applies (log_sigma > 0 and the block sticks out of the mesh)"""
        nb = org.shape[0]
        tier = np.full(nb, 3, dtype=np.int32)
        A6 = np.zeros((nb, 6))
        pack = lambda A: np.array([A[0, 0], A[1, 0], A[2, 0], A[0, 1], A[1, 1], A[2, 1]])
        if not self.tol > 0:
            return tier, A6
        A_g, res_g = self._global
        if res_g < self.tol:
            tier[:] = 1
            A6[:] = pack(A_g)
            return (self._precise(tier, d_org, nb, h, w, d_cand, d_cnt, cap) if precise else tier), A6
        cand = d_cand.to_array((nb, cap), np.int32)
        cnt = np.ascontiguousarray(cnt, dtype=np.int32)
        _lib.check(_lib.load().fb_mesh_block_affines(_lib.ctx(), self.v0.shape[0], _lib.ptr(self.v0), _lib.ptr(self.v1), _lib.ptr(self.tris), nb,
                                                     _lib.ptr(np.ascontiguousarray(org)), h, w, cap, _lib.ptr(cand), _lib.ptr(cnt), self.tol,
                                                     _lib.ptr(tier), _lib.ptr(A6)))
        for b in np.flatnonzero(tier == -1):                      # rank-deficient or flipped vertex set: spatial.py:45-60
            ti = cand[b, :cnt[b]]
            box = np.concatenate((org[b], org[b] + np.array([w, h], dtype=np.float64))) - 0.5
            ti = ti[_sat_hits(self.v0[self.tris[ti]], np.tile(box, (ti.size, 1)))]
            idx = np.unique(self.tris[ti])
            _, A_b = fit_affine(self.v1[idx], self.v0[idx], return_rigid=True, svd_clip=None)
            tier[b] = 3
            if affine_residue(self.v1[idx], self.v0[idx], A_b) < self.tol:
                tier[b] = 2
                A6[b] = pack(A_b)
        return (self._precise(tier, d_org, nb, h, w, d_cand, d_cnt, cap) if precise else tier), A6

    def render_stack_dev(self, bboxes, precise_mask=False):
        """crop_multiple(bboxes, mode=RENDER_FULL, remap_interp=INTER_LINEAR) before its DoG, for blocks of ONE size:
        (stack float32 [N][h][w], mask uint8 [N][h][w]) as device buffers + (N, h, w) + the tiers.  precise_mask = log_sigma > 0
        (crop_field passes it on to crop_field_affine, renderer.py:491-511)"""
        lib, ctx = _lib.load(), _lib.ctx()
        bboxes = np.asarray(bboxes).reshape(-1, 4)
        nb = bboxes.shape[0]
        bbox0 = bboxes.astype(np.float64) - np.tile(self._offset, 2)
        w, h = int(round(bbox0[0, 2] - bbox0[0, 0])), int(round(bbox0[0, 3] - bbox0[0, 1]))
        if np.any(np.round(bbox0[:, 2] - bbox0[:, 0]) != w) or np.any(np.round(bbox0[:, 3] - bbox0[:, 1]) != h):
            raise ValueError('render_stack_dev: the blocks of one call share one size (matcher.py:805-822 batches them so)')
        org = np.ascontiguousarray(bbox0[:, :2])
        d_org = _lib.DeviceBuffer.from_array(org)
        d_cand, d_cnt, cnt, cap = self._candidates(d_org, nb, h, w)
        tier, A6 = self._tiers(org, d_org, h, w, d_cand, d_cnt, cnt, cap, precise=precise_mask)
        d_tier, d_A6 = _lib.DeviceBuffer.from_array(tier), _lib.DeviceBuffer.from_array(A6)
        d_ext, d_origin = _lib.DeviceBuffer(16 * nb), _lib.DeviceBuffer(8 * nb)
        d_out, d_mask = _lib.DeviceBuffer(4 * nb * h * w), _lib.DeviceBuffer(nb * h * w)
        im = self.image
        try:
            _lib.check(lib.fb_mesh_render_blocks_dev(ctx, im.buf.ptr, im.dtype, im.shape[0], im.shape[1], im.origin[0], im.origin[1],
                                                     self.d_v0.ptr, self.d_v1.ptr, self.d_tris.ptr, nb, d_org.ptr, h, w, d_tier.ptr,
                                                     d_A6.ptr, cap, d_cand.ptr, d_cnt.ptr, d_ext.ptr, d_origin.ptr, d_out.ptr, d_mask.ptr))
            _lib.check(lib.fb_sync(ctx))
        finally:
            for b in (d_org, d_cand, d_cnt, d_tier, d_A6, d_ext, d_origin):
                b.free()
        return d_out, d_mask, (nb, h, w), tier

    def filter_stack_dev(self, d_out, d_mask, shape, log_sigma, mask_range=None):
        """the log_sigma branch of crop_multiple (renderer.py:632-641) in place of the stack"""
        lib, ctx = _lib.load(), _lib.ctx()
        nb, h, w = shape
        if mask_range is not None:                                      # renderer.py:634-637: grey levels outside the range are masked out
            mr = np.atleast_1d(mask_range)
            _lib.check(lib.fb_mask_range_dev(ctx, d_out.ptr, nb * h * w, float(mr[0]), float(mr[-1]), d_mask.ptr))
        d_f = _lib.DeviceBuffer(4 * nb * h * w)
        _lib.check(lib.fb_dog_masks_dev(ctx, d_out.ptr, 1, nb, h, w, float(log_sigma), d_mask.ptr, 1, d_f.ptr))
        _lib.check(lib.fb_sync(ctx))
        return d_f

    def crop_multiple(self, bboxes, **kwargs):
        """renderer.py:601-648 -> float32 [N, h, w] on the host (None when nothing is covered)"""
        log_sigma = kwargs.get('log_sigma', 0)
        d_out, d_mask, shape, _ = self.render_stack_dev(bboxes, precise_mask=log_sigma > 0)
        try:
            if not d_mask.count_nonzero(int(np.prod(shape))):
                return None
            if log_sigma > 0:
                d_f = self.filter_stack_dev(d_out, d_mask, shape, log_sigma, kwargs.get('mask_range', None))
                try:
                    return d_f.to_array(shape, np.float32)
                finally:
                    d_f.free()
            return d_out.to_array(shape, np.float32)
        finally:
            d_out.free(); d_mask.free()
