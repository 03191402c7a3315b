"""Host mirror of the feabas.common helpers that sit on the NCC path.

``masked_dog_filter`` runs on the GPU (fb_dog); the bounding-box helpers are
integer bookkeeping that stays on the host exactly as in the reference.
"""
from collections import namedtuple

import numpy as np

from . import _lib

Match = namedtuple('Match', ('xy0', 'xy1', 'weight'))       # feabas/common.py:18


def masked_dog_filter(img, sigma, mask=None, signed=True):
    """feabas/common.py:353-377 on the GPU.  img: (N x) H x W uint8/float;
    returns float32 of the same shape."""
    img = np.asarray(img)
    if img.ndim < 2:
        raise ValueError('masked_dog_filter expects at least a 2-D image')
    shp = img.shape
    h, w = shp[-2:]
    n = int(np.prod(shp[:-2])) if img.ndim > 2 else 1
    if img.dtype == np.uint8:
        src, dt = np.ascontiguousarray(img), 0
    else:
        # the reference filters floating input in its own precision; the matcher only feeds float32
        src, dt = np.ascontiguousarray(img, dtype=np.float32), 1
    m = None
    if mask is not None:
        m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
        if m.shape != (h, w):
            raise ValueError('mask must be H x W')
    out = np.empty((n, h, w), dtype=np.float32)
    _lib.check(_lib.load().fb_dog(_lib.ctx(), _lib.ptr(src), dt, n, h, w, float(sigma), _lib.ptr(m),
                                  1 if signed else 0, _lib.ptr(out)))
    return out.reshape(shp)


def half_size(n):
    """output length of cv2.resize(fx=0.5): cvRound(n * 0.5), round half to even"""
    return int(round(n * 0.5))


def area_downsample2(img):
    """cv2.resize(img, None, fx=0.5, fy=0.5, INTER_AREA) of uint8 images (feabas/matcher.py:255-256) on the GPU; odd
    sizes follow the integer-scale area path's edge rule (see fb_dog.hip)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    shp = img.shape
    h, w = shp[-2:]
    n = int(np.prod(shp[:-2])) if img.ndim > 2 else 1
    out = np.empty(shp[:-2] + (half_size(h), half_size(w)), dtype=np.uint8)
    _lib.check(_lib.load().fb_area_downsample2(_lib.ctx(), _lib.ptr(img), n, h, w, _lib.ptr(out)))
    return out


def resize_size(n, f):
    """output length of cv2.resize(None, fx=f): cvRound(n f), round half to even"""
    return int(_lib.load().fb_area_resize_size(int(n), float(f)))


def area_resize(img, fx, fy=None):
    """cv2.resize(img, None, fx=fx, fy=fy, interpolation=cv2.INTER_AREA) of uint8 images (last two axes) for shrinking factors
    (feabas/matcher.py:255-256, 320-321) on the GPU: fb_area_resize (integer 1 / f: cell sums; otherwise fractional-coverage
    taps).  x0.5 is area_downsample2."""
    fy = fx if fy is None else fy
    if not (0 < fx <= 1 and 0 < fy <= 1):
        raise NotImplementedError('area_resize: INTER_AREA is restated for shrinking factors (0 < f <= 1) only')
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if fx == 1 and fy == 1:
        return img
    shp = img.shape
    h, w = shp[-2:]
    n = int(np.prod(shp[:-2])) if img.ndim > 2 else 1
    out = np.empty(shp[:-2] + (resize_size(h, fy), resize_size(w, fx)), dtype=np.uint8)
    if out.shape[-1] == 0 or out.shape[-2] == 0:
        raise ValueError(f'area_resize: a {h} x {w} image shrinks to nothing at ({fy}, {fx})')
    if n == 0:
        return out
    _lib.check(_lib.load().fb_area_resize(_lib.ctx(), _lib.ptr(img), n, h, w, float(fx), float(fy), _lib.ptr(out)))
    return out


def nearest_resize_mask(mask, fx, fy=None):
    """cv2.resize(mask.astype(np.uint8), None, fx=fx, fy=fy, interpolation=cv2.INTER_NEAREST).astype(bool)
    (feabas/matcher.py:257-264): source index min(floor(d / f), n - 1)"""
    mask = np.asarray(mask, dtype=bool)
    fy = fx if fy is None else fy
    H, W = mask.shape
    iy = np.minimum(np.floor(np.arange(resize_size(H, fy)) * (1.0 / fy)).astype(np.int64), H - 1)
    ix = np.minimum(np.floor(np.arange(resize_size(W, fx)) * (1.0 / fx)).astype(np.int64), W - 1)
    return mask[np.ix_(iy, ix)]


def scale_coordinates(xy, scale):
    """feabas/spatial.py:77-86: scaling that keeps the centre of the corner pixel at (0, 0)"""
    xy = np.asarray(xy)
    return xy if np.all(scale == 1) else (xy + 0.5) * scale - 0.5


def numpy_array(obj, copy=False):
    return np.array(obj, copy=True) if copy else np.asarray(obj)


def divide_bbox(bbox, **kwargs):
    """feabas/common.py:380-409 -> (x0, y0, x1, y1) of the blocks, row-major over (y, x).  The cut of each axis is made by the
    C++ that also makes the block grids of fb_match_strips (fb_divide_bbox, csrc/fb_match.hip); needs no GPU."""
    block = kwargs.get('block_size', max(bbox[3] - bbox[1], bbox[2] - bbox[0]))
    least = kwargs.get('min_num_blocks', 1)
    rounded = bool(kwargs.get('round_output', True))
    block_hw = np.array(block if hasattr(block, '__len__') else (block, block), dtype=np.float64)
    least_yx = np.array(least if hasattr(least, '__len__') else (least, least), dtype=np.int32)
    box = np.array(bbox, dtype=np.float64)
    counts, steps = np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.int32)
    lib = _lib.load()
    call = lambda xs, ys: _lib.check(lib.fb_divide_bbox(None, _lib.ptr(box), _lib.ptr(block_hw), _lib.ptr(least_yx), float(kwargs.get('shrink_factor', 1)),
                                                        int(rounded), _lib.ptr(counts), _lib.ptr(steps), _lib.ptr(xs), 0 if xs is None else xs.size,
                                                        _lib.ptr(ys), 0 if ys is None else ys.size))
    call(None, None)                                        # sizing
    xs, ys = np.empty(counts[0]), np.empty(counts[1])
    call(xs, ys)
    if rounded:
        xs, ys = xs.astype(np.int32), ys.astype(np.int32)
    x0 = np.tile(xs, ys.size)
    y0 = np.repeat(ys, xs.size)
    return x0, y0, x0 + int(steps[0]), y0 + int(steps[1])


def intersect_bbox(bbox0, bbox1):
    """feabas/common.py:412-417."""
    lo_x, lo_y = max(bbox0[0], bbox1[0]), max(bbox0[1], bbox1[1])
    hi_x, hi_y = min(bbox0[2], bbox1[2]), min(bbox0[3], bbox1[3])
    return (lo_x, lo_y, hi_x, hi_y), (lo_x < hi_x) and (lo_y < hi_y)


def z_order(indices, base=2):
    """feabas/common.py:196-215."""
    indices = np.asarray(indices)
    ndim = indices.shape[-1]
    rem = indices - indices.min(axis=0)
    digits = np.zeros_like(rem)
    level = 0
    while np.any(rem > 0):
        digits = digits + (rem % base) * (base ** (ndim * level))
        rem = np.floor(rem / base)
        level += 1
    score = np.sum(digits * (base ** np.arange(ndim)), axis=-1)
    return np.argsort(score, kind='stable')


def bbox_centers(bboxes):
    """feabas/common.py:687-690 (pixel-centre convention: -0.5)."""
    b = np.asarray(bboxes, dtype=np.float64).reshape(-1, 4)
    return 0.5 * np.stack((b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]), axis=-1) - 0.5


def bbox_sizes(bboxes):
    """feabas/common.py:693-696: (height, width)."""
    b = np.asarray(bboxes).reshape(-1, 4)
    return np.stack((b[:, 3] - b[:, 1], b[:, 2] - b[:, 0]), axis=-1).clip(0, None)


def cross2d(v0, v1):
    return v0[..., 0] * v1[..., 1] - v0[..., 1] * v1[..., 0]


def signed_area(vertices, triangles):
    """feabas/common.py:672-676.  Whole meshes (float64 vertices, int32 triangles) take the host loop of the library
    (fb_signed_area, the same roundings: no fused multiply-add): the numpy gather costs 16 ms per 500 k triangles, and every
    Link of a section asks for the areas of both its meshes (optimizer.py:26-30)."""
    vertices = np.asarray(vertices); triangles = np.asarray(triangles)
    if (vertices.dtype == np.float64 and triangles.dtype == np.int32 and vertices.ndim == 2 and vertices.shape[1] == 2
            and triangles.ndim == 2 and triangles.shape[1] == 3 and triangles.shape[0] >= 4096):
        from . import _lib
        v = np.ascontiguousarray(vertices); t = np.ascontiguousarray(triangles)
        out = np.empty(t.shape[0])
        if _lib.load().fb_signed_area(None, v.shape[0], _lib.ptr(v), t.shape[0], _lib.ptr(t), _lib.ptr(out)) == 0:
            return out
        raise IndexError('signed_area: a triangle names a vertex outside the vertex list')
    p = vertices[triangles]
    return cross2d(p[:, 1, :] - p[:, 0, :], p[:, 2, :] - p[:, 1, :])


def fit_affine(pts0, pts1, return_rigid=False, weight=None, svd_clip=(1, 1), avoid_flip=True):
    """feabas/spatial.py:21-73: pts0 ~ pts1 @ A (3x3, row vectors); with return_rigid also the transform whose 2x2 part
    has its singular values clipped to svd_clip.  Host-side (the reference keeps it on scipy/numpy too)."""
    pts0 = np.asarray(pts0, dtype=np.float64).reshape(-1, 2)
    pts1 = np.asarray(pts1, dtype=np.float64).reshape(-1, 2)
    assert pts0.shape[0] == pts1.shape[0]
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
