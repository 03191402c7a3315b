"""Oracle (CPU restatement) of the NCC path.  TEST INFRASTRUCTURE -- see
``oracle/__init__.py``; never imported by the product package.

Each function cites the reference lines it restates (paths under
/root/reference/feabas/).  Arithmetic types follow the reference: float32
images, complex64 spectra (pocketfft), float32 correlation surfaces, float64
peak coordinates, float32 confidences.
"""
import numpy as np
from scipy import fft as _fft
from scipy.ndimage import gaussian_filter1d as _g1d

FFT_CONF_NONE = 0     # constant.py:39
FFT_CONF_STD = 1      # constant.py:40
FFT_CONF_MIRROR = 2   # constant.py:41


def next_fast_len(n):
    """Smallest 5-smooth integer >= n (scipy.fftpack.next_fast_len, used at
    matcher.py:6,60-62).  Written out so that the host/C++ side can be checked
    against it without scipy."""
    n = int(n)
    if n <= 6:
        return max(n, 0)
    best = 1 << (n - 1).bit_length()
    p5 = 1
    while p5 < best:
        p35 = p5
        while p35 < best:
            # smallest power of two making p35*2^k >= n
            q = -(-n // p35)
            p2 = 1 << max(0, (q - 1).bit_length())
            cand = p35 * p2
            if cand >= n and cand < best:
                best = cand
            p35 *= 3
        p5 *= 5
    return best


def fft_shape(shp0, shp1, pad):
    """matcher.py:59-62."""
    if pad:
        return tuple(next_fast_len(a + b - 1) for a, b in zip(shp0, shp1))
    return tuple(next_fast_len(max(a, b)) for a, b in zip(shp0, shp1))


def xcorr_fft(img0, img1, conf_mode=FFT_CONF_MIRROR, subpixel=False, pad=True,
              return_surfaces=False, normalize=False, mask0=None, mask1=None):
    """matcher.py:22-135 with sigma=0 (the only way any call site uses it:
    matcher.py:153, 213, 846).  normalize (matcher.py:71-81, 119-122; no call
    site enables it, pinned by golden G20): the correlation surface is divided by
    NC = irfft2(conj(M0) M1) of the masks (all ones by default), scaled by its
    maximum (at least 1) and clipped at 0.1; the mirror surface by the same of
    irfft2(M0 M1).

    img0: (N,H0,W0[,C]) float32, img1: (N,H1,W1[,C]).  Returns dx, dy
    (float64, N), conf (float32, N; float64-derived for STD).
    """
    img0 = np.asarray(img0)
    img1 = np.asarray(img1)
    if img0.ndim > 3:                       # matcher.py:50-53
        img0 = np.moveaxis(img0, -1, 1)
    if img1.ndim > 3:
        img1 = np.moveaxis(img1, -1, 1)
    h0, w0 = img0.shape[-2:]
    h1, w1 = img1.shape[-2:]
    fh, fw = fft_shape((h0, w0), (h1, w1), pad)
    F0 = _fft.rfft2(img0, s=(fh, fw), axes=(-2, -1))     # :63
    F1 = _fft.rfft2(img1, s=(fh, fw), axes=(-2, -1))     # :64
    P = np.conj(F0) * F1                                   # :65
    if P.ndim > 3:
        P = P.mean(axis=1)                                 # :66-67
    C = _fft.irfft2(P, s=(fh, fw), axes=(-2, -1))         # :68
    n = C.shape[0]
    Cf = C.reshape(n, -1)
    if normalize:                                          # :70-81
        m0 = np.ones((h0, w0), dtype=img0.dtype) if mask0 is None else np.asarray(mask0)
        m1 = np.ones((h1, w1), dtype=img1.dtype) if mask1 is None else np.asarray(mask1)
        M0 = _fft.rfft2(m0, s=(fh, fw))
        M1 = _fft.rfft2(m1, s=(fh, fw))
        NC = _fft.irfft2(np.conj(M0) * M1, s=(fh, fw)).reshape(-1, fh * fw)
        NC = (NC / (NC.max(axis=-1, keepdims=True).clip(1, None))).clip(0.1, None)
        Cf = Cf / NC
    k = np.argmax(Cf, axis=-1)                             # :82 first max, row-major
    py = k // fw
    px = k % fw
    dy = py.astype(np.float64)
    dx = px.astype(np.float64)
    if subpixel:                                           # :84-106
        oy = np.array([-1, -1, -1, 0, 0, 0, 1, 1, 1])
        ox = np.array([-1, 0, 1, -1, 0, 1, -1, 0, 1])
        cy = (py[:, None] + oy[None, :]) % fh
        cx = (px[:, None] + ox[None, :]) % fw
        Ct = Cf[np.arange(n)[:, None], cy * fw + cx]
        gx = (Ct[:, 5] - Ct[:, 3]) / 2
        gy = (Ct[:, 7] - Ct[:, 1]) / 2
        hxx = Ct[:, 3] + Ct[:, 5] - 2 * Ct[:, 4]
        hyy = Ct[:, 7] + Ct[:, 1] - 2 * Ct[:, 4]
        hxy = (Ct[:, 0] + Ct[:, 8] - Ct[:, 2] - Ct[:, 6]) / 4
        det = hxx * hyy - hxy * hxy
        sx = np.zeros(n, dtype=np.float32)
        sy = np.zeros(n, dtype=np.float32)
        ok = det > 0
        ixx = hyy[ok] / det[ok]
        ixy = -hxy[ok] / det[ok]
        iyy = hxx[ok] / det[ok]
        sx[ok] = -ixx * gx[ok] - ixy * gy[ok]
        sy[ok] = -ixy * gx[ok] - iyy * gy[ok]
        dx = px + sx.clip(-0.5, 0.5)
        dy = py + sy.clip(-0.5, 0.5)
    dy = dy + (h0 - h1) / 2                                # :107
    dx = dx + (w0 - w1) / 2                                # :108
    dy = dy - np.round(dy / fh) * fh                       # :109 (half-to-even)
    dx = dx - np.round(dx / fw) * fw                       # :110
    Cm = None
    if conf_mode == FFT_CONF_NONE:                         # :111-112
        conf = np.ones(n, dtype=np.float32)
    elif conf_mode == FFT_CONF_MIRROR:                     # :113-128
        Q = F0 * F1
        if Q.ndim > 3:
            Q = Q.mean(axis=1)
        Cm = np.abs(_fft.irfft2(Q, s=(fh, fw), axes=(-2, -1))).reshape(n, -1)
        if normalize:                                      # :119-122
            NCm = _fft.irfft2(M0 * M1, s=(fh, fw)).reshape(-1, fh * fw)
            NCm = (NCm / (NCm.max(axis=-1, keepdims=True).clip(1, None))).clip(0.1, None)
            Cm = Cm / NCm
        mx = Cf.max(axis=-1)
        mm = Cm.max(axis=-1)
        conf = np.zeros(n, dtype=np.float32)
        pos = mx > 0
        conf[pos] = 1 - mm[pos] / mx[pos]
        conf = conf.clip(0, 1)
    elif conf_mode == FFT_CONF_STD:                        # :129-134
        sd = Cf.std(axis=-1)
        mx = Cf.max(axis=-1)
        with np.errstate(divide='ignore', invalid='ignore'):
            conf = (1 - np.exp(-mx / sd)) ** (fh * fw)
        conf = conf.clip(0, 1)
    else:
        raise ValueError(conf_mode)
    if return_surfaces:
        return dx, dy, conf, Cf.reshape(n, fh, fw), (None if Cm is None else Cm.reshape(n, fh, fw))
    return dx, dy, conf


def masked_dog_filter(img, sigma, mask=None, signed=True):
    """common.py:353-377.  G(img) - G(G(img)) with scipy gaussian_filter1d,
    mode='nearest', truncate 4 sigma; masked halo suppression."""
    img = np.asarray(img)
    if not np.issubdtype(img.dtype, np.floating):
        img = img.astype(np.float32)                       # :363-364
    g0 = _g1d(_g1d(img, sigma, axis=-1, mode='nearest'), sigma, axis=-2, mode='nearest')
    g1 = _g1d(_g1d(g0, sigma, axis=-1, mode='nearest'), sigma, axis=-2, mode='nearest')
    out = g0 - g1
    if (mask is not None) and (not np.all(mask)):          # :368-374
        halo_src = np.ptp(img) * (mask == 0)
        sc = (2 * sigma * sigma) ** 0.5
        halo = _g1d(_g1d(halo_src, sc, axis=-1, mode='nearest'), sc, axis=-2, mode='nearest') * (sc ** 2) / (sigma ** 2)
        mag = (np.abs(out) - halo).clip(0, None)
        out = mag * np.sign(out)
    if not signed:
        out = np.abs(out)
    return out


def gaussian_taps(sigma, truncate=4.0):
    """The FIR taps scipy.ndimage.gaussian_filter1d builds (float64)."""
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1, dtype=np.float64)
    w = np.exp(-0.5 / (float(sigma) * float(sigma)) * x * x)
    return w / w.sum()


def half_size(n):
    """output length of cv2.resize(fx=0.5) for an input length n: cvRound(n * 0.5), round half to even"""
    return int(round(n * 0.5))


def area_downsample2(img):
    """cv2.resize(img, None, fx=0.5, fy=0.5, INTER_AREA) of a uint8 image (matcher.py:255-256): the integer-scale
    area path (resizeAreaFast).  Output size cvRound(n / 2) per axis (round half to even); a full 2x2 cell is
    (sum + 2) >> 2 (the vector kernel's rounding, used everywhere here); a cell cut by the image edge -- odd sizes whose
    half rounds up -- averages the pixels that exist, float(sum) / count rounded half to even; an odd size whose half
    rounds down drops the last row / column.  cv2 is absent from the build container, so this step is "parity
    unpinned" (SURVEY.md A.4)."""
    img = np.asarray(img)
    h, w = img.shape[-2:]
    ho, wo = half_size(h), half_size(w)
    v = np.zeros(img.shape[:-2] + (2 * ho, 2 * wo), dtype=np.uint16)
    cnt = np.zeros((2 * ho, 2 * wo), dtype=np.uint16)
    hh, ww = min(h, 2 * ho), min(w, 2 * wo)
    v[..., :hh, :ww] = img[..., :hh, :ww]
    cnt[:hh, :ww] = 1
    s = v[..., 0::2, 0::2] + v[..., 0::2, 1::2] + v[..., 1::2, 0::2] + v[..., 1::2, 1::2]
    c = cnt[0::2, 0::2] + cnt[0::2, 1::2] + cnt[1::2, 0::2] + cnt[1::2, 1::2]
    full = (s + 2) >> 2
    part = np.rint(s.astype(np.float32) / np.maximum(c, 1).astype(np.float32))
    return np.where(c == 4, full, part).astype(np.uint8)


def area_resize(img, fx, fy=None):
    """cv2.resize(img, None, fx=fx, fy=fy, interpolation=cv2.INTER_AREA) of a uint8 image for shrinking factors
    (matcher.py:255-256, 320-321), restated from OpenCV's imgproc/resize.cpp; UNPINNED (cv2 is absent here).
    Output size cvRound(n f) per axis.  Integer 1 / f on both axes (resizeAreaFast): integer cell sums times float32
    1 / k^2, rounded half to even (k = 2: (sum + 2) >> 2, the vector kernel); a cell cut by the image edge is
    float32(sum) / count.  Otherwise (computeResizeAreaTab + ResizeArea): per axis every source pixel weighs by the
    fraction of it inside the cell over the cell width (float32), a row is accumulated tap after tap in float32, then the
    rows."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    fy = fx if fy is None else fy
    H, W = img.shape
    Ho, Wo = int(np.rint(H * fy)), int(np.rint(W * fx))
    sx, sy = 1.0 / fx, 1.0 / fy
    kx, ky = int(np.rint(sx)), int(np.rint(sy))
    eps = np.finfo(np.float64).eps
    out = np.zeros((Ho, Wo), dtype=np.uint8)
    sat = lambda v: np.clip(np.rint(v), 0, 255).astype(np.uint8)
    if abs(sx - kx) < eps and abs(sy - ky) < eps:
        scale = np.float32(1.0) / np.float32(kx * ky)
        wfull = W // kx
        for dy in range(Ho):
            y0 = dy * ky
            if y0 >= H:
                continue
            rows = img[y0:min(y0 + ky, H)].astype(np.int64)
            for dx in range(Wo):
                x0 = dx * kx
                if x0 >= W:
                    continue
                cell = rows[:, x0:min(x0 + kx, W)]
                s_ = int(cell.sum())
                if y0 + ky <= H and dx < wfull:
                    out[dy, dx] = (s_ + 2) >> 2 if (kx == 2 and ky == 2) else sat(np.float32(s_) * scale)
                else:
                    out[dy, dx] = sat(np.float32(s_) / np.float32(cell.size))
        return out

    def taps(ssize, dsize, scale):
        tab = []
        for d in range(dsize):
            f1 = d * scale; f2 = f1 + scale
            cell = min(scale, ssize - f1)
            s1, s2 = int(np.ceil(f1)), int(np.floor(f2))
            s2 = min(s2, ssize - 1); s1 = min(s1, s2)
            row = []
            if s1 - f1 > 1e-3:
                row.append((s1 - 1, np.float32((s1 - f1) / cell)))
            for s_ in range(s1, s2):
                row.append((s_, np.float32(1.0 / cell)))
            if f2 - s2 > 1e-3:
                row.append((s2, np.float32(min(min(f2 - s2, 1.0), cell) / cell)))
            tab.append(row)
        return tab
    xt, yt = taps(W, Wo, sx), taps(H, Ho, sy)
    src = img.astype(np.float32)
    # rows first: buf[sy, dx] = sum_k S[sy, si_k] * alpha_k accumulated in float32 in tap order
    buf = np.zeros((H, Wo), dtype=np.float32)
    for dx, row in enumerate(xt):
        acc = np.zeros(H, dtype=np.float32)
        for si, a in row:
            acc = acc + src[:, si] * a
        buf[:, dx] = acc
    for dy, col in enumerate(yt):
        acc = None
        for si, b in col:
            term = b * buf[si]
            acc = term if acc is None else acc + term
        out[dy] = sat(acc if acc is not None else np.zeros(Wo, np.float32))
    return out


def nearest_resize_mask(mask, fx, fy=None):
    """cv2.resize(mask.astype(uint8), None, fx, fy, INTER_NEAREST).astype(bool) (matcher.py:257-264): source index
    min(floor(d / f), n - 1); UNPINNED."""
    mask = np.asarray(mask, dtype=bool)
    fy = fx if fy is None else fy
    H, W = mask.shape
    Ho, Wo = int(np.rint(H * fy)), int(np.rint(W * fx))
    iy = np.minimum(np.floor(np.arange(Ho) * (1.0 / fy)).astype(np.int64), H - 1)
    ix = np.minimum(np.floor(np.arange(Wo) * (1.0 / fx)).astype(np.int64), W - 1)
    return mask[np.ix_(iy, ix)]


def scale_coordinates(xy, s):
    """spatial.scale_coordinates (spatial.py): pixel centres keep their meaning, (xy + 0.5) s - 0.5"""
    return (np.asarray(xy, dtype=np.float64) + 0.5) * s - 0.5


# ------------------------------------------------------------------ bbox helpers
def divide_bbox(bbox, block_size=None, min_num_blocks=1, round_output=True, shrink_factor=1):
    """common.py:380-409."""
    xmin, ymin, xmax, ymax = bbox
    ht = ymax - ymin
    wd = xmax - xmin
    if block_size is None:
        block_size = max(ht, wd)
    if not hasattr(block_size, '__len__'):
        block_size = (block_size, block_size)
    if not hasattr(min_num_blocks, '__len__'):
        min_num_blocks = (min_num_blocks, min_num_blocks)
    nx = max(np.ceil(wd / block_size[1]), min_num_blocks[1])
    ny = max(np.ceil(ht / block_size[0]), min_num_blocks[0])
    dx = int(np.ceil(wd / nx))
    dy = int(np.ceil(ht / ny))
    xt = np.linspace(xmin, xmax - dx, num=int(nx), endpoint=True)
    yt = np.linspace(ymin, ymax - dy, num=int(ny), endpoint=True)
    if shrink_factor != 1:
        dxn = dx * shrink_factor
        dyn = dy * shrink_factor
        xt = xt + (dx - dxn) / 2
        yt = yt + (dy - dyn) / 2
        dx = int(np.ceil(dxn))
        dy = int(np.ceil(dyn))
    if round_output:
        xt = np.round(xt).astype(np.int32)
        yt = np.round(yt).astype(np.int32)
    xx, yy = np.meshgrid(xt, yt)
    return xx.ravel(), yy.ravel(), xx.ravel() + dx, yy.ravel() + dy


def intersect_bbox(b0, b1):
    """common.py:412-417."""
    xmin = max(b0[0], b1[0]); ymin = max(b0[1], b1[1])
    xmax = min(b0[2], b1[2]); ymax = min(b0[3], b1[3])
    return (xmin, ymin, xmax, ymax), (xmin < xmax) and (ymin < ymax)


def z_order(indices, base=2):
    """common.py:196-215: stable argsort of the interleaved-digit score."""
    indices = np.asarray(indices)
    ndim = indices.shape[-1]
    idx = indices - indices.min(axis=0)
    score = np.zeros_like(idx)
    pw = 0
    while np.any(idx > 0):
        score = score + (idx % base) * (base ** (ndim * pw))
        idx = np.floor(idx / base)
        pw += 1
    z = np.sum(score * (base ** np.arange(ndim)), axis=-1)
    return np.argsort(z, kind='stable')


def bbox_centers(bboxes):
    """common.py:687-690."""
    b = np.asarray(bboxes, dtype=np.float64).reshape(-1, 4)
    return np.stack((0.5 * (b[:, 0] + b[:, 2]) - 0.5, 0.5 * (b[:, 1] + b[:, 3]) - 0.5), axis=-1)


def bbox_sizes(bboxes):
    """common.py:693-696: (height, width) per bbox, clipped at 0."""
    b = np.asarray(bboxes).reshape(-1, 4)
    return np.stack((b[:, 3] - b[:, 1], b[:, 2] - b[:, 0]), axis=-1).clip(0, None)


def distributor_cartesian_bbox(bbox0, bbox1, spacing, min_num_blocks=1, shrink_factor=1, zorder=True):
    """matcher.py:865-891 on the two mesh bounding boxes (MOVING gear)."""
    if not hasattr(shrink_factor, '__len__'):
        shrink_factor = (shrink_factor, shrink_factor)
    bbox, valid = intersect_bbox(bbox0, bbox1)
    if not valid:
        return None, None
    bb0 = np.stack(divide_bbox(bbox, block_size=spacing, min_num_blocks=min_num_blocks,
                               shrink_factor=shrink_factor[0]), axis=-1)
    bb1 = np.stack(divide_bbox(bbox, block_size=spacing, min_num_blocks=min_num_blocks,
                               shrink_factor=shrink_factor[1]), axis=-1)
    if zorder:
        xs = bb0[:, 0]; ys = bb0[:, 1]
        xr = np.round((xs - xs.min()) / spacing)
        yr = np.round((ys - ys.min()) / spacing)
        idx = z_order(np.stack((xr, yr), axis=-1))
        bb0 = bb0[idx]; bb1 = bb1[idx]
    return bb0, bb1


def global_translation_matcher(img0, img1, conf_mode=FFT_CONF_MIRROR, conf_thresh=0.3, divide_factor=6):
    """matcher.py:138-221 with sigma=0 (the DoG is applied by the caller,
    matcher.py:273-275)."""
    ht0, wd0 = img0.shape[-2:]
    ht1, wd1 = img1.shape[-2:]
    tx, ty, conf = xcorr_fft(img0[None], img1[None], conf_mode=conf_mode, pad=True)
    tx, ty, conf = tx.item(), ty.item(), conf.item()
    tx = tx + (wd1 - wd0) / 2                              # :155
    ty = ty + (ht1 - ht0) / 2                              # :156
    if conf > conf_thresh:
        return tx, ty, conf
    shp = np.minimum((ht0, wd0), (ht1, wd1))
    if hasattr(divide_factor, '__len__'):
        div_n = tuple(divide_factor[:2])
    else:                                                  # :165-177
        r0 = shp[0] / shp[1]
        best = np.inf
        div_n = None
        for f in range(1, int(divide_factor ** 0.5) + 1):
            if divide_factor % f != 0:
                continue
            rr = abs(np.log(r0 * (f ** 2 / divide_factor)))
            if rr < best:
                best = rr
                div_n = (int(divide_factor / f), int(f))
            rr = abs(np.log(r0 / (f ** 2 / divide_factor)))
            if rr < best:
                best = rr
                div_n = (int(f), int(divide_factor / f))
    x0a, y0a, x0b, y0b = divide_bbox((0, 0, wd0, ht0), min_num_blocks=div_n)
    x1a, y1a, x1b, y1b = divide_bbox((0, 0, wd1, ht1), min_num_blocks=div_n)
    st0, st1, offx, offy = [], [], [], []
    for k in range(x0a.size):                              # :184-210
        wb = max(x0b[k] - x0a[k], x1b[k] - x1a[k])
        hb = max(y0b[k] - y0a[k], y1b[k] - y1a[k])

        def _span(lo, hi, full, limit):
            p = int(np.ceil((full - (hi - lo)) / 2))
            pt = np.array((lo - p, hi + p))
            return (pt - min(pt[0], 0) - max(pt[1] - limit, 0)).clip(0, limit)
        yp0 = _span(y0a[k], y0b[k], hb, ht0)
        xp0 = _span(x0a[k], x0b[k], wb, wd0)
        b0 = img0[yp0[0]:yp0[1], xp0[0]:xp0[1]]
        if np.ptp(b0) == 0:
            continue
        yp1 = _span(y1a[k], y1b[k], hb, ht1)
        xp1 = _span(x1a[k], x1b[k], wb, wd1)
        b1 = img1[yp1[0]:yp1[1], xp1[0]:xp1[1]]
        if np.ptp(b1) == 0:
            continue
        st0.append(b0); st1.append(b1)
        offx.append((np.ptp(xp1) - np.ptp(xp0)) / 2 + xp1[0] - xp0[0])
        offy.append((np.ptp(yp1) - np.ptp(yp0)) / 2 + yp1[0] - yp0[0])
    if not st0:
        return tx, ty, conf
    btx, bty, bconf = xcorr_fft(np.stack(st0), np.stack(st1), conf_mode=conf_mode, pad=True)
    btx = btx + np.array(offx)
    bty = bty + np.array(offy)
    kb = int(np.argmax(bconf))
    if bconf[kb] >= conf:
        tx, ty, conf = btx[kb], bty[kb], bconf[kb]
    return tx, ty, conf


def block_points(bboxes0, bboxes1, dx, dy):
    """matcher.py:840-849: block displacement -> matched point pair."""
    c0 = bbox_centers(bboxes0)
    c1 = bbox_centers(bboxes1)
    s0 = bbox_sizes(bboxes0).astype(np.float64)
    s1 = bbox_sizes(bboxes1).astype(np.float64)
    r = (s0 / (s0 + s1))[:, ::-1]
    dxy = np.stack((dx, dy), axis=-1)
    return c0 - dxy * r, c1 + dxy * (1 - r)


def auto_spacings(shape0, shape1):
    """matcher.py:243-251."""
    shp = np.minimum(shape0, shape1)
    smx = max(shp) * 0.25
    smn = max(min(75, min(shp) / 3), 25)
    if smn > smx:
        return np.array([smn])
    nsp = max(1, round(np.log(smx / smn) / np.log(4)))
    return np.exp(np.linspace(np.log(smn), np.log(smx), num=nsp, endpoint=True))


# ------------------------------------------------------------------ affine patch gather (UNPINNED: cv2 is absent)
def remap_bilinear_cv(img, map_x, map_y):
    """cv2.remap(img, map_x, map_y, INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0) for a float32
    single-channel image, as common.remap calls it (common.py:218-255).  UNPINNED restatement of OpenCV's
    remapBilinear: the float32 maps are quantised to 1/32 px (INTER_BITS = 5, cvRound = round half to even), the four
    taps are blended with float32 table weights, taps outside the image are the border value."""
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape
    mx = np.asarray(map_x, dtype=np.float32)
    my = np.asarray(map_y, dtype=np.float32)
    sx = np.rint(mx * np.float32(32)).astype(np.int64)
    sy = np.rint(my * np.float32(32)).astype(np.int64)
    ix, iy = sx >> 5, sy >> 5
    ax = (sx & 31).astype(np.float32) * np.float32(1 / 32)
    ay = (sy & 31).astype(np.float32) * np.float32(1 / 32)
    one = np.float32(1)
    w00 = (one - ay) * (one - ax); w01 = (one - ay) * ax; w10 = ay * (one - ax); w11 = ay * ax

    def tap(yy, xx):
        inside = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(inside, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(0))
    return ((tap(iy, ix) * w00 + tap(iy, ix + 1) * w01) + tap(iy + 1, ix) * w10) + tap(iy + 1, ix + 1) * w11


def crop_affine(img, x0, y0, h, w, A, t, origin):
    """MeshRenderer.crop_field_affine (renderer.py:419-435) + render_by_subregions (common.py:311-330) for one block:
    output pixel (i, j) samples `img` at (X A[0,0] + Y A[1,0] + t[0], X A[0,1] + Y A[1,1] + t[1]), X = x0 + i, Y = y0 + j;
    `origin` = (xmin, ymin), the integer origin of the sub-image handed to cv2.remap (maps are float32 relative to it)."""
    xs = np.linspace(x0, x0 + w, num=w, endpoint=False, dtype=float)
    ys = np.linspace(y0, y0 + h, num=h, endpoint=False, dtype=float)
    xx, yy = np.meshgrid(xs, ys)
    x_field = xx * A[0, 0] + yy * A[1, 0] + t[0]
    y_field = xx * A[0, 1] + yy * A[1, 1] + t[1]
    mxt = (x_field - origin[0]).astype(np.float32)
    myt = (y_field - origin[1]).astype(np.float32)
    H, W = img.shape
    # the sub-image starts at `origin`: evaluate on the full image with maps shifted back by the integer origin
    sx = np.rint(mxt * np.float32(32)).astype(np.int64); sy = np.rint(myt * np.float32(32)).astype(np.int64)
    ix = (sx >> 5) + int(origin[0]); iy = (sy >> 5) + int(origin[1])
    ax = (sx & 31).astype(np.float32) * np.float32(1 / 32); ay = (sy & 31).astype(np.float32) * np.float32(1 / 32)
    one = np.float32(1)
    w00 = (one - ay) * (one - ax); w01 = (one - ay) * ax; w10 = ay * (one - ax); w11 = ay * ax
    img = np.asarray(img, dtype=np.float32)

    def tap(yy_, xx_):
        inside = (yy_ >= 0) & (yy_ < H) & (xx_ >= 0) & (xx_ < W)
        return np.where(inside, img[np.clip(yy_, 0, H - 1), np.clip(xx_, 0, W - 1)], np.float32(0))
    return ((tap(iy, ix) * w00 + tap(iy, ix + 1) * w01) + tap(iy + 1, ix) * w10) + tap(iy + 1, ix + 1) * w11
