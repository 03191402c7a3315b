"""GPU parity of the NCC path through the C-ABI: golden vectors from the
reference, seeded comparisons with the oracle, edge cases.  Bar: integer peaks
bit-exact, sub-pixel offsets and confidence within 1e-4 (north_star)."""
import numpy as np
import pytest
from scipy.ndimage import gaussian_filter

from conftest import load_golden
from oracle import ncc_ref

pytestmark = pytest.mark.gpu

ATOL = 1e-4


def _check(got, exp, atol=ATOL, conf_atol=None):
    dx, dy, cf = got
    ex, ey, ec = exp
    np.testing.assert_array_equal(np.round(dx), np.round(ex))
    np.testing.assert_array_equal(np.round(dy), np.round(ey))
    np.testing.assert_allclose(dx, ex, atol=atol, rtol=0)
    np.testing.assert_allclose(dy, ey, atol=atol, rtol=0)
    np.testing.assert_allclose(cf, ec, atol=atol if conf_atol is None else conf_atol, rtol=0)


@pytest.mark.parametrize('case', ['A', 'B', 'C'])
@pytest.mark.parametrize('pad', [1, 0])
@pytest.mark.parametrize('sub', [1, 0])
@pytest.mark.parametrize('cm', [0, 1, 2])
def test_xcorr_golden(fb, case, pad, sub, cm):
    g = load_golden('g1_xcorr.npz')
    got = fb.matcher.xcorr_fft(g[f'{case}_img0'], g[f'{case}_img1'], conf_mode=cm, pad=bool(pad), subpixel=bool(sub))
    key = f'{case}_p{pad}_s{sub}_c{cm}'
    catol = None
    if cm == 1:
        # FFT_CONF_STD = (1 - exp(-Cmax/Cstd)) ** (Fh*Fw) with the base rounded to float32 by numpy
        # (matcher.py:133): one float32 ulp of the base moves the result by Fh*Fw*6e-8, which is the
        # reference's own resolution.  No reference call site uses this mode.
        f = {'A': 150 * 150, 'B': 144 * 150, 'C': 256 * 256}[case] if pad else {'A': 75 * 75, 'B': 75 * 75, 'C': 128 * 128}[case]
        catol = 2.5 * f * 6e-8
    _check(got, (g[key + '_dx'], g[key + '_dy'], g[key + '_conf']), conf_atol=catol)


@pytest.mark.parametrize('pad', [1, 0])
def test_xcorr_golden_channels(fb, pad):
    g = load_golden('g1_xcorr.npz')
    got = fb.matcher.xcorr_fft(g['D_img0'], g['D_img1'], conf_mode=2, pad=bool(pad), subpixel=True)
    _check(got, (g[f'D_p{pad}_s1_c2_dx'], g[f'D_p{pad}_s1_c2_dy'], g[f'D_p{pad}_s1_c2_conf']))


def _pairs(rng, n, s0, s1, maxshift):
    H = max(s0[0], s1[0]); W = max(s0[1], s1[1])
    big = gaussian_filter(rng.standard_normal((3 * H + 64, 3 * W + 64)), 1.5).astype(np.float32)
    big -= gaussian_filter(big, 4.0)
    i0 = np.empty((n,) + s0, np.float32); i1 = np.empty((n,) + s1, np.float32)
    for k in range(n):
        y = rng.integers(H, 2 * H); x = rng.integers(W, 2 * W)
        sy, sx = rng.integers(-maxshift, maxshift + 1, 2)
        i0[k] = big[y:y + s0[0], x:x + s0[1]]
        i1[k] = big[y + sy:y + sy + s1[0], x + sx:x + sx + s1[1]]
    i1 += 0.05 * i1.std() * rng.standard_normal(i1.shape).astype(np.float32)
    return i0, i1


@pytest.mark.parametrize('shape,n,pad', [
    (((75, 73), (75, 73)), 385, False),     # the 4k-pair fine class, unpadded round: 75 x 75 (register-resident kernel, fb_ncc_pfa.hip)
    (((75, 71), (75, 71)), 200, False),     # ... after a translation of more than 5 px: 75 x 72
    (((72, 75), (72, 75)), 200, False),     # ... of an up-down pair: 72 x 75
    (((71, 70), (71, 70)), 120, False),     # 72 x 72
    (((75, 70), (73, 72)), 60, False),      # unequal blocks, 75 x 72
    (((75, 73), (75, 73)), 64, True),       # padded 150 x 150
    (((74, 72), (67, 75)), 33, True),       # README config, unequal blocks
    (((280, 280), (280, 280)), 16, True),   # alignment spacing 400
    (((70, 70), (70, 70)), 100, True),      # alignment spacing 100
    (((1, 40), (1, 40)), 3, True),          # degenerate one-row images
])
def test_xcorr_vs_oracle(fb, shape, n, pad):
    rng = np.random.default_rng(hash((shape, n, pad)) % (2 ** 32))
    i0, i1 = _pairs(rng, n, shape[0], shape[1], maxshift=max(1, min(shape[0]) // 4)) if shape[0][0] > 1 else (
        rng.standard_normal((n,) + shape[0]).astype(np.float32), rng.standard_normal((n,) + shape[1]).astype(np.float32))
    for sub in (True, False):
        got = fb.matcher.xcorr_fft(i0, i1, pad=pad, subpixel=sub)
        exp = ncc_ref.xcorr_fft(i0, i1, pad=pad, subpixel=sub)
        _check(got, exp)


@pytest.mark.parametrize('s0,s1,n,pad', [
    ((280, 280), (280, 280), 6, True),      # alignment spacing 400 x 0.7, padded: 576 x 576 = (16 x 9 x 4)^2
    ((70, 70), (70, 70), 40, True),         # alignment spacing 100 x 0.7, padded: 144 x 144 (too large for the on-chip class)
    ((280, 280), (280, 280), 5, False),     # the same blocks in a round without padding: 288 x 288
    ((74, 72), (67, 75), 33, True),         # README grid: 150 x 144 runs at 160 x 144
    ((67, 75), (74, 72), 9, True),          # ... and 144 x 160
    ((300, 90), (280, 100), 7, True),       # unequal blocks, 579 x 189 -> 600 x 192 runs at 640 x 192
    ((400, 130), (400, 130), 3, True),      # 799 x 259 -> 800 x 270 runs at 1024 x 288 (a power of two beside a 9 x 2^a)
])
def test_xcorr_compile_time_mixed_radix_class(fb, s0, s1, n, pad):
    """the block classes of the alignment matcher (configs/default_alignment_configs.yaml:16-23, matcher.py:59-62) and of the
    README grid run on the compile-time mixed-radix streaming kernels (fb_ncc_ct.hip), padded axes at the next length with
    such a plan: integer peaks bit-exact, sub-pixel offsets and confidences within the bars of _check, and the same
    answers as the run-time mixed-radix kernels at the reference's own FFT size"""
    import os
    rng = np.random.default_rng(s0[0] * 7 + s1[1] + n)
    i0, i1 = _pairs(rng, n, s0, s1, maxshift=max(1, min(s0 + s1) // 4))
    for sub in (True, False):
        exp = ncc_ref.xcorr_fft(i0, i1, pad=pad, subpixel=sub)
        got = fb.matcher.xcorr_fft(i0, i1, pad=pad, subpixel=sub)
        _check(got, exp)
        os.environ['FEABAS_HIP_FFT_GENERIC'] = '1'; os.environ['FEABAS_HIP_FFT_EXACT'] = '1'
        try:
            gen = fb.matcher.xcorr_fft(i0, i1, pad=pad, subpixel=sub)
        finally:
            del os.environ['FEABAS_HIP_FFT_GENERIC'], os.environ['FEABAS_HIP_FFT_EXACT']
        np.testing.assert_array_equal(np.round(got[0]), np.round(gen[0])); np.testing.assert_array_equal(np.round(got[1]), np.round(gen[1]))
        np.testing.assert_allclose(got[2], gen[2], atol=2e-5)


@pytest.mark.parametrize('shape,pad', [((75, 73), False), ((75, 72), False), ((70, 75), False), ((72, 71), False), ((75, 73), True), ((70, 70), True), ((256, 120), True), ((150, 150), False), ((280, 280), True)])
def test_xcorr_one_image_almost_blank(fb, shape, pad):
    """a block one side of which is almost blank (a masked or saturated region, a window that only grazes the texture): the
    reference transforms the two images apart (matcher.py:63-64) and its answer does not depend on the magnitude of either;
    the device packs them into ONE complex transform, where the weak image used to drown in the rounding of the strong one
    (its confidence became noise that changed with the FFT length).  With pack_scales (fb_ldsfft.h) every class -- on-chip,
    power-of-two, compile-time mixed radix, run-time mixed radix -- gives the oracle's peak, offset and confidence with one
    image scaled by 1e-6 or 1e+5, and the same answer as for the unscaled pair"""
    rng = np.random.default_rng(shape[0] + 7 * int(pad))
    i0, i1 = _pairs(rng, 6, shape, shape, maxshift=max(1, min(shape) // 5))
    base = fb.matcher.xcorr_fft(i0, i1, pad=pad, subpixel=True)
    for s0, s1 in ((1.0, 1e-6), (1e-6, 1.0), (1e5, 1.0), (1.0, 3e-3)):
        a, b = (i0 * np.float32(s0)).astype(np.float32), (i1 * np.float32(s1)).astype(np.float32)
        got = fb.matcher.xcorr_fft(a, b, pad=pad, subpixel=True)
        exp = ncc_ref.xcorr_fft(a, b, pad=pad, subpixel=True)
        _check(got, exp)
        np.testing.assert_array_equal(np.round(got[0]), np.round(base[0])); np.testing.assert_array_equal(np.round(got[1]), np.round(base[1]))
        np.testing.assert_allclose(got[0], base[0], atol=2e-4); np.testing.assert_allclose(got[1], base[1], atol=2e-4)
        np.testing.assert_allclose(got[2], base[2], atol=1e-4)
    # one image exactly zero (a window inside a masked region): zero surface, first index, confidence 0 -- not the noise of
    # the other image's rounding
    z = np.zeros_like(i1)
    z[3, :shape[0] // 2] = i1[3, :shape[0] // 2]                       # ... also when only some rows of the stack's blocks are blank
    for a, b in ((i0, z), (z, i1)):
        got = fb.matcher.xcorr_fft(a, b, pad=pad, subpixel=True)
        exp = ncc_ref.xcorr_fft(a, b, pad=pad, subpixel=True)
        _check(got, exp)
        assert np.all(got[2][[0, 1, 2, 4, 5]] == 0)


def test_xcorr_streaming_class(fb):
    """coarse classes of the 4k tile pair at reduced count: 1024x510 blocks (FFT 2048x1024)
    and the 2048x255 global strip (FFT 4096x512)."""
    rng = np.random.default_rng(5)
    i0, i1 = _pairs(rng, 2, (1024, 510), (1024, 510), maxshift=20)
    _check(fb.matcher.xcorr_fft(i0, i1, pad=True), ncc_ref.xcorr_fft(i0, i1, pad=True))
    j0, j1 = _pairs(rng, 1, (2048, 255), (2048, 255), maxshift=12)
    _check(fb.matcher.xcorr_fft(j0, j1, pad=True, subpixel=True), ncc_ref.xcorr_fft(j0, j1, pad=True, subpixel=True))


@pytest.mark.parametrize('shape', [(250, 247), (500, 245), (486, 120)])
def test_xcorr_linear_shapes_promoted_to_pow2(fb, shape):
    """padded (linear) correlations whose 5-smooth FFT size sits just below a power of two run at the power of two on
    the device (500 -> 512, 1000 -> 1024, 972 -> 1024): same lags, same peak, same confidences as the reference size"""
    import os
    rng = np.random.default_rng(shape[0])
    i0, i1 = _pairs(rng, 3, shape, shape, maxshift=15)
    for sub in (True, False):
        exp = ncc_ref.xcorr_fft(i0, i1, pad=True, subpixel=sub)
        got = fb.matcher.xcorr_fft(i0, i1, pad=True, subpixel=sub)
        _check(got, exp)
        os.environ['FEABAS_HIP_FFT_EXACT'] = '1'          # the reference's own FFT size on the generic mixed-radix core
        try:
            ref_size = fb.matcher.xcorr_fft(i0, i1, pad=True, subpixel=sub)
        finally:
            del os.environ['FEABAS_HIP_FFT_EXACT']
        _check(ref_size, exp)
        np.testing.assert_array_equal(np.round(got[0]), np.round(ref_size[0]))
        np.testing.assert_allclose(got[2], ref_size[2], atol=2e-5)


def test_xcorr_edge_cases(fb):
    z = np.zeros((2, 20, 24), np.float32)
    dx, dy, cf = fb.matcher.xcorr_fft(z, z, pad=True, subpixel=True)
    ex, ey, ec = ncc_ref.xcorr_fft(z, z, pad=True, subpixel=True)
    np.testing.assert_array_equal(dx, ex); np.testing.assert_array_equal(dy, ey); np.testing.assert_array_equal(cf, ec)
    e = fb.matcher.xcorr_fft(np.zeros((0, 8, 8), np.float32), np.zeros((0, 8, 8), np.float32))
    assert all(a.shape == (0,) for a in e)
    # exact tie: two identical maxima -> first index wins (numpy argmax rule)
    a = np.zeros((1, 16, 16), np.float32); a[0, 4, 4] = 1
    b = np.zeros((1, 16, 16), np.float32); b[0, 6, 9] = 1; b[0, 2, 3] = 1
    _check(fb.matcher.xcorr_fft(a, b, pad=False), ncc_ref.xcorr_fft(a, b, pad=False))


def test_xcorr_shift_property(fb):
    """size-independent property at the full 4k fine-class batch: a circular shift of img1
    moves the unpadded peak by exactly that shift."""
    rng = np.random.default_rng(11)
    t = gaussian_filter(rng.standard_normal((385, 75, 75)), (0, 1.2, 1.2)).astype(np.float32)
    sh = rng.integers(-30, 31, size=(385, 2))
    u = np.stack([np.roll(t[k], tuple(sh[k]), axis=(0, 1)) for k in range(385)])
    dx, dy, cf = fb.matcher.xcorr_fft(t, u, pad=False)
    np.testing.assert_array_equal(dx, sh[:, 1]); np.testing.assert_array_equal(dy, sh[:, 0])
    assert np.all(cf > 0.5)


def test_global_translation_golden(fb):
    g = load_golden('g3_global.npz')
    np.testing.assert_allclose(fb.matcher.global_translation_matcher(g['d0'], g['d1'], conf_thresh=0.3), g['plain'], atol=ATOL)
    np.testing.assert_allclose(fb.matcher.global_translation_matcher(g['d0'], g['e1'], conf_thresh=2.0), g['fallback'], atol=ATOL)
    np.testing.assert_allclose(fb.matcher.global_translation_matcher(g['d0'], g['f1'], conf_thresh=2.0), g['unequal'], atol=ATOL)


def test_dog_golden(fb):
    g = load_golden('g2_dog.npz')

    def rel(a, b):
        return np.abs(a - b).max() / np.abs(b).max()
    for s in (1.25, 2.5, 3.5):
        assert rel(fb.common.masked_dog_filter(g['img'], s), g[f'dog_s{s}']) < 1e-5
    assert rel(fb.common.masked_dog_filter(g['img'], 2.5, mask=g['mask']), g['dog_masked_signed']) < 1e-5
    assert rel(fb.common.masked_dog_filter(g['img'], 2.5, mask=g['mask'], signed=False), g['dog_masked_unsigned']) < 1e-5
    assert rel(fb.common.masked_dog_filter(g['stack'], 2.5), g['dog_stack_s2.5']) < 1e-5
    assert rel(fb.common.masked_dog_filter(g['fimg'], 1.25), g['dog_fimg_s1.25']) < 1e-5


def test_dog_vs_oracle_strip(fb):
    """the 4k strip shapes: fine 4096x510 sigma 2.5 and coarse 2048x255 sigma 1.25; ragged tile edges"""
    rng = np.random.default_rng(3)
    for (h, w, s) in ((4096, 510, 2.5), (2048, 255, 1.25), (97, 131, 2.5), (7, 300, 1.25)):
        img = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        got = fb.common.masked_dog_filter(img, s)
        exp = ncc_ref.masked_dog_filter(img, s)
        assert np.abs(got - exp).max() <= 1e-5 * np.abs(exp).max()
    # all-ones mask: identical to the unmasked filter (common.py:368)
    img = rng.integers(0, 256, size=(64, 64), dtype=np.uint8)
    np.testing.assert_array_equal(fb.common.masked_dog_filter(img, 2.5, mask=np.ones((64, 64), bool)),
                                  fb.common.masked_dog_filter(img, 2.5))


def test_area_downsample(fb):
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, size=(3, 510, 4096), dtype=np.uint8)
    np.testing.assert_array_equal(fb.common.area_downsample2(img), ncc_ref.area_downsample2(img))
    # odd sizes: cvRound(n / 2) outputs (half to even), edge cells average the pixels that exist
    for shape, out_shape in (((2, 511, 4095), (256, 2048)), ((2, 509, 4097), (254, 2048)), ((1, 7, 5), (4, 2)), ((1, 5, 7), (2, 4))):
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        got = fb.common.area_downsample2(img)
        assert got.shape[-2:] == out_shape
        np.testing.assert_array_equal(got, ncc_ref.area_downsample2(img))


@pytest.mark.parametrize('bh,bw,pad', [(75, 73, False), (75, 71, False), (72, 75, False), (70, 72, False), (75, 73, True), (250, 247, True)])
def test_blocks_affine_gather_vs_oracle(fb, bh, bw, pad):
    """fb_ncc_blocks_affine_dev: image 1 sampled through a per-block affine map with cv2.remap's bilinear rule (oracle
    restatement, unpinned) -- on-chip kernel, generic streaming kernels (150 x 150) and power-of-two kernels (512 x 512)"""
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    rng = np.random.default_rng(bh + bw + int(pad))
    n_img, IH, IW = 2, 600, 640
    from scipy.ndimage import gaussian_filter
    base = np.stack([gaussian_filter(rng.standard_normal((IH + 40, IW + 40)), 1.2) for _ in range(n_img)]).astype(np.float32)
    img0 = np.ascontiguousarray(base[:, 20:20 + IH, 20:20 + IW])
    # image 1 = image 0 seen through a slight magnification + rotation (resampled here with scipy, independent of the sampler under test)
    from scipy.ndimage import affine_transform
    th, sc = 0.002, 1.0015
    M = sc * np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    img1 = np.stack([affine_transform(base[k], M, offset=(20 - 1.3, 20 + 2.2), output_shape=(IH, IW), order=3) for k in range(n_img)]).astype(np.float32)
    N = 12
    blk = np.zeros((N, 9), dtype=np.int32)
    aff = np.zeros((N, 10))
    A = np.array([[1.0012, 0.0017], [-0.0019, 0.9991]])
    exp0, exp1 = [], []
    for k in range(N):
        im = k % n_img
        x0 = int(rng.integers(-10, IW - bw + 10)); y0 = int(rng.integers(-10, IH - bh + 10))
        t = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3)])
        blk[k] = (im, x0, y0, bh, bw, x0, y0, bh, bw)
        xmin = int(np.floor(min(x0 * A[0, 0] + y0 * A[1, 0] + t[0], x0 * A[0, 0] + (y0 + bh) * A[1, 0] + t[0]))) - 4
        ymin = int(np.floor(min(x0 * A[0, 1] + y0 * A[1, 1] + t[1], (x0 + bw) * A[0, 1] + y0 * A[1, 1] + t[1]))) - 4
        aff[k] = (x0, y0, A[0, 0], A[1, 0], t[0], A[0, 1], A[1, 1], t[1], xmin, ymin)
        c0 = np.zeros((bh, bw), np.float32)
        ya, yb, xa, xb = max(y0, 0), min(y0 + bh, IH), max(x0, 0), min(x0 + bw, IW)
        c0[ya - y0:yb - y0, xa - x0:xb - x0] = img0[im, ya:yb, xa:xb]
        exp0.append(c0)
        exp1.append(ncc_ref.crop_affine(img1[im], x0, y0, bh, bw, A, t, (xmin, ymin)))
    exp = ncc_ref.xcorr_fft(np.stack(exp0), np.stack(exp1), pad=pad, subpixel=True)
    nfl = ncc_ref.next_fast_len
    Fh, Fw = (nfl(2 * bh - 1), nfl(2 * bw - 1)) if pad else (nfl(bh), nfl(bw))
    d0 = _lib.DeviceBuffer.from_array(img0); d1 = _lib.DeviceBuffer.from_array(img1)
    dblk = _lib.DeviceBuffer.from_array(blk); daff = _lib.DeviceBuffer.from_array(aff)
    out = _lib.DeviceBuffer(N * 20)
    _lib.check(lib.fb_ncc_blocks_affine_dev(ctx, d0.ptr, d1.ptr, IH, IW, IH, IW, N, dblk.ptr, daff.ptr, bh, bw, Fh, Fw, 1, 2,
                                            out.ptr, out.offset(8 * N), out.offset(16 * N)))
    raw = out.to_array((20 * N,), np.uint8)
    got = (raw[:8 * N].view(np.float64), raw[8 * N:16 * N].view(np.float64), raw[16 * N:].view(np.float32))
    _check(got, exp)
    assert np.median(exp[2]) > 0.5                       # the blocks do match
    for b in (d0, d1, dblk, daff, out):
        b.free()


@pytest.mark.parametrize('bh,bw', [(75, 73), (75, 70), (71, 75), (72, 72), (60, 50)])
def test_blocks_crop_across_the_image_border_vs_oracle(fb, bh, bw):
    """fb_ncc_blocks_dev (crop mode, the fine rounds of the strip matcher): windows of both images that stick out of their
    image on every side are zero there (matcher.py:63-64 pads with zeros; the register-resident kernel gets them from the range
    check of its buffer loads), unequal window sizes inside one launch"""
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    rng = np.random.default_rng(bh * 100 + bw)
    n_img, IH, IW = 3, 300, 260
    from scipy.ndimage import gaussian_filter
    base = np.stack([gaussian_filter(rng.standard_normal((IH + 40, IW + 40)), 1.2) for _ in range(n_img)]).astype(np.float32)
    img0 = np.ascontiguousarray(base[:, 20:20 + IH, 20:20 + IW]); img1 = np.ascontiguousarray(base[:, 18:18 + IH, 23:23 + IW])
    N = 40
    blk = np.zeros((N, 9), dtype=np.int32)
    c0 = np.zeros((N, bh, bw), np.float32); c1 = np.zeros((N, bh, bw), np.float32)
    for k in range(N):
        im = k % n_img
        x0 = int(rng.integers(-30, IW - bw + 30)); y0 = int(rng.integers(-30, IH - bh + 30))
        x1 = x0 + int(rng.integers(-4, 5)); y1 = y0 + int(rng.integers(-4, 5))
        h0, w0 = (bh, bw) if k % 5 else (bh - 3, bw - 2)
        h1, w1 = (bh, bw) if (k % 7 != 3 or k % 5 == 0) else (bh - 1, bw - 4)           # never both windows smaller: the FFT size stays that of (bh, bw)
        blk[k] = (im, x0, y0, h0, w0, x1, y1, h1, w1)
        for c, img, xx, yy, hh, ww in ((c0, img0, x0, y0, h0, w0), (c1, img1, x1, y1, h1, w1)):
            ya, yb, xa, xb = max(yy, 0), min(yy + hh, IH), max(xx, 0), min(xx + ww, IW)
            if yb > ya and xb > xa:
                c[k, ya - yy:yb - yy, xa - xx:xb - xx] = img[im, ya:yb, xa:xb]
    nfl = ncc_ref.next_fast_len
    Fh, Fw = nfl(bh), nfl(bw)
    # the reference call block by block (windows of unequal size: matcher.py:59-64 pads both to the common FFT size, 107-110 re-centres)
    exp = [[], [], []]; exp64 = [[], [], []]
    for k in range(N):
        _, _, _, h0, w0, _, _, h1, w1 = blk[k]
        e = ncc_ref.xcorr_fft(c0[k:k + 1, :h0, :w0], c1[k:k + 1, :h1, :w1], pad=False, subpixel=True)
        e64 = ncc_ref.xcorr_fft(c0[k:k + 1, :h0, :w0].astype(np.float64), c1[k:k + 1, :h1, :w1].astype(np.float64), pad=False, subpixel=True)
        for i in range(3):
            exp[i].append(e[i][0]); exp64[i].append(e64[i][0])
    exp = tuple(np.asarray(v) for v in exp); exp64 = tuple(np.asarray(v) for v in exp64)

    def check(got, ref):
        """integer peaks bit-exact; sub-pixel offsets and confidences within 1e-4 where the reference's own float32 arithmetic
        pins them that well -- a window that is mostly outside its image has a flat peak, and there the float32 reference is
        itself some 1e-3 px from the double precision answer: the device may be as far from it as a few times that"""
        np.testing.assert_array_equal(np.round(got[0]), np.round(ref[0])); np.testing.assert_array_equal(np.round(got[1]), np.round(ref[1]))
        for i in range(3):
            slack = ATOL + 4 * np.abs(np.asarray(exp[i], np.float64) - exp64[i])
            bad = np.nonzero(np.abs(np.asarray(got[i], np.float64) - exp64[i]) > slack)[0]
            assert bad.size == 0, (i, bad, blk[bad], np.asarray(got[i])[bad], exp64[i][bad])
    d0 = _lib.DeviceBuffer.from_array(img0); d1 = _lib.DeviceBuffer.from_array(img1); dblk = _lib.DeviceBuffer.from_array(blk)
    out = _lib.DeviceBuffer(N * 20)
    def run():
        _lib.check(lib.fb_ncc_blocks_dev(ctx, d0.ptr, d1.ptr, IH, IW, IH, IW, N, dblk.ptr, bh, bw, Fh, Fw, 1, 2, out.ptr, out.offset(8 * N), out.offset(16 * N)))
        raw = out.to_array((20 * N,), np.uint8)
        return raw[:8 * N].view(np.float64).copy(), raw[8 * N:16 * N].view(np.float64).copy(), raw[16 * N:].view(np.float32).copy()
    got = run()
    check(got, exp)
    # the run-time mixed-radix on-chip kernel (ncc_small_fused) is the second opinion on every block
    import os, subprocess, sys, json
    code = ('import numpy as np, json, sys; from feabas_amd import _lib; lib, ctx = _lib.load(), _lib.ctx(); d = np.load(sys.argv[1]);'
            'N = int(d["blk"].shape[0]); d0 = _lib.DeviceBuffer.from_array(d["img0"]); d1 = _lib.DeviceBuffer.from_array(d["img1"]); db = _lib.DeviceBuffer.from_array(d["blk"]);'
            'out = _lib.DeviceBuffer(N * 20); IH, IW = d["img0"].shape[1:]; bh, bw, Fh, Fw = [int(v) for v in d["geo"]];'
            '_lib.check(lib.fb_ncc_blocks_dev(ctx, d0.ptr, d1.ptr, IH, IW, IH, IW, N, db.ptr, bh, bw, Fh, Fw, 1, 2, out.ptr, out.offset(8 * N), out.offset(16 * N)));'
            'raw = out.to_array((20 * N,), np.uint8); np.save(sys.argv[2], raw)')
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, 'in.npz'), img0=img0, img1=img1, blk=blk, geo=np.asarray([bh, bw, Fh, Fw]))
        env = dict(os.environ, FEABAS_HIP_NO_PFA='1', PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        subprocess.run([sys.executable, '-c', code, os.path.join(td, 'in.npz'), os.path.join(td, 'out.npy')], check=True, env=env)
        raw = np.load(os.path.join(td, 'out.npy'))
    ref = (raw[:8 * N].view(np.float64), raw[8 * N:16 * N].view(np.float64), raw[16 * N:].view(np.float32))
    check(ref, exp)            # (the same bar for the older kernel)
    np.testing.assert_array_equal(np.round(got[0]), np.round(ref[0])); np.testing.assert_array_equal(np.round(got[1]), np.round(ref[1]))
    for b in (d0, d1, dblk, out):
        b.free()


def test_dog_and_downsample_of_unequal_images_in_one_stack(fb):
    """fb_area_downsample2_sizes_dev / fb_dog_sizes_dev: images of unequal size in padded slots are processed as images of
    their own size ('nearest' extension at their own border) -- bit-identical to the equal-size kernels image by image --
    and the rest of every slot is zero"""
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    rng = np.random.default_rng(12)
    sizes = np.array([[300, 200], [257, 131], [299, 199], [64, 70]], dtype=np.int32)
    N, H, W = sizes.shape[0], 300, 200
    stack = rng.integers(0, 256, (N, H, W), dtype=np.uint8)          # the padding holds junk on purpose
    d_in = _lib.DeviceBuffer.from_array(stack); d_sz = _lib.DeviceBuffer.from_array(sizes)
    hc, wc = fb.common.half_size(H), fb.common.half_size(W)
    d_small = _lib.DeviceBuffer.from_array(np.full((N, hc, wc), 99, dtype=np.uint8)); d_dog = _lib.DeviceBuffer.from_array(np.full((N, H, W), 7.0, dtype=np.float32))
    _lib.check(lib.fb_area_downsample2_sizes_dev(ctx, d_in.ptr, N, H, W, d_sz.ptr, d_small.ptr))
    _lib.check(lib.fb_dog_sizes_dev(ctx, d_in.ptr, 0, N, H, W, d_sz.ptr, 2.5, 1, d_dog.ptr))
    small = d_small.to_array((N, hc, wc), np.uint8); dog = d_dog.to_array((N, H, W), np.float32)
    for n, (h, w) in enumerate(sizes):
        img = np.ascontiguousarray(stack[n, :h, :w])
        exp_s = fb.common.area_downsample2(img)
        np.testing.assert_array_equal(small[n, :exp_s.shape[0], :exp_s.shape[1]], exp_s)
        assert not small[n, exp_s.shape[0]:].any() and not small[n, :, exp_s.shape[1]:].any()
        exp_d = fb.common.masked_dog_filter(img, 2.5)
        np.testing.assert_array_equal(dog[n, :h, :w], exp_d)
        assert not dog[n, h:].any() and not dog[n, :, w:].any()
        np.testing.assert_allclose(exp_d, ncc_ref.masked_dog_filter(img, 2.5), atol=1e-5 * np.abs(exp_d).max() + 1e-4)
    for b in (d_in, d_sz, d_small, d_dog):
        b.free()



def test_dog_of_the_half_resolution_image_in_one_kernel(fb):
    """fb_dog_down2_dev = masked_dog_filter(cv2.resize(img, 0.5, INTER_AREA), sigma) with the 2 x 2 average taken in the DoG's
    loader (matcher.py:255-256 + 273-274): bit-identical to area downsample followed by the DoG, even and odd sizes"""
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    rng = np.random.default_rng(8)
    for (n, h2, w2, s) in ((3, 4096, 510, 1.25), (2, 510, 4096, 1.25), (2, 511, 1021, 1.25), (2, 257, 130, 2.5)):
        img = rng.integers(0, 256, size=(n, h2, w2), dtype=np.uint8)
        small = fb.common.area_downsample2(img)
        exp = fb.common.masked_dog_filter(small, s)
        d_in = _lib.DeviceBuffer.from_array(img); d_out = _lib.DeviceBuffer(exp.nbytes)
        _lib.check(lib.fb_dog_down2_dev(ctx, d_in.ptr, n, h2, w2, s, 1, d_out.ptr))
        got = d_out.to_array(exp.shape, np.float32)
        np.testing.assert_array_equal(got, exp)
        ref = ncc_ref.masked_dog_filter(ncc_ref.area_downsample2(img[0]), s)
        assert np.abs(got[0] - ref).max() <= 1e-5 * np.abs(ref).max()
        d_in.free(); d_out.free()


def test_dog_of_two_stacks_in_one_launch(fb):
    """fb_dog_pair_dev / fb_dog_down2_pair_dev: the two strip stacks of a batch of pairs filtered by one launch (images N .. 2N-1
    come from the second stack) -- bit-identical to a launch per stack, whatever row segments the larger launch is cut into"""
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    rng = np.random.default_rng(18)
    for (n, h, w, s) in ((5, 4096, 510, 2.5), (3, 510, 4096, 2.5), (2, 511, 1021, 1.25), (40, 300, 120, 2.5)):
        a = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8); b = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
        da = _lib.DeviceBuffer.from_array(a); db = _lib.DeviceBuffer.from_array(b)
        d_out = _lib.DeviceBuffer(2 * n * h * w * 4)
        _lib.check(lib.fb_dog_pair_dev(ctx, da.ptr, db.ptr, 0, n, h, w, s, 1, d_out.ptr))
        got = d_out.to_array((2, n, h, w), np.float32)
        np.testing.assert_array_equal(got[0], fb.common.masked_dog_filter(a, s)); np.testing.assert_array_equal(got[1], fb.common.masked_dog_filter(b, s))
        # float32 input
        fa = a.astype(np.float32) * 0.5; fb_ = b.astype(np.float32) * 0.25
        dfa = _lib.DeviceBuffer.from_array(fa); dfb = _lib.DeviceBuffer.from_array(fb_)
        _lib.check(lib.fb_dog_pair_dev(ctx, dfa.ptr, dfb.ptr, 1, n, h, w, s, 1, d_out.ptr))
        got = d_out.to_array((2, n, h, w), np.float32)
        np.testing.assert_array_equal(got[0], fb.common.masked_dog_filter(fa, s)); np.testing.assert_array_equal(got[1], fb.common.masked_dog_filter(fb_, s))
        # the x0.5 form
        hs, ws = fb.common.half_size(h), fb.common.half_size(w)
        _lib.check(lib.fb_dog_down2_pair_dev(ctx, da.ptr, db.ptr, n, h, w, 1.25, 1, d_out.ptr))
        got = d_out.to_array((2 * n * hs * ws,), np.float32).reshape(2, n, hs, ws)
        for k, src in enumerate((a, b)):
            np.testing.assert_array_equal(got[k], fb.common.masked_dog_filter(fb.common.area_downsample2(src), 1.25))
        for buf in (da, db, dfa, dfb, d_out):
            buf.free()


@pytest.mark.parametrize('tag', ['masks', 'ones'])
@pytest.mark.parametrize('pad', [1, 0])
@pytest.mark.parametrize('cm', [0, 1, 2])
def test_g20_xcorr_normalized_vs_reference(fb, tag, pad, cm):
    """xcorr_fft(normalize=True) (matcher.py:70-81, 119-122; fb_ncc_batch_normalized) against the reference's own outputs: golden
    G20, the cases the oracle is pinned by (tests/test_oracle_golden.py::test_g20_xcorr_normalized)"""
    g = load_golden('g20_xcorr_normalized.npz')
    if tag == 'masks':
        a, b, kw = g['img0'] * g['mask0'], g['img1'] * g['mask1'], dict(mask0=g['mask0'], mask1=g['mask1'])
    else:
        a, b, kw = g['img0'], g['img1'], {}
    got = fb.matcher.xcorr_fft(a, b, conf_mode=cm, pad=bool(pad), subpixel=True, normalize=True, **kw)
    key = f'{tag}_p{pad}_c{cm}'
    fh, fw = ncc_ref.fft_shape(a.shape[-2:], b.shape[-2:], bool(pad))
    _check(got, (g[key + '_dx'], g[key + '_dy'], g[key + '_conf']), conf_atol=(2.5 * fh * fw * 6e-8 if cm == 1 else None))
    # and against the plain call: the normalisation must have been applied
    if cm == 2 and tag == 'masks':
        plain = fb.matcher.xcorr_fft(a, b, conf_mode=cm, pad=bool(pad), subpixel=True)
        assert np.abs(plain[2] - got[2]).max() > 1e-3
