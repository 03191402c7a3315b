"""The 8192-point shapes of the power-of-two NCC core (fb_ncc_p2.inc): rows of 8192 points, columns of 8192 points as two
half-length items per column pair ("split columns").  matcher.xcorr_fft (matcher.py:22-135) of two whole 4096-pixel tiles, padded,
asks for exactly that shape: next_fast_len(4096 + 4096 - 1) = 8192 (stress variant).  Same bars as
test_gpu_ncc.py: integer peaks bit-exact, sub-pixel offsets and confidences within 1e-4 of the oracle."""
import os

import numpy as np
import pytest
from scipy.ndimage import gaussian_filter

from oracle import ncc_ref

pytestmark = pytest.mark.gpu


def _check(got, exp, atol=1e-4, conf_atol=None):
    np.testing.assert_array_equal(np.round(got[0]), np.round(exp[0]))
    np.testing.assert_array_equal(np.round(got[1]), np.round(exp[1]))
    np.testing.assert_allclose(got[0], exp[0], atol=atol, rtol=0)
    np.testing.assert_allclose(got[1], exp[1], atol=atol, rtol=0)
    np.testing.assert_allclose(got[2], exp[2], atol=atol if conf_atol is None else conf_atol, rtol=0)


def _pairs(rng, n, s0, s1, maxshift):
    H = max(s0[0], s1[0]); W = max(s0[1], s1[1])
    big = gaussian_filter(rng.standard_normal((H + 2 * maxshift + 8, W + 2 * maxshift + 8)).astype(np.float32), 1.5)
    big -= gaussian_filter(big, 4.0)
    i0 = np.empty((n,) + s0, np.float32); i1 = np.empty((n,) + s1, np.float32)
    for k in range(n):
        sy, sx = rng.integers(-maxshift, maxshift + 1, 2)
        y = x = maxshift + 4
        i0[k] = big[y:y + s0[0], x:x + s0[1]]
        i1[k] = big[y + sy:y + sy + s1[0], x + sx:x + sx + s1[1]]
        big = np.roll(big, (17, 29), axis=(0, 1))
    i1 += 0.05 * i1.std() * rng.standard_normal(i1.shape).astype(np.float32)
    return i0, i1


def _launch_shape(fb, s0, s1, pad):
    from feabas_amd import matcher
    fh = matcher.next_fast_len(s0[0] + s1[0] - 1) if pad else matcher.next_fast_len(max(s0[0], s1[0]))
    fw = matcher.next_fast_len(s0[1] + s1[1] - 1) if pad else matcher.next_fast_len(max(s0[1], s1[1]))
    return fh, fw


@pytest.mark.parametrize('s0,s1,n,pad', [
    ((4096, 130), (4096, 130), 2, True),       # columns of 8192 points (split), rows 259 -> 260 run at 512
    ((3000, 300), (3000, 300), 2, True),       # 5999 -> 6000 runs at 8192 (split), 599 -> 600 at 1024
    ((4096, 200), (3500, 180), 2, True),       # unequal crops
    ((130, 4096), (130, 4096), 2, True),       # rows of 8192 points, columns 260 -> 512
    ((128, 8192), (128, 8192), 2, False),      # rows of 8192 points as a circular axis
    ((2300, 2100), (2300, 2100), 1, True),     # 4599 -> 4608 at 8192 (split), 4199 -> 4200 at 8192: both long forms together
])
def test_xcorr_long_axes_vs_oracle(fb, s0, s1, n, pad):
    rng = np.random.default_rng(s0[0] + 3 * s1[1] + n)
    i0, i1 = _pairs(rng, n, s0, s1, maxshift=20)
    for sub, cm in ((True, 2), (False, 0)):
        exp = ncc_ref.xcorr_fft(i0, i1, pad=pad, subpixel=sub, conf_mode=cm)
        got = fb.matcher.xcorr_fft(i0, i1, pad=pad, subpixel=sub, conf_mode=cm)
        _check(got, exp)
    # the same answers as the route these shapes took before (rocFFT at the reference's own size)
    os.environ['FEABAS_HIP_NO_LONG'] = '1'
    try:
        old = fb.matcher.xcorr_fft(i0, i1, pad=pad, subpixel=True, conf_mode=2)
    finally:
        del os.environ['FEABAS_HIP_NO_LONG']
    got = fb.matcher.xcorr_fft(i0, i1, pad=pad, subpixel=True, conf_mode=2)
    np.testing.assert_array_equal(np.round(got[0]), np.round(old[0])); np.testing.assert_array_equal(np.round(got[1]), np.round(old[1]))
    np.testing.assert_allclose(got[0], old[0], atol=1e-4); np.testing.assert_allclose(got[2], old[2], atol=1e-4)


def test_xcorr_long_axes_std_confidence(fb):
    """FFT_CONF_STD depends on the surface size (matcher.py:129-134): only shapes that ARE 8192 as they stand take the long forms"""
    rng = np.random.default_rng(12)
    i0, i1 = _pairs(rng, 1, (4096, 130), (4096, 130), maxshift=10)      # 8192 x 260: the rows are not a power of two -> the old route
    j0, j1 = _pairs(rng, 1, (4096, 128), (4096, 128), maxshift=10)      # 8192 x 256 with pad=False on x would be circular; padded: 255 -> 256
    for a, b in ((i0, i1), (j0, j1)):
        exp = ncc_ref.xcorr_fft(a, b, pad=True, subpixel=True, conf_mode=1)
        got = fb.matcher.xcorr_fft(a, b, pad=True, subpixel=True, conf_mode=1)
        f = 8192 * (260 if a.shape[2] == 130 else 256)
        _check(got, exp, conf_atol=2.5 * f * 6e-8)


def test_xcorr_whole_tiles_padded(fb):
    """two whole 4096 x 4096 tiles, zero padded: FFT 8192 x 8192 -- the stress shape; one pair, with the peak in each of the four
    quadrants of the surface in turn (negative lags sit in rows / columns >= 4096: the upper half of the split inverse)"""
    rng = np.random.default_rng(7)
    base = gaussian_filter(rng.standard_normal((4096 + 80, 4096 + 80)).astype(np.float32), 1.5)
    base -= gaussian_filter(base, 4.0)
    for sy, sx in ((13, -9), (-21, 17)):
        i0 = np.ascontiguousarray(base[40:40 + 4096, 40:40 + 4096][None])
        i1 = np.ascontiguousarray(base[40 + sy:40 + sy + 4096, 40 + sx:40 + sx + 4096][None])
        exp = ncc_ref.xcorr_fft(i0, i1, pad=True, subpixel=True)
        got = fb.matcher.xcorr_fft(i0, i1, pad=True, subpixel=True)
        assert abs(exp[0][0]) > 5 and abs(exp[1][0]) > 5
        _check(got, exp)
