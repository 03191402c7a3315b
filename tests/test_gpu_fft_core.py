"""The LDS mixed-radix FFT core (feabas_amd/csrc/fb_ldsfft.h) against numpy, every radix mix the NCC path uses."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_tc = {}


def _test_lib_and_ctx():
    """fb_debug_fft1d lives in the test build of the library (include/feabas_hip_test.h) and takes a context of THAT instance"""
    if not _tc:
        from feabas_amd import _lib
        lib = _lib.load_test()
        h = lib.fb_create(0)
        assert h
        _tc['v'] = (lib, h)
    return _tc['v']



@pytest.mark.parametrize('n', [2, 8, 16, 64, 75, 72, 150, 144, 135, 500, 512, 576, 1000, 1024, 2048, 3000, 4096])
@pytest.mark.parametrize('pad', [0, 1])
def test_fft1d_vs_numpy(fb, n, pad):
    from feabas_amd import _lib
    lib, ctx = _test_lib_and_ctx()
    rng = np.random.default_rng(n)
    m = max(1, min(4, 8192 // n))
    x = (rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))).astype(np.complex64)
    out = np.empty_like(x)
    _lib.check(lib.fb_debug_fft1d(ctx, _lib.ptr(x), _lib.ptr(out), m, n, 0, pad))
    ref = np.fft.fft(x.astype(np.complex128), axis=-1)
    assert np.abs(out - ref).max() <= 3e-6 * np.abs(ref).max()
    _lib.check(lib.fb_debug_fft1d(ctx, _lib.ptr(x), _lib.ptr(out), m, n, 1, pad))
    ref = np.fft.ifft(x.astype(np.complex128), axis=-1) * n
    assert np.abs(out - ref).max() <= 3e-6 * np.abs(ref).max()


@pytest.mark.parametrize('n', [64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_fft1d_pow2_packed_core_vs_numpy(fb, n):
    """fb_fft2.h: compile-time plans + packed-FP32 butterflies (the streaming-class kernels at power-of-two shapes)"""
    from feabas_amd import _lib
    lib, ctx = _test_lib_and_ctx()
    rng = np.random.default_rng(n + 1)
    m = max(1, min(4, 8192 // n))
    x = (rng.standard_normal((m, n)) + 1j * rng.standard_normal((m, n))).astype(np.complex64)
    out = np.empty_like(x)
    _lib.check(lib.fb_debug_fft1d(ctx, _lib.ptr(x), _lib.ptr(out), m, n, 0, 2))
    ref = np.fft.fft(x.astype(np.complex128), axis=-1)
    assert np.abs(out - ref).max() <= 3e-6 * np.abs(ref).max()
    _lib.check(lib.fb_debug_fft1d(ctx, _lib.ptr(x), _lib.ptr(out), m, n, 1, 2))
    ref = np.fft.ifft(x.astype(np.complex128), axis=-1) * n
    assert np.abs(out - ref).max() <= 3e-6 * np.abs(ref).max()
