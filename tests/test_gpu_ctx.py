"""context services of the C ABI: staged host <-> device copies, the allocation cache, device-side mask counting"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_staged_copies_round_trip(fb):
    """fb_memcpy_h2d / fb_memcpy_d2h go through the context's pinned staging ring in 4 MiB chunks: any size and any host
    alignment comes back bit for bit"""
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    rng = np.random.default_rng(3)
    for n in (1, 31, 32 << 10, (32 << 10) + 1, (4 << 20) - 3, (4 << 20), (4 << 20) + 5, 3 * (4 << 20) + 12345):
        src = rng.integers(0, 256, n + 3, dtype=np.uint8)[3:]            # odd host address
        d = _lib.DeviceBuffer(n)
        _lib.check(lib.fb_memcpy_h2d(ctx, d.ptr, _lib.ptr(np.ascontiguousarray(src)), n))
        back = d.to_array((n,), np.uint8)
        np.testing.assert_array_equal(back, src)
        d.free()


def test_allocation_cache_reuses_blocks(fb):
    """a block handed back by fb_free serves the next fb_malloc of a similar size (no hipMalloc / hipFree per temporary)"""
    from feabas_amd import _lib
    a = _lib.DeviceBuffer(5 << 20)
    pa = a.ptr.value
    a.free()
    b = _lib.DeviceBuffer((5 << 20) - 4096)
    assert b.ptr.value == pa
    c = _lib.DeviceBuffer(5 << 20)
    assert c.ptr.value != pa
    # contents are not cleared: a fresh use writes before it reads
    b.free(); c.free()
    big = _lib.DeviceBuffer(64 << 20)
    assert big.ptr.value not in (pa,)                # a 5 MiB block never serves a 64 MiB request
    big.free()


@pytest.mark.parametrize('n,off', [(0, 0), (1, 0), (15, 1), (4097, 3), (1 << 20, 0), ((1 << 22) + 77, 5)])
def test_count_nonzero(fb, n, off):
    """fb_count_nonzero_dev = np.count_nonzero of a uint8 device array (the mask.any() of MeshRenderer.crop_multiple,
    renderer.py:601-648), at any pointer alignment"""
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    rng = np.random.default_rng(n + off)
    for density in (0.0, 0.001, 0.5):
        m = (rng.random(n + off) < density).astype(np.uint8) * rng.integers(1, 256, n + off, dtype=np.uint8)
        d = _lib.DeviceBuffer.from_array(m) if m.size else _lib.DeviceBuffer(16)
        cnt = C.c_int64(-1)
        _lib.check(lib.fb_count_nonzero_dev(ctx, d.offset(off), n, C.byref(cnt)))
        assert cnt.value == int(np.count_nonzero(m[off:]))
        d.free()
