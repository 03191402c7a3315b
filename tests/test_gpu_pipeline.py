"""GPU parity of the device-resident tile-pair matcher (feabas_amd/stitch_pipeline.py)
against the oracle pipeline, on synthetic strips generated on the device."""
import ctypes as C

import numpy as np
import pytest

from oracle import pipeline_ref

pytestmark = pytest.mark.gpu


def _synth(fb, P, H, W, seed, max_shift, step=1):
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
    _lib.check(lib.fb_synth_strips_dev(ctx, P, 0, H, W, seed, max_shift, step, s0.ptr, s1.ptr, sh.ptr))
    _lib.check(lib.fb_sync(ctx))
    return s0, s1, sh.to_array((P, 2), np.int32)


@pytest.mark.parametrize('H,W,P', [(1024, 256, 6), (256, 1024, 4)])
def test_pipeline_vs_oracle(fb, H, W, P):
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    s0, s1, shifts = _synth(fb, P, H, W, seed=7, max_shift=12)
    m = StripBatchMatcher(P, H, W)
    got = StripBatchMatcher.per_pair(m.match(s0.ptr, s1.ptr))
    h0 = s0.to_array((P, H, W), np.uint8); h1 = s1.to_array((P, H, W), np.uint8)
    assert h0.std() > 20                                     # real texture
    for p in range(P):
        exp = pipeline_ref.match_pair(h0[p], h1[p])
        g = got[p]
        # strip1(x,y) = texture(x+sx, y+sy): mesh0 must move by (-sx, -sy); the x0.5 coarse match is good to +-1 px
        assert (g['tx'], g['ty']) == (exp['tx'], exp['ty'])
        assert abs(g['tx'] + shifts[p, 0]) <= 1 and abs(g['ty'] + shifts[p, 1]) <= 1
        # the fine blocks recover the injected offset exactly: xy1 - xy0 (INITIAL gear) = -(sx, sy)
        d = np.median(g['xy1'] - g['xy0'], axis=0)
        assert np.abs(d + shifts[p]).max() < 0.3
        assert abs(g['conf0'] - exp['conf0']) < 1e-4
        assert g['needs_host'] == exp['needs_host'] == False      # odd offsets relax rigidly (FEM oracle inside pipeline_ref)
        assert g['xy0'].shape == exp['xy0'].shape and g['xy0'].shape[0] > 10
        np.testing.assert_array_equal(np.round(g['xy1'] - g['xy0']), np.round(exp['xy1'] - exp['xy0']))
        np.testing.assert_allclose(g['xy0'], exp['xy0'], atol=1e-4)
        np.testing.assert_allclose(g['xy1'], exp['xy1'], atol=1e-4)
        np.testing.assert_allclose(g['weight'], exp['weight'], atol=1e-4)
    m.free()


def test_pipeline_even_offsets_take_the_device_branch(fb):
    """even offsets are recovered exactly by the x0.5 coarse match, so no pair needs the mesh-relaxation
    branch: this is the branch bench.py times"""
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    P, H, W = 8, 1024, 256
    s0, s1, shifts = _synth(fb, P, H, W, seed=11, max_shift=20, step=2)
    assert np.all(shifts % 2 == 0)
    m = StripBatchMatcher(P, H, W)
    res = m.match(s0.ptr, s1.ptr)
    np.testing.assert_array_equal(res['tx'], -shifts[:, 0]); np.testing.assert_array_equal(res['ty'], -shifts[:, 1])
    assert not res['needs_host'].any() and res['valid'].all()
    h0 = s0.to_array((P, H, W), np.uint8); h1 = s1.to_array((P, H, W), np.uint8)
    got = StripBatchMatcher.per_pair(res)
    for p in (0, 5):
        exp = pipeline_ref.match_pair(h0[p], h1[p])
        np.testing.assert_allclose(got[p]['xy0'], exp['xy0'], atol=1e-4); np.testing.assert_allclose(got[p]['xy1'], exp['xy1'], atol=1e-4)
        np.testing.assert_allclose(got[p]['weight'], exp['weight'], atol=1e-4)
    m.free()
