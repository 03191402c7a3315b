        # (the oracle solves exactly, the device by PCG to 1e-9: the matches agree to 5e-5 px, tools/diag_deformed_bars.py)
"""GPU parity of the device-resident tile-pair matcher (feabas_amd/stitch_pipeline.py)
against the oracle pipeline, on synthetic strips generated on the device."""
import ctypes as C

import numpy as np
import pytest

from oracle import pipeline_ref

pytestmark = pytest.mark.gpu


def _synth(fb, P, H, W, seed, max_shift, step=1, warp=0.0):
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    s0 = _lib.DeviceBuffer(P * H * W); s1 = _lib.DeviceBuffer(P * H * W); sh = _lib.DeviceBuffer(P * 8)
    _lib.check(lib.fb_synth_strips_dev(ctx, P, 0, H, W, seed, max_shift, step, warp, s0.ptr, s1.ptr, sh.ptr))
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
        np.testing.assert_allclose(g['strain'], exp['strain'], rtol=1e-4, atol=1e-8)
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


def test_final_relax_huber_weights_vs_fem_oracle(fb):
    """last-round relaxation + huber residue weights (matcher.py:725-737) for a batch of pairs whose matches
    carry a smooth deformation plus gross outliers, against the exact FEM solve of the oracle, pair by pair"""
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    P, H, W = 3, 256, 1024
    m = StripBatchMatcher(P, H, W)
    rng = np.random.default_rng(5)
    t0 = np.array([[3.0, -2.0], [-7.0, 4.0], [0.0, 0.0]])
    t1 = np.array([[1.0, 0.0], [0.0, -3.0], [0.0, 0.0]])
    pid, xy0, xy1, wt = [], [], [], []
    for p in range(P):
        gx, gy = np.meshgrid(np.arange(40, W - 40, 61.0), np.arange(30, H - 30, 47.0))
        c = np.stack((gx.ravel(), gy.ravel()), -1) + t1[p]
        if p == 2:
            c = c[:5]                                                   # a sparse pair: most of the mesh is unconstrained
        f = 1.5 * np.stack((np.sin(c[:, 0] / 300.0), np.cos(c[:, 1] / 90.0)), -1) + rng.normal(0, 0.2, c.shape)
        bad = rng.random(c.shape[0]) < 0.15
        f[bad] += rng.uniform(6, 25, (bad.sum(), 1)) * np.array([[0.8, -0.6]])
        pid.append(np.full(c.shape[0], p)); xy0.append(c - 0.5 * f); xy1.append(c + 0.5 * f)
        wt.append(rng.uniform(0.35, 1.0, c.shape[0]).astype(np.float32))
    order = rng.permutation(sum(a.size for a in pid))                   # rows of a pair need not be contiguous
    pid = np.concatenate(pid)[order]; xy0 = np.concatenate(xy0)[order]; xy1 = np.concatenate(xy1)[order]; wt = np.concatenate(wt)[order]
    rw, x = m._final_relax(pid, xy0, xy1, wt, t1)
    assert m.last_relax['relres'] < 1e-8
    assert rw.min() < 0.6 and (rw == 1).sum() > rw.size // 2            # outliers damped, inliers untouched
    for p in range(P):
        s = pid == p
        uo, rwo = pipeline_ref.relax_mesh1(W, H, float(np.min(m.spacings)), t0[p], t1[p], xy0[s], xy1[s], wt[s], residue_len=5.0)
        np.testing.assert_allclose(x[p], uo, atol=1e-4)
        np.testing.assert_allclose(rw[s], rwo, atol=1e-5)
    m.free()


def test_final_relax_relaxes_grossly_deformed_region_first(fb):
    """adjust_link_weight_by_residue(relax_first=True) (matcher.py:736, optimizer.py:763-779): a cluster of confident
    but wrong matches bends mesh1 beyond the deformation cutoff; that region is relaxed on its own before the residue
    weights are taken.  Pair 1 stays below the screen and takes the plain path.  Against the oracle's exact chain."""
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    P, H, W = 3, 256, 1024
    m = StripBatchMatcher(P, H, W)
    rng = np.random.default_rng(15)
    t1 = np.array([[0.0, 0.0], [2.0, -1.0], [-3.0, 1.0]])
    pid, xy0, xy1, wt = [], [], [], []
    for p, mag in enumerate((60.0, 0.0, 40.0)):
        gx, gy = np.meshgrid(np.arange(40, W - 40, 61.0), np.arange(30, H - 30, 47.0))
        c = np.stack((gx.ravel(), gy.ravel()), -1) + t1[p]
        f = rng.normal(0, 0.1, c.shape)
        bad = (np.abs(c[:, 0] - 500) < 70) & (np.abs(c[:, 1] - 120) < 50)
        f[bad] += np.array([[mag * 0.8, -mag * 0.6]])
        w = rng.uniform(0.35, 1.0, c.shape[0]).astype(np.float32)
        w[bad] = 1.0
        pid.append(np.full(c.shape[0], p)); xy0.append(c - 0.5 * f); xy1.append(c + 0.5 * f); wt.append(w)
    pid = np.concatenate(pid); xy0 = np.concatenate(xy0); xy1 = np.concatenate(xy1); wt = np.concatenate(wt)
    rw, x = m._final_relax(pid, xy0, xy1, wt, t1)
    assert m.last_relax['relaxed_first'] == 2
    for p in range(P):
        s = pid == p
        uo, rwo = pipeline_ref.relax_mesh1(W, H, float(np.min(m.spacings)), np.zeros(2), t1[p], xy0[s], xy1[s], wt[s], residue_len=5.0)
        scale = max(np.abs(uo).max(), 1.0)
        np.testing.assert_allclose(x[p], uo, atol=1e-4 * scale)
        np.testing.assert_allclose(rw[s], rwo, atol=1e-4)
    m.free()


def test_strain_estimate_vs_fem_oracle(fb):
    """matcher.py:752-777 for a batch: rigid initialisation (fit_affine, pinned by golden G13), relaxation, sqrt(Es / Es0),
    against the oracle's exact chain pair by pair -- matches carry a small rotation and a smooth deformation"""
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    P, H, W = 3, 256, 1024
    m = StripBatchMatcher(P, H, W)
    rng = np.random.default_rng(9)
    txy = np.array([[3.0, -2.0], [-7.0, 4.0], [0.0, 0.0]])
    pid, xy0, xy1, wt = [], [], [], []
    for p in range(P):
        gx, gy = np.meshgrid(np.arange(40, W - 40, 61.0), np.arange(30, H - 30, 47.0))
        c = np.stack((gx.ravel(), gy.ravel()), -1)
        if p == 2:
            c = c[::7]
        th = 0.004 * (p + 1)
        Rm = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
        f = (c - c.mean(0)) @ Rm + c.mean(0) - c + 0.6 * np.stack((np.sin(c[:, 0] / 200.0), np.cos(c[:, 1] / 70.0)), -1)
        pid.append(np.full(c.shape[0], p)); xy0.append(c + 0.5 * f - txy[p]); xy1.append(c - 0.5 * f)
        wt.append(rng.uniform(0.35, 1.0, c.shape[0]).astype(np.float32))
    pid = np.concatenate(pid); xy0 = np.concatenate(xy0); xy1 = np.concatenate(xy1); wt = np.concatenate(wt)
    got = m._strain(pid, xy0, xy1, wt, txy)
    assert m.last_strain_solve['relres'] < 1e-5
    for p in range(P):
        s = pid == p
        exp = pipeline_ref.strain_estimate(W, H, float(np.min(m.spacings)), txy[p], xy0[s], xy1[s], wt[s])
        assert exp > 1e-4
        np.testing.assert_allclose(got[p], exp, rtol=1e-3)
    # the relaxation of the next batch needs the stiffness at the INITIAL shape again
    rw, x = m._final_relax(pid, xy0 + txy[pid], xy1, wt, np.zeros((P, 2)))
    uo, rwo = pipeline_ref.relax_mesh1(W, H, float(np.min(m.spacings)), txy[0], np.zeros(2), (xy0 + txy[pid])[pid == 0], xy1[pid == 0], wt[pid == 0], residue_len=5.0)
    np.testing.assert_allclose(x[0], uo, atol=1e-4)
    m.free()


def test_contexts_per_thread_give_identical_results(fb):
    """two host threads, each with its own context (HIP stream, arena), run the same batch concurrently: bitwise the same
    match table as the process context (shared twiddle tables are built under a lock, kernels are deterministic)"""
    import threading
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    P, H, W = 4, 1024, 256
    s0, s1, shifts = _synth(fb, P, H, W, seed=21, max_shift=9)
    ref = StripBatchMatcher(P, H, W)
    want = ref.match(s0.ptr, s1.ptr)
    out = {}

    def work(k):
        _lib.use_context(_lib.new_context())
        m = StripBatchMatcher(P, H, W)
        for _ in range(3):
            out[k] = m.match(s0.ptr, s1.ptr)
        m.free()
    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for k in range(2):
        for key in ('tx', 'ty', 'conf0', 'pair', 'xy0', 'xy1', 'weight', 'strain'):
            np.testing.assert_array_equal(out[k][key], want[key])
    ref.free()


def test_pipeline_global_translation_second_shot(fb):
    """a pair whose strips agree only inside one sixth of their area: the whole-strip NCC is not confident, the
    block-wise second shot of global_translation_matcher (matcher.py:159-221) is; same decision and numbers as the oracle"""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    from scipy.ndimage import gaussian_filter
    P, H, W = 2, 1536, 256
    rng = np.random.default_rng(77)

    def tex(h, w):
        t = gaussian_filter(rng.standard_normal((h, w)), 1.5) + 0.5 * gaussian_filter(rng.standard_normal((h, w)), 6.0)
        return np.clip(128 + 40 * t / t.std(), 0, 255).astype(np.uint8)
    s0 = np.stack([tex(H, W) for _ in range(P)])
    s1 = np.stack([tex(H, W) for _ in range(P)])                        # unrelated everywhere ...
    s1[0, 1320:1380, :] = np.roll(s0[0], (4, -6), (0, 1))[1320:1380, :]  # ... except 60 rows of pair 0
    s1[1] = np.roll(s0[1], (2, 8), (0, 1))                               # pair 1: an ordinary pair
    d0 = _lib.DeviceBuffer.from_array(s0); d1 = _lib.DeviceBuffer.from_array(s1)
    m = StripBatchMatcher(P, H, W)
    got = StripBatchMatcher.per_pair(m.match(d0.ptr, d1.ptr))
    # the second shot ran inside the C entry (pair 0 may still come back for relax_first: 60 matching rows bend its mesh)
    assert m.last_flags is not None and not (m.last_flags & ~np.uint8(4)).any()
    exp = [pipeline_ref.match_pair(s0[p], s1[p]) for p in range(P)]
    from oracle import ncc_ref
    c0 = ncc_ref.masked_dog_filter(ncc_ref.area_downsample2(s0[0]), 1.25); c1 = ncc_ref.masked_dog_filter(ncc_ref.area_downsample2(s1[0]), 1.25)
    assert ncc_ref.xcorr_fft(c0[None], c1[None], pad=True)[2][0] < 0.1          # the whole-strip shot fails on pair 0
    for p in range(P):
        assert (got[p]['tx'], got[p]['ty']) == (exp[p]['tx'], exp[p]['ty'])
        assert abs(got[p]['conf0'] - exp[p]['conf0']) < 1e-4
    # pair 0 went through the second shot: its offset is the one of the matching rows
    assert (got[0]['tx'], got[0]['ty']) == (-6.0, 4.0) and got[0]['conf0'] > 0.33
    assert (got[1]['tx'], got[1]['ty']) == (8.0, 2.0)
    m.free(); d0.free(); d1.free()


def test_pipeline_with_subpixel_warp_vs_oracle(fb):
    """SURVEY config-2 style pairs: integer offset + smooth <= 0.4 px warp.  The last-round matches then carry sub-pixel
    structure: relaxation, residue weights and a non-zero strain, all equal to the oracle's"""
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    P, H, W = 4, 1024, 256
    s0, s1, shifts = _synth(fb, P, H, W, seed=5, max_shift=10, warp=0.4)
    m = StripBatchMatcher(P, H, W)
    got = StripBatchMatcher.per_pair(m.match(s0.ptr, s1.ptr))
    h0 = s0.to_array((P, H, W), np.uint8); h1 = s1.to_array((P, H, W), np.uint8)
    nz = 0
    for p in range(P):
        exp = pipeline_ref.match_pair(h0[p], h1[p])
        g = got[p]
        assert (g['tx'], g['ty']) == (exp['tx'], exp['ty'])
        assert g['needs_host'] == exp['needs_host']
        if exp['needs_host']:
            continue
        nz += 1
        assert g['xy0'].shape == exp['xy0'].shape
        np.testing.assert_allclose(g['xy0'], exp['xy0'], atol=1e-4)
        np.testing.assert_allclose(g['xy1'], exp['xy1'], atol=1e-4)
        np.testing.assert_allclose(g['weight'], exp['weight'], atol=1e-4)
        np.testing.assert_allclose(g['strain'], exp['strain'], rtol=1e-4, atol=1e-8)
        assert exp['strain'] > 1e-5                      # the warp is seen
        assert np.abs(g['xy1'] - g['xy0'] + shifts[p]).max() < 0.75
    assert nz >= 2
    m.free()


def test_stitching_matcher_drop_in(fb):
    """the per-pair surface stitcher.py:593 calls: (xy0, xy1, weight, strain, phtm), and the no-match convention"""
    P, H, W = 2, 1024, 256
    s0, s1, shifts = _synth(fb, P, H, W, seed=31, max_shift=8, warp=0.3)
    h0 = s0.to_array((P, H, W), np.uint8); h1 = s1.to_array((P, H, W), np.uint8)
    xy0, xy1, wt, strain, phtm = fb.matcher.stitching_matcher(h0[0], h1[0], sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33)
    exp = pipeline_ref.match_pair(h0[0], h1[0])
    assert phtm is None and not exp['needs_host']
    np.testing.assert_allclose(xy0, exp['xy0'], atol=1e-4); np.testing.assert_allclose(xy1, exp['xy1'], atol=1e-4)
    np.testing.assert_allclose(wt, exp['weight'], atol=1e-4); np.testing.assert_allclose(strain, exp['strain'], rtol=1e-4, atol=1e-8)
    # unrelated strips: no match is a value, not an exception (matcher.py:278)
    out = fb.matcher.stitching_matcher(h0[0], h1[1][::-1].copy(), sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33)
    assert out[0] is None and out[1] is None and out[2] == 0.33 and out[3] is None and out[4] is None
    # spacings relative to the overlap (< 1, matcher.py:343-350): the general-mesh route, same displacement
    rel = fb.matcher.stitching_matcher(h0[0], h1[0], sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, spacings=[0.2])
    assert rel[0] is not None and np.abs(np.median(rel[1] - rel[0], axis=0) - np.median(xy1 - xy0, axis=0)).max() < 0.2
    # explicit spacings in pixels (matcher.py:252-253)
    xy0, xy1, wt, strain, _ = fb.matcher.stitching_matcher(h0[0], h1[0], sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, spacings=[60.0, 200.0])
    exp = pipeline_ref.match_pair(h0[0], h1[0], spacings=[60.0, 200.0])
    assert xy0.shape == exp['xy0'].shape
    np.testing.assert_allclose(xy0, exp['xy0'], atol=1e-4); np.testing.assert_allclose(xy1, exp['xy1'], atol=1e-4)
    np.testing.assert_allclose(wt, exp['weight'], atol=1e-4); np.testing.assert_allclose(strain, exp['strain'], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize('H,W', [(3000, 500), (400, 4000)])
def test_pipeline_readme_tile_shapes(fb, H, W):
    """the README 3000 x 4000 tiles: strips 3000 x 500 / 400 x 4000 -> non-power-of-two FFT shapes (3000 x 500 global,
    1200 x 1000 / 800 x 1600 coarse blocks, 150 x 144 / 135 x 150 padded fine blocks) on the generic mixed-radix kernels"""
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    P = 2
    s0, s1, shifts = _synth(fb, P, H, W, seed=3, max_shift=14, step=2, warp=0.3)
    m = StripBatchMatcher(P, H, W)
    got = StripBatchMatcher.per_pair(m.match(s0.ptr, s1.ptr))
    h0 = s0.to_array((P, H, W), np.uint8); h1 = s1.to_array((P, H, W), np.uint8)
    exp = pipeline_ref.match_pair(h0[0], h1[0])
    g = got[0]
    assert (g['tx'], g['ty']) == (exp['tx'], exp['ty']) == (-shifts[0, 0], -shifts[0, 1])
    assert g['needs_host'] == exp['needs_host'] == False
    assert g['xy0'].shape == exp['xy0'].shape and g['xy0'].shape[0] > 100
    np.testing.assert_allclose(g['xy0'], exp['xy0'], atol=1e-4); np.testing.assert_allclose(g['xy1'], exp['xy1'], atol=1e-4)
    np.testing.assert_allclose(g['weight'], exp['weight'], atol=1e-4)
    np.testing.assert_allclose(g['strain'], exp['strain'], rtol=1e-4, atol=1e-8)
    m.free()


def _warped_pair(H, W, seed, shift=(4, -3), warp=3.0, noise=4.0):
    """strip1(x, y) = texture(x + sx + wx, y + sy + wy) with a smooth warp of several pixels: the coarse blocks then
    disagree and the relaxation between the spacings is not a rigid translation"""
    from scipy.ndimage import gaussian_filter, map_coordinates
    rng = np.random.default_rng(seed)
    pad = 64
    tex = gaussian_filter(rng.standard_normal((H + 2 * pad, W + 2 * pad)), 1.6)
    tex += 1.8 * gaussian_filter(rng.standard_normal(tex.shape), 5.0)
    tex = 128 + 45 * tex / tex.std()
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    L = max(H, W)
    u, v = (yy, xx) if H >= W else (xx, yy)                   # u runs along the strip
    wa = warp * np.sin(2 * np.pi * u / L * 1.5 + 0.3) * np.cos(np.pi * v / L)
    wb = warp * np.cos(2 * np.pi * v / L * 1.2 + 0.7) * (0.5 + 0.5 * np.sin(2 * np.pi * u / L))
    wx, wy = (wa, wb) if H >= W else (wb, wa)
    s0 = map_coordinates(tex, [yy + pad, xx + pad], order=1)
    s1 = map_coordinates(tex, [yy + pad + shift[1] + wy, xx + pad + shift[0] + wx], order=3)
    s0 = s0 + rng.normal(0, noise, s0.shape); s1 = s1 + rng.normal(0, noise, s1.shape)
    return np.clip(np.round(s0), 0, 255).astype(np.uint8), np.clip(np.round(s1), 0, 255).astype(np.uint8)


@pytest.mark.parametrize('H,W', [(1536, 120), (120, 1536)])
def test_pipeline_deformed_mesh_between_spacings_vs_oracle(fb, H, W):
    """a warp of a few pixels makes the coarse blocks disagree, so mesh1 is relaxed into a
    non-rigid field (matcher.py:725-741); the fine round crops image 1 through the deformed mesh
    (MeshRenderer.crop_multiple tiers), locates its matches in the deformed triangles and reports them in the INITIAL
    gear.  One batch mixes deformed pairs with a rigid one.  Against the oracle's deformed branch."""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    pairs = [_warped_pair(H, W, 1, (4, -3), 3.0), _warped_pair(H, W, 2, (-6, 2), 0.0), _warped_pair(H, W, 3, (1, 5), 2.0)]
    s0 = np.stack([p[0] for p in pairs]); s1 = np.stack([p[1] for p in pairs])
    P = len(pairs)
    d0 = _lib.DeviceBuffer.from_array(s0); d1 = _lib.DeviceBuffer.from_array(s1)
    m = StripBatchMatcher(P, H, W, residue_len=2.0)
    assert m.spacings.size == 2
    res = m.match(d0.ptr, d1.ptr)
    got = StripBatchMatcher.per_pair(res)
    ndef = 0
    for p in range(P):
        exp = pipeline_ref.match_pair(s0[p], s1[p], residue_len=2.0)
        g = got[p]
        assert (g['tx'], g['ty']) == (exp['tx'], exp['ty'])
        assert g['deformed'] == bool(exp.get('deformed', False))
        assert g['xy0'].shape == exp['xy0'].shape and g['xy0'].shape[0] > 60
        if g['deformed']:
            ndef += 1
            np.testing.assert_array_equal(m.last_tiers[p], exp['tiers'][-1])
            field = m.last_field[p]
            assert np.abs(field - exp['mesh1_field']).max() < 1e-5 * max(1.0, np.abs(exp['mesh1_field']).max())
            assert np.ptp(field[:, 0]) > 0.5 or np.ptp(field[:, 1]) > 0.5        # really not a translation
        # the exact solve of the oracle and the device PCG (1e-9) give affine maps that differ at the 1e-8 level; a
        # sample position that sits on a 1/32-px rounding boundary may then flip, which moves a sub-pixel peak by ~1e-3
        np.testing.assert_allclose(g['xy0'], exp['xy0'], atol=1e-4)
        np.testing.assert_allclose(g['xy1'], exp['xy1'], atol=1e-4)
        np.testing.assert_allclose(g['weight'], exp['weight'], atol=1e-4)
        np.testing.assert_allclose(g['strain'], exp['strain'], rtol=1e-4, atol=1e-8)
    assert ndef == 2
    m.free(); d0.free(); d1.free()


@pytest.mark.parametrize('H,W', [(1536, 120), (120, 1536)])
def test_relaxation_tolerance_of_the_reference_moves_the_sample_grid_not_the_field(fb, H, W):
    """the reference stops the relaxation between spacings at a residual of 0.01 / max(1, max_dis) (matcher.py:685-688) with a
    restarted MINRES; this library converges it (1e-9, so that the result is a property of the system and equals the
    exact-solve oracle).  What the difference can do to the matches: an unconverged mesh1 puts the fine blocks somewhere
    else (by a pixel or more), but the DISPLACEMENT FIELD the matches sample is the same -- here: the matches of a run
    stopped at 1e-2 and at 2e-3 against the field interpolated from the converged run's matches"""
    from scipy.interpolate import RBFInterpolator
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    pairs = [_warped_pair(H, W, 1, (4, -3), 3.0), _warped_pair(H, W, 3, (1, 5), 2.0)]
    s0 = np.stack([p[0] for p in pairs]); s1 = np.stack([p[1] for p in pairs])
    d0 = _lib.DeviceBuffer.from_array(s0); d1 = _lib.DeviceBuffer.from_array(s1)
    runs = {}
    for tol in (1e-9, 1e-2, 2e-3):
        m = StripBatchMatcher(2, H, W, residue_len=2.0, relax_tol=tol)
        runs[tol] = StripBatchMatcher.per_pair(m.match(d0.ptr, d1.ptr))
        m.free()
    for p in range(2):
        ref = runs[1e-9][p]
        assert ref['deformed'] and ref['xy0'].shape[0] > 100
        dref = ref['xy1'] - ref['xy0']
        f = RBFInterpolator(ref['xy0'], dref, neighbors=12, smoothing=0.0, kernel='thin_plate_spline')
        # what "the same field" can mean here: the converged run against itself, one half of its matches interpolated at the other
        half = np.arange(ref['xy0'].shape[0]) % 2 == 0
        fh = RBFInterpolator(ref['xy0'][half], dref[half], neighbors=12, smoothing=0.0, kernel='thin_plate_spline')
        own = np.abs(dref[~half] - fh(ref['xy0'][~half])).max(axis=1)
        inside_lo, inside_hi = ref['xy0'].min(axis=0), ref['xy0'].max(axis=0)
        for tol in (1e-2, 2e-3):
            g = runs[tol][p]
            assert abs(g['xy0'].shape[0] - ref['xy0'].shape[0]) <= 0.05 * ref['xy0'].shape[0]
            keep = np.all((g['xy0'] >= inside_lo) & (g['xy0'] <= inside_hi), axis=1)
            err = np.abs(g['xy1'][keep] - g['xy0'][keep] - f(g['xy0'][keep])).max(axis=1)
            # measured: 0.03-0.07 px median against 0.07-0.10 px of the converged run against itself; strain within 4 %
            assert np.median(err) <= 1.5 * np.median(own) + 0.02 and np.percentile(err, 90) <= 1.5 * np.percentile(own, 90) + 0.05, \
                (tol, np.median(err), np.percentile(err, 90), np.median(own), np.percentile(own, 90))
            assert abs(g['strain'] - ref['strain']) < 0.1 * ref['strain'] + 1e-4
    d0.free(); d1.free()


def test_deformed_round_exact_field_tier_vs_oracle(fb):
    """a mesh1 with kinks that no block affine follows within 0.1 px: those blocks take the exact piecewise-linear tier
    (host field -> fb_remap_dev -> fb_ncc_batch_dev), the others the affine gather inside the NCC loaders; tiers,
    displacements and confidences against the oracle's render_blocks_mesh1 + xcorr_fft"""
    from feabas_amd import _lib, constant as const
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    from oracle import ncc_ref, fem_ref
    H, W, P = 1536, 120, 2
    pairs = [_warped_pair(H, W, 11, (3, -2), 1.0), _warped_pair(H, W, 12, (0, 0), 0.5)]
    s0 = np.stack([p[0] for p in pairs]); s1 = np.stack([p[1] for p in pairs])
    d0 = _lib.DeviceBuffer.from_array(s0); d1 = _lib.DeviceBuffer.from_array(s1)
    m = StripBatchMatcher(P, H, W)
    m._fine_dog(d0.ptr, d1.ptr)
    m._relax_system()
    v = m._mesh.vertices(const.MESH_GEAR_INITIAL)
    rng = np.random.default_rng(4)
    U = np.zeros((P,) + v.shape)
    for p in range(P):
        U[p] = np.stack((2.0 * np.sin(v[:, 1] / 300) + 0.8 * np.cos(v[:, 0] / 40), 1.5 * np.cos(v[:, 1] / 200)), -1)
        k = rng.integers(0, v.shape[0], 8)
        U[p, k] += rng.normal(0, 0.7, (8, 2))
    tx = np.array([-3.0, 0.0]); ty = np.array([2.0, 0.0])
    pad = np.zeros(P, dtype=bool)
    groups = m._match_round_deformed(tx, ty, U, np.arange(P), m.spacings[-1], m.mnb, pad, True, True)
    f0 = m.d_dogf_view.to_array((2 * P, H, W), np.float32)
    seen = 0
    for sel, bb, ddx, ddy, dcf in groups:
        for q, p in enumerate(sel):
            m1 = fem_ref.RefMesh(v, m._mesh.triangles, uid=1)
            m1.set_field(U[p], gear=(fem_ref.GEAR_FIXED, fem_ref.GEAR_MOVING))
            st1, tiers = pipeline_ref.render_blocks_mesh1(m1, f0[P + p], bb[q], 0.1, return_tiers=True)
            np.testing.assert_array_equal(m.last_tiers[int(p)], tiers)
            assert (tiers == 3).sum() >= 3 and (tiers == 2).sum() >= 3
            h, w = st1.shape[1:]
            st0 = np.stack([pipeline_ref._crop(f0[p], int(b[0] - tx[p]), int(b[1] - ty[p]), h, w) for b in bb[q]])
            ex, ey, ec = ncc_ref.xcorr_fft(st0, st1, pad=False, subpixel=True)
            good = ec > 0.5
            assert good.sum() > 0.4 * good.size and good[tiers == 3].sum() >= 2
            np.testing.assert_array_equal(np.round(ddx[q][good]), np.round(ex[good]))
            np.testing.assert_allclose(ddx[q][good], ex[good], atol=1e-4); np.testing.assert_allclose(ddy[q][good], ey[good], atol=1e-4)
            np.testing.assert_allclose(dcf[q], ec, atol=1e-4)
            seen += 1
    assert seen == P
    m.free(); d0.free(); d1.free()


def test_remap_kernel_vs_oracle(fb):
    """fb_remap_dev = common.remap (cv2.remap bilinear, constant border 0) relative to an integer origin, masked"""
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    rng = np.random.default_rng(8)
    IH, IW, N, h, w = 90, 70, 3, 33, 41
    imgs = rng.standard_normal((2, IH, IW)).astype(np.float32)
    ids = np.array([1, 0, 1], dtype=np.int32)
    org = np.array([[-5, -7], [10, 3], [30, 40]], dtype=np.int32)
    mx = rng.uniform(-8, IW + 8, (N, h, w)); my = rng.uniform(-8, IH + 8, (N, h, w))
    mk = (rng.random((N, h, w)) > 0.2)
    mxr = (mx - org[:, None, None, 0]).astype(np.float32); myr = (my - org[:, None, None, 1]).astype(np.float32)
    bufs = [_lib.DeviceBuffer.from_array(np.ascontiguousarray(a)) for a in (imgs, ids, mxr, myr, mk.astype(np.uint8), org)]
    out = _lib.DeviceBuffer(4 * N * h * w)
    _lib.check(lib.fb_remap_dev(ctx, bufs[0].ptr, IH, IW, N, bufs[1].ptr, h, w, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, bufs[5].ptr, out.ptr))
    got = out.to_array((N, h, w), np.float32)
    for n in range(N):
        exp = pipeline_ref.remap_origin(imgs[ids[n]], mx[n], my[n], (int(org[n, 0]), int(org[n, 1])))
        exp = np.where(mk[n], exp, 0)
        np.testing.assert_array_equal(got[n], exp.astype(np.float32))
    for b in bufs + [out]:
        b.free()


def test_stitching_matcher_varied_strip_shapes_and_corner_pairs(fb):
    """strips of real sections differ in shape from pair to pair (stitcher.py:561-571), corner overlaps are square
    (SURVEY config 4: 510 x 510): the per-pair surface follows the oracle on each, and the cache of per-shape device
    buffers stays bounded"""
    from feabas_amd import matcher as mt
    shapes = [(510, 510), (1010, 262), (260, 998), (1024, 256), (512, 300), (300, 512)]
    for k, (H, W) in enumerate(shapes):
        s0, s1 = _warped_pair(H, W, 40 + k, shift=(3 - k, 2 * k - 4), warp=0.3)
        xy0, xy1, wt, strain, _ = fb.matcher.stitching_matcher(s0, s1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2)
        exp = pipeline_ref.match_pair(s0, s1, residue_len=2.0)
        assert xy0.shape == exp['xy0'].shape and xy0.shape[0] >= 9
        np.testing.assert_allclose(xy0, exp['xy0'], atol=1e-4); np.testing.assert_allclose(xy1, exp['xy1'], atol=1e-4)
        np.testing.assert_allclose(wt, exp['weight'], atol=1e-4); np.testing.assert_allclose(strain, exp['strain'], rtol=1e-4, atol=1e-8)
        assert np.abs(np.median(xy1 - xy0, axis=0) + np.array([3 - k, 2 * k - 4])).max() < 0.5
    assert len(mt._pair_matchers) <= mt._PAIR_MATCHER_CACHE


@pytest.mark.parametrize('H,W', [(1023, 255), (1021, 258), (257, 1019), (511, 509)])
def test_stitching_matcher_odd_strip_sizes(fb, H, W):
    """odd strip sizes (stage jitter makes most real overlaps odd in some dimension): the x0.5 coarse image has
    cvRound(n / 2) pixels per axis and its edge cells average the pixels that exist (cv2's integer-scale area path,
    unpinned like the even case); against the oracle"""
    s0, s1 = _warped_pair(H, W, H + W, shift=(4, -5), warp=0.3)
    xy0, xy1, wt, strain, _ = fb.matcher.stitching_matcher(s0, s1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2)
    exp = pipeline_ref.match_pair(s0, s1, residue_len=2.0)
    assert xy0.shape == exp['xy0'].shape and xy0.shape[0] >= 9
    np.testing.assert_allclose(xy0, exp['xy0'], atol=1e-4); np.testing.assert_allclose(xy1, exp['xy1'], atol=1e-4)
    np.testing.assert_allclose(wt, exp['weight'], atol=1e-4); np.testing.assert_allclose(strain, exp['strain'], rtol=1e-4, atol=1e-8)
    assert np.abs(np.median(xy1 - xy0, axis=0) + np.array([4, -5])).max() < 0.5


def test_pipeline_three_spacings_deformed_twice_vs_oracle(fb):
    """a 3600 x 72 strip has three spacings (900, 150, 25): the relaxation after the first round deforms mesh1, the
    second round crops through it and relaxes AGAIN from the deformed state (the stress term of optimizer.py:1417-1418;
    the device solves for the total displacement from the FIXED gear instead), the third round crops through that
    field.  Node field after the second relaxation, tiers of the last round and the final matches against the oracle."""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    H, W = 3600, 72
    pairs = [_warped_pair(H, W, 21, (2, -3), 2.5), _warped_pair(H, W, 22, (-3, 4), 1.5)]
    s0 = np.stack([p[0] for p in pairs]); s1 = np.stack([p[1] for p in pairs])
    P = len(pairs)
    d0 = _lib.DeviceBuffer.from_array(s0); d1 = _lib.DeviceBuffer.from_array(s1)
    m = StripBatchMatcher(P, H, W, residue_len=2.0)
    assert m.spacings.size == 3
    got = StripBatchMatcher.per_pair(m.match(d0.ptr, d1.ptr))
    for p in range(P):
        exp = pipeline_ref.match_pair(s0[p], s1[p], residue_len=2.0)
        g = got[p]
        assert (g['tx'], g['ty']) == (exp['tx'], exp['ty'])
        assert g['deformed'] and exp['deformed'] and len(exp['tiers']) == 2
        np.testing.assert_array_equal(m.last_tiers[p], exp['tiers'][-1])
        assert np.abs(m.last_field[p] - exp['mesh1_field']).max() < 1e-5 * max(1.0, np.abs(exp['mesh1_field']).max())
        assert g['xy0'].shape == exp['xy0'].shape and g['xy0'].shape[0] > 200
        np.testing.assert_allclose(g['xy0'], exp['xy0'], atol=1e-4); np.testing.assert_allclose(g['xy1'], exp['xy1'], atol=1e-4)
        np.testing.assert_allclose(g['weight'], exp['weight'], atol=1e-4)
        np.testing.assert_allclose(g['strain'], exp['strain'], rtol=1e-4, atol=1e-8)
    m.free(); d0.free(); d1.free()


def test_stitching_matcher_masks_and_photometric(fb):
    """mask0 / mask1 (masked DoG with halo suppression at both scales, matcher.py:257-274, 336-337; the coarse mask is
    every second pixel) and compute_photometric (279-314) through the per-pair surface, against the oracle"""
    H, W = 1024, 256
    s0, s1 = _warped_pair(H, W, 77, shift=(-5, 3), warp=0.3)
    mask0 = np.ones((H, W), dtype=bool); mask0[100:180, 30:120] = False; mask0[700:, :40] = False
    mask1 = np.ones((H, W), dtype=bool); mask1[400:520, 150:] = False
    s0 = s0.copy(); s0[~mask0] = 0
    s1 = s1.copy(); s1[~mask1] = 0
    for m0, m1 in ((mask0, mask1), (None, mask1), (np.ones((H, W), bool), None)):
        xy0, xy1, wt, strain, phtm = fb.matcher.stitching_matcher(s0, s1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2,
                                                                   mask0=m0, mask1=m1, compute_photometric=True)
        exp = pipeline_ref.match_pair(s0, s1, residue_len=2.0, mask0=m0, mask1=m1, compute_photometric=True)
        assert xy0.shape == exp['xy0'].shape and xy0.shape[0] > 20
        np.testing.assert_allclose(xy0, exp['xy0'], atol=1e-4); np.testing.assert_allclose(xy1, exp['xy1'], atol=1e-4)
        np.testing.assert_allclose(wt, exp['weight'], atol=1e-4); np.testing.assert_allclose(strain, exp['strain'], rtol=1e-4, atol=1e-8)
        np.testing.assert_allclose(phtm, exp['phtm'], rtol=1e-5)
    # the masks matter: without them the blanked regions change the filtered images
    plain = fb.matcher.stitching_matcher(s0, s1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2)
    masked = fb.matcher.stitching_matcher(s0, s1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2, mask0=mask0, mask1=mask1)
    assert plain[4] is None and (plain[0].shape != masked[0].shape or np.abs(plain[2] - masked[2]).max() > 1e-3)


def test_stitching_matcher_batch_matches_the_per_pair_surface(fb):
    """host-resident pairs of mixed shapes (one masked) through stitching_matcher_batch (shape buckets, chunks dealt to
    host threads with their own contexts, page-locked staging): the same tuples as stitching_matcher, in input order"""
    cfg = dict(sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2, compute_photometric=True)
    shapes = [(1024, 256), (256, 1024), (1024, 256), (510, 510), (1024, 256), (256, 1024), (1024, 256)]
    pairs = []
    for k, (H, W) in enumerate(shapes):
        s0, s1 = _warped_pair(H, W, 60 + k, shift=(k - 3, 2 - k), warp=0.3)
        if k == 2:
            mk = np.ones((H, W), dtype=bool); mk[300:420, 60:200] = False
            pairs.append((s0, s1, mk, None))
        elif k == 4:
            pairs.append((s0, s1[::-1].copy()))                      # unrelated strips: the no-match tuple
        else:
            pairs.append((s0, s1))
    assert fb.matcher.stitching_matcher_batch([], **cfg) == []
    got = fb.matcher.stitching_matcher_batch(pairs, batch=2, threads=2, **cfg)
    assert len(got) == len(pairs)
    for k, pr in enumerate(pairs):
        kw = dict(cfg)
        if len(pr) > 2:
            kw.update(mask0=pr[2], mask1=pr[3])
        exp = fb.matcher.stitching_matcher(pr[0], pr[1], **kw)
        if exp[0] is None:
            assert got[k][0] is None and got[k][2] == exp[2]
            continue
        for a, b in zip(got[k][:4], exp[:4]):
            np.testing.assert_allclose(a, b, atol=1e-6)
        np.testing.assert_allclose(got[k][4], exp[4], rtol=1e-6)
    assert got[4][0] is None
    fb.matcher.stitching_matcher_batch_release()
    assert not fb.matcher._batch_workers


@pytest.mark.timeout(120)
@pytest.mark.parametrize('threads', [3, 4, 6])
def test_stitching_matcher_batch_surfaces_a_matcher_failure(fb, threads, monkeypatch):
    """a matcher thread that fails (here: the third call of StripBatchMatcher.match raises, like a device OOM in one chunk)
    must surface its error in the caller -- with loaders and several matchers the failing thread used to swallow the
    loaders' end marks and the other matchers waited for ever"""
    import threading
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    cfg = dict(sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2)
    pairs = [_warped_pair(512, 256, 90 + k, shift=(k % 3 - 1, 1), warp=0.2) for k in range(12)]
    calls = [0]
    lock = threading.Lock()
    real = StripBatchMatcher.match

    def flaky(self, *a, **kw):
        with lock:
            calls[0] += 1
            n = calls[0]
        if n == 3:
            raise RuntimeError('injected failure in one chunk')
        return real(self, *a, **kw)
    monkeypatch.setattr(StripBatchMatcher, 'match', flaky)
    with pytest.raises(RuntimeError, match='injected failure'):
        fb.matcher.stitching_matcher_batch(pairs, batch=2, threads=threads, **cfg)
    monkeypatch.setattr(StripBatchMatcher, 'match', real)
    # the workers are reusable afterwards
    got = fb.matcher.stitching_matcher_batch(pairs[:4], batch=2, threads=threads, **cfg)
    assert len(got) == 4 and all(g[0] is not None for g in got)
    fb.matcher.stitching_matcher_batch_release()


def test_config0_readme_grid_matching_stage(fb):
    """BASELINE config[0] (plumbing): the README's 3 x 2 grid of 3000 x 4000 tiles at 10 % overlap, stage jitter of up to
    15 px, through the host mirror of the matching stage (stitcher.find_overlaps -> match_list_of_overlaps): 11 overlaps
    (4 left-right, 3 up-down, 4 diagonal corners), every one matched, and the matches carry the injected jitter"""
    from scipy.ndimage import gaussian_filter
    from feabas_amd import stitcher
    rng = np.random.default_rng(0)
    TH, TW = 3000, 4000
    nom = np.array([[x, y] for y in (0, 2700) for x in (0, 3600, 7200)])
    jit = rng.integers(-15, 16, nom.shape)
    pad = 32
    canvas = gaussian_filter(rng.standard_normal((2700 + TH + 2 * pad, 7200 + TW + 2 * pad)).astype(np.float32), 1.5)
    canvas += 1.5 * gaussian_filter(rng.standard_normal(canvas.shape).astype(np.float32), 6.0)
    canvas = 128 + 40 * canvas / canvas.std()
    tiles = []
    for (x, y), (jx, jy) in zip(nom, jit):
        t = canvas[pad + y + jy:pad + y + jy + TH, pad + x + jx:pad + x + jx + TW] + rng.normal(0, 5, (TH, TW))
        tiles.append(np.clip(np.round(t), 0, 255).astype(np.uint8))
    bboxes = np.concatenate((nom, nom + np.array([TW, TH])), axis=1)
    overlaps = stitcher.find_overlaps(bboxes, tile_size=(TH, TW))
    assert overlaps.shape == (11, 2)
    cfg = dict(sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2, residue_mode='huber', pad=True, spacings=None,
               fine_downsample=1.0, compute_photometric=False)                  # default_stitching_configs.yaml:13-23
    matches, strains, phtm, err = stitcher.match_list_of_overlaps(overlaps, tiles, bboxes, min_overlap_width=25, margin=100,
                                                                  matcher_config=cfg, batch=4, threads=2)
    assert not err and len(matches) == 11 and not phtm
    for (i, j), (xy0, xy1, wt) in matches.items():
        exp = (nom[i] + jit[i]) - (nom[j] + jit[j])          # same canvas point: p_j - p_i
        d = np.median(xy1 - xy0, axis=0)
        assert np.abs(d - exp).max() < 0.5, ((i, j), d, exp)
        assert xy0.shape[0] >= 9 and strains[(i, j)] < 0.01
        assert xy0.min() >= -1 and (xy0.max(axis=0) <= np.array([TW, TH])).all()
    # one left-right pair against the oracle, strip for strip
    i, j = (int(v) for v in overlaps[np.argmax([abs(nom[a][0] - nom[b][0]) == 3600 and nom[a][1] == nom[b][1] for a, b in overlaps])])
    bb_ov, wd = stitcher.bbox_intersections(bboxes[i], bboxes[j])
    bb_ov = bb_ov + np.array([-100, -100, 100, 100])
    b0 = stitcher.bbox_intersections(bb_ov, bboxes[i])[0]; b1 = stitcher.bbox_intersections(bb_ov, bboxes[j])[0]
    s0 = tiles[i][b0[1] - bboxes[i][1]:b0[3] - bboxes[i][1], b0[0] - bboxes[i][0]:b0[2] - bboxes[i][0]]
    s1 = tiles[j][b1[1] - bboxes[j][1]:b1[3] - bboxes[j][1], b1[0] - bboxes[j][0]:b1[2] - bboxes[j][0]]
    assert s0.shape == s1.shape == (3000, 500)
    exp = pipeline_ref.match_pair(s0, s1, residue_len=2.0)
    xy0, xy1, wt = matches[(i, j)]
    np.testing.assert_allclose(xy0 - (b0[:2] - bboxes[i][:2]), exp['xy0'], atol=1e-4)
    np.testing.assert_allclose(xy1 - (b1[:2] - bboxes[j][:2]), exp['xy1'], atol=1e-4)
    np.testing.assert_allclose(wt, exp['weight'], atol=1e-4)
    fb.matcher.stitching_matcher_batch_release()
    # the match table crosses to the optimisation stage as the Stitcher HDF5 file (stitcher.py:126-222): written and read back
    import os, tempfile
    from feabas_amd import h5wire
    with tempfile.TemporaryDirectory() as tmp:
        fn = os.path.join(tmp, 'section.h5')
        h5wire.save_stitcher_h5(fn, '/data', 4.0, [f'tile_{k}.png' for k in range(len(tiles))], bboxes, matches, strains)
        m2, s2, _ = h5wire.load_stitcher_matches(fn)
    assert set(m2) == set(matches)
    for key in matches:
        for a_, b_ in zip(matches[key], m2[key]):
            np.testing.assert_array_equal(np.asarray(a_, dtype=np.float32), b_)
        assert s2[key] == np.float32(strains[key])
    matches = {key: tuple(np.asarray(v, dtype=np.float64) for v in val) for key, val in m2.items()}      # what --mode optimization sees
    # ... and the optimisation stage on top of these matches (stitcher.py:1012-1018): one mesh per tile at its nominal stage
    # position, links from the matches, SLM.optimize_linear on the device.  The tiles must end at their TRUE positions
    # (nominal + jitter) relative to tile 0 -- both hot paths, end to end.
    from feabas_amd import mesh, optimizer, constant as const
    meshes = []
    for k in range(len(tiles)):
        mk = mesh.Mesh.from_bbox((0, 0, TW, TH), cartesian=True, mesh_size=300.0, uid=k)
        mk.apply_translation(nom[k].astype(np.float64), const.MESH_GEAR_FIXED)
        meshes.append(mk)
    meshes[0].lock()
    slm = optimizer.SLM(meshes, stiffness_lambda=1.0, crosslink_lambda=-1.0)
    for (i, j), (xy0, xy1, wt) in matches.items():
        assert slm.add_link_from_coordinates(i, j, xy0, xy1, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_INITIAL), weight=wt)
    cost = slm.optimize_linear(tol=1e-8)
    assert cost[1] < 1e-5 * cost[0]
    ctr = np.array([mk.vertices_w_offset(const.MESH_GEAR_MOVING).mean(axis=0) for mk in meshes])
    rel = (ctr - ctr[0]) - (nom - nom[0])                    # what the optimisation moved every tile by, relative to tile 0
    np.testing.assert_allclose(rel, jit - jit[0], atol=0.3)   # = the injected stage jitter


@pytest.mark.parametrize('shapes,cds', [([(1024, 256), (1020, 250), (1016, 252), (1030, 262), (1024, 256)], 0.5),
                                        ([(1024, 256), (1021, 251), (1016, 252)], 1),
                                        ([(3000, 500), (2990, 496), (3011, 505)], 0.5)])
def test_ragged_batch_matches_the_per_pair_surface(fb, shapes, cds):
    """strips of unequal size in ONE batch (RaggedStripBatchMatcher: padded slots, per-image extents in the downsample / DoG
    kernels, per-pair block grids, spacings and mesh geometry inside a shared system) against the same pairs through
    matchers of their own shape"""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import RaggedStripBatchMatcher, StripBatchMatcher
    keys = {RaggedStripBatchMatcher.bucket_key(h, w) for h, w in shapes}
    assert len(keys) == 1
    pairs = [_warped_pair(h, w, 90 + k, shift=(3 - 2 * k, k - 2), warp=0.3) for k, (h, w) in enumerate(shapes)]
    P = len(shapes)
    Hm, Wm = max(h for h, _ in shapes), max(w for _, w in shapes)
    stage = np.zeros((2, P, Hm, Wm), dtype=np.uint8)
    for k, (a, b) in enumerate(pairs):
        stage[0, k, :a.shape[0], :a.shape[1]] = a
        stage[1, k, :b.shape[0], :b.shape[1]] = b
    dev = _lib.DeviceBuffer.from_array(stage)
    junk = _lib.DeviceBuffer.from_array(np.full(4 * 2 * P * Hm * Wm, 77, dtype=np.uint8))     # dirty memory for the pool to hand out
    junk.free()
    m = RaggedStripBatchMatcher(shapes, residue_len=2.0, coarse_downsample=cds)
    got = StripBatchMatcher.per_pair(m.match(dev.ptr, dev.offset(P * Hm * Wm)))
    cfg = dict(sigma=2.5, coarse_downsample=cds, conf_thresh=0.33, residue_len=2)
    for k, (a, b) in enumerate(pairs):
        exp = fb.matcher.stitching_matcher(a, b, **cfg)
        g = got[k]
        assert not g['deferred'] and g['xy0'] is not None and g['xy0'].shape == exp[0].shape
        np.testing.assert_allclose(g['xy0'], exp[0], atol=1e-5); np.testing.assert_allclose(g['xy1'], exp[1], atol=1e-5)
        np.testing.assert_allclose(g['weight'], exp[2], atol=1e-5); np.testing.assert_allclose(g['strain'], exp[3], rtol=1e-4, atol=1e-7)
    m.free(); dev.free()


@pytest.mark.parametrize('cds', [0.5, 1])
def test_ragged_batch_photometric_statistics_match_the_per_pair_surface(fb, cds):
    """compute_photometric (matcher.py:279-314) in a batch of strips of unequal size: the statistics of every pair over the overlap of
    ITS OWN translated extents, against the same pairs through stitching_matcher one by one"""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import RaggedStripBatchMatcher, StripBatchMatcher
    shapes = [(1536, 120), (1526, 122), (1520, 120), (1530, 122)]
    assert len({RaggedStripBatchMatcher.bucket_key(h, w) for h, w in shapes}) == 1
    pairs = [_warped_pair(h, w, 310 + k, shift=(5 - 3 * k, 2 * k - 3), warp=0.3) for k, (h, w) in enumerate(shapes)]
    P = len(shapes)
    Hm, Wm = max(h for h, _ in shapes), max(w for _, w in shapes)
    stage = np.full((2, P, Hm, Wm), 200, dtype=np.uint8)             # the padding of a slot must not enter the statistics
    for k, (a, b) in enumerate(pairs):
        stage[0, k, :a.shape[0], :a.shape[1]] = a
        stage[1, k, :b.shape[0], :b.shape[1]] = b
    dev = _lib.DeviceBuffer.from_array(stage)
    m = RaggedStripBatchMatcher(shapes, residue_len=2.0, coarse_downsample=cds)
    res = m.match(dev.ptr, dev.offset(P * Hm * Wm), compute_photometric=True)
    got = StripBatchMatcher.per_pair(res)
    cfg = dict(sigma=2.5, coarse_downsample=cds, conf_thresh=0.33, residue_len=2, compute_photometric=True)
    for k, (a, b) in enumerate(pairs):
        exp = fb.matcher.stitching_matcher(a, b, **cfg)
        assert got[k]['xy0'] is not None and got[k]['xy0'].shape == exp[0].shape
        np.testing.assert_allclose(got[k]['xy0'], exp[0], atol=1e-4)
        assert exp[4] is not None and res['phtm'][k] is not None
        np.testing.assert_allclose(res['phtm'][k], exp[4], rtol=1e-6)
    m.free(); dev.free()


@pytest.mark.parametrize('cds', [0.5, 1])
def test_ragged_batch_with_masks_matches_the_per_pair_surface(fb, cds):
    """masked pairs (matcher.py:257-274, 336-337: the masked DoG of the coarse and of the fine images) inside a batch of strips of
    unequal size -- every masked image filtered on its own extent of its slot -- with the photometric statistics over the masked
    overlap, against the same pairs through stitching_matcher one by one"""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import RaggedStripBatchMatcher, StripBatchMatcher
    shapes = [(1536, 120), (1526, 122), (1520, 120), (1530, 122)]
    assert len({RaggedStripBatchMatcher.bucket_key(h, w) for h, w in shapes}) == 1
    pairs = [_warped_pair(h, w, 410 + k, shift=(4 - 2 * k, 2 * k - 3), warp=0.3) for k, (h, w) in enumerate(shapes)]
    P = len(shapes)
    masks0, masks1 = [None] * P, [None] * P
    m = np.ones(shapes[0], dtype=bool); m[300:420, 30:90] = False; m[:40, :] = False
    masks0[0] = m
    m = np.ones(shapes[2], dtype=bool); m[900:1100, :50] = False
    masks1[2] = m
    m = np.ones(shapes[3], dtype=bool); m[1400:, 60:] = False
    masks0[3] = m; masks1[3] = ~np.zeros(shapes[3], dtype=bool)                 # (a mask without a zero changes nothing)
    pairs = [(np.where(masks0[k], a, 0).astype(np.uint8) if masks0[k] is not None else a,
              np.where(masks1[k], b, 0).astype(np.uint8) if masks1[k] is not None else b) for k, (a, b) in enumerate(pairs)]
    Hm, Wm = max(h for h, _ in shapes), max(w for _, w in shapes)
    stage = np.full((2, P, Hm, Wm), 200, dtype=np.uint8)
    for k, (a, b) in enumerate(pairs):
        stage[0, k, :a.shape[0], :a.shape[1]] = a
        stage[1, k, :b.shape[0], :b.shape[1]] = b
    dev = _lib.DeviceBuffer.from_array(stage)
    mt = RaggedStripBatchMatcher(shapes, residue_len=2.0, coarse_downsample=cds)
    res = mt.match(dev.ptr, dev.offset(P * Hm * Wm), masks0=masks0, masks1=masks1, compute_photometric=True)
    got = StripBatchMatcher.per_pair(res)
    cfg = dict(sigma=2.5, coarse_downsample=cds, conf_thresh=0.33, residue_len=2, compute_photometric=True)
    for k, (a, b) in enumerate(pairs):
        exp = fb.matcher.stitching_matcher(a, b, mask0=masks0[k], mask1=masks1[k], **cfg)
        assert exp[0] is not None and got[k]['xy0'] is not None and got[k]['xy0'].shape == exp[0].shape
        np.testing.assert_allclose(got[k]['xy0'], exp[0], atol=1e-4); np.testing.assert_allclose(got[k]['xy1'], exp[1], atol=1e-4)
        np.testing.assert_allclose(got[k]['weight'], exp[2], atol=1e-4)
        np.testing.assert_allclose(res['phtm'][k], exp[4], rtol=1e-6)
    mt.free(); dev.free()


def test_ragged_batch_with_deformed_meshes_matches_the_per_pair_surface(fb):
    """strips of unequal size whose mesh1 is relaxed into a non-rigid field between the spacings, in ONE batch (per-pair node
    grids and tolerances in fb_deformed_block_affines / fb_deformed_locate), against matchers of their own shape"""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import RaggedStripBatchMatcher, StripBatchMatcher
    shapes = [(1536, 120), (1526, 122), (1520, 120), (1536, 120)]
    assert len({RaggedStripBatchMatcher.bucket_key(h, w) for h, w in shapes}) == 1
    warps = [3.0, 2.0, 0.0, 2.5]
    pairs = [_warped_pair(h, w, 130 + k, shift=(4 - k, k - 3), warp=warps[k]) for k, (h, w) in enumerate(shapes)]
    P = len(shapes)
    Hm, Wm = max(h for h, _ in shapes), max(w for _, w in shapes)
    stage = np.full((2, P, Hm, Wm), 200, dtype=np.uint8)             # the padding of a slot is never read
    for k, (a, b) in enumerate(pairs):
        stage[0, k, :a.shape[0], :a.shape[1]] = a
        stage[1, k, :b.shape[0], :b.shape[1]] = b
    dev = _lib.DeviceBuffer.from_array(stage)
    m = RaggedStripBatchMatcher(shapes, residue_len=2.0)
    res = m.match(dev.ptr, dev.offset(P * Hm * Wm))
    got = StripBatchMatcher.per_pair(res)
    assert res['deformed'].tolist() == [True, True, False, True] and not res['deferred'].any()
    cfg = dict(sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2)
    for k, (a, b) in enumerate(pairs):
        exp = fb.matcher.stitching_matcher(a, b, **cfg)
        g = got[k]
        assert g['xy0'] is not None and g['xy0'].shape == exp[0].shape and g['xy0'].shape[0] > 60
        np.testing.assert_allclose(g['xy0'], exp[0], atol=1e-4); np.testing.assert_allclose(g['xy1'], exp[1], atol=1e-4)
        np.testing.assert_allclose(g['weight'], exp[2], atol=1e-4); np.testing.assert_allclose(g['strain'], exp[3], rtol=1e-4, atol=1e-8)
    m.free(); dev.free()


def test_stitching_matcher_threshold_residue_mode(fb):
    """residue_mode='threshold' (matcher.py:732-733, optimizer.py:198-200): matches whose residue after the relaxation
    exceeds residue_len are cut -- weight 0, masked out of the returned table (Link.mask) -- instead of damped.  A band of
    strip 1 is displaced by 14 px: its blocks match confidently but inconsistently with the mesh."""
    H, W = 1024, 256
    s0, s1 = _warped_pair(H, W, 210, shift=(3, -2), warp=0.2)
    s1 = s1.copy()
    s1[470:545] = np.roll(s1[470:545], 14, axis=1)
    for mode in ('threshold', 'huber'):
        xy0, xy1, wt, strain, _ = fb.matcher.stitching_matcher(s0, s1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2,
                                                               residue_mode=mode)
        exp = pipeline_ref.match_pair(s0, s1, residue_len=2.0, residue_mode=mode)
        assert xy0.shape == exp['xy0'].shape
        np.testing.assert_allclose(xy0, exp['xy0'], atol=1e-4); np.testing.assert_allclose(xy1, exp['xy1'], atol=1e-4)
        np.testing.assert_allclose(wt, exp['weight'], atol=1e-4); np.testing.assert_allclose(strain, exp['strain'], rtol=1e-4, atol=1e-8)
        if mode == 'threshold':
            n_thr = xy0.shape[0]
            assert (wt > 0).all()
            assert np.abs(xy1 - xy0 + np.array([3, -2])).max() < 6            # the displaced band is gone
        else:
            assert xy0.shape[0] > n_thr and wt.min() < 0.6                    # ... where huber only damps it
    with pytest.raises(ValueError):
        fb.matcher.stitching_matcher(s0, s1, residue_mode='none')


def test_stitching_matcher_unequal_strip_shapes(fb):
    """the two crops of an overlap need not have the same size (an overlap clipped by a tile border on one side only;
    the reference takes np.minimum of the shapes for its spacings and works on both crops as they are, matcher.py:244,
    354-363): such a pair goes through the general-mesh route.  Strip 1 of an ordinary pair is cropped by 6 rows / 4 columns
    at its far ends: the matches inside the common area must be those of the uncropped pair (same coordinates: the crop keeps
    the origin), through the per-pair surface and through the batch surface, which mixes it with equal-shape pairs."""
    H, W = 1024, 256
    s0, s1 = _warped_pair(H, W, 77, shift=(5, -4), warp=0.3)
    full = fb.matcher.stitching_matcher(s0, s1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2)
    s1c = np.ascontiguousarray(s1[:H - 6, :W - 4])
    cut = fb.matcher.stitching_matcher(s0, s1c, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2)
    assert cut[0] is not None and cut[0].shape[0] >= 0.8 * full[0].shape[0]
    d_full = np.median(full[1] - full[0], axis=0); d_cut = np.median(cut[1] - cut[0], axis=0)
    assert np.abs(d_full + np.array([5, -4])).max() < 0.3 and np.abs(d_cut - d_full).max() < 0.15       # the same displacement field
    assert np.all(cut[1][:, 0] <= W - 4) and np.all(cut[1][:, 1] <= H - 6)                              # matches inside the smaller strip
    assert cut[2].min() > 0.3 and abs(cut[3] - full[3]) < 0.01
    t0, t1 = _warped_pair(H, W, 78, shift=(-3, 2), warp=0.3)
    outs = fb.matcher.stitching_matcher_batch([(s0, s1), (s0, s1c), (t0, t1)], batch=2, threads=1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2)
    np.testing.assert_allclose(outs[0][0], full[0], atol=1e-5); np.testing.assert_allclose(outs[1][0], cut[0], atol=1e-5)
    assert outs[2][0] is not None and np.abs(np.median(outs[2][1] - outs[2][0], axis=0) + np.array([-3, 2])).max() < 0.3
    # spacings relative to the overlap (< 1, matcher.py:343-350) take the same route
    rel = fb.matcher.stitching_matcher(s0, s1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2, spacings=[0.25, 75])
    assert rel[0] is not None and np.abs(np.median(rel[1] - rel[0], axis=0) - d_full).max() < 0.15
    # ... from the batch entry too (pair by pair: the spacings follow every pair's own overlap), and they are the pixel spacings
    # they resolve to: 0.25 x the longer side of the overlap of the translated strips
    outs = fb.matcher.stitching_matcher_batch([(s0, s1), (t0, t1)], batch=2, threads=1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2,
                                              spacings=[0.25, 75])
    np.testing.assert_allclose(outs[0][0], rel[0], atol=1e-9); np.testing.assert_allclose(outs[0][1], rel[1], atol=1e-9)
    assert outs[1][0] is not None and np.abs(np.median(outs[1][1] - outs[1][0], axis=0) + np.array([-3, 2])).max() < 0.3
    g_tx, g_ty = np.round(-d_full)
    side = max(W - abs(g_tx), H - abs(g_ty))
    pix = fb.matcher.stitching_matcher(s0, s1, sigma=2.5, coarse_downsample=0.5, conf_thresh=0.33, residue_len=2, spacings=[0.25 * side, 75])
    assert pix[0].shape == rel[0].shape
    np.testing.assert_allclose(pix[0], rel[0], atol=2e-3); np.testing.assert_allclose(pix[1], rel[1], atol=2e-3)


@pytest.mark.parametrize('H,W,P,cds,mode', [(1024, 256, 10, 0.5, 'huber'), (256, 1024, 7, 0.5, 'threshold'), (640, 200, 6, 1, 'huber'),
                                             (1023, 255, 5, 0.5, 'huber')])
def test_native_entry_equals_the_host_statement(fb, H, W, P, cds, mode):
    """fb_match_strips (the whole stitching_matcher sequence behind one C entry) against the numpy statement of the same
    sequence in stitch_pipeline.py: identical tables, including a pair without a match (second shot of
    global_translation_matcher, matcher.py:159-221) and one whose mesh1 deforms between the spacings."""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    s0, s1, shifts = _synth(fb, P, H, W, seed=23, max_shift=14, warp=0.3)
    h0 = s0.to_array((P, H, W), np.uint8); h1 = s1.to_array((P, H, W), np.uint8)
    rng = np.random.default_rng(5)
    h1[1] = rng.integers(0, 256, (H, W), dtype=np.uint8)                   # nothing to find
    w0, w1 = _warped_pair(H, W, 3, (3, -2), 3.0)
    h0[2], h1[2] = w0, w1                                                  # non-rigid between the spacings
    d0 = _lib.DeviceBuffer.from_array(h0); d1 = _lib.DeviceBuffer.from_array(h1)
    kw = dict(coarse_downsample=cds, residue_mode=mode, residue_len=3.0)
    mn = StripBatchMatcher(P, H, W, route='native', **kw)
    mh = StripBatchMatcher(P, H, W, route='host', **kw)
    rn = mn.match(d0.ptr, d1.ptr); rh = mh.match(d0.ptr, d1.ptr)
    # nothing is handed back: the pair without a match takes the second shot of global_translation_matcher inside the entry
    assert not mn.last_flags.any()
    if mn.spacings.size > 1:
        assert rn['deformed'][2]
    for k in ('tx', 'ty', 'conf0', 'valid', 'deformed'):
        np.testing.assert_array_equal(rn[k], rh[k], err_msg=k)
    assert rn['valid'].sum() >= P - 2
    gn = StripBatchMatcher.per_pair(rn); gh = StripBatchMatcher.per_pair(rh)
    for p in range(P):
        if not rh['valid'][p]:
            assert gn[p]['xy0'] is None
            continue
        # (the solve of a deformed pair shares its PCG scalars with the batch: the entry composes it without the pair it hands back)
        exact = not mn.last_flags[p] and not rn['deformed'][p]
        for k in ('xy0', 'xy1', 'weight'):
            if exact:
                np.testing.assert_array_equal(gn[p][k], gh[p][k], err_msg=f'{k} of pair {p}')
            else:
                np.testing.assert_allclose(gn[p][k], gh[p][k], atol=1e-6, err_msg=f'{k} of pair {p}')
        np.testing.assert_allclose(gn[p]['strain'], gh[p]['strain'], rtol=1e-6, atol=1e-10)
    # a second call reuses everything that is resident
    rn2 = mn.match(d0.ptr, d1.ptr)
    np.testing.assert_array_equal(rn2['xy0'], rn['xy0']); np.testing.assert_array_equal(rn2['weight'], rn['weight'])
    mn.free(); mh.free(); d0.free(); d1.free(); s0.free(); s1.free()


@pytest.mark.parametrize('cds', [0.5, 1])
def test_native_entry_masks_and_photometric_equal_the_host_statement(fb, cds):
    """fb_strip_matcher_set_extras: valid-pixel masks (masked DoG of both scales, matcher.py:257-274, 336-337) and the
    photometric statistics (279-314) inside the C entry against the numpy statement of the same batch: identical match
    tables, statistics to the rounding of their float32 / float64 means"""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import StripBatchMatcher
    H, W, P = 1024, 256, 5
    s0, s1, _ = _synth(fb, P, H, W, seed=41, max_shift=10, warp=0.3)
    rng = np.random.default_rng(3)
    masks0 = [None] * P; masks1 = [None] * P
    mk = np.ones((H, W), dtype=np.uint8); mk[:, :40] = 0; mk[700:, :] = 0; masks0[1] = mk
    mk = np.ones((H, W), dtype=bool); mk[100:300, 100:] = False; masks1[1] = mk
    mk = np.ones((H, W), dtype=np.uint8); mk[:90] = 0; masks1[3] = mk * 255
    masks0[4] = np.ones((H, W), dtype=np.uint8)                            # a mask without a zero changes nothing
    kw = dict(coarse_downsample=cds, residue_len=3.0)
    mn = StripBatchMatcher(P, H, W, route='native', **kw); mh = StripBatchMatcher(P, H, W, route='host', **kw)
    rn = mn.match(s0.ptr, s1.ptr, masks0, masks1, compute_photometric=True)
    rh = mh.match(s0.ptr, s1.ptr, masks0, masks1, compute_photometric=True)
    assert not mn.last_flags.any() and rn['valid'].all()
    for k in ('tx', 'ty', 'conf0', 'valid', 'pair', 'xy0', 'xy1', 'weight'):
        np.testing.assert_array_equal(rn[k], rh[k], err_msg=k)
    np.testing.assert_allclose(rn['strain'], rh['strain'], rtol=1e-6, atol=1e-10)
    for p in range(P):
        np.testing.assert_allclose(rn['phtm'][p], rh['phtm'][p], rtol=2e-6)
    # the masked pair really differs from the unmasked run, and the extras do not carry over to the next call
    r0 = mn.match(s0.ptr, s1.ptr)
    assert r0['phtm'] is None and not np.array_equal(StripBatchMatcher.per_pair(r0)[1]['xy0'], StripBatchMatcher.per_pair(rn)[1]['xy0'])
    np.testing.assert_array_equal(StripBatchMatcher.per_pair(r0)[0]['xy0'], StripBatchMatcher.per_pair(rn)[0]['xy0'])
    # photometric statistics alone; a fully masked strip 0 has no statistics (None, matcher.py:296-298)
    none0 = [np.zeros((H, W), dtype=np.uint8)] + [None] * (P - 1)
    rp = mn.match(s0.ptr, s1.ptr, none0, None, compute_photometric=True)
    rq = mh.match(s0.ptr, s1.ptr, none0, None, compute_photometric=True)
    assert rp['phtm'][0] is None and rq['phtm'][0] is None
    np.testing.assert_allclose(rp['phtm'][2], rn['phtm'][2], rtol=1e-12)
    mn.free(); mh.free(); s0.free(); s1.free()


def test_native_entry_second_shot_on_unequal_strips(fb):
    """the second shot of global_translation_matcher (matcher.py:159-221) inside the entry for strips of unequal size: every
    pair has its own 6 x 1 block grid; one pair agrees only inside a sixth of its area (the shot succeeds), one nowhere (it
    stays without a match), the others are ordinary.  Against the numpy statement of the ragged batch, bit for bit."""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import RaggedStripBatchMatcher, StripBatchMatcher
    shapes = [(1536, 120), (1526, 122), (1520, 120), (1530, 121)]
    pairs = [list(_warped_pair(h, w, 310 + k, shift=(3 - k, k - 2), warp=0.2)) for k, (h, w) in enumerate(shapes)]
    rng = np.random.default_rng(9)
    h1, w1 = shapes[1]
    other = _warped_pair(h1, w1, 999, shift=(0, 0), warp=0.0)[0]
    keep = pairs[1][1][1300:1380].copy()
    pairs[1][1] = other.copy(); pairs[1][1][1300:1380] = keep                # pair 1: 80 matching rows, the rest unrelated
    pairs[2][1] = rng.integers(0, 256, shapes[2], dtype=np.uint8)           # pair 2: nothing to find
    P = len(shapes)
    Hm, Wm = max(h for h, _ in shapes), max(w for _, w in shapes)
    stage = np.full((2, P, Hm, Wm), 200, dtype=np.uint8)
    for k, (a, b) in enumerate(pairs):
        stage[0, k, :a.shape[0], :a.shape[1]] = a; stage[1, k, :b.shape[0], :b.shape[1]] = b
    dev = _lib.DeviceBuffer.from_array(stage)
    mn = RaggedStripBatchMatcher(shapes, residue_len=2.0, route='native'); mh = RaggedStripBatchMatcher(shapes, residue_len=2.0, route='host')
    rn = mn.match(dev.ptr, dev.offset(P * Hm * Wm)); rh = mh.match(dev.ptr, dev.offset(P * Hm * Wm))
    assert not mn.last_flags.any()
    for k in ('tx', 'ty', 'conf0', 'valid', 'pair', 'xy0', 'xy1', 'weight'):
        np.testing.assert_array_equal(rn[k], rh[k], err_msg=k)
    assert rn['valid'].tolist() == [True, True, False, True]
    assert rn["tx"][1] == -2.0 and abs(rn["ty"][1] - 1.0) <= 1.0 and rn["conf0"][1] > 0.33     # found by a block of the second shot (coarse scale: even offsets)
    mn.free(); mh.free(); dev.free()


def test_native_entry_randomised_sweep(fb):
    """tools/fuzz_native.py, a short run: random shapes / options / masks / statistics, hard pairs (no texture, 2.5 px warp)
    and ragged batches through fb_match_strips against the numpy statement -- no mismatching batch"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_native.py'), '17', '8'], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert 'mismatching batches 0' in last and 'MISMATCH' not in out.stdout, out.stdout[-2000:]


def test_native_entry_automatic_spacings_and_grid(fb):
    """fb_strip_matcher_create without spacings restates matcher.py:243-251; its relaxation grid is Mesh.from_bbox's"""
    from feabas_amd import _lib
    from feabas_amd.matcher import auto_spacings
    from feabas_amd.stitch_pipeline import grid_counts
    lib, ctx = _lib.load(), _lib.ctx()
    for H, W in [(4096, 510), (510, 4096), (256, 256), (90, 1200), (3000, 500), (1023, 255)]:
        o = _lib.StripOpts(2.5, 1, 0.33, 2, 2, 5.0, 0, 1.0, 1e-9, 1, 0, None)
        h = C.c_void_p()
        _lib.check(lib.fb_strip_matcher_create(ctx, 1, H, W, C.byref(o), C.byref(h)))
        nsp = C.c_int(64); sp = np.zeros(64)
        _lib.check(lib.fb_strip_matcher_info(ctx, h, C.byref(nsp), _lib.ptr(sp), None, None, None, None, None, None))
        exp = np.sort(auto_spacings((H, W), (H, W)))[::-1]
        np.testing.assert_allclose(sp[:nsp.value], exp, rtol=1e-14)
        lib.fb_strip_matcher_destroy(ctx, h)
    with pytest.raises(RuntimeError):
        bad = np.array([0.5])
        o = _lib.StripOpts(2.5, 1, 0.33, 2, 2, 5.0, 0, 1.0, 1e-9, 1, 1, bad.ctypes.data)
        _lib.check(lib.fb_strip_matcher_create(ctx, 1, 256, 256, C.byref(o), C.byref(C.c_void_p())))


@pytest.mark.parametrize('cds', [0.5, 1])
def test_native_entry_ragged_equals_the_host_statement(fb, cds):
    """fb_strip_matcher_create_ragged: strips of unequal size through the C entry against the numpy statement of the ragged
    batch -- per-pair extents, block grids, spacings and mesh geometry; one pair deforms between the spacings and takes the
    deformed-mesh branch inside the entry (node field, tiers and matches bit-identical to the host statement's)"""
    from feabas_amd import _lib
    from feabas_amd.stitch_pipeline import RaggedStripBatchMatcher, StripBatchMatcher
    shapes = [(1536, 120), (1526, 122), (1520, 120), (1530, 121), (1536, 120), (1522, 121)]
    assert len({RaggedStripBatchMatcher.bucket_key(h, w) for h, w in shapes}) == 1
    warps = [0.3, 0.0, 2.5, 0.2, 0.4, 0.1]
    pairs = [_warped_pair(h, w, 210 + k, shift=(4 - k, k - 3), warp=warps[k]) for k, (h, w) in enumerate(shapes)]
    P = len(shapes)
    Hm, Wm = max(h for h, _ in shapes), max(w for _, w in shapes)
    stage = np.full((2, P, Hm, Wm), 200, dtype=np.uint8)
    for k, (a, b) in enumerate(pairs):
        stage[0, k, :a.shape[0], :a.shape[1]] = a
        stage[1, k, :b.shape[0], :b.shape[1]] = b
    dev = _lib.DeviceBuffer.from_array(stage)
    mn = RaggedStripBatchMatcher(shapes, residue_len=2.0, coarse_downsample=cds, route='native')
    mh = RaggedStripBatchMatcher(shapes, residue_len=2.0, coarse_downsample=cds, route='host')
    rn = mn.match(dev.ptr, dev.offset(P * Hm * Wm)); rh = mh.match(dev.ptr, dev.offset(P * Hm * Wm))
    assert mn.last_flags is not None and not mn.last_flags.any()          # the deformed-mesh branch runs inside the entry
    assert np.ptp(mn.last_field[2], axis=0).max() > 0.5 and not mn.last_field[[0, 1, 3, 4, 5]].any() and list(mn.last_tiers) == [2]
    np.testing.assert_array_equal(mn.last_field, mh.last_field)
    np.testing.assert_array_equal(mn.last_tiers[2], mh.last_tiers[2])
    for k in ('tx', 'ty', 'conf0', 'valid', 'deformed'):
        np.testing.assert_array_equal(rn[k], rh[k], err_msg=k)
    assert rn['valid'].all() and rn['deformed'].tolist() == [False, False, True, False, False, False]
    gn = StripBatchMatcher.per_pair(rn); gh = StripBatchMatcher.per_pair(rh)
    for p in range(P):
        for k in ('xy0', 'xy1', 'weight'):
            if mn.last_flags[p]:
                np.testing.assert_allclose(gn[p][k], gh[p][k], atol=1e-6, err_msg=f'{k} of pair {p}')
            else:
                np.testing.assert_array_equal(gn[p][k], gh[p][k], err_msg=f'{k} of pair {p}')
        np.testing.assert_allclose(gn[p]['strain'], gh[p]['strain'], rtol=1e-6, atol=1e-10)
    mn.free(); mh.free(); dev.free()
