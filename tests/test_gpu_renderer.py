"""GPU parity of the general-mesh block renderer / matcher (feabas_amd/renderer.py, matcher.bboxes_mesh_renderer_matcher,
csrc/fb_render.hip) against the oracle restatement of MeshRenderer.crop_multiple + xcorr_fft (oracle/pipeline_ref.py:
render_blocks_mesh1, bboxes_mesh_renderer_matcher; matplotlib's LinearTriInterpolator as the reference uses it)."""
import numpy as np
import pytest
from scipy.ndimage import gaussian_filter
from scipy.spatial import Delaunay

from oracle import fem_ref, ncc_ref, pipeline_ref, region_ref

pytestmark = pytest.mark.gpu


def _texture(rng, h, w):
    a = gaussian_filter(rng.standard_normal((h, w)), 1.5)
    b = gaussian_filter(rng.standard_normal((h, w)), 12)
    t = a / a.std() + 0.7 * b / b.std()
    return np.clip(128 + 40 * t / t.std(), 0, 255).astype(np.uint8)


def _meshes(rng, extent=(600, 480), spacing=45.0, warp=3.0, offset=(7.25, -3.5)):
    """an irregular triangulation (jittered grid -> Delaunay), a smooth non-affine MOVING field and a fractional offset;
    the same state in the oracle's RefMesh and the product's Mesh"""
    from feabas_amd import constant as const
    from feabas_amd.mesh import Mesh
    gx, gy = np.meshgrid(np.arange(0, extent[0] + 1, spacing), np.arange(0, extent[1] + 1, spacing))
    v = np.stack((gx.ravel(), gy.ravel()), axis=-1) + rng.uniform(-0.3, 0.3, (gx.size, 2)) * spacing
    tris = Delaunay(v).simplices.astype(np.int32)
    # orientation as the mesher gives it (positive area), degenerate slivers on the hull dropped
    p = v[tris]
    area = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0])
    tris[area < 0] = tris[area < 0][:, ::-1]
    tris = tris[np.abs(area) > 0.05 * spacing * spacing]
    s = v / np.array(extent)
    field = warp * np.stack((np.sin(2.1 * s[:, 1] + 0.3) + 0.5 * s[:, 0] ** 2, np.cos(1.7 * s[:, 0]) - 0.4 * s[:, 0] * s[:, 1]), axis=-1)
    rot = np.array([[np.cos(0.01), np.sin(0.01)], [-np.sin(0.01), np.cos(0.01)]])
    vm = (v - v.mean(0)) @ rot + v.mean(0) + field
    ref = fem_ref.RefMesh(v, tris)
    ref.set_vertices(vm, fem_ref.GEAR_MOVING); ref.set_offset(np.array([offset]), fem_ref.GEAR_MOVING)
    M = Mesh(v, tris)
    M.set_vertices(vm, const.MESH_GEAR_MOVING); M.set_offset(np.array([offset]), const.MESH_GEAR_MOVING)
    return ref, M


def _blocks(rng, n, size, extent, margin):
    x0 = rng.integers(-margin, extent[0] - size[0] + margin, n)
    y0 = rng.integers(-margin, extent[1] - size[1] + margin, n)
    return np.stack((x0, y0, x0 + size[0], y0 + size[1]), axis=-1)


def _compare_stacks(got, gmask, exp, emask, max_bad=2e-3):
    # pixels within 1e-9 of the hull may fall on either side; pixels whose float32 map coordinate sits on a rounding
    # boundary of the 1/32-px quantisation may take the neighbouring phase
    assert (gmask != emask).mean() < 2e-4
    both = gmask & emask
    d = np.abs(got - exp)[both]
    assert (d > 1e-3 * max(1.0, np.abs(exp).max())).mean() < max_bad
    return d


@pytest.mark.parametrize('dtype', ['u8', 'f32'])
@pytest.mark.parametrize('tol', [0.0, 0.08])
def test_render_stack_vs_oracle(fb, dtype, tol):
    from feabas_amd import renderer
    rng = np.random.default_rng(11)
    ref, M = _meshes(rng)
    img = _texture(rng, 620, 760)
    if dtype == 'f32':
        img = ncc_ref.masked_dog_filter(img, 2.0).astype(np.float32)
    origin = (-60, -50)
    bboxes = _blocks(rng, 24, (72, 56), (600, 480), margin=30)
    exp, emask, etier = pipeline_ref.render_blocks_mesh1(ref, img, bboxes, tol, img_origin=origin, return_mask=True)
    r = renderer.MeshRenderer.from_mesh(M, image_loader=renderer.ResidentImage(img, origin), affine_approx_tol=tol)
    d_out, d_mask, shape, tier = r.render_stack_dev(bboxes)
    got = d_out.to_array(shape, np.float32); gmask = d_mask.to_array(shape, np.uint8).astype(bool)
    d_out.free(); d_mask.free(); r.free()
    assert shape == (24, 56, 72)
    np.testing.assert_array_equal(tier, etier)
    if tol > 0:
        assert set(np.unique(tier)) >= {2, 3}                        # the tolerance splits the blocks between the tiers
    assert 0.5 < emask.mean() < 1.0                                  # some blocks stick out of the mesh
    d = _compare_stacks(got, gmask, exp, emask)
    if dtype == 'u8':
        assert d.max() <= 8                                          # a flipped phase moves a sample by a few grey levels at most
        assert np.all(got == np.rint(got))
    assert np.all(got[~gmask] == 0)


@pytest.mark.parametrize('tol', [0.5, 50.0])
def test_render_stack_precise_mask_of_affine_blocks(fb, tol):
    """log_sigma > 0 makes crop_field_affine mask an affine block pixel by pixel when 1 px^2 or more of it lies outside the
    mesh (renderer.py:437-447): block-affine tier at tol 0.5, the global-affine tier at tol 50.  The oracle
    takes area and containment in image space like the reference, the device in MOVING coordinates: the decisions agree,
    the masks agree except for pixels within the affine tolerance of the mesh border."""
    from feabas_amd import renderer
    rng = np.random.default_rng(19)
    ref, M = _meshes(rng, warp=3.0)
    img = _texture(rng, 620, 760)
    origin = (-60, -50)
    bboxes = _blocks(rng, 32, (72, 56), (600, 480), margin=40)
    exp, emask, etier = pipeline_ref.render_blocks_mesh1(ref, img, bboxes, tol, img_origin=origin, return_mask=True, precise_mask=True)
    plain, pmask, ptier = pipeline_ref.render_blocks_mesh1(ref, img, bboxes, tol, img_origin=origin, return_mask=True)
    r = renderer.MeshRenderer.from_mesh(M, image_loader=renderer.ResidentImage(img, origin), affine_approx_tol=tol)
    d_out, d_mask, shape, tier = r.render_stack_dev(bboxes, precise_mask=True)
    got = d_out.to_array(shape, np.float32); gmask = d_mask.to_array(shape, np.uint8).astype(bool)
    d_out.free(); d_mask.free()
    d_out, d_mask, shape, tier0 = r.render_stack_dev(bboxes)          # log_sigma = 0: every pixel of an affine block is valid
    gmask0 = d_mask.to_array(shape, np.uint8).astype(bool)
    d_out.free(); d_mask.free(); r.free()
    np.testing.assert_array_equal(tier, etier)
    np.testing.assert_array_equal(tier0, ptier)
    assert np.sum(tier >= 10) >= 4 and np.sum((tier == 1) | (tier == 2)) >= 4          # blocks that stick out, blocks that do not
    assert set(np.unique(tier % 10)) <= ({1} if tol > 10 else {2, 3}) and (tol > 10 or np.sum(tier % 10 == 2) >= 8)
    aff = tier0 != 3
    assert gmask0[aff].all() and pmask[aff].all()
    out = tier >= 10
    assert 0.02 < 1 - gmask[out].mean() < 0.9                        # the precise mask removes the part outside the mesh
    assert (gmask != emask).mean() < (2e-3 if tol < 10 else 2e-2)    # border pixels within the affine tolerance may differ
    both = gmask & emask
    assert np.abs(got - exp)[both].max() <= 8 and (np.abs(got - exp)[both] > 1).mean() < 2e-3
    assert np.all(got[~gmask] == 0)


def test_crop_multiple_mask_range(fb):
    """crop_multiple(log_sigma, mask_range=(lo, hi)) (renderer.py:632-641): rendered grey levels outside the range leave the
    mask before the masked DoG -- against the oracle's render + range + masked_dog_filter"""
    from feabas_amd import renderer
    rng = np.random.default_rng(21)
    ref, M = _meshes(rng)
    img = _texture(rng, 620, 760)
    origin = (-60, -50)
    bboxes = _blocks(rng, 12, (72, 56), (600, 480), margin=20)
    st, mk, _ = pipeline_ref.render_blocks_mesh1(ref, img, bboxes, 0.0, img_origin=origin, return_mask=True)
    lo, hi = 70.0, 190.0
    inr = (st >= lo) & (st <= hi)
    assert 0.02 < 1 - inr[mk].mean() < 0.5                            # the range really cuts pixels
    exp = ncc_ref.masked_dog_filter(st, 2.5, mask=mk & inr)
    r = renderer.MeshRenderer.from_mesh(M, image_loader=renderer.ResidentImage(img, origin))
    got = r.crop_multiple(bboxes, log_sigma=2.5, mask_range=(lo, hi))
    plain = r.crop_multiple(bboxes, log_sigma=2.5)
    r.free()
    assert np.abs(got - exp).max() <= 3e-3 * np.abs(exp).max() or (np.abs(got - exp) > 3e-3 * np.abs(exp).max()).mean() < 2e-3
    assert np.abs(got - plain).max() > 0.05 * np.abs(plain).max()      # and changes the result


def test_masked_dog_of_a_stack(fb):
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    rng = np.random.default_rng(12)
    st = rng.integers(0, 255, (9, 80, 96)).astype(np.float32)
    mk = np.ones(st.shape, dtype=bool)
    mk[0, :30] = False; mk[3, :, 50:] = False; mk[5, 20:40, 20:70] = False; mk[8] = False
    st[~mk] = 0
    exp = ncc_ref.masked_dog_filter(st, 2.5, mask=mk)
    d_in, d_m, d_o = _lib.DeviceBuffer.from_array(st), _lib.DeviceBuffer.from_array(mk.astype(np.uint8)), _lib.DeviceBuffer(st.nbytes)
    _lib.check(lib.fb_dog_masks_dev(ctx, d_in.ptr, 1, 9, 80, 96, 2.5, d_m.ptr, 1, d_o.ptr))
    got = d_o.to_array(st.shape, np.float32)
    np.testing.assert_allclose(got, exp, atol=2e-3 * np.abs(exp).max())
    full = np.ones_like(mk)                                          # no zero in any mask: plain DoG (common.py:368)
    d_m2 = _lib.DeviceBuffer.from_array(full.astype(np.uint8))
    _lib.check(lib.fb_dog_masks_dev(ctx, d_in.ptr, 1, 9, 80, 96, 2.5, d_m2.ptr, 1, d_o.ptr))
    np.testing.assert_allclose(d_o.to_array(st.shape, np.float32), ncc_ref.masked_dog_filter(st, 2.5), atol=2e-3 * np.abs(exp).max())
    for b in (d_in, d_m, d_m2, d_o):
        b.free()


@pytest.mark.parametrize('sigma,tol', [(0.0, 0.0), (2.5, 0.0), (2.5, 0.08)])
def test_bboxes_mesh_renderer_matcher_vs_oracle(fb, sigma, tol):
    from feabas_amd import matcher, renderer
    rng = np.random.default_rng(13)
    extent = (600, 480)
    ref0, M0 = _meshes(rng, warp=2.0, offset=(0.0, 0.0))
    ref1, M1 = _meshes(rng, warp=3.0, offset=(4.5, 2.25))
    base = _texture(rng, 700, 820)
    img0 = base[40:660, 30:790]                                       # section 1 is section 0 shifted by (9, -6) + its own noise
    img1 = np.clip(base[46:666, 21:781].astype(np.int32) + rng.integers(-6, 7, (620, 760)), 0, 255).astype(np.uint8)
    if sigma == 0:                                                    # images band-passed beforehand, as the stitching side hands them over
        img0 = ncc_ref.masked_dog_filter(img0, 2.5).astype(np.float32); img1 = ncc_ref.masked_dog_filter(img1, 2.5).astype(np.float32)
    org = (-60, -50)
    # blocks of two sizes (two batches), a few sticking out of mesh1
    b_a = _blocks(rng, 14, (96, 80), extent, margin=10)
    b_b = _blocks(rng, 10, (64, 64), extent, margin=10)
    bboxes0 = np.concatenate((b_a, b_b)); bboxes1 = bboxes0 + np.array([3, -2, 3, -2])
    exp = [pipeline_ref.bboxes_mesh_renderer_matcher(ref0, ref1, img0, img1, b0, b1, sigma=sigma, affine_approx_tol=tol,
                                                     img_origin0=org, img_origin1=org)
           for b0, b1 in ((bboxes0[:14], bboxes1[:14]), (bboxes0[14:], bboxes1[14:]))]
    exy0, exy1, econf = (np.concatenate([e[k] for e in exp]) for k in range(3))
    r0 = renderer.ResidentImage(img0, org); r1 = renderer.ResidentImage(img1, org)
    if tol > 0:                                                       # meshes handed over as a Mesh H5 file / an init dict (matcher.py:792-799)
        import os, tempfile
        from feabas_amd import h5wire
        with tempfile.TemporaryDirectory() as tmp:
            M0.save_to_h5(os.path.join(tmp, 'm0.h5'))
            as_dict = {k: v for k, v in h5wire.mesh_init_dict(M1, save_material=False).items() if k not in ('epsilon', 'name')}
            xy0, xy1, conf = matcher.bboxes_mesh_renderer_matcher(os.path.join(tmp, 'm0.h5'), as_dict, r0, r1, bboxes0, bboxes1, sigma=sigma,
                                                                  affine_approx_tol=tol, batch_size=100)
    else:
        # without the DoG no term depends on the batch: the reference's small batches (split path) must give the same result
        xy0, xy1, conf = matcher.bboxes_mesh_renderer_matcher(M0, M1, r0, r1, bboxes0, bboxes1, sigma=sigma, affine_approx_tol=tol,
                                                              batch_size=7 if sigma == 0 else 100, merge_batches=False)
    r0.free(); r1.free()
    assert xy0.shape == (24, 2) and conf.shape == (24,)
    strong = econf > 0.3
    assert strong.sum() >= 16
    np.testing.assert_array_equal(xy0[strong], exy0[strong])         # integer peaks bit-exact
    np.testing.assert_array_equal(xy1[strong], exy1[strong])
    np.testing.assert_allclose(conf[strong], econf[strong], atol=1e-4)
    d = (xy1 - xy0)[strong]
    assert np.abs(np.median(d, axis=0)).max() < 25                   # a real displacement field, not a constant


@pytest.mark.parametrize('H,W,seed,shift,amp', [(768, 256, 5, (6, -4), 0.0), (1024, 256, 1, (4, -3), 3.0), (3600, 72, 21, (2, -3), 2.5)])
def test_iterative_matcher_general_path_vs_strip_oracle(fb, H, W, seed, shift, amp):
    """matcher.iterative_xcorr_matcher_w_mesh (general meshes: device renderer + SLM) fed the way stitching_matcher feeds it
    (matcher.py:338-363) against the oracle's statement-by-statement pair pipeline (pipeline_ref.match_pair, pinned by the
    golden vectors): the same matches, weights and strain must come out of the general code path -- rigid pair, a pair whose
    mesh1 deforms between the spacings, and three spacings (deformed twice)."""
    from test_gpu_pipeline import _warped_pair
    from feabas_amd import matcher
    from feabas_amd import constant as const
    from feabas_amd.mesh import Mesh
    img0, img1 = _warped_pair(H, W, seed, shift, amp)
    exp = pipeline_ref.match_pair(img0, img1, residue_len=2.0)
    assert exp['xy0'] is not None
    f0 = ncc_ref.masked_dog_filter(img0, 2.5).astype(np.float32); f1 = ncc_ref.masked_dog_filter(img1, 2.5).astype(np.float32)
    spacings = ncc_ref.auto_spacings((H, W), (H, W))
    m0 = Mesh.from_bbox((0, 0, W, H), cartesian=True, mesh_size=float(np.min(spacings)), min_num_blocks=2, uid=0)
    m1 = Mesh.from_bbox((0, 0, W, H), cartesian=True, mesh_size=float(np.min(spacings)), min_num_blocks=2, uid=1)
    m0.apply_translation((exp['tx'], exp['ty']), const.MESH_GEAR_FIXED)
    m0.lock()
    xy0, xy1, weight, strain = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, f0, f1, spacings=spacings, distributor='cartesian_bbox',
                                                                      residue_len=2.0, conf_thresh=0.33, min_num_blocks=2)
    assert xy0 is not None and xy0.shape == exp['xy0'].shape
    np.testing.assert_allclose(xy0, exp['xy0'], atol=1e-4)
    np.testing.assert_allclose(xy1, exp['xy1'], atol=1e-4)
    np.testing.assert_allclose(weight, exp['weight'], atol=1e-4)
    assert abs(strain - exp['strain']) < 1e-4 * max(1e-4, exp['strain'])
    if H == 3600:
        assert exp['deformed'] and len(exp['tiers']) == 2            # the deformed-mesh branch of the oracle, twice


def test_block_uncovered_device_equals_host(fb):
    """fb_mesh_block_uncovered_dev (one thread per block, candidate lists resident) against the host loop
    fb_mesh_block_uncovered on the lists fb_mesh_candidates_dev made: the same areas to rounding, the same precise-mask
    decision (uncovered >= 1 px^2)"""
    from scipy.spatial import Delaunay
    from feabas_amd import _lib
    lib, ctx = _lib.load(), _lib.ctx()
    rng = np.random.default_rng(14)
    v = np.ascontiguousarray(rng.uniform(0, 900, (700, 2)))
    tri = np.ascontiguousarray(Delaunay(v).simplices, dtype=np.int32)
    NB, h, w, cap = 1500, 40, 56, 64
    org = np.ascontiguousarray(rng.uniform(-60, 900, (NB, 2)))
    d_v, d_t, d_o = _lib.DeviceBuffer.from_array(v), _lib.DeviceBuffer.from_array(tri), _lib.DeviceBuffer.from_array(org)
    d_c, d_n, d_u = _lib.DeviceBuffer(4 * NB * cap), _lib.DeviceBuffer(4 * NB), _lib.DeviceBuffer(8 * NB)
    try:
        _lib.check(lib.fb_mesh_candidates_dev(ctx, tri.shape[0], d_v.ptr, d_t.ptr, NB, d_o.ptr, h, w, cap, d_c.ptr, d_n.ptr))
        cnt = d_n.to_array((NB,), np.int32); cand = d_c.to_array((NB, cap), np.int32)
        assert cnt.max() <= cap and cnt.min() == 0 and cnt.max() > 8
        _lib.check(lib.fb_mesh_block_uncovered_dev(ctx, d_v.ptr, d_t.ptr, NB, d_o.ptr, h, w, cap, d_c.ptr, d_n.ptr, d_u.ptr))
        got = d_u.to_array((NB,), np.float64)
    finally:
        for b in (d_v, d_t, d_o, d_c, d_n, d_u):
            b.free()
    exp = np.empty(NB)
    _lib.check(lib.fb_mesh_block_uncovered(ctx, v.shape[0], _lib.ptr(v), _lib.ptr(tri), NB, _lib.ptr(org), h, w, cap, _lib.ptr(cand), _lib.ptr(cnt), _lib.ptr(exp)))
    np.testing.assert_allclose(got, exp, atol=1e-8)
    assert (exp > 1).any() and (exp < 1e-6).any()               # blocks sticking out of the hull and blocks inside it
    np.testing.assert_array_equal(got >= 1.0, exp >= 1.0)


def test_area_resize_vs_oracle(fb):
    """fb_area_resize (cv2.resize INTER_AREA restated: integer cells and fractional-coverage taps) against the oracle's numpy
    statement, bit for bit; x0.5 equals the older fb_area_downsample2"""
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (3, 203, 157), dtype=np.uint8)
    img[1] = np.clip(np.round(128 + 60 * np.sin(np.arange(157)[None, :] / 7.0) * np.cos(np.arange(203)[:, None] / 11.0)), 0, 255).astype(np.uint8)
    for fx, fy in ((0.5, 0.5), (0.25, 0.25), (1 / 3, 1 / 3), (0.25, 0.5), (0.3, 0.3), (0.7, 0.45), (0.9, 1.0), (0.125, 0.125)):
        got = fb.common.area_resize(img, fx, fy)
        for n in range(3):
            exp = ncc_ref.area_resize(img[n], fx, fy)
            assert got[n].shape == exp.shape, (fx, fy)
            np.testing.assert_array_equal(got[n], exp, err_msg=str((fx, fy)))
    np.testing.assert_array_equal(fb.common.area_resize(img, 0.5), fb.common.area_downsample2(img))
    mk = rng.random((203, 157)) > 0.3
    for f in (0.5, 0.25, 0.3):
        np.testing.assert_array_equal(fb.common.nearest_resize_mask(mk, f), ncc_ref.nearest_resize_mask(mk, f))
    with pytest.raises(NotImplementedError):
        fb.common.area_resize(img, 1.5)


@pytest.mark.parametrize('cd,fd', [(0.25, 0.5), (0.5, 0.5), (0.25, 1), (0.4, 0.8)])
def test_stitching_matcher_downsample_factors_vs_oracle(fb, cd, fd):
    """stitching_matcher(coarse_downsample, fine_downsample) away from the defaults (matcher.py:254-266, 318-341, 352, 365-367):
    coarse level through the area resize, fine images shrunk, spacings / residue_len / global translation scaled, matches
    scaled back -- against pipeline_ref.match_pair with the same factors"""
    from test_gpu_pipeline import _warped_pair
    from feabas_amd import matcher
    img0, img1 = _warped_pair(1400, 360, 31, shift=(9, -6), warp=0.6)
    exp = pipeline_ref.match_pair(img0, img1, residue_len=2.0, coarse_downsample=cd, fine_downsample=fd)
    assert exp['xy0'] is not None and exp['xy0'].shape[0] > 20
    xy0, xy1, weight, strain, _ = matcher.stitching_matcher(img0, img1, sigma=2.5, coarse_downsample=cd, fine_downsample=fd, conf_thresh=0.33, residue_len=2.0)
    assert xy0 is not None and xy0.shape == exp['xy0'].shape
    np.testing.assert_allclose(xy0, exp['xy0'], atol=1e-4 / fd)
    np.testing.assert_allclose(xy1, exp['xy1'], atol=1e-4 / fd)
    np.testing.assert_allclose(weight, exp['weight'], atol=1e-4)
    assert abs(strain - exp['strain']) < 1e-4 * max(1e-4, exp['strain'])
    # the matches sit where the strips were shifted (strip pixels, not pixels of the shrunk images)
    d = np.median(xy1 - xy0, axis=0)
    assert abs(d[0] + 9) < 1.0 and abs(d[1] - 6) < 1.0
    # the batch entry takes the same route for such options
    got = matcher.stitching_matcher_batch([(img0, img1)], sigma=2.5, coarse_downsample=cd, fine_downsample=fd, conf_thresh=0.33, residue_len=2.0)
    np.testing.assert_array_equal(got[0][0], xy0)
    if (cd, fd) == (0.25, 0.5):
        # masks shrink by the INTER_NEAREST rule with their images (matcher.py:257-264, 322-329); photometric statistics of the coarse level
        mk0 = np.ones(img0.shape, dtype=bool); mk0[300:420, 40:200] = False
        mk1 = np.ones(img1.shape, dtype=bool); mk1[900:1000, 150:330] = False
        exp = pipeline_ref.match_pair(img0, img1, residue_len=2.0, coarse_downsample=cd, fine_downsample=fd, mask0=mk0, mask1=mk1, compute_photometric=True)
        xy0, xy1, weight, strain, phtm = matcher.stitching_matcher(img0, img1, sigma=2.5, coarse_downsample=cd, fine_downsample=fd, conf_thresh=0.33, residue_len=2.0,
                                                                    mask0=mk0, mask1=mk1, compute_photometric=True)
        assert xy0.shape == exp['xy0'].shape
        np.testing.assert_allclose(xy0, exp['xy0'], atol=1e-4 / fd)
        np.testing.assert_allclose(xy1, exp['xy1'], atol=1e-4 / fd)
        np.testing.assert_allclose(weight, exp['weight'], atol=1e-4)
        np.testing.assert_allclose(phtm, exp['phtm'], rtol=1e-4)


def test_section_matcher_recovers_a_known_field(fb):
    """alignment-scale property test (no oracle at this size): section 1 is section 0 resampled through a known smooth
    field of +-6 px; section_matcher over two irregular meshes (spacings 280 / 70 px, sigma 2.5, residue filter) must
    return matches whose displacement is that field -- through two rounds of render / NCC / relaxation on the device."""
    from scipy.ndimage import map_coordinates
    from feabas_amd import matcher
    rng = np.random.default_rng(41)
    S = 1800
    base = _texture(rng, S, S)
    yy, xx = np.meshgrid(np.arange(S, dtype=np.float64), np.arange(S, dtype=np.float64), indexing='ij')

    def field(x, y):
        return (6.0 * np.sin(2 * np.pi * y / 1500.0 + 0.4) + 2.5 * (x / S) ** 2,
                5.0 * np.cos(2 * np.pi * x / 1300.0) - 2.0 * (x / S) * (y / S))
    ux, uy = field(xx, yy)
    img1 = np.clip(np.rint(map_coordinates(base.astype(np.float32), [yy + uy, xx + ux], order=1, mode='nearest')), 0, 255).astype(np.uint8)
    _, M0 = _meshes(rng, extent=(S - 1, S - 1), spacing=75.0, warp=0.0, offset=(0.0, 0.0))
    _, M1 = _meshes(rng, extent=(S - 1, S - 1), spacing=75.0, warp=0.0, offset=(0.0, 0.0))
    M0.uid, M1.uid = 0.0, 1.0
    for M in (M0, M1):                                              # no prior alignment: MOVING = INITIAL
        M._vertices[1] = None
    trace = []
    xy0, xy1, weight, strain = matcher.section_matcher(M0, M1, base, img1, spacings=[280, 70], conf_thresh=0.3, residue_len=3.0,
                                                       compute_strain=True, trace=trace)
    assert xy0 is not None and len(trace) == 2
    assert trace[0]['blocks'] >= 25 and trace[1]['blocks'] >= 500
    assert xy0.shape[0] > 0.8 * trace[1]['blocks']
    ex, ey = field(xy1[:, 0], xy1[:, 1])
    err = np.hypot(xy1[:, 0] - xy0[:, 0] + ex, xy1[:, 1] - xy0[:, 1] + ey)      # q + u(q) = p
    assert np.median(err) < 0.25 and np.quantile(err, 0.95) < 0.8
    assert trace[1]['max_dis'] < trace[0]['max_dis']               # the first relaxation took most of the field out
    assert 0.0 < strain < 0.05
    assert np.all(weight > 0.3 * 0) and weight.shape[0] == xy0.shape[0]


def test_section_matcher_defaults_on_islands_with_a_hole(fb):
    """the reference's alignment defaults end to end (matcher.py:370-427, default_alignment_configs.yaml:13-27): region-aware
    block lattice (`cartesian_region`, min_boundary_distance), residue_len in units of the section thickness (< 0), soft
    triangles dropped by `stiffness_multiplier_threshold`, and -- with initial matches -- the meshes cut into their connected
    parts, each pair of parts matched on its own.  Two islands (one with a hole); section 1 = section 0 through a known field;
    the images come from a loader with the reference's crop() interface."""
    from scipy.ndimage import map_coordinates
    from feabas_amd import matcher, constant as const
    from feabas_amd.mesh import Mesh
    rng = np.random.default_rng(43)
    SH, SW = 1000, 1900
    base = _texture(rng, SH, SW)
    yy, xx = np.meshgrid(np.arange(SH, dtype=np.float64), np.arange(SW, dtype=np.float64), indexing='ij')

    def field(x, y):
        return (4.0 * np.sin(2 * np.pi * y / 900.0 + 0.4) + 1.5 * (x / SW) ** 2, 3.5 * np.cos(2 * np.pi * x / 1100.0) - 1.0 * (x / SW) * (y / SH))
    ux, uy = field(xx, yy)
    img1 = np.clip(np.rint(map_coordinates(base.astype(np.float32), [yy + uy, xx + ux], order=1, mode='nearest')), 0, 255).astype(np.uint8)

    class Loader:                                                    # dal.StreamLoader's interface (dal.py:1045-1050)
        def __init__(self, img):
            self.img, self.calls = img, 0

        def crop(self, bbox, return_empty=False, **kwargs):
            self.calls += 1
            x0, y0, x1, y1 = (int(v) for v in bbox)
            out = np.zeros((y1 - y0, x1 - x0), dtype=self.img.dtype)
            ya, yb, xa, xb = max(y0, 0), min(y1, self.img.shape[0]), max(x0, 0), min(x1, self.img.shape[1])
            if ya < yb and xa < xb:
                out[ya - y0:yb - y0, xa - x0:xb - x0] = self.img[ya:yb, xa:xb]
            return out

    def islands(seed, uid):
        r = np.random.default_rng(seed)
        vs, ts, soft = [], [], []
        nv = 0
        for (x0, y0, w, h, hole) in ((40, 40, 900, 900, (330, 330, 600, 600)), (1060, 120, 780, 760, None)):
            gx, gy = np.meshgrid(np.arange(x0, x0 + w + 1, 60.0), np.arange(y0, y0 + h + 1, 60.0))
            v = np.stack((gx.ravel(), gy.ravel()), axis=-1)
            inner = (gx.ravel() > x0) & (gx.ravel() < x0 + w) & (gy.ravel() > y0) & (gy.ravel() < y0 + h)
            v[inner] += r.uniform(-0.25, 0.25, (int(inner.sum()), 2)) * 60.0
            t = Delaunay(v).simplices.astype(np.int32)
            p = v[t]
            area = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0])
            t[area < 0] = t[area < 0][:, ::-1]
            c = p.mean(axis=1)
            in_hole = np.zeros(t.shape[0], dtype=bool) if hole is None else ((c[:, 0] > hole[0]) & (c[:, 0] < hole[2]) & (c[:, 1] > hole[1]) & (c[:, 1] < hole[3]))
            vs.append(v); ts.append(t + nv); soft.append(in_hole); nv += v.shape[0]
        t = np.concatenate(ts); soft = np.concatenate(soft)
        # the hole is a soft material (stiffness multiplier 0.01): section_matcher drops its triangles
        return Mesh(np.concatenate(vs), t, uid=uid, tri_model=np.zeros(t.shape[0], np.int32), tri_matmult=np.where(soft, 0.01, 1.0).astype(np.float32))
    M0, M1 = islands(1, 0.0), islands(2, 1.0)
    assert M0.connected_triangles()[0] == 2
    # initial matches: a sparse set of true correspondences (what the thumbnail stage delivers)
    q = np.stack((rng.uniform(80, 1800, 400), rng.uniform(80, 900, 400)), axis=-1)
    ex, ey = field(q[:, 0], q[:, 1])
    ini = (q + np.stack((ex, ey), axis=-1) + rng.normal(0, 0.5, q.shape), q, np.ones(400, dtype=np.float32))
    ld0, ld1 = Loader(base), Loader(img1)
    xy0, xy1, weight, strain = matcher.section_matcher(M0, M1, ld0, ld1, spacings=[200, 70], conf_thresh=0.3, residue_len=-1, section_thickness=12.0,
                                                       min_boundary_distance=20, initial_matches=ini, compute_strain=True, num_workers=4)
    assert ld0.calls == 1 and ld1.calls == 1                          # one read per section, not one per block
    assert xy0 is not None and xy0.shape[0] > 150
    ex, ey = field(xy1[:, 0], xy1[:, 1])
    err = np.hypot(xy1[:, 0] - xy0[:, 0] + ex, xy1[:, 1] - xy0[:, 1] + ey)
    assert np.median(err) < 0.3 and np.quantile(err, 0.9) < 1.0
    # matches of both islands, none in the hole, none in the gap between the islands, all >= 20 px inside the outline
    assert np.any(xy1[:, 0] < 950) and np.any(xy1[:, 0] > 1050)
    assert not np.any((xy1[:, 0] > 345) & (xy1[:, 0] < 585) & (xy1[:, 1] > 345) & (xy1[:, 1] < 585))
    assert not np.any((xy1[:, 0] > 945) & (xy1[:, 0] < 1055))
    with pytest.raises(TypeError):
        matcher.section_matcher(M0, M1, ld0, ld1, spacings=[200], bogus=1)


def test_device_point_location_vs_matplotlib(fb):
    """Mesh.tri_finder on the device (fb_mesh_locate_dev) against matplotlib's trapezoid-map finder, which the reference uses
    (mesh.py:2080-2188): same triangles for random points inside and outside an irregular deformed mesh"""
    from matplotlib.tri import Triangulation
    from feabas_amd import constant as const
    rng = np.random.default_rng(51)
    _, M = _meshes(rng, extent=(900, 700), spacing=30.0, warp=4.0, offset=(12.5, -7.25))
    pts = np.stack((rng.uniform(-40, 960, 20000), rng.uniform(-40, 760, 20000)), axis=-1)
    tid = M.tri_finder(pts, gear=const.MESH_GEAR_MOVING)
    v = M.vertices(const.MESH_GEAR_MOVING)
    p = pts - M.offset(const.MESH_GEAR_MOVING)
    exp = np.asarray(Triangulation(v[:, 0], v[:, 1], M.triangles).get_trifinder()(p[:, 0], p[:, 1]))
    assert (tid >= 0).mean() > 0.7 and (tid < 0).sum() > 500
    differ = np.flatnonzero(tid != exp)
    assert differ.size <= 5                                         # points within 1e-9 of an edge may go to either side
    t2, B = M.cart2bary(pts[tid >= 0], const.MESH_GEAR_MOVING, tid=tid[tid >= 0])
    assert B.min() > -1e-8
    np.testing.assert_allclose(M.bary2cart(t2, B, const.MESH_GEAR_MOVING), pts[tid >= 0], atol=1e-8)
    assert M.tri_finder(np.empty((0, 2)), gear=const.MESH_GEAR_MOVING).shape == (0,)


def test_point_location_culled_kernel_equals_the_plain_walk(fb):
    """fb_mesh_locate_dev takes large point sets through the workgroup-culled kernel (triangles tested against the box of a
    workgroup's 256 points first): a raster, random points and points exactly ON vertices / edge midpoints (several triangles
    contain them: the smallest index wins) get the same triangle as the plain walk, which lists of fewer than 2 048 points take"""
    from feabas_amd import constant as const
    rng = np.random.default_rng(52)
    _, M = _meshes(rng, extent=(1500, 1100), spacing=25.0, warp=5.0, offset=(-3.0, 8.5))
    g = const.MESH_GEAR_MOVING
    v = M.vertices_w_offset(g)
    xs, ys = np.arange(-20, 1530, 7.5), np.arange(-20, 1130, 7.5)
    raster = np.stack(np.meshgrid(xs, ys), -1).reshape(-1, 2)
    t = M.triangles
    mid = 0.5 * (v[t[:, 0]] + v[t[:, 1]])
    pts = np.concatenate((raster, v, mid, np.stack((rng.uniform(-30, 1530, 9000), rng.uniform(-30, 1130, 9000)), -1)))
    assert pts.shape[0] > 2 * 16384
    got = M.tri_finder(pts, gear=g)
    exp = np.concatenate([M.tri_finder(pts[a:a + 1500], gear=g) for a in range(0, pts.shape[0], 1500)])
    np.testing.assert_array_equal(got, exp)
    assert (got >= 0).mean() > 0.8 and (got < 0).sum() > 100


# ------------------------------------------------------------------ a8: the region-aware distributor and section_matcher against their oracle
def _island_pair(rng):
    """two sections of two islands each (one with a hole), jittered Delaunay meshes, different in the two sections; materials:
    a 'wrinkle' band (area_constraint 0.25) in section 0, a zone named 'refine_a' (area_constraint 0.5) in section 1"""
    from feabas_amd.mesh import Mesh

    def islands(r, uid, shift):
        vs, ts, nv = [], [], 0
        for (x0, y0, w, h, hole) in ((30, 30, 520, 520, (200, 210, 330, 340)), (640, 90, 380, 420, None)):
            gx, gy = np.meshgrid(np.arange(x0, x0 + w + 1, 40.0), np.arange(y0, y0 + h + 1, 40.0))
            v = np.stack((gx.ravel(), gy.ravel()), axis=-1) + shift
            inner = (gx.ravel() > x0) & (gx.ravel() < x0 + w) & (gy.ravel() > y0) & (gy.ravel() < y0 + h)
            v[inner] += r.uniform(-0.25, 0.25, (int(inner.sum()), 2)) * 40.0
            t = Delaunay(v).simplices.astype(np.int32)
            p = v[t]
            area = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0])
            t[area < 0] = t[area < 0][:, ::-1]
            c = p.mean(axis=1)
            if hole is not None:
                t = t[~((c[:, 0] > hole[0]) & (c[:, 0] < hole[2]) & (c[:, 1] > hole[1]) & (c[:, 1] < hole[3]))]
            vs.append(v); ts.append(t + nv); nv += v.shape[0]
        v, t = np.concatenate(vs), np.concatenate(ts)
        used = np.unique(t)
        remap = np.full(v.shape[0], -1); remap[used] = np.arange(used.size)
        return v[used], remap[t].astype(np.int32)
    v0, t0 = islands(np.random.default_rng(rng.integers(1 << 30)), 0, np.array([0.0, 0.0]))
    v1, t1 = islands(np.random.default_rng(rng.integers(1 << 30)), 1, np.array([6.0, -4.0]))
    c0, c1 = v0[t0].mean(axis=1), v1[t1].mean(axis=1)
    ids0 = np.where((c0[:, 1] > 380) & (c0[:, 0] < 560), 4, 0).astype(np.int32)
    ids1 = np.where(c1[:, 0] > 860, 7, 0).astype(np.int32)
    names0, cons0 = {'default': 0, 'wrinkle': 4}, {'default': 1.0, 'wrinkle': 0.25}
    names1, cons1 = {'default': 0, 'refine_a': 7}, {'default': 1.0, 'refine_a': 0.5}
    M0 = Mesh(v0, t0, uid=0.0, material_ids=ids0, material_names=names0, material_area_constraints=cons0)
    M1 = Mesh(v1, t1, uid=1.0, material_ids=ids1, material_names=names1, material_area_constraints=cons1)
    mats = ((ids0, {n: (u, cons0[n]) for n, u in names0.items()}), (ids1, {n: (u, cons1[n]) for n, u in names1.items()}))
    return (v0, t0, v1, t1), (M0, M1), mats


@pytest.mark.parametrize('case', ['plain', 'boundary', 'refine_both', 'refine_only', 'shrink_pair'])
def test_distribute_matching_blocks_vs_oracle(fb, case):
    """SURVEY row a8, distributor: matcher.distribute_matching_blocks ('cartesian_region') against oracle/region_ref.py on two
    sections of two islands (one with a hole): the blocks of the two agree EXACTLY -- set, sizes and z-order -- once the oracle
    takes the product's lattice phase part by part (the anchor is GEOS's representative point, which neither side can compute
    without GEOS: INTEGRATION.md), and the product's own anchors sit within its raster step of the oracle's"""
    from feabas_amd import matcher, constant as const
    (v0, t0, v1, t1), (M0, M1), mats = _island_pair(np.random.default_rng(11))
    kw = dict(plain=dict(refine_mode=0), boundary=dict(refine_mode=0, min_boundary_distance=25),
              refine_both=dict(refine_mode=2, min_boundary_distance=15), refine_only=dict(refine_mode=1),
              shrink_pair=dict(refine_mode=0, shrink_factor=(1, 0.6), min_boundary_distance=10))[case]
    sp = 110.0
    g0, g1 = matcher.distribute_matching_blocks(M0, M1, sp, gear=const.MESH_GEAR_INITIAL, **kw)
    assert g0.shape[0] > 20
    # (1) on the raster the product documents for itself (a quarter of a level's lattice step: connected parts and areas), with
    #     the product's lattice phase: identical
    e0, e1 = region_ref.distribute_matching_blocks(v0, t0, v1, t1, sp, materials=mats, anchor_blocks=g0, res=None, **kw)
    np.testing.assert_array_equal(g0, e0)
    np.testing.assert_array_equal(g1, e1)
    if 'refine' in case:
        sides = np.unique(g0[:, 2] - g0[:, 0])
        assert sides.size >= 2 + (case == 'refine_both')                      # levels 0.25, 0.5 (and 1.0): different block sizes
    # (2) on a raster of 1 px (parts as shapely's polygons have them, but for spurs thinner than a pixel): the same blocks except
    #     where a spur of the region is thinner than the product's raster and hangs on another part than the nearest one
    #     (one block of 135 in the 'refine_only' case)
    h0, _ = region_ref.distribute_matching_blocks(v0, t0, v1, t1, sp, materials=mats, anchor_blocks=g0, res=1.0, **kw)
    gs, hs = set(map(tuple, g0)), set(map(tuple, h0))
    assert len(gs ^ hs) <= 0.01 * len(gs) + (1 if case == 'refine_only' else 0), (len(gs), len(gs ^ hs))
    # (3) the product's own lattice phase against the oracle's (no anchors handed over): the same number of blocks up to the
    #     lattice phase (anchors agree to the product's raster step: the documented difference to shapely)
    f0, _ = region_ref.distribute_matching_blocks(v0, t0, v1, t1, sp, materials=mats, res=1.0, **kw)
    assert abs(f0.shape[0] - g0.shape[0]) <= 0.25 * g0.shape[0]
    for side in np.unique(g0[:, 2] - g0[:, 0]):
        a = g0[g0[:, 2] - g0[:, 0] == side]; b = f0[f0[:, 2] - f0[:, 0] == side]
        assert a.shape[0] and b.shape[0]


def test_section_matcher_vs_oracle(fb):
    """SURVEY row a8 end to end: matcher.section_matcher (cartesian_region lattice with a boundary distance, two spacings, DoG,
    block NCC through both meshes, relaxation with huber residues, strain) against oracle/region_ref.section_match -- the
    reference's loop (matcher.py:370-427, 430-778) with every relaxation solved exactly.  Section 0 is LOCKED: the pair then has
    no soft modes and a PCG at 1e-9 sits on the fixed point (a floating pair has a soft common rotation: at 1e-9 its field is
    2.7e-3 off the fixed point -- measured, round 5 -- which moves the sample grid of the next round by hundredths of a pixel).
    Round by round: the same blocks (the oracle takes the lattice phase from the product's blocks, see the distributor test),
    confidences to 1e-4, mesh field after the relaxation, then matches, weights and strain to 1e-4."""
    from scipy.ndimage import map_coordinates
    from feabas_amd import matcher
    rng = np.random.default_rng(17)
    (v0, t0, v1, t1), (M0, M1), _ = _island_pair(rng)
    SH, SW = 600, 1080
    base = _texture(rng, SH, SW)
    yy, xx = np.meshgrid(np.arange(SH, dtype=np.float64), np.arange(SW, dtype=np.float64), indexing='ij')
    ux = 3.0 * np.sin(2 * np.pi * yy / 700.0 + 0.4) + 1.0 * (xx / SW) ** 2
    uy = 2.5 * np.cos(2 * np.pi * xx / 900.0) - 1.0 * (xx / SW) * (yy / SH)
    img1 = np.clip(np.rint(map_coordinates(base.astype(np.float32), [yy + uy, xx + ux], order=1, mode='nearest')), 0, 255).astype(np.uint8)
    for M in (M0, M1):
        M.material_ids = None; M.material_names = {}; M.material_area_constraints = {}
    M0.locked = True
    kw = dict(spacings=[150, 60], sigma=2.5, conf_thresh=0.3, residue_len=3.0, min_boundary_distance=12, stiffness_lambda=0.5)
    trace = []
    xy0, xy1, wt, strain = matcher.section_matcher(M0, M1, base, img1, compute_strain=True, relax_tol=1e-9, merge_batches=False, batch_size=100,
                                                   stiffness_multiplier_threshold=0, trace=trace, **kw)
    assert xy0 is not None and len(trace) == 2 and trace[1]['blocks'] > 80
    r0 = fem_ref.RefMesh(v0, t0, uid=0, locked=True); r1 = fem_ref.RefMesh(v1, t1, uid=1)
    otrace = []
    ex0, ex1, ewt, estrain = region_ref.section_match(r0, r1, base, img1, compute_strain=True, batch_size=100, anchor_rounds=[t['bboxes0'] for t in trace],
                                                      trace=otrace, **kw)
    assert len(otrace) == 2
    for g, e in zip(trace, otrace):
        np.testing.assert_allclose(g['bboxes0'], e['bboxes0'], atol=1e-6)
        np.testing.assert_allclose(g['bboxes1'], e['bboxes1'], atol=1e-6)
        assert g['pad'] == e['pad']
        np.testing.assert_allclose(g['conf'], e['conf'], atol=1e-4)
        if 'field1' in e:
            assert np.abs(g['field1'] - e['field1']).max() < 1e-4 * max(1.0, np.abs(e['field1']).max())
    assert xy0.shape == ex0.shape and xy0.shape[0] > 60
    np.testing.assert_allclose(xy0, ex0, atol=1e-4); np.testing.assert_allclose(xy1, ex1, atol=1e-4)
    np.testing.assert_allclose(wt, ewt, atol=1e-4)
    assert abs(strain - estrain) < 1e-4 * max(1e-4, estrain) + 1e-7
    # and the matches carry the imposed field (q + u(q) = p)
    ex = 3.0 * np.sin(2 * np.pi * xy1[:, 1] / 700.0 + 0.4) + (xy1[:, 0] / SW) ** 2
    ey = 2.5 * np.cos(2 * np.pi * xy1[:, 0] / 900.0) - (xy1[:, 0] / SW) * (xy1[:, 1] / SH)
    assert np.median(np.hypot(xy1[:, 0] - xy0[:, 0] + ex, xy1[:, 1] - xy0[:, 1] + ey)) < 0.3


@pytest.mark.timeout(240)
def test_section_matcher_floating_pair_vs_oracle(fb):
    """a8 as the aligner calls it (aligner.py:47-142: BOTH sections free).  The pair floats: every relaxation has the common
    translations of its link-connected parts in its null space (deflated by the device PCG, csrc/fb_solver.hip) and a soft common
    rotation whose stiffness sits ~50 x above the float32 noise of the reference's arithmetic -- the relaxed fields of two
    implementations agree to about a percent of that rotation, not to 1e-4, and the next round's sample grid moves with them.
    Compared therefore: round 1 exactly (blocks, pads, confidences: nothing relaxed yet); the relaxed field of round 1 up to a
    rigid motion of the pair (what the links do not determine) at 1e-3 px; the final matches through the field they sample
    (the oracle's match displacements interpolated at the product's match positions); both carry the imposed field; strain"""
    from scipy.interpolate import LinearNDInterpolator
    from scipy.ndimage import map_coordinates
    from feabas_amd import matcher
    rng = np.random.default_rng(17)
    (v0, t0, v1, t1), (M0, M1), _ = _island_pair(rng)
    SH, SW = 600, 1080
    base = _texture(rng, SH, SW)
    yy, xx = np.meshgrid(np.arange(SH, dtype=np.float64), np.arange(SW, dtype=np.float64), indexing='ij')
    ux = 3.0 * np.sin(2 * np.pi * yy / 700.0 + 0.4) + 1.0 * (xx / SW) ** 2
    uy = 2.5 * np.cos(2 * np.pi * xx / 900.0) - 1.0 * (xx / SW) * (yy / SH)
    img1 = np.clip(np.rint(map_coordinates(base.astype(np.float32), [yy + uy, xx + ux], order=1, mode='nearest')), 0, 255).astype(np.uint8)
    for M in (M0, M1):
        M.material_ids = None; M.material_names = {}; M.material_area_constraints = {}
    kw = dict(spacings=[150, 60], sigma=2.5, conf_thresh=0.3, residue_len=3.0, min_boundary_distance=12, stiffness_lambda=0.5)
    trace = []
    xy0, xy1, wt, strain = matcher.section_matcher(M0, M1, base, img1, compute_strain=True, relax_tol=1e-11, merge_batches=False, batch_size=100,
                                                   stiffness_multiplier_threshold=0, trace=trace, **kw)
    assert xy0 is not None and len(trace) == 2 and trace[1]['blocks'] > 80
    assert np.all(np.isfinite(xy0)) and np.all(np.isfinite(xy1))
    r0 = fem_ref.RefMesh(v0, t0, uid=0); r1 = fem_ref.RefMesh(v1, t1, uid=1)
    otrace = []
    ex0, ex1, ewt, estrain = region_ref.section_match(r0, r1, base, img1, compute_strain=True, batch_size=100, anchor_rounds=[trace[0]['bboxes0']],
                                                      trace=otrace, **kw)
    assert len(otrace) == 2
    g, e = trace[0], otrace[0]
    np.testing.assert_allclose(g['bboxes0'], e['bboxes0'], atol=1e-6); np.testing.assert_allclose(g['bboxes1'], e['bboxes1'], atol=1e-6)
    assert g['pad'] == e['pad']
    np.testing.assert_allclose(g['conf'], e['conf'], atol=1e-4)
    if 'field1' in e and 'field1' in g:
        # the relaxed fields of mesh 1 up to what the links leave open: per island, a rigid motion (translation + small rotation)
        # fitted to the difference
        from scipy.sparse import coo_matrix, csgraph
        nv = v1.shape[0]
        adj = coo_matrix((np.ones(3 * t1.shape[0]), (t1.ravel(), np.roll(t1, 1, axis=1).ravel())), shape=(nv, nv))
        ncomp, lab = csgraph.connected_components(adj, directed=False)
        assert ncomp == 2
        d = g['field1'] - e['field1']
        for k in range(ncomp):
            sel = lab == k
            c = v1[sel] - v1[sel].mean(axis=0)
            Amat = np.zeros((2 * c.shape[0], 3)); Amat[0::2, 0] = 1; Amat[1::2, 1] = 1; Amat[0::2, 2] = -c[:, 1]; Amat[1::2, 2] = c[:, 0]
            coef, *_ = np.linalg.lstsq(Amat, d[sel].ravel(), rcond=None)
            rest = d[sel].ravel() - Amat @ coef
            assert np.abs(rest).max() < 0.02 and np.abs(coef[:2]).max() < 0.5, (k, np.abs(rest).max(), coef)
    assert abs(xy0.shape[0] - ex0.shape[0]) <= 0.1 * ex0.shape[0] and xy0.shape[0] > 60
    # through the sampled field: the oracle's match displacement (p - q, a function of q) interpolated at the product's q
    de = LinearNDInterpolator(ex1, ex0 - ex1)
    dd = de(xy1)
    ok = np.all(np.isfinite(dd), axis=1)
    assert ok.mean() > 0.6                                   # (the rest lies outside the hull of the oracle's sample points: islands, a hole)
    diff = np.hypot(*((xy0 - xy1)[ok] - dd[ok]).T)
    assert np.median(diff) < 0.05 and np.quantile(diff, 0.95) < 0.2, (np.median(diff), np.quantile(diff, 0.95))
    assert abs(strain - estrain) < 0.05 * max(1e-4, estrain) + 1e-6
    ex = 3.0 * np.sin(2 * np.pi * xy1[:, 1] / 700.0 + 0.4) + (xy1[:, 0] / SW) ** 2
    ey = 2.5 * np.cos(2 * np.pi * xy1[:, 0] / 900.0) - (xy1[:, 0] / SW) * (xy1[:, 1] / SH)
    assert np.median(np.hypot(xy1[:, 0] - xy0[:, 0] + ex, xy1[:, 1] - xy0[:, 1] + ey)) < 0.3


def test_section_matcher_batch_equals_the_per_pair_calls(fb):
    """matcher.section_matcher_batch (pairs dealt to host threads with a context each, like the aligner deals them to its workers):
    the same results as section_matcher pair by pair, in job order; keywords per job; an exception of a job surfaces"""
    from scipy.ndimage import map_coordinates
    from feabas_amd import matcher
    rng = np.random.default_rng(23)
    (v0, t0, v1, t1), (M0, M1), _ = _island_pair(rng)
    SH, SW = 600, 1080
    base = _texture(rng, SH, SW)
    yy, xx = np.meshgrid(np.arange(SH, dtype=np.float64), np.arange(SW, dtype=np.float64), indexing='ij')
    img1 = np.clip(np.rint(map_coordinates(base.astype(np.float32), [yy + 2.0 * np.cos(xx / 150.0), xx + 2.5 * np.sin(yy / 120.0)], order=1, mode='nearest')), 0, 255).astype(np.uint8)
    for M in (M0, M1):
        M.material_ids = None; M.material_names = {}; M.material_area_constraints = {}
    M0.locked = True
    kws = [dict(spacings=[150, 60], conf_thresh=0.3, residue_len=3.0), dict(spacings=[120], conf_thresh=0.3, residue_len=3.0, min_boundary_distance=10)]
    ref = [matcher.section_matcher(M0.copy(), M1.copy(), base, img1, **kw) for kw in kws]
    jobs = [(M0.copy(), M1.copy(), base, img1, kws[k % 2]) for k in range(5)]
    got = matcher.section_matcher_batch(jobs, threads=3)
    assert len(got) == 5
    for k, g in enumerate(got):
        e = ref[k % 2]
        assert g[0].shape == e[0].shape and g[0].shape[0] > 10
        np.testing.assert_array_equal(g[0], e[0]); np.testing.assert_array_equal(g[1], e[1]); np.testing.assert_array_equal(g[2], e[2])
    assert matcher.section_matcher_batch([]) == []
    with pytest.raises(TypeError):
        matcher.section_matcher_batch([(M0.copy(), M1.copy(), base, img1, dict(no_such_keyword=1))] * 2, threads=2)
    matcher.stitching_matcher_batch_release()
