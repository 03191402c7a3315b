"""The PRODUCT's block-matcher loop on the CPU against the reference's (golden G23): feabas_amd.matcher.iterative_xcorr_matcher_w_mesh
with everything of its own that runs on the host -- the block distributor, the round stepper of the library, _PairRelaxation, Link
(point location, barycentric coordinates, residue weights, masks), Mesh's gears, the final extraction of the matches -- and two
stand-ins: the block matcher is the script of the fixture (the reference's was scripted the same way), and SLM.optimize_linear, the
one step that needs the device, is solved exactly through the oracle's assembly of the SAME meshes and links (the device solve has
its own parity tests against that oracle, tests/test_gpu_fem.py).  What the reference's loop did around its block matches --
blocks, flags, the field of mesh 1 going into every round, final matches and weights -- must come out of the product's loop."""
import numpy as np
import pytest

from conftest import load_golden
import feabas_amd.constant as const
from feabas_amd import matcher, optimizer, renderer
from feabas_amd.mesh import Mesh
from oracle import fem_ref
from test_oracle_golden import _g23_scripted_block_matches

GEARS = (const.MESH_GEAR_INITIAL, const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING, const.MESH_GEAR_STAGING)


def _mirror(m):
    r = fem_ref.RefMesh(m._vertices[const.MESH_GEAR_INITIAL], m.triangles, uid=m.uid, locked=m.locked, soft_factor=m.soft_factor,
                        stiffness_multiplier=m.stiffness_multiplier, nu=m.poisson_ratio)
    for g in GEARS:
        r._v[g] = m._vertices[g]
        r._off[g] = np.asarray(m._offsets[g], dtype=np.float64).reshape(1, 2)
    return r


def _exact_optimize_linear(self, **kwargs):
    """SLM.optimize_linear (optimizer.py:1257-1437) with the linear system of the oracle, solved directly"""
    shape_gear = kwargs.get('shape_gear', const.MESH_GEAR_FIXED)
    target_gear = kwargs.get('target_gear', const.MESH_GEAR_MOVING)
    start_gear = kwargs.get('start_gear', target_gear)
    refs = [_mirror(m) for m in self.meshes]
    at = {m.uid: k for k, m in enumerate(self.meshes)}
    links = []
    for lk in self.links:
        if getattr(lk, '_disabled', False):
            continue
        rl = fem_ref.RefLink(refs[at[lk.uids[0]]], refs[at[lk.uids[1]]], lk._tid0, lk._tid1, lk._B0, lk._B1, weight=lk._weight, strain=lk.strain)
        rl.residue_weight = lk._residue_weight
        links.append(rl)
    A, b, _ = fem_ref.linear_system(refs, links, kwargs.get('stiffness_lambda', self._stiffness_lambda), kwargs.get('crosslink_lambda', self._crosslink_lambda),
                                    shape_gear, start_gear, target_gear)
    dd = fem_ref.solve_direct(A, np.asarray(b, dtype=np.float64))
    cost = (float(np.linalg.norm(b)), float(np.linalg.norm(A @ dd - b)))
    self.last_solve = dict(iters=None, relres=cost[1] / cost[0] if cost[0] else 0.0)
    if cost[1] < cost[0]:
        offs, _ = fem_ref.index_offsets(refs)
        for m, o in zip(self.meshes, offs):
            if o >= 0:
                m.set_field(dd[o:o + 2 * m.num_vertices].reshape(-1, 2), gear=(start_gear, target_gear))
    return cost


def _exact_local_stiffness(self, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING), tri_mask=None, **kwargs):
    """Mesh.stiffness_matrix_local_normalized (the masked assembly of relax_mesh, mesh.py / optimizer.py:2110-2154) through the oracle"""
    return _mirror(self).stiffness_matrix_local_normalized(gear=gear, tri_mask=tri_mask)


def _exact_solve(A, b, solver='minres', x0=None, tol=1e-7, atol=None, maxiter=None, M=None, **kwargs):
    """optimizer.solve with the free block solved directly (the held degrees of freedom stay zero)"""
    from scipy import sparse
    A = sparse.csr_matrix(A)
    b = np.asarray(b, dtype=np.float64)
    dof = kwargs.get('extra_dof_constraint', None)
    x = np.zeros_like(b)
    if dof is None:
        return fem_ref.solve_direct(A, b)
    if np.any(b[dof]):
        x[dof] = fem_ref.solve_direct(A[dof][:, dof], b[dof])
    return x


class _NoImage(renderer.ResidentImage):
    def __init__(self):
        pass

    def free(self):
        pass


@pytest.mark.parametrize('case', ['huber', 'threshold3', 'no_residue'])
def test_product_matcher_loop_between_the_block_matches_vs_reference(monkeypatch, case):
    g = load_golden('g23_matcher_loop.npz')
    res_len, seed, ox, oy, thr = g[f'{case}_params']
    m0 = Mesh(g['v0'], g['t0'], uid=0)
    m0.apply_translation((ox, oy), const.MESH_GEAR_FIXED)
    m0.lock()
    m1 = Mesh(g['v1'].copy(), g['t1'], uid=1)
    seen = []

    def scripted(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **kw):
        k = len(seen)
        seen.append(dict(bboxes0=np.array(bboxes0), bboxes1=np.array(bboxes1), pad=bool(kw.get('pad')), subpixel=bool(kw.get('subpixel')),
                         field1=mesh1.vertices_w_offset(const.MESH_GEAR_MOVING) - mesh1.vertices_w_offset(const.MESH_GEAR_INITIAL)))
        return _g23_scripted_block_matches(k, bboxes0, bboxes1, seed)
    monkeypatch.setattr(matcher, 'bboxes_mesh_renderer_matcher', scripted)
    monkeypatch.setattr(optimizer.SLM, 'optimize_linear', _exact_optimize_linear)

    def no_device_relaxation(*a, **k):
        raise AssertionError('relax_mesh needs the device: the fixture does not deform a triangle that far')
    monkeypatch.setattr(optimizer, 'relax_mesh', no_device_relaxation)
    monkeypatch.setattr(Mesh, 'stiffness_energy', lambda self, fields, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING): _oracle_energy(self, fields, gear))
    xy0, xy1, wt, strain = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, _NoImage(), _NoImage(), spacings=g[f'{case}_spacings'], distributor='cartesian_bbox',
                                                                  conf_thresh=0.3, residue_len=float(res_len), residue_mode='threshold' if thr else 'huber',
                                                                  stiffness_lambda=0.5, min_num_blocks=2, compute_strain=True)
    np.testing.assert_allclose(strain, float(g[f'{case}_strain']), rtol=1e-5)
    n = int(g[f'{case}_nrounds'])
    assert len(seen) == n
    for k, r in enumerate(seen):
        np.testing.assert_allclose(r['bboxes0'], g[f'{case}_r{k}_bboxes0'], atol=1e-6)
        np.testing.assert_allclose(r['bboxes1'], g[f'{case}_r{k}_bboxes1'], atol=1e-6)
        assert [r['pad'], r['subpixel']] == g[f'{case}_r{k}_flags'].tolist()
        scale = max(1.0, np.abs(g[f'{case}_r{k}_field1']).max())
        np.testing.assert_allclose(r['field1'], g[f'{case}_r{k}_field1'], atol=1e-6 * scale)
    scale = np.abs(g[f'{case}_field1_final']).max()
    np.testing.assert_allclose(m1.vertices_w_offset(const.MESH_GEAR_MOVING) - m1.vertices_w_offset(const.MESH_GEAR_INITIAL), g[f'{case}_field1_final'], atol=1e-6 * scale)
    assert xy0.shape == g[f'{case}_xy0'].shape
    np.testing.assert_allclose(xy0, g[f'{case}_xy0'], atol=1e-5); np.testing.assert_allclose(xy1, g[f'{case}_xy1'], atol=1e-5)
    np.testing.assert_allclose(wt, g[f'{case}_weight'], atol=1e-5)


@pytest.mark.parametrize('case', ['rigid', 'deformed', 'three'])
def test_product_strip_loop_between_the_block_matches_vs_reference(monkeypatch, case):
    """the same through stitching_matcher's set-up (golden G24, matcher.py:353-364): two cartesian meshes of Mesh.from_bbox, mesh 0
    translated by the global translation and locked, the general-mesh loop on them -- what feabas_amd.matcher._stitching_matcher_general
    runs for strips of unequal shape, and the statement the batched strip pipeline is tested against on the device"""
    from test_oracle_golden import _g24_scripted_strip_blocks
    g = load_golden('g24_strip_loop.npz')
    H, W, tx, ty, res_len = g[f'{case}_params']
    H, W = int(H), int(W)
    spacings = g[f'{case}_spacings']
    m0 = Mesh.from_bbox((0, 0, W, H), cartesian=True, mesh_size=float(np.min(spacings)), min_num_blocks=2, uid=0)
    m1 = Mesh.from_bbox((0, 0, W, H), cartesian=True, mesh_size=float(np.min(spacings)), min_num_blocks=2, uid=1)
    m0.apply_translation((tx, ty), const.MESH_GEAR_FIXED)
    m0.lock()
    seen = []

    def scripted(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **kw):
        k = len(seen)
        seen.append(dict(bboxes0=np.array(bboxes0), bboxes1=np.array(bboxes1), pad=bool(kw.get('pad')), subpixel=bool(kw.get('subpixel')),
                         field1=mesh1.vertices_w_offset(const.MESH_GEAR_MOVING) - mesh1.vertices_w_offset(const.MESH_GEAR_INITIAL)))
        if case == 'rigid' and k > 0:
            # the knife edge of tests/test_oracle_golden.py::test_g24_...: after a whole-pixel round the lattice turns on the sign of the
            # solver's residual (here: of a direct solve); the round goes on with the reference's recorded blocks
            bboxes0, bboxes1 = g[f'{case}_r{k}_bboxes0'], g[f'{case}_r{k}_bboxes1']
        dx, dy, cf = _g24_scripted_strip_blocks(case, k, bboxes0, bboxes1, H, W)
        p0, p1 = matcher.block_displacements_to_points(bboxes0, bboxes1, dx, dy)
        return p0, p1, cf
    monkeypatch.setattr(matcher, 'bboxes_mesh_renderer_matcher', scripted)
    monkeypatch.setattr(optimizer.SLM, 'optimize_linear', _exact_optimize_linear)
    # relax_first (matcher.py:730, optimizer.py:763-772): the product's own relax_mesh_most_deformed / relax_mesh around the two steps that
    # need the device -- the masked assembly and the solve of the free block -- which go through the oracle
    monkeypatch.setattr(Mesh, 'stiffness_matrix_local_normalized', _exact_local_stiffness)
    monkeypatch.setattr(optimizer, 'solve', _exact_solve)
    xy0, xy1, wt, _ = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, _NoImage(), _NoImage(), spacings=spacings, distributor='cartesian_bbox', conf_thresh=0.33,
                                                             residue_len=float(res_len), residue_mode='huber', min_num_blocks=2, compute_strain=False)
    n = int(g[f'{case}_nrounds'])
    assert len(seen) == n
    for k, r in enumerate(seen):
        lattice_tol = 1 if (case == 'rigid' and k > 0) else 1e-6
        assert r['bboxes0'].shape == g[f'{case}_r{k}_bboxes0'].shape
        np.testing.assert_allclose(r['bboxes0'], g[f'{case}_r{k}_bboxes0'], atol=lattice_tol)
        np.testing.assert_allclose(r['bboxes1'], g[f'{case}_r{k}_bboxes1'], atol=lattice_tol)
        assert [r['pad'], r['subpixel']] == g[f'{case}_r{k}_flags'].tolist()
        want = g[f'{case}_r{k}_field1']
        np.testing.assert_allclose(r['field1'], want, atol=1e-6 * max(1.0, np.abs(want).max()))
    assert xy0.shape == g[f'{case}_xy0'].shape
    np.testing.assert_allclose(xy0, g[f'{case}_xy0'], atol=1e-5); np.testing.assert_allclose(xy1, g[f'{case}_xy1'], atol=1e-5)
    np.testing.assert_allclose(wt, g[f'{case}_weight'], atol=1e-5)


def _section_script(rnd, bboxes0, bboxes1):
    """block matches of a section pair as a closed form of the block centres and the round: a smooth field that shrinks from round to
    round, hash-like confidences (some below the threshold) and a few outliers for the residue weights"""
    b0 = np.asarray(bboxes0, dtype=np.float64); b1 = np.asarray(bboxes1, dtype=np.float64)
    c0 = 0.5 * (b0[:, :2] + b0[:, 2:])
    amp = (3.0, 0.8, 0.2)[min(rnd, 2)]
    dx = amp * np.sin(c0[:, 1] / 260.0 + 0.3 + rnd) + 0.4 * amp * (c0[:, 0] / 1000.0)
    dy = amp * np.cos(c0[:, 0] / 300.0 - rnd) - 0.3 * amp * (c0[:, 1] / 600.0)
    h = np.abs(np.modf(np.sin(np.round(c0[:, 0]) * 12.9898 + np.round(c0[:, 1]) * 78.233 + 17.0 * rnd) * 43758.5453)[0])
    conf = (0.2 + 0.8 * h).astype(np.float32)
    dxy = np.stack((dx, dy), axis=-1)
    dxy[h > 0.94] += np.array([7.0, -5.0])
    p0, p1 = matcher.block_displacements_to_points(b0, b1, dxy[:, 0], dxy[:, 1])
    return p0, p1, conf


@pytest.mark.parametrize('locked', [True, False])
def test_product_section_matcher_vs_oracle_with_scripted_blocks(monkeypatch, locked):
    """SURVEY row a8 end to end on the CPU: matcher.section_matcher -- the region-aware distributor ('cartesian_region' with a boundary
    distance) inside the loop, on meshes that move from round to round -- against oracle/region_ref.section_match, the restatement of the
    reference's loop, with the same scripted block matches on both sides and the product's device steps solved through the oracle's
    assembly: the same blocks every round (the oracle takes the lattice phase from the product's blocks, like the distributor test),
    the same field of section 1 after every relaxation, the same final matches and weights.  With section 0 locked and with both
    sections free (the floating system of matcher.py:551: the oracle takes the Jacobi-Krylov limit, the stand-in here the
    minimum-norm-in-diag(A) solution of the same singular system)."""
    from scipy.spatial import Delaunay                                       # noqa: F401  (used by the helper module)
    import test_gpu_renderer as tgr
    from oracle import region_ref
    rng = np.random.default_rng(17)
    (v0, t0, v1, t1), (M0, M1), _ = tgr._island_pair(rng)
    for M in (M0, M1):
        M.material_ids = None; M.material_names = {}; M.material_area_constraints = {}
    M0.locked = locked

    def solve_like_the_oracle(self, **kwargs):
        if any(m.locked for m in self.meshes):
            return _exact_optimize_linear(self, **kwargs)
        # a floating pair: the fixed point a Jacobi-preconditioned Krylov method reaches on the singular system
        refs = [_mirror(m) for m in self.meshes]
        at = {m.uid: k for k, m in enumerate(self.meshes)}
        links = []
        for lk in self.links:
            rl = fem_ref.RefLink(refs[at[lk.uids[0]]], refs[at[lk.uids[1]]], lk._tid0, lk._tid1, lk._B0, lk._B1, weight=lk._weight, strain=lk.strain)
            rl.residue_weight = lk._residue_weight
            links.append(rl)
        A, b, _ = fem_ref.linear_system(refs, links, kwargs.get('stiffness_lambda', self._stiffness_lambda), -1.0, const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING,
                                        const.MESH_GEAR_MOVING)
        dd = region_ref._solve_jacobi_krylov_limit(A, np.asarray(b, dtype=np.float64))
        self.last_solve = dict(iters=None, relres=0.0)
        offs, _ = fem_ref.index_offsets(refs)
        for m, o in zip(self.meshes, offs):
            if o >= 0:
                m.set_field(dd[o:o + 2 * m.num_vertices].reshape(-1, 2), gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_MOVING))
        return (float(np.linalg.norm(b)), float(np.linalg.norm(A @ dd - b)))
    rounds = []

    def scripted_product(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **kw):
        rounds.append(dict(bboxes0=np.array(bboxes0), bboxes1=np.array(bboxes1), pad=bool(kw.get('pad')), subpixel=bool(kw.get('subpixel'))))
        return _section_script(len(rounds) - 1, bboxes0, bboxes1)
    monkeypatch.setattr(matcher, 'bboxes_mesh_renderer_matcher', scripted_product)
    monkeypatch.setattr(optimizer.SLM, 'optimize_linear', solve_like_the_oracle)
    monkeypatch.setattr(Mesh, 'stiffness_matrix_local_normalized', _exact_local_stiffness)
    monkeypatch.setattr(optimizer, 'solve', _exact_solve)
    kw = dict(spacings=[150, 60], conf_thresh=0.3, residue_len=3.0, min_boundary_distance=12, stiffness_lambda=0.5)
    trace = []
    xy0, xy1, wt, _ = matcher.section_matcher(M0, M1, _NoImage(), _NoImage(), compute_strain=False, stiffness_multiplier_threshold=0, trace=trace, **kw)
    assert xy0 is not None and len(trace) == 2 and trace[1]['blocks'] > 80
    r0 = fem_ref.RefMesh(v0, t0, uid=0, locked=locked); r1 = fem_ref.RefMesh(v1, t1, uid=1)
    orounds = []

    def scripted_oracle(rnd, a, b, bb0, bb1, pad, subpixel, tol):
        orounds.append(dict(bboxes0=np.array(bb0), bboxes1=np.array(bb1), pad=bool(pad), subpixel=bool(subpixel)))
        return _section_script(rnd, bb0, bb1)
    otrace = []
    ex0, ex1, ewt, _ = region_ref.section_match(r0, r1, None, None, compute_strain=False, anchor_rounds=[t['bboxes0'] for t in trace], trace=otrace, res=None,
                                                block_matcher=scripted_oracle, **kw)
    assert len(orounds) == len(rounds) == 2
    for g, e, tg, te in zip(rounds, orounds, trace, otrace):
        np.testing.assert_allclose(g['bboxes0'], e['bboxes0'], atol=1e-6)
        np.testing.assert_allclose(g['bboxes1'], e['bboxes1'], atol=1e-6)
        assert (g['pad'], g['subpixel']) == (e['pad'], e['subpixel'])
        if 'field1' in te:
            np.testing.assert_allclose(tg['field1'], te['field1'], atol=1e-6 * max(1.0, np.abs(te['field1']).max()))
    assert xy0.shape == ex0.shape and xy0.shape[0] > 60
    np.testing.assert_allclose(xy0, ex0, atol=1e-5); np.testing.assert_allclose(xy1, ex1, atol=1e-5)
    np.testing.assert_allclose(wt, ewt, atol=1e-5)


def test_product_global_translation_matcher_vs_reference(monkeypatch):
    """matcher.global_translation_matcher's host logic (matcher.py:138-221: the second shot on ~6 sub-blocks when the whole-image
    confidence is low, block extents grown to a common size and slid inside the image, offsets, the best block taking over) against the
    reference's outputs (golden G3) with the correlation itself -- the device step -- served by the oracle's xcorr_fft (pinned by G1)"""
    from oracle import ncc_ref
    monkeypatch.setattr(matcher, 'xcorr_fft', lambda a, b, **kw: ncc_ref.xcorr_fft(a, b, conf_mode=kw.get('conf_mode', const.FFT_CONF_MIRROR),
                                                                                    pad=kw.get('pad', True), subpixel=kw.get('subpixel', False)))
    g = load_golden('g3_global.npz')
    np.testing.assert_allclose(matcher.global_translation_matcher(g['d0'], g['d1'], conf_thresh=0.3), g['plain'], atol=1e-4)
    np.testing.assert_allclose(matcher.global_translation_matcher(g['d0'], g['e1'], conf_thresh=2.0), g['fallback'], atol=1e-4)
    np.testing.assert_allclose(matcher.global_translation_matcher(g['d0'], g['f1'], conf_thresh=2.0), g['unequal'], atol=1e-4)


def test_product_matcher_loop_with_initial_matches_vs_reference(monkeypatch):
    """the seeded start of the loop (matcher.py:552-563: a link from the initial matches, optimize_affine_cascade at the FIXED gear, rigid
    anneal of the MOVING gear, one relaxation) and the rounds after it against the reference (golden G30): section 1 arrives rotated and
    shifted in its own frame; both gears of mesh 1 going into every round, blocks, flags, final matches and weights"""
    from collections import namedtuple
    from test_oracle_golden import _g23_scripted_block_matches as script
    g = load_golden('g30_seeded_loop.npz')
    g23 = load_golden('g23_matcher_loop.npz')
    m0 = Mesh(g23['v0'], g23['t0'], uid=0)
    m0.lock()
    m1 = Mesh(g['v1'].copy(), g23['t1'], uid=1)
    seen = []

    def scripted(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **kw):
        k = len(seen)
        seen.append(dict(bboxes0=np.array(bboxes0), bboxes1=np.array(bboxes1), pad=bool(kw.get('pad')), subpixel=bool(kw.get('subpixel')),
                         moving1=mesh1.vertices_w_offset(const.MESH_GEAR_MOVING), fixed1=mesh1.vertices_w_offset(const.MESH_GEAR_FIXED)))
        return script(k, bboxes0, bboxes1, 5.0)
    monkeypatch.setattr(matcher, 'bboxes_mesh_renderer_matcher', scripted)
    monkeypatch.setattr(optimizer.SLM, 'optimize_linear', _exact_optimize_linear)
    monkeypatch.setattr(Mesh, 'stiffness_matrix_local_normalized', _exact_local_stiffness)
    monkeypatch.setattr(optimizer, 'solve', _exact_solve)
    Match = namedtuple('Match', ('xy0', 'xy1', 'weight'))
    xy0, xy1, wt, _ = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, _NoImage(), _NoImage(), spacings=g['spacings'], distributor='cartesian_bbox', conf_thresh=0.3,
                                                             residue_len=3.0, residue_mode='huber', compute_strain=False, stiffness_lambda=0.5, min_num_blocks=2,
                                                             initial_matches=Match(g['ixy0'], g['ixy1'], g['iw']))
    assert len(seen) == int(g['nrounds'])
    for k, r in enumerate(seen):
        np.testing.assert_allclose(r['fixed1'], g[f'r{k}_fixed1'], atol=1e-6)
        np.testing.assert_allclose(r['moving1'], g[f'r{k}_moving1'], atol=1e-6)
        np.testing.assert_allclose(r['bboxes0'], g[f'r{k}_bboxes0'], atol=1e-6)
        np.testing.assert_allclose(r['bboxes1'], g[f'r{k}_bboxes1'], atol=1e-6)
        assert [r['pad'], r['subpixel']] == g[f'r{k}_flags'].tolist()
    np.testing.assert_allclose(m1.vertices_w_offset(const.MESH_GEAR_MOVING), g['moving1_final'], atol=1e-6)
    assert xy0.shape == g['xy0'].shape
    np.testing.assert_allclose(xy0, g['xy0'], atol=1e-5); np.testing.assert_allclose(xy1, g['xy1'], atol=1e-5)
    np.testing.assert_allclose(wt, g['weight'], atol=1e-5)


@pytest.mark.parametrize('case', ['decay', 'enlarge', 'skip_shrink', 'dwell'])
def test_product_matcher_loop_options_vs_reference(monkeypatch, case):
    """the keywords that change the course of the loop, against the reference (golden G31): link_weight_decay (earlier rounds' links stay,
    decayed), allow_enlarge (a first round whose displacement outruns the largest spacing is repeated with larger blocks before anything
    is linked), max_spacing_skip with a shrink factor and fixed padding, allow_dwell with a fixed sub-pixel flag and three blocks minimum"""
    import json
    from test_oracle_golden import _g23_scripted_block_matches as script
    g = load_golden('g31_loop_options.npz')
    g23 = load_golden('g23_matcher_loop.npz')
    kw = json.loads(str(g['cases']))[case]
    m0 = Mesh(g23['v0'], g23['t0'], uid=0)
    m0.apply_translation((2.0, -1.0), const.MESH_GEAR_FIXED)
    m0.lock()
    m1 = Mesh(g23['v1'].copy(), g23['t1'], uid=1)
    seen = []

    def scripted(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **k_):
        k = len(seen)
        seen.append(dict(bboxes0=np.array(bboxes0), bboxes1=np.array(bboxes1), pad=bool(k_.get('pad')), subpixel=bool(k_.get('subpixel')),
                         field1=mesh1.vertices_w_offset(const.MESH_GEAR_MOVING) - mesh1.vertices_w_offset(const.MESH_GEAR_INITIAL)))
        return script(k, bboxes0, bboxes1, 7.0)
    monkeypatch.setattr(matcher, 'bboxes_mesh_renderer_matcher', scripted)
    monkeypatch.setattr(optimizer.SLM, 'optimize_linear', _exact_optimize_linear)
    monkeypatch.setattr(Mesh, 'stiffness_matrix_local_normalized', _exact_local_stiffness)
    monkeypatch.setattr(optimizer, 'solve', _exact_solve)
    xy0, xy1, wt, _ = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, _NoImage(), _NoImage(), distributor='cartesian_bbox', conf_thresh=0.3, compute_strain=False,
                                                             stiffness_lambda=0.5, **kw)
    n = int(g[f'{case}_nrounds'])
    assert len(seen) == n
    for k, r in enumerate(seen):
        assert r['bboxes0'].shape == g[f'{case}_r{k}_bboxes0'].shape, k
        np.testing.assert_allclose(r['bboxes0'], g[f'{case}_r{k}_bboxes0'], atol=1e-6)
        np.testing.assert_allclose(r['bboxes1'], g[f'{case}_r{k}_bboxes1'], atol=1e-6)
        assert [r['pad'], r['subpixel']] == g[f'{case}_r{k}_flags'].tolist(), k
        want = g[f'{case}_r{k}_field1']
        np.testing.assert_allclose(r['field1'], want, atol=1e-6 * max(1.0, np.abs(want).max()))
    want = g[f'{case}_field1_final']
    np.testing.assert_allclose(m1.vertices_w_offset(const.MESH_GEAR_MOVING) - m1.vertices_w_offset(const.MESH_GEAR_INITIAL), want, atol=1e-6 * max(1.0, np.abs(want).max()))
    assert xy0.shape == g[f'{case}_xy0'].shape
    np.testing.assert_allclose(xy0, g[f'{case}_xy0'], atol=1e-5); np.testing.assert_allclose(xy1, g[f'{case}_xy1'], atol=1e-5)
    np.testing.assert_allclose(wt, g[f'{case}_weight'], atol=1e-5)


def _oracle_cost(self, stiffness_lambda, crosslink_lambda):
    """SLM.cost (optimizer.py:1593-1601): ||lc rhs - ls stress|| of the system at the current gears, through the oracle's assembly"""
    refs = [_mirror(m) for m in self.meshes]
    at = {m.uid: k for k, m in enumerate(self.meshes)}
    links = []
    for lk in self.links:
        rl = fem_ref.RefLink(refs[at[lk.uids[0]]], refs[at[lk.uids[1]]], lk._tid0, lk._tid1, lk._B0, lk._B1, weight=lk._weight, strain=lk.strain)
        rl.residue_weight = lk._residue_weight
        links.append(rl)
    _, _, (K, stress, C, rhs, ls, lc) = fem_ref.linear_system(refs, links, stiffness_lambda, crosslink_lambda, const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING,
                                                              const.MESH_GEAR_MOVING)
    return float(np.linalg.norm(lc * rhs - ls * stress))


@pytest.mark.parametrize('case', ['online_huber', 'online_threshold', 'plain_steps', 'rigid_anneal'])
def test_product_newton_driver_vs_reference(monkeypatch, case):
    """SLM.optimize_Newton_Raphson / optimize_elastic(online_anneal=True) as a DRIVER (optimizer.py:1440-1555) against the reference
    (golden G32) on linear meshes: the per-step ladders of tolerances, lambdas and residue lengths, the resting shape annealed from the
    STAGING gear (with relax_higly_deformed before it), the links re-weighted by their residues between the steps, the cost floor and the
    early last step.  Host control flow of the product; the assembly, the cost and the inner solves -- the device steps -- through the
    oracle (inner solves exact on both sides)."""
    import json
    g = load_golden('g32_newton_driver.npz')
    g23 = load_golden('g23_matcher_loop.npz')
    kw = dict(json.loads(str(g['cases']))[case])
    m0 = Mesh(g23['v0'], g23['t0'], uid=0)
    m0.apply_translation((2.0, -1.0), const.MESH_GEAR_FIXED)
    m0.lock()
    m1 = Mesh(g23['v1'].copy(), g23['t1'], uid=1)
    monkeypatch.setattr(optimizer.SLM, 'optimize_linear', _exact_optimize_linear)
    monkeypatch.setattr(optimizer.SLM, '_assemble', lambda self, *a, **k: None)
    monkeypatch.setattr(optimizer.SLM, 'cost', _oracle_cost)
    monkeypatch.setattr(Mesh, 'stiffness_matrix_local_normalized', _exact_local_stiffness)
    monkeypatch.setattr(optimizer, 'solve', _exact_solve)
    opt = optimizer.SLM([m0, m1], stiffness_lambda=0.7)
    opt.add_link_from_coordinates(0, 1, g['xy0'], g['xy1'], gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_INITIAL), weight=g['w'])
    cost = opt.optimize_Newton_Raphson(**kw) if kw.pop('_newton', False) else opt.optimize_elastic(**kw)
    np.testing.assert_allclose(cost[0], g[f'{case}_cost'][0], rtol=1e-7)
    np.testing.assert_allclose(cost[1], g[f'{case}_cost'][1], rtol=1e-4, atol=1e-6 * g[f'{case}_cost'][0])
    for k, gear in dict(i=const.MESH_GEAR_INITIAL, f=const.MESH_GEAR_FIXED, m=const.MESH_GEAR_MOVING, s=const.MESH_GEAR_STAGING).items():
        np.testing.assert_allclose(m1.vertices_w_offset(gear), g[f'{case}_{k}'], atol=1e-6, err_msg=k)
    np.testing.assert_allclose(opt.links[0].weight(use_mask=False), g[f'{case}_lw'], atol=1e-6)


def _g16_mesh(g):
    return Mesh(g['v'], g['t'], stiffness_multiplier=g['mult'], moving_vertices=g['vmov'].copy(), moving_offset=g['moff'].copy(), uid=3)


@pytest.mark.parametrize('which', ['ft', 'fv'])
def test_product_relax_mesh_vs_reference(monkeypatch, which):
    """optimizer.relax_mesh (optimizer.py:2110-2154) on the CPU against the reference's converged result (golden G16): the product's free-
    vertex selection, the rigid re-alignment of the resting shape around the solve, the field applied to the free vertices only, the locked
    mesh that is relaxed all the same -- with the masked assembly and the free-block solve through the oracle"""
    g = load_golden('g16_relax.npz')
    gear = (const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)
    monkeypatch.setattr(Mesh, 'stiffness_matrix_local_normalized', _exact_local_stiffness)
    monkeypatch.setattr(optimizer, 'solve', _exact_solve)
    m = _g16_mesh(g)
    mod = optimizer.relax_mesh(m, free_triangles=g['free_tri'], gear=gear) if which == 'ft' else optimizer.relax_mesh(m, free_vertices=g['free_vtx'], gear=gear)
    assert mod == bool(g[f'{which}_modified'])
    moved = np.abs(g[f'{which}_vmov'] - g['vmov']).max()
    np.testing.assert_allclose(m.vertices(gear[1]), g[f'{which}_vmov'], atol=1e-6 * moved)
    np.testing.assert_array_equal(m.offset(gear[1]), g[f'{which}_moff'])
    np.testing.assert_array_equal(m.vertices(gear[0]), g['v'])
    m2 = _g16_mesh(g)
    m2.lock()
    assert optimizer.relax_mesh(m2, free_vertices=g['free_vtx'], gear=gear) and m2.locked
    np.testing.assert_allclose(m2.vertices(gear[1]), g['fv_vmov'], atol=1e-6 * moved)
    assert not optimizer.relax_mesh(m, gear=gear)
    assert not optimizer.relax_mesh(m, free_triangles=np.zeros(g['t'].shape[0], dtype=bool), gear=gear)


@pytest.mark.parametrize('name,kw', [('md_flip', dict(deform_cutoff=-1)), ('md_cut', dict(deform_cutoff=0.35)), ('md_iqr', dict(deform_cutoff=0.35, iqr=1.5))])
def test_product_relax_most_deformed_vs_reference(monkeypatch, name, kw):
    """relax_mesh_most_deformed (optimizer.py:2157-2190): the region the product frees (flipped triangles only / beyond the cutoff / capped by
    the inter-quartile rule) and the relaxed field against the reference's converged result (golden G16) on the CPU"""
    g = load_golden('g16_relax.npz')
    gear = (const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)
    monkeypatch.setattr(Mesh, 'stiffness_matrix_local_normalized', _exact_local_stiffness)
    monkeypatch.setattr(optimizer, 'solve', _exact_solve)
    m = _g16_mesh(g)
    assert optimizer.relax_mesh_most_deformed(m, gear=gear, **kw) == bool(g[f'{name}_modified'])
    np.testing.assert_allclose(m.vertices(gear[1]), g[f'{name}_vmov'], atol=1e-6 * np.abs(g[f'{name}_vmov'] - g['vmov']).max())
    if name == 'md_flip':
        assert (m.triangle_area_deform(gear) > 0).all()
        assert not optimizer.relax_mesh_most_deformed(m, gear=gear, deform_cutoff=-1)


def _oracle_energy(self, fields, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)):
    """Mesh.stiffness_energy: x^T K x with the oracle's stiffness matrix of this mesh at `gear`"""
    K, _ = _mirror(self).stiffness_matrix(gear=gear)
    return [float(np.ravel(x) @ (K @ np.ravel(x))) for x in fields]


def test_product_strain_of_matches_vs_reference(monkeypatch):
    """matcher._strain_of_matches (matcher.py:752-777: the untouched pair brought together by a rigid affine cascade, relaxed once, strain =
    sqrt(dv^T K dv / v0^T K v0) over the free mesh) on the CPU against the reference's chain (golden G13): the product's cascade, anneal,
    gear handling and energy bookkeeping; the relaxation and the two energies through the oracle"""
    g = load_golden('g13_strain.npz')
    m0 = Mesh(g['st_v'], g['st_tri'], uid=0)
    m0.apply_translation(g['st_t0'], const.MESH_GEAR_FIXED)
    m0.lock()
    m1 = Mesh(g['st_v'], g['st_tri'], uid=1)
    lk = optimizer.Link(m0, m1, g['st_tid0'], g['st_tid1'], g['st_B0'], g['st_B1'], weight=g['st_w'])
    xy0 = lk.xy0(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True)
    xy1 = lk.xy1(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True)
    monkeypatch.setattr(optimizer.SLM, 'optimize_linear', _exact_optimize_linear)
    monkeypatch.setattr(Mesh, 'stiffness_energy', _oracle_energy)
    strain = matcher._strain_of_matches(m0, m1, xy0, xy1, lk.weight(use_mask=False), 1.0)
    np.testing.assert_allclose(m1.vertices(const.MESH_GEAR_FIXED), g['st_v_fixed'], atol=1e-8)
    np.testing.assert_allclose(m1.offset(const.MESH_GEAR_FIXED), g['st_off_fixed'], atol=1e-8)
    np.testing.assert_allclose(m1.vertices(const.MESH_GEAR_MOVING), g['st_v_moving'], atol=1e-6)
    np.testing.assert_allclose(strain, float(g['st_strain']), rtol=1e-6)


def test_product_render_weights_vs_reference(monkeypatch):
    """render weights of materials (material.py:27-30; mesh.py:1836-1859, 2168-2170; optimizer.py:59) against the reference (golden G33):
    mesh 1 carries a band of a material that weighs 1e-3 in rendering (like soft / wrinkled tissue in the default material table) and a
    corner that is not rendered at all.  The per-triangle weights and the render masks by threshold; which matches Link.from_coordinates
    keeps with its default threshold 0.1, with 0 and with 5e-4; and the matcher loop with render_weight_threshold = 0.1, where the
    matches that land in the band are dropped round after round (no final match in it)."""
    from test_oracle_golden import _g23_scripted_block_matches as script
    g = load_golden('g33_render_weights.npz')
    g23 = load_golden('g23_matcher_loop.npz')

    def mesh1():
        return Mesh(g23['v1'].copy(), g['t1'], uid=1, material_ids=g['mids'], material_names={'default': 0, 'soft_look': 5, 'hidden': 6},
                    material_render_weights={'soft_look': 1.0e-3, 'hidden': -1.0})           # (render = False: -(render_weight + 1))
    m1 = mesh1()
    np.testing.assert_array_equal(m1.weight_multiplier_for_render(), g['weights'])
    for thr in (0.0, 0.1, 1.0e-3, 0.5e-3):
        np.testing.assert_array_equal(m1.triangle_mask_for_render(render_weight_threshold=thr), g[f'mask_{thr}'])
    sub = m1.submesh(m1.triangle_mask_for_render(render_weight_threshold=0.1))
    assert sub.num_triangles == int(g['mask_0.1'].sum()) and np.all(sub.weight_multiplier_for_render() == 1.0)
    assert np.array_equal(m1.copy().weight_multiplier_for_render(), g['weights'])
    m0 = Mesh(g23['v0'], g23['t0'], uid=0)
    for tag, kw in (('default', {}), ('zero', dict(render_weight_threshold=0)), ('low', dict(render_weight_threshold=0.5e-3))):
        lk, mask = optimizer.Link.from_coordinates(m0, m1, g['lp'], g['lq'], weight=g['lw'], **kw)
        np.testing.assert_array_equal(mask, g[f'link_{tag}_mask'])
        np.testing.assert_allclose(lk.xy1(gear=const.MESH_GEAR_INITIAL, use_mask=False, combine=True), g[f'link_{tag}_xy1'], atol=1e-9)
    m0 = Mesh(g23['v0'], g23['t0'], uid=0)
    m0.apply_translation((2.0, -1.0), const.MESH_GEAR_FIXED)
    m0.lock()
    m1 = mesh1()
    seen = []

    def scripted(mesh0, mesh1_, ld0, ld1, bboxes0, bboxes1, **kw):
        k = len(seen)
        seen.append(dict(bboxes0=np.array(bboxes0), rwt=float(kw.get('render_weight_threshold', -1)),
                         field1=mesh1_.vertices_w_offset(const.MESH_GEAR_MOVING) - mesh1_.vertices_w_offset(const.MESH_GEAR_INITIAL)))
        return script(k, bboxes0, bboxes1, 9.0)
    monkeypatch.setattr(matcher, 'bboxes_mesh_renderer_matcher', scripted)
    monkeypatch.setattr(optimizer.SLM, 'optimize_linear', _exact_optimize_linear)
    monkeypatch.setattr(Mesh, 'stiffness_matrix_local_normalized', _exact_local_stiffness)
    monkeypatch.setattr(optimizer, 'solve', _exact_solve)
    xy0, xy1, wt, _ = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, _NoImage(), _NoImage(), spacings=np.array([400.0, 100.0]), distributor='cartesian_bbox',
                                                             conf_thresh=0.3, residue_len=3.0, residue_mode='huber', compute_strain=False, stiffness_lambda=0.5,
                                                             min_num_blocks=2, render_weight_threshold=0.1)
    assert len(seen) == int(g['nrounds'])
    for k, r in enumerate(seen):
        assert r['rwt'] == float(g[f'r{k}_rwt']) == 0.1                      # the block matcher is told the threshold too (renderer.py:59)
        np.testing.assert_allclose(r['bboxes0'], g[f'r{k}_bboxes0'], atol=1e-6)
        want = g[f'r{k}_field1']
        np.testing.assert_allclose(r['field1'], want, atol=1e-6 * max(1.0, np.abs(want).max()))
    assert xy0.shape == g['xy0'].shape
    np.testing.assert_allclose(xy0, g['xy0'], atol=1e-5); np.testing.assert_allclose(xy1, g['xy1'], atol=1e-5)
    np.testing.assert_allclose(wt, g['weight'], atol=1e-5)
    assert not np.any((xy1[:, 0] > 620) & (xy1[:, 0] < 880))


@pytest.mark.parametrize('case', ['late_silence', 'early_silence', 'apart', 'tiny_motion'])
def test_product_matcher_loop_exits_vs_reference(monkeypatch, case):
    """the ways out of the loop (matcher.py:592-598, 671-679, 719-724, 744-751) against the reference (golden G36): a round without a
    confident block after earlier rounds linked (their links are the result) or before anything was linked (no result, weight 0, default
    strain), meshes that do not overlap (no round at all), and matches below 0.1 px (linked, never relaxed: no solve)"""
    import json
    from test_oracle_golden import _g23_scripted_block_matches as script
    g = load_golden('g36_loop_exits.npz')
    g23 = load_golden('g23_matcher_loop.npz')
    cs = json.loads(str(g['cases']))[case]
    m0 = Mesh(g23['v0'], g23['t0'], uid=0)
    m0.lock()
    m1 = Mesh(g23['v1'].copy() + np.array(cs.get('shift1', (0.0, 0.0))), g23['t1'], uid=1)
    rounds, solves = [], []

    def scripted(mesh0, mesh1, ld0, ld1, bboxes0, bboxes1, **kw):
        k = len(rounds)
        rounds.append(np.array(bboxes0))
        xy0, xy1, conf = script(k, bboxes0, bboxes1, 11.0)
        if 'amp' in cs:
            xy1 = xy0 + (xy1 - xy0) * (cs['amp'] / 6.0)
        if k >= cs['silent_from']:
            conf = conf * 0.1
        return xy0, xy1, conf

    def counted(self, **kw):
        solves.append(1)
        return _exact_optimize_linear(self, **kw)
    monkeypatch.setattr(matcher, 'bboxes_mesh_renderer_matcher', scripted)
    monkeypatch.setattr(optimizer.SLM, 'optimize_linear', counted)
    monkeypatch.setattr(Mesh, 'stiffness_matrix_local_normalized', _exact_local_stiffness)
    monkeypatch.setattr(optimizer, 'solve', _exact_solve)
    xy0, xy1, wt, strain = matcher.iterative_xcorr_matcher_w_mesh(m0, m1, _NoImage(), _NoImage(), spacings=np.array(cs['spacings']), distributor='cartesian_bbox',
                                                                  conf_thresh=0.3, residue_len=3.0, residue_mode='huber', compute_strain=False,
                                                                  stiffness_lambda=0.5, min_num_blocks=2)
    assert len(rounds) == int(g[f'{case}_nrounds']) and len(solves) == int(g[f'{case}_nsolves'])
    assert (xy0 is None) == bool(g[f'{case}_none']) and strain == float(g[f'{case}_strain'])
    want = g[f'{case}_field1_final']
    np.testing.assert_allclose(m1.vertices_w_offset(const.MESH_GEAR_MOVING) - m1.vertices_w_offset(const.MESH_GEAR_INITIAL), want, atol=1e-6 * max(1.0, np.abs(want).max()))
    if xy0 is None:
        assert xy1 is None and wt == float(g[f'{case}_wt'])
    else:
        assert xy0.shape == g[f'{case}_xy0'].shape
        np.testing.assert_allclose(xy0, g[f'{case}_xy0'], atol=1e-5); np.testing.assert_allclose(xy1, g[f'{case}_xy1'], atol=1e-5)
        np.testing.assert_allclose(wt, g[f'{case}_weight'], atol=1e-5)


@pytest.mark.parametrize('tag,kw', [('defaults', {}), ('given', dict(spacings=[200, 50], conf_thresh=0.4, sigma=3.5, stiffness_multiplier_threshold=0.0, compute_strain=True,
                                                                      residue_len=-2, shrink_factor=0.7, distributor='cartesian_bbox', stiffness_lambda=0.25))])
def test_product_section_matcher_call_vs_reference(monkeypatch, tag, kw):
    """section_matcher down to its call of the loop (matcher.py:370-396) against the reference (golden G37): the defaults it fills in, what
    it passes through, and the sub-meshes it hands over after dropping the triangles of materials softer than
    stiffness_multiplier_threshold (a 'jelly' region of multiplier 0.05 against the default threshold 0.1) -- the loop replaced by a recorder
    on both sides.  The product's one addition is `relax_tol: None` (its loop's default would converge the relaxations)."""
    import json
    g = load_golden('g37_section_matcher_call.npz')
    g23 = load_golden('g23_matcher_loop.npz')
    seen = {}

    def recorder(mesh0, mesh1, ld0, ld1, **k_):
        seen['kw'] = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in k_.items()}
        seen['ntri'] = [int(mesh0.num_triangles), int(mesh1.num_triangles)]
        return np.zeros((1, 2)), np.ones((1, 2)), np.ones(1), 0.07
    monkeypatch.setattr(matcher, 'iterative_xcorr_matcher_w_mesh', recorder)
    m0 = Mesh(g23['v0'], g23['t0'], uid=0)
    nt = g['t1'].shape[0]
    m1 = Mesh(g23['v1'].copy(), g['t1'], uid=1, material_ids=g['mids'], material_names={'default': 0, 'jelly': 5},
              tri_model=np.zeros(nt, dtype=np.int32), tri_matmult=np.where(g['mids'] == 5, 0.05, 1.0))
    res = matcher.section_matcher(m0, m1, _NoImage(), _NoImage(), **kw)
    want = json.loads(str(g[f'{tag}_kw']))
    got = dict(seen['kw'])
    assert got.pop('relax_tol', None) is None
    assert json.loads(json.dumps(got, sort_keys=True)) == want
    assert seen['ntri'] == g[f'{tag}_ntri'].tolist()
    assert res[3] == float(g[f'{tag}_strain'])
