"""CPU-side checks of the product package: the C-ABI library loads and exports
every symbol include/feabas_hip.h declares (no compute calls), and the host
logic (bbox helpers, next_fast_len) matches the golden vectors."""
import ctypes

import numpy as np
import pytest

from conftest import load_golden
import feabas_amd
from feabas_amd import _lib, common, matcher


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), n
    # and every prototype we bind is declared in the header
    assert set(_lib._PROTOS) <= set(names)


def test_no_cpu_fallback_without_gpu():
    lib = _lib.load()
    if lib.fb_create(0):
        pytest.skip('a GPU is visible')
    with pytest.raises(RuntimeError):
        matcher.xcorr_fft(np.zeros((1, 8, 8), np.float32), np.zeros((1, 8, 8), np.float32))


def test_product_does_not_import_oracle():
    import os, re
    root = os.path.dirname(os.path.abspath(feabas_amd.__file__))
    for fn in os.listdir(root):
        if fn.endswith('.py'):
            src = open(os.path.join(root, fn)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', src, flags=re.M), fn


def test_next_fast_len():
    g = load_golden('g11_bbox.npz')
    got = np.array([matcher.next_fast_len(n) for n in range(1, 4200)])
    np.testing.assert_array_equal(got, g['nfl'])


def test_bbox_helpers_golden():
    g = load_golden('g11_bbox.npz')
    k = 0
    while f'div{k}_in' in g:
        p = g[f'div{k}_in']
        mnb = tuple(int(v) for v in p[6:])
        kw = dict(min_num_blocks=mnb[0] if len(mnb) == 1 else mnb, shrink_factor=p[5])
        if p[4] > 0:
            kw['block_size'] = p[4]
        res = np.stack(common.divide_bbox(tuple(p[:4]), **kw), axis=-1)
        np.testing.assert_array_equal(res, g[f'div{k}_out'])
        k += 1
    np.testing.assert_array_equal(common.z_order(g['z_in']), g['z_out'])
    np.testing.assert_allclose(common.bbox_centers(g['bb_in']), g['bb_centers'])
    np.testing.assert_allclose(common.bbox_sizes(g['bb_in']), g['bb_sizes'])

    class M:
        def __init__(self, bb): self.bb = bb
        def bbox(self, gear=None): return self.bb
    for k in range(2):
        p = g[f'dist{k}_in']
        r0, r1 = matcher.distributor_cartesian_bbox(M(p[:4]), M(p[4:8]), p[8], min_num_blocks=int(p[9]), zorder=True)
        np.testing.assert_array_equal(r0, g[f'dist{k}_bb0'])
        np.testing.assert_array_equal(r1, g[f'dist{k}_bb1'])


def test_mesh_gears_and_field_semantics():
    from feabas_amd.mesh import Mesh
    from feabas_amd import constant as const
    from oracle import fem_ref
    v, t = fem_ref.grid_mesh(5, 4, 10.0)
    m = Mesh(v, t, uid=3)
    r = fem_ref.RefMesh(v, t, uid=3)
    m.apply_translation((1.5, -2.0), const.MESH_GEAR_FIXED)
    r.apply_translation((1.5, -2.0), fem_ref.GEAR_FIXED)
    d = np.random.default_rng(0).standard_normal(v.shape)
    m.set_field(d, gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_MOVING))
    r.set_field(d, gear=(fem_ref.GEAR_MOVING, fem_ref.GEAR_MOVING))
    for gear in (-1, 0, 1, 2):
        np.testing.assert_allclose(m.vertices(gear), r.vertices(gear))
        np.testing.assert_allclose(m.offset(gear), r.offset(gear))
    m.anneal(gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_FIXED), mode=const.ANNEAL_COPY_EXACT)
    r.anneal_copy((fem_ref.GEAR_MOVING, fem_ref.GEAR_FIXED))
    np.testing.assert_allclose(m.vertices_w_offset(0), r.vertices_w_offset(0))
    tid = np.array([0, 5, 11])
    xy = r.bary2cart(tid, np.array([[0.2, 0.3, 0.5]] * 3), 1)
    tid2, B = m.cart2bary(xy, 1, tid=None)
    np.testing.assert_array_equal(tid2, tid)
    np.testing.assert_allclose(B, [[0.2, 0.3, 0.5]] * 3, atol=1e-12)


def test_translation_optimisers_vs_reference_golden():
    """SLM.optimize_translation_lsqr / optimize_translation_w_filtering (optimizer.py:974-1125) are host code in the reference and
    here (scipy lsqr on #tiles unknowns): golden G15 from the reference, no GPU involved"""
    from conftest import load_golden
    import feabas_amd
    from feabas_amd import constant as const
    from feabas_amd.mesh import Mesh
    from feabas_amd.optimizer import Link, SLM
    g = load_golden('g15_translation.npz')

    def system():
        ms = [Mesh(g['v'], g['t'], uid=k) for k in range(6)]
        ms[0].lock()
        links = []
        for k in range(7):
            a, b = g[f'l{k}_ab']
            links.append(Link(ms[a], ms[b], g[f'l{k}_tid0'], g[f'l{k}_tid1'], g[f'l{k}_B0'], g[f'l{k}_B1'], weight=g[f'l{k}_w']))
        return ms, links, SLM(ms, links=links)
    ms, links, slm = system()
    cost, residue = slm.optimize_translation_lsqr(tol=1e-12)
    np.testing.assert_allclose(cost, g['lsqr_cost'], rtol=1e-9)
    np.testing.assert_allclose(residue, g['lsqr_residue'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.stack([m.offset(const.MESH_GEAR_FIXED).ravel() for m in ms]), g['lsqr_offsets'], atol=1e-8)
    ms, links, slm = system()
    nd, cost2 = slm.optimize_translation_w_filtering(tol=1e-12, residue_threshold=1.0)
    assert nd == int(g['filt_disabled'])
    np.testing.assert_array_equal(np.array([lk._disabled for lk in links]), g['filt_link_disabled'])
    np.testing.assert_allclose(cost2[0], g['filt_cost'][0], rtol=1e-9)
    assert cost2[1] < 1e-9
    np.testing.assert_allclose(np.stack([m.offset(const.MESH_GEAR_FIXED).ravel() for m in ms]), g['filt_offsets'], atol=1e-8)


@pytest.mark.parametrize('name,mode', [('grigid', 0), ('gaffine', 1), ('crigid', 2), ('caffine', 3)])
def test_g16_anneal_modes(name, mode):
    """Mesh.anneal rigid / affine, whole mesh and per connected component (mesh.py:2421-2451), and the region
    relax_mesh_most_deformed frees (optimizer.py:2157-2188) -- host logic of the product against the reference"""
    from conftest import load_golden
    from feabas_amd import constant as const
    from feabas_amd.mesh import Mesh
    g = load_golden('g16_relax.npz')
    m = Mesh(g['an_v'], g['an_t'], moving_vertices=g['an_vmov'].copy(), moving_offset=np.array([[1.0, 2.0]]), uid=4)
    m.anneal(gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_FIXED), mode=mode)
    np.testing.assert_allclose(m.vertices(const.MESH_GEAR_FIXED), g[f'an_{name}_vfix'], atol=1e-9)
    np.testing.assert_allclose(m.offset(const.MESH_GEAR_FIXED), g[f'an_{name}_foff'], atol=1e-9)


def test_g16_masked_field_and_measures():
    from conftest import load_golden
    from feabas_amd import constant as const
    from feabas_amd.mesh import Mesh
    g = load_golden('g16_relax.npz')
    gear = (const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)
    m = Mesh(g['v'], g['t'], stiffness_multiplier=g['mult'], moving_vertices=g['vmov'].copy(), moving_offset=g['moff'].copy(), uid=3)
    np.testing.assert_allclose(m.triangle_area_deform(gear), g['area_deform'], rtol=1e-12)
    np.testing.assert_allclose(m.triangle_edge_deform(gear), g['edge_deform'], rtol=1e-12)
    np.testing.assert_allclose(Mesh.svds_to_deform(g['area_deform'].reshape(-1, 1)), g['svd_deform_area'], rtol=1e-12)
    np.testing.assert_allclose(m.effective_stiffness_multiplier(), g['eff_mult'], rtol=1e-7)
    vm = np.zeros(g['v'].shape[0], dtype=bool)
    vm[g['free_vtx']] = True
    d = np.arange(2 * vm.sum(), dtype=np.float64).reshape(-1, 2)
    m.apply_field(d, gear[1], vtx_mask=vm)                  # mesh.py:2393-2396: only the masked vertices move, the offset stays
    np.testing.assert_array_equal(m.vertices(gear[1])[vm], g['vmov'][vm] + d)
    np.testing.assert_array_equal(m.vertices(gear[1])[~vm], g['vmov'][~vm])
    np.testing.assert_array_equal(m.offset(gear[1]), g['moff'])
