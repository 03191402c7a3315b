"""HDF5 wire formats (feabas_amd/h5wire.py): the layouts of stitcher.py:126-222, aligner.py:26-44 / 134-141 and
mesh.py:543-580 / 798-857 written through libhdf5 and read back -- by the module itself, by the reference's reader
arithmetic restated here, and by the h5dump tool of the HDF5 distribution as an independent decoder."""
import json
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from feabas_amd import constant as const
from feabas_amd import h5wire
from feabas_amd.mesh import Mesh


def _h5dump():
    lib = h5wire.library()
    cand = os.path.join(os.path.dirname(os.path.dirname(lib.path)), 'bin', 'h5dump')
    return cand if os.path.exists(cand) else shutil.which('h5dump')


def _dump(args):
    tool = _h5dump()
    if tool is None:
        pytest.skip('no h5dump beside libhdf5')
    return subprocess.run([tool] + args, capture_output=True, text=True, check=True).stdout


def _matches(rng, pairs):
    m, s = {}, {}
    for k, uid in enumerate(pairs):
        n = 3 + 5 * k
        m[uid] = (rng.random((n, 2)) * 100, rng.random((n, 2)) * 100, rng.random(n))
        s[uid] = 0.01 * (k + 1)
    return m, s


def test_library_and_types(tmp_path):
    lib = h5wire.library()
    assert lib.version >= (1, 10, 0) and lib.has_gzip
    fn = str(tmp_path / 't.h5')
    data = {'u8': np.arange(5, dtype=np.uint8), 'i16': np.array([[-3, 4]], dtype=np.int16), 'i32': np.arange(6, dtype=np.int32).reshape(2, 3),
            'i64': np.array([2 ** 40, -1]), 'f32': np.linspace(0, 1, 7, dtype=np.float32), 'f64': np.pi, 'flag': True,
            'flags': np.array([True, False, True]), 'count': 7, 'empty': np.empty((0, 2)), 'be': np.arange(4, dtype='>f4')}
    with h5wire.H5File(fn, 'w') as f:
        for k, v in data.items():
            f.create_dataset('grp/sub/' + k, v, compression='gzip' if np.ndim(v) else None)
        with pytest.raises(h5wire.H5Error):
            f.create_dataset('grp/sub/u8', data['u8'])                 # exists
        with pytest.raises(TypeError):
            f.create_dataset('c', np.zeros(2, dtype=np.complex64))
    with h5wire.H5File(fn) as f:
        assert 'grp/sub/u8' in f and 'grp/nope/u8' not in f and 'grp' in f and f.is_group('grp/sub')
        assert f.keys('grp/sub') == sorted(data)
        for k, v in data.items():
            got = f['grp/sub/' + k]
            want = np.asarray(v)
            assert np.shape(got) == want.shape, k
            assert np.asarray(got).dtype == want.dtype.newbyteorder('='), k
            np.testing.assert_array_equal(got, want)
        assert f.storage('grp/sub/i32') == (True, 1) and f.storage('grp/sub/f64') == (False, 0)
        assert f.storage('grp/sub/empty') == (False, 0)
        with pytest.raises(KeyError):
            f['grp/sub/none']
    with pytest.raises(h5wire.H5Error):
        h5wire.H5File(str(tmp_path / 'missing.h5'))
    head = _dump(['-H', fn])
    assert 'H5T_ENUM' in head and '"FALSE"' in head and 'H5T_STD_I16LE' in head and 'H5T_IEEE_F32LE' in head


def test_stitcher_file_round_trip_and_independent_decode(tmp_path):
    rng = np.random.default_rng(3)
    names = [f'row{k // 4}/tile{k % 4}.png' for k in range(12)]
    bboxes = rng.integers(0, 5000, (12, 4))
    m, s = _matches(rng, [(1, 0), (5, 4), (10, 2), (11, 10)])
    bc = {(1, 0): np.array([1.0, 2.0, 3.0, 4.0]), (5, 4): np.array([0.5, 0.25, 8.0, 9.0])}
    fn = str(tmp_path / 'sec.h5')
    h5wire.save_stitcher_h5(fn, '/data/sec001', 4.0, names, bboxes, m, s, bc)
    m2, s2, bc2 = h5wire.load_stitcher_matches(fn)
    assert list(m2) == sorted(m, key=lambda u: f'{u[0]}_{u[1]}')             # name order, like iterating the h5py group
    for uid in m:
        for a, b in zip(m[uid], m2[uid]):
            np.testing.assert_array_equal(np.asarray(a, np.float32), b)
        assert s2[uid] == np.float32(s[uid])
        if uid in bc:
            np.testing.assert_array_equal(bc[uid], bc2[uid])
    with h5wire.H5File(fn) as f:
        assert h5wire.numpy_to_str_ascii(f['imgrootdir']) == '/data/sec001' and f['resolution'] == 4.0
        assert h5wire.numpy_to_str_ascii(f['imgrelpaths']).split('\n') == names
        np.testing.assert_array_equal(f['init_bboxes'], bboxes)
        assert f.storage('matches/1_0') == (True, 1) and f.storage('imgrootdir') == (False, 0)
    # independent decoder: the raw record of one pair, split with the reader arithmetic of stitcher.py:200-207
    txt = _dump(['-d', '/matches/10_2', '-y', '-w', '0', '-m', '%.9g', fn])
    body = txt[txt.index('DATA {') + 6:txt.rindex('}')]
    vals = np.array([float(v) for v in re.findall(r'[-+0-9.eE]+', body.split('}')[0])], dtype=np.float32)
    npt = int((vals.size - 1) / 5)
    xy0, xy1, w = m[(10, 2)]
    assert npt == len(w)
    np.testing.assert_allclose(vals[:2 * npt].reshape(-1, 2), xy0, rtol=2e-6)
    np.testing.assert_allclose(vals[2 * npt:4 * npt].reshape(-1, 2), xy1, rtol=2e-6)
    np.testing.assert_allclose(vals[4 * npt:5 * npt], w, rtol=2e-6)
    assert abs(vals[-1] - s[(10, 2)]) < 1e-7
    head = _dump(['-H', '-p', fn])
    assert 'COMPRESSION DEFLATE { LEVEL 4 }' in head and 'GROUP "matches"' in head


def test_stitcher_file_check_order(tmp_path):
    rng = np.random.default_rng(4)
    names = ['a.png', 'b.png', 'c.png', 'd.png']
    m, s = _matches(rng, [(1, 0), (2, 1), (3, 2)])
    fn = str(tmp_path / 'sec.h5')
    h5wire.save_stitcher_h5(fn, '', 8.0, names, np.zeros((4, 4), dtype=np.int64), m, s, {(2, 1): np.ones(4)}, compression=False)
    mine = ['d.png', 'c.png', 'b.png']                       # another order, one tile unknown to the caller
    m2, s2, bc2 = h5wire.load_stitcher_matches(fn, imgrelpaths=mine)
    assert set(m2) == {(1, 2), (0, 1)} and set(bc2) == {(1, 2)}
    np.testing.assert_array_equal(m2[(1, 2)][2], np.asarray(m[(2, 1)][2], np.float32))
    with h5wire.H5File(fn) as f:
        assert f.storage('matches/1_0') == (False, 0)
    empty = str(tmp_path / 'none.h5')
    h5wire.save_stitcher_h5(empty, '', 8.0, names, np.zeros((4, 4), dtype=np.int64))
    assert h5wire.load_stitcher_matches(empty) == ({}, {}, {})


def test_section_match_file(tmp_path):
    rng = np.random.default_rng(5)
    xy0, xy1, w = rng.random((40, 2)) * 1000, rng.random((40, 2)) * 1000, rng.random(40)
    fn = str(tmp_path / 's0_s1.h5')
    h5wire.save_section_match_h5(fn, xy0, xy1, w, 16.0, 0.03, 'sec0', 'sec1')
    a, b, c, strain = h5wire.read_matches_from_h5(fn)
    np.testing.assert_array_equal(a, xy0); np.testing.assert_array_equal(b, xy1); np.testing.assert_array_equal(c, w)
    assert strain == 0.03
    a4, b4, _, _ = h5wire.read_matches_from_h5(fn, target_resolution=4.0)                # spatial.py:77-86
    np.testing.assert_allclose(a4, (xy0 + 0.5) * 4 - 0.5)
    np.testing.assert_allclose(b4, (xy1 + 0.5) * 4 - 0.5)
    with h5wire.H5File(fn) as f:
        assert f.keys() == ['name0', 'name1', 'resolution', 'strain', 'weight', 'xy0', 'xy1']
        assert h5wire.numpy_to_str_ascii(f['name1']) == 'sec1' and np.ndim(f['strain']) == 0


def test_mesh_file_round_trip(tmp_path):
    rng = np.random.default_rng(6)
    M = Mesh.from_bbox((0, 0, 200, 120), cartesian=True, mesh_size=25, resolution=8.0, soft_factor=0.5, uid=17)
    nt = M.num_triangles
    M = Mesh(M.vertices(const.MESH_GEAR_INITIAL), M.triangles, resolution=8.0, soft_factor=0.5, uid=17, locked=True,
             initial_offset=np.array([[10.0, -4.0]]), stiffness_multiplier=rng.random(nt) + 0.5,
             tri_model=(np.arange(nt) % 3 == 0).astype(np.int32) * 2, tri_nu=np.where(np.arange(nt) % 3 == 0, 0.3, 0.0),
             tri_matmult=np.where(np.arange(nt) % 3 == 0, 0.1, 1.0))
    M.locked = False
    M.set_vertices(M.vertices(const.MESH_GEAR_INITIAL) + rng.normal(0, 1, (M.num_vertices, 2)), const.MESH_GEAR_MOVING)
    M.set_offset(np.array([[3.0, 5.0]]), const.MESH_GEAR_MOVING)
    M.locked = True
    M.name = 'sec0_tile3'
    fn = str(tmp_path / 'mesh.h5')
    M.save_to_h5(fn)
    with h5wire.H5File(fn) as f:
        keys = f.keys()
        assert {'vertices', 'triangles', 'initial_offset', 'moving_vertices', 'moving_offset', 'stiffness_multiplier', 'material_ids',
                'material_table', 'resolution', 'epsilon', 'name', 'locked', 'uid', 'soft_factor'} == set(keys)
        assert 'fixed_vertices' not in keys                 # the fixed gear aliases the initial one: not saved, mesh.py:553-559
        table = json.loads(h5wire.numpy_to_str_ascii(f['material_table']))
        assert table['default']['uid'] == 0 and table['default']['type'] == 'MATERIAL_MODEL_ENG'
        assert sorted(m['type'] for m in table.values()) == ['MATERIAL_MODEL_ENG', 'MATERIAL_MODEL_NHK']
        assert f['locked'] is np.True_ or f['locked'] == True           # noqa: E712
        assert f.storage('vertices') == (True, 1) and f.storage('uid') == (False, 0)
    N = Mesh.from_h5(fn)
    for g in (const.MESH_GEAR_INITIAL, const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING):
        np.testing.assert_array_equal(N.vertices(g), M.vertices(g))
        np.testing.assert_array_equal(N.offset(g), M.offset(g))
    np.testing.assert_array_equal(N.triangles, M.triangles)
    np.testing.assert_array_equal(N.tri_model, M.tri_model)
    np.testing.assert_array_equal(N.tri_nu, M.tri_nu)
    np.testing.assert_array_equal(N.tri_matmult, M.tri_matmult)
    np.testing.assert_array_equal(N._stiffness_multiplier, M._stiffness_multiplier)
    assert (N.resolution, N.soft_factor, N.uid, N.locked, N.name) == (8.0, 0.5, 17.0, True, 'sec0_tile3')
    # several meshes under prefixes of one file, as the Stitcher file keeps them (stitcher.py:158-163)
    fn2 = str(tmp_path / 'many.h5')
    plain = Mesh.from_bbox((0, 0, 50, 50), cartesian=True, mesh_size=25)
    with h5wire.H5File(fn2, 'w') as f:
        plain.save_to_h5(f, vertex_flags=(const.MESH_GEAR_INITIAL,), prefix='master_meshes/0', save_material=False)
        M.save_to_h5(f, prefix='master_meshes/1')
    with h5wire.H5File(fn2) as f:
        assert f.keys('master_meshes') == ['0', '1'] and 'master_meshes/0/material_ids' not in f
        P = Mesh.from_h5(f, prefix='master_meshes/0')
    assert P.tri_model is None and P.num_triangles == plain.num_triangles


def test_mesh_file_keeps_the_stiffness_functions(tmp_path):
    """a material with a stiffness function (the default "wrinkle" material, default_material_table.yaml:46-56) goes into the
    mesh file the way Material.to_dict writes it (material.py:106-113: factory name + parameters) and comes back as the same
    table on the same triangles; a mesh with one is not linear"""
    from feabas_amd import material
    M0 = Mesh.from_bbox((0, 0, 200, 120), cartesian=True, mesh_size=25)
    nt = M0.num_triangles
    func = np.where(np.arange(nt) % 4 == 0, 0, np.where(np.arange(nt) % 4 == 1, 1, -1)).astype(np.int32)
    wr = material.asymmetrical_elasticity(strain=[0.0, 0.75, 1.0, 1.01], stiffness=[1.5, 1.0, 0.5, 1.0e-7])
    fold = material.asymmetrical_elasticity(strain=[0.2, 0.9, 1.3], stiffness=[3.0, 1.2, 0.25])
    M = Mesh(M0.vertices(const.MESH_GEAR_INITIAL), M0.triangles, uid=5,
             tri_model=np.where(func == 1, 1, 0).astype(np.int32), tri_nu=np.where(func == 1, 0.25, 0.0),
             tri_matmult=np.where(func == 0, 0.4, np.where(func == 1, 1.3, 1.0)), tri_func=func, stiffness_funcs=[wr, fold], func_matmult=[0.4, 1.3])
    assert not M.is_linear and M.linear_triangle_mask.sum() == (func < 0).sum()
    assert float(wr(0.5)) == pytest.approx(1.5 - 0.5 * 0.5 / 0.75) and float(wr(2.0)) == 1.0e-7 and float(wr(-1.0)) == 1.5
    fn = str(tmp_path / 'mesh.h5')
    M.save_to_h5(fn)
    with h5wire.H5File(fn) as f:
        table = json.loads(h5wire.numpy_to_str_ascii(f['material_table']))
    with_f = [m for m in table.values() if m.get('stiffness_func_factory')]
    assert len(with_f) == 2 and all(m['stiffness_func_factory'] == 'feabas.material.asymmetrical_elasticity' for m in with_f)
    assert sorted(m['stiffness_multiplier'] for m in with_f) == [0.4, 1.3]
    N = Mesh.from_h5(fn)
    assert not N.is_linear
    for k in (0, 1):
        sel = func == k
        assert np.all(N.tri_func[sel] == N.tri_func[sel][0]) and np.all(N.tri_func[~sel] != N.tri_func[sel][0])
        assert N.stiffness_funcs[N.tri_func[sel][0]] == (wr, fold)[k]
        assert N.func_matmult[N.tri_func[sel][0]] == (0.4, 1.3)[k]
    np.testing.assert_array_equal(N.tri_model, M.tri_model)
    # the effective multiplier of mesh.py:1600-1621 (median area stretch of the linear triangles as the base)
    N.set_vertices(N.vertices(const.MESH_GEAR_INITIAL) * np.array([[0.9, 1.0]]), const.MESH_GEAR_MOVING)
    eff = N.effective_stiffness_multiplier()
    assert np.allclose(eff[func < 0], 1.0) and np.allclose(eff[func == 0], 0.4 * 0.5)        # uniform compression: stretch / median = 1
    # any other callable is taken as it is (evaluated on the host, Mesh.assemble_into) -- a lambda string, a factory with parameters --
    # but a mesh file can only carry tables, and dill-serialised lambdas need dill
    f = material.stiffness_func_from_spec('lambda x: 2.0 * x')
    assert not isinstance(f, material.StiffnessTable) and f(np.array([0.5]))[0] == 1.0
    g = material.stiffness_func_from_spec(lambda **kw: (lambda x: kw['a'] * x), dict(a=3.0))
    assert g(2.0) == 6.0
    with pytest.raises(NotImplementedError):
        material.stiffness_func_from_spec('<lambda_bytes>00')
    P = Mesh(M.vertices(const.MESH_GEAR_INITIAL), M.triangles, uid=4, tri_func=func, stiffness_funcs=[f, g], func_matmult=[0.4, 1.3])
    with pytest.raises(NotImplementedError):
        P.save_to_h5(str(tmp_path / 'callable.h5'))


def test_mesh_file_keeps_the_named_materials(tmp_path):
    """the names, uids and area constraints of a mesh's material table survive save -> load (the stages meshing ->
    matching -> optimisation hand meshes over through files): ``triangles_of_material(name)`` selects the same triangles
    afterwards, two named materials with the SAME constitutive parameters stay two materials, and a second round trip
    changes nothing"""
    M0 = Mesh.from_bbox((0, 0, 300, 200), cartesian=True, mesh_size=25)
    nt = M0.num_triangles
    c = M0.vertices(const.MESH_GEAR_INITIAL)[M0.triangles].mean(axis=1)
    ids = np.zeros(nt, dtype=np.int32)
    ids[c[:, 0] < 80] = 3                                     # 'wrinkle': soft, refined
    ids[c[:, 0] > 220] = 7                                    # 'refine': same parameters as default, but a refinement region
    ids[(c[:, 1] > 150) & (ids == 0)] = 50                    # 'hold': stiff
    names = {'default': 0, 'wrinkle': 3, 'refine': 7, 'hold': 50, 'absent': 9}
    constraints = {'default': 1.0, 'wrinkle': 0.25, 'refine': 0.5, 'hold': 1.0, 'absent': 0.1}
    M = Mesh(M0.vertices(const.MESH_GEAR_INITIAL), M0.triangles, uid=2,
             tri_model=np.zeros(nt, dtype=np.int32), tri_nu=np.where(ids == 50, 0.2, 0.0),
             tri_matmult=np.where(ids == 3, 0.1, np.where(ids == 50, 8.0, 1.0)),
             material_ids=ids, material_names=names, material_area_constraints=constraints)
    fn = str(tmp_path / 'named.h5')
    M.save_to_h5(fn)
    with h5wire.H5File(fn) as f:
        table = json.loads(h5wire.numpy_to_str_ascii(f['material_table']))
        np.testing.assert_array_equal(np.asarray(f['material_ids']).ravel(), ids)
    assert {n: m['uid'] for n, m in table.items()} == names
    assert {n: m['area_constraint'] for n, m in table.items()} == constraints
    assert table['wrinkle']['stiffness_multiplier'] == pytest.approx(0.1, rel=1e-6) and table['hold']['poisson_ratio'] == 0.2      # (multipliers are float32 in the mesh)
    assert table['refine']['stiffness_multiplier'] == table['default']['stiffness_multiplier'] == 1.0
    N = Mesh.from_h5(fn)
    for name in names:
        np.testing.assert_array_equal(N.triangles_of_material(name), M.triangles_of_material(name))
    assert N.triangles_of_material('refine').sum() > 0 and N.triangles_of_material('absent').sum() == 0
    assert N.material_area_constraints == constraints and N.material_names == names
    np.testing.assert_array_equal(N.tri_matmult, M.tri_matmult.astype(np.float32))
    np.testing.assert_array_equal(N.tri_nu, M.tri_nu)
    fn2 = str(tmp_path / 'named2.h5')
    N.save_to_h5(fn2)
    with h5wire.H5File(fn2) as f:
        assert json.loads(h5wire.numpy_to_str_ascii(f['material_table'])) == table
    # triangles of one uid that disagree in their parameters cannot be written under one name: synthesized entries instead
    M.tri_matmult = M.tri_matmult.copy(); M.tri_matmult[np.flatnonzero(ids == 50)[0]] = 2.0
    fn3 = str(tmp_path / 'named3.h5')
    M.save_to_h5(fn3)
    with h5wire.H5File(fn3) as f:
        t3 = json.loads(h5wire.numpy_to_str_ascii(f['material_table']))
    assert 'hold' not in t3 and 'default' in t3
    np.testing.assert_array_equal(Mesh.from_h5(fn3).tri_matmult, M.tri_matmult.astype(np.float32))


def test_mesh_file_keeps_the_render_weights(tmp_path):
    """the render weight and the render flag of every named material (material.py:27-30, 50-54; `render: false` is carried as the
    negative weight -(render_weight + 1), so that a RENDERED material of weight 0 stays rendered at threshold 0, mesh.py:1850-1854) survive save -> load: the render
    masks by threshold and the per-triangle weights -- what decides where blocks are placed, what is rendered and where matches may land
    (mesh.py:1836-1859, 2168-2170) -- are the same afterwards"""
    M0 = Mesh.from_bbox((0, 0, 300, 200), cartesian=True, mesh_size=25)
    c = M0.vertices(const.MESH_GEAR_INITIAL)[M0.triangles].mean(axis=1)
    ids = np.zeros(M0.num_triangles, dtype=np.int32)
    ids[c[:, 0] < 80] = 3
    ids[c[:, 0] > 220] = 7
    ids[(c[:, 0] > 120) & (c[:, 0] < 160)] = 9
    names = {'default': 0, 'soft': 3, 'hidden': 7, 'weightless': 9}
    M = Mesh(M0.vertices(const.MESH_GEAR_INITIAL), M0.triangles, uid=2, material_ids=ids, material_names=names,
             material_render_weights={'soft': 1.0e-6, 'hidden': -(0.25 + 1.0), 'weightless': 0.0})
    fn = str(tmp_path / 'weights.h5')
    M.save_to_h5(fn)
    with h5wire.H5File(fn) as f:
        table = json.loads(h5wire.numpy_to_str_ascii(f['material_table']))
    assert table['soft']['render'] is True and table['soft']['render_weight'] == pytest.approx(1.0e-6)
    assert table['hidden']['render'] is False and table['hidden']['render_weight'] == pytest.approx(0.25)
    assert table['weightless']['render'] is True and table['weightless']['render_weight'] == 0.0
    assert table['default']['render_weight'] == 1.0
    N = Mesh.from_h5(fn)
    np.testing.assert_array_equal(N.weight_multiplier_for_render(), M.weight_multiplier_for_render())
    for thr in (0.0, 0.1, 1.0e-6):
        np.testing.assert_array_equal(N.triangle_mask_for_render(render_weight_threshold=thr), M.triangle_mask_for_render(render_weight_threshold=thr))
    assert N.triangle_mask_for_render(render_weight_threshold=0.1).sum() == (ids == 0).sum()
    assert N.triangle_mask_for_render().sum() == (ids != 7).sum()             # weight 0 but render = True: rendered at threshold 0
    assert N.weight_multiplier_for_render().min() == 0.0
    # a mesh without weighted materials writes and reads plain weights
    M0.save_to_h5(str(tmp_path / 'plain.h5'))
    assert Mesh.from_h5(str(tmp_path / 'plain.h5')).tri_render_weight is None
