"""HDF5 wire formats on either side of the hot paths, so that a GPU rank writes what the unchanged
``--mode optimization`` / ``rendering`` stages of the reference read, and reads what its ``--mode matching`` wrote:

* the Stitcher file (stitcher.py:126-181 writer, 184-222 reader): ``imgrootdir``, ``resolution``, ``imgrelpaths``,
  ``init_bboxes``, ``matches/<i>_<j>`` = float32 concat(xy0, xy1, weight, strain), brightness / contrast tables;
* the section-pair match file of the aligner (aligner.py:134-141 writer, 26-44 reader): ``xy0 xy1 weight resolution
  strain name0 name1``;
* the Mesh file (mesh.py:543-580 ``get_init_dict``, 822-857 ``save_to_h5``, 798-819 ``from_h5``).

The reference goes through h5py, which this image does not have; libhdf5 (the C library h5py itself wraps) is here, so
``H5File`` below binds the dozen C entry points the three layouts need through ctypes.  Conventions follow what
``h5py.File.create_dataset(name, data=..., compression='gzip')`` produces: little-endian standard types, numpy ``bool`` as
the enum (FALSE=0, TRUE=1) over int8, scalars in a scalar dataspace without filters, gzip level 4 on chunked storage,
intermediate groups created on demand, keys listed in name order.  No library, no file: ``H5Error`` -- there is no other
container format to fall back to.
"""
import ctypes
import ctypes.util
import glob
import json
import os
import threading

import numpy as np

from . import constant as const

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5S_SCALAR = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5I_GROUP, H5I_DATASET = 2, 5
GZIP_LEVEL = 4                        # h5py's default for compression='gzip'
CHUNK_BYTES = 256 << 10
DEFAULT_AVG_DEFORM = 0.05             # feabas/config.py:32


class H5Error(RuntimeError):
    pass


_lock = threading.RLock()             # libhdf5 is not built thread-safe
_h5 = None


def _protos(lib):
    c_char_p, c_int, c_uint, c_void_p, c_size_t = ctypes.c_char_p, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t
    P = ctypes.POINTER
    table = {
        'H5open': (c_int, []),
        'H5get_libversion': (c_int, [P(c_uint)] * 3),
        'H5Eset_auto2': (c_int, [hid_t, c_void_p, c_void_p]),
        'H5Fcreate': (hid_t, [c_char_p, c_uint, hid_t, hid_t]),
        'H5Fopen': (hid_t, [c_char_p, c_uint, hid_t]),
        'H5Fclose': (c_int, [hid_t]),
        'H5Gcreate2': (hid_t, [hid_t, c_char_p, hid_t, hid_t, hid_t]),
        'H5Gclose': (c_int, [hid_t]),
        'H5Gget_info_by_name': (c_int, [hid_t, c_char_p, c_void_p, hid_t]),
        'H5Lexists': (c_int, [hid_t, c_char_p, hid_t]),
        'H5Lget_name_by_idx': (ctypes.c_ssize_t, [hid_t, c_char_p, c_int, c_int, hsize_t, c_char_p, c_size_t, hid_t]),
        'H5Oopen': (hid_t, [hid_t, c_char_p, hid_t]),
        'H5Oclose': (c_int, [hid_t]),
        'H5Iget_type': (c_int, [hid_t]),
        'H5Screate': (hid_t, [c_int]),
        'H5Screate_simple': (hid_t, [c_int, P(hsize_t), P(hsize_t)]),
        'H5Sclose': (c_int, [hid_t]),
        'H5Sget_simple_extent_ndims': (c_int, [hid_t]),
        'H5Sget_simple_extent_dims': (c_int, [hid_t, P(hsize_t), P(hsize_t)]),
        'H5Pcreate': (hid_t, [hid_t]),
        'H5Pclose': (c_int, [hid_t]),
        'H5Pset_chunk': (c_int, [hid_t, c_int, P(hsize_t)]),
        'H5Pset_deflate': (c_int, [hid_t, c_uint]),
        'H5Pget_layout': (c_int, [hid_t]),
        'H5Pget_nfilters': (c_int, [hid_t]),
        'H5Pset_create_intermediate_group': (c_int, [hid_t, c_uint]),
        'H5Zfilter_avail': (c_int, [c_int]),
        'H5Dcreate2': (hid_t, [hid_t, c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        'H5Dopen2': (hid_t, [hid_t, c_char_p, hid_t]),
        'H5Dclose': (c_int, [hid_t]),
        'H5Dget_space': (hid_t, [hid_t]),
        'H5Dget_type': (hid_t, [hid_t]),
        'H5Dget_create_plist': (hid_t, [hid_t]),
        'H5Dwrite': (c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p]),
        'H5Dread': (c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p]),
        'H5Tget_class': (c_int, [hid_t]),
        'H5Tget_size': (c_size_t, [hid_t]),
        'H5Tget_sign': (c_int, [hid_t]),
        'H5Tget_nmembers': (c_int, [hid_t]),
        'H5Tis_variable_str': (c_int, [hid_t]),
        'H5Tcopy': (hid_t, [hid_t]),
        'H5Tclose': (c_int, [hid_t]),
        'H5Tenum_create': (hid_t, [hid_t]),
        'H5Tenum_insert': (c_int, [hid_t, c_char_p, c_void_p]),
    }
    for name, (res, args) in table.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


class _Lib:
    """libhdf5 >= 1.10 (64-bit hid_t).  Search order: $FEABAS_HDF5_LIB, the loader path, the conda runtime of the image."""

    def __init__(self):
        cands = [os.environ.get('FEABAS_HDF5_LIB'), ctypes.util.find_library('hdf5')]
        for pat in ('/opt/conda/lib/libhdf5.so*', '/usr/lib/x86_64-linux-gnu/libhdf5*.so*', '/usr/lib64/libhdf5.so*'):
            cands.extend(sorted(glob.glob(pat)))
        lib, tried = None, []
        for cand in cands:
            if not cand:
                continue
            try:
                lib = ctypes.CDLL(cand)
                if hasattr(lib, 'H5Dcreate2'):
                    self.path = cand
                    break
                lib = None
            except OSError as err:
                tried.append(f'{cand}: {err}')
        if lib is None:
            raise H5Error('libhdf5 not found (set FEABAS_HDF5_LIB to the shared library); tried: ' + '; '.join(tried))
        _protos(lib)
        self.lib = lib
        if lib.H5open() < 0:
            raise H5Error('H5open failed')
        ver = [ctypes.c_uint() for _ in range(3)]
        lib.H5get_libversion(*[ctypes.byref(v) for v in ver])
        self.version = tuple(v.value for v in ver)
        if self.version < (1, 10, 0):
            raise H5Error(f'libhdf5 {self.version} at {self.path}: 1.10 or later is needed (64-bit identifiers)')
        lib.H5Eset_auto2(0, None, None)                # errors come back as return codes; no stack dump on stderr
        g = lambda name: hid_t.in_dll(lib, name).value
        self.std = {np.dtype(k): g(v) for k, v in {
            'u1': 'H5T_STD_U8LE_g', 'i1': 'H5T_STD_I8LE_g', '<u2': 'H5T_STD_U16LE_g', '<i2': 'H5T_STD_I16LE_g',
            '<u4': 'H5T_STD_U32LE_g', '<i4': 'H5T_STD_I32LE_g', '<u8': 'H5T_STD_U64LE_g', '<i8': 'H5T_STD_I64LE_g',
            '<f4': 'H5T_IEEE_F32LE_g', '<f8': 'H5T_IEEE_F64LE_g'}.items()}
        self.dcpl_cls = g('H5P_CLS_DATASET_CREATE_ID_g')
        self.lcpl_cls = g('H5P_CLS_LINK_CREATE_ID_g')
        self.lcpl = lib.H5Pcreate(self.lcpl_cls)
        lib.H5Pset_create_intermediate_group(self.lcpl, 1)
        self.bool_t = lib.H5Tenum_create(self.std[np.dtype('i1')])
        for name, val in ((b'FALSE', 0), (b'TRUE', 1)):
            v = ctypes.c_int8(val)
            lib.H5Tenum_insert(self.bool_t, name, ctypes.byref(v))
        self.has_gzip = lib.H5Zfilter_avail(1) > 0


def library():
    global _h5
    with _lock:
        if _h5 is None:
            _h5 = _Lib()
        return _h5


def _chunk_shape(shape, itemsize):
    """chunks of at most CHUNK_BYTES: halve the longest axis until it fits (readers do not depend on the choice)"""
    chunk = [max(1, int(s)) for s in shape]
    while np.prod(chunk) * itemsize > CHUNK_BYTES:
        k = int(np.argmax(chunk))
        if chunk[k] == 1:
            break
        chunk[k] = (chunk[k] + 1) // 2
    return chunk


class H5File:
    """The slice of ``h5py.File`` the three layouts use: ``create_dataset(name, data=, compression=)``,
    ``create_group``, ``name in f``, ``keys(group)``, ``f[name]`` -> array (what ``f[name][()]`` gives in h5py)."""

    def __init__(self, fname, mode='r'):
        self._h = library()
        self._id = -1
        L = self._h.lib
        name = os.fsencode(fname)
        with _lock:
            if mode == 'r':
                fid = L.H5Fopen(name, H5F_ACC_RDONLY, 0)
            elif mode in ('r+', 'a') and os.path.exists(fname):
                fid = L.H5Fopen(name, H5F_ACC_RDWR, 0)
            elif mode in ('w', 'a'):
                fid = L.H5Fcreate(name, H5F_ACC_TRUNC, 0, 0)
            else:
                raise ValueError(f'mode {mode!r}')
        if fid < 0:
            raise H5Error(f'cannot open {fname!r} with mode {mode!r}')
        self._id = fid
        self.filename = fname

    def close(self):
        if self._id >= 0:
            with _lock:
                self._h.lib.H5Fclose(self._id)
            self._id = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    # ---- structure -------------------------------------------------------------------------------------------------
    def __contains__(self, name):
        L = self._h.lib
        path = ''
        with _lock:
            for part in name.strip('/').split('/'):
                path = part if not path else path + '/' + part
                if L.H5Lexists(self._id, path.encode(), 0) <= 0:
                    return False
        return True

    def _kind(self, name):
        L = self._h.lib
        with _lock:
            oid = L.H5Oopen(self._id, name.encode(), 0)
            if oid < 0:
                raise KeyError(name)
            kind = L.H5Iget_type(oid)
            L.H5Oclose(oid)
        return kind

    def is_group(self, name):
        return self._kind(name) == H5I_GROUP

    def create_group(self, name):
        L = self._h.lib
        with _lock:
            gid = L.H5Gcreate2(self._id, name.encode(), self._h.lcpl, 0, 0)
            if gid < 0:
                raise H5Error(f'cannot create group {name!r}')
            L.H5Gclose(gid)

    def keys(self, group=''):
        """member names of a group in name order (h5py iterates the name index)"""
        L = self._h.lib
        grp = (group.strip('/') or '.').encode()

        class Info(ctypes.Structure):
            _fields_ = [('storage_type', ctypes.c_int), ('nlinks', hsize_t), ('max_corder', ctypes.c_int64), ('mounted', ctypes.c_uint)]
        info = Info()
        out = []
        with _lock:
            if L.H5Gget_info_by_name(self._id, grp, ctypes.byref(info), 0) < 0:
                raise KeyError(group)
            for k in range(info.nlinks):
                n = L.H5Lget_name_by_idx(self._id, grp, 0, 0, k, None, 0, 0)
                buf = ctypes.create_string_buffer(n + 1)
                L.H5Lget_name_by_idx(self._id, grp, 0, 0, k, buf, n + 1, 0)
                out.append(buf.value.decode())
        return out

    # ---- datasets --------------------------------------------------------------------------------------------------
    def create_dataset(self, name, data, compression=None):
        h, L = self._h, self._h.lib
        arr = np.asarray(data)
        if arr.dtype == np.bool_:
            tid, buf = h.bool_t, np.ascontiguousarray(arr, dtype=np.int8)
        else:
            dt = arr.dtype.newbyteorder('<') if arr.dtype.byteorder == '>' else arr.dtype
            if np.dtype(dt) not in h.std:
                raise TypeError(f'{name}: dtype {arr.dtype} has no mapping in the FEABAS layouts')
            tid, buf = h.std[np.dtype(dt)], np.ascontiguousarray(arr, dtype=dt)
        if compression not in (None, False, 'gzip'):
            raise ValueError(f'compression {compression!r}: the reference writes gzip or nothing')
        with _lock:
            if arr.ndim == 0:
                sid = L.H5Screate(H5S_SCALAR)
            else:
                dims = (hsize_t * arr.ndim)(*arr.shape)
                sid = L.H5Screate_simple(arr.ndim, dims, None)
            dcpl = 0
            if compression == 'gzip' and arr.ndim > 0 and arr.size > 0:
                if not h.has_gzip:
                    raise H5Error('this libhdf5 was built without the deflate filter')
                dcpl = L.H5Pcreate(h.dcpl_cls)
                chunk = _chunk_shape(arr.shape, buf.dtype.itemsize)
                L.H5Pset_chunk(dcpl, arr.ndim, (hsize_t * arr.ndim)(*chunk))
                L.H5Pset_deflate(dcpl, GZIP_LEVEL)
            did = L.H5Dcreate2(self._id, name.encode(), tid, sid, h.lcpl, dcpl, 0)
            rc = -1
            if did >= 0:
                rc = 0 if buf.size == 0 else L.H5Dwrite(did, tid, 0, 0, 0, buf.ctypes.data)
                L.H5Dclose(did)
            if dcpl:
                L.H5Pclose(dcpl)
            L.H5Sclose(sid)
        if did < 0 or rc < 0:
            raise H5Error(f'cannot write dataset {name!r} (exists already, or the file is read-only)')

    def __getitem__(self, name):
        h, L = self._h, self._h.lib
        with _lock:
            did = L.H5Dopen2(self._id, name.encode(), 0)
            if did < 0:
                raise KeyError(name)
            try:
                tid = L.H5Dget_type(did)
                sid = L.H5Dget_space(did)
                nd = L.H5Sget_simple_extent_ndims(sid)
                dims = (hsize_t * max(nd, 1))()
                if nd > 0:
                    L.H5Sget_simple_extent_dims(sid, dims, None)
                shape = tuple(int(dims[k]) for k in range(nd))
                cls, size = L.H5Tget_class(tid), int(L.H5Tget_size(tid))
                as_bool = False
                if cls == H5T_INTEGER:
                    dt = np.dtype(('i' if L.H5Tget_sign(tid) else 'u') + str(size))
                    mem = h.std[dt]
                elif cls == H5T_FLOAT and size in (4, 8):
                    dt = np.dtype('f' + str(size))
                    mem = h.std[dt]
                elif cls == H5T_ENUM and size == 1 and L.H5Tget_nmembers(tid) == 2:
                    dt, mem, as_bool = np.dtype('i1'), tid, True
                elif cls == H5T_STRING and L.H5Tis_variable_str(tid) <= 0:
                    dt, mem = np.dtype('S' + str(size)), tid
                else:
                    raise TypeError(f'{name}: HDF5 type class {cls} / size {size} is not part of the FEABAS layouts')
                out = np.empty(shape, dtype=dt)
                if out.size and L.H5Dread(did, mem, 0, 0, 0, out.ctypes.data) < 0:
                    raise H5Error(f'cannot read dataset {name!r}')
                L.H5Sclose(sid)
                L.H5Tclose(tid)
            finally:
                L.H5Dclose(did)
        if as_bool:
            out = out.astype(np.bool_)
        return out[()] if nd == 0 else out

    def storage(self, name):
        """(chunked?, number of filters) of a dataset -- for the layout tests"""
        L = self._h.lib
        with _lock:
            did = L.H5Dopen2(self._id, name.encode(), 0)
            if did < 0:
                raise KeyError(name)
            pl = L.H5Dget_create_plist(did)
            res = (L.H5Pget_layout(pl) == 2, L.H5Pget_nfilters(pl))
            L.H5Pclose(pl)
            L.H5Dclose(did)
        return res


def str_to_numpy_ascii(s):                 # common.py:438-440
    return np.frombuffer(s.encode('ascii'), dtype=np.uint8)


def numpy_to_str_ascii(ar):                # common.py:433-435
    return np.asarray(ar).clip(0, 255).astype(np.uint8).ravel().tobytes().decode('ascii')


# ---- Stitcher file ----------------------------------------------------------------------------------------------------
def save_stitcher_h5(fname, imgrootdir, resolution, imgrelpaths, init_bboxes, matches=None, match_strains=None,
                     brightness_contrast=None, groupings=None, compression=True):
    """stitcher.py:126-181 up to the matches (the meshes of ``--mode optimization`` are written by that stage).
    matches[(i, j)] = (xy0, xy1, weight), match_strains[(i, j)] = strain -- what ``stitcher.match_list_of_overlaps``
    returns."""
    comp = 'gzip' if compression else None
    with H5File(fname, 'w') as f:
        f.create_dataset('imgrootdir', str_to_numpy_ascii(imgrootdir))
        f.create_dataset('resolution', resolution)
        f.create_dataset('imgrelpaths', str_to_numpy_ascii('\n'.join(imgrelpaths)), compression=comp)
        f.create_dataset('init_bboxes', np.asarray(init_bboxes), compression=comp)
        if groupings is not None:
            f.create_dataset('groupings', np.asarray(groupings), compression=comp)
        for uids, (xy0, xy1, weight) in (matches or {}).items():
            strain = match_strains[uids]
            data = np.concatenate((xy0, xy1, weight, strain), axis=None).astype(np.float32, copy=False)
            f.create_dataset('matches/' + '_'.join(str(int(s)) for s in uids), data, compression=comp)
        if brightness_contrast:
            f.create_dataset('matches_brightness_contrast_keys', np.array(list(brightness_contrast.keys())), compression=comp)
            f.create_dataset('matches_brightness_contrast_vals', np.array(list(brightness_contrast.values())), compression=comp)


def load_stitcher_matches(fname, imgrelpaths=None):
    """stitcher.py:184-222.  ``imgrelpaths`` given = ``check_order``: tile indices of the file are mapped onto the
    caller's tile list by name, pairs with an unknown tile dropped.
    Returns (matches, match_strains, brightness_contrast)."""
    matches, strains, bc = {}, {}, {}
    with H5File(fname, 'r') as f:
        if 'matches' not in f:
            return matches, strains, bc
        mapper = None
        if imgrelpaths is not None:
            names = numpy_to_str_ascii(f['imgrelpaths']).split('\n')
            lut = {name: k for k, name in enumerate(imgrelpaths)}
            mapper = np.array([lut.get(s, -1) for s in names])
        for key in f.keys('matches'):
            uid0, uid1 = (int(s) for s in key.split('_'))
            if mapper is not None:
                uid0, uid1 = int(mapper[uid0]), int(mapper[uid1])
            if uid0 < 0 or uid1 < 0:
                continue
            data = f['matches/' + key]
            npt = int((data.size - 1) / 5)
            matches[(uid0, uid1)] = (data[:2 * npt].reshape(-1, 2), data[2 * npt:4 * npt].reshape(-1, 2), data[4 * npt:5 * npt])
            strains[(uid0, uid1)] = data[-1]
        if 'matches_brightness_contrast_keys' in f and 'matches_brightness_contrast_vals' in f:
            keys, vals = f['matches_brightness_contrast_keys'], f['matches_brightness_contrast_vals']
            if mapper is not None:
                keys = mapper[keys]
                ok = np.all(keys >= 0, axis=-1)
                keys, vals = keys[ok], vals[ok]
            bc.update({tuple(int(i) for i in k): v for k, v in zip(keys, vals)})
    return matches, strains, bc


# ---- section-pair match file of the aligner ---------------------------------------------------------------------------
def save_section_match_h5(fname, xy0, xy1, weight, resolution, strain, name0, name1):
    """aligner.py:134-141"""
    with H5File(fname, 'w') as f:
        f.create_dataset('xy0', np.asarray(xy0), compression='gzip')
        f.create_dataset('xy1', np.asarray(xy1), compression='gzip')
        f.create_dataset('weight', np.asarray(weight), compression='gzip')
        f.create_dataset('resolution', resolution)
        f.create_dataset('strain', strain)
        f.create_dataset('name0', str_to_numpy_ascii(name0))
        f.create_dataset('name1', str_to_numpy_ascii(name1))


def read_matches_from_h5(fname, target_resolution=None):
    """aligner.py:26-44 -> (xy0, xy1, weight, strain); coordinates rescaled about the pixel centres (spatial.py:77-86)"""
    with H5File(fname, 'r') as f:
        xy0, xy1 = f['xy0'], f['xy1']
        weight = f['weight'].ravel()
        resolution = np.asarray(f['resolution']).item()
        strain = np.asarray(f['strain']).item() if 'strain' in f else DEFAULT_AVG_DEFORM
    if target_resolution is not None and resolution != target_resolution:
        scale = resolution / target_resolution
        xy0 = (xy0 + 0.5) * scale - 0.5
        xy1 = (xy1 + 0.5) * scale - 0.5
    return xy0, xy1, weight, strain


# ---- Mesh file --------------------------------------------------------------------------------------------------------
_GEAR_NAMES = {const.MESH_GEAR_FIXED: 'fixed', const.MESH_GEAR_MOVING: 'moving', const.MESH_GEAR_STAGING: 'staging'}
_MODEL_NAMES = ('MATERIAL_MODEL_ENG', 'MATERIAL_MODEL_SVK', 'MATERIAL_MODEL_NHK')              # constant.py:37


def _material_entry(mesh, uid, model, nu, mult, fk, area_constraint=1.0):
    entry = {'enable_mesh': True, 'area_constraint': float(area_constraint), 'render': True, 'render_weight': 1.0, 'type': _MODEL_NAMES[model],
             'stiffness_multiplier': mult, 'poisson_ratio': nu, 'uid': int(uid)}
    if fk >= 0:
        f = mesh.stiffness_funcs[fk]
        if not hasattr(f, 'strain'):
            # (the reference serialises arbitrary callables with dill, common.func_to_str: not in the image)
            raise NotImplementedError('a mesh file can carry piecewise-linear stiffness tables only; this mesh has a Python callable as stiffness function')
        entry['stiffness_multiplier'] = float(mesh.func_matmult[fk])
        entry['stiffness_func_factory'] = 'feabas.material.asymmetrical_elasticity'
        entry['stiffness_func_params'] = {'strain': f.strain.tolist(), 'stiffness': f.stiffness.tolist()}
    return entry


def _named_material_entries(mesh, base, tri_func):
    """the table of a mesh that carries NAMED materials (``material_ids`` + ``material_names``, as read from a mesh file or
    handed over by the mesher; mesh.py:257-263): the names, the uids and the area constraints are written back as they are
    -- ``optimize_linear(remove_material_dof=name)`` and the refinement regions of ``distribute_matching_blocks`` select by
    them after the round trip meshing -> matching -> optimisation -- with the constitutive parameters of the triangles of
    each uid.  None when the mesh has no names or when the triangles of one uid disagree (the caller then writes one
    synthesized entry per distinct parameter set)."""
    ids = getattr(mesh, 'material_ids', None)
    names = getattr(mesh, 'material_names', None)
    if ids is None or not names:
        return None
    ids = np.asarray(ids).ravel()
    nt = mesh.num_triangles
    if ids.size != nt:
        return None
    if mesh.tri_model is None:
        rows = np.tile(np.array(base, dtype=np.float64), (nt, 1))
    else:
        fk = np.full(nt, -1.0) if tri_func is None else tri_func.astype(np.float64)
        rows = np.stack((mesh.tri_model.astype(np.float64), mesh.tri_nu, mesh.tri_matmult.astype(np.float64), fk), axis=1)
    constraints = getattr(mesh, 'material_area_constraints', None) or {}
    name_of = {}
    for name, uid in names.items():
        if int(uid) in name_of:
            return None                                       # two names for one uid: not a table the reference writes
        name_of[int(uid)] = name
    table = {}
    for uid in sorted(set(name_of) | set(int(u) for u in np.unique(ids))):
        sel = np.flatnonzero(ids == uid)
        if sel.size:
            r = rows[sel[0]]
            if np.any(rows[sel] != r):
                return None
            combo = (int(r[0]), float(r[1]), float(r[2]), int(r[3]))
        else:
            combo = base                                      # a named material without triangles in this mesh
        name = name_of.get(uid, 'default' if uid == 0 and 'default' not in names else f'material_{uid}')
        table[name] = _material_entry(mesh, uid, *combo, area_constraint=constraints.get(name, 1.0))
        rw = (getattr(mesh, 'material_render_weights', None) or {}).get(name, None)
        if rw is None and sel.size and getattr(mesh, 'tri_render_weight', None) is not None:
            rw = float(mesh.tri_render_weight[sel[0]])
        if rw is not None:                                    # material.py:50-54; render = False is carried as -(render_weight + 1) (Mesh.triangle_mask_for_render)
            table[name]['render'] = bool(rw >= 0)
            table[name]['render_weight'] = float(rw) if rw >= 0 else float(-rw - 1.0)
    lo, hi = int(ids.min(initial=0)), int(ids.max(initial=0))
    dt = np.int8 if -128 <= lo and hi < 128 else (np.int16 if -32768 <= lo and hi < 32768 else np.int32)
    return ids.astype(dt), table


def _material_entries(mesh):
    """(material_ids [T] int8, table dict) of a mesh: one entry per distinct (model, Poisson ratio, multiplier, stiffness
    function); the mesh-wide material is 'default' (uid 0, material.py:333-342).  A stiffness function is written the way the
    reference's Material.to_dict does (material.py:106-113): factory name + the knots of the table."""
    nt = mesh.num_triangles
    base = (0, float(mesh.poisson_ratio), float(mesh.material_multiplier), -1)
    tri_func = getattr(mesh, 'tri_func', None)
    named = _named_material_entries(mesh, base, tri_func)
    if named is not None:
        return named
    if mesh.tri_model is None:
        combos, inv = [base], np.zeros(nt, dtype=np.int64)
    else:
        fk = np.full(nt, -1.0) if tri_func is None else tri_func.astype(np.float64)
        rows = np.stack((mesh.tri_model.astype(np.float64), mesh.tri_nu, mesh.tri_matmult.astype(np.float64), fk), axis=1)
        uniq, inv = np.unique(rows, axis=0, return_inverse=True)
        combos = [(int(r[0]), float(r[1]), float(r[2]), int(r[3])) for r in uniq]
        inv = np.asarray(inv).reshape(-1)
    order = sorted(range(len(combos)), key=lambda k: (combos[k] != base, k))         # the mesh-wide material first
    if combos[order[0]] != base:
        combos, order, inv = [base] + combos, [0] + [k + 1 for k in order], inv + 1
    uid_of = np.empty(len(combos), dtype=np.int64)
    table = {}
    for uid, k in enumerate(order):
        uid_of[k] = uid
        table['default' if uid == 0 else f'material_{uid}'] = _material_entry(mesh, uid, *combos[k])
    return uid_of[inv].astype(np.int8 if len(combos) < 128 else np.int16), table


def mesh_init_dict(mesh, vertex_flags=None, save_material=True):
    """mesh.py:543-580: the keyword dictionary that re-creates the mesh; gears that alias an already saved gear are left
    out, exactly as the reference skips them (``actual_initialized_gears``)."""
    out = {'vertices': mesh._vertices[const.MESH_GEAR_INITIAL]}
    if np.any(mesh._offsets[const.MESH_GEAR_INITIAL]):
        out['initial_offset'] = mesh._offsets[const.MESH_GEAR_INITIAL]
    if vertex_flags is None:
        vertex_flags = [g for g in _GEAR_NAMES if mesh._vertices.get(g) is not None]
    saved = [mesh._vertices[const.MESH_GEAR_INITIAL]]
    for gear in vertex_flags:
        if gear not in _GEAR_NAMES:
            continue
        v = mesh._vertices.get(gear)
        if v is None or any(v is s for s in saved):
            continue
        out[_GEAR_NAMES[gear] + '_vertices'] = v
        out[_GEAR_NAMES[gear] + '_offset'] = mesh._offsets[gear]
        saved.append(v)
    out['triangles'] = mesh.triangles
    if mesh._stiffness_multiplier is not None:
        out['stiffness_multiplier'] = mesh._stiffness_multiplier
    if save_material:
        ids, table = _material_entries(mesh)
        out['material_ids'] = ids
        out['material_table'] = json.dumps(table, indent=2)
    out['resolution'] = mesh.resolution
    out['epsilon'] = getattr(mesh, 'epsilon', 1e-5)              # constant.EPSILON0, mesh.py:267
    name = getattr(mesh, 'name', None)
    if name:
        out['name'] = name
    out['locked'] = bool(mesh.locked)
    out['uid'] = mesh.uid
    out['soft_factor'] = mesh.soft_factor
    return out


def save_mesh_h5(mesh, f, prefix='', vertex_flags=None, save_material=True, compression=True):
    """mesh.py:822-857.  ``f``: a path or an open ``H5File`` (the Stitcher file holds its meshes under prefixes)."""
    if prefix and not prefix.endswith('/'):
        prefix += '/'
    own = not isinstance(f, H5File)
    h = H5File(f, 'w') if own else f
    try:
        for key, val in mesh_init_dict(mesh, vertex_flags, save_material).items():
            if val is None:
                continue
            if isinstance(val, str):
                val = str_to_numpy_ascii(val)
            scalar = np.isscalar(val)
            h.create_dataset(prefix + key, val, compression=None if (scalar or not compression) else 'gzip')
    finally:
        if own:
            h.close()


def load_mesh_h5(f, prefix='', cls=None, **kwargs):
    """mesh.py:798-819: every dataset under the prefix is a constructor keyword."""
    from .mesh import Mesh
    cls = cls or Mesh
    if prefix and not prefix.endswith('/'):
        prefix += '/'
    own = not isinstance(f, H5File)
    h = H5File(f, 'r') if own else f
    try:
        init = {key: h[prefix + key] for key in h.keys(prefix)}
    finally:
        if own:
            h.close()
    vertices, triangles = init.pop('vertices'), init.pop('triangles')
    ids, table = init.pop('material_ids', None), init.pop('material_table', None)
    if 'name' in init:
        init['name'] = numpy_to_str_ascii(init['name'])
    if ids is not None and table is not None:
        table = json.loads(numpy_to_str_ascii(table))
        by_uid = {int(m['uid']): m for m in table.values()}
        model_of = lambda m: _MODEL_NAMES.index(m['type'].upper()) if isinstance(m.get('type', 0), str) else int(m.get('type', 0))
        d = table.get('default', {})
        init['material_ids'] = np.asarray(ids).ravel().astype(np.int32)
        init['material_names'] = {name: int(m['uid']) for name, m in table.items()}
        init['material_area_constraints'] = {name: float(m.get('area_constraint', 1.0)) for name, m in table.items()}
        # render weights (material.py:50-54): which triangles take part in block placement / rendering and where matches may land
        rws = {name: (float(m.get('render_weight', 1.0)) if m.get('render', True) else -(float(m.get('render_weight', 1.0)) + 1.0)) for name, m in table.items()}
        if any(w != 1.0 for w in rws.values()):
            init['material_render_weights'] = rws
        init['poisson_ratio'] = d.get('poisson_ratio', 0.0)
        init['material_multiplier'] = d.get('stiffness_multiplier', 1.0)
        uids = np.unique(ids)
        has_func = any(by_uid[int(u)].get('stiffness_func_factory') is not None for u in uids)
        if uids.size > 1 or model_of(by_uid[int(uids[0])]) != 0 or int(uids[0]) != int(d.get('uid', 0)) or has_func:
            mats = [by_uid[int(u)] for u in ids.ravel()] if uids.size > 8 else None
            pick = (lambda fn, dt: np.array([fn(m) for m in mats], dtype=dt)) if mats is not None else \
                (lambda fn, dt: np.select([ids.ravel() == u for u in uids], [fn(by_uid[int(u)]) for u in uids]).astype(dt))
            init['tri_model'] = pick(model_of, np.int32)
            init['tri_nu'] = pick(lambda m: m.get('poisson_ratio', 0.0), np.float64)
            init['tri_matmult'] = pick(lambda m: m.get('stiffness_multiplier', 1.0), np.float32)
            if has_func:
                # materials with a stiffness function (material.py:60-62): one table per such material
                from .material import stiffness_func_from_spec
                fuids = [int(u) for u in uids if by_uid[int(u)].get('stiffness_func_factory') is not None]
                init['stiffness_funcs'] = [stiffness_func_from_spec(by_uid[u]['stiffness_func_factory'], by_uid[u].get('stiffness_func_params', {})) for u in fuids]
                init['func_matmult'] = [float(by_uid[u].get('stiffness_multiplier', 1.0)) for u in fuids]
                tf = np.full(ids.size, -1, dtype=np.int32)
                for k, u in enumerate(fuids):
                    tf[ids.ravel() == u] = k
                init['tri_func'] = tf
    init.update(kwargs)
    mesh = cls(vertices, triangles, **{k: v for k, v in init.items() if k not in ('name', 'epsilon', 'token')})
    for k in ('name', 'epsilon'):
        if k in init:
            setattr(mesh, k, init[k])
    return mesh
