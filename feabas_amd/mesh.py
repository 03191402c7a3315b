"""Host mirror of the part of feabas.mesh.Mesh that sits on the FEM path.

Vertex state (four "gears" + offsets, feabas/mesh.py:233-293, 1221-1325),
point location helpers (2191-2228), field application (2381-2458) are small
host bookkeeping; ``stiffness_matrix`` (2893-3083) is assembled on the GPU.
Meshing itself (``triangle``, shapely) is out of scope: (vertices, triangles)
are inputs.
"""
import ctypes as C

import numpy as np
from scipy import sparse
from scipy.sparse import csgraph

from . import _lib, common
from .material import StiffnessTable
from . import constant as const


class Mesh:
    uid_counter = 0.0

    def __init__(self, vertices, triangles, **kwargs):
        vertices = np.asarray(vertices, dtype=np.float64).reshape(-1, 2)
        self.triangles = np.ascontiguousarray(np.asarray(triangles).reshape(-1, 3), dtype=np.int32)
        self._vertices = {const.MESH_GEAR_INITIAL: vertices,
                          const.MESH_GEAR_FIXED: kwargs.get('fixed_vertices', vertices),
                          const.MESH_GEAR_MOVING: kwargs.get('moving_vertices', None),
                          const.MESH_GEAR_STAGING: kwargs.get('staging_vertices', None)}
        zero = np.zeros((1, 2), dtype=np.float64)
        self._offsets = {const.MESH_GEAR_INITIAL: kwargs.get('initial_offset', zero)}
        if ('fixed_vertices' not in kwargs) and ('fixed_offset' not in kwargs):       # mesh.py:241-244
            self._offsets[const.MESH_GEAR_FIXED] = kwargs.get('initial_offset', zero)
        else:
            self._offsets[const.MESH_GEAR_FIXED] = kwargs.get('fixed_offset', zero)
        self._offsets[const.MESH_GEAR_MOVING] = kwargs.get('moving_offset', zero)
        self._offsets[const.MESH_GEAR_STAGING] = kwargs.get('staging_offset', zero)
        self._current_gear = const.MESH_GEAR_FIXED
        mult = kwargs.get('stiffness_multiplier', None)
        self._stiffness_multiplier = None if mult is None else np.ascontiguousarray(mult, dtype=np.float32)
        # one linear engineering material per mesh (the default material table)
        self.poisson_ratio = float(kwargs.get('poisson_ratio', 0.0))
        self.material_multiplier = float(kwargs.get('material_multiplier', 1.0))
        # optional per-triangle materials (material.py: model 0 ENG, 1 SVK, 2 NHK; Poisson ratio; multiplier)
        tm = kwargs.get('tri_model', None)
        self.tri_model = None if tm is None else np.ascontiguousarray(tm, dtype=np.int32)
        if self.tri_model is not None:
            nt = self.triangles.shape[0]
            self.tri_nu = np.ascontiguousarray(np.broadcast_to(kwargs.get('tri_nu', self.poisson_ratio), (nt,)), dtype=np.float64)
            self.tri_matmult = np.ascontiguousarray(np.broadcast_to(kwargs.get('tri_matmult', self.material_multiplier), (nt,)), dtype=np.float32)
        # optional stiffness functions (Material._stiffness_func, material.py:128-131): tri_func[t] = index into stiffness_funcs of
        # the function of triangle t's material, -1 = none; func_matmult[k] = that material's multiplier in double precision
        tf = kwargs.get('tri_func', None)
        self.tri_func = None
        if tf is not None and np.any(np.asarray(tf) >= 0):
            from .material import stiffness_func_from_spec
            self.tri_func = np.ascontiguousarray(tf, dtype=np.int32)
            self.stiffness_funcs = [stiffness_func_from_spec(f) for f in kwargs['stiffness_funcs']]
            if self.tri_func.max() >= len(self.stiffness_funcs) or self.tri_func.min() < -1:
                raise ValueError('tri_func names a stiffness function that is not in stiffness_funcs')
            if self.tri_model is None:
                nt = self.triangles.shape[0]
                self.tri_model = np.zeros(nt, dtype=np.int32)
                self.tri_nu = np.full(nt, self.poisson_ratio, dtype=np.float64)
                self.tri_matmult = np.full(nt, self.material_multiplier, dtype=np.float32)
            fm = kwargs.get('func_matmult', None)
            if fm is None:
                fm = [float(self.tri_matmult[np.flatnonzero(self.tri_func == k)[0]]) if np.any(self.tri_func == k) else 1.0
                      for k in range(len(self.stiffness_funcs))]
            self.func_matmult = np.ascontiguousarray(fm, dtype=np.float64)
        # optional names of the materials (mesh.py:257-263, MaterialTable.named_table): material_ids[t] = uid of triangle t's
        # material, material_names = {name: uid} -- what optimize_linear(remove_material_dof=<name>) selects regions by
        mids = kwargs.get('material_ids', None)
        self.material_ids = None if mids is None else np.ascontiguousarray(mids).ravel()
        self.material_names = dict(kwargs.get('material_names', None) or {})
        # area_constraint of the named materials (material.py:22-26): < 1 marks a refinement region of the block distributor
        self.material_area_constraints = dict(kwargs.get('material_area_constraints', None) or {})
        # render weight of every triangle's material (material.py:27-30, 50-54; a material that is NOT rendered carries -(weight + 1)): which triangles
        # take part in block placement and rendering (triangle_mask_for_render) and where matches may land (tri_finder,
        # mesh.py:2168-2170).  Given per triangle, or per named material (material_render_weights = {name: weight}); absent = all 1
        rw = kwargs.get('tri_render_weight', None)
        mrw = kwargs.get('material_render_weights', None)
        if rw is None and mrw and self.material_ids is not None:
            rw = np.ones(self.triangles.shape[0], dtype=np.float32)
            for name, wgt in mrw.items():
                if name in self.material_names:
                    rw[self.material_ids == self.material_names[name]] = wgt
        self.tri_render_weight = None if rw is None else np.ascontiguousarray(rw, dtype=np.float32)
        self.material_render_weights = dict(mrw or {})
        self.resolution = kwargs.get('resolution', 4.0)
        self.locked = kwargs.get('locked', False)
        self.soft_factor = kwargs.get('soft_factor', 1.0)
        uid = kwargs.get('uid', None)
        if uid is None:
            self.uid = float(Mesh.uid_counter)
            Mesh.uid_counter += 1
        else:
            self.uid = float(uid)
            Mesh.uid_counter = float(max(Mesh.uid_counter, uid) + 1)
        self._trifinders = {}

    @classmethod
    def from_bbox(cls, bbox, cartesian=True, **kwargs):
        """feabas/mesh.py:403-435, cartesian branch: a regular node grid over the bounding box (vertices at
        pixel centres - 0.5).  The reference hands the rectangles to `triangle`, whose choice of diagonal is
        implementation defined; here every cell (a b / c d) becomes (a, b, d), (a, d, c).  The node
        coordinates of the grid are kept in ``grid_xs`` / ``grid_ys`` for O(1) point location."""
        if not cartesian:
            raise NotImplementedError('Mesh.from_bbox(cartesian=False) needs the `triangle` mesher (SURVEY.md sec.7)')
        mesh_size = kwargs.pop('mesh_size')
        min_num_blocks = kwargs.pop('min_num_blocks', 2)
        max_aspect_ratio = kwargs.pop('max_aspect_ratio', 2)
        x0, y0 = float(bbox[0]), float(bbox[1])
        wd, ht = float(bbox[2]) - x0, float(bbox[3]) - y0
        nx = max(np.round(wd / mesh_size), min_num_blocks)
        ny = max(np.round(ht / mesh_size), min_num_blocks)
        dx, dy = wd / nx, ht / ny
        if dx > max_aspect_ratio * dy:
            dx = max_aspect_ratio * dy
        elif dy > max_aspect_ratio * dx:
            dy = max_aspect_ratio * dx
        nx = int(np.ceil(wd / dx)) + 1
        ny = int(np.ceil(ht / dy)) + 1
        xs = np.linspace(x0, x0 + wd, num=nx, endpoint=True) - 0.5
        ys = np.linspace(y0, y0 + ht, num=ny, endpoint=True) - 0.5
        vx, vy = np.meshgrid(xs, ys)
        v = np.stack((vx.ravel(), vy.ravel()), axis=-1)
        idx = np.arange(nx * ny).reshape(ny, nx)
        a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
        tri = np.stack((np.stack((a, b, d), -1), np.stack((a, d, c), -1)), axis=1).reshape(-1, 3)
        m = cls(v, tri, **kwargs)
        m.grid_xs, m.grid_ys = xs, ys
        return m

    def locate_cartesian(self, pts):
        """triangle id of points given in the INITIAL gear of a from_bbox(cartesian=True) mesh"""
        xs, ys = self.grid_xs, self.grid_ys
        i = np.clip(np.searchsorted(xs, pts[..., 0], side='right') - 1, 0, xs.size - 2)
        j = np.clip(np.searchsorted(ys, pts[..., 1], side='right') - 1, 0, ys.size - 2)
        u = (pts[..., 0] - xs[i]) / (xs[i + 1] - xs[i])
        w = (pts[..., 1] - ys[j]) / (ys[j + 1] - ys[j])
        return (2 * (j * (xs.size - 1) + i) + (w > u)).astype(np.int32)

    # ------------------------------------------------------------------ state
    @property
    def num_vertices(self):
        return self._vertices[const.MESH_GEAR_INITIAL].shape[0]

    @property
    def num_triangles(self):
        return self.triangles.shape[0]

    @property
    def is_linear(self):                                   # mesh.py:1563-1568: engineering materials without a stiffness function only
        return (self.tri_model is None or not np.any(self.tri_model != const.MATERIAL_MODEL_ENG)) and self.tri_func is None

    def triangles_of_material(self, name):
        """boolean [T]: the triangles whose material carries this name (none if the mesh has no named materials)"""
        if self.material_ids is None or name not in self.material_names:
            return np.zeros(self.num_triangles, dtype=bool)
        return self.material_ids == self.material_names[name]

    @property
    def linear_triangle_mask(self):                        # mesh.py:1571-1580
        m = np.ones(self.num_triangles, dtype=bool)
        if self.tri_model is not None:
            m &= self.tri_model == const.MATERIAL_MODEL_ENG
        if self.tri_func is not None:
            m &= self.tri_func < 0
        return m

    @property
    def stiffness_multiplier(self):
        if self._stiffness_multiplier is None:
            return np.ones(self.num_triangles, dtype=np.float32)
        return self._stiffness_multiplier

    def vertices(self, gear=None):
        gear = self._current_gear if gear is None else gear
        v = self._vertices[gear]
        if v is None:
            if gear == const.MESH_GEAR_MOVING:
                return self._vertices[const.MESH_GEAR_FIXED]
            if gear == const.MESH_GEAR_STAGING:
                return self.vertices(const.MESH_GEAR_MOVING)
        return v

    def offset(self, gear=None):
        gear = self._current_gear if gear is None else gear
        if self._vertices[gear] is None:
            if gear == const.MESH_GEAR_MOVING:
                return self._offsets[const.MESH_GEAR_FIXED]
            return self.offset(const.MESH_GEAR_MOVING)
        return self._offsets[gear]

    def vertices_w_offset(self, gear=None):
        return self.vertices(gear) + self.offset(gear)

    def clear_staging(self):
        """the STAGING gear falls back to MOVING again (what the Newton-Raphson loop does with a mesh after it annealed its
        resting shape from STAGING, optimizer.py:1511-1515)"""
        if self.locked:
            return
        self._vertices[const.MESH_GEAR_STAGING] = None
        self._offsets[const.MESH_GEAR_STAGING] = np.zeros((1, 2), dtype=np.float64)

    def bbox(self, gear=const.MESH_GEAR_MOVING, offsetting=True):
        v = self.vertices_w_offset(gear) if offsetting else self.vertices(gear)
        lo, hi = v.min(axis=0), v.max(axis=0)
        return np.array((lo[0], lo[1], hi[0], hi[1]))

    def lock(self):
        self.locked = True

    def unlock(self):
        self.locked = False

    def copy(self, **override):
        kw = dict(fixed_vertices=self._vertices[const.MESH_GEAR_FIXED],
                  moving_vertices=self._vertices[const.MESH_GEAR_MOVING],
                  staging_vertices=self._vertices[const.MESH_GEAR_STAGING],
                  initial_offset=self._offsets[const.MESH_GEAR_INITIAL],
                  fixed_offset=self._offsets[const.MESH_GEAR_FIXED],
                  moving_offset=self._offsets[const.MESH_GEAR_MOVING],
                  staging_offset=self._offsets[const.MESH_GEAR_STAGING],
                  stiffness_multiplier=self._stiffness_multiplier, poisson_ratio=self.poisson_ratio,
                  tri_model=self.tri_model, tri_nu=getattr(self, 'tri_nu', None) if self.tri_model is not None else self.poisson_ratio,
                  tri_matmult=getattr(self, 'tri_matmult', None) if self.tri_model is not None else self.material_multiplier,
                  material_multiplier=self.material_multiplier, resolution=self.resolution,
                  locked=self.locked, soft_factor=self.soft_factor, uid=self.uid,
                  material_ids=self.material_ids, material_names=self.material_names, material_area_constraints=self.material_area_constraints,
                  tri_render_weight=getattr(self, 'tri_render_weight', None), material_render_weights=getattr(self, 'material_render_weights', None))
        if self.tri_func is not None:
            kw.update(tri_func=self.tri_func, stiffness_funcs=self.stiffness_funcs, func_matmult=self.func_matmult)
        kw.update(override)
        twin = Mesh(self._vertices[const.MESH_GEAR_INITIAL], self.triangles, **kw)
        # what depends on the triangles and the INITIAL vertices alone goes along (read-only arrays): the outline edges and the
        # triangle areas (every Link asks for them, optimizer.py:26-30)
        for name in ('_outline', '_area_initial'):
            if getattr(self, name, None) is not None:
                setattr(twin, name, getattr(self, name))
        return twin

    # ------------------------------------------------------------------ transformations
    def save_to_h5(self, fname, vertex_flags=None, **kwargs):    # mesh.py:822-857 (layout: feabas_amd/h5wire.py)
        from . import h5wire
        h5wire.save_mesh_h5(self, fname, prefix=kwargs.get('prefix', ''), vertex_flags=vertex_flags,
                            save_material=kwargs.get('save_material', True), compression=kwargs.get('compression', True))

    @classmethod
    def from_h5(cls, fname, prefix='', **kwargs):                # mesh.py:798-819
        from . import h5wire
        return h5wire.load_mesh_h5(fname, prefix=prefix, cls=cls, **kwargs)

    def _changed(self, gear):
        self._trifinders.pop(gear, None)
        if gear == const.MESH_GEAR_INITIAL:
            self._area_initial = None

    def set_vertices(self, v, gear, vtx_mask=None):        # mesh.py:2232-2243
        if self.locked:
            return
        if self._vertices[gear] is None:
            self.set_offset(self.offset(gear), gear)
        if vtx_mask is None:
            self._vertices[gear] = v
        else:
            cur = self.vertices(gear).copy()
            cur[vtx_mask] = v
            self._vertices[gear] = cur
        self._changed(gear)

    def set_offset(self, offset, gear):
        if self.locked:
            return
        self._offsets[gear] = offset

    def apply_translation(self, dxy, gear):                # mesh.py:2272-2286 (unmasked)
        dxy = np.asarray(dxy, dtype=np.float64).reshape(1, 2)
        if self.locked or not np.any(dxy):
            return
        v = self.vertices(gear)
        off = self.offset(gear)
        self._vertices[gear] = v
        self.set_offset(off + dxy, gear)

    def apply_field(self, dxy, gear, vtx_mask=None):       # mesh.py:2381-2397
        if self.locked or not np.any(dxy):
            return
        v0 = self.vertices(gear)
        off0 = self.offset(gear)
        if vtx_mask is not None:                           # a masked field moves only those vertices, the offset stays
            self.set_vertices(v0[vtx_mask] + dxy, gear, vtx_mask=vtx_mask)
            self.set_offset(off0, gear)
            return
        d2 = dxy.reshape(-1, 2)
        # (the mean of the field column by column: numpy's axis-0 reduction of an (N, 2) array runs an inner loop of length 2,
        # 2 ms at 250 k nodes against 0.3 ms for two strided sums)
        m = np.array([[d2[:, 0].sum(), d2[:, 1].sum()]]) / max(d2.shape[0], 1)
        self.set_vertices(v0 + (dxy - m), gear)
        self.set_offset(off0 + m, gear)

    def apply_affine(self, A, gear, vtx_mask=None):        # mesh.py:2324-2339
        A = np.asarray(A, dtype=np.float64)
        if self.locked or np.all(A == np.eye(3)):
            return
        v0 = self.vertices(gear)
        off0 = self.offset(gear)
        if vtx_mask is None:
            self.set_vertices(v0 @ A[:-1, :-1], gear)
            self.set_offset(off0 @ A[:-1, :-1] + A[-1, :-1], gear)
        else:
            v1 = v0[vtx_mask] @ A[:-1, :-1] + off0 @ A[:-1, :-1] + A[-1, :-1] - off0
            self.set_vertices(v1, gear, vtx_mask=vtx_mask)
            self.set_offset(off0, gear)

    def set_field(self, dxy, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)):   # mesh.py:2400-2413 (unmasked)
        if self.locked:
            return
        if gear[0] == gear[-1]:
            self.apply_field(dxy, gear[0])
            return
        v0 = self.vertices(gear[0])
        off0 = self.offset(gear[0])
        m = np.mean(dxy.reshape(-1, 2), axis=0, keepdims=True)
        self.set_vertices(v0 + (dxy - m), gear[-1])
        self.set_offset(off0 + m, gear[-1])

    def set_translation(self, dxy, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)):   # mesh.py:2289-2300 (unmasked)
        if self.locked:
            return
        if gear[0] == gear[-1]:
            self.apply_translation(dxy, gear[0])
            return
        dxy = np.asarray(dxy, dtype=np.float64).reshape(1, 2)
        self.set_vertices(self.vertices(gear[0]), gear[-1])
        self.set_offset(self.offset(gear[0]) + dxy, gear[-1])

    def estimate_translation(self, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)):   # mesh.py:2309-2321 (unmasked)
        if gear[0] == gear[-1]:
            return np.zeros(2)
        d = self.vertices(gear[-1]).mean(axis=0) - self.vertices(gear[0]).mean(axis=0)
        return d.ravel() + (self.offset(gear[-1]) - self.offset(gear[0])).ravel()

    def set_affine(self, A, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)):      # mesh.py:2342-2354 (unmasked)
        if self.locked:
            return
        A = np.asarray(A, dtype=np.float64)
        if gear[0] == gear[-1]:
            self.apply_affine(A, gear[0])
            return
        v0 = self.vertices(gear[0])
        off0 = self.offset(gear[0])
        self.set_vertices(v0 @ A[:-1, :-1], gear[-1])
        self.set_offset(off0 @ A[:-1, :-1] + A[-1, :-1], gear[-1])

    def connected_vertices(self):                          # mesh.py:1762-1780 (whole mesh)
        """(number of connected components, component label of every vertex)"""
        t = self.triangles
        e = np.concatenate((t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]), axis=0)
        n = self.num_vertices
        adj = sparse.csr_matrix((np.ones(e.shape[0], dtype=bool), (e[:, 0], e[:, 1])), shape=(n, n))
        return csgraph.connected_components(adj, directed=False, return_labels=True)

    def _edge_table(self, tri_mask=None):
        """undirected edges of the (masked) triangles: (edge vertex pairs [3T, 2] sorted within a pair, owning triangle [3T])"""
        tid = np.arange(self.num_triangles) if tri_mask is None else np.flatnonzero(np.asarray(tri_mask)) if np.asarray(tri_mask).dtype == bool else np.asarray(tri_mask)
        t = self.triangles[tid]
        e = np.sort(np.concatenate((t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]), axis=0), axis=1)
        return e, np.tile(tid, 3)

    def connected_triangles(self, tri_mask=None):          # mesh.py:1784-1791: triangles that share an EDGE are adjacent
        """(number of components, component label of every (masked) triangle)"""
        t = self.triangles if tri_mask is None else self.triangles[tri_mask]
        nt = t.shape[0]
        local = np.tile(np.arange(nt), 3)
        a = np.concatenate((t[:, 0], t[:, 1], t[:, 2])).astype(np.int64)
        b = np.concatenate((t[:, 1], t[:, 2], t[:, 0])).astype(np.int64)
        key = np.minimum(a, b) * np.int64(self.num_vertices) + np.maximum(a, b)
        order = np.argsort(key)                                # (equal keys end up next to each other in any order)
        ks, ls = key[order], local[order]
        same = np.flatnonzero(ks[1:] == ks[:-1])
        adj = sparse.csr_matrix((np.ones(same.size, dtype=bool), (ls[same], ls[same + 1])), shape=(nt, nt))
        return csgraph.connected_components(adj, directed=False, return_labels=True)

    def boundary_edges(self, tri_mask=None):
        """edges that belong to exactly one (masked) triangle: the outline of the region (outer rings and holes), what
        shapely_regions (mesh.py:1876-1905) chains into polygons.  [E, 2] vertex ids."""
        if tri_mask is None:
            # the outline of the whole mesh depends on its triangles only (they never change): kept, every round of a block
            # matcher asks for it twice
            if getattr(self, '_outline', None) is None:
                self._outline = self._boundary_edges_of(None)
            return self._outline
        return self._boundary_edges_of(tri_mask)

    def _boundary_edges_of(self, tri_mask):
        """(sorted by (smaller vertex, larger vertex).  One key per edge, sorted once; an edge is on the outline when its key has
        no equal neighbour -- without the index bookkeeping of np.unique: 395 -> 60 ms at 250 k triangles)"""
        t = self.triangles if tri_mask is None else self.triangles[tri_mask]
        a = np.concatenate((t[:, 0], t[:, 1], t[:, 2])).astype(np.int64)
        b = np.concatenate((t[:, 1], t[:, 2], t[:, 0])).astype(np.int64)
        nv = np.int64(self.num_vertices)
        key = np.sort(np.minimum(a, b) * nv + np.maximum(a, b))
        if key.size == 0:
            return np.empty((0, 2), dtype=self.triangles.dtype)
        differs = key[1:] != key[:-1]
        lone = np.concatenate(([True], differs)) & np.concatenate((differs, [True]))
        k = key[lone]
        return np.stack((k // nv, k % nv), axis=-1).astype(self.triangles.dtype)

    def material_stiffness_multiplier(self):
        """per-triangle stiffness multiplier of the triangle's MATERIAL (material.py: Material.stiffness_multiplier)"""
        if self.tri_model is not None:
            return np.asarray(self.tri_matmult, dtype=np.float32)
        return np.full(self.num_triangles, self.material_multiplier, dtype=np.float32)

    def triangle_mask_for_stiffness(self, **kwargs):       # mesh.py:1863-1873
        thr = kwargs.get('stiffness_multiplier_threshold', 0)
        return ~(self.material_stiffness_multiplier() < thr)

    def weight_multiplier_for_render(self):                # mesh.py:1836-1843
        wt = getattr(self, 'tri_render_weight', None)
        return np.ones(self.num_triangles, dtype=np.float32) if wt is None else np.maximum(np.asarray(wt, dtype=np.float32), 0)

    def triangle_mask_for_render(self, **kwargs):          # mesh.py:1847-1859: materials that are not rendered or weigh less than the threshold
        thr = kwargs.get('render_weight_threshold', 0)
        wt = getattr(self, 'tri_render_weight', None)
        if wt is None:
            return np.ones(self.num_triangles, dtype=bool)
        # (a material with render = False is carried as a NEGATIVE weight, -(render_weight + 1): a rendered material of weight 0 stays
        # rendered at threshold 0, like mesh.py:1850-1854)
        return ~(np.asarray(wt) < thr) & (np.asarray(wt) >= 0)

    def submesh(self, tri_mask, **kwargs):                 # mesh.py:598-626
        """the triangles selected by tri_mask (bool mask or index list) with the vertices they use, every gear kept"""
        tri_mask = np.asarray(tri_mask)
        if tri_mask.dtype == bool:
            if tri_mask.all():
                return self
            sel = np.flatnonzero(tri_mask)
        else:
            sel = tri_mask
            if sel.size == self.num_triangles and np.array_equal(np.sort(sel), np.arange(self.num_triangles)):
                return self
        t = self.triangles[sel]
        vidx, inv = np.unique(t, return_inverse=True)
        pick_v = lambda a: None if a is None else a[vidx]
        pick_t = lambda a: None if (a is None or np.ndim(a) == 0) else np.asarray(a)[sel]
        kw = dict(fixed_vertices=pick_v(self._vertices[const.MESH_GEAR_FIXED]), moving_vertices=pick_v(self._vertices[const.MESH_GEAR_MOVING]),
                  staging_vertices=pick_v(self._vertices[const.MESH_GEAR_STAGING]),
                  initial_offset=self._offsets[const.MESH_GEAR_INITIAL], fixed_offset=self._offsets[const.MESH_GEAR_FIXED],
                  moving_offset=self._offsets[const.MESH_GEAR_MOVING], staging_offset=self._offsets[const.MESH_GEAR_STAGING],
                  stiffness_multiplier=pick_t(self._stiffness_multiplier), poisson_ratio=self.poisson_ratio,
                  material_multiplier=self.material_multiplier, resolution=self.resolution, locked=self.locked,
                  soft_factor=self.soft_factor, uid=self.uid)
        if self.tri_model is not None:
            kw.update(tri_model=self.tri_model[sel], tri_nu=self.tri_nu[sel], tri_matmult=self.tri_matmult[sel])
        if self.tri_func is not None:
            kw.update(tri_func=self.tri_func[sel], stiffness_funcs=self.stiffness_funcs, func_matmult=self.func_matmult)
        if self.material_ids is not None:
            kw.update(material_ids=self.material_ids[sel], material_names=self.material_names, material_area_constraints=self.material_area_constraints)
        if getattr(self, 'tri_render_weight', None) is not None:
            kw.update(tri_render_weight=np.asarray(self.tri_render_weight)[sel], material_render_weights=self.material_render_weights)
        kw.update(kwargs)
        return Mesh(self._vertices[const.MESH_GEAR_INITIAL][vidx], inv.reshape(-1, 3), **kw)

    def divide_disconnected_mesh(self, **kwargs):          # mesh.py:689-704
        n, lab = self.connected_triangles()
        if n == 1:
            return [self]
        lbls = np.unique(lab)
        uids = self.uid + 0.5 * (np.arange(lbls.size) + 1) / (10 ** (np.ceil(np.log10(lbls.size + 1))))
        return [self.submesh(lab == lb, uid=float(u), **kwargs) for lb, u in zip(lbls, uids)]

    def anneal(self, gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_FIXED), mode=const.ANNEAL_CONNECTED_RIGID):
        """feabas/mesh.py:2421-2458: move the resting state gear[1] towards gear[0] -- one rigid / affine fit for the
        whole mesh, one per connected component (after the global rigid one), or an exact copy."""
        if self.locked:
            return
        if mode in (const.ANNEAL_GLOBAL_RIGID, const.ANNEAL_GLOBAL_AFFINE):
            v0 = self.vertices_w_offset(gear[0])
            v1 = self.vertices_w_offset(gear[1])
            if mode == const.ANNEAL_GLOBAL_RIGID:
                _, R = common.fit_affine(v0, v1, return_rigid=True)
                self.apply_affine(R, gear[1])
            else:
                self.apply_affine(common.fit_affine(v0, v1, return_rigid=False), gear[1])
        elif mode in (const.ANNEAL_CONNECTED_RIGID, const.ANNEAL_CONNECTED_AFFINE):
            n_conn, v_conn = self.connected_vertices()
            self.anneal(gear=gear, mode=const.ANNEAL_GLOBAL_RIGID)
            if n_conn == 1 and mode == const.ANNEAL_CONNECTED_RIGID:
                return
            v0 = self.vertices_w_offset(gear[0])
            v1 = self.vertices_w_offset(gear[1])
            for cid in range(n_conn):
                idx = v_conn == cid
                if mode == const.ANNEAL_CONNECTED_RIGID:
                    _, R = common.fit_affine(v0[idx], v1[idx], return_rigid=True)
                    self.apply_affine(R, gear[1], vtx_mask=idx)
                else:
                    self.apply_affine(common.fit_affine(v0[idx], v1[idx], return_rigid=False), gear[1], vtx_mask=idx)
        elif mode == const.ANNEAL_COPY_EXACT:
            off0 = self.offset(gear[0])
            v0 = self.vertices(gear[0])
            self.set_vertices(v0, gear[1])
            self.set_offset(off0, gear[1])
        else:
            raise ValueError(mode)

    # ------------------------------------------------------------------ geometry
    def triangle_areas(self, gear=const.MESH_GEAR_INITIAL):   # mesh.py:1753-1758
        if gear == const.MESH_GEAR_INITIAL:
            # the reference caches per gear (config_cache); the INITIAL vertices never change, and every Link of a large
            # section asks for these areas again (optimizer.py:26-30)
            if getattr(self, '_area_initial', None) is None:
                self._area_initial = common.signed_area(self.vertices(gear), self.triangles)
            return self._area_initial
        return common.signed_area(self.vertices(gear), self.triangles)

    def triangle_area_deform(self, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_MOVING)):   # mesh.py:1979-1986
        return common.signed_area(self.vertices(gear[-1]), self.triangles) / common.signed_area(self.vertices(gear[0]), self.triangles)

    def triangle_edge_deform(self, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_MOVING)):   # mesh.py:1966-1976
        v0, v1 = self.vertices(gear[0]), self.vertices(gear[-1])
        t = self.triangles
        if t.shape[0] >= 1024 and t.dtype == np.int32 and v0.dtype == np.float64 and v1.dtype == np.float64:
            # the six vertex gathers in a host loop of the library (the same roundings); log / exp stay numpy's
            a0, a1, tt = np.ascontiguousarray(v0), np.ascontiguousarray(v1), np.ascontiguousarray(t)
            ratio = np.empty((tt.shape[0], 3))
            if _lib.load().fb_tri_edge_ratio(None, a0.shape[0], _lib.ptr(a0), _lib.ptr(a1), tt.shape[0], _lib.ptr(tt), _lib.ptr(ratio)) != 0:
                raise IndexError('triangle_edge_deform: a triangle names a vertex outside the vertex list')
            return np.exp(np.max(np.abs(0.5 * np.log(ratio)), axis=-1))
        tr = np.roll(t, 1, axis=-1)
        d0 = np.sum((v0[t] - v0[tr]) ** 2, axis=-1)
        d1 = np.sum((v1[t] - v1[tr]) ** 2, axis=-1)
        return np.exp(np.max(np.abs(0.5 * np.log(d1 / d0)), axis=-1))

    @staticmethod
    def svds_to_deform(s):                                 # mesh.py:3358-3365
        """(N, k) singular values -> deformation: 0..1 while not flipped, >= 1 when flipped"""
        s = np.asarray(s, dtype=np.float64)
        d = np.where(s < 1, 1 - s, 1 - 1 / np.where(s == 0, 1, s))
        return np.max(d, axis=-1)

    def effective_stiffness_multiplier(self, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_MOVING)):   # mesh.py:1600-1621
        if self.tri_model is None:
            return self.stiffness_multiplier * np.float32(self.material_multiplier)
        if self.tri_func is None:
            return self.stiffness_multiplier * self.tri_matmult
        # materials with a stiffness function: x f(area stretch / its median over the linear triangles)
        J = self.triangle_area_deform(gear=gear)
        lin = self.linear_triangle_mask
        J = J / (np.median(J[lin]) if np.any(lin) else np.median(J))
        modifier = self.tri_matmult.astype(np.float64)
        for k, f in enumerate(self.stiffness_funcs):
            sel = self.tri_func == k
            modifier[sel] = self.func_matmult[k] * f(J[sel])
        return self.stiffness_multiplier * modifier

    def tri_finder(self, pts, gear=None, render_weight_threshold=0, **kwargs):
        """point -> triangle id (-1 outside).  The reference goes through
        matplotlib's trapezoid-map finder per connected region (mesh.py:2080-2188);
        a single finder over the whole mesh covers the non-overlapping meshes used here.
        render_weight_threshold > 0: a point in a triangle whose material weighs no more than that counts as outside
        (mesh.py:2168-2170 -- what keeps matches out of soft / wrinkled regions, optimizer.py:59)."""
        tid = self._tri_finder(pts, gear)
        if render_weight_threshold > 0 and getattr(self, 'tri_render_weight', None) is not None:
            tid = np.array(tid, dtype=np.int32)
            hit = tid >= 0
            tid[hit] = np.where(self.tri_render_weight[tid[hit]] <= render_weight_threshold, -1, tid[hit])
        return tid

    def _tri_finder(self, pts, gear=None):
        from matplotlib.tri import Triangulation
        gear = self._current_gear if gear is None else gear
        if gear == const.MESH_GEAR_INITIAL and getattr(self, 'grid_xs', None) is not None:
            # the undeformed grid of from_bbox(cartesian=True): the cell follows from the coordinates
            p = (np.asarray(pts, dtype=np.float64) - self.offset(gear)).reshape(-1, 2)
            xs, ys = self.grid_xs, self.grid_ys
            inside = (p[:, 0] >= xs[0]) & (p[:, 0] <= xs[-1]) & (p[:, 1] >= ys[0]) & (p[:, 1] <= ys[-1])
            return np.where(inside, self.locate_cartesian(p), -1).astype(np.int32)
        if _lib._ctx is not None and self.num_triangles > 0:
            # a device context exists: brute-force location on the GPU (fb_mesh_locate_dev) instead of rebuilding matplotlib's
            # trapezoid map every time the gear moved
            p = np.ascontiguousarray((np.asarray(pts, dtype=np.float64) - self.offset(gear)).reshape(-1, 2))
            lib, ctx = _lib.load(), _lib.ctx()
            bufs = [_lib.DeviceBuffer.from_array(np.ascontiguousarray(self.vertices(gear), dtype=np.float64)),
                    _lib.DeviceBuffer.from_array(self.triangles), _lib.DeviceBuffer.from_array(p), _lib.DeviceBuffer(4 * max(1, p.shape[0]))]
            try:
                _lib.check(lib.fb_mesh_locate_dev(ctx, self.num_triangles, bufs[0].ptr, bufs[1].ptr, p.shape[0], bufs[2].ptr, bufs[3].ptr))
                return bufs[3].to_array((p.shape[0],), np.int32)
            finally:
                for b in bufs:
                    b.free()
        if gear not in self._trifinders:
            v = self.vertices(gear)
            self._trifinders[gear] = Triangulation(v[:, 0], v[:, 1], self.triangles).get_trifinder()
        p = (np.asarray(pts, dtype=np.float64) - self.offset(gear)).reshape(-1, 2)
        return np.asarray(self._trifinders[gear](p[:, 0], p[:, 1]), dtype=np.int32)

    def cart2bary(self, xy, gear, tid=None, **kwargs):     # mesh.py:2191-2217
        xy = np.atleast_2d(xy)
        if tid is None:
            tid = self.tri_finder(xy, gear=gear, render_weight_threshold=kwargs.get('render_weight_threshold', 0))
        inside = tid >= 0
        if not np.any(inside):
            return tid, np.full((tid.size, 3), np.nan, dtype=np.float32)
        q = xy[inside, :] - self.offset(gear)
        p = self.vertices(gear)[np.atleast_2d(self.triangles[tid[inside], :])]
        d0, d1, d2 = q - p[:, 0, :], q - p[:, 1, :], q - p[:, 2, :]
        a0, a1, a2 = common.cross2d(d1, d2), common.cross2d(d2, d0), common.cross2d(d0, d1)
        tot = a0 + a1 + a2
        bary = np.stack((a0 / tot, a1 / tot, a2 / tot), axis=-1)
        if np.all(inside):
            return tid, bary
        full = np.full((tid.size, 3), np.nan, dtype=bary.dtype)
        full[inside, :] = bary
        return tid, full

    def bary2cart(self, tid, B, gear, offsetting=True):    # mesh.py:2220-2228
        idx = np.atleast_2d(self.triangles[tid, :])
        v = self.vertices(gear)
        B = np.asarray(B, dtype=np.float64).reshape(-1, 3)
        # sum_k B_k v_k term by term (the offset of the gear is added to the K results, not to all V vertices first)
        out = v[idx[:, 0]] * B[:, 0:1]
        out += v[idx[:, 1]] * B[:, 1:2]
        out += v[idx[:, 2]] * B[:, 2:3]
        if offsetting:
            out += self.offset(gear)
        return out

    # ------------------------------------------------------------------ stiffness (GPU)
    def element_multiplier(self):
        """per-triangle float32 multiplier: mesh multiplier x material multiplier (material.py:168-171)."""
        m = np.full(self.num_triangles, self.material_multiplier, dtype=np.float32)
        if self._stiffness_multiplier is not None:
            m = self._stiffness_multiplier * m
        return np.ascontiguousarray(m, dtype=np.float32)

    def assemble_into(self, sysh, mesh_id, v_shape, v_cur, soft, add=False):
        """numeric assembly of this mesh's stiffness rows into a GPU system (add: on top of the rows' current content,
        for a mesh that shares its degrees of freedom with an earlier member of its group)"""
        lib, ctx = _lib.load(), _lib.ctx()
        if self.tri_model is None:
            fn = lib.fb_sys_assemble_mesh_add if add else lib.fb_sys_assemble_mesh
            _lib.check(fn(ctx, sysh, mesh_id, _lib.ptr(v_shape), _lib.ptr(v_cur),
                          _lib.ptr(self.element_multiplier()), self.poisson_ratio, float(soft)))
        elif add:
            raise NotImplementedError('grouped meshes with non-linear materials')
        elif self.tri_func is not None and not all(isinstance(f, StiffnessTable) for f in self.stiffness_funcs):
            # a stiffness function that is not a table (any Python callable, material.py:128-131): evaluated on the host on the
            # area stretch INITIAL -> current of its triangles (mesh.py:2937-2971) and handed over as the per-triangle material multiplier
            v_init = np.ascontiguousarray(self.vertices(const.MESH_GEAR_INITIAL), dtype=np.float64)
            a1 = common.signed_area(v_shape if v_cur is None else v_cur, self.triangles); a0 = common.signed_area(v_init, self.triangles)
            lin = self.linear_triangle_mask
            # mesh.py:2952-2963, 3030-3040: the stretch of a triangle relative to the ratio of the summed |areas| of the linear triangles
            base = (np.sum(np.abs(a1[lin])) / np.sum(np.abs(a0[lin]))) if np.any(lin) else (np.sum(np.abs(a1)) / np.sum(np.abs(a0)))
            J = (a1 / a0) / base
            matmult = self.tri_matmult.astype(np.float64)
            for k, f in enumerate(self.stiffness_funcs):
                sel = self.tri_func == k
                matmult[sel] = self.func_matmult[k] * np.asarray(f(J[sel]), dtype=np.float64).ravel()
            _lib.check(lib.fb_sys_assemble_mesh_materials(ctx, sysh, mesh_id, _lib.ptr(v_shape), _lib.ptr(v_cur),
                                                          _lib.ptr(np.ascontiguousarray(self.stiffness_multiplier, dtype=np.float32)),
                                                          _lib.ptr(self.tri_model), _lib.ptr(self.tri_nu),
                                                          _lib.ptr(np.ascontiguousarray(matmult, dtype=np.float32)), float(soft)))
        elif self.tri_func is not None:
            # stiffness follows the area stretch INITIAL -> current gear (mesh.py:2937-2971, 3026-3043)
            ptr = np.concatenate(([0], np.cumsum([f.strain.size for f in self.stiffness_funcs]))).astype(np.int32)
            fx = np.concatenate([f.strain for f in self.stiffness_funcs]); fy = np.concatenate([f.stiffness for f in self.stiffness_funcs])
            v_init = np.ascontiguousarray(self.vertices(const.MESH_GEAR_INITIAL), dtype=np.float64)
            _lib.check(lib.fb_sys_assemble_mesh_stretch(ctx, sysh, mesh_id, _lib.ptr(v_shape), _lib.ptr(v_cur), _lib.ptr(v_init),
                                                        _lib.ptr(np.ascontiguousarray(self.stiffness_multiplier, dtype=np.float32)),
                                                        _lib.ptr(self.tri_model), _lib.ptr(self.tri_nu), _lib.ptr(self.tri_matmult),
                                                        _lib.ptr(self.tri_func), len(self.stiffness_funcs), _lib.ptr(ptr), _lib.ptr(fx), _lib.ptr(fy),
                                                        _lib.ptr(self.func_matmult), float(soft)))
        else:
            _lib.check(lib.fb_sys_assemble_mesh_materials(ctx, sysh, mesh_id, _lib.ptr(v_shape), _lib.ptr(v_cur),
                                                          _lib.ptr(np.ascontiguousarray(self.stiffness_multiplier, dtype=np.float32)),
                                                          _lib.ptr(self.tri_model), _lib.ptr(self.tri_nu), _lib.ptr(self.tri_matmult),
                                                          float(soft)))

    def stiffness_matrix(self, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING), **kwargs):
        """feabas/mesh.py:3058-3083 -> (scipy CSR 2V x 2V float64, stress float32),
        assembled by the HIP kernel on a one-mesh system."""
        lib = _lib.load()
        ctx = _lib.ctx()
        sysh = C.c_void_p()
        _lib.check(lib.fb_sys_create(ctx, self.num_vertices, C.byref(sysh)))
        try:
            mid = C.c_int()
            _lib.check(lib.fb_sys_add_mesh(ctx, sysh, 0, _lib.ptr(self.triangles), self.num_vertices,
                                           self.num_triangles, C.byref(mid)))
            _lib.check(lib.fb_sys_set_links(ctx, sysh, 0, None))
            nnzb = C.c_int64()
            _lib.check(lib.fb_sys_finalize(ctx, sysh, C.byref(nnzb)))
            v0 = np.ascontiguousarray(self.vertices(gear[0]), dtype=np.float64)
            v1 = self.vertices(gear[-1])
            v1c = None if v1 is v0 or v1 is self.vertices(gear[0]) else np.ascontiguousarray(v1, dtype=np.float64)
            self.assemble_into(sysh, mid.value, v0, v1c, 1.0)
            K = bsr_download(sysh, 0, self.num_vertices, nnzb.value)
            stress = np.empty(2 * self.num_vertices, dtype=np.float32)
            _lib.check(lib.fb_sys_get(ctx, sysh, 3, _lib.ptr(stress)))
        finally:
            lib.fb_sys_destroy(ctx, sysh)
        return K, stress

    def stiffness_energy(self, fields, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING)):
        """x^T K x for every displacement field x (V x 2) of ``fields``, K = this mesh's stiffness matrix at ``gear``: assembled
        and contracted on the device (fb_sys_group_energy) -- the two energies of the strain estimate, matcher.py:764-777"""
        lib, ctx = _lib.load(), _lib.ctx()
        sysh = C.c_void_p()
        _lib.check(lib.fb_sys_create(ctx, self.num_vertices, C.byref(sysh)))
        try:
            mid = C.c_int()
            _lib.check(lib.fb_sys_add_mesh(ctx, sysh, 0, _lib.ptr(self.triangles), self.num_vertices, self.num_triangles, C.byref(mid)))
            _lib.check(lib.fb_sys_set_links(ctx, sysh, 0, None))
            nnzb = C.c_int64()
            _lib.check(lib.fb_sys_finalize(ctx, sysh, C.byref(nnzb)))
            v0 = np.ascontiguousarray(self.vertices(gear[0]), dtype=np.float64)
            v1 = self.vertices(gear[-1])
            self.assemble_into(sysh, mid.value, v0, None if v1 is self.vertices(gear[0]) else np.ascontiguousarray(v1, dtype=np.float64), 1.0)
            out = []
            e = C.c_double()
            for x in fields:
                x = np.ascontiguousarray(x, dtype=np.float64)
                if x.shape != (self.num_vertices, 2):
                    raise ValueError('a field is V x 2')
                _lib.check(lib.fb_sys_group_energy(ctx, sysh, 1, _lib.ptr(x), C.byref(e)))
                out.append(float(e.value))
        finally:
            lib.fb_sys_destroy(ctx, sysh)
        return out

    def stiffness_matrix_local_normalized(self, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING), tri_mask=None, **kwargs):
        """feabas/mesh.py:3086-3129 -> (scipy CSR, stress float32) of the sub-mesh ``tri_mask``: every triangle as the
        default linear material (Poisson ratio 0, D = m diag(1, 1, 1/2)) with its effective multiplier clipped at
        max / max_stiffness_ratio; shape matrix at gear[0], stress = K (v[gear1] - v[gear0]).  Assembled by the same HIP
        kernel as ``stiffness_matrix`` -- the triangles outside the mask enter with multiplier 0."""
        max_stiffness_ratio = kwargs.get('max_stiffness_ratio', 1000)
        tidx = np.arange(self.num_triangles)
        if tri_mask is not None:
            tidx = tidx[tri_mask]
        if tidx.size == 0:
            return None, None
        mm = np.asarray(self.effective_stiffness_multiplier(), dtype=np.float32)[tidx]
        if max_stiffness_ratio is not None:
            mn = np.max(mm) / max_stiffness_ratio
            if mn == 0:
                mn = 1
            mm = mm.clip(mn, None)
        mult = np.zeros(self.num_triangles, dtype=np.float32)
        mult[tidx] = mm
        lib, ctx = _lib.load(), _lib.ctx()
        sysh = C.c_void_p()
        _lib.check(lib.fb_sys_create(ctx, self.num_vertices, C.byref(sysh)))
        try:
            mid = C.c_int()
            _lib.check(lib.fb_sys_add_mesh(ctx, sysh, 0, _lib.ptr(self.triangles), self.num_vertices, self.num_triangles, C.byref(mid)))
            _lib.check(lib.fb_sys_set_links(ctx, sysh, 0, None))
            nnzb = C.c_int64()
            _lib.check(lib.fb_sys_finalize(ctx, sysh, C.byref(nnzb)))
            v0 = np.ascontiguousarray(self.vertices(gear[0]), dtype=np.float64)
            v1 = np.ascontiguousarray(self.vertices(gear[-1]), dtype=np.float64)
            _lib.check(lib.fb_sys_assemble_mesh(ctx, sysh, mid.value, _lib.ptr(v0), _lib.ptr(v1), _lib.ptr(mult), 0.0, 1.0))
            K = bsr_download(sysh, 0, self.num_vertices, nnzb.value)
            stress = np.empty(2 * self.num_vertices, dtype=np.float32)
            _lib.check(lib.fb_sys_get(ctx, sysh, 3, _lib.ptr(stress)))
        finally:
            lib.fb_sys_destroy(ctx, sysh)
        return K, stress


def bsr_download(sysh, which, nv, nnzb):
    """pattern + 2x2 block values of a system matrix -> scipy CSR (2nv x 2nv)."""
    lib = _lib.load()
    ctx = _lib.ctx()
    rowptr = np.empty(nv + 1, dtype=np.int64)
    col = np.empty(nnzb, dtype=np.int32)
    _lib.check(lib.fb_sys_pattern(ctx, sysh, _lib.ptr(rowptr), _lib.ptr(col)))
    if which == 1:
        c = np.empty(nnzb, dtype=np.float32)
        _lib.check(lib.fb_sys_get(ctx, sysh, 1, _lib.ptr(c)))
        val = np.zeros((nnzb, 2, 2), dtype=np.float32)
        val[:, 0, 0] = c
        val[:, 1, 1] = c
    else:
        val = np.empty((nnzb, 2, 2), dtype=np.float64)
        _lib.check(lib.fb_sys_get(ctx, sysh, which, _lib.ptr(val)))
    M = sparse.bsr_matrix((val, col, rowptr), shape=(2 * nv, 2 * nv)).tocsr()
    M.eliminate_zeros()
    return M
