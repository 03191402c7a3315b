"""Host mirror of feabas.optimizer for the FEM path: ``Link``, ``SLM`` and
``solve`` keep the reference's call surface; assembly and the iterative solve
run on the GPU through the fb_sys_* / fb_pcg entry points.

What is deliberately different from the reference: the solver iterates
Jacobi-PCG to the reference's stopping rule (true residual <= tol*||b||)
instead of restarted MINRES, and none of the reference's wall-clock /
random-perturbation early exits (optimizer.py:1955-1961, 2056-2063) exist, so
results are deterministic.
"""
import ctypes as C
import os

import numpy as np
from scipy import sparse

from . import _lib, common
from . import constant as const
from .mesh import Mesh, bsr_download


class Link:
    """feabas/optimizer.py:17-435 (the part used on the hot path): matched
    points stored as (triangle id, barycentric) on two meshes."""

    def __init__(self, mesh0, mesh1, tid0, tid1, B0, B1, weight=None, **kwargs):
        self.strain = kwargs.get('strain', const.DEFAULT_AVG_DEFORM)
        self.meshes = [mesh0, mesh1]
        self.uids = [mesh0.uid, mesh1.uid]
        self.name = kwargs.get('name', '_'.join(str(s) for s in self.uids))
        self._tid0 = np.asarray(tid0)
        self._tid1 = np.asarray(tid1)
        self._B0 = np.asarray(B0, dtype=np.float64)
        self._B1 = np.asarray(B1, dtype=np.float64)
        se = kwargs.get('sample_err', None)
        if se is None:                                      # optimizer.py:26-30
            a0 = mesh0.triangle_areas(gear=const.MESH_GEAR_INITIAL)[self._tid0]
            a1 = mesh1.triangle_areas(gear=const.MESH_GEAR_INITIAL)[self._tid1]
            se = 0.4387 * (np.minimum(a0, a1)) ** 0.5 * self.strain
        self._sample_err = se
        self._weight = ((self._tid0 >= 0) & (self._tid1 >= 0)).astype(np.float32)
        if weight is not None:
            self._weight = self._weight * weight
        self._residue_weight = np.ones_like(self._weight)
        self._weight_func = None
        self._mask = None
        self._disabled = False

    @classmethod
    def from_coordinates(cls, mesh0, mesh1, xy0, xy1, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_INITIAL),
                         weight=None, **kwargs):
        """optimizer.py:53-82."""
        xy0 = np.asarray(xy0)
        xy1 = np.asarray(xy1)
        if xy0.size == 0:
            return None, None
        # (optimizer.py:59: matches in triangles of a material that weighs no more than 0.1 in rendering -- soft, split, wrinkled
        # regions of the default material table -- are dropped unless the caller asks otherwise)
        rwt = kwargs.pop('render_weight_threshold', 0.1)
        tid0, B0 = mesh0.cart2bary(xy0, gear[0], tid=None, render_weight_threshold=rwt)
        in0 = tid0 >= 0
        if not np.any(in0):
            return None, None
        if not np.all(in0):
            tid0, B0, xy1 = tid0[in0], B0[in0], xy1[in0]
            if isinstance(weight, np.ndarray):
                weight = weight[in0]
        tid1, B1 = mesh1.cart2bary(xy1, gear[1], tid=None, render_weight_threshold=rwt)
        in1 = tid1 >= 0
        if not np.any(in1):
            return None, None
        if not np.all(in1):
            tid0, tid1, B0, B1 = tid0[in1], tid1[in1], B0[in1], B1[in1]
            if isinstance(weight, np.ndarray):
                weight = weight[in1]
            in0[in0] = in1
        kwargs.pop('check_duplicates', None)
        return cls(mesh0, mesh1, tid0, tid1, B0, B1, weight=weight, **kwargs), in0

    # --- views
    @property
    def mask(self):
        if self._mask is None:
            self._mask = (self._weight * self._residue_weight) > 0
        return self._mask

    def _sel(self, arr, use_mask):
        return arr[self.mask] if use_mask else arr

    def tid0(self, use_mask=False):
        return self._sel(self._tid0, use_mask)

    def tid1(self, use_mask=False):
        return self._sel(self._tid1, use_mask)

    def B0(self, use_mask=False):
        return self._sel(self._B0, use_mask)

    def B1(self, use_mask=False):
        return self._sel(self._B1, use_mask)

    def weight(self, use_mask=False):
        return self._sel(self._weight * self._residue_weight, use_mask)

    @property
    def sample_err(self):
        se = np.array(self._sample_err)
        if se.size == 1:
            se = np.full(self._tid0.size, se)
        return se

    @property
    def locked(self):
        return [self.meshes[0].locked, self.meshes[1].locked]

    @property
    def relevant(self):
        return not (self._disabled or np.all(self.locked))

    @property
    def num_matches(self):
        return 0 if self._disabled else int(np.sum(self.mask))

    def xy0(self, gear=const.MESH_GEAR_MOVING, use_mask=True, combine=True):
        xy = self.meshes[0].bary2cart(self.tid0(use_mask), self.B0(use_mask), gear, offsetting=False)
        off = self.meshes[0].offset(gear)
        return xy + off if combine else (xy, off)

    def xy1(self, gear=const.MESH_GEAR_MOVING, use_mask=True, combine=True):
        xy = self.meshes[1].bary2cart(self.tid1(use_mask), self.B1(use_mask), gear, offsetting=False)
        off = self.meshes[1].offset(gear)
        return xy + off if combine else (xy, off)

    def dxy(self, gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_MOVING), use_mask=False):
        """optimizer.py:248-255."""
        if not hasattr(gear, '__len__'):
            gear = (gear, gear)
        p0, o0 = self.xy0(gear=gear[0], use_mask=use_mask, combine=False)
        p1, o1 = self.xy1(gear=gear[1], use_mask=use_mask, combine=False)
        return (p1 - p0) + (o1 - o0)

    # --- residue weighting (optimizer.py:174-205)
    def set_hard_residue_filter(self, residue_len):
        self._weight_func = lambda x: x <= residue_len

    def set_huber_residue_filter(self, residue_len):
        self._weight_func = lambda x: residue_len / np.maximum(x, residue_len)

    def adjust_weight_from_residue(self, gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_MOVING)):
        if self._weight_func is None:
            return False, False
        prev_w = self._residue_weight
        prev_conn = self.num_matches > 0
        d = self.dxy(gear=gear, use_mask=False)
        dis = np.sum(d ** 2, axis=-1) ** 0.5
        dis = ((dis ** 2 - self.sample_err ** 2).clip(0, None)) ** 0.5
        new_w = np.asarray(self._weight_func(dis)).astype(np.float32)
        if np.any(new_w != prev_w):
            self._residue_weight = new_w
            self._mask = None
            return True, (self.num_matches > 0) != prev_conn
        return False, False

    def disable(self):                                     # optimizer.py:208-213
        self._disabled = True

    def enable(self):
        self._disabled = False

    @property
    def weight_sum(self):                                  # optimizer.py:391-395
        return 0 if self._disabled else np.sum(self.weight(use_mask=True))

    def reset_mask(self):
        self._residue_weight = np.ones_like(self._weight)
        self._mask = None


class SLM:
    """feabas/optimizer.py:487-1873, linear-elastic relaxation subset:
    ``optimize_linear`` / ``optimize_elastic`` on linear engineering meshes."""

    def __init__(self, meshes, links=None, **kwargs):
        self.meshes = list(meshes)
        self.links = [] if links is None else links
        self._stiffness_lambda = kwargs.get('stiffness_lambda', 1.0)
        self._crosslink_lambda = kwargs.get('crosslink_lambda', -1.0)
        self._sys = None
        self._sys_key = None
        self.last_solve = None

    def __del__(self):
        self._drop_system()

    def _drop_system(self):
        if getattr(self, '_sys', None) is not None:
            try:
                _lib.load().fb_sys_destroy(_lib.ctx(), self._sys)
            except Exception:
                pass
            self._sys = None
            self._sys_key = None

    # ------------------------------------------------------------------ system manipulation
    def add_link(self, link, **kwargs):
        if link is None:
            return False
        self.links.append(link)
        return True

    def add_link_from_coordinates(self, uid0, uid1, xy0, xy1, gear=(const.MESH_GEAR_INITIAL, const.MESH_GEAR_INITIAL),
                                  weight=None, submesh_exclusive=True, check_duplicates=True, **kwargs):
        """optimizer.py:637-684: matches given by coordinates, between the meshes that carry the two uids -- or, for the uid of a mesh
        that has been cut into its connected parts, its parts: the matches are dealt to the part pairs that hold them (every match
        once with submesh_exclusive).  A link whose ``name`` is loaded already is skipped (check_duplicates); a uid the system does not
        know adds nothing.  Other keywords go to Link.from_coordinates (render_weight_threshold, strain, name)."""
        if check_duplicates and ('name' in kwargs) and any(lk.name == kwargs['name'] for lk in self.links):
            return False
        side0, _ = self.select_mesh_from_uid(uid0)
        side1, _ = self.select_mesh_from_uid(uid1)
        if len(side0) == 0 or len(side1) == 0:
            return False
        xy0, xy1 = np.asarray(xy0), np.asarray(xy1)
        added = False
        for m0 in side0:
            for m1 in side1:
                link, took = Link.from_coordinates(m0, m1, xy0, xy1, gear=gear, weight=weight, **dict(kwargs))
                if link is None:
                    continue
                self.links.append(link)
                added = True
                if submesh_exclusive:
                    xy0, xy1 = xy0[~took], xy1[~took]
                    if isinstance(weight, np.ndarray):
                        weight = weight[~took]
        return added

    # ------------------------------------------------------------------ sub-meshes (optimizer.py:688-754, 1778-1858)
    def select_mesh_from_uid(self, uid):
        """meshes that carry `uid`, or -- for the uid of a mesh that has been cut into parts -- its parts (uids within 0.5
        above the parent's); the flag tells whether the match is exact and unique (optimizer.py:1798-1816)"""
        exact = [m for m in self.meshes if m.uid == uid]
        if exact:
            return exact, len(exact) == 1
        parts = [m for m in self.meshes if np.floor(m.uid) == np.floor(uid)]
        return parts, False

    def link_is_relevant(self, link):                      # optimizer.py:1778-1795
        if link is None or not link.relevant:
            return 0
        for lid in link.uids:
            sel, exact = self.select_mesh_from_uid(lid)
            if len(sel) == 0:
                return 0
            if not exact:
                return -1
        return 1

    @staticmethod
    def distribute_link(mesh0_list, mesh1_list, link, exclusive=True, working_gear=const.MESH_GEAR_INITIAL, **kwargs):
        """optimizer.py:1818-1858: the matches of one link dealt to the pairs of parts that contain them"""
        xy0 = link.xy0(gear=working_gear, use_mask=False, combine=True)
        xy1 = link.xy1(gear=working_gear, use_mask=False, combine=True)
        weight = link.weight(use_mask=False)
        out = []
        for m0 in mesh0_list:
            for m1 in mesh1_list:
                if xy0.shape[0] == 0:
                    break
                lnk, mask = Link.from_coordinates(m0, m1, xy0, xy1, gear=(working_gear, working_gear), weight=weight, strain=link.strain)
                if lnk is None:
                    continue
                lnk._weight_func = link._weight_func
                out.append(lnk)
                if exclusive:
                    xy0, xy1, weight = xy0[~mask], xy1[~mask], weight[~mask]
        return out

    def prune_links(self, **kwargs):
        """optimizer.py:688-717: links that still name whole meshes stay, links an end of which was cut into parts
        (uid 3 -> 3.0, 3.01, ...) are dealt to the pairs of parts that hold their matches (distribute_link), links to meshes
        that are gone are dropped.  Returns whether the list changed."""
        gear = kwargs.get('working_gear', const.MESH_GEAR_INITIAL)
        exclusive = kwargs.get('submesh_exclusive', True)
        kept, changed = [], False
        for lk in self.links:
            state = self.link_is_relevant(lk)
            if state == 1:
                kept.append(lk)
                continue
            changed = True
            if state == -1:
                ends = [self.select_mesh_from_uid(u)[0] for u in lk.uids]
                kept += SLM.distribute_link(ends[0], ends[1], lk, working_gear=gear, exclusive=exclusive)
        if changed:
            self.links = kept
        return changed

    def divide_disconnected_submeshes(self, prune_links=True, **kwargs):
        """optimizer.py:738-754: every free mesh that falls into several connected parts is replaced by its parts (locked
        meshes stay whole); the links follow unless prune_links is False.  Returns whether a mesh was divided."""
        pieces = [[m] if m.locked else list(m.divide_disconnected_mesh()) for m in self.meshes]
        if max((len(p) for p in pieces), default=1) <= 1:
            return False
        self.meshes = [part for p in pieces for part in p]
        self._drop_system()
        if prune_links:
            self.prune_links(**kwargs)
        return True

    def clear_links(self):
        self.links = []

    def set_link_residue_huber(self, residue_len):
        for lk in self.links:
            lk.set_huber_residue_filter(residue_len)

    def set_link_residue_threshold(self, residue_len):
        for lk in self.links:
            lk.set_hard_residue_filter(residue_len)

    def relax_higly_deformed(self, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING), deform_cutoff=const.MAXIMUM_DEFORM_ALLOWED, iqr=0):
        """optimizer.py:763-772 (name as in the reference): relax the most deformed region of every free mesh; the
        cutoff handed down is the already converted threshold, as the reference does"""
        modified = 0
        deform_thresh = 1 - 1 / (abs(deform_cutoff) + 1)
        for m in self.meshes:
            if m.locked:
                continue
            modified = modified + relax_mesh_most_deformed(m, gear=gear, deform_cutoff=deform_thresh, iqr=iqr)
        return modified

    def adjust_link_weight_by_residue(self, gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_MOVING), relax_first=False, **kwargs):
        wm = cm = False
        if relax_first:
            self.relax_higly_deformed()
        for lk in self.links:
            a, b = lk.adjust_weight_from_residue(gear=gear)
            wm |= a
            cm |= b
        return wm, cm

    def anneal(self, gear=(const.MESH_GEAR_MOVING, const.MESH_GEAR_FIXED), mode=const.ANNEAL_CONNECTED_RIGID):
        """optimizer.py:757-760"""
        for m in self.meshes:
            m.anneal(gear=gear, mode=mode)

    def linkage_adjacency(self, directional=False):
        """optimizer.py:1678-1695: mesh x mesh matrix of summed link weights."""
        from scipy import sparse
        uid2idx = {m.uid: k for k, m in enumerate(self.meshes)}
        rows, cols, vals = [], [], []
        for lk in self.links:
            a, b = uid2idx.get(lk.uids[0], -1), uid2idx.get(lk.uids[1], -1)
            if a < 0 or b < 0:
                continue
            rows.append(a); cols.append(b); vals.append(float(np.sum(lk.weight(use_mask=True))))
        n = len(self.meshes)
        A = sparse.csr_matrix((vals, (rows, cols)), shape=(n, n))
        if not directional:
            A = A + A.transpose()
        A.eliminate_zeros()
        return A

    @property
    def connected_subsystems(self):
        """optimizer.py:1698-1703: (labels, count) of the link-connected components."""
        from scipy.sparse import csgraph
        n, labels = csgraph.connected_components(self.linkage_adjacency(), directed=False, return_labels=True)
        return labels, n

    def flag_outcasts(self):
        """optimizer.py:1604-1625 (what the aligner asks before it optimises a window, aligner.py:700): with several link-connected
        subsystems, the meshes that are not connected to a locked mesh -- or, when nothing is locked and nothing was flagged before,
        those outside the largest subsystem -- are outcasts; the flag is kept on the mesh (``Mesh.is_outcast``).  One subsystem: the
        flags as they stand."""
        before = [bool(getattr(m, 'is_outcast', False)) for m in self.meshes]
        labels, n = self.connected_subsystems
        if n == 1:
            return before
        locks = np.asarray(self.lock_flags, dtype=bool)
        if np.any(before) or np.any(locks):
            outcasts = ~np.isin(labels, labels[locks])
        else:
            u, cnt = np.unique(labels, return_counts=True)
            outcasts = labels != u[np.argmax(cnt)]
        for m, flag in zip(self.meshes, outcasts):
            m.is_outcast = bool(flag)
        return outcasts

    def match_residues(self, gear=const.MESH_GEAR_MOVING, use_mask=False, quantile=0.75):
        """optimizer.py:1758-1774."""
        dis = []
        for lk in self.links:
            if use_mask and not lk.relevant:
                dis.append(np.nan)
                continue
            d = np.sum(lk.dxy(gear=gear, use_mask=use_mask) ** 2, axis=-1) ** 0.5
            if d.size == 0:
                dis.append(np.nan)
            elif quantile == 1:
                dis.append(np.max(d))
            elif quantile == 0:
                dis.append(np.min(d))
            else:
                dis.append(np.quantile(d, quantile))
        return np.array(dis)

    def optimize_translation_lsqr(self, **kwargs):
        """optimizer.py:974-1078: one translation per free mesh from the weighted medians of the link residuals, least
        squares (scipy lsqr, host -- SURVEY.md b13: #tiles unknowns, stays on the host like in the reference).  Returns
        (cost, per-link residue) and writes the translations to target_gear."""
        from scipy.sparse.linalg import lsqr
        maxiter = kwargs.get('maxiter', None)
        tol = kwargs.get('tol', 1e-07)
        start_gear = kwargs.get('start_gear', const.MESH_GEAR_FIXED)
        target_gear = kwargs.get('target_gear', const.MESH_GEAR_FIXED)
        return_residue = kwargs.get('return_residue', True)
        locked = self.lock_flags
        free_idx = np.nonzero(~locked)[0]
        nl = len(self.links)
        if nl == 0 or free_idx.size == 0:
            return None, None
        col_of = {self.meshes[k].uid: c for c, k in enumerate(free_idx)}
        known = {m.uid for m in self.meshes}
        labels, _ = self.connected_subsystems
        floating = sorted(set(labels[~locked]).difference(set(labels[locked])))       # components without a locked mesh
        A = sparse.lil_matrix((nl + len(floating), free_idx.size))
        bx = np.zeros(nl + len(floating)); by = np.zeros(nl + len(floating))
        rel = np.zeros(nl, dtype=np.float32)
        row = 0
        for k, lk in enumerate(self.links):
            if (not lk.relevant) or (lk.uids[0] not in known) or (lk.uids[1] not in known):
                continue
            wt = lk.weight_sum ** 0.5
            if wt == 0:
                continue
            gears = []
            for side, sign in ((0, 1.0), (1, -1.0)):
                if lk.uids[side] in col_of:
                    A[row, col_of[lk.uids[side]]] = sign * wt
                    gears.append(start_gear)
                else:
                    gears.append(target_gear)
            d = np.median(lk.dxy(gear=gears, use_mask=True), axis=0)
            bx[row] = d[0] * wt; by[row] = d[1] * wt
            rel[k] = wt
            row += 1
        if row == 0:
            return None, None
        wt = (A.power(2).sum(axis=None) / A.getnnz(axis=None)) ** 0.5
        lab_free = labels[~locked]
        for lbl in floating:                                   # pin one mesh of every floating component to its current estimate
            pos = int(np.nonzero(lab_free == lbl)[0][0])
            A[row, pos] = wt
            t = self.meshes[free_idx[pos]].estimate_translation(gear=(start_gear, target_gear))
            bx[row] = t[0] * wt; by[row] = t[1] * wt
            row += 1
        A = A.tocsr()
        Tx = lsqr(A, bx, atol=tol, btol=tol, iter_lim=maxiter)[0]
        Ty = lsqr(A, by, atol=tol, btol=tol, iter_lim=maxiter)[0]
        rx = A.dot(Tx) - bx; ry = A.dot(Ty) - by
        c0 = c1 = 0.0
        if np.linalg.norm(bx) <= np.linalg.norm(rx):
            Tx = np.zeros_like(Tx)
        else:
            c0 += np.linalg.norm(bx); c1 += np.linalg.norm(rx)
        if np.linalg.norm(by) <= np.linalg.norm(ry):
            Ty = np.zeros_like(Ty)
        else:
            c0 += np.linalg.norm(by); c1 += np.linalg.norm(ry)
        cost = None
        if np.any(Tx != 0) or np.any(Ty != 0):
            for k, tx, ty in zip(free_idx, Tx, Ty):
                self.meshes[k].set_translation((tx, ty), gear=(start_gear, target_gear))
            cost = (float(c0), float(c1))
        residue = None
        if return_residue and cost is not None:
            sel = rel > 0
            nrel = int(np.sum(sel))
            residue = np.zeros(nl, dtype=np.float32)
            residue[sel] = (rx[:nrel] ** 2 + ry[:nrel] ** 2) ** 0.5 / rel[sel]
        return cost, residue

    def optimize_translation_w_filtering(self, **kwargs):
        """optimizer.py:1081-1125: translation least squares, then links whose residue exceeds residue_threshold are
        disabled (the worst first, at most one per mesh and sweep) and the fit is repeated."""
        maxiter = kwargs.get('maxiter', None)
        tol = kwargs.get('tol', 1e-07)
        target_gear = kwargs.get('target_gear', const.MESH_GEAR_FIXED)
        start_gear = kwargs.get('start_gear', target_gear)
        thresh = kwargs.get('residue_threshold', None)
        cost0, residue = self.optimize_translation_lsqr(maxiter=maxiter, tol=tol, start_gear=start_gear, target_gear=target_gear)
        disabled = 0
        if thresh is not None and thresh > 0 and cost0 is not None:
            while True:
                bad = sorted(((residue[k], self.links[k].uids, k) for k in np.flatnonzero(residue > thresh)), reverse=True)
                if not bad:
                    break
                touched = set()
                for _, uids, k in bad:
                    if touched.isdisjoint(uids):
                        self.links[k].disable()
                        disabled += 1
                    touched.update(uids)
                cost1, residue = self.optimize_translation_lsqr(maxiter=maxiter, tol=tol, start_gear=start_gear, target_gear=target_gear)
                if cost1 is None or cost1[1] >= cost1[0]:
                    break
                cost0 = (cost0[0], min(cost1[1], cost0[1]))
        return disabled, cost0

    def optimize_affine_cascade(self, **kwargs):
        """optimizer.py:1128-1189: starting from the meshes linked to locked (or already placed) ones, fit one affine /
        rigid transform per free mesh onto its placed neighbours (spatial.fit_affine, host) and write it to target_gear."""
        target_gear = kwargs.get('target_gear', const.MESH_GEAR_MOVING)
        start_gear = kwargs.get('start_gear', target_gear)
        svd_clip = kwargs.get('svd_clip', (1, 1))
        Adj = self.linkage_adjacency()
        to_optimize = ~self.lock_flags
        uid2idx = {m.uid: k for k, m in enumerate(self.meshes)}
        pairs = np.array([[uid2idx.get(lk.uids[0], -1), uid2idx.get(lk.uids[1], -1)] for lk in self.links], dtype=np.int64).reshape(-1, 2)
        pairs[np.any(pairs < 0, axis=-1)] = -1
        modified = False
        while np.any(to_optimize):
            wsum = Adj.dot((~to_optimize).astype(np.float64)) * to_optimize
            if not np.any(wsum > 0):
                wsum = Adj.dot(np.ones(to_optimize.size)) * to_optimize
                if not np.any(wsum > 0):
                    break
            idx0 = int(np.argmax(wsum))
            placed = ~to_optimize[pairs]
            placed[pairs < 0] = False
            sel = np.nonzero(np.any(pairs == idx0, axis=-1) & np.any(placed, axis=-1))[0]
            if sel.size == 0:
                to_optimize[idx0] = False
                continue
            p_from, p_to, wts = [], [], []
            for li in sel:
                lk = self.links[li]
                if lk.uids[0] == self.meshes[idx0].uid:
                    p_from.append(lk.xy0(gear=start_gear, use_mask=True, combine=True))
                    p_to.append(lk.xy1(gear=target_gear, use_mask=True, combine=True))
                else:
                    p_from.append(lk.xy1(gear=start_gear, use_mask=True, combine=True))
                    p_to.append(lk.xy0(gear=target_gear, use_mask=True, combine=True))
                wts.append(lk.weight(use_mask=True))
            xy_from = np.concatenate(p_from, axis=0)
            if xy_from.size == 0:
                to_optimize[idx0] = False
                continue
            xy_to = np.concatenate(p_to, axis=0)
            _, A = common.fit_affine(xy_to, xy_from, return_rigid=True, weight=np.concatenate(wts, axis=None), svd_clip=svd_clip, avoid_flip=True)
            if (not modified) and np.any(xy_from != xy_to):
                modified = True
            self.meshes[idx0].set_affine(A, gear=(start_gear, target_gear))
            to_optimize[idx0] = False
        return modified

    @property
    def lock_flags(self):
        return np.array([m.locked for m in self.meshes], dtype=bool)

    @property
    def index_offsets(self):
        """optimizer.py:960-970: uid -> first DoF (or -1)."""
        out = {}
        cur = 0
        for m in self.meshes:
            if m.locked:
                out[m.uid] = -1
            else:
                out[m.uid] = cur
                cur += 2 * m.num_vertices
        return out

    @property
    def degree_of_freedom(self):
        return int(sum(2 * m.num_vertices for m in self.meshes if not m.locked))

    def _layout(self, groupings=None):
        """degree-of-freedom layout: {uid: first DoF or -1}, total DoF, and (with groupings, optimizer.py:1378-1415) the set
        of meshes that ADD into rows another member of their group owns, plus mean group size.  Members of a group share
        their DoFs; a group with a locked member is locked as a whole."""
        if groupings is None:
            return self.index_offsets, self.degree_of_freedom, set(), 1.0
        groupings = np.asarray(groupings)
        assert groupings.size == len(self.meshes)
        group_u, indx, group_nm, g_cnt = np.unique(groupings, return_index=True, return_inverse=True, return_counts=True)
        if group_u.size == groupings.size:
            return self.index_offsets, self.degree_of_freedom, set(), 1.0
        glock = np.zeros(group_u.size, dtype=bool)
        np.logical_or.at(glock, group_nm, self.lock_flags)
        vnum = np.array([self.meshes[k].num_vertices * 2 for k in indx]) * (~glock)
        acc = np.cumsum(vnum)
        goff = np.concatenate(([0], acc[:-1]))
        goff[glock] = -1
        offs, adders, seen = {}, set(), set()
        for k, m in enumerate(self.meshes):
            g = group_nm[k]
            if m.locked or goff[g] < 0:
                offs[m.uid] = -1
                continue
            assert m.num_vertices * 2 == vnum[g], 'meshes of a group must have the same number of vertices'
            offs[m.uid] = int(goff[g])
            if g in seen:
                adders.add(m.uid)
            seen.add(g)
        return offs, int(acc[-1]), adders, float(np.mean(g_cnt))

    # ------------------------------------------------------------------ GPU system
    def _active_links(self):
        return [lk for lk in self.links if lk.relevant and lk._tid0.size > 0]

    @staticmethod
    def _link_terms(lk, offs, nodes6, bary6=None, rxy=None, gears=None):
        """rows of one link for fb_sys_set_links / fb_sys_assemble_links (fb_link_terms: vertex ids of the two triangles of
        every match, [B0 | -B1], xy1 - xy0 at the given gears), written into slices of the caller's arrays"""
        m0, m1 = lk.meshes
        tid0 = np.ascontiguousarray(lk._tid0, dtype=np.int64); tid1 = np.ascontiguousarray(lk._tid1, dtype=np.int64)
        B0 = np.ascontiguousarray(lk._B0, dtype=np.float64); B1 = np.ascontiguousarray(lk._B1, dtype=np.float64)
        v0 = v1 = None
        ox = oy = 0.0
        if rxy is not None:
            v0 = np.ascontiguousarray(m0.vertices(gears[0]), dtype=np.float64); v1 = np.ascontiguousarray(m1.vertices(gears[1]), dtype=np.float64)
            d = np.asarray(m1.offset(gears[1]), dtype=np.float64).reshape(-1) - np.asarray(m0.offset(gears[0]), dtype=np.float64).reshape(-1)
            ox, oy = float(d[0]), float(d[1])
        _lib.check(_lib.load().fb_link_terms(_lib.ctx(), tid0.size, _lib.ptr(m0.triangles), m0.num_triangles, _lib.ptr(v0), _lib.ptr(tid0), _lib.ptr(B0),
                                             offs[m0.uid] // 2 if offs[m0.uid] >= 0 else -1, _lib.ptr(m1.triangles), m1.num_triangles, _lib.ptr(v1),
                                             _lib.ptr(tid1), _lib.ptr(B1), offs[m1.uid] // 2 if offs[m1.uid] >= 0 else -1, ox, oy,
                                             _lib.ptr(nodes6), _lib.ptr(bary6), _lib.ptr(rxy)))

    def _ensure_system(self, groupings=None, terms=None):
        """(Re)build the symbolic GPU system when the topology (free meshes, link
        connectivity, groupings) changed; numeric re-assembly reuses it.  terms = (start_gear, target_gear): the numeric
        rows of the links ([B0 | -B1], xy1 - xy0, weights) are produced in the same pass over the matches and kept in
        self._link_rows for _assemble."""
        lib = _lib.load()
        ctx = _lib.ctx()
        offs, dof, adders, gmean = self._layout(groupings)
        self._offs, self._adders, self._gmean = offs, adders, gmean
        links = [lk for lk in self._active_links() if offs[lk.meshes[0].uid] >= 0 or offs[lk.meshes[1].uid] >= 0]
        K = sum(lk._tid0.size for lk in links)
        nodes6 = np.empty((K, 6), dtype=np.int32)
        self._link_rows = None
        if terms is not None and K:
            bary = np.empty((K, 6)); res = np.empty((K, 2)); wts = np.empty(K, dtype=np.float32)
            self._link_rows = (bary, res, wts)
        at = 0
        for lk in links:
            n = lk._tid0.size
            if self._link_rows is not None:
                gears = [terms[1] if offs[m.uid] < 0 else terms[0] for m in lk.meshes]
                # [B0 | -B1] and Link.dxy (optimizer.py:248-255) of the link's matches straight into the rows of the system
                self._link_terms(lk, offs, nodes6[at:at + n], bary[at:at + n], res[at:at + n], gears)
                wts[at:at + n] = lk.weight(use_mask=False)
            else:
                self._link_terms(lk, offs, nodes6[at:at + n])
            at += n
        mesh_key = tuple((m.uid, offs[m.uid], m.num_vertices, m.triangles.ctypes.data) for m in self.meshes)
        if self._sys is not None and self._sys_key == mesh_key:
            # same free meshes: the links are swapped without a new symbolic phase when every coupling they make is in the
            # pattern already -- fb_sys_update_links checks exactly that (in threaded C++, no copy of the table is kept or
            # compared on this side).  Always true for matches with a locked side, whose couplings stay inside one triangle
            # of the free mesh (a section relaxed against locked neighbours, aligner.py:696-727 with one free section), and
            # for the same matches again (Newton-Raphson steps, residue re-weighting)
            rc = lib.fb_sys_update_links(ctx, self._sys, nodes6.shape[0], _lib.ptr(nodes6) if nodes6.shape[0] else None)
            if rc == 0:
                return links
        self._drop_system()
        nv = dof // 2
        sysh = C.c_void_p()
        _lib.check(lib.fb_sys_create(ctx, nv, C.byref(sysh)))
        self._sys = sysh
        self._mesh_ids = {}
        for m in self.meshes:
            if offs[m.uid] < 0:
                continue
            mid = C.c_int()
            _lib.check(lib.fb_sys_add_mesh(ctx, sysh, offs[m.uid] // 2, _lib.ptr(m.triangles), m.num_vertices,
                                           m.num_triangles, C.byref(mid)))
            self._mesh_ids[m.uid] = mid.value
        _lib.check(lib.fb_sys_set_links(ctx, sysh, nodes6.shape[0], _lib.ptr(nodes6)))
        nnzb = C.c_int64()
        _lib.check(lib.fb_sys_finalize(ctx, sysh, C.byref(nnzb)))
        self._nnzb = nnzb.value
        self._nv = nv
        self._sys_key = mesh_key
        return links

    def _assemble(self, shape_gear, start_gear, target_gear, groupings=None):
        """numeric assembly of K, stress, C, rhs on the GPU (optimizer.py:1307-1310)."""
        lib = _lib.load()
        ctx = _lib.ctx()
        links = self._ensure_system(groupings, terms=(start_gear, target_gear))
        offs = self._offs
        for m in self.meshes:
            if offs[m.uid] < 0 or m.uid in getattr(self, '_skip_stiffness', ()):
                continue
            v0 = np.ascontiguousarray(m.vertices(shape_gear), dtype=np.float64)
            v1 = m.vertices(start_gear)
            v1c = None if v1 is m.vertices(shape_gear) else np.ascontiguousarray(v1, dtype=np.float64)
            m.assemble_into(self._sys, self._mesh_ids[m.uid], v0, v1c, float(m.soft_factor), add=m.uid in self._adders)
        if links:
            bary, res, wts = self._link_rows
            self._link_rows = None
            _lib.check(lib.fb_sys_assemble_links(ctx, self._sys, _lib.ptr(bary), _lib.ptr(wts), _lib.ptr(res)))
        else:
            _lib.check(lib.fb_sys_assemble_links(ctx, self._sys, None, None, None))

    # --- exported equation terms (parity / inspection; optimizer.py:802-901)
    def stiffness_matrix(self, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING), **kwargs):
        self._assemble(gear[0], gear[-1], gear[-1])
        K = bsr_download(self._sys, 0, self._nv, self._nnzb)
        stress = np.empty(2 * self._nv, dtype=np.float32)
        _lib.check(_lib.load().fb_sys_get(_lib.ctx(), self._sys, 3, _lib.ptr(stress)))
        return K, stress

    def crosslink_terms(self, start_gear=const.MESH_GEAR_MOVING, target_gear=const.MESH_GEAR_MOVING, **kwargs):
        self._assemble(const.MESH_GEAR_FIXED, start_gear, target_gear)
        Cm = bsr_download(self._sys, 1, self._nv, self._nnzb)
        rhs = np.empty(2 * self._nv, dtype=np.float64)
        _lib.check(_lib.load().fb_sys_get(_lib.ctx(), self._sys, 2, _lib.ptr(rhs)))
        return Cm, rhs

    def relative_lambda_trace(self, stiffness_lambda, crosslink_lambda):
        sl, cl = C.c_double(), C.c_double()
        _lib.check(_lib.load().fb_sys_lambda(_lib.ctx(), self._sys, float(stiffness_lambda), float(crosslink_lambda),
                                             C.byref(sl), C.byref(cl)))
        return sl.value, cl.value

    # ------------------------------------------------------------------ optimize
    def optimize_linear(self, **kwargs):
        """feabas/optimizer.py:1257-1437 including `groupings` (members of a group share their degrees of freedom),
        `remove_extra_dof` and `remove_material_dof`.  Returns (||b||, ||A d - b||) and writes the field into the meshes."""
        maxiter = kwargs.get('maxiter', None)
        tol = kwargs.get('tol', 1e-7)
        atol = kwargs.get('atol', 0.0)
        shape_gear = kwargs.get('shape_gear', const.MESH_GEAR_FIXED)
        target_gear = kwargs.get('target_gear', const.MESH_GEAR_MOVING)
        start_gear = kwargs.get('start_gear', target_gear)
        stiffness_lambda = kwargs.get('stiffness_lambda', self._stiffness_lambda)
        crosslink_lambda = kwargs.get('crosslink_lambda', self._crosslink_lambda)
        precondition = kwargs.get('precondition', os.environ.get('FEABAS_HIP_PRECONDITION', 'jacobi'))
        groupings = kwargs.get('groupings', None)
        multigrid = isinstance(precondition, str) and precondition.lower().startswith(('smooth', 'sa', 'amg', 'multigrid', 'mg'))
        # matcher.py:561 asks for pyamg's smoothed aggregation (optimizer.py:1969-1971); here: the device's aggregation
        # multigrid (csrc/fb_mg.inc: rigid-body modes per aggregate, V(1,1) cycles) as the preconditioner of the same PCG --
        # the same fixed point, an order of magnitude fewer iterations on weakly pinned meshes
        if kwargs.get('distributed', None) is not None:
            return self._optimize_linear_distributed(kwargs['distributed'], kwargs.get('owned', None), tol, atol, maxiter, shape_gear,
                                                     start_gear, target_gear, stiffness_lambda, crosslink_lambda, groupings)
        if np.all(self.lock_flags):
            return 0, 0
        if kwargs.get('remove_material_dof', None) is not None:
            held = self._material_dof_mask(kwargs['remove_material_dof'], groupings)
        else:
            held = self._extra_dof_mask(groupings) if kwargs.get('remove_extra_dof', False) else None
        lib = _lib.load()
        ctx = _lib.ctx()
        if groupings is not None and all(o < 0 for o in self._layout(groupings)[0].values()):
            return 0, 0                                          # optimizer.py:1383-1384: every group holds a locked mesh
        grouped = groupings is not None and np.unique(np.asarray(groupings)).size < len(self.meshes)
        if grouped:
            # relative_lambda_trace reads the cached MESH-level K and C (optimizer.py:1573-1590), the grouped matrices only
            # enter A and b: the lambdas come from the ungrouped system, assembled first
            self._assemble(shape_gear, start_gear, target_gear, None)
            sl, cl = self.relative_lambda_trace(stiffness_lambda, crosslink_lambda)
            self._assemble(shape_gear, start_gear, target_gear, groupings)
        else:
            self._assemble(shape_gear, start_gear, target_gear, None)
            sl, cl = self.relative_lambda_trace(stiffness_lambda, crosslink_lambda)
        _lib.check(lib.fb_sys_form(ctx, self._sys, sl, cl))
        b = np.empty(2 * self._nv, dtype=np.float64)
        _lib.check(lib.fb_sys_get(ctx, self._sys, 5, _lib.ptr(b)))
        dd = np.zeros(2 * self._nv, dtype=np.float64)
        iters, relres = C.c_int(), C.c_double()
        mi = -1 if maxiter is None else int(maxiter)
        # (round 3 kept the hierarchy away from windows without a locked mesh: with its fixed smoother damping the cycle was
        # indefinite there; the damping now follows lambda_max(Dinv A) of every level, csrc/fb_mg.inc::mg_damping)
        auto = isinstance(precondition, str) and precondition.lower() == 'auto'
        # 'auto' (not a reference value): Jacobi-PCG for the iterations a multigrid solve would cost, then the multigrid-PCG from
        # the iterate reached (fb_sys_solve precond 3) -- within about twice the better of the two on any system
        pre = 0 if precondition is None else (2 if multigrid and groupings is None else (3 if auto and groupings is None else 1))
        bn = float(np.linalg.norm(b)) / self._gmean           # grouped terms are divided by mean(count) (optimizer.py:1408-1411)
        if held is not None and not held.all():
            # remove_extra_dof (optimizer.py:1360-1377, 1976-1991): three degrees of freedom of the first mesh of every
            # connected subsystem without a locked member are taken out of the solve -- the assembled A goes through the
            # masked device PCG of `solve` (the rarely taken option does not warrant a masked variant of fb_sys_solve)
            A = bsr_download(self._sys, 4, self._nv, self._nnzb)
            dd = solve(A, b, tol=tol, atol=atol, maxiter=maxiter, M='jacobi', extra_dof_constraint=held)
            res = float(np.linalg.norm(A @ dd - b)) / self._gmean
            cost = (bn, res)
            self.last_solve = dict(iters=None, relres=res / bn if bn else 0.0, stiffness_lambda=sl, crosslink_lambda=cl, held_dofs=int((~held).sum()))
        else:
            rc = lib.fb_sys_solve(ctx, self._sys, _lib.ptr(dd), 0, float(tol), float(atol or 0.0), mi, pre, C.byref(iters), C.byref(relres))
            if pre != 2:
                _lib.check(rc, allow=(_lib.FB_ERR_NOCONV,))
            fell_back = False
            if pre == 2 and (rc != 0 or not relres.value <= max(float(tol), float(atol or 0.0) / bn if bn else 0.0) * 1.0001):
                # the hierarchy is a heuristic: a set-up that cannot be made (a coarsest level that stays too large -- aggregates never
                # join two meshes, so a window of very many free meshes ends there --, a level that stops coarsening), a breakdown or
                # a cycle that stalls all end here, and the plain Jacobi-PCG takes over (any error of ITS run is raised)
                dd[:] = 0.0
                _lib.check(lib.fb_sys_solve(ctx, self._sys, _lib.ptr(dd), 0, float(tol), float(atol or 0.0), mi, 1, C.byref(iters), C.byref(relres)),
                           allow=(_lib.FB_ERR_NOCONV,))
                fell_back = True
            cost = (bn, float(relres.value * bn))
            self.last_solve = dict(iters=iters.value, relres=relres.value, stiffness_lambda=sl, crosslink_lambda=cl)
            if fell_back:
                self.last_solve['multigrid_fell_back'] = True
        if cost[1] < cost[0] and not self._solution_is_sane(dd):
            # not applied: the cost must read as 'no step taken' to the callers that look at it (the Newton driver, the matcher's
            # relaxations; optimizer.py:1421 applies a field exactly when cost[1] < cost[0])
            cost = (cost[0], cost[0])
        if cost[1] < cost[0]:                                   # optimizer.py:1421
            offs = self._offs
            for m in self.meshes:
                o = offs[m.uid]
                if m.locked or o < 0:
                    continue
                m.set_field(dd[o:o + 2 * m.num_vertices].reshape(-1, 2), gear=(start_gear, target_gear))
        return cost

    def _solution_is_sane(self, dd):
        """a displacement field that is not finite, or that moves a node of a FLOATING sub-system -- link-connected meshes none of
        which is locked: the ones with a null space to drift along -- by more than a thousand times the extent of all the meshes
        together, is the drift of a solve that went wrong (a floating system pushed past what doubles can give), never an
        alignment: it is not applied (the meshes keep their state, `optimize_linear` returns a cost that reads 'not improved'), and
        `last_solve['rejected']` says why.  Downstream steps size host and device buffers by where the meshes are."""
        if not np.all(np.isfinite(dd)):
            self.last_solve['rejected'] = 'not finite'
            return False
        if float(np.max(np.abs(dd), initial=0.0)) <= 1e3:
            return True                                       # (extent >= 1: nothing this small is screened -- the common case, no further pass)
        locks = np.asarray(self.lock_flags, dtype=bool)
        labels, _ = self.connected_subsystems
        floating = ~np.isin(labels, labels[locks]) & ~locks
        if not floating.any():
            return True
        offs = getattr(self, '_offs', None) or self.index_offsets
        worst = 0.0
        for m, fl in zip(self.meshes, floating):
            o = offs.get(m.uid, -1)
            if fl and o >= 0:
                worst = max(worst, float(np.max(np.abs(dd[o:o + 2 * m.num_vertices]), initial=0.0)))
        lo = np.min([m.bbox(gear=const.MESH_GEAR_MOVING)[:2] for m in self.meshes], axis=0)
        hi = np.max([m.bbox(gear=const.MESH_GEAR_MOVING)[2:] for m in self.meshes], axis=0)
        extent = float(max(np.max(hi - lo), 1.0))
        if worst > 1e3 * extent:
            self.last_solve['rejected'] = f'displacements up to {worst:.3g} on floating meshes of extent {extent:.3g}'
            return False
        return True

    def _material_dof_mask(self, names, groupings):
        """optimizer.py:1320-1359: the regions of the named materials do not take part in the solve -- every vertex of their
        triangles is held; a name with the suffix '_freeborder' keeps the vertices such a region shares with other materials free
        (and is applied first).  The materials are those a Mesh was given as ``material_ids`` + ``material_names`` (the mesh file
        carries both, h5wire.load_mesh_h5); a mesh without them has no named region.  Returns the selector over the degrees of
        freedom of the free meshes (True = solved)."""
        if isinstance(names, str):
            names = [names]
        elif not isinstance(names, (tuple, list)):
            raise TypeError('remove_material_dof: a material name or a list of names')
        marker = '_freeborder'
        passes = [(s_.replace(marker, ''), True) for s_ in names if marker in s_] + [(s_, False) for s_ in names if marker not in s_]
        parts = []
        for m in self.meshes:
            if m.locked:
                continue
            solved = np.ones(m.num_vertices, dtype=bool)
            for name, free_border in passes:
                region = m.triangles_of_material(name)
                solved[m.triangles[region].ravel()] = False
                if free_border:
                    solved[m.triangles[~region].ravel()] = True
            parts.append(np.repeat(solved, 2))
        return self._fold_dof_selector(parts, groupings)

    def _fold_dof_selector(self, parts, groupings):
        """the selector over the degrees of freedom of the free meshes (one boolean array per free mesh, in mesh order; True =
        solved) in the layout of the system that is solved: concatenated as it is without groupings; with them
        ``edc = (T_m @ edc) > 0`` (optimizer.py:1412-1413) -- the members of a group share their degrees of freedom, and one
        is solved when ANY member has it solved (a hold on one member is undone by a member that does not hold it: the
        reference's rule, golden G21)."""
        if groupings is None or np.unique(np.asarray(groupings)).size == len(self.meshes):
            return np.concatenate(parts)
        offs, ndof, _, _ = self._layout(groupings)
        out = np.zeros(ndof, dtype=bool)
        free = [m for m in self.meshes if not m.locked]
        assert len(free) == len(parts)
        for m, sel in zip(free, parts):
            o = offs[m.uid]
            if o >= 0:                                         # (a free mesh in a group with a locked member is locked with it)
                out[o:o + sel.size] |= sel
        return out

    def _extra_dof_mask(self, groupings):
        """optimizer.py:1360-1377: a connected subsystem without a locked mesh floats (rigid motion costs nothing); the first
        three degrees of freedom -- vertex 0 and the x of vertex 1 -- of its first mesh are held.  Returns the boolean
        selector over the degrees of freedom of the free meshes (True = solved), or None when nothing floats."""
        labels, _ = self.connected_subsystems
        locks = np.asarray(self.lock_flags, dtype=bool)
        first = np.zeros(len(self.meshes), dtype=bool)
        for lbl in np.unique(labels):
            idx = np.flatnonzero(labels == lbl)
            if not locks[idx].any():
                first[idx[0]] = True
        if not first.any():
            return None
        parts = []
        for flg, m in zip(first, self.meshes):
            if m.locked:
                continue
            sel = np.ones(2 * m.num_vertices, dtype=bool)
            if flg:
                sel[:3] = False
            parts.append(sel)
        return self._fold_dof_selector(parts, groupings)

    def _optimize_linear_distributed(self, group, owned, tol, atol, maxiter, shape_gear, start_gear, target_gear, stiffness_lambda, crosslink_lambda,
                                     groupings=None):
        """optimize_linear of a coupled window (aligner.py:510-535, 696-727: all free sections of a window are ONE system)
        with the rows partitioned by section over the ranks of `group` (torch.distributed; True = the default group).
        Every rank holds the sections it OWNS (default: a contiguous shard of the free meshes in list order, every rank
        listing the same meshes; or the uids in `owned`, the rank then only needs its own sections and the ones they are
        linked to) plus the links that touch them.  Its rows -- stiffness of the own sections, link terms of every match that
        touches them, including the blocks that couple to a neighbour's section -- are assembled on the device (fb_sys_*),
        and solved with the row-partitioned Jacobi-PCG of feabas_amd/dist.py (halo exchange with the neighbouring ranks +
        one fused all-reduce per iteration; fused vector kernels on the GPU).  The trace-relative lambdas come from
        all-reduced sums.  Returns (||b||, ||A d - b||) of the whole window; every rank moves the sections it owns.

        `groupings` (optimizer.py:1378-1415; one label per mesh of self.meshes, the same labels on every rank): the members of
        a group share their degrees of freedom, so the unit of ownership is the GROUP -- the shard is one of free groups in order of
        first appearance, `owned` must name every free member of a group it names, and a rank lists every member of the
        groups it touches (a group with a locked member is locked as a whole, as in the reference).  The lambdas come from the
        mesh-level (ungrouped) terms like the reference's (optimizer.py:1573-1590), the costs are divided by the mean group size."""
        import torch.distributed as dist                        # the rendezvous (who owns what); no tensor of torch's is involved
        from . import dist as fdist
        from .mesh import bsr_download
        grp = fdist.host_group(None if group is True else group)
        rank, world = dist.get_rank(grp), dist.get_world_size(grp)
        if groupings is not None:
            labels = np.asarray(groupings).ravel().tolist()
            if len(labels) != len(self.meshes):
                raise ValueError('optimize_linear(distributed, groupings): one label per mesh')
            if len(set(labels)) == len(labels):
                groupings = None                                # every mesh alone in its group
        if groupings is None:
            unit_of = {id(m): m.uid for m in self.meshes}      # the unit of ownership: the mesh
        else:
            unit_of = {id(m): ('g', lb) for m, lb in zip(self.meshes, labels)}
        held_units = {unit_of[id(m)] for m in self.meshes if m.locked}
        units = {}                                              # free units in order of first appearance -> their meshes
        for m in self.meshes:
            if unit_of[id(m)] not in held_units:
                units.setdefault(unit_of[id(m)], []).append(m)
        wards = [m for m in self.meshes if not m.locked and unit_of[id(m)] in held_units]   # free meshes of a locked group: held, but counted by the lambdas
        order = list(units)
        if owned is None:
            a, b = fdist.shard_range(len(order), rank, world)
            own_units = order[a:b]
            my_wards = wards[rank::world]
        else:
            owned = {float(u) for u in owned}
            own_units = [u for u in order if any(m.uid in owned for m in units[u])]
            for u in own_units:
                if not all(m.uid in owned for m in units[u]):
                    raise ValueError(f'optimize_linear(distributed, groupings): group {u[1]} is only partly in `owned`')
            my_wards = [m for m in wards if m.uid in owned]
        own = [m for u in own_units for m in units[u]]
        own_ids = {id(m) for m in own}
        links = [lk for lk in self._active_links() if id(lk.meshes[0]) in own_ids or id(lk.meshes[1]) in own_ids]
        halo, halo_units, seen = [], [], set(own_ids)
        for lk in links:
            for m in lk.meshes:
                u = unit_of[id(m)]
                if u not in held_units and id(m) not in seen:
                    seen.add(id(m)); halo.append(m)
                    if u not in halo_units:
                        halo_units.append(u)
        locked = [m for m in self.meshes if unit_of[id(m)] in held_units]
        unit_dof = {u: 2 * units[u][0].num_vertices for u in order}
        n_own = int(sum(unit_dof[u] for u in own_units))
        sizes = [None] * world
        dist.all_gather_object(sizes, [(u, unit_dof[u]) for u in own_units], group=grp)
        gstart, cur, row_start = {}, 0, 0
        for r_, lst in enumerate(sizes):
            if r_ == rank:
                row_start = cur
            for u, nd in lst:
                gstart[u] = cur
                cur += nd
        missing = [m.uid for m in halo if unit_of[id(m)] not in gstart]
        if missing:
            raise ValueError(f'optimize_linear(distributed): sections {missing} are linked to this rank but owned by no rank')
        gmean = 1.0
        if groupings is not None:
            # mean group size over the whole window (optimizer.py:1408-1411 divides the grouped terms by it): free groups from their
            # owners, held groups from whoever lists them
            seen_held = [None] * world
            dist.all_gather_object(seen_held, {u: sum(1 for m in self.meshes if unit_of[id(m)] == u) for u in held_units}, group=grp)
            held_cnt = {}
            for d_ in seen_held:
                for u, c in d_.items():
                    held_cnt[u] = max(held_cnt.get(u, 0), c)
            nm, ng = fdist.host_sum([len(own), len(own_units)], group=grp)
            gmean = float((nm + sum(held_cnt.values())) / max(1.0, ng + len(held_cnt)))

        def mesh_level_sums(first, others, lks):
            """(sum diag C, sum of diag K where diag C != 0) over the rows of the meshes `first`, from an ungrouped assembly"""
            if not first:
                return np.zeros(2)
            mine = {id(m) for m in first}
            near, got = [], set(mine)
            for lk in lks:
                for m in lk.meshes:
                    if id(m) not in got and not m.locked:
                        got.add(id(m)); near.append(m)
            tmp = SLM(first + near + [m for m in others if m.locked], lks, stiffness_lambda=stiffness_lambda, crosslink_lambda=crosslink_lambda)
            tmp._skip_stiffness = {m.uid for m in near}
            try:
                tmp._assemble(shape_gear, start_gear, target_gear)
                n1 = int(sum(2 * m.num_vertices for m in first))
                dk = bsr_download(tmp._sys, 0, tmp._nv, tmp._nnzb)[:n1, :n1].diagonal()
                dc = bsr_download(tmp._sys, 1, tmp._nv, tmp._nnzb)[:n1, :n1].diagonal()
            finally:
                tmp._drop_system()
            return np.array([dc.sum(), dk[dc != 0].sum()])
        if n_own > 0:
            if groupings is None:
                local = SLM(own + halo + locked, links, stiffness_lambda=stiffness_lambda, crosslink_lambda=crosslink_lambda)
                local_groups = None
            else:
                # local labels in row order: own groups, then the groups of the halo, then the held ones (_layout numbers the
                # groups in sorted label order)
                members = own + halo + locked
                code = {u: k for k, u in enumerate(own_units + halo_units + sorted(held_units, key=repr))}
                local = SLM(members, links, stiffness_lambda=stiffness_lambda, crosslink_lambda=crosslink_lambda)
                local_groups = np.array([code[unit_of[id(m)]] for m in members], dtype=np.int64)
            local._skip_stiffness = {m.uid for m in halo}
            local._assemble(shape_gear, start_gear, target_gear, local_groups)
            nv, nnzb = local._nv, local._nnzb
            K = bsr_download(local._sys, 0, nv, nnzb)[:n_own]
            Cm = bsr_download(local._sys, 1, nv, nnzb)[:n_own]
            rhs = np.empty(2 * nv); stress = np.empty(2 * nv, dtype=np.float32)
            _lib.check(_lib.load().fb_sys_get(_lib.ctx(), local._sys, 2, _lib.ptr(rhs)))
            _lib.check(_lib.load().fb_sys_get(_lib.ctx(), local._sys, 3, _lib.ptr(stress)))
            rhs, stress = rhs[:n_own], stress[:n_own].astype(np.float64)
            if groupings is None:
                dk = K[:, :n_own].diagonal(); dc = Cm[:, :n_own].diagonal()
                sums = np.array([dc.sum(), dk[dc != 0].sum()])
        else:
            sums = np.zeros(2)
        if groupings is not None:
            first = own + my_wards
            ids1 = {id(m) for m in first}
            sums = mesh_level_sums(first, self.meshes, [lk for lk in self._active_links() if id(lk.meshes[0]) in ids1 or id(lk.meshes[1]) in ids1])
        tr_c, sum_k = (float(v) for v in fdist.host_sum(sums, group=grp))
        sl, cl = float(stiffness_lambda), float(crosslink_lambda)
        if sl < 0 or cl < 0:                                    # optimizer.py:1573-1590 on the all-reduced sums
            sl = 0.0 if tr_c == 0 else abs(abs(sl / cl) * tr_c / sum_k)
            cl = 1.0
        if n_own > 0:
            A = (sl * K + cl * Cm).tocsr()
            A.sort_indices()
            bvec = cl * rhs - sl * stress
            # local column -> global DoF: own columns first, then the halo sections (groups) at their owner's offsets
            col_map = np.empty(2 * nv, dtype=np.int64)
            col_map[:n_own] = row_start + np.arange(n_own)
            o = n_own
            for u in ([unit_of[id(m)] for m in halo] if groupings is None else halo_units):
                nd = unit_dof[u]
                col_map[o:o + nd] = gstart[u] + np.arange(nd)
                o += nd
            indptr, gcols, data = A.indptr.astype(np.int64), col_map[A.indices], A.data
            diag = A[:, :n_own].diagonal()
        else:
            indptr, gcols, data, bvec, diag = np.zeros(1, np.int64), np.zeros(0, np.int64), np.zeros(0), np.zeros(0), np.zeros(0)
        part = fdist.RowPartition(indptr, gcols, row_start, group=grp)
        bnorm = float(fdist.host_sum([float(bvec @ bvec)], group=grp)[0]) ** 0.5
        if bnorm == 0 or maxiter == 0:
            return 0.0, 0.0
        rtol = max(float(tol), float(atol or 0.0) / bnorm)
        minv = 1.0 / np.clip(diag, min(1.0, diag.max(initial=0.0) / 1000.0) if diag.size else 1.0, None)       # optimizer.py:1962-1966 on the local rows
        # the loop itself: fb_cgcg_solve_dev on the library's RCCL communicator (halo: fb_sendrecv_dev, scalars:
        # fb_allreduce_f64_dev); ranks that share one device (tests) carry halo and scalars over the host group instead
        comm = fdist.solver_comm(grp)
        rows = fdist.DeviceRows(part, indptr, data)
        try:
            x, it, rel = fdist.pcg_row_partitioned_dev(part, rows, bvec, minv, rtol=rtol, maxiter=10000 if maxiter is None else int(maxiter), comm=comm)
        finally:
            rows.free()
        self.last_solve = dict(iters=int(it), relres=float(rel), stiffness_lambda=sl, crosslink_lambda=cl, rows=n_own, halo=int(part.n_halo),
                               exchange='rccl' if comm is not None else ('none' if world == 1 else 'host group'))
        cost = (bnorm / gmean, float(rel) * bnorm / gmean)
        if cost[1] < cost[0]:
            o = 0
            for u in own_units:
                nd = unit_dof[u]
                for m in units[u]:                              # the members of a group move together (optimizer.py:1425-1432)
                    m.set_field(x[o:o + nd].reshape(-1, 2), gear=(start_gear, target_gear))
                o += nd
        return cost

    @property
    def is_linear(self):
        return all(m.is_linear for m in self.meshes)

    def cost(self, stiffness_lambda, crosslink_lambda):
        """optimizer.py:1593-1601: ||lc rhs - ls stress|| of the assembled terms"""
        lib, ctx = _lib.load(), _lib.ctx()
        sl, cl = self.relative_lambda_trace(stiffness_lambda, crosslink_lambda)
        rhs = np.empty(2 * self._nv); stress = np.empty(2 * self._nv, dtype=np.float32)
        _lib.check(lib.fb_sys_get(ctx, self._sys, 2, _lib.ptr(rhs)))
        _lib.check(lib.fb_sys_get(ctx, self._sys, 3, _lib.ptr(stress)))
        return float(np.linalg.norm(cl * rhs - sl * stress))

    @staticmethod
    def _ladder(value, steps, toward_first=None):
        """Per-step schedule of a Newton-Raphson keyword (SLM.expand_to_list, optimizer.py:1862-1873): a scalar is the value
        of the LAST step, a sequence is right-aligned; the steps before are derived one by one from the step that follows
        them (`toward_first`), or repeat it.  A sequence LONGER than the number of steps is kept whole, as the reference keeps
        it: step k then takes its k-th element from the front and the cost is evaluated with its last one."""
        given = [value] if (not hasattr(value, '__len__') or isinstance(value, str) or len(value) == 0) else list(value)
        sched = list(given)
        while len(sched) < steps:
            sched.insert(0, toward_first(sched[0]) if toward_first is not None else sched[0])
        return sched

    def optimize_Newton_Raphson(self, **kwargs):
        """feabas/optimizer.py:1440-1544.  Every step re-assembles the tangent stiffness and the internal force at the
        current MOVING gear on the device and solves the tangent problem with optimize_linear; the step tolerances tighten
        geometrically from max(tol, 1e-5) to tol, every inner solve also stops at the absolute floor tol * cost0, and a step
        whose out-of-balance force is below that floor is followed by one last step.  Between steps, as asked per step:
        annealing of the resting shape (anneal_mode, after relaxing the most deformed regions when deform_outlier_constant
        > 0 -- the `online_anneal` of optimize_elastic) and residue re-weighting of the links (residue_mode 'huber' /
        'hard' with residue_len, which doubles towards the earlier steps; negative = units of the section thickness over
        the working resolution, keywords section_thickness / working_resolution).  Returns (cost0, cost): the
        out-of-balance force ||lc rhs - ls stress|| before the first step and the smallest one met after a step."""
        n = int(kwargs.pop('max_newtonstep', 5))
        if n < 1:
            raise ValueError('max_newtonstep < 1')
        tol = kwargs.pop('tol', 1e-7)
        atol = kwargs.pop('atol', 0)
        growth = (max(tol, 1e-5) / tol) ** (1.0 / (n - 1)) if n > 1 else 1.0
        maxiter = self._ladder(kwargs.pop('maxiter', None), n)
        step_tol = self._ladder(kwargs.pop('step_tol', tol), n, lambda t: t * growth)
        step_atol = self._ladder(kwargs.pop('step_atol', atol), n)
        ls = self._ladder(kwargs.pop('stiffness_lambda', self._stiffness_lambda), n)
        lc = self._ladder(kwargs.pop('crosslink_lambda', self._crosslink_lambda), n)
        residue_mode = self._ladder(kwargs.pop('residue_mode', None), n, lambda _: None)
        residue_len = self._ladder(kwargs.pop('residue_len', 0), n, lambda r: 2 * r)
        anneal_mode = self._ladder(kwargs.pop('anneal_mode', None), n)
        outlier = self._ladder(kwargs.pop('deform_outlier_constant', 0), n)
        kwargs.pop('check_converge', None); kwargs.pop('inner_cache', None)       # the device PCG always runs to its tolerance
        aspect = kwargs.pop('section_thickness', const.DEFAULT_THICKNESS) / kwargs.pop('working_resolution', self.meshes[0].resolution)
        residue_len = [abs(r) * aspect if r < 0 else r for r in residue_len]
        target_gear = kwargs.pop('target_gear', const.MESH_GEAR_MOVING)
        shape_gear, start_gear = const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING
        if np.all(self.lock_flags):
            return None, None
        self._assemble(shape_gear, start_gear, target_gear)
        cost0 = self.cost(ls[-1], lc[-1])
        floor = atol if tol is None else (cost0 * tol if atol is None else max(cost0 * tol, atol))
        best = np.inf
        ke = 0
        while ke < n:
            bnorm, rnorm = self.optimize_linear(maxiter=maxiter[ke], tol=step_tol[ke], atol=max(step_atol[ke], floor), shape_gear=shape_gear,
                                                start_gear=start_gear, target_gear=target_gear, stiffness_lambda=ls[ke], crosslink_lambda=lc[ke],
                                                **kwargs)
            if bnorm < rnorm:                                # the step made things worse: nothing was applied
                break
            annealed = anneal_mode[ke] is not None
            if annealed:
                if outlier[ke] > 0:
                    self.relax_higly_deformed(gear=(shape_gear, const.MESH_GEAR_STAGING), iqr=outlier[ke])
                self.anneal(gear=(const.MESH_GEAR_STAGING, shape_gear), mode=anneal_mode[ke])
                for m in self.meshes:
                    if not m.locked:
                        m.clear_staging()
            if residue_mode[ke] is not None:
                if residue_len[ke] > 0:
                    (self.set_link_residue_huber if residue_mode[ke] == 'huber' else self.set_link_residue_threshold)(residue_len[ke])
                self.adjust_link_weight_by_residue(gear=(target_gear, target_gear), relax_first=not annealed)
            if start_gear != target_gear:
                self.anneal(gear=(target_gear, start_gear), mode=const.ANNEAL_COPY_EXACT)
            self._assemble(shape_gear, target_gear, target_gear)
            best = min(best, self.cost(ls[-1], lc[-1]))
            ke += 1
            if floor is not None and best < floor:
                ke = max(ke, n - 1)                          # converged: one last step at the final tolerance
            if ke >= len(ls):                                # optimizer.py:1541-1542
                break
        return cost0, best

    def optimize_elastic(self, **kwargs):
        """optimizer.py:1547-1555."""
        online = kwargs.pop('online_anneal', False)
        if online:
            kwargs.setdefault('anneal_mode', const.ANNEAL_COPY_EXACT)
            kwargs.setdefault('deform_outlier_constant', 1.5)
        if self.is_linear and not online:
            return self.optimize_linear(**kwargs)
        return self.optimize_Newton_Raphson(**kwargs)


def solve(A, b, solver='minres', x0=None, tol=1e-7, atol=None, maxiter=None, M=None, **kwargs):
    """feabas/optimizer.py:1945-2080: solve A x = b to ||Ax-b|| <= max(tol, atol/||b||) ||b||
    on the GPU (Jacobi-PCG on 0.5 (A + A^T)).  ``solver`` and ``M`` are accepted for signature
    compatibility: both only choose HOW the reference iterates, the solution to the tolerance is the same.  A bare matrix
    carries no mesh geometry, which the device's aggregation multigrid is built from (csrc/fb_mg.inc), so
    M='smoothed_aggregation' (optimizer.py:1969-1971) is served by the Jacobi-PCG here; the multigrid is reached through
    ``SLM.optimize_linear(precondition='smoothed_aggregation')``, which knows the meshes."""
    edc = kwargs.get('extra_dof_constraint', None)
    A = sparse.csr_matrix(A)
    b = np.ascontiguousarray(b, dtype=np.float64)
    if (maxiter == 0) or (np.linalg.norm(b) == 0):
        return np.zeros_like(b)
    if not (M is None or isinstance(M, str)):
        raise NotImplementedError('solve(M=<operator>): a host preconditioner object cannot run inside the device PCG')
    full_n = b.size
    if edc is not None:                                     # optimizer.py:1976-1991
        if (not isinstance(edc, np.ndarray)) or (edc.dtype != bool):
            sel = edc
            edc = np.zeros(full_n, dtype=bool)
            edc[sel] = True
        if np.all(edc):
            edc = None
        elif not np.any(edc):
            return np.zeros_like(b)
        else:
            A = A[edc][:, edc]
            b = np.ascontiguousarray(b[edc])
            if x0 is not None:
                x0 = x0[edc]
    A.sort_indices()
    n = b.size
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    idx = np.ascontiguousarray(A.indices, dtype=np.int32)
    val = np.ascontiguousarray(A.data, dtype=np.float64)
    x = np.zeros(n, dtype=np.float64) if x0 is None else np.ascontiguousarray(x0, dtype=np.float64).copy()
    iters, relres = C.c_int(), C.c_double()
    _lib.check(_lib.load().fb_pcg(_lib.ctx(), n, _lib.ptr(indptr), _lib.ptr(idx), _lib.ptr(val), _lib.ptr(b), _lib.ptr(x),
                                  0 if x0 is None else 1, float(tol), float(atol or 0.0), -1 if maxiter is None else int(maxiter),
                                  1, 1, C.byref(iters), C.byref(relres)), allow=(_lib.FB_ERR_NOCONV,))
    if edc is not None:
        full = np.zeros(full_n, dtype=np.float64)
        full[edc] = x
        x = full
    return x


def _free_vertex_mask(M, free_vertices, free_triangles):
    """vertices a relaxation may move: the listed ones, or every vertex that no HELD triangle uses"""
    mask = np.zeros(M.num_vertices, dtype=bool)
    if free_vertices is not None:
        mask[free_vertices] = True
    elif free_triangles is not None:
        held = np.zeros(M.num_vertices, dtype=bool)
        held[M.triangles[~np.asarray(free_triangles, dtype=bool)].ravel()] = True
        mask = ~held
    return mask


class _RestingShapeAligned:
    """For the duration of a local relaxation the resting shape (gear[0]) of the mesh is its INITIAL shape laid rigidly,
    connected part by connected part, over the current state (gear[1]); afterwards gear[0] is what it was, and so is the
    lock (optimizer.py:2131-2135, 2149-2153)."""

    def __init__(self, M, gear):
        self.M, self.gear = M, gear

    def __enter__(self):
        M, g = self.M, self.gear
        self.lock, M.locked = M.locked, False
        self.saved = (M.vertices(gear=g[0]), M.offset(gear=g[0]))
        M.anneal(gear=(const.MESH_GEAR_INITIAL, g[0]), mode=const.ANNEAL_COPY_EXACT)
        M.anneal(gear=g[::-1], mode=const.ANNEAL_CONNECTED_RIGID)
        return self

    def __exit__(self, *exc):
        M, g = self.M, self.gear
        if g[0] != g[1]:
            M.set_vertices(self.saved[0], gear=g[0])
            M.set_offset(self.saved[1], gear=g[0])
        M.locked = self.lock
        return False


def relax_mesh(M, free_vertices=None, free_triangles=None, **kwargs):
    """feabas/optimizer.py:2110-2154: relax a region of one mesh with the rest of it held.  The free vertices are the
    given ones, or those no held triangle uses.  The triangles that touch a free vertex are assembled on the device with
    clipped multipliers against the rigidly aligned INITIAL shape (``Mesh.stiffness_matrix_local_normalized``) and the
    held degrees of freedom are taken out by the solver itself (``solve(extra_dof_constraint=)``: the device PCG works on
    the free block); the field is applied when it lowers the residual.  Returns whether the mesh was modified.
    ``tolerated_perturbation`` / ``callback_settings`` are accepted and unused: the PCG runs to ``tol``."""
    gear = kwargs.get('gear', (const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING))
    if free_vertices is None and free_triangles is None:
        return False
    free = _free_vertex_mask(M, free_vertices, free_triangles)
    if not free.any():
        return False
    dof = np.repeat(free, 2)
    with _RestingShapeAligned(M, gear):
        K, stress = M.stiffness_matrix_local_normalized(gear=gear, tri_mask=free[M.triangles].any(axis=1))
        if K is None:
            return False
        force = -stress.astype(np.float64)
        move = solve(K, force, kwargs.get('solver', 'minres'), tol=kwargs.get('tol', 1e-7), maxiter=kwargs.get('maxiter', None),
                     atol=kwargs.get('atol', 0.0), M=kwargs.get('precondition', 'jacobi'), extra_dof_constraint=dof)
        # the held entries of `move` are zero, so (K move)[free] is the free block applied to the free part
        before, after = np.linalg.norm(force[dof]), np.linalg.norm((K @ move - force)[dof])
        if not (after < before and np.any(move[dof] != 0)):
            return False
        M.apply_field(move.reshape(-1, 2)[free], gear[-1], vtx_mask=free)
    return True


def _deformation_beyond_usual(M, gear):
    """per triangle: the larger of its area and edge deformation (0 = none, 1 = flipped), minus the median over the
    normally stiff triangles (multiplier at least half the median one), and that reference set"""
    area = Mesh.svds_to_deform(M.triangle_area_deform(gear=gear).reshape(-1, 1))
    edge = Mesh.svds_to_deform(M.triangle_edge_deform(gear=gear).reshape(-1, 1))
    mult = M.effective_stiffness_multiplier()
    usual = mult >= 0.5 * np.median(mult)
    score = np.maximum(area, edge)
    return score - np.median(score[usual]), usual


def relax_mesh_most_deformed(M, gear=(const.MESH_GEAR_FIXED, const.MESH_GEAR_MOVING), deform_cutoff=const.MAXIMUM_DEFORM_ALLOWED, iqr=0):
    """feabas/optimizer.py:2157-2190.  deform_cutoff < 0: only flipped triangles count and their vertices are freed.
    Otherwise a triangle counts when its deformation exceeds the usual one by more than 1 - 1 / (cutoff + 1) (with
    iqr > 0 also capped at the upper quartile + iqr inter-quartile ranges of the normally stiff triangles; never below
    1e-3), and the triangles that lie wholly on vertices of such triangles are relaxed."""
    if deform_cutoff < 0:
        flipped = np.ravel(Mesh.svds_to_deform(M.triangle_area_deform(gear=gear).reshape(-1, 1)) >= 1)
        return relax_mesh(M, free_vertices=np.unique(M.triangles[flipped]), gear=gear) if flipped.any() else False
    score, usual = _deformation_beyond_usual(M, gear)
    limit = max(1 - 1 / (abs(deform_cutoff) + 1), 0)
    if iqr > 0:
        q1, q3 = np.quantile(score[usual], (0.25, 0.75))
        limit = min(limit, max(q1, q3) + iqr * abs(q3 - q1))
    hot = np.ravel(score > max(limit, 1.0e-3))
    if not hot.any():
        return False
    touched = np.zeros(M.num_vertices, dtype=bool)
    touched[M.triangles[hot].ravel()] = True
    return relax_mesh(M, free_triangles=touched[M.triangles].all(axis=1), gear=gear)
