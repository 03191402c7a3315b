"""Random inputs through the HOST-ONLY entries of the library (no device work, ctx = NULL): the round stepper, divide_bbox, the
deformed-mesh geometry, the general-mesh block fits and uncovered areas, signed areas / edge ratios, the strip packer -- checked
against their numpy statements where one exists.  Meant to run against a build with the host code under AddressSanitizer /
UBSan (tools/asan_host.sh): memory errors of the host loops show up here, without a GPU (the GPU pool refuses sanitizer runs)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
from scipy.spatial import Delaunay

from feabas_amd import _lib, common, deformed


def fuzz_schedule(lib, rng, n):
    for _ in range(n):
        k = int(rng.integers(1, 6))
        sp = np.ascontiguousarray(rng.uniform(5, 800, k))
        h = lib.fb_schedule_create(_lib.ptr(sp), k, int(rng.integers(0, 2)), int(rng.integers(0, 3)), int(rng.integers(0, 3)), int(rng.integers(-1, 2)))
        assert h
        steps = 0
        while steps < 200:
            s_, last, pad = C.c_double(), C.c_int(), C.c_int()
            if not lib.fb_schedule_round(h, C.byref(s_), C.byref(last), C.byref(pad)):
                break
            assert s_.value > 0
            redo = C.c_int()
            lib.fb_schedule_advance(h, float(rng.choice([0.0, 0.05, 1.0, 30.0, 500.0, 1e5])), 4.0, C.byref(redo))
            steps += 1
        assert steps < 200, 'the walk must end'
        lib.fb_schedule_destroy(h)
    assert not lib.fb_schedule_create(_lib.ptr(np.zeros(1)), 0, 0, 0, 0, -1)


def fuzz_divide_bbox(lib, rng, n):
    for _ in range(n):
        lo = rng.uniform(-50, 50, 2); ext = rng.uniform(0.6, 900, 2)
        bbox = np.array([lo[0], lo[1], lo[0] + ext[0], lo[1] + ext[1]])
        blk = np.ascontiguousarray(rng.uniform(3, 300, 2)); mnb = np.ascontiguousarray(rng.integers(1, 4, 2), dtype=np.int32)
        shrink = float(rng.choice([1.0, 0.7, 0.35])); rnd = int(rng.integers(0, 2))
        cnt = np.zeros(2, np.int32); stp = np.zeros(2, np.int32)
        assert lib.fb_divide_bbox(None, _lib.ptr(bbox), _lib.ptr(blk), _lib.ptr(mnb), shrink, rnd, _lib.ptr(cnt), _lib.ptr(stp), None, 0, None, 0) == 0
        xs = np.zeros(cnt[0]); ys = np.zeros(cnt[1])
        assert lib.fb_divide_bbox(None, _lib.ptr(bbox), _lib.ptr(blk), _lib.ptr(mnb), shrink, rnd, _lib.ptr(cnt), _lib.ptr(stp), _lib.ptr(xs), xs.size, _lib.ptr(ys), ys.size) == 0
        x0, y0, x1, y1 = common.divide_bbox(bbox, block_size=(blk[0], blk[1]), min_num_blocks=(int(mnb[0]), int(mnb[1])), round_output=bool(rnd), shrink_factor=shrink)
        np.testing.assert_allclose(np.unique(x0), np.unique(xs), atol=1e-9); np.testing.assert_allclose(np.unique(y0), np.unique(ys), atol=1e-9)
        # a capacity that is too small must be refused, not overrun
        if cnt[0] > 1:
            assert lib.fb_divide_bbox(None, _lib.ptr(bbox), _lib.ptr(blk), _lib.ptr(mnb), shrink, rnd, _lib.ptr(cnt), _lib.ptr(stp), _lib.ptr(xs), int(cnt[0]) - 1, _lib.ptr(ys), ys.size) != 0


def fuzz_general_mesh(lib, rng, n):
    for _ in range(n):
        nv = int(rng.integers(8, 120))
        v = np.ascontiguousarray(rng.uniform(0, 500, (nv, 2)))
        tris = np.ascontiguousarray(Delaunay(v).simplices, dtype=np.int32)
        # a smooth displacement: the triangles of a valid mesh do not overlap (what the uncovered-area sum relies on)
        vm = np.ascontiguousarray(v + 3.0 * np.stack((np.sin(v[:, 1] / 90.0), np.cos(v[:, 0] / 70.0)), axis=-1) + rng.uniform(-5, 5, 2))
        nb = int(rng.integers(0, 12)); h, w = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        org = np.ascontiguousarray(rng.uniform(-60, 520, (nb, 2)))
        cap = int(rng.integers(1, min(40, tris.shape[0]) + 1))
        # candidate lists: distinct triangles in any order (a real list holds every triangle whose box touches the block, once)
        cand = np.ascontiguousarray(np.stack([rng.permutation(tris.shape[0])[:cap] for _ in range(nb)]).reshape(nb, cap), dtype=np.int32) if nb else np.zeros((0, cap), np.int32)
        count = np.ascontiguousarray(rng.integers(0, cap + 1, nb), dtype=np.int32)
        tier = np.full(nb, 3, np.int32); A6 = np.zeros((nb, 6)); unc = np.zeros(nb)
        assert lib.fb_mesh_block_affines(None, nv, _lib.ptr(vm), _lib.ptr(v), _lib.ptr(tris), nb, _lib.ptr(org), h, w, cap, _lib.ptr(cand), _lib.ptr(count),
                                         float(rng.choice([0.0, 0.1, 2.0, 50.0])), _lib.ptr(tier), _lib.ptr(A6)) == 0
        assert set(np.unique(tier)) <= {-1, 2, 3} and np.all(np.isfinite(A6[tier == 2]))
        assert lib.fb_mesh_block_uncovered(None, nv, _lib.ptr(vm), _lib.ptr(tris), nb, _lib.ptr(org), h, w, cap, _lib.ptr(cand), _lib.ptr(count), _lib.ptr(unc)) == 0
        assert np.all(unc >= -1e-6) and np.all(unc <= h * w + 1e-6)
        area = np.zeros(tris.shape[0]); ratio = np.zeros((tris.shape[0], 3))
        tneg = tris.copy(); tneg[::3, 0] -= nv                      # negative indices count from the end, as numpy takes them
        assert lib.fb_signed_area(None, nv, _lib.ptr(v), tris.shape[0], _lib.ptr(np.ascontiguousarray(tneg)), _lib.ptr(area)) == 0
        p = v[tris]
        d1, d2 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 1]
        np.testing.assert_allclose(area, d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0], rtol=1e-12, atol=1e-9)
        assert lib.fb_tri_edge_ratio(None, nv, _lib.ptr(v), _lib.ptr(vm), tris.shape[0], _lib.ptr(tris), _lib.ptr(ratio)) == 0
        assert np.all(ratio > 0)


def fuzz_deformed(lib, rng, n):
    from oracle import pipeline_ref
    for _ in range(n):
        W, H = int(rng.integers(60, 260)), int(rng.integers(200, 1200))
        ms = float(rng.uniform(25, 90))
        v, tri, xs, ys = pipeline_ref.cartesian_mesh(W, H, ms)
        Q = int(rng.integers(1, 4))
        vm = np.ascontiguousarray(v[None] + rng.normal(0, 1.5, (Q,) + v.shape) + rng.uniform(-6, 6, (Q, 1, 2)))
        nblk = int(rng.integers(1, 9)); bh, bw = int(rng.integers(8, 70)), int(rng.integers(8, 70))
        o = np.stack((rng.integers(-40, W + 20, (Q, nblk)), rng.integers(-40, H + 20, (Q, nblk))), axis=-1)
        bb = np.ascontiguousarray(np.concatenate((o, o + np.array([bw, bh])), axis=-1), dtype=np.int32)
        tier = np.empty((Q, nblk), np.int32); A6 = np.empty((Q, nblk, 6)); lo = np.empty((Q, 2))
        tol = float(rng.choice([0.05, 0.5, 4.0]))
        assert lib.fb_deformed_block_affines(None, Q, xs.size, ys.size, _lib.ptr(xs), _lib.ptr(ys), 0, _lib.ptr(vm), nblk, _lib.ptr(bb), tol, None,
                                             _lib.ptr(tier), _lib.ptr(A6), _lib.ptr(lo)) == 0
        for q in range(Q):
            t2, _, _ = deformed.block_affines(vm[q], v, tri, bb[q], tol)
            ok = tier[q] != -1
            np.testing.assert_array_equal(t2[ok], tier[q][ok])
        K = int(rng.integers(0, 400))
        po = np.ascontiguousarray(rng.integers(0, Q, K), dtype=np.int32)
        pts = np.ascontiguousarray(np.stack((rng.uniform(-30, W + 30, K), rng.uniform(-30, H + 30, K)), -1))
        tid = np.empty(max(K, 1), np.int32); B = np.empty((max(K, 1), 3))
        assert lib.fb_deformed_locate(None, Q, xs.size, ys.size, _lib.ptr(xs), _lib.ptr(ys), 0, _lib.ptr(vm), K, _lib.ptr(po), _lib.ptr(pts), _lib.ptr(tid), _lib.ptr(B)) == 0
        assert np.all(tid[:K] < tri.shape[0])
        NB = int(rng.integers(1, 5))
        pair_of = np.ascontiguousarray(rng.integers(0, Q, NB), dtype=np.int32)
        org = np.ascontiguousarray(np.stack((rng.integers(-20, W, NB), rng.integers(-20, H, NB)), -1), dtype=np.int32)
        mx = np.empty((NB, bh, bw)); my = np.empty((NB, bh, bw)); mk = np.empty((NB, bh, bw), np.uint8)
        assert lib.fb_deformed_exact_field(None, Q, xs.size, ys.size, _lib.ptr(xs), _lib.ptr(ys), 0, _lib.ptr(vm), NB, _lib.ptr(pair_of), _lib.ptr(org), bh, bw,
                                           _lib.ptr(mx), _lib.ptr(my), _lib.ptr(mk)) == 0
        assert set(np.unique(mk)) <= {0, 1}


def fuzz_pack(lib, rng, n):
    for _ in range(n):
        k = int(rng.integers(0, 9)); H, W = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        srcs_np = []
        for _ in range(k):
            big = rng.integers(0, 255, (int(rng.integers(1, H + 1)) + 3, int(rng.integers(1, W + 1)) + 5), dtype=np.uint8)
            srcs_np.append(big[:big.shape[0] - 3, :big.shape[1] - 5])           # a view with a pitch larger than its width
        dst = np.full((max(k, 1), H, W), 255, np.uint8)
        srcs = (C.c_void_p * max(k, 1))(*[s.ctypes.data for s in srcs_np])
        hs = np.array([s.shape[0] for s in srcs_np] or [1], dtype=np.int32); ws = np.array([s.shape[1] for s in srcs_np] or [1], dtype=np.int32)
        pit = np.array([s.strides[0] for s in srcs_np] or [1], dtype=np.int64)
        assert lib.fb_host_pack2d(None, _lib.ptr(dst), k, H, W, srcs, _lib.ptr(hs), _lib.ptr(ws), _lib.ptr(pit), int(rng.integers(1, 5))) == 0
        for j, s in enumerate(srcs_np):
            np.testing.assert_array_equal(dst[j, :s.shape[0], :s.shape[1]], s)


def fuzz_multigrid(lib, rng, n):
    """random levels through fb_debug_mg_coarsen (the host half of the multigrid set-up): the invariants of
    tests/test_cpu_host.py::check_mg_coarsening"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    import test_cpu_host as T
    for _ in range(n):
        grids = [(int(rng.integers(2, 60)), int(rng.integers(2, 60)), float(rng.uniform(2, 30)), float(rng.uniform(-500, 500)) + 3000.0 * k, float(rng.uniform(-500, 500)))
                 for k in range(int(rng.integers(1, 4)))]
        xy, comp, A = T._mg_level(rng, grids, jitter=float(rng.choice([0.0, 0.2, 0.45])))
        r = T._mg_coarsen(lib, xy, comp, A, bs=int(rng.choice([2, 3])), fine_scale=float(rng.uniform(0.5, 20)))
        T.check_mg_coarsening(xy, comp, A, r)


def fuzz_strip_host(lib, rng, n):
    """the host arithmetic of fb_match_strips: rigid fits of random match tables against common.fit_affine, spacings, node grids"""
    from feabas_amd import matcher
    from feabas_amd.stitch_pipeline import grid_counts
    for _ in range(n):
        P = int(rng.integers(1, 12))
        pid, p0, p1, wt = [], [], [], []
        for p in range(P):
            k = int(rng.choice([0, 1, 2, 3, 4, 17, 80]))
            q = rng.uniform(-300, 900, (k, 2))
            th = rng.uniform(-0.3, 0.3)
            Rm = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
            t = q @ Rm + rng.uniform(-50, 50, 2) + rng.normal(0, float(rng.choice([0.0, 0.5, 5.0])), (k, 2))
            pid.append(np.full(k, p, np.int32)); p1.append(q); p0.append(t); wt.append(rng.uniform(0.05, 1.0, k).astype(np.float32))
        pid = np.ascontiguousarray(np.concatenate(pid)); p0 = np.ascontiguousarray(np.concatenate(p0).reshape(-1, 2)); p1 = np.ascontiguousarray(np.concatenate(p1).reshape(-1, 2))
        wt = np.ascontiguousarray(np.concatenate(wt))
        R = np.empty((P, 3, 3)); bad = np.empty(P, np.uint8)
        assert lib.fb_debug_rigid_fits(P, pid.size, _lib.ptr(pid), _lib.ptr(p0), _lib.ptr(p1), _lib.ptr(wt), _lib.ptr(R), _lib.ptr(bad)) == 0
        for p in range(P):
            s_ = pid == p
            if bad[p] or s_.sum() < 3:
                continue
            _, Rr = common.fit_affine(p0[s_], p1[s_], return_rigid=True, weight=wt[s_].astype(np.float64), svd_clip=(1, 1))
            np.testing.assert_allclose(R[p], Rr, atol=1e-7, rtol=1e-7)
        H, W = int(rng.integers(30, 5000)), int(rng.integers(30, 5000))
        cnt = C.c_int(); out = np.empty(16)
        assert lib.fb_debug_auto_spacings(H, W, _lib.ptr(out), 16, C.byref(cnt)) == 0
        np.testing.assert_allclose(out[:cnt.value], np.sort(matcher.auto_spacings((H, W), (H, W)))[::-1], rtol=1e-14)
        assert lib.fb_debug_auto_spacings(H, W, _lib.ptr(out), cnt.value - 1, C.byref(cnt)) != 0          # a capacity that is too small is refused
        ms = float(rng.uniform(20, 400)); mnb = int(rng.integers(1, 4))
        nx, ny = C.c_int(), C.c_int()
        assert lib.fb_debug_grid_counts(H, W, ms, mnb, C.byref(nx), C.byref(ny)) == 0
        assert (nx.value, ny.value) == grid_counts(H, W, ms, mnb)


def fuzz_large(lib, rng):
    """sizes at which the host loops go to several threads (>= 4096 blocks, >= 8192 points, >= 131072 triangles): the same checks"""
    from oracle import pipeline_ref
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    import test_cpu_host as T
    xy, comp, A = T._mg_level(rng, [(640, 500, 5.0, 0.0, 0.0), (150, 160, 5.0, 5000.0, 100.0)])
    T.check_mg_coarsening(xy, comp, A, T._mg_coarsen(lib, xy, comp, A))
    nv = 70000
    v = np.ascontiguousarray(rng.uniform(0, 9000, (nv, 2)))
    tris = np.ascontiguousarray(Delaunay(v).simplices, dtype=np.int32)
    vm = np.ascontiguousarray(v + 3.0 * np.stack((np.sin(v[:, 1] / 300.0), np.cos(v[:, 0] / 260.0)), axis=-1))
    # (the slivers Delaunay leaves on the hull of random points fold under any displacement: a valid mesh has none)
    pm = vm[tris]; e1, e2 = pm[:, 1] - pm[:, 0], pm[:, 2] - pm[:, 1]
    pi = v[tris]; f1, f2 = pi[:, 1] - pi[:, 0], pi[:, 2] - pi[:, 1]
    keep = ((e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]) * np.sign(f1[:, 0] * f2[:, 1] - f1[:, 1] * f2[:, 0]) > 1.0) & (np.abs(f1[:, 0] * f2[:, 1] - f1[:, 1] * f2[:, 0]) > 1.0)
    tris = np.ascontiguousarray(tris[keep])
    assert tris.shape[0] > 131072
    area = np.zeros(tris.shape[0]); ratio = np.zeros((tris.shape[0], 3))
    assert lib.fb_signed_area(None, nv, _lib.ptr(v), tris.shape[0], _lib.ptr(tris), _lib.ptr(area)) == 0
    p = v[tris]
    d1, d2 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 1]
    np.testing.assert_allclose(area, d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0], rtol=1e-12, atol=1e-9)
    assert lib.fb_tri_edge_ratio(None, nv, _lib.ptr(v), _lib.ptr(vm), tris.shape[0], _lib.ptr(tris), _lib.ptr(ratio)) == 0
    q = vm[tris]
    for k in range(3):
        e0 = p[:, k] - p[:, k - 1]; e1 = q[:, k] - q[:, k - 1]
        np.testing.assert_allclose(ratio[:, k], np.sum(e1 * e1, axis=1) / np.sum(e0 * e0, axis=1), rtol=1e-12)
    nb, h, w, cap = 6000, 40, 50, 12
    org = np.ascontiguousarray(rng.uniform(0, 8900, (nb, 2)))
    # candidates: the triangles nearest to the block centre (distinct by construction)
    from scipy.spatial import cKDTree
    cen = p.mean(axis=1)
    _, near = cKDTree(cen).query(org + np.array([w / 2, h / 2]), k=cap)
    cand = np.ascontiguousarray(near, dtype=np.int32); count = np.full(nb, cap, np.int32)
    tier = np.full(nb, 3, np.int32); A6 = np.zeros((nb, 6)); unc = np.zeros(nb)
    assert lib.fb_mesh_block_affines(None, nv, _lib.ptr(vm), _lib.ptr(v), _lib.ptr(tris), nb, _lib.ptr(org), h, w, cap, _lib.ptr(cand), _lib.ptr(count), 0.5,
                                     _lib.ptr(tier), _lib.ptr(A6)) == 0
    assert lib.fb_mesh_block_uncovered(None, nv, _lib.ptr(vm), _lib.ptr(tris), nb, _lib.ptr(org), h, w, cap, _lib.ptr(cand), _lib.ptr(count), _lib.ptr(unc)) == 0
    assert np.all(unc >= -1e-6) and np.all(unc <= h * w + 1e-6) and set(np.unique(tier)) <= {-1, 2, 3}
    # the threaded loops must give what one thread gives: the same call on the first 1000 blocks alone
    t1 = np.full(1000, 3, np.int32); A1 = np.zeros((1000, 6)); u1 = np.zeros(1000)
    lib.fb_mesh_block_affines(None, nv, _lib.ptr(vm), _lib.ptr(v), _lib.ptr(tris), 1000, _lib.ptr(org), h, w, cap, _lib.ptr(cand), _lib.ptr(count), 0.5, _lib.ptr(t1), _lib.ptr(A1))
    lib.fb_mesh_block_uncovered(None, nv, _lib.ptr(vm), _lib.ptr(tris), 1000, _lib.ptr(org), h, w, cap, _lib.ptr(cand), _lib.ptr(count), _lib.ptr(u1))
    np.testing.assert_array_equal(t1, tier[:1000]); np.testing.assert_array_equal(A1, A6[:1000]); np.testing.assert_array_equal(u1, unc[:1000])
    W, H = 240, 3000
    vg, tri, xs, ys = pipeline_ref.cartesian_mesh(W, H, 30.0)
    Q, nblk = 3, 1500
    vmq = np.ascontiguousarray(vg[None] + rng.normal(0, 1.0, (Q,) + vg.shape))
    o = np.stack((rng.integers(-20, W, (Q, nblk)), rng.integers(-20, H, (Q, nblk))), axis=-1)
    bb = np.ascontiguousarray(np.concatenate((o, o + np.array([40, 36])), axis=-1), dtype=np.int32)
    tierq = np.empty((Q, nblk), np.int32); A6q = np.empty((Q, nblk, 6)); lo = np.empty((Q, 2))
    assert lib.fb_deformed_block_affines(None, Q, xs.size, ys.size, _lib.ptr(xs), _lib.ptr(ys), 0, _lib.ptr(vmq), nblk, _lib.ptr(bb), 0.4, None,
                                         _lib.ptr(tierq), _lib.ptr(A6q), _lib.ptr(lo)) == 0
    t2, _, _ = deformed.block_affines(vmq[1], vg, tri, bb[1], 0.4)
    ok = tierq[1] != -1
    np.testing.assert_array_equal(t2[ok], tierq[1][ok])
    K = 30000
    po = np.ascontiguousarray(rng.integers(0, Q, K), dtype=np.int32)
    pts = np.ascontiguousarray(np.stack((rng.uniform(-10, W + 10, K), rng.uniform(-10, H + 10, K)), -1))
    tid = np.empty(K, np.int32); B = np.empty((K, 3))
    assert lib.fb_deformed_locate(None, Q, xs.size, ys.size, _lib.ptr(xs), _lib.ptr(ys), 0, _lib.ptr(vmq), K, _lib.ptr(po), _lib.ptr(pts), _lib.ptr(tid), _lib.ptr(B)) == 0
    s_ = po == 2
    tq, Bq = deformed.locate(vmq[2], tri, xs, ys, pts[s_])
    np.testing.assert_array_equal(tq, tid[s_])
    np.testing.assert_allclose(B[s_][tq >= 0], Bq[tq >= 0], atol=1e-12)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    lib = _lib.load_test()
    print('library', _lib.LIB_PATH)
    rng = np.random.default_rng(seed)
    for name, fn in (('schedule', fuzz_schedule), ('divide_bbox', fuzz_divide_bbox), ('general_mesh', fuzz_general_mesh), ('deformed', fuzz_deformed),
                     ('pack', fuzz_pack), ('multigrid', fuzz_multigrid), ('strip_host', fuzz_strip_host)):
        fn(lib, rng, rounds)
        print(f'{name}: {rounds} rounds ok')
    fuzz_large(lib, rng)
    print('large (threaded host loops): ok')
    for n_ in list(range(0, 70)) + [509, 1000, 4095, 8191, 100000]:
        f = lib.fb_next_fast_len(n_)
        assert f >= n_ and (n_ <= 6 or all(f % p for p in (7, 11, 13)))
    print('done')


if __name__ == '__main__':
    main()
