// Host-side geometry of tile pairs whose mesh1 is deformed (feabas_amd/deformed.py states the same functions in numpy
// and is their test oracle on the CPU): the tier decision of MeshRenderer.crop_field with the affine approximator of
// MeshRenderer.from_mesh (feabas/renderer.py:90-109, 397-416, 453-511) for every block of a batch of pairs, and
// Mesh.tri_finder + cart2bary (feabas/mesh.py:2080-2217) on the deformed cartesian mesh.  The reference keeps this on
// the host too (shapely STRtree + numpy lstsq + matplotlib trifinder); here it is plain C++ so that the host threads
// that drive the device do it without the Python interpreter lock.  No device work, ctx may be NULL.
#include "fb_common.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <thread>
#include <vector>

namespace {

struct Fit {
    double a00, a01, a10, a11, t0, t1, res;   // image = moving @ [[a00 a01] [a10 a11]] + (t0, t1)
    bool ok;
};

// spatial.fit_affine(pts0 = v_init[idx], pts1 = vm[idx]) (spatial.py:21-73, unweighted): on centred points the least
// squares is block diagonal -- the 2x2 part solves the normal equations, the translation is mm0 - mm1 @ A
Fit fit_rows(const double* vm, const double* vi, const int* idx, int cnt) {
    Fit f{1, 0, 0, 1, 0, 0, std::numeric_limits<double>::infinity(), false};
    if (cnt < 3) return f;
    double m0x = 0, m0y = 0, m1x = 0, m1y = 0;
    for (int k = 0; k < cnt; ++k) {
        const int v = idx[k];
        m0x += vi[2 * v]; m0y += vi[2 * v + 1]; m1x += vm[2 * v]; m1y += vm[2 * v + 1];
    }
    m0x /= cnt; m0y /= cnt; m1x /= cnt; m1y /= cnt;
    double g00 = 0, g01 = 0, g11 = 0, h00 = 0, h01 = 0, h10 = 0, h11 = 0;
    for (int k = 0; k < cnt; ++k) {
        const int v = idx[k];
        const double c1x = vm[2 * v] - m1x, c1y = vm[2 * v + 1] - m1y, c0x = vi[2 * v] - m0x, c0y = vi[2 * v + 1] - m0y;
        g00 += c1x * c1x; g01 += c1x * c1y; g11 += c1y * c1y;
        h00 += c1x * c0x; h01 += c1x * c0y; h10 += c1y * c0x; h11 += c1y * c0y;
    }
    const double det = g00 * g11 - g01 * g01, scale = std::max(g00, g11);
    if (!(det > 1e-9 * scale * scale)) return f;
    f.a00 = (g11 * h00 - g01 * h10) / det; f.a01 = (g11 * h01 - g01 * h11) / det;
    f.a10 = (g00 * h10 - g01 * h00) / det; f.a11 = (g00 * h11 - g01 * h01) / det;
    if (!(f.a00 * f.a11 - f.a01 * f.a10 > 0)) return f;      // flipped: spatial.py:48-60 takes another route
    f.t0 = m0x - (m1x * f.a00 + m1y * f.a10);
    f.t1 = m0y - (m1x * f.a01 + m1y * f.a11);
    double worst = 0;
    for (int k = 0; k < cnt; ++k) {
        const int v = idx[k];
        const double dx = vi[2 * v] - (vm[2 * v] * f.a00 + vm[2 * v + 1] * f.a10 + f.t0);
        const double dy = vi[2 * v + 1] - (vm[2 * v] * f.a01 + vm[2 * v + 1] * f.a11 + f.t1);
        worst = std::max(worst, dx * dx + dy * dy);
    }
    f.res = std::sqrt(worst);
    f.ok = true;
    return f;
}

// closed triangle against closed box: separating axes = the two box axes and the three edge normals
bool tri_hits_box(const double* p0, const double* p1, const double* p2, double bx0, double by0, double bx1, double by1) {
    const double tx0 = std::min(p0[0], std::min(p1[0], p2[0])), tx1 = std::max(p0[0], std::max(p1[0], p2[0]));
    const double ty0 = std::min(p0[1], std::min(p1[1], p2[1])), ty1 = std::max(p0[1], std::max(p1[1], p2[1]));
    if (tx1 < bx0 || tx0 > bx1 || ty1 < by0 || ty0 > by1) return false;
    const double* tp[3] = {p0, p1, p2};
    const double cx[4] = {bx0, bx1, bx1, bx0}, cy[4] = {by0, by0, by1, by1};
    for (int k = 0; k < 3; ++k) {
        const double ex = tp[(k + 1) % 3][0] - tp[k][0], ey = tp[(k + 1) % 3][1] - tp[k][1];
        const double nx = -ey, ny = ex;
        double tmin = std::numeric_limits<double>::infinity(), tmax = -tmin, bmin = tmin, bmax = -tmin;
        for (int a = 0; a < 3; ++a) { const double d = tp[a][0] * nx + tp[a][1] * ny; tmin = std::min(tmin, d); tmax = std::max(tmax, d); }
        for (int a = 0; a < 4; ++a) { const double d = cx[a] * nx + cy[a] * ny; bmin = std::min(bmin, d); bmax = std::max(bmax, d); }
        if (tmax < bmin || tmin > bmax) return false;
    }
    return true;
}

inline void tri_nodes(int t, int nx, int* n3) {
    const int cell = t >> 1, j = cell / (nx - 1), i = cell - j * (nx - 1), a = j * nx + i;
    n3[0] = a;
    if (t & 1) { n3[1] = a + nx + 1; n3[2] = a + nx; } else { n3[1] = a + 1; n3[2] = a + nx + 1; }
}

}  // namespace

extern "C" {

// vm [Q][V][2]: MOVING vertices (with offset) of mesh1 of every pair, V = nx ny nodes of the grid xs x ys (the INITIAL
// vertices); bboxes [Q][nblk][4] int32 in the MOVING frame.  tier [Q][nblk]: 1 global affine, 2 block affine, 3 exact
// field, -1 = the block's vertex set is degenerate or its fit flipped (caller takes the statement-by-statement route);
// A6 [Q][nblk][6] = {A00, A10, t0, A01, A11, t1} (image x = X A00 + Y A10 + t0, image y = X A01 + Y A11 + t1);
// lo [Q][2] = smallest image x / y any affine block of the pair samples (+inf without one): the remap origin of
// render_by_subregions (common.py:316-321) is floor(lo) - 4 once the exact-field blocks are included.
int fb_deformed_block_affines(fb_ctx* ctx, int Q, int nx, int ny, const double* xs_all, const double* ys_all, int per_pair_grid,
                              const double* vm, int nblk, const int32_t* bboxes, double tol_all, const double* tol_each, int32_t* tier,
                              double* A6, double* lo) {
    FB_CHECK_ARG(ctx, Q >= 0 && nx >= 2 && ny >= 2 && xs_all && ys_all && vm && nblk >= 0 && bboxes && tier && A6 && lo);
    const int V = nx * ny;
    const double inf = std::numeric_limits<double>::infinity();
    // pairs are independent: a batch of deformed pairs (24 k blocks at 64 pairs of the 4k configuration) is dealt to a few host
    // threads -- the calling thread holds the device idle while this runs
    auto work = [&](int q_lo, int q_hi) {
    std::vector<double> vi(2 * (size_t)V);
    std::vector<int> all(V), stamp(V), members;
    for (int v = 0; v < V; ++v) all[v] = v;
    members.reserve(64);
    for (int q = q_lo; q < q_hi; ++q) {
        // the INITIAL node grid of pair q (pairs of unequal strip size have their own)
        const double* xs = xs_all + (per_pair_grid ? (size_t)q * nx : 0);
        const double* ys = ys_all + (per_pair_grid ? (size_t)q * ny : 0);
        const double tol = tol_each ? tol_each[q] : tol_all;
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) { vi[2 * (j * nx + i)] = xs[i]; vi[2 * (j * nx + i) + 1] = ys[j]; }
        const double* v = vm + 2 * (size_t)q * V;
        int32_t* tq = tier + (size_t)q * nblk;
        double* aq = A6 + 6 * (size_t)q * nblk;
        const int32_t* bq = bboxes + 4 * (size_t)q * nblk;
        double lox = inf, loy = inf;
        auto put = [&](int b, const Fit& f) {
            double* a = aq + 6 * (size_t)b;
            a[0] = f.a00; a[1] = f.a10; a[2] = f.t0; a[3] = f.a01; a[4] = f.a11; a[5] = f.t1;
            const double X[2] = {(double)bq[4 * b], (double)bq[4 * b + 2] - 1.0}, Y[2] = {(double)bq[4 * b + 1], (double)bq[4 * b + 3] - 1.0};
            for (int cxi = 0; cxi < 2; ++cxi)
                for (int cyi = 0; cyi < 2; ++cyi) {
                    lox = std::min(lox, X[cxi] * f.a00 + Y[cyi] * f.a10 + f.t0);
                    loy = std::min(loy, X[cxi] * f.a01 + Y[cyi] * f.a11 + f.t1);
                }
        };
        Fit g{1, 0, 0, 1, 0, 0, inf, false};
        if (tol > 0) g = fit_rows(v, vi.data(), all.data(), V);
        if (tol > 0 && g.ok && g.res < tol) {
            for (int b = 0; b < nblk; ++b) { tq[b] = 1; put(b, g); }
        } else {
            double uminx = inf, umaxx = -inf, uminy = inf, umaxy = -inf;
            for (int k = 0; k < V; ++k) {
                const double ux = v[2 * k] - vi[2 * k], uy = v[2 * k + 1] - vi[2 * k + 1];
                uminx = std::min(uminx, ux); umaxx = std::max(umaxx, ux); uminy = std::min(uminy, uy); umaxy = std::max(umaxy, uy);
            }
            std::fill(stamp.begin(), stamp.end(), -1);
            for (int b = 0; b < nblk; ++b) {
                const double bx0 = bq[4 * b] - 0.5, by0 = bq[4 * b + 1] - 0.5, bx1 = bq[4 * b + 2] - 0.5, by1 = bq[4 * b + 3] - 0.5;
                // cells whose deformed extent can reach the box
                int ilo = 0, ihi = nx - 2, jlo = 0, jhi = ny - 2;
                while (ilo < nx - 2 && xs[ilo + 1] + umaxx < bx0) ++ilo;
                while (ihi > 0 && xs[ihi] + uminx > bx1) --ihi;
                while (jlo < ny - 2 && ys[jlo + 1] + umaxy < by0) ++jlo;
                while (jhi > 0 && ys[jhi] + uminy > by1) --jhi;
                members.clear();
                for (int j = jlo; j <= jhi; ++j)
                    for (int i = ilo; i <= ihi; ++i)
                        for (int half = 0; half < 2; ++half) {
                            int n3[3];
                            tri_nodes(2 * (j * (nx - 1) + i) + half, nx, n3);
                            if (!tri_hits_box(v + 2 * n3[0], v + 2 * n3[1], v + 2 * n3[2], bx0, by0, bx1, by1)) continue;
                            for (int a = 0; a < 3; ++a)
                                if (stamp[n3[a]] != b) { stamp[n3[a]] = b; members.push_back(n3[a]); }
                        }
                tq[b] = 3;
                double* a = aq + 6 * (size_t)b;
                a[0] = 1; a[1] = 0; a[2] = 0; a[3] = 0; a[4] = 1; a[5] = 0;
                if (!(tol > 0) || members.empty()) continue;
                std::sort(members.begin(), members.end());
                const Fit f = fit_rows(v, vi.data(), members.data(), (int)members.size());
                if (!f.ok) { tq[b] = -1; continue; }
                if (f.res < tol) { tq[b] = 2; put(b, f); }
            }
        }
        lo[2 * q] = lox; lo[2 * q + 1] = loy;
    }
    };
    const int T = std::max(1, std::min(fb_host_threads(4), (int)(((int64_t)Q * nblk) / 2048)));
    if (T <= 1) work(0, Q);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) pool.emplace_back(work, (int)((int64_t)Q * t / T), (int)((int64_t)Q * (t + 1) / T));
        for (auto& th : pool) th.join();
    }
    return FB_OK;
}

// The same tier decision for GENERAL triangulated meshes (renderer.MeshRenderer): block b = h x w pixels whose first pixel
// sits at org [NB][2] (MOVING frame, offset removed); its candidate triangles cand [NB][cap] / count [NB] come from
// fb_mesh_candidates_dev (host copies).  v_mov / v_img [V][2]: field domain and image-space vertices.  tier [NB]: 2 = block
// affine with residue < tol (A6 filled), 3 = exact field, -1 = degenerate / flipped fit (the caller's statement-by-statement
// route).  The global-affine test (tier 1) is the caller's (one fit per renderer).
int fb_mesh_block_affines(fb_ctx* ctx, int V, const double* v_mov, const double* v_img, const int32_t* tris, int NB, const double* org,
                          int h, int w, int cap, const int32_t* cand, const int32_t* count, double tol, int32_t* tier, double* A6) {
    FB_CHECK_ARG(ctx, V > 0 && v_mov && v_img && tris && NB >= 0 && org && h > 0 && w > 0 && cap > 0 && cand && count && tier && A6);
    auto work = [&](int b_lo, int b_hi) {
    std::vector<int> stamp((size_t)V, -1), members;
    members.reserve(256);
    for (int b = b_lo; b < b_hi; ++b) {
        const double bx0 = org[2 * b] - 0.5, by0 = org[2 * b + 1] - 0.5, bx1 = bx0 + (double)w, by1 = by0 + (double)h;   // bbox0 - 0.5, renderer.py:405
        members.clear();
        const int nc = std::min(count[b], cap);
        for (int k = 0; k < nc; ++k) {
            const int32_t* t3 = tris + 3 * (size_t)cand[(size_t)b * cap + k];
            if (!tri_hits_box(v_mov + 2 * (size_t)t3[0], v_mov + 2 * (size_t)t3[1], v_mov + 2 * (size_t)t3[2], bx0, by0, bx1, by1)) continue;
            for (int a = 0; a < 3; ++a)
                if (stamp[t3[a]] != b) { stamp[t3[a]] = b; members.push_back(t3[a]); }
        }
        tier[b] = 3;
        double* a = A6 + 6 * (size_t)b;
        a[0] = 1; a[1] = 0; a[2] = 0; a[3] = 0; a[4] = 1; a[5] = 0;
        if (!(tol > 0) || members.empty()) continue;
        std::sort(members.begin(), members.end());
        const Fit f = fit_rows(v_mov, v_img, members.data(), (int)members.size());
        if (!f.ok) { tier[b] = -1; continue; }
        if (f.res < tol) { tier[b] = 2; a[0] = f.a00; a[1] = f.a10; a[2] = f.t0; a[3] = f.a01; a[4] = f.a11; a[5] = f.t1; }
    }
    };
    // blocks are independent (their own vertex stamps per thread): a few host threads for the block counts of a whole section
    const int NT = std::max(1, std::min(fb_host_threads(4), NB / 2048));
    if (NT <= 1) work(0, NB);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < NT; ++t) pool.emplace_back(work, (int)((int64_t)NB * t / NT), (int)((int64_t)NB * (t + 1) / NT));
        for (auto& th : pool) th.join();
    }
    return FB_OK;
}

// Area of every block's box (bbox0 - 0.5, renderer.py:438) that no candidate triangle covers, in MOVING coordinates: what the
// precise mask of crop_field_affine (renderer.py:437-447) compares with 1 px^2.  The reference takes the area in image
// space, between the affine image of the box and the mesh region eroded by 0.5 px (shapely); here every candidate
// triangle is clipped to the box (Sutherland-Hodgman against the four box sides) and the areas are summed -- the
// triangles of a valid mesh do not overlap.  uncovered [NB] >= 1 selects the per-pixel mask (tier + 10).
int fb_mesh_block_uncovered(fb_ctx* ctx, int V, const double* v_mov, const int32_t* tris, int NB, const double* org, int h, int w, int cap,
                            const int32_t* cand, const int32_t* count, double* uncovered) {
    FB_CHECK_ARG(ctx, V > 0 && v_mov && tris && NB >= 0 && org && h > 0 && w > 0 && cap > 0 && cand && count && uncovered);
    auto work = [&](int b_lo, int b_hi) {
    for (int b = b_lo; b < b_hi; ++b) {
        const double bx0 = org[2 * b] - 0.5, by0 = org[2 * b + 1] - 0.5, bx1 = bx0 + (double)w, by1 = by0 + (double)h;
        double covered = 0.0;
        const int nc = std::min(count[b], cap);
        for (int k = 0; k < nc; ++k) {
            const int32_t* t3 = tris + 3 * (size_t)cand[(size_t)b * cap + k];
            double px[16], py[16], qx[16], qy[16];
            int n = 3;
            for (int a = 0; a < 3; ++a) { px[a] = v_mov[2 * (size_t)t3[a]]; py[a] = v_mov[2 * (size_t)t3[a] + 1]; }
            // clip against x >= bx0, x <= bx1, y >= by0, y <= by1
            for (int side = 0; side < 4 && n > 0; ++side) {
                int m = 0;
                for (int i = 0; i < n; ++i) {
                    const int j = (i + 1) % n;
                    const double ax = px[i], ay = py[i], cx = px[j], cy = py[j];
                    const double da = side == 0 ? ax - bx0 : side == 1 ? bx1 - ax : side == 2 ? ay - by0 : by1 - ay;
                    const double dc = side == 0 ? cx - bx0 : side == 1 ? bx1 - cx : side == 2 ? cy - by0 : by1 - cy;
                    if (da >= 0) { qx[m] = ax; qy[m] = ay; ++m; }
                    if ((da >= 0) != (dc >= 0)) { const double s_ = da / (da - dc); qx[m] = ax + s_ * (cx - ax); qy[m] = ay + s_ * (cy - ay); ++m; }
                }
                n = m;
                for (int i = 0; i < n; ++i) { px[i] = qx[i]; py[i] = qy[i]; }
            }
            double a2 = 0.0;
            for (int i = 0; i < n; ++i) { const int j = (i + 1) % n; a2 += px[i] * py[j] - px[j] * py[i]; }
            covered += 0.5 * std::fabs(a2);
        }
        uncovered[b] = (double)w * (double)h - covered;
    }
    };
    // blocks are independent: a few host threads for the block counts of a whole section (13 689 blocks: 5 ms on one)
    const int T = std::max(1, std::min(fb_host_threads(4), NB / 2048));
    if (T <= 1) work(0, NB);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) pool.emplace_back(work, (int)((int64_t)NB * t / T), (int)((int64_t)NB * (t + 1) / T));
        for (auto& th : pool) th.join();
    }
    return FB_OK;
}

// common.signed_area (common.py:672-676): cross(p1 - p0, p2 - p1) of every triangle, the two products and their difference
// rounded one by one like the numpy statement (no fused multiply-add), so the result is the same bit for bit.  Host arrays.
int fb_signed_area(fb_ctx* ctx, int V, const double* v, int T, const int32_t* tris, double* area) {
    FB_CHECK_ARG(ctx, V >= 0 && T >= 0 && (T == 0 || (v && tris && area)));
    for (int t = 0; t < T; ++t)
        for (int a = 0; a < 3; ++a) FB_CHECK_ARG(ctx, tris[3 * (size_t)t + a] >= -V && tris[3 * (size_t)t + a] < V);
    auto work = [&](int lo, int hi) {
#pragma clang fp contract(off)
        for (int t = lo; t < hi; ++t) {
            const int32_t* t3 = tris + 3 * (size_t)t;
            const size_t i0 = (size_t)(t3[0] < 0 ? t3[0] + V : t3[0]), i1 = (size_t)(t3[1] < 0 ? t3[1] + V : t3[1]), i2 = (size_t)(t3[2] < 0 ? t3[2] + V : t3[2]);
            const double ax = v[2 * i1] - v[2 * i0], ay = v[2 * i1 + 1] - v[2 * i0 + 1];
            const double bx = v[2 * i2] - v[2 * i1], by = v[2 * i2 + 1] - v[2 * i1 + 1];
            const double p = ax * by, q = ay * bx;
            area[t] = p - q;
        }
    };
    const int NT = std::max(1, std::min(fb_host_threads(4), T / 65536));
    if (NT <= 1) work(0, T);
    else {
        std::vector<std::thread> pool;
        for (int k = 0; k < NT; ++k) pool.emplace_back(work, (int)((int64_t)T * k / NT), (int)((int64_t)T * (k + 1) / NT));
        for (auto& th : pool) th.join();
    }
    return FB_OK;
}

// Squared-length ratio of every triangle edge between two vertex sets (the gathers of Mesh.triangle_edge_deform,
// mesh.py:1966-1976: edge k runs from vertex k - 1 to vertex k of the triangle): ratio [T][3] = |v1[t_k] - v1[t_k-1]|^2 /
// |v0[t_k] - v0[t_k-1]|^2, every operation rounded on its own like the numpy statement.  Host arrays.
int fb_tri_edge_ratio(fb_ctx* ctx, int V, const double* v0, const double* v1, int T, const int32_t* tris, double* ratio) {
    FB_CHECK_ARG(ctx, V >= 0 && T >= 0 && (T == 0 || (v0 && v1 && tris && ratio)));
    for (int t = 0; t < T; ++t)
        for (int a = 0; a < 3; ++a) FB_CHECK_ARG(ctx, tris[3 * (size_t)t + a] >= 0 && tris[3 * (size_t)t + a] < V);
    for (int t = 0; t < T; ++t) {
#pragma clang fp contract(off)
        const int32_t* t3 = tris + 3 * (size_t)t;
        for (int k = 0; k < 3; ++k) {
            const size_t a = (size_t)t3[k], b = (size_t)t3[(k + 2) % 3];
            const double x0 = v0[2 * a] - v0[2 * b], y0 = v0[2 * a + 1] - v0[2 * b + 1];
            const double x1 = v1[2 * a] - v1[2 * b], y1 = v1[2 * a + 1] - v1[2 * b + 1];
            const double xx0 = x0 * x0, yy0 = y0 * y0, xx1 = x1 * x1, yy1 = y1 * y1;
            ratio[3 * (size_t)t + k] = (xx1 + yy1) / (xx0 + yy0);
        }
    }
    return FB_OK;
}

// field_w_weight (renderer.py:259-300) for NB blocks: block e belongs to pair pair_of[e] (index into vm) and covers the
// h x w output pixels at (x0, y0) = org[e]; every pixel is located in the MOVING triangles that can reach the block and
// mapped to the image by linear interpolation of the INITIAL vertices (matplotlib.tri.LinearTriInterpolator in the
// reference).  map_x, map_y [NB][h][w] float64, mask [NB][h][w] uint8 (0 = outside every triangle).
// deformed.exact_field is the numpy statement of the same computation.
int fb_deformed_exact_field(fb_ctx* ctx, int Q, int nx, int ny, const double* xs_all, const double* ys_all, int per_pair_grid,
                            const double* vm, int NB, const int32_t* pair_of, const int32_t* org, int h, int w, double* map_x, double* map_y,
                            uint8_t* mask) {
    FB_CHECK_ARG(ctx, Q >= 0 && nx >= 2 && ny >= 2 && xs_all && ys_all && vm && NB >= 0 && h > 0 && w > 0 && (NB == 0 || (pair_of && org && map_x && map_y && mask)));
    const int V = nx * ny;
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<int> cand;
    for (int e = 0; e < NB; ++e) {
        const int q = pair_of[e];
        if (q < 0 || q >= Q) return fb_fail(ctx, FB_ERR_ARG, "fb_deformed_exact_field: pair %d outside [0, %d)", q, Q);
        const double* xs = xs_all + (per_pair_grid ? (size_t)q * nx : 0);
        const double* ys = ys_all + (per_pair_grid ? (size_t)q * ny : 0);
        const double* v = vm + 2 * (size_t)q * V;
        double uminx = inf, umaxx = -inf, uminy = inf, umaxy = -inf;
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const double ux = v[2 * (j * nx + i)] - xs[i], uy = v[2 * (j * nx + i) + 1] - ys[j];
                uminx = std::min(uminx, ux); umaxx = std::max(umaxx, ux); uminy = std::min(uminy, uy); umaxy = std::max(umaxy, uy);
            }
        const int x0 = org[2 * e], y0 = org[2 * e + 1];
        const double bx0 = x0 - 0.5, by0 = y0 - 0.5, bx1 = x0 + w - 0.5, by1 = y0 + h - 0.5;
        int ilo = 0, ihi = nx - 2, jlo = 0, jhi = ny - 2;
        while (ilo < nx - 2 && xs[ilo + 1] + umaxx < bx0) ++ilo;
        while (ihi > 0 && xs[ihi] + uminx > bx1) --ihi;
        while (jlo < ny - 2 && ys[jlo + 1] + umaxy < by0) ++jlo;
        while (jhi > 0 && ys[jhi] + uminy > by1) --jhi;
        cand.clear();
        for (int j = jlo; j <= jhi; ++j)                      // ascending triangle id, like np.flatnonzero(hits)
            for (int i = ilo; i <= ihi; ++i)
                for (int half = 0; half < 2; ++half) {
                    const int t = 2 * (j * (nx - 1) + i) + half;
                    int n3[3];
                    tri_nodes(t, nx, n3);
                    if (tri_hits_box(v + 2 * n3[0], v + 2 * n3[1], v + 2 * n3[2], bx0, by0, bx1, by1)) cand.push_back(t);
                }
        const size_t base = (size_t)e * h * w;
        for (int r = 0; r < h; ++r)
            for (int c = 0; c < w; ++c) {
                const double px = (double)x0 + c, py = (double)y0 + r;
                double mx = 0.0, my = 0.0;
                uint8_t inside = 0;
                for (size_t ci = 0; ci < cand.size() && !inside; ++ci) {
                    int n3[3];
                    tri_nodes(cand[ci], nx, n3);
                    const double d0x = px - v[2 * n3[0]], d0y = py - v[2 * n3[0] + 1];
                    const double d1x = px - v[2 * n3[1]], d1y = py - v[2 * n3[1] + 1];
                    const double d2x = px - v[2 * n3[2]], d2y = py - v[2 * n3[2] + 1];
                    const double a0 = d1x * d2y - d1y * d2x, a1 = d2x * d0y - d2y * d0x, a2 = d0x * d1y - d0y * d1x;
                    const double tot = a0 + a1 + a2;
                    const double b0 = a0 / tot, b1 = a1 / tot, b2 = a2 / tot;
                    if (b0 >= 0 && b1 >= 0 && b2 >= 0) {
                        const int i0 = n3[0] % nx, j0 = n3[0] / nx, i1 = n3[1] % nx, j1 = n3[1] / nx, i2 = n3[2] % nx, j2 = n3[2] / nx;
                        mx = b0 * xs[i0] + b1 * xs[i1] + b2 * xs[i2];
                        my = b0 * ys[j0] + b1 * ys[j1] + b2 * ys[j2];
                        inside = 1;
                    }
                }
                map_x[base + (size_t)r * w + c] = mx; map_y[base + (size_t)r * w + c] = my; mask[base + (size_t)r * w + c] = inside;
            }
    }
    return FB_OK;
}

// Mesh.tri_finder + cart2bary on the deformed grid mesh: point k belongs to pair pair_of[k] (index into vm [Q][V][2]).
// The cell is guessed by pulling the point back with the displacement of its nearest node (two passes), then the
// triangles of the 3 x 3 cells around it are tested in a fixed order; the first that contains the point wins.
// tid [K] = triangle (cells (a b / c d) -> 2 cell: (a, b, d), 2 cell + 1: (a, d, c)) or -1 outside; B [K][3] (nan outside).
int fb_deformed_locate(fb_ctx* ctx, int Q, int nx, int ny, const double* xs_all, const double* ys_all, int per_pair_grid, const double* vm,
                       int64_t K, const int32_t* pair_of, const double* pts, int32_t* tid, double* B) {
    FB_CHECK_ARG(ctx, Q >= 0 && nx >= 2 && ny >= 2 && xs_all && ys_all && vm && K >= 0 && (K == 0 || (pair_of && pts && tid && B)));
    const int V = nx * ny;
    const double eps = 1e-9, nan = std::numeric_limits<double>::quiet_NaN(), inf = std::numeric_limits<double>::infinity();
    std::vector<double> mean(2 * (size_t)std::max(Q, 1), 0.0);
    for (int q = 0; q < Q; ++q) {
        const double* xs = xs_all + (per_pair_grid ? (size_t)q * nx : 0);
        const double* ys = ys_all + (per_pair_grid ? (size_t)q * ny : 0);
        const double* v = vm + 2 * (size_t)q * V;
        double mx = 0, my = 0;
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) { mx += v[2 * (j * nx + i)] - xs[i]; my += v[2 * (j * nx + i) + 1] - ys[j]; }
        mean[2 * q] = mx / V; mean[2 * q + 1] = my / V;
    }
    static const int order[3] = {0, -1, 1};
    for (int64_t k = 0; k < K; ++k)
        if (pair_of[k] < 0 || pair_of[k] >= Q) return fb_fail(ctx, FB_ERR_ARG, "fb_deformed_locate: pair %d outside [0, %d)", pair_of[k], Q);
    auto work = [&](int64_t k_lo, int64_t k_hi) {
    for (int64_t k = k_lo; k < k_hi; ++k) {
        const int q = pair_of[k];
        const double* xs = xs_all + (per_pair_grid ? (size_t)q * nx : 0);
        const double* ys = ys_all + (per_pair_grid ? (size_t)q * ny : 0);
        const double* v = vm + 2 * (size_t)q * V;
        const double px = pts[2 * k], py = pts[2 * k + 1];
        double qx = px - mean[2 * q], qy = py - mean[2 * q + 1];
        for (int pass = 0; pass < 2; ++pass) {
            int i = (int)std::nearbyint((qx - xs[0]) / (xs[nx - 1] - xs[0]) * (nx - 1));
            int j = (int)std::nearbyint((qy - ys[0]) / (ys[ny - 1] - ys[0]) * (ny - 1));
            i = std::min(std::max(i, 0), nx - 1); j = std::min(std::max(j, 0), ny - 1);
            qx = px - (v[2 * (j * nx + i)] - xs[i]); qy = py - (v[2 * (j * nx + i) + 1] - ys[j]);
        }
        int ci = (int)(std::upper_bound(xs, xs + nx, qx) - xs) - 1, cj = (int)(std::upper_bound(ys, ys + ny, qy) - ys) - 1;
        ci = std::min(std::max(ci, 0), nx - 2); cj = std::min(std::max(cj, 0), ny - 2);
        double best = -inf, bb[3] = {nan, nan, nan};
        int bt = -1;
        for (int dj = 0; dj < 3 && best < 0; ++dj)
            for (int di = 0; di < 3 && best < 0; ++di) {
                const int ii = ci + order[di], jj = cj + order[dj];
                if (ii < 0 || ii >= nx - 1 || jj < 0 || jj >= ny - 1) continue;
                for (int half = 0; half < 2 && best < 0; ++half) {
                    const int t = 2 * (jj * (nx - 1) + ii) + half;
                    int n3[3];
                    tri_nodes(t, nx, n3);
                    const double d0x = px - v[2 * n3[0]], d0y = py - v[2 * n3[0] + 1];
                    const double d1x = px - v[2 * n3[1]], d1y = py - v[2 * n3[1] + 1];
                    const double d2x = px - v[2 * n3[2]], d2y = py - v[2 * n3[2] + 1];
                    const double a0 = d1x * d2y - d1y * d2x, a1 = d2x * d0y - d2y * d0x, a2 = d0x * d1y - d0y * d1x;
                    const double tot = a0 + a1 + a2;
                    const double b0 = a0 / tot, b1 = a1 / tot, b2 = a2 / tot;
                    double score = std::min(b0, std::min(b1, b2));
                    if (std::isnan(score)) score = -inf;
                    if (score > best) { best = score; bt = t; bb[0] = b0; bb[1] = b1; bb[2] = b2; }
                }
            }
        if (best < -eps) { bt = -1; bb[0] = bb[1] = bb[2] = nan; }
        tid[k] = bt; B[3 * k] = bb[0]; B[3 * k + 1] = bb[1]; B[3 * k + 2] = bb[2];
    }
    };
    const int T = std::max(1, std::min(fb_host_threads(4), (int)(K / 4096)));
    if (T <= 1) work(0, K);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) pool.emplace_back(work, K * t / T, K * (t + 1) / T);
        for (auto& th : pool) th.join();
    }
    return FB_OK;
}


// ---- round stepper of the coarse-to-fine block matcher (matcher.py:567-716) ---------------------------------------------------
// The spacings are walked from the largest to the smallest; after a round that measured a largest displacement d the walk jumps
// to the smallest spacing that is still larger than `multiplier` x d (at most 1 + max_skip places ahead), dwells, or moves one
// place on; a first round whose displacement outruns even the largest spacing is repeated once at ceil(multiplier x d).  Blocks
// are zero padded unless the walk jumped over a spacing (the search range then is the block itself).  One state machine for
// every caller: the general-mesh loop of feabas_amd/matcher.py steps it through the C ABI.
struct fb_schedule {
    std::vector<double> sp;          // descending
    int idx = 0;                     // place of the round that is due (-1: the enlarged extra round)
    double cur = 0.0;                // its spacing
    bool enlarged = false;           // the extra round has been spent (or is not allowed)
    int dwelled = 0, allow_dwell = 0, max_skip = 0;
    int pad_fixed = -1;              // -1: by rule
    bool pad = true;
};

fb_schedule* fb_schedule_create(const double* spacings, int n, int allow_enlarge, int allow_dwell, int max_spacing_skip, int pad_fixed) {
    if (!spacings || n <= 0) return nullptr;
    fb_schedule* s = new fb_schedule();
    s->sp.assign(spacings, spacings + n);
    std::sort(s->sp.begin(), s->sp.end(), [](double a, double b) { return a > b; });
    s->idx = 0; s->cur = s->sp[0];
    s->enlarged = !allow_enlarge;
    s->allow_dwell = std::max(0, allow_dwell); s->max_skip = std::max(0, max_spacing_skip);
    s->pad_fixed = pad_fixed < 0 ? -1 : (pad_fixed ? 1 : 0);
    s->pad = s->pad_fixed < 0 ? true : s->pad_fixed != 0;
    return s;
}

void fb_schedule_destroy(fb_schedule* s) { delete s; }

int fb_schedule_round(const fb_schedule* s, double* spacing, int* last, int* pad) {
    if (!s || s->idx >= (int)s->sp.size()) return 0;
    if (spacing) *spacing = s->cur;
    if (last) *last = s->cur == s->sp.back();
    if (pad) *pad = s->pad ? 1 : 0;
    return 1;
}

int fb_schedule_advance(fb_schedule* s, double max_dis, double multiplier, int* redo) {
    if (!s || !redo) return FB_ERR_ARG;
    *redo = 0;
    const int n = (int)s->sp.size();
    const double reach = multiplier * max_dis;                 // the smallest block that still holds the displacement
    int target = -1;                                           // place of the smallest spacing above it
    for (int k = 0; k < n; ++k) target += s->sp[k] > reach;
    auto set_pad = [&](bool v) { if (s->pad_fixed < 0) s->pad = v; };
    if (!s->enlarged && target < 0) {
        s->enlarged = true;
        s->idx = -1;
        s->cur = std::ceil(reach);
        set_pad(true);
        *redo = 1;
        return FB_OK;
    }
    s->enlarged = true;
    if (target > s->idx) {
        target = std::min(target, s->idx + 1 + s->max_skip);
        set_pad(target > s->idx + 1);
        s->idx = target; s->dwelled = 0;
    } else if (s->dwelled >= s->allow_dwell) {
        set_pad(true);
        s->idx += 1; s->dwelled = 0;
    } else {
        set_pad(true);
        s->dwelled += 1;
    }
    if (s->idx >= 0 && s->idx < n) s->cur = s->sp[s->idx];
    return FB_OK;
}

}  // extern "C"
