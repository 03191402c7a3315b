// matcher.stitching_matcher (feabas/matcher.py:224-367) for a BATCH of P equal-shaped overlap-strip pairs resident in
// HBM, as ONE C entry: x0.5 downsample + DoG + global NCC (255-278), fine DoG (336-337), the coarse-to-fine block rounds
// of iterative_xcorr_matcher_w_mesh (578-745: block grids of distributor_cartesian_bbox 865-891, pad / subpixel
// schedule 579-603 and 689-716, block -> point pairs 840-849, rigid relaxations between rounds), the last-round relaxation
// with its residue weights (725-737) and the strain estimate (752-777).  Only host bookkeeping lives here -- every
// device stage is one of the library's own entry points -- but it is O(pairs x blocks) work per round that serialised
// on the interpreter lock of the calling threads when it was written in numpy (feabas_amd/stitch_pipeline.py keeps that
// statement of it for masked / photometric batches and as the route of the pairs this entry hands back).
// A pair whose relaxation between two spacings is not a rigid translation keeps the node field of its mesh1 (725-742) and
// takes the deformed-mesh branch in the rounds that follow: block grid on the deformed bounding box (877), image-1 windows
// through MeshRenderer.crop_multiple's tiers (affine gather inside the NCC loader, exact piecewise-linear field through
// fb_remap_dev), matches located in the deformed triangles (Link.from_coordinates, optimizer.py:51-82).
//
// Pairs this entry does NOT finish are reported in flags[] and left to the caller's general route:
//   FB_STRIP_FOLDED     a block of a deformed mesh1 has a degenerate / flipped affine fit (renderer.py:397-416)
//   FB_STRIP_RELAXFIRST the last relaxation deformed mesh1 beyond the screen of relax_first (optimizer.py:763-779)
//   FB_STRIP_RIGIDFIT   the rigid initialisation of the strain stage is rank deficient / reflected / < 3 matches
#include "fb_common.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>
#include <numeric>
#include <vector>

#pragma clang fp contract(off)

// matcher.py:279-314 (sigma > 0): sums over the overlap of the two translated coarse strips -- grey levels of the raw images,
// |DoG| of the filtered ones -- where both masks are valid, and the number of valid pixels of strip 0 there.  One workgroup
// per (pair, band of rows); part [n][bands][6] = {sum raw0, sum raw1, sum |dog0|, sum |dog1|, count(m0 & m1), count(m0)}
__global__ __launch_bounds__(256) void photometric_kernel(int n, int hc, int wc, const uint8_t* __restrict__ raw, const float* __restrict__ dog,
                                                          const uint8_t* __restrict__ masks, const uint8_t* __restrict__ has_mask,
                                                          const int* __restrict__ txy, int bands, double* __restrict__ part) {
    const int p = blockIdx.x, band = blockIdx.y, tid = threadIdx.x;
    const int tx = txy[2 * p], ty = txy[2 * p + 1];
    const int xa = max(tx, 0), ya = max(ty, 0), xb = min(wc + tx, wc), yb = min(hc + ty, hc);
    const size_t pix = (size_t)hc * wc;
    const uint8_t* r0 = raw + (size_t)p * pix;
    const uint8_t* r1 = raw + (size_t)(n + p) * pix;
    const float* g0 = dog + (size_t)p * pix;
    const float* g1 = dog + (size_t)(n + p) * pix;
    const uint8_t* m0 = has_mask[p] ? masks + (size_t)p * pix : nullptr;
    const uint8_t* m1 = has_mask[n + p] ? masks + (size_t)(n + p) * pix : nullptr;
    const int rows = max(yb - ya, 0), cols = max(xb - xa, 0);
    const int per = (rows + bands - 1) / bands, ra = band * per, rb = min(rows, ra + per);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int r = ra; r < rb; ++r) {
        const int y1 = ya + r, y0 = y1 - ty;
        for (int c = tid; c < cols; c += 256) {
            const int x1 = xa + c, x0 = x1 - tx;
            const size_t i0 = (size_t)y0 * wc + x0, i1 = (size_t)y1 * wc + x1;
            const bool v0 = !m0 || m0[i0], v1 = !m1 || m1[i1];
            if (v0) acc[5] += 1.0;
            if (v0 && v1) {
                acc[0] += (double)r0[i0]; acc[1] += (double)r1[i1];
                acc[2] += (double)fabsf(g0[i0]); acc[3] += (double)fabsf(g1[i1]);
                acc[4] += 1.0;
            }
        }
    }
    __shared__ double red[256];
    for (int k = 0; k < 6; ++k) {
        red[tid] = acc[k];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) part[((size_t)p * bands + band) * 6 + k] = red[0];
        __syncthreads();
    }
}

// value range of a window of a float image: out[2 b] = min, out[2 b + 1] = max of block b = {image, x0, y0, h, w} (the
// `np.ptp(block) == 0` test of global_translation_matcher's second shot, matcher.py:196-205)
__global__ __launch_bounds__(256) void block_range_kernel(const float* __restrict__ imgs, int IH, int IW, const int* __restrict__ blk, float* __restrict__ out) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int* d = blk + 5 * b;
    const float* img = imgs + (size_t)d[0] * IH * IW;
    const int x0 = d[1], y0 = d[2], h = d[3], w = d[4];
    float lo = INFINITY, hi = -INFINITY;
    for (int i = tid; i < h * w; i += 256) {
        const int r = i / w, c = i - r * w;
        const float v = img[(size_t)(y0 + r) * IW + x0 + c];
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    __shared__ float slo[256], shi[256];
    slo[tid] = lo; shi[tid] = hi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { slo[tid] = fminf(slo[tid], slo[tid + s]); shi[tid] = fmaxf(shi[tid], shi[tid + s]); }
        __syncthreads();
    }
    if (tid == 0) { out[2 * b] = slo[0]; out[2 * b + 1] = shi[0]; }
}

struct fb_strip_matcher {
    fb_ctx* owner = nullptr;           // the context that made the matcher: its buffers and its system go back THERE
    int P = 0, H = 0, W = 0, hc = 0, wc = 0;
    double sigma = 2.5;
    int cds2 = 1;
    double conf_thresh = 0.33;
    int mnb = 2, conf_mode = 2;
    double residue_len = 5.0;
    int residue_mode = 0;
    double stiffness_lambda = 1.0, relax_tol = 1e-9;
    int compute_strain = 1;
    std::vector<double> sp;            // spacings in pixels, descending (matcher.py:567): [P][nsp], one row per pair
    int nsp = 0;
    bool ragged = false;               // strips of unequal size in slots of H x W (stitcher.py:561-571)
    std::vector<int> Hs, Ws, hcs, wcs; // per-pair strip extents, full resolution and coarse
    int* d_sizes = nullptr;            // device [2P][2] = {h, w} of every image of the two stacks
    int* d_sizes_c = nullptr;
    size_t b_dogc = 0, b_dogf = 0, b_small = 0, b_blk = 0, b_out = 0, b_sizes = 0;      // byte sizes of the pooled buffers
    float* d_dogc = nullptr;
    float* d_dogf = nullptr;
    uint8_t* d_small = nullptr;
    int* d_blk = nullptr;
    uint8_t* d_out = nullptr;
    size_t max_blocks = 0;
    fb_system* sys = nullptr;
    int gnx = 0, gny = 0;
    std::vector<double> gxs, gys;      // node coordinates of every pair's grid: [P][gnx], [P][gny]
    std::vector<double> es0, se;       // per pair: v0^T K v0 of the centred mesh, sample error of its matches (optimizer.py:26-30)
    double se0 = 0.0;                  // sample error on the template mesh (pair 0)
    std::vector<int32_t> r_pid;
    std::vector<double> r_xy0, r_xy1;
    std::vector<float> r_w;
    int relax_iters = 0, strain_iters = 0;
    double relax_relres = 0.0, strain_relres = 0.0;
    int64_t relax_matches = 0, strain_matches = 0;
    std::vector<uint8_t> raw;          // D2H staging of one launch: [dx f64 N][dy f64 N][conf f32 N]
    // deformed-mesh branch: node field of every pair's mesh1 (MOVING - INITIAL, [P][V][2]; zero for a translated grid), tiers
    // of the blocks of a pair's last deformed round, grow-only device scratch (affine maps; the exact-field tier's stacks)
    std::vector<double> U;
    std::vector<uint8_t> is_def;
    std::vector<std::vector<int32_t>> tiers;
    void* scr[14] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t b_scr[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // extras of the NEXT fb_match_strips call (fb_strip_matcher_set_extras): valid-pixel masks of the strips (host pointers,
    // NULL = none; matcher.py:257-274, 336-337) and the photometric statistics of matcher.py:279-314
    std::vector<const uint8_t*> mask0, mask1;
    bool want_phtm = false;
    std::vector<double> phtm;          // [P][4] of the last call that asked for them
    std::vector<uint8_t> phtm_has;     // [P]: 0 = fewer than 4 valid pixels of strip 0 in the overlap (None in the reference)
    // FEABAS_HIP_MATCH_TRACE=1: wall time of the stages of fb_match_strips, printed by fb_strip_matcher_destroy
    bool trace = false, trace_all = false;             // trace_all (=2): every call counts (matchers that live for one call)
    double t_stage[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // enqueue global, wait global, blocks host, blocks h2d+launch, blocks wait, relax, table+fits, strain
    double t_relax[3] = {0, 0, 0};     // inside the relax stage: locating the rows, fb_pairs_relax_bary, the screen
    int calls = 0;
};

namespace {

const double kDefaultAvgDeform = 0.05;          // feabas/config.py:32

struct StageClock {
    fb_strip_matcher* m;
    std::chrono::steady_clock::time_point t;
    explicit StageClock(fb_strip_matcher* mm) : m(mm), t(std::chrono::steady_clock::now()) {}
    void lap(int stage) {
        if (!m->trace) return;
        const auto now = std::chrono::steady_clock::now();
        m->t_stage[stage] += std::chrono::duration<double, std::milli>(now - t).count();
        t = now;
    }
};

// numpy.linspace(start, stop, num, endpoint=True)[i]
inline double linspace_at(double start, double stop, int num, int i) {
    if (num == 1) return start;
    if (i == num - 1) return stop;
    const double delta = stop - start;
    const double step = delta / (double)(num - 1);
    if (step == 0.0) return ((double)i / (double)(num - 1)) * delta + start;
    return (double)i * step + start;
}

inline int round_i(double v) { return (int)std::nearbyint(v); }      // np.round: half to even

// one axis of common.divide_bbox (feabas/common.py:380-409): the interval [lo, hi) cut into max(ceil(len / block), min_blocks)
// blocks of ceil(len / count) pixels whose starts are numpy.linspace(lo, hi - step, count); a shrink factor keeps the centres
// and scales the blocks
struct AxisCut { int count; long long step0, step; double shift; };      // step0: before the shrink factor (the linspace ends at hi - step0)
inline AxisCut cut_axis(double lo, double hi, double block, double min_blocks, double shrink) {
    const double len = hi - lo;
    const double cnt = std::max(std::ceil(len / block), min_blocks);
    AxisCut c;
    c.count = (int)cnt;
    c.step0 = c.step = (long long)std::ceil(len / cnt);
    c.shift = 0.0;
    if (shrink != 1.0) {
        const double s = (double)c.step0 * shrink;
        c.shift = ((double)c.step0 - s) / 2.0;
        c.step = (long long)std::ceil(s);
    }
    return c;
}
inline double cut_start(double lo, double hi, const AxisCut& c, int i) { return linspace_at(lo, hi - (double)c.step0, c.count, i) + c.shift; }

// the matcher's grid: node counts of Mesh.from_bbox((0, 0, W, H), cartesian=True) (mesh.py:403-435)
void grid_counts(int H, int W, double mesh_size, int mnb, int* nx_out, int* ny_out) {
    const double wd = (double)W, ht = (double)H;
    const double nx0 = std::max(std::nearbyint(wd / mesh_size), (double)mnb), ny0 = std::max(std::nearbyint(ht / mesh_size), (double)mnb);
    double dx = wd / nx0, dy = ht / ny0;
    if (dx > 2.0 * dy) dx = 2.0 * dy;
    else if (dy > 2.0 * dx) dy = 2.0 * dx;
    *nx_out = (int)std::ceil(wd / dx) + 1;
    *ny_out = (int)std::ceil(ht / dy) + 1;
}

// device buffers handed from matcher to matcher through the context (fb_common.h: match_pool)
int pool_take(fb_ctx* ctx, size_t bytes, void** out, size_t* got) {
    int best = -1;
    for (int k = 0; k < (int)ctx->match_pool.size(); ++k) {
        const size_t nb = ctx->match_pool[k].second;
        if (nb >= bytes && nb <= 4 * std::max<size_t>(bytes, 4096) && (best < 0 || nb < ctx->match_pool[best].second)) best = k;
    }
    if (best >= 0) {
        *out = ctx->match_pool[best].first; *got = ctx->match_pool[best].second;
        ctx->match_pool.erase(ctx->match_pool.begin() + best);
        return FB_OK;
    }
    *got = bytes;
    return fb_malloc(ctx, bytes, out);
}
void pool_give(fb_ctx* ctx, void* ptr, size_t bytes) {
    if (!ptr) return;
    ctx->match_pool.emplace_back(ptr, bytes);
    // bound the pool: at most 32 buffers (the smallest goes first) and 16 GB (the largest goes first)
    size_t total = 0;
    for (auto& b : ctx->match_pool) total += b.second;
    while (ctx->match_pool.size() > 32 || (total > ((size_t)16 << 30) && ctx->match_pool.size() > 1)) {
        const bool by_count = ctx->match_pool.size() > 32;
        size_t k = 0;
        for (size_t i = 1; i < ctx->match_pool.size(); ++i)
            if (by_count ? ctx->match_pool[i].second < ctx->match_pool[k].second : ctx->match_pool[i].second > ctx->match_pool[k].second) k = i;
        total -= ctx->match_pool[k].second;
        if (fb_free(ctx, ctx->match_pool[k].first) != FB_OK)
            std::fprintf(stderr, "libfeabas_hip: strip-matcher pool: %s\n", fb_last_error(ctx));     // a pointer of another context: never after the owner rule of fb_strip_matcher_destroy
        ctx->match_pool.erase(ctx->match_pool.begin() + k);
    }
}

// P copies of the grid topology as one block-diagonal system, every copy with the node coordinates of its pair's strip
// (mesh0 is locked, matcher.py:361: the stiffness of a matcher never changes).  The symbolic phase is kept per
// (P, nx, ny) in the context, so a matcher of a new batch of shapes only re-assembles the values.
int ensure_system(fb_ctx* ctx, fb_strip_matcher* m) {
    if (m->sys) return FB_OK;
    const int P = m->P, nsp = m->nsp;
    int nx = 0, ny = 0;
    grid_counts(m->Hs[0], m->Ws[0], *std::min_element(m->sp.begin(), m->sp.begin() + nsp), m->mnb, &nx, &ny);
    for (int p = 1; p < P; ++p) {
        int a = 0, b = 0;
        grid_counts(m->Hs[p], m->Ws[p], *std::min_element(m->sp.begin() + (size_t)p * nsp, m->sp.begin() + (size_t)(p + 1) * nsp), m->mnb, &a, &b);
        if (a != nx || b != ny)
            return fb_fail(ctx, FB_ERR_ARG, "fb_match_strips: pair %d has a %d x %d node grid, pair 0 %d x %d (the pairs of a batch share the mesh topology)", p, a, b, nx, ny);
    }
    const int V = nx * ny, T = 2 * (nx - 1) * (ny - 1);
    m->gnx = nx; m->gny = ny;
    m->gxs.resize((size_t)P * nx); m->gys.resize((size_t)P * ny);
    for (int p = 0; p < P; ++p) {                            // mesh.py:430-431 per pair
        for (int i = 0; i < nx; ++i) m->gxs[(size_t)p * nx + i] = linspace_at(0.0, (double)m->Ws[p], nx, i) - 0.5;
        for (int j = 0; j < ny; ++j) m->gys[(size_t)p * ny + j] = linspace_at(0.0, (double)m->Hs[p], ny, j) - 0.5;
    }
    int rc;
    fb_system* s = nullptr;
    int mid = 0;
    auto key = std::make_tuple(P, nx, ny);
    auto it = ctx->match_systems.find(key);
    if (it != ctx->match_systems.end() && !it->second.empty()) {
        s = it->second.back();
        it->second.pop_back();
    } else {
        std::vector<int32_t> tri((size_t)3 * P * T);
        for (int p = 0; p < P; ++p) {
            int32_t* t = &tri[(size_t)3 * p * T];
            for (int j = 0; j < ny - 1; ++j)
                for (int i = 0; i < nx - 1; ++i) {
                    const int a = p * V + j * nx + i, b = a + 1, c = a + nx, d = c + 1;
                    t[0] = a; t[1] = b; t[2] = d; t[3] = a; t[4] = d; t[5] = c;
                    t += 6;
                }
        }
        int64_t nnzb = 0;
        if ((rc = fb_sys_create(ctx, (int64_t)P * V, &s))) return rc;
        if ((rc = fb_sys_add_mesh(ctx, s, 0, tri.data(), P * V, P * T, &mid)) || (rc = fb_sys_set_links(ctx, s, 0, nullptr)) ||
            (rc = fb_sys_finalize(ctx, s, &nnzb))) {
            fb_sys_destroy(ctx, s);
            return rc;
        }
    }
    std::vector<float> mult((size_t)P * T, 1.0f);
    std::vector<double> v((size_t)2 * P * V), v0((size_t)2 * P * V);
    for (int p = 0; p < P; ++p) {
        const double* gx = &m->gxs[(size_t)p * nx];
        const double* gy = &m->gys[(size_t)p * ny];
        double mx = 0.0, my = 0.0;
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) { mx += gx[i]; my += gy[j]; }
        mx /= V; my /= V;
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const size_t k = 2 * ((size_t)p * V + (size_t)j * nx + i);
                v[k] = gx[i]; v[k + 1] = gy[j];
                v0[k] = gx[i] - mx; v0[k + 1] = gy[j] - my;
            }
    }
    m->es0.resize((size_t)P);
    rc = fb_sys_assemble_mesh(ctx, s, 0, v.data(), nullptr, mult.data(), 0.0, 1.0);
    if (rc || (rc = fb_sys_group_energy(ctx, s, P, v0.data(), m->es0.data()))) {
        fb_sys_destroy(ctx, s);
        return rc;
    }
    m->sys = s;
    // sample error of a match, 0.4387 sqrt(A) strain (optimizer.py:26-30) with A = Mesh.triangle_areas = the cross product
    // of two edges (common.py:672-676): (dx, 0) x (0, dy) for the triangles of a grid cell
    m->se.resize((size_t)P);
    for (int p = 0; p < P; ++p) {
        const double area = std::fabs((m->gxs[(size_t)p * nx + 1] - m->gxs[(size_t)p * nx]) * (m->gys[(size_t)p * ny + 1] - m->gys[(size_t)p * ny]) - 0.0 * 0.0);
        m->se[p] = 0.4387 * std::pow(area, 0.5) * kDefaultAvgDeform;
    }
    m->se0 = m->se[0];
    return FB_OK;
}

// the results of several launches issued one behind the other: launch g wrote (dx [nb_g], dy [nb_g], conf [nb_g]) at byte
// 24 * (blocks of the launches before it) of d_out; one copy brings all of them
int fetch_all(fb_ctx* ctx, fb_strip_matcher* m, size_t total_blocks) {
    m->raw.resize(24 * total_blocks);
    return total_blocks ? fb_memcpy_d2h(ctx, m->raw.data(), m->d_out, 24 * total_blocks) : FB_OK;
}

int fetch(fb_ctx* ctx, fb_strip_matcher* m, size_t nb, const double** dx, const double** dy, const float** cf) {
    m->raw.resize(20 * nb);
    int rc = fb_memcpy_d2h(ctx, m->raw.data(), m->d_out, 20 * nb);
    if (rc) return rc;
    *dx = (const double*)m->raw.data();
    *dy = (const double*)(m->raw.data() + 8 * nb);
    *cf = (const float*)(m->raw.data() + 16 * nb);
    return FB_OK;
}

// triangle (three vertex ids inside the union mesh, in the order of the triangle list) and barycentric coordinates of
// points given in the INITIAL gear of their pair's grid mesh: Mesh.cart2bary (mesh.py:2191-2217) on a uniform grid, the
// cell from one division per axis (the statement feabas_amd/stitch_pipeline.py::_locate_grid makes for ragged batches)
void locate_grid(const fb_strip_matcher* m, size_t K, const int32_t* pid, const double* pts, std::vector<int32_t>& nodes3, std::vector<double>& B) {
    const int nx = m->gnx, ny = m->gny, V = nx * ny;
    nodes3.resize(3 * K); B.resize(3 * K);
    for (size_t k = 0; k < K; ++k) {
        const int p = pid[k];
        const double* gx = &m->gxs[(size_t)p * nx];
        const double* gy = &m->gys[(size_t)p * ny];
        const double cw = (gx[nx - 1] - gx[0]) / (double)(nx - 1), ch = (gy[ny - 1] - gy[0]) / (double)(ny - 1);
        const double fx = (pts[2 * k] - gx[0]) / cw, fy = (pts[2 * k + 1] - gy[0]) / ch;
        const double i = std::min(std::max(std::floor(fx), 0.0), (double)(nx - 2)), j = std::min(std::max(std::floor(fy), 0.0), (double)(ny - 2));
        const double u = fx - i, w = fy - j;
        const bool up = w > u;
        const int na = (int)j * nx + (int)i + p * V;
        nodes3[3 * k] = na; nodes3[3 * k + 1] = up ? na + nx + 1 : na + 1; nodes3[3 * k + 2] = up ? na + nx : na + nx + 1;
        B[3 * k] = up ? 1.0 - w : 1.0 - u; B[3 * k + 1] = up ? u : u - w; B[3 * k + 2] = up ? w - u : w;
    }
}

// Mesh.locate_cartesian + Mesh.cart2bary (feabas_amd/mesh.py; reference mesh.py:2191-2217) of a point given in the INITIAL
// gear of pair p's grid mesh: the cell by bisection of the node coordinates, the triangle of the cell by its diagonal, the
// barycentric coordinates as ratios of cross products.  The statement of uniform batches (`locate_grid` is the one of
// ragged ones: the two agree to rounding, and each is what its host route computes)
void locate_cart(const fb_strip_matcher* m, int p, double x, double y, int32_t n3[3], double B[3]) {
    const int nx = m->gnx, ny = m->gny, V = nx * ny;
    const double* xs = &m->gxs[(size_t)p * nx];
    const double* ys = &m->gys[(size_t)p * ny];
    int i = (int)(std::upper_bound(xs, xs + nx, x) - xs) - 1, j = (int)(std::upper_bound(ys, ys + ny, y) - ys) - 1;
    i = std::min(std::max(i, 0), nx - 2); j = std::min(std::max(j, 0), ny - 2);
    const double u = (x - xs[i]) / (xs[i + 1] - xs[i]), w = (y - ys[j]) / (ys[j + 1] - ys[j]);
    const bool up = w > u;
    const int a = j * nx + i;
    const int loc[3] = {a, up ? a + nx + 1 : a + 1, up ? a + nx : a + nx + 1};
    double dx[3], dy[3];
    for (int k = 0; k < 3; ++k) { dx[k] = x - xs[loc[k] % nx]; dy[k] = y - ys[loc[k] / nx]; n3[k] = loc[k] + p * V; }
    const double a0 = dx[1] * dy[2] - dy[1] * dx[2], a1 = dx[2] * dy[0] - dy[2] * dx[0], a2 = dx[0] * dy[1] - dy[0] * dx[1];
    const double tot = (a0 + a1) + a2;
    B[0] = a0 / tot; B[1] = a1 / tot; B[2] = a2 / tot;
}

// grow-only device scratch of the matcher, by slot
enum : int {
    kScrAff = 0,                                                     // affine maps of a deformed group [nb][10] f64
    kScrIds = 1, kScrOrg = 2, kScrMapX = 3, kScrMapY = 4, kScrMapMask = 5, kScrStack = 6, kScrExactOut = 7,   // exact-field tier
    kScrMaskCoarse = 8, kScrMaskFine0 = 9, kScrMaskFine1 = 10, kScrRaw = 11,                                   // masks, statistics
    kScrDenseIn = 12, kScrDenseOut = 13,                             // a masked image of a ragged batch, copied out of its slot
    kScrTxy = kScrIds, kScrHas = kScrOrg, kScrPart = kScrMapX        // photometric statistics (before any deformed round runs)
};
int scratch(fb_ctx* ctx, fb_strip_matcher* m, int k, size_t bytes, void** out) {
    if (m->b_scr[k] < bytes) {
        if (m->scr[k]) pool_give(ctx, m->scr[k], m->b_scr[k]);
        m->scr[k] = nullptr; m->b_scr[k] = 0;
        void* ptr = nullptr;
        const int rc = pool_take(ctx, bytes + bytes / 4, &ptr, &m->b_scr[k]);
        if (rc) return rc;
        m->scr[k] = ptr;
    }
    *out = m->scr[k];
    return FB_OK;
}

// screen of relax_first (adjust_link_weight_by_residue(relax_first=True), matcher.py:736 -> optimizer.py:763-779) on the node
// field x [P][V][2]: d = the largest displacement difference along a grid edge relative to that edge.  Every triangle's area
// and edge deformation is below 2 d and nothing is freed below 1 - 1 / (1 + (1 - 1 / 1.35)) = 0.206: pairs with d <= 0.1 are
// done, the others take the reference's statements on the caller's general route.
void screen_relax_first(const fb_strip_matcher* m, const std::vector<double>& x, uint8_t* flags) {
    const int n = m->P, nx = m->gnx, ny = m->gny, V = nx * ny;
    for (int p = 0; p < n; ++p) {
        const double* g = &x[(size_t)2 * p * V];
        const double* gx = &m->gxs[(size_t)p * nx];
        const double* gy = &m->gys[(size_t)p * ny];
        double d = 0.0;
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                const double* a = g + 2 * ((size_t)j * nx + i);
                if (i + 1 < nx) d = std::max(d, std::sqrt((a[2] - a[0]) * (a[2] - a[0]) + (a[3] - a[1]) * (a[3] - a[1])) / (gx[i + 1] - gx[i]));
                if (j + 1 < ny) {
                    const double* b = a + 2 * (size_t)nx;
                    d = std::max(d, std::sqrt((b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1])) / (gy[j + 1] - gy[j]));
                }
            }
        if (d > 0.1 || !(d == d)) flags[p] |= FB_STRIP_RELAXFIRST;
    }
}

struct Rows {                                   // a match table: rows of one pair are contiguous
    std::vector<int32_t> pid;
    std::vector<double> xy0, xy1i, xy1;         // mesh0 point (MOVING), mesh1 point (INITIAL), mesh1 point (MOVING)
    std::vector<float> wt;
    std::vector<char> rl;                       // the pair's blocks moved by more than 0.1 px (matcher.py:725)
    std::vector<int32_t> n3;                    // mesh1 triangle of a match located in a DEFORMED mesh1 (ids in the union mesh; -1: grid)
    std::vector<double> B;                      // ... and its barycentric coordinates
    size_t size() const { return pid.size(); }
    void reserve(size_t k) { pid.reserve(k); xy0.reserve(2 * k); xy1i.reserve(2 * k); xy1.reserve(2 * k); wt.reserve(k); rl.reserve(k); n3.reserve(3 * k); B.reserve(3 * k); }
    void push(int32_t p, double x0, double y0, double xi, double yi, double x1, double y1, float w, char r, const int32_t* nodes = nullptr,
              const double* bary = nullptr) {
        pid.push_back(p); xy0.push_back(x0); xy0.push_back(y0); xy1i.push_back(xi); xy1i.push_back(yi);
        xy1.push_back(x1); xy1.push_back(y1); wt.push_back(w); rl.push_back(r);
        for (int a = 0; a < 3; ++a) { n3.push_back(nodes ? nodes[a] : -1); B.push_back(bary ? bary[a] : 0.0); }
    }
    void push_from(const Rows& o, size_t k) {
        push(o.pid[k], o.xy0[2 * k], o.xy0[2 * k + 1], o.xy1i[2 * k], o.xy1i[2 * k + 1], o.xy1[2 * k], o.xy1[2 * k + 1], o.wt[k], o.rl[k], &o.n3[3 * k],
             &o.B[3 * k]);
    }
};

// eigenvalues of a symmetric 3x3 matrix (cyclic Jacobi), ascending
void sym3_eig(const double G[3][3], double ev[3]) {
    double a[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = G[i][j];
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double th = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 3; ++k) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
            }
    }
    ev[0] = a[0][0]; ev[1] = a[1][1]; ev[2] = a[2][2];
    std::sort(ev, ev + 3);
}

// A = G^-1 Hm by elimination with partial pivoting; false when a pivot vanishes
bool solve3(const double G[3][3], const double Hm[3][3], double A[3][3]) {
    double a[3][6];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { a[i][j] = G[i][j]; a[i][3 + j] = Hm[i][j]; }
    for (int c = 0; c < 3; ++c) {
        int piv = c;
        for (int r = c + 1; r < 3; ++r) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        if (piv != c) for (int j = 0; j < 6; ++j) std::swap(a[c][j], a[piv][j]);
        for (int r = c + 1; r < 3; ++r) {
            const double f = a[r][c] / a[c][c];
            for (int j = c; j < 6; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int j = 0; j < 3; ++j)
        for (int r = 2; r >= 0; --r) {
            double v = a[r][3 + j];
            for (int k = r + 1; k < 3; ++k) v -= a[r][k] * A[k][j];
            A[r][j] = v / a[r][r];
        }
    return true;
}

// spatial.fit_affine(p0, p1, return_rigid=True, weight, svd_clip=(1, 1)) (spatial.py:21-73) of every pair through the
// moments of its rows (the statement feabas_amd/stitch_pipeline.py::_rigid_fits makes in numpy).  R [P][9] row major;
// bad[p] = 1 where the fit needs the rank-deficient / reflected branches of the host function.
void rigid_fits(int P, const std::vector<int32_t>& pid, const std::vector<double>& p0, const std::vector<double>& p1, const std::vector<float>& wt,
                std::vector<double>& R, std::vector<char>& bad) {
    R.assign((size_t)9 * P, 0.0);
    bad.assign((size_t)P, 0);
    for (int p = 0; p < P; ++p) { R[9 * (size_t)p] = R[9 * (size_t)p + 4] = R[9 * (size_t)p + 8] = 1.0; }
    const size_t K = pid.size();
    size_t a = 0;
    while (a < K) {
        size_t b = a;
        const int p = pid[a];
        double S[21];
        for (int j = 0; j < 21; ++j) S[j] = 0.0;
        for (; b < K && pid[b] == p; ++b) {
            const double x0 = p0[2 * b], y0 = p0[2 * b + 1], x1 = p1[2 * b], y1 = p1[2 * b + 1], w = (double)wt[b];
            const double wx1 = w * x1, wy1 = w * y1;
            const double F[21] = {1.0, x0, y0, x1, y1, x0 * x0, y0 * y0, x1 * x1, y1 * y1, w, wx1, wy1, w * x0, w * y0,
                                  wx1 * x1, wx1 * y1, wy1 * y1, wx1 * x0, wx1 * y0, wy1 * x0, wy1 * y0};
            for (int j = 0; j < 21; ++j) S[j] += F[j];
        }
        a = b;
        const double cnt = S[0], n = std::max(cnt, 1.0);
        const double m0x = S[1] / n, m0y = S[2] / n, m1x = S[3] / n, m1y = S[4] / n;
        const double var0 = (S[5] / n - m0x * m0x) + (S[6] / n - m0y * m0y);
        const double var1 = (S[7] / n - m1x * m1x) + (S[8] / n - m1y * m1y);
        double scl = std::sqrt(std::max(std::max(var0, var1), 0.0));
        if (scl < 1e-6) scl = 1.0;
        const double sw = S[9], sx1 = S[10], sy1 = S[11], sx0 = S[12], sy0 = S[13];
        const double cx1 = sx1 - m1x * sw, cy1 = sy1 - m1y * sw, cx0 = sx0 - m0x * sw, cy0 = sy0 - m0y * sw;
        const double s2 = scl * scl;
        double G[3][3], Hm[3][3], A[3][3];
        G[0][0] = (S[14] - 2 * m1x * sx1 + m1x * m1x * sw) / s2;
        G[0][1] = G[1][0] = (S[15] - m1x * sy1 - m1y * sx1 + m1x * m1y * sw) / s2;
        G[1][1] = (S[16] - 2 * m1y * sy1 + m1y * m1y * sw) / s2;
        G[0][2] = G[2][0] = cx1 / scl;
        G[1][2] = G[2][1] = cy1 / scl;
        G[2][2] = sw;
        Hm[0][0] = (S[17] - m1x * sx0 - m0x * sx1 + m1x * m0x * sw) / s2;
        Hm[0][1] = (S[18] - m1x * sy0 - m0y * sx1 + m1x * m0y * sw) / s2;
        Hm[1][0] = (S[19] - m1y * sx0 - m0x * sy1 + m1y * m0x * sw) / s2;
        Hm[1][1] = (S[20] - m1y * sy0 - m0y * sy1 + m1y * m0y * sw) / s2;
        Hm[0][2] = cx1 / scl; Hm[1][2] = cy1 / scl;
        Hm[2][0] = cx0 / scl; Hm[2][1] = cy0 / scl; Hm[2][2] = sw;
        bool ok = cnt >= 3;
        if (ok) {
            double ev[3];
            sym3_eig(G, ev);
            ok = ev[0] > 1e-9 * ev[2];
        }
        if (ok) ok = solve3(G, Hm, A);
        if (ok) {
            const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                               A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
            ok = det > 0.0 && (A[0][0] * A[1][1] - A[0][1] * A[1][0]) > 0.0;
        }
        double c = 1.0, s = 0.0;
        if (ok) {
            // u @ vh of the 2x2 block (singular values clipped to 1): the rotation of its polar decomposition
            const double pc = A[0][0] + A[1][1], ps = A[1][0] - A[0][1];
            const double hyp = std::hypot(pc, ps);
            ok = hyp > 0.0 && std::isfinite(hyp);
            if (ok) { c = pc / hyp; s = ps / hyp; }
        }
        if (!ok) { bad[p] = 1; continue; }
        double* r = &R[9 * (size_t)p];
        r[0] = c; r[1] = -s; r[3] = s; r[4] = c;
        r[6] = A[2][0] + m0x - (m1x * r[0] + m1y * r[3]);
        r[7] = A[2][1] + m0y - (m1x * r[1] + m1y * r[4]);
        r[2] = 0.0; r[5] = 0.0; r[8] = 1.0;
    }
}

// matcher.py:725-741 for matches given by their mesh1 triangle and barycentric coordinates: optimize_linear as the TOTAL
// displacement of mesh1 from its FIXED gear (fb_pairs_relax_bary), the screen of relax_first, huber / threshold residue
// weights and -- `resolve`, the rounds before the last -- a second solve for the pairs whose weights changed (737-741).
// rows: indices into `t`; rows whose triangle is not known (a mesh1 that is still a translated grid) are located on the grid.
// rw [rows]; x [P][V][2] (pairs without rows: zero).  Pairs beyond the screen get FB_STRIP_RELAXFIRST.
int relax_general(fb_ctx* ctx, fb_strip_matcher* m, const Rows& t, const std::vector<size_t>& rows, bool resolve, std::vector<float>& rw,
                  std::vector<double>& x, uint8_t* flags) {
    int rc;
    if ((rc = ensure_system(ctx, m))) return rc;
    const int n = m->P, nx = m->gnx, ny = m->gny, V = nx * ny;
    const int64_t K = (int64_t)rows.size();
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](int k) {
        const auto now = std::chrono::steady_clock::now();
        if (m->trace && (m->calls > 3 || m->trace_all)) m->t_relax[k] += std::chrono::duration<double, std::milli>(now - tick).count();
        tick = now;
    };
    std::vector<int32_t> pid((size_t)K), nodes3((size_t)3 * K);
    std::vector<double> B1((size_t)3 * K), dxy0((size_t)2 * K), se_rows, pts;
    std::vector<float> w32((size_t)K);
    for (int64_t k = 0; k < K; ++k) { pid[k] = t.pid[rows[k]]; w32[k] = t.wt[rows[k]]; }
    if (m->ragged) {
        // one mesh geometry per pair: every row is located on its pair's grid from its INITIAL coordinates
        pts.resize((size_t)2 * K); se_rows.resize((size_t)K);
        for (int64_t k = 0; k < K; ++k) { pts[2 * k] = t.xy1i[2 * rows[k]]; pts[2 * k + 1] = t.xy1i[2 * rows[k] + 1]; se_rows[k] = m->se[pid[k]]; }
        locate_grid(m, (size_t)K, pid.data(), pts.data(), nodes3, B1);
    } else {
        for (int64_t k = 0; k < K; ++k) {
            const size_t r = rows[k];
            if (t.n3[3 * r] < 0) locate_cart(m, pid[k], t.xy1i[2 * r], t.xy1i[2 * r + 1], &nodes3[3 * k], &B1[3 * k]);
            else for (int a = 0; a < 3; ++a) { nodes3[3 * k + a] = t.n3[3 * r + a]; B1[3 * k + a] = t.B[3 * r + a]; }
        }
    }
    for (int64_t k = 0; k < K; ++k) {
        const int p = pid[k];
        double fx[3], fy[3];
        for (int a = 0; a < 3; ++a) {
            const int loc = nodes3[3 * k + a] - p * V;
            fx[a] = m->gxs[(size_t)p * nx + loc % nx]; fy[a] = m->gys[(size_t)p * ny + loc / nx];
        }
        const double* b = &B1[3 * (size_t)k];
        dxy0[2 * k] = ((fx[0] * b[0] + fx[1] * b[1]) + fx[2] * b[2]) - t.xy0[2 * rows[k]];          // mesh1 at its FIXED gear (= INITIAL, no offset)
        dxy0[2 * k + 1] = ((fy[0] * b[0] + fy[1] * b[1]) + fy[2] * b[2]) - t.xy0[2 * rows[k] + 1];
    }
    rw.resize((size_t)K);
    x.resize((size_t)2 * n * V);
    lap(0);
    const double rlen = m->residue_len > 0 ? m->residue_len : 1.0;
    if ((rc = fb_pairs_relax_bary(ctx, m->sys, n, K, nodes3.data(), B1.data(), dxy0.data(), w32.data(), rlen, m->residue_mode, m->se0,
                                  m->ragged ? se_rows.data() : nullptr, m->stiffness_lambda, m->relax_tol, rw.data(), x.data(), &m->relax_iters, &m->relax_relres)))
        return rc;
    m->relax_matches = K;
    lap(1);
    if (m->residue_len <= 0) { std::fill(rw.begin(), rw.end(), 1.0f); return FB_OK; }
    screen_relax_first(m, x, flags);
    lap(2);
    bool changed_any = false;
    for (int64_t k = 0; k < K; ++k) changed_any |= rw[k] != 1.0f;
    if (resolve && changed_any) {
        std::vector<char> changed((size_t)n, 0);
        std::vector<float> w2((size_t)K), rw2((size_t)K);
        std::vector<double> x2((size_t)2 * n * V);
        for (int64_t k = 0; k < K; ++k) { if (rw[k] != 1.0f) changed[pid[k]] = 1; w2[k] = w32[k] * rw[k]; }
        if ((rc = fb_pairs_relax_bary(ctx, m->sys, n, K, nodes3.data(), B1.data(), dxy0.data(), w2.data(), rlen, m->residue_mode, m->se0,
                                      m->ragged ? se_rows.data() : nullptr, m->stiffness_lambda, m->relax_tol, rw2.data(), x2.data(), &m->relax_iters,
                                      &m->relax_relres)))
            return rc;
        for (int p = 0; p < n; ++p)
            if (changed[p]) std::copy(x2.begin() + (size_t)2 * p * V, x2.begin() + (size_t)2 * (p + 1) * V, x.begin() + (size_t)2 * p * V);
    }
    return FB_OK;
}

}  // namespace

namespace {

// matcher.py:243-251 with both shapes equal, descending
void auto_spacings(int H, int W, std::vector<double>& out) {
    out.clear();
    const double smax = std::max(H, W) * 0.25, smin = std::max(std::min(75.0, std::min(H, W) / 3.0), 25.0);
    if (smin > smax) out.assign(1, smin);
    else {
        const int count = (int)std::max(1.0, std::nearbyint(std::log(smax / smin) / std::log(4.0)));
        for (int i = 0; i < count; ++i) out.push_back(std::exp(linspace_at(std::log(smin), std::log(smax), count, i)));
    }
    std::sort(out.begin(), out.end());
    std::reverse(out.begin(), out.end());
}

int matcher_create(fb_ctx* ctx, int P, int H, int W, const int32_t* shapes, const fb_strip_opts* o, fb_strip_matcher** out) {
    FB_CHECK_ARG(ctx, P > 0 && H > 1 && W > 1 && o && out && (o->coarse_downsample2 == 0 || o->coarse_downsample2 == 1));
    FB_CHECK_ARG(ctx, o->sigma > 0.0 && o->min_num_blocks >= 1 && (o->residue_mode == 0 || o->residue_mode == 1) && o->nspacings >= 0 &&
                          (o->nspacings == 0 || o->spacings) && o->nspacings <= 64);
    fb_strip_matcher* m = new fb_strip_matcher();
    m->owner = ctx;
    m->P = P; m->H = H; m->W = W;
    m->ragged = shapes != nullptr;
    { const char* e = std::getenv("FEABAS_HIP_MATCH_TRACE"); m->trace = e && (e[0] == '1' || e[0] == '2'); m->trace_all = e && e[0] == '2'; }
    m->sigma = o->sigma; m->cds2 = o->coarse_downsample2; m->conf_thresh = o->conf_thresh; m->mnb = o->min_num_blocks;
    m->conf_mode = o->conf_mode; m->residue_len = o->residue_len; m->residue_mode = o->residue_mode;
    m->stiffness_lambda = o->stiffness_lambda; m->relax_tol = o->relax_tol; m->compute_strain = o->compute_strain;
    m->Hs.resize((size_t)P); m->Ws.resize((size_t)P); m->hcs.resize((size_t)P); m->wcs.resize((size_t)P);
    for (int p = 0; p < P; ++p) {
        m->Hs[p] = shapes ? shapes[2 * p] : H;
        m->Ws[p] = shapes ? shapes[2 * p + 1] : W;
        if (m->Hs[p] < 2 || m->Ws[p] < 2 || m->Hs[p] > H || m->Ws[p] > W) {
            const int hp = m->Hs[p], wp = m->Ws[p];
            delete m;
            return fb_fail(ctx, FB_ERR_ARG, "fb_strip_matcher_create: pair %d: a %d x %d strip does not fit its %d x %d slot", p, hp, wp, H, W);
        }
        // cv2.resize(fx=0.5): cvRound(n / 2), half to even
        m->hcs[p] = m->cds2 ? round_i(m->Hs[p] * 0.5) : m->Hs[p];
        m->wcs[p] = m->cds2 ? round_i(m->Ws[p] * 0.5) : m->Ws[p];
    }
    std::vector<double> one;
    if (o->nspacings) {
        one.assign(o->spacings, o->spacings + o->nspacings);
        for (double v : one)
            if (!(v >= 1.0)) { delete m; return fb_fail(ctx, FB_ERR_ARG, "fb_strip_matcher_create: spacings are pixels (>= 1); relative ones (matcher.py:343-350) are resolved by the caller"); }
        std::sort(one.begin(), one.end());
        std::reverse(one.begin(), one.end());
        m->nsp = (int)one.size();
        for (int p = 0; p < P; ++p) m->sp.insert(m->sp.end(), one.begin(), one.end());
    } else {
        for (int p = 0; p < P; ++p) {
            auto_spacings(m->Hs[p], m->Ws[p], one);
            if (p == 0) m->nsp = (int)one.size();
            if ((int)one.size() != m->nsp) {
                const int nsp0 = m->nsp;                       // (read before the matcher goes: the message used to read it afterwards)
                delete m;
                return fb_fail(ctx, FB_ERR_ARG, "fb_strip_matcher_create: pair %d has %d automatic spacings, pair 0 %d (the pairs of a batch share the number of rounds)", p, (int)one.size(), nsp0);
            }
            m->sp.insert(m->sp.end(), one.begin(), one.end());
        }
    }
    m->hc = m->cds2 ? round_i(H * 0.5) : H;
    m->wc = m->cds2 ? round_i(W * 0.5) : W;
    m->max_blocks = (size_t)P * 1024;
    const size_t n = (size_t)P, cpix = (size_t)m->hc * m->wc, fpix = (size_t)H * W;
    int rc = 0;
    void* ptr = nullptr;
    if (!(rc = pool_take(ctx, 2 * n * cpix * 4, &ptr, &m->b_dogc))) m->d_dogc = (float*)ptr;
    if (!rc && m->cds2 && !(rc = pool_take(ctx, 2 * n * fpix * 4, &ptr, &m->b_dogf))) m->d_dogf = (float*)ptr;
    if (!rc && m->cds2 && !(rc = pool_take(ctx, 2 * n * cpix, &ptr, &m->b_small))) m->d_small = (uint8_t*)ptr;
    if (!rc && !(rc = pool_take(ctx, m->max_blocks * 9 * 4, &ptr, &m->b_blk))) m->d_blk = (int*)ptr;
    if (!rc && !(rc = pool_take(ctx, m->max_blocks * 24, &ptr, &m->b_out))) m->d_out = (uint8_t*)ptr;
    if (!rc && m->ragged) {
        // per-image extents of the two stacks, full resolution and coarse (fb_dog_sizes_dev, fb_area_downsample2_sizes_dev)
        std::vector<int32_t> sz((size_t)8 * P);
        for (int side = 0; side < 2; ++side)
            for (int p = 0; p < P; ++p) {
                sz[2 * ((size_t)side * P + p)] = m->Hs[p]; sz[2 * ((size_t)side * P + p) + 1] = m->Ws[p];
                sz[4 * (size_t)P + 2 * ((size_t)side * P + p)] = m->hcs[p]; sz[4 * (size_t)P + 2 * ((size_t)side * P + p) + 1] = m->wcs[p];
            }
        if (!(rc = pool_take(ctx, sz.size() * 4, &ptr, &m->b_sizes))) {
            m->d_sizes = (int*)ptr;
            m->d_sizes_c = m->d_sizes + 4 * (size_t)P;
            rc = fb_memcpy_h2d(ctx, m->d_sizes, sz.data(), sz.size() * 4);
        }
    }
    if (rc) { fb_strip_matcher_destroy(ctx, m); return rc; }
    *out = m;
    return FB_OK;
}

}  // namespace

namespace {

// second shot of global_translation_matcher (matcher.py:159-221) for the pairs without a confident whole-strip peak: ~6
// sub-blocks on the grid of the most moderate aspect ratio, blocks without contrast dropped, the most confident block wins if
// it is at least as confident as the whole strip.  The two strips of a pair have one shape, so their block grids coincide and
// the re-centred padding of 184-210 is the block itself; pairs of unequal extent (ragged batches) have their own grids and
// are launched per FFT shape.
int second_shot(fb_ctx* ctx, fb_strip_matcher* m, float thr, double* tx, double* ty, float* conf0) {
    const int n = m->P, hc = m->hc, wc = m->wc;
    const size_t cpix = (size_t)hc * wc;
    struct Grid { int p, nx, ny, bw, bh; AxisCut cx, cy; };
    std::map<long long, std::vector<Grid>> by_shape;
    for (int p = 0; p < n; ++p) {
        if (conf0[p] > thr) continue;
        const int hp = m->hcs[p], wp = m->wcs[p];
        int gr = 1, gc = 1;
        {
            const int df = 6;
            const double aspect = (double)hp / (double)wp;
            double best = std::numeric_limits<double>::infinity();
            for (int f = 1; f <= (int)std::sqrt((double)df); ++f) {
                if (df % f) continue;
                const double qf = (double)(f * f) / df;
                const double v1 = std::fabs(std::log(aspect * qf)), v2 = std::fabs(std::log(aspect / qf));
                if (v1 < best) { best = v1; gr = df / f; gc = f; }
                if (v2 < best) { best = v2; gr = f; gc = df / f; }
            }
        }
        const double big = (double)std::max(hp, wp);       // common.divide_bbox: block_size defaults to the larger extent
        Grid g;
        g.p = p;
        g.cx = cut_axis(0.0, (double)wp, big, (double)gc, 1.0); g.cy = cut_axis(0.0, (double)hp, big, (double)gr, 1.0);
        g.nx = g.cx.count; g.ny = g.cy.count; g.bw = (int)g.cx.step; g.bh = (int)g.cy.step;
        if (g.bw < 1 || g.bh < 1) continue;
        by_shape[(long long)fb_next_fast_len(2 * g.bh - 1) * 65536 + fb_next_fast_len(2 * g.bw - 1)].push_back(g);
    }
    int rc;
    for (auto& kv : by_shape) {
        const std::vector<Grid>& gs = kv.second;
        size_t nb2 = 0;
        int hmax = 0, wmax = 0;
        for (const Grid& g : gs) { nb2 += (size_t)g.nx * g.ny; hmax = std::max(hmax, g.bh); wmax = std::max(wmax, g.bw); }
        if (nb2 > m->max_blocks) return fb_fail(ctx, FB_ERR_ARG, "fb_match_strips: %zu blocks in the second shot of the global matcher (limit %zu)", nb2, m->max_blocks);
        std::vector<int32_t> b5(10 * nb2), b9(9 * nb2);
        size_t at = 0;
        for (const Grid& g : gs)
            for (int j = 0; j < g.ny; ++j)
                for (int i = 0; i < g.nx; ++i, ++at) {
                    const int x0 = round_i(cut_start(0.0, (double)m->wcs[g.p], g.cx, i)), y0 = round_i(cut_start(0.0, (double)m->hcs[g.p], g.cy, j));
                    for (int side = 0; side < 2; ++side) {
                        int32_t* d = &b5[5 * (2 * at + side)];
                        d[0] = side * n + g.p; d[1] = x0; d[2] = y0; d[3] = g.bh; d[4] = g.bw;
                    }
                    int32_t* d9 = &b9[9 * at];
                    d9[0] = g.p; d9[1] = x0; d9[2] = y0; d9[3] = g.bh; d9[4] = g.bw; d9[5] = x0; d9[6] = y0; d9[7] = g.bh; d9[8] = g.bw;
                }
        void *d_b5, *d_rng;
        if ((rc = scratch(ctx, m, kScrIds, b5.size() * 4, &d_b5)) || (rc = scratch(ctx, m, kScrOrg, 16 * nb2, &d_rng))) return rc;
        if ((rc = fb_memcpy_h2d(ctx, d_b5, b5.data(), b5.size() * 4))) return rc;
        hipLaunchKernelGGL(block_range_kernel, dim3((unsigned)(2 * nb2)), dim3(256), 0, ctx->stream, (const float*)m->d_dogc, hc, wc, (const int*)d_b5, (float*)d_rng);
        FB_HIP(ctx, hipGetLastError());
        std::vector<float> rng(4 * nb2);
        if ((rc = fb_memcpy_d2h(ctx, rng.data(), d_rng, rng.size() * 4))) return rc;
        if ((rc = fb_memcpy_h2d(ctx, m->d_blk, b9.data(), b9.size() * 4))) return rc;
        if ((rc = fb_ncc_blocks_dev(ctx, m->d_dogc, m->d_dogc + n * cpix, hc, wc, hc, wc, (int)nb2, m->d_blk, hmax, wmax, (int)(kv.first / 65536), (int)(kv.first % 65536), 0,
                                    m->conf_mode, (double*)m->d_out, (double*)(m->d_out + 8 * nb2), (float*)(m->d_out + 16 * nb2))))
            return rc;
        const double *bx, *by; const float* bc;
        if ((rc = fetch(ctx, m, nb2, &bx, &by, &bc))) return rc;
        at = 0;
        for (const Grid& g : gs) {
            const int nbk = g.nx * g.ny;
            int kbest = -1;
            for (int k = 0; k < nbk; ++k) {
                const size_t e = at + k;
                if (rng[4 * e] == rng[4 * e + 1] || rng[4 * e + 2] == rng[4 * e + 3]) continue;      // np.ptp(block) == 0 on either strip
                if (kbest < 0 || bc[e] > bc[at + kbest]) kbest = k;                                  // np.argmax: the first maximum
            }
            if (kbest >= 0 && bc[at + kbest] >= conf0[g.p]) { tx[g.p] = bx[at + kbest]; ty[g.p] = by[at + kbest]; conf0[g.p] = bc[at + kbest]; }
            at += nbk;
        }
    }
    return FB_OK;
}

}  // namespace

extern "C" {

int fb_divide_bbox(fb_ctx* ctx, const double* bbox, const double* block_hw, const int* min_blocks_yx, double shrink_factor, int round_output,
                   int* counts_xy, int* steps_xy, double* x_start, int cap_x, double* y_start, int cap_y) {
    FB_CHECK_ARG(ctx, bbox && block_hw && min_blocks_yx && counts_xy && steps_xy);
    FB_CHECK_ARG(ctx, bbox[2] > bbox[0] && bbox[3] > bbox[1] && block_hw[0] > 0 && block_hw[1] > 0 && shrink_factor > 0);
    const AxisCut cx = cut_axis(bbox[0], bbox[2], block_hw[1], (double)min_blocks_yx[1], shrink_factor);
    const AxisCut cy = cut_axis(bbox[1], bbox[3], block_hw[0], (double)min_blocks_yx[0], shrink_factor);
    counts_xy[0] = cx.count; counts_xy[1] = cy.count;
    steps_xy[0] = (int)cx.step; steps_xy[1] = (int)cy.step;
    if (x_start) {
        FB_CHECK_ARG(ctx, cap_x >= cx.count);
        for (int i = 0; i < cx.count; ++i) { const double v = cut_start(bbox[0], bbox[2], cx, i); x_start[i] = round_output ? std::nearbyint(v) : v; }
    }
    if (y_start) {
        FB_CHECK_ARG(ctx, cap_y >= cy.count);
        for (int j = 0; j < cy.count; ++j) { const double v = cut_start(bbox[1], bbox[3], cy, j); y_start[j] = round_output ? std::nearbyint(v) : v; }
    }
    return FB_OK;
}

#ifdef FB_TEST_HOOKS          // only in libfeabas_hip_test.so (include/feabas_hip_test.h)
#include "feabas_hip_test.h"
// test hooks (host only, no context): the host arithmetic fb_match_strips runs between its kernels -- the rigid fits of
// matcher.py:752-763 (spatial.fit_affine(return_rigid=True, svd_clip=(1, 1)) of every pair's matches; rows of a pair contiguous;
// R [P][3][3], bad [P] = 1 where the pair needs the statement-by-statement route), the automatic spacings of matcher.py:243-251
// (descending) and the node grid of Mesh.from_bbox(cartesian=True) (mesh.py:403-435).
int fb_debug_rigid_fits(int P, int64_t K, const int32_t* pid, const double* p0, const double* p1, const float* wt, double* R, uint8_t* bad) {
    if (P <= 0 || K < 0 || (K && (!pid || !p0 || !p1 || !wt)) || !R || !bad) return FB_ERR_ARG;
    for (int64_t k = 0; k < K; ++k)
        if (pid[k] < 0 || pid[k] >= P) return FB_ERR_ARG;
    std::vector<int32_t> vp(pid, pid + K);
    std::vector<double> v0(p0, p0 + 2 * K), v1(p1, p1 + 2 * K), Rv;
    std::vector<float> vw(wt, wt + K);
    std::vector<char> bv;
    rigid_fits(P, vp, v0, v1, vw, Rv, bv);
    std::copy(Rv.begin(), Rv.end(), R);
    for (int p = 0; p < P; ++p) bad[p] = (uint8_t)bv[p];
    return FB_OK;
}

int fb_debug_auto_spacings(int H, int W, double* out, int cap, int* count) {
    if (H < 1 || W < 1 || !count) return FB_ERR_ARG;
    std::vector<double> sp;
    auto_spacings(H, W, sp);
    *count = (int)sp.size();
    if (out) {
        if (cap < (int)sp.size()) return FB_ERR_ARG;
        std::copy(sp.begin(), sp.end(), out);
    }
    return FB_OK;
}

int fb_debug_grid_counts(int H, int W, double mesh_size, int min_num_blocks, int* nx, int* ny) {
    if (H < 1 || W < 1 || !(mesh_size > 0) || min_num_blocks < 1 || !nx || !ny) return FB_ERR_ARG;
    grid_counts(H, W, mesh_size, min_num_blocks, nx, ny);
    return FB_OK;
}
#endif

int fb_strip_matcher_create(fb_ctx* ctx, int P, int H, int W, const fb_strip_opts* o, fb_strip_matcher** out) {
    FB_LOCK(ctx);
    return matcher_create(ctx, P, H, W, nullptr, o, out);
}

int fb_strip_matcher_create_ragged(fb_ctx* ctx, int P, int H, int W, const int32_t* shapes, const fb_strip_opts* o, fb_strip_matcher** out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, shapes != nullptr);
    return matcher_create(ctx, P, H, W, shapes, o, out);
}

void fb_strip_matcher_destroy(fb_ctx* ctx, fb_strip_matcher* m) {
    if (!m) return;
    // whatever context the caller is on, the matcher's buffers (fb_malloc'ed by its creator) and its relaxation system return
    // to the context that made it: handed to another one, the pool would hold pointers that context does not own (fb_free
    // refuses them, so the bound of the pool was not enforced) and would hand them out after their owner is gone
    if (m->owner && m->owner != ctx) {
        FB_LOCK(ctx);
        hipStreamSynchronize(ctx->stream);                  // the caller's stream may have run the matcher's last kernels
        ctx = m->owner;
    }
    FB_LOCK(ctx);
    if (m->trace && (m->calls > 3 || m->trace_all)) {
        const int c = m->trace_all ? m->calls : m->calls - 3;
        std::fprintf(stderr, "fb_match_strips P=%d %dx%d: %d calls, ms per call (FEABAS_HIP_MATCH_TRACE=1: after the first 3): enqueue %.3f wait-global %.3f blocks-host %.3f blocks-launch %.3f blocks-wait %.3f relax %.3f table %.3f strain %.3f\n",
                     m->P, m->H, m->W, m->calls, m->t_stage[0] / c, m->t_stage[1] / c, m->t_stage[2] / c, m->t_stage[3] / c, m->t_stage[4] / c, m->t_stage[5] / c,
                     m->t_stage[6] / c, m->t_stage[7] / c);
        if (m->t_relax[1] > 0)
            std::fprintf(stderr, "    relax stage, general route: locate rows %.3f fb_pairs_relax_bary %.3f screen %.3f\n", m->t_relax[0] / c, m->t_relax[1] / c, m->t_relax[2] / c);
    }
    hipStreamSynchronize(ctx->stream);                      // nothing of this matcher is in flight when its buffers change hands
    if (m->sys) {
        auto& keep = ctx->match_systems[std::make_tuple(m->P, m->gnx, m->gny)];
        if (keep.size() < 2) keep.push_back(m->sys);
        else fb_sys_destroy(ctx, m->sys);
    }
    pool_give(ctx, m->d_dogc, m->b_dogc); pool_give(ctx, m->d_dogf, m->b_dogf); pool_give(ctx, m->d_small, m->b_small);
    pool_give(ctx, m->d_blk, m->b_blk); pool_give(ctx, m->d_out, m->b_out); pool_give(ctx, m->d_sizes, m->b_sizes);
    for (int k = 0; k < 14; ++k) pool_give(ctx, m->scr[k], m->b_scr[k]);
    delete m;
}

int fb_strip_matcher_info(fb_ctx* ctx, fb_strip_matcher* m, int* nspacings, double* spacings, int* grid_nx, int* grid_ny,
                          int* relax_iters, double* relax_relres, int* strain_iters, double* strain_relres) {
    FB_CHECK_ARG(ctx, m != nullptr);
    if (spacings) FB_CHECK_ARG(ctx, nspacings && *nspacings >= m->nsp);
    if (spacings) std::copy(m->sp.begin(), m->sp.begin() + m->nsp, spacings);      // of pair 0
    if (nspacings) *nspacings = m->nsp;
    if (grid_nx) *grid_nx = m->gnx;
    if (grid_ny) *grid_ny = m->gny;
    if (relax_iters) *relax_iters = m->relax_iters;
    if (relax_relres) *relax_relres = m->relax_relres;
    if (strain_iters) *strain_iters = m->strain_iters;
    if (strain_relres) *strain_relres = m->strain_relres;
    return FB_OK;
}

int fb_match_strips(fb_ctx* ctx, fb_strip_matcher* m, const uint8_t* strips0, const uint8_t* strips1, double* tx, double* ty, float* conf0,
                    uint8_t* valid, uint8_t* flags, double* strain, int64_t* nrows) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, m && strips0 && strips1 && tx && ty && conf0 && valid && flags && strain && nrows);
    struct Trusted {                                        // fb_common.h: trusted_links
        fb_ctx* c;
        explicit Trusted(fb_ctx* cc) : c(cc) { c->trusted_links = 1; }
        ~Trusted() { c->trusted_links = 0; }
    } trusted_guard(ctx);
    const int n = m->P, H = m->H, W = m->W, hc = m->hc, wc = m->wc;
    const size_t cpix = (size_t)hc * wc, fpix = (size_t)H * W;
    const float thr = (float)m->conf_thresh;               // numpy compares float32 confidences with the threshold in float32
    const int nsp = m->nsp;
    int rc;
    StageClock clk(m);
    if (m->trace && !m->trace_all && m->calls == 3) { for (double& t : m->t_stage) t = 0.0; }      // the first calls load code objects and grow arenas
    m->calls++;
    const float* dogf = m->d_dogc;                         // matcher.py:315-317: same image when fine == coarse
    std::vector<int32_t> blk;
    // ---- global translation on the coarse DoG images (matcher.py:255-278); the fine DoG is independent of the answer and
    //      is enqueued before the host waits for the global peaks
    // extras of this call (fb_strip_matcher_set_extras); they do not carry over to the next one
    std::vector<const uint8_t*> mask0, mask1;
    mask0.swap(m->mask0); mask1.swap(m->mask1);
    const bool want_phtm = m->want_phtm;
    m->want_phtm = false;
    bool any_mask = false;
    for (const uint8_t* q : mask0) any_mask |= q != nullptr;
    for (const uint8_t* q : mask1) any_mask |= q != nullptr;
    if (want_phtm && m->ragged) return fb_fail(ctx, FB_ERR_ARG, "fb_match_strips: photometric statistics are not taken on strips of unequal size");
    if (!m->ragged) {
        const bool extras = any_mask || want_phtm;
        uint8_t* d_maskc = nullptr;                          // coarse masks [2n][hc][wc] of the masked images
        std::vector<uint8_t> has_mask((size_t)2 * n, 0);
        if (any_mask) {
            void* ptr = nullptr;
            if ((rc = scratch(ctx, m, kScrMaskCoarse, 2 * (size_t)n * cpix, &ptr))) return rc;
            d_maskc = (uint8_t*)ptr;
        }
        if (m->cds2) {
            const int taps = (int)(4.0 * m->sigma * 0.5 + 0.5);
            if (!extras && (taps == 5 || taps == 6 || taps == 8 || taps == 10)) {
                if ((rc = fb_dog_down2_pair_dev(ctx, strips0, strips1, n, H, W, m->sigma * 0.5, 1, m->d_dogc))) return rc;
            } else {
                // (the coarse uint8 images are kept: the masked DoG and the photometric statistics read them again)
                if ((rc = fb_area_downsample2_dev(ctx, strips0, n, H, W, m->d_small))) return rc;
                if ((rc = fb_area_downsample2_dev(ctx, strips1, n, H, W, m->d_small + n * cpix))) return rc;
                if ((rc = fb_dog_dev(ctx, m->d_small, 0, 2 * n, hc, wc, m->sigma * 0.5, nullptr, 1, m->d_dogc))) return rc;
            }
        } else {
            if ((rc = fb_dog_pair_dev(ctx, strips0, strips1, 0, n, hc, wc, m->sigma, 1, m->d_dogc))) return rc;
        }
        if (any_mask) {
            // masked images: their coarse DoG again with the halo suppression of common.py:368-374; the coarse mask is
            // cv2.resize(mask, fx=0.5, INTER_NEAREST) = every second pixel (matcher.py:257-264)
            std::vector<uint8_t> mc(cpix);
            for (int side = 0; side < 2; ++side)
                for (int p = 0; p < n; ++p) {
                    const uint8_t* mk = (side ? mask1 : mask0).empty() ? nullptr : (side ? mask1 : mask0)[p];
                    if (!mk) continue;
                    const size_t img = (size_t)side * n + p;
                    const int st = m->cds2 ? 2 : 1;
                    for (int y = 0; y < hc; ++y)
                        for (int x = 0; x < wc; ++x) mc[(size_t)y * wc + x] = mk[(size_t)(st * y) * W + st * x] != 0;
                    has_mask[img] = 1;
                    if ((rc = fb_memcpy_h2d(ctx, d_maskc + img * cpix, mc.data(), cpix))) return rc;
                    const void* src = m->cds2 ? (const void*)(m->d_small + img * cpix) : (const void*)((side ? strips1 : strips0) + (size_t)p * fpix);
                    if ((rc = fb_dog_dev(ctx, src, 0, 1, hc, wc, m->cds2 ? m->sigma * 0.5 : m->sigma, d_maskc + img * cpix, 1, m->d_dogc + img * cpix))) return rc;
                }
        }
        if ((rc = fb_ncc_batch_dev(ctx, m->d_dogc, m->d_dogc + n * cpix, n, 1, hc, wc, hc, wc, 1, 0, m->conf_mode, (double*)m->d_out,
                                   (double*)(m->d_out + 8 * (size_t)n), (float*)(m->d_out + 16 * (size_t)n))))
            return rc;
        if (m->cds2) {
            if ((rc = fb_dog_pair_dev(ctx, strips0, strips1, 0, n, H, W, m->sigma, 1, m->d_dogf))) return rc;
            dogf = m->d_dogf;
            if (any_mask) {
                // (matcher.py:336-337: the fine DoG of a masked image with its full-resolution mask)
                void* ptr = nullptr;
                std::vector<uint8_t> mf(fpix);
                int slot = 0;
                for (int side = 0; side < 2; ++side)
                    for (int p = 0; p < n; ++p) {
                        const uint8_t* mk = (side ? mask1 : mask0).empty() ? nullptr : (side ? mask1 : mask0)[p];
                        if (!mk) continue;
                        for (size_t k = 0; k < fpix; ++k) mf[k] = mk[k] != 0;
                        // two slots alternate (copies and kernels are ordered on the context's stream)
                        if ((rc = scratch(ctx, m, (slot & 1) ? kScrMaskFine1 : kScrMaskFine0, fpix, &ptr))) return rc;
                        if ((rc = fb_memcpy_h2d(ctx, ptr, mf.data(), fpix))) return rc;
                        const size_t img = (size_t)side * n + p;
                        if ((rc = fb_dog_dev(ctx, (side ? strips1 : strips0) + (size_t)p * fpix, 0, 1, H, W, m->sigma, (const uint8_t*)ptr, 1, m->d_dogf + img * fpix))) return rc;
                        ++slot;
                    }
            }
        }
        clk.lap(0);
        const double *gx, *gy; const float* gc;
        if ((rc = fetch(ctx, m, (size_t)n, &gx, &gy, &gc))) return rc;
        for (int p = 0; p < n; ++p) { tx[p] = gx[p]; ty[p] = gy[p]; conf0[p] = gc[p]; }
        if ((rc = second_shot(ctx, m, thr, tx, ty, conf0))) return rc;
        if (want_phtm) {
            // matcher.py:279-314 at the coarse scale, with the global translation truncated like int() does
            const int bands = 16;
            std::vector<int32_t> txy((size_t)2 * n);
            for (int p = 0; p < n; ++p) { txy[2 * p] = (int32_t)tx[p]; txy[2 * p + 1] = (int32_t)ty[p]; }
            void *d_txy, *d_part, *d_has;
            if ((rc = scratch(ctx, m, kScrTxy, txy.size() * 4, &d_txy)) || (rc = scratch(ctx, m, kScrPart, (size_t)n * bands * 6 * 8, &d_part)) || (rc = scratch(ctx, m, kScrHas, 2 * (size_t)n, &d_has)))
                return rc;
            if ((rc = fb_memcpy_h2d(ctx, d_txy, txy.data(), txy.size() * 4)) || (rc = fb_memcpy_h2d(ctx, d_has, has_mask.data(), has_mask.size()))) return rc;
            const uint8_t* raw = m->cds2 ? m->d_small : nullptr;
            if (!m->cds2) {
                // coarse == fine: the raw images are the strips themselves, two stacks that need not be adjacent
                void* ptr = nullptr;
                if ((rc = scratch(ctx, m, kScrRaw, 2 * (size_t)n * cpix, &ptr))) return rc;
                if ((rc = fb_memcpy_d2d(ctx, ptr, strips0, (size_t)n * cpix)) || (rc = fb_memcpy_d2d(ctx, (uint8_t*)ptr + (size_t)n * cpix, strips1, (size_t)n * cpix))) return rc;
                raw = (const uint8_t*)ptr;
            }
            hipLaunchKernelGGL(photometric_kernel, dim3(n, bands), dim3(256), 0, ctx->stream, n, hc, wc, raw, (const float*)m->d_dogc, (const uint8_t*)d_maskc,
                               (const uint8_t*)d_has, (const int*)d_txy, bands, (double*)d_part);
            FB_HIP(ctx, hipGetLastError());
            std::vector<double> part((size_t)n * bands * 6);
            if ((rc = fb_memcpy_d2h(ctx, part.data(), d_part, part.size() * 8))) return rc;
            m->phtm.assign((size_t)4 * n, 0.0);
            m->phtm_has.assign((size_t)n, 0);
            for (int p = 0; p < n; ++p) {
                double t[6] = {0, 0, 0, 0, 0, 0};
                for (int bnd = 0; bnd < bands; ++bnd)
                    for (int k = 0; k < 6; ++k) t[k] += part[((size_t)p * bands + bnd) * 6 + k];
                if (!(t[5] > 3.0)) continue;                 // np.sum(m0) <= 3: None
                m->phtm_has[p] = 1;
                for (int k = 0; k < 4; ++k) m->phtm[4 * (size_t)p + k] = t[k] / t[4];       // (an empty joint mask gives nan, like np.mean of nothing)
            }
        } else {
            m->phtm.clear(); m->phtm_has.clear();
        }
        clk.lap(1);
    } else {
        // every stage on each pair's own extent ('nearest' extension at the image's own border, zeros in the rest of its slot)
        if (m->cds2) {
            if ((rc = fb_area_downsample2_sizes_dev(ctx, strips0, n, H, W, m->d_sizes, m->d_small))) return rc;
            if ((rc = fb_area_downsample2_sizes_dev(ctx, strips1, n, H, W, m->d_sizes, m->d_small + n * cpix))) return rc;
            if ((rc = fb_dog_sizes_dev(ctx, m->d_small, 0, 2 * n, hc, wc, m->d_sizes_c, m->sigma * 0.5, 1, m->d_dogc))) return rc;
            if ((rc = fb_dog_sizes_dev(ctx, strips0, 0, n, H, W, m->d_sizes, m->sigma, 1, m->d_dogf))) return rc;
            if ((rc = fb_dog_sizes_dev(ctx, strips1, 0, n, H, W, m->d_sizes, m->sigma, 1, m->d_dogf + n * fpix))) return rc;
            dogf = m->d_dogf;
        } else {
            if ((rc = fb_dog_sizes_dev(ctx, strips0, 0, n, hc, wc, m->d_sizes_c, m->sigma, 1, m->d_dogc))) return rc;
            if ((rc = fb_dog_sizes_dev(ctx, strips1, 0, n, hc, wc, m->d_sizes_c, m->sigma, 1, m->d_dogc + n * cpix))) return rc;
        }
        if (any_mask) {
            // masked images of a ragged batch (mask p: a contiguous uint8 array of the pair's OWN shape [Hs[p]][Ws[p]]): their DoG again
            // with the halo suppression of common.py:368-374 -- the image is copied out of its slot dense, filtered like the uniform
            // branch filters a whole slot, and copied back into the same corner of its float slot.  Coarse mask = every second pixel
            // (cv2.resize(mask, fx=0.5, INTER_NEAREST), matcher.py:257-264), fine DoG with the full-resolution mask (336-337).
            void *d_in = nullptr, *d_o = nullptr, *d_mk = nullptr;
            if ((rc = scratch(ctx, m, kScrDenseIn, fpix, &d_in)) || (rc = scratch(ctx, m, kScrDenseOut, 4 * fpix, &d_o)) || (rc = scratch(ctx, m, kScrMaskFine0, fpix, &d_mk))) return rc;
            std::vector<uint8_t> mb(fpix);
            for (int side = 0; side < 2; ++side)
                for (int p = 0; p < n; ++p) {
                    const uint8_t* mk = (side ? mask1 : mask0).empty() ? nullptr : (side ? mask1 : mask0)[p];
                    if (!mk) continue;
                    const size_t img = (size_t)side * n + p;
                    const int Hp = m->Hs[p], Wp = m->Ws[p], hp = m->hcs[p], wp = m->wcs[p];
                    const int st = m->cds2 ? 2 : 1;
                    // coarse
                    for (int y = 0; y < hp; ++y)
                        for (int x = 0; x < wp; ++x) mb[(size_t)y * wp + x] = mk[(size_t)(st * y) * Wp + st * x] != 0;
                    if ((rc = fb_memcpy_h2d(ctx, d_mk, mb.data(), (size_t)hp * wp))) return rc;
                    const uint8_t* src = m->cds2 ? m->d_small + img * cpix : (side ? strips1 : strips0) + (size_t)p * fpix;
                    if ((rc = fb_memcpy2d_d2d(ctx, d_in, (size_t)wp, src, (size_t)wc, (size_t)wp, (size_t)hp))) return rc;
                    if ((rc = fb_dog_dev(ctx, d_in, 0, 1, hp, wp, m->cds2 ? m->sigma * 0.5 : m->sigma, (const uint8_t*)d_mk, 1, (float*)d_o))) return rc;
                    if ((rc = fb_memcpy2d_d2d(ctx, m->d_dogc + img * cpix, 4 * (size_t)wc, d_o, 4 * (size_t)wp, 4 * (size_t)wp, (size_t)hp))) return rc;
                    if (!m->cds2) continue;
                    // fine
                    for (size_t k = 0; k < (size_t)Hp * Wp; ++k) mb[k] = mk[k] != 0;
                    if ((rc = fb_memcpy_h2d(ctx, d_mk, mb.data(), (size_t)Hp * Wp))) return rc;
                    if ((rc = fb_memcpy2d_d2d(ctx, d_in, (size_t)Wp, (side ? strips1 : strips0) + (size_t)p * fpix, (size_t)W, (size_t)Wp, (size_t)Hp))) return rc;
                    if ((rc = fb_dog_dev(ctx, d_in, 0, 1, Hp, Wp, m->sigma, (const uint8_t*)d_mk, 1, (float*)d_o))) return rc;
                    if ((rc = fb_memcpy2d_d2d(ctx, m->d_dogf + img * fpix, 4 * (size_t)W, d_o, 4 * (size_t)Wp, 4 * (size_t)Wp, (size_t)Hp))) return rc;
                }
        }
        clk.lap(0);
        // whole-strip NCC (matcher.py:153) through block descriptors, one launch per padded FFT shape
        std::map<long long, std::vector<int>> shapes;
        for (int p = 0; p < n; ++p) {
            // (the shape the correlation is RUN at: extents whose 5-smooth lengths differ -- 500 and 512 -- often share it)
            int fh_, fw_;
            fb_ncc_launch_shape(ctx, fb_next_fast_len(2 * m->hcs[p] - 1), fb_next_fast_len(2 * m->wcs[p] - 1), m->hcs[p], m->wcs[p], m->conf_mode, &fh_, &fw_);
            shapes[(long long)fh_ * 65536 + fw_].push_back(p);
        }
        // (all the launches are issued before the one copy that brings their results: a batch of 32 strips around 4096 x 510 has
        // two or three FFT shapes, and a copy between two launches drains the stream every time)
        blk.assign((size_t)n * 9, 0);
        {
            size_t at = 0;
            for (auto& kv : shapes)
                for (int p : kv.second) {
                    int32_t* d = &blk[9 * at++];
                    d[0] = p; d[3] = m->hcs[p]; d[4] = m->wcs[p]; d[7] = m->hcs[p]; d[8] = m->wcs[p];
                }
            if ((rc = fb_memcpy_h2d(ctx, m->d_blk, blk.data(), (size_t)n * 9 * 4))) return rc;
            at = 0;
            for (auto& kv : shapes) {
                const std::vector<int>& sel = kv.second;
                const size_t nb = sel.size();
                int hmax = 0, wmax = 0;
                for (int p : sel) { hmax = std::max(hmax, m->hcs[p]); wmax = std::max(wmax, m->wcs[p]); }
                uint8_t* o = m->d_out + 24 * at;
                if ((rc = fb_ncc_blocks_dev(ctx, m->d_dogc, m->d_dogc + n * cpix, hc, wc, hc, wc, (int)nb, m->d_blk + 9 * at, hmax, wmax, (int)(kv.first / 65536),
                                            (int)(kv.first % 65536), 0, m->conf_mode, (double*)o, (double*)(o + 8 * nb), (float*)(o + 16 * nb))))
                    return rc;
                at += nb;
            }
            if ((rc = fetch_all(ctx, m, (size_t)n))) return rc;
            at = 0;
            for (auto& kv : shapes) {
                const std::vector<int>& sel = kv.second;
                const size_t nb = sel.size();
                const uint8_t* o = m->raw.data() + 24 * at;
                const double *gx = (const double*)o, *gy = (const double*)(o + 8 * nb); const float* gc = (const float*)(o + 16 * nb);
                for (size_t q = 0; q < nb; ++q) { tx[sel[q]] = gx[q]; ty[sel[q]] = gy[q]; conf0[sel[q]] = gc[q]; }
                at += nb;
            }
        }
        if ((rc = second_shot(ctx, m, thr, tx, ty, conf0))) return rc;
        clk.lap(1);
    }
    const float* img1 = dogf + n * fpix;
    {
        const double scale = m->cds2 ? 2.0 : 1.0;          // matcher.py:338-339
        for (int p = 0; p < n; ++p) { tx[p] = tx[p] * scale; ty[p] = ty[p] * scale; }
    }
    std::vector<char> active((size_t)n), live((size_t)n), pad((size_t)n, 1), has_last((size_t)n, 0);
    for (int p = 0; p < n; ++p) {
        flags[p] = 0;
        active[p] = conf0[p] >= thr;                        // matcher.py:277-278
        live[p] = active[p] && !flags[p];
    }
    std::vector<double> t1((size_t)2 * n, 0.0);            // translation of mesh1 acquired by rigid relaxations
    m->is_def.assign((size_t)n, 0);
    m->tiers.assign((size_t)n, std::vector<int32_t>());
    bool any_def = false;
    Rows table, prev;
    bool have_table = false, last_links = false;
    std::vector<int> nxv((size_t)n), nyv((size_t)n), dxv((size_t)n), dyv((size_t)n), fhv((size_t)n), fwv((size_t)n);
    std::vector<double> xminv((size_t)n), yminv((size_t)n), xmaxv((size_t)n), ymaxv((size_t)n);
    std::vector<int> bbx0, bby0, xt, yt, order, ordc;
    std::vector<long long> zk, zkc;
    std::vector<char> in_cur((size_t)n), to_relax((size_t)n);
    std::vector<double> gdx, gdy;                          // results of a deformed group (the exact tier overwrites some)
    std::vector<float> gcf;
    for (int rnd = 0; rnd < nsp; ++rnd) {
        const bool is_last = rnd == nsp - 1;
        const int mnb = is_last ? m->mnb : 1;
        std::fill(to_relax.begin(), to_relax.end(), 0);
        // ---- group the live pairs by block grid and FFT shape (matcher.py:59-62 on the block size); pairs with a deformed
        //      mesh1 form their own groups (per pad flag and block size: their windows are rendered, not cropped)
        std::map<long long, std::vector<int>> groups;
        std::map<std::array<int, 7>, std::vector<int>> dgroups;
        const int gV = m->gnx * m->gny;
        for (int p = 0; p < n; ++p) {
            if (!live[p]) continue;
            const double spc = m->sp[(size_t)p * nsp + rnd];
            const double Wp = (double)m->Ws[p], Hp = (double)m->Hs[p];
            double xmin, xmax, ymin, ymax;
            if (!m->is_def[p]) {
                xmin = std::max(-0.5 + tx[p], -0.5 + t1[2 * p]); xmax = std::min(Wp - 0.5 + tx[p], Wp - 0.5 + t1[2 * p]);
                ymin = std::max(-0.5 + ty[p], -0.5 + t1[2 * p + 1]); ymax = std::min(Hp - 0.5 + ty[p], Hp - 0.5 + t1[2 * p + 1]);
            } else {
                // intersection with the bounding box of the deformed mesh1 (matcher.py:877)
                const double inf = std::numeric_limits<double>::infinity();
                double vx0 = inf, vx1 = -inf, vy0 = inf, vy1 = -inf;
                const double* u = &m->U[(size_t)2 * p * gV];
                for (int j = 0; j < m->gny; ++j)
                    for (int i = 0; i < m->gnx; ++i) {
                        const double vx = m->gxs[(size_t)p * m->gnx + i] + u[2 * (j * m->gnx + i)], vy = m->gys[(size_t)p * m->gny + j] + u[2 * (j * m->gnx + i) + 1];
                        vx0 = std::min(vx0, vx); vx1 = std::max(vx1, vx); vy0 = std::min(vy0, vy); vy1 = std::max(vy1, vy);
                    }
                xmin = std::max(-0.5 + tx[p], vx0); xmax = std::min(Wp - 0.5 + tx[p], vx1);
                ymin = std::max(-0.5 + ty[p], vy0); ymax = std::min(Hp - 0.5 + ty[p], vy1);
            }
            if (!(xmax > xmin && ymax > ymin)) continue;
            // common.divide_bbox (common.py:380-409)
            const AxisCut cx = cut_axis(xmin, xmax, spc, (double)mnb, 1.0), cy = cut_axis(ymin, ymax, spc, (double)mnb, 1.0);
            const double nx = (double)cx.count, ny = (double)cy.count;
            const long long dx = cx.step, dy = cy.step;
            if (dx < 1 || dy < 1 || 2 * dx > 8192 || 2 * dy > 8192 || nx >= 4096 || ny >= 4096)
                return fb_fail(ctx, FB_ERR_ARG, "fb_match_strips: pair %d: block %lld x %lld on a %g x %g grid is outside the block matcher", p, dy, dx, ny, nx);
            xminv[p] = xmin; xmaxv[p] = xmax; yminv[p] = ymin; ymaxv[p] = ymax;
            nxv[p] = (int)nx; nyv[p] = (int)ny; dxv[p] = (int)dx; dyv[p] = (int)dy;
            fhv[p] = fb_next_fast_len(pad[p] ? std::max(2 * (int)dy - 1, 0) : (int)dy);
            fwv[p] = fb_next_fast_len(pad[p] ? std::max(2 * (int)dx - 1, 0) : (int)dx);
            if (!m->is_def[p]) {
                // pairs are grouped by the shape their blocks are RUN at (unequal strips: block sizes a few pixels apart, whose
                // 5-smooth lengths differ, share the promoted power of two of the streaming class)
                if (m->ragged) fb_ncc_launch_shape(ctx, fhv[p], fwv[p], (int)dy, (int)dx, m->conf_mode, &fhv[p], &fwv[p]);
                const long long key = ((((long long)nx * 4096 + (long long)ny) * 8192 + fhv[p]) * 8192 + fwv[p]);
                groups[key].push_back(p);
            } else {
                dgroups[{(int)nx, (int)ny, fhv[p], fwv[p], (int)pad[p], (int)dx, (int)dy}].push_back(p);
            }
        }
        if (groups.empty() && dgroups.empty()) continue;
        Rows cur;
        {
            size_t cap = 0;
            for (auto& kv : groups) cap += kv.second.size() * (size_t)nxv[kv.second[0]] * nyv[kv.second[0]];
            for (auto& kv : dgroups) cap += kv.second.size() * (size_t)nxv[kv.second[0]] * nyv[kv.second[0]];
            cur.reserve(cap + (have_table ? table.size() : 0));
        }
        // block descriptors of pair p (slot q of its group): starts = round(linspace(lo, hi - step, count)) per axis, blocks in the
        // z-order of their index grid (common.z_order, common.py:196-215; stable)
        bool have_c = false;
        size_t goff = 0;                                    // first block of the group being described inside blk / bbx0 / bby0
        auto pair_blocks = [&](int q, int p, int nxi, int nyi, bool deformed) {
            const int nblk = nxi * nyi, dx = dxv[p], dy = dyv[p];
            const double spc = m->sp[(size_t)p * nsp + rnd];
            xt.resize(nxi); yt.resize(nyi);
            for (int i = 0; i < nxi; ++i) xt[i] = round_i(linspace_at(xminv[p], xmaxv[p] - (double)dx, nxi, i));
            for (int j = 0; j < nyi; ++j) yt[j] = round_i(linspace_at(yminv[p], ymaxv[p] - (double)dy, nyi, j));
            const int xlo = *std::min_element(xt.begin(), xt.end()), ylo = *std::min_element(yt.begin(), yt.end());
            zk.resize(nblk);
            for (int j = 0; j < nyi; ++j)
                for (int i = 0; i < nxi; ++i) {
                    unsigned ix = (unsigned)std::nearbyint((double)(xt[i] - xlo) / spc), iy = (unsigned)std::nearbyint((double)(yt[j] - ylo) / spc);
                    long long k = 0;
                    for (int l = 0; ix || iy; ++l, ix >>= 1, iy >>= 1) k += ((long long)(ix & 1) + 2 * (long long)(iy & 1)) << (2 * l);
                    zk[(size_t)j * nxi + i] = k;
                }
            if (!have_c || zk != zkc) {
                order.resize(nblk);
                std::iota(order.begin(), order.end(), 0);
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return zk[a] < zk[b]; });
                ordc = order; zkc = zk; have_c = true;
            }
            const int rtx = round_i(tx[p]), rty = round_i(ty[p]), r1x = round_i(t1[2 * p]), r1y = round_i(t1[2 * p + 1]);
            for (int b = 0; b < nblk; ++b) {
                const int o = ordc[b], x0 = xt[o % nxi], y0 = yt[o / nxi];
                const size_t at = goff + (size_t)q * nblk + b;
                bbx0[at] = x0; bby0[at] = y0;
                int32_t* d = &blk[9 * at];
                d[0] = p; d[1] = x0 - rtx; d[2] = y0 - rty; d[3] = dy; d[4] = dx; d[7] = dy; d[8] = dx;
                if (deformed) { d[5] = 0; d[6] = 0; }          // the window of image 1 comes from the affine / exact gather
                else { d[5] = x0 - r1x; d[6] = y0 - r1y; }
            }
        };
        // spacing schedule of pair p after a round that measured max_dis (matcher.py:689-716), max_spacing_skip = 0
        auto schedule = [&](int p, double max_dis) {
            int next_pos = -1;
            for (int k = 0; k < nsp; ++k) next_pos += m->sp[(size_t)p * nsp + k] > 4.0 * max_dis;
            pad[p] = !(next_pos > rnd);
        };
        // the groups of a round are described, launched and fetched together: one copy of the descriptors, the launches one behind
        // the other, one copy of the results (a chunk of unequal strips has several groups per round; a copy between two
        // launches drained the stream every time)
        std::vector<size_t> gbase;
        size_t gtotal = 0;
        for (auto& kv : groups) {
            gbase.push_back(gtotal);
            gtotal += kv.second.size() * (size_t)nxv[kv.second[0]] * nyv[kv.second[0]];
        }
        if (gtotal > m->max_blocks) return fb_fail(ctx, FB_ERR_ARG, "fb_match_strips: %zu blocks in one round (limit %zu)", gtotal, m->max_blocks);
        if (gtotal) {
            blk.resize(gtotal * 9); bbx0.resize(gtotal); bby0.resize(gtotal);
            size_t gi = 0;
            for (auto& kv : groups) {
                const std::vector<int>& sel = kv.second;
                have_c = false;
                goff = gbase[gi++];
                for (int q = 0; q < (int)sel.size(); ++q) pair_blocks(q, sel[q], nxv[sel[0]], nyv[sel[0]], false);
            }
            goff = 0;
            clk.lap(2);
            if ((rc = fb_memcpy_h2d(ctx, m->d_blk, blk.data(), gtotal * 9 * 4))) return rc;
            gi = 0;
            for (auto& kv : groups) {
                const std::vector<int>& sel = kv.second;
                const size_t nb = sel.size() * (size_t)nxv[sel[0]] * nyv[sel[0]], gb = gbase[gi++];
                int hmax = 0, wmax = 0;
                for (int p : sel) { hmax = std::max(hmax, dyv[p]); wmax = std::max(wmax, dxv[p]); }
                uint8_t* o = m->d_out + 24 * gb;
                if ((rc = fb_ncc_blocks_dev(ctx, dogf, img1, H, W, H, W, (int)nb, m->d_blk + 9 * gb, hmax, wmax, fhv[sel[0]], fwv[sel[0]], is_last ? 1 : 0,
                                            m->conf_mode, (double*)o, (double*)(o + 8 * nb), (float*)(o + 16 * nb))))
                    return rc;
            }
            clk.lap(3);
            if ((rc = fetch_all(ctx, m, gtotal))) return rc;
            clk.lap(4);
        }
        size_t gidx = 0;
        for (auto& kv : groups) {
            const std::vector<int>& sel = kv.second;
            const int Q = (int)sel.size(), nxi = nxv[sel[0]], nyi = nyv[sel[0]], nblk = nxi * nyi;
            const size_t nb = (size_t)Q * nblk, gb = gbase[gidx++];
            const uint8_t* o = m->raw.data() + 24 * gb;
            const double *ddx = (const double*)o, *ddy = (const double*)(o + 8 * nb); const float* dcf = (const float*)(o + 16 * nb);
            const int* gx0 = bbx0.data() + gb; const int* gy0 = bby0.data() + gb;       // block origins of this group
            // ---- blocks -> point pairs (matcher.py:671-683, 840-849), spacing schedule (689-716), rigid relaxation (725-742)
            for (int q = 0; q < Q; ++q) {
                const int p = sel[q];
                const int dx = dxv[p], dy = dyv[p];
                const size_t base = (size_t)q * nblk;
                double dis2max = -1.0;
                int first = -1;
                for (int b = 0; b < nblk; ++b) {
                    if (!(dcf[base + b] > thr)) continue;
                    if (first < 0) first = b;
                    const double hx = ddx[base + b] * 0.5, hy = ddy[base + b] * 0.5;
                    const double cx = 0.5 * (double)(gx0[base + b] + (gx0[base + b] + dx)) - 0.5, cy = 0.5 * (double)(gy0[base + b] + (gy0[base + b] + dy)) - 0.5;
                    const double ex = (cx - hx) - (cx + hx), ey = (cy - hy) - (cy + hy);
                    dis2max = std::max(dis2max, ex * ex + ey * ey);
                }
                const bool has_link = first >= 0;
                const double max_dis = std::sqrt(std::max(dis2max, 0.0));
                if (!has_link) {
                    if (rnd == 0) active[p] = 0;            // invalid_output (matcher.py:672-673, 719-721)
                    live[p] = 0;                            // ... or break with the links so far (674-675, 722-723)
                }
                const double t1x = t1[2 * p], t1y = t1[2 * p + 1];        // mesh1 offset at link creation (matcher.py:748-751)
                if (!is_last) {
                    schedule(p, max_dis);
                    if (has_link && max_dis > 0.1) {
                        // every kept block reports the same displacement: the relaxation of mesh1 is that translation (zero
                        // elastic and zero link energy); any other field is solved below and makes the pair deformed
                        bool nonrigid = false;
                        const double hx0 = ddx[base + first] * 0.5, hy0 = ddy[base + first] * 0.5;
                        const double c0x = 0.5 * (double)(gx0[base + first] + (gx0[base + first] + dx)) - 0.5, c0y = 0.5 * (double)(gy0[base + first] + (gy0[base + first] + dy)) - 0.5;
                        const double u0x = (c0x - hx0) - (c0x + hx0), u0y = (c0y - hy0) - (c0y + hy0);
                        for (int b = 0; b < nblk && !nonrigid; ++b) {
                            if (!(dcf[base + b] > thr)) continue;
                            const double hx = ddx[base + b] * 0.5, hy = ddy[base + b] * 0.5;
                            const double cx = 0.5 * (double)(gx0[base + b] + (gx0[base + b] + dx)) - 0.5, cy = 0.5 * (double)(gy0[base + b] + (gy0[base + b] + dy)) - 0.5;
                            nonrigid = ((cx - hx) - (cx + hx)) != u0x || ((cy - hy) - (cy + hy)) != u0y;
                        }
                        if (!nonrigid) { t1[2 * p] += u0x; t1[2 * p + 1] += u0y; }
                        else to_relax[p] = 1;
                    }
                }
                const char rl = max_dis > 0.1;
                for (int b = 0; b < nblk; ++b) {
                    if (!(dcf[base + b] > thr)) continue;
                    const double hx = ddx[base + b] * 0.5, hy = ddy[base + b] * 0.5;
                    const double cx = 0.5 * (double)(gx0[base + b] + (gx0[base + b] + dx)) - 0.5, cy = 0.5 * (double)(gy0[base + b] + (gy0[base + b] + dy)) - 0.5;
                    const double x1 = cx + hx, y1 = cy + hy;
                    cur.push(p, cx - hx, cy - hy, x1 - t1x, y1 - t1y, x1, y1, dcf[base + b], rl);
                }
                if (has_link) has_last[p] = 1;
            }
        }
        // ---- groups of pairs whose mesh1 is deformed (matcher.py:833-846 -> MeshRenderer.crop_multiple, renderer.py:397-563)
        for (auto& kv : dgroups) {
            const std::vector<int>& sel = kv.second;
            const int Q = (int)sel.size(), nxi = kv.first[0], nyi = kv.first[1], nblk = nxi * nyi, gfh = kv.first[2], gfw = kv.first[3], gpad = kv.first[4];
            const int w = kv.first[5], h = kv.first[6];
            const int gnx = m->gnx, gny = m->gny, V = gnx * gny;
            const size_t nb = (size_t)Q * nblk, px = (size_t)h * w;
            if (nb > m->max_blocks) return fb_fail(ctx, FB_ERR_ARG, "fb_match_strips: %zu blocks in one launch (limit %zu)", nb, m->max_blocks);
            blk.resize(nb * 9); bbx0.resize(nb); bby0.resize(nb);
            have_c = false;
            for (int q = 0; q < Q; ++q) pair_blocks(q, sel[q], nxi, nyi, true);
            // tiers and affine maps of every block (affine_approx_tol = 0.1 in the last round, max(1, 0.02 spacing) before:
            // matcher.py:578-603), on the MOVING nodes of each pair's mesh1
            std::vector<double> vmg((size_t)2 * Q * V), gxs_q((size_t)Q * gnx), gys_q((size_t)Q * gny), tolg((size_t)Q), A6(6 * nb), lo((size_t)2 * Q), aff(10 * nb, 0.0);
            std::vector<int32_t> bb(4 * nb), tier(nb);
            for (int q = 0; q < Q; ++q) {
                const int p = sel[q];
                std::copy(m->gxs.begin() + (size_t)p * gnx, m->gxs.begin() + (size_t)(p + 1) * gnx, gxs_q.begin() + (size_t)q * gnx);
                std::copy(m->gys.begin() + (size_t)p * gny, m->gys.begin() + (size_t)(p + 1) * gny, gys_q.begin() + (size_t)q * gny);
                const double* u = &m->U[(size_t)2 * p * V];
                for (int j = 0; j < gny; ++j)
                    for (int i = 0; i < gnx; ++i) {
                        const size_t k = 2 * ((size_t)q * V + (size_t)j * gnx + i);
                        vmg[k] = gxs_q[(size_t)q * gnx + i] + u[2 * (j * gnx + i)]; vmg[k + 1] = gys_q[(size_t)q * gny + j] + u[2 * (j * gnx + i) + 1];
                    }
                tolg[q] = is_last ? 0.1 : std::max(1.0, 0.02 * m->sp[(size_t)p * nsp + rnd]);
                for (int b = 0; b < nblk; ++b) {
                    const size_t at = (size_t)q * nblk + b;
                    bb[4 * at] = bbx0[at]; bb[4 * at + 1] = bby0[at]; bb[4 * at + 2] = bbx0[at] + w; bb[4 * at + 3] = bby0[at] + h;
                }
            }
            if ((rc = fb_deformed_block_affines(ctx, Q, gnx, gny, gxs_q.data(), gys_q.data(), 1, vmg.data(), nblk, bb.data(), 0.0, tolg.data(), tier.data(),
                                                A6.data(), lo.data())))
                return rc;
            // the exact piecewise-linear field of the blocks no affine map follows (renderer.py:511-563)
            struct Exact { int q, b; size_t at; };
            std::vector<Exact> exact;
            std::vector<double> emx, emy;
            std::vector<uint8_t> emk;
            for (int q = 0; q < Q; ++q) {
                const int p = sel[q];
                bool folded = false;
                for (int b = 0; b < nblk; ++b) folded |= tier[(size_t)q * nblk + b] < 0;
                if (folded) { flags[p] |= FB_STRIP_FOLDED; continue; }      // the statement-by-statement host route renders this pair
                m->tiers[p].assign(tier.begin() + (size_t)q * nblk, tier.begin() + (size_t)(q + 1) * nblk);
                // one remap origin for the whole stack of a pair (render_by_subregions, common.py:316-321): floor(min of the rendered
                // maps) - 4; an affine map takes its extremes at the corner pixels (lo)
                double lo_x = lo[2 * q], lo_y = lo[2 * q + 1];
                std::vector<int32_t> po, org;
                const size_t e0 = exact.size();
                for (int b = 0; b < nblk; ++b)
                    if (tier[(size_t)q * nblk + b] == 3) {
                        exact.push_back({q, b, exact.size() * px});
                        po.push_back(0); org.push_back(bbx0[(size_t)q * nblk + b]); org.push_back(bby0[(size_t)q * nblk + b]);
                    }
                const size_t ne = exact.size() - e0;
                if (ne) {
                    emx.resize(exact.size() * px); emy.resize(exact.size() * px); emk.resize(exact.size() * px);
                    if ((rc = fb_deformed_exact_field(ctx, 1, gnx, gny, &gxs_q[(size_t)q * gnx], &gys_q[(size_t)q * gny], 0, &vmg[(size_t)2 * q * V], (int)ne, po.data(),
                                                      org.data(), h, w, &emx[e0 * px], &emy[e0 * px], &emk[e0 * px])))
                        return rc;
                    for (size_t k = e0 * px; k < exact.size() * px; ++k)
                        if (emk[k]) { lo_x = std::min(lo_x, emx[k]); lo_y = std::min(lo_y, emy[k]); }
                }
                const bool fin = std::isfinite(lo_x);
                for (int b = 0; b < nblk; ++b) {
                    const size_t at = (size_t)q * nblk + b;
                    double* a = &aff[10 * at];
                    a[0] = (double)bbx0[at]; a[1] = (double)bby0[at];
                    for (int k = 0; k < 6; ++k) a[2 + k] = A6[6 * at + k];
                    if (fin) { a[8] = std::floor(lo_x) - 4.0; a[9] = std::floor(lo_y) - 4.0; }
                }
            }
            clk.lap(2);
            void* d_aff = nullptr;
            if ((rc = scratch(ctx, m, kScrAff, aff.size() * 8, &d_aff))) return rc;
            if ((rc = fb_memcpy_h2d(ctx, m->d_blk, blk.data(), nb * 9 * 4)) || (rc = fb_memcpy_h2d(ctx, d_aff, aff.data(), aff.size() * 8))) return rc;
            if ((rc = fb_ncc_blocks_affine_dev(ctx, dogf, img1, H, W, H, W, (int)nb, m->d_blk, (const double*)d_aff, h, w, gfh, gfw, is_last ? 1 : 0, m->conf_mode,
                                               (double*)m->d_out, (double*)(m->d_out + 8 * nb), (float*)(m->d_out + 16 * nb))))
                return rc;
            clk.lap(3);
            {
                const double *ddx, *ddy; const float* dcf;
                if ((rc = fetch(ctx, m, nb, &ddx, &ddy, &dcf))) return rc;
                gdx.assign(ddx, ddx + nb); gdy.assign(ddy, ddy + nb); gcf.assign(dcf, dcf + nb);
            }
            if (!exact.empty()) {
                // both windows of these blocks are materialised (fb_remap_dev; the image-0 window through an integer map) and
                // correlated as a stack (fb_ncc_batch_dev)
                const size_t ne = exact.size(), N2 = 2 * ne;
                std::vector<int32_t> ids(N2), eorg(2 * N2);
                std::vector<float> mxa(N2 * px), mya(N2 * px);
                std::vector<uint8_t> mka(N2 * px, 1);
                for (size_t e = 0; e < ne; ++e) {
                    const Exact& x = exact[e];
                    const size_t at = (size_t)x.q * nblk + x.b;
                    const double ox = aff[10 * at + 8], oy = aff[10 * at + 9];
                    ids[e] = sel[x.q]; eorg[2 * e] = blk[9 * at + 1]; eorg[2 * e + 1] = blk[9 * at + 2];
                    ids[ne + e] = n + sel[x.q]; eorg[2 * (ne + e)] = (int32_t)ox; eorg[2 * (ne + e) + 1] = (int32_t)oy;
                    for (int r = 0; r < h; ++r)
                        for (int c = 0; c < w; ++c) {
                            const size_t k = (size_t)r * w + c;
                            mxa[e * px + k] = (float)c; mya[e * px + k] = (float)r;
                            mxa[(ne + e) * px + k] = (float)(emx[x.at + k] - ox); mya[(ne + e) * px + k] = (float)(emy[x.at + k] - oy);
                            mka[(ne + e) * px + k] = emk[x.at + k];
                        }
                }
                void *d_ids, *d_org, *d_mx, *d_my, *d_mk, *d_st, *d_res;
                if ((rc = scratch(ctx, m, kScrIds, 4 * N2, &d_ids)) || (rc = scratch(ctx, m, kScrOrg, 8 * N2, &d_org)) || (rc = scratch(ctx, m, kScrMapX, 4 * N2 * px, &d_mx)) ||
                    (rc = scratch(ctx, m, kScrMapY, 4 * N2 * px, &d_my)) || (rc = scratch(ctx, m, kScrMapMask, N2 * px, &d_mk)) || (rc = scratch(ctx, m, kScrStack, 4 * N2 * px, &d_st)) ||
                    (rc = scratch(ctx, m, kScrExactOut, 20 * ne, &d_res)))
                    return rc;
                if ((rc = fb_memcpy_h2d(ctx, d_ids, ids.data(), 4 * N2)) || (rc = fb_memcpy_h2d(ctx, d_org, eorg.data(), 8 * N2)) ||
                    (rc = fb_memcpy_h2d(ctx, d_mx, mxa.data(), 4 * N2 * px)) || (rc = fb_memcpy_h2d(ctx, d_my, mya.data(), 4 * N2 * px)) ||
                    (rc = fb_memcpy_h2d(ctx, d_mk, mka.data(), N2 * px)))
                    return rc;
                if ((rc = fb_remap_dev(ctx, dogf, H, W, (int)N2, (const int*)d_ids, h, w, (const float*)d_mx, (const float*)d_my, (const uint8_t*)d_mk, (const int*)d_org,
                                       (float*)d_st)))
                    return rc;
                uint8_t* r8 = (uint8_t*)d_res;
                if ((rc = fb_ncc_batch_dev(ctx, (const float*)d_st, (const float*)d_st + ne * px, (int)ne, 1, h, w, h, w, gpad, is_last ? 1 : 0, m->conf_mode, (double*)r8,
                                           (double*)(r8 + 8 * ne), (float*)(r8 + 16 * ne))))
                    return rc;
                m->raw.resize(20 * ne);
                if ((rc = fb_memcpy_d2h(ctx, m->raw.data(), d_res, 20 * ne))) return rc;
                const double* ex = (const double*)m->raw.data();
                const double* ey = (const double*)(m->raw.data() + 8 * ne);
                const float* ec = (const float*)(m->raw.data() + 16 * ne);
                for (size_t e = 0; e < ne; ++e) {
                    const size_t at = (size_t)exact[e].q * nblk + exact[e].b;
                    gdx[at] = ex[e]; gdy[at] = ey[e]; gcf[at] = ec[e];
                }
            }
            clk.lap(4);
            // ---- blocks -> point pairs; Link.from_coordinates on the MOVING gear of the deformed mesh1 (matcher.py:717,
            //      optimizer.py:51-82): points outside the mesh are dropped, the INITIAL coordinates follow from the barycentric ones
            std::vector<int32_t> pq, ptid;
            std::vector<double> pts, pB;
            std::vector<size_t> pat;
            for (int q = 0; q < Q; ++q) {
                if (flags[sel[q]] & FB_STRIP_FOLDED) continue;
                for (int b = 0; b < nblk; ++b) {
                    const size_t at = (size_t)q * nblk + b;
                    if (!(gcf[at] > thr)) continue;
                    const double cx = 0.5 * (double)(bbx0[at] + (bbx0[at] + w)) - 0.5, cy = 0.5 * (double)(bby0[at] + (bby0[at] + h)) - 0.5;
                    pq.push_back(q); pat.push_back(at);
                    pts.push_back(cx + gdx[at] * 0.5); pts.push_back(cy + gdy[at] * 0.5);
                }
            }
            ptid.resize(pq.size()); pB.resize(3 * pq.size());
            if (!pq.empty() && (rc = fb_deformed_locate(ctx, Q, gnx, gny, gxs_q.data(), gys_q.data(), 1, vmg.data(), (int64_t)pq.size(), pq.data(), pts.data(), ptid.data(),
                                                        pB.data())))
                return rc;
            size_t cursor = 0;
            for (int q = 0; q < Q; ++q) {
                const int p = sel[q];
                if (flags[p] & FB_STRIP_FOLDED) { live[p] = 0; continue; }
                double dis2max = -1.0;
                bool has_link = false;
                const size_t c0 = cursor;
                for (; cursor < pq.size() && pq[cursor] == q; ++cursor) {
                    const size_t at = pat[cursor];
                    const double hx = gdx[at] * 0.5, hy = gdy[at] * 0.5;
                    const double cx = 0.5 * (double)(bbx0[at] + (bbx0[at] + w)) - 0.5, cy = 0.5 * (double)(bby0[at] + (bby0[at] + h)) - 0.5;
                    const double ex = (cx - hx) - (cx + hx), ey = (cy - hy) - (cy + hy);
                    dis2max = std::max(dis2max, ex * ex + ey * ey);      // before the inside test, like the reference
                    has_link |= ptid[cursor] >= 0;
                }
                const double max_dis = std::sqrt(std::max(dis2max, 0.0));
                if (!has_link) {
                    if (rnd == 0) active[p] = 0;
                    live[p] = 0;
                }
                if (!is_last) {
                    schedule(p, max_dis);
                    if (has_link && max_dis > 0.1) to_relax[p] = 1;
                }
                const char rl = max_dis > 0.1;
                for (size_t c = c0; c < cursor; ++c) {
                    if (ptid[c] < 0) continue;
                    const size_t at = pat[c];
                    const double hx = gdx[at] * 0.5, hy = gdy[at] * 0.5;
                    const double cx = 0.5 * (double)(bbx0[at] + (bbx0[at] + w)) - 0.5, cy = 0.5 * (double)(bby0[at] + (bby0[at] + h)) - 0.5;
                    // triangle of the grid cell (a b / c d): 2 cell = (a, b, d), 2 cell + 1 = (a, d, c)
                    const int cell = ptid[c] / 2, cj = cell / (gnx - 1), ci = cell % (gnx - 1), na = cj * gnx + ci;
                    const int loc[3] = {na, (ptid[c] & 1) ? na + gnx + 1 : na + 1, (ptid[c] & 1) ? na + gnx : na + gnx + 1};
                    int32_t nodes[3];
                    double fx[3], fy[3];
                    for (int a = 0; a < 3; ++a) {
                        nodes[a] = loc[a] + p * V;
                        fx[a] = m->gxs[(size_t)p * gnx + loc[a] % gnx]; fy[a] = m->gys[(size_t)p * gny + loc[a] / gnx];
                    }
                    const double* bq = &pB[3 * c];
                    const double xi = (fx[0] * bq[0] + fx[1] * bq[1]) + fx[2] * bq[2], yi = (fy[0] * bq[0] + fy[1] * bq[1]) + fy[2] * bq[2];
                    cur.push(p, cx - hx, cy - hy, xi, yi, cx + hx, cy + hy, gcf[at], rl, nodes, bq);
                }
                if (has_link) has_last[p] = 1;
            }
        }
        prev = std::move(table);
        const bool had_prev = have_table;
        table = std::move(cur);
        have_table = true;
        last_links = false;
        clk.lap(2);
        bool any_relax = false;
        for (int p = 0; p < n; ++p) any_relax |= to_relax[p] != 0;
        if (any_relax) {
            // non-rigid relaxation between spacings (matcher.py:725-741): mesh1 of these pairs keeps the node field
            std::vector<size_t> rows;
            for (size_t k = 0; k < table.size(); ++k) if (to_relax[table.pid[k]]) rows.push_back(k);
            std::vector<float> rw;
            std::vector<double> x;
            if ((rc = relax_general(ctx, m, table, rows, true, rw, x, flags))) return rc;
            const int V = m->gnx * m->gny;
            if (m->U.size() != (size_t)2 * n * V) m->U.assign((size_t)2 * n * V, 0.0);
            for (int p = 0; p < n; ++p) {
                if (!to_relax[p]) continue;
                std::copy(x.begin() + (size_t)2 * p * V, x.begin() + (size_t)2 * (p + 1) * V, m->U.begin() + (size_t)2 * p * V);
                m->is_def[p] = 1; any_def = true;
                if (flags[p]) live[p] = 0;                   // beyond the screen of relax_first: the general route finishes the pair
            }
            for (size_t k = 0; k < rows.size(); ++k) table.wt[rows[k]] = table.wt[rows[k]] * rw[k];      // Link.weight (optimizer.py:313-317)
        }
        if (is_last && m->residue_len > 0 && table.size()) {
            bool any_rl = false;
            for (char r : table.rl) any_rl |= r != 0;
            if (any_rl) {
                // last round (matcher.py:725-737): relaxation + residue weights of every pair at once; all rows enter
                // the block-diagonal system so that the strain stage can reuse the links
                if ((rc = ensure_system(ctx, m))) return rc;
                const int nx = m->gnx, ny = m->gny, V = nx * ny;
                const int64_t K = (int64_t)table.size();
                std::vector<float> rw((size_t)K);
                std::vector<double> x;
                if (!m->ragged && !any_def) {
                    x.resize((size_t)2 * n * V);
                    if ((rc = fb_pairs_relax(ctx, m->sys, n, nx, ny, m->gxs.data(), m->gys.data(), K, table.pid.data(), table.xy0.data(), table.xy1i.data(),
                                             t1.data(), table.wt.data(), m->residue_len, m->residue_mode, m->se0, m->stiffness_lambda, m->relax_tol,
                                             rw.data(), x.data(), &m->relax_iters, &m->relax_relres)))
                        return rc;
                    m->relax_matches = K;
                    last_links = true;
                    screen_relax_first(m, x, flags);
                } else {
                    // one mesh geometry per pair, or matches located in deformed triangles: the matches by triangle and barycentric
                    // coordinates, the unknown the total displacement of mesh1 from its FIXED gear (fb_pairs_relax_bary)
                    std::vector<size_t> rows((size_t)K);
                    std::iota(rows.begin(), rows.end(), (size_t)0);
                    if ((rc = relax_general(ctx, m, table, rows, false, rw, x, flags))) return rc;
                }
                for (int64_t k = 0; k < K; ++k)
                    if (table.rl[k]) table.wt[k] = table.wt[k] * rw[k];                 // Link.weight (optimizer.py:313-317)
            }
        }
        clk.lap(5);
        if (had_prev && prev.size()) {
            // a pair without a confident block in this round keeps the links of its last good round (the reference
            // breaks out of the loop before clear_links, matcher.py:671-679)
            std::fill(in_cur.begin(), in_cur.end(), 0);
            for (int32_t p : table.pid) in_cur[p] = 1;
            for (size_t k = 0; k < prev.size(); ++k)
                if (!in_cur[prev.pid[k]]) { table.push_from(prev, k); last_links = false; }
        }
    }
    // ---- the table in the INITIAL gears (matcher.py:748-751)
    std::vector<char> ok_pair((size_t)n);
    for (int p = 0; p < n; ++p) ok_pair[p] = active[p] && has_last[p] && !flags[p];
    if (m->residue_mode == 1) {
        // threshold mode: matches cut by the residue filter are masked out of the link (optimizer.py:399-402)
        std::vector<char> any((size_t)n, 0);
        for (size_t k = 0; k < table.size(); ++k) if (ok_pair[table.pid[k]] && table.wt[k] > 0) any[table.pid[k]] = 1;
        for (int p = 0; p < n; ++p) ok_pair[p] = ok_pair[p] && any[p];
    }
    m->r_pid.clear(); m->r_xy0.clear(); m->r_xy1.clear(); m->r_w.clear();
    for (size_t k = 0; k < table.size(); ++k) {
        const int p = table.pid[k];
        if (!ok_pair[p] || (m->residue_mode == 1 && !(table.wt[k] > 0))) { last_links = false; continue; }
        m->r_pid.push_back(p);
        m->r_xy0.push_back(table.xy0[2 * k] - tx[p]); m->r_xy0.push_back(table.xy0[2 * k + 1] - ty[p]);
        m->r_xy1.push_back(table.xy1i[2 * k]); m->r_xy1.push_back(table.xy1i[2 * k + 1]);
        m->r_w.push_back(table.wt[k]);
    }
    for (int p = 0; p < n; ++p) { valid[p] = ok_pair[p]; strain[p] = kDefaultAvgDeform; }
    // ---- strain (matcher.py:752-777)
    const size_t K = m->r_pid.size();
    if (m->compute_strain && K) {
        if ((rc = ensure_system(ctx, m))) return rc;
        // rows of a pair are contiguous by construction; a caller-visible guarantee, so check it
        std::vector<char> seen((size_t)n, 0);
        bool contiguous = true;
        for (size_t k = 0; k < K; ++k) {
            if (k && m->r_pid[k] == m->r_pid[k - 1]) continue;
            if (seen[m->r_pid[k]]) contiguous = false;
            seen[m->r_pid[k]] = 1;
        }
        if (!contiguous) {
            std::vector<size_t> o(K);
            std::iota(o.begin(), o.end(), (size_t)0);
            std::stable_sort(o.begin(), o.end(), [&](size_t a, size_t b) { return m->r_pid[a] < m->r_pid[b]; });
            std::vector<int32_t> pid2(K); std::vector<double> a0(2 * K), a1(2 * K); std::vector<float> w2(K);
            for (size_t k = 0; k < K; ++k) {
                pid2[k] = m->r_pid[o[k]]; w2[k] = m->r_w[o[k]];
                a0[2 * k] = m->r_xy0[2 * o[k]]; a0[2 * k + 1] = m->r_xy0[2 * o[k] + 1]; a1[2 * k] = m->r_xy1[2 * o[k]]; a1[2 * k + 1] = m->r_xy1[2 * o[k] + 1];
            }
            m->r_pid.swap(pid2); m->r_xy0.swap(a0); m->r_xy1.swap(a1); m->r_w.swap(w2);
            last_links = false;
        }
        std::vector<double> p0(2 * K), R;
        std::vector<char> bad;
        for (size_t k = 0; k < K; ++k) { p0[2 * k] = m->r_xy0[2 * k] + tx[m->r_pid[k]]; p0[2 * k + 1] = m->r_xy0[2 * k + 1] + ty[m->r_pid[k]]; }   // mesh0 points, FIXED gear
        rigid_fits(n, m->r_pid, p0, m->r_xy1, m->r_w, R, bad);
        clk.lap(6);
        if (!m->ragged) {
            rc = fb_pairs_strain(ctx, m->sys, n, m->gnx, m->gny, m->gxs.data(), m->gys.data(), (int64_t)K, m->r_pid.data(), p0.data(), m->r_xy1.data(),
                                 m->r_w.data(), R.data(), m->stiffness_lambda, m->es0[0], last_links ? 1 : 0, kDefaultAvgDeform, strain, &m->strain_iters,
                                 &m->strain_relres);
        } else {
            std::vector<int32_t> nodes3;
            std::vector<double> B1;
            locate_grid(m, K, m->r_pid.data(), m->r_xy1.data(), nodes3, B1);
            rc = fb_pairs_strain_bary(ctx, m->sys, n, (int64_t)K, m->r_pid.data(), nodes3.data(), B1.data(), p0.data(), m->r_xy1.data(), m->r_w.data(), R.data(),
                                      m->stiffness_lambda, m->es0.data(), kDefaultAvgDeform, strain, &m->strain_iters, &m->strain_relres);
        }
        if (rc) return rc;
        m->strain_matches = (int64_t)K;
        clk.lap(7);
        bool any_bad = false;
        for (int p = 0; p < n; ++p)
            if (bad[p]) { flags[p] |= FB_STRIP_RIGIDFIT; valid[p] = 0; strain[p] = kDefaultAvgDeform; any_bad = true; }
        if (any_bad) {
            size_t w = 0;
            for (size_t k = 0; k < K; ++k) {
                if (bad[m->r_pid[k]]) continue;
                m->r_pid[w] = m->r_pid[k]; m->r_w[w] = m->r_w[k];
                m->r_xy0[2 * w] = m->r_xy0[2 * k]; m->r_xy0[2 * w + 1] = m->r_xy0[2 * k + 1];
                m->r_xy1[2 * w] = m->r_xy1[2 * k]; m->r_xy1[2 * w + 1] = m->r_xy1[2 * k + 1];
                ++w;
            }
            m->r_pid.resize(w); m->r_w.resize(w); m->r_xy0.resize(2 * w); m->r_xy1.resize(2 * w);
        }
    }
    *nrows = (int64_t)m->r_pid.size();
    return FB_OK;
}

int fb_strip_matcher_set_extras(fb_ctx* ctx, fb_strip_matcher* m, const uint8_t* const* masks0, const uint8_t* const* masks1, int photometric) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, m != nullptr);
    if (masks0) m->mask0.assign(masks0, masks0 + m->P); else m->mask0.clear();
    if (masks1) m->mask1.assign(masks1, masks1 + m->P); else m->mask1.clear();
    m->want_phtm = photometric != 0;
    return FB_OK;
}

int fb_match_strips_photometric(fb_ctx* ctx, fb_strip_matcher* m, double* phtm, uint8_t* has) {
    FB_CHECK_ARG(ctx, m && phtm && has);
    if (m->phtm_has.size() != (size_t)m->P) return fb_fail(ctx, FB_ERR_ARG, "fb_match_strips_photometric: the last call did not ask for the statistics");
    std::copy(m->phtm.begin(), m->phtm.end(), phtm);
    std::copy(m->phtm_has.begin(), m->phtm_has.end(), has);
    return FB_OK;
}

int fb_match_strips_deformed(fb_ctx* ctx, fb_strip_matcher* m, uint8_t* deformed, int32_t* ntiers, int* nodes) {
    FB_CHECK_ARG(ctx, m != nullptr);
    const bool have = m->is_def.size() == (size_t)m->P;
    for (int p = 0; p < m->P; ++p) {
        if (deformed) deformed[p] = have ? m->is_def[p] : 0;
        if (ntiers) ntiers[p] = have ? (int32_t)m->tiers[p].size() : 0;
    }
    if (nodes) *nodes = m->gnx * m->gny;
    return FB_OK;
}

int fb_match_strips_field(fb_ctx* ctx, fb_strip_matcher* m, double* field, int32_t* tiers) {
    FB_CHECK_ARG(ctx, m != nullptr);
    const size_t V = (size_t)m->gnx * m->gny;
    const bool have = m->is_def.size() == (size_t)m->P;
    for (int p = 0; p < m->P; ++p) {
        if (field) {
            if (have && m->is_def[p] && m->U.size() == 2 * V * m->P) std::copy(m->U.begin() + 2 * V * p, m->U.begin() + 2 * V * (p + 1), field + 2 * V * p);
            else std::fill(field + 2 * V * p, field + 2 * V * (p + 1), 0.0);
        }
        if (tiers && have) tiers = std::copy(m->tiers[p].begin(), m->tiers[p].end(), tiers);
    }
    return FB_OK;
}

int fb_match_strips_table(fb_ctx* ctx, fb_strip_matcher* m, int32_t* pair, double* xy0, double* xy1, float* weight) {
    FB_CHECK_ARG(ctx, m != nullptr);
    const size_t K = m->r_pid.size();
    if (K == 0) return FB_OK;
    FB_CHECK_ARG(ctx, pair && xy0 && xy1 && weight);
    std::copy(m->r_pid.begin(), m->r_pid.end(), pair);
    std::copy(m->r_xy0.begin(), m->r_xy0.end(), xy0);
    std::copy(m->r_xy1.begin(), m->r_xy1.end(), xy1);
    std::copy(m->r_w.begin(), m->r_w.end(), weight);
    return FB_OK;
}

}  // extern "C"
