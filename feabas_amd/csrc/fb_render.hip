// MeshRenderer for the block matcher of general (triangulated) meshes: the piecewise-linear MOVING -> image field of
// MeshRenderer.from_mesh (feabas/renderer.py:47-166), crop_field (453-563) and the sampling of crop_multiple (601-631)
// -> common.render_by_subregions -> cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) (common.py:218-350), for one region
// without collisions.  The image, the vertices and every intermediate stay in HBM:
//   mesh_cand_kernel   one workgroup per block: the triangles whose box meets the block's box (bbox - 0.5, the query box
//                      of renderer.py:405) -> a capped list per block;
//   mesh_field_kernel  one workgroup per 16 x 16 pixel tile of a block: candidates of the tile staged in LDS with their
//                      barycentric / image coefficients, every pixel keeps the hit of the smallest triangle index
//                      (deterministic for pixels on shared edges).  PASS 0 reduces floor(min) / ceil(max) of the field
//                      (the remap origin of common.py:316-321), PASS 1 samples.
#include <algorithm>
#include <climits>

#include "fb_common.h"

namespace {

constexpr int TILE = 16;             // 256 threads = one pixel each
constexpr int CHUNK = 128;           // candidates staged per sweep
constexpr int REC = 14;              // doubles per staged candidate
constexpr double BARY_EPS = 1e-9;
constexpr int MX_DIS = 16300;        // common.py:264: extent one remap call may cover

// boxes of all triangles, rounded outwards to float32: one coalesced 16-byte read per triangle in the scan below instead of
// three vertex gathers
__global__ void tri_box_kernel(int T, const double* __restrict__ vm, const int* __restrict__ tris, float4* __restrict__ box) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        const int i0 = tris[3 * t], i1 = tris[3 * t + 1], i2 = tris[3 * t + 2];
        const double x0 = vm[2 * i0], y0 = vm[2 * i0 + 1], x1 = vm[2 * i1], y1 = vm[2 * i1 + 1], x2 = vm[2 * i2], y2 = vm[2 * i2 + 1];
        box[t] = make_float4(__double2float_rd(fmin(x0, fmin(x1, x2))), __double2float_rd(fmin(y0, fmin(y1, y2))),
                             __double2float_ru(fmax(x0, fmax(x1, x2))), __double2float_ru(fmax(y0, fmax(y1, y2))));
    }
}

// the list is a superset of the triangles that meet the query box (outward rounding): its users test exactly
__global__ void mesh_cand_kernel(int T, const float4* __restrict__ box, const double* __restrict__ org,
                                 int h, int w, int cap, int* __restrict__ cand, int* __restrict__ count) {
    const int b = blockIdx.x;
    __shared__ int n_hit;
    if (threadIdx.x == 0) n_hit = 0;
    __syncthreads();
    const double qx0 = org[2 * b] - 0.5, qy0 = org[2 * b + 1] - 0.5;
    const float bx0 = __double2float_rd(qx0), by0 = __double2float_rd(qy0), bx1 = __double2float_ru(qx0 + (double)w), by1 = __double2float_ru(qy0 + (double)h);
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const float4 bb = box[t];
        if (bb.z < bx0 || bb.x > bx1 || bb.w < by0 || bb.y > by1) continue;
        const int k = atomicAdd(&n_hit, 1);
        if (k < cap) cand[(size_t)b * cap + k] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) count[b] = n_hit;
}

// Mesh.tri_finder (mesh.py:2080-2188) for K points: every lane owns a point and walks the triangle boxes (a wave reads the
// same box: broadcast loads), the exact barycentric test runs on box hits only; the smallest containing triangle index wins
// (deterministic on shared edges).  blockIdx.y splits the triangle range, merged with atomicMin.
__global__ void mesh_locate_kernel(int K, const double* __restrict__ pts, int T, int chunk, const float4* __restrict__ box,
                                   const double* __restrict__ vm, const int* __restrict__ tris, int* __restrict__ tid_out) {
#pragma clang fp contract(off)
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = k < K;
    const double px = valid ? pts[2 * k] : 0.0, py = valid ? pts[2 * k + 1] : 0.0;
    const float xd = __double2float_rd(px), xu = __double2float_ru(px), yd = __double2float_rd(py), yu = __double2float_ru(py);
    const int t0 = blockIdx.y * chunk, t1 = min(T, t0 + chunk);
    int best = INT_MAX;
    for (int t = t0; t < t1; ++t) {
        const float4 bb = box[t];
        if (!valid || best != INT_MAX || bb.x > xu || bb.z < xd || bb.y > yu || bb.w < yd) continue;
        const int i0 = tris[3 * t], i1 = tris[3 * t + 1], i2 = tris[3 * t + 2];
        const double x0 = vm[2 * i0], y0 = vm[2 * i0 + 1];
        const double e1x = vm[2 * i1] - x0, e1y = vm[2 * i1 + 1] - y0, e2x = vm[2 * i2] - x0, e2y = vm[2 * i2 + 1] - y0;
        const double d = e1x * e2y - e1y * e2x;
        if (d == 0.0) continue;
        const double dx = px - x0, dy = py - y0;
        const double l1 = (dx * e2y - dy * e2x) / d, l2 = (e1x * dy - e1y * dx) / d, l0 = (1.0 - l1) - l2;
        if (l0 >= -BARY_EPS && l1 >= -BARY_EPS && l2 >= -BARY_EPS) best = t;
    }
    if (best != INT_MAX) atomicMin(&tid_out[k], best);
}

// The same search with the triangles culled per workgroup: the 256 points of a workgroup (neighbours in a raster or in a
// z-ordered / sorted list) span a small box; the triangle boxes are tested against THAT box cooperatively, 1024 at a time, the
// survivors go to a list in LDS and every point tests only those exactly -- for a raster of 219 k points over 13 k triangles
// ~100 survivors per workgroup instead of 13 k box tests per point (2.5 ms -> 0.1 ms).  The smallest containing index wins, as before.
__global__ __launch_bounds__(256) void mesh_locate_cull_kernel(int K, const double* __restrict__ pts, int T, const float4* __restrict__ box,
                                                                const double* __restrict__ vm, const int* __restrict__ tris, int* __restrict__ tid_out) {
#pragma clang fp contract(off)
    __shared__ float s_red[4][4];
    __shared__ int s_hits[1024];
    __shared__ int s_n;
    const int k = blockIdx.x * 256 + threadIdx.x;
    const bool valid = k < K;
    const double px = valid ? pts[2 * k] : 0.0, py = valid ? pts[2 * k + 1] : 0.0;
    const float xd = __double2float_rd(px), xu = __double2float_ru(px), yd = __double2float_rd(py), yu = __double2float_ru(py);
    float bx0 = valid ? xd : INFINITY, by0 = valid ? yd : INFINITY, bx1 = valid ? xu : -INFINITY, by1 = valid ? yu : -INFINITY;
    for (int off = 32; off > 0; off >>= 1) {
        bx0 = fminf(bx0, __shfl_down(bx0, off)); by0 = fminf(by0, __shfl_down(by0, off));
        bx1 = fmaxf(bx1, __shfl_down(bx1, off)); by1 = fmaxf(by1, __shfl_down(by1, off));
    }
    if ((threadIdx.x & 63) == 0) { float* r = s_red[threadIdx.x >> 6]; r[0] = bx0; r[1] = by0; r[2] = bx1; r[3] = by1; }
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    bx0 = fminf(fminf(s_red[0][0], s_red[1][0]), fminf(s_red[2][0], s_red[3][0])); by0 = fminf(fminf(s_red[0][1], s_red[1][1]), fminf(s_red[2][1], s_red[3][1]));
    bx1 = fmaxf(fmaxf(s_red[0][2], s_red[1][2]), fmaxf(s_red[2][2], s_red[3][2])); by1 = fmaxf(fmaxf(s_red[0][3], s_red[1][3]), fmaxf(s_red[2][3], s_red[3][3]));
    int best = INT_MAX;
    for (int t0 = 0; t0 < T; t0 += 1024) {
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * 256 + threadIdx.x;
            if (t < T) {
                const float4 bb = box[t];
                if (!(bb.x > bx1 || bb.z < bx0 || bb.y > by1 || bb.w < by0)) s_hits[atomicAdd(&s_n, 1)] = t;
            }
        }
        __syncthreads();
        const int n = s_n;
        for (int h = 0; h < n; ++h) {
            const int t = s_hits[h];
            if (!valid || t > best) continue;
            const float4 bb = box[t];
            if (bb.x > xu || bb.z < xd || bb.y > yu || bb.w < yd) continue;
            const int i0 = tris[3 * t], i1 = tris[3 * t + 1], i2 = tris[3 * t + 2];
            const double x0 = vm[2 * i0], y0 = vm[2 * i0 + 1];
            const double e1x = vm[2 * i1] - x0, e1y = vm[2 * i1 + 1] - y0, e2x = vm[2 * i2] - x0, e2y = vm[2 * i2 + 1] - y0;
            const double d = e1x * e2y - e1y * e2x;
            if (d == 0.0) continue;
            const double dx = px - x0, dy = py - y0;
            const double l1 = (dx * e2y - dy * e2x) / d, l2 = (e1x * dy - e1y * dx) / d, l0 = (1.0 - l1) - l2;
            if (l0 >= -BARY_EPS && l1 >= -BARY_EPS && l2 >= -BARY_EPS) best = t;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
    }
    if (valid) tid_out[k] = best == INT_MAX ? -1 : best;
}

__global__ void locate_finish_kernel(int K, int* tid) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x)
        if (tid[k] == INT_MAX) tid[k] = -1;
}

__global__ void fill_int_kernel(int n, int* p, int v) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

struct RenderArgs {
    const void* img; int dtype, IH, IW, img_x0, img_y0;
    const double* vm; const double* vi; const int* tris;
    const double* org; int h, w;
    const int* tier; const double* A6;
    int cap; const int* cand; const int* count;
    int* ext;            // [NB][4]: floor(min x), floor(min y), ceil(max x), ceil(max y) of the masked field
    const int* origin;   // [NB][2]
    float* out; uint8_t* mask;
    int b0;              // first block of this launch (grid y is limited to 65535 blocks)
    double2* field;      // [NB][h][w] scratch (nullable): pass 0 keeps the field (and the mask) it found, pass 1 reads them back
                         // instead of searching the triangles of every pixel a second time
};

// cv2.remap, CV_8U bilinear: fixed-point table of 1/32-px phases scaled by 2^15 (BilinearTab_i), saturate_cast<short>
// turns the unit weight 32768 into 32767 and the table's sum fix-up adds the missing 1 to the last tap
__device__ __forceinline__ float sample_u8(const uint8_t* __restrict__ img, int IH, int IW, int ix, int iy, int a, int b) {
    int w00 = (32 - a) * (32 - b) * 32, w01 = a * (32 - b) * 32, w10 = (32 - a) * b * 32, w11 = a * b * 32;
    if ((a | b) == 0) { w00 = 32767; w11 = 1; }
    const bool x0ok = ix >= 0 && ix < IW, x1ok = ix + 1 >= 0 && ix + 1 < IW, y0ok = iy >= 0 && iy < IH, y1ok = iy + 1 >= 0 && iy + 1 < IH;
    const int cx0 = min(max(ix, 0), IW - 1), cx1 = min(max(ix + 1, 0), IW - 1), cy0 = min(max(iy, 0), IH - 1), cy1 = min(max(iy + 1, 0), IH - 1);
    const int v00 = (y0ok && x0ok) ? img[(size_t)cy0 * IW + cx0] : 0, v01 = (y0ok && x1ok) ? img[(size_t)cy0 * IW + cx1] : 0;
    const int v10 = (y1ok && x0ok) ? img[(size_t)cy1 * IW + cx0] : 0, v11 = (y1ok && x1ok) ? img[(size_t)cy1 * IW + cx1] : 0;
    const int s = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
    return (float)min(max(s, 0), 255);
}

__device__ __forceinline__ float sample_f32(const float* __restrict__ img, int IH, int IW, int ix, int iy, int a, int b) {
#pragma clang fp contract(off)
    const float ax = (float)a * (1.0f / 32.0f), ay = (float)b * (1.0f / 32.0f);
    const float w00 = (1.0f - ay) * (1.0f - ax), w01 = (1.0f - ay) * ax, w10 = ay * (1.0f - ax), w11 = ay * ax;
    const bool x0ok = ix >= 0 && ix < IW, x1ok = ix + 1 >= 0 && ix + 1 < IW, y0ok = iy >= 0 && iy < IH, y1ok = iy + 1 >= 0 && iy + 1 < IH;
    const int cx0 = min(max(ix, 0), IW - 1), cx1 = min(max(ix + 1, 0), IW - 1), cy0 = min(max(iy, 0), IH - 1), cy1 = min(max(iy + 1, 0), IH - 1);
    const float v00 = img[(size_t)cy0 * IW + cx0], v01 = img[(size_t)cy0 * IW + cx1];
    const float v10 = img[(size_t)cy1 * IW + cx0], v11 = img[(size_t)cy1 * IW + cx1];
    return ((((y0ok && x0ok) ? v00 : 0.f) * w00 + ((y0ok && x1ok) ? v01 : 0.f) * w01) + ((y1ok && x0ok) ? v10 : 0.f) * w10) + ((y1ok && x1ok) ? v11 : 0.f) * w11;
}

template <int PASS>
__global__ __launch_bounds__(TILE* TILE) void mesh_field_kernel(const RenderArgs g) {
#pragma clang fp contract(off)
    __shared__ double rec[CHUNK * REC];
    __shared__ int rec_tid[CHUNK];
    __shared__ int n_rec;
    __shared__ int red[4];
    const int b = g.b0 + blockIdx.y;
    const int tiles_x = (g.w + TILE - 1) / TILE;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int tid = threadIdx.x;
    const int px = tx * TILE + (tid & (TILE - 1)), py = ty * TILE + (tid / TILE);
    const bool live = px < g.w && py < g.h;
    const double ox = g.org[2 * b], oy = g.org[2 * b + 1];
    const double xx = ox + (double)px, yy = oy + (double)py;
    double fx = 0.0, fy = 0.0;
    bool ok = false;
    // tier 1 / 2: affine field, every pixel valid; 11 / 12: affine field, valid where the pixel lies inside the mesh (the
    // precise mask of crop_field_affine, renderer.py:437-447, for a block that sticks out of the mesh); 3: exact field
    const int tr = g.tier[b];
    const bool exact = tr == 3;
    if (!exact) {
        const double* A = g.A6 + 6 * (size_t)b;          // {A00, A10, t0, A01, A11, t1}
        fx = (xx * A[0] + yy * A[1]) + A[2];
        fy = (xx * A[3] + yy * A[4]) + A[5];
        ok = live && tr < 10;
    }
    const size_t o_px = ((size_t)b * g.h + py) * g.w + px;
    if (PASS == 1 && g.field) {
        if (live) { const double2 f = g.field[o_px]; fx = f.x; fy = f.y; ok = g.mask[o_px] != 0; }
    } else if (exact || tr >= 10) {
        // tile box in MOVING coordinates (pixel centres), grown by the inside tolerance
        const double tx0 = ox + (double)(tx * TILE) - 1e-6, ty0 = oy + (double)(ty * TILE) - 1e-6;
        const double tx1 = ox + (double)min(tx * TILE + TILE - 1, g.w - 1) + 1e-6, ty1 = oy + (double)min(ty * TILE + TILE - 1, g.h - 1) + 1e-6;
        const int nc = min(g.count[b], g.cap);
        const int* cl = g.cand + (size_t)b * g.cap;
        int best = INT_MAX;
        int base = 0;
        while (base < nc) {                              // uniform over the workgroup
            if (tid == 0) n_rec = 0;
            __syncthreads();
            // stage the candidates of this sweep that meet the tile: CHUNK looked at => at most CHUNK staged
            const int look = min(nc - base, CHUNK);
            if (tid < look) {
                const int t = cl[base + tid];
                const int i0 = g.tris[3 * t], i1 = g.tris[3 * t + 1], i2 = g.tris[3 * t + 2];
                const double x0 = g.vm[2 * i0], y0 = g.vm[2 * i0 + 1], x1 = g.vm[2 * i1], y1 = g.vm[2 * i1 + 1], x2 = g.vm[2 * i2], y2 = g.vm[2 * i2 + 1];
                const double lx = fmin(x0, fmin(x1, x2)), hx = fmax(x0, fmax(x1, x2));
                const double ly = fmin(y0, fmin(y1, y2)), hy = fmax(y0, fmax(y1, y2));
                const double d = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
                if (!(hx < tx0 || lx > tx1 || hy < ty0 || ly > ty1) && d != 0.0) {
                    const int k = atomicAdd(&n_rec, 1);
                    double* r = rec + k * REC;
                    r[0] = x0; r[1] = y0; r[2] = x1 - x0; r[3] = y1 - y0; r[4] = x2 - x0; r[5] = y2 - y0; r[6] = 1.0 / d;
                    r[7] = g.vi[2 * i0]; r[8] = g.vi[2 * i0 + 1]; r[9] = g.vi[2 * i1]; r[10] = g.vi[2 * i1 + 1];
                    r[11] = g.vi[2 * i2]; r[12] = g.vi[2 * i2 + 1];
                    rec_tid[k] = t;
                }
            }
            __syncthreads();
            base += look;
            const int ns = n_rec;
            if (live) {
                for (int k = 0; k < ns; ++k) {
                    const double* r = rec + k * REC;
                    const double dx = xx - r[0], dy = yy - r[1];
                    const double l1 = (dx * r[5] - dy * r[4]) * r[6];
                    const double l2 = (r[2] * dy - r[3] * dx) * r[6];
                    const double l0 = (1.0 - l1) - l2;
                    if (l0 >= -BARY_EPS && l1 >= -BARY_EPS && l2 >= -BARY_EPS && rec_tid[k] < best) {
                        best = rec_tid[k];
                        if (exact) {
                            fx = (l0 * r[7] + l1 * r[9]) + l2 * r[11];
                            fy = (l0 * r[8] + l1 * r[10]) + l2 * r[12];
                        }
                        ok = true;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (PASS == 0) {
        if (g.field && live) { g.field[o_px] = make_double2(fx, fy); g.mask[o_px] = ok ? 1 : 0; }
        if (tid < 4) red[tid] = (tid < 2) ? INT_MAX : INT_MIN;
        __syncthreads();
        // extent of the tile: reduced inside every wave first (256 pixels hitting four LDS words with atomics serialised the
        // whole pass: 1.65 ms per 40 M pixels against 0.3 ms for the triangle search itself)
        int lo_x = ok ? (int)floor(fx) : INT_MAX, lo_y = ok ? (int)floor(fy) : INT_MAX;
        int hi_x = ok ? (int)ceil(fx) : INT_MIN, hi_y = ok ? (int)ceil(fy) : INT_MIN;
        for (int off = 32; off > 0; off >>= 1) {
            lo_x = min(lo_x, __shfl_xor(lo_x, off)); lo_y = min(lo_y, __shfl_xor(lo_y, off));
            hi_x = max(hi_x, __shfl_xor(hi_x, off)); hi_y = max(hi_y, __shfl_xor(hi_y, off));
        }
        if ((tid & 63) == 0 && lo_x != INT_MAX) {
            atomicMin(&red[0], lo_x); atomicMin(&red[1], lo_y);
            atomicMax(&red[2], hi_x); atomicMax(&red[3], hi_y);
        }
        __syncthreads();
        if (tid < 2 && red[tid] != INT_MAX) atomicMin(&g.ext[4 * b + tid], red[tid]);
        else if (tid >= 2 && tid < 4 && red[tid] != INT_MIN) atomicMax(&g.ext[4 * b + tid], red[tid]);
        return;
    }
    if (!live) return;
    const size_t o = ((size_t)b * g.h + py) * g.w + px;
    g.mask[o] = ok ? 1 : 0;
    if (!ok) { g.out[o] = 0.f; return; }
    const int orx = g.origin[2 * b], ory = g.origin[2 * b + 1];
    const float mx = (float)(fx - (double)orx), my = (float)(fy - (double)ory);
    const int sx = (int)rintf(mx * 32.0f), sy = (int)rintf(my * 32.0f);
    const int ix = (sx >> 5) + orx - g.img_x0, iy = (sy >> 5) + ory - g.img_y0;
    g.out[o] = g.dtype == FB_U8 ? sample_u8((const uint8_t*)g.img, g.IH, g.IW, ix, iy, sx & 31, sy & 31)
                                : sample_f32((const float*)g.img, g.IH, g.IW, ix, iy, sx & 31, sy & 31);
}

// origin of every block: one for the whole stack while the field spans less than MX_DIS (render_by_subregions then makes
// ONE remap call, common.py:305-321), else the block's own floor(min) - 4
__global__ void mesh_origin_kernel(int NB, const int* __restrict__ ext, int* __restrict__ origin) {
    __shared__ int red[4];
    if (threadIdx.x < 4) red[threadIdx.x] = (threadIdx.x < 2) ? INT_MAX : INT_MIN;
    __syncthreads();
    for (int b = threadIdx.x; b < NB; b += blockDim.x) {
        if (ext[4 * b] == INT_MAX) continue;
        atomicMin(&red[0], ext[4 * b]); atomicMin(&red[1], ext[4 * b + 1]);
        atomicMax(&red[2], ext[4 * b + 2]); atomicMax(&red[3], ext[4 * b + 3]);
    }
    __syncthreads();
    const bool one = red[0] != INT_MAX && (red[2] - red[0]) < MX_DIS && (red[3] - red[1]) < MX_DIS;
    for (int b = threadIdx.x; b < NB; b += blockDim.x) {
        const bool any = ext[4 * b] != INT_MAX;
        origin[2 * b] = any ? (one ? red[0] : ext[4 * b]) - 4 : 0;
        origin[2 * b + 1] = any ? (one ? red[1] : ext[4 * b + 1]) - 4 : 0;
    }
}

__global__ void fill_ext_kernel(int NB, int* ext) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 4 * NB; i += gridDim.x * blockDim.x) ext[i] = ((i & 3) < 2) ? INT_MAX : INT_MIN;
}

}  // namespace


// fb_mesh_block_uncovered on the device: one thread per block clips its candidate triangles to the block's box
// (Sutherland-Hodgman against the four sides) and sums the areas, in the order of the candidate list like the host loop
__global__ __launch_bounds__(128) void block_uncovered_kernel(const double* __restrict__ v_mov, const int* __restrict__ tris, int NB,
                                                               const double* __restrict__ org, int h, int w, int cap, const int* __restrict__ cand,
                                                               const int* __restrict__ count, double* __restrict__ uncovered) {
    const int b = blockIdx.x * 128 + threadIdx.x;
    if (b >= NB) return;
    const double bx0 = org[2 * b] - 0.5, by0 = org[2 * b + 1] - 0.5, bx1 = bx0 + (double)w, by1 = by0 + (double)h;
    double covered = 0.0;
    const int nc = min(count[b], cap);
    for (int k = 0; k < nc; ++k) {
        const int* t3 = tris + 3 * (size_t)cand[(size_t)b * cap + k];
        double px[8], py[8], qx[8], qy[8];                    // a triangle cut by four half planes has at most 7 corners
        int n = 3;
        for (int a = 0; a < 3; ++a) { px[a] = v_mov[2 * (size_t)t3[a]]; py[a] = v_mov[2 * (size_t)t3[a] + 1]; }
        for (int side = 0; side < 4 && n > 0; ++side) {
            int m = 0;
            for (int i = 0; i < n; ++i) {
                const int j = (i + 1 == n) ? 0 : i + 1;
                const double ax = px[i], ay = py[i], cx = px[j], cy = py[j];
                const double da = side == 0 ? ax - bx0 : side == 1 ? bx1 - ax : side == 2 ? ay - by0 : by1 - ay;
                const double dc = side == 0 ? cx - bx0 : side == 1 ? bx1 - cx : side == 2 ? cy - by0 : by1 - cy;
                if (da >= 0 && m < 8) { qx[m] = ax; qy[m] = ay; ++m; }
                if ((da >= 0) != (dc >= 0) && m < 8) { const double s_ = da / (da - dc); qx[m] = ax + s_ * (cx - ax); qy[m] = ay + s_ * (cy - ay); ++m; }
            }
            n = m;
            for (int i = 0; i < n; ++i) { px[i] = qx[i]; py[i] = qy[i]; }
        }
        double a2 = 0.0;
        for (int i = 0; i < n; ++i) { const int j = (i + 1 == n) ? 0 : i + 1; a2 += px[i] * py[j] - px[j] * py[i]; }
        covered += 0.5 * fabs(a2);
    }
    uncovered[b] = (double)w * (double)h - covered;
}

extern "C" {

int fb_mesh_candidates_dev(fb_ctx* ctx, int T, const double* v_mov, const int* tris, int NB, const double* org, int h, int w, int cap,
                           int* cand, int* count) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, T >= 0 && NB >= 0 && h > 0 && w > 0 && cap > 0);
    if (NB == 0) return FB_OK;
    FB_CHECK_ARG(ctx, v_mov && tris && org && cand && count);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    if (T == 0) { FB_HIP(ctx, hipMemsetAsync(count, 0, sizeof(int) * NB, ctx->stream)); return FB_OK; }
    float4* box = nullptr;
    { const int rc = fb_malloc(ctx, sizeof(float4) * (size_t)T, (void**)&box); if (rc) return rc; }      // (the context's allocation cache)
    {
        FB_PROF_B(ctx, "mesh_cand", (double)T * 16.0 * NB);
        hipLaunchKernelGGL(tri_box_kernel, dim3(std::min(fb_cdiv(T, 256), 4096)), dim3(256), 0, ctx->stream, T, v_mov, tris, box);
        hipLaunchKernelGGL(mesh_cand_kernel, dim3(NB), dim3(256), 0, ctx->stream, T, box, org, h, w, cap, cand, count);
    }
    const hipError_t e = hipGetLastError();
    fb_free(ctx, box);                                       // waits for the stream
    FB_HIP(ctx, e);
    return FB_OK;
}

int fb_mesh_block_uncovered_dev(fb_ctx* ctx, const double* v_mov, const int* tris, int NB, const double* org, int h, int w, int cap, const int* cand,
                                const int* count, double* uncovered) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, NB >= 0 && h > 0 && w > 0 && cap > 0);
    if (NB == 0) return FB_OK;
    FB_CHECK_ARG(ctx, v_mov && tris && org && cand && count && uncovered);
    FB_PROF(ctx, "mesh_block_uncovered");
    hipLaunchKernelGGL(block_uncovered_kernel, dim3(fb_cdiv(NB, 128)), dim3(128), 0, ctx->stream, v_mov, tris, NB, org, h, w, cap, cand, count, uncovered);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

int fb_mesh_locate_dev(fb_ctx* ctx, int T, const double* v_mov, const int* tris, int K, const double* pts, int* tid) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, T >= 0 && K >= 0);
    if (K == 0) return FB_OK;
    FB_CHECK_ARG(ctx, pts && tid && (T == 0 || (v_mov && tris)));
    FB_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(fill_int_kernel, dim3(fb_cdiv(K, 256)), dim3(256), 0, ctx->stream, K, tid, T == 0 ? -1 : INT_MAX);
    if (T == 0) { FB_HIP(ctx, hipGetLastError()); return FB_OK; }
    float4* box = nullptr;
    { const int rc = fb_malloc(ctx, sizeof(float4) * (size_t)T, (void**)&box); if (rc) return rc; }      // (tri_finder is called a dozen times per section pair)
    {
        FB_PROF(ctx, "mesh_locate");
        hipLaunchKernelGGL(tri_box_kernel, dim3(std::min(fb_cdiv(T, 256), 4096)), dim3(256), 0, ctx->stream, T, v_mov, tris, box);
        const int gx = fb_cdiv(K, 256);
        static const bool plain = getenv("FEABAS_HIP_LOCATE_PLAIN") != nullptr;
        if (gx >= 8 && !plain) {
            // more than a few workgroups of points: triangles culled per workgroup against the box of its 256 points
            hipLaunchKernelGGL(mesh_locate_cull_kernel, dim3(gx), dim3(256), 0, ctx->stream, K, pts, T, box, v_mov, tris, tid);
        } else {
            // few points: every point walks the triangle boxes; the triangle range is split over blockIdx.y to fill the chip
            int gy = std::max(1, std::min(fb_cdiv(T, 2048), fb_cdiv(2048, gx)));
            const int chunk = fb_cdiv(T, gy);
            gy = fb_cdiv(T, chunk);
            hipLaunchKernelGGL(mesh_locate_kernel, dim3(gx, gy), dim3(256), 0, ctx->stream, K, pts, T, chunk, box, v_mov, tris, tid);
            hipLaunchKernelGGL(locate_finish_kernel, dim3(std::min(fb_cdiv(K, 256), 1024)), dim3(256), 0, ctx->stream, K, tid);
        }
    }
    const hipError_t e = hipGetLastError();
    fb_free(ctx, box);                                       // waits for the stream
    FB_HIP(ctx, e);
    return FB_OK;
}

int fb_mesh_render_blocks_dev(fb_ctx* ctx, const void* img, int dtype, int IH, int IW, int img_x0, int img_y0, const double* v_mov,
                              const double* v_img, const int* tris, int NB, const double* org, int h, int w, const int* tier,
                              const double* A6, int cap, const int* cand, const int* count, int* ext, int* origin, float* out,
                              uint8_t* mask) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, NB >= 0 && h > 0 && w > 0 && IH > 0 && IW > 0 && cap > 0);
    FB_CHECK_ARG(ctx, dtype == FB_U8 || dtype == FB_F32);
    if (NB == 0) return FB_OK;
    FB_CHECK_ARG(ctx, img && v_mov && v_img && tris && org && tier && A6 && cand && count && ext && origin && out && mask);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    RenderArgs g{img, dtype, IH, IW, img_x0, img_y0, v_mov, v_img, tris, org, h, w, tier, A6, cap, cand, count, ext, origin, out, mask, 0, nullptr};
    // the field of pass 0 kept for pass 1 (16 B per pixel) while the stack is not larger than 4 GiB of scratch; FEABAS_HIP_RENDER_FIELD=0: search twice
    const size_t fbytes = (size_t)NB * h * w * sizeof(double2);
    void* fbuf = nullptr;
    static const bool keep = [] { const char* e = std::getenv("FEABAS_HIP_RENDER_FIELD"); return !(e && e[0] == '0'); }();
    if (keep && fbytes <= ((size_t)4 << 30) && fb_malloc(ctx, fbytes, &fbuf) == FB_OK) g.field = (double2*)fbuf;
    FB_PROF_B(ctx, "mesh_render", (double)NB * h * w * 5.0);
    hipLaunchKernelGGL(fill_ext_kernel, dim3(fb_cdiv(4 * NB, 256)), dim3(256), 0, ctx->stream, NB, ext);
    constexpr int kMaxY = 32768;                          // blocks per launch: the grid's y extent is limited to 65535
    for (int b0 = 0; b0 < NB; b0 += kMaxY) {
        g.b0 = b0;
        hipLaunchKernelGGL(mesh_field_kernel<0>, dim3(fb_cdiv(w, TILE) * fb_cdiv(h, TILE), std::min(kMaxY, NB - b0)), dim3(TILE * TILE), 0, ctx->stream, g);
    }
    hipLaunchKernelGGL(mesh_origin_kernel, dim3(1), dim3(256), 0, ctx->stream, NB, ext, origin);
    for (int b0 = 0; b0 < NB; b0 += kMaxY) {
        g.b0 = b0;
        hipLaunchKernelGGL(mesh_field_kernel<1>, dim3(fb_cdiv(w, TILE) * fb_cdiv(h, TILE), std::min(kMaxY, NB - b0)), dim3(TILE * TILE), 0, ctx->stream, g);
    }
    const hipError_t e = hipGetLastError();
    if (fbuf) {
        hipStreamSynchronize(ctx->stream);                  // the scratch goes back to the allocation cache
        fb_free(ctx, fbuf);
    }
    FB_HIP(ctx, e);
    return FB_OK;
}

}  // extern "C"
