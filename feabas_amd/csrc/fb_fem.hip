// FEM path, assembly: a spring-linked-mesh system resident on the GPU.
//   stiffness   K = sum_e N_e^T D_e N_e * soft   (material.py:134-182, mesh.py:3058-3083, optimizer.py:802-829)
//   cross links C = S^T diag(w) S (x and y copies), rhs = S^T (w r)   (optimizer.py:832-901)
//   lambdas     relative_lambda_trace            (optimizer.py:1573-1590)
//   system      A = ls K + lc C,  b = lc rhs - ls stress            (optimizer.py:1416-1418)
// One 2x2 block per coupled vertex pair; every vertex row is owned by one thread which walks the
// triangles / matches incident to that vertex, so the assembly has no atomics and a fixed
// summation order (bitwise reproducible).  The symbolic pattern is built once per topology on the
// host (C++), the numeric phases run on the device.
#include "fb_solver.h"

#include <hipcub/hipcub.hpp>

#include <array>
#include <chrono>
#include <climits>
#include <map>
#include <tuple>
#include <thread>
#include <vector>
#include <algorithm>
#include <cmath>
#include <numeric>

struct fb_mesh_blk {
    int voff = 0, V = 0, T = 0;
    std::vector<int> tri;            // host copy (local vertex ids)
    int* d_tri = nullptr;
    int* d_vtptr = nullptr;          // [V+1] incident triangle slots per vertex
    int* d_vtidx = nullptr;          // [3T] encoded 3*t + local
    float* d_mult = nullptr;         // [T]
    int* d_model = nullptr;          // [T] material model per triangle (optional)
    double* d_nu = nullptr;          // [T]
    float* d_matmult = nullptr;      // [T]
    double2* d_vinit = nullptr;      // [V] INITIAL gear (area stretch of the stiffness functions, optional)
    int* d_func = nullptr;           // [T] stiffness function per triangle, -1 = none (optional)
    int* d_fptr = nullptr;           // tables of the stiffness functions: knots [fptr[k], fptr[k+1])
    double* d_fx = nullptr;
    double* d_fy = nullptr;
    double* d_fmm = nullptr;         // material multiplier of function k in double (non-engineering elements)
    size_t ftab_cap = 0, fcnt_cap = 0;
    double2* d_vshape = nullptr;     // [V]
    double2* d_vcur = nullptr;       // [V]
    std::vector<double> h_v;         // [V][2] host copy of the shape coordinates of the last assembly (aggregates of the multigrid)
};

struct fb_system {
    int nv = 0;
    bool finalized = false;
    std::vector<fb_mesh_blk> meshes;
    // links
    int64_t nlink = 0;
    int64_t link_cap = 0;            // capacity of the device link buffers
    std::vector<int> nodes;          // [K][6] global free vertex ids, -1 = locked side
    int* d_nodes = nullptr;
    int* d_vmptr = nullptr;          // [nv+1]
    int* d_vmidx = nullptr;          // encoded 6*i + slot
    double* d_bary = nullptr;        // [K][6] signed (+B0 | -B1)
    float* d_w = nullptr;            // [K]
    double2* d_rxy = nullptr;        // [K]
    // pattern (host copy) + values
    std::vector<int> browptr, bcol;
    double* d_K = nullptr;           // [nnzb][4]
    double* d_Cacc = nullptr;        // [nnzb]
    float* d_C = nullptr;            // [nnzb]
    double2* d_rhs = nullptr;        // [nv]
    float2* d_stress = nullptr;      // [nv]
    double* d_parts = nullptr;       // reduction scratch
    double* d_glambda = nullptr;     // per-group stiffness lambda
    int glambda_cap = 0;
    void* d_gstat = nullptr;         // per-group solve statistics (relres f64, iters i32, flag i32)
    int gstat_cap = 0;
    fb_bsr* M = nullptr;             // A + PCG workspace, shares the pattern
    double* d_Kscr = nullptr;        // scratch rows for meshes that ADD into shared rows (grouped meshes)
    float2* d_sscr = nullptr;
    // host scratch of the batched tile-pair stages (fb_pairs_*)
    std::vector<int32_t> h_nodes6;
    std::vector<double> h_bary6, h_B1, h_dxy, h_x;
};

namespace {

constexpr int kT = 256;

__device__ __forceinline__ int find_col(const int* __restrict__ col, int lo, int hi, int c) {
    for (int j = lo; j < hi; ++j)
        if (col[j] == c) return j;
    return -1;
}

// tangent stiffness rows (2 x 6) and internal force (2) of local vertex `a` of one St-Venant-Kirchhoff (model 1)
// or Neo-Hookean (model 2) element, float32 like the reference (material.py:185-309, DTYPE = float32)
__device__ __forceinline__ void nonlinear_element_rows(const double2* pts, const float* uv, int model, float nu, int a,
                                                      float (&Krow)[2][6], float (&Prow)[2]) {
#pragma clang fp contract(off)
    double ex[3], ey[3];
    ex[0] = pts[1].x - pts[2].x; ey[0] = pts[1].y - pts[2].y;
    ex[1] = pts[2].x - pts[0].x; ey[1] = pts[2].y - pts[0].y;
    ex[2] = pts[0].x - pts[1].x; ey[2] = pts[0].y - pts[1].y;
    const double area2 = fabs(ex[0] * ey[1] - ey[0] * ex[1]);
    const float af = (float)area2;
    float B[4][6];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) B[r][c] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t0 = (float)(ey[i] / area2), t1 = (float)(-ex[i] / area2);
        B[0][2 * i] = t0; B[1][2 * i] = t1; B[2][2 * i + 1] = t0; B[3][2 * i + 1] = t1;
    }
    float g[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) acc += B[r][c] * uv[c];
        g[r] = acc;
    }
    const float F00 = 1.f + g[0], F01 = g[1], F10 = g[2], F11 = 1.f + g[3];      // Ft = (B uv).reshape(2,2) + I
    float M4[4][4];      // K = scale * B^T M4 B (+ Bn^T D Bn for SVK)
    float Pv[4];         // P = scale * B^T Pv     (SVK: handled through Bn)
    float scale;
    float Bn[3][6]; float Dm[3][3]; float S3[3];
    const bool svk = model == 1;
    if (svk) {
        // E = 1/2 (Ft^T Ft - I) in Voigt form, S = D E, geometric term Sg (material.py:261-291)
        const float e00 = 0.5f * (F00 * F00 + F10 * F10 - 1.f), e11 = 0.5f * (F01 * F01 + F11 * F11 - 1.f);
        const float e01 = 0.5f * (F00 * F01 + F10 * F11);
        const float E3[3] = {e00, e11, e01 + e01};
        Dm[0][0] = 1.f; Dm[0][1] = nu; Dm[0][2] = 0.f; Dm[1][0] = nu; Dm[1][1] = 1.f; Dm[1][2] = 0.f;
        Dm[2][0] = 0.f; Dm[2][1] = 0.f; Dm[2][2] = (1.f - nu) / 2.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) S3[r] = Dm[r][0] * E3[0] + Dm[r][1] * E3[1] + Dm[r][2] * E3[2];
        // Bc = [[1,0,1,0],[0,1,0,1]] B ; Fc column 2i+c = FtT[:, c] ; Bn = [Bc*Fc ; sum(Bc*Fc[::-1])]
        const float FtT[2][2] = {{F00, F10}, {F01, F11}};
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const float bc0 = B[0][c] + B[2][c], bc1 = B[1][c] + B[3][c];
            const float fc0 = FtT[0][c & 1], fc1 = FtT[1][c & 1];
            Bn[0][c] = bc0 * fc0; Bn[1][c] = bc1 * fc1; Bn[2][c] = bc0 * fc1 + bc1 * fc0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) M4[r][c] = 0.f;
        M4[0][0] = S3[0]; M4[2][2] = S3[0]; M4[1][1] = S3[1]; M4[3][3] = S3[1];
        M4[0][1] = S3[2]; M4[1][0] = S3[2]; M4[2][3] = S3[2]; M4[3][2] = S3[2];
        scale = af;
    } else {
        // Neo-Hookean (material.py:293-302): K = a/2 B^T (I - U/J + Fu Fu^T / J^2) B, P = a/2 B^T (I - U/J) F
        const float J = F00 * F11 - F01 * F10;
        const float Fv[4] = {F00, F01, F10, F11};
        const float Fu[4] = {F11, -F10, -F01, F00};
        const float U[4][4] = {{0, 0, 0, 1}, {0, 0, -1, 0}, {0, -1, 0, 0}, {1, 0, 0, 0}};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float base = (r == c ? 1.f : 0.f) - U[r][c] / J;
                M4[r][c] = base + (Fu[r] * Fu[c]) / (J * J);
                acc += base * Fv[c];
            }
            Pv[r] = acc;
        }
        scale = 0.5f * af;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int p = 2 * a + k;
        // row p of B^T M4 B
        float bm[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bm[c] = B[0][p] * M4[0][c] + B[1][p] * M4[1][c] + B[2][p] * M4[2][c] + B[3][p] * M4[3][c];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            float v = bm[0] * B[0][q] + bm[1] * B[1][q] + bm[2] * B[2][q] + bm[3] * B[3][q];
            if (svk) {
                float bd[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) bd[c] = Bn[0][p] * Dm[0][c] + Bn[1][p] * Dm[1][c] + Bn[2][p] * Dm[2][c];
                v = (bd[0] * Bn[0][q] + bd[1] * Bn[1][q] + bd[2] * Bn[2][q]) + v;
            }
            Krow[k][q] = scale * v;
        }
        if (svk) Prow[k] = scale * (Bn[0][p] * S3[0] + Bn[1][p] * S3[1] + Bn[2][p] * S3[2]);
        else Prow[k] = scale * (B[0][p] * Pv[0] + B[1][p] * Pv[1] + B[2][p] * Pv[2] + B[3][p] * Pv[3]);
    }
}

// Stiffness that follows the area stretch of the triangle (Material._stiffness_func, material.py:172-173, 307-308; the
// "wrinkle" material of configs/default_material_table.yaml:46-56): per triangle a piecewise-linear function
// (material.asymmetrical_elasticity, material.py:546-551 = scipy interp1d, linear, constant beyond the ends) of
// (area at the current gear / area at the INITIAL gear) / base, base = the same ratio of the summed |areas| of the linear
// triangles (mesh.py:2952-2963, 3030-3040).
struct StretchArgs {
    const double2* vi = nullptr;      // INITIAL vertices
    const int* func = nullptr;        // per triangle, -1 = none
    const int* fptr = nullptr;
    const double* fx = nullptr;
    const double* fy = nullptr;
    const double* fmm = nullptr;
    double base = 1.0;
};
__device__ __forceinline__ double tri_signed_area(const double2 p0, const double2 p1, const double2 p2) {       // common.py:672-676
#pragma clang fp contract(off)
    return (p1.x - p0.x) * (p2.y - p1.y) - (p1.y - p0.y) * (p2.x - p1.x);
}
__device__ __forceinline__ double stretch_factor(const StretchArgs& sa, int k, double x) {
#pragma clang fp contract(off)
    const int o = sa.fptr[k], n = sa.fptr[k + 1] - o;
    const double* xs = sa.fx + o;
    const double* ys = sa.fy + o;
    int hi = 0;
    while (hi < n && xs[hi] < x) ++hi;                   // searchsorted, side = left
    hi = min(max(hi, 1), n - 1);
    const int lo = hi - 1;
    const double slope = (ys[hi] - ys[lo]) / (xs[hi] - xs[lo]);
    double y = slope * (x - xs[lo]) + ys[lo];
    if (x < xs[0]) y = ys[0];
    if (x > xs[n - 1]) y = ys[n - 1];
    return y;
}
// sums of |area| at the INITIAL and at the current gear over the linear triangles (engineering, no function) and over all:
// parts[4 b .. 4 b + 3] of block b = {lin cur, lin ini, all cur, all ini}
__global__ void area_sums_kernel(int T, const int* __restrict__ tri, const double2* __restrict__ vi, const double2* __restrict__ vc,
                                 const int* __restrict__ model, const int* __restrict__ func, double* __restrict__ parts) {
    __shared__ double sh[4][kT];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
        const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
        const double a1 = fabs(tri_signed_area(vc[i0], vc[i1], vc[i2])), a0 = fabs(tri_signed_area(vi[i0], vi[i1], vi[i2]));
        acc[2] += a1; acc[3] += a0;
        if ((model ? model[t] : 0) == 0 && func[t] < 0) { acc[0] += a1; acc[1] += a0; }
    }
    for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int off = kT / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < 4; ++k) parts[4 * blockIdx.x + k] = sh[k][0];
}

// thread per vertex row of one mesh: element stiffness blocks of the incident triangles.
// model == nullptr: every triangle is a linear engineering element with Poisson ratio cnu_u / multiplier mult.
// Otherwise per-triangle model (0 ENG, 1 SVK, 2 NHK), nu and material multiplier (mesh.py:2914-2933, 2992-3054).
__global__ void asm_stiffness_kernel(int voff, int V, const int* __restrict__ tri, const int* __restrict__ vtptr,
                                     const int* __restrict__ vtidx, const double2* __restrict__ vs, const double2* __restrict__ vc,
                                     const float* __restrict__ mult, double c2_u, double cnu_u, double soft, float softf,
                                     const int* __restrict__ model, const double* __restrict__ tri_nu, const float* __restrict__ matmult,
                                     const int* __restrict__ rowptr, const int* __restrict__ col, double* __restrict__ Kval,
                                     float2* __restrict__ stress, const StretchArgs sa) {
#pragma clang fp contract(off)
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int grow = voff + v;
    const int lo = rowptr[grow], hi = rowptr[grow + 1];
    for (int j = lo; j < hi; ++j) reinterpret_cast<double4*>(Kval)[j] = make_double4(0.0, 0.0, 0.0, 0.0);
    double sx = 0.0, sy = 0.0;        // linear part: K_lin (v_cur - v_shape)
    double px = 0.0, py = 0.0;        // internal force of the non-linear elements
    for (int s = vtptr[v]; s < vtptr[v + 1]; ++s) {
        const int enc = vtidx[s];
        const int t = enc / 3, a = enc - 3 * t;
        const int idx[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
        const double2 pts[3] = {vs[idx[0]], vs[idx[1]], vs[idx[2]]};
        const int md = model ? model[t] : 0;
        float m = mult ? mult[t] : 1.0f;
        const int fk = sa.func ? sa.func[t] : -1;
        double fs = 1.0;                                  // f(area stretch) of the triangle's material
        if (fk >= 0) {
            const double2* vcur = vc ? vc : vs;
            const double st = tri_signed_area(vcur[idx[0]], vcur[idx[1]], vcur[idx[2]]) / tri_signed_area(sa.vi[idx[0]], sa.vi[idx[1]], sa.vi[idx[2]]);
            fs = stretch_factor(sa, fk, st / sa.base);
        }
        if (matmult && (fk < 0 || md == 0)) m = m * matmult[t];
        if (md == 0) {
            // e_i = p_{i+1} - p_{i-1}   (material.py:146-148)
            double ex[3], ey[3];
            ex[0] = pts[1].x - pts[2].x; ey[0] = pts[1].y - pts[2].y;
            ex[1] = pts[2].x - pts[0].x; ey[1] = pts[2].y - pts[0].y;
            ex[2] = pts[0].x - pts[1].x; ey[2] = pts[0].y - pts[1].y;
            const double area2 = fabs(ex[0] * ey[1] - ey[0] * ex[1]);
            const double sq = sqrt(area2);
            for (int k = 0; k < 3; ++k) { ex[k] = ex[k] / sq; ey[k] = ey[k] / sq; }
            const double nu = tri_nu ? tri_nu[t] : cnu_u;
            const double c2 = tri_nu ? (1.0 - nu) / 2.0 : c2_u;
            // D = diag(m, m, m(1-nu)/2) + nu*m coupling, stored float32 (material.py:174-180)
            // (with a stiffness function the float32 product of the multipliers goes through double before D is stored)
            const double mf = fk >= 0 ? (double)m * fs : (double)m;
            const double d0 = (double)(float)mf;
            const double d2 = (double)(float)(mf * c2);
            const double dn = (double)(float)(mf * nu);
            const double exa = ex[a], eya = ey[a];
            for (int b = 0; b < 3; ++b) {
                const int j = find_col(col, lo, hi, voff + idx[b]);
                if (j < 0) continue;
                const double exb = ex[b], eyb = ey[b];
                const double kxx = d0 * (eya * eyb) + d2 * (exa * exb);            // (a,x),(b,x)
                const double kxy = -(dn * (eya * exb)) - d2 * (exa * eyb);         // (a,x),(b,y)
                const double kyx = -(dn * (exa * eyb)) - d2 * (eya * exb);         // (a,y),(b,x)
                const double kyy = d0 * (exa * exb) + d2 * (eya * eyb);            // (a,y),(b,y)
                double4 k = reinterpret_cast<double4*>(Kval)[j];
                k.x += kxx; k.y += kxy; k.z += kyx; k.w += kyy;
                reinterpret_cast<double4*>(Kval)[j] = k;
                if (vc) {
                    const double dxv = vc[idx[b]].x - vs[idx[b]].x, dyv = vc[idx[b]].y - vs[idx[b]].y;
                    sx += kxx * dxv + kxy * dyv;
                    sy += kyx * dxv + kyy * dyv;
                }
            }
        } else {
            float uv[6];
            for (int b = 0; b < 3; ++b) {
                uv[2 * b] = vc ? (float)(vc[idx[b]].x - vs[idx[b]].x) : 0.f;
                uv[2 * b + 1] = vc ? (float)(vc[idx[b]].y - vs[idx[b]].y) : 0.f;
            }
            float Krow[2][6], Prow[2];
            nonlinear_element_rows(pts, uv, md, tri_nu ? (float)tri_nu[t] : (float)cnu_u, a, Krow, Prow);
            if (fk < 0) {
                for (int b = 0; b < 3; ++b) {
                    const int j = find_col(col, lo, hi, voff + idx[b]);
                    if (j < 0) continue;
                    double4 k = reinterpret_cast<double4*>(Kval)[j];
                    k.x += (double)(Krow[0][2 * b] * m); k.y += (double)(Krow[0][2 * b + 1] * m);
                    k.z += (double)(Krow[1][2 * b] * m); k.w += (double)(Krow[1][2 * b + 1] * m);
                    reinterpret_cast<double4*>(Kval)[j] = k;
                }
                px += (double)(Prow[0] * m); py += (double)(Prow[1] * m);
            } else {
                // modifier = material multiplier x f(J) in double, times the mesh multiplier (material.py:307-308, mesh.py:3041)
                const double mm = (double)m * (sa.fmm[fk] * fs);
                for (int b = 0; b < 3; ++b) {
                    const int j = find_col(col, lo, hi, voff + idx[b]);
                    if (j < 0) continue;
                    double4 k = reinterpret_cast<double4*>(Kval)[j];
                    k.x += (double)Krow[0][2 * b] * mm; k.y += (double)Krow[0][2 * b + 1] * mm;
                    k.z += (double)Krow[1][2 * b] * mm; k.w += (double)Krow[1][2 * b + 1] * mm;
                    reinterpret_cast<double4*>(Kval)[j] = k;
                }
                px += (double)(float)((double)Prow[0] * mm); py += (double)(float)((double)Prow[1] * mm);
            }
        }
    }
    // stress = K_lin (v_cur - v_shape) -> float32, + internal force, * soft   (mesh.py:3068-3082, optimizer.py:822)
    stress[grow] = make_float2(((float)sx + (float)px) * softf, ((float)sy + (float)py) * softf);
    if (soft != 1.0) {
        for (int j = lo; j < hi; ++j) {
            double4 k = reinterpret_cast<double4*>(Kval)[j];
            k.x *= soft; k.y *= soft; k.z *= soft; k.w *= soft;
            reinterpret_cast<double4*>(Kval)[j] = k;
        }
    }
}

// rows [voff, voff + V): K += Kscr, stress += sscr (a mesh that shares its vertex rows with another mesh of its group)
__global__ void add_rows_kernel(int voff, int V, const int* __restrict__ rowptr, const double* __restrict__ Kscr, const float2* __restrict__ sscr,
                                double* __restrict__ Kval, float2* __restrict__ stress) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int grow = voff + v;
    for (int j = rowptr[grow]; j < rowptr[grow + 1]; ++j) {
        double4 k = reinterpret_cast<double4*>(Kval)[j];
        const double4 a = reinterpret_cast<const double4*>(Kscr)[j];
        k.x += a.x; k.y += a.y; k.z += a.z; k.w += a.w;
        reinterpret_cast<double4*>(Kval)[j] = k;
    }
    const float2 a = sscr[grow];
    float2 t = stress[grow];
    t.x += a.x; t.y += a.y;
    stress[grow] = t;
}

// thread per free vertex: cross-link contributions of the matches incident to it
__global__ void asm_links_kernel(int nv, const int* __restrict__ vmptr, const int* __restrict__ vmidx, const int* __restrict__ nodes,
                                 const double* __restrict__ bary, const float* __restrict__ w, const double2* __restrict__ rxy,
                                 const int* __restrict__ rowptr, const int* __restrict__ col, double* __restrict__ Cacc,
                                 float* __restrict__ Cval, double2* __restrict__ rhs) {
#pragma clang fp contract(off)
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const int lo = rowptr[v], hi = rowptr[v + 1];
    for (int j = lo; j < hi; ++j) Cacc[j] = 0.0;
    double rx = 0.0, ry = 0.0;
    for (int s = vmptr[v]; s < vmptr[v + 1]; ++s) {
        const int enc = vmidx[s];
        const int i = enc / 6, a = enc - 6 * i;
        const float wi = w[i];
        if (wi == 0.0f) continue;                       // masked-out match: contributes exact zeros
        const float sa = (float)bary[6 * (size_t)i + a];   // S is float32 (optimizer.py:896)
        for (int b = 0; b < 6; ++b) {
            const int nbv = nodes[6 * (size_t)i + b];
            if (nbv < 0) continue;
            const float sb = (float)bary[6 * (size_t)i + b];
            // w * s_a * s_b from the float32 factors, kept in double: the reference rounds the products
            // and their sum to float32, which leaves its C (and A) indefinite at the 1e-8 level; the
            // exact rank-one sum is positive semi-definite, which CG needs on systems with a null space.
            const double prod = ((double)sa * (double)sb) * (double)wi;
            const int j = find_col(col, lo, hi, nbv);
            if (j >= 0) Cacc[j] += prod;
        }
        const double2 r = rxy[i];
        rx += (double)sa * ((double)wi * r.x);          // optimizer.py:863-865
        ry += (double)sa * ((double)wi * r.y);
    }
    for (int j = lo; j < hi; ++j) Cval[j] = (float)Cacc[j];
    rhs[v] = make_double2(rx, ry);
}

// partial sums for relative_lambda_trace: [0] sum diag C, [1] sum of K diagonal where C diag != 0
__global__ void lambda_trace_kernel(int nv, const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ Kval,
                                    const float* __restrict__ Cval, double* __restrict__ parts) {
    __shared__ double sh0[kT / 64], sh1[kT / 64];
    double tc = 0.0, tk = 0.0;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += gridDim.x * blockDim.x) {
        const int j = find_col(col, rowptr[v], rowptr[v + 1], v);
        if (j < 0) continue;
        const float c = Cval[j];
        if (c != 0.0f) {
            const double4 k = reinterpret_cast<const double4*>(Kval)[j];
            tc += 2.0 * (double)c;
            tk += k.x + k.w;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { tc += __shfl_down(tc, off); tk += __shfl_down(tk, off); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh0[wave] = tc; sh1[wave] = tk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) { tc += sh0[wv]; tk += sh1[wv]; }
        parts[2 * blockIdx.x] = tc; parts[2 * blockIdx.x + 1] = tk;
    }
}

// A = ls K + lc C (x) I2 ; b = lc rhs - ls stress
__global__ void form_system_kernel(int nv, int64_t nnzb, const double* __restrict__ Kval, const double* __restrict__ Cval,
                                   const double2* __restrict__ rhs, const float2* __restrict__ stress, double ls, double lc,
                                   double* __restrict__ Aval, double2* __restrict__ b) {
#pragma clang fp contract(off)
    const float lsf = (float)ls;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnzb; j += (int64_t)gridDim.x * blockDim.x) {
        const double4 k = reinterpret_cast<const double4*>(Kval)[j];
        const double c = lc * Cval[j];
        reinterpret_cast<double4*>(Aval)[j] = make_double4(ls * k.x + c, ls * k.y, ls * k.z, ls * k.w + c);
    }
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += gridDim.x * blockDim.x) {
        const double2 r = rhs[v];
        const float2 s = stress[v];
        b[v] = make_double2(lc * r.x - (double)(lsf * s.x), lc * r.y - (double)(lsf * s.y));
    }
}

// one workgroup per group of `gs` consecutive vertices: lambda_s of relative_lambda_trace for that group
__global__ void group_lambda_kernel(int gs, const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ Kval,
                                    const float* __restrict__ Cval, double sl, double cl, double* __restrict__ lam) {
    __shared__ double sh0[kT / 64], sh1[kT / 64];
    const int g = blockIdx.x;
    double tc = 0.0, tk = 0.0;
    for (int i = threadIdx.x; i < gs; i += blockDim.x) {
        const int v = g * gs + i;
        const int j = find_col(col, rowptr[v], rowptr[v + 1], v);
        if (j < 0) continue;
        const float c = Cval[j];
        if (c != 0.0f) {
            const double4 k = reinterpret_cast<const double4*>(Kval)[j];
            tc += 2.0 * (double)c;
            tk += k.x + k.w;
        }
    }
    for (int off = 32; off > 0; off >>= 1) { tc += __shfl_down(tc, off); tk += __shfl_down(tk, off); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh0[wave] = tc; sh1[wave] = tk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) { tc += sh0[wv]; tk += sh1[wv]; }
        double l = sl;
        // a range without links keeps its bare stiffness (b = 0 there, so x stays 0) instead of a zero block
        if (sl < 0 || cl < 0) l = (tc == 0.0) ? 1.0 : fabs(fabs(sl / cl) * tc / tk);
        lam[g] = l;
    }
}

// one workgroup per group: e[g] = x^T K x over the group's vertex range (K block-diagonal per group)
__global__ void group_energy_kernel(int gs, const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ Kval,
                                    const double2* __restrict__ x, double* __restrict__ e) {
    __shared__ double sh[kT / 64];
    const int g = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < gs; i += blockDim.x) {
        const int v = g * gs + i;
        double2 y = make_double2(0.0, 0.0);
        for (int j = rowptr[v]; j < rowptr[v + 1]; ++j) {
            const double4 k = reinterpret_cast<const double4*>(Kval)[j];
            const double2 u = x[col[j]];
            y.x += k.x * u.x + k.y * u.y;
            y.y += k.z * u.x + k.w * u.y;
        }
        const double2 xi = x[v];
        acc += xi.x * y.x + xi.y * y.y;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) acc += sh[w];
        e[g] = acc;
    }
}

// thread per vertex row: A = ls[g] K + lc C (x) I2, b = lc rhs - ls[g] stress
__global__ void form_groups_kernel(int nv, int gs, const int* __restrict__ rowptr, const double* __restrict__ Kval,
                                   const double* __restrict__ Cval, const double2* __restrict__ rhs, const float2* __restrict__ stress,
                                   const double* __restrict__ lam, double lc, double* __restrict__ Aval, double2* __restrict__ b) {
#pragma clang fp contract(off)
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const double ls = lam[v / gs];
    const float lsf = (float)ls;
    for (int j = rowptr[v]; j < rowptr[v + 1]; ++j) {
        const double4 k = reinterpret_cast<const double4*>(Kval)[j];
        const double c = lc * Cval[j];
        reinterpret_cast<double4*>(Aval)[j] = make_double4(ls * k.x + c, ls * k.y, ls * k.z, ls * k.w + c);
    }
    const double2 r = rhs[v];
    const float2 s = stress[v];
    b[v] = make_double2(lc * r.x - (double)(lsf * s.x), lc * r.y - (double)(lsf * s.y));
}

template <typename T>
int upload(fb_ctx* ctx, T** dptr, const T* host, size_t count) {
    if (!*dptr) FB_HIP(ctx, hipMalloc((void**)dptr, std::max<size_t>(16, sizeof(T) * count)));
    if (host && count) { const int rc_ = fb_copy_h2d(ctx, *dptr, host, sizeof(T) * count); if (rc_) return rc_; }
    return FB_OK;
}

}  // namespace

// ---- symbolic phase: coupled vertex pairs as sortable keys
__global__ void pattern_tri_keys_kernel(int T, int voff, const int* __restrict__ tri, uint64_t* __restrict__ keys) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int v[3] = {voff + tri[3 * t], voff + tri[3 * t + 1], voff + tri[3 * t + 2]};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) keys[9 * (size_t)t + 3 * a + b] = ((uint64_t)(unsigned)v[a] << 32) | (unsigned)v[b];
}
__global__ void pattern_link_keys_kernel(int64_t K, const int* __restrict__ nodes, uint64_t* __restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    int v[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) v[a] = nodes[6 * i + a];
    const uint64_t dead = (uint64_t)0xffffffffu << 32;                       // a slot of a locked mesh: sorts behind every row
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b)
            keys[36 * (size_t)i + 6 * a + b] = (v[a] >= 0 && v[b] >= 0) ? (((uint64_t)(unsigned)v[a] << 32) | (unsigned)v[b]) : dead;
}
__global__ void pattern_diag_keys_kernel(int nv, uint64_t* __restrict__ keys) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < nv) keys[v] = ((uint64_t)(unsigned)v << 32) | (unsigned)v;
}

// ---- links on the device: the table of matches is uploaded once; the membership test of an update and the index vertex ->
// incident match slots (CSR over free vertices, slots ascending: the assembly sums in a fixed order) are made there.  The host
// statement of the same two steps (36 binary searches per match on a few threads, a counting sort and three more copies) took
// 11 ms of the 19 ms of a numeric re-assembly at 200 k matches.
namespace {

// first match (atomicMin) that names a vertex outside [-1, nv) or couples two vertices outside the pattern
__global__ void link_member_kernel(int64_t K, const int* __restrict__ nodes, int nv, const int* __restrict__ rowptr, const int* __restrict__ col,
                                   int* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    int n6[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) n6[a] = nodes[6 * i + a];
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 6; ++a) ok = ok && n6[a] >= -1 && n6[a] < nv;
    if (ok) {
        for (int a = 0; a < 6 && ok; ++a) {
            const int u = n6[a];
            if (u < 0) continue;
            const int lo0 = rowptr[u], hi0 = rowptr[u + 1];
            for (int b = 0; b < 6 && ok; ++b) {
                const int w = n6[b];
                if (w < 0 || w == u) continue;
                int lo = lo0, hi = hi0;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (col[mid] < w) lo = mid + 1; else hi = mid; }
                ok = lo < hi0 && col[lo] == w;
            }
        }
    }
    if (!ok) atomicMin(bad, (int)i);
}

__global__ void link_count_kernel(int64_t K6, const int* __restrict__ nodes, int* __restrict__ cnt) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K6 && nodes[k] >= 0) atomicAdd(&cnt[nodes[k]], 1);
}

__global__ void link_fill_kernel(int64_t K6, const int* __restrict__ nodes, const int* __restrict__ vmptr, int* __restrict__ cursor,
                                 int* __restrict__ vmidx) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K6) return;
    const int v = nodes[k];
    if (v >= 0) vmidx[vmptr[v] + atomicAdd(&cursor[v], 1)] = (int)k;
}

// the slots of every vertex in ascending order, whatever order the atomics of the fill handed out
__global__ void link_sort_kernel(int nv, const int* __restrict__ vmptr, int* __restrict__ vmidx) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= nv) return;
    const int lo = vmptr[v], hi = vmptr[v + 1];
    for (int i = lo + 1; i < hi; ++i) {
        const int key = vmidx[i];
        int j = i - 1;
        while (j >= lo && vmidx[j] > key) { vmidx[j + 1] = vmidx[j]; --j; }
        vmidx[j + 1] = key;
    }
}

}  // namespace

// device link buffers sized for the current links, the table s->nodes on the device, its vertex index; check != 0: every coupled
// pair must be in the pattern (*bad = first offending match, or -1)
static int build_link_index(fb_ctx* ctx, fb_system* s, int check = 0, int64_t* bad = nullptr) {
    const int nv = s->nv;
    const int64_t K = s->nlink, K6 = 6 * K;
    if (bad) *bad = -1;
    if (K > s->link_cap || !s->d_vmidx) {
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(s->d_vmidx); hipFree(s->d_nodes); hipFree(s->d_bary); hipFree(s->d_w); hipFree(s->d_rxy);
        s->d_vmidx = nullptr; s->d_nodes = nullptr; s->d_bary = nullptr; s->d_w = nullptr; s->d_rxy = nullptr;
        const size_t cap = (size_t)std::max<int64_t>(16, K + K / 4);
        FB_HIP(ctx, hipMalloc((void**)&s->d_vmidx, sizeof(int) * 6 * cap));
        FB_HIP(ctx, hipMalloc((void**)&s->d_nodes, sizeof(int) * 6 * cap));
        FB_HIP(ctx, hipMalloc((void**)&s->d_bary, sizeof(double) * 6 * cap));
        FB_HIP(ctx, hipMalloc((void**)&s->d_w, sizeof(float) * cap));
        FB_HIP(ctx, hipMalloc((void**)&s->d_rxy, sizeof(double2) * cap));
        s->link_cap = (int64_t)cap;
    }
    if (!s->d_vmptr) FB_HIP(ctx, hipMalloc((void**)&s->d_vmptr, sizeof(int) * ((size_t)nv + 1)));
    int rc;
    if (K && (rc = fb_copy_h2d(ctx, s->d_nodes, s->nodes.data(), sizeof(int) * (size_t)K6))) return rc;
    if (check && K) {
        if (!s->M || !s->M->d.rowptr) return fb_fail(ctx, FB_ERR_ARG, "fb_sys_update_links: the system has no pattern on the device yet");
        int* d_bad = reinterpret_cast<int*>(ctx->small);
        const int init = INT_MAX;
        if ((rc = fb_copy_h2d(ctx, d_bad, &init, sizeof(int)))) return rc;
        hipLaunchKernelGGL(link_member_kernel, dim3((unsigned)fb_cdiv(K, kT)), dim3(kT), 0, ctx->stream, K, s->d_nodes, nv, s->M->d.rowptr, s->M->d.col, d_bad);
        FB_HIP(ctx, hipGetLastError());
        int first = INT_MAX;
        if ((rc = fb_copy_d2h(ctx, &first, d_bad, sizeof(int)))) return rc;
        if (first != INT_MAX) { if (bad) *bad = first; return FB_OK; }
    }
    // counts -> exclusive scan -> fill -> per-vertex sort
    void *d_cnt = nullptr, *d_tmp = nullptr;
    if ((rc = fb_malloc(ctx, sizeof(int) * ((size_t)nv + 1), &d_cnt))) return rc;
    FB_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(int) * ((size_t)nv + 1), ctx->stream));
    if (K6) {
        hipLaunchKernelGGL(link_count_kernel, dim3((unsigned)fb_cdiv(K6, kT)), dim3(kT), 0, ctx->stream, K6, s->d_nodes, (int*)d_cnt);
        FB_HIP(ctx, hipGetLastError());
    }
    size_t tmp_bytes = 0;
    FB_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (int*)d_cnt, s->d_vmptr, nv + 1, ctx->stream));
    if ((rc = fb_malloc(ctx, std::max<size_t>(tmp_bytes, 16), &d_tmp))) { fb_free(ctx, d_cnt); return rc; }
    FB_HIP(ctx, hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, (int*)d_cnt, s->d_vmptr, nv + 1, ctx->stream));
    if (K6) {
        FB_HIP(ctx, hipMemsetAsync(d_cnt, 0, sizeof(int) * (size_t)nv, ctx->stream));        // now the fill cursors
        hipLaunchKernelGGL(link_fill_kernel, dim3((unsigned)fb_cdiv(K6, kT)), dim3(kT), 0, ctx->stream, K6, s->d_nodes, s->d_vmptr, (int*)d_cnt, s->d_vmidx);
        hipLaunchKernelGGL(link_sort_kernel, dim3((unsigned)fb_cdiv(nv, kT)), dim3(kT), 0, ctx->stream, nv, s->d_vmptr, s->d_vmidx);
        FB_HIP(ctx, hipGetLastError());
    }
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the scratch goes back to the allocation cache
    fb_free(ctx, d_cnt); fb_free(ctx, d_tmp);
    return FB_OK;
}

#include "fb_mg.inc"

extern "C" {

int fb_sys_create(fb_ctx* ctx, int64_t nvert_free, fb_system** out) {
    FB_CHECK_ARG(ctx, nvert_free > 0 && nvert_free < (1LL << 30) && out);
    fb_system* s = new fb_system();
    s->nv = (int)nvert_free;
    *out = s;
    return FB_OK;
}

void fb_sys_destroy(fb_ctx* ctx, fb_system* s) {
    if (!s) return;
    hipStreamSynchronize(ctx->stream);
    for (auto& m : s->meshes) {
        hipFree(m.d_tri); hipFree(m.d_vtptr); hipFree(m.d_vtidx); hipFree(m.d_mult); hipFree(m.d_vshape); hipFree(m.d_vcur);
        hipFree(m.d_model); hipFree(m.d_nu); hipFree(m.d_matmult);
        hipFree(m.d_vinit); hipFree(m.d_func); hipFree(m.d_fptr); hipFree(m.d_fx); hipFree(m.d_fy); hipFree(m.d_fmm);
    }
    hipFree(s->d_nodes); hipFree(s->d_vmptr); hipFree(s->d_vmidx); hipFree(s->d_bary); hipFree(s->d_w); hipFree(s->d_rxy);
    hipFree(s->d_glambda); hipFree(s->d_gstat); hipFree(s->d_Kscr); hipFree(s->d_sscr); hipFree(s->d_K); hipFree(s->d_Cacc); hipFree(s->d_C); hipFree(s->d_rhs); hipFree(s->d_stress); hipFree(s->d_parts);
    if (s->M) fb_bsr_free(ctx, s->M);
    delete s;
}

int fb_sys_add_mesh(fb_ctx* ctx, fb_system* s, int64_t voff, const int32_t* tri, int V, int T, int* mesh_id) {
    FB_CHECK_ARG(ctx, s && !s->finalized && tri && V > 0 && T >= 0 && voff >= 0 && voff + V <= s->nv);
    for (int i = 0; i < 3 * T; ++i) FB_CHECK_ARG(ctx, tri[i] >= 0 && tri[i] < V);
    fb_mesh_blk m;
    m.voff = (int)voff; m.V = V; m.T = T;
    m.tri.assign(tri, tri + 3 * (size_t)T);
    s->meshes.push_back(std::move(m));
    if (mesh_id) *mesh_id = (int)s->meshes.size() - 1;
    return FB_OK;
}

int fb_sys_set_links(fb_ctx* ctx, fb_system* s, int64_t K, const int32_t* nodes6) {
    FB_CHECK_ARG(ctx, s && !s->finalized && K >= 0 && (K == 0 || nodes6) && K < (1LL << 31) / 6);
    for (int64_t i = 0; i < 6 * K; ++i) FB_CHECK_ARG(ctx, nodes6[i] >= -1 && nodes6[i] < s->nv);
    s->nlink = K;
    s->nodes.assign(nodes6, nodes6 + 6 * K);
    return FB_OK;
}

int fb_sys_finalize(fb_ctx* ctx, fb_system* s, int64_t* nnzb_out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && !s->finalized);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    const int nv = s->nv;
    // FEABAS_HIP_FEM_TRACE=1: wall time of the steps of the symbolic phase on stderr
    const bool trace = std::getenv("FEABAS_HIP_FEM_TRACE") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "fb_sys_finalize %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // ---- symbolic pattern on the device: every coupled vertex pair as a 64-bit key (row << 32 | col) -- 9 per triangle, live^2
    //      per match, the diagonal of every row -- radix sorted and made unique; the sorted keys ARE the block CSR (the
    //      host statement of the same steps, a bucket pass and a sort per row, took 57 of the 64 ms of a first assembly at 1e6 DoF)
    {
        int rc;
        size_t ntri_keys = 0;
        for (auto& m : s->meshes) ntri_keys += 9 * (size_t)m.T;
        const size_t nkeys = ntri_keys + 36 * (size_t)s->nlink + (size_t)nv;
        // (hipcub takes the number of keys as an int: checked before anything is allocated)
        if (nkeys >= (size_t)INT_MAX) return fb_fail(ctx, FB_ERR_ARG, "pattern exceeds int32 block indexing");
        void *d_keys = nullptr, *d_keys2 = nullptr, *d_tmp = nullptr, *d_nodes_tmp = nullptr, *d_tri_tmp = nullptr;
        // every way out of this block -- the HIP error returns included -- hands the temporaries back to the context's cache
        struct Release {
            fb_ctx* c; void **a, **b, **t, **n, **r;
            ~Release() { fb_free(c, *a); fb_free(c, *b); fb_free(c, *t); fb_free(c, *n); fb_free(c, *r); }
        } guard{ctx, &d_keys, &d_keys2, &d_tmp, &d_nodes_tmp, &d_tri_tmp};
        if ((rc = fb_malloc(ctx, sizeof(uint64_t) * nkeys, &d_keys))) return rc;
        if ((rc = fb_malloc(ctx, sizeof(uint64_t) * nkeys, &d_keys2))) return rc;
        size_t at = 0;
        for (auto& m : s->meshes) {
            if (!m.T) continue;
            if ((rc = fb_malloc(ctx, sizeof(int) * 3 * (size_t)m.T, &d_tri_tmp))) return rc;
            if ((rc = fb_copy_h2d(ctx, d_tri_tmp, m.tri.data(), sizeof(int) * 3 * (size_t)m.T))) return rc;
            hipLaunchKernelGGL(pattern_tri_keys_kernel, dim3((unsigned)fb_cdiv(m.T, kT)), dim3(kT), 0, ctx->stream, m.T, m.voff, (const int*)d_tri_tmp,
                               (uint64_t*)d_keys + at);
            at += 9 * (size_t)m.T;
            FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            fb_free(ctx, d_tri_tmp); d_tri_tmp = nullptr;
        }
        if (s->nlink) {
            if ((rc = fb_malloc(ctx, sizeof(int) * 6 * (size_t)s->nlink, &d_nodes_tmp))) return rc;
            if ((rc = fb_copy_h2d(ctx, d_nodes_tmp, s->nodes.data(), sizeof(int) * 6 * (size_t)s->nlink))) return rc;
            hipLaunchKernelGGL(pattern_link_keys_kernel, dim3((unsigned)fb_cdiv(s->nlink, kT)), dim3(kT), 0, ctx->stream, s->nlink, (const int*)d_nodes_tmp,
                               (uint64_t*)d_keys + at);
            at += 36 * (size_t)s->nlink;
        }
        hipLaunchKernelGGL(pattern_diag_keys_kernel, dim3((unsigned)fb_cdiv(nv, kT)), dim3(kT), 0, ctx->stream, nv, (uint64_t*)d_keys + at);
        FB_HIP(ctx, hipGetLastError());
        lap("keys");
        int row_bits = 1;
        while ((1LL << row_bits) < (int64_t)nv + 1) ++row_bits;                        // (the all-ones row of a dead link slot sorts behind every row)
        size_t tmp_sort = 0, tmp_uniq = 0;
        int* d_count = reinterpret_cast<int*>(ctx->small);
        FB_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_sort, (uint64_t*)d_keys, (uint64_t*)d_keys2, (int)nkeys, 0, 32 + row_bits, ctx->stream));
        FB_HIP(ctx, hipcub::DeviceSelect::Unique(nullptr, tmp_uniq, (uint64_t*)d_keys2, (uint64_t*)d_keys, d_count, (int)nkeys, ctx->stream));
        if ((rc = fb_malloc(ctx, std::max<size_t>(std::max(tmp_sort, tmp_uniq), 16), &d_tmp))) return rc;
        FB_HIP(ctx, hipcub::DeviceRadixSort::SortKeys(d_tmp, tmp_sort, (uint64_t*)d_keys, (uint64_t*)d_keys2, (int)nkeys, 0, 32 + row_bits, ctx->stream));
        FB_HIP(ctx, hipcub::DeviceSelect::Unique(d_tmp, tmp_uniq, (uint64_t*)d_keys2, (uint64_t*)d_keys, d_count, (int)nkeys, ctx->stream));
        int nuniq = 0;
        if ((rc = fb_copy_d2h(ctx, &nuniq, d_count, sizeof(int)))) return rc;
        lap("sort + unique");
        // the key of the dead slots is the tail of the sorted list
        std::vector<uint64_t> hk((size_t)nuniq);
        if (nuniq && (rc = fb_copy_d2h(ctx, hk.data(), d_keys, sizeof(uint64_t) * (size_t)nuniq))) return rc;
        while (!hk.empty() && (hk.back() >> 32) >= (uint64_t)nv) hk.pop_back();
        s->browptr.assign((size_t)nv + 1, 0);
        s->bcol.resize(hk.size());
        for (size_t j = 0; j < hk.size(); ++j) {
            s->bcol[j] = (int)(hk[j] & 0xffffffffu);
            s->browptr[(size_t)(hk[j] >> 32) + 1]++;
        }
        for (int v = 0; v < nv; ++v) s->browptr[v + 1] += s->browptr[v];
        lap("keys to host, row pointers");
    }
    const int64_t nnzb = (int64_t)s->bcol.size();
    // ---- device pattern lives in the solver matrix
    int rc = fb_bsr_alloc(ctx, nv, nnzb, &s->M);
    if (rc) return rc;
    { const int rc_ = fb_copy_h2d(ctx, s->M->d.rowptr, s->browptr.data(), sizeof(int) * ((size_t)nv + 1)); if (rc_) return rc_; }
    { int mr = 0; for (int v = 0; v < nv; ++v) mr = std::max(mr, s->browptr[v + 1] - s->browptr[v]); s->M->max_row_blocks = mr; }
    { const int rc_ = fb_copy_h2d(ctx, s->M->d.col, s->bcol.data(), sizeof(int) * (size_t)nnzb); if (rc_) return rc_; }
    FB_HIP(ctx, hipMalloc((void**)&s->d_K, sizeof(double) * 4 * (size_t)nnzb));
    FB_HIP(ctx, hipMalloc((void**)&s->d_Cacc, sizeof(double) * (size_t)nnzb));
    FB_HIP(ctx, hipMalloc((void**)&s->d_C, sizeof(float) * (size_t)nnzb));
    FB_HIP(ctx, hipMalloc((void**)&s->d_rhs, sizeof(double2) * (size_t)nv));
    FB_HIP(ctx, hipMalloc((void**)&s->d_stress, sizeof(float2) * (size_t)nv));
    FB_HIP(ctx, hipMalloc((void**)&s->d_parts, sizeof(double) * 2 * 1024));
    FB_HIP(ctx, hipMemsetAsync(s->d_K, 0, sizeof(double) * 4 * (size_t)nnzb, ctx->stream));
    FB_HIP(ctx, hipMemsetAsync(s->d_C, 0, sizeof(float) * (size_t)nnzb, ctx->stream));
    FB_HIP(ctx, hipMemsetAsync(s->d_rhs, 0, sizeof(double2) * (size_t)nv, ctx->stream));
    FB_HIP(ctx, hipMemsetAsync(s->d_stress, 0, sizeof(float2) * (size_t)nv, ctx->stream));
    lap("matrix buffers");
    // ---- vertex -> incident triangle slots, per mesh
    for (auto& m : s->meshes) {
        std::vector<int> ptr((size_t)m.V + 1, 0), idx(3 * (size_t)m.T);
        for (size_t k = 0; k < 3 * (size_t)m.T; ++k) ptr[(size_t)m.tri[k] + 1]++;
        for (int v = 0; v < m.V; ++v) ptr[v + 1] += ptr[v];
        std::vector<int> f(ptr.begin(), ptr.end() - 1);
        for (int t = 0; t < m.T; ++t)
            for (int a = 0; a < 3; ++a) idx[f[m.tri[3 * (size_t)t + a]]++] = 3 * t + a;
        if ((rc = upload(ctx, &m.d_tri, m.tri.data(), m.tri.size()))) return rc;
        if ((rc = upload(ctx, &m.d_vtptr, ptr.data(), ptr.size()))) return rc;
        if ((rc = upload(ctx, &m.d_vtidx, idx.data(), idx.size()))) return rc;
        FB_HIP(ctx, hipMalloc((void**)&m.d_mult, std::max<size_t>(16, sizeof(float) * (size_t)m.T)));
        FB_HIP(ctx, hipMalloc((void**)&m.d_vshape, sizeof(double2) * (size_t)m.V));
        FB_HIP(ctx, hipMalloc((void**)&m.d_vcur, sizeof(double2) * (size_t)m.V));
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));          // ptr/idx are locals
    }
    lap("vertex -> triangle index");
    // ---- vertex -> incident match slots
    s->link_cap = 0;
    if ((rc = build_link_index(ctx, s))) return rc;
    lap("vertex -> match index");
    s->finalized = true;
    if (nnzb_out) *nnzb_out = nnzb;
    return FB_OK;
}

// Replace the links of a finalized system by links that add no new vertex coupling (every pair of free
// vertices of a match is already in the pattern -- e.g. matches against locked meshes, whose three free
// vertices share a triangle).  Only the vertex -> match index is rebuilt; the symbolic pattern is reused.
// trusted != 0: the caller built the matches from cells of the system's own meshes (fb_pairs_*: a match couples the three
// vertices of ONE grid triangle), so the membership test of every coupled pair is skipped
static int sys_update_links(fb_ctx* ctx, fb_system* s, int64_t K, const int32_t* nodes6, int trusted) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && K >= 0 && (K == 0 || nodes6) && K < (1LL << 31) / 6);
    const int64_t old_K = s->nlink;
    s->nlink = K;
    s->nodes.assign(nodes6, nodes6 + 6 * K);
    int64_t bad = -1;
    const int rc = build_link_index(ctx, s, trusted ? 0 : 1, &bad);
    if (rc) return rc;
    if (bad >= 0) {
        // the system keeps no links: the table is emptied AND the vertex -> match index (it still described the previous
        // link set over buffers that now hold the rejected one) -- a caller that handles the error and assembles again finds a
        // system without links, not a stale index; to use the new links it rebuilds (fb_sys_set_links + fb_sys_finalize)
        s->nlink = 0; (void)old_K;
        s->nodes.clear();
        if (s->d_vmptr) FB_HIP(ctx, hipMemsetAsync(s->d_vmptr, 0, sizeof(int) * ((size_t)s->nv + 1), ctx->stream));
        for (int a = 0; a < 6; ++a) {
            const int u = nodes6[6 * bad + a];
            if (u < -1 || u >= s->nv) return fb_fail(ctx, FB_ERR_ARG, "fb_sys_update_links: match %lld names vertex %d outside [-1, %d)", (long long)bad, u, s->nv);
            if (u < 0) continue;
            const int* lo_ = s->bcol.data() + s->browptr[u];
            const int* hi_ = s->bcol.data() + s->browptr[u + 1];
            for (int b = 0; b < 6; ++b) {
                const int w = nodes6[6 * bad + b];
                if (w < 0 || w == u) continue;
                if (w >= s->nv || !std::binary_search(lo_, hi_, w))
                    return fb_fail(ctx, FB_ERR_ARG, "fb_sys_update_links: match %lld couples vertices %d and %d outside the pattern", (long long)bad, u, w);
            }
        }
        return fb_fail(ctx, FB_ERR_ARG, "fb_sys_update_links: match %lld is outside the pattern", (long long)bad);
    }
    return FB_OK;
}

int fb_sys_update_links(fb_ctx* ctx, fb_system* s, int64_t K, const int32_t* nodes6) { return sys_update_links(ctx, s, K, nodes6, 0); }

// Link.xy0 / xy1 / dxy (optimizer.py:121-135, 248-255) and the rows fb_sys_set_links / fb_sys_assemble_links take, for the
// K matches of one link: nodes6 = global free-vertex ids of the two triangles (-1 on a locked side), bary6 = [B0 | -B1],
// rxy = (B1 . v1[tri1[tid1]] - B0 . v0[tri0[tid0]]) + (ox, oy).  Host only, a few threads: a section of 1e5 matches cost
// 60 ms of numpy gathers per solve.  voff < 0: that mesh is locked.  bary6 / rxy may be NULL (nodes only).
int fb_link_terms(fb_ctx* ctx, int64_t K, const int32_t* tri0, int64_t T0, const double* v0, const int64_t* tid0, const double* B0, int64_t voff0,
                  const int32_t* tri1, int64_t T1, const double* v1, const int64_t* tid1, const double* B1, int64_t voff1, double ox, double oy,
                  int32_t* nodes6, double* bary6, double* rxy) {
#pragma clang fp contract(off)
    FB_CHECK_ARG(ctx, K >= 0 && tri0 && tri1 && T0 > 0 && T1 > 0 && (K == 0 || (tid0 && tid1 && nodes6)));
    FB_CHECK_ARG(ctx, (!bary6 && !rxy) || (B0 && B1));
    FB_CHECK_ARG(ctx, !rxy || (v0 && v1));
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(8, K / 8192));
    std::vector<int64_t> bad((size_t)T, -1);
    auto work = [&](int t) {
        const int64_t lo = K * t / T, hi = K * (t + 1) / T;
        for (int64_t k = lo; k < hi; ++k) {
            int64_t a = tid0[k], b = tid1[k];
            if (a < 0) a += T0;                            // numpy's wrap-around: a match outside its mesh (tid -1) carries weight 0
            if (b < 0) b += T1;
            if (a < 0 || a >= T0 || b < 0 || b >= T1) { bad[t] = k; return; }
            const int32_t* ta = tri0 + 3 * a;
            const int32_t* tb = tri1 + 3 * b;
            int32_t* n6 = nodes6 + 6 * k;
            for (int c = 0; c < 3; ++c) {
                n6[c] = voff0 < 0 ? -1 : (int32_t)(ta[c] + voff0);
                n6[3 + c] = voff1 < 0 ? -1 : (int32_t)(tb[c] + voff1);
            }
            if (bary6) {
                double* b6 = bary6 + 6 * k;
                for (int c = 0; c < 3; ++c) { b6[c] = B0[3 * k + c]; b6[3 + c] = -B1[3 * k + c]; }
            }
            if (rxy) {
                for (int d = 0; d < 2; ++d) {
                    const double p0 = (v0[2 * (size_t)ta[0] + d] * B0[3 * k] + v0[2 * (size_t)ta[1] + d] * B0[3 * k + 1]) + v0[2 * (size_t)ta[2] + d] * B0[3 * k + 2];
                    const double p1 = (v1[2 * (size_t)tb[0] + d] * B1[3 * k] + v1[2 * (size_t)tb[1] + d] * B1[3 * k + 1]) + v1[2 * (size_t)tb[2] + d] * B1[3 * k + 2];
                    rxy[2 * k + d] = (p1 - p0) + (d ? oy : ox);
                }
            }
        }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    for (int t = 0; t < T; ++t)
        if (bad[t] >= 0) return fb_fail(ctx, FB_ERR_ARG, "fb_link_terms: match %lld names a triangle outside its mesh", (long long)bad[t]);
    return FB_OK;
}

int fb_sys_pattern(fb_ctx* ctx, fb_system* s, int64_t* browptr, int32_t* bcol) {
    FB_CHECK_ARG(ctx, s && s->finalized);
    if (browptr) for (size_t i = 0; i < s->browptr.size(); ++i) browptr[i] = s->browptr[i];
    if (bcol) std::copy(s->bcol.begin(), s->bcol.end(), bcol);
    return FB_OK;
}

struct StretchHost {             // host view of the stiffness functions of one assembly (fb_sys_assemble_mesh_stretch)
    const double* v_init = nullptr;
    const int32_t* tri_func = nullptr;
    int nfunc = 0;
    const int32_t* func_ptr = nullptr;
    const double* func_x = nullptr;
    const double* func_y = nullptr;
    const double* func_matmult = nullptr;
};

static int assemble_mesh_impl(fb_ctx* ctx, fb_system* s, int mesh_id, const double* v_shape, const double* v_cur, const float* tri_mult,
                              double nu, double soft, const int32_t* tri_model, const double* tri_nu, const float* tri_matmult,
                              bool accumulate = false, const StretchHost* sh = nullptr) {
    FB_CHECK_ARG(ctx, s && s->finalized && mesh_id >= 0 && mesh_id < (int)s->meshes.size() && v_shape);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    fb_mesh_blk& m = s->meshes[mesh_id];
    { const int rc_ = fb_copy_h2d(ctx, m.d_vshape, v_shape, sizeof(double2) * (size_t)m.V); if (rc_) return rc_; }
    m.h_v.assign(v_shape, v_shape + 2 * (size_t)m.V);
    if (v_cur) { const int rc_ = fb_copy_h2d(ctx, m.d_vcur, v_cur, sizeof(double2) * (size_t)m.V); if (rc_) return rc_; }
    if (tri_mult) { const int rc_ = fb_copy_h2d(ctx, m.d_mult, tri_mult, sizeof(float) * (size_t)m.T); if (rc_) return rc_; }
    if (tri_model) {
        FB_CHECK_ARG(ctx, tri_nu && tri_matmult);
        for (int t = 0; t < m.T; ++t) FB_CHECK_ARG(ctx, tri_model[t] >= 0 && tri_model[t] <= 2);
        int rc;
        if ((rc = upload(ctx, &m.d_model, tri_model, (size_t)m.T))) return rc;
        if ((rc = upload(ctx, &m.d_nu, tri_nu, (size_t)m.T))) return rc;
        if ((rc = upload(ctx, &m.d_matmult, tri_matmult, (size_t)m.T))) return rc;
    }
    StretchArgs sa;
    if (sh) {
        int rc;
        const size_t nk = (size_t)sh->func_ptr[sh->nfunc];
        if (m.ftab_cap < nk) { hipFree(m.d_fx); hipFree(m.d_fy); m.d_fx = m.d_fy = nullptr; m.ftab_cap = nk; }
        if (m.fcnt_cap < (size_t)sh->nfunc) { hipFree(m.d_fptr); hipFree(m.d_fmm); m.d_fptr = nullptr; m.d_fmm = nullptr; m.fcnt_cap = (size_t)sh->nfunc; }
        if ((rc = upload(ctx, &m.d_vinit, reinterpret_cast<const double2*>(sh->v_init), (size_t)m.V))) return rc;
        if ((rc = upload(ctx, &m.d_func, sh->tri_func, (size_t)m.T))) return rc;
        if ((rc = upload(ctx, &m.d_fptr, sh->func_ptr, (size_t)sh->nfunc + 1))) return rc;
        if ((rc = upload(ctx, &m.d_fx, sh->func_x, nk))) return rc;
        if ((rc = upload(ctx, &m.d_fy, sh->func_y, nk))) return rc;
        if ((rc = upload(ctx, &m.d_fmm, sh->func_matmult, (size_t)sh->nfunc))) return rc;
        // base ratio of the area stretch: summed |areas| of the linear triangles (all triangles if there is none), mesh.py:2952-2958
        const int g = std::min(256, std::max(1, fb_cdiv(m.T, kT)));
        if (!s->d_parts) FB_HIP(ctx, hipMalloc((void**)&s->d_parts, sizeof(double) * 2 * 1024));
        hipLaunchKernelGGL(area_sums_kernel, dim3(g), dim3(kT), 0, ctx->stream, m.T, m.d_tri, m.d_vinit, v_cur ? m.d_vcur : m.d_vshape,
                           tri_model ? m.d_model : (const int*)nullptr, m.d_func, s->d_parts);
        std::vector<double> hp(4 * (size_t)g);
        FB_HIP(ctx, hipMemcpyAsync(hp.data(), s->d_parts, sizeof(double) * 4 * g, hipMemcpyDeviceToHost, ctx->stream));
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double acc[4] = {0, 0, 0, 0};
        for (int i = 0; i < g; ++i) for (int k = 0; k < 4; ++k) acc[k] += hp[4 * i + k];
        sa.base = acc[1] > 0.0 ? acc[0] / acc[1] : acc[2] / acc[3];
        sa.vi = m.d_vinit; sa.func = m.d_func; sa.fptr = m.d_fptr; sa.fx = m.d_fx; sa.fy = m.d_fy; sa.fmm = m.d_fmm;
    }
    {
        FB_PROF(ctx, "fem_asm_stiffness");
        hipLaunchKernelGGL(asm_stiffness_kernel, dim3(fb_cdiv(m.V, kT)), dim3(kT), 0, ctx->stream, m.voff, m.V, m.d_tri, m.d_vtptr,
                           m.d_vtidx, m.d_vshape, v_cur ? m.d_vcur : (const double2*)nullptr, tri_mult ? m.d_mult : (const float*)nullptr,
                           (1.0 - nu) / 2.0, nu, soft, (float)soft, tri_model ? m.d_model : (const int*)nullptr,
                           tri_model ? m.d_nu : (const double*)nullptr, tri_model ? m.d_matmult : (const float*)nullptr,
                           s->M->d.rowptr, s->M->d.col, accumulate ? s->d_Kscr : s->d_K, accumulate ? s->d_sscr : s->d_stress, sa);
        if (accumulate)
            hipLaunchKernelGGL(add_rows_kernel, dim3(fb_cdiv(m.V, kT)), dim3(kT), 0, ctx->stream, m.voff, m.V, s->M->d.rowptr, s->d_Kscr, s->d_sscr,
                               s->d_K, s->d_stress);
    }
    FB_HIP(ctx, hipGetLastError());
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FB_OK;
}

int fb_sys_assemble_mesh(fb_ctx* ctx, fb_system* s, int mesh_id, const double* v_shape, const double* v_cur, const float* tri_mult,
                         double nu, double soft) {
    FB_LOCK(ctx);
    return assemble_mesh_impl(ctx, s, mesh_id, v_shape, v_cur, tri_mult, nu, soft, nullptr, nullptr, nullptr);
}

// As fb_sys_assemble_mesh, but the mesh ADDS its stiffness rows and stress to what the rows already hold: meshes that were
// entered at the same vertex offset share their degrees of freedom (`groupings` of SLM.optimize_linear, optimizer.py:1378-1415:
// T K T^T sums the members of a group).  Assemble the first member with fb_sys_assemble_mesh, the others with this.
int fb_sys_assemble_mesh_add(fb_ctx* ctx, fb_system* s, int mesh_id, const double* v_shape, const double* v_cur, const float* tri_mult,
                             double nu, double soft) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized);
    if (!s->d_Kscr) {
        FB_HIP(ctx, hipMalloc((void**)&s->d_Kscr, sizeof(double) * 4 * (size_t)s->M->nnzb));
        FB_HIP(ctx, hipMalloc((void**)&s->d_sscr, sizeof(float2) * (size_t)s->nv));
    }
    return assemble_mesh_impl(ctx, s, mesh_id, v_shape, v_cur, tri_mult, nu, soft, nullptr, nullptr, nullptr, true);
}

int fb_sys_assemble_mesh_materials(fb_ctx* ctx, fb_system* s, int mesh_id, const double* v_shape, const double* v_cur,
                                   const float* tri_mult, const int32_t* tri_model, const double* tri_nu, const float* tri_matmult,
                                   double soft) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, tri_model && tri_nu && tri_matmult);
    return assemble_mesh_impl(ctx, s, mesh_id, v_shape, v_cur, tri_mult, 0.0, soft, tri_model, tri_nu, tri_matmult);
}

int fb_sys_assemble_mesh_stretch(fb_ctx* ctx, fb_system* s, int mesh_id, const double* v_shape, const double* v_cur, const double* v_init,
                                 const float* tri_mult, const int32_t* tri_model, const double* tri_nu, const float* tri_matmult,
                                 const int32_t* tri_func, int nfunc, const int32_t* func_ptr, const double* func_x, const double* func_y,
                                 const double* func_matmult, double soft) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && mesh_id >= 0 && mesh_id < (int)s->meshes.size());
    FB_CHECK_ARG(ctx, tri_model && tri_nu && tri_matmult && v_init && tri_func && nfunc > 0 && func_ptr && func_x && func_y && func_matmult);
    FB_CHECK_ARG(ctx, func_ptr[0] == 0);
    for (int k = 0; k < nfunc; ++k) {
        FB_CHECK_ARG(ctx, func_ptr[k + 1] - func_ptr[k] >= 2);                   // interp1d needs two knots
        for (int i = func_ptr[k] + 1; i < func_ptr[k + 1]; ++i) FB_CHECK_ARG(ctx, func_x[i] > func_x[i - 1]);
    }
    const int T = s->meshes[mesh_id].T;
    for (int t = 0; t < T; ++t) FB_CHECK_ARG(ctx, tri_func[t] >= -1 && tri_func[t] < nfunc);
    StretchHost sh;
    sh.v_init = v_init; sh.tri_func = tri_func; sh.nfunc = nfunc; sh.func_ptr = func_ptr; sh.func_x = func_x; sh.func_y = func_y;
    sh.func_matmult = func_matmult;
    return assemble_mesh_impl(ctx, s, mesh_id, v_shape, v_cur, tri_mult, 0.0, soft, tri_model, tri_nu, tri_matmult, false, &sh);
}

int fb_sys_assemble_links(fb_ctx* ctx, fb_system* s, const double* bary6, const float* w, const double* rxy) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && (s->nlink == 0 || (bary6 && w && rxy)));
    FB_HIP(ctx, hipSetDevice(ctx->device));
    if (s->nlink) {
        int rc_ = fb_copy_h2d(ctx, s->d_bary, bary6, sizeof(double) * 6 * (size_t)s->nlink);
        if (!rc_) rc_ = fb_copy_h2d(ctx, s->d_w, w, sizeof(float) * (size_t)s->nlink);
        if (!rc_) rc_ = fb_copy_h2d(ctx, s->d_rxy, rxy, sizeof(double2) * (size_t)s->nlink);
        if (rc_) return rc_;
    }
    {
        FB_PROF(ctx, "fem_asm_links");
        hipLaunchKernelGGL(asm_links_kernel, dim3(fb_cdiv(s->nv, kT)), dim3(kT), 0, ctx->stream, s->nv, s->d_vmptr, s->d_vmidx, s->d_nodes,
                           s->d_bary, s->d_w, s->d_rxy, s->M->d.rowptr, s->M->d.col, s->d_Cacc, s->d_C, s->d_rhs);
    }
    FB_HIP(ctx, hipGetLastError());
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FB_OK;
}

int fb_sys_lambda(fb_ctx* ctx, fb_system* s, double stiffness_lambda, double crosslink_lambda, double* sl_out, double* cl_out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && sl_out && cl_out);
    double sl = stiffness_lambda, cl = crosslink_lambda;
    if (sl < 0 || cl < 0) {                                          // optimizer.py:1575-1589
        const int g = std::min(1024, std::max(1, fb_cdiv(s->nv, kT)));
        hipLaunchKernelGGL(lambda_trace_kernel, dim3(g), dim3(kT), 0, ctx->stream, s->nv, s->M->d.rowptr, s->M->d.col, s->d_K, s->d_C, s->d_parts);
        std::vector<double> hp(2 * (size_t)g);
        FB_HIP(ctx, hipMemcpyAsync(hp.data(), s->d_parts, sizeof(double) * 2 * g, hipMemcpyDeviceToHost, ctx->stream));
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double tc = 0.0, tk = 0.0;
        for (int i = 0; i < g; ++i) { tc += hp[2 * i]; tk += hp[2 * i + 1]; }
        const double ratio = std::fabs(sl / cl);
        sl = (tc == 0.0) ? 0.0 : std::fabs(ratio * tc / tk);
        cl = 1.0;
    }
    *sl_out = sl; *cl_out = cl;
    return FB_OK;
}

// Block-diagonal systems made of `ngroups` equal vertex ranges (one independent SLM per range, e.g. one tile pair
// each): relative_lambda_trace (optimizer.py:1573-1590) is evaluated per range and A = ls[g] K + lc C,
// b = lc rhs - ls[g] stress is formed with the range's own lambda.
int fb_sys_form_groups(fb_ctx* ctx, fb_system* s, int ngroups, double stiffness_lambda, double crosslink_lambda, double* ls_out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && ngroups > 0 && s->nv % ngroups == 0);
    const int gs = s->nv / ngroups;
    if (!s->d_glambda || s->glambda_cap < ngroups) {
        hipFree(s->d_glambda);
        FB_HIP(ctx, hipMalloc((void**)&s->d_glambda, sizeof(double) * (size_t)ngroups));
        s->glambda_cap = ngroups;
    }
    hipLaunchKernelGGL(group_lambda_kernel, dim3(ngroups), dim3(kT), 0, ctx->stream, gs, s->M->d.rowptr, s->M->d.col, s->d_K, s->d_C,
                       stiffness_lambda, crosslink_lambda, s->d_glambda);
    const double cl = (stiffness_lambda < 0 || crosslink_lambda < 0) ? 1.0 : crosslink_lambda;
    hipLaunchKernelGGL(form_groups_kernel, dim3(fb_cdiv(s->nv, kT)), dim3(kT), 0, ctx->stream, s->nv, gs, s->M->d.rowptr, s->d_K, s->d_Cacc,
                       s->d_rhs, s->d_stress, s->d_glambda, cl, s->M->d.val, s->M->b);
    FB_HIP(ctx, hipGetLastError());
    if (ls_out) {
        FB_HIP(ctx, hipMemcpyAsync(ls_out, s->d_glambda, sizeof(double) * (size_t)ngroups, hipMemcpyDeviceToHost, ctx->stream));
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FB_OK;
}

int fb_sys_form(fb_ctx* ctx, fb_system* s, double sl, double cl) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized);
    FB_PROF(ctx, "fem_form_system");
    const int g = (int)std::min<int64_t>(4096, std::max<int64_t>(1, (s->M->nnzb + kT - 1) / kT));
    hipLaunchKernelGGL(form_system_kernel, dim3(g), dim3(kT), 0, ctx->stream, s->nv, s->M->nnzb, s->d_K, s->d_Cacc, s->d_rhs, s->d_stress, sl, cl,
                       s->M->d.val, s->M->b);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

// Chronopoulos-Gear PCG (the fb_cgcg_* kernels) with one multigrid V-cycle as the preconditioner; x0 in M->x, result in M->x.
// The stopping rule is optimizer.solve's (optimizer.py:1993-1996) on the TRUE residual, re-evaluated at the end of a leg.
static int sys_solve_mg(fb_ctx* ctx, fb_system* s, double rtol, double atol, int maxiter, int* iters, double* relres) {
    fb_bsr* M = s->M;
    const int64_t n = 2 * (int64_t)s->nv;
    fb_mg* mg = nullptr;
    int rc = mg_setup(ctx, s, &mg);
    if (rc) return rc;
    struct Guard { fb_ctx* c; fb_mg* m; ~Guard() { mg_destroy(c, m); } } guard{ctx, mg};
    double* x = reinterpret_cast<double*>(M->x); double* r = reinterpret_cast<double*>(M->r); double* p = reinterpret_cast<double*>(M->p0);
    double* sv = reinterpret_cast<double*>(M->p1); double* w = reinterpret_cast<double*>(M->Ap); double* b = reinterpret_cast<double*>(M->b);
    double* state = M->parts; double* t3 = state + 8; double* scratch = t3 + 8;           // parts holds 7 * kNP doubles: plenty
    MgLevel& L0 = mg->L[0];
    double* own_r = L0.d_r;
    L0.d_r = r;                                              // the cycle reads the residual where the iteration keeps it
    struct Restore { MgLevel& l; double* p; ~Restore() { l.d_r = p; } } restore{L0, own_r};
    const int g = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (s->nv + 255) / 256));
    auto residual = [&](double* rr_out) -> int {             // r = b - A x, rr = ||r||^2
        int e = fb_bsr_spmv_dev(ctx, M, M->x, M->Ap);
        if (e) return e;
        hipLaunchKernelGGL((mg_fine_axpy_kernel<1>), dim3(g), dim3(256), 0, ctx->stream, s->nv, (const double*)nullptr, reinterpret_cast<const double2*>(b), M->Ap, nullptr, M->r, 0.0);
        if ((e = fb_cgcg_dots_dev(ctx, n, r, r, r, scratch, t3))) return e;
        double h[3];
        FB_HIP(ctx, hipMemcpyAsync(h, t3, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *rr_out = h[2];
        return FB_OK;
    };
    // ||b||
    double bb = 0.0;
    if ((rc = fb_cgcg_dots_dev(ctx, n, b, b, b, scratch, t3))) return rc;
    { double h[3]; FB_HIP(ctx, hipMemcpyAsync(h, t3, sizeof(h), hipMemcpyDeviceToHost, ctx->stream)); FB_HIP(ctx, hipStreamSynchronize(ctx->stream)); bb = h[2]; }
    if (iters) *iters = 0;
    if (relres) *relres = 0.0;
    if (bb == 0.0 || maxiter == 0) { FB_HIP(ctx, hipMemsetAsync(M->x, 0, sizeof(double2) * (size_t)s->nv, ctx->stream)); return FB_OK; }
    const double bnorm = std::sqrt(bb);
    double tol = rtol;
    if (atol > 0.0) tol = std::max(tol, atol / bnorm);
    const int limit = maxiter > 0 ? maxiter : 100 * 1000;
    int total = 0;
    double rel = 1.0;
    for (int leg = 0; leg < 6; ++leg) {
        double rr = 0.0;
        if ((rc = residual(&rr))) return rc;
        rel = std::sqrt(rr) / bnorm;
        if (relres) *relres = rel;
        if (rel <= tol || total >= limit) break;
        FB_HIP(ctx, hipMemsetAsync(p, 0, sizeof(double) * (size_t)n, ctx->stream));
        FB_HIP(ctx, hipMemsetAsync(sv, 0, sizeof(double) * (size_t)n, ctx->stream));
        FB_HIP(ctx, hipMemsetAsync(state, 0, sizeof(double) * 8, ctx->stream));
        auto precondition_and_dots = [&](int first) -> int {
            int e = mg_vcycle(ctx, s, mg, 0);                // u = V(r) -> L0.d_e
            if (e) return e;
            if ((e = fb_bsr_spmv_dev(ctx, M, reinterpret_cast<const double2*>(L0.d_e), M->Ap))) return e;
            if ((e = fb_cgcg_dots_dev(ctx, n, r, L0.d_e, w, scratch, t3))) return e;
            return fb_cgcg_scalars_dev(ctx, t3, state, first);
        };
        if ((rc = precondition_and_dots(1))) return rc;
        const double target = 0.81 * tol * tol * bb;          // the recurrence residual a little below the target, like fb_bsr_pcg_dev
        int it = 0;
        double dropped_seen = 0.0;
        while (total + it < limit) {
            const int batch = std::min(8, limit - total - it);
            for (int k = 0; k < batch; ++k) {
                if ((rc = fb_cgcg_update_dev(ctx, n, state, nullptr, x, r, L0.d_e, w, p, sv))) return rc;
                if ((rc = precondition_and_dots(0))) return rc;
            }
            it += batch;
            double hs[8];
            FB_HIP(ctx, hipMemcpyAsync(hs, state, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
            FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (!(hs[3] == hs[3])) return fb_fail(ctx, FB_ERR_BREAKDOWN, "multigrid PCG: the residual is not finite after %d iterations", total + it);
            if (hs[3] <= target) break;
            if (hs[4] - dropped_seen >= (double)batch) return fb_fail(ctx, FB_ERR_BREAKDOWN, "multigrid PCG: every step of the last %d was dropped (p^T A p <= 0): the system is not positive definite", batch);
            dropped_seen = hs[4];
        }
        total += it;
        if (iters) *iters = total;
    }
    if (iters) *iters = total;
    return rel <= tol ? FB_OK : FB_ERR_NOCONV;
}

int fb_sys_solve(fb_ctx* ctx, fb_system* s, double* x, int use_x0, double rtol, double atol, int maxiter, int precond, int* iters,
                 double* relres) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && x);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    if (use_x0) { const int rc_ = fb_copy_h2d(ctx, s->M->x, x, sizeof(double2) * (size_t)s->nv); if (rc_) return rc_; }
    else FB_HIP(ctx, hipMemsetAsync(s->M->x, 0, sizeof(double2) * (size_t)s->nv, ctx->stream));
    int rc;
    if (precond == 3 && maxiter == 0) precond = 1;          // no iterations asked for: the plain path returns the start vector's zeros like every other preconditioner (optimizer.py:1974)
    if (precond == 3) {
        // 'auto': the Jacobi-PCG for as many iterations as a multigrid solve of this size is expected to cost in all (set-up +
        // cycles: ~30 ms; an iteration: 12 us + 0.085 ns per vertex, measured on MI355X), then -- from the iterate reached --
        // the multigrid-PCG; whatever stops the hierarchy (set-up, breakdown, stall) hands the iterate back to the Jacobi-PCG.
        // Never more than about twice the cost of the better of the two; a solve that converges inside the budget is the
        // plain Jacobi-PCG bit for bit.
        static const double budget_ms = [] { const char* e = getenv("FEABAS_HIP_AUTO_SWITCH_MS"); return e ? std::max(0.0, atof(e)) : 30.0; }();
        const int budget = (int)std::min(4000.0, std::max(100.0, budget_ms * 1e-3 / (12e-6 + 0.085e-9 * (double)s->nv)));
        const int cap1 = maxiter > 0 ? std::min(maxiter, budget) : budget;
        int it1 = 0, it2 = 0, it3 = 0;
        double rr = 0.0;
        if ((rc = fb_bsr_setup_jacobi(ctx, s->M, 1))) return rc;
        // ... or fewer: from 128 iterations on the Jacobi leg projects what it still needs from the decay of its residual over the
        // last 64 iterations and ends once the projection passes 1.5 x the budget (CG speeds up as it goes: early projections are high)
        s->M->probe_limit = budget + budget / 2;
        rc = fb_bsr_pcg_dev(ctx, s->M, rtol, atol, cap1, 0, &it1, &rr);
        s->M->probe_limit = 0;
        if (rc) return rc;
        const double bn = s->M->last_bnorm;
        const double tol = std::max(rtol, (atol > 0.0 && bn > 0.0) ? atol / bn : 0.0);
        auto left = [&](int used) { return maxiter > 0 ? maxiter - used : -1; };
        if (rr > tol && (maxiter <= 0 || it1 < maxiter)) {
            double rr_mg = rr;
            rc = fb_bsr_setup_jacobi(ctx, s->M, 2);
            if (!rc) rc = sys_solve_mg(ctx, s, rtol, atol, left(it1), &it2, &rr_mg);
            // what the hierarchy may hand back to the Jacobi leg: a stall, a breakdown, a set-up it refuses (FB_ERR_ARG: nothing to
            // aggregate).  A device or allocation error is an error of the solve.
            if (rc == FB_ERR_HIP || rc == FB_ERR_NOMEM) return rc;
            if (!rc || rc == FB_ERR_NOCONV) rr = rr_mg;       // (a failed call may not have written its residual)
            if (rc || rr > tol) {
                // M->x holds the last iterate the hierarchy reached -- or, when only its SET-UP refused (FB_ERR_ARG: the solve never
                // touched x), the iterate of the first Jacobi leg, which is kept; after a breakdown (or a residual that is not a
                // number) the Jacobi leg restarts from zero
                if (rc == FB_ERR_BREAKDOWN || !(rr == rr)) FB_HIP(ctx, hipMemsetAsync(s->M->x, 0, sizeof(double2) * (size_t)s->nv, ctx->stream));
                else if (rc && rc != FB_ERR_NOCONV && rc != FB_ERR_ARG) return rc;
                const int rem = left(it1 + it2);
                if (maxiter <= 0 || rem > 0) {
                    if ((rc = fb_bsr_setup_jacobi(ctx, s->M, 1))) return rc;
                    rc = fb_bsr_pcg_dev(ctx, s->M, rtol, atol, rem, 0, &it3, &rr);
                    if (rc && rc != FB_ERR_NOCONV) return rc;
                } else rc = FB_OK;
            }
        }
        if (iters) *iters = it1 + it2 + it3;
        if (relres) *relres = rr;
        if (!rc && rr > tol && maxiter < 0) rc = fb_fail(ctx, FB_ERR_NOCONV, "PCG stopped at relative residual %.3e > %.3e after %d iterations", rr, tol, it1 + it2 + it3);
    } else {
        rc = fb_bsr_setup_jacobi(ctx, s->M, precond);
        if (rc) return rc;
        if (precond == 2) rc = sys_solve_mg(ctx, s, rtol, atol, maxiter, iters, relres);
        else rc = fb_bsr_pcg_dev(ctx, s->M, rtol, atol, maxiter, 0, iters, relres);
    }
    if (rc && rc != FB_ERR_NOCONV) return rc;
    { const int rc_ = fb_copy_d2h(ctx, x, s->M->x, sizeof(double2) * (size_t)s->nv); if (rc_) return rc_; }
    return rc;
}

// Batched in-matcher relaxations: the system is block diagonal with `ngroups` equal vertex ranges (fb_sys_form_groups);
// every range is solved to its own tolerance by one workgroup (fb_bsr_pcg_groups).  x: host [2 nv].
// iters_max / relres_max: worst range.  Returns FB_ERR_NOCONV when a range stopped above its tolerance.
int fb_sys_solve_groups(fb_ctx* ctx, fb_system* s, int ngroups, double* x, double rtol, double atol, int maxiter, int precond, int* iters_max,
                        double* relres_max) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && x && ngroups > 0 && s->nv % ngroups == 0);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    if (s->gstat_cap < ngroups) {
        hipFree(s->d_gstat);
        FB_HIP(ctx, hipMalloc((void**)&s->d_gstat, 16 * (size_t)ngroups));
        s->gstat_cap = ngroups;
    }
    double* d_rel = reinterpret_cast<double*>(s->d_gstat);
    int* d_it = reinterpret_cast<int*>(d_rel + ngroups);
    int* d_fl = d_it + ngroups;
    int rc = fb_bsr_pcg_groups(ctx, s->M, ngroups, rtol, atol, maxiter, precond, d_it, d_rel, d_fl);
    if (rc) return rc;
    std::vector<char> hs(16 * (size_t)ngroups);
    { const int rc_ = fb_copy_d2h(ctx, x, s->M->x, sizeof(double2) * (size_t)s->nv); if (rc_) return rc_; }
    FB_HIP(ctx, hipMemcpyAsync(hs.data(), s->d_gstat, hs.size(), hipMemcpyDeviceToHost, ctx->stream));
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double* hr = reinterpret_cast<const double*>(hs.data());
    const int* hi = reinterpret_cast<const int*>(hr + ngroups);
    const int* hf = hi + ngroups;
    int mi = 0, bad = 0; double mr = 0.0;
    for (int g = 0; g < ngroups; ++g) {
        mi = std::max(mi, hi[g]); mr = std::max(mr, hr[g]);
        if (hf[g] == 2) return fb_fail(ctx, FB_ERR_BREAKDOWN, "fb_sys_solve_groups: negative curvature in range %d", g);
        if (hf[g] == 0 && hi[g] > 0) bad = 1;          // iteration cap reached
    }
    if (iters_max) *iters_max = mi;
    if (relres_max) *relres_max = mr;
    return bad ? FB_ERR_NOCONV : FB_OK;
}

// Elastic energy x^T K x of every equal vertex range of a block-diagonal batch (K = the stiffness of the last
// fb_sys_assemble_mesh): the Es / Es0 terms of the matcher's strain estimate (matcher.py:764-777).  x: host [2 nv].
int fb_sys_group_energy(fb_ctx* ctx, fb_system* s, int ngroups, const double* x, double* energy) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && x && energy && ngroups > 0 && s->nv % ngroups == 0);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    if (s->gstat_cap < ngroups) {
        hipFree(s->d_gstat);
        FB_HIP(ctx, hipMalloc((void**)&s->d_gstat, 16 * (size_t)ngroups));
        s->gstat_cap = ngroups;
    }
    // M->z is free outside a solve: staging area for x
    { const int rc_ = fb_copy_h2d(ctx, s->M->z, x, sizeof(double2) * (size_t)s->nv); if (rc_) return rc_; }
    hipLaunchKernelGGL(group_energy_kernel, dim3(ngroups), dim3(kT), 0, ctx->stream, s->nv / ngroups, s->M->d.rowptr, s->M->d.col, s->d_K, s->M->z,
                       reinterpret_cast<double*>(s->d_gstat));
    FB_HIP(ctx, hipGetLastError());
    FB_HIP(ctx, hipMemcpyAsync(energy, s->d_gstat, sizeof(double) * (size_t)ngroups, hipMemcpyDeviceToHost, ctx->stream));
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FB_OK;
}

int fb_sys_solve_fixed(fb_ctx* ctx, fb_system* s, int iters, double* relres) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && iters > 0);
    FB_HIP(ctx, hipMemsetAsync(s->M->x, 0, sizeof(double2) * (size_t)s->nv, ctx->stream));
    int rc = fb_bsr_setup_jacobi(ctx, s->M, 1);
    if (rc) return rc;
    int done = 0;
    return fb_bsr_pcg_dev(ctx, s->M, 0.0, 0.0, 0, iters, &done, relres);
}

// which: 0 K [nnzb][4] f64, 1 C [nnzb] f32, 2 rhs [2nv] f64, 3 stress [2nv] f32, 4 A [nnzb][4] f64, 5 b [2nv] f64
int fb_sys_get(fb_ctx* ctx, fb_system* s, int which, void* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && out && which >= 0 && which <= 5);
    const size_t nnzb = (size_t)s->M->nnzb, nv = (size_t)s->nv;
    const void* src = nullptr;
    size_t bytes = 0;
    switch (which) {
        case 0: src = s->d_K; bytes = sizeof(double) * 4 * nnzb; break;
        case 1: src = s->d_C; bytes = sizeof(float) * nnzb; break;
        case 2: src = s->d_rhs; bytes = sizeof(double2) * nv; break;
        case 3: src = s->d_stress; bytes = sizeof(float2) * nv; break;
        case 4: src = s->M->d.val; bytes = sizeof(double) * 4 * nnzb; break;
        default: src = s->M->b; bytes = sizeof(double2) * nv; break;
    }
    return fb_copy_d2h(ctx, out, src, bytes);
}

int fb_sys_info(fb_ctx* ctx, fb_system* s, int64_t* nv, int64_t* nnzb, int64_t* nlink) {
    FB_CHECK_ARG(ctx, s != nullptr);
    if (nv) *nv = s->nv;
    if (nnzb) *nnzb = s->M ? s->M->nnzb : 0;
    if (nlink) *nlink = s->nlink;
    return FB_OK;
}

// ---------------------------------------------------------------------------------- batched tile-pair stages
// The in-matcher FEM work of a batch of P tile pairs on ONE resident block-diagonal system: P copies of the
// cartesian mesh of matcher.py:354-359 (nx x ny nodes, cells (a b / c d) split into (a, b, d), (a, d, c)), mesh0
// locked, so a match touches only the three mesh1 vertices of its triangle.  Host glue in C++ (the per-match
// bookkeeping is O(K) and would otherwise serialise on the Python interpreter lock of the calling threads).
namespace {
// matches -> (nodes6, B1, bary6): Mesh.cart2bary (mesh.py:2191-2217) on the grid reduces to the cell coordinates
int pairs_build_links(fb_ctx* ctx, fb_system* s, int P, int nx, int ny, const double* xs, const double* ys, int64_t K, const int32_t* pid,
                      const double* xy1_init) {
    const int V = s->nv / P;
    if (nx < 2 || ny < 2 || nx * ny != V) return fb_fail(ctx, FB_ERR_ARG, "fb_pairs: grid %d x %d does not match %d vertices per pair", nx, ny, V);
    s->h_nodes6.resize(6 * (size_t)K);
    s->h_bary6.resize(6 * (size_t)K);
    s->h_B1.resize(3 * (size_t)K);
    for (int64_t k = 0; k < K; ++k) {
        const int p = pid[k];
        if (p < 0 || p >= P) return fb_fail(ctx, FB_ERR_ARG, "fb_pairs: pair id %d outside [0, %d)", p, P);
        const double qx = xy1_init[2 * k], qy = xy1_init[2 * k + 1];
        int i = (int)(std::upper_bound(xs, xs + nx, qx) - xs) - 1;       // np.searchsorted(side='right') - 1
        int j = (int)(std::upper_bound(ys, ys + ny, qy) - ys) - 1;
        i = std::min(std::max(i, 0), nx - 2);
        j = std::min(std::max(j, 0), ny - 2);
        const double u = (qx - xs[i]) / (xs[i + 1] - xs[i]), w = (qy - ys[j]) / (ys[j + 1] - ys[j]);
        const bool up = w > u;
        const int na = j * nx + i + p * V;
        int32_t* n6 = &s->h_nodes6[6 * (size_t)k];
        n6[0] = n6[1] = n6[2] = -1;
        n6[3] = na; n6[4] = up ? na + nx + 1 : na + 1; n6[5] = up ? na + nx : na + nx + 1;
        double* b1 = &s->h_B1[3 * (size_t)k];
        b1[0] = up ? 1.0 - w : 1.0 - u; b1[1] = up ? u : u - w; b1[2] = up ? w - u : w;
        double* b6 = &s->h_bary6[6 * (size_t)k];
        b6[0] = 1.0; b6[1] = 0.0; b6[2] = 0.0; b6[3] = -b1[0]; b6[4] = -b1[1]; b6[5] = -b1[2];
    }
    return sys_update_links(ctx, s, K, s->h_nodes6.data(), 1);     // the three vertices of one grid triangle: in the pattern by construction
}
}  // namespace

namespace {
// shared tail of fb_pairs_relax / fb_pairs_relax_bary: the links (h_nodes6, h_B1, h_bary6) are resident, h_dxy holds the
// residual of every match before the solve.  Solve every pair, then the huber residue weights.
int pairs_relax_core(fb_ctx* ctx, fb_system* s, int P, int64_t K, const float* conf, double residue_len, double sample_err,
                     double stiffness_lambda, double rtol, float* rw, double* x_out, int* iters, double* relres,
                     const double* sample_err_each = nullptr, int residue_mode = 0) {
    int rc;
    if ((rc = fb_sys_assemble_links(ctx, s, s->h_bary6.data(), conf, s->h_dxy.data()))) return rc;
    if ((rc = fb_sys_form_groups(ctx, s, P, stiffness_lambda, -1.0, nullptr))) return rc;
    s->h_x.resize(2 * (size_t)s->nv);
    const int V = s->nv / P;
    rc = fb_sys_solve_groups(ctx, s, P, s->h_x.data(), rtol, 0.0, 20 * V, 1, iters, relres);
    if (rc && rc != FB_ERR_NOCONV) return rc;
    for (int64_t k = 0; k < K; ++k) {
        const int32_t* n6 = &s->h_nodes6[6 * (size_t)k];
        const double* b1 = &s->h_B1[3 * (size_t)k];
        double ux = 0.0, uy = 0.0;
        for (int a = 0; a < 3; ++a) { ux += s->h_x[2 * (size_t)n6[3 + a]] * b1[a]; uy += s->h_x[2 * (size_t)n6[3 + a] + 1] * b1[a]; }
        const double rx = s->h_dxy[2 * k] + ux, ry = s->h_dxy[2 * k + 1] + uy;
        const double se = sample_err_each ? sample_err_each[k] : sample_err;
        const double d2 = rx * rx + ry * ry - se * se;
        const double dis = std::sqrt(d2 > 0.0 ? d2 : 0.0);
        // huber: L / max(dis, L) (optimizer.py:203-205); threshold: dis <= L (optimizer.py:198-200)
        rw[k] = residue_mode == 1 ? (dis <= residue_len ? 1.0f : 0.0f) : (float)(residue_len / std::max(dis, residue_len));
    }
    if (x_out) std::copy(s->h_x.begin(), s->h_x.end(), x_out);
    return FB_OK;
}
}  // namespace

// matcher.py:725-737 for a batch: relax every pair's mesh1 against its matches (optimize_linear to rtol), then the
// huber residue weight L / max(sqrt(max(|r|^2 - sample_err^2, 0)), L) of every match (optimizer.py:174-205).
// xy0_mov: mesh0 points in the MOVING gear, xy1_init: mesh1 points in its INITIAL gear, t1 [P][2]: mesh1 offsets.
int fb_pairs_relax(fb_ctx* ctx, fb_system* s, int P, int nx, int ny, const double* xs, const double* ys, int64_t K, const int32_t* pid,
                   const double* xy0_mov, const double* xy1_init, const double* t1, const float* conf, double residue_len, int residue_mode,
                   double sample_err, double stiffness_lambda, double rtol, float* rw, double* x_out, int* iters, double* relres) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && P > 0 && s->nv % P == 0 && K > 0 && pid && xy0_mov && xy1_init && t1 && conf && rw);
    int rc = pairs_build_links(ctx, s, P, nx, ny, xs, ys, K, pid, xy1_init);
    if (rc) return rc;
    s->h_dxy.resize(2 * (size_t)K);
    for (int64_t k = 0; k < K; ++k) {                       // Link.dxy (optimizer.py:248-255) in the MOVING gears
        const int p = pid[k];
        s->h_dxy[2 * k] = (xy1_init[2 * k] + t1[2 * p]) - xy0_mov[2 * k];
        s->h_dxy[2 * k + 1] = (xy1_init[2 * k + 1] + t1[2 * p + 1]) - xy0_mov[2 * k + 1];
    }
    return pairs_relax_core(ctx, s, P, K, conf, residue_len, sample_err, stiffness_lambda, rtol, rw, x_out, iters, relres, nullptr, residue_mode);
}

// The same relaxation for matches that were located in a DEFORMED mesh1 (Link.from_coordinates on the MOVING gear of a
// mesh that an earlier round relaxed, matcher.py:717; mesh.py:2191-2217): the caller hands over the three mesh1 vertices
// (ids inside the union mesh) and barycentric coordinates of every match and dxy0 = the link residual with mesh1 at its
// FIXED gear (B1 . v_fixed - xy0).  The unknown is the TOTAL displacement from the FIXED gear, so that the elastic
// energy of the earlier deformation (the `stress` term of optimizer.py:1417-1418) is inside the system: the minimiser
// is the reference's MOVING gear after optimize_linear, whatever the starting field.
int fb_pairs_relax_bary(fb_ctx* ctx, fb_system* s, int P, int64_t K, const int32_t* nodes3, const double* B1, const double* dxy0,
                        const float* conf, double residue_len, int residue_mode, double sample_err, const double* sample_err_each,
                        double stiffness_lambda, double rtol, float* rw, double* x_out, int* iters, double* relres) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && P > 0 && s->nv % P == 0 && K > 0 && nodes3 && B1 && dxy0 && conf && rw);
    s->h_nodes6.resize(6 * (size_t)K);
    s->h_bary6.resize(6 * (size_t)K);
    s->h_B1.assign(B1, B1 + 3 * (size_t)K);
    s->h_dxy.assign(dxy0, dxy0 + 2 * (size_t)K);
    for (int64_t k = 0; k < K; ++k) {
        int32_t* n6 = &s->h_nodes6[6 * (size_t)k];
        double* b6 = &s->h_bary6[6 * (size_t)k];
        n6[0] = n6[1] = n6[2] = -1;
        b6[0] = 1.0; b6[1] = 0.0; b6[2] = 0.0;
        for (int a = 0; a < 3; ++a) {
            const int32_t v = nodes3[3 * k + a];
            if (v < 0 || v >= s->nv) return fb_fail(ctx, FB_ERR_ARG, "fb_pairs_relax_bary: vertex %d outside [0, %d)", v, s->nv);
            n6[3 + a] = v;
            b6[3 + a] = -B1[3 * k + a];
        }
    }
    int rc = sys_update_links(ctx, s, K, s->h_nodes6.data(), ctx->trusted_links);
    if (rc) return rc;
    return pairs_relax_core(ctx, s, P, K, conf, residue_len, sample_err, stiffness_lambda, rtol, rw, x_out, iters, relres, sample_err_each, residue_mode);
}

// matcher.py:752-777 for a batch, after the rigid initialisation R [P][3][3] (row vectors: v_fixed = v_initial R[:2,:2],
// offset R[2,:2]) of every pair's mesh1: relaxation to 1e-6 and strain = sqrt(Es / Es0).  The stiffness of a rotated
// mesh is Q K Q^T and the link terms are multiples of I2, so the system is solved with the resident K of the INITIAL
// shape and right-hand sides rotated back by Q^T; es0 = v^T K v of the centred INITIAL mesh (rotation free).
// links_loaded != 0: the links of the preceding fb_pairs_relax call (same K rows) are reused.
namespace {
// tail of fb_pairs_strain / fb_pairs_strain_bary: links resident, h_dxy = Q^T f of every match; es0 [P] or one value
int pairs_strain_core(fb_ctx* ctx, fb_system* s, int P, const std::vector<char>& has, const float* weight, double stiffness_lambda,
                      const double* es0, int es0_stride, double default_strain, double* strain, int* iters, double* relres) {
    int rc;
    if ((rc = fb_sys_assemble_links(ctx, s, s->h_bary6.data(), weight, s->h_dxy.data()))) return rc;
    if ((rc = fb_sys_form_groups(ctx, s, P, stiffness_lambda, -1.0, nullptr))) return rc;
    s->h_x.resize(2 * (size_t)s->nv);
    const int V = s->nv / P;
    rc = fb_sys_solve_groups(ctx, s, P, s->h_x.data(), 1e-6, 0.0, 20 * V, 1, iters, relres);
    if (rc && rc != FB_ERR_NOCONV) return rc;
    for (int p = 0; p < P; ++p) {                           // Mesh.set_field keeps the mean in the offset (mesh.py:2409-2413)
        double mx = 0.0, my = 0.0;
        double* x = &s->h_x[2 * (size_t)p * V];
        for (int v = 0; v < V; ++v) { mx += x[2 * v]; my += x[2 * v + 1]; }
        mx /= V; my /= V;
        for (int v = 0; v < V; ++v) { x[2 * v] -= mx; x[2 * v + 1] -= my; }
    }
    std::vector<double> es((size_t)P);
    if ((rc = fb_sys_group_energy(ctx, s, P, s->h_x.data(), es.data()))) return rc;
    for (int p = 0; p < P; ++p) strain[p] = has[p] ? std::sqrt(std::max(es[p], 0.0) / es0[(size_t)p * es0_stride]) : default_strain;
    return FB_OK;
}
}  // namespace

int fb_pairs_strain(fb_ctx* ctx, fb_system* s, int P, int nx, int ny, const double* xs, const double* ys, int64_t K, const int32_t* pid,
                    const double* xy0_fixed, const double* xy1_init, const float* weight, const double* R, double stiffness_lambda, double es0,
                    int links_loaded, double default_strain, double* strain, int* iters, double* relres) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && P > 0 && s->nv % P == 0 && K > 0 && pid && xy0_fixed && xy1_init && weight && R && strain && es0 > 0.0);
    int rc;
    if (!links_loaded || s->nlink != K || (int64_t)s->h_B1.size() != 3 * K) {
        if ((rc = pairs_build_links(ctx, s, P, nx, ny, xs, ys, K, pid, xy1_init))) return rc;
    }
    s->h_dxy.resize(2 * (size_t)K);
    std::vector<char> has((size_t)P, 0);
    for (int64_t k = 0; k < K; ++k) {
        const int p = pid[k];
        FB_CHECK_ARG(ctx, p >= 0 && p < P);
        has[p] = 1;
        const double* r = R + 9 * (size_t)p;                 // r[0] r[1] / r[3] r[4] = 2x2, r[6] r[7] = offset
        const double x1 = xy1_init[2 * k], y1 = xy1_init[2 * k + 1];
        const double fx = x1 * r[0] + y1 * r[3] + r[6] - xy0_fixed[2 * k];        // Mesh.set_affine on the match point, minus mesh0's
        const double fy = x1 * r[1] + y1 * r[4] + r[7] - xy0_fixed[2 * k + 1];
        s->h_dxy[2 * k] = fx * r[0] + fy * r[1];                                   // Q^T f
        s->h_dxy[2 * k + 1] = fx * r[3] + fy * r[4];
    }
    return pairs_strain_core(ctx, s, P, has, weight, stiffness_lambda, &es0, 0, default_strain, strain, iters, relres);
}

// fb_pairs_strain for pairs whose meshes share a topology but not a geometry (strips of unequal size): the caller locates
// the matches (nodes3, B1 in the INITIAL mesh of its pair) and gives Es0 per pair
int fb_pairs_strain_bary(fb_ctx* ctx, fb_system* s, int P, int64_t K, const int32_t* pid, const int32_t* nodes3, const double* B1,
                         const double* xy0_fixed, const double* xy1_init, const float* weight, const double* R, double stiffness_lambda,
                         const double* es0, double default_strain, double* strain, int* iters, double* relres) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, s && s->finalized && P > 0 && s->nv % P == 0 && K > 0 && pid && nodes3 && B1 && xy0_fixed && xy1_init && weight && R && strain && es0);
    s->h_nodes6.resize(6 * (size_t)K);
    s->h_bary6.resize(6 * (size_t)K);
    s->h_B1.assign(B1, B1 + 3 * (size_t)K);
    s->h_dxy.resize(2 * (size_t)K);
    std::vector<char> has((size_t)P, 0);
    for (int64_t k = 0; k < K; ++k) {
        const int p = pid[k];
        FB_CHECK_ARG(ctx, p >= 0 && p < P);
        has[p] = 1;
        int32_t* n6 = &s->h_nodes6[6 * (size_t)k];
        double* b6 = &s->h_bary6[6 * (size_t)k];
        n6[0] = n6[1] = n6[2] = -1;
        b6[0] = 1.0; b6[1] = 0.0; b6[2] = 0.0;
        for (int a = 0; a < 3; ++a) {
            const int32_t v = nodes3[3 * k + a];
            if (v < 0 || v >= s->nv) return fb_fail(ctx, FB_ERR_ARG, "fb_pairs_strain_bary: vertex %d outside [0, %d)", v, s->nv);
            n6[3 + a] = v;
            b6[3 + a] = -B1[3 * k + a];
        }
        const double* r = R + 9 * (size_t)p;
        const double x1 = xy1_init[2 * k], y1 = xy1_init[2 * k + 1];
        const double fx = x1 * r[0] + y1 * r[3] + r[6] - xy0_fixed[2 * k];
        const double fy = x1 * r[1] + y1 * r[4] + r[7] - xy0_fixed[2 * k + 1];
        s->h_dxy[2 * k] = fx * r[0] + fy * r[1];
        s->h_dxy[2 * k + 1] = fx * r[3] + fy * r[4];
    }
    int rc = sys_update_links(ctx, s, K, s->h_nodes6.data(), ctx->trusted_links);
    if (rc) return rc;
    for (int p = 0; p < P; ++p) FB_CHECK_ARG(ctx, !has[p] || es0[p] > 0.0);
    return pairs_strain_core(ctx, s, P, has, weight, stiffness_lambda, es0, 1, default_strain, strain, iters, relres);
}

#ifdef FB_TEST_HOOKS          // only in libfeabas_hip_test.so (include/feabas_hip_test.h)
#include "feabas_hip_test.h"
// test hook (host only, no context): one coarsening step of the multigrid set-up on a level given as host arrays -- the
// aggregates, the relative node positions and the coarse pattern mg_build_next uploads (mg_host_coarsen, the threaded host half).
// A first call with ccol == NULL sizes the coarse pattern (*cnnz).
int fb_debug_mg_coarsen(int n, int bs, const double* xy, const int32_t* comp, const int32_t* rowptr, const int32_t* col, double fine_scale,
                        int32_t* nc, double* cell, int32_t* agg, double* rel, double* cxy, int32_t* ccomp, int32_t* crowptr, int64_t* cnnz,
                        int32_t* ccol, int64_t ccol_cap, int32_t* maxc) {
    if (n <= 0 || (bs != 2 && bs != 3) || !xy || !comp || !rowptr || !col || !nc || !cnnz) return FB_ERR_ARG;
    MgLevel f, c;
    f.n = n; f.bs = bs;
    f.xy.assign(xy, xy + 2 * (size_t)n);
    f.comp.assign(comp, comp + n);
    f.rowptr.assign(rowptr, rowptr + n + 1);
    f.col.assign(col, col + rowptr[n]);
    f.nnzb = rowptr[n];
    std::vector<int> agg_v, aptr, aidx;
    std::vector<double2> rel_v;
    int mc = 0;
    mg_host_coarsen(f, c, bs, fine_scale, agg_v, aptr, aidx, rel_v, mc, [](const char*) {});
    *nc = f.nc; *cnnz = c.nnzb;
    if (cell) *cell = f.scale;
    if (maxc) *maxc = mc;
    if (agg) std::copy(agg_v.begin(), agg_v.end(), agg);
    if (rel) for (int i = 0; i < n; ++i) { rel[2 * (size_t)i] = rel_v[i].x; rel[2 * (size_t)i + 1] = rel_v[i].y; }
    if (ccol) {
        if (ccol_cap < c.nnzb || !crowptr || !cxy || !ccomp) return FB_ERR_ARG;
        std::copy(c.rowptr.begin(), c.rowptr.end(), crowptr);
        std::copy(c.col.begin(), c.col.end(), ccol);
        std::copy(c.xy.begin(), c.xy.end(), cxy);
        std::copy(c.comp.begin(), c.comp.end(), ccomp);
    }
    return FB_OK;
}
#endif

}  // extern "C"
