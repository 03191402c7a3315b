// FEM path, solver: 2x2-block CSR (one block row per mesh vertex, DoF order [x,y]) SpMV with
// coalesced value/index streams and an LDS row reduction, and a Jacobi-preconditioned CG
// whose iteration is two kernels with the dot products fused in.
// Replaces optimizer.solve (feabas/optimizer.py:1945-2080): same fixed point
// (||Ax-b|| <= max(rtol, atol/||b||) ||b|| on the symmetrised system, reference Jacobi
// M0 = 1/clip(diag, min(1, max/1000)), optimizer.py:1962-1966), not the reference's
// restarted-MINRES iteration path (SURVEY.md sec.7 "Hard parts").
#include "fb_solver.h"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kT = 256;          // threads per workgroup == block rows per row-chunk
constexpr int kCap = 3072;       // LDS capacity in blocks (16 B each = 48 KiB)
constexpr int kMaxWG1 = 1024;    // SpMV workgroups (each loops over row chunks)
constexpr int kWG2 = 256;        // vector-update workgroups (one per CU: 256 / 512 / 1024 give 51.1 / 52.8 / 56.6 us per iteration at 1e6 DoF -- every workgroup of both kernels sums their partials)
constexpr int kNP = 1024;        // partial-sum slots per array

__device__ __forceinline__ double block_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
    return t;
}

// every workgroup sums the same partial array in the same order -> identical scalar everywhere
__device__ __forceinline__ double sum_partials(const double* __restrict__ part, int n, double* sh) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += part[i];
    return block_sum(v, sh);
}

// K partial arrays reduced together: all loads are issued before the one pair of barriers (a prologue of K separate
// reductions costs K dependent round trips at the head of every PCG kernel).  Per array the summation order is that of
// sum_partials, so the scalars are bitwise the same.
template <int K>
__device__ __forceinline__ void sum_partials_k(const double* const* part, const int* n, double* out, double (*sh)[kT / 64]) {
    double v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        v[k] = 0.0;
        for (int i = threadIdx.x; i < n[k]; i += blockDim.x) v[k] += part[k][i];
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) sh[k][wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[k][w];
        out[k] = t;
    }
}

// MODE 0: y = A v.   MODE 1 (PCG step A): p_new = z + beta p_old, Ap = A p_new, partial p.Ap
// MODE 2: r = b - A v  (y := residual), partial r.r into part_out
template <int MODE>
__global__ __launch_bounds__(kT) void bsr_spmv_kernel(
    fb_bsr_dev A, const double2* __restrict__ v, const double2* __restrict__ p_old, double2* __restrict__ p_new,
    double2* __restrict__ y, const double2* __restrict__ bvec, double* __restrict__ part_out, double* __restrict__ part_pp,
    const double* __restrict__ part_rz_new, const double* __restrict__ part_rz_old, const double* __restrict__ part_rr,
    int nparts_in, fb_pcg_state* st, int iter) {
    __shared__ double2 contrib[kCap];
    __shared__ double sh[kT / 64];
    __shared__ double sh3[3][kT / 64];
    __shared__ int sflag;
    double beta = 0.0;
    if (MODE == 1) {
        if (threadIdx.x == 0) sflag = st->flag;
        double rr, rzn = 0.0, rzo = 1.0;
        if (iter > 0) {
            const double* ptrs[3] = {part_rr, part_rz_new, part_rz_old};
            const int ns[3] = {nparts_in, nparts_in, nparts_in};
            double o[3];
            sum_partials_k<3>(ptrs, ns, o, sh3);             // (its barriers also publish sflag)
            rr = o[0]; rzn = o[1]; rzo = o[2];
        } else {
            const double* ptrs[1] = {part_rr};
            const int ns[1] = {nparts_in};
            double o[1];
            sum_partials_k<1>(ptrs, ns, o, sh3);
            rr = o[0];
        }
        if (sflag) return;
        if (rr <= st->tol2bb) {
            if (threadIdx.x == 0 && blockIdx.x == 0) { st->flag = 1; st->iter = iter; st->rr = rr; }
            return;
        }
        if (threadIdx.x == 0 && blockIdx.x == 0) st->rr = rr;         // (the host reads the decay of the residual between its checks)
        if (iter > 0) beta = rzn / rzo;
    }
    double acc_dot = 0.0, acc_pp = 0.0;
    const int nchunks = (A.nb + kT - 1) / kT;
    // Workgroups are dealt round-robin over the 8 XCDs, each with its own L2: with chunk = blockIdx the neighbours of a row
    // chunk (whose vector entries its gathers touch) sit on other XCDs and every L2 ends up holding the whole vector.  The
    // workgroups of one XCD take a contiguous eighth of the chunks instead, so the gathers of a mesh-ordered matrix stay in
    // that XCD's L2.
    int c_first = blockIdx.x, c_stride = gridDim.x, c_limit = nchunks, c_base = 0;
    if (A.xcd_rows && (gridDim.x & 7) == 0 && nchunks >= 64) {
        const int per = (nchunks + 7) >> 3;
        c_base = (blockIdx.x & 7) * per; c_first = blockIdx.x >> 3; c_stride = gridDim.x >> 3; c_limit = min(per, nchunks - c_base);
    }
    for (int cc = c_first; cc < c_limit; cc += c_stride) {
        const int chunk = c_base + cc;
        const int r0 = chunk * kT;
        const int r1 = min(A.nb, r0 + kT);
        const int b0 = A.rowptr[r0], b1 = A.rowptr[r1];
        const int row = r0 + threadIdx.x;
        int lo = 0, hi = 0;
        if (row < r1) { lo = A.rowptr[row]; hi = A.rowptr[row + 1]; }
        double2 acc = make_double2(0.0, 0.0);
        for (int c0 = b0; c0 < b1; c0 += kCap) {
            const int c1 = min(b1, c0 + kCap);
            // four blocks per trip: column indices, values and gathered vector entries of all four are requested
            // before the first product is formed (the col -> gather dependency otherwise serialises the loop)
            for (int jb = c0 + threadIdx.x; jb < c1; jb += 4 * kT) {
                int colv[4]; double4 av[4]; double2 zv[4], pv_old[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = min(jb + u * kT, c1 - 1);
                    colv[u] = A.col[j];
                    av[u] = reinterpret_cast<const double4*>(A.val)[j];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    zv[u] = v[colv[u]];
                    if (MODE == 1 && iter > 0) pv_old[u] = p_old[colv[u]];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = jb + u * kT;
                    if (j < c1) {
                        double2 pv = zv[u];
                        if (MODE == 1 && iter > 0) { pv.x += beta * pv_old[u].x; pv.y += beta * pv_old[u].y; }
                        const double4 a = av[u];
                        contrib[j - c0] = make_double2(a.x * pv.x + a.y * pv.y, a.z * pv.x + a.w * pv.y);
                    }
                }
            }
            __syncthreads();
            const int jl = max(lo, c0), jh = min(hi, c1);
            for (int j = jl; j < jh; ++j) { const double2 c = contrib[j - c0]; acc.x += c.x; acc.y += c.y; }
            __syncthreads();
        }
        if (row < r1) {
            if (MODE == 0) {
                y[row] = acc;
            } else if (MODE == 1) {
                double2 pr = v[row];
                if (iter > 0) { const double2 po = p_old[row]; pr.x += beta * po.x; pr.y += beta * po.y; }
                p_new[row] = pr;
                y[row] = acc;
                acc_dot += pr.x * acc.x + pr.y * acc.y;
                acc_pp += pr.x * pr.x + pr.y * pr.y;
            } else {
                const double2 bb = bvec[row];
                const double2 r = make_double2(bb.x - acc.x, bb.y - acc.y);
                y[row] = r;
                acc_dot += r.x * r.x + r.y * r.y;
            }
        }
    }
    if (MODE != 0) {
        const double t = block_sum(acc_dot, sh);
        if (MODE == 1) {
            // (p.Ap, p.p) of the workgroup side by side: the update kernel sums both with one 16-byte load per entry
            const double t2 = block_sum(acc_pp, sh);
            if (threadIdx.x == 0) reinterpret_cast<double2*>(part_pp)[blockIdx.x] = make_double2(t, t2);
        } else if (threadIdx.x == 0) part_out[blockIdx.x] = t;
    }
}

// PCG step B: alpha = rz/pAp; x += alpha p; r -= alpha Ap; z = minv r; partial r.z, r.r
__global__ __launch_bounds__(kT) void pcg_update_kernel(
    int nb, double2* __restrict__ x, double2* __restrict__ r, double2* __restrict__ z, const double2* __restrict__ p,
    const double2* __restrict__ Ap, const double2* __restrict__ minv, const double* __restrict__ part_pAp,
    const double* __restrict__ part_pp, int np_pAp,
    const double* __restrict__ part_rz_cur, int np_rz, double* __restrict__ part_rz_out, double* __restrict__ part_rr_out,
    fb_pcg_state* st, int iter, int xcd_rows) {
    __shared__ double sh[kT / 64];
    __shared__ double sh2[3][kT / 64];
    __shared__ int sflag;
    if (threadIdx.x == 0) sflag = st->flag;
    double pAp, rz, pp;
    {
        // p.Ap and p.p: [np_pAp] double2 at part_pp (written by the SpMV kernel), r.z: [np_rz] doubles; every workgroup sums them
        // in the same order (bitwise the same scalars everywhere).  (part_pAp is not read any more.)
        const double2* pq = reinterpret_cast<const double2*>(part_pp);
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
        for (int i = threadIdx.x; i < np_pAp; i += blockDim.x) { const double2 a = pq[i]; v0 += a.x; v1 += a.y; }
        for (int i = threadIdx.x; i < np_rz; i += blockDim.x) v2 += part_rz_cur[i];
        for (int off = 32; off > 0; off >>= 1) { v0 += __shfl_down(v0, off); v1 += __shfl_down(v1, off); v2 += __shfl_down(v2, off); }
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();                                     // (also publishes sflag)
        if (lane == 0) { sh2[0][wave] = v0; sh2[1][wave] = v1; sh2[2][wave] = v2; }
        __syncthreads();
        pAp = 0.0; pp = 0.0; rz = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { pAp += sh2[0][w]; pp += sh2[1][w]; rz += sh2[2][w]; }
        (void)part_pAp;
    }
    if (sflag) return;
    if (!(pAp > 1e-5 * st->curv_eps * pp)) {
        // p^T A p <= 0, or positive below what doubles resolve (a curvature under 1e-14 of the largest diagonal entry):
        // rounding noise of a direction inside the null space of a consistent semi-definite system -- a step rz / pAp along
        // it would be as long as the noise is small (flag 3: end the leg, the host re-evaluates the true residual) -- or
        // genuine negative curvature (flag 2: the matrix is not positive semi-definite).
        if (threadIdx.x == 0 && blockIdx.x == 0) { st->flag = (pAp < -st->curv_eps * pp) ? 2 : 3; st->iter = iter; }
        return;
    }
    const double alpha = rz / pAp;
    double s_rz = 0.0, s_rr = 0.0;
    // the rows of one XCD's workgroups are the contiguous eighth that XCD's SpMV workgroups gather from (bsr_spmv_kernel):
    // what this kernel writes (z, r, x) and the SpMV reads next sits in the same L2
    int stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x, base = 0, limit = nb;
    const int nchunks = (nb + kT - 1) / kT;
    if (xcd_rows && (gridDim.x & 7) == 0 && nchunks >= 64) {
        const int per = ((nchunks + 7) >> 3) * kT;
        base = (blockIdx.x & 7) * per; limit = min(per, nb - base);
        first = (blockIdx.x >> 3) * blockDim.x + threadIdx.x; stride = (gridDim.x >> 3) * blockDim.x;
        if (xcd_rows == 2) {
            // every workgroup owns ONE contiguous run of its XCD's eighth (FEABAS_HIP_PCG_CHUNKED=1): the store form that reaches
            // the best write rate in isolation (tools/hbm_store_probe.hip)
            const int nwg = gridDim.x >> 3, wg = blockIdx.x >> 3;
            const int run = ((limit + nwg - 1) / nwg + 2 * (int)blockDim.x - 1) / (2 * (int)blockDim.x) * (2 * (int)blockDim.x);
            base += wg * run; limit = max(0, min(run, limit - wg * run));
            first = threadIdx.x; stride = blockDim.x;
        }
    }
    for (int j0 = first; j0 < limit; j0 += 2 * stride) {
        const int i0 = base + j0, i1 = i0 + stride;
        const bool two = j0 + stride < limit;
        const int ib = two ? i1 : i0;
        const double2 pa = p[i0], apa = Ap[i0], ma = minv[i0], pb = p[ib], apb = Ap[ib], mb = minv[ib];
        double2 xa = x[i0], ra = r[i0], xb = x[ib], rb = r[ib];
        xa.x += alpha * pa.x; xa.y += alpha * pa.y;
        ra.x -= alpha * apa.x; ra.y -= alpha * apa.y;
        const double2 za = make_double2(ma.x * ra.x, ma.y * ra.y);
        x[i0] = xa; r[i0] = ra; z[i0] = za;
        s_rz += ra.x * za.x + ra.y * za.y;
        s_rr += ra.x * ra.x + ra.y * ra.y;
        if (two) {
            xb.x += alpha * pb.x; xb.y += alpha * pb.y;
            rb.x -= alpha * apb.x; rb.y -= alpha * apb.y;
            const double2 zb = make_double2(mb.x * rb.x, mb.y * rb.y);
            x[i1] = xb; r[i1] = rb; z[i1] = zb;
            s_rz += rb.x * zb.x + rb.y * zb.y;
            s_rr += rb.x * rb.x + rb.y * rb.y;
        }
    }
    const double t1 = block_sum(s_rz, sh);
    const double t2 = block_sum(s_rr, sh);
    if (threadIdx.x == 0) { part_rz_out[blockIdx.x] = t1; part_rr_out[blockIdx.x] = t2; }
}

// z = minv r, partial r.z and r.r (start of a PCG leg)
__global__ __launch_bounds__(kT) void pcg_init_kernel(int nb, const double2* __restrict__ r, double2* __restrict__ z,
                                                       const double2* __restrict__ minv, double* __restrict__ part_rz,
                                                       double* __restrict__ part_rr) {
    __shared__ double sh[kT / 64];
    double s_rz = 0.0, s_rr = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const double2 ri = r[i], mi = minv[i];
        const double2 zi = make_double2(mi.x * ri.x, mi.y * ri.y);
        z[i] = zi;
        s_rz += ri.x * zi.x + ri.y * zi.y;
        s_rr += ri.x * ri.x + ri.y * ri.y;
    }
    const double t1 = block_sum(s_rz, sh);
    const double t2 = block_sum(s_rr, sh);
    if (threadIdx.x == 0) { part_rz[blockIdx.x] = t1; part_rr[blockIdx.x] = t2; }
}


// ---- deflation of floating translations (the rescue path of fb_bsr_pcg_dev) ------------------------------------------------------
// comp[i]: the deflated component vertex i belongs to, or -1.  S[c] = {sum v.x, sum v.y, sum (w v).x, sum (w v).y} over the
// vertices of component c (WEIGHTS false: w = minv; true: v -> v / minv, w -> 1 / minv: the sums of the final M-orthogonalisation).
// A thread keeps a running sum while the component does not change and the waves of one component combine before they touch
// memory: a mesh-ordered matrix costs a few thousand atomic adds per pass.
template <bool WEIGHTS>
__global__ __launch_bounds__(kT) void defl_accum_kernel(int nb, const double2* __restrict__ v, const double2* __restrict__ minv,
                                                         const int* __restrict__ comp, double* __restrict__ S) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int cur = -1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const int c = comp[i];
        if (c < 0) continue;
        if (c != cur) {
            if (cur >= 0) { atomicAdd(S + 4 * cur, a0); atomicAdd(S + 4 * cur + 1, a1); atomicAdd(S + 4 * cur + 2, a2); atomicAdd(S + 4 * cur + 3, a3); }
            cur = c; a0 = a1 = a2 = a3 = 0.0;
        }
        const double2 vi = v[i], mi = minv[i];
        if (WEIGHTS) { a0 += vi.x / mi.x; a1 += vi.y / mi.y; a2 += 1.0 / mi.x; a3 += 1.0 / mi.y; }
        else { a0 += vi.x; a1 += vi.y; a2 += mi.x * vi.x; a3 += mi.y * vi.y; }
    }
    const int first = __shfl(cur, 0);
    if (__all(cur == first)) {
        for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_down(a0, off); a1 += __shfl_down(a1, off); a2 += __shfl_down(a2, off); a3 += __shfl_down(a3, off); }
        if ((threadIdx.x & 63) != 0) cur = -1;
    }
    if (cur >= 0) { atomicAdd(S + 4 * cur, a0); atomicAdd(S + 4 * cur + 1, a1); atomicAdd(S + 4 * cur + 2, a2); atomicAdd(S + 4 * cur + 3, a3); }
}

// r <- P r, z <- P (minv r) with P = 'subtract the mean over every deflated (component, axis)', partial r.z and r.r in the slots
// the next SpMV kernel reads; G[c] = {1 / n or 0 for x, for y, sum minv.x, sum minv.y}.  Zeroes the sums of the other parity.
__global__ __launch_bounds__(kT) void defl_apply_kernel(int nb, double2* __restrict__ r, double2* __restrict__ z, const double2* __restrict__ minv,
                                                        const int* __restrict__ comp, const double* __restrict__ S, const double* __restrict__ G,
                                                        double* __restrict__ part_rz, double* __restrict__ part_rr, double* __restrict__ S_next, int ncomp) {
    __shared__ double sh[kT / 64];
    if (blockIdx.x == 0) for (int k = threadIdx.x; k < 4 * ncomp; k += blockDim.x) S_next[k] = 0.0;
    double s_rz = 0.0, s_rr = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const int c = comp[i];
        double2 ri = r[i];
        const double2 mi = minv[i];
        double2 zi;
        if (c >= 0) {
            const double* s = S + 4 * c;
            const double* g = G + 4 * c;
            const double mrx = s[0] * g[0], mry = s[1] * g[1];
            ri.x -= mrx; ri.y -= mry;
            zi.x = mi.x * ri.x - (s[2] - mrx * g[2]) * g[0];
            zi.y = mi.y * ri.y - (s[3] - mry * g[3]) * g[1];
            r[i] = ri;
        } else {
            zi = make_double2(mi.x * ri.x, mi.y * ri.y);
        }
        z[i] = zi;
        s_rz += ri.x * zi.x + ri.y * zi.y;
        s_rr += ri.x * ri.x + ri.y * ri.y;
    }
    const double t1 = block_sum(s_rz, sh);
    const double t2 = block_sum(s_rr, sh);
    if (threadIdx.x == 0) { part_rz[blockIdx.x] = t1; part_rr[blockIdx.x] = t2; }
}

// x <- x - sum_g t_g (t_g^T M x) / (t_g^T M t_g), M = 1 / minv: the solution a Jacobi-preconditioned Krylov method started from
// zero converges to (its iterates never leave minv range(A)); S from defl_accum_kernel<true>
__global__ void defl_morth_kernel(int nb, double2* __restrict__ x, const int* __restrict__ comp, const double* __restrict__ S, const double* __restrict__ G) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const int c = comp[i];
        if (c < 0) continue;
        const double* s = S + 4 * c;
        const double* g = G + 4 * c;
        double2 xi = x[i];
        if (g[0] > 0.0) xi.x -= s[0] / s[2];
        if (g[1] > 0.0) xi.y -= s[1] / s[3];
        x[i] = xi;
    }
}

// x <- P x (the mean over every deflated (component, axis) taken out); S from defl_accum_kernel<false>
__global__ void defl_center_kernel(int nb, double2* __restrict__ x, const int* __restrict__ comp, const double* __restrict__ S, const double* __restrict__ G) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const int c = comp[i];
        if (c < 0) continue;
        double2 xi = x[i];
        xi.x -= S[4 * c] * G[4 * c];
        xi.y -= S[4 * c + 1] * G[4 * c + 1];
        x[i] = xi;
    }
}

__global__ void fill2_kernel(int nb, double2* __restrict__ v, double2 c) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) v[i] = c;
}

// diagonal of the block matrix (for Jacobi) + partial max
__global__ void bsr_diag_kernel(fb_bsr_dev A, double2* __restrict__ diag, double* __restrict__ part_max) {
    __shared__ double sh[kT / 64];
    double mx = -INFINITY;
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < A.nb; row += gridDim.x * blockDim.x) {
        double2 d = make_double2(0.0, 0.0);
        for (int j = A.rowptr[row]; j < A.rowptr[row + 1]; ++j)
            if (A.col[j] == row) { const double4 a = reinterpret_cast<const double4*>(A.val)[j]; d.x += a.x; d.y += a.w; }
        diag[row] = d;
        mx = fmax(mx, fmax(d.x, d.y));
    }
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_down(mx, off));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sh[wave] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mx = fmax(mx, sh[w]);
        part_max[blockIdx.x] = mx;
    }
}

__global__ void jacobi_kernel(int nb, const double2* __restrict__ diag, double2* __restrict__ minv, double floor_, int identity) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        if (identity) { minv[i] = make_double2(1.0, 1.0); continue; }
        const double2 d = diag[i];
        minv[i] = make_double2(1.0 / fmax(d.x, floor_), 1.0 / fmax(d.y, floor_));
    }
}

// ---------------------------------------------------------------------------------- batched small solves
// Block-diagonal systems (one independent spring-linked mesh pair per equal vertex range, e.g. the in-matcher
// relaxations of matcher.py:717-742): ONE workgroup runs the whole Jacobi-PCG of one range with its vectors in
// LDS -- no launch per iteration, no host polling, and every range stops on its own residual like the
// reference's per-pair SLM does.  Reductions follow a fixed tree, so results are reproducible.
constexpr int kGT = 512;

__device__ __forceinline__ double group_sum(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                       // sh may still be read from the previous reduction
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = sh[0];
    for (int w = 1; w < kGT / 64; ++w) s += sh[w];
    return s;
}
// two sums behind ONE barrier (the iteration of pcg_groups_kernel is bound by its barriers: 9 -> 3 per iteration); per value
// the summation tree is that of group_sum, so the results are bitwise the same
// (no barrier in front: the two reductions of an iteration use buffers of their own, and between two uses of one buffer lie the
// barriers of the other reduction and of the direction update -- 3 barriers per iteration)
__device__ __forceinline__ void group_sum2(double& a, double& b, double (*sh2)[kGT / 64]) {
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh2[0][wave] = a; sh2[1][wave] = b; }
    __syncthreads();
    double s = sh2[0][0], t = sh2[1][0];
    for (int w = 1; w < kGT / 64; ++w) { s += sh2[0][w]; t += sh2[1][w]; }
    a = s; b = t;
}
__device__ __forceinline__ double group_max(double v, double* sh) {
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = sh[0];
    for (int w = 1; w < kGT / 64; ++w) s = fmax(s, sh[w]);
    return s;
}

// the same row product with the matrix of the group in LDS, slot-major (ELL): slot k of local row i at [k * gs + i] -- consecutive
// threads read consecutive addresses; rows shorter than K are padded with zero blocks on the diagonal (adding +0.0 at the end of
// the row's sum: the result is bitwise that of the CSR loop)
__device__ __forceinline__ double2 group_row_spmv_lds(const double4* vals, const unsigned short* cols, int K, int gs, int i, const double2* v) {
    double2 acc = make_double2(0.0, 0.0);
    for (int k = 0; k < K; ++k) {
        const double4 a = vals[k * gs + i];
        const double2 u = v[cols[k * gs + i]];
        acc.x += a.x * u.x + a.y * u.y;
        acc.y += a.z * u.x + a.w * u.y;
    }
    return acc;
}

// y_i = sum_j A_ij v_j for the rows of this group; columns are group-local (block-diagonal system)
__device__ __forceinline__ double2 group_row_spmv(const fb_bsr_dev& A, int row, int base, const double2* v) {
    double2 acc = make_double2(0.0, 0.0);
    for (int j = A.rowptr[row]; j < A.rowptr[row + 1]; ++j) {
        const double4 a = reinterpret_cast<const double4*>(A.val)[j];
        const double2 u = v[A.col[j] - base];
        acc.x += a.x * u.x + a.y * u.y;
        acc.y += a.z * u.x + a.w * u.y;
    }
    return acc;
}

// LDSMAT: the K block slots of every row of the group are copied into LDS once (K = longest row of the pattern): the hundreds of
// iterations of a deformed-mesh relaxation then gather nothing from L2 (7 slots x 448 rows x 34 B = 107 KB beside the 36 KB of
// vectors at the strip meshes of the 4k configuration)
template <bool LDSMAT>
__global__ __launch_bounds__(kGT) void pcg_groups_kernel(fb_bsr_dev A, const double2* __restrict__ b, double2* __restrict__ x, int gs, double rtol,
                                                         double atol, int maxiter, int precond, int* __restrict__ iters,
                                                         double* __restrict__ relres, int* __restrict__ flags, int K) {
    extern __shared__ __attribute__((aligned(16))) double2 gl[];
    double2* xs = gl; double2* r = gl + gs; double2* z = gl + 2 * gs; double2* pv = gl + 3 * gs; double2* mi = gl + 4 * gs;
    double4* mvals = reinterpret_cast<double4*>(gl + 5 * gs);                       // [K][gs]
    unsigned short* mcols = reinterpret_cast<unsigned short*>(mvals + (size_t)K * gs);   // [K][gs]
    if (LDSMAT) {
        const int gbase = blockIdx.x * gs;
        for (int i = threadIdx.x; i < gs; i += kGT) {
            const int row = gbase + i, j0 = A.rowptr[row], n = A.rowptr[row + 1] - j0;
            for (int k = 0; k < K; ++k) {
                const bool in = k < n;
                mvals[k * gs + i] = in ? reinterpret_cast<const double4*>(A.val)[j0 + k] : make_double4(0.0, 0.0, 0.0, 0.0);
                mcols[k * gs + i] = (unsigned short)(in ? A.col[j0 + k] - gbase : i);
            }
        }
        __syncthreads();
    }
    __shared__ double sh[kGT / 64];
    __shared__ double sh2[2][kGT / 64];
    __shared__ double sh3[2][kGT / 64];
    const int g = blockIdx.x, base = g * gs, tid = threadIdx.x;
    // Jacobi preconditioner of this range (optimizer.py:1958-1966): 1 / clip(diag, min(1, max / 1000))
    double dmax = -INFINITY;
    for (int i = tid; i < gs; i += kGT) {
        const int row = base + i;
        double2 d = make_double2(0.0, 0.0);
        for (int j = A.rowptr[row]; j < A.rowptr[row + 1]; ++j)
            if (A.col[j] == row) { const double4 a = reinterpret_cast<const double4*>(A.val)[j]; d.x += a.x; d.y += a.w; }
        mi[i] = d;
        dmax = fmax(dmax, fmax(d.x, d.y));
    }
    dmax = group_max(dmax, sh);
    const bool identity = precond == 0 || !(dmax > 0.0);
    const double floor_ = fmin(1.0, dmax / 1000.0);
    const double curv_eps = 1e-9 * fmax(dmax, 0.0);
    double bb = 0.0;
    for (int i = tid; i < gs; i += kGT) {
        const double2 d = mi[i];
        mi[i] = identity ? make_double2(1.0, 1.0) : make_double2(1.0 / fmax(d.x, floor_), 1.0 / fmax(d.y, floor_));
        const double2 bi = b[base + i];
        xs[i] = make_double2(0.0, 0.0);
        r[i] = bi;
        bb += bi.x * bi.x + bi.y * bi.y;
    }
    bb = group_sum(bb, sh);
    int it = 0, flag = 0;
    double rr = bb;
    if (bb > 0.0 && maxiter != 0) {
        const double bnorm = sqrt(bb);
        double tol = rtol;
        if (atol > 0.0) tol = fmax(tol, atol / bnorm);
        const double tol2 = tol * tol * bb;
        const int limit = maxiter > 0 ? maxiter : 100000;
        double rz = 0.0;
        for (int i = tid; i < gs; i += kGT) {
            const double2 ri = r[i], m = mi[i];
            const double2 zi = make_double2(m.x * ri.x, m.y * ri.y);
            z[i] = zi; pv[i] = zi;
            rz += ri.x * zi.x + ri.y * zi.y;
        }
        rz = group_sum(rz, sh);                        // (barriers inside: pv is complete)
        while (it < limit) {
            double pAp = 0.0, pp = 0.0;
            // rows of a thread stay with it through the iteration: Ap is kept in z (z is dead until it is recomputed)
            for (int i = tid; i < gs; i += kGT) {
                const double2 ap = LDSMAT ? group_row_spmv_lds(mvals, mcols, K, gs, i, pv) : group_row_spmv(A, base + i, base, pv);
                const double2 pi = pv[i];
                z[i] = ap;
                pAp += pi.x * ap.x + pi.y * ap.y;
                pp += pi.x * pi.x + pi.y * pi.y;
            }
            group_sum2(pAp, pp, sh2);
            if (!(pAp > curv_eps * pp)) { flag = (pAp < -curv_eps * pp) ? 2 : 3; break; }     // breakdown / semi-definite stagnation
            const double alpha = rz / pAp;
            double rr_new = 0.0, rz_new = 0.0;
            for (int i = tid; i < gs; i += kGT) {
                const double2 pi = pv[i], ap = z[i], m = mi[i];
                double2 xi = xs[i], ri = r[i];
                xi.x += alpha * pi.x; xi.y += alpha * pi.y;
                ri.x -= alpha * ap.x; ri.y -= alpha * ap.y;
                xs[i] = xi; r[i] = ri;
                const double2 zi = make_double2(m.x * ri.x, m.y * ri.y);
                z[i] = zi;
                rr_new += ri.x * ri.x + ri.y * ri.y;
                rz_new += ri.x * zi.x + ri.y * zi.y;
            }
            group_sum2(rr_new, rz_new, sh3);
            rr = rr_new;
            ++it;
            if (rr <= tol2) { flag = 1; break; }
            const double beta = rz_new / rz;
            rz = rz_new;
            for (int i = tid; i < gs; i += kGT) {
                const double2 zi = z[i], pi = pv[i];
                pv[i] = make_double2(zi.x + beta * pi.x, zi.y + beta * pi.y);
            }
            __syncthreads();
        }
        // true residual of the returned iterate
        __syncthreads();
        double rt = 0.0;
        for (int i = tid; i < gs; i += kGT) {
            const double2 ax = LDSMAT ? group_row_spmv_lds(mvals, mcols, K, gs, i, xs) : group_row_spmv(A, base + i, base, xs);
            const double2 bi = b[base + i];
            const double dx = bi.x - ax.x, dy = bi.y - ax.y;
            rt += dx * dx + dy * dy;
        }
        rr = group_sum(rt, sh);
    }
    for (int i = tid; i < gs; i += kGT) x[base + i] = xs[i];
    if (tid == 0) {
        iters[g] = it;
        relres[g] = bb > 0.0 ? sqrt(rr / bb) : 0.0;
        flags[g] = flag;
    }
}

}  // namespace

int fb_bsr_pcg_groups(fb_ctx* ctx, fb_bsr* M, int ngroups, double rtol, double atol, int maxiter, int precond, int* iters_dev, double* relres_dev,
                      int* flags_dev) {
    const int gs = M->d.nb / ngroups;
    const size_t lds = sizeof(double2) * 5 * (size_t)gs;
    if (M->d.nb % ngroups != 0 || lds > 150 * 1024) return fb_fail(ctx, FB_ERR_ARG, "fb_bsr_pcg_groups: %d vertices per group do not fit the LDS", gs);
    // the matrix of a group beside its vectors when both fit (160 KB per CU; the kernel has ~0.3 KB of static LDS); FEABAS_HIP_GROUPS_LDSMAT=0: never
    static const int ldsmat = [] { const char* e = getenv("FEABAS_HIP_GROUPS_LDSMAT"); return e ? atoi(e) : 1; }();
    const int K = M->max_row_blocks;
    const size_t lds_mat = lds + (size_t)K * gs * (sizeof(double4) + sizeof(unsigned short)) + 16;
    FB_PROF(ctx, "pcg_groups");
    if (ldsmat && K > 0 && gs <= 65535 && lds_mat <= 158 * 1024) {
        FB_HIP(ctx, hipFuncSetAttribute((const void*)pcg_groups_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mat));
        hipLaunchKernelGGL(pcg_groups_kernel<true>, dim3(ngroups), dim3(kGT), lds_mat, ctx->stream, M->d, M->b, M->x, gs, rtol, atol, maxiter, precond, iters_dev,
                           relres_dev, flags_dev, K);
    } else {
        FB_HIP(ctx, hipFuncSetAttribute((const void*)pcg_groups_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(pcg_groups_kernel<false>, dim3(ngroups), dim3(kGT), lds, ctx->stream, M->d, M->b, M->x, gs, rtol, atol, maxiter, precond, iters_dev,
                           relres_dev, flags_dev, 0);
    }
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

namespace {

int grid1(const fb_bsr_dev& A) { return std::max(1, std::min(kMaxWG1, (A.nb + kT - 1) / kT)); }

}  // namespace

// ---------------------------------------------------------------------------------- fb_bsr
int fb_bsr_free(fb_ctx* ctx, fb_bsr* M) {
    if (!M) return FB_OK;
    hipStreamSynchronize(ctx->stream);
    hipFree(M->d.rowptr); hipFree(M->d.col); hipFree(M->d.val);
    for (double2* v : {M->x, M->r, M->z, M->p0, M->p1, M->Ap, M->minv, M->b, M->diag, M->xbest}) hipFree(v);
    hipFree(M->parts); hipFree(M->state);
    if (M->pcg_graph) hipGraphExecDestroy(M->pcg_graph);
    delete M;
    return FB_OK;
}

int fb_bsr_alloc(fb_ctx* ctx, int nb, int64_t nnzb, fb_bsr** out) {
    if (nnzb >= (1LL << 31)) return fb_fail(ctx, FB_ERR_ARG, "block count %lld exceeds int32 indexing", (long long)nnzb);
    fb_bsr* M = new fb_bsr();
    M->d.nb = nb;
    { static const int xr = [] { const char* e = getenv("FEABAS_HIP_SPMV_XCD"); return e ? atoi(e) : 1; }(); M->d.xcd_rows = xr; }
    M->nnzb = nnzb;
    hipError_t e = hipSuccess;
    auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 16); };
    A((void**)&M->d.rowptr, sizeof(int) * ((size_t)nb + 1));
    A((void**)&M->d.col, sizeof(int) * (size_t)nnzb);
    A((void**)&M->d.val, sizeof(double) * 4 * (size_t)nnzb);
    for (double2** v : {&M->x, &M->r, &M->z, &M->p0, &M->p1, &M->Ap, &M->minv, &M->b, &M->diag}) A((void**)v, sizeof(double2) * (size_t)nb);
    A((void**)&M->parts, sizeof(double) * kNP * 8);
    A((void**)&M->state, sizeof(fb_pcg_state));
    if (e != hipSuccess) {
        fb_bsr_free(ctx, M);
        return fb_fail(ctx, FB_ERR_NOMEM, "fb_bsr_alloc: %s", hipGetErrorString(e));
    }
    *out = M;
    return FB_OK;
}

int fb_bsr_upload(fb_ctx* ctx, int nb, const std::vector<int>& rowptr, const std::vector<int>& col, const std::vector<double>& val,
                  fb_bsr** out) {
    fb_bsr* M = nullptr;
    int rc = fb_bsr_alloc(ctx, nb, (int64_t)col.size(), &M);
    if (rc) return rc;
    { int mr = 0; for (int v = 0; v < nb; ++v) mr = std::max(mr, rowptr[v + 1] - rowptr[v]); M->max_row_blocks = mr; }
    rc = fb_copy_h2d(ctx, M->d.rowptr, rowptr.data(), sizeof(int) * rowptr.size());
    if (!rc) rc = fb_copy_h2d(ctx, M->d.col, col.data(), sizeof(int) * col.size());
    if (!rc) rc = fb_copy_h2d(ctx, M->d.val, val.data(), sizeof(double) * val.size());
    if (rc) return rc;
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = M;
    return FB_OK;
}

// y = A v on device vectors
int fb_bsr_spmv_dev(fb_ctx* ctx, fb_bsr* M, const double2* v, double2* y) {
    FB_PROF(ctx, "bsr_spmv");
    hipLaunchKernelGGL(bsr_spmv_kernel<0>, dim3(grid1(M->d)), dim3(kT), 0, ctx->stream, M->d, v, nullptr, nullptr, y, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

int fb_bsr_setup_jacobi(fb_ctx* ctx, fb_bsr* M, int precond) {
    const int g = std::min(kNP, std::max(1, (M->d.nb + kT - 1) / kT));
    hipLaunchKernelGGL(bsr_diag_kernel, dim3(g), dim3(kT), 0, ctx->stream, M->d, M->diag, M->parts);
    std::vector<double> pm(g);
    FB_HIP(ctx, hipMemcpyAsync(pm.data(), M->parts, sizeof(double) * g, hipMemcpyDeviceToHost, ctx->stream));
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double mx = -INFINITY;
    for (double v : pm) mx = std::max(mx, v);
    M->diag_max = mx;
    const int identity = (precond == 0) || !(mx > 0.0);       // optimizer.py:1963-1966: M0 = None when max <= 0
    const double floor_ = std::min(1.0, mx / 1000.0);
    hipLaunchKernelGGL(jacobi_kernel, dim3(g), dim3(kT), 0, ctx->stream, M->d.nb, M->diag, M->minv, floor_, identity);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}


// ---- floating sub-systems: which translations lie in the null space ---------------------------------------------------------------
// A system without a locked mesh is singular: every link-connected set of free meshes (and every island of a mesh) can translate
// as a whole, A t = 0.  The reference's matrices are float32 in places (cross-link matrix, stress: SURVEY app. B) and so are ours
// to match them: t^T A t is +-1e-10 of the largest eigenvalue instead of zero, b has a component of ~1e-8 ||b|| along t, and the
// soft rotation of a floating pair sits only a factor ~50 above that noise.  Plain CG then meets p^T A p <= 0 long before it has
// converged (measured: true residual 1e-3 at the end of the first leg, profiles/r06b_pcg_floating_probe_*).  The rescue: find the
// connected components of the matrix graph (host, union-find over the block pattern), test the two translations of each
// (max |A t| <= 1e-5 of the component's largest diagonal entry), and run the PCG on the system deflated by them: r and
// z = minv r are kept orthogonal to every such t (defl_accum_kernel / defl_apply_kernel behind every update), so p has no
// component along t and the noise curvature cannot enter p^T A p.  At the end x is made (1 / minv)-orthogonal to the t's: the
// limit of a Jacobi-preconditioned Krylov method started from zero, which is what the oracle's exact solve defines.
struct fb_deflation {
    int ncomp = 0;
    int* comp = nullptr;       // [nb] deflated component of a vertex or -1
    double* G = nullptr;       // [ncomp][4]: 1/n (or 0) for x, for y, sum minv.x, sum minv.y
    double* S = nullptr;       // [2][ncomp][4] running sums, two parities
    int parity = 0;
    ~fb_deflation() { hipFree(comp); hipFree(G); hipFree(S); }
};

static int defl_setup(fb_ctx* ctx, fb_bsr* M, fb_deflation* D, int g2) {
    const int nb = M->d.nb;
    std::vector<int> rowptr(nb + 1), col((size_t)M->nnzb);
    FB_HIP(ctx, hipMemcpyAsync(rowptr.data(), M->d.rowptr, sizeof(int) * (nb + 1), hipMemcpyDeviceToHost, ctx->stream));
    FB_HIP(ctx, hipMemcpyAsync(col.data(), M->d.col, sizeof(int) * col.size(), hipMemcpyDeviceToHost, ctx->stream));
    // A t_x and A t_y for t = 'every vertex' (components do not couple: the rows of a component see its own t only)
    std::vector<double2> y[2], diag(nb), minv(nb);
    for (int a = 0; a < 2; ++a) {
        y[a].resize(nb);
        hipLaunchKernelGGL(fill2_kernel, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->p0, a == 0 ? make_double2(1.0, 0.0) : make_double2(0.0, 1.0));
        int rc = fb_bsr_spmv_dev(ctx, M, M->p0, M->Ap);
        if (rc) return rc;
        FB_HIP(ctx, hipMemcpyAsync(y[a].data(), M->Ap, sizeof(double2) * nb, hipMemcpyDeviceToHost, ctx->stream));
    }
    FB_HIP(ctx, hipMemcpyAsync(diag.data(), M->diag, sizeof(double2) * nb, hipMemcpyDeviceToHost, ctx->stream));
    FB_HIP(ctx, hipMemcpyAsync(minv.data(), M->minv, sizeof(double2) * nb, hipMemcpyDeviceToHost, ctx->stream));
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> parent(nb);
    for (int i = 0; i < nb; ++i) parent[i] = i;
    auto find = [&](int i) { while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; } return i; };
    for (int i = 0; i < nb; ++i)
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) {
            const int a = find(i), b = find(col[j]);
            if (a != b) parent[std::max(a, b)] = std::min(a, b);
        }
    std::vector<int> label(nb, -1), root_of;
    std::vector<double> dmax, ymax[2];
    for (int i = 0; i < nb; ++i) {
        const int r = find(i);
        if (label[r] < 0) { label[r] = (int)root_of.size(); root_of.push_back(r); dmax.push_back(0.0); ymax[0].push_back(0.0); ymax[1].push_back(0.0); }
        const int c = label[r];
        label[i] = c;
        dmax[c] = std::max(dmax[c], std::max(std::fabs(diag[i].x), std::fabs(diag[i].y)));
        for (int a = 0; a < 2; ++a) ymax[a][c] = std::max(ymax[a][c], std::max(std::fabs(y[a][i].x), std::fabs(y[a][i].y)));
    }
    const int nc_all = (int)root_of.size();
    std::vector<int> newid(nc_all, -1);
    std::vector<double> G;
    int nc = 0;
    for (int c = 0; c < nc_all; ++c) {
        const bool nx = dmax[c] > 0.0 && ymax[0][c] <= 1e-5 * dmax[c], ny = dmax[c] > 0.0 && ymax[1][c] <= 1e-5 * dmax[c];
        if (!nx && !ny) continue;
        newid[c] = nc++;
        G.insert(G.end(), {nx ? 1.0 : 0.0, ny ? 1.0 : 0.0, 0.0, 0.0});          // (counts first, inverted below)
    }
    D->ncomp = nc;
    if (nc == 0) return FB_OK;
    std::vector<double> cnt(nc, 0.0);
    for (int i = 0; i < nb; ++i) {
        const int c = newid[label[i]];
        label[i] = c;
        if (c < 0) continue;
        cnt[c] += 1.0;
        G[4 * c + 2] += minv[i].x; G[4 * c + 3] += minv[i].y;
    }
    for (int c = 0; c < nc; ++c) { G[4 * c] = G[4 * c] > 0.0 ? 1.0 / cnt[c] : 0.0; G[4 * c + 1] = G[4 * c + 1] > 0.0 ? 1.0 / cnt[c] : 0.0; }
    FB_HIP(ctx, hipMalloc((void**)&D->comp, sizeof(int) * nb));
    FB_HIP(ctx, hipMalloc((void**)&D->G, sizeof(double) * 4 * nc));
    FB_HIP(ctx, hipMalloc((void**)&D->S, sizeof(double) * 8 * nc));
    FB_HIP(ctx, hipMemcpyAsync(D->comp, label.data(), sizeof(int) * nb, hipMemcpyHostToDevice, ctx->stream));
    FB_HIP(ctx, hipMemcpyAsync(D->G, G.data(), sizeof(double) * 4 * nc, hipMemcpyHostToDevice, ctx->stream));
    FB_HIP(ctx, hipMemsetAsync(D->S, 0, sizeof(double) * 8 * nc, ctx->stream));
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));          // (the host vectors go out of scope)
    return FB_OK;
}

// Solve A x = b for the vectors resident in M (M->b set, M->x = x0).  fixed_iters > 0: run exactly
// that many iterations with no convergence exit.
int fb_bsr_pcg_dev(fb_ctx* ctx, fb_bsr* M, double rtol, double atol, int maxiter, int fixed_iters, int* iters_out, double* relres_out) {
    const int nb = M->d.nb;
    const int g1 = grid1(M->d);
    static const int wg2_env = [] { const char* e = getenv("FEABAS_HIP_PCG_WG2"); return e ? std::max(1, std::min(kNP, atoi(e))) : 0; }();
    const int g2 = std::min(wg2_env ? wg2_env : kWG2, std::max(1, (nb + kT - 1) / kT));
    double* P = M->parts;
    double* part_pAp = P;                 // [kNP]
    double* part_rz[2] = {P + kNP, P + 2 * kNP};
    double* part_rr[2] = {P + 3 * kNP, P + 4 * kNP};
    double* part_tmp = P + 5 * kNP;
    double* part_pp = P + 6 * kNP;
    const double curv_eps = 1e-9 * std::max(M->diag_max, 0.0);
    // ||b||^2
    std::vector<double> hp(kNP);
    auto host_sum = [&](double* dev, int n, double* out) -> int {
        FB_HIP(ctx, hipMemcpyAsync(hp.data(), dev, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += hp[i];
        *out = s;
        return FB_OK;
    };
    int rc;
    // bb via MODE 2 with v = 0?  cheaper: r = b - A x0 gives rr; bb needs its own pass: use init kernel with minv as weights
    hipLaunchKernelGGL(pcg_init_kernel, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->b, M->z, M->minv, part_tmp, part_rr[0]);
    double bb = 0.0;
    if ((rc = host_sum(part_rr[0], g2, &bb))) return rc;
    M->last_bnorm = std::sqrt(bb);
    if (iters_out) *iters_out = 0;
    if (relres_out) *relres_out = 0.0;
    if (bb == 0.0 || (maxiter == 0 && fixed_iters <= 0)) {          // maxiter: <0 unlimited, 0 -> zeros, >0 cap          // optimizer.py:1974-1975
        FB_HIP(ctx, hipMemsetAsync(M->x, 0, sizeof(double2) * nb, ctx->stream));
        return FB_OK;
    }
    const double bnorm = std::sqrt(bb);
    M->last_bnorm = bnorm;
    M->probe_stopped = false;
    double tol = rtol;
    if (atol > 0.0) tol = std::max(tol, atol / bnorm);               // optimizer.py:1993-1996
    if (fixed_iters > 0) tol = 0.0;
    const int limit = fixed_iters > 0 ? fixed_iters : (maxiter > 0 ? maxiter : 100 * 1000);
    int total_iters = 0;
    double relres = 0.0;
    const int check_every = 32;
    // launch-bound sizes replay the batch as a graph (not under the per-kernel profiler, whose event pairs
    // sit between the launches)
    bool use_graph = !ctx->prof_on && ctx->pcg_graph_max_nb > 0 && nb <= ctx->pcg_graph_max_nb && !M->pcg_graph_off;
    // What the reference keeps across its restarts is the iterate with the best TRUE residual (SLM_Callback.solution /
    // min_cost, optimizer.py:1881-1942, 2040-2047), and it stops when a restart no longer helps (exit codes, 2063-2075).
    // Same here, leg by leg: a tolerance below what the arithmetic can reach on this matrix (a floating pair has soft modes
    // and a null space; asked for 1e-11 the legs used to go on and on -- every leg converges in its recurrence residual, the
    // true residual stops improving and then GROWS as round-off feeds the null space, 348 x ||b|| after eight legs, round 5)
    // ends at the best iterate: a leg that does not improve the true residual is undone, one that does not halve it is the
    // last, and so is one that doubles ||x|| (a correction leg refines x; one that doubles it ran away along a null vector).
    const int trace = [] { const char* e = getenv("FEABAS_HIP_PCG_TRACE"); return e ? atoi(e) : 0; }();      // (read per call: a test switches it on)
    static const int keep_best = [] { const char* e = getenv("FEABAS_HIP_PCG_BEST"); return e ? atoi(e) : 1; }();
    static const int upd_chunked = [] { const char* e = getenv("FEABAS_HIP_PCG_CHUNKED"); return e ? atoi(e) : 0; }();
    static const int deflate = [] { const char* e = getenv("FEABAS_HIP_PCG_DEFLATE"); return e ? atoi(e) : 1; }();
    fb_deflation D;                                  // (frees its device blocks when the solve returns)
    bool deflated = false, breakdown = false;
    // r <- P r, z <- P minv r, partials into the slots of parity `slot` (deflated passes only)
    auto deflate_rz = [&](int slot) {
        double* S_cur = D.S + (size_t)D.parity * 4 * D.ncomp;
        double* S_nxt = D.S + (size_t)(D.parity ^ 1) * 4 * D.ncomp;
        hipLaunchKernelGGL(defl_accum_kernel<false>, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->r, M->minv, D.comp, S_cur);
        hipLaunchKernelGGL(defl_apply_kernel, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->r, M->z, M->minv, D.comp, S_cur, D.G, part_rz[slot],
                           part_rr[slot], S_nxt, D.ncomp);
        D.parity ^= 1;
    };
    for (int pass = 0; pass < 2; ++pass) {
    double best_rel = INFINITY, xx_ref = 0.0;        // best true residual at the start of a leg >= 1 (its iterate is in M->xbest)
    bool troubled = false;                           // a leg ended on noise-level curvature, or the legs stopped helping
    for (int leg = 0; leg < 8; ++leg) {
        // r = b - A x, rr
        {
            FB_PROF(ctx, "bsr_spmv_residual");
            hipLaunchKernelGGL(bsr_spmv_kernel<2>, dim3(g1), dim3(kT), 0, ctx->stream, M->d, M->x, nullptr, nullptr, M->r, M->b,
                               part_tmp, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0);
        }
        double rr = 0.0;
        if (deflated) {
            // the residual that counts is the deflated one: what b holds along the null vectors (float32 noise of the
            // reference's arithmetic, ~1e-8 ||b||) no x can remove
            deflate_rz(1);
            if ((rc = host_sum(part_rr[1], g2, &rr))) return rc;
        } else if ((rc = host_sum(part_tmp, g1, &rr))) return rc;
        relres = std::sqrt(rr) / bnorm;
        if (trace) {
            double xx = 0.0;
            hipLaunchKernelGGL(pcg_init_kernel, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->x, M->p0, M->minv, part_tmp, part_rr[0]);
            if ((rc = host_sum(part_rr[0], g2, &xx))) return rc;
            fprintf(stderr, "[pcg] nb %d%s leg %d iters %d true relres %.3e ||x|| %.6e tol %.1e\n", nb, deflated ? " deflated" : "", leg, total_iters, relres, std::sqrt(xx), tol);
        }
        if (fixed_iters <= 0 && relres <= tol) break;
        if (fixed_iters <= 0 && keep_best && leg >= 1) {
            double xx = 0.0;                             // ||x||^2 (p0 is scratch here: iteration 0 of a leg rewrites it)
            hipLaunchKernelGGL(pcg_init_kernel, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->x, M->p0, M->minv, part_tmp, part_rr[0]);
            if ((rc = host_sum(part_rr[0], g2, &xx))) return rc;
            const bool ran_away = leg >= 2 && !(xx <= 4.0 * xx_ref);
            if (!(relres < best_rel) || ran_away) {          // (a residual that is not a number ends the solve as well)
                if (M->xbest && best_rel < INFINITY) {
                    FB_HIP(ctx, hipMemcpyAsync(M->x, M->xbest, sizeof(double2) * (size_t)nb, hipMemcpyDeviceToDevice, ctx->stream));
                    relres = best_rel;
                }
                troubled = true;
                break;
            }
            const bool stalled = relres > 0.5 * best_rel;
            best_rel = relres;
            if (leg == 1) xx_ref = xx;
            if (stalled) troubled = true;
            if (stalled || total_iters >= limit) break;
            if (!M->xbest) FB_HIP(ctx, hipMalloc((void**)&M->xbest, sizeof(double2) * (size_t)nb));
            FB_HIP(ctx, hipMemcpyAsync(M->xbest, M->x, sizeof(double2) * (size_t)nb, hipMemcpyDeviceToDevice, ctx->stream));
        }
        if (total_iters >= limit) break;
        // start a leg
        fb_pcg_state hs;
        hs.tol2bb = tol * tol * bb;
        // inside a leg the recurrence residual is compared against a slightly tighter target so that the
        // true residual re-evaluated at the end of the leg meets `tol`
        hs.tol2bb *= 0.81;
        hs.flag = 0; hs.iter = 0; hs.rr = rr; hs.curv_eps = curv_eps;
        FB_HIP(ctx, hipMemcpyAsync(M->state, &hs, sizeof(hs), hipMemcpyHostToDevice, ctx->stream));
        if (!deflated) hipLaunchKernelGGL(pcg_init_kernel, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->r, M->z, M->minv, part_rz[1], part_rr[1]);
        int it = 0;
        bool stop = false;
        double hist_rr[3] = {rr, rr, rr}; int hist_it[3] = {0, 0, 0};      // residual^2 at the last three checks of this leg
        auto iterate = [&](int i) {                           // one iteration = two launches; `i` is the index inside the leg
            const int cur = i & 1, prev = cur ^ 1;            // K2 of iteration `i` writes slot cur; slot prev holds r_i.z_i
            double2* p_new = cur ? M->p1 : M->p0;
            double2* p_old = cur ? M->p0 : M->p1;
            {
                FB_PROF_B(ctx, "pcg_spmv_fused", 36.0 * (double)M->nnzb + 4.0 * nb + 5.0 * 16.0 * nb);
                hipLaunchKernelGGL(bsr_spmv_kernel<1>, dim3(g1), dim3(kT), 0, ctx->stream, M->d, M->z, p_old, p_new, M->Ap, nullptr,
                                   part_pAp, part_pp, part_rz[prev], part_rz[cur], part_rr[prev], g2, M->state, i);
            }
            {
                FB_PROF_B(ctx, "pcg_update_fused", 7.0 * 16.0 * nb);
                hipLaunchKernelGGL(pcg_update_kernel, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->x, M->r, M->z, p_new, M->Ap, M->minv,
                                   part_pAp, part_pp, g1, part_rz[prev], g2, part_rz[cur], part_rr[cur], M->state, i, (M->d.xcd_rows && upd_chunked) ? 2 : M->d.xcd_rows);
            }
            if (deflated) deflate_rz(cur);                   // r - alpha P(A p) = P(r - alpha A p): the update kernel's r, z and dots redone on the deflated vectors
        };
        while (!stop) {
            const int batch = std::min(check_every, limit - total_iters - it);
            if (batch <= 0) break;
            int graph_base = -1;                              // >= 0: the batch ran as the graph, whose kernels count from check_every
            if (use_graph && it > 0 && batch == check_every) {
                // every batch but the first of a leg is the same 2 x check_every launches (the slot parity
                // repeats, the scalars live in M->state): replay them as one graph.  The kernels of the graph
                // carry the iteration numbers check_every .. 2 check_every - 1; a stop reports one of those.
                if (!M->pcg_graph) {
                    // A capture that fails is not a failure of the solve: the plain launches below do the same work.  Whatever
                    // state the attempt left behind is cleared (the capture is ended, the sticky error read), the graph path is
                    // switched off for this matrix and the batch runs as launches.
                    hipGraph_t g = nullptr;
                    hipError_t ge = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
                    if (ge == hipSuccess) {
                        for (int k = 0; k < check_every; ++k) iterate(check_every + k);
                        const hipError_t le = hipGetLastError();                   // (a launch refused inside the capture)
                        ge = hipStreamEndCapture(ctx->stream, &g);                // ends the capture also when it was invalidated
                        if (ge == hipSuccess && le != hipSuccess) ge = le;
                        if (ge == hipSuccess && !g) ge = hipErrorUnknown;
                    }
                    if (ge == hipSuccess) {
                        ge = hipGraphInstantiate(&M->pcg_graph, g, nullptr, nullptr, 0);
                        if (ge != hipSuccess) M->pcg_graph = nullptr;
                    }
                    if (g) hipGraphDestroy(g);
                    if (ge != hipSuccess) {
                        (void)hipGetLastError();
                        M->pcg_graph_off = true;
                        use_graph = false;
                    } else {
                        M->pcg_graph_iters = check_every;
                    }
                }
            }
            if (use_graph && it > 0 && batch == check_every && M->pcg_graph) {
                FB_HIP(ctx, hipGraphLaunch(M->pcg_graph, ctx->stream));
                graph_base = it;
                it += batch;
            } else {
                for (int k = 0; k < batch; ++k, ++it) iterate(it);
            }
            FB_HIP(ctx, hipGetLastError());
            FB_HIP(ctx, hipMemcpyAsync(&hs, M->state, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
            FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (hs.flag && graph_base >= 0) hs.iter = graph_base + (hs.iter - check_every);
            if (hs.flag == 1 || hs.flag == 3) { it = hs.iter; stop = true; if (hs.flag == 3) troubled = true; }
            else if (hs.flag == 2) {
                total_iters += hs.iter;
                if (iters_out) *iters_out = total_iters;
                if (pass == 0 && deflate && fixed_iters <= 0) { it = 0; stop = true; troubled = true; breakdown = true; continue; }   // (the float32 noise of a floating system's null vectors can read as negative curvature: the deflated pass decides)
                return fb_fail(ctx, FB_ERR_BREAKDOWN, "PCG breakdown at iteration %d: p^T A p <= 0 (matrix not positive semi-definite)", total_iters);
            } else if (M->probe_limit > 0 && fixed_iters <= 0 && hs.rr > 0.0) {
                // iterations still needed at the decay of the last two checks (64 iterations); a stalled residual projects to infinity
                hist_rr[0] = hist_rr[1]; hist_rr[1] = hist_rr[2]; hist_rr[2] = hs.rr;
                hist_it[0] = hist_it[1]; hist_it[1] = hist_it[2]; hist_it[2] = it;
                if (it >= 4 * check_every && hist_it[2] > hist_it[0]) {
                    const double decay = 0.5 * std::log(hist_rr[2] / hist_rr[0]) / (double)(hist_it[2] - hist_it[0]);     // per iteration, of the norm
                    const double togo = 0.5 * std::log(hs.tol2bb / hist_rr[2]);                                            // (< 0)
                    const double need = decay < 0.0 ? togo / decay : INFINITY;
                    if ((double)(total_iters + it) + need > (double)M->probe_limit) { M->probe_stopped = true; stop = true; }
                }
            }
        }
        total_iters += it;
        if (breakdown) break;
        if (M->probe_stopped) {
            // the true residual of the iterate reached, then out (like an iteration cap)
            hipLaunchKernelGGL(bsr_spmv_kernel<2>, dim3(g1), dim3(kT), 0, ctx->stream, M->d, M->x, nullptr, nullptr, M->r, M->b,
                               part_tmp, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0);
            if ((rc = host_sum(part_tmp, g1, &rr))) return rc;
            relres = std::sqrt(rr) / bnorm;
            break;
        }
        if (fixed_iters > 0) {
            // report the true residual after the fixed number of iterations
            hipLaunchKernelGGL(bsr_spmv_kernel<2>, dim3(g1), dim3(kT), 0, ctx->stream, M->d, M->x, nullptr, nullptr, M->r, M->b,
                               part_tmp, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0);
            if ((rc = host_sum(part_tmp, g1, &rr))) return rc;
            relres = std::sqrt(rr) / bnorm;
            break;
        }
    }
    if (pass == 0 && deflate && fixed_iters <= 0 && troubled && (breakdown || relres > tol) && total_iters < limit && !M->probe_stopped) {
        if ((rc = defl_setup(ctx, M, &D, g2))) return rc;
        if (trace) fprintf(stderr, "[pcg] nb %d: legs in trouble at true relres %.3e%s; %d floating component(s) found\n", nb, relres, breakdown ? " (negative curvature)" : "", D.ncomp);
        if (D.ncomp > 0) {
            // the deflated operator is P A P: the iterate starts in the range of P (what pass 0 left along the null vectors --
            // it can be large -- would meet the float32 noise of A t in every residual) and stays there (p = P p throughout)
            hipLaunchKernelGGL(defl_accum_kernel<false>, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->x, M->minv, D.comp, D.S);
            hipLaunchKernelGGL(defl_center_kernel, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->x, D.comp, D.S, D.G);
            FB_HIP(ctx, hipMemsetAsync(D.S, 0, sizeof(double) * 8 * D.ncomp, ctx->stream));
            D.parity = 0;
            deflated = true; use_graph = false; breakdown = false;
            continue;
        }
    }
    if (breakdown)
        return fb_fail(ctx, FB_ERR_BREAKDOWN, "PCG breakdown at iteration %d: p^T A p <= 0 (matrix not positive semi-definite)", total_iters);
    break;
    }
    if (deflated) {
        // the null-space part of x: the one a Jacobi-preconditioned Krylov method started from zero converges to
        double* S_cur = D.S + (size_t)D.parity * 4 * D.ncomp;
        hipLaunchKernelGGL(defl_accum_kernel<true>, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->x, M->minv, D.comp, S_cur);
        hipLaunchKernelGGL(defl_morth_kernel, dim3(g2), dim3(kT), 0, ctx->stream, nb, M->x, D.comp, S_cur, D.G);
        FB_HIP(ctx, hipGetLastError());
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));      // (D's blocks are freed when this function returns)
    }
    if (iters_out) *iters_out = total_iters;
    if (relres_out) *relres_out = relres;
    if (fixed_iters <= 0 && relres > tol && maxiter < 0 && !M->probe_stopped)
        return fb_fail(ctx, FB_ERR_NOCONV, "PCG stopped at relative residual %.3e > %.3e after %d iterations", relres, tol, total_iters);
    return FB_OK;
}

// ---------------------------------------------------------------------------------- scalar CSR -> BSR (host)
int fb_csr_to_bsr_host(fb_ctx* ctx, int64_t n, const int64_t* indptr, const int32_t* idx, const double* val, int symmetrize,
                       int* nb_out, std::vector<int>& browptr, std::vector<int>& bcol, std::vector<double>& bval) {
    const int64_t nb = (n + 1) / 2;
    if (nb >= (1LL << 31)) return fb_fail(ctx, FB_ERR_ARG, "matrix too large");
    // entries (block row, block col, sub index, value); with symmetrize each entry also contributes its transpose / 2
    struct Ent { int64_t key; int sub; double v; };
    const int64_t nnz = indptr[n];
    std::vector<Ent> ents;
    ents.reserve((size_t)nnz * (symmetrize ? 2 : 1));
    for (int64_t r = 0; r < n; ++r) {
        for (int64_t j = indptr[r]; j < indptr[r + 1]; ++j) {
            const int64_t c = idx[j];
            if (c < 0 || c >= n) return fb_fail(ctx, FB_ERR_ARG, "column index %lld out of range", (long long)c);
            const double v = symmetrize ? 0.5 * val[j] : val[j];
            ents.push_back({(r / 2) * nb + (c / 2), (int)((r & 1) * 2 + (c & 1)), v});
            if (symmetrize) ents.push_back({(c / 2) * nb + (r / 2), (int)((c & 1) * 2 + (r & 1)), v});
        }
    }
    std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.key < b.key; });
    browptr.assign((size_t)nb + 1, 0);
    bcol.clear(); bval.clear();
    int64_t last = -1;
    for (const Ent& e : ents) {
        if (e.key != last) {
            last = e.key;
            bcol.push_back((int)(e.key % nb));
            bval.insert(bval.end(), 4, 0.0);
            browptr[(size_t)(e.key / nb) + 1] += 1;
        }
        bval[bval.size() - 4 + e.sub] += e.v;
    }
    for (int64_t i = 0; i < nb; ++i) browptr[i + 1] += browptr[i];
    // an odd n leaves a dangling DoF: give it a unit diagonal so that the padded system stays regular
    if (n & 1) {
        const int br = (int)(nb - 1);
        bool found = false;
        for (int j = browptr[br]; j < browptr[br + 1]; ++j)
            if (bcol[j] == br) { bval[(size_t)j * 4 + 3] += 1.0; found = true; }
        if (!found) return fb_fail(ctx, FB_ERR_ARG, "odd-sized matrix without a diagonal block in its last row is not supported");
    }
    *nb_out = (int)nb;
    return FB_OK;
}

extern "C" {

int fb_csr_upload(fb_ctx* ctx, int64_t n, const int64_t* indptr, const int32_t* idx, const double* val, int symmetrize, fb_csr** out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, n > 0 && indptr && idx && val && out);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int> browptr, bcol;
    std::vector<double> bval;
    int nb = 0;
    int rc = fb_csr_to_bsr_host(ctx, n, indptr, idx, val, symmetrize, &nb, browptr, bcol, bval);
    if (rc) return rc;
    fb_bsr* M = nullptr;
    rc = fb_bsr_upload(ctx, nb, browptr, bcol, bval, &M);
    if (rc) return rc;
    fb_csr* A = new fb_csr();
    A->n = n;
    A->nnz = indptr[n];
    A->M = M;
    *out = A;
    return FB_OK;
}

void fb_csr_destroy(fb_ctx* ctx, fb_csr* A) {
    if (!A) return;
    fb_bsr_free(ctx, A->M);
    delete A;
}

int fb_csr_info(fb_ctx* ctx, fb_csr* A, int64_t* n, int64_t* nnz, int64_t* nb, int64_t* nnzb) {
    FB_CHECK_ARG(ctx, A != nullptr);
    if (n) *n = A->n;
    if (nnz) *nnz = A->nnz;
    if (nb) *nb = A->M->d.nb;
    if (nnzb) *nnzb = A->M->nnzb;
    return FB_OK;
}

static int upload_vec(fb_ctx* ctx, fb_csr* A, double2* dst, const double* src) {
    const size_t nbytes = sizeof(double) * (size_t)A->n;
    FB_HIP(ctx, hipMemsetAsync(dst, 0, sizeof(double2) * (size_t)A->M->d.nb, ctx->stream));
    if (src) return fb_copy_h2d(ctx, dst, src, nbytes);
    return FB_OK;
}

int fb_spmv(fb_ctx* ctx, fb_csr* A, const double* x_host, double* y_host) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, A && x_host && y_host);
    int rc = upload_vec(ctx, A, A->M->x, x_host);
    if (rc) return rc;
    rc = fb_bsr_spmv_dev(ctx, A->M, A->M->x, A->M->Ap);
    if (rc) return rc;
    return fb_copy_d2h(ctx, y_host, A->M->Ap, sizeof(double) * (size_t)A->n);
}

// y = A x on device-resident vectors (e.g. torch tensors of a distributed solver): no copies, no synchronisation
int fb_spmv_dev(fb_ctx* ctx, fb_csr* A, const double* x_dev, double* y_dev) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, A && x_dev && y_dev);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    return fb_bsr_spmv_dev(ctx, A->M, reinterpret_cast<const double2*>(x_dev), reinterpret_cast<double2*>(y_dev));
}

// ---------------------------------------------------------------------------------------------
// Vector side of the row-partitioned (coupled-window) PCG in its Chronopoulos-Gear form (feabas_amd/dist.py): between two
// halo exchanges a rank runs  update -> [exchange u] -> SpMV -> dots -> [all-reduce 3 scalars] -> scalars,  all on device
// pointers on the context's stream.  state (device double[8]) = {gamma, alpha, beta, r.r, breakdown flag, iterations, -, -}.
namespace {
constexpr int kCgWG = 1024;       // workgroups of the update / dot kernels; partial sums in a fixed order (reproducible)

__global__ __launch_bounds__(256) void cgcg_update_kernel(int64_t n, const double* __restrict__ state, const double* __restrict__ minv,
                                                          double* __restrict__ x, double* __restrict__ r, double* __restrict__ u,
                                                          const double* __restrict__ w, double* __restrict__ p, double* __restrict__ s) {
    const double alpha = state[1], beta = state[2];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double pi = u[i] + beta * p[i];
        const double si = w[i] + beta * s[i];
        const double ri = r[i] - alpha * si;
        p[i] = pi; s[i] = si;
        x[i] += alpha * pi;
        r[i] = ri;
        if (minv) u[i] = minv[i] * ri;                      // minv == NULL: the caller applies its own preconditioner to r
    }
}

__global__ __launch_bounds__(256) void cgcg_dots_kernel(int64_t n, const double* __restrict__ r, const double* __restrict__ u,
                                                        const double* __restrict__ w, double* __restrict__ part) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double ri = r[i], ui = u[i];
        a += ri * ui; b += w[i] * ui; c += ri * ri;
    }
    __shared__ double sh[3][4];
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); c += __shfl_down(c, off); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; sh[2][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x < 3) part[(size_t)threadIdx.x * kCgWG + blockIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void cgcg_dots_final_kernel(int nblk, const double* __restrict__ part, double* __restrict__ out3) {
    __shared__ double sh[256];
    for (int k = 0; k < 3; ++k) {
        double a = 0.0;
        for (int i = threadIdx.x; i < nblk; i += 256) a += part[(size_t)k * kCgWG + i];
        sh[threadIdx.x] = a;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) out3[k] = sh[0];
        __syncthreads();
    }
}

// dst[i] = src[idx[i]]: the entries of u a neighbouring rank needs, packed for one transfer
__global__ __launch_bounds__(256) void gather_f64_kernel(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ src, double* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}

// t3 = (r.u, w.u, r.r) summed over the ranks.  first != 0: the start of the iteration (alpha = gamma / delta, beta = 0)
__global__ void cgcg_scalars_kernel(const double* __restrict__ t3, double* __restrict__ state, int first) {
    const double g_new = t3[0], delta = t3[1];
    double alpha, beta;
    if (first || state[1] == 0.0) {
        // the start of the iteration, or the step after a dropped one: p = u, s = w (beta = 0), alpha = gamma / delta -- a
        // restart of the recurrence from the current iterate.  (A dropped step used to be absorbing: alpha = 0 made the next
        // denominator 0 and every later step was dropped too, up to maxiter.)  state[4] counts the dropped steps.
        beta = 0.0;
        alpha = delta > 0.0 ? g_new / delta : 0.0;
        if (first) { state[4] = 0.0; state[5] = 0.0; } else state[5] += 1.0;
        if (!(delta > 0.0) && g_new != 0.0) state[4] += 1.0;
    } else {
        const double g_old = state[0], a_old = state[1];
        beta = g_old != 0.0 ? g_new / g_old : 0.0;
        const double den = a_old != 0.0 ? delta - beta * g_new / a_old : 0.0;
        // den <= 0 (loss of positive definiteness in floating point, or an exactly converged residual): the step is dropped
        // instead of poisoning x with 0 / 0; the flag tells the host
        alpha = den > 0.0 ? g_new / den : 0.0;
        if (!(den > 0.0) && g_new != 0.0) state[4] += 1.0;
        if (!(den > 0.0)) beta = 0.0;
        state[5] += 1.0;
    }
    state[0] = g_new; state[1] = alpha; state[2] = beta; state[3] = t3[2];
}
}  // namespace

int fb_cgcg_update_dev(fb_ctx* ctx, int64_t n, const double* state, const double* minv, double* x, double* r, double* u, const double* w,
                       double* p, double* s) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, n >= 0 && state && (n == 0 || (x && r && u && w && p && s)));
    if (n == 0) return FB_OK;
    FB_PROF_B(ctx, "cgcg_update", (double)n * 8.0 * 12.0);
    hipLaunchKernelGGL(cgcg_update_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, kCgWG)), dim3(256), 0, ctx->stream, n, state, minv, x, r, u, w, p, s);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

// out3 = (r.u, w.u, r.r) of the local parts; scratch: device double[3 * 1024]
int fb_cgcg_dots_dev(fb_ctx* ctx, int64_t n, const double* r, const double* u, const double* w, double* scratch, double* out3) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, n >= 0 && scratch && out3 && (n == 0 || (r && u && w)));
    const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, kCgWG));
    FB_PROF_B(ctx, "cgcg_dots", (double)n * 8.0 * 3.0);
    hipLaunchKernelGGL(cgcg_dots_kernel, dim3(nblk), dim3(256), 0, ctx->stream, n, r, u, w, scratch);
    hipLaunchKernelGGL(cgcg_dots_final_kernel, dim3(1), dim3(256), 0, ctx->stream, nblk, scratch, out3);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

int fb_cgcg_scalars_dev(fb_ctx* ctx, const double* t3, double* state, int first) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, t3 && state);
    hipLaunchKernelGGL(cgcg_scalars_kernel, dim3(1), dim3(1), 0, ctx->stream, t3, state, first);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

int fb_gather_f64_dev(fb_ctx* ctx, int64_t n, const int32_t* idx, const double* src, double* dst) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, n >= 0 && (n == 0 || (idx && src && dst)));
    if (n == 0) return FB_OK;
    hipLaunchKernelGGL(gather_f64_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, kCgWG)), dim3(256), 0, ctx->stream, n, idx, src, dst);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

// The whole row-partitioned PCG of a coupled window behind one call (the loop of feabas_amd/dist.py in C++): per iteration
//   update (one pass over 12 vectors) -> pack + halo exchange (fb_sendrecv_dev) -> SpMV on [own | halo] -> three inner
//   products -> ONE all-reduce of 3 doubles (fb_allreduce_f64_dev) -> scalar recurrences on the device,
// everything on the context's stream; the host reads the 64-byte state every `check_every` iterations.
int fb_cgcg_solve_dev(fb_ctx* ctx, fb_comm* comm, fb_csr* rows, int64_t n_loc, int64_t n_halo, const double* b, const double* minv, double* x,
                      int nsend, const int* send_peer, const int64_t* send_off, const int32_t* send_idx, int nrecv, const int* recv_peer,
                      const int64_t* recv_off, double rtol, int maxiter, int check_every, int* iters, double* relres, double* bnorm) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, n_loc >= 0 && n_halo >= 0 && (n_loc + n_halo == 0 || (rows && rows->n == n_loc + n_halo)) && (n_loc == 0 || (b && minv && x)));     // a rank may own nothing: it still takes part in every reduction
    FB_CHECK_ARG(ctx, nsend >= 0 && nrecv >= 0 && (nsend == 0 || (send_peer && send_off && send_idx)) && (nrecv == 0 || (recv_peer && recv_off)));
    FB_CHECK_ARG(ctx, (nsend + nrecv == 0) || comm);
    FB_CHECK_ARG(ctx, nrecv == 0 || recv_off[nrecv] <= n_halo);
    if (check_every < 1) check_every = 8;
    FB_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t n = n_loc, m = n_loc + n_halo;
    const int64_t nsb = nsend ? send_off[nsend] : 0;
    // work vectors in one block: ext [m] (u at its head), r, p, s, w [n each], y [m], sendbuf, state[8], t3[4], scratch[3 * 1024]
    const size_t words = (size_t)m + 4 * (size_t)n + (size_t)m + (size_t)nsb + 8 + 4 + 3 * 1024 + 16;
    double* blk = nullptr;
    int rc = fb_malloc(ctx, words * sizeof(double), (void**)&blk);
    if (rc) return rc;
    struct Guard { fb_ctx* c; void* p; ~Guard() { fb_free(c, p); } } guard{ctx, blk};
    FB_HIP(ctx, hipMemsetAsync(blk, 0, words * sizeof(double), ctx->stream));
    double* ext = blk; double* r = ext + m + (m & 1); double* p = r + n; double* s = p + n; double* w = s + n;
    double* y = w + n + (n & 1); double* sb = y + m + (m & 1); double* state = sb + nsb + (nsb & 1); double* t3 = state + 8; double* scratch = t3 + 4;
    double* u = ext;
    if (n) {
        FB_HIP(ctx, hipMemsetAsync(x, 0, sizeof(double) * (size_t)n, ctx->stream));
        FB_HIP(ctx, hipMemcpyAsync(r, b, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    std::vector<const void*> sp((size_t)nsend); std::vector<int64_t> sbytes((size_t)nsend);
    std::vector<void*> rp((size_t)nrecv); std::vector<int64_t> rbytes((size_t)nrecv);
    for (int k = 0; k < nsend; ++k) { sp[k] = sb + send_off[k]; sbytes[k] = 8 * (send_off[k + 1] - send_off[k]); }
    for (int k = 0; k < nrecv; ++k) { rp[k] = ext + n + recv_off[k]; rbytes[k] = 8 * (recv_off[k + 1] - recv_off[k]); }
    auto product_and_dots = [&](int first) -> int {
        int e;
        if (nsend + nrecv) {
            if ((e = fb_gather_f64_dev(ctx, nsb, send_idx, u, sb))) return e;
            if ((e = fb_sendrecv_dev(ctx, comm, nsend, send_peer, sp.data(), sbytes.data(), nrecv, recv_peer, rp.data(), rbytes.data()))) return e;
        }
        if (m && (e = fb_bsr_spmv_dev(ctx, rows->M, reinterpret_cast<const double2*>(ext), reinterpret_cast<double2*>(y)))) return e;
        if (n) FB_HIP(ctx, hipMemcpyAsync(w, y, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
        if ((e = fb_cgcg_dots_dev(ctx, n, r, u, w, scratch, t3))) return e;
        if (comm && (e = fb_allreduce_f64_dev(ctx, comm, t3, t3, 3, FB_REDUCE_SUM))) return e;
        return fb_cgcg_scalars_dev(ctx, t3, state, first);
    };
    // u = minv r through the update kernel with alpha = beta = 0 (state is zero, p = s = 0)
    if ((rc = fb_cgcg_update_dev(ctx, n, state, minv, x, r, u, w, p, s))) return rc;
    if ((rc = product_and_dots(1))) return rc;
    double hs[8];
    FB_HIP(ctx, hipMemcpyAsync(hs, state, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const double bb = hs[3];
    if (bnorm) *bnorm = std::sqrt(bb);
    if (iters) *iters = 0;
    if (relres) *relres = 0.0;
    if (bb == 0.0 || maxiter == 0) return FB_OK;
    if (!(bb == bb) || bb < 0) return fb_fail(ctx, FB_ERR_BREAKDOWN, "fb_cgcg_solve_dev: ||b||^2 = %g", bb);
    const int limit = maxiter > 0 ? maxiter : 100 * 1000;
    int it = 0;
    double rel = 1.0, dropped_seen = 0.0;
    while (it < limit) {
        const int batch = std::min(check_every, limit - it);
        for (int k = 0; k < batch; ++k) {
            if ((rc = fb_cgcg_update_dev(ctx, n, state, minv, x, r, u, w, p, s))) return rc;
            if ((rc = product_and_dots(0))) return rc;
        }
        it += batch;
        FB_HIP(ctx, hipMemcpyAsync(hs, state, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        rel = std::sqrt(hs[3] / bb);
        if (iters) *iters = it;
        if (relres) *relres = rel;
        if (!(rel == rel)) return fb_fail(ctx, FB_ERR_BREAKDOWN, "fb_cgcg_solve_dev: the residual is not finite after %d iterations", it);
        if (rel <= rtol) return FB_OK;
        // dropped steps restart the recurrence (cgcg_scalars_kernel); a system that keeps dropping them is not positive definite
        if (hs[4] - dropped_seen >= (double)batch) return fb_fail(ctx, FB_ERR_BREAKDOWN, "fb_cgcg_solve_dev: every step of the last %d was dropped (p^T A p <= 0): the system is not positive definite", batch);
        dropped_seen = hs[4];
    }
    return FB_ERR_NOCONV;
}

int fb_pcg_csr(fb_ctx* ctx, fb_csr* A, const double* b, double* x, int use_x0, double rtol, double atol, int maxiter, int precond,
               int* iters, double* relres) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, A && b && x);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    int rc = upload_vec(ctx, A, A->M->b, b);
    if (rc) return rc;
    rc = upload_vec(ctx, A, A->M->x, use_x0 ? x : nullptr);
    if (rc) return rc;
    rc = fb_bsr_setup_jacobi(ctx, A->M, precond);
    if (rc) return rc;
    rc = fb_bsr_pcg_dev(ctx, A->M, rtol, atol, maxiter, 0, iters, relres);
    if (rc && rc != FB_ERR_NOCONV) return rc;
    { const int rc_ = fb_copy_d2h(ctx, x, A->M->x, sizeof(double) * (size_t)A->n); if (rc_) return rc_; }
    return rc;
}

int fb_pcg_fixed_iters(fb_ctx* ctx, fb_csr* A, const double* b_host, int iters, double* relres) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, A && iters > 0);
    int rc;
    if (b_host) {
        rc = upload_vec(ctx, A, A->M->b, b_host);
        if (rc) return rc;
        rc = fb_bsr_setup_jacobi(ctx, A->M, 1);
        if (rc) return rc;
    }
    FB_HIP(ctx, hipMemsetAsync(A->M->x, 0, sizeof(double2) * (size_t)A->M->d.nb, ctx->stream));
    int done = 0;
    return fb_bsr_pcg_dev(ctx, A->M, 0.0, 0.0, 0, iters, &done, relres);
}

int fb_pcg(fb_ctx* ctx, int64_t n, const int64_t* indptr, const int32_t* idx, const double* val, const double* b, double* x,
           int use_x0, double rtol, double atol, int maxiter, int precond, int symmetrize, int* iters, double* relres) {
    FB_LOCK(ctx);
    fb_csr* A = nullptr;
    int rc = fb_csr_upload(ctx, n, indptr, idx, val, symmetrize, &A);
    if (rc) return rc;
    rc = fb_pcg_csr(ctx, A, b, x, use_x0, rtol, atol, maxiter, precond, iters, relres);
    fb_csr_destroy(ctx, A);
    return rc;
}

}  // extern "C"
