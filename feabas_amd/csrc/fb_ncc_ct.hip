// NCC path, streaming class, compile-time mixed-radix shapes: the block classes of the alignment matcher
// (configs/default_alignment_configs.yaml:16-23: spacings [400, 100] x 0.7 -> blocks of 280^2 and 70^2, padded FFTs of
// 576^2 = (16 x 9 x 4)^2 and 144^2 = (16 x 9)^2; matcher.py:59-62) and of the README stitching example (150 x 144,
// 135 x 150 -> run at 160 x 144 / 144 x 160).  None of them is a power of two (fb_ncc_p2.inc) and none fits the on-chip
// class (144 x 146 x 8 B > 160 KiB of LDS), so they used to fall to the run-time mixed-radix kernels of fb_ncc.hip,
// whose index arithmetic costs more than the butterflies.  Here the plan of every length is unrolled at compile time
// (fb_fft3.h, plan family 17: radix 16, 9, 8, then 4 / 2 / 5 / 3) with packed-FP32 butterflies; the three passes, the
// T / V layouts and the reductions are those of the generic kernels (matcher.xcorr_fft, feabas/matcher.py:22-135).
#include "fb_ncc_stream.h"
#include "fb_fft3.h"

#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr int kCtFam = 17;          // plan family (p3_pick)
constexpr int kCtThreads = 256;

// row pitch of the LDS work arrays in float2: odd, so that the rows of a tile start on different banks
__host__ __device__ constexpr int ct_pitch(int F) { return F | 1; }
// column pairs per workgroup of the column pass: about 2 K points per transform set
__host__ __device__ constexpr int ct_np(int FH) { return FH >= 1024 ? 1 : 1024 / FH; }
// rows per tile of the row passes: the largest power of two <= 16 with at most 8 K points per tile
__host__ __device__ constexpr int ct_tr(int FW) { int t = 16; while (t > 1 && t * FW > 8192) t >>= 1; return t; }

template <int F>
__device__ __forceinline__ void ct_tables(f2* tw, short* pos, const float2* __restrict__ tw_g, int tid, int nt) {
    for (int i = tid; i < F; i += nt) { tw[i] = (f2){tw_g[i].x, tw_g[i].y}; if (pos) pos[i] = (short)p3_pos<F, kCtFam>(i); }
}

// ---- rows: packed R2C (z = img0 + i img1) of one tile of TR non-zero rows, spectra split and stored transposed as
// interleaved column pairs T[n][kx / 2][y][kx & 1]
// AFF (image 1 gathered through a per-block affine map) is a template parameter and the per-row clamps / validity are taken once
// per row, like in ncc_rows_p2 (fb_ncc_p2.inc): scalar instructions per item 1 900 -> see profiles/README_r06.md
template <int FW, bool AFF = false>
__global__ __launch_bounds__(kCtThreads) void ncc_rows_ct(const StreamGeom g, const float2* __restrict__ tw_g, float2* __restrict__ T0, float2* __restrict__ T1) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    constexpr int pitch = ct_pitch(FW), Sw = FW / 2 + 1, NX = (FW + kCtThreads - 1) / kCtThreads, TR = ct_tr(FW);      // g.TR == TR (fb_ncc_ct_tr)
    f2* G = reinterpret_cast<f2*>(lds);
    f2* tw = G + (size_t)TR * pitch;
    short* posW = reinterpret_cast<short*>(tw + FW);
    const int n = blockIdx.y, y0 = blockIdx.x * TR;
    const int tid = threadIdx.x;
    int h0 = g.H0, w0 = g.W0, h1 = g.H1, w1 = g.W1, ox0 = 0, oy0 = 0, ox1 = 0, oy1 = 0;
    int p0 = w0, p1 = w1, my0 = h0 - 1, mx0 = w0 - 1, my1 = h1 - 1, mx1 = w1 - 1;
    const float* s0; const float* s1;
    if (g.blk) {
        const int* d = g.blk + (size_t)n * kBlkStride;
        s0 = g.img0 + (size_t)d[0] * g.IH0 * g.IW0;
        s1 = g.img1 + (size_t)d[0] * g.IH1 * g.IW1;
        ox0 = d[1]; oy0 = d[2]; h0 = d[3]; w0 = d[4]; ox1 = d[5]; oy1 = d[6]; h1 = d[7]; w1 = d[8];
        p0 = g.IW0; p1 = g.IW1; my0 = g.IH0 - 1; mx0 = g.IW0 - 1; my1 = g.IH1 - 1; mx1 = g.IW1 - 1;
    } else {
        s0 = g.img0 + (size_t)n * h0 * w0;
        s1 = g.img1 + (size_t)n * h1 * w1;
    }
    ct_tables<FW>(tw, posW, tw_g, tid, kCtThreads);
    // packed load, zero padded, branch-free; columns beyond both crops are never read
    const int wmax = max(w0, w1);
    __shared__ __attribute__((aligned(8))) float s_red[2 * (kCtThreads / 64)];
    float m0 = 0.f, m1 = 0.f;
    uint32_t ro0[TR], ro1[TR];                                 // clamped source rows as byte offsets from the image base (uniform)
    bool ry0[TR], ry1[TR];                                     // the row exists in the crop and in the image (uniform)
#pragma unroll
    for (int r = 0; r < TR; ++r) {
        const int y = y0 + r, gy0 = oy0 + y, gy1 = oy1 + y;
        ro0[r] = (uint32_t)(min(max(gy0, 0), my0) * p0) * 4u;
        ro1[r] = (uint32_t)(min(max(gy1, 0), my1) * p1) * 4u;
        ry0[r] = y < h0 && gy0 >= 0 && gy0 <= my0;
        ry1[r] = AFF ? (y < h1) : (y < h1 && gy1 >= 0 && gy1 <= my1);
    }
    const char* b0 = reinterpret_cast<const char*>(s0);
    const char* b1 = reinterpret_cast<const char*>(s1);
#pragma unroll
    for (int c = 0; c < NX; ++c) {
        const int x = c * kCtThreads + tid;
        const int gx0 = ox0 + x, gx1 = ox1 + x;
        const bool vx0 = x < w0 && gx0 >= 0 && gx0 <= mx0, vx1 = x < w1 && gx1 >= 0 && gx1 <= mx1;
        const uint32_t cx0 = (uint32_t)min(max(gx0, 0), mx0) * 4u, cx1 = (uint32_t)min(max(gx1, 0), mx1) * 4u;
        float a[TR], b[TR];
        if (c * kCtThreads < wmax) {
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                a[r] = *reinterpret_cast<const float*>(b0 + (ro0[r] + cx0));
                if (AFF) b[r] = fb_sample_affine(s1, g.IH1, g.IW1, g.aff + (size_t)n * FB_AFFINE_STRIDE, min(x, w1 - 1), min(y0 + r, h1 - 1));
                else b[r] = *reinterpret_cast<const float*>(b1 + (ro1[r] + cx1));
            }
        }
        if (x < FW) {
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const bool in = c * kCtThreads < wmax;
                const bool v0 = in && vx0 && ry0[r];
                const bool v1 = in && (AFF ? (x < w1 && ry1[r]) : (vx1 && ry1[r]));
                const float va = v0 ? a[r] : 0.f, vb = v1 ? b[r] : 0.f;
                m0 = fmaxf(m0, fabsf(va)); m1 = fmaxf(m1, fabsf(vb));
                G[r * pitch + x] = (f2){va, vb};
            }
        }
    }
    wg_max2_post(m0, m1, s_red);
    __syncthreads();
    const float2 mm = wg_max2_read(s_red);
    const float2 sc = pack_scales(mm.x, mm.y);                // (fb_ldsfft.h) != 1 only for a tile one side of which is almost blank
    if (sc.x != 1.f || sc.y != 1.f) {
        for (int i = tid; i < TR * FW; i += kCtThreads) { f2* z = G + (i / FW) * pitch + (i % FW); *z = (f2){z->x * sc.x, z->y * sc.y}; }
        __syncthreads();
    }
    p3_fft<FW, kCtFam, false, false>(G, TR, pitch, tw);
    const size_t tbase = (size_t)n * g.Kp * g.Hs * 2;
    const float ia = mm.x > 0.f ? 0.5f / sc.x : 0.f, ib = mm.y > 0.f ? 0.5f / sc.y : 0.f;
              // powers of two: exact; an image that is exactly zero on the tile gets an exactly zero spectrum (the split leaves rounding noise of the other image there)
    for (int t = tid; t < 2 * g.Kp * TR; t += kCtThreads) {
        const int c = t & 1, r = (t >> 1) & (TR - 1), kp = t / (2 * TR);
        const int kx = 2 * kp + c;
        float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
        if (kx < Sw) {
            const f2 zk = G[r * pitch + posW[kx]];
            const f2 zn = G[r * pitch + posW[kx == 0 ? 0 : FW - kx]];
            a = make_float2(ia * (zk.x + zn.x), ia * (zk.y - zn.y));
            b = make_float2(ib * (zk.y + zn.y), -ib * (zk.x - zn.x));
        }
        const size_t o = tbase + ((size_t)kp * g.Hs + y0 + r) * 2 + c;
        T0[o] = a;
        T1[o] = b;
    }
}

// ---- cols: one workgroup = NP adjacent kx column pairs of one block pair: 4 NP length-FH transforms forward
// (column pair x image 0/1 x column 0/1), products, inverse
template <int FH>
__global__ __launch_bounds__(kCtThreads) void ncc_cols_ct(const StreamGeom g, const float2* __restrict__ tw_g, const float2* __restrict__ T0,
                                                          const float2* __restrict__ T1, float2* __restrict__ V0, float2* __restrict__ V1) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    constexpr int pitch = ct_pitch(FH), NP = ct_np(FH), NE = NP * 2 * FH, NU = (NE + kCtThreads - 1) / kCtThreads;
    f2* U = reinterpret_cast<f2*>(lds);         // [NP][4][pitch]: (img0,c0) (img0,c1) (img1,c0) (img1,c1) per column pair
    f2* tw = U + (size_t)4 * NP * pitch;
    int bx = blockIdx.x, by = blockIdx.y;
    if (g.xg_per8 > 0) {                       // adjacent column groups write adjacent runs of V: one contiguous eighth of the items per XCD
        const int w = (int)(blockIdx.x & 7) * g.xg_per8 + (int)(blockIdx.x >> 3);
        if (w >= g.xg_gx * g.xg_gy) return;
        bx = w % g.xg_gx; by = w / g.xg_gx;
    }
    const int kp0 = bx * NP, n = by;
    const int tid = threadIdx.x;
    const int Hs = g.Hs, Kp = g.Kp;
    ct_tables<FH>(tw, nullptr, tw_g, tid, kCtThreads);
    // e = (j, y, c): a column pair is one contiguous run of 2 Hs float2 in T
#pragma unroll 4
    for (int u = 0; u < NU; ++u) {
        const int e = u * kCtThreads + tid;
        if (e < NE) {
            const int j = e / (2 * FH), rem = e - j * (2 * FH), y = rem >> 1, c = rem & 1;
            const bool in = y < Hs && kp0 + j < Kp;
            const size_t o = (((size_t)n * Kp + min(kp0 + j, Kp - 1)) * Hs + min(y, Hs - 1)) * 2 + c;
            const float2 a = T0[o], b = T1[o];
            U[(4 * j + c) * pitch + y] = in ? (f2){a.x, a.y} : (f2){0.f, 0.f};
            U[(4 * j + 2 + c) * pitch + y] = in ? (f2){b.x, b.y} : (f2){0.f, 0.f};
        }
    }
    __syncthreads();
    p3_fft<FH, kCtFam, false, false>(U, 4 * NP, pitch, tw);
    const bool wq = g.want_q != 0;
    for (int e = tid; e < NE; e += kCtThreads) {
        const int j = e / (2 * FH), rem = e - j * (2 * FH), y = rem >> 1, c = rem & 1;
        f2* pa = U + (4 * j + c) * pitch + y;
        f2* pb = pa + 2 * pitch;
        const f2 a = *pa, b = *pb;
        *pa = pk_cmulc(b, a);                                              // conj(F0) F1
        *pb = wq ? pk_cmul(a, b) : (f2){0.f, 0.f};                         // F0 F1 (mirror confidence)
    }
    __syncthreads();
    p3_fft<FH, kCtFam, true, false>(U, 4 * NP, pitch, tw);
    for (int e = tid; e < NE; e += kCtThreads) {
        const int j = e / (2 * FH), rem = e - j * (2 * FH), y = rem >> 1, c = rem & 1;
        if (kp0 + j >= Kp) continue;
        const size_t o = (((size_t)n * Kp + kp0 + j) * FH + y) * 2 + c;
        const f2 p = U[(4 * j + c) * pitch + y];
        V0[o] = make_float2(p.x, p.y);
        if (wq) { const f2 q = U[(4 * j + 2 + c) * pitch + y]; V1[o] = make_float2(q.x, q.y); }
    }
}

// ---- inverse rows.  ct9 == nullptr: tile of TRI consecutive rows, reduce into part[n][tile].
// ct9 != nullptr (sub-pixel neighbours): the 3 rows around the peak of block n, outputs ct9[n][3][3].
template <int FW>
__global__ __launch_bounds__(kCtThreads) void ncc_inv_ct(const StreamGeom g, const float2* __restrict__ tw_g, const float2* __restrict__ V0,
                                                         const float2* __restrict__ V1, PeakPartial* __restrict__ part,
                                                         const PeakPartial* __restrict__ part_in, int nparts, float* __restrict__ ct9) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    constexpr int pitch = ct_pitch(FW), Sw = FW / 2 + 1, nmir = FW - Sw;
    const int Fh = g.Fh;
    const bool neigh = ct9 != nullptr;
    const int TRI = neigh ? 4 : g.TRI;
    f2* G = reinterpret_cast<f2*>(lds);
    f2* tw = G + (size_t)TRI * pitch;
    short* posW = reinterpret_cast<short*>(tw + FW);
    __shared__ float sv[kCtThreads / 64]; __shared__ int si[kCtThreads / 64]; __shared__ float sm[kCtThreads / 64];
    __shared__ double ssum[kCtThreads / 64]; __shared__ double ssq[kCtThreads / 64];
    __shared__ int s_peak;
    const int n = blockIdx.y;
    const int tid = threadIdx.x;
    ct_tables<FW>(tw, posW, tw_g, tid, kCtThreads);
    const int y0 = blockIdx.x * TRI;
    int py = 0, px = 0;
    if (neigh) {
        if (tid == 0) {
            const PeakPartial* p = part_in + (size_t)n * nparts;
            float v = p[0].vmax; int iv = p[0].imax;
            for (int c = 1; c < nparts; ++c) peak_merge(v, iv, p[c].vmax, p[c].imax);
            if (iv == 0x7fffffff) iv = 0;
            s_peak = iv;
        }
        __syncthreads();
        py = s_peak / FW; px = s_peak - py * FW;
    }
    __syncthreads();
    const bool wq = g.want_q != 0;
    const size_t vb = (size_t)n * g.Kp * Fh * 2;
    const int nitems = 2 * g.Kp * TRI;
    for (int tb = 0; tb < nitems; tb += 8 * kCtThreads) {
        float2 pk[8], qk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = min(tb + u * kCtThreads + tid, nitems - 1);
            const int c = t & 1, r = (t >> 1) & (TRI - 1), kp = t / (2 * TRI);
            int y = y0 + r;
            bool ok = y < Fh;
            if (neigh) { y = (py + r - 1 + Fh) % Fh; ok = r < 3; }
            y = min(y, Fh - 1);
            const size_t o = vb + ((size_t)kp * Fh + y) * 2 + c;
            pk[u] = V0[o];
            qk[u] = wq ? V1[o] : make_float2(0.f, 0.f);
            if (!ok) { pk[u] = make_float2(0.f, 0.f); qk[u] = make_float2(0.f, 0.f); }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = tb + u * kCtThreads + tid;
            const int c = t & 1, r = (t >> 1) & (TRI - 1), kp = t / (2 * TRI);
            const int kx = 2 * kp + c;
            if (t < nitems && kx < Sw) {
                // W = P + i Q on the half spectrum, conj(P) + i conj(Q) on the mirror half (irfft ignores the imaginary part of the self-conjugate bins)
                const bool self = (kx == 0) || (2 * kx == FW);
                G[r * pitch + posW[kx]] = self ? (f2){pk[u].x, qk[u].x} : (f2){pk[u].x - qk[u].y, pk[u].y + qk[u].x};
                if (kx >= 1 && kx <= nmir) G[r * pitch + posW[FW - kx]] = (f2){pk[u].x + qk[u].y, qk[u].x - pk[u].y};
            }
        }
    }
    __syncthreads();
    p3_fft<FW, kCtFam, true, false>(G, TRI, pitch, tw);
    if (neigh) {
        if (tid < 9) ct9[(size_t)n * 9 + tid] = G[(tid / 3) * pitch + (px + (tid % 3 - 1) + FW) % FW].x;
        return;
    }
    float v = -INFINITY; int iv = 0x7fffffff; float mm = 0.f;
    double s = 0.0, ss = 0.0;
    for (int r = 0; r < TRI; ++r) {
        const int y = y0 + r;
        if (y >= Fh) break;
        for (int x = tid; x < FW; x += kCtThreads) {
            const f2 c = G[r * pitch + x];
            if (c.x > v) { v = c.x; iv = y * FW + x; }          // a thread meets its elements in increasing flat index: first maximum
            mm = fmaxf(mm, fabsf(c.y));
            if (g.want_std) { s += (double)c.x; ss += (double)c.x * (double)c.x; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float v2 = __shfl_down(v, off);
        const int i2 = __shfl_down(iv, off);
        peak_merge(v, iv, v2, i2);
        mm = fmaxf(mm, __shfl_down(mm, off));
        if (g.want_std) { s += __shfl_down(s, off); ss += __shfl_down(ss, off); }
    }
    const int wave = tid >> 6, lane = tid & 63;
    if (lane == 0) { sv[wave] = v; si[wave] = iv; sm[wave] = mm; ssum[wave] = s; ssq[wave] = ss; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kCtThreads / 64; ++w) {
            peak_merge(v, iv, sv[w], si[w]);
            mm = fmaxf(mm, sm[w]);
            s += ssum[w]; ss += ssq[w];
        }
        PeakPartial p; p.vmax = v; p.imax = iv; p.mmax = mm; p.pad_ = 0; p.sum = s; p.sumsq = ss;
        part[(size_t)n * gridDim.x + blockIdx.x] = p;
    }
}

// full-length twiddle tables tw[i] = exp(-2 pi i / n), one per (device, length), owned by the process
std::map<std::pair<int, int>, float2*> g_ct_tables;

int ct_table(fb_ctx* ctx, int n, const float2** out) {
    static std::mutex mtx;
    std::lock_guard<std::mutex> lk(mtx);
    auto key = std::make_pair(ctx->device, n);
    auto it = g_ct_tables.find(key);
    if (it == g_ct_tables.end()) {
        std::vector<float2> h((size_t)n);
        for (int k = 0; k < n; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)n;
            h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        float2* d = nullptr;
        FB_HIP(ctx, hipMalloc((void**)&d, sizeof(float2) * (size_t)n));
        FB_HIP(ctx, hipMemcpy(d, h.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
        it = g_ct_tables.emplace(key, d).first;
    }
    *out = it->second;
    return FB_OK;
}

size_t lds_rows(int F, int TR) { return ((size_t)TR * ct_pitch(F) + F) * sizeof(float2) + (size_t)(F + 2) / 2 * 2 * sizeof(short); }
size_t lds_cols(int F) { return ((size_t)4 * ct_np(F) * ct_pitch(F) + F) * sizeof(float2); }

template <int F>
struct CtLaunch {
    static int rows(fb_ctx* ctx, dim3 grid, const StreamGeom& g, const float2* tw, float2* T0, float2* T1) {
        const size_t lds = lds_rows(F, g.TR);
        FB_HIP(ctx, hipFuncSetAttribute((const void*)ncc_rows_ct<F, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        FB_HIP(ctx, hipFuncSetAttribute((const void*)ncc_rows_ct<F, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (g.aff) hipLaunchKernelGGL((ncc_rows_ct<F, true>), grid, dim3(kCtThreads), lds, ctx->stream, g, tw, T0, T1);
        else hipLaunchKernelGGL((ncc_rows_ct<F, false>), grid, dim3(kCtThreads), lds, ctx->stream, g, tw, T0, T1);
        return FB_OK;
    }
    static int cols(fb_ctx* ctx, dim3 grid, const StreamGeom& g, const float2* tw, const float2* T0, const float2* T1, float2* V0, float2* V1) {
        const size_t lds = lds_cols(F);
        FB_HIP(ctx, hipFuncSetAttribute((const void*)ncc_cols_ct<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ncc_cols_ct<F>, grid, dim3(kCtThreads), lds, ctx->stream, g, tw, T0, T1, V0, V1);
        return FB_OK;
    }
    static int inv(fb_ctx* ctx, dim3 grid, const StreamGeom& g, const float2* tw, const float2* V0, const float2* V1, PeakPartial* part,
                   const PeakPartial* part_in, int nparts, float* ct9) {
        const size_t lds = lds_rows(F, std::max(g.TRI, 4));
        FB_HIP(ctx, hipFuncSetAttribute((const void*)ncc_inv_ct<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(ncc_inv_ct<F>, grid, dim3(kCtThreads), lds, ctx->stream, g, tw, V0, V1, part, part_in, nparts, ct9);
        return FB_OK;
    }
};

// the instantiated lengths: 2^a, 3 2^a, 5 2^a, 9 2^a from 64 to 2048, 15 2^a and 27 2^a from 108 (any padded need from 64 to
// 2048 is within 1.2 x of one of them)
#define FB_CT_SWITCH(F, CALL)                                                                                   \
    switch (F) {                                                                                                \
        case 64: rc = CtLaunch<64>::CALL; break;                                                                \
        case 72: rc = CtLaunch<72>::CALL; break;                                                                \
        case 80: rc = CtLaunch<80>::CALL; break;                                                                \
        case 96: rc = CtLaunch<96>::CALL; break;                                                                \
        case 108: rc = CtLaunch<108>::CALL; break;                                                              \
        case 120: rc = CtLaunch<120>::CALL; break;                                                              \
        case 128: rc = CtLaunch<128>::CALL; break;                                                              \
        case 144: rc = CtLaunch<144>::CALL; break;                                                              \
        case 160: rc = CtLaunch<160>::CALL; break;                                                              \
        case 192: rc = CtLaunch<192>::CALL; break;                                                              \
        case 216: rc = CtLaunch<216>::CALL; break;                                                              \
        case 240: rc = CtLaunch<240>::CALL; break;                                                              \
        case 256: rc = CtLaunch<256>::CALL; break;                                                              \
        case 288: rc = CtLaunch<288>::CALL; break;                                                              \
        case 320: rc = CtLaunch<320>::CALL; break;                                                              \
        case 384: rc = CtLaunch<384>::CALL; break;                                                              \
        case 432: rc = CtLaunch<432>::CALL; break;                                                              \
        case 480: rc = CtLaunch<480>::CALL; break;                                                              \
        case 512: rc = CtLaunch<512>::CALL; break;                                                              \
        case 576: rc = CtLaunch<576>::CALL; break;                                                              \
        case 640: rc = CtLaunch<640>::CALL; break;                                                              \
        case 768: rc = CtLaunch<768>::CALL; break;                                                              \
        case 864: rc = CtLaunch<864>::CALL; break;                                                              \
        case 960: rc = CtLaunch<960>::CALL; break;                                                              \
        case 1024: rc = CtLaunch<1024>::CALL; break;                                                            \
        case 1152: rc = CtLaunch<1152>::CALL; break;                                                            \
        case 1280: rc = CtLaunch<1280>::CALL; break;                                                            \
        case 1536: rc = CtLaunch<1536>::CALL; break;                                                            \
        case 1728: rc = CtLaunch<1728>::CALL; break;                                                            \
        case 1920: rc = CtLaunch<1920>::CALL; break;                                                            \
        case 2048: rc = CtLaunch<2048>::CALL; break;                                                            \
        default: rc = fb_fail(ctx, FB_ERR_ARG, "ncc: no compile-time plan for length %d", (F)); break;          \
    }

const int kCtLens[] = {64, 72, 80, 96, 108, 120, 128, 144, 160, 192, 216, 240, 256, 288, 320, 384, 432, 480, 512, 576, 640, 768, 864, 960, 1024, 1152, 1280, 1536, 1728, 1920, 2048};

}  // namespace

bool fb_ncc_ct_len(int n) {
    for (int v : kCtLens) if (v == n) return true;
    return false;
}

int fb_ncc_ct_up(int need, int ref) {
    for (int v : kCtLens) if (v >= need) return 5 * v <= 6 * ref ? v : 0;
    return 0;
}

int fb_ncc_ct_tr(int Fw) { return ct_tr(Fw); }

int fb_ncc_ct_run(fb_ctx* ctx, const StreamGeom& g, int nb, float2* T0, float2* T1, float2* V0, float2* V1, PeakPartial* part,
                  int ntiles, float* ct9, int subpixel, double in_bytes) {
    const int Fh = g.Fh, Fw = g.Fw;
    const float2 *twW = nullptr, *twH = nullptr;
    int rc = ct_table(ctx, Fw, &twW);
    if (rc) return rc;
    rc = ct_table(ctx, Fh, &twH);
    if (rc) return rc;
    const double nq = g.want_q ? 2.0 : 1.0;
    {
        FB_PROF_B(ctx, "ncc_stream_rows", nb * (in_bytes + 16.0 * g.Sw * g.Hs));
        FB_CT_SWITCH(Fw, rows(ctx, dim3(g.Hs / g.TR, nb), g, twW, T0, T1));
        if (rc) return rc;
    }
    {
        FB_PROF_B(ctx, "ncc_stream_cols", (double)nb * g.Sw * (16.0 * g.Hs + 8.0 * nq * Fh));
        const int np = Fh >= 1024 ? 1 : 1024 / Fh;
        static const int xcd_mask = [] { const char* e = getenv("FEABAS_HIP_P2_XCD"); return e ? atoi(e) : 7; }();
        StreamGeom gc = g;
        dim3 grid((g.Kp + np - 1) / np, nb);
        if ((xcd_mask & 2) && (long long)grid.x * grid.y >= 64) {
            gc.xg_gx = (int)grid.x; gc.xg_gy = (int)grid.y; gc.xg_per8 = (int)(((long long)grid.x * grid.y + 7) / 8);
            grid = dim3(8 * gc.xg_per8, 1);
        }
        FB_CT_SWITCH(Fh, cols(ctx, grid, gc, twH, T0, T1, V0, V1));
        if (rc) return rc;
    }
    {
        FB_PROF_B(ctx, "ncc_stream_inv", (double)nb * g.Sw * 8.0 * nq * Fh);
        FB_CT_SWITCH(Fw, inv(ctx, dim3(ntiles, nb), g, twW, V0, V1, part, nullptr, 0, nullptr));
        if (rc) return rc;
    }
    if (subpixel) {
        FB_PROF(ctx, "ncc_stream_neighbors");
        FB_CT_SWITCH(Fw, inv(ctx, dim3(1, nb), g, twW, V0, V1, nullptr, part, ntiles, ct9));
        if (rc) return rc;
    }
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}
