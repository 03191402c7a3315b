// NCC path, streaming class: batched 2-D R2C / C2R through rocFFT with hand-written
// pad-load, spectral-multiply and peak-reduction kernels around it.
// Replaces matcher.xcorr_fft (feabas/matcher.py:22-135); numerics per SURVEY.md A.1.
#include "fb_common.h"
#include "fb_ldsfft.h"
#include "fb_fft2.h"
#include "fb_ncc_stream.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr int kPeakChunks = 64;   // partial reductions per surface
constexpr int kThreads = 256;

// img [n][H][W] -> R [n][Fh][Fw], zero padded bottom/right (rfft2(s=...), matcher.py:63-64)
__global__ void ncc_pad_load(const float* __restrict__ img0, const float* __restrict__ img1,
                             float* __restrict__ R, int NC, int H0, int W0, int H1, int W1, int Fh, int Fw) {
    // NC = slots per side in R (the plan batch), gridDim.y = images actually present
    const int sel = blockIdx.z;
    const int n = blockIdx.y;
    const float* src = sel ? img1 : img0;
    const int H = sel ? H1 : H0, W = sel ? W1 : W0;
    src += (size_t)n * H * W;
    float* dst = R + ((size_t)sel * NC + n) * Fh * Fw;
    const int total = Fh * Fw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / Fw, x = i - y * Fw;
        float v = 0.f;
        if (y < H && x < W) v = src[(size_t)y * W + x];
        dst[i] = v;
    }
}

// crop mode: block n of side `sel` is the h x w window at (x,y) of image blk[n][0] of that side's image stack,
// zero outside the image (dal.StreamLoader fillval=0, matcher.py:342-343), zero padded to Fh x Fw.
__global__ void ncc_crop_load(const float* __restrict__ imgs0, const float* __restrict__ imgs1, const int* __restrict__ blk,
                              float* __restrict__ R, int N, int IH0, int IW0, int IH1, int IW1, int Fh, int Fw) {
    const int sel = blockIdx.z;
    const int n = blockIdx.y;
    const int* d = blk + (size_t)n * kBlkStride;
    const int IH = sel ? IH1 : IH0, IW = sel ? IW1 : IW0;
    const float* src = (sel ? imgs1 : imgs0) + (size_t)d[0] * IH * IW;
    const int ox = d[sel ? 5 : 1], oy = d[sel ? 6 : 2], h = d[sel ? 7 : 3], w = d[sel ? 8 : 4];
    float* dst = R + ((size_t)sel * N + n) * Fh * Fw;
    const int total = Fh * Fw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / Fw, x = i - y * Fw;
        float v = 0.f;
        if (y < h && x < w) {
            const int gy = oy + y, gx = ox + x;
            if (gy >= 0 && gy < IH && gx >= 0 && gx < IW) v = src[(size_t)gy * IW + gx];
        }
        dst[i] = v;
    }
}

__device__ __forceinline__ float2 cmul_conj(float2 a, float2 b) {   // conj(a) * b
    return make_float2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// P = sum_c conj(F0) F1 (matcher.py:65-67), Q = sum_c F0 F1 (matcher.py:114-116).
// C == 1: in place (P over F0, Q over F1).  C > 1: compact outputs Pout/Qout.
__global__ void ncc_spectral_mul(const float2* F0, const float2* F1, float2* Pout,
                                 float2* Qout, long long per_img, int N, int C, int want_q) {
    const long long total = per_img * N;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / per_img, k = i - n * per_img;
        float2 p = make_float2(0.f, 0.f), q = make_float2(0.f, 0.f);
        for (int c = 0; c < C; ++c) {
            const float2 a = F0[(n * C + c) * per_img + k];
            const float2 b = F1[(n * C + c) * per_img + k];
            const float2 pc = cmul_conj(a, b);
            p.x += pc.x; p.y += pc.y;
            if (want_q) {
                const float2 qc = cmul(a, b);
                q.x += qc.x; q.y += qc.y;
            }
        }
        Pout[i] = p;
        if (want_q) Qout[i] = q;
    }
}

// stage 1: per (image, chunk) partial of max/argmax C, max|Cm|, sum, sumsq
__global__ void ncc_peak_partial(const float* __restrict__ Csurf, const float* __restrict__ Msurf,
                                 PeakPartial* __restrict__ part, int F, int want_m, int want_std) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    const float* c = Csurf + (size_t)n * F;
    const float* m = want_m ? Msurf + (size_t)n * F : nullptr;
    const int per = (F + gridDim.x - 1) / gridDim.x;
    const int lo = chunk * per, hi = min(F, lo + per);
    float v = -INFINITY; int iv = 0x7fffffff; float mm = 0.f;
    double s = 0.0, ss = 0.0;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float x = c[i];
        if (x > v) { v = x; iv = i; }
        if (want_m) mm = fmaxf(mm, fabsf(m[i]));
        if (want_std) { s += (double)x; ss += (double)x * (double)x; }
    }
    // wave reduction (64 lanes), then across the 4 waves through LDS
    for (int off = 32; off > 0; off >>= 1) {
        const float v2 = __shfl_down(v, off);
        const int i2 = __shfl_down(iv, off);
        peak_merge(v, iv, v2, i2);
        mm = fmaxf(mm, __shfl_down(mm, off));
        if (want_std) { s += __shfl_down(s, off); ss += __shfl_down(ss, off); }
    }
    __shared__ float sv[kThreads / 64]; __shared__ int si[kThreads / 64]; __shared__ float sm[kThreads / 64];
    __shared__ double ssum[kThreads / 64]; __shared__ double ssq[kThreads / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sv[wave] = v; si[wave] = iv; sm[wave] = mm; ssum[wave] = s; ssq[wave] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            peak_merge(v, iv, sv[w], si[w]);
            mm = fmaxf(mm, sm[w]);
            s += ssum[w]; ss += ssq[w];
        }
        PeakPartial p; p.vmax = v; p.imax = iv; p.mmax = mm; p.pad_ = 0; p.sum = s; p.sumsq = ss;
        part[(size_t)n * gridDim.x + chunk] = p;
    }
}

// numpy's round(): half to even
__device__ __forceinline__ double round_half_even(double x) { return rint(x); }

// stage 2: combine partials; sub-pixel fit; confidence (matcher.py:82-134)
__global__ void ncc_peak_final(const float* __restrict__ Csurf, const PeakPartial* __restrict__ part, int nchunks,
                               int Fh, int Fw, int H0, int W0, int H1, int W1, const int* __restrict__ blk, int subpixel, int conf_mode,
                               double* __restrict__ dx, double* __restrict__ dy, float* __restrict__ conf, int N,
                               const float* __restrict__ ct9 = nullptr) {
#pragma clang fp contract(off)
    // one wave per block pair: lanes stride over the partials, shuffle tree, lane 0 finishes
    const int n = blockIdx.x, lane = threadIdx.x;
    if (n >= N) return;
    const PeakPartial* p = part + (size_t)n * nchunks;
    float v = -INFINITY; int iv = 0x7fffffff; float mm = 0.f; double s = 0.0, ss = 0.0;
    for (int c = lane; c < nchunks; c += 64) {
        peak_merge(v, iv, p[c].vmax, p[c].imax);
        mm = fmaxf(mm, p[c].mmax);
        s += p[c].sum; ss += p[c].sumsq;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float v2 = __shfl_down(v, off);
        const int i2 = __shfl_down(iv, off);
        peak_merge(v, iv, v2, i2);
        mm = fmaxf(mm, __shfl_down(mm, off));
        s += __shfl_down(s, off); ss += __shfl_down(ss, off);
    }
    if (lane != 0) return;
    if (blk) {      // per-block sizes (crop mode)
        const int* d = blk + (size_t)n * kBlkStride;
        H0 = d[3]; W0 = d[4]; H1 = d[7]; W1 = d[8];
    }
    if (iv == 0x7fffffff) iv = 0;      // all -inf / NaN surface: numpy argmax -> 0
    const int py = iv / Fw, px = iv - py * Fw;
    double ddx = (double)px, ddy = (double)py;
    if (subpixel) {
        float ct[9];
        if (Csurf) {
            const float* c = Csurf + (size_t)n * Fh * Fw;
            for (int j = 0; j < 9; ++j) {
                int yy = py + (j / 3 - 1), xx = px + (j % 3 - 1);
                yy = (yy + Fh) % Fh; xx = (xx + Fw) % Fw;
                ct[j] = c[(size_t)yy * Fw + xx];
            }
        } else {
            for (int j = 0; j < 9; ++j) ct[j] = ct9[(size_t)n * 9 + j];
        }
        const float tx = (ct[5] - ct[3]) / 2.f;
        const float ty = (ct[7] - ct[1]) / 2.f;
        const float txx = ct[3] + ct[5] - 2.f * ct[4];
        const float tyy = ct[7] + ct[1] - 2.f * ct[4];
        const float txy = (ct[0] + ct[8] - ct[2] - ct[6]) / 4.f;
        const float det = txx * tyy - txy * txy;
        float ox = 0.f, oy = 0.f;
        if (det > 0.f) {
            const float ixx = tyy / det, ixy = -txy / det, iyy = txx / det;
            ox = -ixx * tx - ixy * ty;
            oy = -ixy * tx - iyy * ty;
        }
        ox = fminf(fmaxf(ox, -0.5f), 0.5f);
        oy = fminf(fmaxf(oy, -0.5f), 0.5f);
        ddx += (double)ox; ddy += (double)oy;
    }
    ddy += (double)(H0 - H1) / 2.0;
    ddx += (double)(W0 - W1) / 2.0;
    ddy -= round_half_even(ddy / (double)Fh) * (double)Fh;
    ddx -= round_half_even(ddx / (double)Fw) * (double)Fw;
    dx[n] = ddx; dy[n] = ddy;
    float cf = 1.f;
    if (conf_mode == FB_CONF_MIRROR) {
        cf = 0.f;
        if (v > 0.f) cf = 1.f - mm / v;
        cf = fminf(fmaxf(cf, 0.f), 1.f);
    } else if (conf_mode == FB_CONF_STD) {
        const double F = (double)Fh * (double)Fw;
        const double mean = s / F;
        double var = ss / F - mean * mean;
        if (var < 0) var = 0;
        // numpy evaluates (1 - exp(-Cmax/Cstd)) in float32 and only the power in float64
        // (matcher.py:130-133): the float32 quantisation of 1 - e^-r dominates the value.
        const float sd32 = (float)sqrt(var);
        const float e32 = expf(-(v / sd32));
        const float base32 = 1.0f - e32;
        double r = pow((double)base32, F);
        if (!(r >= 0.0)) r = (r != r) ? r : 0.0;
        if (r > 1.0) r = 1.0;
        cf = (float)r;
    }
    conf[n] = cf;
}

// ---- xcorr_fft(normalize=True), matcher.py:70-81, 119-122: the surfaces are divided by the overlap of the two masks at every lag
// before the peak is looked for.  part[c] = (largest value of chunk c of NC, of NCm); the surfaces are rocFFT's un-normalised ones
// (F = Fh Fw times the reference's).
__global__ void ncc_norm_partial_max(const float* __restrict__ NC, const float* __restrict__ NCm, float2* __restrict__ part, int F) {
    const int per = (F + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(F, lo + per);
    float a = -INFINITY, b = -INFINITY;
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) { a = fmaxf(a, NC[i]); b = fmaxf(b, NCm[i]); }
    for (int off = 32; off > 0; off >>= 1) { a = fmaxf(a, __shfl_down(a, off)); b = fmaxf(b, __shfl_down(b, off)); }
    __shared__ float sa[kThreads / 64], sb[kThreads / 64];
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) { a = fmaxf(a, sa[w]); b = fmaxf(b, sb[w]); }
        part[blockIdx.x] = make_float2(a, b);
    }
}

// D[0][i] = clip(NC[i] / max(max NC, 1), 0.1), D[1][i] likewise of NCm (both on the reference's scale: raw / F)
__global__ void ncc_norm_divisors(const float* __restrict__ NC, const float* __restrict__ NCm, const float2* __restrict__ part, int nchunks,
                                  int F, float* __restrict__ D) {
    float a = -INFINITY, b = -INFINITY;
    for (int c = 0; c < nchunks; ++c) { a = fmaxf(a, part[c].x); b = fmaxf(b, part[c].y); }
    const float inv = 1.0f / (float)F;
    const float sa = fmaxf(a * inv, 1.0f), sb = fmaxf(b * inv, 1.0f);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < F; i += gridDim.x * blockDim.x) {
        D[i] = fmaxf(NC[i] * inv / sa, 0.1f);
        D[(size_t)F + i] = fmaxf(NCm[i] * inv / sb, 0.1f);
    }
}

__global__ void ncc_norm_apply(float* __restrict__ Csurf, float* __restrict__ Msurf, const float* __restrict__ D, int F, int want_m) {
    const int n = blockIdx.y;
    float* c = Csurf + (size_t)n * F;
    float* m = want_m ? Msurf + (size_t)n * F : nullptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < F; i += gridDim.x * blockDim.x) {
        c[i] = c[i] / D[i];
        if (want_m) m[i] = m[i] / D[(size_t)F + i];
    }
}

int ensure_rocfft(fb_ctx* ctx) {
    if (ctx->rocfft_ready) return FB_OK;
    if (fb_rocfft_acquire() != 0) return fb_fail(ctx, FB_ERR_FFT, "rocfft_setup failed");
    ctx->rocfft_ready = true;               // from here on fb_destroy releases the reference
    FB_FFT(ctx, rocfft_execution_info_create(&ctx->fft_info));
    FB_FFT(ctx, rocfft_execution_info_set_stream(ctx->fft_info, ctx->stream));
    ctx->rocfft_ready = true;
    return FB_OK;
}

int get_plan(fb_ctx* ctx, int Fh, int Fw, int batch, fb_fft_plan** out) {
    auto key = std::make_tuple(Fh, Fw, batch);
    auto it = ctx->plans.find(key);
    if (it == ctx->plans.end()) {
        fb_fft_plan pl;
        const size_t lengths[2] = {(size_t)Fw, (size_t)Fh};
        FB_FFT(ctx, rocfft_plan_create(&pl.fwd, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                                       rocfft_precision_single, 2, lengths, (size_t)batch, nullptr));
        FB_FFT(ctx, rocfft_plan_create(&pl.inv, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                       rocfft_precision_single, 2, lengths, (size_t)batch, nullptr));
        size_t w0 = 0, w1 = 0;
        FB_FFT(ctx, rocfft_plan_get_work_buffer_size(pl.fwd, &w0));
        FB_FFT(ctx, rocfft_plan_get_work_buffer_size(pl.inv, &w1));
        pl.work_bytes = std::max(w0, w1);
        it = ctx->plans.emplace(key, pl).first;
    }
    fb_fft_plan& pl = it->second;
    if (pl.work_bytes > ctx->fft_work_bytes) {
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->fft_work) FB_HIP(ctx, hipFree(ctx->fft_work));
        ctx->fft_work = nullptr;
        ctx->fft_work_bytes = 0;
        FB_HIP(ctx, hipMalloc(&ctx->fft_work, pl.work_bytes));
        ctx->fft_work_bytes = pl.work_bytes;
    }
    if (ctx->fft_work_bytes)
        FB_FFT(ctx, rocfft_execution_info_set_work_buffer(ctx->fft_info, ctx->fft_work, ctx->fft_work_bytes));
    *out = &pl;
    return FB_OK;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// plan batch for a call of N items: the next power of two, capped by what fits the arena budget
int quantised_chunk(fb_ctx* ctx, int N, size_t per_pair) {
    size_t cap = std::max<size_t>(1, ctx->ncc_arena_limit / per_pair);
    size_t capq = 1;
    while (capq * 2 <= cap) capq *= 2;
    size_t q = 1;
    while (q < (size_t)N) q *= 2;
    // large transforms: one fixed plan batch per shape (rocFFT plan creation costs ~1 s each)
    if (per_pair >= ((size_t)4 << 20)) q = std::max<size_t>(q, 64);
    return (int)std::min(q, capq);
}

// one sub-batch of nb pairs through the rocFFT pipeline; all pointers are device pointers
struct CropSrc { const int* blk; int IH0, IW0, IH1, IW1; const double* aff = nullptr; };

int ncc_stream_subbatch(fb_ctx* ctx, const float* img0, const float* img1, int nb, int C, int H0, int W0, int H1,
                        int W1, int Fh, int Fw, int subpixel, int conf_mode, double* dx, double* dy, float* conf,
                        const CropSrc* crop = nullptr, int nbq = 0, const float* divisors = nullptr) {
    // nbq >= nb: number of slots the buffers and the rocFFT plans are built for (quantised so that
    // plans are reused across calls); slots [nb, nbq) hold stale data whose transforms are ignored.
    if (nbq < nb) nbq = nb;
    const int nreal = nb;
    const int Sw = Fw / 2 + 1;
    const size_t F = (size_t)Fh * Fw, S = (size_t)Fh * Sw;
    nb = nbq;
    const int NC = nb * C;
    const int want_q = conf_mode == FB_CONF_MIRROR;
    // arena layout
    size_t off = 0;
    const size_t oR = off; off += align_up(2 * (size_t)NC * F * sizeof(float), 256);
    const size_t oS = off; off += align_up(2 * (size_t)NC * S * sizeof(float2), 256);
    size_t oS2 = 0;
    if (C > 1) { oS2 = off; off += align_up(2 * (size_t)nb * S * sizeof(float2), 256); }
    const size_t oP = off; off += align_up((size_t)nb * kPeakChunks * sizeof(PeakPartial), 256);
    int rc = fb_arena_reserve(ctx, off);
    if (rc) return rc;
    char* base = (char*)ctx->arena;
    float* R = (float*)(base + oR);
    float2* Sp = (float2*)(base + oS);
    float2* S2 = C > 1 ? (float2*)(base + oS2) : nullptr;
    PeakPartial* part = (PeakPartial*)(base + oP);

    if (crop) {
        FB_PROF(ctx, "ncc_crop_load");
        dim3 grid((unsigned)std::min<size_t>((F + kThreads - 1) / kThreads, 4096), (unsigned)(nreal * C), 2);
        hipLaunchKernelGGL(ncc_crop_load, grid, dim3(kThreads), 0, ctx->stream, img0, img1, crop->blk, R, NC, crop->IH0, crop->IW0,
                           crop->IH1, crop->IW1, Fh, Fw);
    } else {
        FB_PROF(ctx, "ncc_pad_load");
        dim3 grid((unsigned)std::min<size_t>((F + kThreads - 1) / kThreads, 4096), (unsigned)(nreal * C), 2);
        hipLaunchKernelGGL(ncc_pad_load, grid, dim3(kThreads), 0, ctx->stream, img0, img1, R, NC, H0, W0, H1, W1, Fh, Fw);
    }
    fb_fft_plan* pl = nullptr;
    rc = get_plan(ctx, Fh, Fw, 2 * NC, &pl);
    if (rc) return rc;
    {
        FB_PROF(ctx, "rocfft_r2c");
        void* in[1] = {R};
        void* out[1] = {Sp};
        FB_FFT(ctx, rocfft_execute(pl->fwd, in, out, ctx->fft_info));
    }
    float2* F0 = Sp;
    float2* F1 = Sp + (size_t)NC * S;
    float2* P = C > 1 ? S2 : F0;
    float2* Q = C > 1 ? S2 + (size_t)nb * S : F1;
    {
        FB_PROF(ctx, "ncc_spectral_mul");
        const long long total = (long long)nreal * S;
        const int blocks = (int)std::min<long long>((total + kThreads - 1) / kThreads, 8192);
        hipLaunchKernelGGL(ncc_spectral_mul, dim3(blocks), dim3(kThreads), 0, ctx->stream, F0, F1, P, Q, (long long)S, nreal, C, want_q);
    }
    // inverse: P -> R[0..nb), Q -> R[nb..2nb)
    float* Csurf = R;
    float* Msurf = R + (size_t)nb * F;
    {
        FB_PROF(ctx, "rocfft_c2r");
        if (want_q && C == 1) {
            // P and Q are contiguous [2][nb] in Sp -> one batched inverse of 2*nb
            void* in[1] = {P};
            void* out[1] = {Csurf};
            FB_FFT(ctx, rocfft_execute(pl->inv, in, out, ctx->fft_info));
        } else {
            fb_fft_plan* pl1 = nullptr;
            const int ninv = want_q ? 2 * nb : nb;
            rc = get_plan(ctx, Fh, Fw, ninv, &pl1);
            if (rc) return rc;
            void* in[1] = {P};
            void* out[1] = {Csurf};
            FB_FFT(ctx, rocfft_execute(pl1->inv, in, out, ctx->fft_info));
        }
    }
    if (divisors) {                          // normalize=True: C / NC and C_mirror / NC_mirror ahead of everything that looks at them
        FB_PROF(ctx, "ncc_norm_apply");
        hipLaunchKernelGGL(ncc_norm_apply, dim3((unsigned)std::min<size_t>((F + kThreads - 1) / kThreads, 2048), nreal), dim3(kThreads), 0, ctx->stream,
                           Csurf, Msurf, divisors, (int)F, want_q);
    }
    {
        FB_PROF(ctx, "ncc_peak_partial");
        hipLaunchKernelGGL(ncc_peak_partial, dim3(kPeakChunks, nreal), dim3(kThreads), 0, ctx->stream, Csurf, Msurf, part,
                           (int)F, want_q, conf_mode == FB_CONF_STD);
    }
    {
        FB_PROF(ctx, "ncc_peak_final");
        hipLaunchKernelGGL(ncc_peak_final, dim3(nreal), dim3(64), 0, ctx->stream, Csurf, part, kPeakChunks, Fh, Fw,
                           H0, W0, H1, W1, crop ? crop->blk : (const int*)nullptr, subpixel, conf_mode, dx, dy, conf, nreal);
    }
    FB_HIP(ctx, hipGetLastError());
    ctx->last_Fh = Fh; ctx->last_Fw = Fw; ctx->last_N = nreal;
    ctx->last_C = Csurf; ctx->last_Cm = want_q ? Msurf : nullptr;
    return FB_OK;
}


// =============================================================================================
// Streaming class without rocFFT: three hand-written kernels per batch.
//   rows : packed R2C along x of the NON-ZERO rows only (both images ride one complex FFT),
//          spectra stored transposed  T[n][kx][y], y < Hs
//   cols : per (n, kx): zero-pad to Fh, forward FFT of both columns, conj/plain products,
//          inverse FFT, stored  V[n][kx][y], y < Fh
//   inv  : per tile of rows: Hermitian-extend + pack W = P + iQ, inverse FFT along x, reduce
//          max/arg-max of C, max |Cm| (the correlation surfaces are never written)
// HBM traffic per block pair = inputs + 16 S_h (T write) + 16 S_h (T read) + 16 S (V write)
// + 16 S (V read), S = Fh (Fw/2+1), S_h = Hs (Fw/2+1): the algorithmic minimum of SURVEY.md 8(d).

__device__ __forceinline__ void load_tw(float2* hi, float2* lo, const float2* ghi, const float2* glo) {
    for (int i = threadIdx.x; i < 64; i += blockDim.x) { hi[i] = ghi[i]; lo[i] = glo[i]; }
}

__global__ __launch_bounds__(kStreamThreads) void ncc_stream_rows(const StreamGeom g, float2* __restrict__ T0, float2* __restrict__ T1) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int Fw = g.Fw, Sw = g.Sw, TR = g.TR, pitch = fft_padx(Fw) + 1;
    float2* G = lds;
    float2* thi = G + (size_t)TR * pitch;
    float2* tlo = thi + 64;
    short* posW = reinterpret_cast<short*>(tlo + 64);
    const int n = blockIdx.y, y0 = blockIdx.x * TR;
    const int tid = threadIdx.x, nt = blockDim.x;
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
    load_tw(thi, tlo, g.twW_hi, g.twW_lo);
    for (int i = tid; i < Fw; i += nt) posW[i] = (short)fft_padx(fft_pos(g.pw, i));
    // packed load z = img0 + i img1 (zero padded), branch-free; all rows of a column chunk are fetched before
    // any LDS store so that the loads overlap
    __shared__ __attribute__((aligned(8))) float s_red[2 * (kStreamThreads / 64)];
    float m0 = 0.f, m1 = 0.f;
    for (int xb = 0; xb < Fw; xb += nt) {
        const int x = xb + tid;
        const int gx0 = ox0 + x, gx1 = ox1 + x;
        const bool vx0 = x < w0 && gx0 >= 0 && gx0 <= mx0, vx1 = x < w1 && gx1 >= 0 && gx1 <= mx1;
        const int cx0 = min(max(gx0, 0), mx0), cx1 = min(max(gx1, 0), mx1);
        float a[16], b[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r < TR) {
                const int y = y0 + r, gy0 = oy0 + y, gy1 = oy1 + y;
                a[r] = s0[(size_t)min(max(gy0, 0), my0) * p0 + cx0];
                if (g.aff) b[r] = fb_sample_affine(s1, g.IH1, g.IW1, g.aff + (size_t)n * FB_AFFINE_STRIDE, min(x, w1 - 1), min(y, h1 - 1));
                else b[r] = s1[(size_t)min(max(gy1, 0), my1) * p1 + cx1];
            }
        }
        if (x < Fw) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r < TR) {
                    const int y = y0 + r, gy0 = oy0 + y, gy1 = oy1 + y;
                    const bool v0 = vx0 && y < h0 && gy0 >= 0 && gy0 <= my0;
                    const bool v1 = g.aff ? (x < w1 && y < h1) : (vx1 && y < h1 && gy1 >= 0 && gy1 <= my1);
                    const float va = v0 ? a[r] : 0.f, vb = v1 ? b[r] : 0.f;
                    m0 = fmaxf(m0, fabsf(va)); m1 = fmaxf(m1, fabsf(vb));
                    G[r * pitch + fft_padx(x)] = make_float2(va, vb);
                }
            }
        }
    }
    wg_max2_post(m0, m1, s_red);
    __syncthreads();
    const float2 mm = wg_max2_read(s_red);
    const float2 sc = pack_scales(mm.x, mm.y);                // (fb_ldsfft.h) != 1 only for a tile one side of which is almost blank
    const float mmx = mm.x, mmy = mm.y;
    if (sc.x != 1.f || sc.y != 1.f) {
        for (int i = tid; i < TR * Fw; i += nt) { float2* z = G + (i / Fw) * pitch + fft_padx(i % Fw); *z = make_float2(z->x * sc.x, z->y * sc.y); }
        __syncthreads();
    }
    fft_batch_tw<false, TwSplit, 16, true>(G, g.pw, TR, 1, pitch, TwSplit{thi, tlo}, false);
    // split the packed spectra and store transposed as interleaved column pairs T[n][kx/2][y][kx&1]:
    // a (pair, tile) is 2*TR consecutive float2 = 128 B at TR = 8, lanes walk it contiguously
    const size_t tbase = (size_t)n * g.Kp * g.Hs * 2;
    const float ia = mmx > 0.f ? 0.5f / sc.x : 0.f, ib = mmy > 0.f ? 0.5f / sc.y : 0.f;
              // powers of two: exact; an image that is exactly zero on the tile gets an exactly zero spectrum (the split leaves rounding noise of the other image there)
    for (int t = tid; t < 2 * g.Kp * TR; t += nt) {
        const int c = t & 1, r = (t >> 1) & (TR - 1), kp = t / (2 * TR);
        const int kx = 2 * kp + c;
        float2 a = make_float2(0.f, 0.f), b = make_float2(0.f, 0.f);
        if (kx < Sw) {
            const float2 zk = G[r * pitch + posW[kx]];
            const float2 zn = G[r * pitch + posW[kx == 0 ? 0 : Fw - kx]];
            a = make_float2(ia * (zk.x + zn.x), ia * (zk.y - zn.y));
            b = make_float2(ib * (zk.y + zn.y), -ib * (zk.x - zn.x));
        }
        const size_t o = tbase + ((size_t)kp * g.Hs + y0 + r) * 2 + c;
        T0[o] = a;
        T1[o] = b;
    }
}

// one workgroup = one PAIR of adjacent kx columns of one block pair: four length-Fh transforms
// (image 0/1 x column 0/1) forward, products, inverse.  Loads and stores are fully contiguous.
__global__ __launch_bounds__(kStreamThreads) void ncc_stream_cols(const StreamGeom g, const float2* __restrict__ T0, const float2* __restrict__ T1,
                                                                  float2* __restrict__ V0, float2* __restrict__ V1) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int Fh = g.Fh, pitch = fft_padx(Fh) + 1;
    float2* U = lds;                      // [4][pitch]: (img0,c0) (img0,c1) (img1,c0) (img1,c1)
    float2* thi = U + 4 * (size_t)pitch;
    float2* tlo = thi + 64;
    const int kp = blockIdx.x, n = blockIdx.y;
    const int tid = threadIdx.x, nt = blockDim.x;
    load_tw(thi, tlo, g.twH_hi, g.twH_lo);
    const size_t tb = ((size_t)n * g.Kp + kp) * g.Hs * 2;
    for (int eb = 0; eb < 2 * Fh; eb += 8 * nt) {
        float2 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = eb + u * nt + tid;            // e = 2 y + c
            const bool in = e < 2 * g.Hs;
            a[u] = in ? T0[tb + e] : make_float2(0.f, 0.f);
            b[u] = in ? T1[tb + e] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = eb + u * nt + tid;
            if (e < 2 * Fh) {
                const int c = e & 1, yp = fft_padx(e >> 1);
                U[c * pitch + yp] = a[u];
                U[(2 + c) * pitch + yp] = b[u];
            }
        }
    }
    __syncthreads();
    fft_batch_tw<false, TwSplit, 16, true>(U, g.ph, 4, 1, pitch, TwSplit{thi, tlo}, false);
    for (int e = tid; e < 2 * Fh; e += nt) {
        const int c = e & 1, yp = fft_padx(e >> 1);
        const float2 a = U[c * pitch + yp], b = U[(2 + c) * pitch + yp];
        U[c * pitch + yp] = make_float2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
        U[(2 + c) * pitch + yp] = g.want_q ? make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    fft_batch_tw<true, TwSplit, 16, true>(U, g.ph, g.want_q ? 4 : 2, 1, pitch, TwSplit{thi, tlo}, false);
    const size_t vb = ((size_t)n * g.Kp + kp) * Fh * 2;
    for (int e = tid; e < 2 * Fh; e += nt) {
        const int c = e & 1, yp = fft_padx(e >> 1);
        V0[vb + e] = U[c * pitch + yp];
        if (g.want_q) V1[vb + e] = U[(2 + c) * pitch + yp];
    }
}

// rows_sel == nullptr: tile of TRI consecutive rows, reduce into part[n][tile].
// rows_sel != nullptr (sub-pixel neighbours): 3 rows around the peak of block n, outputs ct9.
__global__ __launch_bounds__(kStreamThreads) void ncc_stream_inv(const StreamGeom g, const float2* __restrict__ V0, const float2* __restrict__ V1,
                                                                 PeakPartial* __restrict__ part, const PeakPartial* __restrict__ part_in,
                                                                 int nparts, float* __restrict__ ct9) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int Fh = g.Fh, Fw = g.Fw, Sw = g.Sw, pitch = fft_padx(Fw) + 1;
    const bool neigh = ct9 != nullptr;
    const int TRI = neigh ? 4 : g.TRI;
    float2* G = lds;
    float2* thi = G + (size_t)TRI * pitch;
    float2* tlo = thi + 64;
    short* posW = reinterpret_cast<short*>(tlo + 64);
    __shared__ float sv[kStreamThreads / 64]; __shared__ int si[kStreamThreads / 64]; __shared__ float sm[kStreamThreads / 64];
    __shared__ double ssum[kStreamThreads / 64]; __shared__ double ssq[kStreamThreads / 64];
    __shared__ int s_peak;
    const int n = blockIdx.y;
    const int tid = threadIdx.x, nt = blockDim.x;
    load_tw(thi, tlo, g.twW_hi, g.twW_lo);
    for (int i = tid; i < Fw; i += nt) posW[i] = (short)fft_padx(fft_pos(g.pw, i));
    int y0 = blockIdx.x * TRI;
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
        py = s_peak / Fw; px = s_peak - py * Fw;
    }
    __syncthreads();
    const int nmir = Fw - Sw;
    const size_t vb = (size_t)n * g.Kp * Fh * 2;
    const int nitems = 2 * g.Kp * TRI;
    for (int tb = 0; tb < nitems; tb += 8 * nt) {
        float2 pk[8], qk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = min(tb + u * nt + tid, nitems - 1);
            const int c = t & 1, r = (t >> 1) & (TRI - 1), kp = t / (2 * TRI);
            int y = y0 + r;
            bool ok = y < Fh;
            if (neigh) { y = (py + r - 1 + Fh) % Fh; ok = r < 3; }
            y = min(y, Fh - 1);
            const size_t o = vb + ((size_t)kp * Fh + y) * 2 + c;
            pk[u] = V0[o];
            qk[u] = g.want_q ? V1[o] : make_float2(0.f, 0.f);
            if (!ok) { pk[u] = make_float2(0.f, 0.f); qk[u] = make_float2(0.f, 0.f); }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = tb + u * nt + tid;
            const int c = t & 1, r = (t >> 1) & (TRI - 1), kp = t / (2 * TRI);
            const int kx = 2 * kp + c;
            if (t < nitems && kx < Sw) {
                const bool self = (kx == 0) || (2 * kx == Fw);
                G[r * pitch + posW[kx]] = self ? make_float2(pk[u].x, qk[u].x) : make_float2(pk[u].x - qk[u].y, pk[u].y + qk[u].x);
                if (kx >= 1 && kx <= nmir) G[r * pitch + posW[Fw - kx]] = make_float2(pk[u].x + qk[u].y, qk[u].x - pk[u].y);
            }
        }
    }
    __syncthreads();
    fft_batch_tw<true, TwSplit, 16, true>(G, g.pw, TRI, 1, pitch, TwSplit{thi, tlo}, false);
    if (neigh) {
        if (tid < 9) ct9[(size_t)n * 9 + tid] = G[(tid / 3) * pitch + fft_padx((px + (tid % 3 - 1) + Fw) % Fw)].x;
        return;
    }
    float v = -INFINITY; int iv = 0x7fffffff; float mm = 0.f;
    double s = 0.0, ss = 0.0;
    for (int r = 0; r < TRI; ++r) {
        const int y = y0 + r;
        if (y >= Fh) break;
        for (int x = tid; x < Fw; x += nt) {
            const float2 c = G[r * pitch + fft_padx(x)];
            if (c.x > v) { v = c.x; iv = y * Fw + x; }
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
        for (int w = 1; w < (int)(nt >> 6); ++w) {
            peak_merge(v, iv, sv[w], si[w]);
            mm = fmaxf(mm, sm[w]);
            s += ssum[w]; ss += ssq[w];
        }
        PeakPartial p; p.vmax = v; p.imax = iv; p.mmax = mm; p.pad_ = 0; p.sum = s; p.sumsq = ss;
        part[(size_t)n * gridDim.x + blockIdx.x] = p;
    }
}


#include "fb_ncc_p2.inc"

std::map<std::pair<int, int>, std::pair<float2*, float2*>> g_split_tables;

int get_split_table(fb_ctx* ctx, int n, const float2** hi, const float2** lo) {
    static std::mutex mtx;                      // process-wide table shared by every context of the device
    std::lock_guard<std::mutex> lk(mtx);
    auto key = std::make_pair(ctx->device, n);
    auto it = g_split_tables.find(key);
    if (it == g_split_tables.end()) {
        std::vector<float2> h(128);
        for (int k = 0; k < 64; ++k) {
            const double a = -2.0 * M_PI * (double)(64 * k) / (double)n, b = -2.0 * M_PI * (double)k / (double)n;
            h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
            h[64 + k] = make_float2((float)std::cos(b), (float)std::sin(b));
        }
        float2* d = nullptr;
        FB_HIP(ctx, hipMalloc((void**)&d, sizeof(float2) * 128));
        FB_HIP(ctx, hipMemcpy(d, h.data(), sizeof(float2) * 128, hipMemcpyHostToDevice));
        it = g_split_tables.emplace(key, std::make_pair(d, d + 64)).first;
    }
    *hi = it->second.first; *lo = it->second.second;
    return FB_OK;
}

// tw[i] = exp(-2 pi i / n), i < n / 16 (the twiddle bases of the power-of-two core, fb_fft2.h)
std::map<std::pair<int, int>, float2*> g_tw16_tables;
int get_tw16_table(fb_ctx* ctx, int n, const float2** tw) {
    static std::mutex mtx;                      // process-wide table shared by every context of the device
    std::lock_guard<std::mutex> lk(mtx);
    auto key = std::make_pair(ctx->device, n);
    auto it = g_tw16_tables.find(key);
    if (it == g_tw16_tables.end()) {
        const int cnt = std::max(1, n / 16);
        std::vector<float2> h((size_t)cnt);
        for (int k = 0; k < cnt; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)n;
            h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        float2* d = nullptr;
        FB_HIP(ctx, hipMalloc((void**)&d, sizeof(float2) * cnt));
        FB_HIP(ctx, hipMemcpy(d, h.data(), sizeof(float2) * cnt, hipMemcpyHostToDevice));
        it = g_tw16_tables.emplace(key, d).first;
    }
    *tw = it->second;
    return FB_OK;
}

// twS[y] = exp(-2 pi i y / n), y < n / 2 (split columns of length n: the input twiddle of the odd branch, the butterfly of the inverse)
std::map<std::pair<int, int>, float2*> g_tw_split_tables;
int get_tw_split_table(fb_ctx* ctx, int n, const float2** tw) {
    static std::mutex mtx;
    std::lock_guard<std::mutex> lk(mtx);
    auto key = std::make_pair(ctx->device, n);
    auto it = g_tw_split_tables.find(key);
    if (it == g_tw_split_tables.end()) {
        const int cnt = n / 2;
        std::vector<float2> h((size_t)cnt);
        for (int k = 0; k < cnt; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)n;
            h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        float2* d = nullptr;
        FB_HIP(ctx, hipMalloc((void**)&d, sizeof(float2) * cnt));
        FB_HIP(ctx, hipMemcpy(d, h.data(), sizeof(float2) * cnt, hipMemcpyHostToDevice));
        it = g_tw_split_tables.emplace(key, d).first;
    }
    *tw = it->second;
    return FB_OK;
}

int pow2_floor(int v) { int p = 1; while (p * 2 <= v) p *= 2; return p; }

// the 8192-point shapes of the power-of-two core: rows of 8192 points (one per tile), columns of 8192 points whose upper half is
// zero padding as two half-length items (fb_ncc_p2.inc: split columns); the other axis a power of two of the core
bool p2_long_shape(int Fh, int Fw, int hmax) {
    if (getenv("FEABAS_HIP_FFT_GENERIC") || getenv("FEABAS_HIP_NO_LONG")) return false;
    if (!(Fh == kSplitLen || Fw == kSplitLen) || Fw < 256) return false;
    if (!(p2_shape(Fw) || Fw == kSplitLen)) return false;
    if (p2_shape(Fh)) return true;
    const int tr = p2_tr(Fw);                                  // the row pass stores whole tiles of rows
    return Fh == kSplitLen && 2 * ((std::max(1, hmax) + tr - 1) / tr * tr) <= Fh;
}

// A shape with an axis of 4097 .. 8192 points (the reference's next_fast_len of two whole 4096-pixel tiles padded is 8192 itself; of
// 4000-pixel tiles, 8000) at the 8192-point forms: linear (zero-padded) axes go up to the next power of two, like
// promote_linear_shape below and for the same reason; a circular axis, and every axis under FFT_CONF_STD, must be one as it stands.
bool promote_long_shape(int& Fh, int& Fw, int need_h, int need_w, int hmax, int conf_mode) {
    if ((Fh <= 4096 && Fw <= 4096) || Fh > kSplitLen || Fw > kSplitLen) return false;
    const bool exact = conf_mode == FB_CONF_STD || getenv("FEABAS_HIP_FFT_EXACT");
    auto up = [&](int F, int need) {
        if (F < need || exact) return F;
        int P = 64;
        while (P < F) P <<= 1;
        return P;
    };
    const int ph = up(Fh, need_h), pw = up(Fw, need_w);
    if (!p2_long_shape(ph, pw, hmax)) return false;
    Fh = ph; Fw = pw;
    return true;
}

bool stream_custom_supported(int Fh, int Fw, int C) {
    if (C != 1 || Fw > 4096 || Fh > 4096 || Fw < 4 || Fh < 2) return false;
    if ((4 * (size_t)(Fh + Fh / 16 + 1) + 128) * sizeof(float2) > 150 * 1024) return false;   // four columns must fit the LDS   // 2-level twiddles cover N <= 4096; posW is int16
    FftPlan p;
    return fft_make_plan(Fh, &p) && fft_make_plan(Fw, &p);
}

// sub-batch of nb block pairs through the three custom kernels
int ncc_custom_subbatch(fb_ctx* ctx, const float* img0, const float* img1, int nb, int H0, int W0, int H1, int W1, int hmax, int wmax, int Fh,
                        int Fw, int subpixel, int conf_mode, double* dx, double* dy, float* conf, const CropSrc* crop) {
    StreamGeom g;
    g.N = nb; g.Fh = Fh; g.Fw = Fw; g.Sw = Fw / 2 + 1; g.Kp = (g.Sw + 1) / 2;
    const char* trb = getenv("FB_TRB");
    const size_t lds_budget = (trb ? (size_t)atoi(trb) : 70) * 1024;
    g.TR = std::min(16, pow2_floor((int)std::max<size_t>(1, lds_budget / ((size_t)(Fw + Fw / 16 + 1) * sizeof(float2)))));
    const bool p2_long = p2_long_shape(Fh, Fw, hmax);                       // an axis of 8192 points
    const bool split = p2_long && Fh == kSplitLen;                          // ... the columns: two half-length items per column pair
    const bool p2 = p2_long || (p2_shape(Fh) && p2_shape(Fw) && !getenv("FEABAS_HIP_FFT_GENERIC"));
    const bool ct = !p2 && fb_ncc_ct_len(Fh) && fb_ncc_ct_len(Fw) && !getenv("FEABAS_HIP_FFT_GENERIC");     // compile-time mixed-radix plans
    if (p2) g.TR = p2_tr(Fw);
    if (ct) g.TR = fb_ncc_ct_tr(Fw);
    g.TRI = split ? p2_trs(Fw) : g.TR;
    // inverse pass on half tiles (workgroups of 256 threads, twice as many per CU: half as many waves meet at each barrier):
    // 11-13 % faster at FW <= 1024; narrower tiles would cut the contiguous runs of V below 64 B (FB_INV_HALF=0/1 overrides)
    const char* inv_half = getenv("FB_INV_HALF");
    if (p2 && !split && (inv_half ? atoi(inv_half) != 0 : g.TR >= 8)) g.TRI = std::max(1, g.TR / 2);
    if (ct) {
        // inverse pass of the mixed-radix class: tiles of 4 rows where the row pass has 8 (rows of 513 .. 1024 points: 21 KB of LDS
        // instead of 43, six workgroups per CU instead of three): 3.63 -> 3.37 us per 280 x 280 block pair at FFT 576 x 576; tiles of 2
        // rows (32-byte runs of V) and halving the 16-row tiles of shorter rows are slower (FB_CT_TRI = 2 / 4 / 8 / 16 overrides)
        static const int ct_tri = [] { const char* e = getenv("FB_CT_TRI"); return e ? atoi(e) : 0; }();
        if (ct_tri == 2 || ct_tri == 4 || ct_tri == 8 || ct_tri == 16) g.TRI = ct_tri;
        else if (g.TR == 8) g.TRI = 4;
    }
    const int rows = std::min(Fh, std::max(1, hmax));
    g.Hs = (rows + g.TR - 1) / g.TR * g.TR;
    g.H0 = H0; g.W0 = W0; g.H1 = H1; g.W1 = W1;
    int rc = FB_OK;
    if (!p2_long) {                         // (plans and two-level twiddles of the generic kernels: lengths up to 4096)
        if (!fft_make_plan(Fw, &g.pw) || !fft_make_plan(Fh, &g.ph)) return fb_fail(ctx, FB_ERR_ARG, "ncc: FFT shape %dx%d is not 5-smooth", Fh, Fw);
        rc = get_split_table(ctx, Fw, &g.twW_hi, &g.twW_lo);
        if (rc) return rc;
        rc = get_split_table(ctx, Fh, &g.twH_hi, &g.twH_lo);
        if (rc) return rc;
    } else {
        g.twW_hi = g.twW_lo = g.twH_hi = g.twH_lo = nullptr;
        if (split && 2 * g.Hs > Fh) return fb_fail(ctx, FB_ERR_ARG, "ncc: %d rows in columns of %d points", g.Hs, Fh);
    }
    g.img0 = img0; g.img1 = img1;
    g.blk = crop ? crop->blk : nullptr;
    g.aff = crop ? crop->aff : nullptr;
    g.IH0 = crop ? crop->IH0 : 0; g.IW0 = crop ? crop->IW0 : 0; g.IH1 = crop ? crop->IH1 : 0; g.IW1 = crop ? crop->IW1 : 0;
    g.want_q = conf_mode == FB_CONF_MIRROR;
    g.want_std = conf_mode == FB_CONF_STD;
    const size_t nT = (size_t)nb * g.Kp * 2 * g.Hs, nV = (size_t)nb * g.Kp * 2 * Fh;
    const int ntiles = (Fh + g.TRI - 1) / g.TRI;
    size_t off = 0;
    const size_t oT0 = off; off += align_up(nT * sizeof(float2), 256);
    const size_t oT1 = off; off += align_up(nT * sizeof(float2), 256);
    const size_t oV0 = off; off += align_up(nV * sizeof(float2), 256);
    const size_t oV1 = off; off += align_up(nV * sizeof(float2), 256);
    const size_t oP = off; off += align_up((size_t)nb * ntiles * sizeof(PeakPartial), 256);
    const size_t oC = off; off += align_up((size_t)nb * 9 * sizeof(float), 256);
    rc = fb_arena_reserve(ctx, off);
    if (rc) return rc;
    char* base = (char*)ctx->arena;
    float2 *T0 = (float2*)(base + oT0), *T1 = (float2*)(base + oT1), *V0 = (float2*)(base + oV0), *V1 = (float2*)(base + oV1);
    PeakPartial* part = (PeakPartial*)(base + oP);
    float* ct9 = (float*)(base + oC);
    const size_t lds_rows = ((size_t)g.TR * (Fw + Fw / 16 + 1) + 128) * sizeof(float2) + (size_t)(Fw + 2) / 2 * 2 * sizeof(short);
    const size_t lds_cols = (4 * (size_t)(Fh + Fh / 16 + 1) + 128) * sizeof(float2);
    const size_t lds_inv = ((size_t)g.TRI * (Fw + Fw / 16 + 1) + 128) * sizeof(float2) + (size_t)(Fw + 2) / 2 * 2 * sizeof(short);
    if (!p2 && !ct) {
        FB_HIP(ctx, hipFuncSetAttribute((const void*)ncc_stream_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows));
        FB_HIP(ctx, hipFuncSetAttribute((const void*)ncc_stream_cols, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cols));
        FB_HIP(ctx, hipFuncSetAttribute((const void*)ncc_stream_inv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(lds_inv, ((size_t)4 * (Fw + Fw / 16 + 1) + 128) * sizeof(float2) + (size_t)(Fw + 2) * sizeof(short))));
    }
    const double in_bytes = crop ? 8.0 * hmax * wmax : 4.0 * ((double)H0 * W0 + (double)H1 * W1);
    const double nq = g.want_q ? 2.0 : 1.0;
    if (ct) {
        rc = fb_ncc_ct_run(ctx, g, nb, T0, T1, V0, V1, part, ntiles, ct9, subpixel, in_bytes);
        if (rc) return rc;
        FB_PROF(ctx, "ncc_peak_final");
        hipLaunchKernelGGL(ncc_peak_final, dim3(nb), dim3(64), 0, ctx->stream, (const float*)nullptr, part, ntiles, Fh, Fw, H0, W0, H1,
                           W1, crop ? crop->blk : (const int*)nullptr, subpixel, conf_mode, dx, dy, conf, nb, subpixel ? ct9 : (const float*)nullptr);
        FB_HIP(ctx, hipGetLastError());
        ctx->last_C = nullptr; ctx->last_Cm = nullptr;
        return FB_OK;
    }
    P2Geom q;
    size_t lds_rows2 = 0, lds_cols2 = 0, lds_inv2 = 0, lds_n2 = 0;
    // grid: one workgroup per item by default (measured fastest: the hardware dispatcher overlaps one workgroup's loads
    // with its CU neighbour's FFT); FB_P2_SLOTS=k runs k persistent workgroups per CU with register prefetch instead
    const int slots_per_cu = getenv("FB_P2_SLOTS") ? atoi(getenv("FB_P2_SLOTS")) : 0;
    const int wg_slots = slots_per_cu > 0 ? slots_per_cu * ctx->prop.multiProcessorCount : (1 << 30);
    // FEABAS_HIP_P2_XCD: bit 0 rows, bit 1 columns, bit 2 inverse rows -- the items of the pass in one contiguous eighth per XCD
    // (fb_ncc_p2.inc: p2_item); default 7.  Only for the one-workgroup-per-item launches.
    static const int xcd_mask = [] { const char* e = getenv("FEABAS_HIP_P2_XCD"); return e ? atoi(e) : 7; }();
    q.per8 = 0;
    // one persistent workgroup per CU that walks its XCD's eighth of the items (fb_ncc_p2.inc: p2_step / p2_valid)
    auto xcd_persist = [&](int total) {
        const int k = std::max(1, ctx->prop.multiProcessorCount / 8);
        q.per8 = (total + 7) / 8;
        return 8 * std::min(q.per8, k);
    };
    const int persist = !p2_long ? 0 : (getenv("FB_P2_LONG_PERSIST") ? atoi(getenv("FB_P2_LONG_PERSIST")) : 1);    // bit 0: inverse rows, bit 1: rows and columns
    auto xcd_grid = [&](int bit, int total, int grid) {       // grid of the launch; sets q.per8
        q.per8 = 0;
        if (!(xcd_mask & bit) || grid != total || total < 64) return grid;
        q.per8 = (total + 7) / 8;
        return 8 * q.per8;
    };
    if (p2) {
        rc = get_tw16_table(ctx, Fw, &q.twW);
        if (rc) return rc;
        const int Fc = split ? Fh / 2 : Fh;                 // length of the column transforms
        rc = get_tw16_table(ctx, Fc, &q.twH);
        if (rc) return rc;
        q.twS = nullptr;
        if (split) {
            rc = get_tw_split_table(ctx, Fh, &q.twS);
            if (rc) return rc;
        }
        const size_t pw = (size_t)(Fw + Fw / 16 + 1), ph = (size_t)(Fc + Fc / 16 + 1);
        lds_rows2 = ((size_t)g.TR * pw + Fw / 16) * sizeof(float2);
        lds_cols2 = (4 * (size_t)p2_np(Fc) * ph + Fc / 16) * sizeof(float2);
        lds_inv2 = ((size_t)g.TRI * pw + Fw / 16) * sizeof(float2);
        lds_n2 = ((size_t)4 * pw + Fw / 16) * sizeof(float2);
        q.lTRI = 0;
        while ((1 << q.lTRI) < g.TRI) ++q.lTRI;
        q.ntiles_inv = ntiles;
    }
    {
        FB_PROF_B(ctx, "ncc_stream_rows", nb * (in_bytes + 16.0 * g.Sw * g.Hs));
        if (p2) {
            q.tiles = g.Hs / g.TR; q.total = q.tiles * nb;
            // (rows and columns of the 8192-point shapes as one persistent workgroup per CU with register prefetch -- their tiles fill
            // the LDS, no neighbour shares the CU -- were measured at HALF the rate of one workgroup per item: the persistent walk
            // gives up the XCD-contiguous order of the stores.  FB_P2_LONG_PERSIST=3 to repeat; the inverse pass, which only reads, gains 15 %)
            { const int grid = ((persist & 2) && lds_rows2 > 80 * 1024) ? xcd_persist(q.total) : xcd_grid(1, q.total, std::min(q.total, wg_slots));
              FB_P2_SWITCH_W(Fw, rows(ctx->stream, grid, lds_rows2, g, q, T0, T1)); }
        } else hipLaunchKernelGGL(ncc_stream_rows, dim3(g.Hs / g.TR, nb), dim3(kStreamThreads), lds_rows, ctx->stream, g, T0, T1);
    }
    {
        FB_PROF_B(ctx, "ncc_stream_cols", (double)nb * g.Sw * (16.0 * g.Hs + 8.0 * nq * Fh));
        if (split) {
            q.tiles = 2 * g.Kp; q.total = q.tiles * nb;                                  // (column pair, even / odd frequencies)
            const int grid = (persist & 2) ? xcd_persist(q.total) : xcd_grid(2, q.total, q.total);
            p2_cols_split(ctx->stream, grid, lds_cols2, g, q, T0, T1, V0, V1);
        } else if (p2) {
            q.tiles = (g.Kp + p2_np(Fh) - 1) / p2_np(Fh); q.total = q.tiles * nb;       // groups of column pairs
            // zero-padded columns (every padded correlation) take the direct form: first pass from HBM, last pass to HBM
            const bool direct = 2 * g.Hs <= Fh && Fh >= 256 && slots_per_cu == 0 && !(getenv("FB_COLS_STAGED") && atoi(getenv("FB_COLS_STAGED")));
            if (direct) { const int grid = xcd_grid(2, q.total, q.total); FB_P2_SWITCH(Fh, cols2(ctx->stream, grid, lds_cols2, g, q, T0, T1, V0, V1)); }
            else { const int grid = xcd_grid(2, q.total, std::min(q.total, wg_slots)); FB_P2_SWITCH(Fh, cols(ctx->stream, grid, lds_cols2, g, q, T0, T1, V0, V1)); }
        } else hipLaunchKernelGGL(ncc_stream_cols, dim3(g.Kp, nb), dim3(kStreamThreads), lds_cols, ctx->stream, g, T0, T1, V0, V1);
    }
    {
        FB_PROF_B(ctx, "ncc_stream_inv", (double)nb * g.Sw * 8.0 * nq * Fh);
        if (p2) {
            q.tiles = ntiles; q.total = ntiles * nb;
            if (split) {
                // (a tile of rows of 8192 points fills the LDS: one persistent workgroup per CU with register prefetch)
                const bool fills = (size_t)p2_trs(Fw) * (Fw + Fw / 16 + 1) * sizeof(float2) > 80 * 1024;
                const int grid = (fills && (persist & 1)) ? xcd_persist(q.total) : xcd_grid(4, q.total, q.total);
                FB_P2_SWITCH_W(Fw, inv_split(ctx->stream, grid, g, q, V0, V1, part));
            }
            else {
                // (FB_P2_INV_PERSIST=1: a tile that fills the LDS -- rows of 2048 and 4096 points -- on one persistent workgroup per CU)
                static const int inv_persist = [] { const char* e = getenv("FB_P2_INV_PERSIST"); return e ? atoi(e) : 0; }();
                const int grid = (inv_persist && lds_inv2 > 80 * 1024 && slots_per_cu == 0) ? xcd_persist(q.total) : xcd_grid(4, q.total, std::min(q.total, wg_slots));
                FB_P2_SWITCH_W(Fw, inv(ctx->stream, grid, lds_inv2, g, q, V0, V1, part));
            }
        } else hipLaunchKernelGGL(ncc_stream_inv, dim3(ntiles, nb), dim3(kStreamThreads), lds_inv, ctx->stream, g, V0, V1, part, nullptr, 0, nullptr);
    }
    if (subpixel) {
        FB_PROF(ctx, "ncc_stream_neighbors");
        const size_t lds_n = ((size_t)4 * (Fw + Fw / 16 + 1) + 128) * sizeof(float2) + (size_t)(Fw + 2) / 2 * 2 * sizeof(short);
        if (split) { q.per8 = 0; FB_P2_SWITCH_W(Fw, neigh1(ctx->stream, nb, g, q, V0, V1, part, ntiles, ct9, true)); }
        else if (p2) { q.per8 = 0; FB_P2_SWITCH_W(Fw, neigh(ctx->stream, nb, lds_n2, g, q, V0, V1, part, ntiles, ct9)); }
        else hipLaunchKernelGGL(ncc_stream_inv, dim3(1, nb), dim3(kStreamThreads), lds_n, ctx->stream, g, V0, V1, nullptr, part, ntiles, ct9);
    }
    {
        FB_PROF(ctx, "ncc_peak_final");
        hipLaunchKernelGGL(ncc_peak_final, dim3(nb), dim3(64), 0, ctx->stream, (const float*)nullptr, part, ntiles, Fh, Fw, H0, W0, H1,
                           W1, crop ? crop->blk : (const int*)nullptr, subpixel, conf_mode, dx, dy, conf, nb, subpixel ? ct9 : (const float*)nullptr);
    }
    FB_HIP(ctx, hipGetLastError());
    ctx->last_C = nullptr; ctx->last_Cm = nullptr;
    return FB_OK;
}

// A zero-padded (linear) correlation does not depend on the padded length: every lag keeps its value, and the
// peak / sub-pixel / MIRROR-confidence arithmetic only ever looks at lags.  When the reference's 5-smooth
// next_fast_len(s0 + s1 - 1) sits just below a power of two (1000, 972, 960 -> 1024), the power of two runs on
// the packed compile-time FFT core instead of the generic mixed-radix one.  Never applied to circular
// (pad=False) axes or to FFT_CONF_STD, whose result depends on the surface size.
void promote_linear_shape(int& Fh, int& Fw, int need_h, int need_w, int conf_mode) {
    if (conf_mode == FB_CONF_STD || getenv("FEABAS_HIP_FFT_EXACT")) return;
    auto up = [](int F, int need) {
        if (F < need) return F;
        int P = 64;
        while (P < F) P <<= 1;
        return (P <= 4096 && 5 * P <= 6 * F) ? P : F;
    };
    const int ph = up(Fh, need_h), pw = up(Fw, need_w);
    if (p2_shape(ph) && p2_shape(pw)) { Fh = ph; Fw = pw; return; }
    // otherwise a length with a compile-time mixed-radix plan (fb_ncc_ct.hip: 2^a, 3 2^a, 5 2^a, 9 2^a) within 1.2 x, axis by
    // axis; a circular axis must have such a length as it stands
    auto ct = [](int F, int need) { return F < need ? (fb_ncc_ct_len(F) ? F : 0) : fb_ncc_ct_up(need, F); };
    const int ch = ct(Fh, need_h), cw = ct(Fw, need_w);
    if (ch && cw) { Fh = ch; Fw = cw; }
}

size_t custom_bytes_per_pair(int Fh, int Fw, int hmax) {
    const size_t Sw = (size_t)(Fw / 2 + 2) / 2 * 2;      // column pairs
    return 2 * Sw * (size_t)(std::min(Fh, hmax) + 16) * 8 + 2 * Sw * (size_t)Fh * 8 + 4096;
}

}  // namespace

extern "C" {

int fb_next_fast_len(int n) {
    if (n <= 6) return n < 0 ? 0 : n;
    int best = 1;
    while (best < n) best <<= 1;
    for (long long p5 = 1; p5 < best; p5 *= 5) {
        for (long long p35 = p5; p35 < best; p35 *= 3) {
            long long c = p35;
            while (c < n) c <<= 1;
            if (c < best) best = (int)c;
        }
    }
    return best;
}

int fb_ncc_batch_dev(fb_ctx* ctx, const float* img0, const float* img1, int N, int C, int H0, int W0, int H1, int W1,
                     int pad, int subpixel, int conf_mode, double* dx, double* dy, float* conf) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && C >= 1 && H0 > 0 && W0 > 0 && H1 > 0 && W1 > 0);
    FB_CHECK_ARG(ctx, conf_mode >= 0 && conf_mode <= 2);
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, img0 && img1 && dx && dy && conf);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    const int Fh_ref = pad ? fb_next_fast_len(H0 + H1 - 1) : fb_next_fast_len(std::max(H0, H1));
    const int Fw_ref = pad ? fb_next_fast_len(W0 + W1 - 1) : fb_next_fast_len(std::max(W0, W1));
    const int Fh = Fh_ref, Fw = Fw_ref;
    FB_CHECK_ARG(ctx, (long long)Fh * Fw < (1LL << 31));
    if (C == 1 && fb_ncc_small_supported(Fh, Fw, H0, W0, H1, W1, C))
        return fb_ncc_small_launch(ctx, img0, img1, N, H0, W0, H1, W1, Fh, Fw, subpixel, conf_mode, dx, dy, conf);
    int Lh = Fh_ref, Lw = Fw_ref;
    const bool is_long = C == 1 && !ctx->use_rocfft && promote_long_shape(Lh, Lw, H0 + H1 - 1, W0 + W1 - 1, std::max(H0, H1), conf_mode);
    if (is_long || (!ctx->use_rocfft && stream_custom_supported(Fh, Fw, C))) {
        const int hmax = std::max(H0, H1);
        int Fh = Fh_ref, Fw = Fw_ref;
        if (is_long) { Fh = Lh; Fw = Lw; }
        else promote_linear_shape(Fh, Fw, H0 + H1 - 1, W0 + W1 - 1, conf_mode);
        const int nb_max = (int)std::max<size_t>(1, ctx->ncc_arena_limit / custom_bytes_per_pair(Fh, Fw, hmax));
        for (int n0 = 0; n0 < N; n0 += nb_max) {
            const int nb = std::min(nb_max, N - n0);
            int rc2 = ncc_custom_subbatch(ctx, img0 + (size_t)n0 * H0 * W0, img1 + (size_t)n0 * H1 * W1, nb, H0, W0, H1, W1, hmax, std::max(W0, W1), Fh, Fw,
                                          subpixel, conf_mode, dx + n0, dy + n0, conf + n0, nullptr);
            if (rc2) return rc2;
        }
        return FB_OK;
    }
    int rc = ensure_rocfft(ctx);
    if (rc) return rc;
    // sub-batch so that the arena stays under the limit
    const size_t per_pair = (size_t)C * ((size_t)Fh * Fw * 4 + (size_t)Fh * (Fw / 2 + 1) * 8) * 2 + (C > 1 ? (size_t)Fh * (Fw / 2 + 1) * 16 : 0);
    const int nb_max = quantised_chunk(ctx, N, per_pair);
    for (int n0 = 0; n0 < N; n0 += nb_max) {
        const int nb = std::min(nb_max, N - n0);
        rc = ncc_stream_subbatch(ctx, img0 + (size_t)n0 * C * H0 * W0, img1 + (size_t)n0 * C * H1 * W1, nb, C, H0, W0, H1, W1,
                                 Fh, Fw, subpixel, conf_mode, dx + n0, dy + n0, conf + n0, nullptr, nb_max);
        if (rc) return rc;
    }
    return FB_OK;
}

int fb_ncc_batch(fb_ctx* ctx, const float* img0, const float* img1, int N, int C, int H0, int W0, int H1, int W1, int pad,
                 int subpixel, int conf_mode, double* dx, double* dy, float* conf) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && C >= 1 && H0 > 0 && W0 > 0 && H1 > 0 && W1 > 0);
    if (N == 0) return FB_OK;
    FB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t b0 = (size_t)N * C * H0 * W0 * sizeof(float), b1 = (size_t)N * C * H1 * W1 * sizeof(float);
    void *d0 = nullptr, *d1 = nullptr, *dres = nullptr;        // (the context's allocation cache; results in one block: dx, dy, conf)
    int rc = fb_malloc(ctx, b0, &d0);
    if (!rc) rc = fb_malloc(ctx, b1, &d1);
    if (!rc) rc = fb_malloc(ctx, (size_t)N * 24, &dres);
    double* ddx = (double*)dres; double* ddy = ddx ? ddx + N : nullptr; float* dconf = ddy ? (float*)(ddy + N) : nullptr;
    if (!rc) {
        rc = fb_copy_h2d(ctx, d0, img0, b0);
        if (!rc) rc = fb_copy_h2d(ctx, d1, img1, b1);
        if (!rc) rc = fb_ncc_batch_dev(ctx, (const float*)d0, (const float*)d1, N, C, H0, W0, H1, W1, pad, subpixel, conf_mode, ddx, ddy, dconf);
    }
    if (!rc) {
        rc = fb_copy_d2h(ctx, dx, ddx, N * sizeof(double));
        if (!rc) rc = fb_copy_d2h(ctx, dy, ddy, N * sizeof(double));
        if (!rc) rc = fb_copy_d2h(ctx, conf, dconf, N * sizeof(float));
    }
    if (d0) fb_free(ctx, d0);
    if (d1) fb_free(ctx, d1);
    if (dres) fb_free(ctx, dres);
    return rc;
}

int fb_ncc_batch_normalized(fb_ctx* ctx, const float* img0, const float* img1, int N, int C, int H0, int W0, int H1, int W1,
                            const float* mask0, const float* mask1, int pad, int subpixel, int conf_mode, double* dx, double* dy, float* conf) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && C >= 1 && H0 > 0 && W0 > 0 && H1 > 0 && W1 > 0);
    FB_CHECK_ARG(ctx, conf_mode >= 0 && conf_mode <= 2);
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, img0 && img1 && dx && dy && conf);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    const int Fh = pad ? fb_next_fast_len(H0 + H1 - 1) : fb_next_fast_len(std::max(H0, H1));
    const int Fw = pad ? fb_next_fast_len(W0 + W1 - 1) : fb_next_fast_len(std::max(W0, W1));
    const size_t F = (size_t)Fh * Fw;
    FB_CHECK_ARG(ctx, F < ((size_t)1 << 31));
    int rc = ensure_rocfft(ctx);
    if (rc) return rc;
    const size_t b0 = (size_t)N * C * H0 * W0 * sizeof(float), b1 = (size_t)N * C * H1 * W1 * sizeof(float);
    const size_t m0b = (size_t)H0 * W0 * sizeof(float), m1b = (size_t)H1 * W1 * sizeof(float);
    void *d0 = nullptr, *d1 = nullptr, *dm0 = nullptr, *dm1 = nullptr, *dD = nullptr, *dres = nullptr, *dpart = nullptr;
    rc = fb_malloc(ctx, b0, &d0);
    if (!rc) rc = fb_malloc(ctx, b1, &d1);
    if (!rc) rc = fb_malloc(ctx, m0b, &dm0);
    if (!rc) rc = fb_malloc(ctx, m1b, &dm1);
    if (!rc) rc = fb_malloc(ctx, 2 * F * sizeof(float), &dD);
    if (!rc) rc = fb_malloc(ctx, (size_t)(N + 1) * 24, &dres);
    if (!rc) rc = fb_malloc(ctx, kPeakChunks * sizeof(float2), &dpart);
    double* ddx = (double*)dres; double* ddy = ddx ? ddx + (N + 1) : nullptr; float* dconf = ddy ? (float*)(ddy + (N + 1)) : nullptr;
    if (!rc) {
        // the masks (all ones when absent: matcher.py:71-74)
        std::vector<float> ones;
        if (!mask0 || !mask1) ones.assign(std::max((size_t)H0 * W0, (size_t)H1 * W1), 1.0f);
        rc = fb_copy_h2d(ctx, dm0, mask0 ? mask0 : ones.data(), m0b);
        if (!rc) rc = fb_copy_h2d(ctx, dm1, mask1 ? mask1 : ones.data(), m1b);
        if (!rc) rc = fb_copy_h2d(ctx, d0, img0, b0);
        if (!rc) rc = fb_copy_h2d(ctx, d1, img1, b1);
    }
    if (!rc) {
        // NC = irfft2(conj(M0) M1), NC_mirror = irfft2(M0 M1): the correlation pipeline itself on the pair of masks
        rc = ncc_stream_subbatch(ctx, (const float*)dm0, (const float*)dm1, 1, 1, H0, W0, H1, W1, Fh, Fw, 0, FB_CONF_MIRROR, ddx + N, ddy + N, dconf + N, nullptr, 1);
    }
    if (!rc) {
        hipLaunchKernelGGL(ncc_norm_partial_max, dim3(kPeakChunks), dim3(kThreads), 0, ctx->stream, ctx->last_C, ctx->last_Cm, (float2*)dpart, (int)F);
        hipLaunchKernelGGL(ncc_norm_divisors, dim3((unsigned)std::min<size_t>((F + kThreads - 1) / kThreads, 2048)), dim3(kThreads), 0, ctx->stream,
                           ctx->last_C, ctx->last_Cm, (const float2*)dpart, kPeakChunks, (int)F, (float*)dD);
        const hipError_t le = hipGetLastError();               // (no early return: the blocks above are freed below)
        if (le != hipSuccess) rc = fb_fail(ctx, FB_ERR_HIP, "fb_ncc_batch_normalized: %s", hipGetErrorString(le));
    }
    if (!rc) {
        const size_t per_pair = (size_t)C * (F * 4 + (size_t)Fh * (Fw / 2 + 1) * 8) * 2 + (C > 1 ? (size_t)Fh * (Fw / 2 + 1) * 16 : 0);
        const int nb_max = quantised_chunk(ctx, N, per_pair);
        for (int n0 = 0; n0 < N && !rc; n0 += nb_max) {
            const int nb = std::min(nb_max, N - n0);
            rc = ncc_stream_subbatch(ctx, (const float*)d0 + (size_t)n0 * C * H0 * W0, (const float*)d1 + (size_t)n0 * C * H1 * W1, nb, C, H0, W0, H1, W1,
                                     Fh, Fw, subpixel, conf_mode, ddx + n0, ddy + n0, dconf + n0, nullptr, nb_max, (const float*)dD);
        }
    }
    if (!rc) {
        rc = fb_copy_d2h(ctx, dx, ddx, N * sizeof(double));
        if (!rc) rc = fb_copy_d2h(ctx, dy, ddy, N * sizeof(double));
        if (!rc) rc = fb_copy_d2h(ctx, conf, dconf, N * sizeof(float));
    }
    for (void* q : {d0, d1, dm0, dm1, dD, dres, dpart})
        if (q) fb_free(ctx, q);
    return rc;
}

int fb_ncc_blocks_affine_dev(fb_ctx* ctx, const float* imgs0, const float* imgs1, int IH0, int IW0, int IH1, int IW1, int N,
                             const int* blk, const double* aff1, int hmax, int wmax, int Fh, int Fw, int subpixel, int conf_mode, double* dx,
                             double* dy, float* conf) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && IH0 > 0 && IW0 > 0 && IH1 > 0 && IW1 > 0 && Fh > 0 && Fw > 0);
    FB_CHECK_ARG(ctx, conf_mode >= 0 && conf_mode <= 2 && (long long)Fh * Fw < (1LL << 31));
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, imgs0 && imgs1 && blk && dx && dy && conf);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    if (hmax <= 0 || hmax > Fh) hmax = Fh;
    if (wmax <= 0 || wmax > Fw) wmax = Fw;
    if (fb_ncc_small_supported(Fh, Fw, 0, 0, 0, 0, 1))
        return fb_ncc_small_launch_ex(ctx, imgs0, imgs1, N, hmax, wmax, hmax, wmax, blk, IH0, IW0, IH1, IW1, Fh, Fw, subpixel, conf_mode, dx, dy, conf, aff1);
    const bool is_long = !ctx->use_rocfft && promote_long_shape(Fh, Fw, 2 * hmax - 1, 2 * wmax - 1, hmax, conf_mode);
    if (is_long || (!ctx->use_rocfft && stream_custom_supported(Fh, Fw, 1))) {
        if (!is_long) promote_linear_shape(Fh, Fw, 2 * hmax - 1, 2 * wmax - 1, conf_mode);
        const int nbm = (int)std::max<size_t>(1, ctx->ncc_arena_limit / custom_bytes_per_pair(Fh, Fw, hmax));
        CropSrc cs{blk, IH0, IW0, IH1, IW1};
        for (int n0 = 0; n0 < N; n0 += nbm) {
            const int nb = std::min(nbm, N - n0);
            cs.blk = blk + (size_t)n0 * kBlkStride;
            cs.aff = aff1 ? aff1 + (size_t)n0 * FB_AFFINE_STRIDE : nullptr;
            int rc2 = ncc_custom_subbatch(ctx, imgs0, imgs1, nb, 0, 0, 0, 0, hmax, wmax, Fh, Fw, subpixel, conf_mode, dx + n0, dy + n0, conf + n0, &cs);
            if (rc2) return rc2;
        }
        return FB_OK;
    }
    if (aff1) return fb_fail(ctx, FB_ERR_ARG, "fb_ncc_blocks_affine_dev: FFT shape %dx%d has no custom kernel (rocFFT path crops by translation only)", Fh, Fw);
    int rc = ensure_rocfft(ctx);
    if (rc) return rc;
    const size_t per_pair = ((size_t)Fh * Fw * 4 + (size_t)Fh * (Fw / 2 + 1) * 8) * 2;
    const int nb_max = quantised_chunk(ctx, N, per_pair);
    CropSrc crop{blk, IH0, IW0, IH1, IW1};
    for (int n0 = 0; n0 < N; n0 += nb_max) {
        const int nb = std::min(nb_max, N - n0);
        crop.blk = blk + (size_t)n0 * kBlkStride;
        rc = ncc_stream_subbatch(ctx, imgs0, imgs1, nb, 1, 0, 0, 0, 0, Fh, Fw, subpixel, conf_mode, dx + n0, dy + n0, conf + n0, &crop, nb_max);
        if (rc) return rc;
    }
    return FB_OK;
}

}  // extern "C"

// The FFT shape fb_ncc_blocks_dev runs blocks of at most hmax x wmax pixels at when it is asked for Fh x Fw: the shape itself
// (on-chip class, circular axes, FFT_CONF_STD, rocFFT shapes) or the promoted one of promote_linear_shape.  Callers that hold
// several block lists whose shapes promote to the same one may launch them as one list at that shape.
void fb_ncc_launch_shape(fb_ctx* ctx, int Fh, int Fw, int hmax, int wmax, int conf_mode, int* oh, int* ow) {
    *oh = Fh; *ow = Fw;
    if (fb_ncc_small_supported(Fh, Fw, 0, 0, 0, 0, 1)) return;
    if (!ctx->use_rocfft && promote_long_shape(*oh, *ow, 2 * hmax - 1, 2 * wmax - 1, hmax, conf_mode)) return;
    if (!ctx->use_rocfft && stream_custom_supported(Fh, Fw, 1)) promote_linear_shape(*oh, *ow, 2 * hmax - 1, 2 * wmax - 1, conf_mode);
}

extern "C" {

int fb_ncc_blocks_dev(fb_ctx* ctx, const float* imgs0, const float* imgs1, int IH0, int IW0, int IH1, int IW1, int N,
                      const int* blk, int hmax, int wmax, int Fh, int Fw, int subpixel, int conf_mode, double* dx, double* dy, float* conf) {
    return fb_ncc_blocks_affine_dev(ctx, imgs0, imgs1, IH0, IW0, IH1, IW1, N, blk, nullptr, hmax, wmax, Fh, Fw, subpixel, conf_mode, dx, dy, conf);
}

int fb_ncc_last_surfaces(fb_ctx* ctx, float* C_out, float* Cm_out, int* Fh, int* Fw) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, ctx->last_C != nullptr);
    if (Fh) *Fh = ctx->last_Fh;
    if (Fw) *Fw = ctx->last_Fw;
    const size_t bytes = (size_t)ctx->last_N * ctx->last_Fh * ctx->last_Fw * sizeof(float);
    if (C_out) FB_HIP(ctx, hipMemcpy(C_out, ctx->last_C, bytes, hipMemcpyDeviceToHost));
    if (Cm_out && ctx->last_Cm) FB_HIP(ctx, hipMemcpy(Cm_out, ctx->last_Cm, bytes, hipMemcpyDeviceToHost));
    return FB_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// test hook: batched 1-D complex FFT through the LDS mixed-radix core (forward: natural order out)
namespace {
template <bool PAD>
__global__ __launch_bounds__(256) void debug_fft_kernel(const float2* __restrict__ in, float2* __restrict__ out, int M, FftPlan plan,
                                                        const float2* ghi, const float2* glo, int inverse) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int N = plan.n, pitch = (PAD ? fft_padx(N) : N) + 1;
    float2* G = lds;
    float2* thi = G + (size_t)M * pitch;
    float2* tlo = thi + 64;
    load_tw(thi, tlo, ghi, glo);
    for (int i = threadIdx.x; i < M * N; i += blockDim.x) {
        const int m = i / N, e = i - m * N;
        // the inverse consumes digit-reversed input
        const int src = inverse ? fft_pos(plan, e) : e;
        G[m * pitch + (PAD ? fft_padx(src) : src)] = in[i];
    }
    __syncthreads();
    if (inverse) fft_batch_tw<true, TwSplit, 16, PAD>(G, plan, M, 1, pitch, TwSplit{thi, tlo}, false);
    else fft_batch_tw<false, TwSplit, 16, PAD>(G, plan, M, 1, pitch, TwSplit{thi, tlo}, false);
    for (int i = threadIdx.x; i < M * N; i += blockDim.x) {
        const int m = i / N, k = i - m * N;
        const int pos = inverse ? k : fft_pos(plan, k);
        out[i] = G[m * pitch + (PAD ? fft_padx(pos) : pos)];
    }
}
}  // namespace

namespace {
template <int N>
__global__ __launch_bounds__(256) void debug_fft2_kernel(const float2* __restrict__ in, float2* __restrict__ out, int M, FftPlan plan,
                                                         const float2* gtw, int inverse) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int pitch = fft_padx(N) + 1;
    f2* G = reinterpret_cast<f2*>(lds);
    f2* tw = G + (size_t)M * pitch;
    for (int i = threadIdx.x; i < p2_tw_entries(N); i += blockDim.x) tw[i] = (f2){gtw[i].x, gtw[i].y};
    for (int i = threadIdx.x; i < M * N; i += blockDim.x) {
        const int m = i / N, e = i - m * N;
        const int src = inverse ? p2_pos<N>(e) : e;
        G[m * pitch + fft_padx(src)] = (f2){in[i].x, in[i].y};
    }
    __syncthreads();
    if (inverse) p2_fft<N, true>(G, M, pitch, tw);
    else p2_fft<N, false>(G, M, pitch, tw);
    for (int i = threadIdx.x; i < M * N; i += blockDim.x) {
        const int m = i / N, k = i - m * N;
        const int pos = inverse ? k : p2_pos<N>(k);
        const f2 v = G[m * pitch + fft_padx(pos)];
        out[i] = make_float2(v.x, v.y);
    }
}
template <int N>
void launch_debug_fft2(fb_ctx* ctx, size_t lds, const float2* din, float2* dout, int M, const FftPlan& plan, const float2* tw, int inverse) {
    hipFuncSetAttribute((const void*)debug_fft2_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(debug_fft2_kernel<N>, dim3(1), dim3(256), lds, ctx->stream, din, dout, M, plan, tw, inverse);
}
}  // namespace

#ifdef FB_TEST_HOOKS          // only in libfeabas_hip_test.so (include/feabas_hip_test.h)
#include "feabas_hip_test.h"
extern "C" int fb_debug_fft1d(fb_ctx* ctx, const float* in_host, float* out_host, int M, int N, int inverse, int pad) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, M > 0 && N >= 2 && (N <= 4096 || (pad == 2 && N == 8192)) && in_host && out_host);
    FftPlan plan;
    const float2 *hi = nullptr, *lo = nullptr;
    int rc = FB_OK;
    if (N <= 4096) {                        // (the generic plans and two-level twiddles stop there; the packed core has its own)
        if (!fft_make_plan(N, &plan)) return fb_fail(ctx, FB_ERR_ARG, "fb_debug_fft1d: %d is not 5-smooth", N);
        rc = get_split_table(ctx, N, &hi, &lo);
        if (rc) return rc;
    }
    const size_t bytes = sizeof(float2) * (size_t)M * N;
    const size_t lds = ((size_t)M * (N + N / 16 + 2) + 128) * sizeof(float2);
    if (lds > 150 * 1024) return fb_fail(ctx, FB_ERR_ARG, "fb_debug_fft1d: M*N too large for LDS");
    float2 *din = nullptr, *dout = nullptr;
    FB_HIP(ctx, hipMalloc((void**)&din, bytes));
    FB_HIP(ctx, hipMalloc((void**)&dout, bytes));
    FB_HIP(ctx, hipMemcpyAsync(din, in_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    if (pad == 2) {          // power-of-two packed-math core
        const float2* tw;
        rc = get_tw16_table(ctx, N, &tw);
        if (rc) return rc;
        const size_t lds2 = lds + sizeof(float2) * 256;
        switch (N) {
            case 64: launch_debug_fft2<64>(ctx, lds2, din, dout, M, plan, tw, inverse); break;
            case 128: launch_debug_fft2<128>(ctx, lds2, din, dout, M, plan, tw, inverse); break;
            case 256: launch_debug_fft2<256>(ctx, lds2, din, dout, M, plan, tw, inverse); break;
            case 512: launch_debug_fft2<512>(ctx, lds2, din, dout, M, plan, tw, inverse); break;
            case 1024: launch_debug_fft2<1024>(ctx, lds2, din, dout, M, plan, tw, inverse); break;
            case 2048: launch_debug_fft2<2048>(ctx, lds2, din, dout, M, plan, tw, inverse); break;
            case 4096: launch_debug_fft2<4096>(ctx, lds2, din, dout, M, plan, tw, inverse); break;
            case 8192: launch_debug_fft2<8192>(ctx, lds2 + sizeof(float2) * 256, din, dout, M, plan, tw, inverse); break;
            default: hipFree(din); hipFree(dout); return fb_fail(ctx, FB_ERR_ARG, "fb_debug_fft1d: pad=2 needs a power of two in [64, 4096]");
        }
    } else if (pad) {
        FB_HIP(ctx, hipFuncSetAttribute((const void*)debug_fft_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(debug_fft_kernel<true>, dim3(1), dim3(256), lds, ctx->stream, din, dout, M, plan, hi, lo, inverse);
    } else {
        FB_HIP(ctx, hipFuncSetAttribute((const void*)debug_fft_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(debug_fft_kernel<false>, dim3(1), dim3(256), lds, ctx->stream, din, dout, M, plan, hi, lo, inverse);
    }
    FB_HIP(ctx, hipMemcpyAsync(out_host, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    hipFree(din); hipFree(dout);
    return FB_OK;
}
#endif
