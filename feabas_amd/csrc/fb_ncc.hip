// NCC path, streaming class: batched 2-D R2C / C2R through rocFFT with hand-written
// pad-load, spectral-multiply and peak-reduction kernels around it.
// Replaces matcher.xcorr_fft (feabas/matcher.py:22-135); numerics per SURVEY.md A.1.
#include "fb_common.h"

#include <algorithm>
#include <cmath>

namespace {

constexpr int kPeakChunks = 64;   // partial reductions per surface
constexpr int kThreads = 256;

struct PeakPartial {
    float vmax;      // max of C in the chunk
    int imax;        // first index achieving it
    float mmax;      // max |Cm|
    int pad_;
    double sum;      // sum C      (STD confidence)
    double sumsq;    // sum C^2
};

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

// block descriptor of the crop loader: {image, x0, y0, h0, w0, x1, y1, h1, w1}
constexpr int kBlkStride = 9;

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

__device__ __forceinline__ void peak_merge(float& v, int& i, float v2, int i2) {
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
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
                               double* __restrict__ dx, double* __restrict__ dy, float* __restrict__ conf, int N) {
#pragma clang fp contract(off)
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (blk) {      // per-block sizes (crop mode)
        const int* d = blk + (size_t)n * kBlkStride;
        H0 = d[3]; W0 = d[4]; H1 = d[7]; W1 = d[8];
    }
    const PeakPartial* p = part + (size_t)n * nchunks;
    float v = p[0].vmax; int iv = p[0].imax; float mm = p[0].mmax; double s = p[0].sum, ss = p[0].sumsq;
    for (int c = 1; c < nchunks; ++c) {
        peak_merge(v, iv, p[c].vmax, p[c].imax);
        mm = fmaxf(mm, p[c].mmax);
        s += p[c].sum; ss += p[c].sumsq;
    }
    if (iv == 0x7fffffff) iv = 0;      // all -inf / NaN surface: numpy argmax -> 0
    const int py = iv / Fw, px = iv - py * Fw;
    double ddx = (double)px, ddy = (double)py;
    if (subpixel) {
        const float* c = Csurf + (size_t)n * Fh * Fw;
        float ct[9];
        for (int j = 0; j < 9; ++j) {
            int yy = py + (j / 3 - 1), xx = px + (j % 3 - 1);
            yy = (yy + Fh) % Fh; xx = (xx + Fw) % Fw;
            ct[j] = c[(size_t)yy * Fw + xx];
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

int ensure_rocfft(fb_ctx* ctx) {
    if (ctx->rocfft_ready) return FB_OK;
    FB_FFT(ctx, rocfft_setup());
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
struct CropSrc { const int* blk; int IH0, IW0, IH1, IW1; };

int ncc_stream_subbatch(fb_ctx* ctx, const float* img0, const float* img1, int nb, int C, int H0, int W0, int H1,
                        int W1, int Fh, int Fw, int subpixel, int conf_mode, double* dx, double* dy, float* conf,
                        const CropSrc* crop = nullptr, int nbq = 0) {
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
    {
        FB_PROF(ctx, "ncc_peak_partial");
        hipLaunchKernelGGL(ncc_peak_partial, dim3(kPeakChunks, nreal), dim3(kThreads), 0, ctx->stream, Csurf, Msurf, part,
                           (int)F, want_q, conf_mode == FB_CONF_STD);
    }
    {
        FB_PROF(ctx, "ncc_peak_final");
        hipLaunchKernelGGL(ncc_peak_final, dim3(fb_cdiv(nreal, 64)), dim3(64), 0, ctx->stream, Csurf, part, kPeakChunks, Fh, Fw,
                           H0, W0, H1, W1, crop ? crop->blk : (const int*)nullptr, subpixel, conf_mode, dx, dy, conf, nreal);
    }
    FB_HIP(ctx, hipGetLastError());
    ctx->last_Fh = Fh; ctx->last_Fw = Fw; ctx->last_N = nreal;
    ctx->last_C = Csurf; ctx->last_Cm = want_q ? Msurf : nullptr;
    return FB_OK;
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
    FB_CHECK_ARG(ctx, N >= 0 && C >= 1 && H0 > 0 && W0 > 0 && H1 > 0 && W1 > 0);
    FB_CHECK_ARG(ctx, conf_mode >= 0 && conf_mode <= 2);
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, img0 && img1 && dx && dy && conf);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    const int Fh = pad ? fb_next_fast_len(H0 + H1 - 1) : fb_next_fast_len(std::max(H0, H1));
    const int Fw = pad ? fb_next_fast_len(W0 + W1 - 1) : fb_next_fast_len(std::max(W0, W1));
    FB_CHECK_ARG(ctx, (long long)Fh * Fw < (1LL << 31));
    if (C == 1 && fb_ncc_small_supported(Fh, Fw, H0, W0, H1, W1, C))
        return fb_ncc_small_launch(ctx, img0, img1, N, H0, W0, H1, W1, Fh, Fw, subpixel, conf_mode, dx, dy, conf);
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
    FB_CHECK_ARG(ctx, N >= 0 && C >= 1 && H0 > 0 && W0 > 0 && H1 > 0 && W1 > 0);
    if (N == 0) return FB_OK;
    FB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t b0 = (size_t)N * C * H0 * W0 * sizeof(float), b1 = (size_t)N * C * H1 * W1 * sizeof(float);
    float *d0 = nullptr, *d1 = nullptr, *dconf = nullptr;
    double *ddx = nullptr, *ddy = nullptr;
    int rc = FB_OK;
    hipError_t e;
    if ((e = hipMalloc(&d0, b0)) != hipSuccess || (e = hipMalloc(&d1, b1)) != hipSuccess ||
        (e = hipMalloc(&ddx, N * sizeof(double))) != hipSuccess || (e = hipMalloc(&ddy, N * sizeof(double))) != hipSuccess ||
        (e = hipMalloc(&dconf, N * sizeof(float))) != hipSuccess) {
        rc = fb_fail(ctx, FB_ERR_NOMEM, "fb_ncc_batch: hipMalloc: %s", hipGetErrorString(e));
    }
    if (!rc) {
        hipMemcpyAsync(d0, img0, b0, hipMemcpyHostToDevice, ctx->stream);
        hipMemcpyAsync(d1, img1, b1, hipMemcpyHostToDevice, ctx->stream);
        rc = fb_ncc_batch_dev(ctx, d0, d1, N, C, H0, W0, H1, W1, pad, subpixel, conf_mode, ddx, ddy, dconf);
    }
    if (!rc) {
        hipMemcpyAsync(dx, ddx, N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        hipMemcpyAsync(dy, ddy, N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        hipMemcpyAsync(conf, dconf, N * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fb_fail(ctx, FB_ERR_HIP, "fb_ncc_batch: %s", hipGetErrorString(e));
    }
    hipFree(d0); hipFree(d1); hipFree(ddx); hipFree(ddy); hipFree(dconf);
    return rc;
}

int fb_ncc_blocks_dev(fb_ctx* ctx, const float* imgs0, const float* imgs1, int IH0, int IW0, int IH1, int IW1, int N,
                      const int* blk, int Fh, int Fw, int subpixel, int conf_mode, double* dx, double* dy, float* conf) {
    FB_CHECK_ARG(ctx, N >= 0 && IH0 > 0 && IW0 > 0 && IH1 > 0 && IW1 > 0 && Fh > 0 && Fw > 0);
    FB_CHECK_ARG(ctx, conf_mode >= 0 && conf_mode <= 2 && (long long)Fh * Fw < (1LL << 31));
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, imgs0 && imgs1 && blk && dx && dy && conf);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    if (fb_ncc_small_supported(Fh, Fw, 0, 0, 0, 0, 1))
        return fb_ncc_small_launch_ex(ctx, imgs0, imgs1, N, 0, 0, 0, 0, blk, IH0, IW0, IH1, IW1, Fh, Fw, subpixel, conf_mode, dx, dy, conf);
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

int fb_ncc_last_surfaces(fb_ctx* ctx, float* C_out, float* Cm_out, int* Fh, int* Fw) {
    FB_CHECK_ARG(ctx, ctx->last_C != nullptr);
    if (Fh) *Fh = ctx->last_Fh;
    if (Fw) *Fw = ctx->last_Fw;
    const size_t bytes = (size_t)ctx->last_N * ctx->last_Fh * ctx->last_Fw * sizeof(float);
    if (C_out) FB_HIP(ctx, hipMemcpy(C_out, ctx->last_C, bytes, hipMemcpyDeviceToHost));
    if (Cm_out && ctx->last_Cm) FB_HIP(ctx, hipMemcpy(Cm_out, ctx->last_Cm, bytes, hipMemcpyDeviceToHost));
    return FB_OK;
}

}  // extern "C"
