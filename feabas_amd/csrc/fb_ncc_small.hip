// NCC path, on-chip class: one workgroup computes the whole cross-correlation of one block pair
// inside LDS -- load (or crop) both patches, packed 2-D real FFT (both images ride one complex
// transform), conj / plain spectral products, packed inverse (both correlation surfaces ride one
// complex transform), arg-max / mirror-max reduction, 3x3 sub-pixel fit -- and writes 20 bytes.
// HBM traffic = the two patches.  Replaces matcher.xcorr_fft (feabas/matcher.py:22-135) for FFT
// shapes whose working set fits the 160 KiB LDS (e.g. the 75 x 75 fine blocks of the 4k tile pair).
#include "fb_common.h"
#include "fb_ldsfft.h"
#include "fb_fft3.h"

#include <cmath>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace {

constexpr int kSmallThreads = 512;
constexpr int kBlkStrideS = 9;
constexpr size_t kSmallLdsBudget = 150 * 1024;

struct SmallParams {
    int N, Fh, Fw, Sw, RS;
    int H0, W0, H1, W1;                 // stack mode sizes
    int subpixel, conf_mode;
    FftPlan pw, ph;
    const float2* tw_w;
    const float2* tw_h;
    const float* img0;
    const float* img1;
    const int* blk;                     // crop mode when non-null: {img, x0, y0, h0, w0, x1, y1, h1, w1}
    const double* aff;                  // per block affine gather of image 1 (fb_sample_affine) or nullptr
    int IH0, IW0, IH1, IW1;
    double* dx;
    double* dy;
    float* conf;
};

bool small_ct_len(int n) { return n == 64 || n == 72 || n == 75 || n == 80 || n == 81 || n == 90 || n == 96 || n == 100; }

// row pitch (float2 units) of the [Fh][pitch] work array: the 2 Sw packed columns (extra padding columns were measured
// to change nothing: the row passes are not limited by bank conflicts between rows)
int small_pitch(int Fw) {
    const int Sw = Fw / 2 + 1;
    return 2 * Sw;
}

size_t small_lds_bytes(int Fh, int Fw) {
    return ((size_t)Fh * small_pitch(Fw) + Fw + Fh) * sizeof(float2) + 256 + (size_t)(Fw + 2) / 2 * 2 * sizeof(short);
}

__device__ __forceinline__ void merge_peak(float& v, int& i, float v2, int i2) {
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

// The fine-block lengths that occur around the default spacings (blocks of 64..100 px, matcher.py:243-251) run on
// the compile-time packed core (fb_fft3.h, plans of fft_make_plan(n, ., 9)); anything else on the generic core.
#define FB_SMALL_FFT(LEN, INV, COLS, M, TW, ELSE)                                                            \
    switch (LEN) {                                                                                           \
        case 64: p3_fft<64, 9, INV, COLS>(Gp, M, RS, TW); break;                                             \
        case 72: p3_fft<72, 9, INV, COLS>(Gp, M, RS, TW); break;                                             \
        case 75: p3_fft<75, 9, INV, COLS>(Gp, M, RS, TW); break;                                             \
        case 80: p3_fft<80, 9, INV, COLS>(Gp, M, RS, TW); break;                                             \
        case 81: p3_fft<81, 9, INV, COLS>(Gp, M, RS, TW); break;                                             \
        case 90: p3_fft<90, 9, INV, COLS>(Gp, M, RS, TW); break;                                             \
        case 96: p3_fft<96, 9, INV, COLS>(Gp, M, RS, TW); break;                                             \
        case 100: p3_fft<100, 9, INV, COLS>(Gp, M, RS, TW); break;                                           \
        default: ELSE; break;                                                                                \
    }

__global__ __launch_bounds__(kSmallThreads) void ncc_small_fused(const SmallParams prm) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int Fh = prm.Fh, Fw = prm.Fw, Sw = prm.Sw, RS = prm.RS;
    float2* G = lds;                                   // [Fh][RS]
    f2* Gp = reinterpret_cast<f2*>(lds);
    float2* twW = lds + (size_t)Fh * RS;               // [Fw]
    float2* twH = twW + Fw;                            // [Fh]
    float* red = reinterpret_cast<float*>(twH + Fh);   // reduction scratch (<= 256 B)
    short* posW = reinterpret_cast<short*>(red + 64);  // digit-reversed position of every x frequency
    const int n = blockIdx.x;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;

    int h0 = prm.H0, w0 = prm.W0, h1 = prm.H1, w1 = prm.W1;
    const float* s0; const float* s1;
    int ox0 = 0, oy0 = 0, ox1 = 0, oy1 = 0, IH0 = 0, IW0 = 0, IH1 = 0, IW1 = 0;
    if (prm.blk) {
        const int* d = prm.blk + (size_t)n * kBlkStrideS;
        IH0 = prm.IH0; IW0 = prm.IW0; IH1 = prm.IH1; IW1 = prm.IW1;
        s0 = prm.img0 + (size_t)d[0] * IH0 * IW0;
        s1 = prm.img1 + (size_t)d[0] * IH1 * IW1;
        ox0 = d[1]; oy0 = d[2]; h0 = d[3]; w0 = d[4];
        ox1 = d[5]; oy1 = d[6]; h1 = d[7]; w1 = d[8];
    } else {
        s0 = prm.img0 + (size_t)n * h0 * w0;
        s1 = prm.img1 + (size_t)n * h1 * w1;
    }
    // ---- twiddle tables + packed load z = img0 + i img1, zero padded (matcher.py:63-64)
    for (int i = tid; i < Fw; i += nt) { twW[i] = prm.tw_w[i]; posW[i] = (short)fft_pos(prm.pw, i); }
    for (int i = tid; i < Fh; i += nt) twH[i] = prm.tw_h[i];
    // branch-free (clamped address + select) so that the loads of several rows are in flight together
    float fa = 1.f, fb = 1.f;        // 0 for an image that is exactly zero on the block: its spectrum is then exactly zero
    {
        int pitch0, pitch1, maxy0, maxx0, maxy1, maxx1;
        if (prm.blk) { pitch0 = IW0; pitch1 = IW1; maxy0 = IH0 - 1; maxx0 = IW0 - 1; maxy1 = IH1 - 1; maxx1 = IW1 - 1; }
        else { pitch0 = w0; pitch1 = w1; maxy0 = h0 - 1; maxx0 = w0 - 1; maxy1 = h1 - 1; maxx1 = w1 - 1; }
        float m0 = 0.f, m1 = 0.f;
        for (int xc = 0; xc < RS; xc += 64) {
            const int x = xc + lane;
            const int gx0 = ox0 + x, gx1 = ox1 + x;
            const bool vx0 = x < w0 && gx0 >= 0 && gx0 <= maxx0;
            const bool vx1 = x < w1 && gx1 >= 0 && gx1 <= maxx1;
            const int cx0 = min(max(gx0, 0), maxx0), cx1 = min(max(gx1, 0), maxx1);
#pragma unroll 4
            for (int y = wave; y < Fh; y += nwaves) {
                const int gy0 = oy0 + y, gy1 = oy1 + y;
                const bool v0 = vx0 && y < h0 && gy0 >= 0 && gy0 <= maxy0;
                const bool v1 = vx1 && y < h1 && gy1 >= 0 && gy1 <= maxy1;
                const float a = s0[(size_t)min(max(gy0, 0), maxy0) * pitch0 + cx0];
                float b;
                bool v1e = v1;
                if (prm.aff) {          // image 1 through the (deformed, affine-approximated) mesh: bilinear gather, zero outside the image
                    v1e = x < w1 && y < h1;
                    b = fb_sample_affine(s1, IH1, IW1, prm.aff + (size_t)n * FB_AFFINE_STRIDE, min(x, w1 - 1), min(y, h1 - 1));
                } else b = s1[(size_t)min(max(gy1, 0), maxy1) * pitch1 + cx1];
                const float va = v0 ? a : 0.f, vb = v1e ? b : 0.f;
                m0 = fmaxf(m0, fabsf(va)); m1 = fmaxf(m1, fabsf(vb));
                if (x < RS) G[y * RS + x] = make_float2(va, vb);
            }
        }
        // a block one side of which is (almost) blank: the weaker image is brought to the magnitude of the stronger one before
        // the packed transform (pack_scales, fb_ldsfft.h); nothing downstream depends on the scale of either image
        wg_max2_post(m0, m1, red);
        __syncthreads();                                      // G and the tables are complete
        const float2 mm = wg_max2_read(red);
        const float2 sc = pack_scales(mm.x, mm.y);
        fa = mm.x > 0.f ? 1.f : 0.f; fb = mm.y > 0.f ? 1.f : 0.f;
        if (sc.x != 1.f || sc.y != 1.f) {
            for (int i = tid; i < Fh * RS; i += nt) { const float2 z = G[i]; G[i] = make_float2(z.x * sc.x, z.y * sc.y); }
            __syncthreads();
        }
    }
    const int rows_nz = max(h0, h1);
    // ---- forward along x on the non-zero rows
    FB_SMALL_FFT(Fw, false, false, rows_nz, reinterpret_cast<const f2*>(twW), fft_batch<false>(G, prm.pw, rows_nz, 1, RS, twW, false));
    // ---- split the packed row spectra: row y -> [A(kx) | B(kx)], kx < Sw (one wave per row; the wave's
    //      LDS queue is in order, so all reads of a row precede its writes)
    for (int y = wave; y < rows_nz; y += nwaves) {
        float2* row = G + (size_t)y * RS;
        float2 zk[3], zn[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int kx = lane + 64 * q;
            if (kx < Sw) {
                zk[q] = row[posW[kx]];
                zn[q] = row[posW[kx == 0 ? 0 : Fw - kx]];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int kx = lane + 64 * q;
            if (kx < Sw) {
                // the split leaves rounding noise of one image in the other's spectrum: an all-zero image (the reference's rfft2 of
                // zeros) gets exact zeros, so that its correlation surface is exactly zero and the confidence 0 (matcher.py:124-126)
                row[kx] = make_float2(0.5f * fa * (zk[q].x + zn[q].x), 0.5f * fa * (zk[q].y - zn[q].y));            // A = (Zk + conj Zn)/2
                row[Sw + kx] = make_float2(0.5f * fb * (zk[q].y + zn[q].y), -0.5f * fb * (zk[q].x - zn[q].x));     // B = -i (Zk - conj Zn)/2
            }
        }
    }
    __syncthreads();
    // ---- forward along y on the 2 Sw columns
    FB_SMALL_FFT(Fh, false, true, 2 * Sw, reinterpret_cast<const f2*>(twH), fft_batch<false>(G, prm.ph, 2 * Sw, RS, 1, twH, true));
    // ---- spectral products (matcher.py:65, 114): P = conj(F0) F1 over A's slots, Q = F0 F1 over B's
    const bool want_q = prm.conf_mode == FB_CONF_MIRROR;
#ifndef FB_CUT_PRODUCTS
    for (int y = wave; y < Fh; y += nwaves)
    for (int kx = lane; kx < Sw; kx += 64) {
        float2* row = G + y * RS;
        const float2 a = row[kx], b = row[Sw + kx];
        row[kx] = make_float2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
        row[Sw + kx] = want_q ? make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x) : make_float2(0.f, 0.f);
    }
    __syncthreads();
#endif
    // ---- inverse along y
    FB_SMALL_FFT(Fh, true, true, 2 * Sw, reinterpret_cast<const f2*>(twH), fft_batch<true>(G, prm.ph, 2 * Sw, RS, 1, twH, true));
    // ---- Hermitian-extend and pack: W = P + iQ in the digit-reversed order the inverse row pass consumes
    const int nmir = Fw - Sw;                          // kx in [1, nmir] have a mirror Fw - kx >= Sw
    for (int y = wave; y < Fh; y += nwaves) {
        float2* row = G + (size_t)y * RS;
        float2 pk[3], qk[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int kx = lane + 64 * q;
            if (kx < Sw) { pk[q] = row[kx]; qk[q] = row[Sw + kx]; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int kx = lane + 64 * q;
            if (kx < Sw) {
                const bool self = (kx == 0) || (2 * kx == Fw);       // real-valued bins of a real signal (irfft ignores their imag)
                const float2 w = self ? make_float2(pk[q].x, qk[q].x) : make_float2(pk[q].x - qk[q].y, pk[q].y + qk[q].x);
                row[posW[kx]] = w;
                if (kx >= 1 && kx <= nmir)
                    row[posW[Fw - kx]] = make_float2(pk[q].x + qk[q].y, qk[q].x - pk[q].y);     // conj(P) + i conj(Q)
            }
        }
    }
    __syncthreads();
    // ---- inverse along x: row y now holds (C[y][x], Cm[y][x]) x < Fw, un-normalised
    FB_SMALL_FFT(Fw, true, false, Fh, reinterpret_cast<const f2*>(twW), fft_batch<true>(G, prm.pw, Fh, 1, RS, twW, false));
    // ---- reductions (matcher.py:82, 124-125, 130-131)
    float v = -INFINITY; int iv = 0x7fffffff; float mm = 0.f;
    double s = 0.0, ss = 0.0;
    const bool want_std = prm.conf_mode == FB_CONF_STD;
    for (int y = wave; y < Fh; y += nwaves)
    for (int x = lane; x < Fw; x += 64) {
        const float2 c = G[y * RS + x];
        if (c.x > v) { v = c.x; iv = y * Fw + x; }
        mm = fmaxf(mm, fabsf(c.y));
        if (want_std) { s += (double)c.x; ss += (double)c.x * (double)c.x; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float v2 = __shfl_down(v, off);
        const int i2 = __shfl_down(iv, off);
        merge_peak(v, iv, v2, i2);
        mm = fmaxf(mm, __shfl_down(mm, off));
        if (want_std) { s += __shfl_down(s, off); ss += __shfl_down(ss, off); }
    }
    float* rv = red; int* ri = reinterpret_cast<int*>(red + 8); float* rm = red + 16; double* rs = reinterpret_cast<double*>(red + 24);
    if (lane == 0) { rv[wave] = v; ri[wave] = iv; rm[wave] = mm; rs[wave] = s; rs[8 + wave] = ss; }
    __syncthreads();
    if (tid != 0) return;
    {
#pragma clang fp contract(off)
        for (int w = 1; w < nwaves; ++w) {
            merge_peak(v, iv, rv[w], ri[w]);
            mm = fmaxf(mm, rm[w]);
            s += rs[w]; ss += rs[8 + w];
        }
        if (iv == 0x7fffffff) iv = 0;
        const int py = iv / Fw, px = iv - py * Fw;
        double ddx = (double)px, ddy = (double)py;
        if (prm.subpixel) {                            // matcher.py:84-106
            float ct[9];
            for (int j = 0; j < 9; ++j) {
                const int yy = (py + (j / 3 - 1) + Fh) % Fh, xx = (px + (j % 3 - 1) + Fw) % Fw;
                ct[j] = G[(size_t)yy * RS + xx].x;
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
        ddy += (double)(h0 - h1) / 2.0;                // matcher.py:107-110
        ddx += (double)(w0 - w1) / 2.0;
        ddy -= rint(ddy / (double)Fh) * (double)Fh;
        ddx -= rint(ddx / (double)Fw) * (double)Fw;
        prm.dx[n] = ddx; prm.dy[n] = ddy;
        float cf = 1.f;
        if (prm.conf_mode == FB_CONF_MIRROR) {
            cf = 0.f;
            if (v > 0.f) cf = 1.f - mm / v;
            cf = fminf(fmaxf(cf, 0.f), 1.f);
        } else if (prm.conf_mode == FB_CONF_STD) {
            const double F = (double)Fh * (double)Fw;
            const double mean = s / F;
            double var = ss / F - mean * mean;
            if (var < 0) var = 0;
            const float sd32 = (float)sqrt(var);
            const float base32 = 1.0f - expf(-(v / sd32));
            double r = pow((double)base32, F);
            if (!(r >= 0.0)) r = (r != r) ? r : 0.0;
            if (r > 1.0) r = 1.0;
            cf = (float)r;
        }
        prm.conf[n] = cf;
    }
}

// device twiddle tables, one per length, owned by the process (freed at exit with the context's device)
std::map<std::pair<int, int>, float2*> g_tables;

int get_table(fb_ctx* ctx, int n, const float2** out) {
    static std::mutex mtx;                      // process-wide table shared by every context of the device
    std::lock_guard<std::mutex> lk(mtx);
    auto key = std::make_pair(ctx->device, n);
    auto it = g_tables.find(key);
    if (it == g_tables.end()) {
        std::vector<float2> h((size_t)n);
        for (int k = 0; k < n; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)n;
            h[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        float2* d = nullptr;
        FB_HIP(ctx, hipMalloc((void**)&d, sizeof(float2) * (size_t)n));
        FB_HIP(ctx, hipMemcpy(d, h.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
        it = g_tables.emplace(key, d).first;
    }
    *out = it->second;
    return FB_OK;
}

}  // namespace

int fb_ncc_small_supported(int Fh, int Fw, int, int, int, int, int C) {
    if (C != 1 || Fw > 192 || Fh < 1 || Fw < 2) return 0;        // split/pack loops cover Sw <= 192
    FftPlan p;
    if (!fft_make_plan(Fh, &p, 5) || !fft_make_plan(Fw, &p, 5)) return 0;
    return small_lds_bytes(Fh, Fw) <= kSmallLdsBudget;
}

int fb_ncc_small_launch_ex(fb_ctx* ctx, const float* img0, const float* img1, int N, int H0, int W0, int H1, int W1, const int* blk,
                           int IH0, int IW0, int IH1, int IW1, int Fh, int Fw, int subpixel, int conf_mode, double* dx, double* dy,
                           float* conf, const double* aff1) {
    if (fb_ncc_pfa_supported(Fh, Fw, conf_mode))
        return fb_ncc_pfa_launch(ctx, img0, img1, N, H0, W0, H1, W1, blk, IH0, IW0, IH1, IW1, Fh, Fw, subpixel, conf_mode, dx, dy, conf, aff1);
    SmallParams p;
    p.N = N; p.Fh = Fh; p.Fw = Fw; p.Sw = Fw / 2 + 1; p.RS = small_pitch(Fw);
    p.H0 = H0; p.W0 = W0; p.H1 = H1; p.W1 = W1;
    p.subpixel = subpixel; p.conf_mode = conf_mode;
    // positions (posW) must follow the plan the FFT of that length actually runs: radices 9 / 8 / ... on the compile-time
    // core, radix <= 5 on the generic one
    if (!fft_make_plan(Fw, &p.pw, small_ct_len(Fw) ? 9 : 5) || !fft_make_plan(Fh, &p.ph, small_ct_len(Fh) ? 9 : 5))
        return fb_fail(ctx, FB_ERR_ARG, "ncc_small: %dx%d is not 5-smooth", Fh, Fw);
    int rc = get_table(ctx, Fw, &p.tw_w);
    if (rc) return rc;
    rc = get_table(ctx, Fh, &p.tw_h);
    if (rc) return rc;
    p.img0 = img0; p.img1 = img1; p.blk = blk; p.aff = blk ? aff1 : nullptr;
    p.IH0 = IH0; p.IW0 = IW0; p.IH1 = IH1; p.IW1 = IW1;
    p.dx = dx; p.dy = dy; p.conf = conf;
    const size_t lds = small_lds_bytes(Fh, Fw);
    FB_HIP(ctx, hipFuncSetAttribute((const void*)ncc_small_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FB_PROF_B(ctx, "ncc_small_fused", (double)N * (4.0 * ((double)H0 * W0 + (double)H1 * W1) + 20.0));
    static const int nthreads = [] { const char* e = getenv("FB_SMALL_THREADS"); const int v = e ? atoi(e) : 0; return (v >= 64 && v <= kSmallThreads && v % 64 == 0) ? v : kSmallThreads; }();
    hipLaunchKernelGGL(ncc_small_fused, dim3(N), dim3(nthreads), lds, ctx->stream, p);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

int fb_ncc_small_launch(fb_ctx* ctx, const float* img0, const float* img1, int N, int H0, int W0, int H1, int W1, int Fh, int Fw,
                        int subpixel, int conf_mode, double* dx, double* dy, float* conf) {
    return fb_ncc_small_launch_ex(ctx, img0, img1, N, H0, W0, H1, W1, nullptr, 0, 0, 0, 0, Fh, Fw, subpixel, conf_mode, dx, dy, conf);
}
