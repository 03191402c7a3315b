// Workgroup-cooperative batched 1-D FFTs on LDS-resident complex data, lengths 2^a 3^b 5^c.
//
// Forward = decimation in frequency, in place, natural order in -> digit-reversed order out.
// Inverse = decimation in time, in place, digit-reversed order in -> natural order out (unnormalised).
// Pointwise products commute with the permutation, so a forward / multiply / inverse chain never
// un-scrambles; code that needs a particular frequency uses fft_pos().
//
// Element i of transform m lives at  base[m * ms + i * is]  (float2 units).  `lanes_on_m` picks
// which index consecutive lanes walk, so that the unit-stride index is the fast one (rows: is == 1,
// lanes walk i; columns: ms == 1, lanes walk m): ds_read/write_b64 then sweeps consecutive banks.
#pragma once
#include <hip/hip_runtime.h>

constexpr int kFftMaxFactors = 10;

struct FftPlan {
    int n;                          // length
    int nf;                         // number of radix passes
    int radix[kFftMaxFactors];      // in forward (DIF) order
};

// host: factor n (5-smooth) preferring radix 4.  max_radix = 9 selects the plans of the compile-time small core
// (fb_fft3.h): the composite odd radix 9 next to 8 / 4 / 2 / 5 / 3 -- 72 = 8 x 9, 81 = 9 x 9 (a radix 15 for 75 was
// measured too: its registers cost the occupancy it saves in LDS passes).
static inline bool fft_make_plan(int n, FftPlan* p, int max_radix = 16) {
    p->n = n;
    p->nf = 0;
    int m = n;
    if (max_radix == 9) {
        while (m % 8 == 0) { p->radix[p->nf++] = 8; m /= 8; if (p->nf >= kFftMaxFactors) return false; }
        while (m % 9 == 0) { p->radix[p->nf++] = 9; m /= 9; if (p->nf >= kFftMaxFactors) return false; }
        max_radix = 4;
    }
    if (max_radix == 17) {          // compile-time streaming class (fb_ncc_ct.hip, p3_pick(., 17)): 16, 9, 8, then 4 / 2 / 5 / 3
        while (m % 16 == 0) { p->radix[p->nf++] = 16; m /= 16; if (p->nf >= kFftMaxFactors) return false; }
        while (m % 9 == 0) { p->radix[p->nf++] = 9; m /= 9; if (p->nf >= kFftMaxFactors) return false; }
        max_radix = 8;
    }
    while (max_radix >= 16 && m % 16 == 0) { p->radix[p->nf++] = 16; m /= 16; if (p->nf >= kFftMaxFactors) return false; }
    while (max_radix >= 8 && m % 8 == 0) { p->radix[p->nf++] = 8; m /= 8; if (p->nf >= kFftMaxFactors) return false; }
    while (m % 4 == 0) { p->radix[p->nf++] = 4; m /= 4; if (p->nf >= kFftMaxFactors) return false; }
    while (m % 2 == 0) { p->radix[p->nf++] = 2; m /= 2; if (p->nf >= kFftMaxFactors) return false; }
    while (m % 5 == 0) { p->radix[p->nf++] = 5; m /= 5; if (p->nf >= kFftMaxFactors) return false; }
    while (m % 3 == 0) { p->radix[p->nf++] = 3; m /= 3; if (p->nf >= kFftMaxFactors) return false; }
    return m == 1;
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cmulcf(float2 a, float2 b) {   // a * conj(b)
    return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 caddf(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csubf(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward) / +i (inverse)
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }
__device__ __forceinline__ float2 mul_pi(float2 a) { return make_float2(-a.y, a.x); }

// position of frequency k in the digit-reversed output of the forward transform
__device__ __forceinline__ int fft_pos(const FftPlan& p, int k) {
    int pos = 0, stride = p.n;
    for (int s = 0; s < p.nf; ++s) {
        const int r = p.radix[s];
        stride /= r;
        pos += (k % r) * stride;
        k /= r;
    }
    return pos;
}

// r-point DFT kernels; INV selects the conjugate transform
template <bool INV>
__device__ __forceinline__ void dft2(float2* v) {
    const float2 a = v[0], b = v[1];
    v[0] = caddf(a, b); v[1] = csubf(a, b);
}
template <bool INV>
__device__ __forceinline__ void dft3(float2* v) {
    const float c = -0.5f, s = 0.86602540378443864676f;
    const float2 t1 = caddf(v[1], v[2]);
    const float2 t2 = make_float2(v[0].x + c * t1.x, v[0].y + c * t1.y);
    const float2 d = csubf(v[1], v[2]);
    // forward: -i * s * d ; inverse: +i * s * d
    const float2 t3 = INV ? make_float2(-s * d.y, s * d.x) : make_float2(s * d.y, -s * d.x);
    v[0] = caddf(v[0], t1);
    v[1] = caddf(t2, t3);
    v[2] = csubf(t2, t3);
}
template <bool INV>
__device__ __forceinline__ void dft4(float2* v) {
    const float2 a = caddf(v[0], v[2]), b = csubf(v[0], v[2]);
    const float2 c = caddf(v[1], v[3]), d0 = csubf(v[1], v[3]);
    const float2 d = INV ? mul_pi(d0) : mul_mi(d0);
    v[0] = caddf(a, c); v[2] = csubf(a, c);
    v[1] = caddf(b, d); v[3] = csubf(b, d);
}
template <bool INV>
__device__ __forceinline__ void dft5(float2* v) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;    // cos(2pi/5), cos(4pi/5)
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;     // sin(2pi/5), sin(4pi/5)
    const float2 a1 = caddf(v[1], v[4]), b1 = csubf(v[1], v[4]);
    const float2 a2 = caddf(v[2], v[3]), b2 = csubf(v[2], v[3]);
    const float2 m1 = make_float2(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    const float2 m2 = make_float2(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    // n1 = s1 b1 + s2 b2 ; n2 = s2 b1 - s1 b2 ; forward multiplies them by -i, inverse by +i
    const float2 n1 = make_float2(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    const float2 n2 = make_float2(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    const float2 r1 = INV ? mul_pi(n1) : mul_mi(n1);
    const float2 r2 = INV ? mul_pi(n2) : mul_mi(n2);
    v[0] = caddf(v[0], caddf(a1, a2));
    v[1] = caddf(m1, r1); v[4] = csubf(m1, r1);
    v[2] = caddf(m2, r2); v[3] = csubf(m2, r2);
}

// w_16^k = exp(-2 pi i k / 16), k = 0..9 (the products j2 * p1 that occur in the 4 x 4 and 4 x 2 splits)
__device__ __forceinline__ float2 w16c(int k) {
    const float c1 = 0.92387953251128673848f, s1 = 0.38268343236508978178f, h = 0.70710678118654752440f;
    switch (k) {
        case 0: return make_float2(1.f, 0.f);
        case 1: return make_float2(c1, -s1);
        case 2: return make_float2(h, -h);
        case 3: return make_float2(s1, -c1);
        case 4: return make_float2(0.f, -1.f);
        case 6: return make_float2(-h, -h);
        case 9: return make_float2(-c1, s1);
        default: return make_float2(0.f, 0.f);
    }
}

// 8-point DFT as 4 x 2 Cooley-Tukey in registers (natural order in and out)
template <bool INV>
__device__ __forceinline__ void dft8(float2* v) {
    float2 a[2][4];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[j2][q] = v[j2 + 2 * q];
        dft4<INV>(a[j2]);
    }
#pragma unroll
    for (int p1 = 1; p1 < 4; ++p1) {
        const float2 w = w16c(2 * p1);                 // w_8^p1
        a[1][p1] = INV ? cmulcf(a[1][p1], w) : cmulf(a[1][p1], w);
    }
#pragma unroll
    for (int p1 = 0; p1 < 4; ++p1) {
        v[p1] = caddf(a[0][p1], a[1][p1]);
        v[p1 + 4] = csubf(a[0][p1], a[1][p1]);
    }
}

// 16-point DFT as 4 x 4 Cooley-Tukey in registers
template <bool INV>
__device__ __forceinline__ void dft16(float2* v) {
    float2 a[4][4];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[j2][q] = v[j2 + 4 * q];
        dft4<INV>(a[j2]);
    }
#pragma unroll
    for (int j2 = 1; j2 < 4; ++j2)
#pragma unroll
        for (int p1 = 1; p1 < 4; ++p1) {
            const float2 w = w16c(j2 * p1);
            a[j2][p1] = INV ? cmulcf(a[j2][p1], w) : cmulf(a[j2][p1], w);
        }
#pragma unroll
    for (int p1 = 0; p1 < 4; ++p1) {
        float2 b[4];
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) b[j2] = a[j2][p1];
        dft4<INV>(b);
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) v[p1 + 4 * p2] = b[p2];
    }
}

template <int R, bool INV>
__device__ __forceinline__ void dft_r(float2* v) {
    if (R == 2) dft2<INV>(v);
    else if (R == 3) dft3<INV>(v);
    else if (R == 4) dft4<INV>(v);
    else if (R == 5) dft5<INV>(v);
    else if (R == 8) dft8<INV>(v);
    else dft16<INV>(v);
}

// twiddle sources: a full table (small lengths) or a two-level table w^n = hi[n >> 6] * lo[n & 63]
// (long lengths: 2 x 64 entries instead of N)
struct TwTable {
    const float2* t;
    __device__ __forceinline__ float2 operator()(int n) const { return t[n]; }
};
struct TwSplit {
    const float2* hi;
    const float2* lo;
    __device__ __forceinline__ float2 operator()(int n) const { return cmulf(hi[n >> 6], lo[n & 63]); }
};

// w[q] = w1^q for q = 1..R-1 through squarings/products of depth <= log2(R) (error ~ depth * 6e-8):
// one table lookup per butterfly instead of R-1
template <int R>
__device__ __forceinline__ void twiddle_powers(float2 w1, float2* w) {
    w[1] = w1;
#pragma unroll
    for (int q = 2; q < R; ++q) w[q] = (q & 1) ? cmulf(w[q - 1], w1) : cmulf(w[q / 2], w[q / 2]);
}

// ---- packing two real images into one complex transform, z = img0 + i img1 (matcher.py:63-64 transforms them apart)
// The split of the packed spectrum, A = (Z_k + conj Z_-k) / 2, B = -i (Z_k - conj Z_-k) / 2, carries the rounding of the
// LARGER image into the smaller one's spectrum: with one side of a block (almost) blank -- a masked or saturated region, a
// window that only grazes the texture -- its spectrum drowned in 6e-8 x the other side's and the confidence of such a block
// became noise that changed with the FFT length.  So the weaker image of a tile (rows kernels) or block (on-chip kernel) is
// brought to the magnitude of the stronger one by a power of two (exact) before the transform whenever they differ by 2^6 or
// more -- an extra sweep over the tile in LDS, taken by such tiles only: the maxima ride the barrier that precedes the
// transform anyway -- and the rows kernels divide it out again when they store the split spectra; peak, sub-pixel fit and
// confidences do not depend on the scale of either image.  Tiles of comparable images (every block of a textured pair) are
// left exactly as they were.
__device__ __forceinline__ float2 pack_scales(float m0, float m1) {          // m = max |.| of each image over the tile / block
    // exponent fields (biased); 0 = zero or subnormal, 255 = inf / nan: such tiles are left alone
    const int e0 = (int)((__float_as_uint(m0) >> 23) & 0xffu), e1 = (int)((__float_as_uint(m1) >> 23) & 0xffu);
    const bool ok = e0 > 0 && e1 > 0 && e0 < 255 && e1 < 255;
    const int d = e0 - e1;
    const float s1 = (ok && d >= 6) ? __uint_as_float((unsigned)(127 + d) << 23) : 1.f;      // 2^d, exact
    const float s0 = (ok && -d >= 6) ? __uint_as_float((unsigned)(127 - d) << 23) : 1.f;
    return make_float2(s0, s1);
}
// workgroup-wide max of two per-thread values in two halves around a barrier the caller has anyway: wg_max2_post (every
// thread) before it, wg_max2_read after it.  red: 2 * (threads / 64) floats of LDS that nobody writes again before the next
// barrier.
__device__ __forceinline__ void wg_max2_post(float m0, float m1, float* red) {
    for (int off = 32; off > 0; off >>= 1) { m0 = fmaxf(m0, __shfl_down(m0, off)); m1 = fmaxf(m1, __shfl_down(m1, off)); }
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = m0; red[2 * (threadIdx.x >> 6) + 1] = m1; }
}
__device__ __forceinline__ float2 wg_max2_read(const float* red) {
    const int nw = (blockDim.x + 63) >> 6;
    const float2* r2 = reinterpret_cast<const float2*>(red);
    float2 m = r2[0];
    for (int w = 1; w < nw; ++w) { const float2 v = r2[w]; m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); }
    return m;
}

// LDS skew for unit-stride transforms: one spare slot every 16 elements.  Late passes walk the data with
// strides of 8..128 elements; without the skew 4 to 8 lanes of a half-wave land on the same banks.
__device__ __forceinline__ int fft_padx(int e) { return e + (e >> 4); }

// exact t / d for 0 <= t < 2^22 through a float reciprocal (the integer divide costs ~40 VALU instructions)
__device__ __forceinline__ int fdiv_i(int t, int d, float inv) {
    int q = (int)((float)t * inv);
    int r = t - q * d;
    if (r >= d) { ++q; r -= d; }
    if (r < 0) { --q; }
    return q;
}

// one radix-R pass over M transforms.  L = current block length (multiple of R), Lp = L / R.
// tw[n] = exp(-2 pi i n / N), n < N (LDS).  All offsets are 32-bit (LDS).
template <int R, bool INV, typename TW, bool PAD = false>
__device__ __forceinline__ void fft_pass(float2* base, int N, int M, int is, int ms, int L, const TW& tw,
                                         bool lanes_on_m, int tid, int nthreads) {
    const int Lp = L / R;
    const int per = N / R;                // butterflies per transform
    const int total = per * M;
    const int tstep = N / L;              // twiddle index step: w_L^(j p) = tw[j * p * N / L]
    const float inv_M = 1.0f / (float)M, inv_per = 1.0f / (float)per, inv_Lp = 1.0f / (float)Lp;
    const int leg = Lp * is;
    // two butterflies per trip: both operand sets are read before either result is written, so the LDS
    // latencies of the pair overlap (the compiler cannot prove the two in-place updates independent)
    constexpr int U = (R >= 8) ? 1 : 2;
    for (int t = tid; t < total; t += U * nthreads) {
        const int t2 = t + nthreads;
        const bool has2 = (U == 2) && t2 < total;
        int off[U], jj[U], eoff[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int tt = u ? (has2 ? t2 : t) : t;
            int m, bj;
            if (lanes_on_m) { bj = fdiv_i(tt, M, inv_M); m = tt - bj * M; }
            else { m = fdiv_i(tt, per, inv_per); bj = tt - m * per; }
            const int b = (Lp == per) ? 0 : fdiv_i(bj, Lp, inv_Lp);
            jj[u] = bj - b * Lp;
            off[u] = PAD ? (m * ms) : (m * ms + (b * L + jj[u]) * is);
            if (PAD) eoff[u] = b * L + jj[u];
        }
        float2 v[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < R; ++q) v[u][q] = base[PAD ? off[u] + fft_padx(eoff[u] + q * Lp) : off[u] + q * leg];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (INV) {
                if (L > R) {
                    float2 w[R];
                    if (R >= 8) twiddle_powers<R>(tw(jj[u] * tstep), w);
                    else {
#pragma unroll
                        for (int q = 1; q < R; ++q) w[q] = tw(jj[u] * q * tstep);
                    }
#pragma unroll
                    for (int q = 1; q < R; ++q) v[u][q] = cmulcf(v[u][q], w[q]);  // conj twiddle first (DIT)
                }
                dft_r<R, true>(v[u]);
            } else {
                dft_r<R, false>(v[u]);
                if (L > R) {
                    float2 w[R];
                    if (R >= 8) twiddle_powers<R>(tw(jj[u] * tstep), w);
                    else {
#pragma unroll
                        for (int q = 1; q < R; ++q) w[q] = tw(jj[u] * q * tstep);
                    }
#pragma unroll
                    for (int q = 1; q < R; ++q) v[u][q] = cmulf(v[u][q], w[q]);   // twiddle after (DIF)
                }
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) base[PAD ? off[0] + fft_padx(eoff[0] + q * Lp) : off[0] + q * leg] = v[0][q];
        if (has2) {
#pragma unroll
            for (int q = 0; q < R; ++q) base[PAD ? off[U - 1] + fft_padx(eoff[U - 1] + q * Lp) : off[U - 1] + q * leg] = v[U - 1][q];
        }
    }
}

// full batched transform; every pass ends with a workgroup barrier
template <bool INV, typename TW, int MAXR = 16, bool PAD = false>
__device__ __forceinline__ void fft_batch_tw(float2* base, const FftPlan& plan, int M, int is, int ms, const TW& tw, bool lanes_on_m) {
    const int N = plan.n;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (!INV) {
        int L = N;
        for (int s = 0; s < plan.nf; ++s) {
            const int r = plan.radix[s];
            if (MAXR >= 16 && r == 16) fft_pass<16, false, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            else if (MAXR >= 8 && r == 8) fft_pass<8, false, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            else if (r == 4) fft_pass<4, false, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            else if (r == 2) fft_pass<2, false, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            else if (r == 5) fft_pass<5, false, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            else fft_pass<3, false, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            L /= r;
            __syncthreads();
        }
    } else {
        int L = 1;
        for (int s = plan.nf - 1; s >= 0; --s) {
            const int r = plan.radix[s];
            L *= r;
            if (MAXR >= 16 && r == 16) fft_pass<16, true, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            else if (MAXR >= 8 && r == 8) fft_pass<8, true, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            else if (r == 4) fft_pass<4, true, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            else if (r == 2) fft_pass<2, true, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            else if (r == 5) fft_pass<5, true, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            else fft_pass<3, true, TW, PAD>(base, N, M, is, ms, L, tw, lanes_on_m, tid, nt);
            __syncthreads();
        }
    }
}

template <bool INV>
__device__ __forceinline__ void fft_batch(float2* base, const FftPlan& plan, int M, int is, int ms, const float2* tw, bool lanes_on_m) {
    fft_batch_tw<INV, TwTable, 5>(base, plan, M, is, ms, TwTable{tw}, lanes_on_m);      // small lengths: radix <= 5 keeps VGPRs low
}
