// Compile-time mixed-radix LDS FFT (lengths 2^a 3^b 5^c) with packed-FP32 butterflies: the plan of
// fft_make_plan(N, ., MAXR) unrolled at compile time, so that the butterfly / block / twiddle indices of a pass
// are divisions by constants and every operand address is base + constant.  Same pass order, same
// digit-reversed output order and same twiddle table (tw[i] = exp(-2 pi i / N), i < N) as the generic core in
// fb_ldsfft.h, which stays the fallback for lengths that are not instantiated.
//
// Element e of transform m lives at  base[m * ms + e * es].  ROWS: es == 1 and ms == `stride` (lanes walk the
// butterflies of a row); COLS: ms == 1 and es == `stride` (lanes walk the transforms, i.e. the unit-stride index).
#pragma once
#include <hip/hip_runtime.h>

#include "fb_fft2.h"

template <bool INV>
__device__ __forceinline__ void pk_dft3(f2* v) {
    const f2 half = {-0.5f, -0.5f}, s = {0.86602540378443864676f, 0.86602540378443864676f};
    const f2 t1 = v[1] + v[2];
    const f2 t2 = __builtin_elementwise_fma(t1, half, v[0]);
    const f2 d = (v[1] - v[2]) * s;
    v[0] = v[0] + t1;
    v[1] = INV ? pk_add_pi(t2, d) : pk_add_mi(t2, d);       // forward: t2 - i s d
    v[2] = INV ? pk_add_mi(t2, d) : pk_add_pi(t2, d);
}
template <bool INV>
__device__ __forceinline__ void pk_dft5(f2* v) {
    const f2 c1 = {0.30901699437494742410f, 0.30901699437494742410f}, c2 = {-0.80901699437494742410f, -0.80901699437494742410f};
    const f2 s1 = {0.95105651629515357212f, 0.95105651629515357212f}, s2 = {0.58778525229247312917f, 0.58778525229247312917f};
    const f2 a1 = v[1] + v[4], b1 = v[1] - v[4];
    const f2 a2 = v[2] + v[3], b2 = v[2] - v[3];
    const f2 m1 = __builtin_elementwise_fma(a2, c2, __builtin_elementwise_fma(a1, c1, v[0]));
    const f2 m2 = __builtin_elementwise_fma(a2, c1, __builtin_elementwise_fma(a1, c2, v[0]));
    const f2 n1 = __builtin_elementwise_fma(b2, s2, b1 * s1);           // s1 b1 + s2 b2
    const f2 n2 = __builtin_elementwise_fma(b2, -s1, b1 * s2);          // s2 b1 - s1 b2
    v[0] = v[0] + (a1 + a2);
    v[1] = INV ? pk_add_pi(m1, n1) : pk_add_mi(m1, n1);
    v[4] = INV ? pk_add_mi(m1, n1) : pk_add_pi(m1, n1);
    v[2] = INV ? pk_add_pi(m2, n2) : pk_add_mi(m2, n2);
    v[3] = INV ? pk_add_mi(m2, n2) : pk_add_pi(m2, n2);
}
// exp(-2 pi i m / n) as a compile-time constant pair
template <int M_, int N_>
__device__ __forceinline__ f2 pk_wconst() {
    constexpr double a = -6.283185307179586476925286766559 * (double)M_ / (double)N_;
    return (f2){(float)__builtin_cos(a), (float)__builtin_sin(a)};
}
// 9-point DFT as 3 x 3 Cooley-Tukey in registers (natural order in and out): n = 3 n1 + n2, k = k1 + 3 k2
template <bool INV>
__device__ __forceinline__ void pk_dft9(f2* v) {
    f2 y[3][3];
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
#pragma unroll
        for (int n1 = 0; n1 < 3; ++n1) y[n2][n1] = v[3 * n1 + n2];
        pk_dft3<INV>(y[n2]);
    }
    y[1][1] = INV ? pk_cmulc(y[1][1], pk_wconst<1, 9>()) : pk_cmul(y[1][1], pk_wconst<1, 9>());
    y[1][2] = INV ? pk_cmulc(y[1][2], pk_wconst<2, 9>()) : pk_cmul(y[1][2], pk_wconst<2, 9>());
    y[2][1] = INV ? pk_cmulc(y[2][1], pk_wconst<2, 9>()) : pk_cmul(y[2][1], pk_wconst<2, 9>());
    y[2][2] = INV ? pk_cmulc(y[2][2], pk_wconst<4, 9>()) : pk_cmul(y[2][2], pk_wconst<4, 9>());
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        f2 z[3] = {y[0][k1], y[1][k1], y[2][k1]};
        pk_dft3<INV>(z);
#pragma unroll
        for (int k2 = 0; k2 < 3; ++k2) v[k1 + 3 * k2] = z[k2];
    }
}
template <int R, bool INV>
__device__ __forceinline__ void pk_dft_any(f2* v) {
    if (R == 3) pk_dft3<INV>(v);
    else if (R == 5) pk_dft5<INV>(v);
    else if (R == 9) pk_dft9<INV>(v);
    else pk_dft<R, INV>(v);
}

// radix of pass s (DIF order) of fft_make_plan(N, ., MAXR), and the block length that pass works on
constexpr int p3_pick(int m, int maxr) {
    if (maxr == 9)           // the plans of fft_make_plan(., ., 9): 8, 9, then 4 / 2 / 5 / 3
        return (m % 8 == 0) ? 8 : (m % 9 == 0) ? 9 : (m % 4 == 0) ? 4 : (m % 2 == 0) ? 2 : (m % 5 == 0) ? 5 : 3;
    if (maxr == 17)          // the plans of fft_make_plan(., ., 17), streaming class: 16, 9, 8, then 4 / 2 / 5 / 3 (576 = 16 x 9 x 4, 144 = 16 x 9)
        return (m % 16 == 0) ? 16 : (m % 9 == 0) ? 9 : (m % 8 == 0) ? 8 : (m % 4 == 0) ? 4 : (m % 2 == 0) ? 2 : (m % 5 == 0) ? 5 : 3;
    return (maxr >= 16 && m % 16 == 0) ? 16 : (maxr >= 8 && m % 8 == 0) ? 8 : (m % 4 == 0) ? 4 : (m % 2 == 0) ? 2 : (m % 5 == 0) ? 5 : 3;
}
constexpr int p3_nf(int N, int maxr) { int m = N, c = 0; while (m > 1) { m /= p3_pick(m, maxr); ++c; } return c; }
constexpr int p3_len(int N, int maxr, int s) { int m = N; for (int i = 0; i < s; ++i) m /= p3_pick(m, maxr); return m; }   // L of pass s
constexpr int p3_radix(int N, int maxr, int s) { return p3_pick(p3_len(N, maxr, s), maxr); }
constexpr bool p3_smooth(int N) { int m = N; while (m % 2 == 0) m /= 2; while (m % 3 == 0) m /= 3; while (m % 5 == 0) m /= 5; return m == 1 && N >= 2; }

// position of frequency k in the digit-reversed output of the forward transform of plan (N, MAXR)
template <int N, int MAXR, int S = 0>
__device__ __forceinline__ int p3_pos(int k) {
    if constexpr (S >= p3_nf(N, MAXR)) return 0;
    else {
        constexpr int r = p3_radix(N, MAXR, S), stride = p3_len(N, MAXR, S) / r;
        return (k % r) * stride + p3_pos<N, MAXR, S + 1>(k / r);
    }
}

template <int N, int R, int L, bool INV, bool COLS>
__device__ __forceinline__ void p3_pass(f2* base, int M, int stride, const f2* tw, float inv_M, int tid, int nt) {
    constexpr int Lp = L / R, per = N / R, tstep = N / L;
    const int total = per * M;
    const int leg = COLS ? Lp * stride : Lp;
    for (int t = tid; t < total; t += nt) {
        int m, bj;
        if (COLS) { bj = fdiv_i(t, M, inv_M); m = t - bj * M; }
        else { m = t / per; bj = t - m * per; }
        const int b = bj / Lp, jj = bj - b * Lp;
        const int e0 = b * L + jj;
        f2* p = COLS ? base + m + e0 * stride : base + m * stride + e0;
        f2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = p[q * leg];
        if (L > R) {
            f2 w[R];
            if (R >= 8) {
                w[1] = tw[jj * tstep];
#pragma unroll
                for (int q = 2; q < R; ++q) w[q] = (q & 1) ? pk_cmul(w[q - 1], w[1]) : pk_cmul(w[q / 2], w[q / 2]);
            } else {
#pragma unroll
                for (int q = 1; q < R; ++q) w[q] = tw[jj * q * tstep];
            }
            if (INV) {
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = pk_cmulc(v[q], w[q]);
                pk_dft_any<R, true>(v);
            } else {
                pk_dft_any<R, false>(v);
#pragma unroll
                for (int q = 1; q < R; ++q) v[q] = pk_cmul(v[q], w[q]);
            }
        } else {
            pk_dft_any<R, INV>(v);
        }
#pragma unroll
        for (int q = 0; q < R; ++q) p[q * leg] = v[q];
    }
}

template <int N, int MAXR, bool INV, bool COLS, int S, int NF>
struct P3Run {
    static __device__ __forceinline__ void run(f2* base, int M, int stride, const f2* tw, float inv_M, int tid, int nt) {
        constexpr int s = INV ? (NF - 1 - S) : S;                 // DIT walks the passes backwards
        p3_pass<N, p3_radix(N, MAXR, s), p3_len(N, MAXR, s), INV, COLS>(base, M, stride, tw, inv_M, tid, nt);
        __syncthreads();
        P3Run<N, MAXR, INV, COLS, S + 1, NF>::run(base, M, stride, tw, inv_M, tid, nt);
    }
};
template <int N, int MAXR, bool INV, bool COLS, int NF>
struct P3Run<N, MAXR, INV, COLS, NF, NF> {
    static __device__ __forceinline__ void run(f2*, int, int, const f2*, float, int, int) {}
};

// M transforms of length N; every pass ends with a workgroup barrier
template <int N, int MAXR, bool INV, bool COLS>
__device__ __forceinline__ void p3_fft(f2* base, int M, int stride, const f2* tw) {
    P3Run<N, MAXR, INV, COLS, 0, p3_nf(N, MAXR)>::run(base, M, stride, tw, 1.0f / (float)M, threadIdx.x, blockDim.x);
}
