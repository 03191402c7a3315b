// Power-of-two LDS FFT core with compile-time plans and packed-FP32 arithmetic (v_pk_add/mul/fma_f32:
// one instruction per complex add, two per complex multiply).  Same data layout, pass order and
// digit-reversed output order as the generic core in fb_ldsfft.h (radix 16 passes first, then one
// radix 8/4/2 pass), so fft_pos() and the skewed addressing fft_padx() are shared with it.
//
// Every LDS address of a butterfly is `base + compile-time offset`: with L and R powers of two,
// fft_padx(e0 + q Lp) == fft_padx(e0) + q Lp + (q Lp >> 4) for all the (L, Lp) that occur, so the
// reads/writes become ds_read/write_b64 with immediate offsets and the pass spends no VALU on addresses.
#pragma once
#include <hip/hip_runtime.h>

#include "fb_ldsfft.h"

typedef float f2 __attribute__((ext_vector_type(2)));

// a * w
__device__ __forceinline__ f2 pk_cmul(f2 a, f2 w) {
    const f2 t = a.xx * w;
    f2 r;   // r.lo = a.y * (-w.y) + t.lo ; r.hi = a.y * w.x + t.hi
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w)
__device__ __forceinline__ f2 pk_cmulc(f2 a, f2 w) {
    const f2 t = a.xx * w;              // (a.x w.x, a.x w.y)
    f2 r;   // r.lo = a.y * w.y + t.lo ; r.hi = a.y * w.x - t.hi
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ f2 pk_add_mi(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + (+i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ f2 pk_add_pi(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <bool INV>
__device__ __forceinline__ void pk_dft2(f2* v) {
    const f2 a = v[0], b = v[1];
    v[0] = a + b; v[1] = a - b;
}
template <bool INV>
__device__ __forceinline__ void pk_dft4(f2* v) {
    const f2 a = v[0] + v[2], b = v[0] - v[2];
    const f2 c = v[1] + v[3], d = v[1] - v[3];
    v[0] = a + c; v[2] = a - c;
    v[1] = INV ? pk_add_pi(b, d) : pk_add_mi(b, d);
    v[3] = INV ? pk_add_mi(b, d) : pk_add_pi(b, d);
}
__device__ __forceinline__ f2 pk_w16(int k) {
    const float2 w = w16c(k);
    return (f2){w.x, w.y};
}
template <bool INV>
__device__ __forceinline__ f2 pk_twc(f2 a, int k) {   // a * w_16^k (forward) or its conjugate (inverse), k compile-time
    if (k == 0) return a;
    if (k == 4) return INV ? (f2){-a.y, a.x} : (f2){a.y, -a.x};
    const f2 w = pk_w16(k);
    const f2 wc = {w.x, -w.y};
    const f2 ww = INV ? wc : w;
    const f2 t = a.xx * ww;
    const f2 ws = {-ww.y, ww.x};
    return __builtin_elementwise_fma(a.yy, ws, t);     // constants: the compiler folds the swizzles into SGPR pairs
}
template <bool INV>
__device__ __forceinline__ void pk_dft8(f2* v) {
    f2 a[2][4];
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[j2][q] = v[j2 + 2 * q];
        pk_dft4<INV>(a[j2]);
    }
#pragma unroll
    for (int p1 = 1; p1 < 4; ++p1) a[1][p1] = pk_twc<INV>(a[1][p1], 2 * p1);
#pragma unroll
    for (int p1 = 0; p1 < 4; ++p1) {
        v[p1] = a[0][p1] + a[1][p1];
        v[p1 + 4] = a[0][p1] - a[1][p1];
    }
}
template <bool INV>
__device__ __forceinline__ void pk_dft16(f2* v) {
    f2 a[4][4];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[j2][q] = v[j2 + 4 * q];
        pk_dft4<INV>(a[j2]);
    }
#pragma unroll
    for (int j2 = 1; j2 < 4; ++j2)
#pragma unroll
        for (int p1 = 1; p1 < 4; ++p1) a[j2][p1] = pk_twc<INV>(a[j2][p1], j2 * p1);
#pragma unroll
    for (int p1 = 0; p1 < 4; ++p1) {
        f2 b[4];
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) b[j2] = a[j2][p1];
        pk_dft4<INV>(b);
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) v[p1 + 4 * p2] = b[p2];
    }
}
// 16-point DFT whose inputs v[8..15] are zero (the first pass of a transform whose upper half is zero padding):
// the first-stage 4-point transforms see (x0, x1, 0, 0)
template <bool INV>
__device__ __forceinline__ void pk_dft16_lo8(f2* v) {
    f2 a[4][4];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
        const f2 x0 = v[j2], x1 = v[j2 + 4];
        a[j2][0] = x0 + x1; a[j2][2] = x0 - x1;
        a[j2][1] = INV ? pk_add_pi(x0, x1) : pk_add_mi(x0, x1);
        a[j2][3] = INV ? pk_add_mi(x0, x1) : pk_add_pi(x0, x1);
    }
#pragma unroll
    for (int j2 = 1; j2 < 4; ++j2)
#pragma unroll
        for (int p1 = 1; p1 < 4; ++p1) a[j2][p1] = pk_twc<INV>(a[j2][p1], j2 * p1);
#pragma unroll
    for (int p1 = 0; p1 < 4; ++p1) {
        f2 b[4];
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) b[j2] = a[j2][p1];
        pk_dft4<INV>(b);
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) v[p1 + 4 * p2] = b[p2];
    }
}

template <int R, bool INV>
__device__ __forceinline__ void pk_dft(f2* v) {
    if (R == 2) pk_dft2<INV>(v);
    else if (R == 4) pk_dft4<INV>(v);
    else if (R == 8) pk_dft8<INV>(v);
    else pk_dft16<INV>(v);
}

constexpr int p2_log2(int n) { return n <= 1 ? 0 : 1 + p2_log2(n >> 1); }
constexpr int p2_n16(int N) { return p2_log2(N) / 4; }          // radix-16 passes
constexpr int p2_rem(int N) { return N >> (4 * p2_n16(N)); }     // trailing radix: 1, 2, 4 or 8
constexpr int p2_tw_entries(int N) { return N / 16; }            // tw[i] = exp(-2 pi i / N), i < N / 16

// one radix-R pass (block length L) over M transforms of length N; transform m starts at base + m * pitch
// (pitch >= fft_padx(N)); element e of a transform sits at fft_padx(e).
// position of frequency k in the digit-reversed output of the forward transform (== fft_pos of the generic plan)
template <int N>
__device__ __forceinline__ int p2_pos(int k) {
    int pos = 0, stride = N;
#pragma unroll
    for (int s = 0; s < p2_n16(N); ++s) {
        stride >>= 4;
        pos += (k & 15) * stride;
        k >>= 4;
    }
    return pos + k;          // trailing radix (stride 1)
}

struct P2Store {             // default sink of a pass: write the butterfly back in place
    static constexpr bool kStore = true;
    __device__ __forceinline__ void operator()(int, int, f2) const {}
};

// ZH: the elements N/2 .. N-1 of every transform are zero and are NOT stored in LDS (forward radix-16 pass with L == N only)
template <int N, int R, int L, bool INV, typename SINK = P2Store, bool ZH = false>
__device__ __forceinline__ void p2_pass(f2* base, int M, int pitch, const f2* tw, int tid, int nt, const SINK& sink = SINK()) {
    constexpr int Lp = L / R, per = N / R, lper = p2_log2(per), lLp = p2_log2(Lp);
    const int total = per * M;
    for (int t = tid; t < total; t += nt) {
        const int m = t >> lper, bj = t & (per - 1);
        const int b = bj >> lLp, jj = bj & (Lp - 1);
        const int e0 = b * L + jj;
        f2* p = base + m * pitch + e0 + (e0 >> 4);
        f2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = (ZH && q >= R / 2) ? (f2){0.f, 0.f} : p[q * Lp + ((q * Lp) >> 4)];
        if (L > R) {
            // w[q] = w1^q by squarings / products of depth <= log2(R); each power is applied as soon as it exists so
            // that at most R / 2 of them are live
            f2 w[R];
            w[1] = tw[jj * (N / L)];
            if (INV) {
                v[1] = pk_cmulc(v[1], w[1]);                                  // DIT: conjugate twiddle first
#pragma unroll
                for (int q = 2; q < R; ++q) {
                    w[q] = (q & 1) ? pk_cmul(w[q - 1], w[1]) : pk_cmul(w[q / 2], w[q / 2]);
                    v[q] = pk_cmulc(v[q], w[q]);
                }
                pk_dft<R, true>(v);
            } else {
                if (ZH && R == 16) pk_dft16_lo8<false>(v);
                else pk_dft<R, false>(v);
                v[1] = pk_cmul(v[1], w[1]);                                   // DIF: twiddle after
#pragma unroll
                for (int q = 2; q < R; ++q) {
                    w[q] = (q & 1) ? pk_cmul(w[q - 1], w[1]) : pk_cmul(w[q / 2], w[q / 2]);
                    v[q] = pk_cmul(v[q], w[q]);
                }
            }
        } else {
            pk_dft<R, INV>(v);
        }
        if (SINK::kStore) {
#pragma unroll
            for (int q = 0; q < R; ++q) p[q * Lp + ((q * Lp) >> 4)] = v[q];
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) sink(m, e0 + q * Lp, v[q]);      // (transform, element index, value)
        }
    }
}

template <int N, int S, bool INV>
struct P2Stage {       // S-th radix-16 pass counted in DIF order
    static __device__ __forceinline__ void run(f2* base, int M, int pitch, const f2* tw, int tid, int nt) {
        p2_pass<N, 16, (N >> (4 * S)), INV>(base, M, pitch, tw, tid, nt);
    }
};

// geometry of the pass that ends the forward / starts the inverse transform (block length == radix, no twiddles)
constexpr int p2_last_radix(int N) { return p2_rem(N) > 1 ? p2_rem(N) : 16; }

// forward transform without its last pass / inverse transform without its first pass: the caller fuses
// those two with the pointwise work in between (each touches R consecutive elements of a transform)
template <int N>
__device__ __forceinline__ void p2_fft_fwd_head(f2* base, int M, int pitch, const f2* tw, bool upper_half_zero = false) {
    const int tid = threadIdx.x, nt = blockDim.x;
    constexpr int n16 = p2_n16(N), rem = p2_rem(N), nh = rem > 1 ? n16 : n16 - 1;
    if (nh >= 1) {
        if (upper_half_zero) p2_pass<N, 16, N, false, P2Store, true>(base, M, pitch, tw, tid, nt);
        else P2Stage<N, 0, false>::run(base, M, pitch, tw, tid, nt);
        __syncthreads();
    }
    if (nh >= 2) { P2Stage<N, (nh >= 2 ? 1 : 0), false>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
    if (nh >= 3) { P2Stage<N, (nh >= 3 ? 2 : 0), false>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
}
template <int N, typename SINK = P2Store>
__device__ __forceinline__ void p2_fft_inv_tail(f2* base, int M, int pitch, const f2* tw, const SINK& sink = SINK()) {
    const int tid = threadIdx.x, nt = blockDim.x;
    constexpr int n16 = p2_n16(N), rem = p2_rem(N), nh = rem > 1 ? n16 : n16 - 1;
    if (nh >= 3) { P2Stage<N, (nh >= 3 ? 2 : 0), true>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
    if (nh >= 2) { P2Stage<N, (nh >= 2 ? 1 : 0), true>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
    if (nh >= 1) {
        p2_pass<N, 16, N, true, SINK>(base, M, pitch, tw, tid, nt, sink);
        if (SINK::kStore) __syncthreads();
    }
}

// full transform of M rows; every pass ends with a workgroup barrier
// inverse transform whose last pass (radix 16, L = N: outputs in natural order) hands its results to `sink`
// instead of writing them back (N >= 256 so that the last pass is not also the first)
// NT > 0: the workgroup size as a compile-time constant (the trip counts of the passes are then known: a thread's butterflies of
// one pass are unrolled and interleaved instead of running one after the other)
template <int N, typename SINK, int NT = 0>
__device__ __forceinline__ void p2_fft_inv_sink(f2* base, int M, int pitch, const f2* tw, const SINK& sink) {
    const int tid = threadIdx.x, nt = NT > 0 ? NT : (int)blockDim.x;
    constexpr int n16 = p2_n16(N), rem = p2_rem(N);
    if (rem > 1) { p2_pass<N, (rem > 1 ? rem : 2), (rem > 1 ? rem : 2), true>(base, M, pitch, tw, tid, nt); __syncthreads(); }
    if (n16 >= 3) { P2Stage<N, (n16 >= 3 ? 2 : 0), true>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
    if (n16 >= 2) { P2Stage<N, (n16 >= 2 ? 1 : 0), true>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
    p2_pass<N, 16, N, true, SINK>(base, M, pitch, tw, tid, nt, sink);
}

template <int N, bool INV, int NT = 0>
__device__ __forceinline__ void p2_fft(f2* base, int M, int pitch, const f2* tw) {
    const int tid = threadIdx.x, nt = NT > 0 ? NT : (int)blockDim.x;
    constexpr int n16 = p2_n16(N), rem = p2_rem(N);
    if (!INV) {
        if (n16 >= 1) { P2Stage<N, 0, false>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
        if (n16 >= 2) { P2Stage<N, (n16 >= 2 ? 1 : 0), false>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
        if (n16 >= 3) { P2Stage<N, (n16 >= 3 ? 2 : 0), false>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
        if (rem > 1) { p2_pass<N, (rem > 1 ? rem : 2), (rem > 1 ? rem : 2), false>(base, M, pitch, tw, tid, nt); __syncthreads(); }
    } else {
        if (rem > 1) { p2_pass<N, (rem > 1 ? rem : 2), (rem > 1 ? rem : 2), true>(base, M, pitch, tw, tid, nt); __syncthreads(); }
        if (n16 >= 3) { P2Stage<N, (n16 >= 3 ? 2 : 0), true>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
        if (n16 >= 2) { P2Stage<N, (n16 >= 2 ? 1 : 0), true>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
        if (n16 >= 1) { P2Stage<N, 0, true>::run(base, M, pitch, tw, tid, nt); __syncthreads(); }
    }
}
