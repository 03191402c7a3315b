// NCC path, on-chip class, register-resident form for the 75 x 75 transform of the fine blocks of a 4k tile pair
// (blocks of 75 x 73 px, pad = False: matcher.py:59-62, 701-705).  Replaces matcher.xcorr_fft (feabas/matcher.py:22-135)
// for that shape; every other on-chip shape stays on ncc_small_fused (fb_ncc_small.hip).
//
// The cross-correlation of one block pair is, as there, ONE packed complex 2-D transform forward (z = img0 + i img1),
// a pointwise step, and ONE packed complex 2-D transform back (real part = C, imaginary part = the mirror surface of
// matcher.py:113-128).  What is different is how the 75 x 75 transform is cut:
//
//   * 75 = 3 x 25 with coprime factors, so each axis is a 3 x 25 two-dimensional DFT WITHOUT twiddles between the
//     factors (Good-Thomas): sample n sits at digits (n mod 3, 17 n mod 25), frequency k at (k mod 3, k mod 25).  The
//     tile in LDS is the 4-D array [y1][y2][x1][x2] (row pitch 75 complex), and the whole 2-D transform is four
//     separable passes over it: 25-point along x2, 25-point along y2, and the two 3-point ones as one 3 x 3 block.
//   * a thread owns one 25-point transform (or one pair of 3 x 3 blocks) IN REGISTERS: every index, every twiddle of
//     a pass is a compile-time constant, every LDS operand is `per-thread base + immediate offset`.  No index
//     arithmetic, no twiddle table, no digit-reversal table; five workgroup barriers instead of fifteen.
//   * the pointwise step never forms the two spectra.  With Zk = Z(k), Zn = Z(-k): F0 = (Zk + conj Zn) / 2,
//     F1 = -i (Zk - conj Zn) / 2, and the packed inverse input  conj(F0) F1 + i F0 F1  equals
//     (Re F0 - Im F0) (1 + i) F1 -- a real scalar times a rotated F1: six packed instructions per frequency pair.
//     Frequencies k and -k sit in the 3 x 3 blocks of (y2, x2) and (-y2, -x2); one thread takes both blocks, so the
//     forward 3 x 3 step, the pointwise step and the inverse 3 x 3 step are one trip through LDS.
//   * peak / mirror maximum are reduced from the registers of the last pass; only the real surface goes back to LDS
//     (for the 3 x 3 sub-pixel neighbourhood, matcher.py:84-106).
//
// 320 threads (225 transforms of 25 points per pass, 313 block pairs in the 3 x 3 pass), 45 KB of LDS: three
// workgroups per CU.
#include "fb_common.h"
#include "fb_ldsfft.h"
#include "fb_fft3.h"

#include <cmath>
#include <cstdlib>

namespace {

#ifndef FB_PFA_THREADS
#define FB_PFA_THREADS 256
#endif
#ifndef FB_PFA_WPE
#define FB_PFA_WPE 3
#endif
constexpr int kPfaN = 75, kPfaThreads = FB_PFA_THREADS;
constexpr int kPfaSlots = kPfaN * kPfaN;          // complex slots of the tile
constexpr int kBlkStrideP = 9;

struct PfaParams {
    int N;
    int H0, W0, H1, W1;                 // stack mode sizes
    int subpixel, conf_mode;
    const float* img0;
    const float* img1;
    const int* blk;                     // crop mode when non-null: {img, x0, y0, h0, w0, x1, y1, h1, w1}
    const double* aff;                  // per block affine gather of image 1 (fb_sample_affine) or nullptr
    int IH0, IW0, IH1, IW1;
    double* dx;
    double* dy;
    float* conf;
};

// a * exp(-+ 2 pi i M / N) with the constant folded into the instruction operands
template <int M_, int N_, bool INV>
__device__ __forceinline__ f2 pk_mulw(f2 a) {
    constexpr double ang = -6.283185307179586476925286766559 * (double)M_ / (double)N_;
    const float c = (float)__builtin_cos(ang), s = INV ? -(float)__builtin_sin(ang) : (float)__builtin_sin(ang);
    const f2 w = {c, s}, ws = {-s, c};
    return __builtin_elementwise_fma(a.yy, ws, a.xx * w);
}

// 25-point DFT in registers, 5 x 5 Cooley-Tukey.  In: v[n] natural order.  Out: frequency k1 + 5 k2 in v[5 k1 + k2]
// (reg25_out() gives the register of an output index).  INV: conjugate transform, unnormalised.
template <bool INV, int N2 = 1>
struct Reg25Tw {            // twiddles of column n2, applied to its outputs k1 = 1..4
    static __device__ __forceinline__ void run(f2* v) {
        v[5 * 1 + N2] = pk_mulw<1 * N2, 25, INV>(v[5 * 1 + N2]);
        v[5 * 2 + N2] = pk_mulw<2 * N2, 25, INV>(v[5 * 2 + N2]);
        v[5 * 3 + N2] = pk_mulw<3 * N2, 25, INV>(v[5 * 3 + N2]);
        v[5 * 4 + N2] = pk_mulw<4 * N2, 25, INV>(v[5 * 4 + N2]);
        if constexpr (N2 < 4) Reg25Tw<INV, N2 + 1>::run(v);
    }
};
template <bool INV>
__device__ __forceinline__ void reg_fft25(f2* v) {
    // Y[n2][k1] = sum_n1 x[5 n1 + n2] w5^(n1 k1), left in v[5 k1 + n2]
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        f2 t[5] = {v[n2], v[5 + n2], v[10 + n2], v[15 + n2], v[20 + n2]};
        pk_dft5<INV>(t);
        v[n2] = t[0]; v[5 + n2] = t[1]; v[10 + n2] = t[2]; v[15 + n2] = t[3]; v[20 + n2] = t[4];
    }
    Reg25Tw<INV>::run(v);       // Y[n2][k1] *= w25^(n2 k1)
    // X[k1 + 5 k2] = sum_n2 Y[n2][k1] w5^(n2 k2), left in v[5 k1 + k2]
#pragma unroll
    for (int k1 = 0; k1 < 5; ++k1) pk_dft5<INV>(v + 5 * k1);
}
constexpr int reg25_out(int k) { return 5 * (k % 5) + k / 5; }

// 3 x 3 two-dimensional DFT of a block b[3 y1 + x1], in place, natural order
template <bool INV>
__device__ __forceinline__ void reg_dft3x3(f2* b) {
#pragma unroll
    for (int r = 0; r < 3; ++r) pk_dft3<INV>(b + 3 * r);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f2 t[3] = {b[c], b[3 + c], b[6 + c]};
        pk_dft3<INV>(t);
        b[c] = t[0]; b[3 + c] = t[1]; b[6 + c] = t[2];
    }
}

// the packed inverse input at k (-> wk) and at -k (-> wn) from Zk = Z(k), Zn = Z(-k)   (4 x the values of the header)
__device__ __forceinline__ void pfa_pointwise(f2 zk, f2 zn, f2& wk, f2& wn) {
    f2 u, d, ef, s;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(u) : "v"(zk), "v"(zn));                                 // Zk + conj Zn = 2 F0
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(d) : "v"(zk), "v"(zn));                                 // Zk - conj Zn = 2 i F1
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(ef) : "v"(d), "v"(d));     // (d.y + d.x, d.y - d.x)
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(s) : "v"(u), "v"(u));      // (u.x - u.y, u.x + u.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(wk) : "v"(s), "v"(ef));                 // s.x * (e, f)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(wn) : "v"(s), "v"(ef));                 // s.y * (f, e)
}

__device__ __forceinline__ int pfa_digits(int n) {        // slot of sample n along one axis: 25 (n mod 3) + (17 n mod 25)
    const int q3 = (n * 171) >> 9, m = 17 * n;
    return 25 * (n - 3 * q3) + (m - 25 * ((m * 1311) >> 15));
}

// AFF: image 1 is gathered through a per-block affine map (prm.aff)
template <bool AFF>
__global__ __launch_bounds__(kPfaThreads) __attribute__((amdgpu_waves_per_eu(FB_PFA_WPE, FB_PFA_WPE))) void ncc_pfa75(const PfaParams prm) {
    __shared__ __attribute__((aligned(16))) f2 tile[kPfaSlots];
    __shared__ float red[16];
    __shared__ int red_idx;
    const int n = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    int h0 = prm.H0, w0 = prm.W0, h1 = prm.H1, w1 = prm.W1;
    const float* s0; const float* s1;
    int ox0 = 0, oy0 = 0, ox1 = 0, oy1 = 0, IH1 = 0, IW1 = 0;
    int pitch0, pitch1, maxy0, maxx0, maxy1, maxx1;
    if (prm.blk) {
        const int* d = prm.blk + (size_t)n * kBlkStrideP;
        IH1 = prm.IH1; IW1 = prm.IW1;
        s0 = prm.img0 + (size_t)d[0] * prm.IH0 * prm.IW0;
        s1 = prm.img1 + (size_t)d[0] * IH1 * IW1;
        ox0 = d[1]; oy0 = d[2]; h0 = d[3]; w0 = d[4];
        ox1 = d[5]; oy1 = d[6]; h1 = d[7]; w1 = d[8];
        pitch0 = prm.IW0; pitch1 = IW1; maxy0 = prm.IH0 - 1; maxx0 = prm.IW0 - 1; maxy1 = IH1 - 1; maxx1 = IW1 - 1;
    } else {
        s0 = prm.img0 + (size_t)n * h0 * w0;
        s1 = prm.img1 + (size_t)n * h1 * w1;
        pitch0 = w0; pitch1 = w1; maxy0 = h0 - 1; maxx0 = w0 - 1; maxy1 = h1 - 1; maxx1 = w1 - 1;
    }
    // ---- packed load z = img0 + i img1, zero padded (matcher.py:63-64): thread = column x, rows yg, yg + 3, ... (25 of them).
    //      The rows of the block that exist in the image are the records of a buffer descriptor (wave-uniform), so a row
    //      outside the block or the image is dropped by the range check of the load and reads as zero: one address add per
    //      element, nothing else; a column outside them gets an offset that is out of range on every row.
    constexpr int kYG = 3, kRowsPer = kPfaN / kYG;
    static_assert(kPfaThreads >= kYG * kPfaN, "one thread per (column, row group)");
    const int yg = (tid >= 75) + (tid >= 150) + (tid >= 225), x = tid - 75 * yg;
    const bool act = tid < kYG * kPfaN;
    float a[kRowsPer], b[kRowsPer];
    {
        const int ylo0 = max(0, -oy0), nrow0 = max(min(h0, maxy0 + 1 - oy0) - ylo0, 0);
        const auto rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s0 + (ptrdiff_t)(oy0 + ylo0) * pitch0), 0, nrow0 * pitch0 * 4, 0x00020000);
        const int gx0 = ox0 + x;
        const bool vx0 = act && x < w0 && gx0 >= 0 && gx0 <= maxx0;
        const int vo0 = vx0 ? ((yg - ylo0) * pitch0 + gx0) * 4 : (int)0x80000000;
        const int st0 = kYG * pitch0 * 4;
#pragma unroll
        for (int j = 0; j < kRowsPer; ++j) a[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs0, vo0 + j * st0, 0, 0));
        if (AFF) {
            // image 1 through the (deformed, affine-approximated) mesh: bilinear gather, zero outside the image.  One sample
            // at a time (double precision map, four taps each), parked in the thread's own words of the tile
            float* park = reinterpret_cast<float*>(tile) + tid;
            const double* prm_aff = prm.aff + (size_t)n * FB_AFFINE_STRIDE;
            const bool v1x = act && x < w1;
#pragma unroll 1
            for (int j = 0; j < kRowsPer; ++j) {
                const int y = yg + kYG * j;
                const float vb = fb_sample_affine(s1, IH1, IW1, prm_aff, min(x, w1 - 1), min(y, h1 - 1));
                park[j * kPfaThreads] = (v1x && y < h1) ? vb : 0.f;
            }
#pragma unroll
            for (int j = 0; j < kRowsPer; ++j) b[j] = park[j * kPfaThreads];
        } else {
            const int ylo1 = max(0, -oy1), nrow1 = max(min(h1, maxy1 + 1 - oy1) - ylo1, 0);
            const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s1 + (ptrdiff_t)(oy1 + ylo1) * pitch1), 0, nrow1 * pitch1 * 4, 0x00020000);
            const int gx1 = ox1 + x;
            const bool vx1 = act && x < w1 && gx1 >= 0 && gx1 <= maxx1;
            const int vo1 = vx1 ? ((yg - ylo1) * pitch1 + gx1) * 4 : (int)0x80000000;
            const int st1 = kYG * pitch1 * 4;
#pragma unroll
            for (int j = 0; j < kRowsPer; ++j) b[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs1, vo1 + j * st1, 0, 0));
        }
    }
    if (tid == 0) red_idx = 0x7fffffff;
    float m0 = 0.f, m1 = 0.f;
#pragma unroll
    for (int j = 0; j < kRowsPer; ++j) { m0 = fmaxf(m0, fabsf(a[j])); m1 = fmaxf(m1, fabsf(b[j])); }
    // a block one side of which is (almost) blank: the weaker image is brought to the magnitude of the stronger one before
    // the packed transform (pack_scales, fb_ldsfft.h); nothing downstream depends on the scale of either image
    wg_max2_post(m0, m1, red);
    __syncthreads();
    const float2 mm2 = wg_max2_read(red);
    if (!(mm2.x > 0.f) || !(mm2.y > 0.f)) {
        // an image that is exactly zero on the block: its spectrum, both correlation surfaces and the confidence are
        // exactly zero (matcher.py:124-126); the first maximum of the zero surface is index 0
        if (tid == 0) {
            double ddy = (double)(h0 - h1) / 2.0, ddx = (double)(w0 - w1) / 2.0;        // matcher.py:107-110
            ddy -= rint(ddy / (double)kPfaN) * (double)kPfaN;
            ddx -= rint(ddx / (double)kPfaN) * (double)kPfaN;
            prm.dx[n] = ddx; prm.dy[n] = ddy;
            prm.conf[n] = prm.conf_mode == FB_CONF_MIRROR ? 0.f : 1.f;
        }
        return;
    }
    {
        const float2 sc = pack_scales(mm2.x, mm2.y);
        if (sc.x != 1.f || sc.y != 1.f) {
#pragma unroll
            for (int j = 0; j < kRowsPer; ++j) { a[j] *= sc.x; b[j] *= sc.y; }
        }
        // row yg + 3 j sits in row slot 25 yg + (17 yg + j) mod 25 (51 = 1 mod 25): consecutive slots with one wrap
        if (act) {
            const int c = 17 * yg - 25 * (yg == 2);
            f2* q = tile + (25 * yg + c) * kPfaN + pfa_digits(x);
#pragma unroll
            for (int j = 0; j < kRowsPer; ++j) (j >= 25 - c ? q - 25 * kPfaN : q)[j * kPfaN] = (f2){a[j], b[j]};
        }
    }
    __syncthreads();

    // ---- forward, 25-point along x2: item (row slot, x1) = 25 consecutive slots
    if (tid < 225) {
        f2* p = tile + 25 * tid;
        f2 v[25];
#pragma unroll
        for (int q = 0; q < 25; ++q) v[q] = p[q];
        reg_fft25<false>(v);
#pragma unroll
        for (int k = 0; k < 25; ++k) p[k] = v[reg25_out(k)];
    }
    __syncthreads();
    // ---- forward, 25-point along y2: item (y1, column slot), elements one row apart
    const int c_y1 = (tid >= 75) + (tid >= 150), c_x = tid - 75 * c_y1;
    if (tid < 225) {
        f2* p = tile + c_y1 * (25 * kPfaN) + c_x;
        f2 v[25];
#pragma unroll
        for (int q = 0; q < 25; ++q) v[q] = p[q * kPfaN];
        reg_fft25<false>(v);
#pragma unroll
        for (int k = 0; k < 25; ++k) p[k * kPfaN] = v[reg25_out(k)];
    }
    __syncthreads();
    // ---- 3 x 3 forward, pointwise products (matcher.py:65, 114), 3 x 3 inverse: item = the blocks of (y2, x2) and (-y2, -x2)
#pragma unroll 1
    for (int it = tid; it < 313; it += kPfaThreads) {
        int y2, x2;
        if (it < 300) { const int q = (it * 1311) >> 15; y2 = 1 + q; x2 = it - 25 * q; }
        else { y2 = 0; x2 = it - 300; }
        const int y2n = y2 ? 25 - y2 : 0, x2n = x2 ? 25 - x2 : 0;
        f2* pa = tile + y2 * kPfaN + x2;
        f2* pb = tile + y2n * kPfaN + x2n;
        f2 za[9], zb[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) { za[3 * r + c] = pa[r * (25 * kPfaN) + c * 25]; zb[3 * r + c] = pb[r * (25 * kPfaN) + c * 25]; }
        reg_dft3x3<false>(za);
        reg_dft3x3<false>(zb);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int m = 3 * r + c, mn = 3 * ((3 - r) % 3) + (3 - c) % 3;      // -k within the 3 x 3 block
                f2 wk, wn;
                pfa_pointwise(za[m], zb[mn], wk, wn);
                za[m] = wk; zb[mn] = wn;
            }
        reg_dft3x3<true>(za);
        reg_dft3x3<true>(zb);
        // the block pair of (0, 0) is the same block twice: both copies hold the same values
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) { pa[r * (25 * kPfaN) + c * 25] = za[3 * r + c]; pb[r * (25 * kPfaN) + c * 25] = zb[3 * r + c]; }
    }
    __syncthreads();
    // ---- inverse, 25-point along y2
    if (tid < 225) {
        f2* p = tile + c_y1 * (25 * kPfaN) + c_x;
        f2 v[25];
#pragma unroll
        for (int q = 0; q < 25; ++q) v[q] = p[q * kPfaN];
        reg_fft25<true>(v);
#pragma unroll
        for (int k = 0; k < 25; ++k) p[k * kPfaN] = v[reg25_out(k)];
    }
    __syncthreads();
    // ---- inverse, 25-point along x2: the thread of row slot r, x1 now holds (C, Cm) at x = (25 x1 + 3 x2) mod 75, x2 = 0..24;
    //      reductions (matcher.py:82, 124-125) from the registers, real surface back to LDS for the sub-pixel fit
    f2 v[25];
    float vmax = -INFINITY, mmax = 0.f;
    if (tid < 225) {
        f2* p = tile + 25 * tid;
#pragma unroll
        for (int q = 0; q < 25; ++q) v[q] = p[q];
        reg_fft25<true>(v);
#pragma unroll
        for (int q = 0; q < 25; ++q) { vmax = fmaxf(vmax, v[q].x); mmax = fmaxf(mmax, fabsf(v[q].y)); }
        if (prm.subpixel) {
            float* pf = reinterpret_cast<float*>(p);
#pragma unroll
            for (int k = 0; k < 25; ++k) pf[2 * k] = v[reg25_out(k)].x;
        }
    }
    {
        float vw = vmax, mw = mmax;
        for (int off = 32; off > 0; off >>= 1) { vw = fmaxf(vw, __shfl_down(vw, off)); mw = fmaxf(mw, __shfl_down(mw, off)); }
        if (lane == 0) { red[2 * wave] = vw; red[2 * wave + 1] = mw; }
    }
    __syncthreads();
    float V = red[0], MM = red[1];
#pragma unroll
    for (int w = 1; w < kPfaThreads / 64; ++w) { V = fmaxf(V, red[2 * w]); MM = fmaxf(MM, red[2 * w + 1]); }
    if (tid < 225 && vmax == V) {
        // first maximal flat index, row-major (matcher.py:82): rare path, one or a few lanes of the workgroup
        const int rs = tid / 3, x1 = tid - 3 * rs;                 // row slot = 25 y1 + y2
        const int y1 = (rs >= 25) + (rs >= 50), y2r = rs - 25 * y1;
        int y = 25 * y1 + 3 * y2r; y -= 75 * (y >= 75) ; y -= 75 * (y >= 75);
        int best = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < 25; ++k) {
            if (v[reg25_out(k)].x == V) {
                int xx = 25 * x1 + 3 * k; xx -= 75 * (xx >= 75);
                best = min(best, y * kPfaN + xx);
            }
        }
        atomicMin(&red_idx, best);
    }
    __syncthreads();
    if (tid != 0) return;
    {
#pragma clang fp contract(off)
        int iv = red_idx;
        if (iv == 0x7fffffff) iv = 0;
        const int py = iv / kPfaN, px = iv - py * kPfaN;
        double ddx = (double)px, ddy = (double)py;
        if (prm.subpixel) {                            // matcher.py:84-106
            const float* tf = reinterpret_cast<const float*>(tile);
            float ct[9];
            for (int j = 0; j < 9; ++j) {
                const int yy = (py + (j / 3 - 1) + kPfaN) % kPfaN, xx = (px + (j % 3 - 1) + kPfaN) % kPfaN;
                ct[j] = tf[2 * (pfa_digits(yy) * kPfaN + pfa_digits(xx))];
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
        ddy -= rint(ddy / (double)kPfaN) * (double)kPfaN;
        ddx -= rint(ddx / (double)kPfaN) * (double)kPfaN;
        prm.dx[n] = ddx; prm.dy[n] = ddy;
        float cf = 1.f;
        if (prm.conf_mode == FB_CONF_MIRROR) {
            cf = 0.f;
            if (V > 0.f) cf = 1.f - MM / V;
            cf = fminf(fmaxf(cf, 0.f), 1.f);
        }
        prm.conf[n] = cf;
    }
}

}  // namespace

int fb_ncc_pfa_supported(int Fh, int Fw, int conf_mode) {
    static const bool off = [] { const char* e = getenv("FEABAS_HIP_NO_PFA"); return e && atoi(e) != 0; }();
    return !off && Fh == kPfaN && Fw == kPfaN && conf_mode != FB_CONF_STD;
}

int fb_ncc_pfa_launch(fb_ctx* ctx, const float* img0, const float* img1, int N, int H0, int W0, int H1, int W1, const int* blk,
                      int IH0, int IW0, int IH1, int IW1, int subpixel, int conf_mode, double* dx, double* dy, float* conf,
                      const double* aff1) {
    if (H0 > kPfaN || W0 > kPfaN || H1 > kPfaN || W1 > kPfaN || H0 < 1 || W0 < 1 || H1 < 1 || W1 < 1)
        return fb_fail(ctx, FB_ERR_ARG, "ncc_pfa75: blocks of %dx%d / %dx%d do not fit a 75x75 transform", H0, W0, H1, W1);
    PfaParams p;
    p.N = N; p.H0 = H0; p.W0 = W0; p.H1 = H1; p.W1 = W1;
    p.subpixel = subpixel; p.conf_mode = conf_mode;
    p.img0 = img0; p.img1 = img1; p.blk = blk; p.aff = blk ? aff1 : nullptr;
    p.IH0 = IH0; p.IW0 = IW0; p.IH1 = IH1; p.IW1 = IW1;
    p.dx = dx; p.dy = dy; p.conf = conf;
    FB_PROF_B(ctx, "ncc_small_fused", (double)N * (4.0 * ((double)H0 * W0 + (double)H1 * W1) + 20.0));
    if (p.aff) hipLaunchKernelGGL(ncc_pfa75<true>, dim3(N), dim3(kPfaThreads), 0, ctx->stream, p);
    else hipLaunchKernelGGL(ncc_pfa75<false>, dim3(N), dim3(kPfaThreads), 0, ctx->stream, p);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}
