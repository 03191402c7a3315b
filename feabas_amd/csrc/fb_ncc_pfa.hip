// NCC path, on-chip class, register-resident form for the transforms of the fine blocks of a 4k tile pair: blocks of
// 70..75 px on a side (matcher.py:243-251, common.divide_bbox), pad = False, so each axis of the FFT is 72 or 75 long
// (matcher.py:59-62, 701-705).  Replaces matcher.xcorr_fft (feabas/matcher.py:22-135) for those four shapes; every other
// on-chip shape stays on ncc_small_fused (fb_ncc_small.hip).
//
// The cross-correlation of one block pair is, as there, ONE packed complex 2-D transform forward (z = img0 + i img1),
// a pointwise step, and ONE packed complex 2-D transform back (real part = C, imaginary part = the mirror surface of
// matcher.py:113-128).  What is different is how the transform is cut:
//
//   * each axis is 3 x N2 (N2 = 25 or 24).  75 = 3 x 25 has coprime factors: a 3 x 25 two-dimensional DFT WITHOUT twiddles
//     between the factors (Good-Thomas; sample n at digits (n mod 3, 17 n mod 25), frequency k at (k mod 3, k mod 25)).
//     72 = 3 x 24 is a Cooley-Tukey split (sample n at (n mod 3, n / 3), frequency k at (k / 24, k mod 24)) whose
//     twiddles w72^(n1 k2) are applied where the 3-point transforms are.  The tile in LDS is the 4-D array
//     [y1][y2][x1][x2] with physical extents [3][25][3][25] for every shape (row pitch 75 complex: the strides 25 and 75
//     keep all the passes off each other's banks; a 24-long digit leaves its last slot unused), and the whole 2-D
//     transform is separable passes over it: N2-point along x2, N2-point along y2, and the two 3-point ones as one
//     3 x 3 block.
//   * a thread owns one N2-point transform (or one pair of 3 x 3 blocks) IN REGISTERS: every index, every twiddle of a
//     long pass is a compile-time constant, every LDS operand is `per-thread base + immediate offset`.  No index
//     arithmetic, no digit-reversal table; five workgroup barriers instead of fifteen.
//   * the pointwise step never forms the two spectra.  With Zk = Z(k), Zn = Z(-k): F0 = (Zk + conj Zn) / 2,
//     F1 = -i (Zk - conj Zn) / 2, and the packed inverse input  conj(F0) F1 + i F0 F1  equals
//     (Re F0 - Im F0) (1 + i) F1 -- a real scalar times a rotated F1: six packed instructions per frequency pair.
//     Frequencies k and -k sit in the 3 x 3 blocks of (y2, x2) and (-y2, -x2); one thread takes both blocks, so the
//     forward 3 x 3 step, the pointwise step and the inverse 3 x 3 step are one trip through LDS.  On a 72-long axis the
//     block of -k2 takes the conjugate twiddles of the block of k2 (its frequency is -k2, not 24 - k2), which keeps the
//     pairing inside the blocks the same as on a 75-long axis: element m with element (3 - m) mod 3.
//   * peak / mirror maximum are reduced from the registers of the last pass; only the real surface goes back to LDS
//     (for the 3 x 3 sub-pixel neighbourhood, matcher.py:84-106).
//
// 256 threads (<= 225 long transforms per pass, <= 313 block pairs in the 3 x 3 pass), 45 KB of LDS: three workgroups
// per CU.  The rows of a patch are read through a buffer descriptor whose range check IS the zero padding.
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
constexpr int kPfaThreads = FB_PFA_THREADS;
constexpr int kPfaP = 75;                         // row pitch of the tile = 3 x 25 slots, whatever the shape
constexpr int kPfaSlots = kPfaP * kPfaP;          // complex slots of the tile

// one axis of the transform: N = 3 x N2
template <int N> struct PfaAxis;
template <> struct PfaAxis<75> {
    static constexpr int N2 = 25;
    static constexpr bool kTwiddle = false;                                   // Good-Thomas: coprime factors
    static __device__ __forceinline__ int slot(int n) {                       // 25 (n mod 3) + (17 n mod 25), n < 75
        const int q3 = (n * 171) >> 9, m = 17 * n;
        return 25 * (n - 3 * q3) + (m - 25 * ((m * 1311) >> 15));
    }
    static __device__ __forceinline__ int pos(int n1, int n2) { int n = 25 * n1 + 3 * n2; return n - 75 * (n >= 75); }
};
template <> struct PfaAxis<72> {
    static constexpr int N2 = 24;
    static constexpr bool kTwiddle = true;                                    // Cooley-Tukey: w72^(n1 k2) between the factors
    static __device__ __forceinline__ int slot(int n) { const int q3 = (n * 171) >> 9; return 25 * (n - 3 * q3) + q3; }     // n < 75
    static __device__ __forceinline__ int pos(int n1, int n2) { return 3 * n2 + n1; }
};
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
    int per_xcd;                        // > 0: blocks are dealt to the XCDs in contiguous runs of this length (grid = 8 per_xcd)
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

// 24-point DFT in registers, 8 x 3 Cooley-Tukey (n = 3 a + b, k = ka + 8 kb).  In: v[n] natural order.  Out: frequency
// ka + 8 kb in v[3 ka + kb].
template <bool INV, int KA = 1>
struct Reg24Tw {            // twiddles w24^(b ka), b = 1, 2, of the outputs ka of the three 8-point transforms
    static __device__ __forceinline__ void run(f2* v) {
        v[3 * KA + 1] = pk_mulw<KA, 24, INV>(v[3 * KA + 1]);
        v[3 * KA + 2] = pk_mulw<2 * KA, 24, INV>(v[3 * KA + 2]);
        if constexpr (KA < 7) Reg24Tw<INV, KA + 1>::run(v);
    }
};
template <bool INV>
__device__ __forceinline__ void reg_fft24(f2* v) {
    // Y[b][ka] = sum_a x[3 a + b] w8^(a ka), left in v[3 ka + b]
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        f2 t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = v[3 * q + b];
        pk_dft8<INV>(t);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[3 * q + b] = t[q];
    }
    Reg24Tw<INV>::run(v);
    // X[ka + 8 kb] = sum_b Y[b][ka] w3^(b kb), left in v[3 ka + kb]
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) pk_dft3<INV>(v + 3 * ka);
}
constexpr int reg24_out(int k) { return 3 * (k % 8) + k / 8; }

template <int N2, bool INV>
__device__ __forceinline__ void reg_fft(f2* v) {
    if constexpr (N2 == 25) reg_fft25<INV>(v);
    else reg_fft24<INV>(v);
}
template <int N2>
constexpr int reg_out(int k) { return N2 == 25 ? reg25_out(k) : reg24_out(k); }

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

// AFF: image 1 is gathered through a per-block affine map (prm.aff)
template <int NY, int NX, bool AFF>
__global__ __launch_bounds__(kPfaThreads) __attribute__((amdgpu_waves_per_eu(FB_PFA_WPE, FB_PFA_WPE))) void ncc_pfa(const PfaParams prm) {
    using AY = PfaAxis<NY>;
    using AX = PfaAxis<NX>;
    constexpr int N2Y = AY::N2, N2X = AX::N2;
    __shared__ __attribute__((aligned(16))) f2 tile[kPfaSlots];
    __shared__ float red[16];
    __shared__ int red_idx;
    __shared__ f2 tw72[AY::kTwiddle || AX::kTwiddle ? 48 : 1];       // exp(-2 pi i m / 72), m < 48 (72-long axes only)
    // Workgroups go to the XCDs round-robin (blockIdx % 8).  The blocks of a pair follow each other in z-order, and a crop row
    // of ~290 bytes shares its first and last 128-byte line with the crops beside it: with blockIdx as block index the
    // neighbours sit on eight different L2s and each fetches those lines again (2.0 x the algorithmic bytes on the counters);
    // with one contiguous run of blocks per XCD they meet in one L2.
    int n = blockIdx.x;
    if (prm.per_xcd > 0) {
        n = (n & 7) * prm.per_xcd + (n >> 3);
        if (n >= prm.N) return;
    }
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
    // ---- packed load z = img0 + i img1, zero padded (matcher.py:63-64): thread = column x, rows yg, yg + 3, ... (N2Y of them).
    //      The rows of the block that exist in the image are the records of a buffer descriptor (wave-uniform), so a row
    //      outside the block or the image is dropped by the range check of the load and reads as zero: one address add per
    //      element, nothing else; a column outside them gets an offset that is out of range on every row.
    constexpr int kYG = 3, kRowsPer = N2Y;
    static_assert(kPfaThreads >= kYG * NX, "one thread per (column, row group)");
    const int yg = (tid >= NX) + (tid >= 2 * NX) + (tid >= 3 * NX), x = tid - NX * yg;
    const bool act = tid < kYG * NX;
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
    if ((AY::kTwiddle || AX::kTwiddle) && tid < 48) {
        float sn, cs;
        sincospif(-(float)tid * (1.0f / 36.0f), &sn, &cs);
        tw72[tid] = (f2){cs, sn};
    }
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
            ddy -= rint(ddy / (double)NY) * (double)NY;
            ddx -= rint(ddx / (double)NX) * (double)NX;
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
        if (act) {
            f2* q = tile + 25 * yg * kPfaP + AX::slot(x);
            if constexpr (NY == 75) {
                // row yg + 3 j sits in row slot 25 yg + (17 yg + j) mod 25 (51 = 1 mod 25): consecutive slots with one wrap
                const int c = 17 * yg - 25 * (yg == 2);
                q += c * kPfaP;
#pragma unroll
                for (int j = 0; j < kRowsPer; ++j) (j >= 25 - c ? q - 25 * kPfaP : q)[j * kPfaP] = (f2){a[j], b[j]};
            } else {
                // row yg + 3 j sits in row slot 25 yg + j
#pragma unroll
                for (int j = 0; j < kRowsPer; ++j) q[j * kPfaP] = (f2){a[j], b[j]};
            }
        }
    }
    __syncthreads();

    // ---- forward, N2X-point along x2: item (row slot, x1) = N2X consecutive slots
    const int r_row = tid / 3, r_y1 = r_row / N2Y;                               // (y1, y2, x1) of item tid < 9 N2Y
    f2* const p_x = tile + (r_row + (25 - N2Y) * r_y1) * kPfaP + 25 * (tid - 3 * r_row);
    if (tid < 9 * N2Y) {
        f2 v[N2X];
#pragma unroll
        for (int q = 0; q < N2X; ++q) v[q] = p_x[q];
        reg_fft<N2X, false>(v);
#pragma unroll
        for (int k = 0; k < N2X; ++k) p_x[k] = v[reg_out<N2X>(k)];
    }
    __syncthreads();
    // ---- forward, N2Y-point along y2: item (y1, column slot), elements one row apart
    const int c_y1 = (tid >= 3 * N2X) + (tid >= 6 * N2X), c_c = tid - 3 * N2X * c_y1, c_x1 = c_c / N2X;      // (y1, x1, x2) of item tid < 9 N2X
    f2* const p_y = tile + c_y1 * (25 * kPfaP) + c_c + (25 - N2X) * c_x1;
    if (tid < 9 * N2X) {
        f2 v[N2Y];
#pragma unroll
        for (int q = 0; q < N2Y; ++q) v[q] = p_y[q * kPfaP];
        reg_fft<N2Y, false>(v);
#pragma unroll
        for (int k = 0; k < N2Y; ++k) p_y[k * kPfaP] = v[reg_out<N2Y>(k)];
    }
    __syncthreads();
    // ---- 3 x 3 forward, pointwise products (matcher.py:65, 114), 3 x 3 inverse: item = the blocks of (y2, x2) and (-y2, -x2).
    //      Representatives: rows 1 .. (N2Y - 1) / 2 whole, then the first half of the self-negating rows (0, and N2Y / 2 if even)
    constexpr int kHy = (N2Y - 1) / 2, kHx = N2X / 2 + 1, kItems = kHy * N2X + (1 + (N2Y % 2 == 0)) * kHx;
#pragma unroll 1
    for (int it = tid; it < kItems; it += kPfaThreads) {
        int y2, x2;
        if (it < kHy * N2X) { const int q = it / N2X; y2 = 1 + q; x2 = it - N2X * q; }
        else { const int r = it - kHy * N2X, q = r >= kHx; y2 = q ? N2Y / 2 : 0; x2 = r - kHx * q; }
        const int y2n = y2 ? N2Y - y2 : 0, x2n = x2 ? N2X - x2 : 0;
        f2* pa = tile + y2 * kPfaP + x2;
        f2* pb = tile + y2n * kPfaP + x2n;
        f2 za[9], zb[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) { za[3 * r + c] = pa[r * (25 * kPfaP) + c * 25]; zb[3 * r + c] = pb[r * (25 * kPfaP) + c * 25]; }
        // twiddles of the 72-long axes: element (r, c) of the block of k2 by w72^(r k2y + c k2x), of the block of -k2 by its conjugate
        f2 tw[9];
        if constexpr (AY::kTwiddle || AX::kTwiddle) {
            f2 ty[3], tx[3];
            if constexpr (AY::kTwiddle) { ty[1] = tw72[y2]; ty[2] = tw72[2 * y2]; }
            if constexpr (AX::kTwiddle) { tx[1] = tw72[x2]; tx[2] = tw72[2 * x2]; }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const bool hy = AY::kTwiddle && r > 0, hx = AX::kTwiddle && c > 0;
                    if (hy && hx) tw[3 * r + c] = pk_cmul(ty[r], tx[c]);
                    else if (hy) tw[3 * r + c] = ty[r];
                    else if (hx) tw[3 * r + c] = tx[c];
                    if (hy || hx) { za[3 * r + c] = pk_cmul(za[3 * r + c], tw[3 * r + c]); zb[3 * r + c] = pk_cmulc(zb[3 * r + c], tw[3 * r + c]); }
                }
        }
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
        if constexpr (AY::kTwiddle || AX::kTwiddle) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    if ((AY::kTwiddle && r > 0) || (AX::kTwiddle && c > 0)) { za[3 * r + c] = pk_cmulc(za[3 * r + c], tw[3 * r + c]); zb[3 * r + c] = pk_cmul(zb[3 * r + c], tw[3 * r + c]); }
        }
        // a self-negating block is the same block twice: both copies hold the same values
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) { pa[r * (25 * kPfaP) + c * 25] = za[3 * r + c]; pb[r * (25 * kPfaP) + c * 25] = zb[3 * r + c]; }
    }
    __syncthreads();
    // ---- inverse, N2Y-point along y2
    if (tid < 9 * N2X) {
        f2 v[N2Y];
#pragma unroll
        for (int q = 0; q < N2Y; ++q) v[q] = p_y[q * kPfaP];
        reg_fft<N2Y, true>(v);
#pragma unroll
        for (int k = 0; k < N2Y; ++k) p_y[k * kPfaP] = v[reg_out<N2Y>(k)];
    }
    __syncthreads();
    // ---- inverse, N2X-point along x2: the thread of (row slot, x1) now holds (C, Cm) at x = AX::pos(x1, x2), x2 < N2X;
    //      reductions (matcher.py:82, 124-125) from the registers, real surface back to LDS for the sub-pixel fit
    f2 v[N2X];
    float vmax = -INFINITY, mmax = 0.f;
    if (tid < 9 * N2Y) {
#pragma unroll
        for (int q = 0; q < N2X; ++q) v[q] = p_x[q];
        reg_fft<N2X, true>(v);
#pragma unroll
        for (int q = 0; q < N2X; ++q) { vmax = fmaxf(vmax, v[q].x); mmax = fmaxf(mmax, fabsf(v[q].y)); }
        if (prm.subpixel) {
            float* pf = reinterpret_cast<float*>(p_x);
#pragma unroll
            for (int k = 0; k < N2X; ++k) pf[2 * k] = v[reg_out<N2X>(k)].x;
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
    if (tid < 9 * N2Y && vmax == V) {
        // first maximal flat index, row-major (matcher.py:82): rare path, one or a few lanes of the workgroup
        const int x1 = tid - 3 * r_row;
        const int y = AY::pos(r_y1, r_row - N2Y * r_y1);
        int best = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < N2X; ++k)
            if (v[reg_out<N2X>(k)].x == V) best = min(best, y * NX + AX::pos(x1, k));
        atomicMin(&red_idx, best);
    }
    __syncthreads();
    if (tid != 0) return;
    {
#pragma clang fp contract(off)
        int iv = red_idx;
        if (iv == 0x7fffffff) iv = 0;
        const int py = iv / NX, px = iv - py * NX;
        double ddx = (double)px, ddy = (double)py;
        if (prm.subpixel) {                            // matcher.py:84-106
            const float* tf = reinterpret_cast<const float*>(tile);
            float ct[9];
            for (int j = 0; j < 9; ++j) {
                const int yy = (py + (j / 3 - 1) + NY) % NY, xx = (px + (j % 3 - 1) + NX) % NX;
                ct[j] = tf[2 * (AY::slot(yy) * kPfaP + AX::slot(xx))];
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
        ddy -= rint(ddy / (double)NY) * (double)NY;
        ddx -= rint(ddx / (double)NX) * (double)NX;
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

template <int NY, int NX>
void pfa_launch(const PfaParams& p, hipStream_t st) {
    const int grid = p.per_xcd > 0 ? 8 * p.per_xcd : p.N;
    if (p.aff) hipLaunchKernelGGL((ncc_pfa<NY, NX, true>), dim3(grid), dim3(kPfaThreads), 0, st, p);
    else hipLaunchKernelGGL((ncc_pfa<NY, NX, false>), dim3(grid), dim3(kPfaThreads), 0, st, p);
}

}  // namespace

int fb_ncc_pfa_supported(int Fh, int Fw, int conf_mode) {
    static const bool off = [] { const char* e = getenv("FEABAS_HIP_NO_PFA"); return e && atoi(e) != 0; }();
    return !off && (Fh == 72 || Fh == 75) && (Fw == 72 || Fw == 75) && conf_mode != FB_CONF_STD;
}

int fb_ncc_pfa_launch(fb_ctx* ctx, const float* img0, const float* img1, int N, int H0, int W0, int H1, int W1, const int* blk,
                      int IH0, int IW0, int IH1, int IW1, int Fh, int Fw, int subpixel, int conf_mode, double* dx, double* dy, float* conf,
                      const double* aff1) {
    if (!fb_ncc_pfa_supported(Fh, Fw, conf_mode)) return fb_fail(ctx, FB_ERR_ARG, "ncc_pfa: no %dx%d transform", Fh, Fw);
    if (H0 > Fh || W0 > Fw || H1 > Fh || W1 > Fw || H0 < 1 || W0 < 1 || H1 < 1 || W1 < 1)
        return fb_fail(ctx, FB_ERR_ARG, "ncc_pfa: blocks of %dx%d / %dx%d do not fit a %dx%d transform", H0, W0, H1, W1, Fh, Fw);
    if (N <= 0) return FB_OK;
    PfaParams p;
    p.N = N; p.H0 = H0; p.W0 = W0; p.H1 = H1; p.W1 = W1;
    p.subpixel = subpixel; p.conf_mode = conf_mode;
    p.img0 = img0; p.img1 = img1; p.blk = blk; p.aff = blk ? aff1 : nullptr;
    p.IH0 = IH0; p.IW0 = IW0; p.IH1 = IH1; p.IW1 = IW1;
    p.dx = dx; p.dy = dy; p.conf = conf;
    static const bool xcd_runs = [] { const char* e = getenv("FEABAS_HIP_PFA_XCD"); return !e || atoi(e) != 0; }();
    p.per_xcd = (xcd_runs && blk && N >= 64) ? (N + 7) / 8 : 0;
    FB_PROF_B(ctx, "ncc_small_fused", (double)N * (4.0 * ((double)H0 * W0 + (double)H1 * W1) + 20.0));
    if (Fh == 75 && Fw == 75) pfa_launch<75, 75>(p, ctx->stream);
    else if (Fh == 75) pfa_launch<75, 72>(p, ctx->stream);
    else if (Fw == 75) pfa_launch<72, 75>(p, ctx->stream);
    else pfa_launch<72, 72>(p, ctx->stream);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}
