// Band-pass pre-filter of the matcher: common.masked_dog_filter (feabas/common.py:353-377)
// and the 2x area downsample in front of the coarse match (matcher.py:255-256).
//
// DoG = G(img) - G(G(img)), each G = scipy.ndimage.gaussian_filter1d along W then H with
// mode='nearest', truncate 4 sigma.  scipy keeps float32 between passes and accumulates
// each output sample in double, centre tap first, then tap pairs from the outermost in;
// the kernels below reproduce that order.  One workgroup filters a 64x64 output tile
// through all four passes inside LDS (two ping-pong buffers): HBM traffic is one read of
// the (haloed) input and one float32 write per pixel.
#include "fb_common.h"

#include <algorithm>
#include <cmath>
#include <cfloat>
#include <vector>

namespace {

constexpr int kMaxRadius = 40;
struct Taps { double w[kMaxRadius + 1]; };      // w[0] = centre, w[k] = tap at +-k (kernel argument)

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <typename T>
__device__ __forceinline__ float load_px(const T* p, size_t i);
template <>
__device__ __forceinline__ float load_px<uint8_t>(const uint8_t* p, size_t i) { return (float)p[i]; }
template <>
__device__ __forceinline__ float load_px<float>(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float load_px(const uint8_t* p, uint32_t i) { return (float)p[i]; }
__device__ __forceinline__ float load_px(const float* p, uint32_t i) { return p[i]; }

// 1-D correlation at LDS position `base` with element stride `st` (symmetric taps)
__device__ __forceinline__ float fir_sym(const float* s, int base, int st, int r, const Taps& t) {
    double acc = (double)s[base] * t.w[0];
    for (int k = r; k >= 1; --k)
        acc += ((double)s[base - k * st] + (double)s[base + k * st]) * t.w[k];
    return (float)acc;
}

// NPASS = 2: out = G(in).  NPASS = 4: out = G(in) - G(G(in)), optional masked-halo epilogue.
// A value stored at tile-local (ty,tx) always is the value of the stage's image at the
// CLAMPED global coordinate, which is exactly scipy's 'nearest' extension of that stage.
template <typename T, int NPASS>
__global__ void dog_tile(const T* __restrict__ img, float* __restrict__ out, const float* __restrict__ halo,
                         int H, int W, int r, int TY, int TX, int signed_out, float in_scale, const uint8_t* __restrict__ mask, const Taps taps,
                         size_t per_image, const int* __restrict__ ids = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = ids ? ids[blockIdx.z] : blockIdx.z;       // ids: the images of the stack this launch works on
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int hal = (NPASS == 4) ? 2 * r : r;          // input halo
    const int AW = TX + 2 * hal, AH = TY + 2 * hal;     // stage-0 tile
    float* bufA = smem;                                 // AH x AW, later G0: (TY+2r) x (TX+2r)
    float* bufB = smem + AH * AW;                       // AH x (AW-2r), later (TY+2r|TY) x TX
    const T* src = img + (size_t)n * H * W;
    const int tid = threadIdx.x, nt = blockDim.x;
    // per_image = H W: every image has its own mask / halo plane (the N x H x W masks of MeshRenderer.crop_multiple); 0: one shared plane
    if (mask) mask += (size_t)n * per_image;
    if (halo) halo += (size_t)n * per_image;

    // stage 0: load (clamped) input tile.  With a mask, the input is ptp*(mask==0) (common.py:369)
    for (int i = tid; i < AH * AW; i += nt) {
        const int ty = i / AW, tx = i - ty * AW;
        const int gy = clampi(y0 - hal + ty, 0, H - 1), gx = clampi(x0 - hal + tx, 0, W - 1);
        float v;
        if (mask) v = mask[(size_t)gy * W + gx] ? 0.f : in_scale;
        else v = load_px<T>(src, (size_t)gy * W + gx);
        bufA[i] = v;
    }
    __syncthreads();
    // stage 1: rows.  B (AH x BW), BW = AW - 2r; B(ty,tx) <- global x = x0-(hal-r)+tx
    const int BW = AW - 2 * r;
    for (int i = tid; i < AH * BW; i += nt) {
        const int ty = i / BW, tx = i - ty * BW;
        const int gx = clampi(x0 - (hal - r) + tx, 0, W - 1);
        bufB[i] = fir_sym(bufA, ty * AW + (gx - (x0 - hal)), 1, r, taps);
    }
    __syncthreads();
    // stage 2: columns.  G0 (GH x BW), GH = AH - 2r, into bufA
    const int GH = AH - 2 * r;
    for (int i = tid; i < GH * BW; i += nt) {
        const int ty = i / BW, tx = i - ty * BW;
        const int gy = clampi(y0 - (hal - r) + ty, 0, H - 1);
        bufA[i] = fir_sym(bufB, (gy - (y0 - hal)) * BW + tx, BW, r, taps);
    }
    __syncthreads();
    if (NPASS == 2) {
        for (int i = tid; i < TY * TX; i += nt) {
            const int ty = i / TX, tx = i - ty * TX;
            const int gy = y0 + ty, gx = x0 + tx;
            if (gy < H && gx < W) out[((size_t)n * H + gy) * W + gx] = bufA[ty * BW + tx] * 2.0f;   // (sc^2/s0^2) = 2, common.py:371
        }
        return;
    }
    // stage 3: rows of G0 -> D (GH x TX) in bufB
    for (int i = tid; i < GH * TX; i += nt) {
        const int ty = i / TX, tx = i - ty * TX;
        const int gx = clampi(x0 + tx, 0, W - 1);
        bufB[i] = fir_sym(bufA, ty * BW + (gx - (x0 - r)), 1, r, taps);
    }
    __syncthreads();
    // stage 4: columns of D -> G1; out = G0 - G1
    for (int i = tid; i < TY * TX; i += nt) {
        const int ty = i / TX, tx = i - ty * TX;
        const int gy = y0 + ty, gx = x0 + tx;
        if (gy >= H || gx >= W) continue;
        const float g1 = fir_sym(bufB, (gy - (y0 - r)) * TX + tx, TX, r, taps);
        const float g0 = bufA[(ty + r) * BW + (tx + r)];
        float v = g0 - g1;
        if (halo) {                                     // common.py:372-374
            const float a = fmaxf(fabsf(v) - halo[(size_t)gy * W + gx], 0.f);
            v = v > 0.f ? a : (v < 0.f ? -a : 0.f * a);
        }
        if (!signed_out) v = fabsf(v);
        out[((size_t)n * H + gy) * W + gx] = v;
    }
}

// global min / max of the whole input (np.ptp, common.py:369)
template <typename T>
__global__ void minmax_kernel(const T* __restrict__ img, size_t total, float* __restrict__ mm) {
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float v = load_px<T>(img, i);
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_down(lo, off));
        hi = fmaxf(hi, __shfl_down(hi, off));
    }
    if ((threadIdx.x & 63) == 0) {
        // float atomics through the integer ordering trick are avoided: one slot per wave
        const size_t slot = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        mm[2 * slot] = lo; mm[2 * slot + 1] = hi;
    }
}

// non-zero bytes of m[0 .. total): 16 bytes per lane and trip where the pointer allows it, one atomic per wave
__global__ __launch_bounds__(256) void count_nonzero_kernel(const uint8_t* __restrict__ m, size_t total, unsigned long long* __restrict__ count) {
    unsigned long long c = 0;
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(m) & 15)) & 15;
    const size_t h = head < total ? head : total, nv = (total - h) / 16;
    const uint4* v = reinterpret_cast<const uint4*>(m + h);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 q = v[i];
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // one bit per non-zero byte: fold every byte onto its top bit
            unsigned t = w[k];
            t |= t >> 4; t |= t >> 2; t |= t >> 1;                     // bit 0 of each byte = OR of the byte's bits (low bits of t)
            c += __popc(t & 0x01010101u);
        }
    }
    if (blockIdx.x == 0) {
        for (size_t i = threadIdx.x; i < h; i += blockDim.x) c += m[i] != 0;
        for (size_t i = h + 16 * nv + threadIdx.x; i < total; i += blockDim.x) c += m[i] != 0;
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// is any byte of the mask zero?  16 bytes per load on the aligned body ((x - 0x01..01) & ~x & 0x80..80 marks the zero bytes of
// a word), single bytes on the unaligned head and the tail
__global__ void any_zero_kernel(const uint8_t* __restrict__ m, size_t total, int* flag) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
    const size_t head = min(total, (size_t)((16 - (reinterpret_cast<uintptr_t>(m) & 15)) & 15));
    const size_t nvec = (total - head) / 16;
    const uint4* v = reinterpret_cast<const uint4*>(m + head);
    bool hit = false;
    for (size_t i = gid; i < nvec && !hit; i += gsz) {
        const uint4 q = v[i];
        const uint32_t z = ((q.x - 0x01010101u) & ~q.x) | ((q.y - 0x01010101u) & ~q.y) | ((q.z - 0x01010101u) & ~q.z) | ((q.w - 0x01010101u) & ~q.w);
        hit = (z & 0x80808080u) != 0;
    }
    for (size_t i = gid; i < head && !hit; i += gsz) hit = m[i] == 0;
    for (size_t i = head + 16 * nvec + gid; i < total && !hit; i += gsz) hit = m[i] == 0;
    if (hit) *flag = 1;
}

// per image of a stack of masks: does it hold a zero?  (one workgroup per image)
__global__ __launch_bounds__(256) void any_zero_each_kernel(const uint8_t* __restrict__ m, size_t per, int* __restrict__ flags) {
    const uint8_t* p = m + (size_t)blockIdx.x * per;
    bool hit = false;
    for (size_t i = threadIdx.x; i < per && !hit; i += 256) hit = p[i] == 0;
    __shared__ int any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    if (hit) any = 1;
    __syncthreads();
    if (threadIdx.x == 0) flags[blockIdx.x] = any;
}

// one thread = one output pixel; maps are float32 relative to the integer origin of the sub-image the reference
// hands to cv2.remap; 1/32-px quantisation and float32 table weights as in fb_sample_affine (fb_common.h)
__global__ void remap_kernel(const float* __restrict__ imgs, int IH, int IW, int N, const int* __restrict__ img_id, int h, int w,
                             const float* __restrict__ map_x, const float* __restrict__ map_y, const uint8_t* __restrict__ mask,
                             const int* __restrict__ origin, float* __restrict__ out) {
#pragma clang fp contract(off)
    const size_t per = (size_t)h * w, total = (size_t)N * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / per);
        if (mask && !mask[i]) { out[i] = 0.f; continue; }
        const float* img = imgs + (size_t)img_id[n] * IH * IW;
        const int sx = (int)rintf(map_x[i] * 32.0f), sy = (int)rintf(map_y[i] * 32.0f);
        const int ix = (sx >> 5) + origin[2 * n], iy = (sy >> 5) + origin[2 * n + 1];
        const float ax = (float)(sx & 31) * (1.0f / 32.0f), ay = (float)(sy & 31) * (1.0f / 32.0f);
        const float w00 = (1.0f - ay) * (1.0f - ax), w01 = (1.0f - ay) * ax, w10 = ay * (1.0f - ax), w11 = ay * ax;
        const bool x0ok = ix >= 0 && ix < IW, x1ok = ix + 1 >= 0 && ix + 1 < IW, y0ok = iy >= 0 && iy < IH, y1ok = iy + 1 >= 0 && iy + 1 < IH;
        const int cx0 = min(max(ix, 0), IW - 1), cx1 = min(max(ix + 1, 0), IW - 1), cy0 = min(max(iy, 0), IH - 1), cy1 = min(max(iy + 1, 0), IH - 1);
        const float v00 = img[(size_t)cy0 * IW + cx0], v01 = img[(size_t)cy0 * IW + cx1];
        const float v10 = img[(size_t)cy1 * IW + cx0], v11 = img[(size_t)cy1 * IW + cx1];
        out[i] = ((((y0ok && x0ok) ? v00 : 0.f) * w00 + ((y0ok && x1ok) ? v01 : 0.f) * w01) + ((y1ok && x0ok) ? v10 : 0.f) * w10) + ((y1ok && x1ok) ? v11 : 0.f) * w11;
    }
}

// cv2.resize(fx = fy = 0.5, INTER_AREA) on its integer-scale path (resizeAreaFast): output size cvRound(n / 2) per axis
// (round half to even); a full 2x2 cell is (sum + 2) >> 2; a cell cut by the image edge averages the pixels that exist,
// float(sum) / count rounded half to even
__host__ __device__ inline int half_size(int n) { return (n & 1) ? (((n >> 1) & 1) ? (n >> 1) + 1 : (n >> 1)) : (n >> 1); }

// sizes (nullable) [N][2] = {h, w} of image n inside its H x W slot (strips of unequal size share a padded stack): the
// output slot is half_size(H) x half_size(W), image n fills its half_size(h) x half_size(w) corner, the rest is zero
__global__ void area_down2_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int N, int H, int W, const int* __restrict__ sizes) {
    const int Ho = half_size(H), Wo = half_size(W);
    const size_t total = (size_t)N * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / ((size_t)Ho * Wo);
        const int rem = (int)(i - n * (size_t)Ho * Wo);
        const int y = rem / Wo, x = rem - y * Wo;
        const int Hn = sizes ? sizes[2 * n] : H, Wn = sizes ? sizes[2 * n + 1] : W;
        if (y >= half_size(Hn) || x >= half_size(Wn)) { out[i] = 0; continue; }
        const uint8_t* p = in + (n * H + 2 * y) * (size_t)W + 2 * x;
        const bool x1 = 2 * x + 1 < Wn, y1 = 2 * y + 1 < Hn;
        if (x1 && y1) {
            const int s = (int)p[0] + (int)p[1] + (int)p[W] + (int)p[W + 1];
            out[i] = (uint8_t)((s + 2) >> 2);
        } else {
            int s = (int)p[0], c = 1;
            if (x1) { s += (int)p[1]; ++c; }
            if (y1) { s += (int)p[W]; ++c; }
            out[i] = (uint8_t)rintf((float)s / (float)c);
        }
    }
}

// mask &= (lo <= img <= hi): the mask_range of MeshRenderer.crop_multiple (renderer.py:634-637), before the masked DoG
__global__ void mask_range_kernel(const float* __restrict__ img, size_t total, float lo, float hi, uint8_t* __restrict__ mask) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float v = img[i];
        if (!(v >= lo && v <= hi)) mask[i] = 0;
    }
}

int set_taps(fb_ctx* ctx, double sigma, int* radius, Taps* out) {
    const int r = (int)(4.0 * sigma + 0.5);
    if (r < 1 || r > kMaxRadius) return fb_fail(ctx, FB_ERR_ARG, "fb_dog: sigma %.3f gives radius %d outside [1,%d]", sigma, r, kMaxRadius);
    double w[2 * kMaxRadius + 1];
    double sum = 0.0;
    for (int k = -r; k <= r; ++k) {            // scipy.ndimage._gaussian_kernel1d
        w[k + r] = std::exp(-0.5 / (sigma * sigma) * (double)k * (double)k);
        sum += w[k + r];
    }
    for (int k = 0; k <= kMaxRadius; ++k) out->w[k] = k <= r ? w[r + k] / sum : 0.0;
    *radius = r;
    return FB_OK;
}


// ---------------------------------------------------------------------------------------------
// Fast path (unmasked DoG, radius known at compile time): same four passes, but every thread
// produces RUN consecutive outputs along the filter axis from RUN + 2R values held in registers
// (one LDS read per ~0.3 outputs instead of 21), and the taps are applied in float32.  Row passes
// map lanes to rows (pitch/4 odd -> conflict-free ds_read_b128), column passes map lanes to columns.
// The 'nearest' extension is restored after each pass by replicating the image-border column/row.
struct TapsF { float w[kMaxRadius + 1]; };
constexpr int RUN = 8;
constexpr int FT = 64;     // output tile

__host__ __device__ constexpr int pitch_for(int cols) { int p = (cols + 3) / 4 * 4; return (p / 4) % 2 ? p : p + 4; }
__host__ __device__ constexpr int up8(int v) { return (v + RUN - 1) / RUN * RUN; }
__host__ __device__ constexpr int upn(int v, int n) { return (v + n - 1) / n * n; }

template <int R>
struct FastGeom {
    static constexpr int HAL = 2 * R;
    static constexpr int AH = FT + 2 * HAL, AW = FT + 2 * HAL;      // stage-0 tile
    static constexpr int BW = AW - 2 * R, GH = AH - 2 * R;           // after row pass / after column pass
    static constexpr int PA = pitch_for((upn(BW, 12) > up8(BW) ? upn(BW, 12) : up8(BW)) + 2 * R);      // row-pass overread stays inside the row
    static constexpr int PB = pitch_for(BW);
    static constexpr int PD = pitch_for(FT);
    static constexpr int SZ_A = (AH * PA > GH * PB ? AH * PA : GH * PB);
    static constexpr int SZ_B = (AH * PB > GH * PD ? AH * PB : GH * PD);
};

template <int R, int RN>
__device__ __forceinline__ void fir_run(const float (&in)[RN + 2 * R], const TapsF& t, float (&out)[RN]) {
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        float acc = in[j + R] * t.w[0];
#pragma unroll
        for (int k = R; k >= 1; --k) acc = fmaf(in[j + R - k] + in[j + R + k], t.w[k], acc);
        out[j] = acc;
    }
}

// Run lengths per pass.  One pass is one or two sweeps of the 1024 threads over its items, and every wave waits at the
// barrier for the slowest: the items of a pass should fill ONE sweep as evenly as possible.  At the 64 x 64 tile with
// R = 10: pass 1 (104 rows x 84 cols) 7 runs of 12 = 728 items, pass 2 (84 x 84) 12 runs of 7 = 1008 items, pass 3
// (84 x 64) 8 runs of 8 = 672 items, pass 4 (64 x 64) 16 runs of 4 = 1024 items -- 12 + 7 + 8 + 4 = 31 outputs on the
// critical path of a tile instead of 16 + 8 + 8 + 8 = 40 with runs of 8 everywhere (pass 1 needed a second sweep for
// 120 of its 1144 items).  Row passes keep run lengths that are multiples of 4 (16-byte aligned ds_read_b128).
constexpr int RUN1 = 12, RUN2 = 7, RUN3 = 8, RUN4 = 4;

// rows x ncols_out outputs; src/dst are LDS arrays with pitches ps/pd; output col c reads src cols [c, c+2R]
template <int R, int RN>
__device__ __forceinline__ void row_pass(const float* __restrict__ src, int ps, float* __restrict__ dst, int pd, int rows,
                                         int ncols_out, const TapsF& t) {
    static_assert(RN % 4 == 0, "row runs are read and written as float4");
    const int nruns = (ncols_out + RN - 1) / RN;
    for (int item = threadIdx.x; item < rows * nruns; item += blockDim.x) {
        const int run = item / rows, row = item - run * rows;       // lanes walk rows
        float in[RN + 2 * R], o[RN];
        // 16-byte aligned by construction (pitches and run offsets are multiples of 4 floats): ds_read_b128
        const float4* p4 = reinterpret_cast<const float4*>(__builtin_assume_aligned(src, 16)) + ((row * ps + run * RN) >> 2);
        const float* p = src + row * ps + run * RN;
#pragma unroll
        for (int q = 0; q < (RN + 2 * R) / 4; ++q) {
            const float4 v = p4[q];
            in[4 * q] = v.x; in[4 * q + 1] = v.y; in[4 * q + 2] = v.z; in[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int q = (RN + 2 * R) / 4 * 4; q < RN + 2 * R; ++q) in[q] = p[q];
        fir_run<R, RN>(in, t, o);
        float* d = dst + row * pd + run * RN;
        if (run * RN + RN <= ncols_out) {
#pragma unroll
            for (int q = 0; q < RN / 4; ++q) *reinterpret_cast<float4*>(d + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        } else {
#pragma unroll
            for (int j = 0; j < RN; ++j)
                if (run * RN + j < ncols_out) d[j] = o[j];
        }
    }
}

// nrows_out x cols outputs; output row y reads src rows [y, y+2R]
template <int R, int RN>
__device__ __forceinline__ void col_pass(const float* __restrict__ src, int ps, int src_rows, float* __restrict__ dst, int pd,
                                         int nrows_out, int cols, const TapsF& t) {
    const int nruns = (nrows_out + RN - 1) / RN;
    for (int item = threadIdx.x; item < cols * nruns; item += blockDim.x) {
        const int run = item / cols, c = item - run * cols;         // lanes walk columns
        float in[RN + 2 * R], o[RN];
        const float* p = src + c;
#pragma unroll
        for (int q = 0; q < RN + 2 * R; ++q) in[q] = p[min(run * RN + q, src_rows - 1) * ps];
        fir_run<R, RN>(in, t, o);
        float* d = dst + (run * RN) * pd + c;
#pragma unroll
        for (int j = 0; j < RN; ++j)
            if (run * RN + j < nrows_out) d[j * pd] = o[j];
    }
}

template <typename T, int R>
__global__ __launch_bounds__(1024) void dog_fast(const T* __restrict__ img, float* __restrict__ out, int SH, int SW, int signed_out,
                                                const int* __restrict__ sizes, const TapsF taps) {
    using G = FastGeom<R>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* bufA = smem;
    float* bufB = smem + G::SZ_A;
    const int n = blockIdx.z;
    const int x0 = blockIdx.x * FT, y0 = blockIdx.y * FT;
    // image n occupies the H x W corner of its SH x SW slot (sizes == nullptr: the whole slot); the 'nearest' extension
    // and the output are those of the H x W image
    const int H = sizes ? sizes[2 * n] : SH, W = sizes ? sizes[2 * n + 1] : SW;
    if (x0 >= W || y0 >= H) {
        // a tile of the slot outside the image: zero, so that windows cropped past the image border read the fill value
        for (int i = threadIdx.x; i < FT * FT; i += blockDim.x) {
            const int gy = y0 + i / FT, gx = x0 + i % FT;
            if (gy < SH && gx < SW) out[((size_t)n * SH + gy) * SW + gx] = 0.f;
        }
        return;
    }
    const T* src = img + (size_t)n * SH * SW;
    const int tid = threadIdx.x, nt = blockDim.x;
    // stage 0: clamped input tile ('nearest' extension of the image).  Threads = 2 row groups x 128 columns;
    // 8 rows are fetched per trip so that the loads are in flight together (a load -> LDS store per trip
    // exposes the full memory latency every iteration).
    {
        static_assert(G::AW <= 128, "tile wider than the load mapping");
        const int tx = tid & 127, tyo = tid >> 7, nrg = nt >> 7;
        const int gx = clampi(x0 - G::HAL + tx, 0, W - 1);
        constexpr int UB = 8;
        for (int tb = tyo; tb < G::AH; tb += UB * nrg) {
            float val[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                // 32-bit offsets from the (uniform) image base: one image is far below 4 G pixels (checked by the launcher)
                const int ty = min(tb + u * nrg, G::AH - 1);
                val[u] = load_px(src, (uint32_t)(clampi(y0 - G::HAL + ty, 0, H - 1) * SW + gx));
            }
            if (tx < G::AW) {
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int ty = tb + u * nrg;
                    if (ty < G::AH) bufA[ty * G::PA + tx] = val[u];
                }
            }
        }
    }
    __syncthreads();
    // stage 1: rows A -> B (AH x BW); B col tx <-> global x = x0 - R + tx
    row_pass<R, RUN1>(bufA, G::PA, bufB, G::PB, G::AH, G::BW, taps);
    __syncthreads();
    {   // replicate the image-border columns
        const int cl = -(x0 - R), cr = (W - 1) - (x0 - R);          // B columns of global x = 0 and x = W-1
        if (cl > 0 || cr < G::BW - 1) {
            // only the columns outside the image are touched: lanes walk rows
            const int nleft = max(cl, 0), nright = max(G::BW - 1 - cr, 0);
            for (int i = tid; i < G::AH * (nleft + nright); i += nt) {
                const int k = i / G::AH, ty = i - k * G::AH;
                const int tx = k < nleft ? k : cr + 1 + (k - nleft);
                bufB[ty * G::PB + tx] = bufB[ty * G::PB + (k < nleft ? cl : cr)];
            }
            __syncthreads();
        }
    }
    // stage 2: columns B -> G0 (GH x BW) in bufA (pitch PB); G0 row ty <-> global y = y0 - R + ty
    col_pass<R, RUN2>(bufB, G::PB, G::AH, bufA, G::PB, G::GH, G::BW, taps);
    __syncthreads();
    {   // replicate the image-border rows
        const int rt = -(y0 - R), rb = (H - 1) - (y0 - R);
        if (rt > 0 || rb < G::GH - 1) {
            const int ntop = max(rt, 0), nbot = max(G::GH - 1 - rb, 0);
            for (int i = tid; i < G::BW * (ntop + nbot); i += nt) {
                const int k = i / G::BW, tx = i - k * G::BW;
                const int ty = k < ntop ? k : rb + 1 + (k - ntop);
                bufA[ty * G::PB + tx] = bufA[(k < ntop ? rt : rb) * G::PB + tx];
            }
            __syncthreads();
        }
    }
    // stage 3: rows G0 -> D (GH x FT) in bufB (pitch PD); D col tx <-> global x = x0 + tx
    row_pass<R, RUN3>(bufA, G::PB, bufB, G::PD, G::GH, FT, taps);
    __syncthreads();
    // stage 4: columns D -> G1 (FT x FT); out = G0 - G1
    {
        const int nruns = FT / RUN4;
        for (int item = tid; item < FT * nruns; item += nt) {
            const int run = item / FT, c = item - run * FT;
            float in[RUN4 + 2 * R], o[RUN4];
            const float* p = bufB + (run * RUN4) * G::PD + c;
#pragma unroll
            for (int q = 0; q < RUN4 + 2 * R; ++q) in[q] = p[q * G::PD];
            fir_run<R, RUN4>(in, taps, o);
            const int gx = x0 + c;
            if (gx >= SW) continue;
            float* __restrict__ oimg = out + (size_t)n * SH * SW;                 // uniform base, 32-bit offsets below
            const uint32_t off0 = (uint32_t)((y0 + run * RUN4) * SW + gx);
            const int jmax = min(RUN4, SH - (y0 + run * RUN4)), jin = gx < W ? H - (y0 + run * RUN4) : 0;
#pragma unroll
            for (int j = 0; j < RUN4; ++j) {
                if (j >= jmax) break;
                const int ty = run * RUN4 + j;
                float v = bufA[(ty + R) * G::PB + (c + R)] - o[j];
                if (!signed_out) v = fabsf(v);
                oimg[off0 + (uint32_t)(j * SW)] = j < jin ? v : 0.f;                  // slot pixels outside the image: 0
            }
        }
    }
}

template <typename T, int R>
int launch_fast(fb_ctx* ctx, const T* img, float* out, int N, int H, int W, int signed_out, const Taps& taps, const int* sizes) {
    using G = FastGeom<R>;
    TapsF tf;
    for (int k = 0; k <= kMaxRadius; ++k) tf.w[k] = (float)taps.w[k];
    const size_t lds = (size_t)(G::SZ_A + G::SZ_B) * sizeof(float);
    if ((size_t)H * W >= ((size_t)1 << 31)) return fb_fail(ctx, FB_ERR_ARG, "fb_dog: image of %d x %d pixels exceeds the 32-bit offsets of the fast kernel", H, W);
    auto kern = dog_fast<T, R>;
    FB_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(fb_cdiv(W, FT), fb_cdiv(H, FT), N);
    hipLaunchKernelGGL(kern, grid, dim3(1024), lds, ctx->stream, img, out, H, W, signed_out, sizes, tf);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}


// ---------------------------------------------------------------------------------------------
// Streaming form of the fast path.  DoG = G(I) - G(G(I)) = G(I - G(I)) (G is linear, 'nearest' extension included), and
// the two 1-D passes of a G commute, so the four passes run in the order  x, y, (I - .), y, x:
//   A = Gx I        horizontal, out of an LDS staging tile of the input rows
//   B = Gy A        vertical:   one thread per column, the last 2R rows of A live in registers
//   D = I - B       (I delayed by R rows in registers)
//   E = Gy D        vertical, second register window of the same thread
//   out = Gx E      horizontal, out of an LDS staging tile, stored straight to HBM
// A workgroup owns a band of TX output columns and a segment of SY rows and streams down the rows 8 at a time: nothing
// is re-read in y except the 4R warm-up rows of a segment, the x halo costs 1 + 1.5 R / TX in arithmetic (a 64 x 64
// tile with its 2R apron cost 1.55 x), every LDS address is a per-thread base plus an immediate, and the vertical passes
// -- half of the arithmetic -- read each value once.  The 'nearest' extensions: rows / columns of the input outside the
// image are clamped loads (A, B of a virtual column are then the replicas the next stage expects, because a vertical
// pass commutes with column replication); D is replicated explicitly at the top (window initialised with D[0]) and at
// the bottom (D[y > H-1] = D[H-1]).
constexpr int SCH = 8;     // rows per chunk
constexpr int SRN = 8;     // outputs per horizontal run
#ifndef FB_DOG_FIL
#define FB_DOG_FIL 2
#endif
constexpr int FIL = FB_DOG_FIL;   // outputs whose accumulators are interleaved
#ifndef FB_DOG_R4
#define FB_DOG_R4 10
#endif

__host__ __device__ constexpr int spitch(int cols) { int p = (cols + 3) / 4 * 4; return (p / 4) % 2 ? p : p + 4; }     // pitch / 4 odd

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

// Packed-FP32 form of the 1-D pass (v_pk_fma_f32: two multiply-adds per lane and instruction; the scalar v_fma_f32 runs at
// half the FP32 rate of the chip).  Values sit in aligned pairs p[i] = (x[2i], x[2i+1]); an output o[j] = sum_i w[i] x[j+i],
// i = 0 .. 2R, w[i] = tap |i - R|, is accumulated over TAP PAIRS, so that every operand is an aligned pair whatever the
// parity of j:
//   j = 2a      acc  = sum_m (w[2m],   w[2m+1]) * p[a+m],     m < R;   o = acc.x + acc.y + w[2R] x[j+2R]   (= p[a+R].x)
//   j = 2a + 1  acc  = sum_m (w[2m+1], w[2m+2]) * p[a+1+m],   m < R;   o = acc.x + acc.y + w[0]  x[j]      (= p[a].y)
// R + 2 instructions per output instead of 2R + 1.  The two pair tables and the edge tap are kernel arguments (SGPR pairs).
typedef float f2 __attribute__((ext_vector_type(2)));
template <int R>
struct TapsP { f2 we[R], wo[R]; float w0; };

// NO outputs (NO even) from the R + NO / 2 pairs p[]: o[j] = sum_i w[i] x[j + i].  The tap-pair loop is the OUTER loop, so that
// consecutive packed multiply-adds are independent (a dependent v_pk_fma_f32 chain costs a wait state per link).
template <int R, int NO>
__device__ __forceinline__ void fir_run_pk(const f2* p, const TapsP<R>& t, float* o) {
    f2 acc[NO];
#pragma unroll
    for (int a = 0; a < NO / 2; ++a) { acc[2 * a] = p[a] * t.we[0]; acc[2 * a + 1] = p[a + 1] * t.wo[0]; }
#pragma unroll
    for (int m = 1; m < R; ++m) {
#pragma unroll
        for (int a = 0; a < NO / 2; ++a) {
            acc[2 * a] = __builtin_elementwise_fma(p[a + m], t.we[m], acc[2 * a]);
            acc[2 * a + 1] = __builtin_elementwise_fma(p[a + 1 + m], t.wo[m], acc[2 * a + 1]);
        }
    }
#pragma unroll
    for (int a = 0; a < NO / 2; ++a) {
        o[2 * a] = fmaf(p[a + R].x, t.w0, acc[2 * a].x) + acc[2 * a].y;
        o[2 * a + 1] = fmaf(p[a].y, t.w0, acc[2 * a + 1].y) + acc[2 * a + 1].x;
    }
}
// the same in groups of IL outputs (IL / 2 pairs of accumulators live at a time)
template <int R, int NO, int IL>
__device__ __forceinline__ void fir_run_pk_il(const f2* p, const TapsP<R>& t, float* o) {
#pragma unroll
    for (int g = 0; g < NO / IL; ++g) fir_run_pk<R, IL>(p + g * IL / 2, t, o + g * IL);
}
template <int R>
inline TapsP<R> pack_taps(const Taps& taps) {
    float w[2 * R + 1];
    for (int i = 0; i <= 2 * R; ++i) w[i] = (float)taps.w[i < R ? R - i : i - R];
    TapsP<R> t;
    for (int m = 0; m < R; ++m) { t.we[m] = (f2){w[2 * m], w[2 * m + 1]}; t.wo[m] = (f2){w[2 * m + 1], w[2 * m + 2]}; }
    t.w0 = w[0];
    return t;
}

// DS2: `img` is the FULL-resolution uint8 stack [N][H2][W2] and the filtered image is its x0.5 area downsample
// (cv2.resize(fx = fy = 0.5, INTER_AREA), the rule of area_down2_kernel): the 2 x 2 cells are averaged in the loader, so the
// coarse image of matcher.py:255-256 is never written (SH, SW = half_size(H2), half_size(W2))
template <typename T, int R, int NT, bool DS2 = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(R <= FB_DOG_R4 ? 4 : (R <= 12 ? 3 : 2)))) void dog_stream(const T* __restrict__ img, float* __restrict__ out, int SH, int SW, int signed_out,
                                                const int* __restrict__ sizes, const TapsP<R> taps, int TX, int SY, int H2 = 0, int W2 = 0,
                                                const T* __restrict__ img1 = nullptr, int nsplit = 0x7fffffff, int per8 = 0, int gx = 1, int gy = 1, int gz = 1) {
    constexpr int CH = SCH, RN = SRN;
    static_assert(CH % 2 == 0 && RN % 2 == 0, "pairs");
    constexpr int PI = spitch(NT + 2 * R + 4), PA = spitch(NT), PE = spitch(NT + 4);
    constexpr int NQ = (RN + 2 * R + 3) / 4;               // float4 reads per horizontal run (over-reads up to 3 staged values)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* In = smem;                    // [2][CH][PI]  input rows (float), column i <-> global x0 - 2R + i
    float* A = In + 2 * CH * PI;         // [CH][PA]     A = Gx I,           column i <-> global x0 - R + i
    float* E = A + CH * PA;              // [CH][PE]     E = Gy (I - Gy A),  column i <-> global x0 - R + i
    // per8 > 0: a launch of 8 * per8 workgroups in one dimension; workgroup b takes item (b % 8) * per8 + b / 8 of the
    // (band, segment, image) list -- every XCD (workgroups are dealt to them round-robin) filters a contiguous run of whole
    // images, so the bands of a row meet in ONE L2 before they go to memory (fb_ncc_p2.inc: p2_item has the measurement)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (per8 > 0) {
        const int w = (int)(blockIdx.x & 7) * per8 + (int)(blockIdx.x >> 3);
        if (w >= gx * gy * gz) return;
        bx = w % gx; by = (w / gx) % gy; bz = w / (gx * gy);
    }
    const int n = bz, x0 = bx * TX, y0 = by * SY;
    const int H = sizes ? sizes[2 * n] : SH, W = sizes ? sizes[2 * n + 1] : SW;
    const int tid = threadIdx.x;
    float* __restrict__ oimg = out + (size_t)n * SH * SW;
    const int sw = min(TX, SW - x0);                           // band width inside the slot
    if (x0 >= W || y0 >= H) {
        // a band / segment of the slot outside the image: zero, so that windows cropped past the image border read the fill value
        const int rows = min(SY, SH - y0);
        for (int i = tid; i < rows * sw; i += NT) oimg[(uint32_t)((y0 + i / sw) * SW + x0 + i % sw)] = 0.f;
        return;
    }
    // images nsplit .. of the launch come from a second stack (the two strips of a batch of pairs filtered in one launch)
    const T* __restrict__ src = (n < nsplit ? img : img1) + (size_t)(n < nsplit ? n : n - nsplit) * (DS2 ? (size_t)H2 * W2 : (size_t)SH * SW);
    const int bw = min(TX, W - x0);                            // image columns of the band
    const int NV = bw + 2 * R;                                 // columns of the vertical passes (<= NT)
    const int NA = (NV + RN - 1) / RN * RN;                    // columns of A computed by the first pass (<= NT)
    const int NIN = NA + 2 * R;                                // input columns staged
    const int ye = min(y0 + SY, H);
    const bool top = y0 == 0;
    const int a0 = top ? 0 : y0 - 2 * R;                       // first input row fed
    const int nch = (ye + 2 * R - a0 + CH - 1) / CH;
    // per-thread constants
    const int gx_a = min(max(x0 - 2 * R + tid, 0), W - 1), gx_b = min(max(x0 - 2 * R + tid + NT, 0), W - 1);
    const bool ld_b = tid + NT < NIN;
    const int nr1 = NA / RN, nr2 = (bw + RN - 1) / RN;
    // horizontal sweeps: 16 consecutive lanes = 8 consecutive runs x 2 consecutive rows.  A run starts every 32 B and the row
    // pitch is an odd multiple of 16 B, so the sixteen 16-byte reads of every lane group of a ds_read_b128
    // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...) fall on sixteen different bank quads: no conflicts (lanes in run
    // order on one row conflict two ways)
    const int r1 = 2 * ((tid >> 4) % (CH / 2)) + ((tid >> 3) & 1), u1 = 8 * ((tid >> 4) / (CH / 2)) + (tid & 7), r2 = r1, u2 = u1;
    const bool h1_on = u1 < nr1, h2_on = u1 < nr2, v_on = tid < NV;
    const int cxv = min(max(x0 - R + tid, 0), W - 1) - (x0 - R);     // clamped column (index in A / E space) of the vertical thread
    f2 wA[R], wD[R];                                            // the last 2R rows of A and D of the thread's column, as row pairs
    float dI[R], dlast = 0.f;
    float pa[CH], pb[CH];
    // LDS offsets of the thread's items (floats): every access below is one of these plus an immediate
    const int o1r = r1 * PI + u1 * RN, o1w = r1 * PA + u1 * RN, o2r = r2 * PE + u2 * RN;
    const uint32_t absmask = signed_out ? 0xffffffffu : 0x7fffffffu;
    // one coarse pixel from its 2 x 2 cell of the full-resolution image (cells cut by the edge of an odd-sized image average
    // the pixels that exist, round half to even: resizeAreaFast)
    auto cell = [&](int gy, int gx) -> float {
        const uint8_t* p = reinterpret_cast<const uint8_t*>(src) + (uint32_t)(2 * gy * W2 + 2 * gx);
        const bool x1 = 2 * gx + 1 < W2, y1 = 2 * gy + 1 < H2;
        const int a = p[0], b = x1 ? p[1] : 0, c2 = y1 ? p[W2] : 0, d = (x1 && y1) ? p[W2 + 1] : 0;
        if (x1 && y1) return (float)((a + b + c2 + d + 2) >> 2);
        return rintf((float)(a + b + c2) / (float)(1 + (int)x1 + (int)y1));
    };
    auto fetch = [&](int c) {
        const int ya = a0 + c * CH;
        if (DS2) {
            if ((W2 & 1) == 0 && ya >= 0 && 2 * (ya + CH) <= H2) {
                // uniform fast path: every cell of the chunk is a whole 2 x 2 cell and both of its rows start on an even
                // address: two 16-bit loads per cell
                const uint8_t* base = reinterpret_cast<const uint8_t*>(src);
                auto cell2 = [&](uint32_t off) -> float {
                    const uint32_t u = *reinterpret_cast<const uint16_t*>(base + off), v = *reinterpret_cast<const uint16_t*>(base + off + (uint32_t)W2);
                    return (float)(((u & 0xffu) + (u >> 8) + (v & 0xffu) + (v >> 8) + 2u) >> 2);
                };
                const uint32_t oa = (uint32_t)(2 * ya * W2 + 2 * gx_a), ob = (uint32_t)(2 * ya * W2 + 2 * gx_b);
#pragma unroll
                for (int r = 0; r < CH; ++r) pa[r] = cell2(oa + (uint32_t)(2 * r * W2));
                if (ld_b) {
#pragma unroll
                    for (int r = 0; r < CH; ++r) pb[r] = cell2(ob + (uint32_t)(2 * r * W2));
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < CH; ++r) pa[r] = cell(min(max(ya + r, 0), H - 1), gx_a);
            if (ld_b) {
#pragma unroll
                for (int r = 0; r < CH; ++r) pb[r] = cell(min(max(ya + r, 0), H - 1), gx_b);
            }
            return;
        }
        if (ya >= 0 && ya + CH <= H) {                     // uniform: rows inside the image
            const uint32_t oa = (uint32_t)(ya * SW + gx_a), ob = (uint32_t)(ya * SW + gx_b);
#pragma unroll
            for (int r = 0; r < CH; ++r) pa[r] = load_px(src, oa + (uint32_t)(r * SW));
            if (ld_b) {
#pragma unroll
                for (int r = 0; r < CH; ++r) pb[r] = load_px(src, ob + (uint32_t)(r * SW));
            }
        } else {
#pragma unroll
            for (int r = 0; r < CH; ++r) pa[r] = load_px(src, (uint32_t)(min(max(ya + r, 0), H - 1) * SW + gx_a));
            if (ld_b) {
#pragma unroll
                for (int r = 0; r < CH; ++r) pb[r] = load_px(src, (uint32_t)(min(max(ya + r, 0), H - 1) * SW + gx_b));
            }
        }
    };
    auto stage = [&](int c) {
        float* dst = In + (c & 1) * CH * PI + tid;
#pragma unroll
        for (int r = 0; r < CH; ++r) dst[r * PI] = pa[r];
        if (ld_b) {
#pragma unroll
            for (int r = 0; r < CH; ++r) dst[r * PI + NT] = pb[r];
        }
    };
    // input rows of chunk c + 2 are requested between the vertical phase of chunk c and its output sweep and staged after the
    // first sweep of chunk c + 1: in flight over two sweeps, and not live in registers during the vertical phase (the register peak)
    fetch(0);
    stage(0);
    __syncthreads();
    if (nch > 1) fetch(1);
    for (int c = 0; c < nch; ++c) {
        const float* in = In + (c & 1) * CH * PI;
        // ---- A = Gx I: RN outputs from RN + 2R staged values
        if (h1_on) {
            f2 v[2 * NQ];
            const float4* p4 = reinterpret_cast<const float4*>(in + o1r);
#pragma unroll
            for (int q = 0; q < NQ; ++q) { const float4 f = p4[q]; v[2 * q] = (f2){f.x, f.y}; v[2 * q + 1] = (f2){f.z, f.w}; }
            float o[RN];
            fir_run_pk_il<R, RN, FIL>(v, taps, o);
            float4* d4 = reinterpret_cast<float4*>(A + o1w);
            d4[0] = make_float4(o[0], o[1], o[2], o[3]);
            d4[1] = make_float4(o[4], o[5], o[6], o[7]);
        }
        if (c + 1 < nch) stage(c + 1);
        __syncthreads();
        // ---- B = Gy A, D = I - B, E = Gy D on the thread's column
        if (v_on) {
            f2 fa[R + CH / 2], fd[R + CH / 2];                               // row pairs: fa[i] = rows (2i, 2i + 1) of the window
            float fi[R + CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) { fa[R + j / 2][j & 1] = A[j * PA + cxv]; fi[R + j] = in[j * PI + cxv + R]; }
            if (c == 0) {
#pragma unroll
                for (int i = 0; i < R; ++i) wA[i] = fa[R].xx;               // rows above the first fed row: replicas (exact at the image top)
#pragma unroll
                for (int i = 0; i < R; ++i) dI[i] = fi[R];
#pragma unroll
                for (int i = 0; i < R; ++i) wD[i] = (f2){0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < R; ++i) { fa[i] = wA[i]; fd[i] = wD[i]; }
#pragma unroll
            for (int i = 0; i < R; ++i) fi[i] = dI[i];
            const int rD0 = a0 + c * CH - R;                                 // image row of D[0] of this chunk
            {
                float bv[CH];
                fir_run_pk_il<R, CH, FIL>(fa, taps, bv);
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const float d = fi[j] - bv[j];
                    dlast = (rD0 + j <= H - 1) ? d : dlast;                  // D[y > H-1] = D[H-1]
                    fd[R + j / 2][j & 1] = dlast;
                }
            }
            if (top && c == R / CH) {
                // the image's first row of D has just been computed: every row above it is its replica
                const float d0 = fd[R + (R % CH) / 2][(R % CH) & 1];
#pragma unroll
                for (int i = 0; i < 2 * R + R % CH; ++i) fd[i / 2][i & 1] = d0;
            }
            {
                float ev[CH];
                fir_run_pk_il<R, CH, FIL>(fd, taps, ev);
#pragma unroll
                for (int j = 0; j < CH; ++j) E[j * PE + tid] = ev[j];
            }
#pragma unroll
            for (int i = 0; i < R; ++i) { wA[i] = fa[CH / 2 + i]; wD[i] = fd[CH / 2 + i]; }
#pragma unroll
            for (int i = 0; i < R; ++i) dI[i] = fi[CH + i];
        }
        __syncthreads();
        if (c + 2 < nch) fetch(c + 2);
        // ---- out = Gx E, rows a0 + c CH - 2R + r
        if (h2_on) {
            const int gy = a0 + c * CH - 2 * R + r2;
            if (gy >= y0 && gy < ye) {
                f2 v[2 * NQ];
                const float4* p4 = reinterpret_cast<const float4*>(E + o2r);
#pragma unroll
                for (int q = 0; q < NQ; ++q) { const float4 f = p4[q]; v[2 * q] = (f2){f.x, f.y}; v[2 * q + 1] = (f2){f.z, f.w}; }
                float o[RN];
                fir_run_pk_il<R, RN, FIL>(v, taps, o);
#pragma unroll
                for (int j = 0; j < RN; ++j) o[j] = __uint_as_float(__float_as_uint(o[j]) & absmask);
                const int gx = x0 + u2 * RN;
                float* d = oimg + (uint32_t)(gy * SW + gx);
                if (gx + RN <= x0 + bw) {                  // a whole run inside the band (the last run of a band may hang over into the next one)
                    *reinterpret_cast<float4u*>(d) = (float4u){o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<float4u*>(d + 4) = (float4u){o[4], o[5], o[6], o[7]};
                } else {
#pragma unroll
                    for (int j = 0; j < RN; ++j)
                        if (gx + j < x0 + bw) d[j] = o[j];
                }
            }
        }
    }
    // slot pixels of the band right of / below the image (per-image sizes): 0
    if (sizes) {
        if (bw < sw) {
            const int wz = sw - bw;
            for (int i = tid; i < (ye - y0) * wz; i += NT) oimg[(uint32_t)((y0 + i / wz) * SW + x0 + bw + i % wz)] = 0.f;
        }
        const int yz = min(y0 + SY, SH);
        if (ye < yz)
            for (int i = tid; i < (yz - ye) * sw; i += NT) oimg[(uint32_t)((ye + i / sw) * SW + x0 + i % sw)] = 0.f;
    }
}

struct StreamPlan { int NT, TX, nb, SY, nseg; };

// band / segment geometry: TX + 2R <= NT, bands of equal width; segments so that the grid fills the chip in whole rounds
inline StreamPlan plan_stream(int N, int H, int W, int R, int num_cu) {
    StreamPlan best{0, 0, 0, 0, 0};
    double best_cost = 1e300;
    for (int NT : {64, 192}) {
        const int txmax = NT - 2 * R;
        if (txmax < 16) continue;
        // TX even: the tap pairing of an output follows the parity of its position in the band (fir_run_pk), and a pixel's
        // value must not depend on how a launch cuts the image into bands (a slot of another width, a stack of two strips)
        const int nb = (W + txmax - 1) / txmax, TX = ((W + nb - 1) / nb + 1) & ~1;
        // thread slots per chunk of 8 rows: two horizontal sweeps + two vertical passes of NT threads each
        const double per_px = (double)nb * NT / W;
        const double cost = per_px * (NT == 64 ? 1.08 : 1.0);            // small workgroups pay a little more per barrier / prologue
        if (cost < best_cost) { best_cost = cost; best = StreamPlan{NT, TX, nb, 0, 0}; }
    }
    // segments: about 4 workgroups per CU resident; whole rounds of the grid, rows per segment >= 64
    const int lds_wgs = best.NT == 64 ? 8 : 5;
    const double cap = (double)num_cu * lds_wgs;
    int best_seg = 1;
    double best_t = 1e300;
    for (int nseg = 1; nseg <= std::max(1, H / 64); ++nseg) {
        const int SY = ((H + nseg - 1) / nseg + SCH - 1) / SCH * SCH;
        const int ns = (H + SY - 1) / SY;
        const double wgs = (double)N * best.nb * ns;
        const double rounds = std::ceil(wgs / cap);
        const double t = rounds * (SY + 4.0 * R + 16.0);
        if (t < best_t - 1e-9) { best_t = t; best_seg = ns; best.SY = SY; }
    }
    best.nseg = best_seg;
    return best;
}

// FEABAS_HIP_DOG_XCD=1: one contiguous eighth of the (band, segment, image) grid per XCD instead of launch order.  Off by default:
// measured 8.99 -> 8.86 ms of DoG per 512 pairs and no change of the headline (profiles/r06j_dog_xcd_ab.txt) -- the kernel is
// bound by its taps, not by its stores.
inline int dog_xcd_per8(int gx, int gy, int gz) {
    static const int on = [] { const char* e = getenv("FEABAS_HIP_DOG_XCD"); return e ? atoi(e) : 0; }();
    const long long total = (long long)gx * gy * gz;
    if (!on || total < 64 || total > (1LL << 30)) return 0;
    return (int)((total + 7) / 8);
}

template <int R, int NT>
int launch_stream_ds2_nt(fb_ctx* ctx, const uint8_t* img, float* out, int N, int H2, int W2, int signed_out, const TapsP<R>& tf, const StreamPlan& pl) {
    constexpr int PI = spitch(NT + 2 * R + 4), PA = spitch(NT), PE = spitch(NT + 4);
    const size_t lds = (size_t)(2 * SCH * PI + SCH * PA + SCH * PE) * sizeof(float);
    auto kern = dog_stream<uint8_t, R, NT, true>;
    FB_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(pl.nb, pl.nseg, N);
    const int per8 = dog_xcd_per8(pl.nb, pl.nseg, N);
    if (per8 > 0) grid = dim3(8 * per8, 1, 1);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, ctx->stream, img, out, half_size(H2), half_size(W2), signed_out, (const int*)nullptr, tf, pl.TX, pl.SY, H2, W2,
                       (const uint8_t*)ctx->dog_img1, ctx->dog_img1 ? ctx->dog_nsplit : 0x7fffffff, per8, pl.nb, pl.nseg, N);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

template <int R>
int launch_stream_ds2(fb_ctx* ctx, const uint8_t* img, float* out, int N, int H2, int W2, int signed_out, const Taps& taps) {
    const TapsP<R> tf = pack_taps<R>(taps);
    if ((size_t)H2 * W2 >= ((size_t)1 << 30) || N > 65535) return fb_fail(ctx, FB_ERR_ARG, "fb_dog_down2_dev: stack too large for one launch");
    const StreamPlan pl = plan_stream(N, half_size(H2), half_size(W2), R, ctx->prop.multiProcessorCount);
    if (pl.NT == 64) return launch_stream_ds2_nt<R, 64>(ctx, img, out, N, H2, W2, signed_out, tf, pl);
    return launch_stream_ds2_nt<R, 192>(ctx, img, out, N, H2, W2, signed_out, tf, pl);
}

template <typename T, int R, int NT>
int launch_stream_nt(fb_ctx* ctx, const T* img, float* out, int N, int H, int W, int signed_out, const TapsP<R>& tf, const int* sizes, const StreamPlan& pl) {
    constexpr int PI = spitch(NT + 2 * R + 4), PA = spitch(NT), PE = spitch(NT + 4);
    const size_t lds = (size_t)(2 * SCH * PI + SCH * PA + SCH * PE) * sizeof(float);
    auto kern = dog_stream<T, R, NT>;
    FB_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(pl.nb, pl.nseg, N);
    const int per8 = dog_xcd_per8(pl.nb, pl.nseg, N);
    if (per8 > 0) grid = dim3(8 * per8, 1, 1);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, ctx->stream, img, out, H, W, signed_out, sizes, tf, pl.TX, pl.SY, 0, 0, (const T*)ctx->dog_img1,
                       ctx->dog_img1 ? ctx->dog_nsplit : 0x7fffffff, per8, pl.nb, pl.nseg, N);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

template <typename T, int R>
int launch_stream(fb_ctx* ctx, const T* img, float* out, int N, int H, int W, int signed_out, const Taps& taps, const int* sizes) {
    const TapsP<R> tf = pack_taps<R>(taps);
    if ((size_t)H * W >= ((size_t)1 << 30)) return fb_fail(ctx, FB_ERR_ARG, "fb_dog: image of %d x %d pixels exceeds the 32-bit offsets of the fast kernel", H, W);
    if (N > 65535) return fb_fail(ctx, FB_ERR_ARG, "fb_dog: more than 65535 images in one launch");
    const StreamPlan pl = plan_stream(N, H, W, R, ctx->prop.multiProcessorCount);
    if (pl.nseg > 65535) return fb_fail(ctx, FB_ERR_ARG, "fb_dog: image too tall for one launch");
    if (pl.NT == 64) return launch_stream_nt<T, R, 64>(ctx, img, out, N, H, W, signed_out, tf, sizes, pl);
    return launch_stream_nt<T, R, 192>(ctx, img, out, N, H, W, signed_out, tf, sizes, pl);
}

template <typename T>
int launch_fast_any(fb_ctx* ctx, int r, const T* img, float* out, int N, int H, int W, int signed_out, const Taps& taps, bool* done,
                    const int* sizes = nullptr) {
    *done = true;
    if (!ctx->dog_tiles) {
        switch (r) {
            case 5: return launch_stream<T, 5>(ctx, img, out, N, H, W, signed_out, taps, sizes);
            case 6: return launch_stream<T, 6>(ctx, img, out, N, H, W, signed_out, taps, sizes);
            case 8: return launch_stream<T, 8>(ctx, img, out, N, H, W, signed_out, taps, sizes);
            case 10: return launch_stream<T, 10>(ctx, img, out, N, H, W, signed_out, taps, sizes);
            case 12: return launch_stream<T, 12>(ctx, img, out, N, H, W, signed_out, taps, sizes);
            case 14: return launch_stream<T, 14>(ctx, img, out, N, H, W, signed_out, taps, sizes);
            default: *done = false; return FB_OK;
        }
    }
    switch (r) {
        case 5: return launch_fast<T, 5>(ctx, img, out, N, H, W, signed_out, taps, sizes);
        case 6: return launch_fast<T, 6>(ctx, img, out, N, H, W, signed_out, taps, sizes);
        case 8: return launch_fast<T, 8>(ctx, img, out, N, H, W, signed_out, taps, sizes);
        case 10: return launch_fast<T, 10>(ctx, img, out, N, H, W, signed_out, taps, sizes);
        case 12: return launch_fast<T, 12>(ctx, img, out, N, H, W, signed_out, taps, sizes);
        case 14: return launch_fast<T, 14>(ctx, img, out, N, H, W, signed_out, taps, sizes);
        default: *done = false; return FB_OK;
    }
}

template <typename T, int NPASS>
int launch_tile(fb_ctx* ctx, const T* img, float* out, const float* halo, int N, int H, int W, int r, int signed_out,
                float in_scale, const uint8_t* mask, const Taps& taps, size_t per_image = 0, const int* ids = nullptr) {
    // largest square tile whose two LDS buffers fit in 64 KiB (2 workgroups per CU) or, failing that, 150 KiB
    const int hal = (NPASS == 4) ? 2 * r : r;
    int T_ = 64;
    auto lds_bytes = [&](int t) { const int A = t + 2 * hal; return (size_t)(A * A + A * (A - 2 * r)) * sizeof(float); };
    while (T_ > 16 && lds_bytes(T_) > 150 * 1024) T_ -= 16;
    if (lds_bytes(T_) > 150 * 1024) return fb_fail(ctx, FB_ERR_ARG, "fb_dog: radius %d too large for the LDS tile", r);
    const size_t lds = lds_bytes(T_);
    auto kern = dog_tile<T, NPASS>;
    FB_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    constexpr int kMaxZ = 32768;                          // images per launch: the grid's z extent is limited to 65535
    for (int n0 = 0; n0 < N; n0 += kMaxZ) {
        const size_t io = (size_t)n0 * H * W, mo = (size_t)n0 * per_image;
        dim3 grid(fb_cdiv(W, T_), fb_cdiv(H, T_), std::min(kMaxZ, N - n0));
        if (ids) hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, img, out, halo, H, W, r, T_, T_, signed_out, in_scale, mask, taps, per_image, ids + n0);
        else hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, img ? img + io : img, out + io, halo ? halo + mo : halo, H, W, r, T_, T_, signed_out,
                                in_scale, mask ? mask + mo : mask, taps, per_image, (const int*)nullptr);
    }
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

template <typename T>
int dog_dev_t(fb_ctx* ctx, const T* img, int N, int H, int W, double sigma, const uint8_t* mask, int signed_out, float* out,
              bool mask_per_image = false) {
    int rc = FB_OK, r = 0, r_halo = 0;
    Taps taps, taps_halo;
    float* halo = nullptr;
    float ptp_halo = 0.f;
    std::vector<int> sel;                 // images of a per-image mask stack whose mask has a zero (when not all of them)
    int* d_sel = nullptr;
    if (mask) {
        // common.py:368: only when the mask has a zero somewhere
        int* flag = reinterpret_cast<int*>(ctx->small);              // the context's scratch: no hipMalloc / hipFree (a device-wide drain) per call
        FB_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
        const size_t nmask = mask_per_image ? (size_t)N : 1, per_image = mask_per_image ? (size_t)H * W : 0;
        hipLaunchKernelGGL(any_zero_kernel, dim3(1024), dim3(256), 0, ctx->stream, mask, nmask * H * W, flag);
        int hflag = 0;
        FB_HIP(ctx, hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (hflag && mask_per_image && N > 1 && !ctx->dog_exact) {
            // a stack with one mask per image (the rendered blocks of crop_multiple): usually only the blocks at the border of
            // the mesh have holes.  The halo term of an image whose mask has no zero is exactly zero, so only the images with
            // holes take the masked tile kernels; the others go through the streaming kernel with the rest of the stack
            int* d_each = nullptr;
            if (fb_malloc(ctx, sizeof(int) * (size_t)N, (void**)&d_each) == FB_OK) {
                hipLaunchKernelGGL(any_zero_each_kernel, dim3(N), dim3(256), 0, ctx->stream, mask, (size_t)H * W, d_each);
                std::vector<int> each((size_t)N);
                rc = fb_copy_d2h(ctx, each.data(), d_each, sizeof(int) * (size_t)N);
                if (!rc) for (int i = 0; i < N; ++i) if (each[i]) sel.push_back(i);
                if (!rc && (int)sel.size() < N) {
                    if ((rc = fb_copy_h2d(ctx, d_each, sel.data(), sizeof(int) * sel.size()))) { fb_free(ctx, d_each); return rc; }
                    d_sel = d_each;
                } else {
                    sel.clear();
                    fb_free(ctx, d_each);
                    if (rc) return rc;
                }
            }
        }
        if (hflag) {
            const int nblk = 256, nslots = nblk * 4;
            float* mm = reinterpret_cast<float*>(ctx->small) + 64;       // 8 KiB of the 64 KiB scratch
            hipLaunchKernelGGL(minmax_kernel<T>, dim3(nblk), dim3(256), 0, ctx->stream, img, (size_t)N * H * W, mm);
            std::vector<float> hmm(2 * nslots);
            FB_HIP(ctx, hipMemcpyAsync(hmm.data(), mm, sizeof(float) * 2 * nslots, hipMemcpyDeviceToHost, ctx->stream));
            FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
            float lo = INFINITY, hi = -INFINITY;
            for (int i = 0; i < nslots; ++i) { lo = std::min(lo, hmm[2 * i]); hi = std::max(hi, hmm[2 * i + 1]); }
            const float ptp = hi - lo;
            // halo = G_{sigma*sqrt2}(ptp*(mask==0)) * 2   (common.py:369-371)
            const double sc = std::sqrt(2.0 * sigma * sigma);
            rc = set_taps(ctx, sc, &r, &taps);
            if (rc) { if (d_sel) fb_free(ctx, d_sel); return rc; }
            r_halo = r; taps_halo = taps; ptp_halo = ptp;
            FB_HIP(ctx, hipMalloc(&halo, sizeof(float) * nmask * H * W));
            if (d_sel) rc = launch_tile<T, 2>(ctx, (const T*)nullptr, halo, nullptr, (int)sel.size(), H, W, r, 1, ptp, mask, taps, per_image, d_sel);
            else rc = launch_tile<T, 2>(ctx, (const T*)nullptr, halo, nullptr, (int)nmask, H, W, r, 1, ptp, mask, taps, per_image);
            if (rc) { hipFree(halo); if (d_sel) fb_free(ctx, d_sel); return rc; }
        }
    }
    rc = set_taps(ctx, sigma, &r, &taps);
    if (!rc) {
        bool done = false;
        if ((!halo || d_sel) && !ctx->dog_exact) {
            FB_PROF_B(ctx, "dog_fast", (double)N * H * W * (sizeof(T) + 4.0));
            rc = launch_fast_any<T>(ctx, r, img, out, N, H, W, signed_out, taps, &done);
        }
        if (!rc && d_sel && done) {
            // the images with holes again, through the masked kernels (the streaming kernel's values of those are overwritten)
            FB_PROF_B(ctx, "dog_tile", (double)sel.size() * H * W * (sizeof(T) + 4.0));
            rc = launch_tile<T, 4>(ctx, img, out, halo, (int)sel.size(), H, W, r, signed_out, 0.f, nullptr, taps, (size_t)H * W, d_sel);
        } else if (!rc && !done) {
            FB_PROF_B(ctx, "dog_tile", (double)N * H * W * (sizeof(T) + 4.0));
            // (no streaming kernel for this radius: every image through the tile kernels; the halo planes of images without
            // holes were not written when only the selected ones were made -- they are made now)
            if (d_sel) rc = launch_tile<T, 2>(ctx, (const T*)nullptr, halo, nullptr, N, H, W, r_halo, 1, ptp_halo, mask, taps_halo, (size_t)H * W);
            if (!rc) rc = launch_tile<T, 4>(ctx, img, out, halo, N, H, W, r, signed_out, 0.f, nullptr, taps, (halo && mask_per_image) ? (size_t)H * W : 0);
        }
    }
    if (halo || d_sel) {
        hipStreamSynchronize(ctx->stream);
        if (halo) hipFree(halo);
        if (d_sel) fb_free(ctx, d_sel);
    }
    return rc;
}

}  // namespace

extern "C" {

int fb_dog_dev(fb_ctx* ctx, const void* img, int dtype, int N, int H, int W, double sigma, const uint8_t* mask,
               int signed_out, float* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && H > 0 && W > 0 && sigma > 0);
    FB_CHECK_ARG(ctx, dtype == FB_U8 || dtype == FB_F32);
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, img && out);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    if (dtype == FB_U8) return dog_dev_t<uint8_t>(ctx, (const uint8_t*)img, N, H, W, sigma, mask, signed_out, out);
    return dog_dev_t<float>(ctx, (const float*)img, N, H, W, sigma, mask, signed_out, out);
}

// common.masked_dog_filter(stack, sigma, mask=masks) with one mask per image (masks [N][H][W]): what
// MeshRenderer.crop_multiple(log_sigma=sigma) applies to its N x h x w stack (renderer.py:632-641); np.ptp runs over the
// whole stack, as there.
int fb_dog_masks_dev(fb_ctx* ctx, const void* img, int dtype, int N, int H, int W, double sigma, const uint8_t* masks,
                     int signed_out, float* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && H > 0 && W > 0 && sigma > 0);
    FB_CHECK_ARG(ctx, dtype == FB_U8 || dtype == FB_F32);
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, img && out && masks);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    if (dtype == FB_U8) return dog_dev_t<uint8_t>(ctx, (const uint8_t*)img, N, H, W, sigma, masks, signed_out, out, true);
    return dog_dev_t<float>(ctx, (const float*)img, N, H, W, sigma, masks, signed_out, out, true);
}

// DoG of N images of unequal size that share a padded stack: image n is the sizes[n] = {h, w} corner of its H x W slot
// (device int32 [N][2]); each is filtered as an h x w image ('nearest' extension at ITS border), pixels outside are
// left untouched.  Fast path only (the radii of the matcher's sigmas).
int fb_dog_sizes_dev(fb_ctx* ctx, const void* img, int dtype, int N, int H, int W, const int* sizes, double sigma, int signed_out,
                     float* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && H > 0 && W > 0 && sigma > 0 && sizes);
    FB_CHECK_ARG(ctx, dtype == FB_U8 || dtype == FB_F32);
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, img && out);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    int r = 0;
    Taps taps;
    int rc = set_taps(ctx, sigma, &r, &taps);
    if (rc) return rc;
    bool done = false;
    FB_PROF_B(ctx, "dog_fast", (double)N * H * W * ((dtype == FB_U8 ? 1.0 : 4.0) + 4.0));
    if (dtype == FB_U8) rc = launch_fast_any<uint8_t>(ctx, r, (const uint8_t*)img, out, N, H, W, signed_out, taps, &done, sizes);
    else rc = launch_fast_any<float>(ctx, r, (const float*)img, out, N, H, W, signed_out, taps, &done, sizes);
    if (!rc && !done) return fb_fail(ctx, FB_ERR_ARG, "fb_dog_sizes_dev: sigma %.3f (radius %d) has no fast kernel", sigma, r);
    return rc;
}

// masked_dog_filter(cv2.resize(img, fx = fy = 0.5, INTER_AREA), sigma) without the intermediate image: matcher.py:255-256 +
// 273-274 for N resident uint8 images [N][H2][W2]; out float32 [N][half_size(H2)][half_size(W2)]
int fb_dog_down2_dev(fb_ctx* ctx, const uint8_t* img, int N, int H2, int W2, double sigma, int signed_out, float* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && H2 > 1 && W2 > 1 && sigma > 0);
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, img && out);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    int r = 0;
    Taps taps;
    int rc = set_taps(ctx, sigma, &r, &taps);
    if (rc) return rc;
    FB_PROF_B(ctx, "dog_fast", (double)N * ((double)H2 * W2 + 4.0 * half_size(H2) * half_size(W2)));
    switch (r) {
        case 5: return launch_stream_ds2<5>(ctx, img, out, N, H2, W2, signed_out, taps);
        case 6: return launch_stream_ds2<6>(ctx, img, out, N, H2, W2, signed_out, taps);
        case 8: return launch_stream_ds2<8>(ctx, img, out, N, H2, W2, signed_out, taps);
        case 10: return launch_stream_ds2<10>(ctx, img, out, N, H2, W2, signed_out, taps);
        default: return fb_fail(ctx, FB_ERR_ARG, "fb_dog_down2_dev: sigma %.3f (radius %d) has no streaming kernel", sigma, r);
    }
}

// The two strip stacks of a batch of pairs in ONE launch: images 0 .. N-1 from img0, N .. 2N-1 from img1, out [2N].  A launch
// of 2N images is cut into fewer row segments than two of N (the segments of plan_stream fill the chip in whole rounds), so
// fewer warm-up rows are filtered twice: 6-7 % less work at 128 pairs of 4096 x 510.
int fb_dog_pair_dev(fb_ctx* ctx, const void* img0, const void* img1, int dtype, int N, int H, int W, double sigma, int signed_out, float* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && H > 0 && W > 0 && sigma > 0 && (dtype == FB_U8 || dtype == FB_F32));
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, img0 && img1 && out);
    int r = 0;
    Taps taps;
    int rc = set_taps(ctx, sigma, &r, &taps);
    if (rc) return rc;
    const bool streamed = !ctx->dog_tiles && !ctx->dog_exact && (r == 5 || r == 6 || r == 8 || r == 10 || r == 12 || r == 14) && 2 * (long long)N <= 65535;
    if (!streamed) {
        if ((rc = fb_dog_dev(ctx, img0, dtype, N, H, W, sigma, nullptr, signed_out, out))) return rc;
        return fb_dog_dev(ctx, img1, dtype, N, H, W, sigma, nullptr, signed_out, out + (size_t)N * H * W);
    }
    ctx->dog_img1 = img1; ctx->dog_nsplit = N;
    rc = fb_dog_dev(ctx, img0, dtype, 2 * N, H, W, sigma, nullptr, signed_out, out);
    ctx->dog_img1 = nullptr;
    return rc;
}

int fb_dog_down2_pair_dev(fb_ctx* ctx, const uint8_t* img0, const uint8_t* img1, int N, int H2, int W2, double sigma, int signed_out, float* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && H2 > 1 && W2 > 1 && sigma > 0 && 2 * (long long)N <= 65535);
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, img0 && img1 && out);
    ctx->dog_img1 = img1; ctx->dog_nsplit = N;
    const int rc = fb_dog_down2_dev(ctx, img0, 2 * N, H2, W2, sigma, signed_out, out);
    ctx->dog_img1 = nullptr;
    return rc;
}

int fb_mask_range_dev(fb_ctx* ctx, const float* img, size_t n, float lo, float hi, uint8_t* mask) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, n == 0 || (img && mask));
    if (n == 0) return FB_OK;
    hipLaunchKernelGGL(mask_range_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, ctx->stream, img, n, lo, hi, mask);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

int fb_count_nonzero_dev(fb_ctx* ctx, const uint8_t* m, size_t n, int64_t* count) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, count && (n == 0 || m));
    *count = 0;
    if (n == 0) return FB_OK;
    unsigned long long* d = reinterpret_cast<unsigned long long*>(ctx->small);
    FB_HIP(ctx, hipMemsetAsync(d, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(count_nonzero_kernel, dim3((unsigned)std::min<size_t>((n / 16 + 255) / 256 + 1, 2048)), dim3(256), 0, ctx->stream, m, n, d);
    FB_HIP(ctx, hipGetLastError());
    unsigned long long h = 0;
    FB_HIP(ctx, hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *count = (int64_t)h;
    return FB_OK;
}

int fb_dog(fb_ctx* ctx, const void* img, int dtype, int N, int H, int W, double sigma, const uint8_t* mask, int signed_out,
           float* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && H > 0 && W > 0 && sigma > 0);
    FB_CHECK_ARG(ctx, dtype == FB_U8 || dtype == FB_F32);
    if (N == 0) return FB_OK;
    FB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t px = (size_t)N * H * W, esz = dtype == FB_U8 ? 1 : 4;
    void *din = nullptr, *dout = nullptr, *dmask = nullptr;             // (the context's allocation cache: a hipMalloc / hipFree pair costs ~1 ms)
    int rc = fb_malloc(ctx, px * esz, &din);
    if (!rc) rc = fb_malloc(ctx, px * sizeof(float), &dout);
    if (!rc) rc = fb_copy_h2d(ctx, din, img, px * esz);
    if (!rc && mask) {
        rc = fb_malloc(ctx, (size_t)H * W, &dmask);
        if (!rc) rc = fb_copy_h2d(ctx, dmask, mask, (size_t)H * W);
    }
    if (!rc) rc = fb_dog_dev(ctx, din, dtype, N, H, W, sigma, (const uint8_t*)dmask, signed_out, (float*)dout);
    if (!rc) rc = fb_copy_d2h(ctx, out, dout, px * sizeof(float));
    if (din) fb_free(ctx, din);
    if (dout) fb_free(ctx, dout);
    if (dmask) fb_free(ctx, dmask);
    return rc;
}

// common.remap (cv2.remap INTER_LINEAR, BORDER_CONSTANT 0; common.py:218-255, 329-330) of resident float32 images
// through explicit per-pixel maps: the exact piecewise-linear tier of MeshRenderer.crop_multiple (renderer.py:511-563),
// whose field the host evaluates for the few blocks that no affine map approximates within tolerance.
int fb_remap_dev(fb_ctx* ctx, const float* imgs, int IH, int IW, int N, const int* img_id, int h, int w, const float* map_x,
                 const float* map_y, const uint8_t* mask, const int* origin, float* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, imgs && IH > 0 && IW > 0 && N >= 0 && h > 0 && w > 0 && img_id && map_x && map_y && origin && out);
    if (N == 0) return FB_OK;
    FB_PROF(ctx, "remap");
    const size_t total = (size_t)N * h * w;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(remap_kernel, dim3(blocks), dim3(256), 0, ctx->stream, imgs, IH, IW, N, img_id, h, w, map_x, map_y, mask, origin, out);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

int fb_area_downsample2_sizes_dev(fb_ctx* ctx, const uint8_t* img, int N, int H, int W, const int* sizes, uint8_t* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && H > 1 && W > 1);
    if (N == 0) return FB_OK;
    FB_PROF(ctx, "area_down2");
    const size_t total = (size_t)N * half_size(H) * half_size(W);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(area_down2_kernel, dim3(blocks), dim3(256), 0, ctx->stream, img, out, N, H, W, sizes);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

int fb_area_downsample2_dev(fb_ctx* ctx, const uint8_t* img, int N, int H, int W, uint8_t* out) {
    return fb_area_downsample2_sizes_dev(ctx, img, N, H, W, nullptr, out);
}

int fb_area_downsample2(fb_ctx* ctx, const uint8_t* img, int N, int H, int W, uint8_t* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && H > 1 && W > 1);
    if (N == 0) return FB_OK;
    const size_t bi = (size_t)N * H * W, bo = (size_t)N * half_size(H) * half_size(W);
    void *din = nullptr, *dout = nullptr;
    int rc = fb_malloc(ctx, bi, &din);
    if (!rc) rc = fb_malloc(ctx, bo, &dout);
    if (!rc) rc = fb_copy_h2d(ctx, din, img, bi);
    if (!rc) rc = fb_area_downsample2_dev(ctx, (const uint8_t*)din, N, H, W, (uint8_t*)dout);
    if (!rc) rc = fb_copy_d2h(ctx, out, dout, bo);
    if (din) fb_free(ctx, din);
    if (dout) fb_free(ctx, dout);
    return rc;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// cv2.resize(img, None, fx, fy, INTER_AREA) of uint8 images for any shrinking factor (matcher.py:255-256, 320-321:
// coarse_downsample / fine_downsample other than 0.5).  OpenCV's rule as published (imgproc/resize.cpp; cv2 is absent from
// the build image, so this statement is unpinned like the x0.5 one): output size cvRound(n f) per axis; when 1/f is an
// integer k on both axes the cell is summed in integers and scaled, saturate_cast<uchar>(sum * (1.f / k^2)) -- round half to
// even -- with (sum + 2) >> 2 for k = 2 (its vector kernel), and a cell cut by the image edge averages the pixels that exist,
// (float)sum / count; otherwise every axis gets a table of (source index, weight) taps -- the covered fraction of each source
// pixel over the cell width -- and the pixel is accumulated in float, taps of a row first, rows after.
namespace {
struct AreaAxis { std::vector<int> ptr, si; std::vector<float> alpha; };

// computeResizeAreaTab: taps of every output index along one axis
AreaAxis area_axis(int ssize, int dsize, double scale) {
    AreaAxis t;
    t.ptr.push_back(0);
    for (int dx = 0; dx < dsize; ++dx) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
        sx2 = std::min(sx2, ssize - 1); sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) { t.si.push_back(sx1 - 1); t.alpha.push_back((float)((sx1 - fsx1) / cell)); }
        for (int sx = sx1; sx < sx2; ++sx) { t.si.push_back(sx); t.alpha.push_back((float)(1.0 / cell)); }
        if (fsx2 - sx2 > 1e-3) { t.si.push_back(sx2); t.alpha.push_back((float)(std::min(std::min(fsx2 - sx2, 1.0), cell) / cell)); }
        t.ptr.push_back((int)t.si.size());
    }
    return t;
}

__device__ __forceinline__ uint8_t sat_u8(float v) {
    const float r = rintf(v);                                // cvRound: to nearest, ties to even (NaN -> 0 after the clamp)
    return (uint8_t)(r > 255.f ? 255.f : (r > 0.f ? r : 0.f));
}

__global__ __launch_bounds__(256) void area_fast_kernel(const uint8_t* __restrict__ img, uint8_t* __restrict__ out, int N, int H, int W, int Ho, int Wo,
                                                         int kx, int ky) {
#pragma clang fp contract(off)
    const size_t total = (size_t)N * Ho * Wo;
    const float scale = 1.f / (float)(kx * ky);
    const int wfull = W / kx;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int dx = (int)(i % Wo), dy = (int)((i / Wo) % Ho);
        const uint8_t* S = img + (i / ((size_t)Wo * Ho)) * (size_t)H * W;
        const int sy0 = dy * ky, sx0 = dx * kx;
        if (sy0 >= H || sx0 >= W) { out[i] = 0; continue; }
        int sum = 0, count = 0;
        for (int sy = 0; sy < ky && sy0 + sy < H; ++sy)
            for (int sx = 0; sx < kx && sx0 + sx < W; ++sx) { sum += S[(size_t)(sy0 + sy) * W + sx0 + sx]; ++count; }
        const bool full = sy0 + ky <= H && dx < wfull;
        if (full) out[i] = (kx == 2 && ky == 2) ? (uint8_t)((sum + 2) >> 2) : sat_u8((float)sum * scale);
        else out[i] = sat_u8((float)sum / (float)count);
    }
}

__global__ __launch_bounds__(256) void area_taps_kernel(const uint8_t* __restrict__ img, uint8_t* __restrict__ out, int N, int H, int W, int Ho, int Wo,
                                                         const int* __restrict__ xptr, const int* __restrict__ xsi, const float* __restrict__ xal,
                                                         const int* __restrict__ yptr, const int* __restrict__ ysi, const float* __restrict__ yal) {
#pragma clang fp contract(off)                               // products and sums rounded one by one, like the float loops of ResizeArea
    const size_t total = (size_t)N * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int dx = (int)(i % Wo), dy = (int)((i / Wo) % Ho);
        const uint8_t* S = img + (i / ((size_t)Wo * Ho)) * (size_t)H * W;
        float sum = 0.f;
        for (int j = yptr[dy]; j < yptr[dy + 1]; ++j) {
            const uint8_t* row = S + (size_t)ysi[j] * W;
            float buf = 0.f;
            for (int k = xptr[dx]; k < xptr[dx + 1]; ++k) { const float t = (float)row[xsi[k]] * xal[k]; buf = buf + t; }
            const float term = yal[j] * buf;
            sum = j == yptr[dy] ? term : sum + term;
        }
        out[i] = sat_u8(sum);
    }
}
}  // namespace

extern "C" {

int fb_area_resize_size(int n, double f) { return (int)std::lrint((double)n * f); }

int fb_area_resize_dev(fb_ctx* ctx, const uint8_t* img, int N, int H, int W, double fx, double fy, uint8_t* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && H > 0 && W > 0 && fx > 0 && fx <= 1 && fy > 0 && fy <= 1);
    const int Ho = fb_area_resize_size(H, fy), Wo = fb_area_resize_size(W, fx);
    FB_CHECK_ARG(ctx, Ho > 0 && Wo > 0);
    if (N == 0) return FB_OK;
    FB_CHECK_ARG(ctx, img && out);
    FB_PROF(ctx, "area_resize");
    const double sx = 1.0 / fx, sy = 1.0 / fy;
    const int kx = (int)std::lrint(sx), ky = (int)std::lrint(sy);
    const size_t total = (size_t)N * Ho * Wo;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
    if (std::fabs(sx - kx) < DBL_EPSILON && std::fabs(sy - ky) < DBL_EPSILON) {
        hipLaunchKernelGGL(area_fast_kernel, dim3(blocks), dim3(256), 0, ctx->stream, img, out, N, H, W, Ho, Wo, kx, ky);
        FB_HIP(ctx, hipGetLastError());
        return FB_OK;
    }
    const AreaAxis tx = area_axis(W, Wo, sx), ty = area_axis(H, Ho, sy);
    int *d_i = nullptr; float* d_f = nullptr;
    const size_t ni = tx.ptr.size() + tx.si.size() + ty.ptr.size() + ty.si.size(), nf = tx.alpha.size() + ty.alpha.size();
    std::vector<int> hi; hi.reserve(ni);
    hi.insert(hi.end(), tx.ptr.begin(), tx.ptr.end()); hi.insert(hi.end(), tx.si.begin(), tx.si.end());
    hi.insert(hi.end(), ty.ptr.begin(), ty.ptr.end()); hi.insert(hi.end(), ty.si.begin(), ty.si.end());
    std::vector<float> hf(tx.alpha); hf.insert(hf.end(), ty.alpha.begin(), ty.alpha.end());
    int rc = fb_malloc(ctx, sizeof(int) * ni, (void**)&d_i);
    if (!rc) rc = fb_malloc(ctx, sizeof(float) * nf, (void**)&d_f);
    if (!rc) rc = fb_copy_h2d(ctx, d_i, hi.data(), sizeof(int) * ni);
    if (!rc) rc = fb_copy_h2d(ctx, d_f, hf.data(), sizeof(float) * nf);
    if (!rc) {
        const int* xptr = d_i; const int* xsi = xptr + tx.ptr.size(); const int* yptr = xsi + tx.si.size(); const int* ysi = yptr + ty.ptr.size();
        hipLaunchKernelGGL(area_taps_kernel, dim3(blocks), dim3(256), 0, ctx->stream, img, out, N, H, W, Ho, Wo, xptr, xsi, (const float*)d_f, yptr, ysi,
                           (const float*)d_f + tx.alpha.size());
        if (hipGetLastError() != hipSuccess) rc = fb_fail(ctx, FB_ERR_HIP, "area_taps_kernel launch failed");
    }
    if (d_i) fb_free(ctx, d_i);                             // (fb_free waits for the stream)
    if (d_f) fb_free(ctx, d_f);
    return rc;
}

int fb_area_resize(fb_ctx* ctx, const uint8_t* img, int N, int H, int W, double fx, double fy, uint8_t* out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, N >= 0 && H > 0 && W > 0 && fx > 0 && fx <= 1 && fy > 0 && fy <= 1);
    if (N == 0) return FB_OK;
    const size_t bi = (size_t)N * H * W, bo = (size_t)N * fb_area_resize_size(H, fy) * fb_area_resize_size(W, fx);
    FB_CHECK_ARG(ctx, img && out && bo > 0);
    void *din = nullptr, *dout = nullptr;
    int rc = fb_malloc(ctx, bi, &din);
    if (!rc) rc = fb_malloc(ctx, bo, &dout);
    if (!rc) rc = fb_copy_h2d(ctx, din, img, bi);
    if (!rc) rc = fb_area_resize_dev(ctx, (const uint8_t*)din, N, H, W, fx, fy, (uint8_t*)dout);
    if (!rc) rc = fb_copy_d2h(ctx, out, dout, bo);
    if (din) fb_free(ctx, din);
    if (dout) fb_free(ctx, dout);
    return rc;
}

}  // extern "C"
