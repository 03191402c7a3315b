// Synthetic overlap strips for the benchmark / scale tests (BASELINE.json configs 2 and 4):
// band-limited value noise cut at a known integer offset per pair, plus per-strip sensor noise.
// Deterministic in (seed, pair): the host can regenerate any pair.  Not a reference function.
#include "fb_common.h"

#include <algorithm>

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t hash3(int x, int y, uint32_t s) {
    return mix32((uint32_t)x * 0x9e3779b1u ^ mix32((uint32_t)y * 0x85ebca77u ^ s));
}
__device__ __forceinline__ float lattice(int x, int y, uint32_t s) {
    return (float)(hash3(x, y, s) >> 8) * (2.0f / 16777216.0f) - 1.0f;
}
__device__ __forceinline__ float vnoise(float X, float Y, float cell, uint32_t s) {
    const float fx = X / cell, fy = Y / cell;
    const float flx = floorf(fx), fly = floorf(fy);
    const int ix = (int)flx, iy = (int)fly;
    float tx = fx - flx, ty = fy - fly;
    tx = tx * tx * (3.f - 2.f * tx);
    ty = ty * ty * (3.f - 2.f * ty);
    const float a = lattice(ix, iy, s), b = lattice(ix + 1, iy, s);
    const float c = lattice(ix, iy + 1, s), d = lattice(ix + 1, iy + 1, s);
    return (a + (b - a) * tx) + ((c + (d - c) * tx) - (a + (b - a) * tx)) * ty;
}

__global__ void synth_shifts_kernel(int P, int pair0, uint32_t seed, int max_shift, int step, int* __restrict__ shifts) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const uint32_t h0 = hash3(pair0 + p, 17, seed), h1 = hash3(pair0 + p, 91, seed);
    const int half = max_shift / step;
    const int span = 2 * half + 1;
    shifts[2 * p] = step * ((int)(h0 % (uint32_t)span) - half);        // sx
    shifts[2 * p + 1] = step * ((int)(h1 % (uint32_t)span) - half);    // sy
}

// strip1(x, y) = texture(x + sx + wx(x, y), y + sy + wy(x, y)), strip0(x, y) = texture(x, y); both + independent noise.
// (wx, wy) = smooth sub-pixel warp of amplitude `warp` px ("smooth sub-pixel warp <= 0.4 px").
__global__ void synth_strips_kernel(int P, int pair0, int H, int W, uint32_t seed, const int* __restrict__ shifts, float warp,
                                    uint8_t* __restrict__ s0, uint8_t* __restrict__ s1) {
    const size_t per = (size_t)H * W;
    const size_t total = 2 * (size_t)P * per;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int sel = (int)(i / ((size_t)P * per));
        const size_t r = i - (size_t)sel * P * per;
        const int p = (int)(r / per);
        const int pix = (int)(r - (size_t)p * per);
        const int y = pix / W, x = pix - y * W;
        const uint32_t ps = mix32(seed ^ (uint32_t)(pair0 + p) * 0x27d4eb2fu);
        float X = (float)(x + (sel ? shifts[2 * p] : 0)) + 1000.0f;
        float Y = (float)(y + (sel ? shifts[2 * p + 1] : 0)) + 1000.0f;
        if (sel && warp != 0.f) {
            const float ph = (float)(ps & 1023u) * 0.00613592f;                 // per-pair phase
            X += warp * __sinf(6.2831853f * (float)y / (float)max(H, W) * 1.5f + ph) * __cosf(3.1415927f * (float)x / (float)max(H, W));
            Y += warp * __cosf(6.2831853f * (float)x / (float)max(H, W) * 1.2f + 0.7f * ph);
        }
        const float t = 0.62f * vnoise(X, Y, 3.1f, ps) + 0.30f * vnoise(X, Y, 9.7f, ps + 1u) + 0.25f * vnoise(X, Y, 41.0f, ps + 2u);
        const float nz = (float)(hash3(x, y, ps ^ (sel ? 0xa5a5a5a5u : 0x5a5a5a5au)) >> 8) * (1.0f / 16777216.0f) - 0.5f;
        float v = 128.0f + 70.0f * t + 10.0f * nz;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        (sel ? s1 : s0)[r] = (uint8_t)(v + 0.5f);
    }
}

// plain streaming kernel: every element is read from NR streams and written to NW streams.  Every workgroup owns ONE contiguous
// run of every stream: that is the form that reaches the best store rate on this device (5.6-6.0 TB/s against 4.2-4.8 with
// the runs of a grid-strided loop interleaved over the XCDs: tools/hbm_store_probe.hip, profiles/r06g_hbm_store_probe.txt) --
// the calibration is the rate to beat, so it uses the better form (until round 6 it used the grid-strided one).
template <int NR, int NW>
__global__ __launch_bounds__(256) void hbm_mix_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t end = min(n, (size_t)(blockIdx.x + 1) * per);
    for (size_t i = (size_t)blockIdx.x * per + threadIdx.x; i < end; i += 256) {
        float4 acc = make_float4(1.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < NR; ++r) { const float4 v = src[(size_t)r * n + i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
        for (int w = 0; w < NW; ++w) dst[(size_t)w * n + i] = make_float4(acc.x + w, acc.y, acc.z, acc.w);
        if (NW == 0 && acc.x == -12345.f) dst[i] = acc;         // keeps the loads of a read-only run alive
    }
}

}  // namespace

extern "C" {

// Calibration of the HBM roofline: the rate a plain streaming kernel reaches on this device at a given mix of read and
// write streams (nr : nw in {1:0, 2:1, 1:1, 1:2, 0:1}), 1 GiB per stream.  The data-sheet 8 TB/s is a read figure; the
// NCC passes write as much as or twice what they read.  Not a reference function (bench.py reports it beside `roofline`).
int fb_hbm_probe(fb_ctx* ctx, int nr, int nw, double* gbs) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, gbs && nr >= 0 && nr <= 2 && nw >= 0 && nw <= 2 && nr + nw > 0);
    const size_t n = (size_t)64 << 20;
    float4 *src = nullptr, *dst = nullptr;
    FB_HIP(ctx, hipMalloc((void**)&src, n * 16 * 2));
    if (hipMalloc((void**)&dst, n * 16 * 2) != hipSuccess) { hipFree(src); return fb_fail(ctx, FB_ERR_NOMEM, "fb_hbm_probe: hipMalloc"); }
    hipMemsetAsync(src, 0, n * 16 * 2, ctx->stream);
    hipMemsetAsync(dst, 0, n * 16 * 2, ctx->stream);
    auto launch = [&]() {
        const dim3 grid(256 * 16), block(256);
        if (nr == 1 && nw == 0) hipLaunchKernelGGL((hbm_mix_kernel<1, 0>), grid, block, 0, ctx->stream, src, dst, n);
        else if (nr == 2 && nw == 0) hipLaunchKernelGGL((hbm_mix_kernel<2, 0>), grid, block, 0, ctx->stream, src, dst, n);
        else if (nr == 2 && nw == 1) hipLaunchKernelGGL((hbm_mix_kernel<2, 1>), grid, block, 0, ctx->stream, src, dst, n);
        else if (nr == 1 && nw == 1) hipLaunchKernelGGL((hbm_mix_kernel<1, 1>), grid, block, 0, ctx->stream, src, dst, n);
        else if (nr == 1 && nw == 2) hipLaunchKernelGGL((hbm_mix_kernel<1, 2>), grid, block, 0, ctx->stream, src, dst, n);
        else if (nr == 2 && nw == 2) hipLaunchKernelGGL((hbm_mix_kernel<2, 2>), grid, block, 0, ctx->stream, src, dst, n);
        else if (nr == 0 && nw == 1) hipLaunchKernelGGL((hbm_mix_kernel<0, 1>), grid, block, 0, ctx->stream, src, dst, n);
        else hipLaunchKernelGGL((hbm_mix_kernel<0, 2>), grid, block, 0, ctx->stream, src, dst, n);
    };
    launch();
    hipEventRecord(ctx->t0, ctx->stream);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(ctx->t1, ctx->stream);
    hipError_t e = hipEventSynchronize(ctx->t1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, ctx->t0, ctx->t1);
    hipFree(src); hipFree(dst);
    if (e != hipSuccess) return fb_fail(ctx, FB_ERR_HIP, "fb_hbm_probe: %s", hipGetErrorString(e));
    *gbs = (double)reps * n * 16.0 * (nr + nw) / (ms * 1e-3) / 1e9;
    return FB_OK;
}


int fb_synth_strips_dev(fb_ctx* ctx, int P, int pair0, int H, int W, uint32_t seed, int max_shift, int shift_step, float warp,
                        uint8_t* strips0, uint8_t* strips1, int* shifts_dev) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, P > 0 && H > 0 && W > 0 && max_shift >= 0 && shift_step >= 1 && strips0 && strips1 && shifts_dev);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(synth_shifts_kernel, dim3(fb_cdiv(P, 256)), dim3(256), 0, ctx->stream, P, pair0, seed, max_shift, shift_step, shifts_dev);
    const size_t total = 2 * (size_t)P * H * W;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(synth_strips_kernel, dim3(blocks), dim3(256), 0, ctx->stream, P, pair0, H, W, seed, shifts_dev, warp, strips0, strips1);
    FB_HIP(ctx, hipGetLastError());
    return FB_OK;
}

}  // extern "C"
