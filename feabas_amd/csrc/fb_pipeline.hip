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

}  // namespace

extern "C" {

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
