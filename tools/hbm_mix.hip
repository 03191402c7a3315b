// Achievable HBM rate of plain streaming kernels at the read : write mixes of the NCC passes (calibration of the roofline:
// the 8 TB/s of the data sheet is a read-mostly figure).  hipcc -O3 --offload-arch=gfx950 tools/hbm_mix.hip -o /tmp/hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NR, int NW>
__global__ __launch_bounds__(256) void mix(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    // every thread reads NR float4 and writes NW float4, all streams unit-stride across the grid
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < NR; ++r) { const float4 v = src[(size_t)r * n + i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
        for (int w = 0; w < NW; ++w) dst[(size_t)w * n + i] = make_float4(acc.x + w, acc.y, acc.z, acc.w);
    }
}

template <int NR, int NW>
void run(const char* name, float4* src, float4* dst, size_t n) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 16;
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((mix<NR, NW>), dim3(grid), dim3(256), 0, 0, src, dst, n);
    hipEventRecord(e0);
    const int reps = 10;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((mix<NR, NW>), dim3(grid), dim3(256), 0, 0, src, dst, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)reps * n * 16.0 * (NR + NW);
    printf("%-28s %7.1f GB/s  (%d reads : %d writes per element, %.1f MB per launch)\n", name, bytes / (ms * 1e-3) / 1e9, NR, NW, n * 16.0 * (NR + NW) / 1e6);
}

int main() {
    const size_t n = (size_t)64 << 20;            // 64 Mi float4 = 1 GiB per stream
    float4 *src, *dst;
    hipMalloc(&src, n * 16 * 2); hipMalloc(&dst, n * 16 * 2);
    hipMemset(src, 0, n * 16 * 2); hipMemset(dst, 0, n * 16 * 2);
    run<1, 0>("read only", src, dst, n);
    run<2, 1>("2 reads : 1 write", src, dst, n / 2 * 1);
    run<1, 1>("1 read : 1 write (copy)", src, dst, n);
    run<1, 2>("1 read : 2 writes (cols)", src, dst, n);
    run<0, 1>("write only", src, dst, n);
    run<0, 2>("write only, 2 streams", src, dst, n);
    return 0;
}
