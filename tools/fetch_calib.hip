// Calibration of the rocprofv3 FETCH_SIZE counter on gfx950 by load width and access pattern (MI355X_MICROARCH.md, HBM: the
// counter reads exactly 1/2 of the bytes of a 16-B-per-lane streaming read; "other access widths are uncalibrated").  Every
// kernel reads a known number of bytes once, from a buffer far larger than the 256 MiB Infinity Cache:
//   hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o f -- /tmp/fetch_calib      (then tools/fetch_calib_report.py)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void calib_b128(const float4* __restrict__ src, float* __restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_b64(const float2* __restrict__ src, float* __restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float2 v = src[i]; acc += v.x + v.y; }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_b32(const float* __restrict__ src, float* __restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void calib_b8(const unsigned char* __restrict__ src, float* __restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += (float)src[i];
    if (acc == 12345.678f) out[0] = acc;
}
// the crop pattern of the on-chip NCC class: a workgroup reads a window of 75 rows x 73 floats out of an image of pitch 510
// floats with dword buffer-style loads (thread = column, 25 rows each), windows side by side: 7 across, 54 down per image
__global__ __launch_bounds__(256) void calib_crop_b32(const float* __restrict__ src, float* __restrict__ out, int nimg) {
    const int per = 7 * 54, img = blockIdx.x / per, b = blockIdx.x % per, bx = b % 7, by = b / 7;
    if (img >= nimg) return;
    const float* p = src + (size_t)img * 4096 * 510 + (size_t)(by * 75) * 510 + bx * 72;
    const int yg = threadIdx.x / 75, x = threadIdx.x % 75;
    float acc = 0.f;
    if (threadIdx.x < 225 && x < 73)
        for (int j = 0; j < 25; ++j) acc += p[(size_t)(yg + 3 * j) * 510 + x];
    if (acc == 12345.678f) out[0] = acc;
}
// the row pattern of the DoG's uint8 reads: a wave reads 64 consecutive bytes x 4 (16 B per lane) of one row
__global__ __launch_bounds__(256) void calib_u8x16(const uint4* __restrict__ src, float* __restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; acc += (float)(v.x ^ v.y ^ v.z ^ v.w); }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void* src; float* out;
    hipMalloc(&src, bytes); hipMalloc(&out, 256);
    hipMemset(src, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_b128, dim3(grid), dim3(256), 0, 0, (const float4*)src, out, bytes / 16);
        hipLaunchKernelGGL(calib_b64, dim3(grid), dim3(256), 0, 0, (const float2*)src, out, bytes / 8);
        hipLaunchKernelGGL(calib_b32, dim3(grid), dim3(256), 0, 0, (const float*)src, out, bytes / 4);
        hipLaunchKernelGGL(calib_b8, dim3(grid), dim3(256), 0, 0, (const unsigned char*)src, out, bytes / 4);       // a quarter of the buffer
        hipLaunchKernelGGL(calib_u8x16, dim3(grid), dim3(256), 0, 0, (const uint4*)src, out, bytes / 16);
        const int nimg = (int)(bytes / ((size_t)4096 * 510 * 4));
        hipLaunchKernelGGL(calib_crop_b32, dim3(nimg * 7 * 54), dim3(256), 0, 0, (const float*)src, out, nimg);
        hipDeviceSynchronize();
    }
    const int nimg = (int)(bytes / ((size_t)4096 * 510 * 4));
    printf("bytes_read calib_b128 %zu\nbytes_read calib_b64 %zu\nbytes_read calib_b32 %zu\nbytes_read calib_b8 %zu\nbytes_read calib_u8x16 %zu\nbytes_read calib_crop_b32 %zu\n",
           bytes, bytes, bytes, bytes / 4, bytes, (size_t)nimg * 7 * 54 * 75 * 73 * 4);
    return 0;
}
