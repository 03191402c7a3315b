// VALU issue-rate probe for gfx950: scalar v_fma_f32 against v_pk_fma_f32 / v_pk_add_f32 at 1, 2, 4, 8 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o ab/valu_rate ; run: ab/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float a, float b) {
    constexpr int NA = 16;
    float s[NA]; f2 p[NA];
    for (int i = 0; i < NA; ++i) { s[i] = threadIdx.x * 0.001f + i; p[i] = (f2){s[i], s[i] + 0.5f}; }
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
                if (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
                if (MODE == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));
                if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
                if (MODE == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(s[i]) : "v"(s[(i + 1) % NA]));
                if (MODE == 6) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "s"(a), "v"(s[(i + 1) % NA]));
                if (MODE == 7) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "s"(a), "v"(s[(i + 1) % NA]));
                if (MODE == 8) asm volatile("v_pk_fma_f32 %0, %2, %1, %0" : "+v"(p[i]) : "s"(a2), "v"(p[(i + 1) % NA]));
                if (MODE == 9) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(a), "v"(s[(i + 1) % NA]));
                if (MODE == 10) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(s[(i + 1) % NA]));
                if (MODE == 11) asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(s[i]) : "s"(a), "v"(s[(i + 1) % NA]));
                if (MODE == 12) asm volatile("v_pk_fma_f32 %0, %2, %1, %0" : "+v"(p[i]) : "v"(a2), "v"(p[(i + 1) % NA]));
            }
        }
    }
    float acc = 0.f;
    for (int i = 0; i < NA; ++i) acc += s[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, float* d, int wg_per_cu) {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = pr.multiProcessorCount * wg_per_cu;       // 256 threads = 4 waves = 1 wave per SIMD per workgroup
    probe<MODE><<<grid, 256>>>(d, 10, 1.0001f, 0.5f);
    hipEventRecord(e0);
    probe<MODE><<<grid, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * wg_per_cu;            // wave-instructions per SIMD
    printf("%-14s %d waves/SIMD: %7.3f ms  %.2f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, wg_per_cu, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 8 * 4 * 4);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", d, w); run<1>("v_pk_fma_f32", d, w); run<2>("v_pk_add_f32", d, w); run<3>("v_add_f32", d, w); run<4>("v_pk_mul_f32", d, w); run<5>("v_mov_b32", d, w);
        run<6>("fma s,v,acc", d, w); run<7>("fmac s,v", d, w); run<8>("pk_fma v,s,acc", d, w); run<9>("fma v,v,acc", d, w); run<10>("fmac v,v", d, w); run<11>("mul s,v", d, w); run<12>("pk_fma v,v,acc", d, w);
    }
    return 0;
}
