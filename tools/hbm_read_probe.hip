// The read pattern of ncc_inv_p2 in isolation: one workgroup per item, an item = one contiguous run of RUN bytes in each of two
// arrays, every thread loads NU float4 per array (index u * NT + tid), nothing else but a dummy reduction.  Which form of it
// reaches the streaming read rate of the device (6.0-6.6 TB/s)?  Variants: threads per workgroup, LDS per workgroup (limits the
// workgroups per CU like the FFT tile does), nontemporal loads, the XCD item mapping.
// build: hipcc -O3 --offload-arch=gfx950 tools/hbm_read_probe.hip -o ab/hbm_read_probe ; run under `timeout -k 10 120`
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));

template <int NT, int NU, bool NTL, bool TWO>
__global__ __launch_bounds__(NT) void k(const v4* __restrict__ A, const v4* __restrict__ B, v4* __restrict__ out, int items, int per8, int spin) {
    extern __shared__ float lds[];
    int w = blockIdx.x;
    if (per8 > 0) { w = (int)(blockIdx.x & 7) * per8 + (int)(blockIdx.x >> 3); if (w >= items) return; }
    const size_t base = (size_t)w * NT * NU;
    v4 a[NU], b[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) a[u] = NTL ? __builtin_nontemporal_load(A + base + u * NT + threadIdx.x) : A[base + u * NT + threadIdx.x];
    if (TWO) {
#pragma unroll
        for (int u = 0; u < NU; ++u) b[u] = NTL ? __builtin_nontemporal_load(B + base + u * NT + threadIdx.x) : B[base + u * NT + threadIdx.x];
    }
    v4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u) { s += a[u]; if (TWO) s += b[u]; }
    // a compute phase after the loads, like the FFT passes: `spin` dependent FMAs
    float t = s.x;
    for (int i = 0; i < spin; ++i) t = __builtin_fmaf(t, 1.0000001f, 0.5f);
    lds[threadIdx.x] = t;
    __syncthreads();
    if (lds[(threadIdx.x + 1) % NT] == -12345.f) out[w] = s;
}

template <int NT, int NU, bool NTL, bool TWO>
double run(const v4* A, const v4* B, v4* out, size_t bytes_per_array, size_t lds, bool xcd, int spin) {
    const int items = (int)(bytes_per_array / ((size_t)NT * NU * 16));
    const int per8 = xcd ? (items + 7) / 8 : 0;
    const int grid = xcd ? 8 * per8 : items;
    CK(hipFuncSetAttribute((const void*)k<NT, NU, NTL, TWO>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NT, NU, NTL, TWO>), dim3(grid), dim3(NT), lds, 0, A, B, out, items, per8, spin);
    CK(hipEventRecord(e0, 0));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<NT, NU, NTL, TWO>), dim3(grid), dim3(NT), lds, 0, A, B, out, items, per8, spin);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
    return (double)reps * (double)items * NT * NU * 16.0 * (TWO ? 2 : 1) / (ms * 1e-3) / 1e9;
}

int main() {
    const size_t bytes = (size_t)2 << 30;       // 2 GiB per array (V0 and V1 of 128 coarse block pairs are 2.1 GB each)
    v4 *A, *B, *out;
    CK(hipMalloc((void**)&A, bytes)); CK(hipMalloc((void**)&B, bytes)); CK(hipMalloc((void**)&out, 64 << 20));
    CK(hipMemset(A, 0, bytes)); CK(hipMemset(B, 0, bytes));
    printf("GB/s of the two-array item read; columns: LDS per workgroup = 1 KB (no limit), 35 KB (4 per CU), 70 KB (2 per CU), 140 KB (1 per CU)\n");
    const size_t L[4] = {1024, 35 * 1024, 70 * 1024, 140 * 1024};
#define ROW(NAME, NT, NU, NTL, TWO, XCD, SPIN) { printf("%-66s", NAME); for (size_t l : L) printf(" %7.0f", run<NT, NU, NTL, TWO>(A, B, out, bytes, l, XCD, SPIN)); printf("\n"); fflush(stdout); }
    ROW("512 thr x 4 float4 x 2 arrays (64 KB item)  plain", 512, 4, false, true, false, 0)
    ROW("512 thr x 4 float4 x 2 arrays              plain, xcd eighths", 512, 4, false, true, true, 0)
    ROW("512 thr x 4 float4 x 2 arrays              nt", 512, 4, true, true, false, 0)
    ROW("512 thr x 4 float4 x 2 arrays              nt, xcd eighths", 512, 4, true, true, true, 0)
    ROW("256 thr x 4 float4 x 2 arrays (32 KB item)  plain", 256, 4, false, true, false, 0)
    ROW("256 thr x 4 float4 x 2 arrays              nt", 256, 4, true, true, false, 0)
    ROW("512 thr x 8 float4 x 2 arrays (128 KB item) plain", 512, 8, false, true, false, 0)
    ROW("512 thr x 8 float4 x 1 array  (64 KB item)  plain", 512, 8, false, false, false, 0)
    ROW("512 thr x 4 float4 x 2 arrays  + 2000 FMA after the loads, plain", 512, 4, false, true, false, 2000)
    ROW("512 thr x 4 float4 x 2 arrays  + 2000 FMA after the loads, nt", 512, 4, true, true, false, 2000)
    ROW("256 thr x 4 float4 x 2 arrays  + 2000 FMA after the loads, plain", 256, 4, false, true, false, 2000)
    ROW("512 thr x 4 float4 x 2 arrays  + 6000 FMA after the loads, plain", 512, 4, false, true, false, 6000)
    return 0;
}
