// Internal definitions shared by the HIP translation units of libfeabas_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "feabas_hip.h"

struct fb_prof_entry {
    std::string name;
    int launches = 0;
    double total_ms = 0.0;
    double total_bytes = 0.0;    // algorithmic HBM bytes of the launches
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct fb_fft_plan {
    rocfft_plan fwd = nullptr;
    rocfft_plan inv = nullptr;
    size_t work_bytes = 0;
};

// host threads a pure-host helper may use for one call (FEABAS_HIP_HOST_THREADS overrides; 1 = none beside the caller)
inline int fb_host_threads(int cap) {
    static const int env = [] { const char* e = std::getenv("FEABAS_HIP_HOST_THREADS"); return e ? std::atoi(e) : 0; }();
    return env > 0 ? std::min(env, std::max(cap, 1)) : cap;
}

struct fb_ctx {
    // set by fb_match_strips (under the context lock) around its fb_pairs_*_bary calls: the matches it hands over couple the
    // three vertices of ONE grid triangle (locate_grid / fb_deformed_locate), so fb_sys_update_links' membership test of every
    // coupled pair in the pattern -- 36 binary searches per match, 1.5 ms per 12 k matches -- is skipped as in fb_pairs_relax
    int trusted_links = 0;
    // every compute entry point holds this while it enqueues: calls from several host threads interleave at
    // call granularity only (the scratch arena and the profile are shared)
    std::recursive_mutex mtx;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    hipDeviceProp_t prop;
    // scratch arena (grow-only) for the NCC streaming class
    void* arena = nullptr;
    size_t arena_bytes = 0;
    // rocFFT plans keyed by (Fh, Fw, batch)
    std::map<std::tuple<int, int, int>, fb_fft_plan> plans;
    rocfft_execution_info fft_info = nullptr;
    void* fft_work = nullptr;
    size_t fft_work_bytes = 0;
    bool rocfft_ready = false;
    // surfaces of the last streaming-class call (debug aid)
    int last_Fh = 0, last_Fw = 0, last_N = 0;
    const float* last_C = nullptr;
    const float* last_Cm = nullptr;
    // stopwatch + per-kernel profile
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool prof_on = false;
    std::vector<fb_prof_entry> prof;
    std::vector<std::pair<void*, size_t>> owned;        // fb_malloc'ed blocks (pointer, capacity)
    // blocks handed back by fb_free are kept for the next fb_malloc of a similar size (hipMalloc / hipFree cost 0.1-0.3 ms each
    // and hipFree drains the device; the Python surface allocates its temporaries per call), up to free_limit bytes
    std::vector<std::pair<void*, size_t>> free_blocks;
    size_t free_bytes = 0, free_limit = (size_t)4 << 30;
    void* small = nullptr;      // 64 KiB of device scratch for flags and partial reductions (allocated with the context)
    size_t ncc_arena_limit = (size_t)8 << 30;
    int pcg_graph_max_nb = 1 << 18;        // Jacobi-PCG batches of systems up to this many vertices replay as a graph (0: never)
    bool use_rocfft = false;     // FEABAS_HIP_ROCFFT=1: streaming-class FFTs through rocFFT instead of the hand-written kernels
    bool dog_tiles = false;      // FEABAS_HIP_DOG_TILES=1: the 64 x 64 tile kernel (dog_fast) instead of the streaming one (A/B)
    bool dog_exact = false;      // double-precision tap accumulation (scipy's arithmetic) instead of the float fast path
    const void* dog_img1 = nullptr;    // second image stack of the launch being enqueued (fb_dog_pair_dev); set and cleared under the lock
    int dog_nsplit = 0;
    // device buffers and relaxation systems that outlive a strip matcher (fb_match.hip): matchers of ragged batches come and
    // go with every chunk of a section, their buffers and the symbolic phase of their block-diagonal system do not
    // pinned staging ring of the host <-> device copies (fb_copy_h2d / fb_copy_d2h): a pageable buffer HIP sees for the first
    // time costs 8-20 ms per 4 MB to map (measured, tools/prof_h2d.py), whatever its size afterwards -- and every section, every
    // numpy temporary is such a buffer; through the ring a copy costs one host memcpy + one DMA
    std::mutex stage_mtx;
    void* stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    int stage_next = 0;
    std::vector<std::pair<void*, size_t>> match_pool;
    std::map<std::tuple<int, int, int>, std::vector<fb_system*>> match_systems;
};

int fb_fail(fb_ctx* ctx, int code, const char* fmt, ...);
std::string& fb_tls_err();          // last error message of the calling thread
int fb_rocfft_acquire();            // process-wide reference count around rocfft_setup / rocfft_cleanup
void fb_rocfft_release();

#define FB_HIP(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fb_fail((ctx), FB_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,         \
                           hipGetErrorString(e__));                                                \
    } while (0)

#define FB_FFT(ctx, call)                                                                          \
    do {                                                                                           \
        rocfft_status s__ = (call);                                                                \
        if (s__ != rocfft_status_success)                                                          \
            return fb_fail((ctx), FB_ERR_FFT, "%s:%d %s -> rocfft status %d", __FILE__, __LINE__,  \
                           #call, (int)s__);                                                       \
    } while (0)

#define FB_CHECK_ARG(ctx, cond)                                                                    \
    do {                                                                                           \
        if (!(cond)) return fb_fail((ctx), FB_ERR_ARG, "%s:%d invalid argument: %s", __FILE__,     \
                                    __LINE__, #cond);                                              \
    } while (0)

// grow-only scratch arena
int fb_arena_reserve(fb_ctx* ctx, size_t bytes);
// host -> device on the context's stream through the pinned staging ring: returns when `src` may be reused (the DMA of the
// last chunk may still be in flight, ordered on the stream like a hipMemcpyAsync from pageable memory)
int fb_copy_h2d(fb_ctx* ctx, void* dst, const void* src, size_t bytes);
// device -> host through the ring: returns when `dst` holds the data (everything enqueued on the stream before it has completed)
int fb_copy_d2h(fb_ctx* ctx, void* dst, const void* src, size_t bytes);

// per-kernel profiling scope: records an event pair around a launch when enabled
struct fb_prof_scope {
    fb_ctx* ctx;
    int idx = -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    fb_prof_scope(fb_ctx* c, const char* name, double bytes = 0.0);
    ~fb_prof_scope();
};
#define FB_LOCK(ctx) std::lock_guard<std::recursive_mutex> lock_guard__((ctx)->mtx)
#define FB_PROF(ctx, name) fb_prof_scope prof_scope__((ctx), (name))
#define FB_PROF_B(ctx, name, bytes) fb_prof_scope prof_scope__((ctx), (name), (double)(bytes))

static inline int fb_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// cross-TU entry points
int fb_ncc_small_supported(int Fh, int Fw, int H0, int W0, int H1, int W1, int C);
int fb_ncc_small_launch(fb_ctx* ctx, const float* img0, const float* img1, int N, int H0, int W0,
                        int H1, int W1, int Fh, int Fw, int subpixel, int conf_mode, double* dx,
                        double* dy, float* conf);
// register-resident 75 x 75 form of the on-chip class (fb_ncc_pfa.hip); same arguments as fb_ncc_small_launch_ex
int fb_ncc_pfa_supported(int Fh, int Fw, int conf_mode);
// the FFT shape a block list of at most hmax x wmax pixels asked at Fh x Fw is run at (fb_ncc.hip)
void fb_ncc_launch_shape(fb_ctx* ctx, int Fh, int Fw, int hmax, int wmax, int conf_mode, int* oh, int* ow);
int fb_ncc_pfa_launch(fb_ctx* ctx, const float* img0, const float* img1, int N, int H0, int W0, int H1, int W1, const int* blk,
                      int IH0, int IW0, int IH1, int IW1, int Fh, int Fw, int subpixel, int conf_mode, double* dx, double* dy, float* conf,
                      const double* aff1);
// ---- affine patch gather (the affine-approximated branch of MeshRenderer.crop_field, renderer.py:419-451, 499-511,
// followed by common.render_by_subregions -> cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0), common.py:218-350).
// prm[10] per block: x0, y0 (block origin in the renderer's output space, offset removed), A00, A10, t0, A01, A11, t1
// (source x = X A00 + Y A10 + t0, source y = X A01 + Y A11 + t1), xmin, ymin (integer origin of the sub-image the
// reference hands to cv2.remap: the float32 map is taken relative to it).  cv2.remap quantises the float32 map to
// 1/32 px (INTER_BITS = 5, round half to even) and blends the 4 taps with float32 table weights; taps outside the
// image are the border value 0.
#define FB_AFFINE_STRIDE 10
__device__ __forceinline__ float fb_sample_affine(const float* __restrict__ img, int IH, int IW, const double* __restrict__ prm, int i, int j) {
#pragma clang fp contract(off)
    const double xx = prm[0] + (double)i, yy = prm[1] + (double)j;
    const double mx = (xx * prm[2] + yy * prm[3]) + prm[4];
    const double my = (xx * prm[5] + yy * prm[6]) + prm[7];
    const float fx = (float)(mx - prm[8]), fy = (float)(my - prm[9]);
    const int sx = (int)rintf(fx * 32.0f), sy = (int)rintf(fy * 32.0f);
    const int ix = (sx >> 5) + (int)prm[8], iy = (sy >> 5) + (int)prm[9];
    const float ax = (float)(sx & 31) * (1.0f / 32.0f), ay = (float)(sy & 31) * (1.0f / 32.0f);
    const float w00 = (1.0f - ay) * (1.0f - ax), w01 = (1.0f - ay) * ax, w10 = ay * (1.0f - ax), w11 = ay * ax;
    const bool x0ok = ix >= 0 && ix < IW, x1ok = ix + 1 >= 0 && ix + 1 < IW, y0ok = iy >= 0 && iy < IH, y1ok = iy + 1 >= 0 && iy + 1 < IH;
    const int cx0 = min(max(ix, 0), IW - 1), cx1 = min(max(ix + 1, 0), IW - 1), cy0 = min(max(iy, 0), IH - 1), cy1 = min(max(iy + 1, 0), IH - 1);
    const float v00 = img[(size_t)cy0 * IW + cx0], v01 = img[(size_t)cy0 * IW + cx1];
    const float v10 = img[(size_t)cy1 * IW + cx0], v11 = img[(size_t)cy1 * IW + cx1];
    return ((((y0ok && x0ok) ? v00 : 0.f) * w00 + ((y0ok && x1ok) ? v01 : 0.f) * w01) + ((y1ok && x0ok) ? v10 : 0.f) * w10) + ((y1ok && x1ok) ? v11 : 0.f) * w11;
}

int fb_ncc_small_launch_ex(fb_ctx* ctx, const float* img0, const float* img1, int N, int H0, int W0, int H1, int W1,
                           const int* blk, int IH0, int IW0, int IH1, int IW1, int Fh, int Fw, int subpixel,
                           int conf_mode, double* dx, double* dy, float* conf, const double* aff1 = nullptr);
