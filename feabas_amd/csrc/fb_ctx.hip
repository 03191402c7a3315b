// Context, device memory, stopwatch and per-kernel profile of libfeabas_hip.so.
#include "fb_common.h"

#include <algorithm>
#include <atomic>
#include <thread>

#include <cstdlib>

int fb_fail(fb_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    // the message belongs to the calling thread: several host threads may share a context (or use one each), and
    // fb_last_error is called by the thread whose call failed
    fb_tls_err() = buf;
    (void)ctx;
    return code;
}

std::string& fb_tls_err() {
    static thread_local std::string e;
    return e;
}

// rocFFT's setup / cleanup are process-global: counted over the contexts that initialised it, cleaned up by the last
std::atomic<int> g_rocfft_users{0};
int fb_rocfft_acquire() {
    if (g_rocfft_users.fetch_add(1) == 0) return rocfft_setup() == rocfft_status_success ? 0 : -1;
    return 0;
}
void fb_rocfft_release() {
    if (g_rocfft_users.fetch_sub(1) == 1) rocfft_cleanup();
}

int fb_arena_reserve(fb_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->arena_bytes) return FB_OK;
    if (ctx->arena) {
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FB_HIP(ctx, hipFree(ctx->arena));
        ctx->arena = nullptr;
        ctx->arena_bytes = 0;
    }
    size_t want = bytes + (bytes >> 3);
    hipError_t e = hipMalloc(&ctx->arena, want);
    if (e != hipSuccess) {
        want = bytes;
        e = hipMalloc(&ctx->arena, want);
    }
    if (e != hipSuccess) return fb_fail(ctx, FB_ERR_NOMEM, "arena hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    ctx->arena_bytes = want;
    return FB_OK;
}

fb_prof_scope::fb_prof_scope(fb_ctx* c, const char* name, double bytes) : ctx(c) {
    if (!ctx->prof_on) return;
    for (size_t i = 0; i < ctx->prof.size(); ++i)
        if (ctx->prof[i].name == name) idx = (int)i;
    if (idx < 0) {
        ctx->prof.emplace_back();
        ctx->prof.back().name = name;
        idx = (int)ctx->prof.size() - 1;
    }
    ctx->prof[idx].total_bytes += bytes;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, ctx->stream);
}

fb_prof_scope::~fb_prof_scope() {
    if (idx < 0) return;
    hipEventRecord(e1, ctx->stream);
    ctx->prof[idx].pending.emplace_back(e0, e1);
}

static void prof_drain(fb_ctx* ctx) {
    for (auto& p : ctx->prof) {
        for (auto& ev : p.pending) {
            float ms = 0.f;
            hipEventSynchronize(ev.second);
            if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
                p.total_ms += ms;
                p.launches += 1;
            }
            hipEventDestroy(ev.first);
            hipEventDestroy(ev.second);
        }
        p.pending.clear();
    }
}

namespace {
constexpr size_t kStageChunk = (size_t)4 << 20;        // bytes per slot of the staging ring
constexpr size_t kStageDirect = (size_t)32 << 10;      // smaller copies go straight through hipMemcpyAsync (the runtime stages them itself)

int stage_init(fb_ctx* ctx) {
    if (ctx->stage[0]) return FB_OK;
    for (int k = 0; k < 2; ++k) {
        FB_HIP(ctx, hipHostMalloc(&ctx->stage[k], kStageChunk, hipHostMallocDefault));
        FB_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
    }
    return FB_OK;
}
}  // namespace

namespace {
// page-locked host memory (fb_host_alloc, hipHostMalloc / hipHostRegister of the caller) goes straight to the DMA engine
bool host_pinned(const void* p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}
}  // namespace

int fb_copy_h2d(fb_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return FB_OK;
    if (bytes <= kStageDirect) {
        // (the runtime copies a small pageable source into its own staging area before it returns; a small PAGE-LOCKED source
        // is read by the DMA engine later: wait for it below)
        const bool pinned = host_pinned(src);
        FB_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        if (pinned) FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FB_OK;
    }
    if (host_pinned(src)) {
        // straight to the DMA engine -- which reads the source asynchronously: the contract of this function (the caller may reuse
        // `src` on return) needs the copy to have completed
        FB_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FB_OK;
    }
    std::lock_guard<std::mutex> lk(ctx->stage_mtx);
    int rc = stage_init(ctx);
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += kStageChunk) {
        const size_t n = std::min(kStageChunk, bytes - off);
        const int k = ctx->stage_next;
        ctx->stage_next ^= 1;
        FB_HIP(ctx, hipEventSynchronize(ctx->stage_ev[k]));            // the slot's previous DMA has read it
        memcpy(ctx->stage[k], (const char*)src + off, n);
        FB_HIP(ctx, hipMemcpyAsync((char*)dst + off, ctx->stage[k], n, hipMemcpyHostToDevice, ctx->stream));
        FB_HIP(ctx, hipEventRecord(ctx->stage_ev[k], ctx->stream));
    }
    return FB_OK;
}

int fb_copy_d2h(fb_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return FB_OK;
    if (bytes <= kStageDirect || host_pinned(dst)) {
        FB_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FB_OK;
    }
    std::lock_guard<std::mutex> lk(ctx->stage_mtx);
    int rc = stage_init(ctx);
    if (rc) return rc;
    // chunk c + 1 is on its way while chunk c is copied out of its slot
    size_t pend_off = 0, pend_n = 0; int pend_k = -1;
    for (size_t off = 0; off < bytes; off += kStageChunk) {
        const size_t n = std::min(kStageChunk, bytes - off);
        const int k = ctx->stage_next;
        ctx->stage_next ^= 1;
        FB_HIP(ctx, hipEventSynchronize(ctx->stage_ev[k]));
        FB_HIP(ctx, hipMemcpyAsync(ctx->stage[k], (const char*)src + off, n, hipMemcpyDeviceToHost, ctx->stream));
        FB_HIP(ctx, hipEventRecord(ctx->stage_ev[k], ctx->stream));
        if (pend_k >= 0) {
            FB_HIP(ctx, hipEventSynchronize(ctx->stage_ev[pend_k]));
            memcpy((char*)dst + pend_off, ctx->stage[pend_k], pend_n);
        }
        pend_off = off; pend_n = n; pend_k = k;
    }
    FB_HIP(ctx, hipEventSynchronize(ctx->stage_ev[pend_k]));
    memcpy((char*)dst + pend_off, ctx->stage[pend_k], pend_n);
    return FB_OK;
}

extern "C" {

const char* fb_version(void) { return "feabas_hip 0.1 (gfx950)"; }

fb_ctx* fb_create(int device_id) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return nullptr;
    if (device_id < 0 || device_id >= ndev) return nullptr;
    if (hipSetDevice(device_id) != hipSuccess) return nullptr;
    fb_ctx* ctx = new fb_ctx();
    ctx->device = device_id;
    if (const char* e = getenv("FEABAS_HIP_ROCFFT")) ctx->use_rocfft = atoi(e) != 0;
    if (const char* e = getenv("FEABAS_HIP_DOG_EXACT")) ctx->dog_exact = atoi(e) != 0;
    if (const char* e = getenv("FEABAS_HIP_DOG_TILES")) ctx->dog_tiles = atoi(e) != 0;
    if (const char* e = getenv("FEABAS_HIP_MALLOC_CACHE_MB")) ctx->free_limit = (size_t)std::max(0L, atol(e)) << 20;
    if (const char* e = getenv("FEABAS_HIP_PCG_GRAPH_NB")) ctx->pcg_graph_max_nb = std::max(0, atoi(e));
    if (const char* e = getenv("FEABAS_HIP_NCC_ARENA_MB")) { const long mb = atol(e); if (mb >= 16) ctx->ncc_arena_limit = (size_t)mb << 20; }   // sub-batch size of the streaming NCC class (A/B: small chunks keep T and V in the 256 MiB Infinity Cache)
    if (hipGetDeviceProperties(&ctx->prop, device_id) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->t0) != hipSuccess || hipEventCreate(&ctx->t1) != hipSuccess || hipMalloc(&ctx->small, 64 << 10) != hipSuccess) {
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void fb_destroy(fb_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->match_systems)
        for (fb_system* sy : kv.second) fb_sys_destroy(ctx, sy);
    ctx->match_systems.clear();
    prof_drain(ctx);
    for (auto& kv : ctx->plans) {
        if (kv.second.fwd) rocfft_plan_destroy(kv.second.fwd);
        if (kv.second.inv) rocfft_plan_destroy(kv.second.inv);
    }
    if (ctx->fft_info) rocfft_execution_info_destroy(ctx->fft_info);
    if (ctx->fft_work) hipFree(ctx->fft_work);
    if (ctx->arena) hipFree(ctx->arena);
    for (auto& p : ctx->owned) hipFree(p.first);
    for (auto& p : ctx->free_blocks) hipFree(p.first);
    if (ctx->small) hipFree(ctx->small);
    if (ctx->rocfft_ready) fb_rocfft_release();
    hipEventDestroy(ctx->t0);
    hipEventDestroy(ctx->t1);
    for (int k = 0; k < 2; ++k) {
        if (ctx->stage[k]) hipHostFree(ctx->stage[k]);
        if (ctx->stage_ev[k]) hipEventDestroy(ctx->stage_ev[k]);
    }
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* fb_last_error(fb_ctx* ctx) { (void)ctx; return fb_tls_err().c_str(); }

int fb_sync(fb_ctx* ctx) {
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FB_OK;
}

void* fb_stream(fb_ctx* ctx) { return (void*)ctx->stream; }

int fb_device_info(fb_ctx* ctx, char* name, int name_len, int* num_cu, size_t* hbm_bytes) {
    if (name && name_len > 0) {
        snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    }
    if (num_cu) *num_cu = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = ctx->prop.totalGlobalMem;
    return FB_OK;
}

int fb_malloc(fb_ctx* ctx, size_t bytes, void** dptr) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, dptr != nullptr);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t need = bytes ? (bytes + 255) / 256 * 256 : 256;
    // best fit among the blocks fb_free kept: at least `need`, at most twice that (+ 1 MiB)
    int best = -1;
    for (size_t i = 0; i < ctx->free_blocks.size(); ++i) {
        const size_t cap = ctx->free_blocks[i].second;
        if (cap >= need && cap <= 2 * need + ((size_t)1 << 20) && (best < 0 || cap < ctx->free_blocks[best].second)) best = (int)i;
    }
    if (best >= 0) {
        const auto blk = ctx->free_blocks[best];
        ctx->free_blocks.erase(ctx->free_blocks.begin() + best);
        ctx->free_bytes -= blk.second;
        ctx->owned.push_back(blk);
        *dptr = blk.first;
        return FB_OK;
    }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess && !ctx->free_blocks.empty()) {          // out of memory with blocks in the cache: give them back and retry
        (void)hipGetLastError();
        hipStreamSynchronize(ctx->stream);
        for (auto& b : ctx->free_blocks) hipFree(b.first);
        ctx->free_blocks.clear(); ctx->free_bytes = 0;
        e = hipMalloc(&p, need);
    }
    if (e != hipSuccess) return fb_fail(ctx, FB_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    ctx->owned.emplace_back(p, need);
    *dptr = p;
    return FB_OK;
}

// page-locked host staging memory: fb_memcpy_h2d / _d2h from it run at the link rate instead of through the driver's
// bounce buffers (the strips of a tile pair are 4 MB; pageable copies cap the ingest well below the kernels' rate)
int fb_host_alloc(fb_ctx* ctx, size_t bytes, void** hptr) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, hptr != nullptr);
    FB_HIP(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) return fb_fail(ctx, FB_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    *hptr = p;
    return FB_OK;
}

int fb_host_free(fb_ctx* ctx, void* hptr) {
    FB_LOCK(ctx);
    if (!hptr) return FB_OK;
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FB_HIP(ctx, hipHostFree(hptr));
    return FB_OK;
}

// Gather n row-major uint8 images (strips cropped on the host) into the slots of a staging stack [n][H][W]: image k
// (hs[k] x ws[k], row pitch pitches[k] bytes) goes to the top-left corner of slot k.  Plain host memcpy on `threads`
// std::threads (the caller's interpreter lock is released for the whole call): packing 4 MB per pair through the
// interpreter was the largest host cost of the PCIe-inclusive path.
int fb_host_pack2d(fb_ctx* ctx, uint8_t* dst, int n, int H, int W, const void* const* srcs, const int* hs, const int* ws,
                   const int64_t* pitches, int threads) {
    FB_CHECK_ARG(ctx, dst && n >= 0 && H > 0 && W > 0 && (n == 0 || (srcs && hs && ws && pitches)));
    for (int k = 0; k < n; ++k) FB_CHECK_ARG(ctx, srcs[k] && hs[k] > 0 && ws[k] > 0 && hs[k] <= H && ws[k] <= W && pitches[k] >= ws[k]);
    const int T = std::max(1, std::min(threads, 16));
    auto work = [&](int t) {
        // rows are dealt in contiguous blocks so that every thread streams its own part of the stack
        int64_t total = 0;
        for (int k = 0; k < n; ++k) total += hs[k];
        const int64_t lo = total * t / T, hi = total * (t + 1) / T;
        int64_t at = 0;
        for (int k = 0; k < n && at < hi; ++k) {
            const int64_t r0 = std::max<int64_t>(lo - at, 0), r1 = std::min<int64_t>(hi - at, hs[k]);
            const uint8_t* s = (const uint8_t*)srcs[k];
            uint8_t* d = dst + (size_t)k * H * W;
            if (pitches[k] == ws[k] && ws[k] == W && r1 > r0) std::memcpy(d + (size_t)r0 * W, s + (size_t)r0 * W, (size_t)(r1 - r0) * W);
            else
                for (int64_t r = r0; r < r1; ++r) std::memcpy(d + (size_t)r * W, s + (size_t)r * pitches[k], (size_t)ws[k]);
            at += hs[k];
        }
    };
    if (T == 1) { work(0); return FB_OK; }
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t) pool.emplace_back(work, t);
    for (auto& th : pool) th.join();
    return FB_OK;
}

int fb_free(fb_ctx* ctx, void* dptr) {
    FB_LOCK(ctx);
    if (!dptr) return FB_OK;
    for (size_t i = 0; i < ctx->owned.size(); ++i) {
        if (ctx->owned[i].first == dptr) {
            const auto blk = ctx->owned[i];
            ctx->owned.erase(ctx->owned.begin() + i);
            // (from here on the block is in neither list: every error return below first puts it where it can be found again)
            {
                const hipError_t se = hipStreamSynchronize(ctx->stream);       // nothing enqueued on this context still uses the block
                if (se != hipSuccess) { ctx->free_blocks.push_back(blk); ctx->free_bytes += blk.second; return fb_fail(ctx, FB_ERR_HIP, "fb_free: %s", hipGetErrorString(se)); }
            }
            if (blk.second > ctx->free_limit) { FB_HIP(ctx, hipFree(dptr)); return FB_OK; }
            // the block just freed is the one most likely to be asked for again: when the cache is full the OLDEST kept blocks make
            // room (a phase that worked on a few multi-GB buffers otherwise leaves the cache full of blocks nobody fits into, and
            // every temporary of the next phase pays hipMalloc / hipFree: +10-20 ms per section pair at the end of bench.py)
            while (ctx->free_bytes + blk.second > ctx->free_limit && !ctx->free_blocks.empty()) {
                const auto old = ctx->free_blocks.front();
                ctx->free_blocks.erase(ctx->free_blocks.begin());
                ctx->free_bytes -= old.second;
                const hipError_t fe = hipFree(old.first);
                if (fe != hipSuccess) {
                    // the evicted block is gone from the books either way; the block being freed still goes into the cache
                    ctx->free_blocks.push_back(blk);
                    ctx->free_bytes += blk.second;
                    return fb_fail(ctx, FB_ERR_HIP, "fb_free: eviction: %s", hipGetErrorString(fe));
                }
            }
            ctx->free_blocks.push_back(blk);
            ctx->free_bytes += blk.second;
            return FB_OK;
        }
    }
    return fb_fail(ctx, FB_ERR_ARG, "fb_free: pointer not owned by this context");
}

int fb_memcpy_h2d(fb_ctx* ctx, void* dst, const void* src, size_t bytes) {
    int rc = fb_copy_h2d(ctx, dst, src, bytes);
    if (rc) return rc;
    FB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FB_OK;
}

int fb_memcpy_d2h(fb_ctx* ctx, void* dst, const void* src, size_t bytes) {
    return fb_copy_d2h(ctx, dst, src, bytes);
}

int fb_memcpy_d2d(fb_ctx* ctx, void* dst, const void* src, size_t bytes) {
    FB_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return FB_OK;
}

int fb_memcpy2d_d2d(fb_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t rows) {
    FB_CHECK_ARG(ctx, dst && src && dpitch >= width_bytes && spitch >= width_bytes);
    if (width_bytes == 0 || rows == 0) return FB_OK;
    FB_HIP(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyDeviceToDevice, ctx->stream));
    return FB_OK;
}

int fb_memset(fb_ctx* ctx, void* dst, int value, size_t bytes) {
    FB_HIP(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return FB_OK;
}

int fb_timer_start(fb_ctx* ctx) {
    FB_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
    return FB_OK;
}

int fb_timer_stop(fb_ctx* ctx, float* ms) {
    FB_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
    FB_HIP(ctx, hipEventSynchronize(ctx->t1));
    FB_HIP(ctx, hipEventElapsedTime(ms, ctx->t0, ctx->t1));
    return FB_OK;
}

int fb_prof_enable(fb_ctx* ctx, int on) {
    FB_LOCK(ctx);
    if (!on) prof_drain(ctx);
    ctx->prof_on = on != 0;
    return FB_OK;
}

int fb_prof_reset(fb_ctx* ctx) {
    FB_LOCK(ctx);
    prof_drain(ctx);
    ctx->prof.clear();
    return FB_OK;
}

int fb_prof_count(fb_ctx* ctx) {
    FB_LOCK(ctx);
    prof_drain(ctx);
    return (int)ctx->prof.size();
}

int fb_prof_get(fb_ctx* ctx, int index, char* name, int name_len, int* launches, double* total_ms, double* total_bytes) {
    FB_LOCK(ctx);
    prof_drain(ctx);
    FB_CHECK_ARG(ctx, index >= 0 && index < (int)ctx->prof.size());
    const fb_prof_entry& p = ctx->prof[index];
    if (name && name_len > 0) snprintf(name, name_len, "%s", p.name.c_str());
    if (launches) *launches = p.launches;
    if (total_ms) *total_ms = p.total_ms;
    if (total_bytes) *total_bytes = p.total_bytes;
    return FB_OK;
}

}  // extern "C"
