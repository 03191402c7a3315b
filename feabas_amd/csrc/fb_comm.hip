// The exchange steps of a sharded run behind the C ABI: one RCCL communicator per context,
// collectives enqueued on the context's stream, device pointers only.
//   fb_gatherv_dev   : the variable-length match table of every rank -> rank `root` (stitcher.py:144-151 is what
//                      the root then writes), counts known to every rank (fb_allgather_dev of one int64 first)
//   fb_allgather_dev : equal-sized contributions (node displacements of the sections of a rank, aligner.py:588)
//   fb_allreduce_f64_dev : the fused scalar reduction of the coupled-window PCG (aligner.py:510-535, 696-727)
// librccl is bound at run time (dlopen): a single-GPU process never loads the 570 MB library.
#include "fb_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <string>
#include <vector>

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    std::string why;
};

std::mutex g_api_mtx;
RcclApi g_api;

template <typename F>
bool bind(void* h, const char* name, F& fn) {
    fn = reinterpret_cast<F>(dlsym(h, name));
    return fn != nullptr;
}

const RcclApi* rccl_api() {
    std::lock_guard<std::mutex> lk(g_api_mtx);
    if (g_api.handle || !g_api.why.empty()) return &g_api;
    // The RCCL to bind is the one built against the HIP runtime THIS library runs on: a process may hold a second ROCm stack
    // (PyTorch wheels bundle their own libamdhip64 / librccl with the same SONAME), and a communicator of that copy cannot
    // touch this runtime's streams and allocations.  So: the librccl next to our libamdhip64 first, by absolute path, bound
    // to its own symbols (RTLD_DEEPBIND); a bare SONAME -- which the loader may resolve to the other copy -- only last.
    std::vector<std::string> names;
    if (const char* e = getenv("FEABAS_HIP_RCCL")) if (*e) names.push_back(e);
    Dl_info info;
    if (dladdr((const void*)&hipGetDeviceCount, &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t cut = dir.find_last_of('/');
        if (cut != std::string::npos) names.push_back(dir.substr(0, cut) + "/librccl.so.1");
    }
    names.push_back("/opt/rocm/lib/librccl.so.1");
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    void* h = nullptr;
    std::string tried;
    for (const std::string& n : names) {
        h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
        if (h) break;
        tried += n + " ";
    }
    if (!h) {
        g_api.why = std::string("librccl not loadable (tried ") + tried + "): " + (dlerror() ? dlerror() : "?");
        return &g_api;
    }
    RcclApi a;
    const bool ok = bind(h, "ncclGetUniqueId", a.GetUniqueId) && bind(h, "ncclCommInitRank", a.CommInitRank) &&
                    bind(h, "ncclCommDestroy", a.CommDestroy) && bind(h, "ncclGetErrorString", a.GetErrorString) &&
                    bind(h, "ncclGroupStart", a.GroupStart) && bind(h, "ncclGroupEnd", a.GroupEnd) && bind(h, "ncclSend", a.Send) &&
                    bind(h, "ncclRecv", a.Recv) && bind(h, "ncclAllGather", a.AllGather) && bind(h, "ncclAllReduce", a.AllReduce);
    if (!ok) {
        g_api.why = "librccl lacks a required symbol";
        dlclose(h);
        return &g_api;
    }
    a.handle = h;
    g_api = a;
    return &g_api;
}

}  // namespace

struct fb_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

#define FB_NCCL(ctx, api, call)                                                                                     \
    do {                                                                                                            \
        ncclResult_t r__ = (call);                                                                                  \
        if (r__ != ncclSuccess)                                                                                     \
            return fb_fail((ctx), FB_ERR_COMM, "%s:%d %s -> %s", __FILE__, __LINE__, #call, (api)->GetErrorString(r__)); \
    } while (0)

extern "C" {

int fb_comm_unique_id(fb_ctx* ctx, void* id128) {
    FB_CHECK_ARG(ctx, id128 != nullptr);
    const RcclApi* api = rccl_api();
    if (!api->handle) return fb_fail(ctx, FB_ERR_COMM, "%s", api->why.c_str());
    static_assert(sizeof(ncclUniqueId) == FB_COMM_ID_BYTES, "id size");
    FB_NCCL(ctx, api, api->GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128)));
    return FB_OK;
}

int fb_comm_create(fb_ctx* ctx, const void* id128, int rank, int world, fb_comm** out) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, id128 && out && world >= 1 && rank >= 0 && rank < world);
    const RcclApi* api = rccl_api();
    if (!api->handle) return fb_fail(ctx, FB_ERR_COMM, "%s", api->why.c_str());
    FB_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    fb_comm* c = new fb_comm();
    c->rank = rank; c->world = world;
    ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fb_fail(ctx, FB_ERR_COMM, "ncclCommInitRank(rank %d of %d): %s", rank, world, api->GetErrorString(r));
    }
    *out = c;
    return FB_OK;
}

void fb_comm_destroy(fb_ctx* ctx, fb_comm* comm) {
    if (!comm) return;
    const RcclApi* api = rccl_api();
    if (ctx) hipStreamSynchronize(ctx->stream);
    if (api->handle && comm->comm) api->CommDestroy(comm->comm);
    delete comm;
}

int fb_comm_info(fb_ctx* ctx, fb_comm* comm, int* rank, int* world) {
    FB_CHECK_ARG(ctx, comm != nullptr);
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return FB_OK;
}

int fb_allgather_dev(fb_ctx* ctx, fb_comm* comm, const void* send, void* recv, size_t bytes_per_rank) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, comm && (bytes_per_rank == 0 || (send && recv)));
    if (bytes_per_rank == 0) return FB_OK;
    const RcclApi* api = rccl_api();
    FB_PROF_B(ctx, "rccl_allgather", (double)bytes_per_rank * comm->world);
    FB_NCCL(ctx, api, api->AllGather(send, recv, bytes_per_rank, ncclInt8, comm->comm, ctx->stream));
    return FB_OK;
}

// counts [world] (host, bytes of every rank's contribution, identical on all ranks); recv (root only) receives the
// contributions back to back in rank order.  One grouped set of point-to-point transfers: nothing is padded and only
// the root receives.
int fb_gatherv_dev(fb_ctx* ctx, fb_comm* comm, const void* send, const int64_t* counts, void* recv, int root) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, comm && counts && root >= 0 && root < comm->world);
    for (int r = 0; r < comm->world; ++r) FB_CHECK_ARG(ctx, counts[r] >= 0);
    FB_CHECK_ARG(ctx, counts[comm->rank] == 0 || send != nullptr);
    const RcclApi* api = rccl_api();
    int64_t total = 0;
    for (int r = 0; r < comm->world; ++r) total += counts[r];
    if (comm->rank == root) FB_CHECK_ARG(ctx, total == 0 || recv != nullptr);
    FB_PROF_B(ctx, "rccl_gatherv", (double)(comm->rank == root ? total : counts[comm->rank]));
    if (comm->rank == root) {
        // own part: a device copy on the same stream
        int64_t off = 0;
        for (int r = 0; r < root; ++r) off += counts[r];
        if (counts[root] > 0)
            FB_HIP(ctx, hipMemcpyAsync((char*)recv + off, send, (size_t)counts[root], hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (comm->world == 1) return FB_OK;
    FB_NCCL(ctx, api, api->GroupStart());
    ncclResult_t bad = ncclSuccess;
    if (comm->rank == root) {
        int64_t off = 0;
        for (int r = 0; r < comm->world; ++r) {
            if (r != root && counts[r] > 0) {
                ncclResult_t e = api->Recv((char*)recv + off, (size_t)counts[r], ncclInt8, r, comm->comm, ctx->stream);
                if (e != ncclSuccess) bad = e;
            }
            off += counts[r];
        }
    } else if (counts[comm->rank] > 0) {
        bad = api->Send(send, (size_t)counts[comm->rank], ncclInt8, root, comm->comm, ctx->stream);
    }
    ncclResult_t e = api->GroupEnd();
    if (bad != ncclSuccess) e = bad;
    if (e != ncclSuccess) return fb_fail(ctx, FB_ERR_COMM, "fb_gatherv_dev: %s", api->GetErrorString(e));
    return FB_OK;
}

int fb_allreduce_f64_dev(fb_ctx* ctx, fb_comm* comm, const double* send, double* recv, size_t n, int op) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, comm && (n == 0 || (send && recv)) && (op == FB_REDUCE_SUM || op == FB_REDUCE_MAX));
    if (n == 0) return FB_OK;
    const RcclApi* api = rccl_api();
    FB_PROF(ctx, "rccl_allreduce");
    FB_NCCL(ctx, api, api->AllReduce(send, recv, n, ncclFloat64, op == FB_REDUCE_SUM ? ncclSum : ncclMax, comm->comm, ctx->stream));
    return FB_OK;
}

// point-to-point halo exchange of the coupled-window solver: nsend / nrecv transfers in one group (direct xGMI links)
int fb_sendrecv_dev(fb_ctx* ctx, fb_comm* comm, int nsend, const int* send_peer, const void* const* send_ptr, const int64_t* send_bytes,
                    int nrecv, const int* recv_peer, void* const* recv_ptr, const int64_t* recv_bytes) {
    FB_LOCK(ctx);
    FB_CHECK_ARG(ctx, comm && nsend >= 0 && nrecv >= 0);
    if (nsend + nrecv == 0) return FB_OK;
    FB_CHECK_ARG(ctx, (nsend == 0 || (send_peer && send_ptr && send_bytes)) && (nrecv == 0 || (recv_peer && recv_ptr && recv_bytes)));
    const RcclApi* api = rccl_api();
    FB_PROF(ctx, "rccl_sendrecv");
    FB_NCCL(ctx, api, api->GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int k = 0; k < nsend; ++k) {
        if (send_bytes[k] <= 0) continue;
        ncclResult_t e = api->Send(send_ptr[k], (size_t)send_bytes[k], ncclInt8, send_peer[k], comm->comm, ctx->stream);
        if (e != ncclSuccess) bad = e;
    }
    for (int k = 0; k < nrecv; ++k) {
        if (recv_bytes[k] <= 0) continue;
        ncclResult_t e = api->Recv(recv_ptr[k], (size_t)recv_bytes[k], ncclInt8, recv_peer[k], comm->comm, ctx->stream);
        if (e != ncclSuccess) bad = e;
    }
    ncclResult_t e = api->GroupEnd();
    if (bad != ncclSuccess) e = bad;
    if (e != ncclSuccess) return fb_fail(ctx, FB_ERR_COMM, "fb_sendrecv_dev: %s", api->GetErrorString(e));
    return FB_OK;
}

}  // extern "C"
