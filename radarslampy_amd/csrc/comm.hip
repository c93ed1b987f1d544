// Multi-GPU exchange of the path: one process per GPU, sequences sharded by rank, NO data-path
// collective.  The only exchange the reference's design has is handing a keyframe to a global map
// (Mapping.Map.addKeyframe, reference Mapping.py:118-147): roam_bcast_keyframe broadcasts the DEVICE-RESIDENT keyframe of
// one lane of the owning rank to every rank with ncclBroadcast (RCCL over xGMI) straight from HBM - no host bounce.
// RCCL is bound at run time (dlopen of librccl.so) so that the library loads on hosts without it; every entry point
// fails with ROAM_E_STATE and a message when it is missing.
#include "roam_internal.h"
#include <dlfcn.h>
#include <cstdlib>
#include <rccl/rccl.h>

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = {0};
};

static RcclApi *rccl_api()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    if (getenv("ROAM_DISABLE_RCCL")) { snprintf(api.why, sizeof(api.why), "disabled by ROAM_DISABLE_RCCL"); return &api; }   // (exercises callers' fallbacks)
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
    }
    if (!api.handle) { snprintf(api.why, sizeof(api.why), "librccl.so not found: %s", dlerror()); return &api; }
#define BIND(field, sym)                                                                         \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym));                   \
    if (!api.field) { snprintf(api.why, sizeof(api.why), "librccl.so lacks %s", sym); dlclose(api.handle); api.handle = nullptr; return &api; }
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(CommCount, "ncclCommCount")
    BIND(CommUserRank, "ncclCommUserRank")
    BIND(Broadcast, "ncclBroadcast")
    BIND(AllReduce, "ncclAllReduce")
    BIND(AllGather, "ncclAllGather")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    return &api;
}

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double *red = nullptr;            // 8-double device buffer for the small reductions
};

#define NCCL_TRY(ctx, call)                                                                                   \
    do {                                                                                                      \
        ncclResult_t r_ = (call);                                                                             \
        if (r_ != ncclSuccess) {                                                                              \
            ROAM_SET_ERR(ctx, "%s failed: %s (%s:%d)", #call, rccl_api()->GetErrorString(r_), __FILE__, __LINE__); \
            return ROAM_E_HIP;                                                                                \
        }                                                                                                     \
    } while (0)

static_assert(sizeof(ncclUniqueId) == ROAM_COMM_ID_BYTES, "ncclUniqueId size");

extern "C" {

// 1 when librccl.so loads and exports what roam_comm_* needs (no communicator, no GPU touched): ranks vote on this BEFORE anyone
// enters the collective ncclCommInitRank, which has no timeout
int32_t roam_comm_available(void) { return rccl_api()->handle ? 1 : 0; }

int32_t roam_comm_unique_id(uint8_t *id_out)
{
    if (!id_out) return ROAM_E_ARG;
    RcclApi *a = rccl_api();
    if (!a->handle) return ROAM_E_STATE;
    ncclUniqueId id;
    if (a->GetUniqueId(&id) != ncclSuccess) return ROAM_E_HIP;
    memcpy(id_out, &id, sizeof(id));
    return ROAM_OK;
}

int32_t roam_comm_init(roam_ctx *ctx, const uint8_t *id_bytes, int32_t rank, int32_t world)
{
    if (!ctx) return ROAM_E_ARG;
    ARG_CHECK(ctx, id_bytes && world >= 1 && rank >= 0 && rank < world);
    RcclApi *a = rccl_api();
    if (!a->handle) { ROAM_SET_ERR(ctx, "RCCL unavailable: %s", a->why); return ROAM_E_STATE; }
    if (ctx->comm) { ROAM_SET_ERR(ctx, "communicator already initialised"); return ROAM_E_STATE; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Comm *c = new Comm();
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclResult_t r = a->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { ROAM_SET_ERR(ctx, "ncclCommInitRank failed: %s", a->GetErrorString(r)); delete c; return ROAM_E_HIP; }
    c->rank = rank; c->world = world;
    if (hipMalloc(reinterpret_cast<void **>(&c->red), 8 * sizeof(double)) != hipSuccess) {
        a->CommDestroy(c->comm); delete c; ROAM_SET_ERR(ctx, "hipMalloc failed"); return ROAM_E_HIP;
    }
    ctx->comm = c;
    return ROAM_OK;
}

int32_t roam_comm_info(roam_ctx *ctx, int32_t *rank, int32_t *world)
{
    if (!ctx) return ROAM_E_ARG;
    if (!ctx->comm) { ROAM_SET_ERR(ctx, "communicator not initialised"); return ROAM_E_STATE; }
    int r = -1, n = -1;                                   // what RCCL itself reports
    NCCL_TRY(ctx, rccl_api()->CommUserRank(ctx->comm->comm, &r));
    NCCL_TRY(ctx, rccl_api()->CommCount(ctx->comm->comm, &n));
    if (rank) *rank = r;
    if (world) *world = n;
    return ROAM_OK;
}

// in-place all-reduce of n <= 8 doubles (op 0 = max, 1 = sum); blocking.  Used for the max-over-ranks wall time and as
// the barrier of bench.py (no torch.distributed in the measured program).
int32_t roam_comm_allreduce_f64(roam_ctx *ctx, double *inout, int32_t n, int32_t op)
{
    if (!ctx) return ROAM_E_ARG;
    ARG_CHECK(ctx, inout && n >= 1 && n <= 8 && (op == 0 || op == 1));
    if (!ctx->comm) { ROAM_SET_ERR(ctx, "communicator not initialised"); return ROAM_E_STATE; }
    Comm *c = ctx->comm;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(c->red, inout, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    NCCL_TRY(ctx, rccl_api()->AllReduce(c->red, c->red, (size_t)n, ncclDouble, op == 0 ? ncclMax : ncclSum, c->comm, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(inout, c->red, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return ROAM_OK;
}

int32_t roam_comm_barrier(roam_ctx *ctx)
{
    double one = 1.0;
    return roam_comm_allreduce_f64(ctx, &one, 1, 1);
}

int32_t roam_comm_destroy(roam_ctx *ctx)
{
    if (!ctx) return ROAM_E_ARG;
    if (!ctx->comm) return ROAM_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    rccl_api()->CommDestroy(ctx->comm->comm);
    hipFree(ctx->comm->red);
    delete ctx->comm;
    ctx->comm = nullptr;
    return ROAM_OK;
}

}  // extern "C"

// used by engine.hip (the keyframe payload lives in the engine's buffers)
int32_t roam_comm_bcast_bytes(roam_ctx *ctx, void *dev_buf, size_t bytes, int root)
{
    if (!ctx->comm) { ROAM_SET_ERR(ctx, "communicator not initialised"); return ROAM_E_STATE; }
    ARG_CHECK(ctx, root >= 0 && root < ctx->comm->world);
    NCCL_TRY(ctx, rccl_api()->Broadcast(dev_buf, dev_buf, bytes, ncclChar, root, ctx->comm->comm, ctx->stream));
    return ROAM_OK;
}

int roam_comm_rank(const roam_ctx *ctx) { return ctx->comm ? ctx->comm->rank : 0; }
int roam_comm_world(const roam_ctx *ctx) { return ctx->comm ? ctx->comm->world : 1; }

// every rank's `bytes` at send -> recv[rank * bytes ..] on every rank, enqueued on `st` (nothing waits on the host)
int32_t roam_comm_allgather_bytes(roam_ctx *ctx, const void *send, void *recv, size_t bytes, hipStream_t st)
{
    if (!ctx->comm) { ROAM_SET_ERR(ctx, "communicator not initialised"); return ROAM_E_STATE; }
    NCCL_TRY(ctx, rccl_api()->AllGather(send, recv, bytes, ncclChar, ctx->comm->comm, st));
    return ROAM_OK;
}
