// Collective for a sample whose fragments are sharded over the GPUs of a node: every rank
// holds a contiguous block of X's rows and the full O(n) state; per likelihood pass the partial gradients
// (K*n f32) and log-likelihoods are summed over ranks with ONE all-reduce; everything else is replicated.
// RCCL is bound at run time (dlopen) so that single-GPU users need not have it; the prototypes, the id type and the
// enumerators come from its own header, <rccl/rccl.h>.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "comm_internal.hpp"

namespace polee {

namespace {
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
};

Rccl *rccl(std::string &err)
{
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        // RCCL must sit on the SAME HIP runtime as this library (its streams and device pointers are ours).  A process
        // may hold two ROCm stacks -- /opt/rocm and the one bundled with PyTorch -- so first take the librccl that lies
        // next to the libamdhip64 this library is bound to; an RCCL copy picked up by name alone may belong to the other
        // stack (ncclCommInitRank then fails with "unhandled cuda error").
        Dl_info hip_info;
        if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &hip_info) && hip_info.dli_fname) {
            std::string dir(hip_info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                for (const char *name : {"librccl.so.1", "librccl.so"}) {
                    r.lib = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
                    if (r.lib) break;
                }
            }
        }
        if (!r.lib)
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.lib) break;
            }
        if (r.lib) {
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
            r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
            r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(dlsym(r.lib, "ncclReduceScatter"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
            r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
            r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(r.lib, "ncclCommUserRank"));
        }
    }
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
        err = "librccl.so could not be loaded (needed only for row-sharded fits)";
        return nullptr;
    }
    return &r;
}
}  // namespace

polee_status comm_allreduce_device(polee_comm *c, void *buf, size_t count, bool f64)
{
    if (c->host_allreduce) {
        // host-staged communicator (polee_comm_create_host): the buffer goes through host memory and the caller's
        // all-reduce (MPI, gloo, ...) -- for clusters without RCCL between the ranks, and for two ranks sharing one GPU
        const size_t bytes = count * (f64 ? sizeof(double) : sizeof(float));
        c->staging.resize(bytes);
        POLEE_HIP_TRY(c->ctx, hipMemcpyAsync(c->staging.data(), buf, bytes, hipMemcpyDeviceToHost, c->ctx->stream));
        POLEE_HIP_TRY(c->ctx, hipStreamSynchronize(c->ctx->stream));
        const int rc = c->host_allreduce(c->host_user, c->staging.data(), (int64_t)count, f64 ? 1 : 0);
        if (rc != 0) return fail(c->ctx, POLEE_ERR_HIP, "the host all-reduce callback failed (%d)", rc);
        POLEE_HIP_TRY(c->ctx, hipMemcpyAsync(buf, c->staging.data(), bytes, hipMemcpyHostToDevice, c->ctx->stream));
        POLEE_HIP_TRY(c->ctx, hipStreamSynchronize(c->ctx->stream));
        return POLEE_OK;
    }
    std::string err;
    Rccl *r = rccl(err);
    if (!r) return fail(c->ctx, POLEE_ERR_UNSUPPORTED, "%s", err.c_str());
    const ncclDataType_t dt = f64 ? ncclFloat64 : ncclFloat32;
    const ncclComm_t comm = static_cast<ncclComm_t>(c->comm);
    // POLEE_COMM_ALGO=rs_ag: the exchange as an explicit reduce-scatter + all-gather, in place
    // (every rank reduces its own 1 / nranks of the buffer, then the parts are gathered) -- what a ring all-reduce does inside,
    // as two calls, so that the first multi-GPU runs can compare both; a count the ranks do not divide takes the all-reduce.
    static const bool rs_ag = getenv("POLEE_COMM_ALGO") && strcmp(getenv("POLEE_COMM_ALGO"), "rs_ag") == 0;
    if (rs_ag && r->ReduceScatter && r->AllGather && count % (size_t)c->nranks == 0) {
        const size_t chunk = count / (size_t)c->nranks;
        char *mine = static_cast<char *>(buf) + (size_t)c->rank * chunk * (f64 ? sizeof(double) : sizeof(float));
        ncclResult_t rc = r->ReduceScatter(buf, mine, chunk, dt, ncclSum, comm, c->ctx->stream);
        if (rc == ncclSuccess) rc = r->AllGather(mine, buf, chunk, dt, comm, c->ctx->stream);
        if (rc != ncclSuccess)
            return fail(c->ctx, POLEE_ERR_COMM, "ncclReduceScatter / ncclAllGather failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?");
        return POLEE_OK;
    }
    const ncclResult_t rc = r->AllReduce(buf, buf, count, dt, ncclSum, comm, c->ctx->stream);
    if (rc != ncclSuccess)
        return fail(c->ctx, POLEE_ERR_HIP, "ncclAllReduce failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?");
    return POLEE_OK;
}

}  // namespace polee

using namespace polee;

extern "C" {

polee_status polee_comm_unique_id(uint8_t id[POLEE_COMM_ID_BYTES])
{
    if (!id) return fail(nullptr, POLEE_ERR_BAD_ARG, "null argument");
    std::string err;
    Rccl *r = rccl(err);
    if (!r) return fail(nullptr, POLEE_ERR_UNSUPPORTED, "%s", err.c_str());
    static_assert(sizeof(ncclUniqueId) == POLEE_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId uid;
    const ncclResult_t rc = r->GetUniqueId(&uid);
    if (rc != ncclSuccess) return fail(nullptr, POLEE_ERR_HIP, "ncclGetUniqueId failed (%d)", (int)rc);
    memcpy(id, &uid, POLEE_COMM_ID_BYTES);
    return POLEE_OK;
}

polee_status polee_comm_create(polee_ctx *ctx, int32_t nranks, int32_t rank, const uint8_t id[POLEE_COMM_ID_BYTES],
                               polee_comm **out)
{
    POLEE_TRY(use_device(ctx));
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(ctx, POLEE_ERR_BAD_ARG, "polee_comm_create: bad argument (rank %d of %d)", rank, nranks);
    std::string err;
    Rccl *r = rccl(err);
    if (!r) return fail(ctx, POLEE_ERR_UNSUPPORTED, "%s", err.c_str());
    ncclUniqueId uid;
    memcpy(&uid, id, POLEE_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = r->CommInitRank(&comm, nranks, uid, rank);
    if (rc != ncclSuccess || !comm)
        return fail(ctx, POLEE_ERR_HIP, "ncclCommInitRank failed: %s", r->GetErrorString ? r->GetErrorString(rc) : "?");
    polee_comm *c = new (std::nothrow) polee_comm();
    if (!c) {
        r->CommDestroy(comm);
        return fail(ctx, POLEE_ERR_OOM, "out of host memory");
    }
    c->ctx = ctx;
    c->comm = comm;
    c->nranks = nranks;
    c->rank = rank;
    ctx_retain(ctx);
    *out = c;
    return POLEE_OK;
}

polee_status polee_comm_create_host(polee_ctx *ctx, int32_t nranks, int32_t rank, polee_host_allreduce_fn allreduce, void *user,
                                    polee_comm **out)
{
    POLEE_TRY(use_device(ctx));
    if (!out || !allreduce || nranks < 1 || rank < 0 || rank >= nranks)
        return fail(ctx, POLEE_ERR_BAD_ARG, "polee_comm_create_host: bad argument (rank %d of %d)", rank, nranks);
    polee_comm *c = new (std::nothrow) polee_comm();
    if (!c) return fail(ctx, POLEE_ERR_OOM, "out of host memory");
    c->ctx = ctx;
    c->host_allreduce = allreduce;
    c->host_user = user;
    c->nranks = nranks;
    c->rank = rank;
    ctx_retain(ctx);
    *out = c;
    return POLEE_OK;
}

void polee_comm_destroy(polee_comm *c)
{
    if (!c) return;
    if (--c->refs > 0) return;
    std::string err;
    if (c->comm) {
        if (Rccl *r = rccl(err)) {
            (void)hipStreamSynchronize(c->ctx->stream);
            r->CommDestroy(static_cast<ncclComm_t>(c->comm));
        }
    }
    polee_ctx *ctx = c->ctx;
    delete c;
    ctx_release(ctx);
}

// bench hook (include/polee_hip_debug.h): `reps` all-reduces of `count` f32 of a zeroed device buffer on the library's stream,
// bracketed by HIP events: *ms_avg = the exchange's own time per call (what a row-sharded pass or a regression step adds)
polee_status polee_debug_comm_allreduce_ms(polee_comm *c, int64_t count, int32_t reps, double *ms_avg)
{
    if (!c) return fail(nullptr, POLEE_ERR_BAD_ARG, "null communicator");
    polee_ctx *ctx = c->ctx;
    POLEE_TRY(use_device(ctx));
    if (count < 1 || reps < 1 || !ms_avg) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    DevBuf<float> d;
    POLEE_TRY(d.alloc(ctx, (size_t)count));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(d.p, 0, sizeof(float) * (size_t)count, ctx->stream));
    POLEE_TRY(comm_allreduce_device(c, d.p, (size_t)count, false));  // (first call: connection set-up, not timed)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    POLEE_HIP_TRY(ctx, hipEventCreate(&e0));
    hipError_t he = hipEventCreate(&e1);
    polee_status st = POLEE_OK;
    if (he == hipSuccess) he = hipEventRecord(e0, ctx->stream);
    for (int32_t i = 0; i < reps && he == hipSuccess && st == POLEE_OK; ++i) st = comm_allreduce_device(c, d.p, (size_t)count, false);
    if (he == hipSuccess) he = hipEventRecord(e1, ctx->stream);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st != POLEE_OK) return st;
    if (he != hipSuccess) return fail(ctx, POLEE_ERR_HIP, "all-reduce timing failed: %s", hipGetErrorString(he));
    *ms_avg = (double)ms / reps;
    return POLEE_OK;
}

polee_status polee_allreduce_sum_f32(polee_comm *c, float *buf, int64_t count)
{
    if (!c) return fail(nullptr, POLEE_ERR_BAD_ARG, "null communicator");
    polee_ctx *ctx = c->ctx;
    POLEE_TRY(use_device(ctx));
    if (!buf || count < 0) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    if (count == 0) return POLEE_OK;
    DevBuf<float> d;
    POLEE_TRY(d.upload(ctx, buf, (size_t)count));
    POLEE_TRY(comm_allreduce_device(c, d.p, (size_t)count, false));
    return d.download(ctx, buf, (size_t)count);
}

// What the TRANSPORT says (first-contact evidence for the multi-GPU runs nobody could make yet):
// *transport = 1 RCCL, 2 host-staged callback; *count / *user_rank = ncclCommCount / ncclCommUserRank of the RCCL
// communicator (the creation arguments for a host communicator).
polee_status polee_comm_info(const polee_comm *c, int32_t *transport, int32_t *count, int32_t *user_rank)
{
    if (!c) return fail(nullptr, POLEE_ERR_BAD_ARG, "null communicator");
    int t = c->host_allreduce ? 2 : 1, n = c->nranks, rk = c->rank;
    if (!c->host_allreduce && c->comm) {
        std::string err;
        Rccl *r = rccl(err);
        if (r && r->CommCount && r->CommUserRank) {
            if (r->CommCount(static_cast<ncclComm_t>(c->comm), &n) != ncclSuccess ||
                r->CommUserRank(static_cast<ncclComm_t>(c->comm), &rk) != ncclSuccess)
                return fail(c->ctx, POLEE_ERR_COMM, "ncclCommCount / ncclCommUserRank failed");
        }
    }
    if (transport) *transport = t;
    if (count) *count = n;
    if (user_rank) *user_rank = rk;
    return POLEE_OK;
}

int32_t polee_comm_rank(const polee_comm *c) { return c ? c->rank : -1; }
int32_t polee_comm_size(const polee_comm *c) { return c ? c->nranks : 0; }

}  // extern "C"
