// Context: device, stream, error state, HIP-event stopwatch.
#include "common.hpp"

namespace polee {
std::string &global_error()
{
    static thread_local std::string e;
    return e;
}
void ctx_retain(polee_ctx *ctx)
{
    if (ctx) ++ctx->refs;
}
void ctx_release(polee_ctx *ctx)
{
    if (!ctx || --ctx->refs > 0) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}
}  // namespace polee

using namespace polee;

extern "C" {

#ifndef POLEE_BUILD_INFO
#define POLEE_BUILD_INFO "unknown compiler"
#endif
// (the build records the hipcc version and the backend tuning flags loglik.hip was compiled with: csrc/Makefile)
const char *polee_version(void) { return "polee_hip 0.3 (gfx950); " POLEE_BUILD_INFO; }

// the host builders keep their large scratch blocks for the next sample (common.hpp HugeBlockCache)
void polee_host_cache_trim(void)
{
    polee::HugeBlockCache::get().trim();
    polee::DevBlockCache::get().trim();  // (the kept device buffers too: common.hpp)
}
int64_t polee_host_cache_configure(int64_t cap_mb)
{
    polee::HugeBlockCache &c = polee::HugeBlockCache::get();
    if (cap_mb >= 0) c.set_cap((size_t)cap_mb << 20);
    return (int64_t)(c.cap >> 20);
}
int64_t polee_host_cache_bytes(void) { return (int64_t)polee::HugeBlockCache::get().cached_bytes(); }
int64_t polee_device_cache_bytes(void) { return (int64_t)polee::DevBlockCache::get().kept_bytes(); }

polee_status polee_ctx_create(int device, polee_ctx **out)
{
    if (!out) return fail(nullptr, POLEE_ERR_BAD_ARG, "polee_ctx_create: null out pointer");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, POLEE_ERR_HIP, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= count)
        return fail(nullptr, POLEE_ERR_BAD_ARG, "device %d out of range (%d devices)", device, count);
    polee_ctx *ctx = new (std::nothrow) polee_ctx();
    if (!ctx) return fail(nullptr, POLEE_ERR_OOM, "out of host memory");
    ctx->device = device;
    auto bail = [&](hipError_t err, const char *what) {
        polee_status s = fail(nullptr, POLEE_ERR_HIP, "%s failed: %s", what, hipGetErrorString(err));
        if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
        if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return s;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
    ctx->num_cus = prop.multiProcessorCount;
    ctx->lds_per_block = prop.sharedMemPerBlock;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail(e, "hipStreamCreate");
    if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
    *out = ctx;
    return POLEE_OK;
}

void polee_ctx_destroy(polee_ctx *ctx) { ctx_release(ctx); }

const char *polee_last_error(const polee_ctx *ctx)
{
    if (ctx && !ctx->err.empty()) return ctx->err.c_str();
    return global_error().c_str();
}

polee_status polee_ctx_synchronize(polee_ctx *ctx)
{
    POLEE_TRY(use_device(ctx));
    POLEE_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return POLEE_OK;
}

polee_status polee_ctx_mem_info(polee_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes)
{
    POLEE_TRY(use_device(ctx));
    size_t f = 0, t = 0;
    POLEE_HIP_TRY(ctx, hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return POLEE_OK;
}

void *polee_ctx_stream(polee_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

polee_status polee_ctx_timer_start(polee_ctx *ctx)
{
    POLEE_TRY(use_device(ctx));
    POLEE_HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return POLEE_OK;
}

polee_status polee_ctx_timer_stop(polee_ctx *ctx, double *elapsed_ms)
{
    POLEE_TRY(use_device(ctx));
    POLEE_HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    POLEE_HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    POLEE_HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (elapsed_ms) *elapsed_ms = (double)ms;
    return POLEE_OK;
}

}  // extern "C"
