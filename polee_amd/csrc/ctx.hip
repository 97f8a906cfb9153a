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
// ---- debug mode of the kept device buffers (common.hpp, DevBlockCache; POLEE_DEVICE_CACHE_POISON=1) -------------------------
constexpr uint32_t POISON_WORD = 0xA5C3F00Du;
__global__ void poison_fill_kernel(uint32_t *p, size_t words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = POISON_WORD;
}
__global__ void poison_check_kernel(const uint32_t *p, size_t words, unsigned long long *bad)
{
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) b += p[i] != POISON_WORD;
    if (b) atomicAdd(bad, b);
}
static std::atomic<long long> g_poison_checked{0}, g_poison_bad_blocks{0}, g_poison_bad_words{0};
void dev_poison_fill(void *p, size_t bytes, hipStream_t stream)
{
    const size_t words = bytes / 4;
    if (!words) return;
    hipLaunchKernelGGL(poison_fill_kernel, dim3((unsigned)std::min<size_t>((words + 255) / 256, 4096)), dim3(256), 0, stream, (uint32_t *)p, words);
    (void)hipGetLastError();
}
size_t dev_poison_bad_words(void *p, size_t bytes)
{
    const size_t words = bytes / 4;
    if (!words) return 0;
    unsigned long long *d_bad = nullptr, bad = 0;
    if (hipMalloc((void **)&d_bad, sizeof bad) != hipSuccess) return 0;
    (void)hipMemset(d_bad, 0, sizeof bad);
    hipLaunchKernelGGL(poison_check_kernel, dim3((unsigned)std::min<size_t>((words + 255) / 256, 4096)), dim3(256), 0, nullptr, (const uint32_t *)p, words, d_bad);
    (void)hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost);  // (the block's own event has been waited for by the caller)
    (void)hipFree(d_bad);
    (void)hipGetLastError();
    return (size_t)bad;
}
void dev_poison_report(size_t granted, size_t bad_words, const char *where)
{
    ++g_poison_checked;
    if (!bad_words) return;
    ++g_poison_bad_blocks;
    g_poison_bad_words += (long long)bad_words;
    fprintf(stderr, "*** polee_hip: a kept device block of %zu bytes was WRITTEN after its owner released it (%zu words differ; %s)\n",
            granted, bad_words, where);
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

// debug mode POLEE_DEVICE_CACHE_POISON=1: blocks verified so far, blocks found overwritten, words overwritten (0 when off)
void polee_debug_device_cache_poison(int64_t *checked_blocks, int64_t *bad_blocks, int64_t *bad_words)
{
    if (checked_blocks) *checked_blocks = (int64_t)polee::g_poison_checked.load();
    if (bad_blocks) *bad_blocks = (int64_t)polee::g_poison_bad_blocks.load();
    if (bad_words) *bad_words = (int64_t)polee::g_poison_bad_words.load();
}

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
