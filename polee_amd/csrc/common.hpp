// Shared host-side plumbing of libpolee_hip: context, error state, device buffers.
#pragma once
#include <mutex>
#include <sys/mman.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <vector>

#include "../../include/polee_hip.h"

namespace polee {

std::string &global_error();  // last error raised before / without a context

}  // namespace polee

struct polee_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    int num_cus = 256;
    size_t lds_per_block = 65536;
    int refs = 1;  // the creator's reference + one per live child handle
};

namespace polee {

inline polee_status fail(polee_ctx *ctx, polee_status code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    global_error() = buf;
    return code;
}

#define POLEE_HIP_TRY(ctx, expr)                                                          \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess)                                                            \
            return ::polee::fail((ctx), e__ == hipErrorOutOfMemory ? POLEE_ERR_OOM        \
                                                                  : POLEE_ERR_HIP,       \
                                 "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),  \
                                 __FILE__, __LINE__);                                     \
    } while (0)

#define POLEE_TRY(expr)                     \
    do {                                    \
        polee_status s__ = (expr);          \
        if (s__ != POLEE_OK) return s__;    \
    } while (0)

#define POLEE_KERNEL_CHECK(ctx)                                                            \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess)                                                             \
            return ::polee::fail((ctx), POLEE_ERR_HIP, "kernel launch failed: %s (%s:%d)", \
                                 hipGetErrorString(e__), __FILE__, __LINE__);              \
    } while (0)

// Debug mode of the kept device buffers (POLEE_DEVICE_CACHE_POISON=1): ctx.hip
void dev_poison_fill(void *p, size_t bytes, hipStream_t stream);  // fills the block with a pattern, on the stream
size_t dev_poison_bad_words(void *p, size_t bytes);               // 32-bit words that no longer hold it (synchronous)
void dev_poison_report(size_t granted, size_t bad_words, const char *where);

// Freed device buffers are KEPT (process-wide, per device, by size class) instead of going back through hipFree: the builders
// allocate and free hundreds of buffers per sample, hipFree waits for the whole device, and with plain hipMalloc / hipFree every
// few samples one of the first large allocations of a sample stalled for 2 - 3 s (tools/probe/prep_stages_one.py, 20 samples).
// A block is given back together with an event recorded on the stream its owner worked on; whoever takes it next waits for
// that event first (normally long complete) -- what hipFree guaranteed, for that one block only.  (The runtime's own
// stream-ordered pool, hipMallocAsync with an infinite release threshold, was tried first: under it freshly uploaded arrays read
// back partly zeroed -- as if a block were handed out while a neighbour's memset still covered it -- in every run; not used.)
// POLEE_DEVICE_CACHE_MB: cap of the kept bytes (default: a quarter of the device's memory, at most 64 GiB; 0 = plain hipMalloc /
// hipFree); polee_host_cache_trim() frees them, and so does every allocation of this library that fails.
class DevBlockCache {
public:
    static DevBlockCache &get()
    {
        static DevBlockCache *c = new DevBlockCache();  // (never destroyed: buffers may be released during process exit)
        return *c;
    }
    static size_t size_class(size_t bytes)
    {
        if (bytes <= 4096) return 4096;
        int p = 63 - __builtin_clzll((unsigned long long)bytes);
        const size_t step = (size_t)1 << (p - 3);
        return (bytes + step - 1) / step * step;
    }
    // a kept block of exactly this class on this device, or null
    void *take(int device, size_t granted)
    {
        Block blk;
        {
            std::lock_guard<std::mutex> g(mu_);
            auto range = free_.equal_range(granted);
            auto it = range.first;
            for (; it != range.second; ++it)
                if (it->second.device == device) break;
            if (it == range.second) return nullptr;
            blk = it->second;
            free_.erase(it);
            kept_ -= granted;
            kept_dev_[device] -= granted;
        }
        if (blk.ev) {
            if (hipEventSynchronize(blk.ev) != hipSuccess) {  // (should not happen: then the whole device, as hipFree would have waited)
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
                (void)hipGetLastError();
            }
            (void)hipEventDestroy(blk.ev);
        }
        if (poison_) dev_poison_report(granted, dev_poison_bad_words(blk.p, granted), "taken from the cache");
        return blk.p;
    }
    // false: not kept (cache off or full, or the stream is gone) -- the caller frees it
    bool give(int device, void *p, size_t granted, hipStream_t stream)
    {
        if (cap_ == 0 || granted > cap_) return false;
        // a stream that is being captured into a graph: an event recorded there is a graph node, not a point in time -- the plain
        // free then, as before (the regression model's captured step releases a temporary)
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return false;
        }
        // debug mode: the block is filled with a pattern ON THE OWNER'S STREAM -- behind everything the owner queued there --
        // and the pattern is verified when the block is taken again or freed: a kernel of ANOTHER stream (or a late one of
        // this stream) that still writes to the block after its owner released it shows up as overwritten words
        if (poison_) dev_poison_fill(p, granted, stream);
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, stream) != hipSuccess) {
            (void)hipGetLastError();
            if (ev) (void)hipEventDestroy(ev);
            return false;
        }
        std::lock_guard<std::mutex> g(mu_);
        learn_cap();
        if (kept_dev_[device] + granted > cap_) {  // (the cap is per device: a quarter of ITS memory, whatever the other devices keep)
            (void)hipEventDestroy(ev);
            return false;
        }
        free_.emplace(granted, Block{p, ev, device});
        kept_ += granted;
        kept_dev_[device] += granted;
        return true;
    }
    void trim()
    {
        std::multimap<size_t, Block> all;
        {
            std::lock_guard<std::mutex> g(mu_);
            all.swap(free_);
            kept_ = 0;
            kept_dev_.clear();
        }
        for (auto &kv : all) {
            if (kv.second.ev) {
                (void)hipEventSynchronize(kv.second.ev);
                (void)hipEventDestroy(kv.second.ev);
            }
            if (poison_) dev_poison_report(kv.first, dev_poison_bad_words(kv.second.p, kv.first), "freed by trim");
            (void)hipFree(kv.second.p);
        }
    }
    size_t kept_bytes()
    {
        std::lock_guard<std::mutex> g(mu_);
        return kept_;
    }
    bool enabled() const { return cap_ != 0; }
    bool poisoning() const { return poison_; }

private:
    struct Block {
        void *p;
        hipEvent_t ev;
        int device;
    };
    DevBlockCache()
    {
        const char *pz = getenv("POLEE_DEVICE_CACHE_POISON");
        poison_ = pz && atoi(pz) != 0;
        const char *e = getenv("POLEE_DEVICE_CACHE_MB");
        if (e) {
            cap_ = (size_t)atoll(e) << 20;
            cap_known_ = true;
        } else {
            cap_ = (size_t)64 << 30;  // (until the first give() on a device: then a share of that device's memory, below)
        }
    }
    // Default cap, PER DEVICE: a quarter of a device's TOTAL memory -- 72 GiB on a 288 GB MI355X, never more than 64 GiB;
    // the devices of one node are alike, so the figure is asked once -- asked of the runtime at the first block given back (the constructor may run before any device is selected).  Other
    // users of the device (RCCL, PyTorch, another process) see the kept bytes as used memory: polee_host_cache_trim() returns
    // them, every failed allocation of this library trims and retries, and a cohort's worker processes get their shares
    // from the same query (polee_amd/cohort.py).
    void learn_cap()
    {
        if (cap_known_) return;
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) == hipSuccess && t > 0) cap_ = std::min((size_t)64 << 30, t / 4);
        (void)hipGetLastError();
        cap_known_ = true;
    }
    std::mutex mu_;
    std::multimap<size_t, Block> free_;
    size_t kept_ = 0, cap_ = 0;      // kept_: all devices (kept_bytes()); cap_: per device
    std::map<int, size_t> kept_dev_;  // bytes kept per device (one process may drive several GPUs)
    bool cap_known_ = false;
    bool poison_ = false;
};

// Device buffer owned by a handle.
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    size_t granted = 0;             // bytes of the block behind p (its size class)
    hipStream_t owner_stream = nullptr;  // the stream its owner works on (an event there marks the block free)
    int owner_device = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p && !(granted && DevBlockCache::get().give(owner_device, p, granted, owner_stream))) (void)hipFree(p);
        p = nullptr;
        n = 0;
        granted = 0;
        owner_stream = nullptr;
    }
    polee_status alloc(polee_ctx *ctx, size_t count)
    {
        if (count <= n && p) return POLEE_OK;
        release();
        if (count == 0) return POLEE_OK;
        const size_t bytes = count * sizeof(T);
        if (ctx && DevBlockCache::get().enabled()) {
            granted = DevBlockCache::size_class(bytes);
            owner_stream = ctx->stream;
            owner_device = ctx->device;
            p = static_cast<T *>(DevBlockCache::get().take(ctx->device, granted));
        }
        if (!p) {
            hipError_t e = hipMalloc((void **)&p, granted ? granted : bytes);
            if (e != hipSuccess && DevBlockCache::get().kept_bytes() > 0) {  // (memory is short: what is kept goes back first)
                (void)hipGetLastError();
                DevBlockCache::get().trim();
                e = hipMalloc((void **)&p, granted ? granted : bytes);
            }
            if (e != hipSuccess) {
                p = nullptr;
                granted = 0;
                (void)hipGetLastError();
                return fail(ctx, POLEE_ERR_OOM, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
            }
        }
        n = count;
        return POLEE_OK;
    }
    void take(DevBuf &other)  // (ownership moves here)
    {
        release();
        p = other.p;
        n = other.n;
        granted = other.granted;
        owner_stream = other.owner_stream;
        owner_device = other.owner_device;
        other.p = nullptr;
        other.n = 0;
        other.granted = 0;
        other.owner_stream = nullptr;
    }
    polee_status upload(polee_ctx *ctx, const T *host, size_t count)
    {
        POLEE_TRY(alloc(ctx, count));
        if (count)
            POLEE_HIP_TRY(ctx, hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice,
                                              ctx->stream));
        // host memory is borrowed for the call only: make the copy complete before returning
        POLEE_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return POLEE_OK;
    }
    polee_status upload(polee_ctx *ctx, const std::vector<T> &v) { return upload(ctx, v.data(), v.size()); }
    polee_status download(polee_ctx *ctx, T *host, size_t count) const
    {
        if (count)
            POLEE_HIP_TRY(ctx, hipMemcpyAsync(host, p, count * sizeof(T), hipMemcpyDeviceToHost,
                                              ctx->stream));
        POLEE_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return POLEE_OK;
    }
};

// rocPRIM's temporary storage of the device builders, grown on demand: a DevBuf, so that it comes from (and goes back to) the kept
// blocks instead of hipMalloc / hipFree -- hipFree waits for the whole device -- and an allocation that fails trims the cache and
// retries like every other (the builders' own Scratch used plain hipMalloc and returned the error).
struct DevScratch {
    polee_ctx *ctx;
    DevBuf<uint8_t> buf;
    void *p = nullptr;
    size_t bytes = 0;
    explicit DevScratch(polee_ctx *c) : ctx(c) {}
    hipError_t need(size_t b)
    {
        if (b <= bytes) return hipSuccess;
        p = nullptr;
        bytes = 0;
        if (buf.alloc(ctx, b) != POLEE_OK) return hipErrorOutOfMemory;  // (releases the smaller block first)
        p = buf.p;
        bytes = b;
        return hipSuccess;
    }
};

// host threads for the one-off layout work (bounded: the loops are memory-bound well before 64 threads)
// CPUs this process may actually use: a container often shows every core of the host but runs under a CFS bandwidth
// quota (cgroup v2 cpu.max "quota period", v1 cpu.cfs_quota_us / cpu.cfs_period_us) -- threads beyond it only get
// throttled in bursts.  0 = no quota found.
inline unsigned cgroup_cpu_quota()
{
    long quota = -1, period = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (fscanf(f, "%31s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
        fclose(f);
    } else {
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(g, "%ld", &quota) != 1) quota = -1;
            fclose(g);
        }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(g, "%ld", &period) != 1) period = -1;
            fclose(g);
        }
    }
    if (quota <= 0 || period <= 0) return 0;
    return (unsigned)std::max(1L, (quota + period - 1) / period);
}
inline unsigned host_threads()
{
    static const unsigned n = [] {
        unsigned cap = 48;
        if (const char *e = getenv("POLEE_HOST_THREADS"))
            cap = (unsigned)std::max(1, atoi(e));
        else if (const unsigned q = cgroup_cpu_quota())
            cap = std::min(cap, q);
        return std::max(1u, std::min(cap, std::thread::hardware_concurrency()));
    }();
    return n;
}

// Large blocks (>= 1 MiB) come straight from mmap with MADV_HUGEPAGE: the one-off host builders fill gigabytes of fresh
// memory once, and with 4 KiB pages that is a million page faults in (and a million page frees out) per gigabyte-sized
// buffer -- measured at BASELINE's C2: the two passes of the CSC -> rows transposition 0.36 -> 0.05 s.
// Freed blocks are kept (up to POLEE_HOST_CACHE_MB; default: a quarter of the available memory, at most 8192) and handed out again: unmapping a 3 GB staging buffer
// costs 0.14 s of kernel time, and doing it on a helper thread only moves the cost (the unmap holds the address-space
// lock that every page fault of the next phase needs: measured, the next phase got 0.14 s slower).  A reused block is
// already resident, so the second use does not fault either -- within one build (the sort's buffers reuse the
// transposition's) and from one sample of a cohort to the next.  polee_host_cache_trim() releases everything.
struct HugeBlockCache {
    static constexpr size_t HUGE_PAGE = (size_t)2 << 20;
    struct Blk {
        void *p;
        size_t len;
    };
    std::mutex mu;
    std::vector<Blk> free_;             // oldest first
    std::vector<Blk> live_;             // blocks handed out (a dozen at a time: linear search)
    size_t cached = 0, cap;
    // Default cap: a quarter of the memory this process can still get -- MemAvailable, cut to what the cgroup
    // leaves (memory.max - memory.current) -- and never more than 8 GiB; POLEE_HOST_CACHE_MB or polee_host_cache_configure()
    // override it (a cohort's worker processes each get their share, polee_amd/cohort.py).
    static size_t default_cap()
    {
        long long avail = -1;
        if (FILE *f = fopen("/proc/meminfo", "r")) {
            char line[128];
            while (fgets(line, sizeof line, f))
                if (sscanf(line, "MemAvailable: %lld kB", &avail) == 1) {
                    avail <<= 10;
                    break;
                }
            fclose(f);
        }
        long long mx = -1, cur = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/memory.max", "r")) {
            char q[32] = {0};
            if (fscanf(f, "%31s", q) == 1 && strcmp(q, "max") != 0) mx = atoll(q);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/memory/memory.limit_in_bytes", "r")) {
            if (fscanf(g, "%lld", &mx) != 1 || mx > (1LL << 60)) mx = -1;
            fclose(g);
        }
        if (mx > 0) {
            if (FILE *f = fopen("/sys/fs/cgroup/memory.current", "r")) {
                if (fscanf(f, "%lld", &cur) != 1) cur = 0;
                fclose(f);
            } else if (FILE *g = fopen("/sys/fs/cgroup/memory/memory.usage_in_bytes", "r")) {
                if (fscanf(g, "%lld", &cur) != 1) cur = 0;
                fclose(g);
            }
            const long long left = std::max(0LL, mx - cur);
            avail = avail < 0 ? left : std::min(avail, left);
        }
        const size_t hard = (size_t)8192 << 20;
        return avail < 0 ? hard : std::min(hard, (size_t)(avail / 4));
    }
    HugeBlockCache()
    {
        const char *e = getenv("POLEE_HOST_CACHE_MB");
        cap = e ? (size_t)std::max(0L, atol(e)) << 20 : default_cap();
    }
    // new cap (bytes); frees cached blocks, oldest first, until the cache fits
    void set_cap(size_t bytes)
    {
        std::vector<Blk> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            cap = bytes;
            while (cached > cap && !free_.empty()) {
                drop.push_back(free_.front());
                cached -= free_.front().len;
                free_.erase(free_.begin());
            }
        }
        for (const Blk &d : drop) (void)munmap(d.p, d.len);
    }
    size_t cached_bytes()
    {
        std::lock_guard<std::mutex> g(mu);
        return cached;
    }
    static HugeBlockCache &get()
    {
        static HugeBlockCache *c = new HugeBlockCache();  // (never destroyed: handles may be released after static destructors ran)
        return *c;
    }
    void *take(size_t bytes)
    {
        const size_t len = (bytes + HUGE_PAGE - 1) / HUGE_PAGE * HUGE_PAGE;
        {
            std::lock_guard<std::mutex> g(mu);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); ++i)  // smallest block that fits without wasting more than half of it
                if (free_[i].len >= len && free_[i].len <= 2 * len && (best == free_.size() || free_[i].len < free_[best].len))
                    best = i;
            if (best != free_.size()) {
                const Blk b = free_[best];
                free_.erase(free_.begin() + (ptrdiff_t)best);
                cached -= b.len;
                live_.push_back(b);
                return b.p;
            }
        }
        void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) return nullptr;
        (void)madvise(p, len, MADV_HUGEPAGE);
        std::lock_guard<std::mutex> g(mu);
        live_.push_back(Blk{p, len});
        return p;
    }
    void give(void *p)
    {
        std::vector<Blk> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            Blk b{p, 0};
            for (size_t i = 0; i < live_.size(); ++i)
                if (live_[i].p == p) {
                    b = live_[i];
                    live_[i] = live_.back();
                    live_.pop_back();
                    break;
                }
            if (b.len == 0) return;  // (not ours: cannot happen)
            if (b.len > cap)
                drop.push_back(b);
            else {
                while (cached + b.len > cap) {
                    drop.push_back(free_.front());
                    cached -= free_.front().len;
                    free_.erase(free_.begin());
                }
                free_.push_back(b);
                cached += b.len;
            }
        }
        for (const Blk &d : drop) (void)munmap(d.p, d.len);
    }
    void trim()
    {
        std::vector<Blk> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            drop.swap(free_);
            cached = 0;
        }
        for (const Blk &d : drop) (void)munmap(d.p, d.len);
    }
};

template <class T>
struct huge_allocator : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = huge_allocator<U>;
    };
    using std::allocator<T>::allocator;
    static constexpr size_t HUGE_MIN = (size_t)1 << 20;
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        if (bytes < HUGE_MIN) return std::allocator<T>::allocate(n);
        void *p = HugeBlockCache::get().take(bytes);
        if (!p) throw std::bad_alloc();
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t n)
    {
        if (n * sizeof(T) < HUGE_MIN)
            std::allocator<T>::deallocate(p, n);
        else
            HugeBlockCache::get().give(p);
    }
};
// a std::vector that behaves like one (value-initialises) on that allocator
template <class T>
using BVec = std::vector<T, huge_allocator<T>>;

// std::vector allocator that leaves trivially constructible elements uninitialised on resize (a 1 GB resize otherwise
// spends its time writing zeros that are overwritten at once)
template <class T>
struct default_init_allocator : huge_allocator<T> {
    template <class U>
    struct rebind {
        using other = default_init_allocator<U>;
    };
    using huge_allocator<T>::huge_allocator;
    template <class U>
    void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value)
    {
        ::new (static_cast<void *>(p)) U;
    }
    template <class U, class... Args>
    void construct(U *p, Args &&...args)
    {
        ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...);
    }
};

template <class T>
using RawVec = std::vector<T, default_init_allocator<T>>;

// f(lo, hi, thread) over [0, count) in chunks of `grain` on a few host threads (dynamic assignment)
template <class F>
inline void parallel_chunks(size_t count, size_t grain, F &&f)
{
    const unsigned hw = host_threads();
    const size_t nchunks = (count + grain - 1) / grain;
    if (hw == 1 || nchunks <= 1) {
        for (size_t c = 0; c < nchunks; ++c) f(c * grain, std::min(count, (c + 1) * grain), 0u);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    // an exception in a worker (std::bad_alloc from a builder's vector) must not reach std::terminate: the first one is
    // kept, the other workers stop taking chunks, and it is rethrown on the calling thread after the join
    std::exception_ptr first;
    std::mutex first_mu;
    std::atomic<bool> stop{false};
    for (unsigned th = 0; th < std::min<size_t>(hw, nchunks); ++th)
        pool.emplace_back([&, th]() {
            try {
                for (size_t c = next++; c < nchunks && !stop.load(std::memory_order_relaxed); c = next++)
                    f(c * grain, std::min(count, (c + 1) * grain), th);
            } catch (...) {
                std::lock_guard<std::mutex> g(first_mu);
                if (!first) first = std::current_exception();
                stop = true;
            }
        });
    for (auto &t : pool) t.join();
    if (first) std::rethrow_exception(first);
}


inline polee_status use_device(polee_ctx *ctx)
{
    if (!ctx) return fail(nullptr, POLEE_ERR_BAD_ARG, "null context");
    // (every entry point starts here: an error some earlier call left behind on this thread -- a finaliser's, a call whose status
    // nobody read -- must not be taken for the failure of this call's first kernel launch)
    (void)hipGetLastError();
    POLEE_HIP_TRY(ctx, hipSetDevice(ctx->device));
    return POLEE_OK;
}

// Handles keep their parents alive: destroying a context (tree, matrix) while a child still
// exists only drops the caller's reference (garbage-collected host languages finalise in
// arbitrary order).
void ctx_retain(polee_ctx *ctx);
void ctx_release(polee_ctx *ctx);

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Nothing may unwind through the C ABI: the host-side builders allocate gigabytes through std::vector, on
// the calling thread and in parallel_chunks workers (which hand their first exception back to the caller).
template <class F>
inline polee_status guarded(polee_ctx *ctx, const char *what, F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return fail(ctx, POLEE_ERR_OOM, "%s: out of host memory", what);
    } catch (const std::exception &e) {
        return fail(ctx, POLEE_ERR_UNSUPPORTED, "%s: %s", what, e.what());
    } catch (...) {
        return fail(ctx, POLEE_ERR_UNSUPPORTED, "%s: unknown exception", what);
    }
}

// accessors of the approximation handle (approx.hip) for its consumers
polee_ctx *approx_ctx(const polee_approx *ap);
void approx_dims(const polee_approx *ap, int32_t *S, int32_t *n);
polee_status approx_set_genes(polee_approx *ap, const int32_t *gene_of, int32_t G);
polee_status approx_gene_logprob_device(polee_approx *ap, const float *d_xg, float *d_xi, float *d_lp, float *d_gg);

}  // namespace polee
