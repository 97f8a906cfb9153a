// Batched prefix sums for the tree kernels (gfx950, wave64).
//
// Every Polya-tree operation of the reference is a serial pointer-chasing walk
// (src/ptt.jl:134-155,175-207; hsb_ops.cpp:87-109,212-238,342-391).  Because the nodes
// are in DFS pre-order, those walks are prefix sums over (a) the Euler tour of the tree
// or (b) the leaves in DFS order, so one three-phase scan serves all of them:
//   scan_reduce  : one workgroup per 1024-element chunk -> chunk total
//   scan_spine   : one workgroup per batch row -> exclusive scan of chunk totals
//   scan_apply   : re-loads the chunk, scans it with the chunk offset, hands every
//                  element's exclusive/inclusive prefix to an emit functor
// Element types: double (log-space products) and dd (double-double, ~1e-32 relative)
// for sums whose differences are taken afterwards -- subtree sums come out with full
// f64 relative accuracy, as the reference's exact tree recursion gives.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace polee {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 2;
constexpr int SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;

struct dd {
    double hi, lo;
};

__host__ __device__ inline dd dd_make(double a) { return dd{a, 0.0}; }

// Error-free transformation based double-double addition (Knuth TwoSum + renormalise).
__host__ __device__ inline dd dd_add(dd a, dd b)
{
    double s = a.hi + b.hi;
    double bb = s - a.hi;
    double e = (a.hi - (s - bb)) + (b.hi - bb);
    e += a.lo + b.lo;
    double hi = s + e;
    double lo = e - (hi - s);
    return dd{hi, lo};
}
__host__ __device__ inline dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }
// (a - b) rounded to double
__host__ __device__ inline double dd_diff(dd a, dd b)
{
    dd r = dd_add(a, dd_neg(b));
    return r.hi + r.lo;
}

// A double moved between lanes by DPP (data-parallel primitives: the lane permutation is part of a VALU move, no LDS round trip as
// ds_bpermute makes).  Lanes whose source lies outside the permutation's range, or whose row is masked off, receive 0 -- the
// identity of every scan here.  CTRL: 0x110 + n = row_shr:n (within rows of 16 lanes), 0x142 / 0x143 = row_bcast:15 / :31 (the last
// lane of a row / of the first two rows to the rows behind: gfx9), 0x138 = wave_shr:1.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_f64(double x)
{
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false));
}

template <typename T>
struct ScanOps;
template <>
struct ScanOps<double> {
    __device__ static double zero() { return 0.0; }
    __device__ static double add(double a, double b) { return a + b; }
    __device__ static double shfl_up(double v, int d) { return __shfl_up(v, d, 64); }
    template <int CTRL, int ROW_MASK>
    __device__ static double dpp(double v) { return dpp_f64<CTRL, ROW_MASK>(v); }
};
template <>
struct ScanOps<dd> {
    __device__ static dd zero() { return dd{0.0, 0.0}; }
    __device__ static dd add(dd a, dd b) { return dd_add(a, b); }
    __device__ static dd shfl_up(dd v, int d) { return dd{__shfl_up(v.hi, d, 64), __shfl_up(v.lo, d, 64)}; }
    template <int CTRL, int ROW_MASK>
    __device__ static dd dpp(dd v) { return dd{dpp_f64<CTRL, ROW_MASK>(v.hi), dpp_f64<CTRL, ROW_MASK>(v.lo)}; }
};

// Inclusive scan across the 64 lanes of a wave: four shifts inside the rows of 16 lanes, then the rows' totals handed on
// (row 0 -> 1 and 2 -> 3, then rows 0-1 -> 2-3) -- six adds per lane like the shuffle form it replaces (round 6), whose six
// ds_bpermute round trips and per-step selects it does without.
template <typename T>
__device__ inline T wave_inclusive_scan(T v)
{
    v = ScanOps<T>::add(ScanOps<T>::template dpp<0x111, 0xf>(v), v);
    v = ScanOps<T>::add(ScanOps<T>::template dpp<0x112, 0xf>(v), v);
    v = ScanOps<T>::add(ScanOps<T>::template dpp<0x114, 0xf>(v), v);
    v = ScanOps<T>::add(ScanOps<T>::template dpp<0x118, 0xf>(v), v);
    v = ScanOps<T>::add(ScanOps<T>::template dpp<0x142, 0xa>(v), v);
    v = ScanOps<T>::add(ScanOps<T>::template dpp<0x143, 0xc>(v), v);
    return v;
}
// the value of the lane below (lane 0: the zero element)
template <typename T>
__device__ inline T wave_shift_up_one(T v)
{
    return ScanOps<T>::template dpp<0x138, 0xf>(v);
}

// Exclusive scan of one value per thread across a 256-thread block; also returns the
// block total.  `smem` must hold 4 T's.
template <typename T>
__device__ inline T block_exclusive_scan(T v, T *smem, T *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = wave_inclusive_scan<T>(v);
    if (lane == 63) smem[wave] = inc;
    __syncthreads();
    // (the same sums in the same order as adding every wave's total under a mask -- 0 + x is exact -- in about half the adds: the
    // offset's loop has a wave-uniform trip count; a K-wide double-double add is ~70 instructions and these scans are bound by
    // the instructions they issue)
    T woff = ScanOps<T>::zero();
    for (int w = 0; w < wave; ++w) woff = ScanOps<T>::add(woff, smem[w]);
    T tot = smem[0];
#pragma unroll
    for (int w = 1; w < SCAN_THREADS / 64; ++w) tot = ScanOps<T>::add(tot, smem[w]);
    __syncthreads();
    // exclusive = (wave offset) + (inclusive of previous lane)
    const T prev = wave_shift_up_one<T>(inc);
    *total = tot;
    return ScanOps<T>::add(woff, prev);
}

// Load functor: T operator()(int row, int64_t idx) -- value of element idx (< len).
template <typename T, typename Load>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(Load load, int64_t len, int nchunks,
                                                                  T *chunk_sums)
{
    __shared__ T smem[SCAN_THREADS / 64];
    const int row = blockIdx.y, chunk = blockIdx.x;
    const int64_t base = (int64_t)chunk * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    T acc = ScanOps<T>::zero();
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j)
        if (base + j < len) acc = ScanOps<T>::add(acc, load(row, base + j));
    T tot;
    (void)block_exclusive_scan<T>(acc, smem, &tot);
    if (threadIdx.x == 0) chunk_sums[(int64_t)row * nchunks + chunk] = tot;
}

// In-place exclusive scan of each row's chunk totals; one workgroup per row.
template <typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_spine_kernel(T *chunk_sums, int nchunks)
{
    __shared__ T smem[SCAN_THREADS / 64];
    T *row = chunk_sums + (int64_t)blockIdx.x * nchunks;
    const int per = (nchunks + SCAN_THREADS - 1) / SCAN_THREADS;
    const int b = threadIdx.x * per, e = min(b + per, nchunks);
    T acc = ScanOps<T>::zero();
    for (int i = b; i < e; ++i) acc = ScanOps<T>::add(acc, row[i]);
    T tot;
    T off = block_exclusive_scan<T>(acc, smem, &tot);
    for (int i = b; i < e; ++i) {
        T v = row[i];
        row[i] = off;
        off = ScanOps<T>::add(off, v);
    }
}

// Emit functor: void operator()(int row, int64_t idx, T exclusive, T inclusive).
// `fin` (optional) is called once per row by the thread that owns the last element
// with the row total: void fin(int row, T total).
template <typename T, typename Load, typename Emit>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(Load load, Emit emit, int64_t len, int nchunks,
                                                                 const T *chunk_offsets)
{
    __shared__ T smem[SCAN_THREADS / 64];
    const int row = blockIdx.y, chunk = blockIdx.x;
    const int64_t base = (int64_t)chunk * SCAN_CHUNK + (int64_t)threadIdx.x * SCAN_ITEMS;
    T v[SCAN_ITEMS];
    T acc = ScanOps<T>::zero();
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        v[j] = (base + j < len) ? load(row, base + j) : ScanOps<T>::zero();
        acc = ScanOps<T>::add(acc, v[j]);
    }
    T tot;
    T off = block_exclusive_scan<T>(acc, smem, &tot);
    off = ScanOps<T>::add(chunk_offsets[(int64_t)row * nchunks + chunk], off);
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        T inc = ScanOps<T>::add(off, v[j]);
        if (base + j < len) emit(row, base + j, off, inc);
        off = inc;
    }
}

inline int scan_num_chunks(int64_t len) { return (int)((len + SCAN_CHUNK - 1) / SCAN_CHUNK); }

// Runs the three phases on `stream`.  chunk_buf must hold rows * scan_num_chunks(len) T's.
template <typename T, typename Load, typename Emit>
inline hipError_t run_scan(hipStream_t stream, int rows, int64_t len, T *chunk_buf, Load load, Emit emit)
{
    if (rows <= 0 || len <= 0) return hipSuccess;
    const int nchunks = scan_num_chunks(len);
    dim3 grid(nchunks, rows);
    if (nchunks > 1) {
        hipLaunchKernelGGL((scan_reduce_kernel<T, Load>), grid, dim3(SCAN_THREADS), 0, stream, load, len, nchunks,
                           chunk_buf);
        hipLaunchKernelGGL((scan_spine_kernel<T>), dim3(rows), dim3(SCAN_THREADS), 0, stream, chunk_buf, nchunks);
    } else {
        hipError_t e = hipMemsetAsync(chunk_buf, 0, sizeof(T) * rows, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((scan_apply_kernel<T, Load, Emit>), grid, dim3(SCAN_THREADS), 0, stream, load, emit, len,
                       nchunks, chunk_buf);
    return hipGetLastError();
}

// Block-wide sum of doubles (256 threads); result valid in thread 0.
// log(x) in double, ~2 ulp: x = m 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s),
// s = (m - 1)/(m + 1), |s| <= 0.172, odd series to s^21.  About a third of the f64 instructions of libm's log (the
// tree kernels take two or three double logs per node and draw, and an f64 VALU op holds a SIMD for 8 cycles).
// fast_log for a normal positive argument (no zero / negative / infinite / NaN handling): the VI loop's forward kernel calls it
// on y and 1 - y with y clamped to [y_eps, 1 - y_eps] -- fourteen times per thread, and the four special cases were a tenth of it
__device__ inline double fast_log_pos(double x)
{
    int e;
    double m = frexp(x, &e);  // [0.5, 1)
    if (m < 0.70710678118654752440) {
        m *= 2.0;
        e -= 1;
    }
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double s = num * r;
    s = fma(fma(-den, s, num), r, s);
    const double s2 = s * s;
    double p = 1.0 / 21.0;
    p = fma(p, s2, 1.0 / 19.0);
    p = fma(p, s2, 1.0 / 17.0);
    p = fma(p, s2, 1.0 / 15.0);
    p = fma(p, s2, 1.0 / 13.0);
    p = fma(p, s2, 1.0 / 11.0);
    p = fma(p, s2, 1.0 / 9.0);
    p = fma(p, s2, 1.0 / 7.0);
    p = fma(p, s2, 1.0 / 5.0);
    p = fma(p, s2, 1.0 / 3.0);
    p = fma(p, s2, 1.0);
    const double ed = (double)e;
    return fma(ed, 0x1.62e42fee00000p-1, fma(ed, 0x1.a39ef35793c76p-33, 2.0 * s * p));
}

__device__ inline double fast_log(double x)
{
    int e;
    double m = frexp(x, &e);  // [0.5, 1)
    if (m < 0.70710678118654752440) {
        m *= 2.0;
        e -= 1;
    }
    const double num = m - 1.0, den = m + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    double s = num * r;
    s = fma(fma(-den, s, num), r, s);
    const double s2 = s * s;
    double p = 1.0 / 21.0;
    p = fma(p, s2, 1.0 / 19.0);
    p = fma(p, s2, 1.0 / 17.0);
    p = fma(p, s2, 1.0 / 15.0);
    p = fma(p, s2, 1.0 / 13.0);
    p = fma(p, s2, 1.0 / 11.0);
    p = fma(p, s2, 1.0 / 9.0);
    p = fma(p, s2, 1.0 / 7.0);
    p = fma(p, s2, 1.0 / 5.0);
    p = fma(p, s2, 1.0 / 3.0);
    p = fma(p, s2, 1.0);
    const double ed = (double)e;
    const double r0 = fma(ed, 0x1.62e42fee00000p-1, fma(ed, 0x1.a39ef35793c76p-33, 2.0 * s * p));
    // libm's edge cases: log(0) = -inf, log(x < 0) = log(nan) = nan, log(inf) = inf
    return x > 0.0 ? (x < HUGE_VAL ? r0 : x) : (x == 0.0 ? -HUGE_VAL : __builtin_nan(""));
}

// exp(x) in double, ~1 ulp, for the forward kernel's path sums (x = log u <= ~0): x = n ln 2 + r with |r| <= ln 2 / 2 (Cody-Waite
// in two fmas), the Taylor series to r^12 (the next term is 1.7e-16 of the result), v_ldexp.  20 instructions against libm's
// ~45 -- the forward kernel takes one per leaf and draw in lanes that idle two thirds of the time (a wave's leaves are a third
// of its tour entries), so every instruction of it is paid three times.  exp(x < -745.2) = 0 (and exp(-inf)); a NaN stays one.
__device__ inline double fast_exp(double x)
{
    const double n = rint(x * 0x1.71547652b82fep+0);
    double r = fma(n, -0x1.62e42fefa39efp-1, x);
    r = fma(n, -0x1.abc9e3b39803fp-56, r);
    double p = 1.0 / 479001600.0;
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double v = ldexp(p, (int)fmax(fmin(n, 2000.0), -2000.0));
    return x < -745.2 ? 0.0 : v;
}

__device__ inline double block_sum_f64(double v, double *smem4)
{
    // (a wave's sum by a DPP scan -- it lands in the last lane -- instead of a six-step shuffle tree through LDS)
    v = wave_inclusive_scan<double>(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 63) smem4[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += smem4[w];
    __syncthreads();
    return r;
}

}  // namespace polee
