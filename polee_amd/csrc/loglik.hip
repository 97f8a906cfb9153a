// Sparse fragment x transcript log-likelihood and gradient on gfx950.
// Replaces pAt_mul_B!/pAt_mulinv_B! (src/sparse.jl:6-40) and log_likelihood /
// factored_log_likelihood (src/likelihood.jl:36-85) with ONE pass over X that evaluates
// K expression vectors at once:
//     s_i[k] = sum_j X_ij x_j[k]
//     lp[k] += ks_i log s_i[k]
//     g_j[k] += X_ij ks_i / s_i[k]      (accumulated per tile in LDS, flushed once)
// The reference makes two passes (CSR for s, CSC for g) per draw, i.e. 2*K passes per VI
// step; this file makes one: loglik_stream_kernel, a persistent launch that streams the
// uniform slices of the PSELL layout (loglik_internal.hpp) through LDS rings.
// Roofline: HBM-bound by bytes (0.25 flop/B); the two small dense products per slice run on the
// exact-f32 matrix instruction because that removes the cross-lane sums, not for flops.
#include "loglik_internal.hpp"
#include "psell_device.hpp"
#include "wave.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <queue>
#include <type_traits>

namespace polee {

template <int CTRL, int ROW_MASK>
__device__ inline float dpp_mov0(float v)  // lanes without a source (or in masked-off rows) read 0
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}

// Adds q[k] of every lane into gw[c*K + k].  Lanes holding the same column id are contiguous
// (rows are pattern-sorted), so contributions are first summed per run of equal ids with a
// segmented DPP scan; only the last lane of each run touches LDS.
template <int K>
__device__ inline void scatter_runs(int c, float (&q)[K], float *gw, int lane)
{
    const int c0 = __builtin_amdgcn_readfirstlane(c);
    if (__all(c == c0)) {  // one column for the whole wavefront: plain wave sum
        float *gr = gw + c0 * K;
        wave_sum_to_lane63_n<K>(q);
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < K; ++k) atomicAdd(gr + k, q[k]);
        }
        return;
    }
    const int cprev = __builtin_amdgcn_update_dpp(-1, c, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const unsigned long long heads = __ballot(c != cprev);  // lane 0 compares with -1: always a head
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const int dist = lane - (63 - __clzll(heads & upto));  // distance to the head of this lane's run
    const int rl = lane & 15;
    const float m1 = dist >= 1 ? 1.f : 0.f, m2 = dist >= 2 ? 1.f : 0.f, m4 = dist >= 4 ? 1.f : 0.f,
                m8 = dist >= 8 ? 1.f : 0.f;
    const float mb15 = dist > rl ? 1.f : 0.f;           // run started in an earlier row of 16
    const float mb31 = dist > (lane & 31) ? 1.f : 0.f;  // run started before lane 32
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x111, 0xf>(q[k]), m1, q[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x112, 0xf>(q[k]), m2, q[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x114, 0xf>(q[k]), m4, q[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x118, 0xf>(q[k]), m8, q[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x142, 0xa>(q[k]), mb15, q[k]);  // row_bcast:15 -> rows 1, 3
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x143, 0xc>(q[k]), mb31, q[k]);  // row_bcast:31 -> rows 2, 3
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    if (tail) {
        float *gr = gw + c * K;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (q[k] != 0.0f) atomicAdd(gr + k, q[k]);
    }
}

// sacc[k] += v * row[k]; rows of K floats start 8-byte aligned when K is even -> 8-byte LDS reads
template <int K>
__device__ inline void fma_row(float v, const float *row, float (&sacc)[K])
{
    if constexpr (K % 2 == 0) {
        const float2 *r2 = reinterpret_cast<const float2 *>(row);
#pragma unroll
        for (int k = 0; k < K / 2; ++k) {
            const float2 x = r2[k];
            sacc[2 * k] = fmaf(v, x.x, sacc[2 * k]);
            sacc[2 * k + 1] = fmaf(v, x.y, sacc[2 * k + 1]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) sacc[k] = fmaf(v, row[k], sacc[k]);
    }
}

__device__ inline float fast_weight(float ksv, float s)
{
    // ks / s with v_rcp_f32 (1 ulp): well inside the 1e-4 budget, 10x fewer instructions than a division
    return s > 0.0f ? ksv * __builtin_amdgcn_rcpf(s) : 0.0f;  // padded lanes (and empty rows) have s = 0
}
// Row sums of the matrix-core path start from FLT_MIN instead of 0: a padded lane (no fragment: only zero values) then
// has s = FLT_MIN and a finite weight 1 / FLT_MIN, which multiplies zeros; a real row sum (values >= 1e-12, x >= 1e-16:
// s >= 1e-28) is unchanged by the addend, bit for bit.  The weight is then one v_rcp_f32, no compare or select.
constexpr float ROWSUM_FLOOR = 1.17549435e-38f;

// Sum of the logs of a lane's row sums WITHOUT a logarithm in the slice loop.  log s = exponent(s) ln 2 + log mantissa(s), so
// a lane keeps the PRODUCT of the mantissas (brought back into [0.5, 1) at every step) and the SUM of the exponents -- two
// v_frexp pairs, a multiply and two integer adds per row sum, no float64 -- and takes one logarithm when its share of the
// tile is done.  The mantissa product is rounded to float32 at every step: a relative 6e-8 per factor, i.e. an absolute
// ~1e-6 on the log of a lane's ~100 row sums, against log-likelihoods of 1e6 .. 1e9.  (The float64 log per row sum that
// stood here kept ~30 registers live inside the loop: with the masked streams in the same kernel the instances that
// return lp spilled, and a spill's reload drains the LDS-DMA ring -- the pass took 1.40 ms instead of 0.25.)
struct LogAcc {
    float m;  // in [0.5, 1)
    int e;
    __device__ inline void init()
    {
        m = 0.5f;
        e = 1;
    }
    __device__ inline void mul(float s, bool valid)  // *= s (valid) or *= 1
    {
        const float sv = valid ? s : 1.0f;
        const float p = m * __builtin_amdgcn_frexp_mantf(sv);
        m = __builtin_amdgcn_frexp_mantf(p);
        e += __builtin_amdgcn_frexp_expf(sv) + __builtin_amdgcn_frexp_expf(p);
    }
    // (the mantissa's log with the hardware log2 -- an absolute 1e-7, once per lane and tile: no float64 logarithm, with its
    // thirty live registers, anywhere in the kernel)
    __device__ inline double log_value() const { return ((double)e + (double)__builtin_amdgcn_logf(m)) * 0.693147180559945309417; }
};
// multiplicities: ks log s with the hardware log2 (1 ulp of float32), summed in float64; x ln 2 at the end
__device__ inline double ks_log2(float ksv, float s) { return (double)(ksv * __builtin_amdgcn_logf(s)); }

// ---- stream B (and fallback for very wide rows): mixed slices ---------------------------------------
// Two sweeps over each slice straight from global memory (the second one hits L1/L2); contributions
// are summed per run of equal transcript ids with DPP before touching LDS.  Few registers -> high
// occupancy hides the latency.
struct PsellArgs {
    const uint8_t *data;
    const uint32_t *slice_off, *tile_slice, *tile_dict, *dict;
    const float *slice_ks;
    const float *x;
    float *g;
    double *lp;
    int lcap;
    int tiles_a;  // tiles [0, tiles_a) use the compact uniform slice layouts
    // the persistent streaming kernel
    int tiles_a1;           // tiles [0, tiles_a1): stream A1 (dense narrow), [tiles_a1, tiles_a1m): A1M (masked narrow),
    int tiles_a1m, tiles_a2;  // [tiles_a1m, tiles_a2): A2 (dense wide), [tiles_a2, tiles_a): A2M (masked wide)
    int tiles_s;              // [tiles_a, tiles_s): BN (mixed narrow); the persistent launch's share ends here
    const float *xwin;      // x window of every tile: xwin[e * K + k] = x[dict[e]][k]
    const PosDesc *sched;   // [rounds + 1][grid] static schedule, POS_NONE-terminated columns
    // dynamic schedule (the default; the deterministic mode keeps the static one only when lp is wanted): the tiles in descending order of their
    // cost, POS_NONE behind them; workgroup b starts with positions b, b + G, b + 2 G and draws every further position
    // from the counter dyn_ctr[0] (+ 3 G).  The counter is NEVER reset: a launch makes exactly one draw per position of the list
    // (a workgroup draws once per tile it takes), so the next launch's draws start at dyn_base + positions -- the host keeps
    // dyn_base (round 5: the last workgroup used to reset the counter behind an acquire-release arrival count; that fence --
    // an L2 write-back and invalidate per WORKGROUP, 1 024 per launch -- cost 25 us of a 250 us pass, see launch_stream)
    const PosDesc *sched_dyn;
    unsigned int *dyn_ctr;
    unsigned int dyn_base;
    unsigned int dyn_last;  // the list's last (POS_NONE) slot: a draw is clamped to it, so that a counter that has fallen out of step
                            // with the host's base (an asynchronous fault, two evaluations of one handle at once) ends the
                            // workgroup's walk instead of indexing past the list
    // deterministic mode: every tile's window is stored (not added) and a second kernel sums the windows of a transcript
    float *gwin;            // [dict entries][K], laid out like xwin
    double *lpwin;          // [grid][K] per-workgroup log-likelihood sums, then [stream B's tiles][K] (lp_slot0 on)
    int lp_slot0;
};

template <int K, bool WANT_LP, bool HAS_KS>
__device__ inline void psell_tile_body(const PsellArgs &A, int tile, float *xw, float *gw, double *lp_red)
{
    const uint8_t *__restrict__ data = A.data;
    const uint32_t *__restrict__ slice_off = A.slice_off;
    const uint32_t *__restrict__ tile_slice = A.tile_slice;
    const uint32_t *__restrict__ tile_dict = A.tile_dict;
    const uint32_t *__restrict__ dict = A.dict;
    const float *__restrict__ slice_ks = A.slice_ks;
    const float *__restrict__ x = A.x;
    float *__restrict__ g = A.g;
    double *__restrict__ lp = A.lp;

    const uint32_t d0 = tile_dict[tile];
    const int L = (int)(tile_dict[tile + 1] - d0);
    for (int i = threadIdx.x; i < L * K; i += 256) {
        const int l = i / K, k = i - l * K;
        xw[i] = x[(size_t)dict[d0 + l] * K + k];
        gw[i] = 0.0f;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t s0 = tile_slice[tile], s1 = tile_slice[tile + 1];
    const bool compact = tile < A.tiles_a;
    double lpacc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) lpacc[k] = 0.0;

    const int stream = tile < A.tiles_a1 ? PSELL_A1 : (tile < A.tiles_a1m ? PSELL_A1M : (tile < A.tiles_a2 ? PSELL_A2 : (tile < A.tiles_a ? PSELL_A2M : (tile < A.tiles_s ? PSELL_BN : PSELL_B))));
    // Deterministic mode, stream B (rows of more than 32 transcripts: rare, a fraction of a per cent of the non-zeros where they
    // occur at all): ONE wave takes all slices of the tile in order -- a wave's LDS adds retire in program order, four waves' do
    // not -- and the tile's sums are STORED to its slots of gwin, which gwin_reduce_kernel adds in tile order like every other
    // tile's; its lp goes to a slot of lpwin.  (Round 5: these tiles added with float atomics, and a sample that had any was not
    // bitwise reproducible -- one gradient entry moving by an ulp between launches on a 3 M-fragment sample with 699 such rows.)
    const bool det = A.gwin != nullptr && tile >= A.tiles_s;
    for (uint32_t s = det ? (wave == 0 ? s0 : s1) : s0 + wave; s < s1; s += det ? 1u : 4u) {
        const uint32_t off = slice_off[s] & PSELL_OFF_MASK;
        const uint32_t units = (slice_off[s + 1] & PSELL_OFF_MASK) - off;
        // compact slices (uniform streams): uint16 lcol[128] header, then float val[w][64];
        // masked slices: one (A1M) or two (A2M) rows of uint32 hw[64] (low half: 16 bits of the lane's mask, high half:
        // transcript ids), then float val[i][64] = the lane's i-th non-zero;
        // mixed slices: float val[w][64]; uint16 lcol[w][64]
        const bool masked = stream == PSELL_A1M || stream == PSELL_A2M || ((slice_off[s] >> PSELL_FLAG_MASKED_BIT) & 1u) != 0;
        const int hrows = stream == PSELL_A2M ? 2 : 1;
        const int nrows = compact ? (int)(units / 2u) - hrows - (HAS_KS ? 1 : 0)  // (+ a ks row when factored: uniform streams and BN)
                                  : (int)((units - (HAS_KS && stream == PSELL_BN ? 2u : 0u)) / 3u);
        const uint16_t *hdr = reinterpret_cast<const uint16_t *>(data + (size_t)off * 128);
        int w = nrows;
        uint32_t mk = 0;
        auto hdr_id = [&](int t) -> uint16_t { return hdr[128 * (t >> 4) + 2 * (t & 15) + 1]; };  // (masked slices)
        if (masked) {
            w = 0;
            while (w < 16 * hrows && hdr_id(w) != PSELL_NO_COL) ++w;
            mk = hdr[2 * lane];
            if (hrows == 2) mk |= (uint32_t)hdr[128 + 2 * lane] << 16;
        }
        // (compact slices store element r of row t at position psell_row_pos(stream, t, r) of the row)
        const float *vbase = reinterpret_cast<const float *>(data + (size_t)off * 128 + (compact ? 256 * hrows : 0));
        auto vat = [&](int t) -> float {
            if (masked) return (mk >> t) & 1u ? vbase[__popc(mk & ((1u << t) - 1u)) * 64 + lane] : 0.0f;
            return vbase[t * 64 + (compact ? (int)psell_row_pos(stream, (uint32_t)t, (uint32_t)lane) : lane)];
        };
        const uint16_t *cols = compact ? hdr : reinterpret_cast<const uint16_t *>(data + (size_t)off * 128 + (size_t)w * 256) + lane;
        const int cstride = compact ? 1 : 64;
        auto cat = [&](int t) -> int { return masked ? (int)hdr_id(t) : (int)cols[t * cstride]; };

        // sweep 1: row sums s[k] = sum_t v[t] * x[c[t]][k]
        float sacc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) sacc[k] = 0.0f;
        int t = 0;
        for (; t + 4 <= w; t += 4) {
            float v[4];
            int c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = vat(t + u);
                c[u] = cat(t + u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) fma_row<K>(v[u], xw + c[u] * K, sacc);
        }
        for (; t < w; ++t) fma_row<K>(vat(t), xw + cat(t) * K, sacc);
        const float ksv = HAS_KS ? slice_ks[(size_t)s * 64 + lane] : 1.0f;
        float wk[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            wk[k] = fast_weight(ksv, sacc[k]);
            if (WANT_LP && sacc[k] > 0.0f) lpacc[k] += (double)ksv * log((double)sacc[k]);
        }
        // sweep 2 (slice is L1/L2 resident): g[c[t]][k] += v[t] * w[k], summed per run of equal ids
        for (t = 0; t < w; ++t) {
            const float v = vat(t);
            const int c = cat(t);
            float q[K];
#pragma unroll
            for (int k = 0; k < K; ++k) q[k] = v * wk[k];
            scatter_runs<K>(c, q, gw, lane);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < L * K; i += 256) {
        const int l = i / K, k = i - l * K;
        const float v = gw[i];
        if (det) A.gwin[(size_t)(d0 + l) * K + k] = v;
        else if (v != 0.0f) atomicAdd(g + (size_t)dict[d0 + l] * K + k, v);
    }
    if (WANT_LP) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double v = lpacc[k];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
            if (lane == 0) lp_red[wave] = v;
            __syncthreads();
            if (threadIdx.x == 0) {
                const double sum = lp_red[0] + lp_red[1] + lp_red[2] + lp_red[3];
                if (det) A.lpwin[(size_t)(A.lp_slot0 + (tile - A.tiles_s)) * K + k] = sum;
                else atomicAdd(lp + k, sum);
            }
            __syncthreads();
        }
    }
}

template <int K, bool WANT_LP, bool HAS_KS>
__global__ __launch_bounds__(256) void loglik_psell_kernel(PsellArgs A, int tile_base, const uint32_t *tile_ids)
{
    extern __shared__ float lds[];
    __shared__ double lp_red[4];
    const int tile = tile_ids ? (int)tile_ids[blockIdx.x] : tile_base + (int)blockIdx.x;
    psell_tile_body<K, WANT_LP, HAS_KS>(A, tile, lds, lds + (size_t)A.lcap * K, lp_red);
}

// ---- stream C: rows kept in CSR (fragments without any structure; loglik_internal.hpp) ------------------------------------------
// Lane = fragment: row sums by gathers of x rows from global memory (x is a few MB: L2), then one float atomic per entry
// and draw.  The general fallback -- any sparsity pattern, CSR's bytes -- and slow: nothing with structure ends up here.
template <int K, bool WANT_LP, bool HAS_KS>
__global__ __launch_bounds__(256) void loglik_csr_kernel(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ col,
                                                        const float *__restrict__ val, const float *__restrict__ ks, int64_t rows,
                                                        const float *__restrict__ x, float *__restrict__ g, double *__restrict__ lp)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double lpacc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) lpacc[k] = 0.0;
    if (i < rows) {
        const uint32_t b = rowptr[i], e = rowptr[i + 1];
        float sacc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) sacc[k] = 0.0f;
        for (uint32_t p = b; p < e; ++p) fma_row<K>(val[p], x + (size_t)col[p] * K, sacc);
        const float ksv = HAS_KS ? ks[i] : 1.0f;
        float wk[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            wk[k] = fast_weight(ksv, sacc[k]);
            if (WANT_LP && sacc[k] > 0.0f) lpacc[k] = (double)ksv * log((double)sacc[k]);
        }
        for (uint32_t p = b; p < e; ++p) {
            const float v = val[p];
            float *gr = g + (size_t)col[p] * K;
#pragma unroll
            for (int k = 0; k < K; ++k) atomicAdd(gr + k, v * wk[k]);
        }
    }
    if (WANT_LP) {
        __shared__ double red[4];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double v = lpacc[k];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
            __syncthreads();
            if (threadIdx.x == 0) atomicAdd(lp + k, red[0] + red[1] + red[2] + red[3]);
            __syncthreads();
        }
    }
}

// ---- the uniform streams in ONE persistent launch --------------------------------------------------------------
// A uniform slice holds up to 64 fragments (one per lane) that share ONE transcript set (c_0..c_{w-1}); runs of
// consecutive slices with the same set are marked by the builder.  For such a slice V[t][r] (w x 64):
//     S[r][k]  = sum_t V[t][r] x[c_t][k]          (phase 1)
//     G[t][k] += sum_r V[t][r] ks_r / S[r][k]     (phase 2 = the transpose of phase 1)
// Both run on the matrix cores (exact-f32 MFMA), the weights never leave the registers, and the partial G stays
// in registers for the whole run; it is added to the tile's LDS window when the run ends.
//
// Streaming: each wave owns a contiguous byte range of the tile's slice stream and pulls it through a private LDS
// ring with `global_load_lds_dwordx4` (1 KiB per wave-instruction, no VGPR destination), ring-size ahead; HBM sees
// every byte of X exactly once.
//
// Persistence: the grid is (workgroups per CU) x (CUs); every workgroup walks its own column of a static schedule
// (sched[pos], pos = block + round * grid: tiles dealt in snake order of their cost, small tiles spread between the
// large ones).  While a tile streams, everything the NEXT tile needs arrives in the background by LDS-DMA: its x
// window (one contiguous piece of xwin, gathered once per pass by xwin_gather_kernel), the transcript ids of its
// dictionary (for the flush) and the slice offsets of each wave's share; a wave that has finished its slices
// starts the next tile's ring before the workgroup's barrier.  A tile therefore has no dependent global latency in
// front of its first slice; the counted `s_waitcnt vmcnt(N)` of the slice loop account for these extra operations.
#ifndef POLEE_DMA_POLICY
#define POLEE_DMA_POLICY " nt"  // X is read once per pass: non-temporal keeps it from evicting x / g lines
#endif
// a wave-uniform pointer the compiler cannot prove uniform, as an SGPR pair
__device__ inline const void *uniform_ptr(const void *p)
{
    const uint64_t a = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    return (const void *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
// LDS-DMA forms: wave-uniform 64-bit base in SGPRs + a 32-bit byte offset per lane; the LDS destination is
// M0 + lane * (bytes per lane).  The leading s_nop 4 covers a base that has just come out of v_readfirstlane
// (VALU-written SGPR -> VMEM read: 5 wait states, which the compiler does not insert for an asm statement).  (M0 is not restored, and it
// cannot be declared clobbered: hipcc treats M0 as a reserved register and warns that such a clobber is not honoured.
// Nothing else in these kernels uses it -- LDS instructions on gfx9 do not, there is no movrel / sendmsg -- and the base /
// offset registers are not rewritten per piece.)
#ifdef POLEE_DMA_BUILTIN
// A/B build (`make dmabuiltin`): the LDS-DMA through the compiler's builtin, which models M0 (no
// unmodelled write) but takes a per-lane 64-bit global address instead of SGPR base + 32-bit lane offset.
__device__ inline void dma_builtin(const void *base_uniform, uint32_t voff, uint32_t lds_dst_any, int bytes16)
{
    typedef __attribute__((address_space(3))) void *lds_vp;
    typedef const __attribute__((address_space(1))) void *glb_vp;
    const uint32_t lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_any);
    glb_vp src = (glb_vp)(reinterpret_cast<const char *>(base_uniform) + voff);
    if (bytes16)
        __builtin_amdgcn_global_load_lds(src, (lds_vp)(uintptr_t)lds_dst, 16, 0, 0);
    else
        __builtin_amdgcn_global_load_lds(src, (lds_vp)(uintptr_t)lds_dst, 4, 0, 0);
}
__device__ inline void dma_1k(const void *b, uint32_t v, uint32_t l) { dma_builtin(b, v, l, 1); }
__device__ inline void dma_1k_keep(const void *b, uint32_t v, uint32_t l) { dma_builtin(b, v, l, 1); }
__device__ inline void dma_256(const void *b, uint32_t v, uint32_t l) { dma_builtin(b, v, l, 0); }
#else
__device__ inline void dma_1k(const void *base_uniform, uint32_t voff, uint32_t lds_dst_any)
{
    const uint32_t lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_any);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" POLEE_DMA_POLICY
                 :
                 : "s"(base_uniform), "v"(voff), "s"(lds_dst)
                 : "memory");
}
// the same with the default cache policy (x windows: written by the previous kernel)
__device__ inline void dma_1k_keep(const void *base_uniform, uint32_t voff, uint32_t lds_dst_any)
{
    const uint32_t lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_any);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" : : "s"(base_uniform), "v"(voff), "s"(lds_dst) : "memory");
}
// 4 bytes per lane: 64 dwords -> 256 contiguous LDS bytes
__device__ inline void dma_256(const void *base_uniform, uint32_t voff, uint32_t lds_dst_any)
{
    const uint32_t lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_any);
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" : : "s"(base_uniform), "v"(voff), "s"(lds_dst) : "memory");
}
#endif
__device__ inline uint32_t lds_addr(const void *p)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) const char *)p);
}

// waits until at most `allowed` of this wave's vector-memory operations are outstanding (rounded down
// to an encodable step: waiting for fewer outstanding operations is always safe)
__device__ inline void wait_vm_outstanding(int allowed)
{
    if (allowed >= 8) {
        if (allowed >= 14)
            asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else if (allowed >= 12)
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (allowed >= 10)
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (allowed >= 4) {
        if (allowed >= 7)
            asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else if (allowed >= 6)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (allowed >= 5)
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        if (allowed >= 3)
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (allowed >= 2)
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (allowed >= 1)
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// This lane's index in its wave, computed where it is used (two instructions): a lane index kept in a register across
// the tile loop is spilled under the slice loop's register pressure, and its reload comes with an s_waitcnt vmcnt(0) that
// drains the LDS-DMA ring which has just been started.
__device__ inline int wave_lane()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// workgroup barrier for LDS hand-offs only: waits for this wave's LDS operations, not for its vector-memory queue
// (__syncthreads() would drain the LDS-DMA ring that has just been started for the next tile)
__device__ inline void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef POLEE_STAMPS
// diagnostic build only: where does a wave of the streaming kernel spend its cycles?
__device__ unsigned long long g_stamps[24];

#define STAMP(i)                                                          \
    do {                                                                  \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime();    \
        st_acc[i] += now__ - st_last;                                     \
        st_last = now__;                                                  \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#ifdef POLEE_TILE_CYCLES
// second diagnostic build (cheap: one clock read per tile): how long does every tile / every workgroup take?
__device__ unsigned long long g_tile_cycles[1 << 17];  // per tile: wave 0's time from the previous tile's end to this tile's end
__device__ unsigned long long g_wg_cycles[4096];       // per workgroup: wave 0's time in the kernel
#endif
constexpr int NSTAMP = 16;  // (diagnostic build)

// what a wave knows about its share of the current uniform tile
struct WaveStream {
    uint32_t ent;         // slice offset (+ flags) of slice sb + lane, one per lane
    int nsl;              // slices owned by this wave
    int npieces;          // 1 KiB pieces of its byte range
    int issued, islot;    // pieces requested so far; ring slot of the next one
    int primed;           // pieces requested before the tile's loop started
    const uint8_t *gsrc;  // start of the byte range (wave-uniform)
};

template <uint32_t RB>
__device__ inline void ring_refill(WaveStream &ws, uint32_t ring_lds, int target)
{
    constexpr int RP = (int)(RB / 1024u);
    // (wave-uniform counters: say so, or the loop below is compiled as a divergent loop on vector registers)
    int issued = __builtin_amdgcn_readfirstlane(ws.issued), islot = __builtin_amdgcn_readfirstlane(ws.islot);
    target = __builtin_amdgcn_readfirstlane(target);
    const uint32_t voff = (uint32_t)wave_lane() * 16u;
    const uint8_t *src = reinterpret_cast<const uint8_t *>(uniform_ptr(ws.gsrc + (size_t)issued * 1024));
    for (; issued < target; ++issued) {
        dma_1k(src, voff, ring_lds + (uint32_t)islot * 1024u);
        src += 1024;
        islot = islot + 1 == RP ? 0 : islot + 1;
    }
    ws.issued = issued;
    ws.islot = islot;
}

// The slice loop of one wave over its share of a tile of the WIDE stream (A2: transcript sets of 17..32; 16 x 16 x 4 matrix
// tiles -- the narrow stream has its own loop, narrow_stream below).  `extras` = vector-memory operations issued AFTER the
// primed ring pieces and before the first refill (the previous tile's flush, the next tile's prefetch): they are younger than
// the primed pieces and older than every other piece, so only waits for primed pieces have to allow for them.
//
// Round 5: ALL FOUR waves of the workgroup work on a wide tile, each with the narrow streams' 7 KiB ring (until round 4 two
// waves with 14 KiB rings did, the other two waited at the tile's barrier: 15 % of all wave time on the SURVEY 8(d)
// generator, 40 % on inputs of wide sets).  A wide slice is up to 2 + 32 rows of 256 bytes and does not fit such a ring
// together with any look-ahead, so it passes through in TWO STAGES:
//   stage A   header + transcripts 0..15 (4 352 bytes): the phase-1 operands of steps 0..3 AND the phase-2 operands of the
//             first 16-row tile are read into registers (32), the bytes are released, the DMA for what follows goes out,
//             phase 1 runs over the first sixteen transcripts;
//   stage B   transcripts 16..w-1 (+ the multiplicities): phase 1 over them, the weights, the phase-2 operands of the
//             second tile (in the registers the phase-1 operands have left), release, then both tiles' phase 2.
// At any time the ring holds at most 4 352 bytes of the current slice + the look-ahead.
template <int K, uint32_t RB, bool WANT_LP, bool HAS_KS>
__device__ inline void wide_stream(WaveStream &ws, const char *ring, int extras, const float *xw, float *gw, double &lpacc, int dbg
#ifdef POLEE_STAMPS
                                   , unsigned long long (&st_acc)[NSTAMP], unsigned long long &st_last
#endif
                                   )
{
    constexpr int RP = (int)(RB / 1024u);
    constexpr uint32_t STAGE_A = 256u + 16u * 256u;  // header + sixteen rows
    static_assert(STAGE_A + 2046u <= RB && (PSELL_WIDE_MAX - 16 + 1) * 256u + 2046u <= RB, "a stage of a wide slice (+ ks row) must fit the ring");
    // (computed here, opaquely: the lane constants below are then recomputed per tile -- a few dozen instructions --
    // instead of being hoisted out of the kernel's tile loop, kept alive across it and spilled)
    const int lane = wave_lane();
    const uint32_t ring_lds = lds_addr(ring);

    // lane l = (tt = l & 15, q = l >> 4).  A slice is a (w x 64) block V[t][r]:
    //   phase 1   S[r][k] = sum_t V[t][r] x[c_t][k]      M = r (4 tiles: tile e holds rows 4 i + e), N = k, inner = t
    //       A[i][kk=q] = V[4 step + q][4 i + e]  -- one 16-byte LDS read per step feeds the 4 row tiles
    //       B[kk=q][tt] = x[c_{4 step + q}][tt]  -- constant over a run, kept in registers (xq)
    //       D1[e]: lane (tt, q), register v  =  S[16 q + 4 v + e][tt]
    //   weights   W = ks / S, in place (v_rcp_f32)
    //   phase 2   G[t][k] += sum_r V[t][r] W[k][r]      M = t, N = k, inner = r = 16 q + (0..15)
    //       A[tt][kk=q] = V[16 mt + tt][16 q + 4 j + e]   -- four 16-byte LDS reads per 16-row tile mt
    //       B[kk=q][tt] = W[tt][16 q + 4 j + e]   = D1[e][j] of THIS lane: the weights never leave the registers
    //       D2 (rows 4 q + v, column tt) stays in registers for the whole run of slices sharing the set.
    // Rows are stored ROTATED (element r of row t at position (r + 4 t) & 63) so that the 16 lanes of every
    // 16-byte read hit 16 different bank groups.  Columns tt >= K are padding (B = 0 there).
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int NT = 2;  // 16-row tiles of transcripts (phase 2)
    constexpr int NS = 8;  // steps of 4 transcripts (phase 1)
    const int tt = lane & 15, q = lane >> 4;
    f32x4 acc0[NT], acc1[NT];  // two accumulation chains per tile (dependent MFMA latency 40 > issue 32)
    uint2 colq[NT];            // tile-local ids of transcripts 16 mt + 4 q + (0..3) of the current run, 16 bit each
    float xq[NS];              // x[c_{4 step + q}][tt] of the current run
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        acc0[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        colq[mt] = make_uint2(0u, 0u);
    }
#pragma unroll
    for (int st = 0; st < NS; ++st) xq[st] = 0.0f;
    int pend_w = 0;  // transcripts of the current run (0: no run open)
    LogAcc lpl;       // (WANT_LP) log of the product of this lane's row sums; with multiplicities: lp2 = sum ks log2 s
    lpl.init();
    double lp2 = 0.0;
    auto flush = [&]() {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
            if (16 * mt < pend_w) {
                const f32x4 sum = acc0[mt] + acc1[mt];
                const unsigned cid[4] = {colq[mt].x & 0xffffu, colq[mt].x >> 16, colq[mt].y & 0xffffu, colq[mt].y >> 16};
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (tt < K && 16 * mt + 4 * q + v < pend_w && sum[v] != 0.0f && !(dbg & 2))
                        atomicAdd(gw + cid[v] * K + tt, sum[v]);
                acc0[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc1[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        pend_w = 0;
    };

    // Operand addresses.  A slice starts at a multiple of 256 bytes and the ring holds whole 256-byte rows, so a row never
    // straddles the ring's end: address = ring + wrap(stage base + 256 x row) + (offset inside the row).  Rows t >= w of a
    // step (B = 0 there) or of the second tile (their D2 rows are never used) read whatever follows the slice in the ring --
    // stream bytes or the zeros the ring was initialised with, always finite.
    const uint32_t q256 = 256u * (uint32_t)q;
    auto k2f = [&](int st) -> uint32_t { return ring_lds + (uint32_t)((tt + 4 * st + q) & 15) * 16u; };  // phase 1, step st: row 4 st + q, chunk (tt + row) & 15
    auto c2f = [&](int mt, int j) -> uint32_t { return ring_lds + (uint32_t)((4 * q + j + 16 * mt + tt) & 15) * 16u; };  // phase 2, tile mt: row 16 mt + tt
    auto lds_f4 = [](uint32_t a) -> f32x4 {
        return *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>((uintptr_t)a);
    };
    auto wrap = [&](uint32_t a) -> uint32_t { return min(a, a - RB); };  // a < 2 RB: a mod RB (unsigned wrap-around)

    uint32_t pos = 0;    // byte offset of the ring's tail inside this wave's range (a slice start, or STAGE_A behind one)
    uint32_t pos_r = 0;  // pos modulo the ring size
    auto wait_for = [&](uint32_t upto) {  // bytes [pos, pos + upto) of the wave's range must have landed
        const int need = (int)((pos + upto + 1023u) >> 10);
        if (ws.issued < need) ring_refill<RB>(ws, ring_lds, need);  // (only with a shortened look-ahead: experiments)
        // (wave-uniform: said explicitly, or the ladder below is compiled with vector compares and exec masks)
        int allowed = ws.issued - need + (need <= ws.primed ? extras : 0);
        wait_vm_outstanding(__builtin_amdgcn_readfirstlane(allowed));
    };
    auto release = [&](uint32_t nbytes) {  // the first nbytes behind the tail are consumed: the DMA for what follows goes out
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pos += nbytes;
        pos_r += nbytes;
        pos_r = pos_r >= RB ? pos_r - RB : pos_r;
        ring_refill<RB>(ws, ring_lds, min(ws.npieces, (int)(pos >> 10) + RP));
    };
    for (int si = 0; si < ws.nsl; ++si) {
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)ws.ent, si);
        const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)ws.ent, si + 1);
        const uint32_t off = e0 & PSELL_OFF_MASK, off_next = e1 & PSELL_OFF_MASK;
        const int flags = (int)(e0 >> 30);
        const uint32_t units = off_next - off;
        const int w = (int)(units / 2u) - 1 - (HAS_KS ? 1 : 0);  // 256-byte header (column ids) + w rows of 64 values (+ ks row)
        const uint32_t bytes = units * 128u;
        const bool two = w > 16;  // (always, as the builders fill this stream; a narrower slice passes in one stage)
        if (!(dbg & 16)) __builtin_amdgcn_s_setprio(3);  // the short non-matrix sections of a slice win the issue arbitration
        STAMP(1);  // slice bookkeeping
        wait_for(two ? STAGE_A : bytes);
        STAMP(2);  // waiting for the DMA
        if (pend_w != 0 && !(flags & 2)) flush();
        if (pend_w == 0) {  // a new run: the tile-local ids of its transcripts and their x rows (all in the header)
            const char *hdr = ring + pos_r;
#pragma unroll
            for (int mt = 0; mt < NT; ++mt) colq[mt] = *reinterpret_cast<const uint2 *>(hdr + 32 * mt + 8 * q);
            int cl[NS];
#pragma unroll
            for (int st = 0; st < NS; ++st) cl[st] = *reinterpret_cast<const uint16_t *>(hdr + 2 * min(4 * st + q, w - 1));
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                const float xv = xw[cl[st] * K + min(tt, K - 1)];
                xq[st] = (tt < K && 4 * st + q < w) ? xv : 0.0f;
            }
        }
        STAMP(3);  // run change: flush + column lookup

        f32x4 d1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) d1[e] = f32x4{ROWSUM_FLOOR, ROWSUM_FLOOR, ROWSUM_FLOOR, ROWSUM_FLOOR};
        f32x4 av1[4], av2a[4], av2b[4];
        auto phase1 = [&](int st0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (4 * (st0 + u) < w) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) d1[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[u][e], xq[st0 + u], d1[e], 0, 0, 0);
                }
            }
        };
        // ---- stage A: transcripts 0..15
        {
            const uint32_t rows = pos_r + 256u;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (4 * u < w) av1[u] = lds_f4(wrap(rows + 1024u * (uint32_t)u + q256) + k2f(u));
            const uint32_t row = wrap(rows + 256u * (uint32_t)tt);
#pragma unroll
            for (int j = 0; j < 4; ++j) av2a[j] = lds_f4(row + c2f(0, j));
        }
        // weights, in place: d1[e][v] belongs to fragment r = 16 q + 4 v + e, draw tt.  With multiplicities (the slice's last row:
        // ks of fragments 16 q + 4 v + (0..3) at `kr`) they are read four at a time, so the weights come BEFORE the release.
        auto weights = [&](uint32_t kr) {
            if (HAS_KS) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4 kv = lds_f4(ring_lds + kr + 16u * (uint32_t)v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sv = d1[e][v];
                        if (WANT_LP && sv > 2.0f * ROWSUM_FLOOR) lp2 += ks_log2(kv[e], sv);
                        d1[e][v] = kv[e] * __builtin_amdgcn_rcpf(sv);  // (padded lanes: ks = 0)
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const float sv = d1[e][v];
                        if (WANT_LP) lpl.mul(sv, sv > 2.0f * ROWSUM_FLOOR);
                        d1[e][v] = __builtin_amdgcn_rcpf(sv);
                    }
            }
        };
        if (two) {
            release(STAGE_A);
            STAMP(6);  // refill
            if (!(dbg & 16)) __builtin_amdgcn_s_setprio(0);
            phase1(0);
            // ---- stage B: transcripts 16..w-1; the tail now stands at row 16
            if (!(dbg & 16)) __builtin_amdgcn_s_setprio(3);
            STAMP(14);  // phase 1 MFMAs
            wait_for(bytes - STAGE_A);
            STAMP(2);  // waiting for the DMA
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (4 * (4 + u) < w) av1[u] = lds_f4(wrap(pos_r + 1024u * (uint32_t)u + q256) + k2f(4 + u));
            if (!(dbg & 16)) __builtin_amdgcn_s_setprio(0);
            phase1(4);
            // the second tile's phase-2 operands (rows t >= w: whatever lies there), in the registers the phase-1 operands have left
            const uint32_t row = wrap(pos_r + 256u * (uint32_t)tt);
#pragma unroll
            for (int j = 0; j < 4; ++j) av2b[j] = lds_f4(row + c2f(1, j));
            if (HAS_KS) weights(wrap(pos_r + 256u * (uint32_t)(w - 16)) + 64u * (uint32_t)q);
            release(bytes - STAGE_A);
            STAMP(6);  // refill
            if (!HAS_KS) weights(0u);
        } else if (HAS_KS) {
            if (!(dbg & 16)) __builtin_amdgcn_s_setprio(0);
            phase1(0);
            weights(wrap(pos_r + 256u + 256u * (uint32_t)w) + 64u * (uint32_t)q);
            release(bytes);
            STAMP(6);  // refill
        } else {
            release(bytes);
            STAMP(6);  // refill
            if (!(dbg & 16)) __builtin_amdgcn_s_setprio(0);
            phase1(0);
            weights(0u);
        }
        STAMP(4);  // phase 1 + weights
        if (!(dbg & 2)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2a[j][0], d1[0][j], acc0[0], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2a[j][1], d1[1][j], acc1[0], 0, 0, 0);
                acc0[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2a[j][2], d1[2][j], acc0[0], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2a[j][3], d1[3][j], acc1[0], 0, 0, 0);
            }
            if (two) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc0[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2b[j][0], d1[0][j], acc0[1], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2b[j][1], d1[1][j], acc1[1], 0, 0, 0);
                    acc0[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2b[j][2], d1[2][j], acc0[1], 0, 0, 0);
                    acc1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av2b[j][3], d1[3][j], acc1[1], 0, 0, 0);
                }
            }
        }
        pend_w = w;
        STAMP(5);  // phase 2
    }
    if (pend_w != 0) flush();
    if (WANT_LP) lpacc += HAS_KS ? lp2 * 0.693147180559945309417 : lpl.log_value();
}


// ---- the slice loop of the narrow stream (transcript sets of <= 16): batched outer products ---------------------------
// With K <= 8 draws the 16-column MFMA tile of wide_stream is at most half used.  v_mfma_f32_4x4x1_16b_f32 computes
// sixteen independent 4 x 4 outer products D[b][i][j] += A[b][i] B[b][j] (A in lane 4 b + i, B in lane 4 b + j, D in
// register i of lane 4 b + j) in 8 cycles, a quarter of the 16 x 16 x 4 tile's 32, and fits the problem exactly:
//   phase 1   S[r][k] = sum_t V[t][r] x[c_t][k]    block b = fragments 4 b + (0..3), i = fragment, j = draw (k = 4 kg + j,
//             kg < ceil(K / 4)): one instruction per transcript t and draw group, A = V[t][lane] -- one 4-byte LDS read
//             per lane and transcript -- and B = x[c_t][4 kg + (lane & 3)], constant over a run (registers);
//             d1[kg], register i of lane (b, j) = S[4 b + i][4 kg + j]
//   weights   W = ks / S in place (v_rcp_f32): 4 ceil(K / 4) reciprocals per lane instead of 16
//   phase 2   G[t][k] += sum_r V[t][r] W[r][k]     per group g of 4 transcripts and fragment index i: block b adds the
//             outer product of V[4 g + (0..3)][4 b + i] with W[4 b + i][4 kg + (0..3)] -- B is d1[kg][i] as it stands,
//             A is a 4-byte LDS read of row 4 g + (lane & 3) -- into acc[g][kg]; the sixteen blocks hold partial sums
//             over their four fragments, added across lanes (DPP) when the run is flushed.
// MFMA issue cycles of a slice of w transcripts at K = 6:  64 ceil(w / 4) + 64 ceil(w / 4)  against  128 ceil(w / 4) +
// 256 or 512 before.  Element r of row t is stored at position r ^ (t & 3) of the row (psell_row_pos, stream 0): both
// read patterns -- lane l reads position l ^ (t & 3) of row t; lane (b, i') reads position (4 b + i) ^ i' of row
// 4 g + i' -- then touch 64 different banks.
//
// The loop is written for a short instruction stream (a wave issues at most one instruction every four cycles, and the
// bookkeeping around the matrix instructions was three quarters of what it issued):
//   * the slice body exists in 4 x 2 straight-line versions: groups of four transcripts (1..4) x "the slice's bytes are
//     contiguous in the ring" (every operand address is one per-slice base register plus an immediate offset) or "they
//     wrap around its end" (a third of the slices: addresses computed row by row);
//   * at a run's start lanes t < 16 compute, once, the LDS addresses of transcript t's x row and gradient row (rows
//     t >= w of the last group point at a row of zeros / at a scratch row: no masks in the loads and in the flush);
//   * the flush adds a group's four transcripts under one exec mask.
// `aux_lds`: LDS address of 32 bytes of zeros followed by 32 bytes of scratch.
// 4 x 4 transpose inside every quad of lanes: lane j of a quad ends up with register j of the quad's lanes 0..3 in
// a0..a3 (out[i] of lane j = in[j] of lane i).  Two rounds of "keep or take the neighbour's" with DPP quad permutes
// folded into the selects (v_cndmask_b32_dpp: D = vcc ? src1 : permuted src0) -- 8 vector instructions for 16 values.
// (Inline assembly: the compiler's hazard recogniser does not look inside, so the block starts with the two wait states a
// DPP read of a freshly written VGPR needs; inside it every such pair is at least two instructions apart.)
__device__ inline void quad_transpose(float &a0, float &a1, float &a2, float &a3)
{
    float y0, y1, y2, y3;
    const uint64_t E = 0x5555555555555555ull, L = 0x3333333333333333ull;  // lanes with bit 0 / bit 1 of their index clear
    asm volatile("s_nop 1\n\t"
                 "s_mov_b64 vcc, %8\n\t"
                 "v_cndmask_b32_dpp %4, %1, %0, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  // y0 = even ? a0 : a1 of lane ^ 1
                 "v_cndmask_b32_dpp %6, %3, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  // y2 = even ? a2 : a3 of lane ^ 1
                 "s_not_b64 vcc, vcc\n\t"
                 "v_cndmask_b32_dpp %5, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  // y1 = odd ? a1 : a0 of lane ^ 1
                 "v_cndmask_b32_dpp %7, %2, %3, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"  // y3 = odd ? a3 : a2 of lane ^ 1
                 "s_mov_b64 vcc, %9\n\t"
                 "v_cndmask_b32_dpp %0, %6, %4, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"  // z0 = low ? y0 : y2 of lane ^ 2
                 "v_cndmask_b32_dpp %1, %7, %5, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"  // z1 = low ? y1 : y3 of lane ^ 2
                 "s_not_b64 vcc, vcc\n\t"
                 "v_cndmask_b32_dpp %2, %4, %6, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"  // z2 = high ? y2 : y0 of lane ^ 2
                 "v_cndmask_b32_dpp %3, %5, %7, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"        // z3 = high ? y3 : y1 of lane ^ 2
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3)
                 : "s"(E), "s"(L)
                 : "vcc", "scc");
}

// MASKED: the tile's slices are masked slices (stream A1M, loglik_internal.hpp): a fragment's values are stored packed
// (its i-th non-zero in row i) with a 16-bit mask of the union's transcripts it has.  Lane r expands its own fragment
// for phase 1 -- V[t][r] = bit t of its mask ? row (number of lower mask bits) : 0, one LDS read at a running address per
// transcript, conflict free because every lane reads its own column -- and phase 2's operands V[4 g + j][4 b + i] are
// exactly the quad transposes of the phase-1 registers: no second LDS pass, and the slice's ring bytes are free before
// the first matrix instruction.
template <int K, uint32_t RB, bool WANT_LP, bool HAS_KS, bool MASKED>
__device__ inline void narrow_stream(WaveStream &ws, const char *ring, int extras, const float *xw, float *gw,
                                     uint32_t aux_lds, double &lpacc, int dbg
#ifdef POLEE_STAMPS
                                     , unsigned long long (&st_acc)[NSTAMP], unsigned long long &st_last
#endif
                                     )
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) float *lds_cfp;
    typedef __attribute__((address_space(3))) float *lds_fp;
    constexpr int WMAX = PSELL_NARROW_MAX;
    static_assert(WMAX == 16, "four groups of four transcripts");
    constexpr int KG = (K + 3) / 4;
    constexpr int RP = (int)(RB / 1024u);
    const int lane = wave_lane();  // (see wide_stream: per-tile lane constants instead of spilled ones)
    const uint32_t ring_lds = lds_addr(ring);
    const uint32_t xw_lds = lds_addr(xw), gw_lds = lds_addr(gw);
    const int j = lane & 3, b = lane >> 2;
    const uint32_t j4 = 4u * (uint32_t)j;
    // phase 1 reads position lane ^ (t & 3) of row t; phase 2, fragment index u, position (4 b + u) ^ j = lane ^ u of
    // row 4 g + j: the same four lane constants, plus 256 j
    uint32_t lc1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) lc1[u] = ring_lds + ((uint32_t)(lane ^ u) << 2);
    const uint32_t j256 = 256u * (uint32_t)j;
    auto lds_f = [](uint32_t a) -> float { return *reinterpret_cast<lds_cfp>((uintptr_t)a); };
    auto wrap_u = [&](uint32_t a) -> uint32_t {  // uniform ring offset a < 2 RB
        a = (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
        return a >= RB ? a - RB : a;
    };

    f32x4 acc[4][KG];  // acc[g][kg], register v of lane (b, j): block b's part of G[4 g + v][4 kg + j]
    float xq[8][KG];   // x[c_t][4 kg + j] of the run's first eight transcripts (the others': re-read per slice, see below)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) acc[g][kg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) xq[t][kg] = 0.0f;
    uint32_t xav = aux_lds, gav = aux_lds + 32u;  // lane t < 16: LDS addresses of the x row / gradient row of the run's transcript t
    int pend_w = 0;
    int run_w = 0;  // (MASKED) transcripts of the current run's union
    LogAcc lpl[KG];   // (WANT_LP) per draw group: log of the product of this lane's row sums (LogAcc above)
    double lp2[KG];   // ... with multiplicities: sum of ks log2 s
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        lpl[kg].init();
        lp2[kg] = 0.0;
    }

    // Flush of a group of four transcripts: acc[g][kg][v], lane (b, j) is block b's part of G[4 g + v][4 kg + j].  The
    // four registers v are first added across the wave's four rows of 16 lanes and scattered (gfx950 lane swaps: row r
    // is left with transcript VROW[r]'s sums, block by block), then across a row's four blocks (DPP): lanes 12..15 of
    // every row hold one finished sum each and ONE LDS add per (group, draw group) writes 16 different addresses.
    // (Adding the rows' partial sums to the same addresses instead -- four-way conflicts in four times as many LDS
    // atomics -- cost a quarter of the kernel's time.)
    auto flush_group = [&](int g) {
        const uint32_t ga0 = (uint32_t)__builtin_amdgcn_readlane((int)gav, 4 * g + 0), ga2 = (uint32_t)__builtin_amdgcn_readlane((int)gav, 4 * g + 2);
        const uint32_t ga1 = (uint32_t)__builtin_amdgcn_readlane((int)gav, 4 * g + 1), ga3 = (uint32_t)__builtin_amdgcn_readlane((int)gav, 4 * g + 3);
        // row 0: transcript 4 g, row 1: 4 g + 2, row 2: 4 g + 1, row 3: 4 g + 3
        const uint32_t ga = ((lane & 32) ? ((lane & 16) ? ga3 : ga1) : ((lane & 16) ? ga2 : ga0)) + j4;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[g][kg][0]), __float_as_uint(acc[g][kg][1]), false, false);
            const float ab = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);
            auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[g][kg][2]), __float_as_uint(acc[g][kg][3]), false, false);
            const float cd = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
            auto r3 = __builtin_amdgcn_permlane16_swap(__float_as_uint(ab), __float_as_uint(cd), false, false);
            float q = __uint_as_float(r3[0]) + __uint_as_float(r3[1]);
            q += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), 0x114, 0xf, 0xf, true));  // row_shr:4
            q += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), 0x118, 0xf, 0xf, true));  // row_shr:8
            if ((lane & 12) == 12 && 4 * kg + j < K && !(dbg & 2))
                __hip_atomic_fetch_add(reinterpret_cast<lds_fp>((uintptr_t)(ga + 16u * (uint32_t)kg)), q, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            acc[g][kg] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto flush = [&]() {
        flush_group(0);
        if (pend_w > 4) flush_group(1);
        if (pend_w > 8) flush_group(2);
        if (pend_w > 12) flush_group(3);
        pend_w = 0;
    };
    auto load_x_group = [&](int g, float (*dst)[KG]) {  // x[c_t][4 kg + j] of transcripts 4 g .. 4 g + 3
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)xav, 4 * g + u) + j4;
            dst[u][0] = lds_f(a);
            if (KG > 1) dst[u][KG > 1 ? 1 : 0] = lds_f(a + 16u);  // (draws >= K: a neighbour's values, in columns that are never used)
        }
    };

    uint32_t pos = 0;    // byte offset of the current slice inside this wave's range
    uint32_t pos_r = 0;  // pos modulo the ring size
    for (int si = 0; si < ws.nsl; ++si) {
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)ws.ent, si);
        const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)ws.ent, si + 1);
        const uint32_t off = e0 & PSELL_OFF_MASK, off_next = e1 & PSELL_OFF_MASK;
        const int flags = (int)(e0 >> 30);
        const uint32_t units = off_next - off;
        const int nrows = (int)(units / 2u) - 1 - (HAS_KS ? 1 : 0);  // rows of 64 values: the set's transcripts / (MASKED) the longest fragment
        int w = MASKED ? run_w : nrows;
        const uint32_t bytes = units * 128u;
        if (!(dbg & 16)) __builtin_amdgcn_s_setprio(3);
        STAMP(1);  // slice bookkeeping
        {
            const int need = (int)((pos + bytes + 1023u) >> 10);
            int allowed = ws.issued - need + (need <= ws.primed ? extras : 0);
            wait_vm_outstanding(__builtin_amdgcn_readfirstlane(allowed));
        }
        STAMP(2);  // waiting for the DMA
        if (pend_w != 0 && !(flags & 2)) flush();
        if (pend_w == 0 && !((dbg & 64) && si > 0)) {
            // a new run: lane t < 16 reads transcript t's tile-local id from the slice's header and turns it into the
            // addresses of its x row and its gradient row; then the x values of the run, four transcripts at a time
            const uint32_t cid = *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>(
                (uintptr_t)(ring_lds + pos_r + (MASKED ? 4u * (uint32_t)(lane & 15) + 2u : 2u * (uint32_t)(lane & 15))));
            const bool live = MASKED ? cid != (uint32_t)PSELL_NO_COL : (lane & 15) < w;
            if (MASKED) {
                run_w = __builtin_popcount((uint32_t)__ballot(live) & 0xffffu);
                w = run_w;
            }
            xav = live ? xw_lds + cid * (uint32_t)(K * 4) : aux_lds;
            gav = live ? gw_lds + cid * (uint32_t)(K * 4) : aux_lds + 32u;
            load_x_group(0, xq);
            if (w > 4) load_x_group(1, xq + 4);
        }
        STAMP(3);  // run change: flush + column lookup

        const uint32_t row0 = wrap_u(pos_r + 256u);
        f32x4 d1[KG];
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) d1[kg] = f32x4{ROWSUM_FLOOR, ROWSUM_FLOOR, ROWSUM_FLOOR, ROWSUM_FLOOR};
        f32x4 kv = f32x4{1.f, 1.f, 1.f, 1.f};
        auto consumed = [&]() {
            // every operand of the slice is in registers: its ring bytes are free, the DMA for the pieces behind it goes
            // out before the (rest of the) matrix phases
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAMP(11);  // operand reads landed
            pos += bytes;
            pos_r += bytes;
            pos_r = pos_r >= RB ? pos_r - RB : pos_r;
            ring_refill<RB>(ws, ring_lds, min(ws.npieces, (int)(pos >> 10) + RP));
            STAMP(6);  // refill
            if (!(dbg & 16)) __builtin_amdgcn_s_setprio(0);
        };
        // NG = groups of four transcripts of the slice; FAST = its bytes do not wrap around the ring's end
        auto body = [&](auto NGc, auto FASTc) {
            constexpr int NG = decltype(NGc)::value;
            constexpr bool FAST = decltype(FASTc)::value;
            float pv[8], qv[2][4];  // the operands of two groups at a time
            uint32_t bp[4];
            if (FAST) {
#pragma unroll
                for (int u = 0; u < 4; ++u) bp[u] = lc1[u] + pos_r;
            }
            auto read_p = [&](int g0, int g1) {  // phase-1 operands of groups g0 .. g1 - 1: V[t][lane]
#pragma unroll
                for (int g = g0; g < g1; ++g) {
                    if (FAST) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) pv[4 * (g - g0) + u] = lds_f(bp[u] + 256u + 1024u * (uint32_t)g + 256u * (uint32_t)u);
                    } else {
                        const uint32_t rg = wrap_u(row0 + 1024u * (uint32_t)g);  // row 4 g
#pragma unroll
                        for (int u = 0; u < 4; ++u) pv[4 * (g - g0) + u] = lds_f(lc1[u] + wrap_u(rg + 256u * (uint32_t)u));
                    }
                }
            };
            auto read_q = [&](int g0, int g1) {  // phase-2 operands: V[4 g + j][4 b + u]
#pragma unroll
                for (int g = g0; g < g1; ++g) {
                    if (FAST) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) qv[g - g0][u] = lds_f(bp[u] + j256 + 256u + 1024u * (uint32_t)g);
                    } else {
                        const uint32_t rg = wrap_u(row0 + 1024u * (uint32_t)g);
                        const uint32_t jr = rg + 256u * (uint32_t)j;
                        const uint32_t back = jr >= RB ? RB : 0u;  // (the ring's end may fall inside this group of rows)
#pragma unroll
                        for (int u = 0; u < 4; ++u) qv[g - g0][u] = lds_f(lc1[u] + j256 + rg - back);
                    }
                }
            };
            auto phase1 = [&](const float *p4, const float (*x4)[KG]) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg) d1[kg] = __builtin_amdgcn_mfma_f32_4x4x1f32(p4[u], x4[u][kg], d1[kg], 0, 0, 0);
            };
            auto weights = [&]() {  // in place: d1[kg][i] of lane (b, j) belongs to fragment 4 b + i, draw 4 kg + j
#pragma unroll
                for (int kg = 0; kg < KG; ++kg)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float sv = d1[kg][i];
                        if (WANT_LP) {
                            const bool valid = 4 * kg + j < K && sv > 2.0f * ROWSUM_FLOOR;
                            if (HAS_KS) {
                                if (valid) lp2[kg] += ks_log2(kv[i], sv);
                            } else {
                                lpl[kg].mul(sv, valid);
                            }
                        }
                        d1[kg][i] = HAS_KS ? kv[i] * __builtin_amdgcn_rcpf(sv) : __builtin_amdgcn_rcpf(sv);  // (padded fragments: ks = 0)
                    }
            };
            auto phase2 = [&](int g, const float *q4) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg) acc[g][kg] = __builtin_amdgcn_mfma_f32_4x4x1f32(q4[i], d1[kg][i], acc[g][kg], 0, 0, 0);
            };
            if (HAS_KS)  // the multiplicities travel with the slice (its last row, stored in fragment order)
                kv = *reinterpret_cast<const f32x4 *>(ring + wrap_u(row0 + 256u * (uint32_t)nrows) + 16u * (uint32_t)b);
            if (MASKED) {
                // this lane's fragment, expanded by its mask: transcript t of the union is in row (number of mask bits
                // below t) of the slice when bit t is set; `a` runs through the rows' addresses.  Rows past the fragment's
                // last non-zero (and past the slice: the bytes behind it in the ring) are read and masked away.
                const uint32_t mk = *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>(
                    (uintptr_t)(lc1[0] + pos_r));  // low half of header word `lane`
                float mv[4 * NG];
                uint32_t a = lc1[0] + row0;
                const uint32_t lim = lc1[0] + RB;  // (slices that wrap around the ring's end)
#pragma unroll
                for (int t = 0; t < 4 * NG; ++t) {
                    mv[t] = lds_f(a);
                    a += ((mk >> t) & 1u) << 8;
                    if (!FAST) a = a >= lim ? a - RB : a;
                }
                consumed();
#pragma unroll
                for (int t = 0; t < 4 * NG; ++t) mv[t] = __uint_as_float(__float_as_uint(mv[t]) & (uint32_t)(((int)(mk << (31 - t))) >> 31));
                phase1(mv, xq);
                if (NG > 1) phase1(mv + 4, xq + 4);
                if (NG > 2) {  // (registers hold the x rows of eight transcripts: the others' are re-read per slice)
                    float xr[8][KG];
                    load_x_group(2, xr);
                    if (NG > 3) load_x_group(3, xr + 4);
                    phase1(mv + 8, xr);
                    if (NG > 3) phase1(mv + (NG > 3 ? 12 : 0), xr + 4);
                }
                weights();
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    quad_transpose(mv[4 * g], mv[4 * g + 1], mv[4 * g + 2], mv[4 * g + 3]);
                    phase2(g, mv + 4 * g);
                }
            } else if (NG <= 2) {
                // all operands at once; the slice's ring bytes are free before the matrix phases
                read_p(0, NG);
                read_q(0, NG);
                consumed();
                phase1(pv, xq);
                if (NG > 1) phase1(pv + 4, xq + 4);
                weights();
                phase2(0, qv[0]);
                if (NG > 1) phase2(1, qv[1]);
            } else {
                // (registers hold the x rows of eight transcripts: those of the third and fourth group are re-read from
                // the window for every slice.)  Two LDS round trips: the phase-1 operands, then -- overlapping the
                // reciprocals -- the phase-2 operands, in the registers the first ones have left
                float pw[8], xr[8][KG], qw[2][4];
                read_p(0, 2);
#pragma unroll
                for (int u = 0; u < 8; ++u) pw[u] = pv[u];
                read_p(2, NG);
                if (!(dbg & 4)) {
                    load_x_group(2, xr);
                    if (NG > 3) load_x_group(3, xr + 4);
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int kg = 0; kg < KG; ++kg) xr[u][kg] = xq[u][kg];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (!(dbg & 16)) __builtin_amdgcn_s_setprio(0);
                phase1(pw, xq);
                phase1(pw + 4, xq + 4);
                phase1(pv, xr);
                if (NG > 3) phase1(pv + 4, xr + 4);
                read_q(0, 2);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    qw[0][u] = qv[0][u];
                    qw[1][u] = qv[1][u];
                }
                read_q(2, NG);
                weights();
                consumed();
                phase2(0, qw[0]);
                phase2(1, qw[1]);
                phase2(2, qv[0]);
                if (NG > 3) phase2(3, qv[1]);
            }
        };
        using std::integral_constant;
        const bool fast = pos_r + bytes <= RB;
        if (fast) {
            if (w <= 4) body(integral_constant<int, 1>(), integral_constant<bool, true>());
            else if (w <= 8) body(integral_constant<int, 2>(), integral_constant<bool, true>());
            else if (w <= 12) body(integral_constant<int, 3>(), integral_constant<bool, true>());
            else body(integral_constant<int, 4>(), integral_constant<bool, true>());
        } else {
            if (w <= 4) body(integral_constant<int, 1>(), integral_constant<bool, false>());
            else if (w <= 8) body(integral_constant<int, 2>(), integral_constant<bool, false>());
            else if (w <= 12) body(integral_constant<int, 3>(), integral_constant<bool, false>());
            else body(integral_constant<int, 4>(), integral_constant<bool, false>());
        }
        pend_w = w;
        STAMP(5);  // phase 2
    }
    if (pend_w != 0) flush();
    if (WANT_LP) {
        // into the caller's accumulator, whose lane l < 16 collects draw l: the sixteen blocks' sums of draw 4 kg + j
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            double v = HAS_KS ? lp2[kg] * 0.693147180559945309417 : lpl[kg].log_value();
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16 && (lane >> 2) == kg && lane < K) lpacc += v;
        }
    }
}

// ---- the slice loop of the MIXED NARROW stream (BN: unrelated fragments of <= 16 transcripts) ----------------------------
// Lane = fragment, any 64 fragments of the tile: val[w][64]; lcol[w][64] (tile-local ids, 16 bit).  Two sweeps over the
// slice in the wave's LDS ring: row sums by gathers from the tile's x window, then the gradient contributions summed per
// run of lanes with the same transcript (segmented DPP scan) and added to the tile's gradient window by the last lane of
// every run.  ~3x the instructions of a matrix-core slice per entry, but 64 fragments per slice whatever their sets: the
// place for fragments without company, inside the same launch.  Every LDS access names its address space: a generic
// (flat) access would count against vmcnt and break the ring's counted waits.
template <int K>
__device__ inline void scatter_runs_lds(int c, float (&q)[K], uint32_t gw_lds, int lane)
{
    typedef __attribute__((address_space(3))) float *lds_fp;
    auto add = [](uint32_t a, float v) {
        __hip_atomic_fetch_add(reinterpret_cast<lds_fp>((uintptr_t)a), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const int c0 = __builtin_amdgcn_readfirstlane(c);
    if (__all(c == c0)) {  // one column for the whole wavefront: plain wave sum
        wave_sum_to_lane63_n<K>(q);
        if (lane == 63) {
#pragma unroll
            for (int k = 0; k < K; ++k) add(gw_lds + (uint32_t)(c0 * K + k) * 4u, q[k]);
        }
        return;
    }
    const int cprev = __builtin_amdgcn_update_dpp(-1, c, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const unsigned long long heads = __ballot(c != cprev);  // lane 0 compares with -1: always a head
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const int dist = lane - (63 - __clzll(heads & upto));  // distance to the head of this lane's run
    const int rl = lane & 15;
    const float m1 = dist >= 1 ? 1.f : 0.f, m2 = dist >= 2 ? 1.f : 0.f, m4 = dist >= 4 ? 1.f : 0.f,
                m8 = dist >= 8 ? 1.f : 0.f;
    const float mb15 = dist > rl ? 1.f : 0.f;           // run started in an earlier row of 16
    const float mb31 = dist > (lane & 31) ? 1.f : 0.f;  // run started before lane 32
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x111, 0xf>(q[k]), m1, q[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x112, 0xf>(q[k]), m2, q[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x114, 0xf>(q[k]), m4, q[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x118, 0xf>(q[k]), m8, q[k]);
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x142, 0xa>(q[k]), mb15, q[k]);  // row_bcast:15 -> rows 1, 3
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmaf(dpp_mov0<0x143, 0xc>(q[k]), mb31, q[k]);  // row_bcast:31 -> rows 2, 3
    const bool tail = lane == 63 || ((heads >> (lane + 1)) & 1ull);
    if (tail) {
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (q[k] != 0.0f) add(gw_lds + (uint32_t)(c * K + k) * 4u, q[k]);
    }
}

template <int K, uint32_t RB, bool WANT_LP, bool HAS_KS>
__device__ inline void mixed_stream(WaveStream &ws, const char *ring, int extras, const float *xw, float *gw, double &lpacc, int dbg
#ifdef POLEE_STAMPS
                                    , unsigned long long (&st_acc)[NSTAMP], unsigned long long &st_last
#endif
                                    )
{
    typedef const __attribute__((address_space(3))) float *lds_cfp;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(3))) f32x2 *lds_cf2p;
    typedef const __attribute__((address_space(3))) uint16_t *lds_cu16p;
    constexpr int RP = (int)(RB / 1024u);
    const int lane = wave_lane();
    const uint32_t ring_lds = lds_addr(ring);
    const uint32_t xw_lds = lds_addr(xw), gw_lds = lds_addr(gw);
    const uint32_t la4 = ring_lds + 4u * (uint32_t)lane, la2 = ring_lds + 2u * (uint32_t)lane;
    auto wrap_u = [&](uint32_t a) -> uint32_t {  // uniform ring offset a < 2 RB
        a = (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
        return a >= RB ? a - RB : a;
    };
    auto x_row_fma = [&](float v, uint32_t c, float (&sacc)[K]) {  // sacc[k] += v x[c][k]
        const uint32_t a = xw_lds + c * (uint32_t)(K * 4);
        if constexpr (K % 2 == 0) {
#pragma unroll
            for (int k = 0; k < K / 2; ++k) {
                const f32x2 x = *reinterpret_cast<lds_cf2p>((uintptr_t)(a + 8u * (uint32_t)k));
                sacc[2 * k] = fmaf(v, x.x, sacc[2 * k]);
                sacc[2 * k + 1] = fmaf(v, x.y, sacc[2 * k + 1]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) sacc[k] = fmaf(v, *reinterpret_cast<lds_cfp>((uintptr_t)(a + 4u * (uint32_t)k)), sacc[k]);
        }
    };
    LogAcc lpl[WANT_LP ? K : 1];  // (WANT_LP) per draw (LogAcc above); with multiplicities lp2 = sum ks log2 s
    double lp2[WANT_LP ? K : 1];
#pragma unroll
    for (int k = 0; k < (WANT_LP ? K : 1); ++k) {
        lpl[k].init();
        lp2[k] = 0.0;
    }

    uint32_t pos = 0, pos_r = 0;
    for (int si = 0; si < ws.nsl; ++si) {
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)ws.ent, si);
        const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)ws.ent, si + 1);
        const uint32_t units = (e1 & PSELL_OFF_MASK) - (e0 & PSELL_OFF_MASK);
        const int w = (int)((units - (HAS_KS ? 2u : 0u)) / 3u);  // float val[w][64]; uint16 lcol[w][64]; padding to 256 B; (float ks[64])
        const uint32_t bytes = units * 128u;
        STAMP(1);
        {
            const int need = (int)((pos + bytes + 1023u) >> 10);
            if (ws.issued < need) ring_refill<RB>(ws, ring_lds, need);
            int allowed = ws.issued - need + (need <= ws.primed ? extras : 0);
            wait_vm_outstanding(__builtin_amdgcn_readfirstlane(allowed));
        }
        STAMP(2);
        const uint32_t cols0 = pos_r + 256u * (uint32_t)w;  // (not wrapped yet)
        float sacc[K];
#pragma unroll
        for (int k = 0; k < K; ++k) sacc[k] = 0.0f;
        for (int t = 0; t < w; ++t) {
            const float v = *reinterpret_cast<lds_cfp>((uintptr_t)(la4 + wrap_u(pos_r + 256u * (uint32_t)t)));
            const uint32_t c = *reinterpret_cast<lds_cu16p>((uintptr_t)(la2 + wrap_u(cols0 + 128u * (uint32_t)t)));
            x_row_fma(v, c, sacc);
        }
        float ksv = 1.0f;
        if (HAS_KS) ksv = *reinterpret_cast<lds_cfp>((uintptr_t)(la4 + wrap_u(pos_r + bytes - 256u)));
        float wk[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            wk[k] = fast_weight(ksv, sacc[k]);  // (lanes without a fragment: s = 0, weight 0)
            if (WANT_LP) {
                if (HAS_KS) {
                    if (sacc[k] > 0.0f) lp2[WANT_LP ? k : 0] += ks_log2(ksv, sacc[k]);
                } else {
                    lpl[WANT_LP ? k : 0].mul(sacc[k], sacc[k] > 0.0f);
                }
            }
        }
        STAMP(4);
        for (int t = 0; t < w; ++t) {
            const float v = *reinterpret_cast<lds_cfp>((uintptr_t)(la4 + wrap_u(pos_r + 256u * (uint32_t)t)));
            const int c = (int)*reinterpret_cast<lds_cu16p>((uintptr_t)(la2 + wrap_u(cols0 + 128u * (uint32_t)t)));
            float q[K];
#pragma unroll
            for (int k = 0; k < K; ++k) q[k] = v * wk[k];
            if (!(dbg & 2)) scatter_runs_lds<K>(c, q, gw_lds, lane);
        }
        STAMP(5);
        // the slice is consumed: refill the ring behind it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pos += bytes;
        pos_r += bytes;
        pos_r = pos_r >= RB ? pos_r - RB : pos_r;
        ring_refill<RB>(ws, ring_lds, min(ws.npieces, (int)(pos >> 10) + RP));
        STAMP(6);
    }
    if (WANT_LP) {
        // into the caller's accumulator, whose lane l < 16 collects draw l
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double v = HAS_KS ? lp2[WANT_LP ? k : 0] * 0.693147180559945309417 : lpl[WANT_LP ? k : 0].log_value();
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
            if (lane == k) lpacc += v;
        }
    }
}

// ---- the slice loop of the WIDE MASKED stream (A2M: unions of 17..32 transcripts, fragments whose sets differ) ------------
// The narrow masked formulation (lane = fragment, batched 4 x 4 outer products, phase-2 operands by quad transposes) with
// the union's transcripts in two halves: phase 1 runs over all of them, the weights follow, then phase 2 and the flush
// of the first sixteen transcripts' gradients and of the rest -- one set of accumulators serves both halves, so nothing
// stays in registers across slices (leftover fragments rarely come in runs of slices with the same union; when they do,
// only the header work is saved).  Header: two rows of uint32 hw[64]: the low halves are bits 0..15 / 16..31 of the
// fragment's mask, the high halves of the first sixteen words of a row the tile-local ids of transcripts 0..15 / 16..31
// (PSELL_NO_COL past the union) -- every word a finite float, see loglik_internal.hpp.  Two waves of the workgroup work on
// such a tile, with the wide stream's 14 KiB rings (a slice is up to 2 + 32 + 1 rows).
template <int K, uint32_t RB, bool WANT_LP, bool HAS_KS>
__device__ inline void wide_masked_stream(WaveStream &ws, const char *ring, int extras, const float *xw, float *gw,
                                          uint32_t aux_lds, double &lpacc, int dbg
#ifdef POLEE_STAMPS
                                          , unsigned long long (&st_acc)[NSTAMP], unsigned long long &st_last
#endif
                                          )
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) float *lds_cfp;
    typedef __attribute__((address_space(3))) float *lds_fp;
    typedef const __attribute__((address_space(3))) uint16_t *lds_cu16p;
    constexpr int KG = (K + 3) / 4;
    constexpr int RP = (int)(RB / 1024u);
    const int lane = wave_lane();
    const uint32_t ring_lds = lds_addr(ring);
    const uint32_t xw_lds = lds_addr(xw), gw_lds = lds_addr(gw);
    const int j = lane & 3, b = lane >> 2;
    const uint32_t j4 = 4u * (uint32_t)j;
    const uint32_t la = ring_lds + 4u * (uint32_t)lane;  // this lane's column of a row of the ring
    auto lds_f = [](uint32_t a) -> float { return *reinterpret_cast<lds_cfp>((uintptr_t)a); };
    auto wrap_u = [&](uint32_t a) -> uint32_t {  // uniform ring offset a < 2 RB
        a = (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
        return a >= RB ? a - RB : a;
    };
    f32x4 acc[4][KG];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) acc[g][kg] = f32x4{0.f, 0.f, 0.f, 0.f};
    // lane t < 16: LDS addresses of the x row / gradient row of the union's transcripts t and 16 + t
    uint32_t xav0 = aux_lds, gav0 = aux_lds + 32u, xav1 = aux_lds, gav1 = aux_lds + 32u;
    int run_w = 0;
    LogAcc lpl[KG];  // (WANT_LP) see narrow_stream
    double lp2[KG];
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        lpl[kg].init();
        lp2[kg] = 0.0;
    }

    auto flush_group = [&](int g, uint32_t gav) {  // (as in narrow_stream)
        const uint32_t ga0 = (uint32_t)__builtin_amdgcn_readlane((int)gav, 4 * g + 0), ga2 = (uint32_t)__builtin_amdgcn_readlane((int)gav, 4 * g + 2);
        const uint32_t ga1 = (uint32_t)__builtin_amdgcn_readlane((int)gav, 4 * g + 1), ga3 = (uint32_t)__builtin_amdgcn_readlane((int)gav, 4 * g + 3);
        const uint32_t ga = ((lane & 32) ? ((lane & 16) ? ga3 : ga1) : ((lane & 16) ? ga2 : ga0)) + j4;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[g][kg][0]), __float_as_uint(acc[g][kg][1]), false, false);
            const float ab = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);
            auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[g][kg][2]), __float_as_uint(acc[g][kg][3]), false, false);
            const float cd = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
            auto r3 = __builtin_amdgcn_permlane16_swap(__float_as_uint(ab), __float_as_uint(cd), false, false);
            float q = __uint_as_float(r3[0]) + __uint_as_float(r3[1]);
            q += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), 0x114, 0xf, 0xf, true));  // row_shr:4
            q += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(q), 0x118, 0xf, 0xf, true));  // row_shr:8
            if ((lane & 12) == 12 && 4 * kg + j < K && !(dbg & 2))
                __hip_atomic_fetch_add(reinterpret_cast<lds_fp>((uintptr_t)(ga + 16u * (uint32_t)kg)), q, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
            acc[g][kg] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto load_x_group = [&](int g, uint32_t xav, float (*dst)[KG]) {  // x[c_t][4 kg + j] of transcripts 4 g .. 4 g + 3 of a half
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)xav, 4 * g + u) + j4;
            dst[u][0] = lds_f(a);
            if (KG > 1) dst[u][KG > 1 ? 1 : 0] = lds_f(a + 16u);
        }
    };

    uint32_t pos = 0, pos_r = 0;
    for (int si = 0; si < ws.nsl; ++si) {
        const uint32_t e0 = (uint32_t)__builtin_amdgcn_readlane((int)ws.ent, si);
        const uint32_t e1 = (uint32_t)__builtin_amdgcn_readlane((int)ws.ent, si + 1);
        const uint32_t off = e0 & PSELL_OFF_MASK, off_next = e1 & PSELL_OFF_MASK;
        const int flags = (int)(e0 >> 30);
        const uint32_t units = off_next - off;
        const int nrows = (int)(units / 2u) - 2 - (HAS_KS ? 1 : 0);  // rows of 64 values = the longest fragment of the slice
        const uint32_t bytes = units * 128u;
        if (!(dbg & 16)) __builtin_amdgcn_s_setprio(3);
        STAMP(1);
        {
            const int need = (int)((pos + bytes + 1023u) >> 10);
            if (ws.issued < need) ring_refill<RB>(ws, ring_lds, need);
            int allowed = ws.issued - need + (need <= ws.primed ? extras : 0);
            wait_vm_outstanding(__builtin_amdgcn_readfirstlane(allowed));
        }
        STAMP(2);
        const uint32_t hdr1 = wrap_u(pos_r + 256u);  // second header row
        if (run_w == 0 || !(flags & 2)) {
            // a new union: lane t < 16 turns the tile-local ids of transcripts t and 16 + t into the addresses of their x
            // rows and gradient rows
            const uint32_t c0 = *reinterpret_cast<lds_cu16p>((uintptr_t)(ring_lds + pos_r + 4u * (uint32_t)(lane & 15) + 2u));
            const uint32_t c1 = *reinterpret_cast<lds_cu16p>((uintptr_t)(ring_lds + hdr1 + 4u * (uint32_t)(lane & 15) + 2u));
            const bool live0 = c0 != (uint32_t)PSELL_NO_COL, live1 = c1 != (uint32_t)PSELL_NO_COL;
            run_w = __builtin_popcount((uint32_t)__ballot(live0) & 0xffffu) + __builtin_popcount((uint32_t)__ballot(live1) & 0xffffu);
            xav0 = live0 ? xw_lds + c0 * (uint32_t)(K * 4) : aux_lds;
            gav0 = live0 ? gw_lds + c0 * (uint32_t)(K * 4) : aux_lds + 32u;
            xav1 = live1 ? xw_lds + c1 * (uint32_t)(K * 4) : aux_lds;
            gav1 = live1 ? gw_lds + c1 * (uint32_t)(K * 4) : aux_lds + 32u;
        }
        STAMP(3);
        const uint32_t mk = (uint32_t)*reinterpret_cast<lds_cu16p>((uintptr_t)(la + pos_r)) |
                            ((uint32_t)*reinterpret_cast<lds_cu16p>((uintptr_t)(la + hdr1)) << 16);
        const uint32_t row0 = wrap_u(pos_r + 512u);
        f32x4 d1[KG];
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) d1[kg] = f32x4{ROWSUM_FLOOR, ROWSUM_FLOOR, ROWSUM_FLOOR, ROWSUM_FLOOR};
        f32x4 kv = f32x4{1.f, 1.f, 1.f, 1.f};
        auto consumed = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            STAMP(11);
            pos += bytes;
            pos_r += bytes;
            pos_r = pos_r >= RB ? pos_r - RB : pos_r;
            ring_refill<RB>(ws, ring_lds, min(ws.npieces, (int)(pos >> 10) + RP));
            STAMP(6);
            if (!(dbg & 16)) __builtin_amdgcn_s_setprio(0);
        };
        auto body = [&](auto NGHc, auto FASTc) {
            constexpr int NGH = decltype(NGHc)::value;  // groups of four transcripts in the second half
            constexpr bool FAST = decltype(FASTc)::value;
            float mvA[16], mvB[4 * NGH];
            uint32_t a = la + row0;
            const uint32_t lim = la + RB;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                mvA[t] = lds_f(a);
                a += ((mk >> t) & 1u) << 8;
                if (!FAST) a = a >= lim ? a - RB : a;
            }
#pragma unroll
            for (int t = 0; t < 4 * NGH; ++t) {
                mvB[t] = lds_f(a);
                a += ((mk >> (16 + t)) & 1u) << 8;
                if (!FAST) a = a >= lim ? a - RB : a;
            }
            if (HAS_KS) kv = *reinterpret_cast<const f32x4 *>(ring + wrap_u(row0 + 256u * (uint32_t)nrows) + 16u * (uint32_t)b);
            consumed();
#pragma unroll
            for (int t = 0; t < 16; ++t) mvA[t] = __uint_as_float(__float_as_uint(mvA[t]) & (uint32_t)(((int)(mk << (31 - t))) >> 31));
#pragma unroll
            for (int t = 0; t < 4 * NGH; ++t) mvB[t] = __uint_as_float(__float_as_uint(mvB[t]) & (uint32_t)(((int)(mk << (15 - t))) >> 31));
            auto phase1 = [&](const float *p4, const float (*x4)[KG]) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg) d1[kg] = __builtin_amdgcn_mfma_f32_4x4x1f32(p4[u], x4[u][kg], d1[kg], 0, 0, 0);
            };
            auto phase2 = [&](int g, const float *q4) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kg = 0; kg < KG; ++kg) acc[g][kg] = __builtin_amdgcn_mfma_f32_4x4x1f32(q4[i], d1[kg][i], acc[g][kg], 0, 0, 0);
            };
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                float xr[8][KG];
                load_x_group(g, xav0, xr);
                load_x_group(g + 1, xav0, xr + 4);
                phase1(mvA + 4 * g, xr);
                phase1(mvA + 4 * g + 4, xr + 4);
            }
#pragma unroll
            for (int g = 0; g < NGH; g += 2) {
                float xr[8][KG];
                load_x_group(g, xav1, xr);
                if (g + 1 < NGH) load_x_group(g + 1, xav1, xr + 4);
                phase1(mvB + 4 * g, xr);
                if (g + 1 < NGH) phase1(mvB + (g + 1 < NGH ? 4 * g + 4 : 0), xr + 4);
            }
            // weights, in place: d1[kg][i] of lane (b, j) belongs to fragment 4 b + i, draw 4 kg + j
#pragma unroll
            for (int kg = 0; kg < KG; ++kg)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sv = d1[kg][i];
                    if (WANT_LP) {
                        const bool valid = 4 * kg + j < K && sv > 2.0f * ROWSUM_FLOOR;
                        if (HAS_KS) {
                            if (valid) lp2[kg] += ks_log2(kv[i], sv);
                        } else {
                            lpl[kg].mul(sv, valid);
                        }
                    }
                    d1[kg][i] = HAS_KS ? kv[i] * __builtin_amdgcn_rcpf(sv) : __builtin_amdgcn_rcpf(sv);
                }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                quad_transpose(mvA[4 * g], mvA[4 * g + 1], mvA[4 * g + 2], mvA[4 * g + 3]);
                phase2(g, mvA + 4 * g);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) flush_group(g, gav0);
#pragma unroll
            for (int g = 0; g < NGH; ++g) {
                quad_transpose(mvB[4 * g], mvB[4 * g + 1], mvB[4 * g + 2], mvB[4 * g + 3]);
                phase2(g, mvB + 4 * g);
            }
#pragma unroll
            for (int g = 0; g < NGH; ++g) flush_group(g, gav1);
        };
        using std::integral_constant;
        const bool fast = pos_r + bytes <= RB;
        const int w = run_w;
        if (fast) {
            if (w <= 20) body(integral_constant<int, 1>(), integral_constant<bool, true>());
            else if (w <= 24) body(integral_constant<int, 2>(), integral_constant<bool, true>());
            else if (w <= 28) body(integral_constant<int, 3>(), integral_constant<bool, true>());
            else body(integral_constant<int, 4>(), integral_constant<bool, true>());
        } else {
            if (w <= 20) body(integral_constant<int, 1>(), integral_constant<bool, false>());
            else if (w <= 24) body(integral_constant<int, 2>(), integral_constant<bool, false>());
            else if (w <= 28) body(integral_constant<int, 3>(), integral_constant<bool, false>());
            else body(integral_constant<int, 4>(), integral_constant<bool, false>());
        }
        STAMP(5);
    }
    if (WANT_LP) {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            double v = HAS_KS ? lp2[kg] * 0.693147180559945309417 : lpl[kg].log_value();
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lane < 16 && (lane >> 2) == kg && lane < K) lpacc += v;
        }
    }
}

#ifdef POLEE_TILE_CYCLES
extern "C" int polee_debug_read_tile_cycles(unsigned long long *tiles, int ntiles, unsigned long long *wgs, int nwgs)
{
    if (hipMemcpyFromSymbol(tiles, HIP_SYMBOL(g_tile_cycles), sizeof(unsigned long long) * (size_t)ntiles) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(wgs, HIP_SYMBOL(g_wg_cycles), sizeof(unsigned long long) * (size_t)nwgs) != hipSuccess) return 1;
    static unsigned long long z[1 << 17];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_cycles), z, sizeof(unsigned long long) * (1 << 17)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wg_cycles), z, sizeof(unsigned long long) * 4096) == hipSuccess ? 0 : 1;
}
#endif
#ifdef POLEE_STAMPS
extern "C" int polee_debug_read_stamps(unsigned long long *out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 24) != hipSuccess) return 1;
    unsigned long long z[24] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z) == hipSuccess ? 0 : 1;
}
#endif

// x window of every tile, contiguous: xwin[e * K + k] = x[dict[e]][k] for every dictionary entry e (tiles' dictionaries
// start at multiples of 4 entries, so a tile's window starts 16-byte aligned).  One pass over 4 B x K per entry.
template <int K>
__global__ __launch_bounds__(256) void xwin_gather_kernel(const uint32_t *__restrict__ dict, const float *__restrict__ x,
                                                         int64_t entries, float *__restrict__ xwin,
                                                         const double *__restrict__ side_part, int side_nparts,
                                                         double *__restrict__ side_out)
{
    // a thread per dictionary entry: K adjacent floats in, K adjacent floats out (8-byte accesses when K is even)
    const int64_t e = ((int64_t)blockIdx.x - (side_part ? 1 : 0)) * blockDim.x + threadIdx.x;
    if (side_part && blockIdx.x == 0) {
        // the caller's side job (polee_loglik::side_part), one workgroup in front of the gather's (it starts first): column sums of side_part in a fixed
        // order (thread t adds rows t, t + 256, ...; a wave's 64 partial sums by a fixed shuffle tree, the four waves in order) and
        // their reciprocals
        __shared__ double sm[4 * K];
        double c[K];
#pragma unroll
        for (int d = 0; d < K; ++d) c[d] = 0.0;
        for (int r = threadIdx.x; r < side_nparts; r += 256)
#pragma unroll
            for (int d = 0; d < K; ++d) c[d] += side_part[(size_t)r * K + d];
#pragma unroll
        for (int d = 0; d < K; ++d) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) c[d] += __shfl_down(c[d], o, 64);
            if ((threadIdx.x & 63) == 0) sm[(threadIdx.x >> 6) * K + d] = c[d];
        }
        __syncthreads();
        if ((int)threadIdx.x < K) {
            const double t = ((sm[threadIdx.x] + sm[K + threadIdx.x]) + sm[2 * K + threadIdx.x]) + sm[3 * K + threadIdx.x];
            side_out[threadIdx.x] = t;
            side_out[K + threadIdx.x] = 1.0 / t;
        }
        return;
    }
    if (e >= entries) return;
    const float *src = x + (size_t)dict[e] * K;
    float *dst = xwin + (size_t)e * K;
    if constexpr (K % 2 == 0) {
#pragma unroll
        for (int k = 0; k < K / 2; ++k) reinterpret_cast<float2 *>(dst)[k] = reinterpret_cast<const float2 *>(src)[k];
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k] = src[k];
    }
}

// deterministic mode, second kernel: g[j][k] += the windows' values of transcript j (tslot lists its dictionary entries,
// ascending = tile order).  A thread per (j, k) walks a short list; a transcript present in more than GWIN_HEAVY tiles
// gets a block per draw (gwin_reduce_heavy_kernel): thread t sums entries t, t + 256, ... in order and a wave's 64 partial sums are
// combined by a fixed shuffle tree -- a fixed order either way.  Block 0 also adds the workgroups' log-likelihood sums in
// workgroup order.
constexpr uint32_t GWIN_HEAVY = 32;
__global__ void gwin_reduce_kernel(const uint32_t *__restrict__ tslot_ptr, const uint32_t *__restrict__ tslot,
                                   const float *__restrict__ gwin, int K, int64_t n, float *__restrict__ g,
                                   const double *__restrict__ lpwin, int nwg, double *__restrict__ lp,
                                   const uint32_t *__restrict__ gmap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lp && blockIdx.x == 0 && threadIdx.x < K) {
        double s = 0.0;
        for (int b = 0; b < nwg; ++b) s += lpwin[(size_t)b * K + threadIdx.x];
        lp[threadIdx.x] += s;
    }
    if (i >= n * K) return;
    const int64_t j = i / K;
    const int k = (int)(i - j * K);
    const uint32_t b = tslot_ptr[j], e1 = tslot_ptr[j + 1];
    if (e1 - b > GWIN_HEAVY) return;
    // (four entries' loads in flight at a time; the additions stay in list order)
    float s = 0.0f;
    uint32_t e = b;
    for (; e + 4 <= e1; e += 4) {
        const uint32_t t0 = tslot[e], t1 = tslot[e + 1], t2 = tslot[e + 2], t3 = tslot[e + 3];
        const float a0 = gwin[(size_t)t0 * K + k], a1 = gwin[(size_t)t1 * K + k], a2 = gwin[(size_t)t2 * K + k], a3 = gwin[(size_t)t3 * K + k];
        s = (((s + a0) + a1) + a2) + a3;
    }
    if (e + 2 <= e1) {
        const uint32_t t0 = tslot[e], t1 = tslot[e + 1];
        const float a0 = gwin[(size_t)t0 * K + k], a1 = gwin[(size_t)t1 * K + k];
        s = (s + a0) + a1;
        e += 2;
    }
    if (e < e1) s += gwin[(size_t)tslot[e] * K + k];
    g[gmap ? (size_t)gmap[j] * K + k : (size_t)i] += s;
}
// (round 5: a block of 256 threads per (heavy transcript, draw) -- it was one wave per transcript looping over the draws, 25 us per
// pass for a handful of transcripts; the order is still fixed: thread t sums entries t, t + 256, ..., a wave's 64 partial sums meet
// in a fixed shuffle tree, the four waves' sums are added in wave order)
__global__ __launch_bounds__(256) void gwin_reduce_heavy_kernel(const uint32_t *__restrict__ heavy,
                                                              const uint32_t *__restrict__ tslot_ptr,
                                                              const uint32_t *__restrict__ tslot,
                                                              const float *__restrict__ gwin, int K, float *__restrict__ g,
                                                              const uint32_t *__restrict__ gmap)
{
    __shared__ float part[4];
    const uint32_t j = heavy[blockIdx.x];
    const int k = (int)blockIdx.y;
    const uint32_t b = tslot_ptr[j], e1 = tslot_ptr[j + 1];
    float s = 0.0f;
    for (uint32_t e = b + threadIdx.x; e < e1; e += 256) s += gwin[(size_t)tslot[e] * K + k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) g[(size_t)(gmap ? gmap[j] : j) * K + k] += ((part[0] + part[1]) + part[2]) + part[3];
}

// Stream S (loglik_internal.hpp): fragments with ONE compatible transcript were collapsed at build time into cnt[j] (the
// sum of their multiplicities) and a constant; a pass adds cnt_j / x_j[k] to the gradient (the reference's
// X_ij / (X_ij x_j), sparse.jl:36 after likelihood.jl:41) and cnt_j log x_j[k] (+ the constant sum of log X_ij) to lp.
// A thread per transcript; lp: per-block partial sums, added in block order by single_lp_finish_kernel (a fixed order:
// the deterministic mode stays bitwise reproducible).
constexpr int SINGLE_THREADS = 256;
__global__ __launch_bounds__(SINGLE_THREADS) void single_rows_kernel(const float *__restrict__ cnt, const float *__restrict__ x,
                                                                    int K, int64_t n, float *__restrict__ g,
                                                                    double *__restrict__ part)
{
    __shared__ double sm[SINGLE_THREADS / 64][PSELL_MAX_K];
    const int64_t j = (int64_t)blockIdx.x * SINGLE_THREADS + threadIdx.x;
    const float c = j < n ? cnt[j] : 0.0f;
    double ls[PSELL_MAX_K];
#pragma unroll
    for (int k = 0; k < PSELL_MAX_K; ++k) ls[k] = 0.0;
    if (c != 0.0f) {
        const float *xr = x + (size_t)j * K;
        float *gr = g + (size_t)j * K;
#pragma unroll
        for (int k = 0; k < PSELL_MAX_K; ++k)
            if (k < K) {
                const float xv = xr[k];
                if (g) gr[k] += c / xv;
                if (part) ls[k] = (double)c * log((double)xv);
            }
    }
    if (!part) return;
#pragma unroll
    for (int k = 0; k < PSELL_MAX_K; ++k) {
        double v = ls[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) part[(size_t)blockIdx.x * K + threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
}
__global__ void single_lp_finish_kernel(const double *__restrict__ part, int nblocks, int K, double logsum, double *__restrict__ lp)
{
    const int k = threadIdx.x;
    if (k >= K) return;
    double s = logsum;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * K + k];
    lp[k] += s;
}

// LDS layout of the streaming kernel:
//   [rings: 4 x 7 KiB (all kinds but A2M) or 2 x 14 KiB (A2M)][xw 0][xw 1][gw (x 4 in deterministic mode)][ids 0][ids 1][ent 4 x 64][desc 2 x 64][aux 16]
constexpr uint32_t STREAM_RB1 = 7168u, STREAM_RB2 = 14336u, STREAM_RINGS = 28672u;
template <int K>
constexpr uint32_t stream_ring_total()
{
    return STREAM_RINGS;
}
template <int K>
constexpr uint32_t stream_xw_bytes()  // whole 1 KiB pieces
{
    return ((uint32_t)PSELL_TILE_COLS_TARGET * K * 4u + 1023u) & ~1023u;
}
template <int K, bool DET>
constexpr uint32_t stream_lds_bytes()  // (deterministic mode: one gradient window per wave)
{
    return stream_ring_total<K>() + (DET ? 6u : 3u) * stream_xw_bytes<K>() + 2u * PSELL_TILE_COLS_TARGET * 4u + 4u * 256u + 2u * 256u + 64u;
}

// DET: the deterministic mode -- bitwise reproducible gradients: each wave accumulates into its own LDS window (a wave's
// LDS adds retire in program order), the tile's flush sums the four windows in wave order and STORES the result to the
// tile's slot of gwin, and gwin_reduce_kernel adds a transcript's slots in tile order.
template <int K, bool WANT_LP, bool HAS_KS, bool DET>
// (The instances that also return lp carry the log accumulators across the slice loops: at four waves per SIMD -- 128
// registers -- they spill, and a spill's reload drains the LDS-DMA ring (measured: 1.40 ms per pass instead of 0.25).  They run
// three workgroups per CU, 170 registers, no scratch.  Round 5: so do the instances with multiplicities (the factored likelihood of
// salmon's equivalence classes) -- the wide stream's two-stage body holds 32 operand registers across its second stage, and with the
// ks row on top the 128-register instance spilled 11 registers inside the slice loops.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((WANT_LP || HAS_KS) ? 3 : 4, (WANT_LP || HAS_KS) ? 3 : 4)))
void loglik_stream_kernel(PsellArgs A, int dbg)
{
    extern __shared__ float lds[];
    constexpr uint32_t XWB = stream_xw_bytes<K>();
    constexpr uint32_t GWN = DET ? 4u : 1u;  // gradient windows
    char *const base = reinterpret_cast<char *>(lds);
    char *const rings = base;
    auto xw_of = [&](int b) -> float * { return reinterpret_cast<float *>(base + stream_ring_total<K>() + (uint32_t)b * XWB); };
    float *const gw = reinterpret_cast<float *>(base + stream_ring_total<K>() + 2 * XWB);
    auto ids_of = [&](int b) -> uint32_t * {
        return reinterpret_cast<uint32_t *>(base + stream_ring_total<K>() + (2 + GWN) * XWB + (uint32_t)b * (PSELL_TILE_COLS_TARGET * 4));
    };
    uint32_t *const entb = reinterpret_cast<uint32_t *>(base + stream_ring_total<K>() + (2 + GWN) * XWB + 2 * PSELL_TILE_COLS_TARGET * 4);
    uint32_t *const descb = entb + 4 * 64;  // 2 x 64 words: the schedule entry two rounds ahead, by LDS-DMA
    float *const auxz = reinterpret_cast<float *>(descb + 2 * 64);  // 8 zeros, 8 words of scratch (narrow_stream)

    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t G = gridDim.x;
    // Static schedule: every workgroup walks its own column.  Dynamic schedule: one list, longest tile first; a workgroup
    // owns its next two tiles (the next one's x window, ids and offsets are on their way while the current one streams)
    // and wave 0 draws the position after those from a global counter -- one atomic per tile, issued when the tile's slices are
    // done and read at the next tile's start, behind the barriers and the flush.  Workgroups then finish within one (small, late) tile of each other whatever the
    // cost model says: with the static lists the slowest workgroup was 20 - 30 % above the mean (tile times depend on
    // what the neighbours on the CU are doing, not only on the tile).
    const bool dyn = A.dyn_ctr != nullptr;
    const PosDesc *__restrict__ sched = dyn ? A.sched_dyn : A.sched;
    uint32_t p2 = blockIdx.x + 2u * G;  // (wave 0) position of the tile after the next one
    const uint8_t *__restrict__ xwin_b = reinterpret_cast<const uint8_t *>(A.xwin);
#ifdef POLEE_STAMPS
    unsigned long long st_acc[NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = __builtin_amdgcn_s_memtime();
    unsigned long long n_slices = 0, n_tiles = 0;
#endif
#ifdef POLEE_TILE_CYCLES
    const unsigned long long t_wg0 = __builtin_amdgcn_s_memtime();
    unsigned long long t_tile0 = t_wg0;
#endif

    // 0 = A1 (dense narrow), 2 = A1M (masked narrow), 1 = A2 (dense wide, in two stages per slice: wide_stream), 4 = BN (mixed
    // narrow): four waves, 7 KiB rings; 3 = A2M (masked wide): two active waves, 14 KiB rings; the schedule holds no others
    auto kind_of = [&](uint32_t tile) -> int { return (int)tile < A.tiles_a1 ? 0 : ((int)tile < A.tiles_a1m ? 2 : ((int)tile < A.tiles_a2 ? 1 : ((int)tile < A.tiles_a ? 3 : 4))); };
    // this wave's share [sb, se) of an A tile's slices: a contiguous block, so that runs stay inside one wave
    auto share = [&](int kind, const PosDesc &t, uint32_t &sb, uint32_t &se) {
        const int nw = kind == 3 ? 2 : 4;
        uint32_t a1 = t.c1, a2 = t.c2, a3 = t.c3;
        asm volatile("" : "+s"(a1), "+s"(a2), "+s"(a3));  // (opaque: or the selects below become an indexed load of a PosDesc kept in scratch memory)
        const uint32_t lo = wave == 0 ? t.s0 : (wave == 1 ? a1 : (wave == 2 ? a2 : a3));
        const uint32_t hi = wave + 1 >= nw ? t.s1 : (wave == 0 ? a1 : (wave == 1 ? a2 : a3));
        sb = wave < nw ? lo : t.s1;
        se = wave < nw ? hi : t.s1;
    };
    // requests everything tile `t` needs besides its slice stream; returns the number of vector-memory operations
    auto prefetch = [&](const PosDesc &t, int buf) -> int {
        int cnt = 0;
        const int npx = (int)((t.L * (uint32_t)K * 4u + 1023u) >> 10);
        const uint8_t *src = xwin_b + (size_t)t.d0 * K * 4;
        const uint32_t dst = lds_addr(xw_of(buf));
        for (int p = wave; p < npx; p += 4) {
            dma_1k_keep(uniform_ptr(src + (size_t)p * 1024), (uint32_t)wave_lane() * 16u, dst + (uint32_t)p * 1024u);
            ++cnt;
        }
        if ((uint32_t)wave * 64u < t.L) {  // transcript ids of the dictionary (for the flush)
            dma_256(uniform_ptr(A.dict + t.d0 + (uint32_t)wave * 64u), min((uint32_t)wave_lane(), t.L - 1u - (uint32_t)wave * 64u) * 4u,
                    lds_addr(ids_of(buf)) + (uint32_t)wave * 256u);
            ++cnt;
        }
        const int kind = kind_of(t.tile);
        {  // slice offsets of this wave's share, one per lane
            uint32_t sb, se;
            share(kind, t, sb, se);
            dma_256(uniform_ptr(A.slice_off + sb), min((uint32_t)wave_lane(), se - sb) * 4u, lds_addr(entb) + (uint32_t)wave * 256u);
            ++cnt;
        }
        return cnt;
    };
    // after the prefetch has landed: this wave's stream state for tile `t`, ring started
    WaveStream ws;
    const int ahead = (dbg >> 8) & 15;  // (experiment: pieces requested ahead; 0 = the whole ring)
    auto start_ring = [&](const PosDesc &t) {
        const int kind = kind_of(t.tile);
        ws.nsl = 0; ws.npieces = 0; ws.issued = 0; ws.islot = 0; ws.primed = 0; ws.ent = 0u; ws.gsrc = A.data;
        uint32_t sb, se;
        share(kind, t, sb, se);
        ws.ent = entb[wave * 64 + wave_lane()];
        ws.nsl = (int)(se - sb);
        const uint32_t cb = (uint32_t)__builtin_amdgcn_readlane((int)ws.ent, 0) & PSELL_OFF_MASK;
        const uint32_t ce = (uint32_t)__builtin_amdgcn_readlane((int)ws.ent, ws.nsl) & PSELL_OFF_MASK;  // 128-byte units
        ws.npieces = (int)(((ce - cb) * 128u + 1023u) >> 10);
        ws.gsrc = reinterpret_cast<const uint8_t *>(uniform_ptr(A.data + (size_t)cb * 128));
        if (kind != 3) {
            ring_refill<STREAM_RB1>(ws, lds_addr(rings + wave * STREAM_RB1), min(ws.npieces, ahead ? min(ahead, (int)(STREAM_RB1 / 1024u)) : (int)(STREAM_RB1 / 1024u)));
        } else {
            ring_refill<STREAM_RB2>(ws, lds_addr(rings + (wave < 2 ? wave : 0) * STREAM_RB2), min(ws.npieces, ahead ? min(2 * ahead, (int)(STREAM_RB2 / 1024u)) : (int)(STREAM_RB2 / 1024u)));
        }
        ws.primed = ws.issued;
    };

    double lp_a = 0.0;  // lane (tt, q) holds the share of draw tt

    PosDesc cur = sched[blockIdx.x];
    if (cur.tile == POS_NONE) return;  // (the grid never exceeds the number of tiles)
    PosDesc nxt = sched[blockIdx.x + G];
    for (int i = threadIdx.x; i < (int)(GWN * XWB / 4u); i += 256) gw[i] = 0.0f;
    // the rings start out as zeros: operand rows past a slice's last transcript are read (and multiplied by 0), so
    // whatever lies behind a slice in the ring has to be finite
    for (int i = threadIdx.x; i < (int)(stream_ring_total<K>() / 16u); i += 256)
        reinterpret_cast<float4 *>(rings)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x < 16) auxz[threadIdx.x] = 0.0f;
    lds_barrier();
    (void)prefetch(cur, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    start_ring(cur);
    int young = 0;  // vector-memory operations issued after the ring was started (see wide_stream)
    lds_barrier();
    int buf = 0;
    for (uint32_t round = 0;; ++round) {
        const bool more = nxt.tile != POS_NONE;
        const int kind = kind_of(cur.tile);
        if (more) young += prefetch(nxt, buf ^ 1);
        if (wave == 0) {
            // the schedule entry after the next (a scalar load here would stall every tile by its latency; through LDS it
            // arrives in the background like everything else and is read after the tile's barriers)
            uint32_t off8;  // (lane & 7) * 4, computed here: hoisted out of the tile loop it was spilled, and its reload
                            // (a scratch load the compiler waits for with vmcnt(0)) drained this wave's ring every tile
            asm volatile("v_and_b32_e32 %0, 7, %1\n\tv_lshlrev_b32_e32 %0, 2, %0" : "=v"(off8) : "v"(wave_lane()));
            dma_256(uniform_ptr(sched + p2), off8, lds_addr(descb + (round & 1u) * 64u));
            ++young;
        }
        STAMP(0);  // between tiles: prefetch issue
        // (Round 5 tried WORK STEALING inside the workgroup here -- a wave that had finished its share of the tile's slices stole the
        // tail half of the longest remaining share with an LDS compare-and-swap, the owners taking every slice with an LDS atomic
        // add: correct, and 5 - 12 % SLOWER on every input, profiles/r05_work_stealing_ab.txt.  The static shares stay.)
        {
            const int extras = young;
            if (kind == 0) {
                narrow_stream<K, STREAM_RB1, WANT_LP, HAS_KS, false>(ws, rings + wave * STREAM_RB1, extras, xw_of(buf), gw + (DET ? (uint32_t)wave * (XWB / 4u) : 0u), lds_addr(auxz), lp_a, dbg
#ifdef POLEE_STAMPS
                                                                                 , st_acc, st_last
#endif
                );
            } else if (kind == 2) {
                narrow_stream<K, STREAM_RB1, WANT_LP, HAS_KS, true>(ws, rings + wave * STREAM_RB1, extras, xw_of(buf), gw + (DET ? (uint32_t)wave * (XWB / 4u) : 0u), lds_addr(auxz), lp_a, dbg
#ifdef POLEE_STAMPS
                                                                                 , st_acc, st_last
#endif
                );
            } else if (kind == 4) {
                mixed_stream<K, STREAM_RB1, WANT_LP, HAS_KS>(ws, rings + wave * STREAM_RB1, extras, xw_of(buf), gw + (DET ? (uint32_t)wave * (XWB / 4u) : 0u), lp_a, dbg
#ifdef POLEE_STAMPS
                                                             , st_acc, st_last
#endif
                );
            } else if (kind == 3) {
                wide_masked_stream<K, STREAM_RB2, WANT_LP, HAS_KS>(ws, rings + (wave < 2 ? wave : 0) * STREAM_RB2, extras, xw_of(buf), gw + (DET ? (uint32_t)wave * (XWB / 4u) : 0u), lds_addr(auxz), lp_a, dbg
#ifdef POLEE_STAMPS
                                                                   , st_acc, st_last
#endif
                );
            } else {
                wide_stream<K, STREAM_RB1, WANT_LP, HAS_KS>(ws, rings + wave * STREAM_RB1, extras, xw_of(buf), gw + (DET ? (uint32_t)wave * (XWB / 4u) : 0u), lp_a, dbg
#ifdef POLEE_STAMPS
                                                                               , st_acc, st_last
#endif
                );
            }
        }
#ifdef POLEE_STAMPS
        n_slices += (unsigned long long)ws.nsl;
        ++n_tiles;
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (also: the next tile's prefetch has landed)
        STAMP(7);  // draining the queue after the last slice
        young = 0;
        unsigned int drawn = 0u;  // (this tile's draw: defined and consumed inside one iteration, or it is carried -- and spilled -- around the loop)
        if (dyn && wave == 0) {
            // The draw of the position after the next two (one per tile: the host counts on that, dyn_base).  Issued HERE, with the
            // queue empty, and read at the end of this iteration, behind the barriers and the flush: the result occupies a register
            // only in between (until round 5 it was issued at the tile's start and lived across the slice loops).  A compiler-visible
            // atomic: the compiler waits for it with vmcnt(0) where it is read -- by then the next ring's first pieces, requested
            // before the barrier, have had the barrier, the flush and the second barrier to arrive.  (An inline-assembly atomic with a
            // counted wait was tried first: the compiler, unaware that the register is still in flight, copied it -- wrong tickets,
            // tiles skipped; caught by the C2-size oracle tests.)
            if (wave_lane() == 0) drawn = __hip_atomic_fetch_add(A.dyn_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // a wave that is done starts the next tile's ring BEFORE the barrier when that ring is its own LDS (same kind
        // of uniform tile); otherwise the ring area may still be in use by a slower wave
        const bool early = more && (kind_of(nxt.tile) == 3) == (kind == 3);  // (all kinds but the wide masked one share a ring geometry)
        if (early) start_ring(nxt);
        STAMP(13);  // starting the next ring (before the barrier)
        lds_barrier();  // every wave's contributions are in gw
        STAMP(8);  // waiting for the other waves of the workgroup
        if (more && !early) start_ring(nxt);
        STAMP(13);  // starting the next ring (after the barrier)
        {
            const int LK = (int)cur.L * K;
            const uint32_t *ids = ids_of(buf);
            constexpr int NBF = (PSELL_TILE_COLS_TARGET * K + 255) / 256;
#pragma unroll
            for (int b = 0; b < NBF; ++b) {
                const int i0 = b * 256 + wave * 64;
                if (i0 < LK) {  // (wave-uniform; lane 0 of the wave is active, so the wave issues exactly one atomic here)
                    int i = i0 + wave_lane();
                    asm volatile("" : "+v"(i));  // (keeps i / K from being hoisted out of the tile loop and spilled)
                    if (i < LK) {
                        const int l = i / K;
                        const int k = i - l * K;
                        float v = gw[i];
                        gw[i] = 0.0f;
                        if (DET) {  // the four waves' windows, in wave order; stored to the tile's slot
#pragma unroll
                            for (uint32_t wv = 1; wv < 4; ++wv) {
                                v += gw[wv * (XWB / 4u) + i];
                                gw[wv * (XWB / 4u) + i] = 0.0f;
                            }
                            float *dst = A.gwin + (size_t)cur.d0 * K + i;
                            if (!(dbg & 1)) asm volatile("global_store_dword %0, %1, off" ::"v"(dst), "v"(v) : "memory");
                        } else {
                            float *dst = A.g + (size_t)ids[l] * K + k;
                            if (!(dbg & 1)) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
                        }
                    }
                    if (!(dbg & 1)) ++young;  // (younger than the pieces of the ring that has just been started)
                }
            }
        }
        STAMP(15);  // flush issue
        lds_barrier();  // gw is zero again, the next x window is complete
        STAMP(9);  // barrier B
#ifdef POLEE_TILE_CYCLES
        if (threadIdx.x == 0 && cur.tile < (1u << 17)) {
            const unsigned long long now_t = __builtin_amdgcn_s_memtime();
            atomicAdd(&g_tile_cycles[cur.tile], now_t - t_tile0);
            t_tile0 = now_t;
        }
#endif
        if (!more) break;
        cur = nxt;
        if (wave == 0) {
            if (dyn) {
                p2 = min(3u * G + ((uint32_t)__builtin_amdgcn_readfirstlane((int)drawn) - A.dyn_base), A.dyn_last);
            } else {
                p2 += G;
            }
        }
        {
            const uint32_t *dp = descb + (round & 1u) * 64u;
            nxt.tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)dp[0]);
            nxt.s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dp[1]);
            nxt.s1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dp[2]);
            nxt.d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dp[3]);
            nxt.L = (uint32_t)__builtin_amdgcn_readfirstlane((int)dp[4]);
            nxt.c1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dp[5]);
            nxt.c2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dp[6]);
            nxt.c3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dp[7]);
        }
        buf ^= 1;
    }
    if (WANT_LP && DET) {
        // per-workgroup sums in wave order, stored; gwin_reduce_kernel adds them in workgroup order
        double v = lp_a;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        double *lpw = reinterpret_cast<double *>(gw);  // (the windows are idle now)
        lds_barrier();
        if (wave_lane() < K) lpw[wave * K + wave_lane()] = v;
        lds_barrier();
        if (threadIdx.x < K) A.lpwin[(size_t)blockIdx.x * K + threadIdx.x] = ((lpw[threadIdx.x] + lpw[K + threadIdx.x]) + lpw[2 * K + threadIdx.x]) + lpw[3 * K + threadIdx.x];
    } else if (WANT_LP) {
        double v = lp_a;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (wave_lane() < K) atomicAdd(A.lp + wave_lane(), v);
    }
#ifdef POLEE_TILE_CYCLES
    if (threadIdx.x == 0 && blockIdx.x < 4096) atomicAdd(&g_wg_cycles[blockIdx.x], __builtin_amdgcn_s_memtime() - t_wg0);
#endif
#ifdef POLEE_STAMPS
    STAMP(10);
    if (wave_lane() == 0) {
        for (int i = 0; i < NSTAMP; ++i) atomicAdd(&g_stamps[i], st_acc[i]);
        atomicAdd(&g_stamps[16], n_slices);
        atomicAdd(&g_stamps[17], n_tiles);
        atomicAdd(&g_stamps[18], 1ull);
    }
#endif
}


// Cost model of the schedules, MEASURED: time of every tile of a C2 sample from a clock read per tile (diagnostic build
// POLEE_TILE_CYCLES, tools/probe/tile_cycles.py), regressed per stream on the bytes the tile streams: cycles = fixed + per KiB x
// KiB, R^2 0.6 - 0.9, mean error per tile ~15 % (round 5, profiles/r05_tile_cycles_*.txt: the wide stream on four waves):
//     A1 8.7 k + 350   A1M 9 k + 750   A2 8.7 k + 310   A2M 5 k + 1400   BN 2 k + 3100
// (dense slices cost what their bytes cost; masked ones also the matrix-core work of their union; the wide masked stream runs
// on two of the four waves; BN sweeps lane per fragment).  The first model -- bytes + 4 KiB, x 1.5 for the wide streams -- left
// the slowest workgroup 25 - 30 % above the mean.
static const double TILE_COST_FIXED[PSELL_NSTREAMS] = {8700.0, 9000.0, 8700.0, 5000.0, 2000.0, 2000.0};
static const double TILE_COST_PER_KIB[PSELL_NSTREAMS] = {350.0, 750.0, 310.0, 1400.0, 3100.0, 3100.0};
static double slices_cost(const PsellHost &h, int stream, uint32_t sa, uint32_t sb)
{
    const double kib = 0.125 * (double)((h.slice_off[sb] & PSELL_OFF_MASK) - (h.slice_off[sa] & PSELL_OFF_MASK));
    static const bool old_cost = getenv("POLEE_OLD_COST") != nullptr;  // (A/B)
    if (old_cost) return (kib * 1024.0 + 4096.0) * (stream == PSELL_A2 || stream == PSELL_A2M ? 1.5 : 1.0);
    return TILE_COST_FIXED[stream] + TILE_COST_PER_KIB[stream] * kib;
}
// The waves of a workgroup take contiguous blocks of the slices [sa, sb) of tile t with about equal matrix-core work (phase 1:
// 4 ceil(w / 4) instructions, phase 2: 8 for w <= 8, else 16 per 16 transcripts; + a fixed part): cut[0..2] = the boundaries.
// (needs the slice metadata: only while the handle is being created)
static void wave_cuts(const PsellHost &h, int64_t t, uint32_t s0, uint32_t s1, uint32_t cut[3])
{
    cut[0] = cut[1] = cut[2] = s1;
    if (t >= h.num_tiles_s) return;
    static const bool bytes_cut = getenv("POLEE_BYTES_CUT") != nullptr;  // (A/B: the waves' shares balanced on bytes)
    const int nw = h.stream_of_tile(t) != PSELL_A2M ? 4 : 2;
    // (experiment, POLEE_CUT_MODEL="a,b,c": cost = a + b x groups of four transcripts + c x (1 + groups) when the slice
    // starts a new run -- the flush of the previous set and the column lookup of the new one)
    static const char *cut_env = getenv("POLEE_CUT_MODEL");
    static double cm[3] = {0, 0, 0};
    static const bool cut_custom = cut_env && sscanf(cut_env, "%lf,%lf,%lf", &cm[0], &cm[1], &cm[2]) == 3;
    auto cost = [&](uint32_t sl) {
        const int w = h.slice_w[sl];
        if (bytes_cut) return 512.0 + 128.0 * (double)((h.slice_off[sl + 1] & PSELL_OFF_MASK) - (h.slice_off[sl] & PSELL_OFF_MASK));
        if (h.stream_of_tile(t) == PSELL_BN) return 6.0 * w + 4.0;  // (instructions per entry, not matrix-core work)
        if (cut_custom) {
            const double ng = (double)((w + 3) / 4);
            const bool cont = (h.slice_flags[sl] & 2) != 0 && sl > s0;
            return cm[0] + cm[1] * ng + (cont ? 0.0 : cm[2] * (1.0 + ng));
        }
        return 4.0 * ((w + 3) / 4) + (w <= 8 ? 8.0 : 16.0 * ((w + 15) / 16)) + 10.0;
    };
    double total = 0.0;
    for (uint32_t sl = s0; sl < s1; ++sl) total += cost(sl);
    double acc = 0.0;
    int wv = 1;
    for (uint32_t sl = s0; sl < s1 && wv < nw; ++sl) {
        acc += cost(sl);
        // a wave owns at most 63 slices (one offset per lane + the end)
        while (wv < nw && (acc >= total * wv / nw || sl + 1 - (wv == 1 ? s0 : cut[wv - 2]) >= 63u)) cut[wv++ - 1] = sl + 1;
    }
    // (the last wave takes what is left: if that is more than 63 slices -- cheap slices at the tile's start -- cut
    // by count instead; the builder keeps a tile within 63 slices per active wave)
    if (s1 - (nw > 1 ? cut[nw - 2] : s0) > 63u) {
        const uint32_t per = (s1 - s0 + (uint32_t)nw - 1) / (uint32_t)nw;
        for (int q = 1; q < nw; ++q) cut[q - 1] = std::min(s1, s0 + per * (uint32_t)q);
    }
}

// Static schedule of the streaming kernel for a grid of G workgroups: the uniform tiles sorted by cost, dealt to the
// workgroups in snake order (equal sums), and inside every workgroup's list the wide tiles (stream A2, two active
// waves) spread evenly between the A1 tiles.
static polee_status ensure_schedule(polee_loglik *ll, int G)
{
    if (ll->sched_grid == G && ll->d_sched.p) return POLEE_OK;
    const PsellHost &h = ll->host;
    std::vector<uint32_t> order;
    order.reserve((size_t)h.num_tiles_s);
    for (int64_t t = 0; t < h.num_tiles_s; ++t) order.push_back((uint32_t)t);  // (the wide mixed tiles behind them go to the per-tile kernel)
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return ll->tile_cost[a] > ll->tile_cost[b]; });
    // longest tile first, each to the workgroup with the least work so far (ties: the lowest index): with a handful of
    // tiles per workgroup the makespan of this greedy rule is within a few per cent of the mean load, where dealing the
    // sorted tiles out in rounds left the last round's granularity as a tail
    std::vector<std::vector<uint32_t>> lists((size_t)G);
    {
        typedef std::pair<double, int> Load;  // (work so far, workgroup)
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
        for (int b = 0; b < G; ++b) heap.push(Load(0.0, b));
        for (uint32_t t : order) {
            Load l = heap.top();
            heap.pop();
            lists[(size_t)l.second].push_back(t);
            l.first += (double)ll->tile_cost[t];
            heap.push(l);
        }
    }
    size_t rounds = 0;
    for (auto &l : lists) rounds = std::max(rounds, l.size());
    std::vector<PosDesc> sched((rounds + 3) * (size_t)G);  // (the kernel reads up to two rounds past a column's end)
    for (auto &d : sched) {
        d = PosDesc();
        d.tile = POS_NONE;
    }
    for (int b = 0; b < G; ++b) {
        std::vector<uint32_t> big, small;
        for (uint32_t t : lists[b]) (h.stream_of_tile(t) != PSELL_A2 && h.stream_of_tile(t) != PSELL_A2M ? big : small).push_back(t);
        const size_t n = big.size() + small.size();
        size_t ib = 0, is = 0;
        for (size_t j = 0; j < n; ++j) {
            // position j takes a small tile when the running share of small tiles falls behind (phase shifted per block)
            const size_t want = ((j + 1 + (size_t)(b % 3)) * small.size()) / (n + 2);
            const bool take_small = (is < small.size()) && (ib >= big.size() || is < want);
            const uint32_t t = take_small ? small[is++] : big[ib++];
            PosDesc &d = sched[j * (size_t)G + (size_t)b];
            d.tile = t;
            d.s0 = h.tile_slice[t];
            d.s1 = h.tile_slice[t + 1];
            d.d0 = h.tile_dict[t];
            d.L = h.tile_cols[t];
            d.c1 = ll->tile_cut[(size_t)3 * t];
            d.c2 = ll->tile_cut[(size_t)3 * t + 1];
            d.c3 = ll->tile_cut[(size_t)3 * t + 2];
        }
    }
    POLEE_TRY(ll->d_sched.upload(ll->ctx, sched));
    ll->sched_grid = G;
    if (!ll->d_sched_dyn.p) {
        // the dynamic schedule's list (any grid): the tiles by descending cost, POS_NONE behind them
        // Order: inside every stream by descending cost; the streams MERGED at equal pace (a tile's key is its rank
        // among its stream's tiles divided by their number), so that every stretch of the list holds the streams in
        // their overall proportions -- while a workgroup is on a wide tile (two of its four waves at work) its
        // neighbours on the CU are mostly on narrow ones -- and the list ends with the cheapest tiles of every stream.
        std::vector<uint32_t> dorder(order);
        {
            static const bool plain = getenv("POLEE_DYN_PLAIN_ORDER") != nullptr;  // (A/B: descending cost over all streams)
            std::vector<double> key((size_t)h.num_tiles_s, 0.0);
            size_t cnt[PSELL_NSTREAMS] = {}, seen[PSELL_NSTREAMS] = {};
            for (uint32_t t : order) ++cnt[h.stream_of_tile(t)];
            for (uint32_t t : order) {
                const int st = h.stream_of_tile(t);
                key[t] = ((double)seen[st] + 0.5) / (double)cnt[st];
                ++seen[st];
            }
            if (!plain) std::stable_sort(dorder.begin(), dorder.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
        }
        // (Round 5 tried GUIDED positions -- towards the end of the list a position was a PART of a tile, at most (work still ahead)
        // / (workgroups) cycles, so that the workgroups finish closer together: the extra positions' fixed cost outweighed the
        // better balance on every input, +0.7 .. +4 % kernel time, profiles/r05_guided_positions_ab.txt.  Whole tiles.)
        std::vector<PosDesc> plist;
        plist.reserve(dorder.size());
        for (size_t i = 0; i < dorder.size(); ++i) {
            const uint32_t t = dorder[i];
            PosDesc d = PosDesc();
            d.tile = t;
            d.s0 = h.tile_slice[t];
            d.s1 = h.tile_slice[t + 1];
            d.d0 = h.tile_dict[t];
            d.L = h.tile_cols[t];
            d.c1 = ll->tile_cut[(size_t)3 * t];
            d.c2 = ll->tile_cut[(size_t)3 * t + 1];
            d.c3 = ll->tile_cut[(size_t)3 * t + 2];
            plist.push_back(d);
        }
        ll->dyn_positions = plist.size();
        std::vector<PosDesc> dynl(plist.size() + (size_t)4 * 4 * 256 + 64);
        for (auto &d : dynl) {
            d = PosDesc();
            d.tile = POS_NONE;
        }
        std::copy(plist.begin(), plist.end(), dynl.begin());
        ll->dyn_pad = dynl.size() - plist.size();
        POLEE_TRY(ll->d_sched_dyn.upload(ll->ctx, dynl));
        std::vector<unsigned int> zero(2, 0u);
        POLEE_TRY(ll->d_dyn_ctr.upload(ll->ctx, zero));
    }
    return POLEE_OK;
}

template <int K, bool LP, bool KS, bool DET>
static polee_status launch_stream(polee_loglik *ll, PsellArgs &A, int dbg)
{
    polee_ctx *ctx = ll->ctx;
    const PsellHost &h = ll->host;
    hipStream_t st = ctx->stream;
    // a uniform slice of w transcripts occupies (w+1)*256 bytes and may start 768 bytes into a 1 KiB piece
    static_assert((PSELL_NARROW_MAX + 2) * 256 + 1024 <= STREAM_RB1, "A1 slices (+ ks row) must fit their ring");
    static_assert(((PSELL_MIXED_NARROW_MAX * 384 + 255) & ~255) + 256 + 1024 <= STREAM_RB1, "BN slices (+ ks row) must fit their ring");
    static_assert((PSELL_WIDE_MAX + 3) * 256 + 1024 <= STREAM_RB2, "A2M slices (+ ks row) must fit their ring");  // (A2: wide_stream's own assertion)
    const size_t lds = stream_lds_bytes<K, DET>();
    int &occ = ll->occ_cache[K][LP ? 1 : 0][KS ? 1 : 0][DET ? 1 : 0];
    if (occ == 0) {
        int nb = 0;
        POLEE_HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, loglik_stream_kernel<K, LP, KS, DET>, 256, lds));
        occ = std::max(1, std::min(nb, 4));
        if (getenv("POLEE_DEBUG_PRINT"))
            fprintf(stderr, "[loglik] stream kernel K=%d%s: %d workgroups per CU by the occupancy query, LDS %zu B, grid %d x %d\n",
                    K, DET ? " (deterministic)" : "", nb, lds, occ, ctx->num_cus);
    }
    static const int wg_env = getenv("POLEE_STREAM_WGS_PER_CU") ? atoi(getenv("POLEE_STREAM_WGS_PER_CU")) : 0;  // (A/B)
    const int per_cu = wg_env > 0 ? std::min(wg_env, occ) : occ;
    const int G = (int)std::min<int64_t>((int64_t)per_cu * ctx->num_cus, std::max<int64_t>(h.num_tiles_s, 1));
    POLEE_TRY(ensure_schedule(ll, G));  // (built at creation for the usual grid: no host work here)
    A.sched = ll->d_sched.p;
    static const bool static_sched = getenv("POLEE_STATIC_SCHED") != nullptr;  // (A/B)
    // (The deterministic mode draws its tiles too: a tile's gradient goes to the TILE's slot of gwin, in wave order, whichever
    // workgroup works on it.  Only lp with DET keeps the static lists -- its partial sums are per workgroup.)
    static const bool det_static = getenv("POLEE_DET_STATIC") != nullptr;  // (A/B)
    if (!(DET && (LP || det_static)) && !static_sched && (size_t)3 * (size_t)G + 8 <= ll->dyn_pad) {
        A.sched_dyn = ll->d_sched_dyn.p;
        A.dyn_ctr = ll->d_dyn_ctr.p;
        // (the counter runs on from launch to launch, modulo 2^32: this launch's draws are dyn_base, dyn_base + 1, ...)
        A.dyn_base = ll->dyn_base;
        A.dyn_last = (unsigned int)(ll->dyn_positions + ll->dyn_pad - 1);
        ll->dyn_base += (uint32_t)ll->dyn_positions;
    }
    // (DET: A.gwin / A.lpwin are the caller's, and so is the reduce launch behind this one and stream B's: launch_variant)
    if (!ll->xwin_ready) {
        const bool side = ll->side_part != nullptr && !ll->side_done;
        hipLaunchKernelGGL((xwin_gather_kernel<K>), dim3((unsigned)ceil_div(ll->dict_len, 256) + (side ? 1u : 0u)), dim3(256), 0, st, A.dict,
                           A.x, ll->dict_len, ll->d_xwin.p, side ? ll->side_part : nullptr, ll->side_nparts, ll->side_out);
        if (side) ll->side_done = true;
    }
    if (ll->cur_e0) (void)hipEventRecord(ll->cur_e0, st);
    hipLaunchKernelGGL((loglik_stream_kernel<K, LP, KS, DET>), dim3((unsigned)G), dim3(256), lds, st, A, dbg);
    if (ll->cur_e1) (void)hipEventRecord(ll->cur_e1, st);
    A.lp_slot0 = G;  // (DET: stream B's tiles put their lp behind the workgroups')
    return POLEE_OK;
}

template <int K, bool LP, bool KS>
static polee_status launch_variant(polee_loglik *ll, const float *d_x, float *d_g, double *d_lp)
{
    polee_ctx *ctx = ll->ctx;
    const PsellHost &h = ll->host;
    const int lcap_all = std::max(h.max_tile_cols, 1);
    hipStream_t st = ctx->stream;
    static const bool no_ring_env = getenv("POLEE_NO_RING") != nullptr;
    const bool no_ring = no_ring_env || ll->force_mixed;
    static const int dbg = getenv("POLEE_DBG_ABLATE") ? atoi(getenv("POLEE_DBG_ABLATE")) : 0;
    const LoglikRemap *rm = ll->cur_remap;
    const uint32_t *csr_col = rm && rm->csr_col ? rm->csr_col : ll->d_csr_col.p;
    PsellArgs A{ll->d_data.p, ll->d_slice_off.p, ll->d_tile_slice.p, ll->d_tile_dict.p, rm ? rm->dict : ll->d_dict.p,
                ll->d_slice_ks.p, d_x, d_g, d_lp, lcap_all, (int)h.num_tiles_a,
                (int)h.num_tiles_a1, (int)h.num_tiles_a1m, (int)h.num_tiles_a2, (int)h.num_tiles_s, ll->d_xwin.p, nullptr, nullptr, nullptr, 0u, 0u, nullptr, nullptr, 0};
    const size_t lds_psell = (size_t)2 * lcap_all * K * sizeof(float);
    // The attribute belongs to (device, kernel instance): set before every launch that needs it (a host-side table
    // write), so that a second context on another GPU of the same process gets it too; checked.
    const int64_t tiles_b = h.num_tiles - h.num_tiles_s;
    if (lds_psell > 48 * 1024 && (tiles_b > 0 || no_ring))
        POLEE_HIP_TRY(ctx, hipFuncSetAttribute((const void *)loglik_psell_kernel<K, LP, KS>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    if (!no_ring) {
        const bool det = ll->deterministic;
        if (det) {  // every tile's sums go to its slots of gwin; a transcript's slots are added in tile order afterwards
            POLEE_TRY(ll->d_gwin.alloc(ctx, (size_t)ll->dict_len * PSELL_MAX_K + 512));
            POLEE_TRY(ll->d_lpwin.alloc(ctx, ((size_t)4 * ctx->num_cus + (size_t)tiles_b) * PSELL_MAX_K));
            A.gwin = ll->d_gwin.p;
            A.lpwin = ll->d_lpwin.p;
        }
        if (h.num_tiles_s > 0) {
            if (det)
                POLEE_TRY((launch_stream<K, LP, KS, true>(ll, A, dbg)));
            else
                POLEE_TRY((launch_stream<K, LP, KS, false>(ll, A, dbg)));
        }
        if (tiles_b > 0) {
            // rows with more than 32 transcripts (rare) live in mixed tiles, which the per-tile kernel takes (float atomics; in the
            // deterministic mode one wave per tile and stores to the tile's slots, see psell_tile_body)
            hipLaunchKernelGGL((loglik_psell_kernel<K, LP, KS>), dim3((unsigned)tiles_b), dim3(256), lds_psell, st, A,
                               (int)h.num_tiles_s, (const uint32_t *)nullptr);
        }
        if (det) {
            const uint32_t *gmap = ll->cur_remap ? ll->cur_remap->index_of : nullptr;
            hipLaunchKernelGGL(gwin_reduce_kernel, dim3((unsigned)ceil_div(ll->n * K, 256)), dim3(256), 0, st, ll->d_tslot_ptr.p,
                               ll->d_tslot.p, ll->d_gwin.p, K, ll->n, A.g, LP ? ll->d_lpwin.p : nullptr, A.lp_slot0 + (int)tiles_b, A.lp, gmap);
            if (ll->d_theavy.n > 0)
                hipLaunchKernelGGL(gwin_reduce_heavy_kernel, dim3((unsigned)ll->d_theavy.n, (unsigned)K), dim3(256), 0, st, ll->d_theavy.p,
                                   ll->d_tslot_ptr.p, ll->d_tslot.p, ll->d_gwin.p, K, A.g, gmap);
        }
        if (ll->csr_rows > 0)  // stream C: rows kept in CSR (float atomics: like stream B, outside the deterministic guarantee)
            hipLaunchKernelGGL((loglik_csr_kernel<K, LP, KS>), dim3((unsigned)ceil_div(ll->csr_rows, 256)), dim3(256), 0, st,
                               ll->d_csr_rowptr.p, csr_col, ll->d_csr_val.p, ll->d_csr_ks.p, ll->csr_rows, d_x, d_g, d_lp);
    } else {
        if (ll->csr_rows > 0)
            hipLaunchKernelGGL((loglik_csr_kernel<K, LP, KS>), dim3((unsigned)ceil_div(ll->csr_rows, 256)), dim3(256), 0, st,
                               ll->d_csr_rowptr.p, csr_col, ll->d_csr_val.p, ll->d_csr_ks.p, ll->csr_rows, d_x, d_g, d_lp);
        // the cross-check switch: every tile as mixed slices with the per-run DPP kernel
        if (ll->cur_e0) (void)hipEventRecord(ll->cur_e0, st);
        hipLaunchKernelGGL((loglik_psell_kernel<K, LP, KS>), dim3((unsigned)h.num_tiles), dim3(256), lds_psell, st, A, 0,
                           (const uint32_t *)nullptr);
        if (ll->cur_e1) (void)hipEventRecord(ll->cur_e1, st);
    }
    POLEE_KERNEL_CHECK(ctx);
    return POLEE_OK;
}

template <int K>
static polee_status launch_k(polee_loglik *ll, const float *d_x, float *d_g, double *d_lp)
{
    if (ll->host.num_tiles == 0 && ll->csr_rows == 0) return POLEE_OK;
    if (d_lp) return ll->has_ks ? launch_variant<K, true, true>(ll, d_x, d_g, d_lp)
                                : launch_variant<K, true, false>(ll, d_x, d_g, d_lp);
    return ll->has_ks ? launch_variant<K, false, true>(ll, d_x, d_g, d_lp)
                      : launch_variant<K, false, false>(ll, d_x, d_g, d_lp);
}

polee_status loglik_eval_device(polee_loglik *ll, const float *d_x, int K, float *d_g, double *d_lp, bool xwin_ready,
                                const LoglikRemap *remap)
{
    ll->xwin_ready = xwin_ready;
    ll->cur_remap = remap;
    polee_ctx *ctx = ll->ctx;
    if (K < 1 || K > PSELL_MAX_K) return fail(ctx, POLEE_ERR_BAD_ARG, "K must be in 1..8 (got %d)", K);
    // (the pass is a single launch: one pair of events brackets both the kernel and the pass)
    // (profiling: one pair of events brackets the dominant launch, a second pair the whole pass -- the x-window gather in
    // front of it and the mixed stream's launch behind it)
    hipEvent_t e0 = nullptr, e1 = nullptr, p0 = nullptr, p1 = nullptr;
    // (profile = N > 1: every N-th pass only.  Four event records per pass are four barrier packets in the stream -- measured
    // with rocprofv3, round 5: 22 us of idle gaps per VI iteration when EVERY pass is bracketed, 5 % of a C2 step)
    if (ll->profile && (ll->prof_every <= 1 || ll->prof_tick++ % (uint64_t)ll->prof_every == 0)) {
        if (ll->prof_used + 4 > ll->prof_events.size()) {
            if (ll->prof_events.size() >= 8192) POLEE_TRY(ll->profile_collect());
            while (ll->prof_used + 4 > ll->prof_events.size()) {
                hipEvent_t a;
                POLEE_HIP_TRY(ctx, hipEventCreate(&a));
                ll->prof_events.push_back(a);
            }
        }
        e0 = ll->prof_events[ll->prof_used];
        e1 = ll->prof_events[ll->prof_used + 1];
        p0 = ll->prof_events[ll->prof_used + 2];
        p1 = ll->prof_events[ll->prof_used + 3];
        ll->prof_used += 4;
    }
    ll->cur_e0 = e0;
    ll->cur_e1 = e1;
    if (p0) (void)hipEventRecord(p0, ctx->stream);
    polee_status st;
    switch (K) {
        case 1: st = launch_k<1>(ll, d_x, d_g, d_lp); break;
        case 2: st = launch_k<2>(ll, d_x, d_g, d_lp); break;
        case 3: st = launch_k<3>(ll, d_x, d_g, d_lp); break;
        case 4: st = launch_k<4>(ll, d_x, d_g, d_lp); break;
        case 5: st = launch_k<5>(ll, d_x, d_g, d_lp); break;
        case 6: st = launch_k<6>(ll, d_x, d_g, d_lp); break;
        case 7: st = launch_k<7>(ll, d_x, d_g, d_lp); break;
        default: st = launch_k<8>(ll, d_x, d_g, d_lp); break;
    }
    if (st != POLEE_OK && ll->d_dyn_ctr.p) {  // a failed launch may have made only some of its draws: counter and base start over
        (void)hipMemsetAsync(ll->d_dyn_ctr.p, 0, 2 * sizeof(unsigned int), ctx->stream);
        ll->dyn_base = 0;
    }
    ll->cur_remap = nullptr;
    // stream S; a caller whose forward kernel has already written cnt / x into g only needs the log-likelihood's share
    if (st == POLEE_OK && ll->has_singles && !(remap && remap->singles_in_g && !d_lp)) {
        const int nb = (int)ceil_div(ll->n, SINGLE_THREADS);
        hipLaunchKernelGGL(single_rows_kernel, dim3((unsigned)nb), dim3(SINGLE_THREADS), 0, ctx->stream,
                           remap && remap->single_cnt ? remap->single_cnt : ll->d_single_cnt.p, d_x, K,
                           ll->n, remap && remap->singles_in_g ? nullptr : d_g, d_lp ? ll->d_single_part.p : nullptr);
        if (d_lp)
            hipLaunchKernelGGL(single_lp_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, ll->d_single_part.p, nb, K,
                               ll->host.single_logsum, d_lp);
        POLEE_KERNEL_CHECK(ctx);
    }
    if (p1) (void)hipEventRecord(p1, ctx->stream);
    return st;
}

// [rows][n] <-> [n][rows] re-layout between the host API (one expression vector per row)
// and the kernel's transcript-major layout.
__global__ void rows_to_aos_kernel(const float *in, int K, int64_t n, float *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K) return;
    const int64_t j = i / K;
    const int k = (int)(i - j * K);
    out[i] = in[(int64_t)k * n + j];
}
__global__ void aos_to_rows_f64_kernel(const float *in, int K, int64_t n, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * K) return;
    const int k = (int)(i / n);
    const int64_t j = i - (int64_t)k * n;
    out[i] = (double)in[j * K + k];
}

// effective_length_jacobian_adjustment! (src/likelihood.jl:93-110), host-pointer form.
__global__ void efflen_sum_kernel(const float *efflens, const float *xs, int64_t n, double *sums)
{
    __shared__ double smd[4];
    const int row = blockIdx.y;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += (double)(xs[(int64_t)row * n + i] / efflens[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if ((threadIdx.x & 63) == 0) smd[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[row], smd[0] + smd[1] + smd[2] + smd[3]);
}
__global__ void efflen_adjust_kernel(const float *efflens, const float *xs, const double *sums, int64_t n,
                                     double *x_grad, float *xls)
{
    const int row = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double c = sums[row];
    const float inv_l = 1.0f / efflens[i];
    x_grad[(int64_t)row * n + i] -= (double)((float)n * inv_l) / c;  // n * (1/efflens[i]) is Float32 in the reference
    if (xls) xls[(int64_t)row * n + i] = (float)((double)(xs[(int64_t)row * n + i] / efflens[i]) / c);
}

}  // namespace polee

namespace polee {
void loglik_retain(polee_loglik *ll)
{
    if (ll) ++ll->refs;
}
void loglik_release(polee_loglik *ll)
{
    if (!ll || --ll->refs > 0) return;
    polee_ctx *ctx = ll->ctx;
    if (ctx) (void)hipSetDevice(ctx->device);
    for (hipEvent_t e : ll->prof_events) (void)hipEventDestroy(e);
    delete ll;
    ctx_release(ctx);
}
}  // namespace polee

using namespace polee;

polee_status polee_loglik::profile_collect()
{
    if (prof_used == 0) return POLEE_OK;
    POLEE_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i + 3 < prof_used; i += 4) {
        float ms = 0.f, pass_ms = 0.f;
        // (a sample without uniform tiles never records the kernel pair: its pass is the mixed launch)
        if (hipEventElapsedTime(&ms, prof_events[i], prof_events[i + 1]) != hipSuccess) ms = 0.f;
        POLEE_HIP_TRY(ctx, hipEventElapsedTime(&pass_ms, prof_events[i + 2], prof_events[i + 3]));
        prof_ms_total += ms;
        prof_pass_ms_total += pass_ms;
        ++prof_launches;
    }
    prof_used = 0;
    return POLEE_OK;
}

static double wall_now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static polee_status loglik_finish_create(polee_ctx *ctx, polee_loglik *ll, polee_loglik **out)
{
    PsellHost &h = ll->host;
    polee_status s;
    static const bool timing = getenv("POLEE_BUILD_TIMING") != nullptr;
    const double t_begin = wall_now();
    if (!ll->device_built) h.data.resize(h.data.size() + 2048, 0);  // slack: the LDS-DMA stream reads whole 1 KiB pieces
    // (device_built: the slice stream and the multiplicities were laid out on the device, psell_device.hip, and are in place)
    if ((!ll->device_built && (s = ll->d_data.upload(ctx, h.data.data(), h.data.size()))) || (s = ll->d_slice_off.upload(ctx, h.slice_off)) ||
        (s = ll->d_tile_slice.upload(ctx, h.tile_slice)) || (s = ll->d_tile_dict.upload(ctx, h.tile_dict)) ||
        (s = ll->d_dict.upload(ctx, h.dict)) ||
        (ll->has_ks && !ll->device_built && (s = ll->d_slice_ks.upload(ctx, h.slice_ks))) ||
        (!h.csr_rows.empty() && ((s = ll->d_csr_rowptr.upload(ctx, h.csr_rowptr)) || (s = ll->d_csr_col.upload(ctx, h.csr_col)) ||
                                 (s = ll->d_csr_val.upload(ctx, h.csr_val)) || (ll->has_ks && (s = ll->d_csr_ks.upload(ctx, h.csr_ks)))))) {
        loglik_release(ll);
        return s;
    }
    ll->csr_rows = (int64_t)h.csr_rows.size();
    ll->csr_nnz = (int64_t)h.csr_col.size();
    if (!h.single_cnt.empty()) {
        if ((s = ll->d_single_cnt.upload(ctx, h.single_cnt)) ||
            (s = ll->d_single_part.alloc(ctx, (size_t)ceil_div(ll->n, SINGLE_THREADS) * PSELL_MAX_K))) {
            loglik_release(ll);
            return s;
        }
        ll->has_singles = true;
        std::vector<float>().swap(h.single_cnt);
    }
    if (timing) fprintf(stderr, "[loglik create] %-28s %.3f s\n", "upload", wall_now() - t_begin);
    // what the streaming kernel's schedule needs, before the bulk vectors go: the relative cost of every tile
    // (bytes it streams; the latency-bound streams weigh more per byte), the x windows, the usual grid's schedule
    ll->dict_len = (int64_t)h.dict.size();
    ll->tile_cost.assign((size_t)h.num_tiles, 0.0f);
    for (int64_t t = 0; t < h.num_tiles; ++t)
        ll->tile_cost[(size_t)t] = (float)slices_cost(h, h.stream_of_tile(t), h.tile_slice[t], h.tile_slice[t + 1]);
    ll->tile_cut.assign((size_t)3 * h.num_tiles, 0u);
    for (int64_t t = 0; t < h.num_tiles; ++t) wave_cuts(h, t, h.tile_slice[t], h.tile_slice[t + 1], &ll->tile_cut[(size_t)3 * t]);
    {   // deterministic mode: the dictionary entries of every transcript, ascending (= tile order), padding left out
        std::vector<uint32_t> ptr((size_t)ll->n + 1, 0), slots;
        for (int64_t t = 0; t < h.num_tiles; ++t)  // (stream B's tiles too: their sums go through gwin in the deterministic mode)
            for (uint32_t l = 0; l < h.tile_cols[t]; ++l) ++ptr[(size_t)h.dict[h.tile_dict[t] + l] + 1];
        for (int64_t j = 0; j < ll->n; ++j) ptr[(size_t)j + 1] += ptr[(size_t)j];
        slots.resize(ptr[(size_t)ll->n]);
        std::vector<uint32_t> cur(ptr.begin(), ptr.end() - 1);
        for (int64_t t = 0; t < h.num_tiles; ++t)
            for (uint32_t l = 0; l < h.tile_cols[t]; ++l) {
                const uint32_t e = h.tile_dict[t] + l;
                slots[cur[h.dict[e]]++] = e;
            }
        std::vector<uint32_t> heavy;
        for (int64_t j = 0; j < ll->n; ++j)
            if (ptr[(size_t)j + 1] - ptr[(size_t)j] > GWIN_HEAVY) heavy.push_back((uint32_t)j);
        if ((s = ll->d_tslot_ptr.upload(ctx, ptr)) || (s = ll->d_tslot.upload(ctx, slots)) || (s = ll->d_theavy.upload(ctx, heavy))) {
            loglik_release(ll);
            return s;
        }
    }
    if ((s = ll->d_xwin.alloc(ctx, (size_t)ll->dict_len * PSELL_MAX_K + 512)) ||
        // (the dictionaries' padding entries are never referenced by a slice; they are read into LDS with their tile's
        // window, and the VI loop's forward kernel -- which fills the windows through the slot lists -- does not write them)
        (hipMemsetAsync(ll->d_xwin.p, 0, ((size_t)ll->dict_len * PSELL_MAX_K + 512) * sizeof(float), ctx->stream) != hipSuccess &&
         (s = fail(ctx, POLEE_ERR_HIP, "hipMemset failed")) != POLEE_OK) ||
        (s = ensure_schedule(ll, (int)std::min<int64_t>((int64_t)4 * ctx->num_cus, std::max<int64_t>(h.num_tiles, 1))))) {
        loglik_release(ll);
        return s;
    }
    if (timing) fprintf(stderr, "[loglik create] %-28s %.3f s\n", "upload + schedule + lists", wall_now() - t_begin);
    // keep only metadata on the host
    decltype(h.data)().swap(h.data);
    std::vector<uint32_t>().swap(h.slice_off);
    std::vector<uint32_t>().swap(h.dict);
    std::vector<float>().swap(h.slice_ks);
    std::vector<uint8_t>().swap(h.slice_flags);
    std::vector<uint8_t>().swap(h.slice_w);
    std::vector<uint32_t>().swap(h.row_order);
    std::vector<uint32_t>().swap(h.csr_rowptr);
    std::vector<uint32_t>().swap(h.csr_col);
    std::vector<uint32_t>().swap(h.csr_rows);
    std::vector<float>().swap(h.csr_val);
    std::vector<float>().swap(h.csr_ks);
    *out = ll;
    return POLEE_OK;
}

namespace polee {
// CSC (1-based, as in the HDF5) -> CSR (0-based).  Columns stay ascending within a row.
std::string csc_to_csr(int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                       const float *nzval, BVec<uint64_t> &rowptr, RawVec<uint32_t> &col, RawVec<float> &val)
{
    auto cp = [&](int64_t j) -> uint64_t {
        return colptr_bytes == 4 ? (uint64_t) reinterpret_cast<const uint32_t *>(colptr)[j]
                                 : reinterpret_cast<const uint64_t *>(colptr)[j];
    };
    if (colptr_bytes != 4 && colptr_bytes != 8) return "colptr_bytes must be 4 or 8";
    if (cp(0) != 1) return "colptr[0] must be 1 (1-based)";
    const double t_enter = wall_now();
    const uint64_t nnz = cp(n) - 1;
    for (int64_t j = 0; j < n; ++j)
        if (cp(j + 1) < cp(j)) return "colptr is not monotone";
    rowptr.assign(m + 1, 0);
    // Transposition in two partitioned passes on the host threads, without atomics and without a sort:
    //   the columns are cut into chunks of equal nnz, the rows into buckets of 2^sh consecutive rows (a bucket's cursors
    //   stay in a core's L2);
    //   pass 1: every chunk counts its entries per bucket; a prefix over (bucket, chunk) gives every (chunk, bucket) pair
    //           its own contiguous piece of a staging array, bucket-major, chunks in column order inside a bucket;
    //   pass 2: every chunk writes (row, column, value) records into its pieces -- a few hundred sequential write
    //           streams per thread instead of 240 M random ones;
    //   pass 3: every bucket counts its rows, prefixes (that IS its part of rowptr) and places its records: they arrive
    //           in ascending column order, so every row comes out sorted by transcript, as a sequential transposition
    //           would leave it.
    col.clear();
    val.clear();
    if (nnz == 0) return "";
    if (m < 1) return "rowval out of range";
    static const bool timing = getenv("POLEE_BUILD_TIMING") != nullptr;
    double t_prev = t_enter;
    auto lap = [&](const char *what) {
        if (timing) fprintf(stderr, "[CSC -> rows] %-30s %.3f s\n", what, wall_now() - t_prev);
        t_prev = wall_now();
    };
    int sh = 0;
    while (((uint64_t)m >> sh) > 512) ++sh;  // <= 512 buckets (+1)
    if (sh < 12) sh = 12;
    const size_t B = (size_t)(((uint64_t)m - 1) >> sh) + 1;
    const size_t NC = std::max<size_t>(1, std::min<size_t>(4 * host_threads(), (size_t)(nnz >> 16) + 1));
    std::vector<int64_t> cstart(NC + 1, n);  // chunk c = columns [cstart[c], cstart[c+1])
    for (size_t c = 0; c <= NC; ++c) {
        const uint64_t target = 1 + nnz * c / NC;  // first column whose start is >= target
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) / 2;
            if (cp(mid) < target) lo = mid + 1; else hi = mid;
        }
        cstart[c] = c == NC ? n : lo;
    }
    cstart[0] = 0;
    std::vector<uint64_t> cnt(NC * B, 0);
    std::atomic<int> err{0};
    parallel_chunks(NC, 1, [&](size_t clo, size_t chi, unsigned) {
        for (size_t c = clo; c < chi; ++c) {
            uint64_t *cc = cnt.data() + c * B;
            for (uint64_t k = cp(cstart[c]) - 1; k < cp(cstart[c + 1]) - 1; ++k) {
                const uint32_t r = rowval[k];
                if (r < 1 || (int64_t)r > m) {
                    err = 1;
                    continue;
                }
                ++cc[(r - 1) >> sh];
            }
        }
    });
    if (err) return "rowval out of range";
    lap("checks + count per (chunk, bucket)");
    std::vector<uint64_t> bstart(B + 1, 0);
    {
        uint64_t run = 0;
        for (size_t b = 0; b < B; ++b) {
            bstart[b] = run;
            for (size_t c = 0; c < NC; ++c) {
                const uint64_t t = cnt[c * B + b];
                cnt[c * B + b] = run;  // becomes the write cursor of (chunk, bucket)
                run += t;
            }
        }
        bstart[B] = run;
    }
    struct Rec {
        uint32_t row, col;
        float val;
    };
    std::vector<Rec, default_init_allocator<Rec>> stage((size_t)nnz);
    lap("allocate staging");
    parallel_chunks(NC, 1, [&](size_t clo, size_t chi, unsigned) {
        for (size_t c = clo; c < chi; ++c) {
            uint64_t *cur = cnt.data() + c * B;
            for (int64_t j = cstart[c]; j < cstart[c + 1]; ++j)
                for (uint64_t k = cp(j) - 1; k < cp(j + 1) - 1; ++k) {
                    const uint32_t r = rowval[k] - 1;
                    stage[cur[r >> sh]++] = Rec{r, (uint32_t)j, nzval[k]};
                }
        }
    });
    lap("scatter into buckets");
    col.resize(nnz);
    val.resize(nnz);
    lap("allocate rows");
    parallel_chunks(B, 1, [&](size_t blo, size_t bhi, unsigned) {
        std::vector<uint32_t> local;
        for (size_t b = blo; b < bhi; ++b) {
            const uint64_t r0 = (uint64_t)b << sh, r1 = std::min<uint64_t>((uint64_t)m, r0 + ((uint64_t)1 << sh));
            local.assign((size_t)(r1 - r0) + 1, 0);
            for (uint64_t p = bstart[b]; p < bstart[b + 1]; ++p) ++local[stage[p].row - r0 + 1];
            for (size_t i = 1; i < local.size(); ++i) local[i] += local[i - 1];
            for (uint64_t i = 0; i < r1 - r0; ++i) rowptr[r0 + i + 1] = bstart[b] + local[i + 1];
            for (uint64_t p = bstart[b]; p < bstart[b + 1]; ++p) {
                const Rec &e = stage[p];
                const uint64_t q = bstart[b] + local[e.row - r0]++;
                col[q] = e.col;
                val[q] = e.val;
            }
        }
    });
    rowptr[0] = 0;
    lap("place within buckets");
    return "";
}
}  // namespace polee

extern "C" {

polee_status polee_loglik_set_deterministic(polee_loglik *ll, int on)
{
    if (!ll) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    ll->deterministic = on != 0;
    return POLEE_OK;
}

polee_status polee_debug_loglik_force_mixed(polee_loglik *ll, int on)
{
    if (!ll) return fail(nullptr, POLEE_ERR_BAD_ARG, "null handle");
    ll->force_mixed = on != 0;
    return POLEE_OK;
}

// The layout built on the device from X by rows in device memory (psell_device.hip).  done = false (and POLEE_OK): the matrix
// is the host builder's case -- a real share of fragments without any structure.
static polee_status loglik_create_on_device(polee_ctx *ctx, const PsellDevIn &X, bool has_ks, polee_loglik **out, bool &done)
{
    done = false;
    polee_loglik *ll = new (std::nothrow) polee_loglik();
    if (!ll) return fail(ctx, POLEE_ERR_OOM, "out of host memory");
    ll->ctx = ctx;
    ctx_retain(ctx);
    ll->m = X.m;
    ll->n = X.n;
    ll->has_ks = has_ks;
    PsellDevOut D;
    bool needs_host = false;
    polee_status st = psell_device_build(ctx, X, ll->host, D, false, needs_host);
    if (st != POLEE_OK || needs_host) {
        loglik_release(ll);
        return st;
    }
    ll->nnz = ll->host.nnz;
    ll->device_built = true;
    ll->d_data.take(D.data);
    if (has_ks) ll->d_slice_ks.take(D.slice_ks);
    done = true;
    return loglik_finish_create(ctx, ll, out);
}

static polee_status polee_loglik_create_from_xt_impl(polee_ctx *ctx, int64_t m, int64_t n, const uint64_t *tcolptr,
                                         const uint32_t *trowval, const float *tnzval, const int64_t *ks,
                                         polee_loglik **out)
{
    POLEE_TRY(use_device(ctx));
    if (!tcolptr || !out || m < 0 || n < 1 || (tcolptr[m] > 1 && (!trowval || !tnzval)))
        return fail(ctx, POLEE_ERR_BAD_ARG, "polee_loglik_create_from_xt: bad argument");
    if (tcolptr[0] != 1) return fail(ctx, POLEE_ERR_BAD_ARG, "tcolptr[0] must be 1 (1-based)");
    const uint64_t nnz = tcolptr[m] - 1;
    const double t_begin = wall_now();
    if (psell_device_enabled() && nnz < (1ull << 32) - 1 && m < ((int64_t)1 << 32) - 1) {
        polee_status st = POLEE_OK;
        bool done = false;
        {
            PsellDevCSR C;
            if ((st = psell_device_rows_from_xt(ctx, m, n, tcolptr, trowval, tnzval, ks, false, C)) != POLEE_OK) return st;
            st = loglik_create_on_device(ctx, C.view(), ks != nullptr, out, done);
        }
        if (st != POLEE_OK || done) return st;
    }
    // (0-based copies; vectors that do not zero-fill a gigabyte first, filled on several threads)
    std::vector<uint64_t, default_init_allocator<uint64_t>> rowptr(m + 1);
    parallel_chunks((size_t)m + 1, (size_t)1 << 20, [&](size_t lo, size_t hi, unsigned) {
        for (size_t i = lo; i < hi; ++i) rowptr[i] = tcolptr[i] - 1;
    });
    std::vector<uint32_t, default_init_allocator<uint32_t>> col(nnz);
    {
        std::atomic<int> bad{0};
        parallel_chunks((size_t)nnz, (size_t)1 << 20, [&](size_t lo, size_t hi, unsigned) {
            for (size_t k = lo; k < hi; ++k) {
                if (trowval[k] < 1) bad = 1;
                col[k] = trowval[k] - 1;
            }
        });
        if (bad) return fail(ctx, POLEE_ERR_BAD_ARG, "trowval must be 1-based");
    }
    polee_loglik *ll = new (std::nothrow) polee_loglik();
    if (!ll) return fail(ctx, POLEE_ERR_OOM, "out of host memory");
    ll->ctx = ctx;
    ctx_retain(ctx);
    ll->m = m;
    ll->n = n;
    ll->nnz = (int64_t)nnz;
    ll->has_ks = ks != nullptr;
    if (getenv("POLEE_BUILD_TIMING")) fprintf(stderr, "[loglik create] %-28s %.3f s\n", "0-based copies of the input", wall_now() - t_begin);
    std::string err = build_psell(m, n, rowptr.data(), col.data(), tnzval, ks, ll->host);
    if (!err.empty()) {
        loglik_release(ll);
        return fail(ctx, err.find("more than") != std::string::npos ? POLEE_ERR_UNSUPPORTED : POLEE_ERR_BAD_ARG,
                    "likelihood matrix: %s", err.c_str());
    }
    return loglik_finish_create(ctx, ll, out);
}

int polee_loglik_built_on_device(const polee_loglik *ll) { return ll && ll->device_built ? 1 : 0; }

static polee_status polee_loglik_create_from_xbuild_impl(polee_ctx *ctx, const polee_xbuild *xb, const int64_t *ks, polee_loglik **out)
{
    POLEE_TRY(use_device(ctx));
    if (!xb || !out) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_loglik_create_from_xbuild: null argument");
    polee_ctx *xctx = nullptr;
    int64_t m = 0, n = 0;
    const uint64_t *tcolptr = nullptr;
    const uint32_t *trowval = nullptr;
    const float *tnzval = nullptr;
    POLEE_TRY(xbuild_device_view(xb, &xctx, &m, &n, &tcolptr, &trowval, &tnzval));
    if (xctx->device != ctx->device) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_loglik_create_from_xbuild: the xbuild result lives on another device");
    POLEE_HIP_TRY(ctx, hipStreamSynchronize(xctx->stream));
    bool done = false;
    polee_status st = POLEE_OK;
    uint64_t nnz_dev = 0;  // (the device builder numbers non-zeros in 32 bits, as the from_xt path checks: ADVICE r4)
    if (m > 0) {
        POLEE_HIP_TRY(ctx, hipMemcpy(&nnz_dev, tcolptr + m, sizeof nnz_dev, hipMemcpyDeviceToHost));
        nnz_dev -= 1;
    }
    if (psell_device_enabled() && nnz_dev < (1ull << 32) - 1 && m < ((int64_t)1 << 32) - 1) {
        PsellDevCSR C;
        DevBuf<int64_t> d_ks;
        if (ks) POLEE_TRY(d_ks.upload(ctx, ks, (size_t)m));
        if ((st = psell_device_rows_from_xt(ctx, m, n, tcolptr, trowval, tnzval, ks ? d_ks.p : nullptr, true, C)) != POLEE_OK) return st;
        st = loglik_create_on_device(ctx, C.view(), ks != nullptr, out, done);
        if (st != POLEE_OK || done) return st;
    }
    // the host builder's case: through the host arrays
    std::vector<uint64_t> h_ptr((size_t)m + 1);
    POLEE_HIP_TRY(ctx, hipMemcpy(h_ptr.data(), tcolptr, ((size_t)m + 1) * 8, hipMemcpyDeviceToHost));
    const size_t nnz = (size_t)(h_ptr[(size_t)m] - 1);
    std::vector<uint32_t> h_col(nnz);
    std::vector<float> h_val(nnz);
    if (nnz) {
        POLEE_HIP_TRY(ctx, hipMemcpy(h_col.data(), trowval, nnz * 4, hipMemcpyDeviceToHost));
        POLEE_HIP_TRY(ctx, hipMemcpy(h_val.data(), tnzval, nnz * 4, hipMemcpyDeviceToHost));
    }
    return polee_loglik_create_from_xt(ctx, m, n, h_ptr.data(), h_col.data(), h_val.data(), ks, out);
}

polee_status polee_loglik_create_from_xbuild(polee_ctx *ctx, const polee_xbuild *xb, const int64_t *ks, polee_loglik **out)
{
    return guarded(ctx, "polee_loglik_create_from_xbuild", [&] { return polee_loglik_create_from_xbuild_impl(ctx, xb, ks, out); });
}

polee_status polee_loglik_create_from_xt(polee_ctx *ctx, int64_t m, int64_t n, const uint64_t *tcolptr,
                                         const uint32_t *trowval, const float *tnzval, const int64_t *ks,
                                         polee_loglik **out)
{
    return guarded(ctx, "polee_loglik_create_from_xt", [&] { return polee_loglik_create_from_xt_impl(ctx, m, n, tcolptr, trowval, tnzval, ks, out); });
}

static polee_status polee_loglik_create_impl(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes,
                                 const uint32_t *rowval, const float *nzval, const int64_t *ks, polee_loglik **out)
{
    POLEE_TRY(use_device(ctx));
    if (!colptr || !out || m < 0 || n < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_loglik_create: bad argument");
    BVec<uint64_t> rowptr;
    RawVec<uint32_t> col;  // (resize leaves them uninitialised: 1.9 GB of zeros would be written by one thread)
    RawVec<float> val;
    static const bool timing = getenv("POLEE_BUILD_TIMING") != nullptr;
    const double t_begin = wall_now();
    if (psell_device_enabled()) {
        polee_status st = POLEE_OK;
        bool done = false;
        {
            PsellDevCSR C;
            bool needs_host = false;
            if ((st = psell_device_rows_from_csc(ctx, m, n, colptr, colptr_bytes, rowval, nzval, ks, C, needs_host)) != POLEE_OK) return st;
            if (timing) fprintf(stderr, "[loglik create] %-28s %.3f s\n", "X onto the device, by rows", wall_now() - t_begin);
            if (!needs_host) st = loglik_create_on_device(ctx, C.view(), ks != nullptr, out, done);
        }
        if (timing && done) fprintf(stderr, "[loglik create] %-28s %.3f s\n", "total (device build)", wall_now() - t_begin);
        if (st != POLEE_OK || done) return st;
    }
    std::string err = csc_to_csr(m, n, colptr, colptr_bytes, rowval, nzval, rowptr, col, val);
    if (timing) fprintf(stderr, "[loglik create] %-28s %.3f s\n", "CSC -> rows", wall_now() - t_begin);
    if (!err.empty()) return fail(ctx, POLEE_ERR_BAD_ARG, "likelihood matrix: %s", err.c_str());
    polee_loglik *ll = new (std::nothrow) polee_loglik();
    if (!ll) return fail(ctx, POLEE_ERR_OOM, "out of host memory");
    ll->ctx = ctx;
    ctx_retain(ctx);
    ll->m = m;
    ll->n = n;
    ll->nnz = (int64_t)col.size();
    ll->has_ks = ks != nullptr;
    err = build_psell(m, n, rowptr.data(), col.data(), val.data(), ks, ll->host);
    if (!err.empty()) {
        loglik_release(ll);
        return fail(ctx, err.find("more than") != std::string::npos ? POLEE_ERR_UNSUPPORTED : POLEE_ERR_BAD_ARG,
                    "likelihood matrix: %s", err.c_str());
    }
    const polee_status st = loglik_finish_create(ctx, ll, out);
    if (timing) fprintf(stderr, "[loglik create] %-28s %.3f s\n", "total", wall_now() - t_begin);
    return st;
}

polee_status polee_loglik_create(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes,
                                 const uint32_t *rowval, const float *nzval, const int64_t *ks, polee_loglik **out)
{
    return guarded(ctx, "polee_loglik_create", [&] { return polee_loglik_create_impl(ctx, m, n, colptr, colptr_bytes, rowval, nzval, ks, out); });
}

// ---- X on the device, once, for the tree and the layout ------------------------------------------------------
polee_status polee_devx_upload(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                               const float *nzval, polee_devx **out)
{
    return guarded(ctx, "polee_devx_upload", [&]() -> polee_status {
        POLEE_TRY(use_device(ctx));
        if (!colptr || !out || m < 0 || n < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_devx_upload: bad argument");
        std::vector<uint64_t> cp;
        uint64_t nnz = 0;
        POLEE_TRY(psell_check_colptr(ctx, n, colptr, colptr_bytes, cp, nnz));
        if (nnz > 0 && !rowval) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_devx_upload: bad argument");
        if (nnz >= (1ull << 32) - 1 || m >= ((int64_t)1 << 32) - 1)
            return fail(ctx, POLEE_ERR_UNSUPPORTED, "polee_devx_upload: the device builders number rows and non-zeros in 32 bits (use polee_loglik_create / polee_hclust_parallel)");
        polee_devx *dx = new (std::nothrow) polee_devx();
        if (!dx) return fail(ctx, POLEE_ERR_OOM, "out of host memory");
        dx->ctx = ctx;
        dx->m = m;
        dx->n = n;
        dx->nnz = nnz;
        polee_status st = dx->cp.upload(ctx, cp.data(), cp.size());
        if (st == POLEE_OK && nnz) st = dx->rowval.upload(ctx, rowval, (size_t)nnz);
        if (st == POLEE_OK && nnz && nzval) st = dx->nzval.upload(ctx, nzval, (size_t)nnz);
        if (st != POLEE_OK) {
            delete dx;
            return st;
        }
        ctx_retain(ctx);
        *out = dx;
        return POLEE_OK;
    });
}

polee_status polee_devx_upload_values(polee_devx *dx, const float *nzval)
{
    if (!dx) return fail(nullptr, POLEE_ERR_BAD_ARG, "polee_devx_upload_values: null argument");
    polee_ctx *ctx = dx->ctx;
    return guarded(ctx, "polee_devx_upload_values", [&]() -> polee_status {
        POLEE_TRY(use_device(ctx));
        if (dx->nnz && !nzval) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_devx_upload_values: null argument");
        if (dx->nnz) POLEE_TRY(dx->nzval.upload(ctx, nzval, (size_t)dx->nnz));
        return POLEE_OK;
    });
}

void polee_devx_destroy(polee_devx *dx)
{
    if (!dx) return;
    polee_ctx *ctx = dx->ctx;
    (void)hipSetDevice(ctx->device);
    dx->cp.release();
    dx->rowval.release();
    dx->nzval.release();
    delete dx;
    ctx_release(ctx);
}

polee_status polee_loglik_create_from_devx(polee_ctx *ctx, polee_devx *dx, const float *nzval, const int64_t *ks, polee_loglik **out)
{
    return guarded(ctx, "polee_loglik_create_from_devx", [&]() -> polee_status {
        POLEE_TRY(use_device(ctx));
        if (!dx || !out) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_loglik_create_from_devx: null argument");
        if (dx->ctx->device != ctx->device) return fail(ctx, POLEE_ERR_BAD_ARG, "polee_loglik_create_from_devx: X lives on another device");
        if (dx->nnz && !dx->nzval.p && !nzval)
            return fail(ctx, POLEE_ERR_BAD_ARG, "polee_loglik_create_from_devx: the values have not been uploaded (polee_devx_upload_values, or pass them here)");
        const float *late = dx->nnz && !dx->nzval.p ? nzval : nullptr;  // (go up beside the first kernels, into the handle)
        // (device builder switched off -- POLEE_DEVICE_BUILD=0 or a host-builder knob: straight to the host layout builder below,
        // as polee_loglik_create would; the handle's copy of X still saves the tree builder its upload)
        if (psell_device_enabled()) {
            bool done = false;
            polee_status st = POLEE_OK;
            {
                PsellDevCSR C;
                if ((st = psell_device_rows_from_dev_csc(ctx, dx->m, dx->n, dx->cp.p, dx->nnz, dx->rowval.p, dx->nzval.p, ks, C, nullptr, late, &dx->nzval)) != POLEE_OK) return st;
                st = loglik_create_on_device(ctx, C.view(), ks != nullptr, out, done);
            }
            if (st != POLEE_OK || done) return st;
        }
        // the host builder's case (a real share of rows without any structure): through host arrays, as polee_loglik_create would
        std::vector<uint64_t> h_cp((size_t)dx->n + 1);
        std::vector<uint32_t> h_row((size_t)dx->nnz);
        std::vector<float> h_val;
        POLEE_TRY(dx->cp.download(ctx, h_cp.data(), h_cp.size()));
        const float *vals = nullptr;
        if (dx->nnz) {
            POLEE_TRY(dx->rowval.download(ctx, h_row.data(), h_row.size()));
            if (dx->nzval.p) {
                h_val.resize((size_t)dx->nnz);
                POLEE_TRY(dx->nzval.download(ctx, h_val.data(), h_val.size()));
                vals = h_val.data();
            } else {
                vals = nzval;  // (values that were to go up late never did: the caller's host array is the source)
            }
        }
        return polee_loglik_create_impl(ctx, dx->m, dx->n, h_cp.data(), 8, h_row.data(), vals, ks, out);
    });
}

void polee_loglik_destroy(polee_loglik *ll) { loglik_release(ll); }

polee_status polee_loglik_get_info(const polee_loglik *ll, polee_loglik_info *info)
{
    if (!ll || !info) return fail(nullptr, POLEE_ERR_BAD_ARG, "null argument");
    const PsellHost &h = ll->host;
    info->m = ll->m;
    info->n = ll->n;
    info->nnz = ll->nnz;
    info->num_slices = h.num_slices;
    info->num_tiles = h.num_tiles;
    info->padded_nnz = h.padded_nnz;
    info->stream_bytes = (int64_t)(ll->d_data.n + 4 * (ll->d_slice_off.n + ll->d_tile_slice.n + ll->d_tile_dict.n +
                                                        ll->d_dict.n + ll->d_slice_ks.n + ll->d_csr_rowptr.n + ll->d_csr_col.n +
                                                        ll->d_csr_val.n + ll->d_csr_ks.n));
    info->device_bytes = info->stream_bytes;
    info->num_empty_rows = h.empty_rows;
    info->max_row_nnz = h.max_row;
    info->max_tile_cols = h.max_tile_cols;
    const int64_t tiles[PSELL_NSTREAMS] = {h.num_tiles_a1, h.num_tiles_a1m - h.num_tiles_a1, h.num_tiles_a2 - h.num_tiles_a1m,
                                           h.num_tiles_a - h.num_tiles_a2, h.num_tiles_s - h.num_tiles_a, h.num_tiles - h.num_tiles_s, 0};
    for (int i = 0; i < 8; ++i) info->stream_rows[i] = info->stream_nnz[i] = info->stream_tiles[i] = info->stream_bytes_hbm[i] = 0;
    for (int i = 0; i < PSELL_NSTREAMS; ++i) {
        info->stream_rows[i] = h.stream_rows[i];
        info->stream_nnz[i] = h.stream_nnz[i];
        info->stream_tiles[i] = tiles[i];
        info->stream_bytes_hbm[i] = h.stream_bytes[i];
    }
    info->dict_entries = ll->dict_len;
    return POLEE_OK;
}

polee_status polee_loglik_eval(polee_loglik *ll, const float *xs, int32_t K, double *x_grad, double *lp)
{
    if (!ll) return fail(nullptr, POLEE_ERR_BAD_ARG, "null likelihood handle");
    polee_ctx *ctx = ll->ctx;
    POLEE_TRY(use_device(ctx));
    if (!xs || !x_grad || K < 1 || K > PSELL_MAX_K)
        return fail(ctx, POLEE_ERR_BAD_ARG, "polee_loglik_eval: bad argument (K must be 1..8)");
    const size_t n = ll->n, tot = n * K;
    POLEE_TRY(ll->d_x_rows.upload(ctx, xs, tot));
    POLEE_TRY(ll->d_x_aos.alloc(ctx, tot));
    POLEE_TRY(ll->d_g_aos.alloc(ctx, tot));
    POLEE_TRY(ll->d_g_rows.alloc(ctx, tot));
    POLEE_TRY(ll->d_lp.alloc(ctx, PSELL_MAX_K));
    const unsigned nb = (unsigned)ceil_div(tot, 256);
    hipLaunchKernelGGL(rows_to_aos_kernel, dim3(nb), dim3(256), 0, ctx->stream, ll->d_x_rows.p, K, (int64_t)n,
                       ll->d_x_aos.p);
    POLEE_HIP_TRY(ctx, hipMemsetAsync(ll->d_g_aos.p, 0, tot * sizeof(float), ctx->stream));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(ll->d_lp.p, 0, PSELL_MAX_K * sizeof(double), ctx->stream));
    POLEE_TRY(loglik_eval_device(ll, ll->d_x_aos.p, K, ll->d_g_aos.p, lp ? ll->d_lp.p : nullptr));
    hipLaunchKernelGGL(aos_to_rows_f64_kernel, dim3(nb), dim3(256), 0, ctx->stream, ll->d_g_aos.p, K, (int64_t)n,
                       ll->d_g_rows.p);
    POLEE_KERNEL_CHECK(ctx);
    POLEE_TRY(ll->d_g_rows.download(ctx, x_grad, tot));
    if (lp) {
        POLEE_TRY(ll->d_lp.download(ctx, lp, K));
        for (int k = 0; k < K; ++k)
            if (!std::isfinite(lp[k]))
                return fail(ctx, POLEE_ERR_NONFINITE, "log-likelihood is not finite (likelihood.jl:50)");
    }
    return POLEE_OK;
}

polee_status polee_efflen_jacobian_adjustment(polee_ctx *ctx, const float *efflens, const float *xs, int32_t K,
                                              int64_t n, double *x_grad, float *xls)
{
    POLEE_TRY(use_device(ctx));
    if (!efflens || !xs || !x_grad || K < 1 || n < 1) return fail(ctx, POLEE_ERR_BAD_ARG, "bad argument");
    DevBuf<float> d_l, d_x, d_xls;
    DevBuf<double> d_g, d_s;
    const size_t tot = (size_t)n * K;
    POLEE_TRY(d_l.upload(ctx, efflens, n));
    POLEE_TRY(d_x.upload(ctx, xs, tot));
    POLEE_TRY(d_g.upload(ctx, x_grad, tot));
    POLEE_TRY(d_s.alloc(ctx, K));
    if (xls) POLEE_TRY(d_xls.alloc(ctx, tot));
    POLEE_HIP_TRY(ctx, hipMemsetAsync(d_s.p, 0, K * sizeof(double), ctx->stream));
    const unsigned nb = (unsigned)std::min<int64_t>(ceil_div(n, 256), 1024);
    hipLaunchKernelGGL(efflen_sum_kernel, dim3(nb, K), dim3(256), 0, ctx->stream, d_l.p, d_x.p, n, d_s.p);
    hipLaunchKernelGGL(efflen_adjust_kernel, dim3((unsigned)ceil_div(n, 256), K), dim3(256), 0, ctx->stream, d_l.p,
                       d_x.p, d_s.p, n, d_g.p, xls ? d_xls.p : nullptr);
    POLEE_KERNEL_CHECK(ctx);
    POLEE_TRY(d_g.download(ctx, x_grad, tot));
    if (xls) POLEE_TRY(d_xls.download(ctx, xls, tot));
    return POLEE_OK;
}

}  // extern "C"
