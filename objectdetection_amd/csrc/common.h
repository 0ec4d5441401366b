// common.h -- shared device helpers for liblcf (gfx950 only: wave64 hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lcf.h"

#define LCF_WAVE 64
#define LCF_EMPTY_KEY 0xFFFFFFFFu

#define LCF_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return LCF_ERR_LAUNCH; \
    } while (0)

// Device fill used instead of hipMemsetAsync for every internal reset: a plain kernel node
// behaves the same eagerly and under hipGraph capture / replay.  `bytes` must be a multiple of 4
// and `p` 4-byte aligned (every workspace region is 256-byte aligned).
__global__ void lcf_fill32_kernel(uint32_t *p, uint32_t v, size_t n32);
int lcf_fill32(void *p, uint32_t v, size_t bytes, hipStream_t st);

// ---- per-call options.  The caller owns an lcf_ctx (include/lcf.h) and passes it to the entry
// points that have options; nothing here is process-global except the four kernel-selection
// DEFAULTS that lcf_set_option keeps for benchmarks and tests (atomics, read once per call).
struct LcfOpts {
    int conv_split, conv_impl, conv_tile, conv_sets; // kernel selection (0 = choose)
    int rows_hint;                                   // expected rows of the next conv launches
    int ws_clean, nbr_clean;                         // caller-guaranteed buffer states
    int rb_unfused;                                  // profiling: lcf_rulebook_chain, one role per launch
    const int *plan;                                 // tile plan of the call in progress (not an option:
    int plan_tr;                                     // set by lcf_indice_conv_fwd_planned), its tile height
};
#define LCF_FILL_MAX_SEGS 16
struct LcfFillSeg {
    uint32_t *p;
    uint32_t v;
    unsigned blk0; // first workgroup of the segment
    unsigned long long n32;
};
// one optional "clear the rows the previous frame wrote" record per batch (lcf_dense_clear_rows):
// the workgroups from blk0 on zero dense[b, :, z, y, x] for every row of `prev` and then make
// `prev` this frame's coordinates
struct LcfClearSeg {
    float *dense;
    int *prev;
    const int *cur;
    const int *k_dev;
    int cap, c, batch, D, H, W;
    unsigned blk0; // first workgroup of the record; 0xFFFFFFFF = none
};
struct LcfFillPlan {
    LcfFillSeg s[LCF_FILL_MAX_SEGS];
    LcfClearSeg clr;
    int n;
};
struct lcf_ctx {
    LcfOpts o;
    LcfFillPlan plan; // fills recorded between lcf_ctx_fill_batch_begin and _flush
    int batching;
    unsigned fill_blocks;
};
// options of the call in progress on this thread (set by LcfCall for the duration of one entry
// point; host helpers deep in the launch code read them without having them threaded through)
extern thread_local const LcfOpts *t_lcf_opts;
LcfOpts lcf_default_opts(); // process-wide defaults (lcf_set_option)
struct LcfCall {
    LcfOpts eff;
    const LcfOpts *prev;
    explicit LcfCall(const lcf_ctx *c)
    {
        eff = lcf_default_opts();
        if (c) {
            if (c->o.conv_split) eff.conv_split = c->o.conv_split;
            if (c->o.conv_impl) eff.conv_impl = c->o.conv_impl;
            if (c->o.conv_tile) eff.conv_tile = c->o.conv_tile;
            if (c->o.conv_sets) eff.conv_sets = c->o.conv_sets;
            eff.rows_hint = c->o.rows_hint;
            eff.ws_clean = c->o.ws_clean;
            eff.nbr_clean = c->o.nbr_clean;
            eff.rb_unfused = c->o.rb_unfused;
        }
        prev = t_lcf_opts;
        t_lcf_opts = &eff;
    }
    ~LcfCall() { t_lcf_opts = prev; }
    LcfCall(const LcfCall &) = delete;
    LcfCall &operator=(const LcfCall &) = delete;
};
static inline LcfOpts lcf_opts() { return t_lcf_opts ? *t_lcf_opts : lcf_default_opts(); }
// a fill that joins the context's batch when one is open, else launches now
int lcf_fill32_ctx(lcf_ctx *ctx, void *p, uint32_t v, size_t bytes, hipStream_t st);

static inline size_t lcf_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline uint32_t lcf_pow2_ge(uint32_t x)
{
    uint32_t p = 1;
    while (p < x && p < 0x80000000u) p <<= 1; // clamps at 2^31: never loops on overflow
    return p;
}

// Row counts come as (capacity, optional device pointer): with a device pointer the kernels read
// the actual count from memory (clamped to the capacity) and the launch geometry depends on the
// capacity only, so no host synchronisation is needed to size a launch.
__device__ __forceinline__ int lcf_count(int cap, const int *dev)
{
    if (!dev) return cap;
    int v = *dev;
    v = v < 0 ? 0 : v;
    return v < cap ? v : cap;
}

// fp32 -> bf16 bits, round to nearest even (NaN stays NaN): what torch's float -> bfloat16 copy does
__device__ __forceinline__ unsigned lcf_bf16_rne(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// 32-bit integer hash (murmur3 finaliser)
__device__ __forceinline__ uint32_t lcf_hash32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

// Insert `key` into an open-addressing table of `mask+1` slots (keys initialised to
// LCF_EMPTY_KEY); returns the slot.  Concurrent inserts of equal keys return the same slot.
__device__ __forceinline__ uint32_t lcf_hash_insert(uint32_t *keys, uint32_t mask, uint32_t key)
{
    uint32_t h = lcf_hash32(key) & mask;
    while (true) {
        uint32_t old = atomicCAS(&keys[h], LCF_EMPTY_KEY, key);
        if (old == LCF_EMPTY_KEY || old == key) return h;
        h = (h + 1) & mask;
    }
}

// Find `key`; returns slot or 0xFFFFFFFF when absent.  Table must not be full.
__device__ __forceinline__ uint32_t lcf_hash_find(const uint32_t *keys, uint32_t mask,
                                                  uint32_t key)
{
    uint32_t h = lcf_hash32(key) & mask;
    while (true) {
        uint32_t cur = keys[h];
        if (cur == key) return h;
        if (cur == LCF_EMPTY_KEY) return 0xFFFFFFFFu;
        h = (h + 1) & mask;
    }
}

// Exclusive scan of one int per thread across a 256-thread block (4 waves).
// `lds` must hold >= 5 ints.  Returns the exclusive prefix; *total = block sum.
__device__ __forceinline__ int lcf_block_excl_scan_256(int v, int *lds, int *total)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        int s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}
