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
// LCF_OPT_WS_CLEAN: the caller hands in workspaces already in their reset state (misc.hip)
extern int g_lcf_ws_clean;
// LCF_OPT_NBR_CLEAN: lcf_rulebook_conv_fill's neighbour tables arrive filled with -1 (misc.hip)
extern int g_lcf_nbr_clean;

static inline size_t lcf_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline uint32_t lcf_pow2_ge(uint32_t x)
{
    uint32_t p = 1;
    while (p < x) p <<= 1;
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
