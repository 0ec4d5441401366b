// tc_plan.h -- the per-tile set-up of the tile-compacted convolutions (conv.hip: cv_tc_kernel) and
// the layout of a tile plan record, shared by conv.hip (per launch, or cv_plan_kernel once per
// rulebook) and rulebook_chain.hip (built in the launch that computes the tile's neighbours).
#pragma once
#include "common.h"

// One plan record per tile = the image of the conv kernel's LDS block: ent [32 * kvol] | cnt [64] |
// gbase [64] | soff [64] | nsteps, padded to 16 bytes (ints).
static inline int tc_plan_stride(int kvol) { return (32 * kvol + 193 + 3) & ~3; }
// LDS bytes a plan builder needs: nl [32 * kvol] (4-aligned) + one record + slack
static inline size_t tc_plan_lds_bytes(int kvol)
{
    return (size_t)(((32 * kvol + 3) & ~3) + tc_plan_stride(kvol) + 4) * 4;
}

// The set-up of one tile of the tile-compacted kernels: neighbour tile -> LDS, rows per offset,
// steps (= active offsets ascending), and the gather entries of every 16-row group.  Leaves, in
// LDS: ent [items][16] ((input row << 7) | tile row; a group's padding entries gather a valid row
// and point at the dump row), cnt / gbase [64] per offset, soff [64] per step, *nsteps_p.
// Used by cv_tc_kernel (per launch) and by cv_plan_kernel (once per rulebook).
// tc_build_steps_lds: the same from a neighbour tile nl [tr][kvol] that is already in LDS (the
// rulebook chain builds the plan in the launch that computes the tile's neighbours).
template <int NWT, int TT>
__device__ __forceinline__ void tc_build_steps_lds(int tr, int kvol, int *nl, unsigned *ent,
                                                   int *cnt, int *gbase, int *soff,
                                                   int *nsteps_p, int DUMP)
{
    const int tidw = threadIdx.x;
    const int lane = tidw & 63;
    const int wavew = __builtin_amdgcn_readfirstlane(tidw >> 6);
    // ---- rows per offset.  A wave takes offsets wavew, wavew + NWT, ...: the (up to MAXO) neighbour
    //      ids of its lane's row are read once and kept, with their ballots, for the entry pass
    //      below (the reads are independent: one LDS round trip, not one per offset)
    constexpr int MAXO = (LCF_MAX_KVOL + NWT - 1) / NWT;
    int idv[MAXO];
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
        const int o = wavew + i * NWT;
        idv[i] = (o < kvol && lane < tr) ? nl[lane * kvol + o] : -1;
    }
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
        const int o = wavew + i * NWT;
        if (o < kvol) { // wave-uniform
            const unsigned long long m = __ballot(idv[i] >= 0);
            if (lane == 0) cnt[o] = __popcll(m);
        }
    }
    __syncthreads();
    // ---- steps = active offsets ascending; items = their 16-row groups
    if (wavew == 0) {
        const int c = lane < kvol ? cnt[lane] : 0;
        const int ng = (c + 15) >> 4;
        int incl = ng, sincl = c > 0 ? 1 : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int t = __shfl_up(incl, d, 64);
            int u = __shfl_up(sincl, d, 64);
            if (lane >= d) {
                incl += t;
                sincl += u;
            }
        }
        gbase[lane] = incl - ng;
        if (c > 0) soff[sincl - 1] = lane;
        if (lane == 63) *nsteps_p = sincl;
    }
    __syncthreads();
    // ---- item entries: (input row << 7 | output row of the tile); a group's padding entries
    //      gather a valid row and dump the result
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
        const int o = wavew + i * NWT;
        if (o >= kvol) continue; // wave-uniform
        const int id = idv[i];
        const bool has = id >= 0;
        const unsigned long long m = __ballot(has);
        if (m == 0ull) continue;
        const int c = __popcll(m);
        const int p = __popcll(m & ((1ull << lane) - 1ull));
        unsigned *dst = ent + gbase[o] * 16;
        if (has) dst[p] = ((unsigned)id << 7) | (unsigned)lane;
        const int first = __shfl(id, __builtin_ctzll(m), 64);
        const int padded = ((c + 15) >> 4) << 4;
        if (c + lane < padded) dst[c + lane] = ((unsigned)first << 7) | (unsigned)DUMP;
    }
    __syncthreads();
}


template <int NWT, int TT>
__device__ __forceinline__ void tc_build_steps(const int *__restrict__ nbr, int row0, int tr,
                                               int kvol, int k_out, int *nl, unsigned *ent,
                                               int *cnt, int *gbase, int *soff, int *nsteps_p,
                                               int DUMP)
{
    // ---- neighbour tile -> LDS (coalesced; rows past k_out read as -1)
    {
        const int total = tr * kvol;
        const long long gbase_e = (long long)row0 * kvol;
        const long long glimit = (long long)k_out * kvol;
        for (int e = threadIdx.x; e < total; e += TT) {
            int v = -1;
            if (gbase_e + e < glimit) v = nbr ? nbr[gbase_e + e] : row0 + e; // NULL: identity
            nl[e] = v;
        }
    }
    __syncthreads();
    tc_build_steps_lds<NWT, TT>(tr, kvol, nl, ent, cnt, gbase, soff, nsteps_p, DUMP);
}
