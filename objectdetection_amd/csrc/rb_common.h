// rb_common.h -- what rulebook.hip (one rulebook per call, spconv-API path) and
// rulebook_chain.hip (all rulebooks of a SparseSequential in fused launches) share.
#pragma once
#include "common.h"

#define RB_BLOCK 256
#define RB_ITEMS 4
#define RB_EMPTY64 0xFFFFFFFFFFFFFFFFull

struct ConvGeom {
    int shape[3];  // input spatial shape  (z,y,x)
    int oshape[3]; // output spatial shape
    int ks[3], st[3], pd[3], dl[3];
    int kvol;
    int E;        // max candidates per input = prod ceil(k/s)
    int ecs[3];   // per-dim candidate box bound ceil(k/s)
};

__device__ __forceinline__ uint32_t rb_key(int b, int z, int y, int x, const int *shape)
{
    return (((uint32_t)b * (uint32_t)shape[0] + (uint32_t)z) * (uint32_t)shape[1] + (uint32_t)y) *
               (uint32_t)shape[2] +
           (uint32_t)x;
}

// Candidate (j, e): e-th position of getValidOutPos's enumeration box for input row j.
// Returns false when e lies outside the box or the output position is out of range.
__device__ __forceinline__ bool rb_candidate(const int4 c, int e, const ConvGeom &g, int *oz,
                                             int *oy, int *ox, int *off)
{
    const int in[3] = {c.y, c.z, c.w};
    int lo[3], up[3], cs[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        lo[d] = (in[d] - (g.ks[d] - 1) * g.dl[d] - 1 + g.st[d] + g.pd[d]) / g.st[d]; // trunc div
        up[d] = (in[d] + g.pd[d]) / g.st[d];
        cs[d] = (up[d] - lo[d]) / g.dl[d] + 1;
    }
    if (cs[0] <= 0 || cs[1] <= 0 || cs[2] <= 0) return false;
    // e enumerates the (ecs0, ecs1, ecs2) bounding box, x fastest; positions outside this
    // input's own (cs0,cs1,cs2) box do not exist.  The order of the existing ones is unchanged.
    int c2 = e % g.ecs[2];
    int c1 = (e / g.ecs[2]) % g.ecs[1];
    int c0 = e / (g.ecs[2] * g.ecs[1]);
    if (c0 >= cs[0] || c1 >= cs[1] || c2 >= cs[2]) return false;
    int v[3] = {up[0] - c0 * g.dl[0], up[1] - c1 * g.dl[1], up[2] - c2 * g.dl[2]};
    int m = 1, o = 0;
    bool valid = true;
#pragma unroll
    for (int d = 2; d >= 0; --d) {
        if (v[d] < 0 || v[d] > g.oshape[d] - 1) valid = false;
        o += m * (in[d] - v[d] * g.st[d] + g.pd[d]) / g.dl[d];
        m *= g.ks[d];
    }
    *oz = v[0];
    *oy = v[1];
    *ox = v[2];
    *off = o;
    return valid;
}

__device__ __forceinline__ int rb_is_discoverer(const unsigned long long *tab,
                                                const int *cand_slot, long long t,
                                                long long total)
{
    if (t >= total) return 0;
    int h = cand_slot[t];
    if (h < 0) return 0;
    return (uint32_t)(tab[h] & 0xFFFFFFFFull) == (uint32_t)t;
}

static int rb_geom(ConvGeom *g, int batch, const int shape[3], const int ksize[3],
                   const int stride[3], const int pad[3], const int dil[3], bool subm)
{
    if (!shape || !ksize || !dil || batch <= 0) return LCF_ERR_ARG;
    long long kv = 1, E = 1;
    for (int d = 0; d < 3; ++d) {
        g->shape[d] = shape[d];
        g->ks[d] = ksize[d];
        g->dl[d] = dil[d];
        g->st[d] = subm ? 1 : stride[d];
        g->pd[d] = subm ? (ksize[d] / 2) * dil[d] : pad[d];
        if (shape[d] <= 0 || ksize[d] <= 0 || dil[d] <= 0 || g->st[d] <= 0 || g->pd[d] < 0)
            return LCF_ERR_ARG;
        if (g->st[d] != 1 && g->dl[d] != 1) return LCF_ERR_UNSUPPORTED; // as spconv asserts
        g->oshape[d] =
            subm ? shape[d]
                 : (shape[d] + 2 * g->pd[d] - g->dl[d] * (g->ks[d] - 1) - 1) / g->st[d] + 1;
        if (g->oshape[d] <= 0) return LCF_ERR_ARG;
        kv *= ksize[d];
        g->ecs[d] = (g->ks[d] + g->st[d] - 1) / g->st[d];
        E *= g->ecs[d];
    }
    if (kv > LCF_MAX_KVOL) return LCF_ERR_UNSUPPORTED;
    g->kvol = (int)kv;
    g->E = (int)E;
    unsigned long long cin = (unsigned long long)batch * shape[0] * shape[1] * shape[2];
    unsigned long long cout =
        (unsigned long long)batch * g->oshape[0] * g->oshape[1] * g->oshape[2];
    if (cin >= 0xFFFFFFFFull || cout >= 0xFFFFFFFFull) return LCF_ERR_RANGE;
    return LCF_OK;
}

static uint32_t rb_capacity(long long items)
{
    long long need = items * 2;
    if (need < 1024) need = 1024;
    if (need > 0x80000000ll) need = 0x80000000ll; // callers reject such sizes (LCF_ERR_RANGE)
    return lcf_pow2_ge((uint32_t)need);
}

