// rulebook_chain.hip -- ALL rulebooks of a SparseSequential (second/pytorch/models/middle.py:145-192:
// 4 submanifold keys + 4 strided convs for SpMiddleFHD) built by one call, in few, fused launches.
//
// Same algorithm and the same canonical result as rulebook.hip (coordinate hashing, 64-bit
// atomicMin on key<<32|stamp, discoverer flags + scan = spconv's CPU numbering), restructured for
// a frame that is bound by launch count and by what the branch costs the convolutions it runs
// beside:
//   * neighbour tables are written OUTPUT-STATIONARY: thread (output row n, offset o) computes the
//     input position and probes the previous level's coordinate table -- every entry is written
//     (row id or -1), so no table is pre-filled with -1 and nothing is scattered;
//   * a strided conv's own output table (key -> output id) IS the coordinate table of the next
//     level: its submanifold rulebook probes it directly, no second hash is built;
//   * independent steps share a launch ("roles" = block ranges of one kernel): the fill of level
//     l, the submanifold look-ups of level l and the candidate inserts of level l + 1 all depend
//     on the ids of level l only.
//   * the output table is sized by the OUTPUT capacity (2 x cap_out slots), not by the number of
//     candidates (2 x k_in x E): 8-16x smaller, L2-resident; a table that fills up can only do so
//     when the level overflows, which is reported; probing is bounded.
//   * a level whose tile plan is wanted (lcf_rb_level.plan: the per-tile set-up of the
//     tile-compacted convs, tc_plan.h) gets it from its look-up role: one workgroup per tile
//     computes the tile's table rows, keeps them in LDS and writes the plan record -- no plan
//     launch, no second read of the table (the stand-alone cv_plan_kernel remains for
//     lcf_conv_tile_plan and LCF_OPT_RB_UNFUSED).
// Launches for SpMiddleFHD: 2 + 3 per strided level = 14 (was 24 + 2 fills, + 6 plan launches).
// Measured and NOT kept (tools/bench_chain.py, per-step kernel times): several independent items
// per thread (fewer, longer waves: every step 30-60% slower -- a frame's tables are small enough
// that all waves are resident at once, so more waves beat more chains per wave); a relaxed
// agent-scope load in front of the CAS (inserts 9-16 -> 11-16 us); table entries rewritten as
// key<<32|id by the assign step so that look-ups need one load (look-ups -10%, assign +40%).
#include "rb_common.h"
#include "tc_plan.h"

#define RBC_MAX_ROLES 4
#define RBC_PROBE_LIMIT 2048u
#define RBC_NOFAIL 0xFFFFFFFFu

enum { RBC_INS32 = 1, RBC_INS64, RBC_COUNT, RBC_ASSIGN, RBC_LOOKUP, RBC_LOOKUP_PLAN };

struct RbcRole {
    int kind;
    unsigned blk0;
    // rows the role walks: coordinates of one level
    const int *coors;
    int k_cap;
    const int *k_dev;
    int batch;
    ConvGeom g;
    // level-0 table: 32-bit keys -> row
    uint32_t *keys32;
    int *vals32;
    uint32_t mask32;
    // a strided conv's output table: key<<32|stamp -> output id
    unsigned long long *tab;
    int *vals;
    uint32_t mask;
    uint32_t *fail; // RBC_NOFAIL until an insert ran out of probes
    int *cand_slot;
    int *bsums;
    int cap_out;
    int *out_coors, *k_out, *overflow;
    // LOOKUP: the table probed maps positions of shape g.shape to rows valid below this count
    int table64;
    int lim_cap;
    const int *lim_dev;
    int *nbr;
    // LOOKUP_PLAN: one workgroup per tile of plan_tr rows also writes the tile's plan record
    int *plan;
    int plan_tr, plan_stride;
};

struct RbcLaunch {
    RbcRole r[RBC_MAX_ROLES];
    int n;
};

__device__ __forceinline__ void rbc_ins32(const RbcRole &R, unsigned blk)
{
    const int k = lcf_count(R.k_cap, R.k_dev);
    int j = blk * RB_BLOCK + threadIdx.x;
    if (j >= k) return;
    int4 c = reinterpret_cast<const int4 *>(R.coors)[j];
    if (c.x < 0 || c.x >= R.batch || c.y < 0 || c.y >= R.g.shape[0] || c.z < 0 ||
        c.z >= R.g.shape[1] || c.w < 0 || c.w >= R.g.shape[2])
        return;
    uint32_t h = lcf_hash_insert(R.keys32, R.mask32, rb_key(c.x, c.y, c.z, c.w, R.g.shape));
    R.vals32[h] = j;
}

__device__ __forceinline__ void rbc_ins64(const RbcRole &R, unsigned blk)
{
    const int k_in = lcf_count(R.k_cap, R.k_dev);
    long long t = (long long)blk * RB_BLOCK + threadIdx.x;
    if (t >= (long long)k_in * R.g.E) return;
    int j = (int)(t / R.g.E);
    int e = (int)(t - (long long)j * R.g.E);
    int4 c = reinterpret_cast<const int4 *>(R.coors)[j];
    int oz, oy, ox, off;
    bool ok = c.x >= 0 && c.x < R.batch && rb_candidate(c, e, R.g, &oz, &oy, &ox, &off);
    if (!ok) {
        R.cand_slot[t] = -1;
        return;
    }
    uint32_t key = rb_key(c.x, oz, oy, ox, R.g.oshape);
    unsigned long long mine = ((unsigned long long)key << 32) | (unsigned long long)(uint32_t)t;
    uint32_t h = lcf_hash32(key) & R.mask;
    int slot = -1;
    for (uint32_t probe = 0; probe < RBC_PROBE_LIMIT; ++probe) {
        unsigned long long old = atomicCAS(&R.tab[h], RB_EMPTY64, mine);
        if (old == RB_EMPTY64) {
            slot = (int)h;
            break;
        }
        if ((uint32_t)(old >> 32) == key) {
            if (old > mine) atomicMin(&R.tab[h], mine);
            slot = (int)h;
            break;
        }
        h = (h + 1) & R.mask;
    }
    if (slot < 0) R.fail[0] = 0u; // table (sized for cap_out outputs) is full: level overflows
    R.cand_slot[t] = slot;
}

__device__ __forceinline__ void rbc_count(const RbcRole &R, unsigned blk, int *lds)
{
    const long long total = (long long)lcf_count(R.k_cap, R.k_dev) * R.g.E;
    long long base = ((long long)blk * RB_BLOCK + threadIdx.x) * RB_ITEMS;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < RB_ITEMS; ++i) cnt += rb_is_discoverer(R.tab, R.cand_slot, base + i, total);
    int tot;
    lcf_block_excl_scan_256(cnt, lds, &tot);
    if (threadIdx.x == 0) R.bsums[blk] = tot;
}

__device__ __forceinline__ void rbc_assign(const RbcRole &R, unsigned blk, unsigned nblk, int *lds)
{
    const long long total = (long long)lcf_count(R.k_cap, R.k_dev) * R.g.E;
    int part = 0;
    for (int b = threadIdx.x; b < (int)blk; b += RB_BLOCK) part += R.bsums[b];
    int blockbase;
    lcf_block_excl_scan_256(part, lds, &blockbase);
    long long base = ((long long)blk * RB_BLOCK + threadIdx.x) * RB_ITEMS;
    int fl[RB_ITEMS];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < RB_ITEMS; ++i) {
        fl[i] = rb_is_discoverer(R.tab, R.cand_slot, base + i, total);
        cnt += fl[i];
    }
    int tot;
    int ex = lcf_block_excl_scan_256(cnt, lds, &tot) + blockbase;
#pragma unroll
    for (int i = 0; i < RB_ITEMS; ++i) {
        if (fl[i]) {
            int h = R.cand_slot[base + i];
            R.vals[h] = ex;
            if (ex < R.cap_out) {
                uint32_t key = (uint32_t)(R.tab[h] >> 32);
                uint32_t W = R.g.oshape[2], H = R.g.oshape[1], D = R.g.oshape[0];
                int4 oc;
                oc.w = (int)(key % W);
                oc.z = (int)((key / W) % H);
                oc.y = (int)((key / (W * H)) % D);
                oc.x = (int)(key / (W * H * D));
                reinterpret_cast<int4 *>(R.out_coors)[ex] = oc;
            }
            ++ex;
        }
    }
    if (blk == nblk - 1 && threadIdx.x == 0) {
        const int n = blockbase + tot;
        R.k_out[0] = n < R.cap_out ? n : R.cap_out; // outputs past the capacity are dropped ...
        // ... and reported (written 0 or 1 every time: the caller does not clear it)
        if (R.overflow) R.overflow[0] = (n > R.cap_out || R.fail[0] != RBC_NOFAIL) ? 1 : 0;
    }
}

// (row n, offset o): the row of the previous level that feeds output n through offset o
// (in = out * stride - pad + k * dil; a submanifold conv is stride 1, pad (k/2) * dil), or -1
__device__ __forceinline__ int rbc_lookup_one(const RbcRole &R, int n, int o)
{
    int4 c = reinterpret_cast<const int4 *>(R.coors)[n];
    int kx = o % R.g.ks[2];
    int ky = (o / R.g.ks[2]) % R.g.ks[1];
    int kz = o / (R.g.ks[2] * R.g.ks[1]);
    int z = c.y * R.g.st[0] - R.g.pd[0] + kz * R.g.dl[0];
    int y = c.z * R.g.st[1] - R.g.pd[1] + ky * R.g.dl[1];
    int x = c.w * R.g.st[2] - R.g.pd[2] + kx * R.g.dl[2];
    int res = -1;
    if (c.x >= 0 && c.x < R.batch && z >= 0 && z < R.g.shape[0] && y >= 0 && y < R.g.shape[1] &&
        x >= 0 && x < R.g.shape[2]) {
        const uint32_t key = rb_key(c.x, z, y, x, R.g.shape);
        if (R.table64) {
            uint32_t h = lcf_hash32(key) & R.mask;
            for (uint32_t probe = 0; probe < RBC_PROBE_LIMIT; ++probe) {
                const unsigned long long cur = R.tab[h];
                if ((uint32_t)(cur >> 32) == key) {
                    res = R.vals[h];
                    break;
                }
                if (cur == RB_EMPTY64) break;
                h = (h + 1) & R.mask;
            }
            if (res >= lcf_count(R.lim_cap, R.lim_dev)) res = -1; // an output past the capacity
        } else {
            uint32_t h = lcf_hash_find(R.keys32, R.mask32, key);
            if (h != 0xFFFFFFFFu) res = R.vals32[h];
        }
    }
    return res;
}

__device__ __forceinline__ void rbc_lookup(const RbcRole &R, unsigned blk)
{
    const int k = lcf_count(R.k_cap, R.k_dev);
    long long t = (long long)blk * RB_BLOCK + threadIdx.x;
    if (t >= (long long)k * R.g.kvol) return;
    int n = (int)(t / R.g.kvol);
    int o = (int)(t - (long long)n * R.g.kvol);
    R.nbr[t] = rbc_lookup_one(R, n, o);
}

// The same table rows, one workgroup per TILE of plan_tr rows of the conv kernels, which then
// builds the tile's plan record (tc_plan.h) from the neighbours it has just computed and still
// holds in LDS: the plan costs no launch and no second read of the table.
__device__ __forceinline__ void rbc_lookup_plan(const RbcRole &R, unsigned blk, unsigned char *smem)
{
    const int kvol = R.g.kvol, tr = R.plan_tr;
    int *nl = reinterpret_cast<int *>(smem);                   // [32][kvol]
    unsigned *ent = reinterpret_cast<unsigned *>(nl + ((32 * kvol + 3) & ~3));
    int *cnt = reinterpret_cast<int *>(ent + 32 * kvol);
    int *gbase = cnt + 64;
    int *soff = gbase + 64;
    int *nsteps_p = soff + 64;
    const int k = lcf_count(R.k_cap, R.k_dev);
    const int row0 = (int)blk * tr;
    if (row0 >= k) return; // the conv kernels leave such a tile before reading its record
    const int total = tr * kvol;
    for (int e = threadIdx.x; e < total; e += RB_BLOCK) {
        const int n = row0 + e / kvol;
        int res = -1;
        if (n < k) {
            res = rbc_lookup_one(R, n, e % kvol);
            R.nbr[(long long)row0 * kvol + e] = res;
        }
        nl[e] = res;
    }
    for (int e = threadIdx.x; e < 32 * kvol; e += RB_BLOCK) ent[e] = 0u; // unused entries: defined
    if (threadIdx.x < 64) {
        cnt[threadIdx.x] = 0;
        gbase[threadIdx.x] = 0;
        soff[threadIdx.x] = 0;
        if (threadIdx.x < 4) nsteps_p[threadIdx.x] = 0; // the count and the record's padding
    }
    __syncthreads();
    tc_build_steps_lds<4, RB_BLOCK>(tr, kvol, nl, ent, cnt, gbase, soff, nsteps_p, 32);
    int4 *dst = reinterpret_cast<int4 *>(R.plan + (size_t)blk * R.plan_stride);
    const int4 *src = reinterpret_cast<const int4 *>(ent);
    for (int e = threadIdx.x; e < R.plan_stride / 4; e += RB_BLOCK) dst[e] = src[e];
}

// TAG only names the launch for profilers: 0 = level-0 table + first inserts, 1 = count (+ level-0
// look-ups), 2 = assign, 3 = fill + look-ups + next inserts; 10 + kind = one role alone
// (LCF_OPT_RB_UNFUSED, profiling).
template <int TAG> __global__ void __launch_bounds__(RB_BLOCK) rbc_kernel(const RbcLaunch L)
{
    __shared__ int lds[8];
    extern __shared__ __align__(16) unsigned char rbc_smem[]; // plan-building roles only
    int i = 0;
#pragma unroll
    for (int j = 1; j < RBC_MAX_ROLES; ++j)
        if (j < L.n && blockIdx.x >= L.r[j].blk0) i = j;
    const RbcRole &R = L.r[i];
    const unsigned blk = blockIdx.x - R.blk0;
    const unsigned nblk = (i + 1 < L.n ? L.r[i + 1].blk0 : gridDim.x) - R.blk0;
    switch (R.kind) { // uniform per workgroup
    case RBC_INS32: rbc_ins32(R, blk); break;
    case RBC_INS64: rbc_ins64(R, blk); break;
    case RBC_COUNT: rbc_count(R, blk, lds); break;
    case RBC_ASSIGN: rbc_assign(R, blk, nblk, lds); break;
    case RBC_LOOKUP: rbc_lookup(R, blk); break;
    case RBC_LOOKUP_PLAN: rbc_lookup_plan(R, blk, rbc_smem); break;
    default: break;
    }
}

// ---------------------------------------------------------------- host side

struct RbcWs { // workspace of one strided level
    unsigned long long *tab;
    uint32_t *fail;
    int *vals, *cand_slot, *bsums;
    uint32_t hc;
    int nblk;
    long long total;
    size_t bytes, arm_bytes; // arm_bytes: the prefix a reset fills with 0xFF (table + fail word)
};

static long long rbc_E(const int ksize[3], const int stride[3])
{
    long long e = 1;
    for (int d = 0; d < 3; ++d) {
        if (ksize[d] <= 0 || stride[d] <= 0) return -1;
        e *= (ksize[d] + stride[d] - 1) / stride[d];
    }
    return e;
}

static RbcWs rbc_carve(void *ws, int k_in, long long E, int cap_out)
{
    RbcWs w;
    w.total = (long long)k_in * E;
    long long outs = w.total < cap_out ? w.total : cap_out;
    w.hc = rb_capacity(outs);
    w.nblk = (int)((w.total + RB_BLOCK * RB_ITEMS - 1) / (RB_BLOCK * RB_ITEMS));
    if (w.nblk < 1) w.nblk = 1;
    char *p = (char *)ws;
    size_t off = 0;
    w.tab = (unsigned long long *)(p + off);
    off += lcf_align_up((size_t)w.hc * 8, 256);
    w.fail = (uint32_t *)(p + off);
    off += 256;
    w.arm_bytes = off;
    w.vals = (int *)(p + off);
    off += lcf_align_up((size_t)w.hc * 4, 256);
    w.cand_slot = (int *)(p + off);
    off += lcf_align_up((size_t)(w.total + RB_ITEMS) * 4, 256);
    w.bsums = (int *)(p + off);
    off += lcf_align_up((size_t)w.nblk * 4, 256);
    w.bytes = off;
    return w;
}

extern "C" size_t lcf_rulebook_chain_workspace_bytes(int k_in, const int ksize[3],
                                                     const int stride[3], int cap_out)
{
    if (k_in < 0 || cap_out < 0 || !ksize || !stride) return 0;
    long long E = rbc_E(ksize, stride);
    if (E < 0 || (long long)k_in * E >= 0x3FFFFFFFll) return 0;
    return rbc_carve(nullptr, k_in, E, cap_out).bytes;
}

struct RbcPlanLevel {
    ConvGeom g;
    RbcWs w;
    int k_in_cap; // rows of the level this conv reads
};

static unsigned rbc_blocks(long long items)
{
    long long nb = (items + RB_BLOCK - 1) / RB_BLOCK;
    return nb < 1 ? 1u : (unsigned)nb;
}

template <int TAG> static int rbc_launch_tag(const RbcLaunch &L, unsigned blocks, hipStream_t st)
{
    size_t lds = 0; // dynamic LDS: what the launch's plan-building roles need
    for (int i = 0; i < L.n; ++i)
        if (L.r[i].kind == RBC_LOOKUP_PLAN) {
            const size_t b = tc_plan_lds_bytes(L.r[i].g.kvol);
            lds = b > lds ? b : lds;
        }
    rbc_kernel<TAG><<<blocks, RB_BLOCK, lds, st>>>(L);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

static int rbc_launch(RbcLaunch &L, unsigned blocks, hipStream_t st, int tag, bool unfused)
{
    if (L.n == 0) return LCF_OK;
    if (unfused) { // profiling: every role as a launch of its own, named by its kind
        for (int i = 0; i < L.n; ++i) {
            RbcLaunch one{};
            one.n = 1;
            one.r[0] = L.r[i];
            const unsigned nb = (i + 1 < L.n ? L.r[i + 1].blk0 : blocks) - L.r[i].blk0;
            one.r[0].blk0 = 0;
            int rc;
            switch (one.r[0].kind) {
            case RBC_INS32: rc = rbc_launch_tag<11>(one, nb, st); break;
            case RBC_INS64: rc = rbc_launch_tag<12>(one, nb, st); break;
            case RBC_COUNT: rc = rbc_launch_tag<13>(one, nb, st); break;
            case RBC_ASSIGN: rc = rbc_launch_tag<14>(one, nb, st); break;
            case RBC_LOOKUP_PLAN: rc = rbc_launch_tag<16>(one, nb, st); break;
            default: rc = rbc_launch_tag<15>(one, nb, st); break;
            }
            if (rc) return rc;
        }
        return LCF_OK;
    }
    switch (tag) {
    case 0: return rbc_launch_tag<0>(L, blocks, st);
    case 1: return rbc_launch_tag<1>(L, blocks, st);
    case 2: return rbc_launch_tag<2>(L, blocks, st);
    default: return rbc_launch_tag<3>(L, blocks, st);
    }
}

// state of the level the walk stands on
struct RbcCur {
    const int *coors;
    int k_cap;
    const int *k_dev;
    int shape[3];
    bool t64; // its coordinate table: the producing conv's output table, or the level-0 table
    const RbcPlanLevel *prod;
};

static void rbc_role_common(RbcRole &R, const RbcCur &cur, int batch)
{
    R.coors = cur.coors;
    R.k_cap = cur.k_cap;
    R.k_dev = cur.k_dev;
    R.batch = batch;
}

extern "C" int lcf_rulebook_chain_reset(const lcf_rb_level *lv, int n, int k, void *ws0,
                                        size_t ws0_bytes, void *stream, lcf_ctx *ctx)
{
    if (!lv || n <= 0 || k < 0 || !ws0) return LCF_ERR_ARG;
    if (ws0_bytes < lcf_rulebook_subm_workspace_bytes(k)) return LCF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int rc = lcf_fill32_ctx(ctx, ws0, 0xFFFFFFFFu, (size_t)rb_capacity(k) * 4, st);
    if (rc != LCF_OK) return rc;
    int cap = k;
    for (int i = 0; i < n; ++i) {
        if (lv[i].subm) continue;
        long long E = rbc_E(lv[i].ksize, lv[i].stride);
        if (E < 0 || !lv[i].ws) return LCF_ERR_ARG;
        if ((long long)cap * E >= 0x3FFFFFFFll) return LCF_ERR_RANGE;
        RbcWs w = rbc_carve(lv[i].ws, cap, E, lv[i].cap_out);
        if (lv[i].ws_bytes < w.bytes) return LCF_ERR_WORKSPACE;
        rc = lcf_fill32_ctx(ctx, w.tab, 0xFFFFFFFFu, w.arm_bytes, st);
        if (rc != LCF_OK) return rc;
        cap = lv[i].cap_out;
    }
    return LCF_OK;
}

// Parts of the walk: part 0 = level-0 table, the first strided conv's inserts and count, the
// submanifold look-ups of the input resolution; part q >= 1 = ids, fill and look-ups of the q-th
// strided level, the next one's inserts and count.  Part q needs parts 0 .. q-1 enqueued before it
// on the same stream.
extern "C" int lcf_rulebook_chain_num_parts(const lcf_rb_level *lv, int n)
{
    if (!lv || n <= 0) return 0;
    int parts = 1;
    for (int i = 0; i < n; ++i) parts += lv[i].subm ? 0 : 1;
    return parts;
}

extern "C" int lcf_rulebook_chain_parts(const int *coors, int k, const int *k_dev, int batch,
                                        const int shape[3], const lcf_rb_level *lv, int n,
                                        void *ws0, size_t ws0_bytes, int part_first,
                                        int part_last, void *stream, lcf_ctx *ctx)
{
    if (!lv || n <= 0 || n > 64 || k < 0 || batch <= 0 || !shape || !ws0 || (k > 0 && !coors) ||
        part_first < 0 || part_last < part_first)
        return LCF_ERR_ARG;
    if (k >= (1 << 25)) return LCF_ERR_RANGE;
    int part = 0; // the part the walk stands in; launches and records outside the range are skipped
    auto on = [&]() { return part >= part_first && part <= part_last; };
    if (ws0_bytes < lcf_rulebook_subm_workspace_bytes(k)) return LCF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const bool unfused = ctx && ctx->o.rb_unfused;
    // strided levels (from the first) whose own table is launched ahead of the rest of launch X:
    // all of them (in the frame 0.627 -> 0.616 ms with the first, 0.613-0.616 with all, A/B in one
    // call; LCF_RB_SPLIT_FILL=0 in the environment restores the single launch)
    static const int split_fill =
        getenv("LCF_RB_SPLIT_FILL") ? atoi(getenv("LCF_RB_SPLIT_FILL")) : 64;

    // ---- plan: geometry and workspace of every level, validated before anything is enqueued
    RbcPlanLevel plan[64];
    {
        int cap = k;
        int sh[3] = {shape[0], shape[1], shape[2]};
        for (int i = 0; i < n; ++i) {
            const lcf_rb_level &l = lv[i];
            RbcPlanLevel &p = plan[i];
            int rc = rb_geom(&p.g, batch, sh, l.ksize, l.stride, l.pad, l.dil, l.subm != 0);
            if (rc) return rc;
            p.k_in_cap = cap;
            if (cap > 0 && !l.out_nbr) return LCF_ERR_ARG;
            if (l.plan && (l.plan_tile_rows < 1 || l.plan_tile_rows > 32 ||
                           ((uintptr_t)l.plan & 15)))
                return LCF_ERR_ARG;
            if ((long long)cap * p.g.kvol >= 0x7FFFFFFFll) return LCF_ERR_RANGE;
            if (l.subm) continue;
            if (!l.ws || !l.k_out || !l.overflow || l.cap_out <= 0 || !l.out_coors)
                return LCF_ERR_ARG;
            if ((long long)cap * p.g.E >= 0x3FFFFFFFll || l.cap_out >= (1 << 25))
                return LCF_ERR_RANGE;
            if ((long long)l.cap_out * p.g.kvol >= 0x7FFFFFFFll) return LCF_ERR_RANGE;
            p.w = rbc_carve(l.ws, cap, p.g.E, l.cap_out);
            if (l.ws_bytes < p.w.bytes) return LCF_ERR_WORKSPACE;
            cap = l.cap_out;
            for (int d = 0; d < 3; ++d) sh[d] = p.g.oshape[d];
        }
    }
    const uint32_t hc0 = rb_capacity(k);
    uint32_t *keys0 = (uint32_t *)ws0;
    int *vals0 = (int *)((char *)ws0 + lcf_align_up((size_t)hc0 * 4, 256));
    if (!(ctx && ctx->o.ws_clean) && part_first == 0) {
        int rc = lcf_rulebook_chain_reset(lv, n, k, ws0, ws0_bytes, stream, nullptr);
        if (rc != LCF_OK) return rc;
    }

    RbcCur cur;
    cur.coors = coors;
    cur.k_cap = k;
    cur.k_dev = k_dev;
    for (int d = 0; d < 3; ++d) cur.shape[d] = shape[d];
    cur.t64 = false;
    cur.prod = nullptr;

    // a level whose tile plan is wanted gets it from its look-up role (unless LCF_OPT_RB_UNFUSED
    // asks for the stand-alone kernels)
    auto add_lookup = [&](RbcLaunch &L, unsigned &blocks, const RbcCur &rows, const RbcCur &table,
                          const RbcPlanLevel &p, int *nbr, bool strided, const lcf_rb_level &lvl) {
        // rows: the level whose rows own the table being written; table: the level probed
        RbcRole &R = L.r[L.n++];
        R = RbcRole{};
        R.kind = RBC_LOOKUP;
        R.blk0 = blocks;
        rbc_role_common(R, rows, batch);
        R.g = p.g; // strided: g.shape = input (probed) shape; subm: shape = oshape
        if (!strided) {
            for (int d = 0; d < 3; ++d) R.g.st[d] = 1;
        }
        R.table64 = table.t64 ? 1 : 0;
        if (table.t64) {
            R.tab = table.prod->w.tab;
            R.vals = table.prod->w.vals;
            R.mask = table.prod->w.hc - 1;
            R.lim_cap = table.k_cap;
            R.lim_dev = table.k_dev;
        } else {
            R.keys32 = keys0;
            R.vals32 = vals0;
            R.mask32 = hc0 - 1;
        }
        R.nbr = nbr;
        if (lvl.plan && !unfused) {
            R.kind = RBC_LOOKUP_PLAN;
            R.plan = lvl.plan;
            R.plan_tr = lvl.plan_tile_rows;
            R.plan_stride = tc_plan_stride(p.g.kvol);
            const unsigned tiles = (unsigned)((rows.k_cap + R.plan_tr - 1) / R.plan_tr);
            blocks += tiles < 1 ? 1u : tiles;
        } else {
            blocks += rbc_blocks((long long)rows.k_cap * p.g.kvol);
        }
    };
    auto add_ins64 = [&](RbcLaunch &L, unsigned &blocks, const RbcCur &rows,
                         const RbcPlanLevel &p) {
        RbcRole &R = L.r[L.n++];
        R = RbcRole{};
        R.kind = RBC_INS64;
        R.blk0 = blocks;
        rbc_role_common(R, rows, batch);
        R.g = p.g;
        R.tab = p.w.tab;
        R.mask = p.w.hc - 1;
        R.fail = p.w.fail;
        R.cand_slot = p.w.cand_slot;
        blocks += rbc_blocks(p.w.total);
    };
    // rows of the level a table belongs to: (capacity, device count)
    int rows_cap[64];
    const int *rows_dev[64];
    {
        int cap = k;
        const int *dev = k_dev;
        for (int q = 0; q < n; ++q) {
            if (!lv[q].subm) {
                cap = lv[q].cap_out;
                dev = lv[q].k_out;
            }
            rows_cap[q] = cap;
            rows_dev[q] = dev;
        }
    }
    auto record = [&](int i) -> int {
        if (!on()) return LCF_OK;
        if (lv[i].plan && unfused) { // stand-alone plan kernel (fused into the look-ups otherwise)
            int rc = lcf_conv_tile_plan(lv[i].out_nbr, rows_cap[i], rows_dev[i], plan[i].g.kvol,
                                        lv[i].plan_tile_rows, lv[i].plan, stream);
            if (rc) return rc;
        }
        if (lv[i].ready_event &&
            hipEventRecord((hipEvent_t)lv[i].ready_event, st) != hipSuccess)
            return LCF_ERR_LAUNCH;
        return LCF_OK;
    };
    if (k == 0) { // nothing to build: every strided level is empty
        if (part_first > 0) return LCF_OK; // all of it was part 0
        for (int i = 0; i < n; ++i) {
            if (!lv[i].subm) {
                if (lcf_fill32(lv[i].k_out, 0u, 4, st) != LCF_OK) return LCF_ERR_LAUNCH;
                if (lcf_fill32(lv[i].overflow, 0u, 4, st) != LCF_OK) return LCF_ERR_LAUNCH;
            }
            int rc = record(i);
            if (rc) return rc;
        }
        return LCF_OK;
    }

    // ---- walk.  `i` = first level not yet handled at the current resolution.
    int i = 0;
    // next strided level at or after i (n = none)
    auto next_strided = [&](int from) {
        int s = from;
        while (s < n && lv[s].subm) ++s;
        return s;
    };
    int s = next_strided(0);
    {
        // launch A: level-0 table + the first strided conv's candidate inserts
        RbcLaunch L{};
        unsigned blocks = 0;
        RbcRole &R = L.r[L.n++];
        R = RbcRole{};
        R.kind = RBC_INS32;
        R.blk0 = 0;
        rbc_role_common(R, cur, batch);
        for (int d = 0; d < 3; ++d) R.g.shape[d] = cur.shape[d];
        R.keys32 = keys0;
        R.vals32 = vals0;
        R.mask32 = hc0 - 1;
        blocks += rbc_blocks(k);
        if (s < n) add_ins64(L, blocks, cur, plan[s]);
        int rc = on() ? rbc_launch(L, blocks, st, 0, unfused) : LCF_OK;
        if (rc) return rc;
    }
    while (true) {
        // launch B: the strided conv's count + the submanifold look-ups of this resolution that
        // do not need anything newer than the current level's table.  At level 0 the table was
        // built by launch A; at level >= 1 the look-ups were already issued with the fill.
        {
            RbcLaunch L{};
            unsigned blocks = 0;
            if (s < n) {
                RbcRole &R = L.r[L.n++];
                R = RbcRole{};
                R.kind = RBC_COUNT;
                R.blk0 = blocks;
                rbc_role_common(R, cur, batch);
                R.g = plan[s].g;
                R.tab = plan[s].w.tab;
                R.cand_slot = plan[s].w.cand_slot;
                R.bsums = plan[s].w.bsums;
                blocks += (unsigned)plan[s].w.nblk;
            }
            int first_unrecorded = i;
            if (!cur.t64 || true) {
                // submanifold levels i .. s-1 not yet emitted (level 0, or more than fitted into
                // the fill launch)
                while (i < s) {
                    if (L.n == RBC_MAX_ROLES) {
                        int rc = on() ? rbc_launch(L, blocks, st, 1, unfused) : LCF_OK;
                        if (rc) return rc;
                        L = RbcLaunch{};
                        blocks = 0;
                    }
                    add_lookup(L, blocks, cur, cur, plan[i], lv[i].out_nbr, false, lv[i]);
                    ++i;
                }
            }
            int rc = on() ? rbc_launch(L, blocks, st, 1, unfused) : LCF_OK;
            if (rc) return rc;
            for (int j = first_unrecorded; j < s; ++j) {
                rc = record(j);
                if (rc) return rc;
            }
        }
        if (s >= n) break;
        ++part;
        if (part > part_last) break;
        // launch C: ids of the strided level (needs every block's count)
        {
            RbcLaunch L{};
            RbcRole &R = L.r[L.n++];
            R = RbcRole{};
            R.kind = RBC_ASSIGN;
            R.blk0 = 0;
            rbc_role_common(R, cur, batch);
            R.g = plan[s].g;
            R.tab = plan[s].w.tab;
            R.vals = plan[s].w.vals;
            R.fail = plan[s].w.fail;
            R.cand_slot = plan[s].w.cand_slot;
            R.bsums = plan[s].w.bsums;
            R.cap_out = lv[s].cap_out;
            R.out_coors = lv[s].out_coors;
            R.k_out = lv[s].k_out;
            R.overflow = lv[s].overflow;
            int rc = on() ? rbc_launch(L, (unsigned)plan[s].w.nblk, st, 2, unfused) : LCF_OK;
            if (rc) return rc;
        }
        // launch X: fill of the strided level (probes the level it read), submanifold look-ups of
        // the new level (probe its own table) and the next strided conv's inserts
        RbcCur nxt;
        nxt.coors = lv[s].out_coors;
        nxt.k_cap = lv[s].cap_out;
        nxt.k_dev = lv[s].k_out;
        for (int d = 0; d < 3; ++d) nxt.shape[d] = plan[s].g.oshape[d];
        nxt.t64 = true;
        nxt.prod = &plan[s];
        const int s2 = next_strided(s + 1);
        {
            RbcLaunch L{};
            unsigned blocks = 0;
            add_lookup(L, blocks, nxt, cur, plan[s], lv[s].out_nbr, true, lv[s]);
            // The strided conv's own table is what the feature path needs first: for the leading
            // levels, where that path runs right behind the chain, it goes out as a launch of its
            // own (and its event is recorded) ahead of the new level's look-ups.
            bool fill_alone = part - 1 < split_fill;
            if (fill_alone) {
                int rc = on() ? rbc_launch(L, blocks, st, 3, unfused) : LCF_OK;
                if (rc) return rc;
                rc = record(s);
                if (rc) return rc;
                L = RbcLaunch{};
                blocks = 0;
            }
            int j = s + 1;
            while (j < s2 && L.n < RBC_MAX_ROLES - (s2 < n ? 1 : 0)) {
                add_lookup(L, blocks, nxt, nxt, plan[j], lv[j].out_nbr, false, lv[j]);
                ++j;
            }
            if (s2 < n) add_ins64(L, blocks, nxt, plan[s2]);
            int rc = on() ? rbc_launch(L, blocks, st, 3, unfused) : LCF_OK;
            if (rc) return rc;
            for (int q = fill_alone ? s + 1 : s; q < j; ++q) {
                rc = record(q);
                if (rc) return rc;
            }
            i = j; // submanifold levels j .. s2-1 (if any) go into the next launch B
        }
        cur = nxt;
        s = s2;
    }
    return LCF_OK;
}

extern "C" int lcf_rulebook_chain(const int *coors, int k, const int *k_dev, int batch,
                                  const int shape[3], const lcf_rb_level *lv, int n, void *ws0,
                                  size_t ws0_bytes, void *stream, lcf_ctx *ctx)
{
    return lcf_rulebook_chain_parts(coors, k, k_dev, batch, shape, lv, n, ws0, ws0_bytes, 0,
                                    0x7FFFFFFF, stream, ctx);
}
