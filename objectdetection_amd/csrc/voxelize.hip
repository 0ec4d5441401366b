// voxelize.hip -- hash-based point-to-voxel scatter, bit-identical to the sequential first-come
// loop of the reference (second/utils/simplevis.py:23-50 twin of points_to_voxel;
// call contract second/data/preprocess.py:626-639).
//
// Determinism recipe (SURVEY.md section 7 "order determinism"):
//   1. insert : every in-range point claims its cell in an open-addressing hash (CAS on the
//               32-bit linear cell key) and pushes its index through the cell's T-slot sorted
//               list with a chain of atomicMin's (slot s keeps the min, the loser moves on to
//               slot s+1).  Whatever the interleaving, slot s ends up holding the (s+1)-th
//               smallest point index of the cell, i.e. exactly the first T points the sequential
//               loop would have kept, in its order.
//   2. count/assign : a point is "first" iff it sits in slot 0 of its cell; the exclusive scan of
//               the first-flags over the point index IS the sequential voxel id.  The first point
//               whose id equals max_voxels marks where the sequential loop `break`s (i*).
//   3. emit   : every kept point copies its row to voxels[vid][slot]; the first point of the cell
//               also writes coors / num_points and zero-fills the unused slots.
//
// Appending (lcf_voxelize_append): the recipe is incremental as it stands.
// The hash, the slot lists, cellslot / pvid and two counters (i*, number of first points so far)
// stay in the workspace; a call for the points [n_begin, n_end) inserts only those (their indices
// are larger than every resident one, so they queue behind them in the lists), scans first-flags
// over those only on top of the resident count, and emits only those -- the first appended point
// of a cell refreshes the cell's num_points.  The outputs equal a from-scratch run on [0, n_end).
#include "common.h"

#include <mutex>

#define VOX_SENT 0x7F7F7F7F /* list sentinel: byte-pattern memset, larger than any point index */
#define VOX_ITEMS 4          /* points per thread in the count / assign kernels */
#define VOX_BLOCK 256

struct VoxParams {
    float lo[3];
    float vs[3];
    int grid[3]; // x,y,z
};

__global__ void __launch_bounds__(VOX_BLOCK)
vox_insert_kernel(const float *__restrict__ pts, int n_begin, int n_cap,
                  const int *__restrict__ n_dev, int f, VoxParams p, int max_pts,
                  uint32_t *__restrict__ keys, uint32_t mask, int *__restrict__ lists,
                  int *__restrict__ cellslot, int *__restrict__ scan_state, int scan_nblk,
                  int *__restrict__ ctrl)
{
    const int n = lcf_count(n_cap, n_dev);
    // what vox_scan_kernel (next launch) expects: its per-block states at 0, and i* = "never" at
    // the start of a session
    const int gid = blockIdx.x * VOX_BLOCK + threadIdx.x;
    if (gid < scan_nblk) scan_state[gid] = 0;
    if (gid == 0) {
        // ctrl[0] = i*, ctrl[1] = first points so far, ctrl[2] = that count before this call,
        // ctrl[3] = blocks of vox_scan_kernel that gave up waiting and counted for themselves
        if (n_begin == 0) ctrl[0] = 0x7FFFFFFF, ctrl[3] = 0;
        ctrl[2] = n_begin == 0 ? 0 : ctrl[1];
    }
    int i = n_begin + gid;
    if (i >= n) return;
    int c[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // IEEE fp32 subtract and true divide, no contraction: bit-exact at voxel edges
        float d = __fsub_rn(pts[(size_t)i * f + j], p.lo[j]);
        float q = __fdiv_rn(d, p.vs[j]);
        float fl = floorf(q);
        if (!(fl >= 0.0f) || fl >= (float)p.grid[j]) ok = false;
        c[j] = (int)fl;
    }
    if (!ok) {
        cellslot[i] = -1;
        return;
    }
    uint32_t key = ((uint32_t)c[2] * (uint32_t)p.grid[1] + (uint32_t)c[1]) * (uint32_t)p.grid[0] +
                   (uint32_t)c[0];
    uint32_t h = lcf_hash_insert(keys, mask, key);
    cellslot[i] = (int)h;
    int cur = i;
    int *l = lists + (size_t)h * max_pts;
    for (int s = 0; s < max_pts; ++s) {
        int old = atomicMin(&l[s], cur);
        if (old == VOX_SENT) break; // landed in an empty slot
        cur = old > cur ? old : cur; // loser moves on
    }
}

__device__ __forceinline__ int vox_is_first(const int *cellslot, const int *lists, int max_pts,
                                            int i, int n)
{
    if (i >= n) return 0;
    int h = cellslot[i];
    if (h < 0) return 0;
    return lists[(size_t)h * max_pts] == i;
}

__global__ void __launch_bounds__(VOX_BLOCK)
vox_count_kernel(const int *__restrict__ cellslot, const int *__restrict__ lists, int max_pts,
                 int n_begin, int n_cap, const int *__restrict__ n_dev, int *__restrict__ bsums,
                 int *__restrict__ ctrl)
{
    __shared__ int lds[8];
    const int n = lcf_count(n_cap, n_dev);
    int base = n_begin + (blockIdx.x * VOX_BLOCK + threadIdx.x) * VOX_ITEMS;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < VOX_ITEMS; ++j) cnt += vox_is_first(cellslot, lists, max_pts, base + j, n);
    int total;
    lcf_block_excl_scan_256(cnt, lds, &total);
    if (threadIdx.x == 0) {
        bsums[blockIdx.x] = total;
        // ctrl[0] = i*, ctrl[1] = first points so far, ctrl[2] = that count before this call
        if (blockIdx.x == 0) {
            if (n_begin == 0) ctrl[0] = 0x7FFFFFFF; // i* = "never"
            ctrl[2] = n_begin == 0 ? 0 : ctrl[1];
        }
    }
}

__global__ void __launch_bounds__(VOX_BLOCK)
vox_assign_kernel(const int *__restrict__ cellslot, const int *__restrict__ lists, int max_pts,
                  int n_begin, int n_cap, const int *__restrict__ n_dev,
                  const int *__restrict__ bsums, int max_vox, int *__restrict__ pvid,
                  int *__restrict__ ctrl, int *__restrict__ voxel_num)
{
    __shared__ int lds[8];
    const int n = lcf_count(n_cap, n_dev);
    // prefix of the preceding blocks' sums, on top of the first points of the earlier calls
    int part = threadIdx.x == 0 ? ctrl[2] : 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += VOX_BLOCK) part += bsums[b];
    int blockbase;
    lcf_block_excl_scan_256(part, lds, &blockbase);
    int base = n_begin + (blockIdx.x * VOX_BLOCK + threadIdx.x) * VOX_ITEMS;
    int fl[VOX_ITEMS];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < VOX_ITEMS; ++j) {
        fl[j] = vox_is_first(cellslot, lists, max_pts, base + j, n);
        cnt += fl[j];
    }
    int total;
    int ex = lcf_block_excl_scan_256(cnt, lds, &total) + blockbase;
#pragma unroll
    for (int j = 0; j < VOX_ITEMS; ++j) {
        if (fl[j]) {
            pvid[base + j] = ex;
            if (ex == max_vox) ctrl[0] = base + j; // unique writer: the sequential loop stops here
            ++ex;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        int nfirst = blockbase + total;
        ctrl[1] = nfirst;
        voxel_num[0] = nfirst < max_vox ? nfirst : max_vox;
    }
}

// vox_count_kernel + vox_assign_kernel as ONE launch (decoupled look-back over the per-block
// counts: a block publishes its count, adds up what its predecessors published and hands out the
// ids; the same ids as the two-launch form).  state[b] = status << 30 | value: 0 nothing yet
// (cleared by vox_insert_kernel, one launch earlier), 1 the block's own count, 2 the count up to
// and including b (block 0: on top of the first points of the earlier calls of the session).
// Blocks are normally dispatched in index order and publish before they wait, so a wait is short;
// HIP does not promise that order, so a wait that does not end within `spin_limit` polls stops
// waiting and the block COUNTS THE FIRST POINTS IN FRONT OF IT ITSELF (the inserts finished a
// launch ago: flags of other blocks' points are plain reads).  The ids are therefore right
// whatever the dispatch order, the call never hangs, and ctrl[3] counts the blocks that took
// the slow way (lcf_voxelize_scan_fallbacks; FrameEngine.check() reports it).  The voxelizer is
// the head of a frame's critical path: one launch less is 5 us.
#define VOX_SPIN_LIMIT (1 << 16)
__global__ void __launch_bounds__(VOX_BLOCK)
vox_scan_kernel(const int *__restrict__ cellslot, const int *__restrict__ lists, int max_pts,
                int n_begin, int n_cap, const int *__restrict__ n_dev, int *__restrict__ bsums,
                int max_vox, int *__restrict__ pvid, int *__restrict__ ctrl,
                int *__restrict__ voxel_num, int have_insert, int spin_limit)
{
    __shared__ int lds[8];
    const int n = lcf_count(n_cap, n_dev);
    const int blk = blockIdx.x;
    int base = n_begin + (blk * VOX_BLOCK + threadIdx.x) * VOX_ITEMS;
    int fl[VOX_ITEMS];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < VOX_ITEMS; ++j) {
        fl[j] = vox_is_first(cellslot, lists, max_pts, base + j, n);
        cnt += fl[j];
    }
    int total;
    int ex = lcf_block_excl_scan_256(cnt, lds, &total);
    unsigned *state = reinterpret_cast<unsigned *>(bsums);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int sum = 0;
        if (blk == 0) {
            if (lane == 0) {
                // the first points of the session's earlier calls: ctrl[2], written by the insert
                // launch; with no insert launch (nothing new) this is the grid's only block
                if (have_insert) {
                    sum = ctrl[2];
                } else {
                    sum = n_begin == 0 ? 0 : ctrl[1];
                    ctrl[2] = sum;
                    if (n_begin == 0) ctrl[0] = 0x7FFFFFFF, ctrl[3] = 0;
                }
                atomicExch(&state[0], (2u << 30) | (unsigned)(sum + total));
            }
            sum = __shfl(sum, 0, 64);
        } else {
            if (lane == 0) atomicExch(&state[blk], (1u << 30) | (unsigned)total);
            int pos = blk - 1, polls = 0;
            bool done = false, gave_up = false;
            while (!done) {
                const int idx = pos - lane;
                unsigned v = 2u << 30;
                if (idx >= 0)
                    v = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned stt = v >> 30;
                if (__ballot(stt == 0u) != 0ull) {
                    if (++polls > spin_limit) done = gave_up = true;
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                const unsigned long long full = __ballot(stt == 2u);
                const int first = full ? __builtin_ctzll(full) : 64;
                int val = lane <= first ? (int)(v & 0x3FFFFFFFu) : 0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d, 64);
                sum += val;
                if (full)
                    done = true;
                else
                    pos -= 64;
            }
            if (gave_up) {
                // a predecessor has not published: count every first point in front of this
                // block directly, on top of the earlier calls' count
                const int end = n_begin + blk * VOX_BLOCK * VOX_ITEMS;
                int c = 0;
                for (int i = n_begin + lane; i < end; i += 64)
                    c += vox_is_first(cellslot, lists, max_pts, i, n);
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
                sum = c + ctrl[2];
                if (lane == 0) atomicAdd(&ctrl[3], 1);
            }
            if (lane == 0) atomicExch(&state[blk], (2u << 30) | (unsigned)(sum + total));
        }
        if (lane == 0) lds[4] = sum;
    }
    __syncthreads();
    const int blockbase = lds[4];
    ex += blockbase;
#pragma unroll
    for (int j = 0; j < VOX_ITEMS; ++j) {
        if (fl[j]) {
            pvid[base + j] = ex;
            if (ex == max_vox) ctrl[0] = base + j; // unique writer: the sequential loop stops here
            ++ex;
        }
    }
    if (blk == (int)gridDim.x - 1 && threadIdx.x == 0) {
        int nfirst = blockbase + total;
        ctrl[1] = nfirst;
        voxel_num[0] = nfirst < max_vox ? nfirst : max_vox;
    }
}

__global__ void __launch_bounds__(VOX_BLOCK)
vox_emit_kernel(const float *__restrict__ pts, int n_begin, int n_cap,
                const int *__restrict__ n_dev, int f,
                VoxParams p, int max_pts, int max_vox, int cap_mode, int batch_idx, int cstride,
                const uint32_t *__restrict__ keys,
                const int *__restrict__ lists, const int *__restrict__ cellslot,
                const int *__restrict__ pvid, const int *__restrict__ ctrl,
                const int *__restrict__ voxel_num, float *__restrict__ voxels,
                int *__restrict__ coors, int *__restrict__ num_pts)
{
    const int n = lcf_count(n_cap, n_dev);
    int i = blockIdx.x * VOX_BLOCK + threadIdx.x;
    // (a) zero the rows beyond voxel_num (padded outputs of generate_multi_gpu); an appending
    //     call finds them zero already
    if (n_begin == 0 && i < max_vox && i >= voxel_num[0]) {
        float *row = voxels + (size_t)i * max_pts * f;
        for (int e = 0; e < max_pts * f; ++e) row[e] = 0.f;
        for (int e = 0; e < cstride; ++e) coors[i * cstride + e] = 0;
        num_pts[i] = 0;
    }
    i += n_begin;
    if (i >= n) return;
    int h = cellslot[i];
    if (h < 0) return;
    const int istar = cap_mode == 0 ? ctrl[0] : 0x7FFFFFFF;
    if (i >= istar) return; // the sequential loop had already stopped
    const int *l = lists + (size_t)h * max_pts;
    int first = l[0];
    int vid = pvid[first];
    if (vid >= max_vox) return; // voxel never created
    int slot = -1;
    for (int s = 0; s < max_pts; ++s) {
        if (l[s] == i) {
            slot = s;
            break;
        }
    }
    if (slot < 0) return; // voxel already full when this point arrived
    float *dst = voxels + ((size_t)vid * max_pts + slot) * f;
    const float *src = pts + (size_t)i * f;
    for (int e = 0; e < f; ++e) dst[e] = src[e];
    // the first point of this call in its cell (= the cell's first point in a from-scratch
    // call) owns the cell's count; the cell's first point also owns its coordinates and padding
    if (slot == 0 || l[slot - 1] < n_begin) {
        int cnt = 0;
        for (int s = 0; s < max_pts; ++s) {
            int idx = l[s];
            if (idx != VOX_SENT && idx < istar) ++cnt;
        }
        num_pts[vid] = cnt;
        if (slot != 0) return;
        float *row = voxels + (size_t)vid * max_pts * f;
        for (int e = cnt * f; e < max_pts * f; ++e) row[e] = 0.f;
        uint32_t key = keys[h];
        uint32_t gx = (uint32_t)p.grid[0], gy = (uint32_t)p.grid[1];
        int *cr = coors + (size_t)vid * cstride + (cstride - 3);
        cr[2] = (int)(key % gx);
        cr[1] = (int)((key / gx) % gy);
        cr[0] = (int)(key / (gx * gy));
        if (cstride == 4) coors[(size_t)vid * 4] = batch_idx; // merge_second_batch's batch column
    }
}

struct VoxWs {
    uint32_t *keys;
    int *lists, *cellslot, *pvid, *bsums, *ctrl;
    uint32_t hc;
    int nblk;
    size_t bytes;
};

static VoxWs vox_carve(void *ws, int n, int max_pts)
{
    VoxWs w;
    uint32_t need = (uint32_t)(n > 0 ? n : 1) * 2u;
    w.hc = lcf_pow2_ge(need < 1024u ? 1024u : need);
    w.nblk = (n + VOX_BLOCK * VOX_ITEMS - 1) / (VOX_BLOCK * VOX_ITEMS);
    if (w.nblk < 1) w.nblk = 1;
    char *p = (char *)ws;
    size_t off = 0;
    w.keys = (uint32_t *)(p + off);
    off += lcf_align_up((size_t)w.hc * 4, 256);
    w.lists = (int *)(p + off);
    off += lcf_align_up((size_t)w.hc * max_pts * 4, 256);
    w.cellslot = (int *)(p + off);
    off += lcf_align_up((size_t)(n + VOX_ITEMS) * 4, 256);
    w.pvid = (int *)(p + off);
    off += lcf_align_up((size_t)(n + VOX_ITEMS) * 4, 256);
    w.bsums = (int *)(p + off);
    off += lcf_align_up((size_t)w.nblk * 4, 256);
    w.ctrl = (int *)(p + off);
    off += 256;
    w.bytes = off;
    return w;
}

extern "C" size_t lcf_voxelize_workspace_bytes(int n, int max_pts)
{
    if (n < 0 || max_pts <= 0) return 0;
    return vox_carve(nullptr, n, max_pts).bytes;
}

static int vox_reset(const VoxWs &w, int max_pts, hipStream_t st, lcf_ctx *ctx = nullptr)
{
    if (lcf_fill32_ctx(ctx, w.keys, 0xFFFFFFFFu, (size_t)w.hc * 4, st) != LCF_OK)
        return LCF_ERR_LAUNCH;
    if (lcf_fill32_ctx(ctx, w.lists, (uint32_t)VOX_SENT, (size_t)w.hc * max_pts * 4, st) != LCF_OK)
        return LCF_ERR_LAUNCH;
    return LCF_OK;
}

extern "C" int lcf_voxelize_workspace_reset(void *ws, size_t ws_bytes, int n, int max_pts,
                                            void *stream, lcf_ctx *ctx)
{
    if (!ws || n < 0 || max_pts <= 0) return LCF_ERR_ARG;
    VoxWs w = vox_carve(ws, n, max_pts);
    if (ws_bytes < w.bytes) return LCF_ERR_WORKSPACE;
    return vox_reset(w, max_pts, (hipStream_t)stream, ctx);
}

// Points [n_begin, n) of a cloud whose workspace is carved for n_carve points; n_begin = 0 is
// the plain voxelizer.
static int vox_run(const float *points, int n, const int *n_dev, int f, const float vs[3],
                   const float range[6], int max_pts, int max_vox, int cap_mode, int batch_idx,
                   int coor_stride, float *voxels, int *coors, int *num_pts, int *voxel_num,
                   void *ws, size_t ws_bytes, bool reset, hipStream_t st, int n_begin = 0,
                   int n_carve = -1, lcf_ctx *ctx = nullptr)
{
    if (n_carve < 0) n_carve = n;
    if (n_begin < 0 || n_begin > n || n > n_carve) return LCF_ERR_ARG;
    if (n < 0 || f < 3 || max_pts <= 0 || max_vox <= 0 || !vs || !range || !voxels || !coors ||
        !num_pts || !voxel_num || !ws || (n > 0 && !points) || cap_mode < 0 || cap_mode > 1 ||
        (coor_stride != 3 && coor_stride != 4))
        return LCF_ERR_ARG;
    if (n >= VOX_SENT - VOX_BLOCK * VOX_ITEMS) return LCF_ERR_RANGE;
    VoxParams p;
    unsigned long long cells = 1;
    for (int j = 0; j < 3; ++j) {
        p.lo[j] = range[j];
        p.vs[j] = vs[j];
        // grid_size = round((hi - lo) / vs) in fp32 (simplevis.py:26-29)
        volatile float g = (range[3 + j] - range[j]) / vs[j];
        p.grid[j] = (int)rintf(g);
        if (p.grid[j] <= 0) return LCF_ERR_ARG;
        cells *= (unsigned long long)p.grid[j];
    }
    if (cells >= 0xFFFFFFFFull) return LCF_ERR_RANGE;
    VoxWs w = vox_carve(ws, n_carve, max_pts);
    if (ws_bytes < w.bytes) return LCF_ERR_WORKSPACE;
    if (reset && vox_reset(w, max_pts, st, ctx) != LCF_OK) return LCF_ERR_LAUNCH;
    const int n_new = n - n_begin;
    int nblk = (n_new + VOX_BLOCK * VOX_ITEMS - 1) / (VOX_BLOCK * VOX_ITEMS);
    if (nblk < 1) nblk = 1;
    // LCF_VOX_TWO_PASS (environment, A/B and fallback): count and assign as two launches
    static const bool two_pass = getenv("LCF_VOX_TWO_PASS") != nullptr;
    // LCF_VOX_SPIN_LIMIT (environment, tests): polls a block of the one-launch scan waits for a
    // predecessor before it counts for itself; 0 forces that path wherever a wait would start
    static const int spin_limit =
        getenv("LCF_VOX_SPIN_LIMIT") ? atoi(getenv("LCF_VOX_SPIN_LIMIT")) : VOX_SPIN_LIMIT;
    if (n_new > 0) {
        vox_insert_kernel<<<(n_new + VOX_BLOCK - 1) / VOX_BLOCK, VOX_BLOCK, 0, st>>>(
            points, n_begin, n, n_dev, f, p, max_pts, w.keys, w.hc - 1, w.lists, w.cellslot,
            w.bsums, two_pass ? 0 : nblk, w.ctrl);
        LCF_CHECK_LAUNCH();
    }
    if (two_pass) {
        vox_count_kernel<<<nblk, VOX_BLOCK, 0, st>>>(w.cellslot, w.lists, max_pts, n_begin, n,
                                                     n_dev, w.bsums, w.ctrl);
        LCF_CHECK_LAUNCH();
        vox_assign_kernel<<<nblk, VOX_BLOCK, 0, st>>>(w.cellslot, w.lists, max_pts, n_begin, n,
                                                      n_dev, w.bsums, max_vox, w.pvid, w.ctrl,
                                                      voxel_num);
    } else {
        vox_scan_kernel<<<nblk, VOX_BLOCK, 0, st>>>(w.cellslot, w.lists, max_pts, n_begin, n, n_dev,
                                                    w.bsums, max_vox, w.pvid, w.ctrl, voxel_num,
                                                    n_new > 0 ? 1 : 0, spin_limit);
    }
    LCF_CHECK_LAUNCH();
    int m = n_begin == 0 && max_vox > n_new ? max_vox : n_new;
    if (m > 0) {
        vox_emit_kernel<<<(m + VOX_BLOCK - 1) / VOX_BLOCK, VOX_BLOCK, 0, st>>>(
            points, n_begin, n, n_dev, f, p, max_pts, max_vox, cap_mode, batch_idx, coor_stride,
            w.keys, w.lists, w.cellslot, w.pvid, w.ctrl, voxel_num, voxels, coors, num_pts);
    }
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_voxelize(const float *points, int n, const int *n_dev, int f,
                            const float vs[3], const float range[6], int max_pts, int max_vox,
                            int cap_mode, int batch_idx, int coor_stride, float *voxels,
                            int *coors, int *num_pts, int *voxel_num, void *ws, size_t ws_bytes,
                            void *stream, lcf_ctx *ctx)
{
    const bool ws_clean = ctx && ctx->o.ws_clean;
    return vox_run(points, n, n_dev, f, vs, range, max_pts, max_vox, cap_mode, batch_idx,
                   coor_stride, voxels, coors, num_pts, voxel_num, ws, ws_bytes, !ws_clean,
                   (hipStream_t)stream);
}

// Append the points [n_begin, n_end) of a resident cloud to a resident voxelisation (see the
// header of this file).  n_begin == 0 starts a session: the workspace is re-armed first.
extern "C" int lcf_voxelize_append(const float *points, int n_capacity, int n_begin, int n_end,
                                   int f, const float vs[3], const float range[6], int max_pts,
                                   int max_vox, int cap_mode, int batch_idx, int coor_stride,
                                   float *voxels, int *coors, int *num_pts, int *voxel_num,
                                   void *ws, size_t ws_bytes, void *stream, lcf_ctx *ctx)
{
    if (n_capacity < 0 || n_begin < 0 || n_end < n_begin || n_end > n_capacity) return LCF_ERR_ARG;
    const bool ws_clean = ctx && ctx->o.ws_clean;
    return vox_run(points, n_end, nullptr, f, vs, range, max_pts, max_vox, cap_mode, batch_idx,
                   coor_stride, voxels, coors, num_pts, voxel_num, ws, ws_bytes,
                   n_begin == 0 && !ws_clean, (hipStream_t)stream, n_begin, n_capacity);
}

// Device address of the workspace's counter of vox_scan_kernel blocks that stopped waiting for a
// predecessor and counted for themselves since the session began (results are right either way;
// a non-zero count says the one-launch scan ran slowly, e.g. under heavy concurrent streams).
extern "C" int lcf_voxelize_scan_fallbacks(void *ws, size_t ws_bytes, int n, int max_pts,
                                           const int **counter_dev)
{
    if (!ws || n < 0 || max_pts <= 0 || !counter_dev) return LCF_ERR_ARG;
    VoxWs w = vox_carve(ws, n, max_pts);
    if (ws_bytes < w.bytes) return LCF_ERR_WORKSPACE;
    *counter_dev = w.ctrl + 3;
    return LCF_OK;
}

// ---------------------------------------------------------------- the entry point that allocates
// nothing: lcf_voxelize with its hash workspace in a file-scope device array instead of a caller
// buffer, for callers that cannot allocate (code under hipGraph capture, tight serving loops,
// harnesses that forbid hipMalloc).  Every GPU the library runs on holds its own copy of the
// array until the process ends; all calls on one GPU share it (overlapping calls race).
static __device__ __align__(256) unsigned char scratch[LCF_VOXELIZE_STATIC_BYTES];

#define VOX_MAX_DEVICES 64
static void *g_scratch_ptr[VOX_MAX_DEVICES]; // device address of `scratch`, per device
static std::mutex g_scratch_mu;

// The array's name is a host-side stand-in, not a device address: resolve the address once per
// device, with that device current, and use only the resolved pointer afterwards.
static int vox_scratch(void **out)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VOX_MAX_DEVICES)
        return LCF_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        if (hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr)
            return LCF_ERR_LAUNCH;
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return LCF_OK;
}

extern "C" size_t lcf_voxelize_static_max_bytes(void) { return LCF_VOXELIZE_STATIC_BYTES; }

extern "C" int lcf_voxelize_static(const float *points, int n, const int *n_dev, int f,
                                   const float vs[3], const float range[6], int max_pts,
                                   int max_vox, int cap_mode, int batch_idx, int coor_stride,
                                   float *voxels, int *coors, int *num_pts, int *voxel_num,
                                   void *stream)
{
    if (n < 0 || max_pts <= 0) return LCF_ERR_ARG;
    // sizes beyond the array: a clear error, never a partial run
    if (lcf_voxelize_workspace_bytes(n, max_pts) > (size_t)LCF_VOXELIZE_STATIC_BYTES)
        return LCF_ERR_STATIC_LIMIT;
    void *ptr = nullptr;
    int rc = vox_scratch(&ptr);
    if (rc != LCF_OK) return rc;
    // the array keeps the previous call's contents: always re-armed here (reset = true)
    return vox_run(points, n, n_dev, f, vs, range, max_pts, max_vox, cap_mode, batch_idx,
                   coor_stride, voxels, coors, num_pts, voxel_num, ptr,
                   (size_t)LCF_VOXELIZE_STATIC_BYTES, true, (hipStream_t)stream);
}
