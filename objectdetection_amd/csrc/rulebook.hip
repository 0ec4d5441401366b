// rulebook.hip -- get_indice_pairs for gfx950: coordinate hashing + neighbour tables, and the
// conversion to spconv's canonical compacted pair lists.
//
// Follows the spconv 1.x CPU algorithm (getValidOutPos / getIndicePairsConv /
// getIndicePairsSubM; spconv is an un-vendored dependency of the reference, README.md:29) as
// restated in oracle/lcf_oracle.c; call sites second/pytorch/models/middle.py:145-192.
//
// Canonical order (what "bit-exact rulebook" means here):
//   * strided conv outputs are numbered in CPU discovery order: ascending input row j, and for
//     one j the getValidOutPos enumeration (every dim from its upper bound downward, x fastest).
//     Parallel form: each candidate (j,e) carries the stamp j*E+e; an output cell keeps the
//     minimum stamp (64-bit atomicMin on key<<32|stamp); the exclusive scan of the
//     "I am my cell's minimum" flags over the stamp order is the sequential output id.
//   * pairs of one kernel offset are listed by ascending input row.
#include "rb_common.h"

// ---------------------------------------------------------------- submanifold

__global__ void __launch_bounds__(RB_BLOCK)
rb_subm_insert_kernel(const int *__restrict__ coors, int k_cap, const int *__restrict__ k_dev,
                      int batch, ConvGeom g, uint32_t *__restrict__ keys, uint32_t mask,
                      int *__restrict__ vals)
{
    const int k = lcf_count(k_cap, k_dev);
    int j = blockIdx.x * RB_BLOCK + threadIdx.x;
    if (j >= k) return;
    int4 c = reinterpret_cast<const int4 *>(coors)[j];
    if (c.x < 0 || c.x >= batch || c.y < 0 || c.y >= g.shape[0] || c.z < 0 ||
        c.z >= g.shape[1] || c.w < 0 || c.w >= g.shape[2])
        return;
    uint32_t h = lcf_hash_insert(keys, mask, rb_key(c.x, c.y, c.z, c.w, g.shape));
    vals[h] = j; // coordinates are unique per row (voxel ids); last writer wins otherwise
}

__global__ void __launch_bounds__(RB_BLOCK)
rb_subm_lookup_kernel(const int *__restrict__ coors, int k_cap, const int *__restrict__ k_dev,
                      int batch, ConvGeom g, const uint32_t *__restrict__ keys, uint32_t mask,
                      const int *__restrict__ vals, int *__restrict__ nbr)
{
    const int k = lcf_count(k_cap, k_dev);
    long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x;
    if (t >= (long long)k * g.kvol) return;
    int n = (int)(t / g.kvol);
    int o = (int)(t - (long long)n * g.kvol);
    int4 c = reinterpret_cast<const int4 *>(coors)[n];
    int kx = o % g.ks[2];
    int ky = (o / g.ks[2]) % g.ks[1];
    int kz = o / (g.ks[2] * g.ks[1]);
    // input position feeding output n through offset o: in = out - pad + k*dil (stride 1)
    int z = c.y - g.pd[0] + kz * g.dl[0];
    int y = c.z - g.pd[1] + ky * g.dl[1];
    int x = c.w - g.pd[2] + kx * g.dl[2];
    int res = -1;
    if (c.x >= 0 && c.x < batch && z >= 0 && z < g.shape[0] && y >= 0 && y < g.shape[1] &&
        x >= 0 && x < g.shape[2]) {
        uint32_t h = lcf_hash_find(keys, mask, rb_key(c.x, z, y, x, g.shape));
        if (h != 0xFFFFFFFFu) res = vals[h];
    }
    nbr[t] = res;
}

__global__ void __launch_bounds__(RB_BLOCK)
rb_mirror_kernel(const int *__restrict__ nbr, long long total, int kvol, int *__restrict__ out)
{
    long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x;
    if (t >= total) return;
    long long n = t / kvol;
    int o = (int)(t - n * kvol);
    out[t] = nbr[n * kvol + (kvol - 1 - o)];
}

// ---------------------------------------------------------------- strided conv

__global__ void __launch_bounds__(RB_BLOCK)
rb_conv_insert_kernel(const int *__restrict__ coors, int k_cap, const int *__restrict__ k_dev,
                      int batch, ConvGeom g, unsigned long long *__restrict__ tab, uint32_t mask,
                      int *__restrict__ cand_slot)
{
    const int k_in = lcf_count(k_cap, k_dev);
    long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x;
    if (t >= (long long)k_in * g.E) return;
    int j = (int)(t / g.E);
    int e = (int)(t - (long long)j * g.E);
    int4 c = reinterpret_cast<const int4 *>(coors)[j];
    int oz, oy, ox, off;
    bool ok = c.x >= 0 && c.x < batch && rb_candidate(c, e, g, &oz, &oy, &ox, &off);
    if (!ok) {
        cand_slot[t] = -1;
        return;
    }
    uint32_t key = rb_key(c.x, oz, oy, ox, g.oshape);
    unsigned long long mine = ((unsigned long long)key << 32) | (unsigned long long)(uint32_t)t;
    uint32_t h = lcf_hash32(key) & mask;
    while (true) {
        unsigned long long old = atomicCAS(&tab[h], RB_EMPTY64, mine);
        if (old == RB_EMPTY64) break;
        if ((uint32_t)(old >> 32) == key) {
            atomicMin(&tab[h], mine);
            break;
        }
        h = (h + 1) & mask;
    }
    cand_slot[t] = (int)h;
}

__global__ void __launch_bounds__(RB_BLOCK)
rb_conv_count_kernel(const unsigned long long *__restrict__ tab,
                     const int *__restrict__ cand_slot, int k_cap,
                     const int *__restrict__ k_dev, int E, int *__restrict__ bsums)
{
    __shared__ int lds[8];
    const long long total = (long long)lcf_count(k_cap, k_dev) * E;
    long long base = ((long long)blockIdx.x * RB_BLOCK + threadIdx.x) * RB_ITEMS;
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < RB_ITEMS; ++i) cnt += rb_is_discoverer(tab, cand_slot, base + i, total);
    int tot;
    lcf_block_excl_scan_256(cnt, lds, &tot);
    if (threadIdx.x == 0) bsums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(RB_BLOCK)
rb_conv_assign_kernel(const unsigned long long *__restrict__ tab,
                      const int *__restrict__ cand_slot, int k_cap,
                      const int *__restrict__ k_dev, const int *__restrict__ bsums, ConvGeom g,
                      int cap_out, int *__restrict__ vals, int *__restrict__ out_coors,
                      int *__restrict__ k_out, int *__restrict__ overflow)
{
    __shared__ int lds[8];
    const long long total = (long long)lcf_count(k_cap, k_dev) * g.E;
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += RB_BLOCK) part += bsums[b];
    int blockbase;
    lcf_block_excl_scan_256(part, lds, &blockbase);
    long long base = ((long long)blockIdx.x * RB_BLOCK + threadIdx.x) * RB_ITEMS;
    int fl[RB_ITEMS];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < RB_ITEMS; ++i) {
        fl[i] = rb_is_discoverer(tab, cand_slot, base + i, total);
        cnt += fl[i];
    }
    int tot;
    int ex = lcf_block_excl_scan_256(cnt, lds, &tot) + blockbase;
#pragma unroll
    for (int i = 0; i < RB_ITEMS; ++i) {
        if (fl[i]) {
            int h = cand_slot[base + i];
            vals[h] = ex;
            if (ex < cap_out) {
                uint32_t key = (uint32_t)(tab[h] >> 32);
                uint32_t W = g.oshape[2], H = g.oshape[1], D = g.oshape[0];
                int4 oc;
                oc.w = (int)(key % W);
                oc.z = (int)((key / W) % H);
                oc.y = (int)((key / (W * H)) % D);
                oc.x = (int)(key / (W * H * D));
                reinterpret_cast<int4 *>(out_coors)[ex] = oc;
            }
            ++ex;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const int n = blockbase + tot;
        k_out[0] = n < cap_out ? n : cap_out; // outputs past the capacity are dropped ...
        if (n > cap_out && overflow) overflow[0] = 1; // ... and reported
    }
}

__global__ void __launch_bounds__(RB_BLOCK)
rb_conv_fill_kernel(const int *__restrict__ coors, int k_cap, const int *__restrict__ k_dev,
                    int batch, ConvGeom g, int k_out_cap, const int *__restrict__ k_out_dev,
                    const int *__restrict__ cand_slot, const int *__restrict__ vals,
                    int *__restrict__ out_nbr, int *__restrict__ in_nbr)
{
    const int k_in = lcf_count(k_cap, k_dev);
    const int k_out = lcf_count(k_out_cap, k_out_dev);
    long long t = (long long)blockIdx.x * RB_BLOCK + threadIdx.x;
    if (t >= (long long)k_in * g.E) return;
    int h = cand_slot[t];
    if (h < 0) return;
    int j = (int)(t / g.E);
    int e = (int)(t - (long long)j * g.E);
    int4 c = reinterpret_cast<const int4 *>(coors)[j];
    int oz, oy, ox, off;
    if (!rb_candidate(c, e, g, &oz, &oy, &ox, &off)) return;
    int n = vals[h];
    if (n >= k_out) return;
    out_nbr[(size_t)n * g.kvol + off] = j;
    if (in_nbr) in_nbr[(size_t)j * g.kvol + off] = n;
}

// ---------------------------------------------------------------- canonical pair lists

__global__ void __launch_bounds__(RB_BLOCK)
rb_pairs_count_kernel(const int *__restrict__ in_nbr, int k_in, int kvol,
                      int *__restrict__ bcnt /*[nblk][kvol]*/)
{
    __shared__ int wcnt[4][LCF_MAX_KVOL];
    int j = blockIdx.x * RB_BLOCK + threadIdx.x;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 0; o < kvol; ++o) {
        bool v = j < k_in && in_nbr[(size_t)j * kvol + o] >= 0;
        unsigned long long m = __ballot(v);
        if (lane == 0) wcnt[wave][o] = __popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < kvol)
        bcnt[(size_t)blockIdx.x * kvol + threadIdx.x] = wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] +
                                                        wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
}

// one block: thread o turns bcnt[:, o] into exclusive prefixes and writes pair_num[o]
__global__ void rb_pairs_scan_kernel(int *__restrict__ bcnt, int nblk, int kvol,
                                     int *__restrict__ pair_num)
{
    int o = threadIdx.x;
    if (o >= kvol) return;
    int run = 0;
    for (int b = 0; b < nblk; ++b) {
        int v = bcnt[(size_t)b * kvol + o];
        bcnt[(size_t)b * kvol + o] = run;
        run += v;
    }
    pair_num[o] = run;
}

__global__ void __launch_bounds__(RB_BLOCK)
rb_pairs_write_kernel(const int *__restrict__ in_nbr, int k_in, int kvol,
                      const int *__restrict__ bbase, int *__restrict__ pairs)
{
    __shared__ int wcnt[4][LCF_MAX_KVOL];
    int j = blockIdx.x * RB_BLOCK + threadIdx.x;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    // pass 1: wave counts
    for (int o = 0; o < kvol; ++o) {
        bool v = j < k_in && in_nbr[(size_t)j * kvol + o] >= 0;
        unsigned long long m = __ballot(v);
        if (lane == 0) wcnt[wave][o] = __popcll(m);
    }
    __syncthreads();
    for (int o = 0; o < kvol; ++o) {
        int n = j < k_in ? in_nbr[(size_t)j * kvol + o] : -1;
        bool v = n >= 0;
        unsigned long long m = __ballot(v);
        if (v) {
            int pos = bbase[(size_t)blockIdx.x * kvol + o] + __popcll(m & lt);
            for (int w = 0; w < wave; ++w) pos += wcnt[w][o];
            pairs[(size_t)(0 * kvol + o) * k_in + pos] = j;
            pairs[(size_t)(1 * kvol + o) * k_in + pos] = n;
        }
    }
}

// ---------------------------------------------------------------- host side

extern "C" void lcf_conv_out_shape(const int shape[3], const int ksize[3], const int stride[3],
                                   const int pad[3], const int dil[3], int out_shape[3])
{
    for (int d = 0; d < 3; ++d)
        out_shape[d] = (shape[d] + 2 * pad[d] - dil[d] * (ksize[d] - 1) - 1) / stride[d] + 1;
}

extern "C" size_t lcf_rulebook_subm_workspace_bytes(int k)
{
    if (k < 0 || k >= 0x3FFFFFFF) return 0;
    uint32_t hc = rb_capacity(k);
    return lcf_align_up((size_t)hc * 4, 256) * 2;
}

extern "C" int lcf_rulebook_subm(const int *coors, int k, const int *k_dev, int batch,
                                 const int shape[3], const int ksize[3], const int dil[3],
                                 int *out_nbr, void *ws, size_t ws_bytes, void *stream,
                                 lcf_ctx *ctx)
{
    if (k < 0 || (k > 0 && (!coors || !out_nbr)) || !ws) return LCF_ERR_ARG;
    ConvGeom g;
    int rc = rb_geom(&g, batch, shape, ksize, nullptr, nullptr, dil, true);
    if (rc) return rc;
    // rows < 2^25: what the conv kernels' packed gather entries can address (conv.hip)
    if ((long long)k * g.kvol >= 0x7FFFFFFFll || k >= (1 << 25)) return LCF_ERR_RANGE;
    if (ws_bytes < lcf_rulebook_subm_workspace_bytes(k)) return LCF_ERR_WORKSPACE;
    if (k == 0) return LCF_OK;
    uint32_t hc = rb_capacity(k);
    uint32_t *keys = (uint32_t *)ws;
    int *vals = (int *)((char *)ws + lcf_align_up((size_t)hc * 4, 256));
    hipStream_t st = (hipStream_t)stream;
    if (!(ctx && ctx->o.ws_clean) && lcf_fill32(keys, 0xFFFFFFFFu, (size_t)hc * 4, st) != LCF_OK)
        return LCF_ERR_LAUNCH;
    rb_subm_insert_kernel<<<(k + RB_BLOCK - 1) / RB_BLOCK, RB_BLOCK, 0, st>>>(
        coors, k, k_dev, batch, g, keys, hc - 1, vals);
    LCF_CHECK_LAUNCH();
    long long total = (long long)k * g.kvol;
    rb_subm_lookup_kernel<<<(unsigned)((total + RB_BLOCK - 1) / RB_BLOCK), RB_BLOCK, 0, st>>>(
        coors, k, k_dev, batch, g, keys, hc - 1, vals, out_nbr);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_rulebook_subm_workspace_reset(void *ws, size_t ws_bytes, int k, void *stream,
                                                 lcf_ctx *ctx)
{
    if (!ws || k < 0) return LCF_ERR_ARG;
    if (ws_bytes < lcf_rulebook_subm_workspace_bytes(k)) return LCF_ERR_WORKSPACE;
    return lcf_fill32_ctx(ctx, ws, 0xFFFFFFFFu, (size_t)rb_capacity(k) * 4, (hipStream_t)stream);
}

extern "C" int lcf_rulebook_mirror(const int *nbr, int k, int kvol, int *mirrored, void *stream)
{
    if (k < 0 || kvol <= 0 || (k > 0 && (!nbr || !mirrored))) return LCF_ERR_ARG;
    if (k == 0) return LCF_OK;
    long long total = (long long)k * kvol;
    rb_mirror_kernel<<<(unsigned)((total + RB_BLOCK - 1) / RB_BLOCK), RB_BLOCK, 0,
                       (hipStream_t)stream>>>(nbr, total, kvol, mirrored);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_rulebook_conv_max_out(int k_in, const int ksize[3], const int stride[3])
{
    long long e = 1;
    for (int d = 0; d < 3; ++d) e *= (ksize[d] + stride[d] - 1) / stride[d];
    long long m = (long long)k_in * e;
    return m > 0x7FFFFFFFll ? 0x7FFFFFFF : (int)m;
}

struct ConvWs {
    unsigned long long *tab;
    int *vals, *cand_slot, *bsums;
    uint32_t hc;
    int nblk;
    long long total;
    size_t bytes;
};

static ConvWs rb_conv_carve(void *ws, int k_in, int E)
{
    ConvWs w;
    w.total = (long long)k_in * E;
    w.hc = rb_capacity(w.total);
    w.nblk = (int)((w.total + RB_BLOCK * RB_ITEMS - 1) / (RB_BLOCK * RB_ITEMS));
    if (w.nblk < 1) w.nblk = 1;
    char *p = (char *)ws;
    size_t off = 0;
    w.tab = (unsigned long long *)(p + off);
    off += lcf_align_up((size_t)w.hc * 8, 256);
    w.vals = (int *)(p + off);
    off += lcf_align_up((size_t)w.hc * 4, 256);
    w.cand_slot = (int *)(p + off);
    off += lcf_align_up((size_t)(w.total + RB_ITEMS) * 4, 256);
    w.bsums = (int *)(p + off);
    off += lcf_align_up((size_t)w.nblk * 4, 256);
    w.bytes = off;
    return w;
}

extern "C" size_t lcf_rulebook_conv_workspace_bytes(int k_in, const int ksize[3],
                                                    const int stride[3])
{
    if (k_in < 0 || !ksize || !stride) return 0;
    long long e = 1;
    for (int d = 0; d < 3; ++d) {
        if (ksize[d] <= 0 || stride[d] <= 0) return 0;
        e *= (ksize[d] + stride[d] - 1) / stride[d];
    }
    if ((long long)k_in * e >= 0x3FFFFFFFll) return 0; // LCF_ERR_RANGE at the call
    return rb_conv_carve(nullptr, k_in, (int)e).bytes;
}

extern "C" int lcf_rulebook_conv_workspace_reset(void *ws, size_t ws_bytes, int k_in,
                                                 const int ksize[3], const int stride[3],
                                                 void *stream, lcf_ctx *ctx)
{
    if (!ws || k_in < 0 || !ksize || !stride) return LCF_ERR_ARG;
    long long e = 1;
    for (int d = 0; d < 3; ++d) {
        if (ksize[d] <= 0 || stride[d] <= 0) return LCF_ERR_ARG;
        e *= (ksize[d] + stride[d] - 1) / stride[d];
    }
    if ((long long)k_in * e >= 0x3FFFFFFFll) return LCF_ERR_RANGE;
    ConvWs w = rb_conv_carve(ws, k_in, (int)e);
    if (ws_bytes < w.bytes) return LCF_ERR_WORKSPACE;
    return lcf_fill32_ctx(ctx, w.tab, 0xFFFFFFFFu, (size_t)w.hc * 8, (hipStream_t)stream);
}

extern "C" int lcf_rulebook_conv_discover(const int *coors, int k_in, const int *k_in_dev,
                                          int batch, const int shape[3], const int ksize[3],
                                          const int stride[3], const int pad[3],
                                          const int dil[3], int *out_coors, int cap_out,
                                          int *k_out, int *overflow, void *ws, size_t ws_bytes,
                                          void *stream, lcf_ctx *ctx)
{
    if (k_in < 0 || !stride || !pad || !k_out || !ws || (k_in > 0 && (!coors || !out_coors)))
        return LCF_ERR_ARG;
    ConvGeom g;
    int rc = rb_geom(&g, batch, shape, ksize, stride, pad, dil, false);
    if (rc) return rc;
    // 32-bit stamps and a 2x table; rows < 2^25 (the conv kernels' packed gather entries)
    if ((long long)k_in * g.E >= 0x3FFFFFFFll || k_in >= (1 << 25)) return LCF_ERR_RANGE;
    ConvWs w = rb_conv_carve(ws, k_in, g.E);
    if (ws_bytes < w.bytes) return LCF_ERR_WORKSPACE;
    if (cap_out < 0 || (!overflow && cap_out < lcf_rulebook_conv_max_out(k_in, ksize, stride)))
        return LCF_ERR_ARG; // a capacity below the worst case needs somewhere to report overflow
    hipStream_t st = (hipStream_t)stream;
    if (k_in == 0) {
        if (lcf_fill32(k_out, 0u, 4, st) != LCF_OK) return LCF_ERR_LAUNCH;
        return LCF_OK;
    }
    if (!(ctx && ctx->o.ws_clean) &&
        lcf_fill32(w.tab, 0xFFFFFFFFu, (size_t)w.hc * 8, st) != LCF_OK)
        return LCF_ERR_LAUNCH;
    unsigned nb = (unsigned)((w.total + RB_BLOCK - 1) / RB_BLOCK);
    rb_conv_insert_kernel<<<nb, RB_BLOCK, 0, st>>>(coors, k_in, k_in_dev, batch, g, w.tab,
                                                   w.hc - 1, w.cand_slot);
    LCF_CHECK_LAUNCH();
    rb_conv_count_kernel<<<w.nblk, RB_BLOCK, 0, st>>>(w.tab, w.cand_slot, k_in, k_in_dev, g.E,
                                                      w.bsums);
    LCF_CHECK_LAUNCH();
    rb_conv_assign_kernel<<<w.nblk, RB_BLOCK, 0, st>>>(w.tab, w.cand_slot, k_in, k_in_dev, w.bsums,
                                                       g, cap_out, w.vals, out_coors, k_out,
                                                       overflow);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_rulebook_conv_fill(const int *coors, int k_in, const int *k_in_dev, int batch,
                                      const int shape[3], const int ksize[3],
                                      const int stride[3], const int pad[3], const int dil[3],
                                      int k_out, const int *k_out_dev, int *out_nbr,
                                      int *in_nbr, void *ws, size_t ws_bytes, void *stream,
                                      lcf_ctx *ctx)
{
    if (k_in < 0 || k_out < 0 || !stride || !pad || !ws || (k_in > 0 && !coors) ||
        (k_out > 0 && !out_nbr))
        return LCF_ERR_ARG;
    ConvGeom g;
    int rc = rb_geom(&g, batch, shape, ksize, stride, pad, dil, false);
    if (rc) return rc;
    if ((long long)k_in * g.E >= 0x3FFFFFFFll) return LCF_ERR_RANGE;
    ConvWs w = rb_conv_carve(ws, k_in, g.E);
    if (ws_bytes < w.bytes) return LCF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!(ctx && ctx->o.nbr_clean)) { // LCF_OPT_NBR_CLEAN: the caller filled both tables with -1
        if (k_out > 0 &&
            lcf_fill32(out_nbr, 0xFFFFFFFFu, (size_t)k_out * g.kvol * 4, st) != LCF_OK)
            return LCF_ERR_LAUNCH;
        if (k_in > 0 && in_nbr &&
            lcf_fill32(in_nbr, 0xFFFFFFFFu, (size_t)k_in * g.kvol * 4, st) != LCF_OK)
            return LCF_ERR_LAUNCH;
    }
    if (k_in == 0 || k_out == 0) return LCF_OK;
    unsigned nb = (unsigned)((w.total + RB_BLOCK - 1) / RB_BLOCK);
    rb_conv_fill_kernel<<<nb, RB_BLOCK, 0, st>>>(coors, k_in, k_in_dev, batch, g, k_out, k_out_dev,
                                                 w.cand_slot, w.vals, out_nbr, in_nbr);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" size_t lcf_rulebook_pairs_workspace_bytes(int k_in, int kvol)
{
    if (k_in < 0 || kvol <= 0) return 0;
    size_t nblk = (size_t)(k_in + RB_BLOCK - 1) / RB_BLOCK;
    if (nblk < 1) nblk = 1;
    return lcf_align_up(nblk * kvol * 4, 256);
}

extern "C" int lcf_rulebook_pairs(const int *in_nbr, int k_in, int kvol, int *pairs,
                                  int *pair_num, void *ws, size_t ws_bytes, void *stream)
{
    if (k_in < 0 || kvol <= 0 || !pair_num || !ws || (k_in > 0 && (!in_nbr || !pairs)))
        return LCF_ERR_ARG;
    if (kvol > LCF_MAX_KVOL) return LCF_ERR_UNSUPPORTED;
    if (ws_bytes < lcf_rulebook_pairs_workspace_bytes(k_in, kvol)) return LCF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (k_in == 0) {
        if (lcf_fill32(pair_num, 0u, (size_t)kvol * 4, st) != LCF_OK) return LCF_ERR_LAUNCH;
        return LCF_OK;
    }
    int nblk = (k_in + RB_BLOCK - 1) / RB_BLOCK;
    int *bcnt = (int *)ws;
    if (lcf_fill32(pairs, 0xFFFFFFFFu, (size_t)2 * kvol * k_in * 4, st) != LCF_OK)
        return LCF_ERR_LAUNCH;
    rb_pairs_count_kernel<<<nblk, RB_BLOCK, 0, st>>>(in_nbr, k_in, kvol, bcnt);
    LCF_CHECK_LAUNCH();
    rb_pairs_scan_kernel<<<1, 64, 0, st>>>(bcnt, nblk, kvol, pair_num);
    LCF_CHECK_LAUNCH();
    rb_pairs_write_kernel<<<nblk, RB_BLOCK, 0, st>>>(in_nbr, k_in, kvol, bcnt, pairs);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}
