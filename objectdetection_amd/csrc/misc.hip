// misc.hip -- dense() scatter, VFE streaming kernels, error strings.
#include "common.h"

extern "C" const char *lcf_strerror(int rc)
{
    switch (rc) {
    case LCF_OK: return "ok";
    case LCF_ERR_ARG: return "lcf: invalid argument";
    case LCF_ERR_WORKSPACE: return "lcf: workspace too small";
    case LCF_ERR_UNSUPPORTED: return "lcf: unsupported shape for the gfx950 kernels";
    case LCF_ERR_RANGE: return "lcf: grid or index range exceeds 32-bit keys";
    case LCF_ERR_LAUNCH: return "lcf: HIP launch error";
    case LCF_ERR_STATIC_LIMIT:
        return "lcf: input exceeds the static scratch array of the no-allocation entry point "
               "(LCF_VOXELIZE_STATIC_BYTES); use lcf_voxelize with a caller workspace";
    default: return "lcf: unknown error";
    }
}

extern "C" int lcf_version(void) { return 100; }

__global__ void __launch_bounds__(256) lcf_fill32_kernel(uint32_t *p, uint32_t v, size_t n32)
{
    // 16-byte stores over the aligned body, scalar head/tail
    size_t head = ((16 - ((uintptr_t)p & 15)) & 15) / 4;
    if (head > n32) head = n32;
    size_t body = (n32 - head) / 4;
    uint4 *p4 = reinterpret_cast<uint4 *>(p + head);
    const uint4 v4 = make_uint4(v, v, v, v);
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = t; i < body; i += stride) p4[i] = v4;
    if (t < head) p[t] = v;
    size_t tail0 = head + body * 4;
    if (t < n32 - tail0) p[tail0 + t] = v;
}

// ---------------------------------------------------------------- contexts and options
#include <atomic>
#include <new>

thread_local const LcfOpts *t_lcf_opts = nullptr;
static std::atomic<int> g_def_split{0}, g_def_impl{0}, g_def_tile{0}, g_def_sets{0};

LcfOpts lcf_default_opts()
{
    LcfOpts o{};
    o.conv_split = g_def_split.load(std::memory_order_relaxed);
    o.conv_impl = g_def_impl.load(std::memory_order_relaxed);
    o.conv_tile = g_def_tile.load(std::memory_order_relaxed);
    o.conv_sets = g_def_sets.load(std::memory_order_relaxed);
    return o;
}

static int opt_store(int option, int value, int *split, int *impl, int *tile, int *sets)
{
    switch (option) {
    case LCF_OPT_CONV_SPLIT:
        if (value != 0 && value != 1 && value != 2 && value != 4) return LCF_ERR_ARG;
        *split = value;
        return LCF_OK;
    case LCF_OPT_CONV_IMPL:
        if (value < 0 || value > 5) return LCF_ERR_ARG;
        *impl = value;
        return LCF_OK;
    case LCF_OPT_CONV_TILE:
        if (value != 0 && (value < 16 || value > 32)) return LCF_ERR_ARG;
        *tile = value;
        return LCF_OK;
    case LCF_OPT_CONV_SETS:
        if (value < 0 || value > 4) return LCF_ERR_ARG;
        *sets = value;
        return LCF_OK;
    default: return LCF_ERR_ARG;
    }
}

// Process-wide DEFAULTS of the four kernel-selection knobs only (benchmarks, tests).  The options
// that change what a call assumes about its buffers (WS_CLEAN, NBR_CLEAN) and the rows hint are
// per context: a process-wide switch for them would change the behaviour of other threads' calls.
extern "C" int lcf_set_option(int option, int value)
{
    int split = g_def_split, impl = g_def_impl, tile = g_def_tile, sets = g_def_sets;
    int rc = opt_store(option, value, &split, &impl, &tile, &sets);
    if (rc != LCF_OK) return rc;
    g_def_split = split;
    g_def_impl = impl;
    g_def_tile = tile;
    g_def_sets = sets;
    return LCF_OK;
}

extern "C" int lcf_ctx_create(lcf_ctx **out)
{
    if (!out) return LCF_ERR_ARG;
    lcf_ctx *c = new (std::nothrow) lcf_ctx();
    if (!c) return LCF_ERR_ARG;
    c->plan.clr.blk0 = 0xFFFFFFFFu; // no clear record
    *out = c;
    return LCF_OK;
}

extern "C" void lcf_ctx_destroy(lcf_ctx *ctx) { delete ctx; }

extern "C" int lcf_ctx_set_option(lcf_ctx *ctx, int option, int value)
{
    if (!ctx) return LCF_ERR_ARG;
    switch (option) {
    case LCF_OPT_WS_CLEAN:
        if (value != 0 && value != 1) return LCF_ERR_ARG;
        ctx->o.ws_clean = value;
        return LCF_OK;
    case LCF_OPT_NBR_CLEAN:
        if (value != 0 && value != 1) return LCF_ERR_ARG;
        ctx->o.nbr_clean = value;
        return LCF_OK;
    case LCF_OPT_CONV_ROWS_HINT:
        if (value < 0) return LCF_ERR_ARG;
        ctx->o.rows_hint = value;
        return LCF_OK;
    case LCF_OPT_RB_UNFUSED:
        if (value != 0 && value != 1) return LCF_ERR_ARG;
        ctx->o.rb_unfused = value;
        return LCF_OK;
    default:
        return opt_store(option, value, &ctx->o.conv_split, &ctx->o.conv_impl, &ctx->o.conv_tile,
                         &ctx->o.conv_sets);
    }
}

// Batched fills: between lcf_ctx_fill_batch_begin() and lcf_ctx_fill_batch_flush() every fill the
// library would launch FOR THIS CONTEXT (workspace resets, lcf_fill_u32) is recorded in the
// context instead, and the flush runs them as ONE kernel -- the engine re-arms ten hash
// workspaces and clears the dense output behind a frame's rulebook branch: eleven 4-5 us launches
// there become one.
// Rows-only clear of a dense() output (no 18 MB zero fill per frame).
// `prev` [cap,4] holds the coordinates the PREVIOUS frame's rows were scattered to (-1 = no row);
// a workgroup zeroes the cells of its four rows in every channel plane, then replaces its rows of
// `prev` by this frame's coordinates (cur, rows < k; -1 beyond).  Rows belong to one workgroup, so
// nothing crosses workgroups and no count of the previous frame is needed.
#define LCF_CLEAR_ROWS_PER_WG 4
__device__ __forceinline__ void lcf_clear_rows_role(const LcfClearSeg &q, unsigned blk)
{
    const int row = (int)blk * LCF_CLEAR_ROWS_PER_WG + (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int4 pc = make_int4(-1, -1, -1, -1);
    if (row < q.cap) pc = reinterpret_cast<const int4 *>(q.prev)[row];
    if (pc.x >= 0 && pc.x < q.batch && pc.y >= 0 && pc.y < q.D && pc.z >= 0 && pc.z < q.H &&
        pc.w >= 0 && pc.w < q.W) {
        const size_t svol = (size_t)q.D * q.H * q.W;
        const size_t sp = ((size_t)pc.y * q.H + pc.z) * q.W + pc.w;
        float *base = q.dense + (size_t)pc.x * q.c * svol + sp;
        for (int ch = lane; ch < q.c; ch += 64) base[(size_t)ch * svol] = 0.f;
    }
    __syncthreads();
    if (lane == 0 && row < q.cap) {
        const int k = lcf_count(q.cap, q.k_dev);
        int4 nc = make_int4(-1, -1, -1, -1);
        if (row < k) nc = reinterpret_cast<const int4 *>(q.cur)[row];
        reinterpret_cast<int4 *>(q.prev)[row] = nc;
    }
}

__global__ void __launch_bounds__(256) lcf_clear_rows_kernel(LcfClearSeg q)
{
    lcf_clear_rows_role(q, blockIdx.x);
}

__global__ void __launch_bounds__(256) lcf_fill32_multi_kernel(LcfFillPlan plan)
{
    if (blockIdx.x >= plan.clr.blk0) {
        lcf_clear_rows_role(plan.clr, blockIdx.x - plan.clr.blk0);
        return;
    }
    int i = 0;
#pragma unroll 1
    for (int j = 1; j < plan.n; ++j)
        if (blockIdx.x >= plan.s[j].blk0) i = j;
    const LcfFillSeg sg = plan.s[i];
    const unsigned fill_end = plan.clr.blk0 < gridDim.x ? plan.clr.blk0 : gridDim.x;
    const unsigned nblk = (i + 1 < plan.n ? plan.s[i + 1].blk0 : fill_end) - sg.blk0;
    uint32_t *p = sg.p;
    const size_t n32 = sg.n32;
    size_t head = ((16 - ((uintptr_t)p & 15)) & 15) / 4;
    if (head > n32) head = n32;
    const size_t body = (n32 - head) / 4;
    uint4 *p4 = reinterpret_cast<uint4 *>(p + head);
    const uint4 v4 = make_uint4(sg.v, sg.v, sg.v, sg.v);
    const size_t stride = (size_t)nblk * 256;
    const size_t t = (size_t)(blockIdx.x - sg.blk0) * 256 + threadIdx.x;
    for (size_t e = t; e < body; e += stride) p4[e] = v4;
    if (t < head) p[t] = sg.v;
    const size_t tail0 = head + body * 4;
    if (t < n32 - tail0) p[tail0 + t] = sg.v;
}

static unsigned clear_rows_blocks(int cap)
{
    return (unsigned)((cap + LCF_CLEAR_ROWS_PER_WG - 1) / LCF_CLEAR_ROWS_PER_WG);
}

static int fill_plan_launch(lcf_ctx *ctx, hipStream_t st)
{
    const bool clr = ctx->plan.clr.blk0 != 0xFFFFFFFFu;
    if (ctx->plan.n == 0 && !clr) return LCF_OK;
    unsigned blocks = ctx->fill_blocks;
    if (clr) { // the clear record's workgroups go behind the fills'
        ctx->plan.clr.blk0 = blocks;
        blocks += clear_rows_blocks(ctx->plan.clr.cap);
    }
    lcf_fill32_multi_kernel<<<blocks, 256, 0, st>>>(ctx->plan);
    ctx->plan.n = 0;
    ctx->plan.clr.blk0 = 0xFFFFFFFFu;
    ctx->fill_blocks = 0;
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_ctx_fill_batch_begin(lcf_ctx *ctx)
{
    if (!ctx) return LCF_ERR_ARG;
    ctx->plan.n = 0;
    ctx->plan.clr.blk0 = 0xFFFFFFFFu;
    ctx->fill_blocks = 0;
    ctx->batching = 1;
    return LCF_OK;
}

extern "C" int lcf_ctx_fill_batch_flush(lcf_ctx *ctx, void *stream)
{
    if (!ctx) return LCF_ERR_ARG;
    ctx->batching = 0;
    return fill_plan_launch(ctx, (hipStream_t)stream);
}

int lcf_fill32(void *p, uint32_t v, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return LCF_OK;
    if ((bytes & 3) || ((uintptr_t)p & 3)) return LCF_ERR_ARG;
    size_t n32 = bytes / 4;
    size_t nb = (n32 / 4 + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 4096) nb = 4096;
    lcf_fill32_kernel<<<(unsigned)nb, 256, 0, st>>>((uint32_t *)p, v, n32);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

int lcf_fill32_ctx(lcf_ctx *ctx, void *p, uint32_t v, size_t bytes, hipStream_t st)
{
    if (!ctx || !ctx->batching) return lcf_fill32(p, v, bytes, st);
    if (bytes == 0) return LCF_OK;
    if ((bytes & 3) || ((uintptr_t)p & 3)) return LCF_ERR_ARG;
    if (ctx->plan.n == LCF_FILL_MAX_SEGS) { // full: run the recorded fills (a clear record stays)
        LcfClearSeg keep = ctx->plan.clr;
        ctx->plan.clr.blk0 = 0xFFFFFFFFu;
        int rc = fill_plan_launch(ctx, st);
        ctx->plan.clr = keep;
        if (rc != LCF_OK) return rc;
    }
    const size_t n32 = bytes / 4;
    size_t nbs = (n32 / 4 + 255) / 256;
    if (nbs < 1) nbs = 1;
    if (nbs > 2048) nbs = 2048;
    LcfFillSeg &sg = ctx->plan.s[ctx->plan.n++];
    sg.p = (uint32_t *)p;
    sg.v = v;
    sg.blk0 = ctx->fill_blocks;
    sg.n32 = n32;
    ctx->fill_blocks += (unsigned)nbs;
    return LCF_OK;
}

extern "C" int lcf_fill_u32(void *p, uint32_t value, size_t bytes, void *stream, lcf_ctx *ctx)
{
    if (!p && bytes) return LCF_ERR_ARG;
    return lcf_fill32_ctx(ctx, p, value, bytes, (hipStream_t)stream);
}

extern "C" int lcf_dense_clear_rows(float *dense, int *prev_coors, const int *coors, int cap,
                                    const int *k_dev, int c, int batch, const int shape[3],
                                    void *stream, lcf_ctx *ctx)
{
    if (!dense || !prev_coors || !coors || cap <= 0 || c <= 0 || batch <= 0 || !shape ||
        shape[0] <= 0 || shape[1] <= 0 || shape[2] <= 0)
        return LCF_ERR_ARG;
    LcfClearSeg q;
    q.dense = dense;
    q.prev = prev_coors;
    q.cur = coors;
    q.k_dev = k_dev;
    q.cap = cap;
    q.c = c;
    q.batch = batch;
    q.D = shape[0];
    q.H = shape[1];
    q.W = shape[2];
    q.blk0 = 0;
    if (ctx && ctx->batching) {
        if (ctx->plan.clr.blk0 != 0xFFFFFFFFu) return LCF_ERR_UNSUPPORTED; // one record per batch
        ctx->plan.clr = q;
        return LCF_OK;
    }
    lcf_clear_rows_kernel<<<clear_rows_blocks(cap), 256, 0, (hipStream_t)stream>>>(q);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

// ---------------------------------------------------------------- frame hand-over
// One launch in front of a captured frame: the new points into the engine's static point buffer
// (dst = where they go: the buffer's start, or behind the cloud already there) and the total
// point count into the device-side counter (instead of a copy and a fill).
__global__ void __launch_bounds__(256)
load_points_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, long long n4,
                   int n_total, int *__restrict__ n_dev)
{
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t == 0) n_dev[0] = n_total;
    if (t < n4) dst[t] = src[t];
}

extern "C" int lcf_load_points(const float *points, int n, int f, float *dst, int n_total,
                               int *n_dev, void *stream)
{
    if (n < 0 || f <= 0 || n_total < n || !dst || !n_dev || (n > 0 && !points))
        return LCF_ERR_ARG;
    const long long elems = (long long)n * f;
    if ((elems & 3) || ((uintptr_t)points & 15) || ((uintptr_t)dst & 15))
        return LCF_ERR_UNSUPPORTED; // float4 copies: n * f a multiple of 4, 16-byte aligned
    const long long n4 = elems / 4;
    const unsigned nb = (unsigned)((n4 + 255) / 256);
    load_points_kernel<<<nb ? nb : 1, 256, 0, (hipStream_t)stream>>>(
        (const float4 *)points, (float4 *)dst, n4, n_total, n_dev);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

// ---------------------------------------------------------------- dense
// SparseConvTensor.dense() (second/pytorch/models/middle.py:206): [B,C,D,H,W] contiguous.
// Thread = (row, channel); a row's C values land C*svol*4 bytes apart, so writes are scattered by
// construction -- the bulk traffic is the zero fill, done by a 16-byte-store fill kernel.
__global__ void __launch_bounds__(256)
dense_scatter_kernel(const float *__restrict__ feat, const int *__restrict__ coors, int k_cap,
                     const int *__restrict__ k_dev, int c, int batch, int D, int H, int W,
                     float *__restrict__ out)
{
    const int k = lcf_count(k_cap, k_dev);
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)k * c) return;
    int row = (int)(t / c), ch = (int)(t - (long long)row * c);
    int4 q = reinterpret_cast<const int4 *>(coors)[row];
    if (q.x < 0 || q.x >= batch || q.y < 0 || q.y >= D || q.z < 0 || q.z >= H || q.w < 0 ||
        q.w >= W)
        return;
    size_t svol = (size_t)D * H * W;
    size_t sp = ((size_t)q.y * H + q.z) * W + q.w;
    out[((size_t)q.x * c + ch) * svol + sp] = feat[t];
}

__global__ void __launch_bounds__(256)
dense_gather_kernel(const float *__restrict__ dout, const int *__restrict__ coors, int k, int c,
                    int batch, int D, int H, int W, float *__restrict__ dfeat)
{
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)k * c) return;
    int row = (int)(t / c), ch = (int)(t - (long long)row * c);
    int4 q = reinterpret_cast<const int4 *>(coors)[row];
    float v = 0.f;
    if (q.x >= 0 && q.x < batch && q.y >= 0 && q.y < D && q.z >= 0 && q.z < H && q.w >= 0 &&
        q.w < W) {
        size_t svol = (size_t)D * H * W;
        size_t sp = ((size_t)q.y * H + q.z) * W + q.w;
        v = dout[((size_t)q.x * c + ch) * svol + sp];
    }
    dfeat[t] = v;
}

extern "C" int lcf_dense(const float *feat, const int *coors, int k, const int *k_dev, int c,
                         int batch, const int shape[3], int zero_fill, float *out, void *stream)
{
    if (k < 0 || c <= 0 || batch <= 0 || !shape || !out || (k > 0 && (!feat || !coors)))
        return LCF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    size_t bytes = (size_t)batch * c * shape[0] * shape[1] * shape[2] * 4;
    if (zero_fill && lcf_fill32(out, 0u, bytes, st) != LCF_OK) return LCF_ERR_LAUNCH;
    if (k == 0) return LCF_OK;
    long long total = (long long)k * c;
    dense_scatter_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(
        feat, coors, k, k_dev, c, batch, shape[0], shape[1], shape[2], out);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_dense_bwd(const float *dout, const int *coors, int k, int c, int batch,
                             const int shape[3], float *dfeat, void *stream)
{
    if (k < 0 || c <= 0 || batch <= 0 || !shape || !dout || (k > 0 && (!dfeat || !coors)))
        return LCF_ERR_ARG;
    if (k == 0) return LCF_OK;
    long long total = (long long)k * c;
    dense_gather_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        dout, coors, k, c, batch, shape[0], shape[1], shape[2], dfeat);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

// dense() in the layout the RPN's first convolution reads under bf16 autocast: the
// [B, C*D, H, W] view of middle.py:207-209 as channels_last bf16, i.e. memory [B, H, W, C*D] with
// element (b, c, z, y, x) at ((b*H + y)*W + x)*(C*D) + c*D + z.  One pass instead of fp32 NCDHW
// scatter + layout copy + dtype copy (and half the zero fill); the two z-rows of a pillar fill
// one 2*C*D-byte run.  Rounding: round-to-nearest-even, as torch's float -> bfloat16 copy.
template <typename TF>
__global__ void __launch_bounds__(256)
dense_nhwc_bf16_kernel(const TF *__restrict__ feat, const int *__restrict__ coors, int k_cap,
                       const int *__restrict__ k_dev, int c, int batch, int D, int H, int W,
                       __bf16 *__restrict__ out)
{
    const int k = lcf_count(k_cap, k_dev);
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)k * c) return;
    int row = (int)(t / c), ch = (int)(t - (long long)row * c);
    int4 q = reinterpret_cast<const int4 *>(coors)[row];
    if (q.x < 0 || q.x >= batch || q.y < 0 || q.y >= D || q.z < 0 || q.z >= H || q.w < 0 ||
        q.w >= W)
        return;
    size_t pix = ((size_t)q.x * H + q.z) * W + q.w;
    out[pix * ((size_t)c * D) + (size_t)ch * D + q.y] = (__bf16)(float)feat[t];
}

template <typename TF>
__global__ void __launch_bounds__(256)
dense_nhwc_bf16_gather_kernel(const __bf16 *__restrict__ dout, const int *__restrict__ coors,
                              int k, int c, int batch, int D, int H, int W,
                              TF *__restrict__ dfeat)
{
    long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)k * c) return;
    int row = (int)(t / c), ch = (int)(t - (long long)row * c);
    int4 q = reinterpret_cast<const int4 *>(coors)[row];
    float v = 0.f;
    if (q.x >= 0 && q.x < batch && q.y >= 0 && q.y < D && q.z >= 0 && q.z < H && q.w >= 0 &&
        q.w < W) {
        size_t pix = ((size_t)q.x * H + q.z) * W + q.w;
        v = (float)dout[pix * ((size_t)c * D) + (size_t)ch * D + q.y];
    }
    dfeat[t] = (TF)v;
}

static int dense_nhwc_run(const void *feat, int feat_bf16, const int *coors, int k,
                          const int *k_dev, int c, int batch, const int shape[3], int zero_fill,
                          void *out_bf16, void *stream);

extern "C" int lcf_dense_nhwc_bf16(const float *feat, const int *coors, int k, const int *k_dev,
                                   int c, int batch, const int shape[3], int zero_fill,
                                   void *out_bf16, void *stream)
{
    return dense_nhwc_run(feat, 0, coors, k, k_dev, c, batch, shape, zero_fill, out_bf16, stream);
}

extern "C" int lcf_dense_nhwc_bf16_rows(const void *feat, int feat_bf16, const int *coors, int k,
                                        const int *k_dev, int c, int batch, const int shape[3],
                                        int zero_fill, void *out_bf16, void *stream)
{
    return dense_nhwc_run(feat, feat_bf16, coors, k, k_dev, c, batch, shape, zero_fill, out_bf16,
                          stream);
}

static int dense_nhwc_run(const void *feat, int feat_bf16, const int *coors, int k,
                          const int *k_dev, int c, int batch, const int shape[3], int zero_fill,
                          void *out_bf16, void *stream)
{
    if (k < 0 || c <= 0 || batch <= 0 || !shape || !out_bf16 || (k > 0 && (!feat || !coors)))
        return LCF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    size_t bytes = (size_t)batch * c * shape[0] * shape[1] * shape[2] * 2;
    if (bytes & 3) return LCF_ERR_UNSUPPORTED; // the fill works on 32-bit words
    if (zero_fill && lcf_fill32(out_bf16, 0u, bytes, st) != LCF_OK) return LCF_ERR_LAUNCH;
    if (k == 0) return LCF_OK;
    long long total = (long long)k * c;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (feat_bf16)
        dense_nhwc_bf16_kernel<__bf16><<<grid, 256, 0, st>>>(
            (const __bf16 *)feat, coors, k, k_dev, c, batch, shape[0], shape[1], shape[2],
            (__bf16 *)out_bf16);
    else
        dense_nhwc_bf16_kernel<float><<<grid, 256, 0, st>>>(
            (const float *)feat, coors, k, k_dev, c, batch, shape[0], shape[1], shape[2],
            (__bf16 *)out_bf16);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_dense_nhwc_bf16_bwd_rows(const void *dout_bf16, const int *coors, int k, int c,
                                            int batch, const int shape[3], void *dfeat,
                                            int dfeat_bf16, void *stream)
{
    if (k < 0 || c <= 0 || batch <= 0 || !shape || !dout_bf16 || (k > 0 && (!dfeat || !coors)))
        return LCF_ERR_ARG;
    if (k == 0) return LCF_OK;
    long long total = (long long)k * c;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (dfeat_bf16)
        dense_nhwc_bf16_gather_kernel<__bf16><<<grid, 256, 0, (hipStream_t)stream>>>(
            (const __bf16 *)dout_bf16, coors, k, c, batch, shape[0], shape[1], shape[2],
            (__bf16 *)dfeat);
    else
        dense_nhwc_bf16_gather_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(
            (const __bf16 *)dout_bf16, coors, k, c, batch, shape[0], shape[1], shape[2],
            (float *)dfeat);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_dense_nhwc_bf16_bwd(const void *dout_bf16, const int *coors, int k, int c,
                                       int batch, const int shape[3], float *dfeat, void *stream)
{
    return lcf_dense_nhwc_bf16_bwd_rows(dout_bf16, coors, k, c, batch, shape, dfeat, 0, stream);
}

// ---------------------------------------------------------------- VFE
// SimpleVoxel.forward (voxel_encoder.py:220-225): sum over T (padding rows are zero) / num.
// The sum runs t = 0..T-1 in order, as torch's reduction over a 5-long dim does.
__global__ void __launch_bounds__(256)
vfe_mean_kernel(const float *__restrict__ voxels, const int *__restrict__ num, int k_cap,
                const int *__restrict__ k_dev, int t, int f, int c, float *__restrict__ out)
{
    const int k = lcf_count(k_cap, k_dev);
    long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)k * c) return;
    int v = (int)(id / c), ch = (int)(id - (long long)v * c);
    const float *p = voxels + (size_t)v * t * f + ch;
    float s = 0.f;
    for (int i = 0; i < t; ++i) s += p[(size_t)i * f];
    out[id] = __fdiv_rn(s, (float)num[v]);
}

// VoxelFeatureExtractor.forward :116-124: out[k,t,:] = [voxels[k,t,:f], voxels[k,t,:3]-mean]
__global__ void __launch_bounds__(256)
vfe_augment_kernel(const float *__restrict__ voxels, const int *__restrict__ num, int k_cap,
                   const int *__restrict__ k_dev, int t, int f, float *__restrict__ out)
{
    const int k = lcf_count(k_cap, k_dev);
    long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)k * t) return;
    int v = (int)(id / t);
    const float *base = voxels + (size_t)v * t * f;
    const float *p = base + (size_t)(id - (long long)v * t) * f;
    float *o = out + (size_t)id * (f + 3);
    float n = (float)num[v];
    for (int e = 0; e < f; ++e) o[e] = p[e];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        float s = 0.f;
        for (int i = 0; i < t; ++i) s += base[(size_t)i * f + e];
        o[f + e] = p[e] - __fdiv_rn(s, n);
    }
}

// Tail of VFELayer.forward (:69-80) + mask (:130), or the final masked max (:134-139).
// Thread = (voxel, channel); walks the T points of its voxel (stride u floats: the 64 lanes of a
// wave read 64 consecutive channels = 256 contiguous bytes per point row).
__global__ void __launch_bounds__(256)
vfe_pool_kernel(const float *__restrict__ x, const int *__restrict__ num, int k_cap,
                const int *__restrict__ k_dev, int t, int u, const float *__restrict__ scale,
                const float *__restrict__ shift, int mode, float *__restrict__ out)
{
    const int k = lcf_count(k_cap, k_dev);
    long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)k * u) return;
    int v = (int)(id / u), ch = (int)(id - (long long)v * u);
    const float sc = scale ? scale[ch] : 1.f;
    const float sh = shift ? shift[ch] : 0.f;
    const int n = num[v];
    const float *p = x + (size_t)v * t * u + ch;
    if (mode == 0) {
        float mx = -INFINITY;
        for (int i = 0; i < t; ++i) {
            float y = fmaxf(p[(size_t)i * u] * sc + sh, 0.f);
            mx = fmaxf(mx, y);
        }
        float *o = out + (size_t)v * t * 2 * u;
        for (int i = 0; i < t; ++i) {
            float y = fmaxf(p[(size_t)i * u] * sc + sh, 0.f);
            float m = i < n ? 1.f : 0.f;
            o[(size_t)i * 2 * u + ch] = y * m;
            o[(size_t)i * 2 * u + u + ch] = mx * m;
        }
    } else {
        float mx = -INFINITY;
        for (int i = 0; i < t; ++i) {
            float y = fmaxf(p[(size_t)i * u] * sc + sh, 0.f);
            y = i < n ? y : 0.f;
            mx = fmaxf(mx, y);
        }
        out[id] = mx;
    }
}

extern "C" int lcf_vfe_mean(const float *voxels, const int *num, int k, const int *k_dev, int t,
                            int f, int c, float *out, void *stream)
{
    if (k < 0 || t <= 0 || f <= 0 || c <= 0 || c > f || (k > 0 && (!voxels || !num || !out)))
        return LCF_ERR_ARG;
    if (k == 0) return LCF_OK;
    long long total = (long long)k * c;
    vfe_mean_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        voxels, num, k, k_dev, t, f, c, out);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_vfe_augment(const float *voxels, const int *num, int k, const int *k_dev,
                               int t, int f, float *out, void *stream)
{
    if (k < 0 || t <= 0 || f < 3 || (k > 0 && (!voxels || !num || !out))) return LCF_ERR_ARG;
    if (k == 0) return LCF_OK;
    long long total = (long long)k * t;
    vfe_augment_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        voxels, num, k, k_dev, t, f, out);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_vfe_pool(const float *x, const int *num, int k, const int *k_dev, int t,
                            int u, const float *scale, const float *shift, int mode, float *out,
                            void *stream)
{
    if (k < 0 || t <= 0 || u <= 0 || mode < 0 || mode > 1 || (k > 0 && (!x || !num || !out)))
        return LCF_ERR_ARG;
    if (k == 0) return LCF_OK;
    long long total = (long long)k * u;
    vfe_pool_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        x, num, k, k_dev, t, u, scale, shift, mode, out);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}
