// bn.hip -- training-mode BatchNorm1d + ReLU over the active voxels of a sparse tensor, the pair
// that sits between every two sparse convs of the middle extractor
// (second/pytorch/models/middle.py:124-125,147-148: BatchNorm1d(eps=1e-3, momentum=0.01), ReLU;
// SURVEY.md section 8 row a13).  Four HBM-streaming kernels over rows [K, C] fp32:
//   forward   stats partials -> (finalise +) y = relu(x * scale + shift)
//   backward  reduce partials of g = dy * (y > 0): sum g, sum g * xhat -> (finalise +) dx
// Per-channel sums are taken over a FIXED partition of the rows in a fixed order (per-thread
// registers -> LDS tree -> partials in HBM -> double-precision sum of the partials), so results
// are run-to-run bit-identical; a one-wave-per-column kernel sums the partials in between.
#include "common.h"

#define BN_BLOCK 256
#define BN_MAX_C 128
#define BN_MAX_PARTS 512

// thread -> (row lane rl, float4 column group c4); a block walks rows rl, rl + RL, ... of its chunk
struct BnMap {
    int c4n; // float4 groups per row
    int rl;  // row lane of this thread
    int c4;  // column group of this thread
    int RL;  // row lanes per block
};

__device__ __forceinline__ BnMap bn_map(int c)
{
    BnMap m;
    m.c4n = c >> 2;
    m.RL = BN_BLOCK / m.c4n;
    m.rl = threadIdx.x / m.c4n;
    m.c4 = threadIdx.x % m.c4n;
    return m;
}

// Rows are fp32 or bf16 (round 3: the activations a sparse conv GATHERS -- the output y of this
// pair and the gradient dx it hands back -- can be stored as bf16 so that the gathers move
// 128-byte rows; statistics, sums and the arithmetic stay fp32 / double).  T = float | __bf16.
template <typename T>
__device__ __forceinline__ float4 bn_ld4(const T *row, int c4)
{
    if constexpr (sizeof(T) == 4) {
        return reinterpret_cast<const float4 *>(row)[c4];
    } else {
        const uint2 u = reinterpret_cast<const uint2 *>(row)[c4];
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
    }
}

template <typename T>
__device__ __forceinline__ void bn_st4(T *row, int c4, float4 v)
{
    if constexpr (sizeof(T) == 4) {
        reinterpret_cast<float4 *>(row)[c4] = v;
    } else {
        uint2 u;
        u.x = lcf_bf16_rne(v.x) | (lcf_bf16_rne(v.y) << 16);
        u.y = lcf_bf16_rne(v.z) | (lcf_bf16_rne(v.w) << 16);
        reinterpret_cast<uint2 *>(row)[c4] = u;
    }
}

// block-level sum of two float4 per thread over the row lanes -> part[0:c] and part[c:2c]
__device__ __forceinline__ void bn_block_reduce(const BnMap &m, int c, float4 a, float4 b,
                                                float *lds, float *part)
{
    // lds [RL][2c]
    float4 *l4 = reinterpret_cast<float4 *>(lds);
    const bool live = m.rl < m.RL;
    if (live) {
        l4[m.rl * (2 * m.c4n) + m.c4] = a;
        l4[m.rl * (2 * m.c4n) + m.c4n + m.c4] = b;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * c; e += BN_BLOCK) {
        float s = 0.f;
        for (int r = 0; r < m.RL; ++r) s += lds[r * 2 * c + e];
        part[e] = s;
    }
}

// ---- forward 1: partial sums of x and x^2
template <typename TX>
__global__ void __launch_bounds__(BN_BLOCK)
bn_stats_kernel(const TX *__restrict__ x, int k, int c, int rows_per_block,
                float *__restrict__ parts)
{
    __shared__ __align__(16) float lds[BN_BLOCK * 8];
    const BnMap m = bn_map(c);
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(k, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (m.rl < m.RL) {
        for (int r = r0 + m.rl; r < r1; r += m.RL) {
            const float4 v = bn_ld4(x + (size_t)r * c, m.c4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
        }
    }
    bn_block_reduce(m, c, s, q, lds, parts + (size_t)blockIdx.x * 2 * c);
}

// ---- sums[e] = sum over the partials of column e (double); one wave per column
__global__ void __launch_bounds__(64)
bn_finalize_kernel(const float *__restrict__ parts, int nparts, int c2, double *__restrict__ sums)
{
    const int e = blockIdx.x;
    double s = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 64) s += (double)parts[(size_t)p * c2 + e];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if (threadIdx.x == 0) sums[e] = s;
}

// ---- forward 2: mean / biased var from the partials, y = relu(x * scale + shift)
// running statistics of nn.BatchNorm1d, updated by block 0 when the caller passes them (what the
// module does after its forward: unbiased variance, momentum; five tiny launches otherwise)
struct BnRunning {
    float *mean, *var;
    long long *batches;
    float momentum;
};

template <typename TX, typename TY>
__global__ void __launch_bounds__(BN_BLOCK)
bn_apply_relu_kernel(const TX *__restrict__ x, int k, int c, int rows_per_block,
                     const double *__restrict__ sums,
                     const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                     int relu, TY *__restrict__ y, float *__restrict__ stats, BnRunning run)
{
    __shared__ __align__(16) float sc[BN_MAX_C], sh[BN_MAX_C];
    for (int e = threadIdx.x; e < c; e += BN_BLOCK) {
        const double mean = sums[e] / k;
        double var = sums[c + e] / k - mean * mean;
        if (var < 0.0) var = 0.0;
        const float inv = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[e] : 1.f;
        sc[e] = g * inv;
        sh[e] = (beta ? beta[e] : 0.f) - (float)mean * g * inv;
        if (blockIdx.x == 0) {
            stats[e] = (float)mean;
            stats[c + e] = (float)var;
            stats[2 * c + e] = inv;
            if (run.mean) {
                const float mo = run.momentum;
                run.mean[e] = (1.f - mo) * run.mean[e] + mo * (float)mean;
                run.var[e] = (1.f - mo) * run.var[e] +
                             mo * (float)(var * ((double)k / (double)(k > 1 ? k - 1 : 1)));
                if (e == 0 && run.batches) run.batches[0] += 1;
            }
        }
    }
    __syncthreads();
    const BnMap m = bn_map(c);
    if (m.rl >= m.RL) return;
    const float4 a = reinterpret_cast<const float4 *>(sc)[m.c4];
    const float4 b = reinterpret_cast<const float4 *>(sh)[m.c4];
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(k, r0 + rows_per_block);
    for (int r = r0 + m.rl; r < r1; r += m.RL) {
        float4 v = bn_ld4(x + (size_t)r * c, m.c4);
        v.x = v.x * a.x + b.x; v.y = v.y * a.y + b.y;
        v.z = v.z * a.z + b.z; v.w = v.w * a.w + b.w;
        if (relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        bn_st4(y + (size_t)r * c, m.c4, v);
    }
}

// ---- backward 1: partial sums of g = dy * (y > 0) and of g * xhat, xhat = (x - mean) * inv
template <typename TX, typename TY, typename TG>
__global__ void __launch_bounds__(BN_BLOCK)
bn_bwd_reduce_kernel(const TX *__restrict__ x, const TY *__restrict__ y,
                     const TG *__restrict__ dy, int k, int c, int rows_per_block,
                     const float *__restrict__ stats, int relu, float *__restrict__ parts)
{
    __shared__ __align__(16) float lds[BN_BLOCK * 8];
    const BnMap m = bn_map(c);
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(k, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (m.rl < m.RL) {
        const float4 mean = reinterpret_cast<const float4 *>(stats)[m.c4];
        const float4 inv = reinterpret_cast<const float4 *>(stats + 2 * c)[m.c4];
        for (int r = r0 + m.rl; r < r1; r += m.RL) {
            const float4 xv = bn_ld4(x + (size_t)r * c, m.c4);
            float4 g = bn_ld4(dy + (size_t)r * c, m.c4);
            if (relu) {
                const float4 yv = bn_ld4(y + (size_t)r * c, m.c4);
                g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
                g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
            }
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            q.x += g.x * (xv.x - mean.x) * inv.x; q.y += g.y * (xv.y - mean.y) * inv.y;
            q.z += g.z * (xv.z - mean.z) * inv.z; q.w += g.w * (xv.w - mean.w) * inv.w;
        }
    }
    bn_block_reduce(m, c, s, q, lds, parts + (size_t)blockIdx.x * 2 * c);
}

// ---- backward 2: dx = gamma * inv * (g - mean(g) - xhat * mean(g * xhat)); dgamma, dbeta
template <typename TX, typename TY, typename TG, typename TD>
__global__ void __launch_bounds__(BN_BLOCK)
bn_bwd_apply_kernel(const TX *__restrict__ x, const TY *__restrict__ y,
                    const TG *__restrict__ dy, int k, int c, int rows_per_block,
                    const float *__restrict__ stats, const double *__restrict__ sums,
                    const float *__restrict__ gamma, int relu, TD *__restrict__ dx,
                    float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    __shared__ __align__(16) float mg[BN_MAX_C], mgx[BN_MAX_C], gi[BN_MAX_C];
    for (int e = threadIdx.x; e < c; e += BN_BLOCK) {
        const double sg = sums[e];
        const double sgx = sums[c + e];
        mg[e] = (float)(sg / k);
        mgx[e] = (float)(sgx / k);
        gi[e] = (gamma ? gamma[e] : 1.f) * stats[2 * c + e];
        if (blockIdx.x == 0) {
            if (dgamma) dgamma[e] = (float)sgx;
            if (dbeta) dbeta[e] = (float)sg;
        }
    }
    __syncthreads();
    const BnMap m = bn_map(c);
    if (m.rl >= m.RL) return;
    const float4 mean = reinterpret_cast<const float4 *>(stats)[m.c4];
    const float4 inv = reinterpret_cast<const float4 *>(stats + 2 * c)[m.c4];
    const float4 a = reinterpret_cast<const float4 *>(mg)[m.c4];
    const float4 b = reinterpret_cast<const float4 *>(mgx)[m.c4];
    const float4 w = reinterpret_cast<const float4 *>(gi)[m.c4];
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(k, r0 + rows_per_block);
    for (int r = r0 + m.rl; r < r1; r += m.RL) {
        const float4 xv = bn_ld4(x + (size_t)r * c, m.c4);
        float4 g = bn_ld4(dy + (size_t)r * c, m.c4);
        if (relu) {
            const float4 yv = bn_ld4(y + (size_t)r * c, m.c4);
            g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
            g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
        }
        float4 o;
        o.x = w.x * (g.x - a.x - (xv.x - mean.x) * inv.x * b.x);
        o.y = w.y * (g.y - a.y - (xv.y - mean.y) * inv.y * b.y);
        o.z = w.z * (g.z - a.z - (xv.z - mean.z) * inv.z * b.z);
        o.w = w.w * (g.w - a.w - (xv.w - mean.w) * inv.w * b.w);
        bn_st4(dx + (size_t)r * c, m.c4, o);
    }
}

// ---- all-bf16 rows, eight channels (16 bytes) per thread: the dense RPN's BatchNorm2d + ReLU on
// channels_last activations (rows = pixels, C = 128: 72 MB per tensor at batch 8).  Same
// partition, partials and finalise as above; the backward takes the ReLU mask from the sign of
// the recomputed output instead of reading y (one 72 MB read less in each of its two passes).
struct Bn8 {
    float v[8];
};

__device__ __forceinline__ Bn8 bn_ld8(const __bf16 *row, int c8)
{
    const uint4 u = reinterpret_cast<const uint4 *>(row)[c8];
    Bn8 r;
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xFFFF0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xFFFF0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xFFFF0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xFFFF0000u);
    return r;
}

__device__ __forceinline__ void bn_st8(__bf16 *row, int c8, const Bn8 &r)
{
    uint4 u;
    u.x = lcf_bf16_rne(r.v[0]) | (lcf_bf16_rne(r.v[1]) << 16);
    u.y = lcf_bf16_rne(r.v[2]) | (lcf_bf16_rne(r.v[3]) << 16);
    u.z = lcf_bf16_rne(r.v[4]) | (lcf_bf16_rne(r.v[5]) << 16);
    u.w = lcf_bf16_rne(r.v[6]) | (lcf_bf16_rne(r.v[7]) << 16);
    reinterpret_cast<uint4 *>(row)[c8] = u;
}

// thread -> (row lane, group of 8 channels); sums a[8] | b[8] per thread -> part[0:c] | part[c:2c]
__device__ __forceinline__ void bn8_block_reduce(int c, int c8n, int rl, int c8, int RL,
                                                 const Bn8 &a, const Bn8 &b, float *lds, float *part)
{
    if (rl < RL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            lds[rl * 2 * c + c8 * 8 + j] = a.v[j];
            lds[rl * 2 * c + c + c8 * 8 + j] = b.v[j];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * c; e += BN_BLOCK) {
        float s = 0.f;
        for (int r = 0; r < RL; ++r) s += lds[r * 2 * c + e];
        part[e] = s;
    }
}

__global__ void __launch_bounds__(BN_BLOCK)
bn8_stats_kernel(const __bf16 *__restrict__ x, int k, int c, int rows_per_block,
                 float *__restrict__ parts)
{
    __shared__ __align__(16) float lds[BN_BLOCK * 16];
    const int c8n = c >> 3, RL = BN_BLOCK / c8n, rl = threadIdx.x / c8n, c8 = threadIdx.x % c8n;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(k, r0 + rows_per_block);
    Bn8 s, q;
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] = q.v[j] = 0.f;
    if (rl < RL) {
        for (int r = r0 + rl; r < r1; r += RL) {
            const Bn8 v = bn_ld8(x + (size_t)r * c, c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s.v[j] += v.v[j];
                q.v[j] += v.v[j] * v.v[j];
            }
        }
    }
    bn8_block_reduce(c, c8n, rl, c8, RL, s, q, lds, parts + (size_t)blockIdx.x * 2 * c);
}

__global__ void __launch_bounds__(BN_BLOCK)
bn8_apply_relu_kernel(const __bf16 *__restrict__ x, int k, int c, int rows_per_block,
                      const double *__restrict__ sums, const float *__restrict__ gamma,
                      const float *__restrict__ beta, float eps, int relu,
                      __bf16 *__restrict__ y, float *__restrict__ stats, BnRunning run)
{
    __shared__ __align__(16) float sc[BN_MAX_C], sh[BN_MAX_C];
    for (int e = threadIdx.x; e < c; e += BN_BLOCK) {
        const double mean = sums[e] / k;
        double var = sums[c + e] / k - mean * mean;
        if (var < 0.0) var = 0.0;
        const float inv = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma ? gamma[e] : 1.f;
        sc[e] = g * inv;
        sh[e] = (beta ? beta[e] : 0.f) - (float)mean * g * inv;
        if (blockIdx.x == 0) {
            stats[e] = (float)mean;
            stats[c + e] = (float)var;
            stats[2 * c + e] = inv;
            if (run.mean) {
                const float mo = run.momentum;
                run.mean[e] = (1.f - mo) * run.mean[e] + mo * (float)mean;
                run.var[e] = (1.f - mo) * run.var[e] +
                             mo * (float)(var * ((double)k / (double)(k > 1 ? k - 1 : 1)));
                if (e == 0 && run.batches) run.batches[0] += 1;
            }
        }
    }
    __syncthreads();
    const int c8n = c >> 3, RL = BN_BLOCK / c8n, rl = threadIdx.x / c8n, c8 = threadIdx.x % c8n;
    if (rl >= RL) return;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = sc[c8 * 8 + j];
        b[j] = sh[c8 * 8 + j];
    }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(k, r0 + rows_per_block);
    for (int r = r0 + rl; r < r1; r += RL) {
        Bn8 v = bn_ld8(x + (size_t)r * c, c8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v.v[j] = __fmaf_rn(v.v[j], a[j], b[j]);
            if (relu) v.v[j] = fmaxf(v.v[j], 0.f);
        }
        bn_st8(y + (size_t)r * c, c8, v);
    }
}

// g = dy * (y > 0) with y's sign recomputed as in the forward (same fused multiply-add)
__global__ void __launch_bounds__(BN_BLOCK)
bn8_bwd_reduce_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy, int k, int c,
                      int rows_per_block, const float *__restrict__ stats,
                      const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                      float *__restrict__ parts)
{
    __shared__ __align__(16) float lds[BN_BLOCK * 16];
    const int c8n = c >> 3, RL = BN_BLOCK / c8n, rl = threadIdx.x / c8n, c8 = threadIdx.x % c8n;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(k, r0 + rows_per_block);
    Bn8 s, q;
#pragma unroll
    for (int j = 0; j < 8; ++j) s.v[j] = q.v[j] = 0.f;
    if (rl < RL) {
        float mean[8], inv[8], a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = c8 * 8 + j;
            mean[j] = stats[e];
            inv[j] = stats[2 * c + e];
            const float g = gamma ? gamma[e] : 1.f;
            a[j] = g * inv[j];
            b[j] = (beta ? beta[e] : 0.f) - mean[j] * g * inv[j];
        }
        for (int r = r0 + rl; r < r1; r += RL) {
            const Bn8 xv = bn_ld8(x + (size_t)r * c, c8);
            const Bn8 gv = bn_ld8(dy + (size_t)r * c, c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float g = gv.v[j];
                if (relu && !(__fmaf_rn(xv.v[j], a[j], b[j]) > 0.f)) g = 0.f;
                s.v[j] += g;
                q.v[j] += g * (xv.v[j] - mean[j]) * inv[j];
            }
        }
    }
    bn8_block_reduce(c, c8n, rl, c8, RL, s, q, lds, parts + (size_t)blockIdx.x * 2 * c);
}

__global__ void __launch_bounds__(BN_BLOCK)
bn8_bwd_apply_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy, int k, int c,
                     int rows_per_block, const float *__restrict__ stats,
                     const double *__restrict__ sums, const float *__restrict__ gamma,
                     const float *__restrict__ beta, int relu, __bf16 *__restrict__ dx,
                     float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    __shared__ __align__(16) float mg[BN_MAX_C], mgx[BN_MAX_C], gi[BN_MAX_C];
    for (int e = threadIdx.x; e < c; e += BN_BLOCK) {
        const double sg = sums[e], sgx = sums[c + e];
        mg[e] = (float)(sg / k);
        mgx[e] = (float)(sgx / k);
        gi[e] = (gamma ? gamma[e] : 1.f) * stats[2 * c + e];
        if (blockIdx.x == 0) {
            if (dgamma) dgamma[e] = (float)sgx;
            if (dbeta) dbeta[e] = (float)sg;
        }
    }
    __syncthreads();
    const int c8n = c >> 3, RL = BN_BLOCK / c8n, rl = threadIdx.x / c8n, c8 = threadIdx.x % c8n;
    if (rl >= RL) return;
    float mean[8], inv[8], a[8], b[8], m1[8], m2[8], w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = c8 * 8 + j;
        mean[j] = stats[e];
        inv[j] = stats[2 * c + e];
        const float g = gamma ? gamma[e] : 1.f;
        a[j] = g * inv[j];
        b[j] = (beta ? beta[e] : 0.f) - mean[j] * g * inv[j];
        m1[j] = mg[e];
        m2[j] = mgx[e];
        w[j] = gi[e];
    }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(k, r0 + rows_per_block);
    for (int r = r0 + rl; r < r1; r += RL) {
        const Bn8 xv = bn_ld8(x + (size_t)r * c, c8);
        const Bn8 gv = bn_ld8(dy + (size_t)r * c, c8);
        Bn8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float g = gv.v[j];
            if (relu && !(__fmaf_rn(xv.v[j], a[j], b[j]) > 0.f)) g = 0.f;
            o.v[j] = w[j] * (g - m1[j] - (xv.v[j] - mean[j]) * inv[j] * m2[j]);
        }
        bn_st8(dx + (size_t)r * c, c8, o);
    }
}

// rows per block / number of partials for k rows: enough blocks to fill the chip, at most
// BN_MAX_PARTS partials
static void bn_partition(int k, int *rows_per_block, int *nparts)
{
    int nb = (k + 255) / 256;
    if (nb > BN_MAX_PARTS) nb = BN_MAX_PARTS;
    if (nb < 1) nb = 1;
    int rpb = (k + nb - 1) / nb;
    *rows_per_block = rpb;
    *nparts = (k + rpb - 1) / rpb;
}

static int bn_check(int k, int c)
{
    if (k <= 0 || c <= 0) return LCF_ERR_ARG;
    if ((c & 3) || c > BN_MAX_C) return LCF_ERR_UNSUPPORTED;
    return LCF_OK;
}

extern "C" size_t lcf_bn_workspace_bytes(int k, int c)
{
    if (k <= 0 || c <= 0) return 0;
    return (size_t)BN_MAX_PARTS * 2 * c * sizeof(float) + (size_t)2 * c * sizeof(double);
}

// dtype codes of the _ex entry points: 0 = fp32 rows, 1 = bf16 rows
static int bn_fwd_run(const void *x, int x_bf16, int k, int c, const float *gamma,
                      const float *beta, float eps, int relu, void *y, int y_bf16, float *stats,
                      BnRunning run, void *ws, size_t ws_bytes, void *stream)
{
    int rc = bn_check(k, c);
    if (rc) return rc;
    if (!x || !y || !stats || !ws) return LCF_ERR_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)stats & 15)) return LCF_ERR_ARG;
    if (x_bf16 < 0 || x_bf16 > 1 || y_bf16 < 0 || y_bf16 > 1) return LCF_ERR_ARG;
    if ((x_bf16 || y_bf16) && (c & 7)) return LCF_ERR_UNSUPPORTED; // 16-byte aligned bf16 rows
    if (ws_bytes < lcf_bn_workspace_bytes(k, c)) return LCF_ERR_WORKSPACE;
    int rpb, np;
    bn_partition(k, &rpb, &np);
    hipStream_t st = (hipStream_t)stream;
    float *parts = (float *)ws;
    double *sums = (double *)((char *)ws + (size_t)BN_MAX_PARTS * 2 * c * sizeof(float));
    if (x_bf16 && y_bf16) { // all-bf16 rows: the 16-byte-per-thread kernels
        bn8_stats_kernel<<<np, BN_BLOCK, 0, st>>>((const __bf16 *)x, k, c, rpb, parts);
        LCF_CHECK_LAUNCH();
        bn_finalize_kernel<<<2 * c, 64, 0, st>>>(parts, np, 2 * c, sums);
        LCF_CHECK_LAUNCH();
        bn8_apply_relu_kernel<<<np, BN_BLOCK, 0, st>>>((const __bf16 *)x, k, c, rpb, sums, gamma,
                                                      beta, eps, relu, (__bf16 *)y, stats, run);
        LCF_CHECK_LAUNCH();
        return LCF_OK;
    }
    if (x_bf16)
        bn_stats_kernel<__bf16><<<np, BN_BLOCK, 0, st>>>((const __bf16 *)x, k, c, rpb, parts);
    else
        bn_stats_kernel<float><<<np, BN_BLOCK, 0, st>>>((const float *)x, k, c, rpb, parts);
    LCF_CHECK_LAUNCH();
    bn_finalize_kernel<<<2 * c, 64, 0, st>>>(parts, np, 2 * c, sums);
    LCF_CHECK_LAUNCH();
#define BN_APPLY(TX, TY)                                                                       \
    bn_apply_relu_kernel<TX, TY><<<np, BN_BLOCK, 0, st>>>((const TX *)x, k, c, rpb, sums, gamma, \
                                                          beta, eps, relu, (TY *)y, stats, run)
    if (x_bf16 && y_bf16) BN_APPLY(__bf16, __bf16);
    else if (x_bf16) BN_APPLY(__bf16, float);
    else if (y_bf16) BN_APPLY(float, __bf16);
    else BN_APPLY(float, float);
#undef BN_APPLY
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_bn_relu_fwd(const float *x, int k, int c, const float *gamma,
                               const float *beta, float eps, int relu, float *y, float *stats,
                               void *ws, size_t ws_bytes, void *stream)
{
    BnRunning run = {nullptr, nullptr, nullptr, 0.f};
    return bn_fwd_run(x, 0, k, c, gamma, beta, eps, relu, y, 0, stats, run, ws, ws_bytes, stream);
}

extern "C" int lcf_bn_relu_fwd_ex(const void *x, int x_bf16, int k, int c, const float *gamma,
                                  const float *beta, float eps, int relu, void *y, int y_bf16,
                                  float *stats, float *running_mean, float *running_var,
                                  long long *num_batches_tracked, float momentum, void *ws,
                                  size_t ws_bytes, void *stream)
{
    if ((running_mean == nullptr) != (running_var == nullptr)) return LCF_ERR_ARG;
    BnRunning run = {running_mean, running_var, num_batches_tracked, momentum};
    return bn_fwd_run(x, x_bf16, k, c, gamma, beta, eps, relu, y, y_bf16, stats, run, ws,
                      ws_bytes, stream);
}

template <typename TX, typename TY, typename TG, typename TD>
static int bn_bwd_launch(const void *x, const void *y, const void *dy, int k, int c,
                         const float *gamma, const float *stats, int relu, void *dx,
                         float *dgamma, float *dbeta, void *ws, hipStream_t st)
{
    int rpb, np;
    bn_partition(k, &rpb, &np);
    float *parts = (float *)ws;
    double *sums = (double *)((char *)ws + (size_t)BN_MAX_PARTS * 2 * c * sizeof(float));
    bn_bwd_reduce_kernel<TX, TY, TG><<<np, BN_BLOCK, 0, st>>>(
        (const TX *)x, (const TY *)y, (const TG *)dy, k, c, rpb, stats, relu, parts);
    LCF_CHECK_LAUNCH();
    bn_finalize_kernel<<<2 * c, 64, 0, st>>>(parts, np, 2 * c, sums);
    LCF_CHECK_LAUNCH();
    bn_bwd_apply_kernel<TX, TY, TG, TD><<<np, BN_BLOCK, 0, st>>>(
        (const TX *)x, (const TY *)y, (const TG *)dy, k, c, rpb, stats, sums, gamma, relu,
        (TD *)dx, dgamma, dbeta);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_bn_relu_bwd_ex(const void *x, int x_bf16, const void *y, int y_bf16,
                                  const void *dy, int dy_bf16, int k, int c, const float *gamma,
                                  const float *stats, int relu, void *dx, int dx_bf16,
                                  float *dgamma, float *dbeta, void *ws, size_t ws_bytes,
                                  void *stream)
{
    int rc = bn_check(k, c);
    if (rc) return rc;
    if (!x || !dy || !dx || !stats || !ws || (relu && !y)) return LCF_ERR_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15) ||
        ((uintptr_t)y & 15) || ((uintptr_t)stats & 15))
        return LCF_ERR_ARG;
    const int code = (x_bf16 ? 8 : 0) | (y_bf16 ? 4 : 0) | (dy_bf16 ? 2 : 0) | (dx_bf16 ? 1 : 0);
    if (code && (c & 7)) return LCF_ERR_UNSUPPORTED;
    if (ws_bytes < lcf_bn_workspace_bytes(k, c)) return LCF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    // the forms the training paths use: all fp32 (the module path); x fp32 with y / dy / dx in
    // any mix of fp32 and bf16 (the fused conv + BN + ReLU step)
    switch (code) {
    case 0: return bn_bwd_launch<float, float, float, float>(x, y, dy, k, c, gamma, stats, relu, dx, dgamma, dbeta, ws, st);
    case 1: return bn_bwd_launch<float, float, float, __bf16>(x, y, dy, k, c, gamma, stats, relu, dx, dgamma, dbeta, ws, st);
    case 2: return bn_bwd_launch<float, float, __bf16, float>(x, y, dy, k, c, gamma, stats, relu, dx, dgamma, dbeta, ws, st);
    case 3: return bn_bwd_launch<float, float, __bf16, __bf16>(x, y, dy, k, c, gamma, stats, relu, dx, dgamma, dbeta, ws, st);
    case 4: return bn_bwd_launch<float, __bf16, float, float>(x, y, dy, k, c, gamma, stats, relu, dx, dgamma, dbeta, ws, st);
    case 5: return bn_bwd_launch<float, __bf16, float, __bf16>(x, y, dy, k, c, gamma, stats, relu, dx, dgamma, dbeta, ws, st);
    case 6: return bn_bwd_launch<float, __bf16, __bf16, float>(x, y, dy, k, c, gamma, stats, relu, dx, dgamma, dbeta, ws, st);
    case 7: return bn_bwd_launch<float, __bf16, __bf16, __bf16>(x, y, dy, k, c, gamma, stats, relu, dx, dgamma, dbeta, ws, st);
    // everything bf16: the dense RPN's BatchNorm2d + ReLU on channels_last activations under
    // bf16 autocast (rows = pixels; objectdetection_amd/rpn.py)
    case 15: return bn_bwd_launch<__bf16, __bf16, __bf16, __bf16>(x, y, dy, k, c, gamma, stats, relu, dx, dgamma, dbeta, ws, st);
    default: return LCF_ERR_UNSUPPORTED; // other mixes with a bf16 x are not built
    }
}

extern "C" int lcf_bn_relu_bwd_b16(const void *x_bf16, const void *dy_bf16, int k, int c,
                                   const float *gamma, const float *beta, const float *stats,
                                   int relu, void *dx_bf16, float *dgamma, float *dbeta, void *ws,
                                   size_t ws_bytes, void *stream)
{
    int rc = bn_check(k, c);
    if (rc) return rc;
    if (c & 7) return LCF_ERR_UNSUPPORTED;
    if (!x_bf16 || !dy_bf16 || !dx_bf16 || !stats || !ws) return LCF_ERR_ARG;
    if (((uintptr_t)x_bf16 & 15) || ((uintptr_t)dy_bf16 & 15) || ((uintptr_t)dx_bf16 & 15) ||
        ((uintptr_t)stats & 15))
        return LCF_ERR_ARG;
    if (ws_bytes < lcf_bn_workspace_bytes(k, c)) return LCF_ERR_WORKSPACE;
    int rpb, np;
    bn_partition(k, &rpb, &np);
    hipStream_t st = (hipStream_t)stream;
    float *parts = (float *)ws;
    double *sums = (double *)((char *)ws + (size_t)BN_MAX_PARTS * 2 * c * sizeof(float));
    bn8_bwd_reduce_kernel<<<np, BN_BLOCK, 0, st>>>((const __bf16 *)x_bf16,
                                                   (const __bf16 *)dy_bf16, k, c, rpb, stats,
                                                   gamma, beta, relu, parts);
    LCF_CHECK_LAUNCH();
    bn_finalize_kernel<<<2 * c, 64, 0, st>>>(parts, np, 2 * c, sums);
    LCF_CHECK_LAUNCH();
    bn8_bwd_apply_kernel<<<np, BN_BLOCK, 0, st>>>((const __bf16 *)x_bf16, (const __bf16 *)dy_bf16,
                                                  k, c, rpb, stats, sums, gamma, beta, relu,
                                                  (__bf16 *)dx_bf16, dgamma, dbeta);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_bn_relu_bwd(const float *x, const float *y, const float *dy, int k, int c,
                               const float *gamma, const float *stats, int relu, float *dx,
                               float *dgamma, float *dbeta, void *ws, size_t ws_bytes,
                               void *stream)
{
    return lcf_bn_relu_bwd_ex(x, 0, y, 0, dy, 0, k, c, gamma, stats, relu, dx, 0, dgamma, dbeta,
                              ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------
// VFE training (voxel_encoder.py:69-80, 134-139 over the real points only, see
// objectdetection_amd/voxel_encoder.py::_forward_compact): per layer, after the Linear and the
// batch statistics, y = relu(x * scale + shift) and the per-voxel maximum agg[seg[r]] =
// max(agg[seg[r]], y[r]) -- one pass here instead of mul / add / relu / scatter-max, and one pass
// plus a column sum backward instead of nine elementwise / reduction launches.  agg arrives
// initialised (what the voxel's padding rows contribute, or a negative sentinel); y >= 0, so the
// maximum is an integer max on the float bits (order-independent: deterministic).
__global__ void __launch_bounds__(BN_BLOCK)
vfe_ars_fwd_kernel(const float *__restrict__ x, int p, int c, int rows_per_block,
                   const float *__restrict__ scale, const float *__restrict__ shift,
                   const int *__restrict__ seg, float *__restrict__ y, float *__restrict__ agg)
{
    const BnMap m = bn_map(c);
    if (m.rl >= m.RL) return;
    const float4 a = reinterpret_cast<const float4 *>(scale)[m.c4];
    const float4 b = reinterpret_cast<const float4 *>(shift)[m.c4];
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(p, r0 + rows_per_block);
    for (int r = r0 + m.rl; r < r1; r += m.RL) {
#pragma clang fp contract(off)
        float4 v = reinterpret_cast<const float4 *>(x + (size_t)r * c)[m.c4];
        // product and sum rounded separately (no fma contraction): the values of the torch
        // composition mul -> add -> relu, bit for bit
        const float4 t = make_float4(v.x * a.x, v.y * a.y, v.z * a.z, v.w * a.w);
        v.x = fmaxf(t.x + b.x, 0.f);
        v.y = fmaxf(t.y + b.y, 0.f);
        v.z = fmaxf(t.z + b.z, 0.f);
        v.w = fmaxf(t.w + b.w, 0.f);
        reinterpret_cast<float4 *>(y + (size_t)r * c)[m.c4] = v;
        int *dst = reinterpret_cast<int *>(agg + (size_t)seg[r] * c) + m.c4 * 4;
        atomicMax(dst + 0, __float_as_int(v.x));
        atomicMax(dst + 1, __float_as_int(v.y));
        atomicMax(dst + 2, __float_as_int(v.z));
        atomicMax(dst + 3, __float_as_int(v.w));
    }
}

// g = [y > 0] * (gy + [y == agg[seg]] * gagg[seg]);  dx = g * scale;  partial sums of g, g * x
__global__ void __launch_bounds__(BN_BLOCK)
vfe_ars_bwd_kernel(const float *__restrict__ x, const float *__restrict__ y,
                   const float *__restrict__ agg, const float *__restrict__ gy,
                   const float *__restrict__ gagg, const int *__restrict__ seg, int p, int c,
                   int rows_per_block, const float *__restrict__ scale, float *__restrict__ dx,
                   float *__restrict__ parts)
{
    __shared__ __align__(16) float lds[BN_BLOCK * 8];
    const BnMap m = bn_map(c);
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(p, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (m.rl < m.RL) {
        const float4 a = reinterpret_cast<const float4 *>(scale)[m.c4];
        for (int r = r0 + m.rl; r < r1; r += m.RL) {
            const float4 xv = reinterpret_cast<const float4 *>(x + (size_t)r * c)[m.c4];
            const float4 yv = reinterpret_cast<const float4 *>(y + (size_t)r * c)[m.c4];
            const size_t so = (size_t)seg[r] * c;
            const float4 av = reinterpret_cast<const float4 *>(agg + so)[m.c4];
            const float4 ga = reinterpret_cast<const float4 *>(gagg + so)[m.c4];
            float4 g = gy ? reinterpret_cast<const float4 *>(gy + (size_t)r * c)[m.c4]
                          : make_float4(0.f, 0.f, 0.f, 0.f);
            g.x = yv.x > 0.f ? g.x + (yv.x == av.x ? ga.x : 0.f) : 0.f;
            g.y = yv.y > 0.f ? g.y + (yv.y == av.y ? ga.y : 0.f) : 0.f;
            g.z = yv.z > 0.f ? g.z + (yv.z == av.z ? ga.z : 0.f) : 0.f;
            g.w = yv.w > 0.f ? g.w + (yv.w == av.w ? ga.w : 0.f) : 0.f;
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            q.x += g.x * xv.x; q.y += g.y * xv.y; q.z += g.z * xv.z; q.w += g.w * xv.w;
            reinterpret_cast<float4 *>(dx + (size_t)r * c)[m.c4] =
                make_float4(g.x * a.x, g.y * a.y, g.z * a.z, g.w * a.w);
        }
    }
    bn_block_reduce(m, c, s, q, lds, parts + (size_t)blockIdx.x * 2 * c);
}

// dshift[e] = sum of the partials of column e, dscale[e] = of column c + e; one wave per column
__global__ void __launch_bounds__(64)
vfe_ars_finalize_kernel(const float *__restrict__ parts, int nparts, int c,
                        float *__restrict__ dscale, float *__restrict__ dshift)
{
    const int e = blockIdx.x;
    double s = 0.0;
    for (int q = threadIdx.x; q < nparts; q += 64) s += (double)parts[(size_t)q * 2 * c + e];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
    if (threadIdx.x == 0) {
        if (e < c)
            dshift[e] = (float)s;
        else
            dscale[e - c] = (float)s;
    }
}

extern "C" int lcf_vfe_affine_relu_segmax_fwd(const float *x, int p, int c, const float *scale,
                                              const float *shift, const int *seg, float *y,
                                              float *agg, void *stream)
{
    int rc = bn_check(p, c);
    if (rc) return rc;
    if (!x || !scale || !shift || !seg || !y || !agg) return LCF_ERR_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)scale & 15) ||
        ((uintptr_t)shift & 15) || ((uintptr_t)agg & 15))
        return LCF_ERR_ARG;
    int rpb, np;
    bn_partition(p, &rpb, &np);
    vfe_ars_fwd_kernel<<<np, BN_BLOCK, 0, (hipStream_t)stream>>>(x, p, c, rpb, scale, shift, seg,
                                                                 y, agg);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_vfe_affine_relu_segmax_bwd(const float *x, const float *y, const float *agg,
                                              const float *gy, const float *gagg, const int *seg,
                                              int p, int c, const float *scale, float *dx,
                                              float *dscale, float *dshift, void *ws,
                                              size_t ws_bytes, void *stream)
{
    int rc = bn_check(p, c);
    if (rc) return rc;
    if (!x || !y || !agg || !gagg || !seg || !scale || !dx || !dscale || !dshift || !ws)
        return LCF_ERR_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)agg & 15) ||
        ((uintptr_t)gy & 15) || ((uintptr_t)gagg & 15) || ((uintptr_t)scale & 15) ||
        ((uintptr_t)dx & 15))
        return LCF_ERR_ARG;
    if (ws_bytes < lcf_bn_workspace_bytes(p, c)) return LCF_ERR_WORKSPACE;
    int rpb, np;
    bn_partition(p, &rpb, &np);
    hipStream_t st = (hipStream_t)stream;
    float *parts = (float *)ws;
    vfe_ars_bwd_kernel<<<np, BN_BLOCK, 0, st>>>(x, y, agg, gy, gagg, seg, p, c, rpb, scale, dx,
                                                parts);
    LCF_CHECK_LAUNCH();
    vfe_ars_finalize_kernel<<<2 * c, 64, 0, st>>>(parts, np, c, dscale, dshift);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

// ------------------------------------------------------------------------------------------
// A whole VFE layer tail in TRAINING mode over the real rows only (voxel_encoder.py:65-80 and
// 126-139 as _forward_compact restates them): batch statistics over the n = K * T rows of the
// reference's dense tensor, of which only the p real rows (and, first layer, one padding
// representative per voxel with its multiplicity) are materialised; BatchNorm affine + ReLU;
// per-voxel maximum including what the voxel's padding rows contribute.  One call forward, one
// backward (was ~35 torch launches per layer and direction).
//   mode 0 (first layer): the padding rows of voxel v are wp[v] copies of xp[v]: they enter the
//           statistics with that weight and the maximum once, as relu(xp * scale + shift);
//   mode 1 (later layer): padding rows are zeros (masked input, Linear without bias): n - p zeros
//           in the statistics, relu(shift) in the maximum of every voxel with wp[v] > 0;
//   mode 2 (final Linear): statistics as mode 1; the maximum starts at 0 (masked output >= 0).
// Rows of one voxel are contiguous (seg non-decreasing): the first row of a run takes the run's
// maximum in registers and writes agg[v] = max(init, run) with plain stores -- no atomics.
#define VFE_NEG (-3.402823466e+38f)

__global__ void __launch_bounds__(BN_BLOCK)
vfe_wstats_kernel(const float *__restrict__ x, const float *__restrict__ w, int k, int c,
                  int rows_per_block, float *__restrict__ parts)
{
    __shared__ __align__(16) float lds[BN_BLOCK * 8];
    const BnMap m = bn_map(c);
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(k, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (m.rl < m.RL) {
        for (int r = r0 + m.rl; r < r1; r += m.RL) {
            const float wv = w ? w[r] : 1.f;
            if (wv == 0.f) continue;
            const float4 v = reinterpret_cast<const float4 *>(x + (size_t)r * c)[m.c4];
            s.x += wv * v.x; s.y += wv * v.y; s.z += wv * v.z; s.w += wv * v.w;
            q.x += wv * v.x * v.x; q.y += wv * v.y * v.y;
            q.z += wv * v.z * v.z; q.w += wv * v.w * v.w;
        }
    }
    bn_block_reduce(m, c, s, q, lds, parts + (size_t)blockIdx.x * 2 * c);
}

// stats [5c] = scale | shift | mean | biased var | inv  (n rows in all)
__global__ void __launch_bounds__(BN_MAX_C)
vfe_coeffs_fwd_kernel(const double *__restrict__ sums, double n, int c,
                      const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                      float *__restrict__ stats)
{
    const int e = threadIdx.x;
    if (e >= c) return;
    const double mean = sums[e] / n;
    double var = sums[c + e] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float inv = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = (gamma ? gamma[e] : 1.f) * inv;
    stats[e] = sc;
    stats[c + e] = (beta ? beta[e] : 0.f) - (float)mean * sc;
    stats[2 * c + e] = (float)mean;
    stats[3 * c + e] = (float)var;
    stats[4 * c + e] = inv;
}

__device__ __forceinline__ float4 vfe_affine_relu(float4 v, float4 a, float4 b)
{
#pragma clang fp contract(off)
    const float4 t = make_float4(v.x * a.x, v.y * a.y, v.z * a.z, v.w * a.w);
    return make_float4(fmaxf(t.x + b.x, 0.f), fmaxf(t.y + b.y, 0.f), fmaxf(t.z + b.z, 0.f),
                       fmaxf(t.w + b.w, 0.f));
}

__device__ __forceinline__ float4 vfe_max4(float4 a, float4 b)
{
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

// what the padding rows of voxel v put into its maximum
__device__ __forceinline__ float4 vfe_pad_value(int mode, const float *xp, const float *wp, int v,
                                                int c, int c4, float4 a, float4 b)
{
    if (mode == 2) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(wp[v] > 0.f)) return make_float4(VFE_NEG, VFE_NEG, VFE_NEG, VFE_NEG);
    if (mode == 1) return make_float4(fmaxf(b.x, 0.f), fmaxf(b.y, 0.f), fmaxf(b.z, 0.f),
                                      fmaxf(b.w, 0.f));
    return vfe_affine_relu(reinterpret_cast<const float4 *>(xp + (size_t)v * c)[c4], a, b);
}

// agg[v] = the padding rows' contribution, for every voxel (also those without a real row)
__global__ void __launch_bounds__(BN_BLOCK)
vfe_init_kernel(const float *__restrict__ xp, const float *__restrict__ wp, int k, int c, int mode,
                const float *__restrict__ stats, float *__restrict__ agg)
{
    const int c4n = c >> 2;
    const long long t = (long long)blockIdx.x * BN_BLOCK + threadIdx.x;
    if (t >= (long long)k * c4n) return;
    const int v = (int)(t / c4n), c4 = (int)(t % c4n);
    const float4 a = reinterpret_cast<const float4 *>(stats)[c4];
    const float4 b = reinterpret_cast<const float4 *>(stats + c)[c4];
    reinterpret_cast<float4 *>(agg + (size_t)v * c)[c4] =
        vfe_pad_value(mode, xp, wp, v, c, c4, a, b);
}

__global__ void __launch_bounds__(BN_BLOCK)
vfe_layer_fwd_kernel(const float *__restrict__ x, int p, int c, int rows_per_block,
                     const float *__restrict__ stats, const int *__restrict__ seg,
                     float *__restrict__ y, int ldy, float *__restrict__ agg)
{
    const BnMap m = bn_map(c);
    if (m.rl >= m.RL) return;
    const float4 a = reinterpret_cast<const float4 *>(stats)[m.c4];
    const float4 b = reinterpret_cast<const float4 *>(stats + c)[m.c4];
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(p, r0 + rows_per_block);
    for (int r = r0 + m.rl; r < r1; r += m.RL) {
        const float4 v =
            vfe_affine_relu(reinterpret_cast<const float4 *>(x + (size_t)r * c)[m.c4], a, b);
        if (y) reinterpret_cast<float4 *>(y + (size_t)r * ldy)[m.c4] = v;
        const int sv = seg[r];
        if (r == 0 || seg[r - 1] != sv) { // first row of the voxel's run: the run's maximum
            float4 mx = v;
            for (int r2 = r + 1; r2 < p && seg[r2] == sv; ++r2)
                mx = vfe_max4(mx, vfe_affine_relu(
                    reinterpret_cast<const float4 *>(x + (size_t)r2 * c)[m.c4], a, b));
            float4 *d = reinterpret_cast<float4 *>(agg + (size_t)sv * c) + m.c4;
            *d = vfe_max4(*d, mx); // one writer per (voxel, column group)
        }
    }
}

// g = [y > 0] * (gy + [y == agg[seg]] * gagg[seg]) with y recomputed; dx = g * scale; partial
// sums of g and g * x
__global__ void __launch_bounds__(BN_BLOCK)
vfe_layer_bwd_kernel(const float *__restrict__ x, const float *__restrict__ agg,
                     const float *__restrict__ gy, const float *__restrict__ gagg,
                     const int *__restrict__ seg, int p, int c, int rows_per_block, int ldg,
                     const float *__restrict__ stats, float *__restrict__ dx,
                     float *__restrict__ parts)
{
    __shared__ __align__(16) float lds[BN_BLOCK * 8];
    const BnMap m = bn_map(c);
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(p, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (m.rl < m.RL) {
        const float4 a = reinterpret_cast<const float4 *>(stats)[m.c4];
        const float4 b = reinterpret_cast<const float4 *>(stats + c)[m.c4];
        for (int r = r0 + m.rl; r < r1; r += m.RL) {
            const float4 xv = reinterpret_cast<const float4 *>(x + (size_t)r * c)[m.c4];
            const float4 yv = vfe_affine_relu(xv, a, b);
            const size_t so = (size_t)seg[r] * c;
            const float4 av = reinterpret_cast<const float4 *>(agg + so)[m.c4];
            const float4 ga = reinterpret_cast<const float4 *>(gagg + so)[m.c4];
            float4 g = gy ? reinterpret_cast<const float4 *>(gy + (size_t)r * ldg)[m.c4]
                          : make_float4(0.f, 0.f, 0.f, 0.f);
            g.x = yv.x > 0.f ? g.x + (yv.x == av.x ? ga.x : 0.f) : 0.f;
            g.y = yv.y > 0.f ? g.y + (yv.y == av.y ? ga.y : 0.f) : 0.f;
            g.z = yv.z > 0.f ? g.z + (yv.z == av.z ? ga.z : 0.f) : 0.f;
            g.w = yv.w > 0.f ? g.w + (yv.w == av.w ? ga.w : 0.f) : 0.f;
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            q.x += g.x * xv.x; q.y += g.y * xv.y; q.z += g.z * xv.z; q.w += g.w * xv.w;
            reinterpret_cast<float4 *>(dx + (size_t)r * c)[m.c4] =
                make_float4(g.x * a.x, g.y * a.y, g.z * a.z, g.w * a.w);
        }
    }
    bn_block_reduce(m, c, s, q, lds, parts + (size_t)blockIdx.x * 2 * c);
}

// out[r, c : 2c] = agg[seg[r]]: the "repeated" half of a VFE layer's output
// (voxel_encoder.py:77-80)
__global__ void __launch_bounds__(BN_BLOCK)
vfe_repeat_kernel(const float *__restrict__ agg, const int *__restrict__ seg, int p, int c,
                  float *__restrict__ out)
{
    const int c4n = c >> 2;
    const long long t = (long long)blockIdx.x * BN_BLOCK + threadIdx.x;
    if (t >= (long long)p * c4n) return;
    const int r = (int)(t / c4n), c4 = (int)(t % c4n);
    reinterpret_cast<float4 *>(out + (size_t)r * 2 * c + c)[c4] =
        reinterpret_cast<const float4 *>(agg + (size_t)seg[r] * c)[c4];
}

// its backward: gagg[v] = sum over the voxel's run of rows of gout[r, c : 2c] (gagg zeroed
// before; the first row of a run sums it in row order: deterministic, no atomics)
__global__ void __launch_bounds__(BN_BLOCK)
vfe_runsum_kernel(const float *__restrict__ gout, const int *__restrict__ seg, int p, int c,
                  float *__restrict__ gagg)
{
    const int c4n = c >> 2;
    const long long t = (long long)blockIdx.x * BN_BLOCK + threadIdx.x;
    if (t >= (long long)p * c4n) return;
    const int r = (int)(t / c4n), c4 = (int)(t % c4n);
    const int sv = seg[r];
    if (r != 0 && seg[r - 1] == sv) return;
    float4 acc = reinterpret_cast<const float4 *>(gout + (size_t)r * 2 * c + c)[c4];
    for (int r2 = r + 1; r2 < p && seg[r2] == sv; ++r2) {
        const float4 v = reinterpret_cast<const float4 *>(gout + (size_t)r2 * 2 * c + c)[c4];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4 *>(gagg + (size_t)sv * c)[c4] = acc;
}

// the padding representatives' share of the same sums (modes 0 and 1); mode 0 also starts dxp
__global__ void __launch_bounds__(BN_BLOCK)
vfe_pad_bwd_kernel(const float *__restrict__ xp, const float *__restrict__ wp,
                   const float *__restrict__ agg, const float *__restrict__ gagg, int k, int c,
                   int rows_per_block, int mode, const float *__restrict__ stats,
                   float *__restrict__ dxp, float *__restrict__ parts)
{
    __shared__ __align__(16) float lds[BN_BLOCK * 8];
    const BnMap m = bn_map(c);
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(k, r0 + rows_per_block);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
    if (m.rl < m.RL) {
        const float4 a = reinterpret_cast<const float4 *>(stats)[m.c4];
        const float4 b = reinterpret_cast<const float4 *>(stats + c)[m.c4];
        for (int v = r0 + m.rl; v < r1; v += m.RL) {
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 xv = g;
            if (wp[v] > 0.f) {
                const float4 pv = vfe_pad_value(mode, xp, wp, v, c, m.c4, a, b);
                const float4 av = reinterpret_cast<const float4 *>(agg + (size_t)v * c)[m.c4];
                const float4 ga = reinterpret_cast<const float4 *>(gagg + (size_t)v * c)[m.c4];
                g.x = pv.x > 0.f && pv.x == av.x ? ga.x : 0.f;
                g.y = pv.y > 0.f && pv.y == av.y ? ga.y : 0.f;
                g.z = pv.z > 0.f && pv.z == av.z ? ga.z : 0.f;
                g.w = pv.w > 0.f && pv.w == av.w ? ga.w : 0.f;
                if (mode == 0) xv = reinterpret_cast<const float4 *>(xp + (size_t)v * c)[m.c4];
            }
            s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
            q.x += g.x * xv.x; q.y += g.y * xv.y; q.z += g.z * xv.z; q.w += g.w * xv.w;
            if (dxp)
                reinterpret_cast<float4 *>(dxp + (size_t)v * c)[m.c4] =
                    make_float4(g.x * a.x, g.y * a.y, g.z * a.z, g.w * a.w);
        }
    }
    bn_block_reduce(m, c, s, q, lds, parts + (size_t)blockIdx.x * 2 * c);
}

// sums [2c] = dshift | dscale  ->  dgamma, dbeta, and coef [2c] = A | B of the statistics' share
// of the row gradients, dx += A * x + B:
//   scale = gamma * inv, shift = beta - mean * scale, inv = (var + eps)^-1/2
//   dgamma = inv * (dscale - mean * dshift), dbeta = dshift,
//   dmean = -scale * dshift, dvar = -inv^3 / 2 * gamma * (dscale - mean * dshift),
//   A = 2 dvar / n, B = dmean / n - A * mean
__global__ void __launch_bounds__(BN_MAX_C)
vfe_coeffs_bwd_kernel(const double *__restrict__ sums, double n, int c,
                      const float *__restrict__ gamma, const float *__restrict__ stats,
                      float *__restrict__ dgamma, float *__restrict__ dbeta,
                      float *__restrict__ coef)
{
    const int e = threadIdx.x;
    if (e >= c) return;
    const double dshift = sums[e], dscale = sums[c + e];
    const double mean = stats[2 * c + e], inv = stats[4 * c + e];
    const double g = gamma ? gamma[e] : 1.0;
    const double t = dscale - mean * dshift;
    if (dgamma) dgamma[e] = (float)(inv * t);
    if (dbeta) dbeta[e] = (float)dshift;
    const double dmean = -(double)stats[e] * dshift;
    const double dvar = -0.5 * inv * inv * inv * g * t;
    const double A = 2.0 * dvar / n;
    coef[e] = (float)A;
    coef[c + e] = (float)(dmean / n - A * mean);
}

// dx[r] += w[r] * (A * x[r] + B)   (w NULL: 1)
__global__ void __launch_bounds__(BN_BLOCK)
vfe_axpb_kernel(const float *__restrict__ x, const float *__restrict__ w, int k, int c,
                const float *__restrict__ coef, float *__restrict__ dx)
{
    const int c4n = c >> 2;
    const long long t = (long long)blockIdx.x * BN_BLOCK + threadIdx.x;
    if (t >= (long long)k * c4n) return;
    const int r = (int)(t / c4n), c4 = (int)(t % c4n);
    const float wv = w ? w[r] : 1.f;
    if (wv == 0.f) return;
    const float4 A = reinterpret_cast<const float4 *>(coef)[c4];
    const float4 B = reinterpret_cast<const float4 *>(coef + c)[c4];
    const float4 xv = reinterpret_cast<const float4 *>(x + (size_t)r * c)[c4];
    float4 *d = reinterpret_cast<float4 *>(dx + (size_t)r * c) + c4;
    float4 o = *d;
    o.x += wv * (A.x * xv.x + B.x);
    o.y += wv * (A.y * xv.y + B.y);
    o.z += wv * (A.z * xv.z + B.z);
    o.w += wv * (A.w * xv.w + B.w);
    *d = o;
}

struct VfeLayerWs {
    float *parts;  // [(np_rows + np_vox)][2c]
    double *sums;  // [2c]
    float *coef;   // [2c]
    float *gagg;   // [k][c] (backward of modes 0 and 1)
    int rpb_r, np_r, rpb_v, np_v;
    size_t bytes;
};

static VfeLayerWs vfe_layer_carve(void *ws, int p, int k, int c)
{
    VfeLayerWs w;
    bn_partition(p, &w.rpb_r, &w.np_r);
    bn_partition(k, &w.rpb_v, &w.np_v);
    char *b = (char *)ws;
    size_t off = 0;
    w.parts = (float *)(b + off);
    off += lcf_align_up((size_t)(w.np_r + w.np_v) * 2 * c * sizeof(float), 256);
    w.sums = (double *)(b + off);
    off += lcf_align_up((size_t)2 * c * sizeof(double), 256);
    w.coef = (float *)(b + off);
    off += lcf_align_up((size_t)2 * c * sizeof(float), 256);
    w.gagg = (float *)(b + off);
    off += lcf_align_up((size_t)k * c * sizeof(float), 256);
    w.bytes = off;
    return w;
}

extern "C" size_t lcf_vfe_layer_workspace_bytes(int p, int k, int c)
{
    if (p <= 0 || k <= 0 || c <= 0) return 0;
    return vfe_layer_carve(nullptr, p, k, c).bytes;
}

static int vfe_layer_check(const float *x, int p, const float *xp, const float *wp, int k, int c,
                           long long n, const int *seg, int mode)
{
    int rc = bn_check(p, c);
    if (rc) return rc;
    if (k <= 0 || n < p || mode < 0 || mode > 2 || !x || !seg) return LCF_ERR_ARG;
    if (mode != 2 && !wp) return LCF_ERR_ARG;
    if (mode == 0 && !xp) return LCF_ERR_ARG;
    if (((uintptr_t)x & 15) || ((uintptr_t)xp & 15)) return LCF_ERR_ARG;
    return LCF_OK;
}

extern "C" int lcf_vfe_layer_fwd(const float *x, int p, const float *xp, const float *wp, int k,
                                 int c, long long n, const int *seg, const float *gamma,
                                 const float *beta, float eps, int mode, float *out, float *agg,
                                 float *stats, void *ws, size_t ws_bytes, void *stream)
{
    int rc = vfe_layer_check(x, p, xp, wp, k, c, n, seg, mode);
    if (rc) return rc;
    if (!agg || !stats || !ws || (mode != 2 && !out) || ((uintptr_t)out & 15) ||
        ((uintptr_t)agg & 15) || ((uintptr_t)stats & 15))
        return LCF_ERR_ARG;
    VfeLayerWs w = vfe_layer_carve(ws, p, k, c);
    if (ws_bytes < w.bytes) return LCF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int np = w.np_r;
    vfe_wstats_kernel<<<w.np_r, BN_BLOCK, 0, st>>>(x, nullptr, p, c, w.rpb_r, w.parts);
    LCF_CHECK_LAUNCH();
    if (mode == 0) {
        vfe_wstats_kernel<<<w.np_v, BN_BLOCK, 0, st>>>(xp, wp, k, c, w.rpb_v,
                                                       w.parts + (size_t)w.np_r * 2 * c);
        LCF_CHECK_LAUNCH();
        np += w.np_v;
    }
    bn_finalize_kernel<<<2 * c, 64, 0, st>>>(w.parts, np, 2 * c, w.sums);
    LCF_CHECK_LAUNCH();
    vfe_coeffs_fwd_kernel<<<1, BN_MAX_C, 0, st>>>(w.sums, (double)n, c, gamma, beta, eps, stats);
    LCF_CHECK_LAUNCH();
    const long long items = (long long)k * (c >> 2);
    vfe_init_kernel<<<(unsigned)((items + BN_BLOCK - 1) / BN_BLOCK), BN_BLOCK, 0, st>>>(
        xp, wp, k, c, mode, stats, agg);
    LCF_CHECK_LAUNCH();
    vfe_layer_fwd_kernel<<<w.np_r, BN_BLOCK, 0, st>>>(x, p, c, w.rpb_r, stats, seg,
                                                      mode == 2 ? nullptr : out, 2 * c, agg);
    LCF_CHECK_LAUNCH();
    if (mode != 2) {
        const long long ir = (long long)p * (c >> 2);
        vfe_repeat_kernel<<<(unsigned)((ir + BN_BLOCK - 1) / BN_BLOCK), BN_BLOCK, 0, st>>>(
            agg, seg, p, c, out);
        LCF_CHECK_LAUNCH();
    }
    return LCF_OK;
}

extern "C" int lcf_vfe_layer_bwd(const float *x, const float *agg, const float *gout, int p,
                                 const float *xp, const float *wp, int k, int c, long long n,
                                 const int *seg, const float *gamma, const float *stats, int mode,
                                 float *dx, float *dxp, float *dgamma, float *dbeta, void *ws,
                                 size_t ws_bytes, void *stream)
{
    int rc = vfe_layer_check(x, p, xp, wp, k, c, n, seg, mode);
    if (rc) return rc;
    if (!agg || !gout || !stats || !dx || !ws || (mode == 0 && !dxp)) return LCF_ERR_ARG;
    if (((uintptr_t)agg & 15) || ((uintptr_t)gout & 15) || ((uintptr_t)dx & 15) ||
        ((uintptr_t)dxp & 15) || ((uintptr_t)stats & 15))
        return LCF_ERR_ARG;
    VfeLayerWs w = vfe_layer_carve(ws, p, k, c);
    if (ws_bytes < w.bytes) return LCF_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int np = w.np_r;
    // modes 0 / 1: gout [p, 2c] = gradient of (y | agg[seg]); mode 2: gout [k, c] = gradient of agg
    const float *gy = nullptr, *gagg = gout;
    if (mode != 2) {
        if (lcf_fill32(w.gagg, 0u, (size_t)k * c * 4, st) != LCF_OK) return LCF_ERR_LAUNCH;
        const long long ir = (long long)p * (c >> 2);
        vfe_runsum_kernel<<<(unsigned)((ir + BN_BLOCK - 1) / BN_BLOCK), BN_BLOCK, 0, st>>>(
            gout, seg, p, c, w.gagg);
        LCF_CHECK_LAUNCH();
        gy = gout;
        gagg = w.gagg;
    }
    vfe_layer_bwd_kernel<<<w.np_r, BN_BLOCK, 0, st>>>(x, agg, gy, gagg, seg, p, c, w.rpb_r, 2 * c,
                                                      stats, dx, w.parts);
    LCF_CHECK_LAUNCH();
    if (mode != 2) {
        vfe_pad_bwd_kernel<<<w.np_v, BN_BLOCK, 0, st>>>(xp, wp, agg, gagg, k, c, w.rpb_v, mode,
                                                        stats, mode == 0 ? dxp : nullptr,
                                                        w.parts + (size_t)w.np_r * 2 * c);
        LCF_CHECK_LAUNCH();
        np += w.np_v;
    }
    bn_finalize_kernel<<<2 * c, 64, 0, st>>>(w.parts, np, 2 * c, w.sums);
    LCF_CHECK_LAUNCH();
    vfe_coeffs_bwd_kernel<<<1, BN_MAX_C, 0, st>>>(w.sums, (double)n, c, gamma, stats, dgamma,
                                                  dbeta, w.coef);
    LCF_CHECK_LAUNCH();
    const long long ir = (long long)p * (c >> 2);
    vfe_axpb_kernel<<<(unsigned)((ir + BN_BLOCK - 1) / BN_BLOCK), BN_BLOCK, 0, st>>>(
        x, nullptr, p, c, w.coef, dx);
    LCF_CHECK_LAUNCH();
    if (mode == 0) {
        const long long iv = (long long)k * (c >> 2);
        vfe_axpb_kernel<<<(unsigned)((iv + BN_BLOCK - 1) / BN_BLOCK), BN_BLOCK, 0, st>>>(
            xp, wp, k, c, w.coef, dxp);
        LCF_CHECK_LAUNCH();
    }
    return LCF_OK;
}
