// conv.hip -- indice_conv forward / dgrad / wgrad on the gfx950 fp32 MFMA pipe.
//
// Replaces torch.ops.spconv.indice_conv(+_backward) as used by spconv.SparseConv3d/SubMConv3d
// in second/pytorch/models/middle.py:127-192 (per-offset gather -> mm -> scatter-add).
//
// Forward is OUTPUT-STATIONARY over the neighbour table: a 16-row output tile per wave keeps its
// accumulators in registers while it walks the kernel offsets; per offset it gathers the 16
// input rows named by out_nbr (zeros where -1) straight into the MFMA A operand and multiplies
// by W[o], which the workgroup stages through LDS (double buffered, one barrier per offset) in
// an image laid out so every B fragment read is one conflict-free ds_read_b128.  The "scatter"
// is the tile's own store, so there are no atomics and the result is run-to-run deterministic.
// v_mfma_f32_16x16x4_f32 is an exact fp32 FMA chain (MI355X guide section 3), so results agree
// with the CPU oracle to fp32 summation-order differences (<= 1e-4 relative is the test bar).
#include <stdlib.h>

#include "common.h"
#include "tc_plan.h"

#include <mutex>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define CV_ROWS_PER_WAVE 16
#define CV_WAVES 4
#define CV_ROWS (CV_ROWS_PER_WAVE * CV_WAVES)
#define CV_BLOCK (CV_WAVES * 64)

static int cv_pad_cin(int c)
{
    const int s[] = {4, 8, 16, 32, 64, 128};
    for (int i = 0; i < 6; ++i)
        if (c <= s[i]) return s[i];
    return -1;
}
static int cv_pad_cout(int c)
{
    const int s[] = {16, 32, 64, 128};
    for (int i = 0; i < 4; ++i)
        if (c <= s[i]) return s[i];
    return -1;
}

// position (in floats) of W[k][c] inside one offset's image
__host__ __device__ __forceinline__ int cv_img_pos(int k, int c, int CIN, int COUT)
{
    (void)COUT;
    const int KL = CIN / 4;
    const int VW = KL < 4 ? KL : 4;
    const int SG = KL / VW;
    int q = k / KL, s = k % KL;
    int sg = s / VW, v = s % VW;
    int t = c / 16, jj = c % 16;
    return ((((t * SG + sg) * 4 + q) * 16 + jj) * VW) + v;
}

__global__ void __launch_bounds__(256)
cv_weight_image_kernel(const float *__restrict__ w, int kvol, int cin, int cout, int CIN,
                       int COUT, int transpose, float *__restrict__ img)
{
    // image element index e -> (o, k, c) of the (possibly transposed) padded matrix
    long long total = (long long)kvol * CIN * COUT;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        int o = (int)(e / (CIN * COUT));
        int r = (int)(e - (long long)o * CIN * COUT);
        int k = r / COUT, c = r % COUT;
        float v = 0.f;
        if (!transpose) {
            if (k < cin && c < cout) v = w[((size_t)o * cin + k) * cout + c];
        } else {
            // image of W[o]^T : rows = original cout (k), cols = original cin (c)
            if (k < cout && c < cin) v = w[((size_t)o * cin + c) * cout + k];
        }
        img[(size_t)o * CIN * COUT + cv_img_pos(k, c, CIN, COUT)] = v;
    }
}

// NSPLIT = how many waves share one 16-row tile by splitting its output columns.  More splits
// = smaller workgroups (16*4/NSPLIT rows) = finer load balance over the 256 CUs when K is only a
// few tens of thousands of rows, at the price of re-staging W[o] for fewer rows.
template <int CIN, int COUT, int NSPLIT>
__global__ void __launch_bounds__(CV_BLOCK)
cv_fwd_kernel(const float *__restrict__ feat, const float *__restrict__ wimg,
              const int *__restrict__ nbr, int k_cap, const int *__restrict__ k_dev, int kvol,
              int cin, int cout, const float *__restrict__ scale,
              const float *__restrict__ shift, int relu, float *__restrict__ out)
{
    const int k_out = lcf_count(k_cap, k_dev);
    constexpr int KL = CIN / 4;          // k values per lane quarter
    constexpr int VW = KL < 4 ? KL : 4;  // floats per LDS fragment read
    constexpr int SG = KL / VW;          // fragment reads per column tile
    constexpr int NT = COUT / 16 / NSPLIT; // column tiles per wave
    constexpr int RT = CV_WAVES / NSPLIT;  // row tiles per workgroup
    constexpr int ROWS = RT * 16;          // rows per workgroup
    constexpr int WELEMS = CIN * COUT;   // floats per offset image
    constexpr int WVEC = WELEMS / 4;     // float4 per offset image
    constexpr int WPT = (WVEC + CV_BLOCK - 1) / CV_BLOCK; // float4 per thread to stage
    constexpr bool WFULL = (WVEC % CV_BLOCK) == 0;
    constexpr int A4 = KL >= 4 ? KL / 4 : 1; // float4 registers of A per lane
    static_assert(NT >= 1 && RT >= 1, "bad split");

    extern __shared__ __align__(16) unsigned char smem[];
    float *wl = reinterpret_cast<float *>(smem);                 // [2][WELEMS]
    int *nl = reinterpret_cast<int *>(smem + 2 * WELEMS * 4);    // [ROWS][kvol]
    int *act = nl + ROWS * kvol;                                 // [kvol] active offsets
    int *nact_p = act + kvol;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rt = wave / NSPLIT;        // row tile of this wave
    const int cs = wave % NSPLIT;        // column slice of this wave
    const int r16 = lane & 15;
    const int q = lane >> 4;
    const int row0 = blockIdx.x * ROWS;
    if (row0 >= k_out) return; // capacity-sized grid: workgroups past the actual count exit

    // ---- neighbour tile -> LDS (coalesced), rows past k_out read as -1
    {
        const int total = ROWS * kvol;
        const long long gbase = (long long)row0 * kvol;
        const long long glimit = (long long)k_out * kvol;
        for (int e = tid; e < total; e += CV_BLOCK)
            nl[e] = (gbase + e < glimit) ? nbr[gbase + e] : -1;
    }
    __syncthreads();
    // ---- workgroup-uniform list of offsets that have at least one neighbour in the tile
    if (wave == 0) {
        bool any = false;
        if (lane < kvol) {
            for (int r = 0; r < ROWS; ++r) any |= nl[r * kvol + lane] >= 0;
        }
        unsigned long long m = __ballot(any);
        if (lane == 0) {
            int n = 0;
            while (m) {
                int o = __builtin_ctzll(m);
                m &= m - 1;
                act[n++] = o;
            }
            *nact_p = n;
        }
    }
    __syncthreads();
    const int nact = *nact_p;

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool exact = (cin == CIN);
    const int myrow = rt * 16 + r16;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // A operand of one offset: KL floats per lane (k = q*KL .. q*KL+KL-1 of the gathered row)
    struct AFrag { float4 v[A4]; };

    auto gather = [&](int o, bool &any) -> AFrag {
        AFrag a;
        int id = nl[myrow * kvol + o];
        bool has = id >= 0;
#pragma unroll
        for (int i = 0; i < A4; ++i) a.v[i] = zero4;
        if (has) {
            const float *src = feat + (size_t)id * cin + q * KL;
            if (exact) {
                if constexpr (KL >= 4) {
#pragma unroll
                    for (int i = 0; i < A4; ++i) a.v[i] = reinterpret_cast<const float4 *>(src)[i];
                } else if constexpr (KL == 2) {
                    float2 x = *reinterpret_cast<const float2 *>(src);
                    a.v[0].x = x.x;
                    a.v[0].y = x.y;
                } else {
                    a.v[0].x = src[0];
                }
            } else {
                float tmp[A4 * 4];
#pragma unroll
                for (int s = 0; s < A4 * 4; ++s)
                    tmp[s] = (s < KL && q * KL + s < cin) ? src[s] : 0.f;
#pragma unroll
                for (int i = 0; i < A4; ++i)
                    a.v[i] = make_float4(tmp[4 * i], tmp[4 * i + 1], tmp[4 * i + 2], tmp[4 * i + 3]);
            }
        }
        any = __ballot(has) != 0ull; // wave-uniform: does this 16-row tile touch offset o?
        return a;
    };

    struct WFrag { float4 v[WPT]; };
    auto wload = [&](int o) -> WFrag {
        WFrag w;
        const float4 *src = reinterpret_cast<const float4 *>(wimg + (size_t)o * WELEMS);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            if constexpr (WFULL) {
                w.v[i] = src[tid + i * CV_BLOCK];
            } else {
                int e = tid + i * CV_BLOCK;
                w.v[i] = src[e < WVEC ? e : 0];
            }
        }
        return w;
    };
    auto wstore = [&](int buf, const WFrag &w) {
        float4 *dst = reinterpret_cast<float4 *>(wl + buf * WELEMS);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            if constexpr (WFULL) {
                dst[tid + i * CV_BLOCK] = w.v[i];
            } else {
                int e = tid + i * CV_BLOCK;
                if (e < WVEC) dst[e] = w.v[i];
            }
        }
    };
    auto afloat = [&](const AFrag &a, int s) -> float {
        const float4 x = a.v[s >> 2];
        switch (s & 3) {
        case 0: return x.x;
        case 1: return x.y;
        case 2: return x.z;
        default: return x.w;
        }
    };
    auto compute = [&](int buf, const AFrag &a) {
        const float *wb = wl + buf * WELEMS;
#pragma unroll
        for (int sg = 0; sg < SG; ++sg) {
            float b[NT][VW];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tg = cs * NT + t;
                const float *p = wb + (((tg * SG + sg) * 4 + q) * 16 + r16) * VW;
                if constexpr (VW == 4) {
                    float4 x = *reinterpret_cast<const float4 *>(p);
                    b[t][0] = x.x;
                    b[t][1] = x.y;
                    b[t][2] = x.z;
                    b[t][3] = x.w;
                } else if constexpr (VW == 2) {
                    float2 x = *reinterpret_cast<const float2 *>(p);
                    b[t][0] = x.x;
                    b[t][1] = x.y;
                } else {
                    b[t][0] = p[0];
                }
            }
            // k-step outer, column tile inner: consecutive MFMAs hit different accumulators
#pragma unroll
            for (int v = 0; v < VW; ++v)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afloat(a, sg * VW + v), b[t][v],
                                                                  acc[t], 0, 0, 0);
        }
    };

    if (nact == 0) goto epilogue_;
    {
        bool hcur = false, hnext = false;
        WFrag w = wload(act[0]);
        AFrag acur = gather(act[0], hcur);
        wstore(0, w);
        __syncthreads();
        // Offset idx is computed from LDS buffer idx&1 while W / A of offset idx+1 are in
        // flight to registers; they land in the other buffer before the (single) barrier.
        for (int idx = 0; idx < nact; ++idx) {
            const int nxt = act[idx + 1 < nact ? idx + 1 : idx]; // clamped: last trip reloads
            w = wload(nxt);
            AFrag anext = gather(nxt, hnext);
            if (hcur) compute(idx & 1, acur);
            wstore((idx + 1) & 1, w);
            __syncthreads();
            acur = anext;
            hcur = hnext;
        }
    }
epilogue_:

    // ---- epilogue: (acc * scale + shift), relu, store.  C/D map: col = lane&15, row = q*4+reg
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = (cs * NT + t) * 16 + r16;
        if (col < cout) {
            const float sc = scale ? scale[col] : 1.f;
            const float sh = shift ? shift[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + rt * 16 + q * 4 + r;
                if (row < k_out) {
                    float v = acc[t][r];
                    if (scale) v = v * sc;
                    if (shift) v = v + sh;
                    if (relu) v = fmaxf(v, 0.f);
                    out[(size_t)row * cout + col] = v;
                }
            }
        }
    }
}


// -------------------------------------------------------------------------- split-slab fwd
// Thin layers (Cout <= 32: the first SECOND levels, 128->16, 16->16, 16->32) have few rows, few
// neighbours per row and little arithmetic: what costs there is the serial chain of offsets, each
// a global round trip, on a machine that the launch does not fill.  Here a workgroup owns ONE
// 16-row slab and its four waves split the reduction instead of the rows or columns:
//   * Cin >= 64: wave w takes a quarter of the input channels of every active offset,
//   * Cin <  64: wave w takes every fourth active offset,
// each wave walking its items on its own (operands straight into the MFMA fragments: lane
// (row, q) reads consecutive channels of its row, W[o] fragments stream from the image in L2,
// next item in flight behind this item's MFMAs; no barrier in the loop).  The four partial
// 16 x Cout tiles meet in LDS once and are summed in a fixed order: deterministic, 4x the waves
// of the slab form and 4x shorter chains.
template <int CIN, int COUT>
__global__ void __launch_bounds__(CV_BLOCK)
cv_ss_kernel(const float *__restrict__ feat, const float *__restrict__ wimg,
             const int *__restrict__ nbr, int k_cap, const int *__restrict__ k_dev, int kvol,
             int cin, int cout, const float *__restrict__ scale,
             const float *__restrict__ shift, int relu, float *__restrict__ out)
{
    const int k_out = lcf_count(k_cap, k_dev);
    constexpr int KL = CIN / 4;               // k values per lane quarter (image convention)
    constexpr int VW = KL < 4 ? KL : 4;
    constexpr int SG = KL / VW;
    constexpr int NT = COUT / 16;
    constexpr int WELEMS = CIN * COUT;
    constexpr bool KSPLIT = CIN >= 64;        // waves split the channels, else the offsets
    constexpr int KW = KSPLIT ? KL / CV_WAVES : KL; // k values per lane handled by one wave
    constexpr int A4 = KW >= 4 ? KW / 4 : 1;
    const int row0 = blockIdx.x * 16;
    if (row0 >= k_out) return;

    extern __shared__ __align__(16) unsigned char smem[];
    float *red = reinterpret_cast<float *>(smem);                          // [4][16][COUT]
    int *nl = reinterpret_cast<int *>(smem + CV_WAVES * 16 * COUT * 4);    // [16][kvol]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15;
    const int q = lane >> 4;
    // (scale | shift of the epilogue are requested with the table rows: a thread's column there is
    // tid % COUT whatever the row, and read in the epilogue they were one more round trip at the
    // end of every workgroup's life)
    static_assert(CV_BLOCK % COUT == 0, "a thread keeps its column");
    const int ecol = tid % COUT;
    const float esc = (scale && ecol < cout) ? scale[ecol] : 1.f;
    const float esh = (shift && ecol < cout) ? shift[ecol] : 0.f;
    {
        const long long gb = (long long)row0 * kvol, gl = (long long)k_out * kvol;
        for (int e = tid; e < 16 * kvol; e += CV_BLOCK) nl[e] = (gb + e < gl) ? nbr[gb + e] : -1;
    }
    __syncthreads();
    // offsets with a neighbour in the slab: lane o scans the 16 rows of offset o
    unsigned long long mask;
    {
        bool any = false;
        if (lane < kvol) {
#pragma unroll
            for (int r = 0; r < 16; ++r) any |= nl[r * kvol + lane] >= 0;
        }
        mask = __ballot(any);
        const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)mask);
        const unsigned mhi = __builtin_amdgcn_readfirstlane((unsigned)(mask >> 32));
        mask = ((unsigned long long)mhi << 32) | mlo;
    }
    // next item of this wave (-1: none): every active offset, or every fourth starting at #wave
    auto pop = [&]() -> int {
        if (mask == 0ull) return -1;
        const int o = __builtin_ctzll(mask);
        mask &= mask - 1ull;
        if constexpr (!KSPLIT) {
#pragma unroll
            for (int j = 1; j < CV_WAVES; ++j) mask &= mask - 1ull; // x & (x-1) of 0 stays 0
        }
        return o;
    };
    if constexpr (!KSPLIT) {
        for (int j = 0; j < wave; ++j) mask &= mask - 1ull;
    }
    const int ks0 = KSPLIT ? wave * KW : 0; // first k (within the lane's slice) of this wave

    const bool exact = (cin == CIN);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    struct AFrag { float4 v[A4]; };
    struct BFrag { float b[NT][KW]; };

    auto gather = [&](int o) -> AFrag {
        AFrag a;
        const int id = nl[r16 * kvol + o];
#pragma unroll
        for (int i = 0; i < A4; ++i) a.v[i] = zero4;
        if (id >= 0) {
            const float *src = feat + (size_t)id * cin + q * KL + ks0;
            if (exact) {
                if constexpr (KW >= 4) {
#pragma unroll
                    for (int i = 0; i < A4; ++i) a.v[i] = reinterpret_cast<const float4 *>(src)[i];
                } else if constexpr (KW == 2) {
                    float2 x = *reinterpret_cast<const float2 *>(src);
                    a.v[0].x = x.x;
                    a.v[0].y = x.y;
                } else {
                    a.v[0].x = src[0];
                }
            } else {
                float tmp[A4 * 4];
#pragma unroll
                for (int s = 0; s < A4 * 4; ++s)
                    tmp[s] = (s < KW && q * KL + ks0 + s < cin) ? src[s] : 0.f;
#pragma unroll
                for (int i = 0; i < A4; ++i)
                    a.v[i] = make_float4(tmp[4 * i], tmp[4 * i + 1], tmp[4 * i + 2], tmp[4 * i + 3]);
            }
        }
        return a;
    };
    auto bload = [&](int o) -> BFrag {
        BFrag w;
        const float *wo = wimg + (size_t)o * WELEMS;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (VW == 4) {
#pragma unroll
                for (int j = 0; j < KW / 4; ++j) {
                    const int sg = ks0 / 4 + j;
                    float4 x = *reinterpret_cast<const float4 *>(
                        wo + (((t * SG + sg) * 4 + q) * 16 + r16) * 4);
                    w.b[t][j * 4 + 0] = x.x;
                    w.b[t][j * 4 + 1] = x.y;
                    w.b[t][j * 4 + 2] = x.z;
                    w.b[t][j * 4 + 3] = x.w;
                }
            } else if constexpr (VW == 2) {
                float2 x = *reinterpret_cast<const float2 *>(wo + ((t * 4 + q) * 16 + r16) * 2);
                w.b[t][0] = x.x;
                w.b[t][1] = x.y;
            } else {
                w.b[t][0] = wo[(t * 4 + q) * 16 + r16];
            }
        }
        return w;
    };
    auto afloat = [&](const AFrag &a, int s) -> float {
        const float4 x = a.v[s >> 2];
        switch (s & 3) {
        case 0: return x.x;
        case 1: return x.y;
        case 2: return x.z;
        default: return x.w;
        }
    };

    f32x4 acc[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    int o = pop();
    if (o >= 0) {
        AFrag acur = gather(o);
        BFrag bcur = bload(o);
        for (;;) {
            const int on = pop(); // wave-uniform
            AFrag an = acur;
            BFrag bn = bcur;
            if (on >= 0) {
                an = gather(on);
                bn = bload(on);
            }
#pragma unroll
            for (int s = 0; s < KW; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t][s & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        afloat(acur, s), bcur.b[t][s], acc[t][s & 1], 0, 0, 0);
            if (on < 0) break;
            acur = an;
            bcur = bn;
        }
    }
    // ---- partial tiles -> LDS.  C/D map: col = lane&15, row = q*4+reg
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            red[(wave * 16 + q * 4 + r) * COUT + t * 16 + r16] = acc[t][0][r] + acc[t][1][r];
    __syncthreads();
    // ---- fixed-order sum, (x*scale + shift), relu, coalesced store
    for (int e = tid; e < 16 * COUT; e += CV_BLOCK) {
        const int r = e / COUT, c = e % COUT;
        const int row = row0 + r;
        if (row < k_out && c < cout) {
            float v = ((red[e] + red[16 * COUT + e]) + red[2 * 16 * COUT + e]) +
                      red[3 * 16 * COUT + e];
            if (scale) v *= esc;
            if (shift) v += esh;
            if (relu) v = fmaxf(v, 0.f);
            out[(size_t)row * cout + c] = v;
        }
    }
}

template <int CIN, int COUT>
static int cv_launch_ss(const float *feat, const float *wimg, const int *nbr, int k_out,
                        const int *k_dev, int kvol, int cin, int cout, const float *scale,
                        const float *shift, int relu, float *out, hipStream_t st)
{
    if constexpr (COUT <= 32) {
        const size_t lds = (size_t)CV_WAVES * 16 * COUT * 4 + (size_t)16 * kvol * 4;
        const int nb = (k_out + 15) / 16;
        cv_ss_kernel<CIN, COUT><<<nb, CV_BLOCK, lds, st>>>(feat, wimg, nbr, k_out, k_dev, kvol, cin,
                                                         cout, scale, shift, relu, out);
        LCF_CHECK_LAUNCH();
        return LCF_OK;
    }
    return LCF_ERR_UNSUPPORTED;
}

// -------------------------------------------------------------------------- thin-tile fwd
// The same thin layers (Cout <= 32), third form.  What bounds the split-slab kernel on 128 -> 16 is
// neither its chain nor the gathers (splitting offsets instead of channels over the waves changes
// nothing: 22.4 vs 22.8 us) but operands per useful row: a 16-row slab has 1.2 rows with a
// neighbour at an average active offset, so every (slab, offset) item streams a whole W[o]
// (8 KB at 128 x 16: ~190 MB of L2 traffic per launch) and runs full 16-row MFMAs for it.  Here a
// workgroup owns a TILE of up to 64 output rows and compacts, per offset, the rows that have a
// neighbour (as the tile-compacted kernel does): an item is 16 COMPACTED rows of one offset, W[o]
// is fetched once per item of ~4x more useful rows, and the MFMA work drops by the same factor.
// Items go round-robin to the four waves, each walking its items alone (operands straight into
// the MFMA fragments, next item in flight behind this item's MFMAs); an item's 16 x Cout result
// is added into the wave's OWN copy of the output tile in LDS, and the four copies are summed in
// a fixed order at the end: deterministic, no atomics.
// What it buys is modest (128 -> 16: 22.6 -> 19.9 us; 16 -> 32: 18.8 -> 15.4): the layer's 27
// weight matrices (221 KB at 128 x 16) are larger than a tile's rows (32 KB), every tile needs
// all of them, and an item still fetches its own W[o]: ~120 MB of L2 reads per launch instead of
// ~190 MB.  The floor of any tile-owns-rows form is tiles x 221 KB.
#define TN_TR 64
template <int CIN, int COUT, int NW>
__global__ void __launch_bounds__(NW * 64)
cv_tn_kernel(const float *__restrict__ feat, const float *__restrict__ wimg,
             const int *__restrict__ nbr, int k_cap, const int *__restrict__ k_dev, int kvol,
             int cin, int cout, const float *__restrict__ scale,
             const float *__restrict__ shift, int relu, float *__restrict__ out, int tr)
{
    const int k_out = lcf_count(k_cap, k_dev);
    constexpr int KL = CIN / 4;               // k values per lane quarter (image convention)
    constexpr int VW = KL < 4 ? KL : 4;
    constexpr int SG = KL / VW;
    constexpr int NT = COUT / 16;
    constexpr int WELEMS = CIN * COUT;
    constexpr int A4 = KL >= 4 ? KL / 4 : 1;
    constexpr int OSTR = COUT + 1;            // output-tile row stride (floats)
    constexpr int OTE = (TN_TR + 1) * OSTR;   // one wave's tile (+ the dump row of padding rows)
    const int row0 = blockIdx.x * tr;
    if (row0 >= k_out) return;

    extern __shared__ __align__(16) unsigned char smem[];
    float *ot = reinterpret_cast<float *>(smem);                       // [4][TN_TR + 1][OSTR]
    int *nl = reinterpret_cast<int *>(ot + NW * OTE);            // [TN_TR][kvol]
    unsigned *ent = reinterpret_cast<unsigned *>(nl + TN_TR * kvol);   // [kvol][TN_TR]
    int *cnt = reinterpret_cast<int *>(ent + TN_TR * kvol);            // [64] rows per offset
    int *ibase = cnt + 64;                                             // [64] first item of offset
    int *nitems_p = ibase + 64;
    int *iof = nitems_p + 4;                                           // [kvol * 4] offset of item
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15;
    const int q = lane >> 4;
    {
        const int total = tr * kvol;
        const long long gb = (long long)row0 * kvol, gl = (long long)k_out * kvol;
        for (int e = tid; e < total; e += NW * 64) nl[e] = (gb + e < gl) ? nbr[gb + e] : -1;
        for (int e = tid; e < NW * OTE; e += NW * 64) ot[e] = 0.f;
    }
    __syncthreads();
    // ---- compaction: wave w takes offsets w, w + 4, ...; lane = tile row (tr <= 64)
    for (int o = wave; o < kvol; o += NW) {
        const int id = lane < tr ? nl[lane * kvol + o] : -1;
        const unsigned long long m = __ballot(id >= 0);
        const int c = __popcll(m);
        if (id >= 0)
            ent[o * TN_TR + __popcll(m & ((1ull << lane) - 1ull))] =
                ((unsigned)id << 7) | (unsigned)lane;
        if (lane == 0) cnt[o] = c;
    }
    __syncthreads();
    if (wave == 0) { // items = 16-row groups, offsets ascending
        const int c = lane < kvol ? cnt[lane] : 0;
        const int ng = (c + 15) >> 4;
        int incl = ng;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        ibase[lane] = incl - ng;
        for (int g = 0; g < ng; ++g) iof[incl - ng + g] = lane;
        if (lane == 63) *nitems_p = incl;
    }
    __syncthreads();
    const int nitems = __builtin_amdgcn_readfirstlane(*nitems_p);
    float *myot = ot + wave * OTE;

    const bool exact = (cin == CIN);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    struct AFrag { float4 v[A4]; };
    struct BFrag { float b[NT][KL]; };
    // item -> (offset, group), wave-uniform
    auto locate = [&](int it, int &o, int &g) {
        o = __builtin_amdgcn_readfirstlane(iof[it]);
        g = it - __builtin_amdgcn_readfirstlane(ibase[o]);
    };
    auto gather = [&](int o, int g, unsigned &e_out) -> AFrag {
        AFrag a;
        const int p = g * 16 + r16;
        const bool has = p < cnt[o];
        const unsigned e = has ? ent[o * TN_TR + p] : (unsigned)TN_TR; // padding: the dump row
        e_out = e;
#pragma unroll
        for (int i = 0; i < A4; ++i) a.v[i] = zero4;
        if (has) {
            const float *src = feat + (size_t)(e >> 7) * cin + q * KL;
            if (exact) {
                if constexpr (KL >= 4) {
#pragma unroll
                    for (int i = 0; i < A4; ++i) a.v[i] = reinterpret_cast<const float4 *>(src)[i];
                } else if constexpr (KL == 2) {
                    const float2 x = *reinterpret_cast<const float2 *>(src);
                    a.v[0].x = x.x;
                    a.v[0].y = x.y;
                } else {
                    a.v[0].x = src[0];
                }
            } else {
                float tmp[A4 * 4];
#pragma unroll
                for (int s = 0; s < A4 * 4; ++s)
                    tmp[s] = (s < KL && q * KL + s < cin) ? src[s] : 0.f;
#pragma unroll
                for (int i = 0; i < A4; ++i)
                    a.v[i] = make_float4(tmp[4 * i], tmp[4 * i + 1], tmp[4 * i + 2], tmp[4 * i + 3]);
            }
        }
        return a;
    };
    auto bload = [&](int o) -> BFrag {
        BFrag w;
        const float *wo = wimg + (size_t)o * WELEMS;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (VW == 4) {
#pragma unroll
                for (int j = 0; j < KL / 4; ++j) {
                    const float4 x = *reinterpret_cast<const float4 *>(
                        wo + (((t * SG + j) * 4 + q) * 16 + r16) * 4);
                    w.b[t][j * 4 + 0] = x.x;
                    w.b[t][j * 4 + 1] = x.y;
                    w.b[t][j * 4 + 2] = x.z;
                    w.b[t][j * 4 + 3] = x.w;
                }
            } else if constexpr (VW == 2) {
                const float2 x = *reinterpret_cast<const float2 *>(wo + ((t * 4 + q) * 16 + r16) * 2);
                w.b[t][0] = x.x;
                w.b[t][1] = x.y;
            } else {
                w.b[t][0] = wo[(t * 4 + q) * 16 + r16];
            }
        }
        return w;
    };
    auto afloat = [&](const AFrag &a, int s) -> float {
        const float4 x = a.v[s >> 2];
        switch (s & 3) {
        case 0: return x.x;
        case 1: return x.y;
        case 2: return x.z;
        default: return x.w;
        }
    };

    int it = wave;
    if (it < nitems) {
        int o, g;
        locate(it, o, g);
        unsigned ecur;
        AFrag acur = gather(o, g, ecur);
        BFrag bcur = bload(o);
        for (;;) {
            const int itn = it + NW;
            const bool more = itn < nitems; // wave-uniform
            AFrag an = acur;
            BFrag bn = bcur;
            unsigned en = ecur;
            if (more) {
                int on, gn;
                locate(itn, on, gn);
                an = gather(on, gn, en);
                bn = bload(on);
            }
            f32x4 acc[NT][2];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int s = 0; s < KL; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t][s & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        afloat(acur, s), bcur.b[t][s], acc[t][s & 1], 0, 0, 0);
            // C/D map: col = lane & 15, group row = q * 4 + reg -> the tile row of that entry
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned er = __shfl(ecur, q * 4 + r, 64); // entry of group row q*4+r
                float *dst = myot + (er & 127u) * OSTR + r16;
#pragma unroll
                for (int t = 0; t < NT; ++t) dst[t * 16] += acc[t][0][r] + acc[t][1][r];
            }
            if (!more) break;
            it = itn;
            acur = an;
            bcur = bn;
            ecur = en;
        }
    }
    __syncthreads();
    // ---- fixed-order sum of the waves' copies, (x*scale + shift), relu, coalesced store
    for (int e = tid; e < tr * COUT; e += NW * 64) {
        const int r = e / COUT, c = e % COUT;
        const int row = row0 + r;
        if (row < k_out && c < cout) {
            const int a = r * OSTR + c;
            float v = ot[a];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += ot[w * OTE + a]; // fixed order
            if (scale) v *= scale[c];
            if (shift) v += shift[c];
            if (relu) v = fmaxf(v, 0.f);
            out[(size_t)row * cout + c] = v;
        }
    }
}

static size_t tn_lds_bytes(int COUT, int kvol, int nw)
{
    return (size_t)nw * (TN_TR + 1) * (COUT + 1) * 4 + (size_t)2 * TN_TR * kvol * 4 +
           (size_t)(64 + 64 + 4 + 4 * kvol) * 4;
}

template <int CIN, int COUT, int NW>
static int cv_launch_tn_n(const float *feat, const float *wimg, const int *nbr, int k_out,
                          const int *k_dev, int kvol, int cin, int cout, const float *scale,
                          const float *shift, int relu, float *out, hipStream_t st)
{
    const size_t lds = tn_lds_bytes(COUT, kvol, NW);
    if (lds > 160 * 1024) return LCF_ERR_UNSUPPORTED;
    {
        static size_t lds_set = 0;
        static std::mutex lds_mu;
        std::lock_guard<std::mutex> lock(lds_mu);
        if (lds > 64 * 1024 && lds > lds_set) {
            if (hipFuncSetAttribute((const void *)cv_tn_kernel<CIN, COUT, NW>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return LCF_ERR_LAUNCH;
            lds_set = lds;
        }
    }
    const LcfOpts op = lcf_opts();
    const int rows = op.rows_hint > 0 && op.rows_hint < k_out ? op.rows_hint : k_out;
    // tile height: a whole number of tiles per CU where the launch is small
    int tr = TN_TR;
    const int m = (rows + 256 * TN_TR - 1) / (256 * TN_TR);
    if (m >= 1) {
        tr = (rows + 256 * m - 1) / (256 * m);
        if (tr < 16) tr = 16;
        if (tr > TN_TR) tr = TN_TR;
    }
    const int nb = (k_out + tr - 1) / tr;
    cv_tn_kernel<CIN, COUT, NW><<<nb, NW * 64, lds, st>>>(feat, wimg, nbr, k_out, k_dev, kvol, cin,
                                                        cout, scale, shift, relu, out, tr);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

template <int CIN, int COUT>
static int cv_launch_tn(const float *feat, const float *wimg, const int *nbr, int k_out,
                        const int *k_dev, int kvol, int cin, int cout, const float *scale,
                        const float *shift, int relu, float *out, hipStream_t st)
{
    if constexpr (COUT <= 32) {
        // four waves: 8 and 16 waves per workgroup (shorter chains per wave, more private tiles)
        // were measured and lose (128 -> 16: 19.9 / 22.5 / 83 us)
        return cv_launch_tn_n<CIN, COUT, 4>(feat, wimg, nbr, k_out, k_dev, kvol, cin, cout, scale,
                                            shift, relu, out, st);
    }
    return LCF_ERR_UNSUPPORTED;
}

// -------------------------------------------------------------------------- tile-compacted fwd
// Second-generation forward ("tc"): the workgroup owns a tile of TR (32 or 64) output rows.
// For every kernel offset it COMPACTS the rows of the tile that have a neighbour at that offset
// (wave ballot + popcount) and runs the MFMAs only over the resulting ceil(c/16) dense groups of
// 16 gathered rows, instead of over every 16-row slab that has at least one neighbour: MFMA work
// ~ pairs*Cin*Cout rounded up to 16-row groups per (tile, offset), 2-4x less than the slab form
// on the SECOND rulebooks.
//   * One STEP = one active offset: all its groups are gathered into LDS (double buffered; the
//     next offset's rows are prefetched to registers behind this offset's MFMAs), one barrier
//     per step.
//   * The waves of the workgroup split the OUTPUT COLUMNS and share the gathered rows.
//   * W[o] is NOT staged in LDS: the weight image is stored in fragment order, so each wave
//     streams its own column slice of W[o] straight from L2 into registers with coalesced
//     16-byte loads, one offset ahead.
//   * Each group's 16 x Cout result is added into an fp32 output tile in LDS.  Every element is
//     owned by exactly one wave and touched in ascending offset order, so the result is
//     run-to-run deterministic and equals sum_o(dot_o), which is how the reference's per-offset
//     GEMM + scatter-add sums.
// Row counts may live on the device (k_dev): the grid is sized by the capacity and workgroups
// past the actual count exit at once, so the launch needs no host synchronisation.

// Optional dense target of the epilogue (SparseConvTensor.dense() fused into the last conv of the
// stack): with coors != NULL the tile is written to
// out[b, c, z, y, x] of a [batch, cout, D, H, W] tensor instead of out[row, c].
struct TcDense {
    const int *coors; // [k_out, 4] (b, z, y, x) of the output rows, or NULL
    int batch, D, H, W;
};

template <int CIN, int COUT, int TR>
struct TcCfg {
    static constexpr int NW = (COUT / 16 >= 4) ? 4 : COUT / 16; // waves (column split)
    static constexpr int NT = COUT / 16 / NW;                   // column tiles per wave
    static constexpr int THREADS = NW * 64;
    static constexpr int MAXG = TR / 16;              // groups per offset at most
    static constexpr int KL = CIN / 4;
    static constexpr int VW = KL < 4 ? KL : 4;
    static constexpr int SG = KL / VW;
    static constexpr int CH = CIN / 4;                // float4 chunks per gathered row
    static constexpr int ACH = 16 * CH;               // chunks per 16-row group
    static constexpr int APT = (ACH + THREADS - 1) / THREADS;
    static constexpr int ASTR = CIN + 4;              // padded A row stride (floats)
    static constexpr int WELEMS = CIN * COUT;
    static constexpr int OSTR = COUT + 4;             // padded output-tile row stride
};

static size_t tc_max(size_t a, size_t b) { return a > b ? a : b; }

static size_t tc_lds_bytes(int CIN, int COUT, int TR, int kvol, bool bf16 = false, int nsets = 1,
                           int planes = 1)
{
    // one wave set double-buffers its rows; two sets, and the three-plane form, do not
    const int abufs = (nsets == 1 && planes < 3) ? 2 : 1;
    size_t b = bf16 ? (size_t)nsets * abufs * planes * TR * (CIN + 8) * 2 // A rows, bf16
                    : (size_t)nsets * abufs * TR * (CIN + 4) * 4;         // A rows of one offset
    b += tc_max((size_t)nsets * (TR + 1) * (COUT + 4) * 4, (size_t)TR * kvol * 4); // tiles | nbr
    b = (b + 15) & ~(size_t)15;
    b += (size_t)TR * kvol * 4;             // item entries: pad16(cnt) <= TR per offset
    b += (size_t)64 * 4 * 3 + 16;                                         // cnt, gbase, soff, n
    return b;
}

#define TC_THREADS(COUT) (((COUT) / 16 >= 4 ? 4 : (COUT) / 16) * 64)

// (tc_build_steps: tc_plan.h, shared with rulebook_chain.hip)
// BF16: operands rounded to bf16 (features when they enter LDS, weights in a bf16 image in
// v_mfma_f32_16x16x32_bf16 fragment order), fp32 accumulation and fp32 output: the training
// variant BASELINE.json configs[2] names.  Needs Cin a multiple of 32.
// NP = 2 (with BF16): fp32 operands split into two bf16 planes, x = hi + lo with hi = bf16(x),
// lo = bf16(x - hi) (|x - hi - lo| <= 2^-18 |x|), and the product taken as
// lo_a*hi_w + hi_a*lo_w + hi_a*hi_w on the bf16 matrix cores with fp32 accumulation: three
// v_mfma_f32_16x16x32_bf16 (48 cycles) where fp32 operands need eight v_mfma_f32_16x16x4_f32
// (256 cycles), and unlike those they overlap the loop's vector instructions.  Dropped terms are
// <= 3 * 2^-18 of |a||w| per product (1.1e-5; fp32 products: 6e-8).
// NP = 3: three planes, x = hi + mid + lo EXACTLY (8 + 8 + 8 significand bits; each residual is
// exact in fp32), and the eight products down to 2^-24 |a||w| (hi*hi, hi*mid, mid*hi, hi*lo,
// mid*mid, lo*hi, mid*lo, lo*mid; dropped: lo*lo, 2^-32): what is left is smaller than the
// rounding of fp32 operands' own products, at 8 x 16 = 128 matrix-core cycles per 16x16x32 block
// instead of 256 fp32-MFMA cycles.
// NSETS = 2: a second set of waves takes every other step (offset), with its own row buffer and
// its own output tile; the two tiles are summed (set 0 + set 1) in the epilogue.  Halves the
// serial chain of a tile, which is what bounds layers with few tiles per CU.
template <int CIN, int COUT, int TR, bool EXACT, bool BF16, int NSETS, int NP>
__global__ void __launch_bounds__(TC_THREADS(COUT) * NSETS)
cv_tc_kernel(const float *__restrict__ feat, const float *__restrict__ wimg,
             const int *__restrict__ nbr, int k_cap, const int *__restrict__ k_dev, int kvol,
             int cin, int cout, const float *__restrict__ scale,
             const float *__restrict__ shift, int relu, float *__restrict__ out, int tr,
             TcDense dn, const int *__restrict__ plan, int plan_stride)
{
    using C = TcCfg<CIN, COUT, TR>;
    constexpr int THREADS = C::THREADS, NT = C::NT, KL = C::KL, VW = C::VW, SG = C::SG;
    constexpr int CH = C::CH, ACH = C::ACH, APT = C::APT, ASTR = C::ASTR, OSTR = C::OSTR;
    constexpr int WELEMS = C::WELEMS, MAXG = C::MAXG;
    constexpr bool AFULL = (ACH % THREADS) == 0;
    constexpr int DUMP = TR; // output-tile row that absorbs a group's padding entries
    constexpr int ABUFS = (NSETS == 1 && NP < 3) ? 2 : 1; // row buffers per wave set
    constexpr int TT = THREADS * NSETS;         // threads of the workgroup
    constexpr int NWT = C::NW * NSETS;          // waves of the workgroup
    constexpr int OTE = (TR + 1) * OSTR;        // floats of one output tile

    const int k_out = lcf_count(k_cap, k_dev);
    // tr <= TR rows per tile (host: chosen so that the tiles divide evenly over the CUs); every
    // buffer is laid out for TR rows
    // (tile = blockIdx.x: workgroups go round-robin to the 8 XCDs.  Giving XCD x the x-th
    // contiguous eighth of the tiles instead, so that one L2 holds a neighbourhood's rows, was
    // measured: 7 % slower on one frame (38k rows: the features fit the L2s either way and
    // round-robin balances the per-tile work) and no different at batch 8 (294k rows).)
    const int row0 = blockIdx.x * tr;
    if (row0 >= k_out) return;
#if defined(TC_ABLATE) && TC_ABLATE == 3 // timing ablation: launch + row count only
    if (k_out >= 0) return;
#endif

    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int ASTRH = CIN + 8;                               // bf16 A row stride (halfs)
    constexpr int KS = CIN >= 32 ? CIN / 32 : 1;                 // bf16 k-steps of 32
    constexpr int APL = TR * ASTRH;                              // halfs of one bf16 plane
    constexpr size_t ABYTES = BF16 ? (size_t)NSETS * ABUFS * NP * APL * 2
                                   : (size_t)NSETS * ABUFS * TR * ASTR * 4;
    float *ot0 = reinterpret_cast<float *>(smem + ABYTES);       // [NSETS][TR + dump row][OSTR]
    int *nl = reinterpret_cast<int *>(ot0);                      // [TR][kvol], dead before ot lives
    size_t otb = (size_t)NSETS * OTE * 4, nlb = (size_t)TR * kvol * 4;
    size_t off = ABYTES + (otb > nlb ? otb : nlb);
    off = (off + 15) & ~(size_t)15;
    unsigned *ent = reinterpret_cast<unsigned *>(smem + off);    // [items][16] entries
    int *cnt = reinterpret_cast<int *>(ent + TR * kvol);         // [64] rows per offset
    int *gbase = cnt + 64;                                       // [64] first item of an offset
    int *soff = gbase + 64;                                      // [64] offset of step s
    int *nsteps_p = soff + 64;

    const int tidw = threadIdx.x;          // thread of the workgroup
    const int lane = tidw & 63;
    const int wavew = __builtin_amdgcn_readfirstlane(tidw >> 6); // wave of the workgroup (SGPR)
    const int set = wavew / C::NW;         // wave set (0 when NSETS == 1)
    const int wave = wavew % C::NW;        // = column slice
    const int tid = tidw - set * THREADS;  // thread of the set
    const int r16 = lane & 15;
    const int q = lane >> 4;
    float *al = reinterpret_cast<float *>(smem) + set * ABUFS * TR * ASTR;    // [ABUFS][TR][ASTR]
    __bf16 *alh = reinterpret_cast<__bf16 *>(smem) + set * ABUFS * NP * APL; // [ABUFS][NP][TR][ASTRH]
    float *ot = ot0 + set * OTE;                                 // this set's output tile

    if (plan) {
        // the compaction of this tile was done once per rulebook (lcf_conv_tile_plan: shared by the
        // convs of one indice_key, built beside the feature path): copy its image into LDS
        const int4 *src = reinterpret_cast<const int4 *>(plan + (size_t)blockIdx.x * plan_stride);
        int4 *dst = reinterpret_cast<int4 *>(ent);
        for (int e = tidw; e < plan_stride / 4; e += TT) dst[e] = src[e];
        __syncthreads();
    } else {
        tc_build_steps<NWT, TT>(nbr, row0, tr, kvol, k_out, nl, ent, cnt, gbase, soff, nsteps_p,
                                DUMP);
    }
#if defined(TC_ABLATE) && TC_ABLATE == 5 // timing ablation: ... + counts, steps, entries
    if (k_out >= 0) return;
#endif
    for (int e = tidw; e < NSETS * OTE; e += TT) ot0[e] = 0.f; // nl is dead now
    const int n_steps = __builtin_amdgcn_readfirstlane(*nsteps_p);
    __syncthreads();
#if defined(TC_ABLATE) && TC_ABLATE == 1 // timing ablation: the set-up alone (results wrong)
    if (n_steps >= 0) return;
#endif

    struct AReg { float4 v[MAXG][APT]; };
    struct BReg {
        float b[NT][BF16 ? 1 : KL];
        bf16x8 h[NT][BF16 ? KS * NP : 1]; // [ks * NP + plane]
    };

    // rows of the ng groups starting at item i0: thread -> (row r of a group, 16-byte chunk cc)
    auto agather = [&](int i0, int ng) -> AReg {
        AReg a;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
#pragma unroll
            for (int x = 0; x < APT; ++x) {
                const int ch = tid + x * THREADS;
                if (g < ng && (AFULL || ch < ACH)) { // astore reads a.v[g][x] under the same test
                    const int r = ch / CH, cc = ch % CH;
                    const unsigned e = ent[(i0 + g) * 16 + r];
                    const float *src = feat + (size_t)(e >> 7) * cin + cc * 4;
                    if constexpr (EXACT) {
                        a.v[g][x] = *reinterpret_cast<const float4 *>(src);
                    } else {
                        const int k0 = cc * 4;
                        float t0 = k0 + 0 < cin ? src[0] : 0.f;
                        float t1 = k0 + 1 < cin ? src[1] : 0.f;
                        float t2 = k0 + 2 < cin ? src[2] : 0.f;
                        float t3 = k0 + 3 < cin ? src[3] : 0.f;
                        a.v[g][x] = make_float4(t0, t1, t2, t3);
                    }
                }
            }
        }
        return a;
    };
    auto astore = [&](int buf, const AReg &a, int ng) {
        float *dst = al + buf * TR * ASTR;
        __bf16 *dsth = alh + buf * NP * APL;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
#pragma unroll
            for (int x = 0; x < APT; ++x) {
                const int ch = tid + x * THREADS;
                if (g < ng && (AFULL || ch < ACH)) {
                    const int r = ch / CH, cc = ch % CH;
                    if constexpr (BF16) {
                        const float4 v = a.v[g][x];
                        bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                        *reinterpret_cast<bf16x4 *>(dsth + (g * 16 + r) * ASTRH + cc * 4) = h;
                        if constexpr (NP >= 2) {
                            const float4 r1 = make_float4(v.x - (float)h[0], v.y - (float)h[1],
                                                          v.z - (float)h[2], v.w - (float)h[3]);
                            bf16x4 l = {(__bf16)r1.x, (__bf16)r1.y, (__bf16)r1.z, (__bf16)r1.w};
                            *reinterpret_cast<bf16x4 *>(dsth + APL + (g * 16 + r) * ASTRH +
                                                        cc * 4) = l;
                            if constexpr (NP == 3) {
                                bf16x4 l2 = {(__bf16)(r1.x - (float)l[0]),
                                             (__bf16)(r1.y - (float)l[1]),
                                             (__bf16)(r1.z - (float)l[2]),
                                             (__bf16)(r1.w - (float)l[3])};
                                *reinterpret_cast<bf16x4 *>(dsth + 2 * APL + (g * 16 + r) * ASTRH +
                                                            cc * 4) = l2;
                            }
                        }
                    } else {
                        *reinterpret_cast<float4 *>(dst + (g * 16 + r) * ASTR + cc * 4) =
                            a.v[g][x];
                    }
                }
            }
        }
    };
    // this wave's column slice of W[o], in fragment order, straight from the image (L2)
    auto bload = [&](int o) -> BReg {
        BReg w;
        if constexpr (BF16) {
            // image [plane][o][tg][ks][q][r16][8]: lane (q, r16) holds B[k = 32ks + 8q + j][col]
            const __bf16 *wo = reinterpret_cast<const __bf16 *>(wimg) + (size_t)o * WELEMS;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tg = wave * NT + t;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl)
                        w.h[t][ks * NP + pl] = *reinterpret_cast<const bf16x8 *>(
                            wo + (size_t)pl * kvol * WELEMS +
                            (((tg * KS + ks) * 4 + q) * 16 + r16) * 8);
            }
            return w;
        }
        const float *wo = wimg + (size_t)o * WELEMS;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tg = wave * NT + t;
#pragma unroll
            for (int sg = 0; sg < SG; ++sg) {
                const float *p = wo + (((tg * SG + sg) * 4 + q) * 16 + r16) * VW;
                if constexpr (VW == 4) {
                    float4 x = *reinterpret_cast<const float4 *>(p);
                    w.b[t][sg * 4 + 0] = x.x;
                    w.b[t][sg * 4 + 1] = x.y;
                    w.b[t][sg * 4 + 2] = x.z;
                    w.b[t][sg * 4 + 3] = x.w;
                } else if constexpr (VW == 2) {
                    float2 x = *reinterpret_cast<const float2 *>(p);
                    w.b[t][0] = x.x;
                    w.b[t][1] = x.y;
                } else {
                    w.b[t][0] = p[0];
                }
            }
        }
        return w;
    };
    // group g of the step in LDS buffer abuf; item index i names its entries
    auto compute = [&](int abuf, int g, const BReg &w, int i) {
        const float *ap = al + (abuf * TR + g * 16 + r16) * ASTR + q * KL;
        float a[KL];
        bf16x8 ah[KS * NP];
        if constexpr (BF16) {
            // lane (r16, q) holds A[row r16][k = 32ks + 8q + j]
            const __bf16 *aph = alh + abuf * NP * APL + (g * 16 + r16) * ASTRH + q * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    ah[ks * NP + pl] = *reinterpret_cast<const bf16x8 *>(aph + pl * APL + ks * 32);
        } else if constexpr (KL >= 4) {
#pragma unroll
            for (int j = 0; j < KL / 4; ++j) {
                float4 x = reinterpret_cast<const float4 *>(ap)[j];
                a[4 * j] = x.x;
                a[4 * j + 1] = x.y;
                a[4 * j + 2] = x.z;
                a[4 * j + 3] = x.w;
            }
        } else if constexpr (KL == 2) {
            float2 x = *reinterpret_cast<const float2 *>(ap);
            a[0] = x.x;
            a[1] = x.y;
        } else {
            a[0] = ap[0];
        }
        // output rows of this lane's 4 accumulator rows (C/D map: row = q*4 + reg)
        const uint4 e4 = *reinterpret_cast<const uint4 *>(ent + i * 16 + q * 4);
        const int orow[4] = {(int)(e4.x & 127u), (int)(e4.y & 127u), (int)(e4.z & 127u),
                             (int)(e4.w & 127u)};
        // one accumulator chain per column tile: an MFMA whose C operand is the previous
        // one's result issues back to back, and a second chain would cost four VALU adds per
        // group to merge (VALU time is what this kernel is short of)
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (BF16) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if constexpr (NP == 2) { // lo*hi + hi*lo, then hi*hi
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            ah[ks * 2 + 1], w.h[t][ks * 2], acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            ah[ks * 2], w.h[t][ks * 2 + 1], acc[t], 0, 0, 0);
                    }
                    if constexpr (NP == 3) { // smallest terms first; (plane a, plane w)
                        constexpr int TA[7] = {2, 1, 2, 1, 0, 1, 0};
                        constexpr int TW[7] = {1, 2, 0, 1, 2, 0, 1};
#pragma unroll
                        for (int i = 0; i < 7; ++i)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                ah[ks * 3 + TA[i]], w.h[t][ks * 3 + TW[i]], acc[t], 0, 0, 0);
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks * NP], w.h[t][ks * NP],
                                                                     acc[t], 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int s = 0; s < KL; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], w.b[t][s], acc[t], 0, 0, 0);
        }
        // tile[orow][cols of this wave] += group result (col = lane&15); exclusive ownership
        float cur[4][NT];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                cur[r][t] = ot[orow[r] * OSTR + wave * NT * 16 + r16 + t * 16];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                ot[orow[r] * OSTR + wave * NT * 16 + r16 + t * 16] = cur[r][t] + acc[t][r];
    };

    if (n_steps > 0) {
        // set j walks steps j, j + NSETS, ...; the workgroup walks them in lockstep rounds.
        // Step metadata is wave-uniform: kept in SGPRs so that the tests below are scalar
        // branches, not exec-mask updates (fp32 MFMAs and VALU work do not overlap on a SIMD,
        // see DESIGN.md: every VALU instruction of this loop is paid in full).
        const int n_rounds = (n_steps + NSETS - 1) / NSETS;
        int s = set;
        int o = __builtin_amdgcn_readfirstlane(soff[s < n_steps ? s : 0]);
        int i0 = __builtin_amdgcn_readfirstlane(gbase[o]);
        int ng = s < n_steps ? __builtin_amdgcn_readfirstlane((cnt[o] + 15) >> 4) : 0;
        BReg b0 = bload(o), b1;
        {
            AReg a0 = agather(i0, ng);
            astore(0, a0, ng);
        }
        __syncthreads();
        // one round: compute step s from bcur while step s + NSETS loads into bnext
        auto round = [&](int u, const BReg &bcur, BReg &bnext) {
            const bool more = s + NSETS < n_steps; // uniform over the set
            AReg an;
            int on = o, i0n = i0, ngn = 0;
            if (more) {
                on = __builtin_amdgcn_readfirstlane(soff[s + NSETS]);
                i0n = __builtin_amdgcn_readfirstlane(gbase[on]);
                ngn = __builtin_amdgcn_readfirstlane((cnt[on] + 15) >> 4);
                bnext = bload(on);
                an = agather(i0n, ngn);
            }
            const int abuf = ABUFS == 2 ? (u & 1) : 0;
#pragma unroll
            for (int g = 0; g < MAXG; ++g)
                if (g < ng) compute(abuf, g, bcur, i0 + g);
            if constexpr (ABUFS == 1) __syncthreads(); // single buffer: everyone done reading
            if (more) astore(ABUFS == 2 ? ((u + 1) & 1) : 0, an, ngn);
            __syncthreads();
            o = on;
            i0 = i0n;
            ng = ngn;
            s += NSETS;
        };
        for (int u = 0; u < n_rounds; u += 2) { // W fragments alternate between b0 and b1
            round(u, b0, b1);
            if (u + 1 < n_rounds) round(u + 1, b1, b0);
        }
    }

#if defined(TC_ABLATE) && TC_ABLATE == 2 // timing ablation: no epilogue (results wrong)
    if (k_out >= 0) return;
#endif
    // ---- epilogue: (x*scale + shift), relu, coalesced store of the tile
    if (cout == COUT) {
        constexpr int C4 = COUT / 4;
        for (int e = tidw; e < tr * C4; e += TT) {
            const int r = e / C4, c4 = e % C4;
            const int row = row0 + r;
            if (row < k_out) {
                float4 v = *reinterpret_cast<const float4 *>(ot0 + r * OSTR + c4 * 4);
                if constexpr (NSETS == 2) {
                    const float4 v1 =
                        *reinterpret_cast<const float4 *>(ot0 + OTE + r * OSTR + c4 * 4);
                    v.x += v1.x; v.y += v1.y; v.z += v1.z; v.w += v1.w;
                }
                if (scale) {
                    const float4 s = *reinterpret_cast<const float4 *>(scale + c4 * 4);
                    v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
                }
                if (shift) {
                    const float4 s = *reinterpret_cast<const float4 *>(shift + c4 * 4);
                    v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
                }
                if (relu) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
                    v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                }
                if (dn.coors) {
                    const int4 q = reinterpret_cast<const int4 *>(dn.coors)[row];
                    if (q.x >= 0 && q.x < dn.batch && q.y >= 0 && q.y < dn.D && q.z >= 0 &&
                        q.z < dn.H && q.w >= 0 && q.w < dn.W) {
                        const size_t svol = (size_t)dn.D * dn.H * dn.W;
                        float *d = out + ((size_t)q.x * cout + c4 * 4) * svol +
                                   ((size_t)q.y * dn.H + q.z) * dn.W + q.w;
                        d[0] = v.x;
                        d[svol] = v.y;
                        d[2 * svol] = v.z;
                        d[3 * svol] = v.w;
                    }
                } else {
                    *reinterpret_cast<float4 *>(out + (size_t)row * cout + c4 * 4) = v;
                }
            }
        }
    } else {
        for (int e = tidw; e < tr * cout; e += TT) {
            const int r = e / cout, c = e % cout;
            const int row = row0 + r;
            if (row < k_out) {
                float v = ot0[r * OSTR + c];
                if constexpr (NSETS == 2) v += ot0[OTE + r * OSTR + c];
                if (scale) v *= scale[c];
                if (shift) v += shift[c];
                if (relu) v = fmaxf(v, 0.f);
                out[(size_t)row * cout + c] = v;
            }
        }
    }
}

// ---- tile plans: the set-up of every tile of a rulebook, computed once (lcf_conv_tile_plan) and
// read by every tile-compacted launch over that rulebook (the two or three convs of one
// indice_key, forward passes of later frames never: a plan belongs to one frame's rulebook).
// One record per tile = the image of the kernel's LDS block: ent [32 * kvol] | cnt [64] |
// gbase [64] | soff [64] | nsteps, padded to 16 bytes.
static int tc_tile_rows(int rows);

__global__ void __launch_bounds__(256)
cv_plan_kernel(const int *__restrict__ nbr, int k_cap, const int *__restrict__ k_dev, int kvol,
               int tr, int *__restrict__ plan, int plan_stride)
{
    extern __shared__ __align__(16) unsigned char smem[];
    int *nl = reinterpret_cast<int *>(smem);                   // [32][kvol]
    unsigned *ent = reinterpret_cast<unsigned *>(nl + ((32 * kvol + 3) & ~3));
    int *cnt = reinterpret_cast<int *>(ent + 32 * kvol);
    int *gbase = cnt + 64;
    int *soff = gbase + 64;
    int *nsteps_p = soff + 64;
    const int k_out = lcf_count(k_cap, k_dev);
    const int row0 = blockIdx.x * tr;
    int4 *dst = reinterpret_cast<int4 *>(plan + (size_t)blockIdx.x * plan_stride);
    if (row0 >= k_out) { // the conv kernels leave such a tile before reading its record
        return;
    }
    for (int e = threadIdx.x; e < 32 * kvol; e += 256) ent[e] = 0u; // unused entries: defined
    if (threadIdx.x < 64) {
        cnt[threadIdx.x] = 0;
        gbase[threadIdx.x] = 0;
        soff[threadIdx.x] = 0;
        if (threadIdx.x < 4) nsteps_p[threadIdx.x] = 0; // the count and the record's padding
    }
    __syncthreads();
    tc_build_steps<4, 256>(nbr, row0, tr, kvol, k_out, nl, ent, cnt, gbase, soff, nsteps_p, 32);
    const int4 *src = reinterpret_cast<const int4 *>(ent);
    for (int e = threadIdx.x; e < plan_stride / 4; e += 256) dst[e] = src[e];
}

extern "C" size_t lcf_conv_tile_plan_ints(int k_cap, int kvol, int tile_rows)
{
    if (k_cap < 0 || kvol <= 0 || kvol > LCF_MAX_KVOL || tile_rows < 1 || tile_rows > 32) return 0;
    const size_t tiles = ((size_t)k_cap + tile_rows - 1) / tile_rows;
    return (tiles ? tiles : 1) * (size_t)tc_plan_stride(kvol);
}

// fewer offsets than this: the tile-compacted kernel (the 3-offset conv in front of dense():
// 0.627 -> 0.631 ms per frame on the quad-row form, tools/ab_lib.sh)
#define Q4_KVOL_MIN 9

// Tile height of the quad-row form: full 32-row tiles, lower only while that fills three
// workgroups per CU (18.6k rows: 25-row tiles; in the frame 0.6298 -> 0.6274 ms, 1024 or 1280
// tiles are no better / worse: tools/ab_env.sh LCF_ENGINE_Q4_FILL at the time)
static int q4_tile_rows(int rows)
{
    const int fixed = lcf_opts().conv_tile;
    if (fixed >= 16 && fixed <= 32) return fixed;
    int tr = (rows + 767) / 768;
    return tr < 16 ? 16 : (tr > 32 ? 32 : tr);
}

static bool q4_default_for(int kvol, int cin, int cout)
{
    const int impl = lcf_opts().conv_impl;
    return (cin == 32 || cin == 64) && (cout == 32 || cout == 64) &&
           (impl == 5 || (impl == 0 && kvol >= Q4_KVOL_MIN));
}

extern "C" int lcf_conv_tile_rows_for(int rows, int kvol, int cin, int cout, int math)
{
    // the height the default forward of that shape and arithmetic (math as in
    // lcf_indice_conv_fwd_planned) picks for a launch expecting `rows`
    LcfCall call(nullptr);
    if (math == 0 && q4_default_for(kvol, cin, cout)) return q4_tile_rows(rows);
    return tc_tile_rows(rows);
}

extern "C" int lcf_conv_tile_rows(int rows)
{
    // the height the tile-compacted forward picks for a launch that expects `rows` rows
    LcfCall call(nullptr); // (the process-wide LCF_OPT_CONV_TILE, if one is set, applies)
    return tc_tile_rows(rows);
}

extern "C" int lcf_conv_tile_plan(const int *nbr, int k, const int *k_dev, int kvol,
                                  int tile_rows, int *plan, void *stream)
{
    if (k < 0 || kvol <= 0 || tile_rows < 1 || tile_rows > 32 || !plan || (k > 0 && !nbr) ||
        ((uintptr_t)plan & 15))
        return LCF_ERR_ARG;
    if (kvol > LCF_MAX_KVOL) return LCF_ERR_UNSUPPORTED;
    if (k >= (1 << 25)) return LCF_ERR_RANGE;
    if (k == 0) return LCF_OK;
    const int nb = (k + tile_rows - 1) / tile_rows;
    const int stride = tc_plan_stride(kvol);
    const size_t lds = tc_plan_lds_bytes(kvol);
    cv_plan_kernel<<<nb, 256, lds, (hipStream_t)stream>>>(nbr, k, k_dev, kvol, tile_rows, plan,
                                                          stride);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

// Launch-time choices come from the options of the call in progress (lcf_opts(), common.h):
// conv_tile (LCF_OPT_CONV_TILE: 0 = balance over the CUs, 16..32 = fixed height), conv_sets
// (LCF_OPT_CONV_SETS: 0 = choose from the row count, 1 or 2 wave sets), rows_hint
// (LCF_OPT_CONV_ROWS_HINT: expected rows, 0 = the capacity passed).

template <int CIN, int COUT, int TR, bool BF16 = false, int NSETS = 1, int NP = 1>
static int cv_launch_tc_n(const float *feat, const float *wimg, const int *nbr, int k_cap,
                          const int *k_dev, int kvol, int cin, int cout, const float *scale,
                          const float *shift, int relu, float *out, hipStream_t st, int tr = TR,
                          TcDense dn = TcDense{nullptr, 0, 0, 0, 0})
{
    using C = TcCfg<CIN, COUT, TR>;
    static_assert(NP == 1 || BF16, "operand planes are a bf16 notion");
    size_t lds = tc_lds_bytes(CIN, COUT, TR, kvol, BF16, NSETS, NP);
    if (lds > 160 * 1024) return LCF_ERR_UNSUPPORTED;
    static size_t lds_set = 0; // per instantiation: largest dynamic LDS size opted into so far
    static std::mutex lds_mu;  // two threads' first launches must not interleave set + record
    std::unique_lock<std::mutex> lds_lock(lds_mu);
    if (lds > 64 * 1024 && lds > lds_set) {
        if (hipFuncSetAttribute((const void *)cv_tc_kernel<CIN, COUT, TR, true, BF16, NSETS, NP>,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return LCF_ERR_LAUNCH;
        if constexpr (!BF16) {
            if (hipFuncSetAttribute((const void *)cv_tc_kernel<CIN, COUT, TR, false, false, NSETS, 1>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return LCF_ERR_LAUNCH;
        }
        lds_set = lds;
    }
    lds_lock.unlock();
    if (tr < 1 || tr > TR) return LCF_ERR_ARG;
    if (dn.coors && cout != COUT) return LCF_ERR_UNSUPPORTED;
    int nb = (k_cap + tr - 1) / tr;
    constexpr int TT = C::THREADS * NSETS;
    // a tile plan (lcf_conv_tile_plan) of the call in progress replaces the per-launch set-up
    const LcfOpts op = lcf_opts();
    const int *plan = op.plan;
    const int pstride = plan ? tc_plan_stride(kvol) : 0;
    if (plan && (op.plan_tr != tr || TR != 32)) return LCF_ERR_ARG;
    if constexpr (BF16) {
        if (cin != CIN) return LCF_ERR_UNSUPPORTED;
        cv_tc_kernel<CIN, COUT, TR, true, true, NSETS, NP><<<nb, TT, lds, st>>>(
            feat, wimg, nbr, k_cap, k_dev, kvol, cin, cout, scale, shift, relu, out, tr, dn, plan,
            pstride);
    } else if (cin == CIN)
        cv_tc_kernel<CIN, COUT, TR, true, false, NSETS, 1><<<nb, TT, lds, st>>>(
            feat, wimg, nbr, k_cap, k_dev, kvol, cin, cout, scale, shift, relu, out, tr, dn, plan,
            pstride);
    else
        cv_tc_kernel<CIN, COUT, TR, false, false, NSETS, 1><<<nb, TT, lds, st>>>(
            feat, wimg, nbr, k_cap, k_dev, kvol, cin, cout, scale, shift, relu, out, tr, dn, plan,
            pstride);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

// Rows per tile: the largest height <= 32 for which the tiles are a whole number of rounds over
// the 256 CUs (581 tiles of 32 rows are 2 on most CUs and 3 on some, and the launch lasts as
// long as the CUs with 3; 744 tiles of 25 rows are 3 everywhere, each 22% shorter).
static int tc_tile_rows(int rows)
{
    const LcfOpts op = lcf_opts();
    if (op.plan) return op.plan_tr; // the plan was built for this height
    const int fixed = op.conv_tile;
    if (fixed >= 16 && fixed <= 32) return fixed;
    const int per_round = 256;
    const int m = (rows + per_round * 32 - 1) / (per_round * 32); // tiles per CU at 32 rows
    if (m < 1) return 32;
    int tr = (rows + per_round * m - 1) / (per_round * m);
    if (tr < 16) tr = 16;
    if (tr > 32) tr = 32;
    return tr;
}

// Two wave sets pay when every tile of the launch is resident at once (the chain of a tile, not
// the number of slots, is then what bounds the launch): workgroups per CU by LDS and by waves
// (6 per SIMD at this register count) x 256 CUs against the number of tiles.
static bool tc_two_sets_fit(int CIN, int COUT, int kvol, int rows, int tr, bool bf16 = false,
                            int planes = 1)
{
    const int nw = (COUT / 16 >= 4 ? 4 : COUT / 16) * 2;
    const size_t lds = tc_lds_bytes(CIN, COUT, 32, kvol, bf16, 2, planes);
    int per_cu = (int)((size_t)160 * 1024 / lds);
    if (24 / nw < per_cu) per_cu = 24 / nw;
    return (long long)per_cu * 256 >= (rows + tr - 1) / tr;
}

#include "conv_q4.h"

template <int CIN, int COUT>
static int cv_launch_tc(const float *feat, const float *wimg, const int *nbr, int k_cap,
                        const int *k_dev, int kvol, int cin, int cout, const float *scale,
                        const float *shift, int relu, float *out, hipStream_t st,
                        TcDense dn = TcDense{nullptr, 0, 0, 0, 0})
{
    // tiles of up to 32 rows: measured against 64-row tiles on every SECOND layer (half the
    // workgroups with the same serial chain lose)
    const LcfOpts op = lcf_opts();
    const int rows = op.rows_hint > 0 && op.rows_hint < k_cap ? op.rows_hint : k_cap;
    const int tr = tc_tile_rows(rows);
    // quad-row form (conv_q4.h): the default for the 3x3x3 convs of 32 / 64 channels
    if (q4_shape_ok<CIN, COUT>() && cin == CIN && cout == COUT && q4_default_for(kvol, cin, cout)) {
        // its own tile heights (q4_tile_rows): no per-CU balance to strike in this form (32 rows:
        // 0.6336 -> 0.6292 ms per frame against the balanced heights, tools/ab_tile.py)
        const int trq = op.plan ? op.plan_tr : q4_tile_rows(rows);
        return cv_launch_q4<CIN, COUT>(feat, wimg, nbr, k_cap, k_dev, kvol, scale, shift, relu,
                                       out, st, trq, dn);
    }
    const bool two = op.conv_sets == 2 ||
                     (op.conv_sets == 0 && tc_two_sets_fit(CIN, COUT, kvol, rows, tr));
    if (two && tc_lds_bytes(CIN, COUT, 32, kvol, false, 2) <= 160 * 1024)
        return cv_launch_tc_n<CIN, COUT, 32, false, 2>(feat, wimg, nbr, k_cap, k_dev, kvol, cin,
                                                       cout, scale, shift, relu, out, st, tr, dn);
    return cv_launch_tc_n<CIN, COUT, 32>(feat, wimg, nbr, k_cap, k_dev, kvol, cin, cout, scale,
                                         shift, relu, out, st, tr, dn);
}

// -------------------------------------------------------------------------- bf16 entry points
__global__ void __launch_bounds__(256)
cv_weight_image_bf16_kernel(const float *__restrict__ w, int kvol, int cin, int cout, int CIN,
                            int COUT, int transpose, __bf16 *__restrict__ img)
{
    // image [o][tg][ks][q][r16][j] <- W'[k = 32ks + 8q + j][c = 16tg + r16], W' = W[o] or W[o]^T
    const int KS = CIN / 32;
    long long total = (long long)kvol * CIN * COUT;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        int o = (int)(e / (CIN * COUT));
        int r = (int)(e - (long long)o * CIN * COUT);
        int j = r & 7, r16 = (r >> 3) & 15, q = (r >> 7) & 3;
        int ks = (r >> 9) % KS, tg = (r >> 9) / KS;
        int k = 32 * ks + 8 * q + j, c = 16 * tg + r16;
        float v = 0.f;
        if (!transpose) {
            if (k < cin && c < cout) v = w[((size_t)o * cin + k) * cout + c];
        } else {
            if (k < cout && c < cin) v = w[((size_t)o * cin + c) * cout + k];
        }
        img[e] = (__bf16)v;
    }
}

static bool cv_bf16_ok(int c) { return c == 32 || c == 64 || c == 128; }

extern "C" int lcf_weight_image_bf16(const float *w, int kvol, int cin, int cout, int transpose,
                                     void *img, void *stream)
{
    if (!w || !img || kvol <= 0 || cin <= 0 || cout <= 0) return LCF_ERR_ARG;
    if (!cv_bf16_ok(cin) || !cv_bf16_ok(cout)) return LCF_ERR_UNSUPPORTED;
    int CIN = transpose ? cout : cin, COUT = transpose ? cin : cout;
    long long total = (long long)kvol * CIN * COUT;
    int nb = (int)((total + 255) / 256);
    if (nb > 2048) nb = 2048;
    cv_weight_image_bf16_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(
        w, kvol, cin, cout, CIN, COUT, transpose, (__bf16 *)img);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

template <int CIN, int COUT>
static int cv_launch_tc_bf16(const float *feat, const float *wimg, const int *nbr, int k_cap,
                             const int *k_dev, int kvol, int cin, int cout, const float *scale,
                             const float *shift, int relu, float *out, hipStream_t st)
{
    // (64-row tiles, which halve the W stream per row, were measured at batch 8: 212 -> 286 us
    // on the 64->64 layers -- fewer, longer chains per CU lose more than the traffic saves)
    const LcfOpts op = lcf_opts();
    return cv_launch_tc_n<CIN, COUT, 32, true>(feat, wimg, nbr, k_cap, k_dev, kvol, cin, cout,
                                               scale, shift, relu, out, st,
                                               op.plan ? op.plan_tr : 32);
}

static int cv_fwd_bf16(const float *feat, const void *wimg_bf16, const int *nbr, int k_out,
                       const int *k_out_dev, int kvol, int cin, int cout, const float *scale,
                       const float *shift, int relu, float *out, void *stream)
{
    if (k_out < 0 || kvol <= 0 || !wimg_bf16 || (k_out > 0 && (!feat || !out)) ||
        (!nbr && kvol != 1))
        return LCF_ERR_ARG;
    if (kvol > LCF_MAX_KVOL || !cv_bf16_ok(cin) || !cv_bf16_ok(cout)) return LCF_ERR_UNSUPPORTED;
    if (((uintptr_t)feat & 15) || ((uintptr_t)wimg_bf16 & 15) || ((uintptr_t)out & 15) ||
        ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15))
        return LCF_ERR_ARG;
    if (k_out >= (1 << 25)) return LCF_ERR_RANGE;
    if (k_out == 0) return LCF_OK;
    hipStream_t st = (hipStream_t)stream;
    const float *wi = (const float *)wimg_bf16;
#define CV_BF16_CASE(A, B)                                                                  \
    if (cin == A && cout == B)                                                              \
        return cv_launch_tc_bf16<A, B>(feat, wi, nbr, k_out, k_out_dev, kvol, cin, cout, scale, \
                                       shift, relu, out, st);
    CV_BF16_CASE(32, 32) CV_BF16_CASE(32, 64) CV_BF16_CASE(32, 128)
    CV_BF16_CASE(64, 32) CV_BF16_CASE(64, 64) CV_BF16_CASE(64, 128)
    CV_BF16_CASE(128, 32) CV_BF16_CASE(128, 64) CV_BF16_CASE(128, 128)
#undef CV_BF16_CASE
    return LCF_ERR_UNSUPPORTED;
}

extern "C" int lcf_indice_conv_fwd_bf16(const float *feat, const void *wimg_bf16, const int *nbr,
                                        int k_out, const int *k_out_dev, int kvol, int cin,
                                        int cout, const float *scale, const float *shift,
                                        int relu, float *out, void *stream, lcf_ctx *ctx)
{
    LcfCall call(ctx);
    return cv_fwd_bf16(feat, wimg_bf16, nbr, k_out, k_out_dev, kvol, cin, cout, scale, shift,
                       relu, out, stream);
}

// -------------------------------------------------------------------------- bf16x3 entry points
__global__ void __launch_bounds__(256)
cv_weight_image_bf16x3_kernel(const float *__restrict__ w, int kvol, int cin, int cout, int CIN,
                              int COUT, int transpose, __bf16 *__restrict__ img, int planes)
{
    // two or three planes of the bf16 image (cv_weight_image_bf16_kernel): hi = bf16(w),
    // mid = bf16(w - hi), lo = bf16(w - hi - mid)
    const int KS = CIN / 32;
    long long total = (long long)kvol * CIN * COUT;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        int o = (int)(e / (CIN * COUT));
        int r = (int)(e - (long long)o * CIN * COUT);
        int j = r & 7, r16 = (r >> 3) & 15, q = (r >> 7) & 3;
        int ks = (r >> 9) % KS, tg = (r >> 9) / KS;
        int k = 32 * ks + 8 * q + j, c = 16 * tg + r16;
        float v = 0.f;
        if (!transpose) {
            if (k < cin && c < cout) v = w[((size_t)o * cin + k) * cout + c];
        } else {
            if (k < cout && c < cin) v = w[((size_t)o * cin + c) * cout + k];
        }
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 l = (__bf16)r1;
        img[e] = h;
        img[total + e] = l;
        if (planes == 3) img[2 * total + e] = (__bf16)(r1 - (float)l);
    }
}

static int cv_weight_image_split(const float *w, int kvol, int cin, int cout, int transpose,
                                 void *img, void *stream, int planes)
{
    if (!w || !img || kvol <= 0 || cin <= 0 || cout <= 0) return LCF_ERR_ARG;
    if (!cv_bf16_ok(cin) || !cv_bf16_ok(cout)) return LCF_ERR_UNSUPPORTED;
    int CIN = transpose ? cout : cin, COUT = transpose ? cin : cout;
    long long total = (long long)kvol * CIN * COUT;
    int nb = (int)((total + 255) / 256);
    if (nb > 2048) nb = 2048;
    cv_weight_image_bf16x3_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(
        w, kvol, cin, cout, CIN, COUT, transpose, (__bf16 *)img, planes);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_weight_image_bf16x3(const float *w, int kvol, int cin, int cout, int transpose,
                                       void *img, void *stream)
{
    return cv_weight_image_split(w, kvol, cin, cout, transpose, img, stream, 2);
}

extern "C" int lcf_weight_image_bf16x8(const float *w, int kvol, int cin, int cout, int transpose,
                                       void *img, void *stream)
{
    return cv_weight_image_split(w, kvol, cin, cout, transpose, img, stream, 3);
}

// same launch-time choices as the fp32 form (cv_launch_tc): balanced tile height, second wave set
template <int CIN, int COUT, int NP>
static int cv_launch_tc_split(const float *feat, const float *wimg, const int *nbr, int k_cap,
                              const int *k_dev, int kvol, int cin, int cout, const float *scale,
                              const float *shift, int relu, float *out, hipStream_t st,
                              TcDense dn)
{
    const LcfOpts op = lcf_opts();
    const int rows = op.rows_hint > 0 && op.rows_hint < k_cap ? op.rows_hint : k_cap;
    const int tr = tc_tile_rows(rows);
    // three planes: a second wave set never paid (tools/bench_conv.py: 42 vs 50 us at 18.6k rows)
    const bool two = op.conv_sets == 2 || (op.conv_sets == 0 && NP < 3 &&
                                           tc_two_sets_fit(CIN, COUT, kvol, rows, tr, true, NP));
    if (two && tc_lds_bytes(CIN, COUT, 32, kvol, true, 2, NP) <= 160 * 1024)
        return cv_launch_tc_n<CIN, COUT, 32, true, 2, NP>(feat, wimg, nbr, k_cap, k_dev, kvol, cin,
                                                          cout, scale, shift, relu, out, st, tr,
                                                          dn);
    return cv_launch_tc_n<CIN, COUT, 32, true, 1, NP>(feat, wimg, nbr, k_cap, k_dev, kvol, cin,
                                                      cout, scale, shift, relu, out, st, tr, dn);
}

static int cv_fwd_split(const float *feat, const void *wimg_bf16x3, const int *nbr, int k_out,
                        const int *k_out_dev, int kvol, int cin, int cout, const float *scale,
                        const float *shift, int relu, float *out, const int *dense_coors,
                        int batch, const int shape[3], void *stream, int planes)
{
    if (k_out < 0 || kvol <= 0 || !wimg_bf16x3 || (k_out > 0 && (!feat || !out)) ||
        (!nbr && kvol != 1) || (dense_coors && (!shape || batch <= 0)))
        return LCF_ERR_ARG;
    if (kvol > LCF_MAX_KVOL || !cv_bf16_ok(cin) || !cv_bf16_ok(cout)) return LCF_ERR_UNSUPPORTED;
    if (((uintptr_t)feat & 15) || ((uintptr_t)wimg_bf16x3 & 15) || ((uintptr_t)out & 15) ||
        ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15) || ((uintptr_t)dense_coors & 15))
        return LCF_ERR_ARG;
    if (k_out >= (1 << 25)) return LCF_ERR_RANGE;
    if (k_out == 0) return LCF_OK;
    hipStream_t st = (hipStream_t)stream;
    const float *wi = (const float *)wimg_bf16x3;
    TcDense dn{nullptr, 0, 0, 0, 0};
    if (dense_coors) dn = TcDense{dense_coors, batch, shape[0], shape[1], shape[2]};
#define CV_X3_CASE(A, B)                                                                      \
    if (cin == A && cout == B)                                                                \
        return planes == 3 ? cv_launch_tc_split<A, B, 3>(feat, wi, nbr, k_out, k_out_dev, kvol, \
                                                         cin, cout, scale, shift, relu, out,  \
                                                         st, dn)                              \
                           : cv_launch_tc_split<A, B, 2>(feat, wi, nbr, k_out, k_out_dev, kvol, \
                                                         cin, cout, scale, shift, relu, out,  \
                                                         st, dn);
    CV_X3_CASE(32, 32) CV_X3_CASE(32, 64) CV_X3_CASE(32, 128)
    CV_X3_CASE(64, 32) CV_X3_CASE(64, 64) CV_X3_CASE(64, 128)
    CV_X3_CASE(128, 32) CV_X3_CASE(128, 64) CV_X3_CASE(128, 128)
#undef CV_X3_CASE
    return LCF_ERR_UNSUPPORTED;
}

extern "C" int lcf_indice_conv_fwd_bf16x3(const float *feat, const void *wimg_bf16x3,
                                          const int *nbr, int k_out, const int *k_out_dev,
                                          int kvol, int cin, int cout, const float *scale,
                                          const float *shift, int relu, float *out,
                                          const int *dense_coors, int batch, const int shape[3],
                                          void *stream, lcf_ctx *ctx)
{
    LcfCall call(ctx);
    return cv_fwd_split(feat, wimg_bf16x3, nbr, k_out, k_out_dev, kvol, cin, cout, scale, shift,
                        relu, out, dense_coors, batch, shape, stream, 2);
}

extern "C" int lcf_indice_conv_fwd_bf16x8(const float *feat, const void *wimg_bf16x8,
                                          const int *nbr, int k_out, const int *k_out_dev,
                                          int kvol, int cin, int cout, const float *scale,
                                          const float *shift, int relu, float *out,
                                          const int *dense_coors, int batch, const int shape[3],
                                          void *stream, lcf_ctx *ctx)
{
    LcfCall call(ctx);
    return cv_fwd_split(feat, wimg_bf16x8, nbr, k_out, k_out_dev, kvol, cin, cout, scale, shift,
                        relu, out, dense_coors, batch, shape, stream, 3);
}

// -------------------------------------------------------------------------- wgrad
// dw[o][ci][co] += sum_j feat[j][ci] * dout[in_nbr[j][o]][co]
// Workgroup = (kernel offset o, chunk of wg_rows input rows); each of its 4 waves walks a quarter
// of the chunk 64 rows at a time: the rows that have a pair at offset o are COMPACTED inside the
// wave (ballot + ds_permute: no LDS round trip), so the MFMAs (K dimension = pairs, 4 per
// instruction) run over existing pairs only -- 2x fewer than rows on the submanifold layers,
// 7x fewer on the strided ones.  Operands of the next 4 pairs are loaded before the MFMAs of the
// current 4.  The four partial dW[o] are summed in LDS and added to HBM once per workgroup
// (fp32 atomics: the sum order across workgroups is not reproducible in the last bits).

// Workgroup -> (row chunk, offset).  Workgroups go round-robin to the 8 XCDs (id % 8), each with
// its own 4 MB L2; the 27 offsets of one chunk read the same feat rows and neighbouring dout
// rows, so they are given consecutive slots on ONE XCD: id = 8 * (group * kvol + o) + xcd,
// chunk = 8 * group + xcd.  Against a (chunk, offset) grid in launch order, batch 8 (294k rows,
// 150 MB of operands, far beyond the L2s): 32->32 256 -> 168 us, strided 32->64 210 -> 100 us,
// the split-operand 64->64 392 -> 300 us.  Grid: 8 * ceil(nchunk / 8) * kvol workgroups.
__device__ __forceinline__ bool wg_chunk_offset(int nchunk, int kvol, int &chunk, int &o)
{
    const int id = blockIdx.x, t = id >> 3;
    const int g = t / kvol;
    o = t - g * kvol;
    chunk = g * 8 + (id & 7);
    return chunk < nchunk;
}

template <int CIN, int COUT>
__global__ void __launch_bounds__(256)
cv_wgrad_kernel(const float *__restrict__ feat, const float *__restrict__ dout,
                const int *__restrict__ in_nbr, int k_in, int kvol, int cin, int cout,
                int wg_rows, float *__restrict__ dw, int ldf, int ldd, int ldw)
{
    // ldf / ldd / ldw: row strides (floats) of feat, dout and dw[o]: cin, cout, cout for the
    // sparse conv; column blocks of wider matrices for lcf_linear_wgrad.  in_nbr == NULL: the
    // identity table (dense rows: dw = feat^T dout).
    constexpr int MT = CIN / 16;  // row tiles of dW (ci)
    constexpr int NT = COUT / 16; // col tiles of dW (co)
    __shared__ float red[CIN * COUT];
    __shared__ int any_s;
    int chunk, o;
    if (!wg_chunk_offset((k_in + wg_rows - 1) / wg_rows, kvol, chunk, o)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15;
    const int q = lane >> 4;
    const int wbeg = chunk * wg_rows + wave * (wg_rows / 4);

    for (int e = tid; e < CIN * COUT; e += 256) red[e] = 0.f;
    if (tid == 0) any_s = 0;
    __syncthreads();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool any_total = false;

    // Operand fetch for pairs 4i..4i+3: lane quarter q takes pair 4i+q and reads MT (NT)
    // CONSECUTIVE channels of its rows -- one 16-byte load per operand at 64 channels, the 16
    // lanes of a quarter covering the whole row.  dW tile (m, n) therefore holds
    // ci = MT*row + m, co = NT*col + n (un-permuted when written out).
    const bool vec_ok = (cin == CIN) && (cout == COUT) && !(ldf & 3) && !(ldd & 3);
    auto fetch = [&](int cj, int cn, int c, int i, float *a, float *b) {
        const int s = 4 * i + q;
        const int j = __shfl(cj, s, 64);
        const int n = __shfl(cn, s, 64);
        const bool ok = s < c;
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = 0.f;
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) b[nn] = 0.f;
        if (ok) {
            const float *fa = feat + (size_t)j * ldf + MT * r16;
            const float *fb = dout + (size_t)n * ldd + NT * r16;
            if (vec_ok) {
                if constexpr (MT % 4 == 0) {
#pragma unroll
                    for (int m = 0; m < MT; m += 4) {
                        float4 x = *reinterpret_cast<const float4 *>(fa + m);
                        a[m] = x.x; a[m + 1] = x.y; a[m + 2] = x.z; a[m + 3] = x.w;
                    }
                } else if constexpr (MT == 2) {
                    float2 x = *reinterpret_cast<const float2 *>(fa);
                    a[0] = x.x; a[1] = x.y;
                } else {
                    a[0] = fa[0];
                }
                if constexpr (NT % 4 == 0) {
#pragma unroll
                    for (int nn = 0; nn < NT; nn += 4) {
                        float4 x = *reinterpret_cast<const float4 *>(fb + nn);
                        b[nn] = x.x; b[nn + 1] = x.y; b[nn + 2] = x.z; b[nn + 3] = x.w;
                    }
                } else if constexpr (NT == 2) {
                    float2 x = *reinterpret_cast<const float2 *>(fb);
                    b[0] = x.x; b[1] = x.y;
                } else {
                    b[0] = fb[0];
                }
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (MT * r16 + m < cin) a[m] = fa[m];
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
                    if (NT * r16 + nn < cout) b[nn] = fb[nn];
            }
        }
    };

    for (int blk = 0; blk < wg_rows / 4 / 64; ++blk) {
        const int row = wbeg + blk * 64 + lane;
        const int nb = row < k_in ? (in_nbr ? in_nbr[(size_t)row * kvol + o] : row) : -1;
        const unsigned long long m = __ballot(nb >= 0);
        if (m == 0ull) continue; // wave-uniform
        any_total = true;
        const int c = __popcll(m);
        const unsigned long long lt = (1ull << lane) - 1ull;
        // compacted slot of this lane: pairs first (in row order), the rest behind them
        const int slot = nb >= 0 ? __popcll(m & lt) : c + __popcll(~m & lt);
        const int cj = __builtin_amdgcn_ds_permute(slot * 4, row);
        const int cn = __builtin_amdgcn_ds_permute(slot * 4, nb);
        const int steps = (c + 3) >> 2;
        // operands of the next 3 steps are in flight while one step's MFMAs issue
        float a[4][MT], b[4][NT];
        fetch(cj, cn, c, 0, a[0], b[0]);
        fetch(cj, cn, c, 1, a[1], b[1]);
        fetch(cj, cn, c, 2, a[2], b[2]);
        for (int i = 0; i < steps; i += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                fetch(cj, cn, c, i + u + 3, a[(u + 3) & 3], b[(u + 3) & 3]); // past the end: masked
                if (i + u < steps) {
#pragma unroll
                    for (int mm = 0; mm < MT; ++mm)
#pragma unroll
                        for (int nn = 0; nn < NT; ++nn)
                            acc[mm][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                a[u][mm], b[u][nn], acc[mm][nn], 0, 0, 0);
                }
            }
        }
    }
    // sum the four waves' partial dW[o] in LDS, one wave after the other (plain RMW)
    if (any_total) any_s = 1;
    for (int w = 0; w < 4; ++w) {
        if (wave == w && any_total) {
#pragma unroll
            for (int mm = 0; mm < MT; ++mm)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[(MT * (q * 4 + r) + mm) * COUT + NT * r16 + nn] += acc[mm][nn][r];
        }
        __syncthreads();
    }
    if (!any_s) return;
    float *dwo = dw + (size_t)o * cin * ldw;
    for (int e = tid; e < CIN * COUT; e += 256) {
        const int ci = e / COUT, co = e % COUT;
        if (ci < cin && co < cout) atomicAdd(&dwo[(size_t)ci * ldw + co], red[e]);
    }
}

// wgrad on the bf16 matrix cores.  Same decomposition as cv_wgrad_kernel (workgroup = offset x
// row chunk, pairs compacted inside the wave), but the K dimension of one
// v_mfma_f32_16x16x32_bf16 is 32 pairs: lane quarter q takes pairs 8q .. 8q+7 of the step and
// issues its 16 row loads (8 of feat, 8 of dout, 16 bytes each at 64 channels) at once; component
// m of the eight loaded vectors is the A fragment of dW tile row m (B alike), so the fragments are
// built in registers with no LDS transpose.  NP = 1: operands rounded to bf16 (the training
// variant of BASELINE.json configs[2]).  NP = 2: operands split hi + lo as in the bf16x3 forward,
// three products per term: fp32-grade gradients (<= 3 * 2^-18 per product) at 3/16 of the fp32
// MFMA cycles, and on the pipe that overlaps the conversions.
template <int CIN, int COUT, int NP>
__global__ void __launch_bounds__(256)
cv_wgrad_mc_kernel(const float *__restrict__ feat, const float *__restrict__ dout,
                   const int *__restrict__ in_nbr, int k_in, int kvol, int wg_rows,
                   float *__restrict__ dw)
{
    constexpr int MT = CIN / 16, NT = COUT / 16;
    static_assert(MT % 2 == 0 && NT % 2 == 0, "32, 64 or 128 channels");
    __shared__ float red[CIN * COUT];
    __shared__ int any_s;
    int chunk, o;
    if (!wg_chunk_offset((k_in + wg_rows - 1) / wg_rows, kvol, chunk, o)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15;
    const int q = lane >> 4;
    const int wbeg = chunk * wg_rows + wave * (wg_rows / 4);

    for (int e = tid; e < CIN * COUT; e += 256) red[e] = 0.f;
    if (tid == 0) any_s = 0;
    __syncthreads();

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bool any_total = false;

    // one row's MT (NT) consecutive channels of this lane
    auto load_row = [&](const float *p, float *v, auto cnt) {
        constexpr int N = decltype(cnt)::value;
        if constexpr (N % 4 == 0) {
#pragma unroll
            for (int m = 0; m < N; m += 4) {
                float4 x = *reinterpret_cast<const float4 *>(p + m);
                v[m] = x.x; v[m + 1] = x.y; v[m + 2] = x.z; v[m + 3] = x.w;
            }
        } else {
            float2 x = *reinterpret_cast<const float2 *>(p);
            v[0] = x.x; v[1] = x.y;
        }
    };

    for (int blk = 0; blk < wg_rows / 4 / 64; ++blk) {
        const int row = wbeg + blk * 64 + lane;
        const int nb = row < k_in ? in_nbr[(size_t)row * kvol + o] : -1;
        const unsigned long long mk = __ballot(nb >= 0);
        if (mk == 0ull) continue; // wave-uniform
        any_total = true;
        const int c = __popcll(mk);
        const unsigned long long lt = (1ull << lane) - 1ull;
        const int slot = nb >= 0 ? __popcll(mk & lt) : c + __popcll(~mk & lt);
        const int cj = __builtin_amdgcn_ds_permute(slot * 4, row);
        const int cn = __builtin_amdgcn_ds_permute(slot * 4, nb);
        for (int s0 = 0; s0 < c; s0 += 32) { // one or two steps of 32 pairs
            float xa[8][MT], xb[8][NT];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int s = s0 + 8 * q + j;
                const int rj = __shfl(cj, s & 63, 64);
                const int rn = __shfl(cn, s & 63, 64);
                if (s < c) {
                    load_row(feat + (size_t)rj * CIN + MT * r16, xa[j],
                             std::integral_constant<int, MT>{});
                    load_row(dout + (size_t)rn * COUT + NT * r16, xb[j],
                             std::integral_constant<int, NT>{});
                } else {
#pragma unroll
                    for (int m = 0; m < MT; ++m) xa[j][m] = 0.f;
#pragma unroll
                    for (int n = 0; n < NT; ++n) xb[j][n] = 0.f;
                }
            }
            bf16x8 bh[NT], bl[NP == 2 ? NT : 1];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const __bf16 h = (__bf16)xb[j][n];
                    bh[n][j] = h;
                    if constexpr (NP == 2) bl[n][j] = (__bf16)(xb[j][n] - (float)h);
                }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                bf16x8 ah, al;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const __bf16 h = (__bf16)xa[j][m];
                    ah[j] = h;
                    if constexpr (NP == 2) al[j] = (__bf16)(xa[j][m] - (float)h);
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if constexpr (NP == 2) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[n], acc[m][n],
                                                                            0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[n], acc[m][n],
                                                                            0, 0, 0);
                    }
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[n], acc[m][n], 0,
                                                                        0, 0);
                }
            }
        }
    }
    if (any_total) any_s = 1;
    for (int w = 0; w < 4; ++w) {
        if (wave == w && any_total) {
#pragma unroll
            for (int mm = 0; mm < MT; ++mm)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        red[(MT * (q * 4 + r) + mm) * COUT + NT * r16 + nn] += acc[mm][nn][r];
        }
        __syncthreads();
    }
    if (!any_s) return;
    float *dwo = dw + (size_t)o * CIN * COUT;
    for (int e = tid; e < CIN * COUT; e += 256) atomicAdd(&dwo[e], red[e]);
}

#include "conv_bt.h" // bf16-row training kernels (forward / dgrad / wgrad), round 3
#include "conv_bx.h" // inference on three exact bf16 planes, six products (round 3)

// -------------------------------------------------------------------------- host side

extern "C" size_t lcf_weight_image_elems(int kvol, int cin, int cout, int transpose)
{
    int ci = transpose ? cout : cin, co = transpose ? cin : cout;
    int CIN = cv_pad_cin(ci), COUT = cv_pad_cout(co);
    if (kvol <= 0 || CIN < 0 || COUT < 0) return 0;
    return (size_t)kvol * CIN * COUT;
}

extern "C" int lcf_weight_image(const float *w, int kvol, int cin, int cout, int transpose,
                                float *img, void *stream)
{
    if (!w || !img || kvol <= 0 || cin <= 0 || cout <= 0) return LCF_ERR_ARG;
    int ci = transpose ? cout : cin, co = transpose ? cin : cout;
    int CIN = cv_pad_cin(ci), COUT = cv_pad_cout(co);
    if (CIN < 0 || COUT < 0) return LCF_ERR_UNSUPPORTED;
    long long total = (long long)kvol * CIN * COUT;
    int nb = (int)((total + 255) / 256);
    if (nb > 2048) nb = 2048;
    cv_weight_image_kernel<<<nb, 256, 0, (hipStream_t)stream>>>(w, kvol, cin, cout, CIN, COUT,
                                                                transpose, img);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

template <int CIN, int COUT, int NSPLIT>
static int cv_launch_fwd_n(const float *feat, const float *wimg, const int *nbr, int k_out,
                           const int *k_dev, int kvol, int cin, int cout, const float *scale,
                           const float *shift, int relu, float *out, hipStream_t st)
{
    constexpr int ROWS = 16 * CV_WAVES / NSPLIT;
    size_t lds = (size_t)2 * CIN * COUT * 4 + (size_t)ROWS * kvol * 4 + (size_t)kvol * 4 + 16;
    int nb = (k_out + ROWS - 1) / ROWS;
    {
        // more than the 64 KiB default: opt in (gfx950 has 160 KiB of LDS per CU); monotonic per
        // instantiation and under a lock, so that two threads' launches cannot lower it
        static size_t lds_set = 0;
        static std::mutex lds_mu;
        std::lock_guard<std::mutex> lock(lds_mu);
        if (lds > 64 * 1024 && lds > lds_set) {
            if (hipFuncSetAttribute((const void *)cv_fwd_kernel<CIN, COUT, NSPLIT>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return LCF_ERR_LAUNCH;
            lds_set = lds;
        }
    }
    cv_fwd_kernel<CIN, COUT, NSPLIT><<<nb, CV_BLOCK, lds, st>>>(
        feat, wimg, nbr, k_out, k_dev, kvol, cin, cout, scale, shift, relu, out);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

// Tuning knob LCF_OPT_CONV_SPLIT (1|2|4; 0 = pick by size) of the call in progress.
static int cv_split_override() { return lcf_opts().conv_split; }

template <int CIN, int COUT>
static int cv_launch_fwd(const float *feat, const float *wimg, const int *nbr, int k_out,
                         const int *k_dev, int kvol, int cin, int cout, const float *scale,
                         const float *shift, int relu, float *out, hipStream_t st)
{
    // Smaller workgroups while the launch would otherwise leave CUs unevenly loaded: aim for
    // >= ~4 workgroups per CU (256 CUs).
    int split = 1;
    constexpr int MAXSPLIT = COUT / 16 >= 4 ? 4 : COUT / 16;
    while (split < MAXSPLIT && (long long)k_out * split < 64ll * 256 * 4) split *= 2;
    int ov = cv_split_override();
    if (ov == 1 || ov == 2 || ov == 4) split = ov <= MAXSPLIT ? ov : MAXSPLIT;
    if constexpr (MAXSPLIT >= 4) {
        if (split == 4)
            return cv_launch_fwd_n<CIN, COUT, 4>(feat, wimg, nbr, k_out, k_dev, kvol, cin, cout, scale,
                                                 shift, relu, out, st);
    }
    if constexpr (MAXSPLIT >= 2) {
        if (split == 2)
            return cv_launch_fwd_n<CIN, COUT, 2>(feat, wimg, nbr, k_out, k_dev, kvol, cin, cout, scale,
                                                 shift, relu, out, st);
    }
    return cv_launch_fwd_n<CIN, COUT, 1>(feat, wimg, nbr, k_out, k_dev, kvol, cin, cout, scale, shift,
                                         relu, out, st);
}

#define CV_DISPATCH(FN, ...)                                                          \
    switch (CIN * 1000 + COUT) {                                                      \
    case 4016: return FN<4, 16>(__VA_ARGS__);                                         \
    case 4032: return FN<4, 32>(__VA_ARGS__);                                         \
    case 4064: return FN<4, 64>(__VA_ARGS__);                                         \
    case 4128: return FN<4, 128>(__VA_ARGS__);                                        \
    case 8016: return FN<8, 16>(__VA_ARGS__);                                         \
    case 8032: return FN<8, 32>(__VA_ARGS__);                                         \
    case 8064: return FN<8, 64>(__VA_ARGS__);                                         \
    case 8128: return FN<8, 128>(__VA_ARGS__);                                        \
    case 16016: return FN<16, 16>(__VA_ARGS__);                                       \
    case 16032: return FN<16, 32>(__VA_ARGS__);                                       \
    case 16064: return FN<16, 64>(__VA_ARGS__);                                       \
    case 16128: return FN<16, 128>(__VA_ARGS__);                                      \
    case 32016: return FN<32, 16>(__VA_ARGS__);                                       \
    case 32032: return FN<32, 32>(__VA_ARGS__);                                       \
    case 32064: return FN<32, 64>(__VA_ARGS__);                                       \
    case 32128: return FN<32, 128>(__VA_ARGS__);                                      \
    case 64016: return FN<64, 16>(__VA_ARGS__);                                       \
    case 64032: return FN<64, 32>(__VA_ARGS__);                                       \
    case 64064: return FN<64, 64>(__VA_ARGS__);                                       \
    case 64128: return FN<64, 128>(__VA_ARGS__);                                      \
    case 128016: return FN<128, 16>(__VA_ARGS__);                                     \
    case 128032: return FN<128, 32>(__VA_ARGS__);                                     \
    case 128064: return FN<128, 64>(__VA_ARGS__);                                     \
    case 128128: return FN<128, 128>(__VA_ARGS__);                                    \
    default: return LCF_ERR_UNSUPPORTED;                                              \
    }

extern "C" int lcf_indice_conv_fwd(const float *feat, const float *wimg, const int *nbr,
                                   int k_out, const int *k_out_dev, int kvol, int cin, int cout,
                                   const float *scale, const float *shift, int relu, float *out,
                                   void *stream, lcf_ctx *ctx)
{
    LcfCall call(ctx);
    const int g_conv_impl = call.eff.conv_impl;
    if (k_out < 0 || kvol <= 0 || cin <= 0 || cout <= 0 || !wimg ||
        (k_out > 0 && (!feat || !out)) || (!nbr && kvol != 1))
        return LCF_ERR_ARG;
    if (kvol > LCF_MAX_KVOL) return LCF_ERR_UNSUPPORTED;
    if (((uintptr_t)feat & 15) || ((uintptr_t)wimg & 15) || ((uintptr_t)out & 15) ||
        ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15))
        return LCF_ERR_ARG;
    // compacted entries pack (input row << 7 | tile row) in 32 bits: rows < 2^25 on both sides
    // (the rulebook builders refuse to build tables over more input rows than that)
    if (k_out >= (1 << 25)) return LCF_ERR_RANGE;
    int CIN = cv_pad_cin(cin), COUT = cv_pad_cout(cout);
    if (CIN < 0 || COUT < 0) return LCF_ERR_UNSUPPORTED;
    if (k_out == 0) return LCF_OK;
    hipStream_t st = (hipStream_t)stream;
    // other narrow shapes (< 32 inputs with >= 64 outputs) stay on the slab form
    const bool slab_ok = nbr && CIN * COUT <= 8192;
    const bool use_slab = g_conv_impl == 1 ? slab_ok : (g_conv_impl == 2 ? false
                                                                        : slab_ok && (COUT < 32 || CIN < 32));
    // thin layers, compacted 64-row tiles: only on request (LCF_OPT_CONV_IMPL = 4).  Alone they
    // beat the split-slab form on two of the three SECOND layers (128 -> 16 19.9 vs 22.6 us,
    // 16 -> 32 15.4 vs 18.8, 16 -> 16 9.5 vs 7.1), inside the frame -- beside the rulebook branch
    // -- they lose (0.6977 vs 0.6936 ms per frame, A/B in one call, tools/ab_impl.py)
    if (nbr && COUT <= 32 && g_conv_impl == 4) {
        CV_DISPATCH(cv_launch_tn, feat, wimg, nbr, k_out, k_out_dev, kvol, cin, cout, scale,
                    shift, relu, out, st);
    }
    // thin layers: one 16-row slab per workgroup, waves split the reduction
    if (nbr && COUT <= 32 && (g_conv_impl == 3 || (g_conv_impl == 0 && (COUT < 32 || CIN < 32)))) {
        CV_DISPATCH(cv_launch_ss, feat, wimg, nbr, k_out, k_out_dev, kvol, cin, cout, scale,
                    shift, relu, out, st);
    }
    if (use_slab) {
        CV_DISPATCH(cv_launch_fwd, feat, wimg, nbr, k_out, k_out_dev, kvol, cin, cout, scale,
                    shift, relu, out, st);
    }
    CV_DISPATCH(cv_launch_tc, feat, wimg, nbr, k_out, k_out_dev, kvol, cin, cout, scale, shift,
                relu, out, st);
}

extern "C" int lcf_indice_conv_fwd_dense(const float *feat, const float *wimg, const int *nbr,
                                         int k_out, const int *k_out_dev, int kvol, int cin,
                                         int cout, const float *scale, const float *shift,
                                         int relu, const int *out_coors, int batch,
                                         const int shape[3], float *dense, void *stream,
                                         lcf_ctx *ctx)
{
    LcfCall call(ctx);
    if (k_out < 0 || kvol <= 0 || cin <= 0 || cout <= 0 || !wimg || !nbr || !shape ||
        batch <= 0 || (k_out > 0 && (!feat || !dense || !out_coors)))
        return LCF_ERR_ARG;
    if (kvol > LCF_MAX_KVOL) return LCF_ERR_UNSUPPORTED;
    if (((uintptr_t)feat & 15) || ((uintptr_t)wimg & 15) || ((uintptr_t)out_coors & 15) ||
        ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15))
        return LCF_ERR_ARG;
    if (k_out >= (1 << 25)) return LCF_ERR_RANGE;
    int CIN = cv_pad_cin(cin), COUT = cv_pad_cout(cout);
    if (CIN < 0 || COUT < 0 || COUT != cout) return LCF_ERR_UNSUPPORTED;
    if (k_out == 0) return LCF_OK;
    const TcDense dn{out_coors, batch, shape[0], shape[1], shape[2]};
    CV_DISPATCH(cv_launch_tc, feat, wimg, nbr, k_out, k_out_dev, kvol, cin, cout, scale, shift,
                relu, dense, (hipStream_t)stream, dn);
}

// Forward over a tile plan (lcf_conv_tile_plan) instead of a neighbour table: the same kernels and
// the same arithmetic, bit for bit, without the per-launch compaction of each tile.  math 0: fp32
// MFMA (lcf_indice_conv_fwd / _dense), 1: bf16 operands, 2: bf16x3, 3: bf16x8 (wimg is the image
// of that form).  Shapes the tile-compacted kernels serve: cin, cout in {32, 64, 128}.
extern "C" int lcf_indice_conv_fwd_planned(const float *feat, const void *wimg, int math,
                                           const int *plan, int tile_rows, int k_out,
                                           const int *k_out_dev, int kvol, int cin, int cout,
                                           const float *scale, const float *shift, int relu,
                                           float *out, const int *dense_coors, int batch,
                                           const int shape[3], void *stream, lcf_ctx *ctx)
{
    LcfCall call(ctx);
    if (!plan || ((uintptr_t)plan & 15) || tile_rows < 1 || tile_rows > 32 || math < 0 || math > 3)
        return LCF_ERR_ARG;
    if (!cv_bf16_ok(cin) || !cv_bf16_ok(cout)) return LCF_ERR_UNSUPPORTED;
    call.eff.plan = plan;
    call.eff.plan_tr = tile_rows;
    // (tile-compacted kernels only: 0 / 5 let cv_launch_tc pick the quad-row form)
    if (call.eff.conv_impl != 5 && call.eff.conv_impl != 0) call.eff.conv_impl = 2;
    const int *nbr = plan;  // never read: the kernels take the plan's entries
    if (math == 0) {
        if (k_out < 0 || kvol <= 0 || !wimg || (k_out > 0 && (!feat || !out)) ||
            (dense_coors && (!shape || batch <= 0)))
            return LCF_ERR_ARG;
        if (kvol > LCF_MAX_KVOL) return LCF_ERR_UNSUPPORTED;
        if (((uintptr_t)feat & 15) || ((uintptr_t)wimg & 15) || ((uintptr_t)out & 15) ||
            ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15) || ((uintptr_t)dense_coors & 15))
            return LCF_ERR_ARG;
        if (k_out >= (1 << 25)) return LCF_ERR_RANGE;
        if (k_out == 0) return LCF_OK;
        const int CIN = cin, COUT = cout;
        TcDense dn{nullptr, 0, 0, 0, 0};
        if (dense_coors) dn = TcDense{dense_coors, batch, shape[0], shape[1], shape[2]};
        CV_DISPATCH(cv_launch_tc, feat, (const float *)wimg, nbr, k_out, k_out_dev, kvol, cin,
                    cout, scale, shift, relu, out, (hipStream_t)stream, dn);
    }
    if (math == 1) {
        if (dense_coors) return LCF_ERR_UNSUPPORTED;
        return cv_fwd_bf16(feat, wimg, nbr, k_out, k_out_dev, kvol, cin, cout, scale, shift, relu,
                           out, stream);
    }
    return cv_fwd_split(feat, wimg, nbr, k_out, k_out_dev, kvol, cin, cout, scale, shift, relu,
                        out, dense_coors, batch, shape, stream, math == 3 ? 3 : 2);
}

template <int CIN, int COUT>
static int cv_launch_wgrad(const float *feat, const float *dout, const int *in_nbr, int k_in,
                           int kvol, int cin, int cout, float *dw, hipStream_t st, int ldf = 0,
                           int ldd = 0, int ldw = 0)
{
    if (!ldf) ldf = cin;
    if (!ldd) ldd = cout;
    if (!ldw) ldw = cout;
    // rows per workgroup (multiple of 256): ~3000 workgroups per launch, so that the per-
    // workgroup reduction and its atomics stay small next to the MFMA work
    long long want = ((long long)k_in * kvol + 2999) / 3000;
    int wg_rows = (int)((want + 255) / 256 * 256);
    if (wg_rows < 256) wg_rows = 256;
    if (wg_rows > 16384) wg_rows = 16384;
    const int nchunk = (k_in + wg_rows - 1) / wg_rows;
    const dim3 grid(((nchunk + 7) / 8) * 8 * kvol, 1);
    cv_wgrad_kernel<(CIN < 16 ? 16 : CIN), COUT><<<grid, 256, 0, st>>>(
        feat, dout, in_nbr, k_in, kvol, cin, cout, wg_rows, dw, ldf, ldd, ldw);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

template <int CIN, int COUT>
static int cv_launch_wgrad_mc(const float *feat, const float *dout, const int *in_nbr, int k_in,
                              int kvol, int planes, float *dw, hipStream_t st)
{
    long long want = ((long long)k_in * kvol + 2999) / 3000;
    int wg_rows = (int)((want + 255) / 256 * 256);
    if (wg_rows < 256) wg_rows = 256;
    if (wg_rows > 16384) wg_rows = 16384;
    const int nchunk = (k_in + wg_rows - 1) / wg_rows;
    const dim3 grid(((nchunk + 7) / 8) * 8 * kvol, 1);
    if (planes == 2)
        cv_wgrad_mc_kernel<CIN, COUT, 2><<<grid, 256, 0, st>>>(feat, dout, in_nbr, k_in, kvol,
                                                               wg_rows, dw);
    else
        cv_wgrad_mc_kernel<CIN, COUT, 1><<<grid, 256, 0, st>>>(feat, dout, in_nbr, k_in, kvol,
                                                               wg_rows, dw);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_indice_conv_wgrad_bf16(const float *feat, const float *dout,
                                          const int *in_nbr, int k_in, int kvol, int cin,
                                          int cout, int planes, float *dw, void *stream)
{
    if (k_in < 0 || kvol <= 0 || cin <= 0 || cout <= 0 || !dw || (planes != 1 && planes != 2) ||
        (k_in > 0 && (!feat || !dout || !in_nbr)))
        return LCF_ERR_ARG;
    if (kvol > LCF_MAX_KVOL || !cv_bf16_ok(cin) || !cv_bf16_ok(cout) || cin * cout > 4096)
        return LCF_ERR_UNSUPPORTED;
    if (((uintptr_t)feat & 15) || ((uintptr_t)dout & 15)) return LCF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (lcf_fill32(dw, 0u, (size_t)kvol * cin * cout * 4, st) != LCF_OK) return LCF_ERR_LAUNCH;
    if (k_in == 0) return LCF_OK;
#define CV_WMC_CASE(A, B)                                                                     \
    if (cin == A && cout == B)                                                                \
        return cv_launch_wgrad_mc<A, B>(feat, dout, in_nbr, k_in, kvol, planes, dw, st);
    CV_WMC_CASE(32, 32) CV_WMC_CASE(32, 64) CV_WMC_CASE(64, 32) CV_WMC_CASE(64, 64)
    CV_WMC_CASE(32, 128) CV_WMC_CASE(128, 32)
#undef CV_WMC_CASE
    return LCF_ERR_UNSUPPORTED;
}

// cols columns of dout / dW[o] starting at the given pointers, rows ld floats apart
template <int CIN, int COUT>
static int cv_launch_wgrad_b16(const void *feat, const void *dout, const int *in_nbr, int k_in,
                               int kvol, float *dw, hipStream_t st)
{
    if (lcf_fill32(dw, 0u, (size_t)kvol * CIN * COUT * 4, st) != LCF_OK) return LCF_ERR_LAUNCH;
    if (k_in == 0) return LCF_OK;
    long long want = ((long long)k_in * kvol + 2999) / 3000; // ~3000 workgroups, as the mc form
    int wg_rows = (int)((want + 255) / 256 * 256);
    if (wg_rows < 256) wg_rows = 256;
    if (wg_rows > 16384) wg_rows = 16384;
    const int nchunk = (k_in + wg_rows - 1) / wg_rows;
    const unsigned grid = (unsigned)(8 * ((nchunk + 7) / 8) * kvol);
    cv_wgrad_b16_kernel<CIN, COUT><<<grid, 256, 0, st>>>(
        (const __bf16 *)feat, (const __bf16 *)dout, in_nbr, k_in, kvol, wg_rows, dw);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" int lcf_indice_conv_wgrad_b16(const void *feat_bf16, const void *dout_bf16,
                                         const int *in_nbr, int k_in, int kvol, int cin, int cout,
                                         float *dw, void *stream)
{
    if (k_in < 0 || kvol <= 0 || kvol > LCF_MAX_KVOL || !dw || !in_nbr ||
        (k_in > 0 && (!feat_bf16 || !dout_bf16)))
        return LCF_ERR_ARG;
    if (((uintptr_t)feat_bf16 & 15) || ((uintptr_t)dout_bf16 & 15)) return LCF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
#define WB_CASE(A, B) \
    if (cin == A && cout == B) \
        return cv_launch_wgrad_b16<A, B>(feat_bf16, dout_bf16, in_nbr, k_in, kvol, dw, st)
    WB_CASE(32, 32);
    WB_CASE(32, 64);
    WB_CASE(64, 32);
    WB_CASE(64, 64);
#undef WB_CASE
    return LCF_ERR_UNSUPPORTED;
}

static int cv_wgrad_cols(const float *feat, const float *dout, const int *in_nbr, int k_in, int kvol,
                         int cin, int cols, int ld, float *dw, hipStream_t st)
{
    int CIN = cv_pad_cin(cin), COUT = cv_pad_cout(cols);
    if (CIN < 0 || COUT < 0 || CIN * COUT > 8192) return LCF_ERR_UNSUPPORTED;
    const int cout = cols;
    CV_DISPATCH(cv_launch_wgrad, feat, dout, in_nbr, k_in, kvol, cin, cout, dw, st, cin, ld, ld);
}

extern "C" int lcf_indice_conv_wgrad(const float *feat, const float *dout, const int *in_nbr,
                                     int k_in, int kvol, int cin, int cout, float *dw,
                                     void *stream)
{
    if (k_in < 0 || kvol <= 0 || cin <= 0 || cout <= 0 || !dw ||
        (k_in > 0 && (!feat || !dout || !in_nbr)))
        return LCF_ERR_ARG;
    if (kvol > LCF_MAX_KVOL) return LCF_ERR_UNSUPPORTED;
    int CIN = cv_pad_cin(cin), COUT = cv_pad_cout(cout);
    if (CIN < 0 || COUT < 0) return LCF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (lcf_fill32(dw, 0u, (size_t)kvol * cin * cout * 4, st) != LCF_OK) return LCF_ERR_LAUNCH;
    if (k_in == 0) return LCF_OK;
    if (CIN * COUT <= 8192) {
        CV_DISPATCH(cv_launch_wgrad, feat, dout, in_nbr, k_in, kvol, cin, cout, dw, st);
    }
    // wider than one wave's accumulators (128 x 128): two column halves of dout and dW
    if (cout & 1) return LCF_ERR_UNSUPPORTED;
    const int h = cout / 2;
    int rc = cv_wgrad_cols(feat, dout, in_nbr, k_in, kvol, cin, h, cout, dw, st);
    if (rc != LCF_OK) return rc;
    return cv_wgrad_cols(feat, dout + h, in_nbr, k_in, kvol, cin, h, cout, dw + h, st);
}

static int cv_linear_wgrad_block(const float *x, int ldx, const float *dy, int ldy, int n, int cin,
                                 int cout, float *dw, int ldw, hipStream_t st)
{
    int CIN = cv_pad_cin(cin), COUT = cv_pad_cout(cout);
    if (CIN < 0 || COUT < 0 || CIN * COUT > 8192) return LCF_ERR_UNSUPPORTED;
    CV_DISPATCH(cv_launch_wgrad, x, dy, nullptr, n, 1, cin, cout, dw, st, ldx, ldy, ldw);
}

// dw [cin, cout] = x^T dy over n dense rows: the weight gradient of a row-wise Linear layer (the
// VFE's nn.Linear, voxel_encoder.py:64, 127; with x = dy and dy = x it is nn.Linear's own
// [out, in] layout).  The reduction runs over ~10^5 rows into a matrix of at most 128 x 128, a
// shape library GEMMs serve with a handful of workgroups; here it is the sparse wgrad kernel on
// the identity table: thousands of row chunks, fp32 MFMA, one atomic add per chunk and element.
extern "C" int lcf_linear_wgrad(const float *x, const float *dy, int n, int cin, int cout,
                                float *dw, void *stream)
{
    if (n < 0 || cin <= 0 || cout <= 0 || !dw || (n > 0 && (!x || !dy))) return LCF_ERR_ARG;
    if (cin > 128 || cout > 128) return LCF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (lcf_fill32(dw, 0u, (size_t)cin * cout * 4, st) != LCF_OK) return LCF_ERR_LAUNCH;
    if (n == 0) return LCF_OK;
    int CIN = cv_pad_cin(cin), COUT = cv_pad_cout(cout);
    if (CIN < 0 || COUT < 0) return LCF_ERR_UNSUPPORTED;
    if (CIN * COUT <= 8192) return cv_linear_wgrad_block(x, cin, dy, cout, n, cin, cout, dw, cout, st);
    // 128 x 128: two column halves of dy (the accumulators of one wave cover 8192 elements)
    const int h = cout / 2;
    if (cout & 1) return LCF_ERR_UNSUPPORTED;
    int rc = cv_linear_wgrad_block(x, cin, dy, cout, n, cin, h, dw, cout, st);
    if (rc != LCF_OK) return rc;
    return cv_linear_wgrad_block(x, cin, dy + h, cout, n, cin, h, dw + h, cout, st);
}
