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

typedef float f32x4 __attribute__((ext_vector_type(4)));

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


// -------------------------------------------------------------------------- tile-compacted fwd
// Second-generation forward ("tc"): the workgroup owns a tile of 64 output rows.  For every
// kernel offset it COMPACTS the rows of the tile that have a neighbour at that offset (wave
// ballot + popcount) and runs the MFMAs only over the resulting ceil(c/16) dense groups of 16
// gathered rows, instead of over every 16-row slab that has at least one neighbour.  The waves
// of the workgroup split the output columns, share the gathered A group through LDS and sum
// each group's 16xCOUT result into an fp32 output tile kept in LDS (every element is owned by
// exactly one wave and touched in ascending offset order, so the result is deterministic and
// equals sum_o(dot_o) as the reference's per-offset GEMM + scatter-add computes it).  One
// barrier per group; W[o] and the next group's rows are prefetched to registers behind the
// MFMAs of the current group.  MFMA work ~ pairs*Cin*Cout rounded up to 16-row groups per
// (tile, offset), which is 2-4x less than the slab form on the SECOND rulebooks.
//
// Row counts may live on the device (k_dev): the grid is sized by the capacity and workgroups
// past the actual count exit at once, so the launch needs no host synchronisation.
#define TC_ROWS 64
#define TC_DEPTH 4 /* gathered groups in flight per thread */

template <int CIN, int COUT>
struct TcCfg {
    static constexpr int NW = (COUT / 16 >= 4) ? 4 : COUT / 16; // waves (column split)
    static constexpr int NT = COUT / 16 / NW;                   // column tiles per wave
    static constexpr int THREADS = NW * 64;
    static constexpr int KL = CIN / 4;
    static constexpr int VW = KL < 4 ? KL : 4;
    static constexpr int SG = KL / VW;
    static constexpr int A4 = KL >= 4 ? KL / 4 : 1;
    static constexpr int CH = CIN / 4;                // float4 chunks per gathered row
    static constexpr int ACH = 16 * CH;               // chunks per 16-row group
    static constexpr int APT = (ACH + THREADS - 1) / THREADS;
    static constexpr int ASTR = CIN + 4;              // padded A row stride (floats)
    static constexpr int WELEMS = CIN * COUT;
    static constexpr int WVEC = WELEMS / 4;
    static constexpr int WPT = (WVEC + THREADS - 1) / THREADS;
    static constexpr int OSTR = COUT + 4;             // padded output-tile row stride
};

static size_t tc_lds_bytes(int CIN, int COUT, int kvol, int wbufs)
{
    size_t b = (size_t)wbufs * CIN * COUT * 4;       // W[o] images
    b += (size_t)2 * 16 * (CIN + 4) * 4;             // A groups (double buffered)
    b += (size_t)(TC_ROWS + 1) * (COUT + 4) * 4;     // output tile + dump row
    b += (size_t)TC_ROWS * kvol * 4 * 2;             // neighbour tile + compacted lists
    b += (size_t)64 * 4 + (size_t)kvol * 4 * 4 + 16; // counts, items, n_items
    return b;
}

#define TC_THREADS(COUT) (((COUT) / 16 >= 4 ? 4 : (COUT) / 16) * 64)

template <int CIN, int COUT, bool EXACT>
__global__ void __launch_bounds__(TC_THREADS(COUT))
cv_tc_kernel(const float *__restrict__ feat, const float *__restrict__ wimg,
             const int *__restrict__ nbr, int k_cap, const int *__restrict__ k_dev, int kvol,
             int cin, int cout, const float *__restrict__ scale,
             const float *__restrict__ shift, int relu, float *__restrict__ out, int wbufs)
{
    using C = TcCfg<CIN, COUT>;
    constexpr int THREADS = C::THREADS, NT = C::NT, KL = C::KL, VW = C::VW, SG = C::SG;
    constexpr int CH = C::CH, ACH = C::ACH, APT = C::APT, ASTR = C::ASTR, OSTR = C::OSTR;
    constexpr int WELEMS = C::WELEMS, WVEC = C::WVEC, WPT = C::WPT;
    constexpr bool WFULL = (WVEC % THREADS) == 0;
    constexpr bool AFULL = (ACH % THREADS) == 0;

    int k_out = k_cap;
    if (k_dev) {
        int kd = *k_dev;
        k_out = kd < k_cap ? kd : k_cap;
    }
    const int row0 = blockIdx.x * TC_ROWS;
    if (row0 >= k_out) return;

    extern __shared__ __align__(16) unsigned char smem[];
    float *wl = reinterpret_cast<float *>(smem);                 // [wbufs][WELEMS]
    float *al = wl + (size_t)wbufs * WELEMS;                     // [2][16][ASTR]
    float *ot = al + 2 * 16 * ASTR;                              // [64 + dump row][OSTR]
    int *nl = reinterpret_cast<int *>(ot + (TC_ROWS + 1) * OSTR); // [64][kvol]
    unsigned *pl = reinterpret_cast<unsigned *>(nl + TC_ROWS * kvol); // [kvol][64]
    int *cnt = reinterpret_cast<int *>(pl + TC_ROWS * kvol);     // [64]
    int *items = cnt + 64;                                       // [kvol*4]
    int *nitems_p = items + kvol * 4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6; // = column slice
    const int r16 = lane & 15;
    const int q = lane >> 4;

    // ---- neighbour tile -> LDS (coalesced; rows past k_out read as -1), output tile = 0
    {
        const int total = TC_ROWS * kvol;
        const long long gbase = (long long)row0 * kvol;
        const long long glimit = (long long)k_out * kvol;
        for (int e = tid; e < total; e += THREADS) {
            int v = -1;
            if (gbase + e < glimit) v = nbr ? nbr[gbase + e] : row0 + e; // NULL table: identity
            nl[e] = v;
        }
        for (int e = tid; e < (TC_ROWS + 1) * OSTR; e += THREADS) ot[e] = 0.f;
    }
    __syncthreads();
    // ---- per offset: compact the rows that have a neighbour (entry = input row << 6 | row)
    for (int o = wave; o < kvol; o += C::NW) {
        const int id = nl[lane * kvol + o];
        const bool has = id >= 0;
        const unsigned long long m = __ballot(has);
        const int p = __popcll(m & ((1ull << lane) - 1ull));
        if (has) pl[o * 64 + p] = ((unsigned)id << 6) | (unsigned)lane;
        if (lane == 0) cnt[o] = __popcll(m);
    }
    __syncthreads();
    // ---- work items (offset, 16-row group), offsets ascending
    if (wave == 0) {
        const int ng = lane < kvol ? (cnt[lane] + 15) >> 4 : 0;
        int incl = ng;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        const int ex = incl - ng;
        for (int g = 0; g < ng; ++g) items[ex + g] = lane * 4 + g;
        if (lane == 63) *nitems_p = incl;
    }
    __syncthreads();
    const int n_items = *nitems_p;

    constexpr bool exact_in = EXACT; // cin == CIN: rows are whole 16-byte chunks
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    struct AReg { float4 v[APT]; };
    struct WReg { float4 v[WPT]; };

    auto agather = [&](int o, int g) -> AReg {
        AReg a;
        const int c = cnt[o];
#pragma unroll
        for (int x = 0; x < APT; ++x) {
            const int ch = tid + x * THREADS;
            a.v[x] = zero4;
            if (AFULL || ch < ACH) {
                const int r = ch / CH, cc = ch % CH;
                int pos = g * 16 + r;
                pos = pos < c ? pos : c - 1;
                const unsigned e = pl[o * 64 + pos];
                const float *src = feat + (size_t)(e >> 6) * cin + cc * 4;
                if constexpr (exact_in) {
                    a.v[x] = *reinterpret_cast<const float4 *>(src);
                } else {
                    const int k0 = cc * 4;
                    float t0 = k0 + 0 < cin ? src[0] : 0.f;
                    float t1 = k0 + 1 < cin ? src[1] : 0.f;
                    float t2 = k0 + 2 < cin ? src[2] : 0.f;
                    float t3 = k0 + 3 < cin ? src[3] : 0.f;
                    a.v[x] = make_float4(t0, t1, t2, t3);
                }
            }
        }
        return a;
    };
    auto astore = [&](int buf, const AReg &a) {
        float *dst = al + buf * 16 * ASTR;
#pragma unroll
        for (int x = 0; x < APT; ++x) {
            const int ch = tid + x * THREADS;
            if (AFULL || ch < ACH) {
                const int r = ch / CH, cc = ch % CH;
                *reinterpret_cast<float4 *>(dst + r * ASTR + cc * 4) = a.v[x];
            }
        }
    };
    auto wload = [&](int o) -> WReg {
        WReg w;
        const float4 *src = reinterpret_cast<const float4 *>(wimg + (size_t)o * WELEMS);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * THREADS;
            w.v[i] = src[(WFULL || e < WVEC) ? e : 0];
        }
        return w;
    };
    auto wstore = [&](int buf, const WReg &w) {
        float4 *dst = reinterpret_cast<float4 *>(wl + (size_t)buf * WELEMS);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int e = tid + i * THREADS;
            if (WFULL || e < WVEC) dst[e] = w.v[i];
        }
    };
    auto compute = [&](int abuf, int wb, int o, int g) {
        const float *ap = al + abuf * 16 * ASTR + r16 * ASTR + q * KL;
        // every LDS read of the item is issued before the first MFMA
        float a[KL];
        if constexpr (KL >= 4) {
#pragma unroll
            for (int i = 0; i < KL / 4; ++i) {
                float4 x = reinterpret_cast<const float4 *>(ap)[i];
                a[4 * i] = x.x;
                a[4 * i + 1] = x.y;
                a[4 * i + 2] = x.z;
                a[4 * i + 3] = x.w;
            }
        } else if constexpr (KL == 2) {
            float2 x = *reinterpret_cast<const float2 *>(ap);
            a[0] = x.x;
            a[1] = x.y;
        } else {
            a[0] = ap[0];
        }
        const float *wb_p = wl + (size_t)wb * WELEMS;
        float b[NT][KL];
#pragma unroll
        for (int sg = 0; sg < SG; ++sg) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int tg = wave * NT + t;
                const float *p = wb_p + (((tg * SG + sg) * 4 + q) * 16 + r16) * VW;
                if constexpr (VW == 4) {
                    float4 x = *reinterpret_cast<const float4 *>(p);
                    b[t][sg * 4 + 0] = x.x;
                    b[t][sg * 4 + 1] = x.y;
                    b[t][sg * 4 + 2] = x.z;
                    b[t][sg * 4 + 3] = x.w;
                } else if constexpr (VW == 2) {
                    float2 x = *reinterpret_cast<const float2 *>(p);
                    b[t][0] = x.x;
                    b[t][1] = x.y;
                } else {
                    b[t][0] = p[0];
                }
            }
        }
        // output rows of this lane's 4 accumulator rows (compacted positions g*16+q*4 .. +3);
        // positions past the offset's count go to the dump row
        const int c = cnt[o];
        const uint4 e4 = *reinterpret_cast<const uint4 *>(pl + o * 64 + g * 16 + q * 4);
        const int p0 = g * 16 + q * 4;
        const int orow[4] = {p0 + 0 < c ? (int)(e4.x & 63u) : TC_ROWS,
                             p0 + 1 < c ? (int)(e4.y & 63u) : TC_ROWS,
                             p0 + 2 < c ? (int)(e4.z & 63u) : TC_ROWS,
                             p0 + 3 < c ? (int)(e4.w & 63u) : TC_ROWS};
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
                acc[t][s & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t][s], acc[t][s & 1],
                                                                     0, 0, 0);
        // sum the group's 16 x (NT*16) block into the output tile: C/D map col = lane&15,
        // row = q*4 + reg.  Every element has ONE owner wave and is touched once per item, in
        // item order, so the no-return LDS adds are deterministic.
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
                ot[orow[r] * OSTR + wave * NT * 16 + r16 + t * 16] =
                    cur[r][t] + (acc[t][0][r] + acc[t][1][r]);
    };

    if (n_items > 0) {
        // Software pipeline: the rows of item i+1 .. i+TC_DEPTH are in flight to registers
        // (ring, statically indexed) while item i computes; W of the next distinct offset is in
        // flight during the whole current offset.
        constexpr int D = TC_DEPTH;
        AReg ring[D];
        WReg wreg;
        auto item_o = [&](int i) { return items[i] >> 2; };
        auto item_g = [&](int i) { return items[i] & 3; };
        // index of the first item of the next distinct offset after item i (or n_items)
        auto next_change = [&](int i) {
            const int oc = item_o(i);
            int j = i + 1;
            while (j < n_items && item_o(j) == oc) ++j;
            return j;
        };
        int o = item_o(0), g = item_g(0);
        {
            AReg a0 = agather(o, g);
            WReg w0 = wload(o);
#pragma unroll
            for (int s = 1; s <= D; ++s) // item j waits in ring[j % D]
                if (s < n_items) ring[s % D] = agather(item_o(s), item_g(s));
            astore(0, a0);
            wstore(0, w0);
        }
        int wnext = next_change(0); // first item that needs the W now being prefetched
        if (wnext < n_items) wreg = wload(item_o(wnext));
        __syncthreads();
        int wb = 0;
        for (int i0 = 0; i0 < n_items; i0 += D) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                const int i = i0 + s;
                if (i < n_items) {          // workgroup-uniform
                    compute(i & 1, wb, o, g);
                    const int in = i + 1;
                    if (in < n_items) {
                        const int on = item_o(in), gn = item_g(in);
                        astore(in & 1, ring[(s + 1) % D]);
                        if (in + D < n_items)
                            ring[(s + 1) % D] = agather(item_o(in + D), item_g(in + D));
                        if (in == wnext) {
                            wstore(wb ^ 1, wreg);
                            wnext = next_change(in);
                            if (wnext < n_items) wreg = wload(item_o(wnext));
                        }
                        __syncthreads();
                        if (on != o) wb ^= 1;
                        o = on;
                        g = gn;
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: (x*scale + shift), relu, coalesced store of the tile
    if (cout == COUT) {
        constexpr int C4 = COUT / 4;
        for (int e = tid; e < TC_ROWS * C4; e += THREADS) {
            const int r = e / C4, c4 = e % C4;
            const int row = row0 + r;
            if (row < k_out) {
                float4 v = *reinterpret_cast<const float4 *>(ot + r * OSTR + c4 * 4);
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
                *reinterpret_cast<float4 *>(out + (size_t)row * cout + c4 * 4) = v;
            }
        }
    } else {
        for (int e = tid; e < TC_ROWS * cout; e += THREADS) {
            const int r = e / cout, c = e % cout;
            const int row = row0 + r;
            if (row < k_out) {
                float v = ot[r * OSTR + c];
                if (scale) v *= scale[c];
                if (shift) v += shift[c];
                if (relu) v = fmaxf(v, 0.f);
                out[(size_t)row * cout + c] = v;
            }
        }
    }
}

template <int CIN, int COUT>
static int cv_launch_tc(const float *feat, const float *wimg, const int *nbr, int k_cap,
                        const int *k_dev, int kvol, int cin, int cout, const float *scale,
                        const float *shift, int relu, float *out, hipStream_t st)
{
    using C = TcCfg<CIN, COUT>;
    int wbufs = kvol == 1 ? 1 : 2;
    size_t lds = tc_lds_bytes(CIN, COUT, kvol, wbufs);
    if (lds > 160 * 1024) return LCF_ERR_UNSUPPORTED;
    static size_t lds_set = 0; // per instantiation: largest dynamic LDS size opted into so far
    if (lds > 64 * 1024 && lds > lds_set) {
        if (hipFuncSetAttribute((const void *)cv_tc_kernel<CIN, COUT, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess ||
            hipFuncSetAttribute((const void *)cv_tc_kernel<CIN, COUT, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return LCF_ERR_LAUNCH;
        lds_set = lds;
    }
    int nb = (k_cap + TC_ROWS - 1) / TC_ROWS;
    if (cin == CIN)
        cv_tc_kernel<CIN, COUT, true><<<nb, C::THREADS, lds, st>>>(
            feat, wimg, nbr, k_cap, k_dev, kvol, cin, cout, scale, shift, relu, out, wbufs);
    else
        cv_tc_kernel<CIN, COUT, false><<<nb, C::THREADS, lds, st>>>(
            feat, wimg, nbr, k_cap, k_dev, kvol, cin, cout, scale, shift, relu, out, wbufs);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

// -------------------------------------------------------------------------- wgrad
// dw[o][ci][co] += sum_j feat[j][ci] * dout[in_nbr[j][o]][co]
// One wave owns (offset o, chunk of WG_ROWS rows) and all (ci,co) tiles; K dimension = rows.
#define WG_ROWS 512

template <int CIN, int COUT>
__global__ void __launch_bounds__(64)
cv_wgrad_kernel(const float *__restrict__ feat, const float *__restrict__ dout,
                const int *__restrict__ in_nbr, int k_in, int kvol, int cin, int cout,
                float *__restrict__ dw)
{
    constexpr int MT = CIN / 16 > 0 ? CIN / 16 : 1; // row tiles of dW (ci)
    constexpr int NT = COUT / 16;                   // col tiles of dW (co)
    const int o = blockIdx.y;
    const int lane = threadIdx.x;
    const int i16 = lane & 15;
    const int kq = lane >> 4;
    const int rbeg = blockIdx.x * WG_ROWS;
    const int rend = min(rbeg + WG_ROWS, k_in);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    bool any_total = false;
    for (int r = rbeg; r < rend; r += 4) {
        const int row = r + kq;
        int nb = -1;
        if (row < rend) nb = in_nbr[(size_t)row * kvol + o];
        if (__ballot(nb >= 0) == 0ull) continue; // none of these 4 rows has a pair at offset o
        any_total = true;
        float a[MT], b[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int ci = m * 16 + i16;
            a[m] = (nb >= 0 && ci < cin) ? feat[(size_t)row * cin + ci] : 0.f;
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            int co = n * 16 + i16;
            b[n] = (nb >= 0 && co < cout) ? dout[(size_t)nb * cout + co] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    if (!any_total) return;
    float *dwo = dw + (size_t)o * cin * cout;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int ci = m * 16 + kq * 4 + r;
                int co = n * 16 + i16;
                if (ci < cin && co < cout) atomicAdd(&dwo[(size_t)ci * cout + co], acc[m][n][r]);
            }
}

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
    if (lds > 64 * 1024) {
        // more than the 64 KiB default: opt in (gfx950 has 160 KiB of LDS per CU)
        if (hipFuncSetAttribute((const void *)cv_fwd_kernel<CIN, COUT, NSPLIT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return LCF_ERR_LAUNCH;
    }
    cv_fwd_kernel<CIN, COUT, NSPLIT><<<nb, CV_BLOCK, lds, st>>>(
        feat, wimg, nbr, k_out, k_dev, kvol, cin, cout, scale, shift, relu, out);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

// Tuning knob (lcf_set_option(LCF_OPT_CONV_SPLIT, 1|2|4)); 0 = pick by size.
static int g_conv_split = 0;
static int g_conv_impl = 1; // 1 = slab form (default), 0 = tile-compacted everywhere
static int cv_split_override() { return g_conv_split; }

extern "C" int lcf_set_option(int option, int value)
{
    if (option == LCF_OPT_CONV_SPLIT) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return LCF_ERR_ARG;
        g_conv_split = value;
        return LCF_OK;
    }
    if (option == LCF_OPT_CONV_IMPL) {
        if (value != 0 && value != 1) return LCF_ERR_ARG;
        g_conv_impl = value;
        return LCF_OK;
    }
    return LCF_ERR_ARG;
}

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
                                   void *stream)
{
    if (k_out < 0 || kvol <= 0 || cin <= 0 || cout <= 0 || !wimg ||
        (k_out > 0 && (!feat || !out)) || (!nbr && kvol != 1))
        return LCF_ERR_ARG;
    if (kvol > LCF_MAX_KVOL) return LCF_ERR_UNSUPPORTED;
    if (((uintptr_t)feat & 15) || ((uintptr_t)wimg & 15) || ((uintptr_t)out & 15) ||
        ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15))
        return LCF_ERR_ARG;
    if (k_out >= (1 << 26)) return LCF_ERR_RANGE; // compacted entries pack row << 6 | lane
    int CIN = cv_pad_cin(cin), COUT = cv_pad_cout(cout);
    if (CIN < 0 || COUT < 0) return LCF_ERR_UNSUPPORTED;
    if (k_out == 0) return LCF_OK;
    hipStream_t st = (hipStream_t)stream;
    if (g_conv_impl == 1 && nbr && CIN * COUT <= 8192) {
        CV_DISPATCH(cv_launch_fwd, feat, wimg, nbr, k_out, k_out_dev, kvol, cin, cout, scale,
                    shift, relu, out, st);
    }
    CV_DISPATCH(cv_launch_tc, feat, wimg, nbr, k_out, k_out_dev, kvol, cin, cout, scale, shift,
                relu, out, st);
}

template <int CIN, int COUT>
static int cv_launch_wgrad(const float *feat, const float *dout, const int *in_nbr, int k_in,
                           int kvol, int cin, int cout, float *dw, hipStream_t st)
{
    dim3 grid((k_in + WG_ROWS - 1) / WG_ROWS, kvol);
    cv_wgrad_kernel<(CIN < 16 ? 16 : CIN), COUT><<<grid, 64, 0, st>>>(feat, dout, in_nbr, k_in,
                                                                       kvol, cin, cout, dw);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
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
    if (CIN < 0 || COUT < 0 || CIN * COUT > 8192) return LCF_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (lcf_fill32(dw, 0u, (size_t)kvol * cin * cout * 4, st) != LCF_OK) return LCF_ERR_LAUNCH;
    if (k_in == 0) return LCF_OK;
    CV_DISPATCH(cv_launch_wgrad, feat, dout, in_nbr, k_in, kvol, cin, cout, dw, st);
}
