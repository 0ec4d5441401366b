// conv_bt.h -- indice_conv for the TRAINING step on rows stored as bf16 (round 3; included by
// conv.hip).  BASELINE.json configs[2]: batch-8 training, bf16 MFMA indice_conv; reference call
// sites second/pytorch/models/middle.py:145-192 (SparseConvolution.forward -> indice_conv and its
// backward), training loop second/pytorch/train.py:316-343.
//
// At batch 8 a level holds 150k-300k rows (40-80 MB of fp32 features): the tile-compacted kernel
// (cv_tc_kernel<.., BF16>) spends its time in the gather -> LDS -> barrier chain of 27 steps per
// 32-row tile, moving 256-byte fp32 rows that it rounds to bf16 on the way in.  Here the
// activations a conv gathers are STORED as bf16 by their producer (lcf_bn_relu_fwd_ex /
// _bwd_ex), so a gathered row is 128 bytes = one cache line, and the kernel is rebuilt around
// that:
//   * a wave owns G groups of 16 output rows for the whole launch; their accumulators (16 x COUT
//     fp32 per group) stay in registers across all kernel offsets -- no output tile in LDS, no
//     read-add-write per group;
//   * for each offset a lane loads its row's A fragment STRAIGHT from global memory into the
//     registers v_mfma_f32_16x16x32_bf16 reads (lane (r16, q) holds 8 consecutive channels of row
//     r16: one 16-byte load per 32 input channels), one offset ahead of the MFMAs.  No LDS
//     staging of rows, no compaction: a row without a neighbour at the offset contributes zeros
//     (masked load), a group with none skips its MFMAs.  bf16 MFMAs are cheap enough that the
//     uncompacted work (27 offsets x all groups) is ~25 us of matrix pipe at 300k rows;
//   * W[o] (8 KB at 64 -> 64) is what the four waves share: double-buffered in LDS, copied in by
//     the whole workgroup one offset ahead, ONE barrier per offset; fragments are read with
//     conflict-free ds_read_b128 (the image is in fragment order);
//   * the workgroup's slice of the neighbour table ([64 G rows][kvol], contiguous in memory) is
//     copied to LDS once, coalesced.
// dgrad is the same kernel over in_nbr with the transposed weight image; OUT_BF16 writes the
// result rounded to bf16 (the gradient goes to a BatchNorm backward that reads it once and to
// nobody who needs fp32).  Deterministic: every output element is one lane's register, summed in
// ascending offset order.
#pragma once

template <int CIN, int COUT, int G, bool OUT_BF16>
__global__ void __launch_bounds__(256)
cv_bt_kernel(const __bf16 *__restrict__ feat, const __bf16 *__restrict__ wimg,
             const int *__restrict__ nbr, int k_out, int kvol, void *__restrict__ out,
             int tiles_per_xcd, unsigned feat_bytes)
{
    constexpr int KS = CIN / 32;           // k-steps of 32 input channels
    constexpr int NT = COUT / 16;          // 16-column tiles
    constexpr int WELEMS = CIN * COUT;     // bf16 elements of one W[o]
    constexpr int WV = WELEMS / 8;         // 16-byte pieces of one W[o]
    constexpr int WPT = (WV + 255) / 256;  // pieces per thread
    constexpr int ROWS = 64 * G;           // rows of the workgroup
    extern __shared__ __align__(16) unsigned char smem[];
    uint4 *wl = reinterpret_cast<uint4 *>(smem);                       // [2][WV]
    int *nl = reinterpret_cast<int *>(smem + (size_t)2 * WELEMS * 2);  // [ROWS][kvol]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15;
    const int q = lane >> 4;
    // Workgroups go round-robin to the 8 XCDs (id % 8), each with its own 4 MB L2.  XCD x takes
    // the x-th contiguous eighth of the tiles, so that one L2 serves one neighbourhood of rows (at
    // batch 8: about one frame) instead of every L2 seeing rows from everywhere; tiles_per_xcd = 0
    // keeps tile = id.
    const int tile = tiles_per_xcd ? (int)(blockIdx.x & 7) * tiles_per_xcd + (int)(blockIdx.x >> 3)
                                   : (int)blockIdx.x;
    const int row0 = tile * ROWS;
    if (row0 >= k_out) return;

    // the workgroup's rows of the neighbour table: one contiguous run of ints
    {
        const int live = (k_out - row0 < ROWS ? k_out - row0 : ROWS) * kvol;
        const int *src = nbr + (size_t)row0 * kvol;
        for (int e = tid; e < ROWS * kvol; e += 256) nl[e] = e < live ? src[e] : -1;
    }
    const uint4 *wg = reinterpret_cast<const uint4 *>(wimg);
#pragma unroll
    for (int x = 0; x < WPT; ++x) {
        const int e = tid + x * 256;
        if (WV % 256 == 0 || e < WV) wl[e] = wg[e];
    }
    __syncthreads();

    f32x4 acc[G][NT];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    struct AReg {
        uint4 v[G][KS];
        int idx[G];
    };
    // Rows of offset o for this lane's G rows: buffer loads through a descriptor of the feature
    // matrix.  A row without a neighbour takes an offset beyond the descriptor's range: the range
    // check returns zeros and the lane costs the texture addresser nothing -- with plain loads of a
    // dummy row for those lanes (53 % of them on a submanifold rulebook) the kernel sat at the
    // per-CU gather rate (~30 B/clk/CU counting the dummies; tools/bench_bt.py, LCF_BT_LDS_PAD).
    // Every load is issued unconditionally (no branch around it): a load behind a branch would
    // make every later s_waitcnt assume the worst, and the point of this loop is that the next
    // offset's rows stay in flight across the barrier.
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16 *>(feat), 0, feat_bytes, 0x00020000);
    typedef unsigned bt_u4 __attribute__((ext_vector_type(4)));
    auto aload = [&](int o, AReg &a) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int idx = nl[((wave * G + g) * 16 + r16) * kvol + o];
            a.idx[g] = idx;
#if defined(BT_ABLATE) && BT_ABLATE == 2 // timing ablation: every gather hits one cached row
            const int voff = (idx & 63) * (CIN * 2) + q * 16;
#else
            const int voff = idx < 0 ? (int)0x80000000 : idx * (CIN * 2) + q * 16;
#endif
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bt_u4 w = __builtin_amdgcn_raw_buffer_load_b128(frs, voff + ks * 64, 0, 0);
                a.v[g][ks] = make_uint4(w.x, w.y, w.z, w.w);
            }
        }
    };
    // W[o] travels global -> registers -> LDS one step AHEAD of the rows: step o stores W[o+1]
    // (requested during step o-1, so its L2 round trip had a whole step to land) and requests
    // W[o+2] into the same registers.  (Named registers, not an array: the array form was left in
    // scratch memory, and a scratch load drains vmcnt like any other.)
    static_assert(WPT <= 4, "W[o] of at most 128 x 64");
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0;
    auto wload = [&](int o) {
#if defined(BT_ABLATE) && BT_ABLATE == 3 // timing ablation: workgroups read different W[o] at a time
        const uint4 *wsrc = wg + (size_t)(((o < kvol ? o : kvol - 1) + blockIdx.x) % kvol) * WV + tid;
#elif defined(BT_ABLATE) && BT_ABLATE == 4 // timing ablation: no W traffic after the first two
        const uint4 *wsrc = wg + (size_t)(o & 1) * WV + tid;
#else
        const uint4 *wsrc = wg + (size_t)(o < kvol ? o : kvol - 1) * WV + tid;
#endif
        if (WV >= 256 || tid < WV) w0 = wsrc[0];
        if constexpr (WPT > 1) w1 = wsrc[256];
        if constexpr (WPT > 2) w2 = wsrc[512];
        if constexpr (WPT > 3) w3 = wsrc[768];
    };
    auto wstore = [&](int o) {
        uint4 *wn = wl + (o & 1) * WV + tid;
        if (WV >= 256 || tid < WV) wn[0] = w0;
        if constexpr (WPT > 1) wn[256] = w1;
        if constexpr (WPT > 2) wn[512] = w2;
        if constexpr (WPT > 3) wn[768] = w3;
    };
    auto step = [&](int o, const AReg &cur, AReg &next) {
        wstore(o + 1); // W[o+1] -> the buffer W[o-1] was read from (everyone is past that barrier)
        wload(o + 2);
        aload(o + 1 < kvol ? o + 1 : o, next); // (the last step reloads its own: harmless)
        // this offset's W fragments: [tg][ks][q][r16] pieces of 8 bf16
        const uint4 *wb = wl + (o & 1) * WV;
        uint4 wf[NT][KS];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wf[t][ks] = wb[((t * KS + ks) * 4 + q) * 16 + r16];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bool has = cur.idx[g] >= 0;
            if (__ballot(has) == 0ull) continue; // no row of the group has a pair here
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 av = __builtin_bit_cast(bf16x8, cur.v[g][ks]);
#if defined(BT_ABLATE) && BT_ABLATE == 1 // timing ablation: one MFMA per k-step instead of NT
                acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    av, __builtin_bit_cast(bf16x8, wf[ks % NT][ks]), acc[g][0], 0, 0, 0);
#pragma unroll
                for (int t = 1; t < NT; ++t) acc[g][t][0] += __uint_as_float(wf[t][ks].x);
#else
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        av, __builtin_bit_cast(bf16x8, wf[t][ks]), acc[g][t], 0, 0, 0);
#endif
            }
        }
        // LDS accesses done, then the barrier -- NOT __syncthreads(), whose fence also drains
        // vmcnt(0), i.e. would wait for the rows and the W[o+2] just requested
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    AReg a0, a1;
    wload(1);
    aload(0, a0);
    int o = 0;
    for (; o + 1 < kvol; o += 2) { // row registers alternate between a0 and a1
        step(o, a0, a1);
        step(o + 1, a1, a0);
    }
    if (o < kvol) step(o, a0, a1);

    // ---- epilogue: lane (col r16, q) holds rows 4q .. 4q+3 of each 16 x 16 tile
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int rbase = row0 + (wave * G + g) * 16 + q * 4;
        if constexpr (!OUT_BF16) {
            float *o32 = reinterpret_cast<float *>(out);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rbase + i;
                if (row < k_out) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        o32[(size_t)row * COUT + t * 16 + r16] = acc[g][t][i];
                }
            }
        } else {
            // two adjacent columns per 4-byte store: even lanes take rows 4q, 4q+2, odd lanes
            // rows 4q+1, 4q+3 of the column pair (r16 & ~1, +1)
            unsigned *o16 = reinterpret_cast<unsigned *>(out);
            const bool odd = r16 & 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const float mine = odd ? acc[g][t][i + 1] : acc[g][t][i];
                    const float give = odd ? acc[g][t][i] : acc[g][t][i + 1];
                    const float got = __shfl_xor(give, 1, 64);
                    const int row = rbase + i + (odd ? 1 : 0);
                    const float lo = odd ? got : mine, hi = odd ? mine : got;
                    if (row < k_out)
                        o16[((size_t)row * COUT + t * 16 + (r16 & ~1)) >> 1] =
                            lcf_bf16_rne(lo) | (lcf_bf16_rne(hi) << 16);
                }
            }
        }
    }
}

static size_t bt_lds_bytes(int cin, int cout, int g, int kvol)
{
    return (size_t)2 * cin * cout * 2 + (size_t)64 * g * kvol * 4;
}

template <int CIN, int COUT, int G>
static int cv_launch_bt_g(const void *feat, const void *wimg, const int *nbr, int k_out, int kvol,
                          void *out, int out_bf16, unsigned feat_bytes, hipStream_t st)
{
    // LCF_BT_LDS_PAD (environment, diagnosis): extra LDS bytes per workgroup, to cap how many share a CU
    static const size_t pad = getenv("LCF_BT_LDS_PAD") ? (size_t)atoi(getenv("LCF_BT_LDS_PAD")) : 0;
    const size_t lds = bt_lds_bytes(CIN, COUT, G, kvol) + pad;
    const int tiles = (k_out + 64 * G - 1) / (64 * G);
    static const bool xcd_off = getenv("LCF_BT_XCD") && atoi(getenv("LCF_BT_XCD")) == 0;
    const int per = xcd_off ? 0 : (tiles + 7) / 8;
    const unsigned grid = (unsigned)(per ? 8 * per : tiles);
    if (out_bf16)
        cv_bt_kernel<CIN, COUT, G, true><<<grid, 256, lds, st>>>(
            (const __bf16 *)feat, (const __bf16 *)wimg, nbr, k_out, kvol, out, per, feat_bytes);
    else
        cv_bt_kernel<CIN, COUT, G, false><<<grid, 256, lds, st>>>(
            (const __bf16 *)feat, (const __bf16 *)wimg, nbr, k_out, kvol, out, per, feat_bytes);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

template <int CIN, int COUT>
static int cv_launch_bt(const void *feat, const void *wimg, const int *nbr, int k_out, int kvol,
                        void *out, int out_bf16, unsigned feat_bytes, hipStream_t st)
{
    // groups per wave: LCF_BT_GROUPS (environment, A/B), else 3 at 64 -> 64 channels and 2
    // elsewhere (tools/bench_bt.py at batch 8, us for G = 1 / 2 / 3 / 4: 64 -> 64 at 275k rows
    // 132 / 119 / 112 / 126, at 136k rows 71 / 68 / 64 / -, 32 -> 32 at 348k rows - / 54 / 57 / 60:
    // more rows per wave amortise the W[o] fragment reads, fewer waves per SIMD hide less)
    static const int genv = getenv("LCF_BT_GROUPS") ? atoi(getenv("LCF_BT_GROUPS")) : 0;
    const int g = genv ? genv : (CIN == 64 && COUT == 64 ? 3 : 2);
    if (g == 1) return cv_launch_bt_g<CIN, COUT, 1>(feat, wimg, nbr, k_out, kvol, out, out_bf16, feat_bytes, st);
    if (g == 3) return cv_launch_bt_g<CIN, COUT, 3>(feat, wimg, nbr, k_out, kvol, out, out_bf16, feat_bytes, st);
    if (g == 4) return cv_launch_bt_g<CIN, COUT, 4>(feat, wimg, nbr, k_out, kvol, out, out_bf16, feat_bytes, st);
    return cv_launch_bt_g<CIN, COUT, 2>(feat, wimg, nbr, k_out, kvol, out, out_bf16, feat_bytes, st);
}

extern "C" int lcf_indice_conv_fwd_b16(const void *feat_bf16, int k_in, const void *wimg_bf16,
                                       const int *nbr, int k_out, int kvol, int cin, int cout,
                                       void *out, int out_bf16, void *stream)
{
    // (k_in: rows of feat -- the gathers go through a range-checked buffer descriptor of
    // k_in * cin * 2 bytes, which must stay below 2 GB)
    if (k_in < 0 || (unsigned long long)k_in * (unsigned)cin * 2ull >= 0x80000000ull)
        return LCF_ERR_RANGE;
    const unsigned feat_bytes = (unsigned)k_in * (unsigned)cin * 2u;
    if (k_out < 0 || kvol <= 0 || kvol > LCF_MAX_KVOL || !wimg_bf16 || !nbr || !out ||
        (k_out > 0 && !feat_bf16))
        return LCF_ERR_ARG;
    if (((uintptr_t)feat_bf16 & 15) || ((uintptr_t)wimg_bf16 & 15) || ((uintptr_t)out & 15))
        return LCF_ERR_ARG;
    if (k_out == 0) return LCF_OK;
    hipStream_t st = (hipStream_t)stream;
#define BT_CASE(A, B) \
    if (cin == A && cout == B) \
        return cv_launch_bt<A, B>(feat_bf16, wimg_bf16, nbr, k_out, kvol, out, out_bf16, feat_bytes, st)
    BT_CASE(32, 32);
    BT_CASE(32, 64);
    BT_CASE(64, 32);
    BT_CASE(64, 64);
    BT_CASE(64, 128);
    BT_CASE(128, 64);
#undef BT_CASE
    return LCF_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------- wgrad on bf16 rows
// cv_wgrad_mc_kernel's decomposition (workgroup = offset x row chunk, pairs compacted inside the
// wave, K dimension of one MFMA = 32 pairs, fragments built in registers from each pair's rows,
// the permuted dW tile) with both operands READ as bf16: a lane's piece of a row is MT (NT)
// consecutive bf16 -- 8 bytes at 64 channels, the 16 lanes of a quarter covering the 128-byte
// row -- and component m of the eight loaded pieces is the fragment of dW tile row m.  No
// conversions and half the operand traffic of the fp32-row form, which is what bounded it.
template <int N>
struct BtPiece {
    unsigned w[N / 2]; // N bf16
};

template <int CIN, int COUT>
__global__ void __launch_bounds__(256)
cv_wgrad_b16_kernel(const __bf16 *__restrict__ feat, const __bf16 *__restrict__ dout,
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

    auto load_piece = [&](const __bf16 *p, auto &dst) {
        constexpr int W = sizeof(dst.w) / 4;
        if constexpr (W == 4) {
            const uint4 x = *reinterpret_cast<const uint4 *>(p);
            dst.w[0] = x.x; dst.w[1] = x.y; dst.w[2] = x.z; dst.w[3] = x.w;
        } else if constexpr (W == 2) {
            const uint2 x = *reinterpret_cast<const uint2 *>(p);
            dst.w[0] = x.x; dst.w[1] = x.y;
        } else {
            dst.w[0] = *reinterpret_cast<const unsigned *>(p);
        }
    };
    // bf16 number m of a piece, in the low half of a dword
    auto half = [](const unsigned *w, int m) -> unsigned {
        return (m & 1) ? (w[m >> 1] >> 16) : (w[m >> 1] & 0xFFFFu);
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
            BtPiece<MT> xa[8];
            BtPiece<NT> xb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int s = s0 + 8 * q + j;
                const int rj = __shfl(cj, s & 63, 64);
                const int rn = __shfl(cn, s & 63, 64);
#pragma unroll
                for (int e = 0; e < MT / 2; ++e) xa[j].w[e] = 0u;
#pragma unroll
                for (int e = 0; e < NT / 2; ++e) xb[j].w[e] = 0u;
                if (s < c) {
                    load_piece(feat + (size_t)rj * CIN + MT * r16, xa[j]);
                    load_piece(dout + (size_t)rn * COUT + NT * r16, xb[j]);
                }
            }
            bf16x8 bh[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                uint4 u;
                u.x = half(xb[0].w, n) | (half(xb[1].w, n) << 16);
                u.y = half(xb[2].w, n) | (half(xb[3].w, n) << 16);
                u.z = half(xb[4].w, n) | (half(xb[5].w, n) << 16);
                u.w = half(xb[6].w, n) | (half(xb[7].w, n) << 16);
                bh[n] = __builtin_bit_cast(bf16x8, u);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                uint4 u;
                u.x = half(xa[0].w, m) | (half(xa[1].w, m) << 16);
                u.y = half(xa[2].w, m) | (half(xa[3].w, m) << 16);
                u.z = half(xa[4].w, m) | (half(xa[5].w, m) << 16);
                u.w = half(xa[6].w, m) | (half(xa[7].w, m) << 16);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, u);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[n], acc[m][n], 0, 0,
                                                                        0);
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
