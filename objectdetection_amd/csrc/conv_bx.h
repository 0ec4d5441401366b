// conv_bx.h -- indice_conv forward for INFERENCE on rows stored as three bf16 planes (round 3;
// included by conv.hip).  BASELINE.json configs[1]: one frame through SpMiddleFHD, reference
// call sites second/pytorch/models/middle.py:145-192 (SparseConvolution.forward -> indice_conv).
//
// An fp32 value is EXACTLY the sum of three bf16 numbers: h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m) (8 + 8 + 8 significand bits).  With both operands split that way,
//     x * w = hH + (hM + mH) + (hL + lH + mM) + O(2^-24 |x w|)
// and the six products kept here are exact in the fp32 accumulator of v_mfma_f32_16x16x32_bf16;
// the three dropped ones are below the rounding error of a single fp32 product.  The result is
// fp32-grade (tests/test_gpu_conv.py measures it against float64 beside the fp32 MFMA kernel) at
// 6/16 of the fp32 MFMA pipe time per useful flop -- which pays for dropping the compaction of the
// fp32 kernels altogether and taking the structure of the training kernel (conv_bt.h):
//   * a wave owns 16 output rows for the whole launch, accumulators in registers over all offsets
//     (no output tile in LDS, no per-group read-add-write, no tile plan, no set-up);
//   * per offset a lane loads its row's A fragments straight from global memory into MFMA
//     registers -- one range-checked 16-byte buffer load per plane and 32 input channels (a row
//     without a neighbour: out-of-range offset, zeros, no memory access) -- one offset ahead;
//   * W[o] (three planes, 24 KB at 64 -> 64) is double-buffered in LDS, requested two offsets
//     ahead, one barrier per offset (s_waitcnt lgkmcnt(0) + s_barrier: the loads stay in flight);
//   * the workgroup's slice of the neighbour table is copied to LDS once, coalesced; rows are
//     requested TWO offsets ahead (three register stages: a round trip to L2 is longer than an
//     offset's MFMAs at one frame's row counts).
// The PRODUCER writes the planes: this kernel's epilogue (scale, shift, ReLU, split, 2-byte
// pairs packed with one shuffle) or lcf_split_planes for rows that come from another kernel.
// Deterministic: every output element is one lane's register, products and offsets in a fixed
// order.
#pragma once

template <int CIN, int COUT, int G, bool OUT_PLANES>
__global__ void __launch_bounds__(256)
cv_bx_kernel(const __bf16 *__restrict__ feat, unsigned plane_bytes,
             const __bf16 *__restrict__ wimg, unsigned wplane_elems, const int *__restrict__ nbr,
             int k_cap, const int *__restrict__ k_dev, int kvol, const float *__restrict__ scale,
             const float *__restrict__ shift, int relu, void *__restrict__ out,
             unsigned out_plane_elems, int tiles_per_xcd)
{
    constexpr int KS = CIN / 32;           // k-steps of 32 input channels
    constexpr int NT = COUT / 16;          // 16-column tiles
    constexpr int WELEMS = CIN * COUT;     // bf16 elements of one plane of W[o]
    constexpr int WV = WELEMS / 8;         // 16-byte pieces of one plane
    constexpr int WV3 = 3 * WV;            // ... of the three planes
    constexpr int WPT = (WV3 + 255) / 256; // pieces per thread
    constexpr int ROWS = 64 * G;           // a wave owns G groups of 16 rows
    static_assert(WPT <= 6, "W[o] of at most 64 x 64 x 3 planes");
    extern __shared__ __align__(16) unsigned char smem[];
    uint4 *wl = reinterpret_cast<uint4 *>(smem);                          // [2][3][WV]
    int *nl = reinterpret_cast<int *>(smem + (size_t)2 * 3 * WELEMS * 2); // [ROWS][kvol]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int r16 = lane & 15;
    const int q = lane >> 4;
    const int k_out = lcf_count(k_cap, k_dev);
    // Workgroups go round-robin to the 8 XCDs (id % 8); XCD x takes the x-th contiguous eighth of
    // the tiles THE COUNT ON THE DEVICE gives (the launch is sized by the capacity: an eighth of
    // that would leave the upper XCDs without rows).  tiles_per_xcd = 0 keeps tile = id.
    int tile = (int)blockIdx.x;
    if (tiles_per_xcd) {
        const int per = ((k_out + ROWS - 1) / ROWS + 7) / 8;
        if ((int)(blockIdx.x >> 3) >= per) return;
        tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    }
    const int row0 = tile * ROWS;
    if (row0 >= k_out) return;

    // the workgroup's rows of the neighbour table: one contiguous run of ints
    {
        const int live = (k_out - row0 < ROWS ? k_out - row0 : ROWS) * kvol;
        const int *src = nbr + (size_t)row0 * kvol;
        for (int e = tid; e < ROWS * kvol; e += 256) nl[e] = e < live ? src[e] : -1;
    }

    // piece e of W[o]'s three planes: plane e / WV, piece e % WV
    const uint4 *wg = reinterpret_cast<const uint4 *>(wimg);
    const unsigned wplane_v = wplane_elems / 8;
    auto wsrc = [&](int o, int e) -> const uint4 * {
        const int p = e / WV, r = e - p * WV;
        return wg + (size_t)p * wplane_v + (size_t)o * WV + r;
    };
    // W[x] travels global -> registers -> LDS one step ahead of its use: step o stores W[o+1]
    // (requested during step o-1) and requests W[o+2].  (Named registers, not a struct or an
    // array: those were left in scratch memory, and a scratch load drains vmcnt like any other.
    // A second register set -- two steps per request -- measured no different.)
    uint4 w0 = make_uint4(0, 0, 0, 0), w1 = w0, w2 = w0, w3 = w0, w4 = w0, w5 = w0;
    auto wload = [&](int o) {
        const int oo = o < kvol ? o : kvol - 1;
        if (WV3 >= 256 || tid < WV3) w0 = *wsrc(oo, tid);
        if constexpr (WPT > 1) { if (WV3 >= 512 || tid + 256 < WV3) w1 = *wsrc(oo, tid + 256); }
        if constexpr (WPT > 2) { if (WV3 >= 768 || tid + 512 < WV3) w2 = *wsrc(oo, tid + 512); }
        if constexpr (WPT > 3) { if (WV3 >= 1024 || tid + 768 < WV3) w3 = *wsrc(oo, tid + 768); }
        if constexpr (WPT > 4) { if (WV3 >= 1280 || tid + 1024 < WV3) w4 = *wsrc(oo, tid + 1024); }
        if constexpr (WPT > 5) { if (WV3 >= 1536 || tid + 1280 < WV3) w5 = *wsrc(oo, tid + 1280); }
    };
    auto wstore = [&](int o) {
        uint4 *wn = wl + (o & 1) * WV3 + tid;
        if (WV3 >= 256 || tid < WV3) wn[0] = w0;
        if constexpr (WPT > 1) { if (WV3 >= 512 || tid + 256 < WV3) wn[256] = w1; }
        if constexpr (WPT > 2) { if (WV3 >= 768 || tid + 512 < WV3) wn[512] = w2; }
        if constexpr (WPT > 3) { if (WV3 >= 1024 || tid + 768 < WV3) wn[768] = w3; }
        if constexpr (WPT > 4) { if (WV3 >= 1280 || tid + 1024 < WV3) wn[1024] = w4; }
        if constexpr (WPT > 5) { if (WV3 >= 1536 || tid + 1280 < WV3) wn[1280] = w5; }
    };

    // range-checked descriptors, one per plane of the rows: a row without a neighbour takes an
    // offset beyond the range -- zeros, no memory access (conv_bt.h)
    typedef unsigned bx_u4 __attribute__((ext_vector_type(4)));
    const char *fb = reinterpret_cast<const char *>(feat);
    const __amdgpu_buffer_rsrc_t fr0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(fb), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fr1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(fb + plane_bytes), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t fr2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(fb + 2 * (size_t)plane_bytes), 0, plane_bytes, 0x00020000);
    struct AReg {
        uint4 h[G][KS], m[G][KS], l[G][KS];
        int idx[G]; // neighbour, -1 = none
    };
    // rows of offset o (clamped: the last steps reload the last offset, harmless)
    auto aload = [&](int o, AReg &a) {
        const int oo = o < kvol ? o : kvol - 1;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int idx = nl[((wave * G + g) * 16 + r16) * kvol + oo];
            a.idx[g] = idx;
#if defined(BX_ABLATE) && (BX_ABLATE & 2) // timing ablation: every gather from 64 cached rows
            const int voff = idx < 0 ? (int)0x80000000 : (idx & 63) * (CIN * 2) + q * 16;
#else
            const int voff = idx < 0 ? (int)0x80000000 : idx * (CIN * 2) + q * 16;
#endif
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bx_u4 x = __builtin_amdgcn_raw_buffer_load_b128(fr0, voff + ks * 64, 0, 0);
                const bx_u4 y = __builtin_amdgcn_raw_buffer_load_b128(fr1, voff + ks * 64, 0, 0);
                const bx_u4 z = __builtin_amdgcn_raw_buffer_load_b128(fr2, voff + ks * 64, 0, 0);
                a.h[g][ks] = make_uint4(x.x, x.y, x.z, x.w);
                a.m[g][ks] = make_uint4(y.x, y.y, y.z, y.w);
                a.l[g][ks] = make_uint4(z.x, z.y, z.z, z.w);
            }
        }
    };

    f32x4 acc[G][NT];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // W[0] into LDS, W[1] requested; rows of offsets 0 and 1 requested.  A round trip to L2 is
    // ~0.6 us here and a step's MFMAs 0.35: the rows are requested TWO steps ahead (three register
    // stages), with one step the kernel waited for memory at every offset.
    wload(0);
    wstore(0);
    wload(1);
    __syncthreads(); // (the table rows and W[0])
    AReg a0, a1, a2;
    aload(0, a0);
    aload(1, a1);

    auto step = [&](int o, const AReg &cur, AReg &fill) {
#if !(defined(BX_ABLATE) && (BX_ABLATE & 4)) // timing ablation: W[o] not staged after the first two
        wstore(o + 1); // W[o+1] -> the buffer W[o-1] was read from (everyone is past that barrier)
        wload(o + 2);
#endif
        aload(o + 2, fill);
        const uint4 *wb = wl + (o & 1) * WV3;
        bool on[G];
        bool any = false;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            on[g] = __ballot(cur.idx[g] >= 0) != 0ull; // (a group without any pair here skips its MFMAs)
            any |= on[g];
        }
#if defined(BX_ABLATE) && (BX_ABLATE & 1) // timing ablation: no fragment reads, no MFMAs
        if (cur.idx[0] == 0x7fffffff)
            acc[0][0][0] += __uint_as_float(cur.h[0][0].x ^ cur.m[G - 1][KS - 1].y ^ cur.l[0][0].z);
        any = false;
#endif
        if (any) {
            // The offset's products are summed in accumulators of their own, starting from zero,
            // and added to the running sums with fp32 VALU adds (round to nearest even).  The bf16
            // MFMA does not round its accumulator that way: 54 accumulations into ONE register
            // measured 4e-7 .. 1e-6 of the result's magnitude from float64, the fp32 MFMA kernels
            // 1e-7 .. 2e-7.
            f32x4 part[G][NT];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int t = 0; t < NT; ++t) part[g][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // fragment (plane p, tile t, k-step ks): [p][t][ks][q][r16] pieces of 8 bf16
                bf16x8 wh[NT], wm[NT], wlo[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int e = ((t * KS + ks) * 4 + q) * 16 + r16;
                    wh[t] = __builtin_bit_cast(bf16x8, wb[e]);
                    wm[t] = __builtin_bit_cast(bf16x8, wb[WV + e]);
                    wlo[t] = __builtin_bit_cast(bf16x8, wb[2 * WV + e]);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (!on[g]) continue;
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, cur.h[g][ks]);
                    const bf16x8 am = __builtin_bit_cast(bf16x8, cur.m[g][ks]);
                    const bf16x8 al = __builtin_bit_cast(bf16x8, cur.l[g][ks]);
                    // smallest terms first; consecutive MFMAs go to different accumulators
#define BX_MAC(A, B)                                                                          \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) part[g][t] =                               \
        __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B[t], part[g][t], 0, 0, 0)
                    BX_MAC(al, wh);
                    BX_MAC(ah, wlo);
                    BX_MAC(am, wm);
                    BX_MAC(am, wh);
                    BX_MAC(ah, wm);
                    BX_MAC(ah, wh);
#undef BX_MAC
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[g][t] += part[g][t];
        }
        // LDS accesses done, then the barrier -- NOT __syncthreads(), whose fence also drains
        // vmcnt(0), i.e. would wait for the rows and the W[o+2] just requested
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    int o = 0;
    for (; o + 2 < kvol; o += 3) { // the row registers rotate through a0, a1, a2
        step(o, a0, a2);
        step(o + 1, a1, a0);
        step(o + 2, a2, a1);
    }
    if (o < kvol) step(o, a0, a2);
    if (o + 1 < kvol) step(o + 1, a1, a0);

    // ---- epilogue: lane (col r16, q) holds rows 4q .. 4q+3 of each 16 x 16 tile
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int rbase = row0 + (wave * G + g) * 16 + q * 4;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 16 + r16;
            const float sc = scale ? scale[col] : 1.f;
            const float sh = shift ? shift[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc[g][t][i];
                if (scale) v *= sc;
                if (shift) v += sh;
                if (relu) v = fmaxf(v, 0.f);
                acc[g][t][i] = v;
            }
        }
        if constexpr (!OUT_PLANES) {
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
            // two adjacent columns per 4-byte store: even lanes take rows 4q, 4q+2, odd lanes rows
            // 4q+1, 4q+3 of the column pair (r16 & ~1, +1); the same for each of the three planes
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
                    float lo = odd ? got : mine, hi = odd ? mine : got;
                    const size_t at = ((size_t)row * COUT + t * 16 + (r16 & ~1)) >> 1;
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const unsigned bl = lcf_bf16_rne(lo), bh = lcf_bf16_rne(hi);
                        if (row < k_out)
                            o16[(size_t)p * (out_plane_elems >> 1) + at] = bl | (bh << 16);
                        lo -= __uint_as_float(bl << 16);
                        hi -= __uint_as_float(bh << 16);
                    }
                }
            }
        }
    }
}

// fp32 rows -> three bf16 planes [3][plane_rows][c] (rows from a kernel that writes fp32)
__global__ void __launch_bounds__(256)
bx_split_kernel(const float *__restrict__ x, int k_cap, const int *__restrict__ k_dev, int c,
                unsigned *__restrict__ planes, unsigned plane_elems)
{
    const int k = lcf_count(k_cap, k_dev);
    const size_t total = (size_t)k * c / 2; // pairs of channels
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        float2 v = reinterpret_cast<const float2 *>(x)[e];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned a = lcf_bf16_rne(v.x), b = lcf_bf16_rne(v.y);
            planes[(size_t)p * (plane_elems >> 1) + e] = a | (b << 16);
            v.x -= __uint_as_float(a << 16);
            v.y -= __uint_as_float(b << 16);
        }
    }
}

extern "C" int lcf_split_planes(const float *feat, int k, const int *k_dev, int c, void *planes,
                                int plane_rows, void *stream)
{
    if (k < 0 || c <= 0 || (c & 1) || plane_rows < k || !planes || (k > 0 && !feat))
        return LCF_ERR_ARG;
    if (((uintptr_t)feat & 7) || ((uintptr_t)planes & 3)) return LCF_ERR_ARG;
    if (k == 0) return LCF_OK;
    const size_t total = (size_t)k * c / 2;
    size_t nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    bx_split_kernel<<<(unsigned)nb, 256, 0, (hipStream_t)stream>>>(
        feat, k, k_dev, c, (unsigned *)planes, (unsigned)plane_rows * (unsigned)c);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

template <int CIN, int COUT, int G>
static int cv_launch_bx_g(const void *feat, unsigned plane_bytes, const void *wimg,
                          unsigned wplane_elems, const int *nbr, int k_cap, const int *k_dev,
                          int kvol, const float *scale, const float *shift, int relu, void *out,
                          int out_planes, unsigned out_plane_elems, hipStream_t st)
{
    const size_t lds = (size_t)2 * 3 * CIN * COUT * 2 + (size_t)64 * G * kvol * 4;
    const int tiles = (k_cap + 64 * G - 1) / (64 * G);
    static const bool xcd_off = getenv("LCF_BX_XCD") && atoi(getenv("LCF_BX_XCD")) == 0;
    const int per = xcd_off ? 0 : (tiles + 7) / 8;
    const unsigned grid = (unsigned)(per ? 8 * per : tiles);
    if (out_planes)
        cv_bx_kernel<CIN, COUT, G, true><<<grid, 256, lds, st>>>(
            (const __bf16 *)feat, plane_bytes, (const __bf16 *)wimg, wplane_elems, nbr, k_cap,
            k_dev, kvol, scale, shift, relu, out, out_plane_elems, per);
    else
        cv_bx_kernel<CIN, COUT, G, false><<<grid, 256, lds, st>>>(
            (const __bf16 *)feat, plane_bytes, (const __bf16 *)wimg, wplane_elems, nbr, k_cap,
            k_dev, kvol, scale, shift, relu, out, out_plane_elems, per);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

template <int CIN, int COUT>
static int cv_launch_bx(const void *feat, unsigned plane_bytes, const void *wimg,
                        unsigned wplane_elems, const int *nbr, int k_cap, const int *k_dev,
                        int kvol, const float *scale, const float *shift, int relu, void *out,
                        int out_planes, unsigned out_plane_elems, hipStream_t st)
{
    // one group of 16 rows per wave.  (Two groups per wave, sharing every fragment read, measured
    // 79 against 69 us at 64 -> 64 / 38k rows: half as many waves to overlap one another's waits.)
    return cv_launch_bx_g<CIN, COUT, 1>(feat, plane_bytes, wimg, wplane_elems, nbr, k_cap, k_dev,
                                        kvol, scale, shift, relu, out, out_planes, out_plane_elems,
                                        st);
}

extern "C" int lcf_indice_conv_fwd_x6(const void *feat_planes, int plane_rows,
                                      const void *wimg_planes, const int *nbr, int k_out,
                                      const int *k_out_dev, int kvol, int cin, int cout,
                                      const float *scale, const float *shift, int relu, void *out,
                                      int out_planes, int out_plane_rows, void *stream)
{
    if (plane_rows < 0 || (unsigned long long)plane_rows * (unsigned)cin * 2ull >= 0x80000000ull)
        return LCF_ERR_RANGE;
    if (k_out < 0 || (unsigned long long)k_out * (unsigned)kvol * 4ull >= 0x80000000ull)
        return LCF_ERR_RANGE;
    if (kvol <= 0 || kvol > LCF_MAX_KVOL || !wimg_planes || !nbr || !out ||
        (k_out > 0 && !feat_planes) || (out_planes && out_plane_rows < k_out))
        return LCF_ERR_ARG;
    if (((uintptr_t)feat_planes & 15) || ((uintptr_t)wimg_planes & 15) || ((uintptr_t)out & 15) ||
        (((size_t)plane_rows * cin * 2) & 15))
        return LCF_ERR_ARG;
    if (k_out == 0) return LCF_OK;
    const unsigned plane_bytes = (unsigned)plane_rows * (unsigned)cin * 2u;
    const unsigned wplane_elems = (unsigned)kvol * (unsigned)cin * (unsigned)cout;
    const unsigned out_plane_elems = (unsigned)out_plane_rows * (unsigned)cout;
    hipStream_t st = (hipStream_t)stream;
#define BX_CASE(A, B)                                                                         \
    if (cin == A && cout == B)                                                                \
        return cv_launch_bx<A, B>(feat_planes, plane_bytes, wimg_planes, wplane_elems, nbr,   \
                                  k_out, k_out_dev, kvol, scale, shift, relu, out, out_planes, \
                                  out_plane_elems, st)
    BX_CASE(32, 32);
    BX_CASE(32, 64);
    BX_CASE(64, 64);
#undef BX_CASE
    return LCF_ERR_UNSUPPORTED;
}
