// conv_q4.h -- indice_conv, quad-row form (cv_q4_kernel; included by conv.hip).
//
// The tile-compacted kernel pads every (tile, offset) to 16-row groups because v_mfma_f32_16x16x4
// has 16 rows: at 14 rows per offset and 32-row tile that is 70-75 % efficiency, and it stages the
// gathered rows through LDS with a barrier per offset.  v_mfma_f32_4x4x1_16b_f32 is sixteen
// independent 4x4 rank-1 updates at the same flop rate (512 flop in 2 passes); with the A-matrix
// broadcast (cbsz = 4, abid = b) all sixteen blocks take block b's A operand, which makes ONE
// instruction a 4-row x 64-column rank-1 update:
//     D[vgpr i][lane l] += A[lane 4b + i] * B[lane l]      (A = 4 rows at channel k, B = W[k][l])
// So groups are 4 rows (8 at 32 output columns: two row quads x 8 column blocks, cbsz = 3), a
// group's rows are ONE load per lane straight from global memory into the A registers (lane
// (i, b) holds CPL consecutive channels of row i: no LDS staging, no barrier, no redundancy), and
// abid walks the channels.
// A workgroup is one tile of <= 32 rows; its NSETS waves each take every NSETS-th active offset
// alone: W[o] in Cin registers (read from the same fragment-order image as the other kernels),
// an output tile of its own in LDS (rows added in ascending offset order; the tiles are summed
// wave 0 .. NSETS-1 in the epilogue: deterministic), and no barrier between set-up and epilogue.
// (KS > 1 splits the input channels over KS waves per set, so that the next step's W fits beside
// this step's; built and measured, not instantiated: twice the output tiles and gathers cost
// more than the hidden round trip gains.)
// tools/ubench/mfma4x4.hip: semantics of the broadcast and the rate (146 cycles per 16
// instructions with four chains, 137 with two waves per SIMD; 128 is the 16x16x4 rate).
// What it measures (tools/bench_q4.py): level with the tile-compacted kernel
// on the submanifold layers, ahead of it on the strided ones (few rows per offset: 16-row groups
// are mostly padding there); inside the frame, beside the rulebook branch, ahead on all of them
// (0.695 -> 0.636 ms per frame, tools/ab_q4.py): the default for 27-offset convs with 32 / 64
// channels.  LCF_OPT_CONV_IMPL = 2 keeps the tile-compacted kernel, 5 asks for this one;
// LCF_OPT_CONV_SETS = 1..4 waves per tile (default 3 at 32 input channels, 4 at 64).
// -DQ4_ABLATE=<bits> (round 3; timing only, results are wrong): 1 = W[o] loaded once per wave,
// 2 = every gather from 64 cached rows, 4 = results summed in registers (no LDS tile), 8 = set-up
// and epilogue only, 16 = no MFMAs.  tools/ubench/q4_ablate.sh runs such builds through
// tools/bench_q4.py; DESIGN.md section 8 (round 3) has the numbers.
#pragma once

template <int CIN, int COUT, int KS_>
struct Q4Cfg {
    static constexpr int KS = KS_;                 // waves that split the input channels
    static constexpr int CW = CIN / KS;            // channels (= W registers) of a wave: 32
    static constexpr int NB = COUT / 4;            // blocks that share an A operand: 16 or 8
    static constexpr int CBSZ = NB == 16 ? 4 : 3;
    static constexpr int R = 4 * (16 / NB);        // rows per group: 4 or 8
    static constexpr int CPL = CW / NB;            // channels of its row a lane holds: 2 or 4
    static constexpr int OSTR = COUT;              // output-tile row stride (lane = column)
    static constexpr int OTE = 33 * OSTR;          // floats of one output tile (32 rows + dump)
    static constexpr int KL = CIN / 4;             // weight image: [tg][sg][q][r16][4],
    static constexpr int SG = KL / 4;              //   channel = q * KL + sg * 4 + v
};

static size_t q4_lds_bytes(int COUT, int kvol, int nw)
{
    // plan record + the waves' output tiles + scale | shift of the epilogue
    return (size_t)tc_plan_stride(kvol) * 4 + (size_t)nw * 33 * COUT * 4 + (size_t)2 * COUT * 4;
}

// sb-th block's channels of a group (or of two groups with the same W registers)
template <int CIN, int COUT, int KS, int SB>
struct Q4Mac {
    using C = Q4Cfg<CIN, COUT, KS>;
    // Every group's result is (sum over the even blocks' channels) + (sum over the odd blocks'),
    // each in ascending channel order, whether the group runs alone or beside another: with one
    // wave per tile a row's value does not depend on how the tiles were cut; with several it
    // depends on the tile's list of active offsets (which wave adds which), so the engine and the
    // module path both cut 32-row tiles and agree bit for bit.
    static __device__ __forceinline__ void two(const float (&a0)[C::CPL], const float (&a1)[C::CPL],
                                               const float (&w)[C::CW], f32x4 (&c0)[2],
                                               f32x4 (&c1)[2])
    {
#pragma unroll
        for (int j = 0; j < C::CPL; ++j) {
            c0[SB & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[j], w[SB * C::CPL + j], c0[SB & 1],
                                                            C::CBSZ, SB, 0);
            c1[SB & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[j], w[SB * C::CPL + j], c1[SB & 1],
                                                            C::CBSZ, SB, 0);
        }
        if constexpr (SB + 1 < C::NB) Q4Mac<CIN, COUT, KS, SB + 1>::two(a0, a1, w, c0, c1);
    }
    // one group: two chains over alternate blocks (a dependent chain issues at 14.6 cycles per
    // instruction, two at 9.5)
    static __device__ __forceinline__ void one(const float (&a0)[C::CPL], const float (&w)[C::CW],
                                               f32x4 &c0, f32x4 &c1)
    {
#pragma unroll
        for (int j = 0; j < C::CPL; ++j) {
            if constexpr ((SB & 1) == 0)
                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[j], w[SB * C::CPL + j], c0, C::CBSZ, SB, 0);
            else
                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[j], w[SB * C::CPL + j], c1, C::CBSZ, SB, 0);
        }
        if constexpr (SB + 1 < C::NB) Q4Mac<CIN, COUT, KS, SB + 1>::one(a0, w, c0, c1);
    }
};

template <int CIN, int COUT, int NSETS, int KS_>
__global__ void __launch_bounds__(NSETS * KS_ * 64)
cv_q4_kernel(const float *__restrict__ feat, const float *__restrict__ wimg,
             const int *__restrict__ nbr, int k_cap, const int *__restrict__ k_dev, int kvol,
             const float *__restrict__ scale, const float *__restrict__ shift, int relu,
             float *__restrict__ out, int tr, TcDense dn, const int *__restrict__ plan,
             int plan_stride)
{
    using C = Q4Cfg<CIN, COUT, KS_>;
    constexpr int NB = C::NB, R = C::R, CPL = C::CPL, OSTR = C::OSTR, OTE = C::OTE;
    constexpr int KS = C::KS, CW = C::CW;
    constexpr int NW = NSETS * KS;
    constexpr int TT = NW * 64;
    constexpr int DUMP = 32;

    const int k_out = lcf_count(k_cap, k_dev);
    const int row0 = blockIdx.x * tr;
    if (row0 >= k_out) return;

    extern __shared__ __align__(16) unsigned char smem[];
    unsigned *ent = reinterpret_cast<unsigned *>(smem);          // the tile's plan record
    int *cnt = reinterpret_cast<int *>(ent + 32 * kvol);
    int *gbase = cnt + 64;
    int *soff = gbase + 64;
    int *nsteps_p = soff + 64;
    const int rec = (32 * kvol + 193 + 3) & ~3;                  // = tc_plan_stride(kvol)
    float *ot0 = reinterpret_cast<float *>(smem) + rec;          // [NW][33][OSTR]

    const int tidw = threadIdx.x;
    const int lane = tidw & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tidw >> 6);
    const int set = wave / KS;      // steps set, set + NSETS, ...
    const int kh = wave % KS;       // input channels [kh * CW, (kh + 1) * CW)
    float *ot = ot0 + wave * OTE;

    // scale | shift of the epilogue travel to LDS with the set-up (read from global memory in the
    // epilogue they were one more round trip at the end of every tile's life)
    float *ssl = ot0 + NW * OTE; // [2][COUT]
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (plan) {
        // the record's loads are in flight while the output tiles are cleared: one barrier
        const int4 *src = reinterpret_cast<const int4 *>(plan + (size_t)blockIdx.x * plan_stride);
        int4 *dst = reinterpret_cast<int4 *>(ent);
        const int nv = plan_stride / 4; // <= 2 * TT at 27 offsets with >= 3 waves, else a loop
        const int4 r0 = tidw < nv ? src[tidw] : make_int4(0, 0, 0, 0);
        const int4 r1 = tidw + TT < nv ? src[tidw + TT] : make_int4(0, 0, 0, 0);
        auto ss_at = [&](int e) -> float {
            return e < COUT ? (scale ? scale[e] : 1.f) : (shift ? shift[e - COUT] : 0.f);
        };
        const float sv0 = tidw < 2 * COUT ? ss_at(tidw) : 0.f;             // (TT >= 64,
        const float sv1 = tidw + TT < 2 * COUT ? ss_at(tidw + TT) : 0.f;    //  2 COUT <= 128)
        for (int e = tidw; e < NW * OTE / 4; e += TT) reinterpret_cast<float4 *>(ot0)[e] = zero4;
        if (tidw < nv) dst[tidw] = r0;
        if (tidw + TT < nv) dst[tidw + TT] = r1;
        for (int e = tidw + 2 * TT; e < nv; e += TT) dst[e] = src[e];
        if (tidw < 2 * COUT) ssl[tidw] = sv0;
        if (tidw + TT < 2 * COUT) ssl[tidw + TT] = sv1;
    } else {
        tc_build_steps<NW, TT>(nbr, row0, tr, kvol, k_out, reinterpret_cast<int *>(ot0), ent, cnt,
                               gbase, soff, nsteps_p, DUMP);
        // (the neighbour tile is dead now)
        for (int e = tidw; e < NW * OTE / 4; e += TT) reinterpret_cast<float4 *>(ot0)[e] = zero4;
        for (int e = tidw; e < 2 * COUT; e += TT)
            ssl[e] = e < COUT ? (scale ? scale[e] : 1.f) : (shift ? shift[e - COUT] : 0.f);
    }
    __syncthreads();
    const int n_steps = __builtin_amdgcn_readfirstlane(*nsteps_p);

    // lane -> (row of the group, block = channel slot, output column)
    const int h = COUT == 64 ? 0 : (lane >> 5);     // row quad of the group (8-row groups)
    const int rg = h * 4 + (lane & 3);
    const int sb = (lane >> 2) & (NB - 1);
    const int col = lane & (COUT - 1);
    const float *fbase = feat + kh * CW + sb * CPL;

    struct AReg { float v[CPL]; };
    auto gather = [&](int ibase, int g) -> AReg {
        AReg a;
        const unsigned e = ent[ibase + R * g + rg];
#if defined(Q4_ABLATE) && (Q4_ABLATE & 2) // timing ablation: every gather hits one of 64 cached rows
        const float *src = fbase + (size_t)((e >> 7) & 63) * CIN;
#else
        const float *src = fbase + (size_t)(e >> 7) * CIN;
#endif
        if constexpr (CPL == 8) {
            const float4 x = reinterpret_cast<const float4 *>(src)[0];
            const float4 y = reinterpret_cast<const float4 *>(src)[1];
            a.v[0] = x.x; a.v[1] = x.y; a.v[2] = x.z; a.v[3] = x.w;
            a.v[4] = y.x; a.v[5] = y.y; a.v[6] = y.z; a.v[7] = y.w;
        } else if constexpr (CPL == 4) {
            const float4 x = *reinterpret_cast<const float4 *>(src);
            a.v[0] = x.x; a.v[1] = x.y; a.v[2] = x.z; a.v[3] = x.w;
        } else {
            const float2 x = *reinterpret_cast<const float2 *>(src);
            a.v[0] = x.x; a.v[1] = x.y;
        }
        return a;
    };
    // W[o][k][col] for this wave's channels k = kh * CW + c, from the fragment-order image
    // [o][tg][sg][q][r16][4] (channel = q * KL + sg * 4 + v): piece m = channels [4m, 4m + 4) of
    // the wave = q (4 / KS) * kh + 4m / KL, sg = (4m % KL) / 4
    const int wlane = ((col >> 4) * C::SG * 4 + kh * (4 / KS)) * 16 + (col & 15);
    auto bload = [&](int o, float (&w)[CW]) {
        const float4 *wo =
            reinterpret_cast<const float4 *>(wimg + (size_t)o * CIN * COUT) +
            wlane;
#pragma unroll
        for (int m = 0; m < CW / 4; ++m) {
            const int q = (4 * m) / C::KL, sg = ((4 * m) % C::KL) / 4;
            const float4 x = wo[(sg * 4 + q) * 16];
            w[4 * m + 0] = x.x;
            w[4 * m + 1] = x.y;
            w[4 * m + 2] = x.z;
            w[4 * m + 3] = x.w;
        }
    };
    // tile[output rows of the group][col] += result (vgpr i = row 4h + i of the group)
#if defined(Q4_ABLATE) && (Q4_ABLATE & 4) // timing ablation: results summed in registers, no LDS tile
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
#endif
    auto add_out = [&](const f32x4 &acc, int ibase, int g) {
#if defined(Q4_ABLATE) && (Q4_ABLATE & 4)
        sink += acc;
        return;
#endif
        const uint4 e4 = *reinterpret_cast<const uint4 *>(ent + ibase + R * g + 4 * h);
        float *p0 = ot + (e4.x & 127u) * OSTR + col;
        float *p1 = ot + (e4.y & 127u) * OSTR + col;
        float *p2 = ot + (e4.z & 127u) * OSTR + col;
        float *p3 = ot + (e4.w & 127u) * OSTR + col;
        const float c0 = *p0, c1 = *p1, c2 = *p2, c3 = *p3;
        *p0 = c0 + acc[0];
        *p1 = c1 + acc[1];
        *p2 = c2 + acc[2];
        *p3 = c3 + acc[3];
    };

#if defined(Q4_ABLATE) && (Q4_ABLATE & 8) // timing ablation: set-up and epilogue only
    if (false) {
#else
    if (set < n_steps) {
#endif
        // Everything that steers is wave-uniform (SGPRs).  Step metadata of this step and the next.
        int s = set;
        int o = __builtin_amdgcn_readfirstlane(soff[s]);
        int ng = (__builtin_amdgcn_readfirstlane(cnt[o]) + R - 1) / R;
        int ib = __builtin_amdgcn_readfirstlane(gbase[o]) * 16;
        bool have_next;
        int no = o, nng = 0, nib = 0;
        auto next_meta = [&]() {
            have_next = s + NSETS < n_steps;
            no = __builtin_amdgcn_readfirstlane(soff[have_next ? s + NSETS : s]);
            nng = (__builtin_amdgcn_readfirstlane(cnt[no]) + R - 1) / R;
            nib = __builtin_amdgcn_readfirstlane(gbase[no]) * 16;
        };
        next_meta();
        // Groups go in pairs (one W register feeds two independent accumulator chains).  The
        // rows of the NEXT pair -- of this step or the first of the next -- are loaded behind
        // this pair's MFMAs; both loads always go (a missing group repeats its neighbour:
        // cached), so that the number of loads in flight at every wait is known when the kernel
        // is compiled (behind a branch the waits would have to assume they were not issued and
        // drain the queue).  W[o] is reloaded at the step boundary: that round trip is covered by
        // the SIMD's other waves, not by this one.
        float w[CW];
        bload(o, w);
        AReg n0 = gather(ib, 0), n1 = gather(ib, ng > 1 ? 1 : 0);
        int g = 0;
        for (;;) {
            const AReg a0 = n0, a1 = n1; // waits for this pair's rows
            // the pair after this one
            const bool newstep = g + 2 >= ng;
            const bool more = !newstep || have_next;
            const int pib = newstep ? nib : ib;
            const int png = newstep ? nng : ng;
            const int pg = newstep ? 0 : g + 2;
            n0 = gather(pib, pg);
            n1 = gather(pib, pg + 1 < png ? pg + 1 : pg);
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#if defined(Q4_ABLATE) && (Q4_ABLATE & 16) // timing ablation: no MFMAs (operands kept alive)
            if (true) {
                c0[0] = a0.v[0] + w[0] + w[CW - 1];
                c1[0] = a1.v[CPL - 1] + w[1] + w[CW / 2];
                add_out(c0, ib, g);
                if (g + 1 < ng) add_out(c1, ib, g + 1);
            } else
#endif
            if (g + 1 < ng) {
                f32x4 p0[2] = {c0, c0}, p1[2] = {c0, c0};
                Q4Mac<CIN, COUT, KS, 0>::two(a0.v, a1.v, w, p0, p1);
                c0 = p0[0] + p0[1];
                c1 = p1[0] + p1[1];
                add_out(c0, ib, g);
                add_out(c1, ib, g + 1);
            } else {
                Q4Mac<CIN, COUT, KS, 0>::one(a0.v, w, c0, c1);
                c0[0] += c1[0];
                c0[1] += c1[1];
                c0[2] += c1[2];
                c0[3] += c1[3];
                add_out(c0, ib, g);
            }
            if (!more) break;
            g = pg;
            if (newstep) {
#if !(defined(Q4_ABLATE) && (Q4_ABLATE & 1)) // timing ablation: W[o] loaded once per wave
                bload(no, w);
#endif
                s += NSETS;
                o = no;
                ng = nng;
                ib = nib;
                next_meta();
            }
        }
    }
#if defined(Q4_ABLATE) && (Q4_ABLATE & 4)
    ot[col] = sink[0] + sink[1] + sink[2] + sink[3];
#endif
    __syncthreads();

    // ---- epilogue: sum of the waves' tiles (wave 0 first), (x*scale + shift), relu, store
    constexpr int C4 = COUT / 4;
    for (int e = tidw; e < tr * C4; e += TT) {
        const int r = e / C4, c4 = e % C4;
        const int row = row0 + r;
        if (row < k_out) {
            float4 v = *reinterpret_cast<const float4 *>(ot0 + r * OSTR + c4 * 4);
#pragma unroll
            for (int x = 1; x < NW; ++x) {
                const float4 v1 =
                    *reinterpret_cast<const float4 *>(ot0 + x * OTE + r * OSTR + c4 * 4);
                v.x += v1.x; v.y += v1.y; v.z += v1.z; v.w += v1.w;
            }
            if (scale) {
                const float4 s = *reinterpret_cast<const float4 *>(ssl + c4 * 4);
                v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
            }
            if (shift) {
                const float4 s = *reinterpret_cast<const float4 *>(ssl + COUT + c4 * 4);
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
                    float *d = out + ((size_t)q.x * COUT + c4 * 4) * svol +
                               ((size_t)q.y * dn.H + q.z) * dn.W + q.w;
                    d[0] = v.x;
                    d[svol] = v.y;
                    d[2 * svol] = v.z;
                    d[3 * svol] = v.w;
                }
            } else {
                *reinterpret_cast<float4 *>(out + (size_t)row * COUT + c4 * 4) = v;
            }
        }
    }
}

template <int CIN, int COUT>
static constexpr bool q4_shape_ok()
{
    return (CIN == 32 || CIN == 64) && (COUT == 32 || COUT == 64);
}

template <int CIN, int COUT, int NSETS, int KS>
static int cv_launch_q4_n(const float *feat, const float *wimg, const int *nbr, int k_cap,
                          const int *k_dev, int kvol, const float *scale, const float *shift,
                          int relu, float *out, hipStream_t st, int tr, TcDense dn)
{
    constexpr int NW = NSETS * KS;
    const size_t lds = q4_lds_bytes(COUT, kvol, NW);
    if (tr < 1 || tr > 32) return LCF_ERR_ARG;
    static size_t lds_set = 0; // per instantiation: largest dynamic LDS size opted into so far
    static std::mutex lds_mu;
    {
        std::lock_guard<std::mutex> lock(lds_mu);
        if (lds > 64 * 1024 && lds > lds_set) {
            if (hipFuncSetAttribute((const void *)cv_q4_kernel<CIN, COUT, NSETS, KS>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return LCF_ERR_LAUNCH;
            lds_set = lds;
        }
    }
    const LcfOpts op = lcf_opts();
    const int *plan = op.plan;
    const int pstride = plan ? tc_plan_stride(kvol) : 0;
    if (plan && op.plan_tr != tr) return LCF_ERR_ARG;
    const int nb = (k_cap + tr - 1) / tr;
    cv_q4_kernel<CIN, COUT, NSETS, KS><<<nb, NW * 64, lds, st>>>(feat, wimg, nbr, k_cap, k_dev, kvol,
                                                             scale, shift, relu, out, tr, dn,
                                                             plan, pstride);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

template <int CIN, int COUT>
static int cv_launch_q4(const float *feat, const float *wimg, const int *nbr, int k_cap,
                        const int *k_dev, int kvol, const float *scale, const float *shift,
                        int relu, float *out, hipStream_t st, int tr, TcDense dn)
{
    if constexpr (q4_shape_ok<CIN, COUT>()) {
        const int sets = lcf_opts().conv_sets;
#define Q4_GO(NS)                                                                                  \
    return cv_launch_q4_n<CIN, COUT, NS, 1>(feat, wimg, nbr, k_cap, k_dev, kvol, scale, shift,     \
                                            relu, out, st, tr, dn)
        if (sets == 1) Q4_GO(1);
        if (sets == 2) Q4_GO(2);
#ifdef Q4_EXP
        {
            const LcfOpts op = lcf_opts();
            const int rows = op.rows_hint > 0 && op.rows_hint < k_cap ? op.rows_hint : k_cap;
            const int tiles = (rows + tr - 1) / tr;
            // three waves: 29.6 KB of LDS, five workgroups per CU -- every tile resident at once
            if (sets == 0 && CIN == 64 && tiles > 1024 && tiles <= 1280) Q4_GO(3);
        }
#endif
        if (sets == 3 || (sets == 0 && CIN == 32)) Q4_GO(3);
        Q4_GO(4); // (5 and 6 waves lose to 4, 8 is level with it: tools/ab_q4.py)
#undef Q4_GO
    }
    return LCF_ERR_UNSUPPORTED;
}
