// lc_policy.hip -- the light-curtain placement step next to the detector:
// entropy of the detector's confidence map and the dynamic programme that picks one design point
// per camera ray, on the device, so that LC_PROCESS (second/light_curtain/utils.py:88-104) no
// longer brings the class scores to the host between two detector passes.
//
// Follows second/light_curtain/utils.py:177-237 (_get_confidence_map: masking, the flip to
// increasing camera X) + second/light_curtain/policy.py:60-66 (confidence2entropy) for the
// look-up table, and policy.py:283-434 (DPOptimizedPolicy.get_design_points) for the DP:
//   backward over the rays t = T-2 .. 0, N points per ray,
//     q_unc[i][j] = r_unc[i] + v_unc'[j]              if |dtheta_las(i, j)| < max_dtheta else -inf
//     v_unc[i]    = max_j q_unc[i][j]                                        (float32, as numpy)
//     q_las[i][j] = dtheta_las(i, j)^2 + v_las'[j]    if q_unc[i][j] == v_unc[i] else +inf
//     v_las[i], next[i] = min / first argmin over j                          (float64, as numpy)
//   forward: start at the first i with v_unc maximal and, among those, v_las minimal; follow next.
// The graph (node -> map cell k, laser angle of every node) depends on the device geometry only
// and is built once on the host (policy.py:160-281), exactly as the reference caches it.
#include "common.h"
#include <mutex>

#define LCP_THREADS 1024
#define LCP_MAXN 256

// ulookup[0] = 0 (points outside the map), ulookup[1 + xi * Z + zi] = entropy of the cell at
// camera-x index xi = Y - 1 - y (np.flip, utils.py:235) and camera-z index zi = x; Z = X.
__global__ void __launch_bounds__(256)
lc_entropy_kernel(const float *__restrict__ cls, int K, int Y, int X,
                  const unsigned char *__restrict__ mask, int apply_sigmoid,
                  float *__restrict__ ulookup)
{
#pragma clang fp contract(off) // numpy rounds every product and sum; no fused multiply-adds
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t == 0) ulookup[0] = 0.f;
    if (t >= Y * X) return;
    const int y = t / X, x = t - y * X;
    bool in_mask = true; // utils.py:213-219: a location counts only if ALL its anchors are in
    if (mask)
        for (int k = 0; k < K; ++k) in_mask = in_mask && mask[((size_t)k * Y + y) * X + x] != 0;
    const float lo = 1e-5f, hi = 1.f - 1e-5f; // policy.py:62
    float sum = 0.f;
    for (int k = 0; k < K; ++k) {
        float p = cls[((size_t)k * Y + y) * X + x];
        if (apply_sigmoid) p = 1.f / (1.f + expf(-p)); // utils.py:95 (torch.sigmoid)
        if (!in_mask) p = 0.f;
        p = fminf(fmaxf(p, lo), hi);
        const float e = -p * log2f(p) - (1.f - p) * log2f(1.f - p); // policy.py:63
        sum += e;
    }
    ulookup[1 + (size_t)(Y - 1 - y) * X + x] = sum / (float)K; // policy.py:64 (mean over K)
}

struct MinIdx {
    double v;
    int i;
};

__device__ __forceinline__ MinIdx lcp_min(MinIdx a, MinIdx b)
{
    // numpy argmin: the first index of the minimum
    if (b.v < a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

// Per node (t, i), in parallel over the whole chip before the serial walk:
//   r_unc = ulookup[node_k] (policy.py:352), so the walk reads one stream instead of chasing
//   node_k -> ulookup on its critical path;
//   win = (lo | hi << 8), the first and last successor j that the velocity limit allows
//   (policy.py:266-268), when the allowed set is the whole non-empty interval [lo, hi] (laser
//   angles are monotonic along a ray, so it is); (1 | 0 << 8) = "no window" otherwise (no allowed
//   successor, or a set with holes): the walk then scans all N successors with the test.
__global__ void __launch_bounds__(256)
lc_prepare_kernel(const float *__restrict__ ulookup, int n_lookup,
                  const int *__restrict__ node_k,
                  const double *__restrict__ theta_las, int T, int N, double max_dtheta,
                  float *__restrict__ r_unc, unsigned short *__restrict__ win)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= T * N) return;
    // (a graph built for another map shape must not read beyond the table: an index outside it
    // counts as "outside the map", entry 0, whose value is 0)
    const int k = node_k[e];
    r_unc[e] = ulookup[(unsigned)k < (unsigned)n_lookup ? k : 0];
    const int t = e / N;
    unsigned short w = 1;
    if (t < T - 1) {
        const double t_i = theta_las[e];
        const double *nx = theta_las + (size_t)(t + 1) * N;
        int lo = N, hi = -1, cnt = 0;
        for (int j = 0; j < N; ++j) {
            const double d = nx[j] - t_i;
            if (d < max_dtheta && d > -max_dtheta) {
                lo = min(lo, j);
                hi = j;
                ++cnt;
            }
        }
        if (cnt > 0 && cnt == hi - lo + 1) w = (unsigned short)(lo | hi << 8);
    }
    win[e] = w;
}

// One workgroup walks the rays backwards.  Rays are numbered q = T-1-t (q = 0 is the last ray).
//   * 8 (N <= 128) or 16 lanes per point i of the current ray, DPP butterflies inside the lane
//     group, 128 (64) points per round.  A point with a window (lc_prepare_kernel) reads only
//     v_unc' over the window for the max -- max_j (r + v'[j]) = r + max_j v'[j], rounding is
//     monotonic -- and computes laser costs only for the j that attain it (usually one).  A
//     point without one, or whose successors are all -inf (the reference then ties over every j,
//     allowed or not, policy.py:368), takes the general scan: every 8th / 16th point j of the
//     next ray in registers, max of q_unc, arg-min of the cost among the maximisers.
//     One CU does the whole walk: its LDS and VALU throughput bound a step, which is why the
//     windowed path matters (34 of 80 successors on average on the KITTI-shaped graph).
//   * laser angles and r_unc of the next LCP_CHUNK rays are prefetched into registers while the
//     current chunk is computed and dropped into the other half of an LDS double buffer at the
//     chunk boundary: no global load sits on the serial path.
//   * next_idx lives in LDS as bytes (N <= 256) when T*N fits, so the forward walk over T
//     dependent look-ups costs LDS latency, not L2 latency; otherwise it goes through `next_g`.
// Dynamic LDS: th[2][C][N] f64 | vl[2][N] f64 | ru[2][C][N] f32 | vu[2][N] f32 |
//              wn[2][C][N] u16 | nx[T*N] u8.
#define LCP_CHUNK 8
#define LCP_PF ((LCP_CHUNK * LCP_MAXN + LCP_THREADS - 1) / LCP_THREADS)

// Butterflies inside a group of 8 or 16 lanes on the DPP path (a VALU move each; a
// ds_bpermute round trip per step would dominate the serial loop): row_mirror pairs lane l with
// 15 - l, row_half_mirror with 7 - l, the two quad permutations with l ^ 1 and l ^ 2.
#define LCP_DPP_ROW_MIRROR 0x140
#define LCP_DPP_HALF_MIRROR 0x141
#define LCP_DPP_XOR1 0xB1
#define LCP_DPP_XOR2 0x4E

template <int CTRL> __device__ __forceinline__ int lcp_dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL> __device__ __forceinline__ float lcp_dpp(float v)
{
    return __int_as_float(lcp_dpp<CTRL>(__float_as_int(v)));
}
template <int CTRL> __device__ __forceinline__ MinIdx lcp_dpp(MinIdx m)
{
    const int lo = lcp_dpp<CTRL>(__double2loint(m.v)), hi = lcp_dpp<CTRL>(__double2hiint(m.v));
    return MinIdx{__hiloint2double(hi, lo), lcp_dpp<CTRL>(m.i)};
}

// LPP lanes per point, NS points j per lane: LPP * NS >= N.
template <int LPP, int NS>
__global__ void __launch_bounds__(LCP_THREADS)
lc_dp_kernel(const float *__restrict__ r_unc_g, const unsigned short *__restrict__ win_g,
             const double *__restrict__ theta_las, int T, int N, double max_dtheta, int nx_in_lds, int *__restrict__ next_g /*[T][N]*/,
             int *__restrict__ choice /*[T]*/, float *__restrict__ v_unc_out /*[N]*/,
             double *__restrict__ v_las_out /*[N]*/)
{
#pragma clang fp contract(off) // d * d + v_las' is a rounded product, then a rounded sum
    extern __shared__ __align__(16) unsigned char lcp_lds[];
    constexpr int C = LCP_CHUNK;
    double *th = (double *)lcp_lds;          // [2][C][N]
    double *vl = th + 2 * C * N;             // [2][N]
    float *ru = (float *)(vl + 2 * N);       // [2][C][N]
    float *vu = ru + 2 * C * N;              // [2][N]
    unsigned short *wn = (unsigned short *)(vu + 2 * N); // [2][C][N]
    unsigned char *nx = (unsigned char *)(wn + 2 * C * N);
    const int tid = threadIdx.x, grp = tid / LPP, ll = tid % LPP;
    const int CN = C * N;
    const int nchunks = (T + C - 1) / C;

    // chunk 0 straight into the buffer; ray q lives at th[((q / C) & 1) * CN + (q % C) * N + j]
    for (int e = tid; e < CN; e += LCP_THREADS) {
        const int q = e / N, j = e - q * N;
        if (q < T) {
            th[e] = theta_las[(size_t)(T - 1 - q) * N + j];
            ru[e] = r_unc_g[(size_t)(T - 1 - q) * N + j];
            wn[e] = win_g[(size_t)(T - 1 - q) * N + j];
        }
    }
    __syncthreads();
    for (int j = tid; j < N; j += LCP_THREADS) { // the last ray (policy.py:331-336)
        vu[j] = ru[j];
        vl[j] = 0.0;
    }
    __syncthreads();
    int cur = 0; // vu[cur * N ..], vl[cur * N ..] belong to ray q - 1
    for (int c = 0; c < nchunks; ++c) {
        double pf_th[LCP_PF];
        float pf_ru[LCP_PF];
        unsigned short pf_wn[LCP_PF];
        const bool more = c + 1 < nchunks;
        if (more) {
#pragma unroll
            for (int u = 0; u < LCP_PF; ++u) {
                const int e = tid + u * LCP_THREADS;
                const int qq = e / N, j = e - qq * N;
                const int q = (c + 1) * C + qq;
                pf_th[u] = 0.0;
                pf_ru[u] = 0.f;
                pf_wn[u] = 1;
                if (e < CN && q < T) {
                    pf_th[u] = theta_las[(size_t)(T - 1 - q) * N + j];
                    pf_ru[u] = r_unc_g[(size_t)(T - 1 - q) * N + j];
                    pf_wn[u] = win_g[(size_t)(T - 1 - q) * N + j];
                }
            }
        }
        const int q_lo = c * C > 1 ? c * C : 1, q_hi = (c + 1) * C < T ? (c + 1) * C : T;
        for (int q = q_lo; q < q_hi; ++q) {
            const double *th_i = th + ((q / C) & 1) * CN + (q % C) * N;             // ray t
            const float *ru_i = ru + ((q / C) & 1) * CN + (q % C) * N;
            const unsigned short *wn_i = wn + ((q / C) & 1) * CN + (q % C) * N;
            const double *th_j = th + (((q - 1) / C) & 1) * CN + ((q - 1) % C) * N; // ray t + 1
            const float *vu_c = vu + cur * N;
            const double *vl_c = vl + cur * N;
            const int nxt = cur ^ 1;
#ifdef LCP_TIMING // timing build only (tools/bench_lc.py --timing): shader and wall clocks per step
            if (tid == 0 && nx_in_lds) {
                next_g[8 * q] = (int)clock64();
                next_g[8 * q + 1] = (int)wall_clock64();
            }
#define LCP_STAMP(k) if (tid == 0 && nx_in_lds) next_g[8 * q + k] = (int)clock64();
#else
#define LCP_STAMP(k)
#endif
            for (int i = grp; i < N; i += LCP_THREADS / LPP) {
                const float r = ru_i[i];
                const double t_i = th_i[i];
                const int w = wn_i[i], lo = w & 255, hi = w >> 8;
                LCP_STAMP(2 + (w & 0))
                float best = -INFINITY;
                MinIdx m{INFINITY, 0x7FFFFFFF};
                bool general = lo > hi;
                if (!general) {
                    constexpr int W = 64 / LPP < 8 ? 8 : 64 / LPP; // window slots per lane in registers
                    const bool one_block = hi - lo < W * LPP;
                    float v[W];
                    float vmax = -INFINITY;
                    if (one_block) {
#pragma unroll
                        for (int u = 0; u < W; ++u) {
                            const int j = lo + ll + u * LPP;
                            v[u] = j <= hi ? vu_c[j] : -INFINITY;
                            vmax = fmaxf(vmax, v[u]);
                        }
                    } else {
                        for (int j = lo + ll; j <= hi; j += LPP) vmax = fmaxf(vmax, vu_c[j]);
                    }
                    if (LPP == 16) vmax = fmaxf(vmax, lcp_dpp<LCP_DPP_ROW_MIRROR>(vmax));
                    if (LPP >= 8) vmax = fmaxf(vmax, lcp_dpp<LCP_DPP_HALF_MIRROR>(vmax));
                    vmax = fmaxf(vmax, lcp_dpp<LCP_DPP_XOR2>(vmax));
                    vmax = fmaxf(vmax, lcp_dpp<LCP_DPP_XOR1>(vmax));
                    best = r + vmax; // policy.py:354-360
                    LCP_STAMP(3 + (__float_as_int(best) & 0))
                    if (!(best > -INFINITY)) {
                        general = true; // every successor ties at -inf (or NaN): the full scan
                    } else if (one_block) {
                        unsigned cand = 0;
#pragma unroll
                        for (int u = 0; u < W; ++u)
                            if (lo + ll + u * LPP <= hi && r + v[u] == best) cand |= 1u << u;
                        while (cand) { // ascending j: the first arg-min wins (policy.py:374)
                            const int u = __ffs(cand) - 1;
                            cand &= cand - 1;
                            const int j = lo + ll + u * LPP;
                            const double d = th_j[j] - t_i;
                            m = lcp_min(m, MinIdx{d * d + vl_c[j], j}); // :366
                        }
                    } else {
                        for (int j = lo + ll; j <= hi; j += LPP)
                            if (r + vu_c[j] == best) {
                                const double d = th_j[j] - t_i;
                                m = lcp_min(m, MinIdx{d * d + vl_c[j], j});
                            }
                    }
                }
                if (general) {
                    float qv[NS];
                    double cost[NS];
                    best = -INFINITY; // v_unc = max over the allowed j (policy.py:354-360)
#pragma unroll
                    for (int u = 0; u < NS; ++u) {
                        const int j = ll + u * LPP;
                        qv[u] = -INFINITY;
                        cost[u] = INFINITY;
                        if (j < N) {
                            const double d = th_j[j] - t_i;                    // :266
                            const bool ok = d < max_dtheta && d > -max_dtheta; // :267-268
                            if (ok) qv[u] = r + vu_c[j];
                            cost[u] = d * d + vl_c[j];                         // :366
                            best = fmaxf(best, qv[u]);
                        }
                    }
                    if (LPP == 16) best = fmaxf(best, lcp_dpp<LCP_DPP_ROW_MIRROR>(best));
                    if (LPP >= 8) best = fmaxf(best, lcp_dpp<LCP_DPP_HALF_MIRROR>(best));
                    best = fmaxf(best, lcp_dpp<LCP_DPP_XOR2>(best));
                    best = fmaxf(best, lcp_dpp<LCP_DPP_XOR1>(best));
                    // among the maximisers (every j when all are -inf, :368) the smallest laser
                    // cost, first index (:374)
                    m = MinIdx{INFINITY, 0x7FFFFFFF};
#pragma unroll
                    for (int u = 0; u < NS; ++u) {
                        const int j = ll + u * LPP;
                        if (j < N && qv[u] == best) m = lcp_min(m, MinIdx{cost[u], j});
                    }
                }
                LCP_STAMP(4 + (m.i & 0))
                if (LPP == 16) m = lcp_min(m, lcp_dpp<LCP_DPP_ROW_MIRROR>(m));
                if (LPP >= 8) m = lcp_min(m, lcp_dpp<LCP_DPP_HALF_MIRROR>(m));
                m = lcp_min(m, lcp_dpp<LCP_DPP_XOR2>(m));
                m = lcp_min(m, lcp_dpp<LCP_DPP_XOR1>(m));
                LCP_STAMP(5 + (m.i & 0))
                if (ll == 0) {
                    vu[nxt * N + i] = best;
                    vl[nxt * N + i] = m.v;
                    if (nx_in_lds) nx[(size_t)(T - 1 - q) * N + i] = (unsigned char)m.i;
                    else next_g[(size_t)(T - 1 - q) * N + i] = m.i;
                }
            }
            LCP_STAMP(6)
            __syncthreads();
            cur = nxt;
        }
        if (more) { // chunk c - 1's half is free: the steps above only looked one ray back
#pragma unroll
            for (int u = 0; u < LCP_PF; ++u) {
                const int e = tid + u * LCP_THREADS;
                if (e < CN) {
                    th[((c + 1) & 1) * CN + e] = pf_th[u];
                    ru[((c + 1) & 1) * CN + e] = pf_ru[u];
                    wn[((c + 1) & 1) * CN + e] = pf_wn[u];
                }
            }
            __syncthreads();
        }
    }
    // forward walk (policy.py:399-415)
    if (!nx_in_lds) __threadfence();
    __syncthreads();
    for (int j = tid; j < N; j += LCP_THREADS) {
        v_unc_out[j] = vu[cur * N + j];
        v_las_out[j] = vl[cur * N + j];
    }
    if (tid == 0) {
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) mx = fmaxf(mx, vu[cur * N + j]);
        MinIdx m{INFINITY, 0x7FFFFFFF};
        for (int j = 0; j < N; ++j)
            if (vu[cur * N + j] == mx) m = lcp_min(m, MinIdx{vl[cur * N + j], j});
        int cpt = m.i;
        for (int t = 0; t < T; ++t) {
            choice[t] = cpt;
            if (t < T - 1)
                cpt = nx_in_lds ? (int)nx[(size_t)t * N + cpt]
                                : __hip_atomic_load(&next_g[(size_t)t * N + cpt], __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

extern "C" int lcf_lc_entropy_lookup(const float *cls, int K, int Y, int X,
                                     const unsigned char *anchors_mask, int apply_sigmoid,
                                     float *ulookup, void *stream)
{
    if (!cls || !ulookup || K <= 0 || Y <= 0 || X <= 0) return LCF_ERR_ARG;
    if ((long long)Y * X >= 0x7FFFFFFFll) return LCF_ERR_RANGE;
    const int n = Y * X;
    lc_entropy_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(
        cls, K, Y, X, anchors_mask, apply_sigmoid, ulookup);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" size_t lcf_lc_dp_workspace_bytes(int T, int N)
{
    if (T <= 0 || N <= 0) return 0;
    // next_idx [T*N] i32 | v_unc [N] f32 | v_las [N] f64 | r_unc [T*N] f32 | win [T*N] u16
    return 2 * lcf_align_up((size_t)T * N * 4, 256) + lcf_align_up((size_t)N * 4, 256) +
           lcf_align_up((size_t)N * 8, 256) + lcf_align_up((size_t)T * N * 2, 256);
}

static size_t lcp_lds_bytes(int N) // without the next_idx bytes
{
    return (size_t)(2 * LCP_CHUNK * N + 2 * N) * (8 + 4) + (size_t)2 * LCP_CHUNK * N * 2;
}

extern "C" int lcf_lc_dp_design(const float *ulookup, int n_lookup, const int *node_k,
                                const double *theta_las, int T, int N, double max_dtheta,
                                int *choice, void *ws, size_t ws_bytes, void *stream)
{
    if (!ulookup || n_lookup <= 0 || !node_k || !theta_las || !choice || !ws || T <= 0 || N <= 0)
        return LCF_ERR_ARG;
    if (N > LCP_MAXN) return LCF_ERR_UNSUPPORTED;
    if ((long long)T * N >= (1ll << 29)) return LCF_ERR_RANGE;
    if (ws_bytes < lcf_lc_dp_workspace_bytes(T, N)) return LCF_ERR_WORKSPACE;
    char *p = (char *)ws;
    int *next_idx = (int *)p;
    p += lcf_align_up((size_t)T * N * 4, 256);
    float *vu = (float *)p; // value functions of ray 0, kept for inspection / tests
    p += lcf_align_up((size_t)N * 4, 256);
    double *vl = (double *)p;
    p += lcf_align_up((size_t)N * 8, 256);
    float *r_unc = (float *)p;
    p += lcf_align_up((size_t)T * N * 4, 256);
    unsigned short *win = (unsigned short *)p;
    const size_t base = lcp_lds_bytes(N), lds_max = 160 * 1024;
    const int nx_in_lds = base + (size_t)T * N <= lds_max;
    const size_t lds = nx_in_lds ? base + (size_t)T * N : base;
    auto kern = N <= 80 ? lc_dp_kernel<8, 10> : N <= 128 ? lc_dp_kernel<8, 16>
                                                          : lc_dp_kernel<16, 16>;
    const int which = N <= 80 ? 0 : N <= 128 ? 1 : 2;
    {
        static size_t lds_set[3] = {0, 0, 0};
        static std::mutex lds_mu;
        std::lock_guard<std::mutex> lock(lds_mu);
        if (lds > 64 * 1024 && lds > lds_set[which]) {
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return LCF_ERR_LAUNCH;
            lds_set[which] = lds;
        }
    }
    const int n = T * N;
    lc_prepare_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(
        ulookup, n_lookup, node_k, theta_las, T, N, max_dtheta, r_unc, win);
    LCF_CHECK_LAUNCH();
    kern<<<1, LCP_THREADS, lds, (hipStream_t)stream>>>(
        r_unc, win, theta_las, T, N, max_dtheta, nx_in_lds, next_idx, choice, vu, vl);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}
