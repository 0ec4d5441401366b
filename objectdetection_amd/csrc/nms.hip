// nms.hip -- rotated / axis-aligned IoU and non-maximum suppression on the device
// (second/core/non_max_suppression/nms_gpu.py:404-508 rotate_nms_kernel /
// rotate_iou_kernel, :70-163 nms_kernel / nms_postprocess; called from
// second/pytorch/core/box_torch_ops.py:473-515 at predict time, where the reference goes through
// the host).  Geometry as in the reference (nms_gpu.py:166-401): a box (cx, cy, w, h, angle) is
// turned into four corners, the intersection polygon of two boxes is the set of corners of either
// that lie inside the other plus the pairwise edge intersections (at most 8 + 16 candidates, 8
// vertices), ordered around their centroid and summed as a triangle fan.
//
// One wave per 64 x 64 block of pairs: lane = row box, the 64 column boxes (corners precomputed
// once per block) sit in LDS; the suppression matrix is one 64-bit word per (row, column block),
// only blocks on or above the diagonal are computed.  The sequential pass over the sorted boxes
// (nms_postprocess, a host loop in the reference) runs in one wave on the device, so neither the
// matrix nor the scores leave HBM.
#include "common.h"

struct NmsBox {
    float c[8]; // corners x0,y0 .. x3,y3 (clockwise)
    float area;
};

__device__ __forceinline__ NmsBox nms_corners(const float *b)
{
    // nms_gpu.py:353-376
    NmsBox r;
    const float ac = cosf(b[4]), as = sinf(b[4]);
    const float hx = b[2] * 0.5f, hy = b[3] * 0.5f;
    const float xs[4] = {-hx, -hx, hx, hx};
    const float ys[4] = {-hy, hy, hy, -hy};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r.c[2 * i] = ac * xs[i] + as * ys[i] + b[0];
        r.c[2 * i + 1] = -as * xs[i] + ac * ys[i] + b[1];
    }
    r.area = b[2] * b[3];
    return r;
}

// a box given by its four corners (x0,y0 .. x3,y3, clockwise): spconv's rbbox_iou /
// rotate_non_max_suppression_cpu take boxes in this form (nms_cpu.py:21-32)
__device__ __forceinline__ NmsBox nms_from_corners(const float *c)
{
    NmsBox r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.c[i] = c[i];
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    r.area = sqrtf(ab0 * ab0 + ab1 * ab1) * sqrtf(ad0 * ad0 + ad1 * ad1);
    return r;
}

__device__ __forceinline__ NmsBox nms_load(const float *b, int corners)
{
    return corners ? nms_from_corners(b) : nms_corners(b);
}

__device__ __forceinline__ bool nms_inside(float px, float py, const float *q)
{
    // nms_gpu.py:308-325: projections on the edges AB and AD of the rectangle, eps = -1e-6
    const float ab0 = q[2] - q[0], ab1 = q[3] - q[1];
    const float ad0 = q[6] - q[0], ad1 = q[7] - q[1];
    const float ap0 = px - q[0], ap1 = py - q[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    const float eps = -1e-6f;
    return abab - abap >= eps && abap >= eps && adad - adap >= eps && adap >= eps;
}

// area of the intersection of two rectangles given by their corners (nms_gpu.py:329-393)
__device__ float nms_inter_area(const float *p, const float *q)
{
    float px[24], py[24]; // 8 corners + 16 edge crossings at most (degenerate overlaps)
    int n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (nms_inside(p[2 * i], p[2 * i + 1], q)) {
            px[n] = p[2 * i];
            py[n] = p[2 * i + 1];
            ++n;
        }
        if (nms_inside(q[2 * i], q[2 * i + 1], p)) {
            px[n] = q[2 * i];
            py[n] = q[2 * i + 1];
            ++n;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float ax = p[2 * i], ay = p[2 * i + 1];
        const float bx = p[2 * ((i + 1) & 3)], by = p[2 * ((i + 1) & 3) + 1];
        const float ba0 = bx - ax, ba1 = by - ay;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // nms_gpu.py:225-264: orientation tests, then Cramer's rule
            const float cx = q[2 * j], cy = q[2 * j + 1];
            const float dx = q[2 * ((j + 1) & 3)], dy = q[2 * ((j + 1) & 3) + 1];
            const float da0 = dx - ax, ca0 = cx - ax, da1 = dy - ay, ca1 = cy - ay;
            const bool acd = da1 * ca0 > ca1 * da0;
            const bool bcd = (dy - by) * (cx - bx) > (cy - by) * (dx - bx);
            if (acd != bcd) {
                const bool abc = ca1 * ba0 > ba1 * ca0;
                const bool abd = da1 * ba0 > ba1 * da0;
                if (abc != abd) {
                    const float dc0 = dx - cx, dc1 = dy - cy;
                    const float abba = ax * by - bx * ay;
                    const float cddc = cx * dy - dx * cy;
                    const float dh = ba1 * dc0 - ba0 * dc1;
                    px[n] = (abba * dc0 - ba0 * cddc) / dh;
                    py[n] = (abba * dc1 - ba1 * cddc) / dh;
                    ++n;
                }
            }
        }
    }
    if (n < 3) return 0.f;
    // order around the centroid (nms_gpu.py:183-219): key = unit-vector x, mirrored for y < 0
    float mx = 0.f, my = 0.f;
    for (int i = 0; i < n; ++i) {
        mx += px[i];
        my += py[i];
    }
    mx /= n;
    my /= n;
    float key[24];
    for (int i = 0; i < n; ++i) {
        float vx = px[i] - mx, vy = py[i] - my;
        const float d = sqrtf(vx * vx + vy * vy);
        vx /= d;
        vy /= d;
        key[i] = vy < 0.f ? -2.f - vx : vx;
    }
    for (int i = 1; i < n; ++i) { // insertion sort, ascending key
        if (key[i - 1] > key[i]) {
            const float t = key[i], tx = px[i], ty = py[i];
            int j = i;
            while (j > 0 && key[j - 1] > t) {
                key[j] = key[j - 1];
                px[j] = px[j - 1];
                py[j] = py[j - 1];
                --j;
            }
            key[j] = t;
            px[j] = tx;
            py[j] = ty;
        }
    }
    float area = 0.f; // triangle fan from vertex 0 (nms_gpu.py:166-179)
    for (int i = 0; i < n - 2; ++i)
        area += fabsf((px[0] - px[i + 2]) * (py[i + 1] - py[i + 2]) -
                      (py[0] - py[i + 2]) * (px[i + 1] - px[i + 2])) * 0.5f;
    return area;
}

__device__ __forceinline__ float nms_ratio(float inter, float a1, float a2, int criterion)
{
    // nms_gpu.py:549-561 (a1: query box, a2: box)
    if (criterion == -1) return inter / (a1 + a2 - inter);
    if (criterion == 0) return inter / a1;
    if (criterion == 1) return inter / a2;
    return inter;
}

// ---- iou[n, k] of rotated boxes [n,5] against query boxes [k,5]
__global__ void __launch_bounds__(64)
nms_rotate_iou_kernel(const float *__restrict__ boxes, int n, const float *__restrict__ qboxes,
                      int k, int corners, int criterion, float *__restrict__ iou)
{
    const int bs = corners ? 8 : 5;
    __shared__ NmsBox qs[64];
    const int tx = threadIdx.x;
    const int row = blockIdx.x * 64 + tx;
    const int col0 = blockIdx.y * 64;
    const int ncol = min(64, k - col0);
    if (tx < ncol) qs[tx] = nms_load(qboxes + (size_t)(col0 + tx) * bs, corners);
    __syncthreads();
    if (row >= n) return;
    const NmsBox b = nms_load(boxes + (size_t)row * bs, corners);
    for (int j = 0; j < ncol; ++j) {
        const float inter = nms_inter_area(qs[j].c, b.c);
        iou[(size_t)row * k + col0 + j] = nms_ratio(inter, qs[j].area, b.area, criterion);
    }
}

// ---- suppression matrix of boxes sorted by descending score.
// dets: rows of `stride` floats; mode 0: (x1, y1, x2, y2, ...) axis-aligned with the +eps pixel
// convention (nms_gpu.py:21-32 uses eps = 1), mode 1: (cx, cy, w, h, angle, ...) rotated,
// mode 2: rotated, given as 8 corner coordinates.
// mask[i, jb] bit j: box 64*jb + j (after i in the order) overlaps box i by more than thresh
// (inclusive: at least thresh).
__global__ void __launch_bounds__(64)
nms_mask_kernel(const float *__restrict__ dets, int n, int stride, int mode, float thresh,
                float eps, int inclusive, unsigned long long *__restrict__ mask)
{
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return; // only pairs (i, j > i) are ever read
    __shared__ NmsBox cs[64];
    __shared__ float cr[64][4];
    const int tx = threadIdx.x;
    const int ncol = min(64, n - cb * 64);
    const int col_blocks = (n + 63) / 64;
    if (tx < ncol) {
        const float *d = dets + (size_t)(cb * 64 + tx) * stride;
        if (mode != 0) {
            cs[tx] = nms_load(d, mode == 2);
        } else {
            cr[tx][0] = d[0];
            cr[tx][1] = d[1];
            cr[tx][2] = d[2];
            cr[tx][3] = d[3];
        }
    }
    __syncthreads();
    const int row = rb * 64 + tx;
    if (row >= n) return;
    const float *d = dets + (size_t)row * stride;
    unsigned long long t = 0ull;
    const int start = rb == cb ? tx + 1 : 0;
    if (mode != 0) {
        const NmsBox b = nms_load(d, mode == 2);
        for (int j = start; j < ncol; ++j) {
            const float inter = nms_inter_area(b.c, cs[j].c);
            const float ov = inter / (b.area + cs[j].area - inter);
            if (inclusive ? ov >= thresh : ov > thresh) t |= 1ull << j;
        }
    } else {
        const float x1 = d[0], y1 = d[1], x2 = d[2], y2 = d[3];
        const float sa = (x2 - x1 + eps) * (y2 - y1 + eps);
        for (int j = start; j < ncol; ++j) {
            const float w = fmaxf(fminf(x2, cr[j][2]) - fmaxf(x1, cr[j][0]) + eps, 0.f);
            const float h = fmaxf(fminf(y2, cr[j][3]) - fmaxf(y1, cr[j][1]) + eps, 0.f);
            const float inter = w * h;
            const float sb = (cr[j][2] - cr[j][0] + eps) * (cr[j][3] - cr[j][1] + eps);
            const float ov = inter / (sa + sb - inter);
            if (inclusive ? ov >= thresh : ov > thresh) t |= 1ull << j;
        }
    }
    mask[(size_t)row * col_blocks + cb] = t;
}

// ---- the sequential pass (nms_postprocess, nms_gpu.py:109-126) in one wave: walk the boxes in
// score order, keep a box unless an earlier kept box removed it, OR its row into the removed set
#define NMS_MAX_BLOCKS 1024 /* 65536 boxes */
__global__ void __launch_bounds__(64)
nms_select_kernel(const unsigned long long *__restrict__ mask, int n, int max_keep,
                  int *__restrict__ keep, int *__restrict__ num_keep)
{
    __shared__ unsigned long long remv[NMS_MAX_BLOCKS];
    const int lane = threadIdx.x;
    const int col_blocks = (n + 63) / 64;
    for (int j = lane; j < col_blocks; j += 64) remv[j] = 0ull;
    __syncthreads();
    int cnt = 0;
    for (int i = 0; i < n && cnt < max_keep; ++i) {
        const int nb = i >> 6;
        const unsigned long long w = remv[nb]; // wave-uniform read
        if (!((w >> (i & 63)) & 1ull)) {
            if (lane == 0) keep[cnt] = i;
            ++cnt;
            for (int j = nb + lane; j < col_blocks; j += 64)
                remv[j] |= mask[(size_t)i * col_blocks + j];
            __syncthreads(); // one wave: orders the LDS updates before the next test
        }
    }
    if (lane == 0) *num_keep = cnt;
}

extern "C" int lcf_rotate_iou(const float *boxes, int n, const float *qboxes, int k, int corners,
                              int criterion, float *iou, void *stream)
{
    if (n < 0 || k < 0 || criterion < -1 || criterion > 2 || (corners != 0 && corners != 1))
        return LCF_ERR_ARG;
    if (n == 0 || k == 0) return LCF_OK;
    if (!boxes || !qboxes || !iou) return LCF_ERR_ARG;
    dim3 grid((n + 63) / 64, (k + 63) / 64);
    nms_rotate_iou_kernel<<<grid, 64, 0, (hipStream_t)stream>>>(boxes, n, qboxes, k, corners,
                                                                criterion, iou);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}

extern "C" size_t lcf_nms_workspace_bytes(int n)
{
    if (n <= 0) return 0;
    return (size_t)n * ((n + 63) / 64) * sizeof(unsigned long long);
}

extern "C" int lcf_nms(const float *dets, int n, int stride, int mode, float thresh, float eps,
                       int inclusive, int max_keep, int *keep, int *num_keep, void *ws,
                       size_t ws_bytes, void *stream)
{
    if (n < 0 || !num_keep || mode < 0 || mode > 2 || stride < (mode == 2 ? 8 : mode ? 5 : 4))
        return LCF_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return lcf_fill32(num_keep, 0u, 4, st);
    if (!dets || !keep || !ws) return LCF_ERR_ARG;
    const int col_blocks = (n + 63) / 64;
    if (col_blocks > NMS_MAX_BLOCKS) return LCF_ERR_RANGE;
    if (ws_bytes < lcf_nms_workspace_bytes(n)) return LCF_ERR_WORKSPACE;
    if (max_keep <= 0 || max_keep > n) max_keep = n;
    unsigned long long *mask = (unsigned long long *)ws;
    dim3 grid(col_blocks, col_blocks);
    nms_mask_kernel<<<grid, 64, 0, st>>>(dets, n, stride, mode, thresh, eps, inclusive, mask);
    LCF_CHECK_LAUNCH();
    nms_select_kernel<<<1, 64, 0, st>>>(mask, n, max_keep, keep, num_keep);
    LCF_CHECK_LAUNCH();
    return LCF_OK;
}
