/*
 * lcf_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, sequential, fp32) of
 * the SECOND sparse front end used by CMU-Light-Curtains/ObjectDetection.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (objectdetection_amd/csrc, liblcf.so) never links, loads or calls it.
 *
 * What each function follows (paths relative to /root/reference):
 *   orc_voxelize        second/utils/simplevis.py:23-50 (the in-tree twin of the missing
 *                       second/core/point_cloud/point_cloud_ops.py points_to_voxel kernel: same
 *                       coordinate maths, first-come voxel ids, `max_voxels` break) and the call
 *                       contracts at second/data/preprocess.py:626-639,
 *                       second/kittiviewer/viewer.py:389-396.
 *   orc_rulebook_*      spconv 1.x get_indice_pairs CPU path (getValidOutPos / getIndicePairsConv /
 *                       getIndicePairsSubM).  spconv is a third-party dependency of the reference,
 *                       NOT vendored and NOT version pinned (README.md:29); this is a restatement
 *                       of its published algorithm, anchored on the call sites
 *                       second/pytorch/models/middle.py:145-192.
 *   orc_indice_conv_*   spconv 1.x indice_conv forward / backward (gather -> mm -> scatter-add per
 *                       kernel offset, offsets ascending), weights [kD,kH,kW,Cin,Cout].
 *   orc_dense           SparseConvTensor.dense(): scatter to [B,D,H,W,C] then permute to
 *                       [B,C,D,H,W] (middle.py:206).
 *
 * PARITY PIN STATUS: the reference holds no tests, fixtures or golden vectors for this path
 *  and spconv/numba are absent, so the voxelizer / rulebook / indice_conv parts of
 * this oracle are "parity unpinned" against the original binaries.  They are pinned instead by
 * hand-derived known-answer tests and by equivalence with torch.nn.functional.conv3d
 * (tests/test_oracle_*.py).  The VFE oracle (oracle/vfe_oracle.py) IS pinned against vectors
 * generated from the real reference file (tests/golden/vfe_*.npz).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_KVOL 343

/* ------------------------------------------------------------------ voxelizer */

/* cap_mode 0: `break` when a new voxel would exceed max_voxels (numba twin, simplevis.py:47-48)
 * cap_mode 1: `continue` (spconv C++ points_to_voxel_3d_np: late points still join old voxels) */
int orc_voxelize(const float *pts, int n, int f, const float vs[3], const float range[6],
                 int max_pts, int max_vox, int cap_mode, float *voxels, int *coors,
                 int *num_pts, int *voxel_num)
{
    int grid[3];
    for (int j = 0; j < 3; ++j) {
        /* grid_size = np.round((hi - lo) / vs) in float32 (simplevis.py:26-29) */
        float g = (range[3 + j] - range[j]) / vs[j];
        grid[j] = (int)rintf(g);
        if (grid[j] <= 0) return -1;
    }
    size_t cells = (size_t)grid[0] * grid[1] * grid[2];
    int *lut = (int *)malloc(cells * sizeof(int));
    if (!lut) return -2;
    memset(lut, 0xff, cells * sizeof(int)); /* -1 */
    memset(voxels, 0, (size_t)max_vox * max_pts * f * sizeof(float));
    memset(coors, 0, (size_t)max_vox * 3 * sizeof(int));
    memset(num_pts, 0, (size_t)max_vox * sizeof(int));
    int nvox = 0;
    for (int i = 0; i < n; ++i) {
        int c[3];
        int failed = 0;
        for (int j = 0; j < 3; ++j) {
            volatile float d = pts[(size_t)i * f + j] - range[j]; /* fp32 subtract */
            volatile float q = d / vs[j];                         /* fp32 true divide */
            float fl = floorf(q);
            if (!(fl >= 0.0f) || fl >= (float)grid[j]) { failed = 1; break; } /* NaN dropped */
            c[2 - j] = (int)fl; /* zyx */
        }
        if (failed) continue;
        size_t cell = ((size_t)c[0] * grid[1] + c[1]) * grid[0] + c[2];
        int vid = lut[cell];
        if (vid == -1) {
            if (nvox >= max_vox) {
                if (cap_mode == 0) break;
                continue;
            }
            vid = nvox++;
            lut[cell] = vid;
            coors[vid * 3 + 0] = c[0];
            coors[vid * 3 + 1] = c[1];
            coors[vid * 3 + 2] = c[2];
        }
        int m = num_pts[vid];
        if (m < max_pts) {
            memcpy(voxels + ((size_t)vid * max_pts + m) * f, pts + (size_t)i * f,
                   f * sizeof(float));
            num_pts[vid] = m + 1;
        }
    }
    free(lut);
    *voxel_num = nvox;
    return 0;
}

/* ------------------------------------------------------------------ rulebooks */

static int out_size(int in, int k, int s, int p, int d)
{
    /* spconv get_conv_output_size: (in + 2p - d(k-1) - 1) / s + 1 */
    return (in + 2 * p - d * (k - 1) - 1) / s + 1;
}

void orc_conv_out_shape(const int shape[3], const int ksize[3], const int stride[3],
                        const int pad[3], const int dil[3], int out[3])
{
    for (int i = 0; i < 3; ++i) out[i] = out_size(shape[i], ksize[i], stride[i], pad[i], dil[i]);
}

/* getValidOutPos: enumerate output positions covered by one input position.  Order: from the
 * upper bound downward in every dim, last dim fastest; truncating C division for the bounds.
 * out[] receives (z,y,x,offset) quadruples, returns the number of valid ones. */
static int valid_out_pos(const int *in, const int *ks, const int *st, const int *pd,
                         const int *dl, const int *oshape, int *out)
{
    int lo[3], up[3], cs[3], cnt[3] = {0, 0, 0};
    int npts = 1, pc = 0;
    for (int i = 0; i < 3; ++i) {
        lo[i] = (in[i] - (ks[i] - 1) * dl[i] - 1 + st[i] + pd[i]) / st[i];
        up[i] = (in[i] + pd[i]) / st[i];
    }
    for (int i = 0; i < 3; ++i) {
        cs[i] = (up[i] - lo[i]) / dl[i] + 1;
        npts *= cs[i];
    }
    for (int i = 0; i < npts; ++i) {
        int valid = 1, m = 1, off = 0;
        for (int j = 2; j >= 0; --j) {
            int val = up[j] - cnt[j] * dl[j];
            out[pc * 4 + j] = val;
            if (val < 0 || val > oshape[j] - 1) valid = 0;
            off += m * (in[j] - val * st[j] + pd[j]) / dl[j];
            m *= ks[j];
        }
        out[pc * 4 + 3] = off;
        if (valid) ++pc;
        cnt[2] += 1;
        for (int c = 2; c >= 0; --c) {
            if (cnt[c] == cs[c] && c > 0) {
                cnt[c - 1] += 1;
                cnt[c] = 0;
            }
        }
    }
    return pc;
}

/* Strided / ordinary sparse conv rulebook, CPU discovery order (canonical).
 * coors [k_in,4] (b,z,y,x); pairs [2,kvol,k_in] filled with -1; pair_num [kvol];
 * out_coors capacity k_in*kvol rows. */
int orc_rulebook_conv(const int *coors, int k_in, int batch, const int shape[3],
                      const int ksize[3], const int stride[3], const int pad[3],
                      const int dil[3], int *out_coors, int *k_out, int *pairs, int *pair_num)
{
    int oshape[3];
    orc_conv_out_shape(shape, ksize, stride, pad, dil, oshape);
    int kvol = ksize[0] * ksize[1] * ksize[2];
    if (kvol > ORC_MAX_KVOL) return -1;
    size_t svol = (size_t)oshape[0] * oshape[1] * oshape[2];
    int *grid = (int *)malloc(svol * batch * sizeof(int));
    if (!grid) return -2;
    memset(grid, 0xff, svol * batch * sizeof(int));
    for (size_t i = 0; i < (size_t)2 * kvol * k_in; ++i) pairs[i] = -1;
    memset(pair_num, 0, kvol * sizeof(int));
    int *vp = (int *)malloc((size_t)kvol * 4 * sizeof(int) + 16);
    int nact = 0;
    for (int j = 0; j < k_in; ++j) {
        int b = coors[j * 4];
        int nv = valid_out_pos(coors + j * 4 + 1, ksize, stride, pad, dil, oshape, vp);
        for (int i = 0; i < nv; ++i) {
            const int *p = vp + i * 4;
            int off = p[3];
            size_t idx = ((size_t)p[0] * oshape[1] + p[1]) * oshape[2] + p[2] + svol * b;
            if (grid[idx] == -1) {
                out_coors[nact * 4 + 0] = b;
                out_coors[nact * 4 + 1] = p[0];
                out_coors[nact * 4 + 2] = p[1];
                out_coors[nact * 4 + 3] = p[2];
                grid[idx] = nact++;
            }
            int pos = pair_num[off]++;
            pairs[(size_t)(0 * kvol + off) * k_in + pos] = j;
            pairs[(size_t)(1 * kvol + off) * k_in + pos] = grid[idx];
        }
    }
    free(vp);
    free(grid);
    *k_out = nact;
    return 0;
}

/* Submanifold rulebook: outputs = inputs, stride 1, padding = dil*(k/2) forced. */
int orc_rulebook_subm(const int *coors, int k, int batch, const int shape[3],
                      const int ksize[3], const int dil[3], int *pairs, int *pair_num)
{
    int stride[3] = {1, 1, 1}, pad[3];
    for (int i = 0; i < 3; ++i) pad[i] = (ksize[i] / 2) * dil[i];
    int kvol = ksize[0] * ksize[1] * ksize[2];
    if (kvol > ORC_MAX_KVOL) return -1;
    size_t svol = (size_t)shape[0] * shape[1] * shape[2];
    int *grid = (int *)malloc(svol * batch * sizeof(int));
    if (!grid) return -2;
    memset(grid, 0xff, svol * batch * sizeof(int));
    for (size_t i = 0; i < (size_t)2 * kvol * k; ++i) pairs[i] = -1;
    memset(pair_num, 0, kvol * sizeof(int));
    for (int j = 0; j < k; ++j) {
        const int *c = coors + j * 4;
        size_t idx = ((size_t)c[1] * shape[1] + c[2]) * shape[2] + c[3] + svol * c[0];
        grid[idx] = j;
    }
    int *vp = (int *)malloc((size_t)kvol * 4 * sizeof(int) + 16);
    for (int j = 0; j < k; ++j) {
        int b = coors[j * 4];
        int nv = valid_out_pos(coors + j * 4 + 1, ksize, stride, pad, dil, shape, vp);
        for (int i = 0; i < nv; ++i) {
            const int *p = vp + i * 4;
            int off = p[3];
            size_t idx = ((size_t)p[0] * shape[1] + p[1]) * shape[2] + p[2] + svol * b;
            if (grid[idx] > -1) {
                int pos = pair_num[off]++;
                pairs[(size_t)(0 * kvol + off) * k + pos] = j;
                pairs[(size_t)(1 * kvol + off) * k + pos] = grid[idx];
            }
        }
    }
    free(vp);
    free(grid);
    return 0;
}

/* ------------------------------------------------------------------ indice_conv */

/* out[k_out,cout] = sum over offsets (ascending) of gather(in) @ W[o], scatter-added. */
void orc_indice_conv_fwd(const float *feat, const float *w, const int *pairs,
                         const int *pair_num, int k_in, int k_out, int cin, int cout, int kvol,
                         float *out)
{
    memset(out, 0, (size_t)k_out * cout * sizeof(float));
    for (int o = 0; o < kvol; ++o) {
        const int *pi = pairs + (size_t)(0 * kvol + o) * k_in;
        const int *po = pairs + (size_t)(1 * kvol + o) * k_in;
        const float *wo = w + (size_t)o * cin * cout;
        for (int p = 0; p < pair_num[o]; ++p) {
            const float *x = feat + (size_t)pi[p] * cin;
            float *y = out + (size_t)po[p] * cout;
            for (int c = 0; c < cout; ++c) {
                float acc = 0.f;
                for (int k = 0; k < cin; ++k) acc += x[k] * wo[(size_t)k * cout + c];
                y[c] += acc;
            }
        }
    }
}

/* din[i,:] += dout[o,:] @ W[k]^T ; dW[k] += in[i,:]^T dout[o,:] */
void orc_indice_conv_bwd(const float *feat, const float *w, const float *dout, const int *pairs,
                         const int *pair_num, int k_in, int k_out, int cin, int cout, int kvol,
                         float *din, float *dw)
{
    (void)k_out;
    memset(din, 0, (size_t)k_in * cin * sizeof(float));
    memset(dw, 0, (size_t)kvol * cin * cout * sizeof(float));
    for (int o = 0; o < kvol; ++o) {
        const int *pi = pairs + (size_t)(0 * kvol + o) * k_in;
        const int *po = pairs + (size_t)(1 * kvol + o) * k_in;
        const float *wo = w + (size_t)o * cin * cout;
        float *dwo = dw + (size_t)o * cin * cout;
        for (int p = 0; p < pair_num[o]; ++p) {
            const float *x = feat + (size_t)pi[p] * cin;
            const float *g = dout + (size_t)po[p] * cout;
            float *dx = din + (size_t)pi[p] * cin;
            for (int k = 0; k < cin; ++k) {
                float acc = 0.f;
                for (int c = 0; c < cout; ++c) {
                    acc += g[c] * wo[(size_t)k * cout + c];
                    dwo[(size_t)k * cout + c] += x[k] * g[c];
                }
                dx[k] += acc;
            }
        }
    }
}

/* ------------------------------------------------------------------ dense */

/* out [B,C,D,H,W] contiguous, zero filled then scattered. */
void orc_dense(const float *feat, const int *coors, int k, int c, int batch, const int shape[3],
               float *out)
{
    size_t svol = (size_t)shape[0] * shape[1] * shape[2];
    memset(out, 0, (size_t)batch * c * svol * sizeof(float));
    for (int i = 0; i < k; ++i) {
        const int *q = coors + i * 4;
        size_t sp = ((size_t)q[1] * shape[1] + q[2]) * shape[2] + q[3];
        for (int ch = 0; ch < c; ++ch)
            out[((size_t)q[0] * c + ch) * svol + sp] = feat[(size_t)i * c + ch];
    }
}
