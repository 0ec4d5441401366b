/*
 * lcf.h -- C ABI of liblcf.so: the MI355X (gfx950) native sparse front end of the SECOND
 * detector (voxelize -> VFE -> sparse 3D middle extractor) as used by
 * CMU-Light-Curtains/ObjectDetection.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless the comment says "host".
 *   - The caller (PyTorch host side) owns every buffer: inputs, outputs, workspaces.  No entry
 *     point allocates or frees device memory, none synchronises the device; all work is
 *     enqueued on `stream` (a hipStream_t passed as void*; NULL = the legacy default stream).
 *   - Re-entrant: all state lives in caller buffers and in the caller's lcf_ctx (options, open
 *     fill batch), passed as the LAST argument where an entry point has options (NULL = defaults).
 *   - Return 0 on success, a negative LCF_ERR_* otherwise; lcf_strerror() names it.
 *   - Data-dependent sizes (number of voxels, number of active outputs) come back through
 *     device int32 scalars; the caller passes capacities.
 *   - Row counts go IN as a pair (int k, const int *k_dev).  With k_dev == NULL, k is the count.
 *     With k_dev != NULL the kernels read the count from device memory, clamped to k, which is
 *     then the CAPACITY of the buffers; launch geometry depends on k only.  A whole frame can
 *     therefore be enqueued (or captured into a hipGraph) without one host synchronisation.
 *
 * The reference binds this path through the third-party `spconv` python package (1.x, not
 * vendored, README.md:29) and a numba kernel; each entry point below names the reference
 * interface / call site it replaces (paths relative to the reference repo root).
 */
#ifndef LCF_H_
#define LCF_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCF_OK 0
#define LCF_ERR_ARG (-1)         /* bad argument (null pointer, non-positive size, ...) */
#define LCF_ERR_WORKSPACE (-2)   /* workspace too small */
#define LCF_ERR_UNSUPPORTED (-3) /* shape outside what the kernels are built for */
#define LCF_ERR_RANGE (-4)       /* grid too large for 32-bit cell keys */
#define LCF_ERR_LAUNCH (-5)      /* HIP reported a launch error */
#define LCF_ERR_STATIC_LIMIT (-6) /* input larger than the no-allocation entry point's array */

#define LCF_MAX_KVOL 64 /* kernel volume limit of the rulebook / conv kernels (3x3x3 = 27) */

const char *lcf_strerror(int rc);
int lcf_version(void);

/* ------------------------------------------------------------------------------------------
 * Options and contexts (no counterpart in the reference).
 *
 * The library is re-entrant: nothing an entry point does depends on process-wide mutable state.
 * Every option travels in an lcf_ctx -- an opaque HOST object the caller creates, owns and passes
 * (as the last argument) to the entry points that have options; NULL = all defaults.  One context
 * must not be used by two threads at once; different contexts (e.g. one per FrameEngine, one per
 * thread) never interact.  A context also carries the caller's open fill batch (below).
 *
 * Options (lcf_ctx_set_option; 0 = default):
 * LCF_OPT_CONV_SPLIT: how many waves share one 16-row tile of the slab-form indice_conv forward
 *   by splitting its output columns: 1, 2 or 4; 0 = choose from the row count.
 * LCF_OPT_CONV_IMPL: forward kernel: 0 = per shape (quad-row form for >= 9 offsets with 32 / 64
 *   channels on both sides, tile-compacted for the other shapes with both channel counts >= 32,
 *   split-slab for thin layers with <= 32 output channels, slab form for the rest),
 *   1 = slab form wherever it applies, 2 = tile-compacted, 3 = split-slab wherever it applies,
 *   4 = thin-tile (compacted 64-row tiles) wherever it applies (<= 32 output channels),
 *   5 = quad-row form wherever it applies (32 / 64 channels on both sides, any offset count).
 * LCF_OPT_CONV_TILE: output rows per workgroup of the compacted forwards: 16..32, or 0 = the
 *   largest height <= 32 that deals the tiles evenly over the 256 CUs (tile-compacted form) /
 *   32 (quad-row form).
 * LCF_OPT_CONV_SETS: waves (quad-row form: 1..4, 0 = 3 at 32 input channels, 4 at 64) or wave
 *   sets (tile-compacted form: 1, 2 or 0 = two when every tile of the launch is resident at once,
 *   judged from the rows the launch expects) that share a tile by taking every n-th kernel
 *   offset of it.
 * LCF_OPT_CONV_ROWS_HINT: the number of output rows the forward calls made with this context
 *   expect, for callers that pass a capacity as k_out and keep the real count on the device; only
 *   kernel selection reads it (never bounds).  0 = use the k_out passed to the call.
 * LCF_OPT_WS_CLEAN: 1 = lcf_voxelize, lcf_rulebook_subm and lcf_rulebook_conv_discover skip the
 *   reset of their hash workspace: the caller guarantees it is in the reset state on entry
 *   (lcf_*_workspace_reset, e.g. enqueued behind the previous frame, off its critical path); the
 *   calls leave it dirty.  0 = every call resets its workspace itself.
 * LCF_OPT_NBR_CLEAN: 1 = lcf_rulebook_conv_fill skips filling out_nbr / in_nbr with -1: the
 *   caller has done it (e.g. all tables of a frame in one batched fill).
 * LCF_OPT_RB_UNFUSED: 1 = lcf_rulebook_chain launches every step by itself instead of sharing
 *   launches (profiling: each step then shows up under its own kernel name).  Same results.
 *
 * lcf_set_option() sets process-wide DEFAULTS of the four kernel-selection knobs only (SPLIT,
 * IMPL, TILE, SETS; what a context leaves at 0 takes the default) -- for benchmarks and tests that
 * pin a kernel variant.  It rejects ROWS_HINT, WS_CLEAN and NBR_CLEAN (LCF_ERR_ARG): those change
 * what a call assumes about the caller's buffers and exist per context only. */
#define LCF_OPT_CONV_SPLIT 0
#define LCF_OPT_CONV_IMPL 1
#define LCF_OPT_CONV_TILE 2
#define LCF_OPT_CONV_SETS 3
#define LCF_OPT_CONV_ROWS_HINT 4
#define LCF_OPT_WS_CLEAN 5
#define LCF_OPT_NBR_CLEAN 6
#define LCF_OPT_RB_UNFUSED 7
typedef struct lcf_ctx lcf_ctx;
int lcf_ctx_create(lcf_ctx **out);
void lcf_ctx_destroy(lcf_ctx *ctx);
int lcf_ctx_set_option(lcf_ctx *ctx, int option, int value);
int lcf_set_option(int option, int value);

/* Batched fills: between lcf_ctx_fill_batch_begin(ctx) and lcf_ctx_fill_batch_flush(ctx, stream)
 * the fills that lcf_fill_u32 and the lcf_*_workspace_reset calls would launch WHEN GIVEN THIS
 * CONTEXT are recorded in it (their stream argument is ignored) and the flush runs all of them
 * as ONE kernel on `stream`.  Calls without the context, or with another one, are unaffected. */
int lcf_ctx_fill_batch_begin(lcf_ctx *ctx);
int lcf_ctx_fill_batch_flush(lcf_ctx *ctx, void *stream);

/* ------------------------------------------------------------------------------------------
 * Voxelization.
 * Replaces: spconv.utils.VoxelGeneratorV2.generate / generate_multi_gpu
 *           (second/builder/voxel_builder.py:23-33, second/data/preprocess.py:626-639) and
 *           second.core.point_cloud.point_cloud_ops.points_to_voxel
 *           (second/kittiviewer/viewer.py:389-396; loop shape second/utils/simplevis.py:23-50).
 *
 * points [n,f] f32 row-major (x,y,z,...).  Outputs (capacity max_vox rows, every row written:
 * rows >= *voxel_num are zero): voxels [max_vox,max_pts,f] f32, coors [max_vox,3] i32 (z,y,x),
 * num_pts [max_vox] i32, voxel_num [1] i32.  Results are bit-identical to the sequential
 * first-come loop: voxel ids in order of first appearance, the first max_pts points (by index)
 * kept per voxel.  cap_mode 0 = `break` when a new voxel would exceed max_vox (numba twin),
 * 1 = `continue` (spconv C++ behaviour).  vs / range are HOST arrays.
 * coor_stride 3 writes (z,y,x); coor_stride 4 writes (batch_idx,z,y,x), i.e. the batch column
 * that merge_second_batch (second/data/preprocess.py:44-50) pads on the host.
 */
size_t lcf_voxelize_workspace_bytes(int n, int max_pts);
/* Puts a workspace of lcf_voxelize(n, max_pts) into its reset state (see LCF_OPT_WS_CLEAN). */
int lcf_voxelize_workspace_reset(void *ws, size_t ws_bytes, int n, int max_pts, void *stream,
                                 lcf_ctx *ctx);
int lcf_voxelize(const float *points, int n, const int *n_dev, int f, const float vs[3],
                 const float range[6], int max_pts, int max_vox, int cap_mode, int batch_idx,
                 int coor_stride, float *voxels, int *coors, int *num_pts, int *voxel_num,
                 void *ws, size_t ws_bytes, void *stream, lcf_ctx *ctx);
/* Diagnostics of the one-launch count + assign step (a decoupled look-back scan): a block that
 * waits too long for a predecessor's count (HIP does not promise dispatch in index order) counts
 * the first points in front of it itself -- the outputs are right either way -- and bumps a
 * counter in the workspace, cleared when a call (or an appending session) starts.  This returns
 * the counter's DEVICE address inside `ws` (one int); a non-zero value means frames ran the slow
 * way.  LCF_VOX_SPIN_LIMIT (environment) sets the polls waited, LCF_VOX_TWO_PASS selects the
 * two-launch form.  No counterpart in the reference. */
int lcf_voxelize_scan_fallbacks(void *ws, size_t ws_bytes, int n, int max_pts,
                                const int **counter_dev);

/* lcf_voxelize without a workspace argument, for callers that CANNOT ALLOCATE device memory (code
 * under hipGraph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart
 * in the reference.  The hash workspace lives in a file-scope device array of
 * LCF_VOXELIZE_STATIC_BYTES (16 MiB, held by every GPU the library runs on until the process
 * ends); the call allocates nothing and re-arms the array itself.  Inputs whose
 * lcf_voxelize_workspace_bytes(n, max_pts) exceed the array (it holds n up to 262144 points at
 * max_pts = 5, 32768 at max_pts = 35) return LCF_ERR_STATIC_LIMIT and run nothing.  All calls on
 * one GPU share the array: calls that overlap on one GPU (two streams, two threads) race.
 * Call it with the stream's device current.  Same outputs, bit for bit, as lcf_voxelize. */
#define LCF_VOXELIZE_STATIC_BYTES (16u << 20)
size_t lcf_voxelize_static_max_bytes(void);
int lcf_voxelize_static(const float *points, int n, const int *n_dev, int f, const float vs[3],
                        const float range[6], int max_pts, int max_vox, int cap_mode,
                        int batch_idx, int coor_stride, float *voxels, int *coors, int *num_pts,
                        int *voxel_num, void *stream);

/* ------------------------------------------------------------------------------------------
 * Rulebooks (spconv.ops.get_indice_pairs as called by SparseConvolution.forward for
 * second/pytorch/models/middle.py:145-192).
 *
 * The native rulebook is a pair of NEIGHBOUR TABLES instead of spconv's compacted pair lists:
 *   out_nbr [k_out,kvol] i32: input row feeding output row n through kernel offset o, or -1
 *   in_nbr  [k_in ,kvol] i32: output row fed by input row j through kernel offset o, or -1
 * (kernel offset o = kz*KH*KW + ky*KW + kx, k_d = in_d - out_d*stride_d + pad_d).
 * lcf_rulebook_pairs() converts in_nbr into spconv's canonical indice_pairs [2,kvol,k_in] /
 * indice_pair_num [kvol] (pairs of one offset in ascending input row order, -1 padded),
 * bit-identical to spconv's CPU get_indice_pairs.
 *
 * coors [k,4] i32 (b,z,y,x).  shape/ksize/stride/pad/dil are HOST int[3] (z,y,x order).
 * Limits (LCF_ERR_RANGE beyond): fewer than 2^25 rows on either side of a rulebook (the conv
 * kernels pack a gathered row id into 25 bits), k_in * prod(ceil(ksize/stride)) < 2^30, and
 * batch * D * H * W < 2^32 - 1 cells.
 */
size_t lcf_rulebook_subm_workspace_bytes(int k);
int lcf_rulebook_subm_workspace_reset(void *ws, size_t ws_bytes, int k, void *stream,
                                      lcf_ctx *ctx);
int lcf_rulebook_subm(const int *coors, int k, const int *k_dev, int batch, const int shape[3],
                      const int ksize[3], const int dil[3], int *out_nbr, void *ws,
                      size_t ws_bytes, void *stream, lcf_ctx *ctx);

/* Strided / ordinary sparse conv, two phases.  Either the caller reads *k_out between them to
 * size the tables exactly (as spconv synchronises to size its output), or it passes capacities
 * and the device count to `fill` and never synchronises.
 *   discover: out_coors [cap_out,4] in CPU-discovery (canonical) order, *k_out (device i32,
 *             clamped to cap_out).  cap_out may be below the worst case
 *             lcf_rulebook_conv_max_out(k_in, ksize, stride) only when `overflow` (device i32,
 *             zeroed by the caller) is given: it is set to 1 if outputs were dropped.
 *   fill:     out_nbr [k_out,kvol], in_nbr [k_in,kvol] (NULL = not wanted: inference needs
 *             out_nbr only); needs the SAME ws untouched in between.
 */
int lcf_rulebook_conv_max_out(int k_in, const int ksize[3], const int stride[3]);
void lcf_conv_out_shape(const int shape[3], const int ksize[3], const int stride[3],
                        const int pad[3], const int dil[3], int out_shape[3]);
size_t lcf_rulebook_conv_workspace_bytes(int k_in, const int ksize[3], const int stride[3]);
int lcf_rulebook_conv_workspace_reset(void *ws, size_t ws_bytes, int k_in, const int ksize[3],
                                      const int stride[3], void *stream, lcf_ctx *ctx);
int lcf_rulebook_conv_discover(const int *coors, int k_in, const int *k_in_dev, int batch,
                               const int shape[3], const int ksize[3], const int stride[3],
                               const int pad[3], const int dil[3], int *out_coors, int cap_out,
                               int *k_out, int *overflow, void *ws, size_t ws_bytes,
                               void *stream, lcf_ctx *ctx);
int lcf_rulebook_conv_fill(const int *coors, int k_in, const int *k_in_dev, int batch,
                           const int shape[3], const int ksize[3], const int stride[3],
                           const int pad[3], const int dil[3], int k_out, const int *k_out_dev,
                           int *out_nbr, int *in_nbr, void *ws, size_t ws_bytes, void *stream,
                           lcf_ctx *ctx);

/* All rulebooks of a SparseSequential in ONE call (the eight get_indice_pairs calls that
 * SparseConvolution.forward makes for second/pytorch/models/middle.py:145-192: four submanifold
 * keys + four strided convs), enqueued as few fused launches and without a host synchronisation.
 * Same canonical result as lcf_rulebook_subm / lcf_rulebook_conv_* (out_nbr tables, out_coors in
 * CPU-discovery order, device-side counts); builds out_nbr only (inference).
 *
 * `levels` (HOST array) lists the convs in network order.  A submanifold level (subm = 1) builds
 * out_nbr [rows_cap, kvol] over the coordinates the walk currently stands on (rows_cap = k for
 * the input level, the producing strided level's cap_out after one); a strided level produces the
 * next coordinates: out_coors [cap_out,4], k_out [1] (clamped to cap_out), overflow [1] (written
 * 0 or 1 on every call: the caller does not clear it), out_nbr [cap_out, kvol], and needs a
 * workspace of lcf_rulebook_chain_workspace_bytes(k_in_cap, ksize, stride, cap_out) bytes, where
 * k_in_cap is the row capacity of the level it reads.  Its output hash table is sized by cap_out:
 * cap_out below the real number of outputs is reported through `overflow`, never a fault.
 * ready_event (a hipEvent_t, or NULL) is recorded on `stream` as soon as the level's out_nbr (and
 * its tile plan, if `plan` is given: see lcf_conv_tile_plan) is complete, so that a consumer on
 * another stream can start before the whole chain is done.
 * ws0: lcf_rulebook_subm_workspace_bytes(k) bytes (coordinate table of the input level).
 * LCF_OPT_WS_CLEAN in ctx: the caller guarantees that lcf_rulebook_chain_reset has run on these
 * workspaces since their last use (otherwise the call resets them first). */
typedef struct lcf_rb_level {
    int subm;
    int ksize[3], stride[3], pad[3], dil[3]; /* z, y, x; stride / pad ignored for subm */
    int *out_nbr;
    int *out_coors; /* strided levels only, from here on */
    int cap_out;
    int *k_out;
    int *overflow;
    void *ws;
    size_t ws_bytes;
    void *ready_event;
    int *plan;          /* optional: lcf_conv_tile_plan of this level's out_nbr, built before */
    int plan_tile_rows; /* ready_event is recorded (tile height lcf_conv_tile_rows(...)) */
} lcf_rb_level;
size_t lcf_rulebook_chain_workspace_bytes(int k_in_cap, const int ksize[3], const int stride[3],
                                          int cap_out);
int lcf_rulebook_chain_reset(const lcf_rb_level *levels, int n, int k, void *ws0, size_t ws0_bytes,
                             void *stream, lcf_ctx *ctx);
int lcf_rulebook_chain(const int *coors, int k, const int *k_dev, int batch, const int shape[3],
                       const lcf_rb_level *levels, int n, void *ws0, size_t ws0_bytes,
                       void *stream, lcf_ctx *ctx);
/* The same walk enqueued in pieces, for a caller that interleaves it with the consumers of its
 * tables on another stream (a graph executor may order a cross-stream wait behind everything that
 * was enqueued on the producing stream before the wait, so enqueue only what the next consumer
 * needs): part 0 = level-0 table, the submanifold tables of the input resolution, the first
 * strided conv's inserts and count; part q >= 1 = ids, fill and submanifold tables of the q-th
 * strided level (+ the next one's inserts and count).  lcf_rulebook_chain_num_parts = 1 + number
 * of strided levels.  Call with the same arguments for every piece, parts in ascending order on
 * one stream; part_first = 0 re-arms the workspaces (unless LCF_OPT_WS_CLEAN). */
int lcf_rulebook_chain_num_parts(const lcf_rb_level *lv, int n);
int lcf_rulebook_chain_parts(const int *coors, int k, const int *k_dev, int batch,
                             const int shape[3], const lcf_rb_level *lv, int n, void *ws0,
                             size_t ws0_bytes, int part_first, int part_last, void *stream,
                             lcf_ctx *ctx);

/* mirror a submanifold table: in_nbr[j][o] = out_nbr[j][kvol-1-o] */
int lcf_rulebook_mirror(const int *nbr, int k, int kvol, int *mirrored, void *stream);

size_t lcf_rulebook_pairs_workspace_bytes(int k_in, int kvol);
int lcf_rulebook_pairs(const int *in_nbr, int k_in, int kvol, int *pairs, int *pair_num,
                       void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * indice_conv (torch.ops.spconv.indice_conv / indice_conv_backward as used by
 * spconv.SparseConv3d / SubMConv3d in second/pytorch/models/middle.py:127-192).
 *
 * Weights arrive in the reference layout [kD,kH,kW,Cin,Cout] = [kvol,Cin,Cout] f32 and are
 * first re-laid into an MFMA "weight image" (lcf_weight_image; transpose=1 builds the image of
 * W[o]^T for dgrad).  Image size: lcf_weight_image_elems(kvol,cin,cout) floats.
 *
 * fwd : out[n,:] = act( (sum_o feat[out_nbr[n][o],:] @ W[o]) * scale + shift ), offsets
 *       ascending, fp32 FMA accumulation on the fp32 MFMA pipe.  scale/shift [cout] may be
 *       NULL (identity; shift alone = bias); relu != 0 applies max(x,0).  This one kernel is
 *       the fused gather-GEMM-scatter: the scatter is the row's own store.  nbr == NULL with
 *       kvol == 1 is the identity table: a row-wise linear layer (spconv's kernel_size == 1
 *       short cut to torch.mm, second/pytorch/models/resnet.py:20-29; the VFE Linear layers).
 * dgrad = the same entry point with in_nbr and the transposed image.
 * wgrad: dw[o] (+)= feat[j,:]^T dout[in_nbr[j][o],:]  -> dw [kvol,cin,cout] f32 (zeroed inside).
 */
size_t lcf_weight_image_elems(int kvol, int cin, int cout, int transpose);
int lcf_weight_image(const float *w, int kvol, int cin, int cout, int transpose, float *img,
                     void *stream);
int lcf_indice_conv_fwd(const float *feat, const float *wimg, const int *nbr, int k_out,
                        const int *k_out_dev, int kvol, int cin, int cout, const float *scale,
                        const float *shift, int relu, float *out, void *stream, lcf_ctx *ctx);
/* bf16-operand variant (BASELINE.json configs[2]: "bf16 MFMA indice_conv" for training): the
 * features stay fp32 in HBM and are rounded to bf16 when they enter LDS, the weights come as a
 * bf16 image (lcf_weight_image_bf16: kvol*cin*cout 2-byte elements in
 * v_mfma_f32_16x16x32_bf16 fragment order), accumulation and output are fp32.  cin and cout must
 * be 32, 64 or 128 (LCF_ERR_UNSUPPORTED otherwise: the caller keeps the fp32 entry point).
 * Stated tolerance vs the fp32 oracle: 2e-2 of the output's max magnitude (tests). */
int lcf_weight_image_bf16(const float *w, int kvol, int cin, int cout, int transpose, void *img,
                          void *stream);
int lcf_indice_conv_fwd_bf16(const float *feat, const void *wimg_bf16, const int *nbr, int k_out,
                             const int *k_out_dev, int kvol, int cin, int cout,
                             const float *scale, const float *shift, int relu, float *out,
                             void *stream, lcf_ctx *ctx);
/* Split-operand variant ("bf16x3") of the fp32 forward, for inference at fp32 tolerance on the
 * bf16 matrix cores: every fp32 operand is x = hi + lo, hi = bf16(x), lo = bf16(x - hi)
 * (|x - hi - lo| <= 2^-18 |x|); features are split when they enter LDS, the weights come as a
 * two-plane bf16 image (lcf_weight_image_bf16x3: 2 * kvol*cin*cout 2-byte elements, as many
 * bytes as the fp32 image), and each product is lo_a*hi_w + hi_a*lo_w + hi_a*hi_w with fp32
 * accumulation.  Error per product <= 3 * 2^-18 |a||w| (1.1e-5; measured on the network's
 * features: <= 1e-5 of the tensor's max magnitude against the float64-accumulating oracle, the
 * north star's bound is 1e-4).  cin, cout in {32, 64, 128} (LCF_ERR_UNSUPPORTED otherwise: the
 * caller keeps lcf_indice_conv_fwd).  dense_coors != NULL: `out` is the cleared dense
 * [batch, cout, D, H, W] tensor of lcf_indice_conv_fwd_dense; NULL: out [k_out, cout], and
 * batch / shape are ignored.  No counterpart in the reference (its spconv runs fp32 GEMMs). */
int lcf_weight_image_bf16x3(const float *w, int kvol, int cin, int cout, int transpose,
                            void *img, void *stream);
int lcf_indice_conv_fwd_bf16x3(const float *feat, const void *wimg_bf16x3, const int *nbr,
                               int k_out, const int *k_out_dev, int kvol, int cin, int cout,
                               const float *scale, const float *shift, int relu, float *out,
                               const int *dense_coors, int batch, const int shape[3],
                               void *stream, lcf_ctx *ctx);
/* Three-plane form of the same ("bf16x8"): x = hi + mid + lo exactly (three bf16 planes, 8 + 8 + 8
 * significand bits) and the eight products down to 2^-24 |a||w| -- what is dropped (lo*lo, 2^-32)
 * is far below the rounding of fp32 operands' own products, so results are fp32-grade (measured:
 * the same distance from the float64 oracle as lcf_indice_conv_fwd), at half the matrix-pipe
 * cycles of the fp32 MFMA.  Image: 3 * kvol*cin*cout 2-byte elements.  Same arguments, shapes
 * and dense-target option as the bf16x3 pair. */
int lcf_weight_image_bf16x8(const float *w, int kvol, int cin, int cout, int transpose,
                            void *img, void *stream);
int lcf_indice_conv_fwd_bf16x8(const float *feat, const void *wimg_bf16x8, const int *nbr,
                               int k_out, const int *k_out_dev, int kvol, int cin, int cout,
                               const float *scale, const float *shift, int relu, float *out,
                               const int *dense_coors, int batch, const int shape[3],
                               void *stream, lcf_ctx *ctx);
/* fwd with SparseConvTensor.dense() (second/pytorch/models/middle.py:206, right after the last
 * SparseConv3d + BatchNorm1d + ReLU of the stack, :186-192) fused into its epilogue: instead of
 * out[row,:] the result lands in dense[b, :, z, y, x] of a [batch, cout, D, H, W] f32 tensor,
 * (b,z,y,x) = out_coors[row] ([k_out,4] int32, 16-byte aligned; rows with a coordinate outside
 * the shape are dropped, as lcf_dense drops them).  The caller clears `dense` beforehand.  cout
 * must be 16, 32, 64 or 128 and nbr non-NULL (LCF_ERR_UNSUPPORTED / LCF_ERR_ARG otherwise: the
 * caller keeps lcf_indice_conv_fwd + lcf_dense).  Same arithmetic as lcf_indice_conv_fwd:
 * bit-identical values. */
int lcf_indice_conv_fwd_dense(const float *feat, const float *wimg, const int *nbr, int k_out,
                              const int *k_out_dev, int kvol, int cin, int cout,
                              const float *scale, const float *shift, int relu,
                              const int *out_coors, int batch, const int shape[3], float *dense,
                              void *stream, lcf_ctx *ctx);
/* Tile plans (no counterpart in the reference: spconv re-derives its gather lists per call).
 * The tile-compacted forward kernels start every tile by compacting its rows per kernel offset;
 * the result depends on the neighbour table only, so it can be computed ONCE per rulebook and
 * shared by every conv over it (the two or three SubMConv3d of one indice_key,
 * second/pytorch/models/middle.py:146-183) and built off the critical path.
 *   lcf_conv_tile_rows_for(rows, kvol, cin, cout, math): the tile height the default forward of
 *     that shape and arithmetic picks for a launch expecting `rows` (lcf_conv_tile_rows(rows):
 *     the tile-compacted kernels' choice, whatever the shape).
 *   lcf_conv_tile_plan: plan [lcf_conv_tile_plan_ints(k, kvol, tile_rows)] int32 (16-byte
 *     aligned) from nbr [k, kvol] (out_nbr for the forward, in_nbr for dgrad); k / k_dev as
 *     everywhere.
 *   lcf_indice_conv_fwd_planned: the forward over a plan instead of a table -- same kernels, same
 *     arithmetic and summation order, bit-identical results.  math 0: fp32 MFMA
 *     (lcf_indice_conv_fwd, or _dense when dense_coors != NULL), 1: bf16 operands (no dense
 *     target), 2: bf16x3, 3: bf16x8; wimg is the weight image of that form.  cin, cout in
 *     {32, 64, 128}; k_out (capacity) and tile_rows must be the ones the plan was built with. */
size_t lcf_conv_tile_plan_ints(int k_cap, int kvol, int tile_rows);
int lcf_conv_tile_rows(int rows);
int lcf_conv_tile_rows_for(int rows, int kvol, int cin, int cout, int math);
int lcf_conv_tile_plan(const int *nbr, int k, const int *k_dev, int kvol, int tile_rows, int *plan,
                       void *stream);
int lcf_indice_conv_fwd_planned(const float *feat, const void *wimg, int math, const int *plan,
                                int tile_rows, int k_out, const int *k_out_dev, int kvol, int cin,
                                int cout, const float *scale, const float *shift, int relu,
                                float *out, const int *dense_coors, int batch, const int shape[3],
                                void *stream, lcf_ctx *ctx);
int lcf_indice_conv_wgrad(const float *feat, const float *dout, const int *in_nbr, int k_in,
                          int kvol, int cin, int cout, float *dw, void *stream);

/* dw [cin, cout] f32 = x^T dy over n dense rows (x [n, cin], dy [n, cout] row-major f32; cin,
 * cout <= 128): weight gradient of the VFE's row-wise Linear layers (voxel_encoder.py:64, 127)
 * -- called with x = dY and dy = X it yields nn.Linear's [out, in] layout.  fp32 MFMA, atomic
 * accumulation over row chunks (not bit-reproducible); dw is zeroed inside. */
int lcf_linear_wgrad(const float *x, const float *dy, int n, int cin, int cout, float *dw,
                     void *stream);

/* wgrad on the bf16 matrix cores (K dimension of one MFMA = 32 pairs).  planes = 1: operands
 * rounded to bf16, fp32 accumulation (BASELINE.json configs[2], tolerance 2e-2 of max |dw|);
 * planes = 2: operands split hi + lo as in lcf_indice_conv_fwd_bf16x3, three products per term
 * (fp32 tolerance 1e-4; measured ~1e-5).  feat / dout row-major f32 with exactly cin / cout
 * columns, cin, cout in {32, 64, 128} and cin * cout <= 4096 (LCF_ERR_UNSUPPORTED otherwise: the
 * caller keeps lcf_indice_conv_wgrad).  dw [kvol,cin,cout] f32 is zeroed inside. */
int lcf_indice_conv_wgrad_bf16(const float *feat, const float *dout, const int *in_nbr, int k_in,
                               int kvol, int cin, int cout, int planes, float *dw,
                               void *stream);

/* indice_conv for the training step on rows STORED as bf16 (round 3; BASELINE.json configs[2]).
 * Replaces the same reference calls as lcf_indice_conv_fwd / _wgrad (indice_conv and
 * indice_conv_backward inside spconv's SparseConvolution, second/pytorch/models/middle.py:145-192)
 * when the activations a conv gathers have been written as bf16 by their producer
 * (lcf_bn_relu_fwd_ex / _bwd_ex): gathered rows are 128 bytes at 64 channels, operands go to
 * v_mfma_f32_16x16x32_bf16 without conversion, accumulation is fp32.
 * fwd_b16: feat [k_in,cin] bf16 (k_in * cin * 2 < 2 GB: the gathers are range-checked buffer
 *   loads, a missing neighbour reads as zeros without touching memory; every entry of nbr must be
 *   < k_in), wimg = lcf_weight_image_bf16, nbr [k_out,kvol]; out [k_out,cout]
 *   fp32 (out_bf16 = 0) or bf16 rounded to nearest even (1).  dgrad: the same call over in_nbr
 *   with the transposed image and cin / cout exchanged.  cin, cout in {32, 64, 128}.  No BatchNorm
 *   epilogue (training BatchNorm needs the batch statistics of `out` first).  Deterministic.
 * wgrad_b16: feat [k_in,cin] bf16, dout [k_out,cout] bf16, in_nbr [k_in,kvol] -> dw
 *   [kvol,cin,cout] fp32 (zeroed inside; fp32 atomics across workgroups); cin, cout in {32, 64}.
 * Stated tolerance of the bf16 forms: 2e-2 of the result's max magnitude against the fp32
 * oracle (operands carry 8 significand bits); measured in tests/test_gpu_conv.py. */
int lcf_indice_conv_fwd_b16(const void *feat_bf16, int k_in, const void *wimg_bf16,
                            const int *nbr, int k_out, int kvol, int cin, int cout, void *out,
                            int out_bf16, void *stream);
int lcf_indice_conv_wgrad_b16(const void *feat_bf16, const void *dout_bf16, const int *in_nbr,
                              int k_in, int kvol, int cin, int cout, float *dw, void *stream);

/* Inference on three exact bf16 planes, six products ("bf16x6"; round 3).  Replaces, for the
 * 27-offset convs of 32 / 64 channels of one frame, the same reference call as
 * lcf_indice_conv_fwd (spconv SparseConvolution.forward -> indice_conv,
 * second/pytorch/models/middle.py:145-192) with fp32-GRADE arithmetic on the bf16 matrix cores:
 * an fp32 value is exactly h + m + l with three bf16 numbers, and x*w = hH + (hM + mH) +
 * (hL + lH + mM) up to 2^-24 |x*w|; all six products are exact in the fp32 accumulator.
 * Measured distance from a float64 reference: the same as the fp32 MFMA kernels' (a few 1e-7 of
 * the result's max magnitude; tests/test_gpu_conv.py).
 * Rows travel BETWEEN such convs as planes: [3][plane_rows][c] bf16, plane p at byte offset
 * p * plane_rows * c * 2 (plane_rows * c * 2 < 2 GB and a multiple of 16).
 * lcf_split_planes: feat [k,c] fp32 -> planes (rows >= the count are left untouched).
 * lcf_indice_conv_fwd_x6: feat_planes as above, wimg_planes = lcf_weight_image_bf16x8 (its three
 *   planes), nbr [k_out,kvol] (every entry < plane_rows), k_out / k_out_dev as everywhere;
 *   y = relu?((sum) * scale + shift) (scale / shift may be NULL); out = fp32 rows [k_out,cout]
 *   (out_planes = 0) or planes [3][out_plane_rows][cout] (out_planes = 1, out_plane_rows >= k_out).
 *   (cin, cout) in {(32,32), (32,64), (64,64)}.  Deterministic, no workspace, no tile plan. */
int lcf_split_planes(const float *feat, int k, const int *k_dev, int c, void *planes,
                     int plane_rows, void *stream);
int lcf_indice_conv_fwd_x6(const void *feat_planes, int plane_rows, const void *wimg_planes,
                           const int *nbr, int k_out, const int *k_out_dev, int kvol, int cin,
                           int cout, const float *scale, const float *shift, int relu, void *out,
                           int out_planes, int out_plane_rows, void *stream);

/* Frame hand-over for a caller that keeps a static point buffer and a device-side point count
 * (FrameEngine; no counterpart in the reference): dst[0 : n*f] = points[0 : n*f] and
 * n_dev[0] = n_total (>= n: dst may point behind points already in the buffer, the light-curtain
 * loop's append), one launch.  n * f must be a multiple of 4 and both pointers 16-byte aligned
 * (LCF_ERR_UNSUPPORTED otherwise: copy and fill separately). */
int lcf_load_points(const float *points, int n, int f, float *dst, int n_total, int *n_dev,
                    void *stream);

/* Fill `bytes` (multiple of 4, 4-byte aligned) with a 32-bit pattern: a plain kernel, usable
 * where hipMemsetAsync is not wanted (graph capture; no counterpart in the reference). */
int lcf_fill_u32(void *p, uint32_t value, size_t bytes, void *stream, lcf_ctx *ctx);

/* ------------------------------------------------------------------------------------------
 * SparseConvTensor.dense() (second/pytorch/models/middle.py:206): out [B,C,D,H,W] f32 is
 * zero-filled (zero_fill != 0; a caller that has already cleared `out`, e.g. on another stream,
 * passes 0) and scattered from feat [k,c], coors [k,4].  dense_bwd gathers it back.
 */
int lcf_dense(const float *feat, const int *coors, int k, const int *k_dev, int c, int batch,
              const int shape[3], int zero_fill, float *out, void *stream);
/* Rows-only clear of a dense() output that is re-used frame after frame (SURVEY.md section 8f-1;
 * the reference's dense() allocates and zero-fills the whole [B,C,D,H,W] tensor per call,
 * second/pytorch/models/middle.py:206 -> torchplus/ops/array_ops.py:7-18).  `prev_coors` [cap,4]
 * i32 is caller-owned state: the (b,z,y,x) the previous frame's rows were scattered to, -1 rows =
 * none (initialise to -1 together with a zeroed `dense`).  The call zeroes dense[b,:,z,y,x] for
 * every row of prev_coors, then sets prev_coors = coors[0 : k] (-1 beyond k; k = *k_dev clamped
 * to cap, or cap when k_dev is NULL) -- the rows the caller is about to scatter.  Afterwards
 * `dense` is all zero again provided every earlier scatter went through this bookkeeping.  With
 * an open fill batch on `ctx` the call is recorded (one per batch) and runs inside the flush's
 * launch; otherwise it is one launch on `stream`.  c = channels, shape = (D,H,W). */
int lcf_dense_clear_rows(float *dense, int *prev_coors, const int *coors, int cap,
                         const int *k_dev, int c, int batch, const int shape[3], void *stream,
                         lcf_ctx *ctx);
int lcf_dense_bwd(const float *dout, const int *coors, int k, int c, int batch,
                  const int shape[3], float *dfeat, void *stream);

/* VFE training, the tail of a layer after the Linear and the batch statistics
 * (voxel_encoder.py:69-80, 134-139, over the real points only): y [p,c] = relu(x * scale + shift)
 * and agg[seg[r], :] = max(agg[seg[r], :], y[r, :]) in one pass.  seg [p] int32 voxel id of each
 * row; agg [k,c] arrives initialised (values >= 0, or any negative number for "no contribution")
 * and every seg[r] must lie in [0, k).  c % 4 == 0, c <= 128.  _bwd: with
 * g = [y > 0] * (gy + [y == agg[seg]] * gagg[seg]) (gy may be NULL = zeros):
 * dx = g * scale, dscale = sum_r g * x, dshift = sum_r g (fixed-order sums, deterministic).
 * Workspace: lcf_bn_workspace_bytes(p, c). */
int lcf_vfe_affine_relu_segmax_fwd(const float *x, int p, int c, const float *scale,
                                   const float *shift, const int *seg, float *y, float *agg,
                                   void *stream);
int lcf_vfe_affine_relu_segmax_bwd(const float *x, const float *y, const float *agg,
                                   const float *gy, const float *gagg, const int *seg, int p,
                                   int c, const float *scale, float *dx, float *dscale,
                                   float *dshift, void *ws, size_t ws_bytes, void *stream);

/* dense() handed to the RPN's first convolution without a layout or dtype pass (SURVEY.md
 * section 8f-1; middle.py:206-209 + rpn.py:486-495 under bf16 autocast): the [B, C*D, H, W] view
 * as channels_last bf16, memory [B, H, W, C*D], element (b, c, z, y, x) at
 * ((b*H + y)*W + x)*(C*D) + c*D + z, rounded to nearest even.  _bwd gathers the gradient of that
 * tensor (same layout, bf16) back into dfeat [k, c] f32.  B*C*D*H*W must be even. */
int lcf_dense_nhwc_bf16(const float *feat, const int *coors, int k, const int *k_dev, int c,
                        int batch, const int shape[3], int zero_fill, void *out_bf16,
                        void *stream);
int lcf_dense_nhwc_bf16_bwd(const void *dout_bf16, const int *coors, int k, int c, int batch,
                            const int shape[3], float *dfeat, void *stream);
/* The same pair with the feature rows (feat / dfeat) stored as fp32 (0) or bf16 (1): the bf16
 * training path hands its last layer's bf16 rows over as they are and takes the gradient back as
 * bf16 rows. */
int lcf_dense_nhwc_bf16_rows(const void *feat, int feat_bf16, const int *coors, int k,
                             const int *k_dev, int c, int batch, const int shape[3],
                             int zero_fill, void *out_bf16, void *stream);
int lcf_dense_nhwc_bf16_bwd_rows(const void *dout_bf16, const int *coors, int k, int c, int batch,
                                 const int shape[3], void *dfeat, int dfeat_bf16, void *stream);

/* ------------------------------------------------------------------------------------------
 * VFE streaming kernels (second/pytorch/models/voxel_encoder.py).
 * vfe_mean : SimpleVoxel.forward (:220-225): out[k,:c] = sum_t voxels[k,t,:c] / num[k].
 * vfe_augment: VoxelFeatureExtractor.forward :116-124: out[k,t,:] = concat(voxels[k,t,:f],
 *            voxels[k,t,:3] - mean_t) ([k,t,f+3]).
 * vfe_pool : the tail of VFELayer.forward (:69-80) and of VoxelFeatureExtractor.forward
 *            (:134-139) after the Linear: y = relu(x*scale + shift) per channel (BatchNorm in
 *            affine form), then
 *              mode 0: out [k,t,2u] = concat(y, max_t y) * mask[k,t]      (VFELayer + mask)
 *              mode 1: out [k,u]    = max_t (y * mask[k,t])               (final max-pool)
 *            with mask[k,t] = t < num[k]  (get_paddings_indicator :27-48).
 */
int lcf_vfe_mean(const float *voxels, const int *num, int k, const int *k_dev, int t, int f,
                 int c, float *out, void *stream);
int lcf_vfe_augment(const float *voxels, const int *num, int k, const int *k_dev, int t, int f,
                    float *out, void *stream);
int lcf_vfe_pool(const float *x, const int *num, int k, const int *k_dev, int t, int u,
                 const float *scale, const float *shift, int mode, float *out, void *stream);

/* VoxelFeatureExtractor.forward (voxel_encoder.py:113-140) in ONE persistent kernel for the
 * shipped shape: f (+3 mean-centred) <= 8 inputs, VFE layers of u1 = 16 and u2 = 64 units, final
 * Linear of u3 = 128 (num_filters [32,128]); any other shape, or t too large for the group to
 * stay in LDS, returns LCF_ERR_UNSUPPORTED and the caller uses the layer-by-layer entry points.
 * w?img: lcf_weight_image of the Linear weights as [1,in,out]; sc?/sh?: eval-mode BatchNorm in
 * affine form (NULL = identity / bias only).  out [k,128]. */
int lcf_vfe_fused(const float *voxels, const int *num, int k, const int *k_dev, int t, int f,
                  int u1, int u2, int u3, const float *w1img, const float *sc1, const float *sh1,
                  const float *w2img, const float *sc2, const float *sh2, const float *w3img,
                  const float *sc3, const float *sh3, float *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Training-mode BatchNorm1d + ReLU over the active rows of a sparse tensor.
 * Replaces: the nn.BatchNorm1d(eps=1e-3, momentum=0.01) -> nn.ReLU pair between the sparse convs
 * (second/pytorch/models/middle.py:124-125,147-148; torch's native_batch_norm /
 * native_batch_norm_backward + threshold_backward), batch statistics over the K rows.
 * x, y, dy, dx [k,c] fp32, c a multiple of 4 and <= 128 (else LCF_ERR_UNSUPPORTED); gamma / beta
 * [c] or NULL (1 / 0); relu 0 = plain BatchNorm.
 * fwd: y = relu((x - mean) * inv * gamma + beta); stats [3c] = mean, biased variance,
 *      inv = 1/sqrt(var + eps) (the caller updates running statistics from it).
 * bwd: with g = dy * (y > 0): dx, dgamma = sum g * xhat, dbeta = sum g (either may be NULL).
 * Deterministic (fixed partition of the rows, fixed order).  ws: lcf_bn_workspace_bytes. */
size_t lcf_bn_workspace_bytes(int k, int c);
int lcf_bn_relu_fwd(const float *x, int k, int c, const float *gamma, const float *beta,
                    float eps, int relu, float *y, float *stats, void *ws, size_t ws_bytes,
                    void *stream);
int lcf_bn_relu_bwd(const float *x, const float *y, const float *dy, int k, int c,
                    const float *gamma, const float *stats, int relu, float *dx, float *dgamma,
                    float *dbeta, void *ws, size_t ws_bytes, void *stream);
/* The same pair with rows stored as fp32 or bf16 (*_bf16 = 1: that tensor's rows are bf16, c a
 * multiple of 8) and the module's running statistics updated by the forward itself.  bf16 rows
 * are for the tensors a sparse conv GATHERS: this pair's output y (the next conv's input) and
 * the gradient dx it hands back (the dgrad / wgrad operand of the conv in front), so that the
 * gathers of the training step move 128-byte rows (BASELINE.json configs[2]: bf16 MFMA
 * indice_conv); conv outputs x stay fp32 and every statistic, sum and product here is fp32 /
 * double.  Values are rounded to nearest even when stored.
 * running_mean / running_var [c] (both or neither) and num_batches_tracked [1] int64 (or NULL):
 * mean <- (1-m) mean + m batch_mean, var <- (1-m) var + m unbiased batch var, batches += 1, as
 * nn.BatchNorm1d does after its forward (torch/nn/modules/batchnorm.py; the reference's
 * momentum 0.01, second/pytorch/models/middle.py:124). */
int lcf_bn_relu_fwd_ex(const void *x, int x_bf16, int k, int c, const float *gamma,
                       const float *beta, float eps, int relu, void *y, int y_bf16, float *stats,
                       float *running_mean, float *running_var, long long *num_batches_tracked,
                       float momentum, void *ws, size_t ws_bytes, void *stream);
int lcf_bn_relu_bwd_ex(const void *x, int x_bf16, const void *y, int y_bf16, const void *dy,
                       int dy_bf16, int k, int c, const float *gamma, const float *stats, int relu,
                       void *dx, int dx_bf16, float *dgamma, float *dbeta, void *ws,
                       size_t ws_bytes, void *stream);
/* Backward of the pair on ALL-bf16 rows (x, dy, dx bf16; 16 bytes per thread): the form the dense
 * RPN's BatchNorm2d + ReLU takes on channels_last activations under bf16 autocast (rows = pixels,
 * second/pytorch/models/rpn.py:467-497 in training; its forward is lcf_bn_relu_fwd_ex with x_bf16
 * = y_bf16 = 1).  The ReLU mask is the sign of the forward's output RECOMPUTED from x, gamma, beta
 * and stats with the forward's own fused multiply-add, so y is not read (beta is needed for
 * that).  c a multiple of 8 and <= 128. */
int lcf_bn_relu_bwd_b16(const void *x_bf16, const void *dy_bf16, int k, int c, const float *gamma,
                        const float *beta, const float *stats, int relu, void *dx_bf16,
                        float *dgamma, float *dbeta, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Rotated / axis-aligned IoU and non-maximum suppression (the predict-time
 * post-processing that the reference runs through the host).
 * lcf_rotate_iou replaces rotate_iou_gpu / rotate_iou_gpu_eval
 *   (second/core/non_max_suppression/nms_gpu.py:478-640): boxes [n,5], qboxes [k,5] as
 *   (cx, cy, w, h, angle), or with corners != 0 [n,8] / [k,8] corner coordinates (the form
 *   spconv.utils.rbbox_iou takes); iou [n,k]; criterion -1: IoU, 0: inter / area(qbox),
 *   1: inter / area(box), 2: intersection area.
 * lcf_nms replaces nms_gpu / rotate_nms_gpu (nms_gpu.py:129-163, :440-475: pairwise kernel +
 *   the host loop nms_postprocess) and spconv's non_max_suppression_cpu /
 *   rotate_non_max_suppression_cpu as called from second/pytorch/core/box_torch_ops.py:473-515.
 *   dets: n rows of `stride` floats ALREADY sorted by descending score; mode 0: (x1, y1, x2, y2)
 *   with the +eps pixel convention (eps = 1 in the reference's callers), mode 1: (cx, cy, w, h,
 *   angle), mode 2: 8 corner coordinates.  A box is dropped when an earlier kept box overlaps it by more than thresh
 *   (inclusive != 0: at least thresh, spconv's / nms_jit's test).  keep [min(n, max_keep)] gets
 *   the kept positions in order, num_keep [1] their number (device memory; max_keep <= 0: all).
 *   ws: lcf_nms_workspace_bytes(n) (the n x n/64 suppression matrix); n <= 65536. */
int lcf_rotate_iou(const float *boxes, int n, const float *qboxes, int k, int corners,
                   int criterion, float *iou, void *stream);
size_t lcf_nms_workspace_bytes(int n);
int lcf_nms(const float *dets, int n, int stride, int mode, float thresh, float eps,
            int inclusive, int max_keep, int *keep, int *num_keep, void *ws, size_t ws_bytes,
            void *stream);

/* A VFE layer's tail in TRAINING mode over the real rows only: batch statistics, BatchNorm affine,
 * ReLU, per-voxel maximum and the (pointwise | repeated) concatenation of
 * second/pytorch/models/voxel_encoder.py:65-80 (VFELayer.forward) and :126-139 (the final Linear +
 * norm + masked max), restated over the p rows that hold a point (the reference's [K, T, C]
 * tensors are 78 % padding on a 20k-point frame; objectdetection_amd/voxel_encoder.py
 * ::_forward_compact).  x [p, c] = Linear output of the real rows, seg [p] = voxel of each row,
 * non-decreasing (a voxel's rows are contiguous) and every voxel 0..k-1 owning a row or not;
 * n = K * T rows enter the statistics:
 *   mode 0 (first layer): voxel v's padding rows are wp[v] copies of xp[v] (Linear output of its
 *          padding representative): weight wp[v] in the statistics, once in the maximum;
 *   mode 1 (later layer): padding rows are zeros: n - p zeros in the statistics, relu(shift) in
 *          the maximum of every voxel with wp[v] > 0;
 *   mode 2 (final Linear): statistics as mode 1, maxima start at 0.
 * fwd: stats [5c] = scale | shift | mean | biased var | inv; modes 0 / 1: out [p, 2c] =
 *   (relu(x * scale + shift) | agg[seg]) and agg [k, c]; mode 2: agg [k, c] is the result.
 * bwd: gout = gradient of out [p, 2c] (modes 0 / 1) or of agg [k, c] (mode 2); dx [p, c], dxp [k, c]
 *   (mode 0), dgamma / dbeta [c] -- gradients THROUGH the batch statistics included.
 * c % 4 == 0, c <= 128.  ws: lcf_vfe_layer_workspace_bytes(p, k, c).  Deterministic. */
size_t lcf_vfe_layer_workspace_bytes(int p, int k, int c);
int lcf_vfe_layer_fwd(const float *x, int p, const float *xp, const float *wp, int k, int c,
                      long long n, const int *seg, const float *gamma, const float *beta, float eps,
                      int mode, float *out, float *agg, float *stats, void *ws, size_t ws_bytes,
                      void *stream);
int lcf_vfe_layer_bwd(const float *x, const float *agg, const float *gout, int p, const float *xp,
                      const float *wp, int k, int c, long long n, const int *seg,
                      const float *gamma, const float *stats, int mode, float *dx, float *dxp,
                      float *dgamma, float *dbeta, void *ws, size_t ws_bytes, void *stream);

/* Incremental voxelisation of a growing cloud.  The light-curtain loop
 * stacks every curtain's returns under the cumulative cloud and voxelises the whole cloud again on
 * the host (second/light_curtain/utils.py:121-124 -> dataset._prep_func_main ->
 * second/data/preprocess.py:626-639).  Here the cloud [n_capacity, f], the workspace
 * (lcf_voxelize_workspace_bytes(n_capacity, max_pts)) and the four outputs stay resident on the
 * device; a call voxelises only the points [n_begin, n_end) the caller has just written under the
 * resident ones and leaves the outputs EQUAL to lcf_voxelize on points[0 .. n_end) from scratch
 * (same voxel order, slots, counts and max_vox cut-off).  n_begin = 0 starts a session (the
 * workspace is re-armed, rows beyond voxel_num zeroed); later calls must pass the previous call's
 * n_end as n_begin and the same geometry arguments and buffers.  Point counts are host values. */
int lcf_voxelize_append(const float *points, int n_capacity, int n_begin, int n_end, int f,
                        const float vs[3], const float range[6], int max_pts, int max_vox,
                        int cap_mode, int batch_idx, int coor_stride, float *voxels, int *coors,
                        int *num_pts, int *voxel_num, void *ws, size_t ws_bytes, void *stream,
                        lcf_ctx *ctx);

/* ------------------------------------------------------------------------------------------
 * Light-curtain placement next to the detector: what LC_PROCESS
 * (second/light_curtain/utils.py:88-104) does on the host between two detector passes.
 * lcf_lc_entropy_lookup replaces torch.sigmoid(cls_preds).cpu() + _get_confidence_map's masking
 *   and flip (utils.py:93-95, 208-235) + Policy.confidence2entropy (policy.py:60-66) +
 *   DPOptimizedPolicy._umap2ulookup (policy.py:154-158): cls [K,Y,X] f32 class scores (logits
 *   when apply_sigmoid != 0), anchors_mask [K,Y,X] u8 or NULL; ulookup [Y*X + 1] f32 with
 *   ulookup[0] = 0 and ulookup[1 + xi*X + zi] = mean_k H(clip(p_k, 1e-5, 1 - 1e-5)) of the cell at
 *   camera-x index xi = Y-1-y, camera-z index zi = x.
 * lcf_lc_dp_design replaces DPOptimizedPolicy.get_design_points (policy.py:283-434) given the
 *   cached graph (policy.py:160-281, built on the host from the device geometry): node_k [T,N]
 *   i32 = index of every ray point into ulookup [n_lookup] (an index outside the table -- a
 *   graph built for another map shape -- reads entry 0, "outside the map"), theta_las [T,N] f64 = its laser angle in degrees,
 *   max_dtheta = laser_vel_limit * laser_timestep; choice [T] i32 = the point chosen on each
 *   ray t (rays in the graph's order; the reference returns their (x, z) flipped).  Value
 *   functions in float32 (uncertainty) and float64 (laser cost) as numpy computes them: the
 *   trajectory is the reference's, tie-breaks included.  N <= 256.  ws:
 *   lcf_lc_dp_workspace_bytes(T, N).  One workgroup: latency ~0.3 us per ray. */
int lcf_lc_entropy_lookup(const float *cls, int K, int Y, int X,
                          const unsigned char *anchors_mask, int apply_sigmoid, float *ulookup,
                          void *stream);
size_t lcf_lc_dp_workspace_bytes(int T, int N);
int lcf_lc_dp_design(const float *ulookup, int n_lookup, const int *node_k,
                     const double *theta_las, int T, int N, double max_dtheta, int *choice,
                     void *ws, size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LCF_H_ */
