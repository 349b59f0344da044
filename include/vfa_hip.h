/*
 * vfa_hip.h -- C ABI of the MI355X (gfx950) multiview feature -> voxel projection + aggregation path.
 *
 * The reference (Jiahao-Ma/VFA) has no FFI on this path: its boundary is the Python call
 * VFA.forward (vfa/model/vfa_op.py:61-125) and the camera loop of VFANet.forward
 * (vfa/model/vfanet.py:64-82).  This header is the boundary the MI355X build introduces underneath
 * that call; vfa_amd/vfa_op.py binds it with ctypes (see INTEGRATION.md for the stub a maintainer of
 * the reference would add).  Each entry point names the reference lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm allocations in practice);
 *     the library allocates nothing, keeps no global state (every tuning choice is a per-call flag) and is re-entrant;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is stream-ordered
 *     and asynchronous, nothing synchronises the device;
 *   - return value: 0 on success, otherwise a hipError_t (launch failure) or VFA_ERR_* (bad argument);
 *     nothing throws, nothing exits;
 *   - fp32 everywhere, with the rounding sequence of the reference's PyTorch CPU path (SURVEY.md
 *     Appendix A): the STAGE entry points (integral images, box parameters, vfa_gather_f32 / vfa_project_gather_f32 /
 *     vfa_pool_windows_f32: the voxel features) are bit-identical to the reference up to the sign of zero of masked
 *     voxels.  The FUSED frame entry points (vfa_pool_collapse_relu_sum_f32, vfa_pipe_collapse_relu_sum_f32) never
 *     write voxel features; inside them the tap chains and the box sum are the same exact sequence, but the quotient is
 *     v * RN(1 / area) where the reference divides: their on-chip voxel features are within ONE unit in the last place of
 *     the reference's (79-87 % of them identical: VFA_FLAG_DUMP_VOX, tests/test_fused_frame.py), and
 *     their output is compared with the reference within the post-GEMM tolerance (rtol 1e-4, atol 1e-5 max|ref|);
 *   - `n_views` batches cameras that share feature-map and grid shapes (one scale of one frame).
 *
 * Layouts
 *   feature   (n_views, C, Hf, Wf)          NCHW, as the reference's lateral maps
 *   integral  (n_views, Hf+2, Wf+2, C)      channels-last with a one-pixel ZERO border: one tap of one box is C
 *                                           contiguous floats and grid_sample's zeros padding is a plain load
 *   box       (n_views, nl, n_cells, 4)     left, top, right, bottom in normalised [-1,1] image coords
 *   area      (n_views, nl, n_cells)
 *   visible   (n_views, nl, n_cells)        0/1 bytes
 *   vox       (n_views, cell_count, nl*C)   column = layer*C + c (VFA_VOX_LAYER_MAJOR) or
 *                                           column = c*nl + layer (VFA_VOX_REFERENCE, vfa_op.py:120)
 */
#ifndef VFA_HIP_H
#define VFA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFA_ABI_VERSION 9

/* world-unit conversion of the grid, reference vfa_op.py:23-44 (chosen by args.data) */
#define VFA_CONV_MULTIVIEWC 0 /* x / 1.0                                   */
#define VFA_CONV_MULTIVIEWX 1 /* x / 40.0 (true division)                  */
#define VFA_CONV_WILDTRACK 2  /* x*2.5-300, y*2.5-900, z*2.5               */

#define VFA_VOX_REFERENCE 0   /* vox[cell, c*nl + layer]   (vfa_op.py:120) */
#define VFA_VOX_LAYER_MAJOR 1 /* vox[cell, layer*C + c]    (coalesced; collapse.weight columns permuted by the host) */

/* optional kernel choice for vfa_project_gather_f32, OR-ed into vox_layout (same results bit for bit):
 * neither flag = library default (tap cache on single-layer grids with C = 256, direct kernel otherwise) */
#define VFA_VOX_KERNEL_DIRECT 0x100
#define VFA_VOX_KERNEL_TAP_CACHE 0x200

#define VFA_ERR_BAD_ARGUMENT 10001
#define VFA_ERR_UNSUPPORTED 10002 /* shape outside what a specialised kernel was built for; use the general entry points */

/* ABI version of the loaded library (VFA_ABI_VERSION at build time). */
int vfa_abi_version(void);

/* Per-call `flags` of the MFMA collapse entry points (there is no process-wide state):
 *   bits 0-3   arithmetic of the `collapse` product (reference: an fp32 nn.Linear, vfa_op.py:59, :123):
 *                0 / 2  (default) TWO FP16 PIECES per operand, three products hi.hi + hi.lo + lo.hi, fp32 accumulation -- with a
 *                       power-of-two scale per operand and feature scale the split is exact to 22 bits and the result has the error
 *                       of an sgemm (2e-7 normwise against float64 at K = 256): the reference's arithmetic width.  The fused frame
 *                       kernels only (vfa_pool_collapse_relu_sum_f32, vfa_pipe_collapse_relu_sum_f32 and their geometry calls);
 *                       the unfused product kernels (vfa_collapse_relu_sum_f32, vfa_collapse_gemm_*) read 0 as 3.
 *                3      two bf16 pieces, three products (16 bits: ~4e-6 normwise; inside the path's 1e-4 / 1e-5 tolerance)
 *                4      ... plus lo.lo
 *                6      three bf16 pieces, six products (sgemm class at twice the matrix work; vfa_pipe_* only)
 *   bits 8-15  VFA_FLAG_RESERVED_CUS(n): the persistent kernels (one workgroup per CU with all of its LDS) launch on at
 *              most n_cu - n CUs whenever that does not add a round of tiles.  Multi-GPU: leaves room for the RCCL
 *              kernels of the all-reduce that overlaps the next frame, which could otherwise only start at a kernel
 *              boundary.  Results are unchanged. */
#define VFA_FLAG_TERMS_MASK 0xf
#define VFA_FLAG_RESERVED_CUS(n) (((n) & 0xff) << 8)
/* vfa_pool_collapse_relu_sum_f32 only, DIAGNOSTIC: bits 16-27 select a profiling build of the kernel (phase ablations,
 * in-kernel cycle stamps written behind the records in the workspace); its results are meaningless.  tools/ use it. */
#define VFA_FLAG_DEBUG(mask) (((mask) & 0xfff) << 16)
/* Both fused entry points, for TESTS: with ONE view, ONE scale (and one layer) `out` receives the pooled fp32 voxel features (cell,
 * channel) exactly as the kernel's own pooling code forms them in front of the operand split, instead of the map (a diagnostic
 * build of the same source).  That is how the fused kernels' pre-GEMM arithmetic is compared with the reference's. */
#define VFA_FLAG_DUMP_VOX (1 << 30)
/* vfa_pool_collapse_relu_sum_f32 only: the entry point in two calls -- first ROWS_ONLY (the pre-pass over the direct items: needs
 * the box records of the frame, not its work cuts), later SKIP_ROWS (everything else).  Lets a caller that computes the geometry
 * on a second stream wait for vfa_frame_boxes_f32 before the first call and for vfa_frame_cuts_f32 only before the second. */
#define VFA_FLAG_ROWS_ONLY (1 << 28)
#define VFA_FLAG_SKIP_ROWS (1 << 29)
/* `flags` of vfa_project_gather_backward_f32: bit 0 = accumulate into grad_integral (otherwise it is zeroed first);
 * VFA_VOX_KERNEL_DIRECT selects the per-box atomic kernel instead of the LDS-privatised one (C = 256). */
#define VFA_BWD_ACCUMULATE 1

/* Integral image of every feature map: cumsum over W then over H, double accumulator rounded to
 * fp32 at every element (what ATen's CPU cumsum does), written channels-last inside a zero border.
 *                                                               replaces vfa_op.py:110, 172-173 */
int vfa_integral_image_f32(const float *feature, float *integral, int n_views, int C, int Hf, int Wf,
                           void *stream);

/* Producer fusion: the same integral image, of  relu(x * scale[v, c] + shift[v, c])  -- the GroupNorm
 * affine + ReLU of the lateral branch applied while the rows are scanned (two separately rounded fp32 operations), so the
 * lateral map is never materialised.  x (n_views, C, Hf, Wf) = the lateral 1x1-conv output; scale, shift (n_views, C) with
 * scale = gamma * rstd, shift = beta - mean * scale of the channel's group.   replaces vfanet.py:72-74 (norm + ReLU) + vfa_op.py:110 */
int vfa_affine_relu_integral_image_f32(const float *x, const float *scale, const float *shift, float *integral, int n_views, int C,
                                       int Hf, int Wf, void *stream);

/* The integral images of all the feature maps of a frame (one per stride) in ONE launch pair: features[s] (n_views, C, H_s, W_s),
 * integrals[s] (n_views, H_s + 2, W_s + 2, C), feat_hw = {H_0, W_0, H_1, W_1, ...} (host array), n_maps <= 4.  scales / shifts:
 * both NULL (plain maps, vfa_integral_image_f32 of each) or both arrays of n_maps device pointers (n_views, C) (producer fusion,
 * vfa_affine_relu_integral_image_f32 of each).  Results are bit-identical to the per-map entry points; the small maps share the
 * launch of the large one instead of paying a launch pair each.   replaces the three vfa_op.py:110 calls of vfanet.py:76-78 */
/* absmax (ABI v6): NULL, or a HOST array of n_maps device pointers (entries may be NULL), absmax[s] -> vfa_feature_stats_count(n_views,
 * C, H_s) uint32: the call leaves there partial maxima of |feature| (fp32 bit patterns, sign cleared; the maximum of all entries is
 * the map's largest |value| AFTER the affine + ReLU) -- the fused frame kernels scale their fp16 operand split by it.  No
 * initialisation needed, every entry is written. */
size_t vfa_feature_stats_count(int n_views, int C, int Hf);
int vfa_integral_images_f32(const float *const *features, const float *const *scales, const float *const *shifts,
                            float *const *integrals, unsigned *const *absmax, int n_views, int C, int n_maps, const int *feat_hw,
                            void *stream);
/* The same statistic from a finished integral image (n_views, Hf+2, Wf+2, C), for callers that kept no feature map: second
 * differences of the integral image, exact up to its rounding -- enough for a power-of-two scale.  absmax: as above. */
int vfa_integral_absmax_f32(const float *integral, unsigned *absmax, int n_views, int C, int Hf, int Wf, void *stream);

/* The same for CHANNELS-LAST inputs features_hwc[s] (n_views, H_s, W_s, C) -- what vfa_lateral_conv_f32 writes --: no NCHW copy of the
 * lateral convolution exists.  C a multiple of 64.  Bit-identical to vfa_integral_images_f32 of the permuted input. */
int vfa_integral_images_hwc_f32(const float *const *features_hwc, const float *const *scales, const float *const *shifts,
                                float *const *integrals, unsigned *const *absmax, int n_views, int C, int n_maps, const int *feat_hw,
                                void *stream);

/* The lateral branch of one feature scale, as far as the integral image needs it:
 *   y = conv1x1(feat) + bias            feat (n_views, K, Hf, Wf) NCHW (the trunk's output), weight (256, K) = lat.weight.view(256, K)
 *   out_hwc (n_views, Hf, Wf, 256) = y, channels-last;   scale, shift (n_views, 256): the nn.GroupNorm(16, 256) affine of y,
 *   scale = gamma * rstd(group), shift = beta - mean(group) * scale   (biased variance, eps inside the root)
 * so that relu(y * scale + shift) = relu(bn(lat(feat))) -- applied by vfa_integral_images_hwc_f32 while it scans the rows.
 * The product: six bf16 MFMA products of a three-piece split of both operands, fp32 accumulation (x = p0 + p1 + p2 to 2^-25 |x|; what
 * is dropped is <= 2^-23 of a product: the class of an sgemm); the statistics are gathered in the epilogue (double partial sums, added
 * in a fixed order: the same bits on every run).  K a multiple of 32, <= 1024.
 * workspace: vfa_lateral_conv_workspace_bytes(n_views, Hf, Wf).
 *   replaces vfa/model/vfanet.py:37-42, 72-74 (self.lat8/16/32 + self.bn8/16/32; the ReLU rides in the integral image) */
size_t vfa_lateral_conv_workspace_bytes(int n_views, int Hf, int Wf);
int vfa_lateral_conv_f32(const float *feat, const float *weight, const float *bias, const float *gamma, const float *beta, float eps,
                         float *out_hwc, float *scale, float *shift, void *workspace, size_t workspace_bytes, int n_views, int K,
                         int Hf, int Wf, void *stream);
/* The lateral branches of ALL the feature scales of a frame (n_maps <= 3) in three launches instead of three per scale: HOST arrays
 * of n_maps device pointers / values with the meaning of vfa_lateral_conv_f32's arguments, Ks[s], feat_hw = {H_0, W_0, H_1, W_1, ...},
 * workspaces[s] of vfa_lateral_conv_workspace_bytes(n_views, H_s, W_s) bytes each.  Bit-identical to the per-scale calls; the small
 * maps' workgroups fill the tail of the first (largest) map's instead of paying a launch of their own (ABI v7).
 *   replaces vfa/model/vfanet.py:72-74 for the three scales */
int vfa_lateral_convs_f32(int n_maps, const float *const *feats, const float *const *weights, const float *const *biases,
                          const float *const *gammas, const float *const *betas, const float *eps, float *const *outs_hwc,
                          float *const *scales, float *const *shifts, void *const *workspaces, const size_t *workspace_bytes, int n_views,
                          const int *Ks, const int *feat_hw, void *stream);

/* Cube corners -> world units -> 3x4 projection -> normalise/clamp -> 2-D bounding box, area and
 * visibility of every (view, layer, cell).                      replaces vfa_op.py:64-88, 104-106
 * and vfa/utils.py:50-59 (project).
 *   calibs (n_views, 3, 4); grid (n_cells, 3) cell origins in grid units; z_layers (nl);
 *   corner_off (8, 3) in generate_cube order (vfa_op.py:127-133); img_w/img_h = args.image_size[::-1]. */
int vfa_box_params_f32(const float *calibs, const float *grid, const float *z_layers, const float *corner_off,
                       int n_views, int n_cells, int nl, int conv_kind, float img_w, float img_h, int Hf, int Wf,
                       float cmin, float cmax, float *box, float *area, uint8_t *visible, void *stream);

/* Box pooling: four bilinear samples of the integral image at the box corners,
 * vox = (((lt + rb) - rt) - lb) / area * visible.               replaces vfa_op.py:112-120
 * Processes cells [cell_begin, cell_begin + cell_count) of every view so that the caller can bound
 * the size of `vox` on large grids. */
int vfa_gather_f32(const float *integral, const float *box, const float *area, const uint8_t *visible, float *vox,
                   int n_views, int C, int Hf, int Wf, int nl, int n_cells, int cell_begin, int cell_count,
                   int vox_layout, void *stream);

/* Fused form of the two entry points above: each workgroup computes the box parameters of its tile of
 * boxes itself (one thread per box) and stages them in LDS; box/area/visible never touch HBM.
 *                                                               replaces vfa_op.py:64-120 */
int vfa_project_gather_f32(const float *integral, const float *calibs, const float *grid, const float *z_layers,
                           const float *corner_off, float *vox, int n_views, int C, int Hf, int Wf, int nl,
                           int n_cells, int cell_begin, int cell_count, int conv_kind, float img_w, float img_h,
                           float cmin, float cmax, int vox_layout, void *stream);

/* Fused projection + box pooling + collapse product (fp32 MFMA) for C = c_out = 256:
 *   lin[view, cell, :] = sum_layer vox[view, cell, layer, :] . W_layer^T          (no bias, no ReLU)
 * without materialising vox.  weight_t is collapse.weight re-laid as (nl*C, c_out): row layer*C + c, i.e. the
 * transpose of the layer-major weight.  Voxel features are formed exactly as in vfa_project_gather_f32; the
 * product is a k-ordered fp32 fmaf chain (within the collapse tolerance, not bitwise -- no GEMM order is).
 * Returns VFA_ERR_UNSUPPORTED for other channel counts.               replaces vfa_op.py:64-123 */
int vfa_project_collapse_f32(const float *integral, const float *calibs, const float *grid, const float *z_layers,
                             const float *corner_off, const float *weight_t, float *lin, int n_views, int C, int Hf, int Wf,
                             int nl, int n_cells, int c_out, int conv_kind, float img_w, float img_h, float cmin, float cmax,
                             void *stream);

/* Two-kernel form of vfa_project_gather_f32: a records kernel writes one 128-byte record per box into `workspace`
 * (vfa_gather_workspace_bytes() bytes, caller-owned scratch) and the pooling waves fetch them with scalar loads.
 * Same results, bit for bit. */
size_t vfa_gather_workspace_bytes(int n_views, int nl, int cell_count);
int vfa_project_gather_ws_f32(const float *integral, const float *calibs, const float *grid, const float *z_layers,
                              const float *corner_off, float *vox, void *workspace, size_t workspace_bytes, int n_views,
                              int C, int Hf, int Wf, int nl, int n_cells, int cell_begin, int cell_count, int conv_kind,
                              float img_w, float img_h, float cmin, float cmax, int vox_layout, void *stream);

/* Backward of vfa_project_gather_f32 with respect to the integral images (training: the reference back-propagates
 * through the path with autograd, trainer.py:41; calib and grid carry no gradient).  grad_vox is layer-major
 * (n_views, cell_count, nl*C); grad_integral (n_views, Hf+2, Wf+2, C) is zeroed first unless VFA_BWD_ACCUMULATE is set in `flags`.
 * Scatter-add with float atomics: results are not bit-reproducible run to run. */
int vfa_project_gather_backward_f32(const float *grad_vox, const float *calibs, const float *grid, const float *z_layers,
                                    const float *corner_off, float *grad_integral, int n_views, int C, int Hf, int Wf,
                                    int nl, int n_cells, int cell_begin, int cell_count, int conv_kind, float img_w,
                                    float img_h, float cmin, float cmax, int flags, void *stream);
/* The same, told the width of the ground grid (ABI v9): grid_w = cells per row of the (rows, grid_w) grid the n_cells cells come from,
 * row-major (n_cells a multiple of it; 0 = unknown: as above).  The scatter then works on patches of 4 x 8 cells instead of cells in a
 * line: neighbouring boxes share taps in both directions, and the kernel runs at the rate of its atomic rows (bench frame: 1.7 / 0.9 /
 * 0.5 distinct taps per box on strides 8 / 16 / 32 against 4.5 / 2.7 / 1.7).  Same sums up to the order of the float atomics. */
int vfa_project_gather_backward_grid_f32(const float *grad_vox, const float *calibs, const float *grid, const float *z_layers,
                                    const float *corner_off, float *grad_integral, int n_views, int C, int Hf, int Wf,
                                    int nl, int n_cells, int cell_begin, int cell_count, int grid_w, int conv_kind, float img_w,
                                    float img_h, float cmin, float cmax, int flags, void *stream);

/* Backward of vfa_integral_image_f32: reverse cumsum over H (in place on grad_integral, which is destroyed) then over
 * W, written as NCHW grad_feature (n_views, C, Hf, Wf). */
int vfa_integral_image_backward_f32(float *grad_integral, float *grad_feature, int n_views, int C, int Hf, int Wf,
                                    void *stream);

/* Backward of the two epilogues below with respect to `lin` and `bias` (per scale):
 *   grad_lin[v] = grad * (lin[v] + bias > 0);  grad_bias = column sums of grad_lin (zeroed first; NULL to skip).
 * grad (M, N); lin, grad_lin (n_views, M, N).  Needs 4 | N and N | 1024, otherwise VFA_ERR_UNSUPPORTED. */
int vfa_relu_mask_backward_f32(const float *grad, const float *lin, const float *bias, float *grad_lin, float *grad_bias,
                               int n_views, size_t M, int N, void *stream);

/* Epilogue of `collapse` for one VFA call batch:  out = (accumulate ? out : 0) + sum_v relu(lin[v] + bias)
 * with views added in index order.            replaces vfa_op.py:124 (ReLU) and vfanet.py:82 (view sum)
 *   lin (n_views, M, N) = vox . W^T without bias; bias (N) or NULL; out (M, N). */
int vfa_bias_relu_accumulate_f32(const float *lin, const float *bias, float *out, int n_views, size_t M, int N,
                                 int accumulate, void *stream);

/* Scale sum + view sum in the reference's order:
 *   ortho = sum_v ((relu(lin8[v]+b8) + relu(lin16[v]+b16)) + relu(lin32[v]+b32))
 *                                                               replaces vfanet.py:79 and :82 */
int vfa_scale_view_sum_f32(const float *lin8, const float *lin16, const float *lin32, const float *bias8,
                           const float *bias16, const float *bias32, float *ortho, int n_views, size_t M, int N,
                           int accumulate, void *stream);

/* `collapse` + ReLU + view sum in one MFMA kernel, for K = N = 256 (single-layer grids, C = 256):
 *   out[m, :] = (accumulate ? out[m, :] : 0) + sum_v relu(vox[v, m, :] . weight^T + bias)
 *                                              replaces vfa_op.py:121-124 (Linear, ReLU) and vfanet.py:82 (view sum)
 *   vox (n_views, M, K) layer-major; weight (N, K) = collapse.weight (layer-major columns; identical to the reference's
 *   for one layer); bias (N) or NULL; out (M, N).
 * fp32 in, fp32 out, fp32 accumulation; each fp32 product is formed from `terms` bf16 MFMA products of an exact
 * hi/lo split of both operands (`flags` bits 0-3: 3 = default when 0 is passed, 4 adds lo*lo): error ~5e-6 of max|out| (the fp32 library
 * GEMM: 1e-6), inside the 1e-4 / 1e-5 max tolerance of the path; not bitwise -- no GEMM order is.  Inf inputs give
 * NaN (Inf - Inf in the split).  Returns VFA_ERR_UNSUPPORTED for other K, N: use a GEMM + the epilogues above. */
int vfa_collapse_relu_sum_f32(const float *vox, const float *weight, const float *bias, float *out, int n_views, size_t M,
                              int K, int N, int accumulate, int flags, void *stream);

/* The `collapse` product alone for any layer count:  lin[m, :] = vox[m, :] . weight^T,  m < M = n_views * cells,
 * K = nl * C a multiple of 128, N = 256 (other shapes: VFA_ERR_UNSUPPORTED, use a library GEMM).  No bias, no ReLU: the
 * two epilogue entry points above add them while summing views.         replaces vfa_op.py:121-123 (nn.Linear)
 * Same arithmetic as vfa_collapse_relu_sum_f32 (bf16-split MFMA, fp32 accumulation, same `flags`), as a K-looped
 * 128-row tile GEMM.  `workspace`: vfa_collapse_gemm_workspace_bytes(K, N) bytes of caller-owned scratch (the weight,
 * split into bf16 planes in MFMA fragment order, rewritten by every call). */
size_t vfa_collapse_gemm_workspace_bytes(int K, int N);
int vfa_collapse_gemm_f32(const float *vox, const float *weight, float *lin, void *workspace, size_t workspace_bytes, size_t M,
                          int K, int N, int flags, void *stream);

/* Training backward behind the fused forward (which kept no pre-activations): the same product, recomputed, with the ReLU mask as
 * its epilogue --  grad_lin (n_views, cells, 256) = (vox . weight^T + bias > 0) ? grad_out[cell] : 0,  grad_bias (256) += column
 * sums (atomics; may be NULL).  `lin` is never written.  vox (n_views * cells, K), weight (256, K), grad_out (cells, 256).
 *   replaces the autograd of vfa_op.py:123-124 (Linear + ReLU) for the gradient w.r.t. the pre-activation; trainer.py:41 */
int vfa_collapse_gemm_relu_backward_f32(const float *vox, const float *weight, const float *bias, const float *grad_out,
                                        float *grad_lin, float *grad_bias, void *workspace, size_t workspace_bytes, int n_views,
                                        size_t cells, int K, int N, int flags, void *stream);
/* The same with the product of the FUSED FRAME KERNELS -- two fp16 pieces per operand under the frame's power-of-two scales (ABI v8) --
 * so that the recomputed pre-activation, and with it the ReLU mask of the backward, is the forward's bit for bit: feat_absmax /
 * absmax_count = the feature statistics of this scale (vfa_integral_images_f32: what the frame kernel reduced to its 2^ea), row_shift
 * = the sliver shift of every row of the product (n_views * cells bytes: vfa_sliver_shifts_u8, its per-cell form repeated per view;
 * NULL: none), tile_any = one byte per 128 rows, non-zero where a row of that block has a shift (NULL: every row is looked up).  Operands,
 * scales, accumulator start (bias 2^(ea+ew-shift)) and the order of the three MFMA products are the forward's.  flags: reserved CUs. */
int vfa_collapse_gemm_relu_backward_f16_f32(const float *vox, const float *weight, const float *bias, const float *grad_out,
                                            float *grad_lin, float *grad_bias, void *workspace, size_t workspace_bytes, int n_views,
                                            size_t cells, int K, int N, const unsigned *feat_absmax, int absmax_count,
                                            const unsigned char *row_shift, const unsigned char *tile_any, int flags, void *stream);
/* The sliver shifts of a frame, per output row of that product (vfa_geom.h: sliver_shift): the binary places the
 * fp16 operand split of the frame kernels gave up for the noisiest visible box of the row's item.  per_item = 1: the serial kernel's
 * items (view, 8 x 4-cell tile) of ONE scale (feature map Hf x Wf), single-layer grids -> shift (n_views, L * W); per_item = 0: the
 * pipelined kernel's (tile, scale) over all views and layers -> shift (L * W), scratch = vfa_sliver_shifts_scratch_bytes(L, W) bytes.
 *   restates vfa_op.py:64-106 for the training backward; same device code as the geometry passes */
size_t vfa_sliver_shifts_scratch_bytes(int L, int W);
int vfa_sliver_shifts_u8(const float *calibs, const float *grid, const float *z_layers, int n_layers, const float *corner_off, int n_views,
                         int L, int W, int conv_kind, float img_w, float img_h, float cmin, float cmax, int Hf, int Wf, int per_item,
                         unsigned char *shift, void *scratch, size_t scratch_bytes, void *stream);

/* ---- single-layer grids (nl = 1), C = c_out = 256, inference: the whole frame in two launches ------------------------------
 *
 * vfa_frame_records_f32: geometry ONCE per frame.  Every (view, cell) cube is projected once (vfa_op.py:64-88,
 * utils.py:50-59) and turned, per feature scale, into a 96-byte box record (16 bilinear tap weights, 1 / area, visibility,
 * tap coordinates: vfa_op.py:104-106 and the set-up of the four F.grid_sample calls :112-115) plus, per (view, 8 x 4-cell
 * tile, scale), the window of the integral image holding the tile's taps; `weights[k]` (collapse.weight of scale k, (256,
 * 256) fp32, may be NULL as a whole to skip) is split into bf16 hi / lo planes in MFMA fragment order.  All of it goes to
 * `workspace` (vfa_frame_workspace_bytes() bytes of caller-owned device memory, valid until the next call that uses it).
 *   n_scales in 1..3; feat_hw = HOST array {Hf0, Wf0, Hf1, Wf1, ...}; weights = HOST array of n_scales device pointers.
 *   grid (L * W, 3) row-major; z_layers[0] is the single layer.  n_views <= 32 (else VFA_ERR_UNSUPPORTED).
 *
 * vfa_pool_collapse_relu_sum_f32: box pooling + Linear + bias + ReLU + view sum + scale sum in ONE persistent kernel,
 *   out[cell, :] = (accumulate ? out[cell, :] : 0) + sum_scale sum_view relu(vox_{scale,view}[cell, :] . W_scale^T + b_scale)
 *                                                replaces vfa_op.py:112-125 and vfanet.py:79, 82 for every scale and camera
 * The voxel features never reach HBM: per (tile, scale, view) the tap window is brought into LDS by LDS-DMA, the 32 boxes are
 * pooled with the reference's FMA chains into bf16 hi / lo planes and multiplied on the matrix cores (same bf16-split
 * arithmetic and `flags` as vfa_collapse_relu_sum_f32); the quotient is v * RN(1 / area) instead of a division (<= 1.5 ulp,
 * far below the split).  Results: within the path's post-GEMM tolerance, not bitwise.
 *   integrals / biases = HOST arrays of n_scales device pointers ((n_views, Hf+2, Wf+2, 256) each / (256) or NULL);
 *   workspace = what vfa_frame_records_f32 filled for the same (n_views, L, W, n_scales, feat_hw). */
size_t vfa_frame_workspace_bytes(int n_views, int L, int W, int n_scales);
/* (That is the recommended size.  The last region holds the pooled rows of the direct items -- tiles whose tap window exceeds LDS
 * -- 32 KiB per slot; any size from offsets[21] of vfa_frame_workspace_layout upwards is accepted by the three entry points
 * below, which derive the number of row slots from the size they are given: pass the SAME size to all of them.  Direct items
 * without a slot take a slower second launch.) */
/* Where things are inside that workspace (tests and tools/ read the records back): offsets[25]: offsets[5 k + {0..4}] = live-view
 * masks, direct-item masks, tile headers (32 B), box records (96 B), split weight of scale k; offsets[15] diagnostics, offsets[16]
 * total bytes; offsets[17 + k] = masks of the direct items without a row slot, offsets[20] the direct-item counter,
 * offsets[21] the pooled rows of the direct items (slot x 32 boxes x 256 fp32), offsets[22] = the number of row slots,
 * offsets[23] / [24] the work cuts (tile, rank of the first item inside it), tiles[3] + 1 of them;
 * tiles[4] = {tile rows, tile columns, tap-window capacity in slots, number of work pieces}.  Tiles are 4 x 8 cells. */
int vfa_frame_workspace_layout(int n_views, int L, int W, int n_scales, size_t *offsets, int *tiles);
int vfa_frame_records_f32(const float *calibs, const float *grid, const float *z_layers, const float *corner_off, int n_views, int L,
                          int W, int conv_kind, float img_w, float img_h, float cmin, float cmax, int n_scales,
                          const int *feat_hw, const float *const *weights, int flags, void *workspace, size_t workspace_bytes,
                          void *stream);
/* The same in two calls (vfa_frame_records_f32 = boxes, then cuts): vfa_frame_boxes_f32 projects the boxes and writes records,
 * headers, masks and the list of direct items; vfa_frame_cuts_f32 forms the work cuts of the persistent kernel from them and
 * splits the collapse weights (`weights` NULL: left as they are in the workspace).                 replaces vfa_op.py:64-106 */
int vfa_frame_boxes_f32(const float *calibs, const float *grid, const float *z_layers, const float *corner_off, int n_views, int L,
                        int W, int conv_kind, float img_w, float img_h, float cmin, float cmax, int n_scales,
                        const int *feat_hw, void *workspace, size_t workspace_bytes, void *stream);
int vfa_frame_cuts_f32(int n_views, int L, int W, int n_scales, const float *const *weights, int flags, void *workspace,
                       size_t workspace_bytes, void *stream);
/* (flags, ABI v6: VFA_FLAG_TERMS_MASK -- the arithmetic the weights are split for; pass what goes to vfa_pool_collapse_relu_sum_f32.
 * vfa_pool_collapse_relu_sum_f32 / vfa_pipe_collapse_relu_sum_f32 take `feat_absmax`: NULL, or a HOST array of n_scales device
 * pointers to the statistics vfa_integral_images_f32 left for the SAME integral images (entries may be NULL).  With the default
 * fp16 arithmetic a scale without statistics costs one extra pass over its integral image inside the call.) */
/* Box pooling alone from the same workspace: vox (n_views, L * W, 256) fp32, BIT-IDENTICAL to vfa_project_gather_f32 (layer-major,
 * nl = 1): four bilinear samples of the integral image of scale `scale`, (((lt + rb) - rt) - lb) / area * visible.
 *                                                                                            replaces vfa_op.py:112-120
 * One 64-channel quarter of one (view, tile) per workgroup; the tile's tap window comes in by LDS-DMA, so every distinct tap is
 * read from L2 / HBM once; bound by the HBM write of `vox`.  Hf, Wf must be the sizes the records of `scale` were built for. */
int vfa_pool_windows_f32(const float *integral, const void *workspace, size_t workspace_bytes, float *vox, int n_views, int L, int W,
                         int n_scales, int scale, int Hf, int Wf, void *stream);
int vfa_pool_collapse_relu_sum_f32(const float *const *integrals, const unsigned *const *feat_absmax, const float *const *biases,
                                   const void *workspace, size_t workspace_bytes, float *out, int n_views, int L, int W, int n_scales,
                                   const int *feat_hw, int accumulate, int flags, void *stream);

/* ---- the frame as a producer / consumer pipeline, any number of z-layers (vfa_pipe.hip) ----------------------------------------
 *
 * The inference hot path for C = 256 and ANY K = n_layers * 256 (the reference builds collapse = Linear(C * nl -> C),
 * vfa_op.py:50-59, and every shipped config has nl > 1: vfa/config.py:22-24, 49-52, 77-80):
 *
 *   vfa_pipe_boxes_f32     geometry once per frame: every (view, cell, LAYER) cube projected once, a 48-byte box record per scale
 *                          and a 32-byte tap-window header per (tile, layer, view, scale)        replaces vfa_op.py:64-106
 *   vfa_pipe_cuts_f32      cost-balanced work cuts + collapse.weight of every scale as 16-bit MFMA fragments (fp16 hi / lo under a
 *                          power-of-two scale by default, bf16 pieces for terms 3 / 4 / 6).
 *                          weights[k]: (256, 256 * n_layers) fp32 in the REFERENCE layout, column = c * n_layers + layer
 *                          (vfa_op.py:59, :120) -- no host-side permutation
 *   vfa_pipe_records_f32   both of the above
 *   vfa_pipe_collapse_relu_sum_f32
 *                          out (L * W, 256) (+)= sum_scale sum_view relu(vox . W^T + b): one persistent kernel, 16 waves per
 *                          CU -- eight pool boxes (the reference's exact fp32 FMA chains and its correctly rounded quotient)
 *                          while eight multiply the previous 64 rows x 64 channels on the matrix cores (default: two fp16 pieces
 *                          per operand under a power-of-two scale, three products, fp32 accumulation: the width of the reference's
 *                          fp32 nn.Linear; vfa_split.h).  The accumulators of a GROUP of four sub-tiles (a sub-tile = one live
 *                          view of one (tile, scale)) stay in registers across all layers: `relu` follows the whole
 *                          K = n_layers * 256.  Round 6: a group takes its sub-tiles from a RUN of 1, 2 or 4 consecutive tiles
 *                          (vfa_pipe_seq.h: run_tiles_of -- chosen per frame from n_views, the tile count and n_layers), so a frame
 *                          of one or two views (a rank's share of a camera-sharded rig) fills its groups from four tiles instead of
 *                          streaming the weight for one view's 32 rows; the voxel features never reach HBM
 *                                                                                replaces vfa_op.py:110-125, vfanet.py:79, 82
 *
 * workspace: caller-owned, vfa_pipe_workspace_bytes(); the geometry calls fill it, the kernel reads it and uses three areas of its
 * own: the CONTRIBUTIONS of a workgroup to the tiles of the run it is in (one 32-row x 256-column buffer per (tile of the run,
 * scale, group): plain stores at the end of a group, read back and added in (scale, group) order when the workgroup leaves the
 * run), the PARTS of a run cut between workgroups (every matrix wave stores its 32 columns, draws a ticket of its own, and the wave
 * that arrives last adds the parts in workgroup order and writes the tiles: nobody waits, no barrier), and the balance state.
 * integrals[k]: zero-bordered channels-last (n_views, Hf+2, Wf+2, 256).  n_views <= 32.  flags: VFA_FLAG_TERMS_MASK |
 * VFA_FLAG_RESERVED_CUS(n) | VFA_FLAG_DEBUG(mask) | VFA_FLAG_DUMP_VOX.  Terms: 0 / 2 = two fp16 pieces per operand, three products
 * (1e-7 ... 3e-7 normwise against float64: an fp32 sgemm's error); 3 = two bf16 pieces, three products (16-bit operands, ~3e-6 of
 * max|out|, inside the path's 1e-5 but narrower than the reference); 4 adds lo.lo; 6 = THREE bf16 pieces per operand (x = p0 + p1 +
 * p2 to 2^-25) and the six products down to 2^-16 of the largest, at twice the matrix work.  The geometry calls take the same terms
 * in `flags` (the weight fragments are split for one arithmetic, and the three-piece variant has smaller LDS tap windows; pass the
 * value that goes to vfa_pipe_collapse_relu_sum_f32: a launch that asks for the other arithmetic returns a map of NaNs).  Results are
 * deterministic for a given geometry, launch size and balance state; the association of the view / scale sum differs from
 * vfa_pool_collapse_relu_sum_f32's (inside the post-GEMM tolerance).
 *
 *   vfa_pipe_balance_f32   (ABI v5) mode 1: from the work cuts the geometry call has just left (estimated cost per group), the BOUNDS
 *                          of the workgroups' shares that minimise the heaviest share, for a launch of the size `reserved_cus`
 *                          gives; the frame kernel uses them for every later frame whose cuts have the same total cost and whose
 *                          launch has the same number of workgroups, and falls back to the uniform split otherwise (cameras and grid
 *                          of a frame stream stand still).  mode 0 clears the state: call it once on a fresh workspace (the geometry
 *                          calls never touch the state).  Deterministic.  Results stay inside the path's tolerance but are NOT
 *                          bitwise the unbalanced ones: a run cut between two workgroups is summed in another association when the
 *                          cut moves.  offsets[18] of vfa_pipe_workspace_layout: the state (8 KiB: int bounds[513], launch size,
 *                          cost signature; at byte 4096 a u64 cycle count per workgroup of the last launch: diagnostics), for
 *                          callers that keep one state per band of a banded frame.  offsets must hold 19 entries since ABI v5, 22
 *                          since ABI v8: offsets[19 + k] = the sliver shifts of scale k, one unsigned per tile (binary places the
 *                          fp16 operand split of the frame kernel gives up for the noisiest visible box of (tile, scale): 0
 *                          everywhere but next to boxes of ~1e-5 pixels).  VFA_AMD_PIPE_RT = 1 | 2 | 4 in the environment overrides
 *                          the run length (geometry call and frame kernel read it alike): measurements only. */
size_t vfa_pipe_workspace_bytes(int n_views, int L, int W, int n_layers, int n_scales);
int vfa_pipe_workspace_layout(int n_views, int L, int W, int n_layers, int n_scales, size_t *offsets, int *tiles);
int vfa_pipe_boxes_f32(const float *calibs, const float *grid, const float *z_layers, int n_layers, const float *corner_off, int n_views,
                       int L, int W, int conv_kind, float img_w, float img_h, float cmin, float cmax, int n_scales, const int *feat_hw,
                       int flags, void *workspace, size_t workspace_bytes, void *stream);
int vfa_pipe_cuts_f32(int n_views, int L, int W, int n_layers, int n_scales, const float *const *weights, int flags, void *workspace,
                      size_t workspace_bytes, void *stream);
int vfa_pipe_records_f32(const float *calibs, const float *grid, const float *z_layers, int n_layers, const float *corner_off,
                         int n_views, int L, int W, int conv_kind, float img_w, float img_h, float cmin, float cmax, int n_scales,
                         const int *feat_hw, const float *const *weights, int flags, void *workspace, size_t workspace_bytes, void *stream);
int vfa_pipe_collapse_relu_sum_f32(const float *const *integrals, const unsigned *const *feat_absmax, const float *const *biases,
                                   void *workspace, size_t workspace_bytes, float *out, int n_views, int L, int W, int n_layers,
                                   int n_scales, const int *feat_hw, int accumulate, int flags, void *stream);
int vfa_pipe_balance_f32(int n_views, int L, int W, int n_layers, int n_scales, int reserved_cus, int mode, void *workspace,
                         size_t workspace_bytes, void *stream);

/* The weight gradient of `collapse` (training):  g_w (256, K) (+)= g_lin^T . vox  with g_lin (rows, 256) the
 * masked output gradient (vfa_relu_mask_backward_f32 / vfa_collapse_gemm_relu_backward_f32) and vox (rows, K) the voxel features of
 * the same rows, K = n_layers * 256 in the column order of the weight passed to the forward.  Six bf16 MFMA products of a three-piece
 * split of both operands, fp32 accumulation (sgemm class); per-workgroup partial sums added in a fixed order (the same bits on every
 * run).  workspace: vfa_grad_weight_workspace_bytes(rows, K).  K a multiple of 256 (ABI v7).
 *   replaces the autograd of nn.Linear's weight, vfa/model/vfa_op.py:59, :123 under vfa/trainer.py:41 */
size_t vfa_grad_weight_workspace_bytes(long long rows, int K);
int vfa_grad_weight_f32(const float *g_lin, const float *vox, float *g_w, long long rows, int K, int accumulate, void *workspace,
                        size_t workspace_bytes, void *stream);
/* ... and of its input:  g_vox (rows, K) = g_lin (rows, 256) . w (256, K), w = the weight in the column order of the forward.  The same
 * six-product arithmetic; workspace: vfa_grad_input_workspace_bytes(K) (the weight as three bf16 planes in MFMA fragment order,
 * rebuilt by every call).  K a multiple of 256, g_lin 16-byte aligned (ABI v7).
 *   replaces the autograd of nn.Linear's input, vfa/model/vfa_op.py:123 under vfa/trainer.py:41 */
size_t vfa_grad_input_workspace_bytes(int K);
int vfa_grad_input_f32(const float *g_lin, const float *w, float *g_vox, long long rows, int K, void *workspace, size_t workspace_bytes,
                       void *stream);

/* ---- consumers of the path ---------------------------------------------------------------------------
 *
 * vfa_sort_vertices_f32: anticlockwise order of the valid vertices of n convex polygons per batch entry (rectangle x rectangle
 * intersections of the AP/AOS metric).                       replaces vfa/evaluation/pyeval/cuda_op/sort_vert_kernel.cu:42-140
 *   vertices (b, n, m, 2) fp32 normalised around the polygon centre; mask (b, n, m) bytes (1 = valid candidate, the first 8 are box
 *   corners, the rest edge intersections); num_valid (b, n) int32; idx (b, n, 9) int32 out: the sorted indices, the first one
 *   repeated, then padded with an invalid intersection index.  m > 8.  Same selection rule, comparison and corner cases as the
 *   reference kernel; one lane per polygon.
 *
 * vfa_bev_nms_f32: conf (L, W) = sigmoid(heatmap) where it equals its 5 x 5 max-pool (padding 2), else 0.
 *                                                            replaces vfa/data/encoder.py:230-232 + the sigmoid of :238 / :278 */
int vfa_sort_vertices_f32(const float *vertices, const uint8_t *mask, const int *num_valid, int *idx, int b, int n, int m, void *stream);
int vfa_bev_nms_f32(const float *heatmap, float *conf, int L, int W, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VFA_HIP_H */
