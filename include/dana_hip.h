/* libdana_hip.so -- C ABI of the MI355X (gfx950) DAnA forward hot path.
 *
 * Every entry point takes raw DEVICE pointers, explicit sizes/strides, scalars and a HIP stream
 * (passed as void*); returns 0 on success or a negative DANA_ERR_* code, with the message
 * available from dana_last_error() (thread-local). No entry point allocates device memory or
 * synchronises with the host: scratch comes from the caller (`*_workspace_bytes`), launches go
 * to the caller's stream, so the functions are re-entrant across threads/devices (the reference
 * is driven from one Python thread per device under nn.DataParallel, train.py:104-105).
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference
 * repository root). INTEGRATION.md shows the binding a reference maintainer would add.
 */
#ifndef DANA_HIP_H
#define DANA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dana_stream_t; /* hipStream_t */

#define DANA_LAYOUT_NCHW 0
#define DANA_LAYOUT_NHWC 1

/* epilogue flags of dana_conv2d_nhwc / dana_gemm_nt */
#define DANA_EPI_RELU 1
#define DANA_CONV_STEM7 2 /* 7x7/2 stem over NHWC4 input, weight packed [cout][7][8][4] */
/* the weight / `b` / `u` argument points at dana_split_weight's output instead of fp32 rows: three bf16 planes per
 * 16-wide K-step, [kp / 16][3][n][16], of the SAME values (exact split, kp = k rounded up to 16; dana_gemm_nt: ldb = kp,
 * batch_b = bf16 elements between the slices' first K-steps). Split kernel only (dana_set_mfma_mode != 0);
 * results are bit-identical to the fp32-weight call, the K loop just does not repeat the split per tile and step. */
#define DANA_W_SPLIT3 256
/* dana_gemm_nt only: `a` holds the activation rows as split planes as well, [lda / 16][3][m][16] with lda = k rounded up to
 * 16 and batch_a in bf16 elements (the layout of dana_split_weight with n = m): planes x planes form of igemm_dma_kernel -- both
 * operands by LDS-DMA, no split and no staging registers in the K loop. Needs DANA_W_SPLIT3. Bit-identical to the fp32-rows
 * call on the values the planes were split from. (Measured in round 5, profiles/r5_activation_planes.md: the GEMM itself is
 * 6-19 % faster, a PRODUCER that writes planes pays 1.5x the bytes -- not used on the model's path.) */
#define DANA_A_SPLIT3 512

const char* dana_last_error(void);
int dana_abi_version(void);

/* How the fp32 contractions (every conv / GEMM below) use the matrix cores. 1 (default; DANA_MFMA_SPLIT overrides the
 * initial value): each fp32 operand is split EXACTLY into three bf16 numbers and the six products of weight >= 2^-16
 * run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (error vs an fp64 contraction equal to the f32 kernel's,
 * tests/test_gpu_contractions.py); 0: v_mfma_f32_32x32x2_f32. (Forced tile shapes, the LDS epilogue form, the sort dispatch and the per-block trace are
 * debug switches: include/dana_hip_debug.h.)
 * Replaces nothing in the reference: cuDNN picks its own algorithm / math mode there (lib/model/framework/resnet.py). */
int dana_set_mfma_mode(int mode);
int dana_get_mfma_mode(void);

/* ---- native operators: lib/model/csrc/vision.cpp:7-13 (module `model._C`) ------------------- */

/* ROIAlign_forward: lib/model/csrc/ROIAlign.h:11-25 -> cuda/ROIAlign_cuda.cu:257-305.
 * rois[num_rois][5] = (batch_idx, x1, y1, x2, y2) in image pixels. layout NCHW: input
 * [B][C][H][W] -> output [R][C][PH][PW] (the reference contract). layout NHWC: pixel (b,y,x) at
 * input + ((b*H+y)*W+x)*in_pix_stride (stride 0 => C) -> output [R][PH*PW][out_pix_stride];
 * optional output2 = output + add2[PH*PW][C] written in the same pass (fused positional
 * encoding, dana.py:258-259).
 * Arithmetic: the reference's per-sample loop in the reference's order -- bit-identical to cpu/ROIAlign_cpu.cpp in both
 * layouts. */
int dana_roi_align_forward(const float* input, const float* rois, float* output, int batch, int channels,
                           int height, int width, int num_rois, float spatial_scale, int pooled_h,
                           int pooled_w, int sampling_ratio, int layout, long in_pix_stride,
                           long out_pix_stride, float* output2, const float* add2, long out2_pix_stride,
                           dana_stream_t stream);

/* ROIAlign_backward: lib/model/csrc/ROIAlign.h:27-45 -> cuda/ROIAlign_cuda.cu:308-346.
 * grad_in [B][C][H][W] (NCHW) or [B][H][W][C] (NHWC) is completely (over)written: NCHW zero-fills it and scatters with
 * fp32 atomics like the reference (unordered sum); NHWC (channels % 4 == 0, pooled sides <= 8) gathers per feature cell
 * in a fixed (roi, bin) order -- no atomics, bit-reproducible. */
int dana_roi_align_backward(const float* grad_out, const float* rois, float* grad_in, int batch, int channels,
                            int height, int width, int num_rois, float spatial_scale, int pooled_h,
                            int pooled_w, int sampling_ratio, int layout, dana_stream_t stream);

/* ROIPool_forward / ROIPool_backward: lib/model/csrc/ROIPool.h:11-46 -> cuda/ROIPool_cuda.cu:111-202. NCHW. */
int dana_roi_pool_forward(const float* input, const float* rois, float* output, int* argmax, int batch,
                          int channels, int height, int width, int num_rois, float spatial_scale, int pooled_h,
                          int pooled_w, dana_stream_t stream);
int dana_roi_pool_backward(const float* grad_out, const int* argmax, const float* rois, float* grad_in,
                           int batch, int channels, int height, int width, int num_rois, int pooled_h,
                           int pooled_w, dana_stream_t stream);

/* nms: lib/model/csrc/nms.h:10-28 -> cuda/nms.cu:70-131 (nms_cuda) / cpu/nms_cpu.cpp:5-75.
 * boxes[problems][n][4] must already be in descending-score order (the reference sorts inside
 * nms_cuda, nms.cu:73-75; here the sort is dana_sort_desc). keep[p][0..num_keep[p]) receives
 * the kept POSITIONS, ascending; at most max_keep (<=0: all). inclusive=0: suppress IoU > thr
 * (nms.cu:60); inclusive=1: IoU >= thr (nms_cpu.cpp:60).
 * With max_keep < n the mask is filled and scanned in column bands (the first covers ~2.5 x max_keep boxes; the later
 * ones leave at once when max_keep is reached): the result is the same greedy NMS truncated at max_keep, only the
 * work differs. A band of <= 128 column blocks (every band of the proposal layer's problems) is scanned by a barrier-free
 * dataflow of one workgroup's waves -- resolver, word loaders, row workers over LDS flags --, a wider one by the
 * resolver + workers kernel with one barrier per 64-box block. */
size_t dana_nms_workspace_bytes(int n, int problems);
int dana_nms(const float* boxes, int n, int problems, float thr, int inclusive, int max_keep, int* keep,
             int keep_stride, int* num_keep, void* workspace, size_t workspace_bytes, dana_stream_t stream);

/* ---- RPN proposal path: lib/model/rpn/proposal_layer.py:49-190 ------------------------------- */

/* Anchor grid + bbox_transform_inv (bbox_transform.py:77-103) + clip_boxes (:125-133) + fg score.
 * cls element (b, ch, k=h*W+w) at cls + b*cls_sb + ch*cls_sc + k*cls_sp (2A channels: bg a, fg A+a);
 * cls_is_prob=0 applies the pair softmax of rpn.py:67-69. bbox likewise with 4A channels.
 * proposals[B][H*W*A][4], scores[B][H*W*A], index = k*A + a (proposal_layer.py:98-103). */
int dana_rpn_decode(const float* cls, long cls_sb, long cls_sc, long cls_sp, int cls_is_prob, const float* bbox,
                    long bbox_sb, long bbox_sc, long bbox_sp, const float* im_info, const float* base_anchors,
                    int B, int A, int H, int W, int feat_stride, float* proposals, float* scores,
                    dana_stream_t stream);

/* torch.sort(scores, 1, True) (proposal_layer.py:135): stable, per row. sorted_scores may be NULL. (= dana_topk_desc with
 * topn = n) */
size_t dana_sort_desc_workspace_bytes(int B, int n);
int dana_sort_desc(const float* scores, int B, int n, int* order, float* sorted_scores, void* workspace,
                   size_t workspace_bytes, dana_stream_t stream);
/* proposal_layer.py:135-150 as one operation: order[b][0..min(topn, n)) = the indices of row b's `topn` largest scores in
 * descending score order, equal scores in ascending index order -- the first topn entries of torch.sort(scores, 1, True)
 * (row stride of order / sorted_scores: order_stride >= min(topn, n); sorted_scores may be NULL). Two hand-written sorts, no
 * library: rows of <= 4 096 scores take ONE workgroup per row (radix select in registers + stable LDS radix sort, one
 * launch: 19 us at n = 300); longer rows a sample sort over the whole chip (splitters from 2 048 sampled keys, classify,
 * scatter, per-bucket rank: four launches, `workspace` of dana_topk_desc_workspace_bytes; 4 x 21 546 -> 12 000 in 45 us
 * where the single-workgroup kernel takes 122 us and rocPRIM's device-wide radix sort, used through round 4, took 88 us). */
size_t dana_topk_desc_workspace_bytes(int B, int n, int topn);
int dana_topk_desc(const float* scores, int B, int n, int topn, int* order, int order_stride, float* sorted_scores,
                   void* workspace, size_t workspace_bytes, dana_stream_t stream);

/* proposals_single[order_single[:topn]] (proposal_layer.py:148-151) */
int dana_gather_boxes(const float* src, const int* order, int B, int n, int order_stride, int topn, float* dst,
                      dana_stream_t stream);

/* output[i,:,0]=i; output[i,:num,1:]=kept boxes; zero padding (proposal_layer.py:136,183-188) */
int dana_rois_assemble(const float* sorted_boxes, const int* keep, const int* num_keep, int B, int topn,
                       int keep_stride, int post_n, float* rois, dana_stream_t stream);

/* _ProposalLayer.forward as one call: decode -> sort -> top-N -> NMS -> rois[B][post_nms_topn][5]. */
size_t dana_proposal_layer_workspace_bytes(int B, int A, int H, int W, int pre_nms_topn, int post_nms_topn);
int dana_proposal_layer(const float* cls, long cls_sb, long cls_sc, long cls_sp, int cls_is_prob, const float* bbox,
                        long bbox_sb, long bbox_sc, long bbox_sp, const float* im_info, const float* base_anchors,
                        int B, int A, int H, int W, int feat_stride, int pre_nms_topn, int post_nms_topn,
                        float nms_thresh, int nms_inclusive, float* rois, void* workspace, size_t workspace_bytes,
                        dana_stream_t stream);

/* Inference post-processing for one image (inference.py:106-140, utils.py:312-317): deltas*stds+means ->
 * bbox_transform_inv on the rois -> clip_boxes -> / im_scale; rows with cls_prob[:,1] <= score_thresh sort last
 * (score -inf); descending sort; NMS(nms_thresh). dets[R][5] holds every row in sorted order, keep_pos the kept
 * positions; meta[0] = rows above the threshold, meta[1] = rows kept: the detections are
 * dets[keep_pos[i]] for the i with keep_pos[i] < meta[0]. stds4 / means4 are HOST float[4]. */
size_t dana_detect_postprocess_workspace_bytes(int R);
int dana_detect_postprocess(const float* rois, const float* cls_prob, const float* bbox_pred, const float* im_info,
                            int R, const float* stds4, const float* means4, int normalize, float score_thresh,
                            float nms_thresh, int nms_inclusive, float* dets, int* keep_pos, int* meta,
                            void* workspace, size_t workspace_bytes, dana_stream_t stream);

/* ---- dense contractions on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) --------------------- */

/* nn.Conv2d + frozen BatchNorm2d/bias + residual + ReLU (resnet.py:66-102 Bottleneck, :109-112 stem;
 * rpn.py:28,63 RPN_Conv). input NHWC [batch][in_h][in_w][in_pix_stride>=cin]; weight [cout][kh][kw][cin]
 * (see dana_pack_conv_weight); out[m][n] = relu?( acc*scale[n] + shift[n] + residual[m][n] ) written
 * with row stride out_pix_stride. scale/shift/residual may be NULL. cin % 32 == 0 unless STEM7. */
int dana_conv2d_nhwc(const float* input, const float* weight, float* output, const float* scale,
                     const float* shift, const float* residual, int batch, int in_h, int in_w, int cin,
                     int cout, int kh, int kw, int stride, int pad, long in_pix_stride, long out_pix_stride,
                     long res_pix_stride, int flags, dana_stream_t stream);

/* dana_conv2d_nhwc with the ReLU adjoint fused into the epilogue: out = mask_act > 0 ? conv(...) + residual : 0.
 * Used for data gradients (backward of resnet.py:83-100): mask_act is the activation whose ReLU sits in front of the
 * differentiated conv's input, [batch*oh*ow][mask_pix_stride]. */
int dana_conv2d_nhwc_masked(const float* input, const float* weight, float* output, const float* scale,
                            const float* shift, const float* residual, const float* mask_act, int batch, int in_h,
                            int in_w, int cin, int cout, int kh, int kw, int stride, int pad, long in_pix_stride,
                            long out_pix_stride, long res_pix_stride, long mask_pix_stride, int flags,
                            dana_stream_t stream);
/* The tail of a Bottleneck in ONE launch (resnet.py:92-100): out = relu?( bn3(conv3(relu(bn2(conv2(x))))) + residual ),
 * conv2 3x3 / stride 1 / pad 1 with cmid = 64 output channels, conv3 1x1 cmid -> cout. Both weights are dana_split_weight
 * planes (of the packed [cmid][3][3][cin] rows and of the [cout][cmid] rows); scale2 / shift2 and scale3 / shift3 are the
 * folded frozen BatchNorms. conv2's output tile stays in LDS as conv3's A operand -- the [M][64] map is never written --
 * and the result is bit-identical to dana_conv2d_nhwc (3x3, DANA_EPI_RELU) followed by dana_conv2d_nhwc (1x1, residual).
 * flags: DANA_EPI_RELU = the final ReLU. Split kernel only (dana_set_mfma_mode != 0). */
int dana_bottleneck_tail_nhwc(const float* input, const float* w2_split, const float* scale2, const float* shift2,
                              const float* w3_split, const float* scale3, const float* shift3, const float* residual,
                              float* output, int batch, int h, int w, int cin, int cmid, int cout, long in_pix_stride,
                              long out_pix_stride, long res_pix_stride, int flags, dana_stream_t stream);

/* The same conv over TWO image groups in one launch: input rows = batch0 images of h0 x w0 followed by
 * batch1 images of h1 x w1 (same channels / pixel stride); outputs go to out0 / out1 with their own row
 * strides. RCNN_base is applied to the query batch and to the support batch (dana.py:98,100): sharing
 * each layer's launch doubles the tile count and halves the tail on the 256 CUs. */
int dana_conv2d_nhwc_dual(const float* input, const float* weight, float* out0, float* out1, const float* scale,
                          const float* shift, const float* res0, const float* res1, int batch0, int h0, int w0,
                          int batch1, int h1, int w1, int cin, int cout, int kh, int kw, int stride, int pad,
                          long in_pix_stride, long out0_stride, long out1_stride, long res0_stride,
                          long res1_stride, int flags, dana_stream_t stream);

/* Winograd F(2x2,3x3) path for the stride-1 pad-1 3x3 convs with many channels (layer3/layer4 conv2,
 * RPN_Conv): u = dana_winograd_filter_transform(packed weight [cout][3][3][cin]) -> [16][cout][cin], then
 * out = relu?(conv3x3(input) * scale + shift) with 2.25x fewer multiplies (the 16 GEMMs run on the igemm kernels, see dana_set_mfma_mode). */
int dana_winograd_filter_transform(const float* w_packed, float* u, int cout, int cin, dana_stream_t stream);
size_t dana_conv3x3_winograd_workspace_bytes(int batch, int h, int w, int cin, int cout);
int dana_conv3x3_winograd_nhwc(const float* input, const float* u, float* output, const float* scale,
                               const float* shift, int batch, int h, int w, int cin, int cout, long in_pix_stride,
                               long out_pix_stride, int flags, void* workspace, size_t workspace_bytes,
                               dana_stream_t stream);
/* same, with the ReLU adjoint fused into the output transform (out = mask_act > 0 ? conv : 0): the data gradient of
 * a 3x3 conv is a 3x3 conv on flipped weights, masked by the activation in front of the differentiated conv */
int dana_conv3x3_winograd_nhwc_masked(const float* input, const float* u, float* output, const float* scale,
                                      const float* shift, const float* mask_act, int batch, int h, int w, int cin,
                                      int cout, long in_pix_stride, long out_pix_stride, long mask_pix_stride,
                                      int flags, void* workspace, size_t workspace_bytes, dana_stream_t stream);

/* F(4x4,3x3) variant: u = dana_winograd4_filter_transform(...) -> [36][cout][cin]; 4x fewer multiplies than the
 * direct conv and smaller transformed tensors than F(2x2,3x3), at ~1e-5 relative error (transform constants up to 8) */
int dana_winograd4_filter_transform(const float* w_packed, float* u, int cout, int cin, dana_stream_t stream);
size_t dana_conv3x3_winograd4_workspace_bytes(int batch, int h, int w, int cin, int cout);
int dana_conv3x3_winograd4_nhwc_masked(const float* input, const float* u, float* output, const float* scale,
                                       const float* shift, const float* mask_act, int batch, int h, int w, int cin,
                                       int cout, long in_pix_stride, long out_pix_stride, long mask_pix_stride,
                                       int flags, void* workspace, size_t workspace_bytes, dana_stream_t stream);
/* F(4x4,3x3) over TWO image groups (input: group 0's [n0][h0][w0] pixels, then group 1's [n1][h1][w1]): two input
 * transforms, ONE batched plane GEMM over all tiles, two output transforms (resnet.py:92-94 on the query and the support
 * batch of dana.py:98,100 with the same filters) */
size_t dana_conv3x3_winograd4_dual_workspace_bytes(int n0, int h0, int w0, int n1, int h1, int w1, int cin, int cout);
int dana_conv3x3_winograd4_nhwc_dual(const float* input, const float* u, float* out0, float* out1, const float* scale,
                                     const float* shift, int n0, int h0, int w0, int n1, int h1, int w1, int cin,
                                     int cout, long in_pix_stride, long out0_pix_stride, long out1_pix_stride, int flags,
                                     void* workspace, size_t workspace_bytes, dana_stream_t stream);
/* ... with the ReLU adjoint of the masked data-gradient form (dana_conv3x3_winograd4_nhwc_masked) per group: out = mask > 0 ?
 * conv : 0. The merged backward of an identity bottleneck (resnet.py:84-100 differentiated over the [query | support]
 * buffers) runs its 3x3 data gradient for both batches as ONE batched plane GEMM with it. mask0 / mask1 may be NULL. */
int dana_conv3x3_winograd4_nhwc_dual_masked(const float* input, const float* u, float* out0, float* out1,
                                            const float* scale, const float* shift, const float* mask0,
                                            const float* mask1, int n0, int h0, int w0, int n1, int h1, int w1, int cin,
                                            int cout, long in_pix_stride, long out0_pix_stride, long out1_pix_stride,
                                            long mask0_pix_stride, long mask1_pix_stride, int flags, void* workspace,
                                            size_t workspace_bytes, dana_stream_t stream);

/* weight gradient of a stride-1 pad-1 3x3 conv in the F(4x4,3x3) domain: dU[36] = sum over tiles of
 * (A dY A^T)^T (B^T x B), dW = G^T dU G -- 4x fewer multiplies than dana_conv2d_wgrad_nhwc; same packed
 * [cout][3][3][cin] result, same row_scale / accumulate semantics. cin % 64 == 0, cout % 4 == 0. */
size_t dana_conv3x3_wgrad_winograd4_workspace_bytes(int batch, int h, int w, int cin, int cout);
int dana_conv3x3_wgrad_winograd4(const float* grad_out, const float* input, float* grad_weight, int batch, int h, int w,
                                 int cin, int cout, long in_pix_stride, long grad_pix_stride, const float* row_scale,
                                 int accumulate, void* workspace, size_t workspace_bytes, dana_stream_t stream);
/* ... with the input's transform handed in: v = the V planes [36][tiles][cin] that the forward's
 * dana_conv3x3_winograd4_nhwc(_masked) call over the same input left in the FIRST 36*tiles*cin floats of its workspace
 * (the caller keeps that buffer until the backward) -- the weight gradient skips its own input transform. Same workspace
 * size as dana_conv3x3_wgrad_winograd4. */
int dana_conv3x3_wgrad_winograd4_v(const float* grad_out, const float* v, float* grad_weight, int batch, int h, int w,
                                   int cin, int cout, long grad_pix_stride, const float* row_scale, int accumulate,
                                   void* workspace, size_t workspace_bytes, dana_stream_t stream);

/* nn.Linear / torch.bmm (dana.py:124,140,142,147,266-290): c[z][m][n] = epi(alpha * sum_k a[z][m][k]*b[z][n][k]).
 * Both operands K-contiguous ("NT"); nn.Linear weights [out][in] are used as stored. k % 4 == 0. */
int dana_gemm_nt(const float* a, const float* b, float* c, const float* scale, const float* shift,
                 const float* residual, int m, int n, int k, long lda, long ldb, long ldc, long ldr, int batch,
                 long batch_a, long batch_b, long batch_c, float alpha, int flags, dana_stream_t stream);

/* ---- HBM-bound layout / pooling / packing kernels ---------------------------------------------- */

/* boundary conversions between the reference's NCHW tensors and this build's NHWC activations */
int dana_nchw_to_nhwc(const float* in, float* out, int batch, int channels, int height, int width, int cpad,
                      long out_pix_stride, dana_stream_t stream);
int dana_nhwc_to_nchw(const float* in, float* out, int batch, int channels, int height, int width,
                      long in_pix_stride, dana_stream_t stream);
/* nn.MaxPool2d(3, 2, padding=0, ceil_mode=True): resnet.py:113 */
int dana_maxpool3x3s2_ceil_nhwc(const float* in, float* out, int batch, int height, int width, int channels,
                                dana_stream_t stream);
/* nn.AvgPool2d(14, stride=1) on the 20x20 support maps: dana.py:42,105-108 */
int dana_avgpool_nhwc(const float* in, float* out, int batch, int height, int width, int channels, int k, int stride,
                      dana_stream_t stream);
/* RCNN_top(pool5).mean(3).mean(2): dana.py:387-389; in[groups][positions][stride] -> out[groups][channels] */
int dana_spatial_mean_nhwc(const float* in, float* out, int groups, int positions, int channels, long in_pix_stride,
                           dana_stream_t stream);
/* sibling model `fgn` (framework/fgn.py:36-39,156-161): its head's nn.BatchNorm2d layers are NOT frozen. Batch
 * statistics over rows (= N*H*W) per channel, two passes (mean, then biased variance about it); workspace:
 * dana_colsum_workspace_bytes(rows, channels). dana_bn_fold turns (gamma, beta, mean, var) into scale / shift for
 * dana_scale_shift_relu: x = relu?(x * scale + shift), in place. */
int dana_batch_stats(const float* x, float* mean, float* var_biased, long rows, int channels, long ld, void* workspace,
                     size_t workspace_bytes, dana_stream_t stream);
int dana_scale_shift_relu(float* x, const float* scale, const float* shift, long rows, int channels, int relu,
                          dana_stream_t stream);
/* adjoint of a TRAIN-mode nn.BatchNorm2d over NHWC rows (batch statistics mean / var_biased of x, as dana_batch_stats
 * returns them): grad_x = gamma / sqrt(var + eps) * (g - mean(g) - xhat * mean(g * xhat)), grad_gamma (+)= sum g * xhat,
 * grad_beta (+)= sum g (fgn.py:147-153: the head's bn1 / bn2 are ordinary, trainable BatchNorm layers) */
int dana_bn_train_backward(const float* grad_out, const float* x, const float* mean, const float* var_biased,
                           const float* gamma, float eps, long rows, int channels, float* grad_x, float* grad_gamma,
                           float* grad_beta, int accumulate, dana_stream_t stream);
/* sibling model `fsod` (framework/fsod.py:109-116, 207-214): depth-wise "valid" cross-correlation
 * F.conv2d(feat, kernel.view(C, 1, kh, kw), groups=C): out[n][oh][ow][c] = sum feat[n][oh+i][ow+j][c] *
 * kernels[n / maps_per_kernel][i][j][c]; feat [n_maps][height][width][feat_pix_stride], out [n_maps][oh][ow][channels] */
int dana_depthwise_corr_nhwc(const float* feat, const float* kernels, float* out, long n_maps, int height, int width,
                             int channels, int kh, int kw, long maps_per_kernel, long feat_pix_stride,
                             dana_stream_t stream);
/* adjoints of dana_depthwise_corr_nhwc: grad_feat [n_maps][height][width][channels] (dense; null to skip) and
 * grad_kernels [ceil(n_maps / maps_per_kernel)][kh][kw][channels] (null to skip; accumulated when accumulate_kernels) */
int dana_depthwise_corr_backward_nhwc(const float* grad_out, const float* feat, const float* kernels, float* grad_feat,
                                      float* grad_kernels, long n_maps, int height, int width, int channels, int kh,
                                      int kw, long maps_per_kernel, long feat_pix_stride, int accumulate_kernels,
                                      dana_stream_t stream);
/* sibling model `meta` (framework/meta.py): nn.MaxPool2d(2) of the PRN (:203,246), nn.Sigmoid (:202,250), and the
 * channel-wise product of the RoI features with their image's class-attentive vector (:136-140) */
int dana_maxpool2x2s2_nhwc(const float* in, float* out, int batch, int height, int width, int channels,
                           dana_stream_t stream);
/* adjoint of dana_maxpool2x2s2_nhwc: grad_in[b][y][x][c] = grad_out of the window whose (first) maximum the element is */
int dana_maxpool2x2s2_backward_nhwc(const float* in, const float* grad_out, float* grad_in, int batch, int height,
                                    int width, int channels, dana_stream_t stream);
int dana_sigmoid(float* x, long n, dana_stream_t stream);
int dana_scale_rows_by_group(const float* x, const float* group_vec, float* out, long rows, long rows_per_group,
                             int channels, dana_stream_t stream);
/* PositionalEncoding.forward: dana.py:322-324; out[r] = in[r] + pe[r % length] */
int dana_add_pe(const float* in, const float* pe, float* out, long rows, int length, int channels,
                long in_stride, long out_stride, dana_stream_t stream);
/* ... for `groups` row groups whose inputs sit in_group_stride floats apart (the positive supports of every image of the
 * batch, way * shot maps apart: dana.py:103,126-130) in one launch; out[g][r] = in[g][r] + pe[r % length] */
int dana_add_pe_groups(const float* in, const float* pe, float* out, long groups, long rows_per_group, int length, int channels,
                       long in_group_stride, long out_group_stride, dana_stream_t stream);
/* x - x.mean(1, keepdim=True): dana.py:125,141,267,272; x[groups][length][ld], in place */
size_t dana_colmean_sub_workspace_bytes(int groups, int length, int dim);
int dana_colmean_sub(float* x, int groups, int length, int dim, long ld, void* workspace, size_t workspace_bytes,
                     dana_stream_t stream);
/* .transpose(1, 2).contiguous() of [groups][rows][cols] -> [groups][cols][ldo] */
int dana_transpose_batched(const float* in, float* out, int groups, int rows, int cols, long ldi, long ldo,
                           long in_batch, long out_batch, dana_stream_t stream);
/* nn.Conv2d weight OIHW -> [O][KH][KW][I] (stem7: [O][7][8][4], zero padded) for dana_conv2d_nhwc */
int dana_pack_conv_weight(const float* w_oihw, float* out, int cout, int cin, int kh, int kw, int stem7,
                          dana_stream_t stream);
/* eval-mode nn.BatchNorm2d (dana.py:362-385) -> scale = gamma/sqrt(var+eps), shift = beta - mean*scale */
int dana_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                 float* shift, int n, dana_stream_t stream);

/* ---- BA block + CISA reductions: dana.py:133-147, 266-279 -------------------------------------- */

/* nn.Linear(dim, 1): out[r] = x[r].w + bias[0] (rpn_unary_layer, rcnn_unary_layer, rpn_channel_k_layer) */
int dana_rowdot(const float* x, const float* w, const float* bias, float* out, long rows, int dim, long ld,
                dana_stream_t stream);
/* F.softmax(x, last dim), in place, x[groups][length] */
int dana_softmax_rows(float* x, long groups, int length, long ld, dana_stream_t stream);
/* out-of-place: out[g][:length] = softmax(x[g][:length]) (cls_prob next to the untouched cls_score, dana.py:290-292) */
int dana_softmax_rows_to(const float* x, float* out, long groups, int length, long ld_in, long ld_out, dana_stream_t stream);
/* rois_label of the training forward (dana.py:191-194): out[0..n) = (int64) labels[i], out[n..2n) = 0 */
int dana_labels_posneg_i64(const float* labels, long long* out, long n, dana_stream_t stream);
/* S += gamma * leaky_relu(w^T S): dana.py:136-137; s[groups][length][ld], w[groups][length] */
int dana_ba_apply(float* s, const float* w, int groups, int length, int dim, long ld, float gamma, float slope,
                  dana_stream_t stream);
/* A = (softmax_keys(scores) + unary_gamma * unary^T) * out_scale per `length`-wide segment (one per shot):
 * dana.py:143-146 / :274-278; unary + (row / rows_per_batch)*unary_batch_stride -> [nseg][length]; cols nseg*length..kpad-1 zeroed */
int dana_attn_softmax_unary(float* scores, const float* unary, long rows, long rows_per_batch, long unary_batch_stride,
                            int nseg, int length, long ld, int kpad, float unary_gamma, float out_scale,
                            dana_stream_t stream);

/* ---- training targets for the sampled RoIs: lib/model/rpn/proposal_target_layer_cascade.py:33-213 ---- */

/* candidates = rois[B][n_rois][5] (cols 1..4) followed by gt_boxes[B][n_gt][5] (cols 0..3) (:43-47).
 * max_overlaps / gt_assignment [B][n_rois+n_gt] (bbox_overlaps_batch + torch.max, :122-124),
 * fg_list / bg_list: ascending candidate indices with max_overlap >= fg_thresh / in [bg_lo, bg_hi)
 * (:128-133), counts[B][2] = their lengths -- the only values the host RNG needs. */
int dana_proposal_target_prepare(const float* rois, const float* gt_boxes, int B, int n_rois, int n_gt,
                                 float fg_thresh, float bg_thresh_hi, float bg_thresh_lo, float* max_overlaps,
                                 int* gt_assignment, int* fg_list, int* bg_list, int* counts, dana_stream_t stream);
/* picks[B][rois_per_image]: position in fg_list for slot r < fg_taken[b], in bg_list otherwise (the
 * host draws them with np.random exactly as :143-175). means4/stds4/inside_w4 are HOST float[4].
 * -> rois_out[B][R][5], labels_out[B][R], bbox_targets / inside / outside weights [B][R][4] (:83-91,:183-204). */
int dana_proposal_target_gather(const float* rois, const float* gt_boxes, int B, int n_rois, int n_gt,
                                const int* gt_assignment, const int* fg_list, const int* bg_list, const int* picks,
                                const int* fg_taken, int rois_per_image, const float* means4, const float* stds4,
                                const float* inside_w4, int normalize, float* rois_out, float* labels_out,
                                float* bbox_targets, float* inside_weights, float* outside_weights,
                                dana_stream_t stream);

/* ---- anchor targets + RPN losses: lib/model/rpn/anchor_target_layer.py:48-193, rpn.py:97-115 ---------- */

/* All H*W*A anchors per image in (h, w, a) order. labels[B][n] before subsampling (-1 / 0 / 1; anchors
 * outside image 0's bounds stay -1), max_overlaps (-2 = outside), argmax over gt, ascending fg (label 1) /
 * bg (label 0) lists and counts[B][2] for the host's np.random.permutation draws (:137-156). */
int dana_anchor_target_prepare(const float* gt_boxes, const float* im_info, const float* base_anchors, int B,
                               int A, int H, int W, int feat_stride, int n_gt, float negative_overlap,
                               float positive_overlap, float* labels, float* max_overlaps, int* argmax,
                               int* fg_list, int* bg_list, int* counts, dana_stream_t stream);
/* labels[image][list[pos]] = -1 for n subsampled-away entries; which[e] = image*2 + (0 fg | 1 bg). */
int dana_anchor_target_disable(float* labels, const int* fg_list, const int* bg_list, const int* which,
                               const int* pos, int n, int anchors_per_image, dana_stream_t stream);
/* the same with the number of entries read from DEVICE memory (n_dev[0] <= capacity) and the entries interleaved as
 * (which, pos) pairs: the launch parameters never change, so a captured hipGraph can replay it with each step's draws */
int dana_anchor_target_disable_dev(float* labels, const int* fg_list, const int* bg_list, const int* n_dev,
                                   const int* which_pos_pairs, int capacity, int anchors_per_image,
                                   dana_stream_t stream);
/* _AnchorTargetLayer's four outputs in the reference layouts: labels_out[B][A*H*W] (a-major),
 * bbox_targets / inside / outside weights [B][4A][H*W] (:171-191). */
int dana_anchor_target_outputs(const float* labels, const float* max_overlaps, const int* argmax,
                               const float* gt_boxes, const float* base_anchors, int B, int A, int H, int W,
                               int feat_stride, int n_gt, float inside_weight, float outside_weight, const float* outside_weight_dev,
                               float* labels_out, float* bbox_targets, float* inside_weights,
                               float* outside_weights, dana_stream_t stream);
/* RCNN losses of the training forward (dana.py:199-217), fused: rows 0..n-1 = positive-support head scores
 * [n][2] with the proposal-target labels [n] (0 / 1), rows n..2n-1 = negative-support head scores (labels 0).
 * losses3[0] = cross-entropy over fg + hard-negative-mined bg rows (1:2:1: bg ranked by fg probability, descending, per
 * half; dana.py:204-215), losses3[1] = _smooth_l1_loss(bbox_pred, targets, in, out) (net_utils.py:71-85, sigma,
 * mean over rois), losses3[2] = number of rows kept. Optional gradient seeds (may be NULL): d losses3[0] / d scores
 * and d losses3[1] / d bbox_pred. No host sync (the reference's nonzero / sort / index chain has three). */
/* the plain losses of the sibling detectors (faster_rcnn.py:93-98): losses2[0] = F.cross_entropy(scores [n][n_classes],
 * labels int64 [n]) (mean), losses2[1] = _smooth_l1_loss(sigma) (net_utils.py:71-85); optional gradient seeds
 * d losses2[0] / d scores [n][n_classes] and d losses2[1] / d bbox_pred [n][4]. One launch, no host sync.
 * A label outside 0..n_classes-1 is never used as an index: losses2[0] and that row's score gradient become NaN
 * (F.cross_entropy would raise); there is no ignore_index. */
int dana_plain_rcnn_loss(const float* scores, const long long* labels, const float* bbox_pred, const float* bbox_targets,
                         const float* inside_weights, const float* outside_weights, int n, int n_classes, float sigma,
                         float* losses2, float* grad_scores, float* grad_bbox, dana_stream_t stream);
size_t dana_rcnn_loss_workspace_bytes(int n);
int dana_rcnn_loss(const float* score_pos, const float* score_neg, const float* labels, const float* bbox_pred,
                   const float* bbox_targets, const float* inside_weights, const float* outside_weights, int n,
                   float sigma, float* losses3, float* grad_score_pos, float* grad_score_neg, float* grad_bbox,
                   void* workspace, size_t workspace_bytes, dana_stream_t stream);
/* losses3[0] = F.cross_entropy over labels >= 0 (rpn.py:97-105), losses3[1] = _smooth_l1_loss(sigma, dims 1,2,3)
 * (rpn.py:114), losses3[2] = number of labels >= 0; outside_weight_dev (optional, device) overrides outside_weight; fused over heads[B*H*W][row stride] = (2A cls | 4A bbox) without materialising the targets. */
size_t dana_rpn_loss_workspace_bytes(void);
int dana_rpn_loss(const float* heads, long head_row_stride, const float* labels, const int* argmax,
                  const float* gt_boxes, const float* base_anchors, int B, int A, int H, int W, int feat_stride,
                  int n_gt, float sigma, float inside_weight, float outside_weight, const float* outside_weight_dev, float* losses3, void* workspace,
                  size_t workspace_bytes, dana_stream_t stream);

/* ---- backward building blocks of the conv / Linear layers (training step, SURVEY.md 8d variant S) --------
 * what autograd + cuDNN compute for nn.Conv2d in the reference's loss.backward() (train.py:141-143) */

/* grad_weight[cout][kh][kw][cin] (+)= row_scale[cout] * sum_m grad_out[m][cout] * im2col(input)[m][...]; deterministic
 * split-M reduction through the workspace. row_scale (optional) = the frozen-BN scale of each output channel, so the
 * launch can accumulate straight into a parameter gradient kept in this layout. cin % 64 == 0, cout % 4 == 0. */
size_t dana_conv2d_wgrad_workspace_bytes(int batch, int in_h, int in_w, int cin, int cout, int kh, int kw, int stride,
                                         int pad);
int dana_conv2d_wgrad_nhwc(const float* grad_out, const float* input, float* grad_weight, int batch, int in_h,
                           int in_w, int cin, int cout, int kh, int kw, int stride, int pad, long in_pix_stride,
                           long grad_pix_stride, const float* row_scale, int accumulate, void* workspace, size_t workspace_bytes,
                           dana_stream_t stream);
/* A batch of "TN" GEMMs, out[z][n][k] (+)= sum_m y[z][m][n] * x[z][m][k] for n < n_valid -- the adjoints of torch.bmm
 * w.r.t. its right operand (dana.py:140-150, 270-283: d value / d key of the dual-awareness attention), one plane z per
 * image in ONE launch. y rows have ldy >= n floats (columns n_valid..n-1 may be zero padding), x rows ldx >= k; planes
 * are batch_y / batch_x / batch_out floats apart. k % 64 == 0, n % 4 == 0; deterministic split-M reduction. */
size_t dana_gemm_tn_batched_workspace_bytes(int planes, int m, int n, int k);
int dana_gemm_tn_batched(const float* y, const float* x, float* out, int planes, int m, int n, int k, long ldy, long ldx,
                         long batch_y, long batch_x, long batch_out, int n_valid, int accumulate, void* workspace,
                         size_t workspace_bytes, dana_stream_t stream);
/* weights for the data gradient of a stride-1 conv: out[cin][kh][kw][cout] = w[cout][KH-1-kh][KW-1-kw][cin]*scale[cout];
 * grad_input = dana_conv2d_nhwc(grad_out, out, cin <-> cout swapped, same kernel size / pad) */
int dana_conv2d_dgrad_weight(const float* w_packed, const float* scale, float* out, int cout, int cin, int kh, int kw,
                             dana_stream_t stream);
/* data gradient of a strided 1x1 conv: scatter the compact result to the strided positions, zero elsewhere;
 * mask_act (optional, [batch][ih][iw][channels]): zero the gradient where that activation is <= 0 (ReLU adjoint) */
int dana_upsample_scatter_nhwc(const float* compact, float* out, const float* mask_act, int batch, int oh, int ow,
                               int ih, int iw, int channels, int stride, dana_stream_t stream);

/* the rows a strided 1x1 conv reads (resnet.py:71), compacted: compact[b][oh][ow][:] = x[b][oh*stride][ow*stride][:];
 * its weight gradient is then dana_conv2d_wgrad_nhwc with stride 1 over [batch][oh][ow] */
int dana_downsample_gather_nhwc(const float* x, float* compact, int batch, int ih, int iw, int channels, int stride,
                                long in_pix_stride, dana_stream_t stream);

/* element-wise / reduction glue of the backward pass */
int dana_relu_mask(float* grad, const float* act, long rows, int channels, long ld_grad, long ld_act,
                   dana_stream_t stream);                       /* grad *= (saved output > 0) */
int dana_axpy_rows(float* y, const float* x, long rows, int channels, long ld_y, long ld_x, float alpha,
                   int accumulate, dana_stream_t stream);        /* y (+)= alpha * x, strided rows */
/* y *= x over strided rows: the correlation of attention_type 'product' (dana.py:155-156: base_feat * dense_support_feature;
 * :285-286: query_mat * dense_support_feature), in place in the attended buffer */
int dana_mul_rows(float* y, const float* x, long rows, int channels, long ld_y, long ld_x, dana_stream_t stream);
int dana_rowscale(float* dw, const float* scale, int rows, long cols, dana_stream_t stream); /* frozen-BN scale on dW rows */
int dana_unpack_conv_weight_grad(const float* packed, float* w_oihw, int cout, int cin, int kh, int kw, int accumulate,
                                 dana_stream_t stream);          /* packed [O][KH][KW][I] -> OIHW (.grad layout) */
size_t dana_colsum_workspace_bytes(long rows, int channels);
int dana_colsum(const float* x, float* out, long rows, int channels, long ld, float alpha, int accumulate,
                void* workspace, size_t workspace_bytes, dana_stream_t stream);   /* bias gradients, deterministic */
int dana_avgpool_backward_nhwc(const float* grad_out, float* grad_in, int batch, int height, int width, int channels,
                               int k, int stride, dana_stream_t stream);
/* dana_colsum over `batch` matrices in one launch pair: matrix z at x + z * x_batch, its sums at out + z * out_batch
 * (the unary-term adjoint of the attention, one matrix per image: dana.py:132-137); workspace = batch x the single one */
int dana_colsum_batched(const float* x, float* out, int batch, long rows, int channels, long ld, long x_batch, long out_batch,
                        float alpha, int accumulate, void* workspace, size_t workspace_bytes, dana_stream_t stream);
int dana_softmax_rows_backward(float* grad, const float* prob, long rows, int length, long ld_grad, long ld_prob,
                               dana_stream_t stream);            /* grad <- p * (grad - <p, grad>) */
int dana_gemm_small(const float* a, long a_stride_m, long a_stride_k, const float* b, long b_stride_k, long b_stride_n,
                    float* c, long c_stride_m, long c_stride_n, int m, int n, int k, float alpha, int accumulate,
                    dana_stream_t stream);                       /* skinny heads (N or K of 2 / 4) */

/* torch.optim.SGD with momentum over a flat fp32 segment (train.py:76-87: lr, 2x lr and no decay for biases):
 * g' = grad_scale * g + weight_decay * p; buf = first_step ? g' : momentum * buf + g'; p -= lr * buf */
int dana_sgd_momentum(float* params, const float* grads, float* momentum_buf, long n, float lr, float momentum,
                      float weight_decay, float grad_scale, int first_step, dana_stream_t stream);
/* torch.optim.Adam over a flat fp32 segment (train.py:84-85; defaults betas (0.9, 0.999), eps 1e-8): step = 1, 2, ... */
int dana_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
              float beta2, float eps, float weight_decay, float grad_scale, int step, dana_stream_t stream);
/* adjoint of dana_rowdot: grad_w[dim] (+)= sum_r grad_out[r] x[r]; grad_x[r] += grad_out[r] w (grad_x may be NULL);
 * workspace: dana_colsum_workspace_bytes(rows, dim) */
int dana_rowdot_backward(const float* x, const float* grad_out, const float* w, float* grad_x, float* grad_w, long rows,
                         int dim, long ld_x, long ld_grad_x, int accumulate_w, void* workspace, size_t workspace_bytes,
                         dana_stream_t stream);
/* out[g][p][:] (+)= alpha * in[g][:] : adjoint of dana_spatial_mean_nhwc (alpha = 1/positions) */
int dana_broadcast_rows(const float* in, float* out, long groups, int positions, int channels, float alpha,
                        int accumulate, dana_stream_t stream);
/* adjoint of dana_ba_apply (dana.py:133-137), in place on grad_s [groups*length][dim]: s = the block's INPUT,
 * gvec[groups][dim] = weights^T s, gsum[groups][dim] = column sums of grad_s per group; grad_weights[groups*length] out */
/* ... its two per-group reductions in one launch: gvec[g][:] = weights[g]^T s[g], gsum[g][:] = column sums of grad_s[g] */
int dana_ba_backward_prep(const float* s, const float* weights, const float* grad_s, float* gvec, float* gsum, long groups,
                          int length, int dim, dana_stream_t stream);
int dana_ba_backward(float* grad_s, const float* s, const float* weights, const float* gvec, const float* gsum,
                     float* grad_weights, long groups, int length, int dim, float gamma, float slope,
                     dana_stream_t stream);
/* adjoint of dana_attn_softmax_unary, in place on grad_a: -> alpha * dL/d(raw scores) (alpha = the 1/sqrt(d) of QK^T) */
int dana_attn_softmax_unary_backward(float* grad_a, const float* a, const float* unary, long rows, long rows_per_batch,
                                     long unary_batch_stride, int nseg, int length, long ld, int kpad,
                                     float unary_gamma, float out_scale, float alpha, dana_stream_t stream);
/* adjoint of dana_rpn_loss w.r.t. the head buffer: grad_heads[B*H*W][row stride] (zero-filled here);
 * losses3 = dana_rpn_loss's DEVICE output (its [2] is the cross-entropy's divisor); grad_cls / grad_box = upstream scalars,
 * or grad_scales_dev (2 floats in device memory) when given */
int dana_rpn_loss_backward(const float* heads, long head_row_stride, const float* labels, const int* argmax,
                           const float* gt_boxes, const float* base_anchors, int B, int A, int H, int W,
                           int feat_stride, int n_gt, float sigma, float inside_weight, float outside_weight, const float* outside_weight_dev,
                           const float* losses3, float grad_cls, float grad_box, const float* grad_scales_dev,
                           float* grad_heads, dana_stream_t stream);
/* x[0..n) *= scalar_dev[0]: applies an upstream loss gradient that lives in device memory (no host sync) */
int dana_scale_by_device_scalar(float* x, long n, const float* scalar_dev, dana_stream_t stream);

/* ---- episode input pipeline: the loaders' per-image cv2 / numpy work ------------------------ */

/* minibatch.py:70-84 + blob.py:35-52 (prep_im_for_blob): RGB uint8 [h][w][3] -> BGR, optional horizontal flip,
 * fp32 minus cfg.PIXEL_MEANS (3 HOST floats, BGR order), cv2.resize(fx = fy = im_scale, INTER_LINEAR)
 * -> out [out_h][out_w][3] fp32 (the caller computes out_h/out_w = round(h * im_scale) like cv2 does) */
int dana_prep_image(const unsigned char* rgb_hwc, int height, int width, long row_stride_bytes, int flipped,
                    const float* pixel_means_bgr, double im_scale, float* out_hwc, int out_h, int out_w,
                    dana_stream_t stream);
/* fs_loader.py:118-139: crop [y_min..y_max] x [x_min..x_max] (inclusive) out of a prepared fp32 HWC image,
 * cv2.resize(dsize = (resized_w, resized_h), INTER_LINEAR), transpose to CHW, zero-pad to [3][target][target] */
int dana_crop_resize_pad(const float* im_hwc, int height, int width, int x_min, int y_min, int x_max, int y_max,
                         int resized_w, int resized_h, int target, float* out_chw, dana_stream_t stream);
/* fs_loader.py:186-280,318: crop window (start, size) of a prepared fp32 HWC image, zero-padded to [out_h][out_w] and
 * permuted to CHW: one plane set of the batch holder */
int dana_crop_pad_chw(const float* im_hwc, int height, int width, int y_start, int x_start, int crop_h, int crop_w,
                      float* out_chw, int out_h, int out_w, dana_stream_t stream);

/* ---- counter-based device RNG for the training target layers (opt-in) ---------------------------
 * Philox-4x32-10 keyed by (seed, offset, image): the draws of anchor_target_layer.py:137-156 and
 * proposal_target_layer_cascade.py:143-175 without reading the fg / bg counts back to the host. */

/* counts[B][2] (fg, bg) from dana_proposal_target_prepare -> picks[B][rois_per_image] (positions in the fg list for
 * the first fg_taken[b] slots, in the bg list for the rest) for dana_proposal_target_gather */
int dana_proposal_target_sample(const int* counts, int B, int n_candidates, int rois_per_image, int fg_rois_per_image,
                                unsigned long long seed, unsigned long long offset, int* picks, int* fg_taken,
                                dana_stream_t stream);
/* labels[B][anchors_per_image] from dana_anchor_target_prepare: keep a uniform random num_fg-subset of the positives and
 * (rpn_batchsize - kept) of the negatives, label the rest -1; inv_num_examples[0] = 1 / (examples of the LAST image)
 * for the outside_weight_dev argument of dana_rpn_loss / dana_rpn_loss_backward / dana_anchor_target_outputs */
int dana_anchor_target_subsample(float* labels, const int* fg_list, const int* bg_list, const int* counts, int B,
                                 int anchors_per_image, int rpn_batchsize, int num_fg, unsigned long long seed,
                                 unsigned long long offset, float* inv_num_examples, dana_stream_t stream);
/* fp32 weight rows [batch][n][k] (row stride ldw, batch stride batch_w floats) -> bf16 planes [batch][kp / 16][3][n][16]
 * (out: dana_split_weight_bytes bytes) for the DANA_W_SPLIT3 flag of the contraction entry points. A weight of the
 * reference (conv: packed [cout][kh*kw*cin]; nn.Linear: [out][in]) is split once per weight version. */
size_t dana_split_weight_bytes(int n, int k, int batch);
int dana_split_weight(const float* w, long ldw, int n, int k, int batch, long batch_w, void* out, dana_stream_t stream);

/* One contraction over TWO concatenated channel segments: out[m][n] = epi(sum_{k<k0} a0[m][k] w[n][k] +
 * sum_{k<k1} a1[pix1(m)][k] w[n][k0+k]) where a0 is an NHWC map on the OUTPUT grid [batch][oh][ow] (pixel stride
 * a0_pix_stride) and a1 an NHWC map [batch][h1][w1] sampled at (oh*stride1, ow*stride1). This is a Caffe-ResNet
 * bottleneck's 1x1 expand conv and its (strided) 1x1 downsample conv (resnet.py:84-100: out = relu(bn3(conv3(t)) +
 * bn_d(downsample(x)))) with both frozen-BN scales folded into the weight rows [cout][k0 + k1] (dana_pack_cat2_weight)
 * and the two shifts added: the downsample's output never goes through HBM. Split kernel only (dana_set_mfma_mode != 0). */
int dana_conv1x1_cat2_nhwc(const float* a0, long a0_pix_stride, int k0, const float* a1, long a1_pix_stride, int k1,
                           int batch, int h1, int w1, int stride1, const float* weight, float* output,
                           const float* scale, const float* shift, const float* residual, long out_pix_stride,
                           long res_pix_stride, int cout, int flags, dana_stream_t stream);
/* the same over TWO image groups in one launch (query batch + support batch of one trunk layer): a0 holds group 0's
 * output-grid rows, then group 1's; a1 group 0's [n0][h0][w0] pixels, then group 1's [n1][h1][w1]; results go to out0 /
 * out1 with their own row strides (dana.py:98,100: RCNN_base runs on both batches with the same weights) */
int dana_conv1x1_cat2_nhwc_dual(const float* a0, long a0_pix_stride, int k0, const float* a1, long a1_pix_stride, int k1,
                                int n0, int h0, int w0, int n1, int h1, int w1, int stride1, const float* weight,
                                float* out0, float* out1, const float* scale, const float* shift, long out0_pix_stride,
                                long out1_pix_stride, int cout, int flags, dana_stream_t stream);
/* Second half of a split-K contraction: out[m][n] = epi(alpha * sum_s partials[s][m][n]) with the same epilogue as
 * dana_gemm_nt (scale, shift, residual, DANA_EPI_RELU), summed in slice order (deterministic). The slices themselves
 * are one batched dana_gemm_nt launch whose batch strides walk K (batch_a = batch_b = K / slices). n % 4 == 0. */
int dana_splitk_reduce(const float* partials, int slices, int m, int n, float* out, long ldc, const float* scale,
                       const float* shift, const float* residual, long ldr, float alpha, int flags,
                       dana_stream_t stream);
/* w_cat[n][0:k0] = s0[n] * w0[n][:], w_cat[n][k0:] = s1[n] * w1[n][:], shift[n] = b0[n] + b1[n] (w0 / w1 packed [cout][k]) */
int dana_pack_cat2_weight(const float* w0, const float* s0, const float* b0, int k0, const float* w1, const float* s1,
                          const float* b1, int k1, int cout, float* w_cat, float* shift, dana_stream_t stream);
/* hipGraph-friendly forms of the two samplers: the per-call Philox offset is `offset + counter_dev[0]`, with the call
 * counter living in device memory and advanced by dana_counter_add inside the same captured graph, so that every replay
 * draws fresh samples */
int dana_proposal_target_sample_ctr(const int* counts, int B, int n_candidates, int rois_per_image,
                                    int fg_rois_per_image, unsigned long long seed, unsigned long long offset,
                                    const unsigned long long* counter_dev, int* picks, int* fg_taken,
                                    dana_stream_t stream);
int dana_anchor_target_subsample_ctr(float* labels, const int* fg_list, const int* bg_list, const int* counts, int B,
                                     int anchors_per_image, int rpn_batchsize, int num_fg, unsigned long long seed,
                                     unsigned long long offset, const unsigned long long* counter_dev,
                                     float* inv_num_examples, dana_stream_t stream);
int dana_counter_add(unsigned long long* counter_dev, unsigned long long inc, dana_stream_t stream);

/* ---- launch programs -------------------------------------------------------------------------------------------------
 * The reference's training iteration (train.py:125-143) is ~1 500 launches on this library; issued one by one from the host
 * language they cost more host time than the GPU needs for them at eight ranks per host. A program is a recorded list of
 * calls of THIS header's `int dana_*(...)` entry points (the stream is one of their arguments), hipEventRecord and
 * hipStreamWaitEvent operations, re-issued in order from one C loop: the recorded step's launches with its arguments on
 * its streams behind its event edges (dana_amd/program.py records them from the eager step). Host-only; no device memory,
 * no synchronisation. A program is replayed by one thread at a time.
 *   signature: one character per argument of the entry point -- i (int), l (long / size_t / unsigned long long), p (pointer /
 *   dana_stream_t), f (float), d (double); words[k]: argument k as a 64-bit word (integers sign-extended, a float's bits in
 *   the low half, a double's bits). dana_program_run re-issues entries [begin, end) and returns the first non-zero status. */
int dana_program_create(void** program_out);
int dana_program_destroy(void* program);
int dana_program_add_call(void* program, void* entry_point, const char* signature, const unsigned long long* words, int nargs);
int dana_program_add_event_record(void* program, void* event, dana_stream_t stream);
int dana_program_add_event_wait(void* program, void* event, dana_stream_t stream);
int dana_program_size(void* program);
/* device memory zero fill / device-to-device copy on a stream (hipMemsetAsync / hipMemcpyAsync): what torch's zeros / zero_
 * and copy_ / clone do inside a recorded backward, as entry points a launch program can re-issue from its C loop */
int dana_fill_zero(void* dst, size_t bytes, dana_stream_t stream);
int dana_copy_d2d(void* dst, const void* src, size_t bytes, dana_stream_t stream);
int dana_program_run(void* program, int begin, int end);

#ifdef __cplusplus
}
#endif
#endif /* DANA_HIP_H */
