/*
 * cddmsl_hip.h -- C-ABI of libcddmsl_hip.so, the MI355X (gfx950) kernels of the CDDMSL training hot path.
 *
 * Boundary: the reference reaches its native ops through torch.autograd.Function wrappers over
 * `_C.<op>_forward/_backward(Tensor..., scalars...)` (detectron2/layers/roi_align_rotated.py:11-47; the pybind
 * file csrc/vision.cpp is absent from the tree) and through torchvision/ATen for the hot path.  This header is
 * the plain-C equivalent: device pointers + sizes + the HIP stream to enqueue on, no torch types.  Every function
 * returns 0 on success (1 = bad argument, 2 = launch failure), never synchronises, owns no memory (the caller
 * allocates outputs and workspaces) and keeps no global mutable state (one thread-local diagnostic id excepted, see
 * cddmsl_last_kernel), so calls are re-entrant and stream-ordered.
 *
 * Conventions
 *   dtype      0 = bf16 (throughput path), 1 = f32 (exact-f32 MFMA parity path)
 *   activations NHWC [N][H][W][C] in `dtype`; C * sizeof(dtype) must be a multiple of 16 bytes
 *   weights    [Cout][KH][KW][Cin] (= torch channels_last OIHW); f32 masters, `dtype` prepared copies
 *   rois       [K][5] f32 (batch_idx, x0, y0, x1, y1), grouped by image
 *   stream     hipStream_t passed as void*
 * Paths below are relative to /root/reference/detectron2/.
 */
#ifndef CDDMSL_HIP_H
#define CDDMSL_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

int cddmsl_abi_version(void);

/* ---- implicit-GEMM convolution / linear (bf16 or exact-f32 MFMA) ----------------------------------------------
 * replaces ATen conv2d / F.linear + FrozenBatchNorm2d + ReLU (+ residual add, + AvgPool2d) as called from
 * modeling/backbone/clip_backbone.py:57-70,193-219, layers/batch_norm.py:45-66,
 * modeling/proposal_generator/rpn.py:158-177, modeling/backbone/clipcap/clipcap.py:39-163.
 * y[m][n] = relu?( acc*scale[n] + bias[n] + residual[m][n] ), zeroed where relu_mask[m][n] <= 0 (ReLU backward);
 * pool=1: 1x1 conv over the 2x2 average-pooled input.  dgrad = the same entry point on weight_prep's w_dgrad.
 * out_f32: bit 0 = y is f32; bit 1 (bf16 kernels, with bit 0, no relu_mask, leading dims multiples of 8) = the residual rows are
 * f32 -- the mapper's residual stream x + f(LN(x)) stays f32 (clipcap.py:88-100) and its add rides in the GEMM epilogue;
 * bit 2 (leading dims multiples of 8, pooled tensor < 2 GiB) = `residual` is [Nimg][Ho/2][Wo/2][ldr] and row m adds a quarter
 * of its pooled pixel: the backward of the Bottleneck's AvgPool2d(stride) on the downsample path (clip_backbone.py:45-52). */
int cddmsl_conv_fwd(const void* x, const void* w, void* y, const float* scale, const float* bias, const void* residual,
                    const void* relu_mask, int Nimg, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad,
                    int pool, int ldy, int ldr, int ldm, int relu, int out_f32, int dtype, void* stream);
/* Device scratch for the weight-gradient kernels' split reductions (no counterpart in the reference: ATen's conv backward owns its
 * workspace, aten/src/ATen/native/cudnn).  With `bytes` of 16-byte-aligned device memory registered, a split reduction stores its
 * partial tiles there and a second kernel sums them into dW (deterministic; f32 atomics otherwise, and whenever the launch needs more
 * than `bytes`).  One workspace per process; launches that use it must be stream-ordered.  (nullptr, 0) unregisters. */
int cddmsl_set_workspace(void* ptr, long bytes);
/* dW[n][k] (f32, accumulated) += scale[n] * sum_m dY[m][n] * im2col(x)[m][k] */
int cddmsl_conv_wgrad(const void* x, const void* dy, float* dw, const float* scale, int Nimg, int Hi, int Wi, int Cin,
                      int Cout, int KH, int KW, int stride, int pad, int pool, int ldd, int dtype, void* stream);
/* batched GEMMs on the same kernels (strides in elements; used by the reassociated attention pool):
 *   nt: C_b[m][n] = sum_k A_b[m][k] B_b[n][k] (+ bias[n]);   tn: out_b[n][k] (+)= sum_m A_b[m][n] B_b[m][k]
 *   tn mode 0 = f32 atomic accumulate, 1 = f32 store, 2 = `dtype` store */
int cddmsl_gemm_nt_batched(const void* a, const void* w, void* c, const float* bias, int M, int N, int K, int lda, int ldb, int ldc,
                           int batch, long sa, long sw, long sc, int out_f32, int dtype, void* stream);
int cddmsl_gemm_tn_batched(const void* a, const void* b, void* out, int M, int N, int K, int lda, int ldb, int ldo, int batch, long sa,
                           long sb, long so, int mode, int dtype, void* stream);
/* diagnostic: the kernel the calling thread's last conv / GEMM entry point launched -- 1 k_conv_fwd (128x128 LDS-DMA),
 * 2 k_conv_fwd_reg (fused avg-pool loader), 3 k_conv_fwd256 (256x256 ping-pong), 4 k_conv_wgrad, 5 k_conv_wgrad_dma,
 * 6 k_wgrad256, 7 k_gemm_tn_stream, 8 k_conv3x3_small (few-channel 3x3 layers: the CLIP stem), 9 k_gemm_tn_small.  bench.py uses it to attribute HIP-event times to kernels. */
int cddmsl_last_kernel(void);
/* diagnostic: while on (per thread), the conv / GEMM entry points above choose their kernel (cddmsl_last_kernel) and return
 * without launching; bench.py asks this way BEFORE a launch whether it is the kernel whose launches it is timing, so only those
 * launches carry HIP events inside the timed region.  Returns the previous setting. */
int cddmsl_plan_only(int on);
/* f32 master -> `dtype` forward weights and flipped/transposed dgrad weights scaled by the FrozenBN scale */
int cddmsl_weight_prep(const float* w, const float* scale, void* w_fwd, void* w_dgrad, int Cout, int KH, int KW, int Cin,
                       int dtype, void* stream);
/* the same for `count` weights in one launch; device table of 8 x int64 per weight: {w, scale, w_fwd, w_dgrad (pointers, 0 = skip),
 * Cout, KH, KW, Cin}.  Used once per step after the optimizer update. */
int cddmsl_weight_prep_multi(const long long* table, int count, int dtype, void* stream);

/* ---- OCP e4m3 (fp8) configuration: BASELINE.json configs[4] ------------------------------------------------------------------
 * The reference has no fp8 path (AMP is off, config/defaults.py:697): these entry points have no counterpart there; they replace,
 * for the MFMA-bound forward convolutions of layer3 / layer4 / the RoI head's layer4 (clip_backbone.py:57-70) and for the region x
 * text-embedding contraction (roi_heads/fast_rcnn.py:546-572), the bf16 launches of cddmsl_conv_fwd / cddmsl_cosine_logits_fwd.
 *   cddmsl_quantize_fp8   x (src_dtype 0 bf16 / 1 f32, numel % 8 == 0) -> y e4m3 bytes = sat(x * scale[0]); scale (device, nullable = 1)
 *                         and amax (device, nullable): max|x| is recorded with an atomic max (delayed scaling, no host round trip)
 *   cddmsl_conv_fwd_fp8   e4m3 x [Nimg][Hi][Wi][Cin] * e4m3 w [Cout][KH][KW][Cin], stride 1 -> bf16 (f32 with out_f32) y with the
 *                         epilogue of cddmsl_conv_fwd (scale / bias f32 per channel -- the caller folds both dequantisation
 *                         factors into scale --, bf16 residual, ReLU, bf16 ReLU mask); v_mfma_scale_f32_32x32x64_f8f6f4, f32 accumulate.
 *                         Cin % 128 == 0, Cout % 256 == 0, <= 31 taps: other shapes are CDDMSL_ERR_ARG
 *   cddmsl_fp8_dot_nt     c [R][ldc] f32 (columns < N) = alpha[0] * a [R][K] . b [N][K]^T, N <= 32, K % 64 == 0
 *   y8 / q8 / amax8 (nullable together; cddmsl_conv_fwd_fp8 and cddmsl_conv_fwd_q8): a second output y8 [M][Cout] = e4m3 of
 *                         sat(y * q8[0]) for the NEXT convolution, written by the same epilogue (no separate quantisation pass);
 *                         max|y| goes to amax8.  Every amax buffer is 64 floats (atomics are spread by block; the owner takes the max).
 *   cddmsl_conv_fwd_q8    = cddmsl_conv_fwd for bf16 with that second output; only launches the 256x256 kernel takes
 *   cddmsl_conv_wgrad_fp8 dW[n][k] (f32, accumulated) += scale[n] * sum_m dy8[m][n] * im2col(x8)[m][k] on the e4m3 copies of both
 *                         operands (x8 [Nimg][Hi][Wi][Cin], dy8 rows ldd bytes apart); "same" convolutions (stride 1, 2 pad = KH - 1),
 *                         Cin % 256 == 0, Cout % 256 == 0 (cddmsl_conv_wgrad_fp8_ok returns 1), else CDDMSL_ERR_ARG; the caller folds
 *                         both dequantisation factors into scale.  Replaces, for the fp8 configuration, the weight-gradient half of
 *                         torch's conv2d backward behind modeling/backbone/clip_backbone.py:57-70 */
int cddmsl_quantize_fp8(const void* x, void* y, const float* scale, float* amax, long numel, int src_dtype, void* stream);
int cddmsl_conv_fwd_fp8(const void* x, const void* w, void* y, const float* scale, const float* bias, const void* residual,
                        const void* relu_mask, int Nimg, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int pad, int relu,
                        int out_f32, void* y8, const float* q8, float* amax8, void* stream);
int cddmsl_conv_fwd_q8(const void* x, const void* w, void* y, const float* scale, const float* bias, const void* residual,
                       const void* relu_mask, int Nimg, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride, int pad,
                       int relu, void* y8, const float* q8, float* amax8, void* stream);
int cddmsl_fp8_dot_nt(const void* a, const void* b, float* c, const float* alpha, int R, int N, int K, int ldc, void* stream);
int cddmsl_conv_wgrad_fp8_ok(int Cin, int Cout, int KH, int KW, int pad, int ldd);
int cddmsl_conv_wgrad_fp8(const void* x8, const void* dy8, float* dw, const float* scale, int Nimg, int Hi, int Wi, int Cin, int Cout,
                          int KH, int KW, int pad, int ldd, void* stream);

/* ---- RoIAlign  (layers/roi_align.py:49-65 -> torchvision.ops.roi_align; modeling/poolers.py:190-229) --------- */
/* y_pooled (nullable, [K][ph/2][pw/2][C], ph and pw even): AvgPool2d(2) of y, formed from the rounded outputs in the pooling
 * kernel's order -- what the first Bottleneck of the RoI head's layer4 reads on its downsample path (clip_backbone.py:45-52) */
int cddmsl_roi_align_forward(const void* x, const float* rois, void* y, void* y_pooled, int* dbg_grid, int N, int C, int H, int W,
                             int K, int ph, int pw, float spatial_scale, int sampling_ratio, int aligned, int dtype, void* stream);
int cddmsl_roi_align_backward(const void* dy, const float* rois, const int* roi_start, void* dx, float* ws_ay, float* ws_ax,
                              int* ws_fp, int N, int C, int H, int W, int K, int ph, int pw, float spatial_scale,
                              int sampling_ratio, int aligned, int dtype, void* stream);
/* The RoI head's entry (modeling/roi_heads/clip_roi_heads.py:113-115: pooler, then backbone.layer4, whose first Bottleneck
 * starts with a 1x1 conv and pools 2x2 on its downsample path, clip_backbone.py:45-52,57-70) with the 1x1 conv moved in FRONT of
 * the pooling -- RoIAlign is linear over pixels, a 1x1 conv over channels: roi_align(x) W = roi_align(x W) -- so the
 * [K][14][14][1024] crop tensor (3.3 GB at 8192 RoIs) is never formed:
 *   _affine:  y = relu?(scale[c] * roi_align(x)[.., c] + bias[c])  (FrozenBN + ReLU of that conv; scale / bias nullable),
 *             y nullable when only y_pooled (the downsample path's AvgPool2d(2) of the crops) is wanted;
 *   _backward_pooled: dy is the gradient of the pooled map [K][ph][pw][C] (RoIAlign grid 2ph x 2pw): roi_align_backward of the
 *             AvgPool2d backward of dy, without forming it;
 *             y8 / q8 / amax8 (nullable, bf16 only): e4m3 copy of y for the consuming convolution, as cddmsl_conv_fwd_q8. */
int cddmsl_roi_align_forward_affine(const void* x, const float* rois, void* y, void* y_pooled, const float* scale,
                                    const float* bias, int relu, int N, int C, int H, int W, int K, int ph, int pw,
                                    float spatial_scale, int sampling_ratio, int aligned, int dtype, void* y8, const float* q8,
                                    float* amax8, void* stream);
int cddmsl_roi_align_backward_pooled(const void* dy, const float* rois, const int* roi_start, void* dx, float* ws_ay,
                                     float* ws_ax, int* ws_fp, int N, int C, int H, int W, int K, int ph, int pw,
                                     float spatial_scale, int sampling_ratio, int aligned, int dtype, void* stream);
/* torchvision.ops.roi_align with the signature the reference calls (layers/roi_align.py:58-65 -> roi_align(input, rois, output_size,
 * spatial_scale, sampling_ratio, aligned)): input [N][C][H][W] (NCHW, any C), rois [K][5] (batch_idx, x0, y0, x1, y1) in ANY order ->
 * output [K][C][ph][pw]; backward: grad [K][C][ph][pw] -> grad_input [N][C][H][W] (what torchvision's autograd Function hands back).
 * They re-lay the operands channels-last in caller-provided scratch (RoIs ranked stably by image for the gather backward) and run
 * the kernels above: same arithmetic.  Call with temp == NULL to get *temp_bytes.  The training path keeps the channels-last,
 * grouped-by-image entry points (poolers.py:68-95 emits RoIs in that order anyway). */
int cddmsl_roi_align_nchw_anyorder(const void* input, const float* rois, void* output, int N, int C, int H, int W, int K, int ph, int pw,
                                   float spatial_scale, int sampling_ratio, int aligned, int dtype, void* temp, size_t* temp_bytes,
                                   void* stream);
int cddmsl_roi_align_backward_nchw_anyorder(const void* grad, const float* rois, void* grad_input, int N, int C, int H, int W, int K,
                                            int ph, int pw, float spatial_scale, int sampling_ratio, int aligned, int dtype, void* temp,
                                            size_t* temp_bytes, void* stream);

/* ---- RPN / matcher index stages --------------------------------------------------------------------------------
 * modeling/anchor_generator.py:161-228, modeling/box_regression.py:77-115, modeling/proposal_generator/rpn.py:514-533,
 * modeling/proposal_generator/proposal_utils.py:22-130, layers/nms.py:19-39 (torchvision nms),
 * structures/boxes.py:322-367 + modeling/matcher.py:61-126 */
int cddmsl_anchors(const float* cell, float* out, int Hf, int Wf, int A, float stride, float offset, void* stream);
/* stable descending sort of every row of keys_in [N][total] (torch.sort(descending=True, stable=True) per image,
 * proposal_utils.py:66-70): one device-wide radix sort over (image, score) composite keys; `offsets` is unused (dense rows);
 * temp == NULL returns the workspace size in *temp_bytes */
int cddmsl_sort_desc(const float* keys_in, float* keys_out, int* idx_scratch, int* order_out, const int* offsets, int N,
                     int total, void* temp, size_t* temp_bytes, void* stream);
int cddmsl_rpn_decode(const int* order, const float* deltas, const float* cell, const int* img_hw, float* boxes,
                      unsigned char* valid, int N, int Hf, int Wf, int A, int topk, float stride, float offset, float wx,
                      float wy, float ww, float wh, float scale_clamp, float min_size, void* stream);
int cddmsl_nms(const float* boxes, const unsigned char* valid, unsigned long long* mask_ws, int* keep, int* nkeep, int N,
               int n, float thr, int max_keep, void* stream);
/* torchvision.ops.nms as the reference calls it (layers/nms.py:6-7,30,35): boxes [K][4] and scores [K] in ANY order ->
 * keep [K] int64 = indices of the kept boxes in descending-score order (entries past *nkeep are -1), nkeep [1] (device).
 * Scratch from the caller: call with temp == NULL to get *temp_bytes.  K <= 12288. */
int cddmsl_nms_anyorder(const float* boxes, const float* scores, long* keep, int* nkeep, int K, float iou_threshold, void* temp,
                        size_t* temp_bytes, void* stream);
int cddmsl_iou_match(const float* gt, int G, const float* preds, int P, long* matches, signed char* labels,
                     unsigned int* best_ws, int nthr, float t0, float t1, int l0, int l1, int l2, int allow_low_quality,
                     void* stream);

/* the same for all images of a batch in one launch pair: boxes concatenated with offsets gt_off[N+1]; predictions shared by all
 * images (pred_off NULL: the anchors, outputs [N][P]) or concatenated with offsets pred_off[N+1] (proposals, outputs [sum P]) */
int cddmsl_iou_match_batched(const float* gt, const int* gt_off, const float* preds, const int* pred_off, long* matches,
                             signed char* labels, unsigned int* best_ws, int N, int P, int maxG, int totalG, int nthr, float t0,
                             float t1, int l0, int l1, int l2, int allow_low_quality, void* stream);

/* ---- CLIP attention pool (modeling/backbone/clip_backbone.py:83-107): token build/backward; the query-0 attention itself is
 * reassociated into the batched GEMMs above (cddmsl_amd/layers.py::AttnPoolFn) ------------------------------------- */
int cddmsl_attn_tokens_fwd(const void* x, const float* pos, void* tok, int K, int P, int TP, int C, int dtype, void* stream);
/* the same, also writing mbits [K][C] (64-bit words, P <= 64): bit p = (x[k][p][col] > 0), the ReLU mask of the pooled map in the
 * form cddmsl_attnpool_dx reads */
int cddmsl_attn_tokens_fwd_mask(const void* x, const float* pos, void* tok, unsigned long long* mbits, int K, int P, int TP, int C,
                                int dtype, void* stream);
/* The pool's input gradient in ONE pass (bf16, P = 49, TP = 56, C % 128 == 0): the batched product  dtok[k] = [p ; ds][k]^T . [dZ ; U][k]
 * (pds [K][H2][TP], zu [K][H2][C], H2 = 2 x heads <= 64) with the epilogue  dx[k][t-1] = dtok[t] + (dtok[0] + g0[k]) / P  for t = 1..P,
 * zeroed where bit t-1 of mbits[k][col] is clear, and  gpos[t] += sum_k dtok[k][t]  (t = 0: + g0[k]; f32 atomics, nullable) --
 * replaces the stored token gradients + cddmsl_attn_tokens_bwd.  g0 [K][C] f32: the query path's gradient of the mean token. */
int cddmsl_attnpool_dx(const void* pds, const void* zu, const float* g0, const unsigned long long* mbits, void* dx, float* gpos, int K,
                       int H2, int P, int TP, int C, int dtype, void* stream);
/* relu_mask (nullable, [K][P][C]): dx is zeroed where it is <= 0 -- the pooled map when it is a ReLU output (layer4 -> attnpool,
 * clip_roi_heads.py:160-165), so the stage's ReLU backward needs no pass of its own.  gpos (nullable, f32 [P+1][C]) accumulates
 * (atomics) the positional embedding's gradient sum_k dtok[k][t][:] in the same pass; dx may be NULL when only gpos is wanted */
int cddmsl_attn_tokens_bwd(const void* dtok, const void* relu_mask, void* dx, float* gpos, int K, int P, int TP, int C, int dtype,
                           void* stream);
/* softmax of the query-0 scores (clip_backbone.py:95-105, F.multi_head_attention_forward's softmax over the 50 keys) between the
 * batched products: S [K][H][TP] f32 -> p [K][H][P1] f32 = softmax(S[..., :P1] * scale) and pT [K][TP][H] (dtype; zero rows past
 * P1).  Backward: ds = p * (dP - sum p dP) * scale -> dsT [K][TP][H] and pds [K][2H][TP] = [p ; ds] (dtype; zero columns past P1) */
int cddmsl_attnpool_softmax_fwd(const float* S, float* p, void* pT, long K, int H, int P1, int TP, float scale, int dtype, void* stream);
int cddmsl_attnpool_softmax_bwd(const float* p, const float* dP, void* dsT, void* pds, long K, int H, int P1, int TP, float scale,
                                int dtype, void* stream);

/* ---- fused multi-head attention for short sequences: the ClipCap mapper's softmax(QK^T * scale) V (no mask), 80 tokens x 8
 * heads of 96 (modeling/backbone/clipcap/clipcap.py:59-83).  bf16 only (dtype 0), dh == 96, t <= 96; element (s, i, h, c) of
 * q / k / v / o sits at ((s*t + i) * ld + h*dh + c); the backward recomputes the probabilities from q, k. */
int cddmsl_attn_small_fwd(const void* q, const void* k, const void* v, void* o, int nseq, int t, int heads, int dh, int ldq,
                          int ldk, int ldv, int ldo, float scale, int dtype, void* stream);
int cddmsl_attn_small_bwd(const void* q, const void* k, const void* v, const void* dout, void* dq, void* dk, void* dv, int nseq,
                          int t, int heads, int dh, int ldq, int ldk, int ldv, int ldo, float scale, int dtype, void* stream);
/* the mapper's LAST layer evaluated for the last token only (v2l keeps one of the mapped tokens, clipcap.py:714-719; attention of
 * clipcap.py:59-83 with ONE query row per sequence): p [n][heads][t] f32 = softmax(q . K^T * scale), o [n][ldo] = p . V.  q [n][ldq],
 * kv [n*t][ldkv] bf16 with K of head h at column h*dh and V at column voff + h*dh; bf16 only (dtype 0), t <= 128, dh <= 128, dh % 8 == 0.
 * Backward: dq [n][ldq], dkv [n*t][ldkv] (every element written) from dout [n][ldo] and the saved p. */
int cddmsl_attn_last_fwd(const void* q, const void* kv, void* o, float* p, int n, int t, int heads, int dh, int ldq, int ldkv, int voff,
                         int ldo, float scale, int dtype, void* stream);
int cddmsl_attn_last_bwd(const void* q, const void* kv, const void* dout, const float* p, void* dq, void* dkv, int n, int t, int heads,
                         int dh, int ldq, int ldkv, int voff, int ldo, float scale, int dtype, void* stream);

/* ---- fp32 heads: cosine-logit classifier (modeling/roi_heads/fast_rcnn.py:546-572) and the contrastive loss over
 * the cosine-similarity matrix (modeling/meta_arch/rcnn.py:308-317,458-468) ------------------------------------ */
int cddmsl_l2norm_fwd(const float* x, float* y, float* inv, long R, int D, float eps, void* stream);
int cddmsl_l2norm_bwd(const float* dy, const float* y, const float* inv, float* dx, long R, int D, void* stream);
int cddmsl_cosine_logits_fwd(const float* x, const float* wn, float* scores, float* inv, long R, int D, int Kc,
                             float temperature, float eps, void* stream);
int cddmsl_cosine_logits_bwd(const float* ds, const float* x, const float* wn, const float* inv, float* dx, long R, int D,
                             int Kc, float temperature, int accumulate, void* stream);
/* mapper LayerNorm (modeling/backbone/clipcap/clipcap.py:97-100; frozen affine) and the focal-scaled, background-weighted
 * classification loss (modeling/roi_heads/fast_rcnn.py:624-644): row_loss = CE * (1-p_t)^gamma * w; probs saved for backward */
int cddmsl_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, long R, int D,
                         float eps, int dtype, void* stream);
int cddmsl_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                         long R, int D, int accumulate, int dtype, void* stream);
int cddmsl_focal_ce_fwd(const float* logits, const long* target, float* row_loss, float* probs, long R, int C, float gamma,
                        int bg_class, float bg_weight, void* stream);
int cddmsl_focal_ce_bwd(const float* logits, const long* target, const float* probs, const float* gscale, float* dlogits, long R,
                        int C, float gamma, int bg_class, float bg_weight, void* stream);
int cddmsl_contrastive_fwd(const float* S, float* rlse, float* clse, float* loss, int n, int ld, void* stream);
int cddmsl_contrastive_bwd(const float* S, const float* rlse, const float* clse, const float* gloss, float* dS, int n, int ld,
                           void* stream);
/* Losses over SAMPLED rows with known index lists (no dense pass, no boolean-mask gathers):
 *   cddmsl_rpn_losses: RPN.losses (modeling/proposal_generator/rpn.py:365-429; _dense_box_regression_loss box_regression.py:229-270, smooth-L1 beta 0):
 *     out2 = (sum_{sampled} BCE-with-logits, sum_{positives} |delta - get_deltas(anchor, matched gt)|) * inv_norm.  logits [N*A], deltas [N*A][4],
 *     pos / neg = global anchor indices (image * A + anchor), midx [N*A] matched gt index within the image, gt [G][4], gt_off [N], anchors [A][4].
 *   cddmsl_box_l1: FastRCNNOutputLayers.box_reg_loss (modeling/roi_heads/fast_rcnn.py:646-689): out1 = sum_{fg rows} |delta[4 cls[r] ..] - get_deltas(src, tgt)| * inv_norm;
 *     cls nullable (class-agnostic).  Backward = the same entry point with gout given: gradients are scattered into CALLER-ZEROED tensors. */
int cddmsl_rpn_losses(const float* logits, const float* deltas, const long* pos, int npos, const long* neg, int nneg, const long* midx,
                      const float* gt, const long* gt_off, const float* anchors, long A, float wx, float wy, float ww, float wh, float inv_norm,
                      float* out2, const float* gout2, float* dlogits, float* ddeltas, void* stream);
int cddmsl_box_l1(const float* deltas, int ld, const long* fg, int nfg, const long* cls, const float* src, const float* tgt, float wx, float wy,
                  float ww, float wh, float inv_norm, float* out1, const float* gout1, float* ddeltas, void* stream);

/* ---- elementwise: preprocessing (modeling/meta_arch/rcnn.py:161-179,758-768, structures/image_list.py:72-124),
 * AvgPool2d(2) (clip_backbone.py:36,46,147), ReLU backward, column sums, fused clip+SGD (solver/build.py:59-130) -- */
int cddmsl_preprocess(const unsigned char* img, void* out, int n, int h, int w, int Hp, int Wp, int Cp, const float* mean3,
                      const float* std3, int div255, int dtype, void* stream);
int cddmsl_preprocess224(const unsigned char* img, void* out, int n, int h, int w, int Hp, int Wp, int RH, int RW, int top,
                         int left, int S, int Cp, const float* mean3, const float* std3, int dtype, void* stream);
/* the same two for a whole batch in one launch: imgs / hs / ws are HOST arrays of N device pointers / heights / widths (the images are
 * separate tensors of different sizes: ImageList.from_tensors, structures/image_list.py:72-124); image j is written to out[j] */
int cddmsl_preprocess_batch(const unsigned char* const* imgs, const int* hs, const int* ws, int N, void* out, int Hp, int Wp, int Cp,
                            const float* mean3, const float* std3, int div255, int dtype, void* stream);
int cddmsl_preprocess224_batch(const unsigned char* const* imgs, const int* hs, const int* ws, int N, void* out, int Hp, int Wp, int RH,
                               int RW, int top, int left, int S, int Cp, const float* mean3, const float* std3, int dtype, void* stream);
int cddmsl_avgpool2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);
int cddmsl_avgpool2_bwd(const void* dy, const void* mask, const void* add, void* dx, int N, int H, int W, int C, int dtype,
                        void* stream);
/* fp8 configuration: the same pass (bf16 only) with a second output, y8 = e4m3 of sat(dx * q8[0]), max|dx| recorded in amax8 (64 floats) */
int cddmsl_avgpool2_bwd_q8(const void* dy, const void* mask, const void* add, void* dx, int N, int H, int W, int C, void* y8,
                           const float* q8, float* amax8, void* stream);
/* stock Detectron2 R50-C4 pieces (config #1): BasicStem max-pool (modeling/backbone/resnet.py:355-358), the input-gradient
 * scatter of stride-2 1x1 convs (STRIDE_IN_1X1 bottlenecks, resnet.py:100-210), Res5ROIHeads mean pool (roi_heads.py:487) */
int cddmsl_maxpool3s2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);
int cddmsl_upsample_zero2(const void* t, const void* mask, const void* add, void* dx, int N, int H, int W, int C, int dtype,
                          void* stream);
int cddmsl_meanpool_fwd(const void* x, float* y, long K, int P, int C, int dtype, void* stream);
int cddmsl_meanpool_bwd(const float* dy, void* dx, long K, int P, int C, int dtype, void* stream);
int cddmsl_relu_bwd(const void* g, const void* y, void* dx, long numel, int g_f32, int dtype, void* stream);
int cddmsl_colsum(const void* x, float* out, long rows, int cols, int period, int dtype, void* stream);
int cddmsl_sgd_clip_step(float** params, const float** grads, float** moms, const long* sizes, int count, float* norm_ws,
                         float lr, float momentum, float wd, float clip, int first_step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CDDMSL_HIP_H */
