/* C ABI of libeod_hip.so  --  the MI355X (gfx950) kernels of the embodied-detector hot path.
 *
 * The reference (nhcha6/embodied-object-detection) has NO native code on this path: every op is a stock
 * PyTorch / torchvision / detectron2 call made from Python.  There is therefore no
 * reference FFI to mirror; each entry point below cites the reference Python call site whose arithmetic it
 * replaces.  Contract for every function:
 *   - the caller owns every buffer (device pointers, allocated by the host framework); kernels never
 *     allocate, free or synchronise; work is enqueued on `stream`;
 *   - returns 0 on success, a negative EOD_ERR_* code on bad dimensions / misalignment / launch failure;
 *     never throws;
 *   - thread-compatible (one host thread per process / GPU, as detectron2's `launch` runs it,
 *     Detic/train_mp3d.py:850).
 * All activations are NHWC fp32.  "count" pointers are device ints holding a dynamic number of valid rows
 * (ROIs / detections) so that data-dependent sizes never force a host sync.
 */
#ifndef EOD_HIP_H
#define EOD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* eod_stream_t; /* hipStream_t */

#define EOD_OK 0
#define EOD_ERR_BAD_DIMS (-1)
#define EOD_ERR_ALIGN (-2)
#define EOD_ERR_LAUNCH (-3)
#define EOD_ERR_NULL (-4)
#define EOD_ERR_CAPACITY (-5)

/* Batches (B independent scenes in lock-step, BASELINE configs[4]; Detic/SMNet/loader.py:289-293: only independent sequences may be
 * batched).  Every entry point with a `batch` argument (or descriptor field) treats EVERY buffer of the call -- inputs, outputs,
 * counts, workspaces, recurrent state -- as `batch` single-scene buffers laid back to back, scene b at offset b x (single-scene
 * size), and runs the B problems in ONE launch per stage; batch <= 1 is the single-scene call.  Weights are shared.  Results of
 * scene b are bitwise those of the single-scene call on scene b's buffers. */
#define EOD_MAX_BATCH 8
#define EOD_MAX_LEVELS 40   /* 5 pyramid levels x EOD_MAX_BATCH scenes */

int eod_abi_version(void);

/* ---- dense contraction: implicit-GEMM convolution / linear layer on fp32 MFMA ------------------------
 * Replaces every nn.Conv2d / nn.Linear / ConvTranspose2d(2,2) of the path: timm ResNet-50
 * (Detic/detic/modeling/backbone/timm.py:277-299), FPN lateral/output convs (timm.py:118-136), memory 1x1
 * projection + scale + sum fusion (timm.py:174-186), LastLevelP6P7_P5 (timm.py:359-364), CenterNetHead convs
 * (centernet_head.py:141-161), FastRCNNConvFCHead / bbox_pred / cls linear
 * (detic_fast_rcnn.py:437-466, zero_shot_classifier.py:78), mask head convs + deconv (d2
 * MaskRCNNConvUpsampleHead via detic_roi_heads.py:257,268).
 *   y[m][n] = act( (sum_k A[m][k] * w[n][k] + bias[n]) * out_scale + res[m][n] )
 * A is the im2col view of x (NHWC), k ordered (ky, kx, c); w is [Cout][Kpad] with Kpad % 32 == 0.          */
typedef struct EodConvDesc {
  const float* x;      /* [N,H,W,Cin] */
  const float* w;      /* [Cout,Kpad] */
  const float* bias;   /* [Cout] or NULL ([Cout/4] for out_mode 1) */
  const float* res;    /* residual or NULL */
  float* y;            /* [N,OH,OW,Cout] (out_mode 0) / [N,2OH,2OW,Cout/4] (out_mode 1) */
  float* workspace;    /* split-K slabs, >= workspace_bytes */
  size_t workspace_bytes;
  const int32_t* m_count; /* optional device int: number of valid units (each m_unit output rows) */
  int32_t m_unit;
  /* 0 / 1, or B: the rows are B independent unit lists back to back (B scenes in lock-step, each N / B images); m_count then
   * holds B counts and list b has work in its first m_count[b] units only.  Rows without work are neither computed nor written. */
  int32_t m_segments;
  int32_t N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, Kpad;
  int32_t relu;      /* ReLU on the output */
  int32_t res_mode;  /* 0 none, 1 same-shape add, 2 add nearest-x2-upsampled res [N,OH/2,OW/2,Cout] */
  int32_t in_relu;   /* ReLU applied to x on load (p7 = conv(relu(p6)), timm.py:362) */
  int32_t out_mode;  /* 0 NHWC, 1 ConvTranspose2d(k2,s2) scatter: n = (dy*2+dx)*Cout/4 + co, 2 = mode 1 fused with the mask
                        predictor (see fuse_w below) */
  int32_t tap4;      /* 1: Cin == 4 (stem, RGB padded to 4): one float4 per tap */
  int32_t force_tile; /* 0 auto, else tile + 10 * variant: tile 1=128x128 2=128x64 3=64x64; variant 0 default, 1 BK=32, 2 BK=64,
                         5 bf16x3 split math; 6 / 7: the 32x32-tile kernel that splits K over 4 / 8 waves (benchmarks/tests) */
  int32_t force_splitk; /* 0 auto */
  float out_scale;
  /* pyramid mode (levels > 0): x / y are [level_off[levels], C] row lists, level l is a level_h[l] x level_w[l] image;
   * stride-1 'same' convolution with weights shared across levels (CenterNetHead, centernet_head.py:141-161). N must be 1.
   * A batch of B scenes in lock-step passes its 5 B level images as 5 B levels (EOD_MAX_LEVELS) and plan_rows = one scene's rows. */
  int32_t levels;
  int32_t level_off[EOD_MAX_LEVELS + 1];
  int32_t level_h[EOD_MAX_LEVELS];
  int32_t level_w[EOD_MAX_LEVELS];
  /* out_mode 2 -- tail of the mask head in one launch (d2 MaskRCNNConvUpsampleHead deconv + ReLU -> predictor 1x1 conv to one
   * channel, then mask_rcnn_inference's sigmoid; via detic_roi_heads.py:257,268 and custom_rcnn.py:574):
   *   y[u][2oy+dy][2ox+dx] = sigmoid( sum_co relu(deconv(x)[img][2oy+dy][2ox+dx][co]) * fuse_w[co] + fuse_b )
   * y is [units, 2OH, 2OW] probabilities; Cout/4 must be 256; u = out_units ? out_units[img] : img (scatter of a compact
   * ROI list back to per-proposal rows); the 4 x 256-channel deconv activation never goes to memory. */
  const float* fuse_w;
  const int32_t* out_units;
  float fuse_b;
  /* optional: the same weights pre-split for the bf16x3 kernels (eod_conv_split_weights_bf16x3), Cout * Kpad * 6 bytes; used by
   * the 256x128 bf16x3 kernel instead of splitting w on the fly; ignored by every other kernel */
  const void* w_split;
  /* 0, or the number of output rows the tile / split-K plan is made for instead of this call's: a batch of N images planned
   * like ONE image walks K exactly as the single-image call does (same split-K slabs, same summation order), so its results
   * are bitwise those of N separate calls (modeling/batched.py). */
  int32_t plan_rows;
  /* 0, or bytes of LDS the fp32 kernel's launch allocates on top of its tiles and never touches.  LDS is what bounds the
   * kernel's workgroups per CU (18 KB tiles: 8 per 160 KB), so a reserve caps them (e.g. 8 KB -> 6 per CU) and leaves wave slots,
   * registers and LDS on every CU for the small latency-bound kernels other streams launch meanwhile: a grid of thousands of
   * ~150 us workgroups that fills every slot makes a concurrent one-workgroup kernel wait for a slot for ~100 us. */
  int32_t lds_reserve;
  /* optional: the layer is followed by GroupNorm (CenterNet tower, centernet_head.py:76-79).  When the layer's plan splits K into
   * slabs (eod_conv2d_gn_fused(d) == 1), the slab reduce also writes the partial sums eod_groupnorm_relu's statistics launch would
   * write (same chunks, same order: bitwise), into the partial-sum area of that call's `stats` workspace
   * (eod_groupnorm_partial_offset); eod_groupnorm_relu is then called with partial_ready = 1.  Pyramid mode only. */
  double* gn_partial;
  int32_t gn_groups;
  /* optional: two linear layers that read the same input as ONE contraction (the box predictor's cls_score.linear and bbox_pred.0,
   * detic_fast_rcnn.py:437-466): w = [w_a; w_b] stacked along Cout, bias likewise; output columns [0, split_n) go to y
   * [rows, split_n] WITHOUT the ReLU, columns [split_n, Cout) to y2 [rows, Cout - split_n] with it (if relu).  out_mode 0,
   * res_mode 0, no pyramid mode.  Every output column walks K exactly as in a separate call: bitwise the two layers' results. */
  float* y2;
  int32_t split_n;
  /* pipeline variant of the 64x64 fp32 kernel (same results, bitwise): 0 = one LDS buffer, one chunk of register prefetch, two
   * barriers per chunk (default); 1 = two chunks of operands in flight (one more set of prefetch registers); 2 = the operand
   * tiles double buffered in LDS, one barrier per chunk (36 KB per workgroup). */
  int32_t prefetch2;
  /* optional [N,OH,OW,Cout] (out_mode 0, no split_n / gn_partial): after everything else the output is 0 where gate <= 0 -- the
   * backward of a ReLU whose output is `gate`, applied by the input-gradient convolution that produces the gradient (with res_mode 1
   * = the skip connection's gradient: y = relu'(gate) * (conv(x) + res), a bottleneck block's timm.py:277-299 backward in one
   * launch instead of conv, add, mask). */
  const float* gate;
} EodConvDesc;
int eod_conv2d(const EodConvDesc* d, eod_stream_t stream);
int eod_conv2d_gn_fused(const EodConvDesc* d); /* 1 when this layer can carry gn_partial (its plan has a slab reduce), else 0 */
size_t eod_conv2d_workspace_bytes(const EodConvDesc* d);
/* Arithmetic of eod_conv2d when force_tile == 0 (process-wide, read at every call; initial value from the environment variable
 * EOD_CONV_MATH = fp32 | bf16x3):
 *   EOD_MATH_FP32   (0, default) fp32 matrix-core FMAs (v_mfma_f32_32x32x2_f32): the reference's arithmetic class;
 *   EOD_MATH_BF16X3 (1) every fp32 operand split into three bf16 pieces, six bf16 MFMAs per product term set, fp32 accumulate:
 *                   fp32-class accuracy (error vs an fp64 convolution within 2x of the fp32 path's, tests/test_kernels_gpu.py)
 *                   at 16/6 of the fp32-MFMA ceiling.  The 7x7 stem and in_relu convs stay on the fp32 kernel.
 * Returns the previous mode, or EOD_ERR_BAD_DIMS for an unknown one.  Nothing in the reference to mirror (build-defined). */
#define EOD_MATH_FP32 0
#define EOD_MATH_BF16X3 1
int eod_set_conv_math(int mode);
int eod_get_conv_math(void);
/* w [Cout][Kpad] fp32 -> out [Cout][Kpad/32][ xh(32) | xm(32) | xl(32) ] bf16 (three round-to-nearest bf16 pieces of every
 * weight, 192 bytes per 32-wide K chunk = the LDS row image of the bf16x3 kernels).  Static weights are split once. */
size_t eod_conv_split_weights_bytes(int Cout, int Kpad);
int eod_conv_split_weights_bf16x3(const float* w, int Cout, int Kpad, void* out, eod_stream_t stream);

/* ---- small dense / elementwise ops -------------------------------------------------------------------- */
/* d2 GeneralizedRCNN.preprocess_image (custom_rcnn.py:557): u8 CHW RGB -> (x-mean)/std, NHWC4 (4th channel
 * 0), zero padded to [Hp,Wp]. */
int eod_preprocess_image(const uint8_t* img_chw, float* out_nhwc4, int H, int W, int Hp, int Wp,
                         const float* mean3, const float* std3, eod_stream_t stream);
/* timm ResNet maxpool 3x3 s2 p1 (timm.py:281) */
int eod_maxpool3x3s2(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, eod_stream_t stream);
/* GroupNorm(32)+ReLU over the 5 concatenated FPN levels (centernet_head.py:76-79); x,y [P,C], level l owns
 * rows [level_off[l], level_off[l+1]). stats: 8-byte aligned workspace of eod_groupnorm_workspace_bytes() bytes
 * (mean/rstd per (level, group) followed by per-chunk double partial sums). */
size_t eod_groupnorm_workspace_bytes(const int32_t* level_off_host, int levels, int groups);
int eod_groupnorm_relu(const float* x, float* y, const float* gamma, const float* beta, const int32_t* level_off_host,
                       int levels, int C, int groups, float eps, float* stats, int partial_ready, eod_stream_t stream);
/* byte offset of the per-chunk partial sums inside `stats` (EodConvDesc.gn_partial points there); partial_ready = 1: the producing
 * eod_conv2d wrote them, the statistics launch is skipped */
size_t eod_groupnorm_partial_offset(int levels, int groups);
/* Backward of eod_groupnorm_relu (training slices): x / y the forward's input / output, dy = dL/dy, fwd_stats the
 * forward call's `stats` workspace (its partial sums give mean / rstd); -> dx [P,C], dgamma [C], dbeta [C] (sums over all levels: the
 * parameters are shared).  workspace >= eod_groupnorm_backward_workspace_bytes(); three launches, deterministic summation order. */
size_t eod_groupnorm_backward_workspace_bytes(const int32_t* level_off_host, int levels, int C);
int eod_groupnorm_relu_backward(const float* x, const float* y, const float* dy, const float* gamma, const int32_t* level_off_host,
                                int levels, int C, int groups, float eps, const float* fwd_stats, void* workspace, float* dx,
                                float* dgamma, float* dbeta, eod_stream_t stream);
/* mask predictor 1x1 conv -> 1 channel + sigmoid (d2 mask head predictor + mask_rcnn_inference,
 * custom_rcnn.py:574): x [R*784,C] -> prob [R*784] */
int eod_mask_predictor_sigmoid(const float* x, const float* w, float bias, float* prob, int rows, int C,
                               const int32_t* unit_count, int unit_rows, const int32_t* out_units /* optional scatter of units */,
                               eod_stream_t stream);

/* ---- ROIAlignV2 over p3..p5 (d2 ROIPooler, detic_roi_heads.py:332,265) --------------------------------
 * batch > 1: p3..p5 hold `batch` images ([batch,h,w,C]), boxes holds batch x boxes_per_image boxes and box j is pooled from image
 * j / boxes_per_image.  Without box_rows: R_cap = batch x boxes_per_image ROIs, one list per image, count[batch].  With box_rows:
 * ONE compact list of box indices over all images (eod_concat_lists), count[1]. */
/* refine (optional, NULL = off; not with box_rows): ROI r pools the box apply_deltas(boxes[r], deltas[r]) -- the cascade's next-stage
 * proposals (Box2BoxTransform.apply_deltas + clip, detic_roi_heads.py:121-122,314) without a launch of their own -- and the refined
 * boxes are also written to boxes_out [R,4] (bitwise what eod_apply_deltas writes). */
typedef struct EodBoxRefine {
  const float* deltas; /* [R, ld] */
  int32_t ld;
  float wx, wy, ww, wh;
  int32_t clip;
  float img_w, img_h;
  float* boxes_out;    /* [R,4] */
} EodBoxRefine;
int eod_roi_align(const float* p3, const float* p4, const float* p5, int h3, int w3, int C,
                  const float* boxes /*[R,4]*/, const int32_t* box_rows /* optional gather: ROI r pools boxes[box_rows[r]] */,
                  const int32_t* count, int R_cap, int out_size, float* out /*[R,S,S,C]*/, int batch, int boxes_per_image,
                  const EodBoxRefine* refine, eod_stream_t stream);

/* Backward of eod_roi_align for one image (training slices; torchvision roi_align backward under d2's ROIPooler,
 * detic_roi_heads.py:332,265): g [R,S,S,C] = dL/d(out); the gradient is ADDED into dp3..dp5 (the pyramid levels' gradients,
 * [h,w,C] each; zero them first or let several poolers accumulate) with fp32 atomics -- summation order is not fixed, as in torch. */
int eod_roi_align_backward(float* dp3, float* dp4, float* dp5, int h3, int w3, int C, const float* boxes /*[R,4]*/,
                           const int32_t* count, int R_cap, int out_size, const float* g, eod_stream_t stream);

/* Training losses of the proposal generator with their gradients (training slices): CenterNet.losses for ONLY_PROPOSAL +
 * WITH_AGN_HM + NOT_NORM_REG (centernet/modeling/dense_heads/centernet.py:241-318) = binary_heatmap_focal_loss
 * (layers/heatmap_focal_loss.py:52-84) on the agnostic logits + IOULoss 'giou' (layers/iou_loss.py:10-64) on
 * relu(scale_l * bbox_pred) (centernet_head.py:141-161).  The targets (centernet.py:137-239) are inputs. */
typedef struct EodCenterNetLossDesc {
  const float* head_out;      /* [P, head_stride]: col 0 agn_hm logit, cols 1..4 bbox_pred (pre scale / relu) -- the rows eod_centernet_proposals reads */
  int32_t head_stride;        /* >= 5 */
  int32_t P;
  int32_t levels;             /* <= 8 */
  int32_t level_off[9];       /* row offsets of the levels, level_off[levels] == P */
  float level_scale[8];       /* the levels' Scale parameters */
  const float* agn_heatmap;   /* [P] flattened_hms.max(dim=1) */
  const float* reg_targets;   /* [P,4] (left, top, right, bottom); rows whose max is < 0 carry no target (-INF in the reference) */
  const int32_t* pos_inds;    /* [n_pos] rows of the positive locations (a row may appear more than once) */
  int32_t n_pos;
  float hm_focal_alpha, hm_focal_beta, loss_gamma, sigmoid_clamp, ignore_high_fp;    /* MODEL.CENTERNET.* (0.25, 4, 2, 1e-4, 0.85) */
  float pos_weight, neg_weight, reg_weight;                                          /* 0.5, 0.5, 1 in the recurrent yaml */
  float num_pos_avg;          /* max(all-reduced n_pos / world size, 1)  (centernet.py:259-265) */
  float reg_norm;             /* max(all-reduced number of regression rows / world size, 1)  (centernet.py:288-293) */
  /* Or the counts straight from device memory (no host round trip behind eod_centernet_targets): counts_total != NULL replaces
   * num_pos_avg / reg_norm by max(counts_total[0] / world_size, 1) and max(counts_total[1] / world_size, 1) -- counts_total =
   * eod_centernet_targets' `counts`, all-reduced over the ranks -- and n_pos becomes the CAPACITY of pos_inds, of which
   * counts_local[0] entries are read. */
  const int32_t* counts_local;
  const int32_t* counts_total;
  float world_size;
  float* d_head_out;          /* [P, head_stride] dL/d(head_out) of loss_centernet_loc + _agn_pos + _agn_neg; every column written */
  float* losses;              /* [3]: loss_centernet_loc, loss_centernet_agn_pos, loss_centernet_agn_neg */
  void* workspace;
  size_t workspace_bytes;     /* >= eod_centernet_loss_workspace_bytes() */
} EodCenterNetLossDesc;
size_t eod_centernet_loss_workspace_bytes(void);
int eod_centernet_loss(const EodCenterNetLossDesc* d, eod_stream_t stream);

/* CenterNet target assignment for ONLY_PROPOSAL, one image (centernet.py:342-479: _get_ground_truth, _get_label_inds and their
 * helpers): from the ground-truth boxes to what eod_centernet_loss consumes.  Bit-exact with the reference's fp32 arithmetic except
 * the heatmap's expf (<= 2 ulp). */
typedef struct EodCenterNetTargetDesc {
  const float* gt_boxes;      /* [n_boxes,4] x1,y1,x2,y2 on the device */
  int32_t n_boxes;            /* 0 .. 4096 */
  int32_t levels;             /* <= 8 */
  int32_t level_off[9];       /* rows of the pyramid's row list */
  int32_t level_w[8];
  int32_t level_stride[8];    /* MODEL.CENTERNET.FPN_STRIDES */
  float soi_lo[8], soi_hi[8]; /* MODEL.CENTERNET.SOI */
  double hm_min_overlap;      /* 0.8 */
  double min_radius;          /* 4 */
  float* agn_heatmap;         /* [P] */
  float* reg_targets;         /* [P,4] in units of the level's stride; -1e8 / stride where no object claims the position */
  int32_t* pos_inds;          /* [n_boxes * levels] positive rows, box-major / level-minor */
  int32_t* counts;            /* [2]: number of positives, number of positions with a regression target */
} EodCenterNetTargetDesc;
int eod_centernet_targets(const EodCenterNetTargetDesc* d, eod_stream_t stream);

/* Training losses of one cascade stage's box head with their gradients: DeticFastRCNNOutputLayers.losses for USE_SIGMOID_CE +
 * CLS_AGNOSTIC_BBOX_REG (detic_fast_rcnn.py:157-197): sigmoid_cross_entropy_loss (:200-233; class_weight [C] = federated-loss mask x
 * zero-frequency mask, or NULL) and box_reg_loss (:270-303, smooth_l1; beta 0 = L1) against Box2BoxTransform.get_deltas(proposal, gt)
 * with the stage's weights.  scores [B, ld] logits (columns 0..C, C = background), gt_classes [B] in [0, C]; every column of d_scores
 * is written (background and padding: 0).  losses [2] = loss_cls, loss_box_reg.  The sampled / matched proposals are inputs. */
size_t eod_fast_rcnn_loss_workspace_bytes(int B);
int eod_fast_rcnn_loss(const float* scores, int ld, const float* deltas /*[B,4]*/, const float* proposal_boxes, const float* gt_boxes,
                       const int32_t* gt_classes, const float* class_weight, int B, int num_classes, float wx, float wy, float ww,
                       float wh, float smooth_l1_beta, float* d_scores, float* d_deltas, float* losses, void* workspace,
                       size_t workspace_bytes, eod_stream_t stream);

/* The ROI heads' half of the training forward (custom_rcnn.py:642-650 -> detic_roi_heads.py:226-240, 88-147): proposal matching,
 * labelling and sampling, and the classifier's logits -- what stands between the proposals and eod_fast_rcnn_loss.
 * eod_match_label: detectron2's pairwise_iou + Matcher([iou_thresh], [0, 1]) + the labelling of ROIHeads._sample_proposals /
 * CascadeROIHeads._match_and_label_boxes (called at detic_roi_heads.py:232 and :115): proposal i takes the first ground-truth box of
 * maximal IoU; out_classes[i] = its class if that IoU >= iou_thresh, else num_classes (background); out_gt_boxes[i] = the matched box.
 * G = 0: every proposal is background with a zero box.  Bit-exact with torch's fp32 arithmetic. */
int eod_match_label(const float* boxes /*[R,4]*/, int R, const float* gt_boxes /*[G,4]*/, const int32_t* gt_classes /*[G]*/, int G,
                    float iou_thresh, int num_classes, int32_t* matched_idx /*[R]*/, float* matched_iou /*[R]*/,
                    int32_t* out_classes /*[R]*/, float* out_gt_boxes /*[R,4]*/, eod_stream_t stream);
/* The same for a capacity-sized proposal list whose length lives on the device (eod_centernet_proposals' out_count), with
 * detectron2's add_ground_truth_to_proposals folded in (PROPOSAL_APPEND_GT): of the R = cap + (append_gt ? G : 0) rows, row i is
 * proposal i for i < min(*prop_count, cap), ground-truth box i - count for the next G rows, and no row beyond (out_classes -1:
 * eod_sample_proposals ignores it).  all_boxes [R,4] receives the rows' boxes (zeros where there is none). */
int eod_match_label_proposals(const float* prop_boxes /*[cap,4]*/, const int32_t* prop_count /*[1], device*/, int cap,
                              const float* gt_boxes /*[G,4]*/, const int32_t* gt_classes /*[G]*/, int G, int append_gt, float iou_thresh,
                              int num_classes, float* all_boxes, int32_t* matched_idx, float* matched_iou, int32_t* out_classes,
                              float* out_gt_boxes, eod_stream_t stream);
/* eod_sample_proposals: detectron2's subsample_labels (through label_and_sample_proposals, detic_roi_heads.py:232) as a selection by
 * random keys: the min(int(batch * positive_fraction), #foreground) foreground rows and the min(batch - that, #background) background
 * rows with the smallest (key, row); classes [R] as eod_match_label writes them (-1 = ignored).  sampled_idx [batch]: foreground rows
 * first, each kind in ascending row order; counts [2] = number of foreground rows, number of rows.  R <= 8192. */
int eod_sample_proposals(const int32_t* classes, const float* keys /*[R] uniform random*/, int R, int num_classes,
                         int batch_size_per_image, float positive_fraction, int32_t* sampled_idx, int32_t* counts, eod_stream_t stream);
/* eod_zs_logits: the scores DeticFastRCNNOutputLayers.forward returns in training (detic_fast_rcnn.py:437-466 with
 * zero_shot_classifier.py:71-111, NORM_WEIGHT, no bias): logits [B, ld] (columns 0..C1-1) = temp * normalize(feat [B,512]) . zs_weight
 * [512, C1]; featn_out [B,512] (optional) = the normalised, scaled feature.  Any C1. */
int eod_zs_logits(const float* feat, const float* zs_weight, int B, int D /*512*/, int C1, float temp, float* logits, int ld,
                  float* featn_out, eod_stream_t stream);

/* eod_zs_logits_backward: d feat [B,512] of eod_zs_logits given d_logits [B, ld] (F.normalize's and torch.mm's autograd; the class
 * matrix is a buffer, zero_shot_classifier.py:54). */
int eod_zs_logits_backward(const float* feat, const float* zs_weight, const float* d_logits, int ld, int B, int D /*512*/, int C1,
                           float temp, float* d_feat, eod_stream_t stream);

/* ---- CenterNet proposal decode (centernet.py:603-745) ------------------------------------------------- */
typedef struct EodProposalDesc {
  const float* head_out;   /* [P,8]: col 0 agn_hm logit, cols 1..4 bbox_pred (pre scale/relu) */
  int32_t head_stride;     /* 8 */
  int32_t levels;          /* 5 */
  int32_t level_off[6];    /* row offsets per level */
  int32_t level_w[5];
  int32_t level_stride[5];
  float level_scale[5];    /* Scale module values (centernet_head.py:159) */
  float score_thresh;      /* INFERENCE_TH */
  int32_t pre_nms_topk;    /* 1000 */
  int32_t post_nms_topk;   /* 256 */
  float nms_thresh;        /* 0.9 */
  int32_t cap;             /* capacity of out_* (>= post_nms_topk) */
  float* out_boxes;        /* [cap,4] */
  float* out_scores;       /* [cap] */
  int32_t* out_count;      /* [1] */
  void* workspace;         /* >= eod_proposals_workspace_bytes() */
  size_t workspace_bytes;
  /* 0 / 1, or B scenes in lock-step: head_out is level major over the scenes ([level][scene][h_l * w_l] rows: the layout the
   * batched pyramid-mode eod_conv2d writes), out_boxes [B*cap,4], out_scores [B*cap], out_count [B]; one workgroup per (level,
   * scene) and per scene */
  int32_t batch;
} EodProposalDesc;
size_t eod_proposals_workspace_bytes(int total_positions, int levels, int pre_nms_topk, int batch);
int eod_centernet_proposals(const EodProposalDesc* d, eod_stream_t stream);

/* ---- cascade box head glue ---------------------------------------------------------------------------- */
/* ZeroShotClassifier tail (zero_shot_classifier.py:86-106): x=50*feat/max(|feat|,1e-12); logits=x@zs [512,C1];
 * prob_acc (+)= sigmoid(logits) (predict_probs, detic_fast_rcnn.py:325-339).
 * Optional tails of the same launch (NULL / 0 = off):
 *   zs_mem [512,C1] + prop_scores [R] + mem_scores_out [R,C1]: the memory update's CLIP re-score of the proposals on the normalised
 *     feature this launch holds in registers: sqrt(sigmoid(featn . zs_mem) * prop_score), 0 for prop_score >= 1
 *     (inference_with_proposals, custom_rcnn.py:838-861) -- what eod_memory_scores computes from feat_norm_out;
 *   final_inv_stages > 0 (last cascade stage, with prop_scores): prob_acc = sqrt(prob_acc * final_inv_stages * prop_score)
 *     (detic_roi_heads.py:164-173) -- what eod_cascade_scores does in place. */
/* eod_zs_classify + the rest of the stage's predictor in the SAME launch: bbox_pred.2 (Linear hb_dim -> 4 on `hb`, the ReLU'd
 * bbox_pred.0 output; detic_fast_rcnn.py:109-116) and Box2BoxTransform.apply_deltas onto the stage's boxes with the stage's
 * weights, clipped to the image when `clip` (detic_roi_heads.py:121-122,314): three launches of the cascade's chain become one.
 * Every field up to `batch` means what the argument of the same name means in eod_zs_classify. */
typedef struct EodStageTailDesc {
  const float* feat;
  const float* zs;
  float* prob_acc;
  int32_t accumulate;
  float* feat_norm_out;
  const int32_t* count;
  int32_t R_cap, D, C1;
  float temp;
  const float* zs_mem;
  const float* prop_scores;
  float* mem_scores_out;
  float final_inv_stages;
  int32_t batch;
  const float* hb;         /* [R, hb_dim] */
  const float* w2;         /* [4][w2_ld]: bbox_pred.2 weight rows (w2_ld >= hb_dim, the packed conv layout) */
  const float* b2;         /* [4] */
  int32_t hb_dim, w2_ld;
  const float* boxes_in;   /* [R,4] the stage's boxes */
  float* boxes_out;        /* [R,4] refined boxes */
  float* deltas_out;       /* [R,4] or NULL */
  float wx, wy, ww, wh;    /* Box2BoxTransform weights of the stage */
  int32_t clip;
  float img_w, img_h;
} EodStageTailDesc;
int eod_cascade_stage_tail(const EodStageTailDesc* d, eod_stream_t stream);
int eod_zs_classify(const float* feat /*[R,512]*/, const float* zs /*[512,C1]*/, float* prob_acc /*[R,C1]*/, int accumulate,
                    float* feat_norm_out /*[R,512] or NULL*/, const int32_t* count, int R_cap, int D, int C1, float temp,
                    const float* zs_mem, const float* prop_scores, float* mem_scores_out, float final_inv_stages, int batch,
                    eod_stream_t stream);
/* Box2BoxTransform.apply_deltas + optional clip (detic_roi_heads.py:121-122,314) */
int eod_apply_deltas(const float* deltas /*[R,ld]*/, int ld, const float* boxes, float* out, const int32_t* count, int R_cap,
                     float wx, float wy, float ww, float wh, int clip, float img_w, float img_h, int batch, eod_stream_t stream);
/* scores = sqrt(mean_k(prob) * proposal_score) (detic_roi_heads.py:164-173) in place on prob_acc */
int eod_cascade_scores(float* prob_acc, const float* prop_scores, const int32_t* count, int R_cap, int C1, float inv_stages,
                       eod_stream_t stream);

/* d2 fast_rcnn_inference, single image (detic_roi_heads.py:214-221, custom_rcnn.py:862-869): threshold, sort, per-class NMS,
 * top-k and the gather of the kept rows in ONE launch (one workgroup: the sort is followed by a chunked greedy NMS that stops
 * once topk boxes are kept).  topk <= 512. */
typedef struct EodDetDesc {
  const float* boxes;     /* [R,4] class agnostic */
  const float* scores;    /* [R,C1] (last column = background) */
  const int32_t* count;   /* number of valid rows */
  int32_t R_cap, C1;
  float img_w, img_h, score_thresh, nms_thresh;
  int32_t topk;
  float* out_boxes;       /* [topk,4] clipped */
  float* out_scores;      /* [topk] */
  int32_t* out_classes;   /* [topk] */
  int32_t* out_rows;      /* [topk] proposal row of each detection */
  int32_t* out_count;     /* [1] */
  void* workspace; size_t workspace_bytes;
  /* optional (NULL = off): torch.unique of out_rows -- ascending, duplicates removed (custom_rcnn.py:875) -- written by the same
   * launch; needs R_cap <= 512 */
  int32_t* out_unique_rows;   /* [unique_cap] */
  int32_t* out_unique_count;  /* [1] */
  int32_t unique_cap;
  /* optional (NULL = off): detections that come from one proposal row carry the same class-agnostic box (they differ in class and
   * score only), so everything computed from the box alone -- the mask head (CLS_AGNOSTIC_MASK) -- is the same for all of them.
   * out_rep_of[k] = first detection with the row of detection k; out_rep_list = the detections that represent their group,
   * ascending; out_rep_count = their number.  Needs R_cap <= 512. */
  int32_t* out_rep_of;     /* [topk] */
  int32_t* out_rep_list;   /* [topk] */
  int32_t* out_rep_count;  /* [1] */
  int32_t batch;           /* 0 / 1, or B scenes (batch convention: every buffer above B x, one workgroup per scene; indices stay scene-local) */
} EodDetDesc;
size_t eod_detections_workspace_bytes(int R_cap, int C1);
int eod_fast_rcnn_inference(const EodDetDesc* d, eod_stream_t stream);

/* The scene-local index lists of a batch (EodDetDesc.out_unique_rows / out_rep_list: lists [batch, cap_in], counts [batch]) as ONE
 * list of global indices b * id_stride + lists[b][k], scene by scene, with its total length: the mask head then runs once over all
 * scenes' ROIs (eod_roi_align box_rows, EodConvDesc.m_count / out_units). */
int eod_concat_lists(const int32_t* lists, const int32_t* counts, int cap_in, int id_stride, int batch, int32_t* out /*[batch*cap_in]*/,
                     int32_t* out_count /*[1]*/, eod_stream_t stream);

/* d2 detector_postprocess without the mask paste (custom_rcnn.py:579): scale, clip, drop empty boxes.
 * out_src[q] = index of the kept detection in the input list, or remap[that index] when remap is given (EodDetDesc.out_rep_of:
 * the detection whose mask stands for it). cap <= 512.  batch: see the batch convention (one workgroup per scene). */
int eod_detector_postprocess(const float* boxes, const float* scores, const int32_t* classes, const int32_t* count, int cap,
                             float sx, float sy, float out_w, float out_h, float* out_boxes, float* out_scores,
                             int32_t* out_classes, int32_t* out_src, int32_t* out_count, const int32_t* remap, int batch,
                             eod_stream_t stream);
/* paste_masks_in_image (d2, inside detector_postprocess; custom_rcnn.py:579,880): out[k] = grid_sample(prob[rows[k]],
 * box k) >= threshold, u8 [K,H,W].  rows NULL = identity.  batch > 1: rows are scene-local indices into the scene's prob_units masks. */
int eod_paste_masks(const float* prob, const float* boxes, const int32_t* rows, const int32_t* count, int K_cap,
                    int H, int W, float threshold, uint8_t* out, int batch, int prob_units /* batch > 1: masks per scene in prob */,
                    eod_stream_t stream);

/* ---- spatial feature memory --------------------------------------------------------------------------- */
/* a1+a2: ProjectorUtils.pixel_to_world_mapping (SMNet/projector/core.py:177-225) + grid-cell index
 * (SMNet/build_memory_data.py:135-144, robot_demo.py:526-534).  INT result, bit-exact vs oracle/projector.c */
int eod_unproject_grid_index(const float* depth, int H, int W, const float* T16_host, float fx, float fy, float cx, float cy,
                             const float* proj_shift3_host, const float* map_shift3_host, float cell, int map_w, int map_h,
                             int order, float* xyz_or_null, int32_t* idx, eod_stream_t stream);
/* a4 + fp16 cast (custom_rcnn.py:762-774,1036): out_f16[n] = half(obs>1 ? mem/obs : mem) */
int eod_memory_normalize_f16(const float* mem, const float* obs, uint16_t* out_f16, int n_cells, int D, eod_stream_t stream);
/* a4, incremental form: re-normalise only the rows flagged in `dirty` (int32 [n_cells], set by eod_memory_write for every cell
 * whose observation count or accumulator row changed) into the resident fp16 table, and clear the flags.  Equal to
 * eod_memory_normalize_f16 on those rows; the other rows of out_f16 are left as they are (they did not change). */
int eod_memory_normalize_dirty_f16(const float* mem, const float* obs, int32_t* dirty, uint16_t* out_f16, int n_cells, int D,
                                   eod_stream_t stream);
/* device-side error word (OR of flags); kernels that index with caller data clamp and set it instead of faulting */
#define EOD_FLAG_BAD_CELL_INDEX 1   /* a proj_indices entry outside [0, n_cells) was clamped */
/* a8 gather + cascaded average pooling (timm.py:147-168): the values timm.py:168 casts to fp16, for the stride-8, -16 and -32
 * levels, stored in the operand-fragment order eod_memory_project_fuse reads: [level][32-row tile][k-step s<32][hi<2][r<32][8]
 * halves = element (row 32*tile + r, channel 16 s + 8 hi + j); each level starts on a tile boundary, so the buffer holds
 * eod_memory_pooled_halves(H, W) halves.  err_flags: int32 [1] or NULL.  torch_order != 0: every 4x4 block is summed pixel by pixel
 * in row-major order (bit-identical to F.avg_pool2d on every input); 0: a block with <= 4 distinct cells is summed as
 * sum count * row in first-appearance order -- the same 16 numbers with fewer roundings, identical to the sequential sum whenever
 * that sum is exact (exponents inside the block span <= 9 bits per channel); fewer instructions on an issue-bound kernel. */
size_t eod_memory_pooled_halves(int H, int W);
int eod_memory_gather_pool(const uint16_t* mem_f16, const int32_t* proj, int H, int W, int D, int n_cells,
                           uint16_t* pooled_f16, int32_t* err_flags, int torch_order, int batch, eod_stream_t stream);
/* a8 projection + fusion (timm.py:174-189): for the three levels P_l = (pooled_l . W_l^T + b_l) * weight (+ P_l if mode 0
 * "sum"; mode 1 "mem_only" overwrites), in place on the [h8*w8 + h16*w16 + h32*w32, 256] fp32 row list `feats`.
 * `prepared` (eod_memory_project_weights_bytes() bytes, 16-byte aligned) is built once from the three Conv2d(512,256,1)
 * layers `map_merge_projection{1,2,3}` (weights [256,512] fp32, bias [256]).  batch > 1: B tables / index images / pooled
 * buffers back to back (err_flags shared); `feats` is LEVEL MAJOR over the scenes -- level l of scene b starts at row
 * B * level_off[l] + b * rows_l -- the [B,h,w,256] level images the batched FPN convs write and read. */
size_t eod_memory_project_weights_bytes(void);
int eod_memory_project_prepare(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                               const float* b3, void* prepared, eod_stream_t stream);
int eod_memory_project_fuse(const uint16_t* pooled_f16, const void* prepared, float* feats, int H, int W, float weight,
                            int mode, int batch, eod_stream_t stream);
/* a16 scoring of proposals in CLIP space (custom_rcnn.py:848-855): scores = sqrt(sigmoid(featn@zs)*ps) */
int eod_memory_scores(const float* featn /*[R,512]*/, const float* zs, const float* prop_scores, float* scores /*[R,C1]*/,
                      const int32_t* count, int R_cap, int D, int C1, eod_stream_t stream);
/* torch.unique of the kept proposal rows (custom_rcnn.py:875): ascending, duplicates removed. R_cap <= 512. */
int eod_unique_rows(const int32_t* rows, const int32_t* count, int K_cap, int R_cap, int32_t* out_rows, int32_t* out_count,
                    eod_stream_t stream);
/* ---- training forward, first slice: backward of the memory read (timm.py:142-192) ------------------------------
 * g3 / g4 / g5: dL/d(fused P3 / P4 / P5), [P_l, 256] fp32 rows; pooled_f16: the buffer eod_memory_gather_pool wrote in the forward.
 * dW_l [256,512] = weight * G_l^T . E_l, db_l [256] = weight * column sums of G_l (fp32 MFMA, deterministic summation order).
 * The gradient with respect to the pooled operand, weight * G_l . W_l, is a 1x1 convolution with the transposed weights: eod_conv2d. */
int eod_memory_project_backward_weights(const float* g3, const float* g4, const float* g5, const uint16_t* pooled_f16, int H, int W,
                                        float weight, float* dw3, float* db3, float* dw4, float* db4, float* dw5, float* db5,
                                        eod_stream_t stream);
/* The same with a workspace (>= eod_memory_project_backward_weights_workspace_bytes()): every level's positions are cut into 8
 * ranges, one grid slice each, whose partial results are added in range order by a second launch per level (without it the 128
 * workgroups of the finest level each walk all of its positions: 290 us at 640x640 against ~50). */
size_t eod_memory_project_backward_weights_workspace_bytes(void);
int eod_memory_project_backward_weights_ws(const float* g3, const float* g4, const float* g5, const uint16_t* pooled_f16, int H, int W,
                                           float weight, float* dw3, float* db3, float* dw4, float* db4, float* dw5, float* db5,
                                           void* workspace, size_t workspace_bytes, eod_stream_t stream);
/* Backward of the cascaded pools avg_pool2 -> half (timm.py:163-168) with autograd's rounding of gradients that enter half tensors:
 * dec3..5 = weight * G_l . W_l ([P_l,512] fp32 rows) -> ge3..5 (gradients of the half tensors E_3..E_5, [P_l,512] half rows) and
 * ge2 (gradient of the fp32 avg_pool4 output, [(H/4)*(W/4), 512]). */
int eod_memory_pool_backward(const float* dec3, const float* dec4, const float* dec5, int H, int W, uint16_t* ge3_f16, uint16_t* ge4_f16,
                             uint16_t* ge5_f16, float* ge2, eod_stream_t stream);

/* Third slice: backward of a convolution layer (NHWC fp32, Cin and Cout multiples of 32, or Cin == 4: the stem's tap layout,
 * timm.py:279; g on the layer's output grid
 * [N,OH,OW,Cout], OH = (H + 2 pad - KH) / stride + 1): the layers downstream of the memory fusion -- CenterNet tower (centernet_head.py:141-161), FPN output convs (timm.py:118-136),
 * mask head convs (detic_roi_heads.py:257-268).  x [N,H,W,Cin] the layer's input, g [N,H,W,Cout] dL/d(pre-activation output).
 * dw [Cout][KH*KW*Cin] in the layout of the packed forward weights (k = (ky, kx, ci), ci fastest), db [Cout] or NULL (fp32 MFMA,
 * deterministic summation order).  The gradient with respect to x is eod_conv2d of g with the rotated / transposed weights. */
int eod_conv2d_backward_weights(const float* x, const float* g, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad,
                                int stride, float* dw, float* db, eod_stream_t stream);
/* The same with a workspace (>= eod_conv2d_backward_weights_workspace_bytes(...) for the same arguments; 0 bytes: not needed): layers
 * with few channel tiles and many positions (the trunk's first stages) cut the positions into up to 64 ranges, one grid slice
 * each, whose partial results are added in range order by a second launch. */
size_t eod_conv2d_backward_weights_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad, int stride);
int eod_conv2d_backward_weights_ws(const float* x, const float* g, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad,
                                   int stride, float* dw, float* db, void* workspace, size_t workspace_bytes, eod_stream_t stream);
/* Pyramid mode (the weight gradient of a level-shared layer, centernet_head.py:141-161, in one launch): x [rows, Cin] / g [rows, Cout]
 * are row lists, rows [level_off[l], level_off[l+1]) an level_h[l] x level_w[l] image (levels <= 8); stride 1, pad = (KH - 1) / 2;
 * dW / db are summed over the levels.  The workspace (optional, as above) holds the position ranges' partial results. */
size_t eod_conv2d_backward_weights_levels_workspace_bytes(int rows, int Cin, int Cout, int KH, int KW);
int eod_conv2d_backward_weights_levels(const float* x, const float* g, int levels, const int32_t* level_off, const int32_t* level_h,
                                       const int32_t* level_w, int Cin, int Cout, int KH, int KW, int pad, float* dw, float* db,
                                       void* workspace, size_t workspace_bytes, eod_stream_t stream);
/* Input gradient of a strided convolution (stride >= 2: P6 / P7, timm.py:359-364; g is [N,OH,OW,Cout]), gather form, w = the
 * forward's packed weights [Cout][Kpad]; dx [N,H,W,Cin].  Stride-1 'same' layers use eod_conv2d with the rotated weights instead. */
int eod_conv2d_backward_input(const float* g, const float* w, int Kpad, int N, int H, int W, int Cin, int Cout, int KH, int KW, int pad,
                              int stride, float* dx, eod_stream_t stream);
/* Backward of the FPN top-down add (res_mode 2 of eod_conv2d: + nearest-x2 of the coarser level, timm.py:128-133): out [N,h,w,C]
 * (+)= the sum of each 2x2 block of g [N,2h,2w,C].  C % 4 == 0. */
int eod_upsample2_sum_backward(const float* g, float* out, int N, int h, int w, int C, int accumulate, eod_stream_t stream);
/* Backward of eod_maxpool3x3s2 (timm.py:281): the gradient of an output window goes to the first maximum of the window in row-major
 * order (torch's max_pool2d backward); gather form, no atomics. */
int eod_maxpool3x3s2_backward(const float* x, const float* y, const float* g, float* dx, int N, int H, int W, int C, int OH, int OW,
                              eod_stream_t stream);
/* out = g where y > 0, else 0 (backward of the ReLU epilogue); n % 4 == 0, 16-byte aligned */
int eod_relu_backward(const float* g, const float* y, float* out, size_t n, eod_stream_t stream);

/* Optimizer step of the reference's training configuration for ONE parameter tensor (second slice): detectron2's
 * per-parameter gradient clipping by value (SOLVER.CLIP_GRADIENTS.ENABLED, CLIP_TYPE "value"; clip_value <= 0: off) followed by
 * torch.optim.AdamW's single-tensor update (SOLVER.OPTIMIZER ADAMW, Base-C2_L_R5021k_640b64_4x_recurrent.yaml:68-74; built by
 * Detic/detic/custom_solver.py:19-79 with lr = BASE_LR x CUSTOM_MULTIPLIER for the `map_merge` parameters).  step >= 1 is the
 * 1-based update count of this tensor; exp_avg / exp_avg_sq are its state, zero before the first step.  In place, elementwise. */
int eod_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int step, double clip_value, eod_stream_t stream);

/* Weights of a layer's input-gradient convolution from its packed forward weights w [Cout][ld_in] (k = (ky, kx, ci)):
 * out [Cin][ld_out] with k' = (ky', kx', co) and out[ci][k'] = w[co][(KH-1-ky', KW-1-kx', ci)] -- the 180-degree rotated, in/out-transposed
 * kernel that `eod_conv2d` convolves dY with (autograd of every conv of timm.py:118-136,277-299, centernet_head.py:141-161). */
int eod_conv_rotate_weights(const float* w, int Cout, int KH, int KW, int Cin, int ld_in, float* out, int ld_out, eod_stream_t stream);

/* eod_conv_rotate_weights for `count` layers in ceil(count / 24) launches (after an optimizer step every layer with an input-gradient
 * convolution needs it).  `tensors` is a HOST array, read before the call returns. */
typedef struct EodRotateTensor {
  const float* w;
  float* out;
  int32_t Cout, KH, KW, Cin, ld_in, ld_out;
} EodRotateTensor;
int eod_conv_rotate_weights_multi(const EodRotateTensor* tensors, int count, eod_stream_t stream);

/* The same update for `count` parameter tensors in ceil(count / 20) launches (the training step of the recurrent detector has 126):
 * tensors[i] carries its own lr (BASE_LR x multipliers x the schedule's factor), weight decay and 1-based update count.  Element
 * for element the arithmetic of eod_adamw_step.  `tensors` is a HOST array, read before the call returns. */
typedef struct EodAdamWTensor {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  size_t n;
  double lr;
  double weight_decay;
  int32_t step;
  /* optional (NULL = off): the stepped tensor read as [n / cols, cols] rows, times row_scale[row], written to folded_out with row
   * pitch ld_out -- a trunk conv's raw master weight -> the layer's packed weights with its FrozenBatchNorm folded in */
  float* folded_out;
  const float* row_scale;
  int32_t cols, ld_out;
  /* 1: `grad` is the gradient of the FOLDED weights; the master's = grad x row_scale[row] (chain rule of the fold), applied before the
   * clipping.  Needs row_scale and cols. */
  int32_t grad_of_folded;
} EodAdamWTensor;
int eod_adamw_step_multi(const EodAdamWTensor* tensors, int count, double beta1, double beta2, double eps, double clip_value,
                         eod_stream_t stream);

/* a16-a19 write path (custom_rcnn.py:681-760,875-936) */
typedef struct EodMemWriteDesc {
  const float* featn;       /* [R,512] normalised x50 proposal features */
  const float* prop_boxes;  /* [R,4] unclipped proposal boxes */
  const float* prop_masks;  /* [R,28,28] mask probabilities */
  const int32_t* det_rows;  /* [K_cap] proposal rows kept by fast_rcnn_inference (with duplicates) */
  const int32_t* det_count; /* [1] */
  int32_t K_cap, R_cap;
  const int32_t* proj;      /* [H,W] */
  int32_t H, W, D, n_cells;
  float mask_thresh;
  float* mem;               /* [N,512] accumulator (semmap_features / implicit_memory) */
  float* obs;               /* [N] observation counts */
  int32_t* k_out;           /* [1] number of unique instances written (diagnostic) */
  void* workspace; size_t workspace_bytes;
  int32_t* dirty;           /* [N] or NULL: set to 1 for every cell whose observation count changed (see normalize_dirty) */
  int32_t* err_flags;       /* [1] or NULL: EOD_FLAG_* */
  uint16_t* snapshot_f16;   /* [N,512] fp16 or NULL: the normalised table eod_memory_gather_pool reads.  When given, the rows of
                             * every cell whose observation count changed are refreshed by the write itself (the snapshot stays
                             * current: no eod_memory_normalize_dirty_f16 launch before the next read) and `dirty` is not touched */
  int32_t batch;            /* 0 / 1, or B scenes with B independent states (batch convention: every buffer above B x, err_flags
                             * shared; workspace >= B x eod_memory_write_workspace_bytes()); three launches for all scenes */
} EodMemWriteDesc;
size_t eod_memory_write_workspace_bytes(int H, int W, int D, int n_cells, int K_cap, int R_cap);
int eod_memory_write(const EodMemWriteDesc* d, eod_stream_t stream);
/* zero the per-frame cell flags inside the workspace once after allocation (eod_memory_write leaves them zero) */
int eod_memory_write_init(void* workspace, size_t workspace_bytes, int H, int W, int D, int n_cells, int K_cap, int R_cap,
                          eod_stream_t stream);

/* a20: explicit semantic map from the implicit memory (custom_rcnn.py:745-756,938-1017), evaluated lazily (only
 * consumed when TEST_SAVE_SEMMAP): labels[c] = argmax over the first C1-1 classes of (50*mem/|mem|)@zs, or -1 where the
 * min-max normalised observation intensity mean|mem| (/obs if obs>1) is below thresh.  workspace >= n_cells + 4 floats. */
int eod_semmap_labels(const float* mem, const float* obs, const float* zs, int n_cells, int D, int C1, float thresh,
                      int32_t* labels, float* workspace, eod_stream_t stream);

/* utility */
int eod_fill_f32(float* p, float v, size_t n, eod_stream_t stream);
int eod_fill_i32(int32_t* p, int32_t v, size_t n, eod_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EOD_HIP_H */
