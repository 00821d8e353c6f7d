/* mi355det — C ABI of the MI355X-native detection hot path (libmi355det.so).
 *
 * Drop-in boundary for the one hot path of kostas1515/object_detectors.  The
 * reference has no FFI of its own: its boundary is a set of Python callables.  Each entry point
 * below names the reference callable (file:line under /root/reference) it replaces; the Python
 * mirror in object_detectors_amd/ binds them with ctypes.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless it is
 *     marked "host".  The caller owns all buffers (inputs, outputs, workspace).
 *   - every call enqueues on `stream` (hipStream_t passed as void*) and returns immediately;
 *     no call synchronises, allocates or copies to the host.
 *   - return value: 0 = MI355DET_OK, negative = error (mi355det_last_error() gives the text);
 *     the library never throws and never aborts.
 *   - data-dependent output sizes (NMS keeps, compaction) are written to a caller-provided
 *     max-size buffer plus a device-side counter.
 *   - fp32 for boxes / losses / scores; bf16 (NHWC) for convolution activations and weights.
 */
#ifndef MI355DET_H
#define MI355DET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355DET_OK 0
#define MI355DET_EINVAL (-1)   /* bad argument / unsupported shape */
#define MI355DET_ELAUNCH (-2)  /* HIP launch failure */
#define MI355DET_EWORKSPACE (-3) /* workspace too small */

#define MI355DET_MAX_SCALES 4
#define MI355DET_MAX_ANCHORS 8

const char* mi355det_last_error(void);
int mi355det_debug_set(int key, int value);   /* bring-up / test knobs: key 0 = force a conv tile configuration (0 = tuned), 1 = weight gradient
                                                 with per-lane bookkeeping, 2 = stride-2 data gradient as four class launches (tests compare the forms),
                                                 3 = diagnostic build of the phase-staggered conv, 5 = stride-2 data-gradient form, 6 = weight-gradient
                                                 ablations (timing only), 7 = weight-gradient split count for every launch (+ 65536: the 256 x 256
                                                 phase-staggered kernel; fails for shapes it does not take), 8 = 1: the tuner leaves that kernel out */
int mi355det_debug_ptr(int key, void* ptr);   /* key 0 = device buffer for the diagnostic (phase-stamp) conv build */
int mi355det_version(void);

/* ------------------------------------------------------------------------------------------
 * YOLO criterion geometry (yolo/nets/yolo_forw.py:93-119).  Anchor index within an image is
 * off[k] + (y*W+x)*na + a, scales in the order the head returns them (stride 32, 16, 8).
 * anchor_w/h are the NORMALISED anchor sizes exactly as the reference computes them in float32:
 * float32(a_w / (img_size/grid)) / float32(grid).
 */
typedef struct {
  int32_t num_scales, na, num_classes, pad0;
  float img_size, ignore_thr;
  int32_t iou_type, pad1;                 /* 0 IoU, 1 GIoU, 2 DIoU, 3 CIoU (helper.py:224-231) */
  int32_t grid[MI355DET_MAX_SCALES];      /* H == W per scale */
  int32_t off[MI355DET_MAX_SCALES + 1];   /* prefix sums of grid^2*na; off[num_scales] == N */
  float anchor_w[MI355DET_MAX_SCALES][MI355DET_MAX_ANCHORS];
  float anchor_h[MI355DET_MAX_SCALES][MI355DET_MAX_ANCHORS];
} mi355det_yolo_geom;

/* A head tensor in place: element (b, a, attr, pix) at base + b*sb + (a*attrs+attr)*sc + pix*sp
 * (elements).  NCHW [bs,A*attrs,H,W]: sc=H*W, sp=1.  NHWC (engine native): sc=1, sp=row pitch. */
typedef struct {
  void* ptr;
  int64_t sb, sc, sp;
} mi355det_head_view;

typedef struct {
  float lambda_iou, lambda_xy, lambda_wh, lambda_conf, lambda_no_conf, lambda_cls;
  float alpha, gamma;          /* custom.FocalLoss (yolo/utilities/custom.py:40-67) */
  float grad_scale;            /* multiplies every gradient (1/sum(M) is applied internally) */
  int32_t grad_is_bf16;        /* format of the grad views: 0 = fp32, 1 = bf16 (engine default), 2 = IEEE fp16 (engine with fp16 storage) */
  const float* class_weights;  /* device [C] or NULL: nn.CrossEntropyLoss(weight=..., reduction='sum') class weights
                                  (yolo_forw.py:50-62,72: tf-idf / effective-number re-weighting); for class_loss 0 / 2 the same
                                  row is BCEWithLogitsLoss's pos_weight (yolo_forw.py:70-71,74-75) */
  int32_t class_loss;          /* cfg.class_loss (yolo_forw.py:69-77): 0 BCEWithLogitsLoss, 1 CrossEntropyLoss (the default), 2 custom.EQLoss */
  int32_t reduction_mean;      /* cfg.reduction: 0 'sum' (every term / sum(M), yolo_forw.py:158-160), 1 'mean' (each loss over its own
                                  element count, no final division) */
  const float* eq_mask;        /* device [C], class_loss 2 only: custom.EQLoss.eq_mask (custom.py:79-80, 1.0 where the class's image-frequency
                                  share is below 0.0045) */
} mi355det_yolo_loss_cfg;

/* helper.bbox_iou (yolo/utilities/helper.py:221-277), broadcast form [M,1,4] x [1,N,4] -> [M,N]
 * (elementwise form: M==1 rows paired, see `paired`).  xcycwh!=0 converts with get_abs_coord. */
int mi355det_bbox_iou(const float* bb1, const float* bb2, float* out, int64_t m, int64_t n,
                      int iou_type, int xcycwh, int paired, void* stream);

/* YOLOForw.get_target (yolo_forw.py:178-208), all images in one launch.
 *   gt_box [G,4] relative xcycwh, gt_off [bs+1] int32 prefix of per-image GT counts (device).
 * out: best_key [G] u64 scratch (zeroed by the call), obj_idx [G] int64, tgt [G,4],
 *      noobj [bs,N] uint8 (1 = contributes to the no-object loss). */
int mi355det_yolo_assign(const mi355det_yolo_geom* geom, const float* gt_box, const int32_t* gt_off,
                         int32_t bs, int32_t num_gt, int32_t max_gt_per_img, uint64_t* best_key,
                         int64_t* obj_idx, float* tgt, uint8_t* noobj, void* stream);

/* YOLOForw.forward, train branch (yolo_forw.py:121-162) fused forward+backward.
 *   heads/grads: num_scales views (grads may be NULL for forward only; grad buffers must be
 *   zero-filled by the caller — only conf planes and positive rows are written).
 *   gt_label [G] int64, idf [C] or NULL (idf_logits, yolo_forw.py:63-67,136).
 *   partials: float workspace of mi355det_yolo_loss_workspace(bs, N) bytes.
 *   out12: loss, sub_losses[6] (xy, wh, iou, pos_conf, neg_conf, cls; already / sum(M)), stats[5]. */
size_t mi355det_yolo_loss_workspace(int32_t bs, int64_t n_anchors);
int mi355det_yolo_loss(const mi355det_yolo_geom* geom, const mi355det_yolo_loss_cfg* cfg,
                       const mi355det_head_view* heads, const mi355det_head_view* grads,
                       const int32_t* gt_off, const int64_t* gt_label, const int64_t* obj_idx,
                       const float* tgt, const uint8_t* noobj, const float* idf, int32_t bs,
                       int32_t num_gt, void* workspace, size_t workspace_bytes, float* out12,
                       void* stream);

/* YOLOForw.forward, inference branch (yolo_forw.py:163-176): out [bs,N,attrs] fp32 contiguous.
 * softmax_cls!=0: class_loss is CrossEntropy (softmax), else sigmoid.
 * score_out/label_out [bs,N] (optional, channels-last heads only): conf*max(cls) and arg-max class
 * (test_one_epoch.py:25,35) produced in the same pass. */
int mi355det_yolo_decode(const mi355det_yolo_geom* geom, const mi355det_head_view* heads,
                         const float* idf, int32_t bs, int softmax_cls, float* out, float* score_out,
                         int32_t* label_out, void* stream);

/* test_one_epoch.py:24-35: get_abs_coord + score=conf*max(cls) + threshold + row build.
 * pred [bs,N,attrs] (decoded). cand [bs,max_cand,6] = (x1,y1,x2,y2,score,label), count [bs] int32
 * (true number of passing boxes, may exceed max_cand: caller must check).  Candidates keep the
 * anchor order of the reference's boolean mask. */
size_t mi355det_yolo_candidates_workspace(int32_t bs, int64_t n);
int mi355det_yolo_candidates(const float* pred, const float* score_in /* optional: from yolo_decode */,
                             const int32_t* label_in, int32_t bs, int64_t n, int32_t attrs, float conf_thr,
                             float* cand, int32_t* count, int32_t max_cand, void* workspace,
                             size_t workspace_bytes, void* stream);

/* helper.nms_majority (helper.py:280-382), batched over images.
 *   boxes [bs,max_n,6], count [bs] int32 (device).  Equal scores are ordered "stable ascending
 *   argsort" (highest original index first among ties).
 * out: out_rows [bs,max_n,6] kept rows in keep order with the majority relabel applied,
 *      out_idx [bs,max_n] int32 original row index, out_count [bs] int32.
 * max_n <= 131072 (up to 16384 boxes are sorted by one workgroup in LDS, more by chunk sorts + a merge by rank; the n*n/8-byte
 * suppression mask is 2 GiB per image at the cap).  workspace: mi355det_nms_workspace(bs, max_n). */
size_t mi355det_nms_workspace(int32_t bs, int32_t max_n);
int mi355det_nms_majority(const float* boxes, const int32_t* count, int32_t bs, int32_t max_n,
                          float thresh_iou, int32_t num_classes /* labels in [0,num_classes) */, float* out_rows, int32_t* out_idx, int32_t* out_count,
                          void* workspace, size_t workspace_bytes, void* stream);

/* torchvision.ops.boxes.box_iou (call sites: tvision/retinanet.py:409, rpn.py:192, roi_heads.py:633) */
int mi355det_box_iou(const float* boxes1, const float* boxes2, float* out, int64_t m, int64_t n,
                     void* stream);

/* torchvision.ops.boxes.nms / batched_nms (retinanet.py:463, rpn.py:272, roi_heads.py:771).
 *   boxes [n,4] xyxy, scores [n], idxs [n] int64 or NULL (plain nms); n <= 131072.
 *   keep [n] int64 in descending-score order (ties: lower index first), keep_count [1] int32. */
int mi355det_nms(const float* boxes, const float* scores, const int64_t* idxs, int32_t n,
                 float iou_thr, int64_t* keep, int32_t* keep_count, void* workspace,
                 size_t workspace_bytes, void* stream);

/* The same NMS for `bs` independent box sets of `n` boxes each in ONE launch sequence (replaces the per-image loop around
 * box_ops.batched_nms in RegionProposalNetwork.filter_proposals, tvision/rpn.py:259-280): boxes [bs,n,4], scores [bs,n], idxs [bs,n] or
 * NULL, keep [bs,n] (the first keep_count[b] entries of row b are defined), keep_count [bs].  Workspace: mi355det_nms_workspace(bs, n). */
int mi355det_nms_batch(const float* boxes, const float* scores, const int64_t* idxs, int32_t bs, int32_t n, float iou_thr, int64_t* keep,
                       int32_t* keep_count, void* workspace, size_t workspace_bytes, void* stream);

/* RegionProposalNetwork.filter_proposals for the whole batch in one call (tvision/rpn.py:215-280, with the anchor decode of :336-351
 * restricted to the selected anchors): per-level top-k of the objectness logits, decode (BoxCoder weights 1, clamp xform_clip), clip to
 * clip_limits[img] = (w, h, w, h), boxes smaller than min_size or scoring below score_thresh masked out, per-level NMS, the first
 * post_nms_top_n survivors of every image.  objectness [N, A] fp32 logits and deltas [N, A, 4], A = sum(level_counts) with the levels
 * concatenated in pyramid order; anchors [A, 4] xyxy; level_counts [nlev] on the HOST (nlev <= 8).  Outputs: out_boxes [N, post, 4],
 * out_scores [N, post] (sigmoid), out_counts [N] int32 on the device - rows beyond out_counts[img] are zero.  Results are those of
 * mi355det_topk + mi355det_box_decode + the clip / mask chain + mi355det_nms_batch, bit for bit.
 * Workspace: mi355det_rpn_proposals_workspace (0 for invalid arguments). */
size_t mi355det_rpn_proposals_workspace(int32_t n_images, const int64_t* level_counts, int32_t nlev, int32_t pre_nms_top_n);
int mi355det_rpn_proposals(const float* objectness, const float* deltas, const float* anchors, const float* clip_limits, int32_t n_images,
                           const int64_t* level_counts, int32_t nlev, int32_t pre_nms_top_n, int32_t post_nms_top_n, float nms_thresh,
                           float score_thresh, float min_size, float xform_clip, float* out_boxes, float* out_scores, int32_t* out_counts,
                           void* workspace, size_t workspace_bytes, void* stream);

/* RetinaNet.postprocess_detections for the whole batch in one call (tvision/retinanet.py:414-472): per level the scores above the threshold
 * (logit_thresh = logit of score_thresh: sigmoid is monotone), of those the topk_candidates best of the flattened [HWA_l x K] scores of
 * every image, decode of their anchors (BoxCoder weights 1), clip to clip_limits[img] = (w, h, w, h), per-class NMS, the first
 * detections_per_img survivors.  HOST tables per level: cls_logits[l] [N, HWA_l, K] fp32 (already scaled by the tf-idf row if any),
 * bbox_regression[l] [N, HWA_l, 4], anchors[l] [HWA_l, 4], level_anchors[l] = HWA_l.  Outputs: out_boxes [N, det, 4], out_scores [N, det],
 * out_labels [N, det] int64, out_counts [N] int32 on the device; rows beyond out_counts[img] are zero.  The same results as
 * mi355det_topk_ws + mi355det_box_decode + clip + mi355det_nms_batch.  Workspace: mi355det_retina_detections_workspace (0 = bad arguments). */
size_t mi355det_retina_detections_workspace(int32_t n_images, const int64_t* level_anchors, int32_t nlev, int32_t num_classes, int32_t topk_candidates);
int mi355det_retina_detections(const float* const* cls_logits, const float* const* bbox_regression, const float* const* anchors,
                               const int64_t* level_anchors, int32_t nlev, int32_t n_images, int32_t num_classes, const float* clip_limits,
                               float logit_thresh, int32_t topk_candidates, float nms_thresh, int32_t detections_per_img, float xform_clip,
                               float* out_boxes, float* out_scores, int64_t* out_labels, int32_t* out_counts, void* workspace,
                               size_t workspace_bytes, void* stream);

/* RoIHeads.postprocess_detections for the whole batch in one call (tvision/roi_heads.py:715-781): scores [N, P, C] (the reference's score
 * function already applied; the caller pushes class 0 and padded proposals below score_thresh), box_regression [N, P, C, 4], proposals
 * [N, P, 4], clip_limits [N, 4] = (w, h, w, h).  Candidates = the scores above score_thresh, at most max_candidates per image in descending
 * order (candidate_counts[img] == max_candidates means the list may be truncated: the caller then takes the unbounded route); decode with
 * BoxCoder(wx, wy, ww, wh), clip, boxes smaller than min_size masked, per-class NMS, the first detections_per_img survivors ->
 * out_boxes [N, det, 4], out_scores [N, det], out_labels [N, det] int64, out_counts [N], candidate_counts [N] (int32, device). */
size_t mi355det_roi_detections_workspace(int32_t n_images, int32_t max_proposals, int32_t num_classes, int32_t max_candidates);
int mi355det_roi_detections(const float* scores, const float* box_regression, const float* proposals, const float* clip_limits, int32_t n_images,
                            int32_t max_proposals, int32_t num_classes, float score_thresh, int32_t max_candidates, float wx, float wy, float ww,
                            float wh, float xform_clip, float min_size, float nms_thresh, int32_t detections_per_img, float* out_boxes,
                            float* out_scores, int64_t* out_labels, int32_t* out_counts, int32_t* candidate_counts, void* workspace,
                            size_t workspace_bytes, void* stream);

/* RegionProposalNetwork.compute_loss (tvision/rpn.py:282-318) on prepared indices, forward and gradient in one launch: objectness [T] logits,
 * pred_bbox_deltas / regression_targets [T,4], labels [T] (1 / 0 for the sampled anchors), pos_idx [num_pos] and sampled_idx [num_sampled]
 * (positives followed by negatives, unique).  losses[0] = binary_cross_entropy_with_logits(objectness[sampled], labels[sampled]) (mean),
 * losses[1] = smooth_l1(deltas[pos], targets[pos], beta 1/9, sum) / num_sampled; grad_objectness [T] and grad_deltas [T,4] are their
 * gradients (zero outside the sampled / positive anchors), written completely. */
int mi355det_rpn_loss(const float* objectness, const float* pred_bbox_deltas, const float* labels, const float* regression_targets, int64_t total,
                      const int64_t* pos_idx, int32_t num_pos, const int64_t* sampled_idx, int32_t num_sampled, float* losses, float* grad_objectness,
                      float* grad_deltas, void* stream);

/* RoIHeads.select_training_samples for the whole batch (tvision/roi_heads.py:627-713), the two launches around the one host read its
 * sampler needs.  Candidates of image i are its proposals (proposals [N, max_proposals, 4] padded, proposal_counts [N] on the device, as
 * mi355det_rpn_proposals leaves them) followed by its ground truth (add_gt_proposals): gt_boxes [G,4] / gt_labels [G] of all images
 * concatenated, gt_offsets [N+1] on the HOST (1..1024 boxes per image; at most 64 images, 8192 candidates per image).
 *   mi355det_roi_match: box_iou + Matcher(fg_iou_thresh, bg_iou_thresh, no low-quality rescue) + assign_targets_to_proposals ->
 *     matched [N, row_stride] (clamped at 0), labels [N, row_stride] (-1 between the thresholds, 0 background, else the class),
 *     counts [N, 2] = positives (label >= 1), negatives (label == 0).
 *   mi355det_roi_sample: perm_pos[i] / perm_neg[i] (HOST tables of device pointers) are `torch.randperm(positives_i)` /
 *     `randperm(negatives_i)`, of which the first num_pos[i] / num_neg[i] (HOST) are used exactly as BalancedPositiveNegativeSampler
 *     does (tvision/_utils.py:38-73); outputs for the sum(num_pos + num_neg) samples in image order, ascending candidate index inside an
 *     image: rois [S, 5] = (image, box), out_labels [S], out_matched [S], out_regression_targets [S, 4] = BoxCoder(wx, wy, ww, wh).encode. */
int mi355det_roi_match(const float* proposals, const int32_t* proposal_counts, int32_t n_images, int32_t max_proposals, const float* gt_boxes,
                       const int64_t* gt_labels, const int32_t* gt_offsets, float fg_iou_thresh, float bg_iou_thresh, int32_t row_stride,
                       int32_t* matched, int32_t* labels, int32_t* counts, void* stream);
int mi355det_roi_sample(const float* proposals, const int32_t* proposal_counts, int32_t n_images, int32_t max_proposals, const float* gt_boxes,
                        const int32_t* gt_offsets, int32_t row_stride, const int32_t* matched, const int32_t* labels,
                        const int64_t* const* perm_pos, const int64_t* const* perm_neg, const int32_t* num_pos, const int32_t* num_neg, float wx,
                        float wy, float ww, float wh, float* rois, int64_t* out_labels, int64_t* out_matched, float* out_regression_targets,
                        void* stream);

/* box_iou + Matcher.__call__ (+ set_low_quality_matches_) fused, never materialising [M,N]
 * (tvision/_utils.py:271-344 after retinanet.py:409).  gt [M,4], anchors [N,4] xyxy.
 * out matches [N] int64 in {-2,-1,0..M-1}; gt_best [M] uint32 scratch. */
int mi355det_match_anchors(const float* gt, const float* anchors, int32_t m, int64_t n, float high_thr,
                           float low_thr, int allow_low_quality, uint32_t* gt_best, int64_t* matches,
                           void* stream);

/* BoxCoder.encode_single / decode_single (tvision/_utils.py:79-125,190-223).  decode: codes [n,4*k] */
int mi355det_box_encode(const float* reference_boxes, const float* proposals, float* out, int64_t n,
                        float wx, float wy, float ww, float wh, void* stream);
int mi355det_box_decode(const float* codes, const float* boxes, float* out, int64_t n, int32_t k,
                        float wx, float wy, float ww, float wh, float clip, void* stream);

/* AnchorGenerator.grid_anchors for one level (tvision/anchor_utils.py:98-134):
 * out [gh*gw*a,4] = shifts(y outer, x inner) + cell[a] ; cell [a,4] from the host (rounded). */
int mi355det_anchor_grid(const float* cell, int32_t a, int32_t gh, int32_t gw, int32_t stride_h,
                         int32_t stride_w, float* out, void* stream);

/* torchvision.ops.sigmoid_focal_loss (retinanet.py:137-141) fused forward (+sum) and backward.
 *   x,t [n] ; scale [k] or NULL multiplies logits per class column (tfidf, retinanet.py:138) with
 *   n = rows*k; valid [rows] uint8 or NULL row mask; loss_sum [1] (atomically accumulated, caller
 *   zeroes), grad [n] or NULL = d(sum)/dx * grad_scale. */
int mi355det_sigmoid_focal_loss(const float* x, const float* t, const float* scale, const uint8_t* valid,
                                int64_t rows, int32_t k, float alpha, float gamma, float grad_scale,
                                float* loss_sum, float* grad, void* stream);
/* reduction = 'none' - torchvision's DEFAULT (the reference only calls 'sum': tvision/retinanet.py:137-141, roi_heads.py:57-58): the
 * unreduced loss per element and, if grad != NULL, d loss / d x per element; x, t, loss, grad are n contiguous floats. */
int mi355det_sigmoid_focal_loss_elem(const float* x, const float* t, int64_t n, float alpha, float gamma, float* loss,
                                     float* grad, void* stream);

/* RetinaNet classification loss without the dense one-hot target (retinanet.py:107-143):
 *   logits [rows,k], matched [rows] int64 (Matcher output), gt_labels [M] int64. Same outputs. */
int mi355det_retina_cls_loss(const float* logits, const int64_t* matched, const int64_t* gt_labels,
                             const float* scale, int64_t rows, int32_t k, float alpha, float gamma,
                             float grad_scale, float* loss_sum, float* grad, void* stream);

/* RetinaNetHead.compute_loss for a WHOLE batch in three launches (retinanet.py:56-62,107-143,196-223):
 *   cls_logits [n_images, rows_per_image, k], bbox_regression [n_images, rows_per_image, 4] (the level-concatenated head
 *   outputs), anchors [rows_per_image, 4] (same image size for the batch), matched [n_images, rows_per_image] int64
 *   (Matcher output: >=0 GT index within the image, -1 background, -2 between thresholds), gt_boxes [sum M,4] /
 *   gt_labels [sum M] packed over images with gt_offsets [n_images+1].
 *   losses[0] = classification (focal sum / max(1,num_fg) averaged over images), losses[1] = bbox_regression (L1 on
 *   BoxCoder(1,1,1,1) targets, same normalisation); num_fg [n_images] scratch/out; grads (nullable) = d(loss)*grad_scale. */
int mi355det_retina_loss(const float* cls_logits, const float* bbox_regression, const float* anchors, const int64_t* matched,
                         const float* gt_boxes, const int64_t* gt_labels, const int32_t* gt_offsets,
                         const float* class_scale, int32_t n_images, int64_t rows_per_image, int32_t k, float alpha,
                         float gamma, float grad_scale, float* num_fg, float* losses, float* grad_logits,
                         float* grad_regression, void* stream);

/* The same head loss with the classification gradient written where the head's backward reads it (replaces the autograd hand-over between
 * retinanet.py:107-143 and the cls_logits convolution, retinanet.py:75-105): bf16, per pyramid level an NHWC buffer [n_images, pixels, ld]
 * whose channel a*k + c is anchor a, class c of the pixel (exactly the layout of the level-concatenated logits [n, sum HWA, k]).  Saves the
 * fp32 gradient tensor and its cast (2 x 4 B per logit: 9 GB per step for the 1204-class head at batch 8).  Padding channels of the
 * buffers are not written (zero them once).  sum(pixels[q]) * anchors_per_pixel must equal rows_per_image. */
typedef struct {
  int32_t n_levels;             /* 1..8 */
  int32_t anchors_per_pixel;
  int64_t pixels[8];            /* h*w of each level */
  void* grad[8];                /* bf16 [n_images, pixels, grad_ld] */
  int32_t grad_ld[8];           /* >= anchors_per_pixel * k */
} mi355det_level_grads;
int mi355det_retina_loss_lv(const float* cls_logits, const float* bbox_regression, const float* anchors, const int64_t* matched,
                            const float* gt_boxes, const int64_t* gt_labels, const int32_t* gt_offsets, const float* class_scale,
                            int32_t n_images, int64_t rows_per_image, int32_t k, float alpha, float gamma, float grad_scale,
                            float* num_fg, float* losses, const mi355det_level_grads* cls_levels, float* grad_regression, void* stream);

/* torchvision.ops.roi_align / MultiScaleRoIAlign (tvision/frcnn.py:208-211, roi_heads.py:818): NCHW fp32 features.
 *   feats/hs/ws/scales: HOST arrays of num_levels (1..4) device pointers / sizes / spatial scales; with several levels
 *   the LevelMapper (k = floor(4 + log2(sqrt(area)/224) + 1e-6) clamped to [k_min,k_max]) picks the level per RoI.
 *   rois [K,5] = (batch index, x1,y1,x2,y2).  Forward: out [K,C,ph,pw].  Backward (grad_out != NULL): scatters
 *   grad_out into grad_feats (fp32 atomics, caller zeroes), `out` unused. */
int mi355det_roi_align(const float* const* feats, const int32_t* hs, const int32_t* ws, const float* scales,
                       int32_t num_levels, const float* rois, int32_t num_rois, int32_t channels,
                       int32_t pooled_h, int32_t pooled_w, int32_t sampling_ratio, int aligned, int32_t k_min,
                       int32_t k_max, float* out, const float* grad_out, float* const* grad_feats, void* stream);

/* Channels-last form of the same op for the engines' native layout: feats = bf16 NHWC [n,h,w,C] with pixel pitch lds[q]
 * (NULL = C); out / grad_out stay [K,C,ph,pw] fp32 (the order box_head.fc6 consumes, frcnn.py:248-262); grad_feats = dense
 * fp32 NHWC [n,h,w,C] buffers, zeroed by the caller.  Lanes run over channels, so the backward's atomics are contiguous
 * 256-byte runs instead of 64 scattered rows. */
int mi355det_roi_align_nhwc(const void* const* feats, const int32_t* hs, const int32_t* ws, const int32_t* lds,
                            const float* scales, int32_t num_levels, const float* rois, int32_t num_rois,
                            int32_t channels, int32_t pooled_h, int32_t pooled_w, int32_t sampling_ratio, int aligned,
                            int32_t k_min, int32_t k_max, float* out, const float* grad_out, float* const* grad_feats,
                            void* stream);

/* Tensor.topk(k, dim=1) on [rows, n] (tvision/rpn.py:215-228, retinanet.py:437-445): indices (and values) of the k
 * largest entries per row in descending order, ties -> lower index; entries <= min_value are never selected
 * (the "> score_thresh" filter).  idx_out/val_out [rows,k], count_out [rows] = number selected; k <= 16384. */
int mi355det_topk(const float* x, int32_t rows, int64_t n, int64_t row_stride, int32_t k, float min_value,
                  int64_t* idx_out, float* val_out, int32_t* count_out, void* stream);
/* the same for LONG rows (n >= 65536: several workgroups per row, 3 histogram launches + collect + sort); shorter rows are forwarded to
 * mi355det_topk.  workspace: mi355det_topk_workspace(rows) bytes. */
size_t mi355det_topk_workspace(int32_t rows);
int mi355det_topk_ws(const float* x, int32_t rows, int64_t n, int64_t row_stride, int32_t k, float min_value, int64_t* idx_out,
                     float* val_out, int32_t* count_out, void* workspace, size_t workspace_bytes, void* stream);

/* The same selection for rows cut into up to 8 SEGMENTS (the pyramid levels of RegionProposalNetwork._get_top_n_idx, tvision/rpn.py:215-228):
 * segment s of row r is x[r*row_stride + seg_start[s] ...][0..seg_n[s]), its k = seg_k[s] <= seg_n[s]; outputs per segment idx_out[s]
 * [rows, k] (index WITHIN the segment), val_out[s] [rows, k] (val_out or its entries may be NULL), count_out[s] [rows].  Segments shorter
 * than 65536 share ONE launch (a workgroup per row and segment), longer ones take the mi355det_topk_ws route.  seg_* and the pointer
 * tables are HOST arrays.  workspace: mi355det_topk_workspace(rows) bytes. */
int mi355det_topk_segments(const float* x, int32_t rows, int64_t row_stride, int32_t nseg, const int64_t* seg_start, const int64_t* seg_n,
                           const int32_t* seg_k, float min_value, int64_t* const* idx_out, float* const* val_out, int32_t* const* count_out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolution path (yolo/nets/backbone/darknet.py:13-20,41-43,64-66; yolo/nets/yolohead.py:41-61):
 * NHWC bf16 activations, fp32 accumulation on MFMA.  See conv section in DESIGN.md.
 */
typedef struct {
  int32_t n, h, w, cin;        /* input  [n,h,w,cin], pixel pitch in_ld elements */
  int32_t ho, wo, cout;        /* output [n,ho,wo,cout], pixel pitch out_ld */
  int32_t ksize, stride, pad;  /* 1 or 3; 1 or 2; (k-1)/2 */
  int32_t in_ld, out_ld;
} mi355det_conv_shape;

/* Forward implicit GEMM: y = conv(x, w) [+ bias]; w packed [cout_pad][k*k*cin] bf16 (K contiguous).
 *   out_f32 != 0: y is fp32 (head conv_out), else bf16.
 *   stats != NULL: per pixel-tile partial sums / sums of squares of y, stats[row][0][c], stats[row][1][c]
 *   (fp32, row pitch 2*cout_pad, plain stores, deterministic) for training BatchNorm. */
int mi355det_conv_fwd(const mi355det_conv_shape* s, const void* x, const void* w, const float* bias,
                      void* y, int out_f32, float* stats, int32_t cout_pad, void* stream);
/* Forward with a fused affine epilogue: y = relu?( conv(x,w) * scale[co] + shift[co] + residual ) — FrozenBatchNorm2d
 * (+ReLU, + the Bottleneck identity add, utilities/resnet.py:116-141) or a plain bias (FPN / RetinaNet head convs,
 * retinanet.py:86-97) without a separate elementwise pass. */
typedef struct {
  const float* scale;        /* [cout] or NULL (= 1) */
  const float* shift;        /* [cout] or NULL (= 0) */
  const void* residual;      /* bf16 [n,ho,wo,cout] with pixel pitch residual_ld, or NULL; bf16 outputs only */
  int32_t residual_ld;
  int32_t relu;              /* 0: none; 1: ReLU after scale/shift/residual (ResNet); 2: LeakyReLU(slope) after scale/shift and
                                BEFORE the residual (Darknet residual block, darknet.py:23-35) */
  int64_t out_image_stride;  /* fp32 outputs: elements between images of y (0 = ho*wo*out_ld): heads write straight
                                into the level-concatenated [N, sum HWA, K] tensor (retinanet.py:163-170) */
  float slope;               /* relu == 2: negative slope */
} mi355det_conv_epilogue;
int mi355det_conv_fwd_ex(const mi355det_conv_shape* s, const void* x, const void* w, const mi355det_conv_epilogue* e,
                         void* y, int out_f32, int32_t cout_pad, void* stream);
/* plan-build helper (synchronises; never part of the step): while the mode is on, mi355det_conv_fwd / _dgrad time
 * their candidate tile configurations on the caller's buffers and remember the fastest per shape. */
int mi355det_conv_autotune_mode(int on);

/* ---- tune record.  The plan build of both engines (yolo/nets/engine.py:_autotune, tvision/engine.py:_autotune; reference: none - the
 * reference lets cuDNN pick its algorithms per process, torch.backends.cudnn.benchmark, yolo/main.py) chooses by TIMING: the tile configuration
 * of every implicit-GEMM shape, the form of the stride-2 data gradient, the split count of every weight gradient.  Each choice fixes a
 * summation order, so the numbers of a step depend on it.  The record makes the choices data:
 *   export   copies the current choices as 16-byte entries sorted by (table, key) into buf (if cap suffices); returns the bytes needed
 *   import   adds (replace = 0) or replaces (1) the choices from a record
 *   lock     on = 1: a shape that HAS an entry is never timed again (plan builds reuse the record; shapes without an entry are timed and
 *            added as before); returns the previous state
 *   clear    drops every choice and the lock (the next plan build times everything)
 * Host mirror: object_detectors_amd/tune.py (JSON files, MI355DET_TUNE_SAVE / MI355DET_TUNE_LOAD, rank-0 broadcast for N > 1). */
typedef struct {
  uint32_t table;   /* 0 implicit-GEMM tile configuration, 1 stride-2 data-gradient form, 2 weight-gradient split count (+ 65536: the 256 x 256
                       phase-staggered kernel; a shape it does not take falls back to the 128 x 128 kernel with the same split count) */
  int32_t value;
  uint64_t key;     /* shape key of that table */
} mi355det_tune_entry;
size_t mi355det_tune_export(void* buf, size_t cap);
int mi355det_tune_import(const void* buf, size_t bytes, int replace);
int mi355det_tune_lock(int on);
int mi355det_tune_clear(void);
/* number of rows of the `stats` partial buffer [rows][2][cout_pad] the forward writes (one per pixel tile);
 * allocate rows+64: bn_finalize uses the 64 spare rows as scratch for its two-stage reduction */
int mi355det_conv_stats_rows(const mi355det_conv_shape* s, int32_t cout_pad);
/* Darknet stem (darknet.py:41, 3->32 3x3): NCHW fp32 image -> im2col rows [n*h*w][32] bf16
 * (k=(kh*3+kw)*3+c, 27 valid) consumed by conv_fwd / conv_wgrad as a 1x1 convolution with cin=32. */
int mi355det_stem_im2col(const float* img, void* out, int32_t n, int32_t h, int32_t w, void* stream);

/* Darknet stem without a stored pre-BN tensor (darknet.py:41-43,74-76: conv1 -> bn1 -> LeakyReLU; csrc/stem_kernels.hip).  The 3->32 3x3
 * convolution is recomputed from the fp32 NCHW image in every pass (K = 27: cheaper than one pass over the 32-channel tensor), so the
 * step never writes z, the im2col matrix or dz of this layer.  w = packed forward weights [32][32] bf16, k = (kh*3+kw)*3 + c
 * (mi355det_pack_weights of the [32, 32] "1x1" master the engine keeps).  Needs h % 8 == 0 and w % 32 == 0 (MI355DET_EINVAL otherwise).
 *   stem_rows            number of partial rows the kernels write (one per persistent workgroup); allocate rows + 64 for
 *                        bn_finalize / bn_bwd_sum_partials, rows * 1024 floats for the wgrad slab
 *   stem_fwd_stats       partial[rows][2][32]: per-channel sum z, sum z*z (fp32 z)            -> mi355det_bn_finalize
 *   stem_fwd_apply       a = lrelu(z*scale + shift) as bf16 NHWC (pitch a_ld); training AND eval (scale_shift from
 *                        mi355det_bn_finalize or mi355det_bn_eval_scale_shift)
 *   stem_bwd_reduce      partial[rows][2][32]: sum dy, sum dy*xhat, dy = da * lrelu'(z*scale+shift) -> mi355det_bn_bwd_sum_partials
 *   stem_bwd_apply_wgrad dz = scale*(dy - mean dy - xhat*mean(dy*xhat)) feeds the weight-gradient MFMA directly:
 *                        dw[32][32] += dz^T * im2col (fixed order via slab[rows][1024]); dgamma += sums[32..63], dbeta += sums[0..31]
 * All four are fixed-order (bit-reproducible). */
/* Stem activation + the first down-sampling convolution fused (csrc/stem_l1_kernels.hip; darknet.py:41-43,64-66,74-76): per 8 x 16 tile of
 * z1 = layer1.ds_conv(a0) the kernel recomputes a0 = lrelu(bn1(conv1(img))) (scale_shift0 from mi355det_bn_finalize of stem_fwd_stats) into
 * LDS and convolves it there (32 -> 64, 3x3, stride 2, pad 1; w1 = that layer's forward pack [64][9*32]).  Writes z1 (bf16 NHWC, pitch
 * z1_ld >= 64), its BatchNorm partial statistics stats[rows + 64][2][64] (rows = mi355det_stem_l1_rows) and - when a0 != NULL - the
 * activation itself (pitch a0_ld >= 32) for the layer's weight gradient.  Replaces stem_fwd_apply + conv_fwd of that layer in training.
 * Needs h % 16 == 0 and w % 32 == 0. */
int mi355det_stem_l1_rows(int32_t n, int32_t h, int32_t w);
int mi355det_stem_l1_fwd(const float* img, const void* w0, const float* scale_shift0, float slope, const void* w1, void* a0,
                         int32_t a0_ld, void* z1, int32_t z1_ld, float* stats, int32_t n, int32_t h, int32_t w, void* stream);
/* Inference form of the same launch: stem activation (folded BN: scale_shift0 from mi355det_bn_eval_scale_shift) + layer1.ds_conv + ITS folded
 * BN + LeakyReLU; a1 [n, h/2, w/2, 64] bf16 is the layer's ACTIVATION, the stem activation is never stored (839 MB written and read back at
 * batch 32 / 640 px otherwise).  scale_shift1 = scale[64] | shift[64] of layer 1. */
int mi355det_stem_l1_fwd_eval(const float* img, const void* w0, const float* scale_shift0, float slope, const void* w1,
                              const float* scale_shift1, void* a1, int32_t a1_ld, int32_t n, int32_t h, int32_t w, void* stream);
int mi355det_stem_rows(int32_t n, int32_t h, int32_t w);
int mi355det_stem_fwd_stats(const float* img, const void* w, float* partial, int32_t n, int32_t h, int32_t wd, void* stream);
int mi355det_stem_fwd_apply(const float* img, const void* w, const float* scale_shift, float slope, void* a, int32_t a_ld,
                            int32_t n, int32_t h, int32_t wd, void* stream);
int mi355det_stem_bwd_reduce(const float* img, const void* w, const float* scale_shift, float slope, const void* da,
                             int32_t da_ld, float* partial, int32_t n, int32_t h, int32_t wd, void* stream);
int mi355det_stem_bwd_apply_wgrad(const float* img, const void* w, const float* scale_shift, const float* sums, float slope,
                                  const void* da, int32_t da_ld, float* slab, float* dw, float* dgamma, float* dbeta,
                                  int32_t n, int32_t h, int32_t wd, void* stream);

/* The stem's backward in ONE pass over the activation gradient instead of stem_bwd_reduce + stem_bwd_apply_wgrad (same reference lines:
 * the autograd of darknet.py:41-43,74-76).  stem_bwd_fused accumulates, on MFMA over the pixels, A[32][32] = dy^T [im2col | 1] and the Gram
 * matrix G[32][32] = [im2col | 1]^T [im2col | 1] into slab[rows][2][32][32] (rows = mi355det_stem_rows), folds them into ag[2][32][32] and
 * derives sums[64] = (sum dy, sum dy*xhat) from A (sum dy = A[:,27], sum dy*z = <W, A>).  stem_bwd_finish then forms, from ag and the -
 * possibly rank-averaged, SyncBN - sums, dW += scale*(A - mean(dy)*B - mean(dy*xhat)*invstd*(W G - mean*B)) with B = G[27,:], and adds
 * dgamma += sums[32:], dbeta += sums[:32] (both nullable together).  count = n*h*w of THIS rank.  dw is [32][32] fp32 (k padded). */
int mi355det_stem_bwd_fused(const float* img, const void* w, const float* scale_shift, float slope, const void* da, int32_t da_ld,
                            float* slab, float* ag, float* sums, int32_t n, int32_t h, int32_t wd, void* stream);
int mi355det_stem_bwd_finish(const void* w, const float* scale_shift, const float* ag, const float* sums, int64_t count, float* dw,
                             float* dgamma, float* dbeta, void* stream);

/* Data gradient: dx = conv_transpose(dy, w); wt packed for dgrad by mi355det_pack_weights.
 * residual != NULL adds a bf16 tensor (same shape as dx) in the epilogue (residual-block skip). */
int mi355det_conv_dgrad(const mi355det_conv_shape* s, const void* dy, const void* wt, void* dx,
                        const void* residual, int32_t residual_ld, void* stream);

/* The same data gradient with a caller-owned workspace (replaces the same autograd step, torch/nn/grad.py conv2d_input as reached from
 * loss.backward() in detection/engine.py:49-57).  When the shape has few output pixels and a very deep reduction (stride 1, <= 96 tiles
 * of 128 x 128, k*k*cout >= 8192: the 1204-class RetinaNet head on the small pyramid levels, retinanet.py:75-105) the reduction is split
 * over channel ranges into fp32 partial tiles in `workspace` and added in a fixed order (deterministic; the bf16 rounding happens once,
 * after the sum and the residual).  mi355det_conv_dgrad_workspace returns the bytes that form needs, 0 when it does not apply; with a
 * NULL / zero workspace or a shape it does not apply to, conv_dgrad_ws IS conv_dgrad. */
/* Data gradient with the FrozenBatchNorm2d / ReLU backward of the layer that produced the convolution's input folded into the epilogue
 * (replaces, for a bottleneck's conv1 -> conv2 -> conv3 chain and the head towers, the autograd steps of F.relu and of the frozen affine,
 * utilities/resnet.py:107-125, tvision/backbone_utils.py:33-50, between two conv2d_input calls):
 *   dx = bf16(conv2d_input(dy)) * scale[c] * (act > 0)   (relu != 0)   |   bf16(conv2d_input(dy)) * scale[c]   (relu == 0)
 * act = the stored activation of that layer [n,h,w,cin] (pitch act_ld), scale [cin] or NULL.  Bit-identical to mi355det_conv_dgrad followed
 * by mi355det_relu_affine_bwd.  Stride-1 convolutions only. */
int mi355det_conv_dgrad_mask(const mi355det_conv_shape* s, const void* dy, const void* wt, void* dx, const void* act, int32_t act_ld,
                             const float* scale, int32_t relu, void* stream);
size_t mi355det_conv_dgrad_workspace(const mi355det_conv_shape* s);
int mi355det_conv_dgrad_ws(const mi355det_conv_shape* s, const void* dy, const void* wt, void* dx, const void* residual, int32_t residual_ld,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Data gradient that also starts the BatchNorm backward of the layer it feeds: dx is the gradient of an activation
 * a = lrelu(bn(z)); while the dx tile is still on chip the epilogue accumulates that layer's per-channel
 *   sum dy  and  sum dy*xhat,   dy = dx * lrelu'(z*scale+shift),  xhat = (z-mean)*invstd
 * into `partials` [rows+64][2][cin_pad] (plain stores, deterministic; rows = mi355det_conv_dgrad_bn_rows(s), the 64
 * spare rows are scratch), saving the separate bn_act_bwd_reduce pass its re-read of dx.  mi355det_bn_bwd_sum_partials
 * folds them into sums[2*cin] (same layout as bn_act_bwd_reduce).  scale_shift = [4*cin] of the producing layer. */
int mi355det_conv_dgrad_bn_rows(const mi355det_conv_shape* s);
int mi355det_conv_dgrad_bn(const mi355det_conv_shape* s, const void* dy, const void* wt, void* dx, const void* residual,
                           int32_t residual_ld, const void* z, int32_t z_ld, const float* scale_shift, float slope,
                           float* partials, void* stream);
int mi355det_bn_bwd_sum_partials(const float* partials, int32_t rows, int32_t c, int32_t c_pad, float* sums,
                                 void* stream);

/* Weight gradient: dw[cout][k*k*cin] fp32 += x^T dy.  Split over the pixel axis; partial 128x128 tiles go to
 * `workspace` with plain stores and are summed into dw in a fixed order: dw is bit-reproducible from run to run.
 * dbias != NULL: dbias[cout] += sum_pixels dy - this column sum ends in one fp32 atomicAdd per workgroup and channel,
 * so dbias (the three YOLO head biases, the RetinaNet / RPN head biases) is reproducible only to fp32 rounding order. */
size_t mi355det_conv_wgrad_workspace(const mi355det_conv_shape* s);
/* plan-build helper (synchronises; not part of the step): times the candidate split counts for this shape on the
 * caller's buffers and remembers the fastest.  Returns the chosen split count; dw is clobbered. */
int mi355det_conv_wgrad_autotune(const mi355det_conv_shape* s, const void* x, const void* dy, float* dw,
                                 void* workspace, size_t workspace_bytes, void* stream);
int mi355det_conv_wgrad(const mi355det_conv_shape* s, const void* x, const void* dy, float* dw,
                        float* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* fp32 master weights, torch OIHW [cout][cin][k][k] or engine OHWI [cout][k][k][cin] (w_is_ohwi) -> bf16 fwd pack [cout_pad][k][k][cin] and
 * dgrad pack (stride 1: [cin_pad][k][k][cout] taps flipped; stride 2: 4 parity classes). */
size_t mi355det_dgrad_pack_elems(const mi355det_conv_shape* s);
int mi355det_pack_weights(const mi355det_conv_shape* s, const float* w, int w_is_ohwi, void* w_fwd,
                          int32_t cout_pad, void* w_dgrad, void* stream);
/* Batched form: all layers of a model re-packed by ONE launch per step.  The host table (entries + block table) is
 * built once from `items`, copied to the device by the caller, and replayed with mi355det_pack_weights_batched. */
typedef struct {
  const float* w;      /* device fp32 master */
  void* w_fwd;         /* device bf16 forward pack or NULL */
  void* w_dgrad;       /* device bf16 dgrad pack or NULL */
  mi355det_conv_shape shape;
  int32_t cout_pad, w_is_ohwi;
} mi355det_pack_item;
size_t mi355det_pack_table_bytes(const mi355det_pack_item* items, int32_t n, int32_t* n_entries, int32_t* n_blocks);
int mi355det_pack_table_build(const mi355det_pack_item* items, int32_t n, void* host_table, size_t host_bytes);
int mi355det_pack_weights_batched(const void* dev_table, int32_t n_entries, int32_t n_blocks, void* stream);
/* dw fp32 [cout][k][k][cin] -> torch layout [cout][cin][k][k] (param.grad) */
int mi355det_unpack_wgrad(const mi355det_conv_shape* s, const float* dw, float* grad, void* stream);

/* BatchNorm2d (training) + LeakyReLU(0.1) around the conv (darknet.py:15-16,19-20):
 *   bn_finalize: stats -> scale/shift (+ running stats update, momentum 0.1, unbiased var).
 *   bn_act_fwd: a = lrelu(z*scale+shift) [+ residual]   (bf16 in/out, vectorised)
 *   bn_act_bwd_reduce: per-channel sums of dy and dy*xhat where dy = (g1[+g2]) * lrelu'(.)
 *   bn_act_bwd_apply: dz = scale*(dy - mean(dy) - xhat*mean(dy*xhat))  (bf16)
 * Limits and reproducibility:
 *   - c must be a multiple of 8 everywhere (MI355DET_EINVAL otherwise);
 *   - bn_act_bwd_reduce (legacy form) finishes every workgroup with one fp32 atomicAdd per channel into `sums`, so the two sums - and
 *     through them dz, dgamma, dbeta and everything upstream - differ from run to run in the last bits (measured ~6e-4 of max on the
 *     final gradient of a 75-layer step).  mi355det_bn_act_bwd_reduce_det below is the fixed-order form the engines use. */
int mi355det_bn_finalize(const float* stats, int32_t rows, int32_t c, int32_t c_pad, int64_t count,
                         const float* gamma, const float* beta, float eps, float momentum,
                         float* running_mean, float* running_var,
                         float* scale_shift /* [4*c]: scale, shift, mean, invstd */, void* stream);
/* SyncBN forward (apex convert_syncbn_model, yolo/procedures/initialize.py:31-32): the per-rank partial rows are folded in double
 * (exactly bn_finalize's own accumulation) into sums64 [2][c_pad] = (sum x | sum x*x), which the caller all-reduces as fp64 across the
 * ranks; bn_finalize_f64 then produces the same scale / shift / running statistics bn_finalize would from the global batch
 * (count = global element count per channel).  stats needs the 64 spare rows like bn_finalize. */
int mi355det_bn_fold_partials_f64(const float* stats, int32_t rows, int32_t c, int32_t c_pad, double* sums64, void* stream);
int mi355det_bn_finalize_f64(const double* sums64, int32_t c, int32_t c_pad, int64_t count, const float* gamma, const float* beta,
                             float eps, float momentum, float* running_mean, float* running_var, float* scale_shift,
                             void* stream);
/* eval mode (model.eval(), test_one_epoch.py:10): scale/shift from the running statistics */
int mi355det_bn_eval_scale_shift(int32_t c, const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, float eps, float* scale_shift, void* stream);
/* out = a + b on bf16 NHWC channel slices (gradient join of the concat/upsample branches) */
int mi355det_add_bf16(const void* a, int32_t a_ld, const void* b, int32_t b_ld, int32_t c, int64_t pixels,
                      void* out, int32_t out_ld, void* stream);
int mi355det_bn_act_fwd(const void* z, int32_t z_ld, const float* scale_shift, int32_t c, int64_t pixels,
                        float slope, const void* residual, int32_t res_ld, void* out, int32_t out_ld,
                        void* stream);
int mi355det_bn_act_bwd_reduce(const void* g1, int32_t g1_ld, const void* g2, int32_t g2_ld, const void* z,
                               int32_t z_ld, const float* scale_shift, int32_t c, int64_t pixels,
                               float slope, float* sums /* [2*c] zeroed by caller */, void* stream);
/* Fixed-order form of the same sums (what the engines use since round 4; reference: torch's batch_norm_backward is deterministic too,
 * yolo/nets/backbone/darknet.py:15-16): every workgroup stores its partial sums as one row of `workspace`, a second small launch adds the
 * rows in row order and WRITES sums[2*c] (no zeroing needed).  workspace: mi355det_bn_act_bwd_reduce_workspace(c, pixels) bytes, 16-byte
 * aligned, no initialisation; launches sharing a workspace must be ordered on one stream.  Bit-reproducible from run to run. */
size_t mi355det_bn_act_bwd_reduce_workspace(int32_t c, int64_t pixels);
int mi355det_bn_act_bwd_reduce_det(const void* g1, int32_t g1_ld, const void* g2, int32_t g2_ld, const void* z,
                                   int32_t z_ld, const float* scale_shift, int32_t c, int64_t pixels,
                                   float slope, float* sums, void* workspace, size_t workspace_bytes, void* stream);
int mi355det_bn_act_bwd_apply(const void* g1, int32_t g1_ld, const void* g2, int32_t g2_ld, const void* z,
                              int32_t z_ld, const float* scale_shift, const float* sums, const float* gamma,
                              int32_t c, int64_t pixels, float slope, void* dz, int32_t dz_ld,
                              float* dgamma, float* dbeta, void* stream);

/* nn.Upsample(scale_factor=2, nearest) into a channel slice (yolohead.py:32,80-81) and its adjoint */
int mi355det_upsample2x_fwd(const void* x, int32_t x_ld, int32_t n, int32_t h, int32_t w, int32_t c, void* out,
                            int32_t out_ld, void* stream);
int mi355det_upsample2x_bwd(const void* g, int32_t g_ld, int32_t n, int32_t h, int32_t w, int32_t c, void* out,
                            int32_t out_ld, void* stream);

/* ResNet stem as one direct convolution (utilities/resnet.py:173-176,232-234: conv1 7x7 / 2 / pad 3, 3 -> 64 + FrozenBatchNorm2d + ReLU;
 * the normalisation of tvision/transform.py:120-124 when mean / inv_std are given): replaces mi355det_im2col_nchw + the 160-deep GEMM of
 * mi355det_conv_fwd_ex for the frozen stem (no im2col matrix: 819 MB written and re-read at batch 16 / 800 px).  img [n,3,h,w] fp32,
 * w = the forward pack [64][160] bf16 (k = (kh*7+kw)*3 + c), scale / shift [64] or NULL, out [n, h/2, w/2, 64] bf16 (pitch out_ld).
 * h and w multiples of 32 (what GeneralizedRCNNTransform.batch_images pads to).  relu: bit 0 = ReLU; bits 1-3 are timing ablations of
 * tools/bench_rstem.py (skip the fragment compute / the halo fetch / the halo LDS store: wrong results) and must be 0 otherwise. */
int mi355det_resnet_stem_fwd(const float* img, const float* mean, const float* inv_std, const void* w, const float* scale,
                             const float* shift, int32_t relu, void* out, int32_t out_ld, int32_t n, int32_t h, int32_t wd,
                             void* stream);

/* ---- ResNet-FPN / RetinaNet elementwise companions (csrc/resnet_kernels.hip) -------------------------------------------
 * im2col_nchw: NCHW fp32 image, optional per-channel (x-mean)*inv_std (GeneralizedRCNNTransform.normalize,
 *   tvision/transform.py:120-124) -> rows [n*ho*wo][kpad] bf16, k=(kh*ks+kw)*c+ch: the 7x7/2 stem (resnet.py:173) runs
 *   as a 1x1 convolution with cin=kpad on the MFMA path.
 * maxpool3x3s2: nn.MaxPool2d(3,2,1) (resnet.py:176); _bwd: its adjoint (gradient to the first maximum of every window, gather form,
 *   deterministic) - needed when the 7x7 stem is trained (trainable_backbone_layers = 5, backbone_utils.py:100-104).
 * relu_affine_bwd: gm = (g1 [+g2]) * [a>0] (if relu), dz = gm * scale[c] (scale NULL = 1); gm / dz nullable.
 * upsample_nearest_add: out = lateral + interpolate(x, size=(out_h,out_w), nearest) (FPN top-down); _bwd its adjoint
 *   (out = accumulate + sum of g over the pixels that read it).
 * cast_rows_bf16: dst[b][r][0..cols) = src[b*image_stride + r*row_stride + c] * mul, zero fill up to dst_ld. */
int mi355det_im2col_nchw(const float* img, const float* mean, const float* inv_std, void* out, int32_t n, int32_t c,
                         int32_t h, int32_t w, int32_t ksize, int32_t stride, int32_t pad, int32_t kpad, void* stream);
int mi355det_maxpool3x3s2(const void* x, int32_t x_ld, int32_t n, int32_t h, int32_t w, int32_t c, void* out,
                          int32_t out_ld, void* stream);
int mi355det_maxpool3x3s2_bwd(const void* x, int32_t x_ld, const void* g, int32_t g_ld, int32_t n, int32_t h, int32_t w,
                              int32_t c, void* dx, int32_t dx_ld, void* stream);
int mi355det_relu_affine_bwd(const void* g1, int32_t g1_ld, const void* g2, int32_t g2_ld, const void* a, int32_t a_ld,
                             const float* scale, int32_t c, int64_t pixels, int relu, void* dz, int32_t dz_ld,
                             void* gm, int32_t gm_ld, void* stream);
int mi355det_upsample_nearest_add(const void* x, int32_t x_ld, int32_t n, int32_t h, int32_t w, int32_t c,
                                  const void* lateral, int32_t lateral_ld, int32_t out_h, int32_t out_w, void* out,
                                  int32_t out_ld, void* stream);
int mi355det_upsample_nearest_bwd(const void* g, int32_t g_ld, int32_t n, int32_t h, int32_t w, int32_t c, int32_t g_h,
                                  int32_t g_w, const void* accumulate, int32_t accumulate_ld, void* out,
                                  int32_t out_ld, void* stream);
int mi355det_cast_rows_bf16(const float* src, int64_t src_image_stride, int64_t src_row_stride, int32_t n, int64_t rows,
                            int32_t cols, float mul, void* dst, int32_t dst_ld, void* stream);

/* ---- fused optimizer step on the flat fp32 buffers --------------------------------------------------
 * Replaces torch.optim.SGD / Adam .step() (+ zero_grad) of yolo/procedures/initialize.py:38,41 and
 * yolo/procedures/train_one_epoch.py:96: one pass over {param, grad, state}.  grad_scale = 1/loss_scale (apex amp,
 * train_one_epoch.py:89) folded in.  first_step: momentum buffer is initialised with the gradient (torch semantics). */
int mi355det_sgd_step(float* w, float* g, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                      float weight_decay, float grad_scale, int nesterov, int first_step, int zero_grad, void* stream);
int mi355det_adam_step(float* w, float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                       float beta2, float eps, float weight_decay, float grad_scale, int32_t step, int zero_grad,
                       void* stream);
/* The amp half of that row (apex `amp.scale_loss` + dynamic loss scaling around `optimizer.step()`, yolo/procedures/initialize.py:44-45,
 * train_one_epoch.py:88-96): grad_nonfinite sets *flag (device int32, zeroed by the caller) to 1 when any gradient is inf / nan (one read
 * pass); the *_guarded steps read *skip_flag on the device and leave parameters and optimizer state untouched when it is non-zero (the
 * gradients are still cleared when zero_grad is set), so an overflowing step is skipped without a host round trip.  skip_flag NULL = the
 * plain step. */
int mi355det_grad_nonfinite(const float* g, int64_t n, int32_t* flag, void* stream);
int mi355det_sgd_step_guarded(float* w, float* g, float* momentum_buf, int64_t n, float lr, float momentum, float dampening,
                              float weight_decay, float grad_scale, int nesterov, int first_step, int zero_grad,
                              const int32_t* skip_flag, void* stream);
int mi355det_adam_step_guarded(float* w, float* g, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, float grad_scale, int32_t step, int zero_grad,
                               const int32_t* skip_flag, void* stream);


/* ---- Fast R-CNN box-head loss (csrc/frcnn_kernels.hip) ------------------------------------------------------------------
 * Replaces `fastrcnn_loss` (torchvision_models/tvision/roi_heads.py:24-96) as RoIHeads.forward calls it (:826-827):
 *   fastrcnn_loss(tfidf * class_logits, box_regression, labels, regression_targets, weights=classification_weights, loss_type)
 * class_logits [n,k] fp32, box_regression [n,4k], labels [n] int64 (0 = background), regression_targets [n,4];
 * class_scale [k] or NULL = the tf-idf row multiplied into the logits; class_weights [k] or NULL ('ce' only);
 * loss_type 0 'ce', 1 'bce', 2 'focal_loss', 3 'gombit' (incl. its "/4 above 5" branch), 4 'gombit_fl'.
 * losses [2] = (loss_classifier, loss_box_reg); grad_logits [n,k] / grad_box [n,4k] (nullable) = d(loss_classifier)/d(class_logits)
 * (UNSCALED logits) and d(loss_box_reg)/d(box_regression).  Deterministic (no atomics).  Returns EINVAL for n < 1, k < 2, a loss_type
 * outside 0..4 or class weights with a loss other than 'ce'. */
size_t mi355det_fastrcnn_loss_workspace(int32_t n);
int mi355det_fastrcnn_loss(const float* class_logits, const float* box_regression, const int64_t* labels,
                           const float* regression_targets, const float* class_scale, const float* class_weights, int32_t n,
                           int32_t k, int32_t loss_type, float* losses, float* grad_logits, float* grad_box, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ---- input-side transform (csrc/transform_kernels.hip) -------------------------------------------------
 * resize_bilinear: `planes` fp32 planes [planes][h][w] -> out [planes][pad_h][pad_w]: F.interpolate(mode='bilinear', align_corners=False)
 *   to (out_h, out_w) with the sampling scale taken from the two sizes (recompute_scale_factor=True / size=...), zeros in the pad.
 *   mean/std [c] (nullable together): (x - mean[p % c]) / std[p % c] applied to the taps = GeneralizedRCNNTransform.normalize followed by
 *   resize and batch_images (torchvision_models/tvision/transform.py:120-135,215-226) for one image (planes = c = 3) written into its slot
 *   of the padded batch; with mean = NULL and planes = N*C it is the YOLO multi-scale resize (yolo/procedures/train_one_epoch.py:64-69).
 * resize_boxes: transform.py:279-293, boxes [n,4] xyxy scaled by float32 ratios new/orig. */
int mi355det_resize_bilinear(const float* in, int32_t planes, int32_t c, int32_t h, int32_t w, const float* mean, const float* stdv,
                             float* out, int32_t out_h, int32_t out_w, int32_t pad_h, int32_t pad_w, void* stream);
int mi355det_resize_boxes(const float* boxes, float* out, int64_t n, int32_t orig_h, int32_t orig_w, int32_t new_h, int32_t new_w,
                          void* stream);

/* ---- output side: detections -> the numbers of the COCO json dicts (csrc/transform_kernels.hip) ----------
 * yolo/procedures/test_one_epoch.py:41-66 (scale = 1): boxes rows (x1,y1,x2,y2,...) with pitch box_ld in network pixels ->
 *   bbox_xywh [k,4] = (x1/inp_dim*W, y1/inp_dim*H, w, h) in the original image, area [k] = w*h, category_id [k] from the label column
 *   (labels_f32 with pitch label_ld, cast like `.long()`, or labels_i64): label_mode 0 = label + 1, 1 = COCO 80 -> 91 map
 *   (yolo/utilities/helper.py:16-24), 2 = unchanged.
 * torchvision_models/detection/coco_eval.py:83-105,169-171 (scale = 0, label_mode 2): convert_to_xywh.  area / category_id nullable. */
int mi355det_coco_rows(const float* boxes, int32_t box_ld, const float* labels_f32, const int64_t* labels_i64, int32_t label_ld,
                       int64_t k, float inp_dim, float img_h, float img_w, int32_t scale, int32_t label_mode, float* bbox_xywh,
                       float* area, int64_t* category_id, void* stream);

/* layout / dtype converters at the module boundary */
int mi355det_nhwc_to_nchw_f32(const void* x, int x_is_bf16, int32_t x_ld, int32_t n, int32_t c, int32_t h,
                              int32_t w, float* out, void* stream);
int mi355det_nchw_f32_to_nhwc(const float* x, int32_t n, int32_t c, int32_t h, int32_t w, void* out,
                              int out_is_bf16, int32_t out_ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355DET_H */
