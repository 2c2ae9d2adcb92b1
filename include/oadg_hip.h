/*
 * oadg_hip.h - C ABI of liboadg_hip.so: the MI355X (gfx950) kernels behind OA-DG's training hot path.
 *
 * The reference (WoojuLee24/OA-DG, an mmdetection 2.20 fork) has no FFI of its own: its native
 * arithmetic is reached through Python modules resolved by registry name.  Each entry
 * point below names the reference interface it stands in for; the Python modules of `oa-dg_amd/`
 * registered under the reference's names are the callers (INTEGRATION.md shows the ctypes binding).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`; the caller owns all memory
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work: no allocation, no
 *     synchronisation, no global mutable state (re-entrant across streams with distinct workspaces)
 *   - return value: 0 = ok, < 0 = argument error (OADG_EARG -1, OADG_ESIZE -2; the host-side file decoder also
 *     OADG_EIO -3, OADG_EUNSUPPORTED -4), > 0 = hipError_t
 *   - tensors are dense row-major in the stated shape; images/feature maps are NHWC
 */
#ifndef OADG_HIP_H
#define OADG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * OA-Loss: instance-level supervised contrastive loss
 *   replaces ContrastiveLossPlus.forward   mmdet/models/losses/oadg/contrastive_loss_plus.py:31-50
 *            supcontrast / supcontrast_mask mmdet/models/losses/oadg/contrastive_loss.py:147-232
 * feats [B, D] fp32 raw contrastive features (D in {64,128,256}); labels [n_labels] int64, rows
 * n_labels..B-1 (random proposals) take labels[n_labels-1]; ori_size = rows per view of the sampled
 * block (reference: 512*num_views), rp_size = random-proposal rows per view.  out_loss[0] =
 * loss_weight * loss (0 when #foreground rows <= min_samples).  The workspace written by _fwd must be
 * passed unchanged to _bwd; grad_out is a 1-float device scalar (NULL = 1.0).
 */
size_t oadg_supcon_workspace_bytes(int B, int D);
int oadg_supcon_fwd(const float* feats, const int64_t* labels, int B, int D, int n_labels, int ori_size,
                    int rp_size, float temper, int min_samples, float loss_weight, void* workspace,
                    size_t workspace_bytes, float* out_loss, void* stream);
int oadg_supcon_bwd(const int64_t* labels, int B, int D, int n_labels, int ori_size, int rp_size,
                    float temper, float loss_weight, const float* grad_out, void* workspace,
                    size_t workspace_bytes, float* dfeats, void* stream);
int oadg_supcon_status(const void* workspace_host_copy);

/* ------------------------------------------------------------------------------------------------
 * OA-Loss: view-1 cross-entropy + two-view Jensen-Shannon consistency
 *   replaces CrossEntropyLossPlus.forward  mmdet/models/losses/oadg/cross_entropy_loss_plus.py:418-500
 *            cross_entropy :11-58, binary_cross_entropy :82-130, jsdv1_3_2aug :264-319
 * logits [R, C] fp32, rows [0,R/2) = view 1, [R/2,R) = view 2; labels [R] int64 (only the first R/2 are
 * read); weights [R] fp32 or NULL.  mode 0: sigmoid rows (C == 1, target = (label == 0), RPN);
 * mode 1: softmax rows (RoI head).  out3 = {total, ce_part, jsd_part}.
 */
size_t oadg_cls_loss_workspace_bytes(void);
int oadg_ce_jsd_fwd(const float* logits, const int64_t* labels, const float* weights, long R, int C,
                    int mode, float avg_factor, float loss_weight, float lambda_jsd, void* workspace,
                    size_t workspace_bytes, float* out3, void* stream);
int oadg_ce_jsd_bwd(const float* logits, const int64_t* labels, const float* weights, long R, int C,
                    int mode, float avg_factor, float loss_weight, float lambda_jsd, const float* grad_out,
                    float* dlogits, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RoIAlign over an FPN pyramid (aligned / avg / adaptive grid), level assignment included
 *   replaces mmcv.ops.RoIAlign fwd+bwd as built at
 *              mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:54-59
 *            SingleRoIExtractor.map_roi_levels + per-level loop
 *              mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:36-55,89-146
 * feats_host[l] -> [N, H_l, W_l, C] (dtype 0 fp32 / 1 bf16); rois [K,5] fp32 (batch,x1,y1,x2,y2);
 * out [K, PH, PW, C] same dtype.  levels == 1 is the plain single-map RoIAlign.  _bwd accumulates
 * (atomically) into caller-zeroed fp32 maps dfeats_host[l].
 */
int oadg_roi_align_fwd(const void* const* feats_host, const int* heights_host, const int* widths_host,
                       const float* scales_host, int levels, int N, int C, int dtype, float finest_scale,
                       const float* rois, int K, int PH, int PW, int sampling_ratio, int aligned,
                       void* out, const int* order, void* stream);
int oadg_roi_align_bwd(float* const* dfeats_host, const int* heights_host, const int* widths_host,
                       const float* scales_host, int levels, int N, int C, int dtype, float finest_scale,
                       const float* rois, int K, int PH, int PW, int sampling_ratio, int aligned,
                       const void* grad_out, const int* order, void* stream);
/* order (optional, int32 [K], NULL = 0..K-1): workgroup b processes RoI order[b]; results do not depend on it.
 * oadg_roi_order_keys writes an int64 key per RoI (level, image, 16-feature-pixel cell of the centre); processing the RoIs
 * in argsort(keys) order keeps the feature rows / gradient lines of neighbouring RoIs in L2. */
int oadg_roi_order_keys(const float* rois, int K, int n_img, int levels, float finest_scale, long long* keys,
                        void* stream);
/* the same order without a sort call: order [K] = stable argsort of those keys (int32), range [levels * n_img + 1] = the
 * first position in `order` of every (level, image) group - what oadg_roi_align_bwd_tiles takes; K <= 8192 */
int oadg_roi_order(const float* rois, int K, int n_img, int levels, float finest_scale, int* order, int* range,
                   void* stream);
/* The bf16 backward organised by OUTPUT tiles (8 x 8 pixels of one image on one level per workgroup): every element of
 * the bf16 gradient maps dmaps[l] [N,H_l,W_l,C] is WRITTEN (no zero fill, no fp32 maps, no atomics, no cast pass) and the
 * summation order is fixed by `order` (deterministic).  order [K] = RoI indices sorted by the oadg_roi_order_keys keys,
 * range [levels * N + 1] (device int32) = first position in `order` of every (level, image) group.  PH, PW <= 8.
 * tile_boxes: device scratch of 16 K bytes (the tile rectangle of every RoI, written by a first small launch).
 * Same arithmetic as oadg_roi_align_bwd (mmcv RoIAlign backward) up to the fp32 summation order. */
int oadg_roi_align_bwd_tiles(void* const* dmaps, const int* heights, const int* widths, const float* scales,
                             int levels, int N, int C, float finest_scale, const float* rois, int K, int PH, int PW,
                             int sampling_ratio, int aligned, const void* grad_out, const int* order, const int* range,
                             void* tile_boxes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused RPN loss: AnchorHead.loss (mmdet/models/dense_heads/anchor_head.py:402-544) with CrossEntropyLossPlus (sigmoid
 * BCE on the view-1 rows + lambda * jsdv1_3_2aug between the views, cross_entropy_loss_plus.py:82-130,264-319) and
 * L1LossPlus (view-1 rows, smooth_l1_loss_plus.py) for ALL pyramid levels in one forward and one backward launch, reading the
 * RPN head's output in place.  A level: y (dtype 0 fp32 / 1 bf16, logical [B, Cy, H, W], element strides sN, sC, sH, sW;
 * channels [0, A) objectness logits, [A, 5A) box deltas), gy = its gradient map (dense NHWC [B, H, W, Cy], written in full
 * by the backward), first = anchors of the levels before, pix0 = pixels of the levels before.  Images [0, B/2) are view 1,
 * image i pairs with i + B/2.  labels / label_weights [B, At], bbox_targets / bbox_weights [B, At, 4] as produced by
 * oadg_anchor_targets.  out4 = {loss_cls, its BCE part, its JSD part, loss_bbox}: sums over all levels (what
 * BaseDetector._parse_losses makes of the per-level lists, base.py:234-277). */
typedef struct oadg_rpn_loss_level {
    const void* y;
    void* gy;
    long sN, sC, sH, sW;
    int H, W, Cy;
    long first, pix0;
} oadg_rpn_loss_level;
size_t oadg_rpn_loss_workspace_bytes(void);
int oadg_rpn_loss_fwd(const oadg_rpn_loss_level* levels, int n_levels, int B, int A, long At, int dtype,
                      const int64_t* labels, const float* label_weights, const float* bbox_targets,
                      const float* bbox_weights, float avg_factor, float w_cls, float lambda_jsd, float w_box,
                      void* workspace, size_t workspace_bytes, float* out4, void* stream);
int oadg_rpn_loss_bwd(const oadg_rpn_loss_level* levels, int n_levels, int B, int A, long At, int dtype,
                      const int64_t* labels, const float* label_weights, const float* bbox_targets,
                      const float* bbox_weights, float avg_factor, float w_cls, float lambda_jsd, float w_box,
                      const float* grad_cls, const float* grad_box, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RPN proposals after the per-level top-k (mmdet/models/dense_heads/rpn_head.py:103-235, batched over the images;
 * delta2bbox of mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:184-..., mmcv batched_nms's class offset): csrc/proposals.hip.
 * A level descriptor: the level's bbox_pred (fp32 = dtype 0 / bf16 = 1, logical [N, 4A, H, W] with element strides
 * sN, sC, sH, sW - the fused RPN head writes one 128-channel NHWC tensor), its anchors [H*W*A, 4] fp32, and for every
 * image the k kept scores [I, k] in stable descending order with their positions index [I, k] (int64, position
 * i = (h*W + w)*A + a; NULL: position = rank).  first = sum of k over the levels before it.
 *   oadg_rpn_decode: props [I, M, 4] (decoded, clipped to lim [I][2] = (W_i, H_i) when clip), scores [I, M], valid [I, M]
 *                    (w > min_size and h > min_size; min_size < 0: all valid), M = sum of k.
 *   oadg_rpn_order:  the stable descending order of key = valid ? score : -1 as a merge of the sorted levels: order [I, M]
 *                    (int32, candidate index per position), boxes_sorted [I, M, 4] = props + level * (max coordinate of the
 *                    valid boxes + 1) in that order (the input of oadg_nms_batched), counts [I] = valid candidates,
 *                    max_coord [I].
 *   oadg_rpn_gather: dets [I, P, 5]: row j < keep_cnt[i] = (props, score) of candidate order[keep[i][j]], other rows
 *                    (0, 0, 0, 0, -1). */
#define OADG_RPN_MAX_LEVELS 8
typedef struct oadg_rpn_level {
    const void* deltas;
    long sN, sC, sH, sW;
    const float* anchors;
    const float* scores;
    const long long* index;
    int H, W, A, k, dtype, first;
} oadg_rpn_level;
int oadg_rpn_decode(const oadg_rpn_level* levels, int n_levels, int n_img, const float* means4, const float* stds4,
                    float max_ratio, const float* lim, int clip, float min_size, float* props, float* scores,
                    unsigned char* valid, void* stream);
int oadg_rpn_order(const oadg_rpn_level* levels, int n_levels, int n_img, const float* props, const float* scores,
                   const unsigned char* valid, float* boxes_sorted, int* order, int* counts, float* max_coord,
                   void* stream);
int oadg_rpn_gather(int n_img, int M, int P, const float* props, const float* scores, const int* order, const int* keep,
                    const int* keep_cnt, float* dets, void* stream);
/* The per-level top-k in front of it (rpn_head.py:146-155: scores.sort(descending, stable)[:nms_pre] per image and level)
 * for all (image, level) rows in six launches: sigmoid scores of cls[l] (dtype 0 fp32 / 1 bf16, logical [N, A, H, W] with
 * element strides strides[4 l ..] = sN, sC, sH, sW, dims[3 l ..] = H, W, A) in (h, w, a) order, a three-pass radix SELECT of
 * the k-th largest score (ties resolved in index order, as the stable sort does), ordered compaction, one LDS sort of the k
 * candidates per row.  out_scores[l] [n_img, k_l] fp32 and out_index[l] [n_img, k_l] int64, k_l = min(nms_pre, A H W).
 * n_img * n_levels <= 48 rows, k_l <= 16384. */
size_t oadg_rpn_topk_workspace_bytes(const int* level_n, int n_levels, int n_img, int nms_pre);
int oadg_rpn_topk(const void* const* cls, const long* strides, const int* dims, int dtype, int n_levels, int n_img,
                  int nms_pre, float* const* out_scores, long long* const* out_index, void* workspace,
                  size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Greedy NMS, batched over images
 *   replaces mmcv.ops.batched_nms -> nms as called at mmdet/models/dense_heads/rpn_head.py:231
 * boxes [n_images, Mmax, 4] fp32 sorted by descending score per image, class offsets already added;
 * counts [n_images] int32 valid boxes; keep [n_images, Mmax] int32; keep_cnt [n_images] int32.
 */
size_t oadg_nms_workspace_bytes(int n_images, int Mmax);
int oadg_nms_batched(const float* boxes, const int* counts, int n_images, int Mmax, float iou_thr,
                     int max_keep, void* workspace, size_t workspace_bytes, int* keep, int* keep_cnt,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * OA-Mix (object-aware augmentation) device ops.  The host class `OAMix` (oa-dg_amd/pipelines/oa_mix.py,
 * registered under the reference's PIPELINES name) replays the reference's np.random stream and enqueues these.
 * Images are uint8 [H, W, 3] (the reference's BGR bytes); masks are never H x W x 3 floats: a blurred box mask
 * is the outer product My[y] * Mx[x] of two float32 profiles.
 *   _box_profiles  replaces OAMix._get_mask                 mmdet/datasets/pipelines/oa_mix.py:74-93
 *                  qbox [n,4] = box // spatial_ratio as int32 (x1,y1,x2,y2); sigma [n,2] = (sx, sy), <= 0: no blur
 *   _fg_union      max over the fg masks + its uint8(x*255) image   bbox_augmentation.py:257-264
 *   _saliency      replaces cv2.saliency.StaticSaliencySpectralResidual + mean   oa_mix.py:104-111
 *                  boxes [n,4] int32 crop corners; score -1 when a side < min_side
 *   _hist/_luts    per-channel histogram; luts[0] = PIL.ImageOps.autocontrast, luts[1] = .equalize   augmix.py:64-69
 *   _bbox_step     one gt box of a bboxes_only_* op: warp the whole image (minv_host = inverted 2x3, row-major),
 *                  blend it in through the box's blurred mask inside rect (rx0,ry0,rw,rh)   bbox_augmentation.py:31-71
 *   _compose       one depth step of the multi-level chain: up to 2 disjoint random boxes + the outside, each with
 *                  its own op; optionally acc (=|+=) acc_w * float(dst)   oa_mix.py:221-236
 *   _final         object_aware_mixing + clip + uint8 + (Normalize, Pad) -> out_u8 and/or NHWC out_norm
 *                  (out_dtype 0 fp32 / 1 bf16, padded to Hp x Wp with zeros)   oa_mix.py:281-309, transforms.py:618-629,699-701
 *   _normalize     Normalize + Pad of an un-augmented view
 */
enum {
    OADG_OP_COPY = 0,
    OADG_OP_LUT_AUTOCONTRAST = 1,
    OADG_OP_LUT_EQUALIZE = 2,
    OADG_OP_POSTERIZE = 3,   /* param = bit mask */
    OADG_OP_SOLARIZE = 4,    /* param = threshold */
    OADG_OP_IMAGE = 5,       /* image = uint8 [H,W,3] holding the op's full result (bboxes_only_* ops) */
    OADG_OP_BG_WARP = 6,     /* bg_only_* : minv = inverted affine */
    OADG_OP_WARP_NEG = 7,    /* 'invert' of augmix.all: -warpAffine (uint8 wrap) */
    /* PIL.ImageEnhance of augmix.all (augmix.py:192-212): minv[0] = factor; CONTRAST: image -> int64 luma sum */
    OADG_OP_ENH_BRIGHTNESS = 8,
    OADG_OP_ENH_COLOR = 9,
    OADG_OP_ENH_CONTRAST = 10,
    OADG_OP_ENH_SHARPNESS = 11
};
typedef struct {
    int kind;
    int param;
    const void* image;
    double minv[6];
} oadg_region_op;
typedef struct {
    int fg_index;   /* >= 0: blurred fg mask number; -1: sharp rectangle */
    int rect[4];    /* x1,y1,x2,y2 (exclusive) when fg_index < 0 */
    float m_oa;
} oadg_mix_target;

int oadg_oamix_box_profiles(const int* qbox, const double* sigma, int n, int H, int W, int ratio, float* My,
                            float* Mx, void* stream);
int oadg_oamix_fg_union(const float* My, const float* Mx, int n, int H, int W, float* union_f,
                        uint8_t* union_u8, void* stream);
/* the same union from the masks' support rects (device int32 [n][4]: x0, y0, w, h; w <= 0 = empty mask): work
 * proportional to the rect areas instead of n * H * W (a 4096-box image: 4.3 ms -> < 0.1 ms).  Byte-identical outputs. */
int oadg_oamix_fg_union_rects(const float* My, const float* Mx, const int* rects_dev, int n, int H, int W, float* union_f,
                              uint8_t* union_u8, void* stream);
/* workspace: oadg_oamix_saliency_workspace_bytes(n) bytes (the 64x64 maps and integer totals between the launches) */
size_t oadg_oamix_saliency_workspace_bytes(int n);
int oadg_oamix_saliency(const uint8_t* img, int H, int W, const int* boxes, int n, int min_side,
                        double* scores, void* workspace, size_t workspace_bytes, void* stream);
/* the boxes of several images of ONE uint8 [N,H,W,3] batch in one call (round 6): image i at imgs + i * img_stride bytes,
 * box b in image box_img[b] (device int32 [n]); the scores of n / N separate oadg_oamix_saliency calls */
int oadg_oamix_saliency_batch(const uint8_t* imgs, long long img_stride, const int* box_img, int H, int W, const int* boxes,
                              int n, int min_side, double* scores, void* workspace, size_t workspace_bytes, void* stream);
int oadg_oamix_hist(const uint8_t* img, long npix, int* hist, void* stream);
int oadg_oamix_luts(const int* hist, uint8_t* luts, void* stream);
int oadg_oamix_gray_sum(const uint8_t* img, long npix, long long* sum, void* stream);
int oadg_oamix_bbox_step(uint8_t* img, int H, int W, const double* minv_host, int rx0, int ry0, int rw, int rh,
                         const float* My_row, const float* Mx_row, uint8_t* scratch, void* stream);
/* bboxes_only_* for ALL gt boxes of an image in a handful of launches (bbox_augmentation.py:74-88 applies the boxes
 * one after the other; only boxes whose written rect meets another's read footprint are ordered).  steps_dev: the
 * boxes' steps sorted by dependency level; tile_prefix_*[i] = number of 1024-pixel tiles before step i (n + 1 entries,
 * device and host copies); level_first_host[l] .. level_first_host[l + 1] = steps of level l.  Two launches per
 * level (blend into scratch, copy back); rect = x0, y0, width, height; scratch_off = byte offset of the step's rect
 * inside `scratch`, a multiple of 4 (rects of one level are disjoint, so H*W*3 + 4*n + 16 bytes suffice). */
typedef struct {
    double minv[6];        /* inverted affine (cv2.warpAffine) */
    int rect[4];
    int row;               /* profile row: My + row * H, Mx + row * W */
    int pad_;
    long long scratch_off;
} oadg_bbox_step;
/* HOST function (no device work): dependency level of each of the n steps in list order - rects [n][4] = x0, y0, w, h
 * (pixels written), minvs [n][6]; level[j] > level[i] whenever i < j and one writes what the other reads or writes. */
int oadg_oamix_bbox_levels(const int* rects, const double* minvs, int n, int H, int W, int* level);
/* HOST function: the whole host side of one bboxes_only_* op for all gt boxes (bbox_augmentation.py:31-88 with the leaf
 * matrices of augmix.py:83-188): kind 0 rotate / 1 shear_x / 2 shear_y / 3 translate_x / 4 translate_y; ib [n][4] int64
 * box corners; support [n][4] = x0, y0, w, h of every box's mask support (w or h <= 0: empty); draws [2 m] = the (level,
 * sign) uniforms of the m boxes that draw (integer width and height >= 1), in box order.  Writes the level-major step
 * table + the (n_live + 1)-entry tile prefix into `staging` (>= oadg_oamix_bbox_plan_bytes(n), the layout
 * oadg_oamix_bbox_chain consumes after one copy to the device), level_first [n_levels + 1] (host, >= n + 2 entries),
 * out = {n_live, n_levels, tiles}, *area_sum = sum of rect areas.  Same numpy operation order as the reference's
 * expressions: bit-identical matrices. */
size_t oadg_oamix_bbox_plan_bytes(int n);
int oadg_oamix_bbox_plan(int kind, double severity, const long long* ib, const int* support, int n, const double* draws,
                         int n_draws, int H, int W, void* staging, size_t staging_bytes, int* level_first, int* out,
                         long long* area_sum);
int oadg_oamix_bbox_chain(uint8_t* img, int H, int W, const oadg_bbox_step* steps_dev, const int* tile_prefix_dev,
                          const int* level_first_host, int n_levels, const int* tile_prefix_host, const float* My,
                          const float* Mx, uint8_t* scratch, void* stream);
/* the chains of several images advanced in lockstep: level l of every chain in ONE launch pair (the images of a batch
 * are independent; launches per batch = 2 x the deepest chain).  One descriptor per image with the operands of
 * oadg_oamix_bbox_chain; images and scratch buffers pairwise distinct.  Byte-identical to n calls of
 * oadg_oamix_bbox_chain. */
typedef struct {
    uint8_t* img;
    const oadg_bbox_step* steps_dev;
    const int* tile_prefix_dev;
    const int* level_first_host;
    const int* tile_prefix_host;
    const float* My;
    const float* Mx;
    uint8_t* scratch;
    int H, W, n_levels, pad_;
} oadg_bbox_chain;
int oadg_oamix_bbox_chain_multi(const oadg_bbox_chain* chains_host, int n, void* stream);
int oadg_oamix_compose(const uint8_t* src, uint8_t* dst, int H, int W, const oadg_region_op* ops_host,
                       const int* rects_host, int n_rects, const uint8_t* luts, const float* union_f,
                       const uint8_t* union_u8, float* acc, float acc_w, int acc_mode, void* stream);
int oadg_oamix_final(const uint8_t* img, const float* acc, int H, int W, const oadg_mix_target* targets,
                     int n_targets, const float* My, const float* Mx, double m_beta, const float* mean_host,
                     const float* stdinv_host, int to_rgb, uint8_t* out_u8, void* out_norm, int out_dtype,
                     int Hp, int Wp, void* stream);
/* oadg_oamix_final for images with many targets: targets binned into 32 x 32 pixel tiles (in target order), a pixel only
 * visits its tile's list.  fg_rects_dev as for oadg_oamix_fg_union_rects; 1 <= n_targets <= 65535.  Byte-identical. */
size_t oadg_oamix_final_tiles_workspace_bytes(int H, int W, int n_targets);
int oadg_oamix_final_tiles(const uint8_t* img, const float* acc, int H, int W, const oadg_mix_target* targets,
                           int n_targets, const int* fg_rects_dev, const float* My, const float* Mx, double m_beta,
                           const float* mean_host, const float* stdinv_host, int to_rgb, uint8_t* out_u8, void* out_norm,
                           int out_dtype, int Hp, int Wp, void* workspace, size_t workspace_bytes, void* stream);
int oadg_oamix_normalize(const uint8_t* img, int H, int W, const float* mean_host, const float* stdinv_host,
                         int to_rgb, void* out, int out_dtype, int Hp, int Wp, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input files (HOST functions, no device work; csrc/png_decode.hip)
 *   oadg_png_decode_bgr  LoadImageFromFile  mmdet/datasets/pipelines/loading.py:33-78 (mmcv.imfrombytes -> cv2.imdecode:
 *                        colour, BGR byte order)
 * PNG file -> uint8 [H][W][3] in B, G, R order at `out` (any host memory: the data set hands in a slice of a pinned
 * batch buffer) - zlib inflate + the five PNG row filters, one call without the interpreter lock.  8-bit grey / RGB / RGBA,
 * non-interlaced; OADG_EUNSUPPORTED for any other PNG variant (the caller decodes those with PIL), OADG_ESIZE when the
 * file's extent is not H x W, OADG_EIO when it cannot be read.  oadg_png_size: the extent from the header. */
int oadg_png_size(const char* path, int* height, int* width);
int oadg_png_decode_bgr(const char* path, uint8_t* out, int H, int W);

/* ------------------------------------------------------------------------------------------------
 * Robustness-benchmark corruptions (HOST functions, no device work; csrc/corrupt_host.hip) - the two sequential
 * per-pixel loops of `Corrupt`, mmdet/datasets/pipelines/transforms.py:1277-1317 -> imagecorruptions.corrupt
 * (third party, v1.1.2), tools/analysis_tools/test_robustness.py:222-235
 *   oadg_glass_shuffle_u8  glass_blur's local shuffle: `iters` sweeps over h = H - delta ... delta + 1, w = W - delta ...
 *                          delta + 1 (both descending), pixel (h, w) swapped with (h + dy, w + dx); dxdy = the draws in
 *                          loop order, int32 [iters][H - 2 delta][W - 2 delta][2] = (dx, dy), each in [-delta, delta)
 *                          (OADG_EARG otherwise).  In place on uint8 [H][W][C], C <= 4.
 *   oadg_chamfer_l2_5x5    cv2.distanceTransform(src, DIST_L2, 5) of spatter: distance to the nearest ZERO pixel,
 *                          two-pass chamfer with weights 1 / 1.4 / 2.1969 in 16-bit fixed point, float32 [H][W] out. */
int oadg_glass_shuffle_u8(uint8_t* img, int H, int W, int C, int delta, int iters, const int32_t* dxdy);
int oadg_chamfer_l2_5x5(const uint8_t* src, int H, int W, float* dist);

/* ------------------------------------------------------------------------------------------------
 * Geometric pipeline steps in front of OA-Mix on uint8 HWC images
 *   oadg_resize_bilinear_u8  Resize._resize_img  mmdet/datasets/pipelines/transforms.py:210-239
 *                            (mmcv.imrescale / imresize -> cv2.resize INTER_LINEAR, 8-bit fixed-point path)
 *   oadg_flip_u8             RandomFlip.__call__ transforms.py:452-457 (mmcv.imflip); direction 1 horizontal,
 *                            2 vertical, 3 diagonal; src != dst
 */
int oadg_resize_bilinear_u8(const uint8_t* src, int H, int W, int C, uint8_t* dst, int Hn, int Wn, void* stream);
int oadg_flip_u8(const uint8_t* src, int H, int W, int C, uint8_t* dst, int direction, void* stream);

/* ------------------------------------------------------------------------------------------------
 * NHWC bf16 implicit-GEMM convolution on the MFMA cores, fused bias / residual / ReLU epilogue
 *   replaces the cuDNN convolutions of  mmdet/models/necks/fpn.py:112-129, dense_heads/rpn_head.py:54-68,
 *   backbones/resnet.py:166-205 (with the eval-mode BN of :648-657 folded into w / bias)
 * x [N,H,W,C] bf16, w [K,R,S,C] bf16, bias [K] fp32 or NULL, residual [N,Ho,Wo,K] bf16 or NULL, y [N,Ho,Wo,K]
 * bf16, zeros16: >= 16 zero bytes.  Requires C % 64 == 0 and K % 128 == 0 (other shapes: argument error).
 * The stride-1 data gradient is the same call on the flipped/transposed weight with pad' = dil*(R-1) - pad.
 */
int oadg_conv2d_nhwc_bf16(const void* x, const void* w, const float* bias, const void* residual, void* y,
                          const void* zeros16, int N, int H, int W, int C, int K, int R, int S, int stride,
                          int pad, int dil, int relu, void* stream);

/* weight gradient of the same convolution: dw [K,R,S,C] fp32 = sum over output pixels of dy (x) x@tap
 * (transposing LDS reads + split-K partials in `workspace`, summed in fixed order).  C % 128 == 0, K % 128 == 0. */
/* same contract; `variant` picks the kernel: 0 automatic (what oadg_conv2d_nhwc_bf16 does), 1 = 128x128x64 tile
 * (4 waves, two 32 KiB LDS stages, 2 workgroups per CU), 2 = 256x256x64 tile (8 waves, 128 KiB LDS,
 * phase-pipelined; needs K % 256 == 0), 3 = 128-tile with ONE LDS stage at 4 workgroups per CU, 4 = the streaming
 * pointwise kernel (1x1 / stride 1 / pad 0, C in {64, 128, 256}, K % 256 == 0, N*H*W a multiple of 32 and large
 * enough: weights stationary in registers, pixels through an LDS-DMA ring, residual / mask bits prefetched one
 * sub-tile ahead; no bf16 mask operand, not mask_bits together with relu_bits_out; OADG_EARG when not eligible) */
int oadg_conv2d_nhwc_bf16_variant(const void* x, const void* w, const float* bias, const void* residual, void* y,
                                  const void* zeros16, int N, int H, int W, int C, int K, int R, int S, int stride,
                                  int pad, int dil, int relu, int variant, void* stream);
/* data-gradient form with the producer's epilogue backward fused in: y = (conv(x, w) [+ residual]) * (mask > 0)
 * (mask [N,Ho,Wo,K] bf16 = the ReLU output the gradient flows into; torch threshold_backward) and
 * colsum_part [oadg_conv2d_pixel_tiles][K] = per-pixel-tile column sums of the stored y = partial bias / BN-shift
 * gradients of the producer (sum them with oadg_colsum_reduce).  mask / colsum_part may be NULL.
 * mask_bits (may be NULL, instead of mask): the same ReLU mask as ONE BIT per element - [rows][K / 8] bytes, bit e of
 * byte j = channel 8 j + e - i.e. 1/16 of the bf16 mask's bytes on these HBM-bound launches;  relu_bits_out (may be
 * NULL): this launch additionally stores (y > 0) of what it writes in that format (the forward launch of the tensor
 * whose gradient a later launch masks).  Both need K % 8 == 0.
 * relu: bit 0 = ReLU; bit 1 (round 4) = `residual` is a HALF-resolution map [N][H/2][W/2][K] added through the nearest 2x
 * upsampling - pixel (n, y, x) adds row (n, y / 2, x / 2): `laterals[i - 1] += F.interpolate(laterals[i], ...)` of
 * necks/fpn.py:166-175 in the lateral convolution's epilogue.  1x1 / stride 1 / pad 0 problems the streaming kernel takes
 * (oadg_conv2d_auto_variant == 4) with even H, W, no mask operands; OADG_EARG otherwise. */
int oadg_conv2d_nhwc_bf16_ex(const void* x, const void* w, const float* bias, const void* residual, void* y,
                             const void* zeros16, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                             int dil, int relu, int variant, const void* mask, float* colsum_part,
                             const void* mask_bits, void* relu_bits_out, void* stream);
long oadg_conv2d_pixel_tiles(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                             int variant);
int oadg_colsum_reduce(const float* part, long rows, int K, float* out, void* stream);
/* several column-sum reductions in ONE launch (the per-layer launches of a backward pass, deferred to its end):
 * jobs_dev [n] on the DEVICE in ascending first_block order, job i owns blocks [first_block_i, first_block_i +
 * (K_i + 15) / 16); out[k] = sum over rows of part[row][k] in oadg_colsum_reduce's order (bit-identical).  dgamma != NULL:
 * the BN scale gradient of that layer was left as a raw dot product by oadg_prep_conv_weights_bwd(_parts) (w_krsc bit
 * 1) and is finished here: dgamma[k] = (dgamma[k] - out[k] * mean[k]) * rsqrt(var[k] + eps) - the expression those
 * functions evaluate when they are given the reduced bias gradient. */
typedef struct oadg_colsum_job {
    const float* part;         /* [rows][K] */
    float* out;                /* [K] */
    float* dgamma;             /* [K] or NULL */
    const float* mean;
    const float* var;
    int rows, K, first_block;
    float eps;
} oadg_colsum_job;
int oadg_colsum_reduce_multi(const oadg_colsum_job* jobs_dev, int n, int total_blocks, void* stream);
/* the variant (2, 3 or 4) variant 0 resolves to for a problem BY ITS GEOMETRY; 0 = shape not covered.  A launch with a
 * bf16 mask operand (or mask_bits together with relu_bits_out) runs 3 where this says 4: callers that need
 * colsum_part pass the variant they resolved explicitly, to oadg_conv2d_pixel_tiles (variant 4: one row per pixel
 * range) and to the launch. */
int oadg_conv2d_auto_variant(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil);
size_t oadg_conv2d_wgrad_workspace_bytes(int N, int Ho, int Wo, int C, int K, int R, int S);
/* the kernel a weight-gradient problem resolves to: 256 = the 256-tile phase pipeline, 2 / 1 = the 128-tile kernel with
 * two / one LDS stages, 0 = shape not covered */
int oadg_conv2d_wgrad_variant(int N, int Ho, int Wo, int C, int K, int R, int S);
int oadg_conv2d_wgrad_nhwc_bf16(const void* x, const void* dy, float* dw, const void* zeros16, void* workspace,
                                size_t workspace_bytes, int N, int H, int W, int C, int K, int R, int S,
                                int stride, int pad, int dil, void* stream);

/* the weight gradient as split partials + their consumer: oadg_conv2d_wgrad_parts_nhwc_bf16 leaves
 * workspace = [*splits][K][R*S][C] fp32; oadg_prep_conv_weights_bwd_parts sums the splits in fp32 and applies the
 * backward of oadg_prep_conv_weights (dw [K,C,R,S], dgamma) in the same launch - no reduction kernel, no bf16 round
 * trip of the weight gradient.  C*R*S <= 3000 (five fp32 copies of a filter in LDS).  w_krsc: bit 0 = the weight lies
 * [K][R][S][C] in memory (channels_last parameter), bit 1 = gbias is not available yet: dgamma receives the raw dot
 * product and oadg_colsum_reduce_multi finishes it (also for oadg_prep_conv_weights_bwd). */
int oadg_conv2d_wgrad_parts_nhwc_bf16(const void* x, const void* dy, const void* zeros16, void* workspace,
                                      size_t workspace_bytes, int N, int H, int W, int C, int K, int R, int S,
                                      int stride, int pad, int dil, int* splits, void* stream);
int oadg_prep_conv_weights_bwd_parts(const float* part, int splits, const float* gbias, const float* w,
                                     const float* scale, const float* mean, const float* var, float eps, int K, int C,
                                     int R, int S, float* dw, float* dgamma, int w_krsc, void* stream);

/* The weight gradients of SEVERAL layers in one launch (round 4): the small maps of the backbone / neck cannot fill the
 * chip with long workgroups one layer at a time, so a trainer defers them and issues groups.  Host: fill x, dy, N, H, W,
 * C, K, R, S, stride, pad, dil of every job (K % 256 == 0, C % 256 == 0), call oadg_conv2d_wgrad_multi_plan - it fills
 * the remaining fields and returns the length of the workgroup list (target_blocks = the workgroups of one round, i.e. the
 * compute units; the plan takes a list of 1 - 3 rounds, whichever its simulation of the dispatch finds shortest, or one
 * workgroup per weight tile when those alone are more; negative = -OADG_E*) -, point part at splits * K * R * S * C floats per job, copy the table to the device, launch.
 * part has the layout of oadg_conv2d_wgrad_parts_nhwc_bf16's workspace; oadg_prep_conv_weights_bwd_parts_multi is the
 * matching consumer (one launch for the group: splits summed in fp32 in a fixed order, BN-fold chain rule, layout
 * change; w_krsc bits as in oadg_prep_conv_weights_bwd_parts; first_block = sum of K over the jobs before). */
typedef struct oadg_wgrad_job {
    const void* x;             /* bf16 [N,H,W,C] */
    const void* dy;            /* bf16 [N,Ho,Wo,K] */
    void* part;                /* fp32 [splits][K][R*S][C] */
    long P;                    /* plan: N*Ho*Wo */
    int N, H, W, C, K, R, S, stride, pad, dil;
    int Ho, Wo, splits, chunks_per_split, first_block, blocks;      /* plan */
    int strip_rows, pad_;      /* plan: > 0 = a split is this many whole output rows, swept in 64-column strips */
} oadg_wgrad_job;
/* xcd_first (host, 9 ints, may be NULL): the plan's eight contiguous slices of the list, equal in WORK - XCD x runs the
 * entries [xcd_first[x], xcd_first[x + 1]); the launch takes the same array (NULL: slices of equal length) */
long oadg_conv2d_wgrad_multi_plan(oadg_wgrad_job* jobs_host, int n, int target_blocks, int* xcd_first);
int oadg_conv2d_wgrad_multi(const oadg_wgrad_job* jobs_dev, int n, int total_blocks, const int* xcd_first, const void* zeros16,
                            void* stream);
typedef struct oadg_prep_bwd_job {
    const float *part, *gbias, *w, *scale, *mean, *var;
    float *dw, *dgamma;
    float eps;
    int splits, K, C, R, S, w_krsc, first_block;
} oadg_prep_bwd_job;
int oadg_prep_conv_weights_bwd_parts_multi(const oadg_prep_bwd_job* jobs_dev, int n, int total_blocks, int max_crs,
                                           void* stream);

/* one parity class of a strided data gradient: a stride-1 convolution over x (= dy) whose out_h x out_w output pixels
 * per image are stored on the strided grid row = ((n*OH + ho*osh + oph)*OW + wo*osw + opw) of y [N,OH,OW,K]; residual and
 * mask are read at the same rows; input positions past the extent of x read zeros.  Replaces the stride-2 branch of
 * aten.convolution_backward (cuDNN/MIOpen data gradient) for Bottleneck.conv2 / downsample (resnet.py:97-302). */
int oadg_conv2d_nhwc_bf16_scatter(const void* x, const void* w, const float* bias, const void* residual, void* y,
                                  const void* zeros16, int N, int H, int W, int C, int K, int R, int S, int pad, int dil,
                                  int relu, int out_h, int out_w, int OH, int OW, int osh, int osw, int oph, int opw,
                                  const void* mask, float* colsum_part, const void* mask_bits, void* stream);
/* dx of a stride-2 convolution (3x3 / pad 1 or 1x1 / pad 0) in ONE launch: the parity classes of
 * oadg_conv2d_nhwc_bf16_scatter together, equal-sized classes interleaved per XCD so that dy is fetched from HBM once
 *   serves the data gradient of Bottleneck.conv2 / downsample of a stage's first block
 *          mmdet/models/backbones/resnet.py:166-186 (stride on the 3x3, pytorch style) through autograd
 * dy [N,Ho,Wo,K] bf16; wt = oadg_prep_conv_weights wt_mode 2 (class blocks [C][taps][K]); dx [N,H,W,C] bf16 (1x1: only
 * the even rows / columns are written); residual / mask / mask_bits indexed like dx (residual may alias dx: in-place
 * accumulate); colsum_part: per class ceil(N*ha*wa/128) rows, classes back to back.  Bit-identical to the class launches. */
int oadg_conv2d_dgrad_s2_nhwc_bf16(const void* dy, const void* wt, const void* residual, void* dx, const void* zeros16,
                                   int N, int Ho, int Wo, int K, int C, int R, int H, int W, const void* mask,
                                   float* colsum_part, const void* mask_bits, void* stream);

/* per-layer weight preparation for the kernels above (one launch): optional eval-mode BatchNorm fold
 * (resnet.py:648-657: scale = gamma / sqrt(var + eps), bias = beta - mean * scale; gamma == NULL: plain cast with
 * bias_in), fp32 [K,C,R,S] -> bf16 wf [K,R,S,C] and (optional) wt [C,R,S,K] flipped for the data gradient.
 * _bwd: gwf bf16 [K,R,S,C], gbias fp32 [K] -> dw fp32 [K,C,R,S], dgamma fp32 [K] (d beta = gbias).
 * w_krsc != 0: w (and dw) are stored channels-last, [K][R][S][C] in memory (a torch.channels_last parameter), else
 * [K][C][R][S].  wt_mode 2 (3x3 / pad 1 and 1x1 stride-2 layers): wt receives, instead of the flipped copy, the weights of
 * the four parity-class convolutions of the stride-2 data gradient, blocks [C][taps][K] in the order (0,0) (0,1) (1,0)
 * (1,1) with 1, 2, 2, 4 taps (1x1: one block) - see oadg_conv2d_nhwc_bf16_scatter. */
int oadg_prep_conv_weights(const float* w, const float* gamma, const float* beta, const float* mean,
                           const float* var, float eps, const float* bias_in, int K, int C, int R, int S, void* wf,
                           void* wt, float* bias, float* scale, int w_krsc, int wt_mode, void* stream);
/* the same for many layers in ONE launch: a device table of descriptors (the arguments of oadg_prep_conv_weights per
 * layer; first_block = sum of oadg_prep_conv_weights_multi_blocks of the layers before it, ascending), total_blocks =
 * their sum.  Weights change only in
 * optimizer.step(), so a trainer re-prepares every layer once after it instead of per layer inside the forward pass. */
typedef struct oadg_prep_desc {
    const float *w, *gamma, *beta, *mean, *var, *bias_in;
    void *wf, *wt;
    float *bias, *scale;
    float eps;
    int K, C, R, S, w_krsc, wt_mode, first_block;
} oadg_prep_desc;
int oadg_prep_conv_weights_multi_blocks(int K, int C, int R, int S);   /* workgroups of one layer (its first_block step) */
int oadg_prep_conv_weights_multi(const oadg_prep_desc* descs, int n_layers, int total_blocks, void* stream);
int oadg_prep_conv_weights_bwd(const void* gwf, const float* gbias, const float* w, const float* scale,
                               const float* mean, const float* var, float eps, int K, int C, int R, int S, float* dw,
                               float* dgamma, int w_krsc, void* stream);

/* 1x1 convolutions with at most 16 output channels (round 4): the RPN head - rpn_cls + rpn_reg, 3 + 12 channels per
 * pixel (dense_heads/rpn_head.py:54-68) - on 16-channel-wide maps (32-byte pixel rows) instead of a 128-channel tile.
 * x bf16 [M][C] (C = 128 or 256), w16 bf16 [16][C] (rows past the live output channels zero), bias16 fp32 [16] or NULL,
 * y bf16 [M][16].  _dgrad: dx[m][c] = sum_k dy[m][k] wt[c][k] (wt bf16 [C][16]) with the data-gradient epilogue of
 * oadg_conv2d_nhwc_bf16_ex: mask_bits (one bit per element of dx, [M][C / 8] bytes) or NULL, colsum_part
 * [oadg_conv1x1_n16_dgrad_rows(M)][C] partial column sums of the stored dx or NULL (reduce: oadg_colsum_reduce).
 * _wgrad: part [rows][16][C] fp32 split partials of dW[k][c] = sum_m dy[m][k] x[m][c] and bias_part [rows][16] of
 * sum_m dy[m][k], rows = oadg_conv1x1_n16_wgrad_rows(M) (sum the rows in order: deterministic). */
int oadg_conv1x1_n16_fwd(const void* x, const void* w16, const float* bias16, void* y, long M, int C, void* stream);
long oadg_conv1x1_n16_dgrad_rows(long M);
int oadg_conv1x1_n16_dgrad(const void* dy, const void* wt, void* dx, const void* mask_bits, float* colsum_part, long M, int C,
                           void* stream);
long oadg_conv1x1_n16_wgrad_splits(long M);
long oadg_conv1x1_n16_wgrad_rows(long M);
int oadg_conv1x1_n16_wgrad(const void* x, const void* dy, float* part, float* bias_part, const void* zeros16, long M, int C,
                           void* stream);

/* ResNet stem convolution (backbones/resnet.py:585-596 conv1: 7x7, stride 2, padding 3, 3 -> 64 channels) on the matrix
 * cores: x bf16 NHWC [N,H,W,3], wp = prepared weights bf16 [64][7][8][4] (k = filter row, 8 input pixels starting
 * one left of the filter - pixel 0 zero -, 4 channels - channel 3 zero), y = round_bf16(conv) NHWC [N,Ho,Wo,64] without
 * bias, Ho = (H-1)/2+1, Wo = (W-1)/2+1.  Replaces the cuDNN/MIOpen call of the reference for this layer. */
int oadg_stem_conv7x7s2_nhwc_bf16(const void* x, const void* wp, void* y, int N, int H, int W, void* stream);

/* ResNet stem tail in one pass: out [N,Ho,Wo,C] = max_pool2d(relu(x + bias), kernel 3, stride 2, padding 1) for NHWC bf16
 * x [N,H,W,C] (resnet.py:631-637 with the eval-mode BN folded into conv1; bias = BN shift, fp32 [C] or NULL), bit-identical
 * to the element-wise chain add (bf16) -> relu -> max_pool2d.  Ho = (H - 1) / 2 + 1.  C % 8 == 0. */
int oadg_bias_relu_maxpool_nhwc_bf16(const void* x, const float* bias, void* out, int N, int H, int W, int C,
                                     void* stream);

/* fused backward of the conv epilogue y = relu(conv + bias [+ residual]) (Bottleneck.forward resnet.py:285-300,
 * RPNHead.forward_single rpn_head.py:62 `F.relu(x, inplace=True)`, the bias/BN-shift gradient of every conv):
 *   g = dy * (y > 0) as bf16 (y == NULL: no mask), dbias[k] = sum over the M pixels of g[., k]
 * dy: bf16 or fp32 (dy_is_f32) [M,K] rows = NHWC pixels; g may be NULL when only dbias is wanted from a bf16 dy.
 * K % 8 == 0.  Deterministic (fixed-order partials in `workspace`). */
size_t oadg_relu_bias_bwd_workspace_bytes(long M, int K);
int oadg_relu_bias_bwd(const void* dy, int dy_is_f32, const void* y, void* g, float* dbias, void* workspace,
                       size_t workspace_bytes, long M, int K, void* stream);

/* FPN top-down step  laterals[i-1] += F.interpolate(laterals[i], size=..., mode='nearest')  (necks/fpn.py:166-175)
 * in one pass, bf16 NHWC: out[n,h,w,:] = lat[n,h,w,:] + top[n, src(h), src(w), :] with ATen's nearest index
 * src(d) = min(floor(d * in/out), in-1).  _bwd: dtop = the sum of g over the pixels each source feeds
 * (d lat = g itself).  C % 8 == 0. */
int oadg_fpn_topdown_fwd(const void* lat, const void* top, void* out, int N, int H, int W, int Ht, int Wt, int C,
                         void* stream);
int oadg_fpn_topdown_bwd(const void* g, void* dtop, int N, int H, int W, int Ht, int Wt, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MaxIoUAssigner.assign for a batch of images (mmdet/core/bbox/assigners/max_iou_assigner.py:61-213 with
 * BboxOverlaps2D iou_calculators/iou2d_calculator.py:78-...; the callers anchor_head.py:230-232 and
 * standard_roi_head.py:88-95), gt_max_assign_all=True, no ignore boxes.
 * boxes [B][N][4] fp32 (box_stride floats between images, 0 = one shared set), valid [B][N] bytes or NULL
 * (invalid rows get -1 and take no part in any maximum, like the reference's pre-filtered anchors),
 * gts [B][Gmax][4] with gt_counts[B] valid rows, gt_labels [B][Gmax] or NULL.
 * Outputs gt_inds [B][N] int64 (0 negative, -1 ignore, k>0 gt k-1), max_overlaps [B][N], labels [B][N] or NULL,
 * counts [B][2] = #(gt_inds > 0), #(gt_inds == 0) (the sampler's candidate counts).  Bit-identical to the
 * reference's tensor expression in fp32. */
size_t oadg_max_iou_assign_workspace_bytes(int B, int Gmax);
int oadg_max_iou_assign(const float* boxes, long box_stride, const unsigned char* valid, const float* gts,
                        const int* gt_counts, const int64_t* gt_labels, int B, int N, int Gmax, float pos_iou_thr,
                        float neg_iou_lo, float neg_iou_hi, float min_pos_iou, int match_low_quality,
                        void* workspace, size_t workspace_bytes, int64_t* gt_inds, float* max_overlaps,
                        int64_t* labels, int* counts, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RandomSampler selection for a batch (mmdet/core/bbox/samplers/random_sampler.py:32-82, base_sampler.py:38-103).
 * The permutation is drawn on the host (global CPU generator, oadg_host_randperm_prefix); a job asks for the
 * candidates of given RANKS among nonzero(gt_inds > 0) (mode 0) or nonzero(gt_inds == 0) (mode 1) of one image.
 * ranks (ascending, so the output is the reference's sorted index list) live in ranks_dev[rank_off .. +k);
 * all != 0 selects every candidate (k = candidate count, no ranks read).  Output: out[out_off .. +k) int64.
 */
typedef struct oadg_select_job {
    const void* gt_inds;   /* [n] int64, device */
    long n;
    int mode;
    int k;
    int all;
    int rank_off;
    long out_off;
} oadg_select_job;
size_t oadg_sample_select_workspace_bytes(int jobs, long max_n);
int oadg_sample_select(const oadg_select_job* jobs_dev, int jobs, long max_n, int max_k, const int* ranks_dev,
                       int64_t* out, void* workspace, size_t workspace_bytes, void* stream);   /* max_k = largest k */

/* AnchorHead._get_targets_single for all images at once (mmdet/models/dense_heads/anchor_head.py:201-297, the
 * branch with every anchor inside the image, + DeltaXYWHBBoxCoder.encode delta_xywh_bbox_coder.py:119-180).
 * jobs_dev / sel: the job table and output of oadg_sample_select, job 2b = positives and 2b+1 = negatives of
 * image b over the shared anchors [A][4]; gt_inds [B][A] from oadg_max_iou_assign; gts [B][Gmax][4];
 * gt_labels NULL = RPN (positives get label 0).  Outputs labels [B][A] (fill_label elsewhere), label_weights
 * [B][A], bbox_targets / bbox_weights [B][A][4]; means4 / stds4 are HOST pointers.  max_k = largest k of a job. */
int oadg_anchor_targets(const float* anchors, const float* gts, const int64_t* gt_inds, const int64_t* gt_labels,
                        const oadg_select_job* jobs_dev, const int64_t* sel, int B, int A, int Gmax, int max_k,
                        int64_t fill_label, float pos_weight, const float* means4, const float* stds4,
                        int64_t* labels, float* label_weights, float* bbox_targets, float* bbox_weights,
                        void* stream);

/* SGD (momentum, weight decay; no dampening, no nesterov) for a list of fp32 tensors in ONE launch
 *   serves torch.optim.SGD.step as the reference runs it (configs/_base_/schedules/schedule_1x.py optimizer,
 *          mmcv OptimizerHook after mmdet/apis/train.py:150-161), torch/optim/sgd.py _multi_tensor_sgd arithmetic:
 *          g' = fma(wd, p, g); m = g' on a tensor's first step, else m*momentum + g'; p = fma(-lr, m, p)
 * table_dev [n] on the DEVICE, entries in ascending first_block order: entry i owns blocks [first_block_i,
 * first_block_i + oadg_sgd_blocks(numel_i)); total_blocks = their sum.  param / grad / momentum: dense fp32 storage of
 * numel elements in the same element order (any alignment; 16-byte aligned triples take the vector path). */
typedef struct oadg_sgd_tensor {
    void* param;
    const void* grad;
    void* momentum;
    long long numel;
    long long first_block;
    int first_step;            /* != 0: the momentum buffer is uninitialised and receives g' */
    int pad;
} oadg_sgd_tensor;
long long oadg_sgd_blocks(long long numel);
int oadg_sgd_step_multi(const oadg_sgd_tensor* table_dev, int n, long long total_blocks, float lr, float momentum,
                        float weight_decay, void* stream);

/* The first FC of the RoI head on RoIAlign's (ph, pw, c)-ordered features: permute the weight's columns instead of the
 * features
 *   serves ConvFCBBoxHead.forward `x = x.flatten(1)` + shared_fcs[0]   mmdet/models/roi_heads/bbox_heads/convfc_bbox_head.py
 *          (NCHW flatten: column c*P + p; P = roi_feat_area) together with autocast's fp32 -> bf16 weight cast
 * mode 0: src fp32 [O][C][P] -> dst bf16 [O][P][C]        (forward: cast + permute)
 * mode 1: src bf16 [O][P][C] -> dst fp32 [O][C][P]        (the weight gradient back in the parameter's layout)
 * C % 64 == 0, P <= 256, O <= 65535. */
int oadg_fc_weight_permute(const void* src, void* dst, int O, int C, int P, int mode, void* stream);

/* BaseDetector._parse_losses for scalar loss values in one launch
 *   serves mmdet/models/detectors/base.py:234-277: log_vars[name] = value.mean() or sum(v.mean() for v in list),
 *          loss = sum of the variables whose name contains 'loss'
 * values_host [n]: HOST array of device pointers to ONE float each; name_of_host [n]: the index of the variable each
 * value belongs to (ascending; a per-level list has several entries); bit i of is_loss_mask: variable i is part of the
 * total.  packed [n_names + 1] = the variables in order, then the total (also written to total_out [1] when not NULL).
 * The sums are python's: 0 + v0 + v1 + ... */
#define OADG_PARSE_LOSSES_MAX 32
int oadg_parse_losses(const float* const* values_host, const int* name_of_host, int n, int n_names, unsigned is_loss_mask,
                      float* packed, float* total_out, void* stream);

/* RoI head: the regression term of BBoxHead.loss and the logged accuracy in one forward and one backward launch
 *   serves BBoxHead.loss   mmdet/models/roi_heads/bbox_heads/bbox_head.py:397-460 (pos_inds = labels in [0, C),
 *          bbox_pred.view(K, -1, 4)[pos, labels[pos]], loss_bbox(pos_pred, targets[pos], weights[pos], avg_factor=K)),
 *          L1Loss / SmoothL1Loss (mmdet/models/losses/smooth_l1_loss.py) and the fork's L1LossPlus / SmoothL1LossPlus
 *          (losses/oadg/smooth_l1_loss_plus.py:350-552: the leading chunk of the positives = view 1), accuracy()
 *          mmdet/models/losses/accuracy.py (top-1, percent)
 * bbox_pred [K][n_reg] (n_reg = 4 * num_classes, or 4: class-agnostic), cls_score [K][n_cls] (NULL: no accuracy), dtypes
 * 0 = fp32 / 1 = bf16; labels [K] int64; bbox_targets / bbox_weights [K][4] fp32.  Rows r < reg_limit with a label in
 * [0, num_classes) take part in the box loss; beta = 0: L1, > 0: SmoothL1.  out2 = {loss_weight * sum / avg_factor,
 * 100 * #(argmax == label) / K}.  The backward writes grad_pred [K][n_reg] in bbox_pred's dtype (zeros elsewhere);
 * grad_out: the loss's incoming gradient, one float on the device. */
int oadg_roi_reg_acc_fwd(const void* bbox_pred, int pred_dtype, const void* cls_score, int cls_dtype, const int64_t* labels,
                         const float* bbox_targets, const float* bbox_weights, int K, int num_classes, int n_reg,
                         int n_cls, int reg_limit, float beta, float avg_factor, float loss_weight, float* out2,
                         void* stream);
int oadg_roi_reg_bwd(const void* bbox_pred, int pred_dtype, const int64_t* labels, const float* bbox_targets,
                     const float* bbox_weights, int K, int num_classes, int n_reg, int reg_limit, float beta,
                     float avg_factor, float loss_weight, const float* grad_out, void* grad_pred, void* stream);

/* RoI head: MaxIoUAssigner over the proposals of every image + "add the gts as proposals" in three launches
 *   serves StandardRoIHead.forward_train's per-image loop   mmdet/models/roi_heads/standard_roi_head.py:88-101
 *          (bbox_assigner.assign, max_iou_assigner.py:61-213) and the head of BaseSampler.sample
 *          (core/bbox/samplers/base_sampler.py:38-78: bboxes = cat([gt_bboxes, bboxes]), assign_result.add_gt_(gt_labels)
 *          core/bbox/assigners/assign_result.py add_gt_, gt_flags)
 * images_host [B] is a HOST array (device pointers inside): proposals [N][stride] (stride >= 5: column 4 is a score,
 * rows with score < 0 are padding and never candidates), gt_bboxes [num_gts][4], gt_labels [num_gts].
 * Outputs, rows of image b: [Gmax - num_gts_b, Gmax) = its gts (gt_inds j + 1, their labels, overlap 1), [Gmax,
 * Gmax + N) = its proposals with the assignment of oadg_max_iou_assign - so image b's tensors are the contiguous row
 * ranges [Gmax - num_gts_b, Gmax + N) of boxes_full [B][Gmax + N][4], gt_inds_full / labels_full / max_ov_full
 * [B][Gmax + N].  counts [B][2] = candidates among the PROPOSALS only (#gt_inds > 0, #gt_inds == 0).  valid [B][N],
 * gts_pad [B][Gmax][4], gl_pad [B][Gmax], gt_counts [B], workspace (oadg_max_iou_assign_workspace_bytes) are scratch. */
#define OADG_ROI_ASSIGN_MAX_IMAGES 32
typedef struct oadg_roi_assign_image {
    const float* proposals;
    const float* gt_bboxes;
    const int64_t* gt_labels;
    int stride;
    int num_gts;
} oadg_roi_assign_image;
int oadg_roi_assign_add_gt(const oadg_roi_assign_image* images_host, int B, int N, int Gmax, float pos_iou_thr,
                           float neg_iou_lo, float neg_iou_hi, float min_pos_iou, int match_low_quality,
                           float* boxes_full, int64_t* gt_inds_full, int64_t* labels_full, float* max_ov_full,
                           unsigned char* valid, float* gts_pad, int64_t* gl_pad, int* gt_counts, void* workspace,
                           size_t workspace_bytes, int* counts, void* stream);

/* BBoxHead.get_targets for all sampled images + bbox2roi of the sampled boxes in ONE launch
 *   serves BBoxHead._get_target_single / get_targets   mmdet/models/roi_heads/bbox_heads/bbox_head.py:190-257,328-394
 *          (the fork's contrastive head adds the absolute gt boxes: contrastive_head.py get_targets)
 *          bbox2roi                                      mmdet/core/bbox/transforms.py:75-94
 *          SamplingResult.pos_bboxes / pos_gt_bboxes / pos_gt_labels / bboxes   core/bbox/samplers/sampling_result.py
 * entries_host [n_entries] is a HOST array (copied into the launch arguments; device pointers inside).  Entries
 * [0, n_target) are sampled images: rows = positives (pos_inds order) then negatives (neg_inds order), image blocks
 * in entry order; labels (fill_label on negatives), label_weights (pos_weight <= 0 -> 1 on positives, 1 on
 * negatives), bbox_targets = DeltaXYWHBBoxCoder.encode(box, its gt) with the fork's zero-size guard in row-wise form
 * (see oadg_anchor_targets), bbox_weights, absolute (gt box of a positive; may be NULL).  Entries [n_target,
 * n_entries) are raw box lists (npos = 0, nneg rows copied in order, neg_inds NULL) that only produce roi rows
 * (the fork's random proposals, contrastive_roi_head.py:131-137).  rois [all rows][5] = (entry.batch, x1, y1, x2, y2).
 * means4 / stds4 are HOST pointers. */
#define OADG_ROI_TARGET_MAX_ENTRIES 32
typedef struct oadg_roi_target_entry {
    const float* bboxes;       /* [n][stride] candidate boxes, device */
    const float* gt_bboxes;    /* [G][4] */
    const int64_t* gt_inds;    /* [n] 1-based assigned gt of a candidate, 0 = negative */
    const int64_t* labels;     /* [n] assigned labels */
    const int64_t* pos_inds;   /* [npos] */
    const int64_t* neg_inds;   /* [nneg] */
    int npos, nneg;
    int stride;                /* floats per row of bboxes (>= 4) */
    int batch;                 /* value of the roi's batch-index column */
} oadg_roi_target_entry;
int oadg_roi_targets(const oadg_roi_target_entry* entries_host, int n_entries, int n_target, int64_t fill_label,
                     float pos_weight, const float* means4, const float* stds4, float* rois, int64_t* labels,
                     float* label_weights, float* bbox_targets, float* bbox_weights, float* absolute, void* stream);

int oadg_roi_targets_dev(const oadg_roi_target_entry* entries_host, int n_entries, int n_target, int n_src,
                         const int* counts_dev, int64_t fill_label, float pos_weight, const float* means4,
                         const float* stds4, float* rois, int64_t* labels, float* label_weights, float* bbox_targets,
                         float* bbox_weights, float* absolute, void* stream);
/* (the same with the positive / negative split of every sampled image read from device memory: counts_dev [n_src][2] as
 *  written by oadg_roi_sample_device, target entry i uses row i % n_src; entry.npos = the entry's row capacity,
 *  entry.nneg = 0, entry.pos_inds = the image's block of the sampler's `sel`) */

/* RandomSampler.sample of the RoI head for up to 8 images in ONE launch, on the device, consuming ATen's CPU generator
 * stream (bit-identical indices, bit-identical engine state afterwards)
 *   serves BaseSampler.sample / RandomSampler._sample_pos / _sample_neg / random_choice
 *          mmdet/core/bbox/samplers/base_sampler.py:38-103, random_sampler.py:32-82 (torch.randperm(n)[:k] on the CPU
 *          generator: ATen randperm_cpu = forward Fisher-Yates on mt19937 outputs)
 * images_host [B] (HOST array, device pointers inside): gt_inds [n] int64 = the image's assignment with the gts added as
 * proposals in front (> 0 positive, 0 negative, < 0 ignored), n <= oadg_roi_sample_max_rows().  mt_state (device, in,
 * never written) [626] uint32 = at::mt19937 state words [624], left, next; mt_state_out (device, a DIFFERENT buffer)
 * [626] = the advanced state (every workgroup of the launch reads mt_state in no defined order).  Outputs: sel [B][num] int64 = per image the sorted
 * positive indices, then the sorted negative indices; counts [B][2] = k_pos, k_neg; flags [B]: bit 0 = fewer than num rows
 * sampled (the fixed-capacity layout has padding rows), bit 1 = image outside the kernel's domain (nothing sampled). */
typedef struct oadg_roi_sample_image {
    const int64_t* gt_inds;
    int n;
} oadg_roi_sample_image;
int oadg_roi_sample_max_rows(void);
int oadg_roi_sample_device(const oadg_roi_sample_image* images_host, int B, int num, int num_pos_exp, float neg_pos_ub,
                           const uint32_t* mt_state, uint32_t* mt_state_out, int64_t* sel, int* counts, int* flags,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 convolution on the fp32 matrix cores (exact fp32 FMA chains) - the fp32 PARITY path (csrc/conv_f32.hip)
 *   serves the cuDNN convolutions of mmdet/models/backbones/resnet.py:263-302, necks/fpn.py:112-129,
 *          dense_heads/rpn_head.py:54-68 when the model runs in fp32 (the reference's training precision)
 * x [N][H][W][C] fp32 NHWC, C % 4 == 0; w [K][R][S][C]; bias [K] or NULL; y [N][Ho][Wo][K].  transposed = 0: the forward
 * convolution (out_h / out_w ignored).  transposed = 1: the data gradient of a convolution with this stride / pad / dil -
 * x is dy [N][H][W][C] (C = the forward convolution's output channels), w is [K][R][S][C] with K = the forward
 * convolution's INPUT channels (= the forward weight [C][R][S][K] transposed), y = dx [N][out_h][out_w][K]. */
int oadg_conv2d_f32(const float* x, const float* w, const float* bias, float* y, const void* zeros16, int N, int H, int W,
                    int C, int K, int R, int S, int stride, int pad, int dil, int transposed, int out_h, int out_w,
                    void* stream);
/* dw [K][R][S][C] fp32 = sum over output pixels of dy[p][k] * x[p @ tap][c]; workspace: oadg_conv2d_wgrad_f32_splits(...)
 * * K*R*S*C floats (fp32 split partials summed in split order: deterministic).  Any C, K. */
int oadg_conv2d_wgrad_f32_splits(int N, int Ho, int Wo, int C, int K, int R, int S);
int oadg_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int H,
                          int W, int C, int K, int R, int S, int stride, int pad, int dil, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Host helper (no device work): first k entries of ATen's CPU randperm(n) replayed on the MT19937 state
 *   serves RandomSampler.random_choice   mmdet/core/bbox/samplers/random_sampler.py:58
 * state624/left/next are the generator's engine words (in/out); out [k] int64.
 */
int oadg_host_randperm_prefix(uint64_t* state624, int* left, uint64_t* next, int64_t n, int64_t k,
                              int64_t* out);

/* Host helper (no device work): OA-DG's random proposal boxes drawn in place from numpy's legacy MT19937 state
 *   replaces generate_random_bboxes_xy   mmdet/models/detectors/two_stage.py:389-419 (the trial loop of one image)
 * mt = numpy's `mt19937_state*` (np.random.mtrand._rand._bit_generator.ctypes.state_address): the same draws in the same
 * order (randint(0, width), randint(0, height), uniform(scales), uniform(ratios) per trial), the IoU test of
 * mmdet/core/evaluation/bbox_overlaps.py in fp32 against gts [n_gt][4] (n_gt < 0: no test; n_gt == 0 returns -2: the
 * reference raises there); out [num][5] doubles.  Returns the number of boxes written (<= num), < 0 on an error. */
int oadg_np_random_bboxes(void* mt, int img_width, int img_height, int num, const float* gts, int n_gt, double scale_lo,
                          double scale_hi, double ratio_lo, double ratio_hi, int max_iters, double iou_max, double iou_min,
                          double* out);

/* ------------------------------------------------------------------------------------------------
 * A frozen ResNet bottleneck block (identity shortcut, 256 -> 64 -> 64 -> 256, stride 1) as one launch:
 *   mmdet/models/backbones/resnet.py:263-302 Bottleneck.forward with the BatchNorms folded (eval mode, frozen_stages).
 * x, y [N][H][W][256] bf16 NHWC (y must not alias x); w1 [64][256], w2 [64][3][3][64], w3 [256][64] bf16 with the BN
 * scales folded in; b1, b2 [64], b3 [256] fp32 = the folded BN shifts.  Any H, W (edge tiles masked).  No backward pass. */
int oadg_bottleneck_frozen_256(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                               const void* w3, const float* b3, void* y, int N, int H, int W, void* stream);
/* the stage's FIRST block (64 -> 64 -> 64 -> 256 with the 1x1 downsample convolution wd / bd on the shortcut):
 * x [N][H][W][64] (the max-pool output), y [N][H][W][256]; w1 [64][64], w3, wd [256][64]. */
int oadg_bottleneck_frozen_first_64(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                                    const void* w3, const float* b3, const void* wd, const float* bd, void* y, int N, int H,
                                    int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OADG_HIP_H */
