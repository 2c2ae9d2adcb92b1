"""CPU (numpy + Pillow) restatement of OA-Mix.  Test infrastructure only (see oracle/__init__).

Follows, line by line in behaviour (not in text):
  mmdet/datasets/pipelines/oa_mix.py:15-29 (op lists), 34-72 (ctor), 74-93 (_get_mask), 95-120 (get_fg_regions),
    122-184 (get_random_regions), 187-204 (__call__), 207-243 (oamix), 245-262, 264-279 (aug), 281-309
  mmdet/datasets/pipelines/bbox_augmentation.py:31-118 (bbox-only ops), 240-302 (bg-only ops)
  mmdet/datasets/pipelines/augmix.py:32-78,83-212 (leaf ops, level sampling)
  mmdet/core/evaluation/bbox_overlaps.py:5-65

Global ``np.random`` is consumed in exactly the reference's order.  Array dtypes follow what the
reference's expressions evaluate to under NumPy >= 2 promotion (the version installed here and on the GPU box):
float32 for the mask algebra, float64 for the two terms that multiply a Python float into a uint8 image
(oa_mix.py:306 and bbox_augmentation.py:266-270).  OpenCV leaves come from oracle/cvleaves.py (PARITY UNPINNED);
with those leaves plugged into the genuine reference as its ``cv2`` the outputs of the two are bit-identical
(tests/test_oracle_oamix.py), which pins everything else.

Masks are kept as single-channel [H,W] float32 planes (the reference's three channels are identical copies).
"""
import numpy as np
from PIL import Image, ImageEnhance, ImageOps

from . import cvleaves as cv

f32 = np.float32


def bbox_overlaps(b1, b2, eps=1e-6):
    b1 = np.asarray(b1).astype(np.float32).reshape(-1, 4)
    b2 = np.asarray(b2).astype(np.float32).reshape(-1, 4)
    rows, cols = b1.shape[0], b2.shape[0]
    if rows * cols == 0:
        return np.zeros((rows, cols), np.float32)
    swap = rows > cols
    if swap:
        b1, b2 = b2, b1
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    out = np.zeros((b1.shape[0], b2.shape[0]), np.float32)
    for i in range(b1.shape[0]):
        w = np.maximum(np.minimum(b1[i, 2], b2[:, 2]) - np.maximum(b1[i, 0], b2[:, 0]), 0)
        h = np.maximum(np.minimum(b1[i, 3], b2[:, 3]) - np.maximum(b1[i, 1], b2[:, 1]), 0)
        inter = w * h
        out[i] = inter / np.maximum(a1[i] + a2 - inter, eps)
    return out.T if swap else out


def sample_level(n):
    return np.random.uniform(low=0.1, high=n)


def int_parameter(level, maxval):
    return int(level * maxval / 10)


def float_parameter(level, maxval):
    return float(level) * maxval / 10.


COLOR_OPS = ('autocontrast', 'equalize', 'posterize', 'solarize', 'color', 'contrast', 'brightness', 'sharpness')
AUG_LISTS = {
    'augmix': ['autocontrast', 'equalize', 'posterize', 'solarize', 'bboxes_only_rotate',
               'bboxes_only_shear_xy', 'bboxes_only_translate_xy', 'bg_only_rotate', 'bg_only_shear_xy',
               'bg_only_translate_xy'],
    'augmix.all': ['autocontrast', 'equalize', 'posterize', 'solarize', 'invert', 'color', 'contrast',
                   'brightness', 'sharpness', 'bboxes_only_rotate', 'bboxes_only_shear_xy',
                   'bboxes_only_translate_xy', 'bg_only_rotate', 'bg_only_shear_xy', 'bg_only_translate_xy'],
}


def color_op(name, img_u8, severity):
    """augmix.py:64-78,103-105,192-212 through Pillow (version-stable LUT / enhance arithmetic)."""
    pil = Image.fromarray(img_u8, 'RGB')
    if name == 'autocontrast':
        out = ImageOps.autocontrast(pil)
    elif name == 'equalize':
        out = ImageOps.equalize(pil)
    elif name == 'posterize':
        out = ImageOps.posterize(pil, 4 - int_parameter(sample_level(severity), 4))
    elif name == 'solarize':
        out = ImageOps.solarize(pil, 256 - int_parameter(sample_level(severity), 256))
    else:
        factor = float_parameter(sample_level(severity), 1.8) + 0.1
        out = {'color': ImageEnhance.Color, 'contrast': ImageEnhance.Contrast,
               'brightness': ImageEnhance.Brightness, 'sharpness': ImageEnhance.Sharpness}[name](pil).enhance(factor)
    return np.asarray(out)


def geo_matrix(kind, severity, img_size, center=None, size_for_level=None):
    """The affine matrix one leaf draws (augmix.py:83-188): returns (M as the dtype cv2 receives, params)."""
    if kind == 'rotate':
        deg = int_parameter(sample_level(severity), 30)
        if np.random.uniform() > 0.5:
            deg = -deg
        c = center if center is not None else (img_size[0] / 2, img_size[1] / 2)
        return cv.get_rotation_matrix_2d(c, deg, 1.0), ('rotate', deg)
    if kind in ('shear_x', 'shear_y'):
        lvl = float_parameter(sample_level(severity), 0.3)
        if np.random.uniform() > 0.5:
            lvl = -lvl
        if kind == 'shear_x':
            tx = 0 if center is None else -lvl * center[1]
            return np.float32([[1, -lvl, -tx], [0, 1, 0]]), (kind, lvl)
        ty = 0 if center is None else -lvl * center[0]
        return np.float32([[1, 0, 0], [-lvl, 1, -ty]]), (kind, lvl)
    ax = 0 if kind == 'translate_x' else 1
    maxval = img_size[ax] if size_for_level is None else size_for_level[ax]
    lvl = int_parameter(sample_level(severity), maxval / 3)
    if np.random.random() > 0.5:
        lvl = -lvl
    M = np.float32([[1, 0, -lvl], [0, 1, 0]]) if ax == 0 else np.float32([[1, 0, 0], [0, 1, -lvl]])
    return M, (kind, lvl)


class OAMixOracle:

    def __init__(self, version='augmix', num_views=2, keep_orig=True, severity=10, mixture_width=3,
                 mixture_depth=-1, random_box_scale=(0.01, 0.1), random_box_ratio=(3, 1 / 3),
                 oa_random_box_scale=(0.005, 0.1), oa_random_box_ratio=(3, 1 / 3), num_bboxes=(3, 5),
                 spatial_ratio=4, sigma_ratio=0.3, saliency_fn=None, **kwargs):
        self.aug_list = AUG_LISTS[version]
        self.num_views, self.keep_orig, self.severity = num_views, keep_orig, severity
        self.aug_prob_coeff, self.mixture_width, self.mixture_depth = 1.0, mixture_width, mixture_depth
        self.random_box_scale, self.random_box_ratio = random_box_scale, random_box_ratio
        self.oa_random_box_scale, self.oa_random_box_ratio = oa_random_box_scale, oa_random_box_ratio
        self.score_thresh = 10
        self.spatial_ratio, self.sigma_ratio = spatial_ratio, sigma_ratio
        self.saliency_fn = saliency_fn or cv.saliency_score
        self._history = {}
        self.trace = []          # (tag, payload) records, compared against the product's host program

    # ---------------------------------------------------------------- masks / regions
    def blur_mask(self, box, H, W):
        my, mx = cv.box_mask_profiles(box, H, W, self.spatial_ratio, self.sigma_ratio)
        return (my[:, None] * mx[None, :]).astype(np.float32)

    @staticmethod
    def sharp_mask(box, H, W):
        x1, y1, x2, y2 = box
        m = np.zeros((H, W), np.float32)
        m[y1:y2, x1:x2] = 1.0
        return m

    def get_fg_regions(self, img, gt_bboxes):
        H, W = img.shape[:2]
        masks, scores = [], []
        for gt in gt_bboxes:
            x1, y1, x2, y2 = np.array(gt, dtype=np.int32)
            if x2 - x1 < self.spatial_ratio or y2 - y1 < self.spatial_ratio:
                scores.append(-1)
            else:
                scores.append(self.saliency_fn(img[y1:y2, x1:x2]))
            masks.append(self.blur_mask(gt, H, W))
        self.trace.append(('fg_scores', [float(s) for s in scores]))
        return gt_bboxes, masks, scores

    def get_random_regions(self, img, scale, ratio, num_bboxes, return_score=False, fg_box_list=None,
                           fg_score_list=None, max_iters=50, eps=1e-6):
        H, W = img.shape[:2]
        boxes, masks, scores = [], [], []
        target = np.random.randint(*num_bboxes) if isinstance(num_bboxes, tuple) else num_bboxes
        for _ in range(max_iters):
            if len(masks) >= target:
                break
            x1, y1 = np.random.randint(0, W), np.random.randint(0, H)
            _scale = np.random.uniform(*scale) * H * W
            _ratio = np.random.uniform(*ratio)
            bw, bh = int(np.sqrt(_scale / _ratio)), int(np.sqrt(_scale * _ratio))
            if x1 + bw > W or y1 + bh > H:
                continue
            box = np.array([[x1, y1, min(x1 + bw, W), min(y1 + bh, H)]])
            if np.sum(bbox_overlaps(box, np.asarray(boxes))) > eps:
                continue
            if return_score:
                ious = bbox_overlaps(box, fg_box_list)
                final = float('inf')
                if np.sum(ious) > eps:
                    for iou, fb, fs in zip(ious[0], fg_box_list, fg_score_list):
                        if iou == 0.0 or fb[2] - fb[0] < 1 or fb[3] - fb[1] < 1:
                            continue
                        if fs < final:
                            final = fs
                scores.append(final)
            masks.append(self.sharp_mask(box[0], H, W))
            boxes += list(box)
        return (boxes, masks, scores) if return_score else (boxes, masks)

    # ---------------------------------------------------------------- ops
    def bboxes_only(self, img, kind, fg_boxes, fg_masks):
        """bbox_augmentation.py:31-88 for one leaf kind: every gt box in turn warps the WHOLE current image
        about the box centre and blends it in through that box's blurred mask."""
        H, W = img.shape[:2]
        for box, b in zip(fg_boxes, fg_masks):
            x1, y1, x2, y2 = int(box[0]), int(box[1]), int(box[2]), int(box[3])
            if (x2 - x1) < 1 or (y2 - y1) < 1:
                continue
            center = ((x1 + x2) / 2., (y1 + y2) / 2.)
            M, p = geo_matrix(kind, self.severity, (W, H), center, (x2 - x1 + 1, y2 - y1 + 1))
            self.trace.append(('bbox_leaf', p))
            warped = cv.warp_affine(img, M, (W, H))
            mask = (1.0 - b)[..., None]                              # float32
            img = np.asarray(img * mask + warped * (1.0 - mask), dtype=np.uint8)
        return img

    def bg_only(self, img, kind, fg_masks):
        """bbox_augmentation.py:240-272."""
        H, W = img.shape[:2]
        if len(fg_masks) == 0:
            mask = np.zeros((H, W), np.uint8)
        else:
            mask = np.max(fg_masks, axis=0)
        M, p = geo_matrix(kind, self.severity, (W, H))
        self.trace.append(('bg_leaf', p))
        warped = cv.warp_affine(img, M, (W, H))
        wmask = cv.warp_affine(np.asarray(mask * 255, dtype=np.uint8), M, (W, H))
        keep = np.maximum(mask, wmask / 255)[..., None]               # float64
        return np.asarray(keep * img + (1.0 - keep) * warped, dtype=np.uint8)

    def aug(self, img, fg_boxes, fg_masks):
        name = self.aug_list[np.random.choice(len(self.aug_list))]
        self.trace.append(('op', name))
        if name in COLOR_OPS:
            return color_op(name, img, self.severity)
        if name == 'invert':
            tx = 1 if np.random.random() > 0.5 else -1
            ty = 1 if np.random.random() > 0.5 else -1
            return -cv.warp_affine(img, np.float32([[1, 0, tx], [0, 1, ty]]), (0, 0))
        scope, kind = name.split('_only_')
        if kind.endswith('_xy'):
            kind = kind[:-2] + ('x' if np.random.rand() < 0.5 else 'y')
        if scope == 'bboxes':
            return self.bboxes_only(img, kind, fg_boxes, fg_masks)
        return self.bg_only(img, kind, fg_masks)

    # ---------------------------------------------------------------- the transform
    def oamix(self, img, gt_bboxes):
        img = np.asarray(img, dtype=np.uint8)
        H, W = img.shape[:2]
        ws = np.float32(np.random.dirichlet([self.aug_prob_coeff] * self.mixture_width))
        self.trace.append(('ws', ws.tolist()))
        rboxes, rmasks = self.get_random_regions(img, self.random_box_scale, self.random_box_ratio, (1, 3))
        self._history['random_box_list'] = np.stack(rboxes, axis=0)
        fg_boxes, fg_masks, fg_scores = self.get_fg_regions(img, gt_bboxes)
        region = np.full((H, W), -1, np.int32)
        for i, b in enumerate(rboxes):
            region[b[1]:b[3], b[0]:b[2]] = i
        mix = np.zeros(img.shape, np.float32)
        for i in range(self.mixture_width):
            depth = self.mixture_depth if self.mixture_depth > 0 else np.random.randint(1, 4)
            cur = img.copy()
            for _ in range(depth):
                outs = [np.asarray(self.aug(cur, fg_boxes, fg_masks), dtype=np.uint8) for _ in rboxes]
                outside = np.asarray(self.aug(cur, fg_boxes, fg_masks), dtype=np.uint8)
                # img_tmp + (1 - union) * outside with disjoint 0/1 masks is an exact per-region selection
                nxt = outside.copy()
                for k, o in enumerate(outs):
                    sel = region == k
                    nxt[sel] = o[sel]
                cur = nxt
            mix += ws[i] * np.asarray(cur, dtype=np.float32)
        tboxes, tmasks, tscores = self.targets_for_mixing(img, fg_boxes, fg_masks, fg_scores)
        out = self.object_aware_mixing(img, mix, tmasks, tscores)
        return np.asarray(out, dtype=np.uint8)

    def targets_for_mixing(self, img, fg_boxes, fg_masks, fg_scores):
        tb, tm, ts = [], [], []
        for box, mask, score in zip(fg_boxes, fg_masks, fg_scores):
            if score <= self.score_thresh:
                tb.append(box); tm.append(mask); ts.append(score)
        rb, rm, rs = self.get_random_regions(img, self.oa_random_box_scale, self.oa_random_box_ratio,
                                             num_bboxes=min(max(len(tb), 1), 5), return_score=True,
                                             fg_box_list=fg_boxes, fg_score_list=fg_scores)
        self._history['oa_random_box_list'] = rb
        return tb + rb, tm + rm, ts + rs

    def object_aware_mixing(self, img, img_aug, mask_list, score_list):
        m = np.random.beta(self.aug_prob_coeff, self.aug_prob_coeff)
        self.trace.append(('beta', float(m)))
        shape = img.shape[:2] + (1,)
        orig = np.zeros(img.shape, np.float32)
        aug = np.zeros(img.shape, np.float32)
        mask_sum = np.zeros(shape, np.float32)
        mask_max = None
        for mask, score in zip(mask_list, score_list):
            mask = mask[..., None]
            mask_sum = mask_sum + mask
            mask_max = mask if mask_max is None else np.maximum(mask_max, mask)
            overlap = mask_sum - mask_max
            m_oa = np.float32(np.random.uniform(0.0, 0.5)) if score <= self.score_thresh \
                else np.float32(np.random.uniform(0.0, 1.0))
            self.trace.append(('m_oa', float(m_oa)))
            wgt = mask - overlap * 0.5
            orig += (1.0 - m_oa) * img * wgt
            aug += m_oa * img_aug * wgt
            mask_sum = mask_max
        out = orig + aug
        out += (1.0 - m) * img * (1.0 - mask_sum)        # float64 term, rounded into float32 on the add
        out += m * img_aug * (1.0 - mask_sum)
        return np.clip(out, 0, 255)

    def __call__(self, results):
        results['custom_field'] = []
        for i in range(1, self.num_views + 1):
            if i == 1:
                self._history = {}
                if not self.keep_orig:
                    results['img'] = self.oamix(results['img'].copy(), results['gt_bboxes'].copy())
                results['img_fields'] = ['img']
            else:
                results[f'img{i}'] = self.oamix(results['img'].copy(), results['gt_bboxes'].copy())
                results['img_fields'] += [f'img{i}']
                results[f'gt_bboxes{i}'] = results['gt_bboxes'].copy()
                results['oamix_boxes'] = np.stack(self._history['oa_random_box_list'], axis=0)
                results['custom_field'] += [f'img{i}', f'gt_bboxes{i}', 'oamix_boxes']
                results['multilevel_boxes'] = self._history['random_box_list']
                results['custom_field'] += ['multilevel_boxes']
        return results
