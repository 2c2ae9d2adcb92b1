"""Seeded input builders shared by the fixture generators (make_*.py) and the tests.

numpy's legacy RandomState streams are bit-stable across numpy versions, so the big inputs are rebuilt
from their seed instead of being stored; each fixture keeps a float64 checksum of its inputs."""
import numpy as np


def supcon_inputs(seed, n_fg_per_img=100, n_rand=17, few_fg=False, n_img=2, per_img=512, dim=256):
    """Config-1 shape: n_img images x 512 sampled RoIs x 2 views + 2*n_img*n_rand random-proposal rows."""
    rs = np.random.RandomState(seed)
    labs = []
    for _ in range(n_img):
        lab = np.full(per_img, 8, np.int64)
        nfg = 2 if few_fg else n_fg_per_img
        lab[:nfg] = rs.randint(0, 8, nfg)
        labs.append(lab)
    one_view = np.concatenate(labs)
    labels = np.concatenate([one_view, one_view]).reshape(-1, 1)
    K = labels.shape[0]
    B = K + 2 * n_img * n_rand
    feats = rs.standard_normal((B, dim)).astype(np.float32)
    # cross-view twins are correlated, like features of one RoI seen in two views
    feats[K // 2:K] = 0.7 * feats[:K // 2] + 0.3 * feats[K // 2:K]
    return feats, labels


def checksum(*arrays):
    return np.float64(sum(float(np.asarray(a, dtype=np.float64).sum()) for a in arrays))


# ---------------------------------------------------------------------------------------------- model-level
import zlib  # noqa: E402


def named_weights(shapes, scale=None):
    """Deterministic weights keyed by parameter NAME (so the reference model and ours, which share mmdet's
    state_dict layout, get identical values regardless of construction order).  shapes: {name: shape}.
    ``scale``: {parameter name: factor} applied on top (fixture-specific, e.g. DC5_SCALE)."""
    out = {}
    for name, shape in shapes.items():
        rs = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
        shape = tuple(shape)
        if name.endswith('num_batches_tracked'):
            v = np.zeros(shape, np.int64)
        elif name.endswith('running_var'):
            v = (0.5 + rs.rand(*shape)).astype(np.float32)
        elif name.endswith('running_mean'):
            v = (0.1 * rs.standard_normal(shape)).astype(np.float32)
        elif len(shape) == 1 and name.endswith('weight'):       # BN gamma
            v = (1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32)
        elif name.endswith('bias'):
            v = (0.01 * rs.standard_normal(shape)).astype(np.float32)
        else:                                                    # conv / linear weight: He-scaled normal
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            v = (rs.standard_normal(shape) * np.sqrt(2.0 / fan_in) * 0.7).astype(np.float32)
        if scale and name in scale:
            v = (v * np.float32(scale[name])).astype(v.dtype)
        out[name] = v
    return out


# R101-DC5 fixture: the 2048-channel RPN produces |logits| > 6 with the He-scaled weights above, where fp32 sigmoids of
# DIFFERENT logits collapse to the SAME score (~900 of 17,280 anchors in tie groups).  The reference ranks them with
# ``scores.sort(descending=True)`` (rpn_head.py:150), an unstable sort whose tie order depends on the torch build and
# device, so a whole-step comparison is only well defined without such ties: smaller objectness logits.
DC5_SCALE = {'rpn_head.rpn_cls.weight': 0.05, 'rpn_head.rpn_cls.bias': 0.05}


def lowpass_image(rs, h, w, k=8):
    """uint8 [h,w,3] box-filtered uniform noise (non-degenerate histograms / saliency)."""
    x = rs.randint(0, 256, (h + k, w + k, 3)).astype(np.float64)
    c = np.cumsum(np.cumsum(x, 0), 1)
    c = np.pad(c, ((1, 0), (1, 0), (0, 0)))
    s = c[k:k + h, k:k + w] - c[:h, k:k + w] - c[k:k + h, :w] + c[:h, :w]
    s = s / (k * k)
    s = (s - s.min()) / (s.max() - s.min() + 1e-9) * 255.0
    return s.astype(np.uint8)


def synthetic_boxes(rs, n, h, w, wmin=24, wmax=400):
    bw = rs.uniform(wmin, min(wmax, w // 2), n)
    bh = rs.uniform(wmin, min(wmax, h // 2), n)
    x1 = rs.uniform(0, w - bw)
    y1 = rs.uniform(0, h - bh)
    return np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)


def model_batch(seed, n_img, h, w, n_gt=12, n_cls=8):
    """A train_step input in the collated format of the OA-DG pipeline: normalised float
    images for both views, gt lists, OA-Mix box lists."""
    rs = np.random.RandomState(seed)
    mean = np.array([123.675, 116.28, 103.53], np.float32)
    std = np.array([58.395, 57.12, 57.375], np.float32)
    img, img2, gtb, gtl, mlb, oab = [], [], [], [], [], []
    for _ in range(n_img):
        a = lowpass_image(rs, h, w).astype(np.float32)
        b = np.clip(a * rs.uniform(0.7, 1.2) + rs.uniform(-20, 20) + rs.standard_normal(a.shape) * 6, 0, 255)
        img.append(((a - mean) / std).transpose(2, 0, 1).astype(np.float32))
        img2.append(((b.astype(np.float32) - mean) / std).transpose(2, 0, 1).astype(np.float32))
        gtb.append(synthetic_boxes(rs, n_gt, h, w, 16, min(200, w // 3)))
        gtl.append(rs.randint(0, n_cls, n_gt).astype(np.int64))
        mlb.append(np.round(synthetic_boxes(rs, rs.randint(1, 3), h, w, 8, 60)).astype(np.int64))
        oab.append(np.round(synthetic_boxes(rs, rs.randint(1, 4), h, w, 8, 60)).astype(np.int64))
    return dict(img=np.stack(img), img2=np.stack(img2), gt_bboxes=gtb, gt_labels=gtl, multilevel_boxes=mlb,
                oamix_boxes=oab)


def postproc_inputs(seed, n, num_classes, per_class, W=512, H=256):
    """RoI-head outputs for multiclass_nms: clustered boxes [n, C*4 or 4] and softmax-like scores [n, C+1]."""
    rs = np.random.RandomState(4000 + seed)
    centers = rs.uniform([20, 20], [W - 20, H - 20], size=(max(n // 12, 1), 2))
    c = centers[rs.randint(0, len(centers), n)] + rs.normal(0, 6, size=(n, 2))
    wh = rs.uniform(10, 80, size=(n, 2))
    base = np.concatenate([c - wh / 2, c + wh / 2], axis=1)
    if per_class:
        boxes = (base[:, None, :] + rs.normal(0, 2, size=(n, num_classes, 4))).reshape(n, -1)
    else:
        boxes = base
    logits = rs.normal(0, 2.0, size=(n, num_classes + 1))
    e = np.exp(logits - logits.max(1, keepdims=True))
    return boxes.astype(np.float32), (e / e.sum(1, keepdims=True)).astype(np.float32)
