"""COCO-style bounding-box evaluation (AP@[.50:.95], AP50, AP75, APs/m/l, AR1/10/100, ARs/m/l) for the Cityscapes /
COCO-format test sets, and the corruption-benchmark aggregation (P, mPC, rPC).

The reference evaluates through ``CocoDataset.evaluate`` -> ``pycocotools.cocoeval.COCOeval`` (mmdet/datasets/coco.py:
362-...; mmdet/datasets/cityscapes.py:212-... delegates 'bbox' to it) and aggregates the robustness runs with
tools/analysis_tools/robustness_eval.py:37-118 and test_robustness.py:28-64.  pycocotools is not installed here, so
this file restates COCOeval's published algorithm for iouType 'bbox' (evaluateImg / accumulate / summarize): the greedy
score-ordered matching with crowd and area-range "ignore" regions, the 101-point interpolated precision envelope, and
the twelve summary numbers.  It is pinned by hand-computed cases (tests/test_evaluation.py), not by pycocotools.
"""
import numpy as np

IOU_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
AREA_RNG = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
METRICS = ['AP', 'AP50', 'AP75', 'APs', 'APm', 'APl', 'AR1', 'AR10', 'AR100', 'ARs', 'ARm', 'ARl']
# CocoDataset.evaluate (mmdet/datasets/coco.py:362-575, CityscapesDataset delegates 'bbox' to it): proposal_nums =
# (100, 300, 1000) become cocoEval.params.maxDets, mAP is taken at maxDets[-1] = 1000 and the names are mmdet's
MMDET_MAX_DETS = (100, 300, 1000)
MMDET_METRICS = ['mAP', 'mAP_50', 'mAP_75', 'mAP_s', 'mAP_m', 'mAP_l', 'AR@100', 'AR@300', 'AR@1000', 'AR_s@1000',
                 'AR_m@1000', 'AR_l@1000']


def _iou_xywh(d, g, iscrowd):
    """maskUtils.iou for boxes [x, y, w, h]: intersection / union, or / det area for crowd ground truth"""
    if len(d) == 0 or len(g) == 0:
        return np.zeros((len(d), len(g)))
    d, g = np.asarray(d, np.float64), np.asarray(g, np.float64)
    iw = np.minimum(d[:, None, 0] + d[:, None, 2], g[None, :, 0] + g[None, :, 2]) - np.maximum(d[:, None, 0], g[None, :, 0])
    ih = np.minimum(d[:, None, 1] + d[:, None, 3], g[None, :, 1] + g[None, :, 3]) - np.maximum(d[:, None, 1], g[None, :, 1])
    inter = np.clip(iw, 0, None) * np.clip(ih, 0, None)
    da, ga = d[:, 2] * d[:, 3], g[:, 2] * g[:, 3]
    union = np.where(np.asarray(iscrowd, bool)[None, :], da[:, None], da[:, None] + ga[None, :] - inter)
    with np.errstate(divide='ignore', invalid='ignore'):
        out = inter / union
    return np.where(union > 0, out, 0.0)


def _evaluate_img(gts, dts, a_rng, max_det):
    """COCOeval.evaluateImg for one (image, category).  gts: list of dict(bbox xywh, area, iscrowd); dts: list of
    dict(bbox xywh, area, score).  Returns None when both are empty."""
    if not gts and not dts:
        return None
    g_ig = np.array([1 if (g['iscrowd'] or g['area'] < a_rng[0] or g['area'] > a_rng[1]) else 0 for g in gts], int)
    gtind = np.argsort(g_ig, kind='mergesort')
    gts = [gts[i] for i in gtind]
    g_ig = g_ig[gtind]
    dtind = np.argsort([-d['score'] for d in dts], kind='mergesort')[:max_det]
    dts = [dts[i] for i in dtind]
    iscrowd = [int(g['iscrowd']) for g in gts]
    ious = _iou_xywh([d['bbox'] for d in dts], [g['bbox'] for g in gts], iscrowd)
    T, G, D = len(IOU_THRS), len(gts), len(dts)
    gtm, dtm, dt_ig = np.zeros((T, G)), np.zeros((T, D)), np.zeros((T, D))
    if G and D:
        for ti, t in enumerate(IOU_THRS):
            for di in range(D):
                iou, m = min(t, 1 - 1e-10), -1
                for gi in range(G):
                    if gtm[ti, gi] > 0 and not iscrowd[gi]:
                        continue                      # already matched, and not a crowd
                    if m > -1 and g_ig[m] == 0 and g_ig[gi] == 1:
                        break                         # a regular match exists: do not trade it for an ignored gt
                    if ious[di, gi] < iou:
                        continue
                    iou, m = ious[di, gi], gi
                if m == -1:
                    continue
                dt_ig[ti, di] = g_ig[m]
                dtm[ti, di] = m + 1
                gtm[ti, m] = di + 1
    out_of_range = np.array([d['area'] < a_rng[0] or d['area'] > a_rng[1] for d in dts], bool).reshape(1, D)
    dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(out_of_range, T, 0)))
    return dict(dtm=dtm, dt_ig=dt_ig, g_ig=g_ig, scores=np.array([d['score'] for d in dts], np.float64))


def coco_eval_bbox(gt_anns, results, num_classes, max_dets=None, names=None):
    """gt_anns: per image, list of dict(bbox=[x, y, w, h], category=label index, area, iscrowd);
    results: per image, list over classes of arrays [k, 5] = x1, y1, x2, y2, score (``bbox2result`` format).
    Returns dict(metric name -> value) with the twelve COCO numbers (-1 where undefined).
    ``max_dets`` / ``names``: COCOeval's defaults (1, 10, 100) with its own names - what test_robustness.py's
    coco_eval_with_return reports - or MMDET_MAX_DETS / MMDET_METRICS for the numbers of ``CocoDataset.evaluate``
    (tools/test.py --eval bbox)."""
    MAX_DETS = list(max_dets) if max_dets is not None else [1, 10, 100]
    names = list(names) if names is not None else METRICS
    assert len(MAX_DETS) == 3 and len(names) == 12
    n_img = len(gt_anns)
    assert len(results) == n_img
    T, R, K, A, M = len(IOU_THRS), len(REC_THRS), num_classes, len(AREA_RNG), len(MAX_DETS)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    eps = np.spacing(1)
    for k in range(K):
        per_img = []
        for i in range(n_img):
            gts = [g for g in gt_anns[i] if g['category'] == k]
            dets = np.asarray(results[i][k], np.float64).reshape(-1, 5)
            # CocoDataset._det2json / xyxy2xywh + COCO.loadRes (area = w * h)
            dts = [dict(bbox=[d[0], d[1], d[2] - d[0], d[3] - d[1]], area=(d[2] - d[0]) * (d[3] - d[1]), score=d[4])
                   for d in dets]
            dts = [dts[j] for j in np.argsort([-d['score'] for d in dts], kind='mergesort')[:MAX_DETS[-1]]]
            per_img.append((gts, dts))
        for a, a_rng in enumerate(AREA_RNG):
            evs = [_evaluate_img(g, d, a_rng, MAX_DETS[-1]) for g, d in per_img]
            evs = [e for e in evs if e is not None]
            if not evs:
                continue
            for m, max_det in enumerate(MAX_DETS):
                scores = np.concatenate([e['scores'][:max_det] for e in evs])
                inds = np.argsort(-scores, kind='mergesort')
                dtm = np.concatenate([e['dtm'][:, :max_det] for e in evs], axis=1)[:, inds]
                dt_ig = np.concatenate([e['dt_ig'][:, :max_det] for e in evs], axis=1)[:, inds]
                g_ig = np.concatenate([e['g_ig'] for e in evs])
                npig = np.count_nonzero(g_ig == 0)
                if npig == 0:
                    continue
                tps = np.logical_and(dtm, np.logical_not(dt_ig))
                fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                tp_sum = np.cumsum(tps, axis=1).astype(np.float64)
                fp_sum = np.cumsum(fps, axis=1).astype(np.float64)
                for t in range(T):
                    tp, fp = tp_sum[t], fp_sum[t]
                    nd = len(tp)
                    rc = tp / npig
                    pr = tp / (fp + tp + eps)
                    recall[t, k, a, m] = rc[-1] if nd else 0
                    pr = pr.tolist()
                    for i in range(nd - 1, 0, -1):
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    q = np.zeros((R,))
                    for ri, pi in enumerate(np.searchsorted(rc, REC_THRS, side='left')):
                        if pi < nd:
                            q[ri] = pr[pi]
                    precision[t, :, k, a, m] = q

    def summarize(ap, iou=None, area=0, max_det=2):
        s = precision[:, :, :, area, max_det] if ap else recall[:, :, area, max_det]
        if iou is not None:
            s = s[np.where(np.isclose(IOU_THRS, iou))[0]]
        s = s[s > -1]
        return float(np.mean(s)) if s.size else -1.0
    stats = [summarize(1), summarize(1, .5), summarize(1, .75), summarize(1, None, 1), summarize(1, None, 2),
             summarize(1, None, 3), summarize(0, None, 0, 0), summarize(0, None, 0, 1), summarize(0, None, 0, 2),
             summarize(0, None, 1), summarize(0, None, 2), summarize(0, None, 3)]
    return dict(zip(names, stats))


def dataset_gt_anns(dataset, indices=None):
    """ground truth in the evaluator's format.  COCO-format datasets: the raw json annotations of every image (crowd
    regions included, as COCOeval sees them); synthetic datasets: their boxes (area = w * h, no crowds)."""
    idx = range(len(dataset)) if indices is None else indices
    out = []
    for i in idx:
        if hasattr(dataset, 'coco'):
            info = dataset.data_infos[i]
            anns = dataset.coco.load_anns(dataset.coco.get_ann_ids(img_ids=[info['id']]))
            out.append([dict(bbox=list(a['bbox']), category=dataset.cat2label[a['category_id']], area=a['area'],
                             iscrowd=int(a.get('iscrowd', 0))) for a in anns if a['category_id'] in dataset.cat_ids])
        else:
            b, l = dataset.boxes(i)
            out.append([dict(bbox=[float(x[0]), float(x[1]), float(x[2] - x[0]), float(x[3] - x[1])], category=int(c),
                             area=float((x[2] - x[0]) * (x[3] - x[1])), iscrowd=0) for x, c in zip(b, l)])
    return out


# ------------------------------------------------------------------------------------------ corruption benchmark
CORRUPTION_SETS = {   # tools/analysis_tools/test_robustness.py:225-256
    'all': ['gaussian_noise', 'shot_noise', 'impulse_noise', 'defocus_blur', 'glass_blur', 'motion_blur', 'zoom_blur',
            'snow', 'frost', 'fog', 'brightness', 'contrast', 'elastic_transform', 'pixelate', 'jpeg_compression',
            'speckle_noise', 'gaussian_blur', 'spatter', 'saturate'],
    'noise': ['gaussian_noise', 'shot_noise', 'impulse_noise'],
    'blur': ['defocus_blur', 'glass_blur', 'motion_blur', 'zoom_blur'],
    'weather': ['snow', 'frost', 'fog', 'brightness'],
    'digital': ['contrast', 'elastic_transform', 'pixelate', 'jpeg_compression'],
    'holdout': ['speckle_noise', 'gaussian_blur', 'spatter', 'saturate'],
}
CORRUPTION_SETS['benchmark'] = CORRUPTION_SETS['all'][:15]


def select_corruptions(names, severities):
    """the if-chain of test_robustness.py:225-256: first matching group wins; 'None' = clean data only"""
    for key in ('all', 'benchmark', 'noise', 'blur', 'weather', 'digital', 'holdout'):
        if key in names:
            return list(CORRUPTION_SETS[key]), list(severities)
    if 'None' in names:
        return ['None'], [0]
    return list(names), list(severities)


def corrupted_img_prefix(img_prefix, corruption, severity):
    """--load-dataset corrupted (test_robustness.py:283-299): the pre-generated ``cityscapes-c`` / ``coco-c`` trees"""
    if '/cityscapes/' in img_prefix:
        return f"{img_prefix.replace('cityscapes', 'cityscapes-c')}{corruption}/{severity}/"
    if '/cityscapes-c/' in img_prefix:
        return f'{img_prefix}{corruption}/{severity}/'
    if '/coco/' in img_prefix:
        return f"{img_prefix.replace('coco', 'coco-c')}{corruption}/{severity}/"
    if '/coco-c/' in img_prefix:
        return f'{img_prefix}{corruption}/{severity}/'
    raise NotImplementedError("set load_dataset as 'corrupted' but use original dataset.")


def aggregate_robustness(eval_output, task='bbox', metrics=None, aggregate='benchmark'):
    """robustness_eval.py:37-118 get_coco_style_results: eval_output[corruption][severity][task][metric] ->
    (P, mPC, rPC) arrays over ``metrics``: P = clean performance (first corruption, severity 0), mPC = mean over the
    corruptions (the first 15 for 'benchmark') and severities 1..5, rPC = mPC / P."""
    assert aggregate in ('benchmark', 'all')
    metrics = list(METRICS) if metrics is None else list(metrics)
    res = np.zeros((len(eval_output), 6, len(metrics)), dtype='float32')
    for ci, corruption in enumerate(eval_output):
        for severity in eval_output[corruption]:
            for mj, name in enumerate(metrics):
                res[ci, int(severity), mj] = eval_output[corruption][severity][task][name]
    P = res[0, 0, :]
    mPC = np.mean(res[:15, 1:, :], axis=(0, 1)) if aggregate == 'benchmark' else np.mean(res[:, 1:, :], axis=(0, 1))
    with np.errstate(divide='ignore', invalid='ignore'):
        rPC = mPC / P
    return dict(P=dict(zip(metrics, P.tolist())), mPC=dict(zip(metrics, mPC.tolist())),
                rPC=dict(zip(metrics, rPC.tolist())), table=res)
