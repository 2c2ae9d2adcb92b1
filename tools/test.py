#!/usr/bin/env python
"""Inference entry point with the reference's CLI surface (tools/test.py:24-130, single-GPU ``single_gpu_test``
apis/test.py): CONFIG CHECKPOINT [--out results.pkl] [--eval bbox] [--cfg-options k=v ...].

Runs ``model(return_loss=False, rescale=True, **data)`` over the test split through the device test pipeline and
collects one ``list[np.ndarray [k, 5]]`` (per class) per image, the format ``--out`` pickles in the reference.
``--eval bbox`` reports the COCO-style numbers the reference's Cityscapes / COCO datasets report (AP@[.50:.95], AP50,
AP75, APs/m/l, AR...: oadg_amd/evaluation.py, a restatement of pycocotools' COCOeval, which this image does not have);
``--eval mAP`` the VOC-style AP@0.5 of mmdet/core/evaluation/mean_ap.py ``eval_map`` (area mode).
"""
import argparse
import os
import pickle
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from train import DictAction  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='test (and eval) a model')
    p.add_argument('config', help='test config file path')
    p.add_argument('checkpoint', help="checkpoint file ('none' = random init, for smoke runs)")
    p.add_argument('--work-dir', help='the directory to save the evaluation metrics')
    p.add_argument('--out', help='output result file in pickle format')
    p.add_argument('--eval', type=str, nargs='+', help="evaluation metrics: 'bbox' (COCO style), 'mAP' (VOC style AP@0.5)")
    p.add_argument('--cfg-options', nargs='+', action=DictAction, help='override config entries, key=value')
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm', 'mpi'], default='none')
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--max-samples', type=int, default=None, help='stop after this many images (smoke runs)')
    p.add_argument('--amp', default='bf16', choices=['bf16', 'none'])
    p.add_argument('--allow-partial-checkpoint', action='store_true',
                   help='evaluate even if the checkpoint lacks (or mis-sizes) some model tensors')
    a = p.parse_args()
    os.environ.setdefault('LOCAL_RANK', str(a.local_rank))
    return a


def average_precision(recalls, precisions):
    """mean_ap.py:13-56, mode='area'."""
    mrec = np.concatenate([[0.0], recalls, [1.0]])
    mpre = np.concatenate([[0.0], precisions, [0.0]])
    for i in range(len(mpre) - 1, 0, -1):
        mpre[i - 1] = max(mpre[i - 1], mpre[i])
    ind = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[ind + 1] - mrec[ind]) * mpre[ind + 1]))


def eval_map(results, annotations, num_classes, iou_thr=0.5):
    """mean_ap.py:298-440 (tpfp_default, no ignore / difficult flags): per-class AP and their mean."""
    from oadg_amd.core import bbox_overlaps_np
    aps = []
    for c in range(num_classes):
        scores, tp, n_gt = [], [], 0
        for res, (gtb, gtl) in zip(results, annotations):
            dets, gts = res[c], gtb[gtl == c]
            n_gt += len(gts)
            if len(dets) == 0:
                continue
            order = np.argsort(-dets[:, 4])
            flags = np.zeros(len(dets), dtype=np.float32)
            if len(gts):
                ious = bbox_overlaps_np(dets[:, :4], gts)
                best, arg = ious.max(axis=1), ious.argmax(axis=1)
                covered = np.zeros(len(gts), dtype=bool)
                for i in order:
                    if best[i] >= iou_thr and not covered[arg[i]]:
                        covered[arg[i]] = True
                        flags[i] = 1
            scores.append(dets[:, 4])
            tp.append(flags)
        if n_gt == 0:
            continue
        if not scores:
            aps.append(0.0)
            continue
        s, t = np.concatenate(scores), np.concatenate(tp)
        o = np.argsort(-s)
        ctp, cfp = np.cumsum(t[o]), np.cumsum(1 - t[o])
        aps.append(average_precision(ctp / max(n_gt, 1e-12), ctp / np.maximum(ctp + cfp, 1e-12)))
    return (float(np.mean(aps)) if aps else 0.0), aps


def build_model(cfg, checkpoint, dev, amp, allow_partial=False):
    from oadg_amd import build_detector, hip_conv
    cfg.model.pop('pretrained', None)
    model = build_detector(cfg.model, test_cfg=cfg.get('test_cfg'))
    if checkpoint != 'none':
        from oadg_amd.checkpoint import load_checkpoint
        rep = load_checkpoint(model, checkpoint, map_location='cpu', strict=False, logger=print)
        print(f'{rep["path"]}: {rep["loaded"]} tensors loaded, {len(rep["missing"])} missing, '
              f'{len(rep["unexpected"])} unexpected, {len(rep["mismatched"])} size-mismatched')
        lost = [k for k in rep['missing'] + [m[0] for m in rep['mismatched']]
                if k.startswith(('backbone.', 'neck.', 'rpn_head.', 'roi_head.'))]
        if lost and not allow_partial:
            raise RuntimeError(f'{len(lost)} model tensors are not provided by {checkpoint} (e.g. {lost[:4]}): they '
                               f'would be evaluated at their random initial values; pass --allow-partial-checkpoint to '
                               f'do that anyway')
    else:
        model.init_weights(allow_missing_pretrained=True)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    if amp is not None:
        hip_conv.enable()
    return model


def run_test(model, data_cfg, dev, amp, samples_per_gpu=1, max_samples=None):
    """single_gpu_test (apis/test.py:15-70) over one test-split config: (results, dataset, seconds)"""
    from oadg_amd.datasets import build_dataset
    from oadg_amd.pipelines import DevicePipeline
    ds = build_dataset(data_cfg, default_args=dict(seed=12345, device=dev, test_mode=True), synthetic_fallback=True)
    pipe = DevicePipeline(data_cfg['pipeline'], dtype=amp or torch.float32)
    n = len(ds) if max_samples is None else min(len(ds), max_samples)
    results, t0 = [], time.time()
    for i in range(0, n, samples_per_gpu):
        imgs, boxes, labels = ds.batch(list(range(i, min(i + samples_per_gpu, n))))
        data = pipe.test_batch(imgs)
        with torch.no_grad(), torch.autocast('cuda', dtype=amp, enabled=amp is not None):
            results.extend(model(return_loss=False, rescale=True, **data))
    torch.cuda.synchronize()
    return results, ds, time.time() - t0


def evaluate(results, ds, metrics, num_classes, mmdet_style=False):
    """{'bbox': {AP, AP50, ...}} (COCO style) and / or {'mAP': ...} (VOC style AP@0.5).  ``mmdet_style``: the numbers and
    names of CocoDataset.evaluate (maxDets 100 / 300 / 1000, mAP at 1000 detections: mAP, mAP_50, ..., AR@100, ...) -
    what tools/test.py prints, comparable with the reference's logs; False: COCOeval's defaults (1 / 10 / 100), which is
    what the robustness benchmark aggregates (test_robustness.py coco_eval_with_return)."""
    from oadg_amd import evaluation as E
    out = {}
    idx = range(len(results))
    if 'bbox' in metrics:
        kw = dict(max_dets=E.MMDET_MAX_DETS, names=E.MMDET_METRICS) if mmdet_style else {}
        out['bbox'] = E.coco_eval_bbox(E.dataset_gt_anns(ds, idx), results, num_classes, **kw)
    if 'mAP' in metrics:
        anns = []
        for i in idx:
            if hasattr(ds, 'get_ann_info'):
                a_ = ds.get_ann_info(i)
                anns.append((a_['bboxes'], a_['labels']))
            else:
                anns.append(ds.boxes(i))
        m, aps = eval_map(results, anns, num_classes)
        out['mAP'] = dict(mAP=m, per_class=aps)
    return out


def main():
    a = parse_args()
    from oadg_amd import Config
    cfg = Config.fromfile(a.config)
    if a.cfg_options:
        cfg.merge_from_dict(a.cfg_options)
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
    dev = torch.device('cuda', torch.cuda.current_device())
    amp = torch.bfloat16 if a.amp == 'bf16' else None
    model = build_model(cfg, a.checkpoint, dev, amp, a.allow_partial_checkpoint)
    dcfg = cfg.data.test
    results, ds, dt = run_test(model, dcfg, dev, amp, cfg.data.get('samples_per_gpu', 1), a.max_samples)
    n = len(results)
    print(f'{n} images in {dt:.2f} s ({n / dt:.1f} img/s), {sum(len(c) for r in results for c in r)} detections')
    if a.out:
        assert a.out.endswith(('.pkl', '.pickle')), 'The output file must be a pkl file.'
        with open(a.out, 'wb') as f:
            pickle.dump(results, f)
        print(f'writing results to {a.out}')
    if a.eval:
        assert set(a.eval) <= {'bbox', 'mAP'}, "--eval bbox (COCO style) and / or mAP (VOC style AP@0.5)"
        nc = len(getattr(ds, 'CLASSES', None) or range(dcfg.get('num_classes', 8)))
        ev = evaluate(results, ds, a.eval, nc, mmdet_style=True)
        if 'bbox' in ev:
            print('bbox (COCO style): ' + '  '.join(f'{k} {v:.3f}' for k, v in ev['bbox'].items()))
        if 'mAP' in ev:
            print(f"AP50 (eval_map, area): mAP {ev['mAP']['mAP']:.4f}  per class " +
                  ' '.join(f'{v:.3f}' for v in ev['mAP']['per_class']))
        if a.work_dir:
            import json
            os.makedirs(a.work_dir, exist_ok=True)
            with open(os.path.join(a.work_dir, 'eval.json'), 'w') as f:
                json.dump({k: (v if k == 'bbox' else dict(mAP=v['mAP'], per_class=list(v['per_class'])))
                           for k, v in ev.items()}, f, indent=1)


if __name__ == '__main__':
    main()
