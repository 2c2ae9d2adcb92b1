#!/usr/bin/env python
"""Inference entry point with the reference's CLI surface (tools/test.py:24-130, single-GPU ``single_gpu_test``
apis/test.py): CONFIG CHECKPOINT [--out results.pkl] [--eval bbox] [--cfg-options k=v ...].

Runs ``model(return_loss=False, rescale=True, **data)`` over the test split through the device test pipeline and
collects one ``list[np.ndarray [k, 5]]`` (per class) per image, the format ``--out`` pickles in the reference.
``--eval bbox`` reports VOC-style AP@0.5 (mmdet/core/evaluation/mean_ap.py ``eval_map``, area mode) against the
dataset's boxes: the COCO-style Cityscapes evaluator needs pycocotools, which this image does not have.
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
    p.add_argument('--eval', type=str, nargs='+', help="evaluation metrics: 'bbox'")
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


def main():
    a = parse_args()
    from oadg_amd import Config, build_detector, hip_conv
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    cfg = Config.fromfile(a.config)
    if a.cfg_options:
        cfg.merge_from_dict(a.cfg_options)
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
    dev = torch.device('cuda', torch.cuda.current_device())
    cfg.model.pop('pretrained', None)
    model = build_detector(cfg.model, test_cfg=cfg.get('test_cfg'))
    if a.checkpoint != 'none':
        from oadg_amd.checkpoint import load_checkpoint
        rep = load_checkpoint(model, a.checkpoint, map_location='cpu', strict=False, logger=print)
        print(f'{rep["path"]}: {rep["loaded"]} tensors loaded, {len(rep["missing"])} missing, '
              f'{len(rep["unexpected"])} unexpected, {len(rep["mismatched"])} size-mismatched')
        lost = [k for k in rep['missing'] + [m[0] for m in rep['mismatched']]
                if k.startswith(('backbone.', 'neck.', 'rpn_head.', 'roi_head.'))]
        if lost and not a.allow_partial_checkpoint:
            raise RuntimeError(f'{len(lost)} model tensors are not provided by {a.checkpoint} (e.g. {lost[:4]}): they '
                               f'would be evaluated at their random initial values; pass --allow-partial-checkpoint to '
                               f'do that anyway')
    else:
        model.init_weights(allow_missing_pretrained=True)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    amp = torch.bfloat16 if a.amp == 'bf16' else None
    if amp is not None:
        hip_conv.enable()
    from oadg_amd.datasets import build_dataset
    dcfg = cfg.data.test
    ds = build_dataset(dcfg, default_args=dict(seed=12345, device=dev, test_mode=True), synthetic_fallback=True)
    pipe = DevicePipeline(dcfg.pipeline, dtype=amp or torch.float32)
    n = len(ds) if a.max_samples is None else min(len(ds), a.max_samples)
    bs = cfg.data.get('samples_per_gpu', 1)
    results, annotations, t0 = [], [], time.time()
    for i in range(0, n, bs):
        imgs, boxes, labels = ds.batch(list(range(i, min(i + bs, n))))
        data = pipe.test_batch(imgs)
        with torch.no_grad(), torch.autocast('cuda', dtype=amp, enabled=amp is not None):
            results.extend(model(return_loss=False, rescale=True, **data))
        annotations.extend(zip([np.asarray(b) for b in boxes], [np.asarray(l) for l in labels]))
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f'{n} images in {dt:.2f} s ({n / dt:.1f} img/s), {sum(len(c) for r in results for c in r)} detections')
    if a.out:
        assert a.out.endswith(('.pkl', '.pickle')), 'The output file must be a pkl file.'
        with open(a.out, 'wb') as f:
            pickle.dump(results, f)
        print(f'writing results to {a.out}')
    if a.eval:
        assert a.eval == ['bbox'], "only --eval bbox is built"
        m, aps = eval_map(results, annotations, len(getattr(ds, 'CLASSES', None) or range(dcfg.get('num_classes', 8))))
        print(f'AP50 (eval_map, area): mAP {m:.4f}  per class ' + ' '.join(f'{v:.3f}' for v in aps))


if __name__ == '__main__':
    main()
