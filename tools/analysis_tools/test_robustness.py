#!/usr/bin/env python
"""Corruption benchmark loop with the reference's CLI surface (tools/analysis_tools/test_robustness.py:94-429):

    python tools/analysis_tools/test_robustness.py CONFIG CHECKPOINT [--out results.pkl]
        [--corruptions benchmark|all|noise|blur|weather|digital|holdout|None|<names...>] [--severities 0 1 2 3 4 5]
        [--eval bbox] [--load-dataset corrupted|original] [--final-prints P mPC rPC] [--final-prints-aggregate all|benchmark]

For every (corruption, severity) the test split is evaluated (severity 0 = clean data, once) and the COCO-style numbers
are collected as ``aggregated[corruption][severity]['bbox'][metric]``; at the end P (clean), mPC (mean performance under
corruption) and rPC (relative) are printed as robustness_eval.py:37-118 does, and the table is written next to --out.

``--load-dataset corrupted`` (the reference's default workflow for Cityscapes-C: pre-generated image trees,
test_robustness.py:283-299) swaps ``img_prefix`` to ``.../cityscapes-c/.../<corruption>/<severity>/``.
``--load-dataset original`` corrupts on the fly with the ``Corrupt`` transform inserted after the loading step
(test_robustness.py:269-277); the reference's transform calls the ``imagecorruptions`` package, which is not installed
in this image: oadg_amd/pipelines/corrupt.py restates all 19 corruptions on numpy / scipy / Pillow and two host loops of
the library (``frost`` needs the package's six photographs: OADG_FROST_DIR).  On a box without the dataset the synthetic
source is used.
"""
import argparse
import copy
import json
import os
import pickle
import sys

import torch

TOOLS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, TOOLS)
sys.path.insert(0, os.path.dirname(TOOLS))
from test import build_model, evaluate, run_test  # noqa: E402
from train import DictAction  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='MMDet test detector under image corruptions')
    p.add_argument('config', help='test config file path')
    p.add_argument('checkpoint', help="checkpoint file ('none' = random init, for smoke runs)")
    p.add_argument('--out', help='output result file')
    p.add_argument('--corruptions', type=str, nargs='+', default=['benchmark'])
    p.add_argument('--severities', type=int, nargs='+', default=[0, 1, 2, 3, 4, 5])
    p.add_argument('--eval', type=str, nargs='+', default=['bbox'], choices=['bbox', 'mAP'])
    p.add_argument('--load-dataset', default='corrupted', choices=['original', 'corrupted'])
    p.add_argument('--final-prints', type=str, nargs='+', choices=['P', 'mPC', 'rPC'], default=['mPC'])
    p.add_argument('--final-prints-aggregate', type=str, choices=['all', 'benchmark'], default='benchmark')
    p.add_argument('--cfg-options', nargs='+', action=DictAction)
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm', 'mpi'], default='none')
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--max-samples', type=int, default=None, help='stop after this many images per run (smoke runs)')
    p.add_argument('--amp', default='bf16', choices=['bf16', 'none'])
    p.add_argument('--allow-partial-checkpoint', action='store_true')
    a = p.parse_args()
    os.environ.setdefault('LOCAL_RANK', str(a.local_rank))
    return a


def main():
    a = parse_args()
    from oadg_amd import Config
    from oadg_amd import evaluation as E
    from oadg_amd.apis import set_random_seed
    cfg = Config.fromfile(a.config)
    if a.cfg_options:
        cfg.merge_from_dict(a.cfg_options)
    if a.seed is not None:
        set_random_seed(a.seed)
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
    dev = torch.device('cuda', torch.cuda.current_device())
    amp = torch.bfloat16 if a.amp == 'bf16' else None
    corruptions, severities = E.select_corruptions(a.corruptions, a.severities)
    model = build_model(cfg, a.checkpoint, dev, amp, a.allow_partial_checkpoint)
    aggregated = {}
    for ci, corruption in enumerate(corruptions):
        aggregated[corruption] = {}
        for severity in severities:
            if ci > 0 and severity == 0:             # severity 0 (= no corruption) is evaluated once
                aggregated[corruption][0] = aggregated[corruptions[0]][0]
                continue
            dcfg = copy.deepcopy(cfg.data.test.to_dict() if hasattr(cfg.data.test, 'to_dict') else dict(cfg.data.test))
            if severity > 0:
                if a.load_dataset == 'original':
                    # test_robustness.py:269-277: the Corrupt transform right after the loading step (index 1)
                    dcfg['pipeline'].insert(1, dict(type='Corrupt', corruption=corruption, severity=severity))
                elif dcfg.get('img_prefix') and dcfg.get('type') != 'SyntheticCityscapes':
                    dcfg['img_prefix'] = E.corrupted_img_prefix(dcfg['img_prefix'], corruption, severity)
            print(f'\nTesting {corruption} at severity {severity}', flush=True)
            results, ds, dt = run_test(model, dcfg, dev, amp, 1, a.max_samples)
            nc = len(getattr(ds, 'CLASSES', None) or range(dcfg.get('num_classes', 8)))
            ev = evaluate(results, ds, a.eval, nc)
            aggregated[corruption][severity] = ev
            if 'bbox' in ev:
                print('  '.join(f'{k} {v:.3f}' for k, v in ev['bbox'].items()), flush=True)
            if a.out:
                stem = os.path.splitext(a.out)[0]
                with open(f'{stem}_results.pkl', 'wb') as f:        # test_robustness.py:411-416: rewritten after every run
                    pickle.dump(aggregated, f)
    if 'bbox' in a.eval:
        agg = E.aggregate_robustness(aggregated, 'bbox', None, a.final_prints_aggregate)
        for name, title in (('P', 'Performance on Clean Data [P] (bbox)'),
                            ('mPC', 'Mean Performance under Corruption [mPC] (bbox)'),
                            ('rPC', 'Relative Performance under Corruption [rPC] (bbox)')):
            if name in a.final_prints:
                print(f'\n{title}')
                for k, v in agg[name].items():
                    print(f'{k:5} = {v:0.3f}' if name != 'rPC' else f'{k:5} => {v * 100:0.1f} %')
        if a.out:
            with open(os.path.splitext(a.out)[0] + '_summary.json', 'w') as f:
                json.dump({k: agg[k] for k in ('P', 'mPC', 'rPC')}, f, indent=1)


if __name__ == '__main__':
    main()
