"""One tiny pass of the whole hot path (OA-Mix -> detector fwd/bwd -> SGD) on the device, for
``__graft_entry__.smoke()``."""
import os

import numpy as np
import torch


def train_step_smoke(dev):
    from . import Config, build_detector
    from .apis import TrainEngine, build_optimizer, set_random_seed
    from .pipelines import DevicePipeline, SyntheticCityscapes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs', 'oadg', 'faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    set_random_seed(0)
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)
    ds = SyntheticCityscapes(img_shape=(256, 512), num_boxes=8, box_size=(16, 120), device=dev)
    pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
    imgs, boxes, labels = ds.batch([0, 1])
    out = eng.step(pipe(imgs, boxes, labels))
    loss = float(out['loss'])
    assert np.isfinite(loss) and loss > 0, loss
    keys = set(out['log_vars'])
    assert {'loss_rpn_cls', 'loss_rpn_bbox', 'loss_cls', 'acc', 'loss_bbox', 'loss_cont', 'loss'} <= keys, keys
    print('smoke ok: train step loss', round(loss, 4))
