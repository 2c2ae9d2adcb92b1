"""Step factory for bench.py's plumbing run (OADG_BENCH_STEP_FACTORY=ddp_bench_factory:make, tests/test_cli.py): the whole
detector train step on CPU ranks over gloo with the HIP entry points swapped for the oracle - test infrastructure only."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


def make(a, rank, world, distributed):
    import oadg_amd  # noqa: F401
    from oadg_amd import Config, build_detector
    from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
    from oracle.backend import oracle_ops
    from inputs import model_batch
    torch.set_num_threads(3)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    set_random_seed(0)
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det.train()
    eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), distributed=distributed)
    set_random_seed(1 + rank)

    def batch(i):
        b = model_batch(100 * rank + i, a.batch, a.height, a.width, n_gt=4)     # different data on each rank and step
        shape = b['img'].shape[2:] + (3,)
        t = torch.tensor
        return dict(img=t(b['img']), img2=t(b['img2']), gt_bboxes=[t(x) for x in b['gt_bboxes']],
                    gt_bboxes2=[t(x) for x in b['gt_bboxes']], gt_labels=[t(x) for x in b['gt_labels']],
                    multilevel_boxes=[t(x) for x in b['multilevel_boxes']], oamix_boxes=[t(x) for x in b['oamix_boxes']],
                    img_metas=[dict(img_shape=shape, pad_shape=shape, ori_shape=shape, scale_factor=1.0, flip=False)
                               for _ in range(a.batch)])
    w0 = [p.detach().clone() for p in det.parameters() if p.requires_grad]

    def step(i):
        with oracle_ops():
            return eng.step(batch(i))

    def finish():
        ps = [p.detach() for p in det.parameters() if p.requires_grad]
        flat = torch.cat([p.flatten() for p in ps]).double()
        digest = torch.stack([flat.sum(), flat.abs().sum(), (flat * torch.arange(flat.numel(), dtype=torch.float64)).sum()])
        gathered = [torch.zeros_like(digest) for _ in range(world)]
        if distributed:
            dist.all_gather(gathered, digest)
        else:
            gathered = [digest]
        changed = sum(int((p != q).any()) for p, q in zip(ps, w0))
        return {'params_equal_across_ranks': all(bool(torch.equal(g, gathered[0])) for g in gathered),
                'param_tensors_changed': changed, 'param_tensors': len(ps)}
    return dict(step=step, finish=finish)
