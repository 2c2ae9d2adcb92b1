"""``optimizer_config`` / ``fp16`` of a reference config (mmdet/apis/train.py:153-161): grad_clip is honoured with
mmcv OptimizerHook.clip_grads semantics, everything this build cannot honour is rejected by name (never dropped)."""
import os

import pytest
import torch

import oadg_amd  # noqa: F401
from oadg_amd import Config
from oadg_amd.apis import TrainEngine, parse_optimizer_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r50_fpn_1x_cityscapes_oadg.py')


def test_named_config_has_no_clip_and_cfg_options_enable_it():
    cfg = Config.fromfile(CFG)
    assert parse_optimizer_config(cfg) == dict(grad_clip=None)           # schedules/oadg.py:3 grad_clip=None
    cfg.merge_from_dict({'optimizer_config.grad_clip.max_norm': '35', 'optimizer_config.grad_clip.norm_type': '2'})
    assert parse_optimizer_config(cfg) == dict(grad_clip=dict(max_norm=35, norm_type=2))


def test_unsupported_keys_are_rejected_by_name():
    cfg = Config.fromfile(CFG)
    cfg.merge_from_dict({'fp16.loss_scale': '512.'})
    with pytest.raises(NotImplementedError, match='fp16'):
        parse_optimizer_config(cfg)
    cfg = Config.fromfile(CFG)
    cfg.merge_from_dict({'optimizer_config.type': 'GradientCumulativeOptimizerHook'})
    with pytest.raises(NotImplementedError, match='GradientCumulativeOptimizerHook'):
        parse_optimizer_config(cfg)
    cfg = Config.fromfile(CFG)
    cfg.merge_from_dict({'optimizer_config.cumulative_iters': '4'})
    with pytest.raises(NotImplementedError, match='cumulative_iters'):
        parse_optimizer_config(cfg)
    cfg = Config.fromfile(CFG)
    cfg.merge_from_dict({'optimizer_config.grad_clip.norm_type': '2'})
    with pytest.raises(ValueError, match='max_norm'):
        parse_optimizer_config(cfg)


class _Toy(torch.nn.Module):
    train_cfg = {}

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(4, 3)
        self.unused = torch.nn.Linear(2, 2)            # receives no gradient: OptimizerHook filters it out

    def forward(self, img, img_metas, **kw):
        return dict(loss_x=(self.a(img) ** 2).sum() * 100.0)

    def _parse_losses(self, losses):
        return losses['loss_x'], dict(loss=losses['loss_x'].detach())


def _step(clip):
    torch.manual_seed(0)
    m = _Toy()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    eng = TrainEngine(m, opt, grad_clip=clip)
    w0 = m.a.weight.detach().clone()
    eng.step(dict(img=torch.ones(2, 4), img_metas=[{}, {}]))
    return m, w0, eng


def test_train_engine_clips_the_global_norm_before_the_step():
    m, w0, eng = _step(dict(max_norm=0.5, norm_type=2))
    gn = torch.sqrt(m.a.weight.grad.pow(2).sum() + m.a.bias.grad.pow(2).sum())
    assert abs(float(gn) - 0.5) < 1e-4                           # gradients were scaled in place to the bound
    assert float(eng.last_grad_norm) > 0.5                       # the returned norm is the one BEFORE clipping
    assert torch.allclose(m.a.weight, w0 - 0.1 * m.a.weight.grad)
    assert m.unused.weight.grad is None
    m2, w02, eng2 = _step(None)
    assert eng2.last_grad_norm is None
    assert float(torch.sqrt(m2.a.weight.grad.pow(2).sum() + m2.a.bias.grad.pow(2).sum())) > 10.0
