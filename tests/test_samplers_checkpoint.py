"""CPU: training-order samplers against the reference's own (fixture from tests/golden/make_golden_samplers.py),
mmcv-style checkpoint loading (size-mismatched keys dropped and reported, prefixes, unresolvable names fail loudly),
pretrained backbone initialisation, and resume bookkeeping."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import oadg_amd  # noqa: F401
from oadg_amd import checkpoint as CK
from oadg_amd.samplers import DistributedGroupSampler, GroupSampler, batches, build_sampler


class _DS:
    def __init__(self, flag):
        self.flag = np.asarray(flag, dtype=np.uint8)

    def __len__(self):
        return len(self.flag)


def test_samplers_reproduce_the_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'samplers_reference.npz'))
    c = 0
    while f'c{c}_flag' in g.files:
        ds = _DS(g[f'c{c}_flag'])
        spg, world = (int(v) for v in g[f'c{c}_cfg'])
        np.random.seed(100 + c)
        gs = GroupSampler(ds, spg)
        assert np.array_equal(np.array([gs.indices() for _ in range(2)]), g[f'c{c}_group'])
        assert np.random.random() == float(g[f'c{c}_group_after']), 'the global numpy stream was consumed differently'
        seen = []
        for rank in range(world):
            for epoch in (0, 3):
                s = DistributedGroupSampler(ds, spg, world, rank, seed=7)
                s.set_epoch(epoch)
                idx = s.indices()
                assert np.array_equal(idx, g[f'c{c}_dist_r{rank}_e{epoch}']), (c, rank, epoch)
                if epoch == 0:
                    seen += idx
        assert set(seen) == set(range(len(ds)))                 # every sample is visited in an epoch
        for b in batches(seen, spg):                            # a batch never mixes aspect-ratio groups
            assert len({int(ds.flag[i]) for i in b}) == 1
        c += 1
    assert c == 4


def test_build_sampler_picks_the_reference_class():
    ds = _DS(np.ones(9))
    assert isinstance(build_sampler(ds, 2, False), GroupSampler)
    s = build_sampler(ds, 2, True, rank=1, world=2, seed=3)
    assert isinstance(s, DistributedGroupSampler) and len(s) == 6


class _Net(nn.Module):
    def __init__(self, classes=8):
        super().__init__()
        self.backbone = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4))
        self.fc_cls = nn.Linear(4, classes + 1)


def test_load_checkpoint_drops_size_mismatches_and_reports(tmp_path):
    src = _Net(classes=80)
    path = str(tmp_path / 'coco.pth')
    torch.save(dict(state_dict={'module.' + k: v for k, v in src.state_dict().items()}, meta=dict(epoch=12, iter=99)), path)
    dst = _Net(classes=8)
    before = dst.fc_cls.weight.clone()
    msgs = []
    rep = CK.load_checkpoint(dst, path, logger=msgs.append)
    assert torch.equal(dst.backbone[0].weight, src.backbone[0].weight)           # 'module.' stripped, backbone loaded
    assert torch.equal(dst.fc_cls.weight, before)                                # 81- vs 9-row head left alone
    assert {m[0] for m in rep['mismatched']} == {'fc_cls.weight', 'fc_cls.bias'}
    assert rep['meta'] == dict(epoch=12, iter=99) and rep['missing'] == [] and rep['unexpected'] == []
    assert any('size mismatch for fc_cls.weight' in m for m in msgs)
    with pytest.raises(RuntimeError):
        CK.load_checkpoint(dst, path, strict=True, logger=None)
    # prefix: only the backbone of a detector checkpoint
    bb = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4))
    rep = CK.load_checkpoint(bb, path, prefix='backbone', logger=None)
    assert torch.equal(bb[0].weight, src.backbone[0].weight) and not rep['missing'] and not rep['mismatched']
    with pytest.raises(RuntimeError):
        CK.load_checkpoint(bb, path, prefix='neck', logger=None)


def test_unresolvable_pretrained_names_fail_loudly(tmp_path, monkeypatch):
    monkeypatch.setenv('OADG_CHECKPOINT_DIR', str(tmp_path))
    for name in ('torchvision://resnet50', 'open-mmlab://detectron2/resnet50_caffe',
                 'https://download.openmmlab.com/mmdetection/v2.0/faster_rcnn/x/faster_rcnn_r50_fpn_1x_coco.pth',
                 '/nonexistent/file.pth'):
        with pytest.raises(FileNotFoundError):
            CK.resolve_checkpoint(name)
    (tmp_path / 'resnet50-0676ba61.pth').write_bytes(b'x')
    assert CK.resolve_checkpoint('torchvision://resnet50') == str(tmp_path / 'resnet50-0676ba61.pth')


def test_backbone_init_cfg_pretrained_is_honoured(tmp_path, monkeypatch):
    from oadg_amd import Config, build_detector
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs', '_base_', 'faster_rcnn_r50_fpn.py'))      # the stock COCO model config
    assert cfg.model.backbone.init_cfg['checkpoint'] == 'torchvision://resnet50'
    monkeypatch.setenv('OADG_CHECKPOINT_DIR', str(tmp_path))
    monkeypatch.delenv('OADG_ALLOW_RANDOM_INIT', raising=False)
    det = build_detector(cfg.model)
    with pytest.raises(FileNotFoundError):                 # configured, unavailable, not waived: stop
        det.init_weights()
    with pytest.warns(UserWarning, match='RANDOM initialisation'):
        det.init_weights(allow_missing_pretrained=True)
    # a torchvision-style file (no prefix) in the checkpoint directory is loaded into the backbone
    sd = {k: torch.full_like(v, 0.25) if v.dtype.is_floating_point else v for k, v in det.backbone.state_dict().items()}
    sd['fc.weight'] = torch.zeros(1000, 2048)              # torchvision's classifier: unexpected, reported, ignored
    torch.save(sd, str(tmp_path / 'resnet50-test.pth'))
    det.init_weights()
    assert float(det.backbone.layer3[0].conv1.weight.mean()) == 0.25 and float(det.backbone.bn1.running_var.mean()) == 0.25
    # ... and a detector checkpoint contributes its 'backbone.' keys
    torch.save(dict(state_dict={'backbone.' + k: v * 2 if v.dtype.is_floating_point else v for k, v in sd.items()
                                if k != 'fc.weight'}), str(tmp_path / 'resnet50-test.pth'))
    det.init_weights()
    assert float(det.backbone.layer3[0].conv1.weight.mean()) == 0.5
