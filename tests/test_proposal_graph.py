"""Padded RPN proposal generation replayed from a hipGraph (dense_heads.RPNHead._proposals_from_graph, experimental and
off by default - see dense_heads.PROPOSAL_GRAPH) must equal the eager path, replay after replay, on new inputs."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_graphed_proposals_equal_eager(dev):
    from oadg_amd import Config, build_detector, dense_heads
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    det = build_detector(cfg.model).to(dev)
    head = det.rpn_head
    pcfg = det.train_cfg.get('rpn_proposal', det.test_cfg.rpn)
    H, W = 256, 384
    metas = [dict(img_shape=(H, W, 3)) for _ in range(4)]
    g = torch.Generator(device=dev).manual_seed(0)
    sizes = [(H // s, W // s) for s in (4, 8, 16, 32, 64)]
    for it in range(3):
        cls = [torch.randn(4, 3, h, w, device=dev, generator=g) for h, w in sizes]
        box = [torch.randn(4, 12, h, w, device=dev, generator=g) * 0.3 for h, w in sizes]
        got = head._proposals_from_graph(cls, box, metas, pcfg, 2)
        assert got is not None, 'hipGraph capture of the proposal path failed on this stack'
        got = [t.clone() for t in got]
        ref = head.get_bboxes(cls, box, img_metas=metas, cfg=pcfg, num_imgs=2, padded=True)
        assert len(got) == len(ref) == 2
        for a, b in zip(got, ref):
            assert torch.equal(a, b), it
    assert len(head._prop_graphs) == 1
