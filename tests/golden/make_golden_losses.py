"""Generate tests/golden/losses_*.npz by running the REFERENCE's loss modules (by path, see refload.py).

Run here only:  python tests/golden/make_golden_losses.py
Inputs come from numpy's legacy RandomState (bit-stable across numpy versions) and are stored with the
reference outputs, so the fixtures are self-contained data.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refload  # noqa: E402
from inputs import checksum, supcon_inputs  # noqa: E402

refload.install()
CLP = refload.ref('mmdet.models.losses.oadg.contrastive_loss_plus', 'ContrastiveLossPlus')
CEP = refload.ref('mmdet.models.losses.oadg.cross_entropy_loss_plus', 'CrossEntropyLossPlus')
SL1 = refload.ref('mmdet.models.losses.oadg.smooth_l1_loss_plus', 'SmoothL1LossPlus')
L1P = refload.ref('mmdet.models.losses.oadg.smooth_l1_loss_plus', 'L1LossPlus')


def supcon_case(seed, **kw):
    feats, labels = supcon_inputs(seed, **kw)
    f = torch.tensor(feats, requires_grad=True)
    loss = CLP(loss_weight=0.01, temperature=0.06, num_views=2)(f, torch.tensor(labels))
    if loss.requires_grad:
        loss.backward()
        g = f.grad.numpy()
    else:
        g = np.zeros_like(feats)
    return dict(kw=np.array(repr(kw)), seed=np.int64(seed), in_checksum=checksum(feats, labels),
                loss=np.float64(loss.item()), grad_norm=np.float64(np.linalg.norm(g.astype(np.float64))),
                grad_rows=g[::16].copy())


def main():
    out = {}
    for s in range(4):
        c = supcon_case(s, n_fg_per_img=60 + 20 * s, n_rand=13 + s)
        for k, v in c.items():
            out[f'supcon{s}_{k}'] = v
    c = supcon_case(9, few_fg=True)
    for k, v in c.items():
        out[f'supconfew_{k}'] = v
    np.savez_compressed(os.path.join(HERE, 'losses_supcon.npz'), **out)

    out = {}
    rs = np.random.RandomState(100)
    # RoI head: 64 rows x 9 classes, 2 views; label 8 = background; weights 0/1
    x = (rs.standard_normal((64, 9)) * 2).astype(np.float32)
    lab = rs.randint(0, 9, 64).astype(np.int64)
    lab[32:] = lab[:32]
    w = (rs.rand(64) > 0.2).astype(np.float32)
    t = torch.tensor(x, requires_grad=True)
    crit = CEP(use_sigmoid=False, loss_weight=1.0, num_views=2, additional_loss='jsdv1_3_2aug',
               lambda_weight=10, wandb_name='roi_cls')
    avg = max(float((w > 0).sum()), 1.0)
    loss = crit(t, torch.tensor(lab), torch.tensor(w), avg_factor=avg)
    loss.backward()
    out.update(roi_x=x, roi_label=lab, roi_w=w, roi_avg=np.float32(avg), roi_loss=np.float64(loss.item()),
               roi_grad=t.grad.numpy().copy())
    # RPN: 96 rows x 1 logit; labels 0 = fg, 1 = bg; weight 1 on sampled anchors only
    for tag, scale in (('rpn', 1.0), ('rpnwide', 6.0)):
        x = (rs.standard_normal((96, 1)) * scale).astype(np.float32)
        lab = (rs.rand(96) > 0.3).astype(np.int64)
        w = (rs.rand(96) > 0.5).astype(np.float32)
        t = torch.tensor(x, requires_grad=True)
        crit = CEP(use_sigmoid=True, loss_weight=1.0, num_views=2, additional_loss='jsdv1_3_2aug',
                   lambda_weight=0.1, wandb_name='rpn_cls')
        loss = crit(t, torch.tensor(lab), torch.tensor(w), avg_factor=37.0)
        loss.backward()
        out.update({f'{tag}_x': x, f'{tag}_label': lab, f'{tag}_w': w, f'{tag}_avg': np.float32(37.0),
                    f'{tag}_loss': np.float64(loss.item()), f'{tag}_grad': t.grad.numpy().copy()})
    # regression losses on the view-1 chunk
    p = rs.standard_normal((40, 4)).astype(np.float32) * 2
    tg = rs.standard_normal((40, 4)).astype(np.float32)
    w = (rs.rand(40, 4) > 0.3).astype(np.float32)
    for name, mod in (('sl1', SL1(beta=1.0, loss_weight=1.0, num_views=2, additional_loss='None')),
                      ('l1', L1P(loss_weight=1.0, num_views=2, additional_loss='None'))):
        t = torch.tensor(p, requires_grad=True)
        loss = mod(t, torch.tensor(tg), torch.tensor(w), avg_factor=23.0)
        loss.backward()
        out.update({f'{name}_loss': np.float64(loss.item()), f'{name}_grad': t.grad.numpy().copy()})
    out.update(reg_pred=p, reg_target=tg, reg_w=w, reg_avg=np.float32(23.0))
    np.savez_compressed(os.path.join(HERE, 'losses_cls_reg.npz'), **out)
    print('wrote losses_supcon.npz, losses_cls_reg.npz')


if __name__ == '__main__':
    main()
