"""oracle/losses.py against the fixtures produced by the reference's own loss modules (CPU)."""
import os

import numpy as np
import pytest
import torch

from inputs import checksum, supcon_inputs
from oracle import losses as O

REL = 1e-5


def _close(a, b, rel=REL, abs_=1e-7):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.all(np.abs(a - b) <= abs_ + rel * np.abs(b)), float(np.abs(a - b).max())


@pytest.mark.parametrize('tag', ['supcon0', 'supcon1', 'supcon2', 'supcon3', 'supconfew'])
def test_supcon_matches_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, 'losses_supcon.npz'))
    kw = eval(str(g[f'{tag}_kw']))
    feats, labels = supcon_inputs(int(g[f'{tag}_seed']), **kw)
    assert checksum(feats, labels) == float(g[f'{tag}_in_checksum'])
    f = torch.tensor(feats, requires_grad=True)
    loss = O.supcon(f, torch.tensor(labels), temper=0.06, loss_weight=0.01)
    _close(loss.item(), g[f'{tag}_loss'])
    loss.backward()
    gr = f.grad.numpy()
    _close(np.linalg.norm(gr.astype(np.float64)), g[f'{tag}_grad_norm'], rel=1e-4)
    ref_rows = g[f'{tag}_grad_rows']
    assert np.abs(gr[::16] - ref_rows).max() <= 1e-4 * np.abs(ref_rows).max() + 1e-12


def test_supcon_explicit_sizes_equal_literal():
    feats, labels = supcon_inputs(5, n_fg_per_img=50, n_rand=11)
    f = torch.tensor(feats)
    a = O.supcon(f, torch.tensor(labels), temper=0.06)
    b = O.supcon(f, torch.tensor(labels), ori_size=1024, rp_size=22, temper=0.06)
    assert a.item() == b.item()


@pytest.mark.parametrize('tag,sig,lam', [('roi', False, 10.0), ('rpn', True, 0.1), ('rpnwide', True, 0.1)])
def test_ce_jsd_matches_reference(golden_dir, tag, sig, lam):
    g = np.load(os.path.join(golden_dir, 'losses_cls_reg.npz'))
    x = torch.tensor(g[f'{tag}_x'], requires_grad=True)
    tot, ce, js = O.ce_jsd(x, torch.tensor(g[f'{tag}_label']), torch.tensor(g[f'{tag}_w']),
                           float(g[f'{tag}_avg']), sig, 1.0, lam)
    _close(tot.item(), g[f'{tag}_loss'])
    tot.backward()
    _close(x.grad.numpy(), g[f'{tag}_grad'], rel=1e-5, abs_=1e-8)


@pytest.mark.parametrize('name', ['sl1', 'l1'])
def test_reg_losses_match_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'losses_cls_reg.npz'))
    p = torch.tensor(g['reg_pred'], requires_grad=True)
    fn = O.smooth_l1_view1 if name == 'sl1' else O.l1_view1
    loss = fn(p, torch.tensor(g['reg_target']), torch.tensor(g['reg_w']), float(g['reg_avg']))
    _close(loss.item(), g[f'{name}_loss'])
    loss.backward()
    _close(p.grad.numpy(), g[f'{name}_grad'])
