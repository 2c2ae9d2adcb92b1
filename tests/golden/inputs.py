"""Seeded input builders shared by the fixture generators (make_*.py) and the tests.

numpy's legacy RandomState streams are bit-stable across numpy versions, so the big inputs are rebuilt
from their seed instead of being stored; each fixture keeps a float64 checksum of its inputs."""
import numpy as np


def supcon_inputs(seed, n_fg_per_img=100, n_rand=17, few_fg=False, n_img=2, per_img=512, dim=256):
    """Config-1 shape: n_img images x 512 sampled RoIs x 2 views + 2*n_img*n_rand random-proposal rows."""
    rs = np.random.RandomState(seed)
    labs = []
    for _ in range(n_img):
        lab = np.full(per_img, 8, np.int64)
        nfg = 2 if few_fg else n_fg_per_img
        lab[:nfg] = rs.randint(0, 8, nfg)
        labs.append(lab)
    one_view = np.concatenate(labs)
    labels = np.concatenate([one_view, one_view]).reshape(-1, 1)
    K = labels.shape[0]
    B = K + 2 * n_img * n_rand
    feats = rs.standard_normal((B, dim)).astype(np.float32)
    # cross-view twins are correlated, like features of one RoI seen in two views
    feats[K // 2:K] = 0.7 * feats[:K // 2] + 0.3 * feats[K // 2:K]
    return feats, labels


def checksum(*arrays):
    return np.float64(sum(float(np.asarray(a, dtype=np.float64).sum()) for a in arrays))
