"""Generate tests/golden/samplers_reference.npz with the REFERENCE's GroupSampler / DistributedGroupSampler
(mmdet/datasets/samplers/group_sampler.py, loaded by path).  Run here only:  python tests/golden/make_golden_samplers.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refload  # noqa: E402


class DS:
    def __init__(self, flag):
        self.flag = np.asarray(flag, dtype=np.uint8)

    def __len__(self):
        return len(self.flag)


def cases():
    rs = np.random.RandomState(0)
    return [dict(flag=np.ones(37, np.uint8), spg=4, world=1), dict(flag=(rs.rand(50) > 0.3).astype(np.uint8), spg=2, world=2),
            dict(flag=(rs.rand(101) > 0.5).astype(np.uint8), spg=4, world=8), dict(flag=np.zeros(5, np.uint8), spg=2, world=4)]


def main():
    refload.install()
    mod = refload.ref('mmdet.datasets.samplers.group_sampler')
    out = {}
    for c, case in enumerate(cases()):
        ds = DS(case['flag'])
        out[f'c{c}_flag'] = case['flag']
        out[f'c{c}_cfg'] = np.array([case['spg'], case['world']], np.int64)
        np.random.seed(100 + c)
        gs = mod.GroupSampler(ds, case['spg'])
        out[f'c{c}_group'] = np.array([list(iter(gs)) for _ in range(2)], np.int64)       # two epochs on one stream
        out[f'c{c}_group_after'] = np.float64(np.random.random())
        for rank in range(case['world']):
            for epoch in (0, 3):
                s = mod.DistributedGroupSampler(ds, case['spg'], case['world'], rank, seed=7)
                s.set_epoch(epoch)
                out[f'c{c}_dist_r{rank}_e{epoch}'] = np.array(list(iter(s)), np.int64)
    np.savez_compressed(os.path.join(HERE, 'samplers_reference.npz'), **out)
    print('wrote samplers_reference.npz', len(out), 'arrays')


if __name__ == '__main__':
    main()
