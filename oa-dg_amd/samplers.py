"""Training-order samplers of the reference's DataLoader (mmdet/datasets/samplers/group_sampler.py:10-148, chosen by
mmdet/datasets/builder.py:128-160): every epoch visits the samples in a fresh random order, grouped by the
aspect-ratio flag (custom.py:209-221) so that a batch never mixes landscape and portrait images.

* ``GroupSampler`` (non-distributed): draws from the GLOBAL numpy stream of the training process - ``shuffle`` per
  group, ``choice`` for the padding, ``permutation`` of the batches - like the reference.
* ``DistributedGroupSampler``: a torch ``Generator`` seeded ``epoch + seed`` (identical on every rank), ``randperm``
  per group, padding by repetition, ``randperm`` over the batches, then rank r takes the r-th contiguous slice.
Both return plain index lists; ``batches()`` cuts them into ``samples_per_gpu`` chunks.
"""
import math

import numpy as np
import torch


def dataset_flags(dataset):
    flag = getattr(dataset, 'flag', None)
    if flag is None:
        flag = np.ones(len(dataset), dtype=np.uint8)          # Cityscapes-shaped synthetic samples: width > height
    return np.asarray(flag)


class GroupSampler:

    def __init__(self, dataset, samples_per_gpu=1):
        self.samples_per_gpu = samples_per_gpu
        self.flag = dataset_flags(dataset).astype(np.int64)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = 0
        for size in self.group_sizes:
            self.num_samples += int(np.ceil(size / self.samples_per_gpu)) * self.samples_per_gpu

    def set_epoch(self, epoch):
        pass

    def indices(self):
        out = []
        for i, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            ind = np.where(self.flag == i)[0]
            np.random.shuffle(ind)
            extra = int(np.ceil(size / self.samples_per_gpu)) * self.samples_per_gpu - len(ind)
            out.append(np.concatenate([ind, np.random.choice(ind, extra)]))
        out = np.concatenate(out)
        order = np.random.permutation(range(len(out) // self.samples_per_gpu))
        out = np.concatenate([out[i * self.samples_per_gpu:(i + 1) * self.samples_per_gpu] for i in order])
        out = out.astype(np.int64).tolist()
        assert len(out) == self.num_samples
        return out

    def __len__(self):
        return self.num_samples


class DistributedGroupSampler:

    def __init__(self, dataset, samples_per_gpu=1, num_replicas=1, rank=0, seed=0):
        self.samples_per_gpu, self.num_replicas, self.rank = samples_per_gpu, num_replicas, rank
        self.epoch, self.seed = 0, seed if seed is not None else 0
        self.flag = dataset_flags(dataset)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = 0
        for size in self.group_sizes:
            self.num_samples += int(math.ceil(size * 1.0 / samples_per_gpu / num_replicas)) * samples_per_gpu
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def indices(self):
        g = torch.Generator()
        g.manual_seed(self.epoch + self.seed)
        out = []
        for i, size in enumerate(self.group_sizes):
            if size > 0:
                ind = np.where(self.flag == i)[0]
                ind = ind[list(torch.randperm(int(size), generator=g).numpy())].tolist()
                extra = int(math.ceil(size * 1.0 / self.samples_per_gpu / self.num_replicas)) * \
                    self.samples_per_gpu * self.num_replicas - len(ind)
                tmp = ind.copy()
                for _ in range(extra // size):
                    ind.extend(tmp)
                ind.extend(tmp[:extra % size])
                out.extend(ind)
        assert len(out) == self.total_size
        s = self.samples_per_gpu
        out = [out[j] for i in list(torch.randperm(len(out) // s, generator=g)) for j in range(i * s, (i + 1) * s)]
        off = self.num_samples * self.rank
        out = out[off:off + self.num_samples]
        assert len(out) == self.num_samples
        return out

    def __len__(self):
        return self.num_samples


def build_sampler(dataset, samples_per_gpu, distributed, rank=0, world=1, seed=0):
    """datasets/builder.py:128-160 for the EpochBasedRunner with shuffle=True"""
    if distributed:
        return DistributedGroupSampler(dataset, samples_per_gpu, world, rank, seed=seed)
    return GroupSampler(dataset, samples_per_gpu)


def batches(index_list, samples_per_gpu):
    """DataLoader(batch_size=samples_per_gpu, drop_last=False) over the sampler's indices"""
    return [index_list[i:i + samples_per_gpu] for i in range(0, len(index_list), samples_per_gpu)]
