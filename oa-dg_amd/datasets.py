"""Real input path: COCO-format annotation files + image files -> uint8 batches in HBM.

``CocoDataset`` / ``CityscapesDataset`` follow mmdet/datasets/{custom,coco,cityscapes}.py for what the training loop
needs (``load_annotations``, ``_filter_imgs``, ``_parse_ann_info``, ``get_ann_info``, ``prepare_train_img``'s
``img_info`` / ``ann_info``); the COCO index is a plain-json restatement of the few ``pycocotools.COCO`` lookups those
functions use (pycocotools is not installed).  Images are decoded on the host (PIL; PNG is lossless, so the bytes equal
``mmcv.imfrombytes(..., flag='color')`` = BGR order) by a small thread pool and uploaded through pinned memory; from
there on everything is the device pipeline.  ``batch(indices)`` has ``SyntheticCityscapes``' signature.
"""
import json
import os
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .registry import DATASETS, build_from_cfg


class _CocoIndex:
    """The part of pycocotools.coco.COCO that CocoDataset calls (createIndex, getCatIds, getImgIds, getAnnIds, ...)."""

    def __init__(self, ann_file):
        with open(ann_file) as f:
            self.dataset = json.load(f)
        self.anns, self.imgs, self.cats = {}, {}, {}
        self.img_to_anns, self.cat_img_map = defaultdict(list), defaultdict(list)
        for ann in self.dataset.get('annotations', []):
            self.img_to_anns[ann['image_id']].append(ann)
            self.anns[ann['id']] = ann
            self.cat_img_map[ann['category_id']].append(ann['image_id'])
        for img in self.dataset.get('images', []):
            self.imgs[img['id']] = img
        for cat in self.dataset.get('categories', []):
            self.cats[cat['id']] = cat

    def get_cat_ids(self, cat_names):
        # pycocotools filters the category list in FILE order: the result does not follow the order of cat_names
        return [c['id'] for c in self.dataset.get('categories', []) if c['name'] in cat_names]

    def get_img_ids(self):
        return list(self.imgs.keys())

    def get_ann_ids(self, img_ids):
        return [a['id'] for i in img_ids for a in self.img_to_anns.get(i, [])]

    def load_anns(self, ids):
        return [self.anns[i] for i in ids]


@DATASETS.register_module()
class CocoDataset:
    CLASSES = None

    def __init__(self, ann_file, pipeline=None, classes=None, data_root=None, img_prefix='', seg_prefix=None,
                 proposal_file=None, test_mode=False, filter_empty_gt=True, device='cuda', decode_workers=4, **kwargs):
        self.ann_file, self.data_root, self.img_prefix = ann_file, data_root, img_prefix
        self.test_mode, self.filter_empty_gt, self.device = test_mode, filter_empty_gt, device
        self.pipeline_cfg = pipeline
        if classes is not None:
            self.CLASSES = tuple(classes)
        if data_root is not None:
            if not os.path.isabs(self.ann_file):
                self.ann_file = os.path.join(data_root, self.ann_file)
            if self.img_prefix and not os.path.isabs(self.img_prefix):
                self.img_prefix = os.path.join(data_root, self.img_prefix)
        self.data_infos = self.load_annotations(self.ann_file)
        if not test_mode:
            valid = self._filter_imgs()
            self.data_infos = [self.data_infos[i] for i in valid]
        self._pool = ThreadPoolExecutor(decode_workers, thread_name_prefix='oadg-decode')
        # custom.py:209-221 _set_group_flag: images with aspect ratio > 1 form group 1 (the samplers group by it)
        self.flag = np.array([1 if i['width'] / i['height'] > 1 else 0 for i in self.data_infos], dtype=np.uint8)

    # ---- coco.py:40-67
    def load_annotations(self, ann_file):
        self.coco = _CocoIndex(ann_file)
        self.cat_ids = self.coco.get_cat_ids(cat_names=self.CLASSES)
        self.cat2label = {cat_id: i for i, cat_id in enumerate(self.cat_ids)}
        self.img_ids = self.coco.get_img_ids()
        data_infos, total = [], []
        for i in self.img_ids:
            info = dict(self.coco.imgs[i])
            info['filename'] = info['file_name']
            data_infos.append(info)
            total.extend(self.coco.get_ann_ids(img_ids=[i]))
        assert len(set(total)) == len(total), f"Annotation ids in '{ann_file}' are not unique!"
        return data_infos

    # ---- coco.py:112-138
    def _filter_imgs(self, min_size=32):
        valid_inds, valid_img_ids = [], []
        ids_with_ann = set(a['image_id'] for a in self.coco.anns.values())
        ids_in_cat = set()
        for class_id in self.cat_ids:
            ids_in_cat |= set(self.coco.cat_img_map[class_id])
        ids_in_cat &= ids_with_ann
        for i, info in enumerate(self.data_infos):
            if self.filter_empty_gt and self.img_ids[i] not in ids_in_cat:
                continue
            if self._skip_all_crowd(info):
                continue
            if min(info['width'], info['height']) >= min_size:
                valid_inds.append(i)
                valid_img_ids.append(self.img_ids[i])
        self.img_ids = valid_img_ids
        return valid_inds

    def _skip_all_crowd(self, info):
        return False

    def __len__(self):
        return len(self.data_infos)

    def get_ann_info(self, idx):
        img_id = self.data_infos[idx]['id']
        return self._parse_ann_info(self.data_infos[idx], self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id])))

    # ---- coco.py:140-199
    def _parse_ann_info(self, img_info, ann_info):
        gt_bboxes, gt_labels, gt_bboxes_ignore = [], [], []
        for ann in ann_info:
            if ann.get('ignore', False):
                continue
            x1, y1, w, h = ann['bbox']
            if self._degenerate(ann, img_info, x1, y1, w, h):
                continue
            if ann['category_id'] not in self.cat_ids:
                continue
            bbox = [x1, y1, x1 + w, y1 + h]
            if ann.get('iscrowd', False):
                gt_bboxes_ignore.append(bbox)
            else:
                gt_bboxes.append(bbox)
                gt_labels.append(self.cat2label[ann['category_id']])
        return dict(
            bboxes=np.array(gt_bboxes, dtype=np.float32) if gt_bboxes else np.zeros((0, 4), dtype=np.float32),
            labels=np.array(gt_labels, dtype=np.int64) if gt_labels else np.array([], dtype=np.int64),
            bboxes_ignore=np.array(gt_bboxes_ignore, dtype=np.float32) if gt_bboxes_ignore
            else np.zeros((0, 4), dtype=np.float32))

    @staticmethod
    def _degenerate(ann, img_info, x1, y1, w, h):
        inter_w = max(0, min(x1 + w, img_info['width']) - max(x1, 0))
        inter_h = max(0, min(y1 + h, img_info['height']) - max(y1, 0))
        return inter_w * inter_h == 0 or ann['area'] <= 0 or w < 1 or h < 1

    # ---- loading.py:33-78 LoadImageFromFile (color, BGR)
    def decode(self, idx):
        from PIL import Image
        path = os.path.join(self.img_prefix, self.data_infos[idx]['filename']) if self.img_prefix \
            else self.data_infos[idx]['filename']
        with Image.open(path) as im:
            rgb = np.asarray(im.convert('RGB'))
        return np.ascontiguousarray(rgb[:, :, ::-1])

    def batch(self, indices):
        """(uint8 [N,H,W,3] on the device, list of float32 [n,4] boxes, list of int64 labels)."""
        indices = list(indices)
        arrs = list(self._pool.map(self.decode, indices))
        assert all(a.shape == arrs[0].shape for a in arrs), 'one image shape per batch (Cityscapes: 1024x2048)'
        host = torch.from_numpy(np.stack(arrs))
        if torch.device(self.device).type == 'cuda':
            host = host.pin_memory().to(self.device, non_blocking=True)
        anns = [self.get_ann_info(i) for i in indices]
        return host, [a['bboxes'] for a in anns], [a['labels'] for a in anns]


@DATASETS.register_module()
class CityscapesDataset(CocoDataset):
    """cityscapes.py:19-110."""
    CLASSES = ('person', 'rider', 'car', 'truck', 'bus', 'train', 'motorcycle', 'bicycle')

    def _skip_all_crowd(self, info):
        anns = self.coco.load_anns(self.coco.get_ann_ids(img_ids=[info['id']]))
        return self.filter_empty_gt and all(a['iscrowd'] for a in anns)

    @staticmethod
    def _degenerate(ann, img_info, x1, y1, w, h):
        return ann['area'] <= 0 or w < 1 or h < 1


@DATASETS.register_module()
class RepeatDataset:
    """dataset_wrappers.py RepeatDataset: ``times`` passes over the wrapped dataset per epoch."""

    def __init__(self, dataset, times):
        self.dataset, self.times = dataset, times
        self.CLASSES = getattr(dataset, 'CLASSES', None)
        self._ori_len = len(dataset)
        if hasattr(dataset, 'flag'):                      # dataset_wrappers.py:170-171
            self.flag = np.tile(dataset.flag, times)

    def __len__(self):
        return self.times * self._ori_len

    def batch(self, indices):
        return self.dataset.batch([i % self._ori_len for i in indices])

    def get_ann_info(self, idx):
        return self.dataset.get_ann_info(idx % self._ori_len)


def _files_present(cfg):
    ann = cfg.get('ann_file')
    if ann is None:
        return True
    root = cfg.get('data_root')
    return os.path.exists(ann if (root is None or os.path.isabs(ann)) else os.path.join(root, ann))


def build_dataset(cfg, default_args=None, synthetic_fallback=False):
    """datasets/builder.py:54-77 (RepeatDataset + plain datasets).  ``synthetic_fallback``: a dataset whose annotation
    file does not exist on this machine (an unmodified reference config on a box without Cityscapes) is replaced by
    the Cityscapes-shaped synthetic source, loudly."""
    cfg = dict(cfg)
    if cfg.get('type') == 'RepeatDataset' and synthetic_fallback and not _files_present(cfg['dataset']):
        cfg = dict(cfg['dataset'])          # one synthetic pass is as good as eight
    if synthetic_fallback and cfg.get('type') != 'RepeatDataset' and not _files_present(cfg):
        from .pipelines import SyntheticCityscapes
        print(f"[oadg] {cfg.get('ann_file')} not found: using SyntheticCityscapes in place of {cfg.get('type')}", flush=True)
        keep = {k: v for k, v in (default_args or {}).items() if k in ('seed', 'device', 'test_mode')}
        return SyntheticCityscapes(pipeline=cfg.get('pipeline'), **keep)
    if cfg.get('type') == 'RepeatDataset':
        return RepeatDataset(build_dataset(cfg['dataset'], default_args, synthetic_fallback), cfg['times'])
    return build_from_cfg(cfg, DATASETS, default_args)
