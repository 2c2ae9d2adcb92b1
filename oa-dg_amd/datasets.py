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
import threading
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .registry import DATASETS, build_from_cfg


NATIVE_PNG = os.environ.get('OADG_NATIVE_PNG', '1') == '1'     # PNG files through csrc/png_decode.hip (else PIL)


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
                 proposal_file=None, test_mode=False, filter_empty_gt=True, device='cuda', decode_workers=None, **kwargs):
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
        # Decode workers (PIL releases the interpreter lock inside its decoders, numpy inside the channel flip): a
        # 1024 x 2048 PNG costs 30 - 50 ms of inflate on one core and the step consumes a batch of four every ~27 ms, so
        # several BATCHES are decoded at once (tools/train.py keeps `loader_depth` of them in flight).  The workers run
        # on spare cores of the rank's slice of the host - NOT on the CCD the launch threads are pinned to
        # (apis.pin_rank_to_cores) - when there are any; default count: 12, or what the spare cores allow.
        from .apis import spare_cores
        want = 12 if decode_workers is None else int(decode_workers)
        self._decode_cores = spare_cores(want)
        if decode_workers is None and self._decode_cores is not None:
            want = max(4, len(self._decode_cores))
        self.decode_workers = want
        self.ring_slots = 6            # pinned batch buffers per (batch, shape): more than tools/train.py keeps in flight
        self._rings, self._ring_lock = {}, threading.Lock()
        self._pool = ThreadPoolExecutor(want, thread_name_prefix='oadg-decode', initializer=self._place_worker)
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

    def _place_worker(self):
        if self._decode_cores and hasattr(os, 'sched_setaffinity'):
            try:
                os.sched_setaffinity(0, self._decode_cores)      # (pid 0 = the calling thread)
            except OSError:
                pass

    # ---- loading.py:33-78 LoadImageFromFile (color, BGR)
    def _path(self, idx):
        name = self.data_infos[idx]['filename']
        return os.path.join(self.img_prefix, name) if self.img_prefix else name

    def decode(self, idx):
        from PIL import Image
        with Image.open(self._path(idx)) as im:
            rgb = np.asarray(im.convert('RGB'))
        return np.ascontiguousarray(rgb[:, :, ::-1])

    def decode_into(self, idx, dst):
        """image ``idx`` as BGR bytes into ``dst`` (uint8 [H,W,3] view of host memory).  PNG files go through the native
        decoder (csrc/png_decode.hip: one call without the interpreter lock, pixels written at their final place); any other
        format, or a PNG variant it does not cover, through PIL."""
        path = self._path(idx)
        if NATIVE_PNG and path.lower().endswith('.png'):
            from . import _lib
            rc = _lib.lib().oadg_png_decode_bgr(path.encode(), dst.ctypes.data, dst.shape[0], dst.shape[1])
            if rc == 0:
                return
            if rc not in (-4,):                   # (-4: a PNG variant for PIL; anything else is an error worth seeing)
                _lib.check(rc, f'oadg_png_decode_bgr({path})')
        np.copyto(dst, self.decode(idx))

    def _shape(self, idx):
        info = self.data_infos[idx]
        return int(info['height']), int(info['width'])

    def _slot(self, n, H, W):
        """a pinned [n,H,W,3] batch buffer from a small ring (allocated once per shape: pinning 25 MB per batch costs more
        than decoding it); a slot is reused only after the upload that read it has finished"""
        key = (n, H, W)
        with self._ring_lock:
            ring = self._rings.setdefault(key, dict(bufs=[], events=[], next=0))
            if len(ring['bufs']) < self.ring_slots:
                buf = torch.empty((n, H, W, 3), dtype=torch.uint8)
                if torch.device(self.device).type == 'cuda':
                    buf = buf.pin_memory()
                ring['bufs'].append(buf)
                ring['events'].append(None)
                k = len(ring['bufs']) - 1
            else:
                k = ring['next']
                ring['next'] = (k + 1) % self.ring_slots
            ev = ring['events'][k]
        if ev is not None:
            ev.synchronize()
        return ring, k

    def batch(self, indices):
        """(uint8 [N,H,W,3] on the device, list of float32 [n,4] boxes, list of int64 labels)."""
        indices = list(indices)
        shapes = {self._shape(i) for i in indices}
        assert len(shapes) == 1, 'one image shape per batch (Cityscapes: 1024x2048)'
        H, W = shapes.pop()
        ring, k = self._slot(len(indices), H, W)
        host = ring['bufs'][k]
        views = host.numpy()
        list(self._pool.map(lambda a: self.decode_into(a[1], views[a[0]]), enumerate(indices)))
        if torch.device(self.device).type == 'cuda':
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            ring['events'][k] = ev
        else:
            dev = host.clone()
        anns = [self.get_ann_info(i) for i in indices]
        return dev, [a['bboxes'] for a in anns], [a['labels'] for a in anns]


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
