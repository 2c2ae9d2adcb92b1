"""The part of the reference's train pipeline that follows image loading, executed on the device for a whole
batch: OAMix -> Normalize -> Pad -> DefaultFormatBundle -> Collect (+ mmcv collate), SURVEY.md 3.2 / 8a a13.

``Normalize`` / ``Pad`` / ``DefaultFormatBundle`` / ``Collect`` are registered so the reference's pipeline lists
parse unchanged; they only carry their configuration.  ``DevicePipeline`` reads that configuration and produces
the collated batch ``train_step`` consumes, with Normalize+Pad fused into the OA-Mix output kernel
(transforms.py:618-629,699-701; formating.py:217-255,289-357): images leave the pipeline as normalised NHWC
tensors (logical [N,3,H,W], channels_last), never as host arrays.
"""
import ctypes
import os

import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr
from ..registry import DATASETS, PIPELINES, build_from_cfg
from .corrupt import Corrupt
from .geometric import RandomFlip, Resize
from . import oa_mix as _oa_mix
from .oa_mix import OAMix, _ImageState


@PIPELINES.register_module()
class Normalize:
    def __init__(self, mean, std, to_rgb=True):
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb

    def as_args(self):
        # mmcv.imnormalize: (x - mean) * (1 / std) after the BGR->RGB swap
        return dict(mean=[float(v) for v in self.mean],
                    stdinv=[float(np.float32(1.0 / np.float64(s))) for s in self.std], to_rgb=self.to_rgb)


@PIPELINES.register_module()
class Pad:
    def __init__(self, size=None, size_divisor=None, pad_val=0):
        assert size is not None or size_divisor is not None
        self.size, self.size_divisor, self.pad_val = size, size_divisor, pad_val

    def padded(self, h, w):
        if self.size is not None:
            return int(self.size[0]), int(self.size[1])
        d = self.size_divisor
        return int(np.ceil(h / d)) * d, int(np.ceil(w / d)) * d


@PIPELINES.register_module()
class DefaultFormatBundle:
    def __init__(self, img_to_float=True, pad_val=dict(img=0, masks=0, seg=255)):
        self.img_to_float = img_to_float


@PIPELINES.register_module()
class Collect:
    def __init__(self, keys, meta_keys=('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape',
                                        'scale_factor', 'flip', 'flip_direction', 'img_norm_cfg')):
        self.keys, self.meta_keys = keys, meta_keys


@PIPELINES.register_module()
class ImageToTensor:
    def __init__(self, keys):
        self.keys = keys


@PIPELINES.register_module()
class MultiScaleFlipAug:
    """test_time_aug.py:12-121: carries the scales / flips and the inner transform list; DevicePipeline.test_batch expands
    them into one batch per augmentation (the reference's test configs use ``img_scale=(2048, 1024), flip=False``)."""

    def __init__(self, transforms, img_scale=None, scale_factor=None, flip=False, flip_direction='horizontal'):
        assert (img_scale is None) ^ (scale_factor is None), 'Must have but only one variable can be set'
        self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
        self.scale_factor = scale_factor
        self.flip, self.flip_direction = flip, flip_direction
        self.transforms = Compose(transforms)


class Compose:
    """mmdet/datasets/pipelines/compose.py: build each transform from the PIPELINES registry."""

    def __init__(self, transforms):
        self.transforms = [build_from_cfg(t, PIPELINES) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


def _side_stream(device):
    """the pipeline's HIP stream.  (Round 5 tried a CU-masked stream here - hipExtStreamCreateWithCUMask, 16 / 32 / 64 of the
    256 compute units, so that OA-Mix's ~400 small launches stop taking compute units from under the training stream's
    one-workgroup-per-CU kernels: the pipeline itself became the bottleneck, 27.6 -> 42 - 62 ms per step)"""
    return torch.cuda.Stream(device=device)


BATCH_STATES = True         # the image states of a same-shape batch through _ImageState.batch (one launch set per batch)
BATCH_STATES_MAX_BOXES = 1024


class DevicePipeline:
    """Batched device execution of [OAMix,] Normalize, Pad, DefaultFormatBundle, Collect."""

    def __init__(self, pipeline_cfg, dtype=torch.float32, one_scale_per_batch=False, oamix_workers=None):
        """``one_scale_per_batch``: a multi-scale Resize draws its scale once per batch instead of once per sample - an
        opt-in shortcut (one tensor shape per step).  The default follows the reference: one draw per sample
        (transforms.py:177-243), differently sized samples padded into one batch tensor like mmcv's collate does."""
        self.one_scale_per_batch = one_scale_per_batch
        # OA-Mix of the images of a batch on several helper threads, each with its own HIP stream, numpy stream and
        # scratch buffers (the analogue of the reference's ``workers_per_gpu`` DataLoader processes, each of which
        # augments whole samples with a private random stream): see _oamix_parallel.  1 = in the calling thread.
        self.oamix_workers = max(1, int(oamix_workers if oamix_workers is not None
                                        else os.environ.get('OADG_OAMIX_WORKERS', '1')))
        self._helpers = None
        ts = Compose(pipeline_cfg).transforms
        # test_robustness.py:269-277 inserts Corrupt right after the image loading step, before MultiScaleFlipAug
        self.corrupt = next((t for t in ts if isinstance(t, Corrupt)), None)
        self.test_aug = next((t for t in ts if isinstance(t, MultiScaleFlipAug)), None)
        if self.test_aug is not None:        # test pipeline: the transforms live inside MultiScaleFlipAug
            ts = self.test_aug.transforms.transforms
        self.oamix = next((t for t in ts if isinstance(t, OAMix)), None)
        self.resize = next((t for t in ts if isinstance(t, Resize)), None)
        self.flip = next((t for t in ts if isinstance(t, RandomFlip)), None)
        self.norm = next((t for t in ts if isinstance(t, Normalize)), None)
        self.pad = next((t for t in ts if isinstance(t, Pad)), None)
        collect = next((t for t in ts if isinstance(t, Collect)), None)
        self.keys = list(collect.keys) if collect else ['img', 'gt_bboxes', 'gt_labels']
        assert self.norm is not None, 'the pipeline needs a Normalize step'
        self.dtype = dtype

    @torch.no_grad()
    def test_batch(self, imgs_u8):
        """The reference's test pipeline ([Corrupt,] MultiScaleFlipAug(img_scale, flip)[Resize(keep_ratio), RandomFlip,
        Normalize, Pad, ImageToTensor, Collect]) for a resident uint8 batch: returns ``dict(img=[tensor, ...],
        img_metas=[[meta, ...], ...])`` as ``forward_test`` expects - one entry per test-time augmentation, scales outer,
        (no flip, then one flip per direction) inner (test_time_aug.py:99-121).  A ``Corrupt`` step in front of it
        (tools/analysis_tools/test_robustness.py --load-dataset original) corrupts the uint8 images first."""
        assert self.test_aug is not None, 'test_batch needs a test pipeline (MultiScaleFlipAug)'
        L = _lib.lib()
        if self.corrupt is not None:
            imgs_u8 = self.corrupt.batch(imgs_u8)
        N, H0, W0 = imgs_u8.shape[:3]
        aug = self.test_aug
        flips = [(False, None)]
        if aug.flip:
            dirs = aug.flip_direction if isinstance(aug.flip_direction, list) else [aug.flip_direction]
            flips += [(True, d) for d in dirs]
        if aug.scale_factor is not None:
            sfs = aug.scale_factor if isinstance(aug.scale_factor, list) else [aug.scale_factor]
            scales = [(int(W0 * f), int(H0 * f)) for f in sfs]           # test_time_aug.py:101-103
        else:
            scales = aug.img_scale
        na = self.norm.as_args()
        mean, stdinv = (ctypes.c_float * 3)(*na['mean']), (ctypes.c_float * 3)(*na['stdinv'])
        out_imgs, out_metas = [], []
        for scale in scales:
            for flip, direction in flips:
                imgs, metas = [], []
                for i in range(N):
                    im, meta = imgs_u8[i], dict(ori_shape=(H0, W0, 3), flip=False, flip_direction=None,
                                                scale_factor=np.ones(4, dtype=np.float32))
                    if self.resize is not None and scale is not None:
                        im, _, m = self.resize(im, np.zeros((0, 4), np.float32), scale=scale)
                        meta.update(m)
                    if flip:
                        assert self.flip is not None, 'MultiScaleFlipAug(flip=True) needs a RandomFlip step'
                        im, _, m = self.flip(im, np.zeros((0, 4), np.float32), preset=(True, direction))
                        meta.update(m)
                    imgs.append(im)
                    metas.append(meta)
                H, W = imgs[0].shape[:2]
                Hp, Wp = self.pad.padded(H, W) if self.pad is not None else (H, W)
                img = torch.empty((N, 3, Hp, Wp), dtype=self.dtype, device=imgs_u8.device, memory_format=torch.channels_last)
                for i in range(N):
                    check(L.oadg_oamix_normalize(ptr(imgs[i].contiguous()), H, W, mean, stdinv, int(na['to_rgb']),
                                                 ctypes.c_void_p(img.data_ptr() + i * img.stride(0) * img.element_size()),
                                                 1 if self.dtype == torch.bfloat16 else 0, Hp, Wp, stream_ptr()),
                          'oadg_oamix_normalize')
                    metas[i].update(img_shape=(H, W, 3), pad_shape=(Hp, Wp, 3))
                out_imgs.append(img)
                out_metas.append(metas)
        return dict(img=out_imgs, img_metas=out_metas)

    def prefetch(self, imgs_u8, gt_bboxes, gt_labels, worker_seed=None, ready=None):
        """Enqueue the whole pipeline for one batch on a side stream and return a handle (``.get()``).

        This is the device-side analogue of the reference's DataLoader workers (``workers_per_gpu`` processes
        with prefetching, configs/OA-DG/cityscapes/faster_rcnn_r50_fpn_1x_cityscapes.py:27): augmentation of
        batch i+1 overlaps the training step of batch i.  The inputs must already be complete on the device
        (resident batches) or ``ready`` is the event recorded after their producer (decode + upload, synthetic
        generation) on its stream; OA-Mix's small host reads only wait on this stream.

        ``worker_seed`` not None: the host side of the pipeline (OA-Mix's draws and ~1000 launches, ~10 ms) runs
        in a worker THREAD with its own ``RandomState(worker_seed)`` stream - a DataLoader worker likewise owns a
        private numpy stream - so it also overlaps the step's host work; None keeps it in the caller's thread on
        the global numpy stream (bit-reproducible against the oracle with a single ``np.random.seed``)."""
        if getattr(self, '_stream', None) is None:
            self._stream = _side_stream(imgs_u8.device)
            self._stream.wait_stream(torch.cuda.current_stream())
        if worker_seed is None:
            return _Prefetched(*self._run_on_side_stream(imgs_u8, gt_bboxes, gt_labels, ready))
        if getattr(self, '_pool', None) is None:
            from concurrent.futures import ThreadPoolExecutor
            from .oa_mix import use_random_state
            dev, rs = imgs_u8.device, np.random.RandomState(worker_seed)

            def init():
                torch.cuda.set_device(dev)
                use_random_state(rs)
            self._pool = ThreadPoolExecutor(1, thread_name_prefix='oadg-pipeline', initializer=init)
        return _PrefetchedFuture(self._pool.submit(self._run_on_side_stream, imgs_u8, gt_bboxes, gt_labels, ready))

    def _run_on_side_stream(self, imgs_u8, gt_bboxes, gt_labels, ready=None):
        if ready is not None:
            self._stream.wait_event(ready)
        with torch.cuda.stream(self._stream):
            out = self(imgs_u8, gt_bboxes, gt_labels)
            evt = torch.cuda.Event()
            evt.record()
        if imgs_u8.is_cuda:
            # the source images were allocated on their producer's stream and are read HERE: without this their memory could
            # be handed to a new tensor of that stream as soon as the caller drops them (a streaming loader does, right after
            # this call returns) - while this stream's kernels are still reading
            imgs_u8.record_stream(self._stream)
        return out, evt

    def __call__(self, imgs_u8, gt_bboxes, gt_labels):
        """imgs_u8: uint8 [N,H,W,3] cuda tensor (BGR bytes); gt_bboxes: list of float32 [n_i,4] numpy arrays;
        gt_labels: list of int64 numpy arrays.  Returns the collated dict for ``train_step``."""
        L = _lib.lib()
        geo_meta = None
        if self.resize is not None or (self.flip is not None and self.flip.flip_ratio):
            # Resize / RandomFlip per sample, in the reference's order (all draws of a sample's geometric steps, then
            # the next sample; OA-Mix draws follow below - a DataLoader worker interleaves them per sample, which is
            # the same stream split whenever OA-Mix gets its own worker stream)
            imgs, boxes, geo_meta = [], [], []
            for i in range(imgs_u8.shape[0]):
                im, bx, meta = imgs_u8[i], np.asarray(gt_bboxes[i], dtype=np.float32), {}
                if self.resize is not None:
                    shared = geo_meta[0]['scale'] if (self.one_scale_per_batch and geo_meta) else None
                    im, bx, m = self.resize(im, bx, scale=shared)
                    meta.update(m)
                if self.flip is not None and self.flip.flip_ratio:
                    im, bx, m = self.flip(im, bx)
                    meta.update(m)
                imgs.append(im)
                boxes.append(bx)
                geo_meta.append(meta)
            imgs_u8 = torch.stack(imgs) if all(im.shape == imgs[0].shape for im in imgs) else imgs
            gt_bboxes = boxes
        # per-sample image shapes (the reference's multi-scale Resize draws a scale per SAMPLE, transforms.py:177-243):
        # every image is padded to its own multiple of size_divisor (Pad, transforms.py:699-701) and the batch tensor
        # takes the largest padded extent, zero-filled right / bottom - what mmcv's collate does with padded
        # DataContainers; img_metas carry the per-image img_shape / pad_shape the RPN's valid flags and proposal
        # clipping read
        per_image = isinstance(imgs_u8, list)
        N = len(imgs_u8)
        shapes = [tuple(int(v) for v in im.shape[:2]) for im in imgs_u8]
        pads = [self.pad.padded(h, w) if self.pad is not None else (h, w) for h, w in shapes]
        Hp, Wp = max(p[0] for p in pads), max(p[1] for p in pads)
        H, W = shapes[0]
        dev = imgs_u8[0].device
        na = self.norm.as_args()
        mean = (ctypes.c_float * 3)(*na['mean'])
        stdinv = (ctypes.c_float * 3)(*na['stdinv'])
        dt = 1 if self.dtype == torch.bfloat16 else 0
        mk = lambda: torch.empty((N, 3, Hp, Wp), dtype=self.dtype, device=dev,  # noqa: E731
                                 memory_format=torch.channels_last)
        out = dict()
        two_views = self.oamix is not None and self.oamix.num_views > 1
        if two_views:
            # both views in one allocation, originals first: integrate_data's torch.cat([img, img2]) (base.py:22-48) becomes
            # a view of it (detectors.integrate_data) instead of a 100 MB copy per step
            both = torch.empty((2 * N, 3, Hp, Wp), dtype=self.dtype, device=dev, memory_format=torch.channels_last)
            img = both[:N]
            img._oadg_batch = both
        else:
            img = mk()
        for i in range(N):   # physical NHWC slice i is contiguous
            check(L.oadg_oamix_normalize(ptr(imgs_u8[i].contiguous() if per_image else imgs_u8[i]), shapes[i][0],
                                         shapes[i][1], mean, stdinv, int(na['to_rgb']),
                                         ctypes.c_void_p(img.data_ptr() + i * img.stride(0) * img.element_size()),
                                         dt, Hp, Wp, stream_ptr()), 'oadg_oamix_normalize')
        out['img'] = img
        # the host copies of the gt boxes travel in img_metas: the random-proposal generator and OA-Mix need
        # them on the host, and reading them back from the device would stall the stream
        out['img_metas'] = [dict(img_shape=shapes[i] + (3,), pad_shape=pads[i] + (3,) if per_image else (Hp, Wp, 3),
                                 ori_shape=shapes[i] + (3,), scale_factor=1.0, flip=False,
                                 gt_bboxes_np=np.ascontiguousarray(gt_bboxes[i], dtype=np.float32))
                            for i in range(N)]
        if geo_meta is not None:
            for m, g in zip(out['img_metas'], geo_meta):
                m.update({k: v for k, v in g.items() if k != 'img_shape'})
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).pin_memory().to(  # noqa: E731
            dev, non_blocking=True)
        out['gt_bboxes'] = [up(b, np.float32) for b in gt_bboxes]
        out['gt_labels'] = [up(l, np.int64) for l in gt_labels]
        if self.oamix is not None and self.oamix.num_views > 1:
            assert self.oamix.num_views == 2 and self.oamix.keep_orig
            om = self.oamix
            # saliency / mask profiles of every image are enqueued first, so the scores of image i are on the
            # host long before its object-aware mixing step needs them
            if BATCH_STATES and not per_image and N > 1 and torch.is_tensor(imgs_u8) and imgs_u8.is_cuda and \
                    imgs_u8.dim() == 4 and imgs_u8.is_contiguous() and sum(len(g) for g in gt_bboxes) <= BATCH_STATES_MAX_BOXES:
                # one upload / profile launch / saliency launch triple for the boxes of the whole batch (few boxes per image:
                # their launches are 20 workgroups each, pure latency; with thousands of boxes per image the per-image launches
                # are full-size AND let the host plan image 0 while the device still scores image 7 - config 5: 36.3 ms per
                # step per image, 42.6 batched)
                states = _ImageState.batch(imgs_u8, gt_bboxes, om.spatial_ratio, om.sigma_ratio)
            else:
                states = [_ImageState(imgs_u8[i].contiguous(), gt_bboxes[i], om.spatial_ratio, om.sigma_ratio)
                          for i in range(N)]              # (each with its own H x W)
            img2 = both[N:]
            views = [ctypes.c_void_p(img2.data_ptr() + i * img2.stride(0) * img2.element_size()) for i in range(N)]
            if self.oamix_workers > 1 and N > 1:
                ml, oa = self._oamix_parallel(states, views, na, (Hp, Wp))
            elif N > 1 and _oa_mix.LOCKSTEP:
                # the per-box chains of the batch's images advance level by level TOGETHER (OAMix.oamix_many)
                hist = om.oamix_many(states, [_PtrView(v, self.dtype) for v in views], na, (Hp, Wp))
                ml = [torch.from_numpy(np.asarray(h['random_box_list'])) for h in hist]
                oa = [torch.from_numpy(np.stack(h['oa_random_box_list'], axis=0)) for h in hist]
            else:
                ml, oa = [], []
                for i, st in enumerate(states):
                    om._history = {}
                    om.oamix(st, out_u8=None, out_norm=_PtrView(views[i], self.dtype), norm=na, pad_shape=(Hp, Wp))
                    ml.append(torch.from_numpy(np.asarray(om._history['random_box_list'])))
                    oa.append(torch.from_numpy(np.stack(om._history['oa_random_box_list'], axis=0)))
            out['img2'] = img2
            out['gt_bboxes2'] = [b.clone() for b in out['gt_bboxes']]
            out['multilevel_boxes'] = ml
            out['oamix_boxes'] = oa
        return {k: v for k, v in out.items() if k in self.keys or k == 'img_metas'}


def _oamix_parallel(self, states, views, na, pad_shape):
    """OA-Mix of the batch's images on ``oamix_workers`` helper threads.  Helper k owns a HIP stream, an OAMix clone with
    its own scratch buffers and a ``RandomState`` seeded from the caller's stream when the helpers are created; it takes
    the images i = k (mod workers), in order, after an event recorded behind the (already enqueued) image states, and
    the caller's stream waits for every helper's event before the batch is handed on.  The host side of a
    ``bboxes_only_*`` op is mostly two C calls (plan + level launches) that run without the interpreter lock, so the
    helpers overlap: BASELINE configs[4] (4096 boxes per image) is host-bound at one worker.  Draw-for-draw
    reproducibility against a single numpy stream holds for one worker only - as with DataLoader workers, each helper's
    stream is private."""
    import copy
    from concurrent.futures import ThreadPoolExecutor
    from .oa_mix import rng, use_random_state
    dev = states[0].img.device
    K = min(self.oamix_workers, len(states))
    if self._helpers is None or len(self._helpers) < K:
        self._helpers = []
        for k in range(self.oamix_workers):
            om = copy.copy(self.oamix)
            om._bufs, om._history = {}, {}
            seed = int(rng.randint(0, 2 ** 31 - 1))

            def init(seed=seed):
                torch.cuda.set_device(dev)
                use_random_state(np.random.RandomState(seed))
            self._helpers.append(dict(om=om, stream=torch.cuda.Stream(device=dev),
                                      pool=ThreadPoolExecutor(1, thread_name_prefix=f'oadg-oamix-{k}', initializer=init)))
    cur = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(cur)
    results = [None] * len(states)

    lockstep = _oa_mix.LOCKSTEP
    recs = [None] * len(states)

    def task(k):
        h = self._helpers[k]
        om = h['om']
        # counters: a private dict per helper, merged by the caller once the futures are done (the shared dict's
        # read-modify-write updates raced between helper threads)
        om.stats = {} if self.oamix.stats is not None else None
        with torch.cuda.stream(h['stream']):
            h['stream'].wait_event(ready)
            for i in range(k, len(states), K):
                om._history = {}
                out_norm = _PtrView(views[i], self.dtype)
                if lockstep:
                    # round 4: the helper PLANS its images (draws, C plan calls, descriptor uploads on its stream); the
                    # caller then runs the recorded device commands of the whole batch in lockstep (OAMix.execute)
                    recs[i], hist = om.record(states[i], i, out_norm, na, pad_shape)
                else:
                    om.oamix(states[i], out_u8=None, out_norm=out_norm, norm=na, pad_shape=pad_shape)
                    hist = om._history
                results[i] = (torch.from_numpy(np.asarray(hist['random_box_list'])),
                              torch.from_numpy(np.stack(hist['oa_random_box_list'], axis=0)))
            ev = torch.cuda.Event()
            ev.record()
        return ev
    futs = [self._helpers[k]['pool'].submit(task, k) for k in range(K)]
    for f in futs:
        cur.wait_event(f.result())
    if lockstep:
        self.oamix.execute(recs)
    if self.oamix.stats is not None:
        for k in range(K):
            for key, n in (self._helpers[k]['om'].stats or {}).items():
                self.oamix.stats[key] = self.oamix.stats.get(key, 0) + n
    return [r[0] for r in results], [r[1] for r in results]


DevicePipeline._oamix_parallel = _oamix_parallel


class _PrefetchedFuture:
    """A batch whose pipeline is being enqueued by the worker thread."""

    def __init__(self, future):
        self.future = future

    def get(self):
        return _Prefetched(*self.future.result()).get()


class _Prefetched:
    """A batch being produced on the pipeline's side stream."""

    def __init__(self, out, event):
        self.out, self.event = out, event

    def get(self):
        """Make the current stream wait for the batch and hand its tensors over to it."""
        cur = torch.cuda.current_stream()
        cur.wait_event(self.event)
        for v in self.out.values():
            for t in (v if isinstance(v, (list, tuple)) else [v]):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)
        return self.out


class _PtrView:
    """a (device pointer, dtype) pair quacking enough like a tensor for ptr()/dtype checks"""

    def __init__(self, p, dtype):
        self._p, self.dtype = p, dtype

    def data_ptr(self):
        return self._p.value


@DATASETS.register_module()
class SyntheticCityscapes:
    """Cityscapes-shaped synthetic samples: uint8 low-pass noise images resident in HBM, ~20
    boxes per image with sides U(24,400) clipped to the image, labels U{0..7}.  Seeded per (seed, index)."""

    def __init__(self, img_shape=(1024, 2048), num_boxes=20, num_classes=8, length=2975, pipeline=None,
                 box_size=(24, 400), seed=0, device='cuda', test_mode=False):
        self.img_shape, self.num_boxes, self.num_classes = tuple(img_shape), num_boxes, num_classes
        self.length, self.box_size, self.seed, self.device = length, box_size, seed, device
        self.pipeline_cfg = pipeline

    def __len__(self):
        return self.length

    def boxes(self, idx):
        rs = np.random.RandomState((self.seed * 1000003 + idx) & 0x7fffffff)
        H, W = self.img_shape
        n = self.num_boxes
        bw = rs.uniform(self.box_size[0], min(self.box_size[1], W), n)
        bh = rs.uniform(self.box_size[0], min(self.box_size[1], H), n)
        x1 = rs.uniform(0, W - bw)
        y1 = rs.uniform(0, H - bh)
        b = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
        return b, rs.randint(0, self.num_classes, n).astype(np.int64)

    def image(self, idx, k=8):
        """uint8 [H,W,3]: uniform noise, k x k box filter, stretched to 0..255 (on the device)."""
        H, W = self.img_shape
        g = torch.Generator(device=self.device).manual_seed((self.seed * 1000003 + idx) & 0x7fffffff)
        x = torch.rand((1, 3, H + k - 1, W + k - 1), device=self.device, generator=g)
        x = torch.nn.functional.avg_pool2d(x, k, stride=1)
        lo, hi = x.amin(), x.amax()
        x = ((x - lo) / (hi - lo + 1e-9) * 255.0).clamp(0, 255).to(torch.uint8)
        return x[0].permute(1, 2, 0).contiguous()

    def batch(self, indices):
        imgs = torch.stack([self.image(i) for i in indices])
        bl = [self.boxes(i) for i in indices]
        return imgs, [b for b, _ in bl], [l for _, l in bl]
