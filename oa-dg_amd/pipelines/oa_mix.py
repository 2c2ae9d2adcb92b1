"""OA-Mix on the MI355X under the reference's PIPELINES name ``OAMix``.

Mirrors mmdet/datasets/pipelines/oa_mix.py:32-312 (and the leaf/wrapper functions of augmix.py:32-212,
bbox_augmentation.py:31-118,240-302 it dispatches to): same constructor, same ``__call__(results) -> results``
contract, same keys written (``img2``, ``gt_bboxes2``, ``oamix_boxes``, ``multilevel_boxes``, ``img_fields``,
``custom_field``), and the same consumption of the GLOBAL ``np.random`` stream, draw for draw.

What differs is where the pixels are touched: the host only draws random numbers and fills small launch
descriptors; every image byte is produced by the HIP kernels of csrc/oamix.hip through the C ABI.  The n_gt
full-resolution float masks of the reference (25 MB each) are replaced by two 1-D profiles per box.
"""
import ctypes
import math
import os

import numpy as np


class _ActiveRandomState:
    """The random stream OA-Mix draws from: numpy's GLOBAL RandomState (``np.random.*``, like the reference, so a
    seeded run replays it draw for draw) unless the calling thread installed its own with :func:`use_random_state`
    (the pipeline's prefetch worker does, exactly like a DataLoader worker process owns a private stream)."""

    def __getattr__(self, name):
        return getattr(getattr(_TLS, 'rs', None) or np.random.mtrand._rand, name)


import threading  # noqa: E402

_TLS = threading.local()
rng = _ActiveRandomState()


def use_random_state(rs):
    """Route this thread's OA-Mix draws to ``rs`` (a ``np.random.RandomState``); ``None`` = the global stream."""
    _TLS.rs = rs
import torch

from .. import _lib
from .._lib import (OP_BG_WARP, OP_IMAGE, OP_LUT_AUTOCONTRAST, OP_LUT_EQUALIZE, OP_POSTERIZE, OP_SOLARIZE,
                    OP_WARP_NEG, RegionOp, check, ptr, stream_ptr)
from ..core.bbox import bbox_overlaps_np
from ..registry import PIPELINES

AUG_LISTS = {   # oa_mix.py:15-29
    'augmix': ['autocontrast', 'equalize', 'posterize', 'solarize', 'bboxes_only_rotate',
               'bboxes_only_shear_xy', 'bboxes_only_translate_xy', 'bg_only_rotate', 'bg_only_shear_xy',
               'bg_only_translate_xy'],
    'augmix.all': ['autocontrast', 'equalize', 'posterize', 'solarize', 'invert', 'color', 'contrast',
                   'brightness', 'sharpness', 'bboxes_only_rotate', 'bboxes_only_shear_xy',
                   'bboxes_only_translate_xy', 'bg_only_rotate', 'bg_only_shear_xy', 'bg_only_translate_xy'],
}
MIX_TARGET_DTYPE = np.dtype([('fg_index', '<i4'), ('rect', '<i4', (4,)), ('m_oa', '<f4')])   # oadg_mix_target
BBOX_STEP_DTYPE = np.dtype([('minv', '<f8', (6,)), ('rect', '<i4', (4,)), ('row', '<i4'), ('pad_', '<i4'),
                            ('scratch_off', '<i8')])                                          # oadg_bbox_step
BBOX_CHAIN_DTYPE = np.dtype([('img', '<u8'), ('steps_dev', '<u8'), ('tile_prefix_dev', '<u8'), ('level_first_host', '<u8'),
                             ('tile_prefix_host', '<u8'), ('My', '<u8'), ('Mx', '<u8'), ('scratch', '<u8'), ('H', '<i4'),
                             ('W', '<i4'), ('n_levels', '<i4'), ('pad_', '<i4')])                        # oadg_bbox_chain
LOCKSTEP = os.environ.get('OADG_OAMIX_LOCKSTEP', '1') == '1'    # the images of a batch advance their per-box chains together
LANES = int(os.environ.get('OADG_OAMIX_LANES', '3'))    # buffer sets per image inside the lockstep pass (one per mixture
                                                        # chain; 1: the chains share a set and follow each other)
BATCH_BOXES = True      # bboxes_only_*: all boxes of an image in 2 launches per dependency level (False: 2 per box)
PLAN_IN_C = os.environ.get('OADG_OAMIX_PLAN_C', '1') == '1'     # the op's host arithmetic in one C call (else numpy)
PLAN_THREADS = int(os.environ.get('OADG_OAMIX_PLAN_THREADS', '4'))    # planner threads of the lockstep pass (0: none)
MIX_TILES_MIN_TARGETS = 64      # from this many mixing targets on object_aware_mixing visits tile-binned target lists
UNION_RECTS_MIN_BOXES = 64      # from this many boxes on the fg-mask union is built from the masks' support rects
ASYNC_PLAN_MIN_BOXES = 512      # below this a plan call takes less than the hand-over to a thread
_PLAN_POOL = None


def _ids(tensors):
    """resource ids of a command's operands (the buffers' addresses)"""
    return tuple(t.data_ptr() for t in tensors if t is not None)


def _plan_pool():
    global _PLAN_POOL
    if _PLAN_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _PLAN_POOL = ThreadPoolExecutor(max(PLAN_THREADS, 1), thread_name_prefix='oadg-oamix-plan')
    return _PLAN_POOL


def dependency_levels(rects, minvs, H, W):
    """Level of each bbox-only step such that running the levels in order, and the steps of one level in ANY order,
    gives the result of running all steps in list order (bbox_augmentation.py:74-88 is sequential): step j goes after
    every earlier step i whose written rect meets j's read footprint (read after write) or whose read footprint
    meets j's written rect (write after read).  rects: [n,4] x0,y0,w,h (the mask support = the pixels written);
    the read footprint = the rect itself plus the bounding box of its warped corners (affine, so the corners bound
    it) widened by the bilinear taps and the 1/32-px coordinate rounding, clipped to the image."""
    n = len(rects)
    if n == 0:
        return np.zeros((0,), np.int64)
    r = np.asarray(rects, np.float64)
    m = np.asarray(minvs, np.float64).reshape(n, 6)
    x0, y0, x1, y1 = r[:, 0], r[:, 1], r[:, 0] + r[:, 2] - 1, r[:, 1] + r[:, 3] - 1      # inclusive pixel corners
    cx = np.stack([x0, x1, x0, x1], 1)
    cy = np.stack([y0, y0, y1, y1], 1)
    sx = m[:, 0:1] * cx + m[:, 1:2] * cy + m[:, 2:3]
    sy = m[:, 3:4] * cx + m[:, 4:5] * cy + m[:, 5:6]
    fx0 = np.clip(np.minimum(np.floor(sx.min(1)) - 2, x0), 0, W - 1)
    fy0 = np.clip(np.minimum(np.floor(sy.min(1)) - 2, y0), 0, H - 1)
    fx1 = np.clip(np.maximum(np.ceil(sx.max(1)) + 2, x1), 0, W - 1)
    fy1 = np.clip(np.maximum(np.ceil(sy.max(1)) + 2, y1), 0, H - 1)
    level = np.zeros((n,), np.int64)
    for j in range(1, n):
        raw = (x0[:j] <= fx1[j]) & (x1[:j] >= fx0[j]) & (y0[:j] <= fy1[j]) & (y1[:j] >= fy0[j])
        war = (fx0[:j] <= x1[j]) & (fx1[:j] >= x0[j]) & (fy0[:j] <= y1[j]) & (fy1[:j] >= y0[j])
        hit = raw | war
        if hit.any():
            level[j] = level[:j][hit].max() + 1
    return level


def sample_level(n):            # augmix.py:60-61
    return rng.uniform(low=0.1, high=n)


def int_parameter(level, maxval):   # augmix.py:32-43
    return int(level * maxval / 10)


def float_parameter(level, maxval):  # augmix.py:46-57
    return float(level) * maxval / 10.


def invert_affine(M):
    """cv2.warpAffine inverts the forward matrix in double precision before sampling (imgwarp.cpp)."""
    m = [float(v) for v in np.asarray(M, dtype=np.float64).reshape(-1)]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def rotation_matrix(center, angle, scale=1.0):
    """cv2.getRotationMatrix2D (centre is a Point2f)."""
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))
    a = angle * math.pi / 180.0
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy],
                     [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)


def geo_matrix(kind, severity, img_size, center=None, size_for_level=None):
    """The affine matrix one geometric leaf draws (augmix.py:83-188), in the dtype cv2 receives it."""
    if kind == 'rotate':
        deg = int_parameter(sample_level(severity), 30)
        if rng.uniform() > 0.5:
            deg = -deg
        c = center if center is not None else (img_size[0] / 2, img_size[1] / 2)
        return rotation_matrix(c, deg)
    if kind in ('shear_x', 'shear_y'):
        lvl = float_parameter(sample_level(severity), 0.3)
        if rng.uniform() > 0.5:
            lvl = -lvl
        if kind == 'shear_x':
            tx = 0 if center is None else -lvl * center[1]
            return np.float32([[1, -lvl, -tx], [0, 1, 0]])
        ty = 0 if center is None else -lvl * center[0]
        return np.float32([[1, 0, 0], [-lvl, 1, -ty]])
    ax = 0 if kind == 'translate_x' else 1
    maxval = img_size[ax] if size_for_level is None else size_for_level[ax]
    lvl = int_parameter(sample_level(severity), maxval / 3)
    if rng.random() > 0.5:
        lvl = -lvl
    return np.float32([[1, 0, -lvl], [0, 1, 0]]) if ax == 0 else np.float32([[1, 0, 0], [0, 1, -lvl]])


class _PinnedRing:
    """Per-thread ring of pinned staging buffers for the small descriptor uploads: ``tensor.pin_memory()`` allocates
    (hipHostMalloc, ~2 ms for a 4096-box descriptor table) on every call; a slot of the ring is reused once the copy
    that read it has completed (event recorded behind the copy on its stream)."""
    SLOTS = 96

    def __init__(self):
        self.bufs, self.events, self.k = [None] * self.SLOTS, [None] * self.SLOTS, 0

    def slot(self, nbytes):
        """(slot number, pinned uint8 buffer of >= nbytes) - the caller fills it, enqueues the copy and records the slot's
        event behind it"""
        k, self.k = self.k, (self.k + 1) % self.SLOTS
        if self.events[k] is not None:
            self.events[k].synchronize()
        buf = self.bufs[k]
        if buf is None or buf.numel() < nbytes:
            buf = self.bufs[k] = torch.empty((max(4096, 1 << int(max(nbytes, 2) - 1).bit_length()),), dtype=torch.uint8,
                                             pin_memory=True)
        return k, buf

    def stage(self, raw):
        k, self.k = self.k, (self.k + 1) % self.SLOTS
        if self.events[k] is not None:
            self.events[k].synchronize()
        buf = self.bufs[k]
        if buf is None or buf.numel() < raw.size:
            buf = self.bufs[k] = torch.empty((max(4096, 1 << int(raw.size - 1).bit_length()),), dtype=torch.uint8,
                                             pin_memory=True)
        buf.numpy()[:raw.size] = raw
        return k, buf[:raw.size]


def _upload(arr, device):
    """small host array -> device tensor (same dtype / shape) without blocking the host on the stream"""
    arr = np.ascontiguousarray(arr)
    ring = getattr(_TLS, 'ring', None)
    if ring is None:
        ring = _TLS.ring = _PinnedRing()
    if arr.size == 0:
        return torch.from_numpy(arr).to(device)
    k, src = ring.stage(arr.view(np.uint8).reshape(-1))
    dst = torch.empty((src.numel(),), dtype=torch.uint8, device=device)
    dst.copy_(src, non_blocking=True)
    ev = ring.events[k] = ring.events[k] or torch.cuda.Event()
    ev.record()
    return dst.view(torch.from_numpy(arr[:0]).dtype).view(arr.shape)


class _ImageState:
    """Device-side state of one image: profiles of the fg masks, their union, saliency scores (async)."""

    @classmethod
    def batch(cls, imgs, gt_bboxes_list, spatial_ratio, sigma_ratio):
        """the states of the images of ONE contiguous uint8 [N,H,W,3] batch (round 6): the boxes of all images go through
        one upload, ONE mask-profile launch and ONE saliency launch triple (per image: 3 uploads, 5 launches and a
        device -> host copy of their own - 20 workgroups per launch at 20 boxes per image); every image's profiles / scores
        are row ranges of the batch's tensors.  Same values as N separate states."""
        L = _lib.lib()
        N, H, W = int(imgs.shape[0]), int(imgs.shape[1]), int(imgs.shape[2])
        dev = imgs.device
        states = [cls.__new__(cls) for _ in range(N)]
        rows, nb = [], 0
        for i, st in enumerate(states):
            st._host_part(imgs[i], gt_bboxes_list[i], spatial_ratio, sigma_ratio)
            rows.append(nb)
            nb += max(st.n, 1)
        My = torch.empty((nb, H), dtype=torch.float32, device=dev)
        Mx = torch.empty((nb, W), dtype=torch.float32, device=dev)
        tot = sum(st.n for st in states)
        if tot:
            # boxes in PROFILE-ROW order (an image without boxes still owns one unused row): qbox, sigma, integer boxes, image id
            qb = np.zeros((nb, 4), np.int32)
            sg = np.zeros((nb, 2), np.float64)
            for st, r in zip(states, rows):
                qb[r:r + st.n], sg[r:r + st.n] = st._qbox_host, st._sigma_host
            qbox_dev, sigma_dev = _upload(qb, dev), _upload(sg, dev)
            check(L.oadg_oamix_box_profiles(ptr(qbox_dev), ptr(sigma_dev), nb, H, W, spatial_ratio, ptr(My), ptr(Mx),
                                            stream_ptr()), 'oadg_oamix_box_profiles')
        for st, r in zip(states, rows):
            k = max(st.n, 1)
            st.My, st.Mx = My[r:r + k], Mx[r:r + k]
            st._keep = (My, Mx)
            st._union()
        if tot:
            ib = np.concatenate([np.array(st.gt, dtype=np.int32).reshape(-1, 4) for st in states], 0)
            img_of = np.concatenate([np.full((st.n,), i, np.int32) for i, st in enumerate(states)])
            ib_dev, of_dev = _upload(ib, dev), _upload(img_of, dev)
            scores_dev = torch.empty((tot,), dtype=torch.float64, device=dev)
            nbytes = L.oadg_oamix_saliency_workspace_bytes(tot)
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
            check(L.oadg_oamix_saliency_batch(ptr(imgs), int(imgs.stride(0)), ptr(of_dev), H, W, ptr(ib_dev), tot, spatial_ratio,
                                              ptr(scores_dev), ptr(ws), nbytes, stream_ptr()), 'oadg_oamix_saliency_batch')
            host = torch.empty((tot,), dtype=torch.float64, pin_memory=True)
            host.copy_(scores_dev, non_blocking=True)
            evt = torch.cuda.Event()
            evt.record()
            off = 0
            for st in states:
                st._scores_host, st._scores_evt = host[off:off + st.n], evt
                st._keep += (ib_dev, of_dev, scores_dev, ws, qbox_dev, sigma_dev)
                off += st.n
        return states

    def __init__(self, img, gt_bboxes, spatial_ratio, sigma_ratio):
        L = _lib.lib()
        self._host_part(img, gt_bboxes, spatial_ratio, sigma_ratio)
        n, H, W, dev = self.n, self.H, self.W, img.device
        self.My = torch.empty((max(n, 1), H), dtype=torch.float32, device=dev)
        self.Mx = torch.empty((max(n, 1), W), dtype=torch.float32, device=dev)
        if n:
            self._qbox = _upload(self._qbox_host, dev)
            self._sigma = _upload(self._sigma_host, dev)
            check(L.oadg_oamix_box_profiles(ptr(self._qbox), ptr(self._sigma), n, H, W, spatial_ratio,
                                            ptr(self.My), ptr(self.Mx), stream_ptr()), 'oadg_oamix_box_profiles')
        self._union()
        # --- saliency scores (oa_mix.py:98-111): launched now, read when object-aware mixing needs them -----
        if n:
            ib = np.array(self.gt, dtype=np.int32)
            self._ibox = _upload(ib, dev)
            self._scores_dev = torch.empty((n,), dtype=torch.float64, device=dev)
            nb = L.oadg_oamix_saliency_workspace_bytes(n)
            self._sal_ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
            check(L.oadg_oamix_saliency(ptr(img), H, W, ptr(self._ibox), n, spatial_ratio,
                                        ptr(self._scores_dev), ptr(self._sal_ws), nb, stream_ptr()),
                  'oadg_oamix_saliency')
            self._scores_host = torch.empty((n,), dtype=torch.float64, pin_memory=True)
            self._scores_host.copy_(self._scores_dev, non_blocking=True)
            self._scores_evt = torch.cuda.Event()
            self._scores_evt.record()

    def _union(self):
        """union of the blurred fg masks (oa_mix.py:95-120 / bbox_augmentation.py:240-302's np.max over the masks)"""
        L = _lib.lib()
        n, H, W, dev = self.n, self.H, self.W, self.img.device
        self.union_f = torch.empty((H, W), dtype=torch.float32, device=dev)
        self.union_u8 = torch.empty((H, W), dtype=torch.uint8, device=dev)
        self._rects = None
        if n >= UNION_RECTS_MIN_BOXES:
            # many boxes: the union from the masks' support rects (work ~ rect areas, not n x H x W); byte-identical
            check(L.oadg_oamix_fg_union_rects(ptr(self.My), ptr(self.Mx), ptr(self.rects_dev()), n, H, W, ptr(self.union_f),
                                              ptr(self.union_u8), stream_ptr()), 'oadg_oamix_fg_union_rects')
        else:
            check(L.oadg_oamix_fg_union(ptr(self.My), ptr(self.Mx), n, H, W, ptr(self.union_f), ptr(self.union_u8),
                                        stream_ptr()), 'oadg_oamix_fg_union')

    def _host_part(self, img, gt_bboxes, spatial_ratio, sigma_ratio):
        """everything of the state that is host arithmetic on the boxes"""
        self.img = img                                     # uint8 [H,W,3] on the device
        self.H, self.W = int(img.shape[0]), int(img.shape[1])
        self.gt = np.asarray(gt_bboxes, dtype=np.float32).reshape(-1, 4)
        n = self.n = self.gt.shape[0]
        self._scores = None
        H, W = self.H, self.W
        # --- blurred-mask profiles (oa_mix.py:74-93) -------------------------------------------------
        # (vectorised over the boxes: per box  x1, y1, x2, y2 = np.array(gt // ratio, dtype=np.int32);
        #  sx = (x2 - x1) * sigma_ratio / 3 * 2  as in oa_mix.py:83-90)
        qbox = np.array(self.gt // spatial_ratio, dtype=np.int32).reshape(n, 4)
        sxy = np.stack([(qbox[:, 2] - qbox[:, 0]) * sigma_ratio / 3 * 2,
                        (qbox[:, 3] - qbox[:, 1]) * sigma_ratio / 3 * 2], 1).astype(np.float64).reshape(n, 2)
        blur = ~((sxy[:, 0] <= 0) | (sxy[:, 1] <= 0))
        self._qbox_host, self._sigma_host = qbox, np.where(blur[:, None], sxy, 0.0)
        # conservative rect per box where its mask can be non-zero: arrays now (rows int64 [n, 4], empty bool [n]), the list
        # of tuples / None the per-box paths index only when one of them asks (``support``: 3 ms of tolist() at 4096 boxes)
        self._support_rows, self._support_empty = self._supports(qbox, sxy, blur, spatial_ratio)
        self._support_list = None
        assert (qbox >= 0).all(), 'gt boxes must have non-negative coordinates'

    def rects_dev(self):
        """device int32 [n, 4]: (x0, y0, w, h) of every fg mask's support, zeros for an empty mask (uploaded once)"""
        if self._rects is None:
            n = self.n
            rows = self._support_rows.reshape(n, 4)
            ok = ~self._support_empty.reshape(n) & (rows[:, 2] > 0) & (rows[:, 3] > 0)
            self._rects = _upload(np.where(ok[:, None], rows, 0).astype(np.int32).reshape(max(n, 0), 4), self.img.device)
        return self._rects

    def _supports(self, qbox, sxy, blur, ratio):
        """per box (x0, y0, w, h) of the pixels where its (blurred, x4 resized) mask can be non-zero, None if empty"""
        H, W = self.H, self.W
        x1, y1, x2, y2 = (qbox[:, k].astype(np.int64) for k in range(4))
        rad = lambda s_: np.where(blur, ((np.rint(s_ * 4 * 2 + 1).astype(np.int64) | 1) - 1) // 2, 0)  # noqa: E731
        rx, ry = rad(sxy[:, 0]), rad(sxy[:, 1])
        xa = np.maximum(0, ratio * (x1 - rx) - 2 * ratio)
        xb = np.minimum(W, ratio * (x2 + rx) + 2 * ratio)
        ya = np.maximum(0, ratio * (y1 - ry) - 2 * ratio)
        yb = np.minimum(H, ratio * (y2 + ry) + 2 * ratio)
        empty = (x2 <= x1) | (y2 <= y1)                      # empty at reduced resolution: mask is all zero
        return np.stack([xa, ya, xb - xa, yb - ya], 1), empty

    @property
    def support(self):
        """per box (x0, y0, w, h) or None (see _supports)"""
        if self._support_list is None:
            self._support_list = [None if e else tuple(r)
                                  for e, r in zip(self._support_empty.tolist(), self._support_rows.tolist())]
        return self._support_list

    def plan_arrays(self):
        """(ib int64 [n, 4], support int32 [n, 4] with zeros for empty masks, number of boxes that draw) - the per-image
        constants of oadg_oamix_bbox_plan, built once"""
        c = getattr(self, '_plan_arrays', None)
        if c is None:
            ib = np.ascontiguousarray(self.gt.astype(np.int64))            # int() truncation of non-negative float32
            rows = self._support_rows.reshape(self.n, 4)
            ok = ~self._support_empty.reshape(self.n) & (rows[:, 2] > 0) & (rows[:, 3] > 0)
            sup = np.ascontiguousarray(np.where(ok[:, None], rows, 0).astype(np.int32))
            draws = int((~(((ib[:, 2] - ib[:, 0]) < 1) | ((ib[:, 3] - ib[:, 1]) < 1))).sum()) if self.n else 0
            c = self._plan_arrays = (ib, sup, draws)
        return c

    def scores(self):
        """fg_score_list of get_fg_regions: -1 for boxes thinner than spatial_ratio, else the saliency mean."""
        if self._scores is None:
            if self.n == 0:
                self._scores = []
            else:
                self._scores_evt.synchronize()
                self._scores = [float(s) if s >= 0 else -1 for s in self._scores_host.tolist()]
        return self._scores


@PIPELINES.register_module()
class OAMix:

    def __init__(self, version='augmix', num_views=2, keep_orig=True, severity=10, mixture_width=3,
                 mixture_depth=-1, random_box_scale=(0.01, 0.1), random_box_ratio=(3, 1 / 3),
                 oa_random_box_scale=(0.005, 0.1), oa_random_box_ratio=(3, 1 / 3), num_bboxes=(3, 5),
                 spatial_ratio=4, sigma_ratio=0.3, **kwargs):
        if version not in AUG_LISTS:
            raise NotImplementedError(version)
        self.version, self.aug_list = version, AUG_LISTS[version]
        self.num_views, self.keep_orig, self.severity = num_views, keep_orig, severity
        self.aug_prob_coeff, self.mixture_width, self.mixture_depth = 1.0, mixture_width, mixture_depth
        self.random_box_scale, self.random_box_ratio = random_box_scale, random_box_ratio
        self.oa_random_box_scale, self.oa_random_box_ratio = oa_random_box_scale, oa_random_box_ratio
        self.score_thresh = 10
        self.spatial_ratio, self.sigma_ratio = spatial_ratio, sigma_ratio
        self._history = {}
        self.kwargs = kwargs            # unknown kwargs are swallowed like the reference (oa_mix.py:72)
        self._bufs = {}
        self._rec = None                # a list while oamix_many() records an image's device commands
        self._lane = 0                  # which private buffer set the commands being recorded work on (see _buffers)
        self._pending_plans = 0         # plans handed to planner threads whose staging slots are still taken
        self.trace = None               # set to [] to record the op sequence (tests)
        self.stats = None               # set to {} to count compose steps / bbox-step pixels (tools/bench_oamix.py)

    # ------------------------------------------------------------------------------------------ lockstep
    def _do(self, fn, reads=(), writes=()):
        """a device command of the current image: now, or - inside oamix_many() - appended to the image's command list
        together with the buffers it reads / writes (tensors; ``execute`` orders commands by them)"""
        if self._rec is None:
            fn()
        else:
            self._rec.append(('call', fn, _ids(reads), _ids(writes)))

    def oamix_many(self, states, out_norms, norm, pad_shape):
        """``oamix`` for the images of a batch with their ``bboxes_only_*`` chains advanced in LOCKSTEP.  The host side runs
        image by image exactly as before (same draws from the same stream in the same order; uploads are issued right
        away), but the kernels that touch an image's work buffers are recorded per image and then issued round by round:
        every image runs up to its next per-box chain, then level l of all those chains goes out as ONE launch pair
        (csrc oadg_oamix_bbox_chain_multi).  The images own disjoint buffers (``_buffers`` slots) and every image's own
        command order is kept, so the views are byte-identical to the sequential pass; launches per batch drop from
        2 x (sum of the chain depths) to 2 x (sum over rounds of the deepest chain).  Returns the per-image histories."""
        recs, hist = [], []
        for i, st in enumerate(states):
            rec, h = self.record(st, i, out_norms[i], norm, pad_shape)
            recs.append(rec)
            hist.append(h)
        self.execute(recs)
        return hist

    def record(self, st, slot, out_norm, norm, pad_shape):
        """the host half of ``oamix`` for one image: every draw, every plan, every upload - and the list of device commands
        it would have issued (``execute`` runs the lists of a batch in lockstep).  ``slot`` selects the image's private
        work buffers.  Returns (command list, history)."""
        st.slot = slot
        self._lane = 0
        self._history, self._rec = {}, []
        try:
            self.oamix(st, out_u8=None, out_norm=out_norm, norm=norm, pad_shape=pad_shape)
        finally:
            rec, self._rec = self._rec, None
        return rec, self._history

    def execute(self, recs):
        """issue the recorded command lists of a batch's images on the current stream.  Every command carries the buffers
        it reads and writes; a command may go out as soon as every EARLIER command of its image that touches one of them
        in a conflicting way has gone out (one stream: issue order = execution order), so commands on disjoint buffers -
        the three mixture chains of a view, the up to three region ops of a compose step, each on a buffer set of its own -
        overtake each other freely, while e.g. the accumulator still receives the chains' results in the reference's
        order.  All per-box chains that are ready at the same time - of all images, all mixture chains, all region ops -
        advance level by level TOGETHER in one csrc oadg_oamix_bbox_chain_multi call (round 6; rounds 4-5 batched one
        chain per image at a time: a third as many launch rounds now)."""
        import heapq
        L = _lib.lib()
        from .. import hip_ops
        graphs, heaps = [], []
        for rec in recs:
            nodes, last_w, readers = [], {}, {}
            for idx, cmd in enumerate(rec):
                reads, writes = cmd[2], cmd[3]
                deps = set()
                for r in reads:
                    if r in last_w:
                        deps.add(last_w[r])
                for w in writes:
                    if w in last_w:
                        deps.add(last_w[w])
                    deps.update(readers.get(w, ()))
                for d in deps:
                    nodes[d][3].append(idx)
                for r in reads:
                    readers.setdefault(r, []).append(idx)
                for w in writes:
                    last_w[w] = idx
                    readers[w] = []
                nodes.append([cmd[0], cmd[1], len(deps), []])        # kind, payload, open dependencies, successors
            graphs.append(nodes)
            heaps.append([i for i, nd in enumerate(nodes) if nd[2] == 0])        # (ascending = a valid heap)

        def complete(g, i):
            for s_ in graphs[g][i][3]:
                graphs[g][s_][2] -= 1
                if graphs[g][s_][2] == 0:
                    heapq.heappush(heaps[g], s_)

        left = sum(len(nodes) for nodes in graphs)
        while left:
            batch = []
            for g, heap in enumerate(heaps):
                while heap:
                    i = heapq.heappop(heap)
                    kind, payload = graphs[g][i][0], graphs[g][i][1]
                    if kind == 'plan':              # a plan that ran on a planner thread: its chain, or nothing
                        payload = payload()
                        kind = 'chain' if payload is not None else 'done'
                    if kind == 'chain':
                        batch.append((g, i, payload))
                        continue
                    if kind == 'call':
                        payload()
                    left -= 1
                    complete(g, i)
            if not batch:
                assert left == 0, 'OA-Mix command graph did not drain'
                break
            tab = np.zeros((len(batch),), BBOX_CHAIN_DTYPE)
            work = 0.0
            for r, (_, _, c) in zip(tab, batch):
                r['img'], r['steps_dev'], r['tile_prefix_dev'] = c['img'], c['steps_dev'], c['tile_prefix_dev']
                r['level_first_host'], r['tile_prefix_host'] = c['level_first'].ctypes.data, c['tile_prefix'].ctypes.data
                r['My'], r['Mx'], r['scratch'], r['H'], r['W'], r['n_levels'] = c['My'], c['Mx'], c['scratch'], c['H'], c['W'], c['n_levels']
                work += c['work']
            check(hip_ops._timed('oamix_bbox_chain', L.oadg_oamix_bbox_chain_multi, tab.ctypes.data_as(ctypes.c_void_p),
                                 len(batch), stream_ptr(), work=work), 'oadg_oamix_bbox_chain_multi')
            if self.stats is not None:
                self.stats['lockstep_rounds'] = self.stats.get('lockstep_rounds', 0) + 1
                self.stats['lockstep_levels'] = self.stats.get('lockstep_levels', 0) + int(tab['n_levels'].max())
                self.stats['lockstep_chains'] = self.stats.get('lockstep_chains', 0) + len(batch)
            for g, i, _ in batch:
                left -= 1
                complete(g, i)

    # ------------------------------------------------------------------------------------------ buffers
    def _buffers(self, st):
        """the work buffers of (image slot, lane): two ping-pong images, three region-op images with a scratch image each,
        histogram / LUT / grey-sum cells; the fp32 accumulator belongs to the slot (all lanes add into it).  Lane 0 is the
        only one outside the lockstep pass; inside it every mixture chain of a view records on a lane of its own, so that
        ``execute`` may run the chains' per-box levels side by side."""
        lane = self._lane
        key = (st.H, st.W, str(st.img.device), getattr(st, 'slot', 0), lane)
        b = self._bufs.get(key)
        if b is None:
            dev, H, W = st.img.device, st.H, st.W
            u8 = lambda: torch.empty((H, W, 3), dtype=torch.uint8, device=dev)  # noqa: E731
            sc = lambda: torch.empty((H * W * 3 + 4 * 8192 + 64,), dtype=torch.uint8, device=dev)  # noqa: E731
            b = dict(ping=[u8(), u8()], tmp=[u8(), u8(), u8()], scratch=[sc(), sc(), sc()],
                     hist=torch.empty((768,), dtype=torch.int32, device=dev),
                     luts=torch.empty((2 * 768,), dtype=torch.uint8, device=dev),
                     gray=torch.empty((1,), dtype=torch.int64, device=dev))
            if lane == 0:
                b['acc'] = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
            else:
                self._lane = 0
                try:
                    b['acc'] = self._buffers(st)['acc']
                finally:
                    self._lane = lane
            # bounded by BYTES: a lane's set is 24 bytes per pixel (eight uint8 images = 50 MB at 1024 x 2048),
            # the accumulator 12; per-sample multi-scale brings a new shape per image, so the oldest sets go once the cache
            # holds more than OADG_OAMIX_CACHE_MB (default 2048: the lockstep pass of a batch of eight with three lanes
            # each plus headroom).  A set still referenced by recorded commands stays alive through their closures.
            b['bytes'] = (24 + (12 if lane == 0 else 0)) * H * W
            limit = int(os.environ.get('OADG_OAMIX_CACHE_MB', 2048)) << 20
            while self._bufs and sum(v['bytes'] for v in self._bufs.values()) + b['bytes'] > limit:
                self._bufs.pop(next(iter(self._bufs)))
            self._bufs[key] = b
        return b

    # ------------------------------------------------------------------------------------------ regions
    def _random_regions(self, H, W, scale, ratio, num_bboxes, return_score=False, fg_boxes=None,
                        fg_scores=None, max_iters=50, eps=1e-6):
        """get_random_regions (oa_mix.py:122-184) without the H x W x 3 masks."""
        boxes, scores = [], []
        target = rng.randint(*num_bboxes) if isinstance(num_bboxes, tuple) else num_bboxes
        for _ in range(max_iters):
            if len(boxes) >= target:
                break
            x1, y1 = rng.randint(0, W), rng.randint(0, H)
            _scale = rng.uniform(*scale) * H * W
            _ratio = rng.uniform(*ratio)
            bw, bh = int(np.sqrt(_scale / _ratio)), int(np.sqrt(_scale * _ratio))
            if x1 + bw > W or y1 + bh > H:
                continue
            box = np.array([[x1, y1, min(x1 + bw, W), min(y1 + bh, H)]])
            if np.sum(bbox_overlaps_np(box, np.asarray(boxes).reshape(-1, 4))) > eps:
                continue
            if return_score:
                ious = bbox_overlaps_np(box, fg_boxes)
                final = float('inf')
                if np.sum(ious) > eps:          # min score over the overlapped fg boxes with w, h >= 1 (oa_mix.py:157-170)
                    ok = (ious[0] != 0.0) & ~((fg_boxes[:, 2] - fg_boxes[:, 0] < 1) | (fg_boxes[:, 3] - fg_boxes[:, 1] < 1))
                    if ok.any():
                        final = min(final, float(np.min(np.asarray(fg_scores, np.float64)[ok])))
                scores.append(final)
            boxes += list(box)
        return (boxes, scores) if return_score else boxes

    # ------------------------------------------------------------------------------------------ one op
    def _ensure_luts(self, st, src, step):
        if step.get('luts_for') is not src:
            L = _lib.lib()
            b = self._buffers(st)

            def run(src=src, b=b, n=st.H * st.W):
                check(L.oadg_oamix_hist(ptr(src), n, ptr(b['hist']), stream_ptr()), 'oadg_oamix_hist')
                check(L.oadg_oamix_luts(ptr(b['hist']), ptr(b['luts']), stream_ptr()), 'oadg_oamix_luts')
            self._do(run, reads=(src,), writes=(b['hist'], b['luts']))
            step['luts_for'] = src

    def _aug(self, st, src, step):
        """aug (oa_mix.py:264-279): draw the op, draw its parameters, return the region-op descriptor."""
        name = self.aug_list[rng.choice(len(self.aug_list))]
        if self.trace is not None:
            self.trace.append(name)
        op = RegionOp()
        H, W = st.H, st.W
        if name in ('autocontrast', 'equalize'):
            self._ensure_luts(st, src, step)
            op.kind = OP_LUT_AUTOCONTRAST if name == 'autocontrast' else OP_LUT_EQUALIZE
        elif name == 'posterize':
            bits = 4 - int_parameter(sample_level(self.severity), 4)
            op.kind, op.param = OP_POSTERIZE, (~(2 ** (8 - bits) - 1)) & 0xFF
        elif name == 'solarize':
            op.kind, op.param = OP_SOLARIZE, 256 - int_parameter(sample_level(self.severity), 256)
        elif name == 'invert':
            tx = 1 if rng.random() > 0.5 else -1
            ty = 1 if rng.random() > 0.5 else -1
            op.kind = OP_WARP_NEG
            op.minv[:] = invert_affine(np.float32([[1, 0, tx], [0, 1, ty]]))
        elif name in ('color', 'contrast', 'brightness', 'sharpness'):
            # PIL.ImageEnhance.X(img).enhance(factor), augmix.py:192-212
            op.minv[0] = float_parameter(sample_level(self.severity), 1.8) + 0.1
            op.kind = {'color': _lib.OP_ENH_COLOR, 'contrast': _lib.OP_ENH_CONTRAST,
                       'brightness': _lib.OP_ENH_BRIGHTNESS, 'sharpness': _lib.OP_ENH_SHARPNESS}[name]
            if name == 'contrast':
                if step.get('gray_for') is not src:
                    b = self._buffers(st)
                    self._do(lambda src=src, b=b, n=st.H * st.W: check(
                        _lib.lib().oadg_oamix_gray_sum(ptr(src), n, ptr(b['gray']), stream_ptr()), 'oadg_oamix_gray_sum'),
                        reads=(src,), writes=(b['gray'],))
                    step['gray_for'] = src
                op.image = self._buffers(st)['gray'].data_ptr()
        else:
            scope, kind = name.split('_only_')
            if kind.endswith('_xy'):
                kind = kind[:-2] + ('x' if rng.rand() < 0.5 else 'y')
            if scope == 'bg':
                op.kind = OP_BG_WARP
                op.minv[:] = invert_affine(geo_matrix(kind, self.severity, (W, H)))
            else:
                op.kind = OP_IMAGE
                op.image = self._bboxes_only(st, src, kind, step).data_ptr()
        return op

    def _bboxes_only(self, st, src, kind, step):
        """_apply_bboxes_only_augmentation (bbox_augmentation.py:74-88): the gt boxes, in order, each warp the
        whole current image about their centre and blend it in through their blurred mask."""
        L = _lib.lib()
        b = self._buffers(st)
        T = b['tmp'][step['n_tmp']]
        step['scratch'] = b['scratch'][step['n_tmp']]          # (a scratch image per region op: their chains may run side by side)
        step['n_tmp'] += 1
        self._do(lambda T=T, src=src: T.copy_(src), reads=(src,), writes=(T,))
        H, W = st.H, st.W
        if BATCH_BOXES and PLAN_IN_C:
            self._bbox_chain_c(st, T, kind, step)
        elif BATCH_BOXES:
            rows, rects, minvs = self._box_matrices(st, kind)
            if len(rows):
                self._bbox_chain(st, T, rows, rects, minvs, step)
        else:
            for i, box in enumerate(st.gt):
                x1, y1, x2, y2 = int(box[0]), int(box[1]), int(box[2]), int(box[3])
                if (x2 - x1) < 1 or (y2 - y1) < 1:
                    continue                                     # returns before any draw (:45-47)
                center = ((x1 + x2) / 2., (y1 + y2) / 2.)
                M = geo_matrix(kind, self.severity, (W, H), center, (x2 - x1 + 1, y2 - y1 + 1))
                sup = st.support[i]
                if sup is None or sup[2] <= 0 or sup[3] <= 0:
                    continue                                     # mask identically zero: image unchanged
                minv = (ctypes.c_double * 6)(*invert_affine(M))
                self._do(lambda minv=minv, sup=sup, i=i, scr=step['scratch']: check(
                    L.oadg_oamix_bbox_step(ptr(T), H, W, minv, sup[0], sup[1], sup[2], sup[3],
                                           ctypes.c_void_p(st.My.data_ptr() + 4 * i * H),
                                           ctypes.c_void_p(st.Mx.data_ptr() + 4 * i * W), ptr(scr),
                                           stream_ptr()), 'oadg_oamix_bbox_step'), writes=(T, step['scratch']))
        if self.stats is not None:
            self.stats['bbox_ops'] = self.stats.get('bbox_ops', 0) + 1
            sup_ = st.plan_arrays()[1].astype(np.int64)
            self.stats['bbox_px'] = self.stats.get('bbox_px', 0) + int((sup_[:, 2] * sup_[:, 3]).sum())
        return T

    _KIND_ID = dict(rotate=0, shear_x=1, shear_y=2, translate_x=3, translate_y=4)

    def _bbox_chain_c(self, st, T, kind, step):
        """_box_matrices + _bbox_chain with the host arithmetic in ONE C call (csrc/oamix_host.hip oadg_oamix_bbox_plan:
        draws -> matrices -> inverses -> dependency levels -> level-major step table, written into the pinned staging
        slot; ctypes releases the interpreter lock for its duration), one copy to the device, the level launches.
        Inside oamix_many() the call of an image with many boxes runs on a PLANNER THREAD (the draws are taken here, in
        stream order; nothing of the plan is needed before the batch's commands are executed): the ~0.8 ms per op of a
        4096-box image overlap the recording of the following ops and images."""
        L = _lib.lib()
        scratch = step['scratch']
        H, W = st.H, st.W
        ib, sup, m = st.plan_arrays()
        n = st.n
        r = rng.random_sample(2 * m) if m else np.zeros((0,))            # the boxes' (level, sign) draws, in box order
        if n == 0:
            return
        nbytes = L.oadg_oamix_bbox_plan_bytes(n)
        deferred = self._rec is not None and PLAN_THREADS > 0 and n >= ASYNC_PLAN_MIN_BOXES
        name = 'plan_ring' if deferred else 'ring'          # (deferred plans hold their slot until execute(): own ring)
        ring = getattr(_TLS, name, None)
        if ring is None:
            ring = _PinnedRing()
            setattr(_TLS, name, ring)
        if deferred and self._pending_plans >= ring.SLOTS - 1:
            deferred = False
            ring = getattr(_TLS, 'ring', None) or _PinnedRing()
            _TLS.ring = ring
        k, buf = ring.slot(nbytes)
        if deferred:
            lf = np.empty((n + 2,), np.int32)
        else:
            lf = getattr(_TLS, 'level_first', None)
            if lf is None or lf.size < n + 2:
                lf = _TLS.level_first = np.empty((max(n + 2, 64),), np.int32)
        out = (ctypes.c_int * 3)()
        area = ctypes.c_longlong(0)
        r = np.ascontiguousarray(r, np.float64)
        args = (self._KIND_ID[kind], float(self.severity), ib.ctypes.data, sup.ctypes.data, n, r.ctypes.data, int(r.size), H, W,
                buf.data_ptr(), nbytes, lf.ctypes.data, out, ctypes.byref(area))
        keep = (ib, sup, r, buf, lf, out, area)             # the call's operands stay alive until it has run

        def finish(rc):
            """the plan has run: statistics, the table's copy to the device, the chain's launch operands (or None)"""
            check(rc, 'oadg_oamix_bbox_plan')
            n_live, n_levels = int(out[0]), int(out[1])
            if self.stats is not None:
                self.stats['bbox_levels'] = self.stats.get('bbox_levels', 0) + n_levels
                self.stats['bbox_steps'] = self.stats.get('bbox_steps', 0) + n_live
            if n_live == 0:
                return None
            used = n_live * BBOX_STEP_DTYPE.itemsize + (n_live + 1) * 4
            dst = torch.empty((used,), dtype=torch.uint8, device=st.img.device)
            dst.copy_(buf[:used], non_blocking=True)
            ev = ring.events[k] = ring.events[k] or torch.cuda.Event()
            ev.record()
            step.setdefault('keepalive', []).append(dst)          # descriptor tensor lives until the step's launches ran
            tiles_off = n_live * BBOX_STEP_DTYPE.itemsize
            return dict(dst=dst, tiles_off=tiles_off, n_live=n_live, n_levels=n_levels, work=float(9 * area.value))

        def chain_rec(f):
            # lockstep: this chain waits for the other images of the batch (oamix_many).  The plan's host tables live in
            # buffers the next plan reuses (thread-local level list, pinned staging slot): private copies travel along
            dst, tiles_off, n_live, n_levels = f['dst'], f['tiles_off'], f['n_live'], f['n_levels']
            return dict(
                img=T.data_ptr(), H=H, W=W, steps_dev=dst.data_ptr(), tile_prefix_dev=dst.data_ptr() + tiles_off,
                level_first=lf[:n_levels + 1].copy(),
                tile_prefix=buf[tiles_off:tiles_off + (n_live + 1) * 4].numpy().view(np.int32).copy(), n_levels=n_levels,
                My=st.My.data_ptr(), Mx=st.Mx.data_ptr(), scratch=scratch.data_ptr(), work=f['work'], keep=(T, dst, st, scratch))

        if deferred:
            fut = _plan_pool().submit(L.oadg_oamix_bbox_plan, *args)
            self._pending_plans += 1

            def resolve(fut=fut, keep=keep):
                self._pending_plans -= 1
                f = finish(fut.result())
                return None if f is None else chain_rec(f)
            self._rec.append(('plan', resolve, (), _ids((T, scratch))))
            return
        f = finish(L.oadg_oamix_bbox_plan(*args))
        if f is None:
            return
        if self._rec is not None:
            self._rec.append(('chain', chain_rec(f), (), _ids((T, scratch))))
            return
        dst, tiles_off = f['dst'], f['tiles_off']
        from .. import hip_ops
        check(hip_ops._timed('oamix_bbox_chain', L.oadg_oamix_bbox_chain, ptr(T), H, W, dst.data_ptr(),
                             dst.data_ptr() + tiles_off, lf.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), f['n_levels'],
                             ctypes.cast(buf.data_ptr() + tiles_off, ctypes.POINTER(ctypes.c_int)), ptr(st.My),
                             ptr(st.Mx), ptr(scratch), stream_ptr(), work=f['work']), 'oadg_oamix_bbox_chain')

    def _box_matrices(self, st, kind):
        """The per-box loop above for ALL boxes at once: (rows, rects [m,4], inverted matrices [m,6]) of the boxes that
        take a step.  Every box with integer width and height >= 1 draws ``uniform(.1, 10)`` and a sign draw, i.e. TWO
        consecutive doubles of the stream (numpy: uniform(a, b) = a + (b - a) * next_double(), uniform() and random() =
        next_double()), in box order - one ``random_sample(2 m)`` call consumes exactly the same doubles.  The matrix
        arithmetic repeats geo_matrix / invert_affine operation for operation in float64 / float32 numpy (no fused
        multiply-adds), cos / sin of the integer angle through the same libm calls."""
        gt = st.gt
        n = len(gt)
        if n == 0:
            return [], np.zeros((0, 4), np.int32), np.zeros((0, 6))
        ib = gt.astype(np.int64)                    # int() truncation of non-negative float32 coordinates
        x1, y1, x2, y2 = ib[:, 0], ib[:, 1], ib[:, 2], ib[:, 3]
        draws = ~(((x2 - x1) < 1) | ((y2 - y1) < 1))
        m = int(draws.sum())
        r = rng.random_sample(2 * m) if m else np.zeros((0,))
        level = 0.1 + (self.severity - 0.1) * r[0::2]
        flip = r[1::2] > 0.5
        x1, y1, x2, y2 = (v[draws].astype(np.float64) for v in (x1, y1, x2, y2))
        cx, cy = (x1 + x2) / 2., (y1 + y2) / 2.
        M = np.zeros((m, 6), np.float64)
        if kind == 'rotate':
            deg = np.trunc(level * 30 / 10).astype(np.int64)
            deg = np.where(flip, -deg, deg)
            cxf, cyf = cx.astype(np.float32).astype(np.float64), cy.astype(np.float32).astype(np.float64)   # Point2f
            tab = {d: (math.cos(d * math.pi / 180.0) * 1.0, math.sin(d * math.pi / 180.0) * 1.0) for d in np.unique(deg)}
            alpha = np.array([tab[d][0] for d in deg], np.float64)
            beta = np.array([tab[d][1] for d in deg], np.float64)
            M[:, 0], M[:, 1], M[:, 2] = alpha, beta, (1 - alpha) * cxf - beta * cyf
            M[:, 3], M[:, 4], M[:, 5] = -beta, alpha, beta * cxf + (1 - alpha) * cyf
        elif kind in ('shear_x', 'shear_y'):
            lvl = level * 0.3 / 10.
            lvl = np.where(flip, -lvl, lvl)
            f32 = lambda v: np.asarray(v, np.float64).astype(np.float32).astype(np.float64)  # noqa: E731
            if kind == 'shear_x':
                M[:, 0], M[:, 1], M[:, 2], M[:, 4] = 1.0, f32(-lvl), f32(-(-lvl * cy)), 1.0
            else:
                M[:, 0], M[:, 3], M[:, 4], M[:, 5] = 1.0, f32(-lvl), 1.0, f32(-(-lvl * cx))
        else:
            size = (x2 - x1 + 1) if kind == 'translate_x' else (y2 - y1 + 1)
            lvl = np.trunc(level * (size / 3) / 10)
            lvl = np.where(flip, -lvl, lvl)
            M[:, 0], M[:, 4] = 1.0, 1.0
            M[:, 2 if kind == 'translate_x' else 5] = (-lvl).astype(np.float32).astype(np.float64)
        # invert_affine, vectorised in the same operation order
        D = M[:, 0] * M[:, 4] - M[:, 1] * M[:, 3]
        with np.errstate(divide='ignore'):
            D = np.where(D != 0, 1.0 / D, 0.0)
        A11, A22 = M[:, 4] * D, M[:, 0] * D
        m1, m3 = M[:, 1] * -D, M[:, 3] * -D
        b1 = -A11 * M[:, 2] - m1 * M[:, 5]
        b2 = -m3 * M[:, 2] - A22 * M[:, 5]
        minv = np.stack([A11, m1, b1, m3, A22, b2], 1)
        sup = [st.support[i] for i in np.nonzero(draws)[0]]
        live = np.array([s_ is not None and s_[2] > 0 and s_[3] > 0 for s_ in sup], bool)
        rows = np.nonzero(draws)[0][live]
        rects = np.array([s_ for s_, ok in zip(sup, live) if ok], np.int32).reshape(-1, 4)
        return rows, rects, minv[live]

    def _bbox_chain(self, st, T, rows, rects, minvs, step):
        """all boxes of one bboxes_only_* op: steps sorted by dependency level, one descriptor upload, 2 launches per
        level (csrc oadg_oamix_bbox_chain)"""
        L = _lib.lib()
        scratch = step['scratch']
        H, W = st.H, st.W
        n = len(rows)
        rects = np.ascontiguousarray(rects, np.int32)
        minvs = np.ascontiguousarray(minvs, np.float64)
        level = np.zeros((n,), np.int32)
        check(L.oadg_oamix_bbox_levels(rects.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                       minvs.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n, H, W,
                                       level.ctypes.data_as(ctypes.POINTER(ctypes.c_int))), 'oadg_oamix_bbox_levels')
        order = np.argsort(level, kind='stable')
        steps = np.zeros((n,), BBOX_STEP_DTYPE)
        steps['minv'] = np.asarray(minvs, np.float64)[order]
        steps['rect'] = np.asarray(rects, np.int32)[order]
        steps['row'] = np.asarray(rows, np.int32)[order]
        area = steps['rect'][:, 2].astype(np.int64) * steps['rect'][:, 3]
        lv = level[order]
        n_levels = int(lv[-1]) + 1
        if self.stats is not None:
            self.stats['bbox_levels'] = self.stats.get('bbox_levels', 0) + n_levels
            self.stats['bbox_steps'] = self.stats.get('bbox_steps', 0) + n
        first = np.searchsorted(lv, np.arange(n_levels + 1)).astype(np.int32)
        # the rects of one level are disjoint, so their packed scratch images fit the H*W*3 scratch buffer
        padded = (3 * area + 3) // 4 * 4                   # every rect starts on a 4-byte boundary of the scratch image
        cum = np.cumsum(padded) - padded
        steps['scratch_off'] = cum - cum[first[:-1]][lv]
        tiles = np.zeros((n + 1,), np.int32)
        tiles[1:] = np.cumsum((area + 1023) // 1024)        # one workgroup = 256 threads x 4 pixels
        dev = st.img.device
        keep = step.setdefault('keepalive', [])              # descriptor tensors live until the step's launches ran
        steps_dev = _upload(steps.view(np.uint8).reshape(-1), dev)
        tiles_dev = _upload(tiles, dev)
        keep += [steps_dev, tiles_dev]
        from .. import hip_ops
        # model bytes of the chain (SURVEY 8d's sum 3 w h term, per blend: the rect is read, its warped source is read,
        # the result is written) + the two mask profiles of every step
        work = float(9 * area.sum() + 4 * (steps['rect'][:, 2].sum() + steps['rect'][:, 3].sum()))
        if self._rec is not None:
            self._rec.append(('chain', dict(
                img=T.data_ptr(), H=H, W=W, steps_dev=steps_dev.data_ptr(), tile_prefix_dev=tiles_dev.data_ptr(),
                level_first=first, tile_prefix=tiles, n_levels=n_levels, My=st.My.data_ptr(), Mx=st.Mx.data_ptr(),
                scratch=scratch.data_ptr(), work=work, keep=(T, steps_dev, tiles_dev, st, scratch)), (), _ids((T, scratch))))
            return
        check(hip_ops._timed('oamix_bbox_chain', L.oadg_oamix_bbox_chain, ptr(T), H, W, ptr(steps_dev), ptr(tiles_dev),
                             first.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), n_levels,
                             tiles.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ptr(st.My), ptr(st.Mx),
                             ptr(scratch), stream_ptr(), work=work), 'oadg_oamix_bbox_chain')

    # ------------------------------------------------------------------------------------------ one view
    def oamix(self, st, out_u8=None, out_norm=None, norm=None, pad_shape=None):
        """oamix (oa_mix.py:207-243) for one image state; writes the uint8 view and/or its normalised tensor."""
        L = _lib.lib()
        H, W = st.H, st.W
        b = self._buffers(st)
        ws = np.float32(rng.dirichlet([self.aug_prob_coeff] * self.mixture_width))
        rboxes = self._random_regions(H, W, self.random_box_scale, self.random_box_ratio, (1, 3))
        self._history['random_box_list'] = np.stack(rboxes, axis=0)
        assert len(rboxes) <= 2
        rects = (ctypes.c_int * 8)(*[int(v) for bx in rboxes for v in bx], *([0] * (8 - 4 * len(rboxes))))
        for i in range(self.mixture_width):
            depth = self.mixture_depth if self.mixture_depth > 0 else rng.randint(1, 4)
            cur = st.img
            # every mixture chain starts from the ORIGINAL image (oa_mix.py:224-231) and only meets the others in the
            # accumulator: inside the lockstep pass it records on a buffer set of its own, so that execute() may advance the
            # chains' per-box levels together
            self._lane = i % LANES if self._rec is not None else 0
            lb = self._buffers(st)
            for d in range(depth):
                step = dict(n_tmp=0, luts_for=None)
                ops = (RegionOp * 3)()
                for k in range(len(rboxes)):
                    ops[k] = self._aug(st, cur, step)
                ops[2] = self._aug(st, cur, step)
                dst = lb['ping'][d & 1]
                last = d == depth - 1
                mode = (1 if i == 0 else 2) if last else 0
                self._do(lambda cur=cur, dst=dst, ops=ops, w=float(ws[i]), mode=mode, lb=lb: check(
                    L.oadg_oamix_compose(ptr(cur), ptr(dst), H, W, ops, rects, len(rboxes), ptr(lb['luts']),
                                         ptr(st.union_f), ptr(st.union_u8), ptr(lb['acc']), w, mode, stream_ptr()),
                    'oadg_oamix_compose'),
                    reads=(cur, lb['luts'], lb['gray'], *lb['tmp']), writes=(dst,) + ((lb['acc'],) if mode else ()))
                cur = dst
                if self.stats is not None:
                    self.stats['compose_steps'] = self.stats.get('compose_steps', 0) + 1
        self._lane = 0
        # get_regions_for_object_aware_mixing (oa_mix.py:245-262)
        scores = st.scores()
        targets = []
        for idx, score in enumerate(scores):
            if score <= self.score_thresh:
                targets.append((idx, (0, 0, 0, 0), score))
        rb, rs = self._random_regions(H, W, self.oa_random_box_scale, self.oa_random_box_ratio,
                                      num_bboxes=min(max(len(targets), 1), 5), return_score=True,
                                      fg_boxes=st.gt, fg_scores=scores)
        self._history['oa_random_box_list'] = rb
        for bx, sc in zip(rb, rs):
            targets.append((-1, tuple(int(v) for v in bx), sc))
        # object_aware_mixing (oa_mix.py:281-309)
        m = rng.beta(self.aug_prob_coeff, self.aug_prob_coeff)
        tg = np.zeros((len(targets),), MIX_TARGET_DTYPE)
        if len(targets) >= MIX_TILES_MIN_TARGETS:
            # the targets' mixing weights in one draw: uniform(0, hi) = 0.0 + (hi - 0.0) * next_double() per target
            # (numpy's legacy uniform), i.e. the next len(targets) doubles of the stream in target order
            hi = np.where(np.array([sc for _, _, sc in targets], np.float64) <= self.score_thresh, 0.5, 1.0)
            tg['fg_index'] = [idx for idx, _, _ in targets]
            tg['rect'] = [rect for _, rect, _ in targets]
            tg['m_oa'] = (0.0 + (hi - 0.0) * rng.random_sample(len(targets))).astype(np.float32)
        else:
            for t, (idx, rect, score) in enumerate(targets):
                hi = 0.5 if score <= self.score_thresh else 1.0
                tg[t] = (idx, rect, np.float32(rng.uniform(0.0, hi)))
        tg_dev = _upload(tg.view(np.uint8).reshape(-1), st.img.device) if len(targets) else None
        mean = stdinv = None
        to_rgb, dt, Hp, Wp = 0, 0, H, W
        if out_norm is not None:
            mean = (ctypes.c_float * 3)(*norm['mean'])
            stdinv = (ctypes.c_float * 3)(*norm['stdinv'])
            to_rgb = int(norm['to_rgb'])
            dt = 1 if out_norm.dtype == torch.bfloat16 else 0
            Hp, Wp = pad_shape
        if MIX_TILES_MIN_TARGETS <= len(targets) <= 65535 and st.n:
            # many targets: binned into 32 x 32 pixel tiles first (csrc mix_bins_kernel), byte-identical
            nb = int(L.oadg_oamix_final_tiles_workspace_bytes(H, W, len(targets)))
            ws = b.get('mix_ws')
            if ws is None or ws.numel() < nb:
                ws = b['mix_ws'] = torch.empty((nb,), dtype=torch.uint8, device=st.img.device)
            fg_rects = st.rects_dev()
            self._do(lambda: check(
                L.oadg_oamix_final_tiles(ptr(st.img), ptr(b['acc']), H, W, ptr(tg_dev), len(targets), ptr(fg_rects), ptr(st.My),
                                         ptr(st.Mx), float(m), mean, stdinv, to_rgb, ptr(out_u8), ptr(out_norm), dt, Hp, Wp,
                                         ptr(ws), nb, stream_ptr()), 'oadg_oamix_final_tiles'), reads=(b['acc'],))
            return out_u8
        self._do(lambda: check(
            L.oadg_oamix_final(ptr(st.img), ptr(b['acc']), H, W, ptr(tg_dev), len(targets), ptr(st.My),
                               ptr(st.Mx), float(m), mean, stdinv, to_rgb, ptr(out_u8), ptr(out_norm), dt, Hp,
                               Wp, stream_ptr()), 'oadg_oamix_final'), reads=(b['acc'],))
        return out_u8

    # ------------------------------------------------------------------------------------------ dict API
    def __call__(self, results, *args, **kwargs):
        """oa_mix.py:187-204.  ``results['img']`` may be a uint8 HWC numpy array (uploaded, the new view is
        returned as numpy) or a uint8 HWC cuda tensor (everything stays on the device)."""
        img = results['img']
        as_numpy = isinstance(img, np.ndarray)
        dimg = torch.from_numpy(np.ascontiguousarray(img)).cuda() if as_numpy else img.contiguous()
        if dimg.dtype != torch.uint8 or dimg.dim() != 3 or dimg.shape[2] != 3 or not dimg.is_cuda:
            raise TypeError('OAMix expects a uint8 HxWx3 image (numpy or cuda tensor)')
        gts = results['gt_bboxes']
        gts = gts.detach().cpu().numpy() if isinstance(gts, torch.Tensor) else np.asarray(gts)
        state = None
        results['custom_field'] = []
        for i in range(1, self.num_views + 1):
            if i == 1:
                self._history = {}
                if not self.keep_orig:
                    state = state or _ImageState(dimg, gts, self.spatial_ratio, self.sigma_ratio)
                    out = self.oamix(state, out_u8=torch.empty_like(dimg))
                    results['img'] = out.cpu().numpy() if as_numpy else out
                results['img_fields'] = ['img']
            else:
                state = state or _ImageState(dimg, gts, self.spatial_ratio, self.sigma_ratio)
                out = self.oamix(state, out_u8=torch.empty_like(dimg))
                results[f'img{i}'] = out.cpu().numpy() if as_numpy else out
                results['img_fields'] += [f'img{i}']
                results[f'gt_bboxes{i}'] = results['gt_bboxes'].copy() if as_numpy or isinstance(
                    results['gt_bboxes'], np.ndarray) else results['gt_bboxes'].clone()
                results['oamix_boxes'] = np.stack(self._history['oa_random_box_list'], axis=0)
                results['custom_field'] += [f'img{i}', f'gt_bboxes{i}', 'oamix_boxes']
                results['multilevel_boxes'] = self._history['random_box_list']
                results['custom_field'] += ['multilevel_boxes']
        return results

    def __repr__(self):
        return self.__class__.__name__
