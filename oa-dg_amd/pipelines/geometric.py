"""Resize / RandomFlip of the reference's train pipeline, on uint8 images resident in HBM
(mmdet/datasets/pipelines/transforms.py:45-310 and :316-470).

The random draws follow the reference call for call (``np.random.randint`` x2 for a scale range /
``random_sample`` for a ratio range / ``randint`` for a value list; then ``np.random.choice`` over the flip
directions) through the same random stream OA-Mix uses (pipelines.oa_mix.rng); the pixels come from
csrc/imgxform.hip (cv2.resize INTER_LINEAR 8-bit path, mmcv.imflip); box arithmetic is the reference's numpy float32
expression.  ``LoadImageFromFile`` / ``LoadAnnotations`` only carry their configuration: samples reach this
pipeline as device tensors (decoding is outside the path).
"""
import numpy as np
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr
from ..registry import PIPELINES
from .oa_mix import rng


@PIPELINES.register_module()
class LoadImageFromFile:
    def __init__(self, to_float32=False, color_type='color', file_client_args=None, **kwargs):
        self.to_float32 = to_float32


@PIPELINES.register_module()
class LoadAnnotations:
    def __init__(self, with_bbox=True, with_label=True, with_mask=False, with_seg=False, poly2mask=True, **kwargs):
        assert not with_mask and not with_seg, 'bbox-only detector'


def rescale_size(w, h, scale):
    """mmcv.rescale_size: (new_w, new_h) for a number or a (long edge, short edge) tuple."""
    if isinstance(scale, (float, int)):
        f = scale
    else:
        f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


@PIPELINES.register_module()
class Resize:
    """transforms.py:45-310 (cv2 backend, images + bboxes)."""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True,
                 bbox_clip_border=True, backend='cv2', override=False):
        assert backend == 'cv2'
        self.img_scale = None if img_scale is None else \
            [tuple(s) for s in (img_scale if isinstance(img_scale, list) else [img_scale])]
        if ratio_range is not None:
            assert len(self.img_scale) == 1
        else:
            assert multiscale_mode in ('value', 'range')
        self.multiscale_mode, self.ratio_range, self.keep_ratio = multiscale_mode, ratio_range, keep_ratio
        self.bbox_clip_border, self.override = bbox_clip_border, override

    def draw_scale(self):
        """_random_scale (:177-208), same draws in the same order."""
        if self.ratio_range is not None:
            lo, hi = self.ratio_range
            ratio = rng.random_sample() * (hi - lo) + lo
            return int(self.img_scale[0][0] * ratio), int(self.img_scale[0][1] * ratio)
        if len(self.img_scale) == 1:
            return self.img_scale[0]
        if self.multiscale_mode == 'range':
            assert len(self.img_scale) == 2
            longs, shorts = [max(s) for s in self.img_scale], [min(s) for s in self.img_scale]
            long_edge = rng.randint(min(longs), max(longs) + 1)
            short_edge = rng.randint(min(shorts), max(shorts) + 1)
            return long_edge, short_edge
        return self.img_scale[rng.randint(len(self.img_scale))]

    def plan(self, H, W, scale=None):
        """Host half of _resize_img (:210-239): draws the scale (unless given), returns (scale, Wn, Hn, scale_factor)."""
        if scale is None:
            scale = self.draw_scale()
        if self.keep_ratio:
            Wn, Hn = rescale_size(W, H, scale)
        else:
            Wn, Hn = int(scale[0]), int(scale[1])
        w_scale, h_scale = Wn / W, Hn / H
        return scale, Wn, Hn, np.array([w_scale, h_scale, w_scale, h_scale], dtype=np.float32)

    def resize_bboxes(self, bboxes, sf, Hn, Wn):
        """_resize_bboxes (:241-249)."""
        b = np.asarray(bboxes, dtype=np.float32) * sf
        if self.bbox_clip_border:
            b[:, 0::2] = np.clip(b[:, 0::2], 0, Wn)
            b[:, 1::2] = np.clip(b[:, 1::2], 0, Hn)
        return b

    def __call__(self, img, bboxes, scale=None):
        """img uint8 [H,W,3] cuda tensor, bboxes float32 [n,4] numpy -> (img', bboxes', meta)."""
        H, W = img.shape[:2]
        scale, Wn, Hn, sf = self.plan(H, W, scale)
        out = torch.empty((Hn, Wn, img.shape[2]), dtype=torch.uint8, device=img.device)
        check(_lib.lib().oadg_resize_bilinear_u8(ptr(img.contiguous()), H, W, img.shape[2], ptr(out), Hn, Wn,
                                                 stream_ptr()), 'oadg_resize_bilinear_u8')
        return out, self.resize_bboxes(bboxes, sf, Hn, Wn), \
            dict(scale=scale, scale_factor=sf, keep_ratio=self.keep_ratio, img_shape=(Hn, Wn, img.shape[2]))


@PIPELINES.register_module()
class RandomFlip:
    """transforms.py:316-470 (images + bboxes)."""
    _CODES = dict(horizontal=1, vertical=2, diagonal=3)

    def __init__(self, flip_ratio=None, direction='horizontal'):
        if isinstance(flip_ratio, list):
            assert 0 <= sum(flip_ratio) <= 1
        elif flip_ratio is not None:
            assert isinstance(flip_ratio, float) and 0 <= flip_ratio <= 1
        self.flip_ratio, self.direction = flip_ratio, direction
        if isinstance(flip_ratio, list):
            assert len(flip_ratio) == len(direction)

    def draw(self):
        """:425-444: one ``np.random.choice`` over the directions + None."""
        direction_list = (self.direction if isinstance(self.direction, list) else [self.direction]) + [None]
        if isinstance(self.flip_ratio, list):
            ratios = self.flip_ratio + [1 - sum(self.flip_ratio)]
        else:
            single = self.flip_ratio / (len(direction_list) - 1)
            ratios = [single] * (len(direction_list) - 1) + [1 - self.flip_ratio]
        return rng.choice(direction_list, p=ratios)

    @staticmethod
    def bbox_flip(bboxes, img_shape, direction):
        flipped = bboxes.copy()
        h, w = img_shape[:2]
        if direction in ('horizontal', 'diagonal'):
            flipped[..., 0::4] = w - bboxes[..., 2::4]
            flipped[..., 2::4] = w - bboxes[..., 0::4]
        if direction in ('vertical', 'diagonal'):
            flipped[..., 1::4] = h - bboxes[..., 3::4]
            flipped[..., 3::4] = h - bboxes[..., 1::4]
        return flipped

    def __call__(self, img, bboxes, preset=None):
        """``preset`` (test-time augmentation: MultiScaleFlipAug sets results['flip'] / ['flip_direction'] itself,
        transforms.py:446-450): (flip, direction) instead of a draw"""
        cur = self.draw() if preset is None else (preset[1] if preset[0] else None)
        if cur is None:
            return img, bboxes, dict(flip=False, flip_direction=None)
        H, W = img.shape[:2]
        out = torch.empty_like(img)
        check(_lib.lib().oadg_flip_u8(ptr(img.contiguous()), H, W, img.shape[2], ptr(out), self._CODES[cur],
                                      stream_ptr()), 'oadg_flip_u8')
        return out, self.bbox_flip(np.asarray(bboxes, dtype=np.float32), (H, W), cur), \
            dict(flip=True, flip_direction=str(cur))
