"""COCO-format / Cityscapes dataset loading (oadg_amd/datasets.py; mmdet/datasets/{coco,cityscapes}.py): a tiny
dataset written to disk - annotation filtering rules, image filtering, PNG decode in BGR order, RepeatDataset."""
import json
import os

import numpy as np
import pytest
import torch


def _write(tmp_path):
    from PIL import Image
    rs = np.random.RandomState(0)
    imgs, images, anns = {}, [], []
    for i, (h, w) in enumerate([(40, 64), (40, 64), (40, 64), (40, 64), (20, 64)]):
        arr = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)         # RGB on disk
        os.makedirs(tmp_path / 'img' / 'city', exist_ok=True)
        Image.fromarray(arr).save(tmp_path / 'img' / 'city' / f'{i}.png')
        imgs[i] = arr
        images.append(dict(id=100 + i, file_name=f'city/{i}.png', height=h, width=w, segm_file=f'{i}_seg.png'))
    cats = [dict(id=24, name='person'), dict(id=26, name='car'), dict(id=99, name='unicorn')]

    def ann(aid, img, cat, box, crowd=0, area=None, **kw):
        return dict(id=aid, image_id=100 + img, category_id=cat, bbox=box, iscrowd=crowd,
                    area=box[2] * box[3] if area is None else area, segmentation=[], **kw)
    anns += [ann(1, 0, 26, [5, 6, 20, 10]), ann(2, 0, 24, [1.5, 2.5, 8, 9]), ann(3, 0, 26, [30, 3, 10, 10], crowd=1),
             ann(4, 0, 26, [3, 3, 0.5, 9]), ann(5, 0, 99, [3, 3, 9, 9]), ann(6, 0, 24, [3, 3, 9, 9], area=0),
             ann(7, 0, 24, [3, 3, 9, 9], ignore=True)]
    anns += [ann(8, 1, 26, [10, 10, 12, 12], crowd=1)]           # image 1: only crowd -> dropped by Cityscapes
    anns += [ann(9, 2, 99, [10, 10, 12, 12])]                    # image 2: no annotation of a wanted class
    anns += [ann(10, 4, 26, [10, 5, 12, 12])]                    # image 4: smaller than min_size 32
    # image 3: no annotations at all
    with open(tmp_path / 'ann.json', 'w') as f:
        json.dump(dict(images=images, annotations=anns, categories=cats), f)
    return imgs


def test_cityscapes_dataset_parsing_and_decode(tmp_path):
    from oadg_amd.datasets import build_dataset
    imgs = _write(tmp_path)
    cfg = dict(type='RepeatDataset', times=3,
               dataset=dict(type='CityscapesDataset', ann_file=str(tmp_path / 'ann.json'),
                            img_prefix=str(tmp_path / 'img') + '/', pipeline=[]))
    ds = build_dataset(cfg, default_args=dict(device='cpu'))
    inner = ds.dataset
    assert inner.cat_ids == [24, 26] and inner.cat2label == {24: 0, 26: 1}      # file order, not CLASSES order
    assert [d['id'] for d in inner.data_infos] == [100] and len(ds) == 3
    a = inner.get_ann_info(0)
    assert np.array_equal(a['bboxes'], np.array([[5, 6, 25, 16], [1.5, 2.5, 9.5, 11.5]], np.float32))
    assert np.array_equal(a['labels'], np.array([1, 0])) and a['labels'].dtype == np.int64
    assert np.array_equal(a['bboxes_ignore'], np.array([[30, 3, 40, 13]], np.float32))
    batch, boxes, labels = ds.batch([0, 1, 2])
    assert batch.shape == (3, 40, 64, 3) and batch.dtype == torch.uint8
    assert np.array_equal(batch[1].numpy(), imgs[0][:, :, ::-1])                # BGR, as cv2 / mmcv.imfrombytes
    assert np.array_equal(boxes[2], a['bboxes'])
    # test_mode keeps every image, in file order
    t = build_dataset(dict(type='CityscapesDataset', ann_file=str(tmp_path / 'ann.json'), test_mode=True,
                           img_prefix=str(tmp_path / 'img')), default_args=dict(device='cpu'))
    assert len(t) == 5 and t.get_ann_info(3)['bboxes'].shape == (0, 4)
    # plain CocoDataset keeps the crowd-only image (coco.py:99-121) and applies the in-image intersection rule
    c = build_dataset(dict(type='CocoDataset', classes=('person', 'car'), ann_file='ann.json', data_root=str(tmp_path),
                           img_prefix='img'), default_args=dict(device='cpu'))
    assert [d['id'] for d in c.data_infos] == [100, 101]


def test_missing_files_fall_back_to_synthetic(capsys):
    from oadg_amd.datasets import build_dataset
    from oadg_amd.pipelines import SyntheticCityscapes
    cfg = dict(type='RepeatDataset', times=8,
               dataset=dict(type='CityscapesDataset', ann_file='/nonexistent/ann.json', img_prefix='/nonexistent/', pipeline=[]))
    ds = build_dataset(cfg, default_args=dict(device='cpu', seed=3), synthetic_fallback=True)
    assert isinstance(ds, SyntheticCityscapes) and 'not found' in capsys.readouterr().out
    with pytest.raises(FileNotFoundError):
        build_dataset(cfg, default_args=dict(device='cpu'))


@pytest.mark.gpu
def test_real_files_through_the_reference_pipeline_and_a_train_step(dev, tmp_path):
    """PNG files + COCO json -> CityscapesDataset -> the reference's full train pipeline list (Resize, RandomFlip, OAMix,
    Normalize, Pad) on the device -> one bf16 train step with finite losses."""
    from PIL import Image
    from oadg_amd import Config, build_detector, hip_conv
    from oadg_amd.apis import TrainEngine, build_optimizer
    from oadg_amd.datasets import build_dataset
    from oadg_amd.pipelines import DevicePipeline
    from inputs import lowpass_image, synthetic_boxes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg_multiscale.py'))
    rs = np.random.RandomState(1)
    H, W, images, anns = 256, 512, [], []
    os.makedirs(tmp_path / 'img', exist_ok=True)
    for i in range(2):
        Image.fromarray(lowpass_image(rs, H, W, 4)[:, :, ::-1]).save(tmp_path / 'img' / f'{i}.png')
        images.append(dict(id=i, file_name=f'{i}.png', height=H, width=W, segm_file=''))
        for j, b in enumerate(synthetic_boxes(rs, 6, H, W, 24, 120)):
            anns.append(dict(id=10 * i + j, image_id=i, category_id=24 + (j % 8), iscrowd=0, segmentation=[],
                             bbox=[float(b[0]), float(b[1]), float(b[2] - b[0]), float(b[3] - b[1])],
                             area=float((b[2] - b[0]) * (b[3] - b[1]))))
    cats = [dict(id=24 + k, name=n) for k, n in enumerate(('person', 'rider', 'car', 'truck', 'bus', 'train',
                                                            'motorcycle', 'bicycle'))]
    with open(tmp_path / 'ann.json', 'w') as f:
        json.dump(dict(images=images, annotations=anns, categories=cats), f)
    ds = build_dataset(dict(type='CityscapesDataset', ann_file=str(tmp_path / 'ann.json'), img_prefix=str(tmp_path / 'img')),
                       default_args=dict(device=dev))
    pipeline = [dict(t) for t in cfg.data.train.pipeline]
    pipeline[2] = dict(type='Resize', img_scale=[(512, 200), (512, 256)], keep_ratio=True)
    pipe = DevicePipeline(pipeline, dtype=torch.bfloat16, one_scale_per_batch=True)
    np.random.seed(0)
    torch.manual_seed(0)
    data = pipe(*ds.batch([0, 1]))
    assert data['img'].shape[0] == 2 and data['img'].shape[2] % 32 == 0 and data['img2'].shape == data['img'].shape
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    hip_conv.enable()
    out = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16).step(data)
    assert np.isfinite(float(out['loss'])) and float(out['loss']) > 0


def test_integrate_data_uses_the_shared_allocation_of_adjacent_views():
    """base.py:22-48 concatenates the views; when the pipeline wrote them into one allocation (tagged on the first view)
    integrate_data hands that tensor out instead of copying - same values, same order - and still concatenates anything
    that is not laid out that way"""
    import torch
    from oadg_amd.config import ConfigDict
    from oadg_amd.detectors import integrate_data
    both = torch.arange(4 * 3 * 2 * 5, dtype=torch.float32).reshape(4, 3, 2, 5).contiguous(memory_format=torch.channels_last)
    img, img2 = both[:2], both[2:]
    img._oadg_batch = both

    def data(a, b):
        return dict(img=a, img2=b, img_metas=[dict(), dict()], gt_bboxes=[torch.zeros(1, 4)] * 2, gt_labels=[torch.zeros(1)] * 2)
    cfg = ConfigDict(dict())
    out = integrate_data(data(img, img2), cfg)
    assert out['img'].data_ptr() == both.data_ptr() and torch.equal(out['img'], both)
    assert out['num_views'] == 2 and out['batch_size'] == 2 and len(out['img_metas']) == 4
    # a second view that lives elsewhere: plain concatenation
    other = img2.clone()
    out = integrate_data(data(img, other), cfg)
    assert out['img'].data_ptr() != both.data_ptr() and torch.equal(out['img'], both)
    # the tag on a tensor whose neighbour has different strides
    odd = img2.contiguous()
    out = integrate_data(data(img, odd), cfg)
    assert torch.equal(out['img'], both)


def test_native_png_decoder_equals_pil(tmp_path):
    """csrc/png_decode.hip (host code: zlib inflate + the five PNG row filters, BGR written in place) against PIL on RGB /
    RGBA / grey files of odd extents, low-pass and noisy content (PIL's encoder picks the row filters adaptively: all five
    occur), several zlib levels; a 16-bit file is declined (-4: the data set then decodes it with PIL), a wrong extent is
    -2, a missing file -3; and CocoDataset.decode_into takes the native path for PNG and PIL for everything else."""
    import ctypes
    from PIL import Image
    from oadg_amd import _lib
    L = _lib.lib()
    rs = np.random.RandomState(3)
    cases = []
    for k, (h, w, mode) in enumerate([(37, 53, 'RGB'), (64, 128, 'RGB'), (31, 17, 'RGBA'), (40, 33, 'L'), (1, 1, 'RGB'),
                                      (129, 255, 'RGB')]):
        ch = {'RGB': 3, 'RGBA': 4, 'L': 1}[mode]
        a = rs.randint(0, 256, (h, w, ch)).astype(np.uint8)
        if k % 2 == 1:                       # smooth content: Sub / Up / Average / Paeth rows
            a = (np.cumsum(np.cumsum(a.astype(np.float64), 0), 1) / (np.arange(1, h + 1)[:, None, None] * np.arange(1, w + 1)[None, :, None])).astype(np.uint8)
        img = Image.fromarray(a[:, :, 0] if ch == 1 else a, mode)
        path = str(tmp_path / f'c{k}.png')
        img.save(path, compress_level=(1, 6, 9)[k % 3])
        cases.append((path, h, w))
    filters = set()
    for path, h, w in cases:
        hh, ww = ctypes.c_int(0), ctypes.c_int(0)
        assert L.oadg_png_size(path.encode(), ctypes.byref(hh), ctypes.byref(ww)) == 0 and (hh.value, ww.value) == (h, w)
        out = np.full((h, w, 3), 7, np.uint8)
        assert L.oadg_png_decode_bgr(path.encode(), out.ctypes.data, h, w) == 0, path
        with Image.open(path) as im:
            ref = np.asarray(im.convert('RGB'))[:, :, ::-1]
        assert np.array_equal(out, ref), path
        assert L.oadg_png_decode_bgr(path.encode(), out.ctypes.data, h + 1, w) == -2
    a16 = (rs.randint(0, 65536, (9, 11)).astype(np.uint16))
    Image.fromarray(a16).save(str(tmp_path / 'deep.png'))
    out = np.zeros((9, 11, 3), np.uint8)
    assert L.oadg_png_decode_bgr(str(tmp_path / 'deep.png').encode(), out.ctypes.data, 9, 11) == -4
    assert L.oadg_png_decode_bgr(str(tmp_path / 'nope.png').encode(), out.ctypes.data, 9, 11) == -3
    # through the data set: PNG natively, JPEG (and the declined PNG) through PIL - same bytes as decode()
    from oadg_amd.datasets import CocoDataset
    os.makedirs(tmp_path / 'img', exist_ok=True)
    b = rs.randint(0, 256, (40, 48, 3)).astype(np.uint8)
    Image.fromarray(b).save(str(tmp_path / 'img' / 'a.png'))
    Image.fromarray(b).save(str(tmp_path / 'img' / 'b.jpg'))
    images = [dict(id=0, file_name='a.png', height=40, width=48), dict(id=1, file_name='b.jpg', height=40, width=48)]
    anns = [dict(id=i, image_id=i, category_id=1, iscrowd=0, area=50.0, bbox=[2, 3, 10, 5]) for i in range(2)]
    with open(tmp_path / 'ann.json', 'w') as f:
        json.dump(dict(images=images, annotations=anns, categories=[dict(id=1, name='car')]), f)
    ds = CocoDataset(str(tmp_path / 'ann.json'), classes=('car',), img_prefix=str(tmp_path / 'img'), device='cpu')
    imgs, boxes, labels = ds.batch([0, 1])
    assert imgs.shape == (2, 40, 48, 3) and np.array_equal(imgs[0].numpy(), b[:, :, ::-1])
    assert np.array_equal(imgs[1].numpy(), ds.decode(1)) and len(boxes) == 2
    imgs2, _, _ = ds.batch([1, 0])                      # the ring's next slot; the first batch's tensor is untouched
    assert np.array_equal(imgs2[1].numpy(), imgs[0].numpy()) and np.array_equal(imgs[0].numpy(), b[:, :, ::-1])
