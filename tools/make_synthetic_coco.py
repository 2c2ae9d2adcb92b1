"""Write a Cityscapes-shaped synthetic dataset to DISK - PNG files + a COCO-format json with the Cityscapes categories -
so that tools/train.py can be run on FILES (decode + upload + Resize / RandomFlip + OA-Mix + train) where the real
Cityscapes tree is not available: the same images and boxes as ``SyntheticCityscapes``.

usage: python tools/make_synthetic_coco.py OUT_DIR [--n 64] [--height 1024] [--width 2048] [--boxes 20]
writes OUT_DIR/img/{i}.png, OUT_DIR/train.json; run e.g.
  python tools/train.py configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py --allow-missing-pretrained \
      --cfg-options data.train.dataset.ann_file=OUT_DIR/train.json data.train.dataset.img_prefix=OUT_DIR/img/
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oadg_amd  # noqa: F401,E402
from oadg_amd.pipelines import SyntheticCityscapes  # noqa: E402

CLASSES = ('person', 'rider', 'car', 'truck', 'bus', 'train', 'motorcycle', 'bicycle')


def main():
    p = argparse.ArgumentParser()
    p.add_argument('out')
    p.add_argument('--n', type=int, default=64)
    p.add_argument('--height', type=int, default=1024)
    p.add_argument('--width', type=int, default=2048)
    p.add_argument('--boxes', type=int, default=20)
    p.add_argument('--compress-level', type=int, default=6, help='zlib level of the PNG files (PIL default 6)')
    a = p.parse_args()
    from PIL import Image
    dev = 'cuda' if torch.cuda.is_available() else 'cpu'
    ds = SyntheticCityscapes(img_shape=(a.height, a.width), num_boxes=a.boxes, device=dev)
    os.makedirs(os.path.join(a.out, 'img'), exist_ok=True)
    images, anns, nbytes = [], [], 0
    for i in range(a.n):
        bgr = ds.image(i).cpu().numpy()
        path = os.path.join(a.out, 'img', f'{i}.png')
        Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(path, compress_level=a.compress_level)
        nbytes += os.path.getsize(path)
        images.append(dict(id=i, file_name=f'{i}.png', height=a.height, width=a.width, segm_file=''))
        boxes, labels = ds.boxes(i)
        for b, l in zip(boxes, labels):
            w, h = float(b[2] - b[0]), float(b[3] - b[1])
            anns.append(dict(id=len(anns), image_id=i, category_id=int(l) + 1, iscrowd=0, area=w * h,
                             bbox=[float(b[0]), float(b[1]), w, h], segmentation=[]))
    cats = [dict(id=k + 1, name=n) for k, n in enumerate(CLASSES)]
    with open(os.path.join(a.out, 'train.json'), 'w') as f:
        json.dump(dict(images=images, annotations=anns, categories=cats), f)
    print(f'{a.n} images of {a.height}x{a.width}, {nbytes / a.n / 1e6:.2f} MB per PNG, {len(anns)} boxes -> {a.out}')


if __name__ == '__main__':
    main()
