#!/usr/bin/env bash
# tools/train.py on FILES: N synthetic 1024x2048 PNGs + a COCO json written to /tmp on the GPU box, the
# CLI for ITERS iterations through CityscapesDataset (decode -> pinned upload -> OA-Mix -> step), then the same run on the
# HBM-resident synthetic source for comparison.  bash tools/probe/train_on_files.sh TAG [N] [ITERS]
TAG=${1:-tmp}; N=${2:-64}; ITERS=${3:-260}
cd $GRAFT_REPO_ROOT
python tools/make_synthetic_coco.py /tmp/oadg_png --n $N > gpurun_out/train_files_$TAG.log 2>&1
CFG=configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py
COMMON="--max-iters $ITERS --seed 0 --allow-missing-pretrained --cfg-options data.samples_per_gpu=4 log_config.interval=20 checkpoint_config.interval=100000 runner.max_epochs=1000"
echo "== files: CityscapesDataset on $N PNGs ${EXTRA:-}" >> gpurun_out/train_files_$TAG.log
python tools/train.py $CFG --work-dir /tmp/wd_files $COMMON data.train.type=CityscapesDataset data.train.ann_file=/tmp/oadg_png/train.json data.train.img_prefix=/tmp/oadg_png/img/ ${EXTRA:-} 2>&1 | grep -v amdgpu.ids | cut -c1-220 >> gpurun_out/train_files_$TAG.log
echo "== synthetic source (HBM-resident generator)" >> gpurun_out/train_files_$TAG.log
python tools/train.py $CFG --work-dir /tmp/wd_syn $COMMON 2>&1 | grep -v amdgpu.ids | cut -c1-220 >> gpurun_out/train_files_$TAG.log
