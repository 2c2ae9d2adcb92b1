#!/usr/bin/env python
"""Train a detector: the reference's entry point (tools/train.py:22-207) for the MI355X build.

    python tools/train.py CONFIG [--work-dir D] [--resume-from F] [--auto-resume] [--no-validate]
                          [--gpus N | --gpu-ids I ...] [--seed S] [--deterministic]
                          [--cfg-options k=v ...] [--launcher none|pytorch|slurm|mpi] [--local_rank R]
                          [--debug_mode] [--max-iters N]

CONFIG may be one of this repo's configs (configs/oadg/*.py) or an unmodified reference config (its
``/ws/external/...`` bases are resolved against the tree the config lives in, or $OADG_CONFIG_ROOT).
``CityscapesDataset`` / ``CocoDataset`` (COCO-format json + image files, oadg_amd/datasets.py) and
``SyntheticCityscapes`` are built; a dataset whose annotation file is absent on this machine is replaced by the
synthetic Cityscapes-shaped source (with a notice) while its pipeline list is honoured.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class DictAction(argparse.Action):
    """key=value pairs -> dict (mmcv.DictAction): ints/floats/bools/lists/tuples are parsed."""

    def __call__(self, parser, namespace, values, option_string=None):
        opts = {}
        for kv in values:
            k, v = kv.split('=', maxsplit=1)
            opts[k] = v
        setattr(namespace, self.dest, opts)


def parse_args():
    p = argparse.ArgumentParser(description='Train a detector')
    p.add_argument('config', help='train config file path')
    p.add_argument('--work-dir', help='the dir to save logs and models')
    p.add_argument('--resume-from', help='the checkpoint file to resume from')
    p.add_argument('--auto-resume', action='store_true', help='resume from the latest checkpoint automatically')
    p.add_argument('--no-validate', action='store_true', help='accepted for compatibility (evaluation is out of scope)')
    g = p.add_mutually_exclusive_group()
    g.add_argument('--gpus', type=int, help='number of gpus to use (only applicable to non-distributed training)')
    g.add_argument('--gpu-ids', type=int, nargs='+', help='ids of gpus to use (non-distributed training)')
    p.add_argument('--seed', type=int, default=None, help='random seed')
    p.add_argument('--deterministic', action='store_true')
    p.add_argument('--options', nargs='+', action=DictAction, help='deprecated alias of --cfg-options')
    p.add_argument('--cfg-options', nargs='+', action=DictAction, help='override config entries, key=value')
    p.add_argument('--launcher', choices=['none', 'pytorch', 'slurm', 'mpi'], default='none')
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--debug_mode', action='store_true')
    p.add_argument('--max-iters', type=int, default=None, help='stop after this many iterations (smoke runs)')
    p.add_argument('--amp', default='bf16', choices=['bf16', 'none'])
    p.add_argument('--one-scale-per-batch', action='store_true',
                   help='multi-scale Resize: draw ONE scale per batch instead of the reference\'s one per sample')
    p.add_argument('--allow-missing-pretrained', action='store_true',
                   help='train from random weights (with a warning) when a configured pretrained / load_from '
                        'checkpoint is not available locally, instead of stopping')
    p.add_argument('--loader-depth', type=int, default=3,
                   help='batches the loader keeps in flight ahead of the augmentation pipeline (decode of real files: a '
                        'batch of 1024x2048 PNGs takes longer to inflate than a step takes to train)')
    p.add_argument('--no-cpu-affinity', action='store_true',
                   help='do not pin this rank\'s threads to a compact set of physical cores (oadg_amd.apis.pin_rank_to_cores)')
    a = p.parse_args()
    if 'LOCAL_RANK' not in os.environ:
        os.environ['LOCAL_RANK'] = str(a.local_rank)
    if a.options and a.cfg_options:
        raise ValueError('--options and --cfg-options cannot be both specified')
    if a.options:
        a.cfg_options = a.options
    return a


def latest_checkpoint(work_dir):
    """mmdet/utils/misc.py:7-38 find_latest_checkpoint."""
    if not work_dir or not os.path.isdir(work_dir):
        return None
    if os.path.exists(os.path.join(work_dir, 'latest.pth')):
        return os.path.join(work_dir, 'latest.pth')
    cks = [f for f in os.listdir(work_dir) if f.endswith('.pth')]
    return os.path.join(work_dir, max(cks, key=lambda f: os.path.getmtime(os.path.join(work_dir, f)))) if cks else None


def main():
    a = parse_args()
    import oadg_amd
    from oadg_amd import Config, build_detector
    from oadg_amd.apis import (TrainEngine, StepLrSchedule, build_optimizer, get_dist_info, init_dist,
                               init_random_seed, parse_optimizer_config, set_random_seed)
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    cfg = Config.fromfile(a.config)
    if a.cfg_options:
        cfg.merge_from_dict(a.cfg_options)
    opt_hook = parse_optimizer_config(cfg)       # grad_clip honoured; fp16 / custom hooks rejected by name, before any work
    work_dir = a.work_dir or cfg.get('work_dir') or os.path.join('./work_dirs',
                                                                 os.path.splitext(os.path.basename(a.config))[0])
    distributed = a.launcher != 'none'
    if distributed:
        init_dist(a.launcher, **cfg.get('dist_params', dict(backend='nccl')))
    rank, world = get_dist_info()
    if not a.no_cpu_affinity and os.environ.get('OADG_BENCH_NO_AFFINITY') != '1':
        # a rank's threads on one CCD's worth of cores of its own (local ranks = LOCAL_WORLD_SIZE under torch.distributed.run)
        from oadg_amd.apis import pin_rank_to_cores
        aff = pin_rank_to_cores(int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('LOCAL_WORLD_SIZE', 1)))
        if rank == 0 and aff:
            print(f'cpu affinity: {aff}', flush=True)
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
    dev = torch.device('cuda', torch.cuda.current_device())
    os.makedirs(work_dir, exist_ok=True)
    seed = init_random_seed(a.seed, device=dev)
    set_random_seed(seed, deterministic=a.deterministic)
    model = build_detector(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    from oadg_amd.checkpoint import load_checkpoint
    log0 = print if rank == 0 else None
    # (None when the flag is absent: ResNet.init_weights then honours OADG_ALLOW_RANDOM_INIT=1)
    model.init_weights(allow_missing_pretrained=True if a.allow_missing_pretrained else None)
    model = model.to(dev).to(memory_format=torch.channels_last).train()
    # the log variables stay on the device and are read when a line is printed (every log_config.interval iterations) - the
    # reference's _parse_losses reads them with .item() on every iteration (base.py:270-275), a full device synchronisation
    # per step; the printed values are the same
    model.log_vars_on_host = False
    resume = a.resume_from or (latest_checkpoint(work_dir) if a.auto_resume else None) or cfg.get('resume_from')
    start_iter = start_epoch = skip_batches = 0
    optimizer = build_optimizer(model, cfg.optimizer)
    load_from = cfg.get('load_from')
    if resume:
        # EpochBasedRunner.resume (mmcv/runner/base_runner.py): weights, optimizer state, epoch AND iteration
        rep = load_checkpoint(model, resume, map_location=dev, strict=False, logger=log0)
        ck = rep['checkpoint']
        if 'optimizer' in ck:
            optimizer.load_state_dict(ck['optimizer'])
        start_iter = int(ck.get('meta', {}).get('iter', 0))
        start_epoch = int(ck.get('meta', {}).get('epoch', 0))
        # a run stopped INSIDE an epoch (--max-iters) saved meta.epoch = the unfinished epoch and meta.inner_iter = the
        # batches of it already trained: that epoch is re-entered and those batches of its (seeded) order are skipped
        skip_batches = int(ck.get('meta', {}).get('inner_iter', 0))
        if rank == 0:
            print(f'resumed epoch {start_epoch}, iter {start_iter}' + (f' (+{skip_batches} batches into the epoch)'
                                                                        if skip_batches else '') + f' from {resume}', flush=True)
    elif load_from:
        # mmdet checkpoints ({'state_dict': ...}, same key names); a head trained for another class count is dropped
        # key by key and reported, as mmcv.load_checkpoint does (apis/train.py:196-197)
        try:
            rep = load_checkpoint(model, load_from, map_location=dev, strict=False, logger=log0)
            if rank == 0:
                print(f'load_from {rep["path"]}: {rep["loaded"]} tensors loaded, {len(rep["missing"])} missing, '
                      f'{len(rep["unexpected"])} unexpected, {len(rep["mismatched"])} size-mismatched', flush=True)
        except FileNotFoundError as e:
            if not a.allow_missing_pretrained:
                raise
            if rank == 0:
                print(f'[oadg] WARNING load_from: {e} -> continuing WITHOUT it (--allow-missing-pretrained)', flush=True)
    amp = torch.bfloat16 if a.amp == 'bf16' else None
    engine = TrainEngine(model, optimizer, distributed=distributed, amp_dtype=amp,
                         find_unused_parameters=cfg.get('find_unused_parameters', False), **opt_hook)
    if rank == 0 and opt_hook['grad_clip']:
        print(f"optimizer_config.grad_clip: clip_grad_norm_({opt_hook['grad_clip']})", flush=True)
    sched = StepLrSchedule(optimizer, **cfg.get('lr_config', dict(policy='step', step=[1 << 30])))
    from oadg_amd.datasets import build_dataset
    ds = build_dataset(cfg.data.train, default_args=dict(seed=seed + rank, device=dev), synthetic_fallback=True)
    dcfg = cfg.data.train
    while 'dataset' in dcfg and dcfg.get('type') in ('RepeatDataset',):
        dcfg = dcfg.dataset
    pipe = DevicePipeline(dcfg.pipeline, dtype=amp or torch.float32, one_scale_per_batch=a.one_scale_per_batch)
    bs = cfg.data.get('samples_per_gpu', 2)
    epochs = cfg.get('runner', dict(max_epochs=1)).get('max_epochs', 1)
    from oadg_amd.samplers import batches, build_sampler
    # shuffled, aspect-ratio-grouped training order (GroupSampler / DistributedGroupSampler, datasets/builder.py:128-160)
    sampler = build_sampler(ds, bs, distributed, rank, world, seed)
    iters_per_epoch = len(sampler) // bs
    interval = cfg.get('log_config', {}).get('interval', 50)

    ck_interval = int(cfg.get('checkpoint_config', {}).get('interval', 0) or 0)

    def save_checkpoint(epoch, it, inner=None):
        """CheckpointHook (by_epoch): epoch_{n}.pth after every ``interval``-th COMPLETED epoch (+ latest.pth).  ``inner``
        (a run cut short by --max-iters inside an epoch): only latest.pth, with meta.epoch = the unfinished epoch and
        meta.inner_iter = its batches already trained, so that a resumed run finishes that epoch instead of skipping it."""
        if rank != 0 or not ck_interval:
            return
        if inner is None:
            if (epoch + 1) % ck_interval != 0 and epoch + 1 != epochs:
                return
            state = dict(state_dict=model.state_dict(), optimizer=optimizer.state_dict(),
                         meta=dict(iter=it, epoch=epoch + 1, mmdet_version='2.20.0-compatible keys'))
            torch.save(state, os.path.join(work_dir, f'epoch_{epoch + 1}.pth'))
        else:
            state = dict(state_dict=model.state_dict(), optimizer=optimizer.state_dict(),
                         meta=dict(iter=it, epoch=epoch, inner_iter=inner, mmdet_version='2.20.0-compatible keys'))
        torch.save(state, os.path.join(work_dir, 'latest.pth'))

    it, t0 = start_iter, time.time()
    t_log, it_log = t0, start_iter          # 'time:' = mean iteration time since the previous log line (mmcv TextLoggerHook)
    # software pipeline, the analogue of the reference's DataLoader workers: a loader thread produces batch i+2 (decode +
    # pinned upload, or the synthetic generator), the pipeline worker augments batch i+1 on its side stream (own numpy
    # stream, seeded like a DataLoader worker: datasets/builder.py:194-199), the main thread trains on batch i
    from concurrent.futures import ThreadPoolExecutor

    def index_lists():
        for epoch in range(start_epoch, epochs):          # a resumed run continues with the epoch after the saved one
            sampler.set_epoch(epoch)
            for k, idx in enumerate(batches(sampler.indices(), bs)):
                if epoch == start_epoch and k < skip_batches:
                    continue                              # trained before the run was interrupted inside this epoch
                yield epoch, k, idx

    loader_stream = torch.cuda.Stream(device=dev)     # decode / upload / synthetic generation beside the training stream

    def load(item):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(loader_stream):
            batch = ds.batch(item[2])
            ready = torch.cuda.Event()
            ready.record()
        return item, batch, ready
    depth = max(1, int(a.loader_depth))
    inner = ds
    while hasattr(inner, 'dataset'):                  # (RepeatDataset and friends)
        inner = inner.dataset
    if hasattr(inner, 'ring_slots'):                  # a pinned batch buffer per load in flight + the ones still uploading
        inner.ring_slots = max(inner.ring_slots, depth + 3)
    loader = ThreadPoolExecutor(depth, thread_name_prefix='oadg-loader')
    wseed = seed + rank + 1000
    todo = index_lists()
    # `depth` batches in flight, consumed in order (the futures' list is FIFO: the training order is the sampler's)
    pending_load = []
    for _ in range(depth):
        nxt = next(todo, None)
        if nxt is not None:
            pending_load.append(loader.submit(load, nxt))
    staged = []          # (epoch, k, prefetched batch)

    def advance():
        if pending_load:
            item, batch, ready = pending_load.pop(0).result()
            nxt = next(todo, None)
            if nxt is not None:
                pending_load.append(loader.submit(load, nxt))
            staged.append((item[0], item[1], pipe.prefetch(*batch, worker_seed=wseed, ready=ready)))
    advance()
    last_epoch, done_in_epoch, cut_short = None, 0, False
    while staged:
        epoch, k, handle = staged.pop(0)
        if a.max_iters is not None and it >= a.max_iters:
            cut_short = done_in_epoch < iters_per_epoch
            break
        if last_epoch is not None and epoch != last_epoch:
            save_checkpoint(last_epoch, it)
        last_epoch = epoch
        sched.set(epoch, it)
        data = handle.get()
        advance()                        # batch i+1 is augmented while this step runs
        out = engine.step(data)
        it += 1
        done_in_epoch = k + 1
        if rank == 0 and it % interval == 0:
            lv = {n: float(v) for n, v in out['log_vars'].items()}      # (the interval's one device synchronisation)
            now = time.time()
            print(f'Epoch [{epoch + 1}][{k + 1}/{iters_per_epoch}] lr: {optimizer.param_groups[0]["lr"]:.3e} '
                  f'time: {(now - t_log) / (it - it_log):.4f} ' +
                  ', '.join(f'{n}: {v:.4f}' for n, v in lv.items()), flush=True)
            if os.environ.get('OADG_TRAIN_TRACE') == '1':      # diagnostic: what an interval's outliers consist of
                ms = torch.cuda.memory_stats()
                print(f'  trace: repeated steps {getattr(engine, "respeculated", 0)}, device segments allocated '
                      f'{ms["segment.all.allocated"]}, reserved {ms["reserved_bytes.all.current"] >> 20} MB, '
                      f'alloc retries {ms["num_alloc_retries"]}', flush=True)
            t_log, it_log = now, it
    if last_epoch is not None:
        save_checkpoint(last_epoch, it, inner=done_in_epoch if cut_short else None)
    loader.shutdown(wait=False, cancel_futures=True)


if __name__ == '__main__':
    main()
