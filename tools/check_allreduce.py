#!/usr/bin/env python
"""Are the gradients that come out of the OVERLAPPED all-reduce the ones that went in?  (round 6.)

    python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 tools/check_allreduce.py [--steps 10]

Every rank runs real training steps of the benchmarked configuration on its own synthetic images through
``apis.TrainEngine`` with the data-parallel reducer - the bucket all-reduces are issued from the autograd hooks while this
library's matrix kernels are still running, which is where RCCL's kernels execute BESIDE them.  Right before each
collective the rank's own bucket is cloned; after the step, on an otherwise idle device, the clones are all-reduced again and
compared with what the overlapped all-reduce left in the flat buffer:

    * "ranks agree": every rank holds the same bits (an all-gather of a checksum per bucket);
    * "overlapped == quiet": the overlapped result equals the all-reduce of the same inputs made while nothing else runs.

Why: profiles/r06_packed_fp32_hazard.txt - kernels that hold packed fp32 instructions can return wrong lanes when matrix-
instruction waves share their SIMD; RCCL's gfx950 code holds such instructions, and one GPU cannot test it (a world of one
does not reduce, RCCL refuses two ranks on one device).  ``OADG_REDUCE_OVERLAP=0`` is the remedy if this reports a
difference.  ``--backend gloo --share-device``: the plumbing of this script on ONE device (gloo reduces on the host: it
validates the script, not RCCL).  Reference: mmdet/apis/train.py:113-121 (MMDistributedDataParallel), tools/dist_train.sh:7-9.
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--backend', default='nccl')
    ap.add_argument('--share-device', action='store_true', help='every rank on cuda:0 (with --backend gloo: a plumbing run)')
    ap.add_argument('--height', type=int, default=1024)
    ap.add_argument('--width', type=int, default=2048)
    ap.add_argument('--batch', type=int, default=4)
    a = ap.parse_args()
    import oadg_amd  # noqa: F401
    from oadg_amd import Config, build_detector
    from oadg_amd.apis import TrainEngine, build_optimizer, init_dist, pin_rank_to_cores, set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    if a.share_device:
        torch.cuda.set_device(0)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(a.backend)
    else:
        init_dist('pytorch', backend=a.backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    local = 0 if a.share_device else int(os.environ.get('LOCAL_RANK', 0))
    pin_rank_to_cores(int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    dev = torch.device('cuda', local)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    set_random_seed(0)
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    det.log_vars_on_host = False
    eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), distributed=True, amp_dtype=torch.bfloat16)
    red = eng.reducer
    assert red is not None and red.overlap, 'this check is about the overlapped reducer (unset OADG_REDUCE_OVERLAP / OADG_USE_TORCH_DDP)'
    ds = SyntheticCityscapes(img_shape=(a.height, a.width), num_boxes=20, num_classes=8, seed=100 + rank, device=dev)
    pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
    set_random_seed(1 + rank)
    snaps = []
    red.pre_collective = lambda r, b: snaps.append((b['start'], b['end'], r.flat[b['start']:b['end']].clone()))
    avg = dist.get_backend() == 'nccl'
    bad_quiet = bad_ranks = buckets = 0
    worst = 0.0
    for step in range(a.steps):
        data = pipe(*ds.batch([(step * a.batch + i) % len(ds) for i in range(a.batch)]))
        snaps.clear()
        # (the optimizer step would consume the gradients: look at them through a hook of finish()'s result instead)
        orig_step = eng.optimizer.step
        kept = {}

        def keep():
            kept['flat'] = red.flat.clone()
        eng.optimizer.step = lambda *x, **k: (keep(), orig_step(*x, **k))[1]
        try:
            eng.step(data)
        finally:
            eng.optimizer.step = orig_step
        torch.cuda.synchronize()
        dist.barrier()                                   # every rank idle: the quiet all-reduce of the same inputs
        for start, end, local_bucket in snaps:
            quiet = local_bucket.clone()
            dist.all_reduce(quiet, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM)
            if not avg:
                quiet.div_(world)
            got = kept['flat'][start:end]
            buckets += 1
            if not torch.equal(got, quiet):
                bad_quiet += 1
                worst = max(worst, float((got - quiet).abs().max() / quiet.abs().max().clamp_min(1e-30)))
            h = torch.stack([got.double().sum(), got.view(torch.int32).to(torch.int64).sum().double()])
            hs = [torch.empty_like(h) for _ in range(world)]
            dist.all_gather(hs, h)
            if any(not torch.equal(x, hs[0]) for x in hs):
                bad_ranks += 1
        torch.cuda.synchronize()
    res = torch.tensor([bad_quiet, bad_ranks, buckets], dtype=torch.int64, device=dev)
    dist.all_reduce(res)
    if rank == 0:
        print(f'check_allreduce: {world} ranks ({dist.get_backend()}), {a.steps} steps, {int(res[2]) // world} buckets per rank: '
              f'overlapped != quiet in {int(res[0])} bucket-checks (all ranks; worst relative difference {worst:.3g} on rank 0), '
              f'ranks disagree in {int(res[1])}', flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if int(res[0]) or int(res[1]) else 0)


if __name__ == '__main__':
    main()
