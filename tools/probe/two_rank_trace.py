"""Two ranks stepping in lockstep on ONE device (gloo barrier before every step, so that their kernels really overlap):
is a training step bit-reproducible there, and if not, where does the difference enter?  (tools/probe, not the product.)

    NRUN=16 HASH_ROI=1 ROI_TWICE=1 python tools/probe/two_rank_trace.py plain      # or: reducer

Every rank repeats the SAME step NRUN times from the same state and compares parameter gradients, the gradients at the
backbone / neck / RoI-head boundaries (tensor hooks) and - with HASH_ROI / HASH_CONV - bit hashes of the operands and
results of RoIAlign's backward and of the data-gradient convolutions; ROI_TWICE runs RoIAlign's backward a second time on
the same operands ("DOUBLE ... A==B") and prints where two results differ ("RACE ...").  OADG_HIP_LIB selects a library
built with other flags.  This is the script that found the packed-fp32 hazard of round 6 (csrc/Makefile,
profiles/r06_packed_fp32_hazard.txt)."""
import os, sys, copy, numpy as np, torch
ROOT = os.getcwd()
def worker(rank, port, q, mode):
    sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests'); sys.path.insert(0, ROOT + '/tests/golden')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=2)
    import oadg_amd
    from oadg_amd import Config, hip_conv
    from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    from test_model_parity import CFG, build_and_load
    dev = torch.device('cuda:0')
    cfg = Config.fromfile(CFG)
    set_random_seed(0)
    det = build_and_load(dev).to(memory_format=torch.channels_last).train()
    det.log_vars_on_host = False
    det.local_log_vars = True
    ds = SyntheticCityscapes(img_shape=(384, 768), num_boxes=8, num_classes=8, box_size=(16, 160), seed=3 + rank, device=dev)
    pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
    set_random_seed(5 + rank)
    batch = pipe(*ds.batch([0, 1]))
    state0 = copy.deepcopy(det.state_dict())
    names = [n for n, p in det.named_parameters() if p.requires_grad]
    caps = {}
    def tap(mod, label):
        orig = mod.forward
        def fwd(*a, **k):
            out = orig(*a, **k)
            outs = out if isinstance(out, (tuple, list)) else (out,)
            for i, o in enumerate(outs):
                if torch.is_tensor(o) and o.requires_grad:
                    o.register_hook(lambda g, i=i: caps.__setitem__(f'{label}.{i}', g.detach().clone()))
            return out
        mod.forward = fwd
    tap(det.backbone, 'backbone'); tap(det.neck, 'neck'); tap(det.roi_head.bbox_roi_extractor, 'roialign'); tap(det.roi_head.bbox_head, 'bbox_head')
    orig_nh = hip_conv.narrow_head_levels
    def nh(xs, *a, **k):
        for i, x in enumerate(xs):
            if x.requires_grad:
                x.register_hook(lambda g, i=i: caps.__setitem__(f'rpn_hidden_grad.{i}', g.detach().clone()))
        ys = orig_nh(xs, *a, **k)
        for i, y in enumerate(ys):
            y.register_hook(lambda g, i=i: caps.__setitem__(f'rpn_dy.{i}', g.detach().clone()))
        return ys
    if os.environ.get('TAP_RPN') == '1':
        hip_conv.narrow_head_levels = nh
    from oadg_amd import hip_ops
    trace = []
    orig_dep = hip_ops._deposit
    def dep(tokens, grads):
        out = orig_dep(tokens, grads)
        trace.append(('roi_deposit', tuple(o is None for o in out)))
        return out
    hip_ops._deposit = dep
    orig_nb = hip_conv._NarrowHead.backward
    def nb(ctx, *gys):
        trace.append(('narrow_bwd', tuple((t.extra is not None) if t is not None else None for t in ctx.toks)))
        return orig_nb(ctx, *gys)
    hip_conv._NarrowHead.backward = staticmethod(nb)
    def hsh(t):
        if t is None: return None
        t = t.detach()
        if t.dtype == torch.bfloat16: v = t.contiguous(memory_format=torch.channels_last).view(torch.int16) if t.dim() == 4 else t.contiguous().view(torch.int16)
        elif t.dtype == torch.float32: v = t.contiguous().view(torch.int32)
        else: v = t.contiguous()
        v = v.to(torch.int64).flatten()
        return int((v * (torch.arange(v.numel(), device=v.device) % 1021 + 1)).sum().item())
    orig_cf = hip_conv.conv_forward
    def cf(x, w, bias, residual, stride, pad, dil, relu, variant=0, mask=None, want_colsum=False, mask_bits=None, bits_out=None, res_up=False):
        out = orig_cf(x, w, bias, residual, stride, pad, dil, relu, variant=variant, mask=mask, want_colsum=want_colsum, mask_bits=mask_bits, bits_out=bits_out, res_up=res_up)
        if os.environ.get('HASH_CONV') == '1' and (residual is not None or want_colsum):
            y = out[0] if want_colsum else out
            trace.append(('conv', tuple(x.shape), tuple(w.shape), hsh(x), hsh(w), hsh(residual), hsh(mask), hsh(mask_bits), hsh(y)))
        return out
    hip_conv.conv_forward = cf
    orig_rb = hip_ops._RoIAlignFPN.backward
    saved_in = {}
    def rb(ctx, gout):
        rois, order, rng_ = ctx.saved_tensors
        if os.environ.get('GOUT_CLONE') == '1':
            gout = gout.clone()
        toks, ctx.tokens = ctx.tokens, None          # return the grads instead of depositing, then deposit by hand
        pre = gout.clone()                            # same stream, no sync: what a consumer launched NOW sees
        outs = orig_rb(ctx, gout)
        torch.cuda.synchronize()
        if not torch.equal(pre, gout):
            d_ = (pre.float() - gout.float()).abs()
            print(f'GOUT-CHANGED rank {rank}: {int((d_ > 0).sum())} elements differ between a clone enqueued before the kernel and gout after; stream {torch.cuda.current_stream()}', flush=True)
        ctx.tokens = toks
        grads = list(outs[7:])
        trace.append(('roi_bwd', hsh(gout), hsh(rois), hsh(order), hsh(rng_), tuple(hsh(g) for g in grads)))
        if os.environ.get('ROI_TWICE') == '1':
            ctx.tokens = None
            again = list(orig_rb(ctx, gout)[7:])
            ctx.tokens = toks
            print(f'DOUBLE rank {rank}: A==B {[bool(torch.equal(x, y)) for x, y in zip(grads, again)]}', flush=True)
            for l, (g1, g2) in enumerate(zip(grads, again)):
                if False:
                  saved_in['ref_done'] = True
                  try:
                      sys.path.insert(0, ROOT)
                      from oracle import roi_align as ora
                      shapes_ = ctx.meta[0]
                      with torch.enable_grad():
                          fc = [torch.zeros(s_, dtype=torch.float64, requires_grad=True) for s_ in shapes_]
                          o_ = ora.roi_align_fpn(fc, rois.cpu(), 7, (4, 8, 16, 32))
                          refg = torch.autograd.grad(o_, fc, gout.detach().cpu().to(o_.dtype), allow_unused=True)
                      r_ = refg[l]
                      dm_ = (g1.float() - g2.float()).abs().cpu()
                      idx_ = dm_.amax(1).nonzero()
                      for (n_, y_, x_) in idx_.tolist()[:6]:
                          chs = dm_[n_, :, y_, x_].nonzero().flatten()[:4]
                          print(f'REF rank {rank} lvl {l} px {(n_, y_, x_)} ch {chs.tolist()} A {g1[n_, :, y_, x_].float().cpu()[chs].tolist()} B {g2[n_, :, y_, x_].float().cpu()[chs].tolist()} ref {r_[n_, chs, y_, x_].tolist()}', flush=True)
                  except Exception as e_:
                    print('REF-ERR', repr(e_)[:300], flush=True)
                if not torch.equal(g1, g2):
                    dm = (g1.float() - g2.float()).abs()
                    idx = dm.amax(1).nonzero()
                    tiles = sorted(set((int(a_), int(b_) // 8, int(c_) // 8) for a_, b_, c_ in idx.tolist()))
                    ch = dm.amax((0, 2, 3)).nonzero().flatten().tolist()
                    px = idx[0].tolist()
                    v1 = g1[px[0], :, px[1], px[2]].float(); v2 = g2[px[0], :, px[1], px[2]].float()
                    rois_l = int(((rng_.numel() > 0)))
                    print(f'RACE rank {rank} level {l}: {idx.shape[0]} px in tiles {tiles[:8]} (n,ty,tx) n_tiles {len(tiles)} channels {ch[:6]}..{ch[-3:]} n_ch {len(ch)} max {float(dm.max()):.4g} of {float(g1.float().abs().max()):.4g}; px {px} v1 {v1[ch[:4]].tolist()} v2 {v2[ch[:4]].tolist()}', flush=True)
        if not saved_in:
            saved_in.update(gout=gout.detach().clone(), rois=rois.clone(), shapes=ctx.meta[0])
        return tuple(outs[:7]) + tuple(hip_ops._deposit(toks, grads))
    if os.environ.get('HASH_ROI') == '1':
        hip_ops._RoIAlignFPN.backward = staticmethod(rb)
    def run():
        trace.clear()
        det.load_state_dict(state0)
        eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), distributed=(mode == 'reducer'), amp_dtype=torch.bfloat16)
        eng.speculative_sampling = False
        hip_conv.refresh_prepared()
        set_random_seed(11 + rank)
        caps.clear()
        dist.barrier()
        out = eng.step({k: (list(v) if isinstance(v, list) else v) for k, v in batch.items()})
        torch.cuda.synchronize()
        g = {n: p.grad.detach().clone() for n, p in det.named_parameters() if p.requires_grad}
        if eng.reducer is not None: eng.reducer.close()
        return g, dict(caps), {k: float(v) for k, v in out['log_vars'].items()}, list(trace)
    rs = [run() for _ in range(int(os.environ.get("NRUN", "6")))]
    lines = []
    for i in range(1, len(rs)):
        a, b = rs[0], rs[i]
        bad = [n for n in names if not torch.equal(a[0][n], b[0][n])]
        capbad = [(k, round(float((a[1][k].float() - b[1][k].float()).abs().max() / a[1][k].float().abs().max().clamp_min(1e-30)), 6)) for k in a[1] if not torch.equal(a[1][k], b[1][k])]
        first = next(((j, x, y) for j, (x, y) in enumerate(zip(a[3], b[3])) if x != y), None)
        lines.append(f'rank {rank} run {i}: trace equal {a[3] == b[3]} n={len(b[3])} FIRST {first} logs equal {a[2] == b[2]} params differing {len(bad)} {bad[:4]} | taps differing {capbad} of {sorted(a[1])}')
    if saved_in and rank == 0 and os.environ.get('HASH_ROI') == '1':
        torch.save({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in saved_in.items()}, ROOT + '/gpurun_out/roi_bwd_inputs.pt')
    q.put('\n'.join(lines))
    dist.barrier(); dist.destroy_process_group()
if __name__ == '__main__':
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn'); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 25311, q, sys.argv[1])) for r in range(2)]
    [p.start() for p in ps]
    for _ in ps: print(q.get(timeout=500))
    [p.join() for p in ps]
