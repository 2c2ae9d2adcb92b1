"""cProfile of the host side of one steady-state bench step (where does the Python time go?)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oadg_amd  # noqa: E402
from oadg_amd import Config, build_detector, hip_conv  # noqa: E402
from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed  # noqa: E402
from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    hip_conv.enable()
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    set_random_seed(0)
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    det.log_vars_on_host = False
    ddp = os.environ.get('OADG_FORCE_DDP') == '1'
    if ddp:
        from oadg_amd.apis import init_dist
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        init_dist('pytorch', backend='nccl')
    eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), distributed=ddp, amp_dtype=torch.bfloat16)
    ds = SyntheticCityscapes(device=dev)
    pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
    batches = [ds.batch(range(i * 4, i * 4 + 4)) for i in range(3)]
    nxt = pipe.prefetch(*batches[0])
    for i in range(4):
        data = nxt.get()
        eng.step(data)
        nxt = pipe.prefetch(*batches[(i + 1) % 3])
    torch.cuda.synchronize()
    if '--callers' in sys.argv:
        # which lines of this package call the small tensor ops?  (python-side calls only; autograd's are not seen)
        import collections
        import traceback
        counts = collections.Counter()

        def wrap(owner, name):
            orig = getattr(owner, name)

            def shim(*a, **k):
                for fr in reversed(traceback.extract_stack(limit=12)[:-1]):
                    if 'oa-dg_amd' in fr.filename:
                        counts[(name, os.path.basename(fr.filename), fr.lineno)] += 1
                        break
                return orig(*a, **k)
            setattr(owner, name, shim)
        for nm in ('cat', 'stack', 'zeros', 'ones', 'full', 'arange', 'where', 'tensor', 'as_tensor', 'empty', 'zeros_like',
                   'ones_like', 'full_like', 'gather', 'sort', 'topk', 'cumsum', 'nonzero', 'clamp', 'exp', 'log', 'abs'):
            wrap(torch, nm)
        for nm in ('new_zeros', 'new_full', 'new_ones', 'new_tensor', 'new_empty', 'float', 'long', 'int', 'to', 'clone',
                   'contiguous', 'sum', 'mean', 'nonzero', 'index_select', 'masked_fill', 'clamp', 'expand', 'repeat',
                   '__getitem__', '__setitem__', '__mul__', '__add__', '__sub__', '__truediv__', 'view', 'reshape',
                   'permute', 'sigmoid', 'softmax', 'index_fill_', 'index_copy_', 'fill_', 'zero_', 'copy_'):
            wrap(torch.Tensor, nm)
        data = nxt.get()
        eng.step(data)
        torch.cuda.synchronize()
        byline = collections.Counter()
        for (nm, f, ln), n in counts.items():
            byline[(f, ln)] += n
        print('---- tensor-op calls per step by source line (top 60)')
        for (f, ln), n in byline.most_common(60):
            ops = ', '.join(f'{nm}x{c}' for (nm, f2, l2), c in counts.items() if (f2, l2) == (f, ln))
            print(f'{n:4d} {f}:{ln}  {ops}')
        return
    if '--small-ops' in sys.argv:
        # who launches the small element-wise kernels?  aten op -> innermost frame of this package, per step
        from torch.profiler import ProfilerActivity, profile
        import collections
        with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as p:
            data = nxt.get()
            eng.step(data)
            torch.cuda.synchronize()
        want = ('aten::zero_', 'aten::fill_', 'aten::cat', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::mul',
                'aten::index', 'aten::index_select', 'aten::gather', 'aten::where', 'aten::sub', 'aten::div')
        agg = collections.Counter()
        for e in p.events():
            if e.name in want:
                fr = [f for f in (e.stack or []) if 'oa-dg_amd' in f or 'oadg_amd' in f]
                where = fr[0].split('/')[-1] if fr else ((e.stack or ['autograd engine'])[0].split('/')[-1])
                agg[(e.name, where[:70], str(e.input_shapes)[:50])] += 1
        for (name, where, shp), n in agg.most_common(70):
            print(f'{n:4d} {name:18s} {where:70s} {shp}')
        print('---- element-wise ops on large tensors (>= 1M elements)')
        big = collections.Counter()
        for e in p.events():
            if e.name.startswith('aten::') and e.name not in ('aten::empty', 'aten::view', 'aten::empty_like', 'aten::as_strided',
                                                              'aten::permute', 'aten::reshape', 'aten::slice', 'aten::narrow',
                                                              'aten::select', 'aten::expand', 'aten::detach', 'aten::alias',
                                                              'aten::empty_strided', 'aten::transpose', 'aten::contiguous',
                                                              'aten::to', 'aten::_to_copy', 'aten::unbind', 'aten::split',
                                                              'aten::chunk', 'aten::zeros', 'aten::clone', 'aten::result_type'):
                n = 0
                for shp in (e.input_shapes or []):
                    if shp:
                        m = 1
                        for d in shp:
                            m *= d
                        n = max(n, m)
                if n >= 1_000_000:
                    big[(e.name, str(e.input_shapes)[:90])] += 1
        for (name, shp), n in sorted(big.items(), key=lambda kv: -kv[1]):
            print(f'{n:4d} {name:28s} {shp}')
        return
    if '--copies' in sys.argv:
        # dtype / layout copies (aten::_to_copy, aten::clone, aten::contiguous, aten::copy_) of >= 256k elements per step
        from torch.profiler import ProfilerActivity, profile
        import collections
        with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as p:
            data = nxt.get()
            eng.step(data)
            torch.cuda.synchronize()
        agg = collections.Counter()
        for e in p.events():
            if e.name in ('aten::_to_copy', 'aten::clone', 'aten::contiguous', 'aten::copy_'):
                n = 0
                for shp in (e.input_shapes or []):
                    m = 1
                    for d in (shp or [0]):
                        m *= d
                    n = max(n, m)
                if n >= 262144:
                    agg[(e.name, str(e.input_shapes)[:80])] += 1
        for (name, shp), n in sorted(agg.items(), key=lambda kv: -kv[1]):
            print(f'{n:4d} {name:18s} {shp}')
        return
    if '--torchprof' in sys.argv:
        from torch.profiler import ProfilerActivity, profile
        # the bench's arrangement: the pipeline of batch i+1 runs in the worker thread while this thread runs step i
        thread = '--no-thread' not in sys.argv
        if thread:
            nxt.get()
            nxt = pipe.prefetch(*batches[0], worker_seed=1000)
            for i in range(4):
                data = nxt.get()
                nxt = pipe.prefetch(*batches[(i + 1) % 3], worker_seed=1000)
                eng.step(data)
            torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
            for i in range(2):
                data = nxt.get()
                if thread:
                    nxt = pipe.prefetch(*batches[(i + 1) % 3], worker_seed=1000)
                eng.step(data)
                if not thread:
                    nxt = pipe.prefetch(*batches[(i + 1) % 3])
            torch.cuda.synchronize()
        p.export_chrome_trace('/tmp/trace.json')
        import collections
        import json
        ev = json.load(open('/tmp/trace.json'))['traceEvents']
        ks = [e for e in ev if e.get('cat') in ('kernel', 'gpu_memcpy', 'gpu_memset')]
        per = collections.defaultdict(list)
        for e in ks:
            per[e['args'].get('stream')].append((e['ts'], e['ts'] + e['dur'], e['name']))
        t0 = min(e['ts'] for e in ks)
        t1 = max(e['ts'] + e['dur'] for e in ks)
        print(f'GPU span {(t1 - t0) / 1e3:.2f} ms for 2 steps')
        for st, lst in per.items():
            busy = sum(b - a for a, b, _ in lst)
            print(f'stream {st}: {len(lst)} launches, busy {busy / 1e3:.2f} ms')
        main = max(per.values(), key=lambda l: sum(b - a for a, b, _ in l))
        for lo, hi in ((0, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1e9)):
            sel = [b - a for a, b, _ in main if lo <= b - a < hi]
            print(f'  main-stream kernels {lo}-{hi} us: {len(sel) / 2:.0f} per step, {sum(sel) / 2e3:.2f} ms per step')
        small = collections.defaultdict(lambda: [0, 0.0])
        for a_, b_, nm in main:
            if b_ - a_ < 10:
                small[nm[:90]][0] += 1
                small[nm[:90]][1] += b_ - a_
        print('kernels < 10 us on the main stream, per step:')
        for nm, (c, t) in sorted(small.items(), key=lambda kv: -kv[1][1])[:28]:
            print(f'  {c / 2:6.0f} x  {t / 2e3:6.3f} ms  {nm}')
        # launches and device time per annotated section (launch API calls inside the section's host interval;
        # the backward section covers the autograd thread's launches)
        secs = [e for e in ev if e.get('cat') == 'user_annotation' and e['name'].startswith('sec:')]
        launches = sorted((e['ts'], e['args'].get('correlation')) for e in ev
                          if e.get('cat') in ('cuda_runtime', 'cuda_driver') and 'aunch' in e['name'])
        kdur = {e['args'].get('correlation'): e['dur'] for e in ks}
        agg = collections.defaultdict(lambda: [0, 0.0])
        for sct in secs:
            for ts, corr in launches:
                if sct['ts'] <= ts <= sct['ts'] + sct['dur']:
                    agg[sct['name']][0] += 1
                    agg[sct['name']][1] += kdur.get(corr, 0.0)
        print('sections (per step): launches, device ms')
        for nm, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f'  {nm:28s} {c / 2:7.0f} {t / 2e3:8.2f}')
        # idle gaps on the busiest stream
        main = max(per.values(), key=lambda l: sum(b - a for a, b, _ in l))
        main.sort()
        gaps = [(main[i + 1][0] - main[i][1], main[i][2][:60], main[i + 1][2][:60]) for i in range(len(main) - 1)]
        print('idle on the main stream: %.2f ms total; gaps > 50 us:' % (sum(g for g, _, _ in gaps if g > 0) / 1e3))
        for g, a_, b_ in sorted(gaps, reverse=True)[:25]:
            print(f'  {g:8.1f} us  after {a_}  before {b_}')
        for i in range(len(main) - 1):          # context of the largest gaps: three kernels either side
            if main[i + 1][0] - main[i][1] > 300:
                print('  gap %.0f us at +%.2f ms:' % (main[i + 1][0] - main[i][1], (main[i][1] - t0) / 1e3))
                for j in range(max(0, i - 3), min(len(main), i + 5)):
                    print('     %s %8.1f us  %s' % ('>>' if j == i + 1 else '  ', main[j][1] - main[j][0], main[j][2][:100]))
        # which host section launched the kernel that ENDS each gap?  (kernel -> launch call by correlation id -> section)
        launch_ts = {e['args'].get('correlation'): e['ts'] for e in ev
                     if e.get('cat') in ('cuda_runtime', 'cuda_driver') and 'aunch' in e['name']}
        kcorr = {(e['ts'], e['name'][:60]): e['args'].get('correlation') for e in ks}
        per_sec = collections.defaultdict(float)
        for i in range(len(main) - 1):
            g = main[i + 1][0] - main[i][1]
            if g <= 5:
                continue
            ts = launch_ts.get(kcorr.get((main[i + 1][0], main[i + 1][2][:60])))
            name = 'unknown'
            if ts is not None:
                inside = [sct for sct in secs if sct['ts'] <= ts <= sct['ts'] + sct['dur']]
                name = min(inside, key=lambda q: q['dur'])['name'] if inside else 'outside sections'
            per_sec[name] += g
        if '--timeline' in sys.argv:      # the backward's host-bound stretch: kernels in time order with the gap before each
            bsecs = sorted([q for q in secs if q['name'] == 'sec:backward'], key=lambda q: q['ts'])
            if bsecs:
                b0 = bsecs[-1]['ts']
                first = next((i for i in range(len(main)) if (launch_ts.get(kcorr.get((main[i][0], main[i][2][:60]))) or 0) >= b0), None)
                if first is not None:
                    print('main-stream kernels from the start of the last backward (gap before, duration, name):')
                    for i in range(first, len(main)):
                        gap = main[i][0] - main[i - 1][1]
                        if gap > 15 or i < first + 3:
                            print(f'  #{i - first:4d} {gap:7.1f} {main[i][1] - main[i][0]:7.1f}  after {main[i - 1][2][:50]:50s} before {main[i][2][:60]}')
        # library / ATen kernels of 15 us and more on the main stream, with the section that launched them
        big = collections.defaultdict(lambda: [0, 0.0])
        for a_, b_, nm in main:
            if b_ - a_ >= 15 and ('at::native' in nm or 'rocclr' in nm or 'Cijk' in nm or 'Memcpy' in nm or 'rocprim' in nm):
                ts = launch_ts.get(kcorr.get((a_, nm[:60])))
                sec = 'unknown'
                if ts is not None:
                    inside = [sct for sct in secs if sct['ts'] <= ts <= sct['ts'] + sct['dur']]
                    sec = min(inside, key=lambda q: q['dur'])['name'] if inside else 'outside sections'
                big[(sec, nm[:110])][0] += 1
                big[(sec, nm[:110])][1] += b_ - a_
        print('ATen / library kernels >= 15 us on the main stream, per step:')
        for (sec, nm), (c, t) in sorted(big.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f'  {c / 2:5.1f} x {t / 2e3:6.3f} ms  [{sec}]  {nm}')
        if '--window' in sys.argv:      # the host-bound stretch: every main-stream kernel from the RoI targets to the first
            # backward convolution of the LAST profiled step, in time order (gap before, duration, name)
            starts = [i for i, m in enumerate(main) if 'roi_targets_kernel' in m[2]]
            if starts:
                i0 = starts[-1]
                print('main-stream kernels from roi_targets_kernel on (gap before us, duration us, name):')
                seen_bwd = 0
                for i in range(i0 - 3, len(main)):
                    gap = main[i][0] - main[i - 1][1]
                    print(f'  {gap:7.1f} {main[i][1] - main[i][0]:7.1f}  {main[i][2][:110]}')
                    if 'conv_wgrad' in main[i][2]:
                        seen_bwd += 1
                        if seen_bwd >= 3:
                            break
        print('idle (gaps > 5 us) by the section that launched the kernel after the gap, ms per step:')
        for nm, g in sorted(per_sec.items(), key=lambda kv: -kv[1]):
            print(f'  {nm:28s} {g / 2e3:7.2f}')
        return
    pr = cProfile.Profile()
    pr.enable()
    for i in range(3):
        data = nxt.get()
        eng.step(data)
        nxt = pipe.prefetch(*batches[(i + 1) % 3])
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(45)
    st.sort_stats('tottime').print_stats(25)


if __name__ == '__main__':
    main()
