"""Training wiring (mmdet/apis/train.py:19-212, tools/train.py:92-207): seeds, optimizer, LR schedule,
data-parallel wrap, iteration loop.  One process per GPU; gradients are all-reduced by torch DDP over RCCL
(``backend='nccl'`` is RCCL on ROCm) in buckets that overlap the backward pass."""
import os
import random
import time

import numpy as np
import torch
from torch.profiler import record_function as _rf
import torch.distributed as dist

from .detectors import integrate_data


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_dist(launcher='pytorch', backend='nccl', **kwargs):
    """tools/train.py:119-132 / mmcv init_dist('pytorch'): env:// rendezvous, one GPU per process."""
    if launcher not in ('pytorch',):
        raise NotImplementedError(f'launcher {launcher}')
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if torch.cuda.is_available() and backend == 'nccl':
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group(backend=backend, **kwargs)


def init_random_seed(seed=None, device='cuda'):
    """apis/train.py:19-49: rank 0's seed is broadcast so every rank shares it."""
    if seed is not None:
        return seed
    rank, world = get_dist_info()
    seed = np.random.randint(2 ** 31)
    if world == 1:
        return seed
    t = torch.tensor(seed if rank == 0 else 0, dtype=torch.int32, device=device)
    dist.broadcast(t, src=0)
    return t.item()


def set_random_seed(seed, deterministic=False):
    """apis/train.py:52-68."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if deterministic:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


def physical_cores(cpus):
    """one logical CPU per physical core of ``cpus`` (the lowest-numbered hardware thread of each sibling set, from sysfs;
    without sysfs: every CPU), sorted - a rank's threads should not share a core's two hardware threads with each other or,
    worse, with another rank (EPYC numbers the second threads of cores 0 .. n-1 as CPUs n .. 2n-1)"""
    cpus = sorted(cpus)
    first = {}
    for c in cpus:
        try:
            with open(f'/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list') as f:
                txt = f.read().strip()
            sib = []
            for part in txt.split(','):
                lo, _, hi = part.partition('-')
                sib += list(range(int(lo), int(hi or lo) + 1))
            key = min(sib)
        except (OSError, ValueError):
            key = c
        first.setdefault(key, c)
    return sorted(first.values())


def _cpu_busy_jiffies():
    """{cpu id: busy jiffies since boot} from /proc/stat (user + nice + system + irq + softirq + steal)"""
    out = {}
    try:
        with open('/proc/stat') as f:
            for line in f:
                if line.startswith('cpu') and line[3:4].isdigit():
                    p = line.split()
                    v = [int(t) for t in p[1:9]]
                    out[int(p[0][3:])] = v[0] + v[1] + v[2] + v[5] + v[6] + v[7]
    except (OSError, ValueError):
        return {}
    return out


def quietest_window(cores, want, probe_s=0.1):
    """index of the window of ``want`` consecutive entries of ``cores`` (physical cores, aligned to multiples of ``want`` -
    a CCD) that was least busy over the next ``probe_s`` seconds, both hardware threads counted; 0 when /proc/stat cannot
    tell.  OADG_AFFINITY_PROBE=0: always the first window."""
    if os.environ.get('OADG_AFFINITY_PROBE', '1') != '1':
        return 0
    a = _cpu_busy_jiffies()
    if not a:
        return 0
    time.sleep(probe_s)
    b = _cpu_busy_jiffies()
    sib = {}
    for c in cores:
        try:
            with open(f'/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list') as f:
                txt = f.read().strip()
            ids = []
            for part in txt.split(','):
                lo, _, hi = part.partition('-')
                ids += list(range(int(lo), int(hi or lo) + 1))
            sib[c] = ids
        except (OSError, ValueError):
            sib[c] = [c]
    busy = [(sum(b.get(t, 0) - a.get(t, 0) for c in cores[s0:s0 + want] for t in sib[c]), s0)
            for s0 in range(0, len(cores) - want + 1, want)]
    floor = min(v for v, _ in busy)
    # the FIRST window that is (nearly) as quiet as the quietest: 3 jiffies = 30 ms of CPU time in the probe over the
    # window's hardware threads is noise; early windows keep a single rank on socket 0 unless it is occupied
    return next(s0 for v, s0 in busy if v <= floor + 3)


_RANK_CORES = dict(mine=None, slice=None)      # what pin_rank_to_cores chose for this process (spare_cores reads it)


def spare_cores(n):
    """up to ``n`` physical cores of this rank's slice of the host that its pinned threads do NOT run on (the decode
    workers of a real-file dataset: a 1024 x 2048 PNG is tens of ms of inflate, which must not sit on the CCD whose L3 the
    launch threads share), or None when the rank is not pinned / its slice has no spare core"""
    mine, sl = _RANK_CORES['mine'], _RANK_CORES['slice']
    if not mine or not sl:
        return None
    rest = [c for c in sl if c not in set(mine)]
    return rest[:max(0, int(n))] or None


def pin_rank_to_cores(local_rank, world, cores_per_rank=8):
    """Every rank runs a main thread (~20 ms of launch work per step), the autograd thread, the pipeline worker and its
    planner threads.  Left to the scheduler on a 2 x 64-core host they wander over both sockets: the SAME binary needed
    18 - 27 ms of host time per step from process to process and the first step after a synchronisation 31 - 35 ms
    (round 5, five boxes: most of the run-to-run spread of this benchmark).  Pinned to a compact set of cores - one CCD's
    worth, ``cores_per_rank`` = 8 physical cores, first hardware thread only - it needs 15 - 18 ms every time and the
    measured step is 27.0 - 27.25 ms instead of 27.1 - 27.6 (29 on some boxes).  The host's physical cores are split into
    ``world`` contiguous slices (ranks never share a core; on a two-socket node ranks 0 .. N/2-1 stay on socket 0) and a rank
    takes the first cores of its slice (os.sched_setaffinity: threads started later inherit it); torch's intra-op pool is
    sized to it.  Used by bench.py and tools/train.py (``--no-cpu-affinity`` / OADG_BENCH_NO_AFFINITY=1: nothing is pinned).
    Returns a description, or None when nothing was pinned."""
    if not hasattr(os, 'sched_setaffinity'):
        return None
    cores = sorted(os.sched_getaffinity(0))
    phys = physical_cores(cores)
    per = len(phys) // max(world, 1)
    if per < 1:
        return None
    want = max(1, int(cores_per_rank))
    # (LOCAL_RANK set without LOCAL_WORLD_SIZE gives local_rank >= world: wrap around instead of an empty slice)
    local_rank = int(local_rank) % max(world, 1)
    mine_slice = phys[local_rank * per:(local_rank + 1) * per]
    # which `want` consecutive cores of the slice?  The quietest ones: another tenant of the host may be sitting on the first
    # ones (a pinned rank cannot walk away from a noisy neighbour: the same box measured 27.2 - 29.9 ms per step while
    # something else ran on CPUs 0-7) - 100 ms of /proc/stat decide.
    start = quietest_window(mine_slice, want) if len(mine_slice) > want else 0
    mine = mine_slice[start:start + want]
    if not mine:
        return None
    try:
        os.sched_setaffinity(0, mine)
    except OSError:                  # (a cpuset that changed under us, a container that forbids it: run unpinned)
        return None
    _RANK_CORES['mine'], _RANK_CORES['slice'] = list(mine), list(mine_slice)
    torch.set_num_threads(max(1, min(len(mine), 16)))
    return (f'{len(mine)} of {len(phys)} physical cores ({len(cores)} CPUs) per rank, slices of {per} '
            f'(rank {local_rank}: CPUs {mine[0]}-{mine[-1]})')


_SGD_TENSOR = np.dtype([('param', np.uint64), ('grad', np.uint64), ('momentum', np.uint64), ('numel', np.int64),
                        ('first_block', np.int64), ('first_step', np.int32), ('pad', np.int32)])      # oadg_sgd_tensor


def _same_layout(a, b):
    """equal strides on every dimension that has more than one element (a size-1 dimension's stride is arbitrary:
    a [K, C, 1, 1] channels_last weight and its gradient need not agree there)"""
    return all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


class FusedSGD(torch.optim.SGD):
    """``torch.optim.SGD`` (the reference's optimizer, schedule_1x.py) whose ``step()`` is ONE launch of csrc/optim.hip over
    all parameters instead of torch's four multi-tensor passes (~13 launches, 0.49 ms per step on R50-FPN): same
    arithmetic operation for operation (bit-identical parameters and momentum buffers, tests/test_hip_optim.py), same
    ``state`` / ``state_dict`` (``momentum_buffer`` per parameter), same ``param_groups``.  Anything the kernel does not
    cover - a closure, nesterov, dampening, maximize, momentum 0, CPU / non-fp32 / differently laid out tensors - takes
    ``torch.optim.SGD.step`` unchanged."""

    FUSED = os.environ.get('OADG_FUSED_SGD', '1') == '1'

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._tables = {}
        self._slot = 0

    def _eligible(self, group):
        if group['momentum'] <= 0 or group['dampening'] != 0 or group['nesterov'] or group.get('maximize', False):
            return None
        ps = [p for p in group['params'] if p.grad is not None]
        f32 = torch.float32
        for p in ps:
            g = p.grad
            if g.dtype is not f32 or g.is_sparse or g.shape != p.shape or g.device != p.device or \
                    not (g.stride() == p.stride() or _same_layout(g, p)):
                return None
            st = self.state.get(p)
            b = st.get('momentum_buffer') if st is not None else None
            # what does not change from step to step - the parameter's own properties, its momentum buffer's layout - is
            # checked once per (parameter, buffer): 160 parameters x 12 attribute reads were 0.4 ms of every step
            if b is not None and not (b.stride() == p.stride() or _same_layout(b, p)):
                return None
            if getattr(p, '_oadg_sgd_checked', None) == (0 if b is None else id(b)):
                continue
            if not (p.is_cuda and p.dtype is f32):
                return None
            if b is not None and not (b.dtype is f32 and b.device == p.device and b.shape == p.shape):
                return None
            dense = getattr(p, '_oadg_dense', None)
            if dense is None:
                from torch._prims_common import is_non_overlapping_and_dense
                dense = p._oadg_dense = bool(is_non_overlapping_and_dense(p))
            if not dense:
                return None
            p._oadg_sgd_checked = 0 if b is None else id(b)
        return ps

    @torch.no_grad()
    def step(self, closure=None):
        lists = [self._eligible(g) for g in self.param_groups] if (self.FUSED and closure is None) else [None]
        if any(l is None for l in lists):
            return super().step(closure)
        from . import _lib
        L = _lib.lib()
        for gi, (group, ps) in enumerate(zip(self.param_groups, lists)):
            if not ps:
                continue
            key = (gi, tuple((id(p), p.numel()) for p in ps))
            ent = self._tables.get(gi)
            if ent is None or ent['key'] != key:
                tab = np.zeros(len(ps), dtype=_SGD_TENSOR)
                blocks = 0
                for i, p in enumerate(ps):
                    tab[i]['numel'], tab[i]['first_block'] = p.numel(), blocks
                    blocks += int(L.oadg_sgd_blocks(p.numel()))
                dev = ps[0].device
                ent = self._tables[gi] = dict(
                    key=key, tab=tab, blocks=blocks,
                    pinned=[torch.empty(tab.nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)],
                    events=[None, None], dev=torch.empty(tab.nbytes, dtype=torch.uint8, device=dev))
            tab = ent['tab']
            bufs, first = [], []
            for p in ps:
                st = self.state[p]
                b = st.get('momentum_buffer')
                first.append(b is None)
                if b is None:
                    b = st['momentum_buffer'] = torch.empty_like(p.grad)       # (receives g' in the kernel: clone(grad))
                bufs.append(b)
            tab['param'] = [p.data_ptr() for p in ps]          # (every step: ``p.data`` may have been re-pointed)
            tab['grad'] = [p.grad.data_ptr() for p in ps]
            tab['momentum'] = [b.data_ptr() for b in bufs]
            tab['first_step'] = first
            self._slot ^= 1
            pin = ent['pinned'][self._slot]
            if ent['events'][self._slot] is not None:
                ent['events'][self._slot].synchronize()        # the copy out of this staging slot two steps ago is done
            pin.numpy()[:] = tab.view(np.uint8)
            ent['dev'].copy_(pin, non_blocking=True)
            ev = ent['events'][self._slot] = ent['events'][self._slot] or torch.cuda.Event()
            ev.record()
            _lib.check(L.oadg_sgd_step_multi(_lib.ptr(ent['dev']), len(ps), ent['blocks'], float(group['lr']),
                                             float(group['momentum']), float(group['weight_decay']), _lib.stream_ptr()),
                       'oadg_sgd_step_multi')
            # the kernel wrote the parameters behind ATen's back: bump their version counters, as the in-place ops of
            # torch.optim.SGD would, so that everything keyed on them sees the update (hip_conv's bank of prepared
            # convolution weights, autograd's saved-tensor checks); the momentum buffers likewise
            both = ps + bufs
            bump = getattr(torch._C._autograd, '_unsafe_set_version_counter', None)
            try:
                if bump is None:
                    raise TypeError
                bump(both, [t._version + 1 for t in both])
            except TypeError:                   # a torch without the hook, or with its older (Tensor, int) signature:
                torch._foreach_add_(both, 0.0)  # an in-place no-op per tensor bumps the counters the same way
        return None


def build_optimizer(model, cfg):
    cfg = dict(cfg)
    t = cfg.pop('type')
    cfg.pop('paramwise_cfg', None)
    params = [p for p in model.parameters() if p.requires_grad]
    if t != 'SGD':
        raise NotImplementedError(f'optimizer {t} (the named configs use SGD)')
    return FusedSGD(params, **cfg)


class StepLrSchedule:
    """mmcv StepLrUpdaterHook + linear warmup (…_cityscapes.py: step=[1], warmup 500 it, ratio 1e-3)."""

    def __init__(self, optimizer, policy='step', step=(8, 11), gamma=0.1, warmup=None, warmup_iters=0,
                 warmup_ratio=0.1, by_epoch=True, **kwargs):
        assert policy == 'step'
        self.opt, self.steps, self.gamma = optimizer, [step] if isinstance(step, int) else list(step), gamma
        self.warmup, self.warmup_iters, self.warmup_ratio = warmup, warmup_iters, warmup_ratio
        # mmcv LrUpdaterHook.before_run: the base rate lives in the param group ('initial_lr'), so it survives an
        # optimizer.load_state_dict() of a checkpoint saved after a decay step
        self.base = [g.setdefault('initial_lr', g['lr']) for g in optimizer.param_groups]

    def set(self, epoch, it):
        exp = sum(1 for s in self.steps if epoch >= s)
        for g, b in zip(self.opt.param_groups, self.base):
            lr = b * self.gamma ** exp
            if self.warmup == 'linear' and it < self.warmup_iters:
                k = (1 - it / self.warmup_iters) * (1 - self.warmup_ratio)
                lr = lr * (1 - k)
            g['lr'] = lr


class FlatGradReducer:
    """Data-parallel gradient averaging for one process per GPU over RCCL (torch.distributed, backend 'nccl'), in
    place of torch DDP's per-parameter reducer (3.5 ms of host work per step here: ~160 parameters, each copied into
    a bucket and bookkept from an autograd hook, plus DDP's forward pre/post passes):

    * parameters are grouped, in reverse registration order (~ the order their gradients become ready), into a few
      buckets of one flat fp32 buffer;
    * the kernels that PRODUCE a gradient (weight-gradient consumers of the convolutions, the cast / permute passes of
      the RoI head's linears) write it straight into the parameter's slice of the flat buffer (``hip_ops.grad_dest``):
      ``param.grad`` is that slice before the collective starts, nothing is packed (round 4 copied all 166 MB with a
      ``torch.cat(out=...)`` per bucket);
    * a post-accumulate hook per parameter only counts down; the hook that completes a bucket copies the few gradients
      that are not in place (bias vectors, anything a torch operator produced) with one multi-tensor launch and starts
      an asynchronous ``all_reduce(AVG)`` on the bucket - the RoI-head FC weights (the largest tensors, first to be
      ready) reduce while the backbone is still in its backward pass;
    * ``finish()`` (before ``optimizer.step()``) makes the compute stream wait for the reductions and re-points each
      ``param.grad`` at its slice of the reduced buffer.  Parameters that received no gradient contribute zeros;
      buckets are all-reduced strictly in index order, so ranks whose graphs differ still issue matching collectives.

    xGMI is point-to-point (7 links per GPU), so a ring all-reduce moves 2(N-1)/N of the 166 MB per GPU: a handful
    of large buckets keeps each collective bandwidth-bound rather than latency-bound."""

    ALIGN = 64          # floats: parameter slices start on 256-byte boundaries of the flat buffer

    def __init__(self, module, bucket_mb=64, group=None, tail_mb=6, overlap=None):
        self.group = group
        # overlap=False (OADG_REDUCE_OVERLAP=0): every collective is issued by finish(), after the backward pass has been
        # enqueued - no RCCL kernel then runs BESIDE this library's matrix kernels.  A safety valve, off by default: round 6
        # found that kernels holding packed fp32 instructions can return wrong lanes beside matrix-instruction waves
        # (profiles/r06_packed_fp32_hazard.txt); RCCL's gfx950 code holds such instructions (108 v_pk_add_f32, 182
        # v_pk_fma_f32), and whether its fp32 sum is affected cannot be tested on one GPU (a world of one does not reduce).
        self.overlap = (os.environ.get('OADG_REDUCE_OVERLAP', '1') == '1') if overlap is None else bool(overlap)
        self.world = dist.get_world_size(group)
        params = [p for p in module.parameters() if p.requires_grad][::-1]
        for t in list(module.parameters()) + list(module.buffers()):       # identical replicas (DDP does the same)
            dist.broadcast(t.data, src=0, group=group)
        self.params = params
        # every parameter's slice starts on a 256-byte boundary of the flat buffer (<= 63 floats of zero padding each, 40 KB
        # in all; the padding is all-reduced with the rest): the gradient producers write their slices with 16-byte stores
        # and the fused SGD kernel reads them with its vector path - with the slices packed back to back, one 9-element bias
        # left every later slice 4-byte aligned (measured: the step 1.4 ms slower with the gradients produced in place)
        slot = lambda p: (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN  # noqa: E731
        total = sum(slot(p) for p in params)
        dev = params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buckets, cur, size, off = [], [], 0, 0
        limit = bucket_mb * (1 << 20) // 4
        # the parameters that become ready last (the first trainable backbone stage) get a small bucket of their own:
        # its all-reduce is the only one that cannot overlap with the backward pass
        tail, acc = len(params), 0
        while tail > 1 and acc + params[tail - 1].numel() <= tail_mb * (1 << 20) // 4:
            tail -= 1
            acc += params[tail].numel()
        for i, p in enumerate(params):
            cur.append(p)
            size += slot(p)
            if size >= limit or i == tail - 1:
                self.buckets.append(dict(params=cur, start=off, end=off + size))
                off, cur, size = off + size, [], 0
        if cur:
            self.buckets.append(dict(params=cur, start=off, end=off + size))
        self.bucket_of, self.views = {}, {}
        for b in self.buckets:
            o = b['start']
            for p in b['params']:
                self.bucket_of[p] = b
                # the slice carries the parameter's own memory layout (channels_last conv weights included): the reduced
                # gradient then has the parameter's strides and the optimizer's multi-tensor kernels keep their fast path
                self.views[p] = self._like_param(self.flat[o:o + p.numel()], p)
                o += slot(p)
            b['pending'], b['work'] = len(b['params']), None
        self._next = 0                      # index of the next bucket to all-reduce (strict order, see _launch_in_order)
        self._hooks = [p.register_post_accumulate_grad_hook(self._ready) for p in params]
        # gradient producers with kernels of their own write into the bucket slices directly (hip_ops.grad_dest)
        self.in_place_bytes = self.packed_bytes = self.steps = 0
        self.pre_collective = None
        if os.environ.get('OADG_GRAD_SINK', '1') == '1':
            from . import hip_ops
            hip_ops.GRAD_SINK = self.views

    def close(self):
        """detach from the module: remove the hooks and stop offering the bucket slices to the gradient producers"""
        from . import hip_ops
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if hip_ops.GRAD_SINK is self.views:
            hip_ops.GRAD_SINK = None
        hip_ops.sink_reset()

    @staticmethod
    def _dense(t):
        return t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last) or \
            (t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d))

    @classmethod
    def _like_param(cls, flat_slice, p):
        if cls._dense(p):
            return flat_slice.as_strided(p.shape, p.stride())
        return flat_slice.view_as(p)

    @classmethod
    def _memory_order(cls, g, p):
        """the gradient as a flat fp32 vector in the PARAMETER's memory order"""
        if g.stride() == p.stride() and cls._dense(g):
            return g.as_strided((g.numel(),), (1,)).float()
        if cls._dense(p):          # rare: a gradient laid out differently from its parameter
            return torch.empty_strided(p.shape, p.stride(), dtype=torch.float32, device=g.device).copy_(g) \
                .as_strided((g.numel(),), (1,))
        return g.reshape(-1).float()

    def _ready(self, p):
        b = self.bucket_of[p]
        b['pending'] -= 1
        if b['pending'] == 0 and self.overlap:
            self._launch_in_order()

    def _launch_in_order(self, force=False):
        """Collectives are issued STRICTLY in bucket-index order on every rank: a bucket whose gradients are complete
        waits for its predecessors (``force``: finish() launches whatever is left, complete or not).  A rank on which
        some parameter received no gradient this step (its bucket never completes in a hook) therefore still issues the
        same all-reduce sequence as its peers - RCCL matches collectives by issue order, not by buffer."""
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b['pending'] > 0 and not force:
                return
            self._launch(b)
            self._next += 1

    def _launch(self, b):
        from . import hip_conv
        hip_conv.flush_deferred()         # weight / bias gradients whose launches were deferred (hip_conv.DEFER_*) are read now
        flat = self.flat[b['start']:b['end']]
        # Gradients whose producers wrote them straight into their bucket slice (hip_ops.grad_dest: convolution weights, BN
        # scales, the RoI head's linears - all but a few hundred KB of a step's gradients) are in place already; the rest
        # (bias / BN-shift vectors, anything a torch operator produced) are packed by ONE multi-tensor copy.  A parameter
        # without a gradient contributes zeros.
        dst, src = [], []
        for p in b['params']:
            g, v = p.grad, self.views[p]
            if g is None:
                v.zero_()
            elif g.data_ptr() == v.data_ptr() and g.stride() == v.stride() and g.dtype == v.dtype:
                self.in_place_bytes += g.numel() * 4
            else:
                dst.append(v)
                src.append(g)
                self.packed_bytes += g.numel() * 4
        if dst:
            f = getattr(torch, '_foreach_copy_', None)
            if f is not None and all(s_.shape == d_.shape for s_, d_ in zip(src, dst)):
                f(dst, src)
            else:
                for d_, s_ in zip(dst, src):
                    d_.copy_(s_)
        if self.pre_collective is not None:        # tests: observe the bucket right before its all-reduce is issued
            self.pre_collective(self, b)
        b['work'] = dist.all_reduce(flat, op=dist.ReduceOp.AVG if self._has_avg() else dist.ReduceOp.SUM,
                                    group=self.group, async_op=True)

    def _has_avg(self):
        return dist.get_backend(self.group) == 'nccl'

    def finish(self):
        """call after loss.backward(): all gradients averaged over the ranks, ``param.grad`` = slices of one buffer"""
        self._launch_in_order(force=True)      # buckets holding a parameter that had no gradient this step
        for b in self.buckets:
            b['work'].wait()
            if not self._has_avg():
                self.flat[b['start']:b['end']].div_(self.world)
            b['work'], b['pending'] = None, len(b['params'])
        self._next = 0
        self.steps += 1
        for p in self.params:
            p.grad = self.views[p]
        from . import hip_ops
        hip_ops.sink_reset()


class TrainEngine:
    """model.train_step + backward + optimizer step for one process/GPU (the runner's hot loop,
    SURVEY.md 3.1): OptimizerHook semantics = zero_grad, loss.backward(), (no grad clip), step."""

    # backends whose all-reduce of the device sampler's flags stays on the device (see _step); tests/test_two_ranks_one_gpu.py
    # adds 'gloo' to run the lockstep logic with two real ranks on the one GPU a test box has
    SPECULATION_BACKENDS = ('nccl',)

    def __init__(self, model, optimizer, distributed=False, amp_dtype=None, bucket_cap_mb=50,
                 find_unused_parameters=False, grad_clip=None):
        self.module = model
        self.optimizer = optimizer
        self.amp_dtype = amp_dtype
        # OptimizerHook.clip_grads (mmcv/runner/hooks/optimizer.py): clip_grad_norm_(**grad_clip) over the parameters that
        # received a gradient, after the gradient average and before optimizer.step(); None (the reference's schedules) = off
        self.grad_clip = dict(grad_clip) if grad_clip else None
        self.last_grad_norm = None
        # OADG_DEVICE_SAMPLER=0: the RoI sampler draws on the host again (one more host read per step)
        self.speculative_sampling = os.environ.get('OADG_DEVICE_SAMPLER', '1') == '1'
        self.respeculated = 0           # steps repeated on the host path because an image came up short
        self.ddp = self.reducer = None
        # (a prioritised stream for the step - kernels dispatched ahead of the data pipeline's side stream - was measured in
        #  rounds 3-5: no gain beyond the run-to-run spread; the step runs on the caller's stream)
        self.stream = None
        from . import hip_conv
        # bf16 training on the MI355X runs its convolutions on the csrc MFMA kernels (OADG_CONV=miopen: library path)
        if amp_dtype is torch.bfloat16 and next(model.parameters()).is_cuda and \
                os.environ.get('OADG_CONV', 'mfma') == 'mfma':
            hip_conv.enable()
        if distributed and os.environ.get('OADG_USE_TORCH_DDP') != '1':
            self.reducer = FlatGradReducer(model, bucket_mb=int(os.environ.get('OADG_BUCKET_MB', 64)))
        elif distributed:
            from torch.nn.parallel import DistributedDataParallel as DDP
            dev = next(model.parameters()).device
            self.ddp = DDP(model, device_ids=[dev.index] if dev.type == 'cuda' else None,
                           broadcast_buffers=False, find_unused_parameters=find_unused_parameters,
                           bucket_cap_mb=int(bucket_cap_mb),
                           gradient_as_bucket_view=True,
                           static_graph=False)

    def forward_losses(self, data):
        data = integrate_data(data, self.module.train_cfg)
        fwd = self.ddp if self.ddp is not None else self.module
        if self.amp_dtype is not None:
            with torch.autocast('cuda', dtype=self.amp_dtype):
                losses = fwd(**data)
        else:
            losses = fwd(**data)
        return self.module._parse_losses(losses), len(data['img_metas'])

    def step(self, data):
        if self.stream is None:
            return self._step(data)
        # the whole step on the prioritised stream, ordered after the caller's stream on entry and before it on exit
        caller = torch.cuda.current_stream()
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            out = self._step(data)
        caller.wait_stream(self.stream)
        for t in [out['loss']] + [v for v in out['log_vars'].values() if torch.is_tensor(v)]:
            t.record_stream(caller)
        return out

    def _lockstep_world(self):
        """number of ranks that step together (their collectives must match call for call)"""
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.reducer.group if self.reducer is not None else None)

    def _foreign_grad_hooks(self):
        """Deferred weight gradients / column sums hand AccumulateGrad tensors that are FILLED LATER (hip_conv.flush_deferred):
        a tensor hook or post-accumulate hook of someone else's would read them too early.  With such a hook on
        any parameter the step runs its gradient launches immediately; the data-parallel reducer's own hooks flush first."""
        ours = set(id(h) for h in ())
        if self.reducer is not None:
            ours = {h.id for h in self.reducer._hooks}
        params = getattr(self, '_param_list', None)
        if params is None:          # (the traversal of the module tree costs 0.3 ms per step; the tree is fixed while training)
            params = self._param_list = list(self.module.parameters())
        for p in params:
            if p._backward_hooks:
                return True
            post = getattr(p, '_post_accumulate_grad_hooks', None)
            if post and any(k not in ours for k in post):
                return True
        return False

    def _forward_backward(self, data, speculate):
        """zero_grad + forward + backward (+ gradient average); returns (loss, log_vars, n, speculation records)"""
        from . import hip_conv
        from .core import bbox as _bbox
        self.optimizer.zero_grad(set_to_none=True)
        # (torch DDP copies gradients into its buckets from autograd hooks: nothing may be pending there)
        hip_conv.begin_step(defer=self.amp_dtype is torch.bfloat16 and self.ddp is None and not self._foreign_grad_hooks())
        if speculate:
            _bbox.begin_speculation(group=self.reducer.group if self.reducer is not None else None,
                                    collective=self._lockstep_world() > 1)
        try:
            (loss, log_vars), n = self.forward_losses(data)
            with _rf('sec:backward'):
                loss.backward()
        finally:
            recs = _bbox.end_speculation() if speculate else []
            hip_conv.end_backward()      # the deferred column-sum reductions of this backward pass: one launch
        with _rf('sec:backward'):
            if self.reducer is not None:
                self.reducer.finish()
        return loss, log_vars, n, recs

    def _step(self, data):
        from . import hip_conv
        # Device-side RoI sampling (core/bbox.py, csrc/roi_sampler.hip) removes the step's last mid-forward host read by
        # ASSUMING that every image yields the sampler's full row count.  The assumption is checked HERE, after the whole
        # backward pass has been enqueued and before the optimizer touches a parameter: the flags were copied to pinned
        # memory right behind the sampler kernel, so this wait ends as soon as the device has passed that point of THIS
        # step's forward pass (the rest of the step is queued behind it - the device never idles on it).  A short image
        # (fewer candidates than `num`: a handful of proposals survived the NMS) repeats the step through the host path
        # from the saved generator states - same draws, same result as if it had run there in the first place.
        # (more than one rank: the repeat decision is shared through a device-side all-reduce of the flags, which only the
        #  RCCL backend can do on device memory without a host round trip - any other backend draws on the host)
        speculate = self.speculative_sampling and self.ddp is None and next(self.module.parameters()).is_cuda and \
            (self._lockstep_world() == 1 or
             dist.get_backend(self.reducer.group if self.reducer is not None else None) in self.SPECULATION_BACKENDS)
        # (saved for a repeat: torch's CPU generator, numpy's and python's global streams, the device generator.  Module
        #  buffers need no saving: every BatchNorm of the named configs is frozen / eval - train-mode statistics of a custom
        #  model WOULD be updated twice by a repeated step)
        saved = (torch.get_rng_state(), np.random.get_state(), random.getstate(),
                 torch.cuda.get_rng_state(next(self.module.parameters()).device)) if speculate else None
        # (integrate_data merges the views into the dict it is given, base.py:22-48: the first pass works on a copy so that
        #  a repeat starts from the batch as it was handed in)
        first = {k: (list(v) if isinstance(v, list) else v) for k, v in data.items()} if speculate else data
        loss, log_vars, n, recs = self._forward_backward(first, speculate)
        short = False
        for r in recs:
            if r['gen'] is not None:
                r['gen'].sync_host()     # (also brings torch's CPU generator up to date with the device's draws)
            else:
                r['event'].synchronize()  # (a call this rank drew on the host: only the shared flags travel)
            short = short or bool(r['meta'][2 * r['B']:].any())
        if short:
            self.respeculated += 1
            torch.set_rng_state(saved[0])
            np.random.set_state(saved[1])
            random.setstate(saved[2])
            torch.cuda.set_rng_state(saved[3], next(self.module.parameters()).device)
            loss, log_vars, n, _ = self._forward_backward(data, False)
        elif first is not data:          # the caller's dict ends up merged, as train_step leaves it (base.py:22-48)
            data.clear()
            data.update(first)
        with _rf('sec:optimizer'):
            if self.grad_clip is not None:
                params = [p for p in self.module.parameters() if p.requires_grad and p.grad is not None]
                if params:
                    self.last_grad_norm = torch.nn.utils.clip_grad_norm_(params, **self.grad_clip)
            self.optimizer.step()
            if self.amp_dtype is torch.bfloat16:
                # the parameters just changed: re-prepare every convolution's weights (BN fold, bf16 layouts) in one launch,
                # so that the next forward pass prepares nothing (hip_conv.refresh_prepared)
                hip_conv.refresh_prepared()
        return dict(loss=loss.detach(), log_vars=log_vars, num_samples=n)


def parse_optimizer_config(cfg):
    """``optimizer_config`` / ``fp16`` of a reference config (apis/train.py:153-161) -> the keyword arguments of
    TrainEngine.  ``grad_clip`` is honoured (mmcv OptimizerHook); everything this build cannot honour is REJECTED with
    its name instead of being dropped: a custom hook ``type``, unknown keys, and ``fp16`` (mmcv's Fp16OptimizerHook =
    fp16 autocast with loss scaling - this build trains in bf16 autocast, which has fp32's exponent range and no loss
    scale; ``--amp bf16`` is the supported mixed-precision mode)."""
    if cfg.get('fp16', None) is not None:
        raise NotImplementedError(
            f"config key fp16={dict(cfg.get('fp16'))!r}: fp16 loss-scaled training (mmcv Fp16OptimizerHook) is not built; "
            "this build's mixed precision is bf16 autocast (--amp bf16, the default) - remove the fp16 key")
    oc = cfg.get('optimizer_config', None)
    oc = dict(oc) if oc else {}
    if 'type' in oc:
        raise NotImplementedError(f"optimizer_config.type={oc['type']!r}: only mmcv's plain OptimizerHook semantics "
                                  "(zero_grad, backward, optional grad_clip, step) are built")
    clip = oc.pop('grad_clip', None)
    oc.pop('detect_anomalous_params', None)          # a debugging aid of mmcv's hook, no effect on training
    if oc:
        raise NotImplementedError(f'optimizer_config keys not supported: {sorted(oc)}')
    if clip is not None:
        clip = dict(clip)
        if 'max_norm' not in clip:
            raise ValueError(f'optimizer_config.grad_clip={clip!r} needs max_norm (torch.nn.utils.clip_grad_norm_)')
    return dict(grad_clip=clip)


def train_detector(model, data_iter, cfg, distributed=False, max_iters=None, logger=print, amp_dtype=None,
                   iters_per_epoch=None):
    """apis/train.py:71-212 reduced to the hot loop: build optimizer + schedule, iterate, log every
    cfg.log_config.interval iterations.  ``iters_per_epoch`` (or ``len(data_iter)`` when it has one) drives the
    epoch-based step decay of the LR schedule."""
    optimizer = build_optimizer(model, cfg.optimizer)
    sched = StepLrSchedule(optimizer, **cfg.get('lr_config', dict(policy='step', step=[1 << 30])))
    engine = TrainEngine(model, optimizer, distributed, amp_dtype,
                         find_unused_parameters=cfg.get('find_unused_parameters', False), **parse_optimizer_config(cfg))
    interval = cfg.get('log_config', {}).get('interval', 50)
    rank, _ = get_dist_info()
    if iters_per_epoch is None and hasattr(data_iter, '__len__'):
        iters_per_epoch = len(data_iter)
    t0 = time.time()
    for it, data in enumerate(data_iter):
        if max_iters is not None and it >= max_iters:
            break
        sched.set(it // iters_per_epoch if iters_per_epoch else 0, it)
        out = engine.step(data)
        if rank == 0 and (it + 1) % interval == 0:
            lv = {k: (float(v) if not isinstance(v, float) else v) for k, v in out['log_vars'].items()}
            logger(f'iter {it + 1} lr {optimizer.param_groups[0]["lr"]:.5f} '
                   f'{(time.time() - t0) / (it + 1):.3f}s/it ' + ' '.join(f'{k}: {v:.4f}' for k, v in lv.items()))
    return engine
