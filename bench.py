"""bench.py - training images/sec of the OA-DG hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over one batch of synthetic Cityscapes-shaped input that is already
resident in HBM: OA-Mix (view 2 of every image) + Normalize/Pad on the device -> Faster R-CNN R50-FPN forward
(both views) -> RPN/RoI losses incl. OA-Loss -> backward -> (DDP all-reduce) -> SGD step.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config r50_fpn|r101_dc5] [--workload train|oamix_stress]

Default = BASELINE configs[1] (R50-FPN, bs 4, 1024x2048), the configuration the metric is quoted on.  ``--config r101_dc5``
runs configs[3] (R101-DC5, bs 2, 736x1280) through the same machinery; ``--workload oamix_stress`` runs configs[4] (OA-Mix
with 4096 boxes per image + the OA-Loss at its 8 x 512 x 2 batch, bs 8).  The line also carries ``clocks``: sclk / socket
power / hotspot temperature sampled inside the timed region (amdsmi), so that runs on different boxes can be compared.

N > 1 is one process per GPU over RCCL: either launched by ``python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or, when ``bench.py --gpus N`` is
run directly (no WORLD_SIZE in the environment), re-launched under torch.distributed.run by this script itself
(tools/dist_train.sh:7-9 of the reference does the same for training).  The JSON line carries ``rccl_ranks`` =
``dist.get_world_size()`` of the 'nccl' (= RCCL) process group.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  "roofline"     - the dominant hand-written kernel of the step, timed live with HIP events on its stream
  "cpu_baseline" - the CPU oracle (kind "port") timed on this box's host cores on a bounded sample (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
METRIC = 'images/sec training, Faster R-CNN R50-FPN + OA-DG, 1024x2048'
# --config: the training workloads BASELINE.json lists.  r50_fpn = configs[1] (the configuration the metric is quoted on: the
# default, and what the driver runs); r101_dc5 = configs[3] at its per-GPU size (bs 2; the reference's DWD configs resize to
# (1280, 720), padded to a multiple of 32: 736 x 1280).  --workload oamix_stress = configs[4], the OA-Mix + OA-Loss isolate.
CONFIGS = {
    'r50_fpn': dict(cfg='faster_rcnn_r50_fpn_1x_cityscapes_oadg.py', batch=4, height=1024, width=2048, boxes=20, classes=8,
                    box_size=(24, 400), roi_strides=[4, 8, 16, 32], roi_channels=256, metric=METRIC,
                    label='faster_rcnn_r50_fpn_1x_cityscapes_oadg: OA-Mix + Faster R-CNN R50-FPN + OA-Loss'),
    'r101_dc5': dict(cfg='faster_rcnn_r101_dc5_1x_dwd_oadg.py', batch=2, height=736, width=1280, boxes=12, classes=7,
                     box_size=(24, 300), roi_strides=[16], roi_channels=2048,
                     metric='images/sec training, Faster R-CNN R101-DC5 + OA-DG, 736x1280 (BASELINE configs[3])',
                     label='faster_rcnn_r101_dc5_1x_dwd_oadg: OA-Mix (augmix.all) + Faster R-CNN R101-DC5 (dilated C5, no FPN, '
                           '15 anchors on one stride-16 level, 2048-channel RoIAlign) + OA-Loss'),
}
CFG = os.path.join(ROOT, 'configs', 'oadg', CONFIGS['r50_fpn']['cfg'])
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA ~2.5 PFLOP/s


def newest_profile_tag():
    """the newest round tag that has committed PMC summaries (profiles/rNN_pmc_FETCH_SIZE.json + _WRITE_SIZE.json)"""
    import glob
    import re
    tags = set()
    for f in glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_FETCH_SIZE.json')):
        m = re.match(r'(r\d+)_pmc_FETCH_SIZE\.json$', os.path.basename(f))
        if m and os.path.exists(os.path.join(ROOT, 'profiles', f'{m.group(1)}_pmc_WRITE_SIZE.json')):
            tags.add(m.group(1))
    return max(tags, key=lambda t: int(t[1:])) if tags else None


def rocprof_name(timer_name):
    """kernel name as rocprofv3 prints it from a timer name of hip_conv (which may carry a '+reduce' / ' xN (...)' note)"""
    return timer_name.split(' x')[0].replace('+reduce', '')


PMC_PREFIX = ''      # '' for the default workload; 'dc5_' ... for the others (profiles/<tag>_<prefix>pmc_*.json)


def pmc_traffic(kernel):
    """HBM-side bytes per launch of ``kernel`` from the committed rocprofv3 PMC summary of this same command
    (profiles/<tag>_<workload prefix>pmc_*.json, written by tools/collect_profiles.sh: one --pmc pass per counter; a
    workload without its own summary reports null - another workload's launches have other shapes).  Counters are in
    KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 wide coalesced reads; WRITE_SIZE is
    taken as reported (uncalibrated).  PMC counters cannot be collected from inside the timed run, so the value is the
    committed measurement of the NEWEST round only (a kernel that is not in it - renamed, new - reports null rather than
    an older round's figure), with its source named."""
    tag = newest_profile_tag()
    if tag is None:
        return {'traffic': None}
    try:
        tot = 0.0
        for c, mul in (('FETCH_SIZE', 2.0), ('WRITE_SIZE', 1.0)):
            rows = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_{PMC_PREFIX}pmc_{c}.json')))
            tot += mul * 1024.0 * next(r['per_dispatch'] for r in rows if r['kernel'] == kernel)
        return {'traffic': int(tot), 'traffic_unit': 'bytes/launch',
                'traffic_source': f'profiles/{tag}_{PMC_PREFIX}pmc_FETCH_SIZE.json x2 + {tag}_{PMC_PREFIX}pmc_WRITE_SIZE.json'}
    except (OSError, StopIteration, KeyError, ValueError):
        return {'traffic': None, 'traffic_note': f'{kernel} is not in profiles/{tag}_{PMC_PREFIX}pmc_*.json'}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='r50_fpn', choices=sorted(CONFIGS),
                    help='r50_fpn: BASELINE configs[1] (default, the metric\'s configuration); r101_dc5: configs[3]')
    ap.add_argument('--workload', default='train', choices=['train', 'oamix_stress'],
                    help='train: the whole training step; oamix_stress: BASELINE configs[4], OA-Mix with 4096 boxes per image '
                         '+ the OA-Loss at its 8 x 512 x 2 contrastive batch, bs 8')
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (default: the configuration\'s, r50_fpn 4 / r101_dc5 2)')
    ap.add_argument('--height', type=int, default=None)
    ap.add_argument('--width', type=int, default=None)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-families', action='store_true', help='skip the per-family table (3 extra steps after the timed region)')
    ap.add_argument('--pipeline-thread', type=int, default=1,
                    help='1: the data pipeline host code runs in a worker thread (DataLoader-worker analogue)')
    ap.add_argument('--conv', default='mfma', choices=['mfma', 'miopen'],
                    help='mfma: hand-written implicit-GEMM kernels where they apply; miopen: torch.conv2d only')
    a = ap.parse_args()
    wl = CONFIGS[a.config]
    for k in ('batch', 'height', 'width'):
        if getattr(a, k) is None:
            setattr(a, k, 8 if (a.workload == 'oamix_stress' and k == 'batch') else
                    (CONFIGS['r50_fpn'][k] if a.workload == 'oamix_stress' else wl[k]))
    return a


class ClockSampler:
    """sclk / socket power / hotspot temperature of the bench's GPU while the timed region runs, from the amdsmi python
    binding (one ``amdsmi_get_gpu_metrics_info`` call every ``period`` seconds on a helper thread; the call is a sysfs
    table read: no subprocess, ~0.1 ms).  The run-to-run spread of a step (28.6 - 31.3 ms between boxes) is larger than
    most single changes; the clock the chip held and the power it drew say whether two runs are comparable
    (MI355X_MICROARCH.md, 'DVFS give-back').  Absent binding / permission: ``summary()`` is {'available': False}."""

    def __init__(self, index=0, period=0.05):
        self.rows, self.period, self._stop, self._thread, self._h, self.err = [], period, False, None, None, None
        try:
            import amdsmi
            self._smi = amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self._h = hs[min(index, len(hs) - 1)]
            self._read()
        except Exception as e:        # noqa: BLE001 (any failure = no sampler, never a failed bench)
            self._h, self.err = None, repr(e)[:120]

    def _read(self):
        m = self._smi.amdsmi_get_gpu_metrics_info(self._h)

        def num(v):
            return float(v) if isinstance(v, (int, float)) and 0 <= v < 60000 else None
        clks = [c for c in (num(v) for v in (m.get('current_gfxclks') or [])) if c]
        if not clks:
            c = num(m.get('current_gfxclk'))
            clks = [c] if c else []
        return (sum(clks) / len(clks) if clks else None,
                num(m.get('current_socket_power')) or num(m.get('average_socket_power')),
                num(m.get('temperature_hotspot')))

    def start(self):
        if self._h is None:
            return self
        import threading

        def loop():
            while not self._stop:
                try:
                    self.rows.append(self._read())
                except Exception as e:  # noqa: BLE001
                    self.err = repr(e)[:120]
                    return
                time.sleep(self.period)
        self.rows, self._stop = [], False
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join()
        return self.summary()

    def summary(self):
        def stat(i, nd=0):
            v = [r[i] for r in self.rows if r[i] is not None]
            return None if not v else {'min': round(min(v), nd), 'mean': round(sum(v) / len(v), nd), 'max': round(max(v), nd)}
        if self._h is None or not self.rows:
            return {'available': False, 'note': self.err or 'no samples'}
        return {'available': True, 'source': "amdsmi gpu_metrics (mean of the XCDs' current_gfxclks), sampled every "
                                             f'{int(self.period * 1e3)} ms inside the timed region',
                'samples': len(self.rows), 'sclk_mhz': stat(0), 'socket_power_w': stat(1), 'hotspot_c': stat(2)}


def roi_algorithmic_bytes(rois, strides, C, elem, finest_scale=56, fwd=False):
    """SURVEY.md 8d RoIAlign row, backward: read 49*C grad elements + read-modify-write of the unique fp32
    input footprint (ceil(w_l)+1)(ceil(h_l)+1)*C on the RoI's level; forward: read that footprint once in the map
    dtype + write 49*C elements."""
    w = (rois[:, 3] - rois[:, 1]).clamp(min=0)
    h = (rois[:, 4] - rois[:, 2]).clamp(min=0)
    lvl = torch.floor(torch.log2(torch.sqrt(w * h) / finest_scale + 1e-6)).clamp(0, len(strides) - 1).long()
    s = torch.tensor(strides, device=rois.device, dtype=torch.float32)[lvl]
    foot = (torch.ceil(w / s) + 1) * (torch.ceil(h / s) + 1)
    if fwd:
        return float((49 * C * elem + elem * C * foot).sum().item())
    return float((49 * C * elem + 2 * 4 * C * foot).sum().item())


def roi_tile_bwd_bytes(n_rois, n_imgs, height, width, strides, C, elem):
    """what roi_align_bwd_tiles_kernel moves: every element of the dense gradient maps of the RoI levels written ONCE
    (no fp32 maps, no read-modify-write) + each RoI's 7 x 7 x C gradient slab read once"""
    maps = sum(n_imgs * (-(-height // s_)) * (-(-width // s_)) * C * elem for s_ in strides)
    return float(maps + n_rois * 49 * C * elem)


def family_of(kernel_name):
    """hip_conv.TIMERS_ONLY_VARIANT key of a kernel name recorded by hip_conv"""
    if kernel_name.startswith('conv_wgrad256'):
        return 'wgrad256'
    if kernel_name.startswith('conv_wgrad'):
        return 'wgrad128'
    if kernel_name.startswith('conv_igemm256'):
        return 2
    if kernel_name.startswith('conv_pw_stream'):
        return 4
    return 3


FAMILIES = (('conv256 forward / data gradient', ('conv_igemm256_kernel',), 'mfma'),
            ('weight gradient 256-tile (single-layer launches + grouped multi-layer launches)',
             ('conv_wgrad256_kernel', 'conv_wgrad256_multi_kernel'), 'mfma'),
            ('weight gradient 128-tile', ('conv_wgrad_kernel',), 'hbm'),
            ('pointwise streaming (1x1, C <= 512)', ('conv_pw_stream_kernel',), 'hbm'),
            ('128-tile convolution', ('conv_igemm_kernel', 'conv_igemm_s2_kernel'), 'mfma'),
            ('16-channel RPN head (rpn_cls + rpn_reg: forward, data gradient, weight gradient)', ('n16_',), 'hbm'),
            ('frozen stage-1 bottleneck blocks (one launch per block)', ('bottleneck_frozen',), 'hbm'))
F32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 matrix (xf32-free) MFMA peak


def families_table(conv_timers, op_timers, roi_sets, steps, elem, geom=None, roi_strides=(4, 8, 16, 32), roi_channels=256):
    """roofline.families: achieved rate of every hand-written kernel family the north star names, from HIP events
    recorded on the launch stream during ``steps`` extra steps after the timed region; PMC traffic (bytes per launch)
    from the committed rocprofv3 counter passes of this same command."""
    out = []
    for title, prefixes, bound in FAMILIES:
        rows = [t for t in conv_timers if t[4].startswith(prefixes)]
        if not rows:
            continue
        ms = sum(t[0].elapsed_time(t[1]) for t in rows)
        fl, by = sum(t[2] for t in rows), sum(t[3] for t in rows)
        names = sorted({t[4] for t in rows})
        e = {'family': title, 'kernels': names, 'bound': bound, 'launches_per_step': round(len(rows) / steps, 1),
             'ms_per_step': round(ms / steps, 3)}
        if bound == 'mfma':
            e.update(achieved=round(fl / ms / 1e9, 1), peak=MFMA_BF16_PEAK_TFLOPS, unit='TFLOP/s',
                     frac=round(fl / ms / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4), hbm_gbs=round(by / ms / 1e6, 1))
        else:
            e.update(achieved=round(by / ms / 1e6, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                     frac=round(by / ms / 1e6 / HBM_PEAK_GBS, 4), tflops=round(fl / ms / 1e9, 1))
        tr = {rocprof_name(n_): pmc_traffic(rocprof_name(n_)) for n_ in names}
        tr = {n_: t_ for n_, t_ in tr.items() if t_.get('traffic')}
        if tr:
            e['traffic'] = {n_: t_['traffic'] for n_, t_ in tr.items()}
            e['traffic_source'] = next(iter(tr.values()))['traffic_source']
        out.append(e)

    def ms_of(key):
        return [(t[0].elapsed_time(t[1]), t[2]) for t in op_timers.get(key, [])]
    # RoIAlign: bytes per launch from the RoIs of the step
    rois_bytes = {'roi_align_fwd': [], 'roi_align_bwd': []}
    for rs_ in roi_sets:
        if rs_ is None:
            continue
        allr = torch.cat([r_.float() for r_ in rs_])
        b_bwd = roi_algorithmic_bytes(allr, list(roi_strides), roi_channels, elem)
        rois_bytes['roi_align_bwd'].append(b_bwd)
        # forward: read the unique footprint in the map dtype, write 49 C elements
        rois_bytes['roi_align_fwd'].append(roi_algorithmic_bytes(allr, list(roi_strides), roi_channels, elem, fwd=True))
    from oadg_amd import hip_ops as _ho
    tiles = bool(_ho.BWD_TILES and elem == 2)
    bwd_kernel = 'roi_align_bwd_tiles_kernel' if tiles else 'roi_align_bwd_kernel'
    if tiles and geom is not None:
        # the default bf16 backward organises the gradient by OUTPUT tiles: its algorithmic bytes are the dense maps written
        # once + the slabs read once, NOT the SURVEY 8d scatter model (49 C read + fp32 read-modify-write of the footprint)
        survey = list(rois_bytes['roi_align_bwd'])
        rois_bytes['roi_align_bwd'] = [roi_tile_bwd_bytes(sum(r_.shape[0] for r_ in rs_), geom['n_imgs'], geom['height'],
                                                          geom['width'], list(roi_strides), roi_channels, elem)
                                       for rs_ in roi_sets if rs_ is not None]
    for key, title in (('roi_align_fwd', 'RoIAlign forward'),
                       ('roi_align_bwd', 'RoIAlign backward (%s)' % bwd_kernel)):
        rows = ms_of(key)
        if rows and rois_bytes[key]:
            ms = sum(r[0] for r in rows)
            by = sum(rois_bytes[key]) / len(rois_bytes[key]) * len(rows)
            kname = bwd_kernel if key == 'roi_align_bwd' else 'roi_align_fwd_rows_kernel'
            e = {'family': title, 'kernels': [kname], 'bound': 'hbm', 'launches_per_step': round(len(rows) / steps, 1),
                 'ms_per_step': round(ms / steps, 3), 'achieved': round(by / ms / 1e6, 1), 'peak': HBM_PEAK_GBS,
                 'unit': 'GB/s', 'frac': round(by / ms / 1e6 / HBM_PEAK_GBS, 4),
                 'algorithmic_bytes_per_launch': int(by / len(rows))}
            if key == 'roi_align_bwd' and tiles and geom is not None:
                e['bytes_model'] = 'dense bf16 gradient maps of the RoI levels written once + 49 C bf16 per RoI read once'
                e['survey_8d_scatter_model_bytes_per_launch'] = int(sum(survey) / len(survey))
            e.update({k: v for k, v in pmc_traffic(kname).items() if v})
            out.append(e)
    for key, title in (('supcon_fwd', 'OA-Loss supcon forward'), ('supcon_bwd', 'OA-Loss supcon backward')):
        rows = ms_of(key)
        if rows:
            ms, fl = sum(r[0] for r in rows), sum(r[1] for r in rows)
            out.append({'family': title, 'kernels': ['supcon_tile_kernel'], 'bound': 'mfma-f32',
                        'launches_per_step': round(len(rows) / steps, 1), 'ms_per_step': round(ms / steps, 3),
                        'achieved': round(fl / ms / 1e9, 2), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(fl / ms / 1e9 / F32_MFMA_PEAK_TFLOPS, 4)})
    rows = ms_of('oamix_bbox_chain')
    if rows:
        ms, by = sum(r[0] for r in rows), sum(r[1] for r in rows)
        e = {'family': 'OA-Mix per-box blend chains (side stream; the images of a batch in lockstep)',
             'kernels': ['bbox_blend_imgs_kernel', 'rect_copy_imgs_kernel'],
             'bound': 'hbm', 'launches_per_step': round(len(rows) / steps, 1), 'ms_per_step': round(ms / steps, 3),
             'achieved': round(by / ms / 1e6, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
             'frac': round(by / ms / 1e6 / HBM_PEAK_GBS, 5), 'algorithmic_bytes_per_step': int(by / steps)}
        out.append(e)
    return out


def cpu_baseline(cfg, wl=None, seconds_budget=30.0):
    """The CPU port of the step (our host logic + oracle/ ops + the OA-Mix oracle) on ONE full-resolution image
    (r50_fpn: 1024x2048, 20 boxes; r101_dc5: 736x1280, 12 boxes; both views): a bounded sample of the same workload, no
    extrapolation."""
    from oadg_amd import build_detector
    from oadg_amd.apis import build_optimizer
    from oadg_amd.detectors import integrate_data
    from oracle import oamix as OO
    from oracle.backend import oracle_ops
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from inputs import lowpass_image, synthetic_boxes
    prev_threads = torch.get_num_threads()
    threads = min(baseline_cores(), 32)  # many-core hosts: small CPU ops do not scale past a few dozen threads
    torch.set_num_threads(threads)
    wl = wl or CONFIGS['r50_fpn']
    H, W, nbox = wl['height'], wl['width'], wl['boxes']
    rs = np.random.RandomState(0)
    img = lowpass_image(rs, H, W)
    gts = synthetic_boxes(rs, nbox, H, W, 12, 200)
    labels = rs.randint(0, wl['classes'], nbox).astype(np.int64)
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det.train()
    det.local_log_vars = True        # a CPU side model: never joins the (RCCL) process group's reductions
    opt = build_optimizer(det, cfg.optimizer)
    mean = np.array([123.675, 116.28, 103.53], np.float32)
    stdinv = (1.0 / np.array([58.395, 57.12, 57.375], np.float64)).astype(np.float32)
    norm = lambda u8: torch.from_numpy(np.ascontiguousarray(  # noqa: E731
        ((u8[..., ::-1].astype(np.float32) - mean) * stdinv).transpose(2, 0, 1)))[None]
    np.random.seed(0)
    torch.manual_seed(0)
    t0 = time.time()
    okw = next(({k: v for k, v in t.items() if k != 'type'} for t in cfg.data.train.pipeline if t['type'] == 'OAMix'), {})
    r = OO.OAMixOracle(**okw)(dict(img=img.copy(), gt_bboxes=gts.copy()))
    t_mix = time.time() - t0
    shape = (H, W, 3)
    data = dict(img=norm(img), img2=norm(r['img2']), gt_bboxes=[torch.from_numpy(gts)],
                gt_bboxes2=[torch.from_numpy(gts.copy())], gt_labels=[torch.from_numpy(labels)],
                multilevel_boxes=[torch.from_numpy(np.asarray(r['multilevel_boxes']))],
                oamix_boxes=[torch.from_numpy(np.asarray(r['oamix_boxes']))],
                img_metas=[dict(img_shape=shape, pad_shape=shape, ori_shape=shape, scale_factor=1.0, flip=False)])
    with oracle_ops():
        out = det.train_step(data, None)
        opt.zero_grad()
        out['loss'].backward()
        opt.step()
    t_all = time.time() - t0
    torch.set_num_threads(prev_threads)
    return dict(value=round(1.0 / t_all, 5), unit='images/s', cores=threads, kind='port',
                sample=f'1 image at {H}x{W} (one of the {wl["batch"]} images of a step): OA-Mix oracle {t_mix:.1f}s + '
                       f'detector step with oracle ops {t_all - t_mix:.1f}s, torch {threads} threads, '
                       f'nproc={os.cpu_count()}.' + ('  For comparison, the GENUINE reference python (tier-C harness, '
                       'SURVEY.md section 0) measured on the build container: 122.6 s per step of 2 images at 1024x2048 '
                       'on 8 cores = 0.016 images/s' if (H, W) == (1024, 2048) else ''))


STRESS_METRIC = 'images/sec, OA-Mix (4096 boxes/image) + OA-Loss (8 x 512 x 2 RoI contrastive batch) isolate, 1024x2048 (BASELINE configs[4])'


def stress_cpu_baseline(H, W, n_boxes_sample=6):
    """the OA-Mix oracle on ONE full-resolution image with ``n_boxes_sample`` of the 4096 boxes (the restated reference warps
    the whole image once per box step: ~4.5 s per box on these cores, 4096 boxes would take hours) + the OA-Loss oracle at
    the full contrastive batch: a bounded sample, nothing extrapolated."""
    from oracle import losses as OL
    from oracle import oamix as OO
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from inputs import lowpass_image, supcon_inputs, synthetic_boxes
    prev_threads = torch.get_num_threads()
    threads = min(baseline_cores(), 32)
    torch.set_num_threads(threads)
    rs = np.random.RandomState(0)
    img = lowpass_image(rs, H, W)
    gts = synthetic_boxes(rs, n_boxes_sample, H, W, 8, 48)
    np.random.seed(0)
    t0 = time.time()
    OO.OAMixOracle(version='augmix')(dict(img=img.copy(), gt_bboxes=gts.copy()))
    t_mix = time.time() - t0
    feats, labels = supcon_inputs(0, n_fg_per_img=100, n_rand=17, n_img=8, per_img=512, dim=256)
    K, B = labels.shape[0], feats.shape[0]
    f = torch.tensor(feats, requires_grad=True)
    t0 = time.time()
    lo = OL.supcon(f, torch.tensor(labels), ori_size=K // 2, rp_size=(B - K) // 2, temper=0.06, loss_weight=0.01)
    lo.backward()
    t_sup = time.time() - t0
    torch.set_num_threads(prev_threads)
    return dict(value=round(1.0 / (t_mix + t_sup / 8.0), 5), unit='images/s', cores=threads, kind='port',
                sample=f'OA-Mix oracle on 1 image at {H}x{W} with {n_boxes_sample} boxes (NOT 4096: one full-image warp per '
                       f'box step) {t_mix:.1f}s + OA-Loss oracle forward + backward at B = {B} rows {t_sup:.1f}s (1/8 of it '
                       f'charged to the image), torch {threads} threads, nproc={os.cpu_count()}')


def oamix_stress_run(a, rank, distributed, dev, affinity=None):
    """BASELINE configs[4]: the augmentation + OA-Loss kernels in isolation.  One step = the device pipeline pass (OA-Mix
    of every image + Normalize / Pad) over ``--batch`` (8) images of 1024 x 2048 with 4096 boxes of 8 - 48 px each, then the
    OA-Loss (supcon forward + backward) at the contrastive batch 8 x 512 RoIs x 2 views + the random RoIs.  Inputs are
    resident in HBM; the same barrier / synchronise / max-over-ranks protocol as the training workload."""
    import oadg_amd  # noqa: F401
    from oadg_amd import Config, hip_ops
    from oadg_amd.apis import set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from inputs import supcon_inputs
    H, W, NB = a.height, a.width, 4096
    cfg = Config.fromfile(CFG)
    set_random_seed(1 + rank)
    ds = SyntheticCityscapes(img_shape=(H, W), num_boxes=NB, num_classes=8, box_size=(8, 48), seed=rank, device=dev)
    pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
    nb = 2
    batches = [ds.batch(range(i * a.batch, (i + 1) * a.batch)) for i in range(nb)]
    feats, labels = supcon_inputs(rank, n_fg_per_img=100, n_rand=17, n_img=8, per_img=512, dim=256)
    K, B = labels.shape[0], feats.shape[0]
    f = torch.tensor(feats, device=dev, requires_grad=True)
    lab = torch.tensor(labels, device=dev)
    torch.cuda.synchronize()
    ev = {'pass': [], 'supcon': []}
    record = [False]

    def step(i):
        es = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if record[0] else None
        if es:
            es[0].record()
        data = pipe(*batches[i % nb])
        if es:
            es[1].record()
        f.grad = None
        loss = hip_ops.supcon_loss(f, lab, K // 2, (B - K) // 2, 0.06, 10, 0.01)
        loss.backward()
        if es:
            es[2].record()
            ev['pass'].append((es[0], es[1]))
            ev['supcon'].append((es[1], es[2]))
        return {'loss': loss, 'img2': data['img2']}
    np.random.seed(1000 + rank)
    for i in range(a.warmup):
        step(i)
    pipe.oamix.stats = {}
    record[0] = True
    clocks = ClockSampler(torch.cuda.current_device()).start() if rank == 0 else None
    dt, out = timed_region(step, a, distributed, dev, torch.cuda.synchronize)
    clock_summary = clocks.stop() if clocks is not None else None
    if rank != 0:
        return
    st = pipe.oamix.stats
    views = a.steps * a.batch
    S = st.get('compose_steps', 0)
    # SURVEY.md 8d OA-Mix row, "materialise every step" model with the S actually drawn: 3 H W (2 S + C + 2) per view + the
    # per-box steps' rect bytes
    model = 3.0 * H * W * (2 * S + (3 + 2) * views) + 3.0 * st.get('bbox_px', 0)
    ms_pass = sum(e0.elapsed_time(e1) for e0, e1 in ev['pass'])
    ms_sup = sum(e0.elapsed_time(e1) for e0, e1 in ev['supcon'])
    flops = 6.0 * B * B * 256 * a.steps
    roof = {'kernel': 'OA-Mix device pass (every oamix.hip launch of the step: box profiles, fg union by rect scatter, saliency, '
                      'per-box blend chains, LUT / compose steps, tile-binned object-aware mixing, normalize)',
            'bound': 'hbm', 'achieved': round(model / (ms_pass * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(model / (ms_pass * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), 'traffic': None,
            'avg_launch_ms': round(ms_pass / a.steps, 3), 'launches': a.steps,
            'algorithmic_bytes_per_launch': int(model / a.steps),
            'bytes_model': 'SURVEY 8d: 3 H W (2 S + C + 2) per view for the S compose steps drawn + 3 w h per per-box step',
            'ms_per_view': round(ms_pass / views, 3), 'compose_steps_per_view': round(S / views, 2),
            'bbox_ops_per_view': round(st.get('bbox_ops', 0) / views, 2),
            'note': 'host-bound since round 6: one pass = ~36 ms of host work (recording 8 x 4096-box images ~17 ms, ~1,070 '
                    'chain launches ~8.5 ms, image states ~7.5 ms: tools/probe/stress_host.py) against ~27.6 ms of kernels - '
                    'see profiles/r06_stress_kernel_stats.csv for the kernel split',
            'families': [{'family': 'OA-Loss supcon forward + backward', 'kernels': ['supcon_tile_kernel'], 'bound': 'mfma-f32',
                          'launches_per_step': 2.0, 'ms_per_step': round(ms_sup / a.steps, 3),
                          'achieved': round(flops / (ms_sup * 1e-3) / 1e12, 2), 'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                          'frac': round(flops / (ms_sup * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                          'rows': f'B = {B} (8 img x 512 RoIs x 2 views + {B - K} random), D = 256'}]}
    res = {'metric': STRESS_METRIC, 'value': round(a.gpus * a.batch * a.steps / dt, 3), 'unit': 'images/s',
           'n_gpus': a.gpus, 'rccl_ranks': (dist.get_world_size() if distributed and dist.get_backend() == 'nccl' else 1),
           'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 2), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u8 (OA-Mix) / f32 (OA-Loss)', 'data': 'synthetic',
           'config': {'workload': f'OA-Mix stress: {NB} boxes/image of 8-48 px, {a.batch} img/GPU, {H}x{W}, view 2 of every image '
                                  f'+ Normalize/Pad; OA-Loss at B = {B} rows', 'global_batch': a.gpus * a.batch,
                      'parallelism': f'dp{a.gpus}', 'final_loss': round(float(out['loss'].detach()), 5), 'cpu_affinity': affinity},
           'clocks': clock_summary, 'roofline': roof,
           'cpu_baseline': None if (a.no_cpu_baseline or a.gpus != 1) else stress_cpu_baseline(H, W)}
    print(json.dumps(res))


_FULL_AFFINITY = None        # the CPUs this process could use before pin_rank_to_cores() narrowed them


def baseline_cores():
    """the CPU baseline leg runs on the host's cores, not on the GPU run's compact set: widen the affinity again (rank 0, N = 1
    only: nobody else is measuring) and return how many CPUs there are"""
    if _FULL_AFFINITY is not None and hasattr(os, 'sched_setaffinity'):
        os.sched_setaffinity(0, _FULL_AFFINITY)
    return len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)


def pin_rank_to_cores(local_rank, world):
    """oadg_amd.apis.pin_rank_to_cores (a compact set of physical cores per rank: its docstring has the measurements); keeps
    the full CPU set for the CPU-baseline leg.  OADG_BENCH_NO_AFFINITY=1: nothing is pinned."""
    if os.environ.get('OADG_BENCH_NO_AFFINITY') == '1':
        return None
    from oadg_amd.apis import pin_rank_to_cores as pin
    global _FULL_AFFINITY
    full = set(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None
    desc = pin(local_rank, world, int(os.environ.get('OADG_BENCH_CORES_PER_RANK', 8)))
    if desc is not None:
        _FULL_AFFINITY = full
    return desc


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: re-run this command line as N ranks on this node"""
    import socket
    import subprocess
    with socket.socket() as sk:                  # a free rendezvous port on the loopback interface
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def timed_region(step, a, distributed, dev, sync):
    """EXACTLY a.steps steps bracketed by barrier + device synchronisation on both sides; returns (seconds - the MAX
    over the ranks, last step's output)"""
    sync()
    if distributed:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    out = None
    for i in range(a.steps):
        out = step(a.warmup + i)
    sync()
    if distributed:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, out


def plumbing_run(a, factory, rank, world, distributed, affinity=None):
    """see OADG_BENCH_STEP_FACTORY in main(): same control flow as the GPU run, CPU tensors, gloo"""
    import importlib
    mod, fn = factory.split(':')
    wl = getattr(importlib.import_module(mod), fn)(a, rank, world, distributed)
    for i in range(a.warmup):
        wl['step'](i)
    dt, out = timed_region(wl['step'], a, distributed, torch.device('cpu'), lambda: None)
    extras = wl['finish']() if 'finish' in wl else {}
    if rank != 0:
        return
    print(json.dumps({
        'metric': METRIC, 'value': round(a.gpus * a.batch * a.steps / dt, 3), 'unit': 'images/s', 'n_gpus': a.gpus,
        'rccl_ranks': 0, 'world_size': dist.get_world_size() if distributed else 1,
        'dist_backend': dist.get_backend() if distributed else None, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': round(dt / a.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'fp32', 'data': 'synthetic',
        'config': dict({'workload': 'PLUMBING RUN on CPU ranks (OADG_BENCH_STEP_FACTORY): NOT a measurement',
                        'global_batch': a.gpus * a.batch, 'parallelism': f'dp{a.gpus}', 'cpu_affinity': affinity,
                        'final_loss': round(float(out['loss']), 4)}, **extras),
        'roofline': None, 'cpu_baseline': None}))


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(a.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # OADG_BENCH_FORCE_DDP=1 (with torch.distributed.run --nproc-per-node 1): exercise the DDP / RCCL path on one GPU
    distributed = world > 1 or os.environ.get('OADG_BENCH_FORCE_DDP') == '1'
    import oadg_amd
    from oadg_amd import Config, build_detector, hip_ops
    from oadg_amd.apis import TrainEngine, build_optimizer, init_dist, set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    # OADG_BENCH_STEP_FACTORY='module:function' (tests only, tests/test_cli.py): the launch / rendezvous / barrier / max-
    # over-ranks / JSON plumbing of THIS script run on CPU ranks over gloo, with the step supplied by the named factory -
    # there is no multi-GPU node to run the RCCL path on.  Its line is marked as a plumbing run, never a measurement.
    plumbing = os.environ.get('OADG_BENCH_STEP_FACTORY')
    if distributed:
        init_dist('pytorch', backend='gloo' if plumbing else 'nccl')
    rank = dist.get_rank() if distributed else 0
    assert a.gpus == world, f'--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    # (OADG_BENCH_PIN_WORLD: pin as one of that many ranks although this process runs alone - tools/probe/eight_rank_host_proxy.sh)
    affinity = pin_rank_to_cores(int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('OADG_BENCH_PIN_WORLD', world)))
    if plumbing:
        return plumbing_run(a, plumbing, rank, world, distributed, affinity)
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    dev = torch.device('cuda', torch.cuda.current_device())
    if a.workload == 'oamix_stress':
        return oamix_stress_run(a, rank, distributed, dev, affinity)
    wl = CONFIGS[a.config]
    global PMC_PREFIX
    PMC_PREFIX = '' if a.config == 'r50_fpn' else 'dc5_'
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'oadg', wl['cfg']))
    amp = torch.bfloat16 if a.dtype == 'bf16' else None
    from oadg_amd import hip_conv
    if a.conv == 'mfma' and amp is not None:
        hip_conv.enable()
    set_random_seed(0)                       # identical initial weights on every rank
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    det.log_vars_on_host = False             # log_vars stay on the device (read at a log interval in training)
    engine = TrainEngine(det, build_optimizer(det, cfg.optimizer), distributed=distributed, amp_dtype=amp)
    if a.conv == 'miopen':
        hip_conv.enable(False)               # (the engine enables the MFMA convolutions for bf16 training)
    set_random_seed(1 + rank)                # per-rank data / augmentation streams
    ds = SyntheticCityscapes(img_shape=(a.height, a.width), num_boxes=wl['boxes'], num_classes=wl['classes'],
                             box_size=wl['box_size'], seed=rank, device=dev)
    pipe = DevicePipeline(cfg.data.train.pipeline, dtype=amp or torch.float32)
    nb = min(a.steps + a.warmup, 6)
    batches = [ds.batch(range(i * a.batch, (i + 1) * a.batch)) for i in range(nb)]   # resident in HBM
    torch.cuda.synchronize()

    # software pipeline, as a DataLoader with prefetching would give: the pipeline of batch i+1 is enqueued on
    # a side stream right after the step of batch i.  Every timed step contains one pipeline pass (OA-Mix +
    # Normalize/Pad of 4 images) and one train step; the batch for the first timed step is produced by the last
    # warm-up step and the last timed step produces one more batch, so K pipeline passes run inside the region.
    wseed = None if a.pipeline_thread == 0 else 1000 + rank     # worker thread with its own numpy stream
    state = {'next': pipe.prefetch(*batches[0], worker_seed=wseed)}

    reuse = os.environ.get('OADG_BENCH_DIAG_REUSE_BATCH') == '1'   # diagnostic only: how much does the concurrent
    fixed = state['next'].get() if reuse else None                  # pipeline cost the step?  (INVALID as a result)

    # OADG_BENCH_STEP_TRACE=1 (diagnostic): host clock at every step's start + an event behind every step on the main
    # stream (no synchronisation): the warm-up curve, host enqueue time and device time step by step -> `step_trace`
    trace = [] if os.environ.get('OADG_BENCH_STEP_TRACE') == '1' else None

    def step(i):
        if trace is not None:
            t_host = time.perf_counter()
            try:
                return _step(i)
            finally:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                trace.append((t_host, time.perf_counter(), ev))
        return _step(i)

    def _step(i):
        if reuse:
            return engine.step(fixed)
        data = state['next'].get()
        if wseed is not None:        # the worker enqueues batch i+1 while this thread runs step i
            state['next'] = pipe.prefetch(*batches[(i + 1) % nb], worker_seed=wseed)
        out = engine.step(data)
        if wseed is None:
            state['next'] = pipe.prefetch(*batches[(i + 1) % nb])
        return out

    # torch DDP instruments its first 10 iterations (Logger.set_runtime_stats_and_log synchronises on timing events
    # in _pre_forward: ~6 ms of host stall per step, measured); a long training run pays that once, so the
    # distributed bench primes past it before the W warm-up steps instead of timing it.
    priming = max(0, 11 - a.warmup) if engine.ddp is not None else 0      # (torch DDP is opt-in: OADG_USE_TORCH_DDP=1)
    for i in range(priming):
        step(i)
    mfma = a.conv == 'mfma' and amp is not None
    for i in range(a.warmup):
        out = step(i)
    # Live HIP events (on the launch stream) inside the timed region, arranged so that `roofline.kernel` is the same on every
    # run: every SAMPLE_EVERY-th timed step carries event pairs around ALL launches of the two kernel families that can be
    # the largest single kernel of the step - the 256-tile forward / data-gradient kernel and the 256-tile weight-gradient
    # kernels (~90 pairs, ~0.4 ms on such a step, ~0.1 ms per step on average) - and the kernel with the largest total
    # over those steps is reported (round 3 chose the family from a 2-step warm-up probe: two kernels within 5 % of each
    # other swapped places from run to run).  The other families are measured on extra steps AFTER the timed region.
    sample_every = int(os.environ.get('OADG_BENCH_SAMPLE_EVERY', 4))
    sampled_steps = [0]
    live = [] if mfma else None
    hip_ops.TIMERS = None
    hip_conv.TIMERS = None
    hip_conv.TIMERS_ONLY_VARIANT = None if os.environ.get('OADG_BENCH_DIAG_CONV') == '1' else (2, 'wgrad256')
    plain_step = step

    def step(i):                                # noqa: F811 (the timed region's step: the sampler around the plain one)
        on = live is not None and (i - a.warmup) % sample_every == 0
        hip_conv.TIMERS = live if on else None
        sampled_steps[0] += int(on)
        try:
            return plain_step(i)
        finally:
            hip_conv.TIMERS = None
    clocks = ClockSampler(torch.cuda.current_device()).start() if rank == 0 else None
    dt, out = timed_region(step, a, distributed, dev, torch.cuda.synchronize)
    clock_summary = clocks.stop() if clocks is not None else None
    step = plain_step
    hip_conv.TIMERS = live
    conv_timers, hip_conv.TIMERS = hip_conv.TIMERS, None
    loss = float(out['loss'])
    assert np.isfinite(loss), 'training diverged'
    # ---- per-family table: DIAG_STEPS more steps with event pairs around every convolution / weight-gradient launch,
    #      RoIAlign forward / backward, the OA-Loss kernels and the OA-Mix per-box chains (worker thread, side stream)
    diag_conv, diag_ops, diag_steps = [], {}, 0
    if rank == 0 and a.gpus == 1 and not a.no_families:
        diag_steps = 3
        hip_ops.TIMERS = {k: [] for k in ('roi_align_fwd', 'roi_align_bwd', 'supcon_fwd', 'supcon_bwd', 'oamix_bbox_chain')}
        if mfma:
            hip_conv.TIMERS, hip_conv.TIMERS_ONLY_VARIANT = [], None
        roi_sets = []
        for i in range(diag_steps):
            step(a.warmup + a.steps + i)
            roi_sets.append(getattr(det.roi_head, '_last_rois', None))
        torch.cuda.synchronize()
        if wseed is not None:
            state['next'].get()          # the worker's last pipeline pass has been enqueued: its events are recorded
            torch.cuda.synchronize()
        diag_conv, hip_conv.TIMERS = (hip_conv.TIMERS or []), None
        diag_ops, hip_ops.TIMERS = hip_ops.TIMERS, None
    if rank != 0:
        return
    # ---- roofline of the dominant hand-written kernel, from HIP events recorded on the kernel's stream inside
    #      the timed region: forward / data-gradient (conv_igemm*) or weight-gradient (conv_wgrad*) family, whichever
    #      takes the most time per step.  Algorithmic FLOPs per launch = 2*M*K*R*S*C summed over the launches / number of
    #      launches, divided by the mean launch duration.
    elem = 2 if amp is not None else 4
    n_s = max(sampled_steps[0], 1)
    if conv_timers:
        per_kernel = {}
        for t in conv_timers:
            per_kernel.setdefault(rocprof_name(t[4]), []).append((t[0].elapsed_time(t[1]), t[2], t[3]))
        name, rows = max(per_kernel.items(), key=lambda kv: sum(r[0] for r in kv[1]))   # the dominant one
        ms = [r[0] for r in rows]
        avg_ms, per_launch = sum(ms) / len(ms), sum(r[1] for r in rows) / len(rows)
        achieved = per_launch / (avg_ms * 1e-3) / 1e12
        roof = {'kernel': name, 'bound': 'mfma', 'achieved': round(achieved, 1),
                'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / MFMA_BF16_PEAK_TFLOPS, 4),
                'traffic': None, 'avg_launch_ms': round(avg_ms, 4), 'launches': len(ms),
                'algorithmic_flops_per_launch': int(per_launch),
                'algorithmic_bytes_per_launch': int(sum(r[2] for r in rows) / len(rows)),
                'launches_per_step': round(len(ms) / n_s, 1),
                'kernel_ms_per_step': round(sum(ms) / n_s, 2),
                'timed_steps': f'{n_s} of the {a.steps} steps of the timed region (every {sample_every}th)',
                'other_conv_kernels': {k: {'launches_per_step': round(len(v) / n_s, 1),
                                           'ms_per_step': round(sum(r[0] for r in v) / n_s, 2),
                                           'achieved': round(sum(r[1] for r in v) / sum(r[0] for r in v) / 1e9, 1)}
                                       for k, v in per_kernel.items() if k != name}}
        # the family mixes shapes (3x3 at P2 ... P4 / layer3 and a few HBM-bound 1x1 launches): its heaviest shape alone
        shp_ = {}
        for t in conv_timers:
            if rocprof_name(t[4]) == name:
                e = shp_.setdefault(t[5], [0, 0.0, 0.0])
                e[0] += 1; e[1] += t[0].elapsed_time(t[1]); e[2] += t[2]
        if shp_:
            k_, e = max(shp_.items(), key=lambda kv: kv[1][1])
            roof['heaviest_shape'] = {**({'N,H,W,C,K,R,stride': list(k_[:7])} if k_[1] else
                                         {'grouped_launch': f'{k_[0]} layers\' weight gradients in one launch'}),
                                      'launches_per_step': round(e[0] / n_s, 1),
                                      'avg_launch_ms': round(e[1] / e[0], 4),
                                      'achieved': round(e[2] / e[1] / 1e9, 1),
                                      'frac': round(e[2] / e[1] / 1e9 / MFMA_BF16_PEAK_TFLOPS, 4)}
        roof.update(pmc_traffic(name))
    else:
        roof = {'kernel': None, 'bound': 'mfma', 'achieved': None, 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': None, 'traffic': None,
                'note': ('fp32 parity path: convolutions on csrc/conv_f32.hip (fp32 MFMA, written for exactness): no family timed'
                         if amp is None else 'library convolutions (--conv miopen): no own conv kernel timed')}
    if diag_steps:
        roof['families'] = families_table(diag_conv, diag_ops, roi_sets, diag_steps, elem,
                                          geom=dict(n_imgs=2 * a.batch, height=a.height, width=a.width),
                                          roi_strides=wl['roi_strides'], roi_channels=wl['roi_channels'])
        # the convolution launches of a step by (kernel, shape), the dozen that take the most time: which SHAPE a family's
        # time sits in (R101-DC5: the dilated 3x3 of layer4, the 2048 -> 2048 RPN convolution, the 75-channel head)
        shp = {}
        for t in diag_conv:
            e = shp.setdefault((t[4],) + tuple(t[5]), [0, 0.0, 0.0, 0.0])
            e[0] += 1; e[1] += t[0].elapsed_time(t[1]); e[2] += t[2]; e[3] += t[3]
        roof['top_shapes'] = [
            {'kernel': k_[0], 'N,H,W,C,K,R,stride': list(k_[1:8]), 'dilation': (k_[10] if len(k_) > 10 else 1),
             'residual': bool(k_[8]), 'mask': bool(k_[9]), 'launches_per_step': round(e[0] / diag_steps, 1),
             'ms_per_step': round(e[1] / diag_steps, 3), 'tflops': round(e[2] / e[1] / 1e9, 1), 'tbs': round(e[3] / e[1] / 1e9, 2)}
            for k_, e in sorted(shp.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('OADG_BENCH_TOP_SHAPES', '12'))]]
        if os.environ.get('OADG_BENCH_DIAG_CONV') == '1':      # per-shape table of the conv launches (stderr)
            for key, blocks in (hip_conv._GROUP_TRACE or {}).items():
                print(f'wgrad group of {len(key)} jobs, {blocks} workgroups: ' +
                      ' '.join(f'[{n}x{h}x{w} C{c} K{k} R{r} /{sp}]' for n, h, w, c, k, r, sp in key), file=sys.stderr)
            for k_, e in sorted(shp.items(), key=lambda kv: -kv[1][1]):
                print(f'{k_[0][5:]:30s} N{k_[1]} {k_[2]}x{k_[3]} C{k_[4]} K{k_[5]} R{k_[6]} s{k_[7]} res{int(k_[8])} mask{int(k_[9])}: '
                      f'{e[0] / diag_steps:5.1f}/step {e[1] / diag_steps:6.3f} ms/step {e[1] / e[0] * 1e3:7.1f} us {e[2] / e[1] / 1e9:7.1f} TF/s '
                      f'{e[3] / e[1] / 1e9:6.2f} TB/s', file=sys.stderr)
    res = {
        'metric': wl['metric'], 'value': round(a.gpus * a.batch * a.steps / dt, 3), 'unit': 'images/s',
        'n_gpus': a.gpus, 'rccl_ranks': (dist.get_world_size() if distributed and dist.get_backend() == 'nccl' else 1),
        'world_size': dist.get_world_size() if distributed else 1, 'dist_backend': dist.get_backend() if distributed else None,
        'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 2),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
        'config': {'workload': f'{wl["label"]}, {a.batch} img/GPU x 2 views, {a.height}x{a.width}, {wl["boxes"]} boxes/img, SGD step',
                   'global_batch': a.gpus * a.batch, 'parallelism': f'dp{a.gpus}', 'priming_steps': priming, 'final_loss': round(loss, 4),
                   'conv': a.conv, 'cpu_affinity': affinity},
        'clocks': clock_summary,
        'roofline': roof,
    }
    if trace:
        torch.cuda.synchronize()
        res['step_trace'] = {
            'note': 'warm-up + timed + families steps in order; host_ms = time inside step() on the host, host_gap_ms = host time '
                    'between consecutive step starts, device_ms = main-stream time between the events behind consecutive steps',
            'host_ms': [round((t[1] - t[0]) * 1e3, 2) for t in trace],
            'host_gap_ms': [round((b[0] - a_[0]) * 1e3, 2) for a_, b in zip(trace, trace[1:])],
            'device_ms': [round(a_[2].elapsed_time(b[2]), 2) for a_, b in zip(trace, trace[1:])]}
    if engine.reducer is not None:       # gradient bytes the producers wrote straight into the buckets vs packed by a copy
        n_ = max(engine.reducer.steps, 1)
        res['config']['grad_mb_in_place_per_step'] = round(engine.reducer.in_place_bytes / n_ / 1e6, 1)
        res['config']['grad_mb_packed_per_step'] = round(engine.reducer.packed_bytes / n_ / 1e6, 1)
    if a.gpus == 1 and not a.no_cpu_baseline:
        res['cpu_baseline'] = cpu_baseline(cfg, dict(wl, batch=a.batch, height=a.height, width=a.width))
    else:
        res['cpu_baseline'] = None
    print(json.dumps(res))


if __name__ == '__main__':
    main()
