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
    det.init_weights()
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    det.log_vars_on_host = False
    eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)
    ds = SyntheticCityscapes(device=dev)
    pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
    batches = [ds.batch(range(i * 4, i * 4 + 4)) for i in range(3)]
    nxt = pipe.prefetch(*batches[0])
    for i in range(4):
        data = nxt.get()
        eng.step(data)
        nxt = pipe.prefetch(*batches[(i + 1) % 3])
    torch.cuda.synchronize()
    if '--torchprof' in sys.argv:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
            for i in range(2):
                data = nxt.get()
                eng.step(data)
                nxt = pipe.prefetch(*batches[(i + 1) % 3])
            torch.cuda.synchronize()
        print(p.key_averages().table(sort_by='self_cuda_time_total', row_limit=40, max_name_column_width=70))
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
