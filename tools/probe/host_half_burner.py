"""A stand-in for the HOST half of one training rank: four Python threads under one interpreter lock with
the duty cycles measured on a real rank (main thread ~17 ms of launch work per 27 ms step, autograd
thread ~6, pipeline worker ~10, planner ~3), pinned like a rank (oadg_amd.apis.pin_rank_to_cores(rank, world)).  Seven of
these beside one real rank load the host the way an 8-GPU job would, on a box that has one GPU.
usage: python tools/probe/host_half_burner.py RANK WORLD SECONDS"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oadg_amd.apis import pin_rank_to_cores  # noqa: E402

rank, world, seconds = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
print('burner', rank, pin_rank_to_cores(rank, world), flush=True)
stop = time.time() + seconds
STEP = 0.027


def worker(busy):
    a = np.random.rand(64, 64).astype(np.float32)
    d = {}
    while time.time() < stop:
        t0 = time.time()
        k = 0
        while time.time() - t0 < busy:          # interpreter-bound work with small array ops in between, like launch glue
            d[k & 255] = (k, k * 2)
            if k % 50 == 0:
                a = a @ a.T * 1e-3 + 1.0
            k += 1
        rest = STEP - (time.time() - t0)
        if rest > 0:
            time.sleep(rest)


ts = [threading.Thread(target=worker, args=(b,)) for b in (0.017, 0.006, 0.010, 0.003)]
for t in ts:
    t.start()
for t in ts:
    t.join()
