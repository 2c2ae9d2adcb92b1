"""CPU: the OA-Mix oracle (oracle/oamix.py) against the fixture produced by the GENUINE reference OAMix
(tests/golden/make_golden_oamix.py): augmented view, box lists and the state of the global numpy stream must be
identical, for both op lists ('augmix' and the DWD config's 'augmix.all')."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
from inputs import lowpass_image, synthetic_boxes  # noqa: E402


def _case_inputs(seed, H, W, n_gt):
    rs = np.random.RandomState(1000 + seed)
    return lowpass_image(rs, H, W), synthetic_boxes(rs, n_gt, H, W, 10, min(H, W) // 2)


@pytest.mark.parametrize('idx', range(8))
def test_oamix_oracle_reproduces_reference_fixture(golden_dir, idx):
    from oracle.oamix import OAMixOracle
    g = np.load(os.path.join(golden_dir, 'oamix_reference.npz'))
    seed, ver, H, W, n_gt = [int(v) for v in g['cases'][idx]]
    img, gts = _case_inputs(seed, H, W, n_gt)
    t = OAMixOracle(version='augmix.all' if ver else 'augmix')
    np.random.seed(seed)
    r = t(dict(img=img.copy(), gt_bboxes=gts.copy()))
    tag = f's{seed}'
    assert np.array_equal(np.asarray(r['multilevel_boxes'], dtype=np.int64), g[tag + '_multilevel_boxes'])
    assert np.array_equal(np.asarray(r['oamix_boxes'], dtype=np.int64), g[tag + '_oamix_boxes'])
    assert np.array_equal(np.asarray(r['gt_bboxes2'], dtype=np.float32), g[tag + '_gt_bboxes2'])
    assert np.array_equal(np.asarray(r['img']), g[tag + '_img'])
    assert np.array_equal(np.asarray(r['img2']), g[tag + '_img2'])
    assert np.random.uniform() == float(g[tag + '_rng_after'][0])


def test_bbox_step_dependency_levels_cover_every_read_and_write():
    """pipelines/oa_mix.dependency_levels: steps of one level must be independent.  Brute force on small rects: the
    true read set of a step = its own pixels + the four bilinear taps of every warped pixel (oracle warp_coords, the
    fixed-point coordinates the kernel uses); two steps i < j may share a level only if neither writes what the other
    reads or writes."""
    import sys
    sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
    import oadg_amd  # noqa: F401
    from oadg_amd.pipelines.oa_mix import dependency_levels, invert_affine, rotation_matrix
    from oracle import cvleaves as cv
    rs = np.random.RandomState(3)
    H, W, n = 96, 160, 40
    rects, minvs = [], []
    for k in range(n):
        w, h = rs.randint(4, 30), rs.randint(4, 30)
        x0, y0 = rs.randint(0, W - w), rs.randint(0, H - h)
        c = (x0 + w / 2.0, y0 + h / 2.0)
        kind = k % 3
        if kind == 0:
            M = rotation_matrix(c, rs.randint(-30, 31))
        elif kind == 1:
            lv = rs.uniform(-0.3, 0.3)
            M = np.float32([[1, -lv, lv * c[1]], [0, 1, 0]])
        else:
            M = np.float32([[1, 0, rs.randint(-w // 3 - 1, w // 3 + 2)], [0, 1, 0]])
        rects.append((x0, y0, w, h))
        minvs.append(invert_affine(M))
    level = dependency_levels(rects, minvs, H, W)
    writes, reads = [], []
    for (x0, y0, w, h), mi in zip(rects, minvs):
        wr = np.zeros((H, W), bool)
        wr[y0:y0 + h, x0:x0 + w] = True
        X, Y = cv.warp_coords(np.asarray(mi).reshape(2, 3), W, H, x0, y0, w, h)
        sx, sy = X >> 5, Y >> 5
        rd = wr.copy()
        for dy in (0, 1):
            for dx in (0, 1):
                yy, xx = sy + dy, sx + dx
                ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
                rd[yy[ok], xx[ok]] = True
        writes.append(wr)
        reads.append(rd)
    assert level.max() >= 1, 'the case must contain dependent steps'
    assert len(np.unique(level)) < n, 'and independent ones'
    for j in range(n):
        for i in range(j):
            if (writes[i] & reads[j]).any() or (reads[i] & writes[j]).any():
                assert level[j] > level[i], (i, j)


def test_vectorised_box_matrices_replay_the_per_box_loop_and_its_rng_stream():
    """pipelines/oa_mix.OAMix._box_matrices (all boxes of a bboxes_only_* op at once) against the per-box loop that
    mirrors bbox_augmentation.py:31-88 + augmix.py:83-188: same inverted matrices bit for bit, same boxes skipped, and
    the global numpy stream left in the same state (two doubles per drawing box)."""
    import oadg_amd  # noqa: F401
    from oadg_amd.pipelines import oa_mix as om

    class St:
        pass
    rs = np.random.RandomState(5)
    H, W = 512, 1024
    for trial in range(8):
        n = rs.randint(0, 40)
        bw, bh = rs.uniform(0.3, 300, n), rs.uniform(0.3, 300, n)
        x1, y1 = rs.uniform(0, W - bw), rs.uniform(0, H - bh)
        st = St()
        st.gt = np.stack([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32).reshape(-1, 4)
        st.H, st.W = H, W
        st.support = [None if rs.rand() < 0.1 else (int(a), int(b), max(1, int(c)), max(1, int(d)))
                      for a, b, c, d in zip(x1, y1, bw, bh)]
        mix = om.OAMix()
        for kind in ['rotate', 'shear_x', 'shear_y', 'translate_x', 'translate_y']:
            np.random.seed(trial)
            rows, rects, minv = mix._box_matrices(st, kind)
            after = np.random.random()
            np.random.seed(trial)
            ref = []
            for i, box in enumerate(st.gt):
                a, b, c, d = int(box[0]), int(box[1]), int(box[2]), int(box[3])
                if (c - a) < 1 or (d - b) < 1:
                    continue
                M = om.geo_matrix(kind, 10, (W, H), ((a + c) / 2., (b + d) / 2.), (c - a + 1, d - b + 1))
                sup = st.support[i]
                if sup is None:
                    continue
                ref.append((i, sup, om.invert_affine(M)))
            assert after == np.random.random(), 'the global numpy stream was consumed differently'
            assert list(rows) == [r_[0] for r_ in ref]
            assert np.array_equal(np.asarray(rects).reshape(-1, 4), np.array([r_[1] for r_ in ref], np.int32).reshape(-1, 4))
            assert np.array_equal(np.asarray(minv).reshape(-1, 6), np.array([r_[2] for r_ in ref]).reshape(-1, 6)), kind


def test_host_bbox_levels_function_equals_the_quadratic_definition():
    """csrc/oamix_host.hip oadg_oamix_bbox_levels (cell grid, exact pair tests) == dependency_levels (all pairs)"""
    import ctypes
    import oadg_amd  # noqa: F401
    from oadg_amd import _lib
    from oadg_amd.pipelines import oa_mix as om
    L = _lib.lib()
    rs = np.random.RandomState(9)
    H, W = 1024, 2048
    for trial in range(6):
        n = [1, 2, 20, 200, 600, 1500][trial]
        big = 200 if n <= 200 else 48
        w, h = rs.randint(4, big, n), rs.randint(4, big, n)
        x0, y0 = rs.randint(0, W - big, n), rs.randint(0, H - big, n)
        rects = np.stack([x0, y0, w, h], 1).astype(np.int32)
        minvs = np.ascontiguousarray([om.invert_affine(om.rotation_matrix((x0[i] + w[i] / 2, y0[i] + h[i] / 2),
                                                                          rs.randint(-30, 31))) for i in range(n)])
        lv = np.full(n, -7, np.int32)
        rc = L.oadg_oamix_bbox_levels(rects.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                      minvs.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n, H, W,
                                      lv.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
        assert rc == 0
        ref = om.dependency_levels(rects, minvs, H, W)
        if n <= 256:
            assert np.array_equal(lv, ref), n
        else:       # conservative cell-map bound: never earlier than the exact level, every dependency ordered
            assert (lv >= ref).all() and lv.max() <= 2 * ref.max() + 2, (n, int(lv.max()), int(ref.max()))
            r = rects.astype(np.int64)
            for j in range(0, n, 7):
                wj = (r[j, 0], r[j, 1], r[j, 0] + r[j, 2] - 1, r[j, 1] + r[j, 3] - 1)
                hit = (r[:j, 0] <= wj[2]) & (r[:j, 0] + r[:j, 2] - 1 >= wj[0]) & (r[:j, 1] <= wj[3]) & \
                    (r[:j, 1] + r[:j, 3] - 1 >= wj[1])
                assert (lv[:j][hit] < lv[j]).all()


def test_oamix_command_scheduler_keeps_every_conflicting_pair_in_program_order():
    """OAMix.execute (round 6): commands carry the buffers they read / write and go out as soon as their conflicting
    predecessors have - random command lists over a handful of buffers: every read-after-write, write-after-read and
    write-after-write pair is issued in program order (so one stream executes them in order), every command exactly once,
    and a planner entry that resolves to nothing is just done.  (Only per-box chains wait for their batch and get overtaken:
    the GPU tests compare such passes byte for byte with the sequential pass.)"""
    from oadg_amd.pipelines.oa_mix import OAMix
    om = OAMix(version='augmix')
    rs = np.random.RandomState(0)
    for trial in range(20):
        recs, logs, meta = [], [], []
        for g in range(3):
            log, rec, m = [], [], []
            for i in range(40):
                lane = int(rs.randint(0, 3))
                reads = tuple(int(v) for v in rs.choice(4, size=rs.randint(0, 3), replace=False) + 10 * lane)
                writes = tuple(int(v) for v in rs.choice(4, size=rs.randint(0, 2), replace=False) + 10 * lane)
                if i % 13 == 12:                         # something every lane meets in (the accumulator)
                    writes = writes + (99,)
                kind = 'plan' if i % 17 == 5 else 'call'
                payload = (lambda: None) if kind == 'plan' else (lambda log=log, i=i: log.append(i))
                rec.append((kind, payload, reads, writes))
                m.append((kind, set(reads), set(writes)))
            recs.append(rec); logs.append(log); meta.append(m)
        om.execute(recs)
        for log, m in zip(logs, meta):
            calls = [i for i, (k, _, _) in enumerate(m) if k == 'call']
            assert sorted(log) == calls
            pos = {i: p for p, i in enumerate(log)}
            for a in calls:
                for b in calls:
                    if a < b:
                        ra, wa = m[a][1], m[a][2]
                        rb, wb = m[b][1], m[b][2]
                        if (wa & rb) or (ra & wb) or (wa & wb):
                            assert pos[a] < pos[b], (trial, a, b)
