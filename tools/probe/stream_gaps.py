"""busy / idle time of the busiest HIP queue (the training stream) per step from a rocprofv3 --kernel-trace csv: total
idle, the largest gaps with the kernels either side, idle by gap-size class.  Steps are delimited by sgd_multi_kernel."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(list)
for r in rows:
    per[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                               r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:70]))
main = max(per.values(), key=lambda l: sum(b - a for a, b, _ in l))
main.sort()
ends = [i for i, k in enumerate(main) if k[2].startswith('sgd_multi_kernel')]
if len(ends) < 4:
    sys.exit('fewer than 4 optimizer steps in the trace')
lo, hi = ends[1], ends[-1]                   # whole steps between the second and the last optimizer launch
steps = len(ends) - 2
seg = main[lo:hi + 1]
span = (seg[-1][0] - seg[0][0]) / 1e6
busy = sum(b - a for a, b, _ in seg[:-1]) / 1e6
gaps = [(seg[i + 1][0] - seg[i][1], seg[i][2], seg[i + 1][2]) for i in range(len(seg) - 1)]
idle = sum(g for g, _, _ in gaps if g > 0) / 1e6
print(f'{steps} steps: span {span / steps:.2f} ms/step, busy {busy / steps:.2f}, idle {idle / steps:.2f}, '
      f'launches {len(seg) / steps:.0f} per step (queues in the trace: {len(per)})')
for lo_, hi_ in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 100), (100, 1e9)):
    sel = [g for g, _, _ in gaps if lo_ * 1e3 <= g < hi_ * 1e3]
    print(f'  gaps {lo_}-{hi_} us: {len(sel) / steps:7.1f} per step, {sum(sel) / 1e6 / steps:6.3f} ms per step')
agg = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    if g >= 10e3:
        agg[(a[:44], b[:44])][0] += 1
        agg[(a[:44], b[:44])][1] += g
print('gaps >= 10 us by (kernel before, kernel after), per step:')
for (a, b), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f'  {n / steps:5.1f} x {t / 1e6 / steps:6.3f} ms   after {a:44s} before {b}')
# the training queue's small launches (average under 25 us): what the "tail" of the step consists of
tab = collections.defaultdict(lambda: [0, 0])
for a, b, k in seg:
    tab[k][0] += 1
    tab[k][1] += b - a
small = {k: v for k, v in tab.items() if v[1] / v[0] < 25e3}
print(f'launches under 25 us on this queue: {sum(v[0] for v in small.values()) / steps:.0f} per step, '
      f'{sum(v[1] for v in small.values()) / 1e6 / steps:.3f} ms per step')
for k, (n, t) in sorted(small.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'  {n / steps:6.1f} x {t / n / 1e3:6.1f} us = {t / 1e6 / steps:6.3f} ms   {k}')
