"""per-step kernel table of a rocprofv3 kernel_stats.csv (tools/probe/kstats.sh): python tools/probe/kstats_show.py TAG [N]"""
import csv
import sys
tag, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(csv.DictReader(open(f'gpurun_out/prof_{tag}/kernel_stats.csv')))
n = max(int(r['Calls']) for r in rows if 'roi_align_bwd' in r['Name'])       # one launch per step
tot = sum(float(r['TotalDurationNs']) for r in rows)
side = sum(float(r['TotalDurationNs']) for r in rows if any(k in r['Name'] for k in (
    'bbox_blend', 'rect_copy', 'compose_kernel', 'saliency', 'luts_kernel', 'hist_kernel', 'fg_union', 'box_profiles')))
print(f'steps {n}; all kernels {tot / 1e6 / n:.2f} ms/step; OA-Mix side stream ~{side / 1e6 / n:.2f}; '
      f'launches/step {sum(int(r["Calls"]) for r in rows) / n:.0f}')
for r in rows[:top]:
    print('%-84s %6.1f/step %8.3f ms/step %8.1f us' % (r['Name'].replace('(anonymous namespace)::', '')[:84], int(r['Calls']) / n,
                                                     float(r['TotalDurationNs']) / 1e6 / n, float(r['AverageNs']) / 1e3))
