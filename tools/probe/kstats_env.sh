# rocprofv3 kernel-time comparison of several environment settings inside ONE gpurun call:
#   bash tools/probe/kstats_env.sh "A=1" "A=0 B=2" ...   -> per setting: ms/step of all kernels, of the main stream, of the
#   kernels matching $KS_MATCH (default: weight-gradient family), launches per step, and the bench's own ms/step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for setting in "$@"; do
  i=$((i+1)); mkdir -p gpurun_out/prof_env_$i; rm -rf /tmp/p_env
  env $setting rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_env -o b -- python bench.py --steps ${KS_STEPS:-20} --warmup 4 --no-cpu-baseline --no-families > gpurun_out/prof_env_$i/bench.log 2>&1
  cp $(find /tmp/p_env -name "*kernel_stats.csv" | head -1) gpurun_out/prof_env_$i/kernel_stats.csv
  python - "$i" "$setting" <<'PY'
import csv, os, re, sys
i, setting = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f'gpurun_out/prof_env_{i}/kernel_stats.csv')))
n = max(int(r['Calls']) for r in rows if 'roi_align_bwd' in r['Name'])
pat = re.compile(os.environ.get('KS_MATCH', 'wgrad|prep_weights_bwd'))
side = ('bbox_blend', 'rect_copy', 'compose_kernel', 'saliency', 'luts_kernel', 'hist_kernel', 'fg_union', 'box_profiles', 'final_mix')
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / n
sd = sum(float(r['TotalDurationNs']) for r in rows if any(k in r['Name'] for k in side)) / 1e6 / n
sel = sum(float(r['TotalDurationNs']) for r in rows if pat.search(r['Name'])) / 1e6 / n
ms = re.findall(r'"ms_per_step": ([0-9.]+)', open(f'gpurun_out/prof_env_{i}/bench.log').read())
print(f'[{setting}] all {tot:.2f} main {tot - sd:.2f} match {sel:.3f} ms/step; launches {sum(int(r["Calls"]) for r in rows) / n:.0f}; bench {ms[-1] if ms else "?"}')
for r in rows:
    if pat.search(r['Name']):
        print('      %-70s %6.1f/step %8.3f ms/step %8.1f us' % (r['Name'].replace('(anonymous namespace)::', '')[:70], int(r['Calls']) / n, float(r['TotalDurationNs']) / 1e6 / n, float(r['AverageNs']) / 1e3))
PY
done
