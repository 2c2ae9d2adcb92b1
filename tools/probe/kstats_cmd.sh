# rocprofv3 kernel stats of ANY bench command: bash tools/probe/kstats_cmd.sh TAG STEPS <bench args...>  -> the top kernels per step
TAG=$1; STEPS=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof_$TAG
rm -rf /tmp/p_stats_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats_$TAG -o b -- python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline "$@" > gpurun_out/prof_$TAG/bench.log 2>&1
cp $(find /tmp/p_stats_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/prof_$TAG/kernel_stats.csv
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_$TAG/bench.log | head -1
python - $TAG $STEPS <<'PY'
import csv, sys
tag, n = sys.argv[1], int(sys.argv[2]) + 2
rows = list(csv.DictReader(open(f'gpurun_out/prof_{tag}/kernel_stats.csv')))
print('all kernels %.2f ms/step' % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / n))
for r in rows[:14]:
    print('%-70s %7.1f/step %8.3f ms/step %8.1f us' % (r['Name'].replace('(anonymous namespace)::', '')[:70], int(r['Calls']) / n,
                                                     float(r['TotalDurationNs']) / 1e6 / n, float(r['AverageNs']) / 1e3))
PY
