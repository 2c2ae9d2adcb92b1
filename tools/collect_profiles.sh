#!/usr/bin/env bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + HBM traffic counters of the bench command.
# PMC counters are collected in their own passes (one --pmc per run, --kernel-trace only), as the MI355X guide
# prescribes; summaries (small CSV/JSON) are written to gpurun_out/prof_$TAG for copying into profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# CONFIG=r101_dc5 PREFIX=dc5_ collects the same summaries for the other bench configuration (files <PREFIX>kernel_stats.csv ...)
CONFIG=${CONFIG:-r50_fpn}
PREFIX=${PREFIX:-}
CMD="python bench.py --config $CONFIG --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline"
if [ "${SKIP_STATS:-0}" != "1" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_${PREFIX}stats -o b -- $CMD > $OUT/${PREFIX}bench_under_rocprof.log 2>&1
cp $(find /tmp/p_${PREFIX}stats -name "*kernel_stats.csv" | head -1) $OUT/${PREFIX}kernel_stats.csv
grep '"metric"' $OUT/${PREFIX}bench_under_rocprof.log | tail -1 > $OUT/${PREFIX}bench_under_rocprof.json
fi
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_${PREFIX}$C -o b -- python bench.py --config $CONFIG --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python - "$C" "$OUT/$PREFIX" "$PREFIX" <<'PY'
import csv, glob, json, sys, collections
c, out, pre = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(f'/tmp/p_{pre}{c}/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != c:
        continue
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if not k.startswith(('conv_', 'n16_')):    # our conv kernels keep their template arguments (one row each)
        k = k.split('<')[0]
    k = k[-70:]
    agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
rows = sorted(((k, v[0], v[1]) for k, v in agg.items()), key=lambda t: -t[1])[:60]
json.dump([dict(kernel=k, counter=c, total=t, dispatches=n, per_dispatch=t / max(n, 1)) for k, t, n in rows],
          open(f'{out}pmc_{c}.json', 'w'), indent=1)
PY
done
ls -la $OUT
