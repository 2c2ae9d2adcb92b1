# FETCH_SIZE / WRITE_SIZE per dispatch of the conv kernels for a bench run (own --pmc passes): bash tools/probe/pmc_fetch.sh TAG
TAG=${1:-tmp}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof_$TAG
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -o b -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python - "$C" "gpurun_out/prof_$TAG" <<'PY'
import csv, glob, json, sys, collections
c, out = sys.argv[1], sys.argv[2]
f = glob.glob(f'/tmp/p_{c}/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != c:
        continue
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if not k.startswith('conv_'):
        k = k.split('<')[0]
    k = k[-70:]
    agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
rows = sorted(((k, v[0], v[1]) for k, v in agg.items()), key=lambda t: -t[1])[:25]
json.dump([dict(kernel=k, counter=c, total=t, dispatches=n, per_dispatch=t / max(n, 1)) for k, t, n in rows],
          open(f'{out}/pmc_{c}.json', 'w'), indent=1)
for k, t, n in rows[:6]:
    print(c, k, n, round(t / max(n, 1)))
PY
done
