#!/usr/bin/env bash
# SQ / MFMA-utilisation counters of the BENCH command per kernel (own --pmc passes, --kernel-trace only, as the
# MI355X guide prescribes): bash tools/probe/pmc_sq.sh TAG [bench args...]  ->  gpurun_out/prof_TAG/pmc_sq_<pass>.json
# MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): the gfx94x MfmaUtil formula, with
# GRBM_GUI_ACTIVE as rocprofv3 reports it on gfx950 = the SUM over the 8 XCDs (3.33 M per 200 us launch = 8 x 2.08 GHz);
# SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 = FLOPs (checked: 467.6 M x 512 = 239.4 GFLOP = the launch's 2 M K R S C);
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves.
TAG=${1:-tmp}; shift
ARGS="${@:---steps 4 --warmup 2 --no-cpu-baseline}"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof_$TAG
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAVES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  i=$((i+1)); rm -rf /tmp/p_sq$i
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_sq$i -o b -- python bench.py $ARGS > /tmp/p_sq$i.log 2>&1
  python - "$i" "gpurun_out/prof_$TAG" "$C" <<'PY'
import csv, glob, json, sys, collections
i, out, names = sys.argv[1], sys.argv[2], sys.argv[3].split()
fs = glob.glob(f'/tmp/p_sq{i}/**/*counter_collection.csv', recursive=True)
if not fs:
    print('pass', i, 'produced no counter file'); print(open(f'/tmp/p_sq{i}.log').read()[-1500:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
dur = collections.defaultdict(float)
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    if not k.startswith(('conv_', 'n16_')):
        k = k.split('<')[0]
    k = k[-70:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    d = r.get('Dispatch_Id')
    if d not in n[k]:
        n[k].add(d)
        if r.get('End_Timestamp') and r.get('Start_Timestamp'):
            dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
rows = []
for k, d in agg.items():
    nd = max(len(n[k]), 1)
    e = dict(kernel=k, dispatches=nd, avg_us=round(dur[k] / nd, 2))
    for c in names:
        if c in d:
            e[c] = d[c] / nd
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in e and e.get('GRBM_GUI_ACTIVE'):
        e['mfma_busy_pct'] = round(100 * e['SQ_VALU_MFMA_BUSY_CYCLES'] / (e['GRBM_GUI_ACTIVE'] / 8 * 1024), 2)
        if e['avg_us']:
            e['eff_clock_ghz'] = round(e['GRBM_GUI_ACTIVE'] / 8 / e['avg_us'] * 1e-3, 3)
            e['mfma_tflops'] = round(e.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0) * 512 / e['avg_us'] * 1e-6, 1)
    if 'SQ_WAVE_CYCLES' in e and e['SQ_WAVE_CYCLES']:
        for c in ('SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY'):
            if c in e:
                e[c + '_pct_of_wave_cycles'] = round(100 * e[c] / e['SQ_WAVE_CYCLES'], 2)
    if e.get('SQ_LDS_IDX_ACTIVE'):
        e['lds_conflict_pct'] = round(100 * e.get('SQ_LDS_BANK_CONFLICT', 0) / e['SQ_LDS_IDX_ACTIVE'], 2)
    rows.append(e)
key = names[0] if names[0] != 'SQ_WAIT_INST_ANY' else 'SQ_WAVE_CYCLES'
rows.sort(key=lambda e: -e.get(key, 0) * e['dispatches'])
json.dump(rows[:40], open(f'{out}/pmc_sq_pass{i}.json', 'w'), indent=1)
for e in rows[:8]:
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items()})
PY
done
