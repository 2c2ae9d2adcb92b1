#!/usr/bin/env bash
# LDS bank-conflict / MFMA-busy counters of the conv kernels on the FPN 3x3 P2 shape (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_conv
mkdir -p $OUT
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$tag -o b -- python tools/probe/one_conv.py > /dev/null 2>&1
  python - "$tag" "$OUT" <<'PY'
import csv, glob, sys, collections
tag, out = sys.argv[1], sys.argv[2]
f = glob.glob(f'/tmp/p_{tag}/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][-40:]
    if 'conv_' not in k: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
with open(f'{out}/{tag}.txt', 'w') as fo:
    for k, d in agg.items():
        for c, v in d.items():
            fo.write(f'{k} {c} {v / n[(k, c)]:.4g} per dispatch ({n[(k, c)]})\n')
print(open(f'{out}/{tag}.txt').read())
PY
done
