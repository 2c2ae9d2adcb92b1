cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 900 python -m pytest tests/test_hip_oamix.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_oamix.py --config both --iters 5 > gpurun_out/r3h/bench_oamix.jsonl 2> gpurun_out/r3h/bench_oamix.err
python - <<'PY'
import json
for l in open('gpurun_out/r3h/bench_oamix.jsonl'):
    d=json.loads(l)
    if d['bench']=='oamix':
        print(d['config'], {m:(d[m]['ms_per_view'], d[m]['host_ms_per_view'], d[m]['frac_of_hbm'], d[m]['levels_per_op']) for m in ('persistent','batched','per_box')})
    else: print(d)
PY
bash tools/probe/ab_env.sh OADG_OAMIX_PERSISTENT 0 1 3
