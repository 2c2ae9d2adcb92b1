cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
python -m pytest tests/test_hip_roi_nms.py tests/test_analytic_known_answers.py tests/test_model_parity.py -m gpu -x -q -s > gpurun_out/r3b/pytest_roi.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3b/pytest_roi.log
for i in 1 2; do
python bench.py --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err; echo "bench rc=$?"
OADG_ROI_BWD_TILES=0 python bench.py --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/r3b/bench_atomic.json 2> gpurun_out/r3b/bench_atomic.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ('bench','bench_atomic'):
    r=json.load(open(f'gpurun_out/r3b/{f}.json'))
    print(f, r['ms_per_step'], r['value'], r['roofline']['kernel'], r['roofline']['frac'])
    for x in r['roofline']['families']:
        if 'RoI' in x['family']: print('   ', x['family'], x['ms_per_step'], x['achieved'], x['frac'])
PY
done
