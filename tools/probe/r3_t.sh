cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3t
python bench.py --no-cpu-baseline --no-families > gpurun_out/r3t/bench_plain.json 2>/dev/null
python bench.py > gpurun_out/r3t/bench_default.json 2> gpurun_out/r3t/bench_default.err; echo "bench rc=$?"
OADG_BENCH_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-families > gpurun_out/r3t/bench_force_ddp.json 2> gpurun_out/r3t/bench_force_ddp.err; echo "ddp rc=$?"
python tools/bench_oamix.py --config both --iters 5 > gpurun_out/r3t/bench_oamix.jsonl 2> gpurun_out/r3t/bench_oamix.err
python tools/profile_host.py --torchprof > gpurun_out/r3t/torchprof.log 2>&1
bash tools/collect_profiles.sh r3t > gpurun_out/r3t/collect.log 2>&1
python - <<'PY'
import json
for f in ('bench_default','bench_force_ddp','bench_plain'):
    try:
        d=json.loads(open(f'gpurun_out/r3t/{f}.json').read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'], d.get('rccl_ranks'), d.get('dist_backend'))
    except Exception as e: print(f, 'ERR', e)
PY
