# round-3 baseline: gpu tests (parity deviations printed), bench, per-shape conv table, torch-profiler host view
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
python -m pytest tests -m gpu -x -q -s -k "model_parity or oamix" > gpurun_out/r3a/pytest_parity.log 2>&1; echo "pytest parity rc=$?"
python bench.py --steps 20 --warmup 6 --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"
OADG_BENCH_DIAG_CONV=1 python bench.py --steps 10 --warmup 6 --no-cpu-baseline > gpurun_out/r3a/bench_diag.json 2> gpurun_out/r3a/bench_diag.err; echo "diag rc=$?"
python tools/profile_host.py --torchprof > gpurun_out/r3a/torchprof.log 2>&1; echo "torchprof rc=$?"
python tools/profile_host.py --small-ops > gpurun_out/r3a/smallops.log 2>&1
bash tools/probe/kstats.sh r3a > gpurun_out/r3a/kstats.log 2>&1
python tools/probe/kstats_show.py r3a 70 > gpurun_out/r3a/kstats_table.log 2>&1
tail -3 gpurun_out/r3a/pytest_parity.log; cut -c1-400 gpurun_out/r3a/bench.json
