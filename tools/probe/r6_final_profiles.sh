#!/usr/bin/env bash
# round-6 evidence set, one GPU call: kernel stats + HBM PMC passes of the bench command (R50-FPN, R101-DC5), kernel stats of the
# stress workload, untraced bench lines of the three workloads -> gpurun_out/prof_r06/
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
CONFIG=r101_dc5 PREFIX=dc5_ bash tools/collect_profiles.sh r06 >> gpurun_out/r06_collect.log 2>&1
bash tools/probe/kstats_cmd.sh r06_stress 6 --workload oamix_stress > gpurun_out/prof_r06/stress_kernel_stats.txt 2>&1
cp gpurun_out/prof_r06_stress/kernel_stats.csv gpurun_out/prof_r06/stress_kernel_stats.csv
python bench.py > gpurun_out/prof_r06/bench_default.json 2> gpurun_out/prof_r06/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r06/bench_default_steps20_warmup5.json 2>/dev/null
python bench.py --config r101_dc5 > gpurun_out/prof_r06/bench_r101_dc5.json 2>/dev/null
python bench.py --workload oamix_stress > gpurun_out/prof_r06/bench_oamix_stress.json 2>/dev/null
OADG_BENCH_DIAG_REUSE_BATCH=1 python bench.py --no-cpu-baseline --no-families 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 > gpurun_out/prof_r06/reuse_batch_ab.txt
python bench.py --no-cpu-baseline --no-families 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 >> gpurun_out/prof_r06/reuse_batch_ab.txt
ls -la gpurun_out/prof_r06
grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_r06/bench_*.json | grep -v ": [0-9]\.[0-9]"
