# rocprofv3 kernel stats of a short bench run -> gpurun_out/prof_$1/kernel_stats.csv (run through gpurun)
TAG=${1:-tmp}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof_$TAG
rm -rf /tmp/p_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o b -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_$TAG/bench.log 2>&1
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) gpurun_out/prof_$TAG/kernel_stats.csv
grep '"metric"' gpurun_out/prof_$TAG/bench.log | tail -1 | cut -c1-160
