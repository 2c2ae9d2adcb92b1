# rocprofv3 kernel-time A/B of one environment switch: bash tools/probe/kstats_ab.sh VAR A B  (prints ms/step of all kernels + launches)
VAR=$1; A=${2:-0}; B=${3:-1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $A $B; do
  mkdir -p gpurun_out/prof_ab_$v; rm -rf /tmp/p_ab
  env $VAR=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ab -o b -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-families > gpurun_out/prof_ab_$v/bench.log 2>&1
  cp $(find /tmp/p_ab -name "*kernel_stats.csv" | head -1) gpurun_out/prof_ab_$v/kernel_stats.csv
  echo "$VAR=$v: $(python tools/probe/kstats_show.py ab_$v 0 | head -1)  bench: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_ab_$v/bench.log | tail -1)"
done
