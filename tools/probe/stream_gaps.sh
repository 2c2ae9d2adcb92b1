# main-stream busy / idle per step from a rocprofv3 kernel trace (no torch profiler in the process: its host overhead
# inflates the host-bound stretches).  bash tools/probe/stream_gaps.sh [TAG]  -> gpurun_out/gaps_TAG.txt
TAG=${1:-tmp}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/p_gaps
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_gaps -o b -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-families > /tmp/gaps_bench.log 2>&1
python tools/probe/stream_gaps.py $(find /tmp/p_gaps -name "*kernel_trace.csv" | head -1) | tee gpurun_out/gaps_$TAG.txt
grep -o '"ms_per_step": [0-9.]*' /tmp/gaps_bench.log | tail -1
