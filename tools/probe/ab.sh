# A/B inside ONE gpurun call (boxes differ by +-1 ms): previous package copy vs working tree, interleaved
run(){ python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['achieved'])"; }
for i in 1 2 3; do
OADG_PKG_DIR=$PWD/tools/probe/ab/prev/oa-dg_amd run prev
run new
done
