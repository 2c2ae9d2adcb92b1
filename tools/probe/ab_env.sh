# interleaved A/B of one environment switch inside ONE gpurun call: bash tools/probe/ab_env.sh VAR [A] [B] [reps]
VAR=$1; A=${2:-0}; B=${3:-1}; REPS=${4:-3}
cd $GRAFT_REPO_ROOT
run(){ env $VAR=$1 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-families 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$VAR=$1', d['ms_per_step'], d['value'])"; }
for i in $(seq $REPS); do run $A; run $B; done
