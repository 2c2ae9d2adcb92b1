run(){ python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['roofline']['achieved'])"; }
OADG_STEP_PRIO=none run none
run high
OADG_STEP_PRIO=none run none
run high
OADG_STEP_PRIO=0 run own_stream_prio0
OADG_PIPE_PRIO=1 run high_pipe_low
run high
