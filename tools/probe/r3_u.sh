cd $GRAFT_REPO_ROOT
run(){ python bench.py "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'])"; }
for i in 1 2 3; do
run --no-cpu-baseline --no-families
run --no-cpu-baseline --no-families --steps 100 --warmup 20
run --no-cpu-baseline
run
done
