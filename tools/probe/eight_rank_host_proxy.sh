#!/usr/bin/env bash
# One real rank pinned to slice 0 of 8 of the host, alone and beside seven stand-ins for the other ranks' host halves
# (tools/probe/host_half_burner.py on slices 1-7): does a rank's step time survive the host load of an 8-GPU job?
# bash tools/probe/eight_rank_host_proxy.sh TAG  -> gpurun_out/eight_rank_TAG.txt
TAG=${1:-tmp}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/eight_rank_$TAG.txt
echo "host: $(nproc) CPUs, load average $(cut -d' ' -f1-3 /proc/loadavg)" > $OUT
one() { OADG_BENCH_PIN_WORLD=8 python bench.py --no-cpu-baseline --no-families 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms per step;', d['config']['cpu_affinity'])"; }
one "alone, slice 0 of 8:" >> $OUT
one "alone, slice 0 of 8:" >> $OUT
for r in 1 2 3 4 5 6 7; do python tools/probe/host_half_burner.py $r 8 75 >> $OUT.burners 2>&1 & done
sleep 3
one "beside 7 host-half stand-ins:" >> $OUT
one "beside 7 host-half stand-ins:" >> $OUT
echo "load average with the stand-ins: $(cut -d' ' -f1-3 /proc/loadavg)" >> $OUT
wait
head -3 $OUT.burners >> $OUT; rm -f $OUT.burners
cat $OUT
