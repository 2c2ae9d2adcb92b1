cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
python tools/profile_host.py --torchprof > gpurun_out/r3k/torchprof.log 2>&1
bash tools/probe/kstats.sh r3k > gpurun_out/r3k/kstats.log 2>&1
python tools/probe/kstats_show.py r3k 80 > gpurun_out/r3k/kstats_table.log 2>&1
python tools/probe/host_phases.py > gpurun_out/r3k/host_phases.log 2>&1
head -3 gpurun_out/r3k/kstats_table.log
