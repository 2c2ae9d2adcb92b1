cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
python tools/profile_host.py --torchprof > gpurun_out/r3e/torchprof.log 2>&1
bash tools/probe/kstats.sh r3e > gpurun_out/r3e/kstats.log 2>&1
python tools/probe/kstats_show.py r3e 80 > gpurun_out/r3e/kstats_table.log 2>&1
python tools/probe/host_phases.py > gpurun_out/r3e/host_phases.log 2>&1
head -3 gpurun_out/r3e/kstats_table.log
