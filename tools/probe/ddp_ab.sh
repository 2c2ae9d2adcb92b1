# reducer overhead at world size 1 (same box, interleaved): plain vs torch.distributed.run + FlatGradReducer over RCCL,
# and the effect of torchrun's OMP_NUM_THREADS=1 default
run(){ "$@" 2>/dev/null | grep '"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for i in 1 2; do
echo -n "plain "; run python bench.py --steps 40 --warmup 10 --no-cpu-baseline
echo -n "plain OMP=1 "; OMP_NUM_THREADS=1 run python bench.py --steps 40 --warmup 10 --no-cpu-baseline
echo -n "torchrun reducer "; OADG_BENCH_FORCE_DDP=1 run python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline
echo -n "torchrun reducer OMP=8 "; OMP_NUM_THREADS=8 OADG_BENCH_FORCE_DDP=1 run python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2952$i bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline
done
