# N > 1 code path of bench.py on a one-GPU box: every rank on cuda:0, gloo carries the exchange (never a reported number)
mkdir -p gpurun_out/r2
for n in 2 4; do
BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 20 --warmup 5 > gpurun_out/r2/reh_$n.json 2> gpurun_out/r2/reh_$n.err
echo "ranks $n rc $?"; tail -c 700 gpurun_out/r2/reh_$n.json | cut -c 1-700; tail -3 gpurun_out/r2/reh_$n.err | cut -c 1-300
done
# the one-rank RCCL branch (all-gather of a one-rank group)
timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --exchange-always > gpurun_out/r2/reh_nccl1.json 2> gpurun_out/r2/reh_nccl1.err; echo "nccl one-rank rc $?"; cut -c 1-300 gpurun_out/r2/reh_nccl1.json; tail -2 gpurun_out/r2/reh_nccl1.err | cut -c 1-300
