#!/bin/bash
# Round-3 profile on the GPU box (one gpurun call): usage  bash tools/r3_profile.sh TAG [WORKLOAD] [extra bench args]
#  1. the driver's command (its JSON line; for C3 with the other configs' blocks) and the 200-step command
#  2. rocprofv3 --kernel-trace --stats of the DRIVER'S command (+ the per-launch kernel trace: grid -> duration)
#  3. PMC passes (one counter group per run; --pmc alone, no tracing) of the bench's launch shape with ONE launch in
#     flight, so that a launch's counters are its own
# Outputs under gpurun_out/prof_$TAG/; tools/r3_pmc_summary.py copies the judged summaries into profiles/.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
TAG=${1:-r03}; WL=${2:-C3}; shift; shift
O=$R/gpurun_out/prof_$TAG; mkdir -p $O/stats
DRIVER="--gpus 1 --steps 20 --warmup 5 --workload $WL $@"
timeout -k 10 500 python3 bench.py $DRIVER > $O/bench_driver.json 2> $O/bench_driver.err || echo "driver bench failed"
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --workload $WL --no-cpu-baseline --workload-steps 0 "$@" > $O/bench_200.json 2> $O/bench_200.err || echo "200-step bench failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py $DRIVER --no-cpu-baseline --workload-steps 0 --no-serial > $O/stats/bench.json 2> $O/stats/err.txt || echo "stats failed"
i=0
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
         "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  i=$((i+1)); mkdir -p $O/pmc/p$i
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $O/pmc/p$i -- python3 bench.py --steps 16 --warmup 8 --workload $WL --frames-in-flight 1 --no-cpu-baseline --no-profile-events --workload-steps 0 --no-serial "$@" > $O/pmc/p$i/log.txt 2>&1 || echo "pmc pass $i ($P) failed"
done
python3 tools/r3_pmc_summary.py $TAG $WL --print-only
echo done
