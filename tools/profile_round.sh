#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel-trace/stats of the default bench command,
# then PMC passes (one counter group per run; one frame in flight, so that a launch's counters are its own). Outputs under gpurun_out/round_profile/.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/round_profile; mkdir -p $O/stats
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || echo bench failed
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/stats/bench.json 2> $O/stats/err.txt || echo stats failed
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d' ' -f1)
  mkdir -p $O/pmc/$N
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $O/pmc/$N -- python3 bench.py --steps 8 --warmup 8 --frames-in-flight 1 --no-cpu-baseline --no-profile-events > $O/pmc/$N/log.txt 2>&1 || echo "pass $N failed"
done
echo done; tail -c 600 $O/bench_default.json
