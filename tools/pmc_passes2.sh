export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
i=0
for P in "TA_TA_BUSY_sum TA_BUSY_avr" "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); mkdir -p gpurun_out/pmc2/p$i
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmc2/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-events > gpurun_out/pmc2/p$i/log.txt 2>&1 || echo "pass $i ($P) failed"
done
echo done
