export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
for P in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  N=$(echo $P | cut -d' ' -f1)
  mkdir -p gpurun_out/pmc/$N
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmc/$N -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-events > gpurun_out/pmc/$N/log.txt 2>&1 || echo "pass $N failed"
done
ls gpurun_out/pmc/*/*/ | head -40
