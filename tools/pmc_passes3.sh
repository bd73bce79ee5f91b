# instruction-issue view of the level kernels (round 1, late): which unit is busy?
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
i=0
for P in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64" "SQ_ACTIVE_INST_VALU2 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_CYCLES SQ_BUSY_CYCLES SQ_LEVEL_WAVES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); mkdir -p gpurun_out/pmc3/p$i
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/pmc3/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-events > gpurun_out/pmc3/p$i/log.txt 2>&1 || echo "pass $i ($P) failed"
done
python3 tools/pmc_summary.py gpurun_out/pmc3 > gpurun_out/pmc3/summary.txt 2>&1
echo done
