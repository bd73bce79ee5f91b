export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
mkdir -p gpurun_out/pmc4/p1
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc4/p1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-events > gpurun_out/pmc4/p1/log.txt 2>&1 || echo failed
python3 tools/pmc_summary.py gpurun_out/pmc4 > gpurun_out/pmc4/summary.txt 2>&1
cat gpurun_out/pmc4/summary.txt
