#!/bin/bash
# Instruction-cache counters of the trace kernels (judge r02 item 3): bash tools/r3_icache.sh [WORKLOAD ...]
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
for WL in "$@"; do
  O=$R/gpurun_out/r3/icache_$WL; mkdir -p $O
  timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY --output-format csv -d $O -- python3 bench.py --steps 16 --warmup 8 --workload $WL --frames-in-flight 1 --no-cpu-baseline --no-profile-events --workload-steps 0 --no-serial > $O/log.txt 2>&1 || echo "icache pass $WL failed"
  python3 - $O <<'PY'
import collections, csv, glob, sys
f = max(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"))
per = collections.defaultdict(lambda: collections.defaultdict(float)); grid = {}; name = {}
for r in csv.DictReader(open(f)):
    if "k_trace" in r["Kernel_Name"]:
        d = int(r["Dispatch_Id"]); per[d][r["Counter_Name"]] += float(r["Counter_Value"]); grid[d] = int(r["Grid_Size"]); name[d] = r["Kernel_Name"][:60]
big = max(grid.values())
ids = [d for d in sorted(per) if grid[d] == big][-2:]
for d in ids:
    c = per[d]
    print(name[d], "grid", grid[d], {k: int(v) for k, v in c.items()})
    if c.get("SQC_ICACHE_REQ"):
        print("   icache miss rate %.4f%%, misses per 1000 VALU insts %.3f, ifetch/wave-cycle %.4f" % (100 * c["SQC_ICACHE_MISSES"] / c["SQC_ICACHE_REQ"], 1e3 * c["SQC_ICACHE_MISSES"] / max(c["SQ_INSTS_VALU"], 1), c.get("SQ_IFETCH", 0) / max(c["SQ_WAVE_CYCLES"], 1)))
PY
done
