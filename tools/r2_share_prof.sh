export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/prof_share; mkdir -p $O
export EMULATE=8 FPL=8 BATCH=8 TILE=3,3 FLIGHTS=3 SKIP=copy
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/exchange_host_cost.py > $O/log.txt 2>&1
grep "host enqueue" $O/log.txt
python3 - <<'PY'
import csv, glob, os
f = max(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_share/*/*_kernel_stats.csv"), key=os.path.getmtime)
print(open(f).read()[:1600])
f = max(glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_share/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = rows[-60:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    n = r["Kernel_Name"].replace("void rt::(anonymous namespace)::", "").replace("rt::(anonymous namespace)::", "")[:44]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} {(int(r["End_Timestamp"])-t0)/1e3:9.1f} s{r["Stream_Id"]:>3} {n}')
PY
