# re-run the two bench commands of a profiled workload after profiles/pmc_<workload>.json was refreshed
cd $GRAFT_REPO_ROOT
for spec in "r02 C3" "r02_C2 C2" "r02_C4 C4" "r02_C5 C5"; do
  set -- $spec; O=gpurun_out/prof_$1; mkdir -p $O
  timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 --workload $2 > $O/bench_driver.json 2> $O/bench_driver.err || echo "driver bench failed $2"
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --workload $2 --no-cpu-baseline > $O/bench_200.json 2> $O/bench_200.err || echo "200-step bench failed $2"
  python3 - $O $2 <<'PY'
import json, sys
o, wl = sys.argv[1:3]
for n in ("bench_driver", "bench_200"):
    d = json.loads(open(f"{o}/{n}.json").read().strip().splitlines()[-1]); r = d["roofline"]
    print(wl, n, d["value"], d["ms_per_step"], "valu", (r.get("valu") or {}).get("insts"), (r.get("valu") or {}).get("frac"), "traffic", r.get("traffic"), "frac", r["frac"])
PY
done
