mkdir -p gpurun_out/r2
for fl in 2 3 4; do
python bench.py --steps 192 --warmup 16 --no-cpu-baseline --frames-in-flight $fl > gpurun_out/r2/b_fl$fl.json 2> gpurun_out/r2/b_fl$fl.err
python - $fl <<'PY'
import json, sys
fl = sys.argv[1]
d = json.load(open(f"gpurun_out/r2/b_fl{fl}.json"))
print("in flight", fl, "Mrays/s", d["value"], "ms", d["ms_per_step"], "launch_us", d["roofline"]["avg_launch_us"])
PY
done
