# driver-argument bench twice + the 200-step run; prints one summary line each
mkdir -p gpurun_out/r2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2/b_driver2.json 2> gpurun_out/r2/b_driver2.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/b_driver3.json 2>/dev/null
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r2/b_200b.json 2> gpurun_out/r2/b_200b.err
for f in b_driver2 b_driver3 b_200b; do python - "$f" <<'PY'
import json, sys
f = sys.argv[1]
d = json.load(open(f"gpurun_out/r2/{f}.json"))
r = d["roofline"]
print(f, d["value"], d["ms_per_step"], d["config"]["launch_frames"], r["avg_launch_us"], r["frac"], r["other_launches"], d["frame_ms_serial"]["mean"])
PY
done
