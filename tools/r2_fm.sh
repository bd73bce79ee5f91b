mkdir -p gpurun_out/r2
for fm in 1 0; do for st in 20 200; do
python bench.py --steps $st --warmup 5 --no-cpu-baseline --set frame_major=$fm > gpurun_out/r2/b_fm${fm}_$st.json 2> gpurun_out/r2/b_fm${fm}_$st.err
python - $fm $st <<'PY'
import json, sys
fm, st = sys.argv[1:3]
d = json.load(open(f"gpurun_out/r2/b_fm{fm}_{st}.json"))
print("frame_major", fm, "steps", st, "Mrays/s", d["value"], "ms", d["ms_per_step"], "serial", d["frame_ms_serial"]["mean"])
PY
done; done
