"""bench.py's one JSON line (the driver's contract): run it as the driver does — a subprocess, a small workload —
and check the fields the contract names, the roofline / cpu_baseline objects and their internal consistency."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def run_bench(*args):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], cwd=ROOT, capture_output=True, text=True,
                         timeout=900, env={**os.environ, "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS", "8")})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout"
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    steps, warmup = 10, 3
    d = run_bench("--gpus", "1", "--steps", str(steps), "--warmup", str(warmup), "--workload", "C2")
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), k
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "Mrays/s" and d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    c = d["config"]
    assert c["workload"] == "C2" and "model" not in c
    sizes = c["launch_frames"]
    assert (sum(sizes) if isinstance(sizes, list) else sum(int(k) * v for k, v in sizes.items())) == steps
    assert c["rays_bvh_traversed"] + c["rays_culled_before_traversal"] == c["rays_per_frame"]
    # value = rays traced over the timed steps / wall time
    assert abs(d["value"] - c["rays_per_frame"] / d["ms_per_step"] / 1e3) <= 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    # achieved = algorithmic bytes per launch / the launches' average duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) <= 0.01 * r["achieved"]
    assert r["chip_wide"]["launch_overlap"] > 0
    s = d["frame_ms_serial"]
    assert 0 < s["min"] <= s["mean"] <= s["max"] and s["default_pose"] > 0
    b = d["cpu_baseline"]
    assert b["kind"] == "port" and b["value"] > 0 and b["value_1core"] > 0 and b["cores"] >= 1 and b["cpu_model"] and b["sample"]
