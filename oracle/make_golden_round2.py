#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors2.npz from the REAL reference code (oracle/_ref).

Container-only (needs /root/reference). TEST INFRASTRUCTURE ONLY. Data only — seeded
inputs and the reference's outputs on them:

  seg_*   sampleSegmentLight (src/light.cpp:19-26): lights, (index, sampleSize), the srand
          seed, the rand() value the reference drew, its position / colour
  par_*   sampleParallelogramLight (src/light.cpp:30-45): same, two draws
  wg_*    weightsGaussian (src/render.cpp:198-210) for several sigmas
  gi_*    getIntersection (src/render.cpp:238-243)
  bmp_*   Screen::writeBitmapToFile (src/screen.cpp:49-60) through the vendored
          stb_image_write.h (oracle/ref_bmp.cpp): the bytes of a small BMP and the SHA-256
          of larger ones, for frames holding negatives, > 1, NaN, infinities, exact k/255
"""
from __future__ import annotations

import hashlib
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent
sys.path.insert(0, str(ROOT))

from oracle.cpu_oracle import Reference, abi  # noqa: E402

GOLDEN = ROOT / "tests" / "golden"


def ray_c(o, d, t):
    r = abi.Ray()
    r.origin[:] = [float(x) for x in o]
    r.direction[:] = [float(x) for x in d]
    r.t = float(t)
    return r


def bmp_frames(rng):
    """name -> (H, W, 3) float32 frame"""
    out = {}
    small = rng.uniform(-0.25, 1.25, size=(5, 7, 3)).astype(np.float32)
    small[0, 0] = [np.nan, 0.5, 1.0]
    small[0, 1] = [np.inf, -np.inf, 0.0]
    small[1, 0] = [1.0, 0.99999994, 1.0000001]
    small[1, 1] = [-0.0, 1e-30, 0.0039215689]          # 1/255 and neighbours
    small[2, 2] = [0.0039215684, 0.0039215693, 0.5019608]
    out["small"] = small
    k = (np.arange(256, dtype=np.float32) / np.float32(255.0))
    ramp = np.zeros((3, 256, 3), np.float32)
    ramp[0, :, 0], ramp[0, :, 1], ramp[0, :, 2] = k, np.nextafter(k, np.float32(2)), np.nextafter(k, np.float32(-1))
    ramp[1] = ramp[0][:, ::-1]
    ramp[2] = rng.uniform(0, 1, size=(256, 3)).astype(np.float32)
    out["ramp"] = ramp
    out["noise"] = rng.uniform(-0.1, 1.1, size=(61, 83, 3)).astype(np.float32)
    out["one_pixel"] = np.array([[[0.25, 0.5, 0.75]]], np.float32)
    return out


def main():
    ref = Reference()
    assert ref.rand_max() == 2147483647
    rng = np.random.default_rng(20261005)
    vec = {}
    N = 3000
    # ---- segment lights
    seg_v = rng.normal(size=(N, 4, 3)).astype(np.float32)
    seg_v[:, 2:] = rng.uniform(0, 1.5, size=(N, 2, 3)).astype(np.float32)
    seg_n = rng.choice(np.array([1, 2, 5, 25, 64, 100], np.float32), size=N)
    seg_i = np.floor(rng.uniform(0, 1, size=N) * seg_n).astype(np.float32)
    seg_seed = rng.integers(0, 2**31, size=N, dtype=np.int64)
    seg_pos, seg_col, seg_r = np.zeros((N, 3), np.float32), np.zeros((N, 3), np.float32), np.zeros(N, np.int64)
    for i in range(N):
        seg_pos[i], seg_col[i], seg_r[i] = ref.sample_segment_light(seg_v[i], seg_seed[i], seg_i[i], seg_n[i])
    vec.update(seg_v=seg_v, seg_n=seg_n, seg_i=seg_i, seg_seed=seg_seed, seg_rand=seg_r, seg_pos=seg_pos,
               seg_col=seg_col)
    # ---- parallelogram lights
    par_v = rng.normal(size=(N, 7, 3)).astype(np.float32)
    par_v[:, 3:] = rng.uniform(0, 1.5, size=(N, 4, 3)).astype(np.float32)
    par_n = rng.choice(np.array([1, 2, 5, 8, 16], np.float32), size=(N, 2))
    par_ik = np.floor(rng.uniform(0, 1, size=(N, 2)) * par_n).astype(np.float32)
    par_seed = rng.integers(0, 2**31, size=N, dtype=np.int64)
    par_pos, par_col = np.zeros((N, 3), np.float32), np.zeros((N, 3), np.float32)
    par_r = np.zeros((N, 2), np.int64)
    for i in range(N):
        par_pos[i], par_col[i], par_r[i, 0], par_r[i, 1] = ref.sample_parallelogram_light(
            par_v[i], par_seed[i], par_ik[i, 0], par_ik[i, 1], par_n[i, 0], par_n[i, 1])
    vec.update(par_v=par_v, par_n=par_n, par_ik=par_ik, par_seed=par_seed, par_rand=par_r, par_pos=par_pos,
               par_col=par_col)
    # ---- weightsGaussian
    sig = np.array([1.0, 0.5, 2.0, 0.84089642, 3.3, 10.0, 0.25], np.float32)
    vec.update(wg_sigma=sig, wg_out=np.stack([ref.weights_gaussian(float(x)) for x in sig]))
    # ---- getIntersection
    M = 2000
    gi_o = (rng.normal(size=(M, 3)) * 2).astype(np.float32)
    gi_d = rng.normal(size=(M, 3)).astype(np.float32)
    gi_n = rng.normal(size=(M, 3)).astype(np.float32)
    gi_n /= np.linalg.norm(gi_n, axis=1, keepdims=True).astype(np.float32)
    gi_D = rng.normal(size=M).astype(np.float32) * 3
    gi_out = np.zeros((M, 3), np.float32)
    for i in range(M):
        gi_out[i] = ref.get_intersection(ray_c(gi_o[i], gi_d[i], 1.0), gi_D[i], gi_n[i])
    vec.update(gi_o=gi_o, gi_d=gi_d, gi_n=gi_n, gi_D=gi_D, gi_out=gi_out)
    # ---- BMP bytes
    with tempfile.TemporaryDirectory() as td:
        for name, frame in bmp_frames(rng).items():
            path = Path(td) / f"{name}.bmp"
            assert ref.write_bmp(path, frame) != 0
            data = path.read_bytes()
            vec[f"bmp_{name}_frame"] = frame
            vec[f"bmp_{name}_sha256"] = np.frombuffer(hashlib.sha256(data).digest(), np.uint8)
            vec[f"bmp_{name}_size"] = np.int64(len(data))
            if len(data) <= 4096:
                vec[f"bmp_{name}_bytes"] = np.frombuffer(data, np.uint8)
    np.savez_compressed(GOLDEN / "ref_vectors2.npz", **vec)
    print("ref_vectors2.npz written:", {k: v.shape for k, v in vec.items() if k.endswith("_out") or k.endswith("_pos")})


if __name__ == "__main__":
    main()
