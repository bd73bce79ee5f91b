"""The C++ host shim with the reference's names (computer-graphics-engine_amd/host):
renderRayTracing / getFinalColor / BvhInterface::intersect called from C++ give
the same bits as the Python binding of the same C ABI (GPU), and the shim compiles
against the header (CPU)."""
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import ROOT, bits

HOST = ROOT / "computer-graphics-engine_amd" / "host"


def test_shim_compiles_without_gpu():
    """syntax + link-level check of the shim sources (no device needed)."""
    r = subprocess.run(["g++", "-std=c++20", "-fsyntax-only", "-Wall", "-Wextra", "reference_api.cpp",
                        "shim_selftest.cpp"], cwd=HOST, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def write_scene_file(path, abi, scene, features, cam_args, rays, depth):
    with open(path, "wb") as f:
        f.write(struct.pack("<I", len(scene.meshes)))
        for m in scene.meshes:
            f.write(struct.pack("<II", len(m.positions), len(m.triangles)))
            f.write(m.positions.tobytes() + m.normals.tobytes() + m.texcoords.tobytes() + m.triangles.tobytes())
            mat = m.material
            f.write(np.asarray(list(mat.kd) + list(mat.ks) + [mat.shininess, mat.transparency], np.float32).tobytes())
            if mat.texture >= 0:
                t = scene.textures[mat.texture]
                f.write(struct.pack("<II", t.shape[1], t.shape[0]) + np.ascontiguousarray(t, np.float32).tobytes())
            else:
                f.write(struct.pack("<II", 0, 0))
        f.write(struct.pack("<I", len(scene.lights)))
        for l in scene.lights:
            f.write(struct.pack("<i", l.kind) + np.asarray(l.v, np.float32).tobytes())
        f.write(bytes(features))
        fovy, aspect, look, dist, rot = cam_args
        f.write(np.asarray([fovy, aspect, *look, dist, *rot], np.float32).tobytes())
        f.write(struct.pack("<I", len(rays)) + rays.tobytes() + struct.pack("<i", depth))


@pytest.mark.gpu
def test_shim_matches_c_abi(rt, abi, scenes, tmp_path):
    subprocess.run(["make", "-C", str(HOST)], check=True, capture_output=True)
    # textured cube + monkey in one Scene (two meshes, one texture, specular materials)
    cube, monkey = scenes.load_fixture_scene("cube_textured"), scenes.load_fixture_scene("monkey")
    scene = abi.SceneData(meshes=[monkey.meshes[0]], lights=list(monkey.lights), textures=list(cube.textures))
    m = cube.meshes[0]
    pos = m.positions * np.float32(0.6) + np.float32([0.0, -1.1, 0.0])
    scene.meshes.append(abi.MeshData(pos, m.normals, m.texcoords, m.triangles, m.material))
    f = abi.Features.make(enableShading=1, enableRecursive=1, enableHardShadow=1, enableNormalInterp=1,
                          enableTextureMapping=1, enableAccelStructure=1)
    W, H, depth = 160, 96, 3
    cam_args = scenes.default_camera_args(W, H)
    rng = np.random.default_rng(3)
    rays = np.zeros(64, abi.RAY_DTYPE)
    o = rng.normal(size=(64, 3)) * 3
    d = rng.normal(size=(64, 3)) * 0.4 - o
    rays["origin"], rays["direction"] = o, d / np.linalg.norm(d, axis=1, keepdims=True)
    rays["t"] = np.float32(3.402823466e+38)
    write_scene_file(tmp_path / "scene.bin", abi, scene, f, cam_args, rays, depth)
    r = subprocess.run([str(HOST / "shim_selftest"), str(tmp_path / "scene.bin"), str(tmp_path / "out.bin"), str(W),
                        str(H)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = (tmp_path / "out.bin").read_bytes()
    w, h, levels, leaves = struct.unpack("<4i", raw[:16])
    frame = np.frombuffer(raw, np.float32, W * H * 3, 16).reshape(H, W, 3)
    recs = np.frombuffer(raw, np.float32, 64 * 8, 16 + W * H * 12).reshape(64, 8)

    bvh = rt.bvh(rt.scene(scene), f)
    assert (w, h, levels, leaves) == (W, H, bvh.num_levels, bvh.num_leaves)
    want = rt.render(bvh, rt.camera(*cam_args), f, abi.RenderParams.make(depth=5), W, H)  # literal 5, render.cpp:318
    assert np.array_equal(bits(frame), bits(want))
    assert (want != 0).any()
    col = rt.trace_batch(bvh, rays, f, abi.RenderParams.make(depth=depth))
    rr, hh = rt.intersect_batch(bvh, rays, f)
    assert np.array_equal(bits(recs[:, 0:3]), bits(col))
    assert np.array_equal(recs[:, 3] != 0, hh["hit"] == 1)
    assert np.array_equal(bits(recs[:, 4]), bits(rr["t"]))
    hit = hh["hit"] == 1
    assert hit.any()
    assert np.array_equal(bits(recs[hit, 5:8]), bits(hh["normal"][hit]))
