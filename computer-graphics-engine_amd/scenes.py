"""Scene inputs of the render path: reference-scene fixtures, the synthetic
dragon-scale stand-in, the default camera and the BASELINE.json configs.

Reference scenes (src/scene.cpp:5-92) are committed as plain-array fixtures
under tests/golden/scene_*.npz, written by oracle/make_golden.py from the
reference's own loader (framework/src/mesh.cpp:52-176). `data/dragon.obj` is
absent from the reference (.MISSING_LARGE_BLOBS), so configs C3–C5 use the
closed-form torus-knot mesh of SURVEY.md §8(d).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

from . import abi
from .abi import LightData, MaterialData, MeshData, SceneData

GOLDEN_DIR = Path(__file__).resolve().parent.parent / "tests" / "golden"

# src/config.h:16-21 — CameraConfig defaults (degrees)
DEFAULT_FOV_DEG = 50.0
DEFAULT_DISTANCE = 3.0
DEFAULT_LOOK_AT = (0.0, 0.0, 0.0)
DEFAULT_ROTATION_DEG = (20.0, 20.0, 0.0)

# monkey.mtl values (Kd .8, Ks .5, Ns 250, d 1)
KNOT_MATERIAL = MaterialData((0.8, 0.8, 0.8), (0.5, 0.5, 0.5), 250.0, 1.0, -1)


def radians(deg: float) -> np.float32:
    """glm::radians<float>: degrees * 0.01745329251994329576923690768489f."""
    return np.float32(deg) * np.float32(0.01745329251994329576923690768489)


def load_fixture_scene(name: str) -> SceneData:
    path = GOLDEN_DIR / f"scene_{name}.npz"
    with np.load(path, allow_pickle=False) as z:
        return SceneData.from_npz(z, name=name)


def center_and_scale_to_unit(positions: np.ndarray) -> np.ndarray:
    """centerAndScaleToUnitMesh (framework/src/mesh.cpp:151-176) in binary32:
    sequential std::accumulate, / float(N), max glm::length, (p - c) / maxD."""
    p = np.ascontiguousarray(positions, np.float32)
    center = np.cumsum(p, axis=0, dtype=np.float32)[-1] / np.float32(len(p))
    d = p - center
    sq = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    max_d = np.sqrt(sq).max()
    return (d / max_d).astype(np.float32)


def torus_knot_mesh(n_u: int = 3200, n_v: int = 128, tube_radius: float = 0.35,
                    material: Optional[MaterialData] = None) -> MeshData:
    """(2,3) torus-knot tube: n_u x n_v quads → 2*n_u*n_v triangles, n_u*n_v
    vertices with analytic normals and (u,v) = (i/n_u, j/n_v); outward winding
    (cross(v1-v0, v2-v0) points out of the tube). Centred and unit-scaled like
    the reference's loader does for its large models."""
    s = (np.arange(n_u, dtype=np.float64) / n_u) * (2.0 * math.pi)
    phi = (np.arange(n_v, dtype=np.float64) / n_v) * (2.0 * math.pi)
    c2, s2, c3, s3 = np.cos(2 * s), np.sin(2 * s), np.cos(3 * s), np.sin(3 * s)
    R = 2.0 + c3
    centre = np.stack([R * c2, R * s2, s3], axis=1)
    dR = -3.0 * s3
    tangent = np.stack([dR * c2 - 2.0 * R * s2, dR * s2 + 2.0 * R * c2, 3.0 * c3], axis=1)
    tangent /= np.linalg.norm(tangent, axis=1, keepdims=True)
    # normal of the torus the knot lies on (perpendicular to the tangent)
    n_t = np.stack([c3 * c2, c3 * s2, s3], axis=1)
    n_t -= (n_t * tangent).sum(1, keepdims=True) * tangent
    n_t /= np.linalg.norm(n_t, axis=1, keepdims=True)
    b = np.cross(tangent, n_t)
    cp, sp = np.cos(phi)[None, :, None], np.sin(phi)[None, :, None]
    normal = cp * n_t[:, None, :] + sp * b[:, None, :]
    pos = centre[:, None, :] + tube_radius * normal
    positions = center_and_scale_to_unit(pos.reshape(-1, 3).astype(np.float32))
    normals = normal.reshape(-1, 3).astype(np.float32)
    uu, vv = np.meshgrid(np.arange(n_u, dtype=np.float32) / np.float32(n_u),
                         np.arange(n_v, dtype=np.float32) / np.float32(n_v), indexing="ij")
    texcoords = np.stack([uu, vv], axis=-1).reshape(-1, 2).astype(np.float32)
    i = np.arange(n_u, dtype=np.int64)[:, None]
    j = np.arange(n_v, dtype=np.int64)[None, :]
    i1, j1 = (i + 1) % n_u, (j + 1) % n_v
    v00, v10, v11, v01 = i * n_v + j, i1 * n_v + j, i1 * n_v + j1, i * n_v + j1
    # cross(d/ds, d/dphi) points INTO the tube, so wind (v00, v11, v10) / (v00, v01, v11)
    tri_a = np.stack([v00, v11, v10], axis=-1)
    tri_b = np.stack([v00, v01, v11], axis=-1)
    triangles = np.stack([tri_a, tri_b], axis=2).reshape(-1, 3).astype(np.uint32)
    return MeshData(positions, normals, texcoords, triangles, material or KNOT_MATERIAL)


def dragon_standin_scene(n_u: int = 3200, n_v: int = 128) -> SceneData:
    """C3 scene: knot + PointLight{(-1,1,-1),(1,1,1)} (src/scene.cpp:72)."""
    s = SceneData(name=f"knot_{n_u}x{n_v}")
    s.meshes.append(torus_knot_mesh(n_u, n_v))
    s.lights.append(LightData.point((-1, 1, -1), (1, 1, 1)))
    return s


def dragon_standin_area_light_scene(n_u: int = 3200, n_v: int = 128) -> SceneData:
    """C4 scene: knot + ParallelogramLight (colours of src/scene.cpp:49-52)."""
    s = SceneData(name=f"knot_area_{n_u}x{n_v}")
    s.meshes.append(torus_knot_mesh(n_u, n_v))
    s.lights.append(LightData.parallelogram((-0.2, 1.0, -0.2), (0.4, 0, 0), (0, 0, 0.4),
                                            (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1)))
    return s


def composite_scene(cube_textured: SceneData, n_u: int = 3200, n_v: int = 128) -> SceneData:
    """C5 scene: knot + cube-textured.obj scaled x3 as a pedestal under it
    (positions*3 then y -= 3.6: top face at y = -0.6, below the unit-scaled
    knot), textures + point light as C3."""
    s = SceneData(name=f"composite_{n_u}x{n_v}")
    s.meshes.append(torus_knot_mesh(n_u, n_v))
    for m in cube_textured.meshes:
        pos = (m.positions * np.float32(3.0)).astype(np.float32)
        pos[:, 1] -= np.float32(3.6)
        mat = MaterialData(m.material.kd, m.material.ks, m.material.shininess, m.material.transparency,
                           m.material.texture)
        s.meshes.append(MeshData(pos, m.normals.copy(), m.texcoords.copy(), m.triangles.copy(), mat))
    s.textures = [t.copy() for t in cube_textured.textures]
    s.lights.append(LightData.point((-1, 1, -1), (1, 1, 1)))
    return s


def geometry_checksum(scene: SceneData) -> int:
    """Order-sensitive 64-bit checksum of all vertex positions and indices
    (guards committed goldens that depend on generated geometry)."""
    h = np.uint64(1469598103934665603)
    with np.errstate(over="ignore"):
        for m in scene.meshes:
            for arr in (m.positions.view(np.uint32).ravel(), m.triangles.ravel()):
                a = arr.astype(np.uint64)
                w = (np.arange(1, len(a) + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
                h = (h ^ np.bitwise_xor.reduce(a * w + (a << np.uint64(17)))) * np.uint64(1099511628211)
    return int(h)


@dataclass
class RenderConfig:
    """One BASELINE.json config: scene factory + features + explicit params."""
    name: str
    scene: str                 # fixture name or synthetic-scene key
    width: int
    height: int
    features: dict
    depth: int = 5
    parallelogram_n: int = 5
    segment_n: int = 25
    seed: int = 0

    def make_features(self) -> abi.Features:
        return abi.Features.make(**self.features)

    def make_params(self) -> abi.RenderParams:
        return abi.RenderParams.make(self.depth, self.parallelogram_n, self.segment_n, self.seed)


# SURVEY.md §8(d) C1..C5
CONFIGS = {
    "C1": RenderConfig("C1", "cornell", 512, 512, dict(enableShading=1, enableAccelStructure=1)),
    "C2": RenderConfig("C2", "monkey", 1024, 1024,
                       dict(enableShading=1, enableHardShadow=1, enableAccelStructure=1)),
    "C3": RenderConfig("C3", "knot", 1920, 1080,
                       dict(enableShading=1, enableRecursive=1, enableHardShadow=1, enableAccelStructure=1), depth=4),
    "C4": RenderConfig("C4", "knot_area", 1920, 1080,
                       dict(enableShading=1, enableSoftShadow=1, enableAccelStructure=1), parallelogram_n=8),
    "C5": RenderConfig("C5", "composite", 3840, 2160,
                       dict(enableShading=1, enableRecursive=1, enableHardShadow=1, enableNormalInterp=1,
                            enableTextureMapping=1, enableAccelStructure=1), depth=6),
}

_scene_cache = {}


def get_scene(key: str, n_u: int = 3200, n_v: int = 128) -> SceneData:
    ck = (key, n_u, n_v)
    if ck not in _scene_cache:
        if key == "knot":
            _scene_cache[ck] = dragon_standin_scene(n_u, n_v)
        elif key == "knot_area":
            _scene_cache[ck] = dragon_standin_area_light_scene(n_u, n_v)
        elif key == "composite":
            _scene_cache[ck] = composite_scene(load_fixture_scene("cube_textured"), n_u, n_v)
        else:
            _scene_cache[ck] = load_fixture_scene(key)
    return _scene_cache[ck]


def default_camera_args(width: int, height: int) -> Tuple:
    """(fovy, aspect, look_at, distance, rotation_euler) of the default
    CameraConfig (src/config.h:16-21) as main.cpp:518-519 passes them."""
    aspect = float(np.float32(width) / np.float32(height))  # Window::getAspectRatio (window.cpp:379-384)
    rot = tuple(float(radians(d)) for d in DEFAULT_ROTATION_DEG)
    return float(radians(DEFAULT_FOV_DEG)), aspect, DEFAULT_LOOK_AT, DEFAULT_DISTANCE, rot
