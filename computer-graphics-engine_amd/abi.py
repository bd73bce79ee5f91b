"""ctypes mirror of include/rt_whitted.h plus the host-side scene containers.

The structures here are the C ABI's descriptors byte for byte; `SceneData`
is the Python-side value of the reference's `Scene` (src/scene.h:28-33):
meshes (framework mesh.h:36-43), spheres (src/common.h:31-35), lights
(src/common.h:37-52), textures (framework image.h:11-18).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

RT_OK, RT_ERR_INVALID, RT_ERR_UNSUPPORTED, RT_ERR_DEVICE, RT_ERR_NOMEM = 0, 1, 2, 3, 4
RT_LIGHT_POINT, RT_LIGHT_SEGMENT, RT_LIGHT_PARALLELOGRAM = 0, 1, 2

FEATURE_FIELDS = (
    "enableShading", "enableRecursive", "enableHardShadow", "enableSoftShadow",
    "enableNormalInterp", "enableTextureMapping", "enableAccelStructure",
    # ExtraFeatures
    "enableEnvironmentMapping", "enableBvhSahBinning", "enableMotionBlur", "enableBloomEffect",
    "enableBilinearTextureFiltering", "enableMipmapTextureFiltering", "enableMultipleRaysPerPixel",
    "enableGlossyReflection", "enableTransparency", "enableDepthOfField",
)


class Features(C.Structure):
    """`Features` + `ExtraFeatures` (src/common.h:54-77): 17 one-byte bools."""
    _fields_ = [(n, C.c_uint8) for n in FEATURE_FIELDS]

    @classmethod
    def make(cls, **flags) -> "Features":
        f = cls()
        for k, v in flags.items():
            if k not in FEATURE_FIELDS:
                raise KeyError(k)
            setattr(f, k, 1 if v else 0)
        return f


class Ray(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("direction", C.c_float * 3), ("t", C.c_float)]


class Material(C.Structure):
    _fields_ = [("kd", C.c_float * 3), ("ks", C.c_float * 3), ("shininess", C.c_float),
                ("transparency", C.c_float), ("texture", C.c_int32)]


class MeshDesc(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("texcoords", C.POINTER(C.c_float)), ("triangles", C.POINTER(C.c_uint32)),
                ("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32), ("material", Material)]


class SphereDesc(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("material", Material)]


class LightDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("v", (C.c_float * 3) * 7)]


class TextureDesc(C.Structure):
    _fields_ = [("pixels", C.POINTER(C.c_float)), ("width", C.c_int32), ("height", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(MeshDesc)), ("n_meshes", C.c_uint32),
                ("spheres", C.POINTER(SphereDesc)), ("n_spheres", C.c_uint32),
                ("lights", C.POINTER(LightDesc)), ("n_lights", C.c_uint32),
                ("textures", C.POINTER(TextureDesc)), ("n_textures", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("quat", C.c_float * 4), ("half_w", C.c_float), ("half_h", C.c_float),
                ("look_at", C.c_float * 3)]


class Hit(C.Structure):
    _fields_ = [("hit", C.c_int32), ("prim", C.c_int32), ("t", C.c_float), ("normal", C.c_float * 3),
                ("material", Material)]


class RenderParams(C.Structure):
    _fields_ = [("depth", C.c_int32), ("parallelogram_n", C.c_int32), ("segment_n", C.c_int32), ("seed", C.c_uint32),
                ("rays_per_pixel_side", C.c_int32), ("samples_dof", C.c_int32), ("focus_plane_distance", C.c_float),
                ("blur_strength", C.c_float), ("bloom_scalar", C.c_float), ("bloom_threshold", C.c_float),
                ("bloom_debug_option", C.c_int32), ("glossy_reflections_cap", C.c_int32),
                ("rays_per_reflection", C.c_int32), ("alpha_modifier", C.c_float)]

    @classmethod
    def make(cls, depth=5, parallelogram_n=5, segment_n=25, seed=0, rays_per_pixel_side=3, samples_dof=5,
             focus_plane_distance=3.0, blur_strength=0.005, bloom_scalar=0.3, bloom_threshold=0.4,
             bloom_debug_option=0, glossy_reflections_cap=3, rays_per_reflection=40,
             alpha_modifier=1.0) -> "RenderParams":
        """Defaults = the reference's globals (src/render.cpp:14-22, src/light.cpp:12-13)."""
        return cls(depth, parallelogram_n, segment_n, seed, rays_per_pixel_side, samples_dof, focus_plane_distance,
                   blur_strength, bloom_scalar, bloom_threshold, bloom_debug_option, glossy_reflections_cap,
                   rays_per_reflection, alpha_modifier)


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays", "rays_primary", "rays_shadow", "rays_reflect", "rays_ref_equiv", "box_tests", "tri_tests",
        "sphere_tests", "node_fetches", "tri_fetches", "hits_shaded", "rays_queued", "group_tests", "rays_culled", "rays_shadow_skipped", "rays_exact")]

    def as_dict(self) -> dict:
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("t", "<f4")])
MATERIAL_DTYPE = np.dtype([("kd", "<f4", 3), ("ks", "<f4", 3), ("shininess", "<f4"), ("transparency", "<f4"),
                           ("texture", "<i4")])
HIT_DTYPE = np.dtype([("hit", "<i4"), ("prim", "<i4"), ("t", "<f4"), ("normal", "<f4", 3),
                      ("material", MATERIAL_DTYPE)])
assert RAY_DTYPE.itemsize == C.sizeof(Ray) == 28
assert HIT_DTYPE.itemsize == C.sizeof(Hit)


# ---------------------------------------------------------------------------
# Host-side scene value
# ---------------------------------------------------------------------------
@dataclass
class MaterialData:
    kd: Sequence[float] = (1.0, 1.0, 1.0)
    ks: Sequence[float] = (0.0, 0.0, 0.0)
    shininess: float = 1.0
    transparency: float = 1.0
    texture: int = -1

    def to_c(self) -> Material:
        m = Material()
        m.kd[:] = [float(x) for x in self.kd]
        m.ks[:] = [float(x) for x in self.ks]
        m.shininess = float(self.shininess)
        m.transparency = float(self.transparency)
        m.texture = int(self.texture)
        return m


@dataclass
class MeshData:
    positions: np.ndarray   # (n,3) f32
    normals: np.ndarray     # (n,3) f32
    texcoords: np.ndarray   # (n,2) f32
    triangles: np.ndarray   # (m,3) u32
    material: MaterialData = field(default_factory=MaterialData)

    def __post_init__(self):
        self.positions = np.ascontiguousarray(self.positions, dtype=np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(self.normals, dtype=np.float32).reshape(-1, 3)
        self.texcoords = np.ascontiguousarray(self.texcoords, dtype=np.float32).reshape(-1, 2)
        self.triangles = np.ascontiguousarray(self.triangles, dtype=np.uint32).reshape(-1, 3)
        assert len(self.normals) == len(self.positions) == len(self.texcoords)


@dataclass
class SphereData:
    center: Sequence[float]
    radius: float
    material: MaterialData = field(default_factory=MaterialData)


@dataclass
class LightData:
    kind: int
    v: np.ndarray  # (7,3) f32, members in declaration order (src/common.h:37-52)

    @classmethod
    def point(cls, position, color) -> "LightData":
        v = np.zeros((7, 3), np.float32)
        v[0], v[1] = position, color
        return cls(RT_LIGHT_POINT, v)

    @classmethod
    def segment(cls, endpoint0, endpoint1, color0, color1) -> "LightData":
        v = np.zeros((7, 3), np.float32)
        v[0], v[1], v[2], v[3] = endpoint0, endpoint1, color0, color1
        return cls(RT_LIGHT_SEGMENT, v)

    @classmethod
    def parallelogram(cls, v0, edge01, edge02, color0, color1, color2, color3) -> "LightData":
        v = np.zeros((7, 3), np.float32)
        v[:] = [v0, edge01, edge02, color0, color1, color2, color3]
        return cls(RT_LIGHT_PARALLELOGRAM, v)

    def to_c(self) -> LightDesc:
        d = LightDesc()
        d.kind = int(self.kind)
        vv = np.asarray(self.v, np.float32).reshape(7, 3)
        for i in range(7):
            d.v[i][:] = [float(x) for x in vv[i]]
        return d


@dataclass
class SceneData:
    meshes: List[MeshData] = field(default_factory=list)
    spheres: List[SphereData] = field(default_factory=list)
    lights: List[LightData] = field(default_factory=list)
    textures: List[np.ndarray] = field(default_factory=list)  # each (h,w,3) f32
    name: str = ""

    @property
    def n_triangles(self) -> int:
        return int(sum(len(m.triangles) for m in self.meshes))

    @property
    def n_prims(self) -> int:
        return self.n_triangles + len(self.spheres)

    # -- npz (de)serialisation: plain arrays only, loadable with allow_pickle=False
    def to_npz_dict(self) -> dict:
        d = {"n_meshes": np.int32(len(self.meshes)), "n_spheres": np.int32(len(self.spheres)),
             "n_lights": np.int32(len(self.lights)), "n_textures": np.int32(len(self.textures))}
        for i, m in enumerate(self.meshes):
            d[f"mesh{i}_positions"], d[f"mesh{i}_normals"] = m.positions, m.normals
            d[f"mesh{i}_texcoords"], d[f"mesh{i}_triangles"] = m.texcoords, m.triangles
            d[f"mesh{i}_material"] = _mat_arr(m.material)
        for i, s in enumerate(self.spheres):
            d[f"sphere{i}"] = np.asarray(list(s.center) + [s.radius], np.float32)
            d[f"sphere{i}_material"] = _mat_arr(s.material)
        for i, l in enumerate(self.lights):
            d[f"light{i}_kind"] = np.int32(l.kind)
            d[f"light{i}_v"] = np.asarray(l.v, np.float32)
        for i, t in enumerate(self.textures):
            d[f"texture{i}"] = np.asarray(t, np.float32)
        return d

    @classmethod
    def from_npz(cls, z, name="") -> "SceneData":
        s = cls(name=name)
        for i in range(int(z["n_meshes"])):
            s.meshes.append(MeshData(z[f"mesh{i}_positions"], z[f"mesh{i}_normals"], z[f"mesh{i}_texcoords"],
                                     z[f"mesh{i}_triangles"], _mat_from(z[f"mesh{i}_material"])))
        for i in range(int(z["n_spheres"])):
            a = z[f"sphere{i}"]
            s.spheres.append(SphereData(a[:3].tolist(), float(a[3]), _mat_from(z[f"sphere{i}_material"])))
        for i in range(int(z["n_lights"])):
            s.lights.append(LightData(int(z[f"light{i}_kind"]), np.asarray(z[f"light{i}_v"], np.float32)))
        for i in range(int(z["n_textures"])):
            s.textures.append(np.asarray(z[f"texture{i}"], np.float32))
        return s


def _mat_arr(m: MaterialData) -> np.ndarray:
    # 9 float32; the texture index is a small integer, exactly representable
    return np.asarray(list(m.kd) + list(m.ks) + [m.shininess, m.transparency, float(m.texture)], np.float32)


def _mat_from(a) -> MaterialData:
    a = np.asarray(a, np.float32)
    return MaterialData(tuple(a[0:3]), tuple(a[3:6]), a[6], a[7], int(a[8]))


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class SceneDescHolder:
    """Owns the ctypes descriptor tree of a SceneData (keeps numpy arrays alive)."""

    def __init__(self, scene: SceneData):
        self.scene = scene
        self._keep = []
        nm, ns, nl, nt = len(scene.meshes), len(scene.spheres), len(scene.lights), len(scene.textures)
        self.meshes = (MeshDesc * max(nm, 1))()
        for i, m in enumerate(scene.meshes):
            d = self.meshes[i]
            d.positions, d.normals, d.texcoords = _fptr(m.positions), _fptr(m.normals), _fptr(m.texcoords)
            d.triangles = m.triangles.ctypes.data_as(C.POINTER(C.c_uint32))
            d.n_vertices, d.n_triangles = len(m.positions), len(m.triangles)
            d.material = m.material.to_c()
        self.spheres = (SphereDesc * max(ns, 1))()
        for i, s in enumerate(scene.spheres):
            self.spheres[i].center[:] = [float(x) for x in s.center]
            self.spheres[i].radius = float(s.radius)
            self.spheres[i].material = s.material.to_c()
        self.lights = (LightDesc * max(nl, 1))()
        for i, l in enumerate(scene.lights):
            self.lights[i] = l.to_c()
        self.textures = (TextureDesc * max(nt, 1))()
        for i, t in enumerate(scene.textures):
            t = np.ascontiguousarray(t, np.float32)
            self._keep.append(t)
            self.textures[i].pixels = _fptr(t)
            self.textures[i].height, self.textures[i].width = t.shape[0], t.shape[1]
        self.desc = SceneDesc(self.meshes, nm, self.spheres, ns, self.lights, nl, self.textures, nt)


def lights_to_c(lights: Sequence[LightData]):
    arr = (LightDesc * max(len(lights), 1))()
    for i, l in enumerate(lights):
        arr[i] = l.to_c()
    return arr
