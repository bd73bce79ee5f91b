"""ctypes binding of the C ABI (include/rt_whitted.h → csrc/librt_whitted.so).

This is plumbing over the shared library, not a second implementation: every
method forwards to one `rt_*` entry point. There is no CPU fallback — if the
library is missing it is built with hipcc, and if no HIP device is usable the
compute calls raise `RtError` (RT_ERR_DEVICE).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from . import abi

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = CSRC / "librt_whitted.so"

EXPORTS = [
    "rt_last_error", "rt_device_info", "rt_set_device", "rt_scene_create", "rt_scene_destroy",
    "rt_scene_update_lights", "rt_bvh_build", "rt_bvh_destroy", "rt_bvh_num_levels", "rt_bvh_num_leaves",
    "rt_bvh_num_nodes", "rt_bvh_num_prims", "rt_bvh_export", "rt_bvh_wide_check", "rt_camera_from_trackball", "rt_render",
    "rt_render_device", "rt_render_device_frames", "rt_frame_rect", "rt_render_frames", "rt_render_sharded", "rt_render_hits", "rt_trace_batch", "rt_intersect_batch", "rt_deinterleave_strips", "rt_deinterleave_frames",
    "rt_rand31", "rt_profile_enable", "rt_profile_read", "rt_profile_read_launches", "rt_debug_div_check", "rt_debug_tile_order", "rt_set_option", "rt_load_obj", "rt_load_preset", "rt_loaded_scene_desc",
    "rt_loaded_scene_destroy", "rt_io_last_error", "rt_write_bmp", "rt_bloom_device",
]


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"rt error {code}: {message}")
        self.code = code


def build_library(force: bool = False) -> Path:
    """Compile csrc/ for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.cpp")) + list(CSRC.glob("*.h")) + \
        [CSRC.parent.parent / "include" / "rt_whitted.h"]
    stale = (not LIB_PATH.exists()) or any(s.stat().st_mtime > LIB_PATH.stat().st_mtime for s in srcs)
    if force or stale:
        r = subprocess.run(["make", "-C", str(CSRC), "librt_whitted.so"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc build of librt_whitted.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


def _f3(x):
    return (C.c_float * 3)(*[float(v) for v in x])


class Scene:
    def __init__(self, lib: "Library", data: abi.SceneData):
        self.lib, self.data = lib, data
        holder = abi.SceneDescHolder(data)
        h = C.c_void_p()
        lib._check(lib.c.rt_scene_create(C.byref(holder.desc), C.byref(h)))
        self.handle = h

    def update_lights(self, lights: Sequence[abi.LightData]):
        arr = abi.lights_to_c(lights)
        self.lib._check(self.lib.c.rt_scene_update_lights(self.handle, arr, len(lights)))

    def close(self):
        if self.handle:
            self.lib.c.rt_scene_destroy(self.handle)
            self.handle = None


class Bvh:
    """≙ BvhInterface (src/bvh_interface.h:12-49)."""

    def __init__(self, lib: "Library", scene: Scene, features: abi.Features):
        self.lib, self.scene = lib, scene
        h = C.c_void_p()
        lib._check(lib.c.rt_bvh_build(scene.handle, C.byref(features), C.byref(h)))
        self.handle = h

    num_levels = property(lambda s: s.lib.c.rt_bvh_num_levels(s.handle))
    num_leaves = property(lambda s: s.lib.c.rt_bvh_num_leaves(s.handle))
    num_nodes = property(lambda s: s.lib.c.rt_bvh_num_nodes(s.handle))
    num_prims = property(lambda s: s.lib.c.rt_bvh_num_prims(s.handle))

    def export(self):
        n, p = self.num_nodes, self.num_prims
        aabb = np.zeros((n, 6), np.float32)
        meta = np.zeros((n, 6), np.int32)
        order = np.zeros(p, np.int32)
        self.lib._check(self.lib.c.rt_bvh_export(self.handle, aabb.ctypes.data, meta.ctypes.data, order.ctypes.data))
        return aabb, meta, order

    def wide_check(self) -> dict:
        """structural self-check of the wide traversal tree (host only); raises RtError when inconsistent"""
        st = (C.c_uint64 * 8)()
        self.lib._check(self.lib.c.rt_bvh_wide_check(self.handle, st))
        return dict(nodes=st[0], groups=st[1], depth=st[2], triangles=st[3], empty_slots=st[4], internal=st[5],
                    has_wide_tree=bool(st[6]))

    def close(self):
        if self.handle:
            self.lib.c.rt_bvh_destroy(self.handle)
            self.handle = None


class Library:
    def __init__(self, build: bool = True):
        path = build_library() if build else LIB_PATH
        if os.environ.get("RT_WHITTED_LIB"):  # diagnostic builds (e.g. -DRT_STAMPS), tools/ only
            path = Path(os.environ["RT_WHITTED_LIB"])
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.
        # Importing torch first makes the loader bind this library's NEEDED
        # libamdhip64.so.7 to the copy torch already mapped, so device pointers and
        # streams of torch tensors are valid here (two runtimes cannot share a GPU).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        self.c = C.CDLL(str(path))
        c = self.c
        c.rt_last_error.restype = C.c_char_p
        c.rt_device_info.argtypes = [C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        c.rt_scene_create.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(C.c_void_p)]
        c.rt_scene_destroy.argtypes = [C.c_void_p]
        c.rt_scene_update_lights.argtypes = [C.c_void_p, C.POINTER(abi.LightDesc), C.c_uint32]
        c.rt_bvh_build.argtypes = [C.c_void_p, C.POINTER(abi.Features), C.POINTER(C.c_void_p)]
        c.rt_bvh_destroy.argtypes = [C.c_void_p]
        for n in ("rt_bvh_num_levels", "rt_bvh_num_leaves", "rt_bvh_num_nodes", "rt_bvh_num_prims"):
            getattr(c, n).argtypes = [C.c_void_p]
        c.rt_bvh_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        c.rt_bvh_wide_check.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        c.rt_bloom_device.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(abi.RenderParams), C.c_void_p]
        c.rt_camera_from_trackball.argtypes = [C.c_float, C.c_float, C.c_float * 3, C.c_float, C.c_float * 3,
                                               C.POINTER(abi.Camera)]
        c.rt_render.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Features),
                                C.POINTER(abi.RenderParams), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                C.POINTER(abi.Counters)]
        c.rt_render_device.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Features),
                                       C.POINTER(abi.RenderParams), C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        c.rt_frame_rect.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.c_int32, C.POINTER(abi.Features),
                                    C.POINTER(abi.RenderParams), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
        c.rt_render_device_frames.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.c_int32, C.POINTER(abi.Features),
                                              C.POINTER(abi.RenderParams), C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                              C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p]
        c.rt_render_frames.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.c_int32, C.POINTER(abi.Features),
                                       C.POINTER(abi.RenderParams), C.c_int32, C.c_int32, C.c_void_p]
        c.rt_render_sharded.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(abi.Camera), C.POINTER(abi.Features),
                                        C.POINTER(abi.RenderParams), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]
        c.rt_render_hits.argtypes = [C.c_void_p, C.POINTER(abi.Camera), C.POINTER(abi.Features), C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_void_p]
        c.rt_trace_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(abi.Features),
                                     C.POINTER(abi.RenderParams), C.c_void_p, C.POINTER(abi.Counters)]
        c.rt_intersect_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(abi.Features), C.c_void_p,
                                         C.POINTER(abi.Counters)]
        c.rt_deinterleave_strips.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_void_p]
        c.rt_deinterleave_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_void_p]
        c.rt_profile_enable.argtypes = [C.c_void_p, C.c_int]
        c.rt_profile_read.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        c.rt_profile_read_launches.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.c_uint64,
                                               C.POINTER(C.c_uint64), C.c_int]
        c.rt_set_option.argtypes = [C.c_char_p, C.c_int]
        c.rt_io_last_error.restype = C.c_char_p
        c.rt_load_obj.argtypes = [C.c_char_p, C.c_int, C.POINTER(abi.LightDesc), C.c_uint32, C.POINTER(C.c_void_p)]
        c.rt_load_preset.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
        c.rt_loaded_scene_desc.argtypes = [C.c_void_p, C.POINTER(abi.SceneDesc)]
        c.rt_loaded_scene_destroy.argtypes = [C.c_void_p]
        c.rt_write_bmp.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        c.rt_debug_div_check.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64)]
        c.rt_debug_tile_order.argtypes = [C.c_int32] * 5 + [C.POINTER(C.c_int32), C.c_void_p, C.c_uint64,
                                                            C.POINTER(C.c_uint32)]
        c.rt_rand31.restype = C.c_uint32
        c.rt_rand31.argtypes = [C.c_uint32] * 6

    # -- errors
    def _check(self, code: int):
        if code != abi.RT_OK:
            raise RtError(code, (self.c.rt_last_error() or b"").decode())

    def device_info(self):
        n = C.c_int()
        buf = C.create_string_buffer(256)
        self._check(self.c.rt_device_info(C.byref(n), buf, 256))
        return n.value, buf.value.decode()

    # -- scene ingestion (host only)
    def _loaded_to_scene_data(self, handle, name) -> abi.SceneData:
        desc = abi.SceneDesc()
        if self.c.rt_loaded_scene_desc(handle, C.byref(desc)) != abi.RT_OK:
            raise RtError(abi.RT_ERR_INVALID, (self.c.rt_io_last_error() or b"").decode())
        scene = abi.SceneData(name=name)
        for i in range(desc.n_meshes):
            m = desc.meshes[i]
            nv, nt = m.n_vertices, m.n_triangles
            scene.meshes.append(abi.MeshData(
                np.ctypeslib.as_array(m.positions, (nv, 3)).copy(), np.ctypeslib.as_array(m.normals, (nv, 3)).copy(),
                np.ctypeslib.as_array(m.texcoords, (nv, 2)).copy(), np.ctypeslib.as_array(m.triangles, (nt, 3)).copy(),
                abi.MaterialData(tuple(m.material.kd), tuple(m.material.ks), m.material.shininess,
                                 m.material.transparency, m.material.texture)))
        for i in range(desc.n_spheres):
            s = desc.spheres[i]
            scene.spheres.append(abi.SphereData(list(s.center), s.radius, abi.MaterialData(
                tuple(s.material.kd), tuple(s.material.ks), s.material.shininess, s.material.transparency)))
        for i in range(desc.n_lights):
            l = desc.lights[i]
            scene.lights.append(abi.LightData(l.kind, np.array([list(l.v[k]) for k in range(7)], np.float32)))
        for i in range(desc.n_textures):
            t = desc.textures[i]
            scene.textures.append(np.ctypeslib.as_array(t.pixels, (t.height, t.width, 3)).copy())
        self.c.rt_loaded_scene_destroy(handle)
        return scene

    def load_preset(self, scene_type: int, data_dir: str, name: str = "") -> abi.SceneData:
        """≙ loadScenePrebuilt (src/scene.cpp:5-92)."""
        h = C.c_void_p()
        if self.c.rt_load_preset(scene_type, str(data_dir).encode(), C.byref(h)) != abi.RT_OK:
            raise RtError(abi.RT_ERR_INVALID, (self.c.rt_io_last_error() or b"").decode())
        return self._loaded_to_scene_data(h, name)

    def load_obj(self, path: str, center_and_normalize: bool = False, lights=()) -> abi.SceneData:
        """≙ loadSceneFromFile / loadMesh (src/scene.cpp:94-103, framework/src/mesh.cpp:52-176)."""
        h = C.c_void_p()
        arr = abi.lights_to_c(lights)
        if self.c.rt_load_obj(str(path).encode(), 1 if center_and_normalize else 0, arr, len(lights),
                              C.byref(h)) != abi.RT_OK:
            raise RtError(abi.RT_ERR_INVALID, (self.c.rt_io_last_error() or b"").decode())
        return self._loaded_to_scene_data(h, Path(path).stem)

    def write_bmp(self, path, frame: np.ndarray):
        """≙ Screen::writeBitmapToFile (src/screen.cpp:49-60); frame = (H, W, 3) fp32, row-flipped."""
        frame = np.ascontiguousarray(frame, np.float32)
        if self.c.rt_write_bmp(str(path).encode(), frame.ctypes.data, frame.shape[1], frame.shape[0]) != abi.RT_OK:
            raise RtError(abi.RT_ERR_INVALID, (self.c.rt_io_last_error() or b"").decode())

    def set_option(self, name: str, value: int):
        self._check(self.c.rt_set_option(name.encode(), int(value)))

    def set_device(self, device: int):
        self._check(self.c.rt_set_device(device))

    # -- objects
    def scene(self, data: abi.SceneData) -> Scene:
        return Scene(self, data)

    def bvh(self, scene: Scene, features: abi.Features) -> Bvh:
        return Bvh(self, scene, features)

    def camera(self, fovy, aspect, look_at, distance, rotation) -> abi.Camera:
        cam = abi.Camera()
        self._check(self.c.rt_camera_from_trackball(fovy, aspect, _f3(look_at), distance, _f3(rotation),
                                                    C.byref(cam)))
        return cam

    def bloom_device(self, rgb_ptr: int, width: int, height: int, params, stream: int = 0):
        """≙ renderBloomFilter on a whole row-flipped frame in device memory."""
        self._check(self.c.rt_bloom_device(C.c_void_p(rgb_ptr), width, height, C.byref(params),
                                           C.c_void_p(stream) if stream else None))

    # -- ≙ renderRayTracing (host framebuffer)
    def render(self, bvh: Bvh, cam, features, params, width, height, rows=None, count=False):
        y0, y1 = rows if rows is not None else (0, height)
        rgb = np.zeros((height, width, 3), np.float32)
        cnt = abi.Counters() if count else None
        self._check(self.c.rt_render(bvh.handle, C.byref(cam), C.byref(features), C.byref(params), width, height,
                                     y0, y1, rgb.ctypes.data, C.byref(cnt) if count else None))
        return (rgb, cnt) if count else rgb

    # -- same, framebuffer in HBM (pointers are ints, e.g. torch.Tensor.data_ptr())
    def render_device(self, bvh: Bvh, cam, features, params, width, height, rgb_ptr: int, rows=None, packed=False,
                      counters_ptr: int = 0, stream: int = 0):
        rows_arr = None
        n_rows = height
        if rows is not None:
            rows_arr = np.ascontiguousarray(rows, np.int32)
            n_rows = len(rows_arr)
        self._check(self.c.rt_render_device(
            bvh.handle, C.byref(cam), C.byref(features), C.byref(params), width, height,
            rows_arr.ctypes.data if rows_arr is not None else None, n_rows, 1 if packed else 0,
            C.c_void_p(rgb_ptr), C.c_void_p(counters_ptr) if counters_ptr else None,
            C.c_void_p(stream) if stream else None))

    # -- one frame on several devices / shards from one process (≙ the C++ host's multi-GPU path)
    def render_sharded(self, bvhs, cam, features, params, width, height, strip_rows=8, out_ptr: int = 0):
        """bvhs: one Bvh per shard, each built on its shard's device. Returns the (H, W, 3) host frame, or writes
        to device memory of shard 0's device when out_ptr is given."""
        arr = (C.c_void_p * len(bvhs))(*[b.handle for b in bvhs])
        rgb = None if out_ptr else np.zeros((height, width, 3), np.float32)
        self._check(self.c.rt_render_sharded(arr, len(bvhs), C.byref(cam), C.byref(features), C.byref(params), width,
                                             height, strip_rows, C.c_void_p(out_ptr) if out_ptr else rgb.ctypes.data,
                                             1 if out_ptr else 0))
        return rgb

    # -- several cameras at once into host memory: (n, H, W, 3)
    def render_frames(self, bvh: Bvh, cams, features, params, width, height, out=None):
        arr = (abi.Camera * len(cams))(*cams)
        rgb = out if out is not None else np.zeros((len(cams), height, width, 3), np.float32)
        assert rgb.dtype == np.float32 and rgb.flags.c_contiguous and rgb.shape == (len(cams), height, width, 3)
        self._check(self.c.rt_render_frames(bvh.handle, arr, len(cams), C.byref(features), C.byref(params), width,
                                            height, rgb.ctypes.data))
        return rgb

    # -- several frames (cameras) in one launch; frame k at rgb_ptr + k * frame bytes
    def render_device_frames(self, bvh: Bvh, cams, features, params, width, height, rgb_ptr: int, rows=None,
                             packed=False, frame_stride_floats: int = 0, stream: int = 0):
        arr = (abi.Camera * len(cams))(*cams)
        rows_arr = None
        n_rows = height
        if rows is not None:
            rows_arr = np.ascontiguousarray(rows, np.int32)
            n_rows = len(rows_arr)
        self._check(self.c.rt_render_device_frames(
            bvh.handle, arr, len(cams), C.byref(features), C.byref(params), width, height,
            rows_arr.ctypes.data if rows_arr is not None else None, n_rows, 1 if packed else 0,
            C.c_void_p(rgb_ptr), frame_stride_floats, C.c_void_p(stream) if stream else None))

    def render_hits(self, bvh: Bvh, cam, features, width, height):
        tb = np.zeros((height, width), np.uint32)
        pr = np.zeros((height, width), np.int32)
        self._check(self.c.rt_render_hits(bvh.handle, C.byref(cam), C.byref(features), width, height,
                                          tb.ctypes.data, pr.ctypes.data))
        return tb, pr

    # -- ≙ getFinalColor on a batch
    def trace_batch(self, bvh: Bvh, rays: np.ndarray, features, params, pixel_ids=None, count=False):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        rgb = np.zeros((len(rays), 3), np.float32)
        pid = None
        if pixel_ids is not None:
            pixel_ids = np.ascontiguousarray(pixel_ids, np.uint32)
            pid = pixel_ids.ctypes.data
        cnt = abi.Counters() if count else None
        self._check(self.c.rt_trace_batch(bvh.handle, rays.ctypes.data, len(rays), pid, C.byref(features),
                                          C.byref(params), rgb.ctypes.data, C.byref(cnt) if count else None))
        return (rgb, cnt) if count else rgb

    # -- ≙ BvhInterface::intersect on a batch
    def intersect_batch(self, bvh: Bvh, rays: np.ndarray, features, count=False):
        rays = np.ascontiguousarray(rays.copy(), dtype=abi.RAY_DTYPE)
        hits = np.zeros(len(rays), abi.HIT_DTYPE)
        cnt = abi.Counters() if count else None
        self._check(self.c.rt_intersect_batch(bvh.handle, rays.ctypes.data, len(rays), C.byref(features),
                                              hits.ctypes.data, C.byref(cnt) if count else None))
        return (rays, hits, cnt) if count else (rays, hits)

    def deinterleave_strips(self, gathered_ptr: int, frame_ptr: int, width, height, strip_rows, n_ranks,
                            stream: int = 0):
        self._check(self.c.rt_deinterleave_strips(C.c_void_p(gathered_ptr), C.c_void_p(frame_ptr), width, height,
                                                  strip_rows, n_ranks, C.c_void_p(stream) if stream else None))

    def deinterleave_frames(self, gathered_ptr: int, frames_ptr: int, width, height, strip_rows, n_ranks, n_frames,
                            batch_frames, stream: int = 0):
        self._check(self.c.rt_deinterleave_frames(C.c_void_p(gathered_ptr), C.c_void_p(frames_ptr), width, height,
                                                  strip_rows, n_ranks, n_frames, batch_frames,
                                                  C.c_void_p(stream) if stream else None))

    def profile_enable(self, bvh: Bvh, enable: bool):
        self._check(self.c.rt_profile_enable(bvh.handle, 1 if enable else 0))

    def profile_read(self, bvh: Bvh, stream: int = 0, reset: bool = True):
        """(summed k_trace_level milliseconds, launches) since the last reset; synchronises the stream."""
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.c.rt_profile_read(bvh.handle, C.c_void_p(stream) if stream else None, C.byref(ms),
                                           C.byref(n), 1 if reset else 0))
        return ms.value, int(n.value)

    def profile_read_launches(self, bvh: Bvh, stream: int = 0, reset: bool = True, cap: int = 65536):
        """durations (ms) of the trace-kernel launches on `stream` since the last reset, in launch order"""
        buf = (C.c_double * cap)()
        n = C.c_uint64()
        self._check(self.c.rt_profile_read_launches(bvh.handle, C.c_void_p(stream) if stream else None, buf, cap,
                                                    C.byref(n), 1 if reset else 0))
        return [buf[i] for i in range(min(cap, int(n.value)))]

    def debug_div_check(self, seed: int, mode: int, n_samples: int):
        """(mismatches, fast-path samples) of the slab test's exact division vs IEEE '/' on the device."""
        bad, fast = C.c_uint64(), C.c_uint64()
        self._check(self.c.rt_debug_div_check(seed, mode, n_samples, C.byref(bad), C.byref(fast)))
        return int(bad.value), int(fast.value), (self.c.rt_last_error() or b"").decode()

    def frame_rect(self, bvh: Bvh, cams, features, params, width: int, height: int):
        """(x0, y0, x1, y1): columns [x0, x1) x image rows [y0, y1) outside of which the frames rendered for the
        cameras `cams` (one, or a sequence: the union) are vec3(0)."""
        cams = list(cams) if isinstance(cams, (list, tuple)) else [cams]
        arr = (abi.Camera * len(cams))(*cams)
        r = (C.c_int32 * 4)()
        self._check(self.c.rt_frame_rect(bvh.handle, arr, len(cams), C.byref(features), C.byref(params), width, height, r))
        return tuple(int(v) for v in r)

    def debug_tile_order(self, width: int, n_rows: int, n_frames: int, tile_wlog: int, tile_hlog: int, rect=None):
        """(order, info) of the tile dispatch order the render calls would build (rt_debug_tile_order)."""
        tw, th = 1 << tile_wlog, 1 << tile_hlog
        cap = ((width + tw - 1) // tw) * ((n_rows + th - 1) // th)
        out = np.zeros(cap, dtype=np.uint32)
        info = (C.c_uint32 * 20)()
        r = (C.c_int32 * 4)(*rect) if rect is not None else None
        self._check(self.c.rt_debug_tile_order(width, n_rows, n_frames, tile_wlog, tile_hlog, r,
                                               out.ctypes.data_as(C.c_void_p), cap, info))
        info = [int(v) for v in info]
        return out[:info[0] * info[1]], info

    def rand31(self, seed, pixel, level, light, sample, dim) -> int:
        return int(self.c.rt_rand31(seed, pixel, level, light, sample, dim))
