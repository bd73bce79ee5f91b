"""The forwarding code compiled against the reference's REAL headers (container only).

computer-graphics-engine_amd/host/reference_drop_in.cpp defines renderRayTracing, getFinalColor and every
BvhInterface member with the signatures of src/render.h:32,35 and src/bvh_interface.h:12-49, on top of the same
forwarding code (host/forwarding.hpp) that runs behind reference_api.hpp on the GPU box. forwarding.hpp's
static_asserts tie rt_features / rt_ray / rt_material to the real Features / Ray / Material (sizeof + offsetof).
"""
import subprocess
from pathlib import Path

import pytest

from conftest import ROOT

REF = Path("/root/reference")
HOST = ROOT / "computer-graphics-engine_amd" / "host"


@pytest.mark.reference
def test_drop_in_compiles_against_the_reference_headers(tmp_path):
    if not REF.exists():
        pytest.skip("needs /root/reference (container only)")
    tp = REF / "framework" / "third_party"
    obj = tmp_path / "drop_in.o"
    cmd = ["g++", "-std=c++20", "-c", "-Wall", "-Wextra", "-fno-access-control", f'-DDATA_DIR="{REF}/data/"',
           f"-I{REF}/src", f"-I{REF}/framework/include", f"-I{REF}/framework/include/framework", f"-I{tp}/glm",
           f"-I{tp}/fmt/include", str(HOST / "reference_drop_in.cpp"), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    # the translation unit defines exactly the reference's symbols for this path
    syms = subprocess.run(["nm", "-C", "--defined-only", str(obj)], capture_output=True, text=True, check=True).stdout
    for want in ("renderRayTracing(Scene const&, Trackball const&, BvhInterface const&, Screen&, Features const&)",
                 "getFinalColor(Scene const&, BvhInterface const&, Ray, Features const&, int)",
                 "BvhInterface::BvhInterface(Scene*, Features const&)",
                 "BvhInterface::intersect(Ray&, HitInfo&, Features const&) const",
                 "BvhInterface::numLevels() const", "BvhInterface::numLeaves() const",
                 "BvhInterface::setRecursionLevel(int) const"):
        assert want in syms, want
    # and needs nothing of the reference's own render / BVH / intersection code
    undef = subprocess.run(["nm", "-C", "--undefined-only", str(obj)], capture_output=True, text=True, check=True).stdout
    for banned in ("intersectRayWith", "computeLightContribution", "BoundingVolumeHierarchy::", "recursiveRayTrace"):
        assert banned not in undef, banned
    for need in ("rt_render", "rt_trace_batch", "rt_intersect_batch", "rt_scene_create", "rt_bvh_build"):
        assert need in undef, need


def test_shim_and_drop_in_share_one_forwarding_source():
    """reference_api.cpp (runs on the GPU box) and reference_drop_in.cpp (real headers) both include forwarding.hpp
    and hold no C-ABI compute calls of their own."""
    for name in ("reference_api.cpp", "reference_drop_in.cpp"):
        text = (HOST / name).read_text()
        assert '#include "forwarding.hpp"' in text
        for call in ("rt_render(", "rt_trace_batch(", "rt_intersect_batch(", "rt_scene_create(", "rt_bvh_build("):
            assert call not in text, (name, call)
