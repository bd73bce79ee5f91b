"""The traversal tree the kernels walk (csrc/host_wide.cpp, csrc/rt_wide.h): structural self-check on the host side
of a built BVH handle (rt_bvh_wide_check), which scenes get one, and that rays the wide engine hands back to the
exact-topology engine are counted (rt_counters::rays_exact). Results themselves are covered by the parity tests:
tests/test_gpu_parity.py, tests/test_gpu_options.py (wide = 0 gives the same bits), tests/test_gpu_fuzz.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,wide", [("cube", True), ("cube_textured", True), ("cornell", True), ("monkey", True),
                                       ("teapot", True), ("cornell_parallelogram", True),
                                       ("single_triangle", False),   # the reference tree's root is a leaf
                                       ("spheres", False)])          # sphere test assumes |d| = 1: exact engine only
def test_wide_tree_is_consistent(rt, abi, scenes, name, wide):
    data = scenes.load_fixture_scene(name)
    for sah in (0, 1):  # the reference tree's split policy decides the reference leaves, not the wide tree's shape
        bvh = rt.bvh(rt.scene(data), abi.Features.make(enableAccelStructure=1, enableBvhSahBinning=sah))
        st = bvh.wide_check()
        assert st["has_wide_tree"] == wide
        if wide:
            assert st["triangles"] == bvh.num_prims == data.n_triangles
            assert st["internal"] == st["nodes"] - 1            # every node but the root is some node's child
            assert st["groups"] * 4 >= st["triangles"] >= st["groups"]
            assert 4 * st["nodes"] == st["internal"] + st["groups"] + st["empty_slots"]


def test_wide_tree_of_the_bench_scene(rt, abi, scenes):
    bvh = rt.bvh(rt.scene(scenes.get_scene("knot")), abi.Features.make(enableAccelStructure=1))
    st = bvh.wide_check()
    assert st["has_wide_tree"] and st["triangles"] == 819200
    assert st["depth"] <= 13 and st["groups"] < 230000          # 3.8 triangles per group, deques of 13 slots


def test_rays_the_wide_engine_cannot_take_are_counted(rt, abi, scenes):
    """axis-parallel rays (a zero direction component) are not all_ok: the exact-topology engine traces them"""
    data = scenes.load_fixture_scene("monkey")
    f = abi.Features.make(enableAccelStructure=1)
    bvh = rt.bvh(rt.scene(data), f)
    rays = np.zeros(256, abi.RAY_DTYPE)
    rng = np.random.default_rng(5)
    rays["origin"] = rng.uniform(-0.5, 0.5, size=(256, 3)) + np.float32([0, 0, 3])
    rays["direction"] = [0.0, 0.0, -1.0]
    rays["direction"][128:] = rng.normal(size=(128, 3)) * 0.05 + np.float32([0, 0, -1])
    rays["t"] = np.float32(3.402823466e+38)
    _, hits, cnt = rt.intersect_batch(bvh, rays, f, count=True)
    assert cnt.rays == 256 and cnt.rays_exact == 128
    assert hits["hit"][:128].any() and hits["hit"][128:].any()
