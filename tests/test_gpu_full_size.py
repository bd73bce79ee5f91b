"""BASELINE.json configs at their FULL sizes (C3 1920x1080 depth 4, C4 1920x1080 with an
8x8 area light, C5 3840x2160 composite with textures + normal interpolation + depth 6).

The oracle needs seconds per 8-row strip at these sizes, so it checks a few strips of the
full-resolution frame (the same rays the full frame traces: rows are rendered through the
row-range entry point), and the rest of the frame is covered by size-independent
properties: the pruned traversal equals the reference policy bit for bit, strips rendered
separately equal the frame rendered at once, and rendering is deterministic."""
import numpy as np
import pytest

from conftest import bits
from test_gpu_parity import assert_colors_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(rt, oracle, scenes):
    cache = {}

    def get(cfg_name):
        if cfg_name not in cache:
            c = scenes.CONFIGS[cfg_name]
            data = scenes.get_scene(c.scene)
            f = c.make_features()
            cache[cfg_name] = (c, data, f, c.make_params(), rt.bvh(rt.scene(data), f), oracle.build(data, f))
        return cache[cfg_name]

    return get


@pytest.mark.parametrize("cfg_name,strips", [("C3", (60, 67, 75)), ("C4", (66,)), ("C5", (120, 150))])
def test_full_resolution_strips_match_oracle(rt, oracle, scenes, full, cfg_name, strips):
    c, data, f, p, bvh, ob = full(cfg_name)
    args = scenes.default_camera_args(c.width, c.height)
    cam, ocam = rt.camera(*args), oracle.camera(*args)
    tb, pr = rt.render_hits(bvh, cam, f, c.width, c.height)
    hits_seen = 0
    for s in strips:
        y0, y1 = 8 * s, 8 * s + 8
        orgb, otb, opr, ocnt = oracle.render(ob, ocam, f, p, c.width, c.height, rows=(y0, y1), want_hits=True)
        sl = slice(c.height - y1, c.height - y0)                      # Screen::setPixel row flip
        assert np.array_equal(pr[sl], opr[sl]), f"strip {s}: prim ids differ"
        assert np.array_equal(tb[sl], otb[sl]), f"strip {s}: t bits differ"
        rgb, cnt = rt.render(bvh, cam, f, p, c.width, c.height, rows=(y0, y1), count=True)
        assert_colors_close(rgb[sl], orgb[sl])
        assert cnt.rays_ref_equiv == ocnt.rays
        hits_seen += int((opr[sl] >= 0).sum())
    assert hits_seen > 1000, "the checked strips must cross the object"


@pytest.mark.parametrize("cfg_name", ["C3", "C4", "C5"])
def test_full_frame_properties(rt, scenes, full, cfg_name):
    c, data, f, p, bvh, ob = full(cfg_name)
    cam = rt.camera(*scenes.default_camera_args(c.width, c.height))
    frame, cnt = rt.render(bvh, cam, f, p, c.width, c.height, count=True)
    # 1. deterministic
    again = rt.render(bvh, cam, f, p, c.width, c.height)
    assert np.array_equal(bits(frame), bits(again))
    # 2. reference traversal policy (no pruning) gives the same bits, with more tests
    rt.set_option("unpruned", 1)
    try:
        ref_policy, cnt_ref = rt.render(bvh, cam, f, p, c.width, c.height, count=True)
        tb_ref, pr_ref = rt.render_hits(bvh, cam, f, c.width, c.height)
    finally:
        rt.set_option("unpruned", 0)
    tb, pr = rt.render_hits(bvh, cam, f, c.width, c.height)
    assert np.array_equal(bits(frame), bits(ref_policy))
    assert np.array_equal(tb, tb_ref) and np.array_equal(pr, pr_ref)
    assert cnt.rays == cnt_ref.rays and cnt.tri_tests < cnt_ref.tri_tests
    # 3. the frame is the union of its strips (what each GPU renders in the sharded run)
    parts = np.zeros_like(frame)
    for y0 in range(0, c.height, 360):
        part = rt.render(bvh, cam, f, p, c.width, c.height, rows=(y0, min(y0 + 360, c.height)))
        parts += part
    assert np.array_equal(bits(parts), bits(frame))
    # 4. hit records are self-consistent: a hit has a finite positive t, a miss keeps FLT_MAX
    t = tb.view(np.float32)
    assert np.all(t[pr >= 0] > 0) and np.all(np.isfinite(t[pr >= 0]))
    assert np.all(tb[pr < 0] == np.float32(3.402823466e+38).view(np.uint32))
    assert cnt.rays_primary == c.width * c.height and cnt.hits_shaded > 0
    if c.features.get("enableRecursive"):
        assert cnt.rays_reflect > 0 and cnt.rays_ref_equiv > cnt.rays


def test_c4_area_light_statistics(rt, scenes, full):
    """C4: 64 jittered shadow rays per hit. Different RNG seeds must agree in the mean
    (what 'statistical parity' with the reference's rand() means) and differ in detail."""
    c, data, f, p, bvh, ob = full("C4")
    W, H = 480, 270
    cam = rt.camera(*scenes.default_camera_args(1920, 1080))
    import ctypes
    imgs = []
    for seed in (0, 1):
        q = type(p)(p.depth, p.parallelogram_n, p.segment_n, seed)
        img, cnt = rt.render(bvh, cam, f, q, W, H, count=True)
        assert cnt.rays_shadow + cnt.rays_shadow_skipped == 64 * cnt.hits_shaded
        assert cnt.rays_shadow_skipped > 0 and cnt.rays_shadow > 0   # part of the knot faces away from the light
        imgs.append(img)
    a, b = imgs
    lit = (a > 0).any(axis=2)
    assert lit.sum() > 1000 and not np.array_equal(a, b)
    blocks = lambda x: x[: H // 16 * 16, : W // 16 * 16].reshape(H // 16, 16, W // 16, 16, 3).mean(axis=(1, 3))
    diff = np.abs(blocks(a) - blocks(b))
    assert diff.mean() < 2e-3


def test_leaves_above_32_primitives_take_the_ungrouped_path(rt, oracle, abi, scenes):
    """MAX_DEPTH caps the tree at 2^15 leaves: a 1.3 M-triangle mesh gets ~40 triangles per leaf,
    above the culling groups' limit of 32 (rt_layout.h kGroupLeafMax), so the traversal tests
    those leaves whole, like the reference. Hit records bit-identical, colours within tolerance,
    pruned == unpruned."""
    data = scenes.get_scene("knot", n_u=4096, n_v=160)       # 1 310 720 triangles
    assert data.n_triangles > 32 * (1 << 15)
    f = abi.Features.make(enableShading=1, enableRecursive=1, enableHardShadow=1, enableAccelStructure=1)
    p = abi.RenderParams.make(depth=2)
    bvh = rt.bvh(rt.scene(data), f)
    ob = oracle.build(data, f)
    assert bvh.num_leaves == ob.num_leaves and bvh.num_levels == ob.num_levels
    W, H = 160, 96
    args = scenes.default_camera_args(W, H)
    cam, ocam = rt.camera(*args), oracle.camera(*args)
    tb, pr = rt.render_hits(bvh, cam, f, W, H)
    orgb, otb, opr, _ = oracle.render(ob, ocam, f, p, W, H, want_hits=True)
    assert (opr >= 0).sum() > 500
    assert np.array_equal(pr, opr) and np.array_equal(tb, otb)
    rgb = rt.render(bvh, cam, f, p, W, H)
    assert_colors_close(rgb, orgb)
    rt.set_option("unpruned", 1)
    try:
        ref_policy = rt.render(bvh, cam, f, p, W, H)
    finally:
        rt.set_option("unpruned", 0)
    assert np.array_equal(bits(ref_policy), bits(rgb))
