"""rt_render_device_frames: several cameras in one launch give, frame by frame, the bits of one
rt_render per camera — the plain case in one launch (lean and general kernel variants, full frames and
packed row lists) and the cases that fall back to one launch per frame."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cams(rt, scenes, W, H, n):
    fovy, aspect, look_at, dist, rot = scenes.default_camera_args(W, H)
    return [rt.camera(fovy, aspect, (0.05 * i, -0.03 * i, 0.0), dist + 0.4 * i, (rot[0] + 13.0 * i, rot[1] - 19.0 * i, rot[2]))
            for i in range(n)]


@pytest.mark.parametrize("scene,flags,n", [
    ("monkey", dict(enableShading=1, enableHardShadow=1, enableRecursive=1, enableAccelStructure=1), 5),      # lean variant
    ("cube_textured", dict(enableShading=1, enableHardShadow=1, enableRecursive=1, enableNormalInterp=1,
                           enableTextureMapping=1, enableAccelStructure=1), 3),                                  # general variant
    ("cornell_parallelogram", dict(enableShading=1, enableSoftShadow=1, enableAccelStructure=1), 8),           # area light, 8 frames
    ("monkey", dict(enableShading=1, enableRecursive=1, enableGlossyReflection=1, enableAccelStructure=1), 2),  # falls back
    ("monkey", dict(enableShading=1, enableAccelStructure=1, enableMultipleRaysPerPixel=1), 2),                 # falls back
])
def test_frames_in_one_launch_match_single_renders(rt, abi, scenes, scene, flags, n):
    data = scenes.load_fixture_scene(scene)
    f = abi.Features.make(**flags)
    p = abi.RenderParams.make(depth=3, parallelogram_n=2, rays_per_reflection=2, rays_per_pixel_side=2)
    bvh = rt.bvh(rt.scene(data), f)
    W, H = 168, 96                       # 96 rows: whole 8-row tiles
    cams = _cams(rt, scenes, W, H, n)
    want = [rt.render(bvh, c, f, p, W, H) for c in cams]
    assert not np.array_equal(want[0], want[1])
    out = torch.zeros((n, H, W, 3), dtype=torch.float32, device="cuda")
    rt.render_device_frames(bvh, cams, f, p, W, H, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for k in range(n):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), f"frame {k}"
    # packed strips of a rank (rows 8..15, 40..47, 72..79), twice in a row (the row list is cached)
    rows = np.concatenate([np.arange(s, s + 8) for s in (8, 40, 72)]).astype(np.int32)
    packed = torch.zeros((n, len(rows), W, 3), dtype=torch.float32, device="cuda")
    for _ in range(2):
        packed.zero_()
        rt.render_device_frames(bvh, cams, f, p, W, H, packed.data_ptr(), rows=rows, packed=True)
        torch.cuda.synchronize()
        gp = packed.cpu().numpy()
        for k in range(n):
            for i, y in enumerate(rows):
                assert np.array_equal(gp[k, i].view(np.uint32), want[k][H - 1 - y].view(np.uint32)), (k, y)
    # a single frame afterwards still goes through the plain launch with its own row list
    one = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rt.render_device(bvh, cams[1], f, p, W, H, one.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(one.cpu().numpy().view(np.uint32), want[1].view(np.uint32))


def test_frames_with_ragged_rows_fall_back(rt, abi, scenes):
    """100 rows are not whole 8-row tiles: one launch per frame, same bits."""
    data = scenes.load_fixture_scene("monkey")
    f = abi.Features.make(enableShading=1, enableHardShadow=1, enableAccelStructure=1)
    p = abi.RenderParams.make(depth=0)
    bvh = rt.bvh(rt.scene(data), f)
    W, H = 120, 100
    cams = _cams(rt, scenes, W, H, 3)
    out = torch.zeros((3, H, W, 3), dtype=torch.float32, device="cuda")
    rt.render_device_frames(bvh, cams, f, p, W, H, out.data_ptr())
    torch.cuda.synchronize()
    for k, c in enumerate(cams):
        assert np.array_equal(out[k].cpu().numpy().view(np.uint32), rt.render(bvh, c, f, p, W, H).view(np.uint32))


def test_host_pointer_frames_call(rt, abi, scenes):
    """rt_render_frames: 11 cameras (one launch of 8, one of 3) into host memory = 11 calls of rt_render."""
    data = scenes.load_fixture_scene("monkey")
    f = abi.Features.make(enableShading=1, enableHardShadow=1, enableRecursive=1, enableAccelStructure=1)
    p = abi.RenderParams.make(depth=2)
    bvh = rt.bvh(rt.scene(data), f)
    W, H = 96, 64
    cams = _cams(rt, scenes, W, H, 11)
    got = rt.render_frames(bvh, cams, f, p, W, H)
    for k, c in enumerate(cams):
        assert np.array_equal(got[k].view(np.uint32), rt.render(bvh, c, f, p, W, H).view(np.uint32)), f"camera {k}"


@pytest.mark.parametrize("scene,flags", [
    ("monkey", dict(enableShading=1, enableHardShadow=1, enableRecursive=1, enableAccelStructure=1)),
    ("cornell_parallelogram", dict(enableShading=1, enableSoftShadow=1, enableRecursive=1, enableNormalInterp=1,
                                   enableAccelStructure=1)),
])
def test_host_frame_in_bands_of_rows_has_the_bits_of_one_launch(rt, abi, scenes, scene, flags):
    """rt_render into a HOST buffer (≙ renderRayTracing -> Screen) renders the frame as bands of rows on alternating
    streams and copies every band back while the later ones render ("host_bands"; default 0 = 4 bands for cheap frames): the bits of the one-launch
    path (host_bands 1) for whole frames and row ranges, band edges that fall on ragged rows included."""
    f = abi.Features.make(**flags)
    p = abi.RenderParams.make(depth=2, parallelogram_n=2, seed=3)
    bvh = rt.bvh(rt.scene(scenes.load_fixture_scene(scene)), f)
    W, H = 404, 517                      # neither a multiple of the tile sizes
    cam = rt.camera(*scenes.default_camera_args(W, H))
    try:
        rt.set_option("host_bands", 1)
        want = rt.render(bvh, cam, f, p, W, H)
        want_rows = rt.render(bvh, cam, f, p, W, H, rows=(37, 451))
        assert (want != 0).any()
        for n in (0, 2, 4, 7):
            rt.set_option("host_bands", n)
            for _ in range(2):           # the second call reuses streams, events and the device frame
                got = rt.render(bvh, cam, f, p, W, H)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), n
            got_rows = rt.render(bvh, cam, f, p, W, H, rows=(37, 451))
            assert np.array_equal(got_rows.view(np.uint32), want_rows.view(np.uint32)), n
    finally:
        rt.set_option("host_bands", 0)
