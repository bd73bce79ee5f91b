"""rt_frame_rect's promise — every pixel outside the rectangle is vec3(0) in the rendered frame — over random
cameras (near, far, off-centre, inside the scene's bounds), several scenes and the pixel-loop variants; and that the
rectangle is the whole frame where no such statement is made."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    ("monkey", dict(enableShading=1, enableHardShadow=1, enableRecursive=1, enableAccelStructure=1), {}),
    ("monkey", dict(enableShading=1, enableAccelStructure=1, enableMultipleRaysPerPixel=1), dict(rays_per_pixel_side=2)),
    ("teapot", dict(enableShading=1, enableNormalInterp=1, enableTextureMapping=1, enableAccelStructure=1), {}),
    ("cornell_parallelogram", dict(enableShading=1, enableSoftShadow=1, enableAccelStructure=1), dict(parallelogram_n=2)),
]


@pytest.mark.parametrize("scene,flags,params", CASES, ids=lambda c: str(c) if isinstance(c, str) else "")
def test_pixels_outside_the_rectangle_are_background(rt, abi, scenes, scene, flags, params):
    f = abi.Features.make(**flags)
    p = abi.RenderParams.make(depth=2, **params)
    bvh = rt.bvh(rt.scene(scenes.load_fixture_scene(scene)), f)
    W, H = 232, 152
    fovy, aspect, look_at, dist, rot = scenes.default_camera_args(W, H)
    rng = np.random.default_rng(7)
    cropped = 0
    for k in range(14):
        d = float(dist * rng.choice([0.05, 0.6, 1.0, 2.5, 6.0]))
        la = tuple(float(look_at[i] + rng.uniform(-1.2, 1.2) * (k % 3 != 0)) for i in range(3))
        r = tuple(float(rot[i] + rng.uniform(-180.0, 180.0)) for i in range(3))
        cam = rt.camera(fovy, aspect, la, d, r)
        frame = rt.render(bvh, cam, f, p, W, H)
        x0, y0, x1, y1 = rt.frame_rect(bvh, cam, f, p, W, H)
        assert 0 <= x0 <= x1 <= W and 0 <= y0 <= y1 <= H
        outside = frame.copy()
        outside[H - y1:H - y0, x0:x1] = 0          # image row y is frame row H - 1 - y
        assert not outside.any(), f"camera {k}: a pixel outside {(x0, y0, x1, y1)} is not background"
        cropped += (x1 - x0) * (y1 - y0) < W * H
        # the union over two cameras covers both
        u = rt.frame_rect(bvh, [cam, rt.camera(fovy, aspect, look_at, dist, rot)], f, p, W, H)
        assert u[0] <= x0 and u[1] <= y0 and u[2] >= x1 and u[3] >= y1
    if scene != "cornell_parallelogram":   # (the camera sits inside the Cornell box: always the whole frame)
        assert cropped >= 3, "some of these cameras must see the object smaller than the frame"


def test_no_statement_means_the_whole_frame(rt, abi, scenes):
    W, H = 96, 64
    cam = rt.camera(*scenes.default_camera_args(W, H))
    p = abi.RenderParams.make(depth=1, samples_dof=2)
    mesh = rt.scene(scenes.load_fixture_scene("monkey"))
    for flags in (dict(enableShading=1, enableAccelStructure=1, enableDepthOfField=1), dict(enableShading=1)):
        f = abi.Features.make(**flags)
        assert rt.frame_rect(rt.bvh(mesh, f), cam, f, p, W, H) == (0, 0, W, H)
    f = abi.Features.make(enableShading=1, enableAccelStructure=1)
    spheres = rt.bvh(rt.scene(scenes.load_fixture_scene("spheres")), f)
    assert rt.frame_rect(spheres, cam, f, p, W, H) == (0, 0, W, H)
    try:
        rt.set_option("unpruned", 1)
        assert rt.frame_rect(rt.bvh(mesh, f), cam, f, p, W, H) == (0, 0, W, H)
    finally:
        rt.set_option("unpruned", 0)
