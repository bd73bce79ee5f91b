"""The library's tuning options (rt_set_option) only change HOW the frame is computed: every setting
gives the bits of the default one — tile shapes (the bench's throughput setting is 8x8), per-level launches
instead of the fused recursion, no merged traversals, row-major tile order, no tile culling / launch rectangle,
XCD super-tiles, leaf-queue thresholds, the exact-topology engine alone (wide=0), the XCD-aware tile order, the
node prefetch, explicit tile shapes against the automatic one (tile_auto) — for full frames, packed strips and several frames per launch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEFAULTS = dict(tile_wlog=2, tile_hlog=2, fused=1, fused_from=0, merge_next=1, tile_order=1, tile_cull=1, launch_rect=1,
                super_log=0, leaf_flush=16, min_waves=4096, wide=1, xcd_order=1, xcd_block=4, prefetch=0, frame_major=1, skip_unlit=1, order_one_launch=1, tile_auto=1)
SETTINGS = [
    dict(tile_wlog=3, tile_hlog=3), dict(tile_wlog=2, tile_hlog=3), dict(tile_wlog=3, tile_hlog=2),
    dict(tile_wlog=1, tile_hlog=3), dict(tile_wlog=0, tile_hlog=0), dict(tile_wlog=3, tile_hlog=0), dict(tile_wlog=0, tile_hlog=3),
    dict(fused=0), dict(fused_from=1), dict(fused_from=2, tile_wlog=3, tile_hlog=3), dict(merge_next=0),
    dict(tile_order=0), dict(tile_cull=0), dict(launch_rect=0), dict(tile_cull=0, launch_rect=0, tile_order=0),
    dict(super_log=1), dict(leaf_flush=1), dict(leaf_flush=32, tile_wlog=3, tile_hlog=3), dict(leaf_flush=36), dict(leaf_flush=4, tile_wlog=3, tile_hlog=3), dict(min_waves=1),
    dict(wide=0), dict(xcd_order=0), dict(xcd_order=0, tile_wlog=3, tile_hlog=3), dict(prefetch=1), dict(xcd_block=0, prefetch=1, launch_rect=0), dict(xcd_block=6), dict(frame_major=0), dict(skip_unlit=0), dict(skip_unlit=0, tile_wlog=3, tile_hlog=3),
    dict(tile_auto=0), dict(tile_hlog=0, tile_wlog=6), dict(tile_hlog=1, tile_wlog=5, leaf_flush=32), dict(tile_wlog=4, tile_hlog=2),
    dict(order_one_launch=0), dict(order_one_launch=2), dict(order_one_launch=2, xcd_order=0, tile_wlog=3, tile_hlog=3),
]


@pytest.fixture(scope="module")
def frames(rt, abi, scenes):
    """default-setting frames of two scene / feature sets (lean and general kernel variants)"""
    cases = []
    for scene, flags in (("monkey", dict(enableShading=1, enableHardShadow=1, enableRecursive=1, enableAccelStructure=1)),
                         ("cornell_parallelogram", dict(enableShading=1, enableSoftShadow=1, enableRecursive=1,
                                                        enableNormalInterp=1, enableAccelStructure=1))):
        f = abi.Features.make(**flags)
        p = abi.RenderParams.make(depth=3, parallelogram_n=2)
        bvh = rt.bvh(rt.scene(scenes.load_fixture_scene(scene)), f)
        W, H = 136, 96
        fovy, aspect, look_at, dist, rot = scenes.default_camera_args(W, H)
        cams = [rt.camera(fovy, aspect, look_at, dist + 0.5 * i, (rot[0] + 9.0 * i, rot[1], rot[2])) for i in range(3)]
        want = [rt.render(bvh, c, f, p, W, H) for c in cams]
        assert (want[0] != 0).any()
        cases.append((bvh, f, p, W, H, cams, want))
    return cases


@pytest.mark.parametrize("setting", SETTINGS, ids=lambda s: ",".join(f"{k}={v}" for k, v in s.items()))
def test_setting_gives_the_default_bits(rt, frames, setting):
    try:
        for k, v in setting.items():
            rt.set_option(k, v)
        for bvh, f, p, W, H, cams, want in frames:
            got = rt.render(bvh, cams[0], f, p, W, H)
            assert np.array_equal(got.view(np.uint32), want[0].view(np.uint32)), "full frame"
            rows = np.concatenate([np.arange(s, s + 8) for s in (0, 24, 88)]).astype(np.int32)
            packed = torch.zeros((len(rows), W, 3), dtype=torch.float32, device="cuda")
            rt.render_device(bvh, cams[1], f, p, W, H, packed.data_ptr(), rows=rows, packed=True)
            out = torch.zeros((3, H, W, 3), dtype=torch.float32, device="cuda")
            rt.render_device_frames(bvh, cams, f, p, W, H, out.data_ptr())
            torch.cuda.synchronize()
            gp = packed.cpu().numpy()
            for i, y in enumerate(rows):
                assert np.array_equal(gp[i].view(np.uint32), want[1][H - 1 - y].view(np.uint32)), f"packed row {y}"
            go = out.cpu().numpy()
            for k in range(3):
                assert np.array_equal(go[k].view(np.uint32), want[k].view(np.uint32)), f"frame {k} of a launch"
    finally:
        for k, v in DEFAULTS.items():
            rt.set_option(k, v)
