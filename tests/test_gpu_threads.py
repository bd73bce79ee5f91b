"""The reference's CLI mode renders its cameras from concurrent std::threads that share one
const BvhInterface& / Scene (src/main.cpp:512-531). The C-ABI calls must therefore be re-entrant on
one BVH handle: host-pointer calls from several threads, device calls on one stream per thread."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cameras(scenes, W, H, n):
    fovy, aspect, look_at, dist, rot = scenes.default_camera_args(W, H)
    return [(fovy, aspect, look_at, dist + 0.3 * i, (rot[0] + 11.0 * i, rot[1] - 17.0 * i, rot[2])) for i in range(n)]


def test_concurrent_host_threads_share_one_bvh(rt, abi, scenes):
    data = scenes.load_fixture_scene("monkey")
    f = abi.Features.make(enableShading=1, enableHardShadow=1, enableRecursive=1, enableAccelStructure=1)
    p = abi.RenderParams.make(depth=3)
    bvh = rt.bvh(rt.scene(data), f)
    W, H, n = 160, 120, 6
    cams = _cameras(scenes, W, H, n)
    serial = [rt.render(bvh, rt.camera(*c), f, p, W, H) for c in cams]
    rays = np.zeros(512, abi.RAY_DTYPE)
    rng = np.random.default_rng(3)
    o = rng.normal(size=(512, 3)) * 3
    d = -o / np.linalg.norm(o, axis=1, keepdims=True)
    rays["origin"], rays["direction"], rays["t"] = o.astype(np.float32), d.astype(np.float32), np.float32(3.402823466e+38)
    _, hits_serial = rt.intersect_batch(bvh, rays, f)
    out, hits, errors = [None] * n, [None] * n, []

    def work(i):
        try:
            for _ in range(3):      # several calls per thread: the threads interleave
                out[i] = rt.render(bvh, rt.camera(*cams[i]), f, p, W, H)
                hits[i] = rt.intersect_batch(bvh, rays, f)[1]
        except Exception as e:  # noqa: BLE001 - reported by the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(n):
        assert np.array_equal(out[i].view(np.uint32), serial[i].view(np.uint32)), f"camera {i}"
        assert np.array_equal(hits[i]["prim"], hits_serial["prim"])
    assert not np.array_equal(serial[0], serial[1]), "the cameras must differ"


def test_concurrent_threads_with_a_stream_each(rt, abi, scenes):
    data = scenes.load_fixture_scene("monkey")
    f = abi.Features.make(enableShading=1, enableHardShadow=1, enableRecursive=1, enableAccelStructure=1)
    p = abi.RenderParams.make(depth=2)
    bvh = rt.bvh(rt.scene(data), f)
    W, H, n = 200, 136, 4
    cams = _cameras(scenes, W, H, n)
    serial = [rt.render(bvh, rt.camera(*c), f, p, W, H) for c in cams]
    frames = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(n)]
    streams = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    errors = []

    def work(i):
        try:
            for _ in range(4):
                rt.render_device(bvh, rt.camera(*cams[i]), f, p, W, H, frames[i].data_ptr(), stream=streams[i].cuda_stream)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for i in range(n):
        assert np.array_equal(frames[i].cpu().numpy().view(np.uint32), serial[i].view(np.uint32)), f"camera {i}"


def test_concurrent_host_frames_in_bands_share_one_bvh(rt, abi, scenes):
    """Frames large enough for the banded host path (rows on alternating streams, copies back on a third), from more
    threads than the handle has host-call slots: every thread gets the bits of its own camera, call after call."""
    data = scenes.load_fixture_scene("monkey")
    f = abi.Features.make(enableShading=1, enableHardShadow=1, enableRecursive=1, enableAccelStructure=1)
    p = abi.RenderParams.make(depth=2)
    bvh = rt.bvh(rt.scene(data), f)
    W, H, n = 352, 304, 7
    cams = _cameras(scenes, W, H, n)
    rt.set_option("host_bands", 1)
    try:
        serial = [rt.render(bvh, rt.camera(*c), f, p, W, H) for c in cams]
    finally:
        rt.set_option("host_bands", 0)
    assert not np.array_equal(serial[0], serial[1])
    errors, bad = [], []

    def work(i):
        try:
            for k in range(4):
                got = rt.render(bvh, rt.camera(*cams[i]), f, p, W, H)
                if not np.array_equal(got.view(np.uint32), serial[i].view(np.uint32)):
                    bad.append((i, k))
                if k == 1:      # a counting call in between (one launch, the slot's counters)
                    _, cnt = rt.render(bvh, rt.camera(*cams[i]), f, p, W, H, count=True)
                    if cnt.rays_primary != W * H:
                        bad.append((i, "count"))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert not bad, bad
