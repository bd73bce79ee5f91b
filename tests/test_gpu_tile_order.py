"""The tile dispatch order is a scheduling choice, but it must start every tile of the launched grid exactly once:
checked directly on the order the device builds (rt_debug_tile_order) — both builders (one launch with the counters
in LDS; count / scan / scatter kernels), with and without the XCD lists, one frame and several per launch, whole
grids and launch rectangles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    # width, rows per frame, frames, tile_wlog, tile_hlog, rect (tile_x0, tile_y0, tiles_w, tiles_h) or None
    (1920, 1080, 1, 2, 2, None),
    (1920, 1080, 1, 2, 2, (37, 51, 275, 190)),
    (1920, 1080, 1, 3, 3, (20, 11, 140, 100)),
    (136, 96, 1, 2, 2, None),
    (136, 96, 3, 3, 3, None),
    (1920, 1080, 8, 3, 3, (30, 0, 151, 8 * 135)),
    (640, 136, 32, 3, 3, None),
    (333, 64, 1, 0, 0, (5, 3, 300, 50)),
    (3840, 2160, 1, 2, 2, (1, 1, 958, 538)),
    (8, 8, 1, 3, 3, None),
]


def tile_xcd(tx, ty, bxlog, bylog):
    return ((tx >> bxlog) + 3 * (ty >> bylog)) & 7


@pytest.mark.parametrize("one_launch", [2, 0])
@pytest.mark.parametrize("xcd", [1, 0])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}x{c[1]}x{c[2]}-t{c[3]}{c[4]}-{'rect' if c[5] else 'full'}")
def test_order_is_a_permutation_of_the_launched_tiles(rt, case, xcd, one_launch):
    W, rows, frames, wl, hl, rect = case
    try:
        rt.set_option("order_one_launch", one_launch)
        rt.set_option("xcd_order", xcd)
        order, info = rt.debug_tile_order(W, rows * frames, frames, wl, hl, rect)
    finally:
        rt.set_option("order_one_launch", 1)
        rt.set_option("xcd_order", 1)
    gw, gh = info[0], info[1]
    n = gw * gh
    assert order.shape == (n,)
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32)), "every tile exactly once"
    if not xcd:
        return
    assert info[2] == 1
    base, count = info[4:12], info[12:20]
    assert sum(count) == n and base[0] == 0 and all(base[k + 1] == base[k] + count[k] for k in range(7))
    assert info[3] == 8 * max(count)
    # list x holds exactly the tiles of XCD x's screen blocks (16x16 pixels: xcd_block 4)
    bxlog, bylog = max(4 - wl, 0), max(4 - hl, 0)
    x0, y0 = (rect[0], rect[1]) if rect else (0, 0)
    period = (rows >> hl) if frames > 1 else gh + y0
    ty, tx = np.divmod(order.astype(np.int64), gw)
    want = tile_xcd(tx + x0, ty % period + y0, bxlog, bylog)
    for k in range(8):
        assert (want[base[k]:base[k] + count[k]] == k).all(), f"list {k}"
    # frame after frame inside a list, each frame centre first (rings 4 tiles wide)
    frame = ty // period
    tw, th = 1 << wl, 1 << hl
    cx, cy = (W // 2) // tw - x0, (rows // 2) // th - y0
    step = 4 * min(tw, th)
    ring = -(-np.maximum(np.abs(tx - cx) * tw, np.abs(ty % period - cy) * th) // step)
    for k in range(8):
        f = frame[base[k]:base[k] + count[k]]
        r = ring[base[k]:base[k] + count[k]]
        assert (np.diff(f) >= 0).all(), f"list {k}: frames in order"
        assert (np.diff(f * (ring.max() + 1) + r) >= 0).all(), f"list {k}: centre first inside a frame"
