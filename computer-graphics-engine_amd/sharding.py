"""Multi-GPU frame sharding: pixel-row strips across ranks, RCCL gather over xGMI.

The reference's only parallelism is image-space: OpenMP over rows
(src/render.cpp:277-281) writing one shared Screen. Here every pixel is still
independent; scene and BVH are replicated per GPU, 8-row strips are dealt
round-robin to the ranks (balances an object in the middle of the frame; row
blocks would not), each rank renders its strips into a packed buffer, ONE
exchange step — an all-gather of equal-sized packed buffers — assembles them,
and rt_deinterleave_strips writes the reference's row-flipped frame. The
area-light RNG is keyed by the global pixel id, so the image does not depend on
the number of ranks. torch.distributed is plumbing here (backend "nccl" = RCCL
on ROCm; "gloo" in the CPU tests of the partition logic).

Overlap: the exchange of frame k runs on a second HIP stream while frame k+1
renders (two packed/gathered buffer pairs), so a sequence of frames costs
max(render, exchange) per frame instead of their sum.
"""
from __future__ import annotations

from typing import List

import numpy as np


def strip_rows_of_rank(height: int, strip_rows: int, rank: int, world: int) -> np.ndarray:
    """Image rows (the reference's loop variable y) rendered by `rank`."""
    n_strips = -(-height // strip_rows)
    rows: List[int] = []
    for s in range(rank, n_strips, world):
        rows.extend(range(s * strip_rows, min((s + 1) * strip_rows, height)))
    return np.asarray(rows, np.int32)


def rows_max(height: int, strip_rows: int, world: int) -> int:
    """Packed rows per rank in the gathered buffer (the largest share)."""
    n_strips = -(-height // strip_rows)
    return -(-n_strips // world) * strip_rows


def deinterleave_numpy(gathered: np.ndarray, height: int, strip_rows: int) -> np.ndarray:
    """Host statement of rt_deinterleave_strips (used by the CPU tests):
    gathered[rank, local_row, x, 3] → row-flipped frame[H, W, 3]."""
    world, _, width, _ = gathered.shape
    frame = np.zeros((height, width, 3), gathered.dtype)
    for y in range(height):
        strip = y // strip_rows
        rank = strip % world
        lrow = (strip // world) * strip_rows + (y - strip * strip_rows)
        frame[height - 1 - y] = gathered[rank, lrow]
    return frame


def gather_strips(gathered, local):
    """The ONE exchange step of a frame: every rank's packed strips → every rank.
    RCCL (backend nccl) takes the flat all-gather; gloo (CPU tests) has no
    _allgather_base, so it gets the list form over views of the same buffer."""
    import torch.distributed as dist

    if dist.get_backend() == "gloo":
        dist.all_gather(list(gathered.unbind(0)), local)
    else:
        dist.all_gather_into_tensor(gathered, local)


class StripRenderer:
    """One rank's share of a frame + the gather. `step()` enqueues a whole frame
    without host synchronisation; `frame` holds the newest assembled frame once the
    device is synchronised (`finish()`)."""

    def __init__(self, rt, bvh, cam, features, params, width, height, rank=0, world=1, strip_rows=8,
                 overlap=True, frames_in_flight=1):
        """frames_in_flight > 1 (one rank only): successive frames go to alternating HIP streams, each
        with its own frame buffer and library scratch, so the tail of frame k — a few long path trees
        on an otherwise idle chip — overlaps the start of frame k + 1. Every frame is rendered in
        full; `frame` is the newest one after finish()."""
        import torch

        self.rt, self.bvh, self.cam, self.f, self.p = rt, bvh, cam, features, params
        self.W, self.H, self.rank, self.world, self.strip_rows = width, height, rank, world, strip_rows
        self.rows = strip_rows_of_rank(height, strip_rows, rank, world)
        self.frame = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
        self.counters = torch.zeros(16, dtype=torch.int64, device="cuda")
        self.overlap = overlap and world > 1
        self.k = 0
        self.flight = max(1, int(frames_in_flight)) if world == 1 else 1
        if self.flight > 1:
            self.streams = [torch.cuda.Stream() for _ in range(self.flight)]
            self.frames = [self.frame] + [torch.zeros_like(self.frame) for _ in range(self.flight - 1)]
        # renderBloomFilter works on the WHOLE frame (render.cpp:326-328): with several ranks the
        # strips are rendered without it and every rank filters its assembled copy of the frame
        self.bloom = bool(features.enableBloomEffect) and world > 1
        self.f_strips = features
        if self.bloom:
            self.f_strips = type(features).from_buffer_copy(bytes(features))
            self.f_strips.enableBloomEffect = 0
        if world > 1:
            rmax = rows_max(height, strip_rows, world)
            nbuf = 2 if self.overlap else 1
            self.local = [torch.zeros((rmax, width, 3), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
            self.gathered = [torch.zeros((world, rmax, width, 3), dtype=torch.float32, device="cuda")
                             for _ in range(nbuf)]
            if self.overlap:
                self.comm_stream = torch.cuda.Stream()
                self.rendered = [torch.cuda.Event() for _ in range(nbuf)]   # buffer i holds a finished render
                self.consumed = [torch.cuda.Event() for _ in range(nbuf)]   # buffer i's exchange has read it
                self.used = [False] * nbuf

    def step(self, stream: int = 0):
        """Render this rank's strips of one frame on `stream` (a raw hipStream_t, the
        current torch stream) and exchange them."""
        import torch

        if self.world == 1:
            if self.flight > 1:
                j = self.k % self.flight
                self.k += 1
                s = self.streams[j]
                if self.k <= self.flight:
                    s.wait_stream(torch.cuda.current_stream())  # whatever set the scene up has finished
                self.frame = self.frames[j]
                self.rt.render_device(self.bvh, self.cam, self.f, self.p, self.W, self.H, self.frame.data_ptr(),
                                      stream=s.cuda_stream)
                return
            self.rt.render_device(self.bvh, self.cam, self.f, self.p, self.W, self.H, self.frame.data_ptr(),
                                  stream=stream)
            return
        i = self.k % len(self.local)
        self.k += 1
        cur = torch.cuda.current_stream()
        if self.overlap and self.used[i]:
            cur.wait_event(self.consumed[i])  # the exchange two frames ago must have read this buffer
        self.rt.render_device(self.bvh, self.cam, self.f_strips, self.p, self.W, self.H, self.local[i].data_ptr(),
                              rows=self.rows, packed=True, stream=stream)
        if not self.overlap:
            gather_strips(self.gathered[i], self.local[i])
            self.rt.deinterleave_strips(self.gathered[i].data_ptr(), self.frame.data_ptr(), self.W, self.H,
                                        self.strip_rows, self.world, stream=stream)
            if self.bloom:
                self.rt.bloom_device(self.frame.data_ptr(), self.W, self.H, self.p, stream=stream)
            return
        self.rendered[i].record(cur)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(self.rendered[i])
            gather_strips(self.gathered[i], self.local[i])
            self.rt.deinterleave_strips(self.gathered[i].data_ptr(), self.frame.data_ptr(), self.W, self.H,
                                        self.strip_rows, self.world, stream=self.comm_stream.cuda_stream)
            if self.bloom:
                self.rt.bloom_device(self.frame.data_ptr(), self.W, self.H, self.p, stream=self.comm_stream.cuda_stream)
            self.consumed[i].record(self.comm_stream)
        self.used[i] = True

    def finish(self):
        import torch

        torch.cuda.synchronize()

    def launch_streams(self, stream: int = 0):
        """raw hipStream_t handles the frames are launched on (bench.py reads the kernel events of each)"""
        return [s.cuda_stream for s in self.streams] if self.flight > 1 else [stream]

    def count_rays(self, stream: int = 0) -> dict:
        """One untimed frame of this rank's rows with the counting kernels."""
        import torch

        abi = __import__("importlib").import_module("computer-graphics-engine_amd.abi")
        self.counters.zero_()
        if self.world == 1:
            self.rt.render_device(self.bvh, self.cam, self.f, self.p, self.W, self.H, self.frame.data_ptr(),
                                  counters_ptr=self.counters.data_ptr(), stream=stream)
        else:
            self.rt.render_device(self.bvh, self.cam, self.f_strips, self.p, self.W, self.H, self.local[0].data_ptr(),
                                  rows=self.rows, packed=True, counters_ptr=self.counters.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        vals = self.counters.cpu().numpy().view(np.uint64)
        names = [n for n, _ in abi.Counters._fields_]
        return {n: int(vals[i]) for i, n in enumerate(names)}
