"""Multi-GPU frame sharding: pixel-row strips across ranks, RCCL gather over xGMI.

The reference's only parallelism is image-space: OpenMP over rows
(src/render.cpp:277-281) writing one shared Screen. Here every pixel is still
independent; scene and BVH are replicated per GPU, 8-row strips are dealt
round-robin to the ranks (balances an object in the middle of the frame; row
blocks would not), each rank renders its strips into a packed buffer, ONE
exchange step per batch of frames assembles them — a gather to rank 0 of the
part of the strips that can differ from the background (gather_strips_window;
gather_strips moves whole strips, to the root or — all-gather — to every rank)
— and rt_deinterleave_strips / rt_deinterleave_frames writes the reference's
row-flipped frame. The area-light RNG is keyed by the global pixel id, so the
image does not depend on the number of ranks. torch.distributed is plumbing
here (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests of the partition
logic and the one-GPU rehearsals).

Overlap: the exchange of one launch's frames runs on a second HIP stream while
the next launches render, so a sequence of frames costs max(render, exchange)
per frame instead of their sum.
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


def gather_strips(gathered, local, root=None, loopback=False):
    """The ONE exchange step of a frame (or of a batch of frames).
    root = None: every rank's packed strips → every rank (all-gather; every rank ends up with the frame).
    root = r:    → rank r only, as one grouped send / receive per peer (ncclGroupStart … ncclSend / ncclRecv …
                 ncclGroupEnd through torch's batch_isend_irecv): each of the root's xGMI links carries exactly one
                 peer's buffer, W*H*12/world bytes each, instead of the (world-1)/world of the whole frame that a
                 ring all-gather pushes through EVERY link — at the render rates of this path the frame gather is
                 what limits scaling, so the bench uses this form.
    RCCL (backend nccl) takes the flat calls; gloo (CPU tests, one-GPU rehearsals) gets list / staged forms.
    loopback (one-rank group, backend nccl): the root is its own peer — its strips travel through the same grouped
    ncclSend / ncclRecv pair a real peer's would (RCCL accepts a send to self inside a group), so a one-GPU box
    executes the point-to-point branch of the N-rank run."""
    import torch
    import torch.distributed as dist

    gloo = dist.get_backend() == "gloo"
    if root is None:
        if gloo:
            dist.all_gather(list(gathered.unbind(0)), local)
        else:
            dist.all_gather_into_tensor(gathered, local)
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    if world == 1 and loopback and not gloo:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, gathered[root], root),
                                         dist.P2POp(dist.isend, local, root)]):
            w.wait()
        return
    if rank == root:
        gathered[root].copy_(local)
    if world == 1:
        return
    if gloo:  # blocking point-to-point calls, staged through host memory for device tensors
        if rank == root:
            for r in range(world):
                if r != root:
                    buf = torch.empty(local.shape, dtype=local.dtype) if local.is_cuda else gathered[r]
                    dist.recv(buf, src=r)
                    if local.is_cuda:
                        gathered[r].copy_(buf)
        else:
            dist.send(local.cpu() if local.is_cuda else local, dst=root)
        return
    if rank == root:
        ops = [dist.P2POp(dist.irecv, gathered[r], r) for r in range(world) if r != root]
    else:
        ops = [dist.P2POp(dist.isend, local, root)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()  # orders the current stream after the transfer; the host does not wait


def rows_in_window(rows, y0: int, y1: int):
    """[l0, l1): positions in a rank's (ascending) image-row list `rows` of the rows that lie in [y0, y1)."""
    return int(np.searchsorted(rows, y0)), int(np.searchsorted(rows, y1))


def stale_region(dirty, new):
    """Bookkeeping of a gathered buffer that windowed exchanges reuse. dirty / new = (x0, y0, x1, y1, frames): what the
    last exchange wrote into the peers' slots / what the next one will overwrite. Returns the region the root has to
    zero first (None when the new region covers the old one)."""
    if dirty is None:
        return None
    covered = new[0] <= dirty[0] and new[1] <= dirty[1] and dirty[2] <= new[2] and dirty[3] <= new[3] and dirty[4] <= new[4]
    return None if covered else dirty


def gather_strips_window(gathered, local, n, window, rows_of, root, send_buf, recv_buf, loopback=False):
    """gather_strips(root=r) for the first n frames of a batch, moving only the part of every rank's packed strips
    that can differ from the background: window = (x0, y0, x1, y1), columns [x0, x1) of the image rows [y0, y1)
    (rt_frame_rect: everything outside is vec3(0) in every rank's strips, so the root's gathered buffer — kept zero
    there by the caller — already holds it). The root's links carry what the frame's content needs, not W*H*12/world
    bytes each: at 1 - 4 ranks the full strips of a 1080p frame keep ONE xGMI link busy for longer than the frame takes
    to render (12.4 MB per frame and link at 2 ranks).
    local [B, rows, W, 3], gathered [world, B, rows, W, 3]; rows_of[r]: rank r's image rows (strip_rows_of_rank);
    send_buf / recv_buf: flat staging tensors (recv_buf [world, B*rows*W*3] on the root).
    loopback (one-rank nccl group): the root's own piece goes through send_buf → grouped send / receive to itself →
    recv_buf → paste, i.e. the path of a peer's piece (its slot of `gathered` is zeroed first, so the paste is what
    fills it).
    Returns the bytes this rank received (root) or sent (peer)."""
    import torch
    import torch.distributed as dist

    x0, y0, x1, y1 = window
    rank, world = dist.get_rank(), dist.get_world_size()
    gloo = dist.get_backend() == "gloo"
    if world == 1 and loopback and not gloo:
        l0, l1 = rows_in_window(rows_of[root], y0, y1)
        shape = (n, max(l1 - l0, 0), max(x1 - x0, 0), 3)
        cnt = shape[0] * shape[1] * shape[2] * 3
        gathered[root].zero_()
        if cnt == 0:
            return 0
        out = send_buf[:cnt]
        out.view(shape).copy_(local[:n, l0:l1, x0:x1, :])
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, recv_buf[root][:cnt], root),
                                         dist.P2POp(dist.isend, out, root)]):
            w.wait()
        gathered[root, :n, l0:l1, x0:x1, :].copy_(recv_buf[root][:cnt].view(shape))
        return 4 * cnt

    def piece(r):
        l0, l1 = rows_in_window(rows_of[r], y0, y1)
        shape = (n, max(l1 - l0, 0), max(x1 - x0, 0), 3)
        return l0, l1, shape, shape[0] * shape[1] * shape[2] * 3

    if rank == root:
        gathered[root].copy_(local)
        peers = [(r,) + piece(r) for r in range(world) if r != root]
        peers = [q for q in peers if q[4] > 0]
        if gloo:
            for r, l0, l1, shape, cnt in peers:
                buf = torch.empty(cnt, dtype=local.dtype)
                dist.recv(buf, src=r)
                recv_buf[r][:cnt].copy_(buf)
        elif peers:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, recv_buf[r][:cnt], r) for r, _, _, _, cnt in peers]):
                w.wait()
        for r, l0, l1, shape, cnt in peers:
            gathered[r, :n, l0:l1, x0:x1, :].copy_(recv_buf[r][:cnt].view(shape))
        return 4 * sum(q[4] for q in peers)
    l0, l1, shape, cnt = piece(rank)
    if cnt == 0:
        return 0
    out = send_buf[:cnt]
    out.view(shape).copy_(local[:n, l0:l1, x0:x1, :])
    if gloo:
        dist.send(out.cpu() if out.is_cuda else out, dst=root)
    else:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, out, root)]):
            w.wait()
    return 4 * cnt


def checked_exchange(p2p, gathered, local, n, root):
    """Run the point-to-point exchange `p2p()` (it fills `gathered` [world, B, rows, W, 3] from every rank's `local`
    [B, rows, W, 3]) and check it against the all-gather form of the same strips: bitwise over the first n frames, on
    the root; one all-reduce (MIN) makes the verdict every rank's. Returns None when the exchange stands, else the
    reason — and then `gathered` holds the all-gather's result on EVERY rank, so the caller can go on with
    exchange="allgather" from this very batch. An exception out of `p2p` counts as a failed check (a rank stuck
    inside RCCL cannot be helped from here; the process group's timeout ends such a run)."""
    import torch
    import torch.distributed as dist

    why = None
    try:
        p2p()
    except Exception as e:   # noqa: BLE001 — whatever RCCL / torch raises: the all-gather form takes over
        why = f"{type(e).__name__}: {e}"[:200]
    ref = torch.zeros_like(gathered)
    gather_strips(ref, local, None)
    ok = why is None
    if ok and dist.get_rank() == root:
        # outside a window the point-to-point form leaves zeros where the strips ARE zero (rt_frame_rect's promise)
        ok = bool(torch.equal(gathered[:, :n].view(torch.int32), ref[:, :n].view(torch.int32)))
        if not ok:
            why = "the point-to-point gather's strips differ from the all-gather's"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=gathered.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        gathered.copy_(ref)
        return why or "another rank's point-to-point gather failed its check"
    return None


def new_streams(n: int, high_priority: bool = False):
    """n HIP streams created HERE, one after the other (hipStreamCreateWithFlags through the HIP runtime
    torch already loaded), wrapped for torch. The runtime deals its hardware queues round-robin in creation
    order (tools/queue_bench.hip), so consecutive streams of one call run side by side — streams taken from
    torch's pool of 32 may share a queue. At most GPU_MAX_HW_QUEUES (HIP runtime setting, default 4) kernels
    of a process run concurrently; bench.py raises it before the runtime starts."""
    import ctypes

    import torch

    # the copy of the HIP runtime this process already runs on (torch brings its own), never a second one
    loaded = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln]
    if not loaded:
        raise RuntimeError("the HIP runtime is not loaded: no GPU build of torch in this process")
    hip = ctypes.CDLL(loaded[0])
    hip.hipStreamCreateWithPriority.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint, ctypes.c_int]
    least, greatest = ctypes.c_int(), ctypes.c_int()
    if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) != 0:
        raise RuntimeError("hipDeviceGetStreamPriorityRange failed")
    out = []
    for _ in range(n):
        h = ctypes.c_void_p()
        # high priority = numerically lowest; the exchange stream takes it so that its short kernels are
        # dispatched ahead of the render workgroups of the frames in flight
        err = hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, greatest.value if high_priority else 0)  # 1 = hipStreamNonBlocking
        if err != 0:
            raise RuntimeError(f"hipStreamCreateWithPriority failed: {err}")
        out.append(torch.cuda.ExternalStream(h.value))
    return out


class StripRenderer:
    """One rank's share of a frame + the gather. `step()` enqueues a whole frame
    without host synchronisation; `frame` holds the newest assembled frame once the
    device is synchronised (`finish()`)."""

    def __init__(self, rt, bvh, cam, features, params, width, height, rank=0, world=1, strip_rows=8,
                 overlap=True, frames_in_flight=1, exchange_always=False,
                 exchange_batch=1, exchange="allgather", window=True, loopback=False, check_exchange=True):
        """frames_in_flight > 1: successive frames go to alternating HIP streams, each with its own
        buffers and library scratch, so the tail of frame k — a few long path trees on an otherwise
        idle chip — overlaps the start of the next frames (with several ranks: the rank's strips of
        the next frames; the smaller the share, the more frames it takes to fill the chip). Every
        frame is rendered in full; `frame` is the newest one after finish().
        exchange_batch > 1 (several ranks): that many successive frames share ONE all-gather and one
        de-interleave launch (rt_deinterleave_frames). At 8 ranks a share of a frame renders in well
        under 0.1 ms while an exchange — short kernels that must find room on a chip full of render
        waves — takes longer than that; the exchanges of successive frames cannot overlap each other
        (one communicator, one order), so they are made fewer. Every frame is still rendered,
        exchanged and assembled; finish() flushes a partial batch.
        exchange_always: take the strips + exchange path with ONE rank too (the all-gather of a
        one-rank group) — how the GPU tests drive the RCCL branch on a one-GPU box.
        exchange: "allgather" — every rank assembles the frame; "gather" — rank 0 only (gather_strips).
        window (gather form with frames in flight): only the part of the strips inside the frames' rt_frame_rect
        travels (gather_strips_window).
        loopback (one rank, backend nccl, gather form): the root is its own peer, so the grouped ncclSend / ncclRecv
        branch of the N-rank run executes on a one-GPU box.
        check_exchange: the FIRST point-to-point exchange is checked against the all-gather form of the same
        strips (bitwise, on the root; the ranks agree on the verdict with one all-reduce); on a mismatch or an
        exception every rank falls back to exchange="allgather" for the rest of the run (`fallback` says why)."""
        import ctypes

        import torch

        abi = __import__("importlib").import_module("computer-graphics-engine_amd.abi")

        self.rt, self.bvh, self.cam, self.f, self.p = rt, bvh, cam, features, params
        self.W, self.H, self.rank, self.world, self.strip_rows = width, height, rank, world, strip_rows
        self.rows = strip_rows_of_rank(height, strip_rows, rank, world)
        self.frame = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
        # the library memsets and atomically adds sizeof(rt_counters) bytes here: sized from the ABI
        self.counters = torch.zeros(-(-ctypes.sizeof(abi.Counters) // 8), dtype=torch.int64, device="cuda")
        self.sharded = world > 1 or exchange_always
        self.root = 0 if exchange == "gather" else None   # who assembles the frame (None: every rank)
        self.loopback = bool(loopback) and world == 1 and self.root is not None
        self.check_exchange = bool(check_exchange) and self.root is not None and (world > 1 or self.loopback)
        self.fallback = None   # why the point-to-point gather was given up for the all-gather (None: it was not)
        self.overlap = overlap and self.sharded
        self.k = 0         # frames stepped (several ranks: position in the exchange batches)
        self.launches = 0  # launches issued: launch n goes to render stream n % flight
        self.log = None    # bench.py: per launch stream, the frame counts of the launches issued on it
        self.flight = max(1, int(frames_in_flight)) if (not self.sharded or self.overlap) else 1
        if self.flight > 1:
            self.streams = new_streams(self.flight)
            self.started = [False] * self.flight
        if not self.sharded and self.flight > 1:
            self.frames = [self.frame] + [torch.zeros_like(self.frame) for _ in range(self.flight - 1)]
        # renderBloomFilter works on the WHOLE frame (render.cpp:326-328): with several ranks the
        # strips are rendered without it and every rank filters its assembled copy of the frame
        self.bloom = bool(features.enableBloomEffect) and self.sharded
        self.f_strips = features
        if self.bloom:
            self.f_strips = type(features).from_buffer_copy(bytes(features))
            self.f_strips.enableBloomEffect = 0
        self.batch = max(1, int(exchange_batch)) if self.overlap else 1
        if self.sharded:
            rmax = rows_max(height, strip_rows, world)
            B = self.batch
            # batch buffers: those the frames in flight are rendering into + one in exchange + slack
            nbuf = (-(-self.flight // B) + 3) if self.overlap else 1
            self.local = [torch.zeros((B, rmax, width, 3), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
            self.gathered = [torch.zeros((world, B, rmax, width, 3), dtype=torch.float32, device="cuda")
                             for _ in range(nbuf)]
            if self.overlap:
                self.out = [torch.zeros((B, height, width, 3), dtype=torch.float32, device="cuda")
                            for _ in range(nbuf)]
                self.comm_stream = new_streams(1, high_priority=True)[0]
                # rendered[i][b]: frame b of batch buffer i is rendered; consumed[i]: its exchange has read it
                self.rendered = [[torch.cuda.Event() for _ in range(B)] for _ in range(nbuf)]
                self.consumed = [torch.cuda.Event() for _ in range(nbuf)]
                self.used = [False] * nbuf
                # windowed gather: the cameras of each open batch, per-buffer staging, and the region of each
                # gathered buffer that may be non-zero in the peers' slots (None: nothing)
                self.window = bool(window) and self.root is not None and (world > 1 or self.loopback)
                self.batch_cams = [[] for _ in range(nbuf)]
                self.exchanges = {"plain": 0, "window": 0, "bytes": 0}  # bytes: received (root) / sent (peer)
                if self.window:
                    self.rows_of = [strip_rows_of_rank(height, strip_rows, r, world) for r in range(world)]
                    flat = B * rmax * width * 3
                    if rank == self.root:
                        self.recv_buf = [torch.zeros((world, flat), dtype=torch.float32, device="cuda") for _ in range(nbuf)]
                        self.send_buf = [torch.zeros(flat, dtype=torch.float32, device="cuda") if self.loopback else None
                                         for _ in range(nbuf)]
                    else:
                        self.recv_buf = [None] * nbuf
                        self.send_buf = [torch.zeros(flat, dtype=torch.float32, device="cuda") for _ in range(nbuf)]
                    self.dirty = [None] * nbuf

    def _render_stream(self, j):
        """Render stream j of the frames in flight; its first use waits for whatever set the scene up."""
        import torch

        s = self.streams[j]
        if not self.started[j]:
            s.wait_stream(torch.cuda.current_stream())
            self.started[j] = True
        return s

    def _multi_buffers(self, frames):
        """Frame buffers of the launches of `frames` frames, one set per launch in flight (allocated once
        per launch size: prepare() does it before anything is timed)."""
        import torch

        if not hasattr(self, "multi"):
            self.multi = {}
        if frames not in self.multi:
            self.multi[frames] = [torch.zeros((frames, self.H, self.W, 3), dtype=torch.float32, device="cuda")
                                  for _ in range(self.flight)]
        return self.multi[frames]

    def prepare(self, stream: int = 0, launch_sizes=(1,)):
        """Everything a steady-state loop must not pay for: allocate every buffer and run one full launch of
        each size in `launch_sizes` on EVERY launch stream (library scratch per stream, row lists, the first
        launch of each kernel variant), then wait for the device. Call before a timed region, whatever
        warm-up follows."""
        for n in launch_sizes:
            if n < 1:
                continue
            if not self.sharded and n > 1:
                self._multi_buffers(n)
            for _ in range(self.flight):
                self.step(stream, frames=n)
                if self.sharded:
                    self.finish()  # the next launch opens a new exchange batch
            self.finish()

    def step(self, stream: int = 0, frames: int = 1, cams=None):
        """Render this rank's strips of one frame on `stream` (a raw hipStream_t, the
        current torch stream) and exchange them. frames > 1: that many successive frames in ONE
        launch (rt_render_device_frames) — a launch of several frames has no idle tail between them,
        which is what a rank's small share of a frame at 8 ranks needs on top of the frames in flight.
        cams: the frames' camera poses (one per frame; default: the renderer's camera for each).
        Several ranks: the frames of one call must lie in one exchange batch."""
        import torch

        cams = list(cams) if cams is not None else [self.cam] * frames
        if len(cams) != frames:
            raise ValueError("one camera per frame")
        if not self.sharded:
            if frames > 1:
                j = self.launches % self.flight
                self.launches += 1
                self._log(j, frames)
                buf = self._multi_buffers(frames)[j]
                raw = self._render_stream(j).cuda_stream if self.flight > 1 else stream
                self.rt.render_device_frames(self.bvh, cams, self.f, self.p, self.W, self.H,
                                             buf.data_ptr(), stream=raw)
                self.frame = buf[frames - 1]
                return
            if self.flight > 1:
                j = self.launches % self.flight
                self.launches += 1
                self._log(j, 1)
                s = self._render_stream(j)
                self.frame = self.frames[j]
                self.rt.render_device(self.bvh, cams[0], self.f, self.p, self.W, self.H, self.frame.data_ptr(),
                                      stream=s.cuda_stream)
                return
            self._log(0, 1)
            self.rt.render_device(self.bvh, cams[0], self.f, self.p, self.W, self.H, self.frame.data_ptr(),
                                  stream=stream)
            return
        if not self.overlap:
            self._log(0, 1)
            self.rt.render_device(self.bvh, cams[0], self.f_strips, self.p, self.W, self.H, self.local[0].data_ptr(),
                                  rows=self.rows, packed=True, stream=stream)
            gather_strips(self.gathered[0], self.local[0], self.root)
            if self.root is None or self.rank == self.root:
                self.rt.deinterleave_strips(self.gathered[0].data_ptr(), self.frame.data_ptr(), self.W, self.H,
                                            self.strip_rows, self.world, stream=stream)
                if self.bloom:
                    self.rt.bloom_device(self.frame.data_ptr(), self.W, self.H, self.p, stream=stream)
            return
        # frames in flight: frame k renders on stream k % flight, so the tail of one rank's share (a
        # few long path trees) overlaps the shares of the next frames; the exchanges follow in frame
        # order on the one exchange stream (every rank issues them in the same order)
        if frames > self.batch:
            raise ValueError("a launch holds at most one exchange batch of frames")
        q, b = divmod(self.k, self.batch)
        if b + frames > self.batch:  # the launch does not fit the open batch: exchange what it holds, open the next
            self._exchange(q % len(self.local), b)
            self.k += self.batch - b
            q, b = divmod(self.k, self.batch)
        i = q % len(self.local)
        j = self.launches % self.flight
        self.launches += 1
        self._log(j, frames)
        self.k += frames
        cur = self._render_stream(j) if self.flight > 1 else torch.cuda.current_stream()
        raw = cur.cuda_stream if self.flight > 1 else stream
        if self.used[i]:
            cur.wait_event(self.consumed[i])  # the exchange that read this buffer last must be over
        if b == 0:
            self.batch_cams[i] = []
        self.batch_cams[i].extend(cams)
        if frames > 1:
            self.rt.render_device_frames(self.bvh, cams, self.f_strips, self.p, self.W, self.H,
                                         self.local[i][b].data_ptr(), rows=self.rows, packed=True,
                                         frame_stride_floats=self.local[i][b].numel(), stream=raw)
        else:
            self.rt.render_device(self.bvh, cams[0], self.f_strips, self.p, self.W, self.H,
                                  self.local[i][b].data_ptr(), rows=self.rows, packed=True, stream=raw)
        for bb in range(b, b + frames):
            self.rendered[i][bb].record(cur)
        if b + frames == self.batch:
            self._exchange(i, self.batch)

    def _exchange(self, i, n):
        """The exchange of batch buffer i, whose first n frames are rendered (or being rendered)."""
        import torch

        with torch.cuda.stream(self.comm_stream):
            for e in self.rendered[i][:n]:
                self.comm_stream.wait_event(e)
            self._gather(i, n)
            raw = self.comm_stream.cuda_stream
            if self.root is None or self.rank == self.root:
                self.rt.deinterleave_frames(self.gathered[i].data_ptr(), self.out[i].data_ptr(), self.W, self.H,
                                            self.strip_rows, self.world, n, self.batch, stream=raw)
                if self.bloom:
                    for b in range(n):
                        self.rt.bloom_device(self.out[i][b].data_ptr(), self.W, self.H, self.p, stream=raw)
            self.consumed[i].record(self.comm_stream)
        self.used[i] = True
        self.frame = self.out[i][n - 1]

    def _gather(self, i, n):
        """The exchange step proper of batch buffer i (first n frames). The first point-to-point exchange of a run
        is checked against the all-gather form (check_exchange)."""
        if not self.check_exchange:
            self._gather_once(i, n)
            return
        self.check_exchange = False
        why = checked_exchange(lambda: self._gather_once(i, n), self.gathered[i], self.local[i], n, self.root)
        if why is not None:
            self.fallback = why
            self.root, self.window, self.loopback = None, False, False

    def _gather_once(self, i, n):
        """Whole strips, or — gather form — only the window that the frames' cameras can see anything in (every rank
        computes the same window from the same cameras; a window of more than 3/4 of the frame goes the plain way)."""
        if not getattr(self, "window", False):
            gather_strips(self.gathered[i], self.local[i], self.root, loopback=self.loopback)
            return
        W, H = self.W, self.H
        cams = self.batch_cams[i][:n]
        win = self.rt.frame_rect(self.bvh, cams, self.f_strips, self.p, W, H) if len(cams) == n else (0, 0, W, H)
        x0, y0, x1, y1 = win
        plain = (x1 - x0) * (y1 - y0) * 4 > W * H * 3
        if self.rank == self.root:
            g, d = self.gathered[i], self.dirty[i]
            new = (0, 0, W, H, self.batch) if plain else (x0, y0, x1, y1, n)
            d = stale_region(d, new)
            if d is not None:
                g[:, :d[4], :, d[0]:d[2], :].zero_()   # what an earlier, different window left in the peers' slots
            self.dirty[i] = new
        self.exchanges["plain" if plain else "window"] += 1
        if plain:
            gather_strips(self.gathered[i], self.local[i], self.root, loopback=self.loopback)
            self.exchanges["bytes"] += 4 * self.local[i].numel() * (max(self.world - 1, 1) if self.rank == self.root else 1)
        else:
            self.exchanges["bytes"] += gather_strips_window(self.gathered[i], self.local[i], n, win, self.rows_of, self.root,
                                                            self.send_buf[i], self.recv_buf[i], loopback=self.loopback)

    def flush(self):
        """Exchange what the open batch holds now (no host synchronisation); the next frame opens a new batch. A caller
        that wants every launch's frames on their way as soon as they are rendered calls this after step()."""
        if self.sharded and self.overlap and self.k % self.batch:
            n = self.k % self.batch
            self._exchange((self.k // self.batch) % len(self.local), n)
            self.k += self.batch - n

    def finish(self):
        """Flush a partial exchange batch and wait for the device."""
        import torch

        self.flush()
        torch.cuda.synchronize()

    def _log(self, j, frames):
        if self.log is not None:
            self.log[j].append(frames)

    def start_log(self):
        self.log = [[] for _ in range(self.flight)]

    def launch_streams(self, stream: int = 0):
        """raw hipStream_t handles the frames are launched on (bench.py reads the kernel events of each)"""
        return [s.cuda_stream for s in self.streams] if self.flight > 1 else [stream]

    def count_rays(self, stream: int = 0, cam=None) -> dict:
        """One untimed frame of this rank's rows (camera `cam`, default the renderer's) with the counting kernels."""
        import torch

        abi = __import__("importlib").import_module("computer-graphics-engine_amd.abi")
        cam = cam if cam is not None else self.cam
        self.counters.zero_()
        if not self.sharded:
            self.rt.render_device(self.bvh, cam, self.f, self.p, self.W, self.H, self.frame.data_ptr(),
                                  counters_ptr=self.counters.data_ptr(), stream=stream)
        else:
            self.rt.render_device(self.bvh, cam, self.f_strips, self.p, self.W, self.H, self.local[0].data_ptr(),
                                  rows=self.rows, packed=True, counters_ptr=self.counters.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        vals = self.counters.cpu().numpy().view(np.uint64)
        names = [n for n, _ in abi.Counters._fields_]
        return {n: int(vals[i]) for i, n in enumerate(names)}
