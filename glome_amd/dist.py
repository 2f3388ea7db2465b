"""Multi-GPU frame rendering: one process per GPU, whole reference tiles sharded round-robin, one RCCL gather.

The reference's only parallelism is `runPar $ parMap renderTile blocks` over 65x65 tiles followed by
`forM_ tiles (blitTile surf)` (GlomeView/Glome.hs:379-386).  Here tile k (renderTiles' order) belongs to rank
k mod N; every rank holds a replica of the flattened scene, renders its tiles into a dense payload (the Tile vectors
of Glome.hs:153-154), and the payloads are gathered to rank 0 over xGMI with ONE torch.distributed gather
(backend "nccl" = RCCL).  Rank 0 blits them into the frame (blitTile).  No other communication exists on the path.
PyTorch is used for device buffers and the collective only.
"""
import ctypes as C

import numpy as np

from . import _lib as L
from . import api


def owned_layout(params, first, stride):
    """Tiles owned by (first, stride): int array [n, 5] = x, y, w, h, pixel offset inside the dense payload."""
    lib = L.load()
    n = lib.glome_tiles_layout(C.byref(params), first, stride, None, 0)
    out = np.zeros((max(n, 1), 5), np.int32)
    lib.glome_tiles_layout(C.byref(params), first, stride, out.ctypes.data_as(L.c_ip), n)
    return out[:n]


def payload_floats(params, first, stride):
    return int(L.load().glome_tiles_payload_floats(C.byref(params), first, stride))


def blit_numpy(frame, payload, layout):
    """Host mirror of glome_tiles_blit_dev for CPU tests: scatter a dense payload into frame[h, w, unit]
    (unit = 5 floats per pixel, or 1 packed word per pixel for a frame[h, w] array)."""
    u = 1 if frame.ndim == 2 else frame.shape[2]
    for x, y, w, h, base in layout:
        frame[y:y + h, x:x + w] = payload[base * u:(base + w * h) * u].reshape((h, w) if frame.ndim == 2 else (h, w, u))
    return frame


def pack_numpy(frame, layout):
    u = 1 if frame.ndim == 2 else frame.shape[2]
    out = np.zeros(int(sum(w * h for _, _, w, h, _ in layout)) * u, frame.dtype)
    for x, y, w, h, base in layout:
        out[base * u:(base + w * h) * u] = frame[y:y + h, x:x + w].reshape(-1)
    return out


def _clone_params(p, **kw):
    q = L.RenderParams()
    C.memmove(C.byref(q), C.byref(p), C.sizeof(L.RenderParams))
    for k, v in kw.items():
        setattr(q, k, v)
    return q


class ShardPlan:
    """Who owns which tiles, how big each rank's payload is, and the one exchange step.  Backend agnostic: the GPU
    path (ShardedFrame) and the CPU multi-process test drive the same plan."""

    def __init__(self, params, rank, world, unit=5):
        """unit: payload words per pixel -- 5 (the float (r, g, b, a, depth) tuple) or 1 (the packed display pixel)."""
        self.rank, self.world, self.unit = rank, world, unit
        self.P = _clone_params(params, tile_first=0, tile_stride=1)
        self.P_local = _clone_params(params, tile_first=rank, tile_stride=world)
        self.sizes = [payload_floats(self.P, r, world) // 5 * unit for r in range(world)]
        self.maxp = max(self.sizes + [unit])  # gather needs equal-sized tensors: pad to the largest shard

    def layout(self, r):
        return owned_layout(self.P, r, self.world)

    def gather(self, payload, gathered, async_op=False):
        """The path's only collective: every rank's (padded) tile payload -> rank 0.  `gathered` (rank 0) is one
        [world, maxp] tensor; returns the work handle when async_op."""
        import torch.distributed as dist
        return dist.gather(payload, list(gathered.unbind(0)) if self.rank == 0 else None, dst=0, async_op=async_op)


class FramePipeline:
    """Frames are independent, so several are kept in flight:

        step k:   on lane k % n:  render(k) -> payload[k % n] ; start gather(k) (asynchronous)
                  then finish frame k-1 on its own lane (wait its gather, blit it)

    Each lane is a (HIP stream, context slot) pair: the render kernel of frame k+1 starts filling the GPU while the tail
    of frame k's persistent kernel drains, and the collective of frame k (on the backend's own stream) runs under the
    render of frame k+1.  `flush()` completes the frames still in flight.  render(slot, payload) and blit(slot, gathered)
    are callables and `lane(slot)` a context manager, so the same control flow is exercised on CPU tensors with gloo
    (tests/test_dist_gloo.py)."""

    def __init__(self, plan, payloads, gathereds, render, blit, lane=None):
        import contextlib
        self.plan, self.payloads, self.gathereds, self.render, self.blit = plan, payloads, gathereds, render, blit
        self.lane = lane if lane is not None else (lambda slot: contextlib.nullcontext())
        self.n = len(payloads)
        self.k = 0
        self.pending = []
        self.done = 0

    def step(self):
        slot = self.k % self.n
        with self.lane(slot):
            self.render(slot, self.payloads[slot])
            work = self.plan.gather(self.payloads[slot], self.gathereds[slot] if self.plan.rank == 0 else None, async_op=True)
        self.pending.append((slot, work))
        self.k += 1
        while len(self.pending) > self.n - 1 or len(self.pending) > 1:
            self._finish(self.pending.pop(0))

    def _finish(self, pending):
        slot, work = pending
        with self.lane(slot):
            work.wait()  # RCCL: this lane's stream waits for the collective; gloo: the host does
            if self.plan.rank == 0:
                self.blit(slot, self.gathereds[slot])
        self.done += 1

    def flush(self):
        while self.pending:
            self._finish(self.pending.pop(0))


class ShardedFrame:
    """Renders one frame per step() over all ranks of the default process group; frames land on rank 0.
    Up to `lanes` frames are in flight (each on its own HIP stream and context slot): call flush() before reading
    `frame` (the most recently completed one on rank 0).

    product: "packed" -- the frame is GlomeView's framebuffer, one 0x00RRGGBB word per pixel (int32 [h, w]); trace and
    blitTile are fused and 4 bytes per pixel cross xGMI.  "rgbad" -- the float (r, g, b, a, depth) tuples ([h, w, 5]),
    20 bytes per pixel."""

    def __init__(self, scene, params, rank, world, device, lanes=4, product="rgbad"):
        import torch
        self.torch = torch
        self.scene, self.ctx, self.lib = scene, scene.ctx, scene.lib
        if product not in ("packed", "rgbad"):
            raise ValueError("product must be 'packed' or 'rgbad'")
        self.packed = product == "packed"
        self.plan = ShardPlan(params, rank, world, unit=1 if self.packed else 5)
        self.rank, self.world, self.device = rank, world, device
        self.P, self.P_local = self.plan.P, self.plan.P_local
        h, w = params.height, params.width
        self.n = max(1, min(int(lanes), 4))
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.n)]
        dt = torch.int32 if self.packed else torch.float32
        self.frames = [torch.zeros((h, w) if self.packed else (h, w, 5), dtype=dt, device=device) for _ in range(self.n)] if rank == 0 else None
        self.last = 0
        self.k = 0
        self.cam = self.lights = self.la = None
        if world > 1:
            payloads = [torch.zeros(self.plan.maxp, dtype=dt, device=device) for _ in range(self.n)]
            gathereds = [torch.zeros((world, self.plan.maxp), dtype=dt, device=device) if rank == 0 else None for _ in range(self.n)]
            self.pipe = FramePipeline(self.plan, payloads, gathereds, self._render, self._blit, self._lane)
        torch.cuda.synchronize(device)

    @property
    def frame(self):
        return self.frames[self.last] if self.frames is not None else None

    def _lane(self, slot):
        s = self.streams[slot]
        self.lib.glome_ctx_use_slot(self.ctx.h, C.c_void_p(s.cuda_stream), slot)
        return self.torch.cuda.stream(s)

    def _render(self, slot, payload, stats=None):
        fn = self.lib.glome_render_tiles_packed_dev if self.packed else self.lib.glome_render_tiles_dev
        rc = fn(self.scene.h, C.byref(self.cam), self.la, len(self.lights), C.byref(self.P_local),
                                             C.c_void_p(payload.data_ptr()), C.byref(stats) if stats is not None else None)
        if rc != 0:
            raise api.GlomeError("glome_render_tiles_dev: " + self.ctx.err())

    def _blit(self, slot, gathered):
        if self.packed:
            rc = self.lib.glome_tiles_blit_all_packed_dev(self.ctx.h, C.byref(self.P), self.world, C.c_void_p(gathered.data_ptr()), self.plan.maxp,
                                                          C.c_void_p(self.frames[slot].data_ptr()))
        else:
            rc = self.lib.glome_tiles_blit_all_dev(self.ctx.h, C.byref(self.P), self.world, C.c_void_p(gathered.data_ptr()), self.plan.maxp,
                                                   C.c_void_p(self.frames[slot].data_ptr()), None)
        if rc != 0:
            raise api.GlomeError("glome_tiles_blit_all_dev: " + self.ctx.err())
        self.last = slot

    def set_view(self, cam, lights):
        self.cam, self.lights = cam, lights
        self.la = (L.Light * max(1, len(lights)))(*lights)

    def step(self, cam, lights, stats=False):
        """One frame.  stats=True renders it alone (no overlap) and returns this rank's stats dict (it synchronises)."""
        if cam is not self.cam or lights is not self.lights:
            self.set_view(cam, lights)
        if stats:
            self.flush()
            self.torch.cuda.synchronize(self.device)
        if self.world == 1:
            slot = self.k % self.n
            self.k += 1
            with self._lane(slot):
                fp = self.frames[slot].data_ptr()
                st = self.scene.render_dev(cam, lights, self.P, None if self.packed else fp, fp if self.packed else None, want_stats=stats)
            self.last = slot
            return st
        if stats:
            st = L.Stats()
            with self._lane(0):
                self._render(0, self.pipe.payloads[0], st)
                self.plan.gather(self.pipe.payloads[0], self.pipe.gathereds[0] if self.rank == 0 else None)
                if self.rank == 0:
                    self._blit(0, self.pipe.gathereds[0])
            return api._stats_dict(st)
        self.pipe.step()
        return None

    def flush(self):
        if self.world > 1:
            self.pipe.flush()
