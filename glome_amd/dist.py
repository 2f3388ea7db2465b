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
    """Host mirror of glome_tiles_blit_dev for CPU tests: scatter a dense payload into frame[h, w, 5]."""
    for x, y, w, h, base in layout:
        frame[y:y + h, x:x + w, :] = payload[base * 5:(base + w * h) * 5].reshape(h, w, 5)
    return frame


def pack_numpy(frame, layout):
    out = np.zeros(int(sum(w * h for _, _, w, h, _ in layout)) * 5, frame.dtype)
    for x, y, w, h, base in layout:
        out[base * 5:(base + w * h) * 5] = frame[y:y + h, x:x + w, :].reshape(-1)
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

    def __init__(self, params, rank, world):
        self.rank, self.world = rank, world
        self.P = _clone_params(params, tile_first=0, tile_stride=1)
        self.P_local = _clone_params(params, tile_first=rank, tile_stride=world)
        self.sizes = [payload_floats(self.P, r, world) for r in range(world)]
        self.maxp = max(self.sizes + [5])  # gather needs equal-sized tensors: pad to the largest shard

    def layout(self, r):
        return owned_layout(self.P, r, self.world)

    def gather(self, payload, gathered):
        """The path's only collective: every rank's (padded) tile payload -> rank 0."""
        import torch.distributed as dist
        dist.gather(payload, gathered if self.rank == 0 else None, dst=0)


class ShardedFrame:
    """Renders one frame per step() over all ranks of the default process group; the frame lands on rank 0."""

    def __init__(self, scene, params, rank, world, device):
        import torch
        self.torch = torch
        self.scene, self.ctx, self.lib = scene, scene.ctx, scene.lib
        self.plan = ShardPlan(params, rank, world)
        self.rank, self.world = rank, world
        self.P, self.P_local = self.plan.P, self.plan.P_local
        h, w = params.height, params.width
        self.frame = torch.zeros((h, w, 5), dtype=torch.float32, device=device) if rank == 0 else None
        if world > 1:
            self.payload = torch.zeros(self.plan.maxp, dtype=torch.float32, device=device)
            self.gathered = [torch.zeros(self.plan.maxp, dtype=torch.float32, device=device) for _ in range(world)] if rank == 0 else None
        # run on torch's current stream: kernels, the collective and the blit are ordered without host syncs
        self.lib.glome_ctx_use_stream(self.ctx.h, C.c_void_p(torch.cuda.current_stream(device).cuda_stream))

    def step(self, cam, lights, stats=False):
        """One frame.  Returns the per-rank stats dict when stats=True (that synchronises)."""
        if self.world == 1:
            return self.scene.render_dev(cam, lights, self.P, self.frame.data_ptr(), None, want_stats=stats)
        la = (L.Light * max(1, len(lights)))(*lights)
        st = L.Stats()
        rc = self.lib.glome_render_tiles_dev(self.scene.h, C.byref(cam), la, len(lights), C.byref(self.P_local),
                                             C.c_void_p(self.payload.data_ptr()), C.byref(st) if stats else None)
        if rc != 0:
            raise api.GlomeError("glome_render_tiles_dev: " + self.ctx.err())
        self.plan.gather(self.payload, self.gathered)  # tile payloads -> rank 0 (RCCL over xGMI)
        if self.rank == 0:
            for r in range(self.world):
                rc = self.lib.glome_tiles_blit_dev(self.ctx.h, C.byref(self.P), r, self.world, C.c_void_p(self.gathered[r].data_ptr()),
                                                   C.c_void_p(self.frame.data_ptr()), None)
                if rc != 0:
                    raise api.GlomeError("glome_tiles_blit_dev: " + self.ctx.err())
        return api._stats_dict(st) if stats else None
