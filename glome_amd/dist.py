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


# rank 0's weight in percent of one other rank's (glome_render_params.rank0_share_pct), by ranks.  renderTile mode; where rank
# 0's and another rank's sustained frame periods met in the one-GPU rehearsal (tools/shard_share.py,
# profiles/r02_g_shard_share.log: until round 3 the pattern was quantised to tens, so 2 ranks ran at 90 where 80 is the meeting
# point, and 8 ranks at 70)
RANK0_SHARE_PCT = {1: 0, 2: 80, 3: 75, 4: 70, 5: 70, 6: 70, 7: 70, 8: 70}  # (2, 4, 8 measured; the others in between)


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
        if payload.is_cuda and dist.get_backend() == "gloo":  # bench.py --rehearse: gloo gathers host tensors only
            host = payload.cpu()
            parts = [host.new_empty(host.shape) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, parts, dst=0)
            if self.rank == 0:
                for r in range(self.world):
                    gathered[r].copy_(parts[r])

            class _Done:
                def wait(self):
                    return True
            return _Done() if async_op else None
        return dist.gather(payload, list(gathered.unbind(0)) if self.rank == 0 else None, dst=0, async_op=async_op)


class FramePipeline:
    """Frames are independent, so they are rendered and moved in groups of G, several groups in flight:

        step(view):  buffer the view; with the G-th view of a group, on lane s = group index % n:
                         render_group(s, payload[s], views)    ONE launch renders the rank's tiles of all G frames
                         ONE gather moves the group's payload   (asynchronous)
                     then the oldest pending group is finished: wait for its gather, blit its G frames.

    A rank's share of one frame is a few thousand work items -- not enough to fill the GPU beyond its slowest item --
    and a collective and a launch per frame cost more host time than the frame's GPU work; a group restores long launches
    and amortises both.  A lane is a (HIP stream, context slot) pair: the launch of group k+1 overlaps the tail of group
    k's, and group k's collective (on the backend's own stream) runs under the render of group k+1.  `flush()` completes
    what is in flight (a partial group included).  render_group(slot, payload, views), blit(slot, row, gathered) and
    lane(slot) (a context manager) are callables, so the same control flow is exercised on CPU tensors with gloo
    (tests/test_dist_gloo.py)."""

    def __init__(self, plan, payloads, gathereds, render_group, blit, lane=None, group=1, blit_group=None):
        import contextlib
        self.plan, self.payloads, self.gathereds, self.render_group, self.blit = plan, payloads, gathereds, render_group, blit
        self.blit_group = blit_group  # optional: blit(slot, nrows, gathered) for a whole group at once instead of row by row
        self.lane = lane if lane is not None else (lambda slot: contextlib.nullcontext())
        self.G = int(group)
        self.n = len(payloads)  # lanes = groups in flight
        self.k = 0              # groups launched
        self.views = []
        self.pending = []
        self.done = 0

    def _launch(self):
        slot = self.k % self.n
        views, self.views = self.views, []
        with self.lane(slot):
            self.render_group(slot, self.payloads[slot], views)
            work = self.plan.gather(self.payloads[slot].view(-1), self.gathereds[slot] if self.plan.rank == 0 else None, async_op=True)
        self.pending.append((slot, len(views), work))
        self.k += 1

    def step(self, view=None):
        self.views.append(view)
        if len(self.views) == self.G:
            self._launch()
            while len(self.pending) > max(self.n - 1, 1) or (self.n == 1 and self.pending):
                self._finish(self.pending.pop(0))

    def _finish(self, pending):
        slot, nrows, work = pending
        with self.lane(slot):
            work.wait()  # RCCL: this lane's stream waits for the collective; gloo: the host does
            if self.plan.rank == 0:
                if self.blit_group is not None:
                    self.blit_group(slot, nrows, self.gathereds[slot])
                else:
                    for g in range(nrows):
                        self.blit(slot, g, self.gathereds[slot])
        self.done += nrows

    def flush(self):
        if self.views:  # a partial group
            self._launch()
        while self.pending:
            self._finish(self.pending.pop(0))


class ShardedFrame:
    """Renders frames over all ranks of the default process group; frames land on rank 0.  `lanes` groups of `group`
    frames are in flight (each group on its own HIP stream and context slot): call flush() before reading `frame` (the
    most recently completed one on rank 0).

    product: "packed" -- the frame is GlomeView's framebuffer, one 0x00RRGGBB word per pixel (int32 [h, w]); trace and
    blitTile are fused and 4 bytes per pixel cross xGMI.  "rgbad" -- the float (r, g, b, a, depth) tuples ([h, w, 5]),
    20 bytes per pixel (one frame per launch)."""

    def __init__(self, scene, params, rank, world, device, lanes=4, product="rgbad", group=1, force_pipeline=False, work_tiles=64, rank0_share_pct=None, direct=None):
        """direct: every rank's render kernel stores its tiles' packed pixels straight into rank 0's frames (HIP IPC mappings of rank
        0's framebuffers, glome_ipc_*): no payload, no gather, no blit, a fair share of the tiles for every rank, and one tiny
        all-reduce per group of frames as the completion signal.  None = try it for the packed renderTile product with several ranks
        and fall back to the gather pipeline -- on EVERY rank, agreed by a collective -- when the mapping cannot be made."""
        import torch
        self.torch = torch
        self.direct = False
        import torch.distributed as _td
        # (by default only inside a process group: a caller that plays several ranks in one process -- tests, rehearsals -- has none)
        want_direct = (world > 1 and product == "packed" and params.mode == 0 and _td.is_available() and _td.is_initialized()) if direct is None else bool(direct)
        if want_direct and world > 1 and product == "packed" and params.mode == 0:
            rank0_share_pct = 0 if rank0_share_pct is None else rank0_share_pct
        # rank 0 also receives and blits every frame: it owns less than a fair share of the tiles (glome_render_params.rank0_share_pct;
        # the defaults are where rank 0's and another rank's sustained frame periods met on one GPU, tools/shard_share.py)
        if rank0_share_pct is None:
            rank0_share_pct = RANK0_SHARE_PCT.get(world, 70) if params.mode == 0 else 0
        params = _clone_params(params, rank0_share_pct=int(rank0_share_pct))
        self.rank0_share_pct = int(rank0_share_pct)
        # renderTile's pixels do not depend on the tile map (the adaptive sampler's do, Q21): the shard unit is then a 64x64
        # *work* tile
        # (64: all 8x8 blocks, no thin leftover strips; tools/shard_balance.py: 32 and 16 lose cache locality between a rank's
        # neighbouring work items, 128 and 256 lose balance at 8 ranks) -- whatever tile size the caller's display loop uses
        if work_tiles and params.mode == 0 and params.blocksize != int(work_tiles):
            params = _clone_params(params, blocksize=int(work_tiles))
        self.scene, self.ctx, self.lib = scene, scene.ctx, scene.lib
        if product not in ("packed", "rgbad"):
            raise ValueError("product must be 'packed' or 'rgbad'")
        self.packed = product == "packed"
        self.plan = ShardPlan(params, rank, world, unit=1 if self.packed else 5)
        self.rank, self.world, self.device = rank, world, device
        # force_pipeline: run the payload / gather / blit pipeline even with one rank (a one-rank collective) -- a rehearsal of
        # the multi-GPU path on a single GPU, never the default
        self.piped = world > 1 or bool(force_pipeline)
        self.P, self.P_local = self.plan.P, self.plan.P_local
        self.h, self.w = params.height, params.width
        self.n = max(1, min(int(lanes), 8))  # the context has 8 launch slots
        self.G = max(1, min(int(group), 32)) if self.packed else 1  # frames per launch (and per gather): kMaxBatchFrames
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.n)]
        dt = torch.int32 if self.packed else torch.float32
        shape = (self.G, self.h, self.w) if self.packed else (self.G, self.h, self.w, 5)
        self.frames = None
        if want_direct and world > 1 and self.packed and params.mode == 0:
            self._setup_direct(shape)
        if not self.direct:
            self.frames = [torch.zeros(shape, dtype=dt, device=device) for _ in range(self.n)] if rank == 0 else None
        self.last = (0, 0)
        self.k = 0
        self.batch = []
        self.lights = self.la = None
        if self.piped and not self.direct:
            payloads = [torch.zeros(self.G * self.plan.maxp, dtype=dt, device=device) for _ in range(self.n)]
            gathereds = [torch.zeros((world, self.G * self.plan.maxp), dtype=dt, device=device) if rank == 0 else None for _ in range(self.n)]
            self.pipe = FramePipeline(self.plan, payloads, gathereds, self._render_group, self._blit, self._lane, group=self.G,
                                      blit_group=self._blit_group if self.packed else None)
        torch.cuda.synchronize(device)

    @property
    def frame(self):
        return self.frames[self.last[0]][self.last[1]] if self.frames is not None else None

    # ---- direct mode: rank 0's framebuffers mapped into every rank's address space ----
    def _setup_direct(self, shape):
        """rank 0 allocates one framebuffer [G, h, w] per lane through glome_ipc_alloc and publishes the handles; the others open them.
        Whether that worked is agreed by all ranks (MIN all-reduce) before anybody commits to the mode."""
        import torch.distributed as dist
        torch = self.torch
        nbytes = int(np.prod(shape)) * 4
        ok, self._ipc_ptrs, handles = 1, [], []
        if self.rank == 0:
            try:
                for _ in range(self.n):
                    p, h = C.c_void_p(), C.create_string_buffer(64)
                    if self.lib.glome_ipc_alloc(self.ctx.h, nbytes, C.byref(p), h) != 0:
                        ok = 0
                        break
                    self._ipc_ptrs.append(p.value); handles.append(h.raw)
            except Exception:  # (whatever goes wrong here, every rank must still reach the collectives below)
                ok = 0
        box = [handles if ok else None]
        dist.broadcast_object_list(box, src=0, **({"device": self.device} if dist.get_backend() != "gloo" else {}))
        if self.rank != 0:
            if box[0] is None:
                ok = 0
            else:
                try:
                    for h in box[0]:
                        p = C.c_void_p()
                        if self.lib.glome_ipc_open(self.ctx.h, h, C.byref(p)) != 0:
                            ok = 0
                            break
                        self._ipc_ptrs.append(p.value)
                except Exception:
                    ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device if dist.get_backend() != "gloo" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            self._close_direct()
            self.direct_error = self.ctx.err() if not ok else "another rank could not map the frames"
            return
        self.direct = True
        if self.rank == 0:  # the frames as tensors over the very memory the other ranks store into (no copy: __cuda_array_interface__)
            class _Raw:
                def __init__(s, ptr, shp):
                    s.__cuda_array_interface__ = {"shape": tuple(shp), "typestr": "<i4", "data": (ptr, False), "version": 2}
            self.frames = [torch.as_tensor(_Raw(p, shape), device=self.device) for p in self._ipc_ptrs]
        self._done = [torch.zeros(1, dtype=torch.int32, device=self.device) for _ in range(self.n)]
        self._pending = []

    def _close_direct(self):
        for p in getattr(self, "_ipc_ptrs", []):
            self.lib.glome_ipc_close(self.ctx.h, C.c_void_p(p), 1 if self.rank == 0 else 0)
        self._ipc_ptrs = []

    def close(self):
        """direct mode: unmap / free the shared framebuffers (after flush(); rank 0's `frames` tensors die with them)"""
        if self.direct:
            self.flush()
            self.torch.cuda.synchronize(self.device)
            import torch.distributed as dist
            dist.barrier()
            self.frames = None
            self._close_direct()
            self.direct = False

    def _launch_direct(self):
        """the rank's tiles of the batch, straight into rank 0's frames of this lane; then the group's completion signal: a one-word
        all-reduce ordered behind the launch on every rank -- when it is through on rank 0's lane stream, every rank's stores are in"""
        import torch.distributed as dist
        slot = self.k % self.n
        self.k += 1
        views, self.batch = self.batch, []
        while len(self._pending) >= self.n:  # the lane's last group must be through before its frames are written again
            self._finish_direct(self._pending.pop(0))
        with self._lane(slot):
            rc = self.lib.glome_render_packed_batch_dev(self.scene.h, self._cams(views), len(views), self.la, len(self.lights), C.byref(self.P_local),
                                                        C.c_void_p(self._ipc_ptrs[slot]), self.h * self.w, None)
            if rc != 0:
                raise api.GlomeError("render (direct): " + self.ctx.err())
            if dist.get_backend() == "gloo":  # bench.py --rehearse: the host stands in for the stream-ordered collective
                self.torch.cuda.current_stream().synchronize()
                dist.barrier()
                work = None
            else:
                work = dist.all_reduce(self._done[slot], async_op=True)
        self._pending.append((slot, len(views), work))

    def _finish_direct(self, pending):
        slot, nrows, work = pending
        if work is not None:
            with self._lane(slot):
                work.wait()  # this lane's stream waits for the collective: all ranks' launches of the group are complete
        self.last = (slot, nrows - 1)

    def _lane(self, slot):
        s = self.streams[slot]
        self.lib.glome_ctx_use_slot(self.ctx.h, C.c_void_p(s.cuda_stream), slot)
        return self.torch.cuda.stream(s)

    def _cams(self, views):
        cams = (L.Camera * len(views))()
        for i, v in enumerate(views):
            C.memmove(C.byref(cams[i]), C.byref(v), C.sizeof(L.Camera))
        return cams

    def _render_group(self, slot, payload, views, stats=None):
        st = C.byref(stats) if stats is not None else None
        if self.packed:
            rc = self.lib.glome_render_tiles_packed_batch_dev(self.scene.h, self._cams(views), len(views), self.la, len(self.lights), C.byref(self.P_local),
                                                              C.c_void_p(payload.data_ptr()), self.plan.maxp, st)
        else:
            rc = self.lib.glome_render_tiles_dev(self.scene.h, C.byref(views[0]), self.la, len(self.lights), C.byref(self.P_local), C.c_void_p(payload.data_ptr()), st)
        if rc != 0:
            raise api.GlomeError("render tiles: " + self.ctx.err())

    def _blit(self, slot, g, gathered):
        # rank r's slab starts at r * (G * maxp); frame g of the group sits g * maxp words into every slab
        base = C.c_void_p(gathered.data_ptr() + g * self.plan.maxp * gathered.element_size())
        stride = self.G * self.plan.maxp
        dst = C.c_void_p(self.frames[slot][g].data_ptr())
        if self.packed:
            rc = self.lib.glome_tiles_blit_all_packed_dev(self.ctx.h, C.byref(self.P), self.world, base, stride, dst)
        else:
            rc = self.lib.glome_tiles_blit_all_dev(self.ctx.h, C.byref(self.P), self.world, base, stride, dst, None)
        if rc != 0:
            raise api.GlomeError("blit: " + self.ctx.err())
        self.last = (slot, g)

    def _blit_group(self, slot, nrows, gathered):
        # one launch for the group's frames: frame g is g * maxp words into every rank's slab and lands in frames[slot][g]
        rc = self.lib.glome_tiles_blit_all_packed_batch_dev(self.ctx.h, C.byref(self.P), self.world, C.c_void_p(gathered.data_ptr()), self.G * self.plan.maxp,
                                                            nrows, self.plan.maxp, C.c_void_p(self.frames[slot].data_ptr()), self.h * self.w)
        if rc != 0:
            raise api.GlomeError("blit: " + self.ctx.err())
        self.last = (slot, nrows - 1)

    def set_lights(self, lights):
        self.lights = lights
        self.la = (L.Light * max(1, len(lights)))(*lights)

    def _launch_local(self):
        """world == 1: the batch goes straight into this lane's frames (no payload, no collective)."""
        slot = self.k % self.n
        self.k += 1
        views, self.batch = self.batch, []
        with self._lane(slot):
            fp = self.frames[slot].data_ptr()
            if self.packed:
                rc = self.lib.glome_render_packed_batch_dev(self.scene.h, self._cams(views), len(views), self.la, len(self.lights), C.byref(self.P),
                                                            C.c_void_p(fp), self.h * self.w, None)
            else:
                rc = self.lib.glome_render_dev(self.scene.h, C.byref(views[0]), self.la, len(self.lights), C.byref(self.P), C.c_void_p(fp), None, None)
            if rc != 0:
                raise api.GlomeError("render: " + self.ctx.err())
        self.last = (slot, len(views) - 1)

    def prime(self, cam, lights):
        """Initialisation, not work: one full group through every lane, so that each lane's context slot (counters, queue,
        overflow and tile tables), stream and kernel instances exist before anything is timed.  Without it a run shorter
        than lanes x group frames pays those first uses inside its timed steps."""
        for _ in range(self.n * self.G):
            self.step(cam, lights)
        self.flush()
        self.torch.cuda.synchronize(self.device)

    def step(self, cam, lights, stats=False):
        """One frame (the view `cam`; the lights are shared by the frames of a group).  stats=True renders it alone (no
        overlap) and returns this rank's stats dict (it synchronises)."""
        if lights is not self.lights:
            self.flush()
            self.set_lights(lights)
        if stats:
            self.flush()
            self.torch.cuda.synchronize(self.device)
            st = L.Stats()
            if self.direct:  # this rank's tiles of the one view into frame 0 of lane 0, then everybody waits for everybody
                import torch.distributed as dist
                with self._lane(0):
                    rc = self.lib.glome_render_packed_batch_dev(self.scene.h, self._cams([cam]), 1, self.la, len(self.lights), C.byref(self.P_local),
                                                                C.c_void_p(self._ipc_ptrs[0]), self.h * self.w, C.byref(st))
                if rc != 0:
                    raise api.GlomeError("render (direct): " + self.ctx.err())
                dist.barrier()
                self.last = (0, 0)
                return api._stats_dict(st)
            with self._lane(0):
                if not self.piped:
                    fp = self.frames[0].data_ptr()
                    rc = self.lib.glome_render_dev(self.scene.h, C.byref(cam), self.la, len(self.lights), C.byref(self.P), None if self.packed else C.c_void_p(fp),
                                                   C.c_void_p(fp) if self.packed else None, C.byref(st))
                    if rc != 0:
                        raise api.GlomeError("render: " + self.ctx.err())
                else:
                    self._render_group(0, self.pipe.payloads[0], [cam], st)
                    self.plan.gather(self.pipe.payloads[0].view(-1), self.pipe.gathereds[0] if self.rank == 0 else None)
                    if self.rank == 0:
                        self._blit(0, 0, self.pipe.gathereds[0])
            self.last = (0, 0)
            return api._stats_dict(st)
        if self.direct:
            self.batch.append(cam)
            if len(self.batch) == self.G:
                self._launch_direct()
            return None
        if not self.piped:
            self.batch.append(cam)
            if len(self.batch) == self.G:
                self._launch_local()
            return None
        self.pipe.step(cam)
        return None

    def flush(self):
        if self.direct:
            if self.batch:
                self._launch_direct()
            while self._pending:
                self._finish_direct(self._pending.pop(0))
        elif self.piped:
            self.pipe.flush()
        elif self.batch:
            self._launch_local()
