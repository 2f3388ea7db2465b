"""Single-GPU rehearsal of the multi-GPU step: per-rank render time of each tile shard (load balance) and rank 0's
per-step cost without the collective.  Not a test."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
from glome_amd import api, scenes, dist, _lib as L
from helpers import product_camera_lights
sd = scenes.s3(224)
b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
cam, lights = product_camera_lights(sd)
dev = torch.device("cuda:0")
P = api.render_params(width=1920, height=1080, maxdepth=1)
la = (L.Light * len(lights))(*lights)
frame = torch.zeros((1080, 1920, 5), dtype=torch.float32, device=dev)
for world in (() if os.environ.get("SHARD_PIPE_ONLY") else (1, 2, 4, 8)):
    plans = [dist.ShardPlan(P, r, world) for r in range(world)]
    gathered = torch.zeros((world, plans[0].maxp), dtype=torch.float32, device=dev)
    per_rank = []
    for r in range(world):
        ms = []
        for i in range(12):
            st = L.Stats()
            assert sc.lib.glome_render_tiles_dev(sc.h, C.byref(cam), la, len(lights), C.byref(plans[r].P_local), C.c_void_p(gathered[r].data_ptr()), C.byref(st)) == 0
            ms.append(st.kernel_ms)
        per_rank.append(min(ms[2:]))
    # rank 0's steady state without the collective: its shard rendered with `lanes` frames in flight + one blit per frame
    for lanes in (1, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
        pay = [torch.zeros(plans[0].maxp, dtype=torch.float32, device=dev) for _ in range(lanes)]
        K = 300
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            sl = i % lanes
            sc.lib.glome_ctx_use_slot(ctx.h, C.c_void_p(streams[sl].cuda_stream), sl)
            sc.lib.glome_render_tiles_packed_dev(sc.h, C.byref(cam), la, len(lights), C.byref(plans[0].P_local), C.c_void_p(pay[sl].data_ptr()), None)
            if world > 1:
                sc.lib.glome_tiles_blit_all_packed_dev(ctx.h, C.byref(P), world, C.c_void_p(gathered.data_ptr()), plans[0].maxp // 5, C.c_void_p(frame.data_ptr()))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        print(json.dumps({"world": world, "lanes": lanes, "rank0_step_ms_no_collective": round(dt, 4)}), flush=True)
    sc.lib.glome_ctx_use_slot(ctx.h, None, 0)
    print(json.dumps({"world": world, "render_ms_per_rank_alone": [round(x, 4) for x in per_rank], "max_over_mean": round(max(per_rank) / (sum(per_rank) / world), 3)}), flush=True)

# rank 0's whole step through dist.ShardedFrame (lanes, frame groups, packed payload, blit), the collective replaced by a
# device copy: the frame period a rank can sustain before RCCL enters
class _Done:
    def wait(self):
        return True
import os
_cfgs = ((8, 4, 16), (8, 2, 16), (8, 3, 16), (4, 4, 16), (2, 4, 16), (8, 4, 8), (8, 4, 4), (4, 4, 4), (4, 4, 8), (2, 4, 8), (2, 4, 4), (1, 4, 1), (1, 4, 2), (1, 2, 4))
if os.environ.get("SHARD_TIMING_CONFIGS"):  # "world,lanes,group;..."
    _cfgs = tuple(tuple(int(x) for x in c.split(",")) for c in os.environ["SHARD_TIMING_CONFIGS"].split(";"))
for world, lanes, group in _cfgs:
    sf = dist.ShardedFrame(sc, P, 0, world, dev, lanes=lanes, product="packed", group=group)
    def fake(payload, gathered, async_op=False):
        gathered[0].copy_(payload)
        return _Done()
    sf.plan.gather = fake
    for i in range(48):
        sf.step(cam, lights)
    sf.flush(); torch.cuda.synchronize()
    K = 480
    t0 = time.perf_counter()
    for i in range(K):
        sf.step(cam, lights)
    sf.flush(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    print(json.dumps({"world": world, "launches_in_flight": sf.n, "frames_per_launch": sf.G, "rank0_pipeline_ms_per_frame_fake_gather": round(dt, 4)}), flush=True)
    sc.lib.glome_ctx_use_slot(ctx.h, None, 0)
