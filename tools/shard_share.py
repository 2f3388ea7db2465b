"""Single-GPU rehearsal of the weighted shard: rank 0's and rank 1's sustained frame period through dist.ShardedFrame (frame
groups, packed payload; the collective replaced by a device copy on rank 0 and by nothing on rank 1) for several values of
glome_render_params.rank0_share_pct.  The share at which the two meet is the default in dist.RANK0_SHARE_PCT.  Not a test."""
import json, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from glome_amd import api, scenes, dist
from helpers import product_camera_lights
sd = scenes.s3(224)
b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
cam, lights = product_camera_lights(sd)
dev = torch.device("cuda:0")
P = api.render_params(width=1920, height=1080, maxdepth=1)
class _Done:
    def wait(self):
        return True
def period(world, rank, pct, group=8, lanes=4):
    sf = dist.ShardedFrame(sc, P, rank, world, dev, lanes=lanes, product="packed", group=group, rank0_share_pct=pct)
    def fake(payload, gathered, async_op=False):
        if rank == 0:
            gathered[0].copy_(payload)
        return _Done()
    sf.plan.gather = fake
    for i in range(64):
        sf.step(cam, lights)
    sf.flush(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        K = 480
        t0 = time.perf_counter()
        for i in range(K):
            sf.step(cam, lights)
        sf.flush(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
    sc.lib.glome_ctx_use_slot(ctx.h, None, 0)
    return best, sf.plan.sizes
import os
GROUP, LANES = int(os.environ.get("GROUP", "8")), int(os.environ.get("LANES", "4"))  # (bench.py at 8 ranks, long runs: 16 frames per launch on 3 lanes)
WORLDS = [int(x) for x in os.environ.get("WORLDS", "8 4 2").split()]
for world, pcts in ((8, (100, 80, 70, 60, 50, 40, 30)), (4, (100, 90, 80, 70, 60, 50)), (2, (100, 95, 90, 85, 80, 75))):
    if world not in WORLDS:
        continue
    for pct in pcts:
        r0, sizes = period(world, 0, pct, GROUP, LANES)
        r1, _ = period(world, 1, pct, GROUP, LANES)
        print(json.dumps({"world": world, "frames_per_launch": GROUP, "lanes": LANES, "rank0_share_pct": pct, "rank0_ms_per_frame": round(r0, 4), "rank1_ms_per_frame": round(r1, 4),
                          "pixels_rank0": sizes[0], "pixels_rank1": sizes[1]}), flush=True)
