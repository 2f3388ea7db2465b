import json, sys, time, os
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
    def wait(self): return True
def period(world, rank, pct, group, lanes, grid):
    sc.lib.glome_ctx_set_grid_per_cu(ctx.h, grid)
    sf = dist.ShardedFrame(sc, P, rank, world, dev, lanes=lanes, product="packed", group=group, rank0_share_pct=pct)
    def fake(payload, gathered, async_op=False):
        if rank == 0: gathered[0].copy_(payload)
        return _Done()
    sf.plan.gather = fake
    for i in range(64): sf.step(cam, lights)
    sf.flush(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        K = 480; t0 = time.perf_counter()
        for i in range(K): sf.step(cam, lights)
        sf.flush(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
    sc.lib.glome_ctx_use_slot(ctx.h, None, 0)
    return best
for grid in (0, 6, 8, 12, 16, 24):
    for lanes in (3, 4):
        print(json.dumps({"world": 8, "group": 16, "lanes": lanes, "grid_per_cu": grid, "rank0": round(period(8, 0, 70, 16, lanes, grid), 4), "rank1": round(period(8, 1, 70, 16, lanes, grid), 4)}), flush=True)
