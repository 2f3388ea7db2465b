"""Single-GPU rehearsal of every rank's side of the multi-GPU step: the frame period rank r sustains on its own shard
(launch groups in flight, packed payload, the collective replaced by a device copy; rank 0 also blits), for each work-tile
size.  The step a real run sustains is the slowest rank's.  (One measurement in eight or so comes out ~40 % high whatever the
rank -- an artefact of rebuilding the pipeline objects in one process; repeat before reading imbalance into it.)  Not a test."""
import json, os, sys, time
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


for wt in [int(x) for x in os.environ.get("WORK_TILES", "64,128").split(",")]:
    for world in (2, 4, 8):
        per = []
        for r in range(world):
            sf = dist.ShardedFrame(sc, P, r, world, dev, lanes=4, product="packed", group=8, work_tiles=wt)
            def fake(payload, gathered, async_op=False):
                if gathered is not None:
                    gathered[0].copy_(payload)
                return _Done()
            sf.plan.gather = fake
            for i in range(32):
                sf.step(cam, lights)
            sf.flush(); torch.cuda.synchronize()
            K = 320
            t0 = time.perf_counter()
            for i in range(K):
                sf.step(cam, lights)
            sf.flush(); torch.cuda.synchronize()
            per.append(round((time.perf_counter() - t0) / K * 1e3, 4))
            sc.lib.glome_ctx_use_slot(ctx.h, None, 0)
        print(json.dumps({"work_tile": wt, "world": world, "ms_per_frame_by_rank": per, "slowest": max(per), "mean": round(sum(per) / world, 4)}), flush=True)
