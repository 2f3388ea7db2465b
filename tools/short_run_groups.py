"""Which frames-per-launch serves a SHORT multi-GPU bench run best: rank 0's side of `K` timed steps (after W warm-up
steps, flush included, collective replaced by a device copy) for each group size.  Not a test."""
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


for world in (8, 4, 2):
    for K, W in ((5, 2), (10, 3), (20, 5), (50, 5), (100, 10)):
        row = {}
        for G in (1, 4, 8, 10, 16):
            best = None
            for rep in range(3):
                sf = dist.ShardedFrame(sc, P, 0, world, dev, lanes=4, product="packed", group=G)
                def fake(payload, gathered, async_op=False):
                    gathered[0].copy_(payload)
                    return _Done()
                sf.plan.gather = fake
                sf.prime(cam, lights)
                for i in range(W):
                    sf.step(cam, lights)
                sf.flush(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(K):
                    sf.step(cam, lights)
                sf.flush(); torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / K * 1e3
                best = dt if best is None or dt < best else best
                sc.lib.glome_ctx_use_slot(ctx.h, None, 0)
            row[G] = round(best, 4)
        print(json.dumps({"world": world, "steps": K, "warmup": W, "ms_per_step_by_group": row}), flush=True)
