"""One-GPU frame loop through dist.ShardedFrame, repeated: is the frame period stable across repetitions / lanes?  Not a test."""
import json, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from glome_amd import api, scenes, dist
cfg = scenes.CONFIGS["S3"]; sd = cfg["make"]()
b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
dev = torch.device("cuda:0")
P = api.render_params(width=1920, height=1080, maxdepth=1)
for rep in range(3):
    for lanes, group in ((4, 1), (8, 1), (4, 2)):
        sf = dist.ShardedFrame(sc, P, 0, 1, dev, lanes=lanes, product="packed", group=group)
        for i in range(40): sf.step(cam, lights)
        sf.flush(); torch.cuda.synchronize()
        K = 400
        t0 = time.perf_counter()
        for i in range(K): sf.step(cam, lights)
        sf.flush(); torch.cuda.synchronize()
        print(json.dumps({"rep": rep, "lanes": lanes, "group": group, "ms_per_frame": round((time.perf_counter() - t0) / K * 1e3, 4)}), flush=True)
        sc.lib.glome_ctx_use_slot(ctx.h, None, 0)
