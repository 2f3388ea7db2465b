"""Minimal workload for rocprofv3 --pmc passes: the launches bench.py times (default: the flagship frame, four frames per
launch, packed framebuffer product), one after the other on one stream (no CPU leg, no torch.distributed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glome_amd import api, scenes, dist
name = os.environ.get("SCENE", "S3")
cfg = scenes.CONFIGS[name]
sd = cfg["make"]()
b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
MODE = int(os.environ.get("MODE", "0"))
P = api.render_params(width=cfg["width"], height=cfg["height"], maxdepth=cfg["maxdepth"], mode=MODE)
G = int(os.environ.get("GROUP", "4"))
ctx.lib.glome_ctx_set_grid_per_cu(ctx.h, int(os.environ.get("GRID_PER_CU", "32")))  # the launch runs alone
sf = dist.ShardedFrame(sc, P, 0, 1, torch.device("cuda:0"), lanes=1, product="packed", group=G)
for i in range(int(os.environ.get("LAUNCHES", "12")) * G):
    sf.step(cam, lights)
sf.flush()
torch.cuda.synchronize()
print("done")
