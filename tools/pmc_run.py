"""Minimal workload for rocprofv3 --pmc passes: N launches of the flagship frame (no CPU leg, no torch.distributed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from glome_amd import api, scenes
name = os.environ.get("SCENE", "S3")
cfg = scenes.CONFIGS[name]
sd = cfg["make"]()
b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
fb = torch.zeros((cfg["height"], cfg["width"], 5), dtype=torch.float32, device="cuda:0")
P = api.render_params(width=cfg["width"], height=cfg["height"], maxdepth=cfg["maxdepth"])
for i in range(int(os.environ.get("LAUNCHES", "12"))):
    sc.render_dev(cam, lights, P, fb.data_ptr(), want_stats=False)
ctx.synchronize()
print("done")
