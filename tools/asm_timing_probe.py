"""GLOME_EXP_ASM_TIMING build: scalar-load wait cycles per branch step of the packet walk (s_memtime around the load).  Not a test."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
from glome_amd import api, scenes
name = os.environ.get("SCENE", "S3")
cfg = scenes.CONFIGS[name]; sd = cfg["make"]()
b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
fb = torch.zeros((cfg["height"], cfg["width"], 5), dtype=torch.float32, device="cuda:0")
P = api.render_params(width=cfg["width"], height=cfg["height"], maxdepth=cfg["maxdepth"])
for i in range(4):
    st = sc.render_dev(cam, lights, P, fb.data_ptr())
print("branch steps %d, load-wait cycles %d -> %.0f cycles per step (includes two s_memtime round trips); kernel %.3f ms" % (st["bih_nodes"], st["prim_tests"], st["prim_tests"] / max(1, st["bih_nodes"]), st["kernel_ms"]))
