"""Phase cycle counters of the GLOME_EXP_TIMING build (closest / shadow / whole item, lane 0 of every wave).  Not a test."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
from glome_amd import api, scenes
from helpers import product_camera_lights
name = os.environ.get("SCENE", "S3")
cfg = scenes.CONFIGS[name]; sd = cfg["make"]()
b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
cam, lights = product_camera_lights(sd)
fb = torch.zeros((cfg["height"], cfg["width"], 5), dtype=torch.float32, device="cuda:0")
P = api.render_params(width=cfg["width"], height=cfg["height"], maxdepth=cfg["maxdepth"])
for i in range(5):
    st = sc.render_dev(cam, lights, P, fb.data_ptr())
print({k: st[k] for k in ("bih_nodes", "prim_tests", "mesh_nodes", "kernel_ms")})
items = 32400.0
print("cycles per item: closest %.0f shadow %.0f total %.0f" % (st["bih_nodes"] / items, st["prim_tests"] / items, st["mesh_nodes"] / items))
