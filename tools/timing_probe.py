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
print({k: st[k] for k in ("bih_nodes", "prim_tests", "mesh_nodes", "kernel_ms", "rays_secondary")})
print("longest item: %d cycles" % st["rays_secondary"])
items = 32400.0
print("cycles per item: closest %.0f shadow %.0f total %.0f" % (st["bih_nodes"] / items, st["prim_tests"] / items, st["mesh_nodes"] / items))

d = fb[..., 4].cpu().numpy()
H, W = d.shape
print("item-cycle map: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (d.mean(), np.percentile(d, 50), np.percentile(d, 90), np.percentile(d, 99), d.max()))
# where are the slow ones: rows / columns of 65-pixel tiles, leftover strips are x % 65 == 64 or y % 65 == 64
ys, xs = np.nonzero(d > np.percentile(d, 99.5))
print("slow pixels: frac on leftover column %.3f, leftover row %.3f" % (np.mean(xs % 65 == 64), np.mean(ys % 65 == 64)))
print("slow pixels y-range", ys.min(), ys.max(), "x-range", xs.min(), xs.max())
hit = fb[..., 3].cpu().numpy() > 0
print("mean cycles: hit pixels %.0f, miss pixels %.0f" % (d[hit].mean(), d[~hit].mean()))
rows = d.reshape(H // 8, 8, W).mean(axis=(1, 2))
print("per 8-row band mean kcycles:", [int(x / 1000) for x in rows[::6]])
