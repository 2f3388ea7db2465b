"""Timing sweep of the flagship frame under debug knobs (GLOME_DEBUG_LB, GLOME_DEBUG_STACK_CAP).  Not a test."""
import json, os, subprocess, sys
code = r'''
import sys, os, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
from glome_amd import api, scenes
from helpers import product_camera_lights
sd = scenes.CONFIGS[os.environ.get("SCENE", "S3")]["make"]()
b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
cam, lights = product_camera_lights(sd)
cfg = scenes.CONFIGS[os.environ.get("SCENE", "S3")]
fb = torch.zeros((cfg["height"], cfg["width"], 5), dtype=torch.float32, device="cuda:0")
P = api.render_params(width=cfg["width"], height=cfg["height"], maxdepth=cfg["maxdepth"])
ms = []
for i in range(30):
    st = sc.render_dev(cam, lights, P, fb.data_ptr()); ms.append(st["kernel_ms"])
ms = np.array(ms[5:])
print(json.dumps({"lb": os.environ.get("GLOME_DEBUG_LB"), "cap": os.environ.get("GLOME_DEBUG_STACK_CAP"), "min_ms": float(ms.min()), "med_ms": float(np.median(ms)), "hit_frac": float((fb[...,3]>0).float().mean())}))
'''
for lb in ("0", "4", "6", "8"):
    for cap in (None, "8", "16"):
        env = dict(os.environ); env["GLOME_DEBUG_LB"] = lb
        if cap: env["GLOME_DEBUG_STACK_CAP"] = cap
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("ERR " + r.stderr[-300:]), flush=True)
