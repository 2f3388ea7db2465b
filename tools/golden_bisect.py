"""Run each golden scene's GPU steps in its own subprocess, stopping at the first failure (debug aid, not a test)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from glome_amd import api
import test_golden as tg
from helpers import product_camera_lights
name, step = sys.argv[1], sys.argv[2]
mg, g = tg.load_gold(name)
sd = mg.SCENES[name]()
b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
print(name, step, sc.info() if hasattr(sc, "info") else "", flush=True)
ro, rd = mg.golden_inputs()
cam, lights = product_camera_lights(sd)
if step == "render":
    img, _, st = sc.render(cam, lights, api.render_params(width=g["image"]["w"], height=g["image"]["h"], maxdepth=g["image"]["maxdepth"]))
    print("render ok", st, flush=True)
elif step == "rayint":
    h = sc.rayint(ro, rd); print("rayint ok", flush=True)
else:
    o = sc.shadow(ro, rd, np.full(len(ro), g["shadow_tmax"], np.float32)); print("shadow ok", flush=True)
ctx.synchronize()
'''
import glob
names = sys.argv[1:] or sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(HERE, "tests", "golden", "*.json")))
for n in names:
    for step in ("render", "rayint", "shadow"):
        r = subprocess.run([sys.executable, "-c", code, n, step], capture_output=True, text=True, cwd=HERE, timeout=120)
        print(n, step, "rc", r.returncode, r.stdout.strip().splitlines()[-1:] , flush=True)
        if r.returncode != 0:
            print(r.stderr[-1500:]); sys.exit(1)
