"""Time the flagship frame with several builds of the library (glome_amd/variants/libglome_<tag>.so, loaded through
GLOME_DEBUG_LIB), one process per build; prints kernel ms (single frame in flight) and a frame checksum.  Not a test."""
import glob, json, os, subprocess, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, os, json, hashlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
from glome_amd import api, scenes
from helpers import product_camera_lights
out = {"lib": os.path.basename(os.environ.get("GLOME_DEBUG_LIB", "default"))}
for name in os.environ.get("SCENES", "S3").split(","):
    cfg = scenes.CONFIGS[name]
    sd = cfg["make"]()
    b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
    cam, lights = product_camera_lights(sd)
    fb = torch.zeros((cfg["height"], cfg["width"], 5), dtype=torch.float32, device="cuda:0")
    P = api.render_params(width=cfg["width"], height=cfg["height"], maxdepth=cfg["maxdepth"])
    ms = []
    for i in range(25):
        st = sc.render_dev(cam, lights, P, fb.data_ptr()); ms.append(st["kernel_ms"])
    ms = np.array(ms[5:])
    out[name] = {"min_ms": round(float(ms.min()), 4), "med_ms": round(float(np.median(ms)), 4),
                 "sha": hashlib.sha1(fb.cpu().numpy().tobytes()).hexdigest()[:12]}
    del sc, ctx
print(json.dumps(out))
'''
libs = sorted(glob.glob(os.path.join(HERE, "glome_amd", "variants", "libglome_*.so")))
for lib in libs:
    env = dict(os.environ); env["GLOME_DEBUG_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=HERE)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("ERR " + r.stderr[-400:]), flush=True)
