"""Frames of the adaptive sampler (and plain renderTile) from this build, written as checksums + raw dumps, so two builds of the
library (GLOME_DEBUG_LIB) can be compared bit for bit.  usage: python ss_compare.py OUT.npz"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from glome_amd import api, scenes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import zoo
ctx = api.Context(0)
out = {}
for name, mk, w, h, md in [("S1", lambda: scenes.s1(nlights=1), 720, 480, 1), ("S3", lambda: scenes.s3(224), 1920, 1080, 1), ("S4", scenes.s4, 520, 390, 3),
                           ("materials", zoo.materials, 400, 300, 3), ("mesh", zoo.mesh_scene, 390, 195, 1), ("S3b", lambda: scenes.s3(64), 1001, 707, 1)]:
    sd = mk(); b = api.Builder(); nm, _ = sd.replay(b); sc = ctx.commit(b, nm[sd.root])
    cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
    for rep in range(3):
        img, packed, st = sc.render(cam, lights, api.render_params(width=w, height=h, mode=1, maxdepth=md))
        key = f"{name}_mode1"
        hsh = hashlib.sha256(img.tobytes() + packed.tobytes()).hexdigest()[:16]
        if key in out and out[key] != hsh: print("NONDETERMINISTIC", key, rep)
        out[key] = hsh
    print(name, out[key], st["rays_primary"], st["rays_shadow"], st["rays_secondary"], "kernel_ms", round(st["kernel_ms"], 3), flush=True)
    sc.release()
np.savez(sys.argv[1], **{k: np.array(v) for k, v in out.items()})
