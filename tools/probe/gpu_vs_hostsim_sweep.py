"""GPU box: many fuzz scenes (both generators, tools/probe/fuzz_gpu.py's rigs) through the library and through the host build
of the same device headers: frames (both render modes) pixel by pixel and random shadow / inside batches.  Prints the scenes
that differ by more than a handful of silhouette pixels.   usage: python tools/probe/gpu_vs_hostsim_sweep.py n_flat n_composites base"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import zoo
from helpers import HostSim, product_camera_lights, random_rays
from glome_amd import api
nf, nc, base = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = api.Context(0)
W, H = 192, 108
e = lambda a, b: (np.abs(a[..., :4] - b[..., :4]) / np.maximum(1, np.abs(b[..., :4]))).max(-1)
worst, bad, skipped, t0 = 0, [], 0, time.time()
for gen, n in ((zoo.random_flat, nf), (zoo.random_composites, nc)):
    for seed in range(base, base + n):
        sd = zoo.random_rig(gen(seed), seed)
        b = api.Builder(); nm, _ = sd.replay(b)
        try:
            sc = ctx.commit(b, nm[sd.root]); hs = HostSim(b, nm[sd.root])
            cam, lights = product_camera_lights(sd)
            img, _, st = sc.render(cam, lights, api.render_params(width=W, height=H, maxdepth=3), want_packed=False)
            him, cnt = hs.render(cam, lights, W, H, 3)
            ro, rd = random_rays(20000, seed)
            sg, sh = sc.shadow(ro, rd, 30.0), hs.shadow(ro, rd, 30.0)
            ig, ih = sc.inside(ro), hs.inside(ro)
        except (api.GlomeError, RuntimeError):
            skipped += 1; continue
        d = int((e(img, him) > 1e-4).sum()); worst = max(worst, d)
        ns, ni = int((sg != sh).sum()), int((ig != ih).sum())
        if d > 16 or ns > 0 or ni > 0: bad.append((gen.__name__, seed, "pixels", d, "shadow rays", ns, "inside", ni))
        sc.release()
print("scenes", nf + nc - skipped, "skipped", skipped, "worst frame differs on", worst, "pixels; beyond a handful:", bad, "secs", round(time.time() - t0, 1), flush=True)
