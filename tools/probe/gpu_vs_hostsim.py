"""GPU box: fuzz scenes (tools/probe/fuzz_gpu.py's cameras and light rigs) rendered by the library and by the host build of
the same device headers (tests/hostsim); pixels that differ by more than 1e-4 are listed with both oracles' values.
usage: python tools/probe/gpu_vs_hostsim.py seed [seed ...]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import zoo
from helpers import HostSim, oracle_for, product_camera_lights
from glome_amd import api
ctx = api.Context(0)
W, H = 192, 108
e = lambda a, b: (np.abs(a[..., :4] - b[..., :4]) / np.maximum(1, np.abs(b[..., :4]))).max(-1)
for seed in [int(x) for x in sys.argv[1:]]:
    sd = zoo.random_composites(seed)
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(0, 4))
    if k == 0: sd.set_camera((0.0, 2.0, 12.0), (0.0, 2.0, 0.0), (0, 1, 0), 45.0)
    elif k == 1: sd.set_camera((float(rng.uniform(-3, 3)), float(rng.uniform(0.5, 3)), float(rng.uniform(-3, 3))), (0.0, 1.0, 0.0), (0, 1, 0), 70.0)
    elif k == 2: sd.set_camera((float(rng.uniform(-9, 9)), float(rng.uniform(3, 9)), float(rng.uniform(8, 14))), (0.0, 1.0, 0.0), (0, 1, 0), float(rng.uniform(30, 60)))
    sd.lights = []
    for _ in range(int(rng.integers(1, 5))):
        sd.add_light((float(rng.uniform(-30, 30)), float(rng.uniform(5, 60)), float(rng.uniform(-10, 60))), tuple(float(x) for x in rng.uniform(20, 900, 3)),
                     rad=float(rng.uniform(15, 60)) if rng.uniform() < 0.3 else 1000000.0, shadow=bool(rng.uniform() < 0.8))
    b = api.Builder(); nm, _ = sd.replay(b)
    sc = ctx.commit(b, nm[sd.root]); hs = HostSim(b, nm[sd.root])
    cam, lights = product_camera_lights(sd)
    for md in (1, 2, 3):
        img, _, st = sc.render(cam, lights, api.render_params(width=W, height=H, maxdepth=md))
        him, cnt = hs.render(cam, lights, W, H, md)
        d = e(img, him) > 1e-4
        print(seed, "maxdepth", md, "tier", sc.info()["tier"], "gpu != hostsim pixels", int(d.sum()), "rays gpu", st["rays_primary"], st["rays_shadow"], st["rays_secondary"], "hostsim", [int(x) for x in cnt[:3]], flush=True)
    o, _, _ = oracle_for(sd); ref, _, _ = o.render(W, H, maxdepth=3, want_packed=False)
    ys, xs = np.nonzero(d)
    for (y, x) in list(zip(ys.tolist(), xs.tolist()))[:6]:
        print("   ", (y, x), "gpu", img[y, x], "hostsim", him[y, x], "oracle", ref[y, x, :5])
    # the hit under those pixels: first-hit records of the primary rays
    sc.release()
