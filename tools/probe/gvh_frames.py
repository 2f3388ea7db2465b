import os, sys
root = "/root/repo"
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import zoo
from helpers import HostSim, product_camera_lights
from glome_amd import api
W, H = 192, 108
e = lambda a, b: (np.abs(a[..., :4] - b[..., :4]) / np.maximum(1, np.abs(b[..., :4]))).max(-1)
ctx = api.Context(0)
for seed in [int(x) for x in sys.argv[1:]]:
    sd = zoo.random_composites(seed)
    b = api.Builder(); nm, _ = sd.replay(b)
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(0, 4))
    if k == 0: sd.set_camera((0.0, 2.0, 12.0), (0.0, 2.0, 0.0), (0, 1, 0), 45.0)
    elif k == 1: sd.set_camera((float(rng.uniform(-3, 3)), float(rng.uniform(0.5, 3)), float(rng.uniform(-3, 3))), (0.0, 1.0, 0.0), (0, 1, 0), 70.0)
    elif k == 2: sd.set_camera((float(rng.uniform(-9, 9)), float(rng.uniform(3, 9)), float(rng.uniform(8, 14))), (0.0, 1.0, 0.0), (0, 1, 0), float(rng.uniform(30, 60)))
    sd.lights = []
    for _ in range(int(rng.integers(1, 5))):
        sd.add_light((float(rng.uniform(-30, 30)), float(rng.uniform(5, 60)), float(rng.uniform(-10, 60))), tuple(float(x) for x in rng.uniform(20, 900, 3)),
                     rad=float(rng.uniform(15, 60)) if rng.uniform() < 0.3 else 1000000.0, shadow=bool(rng.uniform() < 0.8))
    cam, lights = product_camera_lights(sd)
    sc = ctx.commit(b, nm[sd.root])
    img, _, st = sc.render(cam, lights, api.render_params(width=W, height=H, maxdepth=3))
    np.save("gpurun_out/gvh_%s_%d.npy" % (os.environ.get("GVH_TAG", "cur"), seed), img)
    print(os.environ.get("GVH_TAG", "cur"), seed, "tier", sc.info()["tier"], flush=True)
