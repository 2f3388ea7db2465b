"""GPU box: one fuzz scene of zoo.random_composites (fuzz_gpu.py's rig) with the interpreter's packet service on, off altogether
(GLOME_DEBUG_NO_GENERIC_PACKETS) and off for trees of items answered in place only (GLOME_DEBUG_NO_ITEM_PACKETS): the pixels that differ,
with the host build's and the fp64 checker's values.   usage: python tools/probe/packet_onoff_seed.py SEED [SEED ...]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, os.path.join(root, "tools", "probe"))
import numpy as np
import zoo
from helpers import HostSim, oracle_for, product_camera_lights
from glome_amd import api
ctx = api.Context(0)
W, H = 192, 108


rig = zoo.random_rig


for seed in [int(x) for x in sys.argv[1:]]:
    sd = zoo.random_composites(seed); rig(sd, seed)
    cam, lights = product_camera_lights(sd)
    out = {}
    for name, env in (("on", {}), ("off", {"GLOME_DEBUG_NO_GENERIC_PACKETS": "1"}), ("items_off", {"GLOME_DEBUG_NO_ITEM_PACKETS": "1"})):
        os.environ.update(env)
        try:
            b = api.Builder(); nm, _ = sd.replay(b); sc = ctx.commit(b, nm[sd.root])
        finally:
            for k in env: os.environ.pop(k)
        out[name] = [sc.render(cam, lights, api.render_params(width=W, height=H, maxdepth=md, mode=m))[0].copy() for md, m in ((1, 0), (3, 0), (3, 1))]
        info = sc.info(); sc.release()
    hs = HostSim(b, nm[sd.root]); him, _ = hs.render(cam, lights, W, H, 3); him = np.asarray(him).reshape(H, W, 5)
    o, _, _ = oracle_for(sd); ref, _, _ = o.render(W, H, maxdepth=3, want_packed=False)
    for k, what in enumerate(("maxdepth 1", "maxdepth 3", "adaptive")):
        print(seed, what, "on != off:", int((out["on"][k] != out["off"][k]).any(-1).sum()), " on != items_off:", int((out["on"][k] != out["items_off"][k]).any(-1).sum()),
              " off != items_off:", int((out["off"][k] != out["items_off"][k]).any(-1).sum()), flush=True)
    ys, xs = np.nonzero((out["on"][1] != out["off"][1]).any(-1))
    for y, x in list(zip(ys.tolist(), xs.tolist()))[:12]:
        print("  ", (y, x), "on", out["on"][1][y, x], "off", out["off"][1][y, x], "items_off", out["items_off"][1][y, x], "hostsim", him[y, x], "fp64", ref[y, x, :5])
