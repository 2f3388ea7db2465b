"""GPU box: random ray batches through the library's rayint / shadow / inside seams vs the host build of the same headers."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import zoo
from helpers import HostSim, random_rays
from glome_amd import api
ctx = api.Context(0)
for seed in [int(x) for x in sys.argv[1:]]:
    sd = zoo.random_composites(seed)
    b = api.Builder(); nm, _ = sd.replay(b)
    sc = ctx.commit(b, nm[sd.root]); hs = HostSim(b, nm[sd.root])
    ro, rd = random_rays(200000, seed)
    g, h = sc.rayint(ro, rd), hs.rayint(ro, rd)
    bad = np.nonzero((g["t"] != h["t"]) | (g["tex"] != h["tex"]).any(-1))[0]
    sg, sh = sc.shadow(ro, rd, 30.0), hs.shadow(ro, rd, 30.0)
    bads = np.nonzero(sg != sh)[0]
    ig, ih = sc.inside(ro), hs.inside(ro)
    print(seed, "rayint differs", len(bad), "shadow differs", len(bads), "inside differs", int((ig != ih).sum()), flush=True)
    for i in bad[:5]: print("   rayint", i, ro[i], rd[i], "gpu t", g["t"][i], g["prim"][i] if "prim" in g else None, "host t", h["t"][i], h["prim"][i] if "prim" in h else None)
    for i in bads[:8]: print("   shadow", i, ro[i].tolist(), rd[i].tolist(), "gpu", sg[i], "host", sh[i], "host rayint t", h["t"][i], "uid", h.get("prim", [None] * len(ro))[i])
    sc.release()
