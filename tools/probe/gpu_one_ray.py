import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import zoo
from helpers import HostSim, random_rays
from glome_amd import api
ctx = api.Context(0)
seed = 5059
sd = zoo.random_composites(seed)
b = api.Builder(); nm, _ = sd.replay(b)
sc = ctx.commit(b, nm[sd.root]); hs = HostSim(b, nm[sd.root])
ro, rd = random_rays(200000, seed)
sg, sh = sc.shadow(ro, rd, 30.0), hs.shadow(ro, rd, 30.0)
bads = np.nonzero(sg != sh)[0]
print("batch: differs", bads.tolist())
for i in bads[:6]:
    o1, d1 = ro[i:i + 1].copy(), rd[i:i + 1].copy()
    one = sc.shadow(o1, d1, 30.0)[0]
    rep = sc.shadow(np.repeat(o1, 64, 0), np.repeat(d1, 64, 0), 30.0)
    lo = (i // 64) * 64
    wave = sc.shadow(ro[lo:lo + 64].copy(), rd[lo:lo + 64].copy(), 30.0)
    print(i, "host", sh[i], "gpu alone", one, "x64", rep.sum(), "its wave of 64 alone", wave[i - lo], "lane", i - lo)
print(sc.info())
sc.release()
