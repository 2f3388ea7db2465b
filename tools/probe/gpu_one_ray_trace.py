import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import zoo
from helpers import random_rays
from glome_amd import api
ctx = api.Context(0)
seed, i = int(sys.argv[1]), int(sys.argv[2])
sd = zoo.random_composites(seed)
b = api.Builder(); nm, _ = sd.replay(b)
sc = ctx.commit(b, nm[sd.root])
ro, rd = random_rays(200000, seed)
print("BEGIN", flush=True)
print(sc.shadow(ro[i:i + 1].copy(), rd[i:i + 1].copy(), 30.0), flush=True)
ctx.synchronize()
sc.release()
