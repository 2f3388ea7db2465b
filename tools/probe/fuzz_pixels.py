import os, sys
root="/root/repo"
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import zoo
from helpers import HostSim, oracle_for, product_camera_lights
from glome_amd import api
ctx = api.Context(0)
W, H = 192, 108
e = lambda a, b: (np.abs(a[..., :4] - b[..., :4]) / np.maximum(1, np.abs(b[..., :4]))).max(-1)
for spec in sys.argv[1:]:
    kind, seed = spec.split(":"); seed=int(seed)
    sd = zoo.random_rig(zoo.random_flat(seed) if kind=="flat" else zoo.random_composites(seed), seed)
    cam, lights = product_camera_lights(sd)
    b = api.Builder(); nm,_ = sd.replay(b); sc = ctx.commit(b, nm[sd.root]); hs = HostSim(b, nm[sd.root])
    img = sc.render(cam, lights, api.render_params(width=W, height=H, maxdepth=3))[0]
    him,_ = hs.render(cam, lights, W, H, 3); him=np.asarray(him).reshape(H,W,5)
    o,_,_ = oracle_for(sd); ref,_,_ = o.render(W,H,maxdepth=3,want_packed=False)
    of,_,_ = oracle_for(sd, use_float=True); r32,_,_ = of.render(W,H,maxdepth=3,want_packed=False)
    both = (e(img,ref)>1e-4)&(e(img,r32)>1e-4)
    print(spec, "off both", int(both.sum()), "gpu-host", int((e(img,him)>1e-4).sum()), "host off both", int(((e(him,ref)>1e-4)&(e(him,r32)>1e-4)).sum()), "fp32 oracle vs fp64", int((e(r32,ref)>1e-4).sum()))
    ys,xs=np.nonzero(both)
    for y,x in list(zip(ys.tolist(),xs.tolist()))[:10]:
        print("  ",(y,x),"gpu",np.round(img[y,x],4),"host",np.round(him[y,x],4),"fp64",np.round(ref[y,x,:5],4),"fp32",np.round(r32[y,x,:5],4))
