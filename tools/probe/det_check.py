import sys, os, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import product_camera_lights
from glome_amd import api, scenes
sd = scenes.s3(224)
ctx = api.Context(0)
b = api.Builder(); nm, _ = sd.replay(b); sc = ctx.commit(b, nm[sd.root])
cam, lights = product_camera_lights(sd)
P = api.render_params(width=1920, height=1080, maxdepth=1)
a, pa, sa = sc.render(cam, lights, P)
bb, pb, sb = sc.render(cam, lights, P)
f, pf, sf = sc.render(cam, lights, api.render_params(width=1920, height=1080, maxdepth=1, faithful=1))
da, db = (a != f).any(-1), (bb != f).any(-1)
ys, xs = np.nonzero(da | db)
print(os.environ.get("GLOME_DEBUG_LIB", "base"), "a!=f pixels", int(da.sum()), "b!=f pixels", int(db.sum()), "a!=b", int((a != bb).any(-1).sum()),
      "blocks", sorted(set(zip((xs // 8 * 8).tolist(), (ys // 8 * 8).tolist())))[:6], "rays", sa["rays_primary"], sa["rays_shadow"], sf["rays_shadow"], flush=True)
if da.sum():
    y, x = np.argwhere(da)[0]; print(" first bad pixel", x, y, a[y, x], f[y, x])
