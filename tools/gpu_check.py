"""First-contact GPU check (run on the GPU box through gpurun): parity of the C-ABI seams against the oracle on the
benchmark scenes, then timing of the flagship frame.  Prints one line per check; exits non-zero on a parity failure."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from helpers import compare_hits, compare_images, oracle_for, product_camera_lights, random_rays  # noqa: E402

from glome_amd import api, scenes  # noqa: E402

ok = True
ctx = api.Context(0)
print("device:", ctx.device_info(), flush=True)


def check(name, cond, msg):
    global ok
    print(("PASS " if cond else "FAIL ") + name + ": " + msg, flush=True)
    ok = ok and cond


for name, sd, res in [("S1", scenes.s1(nlights=2), (360, 240, 1)), ("S3n32", scenes.s3(32), (320, 180, 1)), ("S3mesh32", scenes.s3(32, as_mesh=True), (320, 180, 1)),
                      ("S4", scenes.s4(), (320, 180, 3))]:
    b = api.Builder()
    nm, mm = sd.replay(b)
    t0 = time.time()
    sc = ctx.commit(b, nm[sd.root])
    info = sc.info()
    print(name, "commit %.3fs" % (time.time() - t0), info, flush=True)
    o, nmap, _ = oracle_for(sd)
    ro, rd = random_rays(50000, 7)
    ref = o.rayint(nmap[sd.root], ro.astype(np.float64), rd.astype(np.float64))
    got = sc.rayint(ro, rd)
    mism, emax, _ = compare_hits(got["t"], ref["t"])
    check(name + " rayint_batch", mism < 1e-3 and emax < 2e-4, f"hit/miss mismatch {mism:.2e}, max rel t err {emax:.2e}")
    tm = np.random.default_rng(3).uniform(1, 30, size=len(ro)).astype(np.float32)
    so = o.shadow(nmap[sd.root], ro.astype(np.float64), rd.astype(np.float64), tm.astype(np.float64))
    sg = sc.shadow(ro, rd, tm)
    check(name + " shadow_batch", np.mean(so != sg) < 1e-3, f"mismatch {np.mean(so != sg):.2e}")
    pts = np.random.default_rng(4).uniform(-6, 6, size=(20000, 3)).astype(np.float32)
    io = o.inside(nmap[sd.root], pts.astype(np.float64))
    ig = sc.inside(pts)
    check(name + " inside_batch", np.mean(io != ig) < 1e-3, f"inside frac {io.mean():.3f} mismatch {np.mean(io != ig):.2e}")
    cam, lights = product_camera_lights(sd)
    w, h, md = res
    for faithful in (0, 1):
        P = api.render_params(width=w, height=h, maxdepth=md, faithful=faithful, count_work=faithful)
        img, packed, st = sc.render(cam, lights, P)
        refimg, refpacked, rc = o.render(w, h, maxdepth=md)
        cmp_ = compare_images(img, refimg)
        rays_ok = (st["rays_primary"], st["rays_shadow"], st["rays_secondary"]) == (rc["rays_primary"], rc["rays_shadow"], rc["rays_secondary"])
        check(f"{name} render faithful={faithful}", cmp_["frac_over"] < 0.02 and cmp_["p999"] < 5e-2,
              f"{cmp_} rays gpu {st['rays_primary']}/{st['rays_shadow']}/{st['rays_secondary']} oracle {rc['rays_primary']}/{rc['rays_shadow']}/{rc['rays_secondary']} "
              f"rays_equal={rays_ok} kernel_ms {st['kernel_ms']:.3f} packed_equal_frac {np.mean(packed == refpacked):.4f}"
              + (f" nodes gpu {st['bih_nodes']} oracle {rc['bih_nodes']} prims gpu {st['prim_tests']} oracle {rc['prim_tests']}" if faithful else ""))
    sc.release()

# flagship frame timing
sd = scenes.s3(224)
b = api.Builder()
t0 = time.time(); nm, mm = sd.replay(b); t1 = time.time()
sc = ctx.commit(b, nm[sd.root]); t2 = time.time()
print("S3 build %.2fs commit %.2fs" % (t1 - t0, t2 - t1), sc.info(), flush=True)
cam, lights = product_camera_lights(sd)
import torch  # device memory only
dev = torch.device("cuda:0")
fb = torch.zeros((1080, 1920, 5), dtype=torch.float32, device=dev)
for faithful in (0, 1):
    P = api.render_params(width=1920, height=1080, maxdepth=1, faithful=faithful, count_work=faithful)
    ms = []
    for it in range(6):
        st = sc.render_dev(cam, lights, P, fb.data_ptr())
        ms.append(st["kernel_ms"])
    rays = st["rays_primary"] + st["rays_shadow"] + st["rays_secondary"]
    print(json.dumps({"scene": "S3", "faithful": faithful, "kernel_ms": ms, "rays": rays, "Mrays_s": rays / (min(ms) * 1e-3) / 1e6, "stats": st}), flush=True)
img = fb.cpu().numpy()
print("S3 frame: hit frac %.3f mean rgb %s" % (float((img[..., 3] > 0).mean()), img[..., :3].mean(axis=(0, 1))), flush=True)
print("ALL PASS" if ok else "SOME FAILED")
sys.exit(0 if ok else 1)
