"""Adaptive sampler: pipelined and lone frame time of a scene for the region shapes in GLOME_DEBUG_SS_REGIONS.
usage: python ss_tune.py SCENE [launches_in_flight] [frames_per_launch]"""
import os, sys, time
import ctypes as C
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
import torch
from glome_amd import api, scenes
from glome_amd import _lib as L
name = sys.argv[1]; lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4; G = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sd, w, h, md = {"S3": (lambda: scenes.s3(224), 1920, 1080, 1), "S5": (lambda: scenes.s3(708), 3840, 2160, 1), "S2": (lambda: scenes.s1(nlights=1), 720, 480, 1)}[name]
sd = sd()
ctx = api.Context(0)
b = api.Builder(); nm, _ = sd.replay(b); sc = ctx.commit(b, nm[sd.root])
cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
la = (L.Light * len(lights))(*lights)
P = api.render_params(width=w, height=h, mode=1, maxdepth=md)
img, packed, st = sc.render(cam, lights, P)
rays = st["rays_primary"] + st["rays_shadow"] + st["rays_secondary"]
out = torch.zeros((lanes, G, h * w), dtype=torch.int32, device="cuda")
streams = [torch.cuda.Stream() for _ in range(lanes)]
cams = (L.Camera * G)(*[cam] * G)
def run(n, lanes_used):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        k = i % lanes_used
        ctx.lib.glome_ctx_use_slot(ctx.h, C.c_void_p(streams[k].cuda_stream), k)
        rc = ctx.lib.glome_render_packed_batch_dev(sc.h, cams, G, la, len(lights), C.byref(P), C.c_void_p(out[k].data_ptr()), h * w, None)
        assert rc == 0, ctx.err()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / (n * G) * 1e3
run(8, lanes)
same = all(np.array_equal(out[k, g].cpu().numpy().view(np.uint32).reshape(h, w), packed) for k in range(lanes) for g in range(G))
pip = min(run(max(8, 80 // G), lanes) for _ in range(3))
lone = min(run(max(4, 20 // G), 1) for _ in range(3))
print(name, os.environ.get("GLOME_DEBUG_SS_REGIONS", "default"), "lanes", lanes, "frames/launch", G, "rays", rays, "pipelined ms/frame", round(pip, 4), "lone ms/frame", round(lone, 4), "frames_equal_single", same, flush=True)
