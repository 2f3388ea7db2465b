import ctypes as C, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch, numpy as np
from glome_amd import api, scenes, _lib as L
for name in ("S3","S5"):
    cfg = scenes.CONFIGS[name]; sd = cfg["make"]()
    b = api.Builder(); ctx = api.Context(0)
    class Dev:
        def __getattr__(self, n): return getattr(b, n)
        def bih(self, ids): return ctx.bih(b, ids)[0] if len(ids) >= 4096 else b.bih(ids)
    nm, _ = sd.replay(Dev()); sc = ctx.commit(b, nm[sd.root])
    cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
    la = (L.Light * len(lights))(*lights)
    W, H = cfg["width"], cfg["height"]
    dev = torch.device("cuda:0")
    for G in (1, 2, 4):
        frames = torch.zeros((G, H, W), dtype=torch.int32, device=dev)
        cams = (L.Camera * G)(*([cam] * G))
        P = api.render_params(width=W, height=H, maxdepth=cfg["maxdepth"])
        for g in (0, 4, 6, 8, 10, 12, 16, 24):
            sc.lib.glome_ctx_set_grid_per_cu(ctx.h, g)
            ts=[]
            for rep in range(9):
                torch.cuda.synchronize(); t0=time.perf_counter()
                assert sc.lib.glome_render_packed_batch_dev(sc.h, cams, G, la, len(lights), C.byref(P), C.c_void_p(frames.data_ptr()), H * W, None) == 0
                torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
            print(name, "frames", G, "grid_per_cu", g, "ms", round(sorted(ts[2:])[3],4), flush=True)
    sc.release()
