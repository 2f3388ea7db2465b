"""How long does ONE render launch take when nothing else runs, by frames per launch and waves per CU?  (The default grid is
sized for several launches in flight; bench.py's roofline times a launch alone.)  SCENE=S3 by default."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from glome_amd import _lib as L, api, scenes
name = os.environ.get("SCENE", "S3")
cfg = scenes.CONFIGS[name]; sd = cfg["make"]()
b = api.Builder(); ctx = api.Context(0)
class Dev:
    def __getattr__(self, n): return getattr(b, n)
    def bih(self, ids): return ctx.bih(b, ids)[0] if len(ids) >= 4096 else b.bih(ids)
nm, _ = sd.replay(Dev()); sc = ctx.commit(b, nm[sd.root])
cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
la = (L.Light * max(1, len(lights)))(*lights)
W, H = cfg["width"], cfg["height"]
P = api.render_params(width=W, height=H, maxdepth=cfg["maxdepth"])
dev = torch.device("cuda:0")
for nf in [int(x) for x in os.environ.get('NF', '1,2,4,8').split(',')]:
    buf = torch.zeros((nf, H, W), dtype=torch.int32, device=dev)
    cams = (L.Camera * nf)(*([cam] * nf))
    for per_cu in [int(x) for x in os.environ.get('PERCU', '0,8,12,16,20,24,32').split(',')]:
        ctx.lib.glome_ctx_set_grid_per_cu(ctx.h, per_cu)
        reps = 10
        for i in range(reps + 2):
            if i == 2: ctx.lib.glome_ctx_timing_begin(ctx.h, reps)
            assert ctx.lib.glome_render_packed_batch_dev(sc.h, cams, nf, la, len(lights), C.byref(P), C.c_void_p(buf.data_ptr()), H * W, None) == 0
            ctx.synchronize()
        ms = np.zeros(reps, np.float32); n = ctx.lib.glome_ctx_timing_end(ctx.h, ms.ctypes.data_as(L.c_fp), reps)
        print(json.dumps({"scene": name, "frames_per_launch": nf, "grid_per_cu": per_cu, "interleave": os.environ.get("GLOME_DEBUG_INTERLEAVE", "default"), "launch_ms_median": round(float(np.median(ms[:n])), 4), "ms_per_frame": round(float(np.median(ms[:n])) / nf, 4)}), flush=True)
