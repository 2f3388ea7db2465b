"""What a work item costs OUTSIDE the tree walks: the flagship scene with the camera turned to the sky (every ray misses the tree's
box: ticket, ray generation, root clip, miss shading, pixel store remain) against the real view, one launch of 8 frames alone.
GLOME_DEBUG_FLAGS (render_loop) needs a library built with -DGLOME_PROBE: tools/build_variants.py probe="-DGLOME_PROBE", GLOME_DEBUG_LIB."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from glome_amd import _lib as L, api, scenes
name = os.environ.get("SCENE", "S3")
cfg = scenes.CONFIGS[name]; sd = cfg["make"]()
b = api.Builder(); ctx = api.Context(0)
class Dev:
    def __getattr__(self, n): return getattr(b, n)
    def bih(self, ids): return ctx.bih(b, ids)[0] if len(ids) >= 4096 else b.bih(ids)
nm, _ = sd.replay(Dev()); sc = ctx.commit(b, nm[sd.root])
pos, at, up, fov = sd.cam
lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
la = (L.Light * max(1, len(lights)))(*lights)
W, H = cfg["width"], cfg["height"]
P = api.render_params(width=W, height=H, maxdepth=cfg["maxdepth"])
nf = 8
prev_words = np.zeros(16)
buf = torch.zeros((nf, H, W), dtype=torch.int32, device=torch.device("cuda:0"))
views = {"scene": api.camera(pos, at, up, fov), "sky": api.camera(pos, (pos[0], pos[1] + 10.0, pos[2] + 0.01), (0, 0, 1), fov)}
for per_cu in [int(x) for x in os.environ.get("PERCU", "4,24").split(",")]:
    ctx.lib.glome_ctx_set_grid_per_cu(ctx.h, per_cu)
    for vname, cam in views.items():
        cams = (L.Camera * nf)(*([cam] * nf))
        reps = 10
        for i in range(reps + 2):
            if i == 2: ctx.lib.glome_ctx_timing_begin(ctx.h, reps)
            st = L.Stats() if i == 0 else None
            assert ctx.lib.glome_render_packed_batch_dev(sc.h, cams, nf, la, len(lights), C.byref(P), C.c_void_p(buf.data_ptr()), H * W, C.byref(st) if st else None) == 0
            ctx.synchronize()
            if st: rays = (st.rays_primary, st.rays_shadow)
        ms = np.zeros(reps, np.float32); n = ctx.lib.glome_ctx_timing_end(ctx.h, ms.ctypes.data_as(L.c_fp), reps)
        out = {"view": vname, "waves_per_cu": per_cu, "rays": rays, "ms_per_frame": round(float(np.median(ms[:n])) / nf, 4)}
        if int(os.environ.get("GLOME_DEBUG_FLAGS", "0")) & 16:  # in-kernel stamps: what a wave spends waiting for tickets
            wd = (C.c_uint64 * 16)(); assert ctx.lib.glome_ctx_debug_words(ctx.h, wd) == 0
            d = np.array(list(wd), dtype=np.float64) - prev_words; prev_words = prev_words + d
            out.update({"ticket_wait_cycles_per_take": round(d[0] / max(d[1], 1), 1), "takes_per_wave": round(d[1] / max(d[3], 1), 2),
                        "ticket_wait_share_of_wave_lifetime": round(d[0] / max(d[2], 1), 4), "wave_lifetime_cycles": round(d[2] / max(d[3], 1), 1),
                        "cycles_per_item": {"lookup": round(d[4] / max(d[1], 1)), "raygen": round(d[5] / max(d[1], 1)), "trace": round(d[6] / max(d[1], 1)),
                                            "all": round(d[2] / max(d[1], 1))}})
        print(json.dumps(out), flush=True)
