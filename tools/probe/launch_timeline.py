"""The timeline of ONE render launch on the clock all CUs share (s_memrealtime, 100 MHz), from a library built with -DGLOME_PROBE
(GLOME_DEBUG_FLAGS=32): when the first and the last wave start, take their last ticket, and end -- where a launch's fixed ~0.3 ms goes."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["GLOME_DEBUG_FLAGS"] = "32"
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
# DCounters::dbg is only ever added to: clear words 8.. by hand through a fresh context per measurement is too slow; instead read
# min / max words, which start at 0 / grow: min words need a large start value, so the first launch is discarded and the probe
# resets through glome_ctx_debug_words' companion below
for nf in (1, 8):
    for per_cu in (8, 24):
        ctx2 = api.Context(0); b2 = api.Builder(); nm2, _ = sd.replay(b2); sc2 = ctx2.commit(b2, nm2[sd.root])
        ctx2.lib.glome_ctx_set_grid_per_cu(ctx2.h, per_cu)
        buf = torch.zeros((nf, H, W), dtype=torch.int32, device=torch.device("cuda:0"))
        cams = (L.Camera * nf)(*([cam] * nf))
        ctx2.lib.glome_ctx_timing_begin(ctx2.h, 1)
        assert ctx2.lib.glome_render_packed_batch_dev(sc2.h, cams, nf, la, len(lights), C.byref(P), C.c_void_p(buf.data_ptr()), H * W, None) == 0
        ctx2.synchronize()
        ms = np.zeros(1, np.float32); ctx2.lib.glome_ctx_timing_end(ctx2.h, ms.ctypes.data_as(L.c_fp), 1)
        w = (C.c_uint64 * 16)(); assert ctx2.lib.glome_ctx_debug_words(ctx2.h, w) == 0
        t0 = w[8]
        us = lambda x: round((x - t0) / 100.0, 1)
        print(json.dumps({"frames": nf, "waves_per_cu": per_cu, "launch_ms_first_launch_of_a_context": round(float(ms[0]), 4), "us_since_first_wave_started": {
            "last_wave_started": us(w[9]), "first_wave_took_its_last_ticket": us(w[10]), "last_wave_took_its_last_ticket": us(w[11]),
            "first_wave_ended": us(w[12]), "last_wave_ended": us(w[13])},
            "cpp_steps_of_all_walks": int(w[14]), "longest_item_us": round((w[15] >> 20) / 100.0, 1), "its_item": int(w[15] & 0xfffff)}), flush=True)
        sc2.release(); ctx2.close()
