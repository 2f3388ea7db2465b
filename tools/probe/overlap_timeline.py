"""Do two render launches on two streams overlap?  (-DGLOME_PROBE library, GLOME_DEBUG_FLAGS=32: every launch records, on the clock all
CUs share, when its first / last wave started, took its last ticket and ended -- per context slot.)  L1 on slot 0, L2 on slot 1, issued
back to back: if the queue drains properly, L2's waves start while L1's long items are still running."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["GLOME_DEBUG_FLAGS"] = "32"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from glome_amd import _lib as L, api, scenes
cfg = scenes.CONFIGS[os.environ.get("SCENE", "S3")]; sd = cfg["make"]()
b = api.Builder(); ctx = api.Context(0)
class Dev:
    def __getattr__(self, n): return getattr(b, n)
    def bih(self, ids): return ctx.bih(b, ids)[0] if len(ids) >= 4096 else b.bih(ids)
nm, _ = sd.replay(Dev()); sc = ctx.commit(b, nm[sd.root])
cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
la = (L.Light * max(1, len(lights)))(*lights)
W, H = cfg["width"], cfg["height"]
P = api.render_params(width=W, height=H, maxdepth=cfg["maxdepth"])
dev = torch.device("cuda", 0)
NS = int(os.environ.get("NSLOTS", "3"))
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
for nf in (4, 10):
    bufs = [torch.zeros((nf, H, W), dtype=torch.int32, device=dev) for _ in range(NS)]
    cams = (L.Camera * nf)(*([cam] * nf))
    def launch(k):
        ctx.lib.glome_ctx_use_slot(ctx.h, C.c_void_p(streams[k].cuda_stream), k)
        assert ctx.lib.glome_render_packed_batch_dev(sc.h, cams, nf, la, len(lights), C.byref(P), C.c_void_p(bufs[k].data_ptr()), H * W, None) == 0
    for rep in range(3):  # warm
        for k in range(NS): launch(k)
        torch.cuda.synchronize(dev)
    # reset the min / max words of every slot, then the measured round
    for k in range(NS):
        ctx.lib.glome_ctx_use_slot(ctx.h, C.c_void_p(streams[k].cuda_stream), k)
        ctx.lib.glome_ctx_debug_reset(ctx.h)
    torch.cuda.synchronize(dev)
    for k in range(NS): launch(k)
    torch.cuda.synchronize(dev)
    rows = []
    for k in range(NS):
        ctx.lib.glome_ctx_use_slot(ctx.h, C.c_void_p(streams[k].cuda_stream), k)
        w = (C.c_uint64 * 16)(); assert ctx.lib.glome_ctx_debug_words(ctx.h, w) == 0
        rows.append([int(x) for x in w])
    t0 = min(r[8] for r in rows)
    us = lambda x: round((x - t0) / 100.0, 1)
    print(json.dumps({"frames_per_launch": nf, "launches": [{"slot": k, "first_wave_started": us(r[8]), "last_wave_started": us(r[9]), "first_wave_took_last_ticket": us(r[10]),
                                                             "last_wave_took_last_ticket": us(r[11]), "first_wave_ended": us(r[12]), "last_wave_ended": us(r[13])} for k, r in enumerate(rows)]}), flush=True)
