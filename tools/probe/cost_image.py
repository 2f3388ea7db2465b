"""Per-item durations of one frame as an image (a -DGLOME_PROBE library, GLOME_DEBUG_FLAGS=64: every pixel receives its work item's
duration): the distribution, and where the slowest items are."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["GLOME_DEBUG_FLAGS"] = "64"
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
ctx.lib.glome_ctx_set_grid_per_cu(ctx.h, int(os.environ.get("PERCU", "24")))
buf = torch.zeros((1, H, W), dtype=torch.int32, device=torch.device("cuda:0"))
cams = (L.Camera * 1)(cam)
for i in range(3):
    assert ctx.lib.glome_render_packed_batch_dev(sc.h, cams, 1, la, len(lights), C.byref(P), C.c_void_p(buf.data_ptr()), H * W, None) == 0
    ctx.synchronize()
img = buf[0].cpu().numpy().astype(np.float64) / 100.0  # microseconds
blk = img[: H // 8 * 8, : W // 8 * 8].reshape(H // 8, 8, W // 8, 8).max(axis=(1, 3))  # one value per 8x8 block
flat = np.sort(blk.ravel())
print(json.dumps({"items": int(flat.size), "mean_us": round(float(flat.mean()), 1), "median_us": round(float(np.median(flat)), 1),
                  "p90": round(float(flat[int(0.9 * flat.size)]), 1), "p99": round(float(flat[int(0.99 * flat.size)]), 1), "p999": round(float(flat[int(0.999 * flat.size)]), 1),
                  "max": round(float(flat[-1]), 1), "sum_ms_over_waves": round(float(flat.sum()) / 1000.0 / 6144, 4)}))
top = np.argsort(blk.ravel())[::-1][:12]
print("slowest blocks (x, y in pixels; us):", [(int(t % (W // 8)) * 8, int(t // (W // 8)) * 8, round(float(blk.ravel()[t]), 1)) for t in top])
rows = blk.mean(axis=1)
print("mean item us by image row band of 64 px:", [round(float(rows[k * 8:(k + 1) * 8].mean()), 1) for k in range(len(rows) // 8)])
