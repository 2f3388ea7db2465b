"""One-GPU rehearsal of a SHORT multi-GPU run in direct mode: bench.py --steps 20 is ONE launch of twenty frames on every rank (its shard
of the tiles), nothing to pipeline it with -- so what the job's 20 steps take is the slowest rank's lone launch.  Timed here per rank
for worlds 1, 2, 4, 8 and for the same twenty frames cut into 2 or 4 launches in flight (GROUPS).   usage: python tools/probe/short_run_shards.py"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, numpy as np
from glome_amd import api, scenes, _lib as L
cfg = scenes.CONFIGS["S3"]; sd = cfg["make"]()
b = api.Builder(); ctx = api.Context(0)
class Dev:
    def __getattr__(self, n): return getattr(b, n)
    def bih(self, ids): return ctx.bih(b, ids)[0] if len(ids) >= 4096 else b.bih(ids)
nm, _ = sd.replay(Dev()); sc = ctx.commit(b, nm[sd.root])
cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
la = (L.Light * len(lights))(*lights)
W, H = cfg["width"], cfg["height"]
dev = torch.device("cuda:0")
STEPS = int(os.environ.get("STEPS", "20"))
GRID = int(os.environ.get("GRID_PER_CU", "0"))  # glome_ctx_set_grid_per_cu: 0 = the library sizes a launch's grid by its work (tuned for several launches in flight)
sc.lib.glome_ctx_set_grid_per_cu(ctx.h, GRID)
for nl in [int(x) for x in os.environ.get("SPLITS", "1 2 4").split()]:   # launches the 20 frames are cut into (all in flight at once)
    G = STEPS // nl
    streams = [torch.cuda.Stream(device=dev) for _ in range(nl)]
    frames = [torch.zeros((G, H, W), dtype=torch.int32, device=dev) for _ in range(nl)]
    cams = (L.Camera * G)(*([cam] * G))
    ref = None
    for world in (1, 2, 4, 8):
        per = []
        for r in (range(world) if world <= 4 else (0, 1, 3, 5, 7)):
            P = api.render_params(width=W, height=H, maxdepth=cfg["maxdepth"], tile_first=r, tile_stride=world, blocksize=64)
            def run():
                for k in range(nl):
                    sc.lib.glome_ctx_use_slot(ctx.h, C.c_void_p(streams[k].cuda_stream), k)
                    assert sc.lib.glome_render_packed_batch_dev(sc.h, cams, G, la, len(lights), C.byref(P), C.c_void_p(frames[k].data_ptr()), H * W, None) == 0
            ts = []
            for rep in range(7):
                torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            per.append(sorted(ts[2:])[len(ts[2:]) // 2])
        if world == 1: ref = per[0]
        print(json.dumps({"grid_per_cu": GRID, "steps": STEPS, "launches": nl, "frames_per_launch": G, "world": world, "ms_for_the_run_by_rank": [round(x, 3) for x in per], "slowest": round(max(per), 3),
                          "ms_per_step": round(max(per) / STEPS, 4), "scaling": round(ref / max(per), 2)}), flush=True)
sc.lib.glome_ctx_use_slot(ctx.h, None, 0)
