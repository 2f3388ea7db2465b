"""One-GPU rehearsal of the multi-GPU step in DIRECT mode (round 4: every rank's kernel stores its tiles straight into rank 0's frames;
no payload, no gather, no blit; a fair share of the tiles each): what a rank does per group of frames is one launch over its shard --
timed here for every rank of worlds 1, 2, 4, 8, pipelined like bench.py (lanes launches in flight, G frames each) -- and a one-word
all-reduce.  The step the job sustains is the slowest rank's; scaling = the one-GPU period / that."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, numpy as np
from glome_amd import api, scenes, dist, _lib as L
name = os.environ.get("SCENE", "S3")
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
lanes = int(os.environ.get("LANES", "4"))
streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
base = None
for G in [int(x) for x in os.environ.get("GROUPS", "32 16").split()]:
    frames = [torch.zeros((G, H, W), dtype=torch.int32, device=dev) for _ in range(lanes)]
    cams = (L.Camera * G)(*([cam] * G))
    for world in (1, 2, 4, 8):
        per = []
        for r in (range(world) if world <= 4 else (0, 1, 3, 5, 7)):
            P = api.render_params(width=W, height=H, maxdepth=cfg["maxdepth"], tile_first=r, tile_stride=world, blocksize=64)
            def launch(k):
                sc.lib.glome_ctx_use_slot(ctx.h, C.c_void_p(streams[k % lanes].cuda_stream), k % lanes)
                assert sc.lib.glome_render_packed_batch_dev(sc.h, cams, G, la, len(lights), C.byref(P), C.c_void_p(frames[k % lanes].data_ptr()), H * W, None) == 0
            for k in range(2 * lanes): launch(k)
            torch.cuda.synchronize()
            K = max(8, 256 // G) * (2 if world >= 4 else 1)
            t0 = time.perf_counter()
            for k in range(K): launch(k)
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / (K * G) * 1e3)
        if world == 1: base = per[0] if base is None or G == 32 else base
        one = per[0] if world == 1 else None
        print(json.dumps({"frames_per_launch": G, "launches_in_flight": lanes, "world": world, "ms_per_frame_by_rank": [round(x, 4) for x in per], "slowest": round(max(per), 4),
                          "scaling_vs_one_gpu_same_G": None if world == 1 else round(ref / max(per), 2)} if (world == 1 and not globals().__setitem__("ref", per[0])) or True else {}), flush=True)
sc.lib.glome_ctx_use_slot(ctx.h, None, 0)
