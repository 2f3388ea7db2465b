"""What a short run pays besides its frames: the driver's invocation (bench.py --steps 20 --warmup 5) timed the way bench.py times it --
barrier + synchronize, K steps, flush, barrier + synchronize -- for K = 10 .. 160 and several (frames per launch, launches in flight):
T(K) = a + b K.  b is the frame period; a is what a run pays once (the last launch's tail, launch and synchronisation latency)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from glome_amd import api, dist, scenes
name = os.environ.get("SCENE", "S3")
cfg = scenes.CONFIGS[name]; sd = cfg["make"]()
b = api.Builder(); ctx = api.Context(0)
class Dev:
    def __getattr__(self, n): return getattr(b, n)
    def bih(self, ids): return ctx.bih(b, ids)[0] if len(ids) >= 4096 else b.bih(ids)
nm, _ = sd.replay(Dev()); sc = ctx.commit(b, nm[sd.root])
cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
P = api.render_params(width=cfg["width"], height=cfg["height"], maxdepth=cfg["maxdepth"])
dev = torch.device("cuda", 0)
for group, lanes in [(10, 4), (5, 4), (16, 4), (10, 2), (20, 2), (32, 2)]:
    try:
        sf = dist.ShardedFrame(sc, P, 0, 1, dev, lanes=lanes, product="packed", group=group)
    except Exception as e:
        print(json.dumps({"group": group, "lanes": lanes, "error": str(e)})); continue
    if sf.G != group:
        print(json.dumps({"group": group, "lanes": lanes, "skipped": "a launch carries at most %d frames" % sf.G})); continue
    sf.step(cam, lights, stats=True)
    sf.prime(cam, lights)
    rows = {}
    for K in (10, 20, 40, 80, 160):
        ts = []
        for rep in range(5):
            for _ in range(5):
                sf.step(cam, lights)
            sf.flush(); torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(K):
                sf.step(cam, lights)
            sf.flush(); torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t0) * 1e3)
        rows[K] = float(np.median(ts))
    ks = np.array(sorted(rows)); tv = np.array([rows[k] for k in ks])
    bfit, afit = np.polyfit(ks, tv, 1)
    print(json.dumps({"group": group, "lanes": lanes, "total_ms_by_steps": {int(k): round(v, 3) for k, v in rows.items()}, "ms_per_step_by_steps": {int(k): round(v / k, 4) for k, v in rows.items()},
                      "fit_fixed_ms": round(float(afit), 3), "fit_ms_per_frame": round(float(bfit), 4)}), flush=True)
