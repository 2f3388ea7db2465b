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
for group, lanes in [tuple(int(x) for x in g.split('x')) for g in os.environ.get('CONFIGS', '32x4 32x2 32x1 16x4 16x2 20x1 10x4').split()]:
    try:
        sf = dist.ShardedFrame(sc, P, 0, 1, dev, lanes=lanes, product="packed", group=group)
    except Exception as e:
        print(json.dumps({"group": group, "lanes": lanes, "error": str(e)})); continue
    if sf.G != group:
        print(json.dumps({"group": group, "lanes": lanes, "skipped": "a launch carries at most %d frames" % sf.G})); continue
    sf.step(cam, lights, stats=True)
    sf.prime(cam, lights)
    rows = {}
    for K in [int(x) for x in os.environ.get('STEPS', '20 32 64 128 200 256').split()]:
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
