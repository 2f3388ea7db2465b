"""GPU box: random groves (tools/probe/fuzz_gpu.py's random_grove, cameras and light rigs) -- BIHs of items the interpreter answers in
place.  Per scene: (1) the frame with the interpreter's packet service on and off (GLOME_DEBUG_NO_GENERIC_PACKETS): must be bit-identical,
both render modes; (2) the GPU against the host build of the same device headers (tests/hostsim): pixels more than 1e-4 apart;
(3) that host build against the fp64 checker.  Also takes fuzz seeds of the other two generators: `flat:SEED`, `comp:SEED`.
usage: python tools/probe/grove_soak.py first_seed n_groves [flat:SEED comp:SEED ...]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import zoo
from helpers import HostSim, oracle_for, product_camera_lights
from glome_amd import api
ctx = api.Context(0)
W, H = 192, 108
e = lambda a, b: (np.abs(a[..., :4] - b[..., :4]) / np.maximum(1, np.abs(b[..., :4]))).max(-1)


rig = zoo.random_rig


jobs = [("grove", s) for s in range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2]))] + [tuple(a.split(":")) for a in sys.argv[3:]]
tot = {"scenes": 0, "on_off_differ": 0, "gpu_vs_host_px": 0, "worst_gpu_vs_host": 0, "host_vs_fp64_px": 0}
for kind, seed in jobs:
    seed = int(seed)
    sd = zoo.grove(n=30 + (seed * 37) % 200, seed=seed) if kind == "grove" else (zoo.random_flat(seed) if kind == "flat" else zoo.random_composites(seed))
    rig(sd, seed)
    cam, lights = product_camera_lights(sd)
    frames = []
    for off in (False, True):
        if off: os.environ["GLOME_DEBUG_NO_GENERIC_PACKETS"] = "1"
        try:
            b = api.Builder(); nm, _ = sd.replay(b)
            sc = ctx.commit(b, nm[sd.root])
        finally:
            os.environ.pop("GLOME_DEBUG_NO_GENERIC_PACKETS", None)
        fr = [sc.render(cam, lights, api.render_params(width=W, height=H, maxdepth=3, mode=m))[0].copy() for m in (0, 1)]
        frames.append(fr); tier = sc.info()["tier"]; depth = sc.info()["max_bih_depth"]
        sc.release()
    same = all(np.array_equal(a, b_) for a, b_ in zip(*frames))
    hs = HostSim(b, nm[sd.root])
    him, cnt = hs.render(cam, lights, W, H, 3)
    o, _, _ = oracle_for(sd); ref, _, _ = o.render(W, H, maxdepth=3, want_packed=False)
    gh, hf, gf = int((e(frames[0][0], him) > 1e-4).sum()), int((e(him, ref) > 1e-4).sum()), int((e(frames[0][0], ref) > 1e-4).sum())
    tot["scenes"] += 1; tot["on_off_differ"] += 0 if same else 1; tot["gpu_vs_host_px"] += gh; tot["worst_gpu_vs_host"] = max(tot["worst_gpu_vs_host"], gh); tot["host_vs_fp64_px"] += hf
    print(kind, seed, "tier", tier, "bih depth", depth, "packets on == off", same, "| pixels > 1e-4 apart: gpu-hostsim", gh, "hostsim-fp64", hf, "gpu-fp64", gf, flush=True)
print(tot, "of", W * H, "pixels a frame", flush=True)
