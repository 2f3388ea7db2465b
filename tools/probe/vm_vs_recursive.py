"""CPU regression check of the generic tier's explicit-frame loop (rt_generic.hpp) against the recursive interpreter it
replaced: both compiled for the host (tests/hostsim), random composite scenes with random cameras and light rigs; frames,
work counters, rayint / shadow / inside batches must be bit-identical.  The recursive build comes from the history:
    rm -rf /tmp/legacy && mkdir -p /tmp/legacy && git archive 58d9928 glome_amd/csrc tests/hostsim include | tar -x -C /tmp/legacy && make -C /tmp/legacy/tests/hostsim
usage: python tools/probe/vm_vs_recursive.py [seed ...]     (default: seeds 3000..3259; 260 scenes, 0 different)"""
import sys, os, ctypes as C
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
import numpy as np
import helpers, zoo
from helpers import HostSim, product_camera_lights, random_rays
from glome_amd import api
new_lib = helpers.hostsim_lib()
old_lib = C.CDLL("/tmp/legacy/tests/hostsim/libhostsim.so"); old_lib.hostsim_commit.restype = C.c_void_p
def mk(lib, b, root):
    helpers.hostsim_lib = lambda: lib
    return HostSim(b, root)
seeds = [int(x) for x in sys.argv[1:]] or list(range(3000, 3260))
nbad = 0
for seed in seeds:
    sd = zoo.random_composites(seed)
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(0, 4))
    if k == 0: sd.set_camera((0.0, 2.0, 12.0), (0.0, 2.0, 0.0), (0, 1, 0), 45.0)
    elif k == 1: sd.set_camera((float(rng.uniform(-3, 3)), float(rng.uniform(0.5, 3)), float(rng.uniform(-3, 3))), (0.0, 1.0, 0.0), (0, 1, 0), 70.0)
    elif k == 2: sd.set_camera((float(rng.uniform(-9, 9)), float(rng.uniform(3, 9)), float(rng.uniform(8, 14))), (0.0, 1.0, 0.0), (0, 1, 0), float(rng.uniform(30, 60)))
    sd.lights = []
    for _ in range(int(rng.integers(1, 5))):
        sd.add_light((float(rng.uniform(-30, 30)), float(rng.uniform(5, 60)), float(rng.uniform(-10, 60))), tuple(float(x) for x in rng.uniform(20, 900, 3)),
                     rad=float(rng.uniform(15, 60)) if rng.uniform() < 0.3 else 1000000.0, shadow=bool(rng.uniform() < 0.8))
    b = api.Builder(); nm, _ = sd.replay(b)
    try:
        hn = mk(new_lib, b, nm[sd.root]); ho = mk(old_lib, b, nm[sd.root])
    except Exception as e:
        continue
    cam, lights = product_camera_lights(sd)
    try:
        a, ca = hn.render(cam, lights, 96, 54, 3); o, co = ho.render(cam, lights, 96, 54, 3)
    except Exception as e:
        print(seed, "render error", e); continue
    same = np.array_equal(a, o, equal_nan=True)
    ro, rd = random_rays(3000, seed)
    def tr(f):
        try: return f()
        except RuntimeError as e: return "ERR"
    r1, r2 = tr(lambda: hn.rayint(ro, rd)), tr(lambda: ho.rayint(ro, rd))
    if r1 == "ERR" or r2 == "ERR":
        print(seed, "limit: new", r1 == "ERR", "old", r2 == "ERR", flush=True); continue
    same_r = all(np.array_equal(r1[k_], r2[k_]) for k_ in ("t", "n", "tex"))
    same_s = np.array_equal(hn.shadow(ro, rd, 20.0), ho.shadow(ro, rd, 20.0))
    i1, i2 = tr(lambda: hn.inside(ro)), tr(lambda: ho.inside(ro))
    same_i = (isinstance(i1, str) or isinstance(i2, str)) or np.array_equal(i1, i2)
    cnt_same = [int(x) for x in ca] == [int(x) for x in co]
    if not (same and same_r and same_s and same_i and cnt_same):
        nbad += 1
        print(seed, "DIFF frame", same, int((a != o).any(-1).sum()), "rayint", same_r, "shadow", same_s, "inside", same_i, "counters", cnt_same, flush=True)
print("scenes", len(seeds), "different", nbad)
