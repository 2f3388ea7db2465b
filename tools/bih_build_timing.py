"""Host (glome_sb_bih) against device (glome_sb_bih_dev) BIH build of the terrain scenes: same tree, time of each.  Not a test."""
import json, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from glome_amd import api, scenes

ctx = api.Context(0)
for n in (224, 708):
    sd = scenes.s3(n)
    b = api.Builder()
    nm, _ = sd.replay(b)
    nid, host, ids = 0, None, None
    for kind, name, args in sd.ops:
        if kind == "N":
            nid += args[0].shape[0]
        elif kind == "n":
            if name == "bih":
                host, ids = nm[nid], [nm[i] for i in args[0]]
            nid += 1
    t0 = time.perf_counter(); again = b.bih(ids); t_host = time.perf_counter() - t0
    best = None
    for rep in range(3):
        t0 = time.perf_counter(); dev, ms = ctx.bih(b, ids); wall = time.perf_counter() - t0
        best = (ms, wall) if best is None or wall < best[1] else best
    da, dd = b.bih_dump(host), b.bih_dump(dev)
    same = all(np.array_equal(da[k], dd[k]) for k in range(5))
    print(json.dumps({"triangles": len(ids), "nodes": int(len(da[0])), "same_tree": bool(same), "host_build_s": round(t_host, 3),
                      "device_build_gpu_ms": round(best[0], 2), "device_build_wall_s_incl_bounds_upload_readback": round(best[1], 3)}), flush=True)

sd = scenes.s3(224, as_mesh=True)
b = api.Builder()
t0 = time.perf_counter(); nm, mm = sd.replay(b); t_all = time.perf_counter() - t0
nid = 0
for kind, name, args in sd.ops:
    if kind == "N":
        nid += args[0].shape[0]
    elif kind == "n":
        if name == "mesh":
            V, Nn, T, mats = args
            host = nm[nid]
            t0 = time.perf_counter(); again = b.mesh(V, Nn, T, [mm[m] for m in mats]); t_host = time.perf_counter() - t0
            t0 = time.perf_counter(); dev, ms = ctx.mesh(b, V, Nn, T, [mm[m] for m in mats]); wall = time.perf_counter() - t0
            print(json.dumps({"mesh_triangles": int(len(T)), "same_tree": b.show(host) == b.show(dev), "host_mesh_call_s": round(t_host, 3), "device_build_gpu_ms": round(ms, 2),
                              "device_mesh_call_s": round(wall, 3)}), flush=True)
        nid += 1
