"""Correctness soak: the 1080p flagship frame rendered whole and as 8 tile shards, over and over, compared on the device.
Reports every mismatch (how many pixels, where, which values).  usage: python soak_shards.py [reps]"""
import ctypes as C, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from glome_amd import api, scenes, dist, _lib as L
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sd = scenes.s3(224)
ctx = api.Context(0)
b = api.Builder(); nm, _ = sd.replay(b); sc = ctx.commit(b, nm[sd.root])
cam = api.camera(*sd.cam); lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
la = (L.Light * len(lights))(*lights)
dev = torch.device("cuda:0")
W, H, world = 1920, 1080, 8
P = api.render_params(width=W, height=H, maxdepth=1)
plans = [dist.ShardPlan(P, r, world, unit=1) for r in range(world)]
ref = torch.zeros((H, W), dtype=torch.int32, device=dev)
sc.render_dev(cam, lights, P, None, ref.data_ptr(), want_stats=False); ctx.synchronize()
bad = 0
t0 = time.time()
for rep in range(reps):
    whole = torch.full((H, W), -1, dtype=torch.int32, device=dev)
    sc.render_dev(cam, lights, P, None, whole.data_ptr(), want_stats=(rep % 7 == 0))
    gathered = torch.full((world, plans[0].maxp), -1, dtype=torch.int32, device=dev)
    for r in range(world):
        st = L.Stats()
        rc = sc.lib.glome_render_tiles_packed_dev(sc.h, C.byref(cam), la, len(lights), C.byref(plans[r].P_local), C.c_void_p(gathered[r].data_ptr()), C.byref(st) if (rep + r) % 5 == 0 else None)
        assert rc == 0, ctx.err()
    frame = torch.full((H, W), -1, dtype=torch.int32, device=dev)
    assert sc.lib.glome_tiles_blit_all_packed_dev(ctx.h, C.byref(P), world, C.c_void_p(gathered.data_ptr()), plans[0].maxp, C.c_void_p(frame.data_ptr())) == 0
    ctx.synchronize()
    for name, t in (("whole", whole), ("shards", frame)):
        if not torch.equal(t, ref):
            d = t != ref
            idx = torch.nonzero(d)
            bad += 1
            print(f"rep {rep} {name}: {int(d.sum())} pixels differ; rows {int(idx[:,0].min())}..{int(idx[:,0].max())} cols {int(idx[:,1].min())}..{int(idx[:,1].max())}; first {idx[:6].cpu().tolist()} got {[hex(t[tuple(i)].item() & 0xffffffff) for i in idx[:4]]} want {[hex(ref[tuple(i)].item() & 0xffffffff) for i in idx[:4]]}", flush=True)
# the float (r, g, b, a, depth) product through the same shards
plans5 = [dist.ShardPlan(P, r, world) for r in range(world)]
ref5 = torch.zeros((H, W, 5), dtype=torch.float32, device=dev)
sc.render_dev(cam, lights, P, ref5.data_ptr(), None, want_stats=False); ctx.synchronize()
for rep in range(reps // 2):
    g5 = torch.full((world, plans5[0].maxp), float("nan"), dtype=torch.float32, device=dev)
    for r in range(world):
        st = L.Stats()
        assert sc.lib.glome_render_tiles_dev(sc.h, C.byref(cam), la, len(lights), C.byref(plans5[r].P_local), C.c_void_p(g5[r].data_ptr()), C.byref(st) if (rep + r) % 3 == 0 else None) == 0
    f5 = torch.full((H, W, 5), float("nan"), dtype=torch.float32, device=dev)
    assert sc.lib.glome_tiles_blit_all_dev(ctx.h, C.byref(P), world, C.c_void_p(g5.data_ptr()), plans5[0].maxp, C.c_void_p(f5.data_ptr()), None) == 0
    ctx.synchronize()
    if not torch.equal(f5, ref5):
        d = (f5 != ref5).any(-1)
        idx = torch.nonzero(d)
        bad += 1
        print(f"float rep {rep}: {int(d.sum())} pixels differ; rows {int(idx[:,0].min())}..{int(idx[:,0].max())} cols {int(idx[:,1].min())}..{int(idx[:,1].max())}; first {idx[:6].cpu().tolist()} got {f5[tuple(idx[0])].cpu().tolist()} want {ref5[tuple(idx[0])].cpu().tolist()}", flush=True)
print("reps", reps, "mismatching frames", bad, "seconds", round(time.time() - t0, 1), flush=True)
