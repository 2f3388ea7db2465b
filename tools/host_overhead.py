"""Per-frame host + launch overhead of the frame pipeline: a tiny frame (GPU work negligible), world = 8 rehearsal with
the collective replaced by a device copy, and the bare C calls.  Not a test."""
import ctypes as C, json, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from glome_amd import api, scenes, dist, _lib as L
from helpers import product_camera_lights
sd = scenes.s3(64)
b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
cam, lights = product_camera_lights(sd)
dev = torch.device("cuda:0")
P = api.render_params(width=130, height=130, maxdepth=1)
la = (L.Light * len(lights))(*lights)
class _Done:
    def wait(self): return True
def timeit(fn, K=2000):
    for i in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(K): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e6
pay = torch.zeros(130 * 130, dtype=torch.int32, device=dev)
Pl = dist._clone_params(P, tile_first=0, tile_stride=1)
print("render_tiles_packed_dev call: %.1f us" % timeit(lambda: sc.lib.glome_render_tiles_packed_dev(sc.h, C.byref(cam), la, len(lights), C.byref(Pl), C.c_void_p(pay.data_ptr()), None)))
fr = torch.zeros((130, 130), dtype=torch.int32, device=dev)
print("blit_all_packed call: %.1f us" % timeit(lambda: sc.lib.glome_tiles_blit_all_packed_dev(ctx.h, C.byref(P), 1, C.c_void_p(pay.data_ptr()), 130 * 130, C.c_void_p(fr.data_ptr()))))
s = torch.cuda.Stream(device=dev)
print("use_slot call: %.1f us" % timeit(lambda: sc.lib.glome_ctx_use_slot(ctx.h, C.c_void_p(s.cuda_stream), 1)))
sc.lib.glome_ctx_use_slot(ctx.h, None, 0)
ev = torch.cuda.Event()
print("event record: %.1f us" % timeit(lambda: ev.record(s)))
print("stream wait_event: %.1f us" % timeit(lambda: s.wait_event(ev)))
def ctxmgr():
    with torch.cuda.stream(s): pass
print("with torch.cuda.stream: %.1f us" % timeit(ctxmgr))
for world, lanes, group in ((8, 8, 4), (8, 8, 1)):
    sf = dist.ShardedFrame(sc, P, 0, world, dev, lanes=lanes, product="packed", group=group)
    def fake(payload, gathered, async_op=False):
        gathered[0].copy_(payload); return _Done()
    sf.plan.gather = fake
    print("ShardedFrame.step world %d lanes %d group %d: %.1f us per frame" % (world, sf.n, sf.G, timeit(lambda: sf.step(cam, lights), 1000)))
    sf.flush(); sc.lib.glome_ctx_use_slot(ctx.h, None, 0)
