"""Child process of test_gpu_parity.py::test_pipeline_through_a_one_rank_rccl_group: dist.ShardedFrame with the real
torch.distributed collective (backend nccl = RCCL, world size 1), frames compared with direct renders."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29544")
os.environ["RANK"] = "0"; os.environ["WORLD_SIZE"] = "1"
import torch
import torch.distributed as tdist

from glome_amd import api, dist, scenes

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
tdist.init_process_group("nccl", device_id=dev)
sd = scenes.s3(64)
b = api.Builder(); nm, _ = sd.replay(b); ctx = api.Context(0); sc = ctx.commit(b, nm[sd.root])
pos, at, up, fov = sd.cam
cams = [api.camera((pos[0] + 2.0 * k, pos[1] + 0.5 * k, pos[2]), at, up, fov) for k in range(11)]
lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
P = api.render_params(width=640, height=360, maxdepth=1)
for product, group in (("packed", 3), ("rgbad", 1)):
    sf = dist.ShardedFrame(sc, P, 0, 1, dev, lanes=2, product=product, group=group, force_pipeline=True)
    st = sf.step(cams[0], lights, stats=True)
    assert st["rays_primary"] == 640 * 360
    for k in range(11):
        sf.step(cams[k], lights)
    sf.flush()
    torch.cuda.synchronize()
    assert sf.pipe.done == 11
    ctx.lib.glome_ctx_use_slot(ctx.h, None, 0)
    want = torch.zeros((360, 640) if product == "packed" else (360, 640, 5), dtype=torch.int32 if product == "packed" else torch.float32, device=dev)
    if product == "packed":
        sc.render_dev(cams[10], lights, P, None, want.data_ptr())
    else:
        sc.render_dev(cams[10], lights, P, want.data_ptr())
    ctx.synchronize()
    assert torch.equal(sf.frame, want), product
tdist.barrier()
tdist.destroy_process_group()
print("rccl one-rank pipeline ok")
