"""Child process of test_gpu_parity.py::test_rccl_branch_of_the_multi_gpu_entry_with_a_stub_transport: the RCCL branch of
glome_multi_render (ncclCommInitAll, one group of ncclSend / ncclRecv per call on the ranks' streams, slab offsets, stream
order) executed on ONE GPU -- GLOME_DEBUG_RCCL_LIB names tests/rcclstub/librccl_stub.so, whose send / recv pairs are
stream-ordered device copies, and GLOME_DEBUG_RCCL_SAME_DEVICE lifts the distinct-device condition.  A fresh process because
the library resolves its transport once."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
stub = os.path.join(ROOT, "tests", "rcclstub", "librccl_stub.so")
os.environ["GLOME_DEBUG_RCCL_LIB"] = stub
os.environ["GLOME_DEBUG_RCCL_SAME_DEVICE"] = "1"
import torch

from glome_amd import api, scenes

dev = torch.device("cuda", 0)
sd = scenes.s3(48)
pos, at, up, fov = sd.cam
cams = [api.camera((pos[0] + 2.0 * k, pos[1] + 0.5 * k, pos[2]), at, up, fov) for k in range(4)]
lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
W, H = 645, 390
pairs_expected = 0
for n in (2, 3, 8):
    ctxs = [api.Context(0) for _ in range(n)]
    scs = []
    for c in ctxs:
        b = api.Builder(); nm, _ = sd.replay(b)
        scs.append(c.commit(b, nm[sd.root]))
    for mode, views in ((0, cams), (1, cams[:1])):
        P = api.render_params(width=W, height=H, mode=mode, maxdepth=1, rank0_share_pct=70 if n == 3 else 0)
        m = api.Multi(scs, P)
        assert m.transport() == "rccl", m.transport()
        out = torch.full((len(views), H, W), -1, dtype=torch.int32, device=dev)
        for rep in range(3):  # later calls reuse the payload buffers: a rank's Send is ordered in front of its next render
            m.render(views, lights, out.data_ptr())
            pairs_expected += n - 1
        m.synchronize()
        for k, cam in enumerate(views):
            want = torch.zeros((H, W), dtype=torch.int32, device=dev)
            scs[0].render_dev(cam, lights, P, None, want.data_ptr())
            ctxs[0].synchronize()
            assert torch.equal(out[k], want), (n, mode, k, int((out[k] != want).sum()))
        m.close()
    for s_ in scs:
        s_.release()
    for c in ctxs:
        c.close()
lib = C.CDLL(stub)  # (the same mapping the product opened)
g, p = C.c_int(0), C.c_int(0)
lib.rccl_stub_counts(C.byref(g), C.byref(p))
assert p.value == pairs_expected and g.value > 0, (g.value, p.value, pairs_expected)
print("rccl stub transport ok: %d groups, %d send/recv pairs" % (g.value, p.value))
