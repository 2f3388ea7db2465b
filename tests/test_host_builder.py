"""Host side of the boundary (no GPU): the product's scene graph, bounds, BIH / Mesh builders and flattening against
the oracle's own restatement of the same reference code, plus the C-ABI surface."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import ROOT, HostSim, O
from glome_amd import api, scenes
import zoo


def both(sd):
    b = api.Builder()
    nm, mm = sd.replay(b)
    o, om, _ = O.load_scene(sd)
    return b, nm, o, om


def sd_bih_nodes(sd):
    return [i for i, op in enumerate([op for op in sd.ops if op[0] == "n"]) if op[1] == "bih"]


@pytest.mark.parametrize("name", ["S1", "S3small", "S4"] + list(zoo.ALL))
def test_bih_trees_bounds_primcount_match_oracle(built, name):
    sd = {"S1": lambda: scenes.s1(), "S3small": lambda: scenes.s3(24), "S4": scenes.s4}.get(name, zoo.ALL.get(name))()
    if any(op[0] == "N" for op in sd.ops):
        pytest.skip("bulk triangles: node numbering differs; covered by test_bih_tree_s3")
    b, nm, o, om = both(sd)
    assert b.primcount(nm[sd.root]) == o.primcount(om[sd.root])
    kinds = [op[1] for op in sd.ops if op[0] == "n"]
    for i, k in enumerate(kinds):
        if k in ("plane", "plane_offset"):
            continue
        assert np.allclose(b.bound(nm[i]), o.bound(om[i]), rtol=0, atol=0), (i, k)
        if k == "bih":
            d1, d2 = o.bih_dump(om[i]), b.bih_dump(nm[i])
            assert np.array_equal(d1[2], d2[2]) and np.array_equal(d1[3], d2[3])      # axes, leaf sizes
            assert np.array_equal(d1[0], d2[0]) and np.array_equal(d1[1], d2[1])      # split planes, bit for bit
            inv1 = {v: k2 for k2, v in enumerate(om)}; inv2 = {v: k2 for k2, v in enumerate(nm)}
            assert [inv1[x] for x in d1[4]] == [inv2[x] for x in d2[4]]                # leaf items in order


def test_bih_tree_s3(built):
    sd = scenes.s3(48)
    b, nm, o, om = both(sd)
    i = sd.n_nodes - 2
    d1, d2 = o.bih_dump(om[i]), b.bih_dump(nm[i])
    for k in range(4):
        assert np.array_equal(d1[k], d2[k])
    inv1 = np.full(max(om) + 1, -1); inv1[om] = np.arange(len(om))
    inv2 = np.full(max(nm) + 1, -1); inv2[nm] = np.arange(len(nm))
    assert np.array_equal(inv1[d1[4]], inv2[d2[4]])
    assert sorted(inv2[d2[4]].tolist()) == list(range(2 * 48 * 48))  # every triangle in exactly one leaf


def test_group_transform_rewrites(built):
    """group [] = Void, group [x] = x, nested groups flatten (Solid.hs:293-302); transform merges instances and bakes
    triangles (Solid.hs:494-496, Triangle.hs:164-168); flatten_transform drops Bound wrappers (Bound.hs:73-74)."""
    b = api.Builder()
    s1, s2, s3 = b.sphere((0, 0, 0), 1), b.sphere((3, 0, 0), 1), b.sphere((6, 0, 0), 1)
    assert b.primcount(b.group([])) == (1, 0, 0)  # Void counts as one primitive by the class default (Solid.hs:251)
    assert b.group([s2]) == s2
    g = b.group([b.group([s1, s2]), s3])
    assert b.primcount(g) == (3, 0, 0)
    t = b.transform(b.transform(s1, [api.translate((1, 0, 0))]), [api.translate((0, 2, 0))])
    assert b.primcount(t) == (1, 1, 0)  # one merged matrix, not two
    assert np.allclose(b.bound(t), [0 - 1e-4, 1 - 1e-4, -1 - 1e-4, 2 + 1e-4, 3 + 1e-4, 1 + 1e-4])  # bbpts pads by delta
    tri = b.transform(b.triangle((0, 0, 0), (1, 0, 0), (0, 1, 0)), [api.scale((2, 2, 2))])
    assert b.primcount(tri) == (1, 0, 0)  # baked: no Instance
    assert np.allclose(b.bound(tri), [-1e-4, -1e-4, -1e-4, 2 + 1e-4, 2 + 1e-4, 1e-4])
    bounded = b.bound_object(b.sphere((0, 0, 0), 5), b.group([s1, s2]))
    assert b.primcount(bounded) == (2, 0, 1)
    inst = b.transform(bounded, [api.translate((0, 1, 0))])
    flat = b.flatten_transform(inst)
    assert b.primcount(flat) == (2, 2, 0)  # bound dropped, the matrix pushed to both leaves
    assert b.primcount(b.tolist(g)) == (3, 0, 0)


def test_constructor_errors_like_the_reference(built):
    b = api.Builder()
    with pytest.raises(api.GlomeError, match="infinite bounding box"):
        b.bih([b.sphere((0, 0, 0), 1), b.plane((0, 0, 0), (0, 1, 0))])  # Bih.hs:319-322
    with pytest.raises(api.GlomeError):
        b.transform(b.sphere((0, 0, 0), 1), [np.concatenate([np.eye(3, 4).ravel(), 3 * np.eye(3, 4).ravel()])])  # check_xfm
    with pytest.raises(api.GlomeError):
        b.tex(b.sphere((0, 0, 0), 1), 99)  # unknown material
    with pytest.raises(api.GlomeError):
        b.group([12345])
    with pytest.raises(api.GlomeError):
        b.mesh([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [], [[0, 1, 7, -1, -1, -1, -1, -1]], [])
    with pytest.raises(api.GlomeError):
        api.xyz_to_uvw((1, 0, 0), (1, 0, 0), (0, 0, 1))  # not orthogonal (Vec.hs:602-622)
    # an intersection bounded by planes is finite (Csg.hs:116-120), so it may enter a bih
    x = b.intersection([b.sphere((0, 0, 0), 2), b.plane_offset((0, 1, 0), 0.5)])
    b.bih([x, b.sphere((4, 0, 0), 1)])


def test_flatten_tiers_and_limits(built):
    b = api.Builder()
    sd = scenes.s1()
    nm, _ = sd.replay(b)
    assert HostSim(b, nm[sd.root]).info()["tier"] == 0
    for name in ("csg", "nested"):  # composites below composites: the interpreter
        sd = zoo.ALL[name]()
        b = api.Builder(); nm, _ = sd.replay(b)
        assert HostSim(b, nm[sd.root]).info()["tier"] == 1, name
    sd = scenes.s4()  # CSG over primitives (and an Instance of one) stays flat: BASELINE configs[3]
    b = api.Builder(); nm, _ = sd.replay(b)
    assert HostSim(b, nm[sd.root]).info()["tier"] == 0
    for name in ("flat_mixed", "mesh", "materials", "quadrics"):  # (a cylinder / cone is an Instance of a canonical quadric)
        sd = zoo.ALL[name]()
        b = api.Builder(); nm, _ = sd.replay(b)
        assert HostSim(b, nm[sd.root]).info()["tier"] == 0, name
    # composites nest as deep as the scene goes (zoo.deep_nest renders in the parity tests): sixteen levels, and a Difference
    # carving them (its get_metainfo walks the sixteen levels over explicit frames like everything else)
    b = api.Builder()
    n = b.sphere((0, 0, 0), 1)
    for _ in range(8):
        n = b.group([b.transform(n, [api.translate((0.1, 0, 0))]), b.sphere((9, 9, 9), 0.1)])
    assert HostSim(b, n).info()["nesting"] == 16
    assert HostSim(b, b.difference(n, b.sphere((0.5, 0, 0), 0.7))).info()["nesting"] == 17
    # what bounds the nesting now is the interpreter's frame memory, estimated at commit from the frames each node needs
    for _ in range(60):  # (2,048 frame words since round 4: a group and the Instance inside it take 34)
        n = b.group([b.transform(n, [api.translate((0.1, 0, 0))]), b.sphere((9, 9, 9), 0.1)])
    with pytest.raises(RuntimeError, match="frame memory"):
        HostSim(b, n)


def test_c_abi_exports_every_declared_symbol(built):
    """The shared library loads and exports every function include/glome_hip.h declares (no compute calls here)."""
    hdr = open(os.path.join(ROOT, "include", "glome_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(glome_[a-z0-9_]+)\s*\(", hdr))
    lib_path = os.path.join(ROOT, "glome_amd", "libglome_hip.so")
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path]).decode()
    exported = set(re.findall(r" T (glome_[a-z0-9_]+)", out))
    assert declared and declared <= exported, sorted(declared - exported)
    from glome_amd import _lib
    lib = _lib.load()
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared <= bound, sorted(declared - bound)
    for name in declared:
        assert getattr(lib, name) is not None


def test_no_gpu_means_loud_failure_not_fallback(built):
    """Without a usable gfx950 device the compute API refuses to run; it never silently computes elsewhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.GlomeError, match="no CPU fallback"):
        api.Context(0)
    src = "".join(open(os.path.join(ROOT, "glome_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "glome_amd")) if f.endswith(".py"))
    assert "oracle" not in src.replace("# the oracle", "")  # the product package never imports the checker


@pytest.mark.parametrize("world,pct", [(2, 90), (3, 70), (4, 75), (8, 60), (8, 35), (5, 0), (8, 100)])
def test_weighted_shards_partition_the_frame(built, world, pct):
    """glome_render_params.rank0_share_pct: every tile of the frame belongs to exactly one rank, rank 0's share is its stated
    weight (percent of one other rank's), the other ranks' shares differ by a few tiles at most, and no rank owns long runs
    of neighbouring tiles (the pattern interleaves)."""
    from glome_amd import dist
    P = api.render_params(width=1920, height=1080, blocksize=64, rank0_share_pct=pct)
    whole = dist.owned_layout(api.render_params(width=1920, height=1080, blocksize=64), 0, 1)
    seen = {}
    counts = []
    for r in range(world):
        lay = dist.owned_layout(P, r, world)
        counts.append(len(lay))
        for x, y, w, h, base in lay:
            assert (x, y) not in seen
            seen[(int(x), int(y))] = r
        assert dist.payload_floats(P, r, world) == 5 * int((lay[:, 2] * lay[:, 3]).sum())
    assert len(seen) == len(whole)
    fair = len(whole) / world
    if pct in (0, 100):
        order = [seen[(int(x), int(y))] for x, y, _, _, _ in whole]
        assert order == [k % world for k in range(len(whole))]  # plain round robin, as before
    else:
        want0 = len(whole) * (pct / 100.0) / (pct / 100.0 + world - 1)
        assert abs(counts[0] - want0) <= 2, (counts, want0)
        assert max(counts[1:]) - min(counts[1:]) <= 2, counts
        order = [seen[(int(x), int(y))] for x, y, _, _, _ in whole]
        runs = max(len(list(g)) for _, g in __import__("itertools").groupby(order))
        assert runs <= 2, runs


def test_every_share_percent_is_its_own_layout(built):
    """the weight is honoured at percent granularity (until round 3 it was quantised to tens: 65..70 were one layout)"""
    from glome_amd import dist
    n0 = []
    for pct in (64, 65, 66, 67, 68, 70, 85, 90, 95):
        P = api.render_params(width=3840, height=2160, blocksize=64, rank0_share_pct=pct)
        n0.append(len(dist.owned_layout(P, 0, 8)))
    assert n0 == sorted(n0) and len(set(n0)) == len(n0), n0
    lay = [dist.owned_layout(api.render_params(width=1920, height=1080, blocksize=64, rank0_share_pct=pct), 0, 2).tolist() for pct in (85, 90)]
    assert lay[0] != lay[1]
