"""N4 / N1: the text GlomeView prints for a scene (`show geom`, Glome.hs:431) as an interchange format -- glome_sb_show writes
it, glome_sb_load_show reads it (glome_amd/csrc/show_format.hpp).  Pinned here against an independent Python reading of
the same rules (tests/showfmt.py), hand-written literals, the builder's own inspection calls and the oracle.  No GHC in
this image: nothing here is a real dump ("parity unpinned" for the format itself)."""
import math

import numpy as np
import pytest

import showfmt
import zoo
from helpers import HostSim, oracle_for, random_rays
from glome_amd import api
from glome_amd.scene import SceneDesc


def _build(sd):
    b = api.Builder()
    nmap, mmap = sd.replay(b)
    return b, nmap[sd.root], mmap


def _floats(v):
    if isinstance(v, float):
        yield v
    elif isinstance(v, (tuple, list)):
        for a in v:
            yield from _floats(a)


def test_doubles_print_like_ghc_show(built):
    # GHC.Float.showFloat: fixed notation for 0.1 <= |x| < 10^7, d.ddde<n> otherwise, shortest digits, negatives
    # parenthesised as constructor arguments (showsPrec 11)
    cases = [0.0, 1.0, 0.1, 0.01, 1e7, 9999999.0, 123.456, 5e-324, 1 / 3, 1e21, 12345678.9, 0.099, 1234567.0, 1.5e300, 4.0001, 2.0 ** -1074 * 3, 0.30000000000000004]
    want = ["0.0", "1.0", "0.1", "1.0e-2", "1.0e7", "9999999.0", "123.456", "5.0e-324", "0.3333333333333333", "1.0e21", "1.23456789e7", "9.9e-2",
            "1234567.0", "1.5e300", "4.0001", "1.5e-323", "0.30000000000000004"]
    assert [showfmt.hs_double(x) for x in cases] == want
    b = api.Builder()
    for x, w in zip(cases, want):
        text = b.show(b.plane_offset((0.0, 1.0, -x), x))
        assert text == "SI Plane (Vec 0.0 1.0 %s) %s" % ("(-%s)" % w, w), text
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.normal(size=200) * 10.0 ** rng.integers(-12, 12, size=200), rng.normal(size=50).astype(np.float32).astype(np.float64)])
    for x in xs:
        text = b.show(b.plane_offset((1.0, 0.0, 0.0), float(x)))
        arg = showfmt.hs_double(float(x))
        assert text == "SI Plane (Vec 1.0 0.0 0.0) " + ("(%s)" % arg if x < 0 else arg)
        assert float(arg) == x  # lossless


def test_hand_written_literal_loads_and_traces_like_the_oracle(built):
    # typed by hand from the derived-Show rules; `cone` is a canonical Cone inside an Instance (Cone.hs:40-67)
    text = ("SI [SI Tex SI Sphere (Vec 0.0 1.0 0.0) 1.0 1.0 Texture,"
            "SI Plane (Vec 0.0 1.0 0.0) (-0.5),"
            "SI Difference SI Box (Bbox {p1 = Vec 2.0 0.0 (-1.0), p2 = Vec 4.0 2.0 1.0}) SI Sphere (Vec 3.0 1.0 1.0) 0.75 1.3333333333333333 True,"
            "SI <Tag SI NoShadow SI Disc (Vec (-3.0) 1.0 0.0) (Vec 0.0 0.0 1.0) 0.25>,"
            "SI Instance SI Intersection [SI Sphere (Vec (-0.5) 0.0 0.0) 1.0 1.0,SI Sphere (Vec 0.5 0.0 0.0) 1.0 1.0] "
            "(Xfm (Matrix 1.0 0.0 0.0 (-2.0) 0.0 1.0 0.0 3.0 0.0 0.0 1.0 0.0) (Matrix 1.0 0.0 0.0 2.0 0.0 1.0 0.0 (-3.0) 0.0 0.0 1.0 0.0))]")
    b = api.Builder()
    m = b.material_surface((1, 0, 0), 1, 0.2, 0.8, 0, 0)
    root, ntex = b.load_show(text, [m])
    assert ntex == 1 and b.show(root) == text
    sd = SceneDesc(round32=False)
    sm = sd.material_surface((1, 0, 0), 1, 0.2, 0.8, 0, 0)
    isect = sd.intersection([sd.sphere((-0.5, 0, 0), 1), sd.sphere((0.5, 0, 0), 1)])
    sd.set_root(sd.group([sd.tex(sd.sphere((0, 1, 0), 1), sm), sd.plane_offset((0, 1, 0), -0.5),
                          sd.difference(sd.box((2, 0, -1), (4, 2, 1)), sd.sphere((3, 1, 1), 0.75)),
                          sd.tag(sd.noshadow(sd.disc((-3, 1, 0), (0, 0, 1), 0.5))),
                          sd.transform(isect, [api.translate(np.float64([-2, 3, 0]))])]))
    sd.set_camera((0, 2, -10), (0, 1, 0), (0, 1, 0), 45)
    o, om, _ = oracle_for(sd)
    assert b.primcount(root) == o.primcount(om[sd.root])
    assert np.array_equal(b.bound(root), o.bound(om[sd.root]))
    ro, rd = random_rays(600, 5, center=(0, 1, 0), radius=9, spread=4)
    got = HostSim(b, root).rayint(ro, rd)
    want = o.rayint(om[sd.root], ro.astype(np.float64), rd.astype(np.float64))
    hit = want["t"] >= 0
    assert np.array_equal(got["t"] >= 0, hit) and hit.sum() > 200
    assert np.allclose(got["t"][hit], want["t"][hit], rtol=2e-4, atol=1e-5)


# Conformance vectors for `show geom` (Glome.hs:431), typed by hand from GHC's rules -- NOT produced by glome_sb_show:
#   * a derived Show instance prints `Con a b ..` with every argument at precedence 11, so an argument that is itself an
#     application, or a negative number, stands in parentheses (Haskell 2010 report 11.4; showSignedFloat parenthesises x < 0 and
#     -0.0 above precedence 6); record syntax prints `Con {f = v, ..}` with the fields at precedence 0 (no parentheses inside),
#     and the record as a whole is parenthesised as an argument (Bbox, Vec.hs:646; Bih, Bih.hs:52);
#   * showFloat: fixed notation for 0.1 <= x < 10^7, otherwise d.ddde<n> with the shortest digits that read back;
#   * SolidItem, Tag and Texture have hand-written `show` (Solid.hs:277-278, Tex.hs:50-51, Solid.hs:101-102): "SI " ++ show s,
#     "<Tag " ++ show s ++ ">", "Texture" -- never parenthesised, whatever the context (the class default showsPrec ignores the
#     precedence);
#   * a list prints as [a,b,c], no spaces, elements at precedence 0 (showList).
# Each row: what the constructors build, and the text GHC prints for it.  Checked three ways: the writer prints exactly this, the
# reader takes it and prints it back, and the reader's scene is the constructors' scene (same primitive count, bound, rays).
_CONFORMANCE = [
    # negative doubles and negative zero as constructor arguments; `plane_offset pt off = Plane pt off` (Plane.hs:24-25)
    (lambda b, m: b.plane_offset((0.0, 1.0, -0.0), -0.5), "SI Plane (Vec 0.0 1.0 (-0.0)) (-0.5)"),
    # the switch between fixed and exponent notation at 0.1 and 10^7; `sphere c r = Sphere c r (1.0/r)` (Sphere.hs:15-17)
    (lambda b, m: b.sphere((1.0e-2, 1.0e7, 9999999.0), 0.1), "SI Sphere (Vec 1.0e-2 1.0e7 9999999.0) 0.1 10.0"),
    (lambda b, m: b.sphere((0.099, 12345678.0, -1234567.5), 4.0), "SI Sphere (Vec 9.9e-2 1.2345678e7 (-1234567.5)) 4.0 0.25"),
    # SI inside derived constructors: no parentheses around an item, `Texture` for the closure
    (lambda b, m: b.tex(b.noshadow(b.onlyshadow(b.sphere((0, 0, 0), 1))), m), "SI Tex SI NoShadow SI OnlyShadow SI Sphere (Vec 0.0 0.0 0.0) 1.0 1.0 Texture"),
    (lambda b, m: b.tag(b.tag(b.sphere((0, 0, 0), 1))), "SI <Tag SI <Tag SI Sphere (Vec 0.0 0.0 0.0) 1.0 1.0>>"),
    # lists: Intersection [SolidItem] (Csg.hs:15), a group inside it (the list instance's own Show, Solid.hs:326)
    (lambda b, m: b.intersection([b.sphere((0, 0, 0), 2), b.plane_offset((1, 0, 0), 1.0), b.group([b.sphere((0, 0, 0), 1), b.sphere((1, 0, 0), 1)])]),
     "SI Intersection [SI Sphere (Vec 0.0 0.0 0.0) 2.0 0.5,SI Plane (Vec 1.0 0.0 0.0) 1.0,SI [SI Sphere (Vec 0.0 0.0 0.0) 1.0 1.0,SI Sphere (Vec 1.0 0.0 0.0) 1.0 1.0]]"),
    # `group []` = Void, `group [x]` = x (Solid.hs:293-296)
    (lambda b, m: b.group([]), "SI Void"),
    (lambda b, m: b.group([b.sphere((0, 0, 0), 1)]), "SI Sphere (Vec 0.0 0.0 0.0) 1.0 1.0"),
    # a record as an argument: parenthesised as a whole, fields bare
    (lambda b, m: b.bound_object(b.sphere((0, 0, 0), 2), b.box((-1, -1, -1), (1, 1, 1))),
     "SI Bound SI Sphere (Vec 0.0 0.0 0.0) 2.0 0.5 SI Box (Bbox {p1 = Vec (-1.0) (-1.0) (-1.0), p2 = Vec 1.0 1.0 1.0})"),
    (lambda b, m: b.innerbound(b.sphere((0, 0, 0), 2), b.box((-1, -1, -1), (1, 1, 1))),
     "SI InnerBound SI Sphere (Vec 0.0 0.0 0.0) 2.0 0.5 SI Box (Bbox {p1 = Vec (-1.0) (-1.0) (-1.0), p2 = Vec 1.0 1.0 1.0})"),
    # Instance (Solid.hs:386) with Xfm = forward and inverse Matrix (Vec.hs:407-414): scale (2,1,1) then translate (-2,3,0)
    (lambda b, m: b.transform(b.sphere((0, 0, 0), 1), [api.scale((2, 1, 1)), api.translate((-2, 3, 0))]),
     "SI Instance SI Sphere (Vec 0.0 0.0 0.0) 1.0 1.0 (Xfm (Matrix 2.0 0.0 0.0 (-2.0) 0.0 1.0 0.0 3.0 0.0 0.0 1.0 0.0) (Matrix 0.5 0.0 0.0 1.0 0.0 1.0 0.0 (-3.0) 0.0 0.0 1.0 0.0))"),
    # Bih {bihbb, bihroot} (Bih.hs:51-57): three objects or fewer stay one leaf (Bih.hs:222)
    (lambda b, m: b.bih([b.sphere((0, 0, 0), 0.5), b.sphere((1, 0, 0), 0.5)]),
     "SI Bih {bihbb = Bbox {p1 = Vec (-0.5) (-0.5) (-0.5), p2 = Vec 1.5 0.5 0.5}, bihroot = BihLeaf [SI Sphere (Vec 0.0 0.0 0.0) 0.5 2.0,SI Sphere (Vec 1.0 0.0 0.0) 0.5 2.0]}"),
    # ... four split once along x: planes lmax + delta = 1.5001, rmin - delta = 9.4999 (Bih.hs:262-266), nested nodes parenthesised
    (lambda b, m: b.bih([b.sphere((0, 0, 0), 0.5), b.sphere((1, 0, 0), 0.5), b.sphere((10, 0, 0), 0.5), b.sphere((11, 0, 0), 0.5)]),
     "SI Bih {bihbb = Bbox {p1 = Vec (-0.5) (-0.5) (-0.5), p2 = Vec 11.5 0.5 0.5}, bihroot = BihBranch 1.5001 9.4999 0 "
     "(BihLeaf [SI Sphere (Vec 0.0 0.0 0.0) 0.5 2.0,SI Sphere (Vec 1.0 0.0 0.0) 0.5 2.0]) (BihLeaf [SI Sphere (Vec 10.0 0.0 0.0) 0.5 2.0,SI Sphere (Vec 11.0 0.0 0.0) 0.5 2.0])}"),
    # Difference a b Bool (Csg.hs:14): `difference` True, `difference_retexture` False (Csg.hs:26-30)
    (lambda b, m: b.difference(b.sphere((0, 0, 0), 1), b.sphere((1, 0, 0), 1)), "SI Difference SI Sphere (Vec 0.0 0.0 0.0) 1.0 1.0 SI Sphere (Vec 1.0 0.0 0.0) 1.0 1.0 True"),
    (lambda b, m: b.difference_retexture(b.sphere((0, 0, 0), 1), b.sphere((1, 0, 0), 1)), "SI Difference SI Sphere (Vec 0.0 0.0 0.0) 1.0 1.0 SI Sphere (Vec 1.0 0.0 0.0) 1.0 1.0 False"),
    (lambda b, m: b.triangle((0, 0, 0), (1, 0, 0), (0, 1, 0)), "SI Triangle (Vec 0.0 0.0 0.0) (Vec 1.0 0.0 0.0) (Vec 0.0 1.0 0.0)"),
    # `disc pos norm r = Disc pos norm (r*r)` (Cone.hs:29-31): the normal as given, the radius squared
    (lambda b, m: b.disc((0, 1, 0), (0, 0, 2), 0.5), "SI Disc (Vec 0.0 1.0 0.0) (Vec 0.0 0.0 2.0) 0.25"),
    # `cylinder` / `cone` = the canonical z-axis solid inside a transform (Cone.hs:40-67): Cylinder r 0 len; Cone r1 0 len (r1*len/(r1-r2))
    (lambda b, m: b.cylinder((0, 0, 0), (0, 0, 2), 0.5),
     "SI Instance SI Cylinder 0.5 0.0 2.0 (Xfm (Matrix 0.0 (-1.0) 0.0 0.0 1.0 0.0 0.0 0.0 0.0 0.0 1.0 0.0) (Matrix 0.0 1.0 0.0 0.0 (-1.0) 0.0 0.0 0.0 0.0 0.0 1.0 0.0))"),
    (lambda b, m: b.cone((0, 0, 0), 1.0, (0, 0, 2), 0.5),
     "SI Instance SI Cone 1.0 0.0 2.0 4.0 (Xfm (Matrix 0.0 (-1.0) 0.0 0.0 1.0 0.0 0.0 0.0 0.0 0.0 1.0 0.0) (Matrix 0.0 1.0 0.0 0.0 (-1.0) 0.0 0.0 0.0 0.0 0.0 1.0 0.0))"),
    # showFloat's names for the non-finite values; a negative one is parenthesised like any negative number
    (lambda b, m: b.plane_offset((0, 1, 0), float("-inf")), "SI Plane (Vec 0.0 1.0 0.0) (-Infinity)"),
]


@pytest.mark.parametrize("k", range(len(_CONFORMANCE)))
def test_show_conformance_vectors(built, k):
    make, literal = _CONFORMANCE[k]
    b = api.Builder()
    m = b.material_surface((1, 0, 0), 1, 0.2, 0.8, 0, 0)
    made = make(b, m)
    assert b.show(made) == literal                       # the writer against the rule
    root, ntex = b.load_show(literal, [m] * literal.count("Texture"))
    assert ntex == literal.count("Texture")
    assert b.show(root) == literal                       # the reader takes it, and loses nothing
    assert showfmt.parse(literal)[0] == "SI"             # (and so does the independent reader)
    assert b.primcount(root) == b.primcount(made)
    if "Infinity" in literal or "Void" in literal:
        return
    assert np.array_equal(b.bound(root), b.bound(made))
    ro, rd = random_rays(300, 17 + k, center=(0.5, 0.5, 0.5), radius=6, spread=3)
    g1, g2 = HostSim(b, made).rayint(ro, rd), HostSim(b, root).rayint(ro, rd)
    for key in ("t", "n", "tex"):
        assert np.array_equal(g1[key], g2[key]), key


def test_reader_takes_text_the_writer_never_prints(built):
    # text a GHC dump can contain that glome_sb_show itself never produces: branch planes and boxes that no builder call made
    # (the reader must take the tree as written, not rebuild it), and a leaf with more than three items
    text = ("SI Bih {bihbb = Bbox {p1 = Vec (-1.0) (-1.0) (-1.0), p2 = Vec 4.0 1.0 1.0}, bihroot = BihBranch 0.75 2.25 0 "
            "(BihLeaf [SI Sphere (Vec 0.0 0.0 0.0) 0.75 1.3333333333333333]) "
            "(BihBranch 0.5 (-0.5) 1 (BihLeaf [SI Sphere (Vec 3.0 0.0 0.0) 0.5 2.0,SI Sphere (Vec 3.0 0.25 0.0) 0.5 2.0,SI Sphere (Vec 3.0 (-0.25) 0.0) 0.5 2.0,SI Sphere (Vec 3.0 0.0 0.25) 0.5 2.0]) (BihLeaf []))}")
    b = api.Builder()
    root, ntex = b.load_show(text, [])
    assert ntex == 0 and b.show(root) == text and b.primcount(root)[0] == 5
    tree = showfmt.parse(text)
    assert tree[1][0] == "Bih" and tree[1][2][0] == "BihBranch" and tree[1][2][3] == 0 and tree[1][2][5][3] == 1

@pytest.mark.parametrize("name", ["flat_mixed", "quadrics", "csg", "nested", "materials", "textures", "soup", "retexture"])
def test_round_trip_is_the_same_scene(built, name):
    sd = zoo.soup(300) if name == "soup" else getattr(zoo, name)()
    b, root, _ = _build(sd)
    text = b.show(root)
    # the independent reader agrees with the writer: structure parses, every double reads back exactly as printed
    tree = showfmt.parse(text)
    assert tree[0] == "SI"
    for x in _floats(tree):
        assert showfmt.hs_double(x) in text
    mats = b.show_tex_materials(root)
    assert len(mats) == sum(1 for c in showfmt.walk(tree) if c[0] == "Tex")
    root2, ntex = b.load_show(text, mats)
    assert ntex == len(mats) and root2 != root
    assert b.show(root2) == text and b.show_tex_materials(root2) == mats
    assert b.primcount(root2) == b.primcount(root) and np.array_equal(b.bound(root2), b.bound(root))
    h1, h2 = HostSim(b, root), HostSim(b, root2)
    i1, i2 = h1.info(), h2.info()
    assert i1 == i2
    ro, rd = random_rays(500, 11)
    r1, r2 = h1.rayint(ro, rd), h2.rayint(ro, rd)
    for k in ("t", "n", "tex"):  # builder ids differ (the copy's nodes are new), everything the shader sees does not
        assert np.array_equal(r1[k], r2[k]), k
    assert np.array_equal(h1.shadow(ro, rd, 20.0), h2.shadow(ro, rd, 20.0))


def test_round_trip_of_random_scenes(built):
    # random wrapper/CSG/instance nestings: either the text is refused for one of the two documented reasons (`show` prints
    # neither vertex normals nor a mesh's default material) or the copy built from it answers every ray identically
    refusals = ("does not print the normals", "needs a default material")
    same = 0
    for gen, seeds in ((zoo.random_composites, range(30)), (zoo.random_flat, range(30))):
        for seed in seeds:
            b, root, _ = _build(gen(seed))
            text = b.show(root)
            try:
                root2, _ = b.load_show(text, b.show_tex_materials(root))
            except api.GlomeError as e:
                assert any(r in str(e) for r in refusals), (gen.__name__, seed, str(e))
                continue
            assert b.show(root2) == text, (gen.__name__, seed)
            h1, h2 = HostSim(b, root), HostSim(b, root2)
            ro, rd = random_rays(300, seed)
            r1, r2 = h1.rayint(ro, rd), h2.rayint(ro, rd)
            for k in ("t", "n", "tex"):
                assert np.array_equal(r1[k], r2[k]), (gen.__name__, seed, k)
            assert np.array_equal(h1.shadow(ro, rd, 20.0), h2.shadow(ro, rd, 20.0)), (gen.__name__, seed)
            same += 1
    assert same >= 20


def test_bih_text_is_the_builders_tree(built):
    # the tree the text spells out == glome_sb_bih_dump's arrays (split planes, axes, leaf sizes), preorder
    rng = np.random.default_rng(8)
    b = api.Builder()
    root = b.bih([b.sphere(tuple(rng.uniform(-20, 20, size=3)), float(rng.uniform(0.2, 3))) for _ in range(400)])
    tree = showfmt.parse(b.show(root))
    bihs = [c for c in showfmt.walk(tree) if c[0] == "Bih"]
    assert len(bihs) == 1
    ls, rs, ax, nl = [], [], [], []

    def pre(n):
        if n[0] == "BihLeaf":
            ls.append(0.0); rs.append(0.0); ax.append(-1); nl.append(len(n[1]))
        else:
            ls.append(n[1]); rs.append(n[2]); ax.append(n[3]); nl.append(0)
            pre(n[4]); pre(n[5])
    pre(bihs[0][2])
    d = b.bih_dump(root)
    assert np.array_equal(d[0], ls) and np.array_equal(d[1], rs) and np.array_equal(d[2], ax) and np.array_equal(d[3], nl)
    assert bihs[0][1] == ("Bbox", ("Vec",) + tuple(b.bound(root)[:3]), ("Vec",) + tuple(b.bound(root)[3:]))


def test_a_tree_given_in_the_text_is_taken_as_printed(built):
    # not rebuilt: a (legal) tree this builder would never make -- one item per leaf, split on z first
    def sph(x, z):
        return "SI Sphere (Vec %s 0.0 %s) 1.0 1.0" % (showfmt.hs_double(x), showfmt.hs_double(z))
    text = ("SI Bih {bihbb = Bbox {p1 = Vec (-1.0) (-1.0) (-1.0), p2 = Vec 5.0 1.0 5.0}, bihroot = "
            "BihBranch 1.0001 2.9999 2 (BihBranch 1.0001 2.9999 0 (BihLeaf [%s]) (BihLeaf [%s])) (BihBranch 1.0001 2.9999 0 (BihLeaf [%s]) (BihLeaf []))}"
            % (sph(0.0, 0.0), sph(4.0, 0.0), sph(0.0, 4.0)))
    b = api.Builder()
    root, ntex = b.load_show(text)
    assert ntex == 0 and b.show(root) == text
    d = b.bih_dump(root)
    assert list(d[2]) == [2, 0, -1, -1, 0, -1, -1] and list(d[3]) == [0, 0, 1, 1, 0, 1, 0]
    hs = HostSim(b, root)
    ro = np.float32([[0, 5, 0.1], [4, 5, 0.1], [0.1, 5, 4], [4, 5, 4]]); rd = np.float32([[0.01, -0.9997, 0.02]] * 4)  # off-axis (Q1: a +0.0 direction component misses the root box)
    t = hs.rayint(ro, rd)["t"]
    assert np.all(t[:3] > 3.9) and np.all(t[:3] < 4.1) and t[3] < 0


def test_flat_mesh_round_trips_and_smooth_mesh_is_refused(built):
    V = np.float64([[0, 0, 0], [1, 0, 0], [0, 0, 1], [1, 0.5, 1], [2, 0, 0], [2, 0.3, 1]])
    T = np.full((4, 8), -1, np.int32)
    T[:, :3] = [[0, 1, 2], [2, 1, 3], [1, 4, 3], [3, 4, 5]]
    b = api.Builder()
    m = b.material_surface((1, 1, 1), 1, 0.2, 0.8, 0, 0)
    T2 = T.copy(); T2[1, 6] = 0
    root = b.mesh(V, np.zeros((0, 3)), T2, [m])
    text = b.show(root)
    assert text.startswith("SI Mesh [Vec 0.0 0.0 0.0,Vec 1.0 0.0 0.0,") and "Tri 2 1 3 (-1) (-1) (-1) 0 (-1)" in text
    tree = showfmt.parse(text)
    assert tree[1][0] == "Mesh" and len(tree[1][1]) == 6 and len(tree[1][2]) == 4 and tree[1][3][0] == "Bbox"
    with pytest.raises(api.GlomeError, match="default material"):
        b.load_show(text)
    root2, _ = b.load_show(text, default_material=m)
    assert b.show(root2) == text
    ro, rd = random_rays(300, 2, center=(1, 0.2, 0.5), radius=4, spread=1)
    r1, r2 = HostSim(b, root).rayint(ro, rd), HostSim(b, root2).rayint(ro, rd)
    assert np.array_equal(r1["t"], r2["t"]) and np.array_equal(r1["n"], r2["n"]) and (r1["t"] >= 0).sum() > 30
    T3 = T.copy(); T3[0, 3:6] = [0, 1, 2]
    smooth = b.mesh(V, np.float64([[0, 1, 0]] * 6), T3, [])
    with pytest.raises(api.GlomeError, match="normals"):
        b.load_show(b.show(smooth))


def test_reader_errors(built):
    b = api.Builder()
    m = b.material_reflect(0.5)
    for text, what in [("SI Sphere (Vec 0.0 0.0 0.0) 1.0", "number"), ("SI Torus 1.0 2.0", "unknown solid"), ("SI Void SI Void", "after the scene"),
                       ("SI Difference SI Void SI Void Maybe", "True or False"), ("SI Tex SI Void Texture", "no material"),
                       ("SI Bih {bihbb = Bbox {p1 = Vec 0.0 0.0 0.0, p2 = Vec 1.0 1.0 1.0}, bihroot = BihBranch 0.5 0.5 3 (BihLeaf []) (BihLeaf [])}", "axis"),
                       ("Sphere (Vec 0.0 0.0 0.0) 1.0 1.0", "expected SI"), ("SI [SI Void,", "expected SI")]:
        with pytest.raises(api.GlomeError, match=what):
            b.load_show(text)
    root, n = b.load_show("SI Tex SI Tex SI Void Texture Texture", [m], default_material=m)
    assert n == 2 and b.show_tex_materials(root) == [m, m]
    with pytest.raises(api.GlomeError):
        b.load_show("SI Void", [99])
    assert math.isinf(showfmt.parse(b.show(b.plane_offset((0, 1, 0), float("inf"))))[1][2])
