"""Two independent restatements of the reference formulas -- C++ (oracle/glome_oracle.hpp) and NumPy
(oracle/np_oracle.py) -- must agree on seeded random rays.  This is the second leg that pins the oracle (the first
is tests/test_oracle_kat.py); the reference ships no vectors of its own."""
import numpy as np
import pytest

from helpers import O
from oracle import np_oracle as NP

N = 4000


def rays(seed, spread=3.0):
    rng = np.random.default_rng(seed)
    o = rng.uniform(-spread, spread, (N, 3))
    d = rng.normal(size=(N, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dist = rng.uniform(0.5, 12, N)
    return o, d, dist


def agree(t_cpp, t_np, rel=1e-11):
    assert np.array_equal(t_cpp >= 0, t_np >= 0)
    h = t_cpp >= 0
    assert np.allclose(t_cpp[h], t_np[h], rtol=rel, atol=1e-12)
    return h


@pytest.fixture()
def o(built):
    return O.Oracle()


def test_sphere(o):
    ro, rd, dist = rays(1)
    c, r = np.array([0.3, -0.2, 0.5]), 1.7
    s = o.sphere(c, r)
    got = o.rayint(s, ro, rd, dist)
    t, n = NP.sphere_rayint(c, r, ro, rd, dist)
    h = agree(got["t"], t)
    assert np.allclose(got["n"][h], n[h], atol=1e-11)
    assert np.array_equal(o.shadow(s, ro, rd, dist), NP.sphere_shadow(c, r, ro, rd, dist))
    assert h.sum() > 200


def test_triangle(o):
    ro, rd, dist = rays(2)
    p = np.array([[-2, -1, 0.2], [2.5, -0.5, -0.3], [0.1, 2.2, 0.4]])
    tr = o.triangle(*p)
    got = o.rayint(tr, ro, rd, dist)
    t, b1, b2 = NP.triangle_rayint(p[0], p[1], p[2], ro, rd, dist)
    h = agree(got["t"], t)
    assert np.array_equal(o.shadow(tr, ro, rd, dist), t >= 0) and h.sum() > 100


def test_box(o):
    ro, rd, dist = rays(3)
    lo, hi = np.array([-1, -0.5, -1.5]), np.array([0.8, 1.2, 0.7])
    b = o.box(lo, hi)
    got = o.rayint(b, ro, rd, dist)
    t, n = NP.box_rayint(lo, hi, ro, rd, dist)
    h = agree(got["t"], t)
    assert np.array_equal(got["n"][h], n[h])
    assert np.array_equal(o.shadow(b, ro, rd, dist), NP.box_shadow(lo, hi, ro, rd, dist))
    inside = np.all((ro > lo) & (ro < hi), axis=1)
    assert np.array_equal(o.inside(b, ro), inside) and inside.sum() > 20 and h.sum() > 300


def test_plane_and_disc(o):
    ro, rd, dist = rays(4)
    n = np.array([0.2, 0.9, -0.1]); n /= np.linalg.norm(n)
    pl = o.plane_offset(n, 0.4)
    agree(o.rayint(pl, ro, rd, dist)["t"], NP.plane_rayint(n, 0.4, ro, rd, dist))
    pt = np.array([0.5, 0.1, -0.3])
    ds = o.disc(pt, n, 1.3)
    agree(o.rayint(ds, ro, rd, dist)["t"], NP.disc_rayint(pt, n, 1.3 * 1.3, ro, rd, dist))


def test_cylinder_and_cone_through_instance(o):
    """cylinder / cone are z-axis primitives inside an Instance (Cone.hs:40-67); check against the NumPy canonical
    forms fed with the NumPy inverse-transformed ray."""
    from glome_amd import api
    ro, rd, dist = rays(5, spread=4.0)
    p1, p2, r = np.array([0.5, -1.0, 0.2]), np.array([1.5, 2.0, -0.8]), 0.7
    cyl = o.cylinder(p1, p2, r)
    axis = p2 - p1; ln = np.linalg.norm(axis); ax1 = axis / ln
    x = np.array([1.0, 0, 0]); y = np.array([0, 1.0, 0])
    ax2 = np.cross(ax1, x) if abs(ax1 @ x) < 0.8 else np.cross(ax1, y)
    ax2 /= np.linalg.norm(ax2); ax3 = np.cross(ax1, ax2)
    xf = api.compose([api.xyz_to_uvw(ax2, ax3, ax1), api.translate(p1)])
    inv = xf[12:].reshape(3, 4)
    lo, ld, ldist, ls = NP.instance_ray(inv, ro, rd, dist)
    t, _ = NP.cylinder_rayint(r, 0.0, ln, lo, ld, ldist)
    t_world = np.where(t >= 0, t / ls, -1.0)
    h = agree(o.rayint(cyl, ro, rd, dist)["t"], t_world, rel=1e-9)
    assert h.sum() > 100
    r1, r2 = 0.9, 0.3
    cone = o.cone(p1, r1, p2, r2)
    height = (r1 * ln) / (r1 - r2)
    t, _ = NP.cone_rayint(r1, 0.0, ln, height, lo, ld, ldist)
    t_world = np.where(t >= 0, t / ls, -1.0)
    h = agree(o.rayint(cone, ro, rd, dist)["t"], t_world, rel=1e-9)
    assert h.sum() > 100


def test_colour_pixel_helpers(built):
    import ctypes as C
    L = O.lib()
    rng = np.random.default_rng(6)
    out = np.zeros(4)
    for _ in range(50):
        a, b = rng.uniform(0, 1, 4), rng.uniform(0, 1.5, 4)
        L.glo_cafold(a.ctypes.data_as(O.c_dp), b.ctypes.data_as(O.c_dp), out.ctypes.data_as(O.c_dp))
        assert np.allclose(out, NP.cafold(a, b), rtol=1e-14)
        w = rng.uniform(0, 1)
        L.glo_caweight(a.ctypes.data_as(O.c_dp), b.ctypes.data_as(O.c_dp), C.c_double(w), out.ctypes.data_as(O.c_dp))
        assert np.allclose(out, NP.caweight(a, b, w), rtol=1e-14)
        cs = rng.uniform(-0.2, 1.3, (3, 4))
        L.glo_casum(np.ascontiguousarray(cs).ctypes.data_as(O.c_dp), C.c_int(3), out.ctypes.data_as(O.c_dp))
        assert np.allclose(out, NP.casum(cs), rtol=1e-13)
        r, g, bb = rng.uniform(-0.1, 1.6, 3)
        assert L.glo_rgbf(C.c_double(r), C.c_double(g), C.c_double(bb)) == NP.rgbf(r, g, bb)
    buf = np.zeros(400, np.int32)
    for size in (1, 64, 65, 66, 130, 720, 1919):
        k = L.glo_chunk(C.c_int(size), C.c_int(65), buf.ctypes.data_as(O.c_ip), C.c_int(200))
        assert [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(k)] == NP.chunk(size, 65)


# ------------------------------------------------------------------ composites and whole frames (oracle/np_scene.py)
def _frames(sd, w, h, maxdepth):
    """The same scene through the C++ oracle and through the independent Python restatement, same fp32-rounded camera."""
    from helpers import oracle_for, product_camera_lights
    from oracle import np_scene as NS
    o, om, _ = oracle_for(sd)
    ref, _, rc = o.render(w, h, maxdepth=maxdepth, want_packed=False)
    sc, nm = NS.load(sd)
    cam, _ = product_camera_lights(sd)
    sc.set_camera_vectors([float(x) for x in cam.pos], [float(x) for x in cam.fwd], [float(x) for x in cam.up], [float(x) for x in cam.right])
    got = np.array(sc.render(w, h, maxdepth))
    return ref, rc, got, sc


_FRAMES = [("S1", 48, 32, 1), ("S3small", 40, 24, 1), ("S3mesh_small", 40, 24, 1), ("S4", 48, 27, 3), ("csg", 40, 30, 3), ("flat_mixed", 40, 30, 3),
           ("materials", 40, 30, 3), ("mesh", 40, 30, 3), ("mirror_terrain", 40, 30, 3), ("nested", 40, 30, 3), ("portal", 56, 42, 3), ("quadrics", 40, 30, 3), ("retexture", 48, 30, 3),
           ("testscene", 64, 48, 3), ("textures", 40, 30, 3)]


@pytest.mark.parametrize("name,w,h,md", _FRAMES)
def test_whole_frames_of_two_independent_restatements_agree(built, name, w, h, md):
    """The benchmark scenes at small sizes and EVERY scene of the zoo (tests/zoo.py: all primitive families incl. smooth
    triangles and discs, cylinders / cones through `orth` + `xyz_to_uvw`, Tex stacks, NoShadow / OnlyShadow, groups, Instances of
    instances, Difference / Intersection over primitives and over an instanced bih with `inside_bih` / `get_metainfo_bih`,
    Bound / InnerBound, a Mesh with vertex normals and per-triangle textures, Reflect / Refract / Blend / AdditiveLayers, the
    perlin and stripe solid textures, Warp with `riray` and its own lights, GlomeView's default scene): the C++ oracle and
    oracle/np_scene.py -- written separately from the Haskell text -- give the same frame to rounding, pixel for pixel, and
    trace the same number of rays."""
    import zoo
    from glome_amd import scenes
    mk = {"S1": lambda: scenes.s1(nlights=2), "S3small": lambda: scenes.s3(12), "S3mesh_small": lambda: scenes.s3(12, as_mesh=True), "S4": scenes.s4,
          "testscene": lambda: zoo.testscene(3)}.get(name) or zoo.ALL[name]
    ref, rc, got, sc = _frames(mk(), w, h, md)
    assert got.shape == ref.shape
    assert np.array_equal(got[..., 4] < 1e6, ref[..., 4] < 1e6)          # the same pixels hit
    assert np.allclose(got[..., 4], ref[..., 4], rtol=1e-11, atol=1e-11)  # at the same depth
    assert np.allclose(got[..., :4], ref[..., :4], rtol=1e-10, atol=1e-12), float(np.abs(got[..., :4] - ref[..., :4]).max())
    assert sc.rays == [rc["rays_primary"], rc["rays_shadow"], rc["rays_secondary"]]


@pytest.mark.parametrize("name,w,h,bs,md", [("S1", 75, 40, 65, 1), ("S3small", 37, 29, 16, 1), ("S4", 40, 22, 65, 3)])
def test_adaptive_sampler_of_two_independent_restatements_agrees(built, name, w, h, bs, md):
    """renderTileSubsample over renderTiles' tile map (Glome.hs:226-323, 371-386) twice: the C++ oracle's pass loops against
    np_scene.render_subsample, written from the Haskell text in its own shape (one dictionary per tile, getc / putc, the five
    loops in the reference's order).  Same pixels to rounding, same number of samples traced -- one frame wider than a tile
    (clipped second tile column), one of several small ragged tiles, one with reflections."""
    from helpers import oracle_for, product_camera_lights
    from oracle import np_scene as NS
    from glome_amd import scenes
    sd = {"S1": lambda: scenes.s1(nlights=2), "S3small": lambda: scenes.s3(12), "S4": scenes.s4}[name]()
    o, om, _ = oracle_for(sd)
    ref, _, rc = o.render(w, h, maxdepth=md, mode=1, blocksize=bs, want_packed=False)
    sc, nm = NS.load(sd)
    cam, _ = product_camera_lights(sd)
    sc.set_camera_vectors([float(x) for x in cam.pos], [float(x) for x in cam.fwd], [float(x) for x in cam.up], [float(x) for x in cam.right])
    got = np.array(sc.render_subsample(w, h, md, blocksize=bs))
    assert got.shape == ref.shape
    assert sc.rays == [rc["rays_primary"], rc["rays_shadow"], rc["rays_secondary"]]  # the same contrast decisions everywhere
    fin = np.isfinite(ref[..., 4]) & (ref[..., 4] < 1e5)
    assert np.array_equal(np.isfinite(got[..., 4]) & (got[..., 4] < 1e5), fin)
    assert np.allclose(got[..., :4], ref[..., :4], rtol=1e-10, atol=1e-12), float(np.abs(got[..., :4] - ref[..., :4]).max())
    assert np.allclose(got[..., 4][fin], ref[..., 4][fin], rtol=1e-10)


@pytest.mark.parametrize("seed", list(range(10)))
def test_random_composite_scenes_through_both_restatements(built, seed):
    """zoo.random_composites(seed) -- random nestings of every construct -- through the C++ oracle and oracle/np_scene.py: the
    same frame (to rounding) and the same ray counts."""
    import zoo
    ref, rc, got, sc = _frames(zoo.random_composites(seed), 64, 36, 3)
    assert np.array_equal(got[..., 4] < 1e6, ref[..., 4] < 1e6)
    assert np.allclose(got[..., :4], ref[..., :4], rtol=1e-10, atol=1e-12), float(np.abs(got[..., :4] - ref[..., :4]).max())
    assert sc.rays == [rc["rays_primary"], rc["rays_shadow"], rc["rays_secondary"]]


def test_bih_trees_of_two_independent_builders_are_the_same(built):
    """build_rec (Bih.hs:211-285) twice: the oracle's tree (through the product's host builder, which test_host_builder pins
    to the oracle's) against np_scene's, node for node -- split planes, axes, leaf contents in order."""
    from glome_amd import api, scenes
    from oracle import np_scene as NS
    import zoo
    for sd in (scenes.s1(nlights=1), scenes.s3(10), zoo.soup(150, 3, floor=False)):
        sc, nm = NS.load(sd)
        b = api.Builder()
        pm, _ = sd.replay(b)
        # the scene's bih node: the last `bih` op
        nid, bih_sd = 0, None
        for kind, opname, args in sd.ops:
            if kind == "N": nid += args[0].shape[0]
            elif kind == "n":
                if opname == "bih": bih_sd = nid
                nid += 1
        ls, rs, ax, nl, lp = b.bih_dump(pm[bih_sd])
        tree = sc.nodes[nm[bih_sd]].root
        # preorder walk of np_scene's tree in the dump's order
        inv = {pm[i]: i for i in range(len(pm))}
        k = [0]; off = [0]
        def walk(n):
            i = k[0]; k[0] += 1
            if n[0] == "leaf":
                assert ax[i] == -1 and nl[i] == len(n[1]), (i, n[1])
                items = [inv[int(x)] for x in lp[off[0]:off[0] + nl[i]]]
                assert items == [sc_uid_to_sd[s.uid] for s in n[1]]
                off[0] += nl[i]
            else:
                assert ax[i] == n[3] and ls[i] == n[1] and rs[i] == n[2], (i, ax[i], n[3], ls[i], n[1])
                walk(n[4]); walk(n[5])
        sc_uid_to_sd = {nm[i]: i for i in range(len(nm))}
        walk(tree)
        assert k[0] == len(ax)
