"""Known-answer tests for the CPU oracle: SURVEY.md Appendix D (hand-derived from the reference formulas; the
reference itself ships no tests -- "parity unpinned").  These pin the restatement before anything is compared to it."""
import ctypes as C
import math

import numpy as np
import pytest

from helpers import O

INF = 1e6


def ray1(o, root, orig, d, dist=INF):
    r = o.rayint(root, [orig], [d], dist)
    return r["t"][0], r["pos"][0], r["n"][0], r


@pytest.fixture()
def o(built):
    return O.Oracle()


def test_D1_D4_sphere(o):
    s = o.sphere([0, 0, 0], 1)
    t, p, n, _ = ray1(o, s, [0, 0, -3], [0, 0, 1])
    assert t == pytest.approx(2, abs=1e-15) and np.allclose(p, [0, 0, -1]) and np.allclose(n, [0, 0, -1])
    # D2: origin at the centre: rayint hits the far side, shadow is False (v = 0)  (Q4)
    t, p, n, _ = ray1(o, s, [0, 0, 0], [0, 0, 1])
    assert t == pytest.approx(1) and np.allclose(p, [0, 0, 1]) and np.allclose(n, [0, 0, 1])
    assert not o.shadow(s, [[0, 0, 0]], [[0, 0, 1]], INF)[0]
    # D3 miss, D4 beyond dist
    assert ray1(o, s, [0, 2, -3], [0, 0, 1])[0] == -1
    assert ray1(o, s, [0, 0, -3], [0, 0, 1], 1.5)[0] == -1
    assert o.shadow(s, [[0, 0, -3]], [[0, 0, 1]], INF)[0]


def test_D5_plane(o):
    p = o.plane([0, 0, 0], [0, 1, 0])
    t, pos, n, _ = ray1(o, p, [0, 2, 0], [0, -1, 0])
    assert t == pytest.approx(2) and np.allclose(pos, [0, 0, 0]) and np.allclose(n, [0, 1, 0])
    assert ray1(o, p, [0, 2, 0], [1, 0, 0])[0] == -1  # hit = -inf


def test_D6_D7_triangle(o):
    tr = o.triangle([0, 0, 0], [1, 0, 0], [0, 1, 0])
    t, pos, n, _ = ray1(o, tr, [.25, .25, -1], [0, 0, 1])
    assert t == pytest.approx(1) and np.allclose(n, [0, 0, 1])  # not flipped toward the ray (Q5)
    assert ray1(o, tr, [.75, .75, -1], [0, 0, 1])[0] == -1
    tn = o.trianglenorm([0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 0], [0, 1, 0])
    t, pos, n, _ = ray1(o, tn, [.25, .25, -1], [0, 0, 1])
    assert np.allclose(n, np.array([1, 1, 2]) / math.sqrt(6), atol=1e-12)


def test_D8_D9_box(o):
    b = o.box([-1, -1, -1], [1, 1, 1])
    d = np.array([1, 1, 1]) / math.sqrt(3)
    t, pos, n, _ = ray1(o, b, [-3, -3, -3], d)
    assert t == pytest.approx(2 * math.sqrt(3), rel=1e-14) and np.allclose(pos, [-1, -1, -1]) and np.allclose(n, [-1, 0, 0])  # corner picks x (Q6)
    # D8b: +0 direction components miss, -0 hit (Q1)
    assert ray1(o, b, [0, 0, -3], [0.0, 0.0, 1])[0] == -1
    t, pos, n, _ = ray1(o, b, [0, 0, -3], [-0.0, -0.0, 1])
    assert t == pytest.approx(2) and np.allclose(pos, [0, 0, -1]) and np.allclose(n, [0, 0, -1])
    # D9: origin inside -> exit hit
    t, pos, n, _ = ray1(o, b, [0, 0, 0], d)
    assert t == pytest.approx(math.sqrt(3)) and np.allclose(pos, [1, 1, 1]) and np.allclose(n, [1, 0, 0])


def test_D10_instance(o):
    from glome_amd import api
    s = o.sphere([0, 0, 0], 1)
    inst = o.transform(s, [api.scale([2, 2, 2]), api.translate([5, 0, 0])])
    t, pos, n, _ = ray1(o, inst, [5, 0, -10], [0, 0, 1])
    assert t == pytest.approx(8) and np.allclose(pos, [5, 0, -2]) and np.allclose(n, [0, 0, -1])


def test_D11_difference_texture_loss(o):
    m = o.material_surface([1, 1, 1], 1, 0.2, 0.8, 0, 0)
    d = o.difference(o.sphere([0, 0, 0], 2), o.sphere([0, 0, -2], 1))
    td = o.tex(d, m)
    t, pos, n, r = ray1(o, td, [0, 0, -10], [0, 0, 1])
    assert t == pytest.approx(9, abs=1e-9) and np.allclose(pos, [0, 0, -1], atol=1e-9) and np.allclose(n, [0, 0, -1])
    assert r["ntex"][0] == 0  # the carved surface takes get_metainfo of A: no textures (Q13)
    # texturing A instead keeps the texture on the carved surface
    d2 = o.difference(o.tex(o.sphere([0, 0, 0], 2), m), o.sphere([0, 0, -2], 1))
    r2 = o.rayint(d2, [[0, 0, -10]], [[0, 0, 1]])
    assert r2["ntex"][0] == 1 and r2["tex"][0][0] == m


def test_D12_intersection(o):
    x = o.intersection([o.sphere([-1, 0, 0], 2), o.sphere([1, 0, 0], 2)])
    t, pos, n, _ = ray1(o, x, [.5, 0, -10], [0, 0, 1])
    assert t == pytest.approx(10 - math.sqrt(1.75), rel=1e-12)
    assert np.allclose(pos, [.5, 0, -math.sqrt(1.75)]) and np.allclose(n, [.75, 0, -math.sqrt(1.75) / 2], atol=1e-12)


def test_D13_bih_equals_nearest(o):
    rng = np.random.default_rng(5)
    ids = [o.sphere(rng.uniform(-3, 3, 3), rng.uniform(0.2, 0.8)) for _ in range(40)]
    b = o.bih(ids)
    g = o.group(ids)
    ro = rng.uniform(-8, 8, (500, 3)); rd = rng.normal(size=(500, 3)); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    a, c = o.rayint(b, ro, rd), o.rayint(g, ro, rd)
    assert np.array_equal(a["t"], c["t"]) and np.array_equal(a["prim"], c["prim"])
    assert np.array_equal(o.shadow(b, ro, rd, 9.0), o.shadow(g, ro, rd, 9.0))


def test_D14_getcoords(built):
    L = O.lib()
    out = np.zeros(2)
    for (px, py), want in {(0, 0): (-1.5, 1), (360, 240): (0, -0.0), (719, 479): (1.4958333333333333, -0.9958333333333333)}.items():
        L.glo_getcoords(C.c_int(720), C.c_int(480), C.c_double(px), C.c_double(py), out.ctypes.data_as(O.c_dp))
        assert np.allclose(out, want, atol=1e-15)


def test_D15_surface_shading(o):
    m = o.material_surface([1, 1, 1], 1, 0.2, 0.8, 0.4, 10)
    o.set_root(o.tex(o.plane([0, 0, 0], [0, 1, 0]), m))
    o.add_light([0, 10, 0], [100, 100, 100])
    # camera straight down from (0,5,0): fwd = (0,-1,0); centre pixel of a 2x2... use a 1x1 image whose pixel (0,0) maps to x=-1,y=1;
    # make up/right zero so every pixel looks along fwd
    o.set_camera_vectors([0, 5, 0], [0, -1, 0], [0, 0, 0], [0, 0, 0])
    img, _, cnt = o.render(1, 1, maxdepth=3, want_packed=False)
    assert np.allclose(img[0, 0, :4], [1.4, 1.4, 1.4, 1.0], atol=1e-12) and img[0, 0, 4] == pytest.approx(5)
    assert cnt["rays_shadow"] == 1


def test_D16_D17_colour(built):
    L = O.lib()
    out = np.zeros(4)
    a, b = np.array([0, 0, 0, 0.]), np.array([.3, .4, .5, .6])
    L.glo_cafold(a.ctypes.data_as(O.c_dp), b.ctypes.data_as(O.c_dp), out.ctypes.data_as(O.c_dp))
    assert np.allclose(out, [.3 * .6, .4 * .6, .5 * .6, .6])
    a, b = np.array([.5, 0, 0, .5]), np.array([0, 1, 0, 1.])
    L.glo_cafold(a.ctypes.data_as(O.c_dp), b.ctypes.data_as(O.c_dp), out.ctypes.data_as(O.c_dp))
    assert np.allclose(out, [.5, .5, 0, 1])
    assert L.glo_rgbf(C.c_double(1.4), C.c_double(0.5), C.c_double(0)) == 16744448


def test_D18_chunk(built):
    L = O.lib()
    want = {720: (12, (715, 5)), 480: (8, (455, 25)), 1920: (30, (1885, 35)), 1080: (17, (1040, 40)), 3840: (60, (3835, 5)), 2160: (34, (2145, 15))}
    buf = np.zeros(200, np.int32)
    for n, (cnt, last) in want.items():
        k = L.glo_chunk(C.c_int(n), C.c_int(65), buf.ctypes.data_as(O.c_ip), C.c_int(100))
        assert k == cnt and (buf[2 * (k - 1)], buf[2 * (k - 1) + 1]) == last and all(buf[2 * i + 1] == 65 for i in range(k - 1))


def test_D19_reflect(built):
    L = O.lib()
    out = np.zeros(3)
    v, n = np.array([0, -1., 0]), np.array([0, 1., 0])
    L.glo_reflect(v.ctypes.data_as(O.c_dp), n.ctypes.data_as(O.c_dp), out.ctypes.data_as(O.c_dp))
    assert np.allclose(out, [0, 1, 0])
    v = np.array([1, -1, 0]) / math.sqrt(2)
    L.glo_reflect(v.ctypes.data_as(O.c_dp), n.ctypes.data_as(O.c_dp), out.ctypes.data_as(O.c_dp))
    assert np.allclose(out, np.array([1, 1, 0]) / math.sqrt(2))


def test_D20_camera(built):
    cv = O.camera_vectors([-2, 4.3, 15], [0, 2, 0], [0, 1, 0], 45)
    pos, fwd, up, right = cv
    t = math.tan(math.radians(22.5))
    assert np.linalg.norm(fwd) == pytest.approx(1) and np.linalg.norm(up) == pytest.approx(t) and np.linalg.norm(right) == pytest.approx(t)
    assert abs(fwd @ up) < 1e-12 and abs(fwd @ right) < 1e-12 and abs(up @ right) < 1e-12


def test_reference_invariants(o):
    """The reference's own constructor-time checks: they `error` out."""
    from glome_amd import api
    with pytest.raises(O.OracleError):
        o.bih([o.sphere([0, 0, 0], 1), o.plane([0, 0, 0], [0, 1, 0])])  # infinite bound inside a bih (Bih.hs:319-322)
    with pytest.raises(api.GlomeError):
        api.rotate([0, 0, 2.0], 0.3)  # unnormalised axis (Vec.hs:577-580)
    bad = np.concatenate([np.eye(3, 4).ravel(), 2 * np.eye(3, 4).ravel()])
    with pytest.raises(O.OracleError):
        o.transform(o.sphere([0, 0, 0], 1), [bad])  # fwd * inv != I (check_xfm, Vec.hs:466-477)


def test_trace_recursion_and_empty_stack(o):
    """Q16: recurs == 0 and an empty texture stack both give transparent; Reflect alpha = child alpha * refl (Q17)."""
    mir = o.material_reflect(0.8)
    matte = o.material_surface([1, 0, 0], 1, 1.0, 0, 0, 0)  # ambient only: colour = (1,0,0)
    floor = o.tex(o.plane([0, 0, 0], [0, 1, 0]), mir)
    ceil = o.tex(o.plane_offset([0, -1, 0], -4), matte)  # plane y = 4 facing down
    o.set_root(o.group([floor, ceil]))
    o.set_camera_vectors([0, 2, 0], [0, -1, 0], [0, 0, 0], [0, 0, 0])
    img, _, c = o.render(1, 1, maxdepth=2, want_packed=False)
    # child = (1,0,0,1); Reflect -> (1,0,0,0.8); the fold is cafold (0,0,0,0) that = (r*a, g*a, b*a, a)  (Clr.hs:106-113)
    assert np.allclose(img[0, 0, :4], [0.8, 0, 0, 0.8]) and c["rays_secondary"] == 1
    img, _, c = o.render(1, 1, maxdepth=1, want_packed=False)  # the bounce is trace ... 0 = transparent
    assert np.allclose(img[0, 0, :4], [0, 0, 0, 0]) and c["rays_secondary"] == 0
    o.set_root(o.plane([0, 0, 0], [0, 1, 0]))  # hit with an empty texture stack
    img, _, _ = o.render(1, 1, maxdepth=3, want_packed=False)
    assert np.allclose(img[0, 0, :4], [0, 0, 0, 0]) and img[0, 0, 4] == pytest.approx(2)


def test_warp_known_answers(o):
    """Warp frame scene' lights' xfm (Shader.hs:157-175), hand-derived.  A wall at z = 0 textured with Warp, seen from
    (0, 0, -5) looking along +z, inside `transform [translate (10, 0, 0)]` so that the hit's own ray (riray) is LOCAL:
      * the frame is traced with the local ray (o = (-10, 0, -5) + ..., d = +z): a red ambient-only sphere at the LOCAL origin
        (0, 0, 3) lies on it, 8 - 1 = 7 away; the world ray would miss it by 10 -- so a red pixel proves riray is local;
      * the other scene is traced from xfm_ray M (Ray pos (vnorm dir)) with M = translate (0, 100, 0): origin (10, 100, 0),
        direction +z, up to the frame's depth: a blue sphere at (10, 100, 4) is 3 away -> nearer than the frame (7): blue;
        moved to (10, 100, 12) it is 11 away -> beyond the frame's depth: the frame's red stays.
      * recurs: Warp's traces run with recurs - 1; at maxdepth 1 both are traceMiss: transparent."""
    from glome_amd import api
    red = o.material_surface([1, 0, 0], 1, 1.0, 0, 0, 0)
    blue = o.material_surface([0, 0, 1], 1, 1.0, 0, 0, 0)
    frame = o.tex(o.sphere([0, 0, 3], 1), red)                     # local coordinates of the wall
    for zs, want in ((4.0, [0, 0, 1, 1]), (12.0, [1, 0, 0, 1])):
        other = o.tex(o.sphere([10, 100, zs], 1), blue)
        warp = o.material_warp(frame, other, [], api.translate((0, 100, 0)))
        wall = o.transform(o.tex(o.plane([0, 0, 0], [0, 0, -1]), warp), [api.translate((10, 0, 0))])  # (a plane: an axis-parallel ray misses a Box, Q1)
        o.set_root(wall)
        o.set_camera_vectors([10, 0, -5], [0, 0, 1], [0, 0, 0], [0, 0, 0])
        img, _, c = o.render(1, 1, maxdepth=2, want_packed=False)
        assert np.allclose(img[0, 0, :4], want) and img[0, 0, 4] == pytest.approx(5) and c["rays_secondary"] == 2, (zs, img[0, 0], c)
        img, _, c = o.render(1, 1, maxdepth=1, want_packed=False)
        assert np.allclose(img[0, 0, :4], [0, 0, 0, 0]) and c["rays_secondary"] == 0


# ------------------------------------------------------------------ solid texture functions (GlomeVec Texture.hs)
def _weight(fn, params, pts):
    import ctypes as C
    from oracle import oracle_py as O
    lib = O.lib()
    pts = np.ascontiguousarray(np.asarray(pts, np.float64).reshape(-1, 3))
    out = np.zeros(len(pts))
    wp = (C.c_double * 4)(*([float(x) for x in params] + [0.0] * (4 - len(params))))
    lib.glo_weight_fn(C.c_int(fn), wp, C.c_int(len(pts)), pts.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def test_perlin_is_half_on_the_lattice_and_stays_in_range():
    # at a lattice point every knot vanishes: the own knot has v = 0 (vdot gamma 0 = 0), the others have an omega(+-1) = 0
    lattice = [(i, j, k) for i in (-3, 0, 2) for j in (-1, 0, 5) for k in (-2, 0, 1)]
    assert np.allclose(_weight(1, [1.0], lattice), 0.5, atol=1e-15)
    rng = np.random.default_rng(7)
    p = rng.uniform(-20, 20, (4000, 3))
    w = _weight(1, [3.0], p)
    assert w.min() >= 0 and w.max() <= 1 and 0.3 < w.mean() < 0.7 and w.std() > 0.05


def test_perlin_matches_an_independent_numpy_restatement():
    phi = [3, 0, 2, 7, 4, 1, 5, 11, 8, 10, 9, 6]
    grad = [np.array((x, y, z), float) for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1) if 1.1 < np.sqrt(x * x + y * y + z * z) < 1.5]
    assert len(grad) == 12

    def omega(t):
        t = abs(t)
        return -6 * t**5 + 15 * t**4 - 10 * t**3 + 1

    def knot(i, j, k, v):
        a = phi[abs(k) % 12]; b = phi[abs(j + a) % 12]; c = phi[abs(i + b) % 12]
        return omega(v[0]) * omega(v[1]) * omega(v[2]) * float(np.dot(grad[c], v))

    def noise(p):
        i, j, k = (int(np.floor(c)) for c in p)
        u, v, w = p[0] - i, p[1] - j, p[2] - k
        return sum(knot(i + a, j + b, k + c, np.array((u - a, v - b, w - c))) for a in (0, 1) for b in (0, 1) for c in (0, 1))

    rng = np.random.default_rng(11)
    pts = rng.uniform(-9, 9, (200, 3))
    want = np.array([(noise(2.5 * p) + 1) * 0.5 for p in pts])
    assert np.allclose(_weight(1, [2.5], pts), want, atol=1e-12)


def test_stripe_waves():
    x = np.array([0.0, 0.25, 0.5, 0.75, 1.25, -0.25])
    pts = np.stack([x, np.zeros_like(x), np.zeros_like(x)], 1)
    assert np.allclose(_weight(2, [1, 0, 0], pts), [0, 0, 1, 1, 0, 1])              # square_wave: offset < 0.5 -> 0
    assert np.allclose(_weight(3, [1, 0, 0], pts), [0, 0.5, 1, 0.5, 0.5, 0.5])      # triangle_wave
    assert np.allclose(_weight(4, [1, 0, 0], pts), np.sin(x * 2 * np.pi) * 0.5 + 0.5)
    assert np.allclose(_weight(3, [4, 8, 5], [[0.1, 0.2, 0.3]]), [1.0])             # vdot = 3.5 -> offset 0.5 -> 2 - 1
