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
