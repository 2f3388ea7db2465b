"""np_oracle.py -- an INDEPENDENT NumPy (float64) restatement of glome's primitive formulas (TEST INFRASTRUCTURE).

Written from the reference text separately from oracle/glome_oracle.hpp, vectorised over rays, so the two restatements
can be cross-checked on seeded random rays (tests/test_np_crosscheck.py).  The reference has no tests of its own
("parity unpinned"); agreement of two independent readings plus the closed-form KATs is what pins the
oracle.  Every function cites the reference file:line.  Returns t = -1 for RayMiss.
"""
import numpy as np

INF = 1000000.0  # Vec.hs:14
DELTA = 0.0001   # Vec.hs:40


def _dot(a, b):
    return (a[..., 0] * b[..., 0]) + (a[..., 1] * b[..., 1]) + (a[..., 2] * b[..., 2])  # Vec.hs:185-187


def _cross(a, b):  # Vec.hs:193-198
    return np.stack([(a[..., 1] * b[..., 2]) - (a[..., 2] * b[..., 1]), (a[..., 2] * b[..., 0]) - (a[..., 0] * b[..., 2]),
                     (a[..., 0] * b[..., 1]) - (a[..., 1] * b[..., 0])], axis=-1)


def _vnorm(a):  # Vec.hs:314-317
    inv = 1.0 / np.sqrt((a[..., 0] * a[..., 0]) + (a[..., 1] * a[..., 1]) + (a[..., 2] * a[..., 2]))
    return a * inv[..., None]


def fmin(a, b):  # Vec.hs:44-45: if a > b then b else a
    return np.where(a > b, b, a)


def fmax(a, b):  # Vec.hs:48-49
    return np.where(a > b, a, b)


def fmin3(a, b, c):  # Vec.hs:52-59
    return np.where(a > b, np.where(b > c, c, b), np.where(a > c, c, a))


def fmax3(a, b, c):  # Vec.hs:62-69
    return np.where(a > b, np.where(a > c, a, c), np.where(b > c, b, c))


def sphere_rayint(c, r, o, d, dist):
    """Sphere.hs:20-41.  Returns (t, normal)."""
    with np.errstate(invalid="ignore"):
        eo = c - o
        v = _dot(eo, d)
        disc = r * r - (_dot(eo, eo) - v * v)
        sq = np.sqrt(np.where(disc < 0, 0.0, disc))
        hit = np.where((v - sq) > 0, v - sq, v + sq)
        miss = (disc < 0) | (hit < 0) | (hit > dist)
        t = np.where(miss, -1.0, hit)
        p = o + d * hit[..., None]
        n = _vnorm(p - c)
    return t, n


def sphere_shadow(c, r, o, d, dist):
    """Sphere.hs:51-71."""
    eo = c - o
    v = _dot(eo, d)
    pre = (dist >= (v - r)) & (v > 0.0)
    t, _ = sphere_rayint(c, r, o, d, dist)
    return pre & (t >= 0)


def triangle_rayint(p1, p2, p3, o, d, dist):
    """Triangle.hs:45-73.  Returns (t, b1, b2)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        e1, e2 = p2 - p1, p3 - p1
        s1 = _cross(d, e2)
        divisor = _dot(s1, e1)
        inv = 1.0 / divisor
        dd = o - p1
        b1 = _dot(dd, s1) * inv
        s2 = _cross(dd, e1)
        b2 = _dot(d, s2) * inv
        t = _dot(e2, s2) * inv
        miss = (divisor == 0) | (b1 < 0) | (b1 > 1) | (b2 < 0) | (b1 + b2 > 1) | (t < 0) | (t > dist)
    return np.where(miss, -1.0, t), b1, b2


def _slabs(lo, hi, o, d):
    with np.errstate(divide="ignore", invalid="ignore"):
        rcp = 1.0 / d
        pos = d > 0
        tin = np.where(pos, (lo - o) * rcp, (hi - o) * rcp)
        tout = np.where(pos, (hi - o) * rcp, (lo - o) * rcp)
    return tin, tout


def bbclip(lo, hi, o, d):
    """bbclip_ub, Vec.hs:743-762 -> (near, far)."""
    tin, tout = _slabs(lo, hi, o, d)
    return fmax3(tin[..., 0], tin[..., 1], tin[..., 2]), fmin3(tout[..., 0], tout[..., 1], tout[..., 2])


def box_rayint(lo, hi, o, d, dist):
    """Box.hs:18-54.  Returns (t, normal)."""
    tin, tout = _slabs(lo, hi, o, d)
    lastin = fmax3(tin[..., 0], tin[..., 1], tin[..., 2])
    firstout = fmin3(tout[..., 0], tout[..., 1], tout[..., 2])
    miss = (lastin > firstout) | (firstout < 0) | (lastin > dist)
    inside = lastin < 0
    n = np.zeros(o.shape)
    sgn = np.where(d > 0, 1.0, -1.0)
    # origin inside: first axis (x, y, z priority) whose out == firstout, normal along +dir; z is the default
    ax_in = np.where(tout[..., 0] == firstout, 0, np.where(tout[..., 1] == firstout, 1, 2))
    ax_out = np.where(tin[..., 0] == lastin, 0, np.where(tin[..., 1] == lastin, 1, 2))
    ax = np.where(inside, ax_in, ax_out)
    s = np.take_along_axis(sgn, ax[..., None], axis=-1)[..., 0] * np.where(inside, 1.0, -1.0)
    np.put_along_axis(n, ax[..., None], s[..., None], axis=-1)
    t = np.where(inside, firstout, lastin)
    return np.where(miss, -1.0, t), n


def box_shadow(lo, hi, o, d, dist):
    """Box.hs:56-62."""
    near, far = bbclip(lo, hi, o, d)
    return ~((near > far) | (far <= 0) | (far > dist))


def plane_rayint(n, off, o, d, dist):
    """Plane.hs:27-32."""
    with np.errstate(divide="ignore", invalid="ignore"):
        hit = -((_dot(n, o) - off) / _dot(n, d))
        miss = (hit < 0) | (hit > dist)
    return np.where(miss, -1.0, hit)


def disc_rayint(point, norm, r2, o, d, dist):
    """Cone.hs:69-79 with plane_int_dist (Vec.hs:391-394)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        t = -(_dot(norm, o - point)) / (_dot(norm, d))
        pos = o + d * t[..., None]
        off = pos - point
        miss = (t < 0) | (t > dist) | (_dot(off, off) > r2)
    return np.where(miss, -1.0, t)


def _quadric(a, b, c, dist):
    """the shared root selection of Cone.hs:109-127 / 162-180: returns (dist or -1)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        disc = b * b - 4 * a * c
        ds = np.sqrt(np.where(disc < 0, 0.0, disc))
        q = np.where(b < 0, (b - ds) * (-0.5), (b + ds) * (-0.5))
        t0p, t1p = q / a, c / q
        t0, t1 = fmin(t0p, t1p), fmax(t0p, t1p)
        sel = np.where(t0 < 0, t1, t0)
        miss = (disc < 0) | (t1 < 0) | (t0 > dist) | (sel < 0) | (sel > dist)
    return np.where(miss, -1.0, sel)


def cylinder_rayint(r, h1, h2, o, d, dist):
    """Cone.hs:104-139 (z axis).  Returns (t, normal)."""
    ox, oy, oz, dx, dy, dz = o[..., 0], o[..., 1], o[..., 2], d[..., 0], d[..., 1], d[..., 2]
    sel = _quadric(dx * dx + dy * dy, 2 * (dx * ox + dy * oy), ox * ox + oy * oy - r * r, dist)
    pos = o + d * sel[..., None]
    side = (sel >= 0) & (pos[..., 2] > h1) & (pos[..., 2] < h2)
    nside = np.stack([pos[..., 0] / r, pos[..., 1] / r, np.zeros_like(sel)], axis=-1)
    up = dz > 0
    bot = disc_rayint(np.array([0, 0, h1]), np.array([0, 0, -1.0]), r * r, o, d, dist)
    top = disc_rayint(np.array([0, 0, h2]), np.array([0, 0, 1.0]), r * r, o, d, dist)
    cap_t = np.where(up, np.where(oz < h1, bot, -1.0), np.where(oz > h2, top, -1.0))
    cap_n = np.where(up[..., None], np.array([0, 0, -1.0]), np.array([0, 0, 1.0]))
    t = np.where(sel < 0, -1.0, np.where(side, sel, cap_t))
    n = np.where(side[..., None], nside, cap_n)
    return t, n


def cone_rayint(r, clip1, clip2, height, o, d, dist):
    """Cone.hs:155-200 (z axis).  Returns (t, normal)."""
    ox, oy, oz, dx, dy, dz = o[..., 0], o[..., 1], o[..., 2], d[..., 0], d[..., 1], d[..., 2]
    kp = r / height
    k = kp * kp
    a = dx * dx + dy * dy - k * dz * dz
    b = 2 * (dx * ox + dy * oy - k * dz * (oz - height))
    c = ox * ox + oy * oy - k * (oz - height) * (oz - height)
    sel = _quadric(a, b, c, dist)
    pos = o + d * sel[..., None]
    side = (sel >= 0) & (pos[..., 2] > clip1) & (pos[..., 2] < clip2)
    invhyp = 1.0 / np.sqrt(height * height + r * r)
    with np.errstate(divide="ignore", invalid="ignore"):
        corr = (height * invhyp) / np.sqrt(pos[..., 0] * pos[..., 0] + pos[..., 1] * pos[..., 1])
    nside = np.stack([pos[..., 0] * corr, pos[..., 1] * corr, np.full_like(sel, r * invhyp)], axis=-1)
    r2 = r * (1 - ((clip2 - clip1) / height))
    up = dz > 0
    bot = disc_rayint(np.array([0, 0, clip1]), np.array([0, 0, -1.0]), r * r, o, d, dist)
    top = disc_rayint(np.array([0, 0, clip2]), np.array([0, 0, 1.0]), r2 * r2, o, d, dist)
    cap_t = np.where(up, np.where(oz < clip1, bot, -1.0), np.where(oz > clip2, top, -1.0))
    cap_n = np.where(up[..., None], np.array([0, 0, -1.0]), np.array([0, 0, 1.0]))
    t = np.where(sel < 0, -1.0, np.where(side, sel, cap_t))
    n = np.where(side[..., None], nside, cap_n)
    return t, n


def instance_ray(inv, o, d, dist):
    """Solid.hs:389-395: inverse-transform, renormalise, rescale tmax.  inv = 3x4 inverse matrix."""
    no = o @ inv[:, :3].T + inv[:, 3]
    nd = d @ inv[:, :3].T
    ls = np.sqrt(_dot(nd, nd))
    return no, nd * (1 / ls)[..., None], dist * ls, ls


def cafold(c1, c2):  # Clr.hs:106-113
    trans = 1 - c1[3]
    return np.array([c1[0] + c2[0] * trans * c2[3], c1[1] + c2[1] * trans * c2[3], c1[2] + c2[2] * trans * c2[3], c1[3] + c2[3] * trans])


def caweight(c1, c2, w):  # Clr.hs:87-91
    return np.asarray(c1) * w + np.asarray(c2) * (1 - w)


def casum(cs):  # Clr.hs:93-103
    cs = np.asarray(cs, dtype=np.float64)
    rgb = (cs[:, :3] * cs[:, 3:4]).sum(axis=0)
    a = 1 - np.prod(1 - np.clip(cs[:, 3], 0, 1))
    return np.array([rgb[0], rgb[1], rgb[2], a])


def get_coords(width, height, xf, yf):  # Glome.hs:119-140
    x = (((xf / float(width)) * 2) - 1) * (float(width) / float(height))
    y = -(((yf / float(height)) * 2) - 1)
    return x, y


def rgbf(r, g, b):  # Glome.hs:98-110
    cap1 = lambda x: (1 - DELTA) if x >= 1 else x
    return (int(np.floor(cap1(r) * 256)) * 65536 + int(np.floor(cap1(g) * 256)) * 256 + int(np.floor(cap1(b) * 256))) & 0xFFFFFFFF


def chunk(size, blocksize):  # Glome.hs:371-377
    out, pos = [], 0
    while True:
        if pos + blocksize >= size:
            out.append((pos, size - pos))
            return out
        out.append((pos, blocksize))
        pos += blocksize
