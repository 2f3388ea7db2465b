"""np_scene.py -- a SECOND, independent restatement of the composite half of glome's hot path (TEST INFRASTRUCTURE).

oracle/glome_oracle.hpp (C++) is the oracle the GPU path is checked against; the reference ships no tests, so nothing pins
that restatement but closed-form KATs and -- for the primitive formulas -- the NumPy restatement in np_oracle.py.  This
module extends the second reading to the rows that matter most and had only one: the BIH builder and its two traversals,
the Mesh BVH builder and traversal, Instance / list / Difference / Intersection, and trace / mpreshade / mpostshade, up to
whole frames (tests/test_np_crosscheck.py asserts frame equality with the C++ oracle on S1 / S3-small / S4 / mesh scenes).

It was written from the Haskell text, function by function in the Haskell's own recursive shape (tuples, folds, no explicit
stacks), in plain Python floats (IEEE double, like `type Flt = Double`, Vec.hs:9) -- deliberately unlike the C++ oracle's
classes and loops, so that a misreading shared by both is unlikely.  It is slow (a 48x27 frame takes seconds); it only has
to be right.  Every function cites the reference file:line.

Backend protocol: the constructor names of glome_amd/scene.py SceneDesc.replay().
"""
import math

INF = 1000000.0  # infinity, Vec.hs:12-14
DELTA = 0.0001   # delta, Vec.hs:40
NAN = float("nan")


# ----------------------------------------------------------------------------------------------- IEEE helpers
def fdiv(a, b):
    """a / b as IEEE does it (Haskell's `/` on Double): x/0 = +-inf, 0/0 = NaN."""
    try:
        return a / b
    except ZeroDivisionError:
        if a == 0 or a != a:
            return NAN
        return math.copysign(math.inf, a) * math.copysign(1.0, b)


def fpow(a, b):
    """Haskell's (**) on Double: NaN where the real power does not exist."""
    try:
        return math.pow(a, b)
    except (ValueError, OverflowError):
        return NAN if a < 0 else math.inf


def fmin(a, b): return b if a > b else a  # Vec.hs:44-45
def fmax(a, b): return a if a > b else b  # Vec.hs:48-49
def fmin3(a, b, c): return (c if b > c else b) if a > b else (c if a > c else a)  # Vec.hs:52-59
def fmax3(a, b, c): return (a if a > c else c) if a > b else (b if b > c else c)  # Vec.hs:62-69


# ----------------------------------------------------------------------------------------------- Vec (Vec.hs:105-342)
def vadd(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])
def vsub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])
def vscale(a, f): return (a[0] * f, a[1] * f, a[2] * f)
def vdot(a, b): return (a[0] * b[0]) + (a[1] * b[1]) + (a[2] * b[2])          # Vec.hs:185-187
def vcross(a, b): return ((a[1] * b[2]) - (a[2] * b[1]), (a[2] * b[0]) - (a[0] * b[2]), (a[0] * b[1]) - (a[1] * b[0]))  # :193-198
def vscaleadd(a, b, f): return (a[0] + (b[0] * f), a[1] + (b[1] * f), a[2] + (b[2] * f))  # :302-306
def vlen(a): return math.sqrt(vdot(a, a))
def vinvert(a): return (-a[0], -a[1], -a[2])
def vnorm(a):  # Vec.hs:314-317
    inv = fdiv(1.0, math.sqrt((a[0] * a[0]) + (a[1] * a[1]) + (a[2] * a[2])))
    return (a[0] * inv, a[1] * inv, a[2] * inv)
def reflect(v, n): return vscaleadd(v, n, (-2.0) * vdot(v, n))  # Vec.hs:340-342
def bisect(a, b): return vnorm(vadd(a, b))                        # Vec.hs:331-332
def vset(v, axis, x): return tuple(x if k == axis else v[k] for k in range(3))


# ----------------------------------------------------------------------------------------------- Xfm (Vec.hs:407-560)
def xfm_point(m, v): return tuple(m[4 * r] * v[0] + m[4 * r + 1] * v[1] + m[4 * r + 2] * v[2] + m[4 * r + 3] for r in range(3))  # :502-509
def xfm_vec(m, v): return tuple(m[4 * r] * v[0] + m[4 * r + 1] * v[1] + m[4 * r + 2] * v[2] for r in range(3))                   # :522-529
def xfm_tvec(m, v): return tuple(m[r] * v[0] + m[4 + r] * v[1] + m[8 + r] * v[2] for r in range(3))  # invxfm_norm: transpose of the inverse, :543-550


def mat_mult(a, b):  # 3x4 affine product a . b (apply b first)
    out = []
    for r in range(3):
        for c in range(4):
            out.append(a[4 * r] * b[c] + a[4 * r + 1] * b[4 + c] + a[4 * r + 2] * b[8 + c] + (a[4 * r + 3] if c == 3 else 0.0))
    return tuple(out)


def compose(xfms):
    """compose [A, B] applies A then B (Vec.hs:461-462): forward = B.A, inverse = A^-1.B^-1.  xfms: 24 doubles each."""
    f = (1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0)
    i = f
    for x in xfms:
        xf, xi = tuple(float(v) for v in x[:12]), tuple(float(v) for v in x[12:24])
        f = mat_mult(xf, f)
        i = mat_mult(i, xi)
    return f, i


# ----------------------------------------------------------------------------------------------- Bbox (Vec.hs:646-762)
EMPTY_BB = ((INF, INF, INF), (-INF, -INF, -INF))       # :706-709
EVERYTHING_BB = ((-INF, -INF, -INF), (INF, INF, INF))   # :712-715
def bbjoin(a, b): return (tuple(fmin(a[0][k], b[0][k]) for k in range(3)), tuple(fmax(a[1][k], b[1][k]) for k in range(3)))    # :652-654
def bboverlap(a, b): return (tuple(fmax(a[0][k], b[0][k]) for k in range(3)), tuple(fmin(a[1][k], b[1][k]) for k in range(3)))  # :657-659
def bbmid(bb): return vscale(vadd(bb[0], bb[1]), 0.5)
def bbsa(bb):  # :694-697
    d = vsub(bb[1], bb[0])
    return max(0.0, 2 * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]))
def bbpts(pts):  # :676-690 (recursion from the back of the list)
    if not pts:
        return EMPTY_BB
    x, y, z = pts[-1]
    lo, hi = (x - DELTA, y - DELTA, z - DELTA), (x + DELTA, y + DELTA, z + DELTA)
    for (x, y, z) in reversed(pts[:-1]):
        lo = (fmin(x - DELTA, lo[0]), fmin(y - DELTA, lo[1]), fmin(z - DELTA, lo[2]))
        hi = (fmax(x + DELTA, hi[0]), fmax(y + DELTA, hi[1]), fmax(z + DELTA, hi[2]))
    return (lo, hi)
def bbclip_ub(o, d, bb):  # :743-762: divides by the direction, branches on d > 0 (Q1)
    ins, outs = [], []
    for k in range(3):
        rcp = fdiv(1.0, d[k])
        if d[k] > 0:
            ins.append((bb[0][k] - o[k]) * rcp); outs.append((bb[1][k] - o[k]) * rcp)
        else:
            ins.append((bb[1][k] - o[k]) * rcp); outs.append((bb[0][k] - o[k]) * rcp)
    return fmax3(*ins), fmin3(*outs)
def bbclip_ub_rcp(o, rcp, bb):  # :725-741: takes reciprocals, branches on their sign
    ins, outs = [], []
    for k in range(3):
        if rcp[k] > 0:
            ins.append((bb[0][k] - o[k]) * rcp[k]); outs.append((bb[1][k] - o[k]) * rcp[k])
        else:
            ins.append((bb[1][k] - o[k]) * rcp[k]); outs.append((bb[0][k] - o[k]) * rcp[k])
    return fmax3(*ins), fmin3(*outs)


# ----------------------------------------------------------------------------------------------- Rayint
# a hit is (depth, pos, norm, texs, uid); a miss is None.  (riray / uvw / tags are not needed for these scenes.)
def ridepth(h): return INF if h is None else h[0]                  # Solid.hs:33-34
def nearest(a, b):                                                  # Solid.hs:37-44: ties -> b
    if b is None: return a
    if a is None: return b
    return a if a[0] < b[0] else b


class Solid:
    uid = -1
    def rayint(self, o, d, dist, texs): raise NotImplementedError
    def shadow(self, o, d, dist): return self.rayint(o, d, dist, ()) is not None   # class default, Solid.hs:218-221
    def inside(self, p): return False
    def bound(self): raise NotImplementedError
    def get_metainfo(self, p): return ()                                           # Solid.hs:254
    def tolist(self): return [self]                                                 # Solid.hs:230
    def transform(self, xf): return Instance(self, xf)                              # Solid.hs:235 (xf = composed (fwd, inv))
    def transform_leaf(self, xf): return self.transform(xf)                         # Solid.hs:240
    def flatten_transform(self): return self.tolist()                               # Solid.hs:246


def flatten_item(s):  # the SolidItem instance, Solid.hs:273: flatten_transform (SolidItem s) = [SolidItem (flatten_transform s)]
    return Group(s.flatten_transform())


class Sphere(Solid):  # Sphere.hs
    def __init__(self, c, r): self.c, self.r = tuple(c), r
    def rayint(self, o, d, dist, texs):  # :20-41 (Q4)
        eo = vsub(self.c, o)
        v = vdot(eo, d)
        disc = self.r * self.r - (vdot(eo, eo) - v * v)
        if disc < 0: return None
        sq = math.sqrt(disc)
        hit = (v - sq) if (v - sq) > 0 else (v + sq)
        if hit < 0 or hit > dist: return None
        p = vscaleadd(o, d, hit)
        return (hit, p, vnorm(vsub(p, self.c)), texs, self.uid, (o, d))  # (.., riray: the ray as this solid received it)
    def shadow(self, o, d, dist):  # :51-71
        eo = vsub(self.c, o)
        v = vdot(eo, d)
        if not ((dist >= (v - self.r)) and (v > 0)): return False
        disc = self.r * self.r - (vdot(eo, eo) - v * v)
        if disc < 0: return False
        sq = math.sqrt(disc)
        hit = (v - sq) if (v - sq) > 0 else (v + sq)
        return not (hit < 0 or hit > dist)
    def inside(self, p): off = vsub(self.c, p); return vdot(off, off) < self.r * self.r  # :73-76
    def bound(self): return (tuple(self.c[k] - self.r for k in range(3)), tuple(self.c[k] + self.r for k in range(3)))  # :78-81


def tri_core(p1, p2, p3, o, d, dist):  # Triangle.hs:45-73: (t, b1, b2) or None
    e1, e2 = vsub(p2, p1), vsub(p3, p1)
    s1 = vcross(d, e2)
    divisor = vdot(s1, e1)
    if divisor == 0: return None
    inv = 1.0 / divisor
    dd = vsub(o, p1)
    b1 = vdot(dd, s1) * inv
    if b1 < 0 or b1 > 1: return None
    s2 = vcross(dd, e1)
    b2 = vdot(d, s2) * inv
    if b2 < 0 or b1 + b2 > 1: return None
    t = vdot(e2, s2) * inv
    if t < 0 or t > dist: return None
    return t, b1, b2


class Triangle(Solid):  # Triangle.hs
    def __init__(self, p1, p2, p3): self.p = (tuple(p1), tuple(p2), tuple(p3))
    def rayint(self, o, d, dist, texs):
        r = tri_core(*self.p, o, d, dist)
        if r is None: return None
        e1, e2 = vsub(self.p[1], self.p[0]), vsub(self.p[2], self.p[0])
        return (r[0], vscaleadd(o, d, r[0]), vnorm(vcross(e1, e2)), texs, self.uid, (o, d))  # :73: not flipped toward the viewer (Q5)
    def shadow(self, o, d, dist): return tri_core(*self.p, o, d, dist) is not None  # :82-107
    def bound(self):  # :147-158
        return (tuple(fmin(fmin(self.p[0][k], self.p[1][k]), self.p[2][k]) - DELTA for k in range(3)),
                tuple(fmax(fmax(self.p[0][k], self.p[1][k]), self.p[2][k]) + DELTA for k in range(3)))
    def transform(self, xf): return Triangle(*[xfm_point(xf[0], q) for q in self.p])  # bakes the matrix in, :164-168


def smooth_normal(n1, n2, n3, b1, b2):  # Triangle.hs:135-139: vnorm (vadd3 (n1 * (1 - (b1 + b2))) (n2 * b1) (n3 * b2))
    a, b, c = vscale(n1, 1 - (b1 + b2)), vscale(n2, b1), vscale(n3, b2)
    return vnorm((a[0] + b[0] + c[0], a[1] + b[1] + c[1], a[2] + b[2] + c[2]))


class TriangleNorm(Triangle):  # Triangle.hs:109-141, 170-178: vertex normals, interpolated by the hit's barycentrics
    def __init__(self, p1, p2, p3, n1, n2, n3): Triangle.__init__(self, p1, p2, p3); self.n = (tuple(n1), tuple(n2), tuple(n3))
    def rayint(self, o, d, dist, texs):
        r = tri_core(*self.p, o, d, dist)
        if r is None: return None
        return (r[0], vscaleadd(o, d, r[0]), smooth_normal(*self.n, r[1], r[2]), texs, self.uid, (o, d))
    def transform(self, xf): return TriangleNorm(*[xfm_point(xf[0], q) for q in self.p], *[vnorm(xfm_vec(xf[0], n)) for n in self.n])


class Box(Solid):  # Box.hs
    def __init__(self, a, b): self.bb = (tuple(a), tuple(b))
    def rayint(self, o, d, dist, texs):  # :18-54 (Q1, Q6)
        ins, outs = [], []
        for k in range(3):
            rcp = fdiv(1.0, d[k])
            if d[k] > 0: ins.append((self.bb[0][k] - o[k]) * rcp); outs.append((self.bb[1][k] - o[k]) * rcp)
            else: ins.append((self.bb[1][k] - o[k]) * rcp); outs.append((self.bb[0][k] - o[k]) * rcp)
        lastin, firstout = fmax3(*ins), fmin3(*outs)
        if lastin > firstout or firstout < 0 or lastin > dist: return None
        axes = ((1.0, 0, 0), (0, 1.0, 0), (0, 0, 1.0))
        if lastin < 0:  # origin inside: the exit face, normal along the direction
            k = 0 if outs[0] == firstout else (1 if outs[1] == firstout else 2)
            n = axes[k] if d[k] > 0 else vinvert(axes[k])
            return (firstout, vscaleadd(o, d, firstout), n, texs, self.uid, (o, d))
        k = 0 if ins[0] == lastin else (1 if ins[1] == lastin else 2)
        n = vinvert(axes[k]) if d[k] > 0 else axes[k]
        return (lastin, vscaleadd(o, d, lastin), n, texs, self.uid, (o, d))
    def shadow(self, o, d, dist):  # :56-62
        near, far = bbclip_ub(o, d, self.bb)
        return not (near > far or far <= 0 or far > dist)
    def inside(self, p): return all(p[k] > self.bb[0][k] and p[k] < self.bb[1][k] for k in range(3))  # :64-68 (strict)
    def bound(self): return self.bb


class Plane(Solid):  # Plane.hs
    def __init__(self, n, off): self.n, self.off = tuple(n), off
    def rayint(self, o, d, dist, texs):  # :27-32 (Q2: a NaN passes both tests)
        hit = -fdiv(vdot(self.n, o) - self.off, vdot(self.n, d))
        if hit < 0 or hit > dist: return None
        return (hit, vscaleadd(o, d, hit), self.n, texs, self.uid, (o, d))
    def inside(self, p): return vdot(vsub(vscale(self.n, self.off), p), self.n) > 0  # :34-38
    def bound(self): return EVERYTHING_BB  # :40-44


class Tex(Solid):  # Tex.hs:53-74
    def __init__(self, s, mat): self.s, self.mat = s, mat
    def rayint(self, o, d, dist, texs): return self.s.rayint(o, d, dist, (self.mat,) + texs)
    def shadow(self, o, d, dist): return self.s.shadow(o, d, dist)
    def inside(self, p): return self.s.inside(p)
    def bound(self): return self.s.bound()
    def get_metainfo(self, p): return (self.mat,) + self.s.get_metainfo(p)


class Group(Solid):  # the list instance, Solid.hs:326-339
    def __init__(self, xs): self.xs = xs
    def rayint(self, o, d, dist, texs):
        best = None
        for s in self.xs: best = nearest(best, s.rayint(o, d, dist, texs))  # foldl' nearest RayMiss; every item sees the same dist (Q9)
        return best
    def shadow(self, o, d, dist):
        acc = False
        for s in self.xs: acc = acc or s.shadow(o, d, dist)
        return acc
    def inside(self, p): return any(s.inside(p) for s in self.xs)
    def bound(self):
        bb = EMPTY_BB
        for s in self.xs: bb = bbjoin(bb, s.bound())
        return bb
    def tolist(self): return [y for s in self.xs for y in s.tolist()]
    def transform_leaf(self, xf): return Group([x.transform_leaf(xf) for x in self.tolist()])   # Solid.hs:334
    def flatten_transform(self): return [flatten_item(x) for x in self.xs]                     # Solid.hs:335: concat (map flatten_transform xs), xs :: [SolidItem]
    def get_metainfo(self, p):
        acc = ()
        for s in self.xs:
            if s.inside(p): acc = s.get_metainfo(p) + acc
        return acc


class Void(Solid):  # Solid.hs:349-360
    def rayint(self, o, d, dist, texs): return None
    def shadow(self, o, d, dist): return False
    def bound(self): return EMPTY_BB
    def tolist(self): return []
    def transform(self, xf): return self


def group(xs):  # Solid.hs:293-302 (Q22)
    flat = [y for s in xs for y in s.tolist()]
    if not flat: return Void()
    if len(flat) == 1: return flat[0]
    return Group(flat)


class Instance(Solid):  # Solid.hs:386-532
    def __init__(self, s, xf): self.s, self.f, self.i = s, xf[0], xf[1]
    def _local(self, o, d):
        nd, no = xfm_vec(self.i, d), xfm_point(self.i, o)
        ls = vlen(nd)
        return no, vscale(nd, 1.0 / ls), ls
    def rayint(self, o, d, dist, texs):  # :388-403 (Q8)
        no, nd, ls = self._local(o, d)
        h = self.s.rayint(no, nd, dist * ls, texs)
        if h is None: return None
        return (h[0] * (1.0 / ls), xfm_point(self.f, h[1]), vnorm(xfm_tvec(self.i, h[2])), h[3], h[4], h[5])  # riray stays the local ray (:397-403)
    def shadow(self, o, d, dist):  # :464-471
        no, nd, ls = self._local(o, d)
        return self.s.shadow(no, nd, dist * ls)
    def inside(self, p): return self.s.inside(xfm_point(self.i, p))
    def bound(self):  # :477-484
        lo, hi = self.s.bound()
        return bbpts([xfm_point(self.f, (x, y, z)) for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    def get_metainfo(self, p): return self.s.get_metainfo(xfm_point(self.i, p))
    def transform(self, xf):  # merges: compose ([xfm2] ++ xfm1), :494-496
        return self.s.transform((mat_mult(xf[0], self.f), mat_mult(self.i, xf[1])))
    def transform_leaf(self, xf):  # :498-500: transform_leaf s [compose ([xfm2] ++ xfm1)]
        return self.s.transform_leaf((mat_mult(xf[0], self.f), mat_mult(self.i, xf[1])))
    def flatten_transform(self): return [self.s.transform_leaf((self.f, self.i))]                # :509-511


NVZ, VZ = (0.0, 0.0, -1.0), (0.0, 0.0, 1.0)


def disc_hit(point, norm, r2, o, d, dist):  # rayint_disc / shadow_disc, Cone.hs:69-91: the distance, or None
    t = fdiv(-vdot(norm, vsub(o, point)), vdot(norm, d))  # plane_int_dist, Vec.hs:391-394
    if t < 0 or t > dist: return None
    off = vsub(vscaleadd(o, d, t), point)
    return None if vdot(off, off) > r2 else t


class Disc(Solid):  # Cone.hs:21, 29-31, 69-102
    def __init__(self, pos, norm, r): self.pos, self.norm, self.r2 = tuple(pos), tuple(norm), r * r
    def rayint(self, o, d, dist, texs):
        t = disc_hit(self.pos, self.norm, self.r2, o, d, dist)
        return None if t is None else (t, vscaleadd(o, d, t), self.norm, texs, self.uid, (o, d))
    def shadow(self, o, d, dist): return disc_hit(self.pos, self.norm, self.r2, o, d, dist) is not None
    def bound(self):  # bound (sphere pos (sqrt rsqr))
        r = math.sqrt(self.r2)
        return (tuple(self.pos[k] - r for k in range(3)), tuple(self.pos[k] + r for k in range(3)))


class NoShadow(Solid):  # Tex.hs:77-85
    def __init__(self, s): self.s = s
    def rayint(self, o, d, dist, texs): return self.s.rayint(o, d, dist, texs)
    def shadow(self, o, d, dist): return False
    def inside(self, p): return self.s.inside(p)
    def bound(self): return self.s.bound()
    def get_metainfo(self, p): return self.s.get_metainfo(p)


class OnlyShadow(Solid):  # Tex.hs:88-96
    def __init__(self, s): self.s = s
    def rayint(self, o, d, dist, texs): return None
    def shadow(self, o, d, dist): return self.s.shadow(o, d, dist)
    def inside(self, p): return self.s.inside(p)
    def bound(self): return self.s.bound()
    def get_metainfo(self, p): return self.s.get_metainfo(p)


class Cone(Solid):  # the canonical cone on the z axis (Cone.hs:155-251): radius r at z = 0, apex at z = height, kept between clip1 and clip2
    def __init__(self, r, clip1, clip2, height): self.r, self.c1, self.c2, self.h = r, clip1, clip2, height
    def _side(self, o, d, dist):  # the quadratic of rayint_cone / shadow_cone: the distance to the infinite cone, or None
        r, height = self.r, self.h
        k = fdiv(r, height); k = k * k
        ox, oy, oz = o; dx, dy, dz = d
        a = dx * dx + dy * dy - k * dz * dz
        b = 2 * (dx * ox + dy * oy - k * dz * (oz - height))
        c = ox * ox + oy * oy - k * (oz - height) * (oz - height)
        disc = b * b - 4 * a * c
        if disc < 0: return None
        ds = math.sqrt(disc)
        q = (b - ds) * (-0.5) if b < 0 else (b + ds) * (-0.5)
        t0_, t1_ = fdiv(q, a), fdiv(c, q)
        t0, t1 = fmin(t0_, t1_), fmax(t0_, t1_)
        if t1 < 0 or t0 > dist: return None
        t = t1 if t0 < 0 else t0
        return None if (t < 0 or t > dist) else t
    def _cap(self, o, d, dist):  # which end disc the ray can still reach when the side hit lies outside the clips (:181-191)
        if d[2] > 0: return ((0.0, 0.0, self.c1), NVZ, self.r * self.r) if o[2] < self.c1 else None
        if o[2] > self.c2:
            r2 = self.r * (1 - fdiv(self.c2 - self.c1, self.h))
            return ((0.0, 0.0, self.c2), VZ, r2 * r2)
        return None
    def rayint(self, o, d, dist, texs):
        t = self._side(o, d, dist)
        if t is None: return None
        pos = vscaleadd(o, d, t)
        if pos[2] > self.c1 and pos[2] < self.c2:
            invhyp = 1 / math.sqrt(self.h * self.h + self.r * self.r)
            up, out = self.r * invhyp, self.h * invhyp
            corr = fdiv(out, math.sqrt(pos[0] * pos[0] + pos[1] * pos[1]))
            return (t, pos, (pos[0] * corr, pos[1] * corr, up), texs, self.uid, (o, d))
        cap = self._cap(o, d, dist)
        if cap is None: return None
        td = disc_hit(cap[0], cap[1], cap[2], o, d, dist)
        return None if td is None else (td, vscaleadd(o, d, td), cap[1], texs, self.uid, (o, d))
    def shadow(self, o, d, dist):
        t = self._side(o, d, dist)
        if t is None: return False
        z = o[2] + d[2] * t
        if z > self.c1 and z < self.c2: return True
        cap = self._cap(o, d, dist)
        return cap is not None and disc_hit(cap[0], cap[1], cap[2], o, d, dist) is not None
    def inside(self, p):  # :244-247
        r = self.r * (1 - fdiv(p[2] - self.c1, self.h))
        return p[2] > self.c1 and p[2] < self.c2 and p[0] * p[0] + p[1] * p[1] < r * r
    def bound(self): return ((-self.r, -self.r, self.c1), (self.r, self.r, self.c2))  # :249-251


class Cylinder(Solid):  # the canonical cylinder on the z axis between h1 and h2 (Cone.hs:104-151); shadow is the class default
    def __init__(self, r, h1, h2): self.r, self.h1, self.h2 = r, h1, h2
    def rayint(self, o, d, dist, texs):
        r = self.r
        ox, oy, oz = o; dx, dy, dz = d
        a = dx * dx + dy * dy
        b = 2 * (dx * ox + dy * oy)
        c = ox * ox + oy * oy - r * r
        disc = b * b - 4 * a * c
        if disc < 0: return None
        ds = math.sqrt(disc)
        q = (b - ds) * (-0.5) if b < 0 else (b + ds) * (-0.5)
        t0_, t1_ = fdiv(q, a), fdiv(c, q)
        t0, t1 = fmin(t0_, t1_), fmax(t0_, t1_)
        if t1 < 0 or t0 > dist: return None
        t = t1 if t0 < 0 else t0
        if t < 0 or t > dist: return None
        pos = vscaleadd(o, d, t)
        if pos[2] > self.h1 and pos[2] < self.h2: return (t, pos, (fdiv(pos[0], r), fdiv(pos[1], r), 0.0), texs, self.uid, (o, d))
        if dz > 0: cap = ((0.0, 0.0, self.h1), NVZ) if oz < self.h1 else None
        else: cap = ((0.0, 0.0, self.h2), VZ) if oz > self.h2 else None
        if cap is None: return None
        td = disc_hit(cap[0], cap[1], r * r, o, d, dist)
        return None if td is None else (td, vscaleadd(o, d, td), cap[1], texs, self.uid, (o, d))
    def inside(self, p): return p[2] > self.h1 and p[2] < self.h2 and p[0] * p[0] + p[1] * p[1] < self.r * self.r
    def bound(self): return ((-self.r, -self.r, self.h1), (self.r, self.r, self.h2))


def cylinder(p1, p2, r):  # Cone.hs:41-49
    axis = vsub(p2, p1)
    ln = vlen(axis)
    ax1 = vscale(axis, 1 / ln)
    ax2, ax3 = orth(ax1)
    return Cylinder(r, 0.0, ln).transform(compose([xyz_to_uvw(ax2, ax3, ax1), translate(p1)]))


class Bound(Solid):  # Bound.hs:27-66: sb is looked at only by rays that start inside sa or hit it
    def __init__(self, a, b): self.a, self.b = a, b
    def rayint(self, o, d, dist, texs): return self.b.rayint(o, d, dist, texs) if (self.a.inside(o) or self.a.shadow(o, d, dist)) else None
    def shadow(self, o, d, dist): return self.b.shadow(o, d, dist) if (self.a.inside(o) or self.a.shadow(o, d, dist)) else False
    def inside(self, p): return self.a.inside(p) and self.b.inside(p)
    def get_metainfo(self, p): return self.b.get_metainfo(p) if self.a.inside(p) else ()
    def bound(self): return bboverlap(self.a.bound(), self.b.bound())
    def transform_leaf(self, xf): return self.b.transform_leaf(xf)                               # Bound.hs:69-71
    def flatten_transform(self): return [flatten_item(self.b)]                                   # Bound.hs:73-74 (sb is a SolidItem)


class InnerBound(Solid):  # Bound.hs:93-108: sb is searched no farther than sa's hit
    def __init__(self, a, b): self.a, self.b = a, b
    def rayint(self, o, d, dist, texs): return self.b.rayint(o, d, ridepth(self.a.rayint(o, d, dist, ())), texs)
    def shadow(self, o, d, dist): return self.a.shadow(o, d, dist) or self.b.shadow(o, d, dist)
    def inside(self, p): return self.a.inside(p) or self.b.inside(p)
    def bound(self): return self.b.bound()
    def transform_leaf(self, xf): return self.b.transform_leaf(xf)                               # Bound.hs:112
    def flatten_transform(self): return [flatten_item(self.b)]                                   # Bound.hs:111


def orth(v1):  # Vec.hs:366-378
    dvx = v1[0]
    v2 = vnorm(vcross(v1, (1.0, 0.0, 0.0))) if (dvx < 0.8 and dvx > -0.8) else vnorm(vcross(v1, (0.0, 1.0, 0.0)))
    return v2, vcross(v1, v2)


def xyz_to_uvw(u, v, w):  # Vec.hs:602-623: (forward, inverse)
    return (u[0], v[0], w[0], 0.0, u[1], v[1], w[1], 0.0, u[2], v[2], w[2], 0.0) + (u[0], u[1], u[2], 0.0, v[0], v[1], v[2], 0.0, w[0], w[1], w[2], 0.0)


def translate(t):  # Vec.hs:564-568
    return (1.0, 0, 0, t[0], 0, 1.0, 0, t[1], 0, 0, 1.0, t[2]) + (1.0, 0, 0, -t[0], 0, 1.0, 0, -t[1], 0, 0, 1.0, -t[2])


def cone(p1, r1, p2, r2):  # Cone.hs:53-67
    if r1 < r2: return cone(p2, r2, p1, r1)
    if r1 - r2 < DELTA: return cylinder(p1, p2, r2)
    axis = vsub(p2, p1)
    ln = vlen(axis)
    ax1 = vscale(axis, 1 / ln)
    ax2, ax3 = orth(ax1)
    height = fdiv(r1 * ln, r1 - r2)
    return Cone(r1, 0.0, ln, height).transform(compose([xyz_to_uvw(ax2, ax3, ax1), translate(p1)]))


# ----------------------------------------------------------------------------------------------- Texture.hs: stripes and Perlin noise
def square_wave(x): return 0.0 if (x - math.floor(x)) < 0.5 else 1.0  # :11-14
def sine_wave(x): return math.sin(x * 2 * math.pi) * 0.5 + 0.5           # :23-24


def triangle_wave(x):  # :16-21
    off = x - math.floor(x)
    return off * 2 if off < 0.5 else 2 - off * 2


PHI = (3, 0, 2, 7, 4, 1, 5, 11, 8, 10, 9, 6)                                                     # :55-56
GRAD = [(x, y, z) for x in (-1.0, 0.0, 1.0) for y in (-1.0, 0.0, 1.0) for z in (-1.0, 0.0, 1.0) if 1.1 < vlen((x, y, z)) < 1.5]  # :58-63


def noise(p):  # :93-108
    def omega(t_):  # :48-52
        t = abs(t_); t2 = t * t; t3 = t2 * t
        return (-6) * t3 * t2 + 15 * t3 * t - 10 * t3 + 1
    def knot(i, j, k, v):  # :65-75
        a = PHI[abs(k) % 12]; b = PHI[abs(j + a) % 12]; c = PHI[abs(i + b) % 12]
        return omega(v[0]) * omega(v[1]) * omega(v[2]) * vdot(GRAD[c], v)
    i, j, k = math.floor(p[0]), math.floor(p[1]), math.floor(p[2])
    u, v, w = p[0] - i, p[1] - j, p[2] - k
    return (knot(i, j, k, (u, v, w)) + knot(i + 1, j, k, (u - 1, v, w)) + knot(i, j + 1, k, (u, v - 1, w)) + knot(i, j, k + 1, (u, v, w - 1)) +
            knot(i + 1, j + 1, k, (u - 1, v - 1, w)) + knot(i + 1, j, k + 1, (u - 1, v, w - 1)) + knot(i, j + 1, k + 1, (u, v - 1, w - 1)) +
            knot(i + 1, j + 1, k + 1, (u - 1, v - 1, w - 1)))


def perlin(v): return (noise(v) + 1) * 0.5  # :110-117


def rayint_advance(s, o, d, dist, texs, adv):  # Solid.hs:85-91
    a = adv + DELTA
    h = s.rayint(vscaleadd(o, d, a), d, dist - a, texs)
    return None if h is None else (h[0] + a,) + h[1:]


class Difference(Solid):  # Csg.hs
    def __init__(self, a, b, useatex=True): self.a, self.b, self.useatex = a, b, useatex  # False: difference_retexture (:29-30)
    def rayint(self, o, d, dist, texs):  # :33-54 (Q13), recursive like the reference
        if self.b.inside(o):
            hb = self.b.rayint(o, d, dist, texs)
            if hb is None: return None
            if self.a.inside(hb[1]) and not self.b.inside(vscaleadd(hb[1], d, DELTA)):
                if not self.useatex: return (hb[0], hb[1], vinvert(hb[2])) + hb[3:]  # `RayHit bd bp (vinvert bn) ray uvw bt btags` (:43)
                return (hb[0], hb[1], vinvert(hb[2]), self.a.get_metainfo(hb[1]), hb[4], hb[5])  # useatex: textures of A at the point
            return rayint_advance(self, o, d, dist, texs, hb[0])
        ha = self.a.rayint(o, d, dist, texs)
        if ha is None: return None
        hb = self.b.rayint(o, d, dist, texs)
        if hb is None: return ha
        if ha[0] < hb[0]: return ha
        return rayint_advance(self, o, d, dist, texs, hb[0])
    def inside(self, p): return self.a.inside(p) and not self.b.inside(p)  # :92-94
    def bound(self): return self.a.bound()
    def get_metainfo(self, p): return self.a.get_metainfo(p) if (self.a.inside(p) and not self.b.inside(p)) else ()


class Intersection(Solid):  # Csg.hs
    def __init__(self, ss): self.ss = list(ss)
    def rayint(self, o, d, dist, texs):  # :68-90 (Q14)
        if not self.ss or dist < 0: return None
        s, rest = self.ss[0], self.ss[1:]
        if not rest: return s.rayint(o, d, dist, texs)
        tail = Intersection(rest)
        hs = s.rayint(o, d, dist, texs)
        if s.inside(o):
            if hs is None: return tail.rayint(o, d, dist, texs)
            r = tail.rayint(o, d, hs[0], texs)
            return rayint_advance(self, o, d, dist, texs, hs[0]) if r is None else r
        if hs is None: return None
        if tail.inside(hs[1]): return hs
        return rayint_advance(self, o, d, dist, texs, hs[0])
    def inside(self, p): return all(s.inside(p) for s in self.ss)  # :96-101 (True when empty)
    def bound(self):
        if not self.ss: return EMPTY_BB
        bb = EVERYTHING_BB
        for s in self.ss: bb = bboverlap(bb, s.bound())
        return bb
    def get_metainfo(self, p):
        if not self.inside(p): return ()
        acc = ()
        for s in self.ss: acc = acc + s.get_metainfo(p)
        return acc


# ----------------------------------------------------------------------------------------------- Bih (Bih.hs)
def build_rec(objs, bb, mid, depth, objcount):  # :211-285 (Q11); a node is ("leaf", [solids]) or ("branch", lsplit, rsplit, axis, l, r)
    if objcount <= 3:
        return ("leaf", [s for (_, s) in objs])
    sa = max(0.0, bbsa(bb))
    parts = []
    for ax in range(3):
        l = [x for x in objs if bbmid(x[0])[ax] < mid[ax]]
        r = [x for x in objs if not (bbmid(x[0])[ax] < mid[ax])]
        parts.append((l, r))
    big = [x for x in objs if max(0.0, bbsa(x[0])) > sa * 0.4]
    small = [x for x in objs if not (max(0.0, bbsa(x[0])) > sa * 0.4)]
    parts.append((big, small))
    cands = []
    for k, (l, r) in enumerate(parts):
        ax = k if k < 3 else 0  # the big / small split is stored as an axis-0 node
        lmax = -INF
        for (b, _) in l: lmax = fmax(lmax, b[1][ax])
        rmin = INF
        for (b, _) in r: rmin = fmin(rmin, b[0][ax])
        lbb = (bb[0], vset(bb[1], ax, lmax))
        rbb = (vset(bb[0], ax, rmin), bb[1])
        cost = ((max(0.0, bbsa(lbb)) * len(l)) + (max(0.0, bbsa(rbb)) * len(r))) * (1.1 if k < 3 else 1.2)
        cands.append((cost, ax, l, lbb, r, rbb, lmax, rmin))
    costx, costy, costz, costb = [c[0] for c in cands]
    costorig = sa * objcount
    if costorig < costx and costorig < costy and costorig < costz and costorig < costb:
        return ("leaf", [s for (_, s) in objs])
    if costx < costy and costx < costz and costx < costb: pick = cands[0]
    elif costy < costz and costy < costb: pick = cands[1]
    elif costy < costb: pick = cands[2]  # as written in the reference: costy, where costz was meant (:283)
    else: pick = cands[3]
    _, ax, l, lbb, r, rbb, lmax, rmin = pick
    return ("branch", lmax + DELTA, rmin - DELTA, ax, build_rec(l, lbb, bbmid(lbb), depth + 1, len(l)), build_rec(r, rbb, bbmid(rbb), depth + 1, len(r)))


class Bih(Solid):
    def __init__(self, slds):  # bih, :309-324
        objs = [(s.bound(), s) for s in slds]
        bb = EMPTY_BB
        for (b, _) in objs: bb = bbjoin(bb, b)
        if any(v == -INF for v in bb[0]) or any(v == INF for v in bb[1]): raise ValueError("bih: infinite bounding box")
        self.bb, self.root = bb, build_rec(objs, bb, bbmid(bb), 0, len(slds))
    def rayint(self, o, d, dist, texs):  # rayint_bih, :332-368 (Q10)
        near, far = bbclip_ub(o, d, self.bb)
        dirrs = (fdiv(1.0, d[0]), fdiv(1.0, d[1]), fdiv(1.0, d[2]))
        def traverse(n, near, far):
            if n[0] == "leaf":
                return Group(n[1]).rayint(o, d, far, texs) if n[1] else None  # rayint s r far: the list instance
            _, lsplit, rsplit, axis, l, r = n
            dirr, oo = dirrs[axis], o[axis]
            dl, dr = (lsplit - oo) * dirr, (rsplit - oo) * dirr
            if near > far: return None
            if dirr > 0:
                return nearest(traverse(l, near, fmin(dl, far)) if near < dl else None, traverse(r, fmax(dr, near), far) if dr < far else None)
            return nearest(traverse(r, near, fmin(dr, far)) if near < dr else None, traverse(l, fmax(dl, near), far) if dl < far else None)
        return traverse(self.root, near, fmin(dist, far))
    def shadow(self, o, d, dist):  # shadow_bih, :510-544
        near, far0 = bbclip_ub(o, d, self.bb)
        def traverse(n, near, far):
            if n[0] == "leaf": return Group(n[1]).shadow(o, d, fmin(dist, far))
            _, lsplit, rsplit, axis, l, r = n
            dirr, oo = fdiv(1.0, d[axis]), o[axis]
            dl, dr = (lsplit - oo) * dirr, (rsplit - oo) * dirr
            if near > far: return False
            if dirr > 0:
                return (traverse(l, near, fmin(dl, far)) if near < dl else False) or (traverse(r, fmax(dr, near), far) if dr < far else False)
            return (traverse(r, near, fmin(dr, far)) if near < dr else False) or (traverse(l, fmax(dl, near), far) if dl < far else False)
        return traverse(self.root, near, fmin(dist, far0))
    def _inbox(self, p): return all(p[k] > self.bb[0][k] and p[k] < self.bb[1][k] for k in range(3))
    def inside(self, p):  # inside_bih, :550-565: a point traversal; a leaf holds a group (the list instance: any)
        def traverse(n):
            if n[0] == "leaf": return any(s.inside(p) for s in n[1])
            _, lsplit, rsplit, axis, l, r = n
            return (traverse(l) if p[axis] < lsplit else False) or (traverse(r) if p[axis] > rsplit else False)
        return self._inbox(p) and traverse(self.root)
    def get_metainfo(self, p):  # get_metainfo_bih, :567-585: left `paircat` right; a leaf like the list instance (Solid.hs:337-339)
        def traverse(n):
            if n[0] == "leaf":
                out = ()
                for s in n[1]:
                    if s.inside(p): out = s.get_metainfo(p) + out
                return out
            _, lsplit, rsplit, axis, l, r = n
            return (traverse(l) if p[axis] < lsplit else ()) + (traverse(r) if p[axis] > rsplit else ())
        return traverse(self.root) if self._inbox(p) else ()
    def bound(self): return self.bb


# ----------------------------------------------------------------------------------------------- Mesh (Mesh.hs)
class Mesh(Solid):
    def __init__(self, verts, tris, texi, texv, norms=(), tnorms=None):  # mesh, :50-134; tris: (a, b, c) vertex indices; texi: per-triangle index into texv, or -1; tnorms: (na, nb, nc) or -1s
        self.verts, self.tris, self.texi, self.texv = [tuple(v) for v in verts], [tuple(t) for t in tris], texi, texv
        self.norms, self.tnorms = list(norms), tnorms
        self.bb = bbpts(self.verts)
        self.tbb = [bbpts([self.verts[a], self.verts[b], self.verts[c]]) for (a, b, c) in self.tris]
        self.bvh = self.build_tree(list(range(len(self.tris))), self.bb)
    def trisbb(self, idx):
        bb = EMPTY_BB
        for i in idx: bb = bbjoin(bb, self.tbb[i])
        return bb
    def build_tree(self, tris, bb):  # :69-113 (Q12)
        if len(tris) < 3: return ("leaf", tris)
        mid, sa = bbmid(bb), bbsa(bb)
        parts = []
        for ax in range(3):
            parts.append(([t for t in tris if bbmid(self.tbb[t])[ax] < mid[ax]], [t for t in tris if not (bbmid(self.tbb[t])[ax] < mid[ax])]))
        parts.append(([t for t in tris if bbsa(self.tbb[t]) > sa * 0.4], [t for t in tris if not (bbsa(self.tbb[t]) > sa * 0.4)]))
        cands = []
        for (l, r) in parts:
            lbb, rbb = self.trisbb(l), self.trisbb(r)
            cands.append(((bbsa(lbb) * len(l) + bbsa(rbb) * len(r)) * 1.1, lbb, rbb, l, r))
        x, y, z, b = [c[0] for c in cands]
        lcost = bbsa(bb) * len(tris)
        if lcost < x and lcost < y and lcost < z and lcost < b: return ("leaf", tris)
        if x < y and x < z and x < b: pick = cands[0]
        elif y < z and y < b: pick = cands[1]
        elif z < b: pick = cands[2]
        else: pick = cands[3]
        _, lbb, rbb, l, r = pick
        return ("branch", lbb, rbb, self.build_tree(l, lbb), self.build_tree(r, rbb))
    def rayint(self, o, d, depth, texs):  # rayint_mesh, :136-198
        rcp = (fdiv(1.0, d[0]), fdiv(1.0, d[1]), fdiv(1.0, d[2]))
        near, far = bbclip_ub_rcp(o, rcp, self.bb)
        if near > far or near > depth or far < 0: return None
        def rayint_tri(i, far):
            a, b, c = (self.verts[k] for k in self.tris[i])
            r = tri_core(a, b, c, o, d, far)
            if r is None: return None
            tex = texs if self.texi[i] == -1 else (self.texv[self.texi[i]],) + texs  # :148-150
            if self.tnorms is None or self.tnorms[i][0] == -1: nrm = vnorm(vcross(vsub(b, a), vsub(c, a)))  # :154-155
            else: nrm = smooth_normal(*(self.norms[k] for k in self.tnorms[i]), r[1], r[2])            # :156-160
            return (r[0], vscaleadd(o, d, r[0]), nrm, tex, self.uid, (o, d))
        def traverse(n, near, far):
            if n[0] == "leaf":
                acc = None
                for i in n[1]: acc = nearest(acc, rayint_tri(i, far))  # every triangle with the box interval's far, not depth
                return acc
            _, lbb, rbb, l, r = n
            ln, lf = bbclip_ub_rcp(o, rcp, lbb)
            rn, rf = bbclip_ub_rcp(o, rcp, rbb)
            lnear, lfar, rnear, rfar = max(near, ln), min(far, lf), max(near, rn), min(far, rf)  # Prelude max / min here
            if lnear < rnear:
                lres = None if (lnear > lfar or lnear > depth or lfar < 0) else traverse(l, lnear, lfar)
                rfar2 = min(rfar, ridepth(lres))
                return nearest(lres, None if (rnear > rfar2 or rnear > depth or rfar2 < 0) else traverse(r, rnear, rfar))  # the unshrunk rfar goes down (:184)
            rres = None if (rnear > rfar or rnear > depth or rfar < 0) else traverse(r, rnear, rfar)
            lfar2 = min(lfar, ridepth(rres))
            return nearest(rres, None if (lnear > lfar2 or lnear > depth or lfar2 < 0) else traverse(l, lnear, lfar))
        return traverse(self.bvh, near, far)
    def shadow(self, o, d, dist): return False  # :210
    def bound(self): return self.bb


# ----------------------------------------------------------------------------------------------- colours (Clr.hs)
def cafold(c1, c2):  # :106-113
    trans = 1 - c1[3]
    return (c1[0] + (c2[0] * trans * c2[3]), c1[1] + (c2[1] * trans * c2[3]), c1[2] + (c2[2] * trans * c2[3]), c1[3] + (c2[3] * trans))
def caweight(a, b, w): return tuple((a[k] * w) + (b[k] * (1 - w)) for k in range(4))  # :87-91
def aclamp(x): return 1.0 if x > 1 else (0.0 if x < 0 else x)                          # :75-79
def casum(cs):  # :93-103
    r = g = b = 0.0
    prod = 1.0
    for c in cs:
        r, g, b = r + c[0] * c[3], g + c[1] * c[3], b + c[2] * c[3]
        prod = prod * (1 - aclamp(c[3]))
    return (r, g, b, 1 - prod)


# ----------------------------------------------------------------------------------------------- trace / shade
class Scene:
    """The backend SceneDesc.replay() drives, then `render`."""
    def __init__(self):
        self.nodes, self.mats, self.lights, self.root, self.cam = [], [], [], None, None
        self.rays = [0, 0, 0]  # primary, shadow, secondary

    # ---- constructors (SceneDesc.replay protocol)
    def _add(self, s): s.uid = len(self.nodes); self.nodes.append(s); return s.uid
    def sphere(self, c, r): return self._add(Sphere(c, r))
    def triangle(self, p1, p2, p3): return self._add(Triangle(p1, p2, p3))
    def trianglenorm(self, p1, p2, p3, n1, n2, n3): return self._add(TriangleNorm(p1, p2, p3, n1, n2, n3))
    def triangles_bulk(self, pts9): return [self._add(Triangle(p[0:3], p[3:6], p[6:9])) for p in pts9.tolist()]
    def box(self, a, b): return self._add(Box(a, b))
    def plane(self, pt, n):  # Plane.hs:16-19: plane orig norm_ = Plane (vnorm norm_) (vdot orig (vnorm norm_))
        nn = vnorm(tuple(n))
        return self._add(Plane(nn, vdot(tuple(pt), nn)))
    def plane_offset(self, n, off): return self._add(Plane(tuple(n), off))  # Plane.hs:23-24
    def group(self, ids): return self._add(group([self.nodes[i] for i in ids]))
    def bih(self, ids): return self._add(Bih([self.nodes[i] for i in ids]) if ids else Void())
    def bih_tolist(self, i):                                                        # `bih (tolist s)`, TestScene.hs:109
        xs = self.nodes[i].tolist()
        return self._add(Bih(xs) if xs else Void())
    def difference(self, a, b): return self._add(Difference(self.nodes[a], self.nodes[b]))
    def difference_retexture(self, a, b): return self._add(Difference(self.nodes[a], self.nodes[b], False))
    def intersection(self, ids): return self._add(Intersection([self.nodes[i] for i in ids]))
    def transform(self, node, xfms): return self._add(self.nodes[node].transform(compose(xfms)))
    def tex(self, node, mat): return self._add(Tex(self.nodes[node], mat))
    def mesh(self, verts, norms, tris, mats):  # Tri a b c na nb nc tex tag (Mesh.hs:27-29)
        return self._add(Mesh([tuple(v) for v in verts.tolist()], [tuple(int(x) for x in t[:3]) for t in tris], [int(t[6]) for t in tris], list(mats),
                              [tuple(v) for v in norms.tolist()], [tuple(int(x) for x in t[3:6]) for t in tris]))
    def material_surface(self, color, alpha, amb, kd, ks, shine): self.mats.append(("surface", tuple(color), alpha, amb, kd, ks, shine)); return len(self.mats) - 1
    def material_reflect(self, refl): self.mats.append(("reflect", refl)); return len(self.mats) - 1
    def material_refract(self, refl, refr, ior): self.mats.append(("refract", refl, refr, ior)); return len(self.mats) - 1
    def material_layers(self, mats): self.mats.append(("layers", list(mats))); return len(self.mats) - 1
    def material_blend(self, a, b, w): self.mats.append(("blend", a, b, w)); return len(self.mats) - 1
    def material_warp(self, frame, scene, lights, xfm):  # Shader.hs:47-50; lights: [(pos, colour, radius, shadow)], xfm: the closure's matrix (3x4 forward first)
        ls = [(tuple(l[0]), tuple(l[1]), l[2], l[3]) for l in lights]
        self.mats.append(("warp", frame, -1 if scene is None else scene, ls, tuple(float(x) for x in list(xfm)[:12])))
        return len(self.mats) - 1
    def tag(self, node): return node  # tags only matter for picking
    def cone(self, p1, r1, p2, r2): return self._add(cone(tuple(p1), r1, tuple(p2), r2))
    def cylinder(self, p1, p2, r): return self._add(cylinder(tuple(p1), tuple(p2), r))
    def disc(self, pos, n, r): return self._add(Disc(pos, n, r))
    def noshadow(self, node): return self._add(NoShadow(self.nodes[node]))
    def onlyshadow(self, node): return self._add(OnlyShadow(self.nodes[node]))
    def flatten_transform(self, i): return self._add(flatten_item(self.nodes[i]))                # `SolidItem (flatten_transform s)`
    def tolist(self, i): return self._add(Group(self.nodes[i].tolist()))
    def bound_object(self, a, b): return self._add(Bound(self.nodes[a], self.nodes[b]))
    def innerbound(self, a, b): return self._add(InnerBound(self.nodes[a], self.nodes[b]))
    def material_blend_fn(self, a, b, fn, params):  # TestScene.hs:213-231: Blend a b (f pos); fn 1 = perlin (pos * s), 3 = stripe axis triangle_wave
        assert fn in (1, 2, 3, 4)
        self.mats.append(("blend_fn", a, b, fn, tuple(float(x) for x in params))); return len(self.mats) - 1

    # ---- Shader.hs
    def mpreshade(self, sld, hit, lights=None):  # :65-80 (Q18)
        out = []
        p, n = hit[1], hit[2]
        for (lpos, lcol, rad, do_shadow) in (self.lights if lights is None else lights):
            lvec = vsub(lpos, p)
            if vdot(lvec, n) < 0: continue
            llen = vlen(lvec)
            ldir = vscale(lvec, 1.0 / llen)
            if llen > rad: continue
            if do_shadow:
                self.rays[1] += 1
                if sld.shadow(vscaleadd(p, n, DELTA), ldir, llen - (2 * DELTA)): continue
            out.append((vscale(lcol, 1.0 / (llen * llen)), ldir))
        return out

    def mpostshade(self, lz, mat, o, d, sld, hit, recurs):  # :82-184 (Q17)
        m = self.mats[mat]
        p, n = hit[1], hit[2]
        eyedir = vinvert(d)
        if m[0] == "surface":
            _, color, alpha, amb, kd, ks, shine = m
            if lz[0] is None: lz[0] = self.mpreshade(sld, hit, lz[1])  # lazy ctxb (Trace.hs:63)
            rgb = vscale(color, amb)
            direct = (0.0, 0.0, 0.0)
            for (lcolor, ldir) in lz[0]:
                half = bisect(ldir, eyedir)
                ldotn = fmax(0, vdot(ldir, n))
                if ks <= DELTA: blinn = 0.0
                else:
                    b = fmax(0, fpow(vdot(half, n), shine) * ldotn)
                    blinn = 0.0 if b != b else b
                direct = vadd(direct, vscale(lcolor, (blinn * ks) + (vdot(ldir, n) * kd)))  # the light term is not tinted by the surface colour
            rgb = vadd(rgb, direct)
            return (rgb[0], rgb[1], rgb[2], alpha)
        if m[0] == "reflect":
            refl = m[1]
            if refl > 0 and recurs > 0:
                out = reflect(d, n)
                c = self.trace(sld, vscaleadd(p, out, DELTA), out, INF, recurs - 1, True, lz[1])[0]
                return (c[0], c[1], c[2], c[3] * refl)
            return (0.0, 0.0, 0.0, 1.0)
        if m[0] == "refract":
            _, refl, refr, ior = m
            if (refl > 0 or refr > 0) and recurs > 0:
                out = reflect(d, n)
                cr = self.trace(sld, vscaleadd(p, out, DELTA), out, INF, recurs - 1, True, lz[1])[0]
                eta = ior if vdot(n, eyedir) > 0 else 1.0 / ior
                c1 = vdot(d, n)
                cs2 = 1 - (eta * eta) * (1 - (c1 * c1))
                if cs2 < 0: ct = (0.0, 0.0, 0.0, 1.0)
                else:
                    t = vadd(vscale(d, eta), vscale(n, eta * c1 - math.sqrt(cs2)))
                    ct = self.trace(sld, vscaleadd(p, t, DELTA), t, INF, recurs - 1, True, lz[1])[0]
                return tuple(cr[k] * refl + ct[k] * refr for k in range(4))
            return (0.0, 0.0, 0.0, 0.0)
        if m[0] == "warp":  # Shader.hs:157-175: the frame through the hit's own (local) ray, then the other scene through the
            _, frame, scene, wlights, xf = m  # warped ray, no farther than the frame's hit; the nearer of the two is shown
            fcolor, fint = self.trace(self.nodes[frame], hit[5][0], hit[5][1], INF, recurs - 1, True, lz[1])
            wo = xfm_point(xf, p)
            wd = vnorm(xfm_vec(xf, vnorm(d)))  # xfm ray hit = xfm_ray M (Ray (pos hit) (vnorm (dir ray))), TestScene.hs:166-172
            wcolor, wint = self.trace(sld if scene < 0 else self.nodes[scene], wo, wd, ridepth(fint), recurs - 1, True, wlights)
            return fcolor if ridepth(fint) < ridepth(wint) else wcolor
        if m[0] == "layers": return casum([self.mpostshade(lz, k, o, d, sld, hit, recurs) for k in m[1]])
        if m[0] == "blend_fn":
            w = perlin(vscale(p, m[4][0])) if m[3] == 1 else {2: square_wave, 3: triangle_wave, 4: sine_wave}[m[3]](vdot(p, m[4][:3]))
            return caweight(self.mpostshade(lz, m[1], o, d, sld, hit, recurs), self.mpostshade(lz, m[2], o, d, sld, hit, recurs), w)
        if m[0] == "blend": return caweight(self.mpostshade(lz, m[1], o, d, sld, hit, recurs), self.mpostshade(lz, m[2], o, d, sld, hit, recurs), m[3])
        raise ValueError(m[0])

    def trace(self, sld, o, d, depth, recurs, secondary=False, lights=None):  # Trace.hs:59-82 (Q16) -> (ColorA, Rayint)
        if recurs == 0: return (0.0, 0.0, 0.0, 0.0), None
        if secondary: self.rays[2] += 1
        hit = sld.rayint(o, d, depth, ())
        if hit is None: return (0.0, 0.0, 0.0, 0.0), None
        lz = [None, lights]  # (the lazily evaluated light list of this hit, the Light list it is made from)
        acc = (0.0, 0.0, 0.0, 0.0)
        for t in hit[3]:
            if acc[3] + DELTA >= 1: break  # opaque (Trace.hs:50-51)
            acc = cafold(acc, self.mpostshade(lz, t, o, d, sld, hit, recurs))
        return acc, hit

    # ---- Scene.hs:48-57, Glome.hs:27-33, 119-128, 162-176
    def set_camera(self, pos, at, up, angle):
        fwd = vnorm(vsub(tuple(at), tuple(pos)))
        right = vnorm(vcross(tuple(up), fwd))
        up_ = vnorm(vcross(fwd, right))
        s = math.tan((math.pi / 180) * (angle / 2))
        self.cam = (tuple(pos), fwd, vscale(up_, s), vscale(right, s))

    def set_camera_vectors(self, pos, fwd, up, right): self.cam = (tuple(pos), tuple(fwd), tuple(up), tuple(right))

    def render(self, width, height, maxdepth):
        """renderTile over the whole frame: rows of (r, g, b, a, depth) -- the get_color tuple, without the d/400 debug term."""
        self.rays = [0, 0, 0]
        pos, fwd, up, right = self.cam
        sld = self.nodes[self.root]
        out = []
        for py in range(height):
            row = []
            for px in range(width):
                x = (((px / width) * 2) - 1) * (width / height)
                y = -(((py / height) * 2) - 1)
                v = (fwd[0] + right[0] * (-x) + up[0] * y, fwd[1] + right[1] * (-x) + up[1] * y, fwd[2] + right[2] * (-x) + up[2] * y)  # vadd3
                self.rays[0] += 1
                c, hit = self.trace(sld, pos, vnorm(v), INF, maxdepth)
                row.append((c[0], c[1], c[2], c[3], ridepth(hit)))
            out.append(row)
        return out

    # ---- renderTileSubsample, Glome.hs:179-323, over the tile map of renderTiles (:371-386)
    def get_color(self, xc, yc, maxdepth):  # Glome.hs:27-33 + 53-55: (ColorA, depth) of one camera-space sample
        pos, fwd, up, right = self.cam
        v = (fwd[0] + right[0] * (-xc) + up[0] * yc, fwd[1] + right[1] * (-xc) + up[1] * yc, fwd[2] + right[2] * (-xc) + up[2] * yc)
        self.rays[0] += 1
        c, hit = self.trace(self.nodes[self.root], pos, vnorm(v), INF, maxdepth)
        return (c[0], c[1], c[2], c[3], ridepth(hit))

    def render_subsample(self, width, height, maxdepth, blocksize=65, thresholds=(0.14, 0.15, 0.16, 0.18)):
        """renderTiles with renderTileSubsample: every tile of `chunk width blocksize` x `chunk height blocksize` sampled
        adaptively in its own buffer (neighbours outside the tile read (0,0,0,0,infinity)), then blitted."""
        self.rays = [0, 0, 0]
        blank = (0.0, 0.0, 0.0, 0.0, INF)

        def chunk(size):  # :371-377
            out, pos = [], 0
            while True:
                if pos + blocksize >= size:
                    out.append((pos, size - pos)); return out
                out.append((pos, blocksize)); pos += blocksize

        def coords(xf, yf):  # getCoords / getCoordsf, :119-140
            return (((xf / width) * 2) - 1) * (width / height), -(((yf / height) * 2) - 1)

        def ccmp(p, q):  # cCmp, :179-189
            def muldiff(a, b):
                if a == 0 and b == 0: return 0.0
                return (fdiv(a, b) - 1) if a > b else (fdiv(b, a) - 1)
            return abs(q[0] - p[0]) + abs(q[1] - p[1]) + abs(q[2] - p[2]) + abs(q[3] - p[3]) + muldiff(p[4], q[4])

        def cavg(a, b, c, d): return tuple((a[k] + b[k] + c[k] + d[k]) * 0.25 for k in range(5))  # :191-197
        def cavg2(a, b): return tuple((a[k] + b[k]) * 0.5 for k in range(5))                         # :199-205

        def decide(thr, xf, yf, a, b, c, d):  # :213-219
            if fmax(ccmp(a, c), ccmp(b, d)) > thr: return self.get_color(*coords(xf, yf), maxdepth)
            return cavg(a, b, c, d)

        frame = [[None] * width for _ in range(height)]
        for (xt, tw) in chunk(width):
            for (yt, th) in chunk(height):
                v = {}

                def getc(buf, x, y):
                    if xt <= x < xt + tw and yt <= y < yt + th: return buf.get((x, y), blank)  # MUV.replicate ... blank
                    return blank

                t1, t2, t3, t4 = thresholds
                for x in range(xt, xt + tw, 2):          # :241-250: the even lattice, (dx + dy) mod 4 == 0, always traced
                    for y in range(yt, yt + th, 2):
                        if ((x - xt) + (y - yt)) % 4 == 0: v[(x, y)] = self.get_color(*coords(float(x), float(y)), maxdepth)
                for x in range(xt, xt + tw, 2):          # :251-263: the rest of the even lattice from its 4 neighbours two away
                    for y in range(yt, yt + th, 2):
                        if ((x - xt) + (y - yt)) % 4 == 2:
                            v[(x, y)] = decide(t1, float(x), float(y), getc(v, x - 2, y), getc(v, x, y + 2), getc(v, x + 2, y), getc(v, x, y - 2))
                for x in range(xt + 1, xt + tw, 2):      # :273-283: odd-odd pixels from their diagonals
                    for y in range(yt + 1, yt + th, 2):
                        v[(x, y)] = decide(t2, float(x), float(y), getc(v, x - 1, y - 1), getc(v, x + 1, y - 1), getc(v, x + 1, y + 1), getc(v, x - 1, y + 1))
                for x in range(xt, xt + tw):             # :285-297: the remaining pixels from their 4 neighbours
                    for y in range(yt, yt + th):
                        if ((x - xt) + (y - yt)) % 2 == 1:
                            v[(x, y)] = decide(t3, float(x), float(y), getc(v, x - 1, y), getc(v, x, y + 1), getc(v, x + 1, y), getc(v, x, y - 1))
                v2 = {}
                for x in range(xt, xt + tw):             # :299-319: supersample between pixels, blend into the copy
                    for y in range(yt, yt + th):
                        a, b, c, d = getc(v, x, y), getc(v, x, y + 1), getc(v, x + 1, y + 1), getc(v, x + 1, y)
                        color = decide(t4, x + 0.5, y + 0.5, a, b, c, d)
                        if x == xt + tw - 1: v2[(x, y)] = color if y == yt + th - 1 else cavg2(color, cavg2(a, b))
                        else: v2[(x, y)] = cavg2(color, cavg2(a, d)) if y == yt + th - 1 else cavg2(color, cavg(a, b, c, d))
                for (x, y), c in v2.items(): frame[y][x] = c
        return frame


def load(sd):
    """SceneDesc -> Scene (the same constants the C++ oracle and the product receive)."""
    sc = Scene()
    nm, _ = sd.replay(sc)
    sc.root = nm[sd.root]
    sc.lights = [(tuple(p), tuple(c), r, s) for (p, c, r, s) in sd.lights]
    sc.set_camera(*sd.cam)
    return sc, nm
