"""Benchmark / parity scenes S1..S5, written in TestScene.hs vocabulary
(GlomeView/TestScene.hs:17-19 lights, :138 cust_cam, :201-245 materials).  Every generator uses only IEEE-exact
+ - * / and integer hashing, evaluated in double and rounded to fp32 by SceneDesc.
"""
import os

import numpy as np

from . import api
from .scene import SceneDesc

LIGHTS = [((-100.0, 70.0, 140.0), (7000.0 * 1.0, 7000.0 * 0.8, 7000.0 * 0.8)),  # TestScene.hs:17
          ((-3.0, 5.0, 8.0), (10.0 * 1.5, 10.0 * 2.0, 10.0 * 2.0))]             # TestScene.hs:18
CUST_CAM = ((-2.0, 4.3, 15.0), (0.0, 2.0, 0.0), (0.0, 1.0, 0.0), 45.0)          # TestScene.hs:138


def _common(sd, nlights):
    for pos, col in LIGHTS[:nlights]:
        sd.add_light(pos, col)
    sd.set_camera(*CUST_CAM)


def materials(sd):
    m = {}
    m["shiny_white"] = sd.material_surface((1, 1, 1), 1, 0.2, 0.8, 0.4, 10)  # TestScene.hs:201-202
    m["shiny_red"] = sd.material_surface((1, 0, 0), 1, 0.2, 0.8, 0.4, 10)    # TestScene.hs:204
    m["mirror"] = sd.material_reflect(0.8)                                   # TestScene.hs:243
    return m


def matte(sd, c):  # m_matte, TestScene.hs:236-237
    return sd.material_surface(c, 1, 0.2, 1, 0, 0)


def s1(nlights=1, shadows=True, n=5):
    """C.1: group [tex plane matte-green, bih [tex (sphere (x,0.5,z) 0.4) mat | x,z in -n..n]]  (S1/S2)."""
    sd = SceneDesc()
    m = materials(sd)
    mats = [m["shiny_white"], m["shiny_red"], matte(sd, (0.5, 0, 1))]
    green = matte(sd, (0, 0.8, 0.3))
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), green)
    sph = []
    for ix, x in enumerate(range(-n, n + 1)):
        for iz, z in enumerate(range(-n, n + 1)):
            sph.append(sd.tex(sd.sphere((float(x), 0.5, float(z)), 0.4), mats[(ix + iz) % 3]))
    root = sd.group([pl, sd.bih(sph)])
    sd.set_root(root)
    _common(sd, nlights)
    if not shadows:
        sd.lights = [(p, c, r, False) for (p, c, r, _) in sd.lights]
    return sd


def heightfield_vertices(N):
    """C.2: (N+1)^2 vertices, evaluated in double."""
    i = np.arange(N + 1, dtype=np.int64)

    def s(k):
        u = (k % 32).astype(np.float64) / 32.0
        sign = np.where((k // 32) % 2 == 0, 1.0, -1.0)
        return sign * (4.0 * u * (1.0 - u))

    x = i.astype(np.float64) * 20.0 / N - 10.0
    I, J = np.meshgrid(i, i, indexing="ij")
    h = ((I.astype(np.uint64) * 73856093) & 0xFFFFFFFF).astype(np.uint32) ^ ((J.astype(np.uint64) * 19349663) & 0xFFFFFFFF).astype(np.uint32)
    y = 1.5 * s(I) * s(J + 17) + 0.05 * ((h & 1023).astype(np.float64) / 1024.0)
    V = np.stack([x[I], y, x[J]], axis=-1)  # V[i, j] = (x_i, y_ij, z_j)
    return V


def heightfield_triangles(N):
    """Cell (i,j) -> (v(i,j), v(i,j+1), v(i+1,j)) and (v(i+1,j), v(i,j+1), v(i+1,j+1)); normals face +y."""
    V = heightfield_vertices(N)
    a, b, c, d = V[:-1, :-1], V[:-1, 1:], V[1:, :-1], V[1:, 1:]
    t1 = np.concatenate([a, b, c], axis=-1)
    t2 = np.concatenate([c, b, d], axis=-1)
    return np.stack([t1, t2], axis=2).reshape(-1, 9)


def s3(N=224, as_mesh=False, nlights=1):
    """C.2: heightfield of 2*N*N triangles as `tex (bih (map triangle ...)) matte` or as `mesh` (S3: N=224, S5: N=708)."""
    sd = SceneDesc()
    mat = matte(sd, (0.8, 0.5, 0.4))
    if as_mesh:
        V = heightfield_vertices(N).reshape(-1, 3)
        idx = np.arange((N + 1) * (N + 1)).reshape(N + 1, N + 1)
        a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1], idx[1:, 1:]
        t1 = np.stack([a, b, c], -1)
        t2 = np.stack([c, b, d], -1)
        tri = np.stack([t1, t2], axis=2).reshape(-1, 3)
        tris = np.full((tri.shape[0], 8), -1, dtype=np.int32)
        tris[:, :3] = tri
        tris[:, 6] = 0  # Tri a b c (-1) (-1) (-1) 0 (-1)
        root = sd.mesh(V, np.zeros((0, 3)), tris, [mat])
    else:
        ids = sd.triangles_bulk(heightfield_triangles(N))
        root = sd.tex(sd.bih(ids), mat)
    sd.set_root(root)
    _common(sd, nlights)
    return sd


def sphereint(sd):  # TestScene.hs:112-115
    return sd.intersection([sd.sphere((-1, 0, 0), 2), sd.sphere((1, 0, 0), 2), sd.sphere((0, -1, 0), 2), sd.sphere((0, 1, 0), 2)])


def s4(nlights=2):
    """C.3: CSG difference / intersection of spheres + boxes, Reflect 0.8, an instanced sphereint (S4)."""
    from . import api
    sd = SceneDesc()
    m = materials(sd)
    green = matte(sd, (0, 0.8, 0.3))
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), green)
    items = [
        sd.difference(sd.tex(sd.sphere((0, 1.5, 0), 1.5), m["mirror"]), sd.sphere((0.9, 2.2, 1.0), 1.0)),
        sd.difference(sd.tex(sd.box((-4, 0, -1), (-2, 2, 1)), matte(sd, (0.4, 0.4, 0.8))), sd.sphere((-3, 1, 1), 0.9)),
        sd.tex(sd.intersection([sd.sphere((3, 1, 0), 1.2), sd.box((2.2, 0.2, -0.8), (3.8, 1.8, 0.8))]), m["shiny_red"]),
        sd.transform(sd.tex(sphereint(sd), matte(sd, (0.5, 0, 1))), [api.scale(np.float32([0.6, 0.6, 0.6])), api.translate(np.float32([-5.2, 1, 5]))]),
        sd.tex(sd.difference(sd.sphere((0, -4, 5), 4.7), sd.sphere((1.5, 1.5, 5.2), 1.6)), m["mirror"]),  # TestScene.hs:127 (Q13 texture loss)
    ]
    root = sd.group([pl, sd.bih(items)])
    sd.set_root(root)
    _common(sd, nlights)
    return sd


def _polyhedron(sd, points, pos, r, name):
    """TestScene.hs:29-54: a sphere cut by one plane per direction"""
    pos = np.array(pos, dtype=np.float64)
    planes = []
    for p in points:
        n = np.array(p, dtype=np.float64)
        n = n / np.sqrt(n @ n)
        planes.append(sd.plane_offset(n, r + float(n @ pos)))
    return sd.tag(sd.intersection([sd.sphere(pos, 1.26 * r)] + planes), name)


class _Draws:
    """The random draws of `oak` (TestScene.hs:68-110).  The reference takes them from System.Random (`mkStdGen 42`, `split`,
    `randomR`): the `random` package is not part of the reference tree and its generator differs between versions, so the
    draws are SUPPLIED AS DATA by a stated generator instead -- NOT GHC's: a SplitMix64 step per draw, `split` = two states
    hashed from the parent's.  Same tree shape and the same ranges as the reference; the individual angles and lengths are
    this generator's."""
    M = (1 << 64) - 1

    def __init__(self, state):
        self.s = state & self.M

    @staticmethod
    def _mix(z):
        M = _Draws.M
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    def split(self):
        return _Draws(self._mix(self.s ^ 0x1234567)), _Draws(self._mix(self.s ^ 0x7654321))

    def randomR(self, lo, hi):
        z = self._mix(self.s)
        u = (z >> 11) / float(1 << 53)
        return lo + (hi - lo) * u, _Draws(z)


def oak(sd, age, seed=42):
    """TestScene.hs:68-110: a tree that branches once per year -- `tree n` = a cone segment and two scaled, rotated copies of
    `tree (n - 1)` under a bounding sphere (bound_object), `tree 1` a green sphere; the whole flattened (flatten_transform,
    tolist) under one bih, matte brown.  age 11.4: 1,023 cones and 1,024 spheres."""
    if age < 0:
        return sd.group([])  # nothing
    year = int(np.floor(age))
    season = age - year
    thickness, minbranch, maxbranch = 0.03, api.deg(10), api.deg(25)
    leaf_mat = matte(sd, (0.2, 1, 0.4))

    def tree(n_, r):
        if n_ == 0:
            return sd.group([])
        if n_ == 1:
            return sd.tex(sd.sphere((0, 0, 0), season), leaf_mat)
        nf = float(n_)
        rng1, rng2 = r.split()
        rng3, rng4 = rng1.split()
        r1, rng5 = rng4.randomR(0.0, 0.5)
        r2, rng6 = rng5.randomR(minbranch, maxbranch)
        r3, rng7 = rng6.randomR(0.8, 0.95)
        r4, _ = rng7.randomR(0.0, 1.0)
        seglen, branchang, scaling = 0.5 + r1, r2, r3
        height, n = (nf / 2, int(np.ceil(nf / 2))) if r4 > 1.0 else (nf, n_)  # (never: r4 is in [0, 1], as in the reference)
        kids = [sd.cone((0, 0, 0), thickness * height, (0, seglen, 0), thickness * (height - 1) * scaling)]
        for sub, ang in ((rng2, branchang), (rng3, -branchang)):
            kids.append(sd.transform(tree(n - 1, sub), [api.scale((scaling, scaling, scaling)), api.rotate((0, 0, 1), ang), api.rotate((0, 1, 0), api.deg(30)),
                                                        api.translate((0, seglen, 0))]))
        return sd.bound_object(sd.sphere((0, height / 2, 0), height / 2), sd.group(kids))

    return sd.tag(sd.tex(sd.bih_tolist(sd.flatten_transform(tree(year, _Draws(seed)))), matte(sd, (0.8, 0.5, 0.4))), "tree")


def testscene(lattice_n=10, with_oak=True, skip=()):
    """GlomeView's own default scene, `geom''` of TestScene.hs:183-197 with TestScene's lights, camera and textures, every
    item of it (the `oak`'s random draws come from a stated generator, not GHC's: see _Draws): the chessboard of 64 textured boxes carved by a sphere (a Difference whose first operand is a
    transformed group), the dodecahedron and the transformed icosahedron (Intersections of a sphere with 12 / 20 planes)
    under the stripe and the perlin Blend textures, a cone, the lattice of (2n+1)^3 spheres under its own bih, rotated,
    scaled and hollowed out by a sphere, the portal (a Warp material looking into this very scene) inside a transform, and
    a refracting sphere squashed by a non-uniform scale.  Nesting reaches four composite levels below the root bih."""
    sd = SceneDesc()
    m = materials(sd)
    dull_gray = sd.material_surface((0.4, 0.3, 0.35), 1, 0.2, 0.8, 0, 0)                                        # :211
    mottled = sd.material_blend_fn(m["mirror"], matte(sd, (0.15, 0.3, 0.5)), api.WEIGHT_PERLIN, [3.0])  # :213-220
    stripe = sd.material_blend_fn(m["shiny_white"], dull_gray, api.WEIGHT_STRIPE_TRIANGLE, [4, 8, 5])          # :225-231
    # chessboard, TestScene.hs:140-150
    xs = [-3.5 + i for i in range(8)]
    squares = []
    for x in xs:
        for z in xs:
            white = (int(np.floor(x)) + int(np.floor(z))) % 2 == 0
            squares.append(sd.tex(sd.box((x - 0.5, -3, z - 0.5), (x + 0.5, (x * z) / 40, z + 0.5)), m["shiny_white"] if white else mottled))
    chessboard = sd.group(squares)
    carved_board = sd.difference(sd.transform(chessboard, [api.scale((2, 1.2, 2))]), sd.tex(sd.sphere((4, 1.5, 3), 3.5), m["shiny_white"]))  # :185
    gr = (1 + 5 ** 0.5) / 2
    r = 1.0
    dod_pts = [(0, y, z) for y in (-r, r) for z in (-gr * r, gr * r)] + [(x, 0, z) for z in (-r, r) for x in (-gr * r, gr * r)] + \
              [(x, y, 0) for x in (-r, r) for y in (-gr * r, gr * r)]
    dodeca = sd.tex(_polyhedron(sd, dod_pts, (-6, 3, 0), r, "dodecahedron"), stripe)                              # :186
    r = 1.5
    ico_pts = [(x, y, z) for x in (-r, r) for y in (-r, r) for z in (-r, r)] + [(0, y, z) for y in (-r / gr, r / gr) for z in (-gr * r, gr * r)] + \
              [(x, y, 0) for x in (-r / gr, r / gr) for y in (-gr * r, gr * r)] + [(x, 0, z) for x in (-gr * r, gr * r) for z in (-r / gr, r / gr)]
    icosa = sd.tex(sd.transform(_polyhedron(sd, ico_pts, (4, 1.5, 3), r, "icosahedron"), [api.rotate((0, 0, 1), api.deg(11)), api.rotate((1, 0, 0), api.deg(7))]), mottled)  # :187-188
    cone = sd.cone((-6, -1, 0), 0.7, (-6, 3, 0), 0)                                                              # :189
    n = int(lattice_n)
    lattice = sd.bih([sd.sphere((float(x), float(y), float(z)), 0.2) for x in range(-n, n + 1) for y in range(-n, n + 1) for z in range(-n, n + 1)])  # :21-26
    hollow = sd.tex(sd.difference(sd.transform(lattice, [api.rotate((0, 0, 1), api.deg(23)), api.rotate((1, 0, 0), api.deg(43)), api.scale((3, 3, 3))]),
                                  sd.sphere((0, 0, 0), 3.2 * n)), m["shiny_red"])                                 # :191-193 (sphere 32 for n = 10)
    # portal 5 2 (1/3), TestScene.hs:152-181, 194-195
    w, h, th, dl = 2.0, 5.0, 1.0 / 3.0, 1e-4
    frame = sd.tag(sd.tex(sd.difference(sd.box((-w, 0, -th), (w, h, th)), sd.box((th - w, th, -(th + dl)), (w - th, h - th, th + dl))), matte(sd, (0.4, 0.4, 0.8))), "door frame")
    surface = sd.box((-w, 0, -dl), (w, h - dl, dl))
    warp = sd.material_warp(frame, None, LIGHTS, api.compose([api.rotate((1, 0, 0), api.deg(-85)), api.translate((8, 40, -4))]))
    door = sd.transform(sd.group([frame, sd.tex(surface, warp)]), [api.rotate((0, 1, 0), api.deg(8)), api.translate((-3, 0.5, -5))])
    glass = sd.transform(sd.tex(sd.sphere((-2.3, 0.3, 4.2), 1.7), sd.material_refract(0.35, 0.8, 1.5)), [api.scale((1, 0.4, 1))])  # :196
    if os.environ.get("GLOME_TS_PLAIN_BOARD"):  # (measurement only, like `skip`: the board left whole -- what its Difference costs on top of its 64 boxes)
        carved_board = sd.transform(chessboard, [api.scale((2, 1.2, 2))])
    items = [carved_board, dodeca, icosa, cone]
    if with_oak:
        items.append(sd.transform(oak(sd, 11.4, 42), [api.scale((2, 2, 2)), api.translate((2, -1, -8))]))                 # :190
    items += [hollow, door, glass]
    if skip:  # (measurement only, tools/probe/ts_parts.sh: the scene without some of its items, by position in the list above)
        items = [it for k, it in enumerate(items) if k not in skip]
    sd.set_root(sd.bih(items))
    _common(sd, 2)
    if with_oak:
        # Bounds the parity checks read (tests/parity.py): every cone segment of the oak ends inside the next one and inside the two
        # child segments' starts, and 1,024 leaf spheres sit on the twig ends -- surfaces that coincide to within fp32 rounding, so
        # which of two primitives an fp32 ray reports at a joint differs from the fp64 checker's on 2-4 % of the hits (same distance:
        # no hit / miss flips, no depth outliers), and a pixel on a joint takes the other primitive's normal
        # (and a twig-end sphere is 0.1-0.2 across, ten transforms deep, 12 units from the eye: 2e-4 of hit distance is 3e-3 of its normal)
        # The adaptive sampler decides per pixel by a contrast threshold: twigs a pixel wide put many contrasts near it, and one flipped
        # decision moves a pixel and the neighbours blended from it (measured on the GPU at 260x195: 1.4 % of the pixels beyond 1e-4)
        # (the wide bounds are bounds against the fp64 checker ALONE; that these pixels are rounding is asked by parity.away_beyond_rounding, in both suites)
        sd.subsample_outlier_max, sd.rel_outlier_max = 2.0e-2, 2.0e-2  # (measured on the GPU: 1.36e-2)
        sd.same_prim_min, sd.pixel_outlier_max, sd.pixel_mean_max, sd.normal_atol = 0.95, 6e-3, 4e-4, 2e-2  # (measured on the GPU at 720x480: 4.2e-3, 2.8e-4)
    return sd


CONFIGS = {
    "S1": dict(make=lambda: s1(nlights=2, shadows=False), width=720, height=480, maxdepth=1),
    "S2": dict(make=lambda: s1(nlights=1), width=720, height=480, maxdepth=1),
    "S3": dict(make=lambda: s3(224), width=1920, height=1080, maxdepth=1),
    "S3mesh": dict(make=lambda: s3(224, as_mesh=True), width=1920, height=1080, maxdepth=1),
    "S4": dict(make=lambda: s4(), width=1920, height=1080, maxdepth=3),
    "S5": dict(make=lambda: s3(708), width=3840, height=2160, maxdepth=1),
    "S5mesh": dict(make=lambda: s3(708, as_mesh=True), width=3840, height=2160, maxdepth=1),  # BASELINE configs[4] as written: a 1M-triangle Mesh
    "TS": dict(make=lambda: testscene(10), width=720, height=480, maxdepth=3),  # GlomeView's default scene at its default window (Glome.hs:112-113)
    "TSnooak": dict(make=lambda: testscene(10, with_oak=False), width=720, height=480, maxdepth=3),  # the same without the oak: what rounds 1-2 timed as TS
    # (probe: GlomeView's default scene without the items whose positions GLOME_TS_SKIP lists -- 0 board, 1 dodecahedron, 2 icosahedron,
    # 3 cone, 4 oak, 5 lattice, 6 door, 7 glass sphere)
    "TSparts": dict(make=lambda: testscene(10, skip=tuple(int(x) for x in os.environ.get("GLOME_TS_SKIP", "").split(",") if x)), width=720, height=480, maxdepth=3),
}
