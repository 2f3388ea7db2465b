"""A zoo of small scenes in TestScene.hs vocabulary that together reach every primitive, composite, wrapper and
material kind on the hot path.  Used by the host-side and the GPU parity tests."""
import numpy as np

from glome_amd import api, scenes
from glome_amd.scene import SceneDesc


def _finish(sd, root, nlights=2):
    sd.set_root(root)
    for pos, col in scenes.LIGHTS[:nlights]:
        sd.add_light(pos, col)
    sd.set_camera(*scenes.CUST_CAM)
    return sd


def flat_mixed():
    """flat tier, mixed leaf class: bih of simple primitives with Tex stacks and shadow flags, plus a plane and a disc."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    blue = scenes.matte(sd, (0.2, 0.3, 0.9))
    items = [
        sd.tex(sd.sphere((-3, 1, 0), 1.0), m["shiny_red"]),
        sd.tex(sd.tex(sd.sphere((0, 1.2, -1), 1.2), m["shiny_white"]), blue),  # nested Tex: innermost first
        sd.tex(sd.triangle((1, 0.1, 2), (3, 0.2, 2.5), (2, 2.5, 1.5)), blue),
        sd.tex(sd.trianglenorm((-2, 0.2, 3), (0, 0.3, 3.5), (-1, 2.0, 3.2), (0, 0, 1), (0.6, 0, 0.8), (0, 0.6, 0.8)), m["shiny_white"]),
        sd.tex(sd.box((3, 0, -3), (4.5, 2, -1.5)), m["shiny_red"]),
        sd.tex(sd.disc((-4.5, 1.5, 2), (0.3, 0.9, 0.3162277), 0.9), blue),
        sd.noshadow(sd.tex(sd.sphere((2, 3.5, 0), 0.7), blue)),
        sd.onlyshadow(sd.box((-1, 3, -0.5), (0, 3.4, 0.5))),
        sd.tag(sd.tex(sd.sphere((5, 0.5, 3), 0.5), m["shiny_white"]), "tagged"),
    ]
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0, 0.8, 0.3)))
    free = sd.tex(sd.disc((0, 4.5, -4), (0, 0.7071068, 0.7071068), 1.5), m["shiny_red"])
    return _finish(sd, sd.group([pl, sd.bih(items), free]))


def quadrics():
    """cylinders / cones (canonical z-axis primitives inside Instances, Cone.hs:40-67), every cap case."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0, 0.8, 0.3)))
    items = [
        sd.tex(sd.cylinder((-3, 0.2, 0), (-3, 2.5, 0.5), 0.6), m["shiny_red"]),
        sd.tex(sd.cone((0, 0.1, 0), 1.0, (0.5, 3.0, 0), 0.0), m["shiny_white"]),        # TestScene.hs:126 style
        sd.tex(sd.cone((3, 0.5, 1), 0.3, (4.5, 2.0, 2), 0.9), scenes.matte(sd, (0.5, 0, 1))),  # swapped radii (r1 < r2)
        sd.tex(sd.cone((-1, 3, -2), 0.5, (1, 3.2, -2), 0.50001), m["shiny_red"]),       # degenerates to a cylinder (r1-r2 < delta)
    ]
    return _finish(sd, sd.group([pl, sd.bih(items)]))


def csg():
    """S4 plus a plane-cut polyhedron (TestScene.hs:45-54), Bound / InnerBound, and nested differences."""
    sd = scenes.s4()
    # rebuild root with extras appended to a new outer group
    m_gold = sd.material_surface((0.9, 0.7, 0.2), 1, 0.2, 0.8, 0.4, 10)
    gr = (1 + 5 ** 0.5) / 2
    pos, r = np.array([6.0, 1.5, -2.0]), 1.0
    pts = [(0, y, z) for y in (-r, r) for z in (-gr * r, gr * r)] + [(x, 0, z) for z in (-r, r) for x in (-gr * r, gr * r)] + \
          [(x, y, 0) for x in (-r, r) for y in (-gr * r, gr * r)]
    planes = []
    for p in pts:
        n = np.array(p, dtype=np.float64); n /= np.linalg.norm(n)
        planes.append(sd.plane_offset(n, r + float(n @ pos)))
    dodeca = sd.tex(sd.tag(sd.intersection([sd.sphere(pos, 1.26 * r)] + planes), "dodecahedron"), m_gold)  # TestScene.hs:45-54
    bounded = sd.bound_object(sd.sphere((-6, 2, -3), 1.6), sd.tex(sd.group([sd.sphere((-6.5, 2, -3), 0.8), sd.box((-6, 1.5, -3.5), (-5, 2.5, -2.5))]), m_gold))
    inner = sd.innerbound(sd.sphere((0, 5, -6), 0.5), sd.tex(sd.sphere((0, 5, -6), 1.0), m_gold))
    nested = sd.tex(sd.difference(sd.difference(sd.box((7, 0, 2), (9, 2, 4)), sd.sphere((8, 2, 3), 0.8)), sd.box((7.5, 0.5, 1.5), (8.5, 1.5, 2.5))), m_gold)
    root = sd.group([sd.root, sd.bih([dodeca, bounded, inner, nested])])
    sd.set_root(root)
    return sd


def nested():
    """bih -> instance -> bih (TestScene.hs lattice style), group inside an instance, a difference whose operand is an
    instanced bih (TestScene.hs:191-193), a rotated + scaled instance, Tex stacks three deep."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    teal, pink = scenes.matte(sd, (0.1, 0.7, 0.7)), scenes.matte(sd, (1, 0.4, 0.7))
    lattice = sd.bih([sd.sphere((float(x), float(y), float(z)), 0.3) for x in range(-2, 3) for y in range(-2, 3) for z in range(-2, 3)])
    carved = sd.tex(sd.difference(sd.transform(lattice, [api.rotate((0, 0, 1), api.deg(23)), api.rotate((1, 0, 0), api.deg(43)), api.scale((0.8, 0.8, 0.8)),
                                                         api.translate((-3, 2.5, 0))]), sd.sphere((-3, 2.5, 0), 1.2)), m["shiny_red"])
    pair = sd.group([sd.tex(sd.sphere((0, 0, 0), 0.6), teal), sd.tex(sd.box((0.5, -0.4, -0.4), (1.4, 0.4, 0.4)), pink)])
    inst_pair = sd.transform(pair, [api.scale((1.5, 0.7, 1.2)), api.rotate((0, 1, 0), api.deg(30)), api.translate((3, 1.0, 1))])
    deep = sd.tex(sd.tex(sd.tex(sd.sphere((0, 1, 4), 0.8), teal), pink), m["shiny_white"])
    instinst = sd.transform(sd.transform(sd.tex(sd.sphere((0, 0, 0), 0.5), pink), [api.translate((1, 0, 0))]), [api.scale((2, 2, 2)), api.translate((2, 3, -3))])  # merges (Solid.hs:494-496)
    tri_moved = sd.tex(sd.transform(sd.triangle((0, 0, 0), (1, 0, 0), (0, 1, 0)), [api.scale((2, 2, 2)), api.translate((-6, 0.5, 3))]), teal)  # bakes (Triangle.hs:164-168)
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0, 0.8, 0.3)))
    return _finish(sd, sd.group([pl, sd.bih([carved, inst_pair, deep, instinst, tri_moved])]))


def mesh_scene():
    """mesh with vertex normals on some triangles and per-triangle textures (Mesh.hs:27-29, 143-161), plus a sphere."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    N = 12
    V = scenes.heightfield_vertices(N).reshape(-1, 3) * np.array([0.5, 1.0, 0.5])
    idx = np.arange((N + 1) * (N + 1)).reshape(N + 1, N + 1)
    a, b, c, d = idx[:-1, :-1], idx[:-1, 1:], idx[1:, :-1], idx[1:, 1:]
    tri = np.stack([np.stack([a, b, c], -1), np.stack([c, b, d], -1)], axis=2).reshape(-1, 3)
    # vertex normals: normalised (0,1,0) + small tilt by position
    nrm = np.stack([0.3 * np.sin(V[:, 0]), np.ones(len(V)), 0.3 * np.cos(V[:, 2])], -1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tris = np.full((len(tri), 8), -1, np.int32)
    tris[:, :3] = tri
    smooth = np.arange(len(tri)) % 3 == 0
    tris[smooth, 3:6] = tri[smooth]
    tris[:, 6] = np.arange(len(tri)) % 2  # alternate two textures
    tris[::7, 6] = -1                     # some triangles carry no texture of their own
    mats = [scenes.matte(sd, (0.8, 0.5, 0.4)), m["shiny_white"]]
    me = sd.tex(sd.mesh(V, nrm, tris, mats), scenes.matte(sd, (0.3, 0.3, 0.3)))
    sp = sd.tex(sd.sphere((0, 2.5, 0), 0.8), m["shiny_red"])
    return _finish(sd, sd.group([me, sp]))


def materials():
    """every Material kind (Shader.hs:43-52 minus Warp): Surface, Reflect, Refract, Blend, AdditiveLayers, stacked Tex."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    glass = sd.material_refract(0.35, 0.8, 1.5)                      # TestScene.hs:196
    blend = sd.material_blend(m["mirror"], scenes.matte(sd, (0.15, 0.3, 0.5)), 0.4)  # t_mottled at a fixed weight (TestScene.hs:220)
    layers = sd.material_layers([sd.material_surface((1, 0.2, 0.2), 0.5, 0.3, 0.7, 0.3, 8), m["mirror"]])
    thin = sd.material_surface((0.2, 0.9, 0.2), 0.4, 0.5, 0.5, 0, 0)  # translucent: the fold continues to the next Tex
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0, 0.8, 0.3)))
    items = [
        sd.tex(sd.sphere((-4, 1, 0), 1.0), m["mirror"]),
        sd.tex(sd.sphere((-1.5, 1, 1), 1.0), glass),
        sd.tex(sd.sphere((1, 1, 0), 1.0), blend),
        sd.tex(sd.sphere((3.5, 1, 1), 1.0), layers),
        sd.tex(sd.tex(sd.box((-1, 0, -4), (1, 2, -3)), thin), m["shiny_red"]),  # thin over red
        sd.tex(sd.box((5, 0, -2), (6, 3, 2)), m["mirror"]),
    ]
    return _finish(sd, sd.group([pl, sd.bih(items)]))


def textures():
    """Solid-texture Blend weights (TestScene.hs:214-234): t_mottled = Blend mirror matte (perlin (3 * pos)), t_stripe =
    Blend shiny_white dull_gray (triangle_wave (pos . (4,8,5))), plus the square and sine waves of Texture.hs:11-24."""
    from glome_amd import api
    sd = SceneDesc()
    m = scenes.materials(sd)
    shiny_white = sd.material_surface((1, 1, 1), 1, 0.2, 0.8, 0.4, 10)       # TestScene.hs:199
    dull_gray = sd.material_surface((0.4, 0.3, 0.35), 1, 0.2, 0.8, 0, 0)     # TestScene.hs:211
    mottled = sd.material_blend_fn(m["mirror"], scenes.matte(sd, (0.15, 0.3, 0.5)), api.WEIGHT_PERLIN, [3.0])
    stripe = sd.material_blend_fn(shiny_white, dull_gray, api.WEIGHT_STRIPE_TRIANGLE, [4, 8, 5])
    square = sd.material_blend_fn(shiny_white, m["shiny_red"], api.WEIGHT_STRIPE_SQUARE, [0, 3, 0])
    sine = sd.material_blend_fn(dull_gray, m["shiny_red"], api.WEIGHT_STRIPE_SINE, [2, 0, 1])
    # a bounded floor: on an infinite plane the horizon's hit points are thousands of units out, where fp32 positions no
    # longer resolve the noise lattice (3 * pos) -- a limit of fp32 hit points, not of the texture code
    pl = sd.tex(sd.box((-9, -0.5, -9), (9, 0, 9)), mottled)
    items = [
        sd.tex(sd.sphere((-4, 1.2, 0), 1.2), stripe),
        sd.tex(sd.sphere((-1, 1.2, 1), 1.2), mottled),
        sd.tex(sd.box((1.2, 0, -1), (3.2, 2, 1)), square),
        sd.tex(sd.cone((5, 0, 0), 1.0, (5, 2.5, 0), 0.2), sine),
    ]
    sd = _finish(sd, sd.group([pl, sd.bih(items)]))
    # square_wave / the stripe edges are step functions of the hit point: an fp32 hit point a few ulp from an edge takes the
    # other material (measured on MI355X: 25 of 57,600 pixels, errors up to 0.2); everything else is at the usual level
    sd.pixel_outlier_max, sd.pixel_mean_max, sd.rel_outlier_max = 1e-3, 2e-4, 1e-2
    return sd


def hall_of_mirrors():
    """The limits lifted in round 3 (the reference has none: Trace.hs:59-82 recurses to any maxdepth, Shader.hs:177-184 nests
    Blend / AdditiveLayers freely): two facing mirrors with objects between them, for maxdepth 8, and a sphere under a Blend of
    an AdditiveLayers of a Blend of an AdditiveLayers -- four levels of nested materials."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    red = sd.material_surface((0.9, 0.2, 0.2), 1, 0.2, 0.7, 0.3, 8)
    blue = scenes.matte(sd, (0.15, 0.3, 0.8))
    l1 = sd.material_layers([sd.material_surface((0.2, 0.9, 0.2), 0.5, 0.3, 0.6, 0.2, 6), blue])
    b2 = sd.material_blend(l1, red, 0.35)
    l3 = sd.material_layers([b2, sd.material_surface((0.9, 0.9, 0.1), 0.3, 0.2, 0.5, 0, 0)])
    b4 = sd.material_blend(l3, m["mirror"], 0.6)
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0.3, 0.7, 0.4)))
    items = [
        sd.tex(sd.box((-6, 0, -3.2), (6, 5, -3.0)), m["mirror"]),   # two walls of mirror facing each other
        sd.tex(sd.box((-6, 0, 3.0), (6, 5, 3.2)), m["mirror"]),
        sd.tex(sd.sphere((-1.5, 1, 0), 1.0), b4),
        sd.tex(sd.sphere((1.8, 0.8, 0.6), 0.8), red),
        sd.tex(sd.box((3.5, 0, -1), (4.2, 2.2, 0)), blue),
    ]
    sd.set_camera((-5.5, 2.2, 2.4), (1.0, 1.2, -3.1), (0, 1, 0), 55)
    return _finish(sd, sd.group([pl, sd.bih(items)]))


def veils(extra_materials=0):
    """Texture stacks of eight (Tex.hs:53-74 conses without a bound; the device stack holds 8 ids in a scene of at most 254
    materials, rt_types.h TexStack): a sphere, a Difference, BIH items and a mesh with per-triangle textures, each under Tex
    wrappers that add up to eight on the way down, the inner ones translucent so that every level reaches the pixel
    (Trace.hs:67-80 folds until opaque).  extra_materials pads the material table (more than 254: 16-bit ids, four levels)."""
    sd = SceneDesc()
    for k in range(extra_materials):
        sd.material_surface((0.001 * k, 0.5, 0.5), 1, 0.2, 0.7, 0, 0)
    cols = [(0.9, 0.2, 0.2), (0.2, 0.9, 0.2), (0.2, 0.2, 0.9), (0.9, 0.9, 0.1), (0.1, 0.9, 0.9), (0.9, 0.1, 0.9), (0.6, 0.6, 0.6)]
    veil = [sd.material_surface(c, 0.22 + 0.03 * k, 0.3, 0.7, 0.3 if k % 2 else 0.0, 6) for k, c in enumerate(cols)]
    solid = scenes.matte(sd, (0.8, 0.7, 0.5))
    deep = 4 if extra_materials > 254 else 8
    def under(n, k, first=0):  # k Tex wrappers, innermost first = veil[first], ...
        for j in range(k):
            n = sd.tex(n, veil[(first + j) % len(veil)])
        return n
    a = sd.tex(under(sd.sphere((-2.5, 1, 0), 1.0), deep - 1), solid)                      # eight straight down
    cut = sd.difference(sd.box((0.2, 0, -0.8), (1.8, 1.6, 0.8)), sd.sphere((1.0, 1.2, 0.0), 0.7))
    inner = sd.bih([under(sd.sphere((3.2 + 0.9 * i, 0.5, -1.5 + 0.8 * i), 0.45), 2, i) for i in range(4)] + [sd.tex(sd.box((3, 0, 1), (4, 0.8, 2)), veil[3])])
    V = np.array([[-1, 0.01, 2], [1, 0.01, 2], [-1, 1.5, 3], [1, 1.5, 3]], np.float64)
    tris = np.full((2, 8), -1, np.int32); tris[0, :3] = (0, 1, 2); tris[1, :3] = (2, 1, 3); tris[0, 6] = 0
    me = sd.mesh(V, np.zeros((0, 3)), tris, [veil[5]])
    half = deep // 2
    grp = sd.group([under(cut, half, 1), under(inner, half - 2, 2), under(me, half - 1, 4)])   # + the wrappers below: deep, deep, deep (- 1 on the mesh's bare triangle)
    grp = sd.tex(under(sd.transform(grp, [api.translate((0.0, 0.0, 0.3))]), deep - half - 1, 3), solid)
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0.3, 0.6, 0.4)))
    sd.set_camera((0.5, 3.0, 8.0), (0.8, 0.8, 0.0), (0, 1, 0), 50)
    return _finish(sd, sd.group([pl, a, grp]))


def instanced_terrain():
    """One triangle BIH (a 20x20-cell heightfield) used three times -- as it is, rotated and squashed by a non-uniform scale,
    and carved by a sphere -- beside a rotated lattice of spheres carved like GlomeView's, a mirror and a refracting sphere:
    the generic tier walks triangle and sphere BIHs inside its interpreter as packets (rt_generic.hpp, vm_run's packet
    service), for primary, shadow and reflected rays; the refracted ones are not unit length and keep the per-lane walk."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    sand, moss = scenes.matte(sd, (0.8, 0.6, 0.4)), scenes.matte(sd, (0.3, 0.6, 0.3))
    ids = sd.triangles_bulk(scenes.heightfield_triangles(20) * np.array([0.3, 0.6, 0.3] * 3))
    terrain = sd.bih(ids)
    t0 = sd.tex(terrain, sand)
    t1 = sd.tex(sd.transform(terrain, [api.rotate((0, 1, 0), api.deg(35)), api.scale((0.6, 1.8, 0.9)), api.translate((5.5, 0.4, -1.0))]), moss)
    t2 = sd.tex(sd.difference(sd.transform(terrain, [api.translate((-6.5, 0.2, 0.5))]), sd.sphere((-6.5, 0.6, 0.5), 1.6)), m["shiny_red"])
    lattice = sd.bih([sd.sphere((float(x), float(y), float(z)), 0.28) for x in range(-3, 4) for y in range(-3, 4) for z in range(-3, 4)])
    hollow = sd.tex(sd.difference(sd.transform(lattice, [api.rotate((0, 0, 1), api.deg(23)), api.rotate((1, 0, 0), api.deg(43)), api.scale((0.45, 0.45, 0.45)), api.translate((0.5, 3.2, -2.5))]),
                                  sd.sphere((0.5, 3.2, -2.5), 1.3)), scenes.matte(sd, (0.9, 0.8, 0.2)))
    mirror = sd.tex(sd.sphere((2.2, 1.4, 2.2), 0.9), m["mirror"])
    glass = sd.tex(sd.sphere((-2.0, 1.3, 3.0), 0.8), sd.material_refract(0.3, 0.8, 1.4))
    sd.set_camera((1.0, 5.5, 11.0), (0.0, 1.0, 0.0), (0, 1, 0), 50)
    return _finish(sd, sd.group([sd.bih([t0, t1, t2, hollow]), mirror, glass]))


ALL = {"flat_mixed": flat_mixed, "quadrics": quadrics, "csg": csg, "nested": nested, "mesh": mesh_scene, "materials": materials, "textures": textures, "veils": veils, "instanced_terrain": instanced_terrain}


def soup(n=2500, seed=5, spheres=False, floor=True):
    """A random soup (irregular tree: overlapping items, a clump of coincident triangles that ends up in leaves of more
    than six items) -- `tex (bih items) matte` lit by one shadow-casting light.  Exercises the packet walk on a tree
    that is nothing like a height field."""
    import numpy as np
    rng = np.random.default_rng(seed)
    sd = SceneDesc()
    mat = scenes.matte(sd, (0.7, 0.6, 0.5))
    items = []
    if spheres:
        c = rng.uniform(-6, 6, (n, 3)); c[:, 1] = np.abs(c[:, 1]) * 0.5 + 0.3
        r = rng.uniform(0.05, 0.5, n)
        items = [sd.sphere(tuple(c[i]), float(r[i])) for i in range(n)]
    else:
        c = rng.uniform(-6, 6, (n, 3)); c[:, 1] = np.abs(c[:, 1]) * 0.5
        e = rng.normal(0, 0.35, (n, 2, 3))
        pts = np.concatenate([c, c + e[:, 0], c + e[:, 1]], axis=1)
        clump = np.tile(pts[:1], (12, 1)) + rng.normal(0, 1e-3, (12, 9))  # twelve nearly coincident triangles
        items = sd.triangles_bulk(np.concatenate([pts, clump]))
    body = sd.tex(sd.bih(items), mat)
    if not floor:  # the all-triangle (or all-sphere) kernel instance
        return _finish(sd, body)
    return _finish(sd, sd.group([sd.tex(sd.box((-8, -0.6, -8), (8, -0.1, 8)), scenes.matte(sd, (0.3, 0.5, 0.3))), body]))


def deep_and_clump(n=2500, seed=9, k=9):
    """Two triangle BIHs side by side in the root list: a deep one (so the flat tier's LDS stack is at its full 12 entries and the
    frame runs on the two-row packet instance) and one whose ROOT is a leaf of `k` > 6 coincident triangles (build_rec keeps
    objects it cannot separate in one leaf, Bih.hs:211-285) -- the one-leaf tree of bih_tri_wave."""
    rng = np.random.default_rng(seed)
    sd = SceneDesc()
    c = rng.uniform(-6, 6, (n, 3)); c[:, 1] = np.abs(c[:, 1]) * 0.5
    e = rng.normal(0, 0.35, (n, 2, 3))
    deep = sd.bih(sd.triangles_bulk(np.concatenate([c, c + e[:, 0], c + e[:, 1]], axis=1)))
    tri = np.array([[-2.0, 0.2, 4.0, 2.5, 0.4, 4.5, 0.0, 3.5, 3.0]])
    clump = sd.bih(sd.triangles_bulk(np.tile(tri, (k, 1))))
    root = sd.group([sd.tex(deep, scenes.matte(sd, (0.7, 0.6, 0.5))), sd.tex(clump, scenes.matte(sd, (0.2, 0.4, 0.9)))])
    return _finish(sd, root, nlights=1)


def mirror_terrain(N=24):
    """Secondary rays through the packet walk (north star: reflection rays compacted back into the wavefront traversal): a
    heightfield of mirrors (Reflect 0.8, TestScene.hs:243) under `bih`, a glass-like Refract patch and matte spheres for the
    mirrors to show; maxdepth 3 bounces between the terrain's own slopes."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    tri = scenes.heightfield_triangles(N) * np.array([0.6, 1.0, 0.6] * 3)
    half = len(tri) // 2
    terrain = sd.tex(sd.bih(sd.triangles_bulk(tri[:half])), m["mirror"])
    glassy = sd.tex(sd.bih(sd.triangles_bulk(tri[half:])), sd.material_refract(0.3, 0.7, 1.3))
    balls = sd.bih([sd.tex(sd.sphere((float(x), 2.2, float(z)), 0.5), scenes.matte(sd, (0.9, 0.3 + 0.1 * x, 0.2))) for x in (-3, 0, 3) for z in (-2, 2)])
    return _finish(sd, sd.group([terrain, glassy, balls]))


ALL["mirror_terrain"] = mirror_terrain


def portal():
    """The portal of TestScene.hs:152-181: a door frame (a Difference of boxes) and, filling it, a box textured with
    `Warp frame scene lights xfm` (Shader.hs:47-50, 157-175) -- a hit traces the frame through the hit's own local ray
    (the portal stands inside a `transform`), then the scene itself through the warped ray (rotated to look down from
    above), and shows the nearer.  The scene looked into is the scene the portal stands in, so views nest until `recurs`
    runs out."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    w, h, th, dl = 2.0, 5.0, 1.0 / 3.0, 1e-4
    frame = sd.tag(sd.tex(sd.difference(sd.box((-w, 0, -th), (w, h, th)), sd.box((th - w, th, -(th + dl)), (w - th, h - th, th + dl))), scenes.matte(sd, (0.4, 0.4, 0.8))), "door frame")
    surface = sd.box((-w, 0, -dl), (w, h - dl, dl))
    xfm = api.compose([api.rotate((1, 0, 0), api.deg(-85)), api.translate((8, 40, -4))])  # TestScene.hs:167-169
    warp = sd.material_warp(frame, None, scenes.LIGHTS, xfm)
    door = sd.transform(sd.group([frame, sd.tex(surface, warp)]), [api.rotate((0, 1, 0), api.deg(8)), api.translate((-3, 0.5, -5))])  # TestScene.hs:193-194
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0, 0.8, 0.3)))
    items = [door,
             sd.tex(sd.sphere((2.5, 1.0, 0.5), 1.0), m["shiny_red"]),
             sd.tex(sd.sphere((-1.0, 0.7, 2.5), 0.7), m["mirror"]),
             sd.tex(sd.box((4, 0, -4), (5.5, 2.5, -2.5)), m["shiny_white"])]
    return _finish(sd, sd.group([pl, sd.bih(items)]))


ALL["portal"] = portal


def retexture():
    """`difference_retexture` = `Difference a b False` (Csg.hs:29-30): the surface hollowed out by b is rendered with b's
    textures -- the stack b's own rayint returned, `bt` (Csg.hs:43) -- where `difference` takes a's at the point (Csg.hs:39-42).
    Every tier that answers a Difference: the flat tier's CSG class (two primitives, bare and inside an Instance), the generic
    tier in place (two primitives below a composite) and over frames (a group carved by a bih of spheres), each beside the same
    solid built with `difference`; and one under an outer Tex, whose id b's stack ends with."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    gold, teal, pink = sd.material_surface((0.9, 0.7, 0.2), 1, 0.2, 0.8, 0.4, 10), scenes.matte(sd, (0.1, 0.7, 0.7)), scenes.matte(sd, (1, 0.4, 0.7))

    def carved(mk, at, outer=None):
        x, z = at
        a = sd.tex(sd.box((x - 1, 0.03, z - 1), (x + 1, 2, z + 1)), gold)  # (clear of the floor: coplanar faces tie differently in fp32)
        b = sd.tex(sd.sphere((x + 0.6, 2.0, z + 0.9), 1.0), m["shiny_red"])
        d = mk(a, b)
        return sd.tex(d, outer) if outer is not None else d
    flat = [carved(sd.difference_retexture, (-4.5, 2)), carved(sd.difference, (-1.5, 2)), carved(sd.difference_retexture, (1.5, 2), outer=teal)]
    inst = sd.transform(carved(sd.difference_retexture, (0, 0)), [api.rotate((0, 1, 0), api.deg(35)), api.scale((1.0, 1.3, 0.8)), api.translate((4.8, 0, 1.5))])
    # generic tier: the first operand a composite, the second a bih of textured spheres
    def blocks(): return sd.tex(sd.group([sd.box((float(i) - 3.0, 0.03, -4.0), (float(i) - 2.1, 1.8, -3.0)) for i in range(6)]), pink)
    def bites(): return sd.bih([sd.tex(sd.sphere((float(i) - 2.55, 1.8, -3.2), 0.55), m["shiny_white"] if i % 2 else teal) for i in range(6)])
    gen = sd.difference_retexture(blocks(), bites())
    gen_plain = sd.transform(sd.difference(blocks(), bites()), [api.translate((0, 0, -2.5))])
    inplace = sd.group([sd.difference_retexture(sd.tex(sd.sphere((-6.0, 1.25, -1.0), 1.2), gold), sd.tex(sd.box((-6.0, 1.0, -1.0), (-4.0, 3.0, 1.0)), m["shiny_red"])), sd.sphere((-6.0, 3.2, -1.0), 0.4)])
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0, 0.8, 0.3)))
    return _finish(sd, sd.group([pl, sd.bih(flat + [inst, gen, gen_plain, inplace])]))


ALL["retexture"] = retexture


def grove(n=150, seed=11):
    """A bih whose items are all answered in place by the interpreter (flatten.hpp item_in_place): plain primitives of every kind,
    Instances of one primitive -- cones and cylinders under non-uniform scales, the quadrics whose answer depends on the tmax they are
    tested with -- and Instances of a group of primitives, with Tex wrappers on either side of the Instance and NoShadow /
    OnlyShadow flags, in a scene the generic tier renders (the same tree once more inside an Instance, and once carved by a sphere).
    The interpreter's packet service walks such a tree with the wave as ONE packet (rt_generic.hpp bih_items_wave) -- what it does
    with the oak of GlomeView's default scene, a bih of 2,047 Instances of cones and spheres."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    mats = [m["shiny_red"], m["shiny_white"], scenes.matte(sd, (0.2, 0.5, 0.9)), scenes.matte(sd, (0.9, 0.8, 0.2)), scenes.matte(sd, (0.4, 0.8, 0.4))]

    def items(count, sd_seed, spread):
        rng = np.random.default_rng(sd_seed)
        out = []
        for k in range(count):
            ang, rad = rng.uniform(0, 2 * np.pi), spread * np.sqrt(rng.uniform(0.02, 1.0))
            pos = (float(rad * np.cos(ang)), float(rng.uniform(1.6, 4.2)), float(rad * np.sin(ang)))  # (clear of the floor)
            sc = tuple(float(x) for x in rng.uniform(0.25, 0.9, 3))
            ax = rng.normal(size=3); ax = tuple(float(x) for x in ax / np.linalg.norm(ax))
            xf = [api.scale(sc), api.rotate(ax, float(rng.uniform(0, np.pi))), api.translate(pos)]
            mat = mats[k % len(mats)]
            kind = k % 11
            if kind == 0: it = sd.transform(sd.cone((0, 0, 0), 0.6, (0, 1.5, 0), 0.15), xf)
            elif kind == 1: it = sd.transform(sd.tex(sd.cylinder((0, 0, 0), (0, 1.2, 0), 0.4), mat), xf)                      # Tex below the Instance
            elif kind == 2: it = sd.tex(sd.transform(sd.sphere((0, 0, 0), 0.8), xf), mat)                                      # Tex above it: an ellipsoid
            elif kind == 3: it = sd.tex(sd.sphere(pos, float(rng.uniform(0.15, 0.4))), mat)                                    # plain primitives
            elif kind == 4: it = sd.tex(sd.box(pos, (pos[0] + sc[0], pos[1] + sc[1], pos[2] + sc[2])), mat)
            elif kind == 5: it = sd.tex(sd.cone(pos, 0.35, (pos[0] + 0.3, pos[1] + 0.9, pos[2] - 0.2), 0.1), mat)
            elif kind == 6: it = sd.tex(sd.cylinder(pos, (pos[0] - 0.4, pos[1] + 0.7, pos[2] + 0.3), 0.2), mat)
            elif kind == 7: it = sd.tex(sd.transform(sd.group([sd.box((-0.4, 0, -0.4), (0.4, 0.5, 0.4)), sd.tex(sd.sphere((0, 0.8, 0), 0.35), mats[(k + 1) % len(mats)]),
                                                               sd.cylinder((0, 0.5, 0), (0, 1.4, 0), 0.12)]), xf), mat)                     # an Instance of a group of primitives
            elif kind == 8: it = sd.noshadow(sd.tex(sd.transform(sd.cone((0, 0, 0), 0.5, (0, 1.0, 0), 0.0), xf), mat))
            elif kind == 9: it = sd.onlyshadow(sd.transform(sd.box((-0.5, 0, -0.5), (0.5, 0.3, 0.5)), xf))
            else: it = sd.tex(sd.disc(pos, ax, 0.4), mat)
            out.append(it)
        return out
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0, 0.8, 0.3)))
    here = sd.bih(items(n, seed, 4.5))
    moved = sd.transform(sd.bih(items(n // 2, seed + 1, 3.0)), [api.rotate((0, 1, 0), api.deg(30)), api.scale((0.8, 1.4, 0.7)), api.translate((6.0, 0.0, -1.0))])
    carved = sd.tex(sd.difference(sd.bih(items(n // 2, seed + 2, 2.5)), sd.sphere((0, 1.2, 0), 1.6)), m["shiny_red"])
    carved = sd.transform(carved, [api.translate((-6.0, 0.0, 1.0))])
    _finish(sd, sd.group([pl, here, moved, carved]))
    sd.set_camera((0.5, 4.0, 9.5), (0.0, 1.0, 0.0), (0, 1, 0), 60)
    # A shadow ray leaves a cone or a cylinder 1e-4 off its surface; the near root of the quadratic, which should come out just
    # negative, is the difference of two products a few units large and in fp32 now and then comes out positive: the surface shadows
    # itself on a pixel where the fp64 checker's does not (every pixel beyond tolerance here lies on a cone or a cylinder, with the same
    # hit and the same depth: 4e-4 of the pixels at 160x90 are off by more than 1e-3).  And the normal of a quadric squashed four to one by
    # its Instance comes back through the inverse transpose: 2.8e-3 of the pixels are between 1e-4 and 1e-3 off (mean error 2.8e-5).  The
    # oak of GlomeView's default scene, cones under ten transforms, states the like (glome_amd/scenes.py).
    sd.pixel_outlier_max, sd.rel_outlier_max, sd.pixel_mean_max = 6e-3, 2.5e-2, 2e-4
    return sd


ALL["grove"] = grove


testscene = scenes.testscene  # GlomeView's default scene (TestScene.hs:183-197), in glome_amd/scenes.py so bench.py can time it


ALL["testscene"] = lambda: testscene(4)  # (a 9x9x9 lattice keeps the CPU oracle's frames in seconds; the GPU tests also run the 21x21x21 one)


def deep_nest(levels=14):
    """Composites nested `levels` deep -- group [instance (bih [...]), sphere] over and over, each level moved, turned and
    scaled a little, and a Difference whose operands are shallow at the very top: the reference recurses as deep as the scene
    goes; the device's rayint / shadow loop keeps its frames in memory and does the same (rt_generic.hpp)."""
    sd = SceneDesc()
    m = scenes.materials(sd)
    teal, pink = scenes.matte(sd, (0.1, 0.7, 0.7)), scenes.matte(sd, (1, 0.4, 0.7))
    n = sd.tex(sd.sphere((0, 0, 0), 0.5), m["shiny_red"])
    for k in range(levels // 2):
        side = sd.tex(sd.sphere((1.1, 0.0, 0.0), 0.35) if k % 2 else sd.box((0.7, -0.3, -0.3), (1.3, 0.3, 0.3)), teal if k % 2 else pink)
        inner = sd.bih([n, side]) if k % 3 == 1 else sd.group([n, side])
        n = sd.group([sd.transform(inner, [api.rotate((0, 0, 1), api.deg(17 + k)), api.scale((0.93, 0.9, 0.95)), api.translate((0.35, 0.25, 0.1))]),
                      sd.tex(sd.sphere((-0.9, 0.2 * k, 0.3), 0.25), m["shiny_white"])])
    carved = sd.group([sd.transform(n, [api.scale((1.6, 1.6, 1.6)), api.translate((0, 1.6, 0))]),
                       sd.tex(sd.difference(sd.sphere((-3.5, 1, 0), 1.0), sd.box((-4.6, 1.0, -1.2), (-2.4, 2.2, 1.2))), teal)])
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0, 0.8, 0.3)))
    return _finish(sd, sd.group([pl, carved]))


ALL["deep_nest"] = deep_nest
ALL["deeper_nest"] = lambda: deep_nest(levels=50)  # (needs ~1,000 of the interpreter's 2,048 frame words per ray: beyond the 768 of rounds 2-3)


def random_composites(seed, n_items=9, max_depth=3):
    """A random scene in TestScene.hs vocabulary: primitives of every family under random Tex stacks, grouped, instanced with
    random (rotate, non-uniform scale, translate) transforms, carved (Difference) and intersected, nested up to `max_depth`
    composite levels, some of it under a `bih`.  Seeded: the same seed gives the same scene on every backend."""
    rng = np.random.default_rng(seed)
    sd = SceneDesc()
    m = scenes.materials(sd)
    mats = [m["shiny_white"], m["shiny_red"], m["mirror"], scenes.matte(sd, (0.2, 0.3, 0.9)), sd.material_refract(0.3, 0.7, 1.4),
            sd.material_blend_fn(m["shiny_white"], m["shiny_red"], api.WEIGHT_STRIPE_TRIANGLE, [2, 3, 1])]
    U = lambda a, b: float(rng.uniform(a, b))

    def prim(c):
        k = int(rng.integers(0, 6))
        if k == 0: return sd.sphere(c, U(0.4, 1.1))
        if k == 1: return sd.box((c[0] - U(0.3, 0.9), c[1] - U(0.3, 0.9), c[2] - U(0.3, 0.9)), (c[0] + U(0.3, 0.9), c[1] + U(0.3, 0.9), c[2] + U(0.3, 0.9)))
        if k == 2: return sd.cone((c[0], c[1] - 0.6, c[2]), U(0.4, 0.9), (c[0] + U(-0.4, 0.4), c[1] + U(0.5, 1.2), c[2] + U(-0.4, 0.4)), U(0.0, 0.35))
        if k == 3: return sd.cylinder((c[0] - 0.5, c[1], c[2]), (c[0] + U(0.3, 0.9), c[1] + U(-0.3, 0.6), c[2] + U(-0.3, 0.3)), U(0.25, 0.6))
        if k == 4: return sd.triangle((c[0] - 0.8, c[1] - 0.4, c[2]), (c[0] + 0.8, c[1] - 0.3, c[2] + U(-0.4, 0.4)), (c[0] + U(-0.3, 0.3), c[1] + 0.9, c[2] + U(-0.3, 0.3)))
        return sd.disc(c, (0.0, 0.6, 0.8), U(0.4, 0.9))

    def solid_prim(c):  # something with an inside (CSG operands)
        k = int(rng.integers(0, 3))
        if k == 0: return sd.sphere(c, U(0.5, 1.1))
        if k == 1: return sd.box((c[0] - U(0.4, 0.9), c[1] - U(0.4, 0.9), c[2] - U(0.4, 0.9)), (c[0] + U(0.4, 0.9), c[1] + U(0.4, 0.9), c[2] + U(0.4, 0.9)))
        return sd.cone((c[0], c[1] - 0.7, c[2]), U(0.5, 0.9), (c[0], c[1] + U(0.6, 1.1), c[2]), U(0.0, 0.3))

    def xf():
        ax = np.array([U(-1, 1), U(-1, 1), U(0.2, 1)]); ax = ax / np.sqrt(ax @ ax)
        return [api.rotate(tuple(float(x) for x in ax), U(-1.2, 1.2)), api.scale((U(0.6, 1.5), U(0.6, 1.5), U(0.6, 1.5))), api.translate((U(-0.8, 0.8), U(-0.2, 0.6), U(-0.8, 0.8)))]

    def tex(n, budget):  # -> (node, textures still allowed above it): at most 4 on any path (the device's texture stack)
        for _ in range(int(rng.integers(0, 3))):
            if budget[0] > 0: n = sd.tex(n, mats[int(rng.integers(0, len(mats)))]); budget[0] -= 1
        return n

    def small_mesh(c):  # a fan of triangles, flat or with vertex normals, some with their own texture (Mesh.hs:27-29)
        nv = int(rng.integers(4, 8))
        verts = [(c[0], c[1] + 0.8, c[2])] + [(c[0] + 1.1 * np.cos(2 * np.pi * q / nv), c[1] + U(-0.2, 0.2), c[2] + 1.1 * np.sin(2 * np.pi * q / nv)) for q in range(nv)]
        smooth = rng.uniform() < 0.5
        norms = []
        if smooth:
            for v in verts:
                nn = np.array([v[0] - c[0], v[1] - c[1] + 0.6, v[2] - c[2]]); norms.append(tuple(float(x) for x in nn / np.sqrt(nn @ nn)))
        mm = [mats[int(rng.integers(0, len(mats)))] for _ in range(2)]
        tris = []
        for q in range(nv):
            a_, b_, c_ = 0, 1 + q, 1 + (q + 1) % nv
            tris.append((a_, b_, c_) + ((a_, b_, c_) if smooth else (-1, -1, -1)) + (int(rng.integers(-1, 2)), -1))
        return sd.mesh(verts, norms, tris, mm)

    def node(c, depth, budget):
        k = int(rng.integers(0, 10)) if depth > 0 else 0
        if k <= 1: return tex(prim(c), budget)
        if k == 7:  # Bound / InnerBound (Bound.hs): a bounding solid around, or inside, something
            inner = node(c, depth - 1, [budget[0]])
            if rng.uniform() < 0.6: return sd.bound_object(sd.sphere(c, U(1.6, 2.6)), inner)
            return sd.innerbound(tex(sd.sphere(c, U(0.2, 0.5)), [0]), inner)
        if k == 8:
            w = node(c, depth - 1, [budget[0]])
            return sd.noshadow(w) if rng.uniform() < 0.5 else sd.onlyshadow(w)
        if k == 9: return tex(small_mesh(c), [max(0, budget[0] - 1)])
        near = lambda s: (c[0] + U(-s, s), c[1] + U(-s, s), c[2] + U(-s, s))
        sub = lambda: [budget[0]]  # every child path starts from what is left here
        def wrap(make):  # textures outside a composite come out of the budget its children may use
            outer = [budget[0]]
            n_out = int(rng.integers(0, 3))
            inner = [max(0, outer[0] - n_out)]
            n = make(inner)
            for _ in range(min(n_out, outer[0])): n = sd.tex(n, mats[int(rng.integers(0, len(mats)))])
            return n
        if k == 2: return wrap(lambda bud: sd.group([node(near(0.9), depth - 1, [bud[0]]) for _ in range(int(rng.integers(2, 4)))]))
        if k == 3: return sd.transform(node(c, depth - 1, [budget[0]]), xf())
        if k == 4:
            def mk(bud):
                a = node(c, depth - 1, [bud[0]]) if rng.uniform() < 0.4 else tex(solid_prim(c), [bud[0]])
                return sd.difference(a, tex(solid_prim(near(0.6)), [bud[0]]))
            return wrap(mk)
        if k == 5: return wrap(lambda bud: sd.intersection([tex(solid_prim(near(0.35)), [bud[0]]) for _ in range(int(rng.integers(2, 4)))]))
        return sd.bih([node(near(1.2), depth - 1, [budget[0]]) for _ in range(int(rng.integers(2, 5)))])

    items = [node((U(-5, 5), U(0.6, 2.6), U(-4, 4)), max_depth, [4]) for _ in range(n_items)]
    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0, 0.8, 0.3)))
    return _finish(sd, sd.group([pl, sd.bih(items)]))


def random_flat(seed, n_items=14):
    """A random scene the flat tier renders: primitives of every family under Tex stacks and shadow flags, cylinders / cones,
    Differences / Intersections over primitives and Instances of those (the flat CSG class), under one `bih`, plus items
    outside it and, sometimes, a second `bih` of triangles only (the packet walk) or of spheres only."""
    rng = np.random.default_rng(1000 + seed)
    sd = SceneDesc()
    m = scenes.materials(sd)
    mats = [m["shiny_white"], m["shiny_red"], m["mirror"], scenes.matte(sd, (0.2, 0.3, 0.9)), sd.material_refract(0.3, 0.7, 1.4)]
    U = lambda a, b: float(rng.uniform(a, b))
    C = lambda: (U(-5, 5), U(0.5, 2.8), U(-4, 4))
    near = lambda c, s: (c[0] + U(-s, s), c[1] + U(-s, s), c[2] + U(-s, s))

    def solid(c):  # CSG operands of the flat tier are primitives proper (a cone is an Instance: the composite fuzz covers those)
        if rng.uniform() < 0.5: return sd.sphere(c, U(0.5, 1.1))
        return sd.box((c[0] - U(0.4, 0.9), c[1] - U(0.4, 0.9), c[2] - U(0.4, 0.9)), (c[0] + U(0.4, 0.9), c[1] + U(0.4, 0.9), c[2] + U(0.4, 0.9)))

    def tex(n, k=None):
        for _ in range(int(rng.integers(0, 3)) if k is None else k): n = sd.tex(n, mats[int(rng.integers(0, len(mats)))])
        return n

    def item():
        c = C()
        k = int(rng.integers(0, 10))
        if k == 0: return tex(sd.sphere(c, U(0.4, 1.0)))
        if k == 1: return tex(sd.box((c[0] - 0.6, c[1] - 0.5, c[2] - 0.7), (c[0] + U(0.3, 0.9), c[1] + U(0.3, 0.9), c[2] + U(0.3, 0.9))))
        if k == 2: return tex(sd.triangle((c[0] - 0.9, c[1] - 0.4, c[2]), (c[0] + 0.9, c[1] - 0.3, c[2] + U(-0.4, 0.4)), (c[0], c[1] + 1.0, c[2] + U(-0.3, 0.3))))
        if k == 3: return tex(sd.cylinder((c[0] - 0.5, c[1], c[2]), (c[0] + U(0.3, 0.9), c[1] + U(-0.3, 0.6), c[2]), U(0.25, 0.6)))
        if k == 4: return tex(sd.cone((c[0], c[1] - 0.6, c[2]), U(0.4, 0.9), (c[0] + U(-0.3, 0.3), c[1] + U(0.5, 1.2), c[2]), U(0.0, 0.35)))
        if k == 5: return tex(sd.disc(c, (0.0, 0.6, 0.8), U(0.4, 0.9)))
        if k == 6: return tex(sd.difference(tex(solid(c), 1), tex(solid(near(c, 0.6)), int(rng.integers(0, 2)))), int(rng.integers(0, 2)))
        if k == 7: return tex(sd.intersection([tex(solid(near(c, 0.35)), int(rng.integers(0, 2))) for _ in range(int(rng.integers(2, 4)))]), 1)
        if k == 8:
            ax = np.array([U(-1, 1), U(-1, 1), U(0.2, 1)]); ax = ax / np.sqrt(ax @ ax)
            inner = tex(solid(c), 1) if rng.uniform() < 0.5 else sd.difference(tex(solid(c), 1), solid(near(c, 0.5)))
            return tex(sd.transform(inner, [api.rotate(tuple(float(x) for x in ax), U(-1, 1)), api.scale((U(0.7, 1.4), U(0.7, 1.4), U(0.7, 1.4)))]), int(rng.integers(0, 2)))
        w = tex(sd.sphere(c, U(0.4, 0.9)), 1)
        return sd.noshadow(w) if rng.uniform() < 0.5 else sd.onlyshadow(w)

    pl = sd.tex(sd.plane((0, 0, 0), (0, 1, 0)), scenes.matte(sd, (0, 0.8, 0.3)))
    parts = [pl, sd.bih([item() for _ in range(n_items)]), item(), item()]
    extra = int(rng.integers(0, 3))
    if extra == 1:
        parts.append(sd.tex(sd.bih([sd.triangle((x, 0.2 + 0.3 * ((x * 7 + z * 3) % 5), z), (x + 1, 0.3, z), (x, 0.4, z + 1)) for x in range(-4, 4) for z in range(-9, -5)]), mats[3]))
    if extra == 2:
        parts.append(sd.bih([sd.tex(sd.sphere((float(x), 0.4, float(z)), 0.35), mats[(x + z) % 2]) for x in range(-4, 4) for z in range(5, 8)]))
    return _finish(sd, sd.group(parts))


def random_rig(sd, seed):
    """a random view and light rig for a fuzz scene (tools/probe/fuzz_gpu.py's): axis-aligned and inside-the-scene cameras, 1-4
    lights, some without shadows, some of finite reach"""
    rng = np.random.default_rng(7000 + seed)
    k = int(rng.integers(0, 4))
    if k == 0: sd.set_camera((0.0, 2.0, 12.0), (0.0, 2.0, 0.0), (0, 1, 0), 45.0)
    elif k == 1: sd.set_camera((float(rng.uniform(-3, 3)), float(rng.uniform(0.5, 3)), float(rng.uniform(-3, 3))), (0.0, 1.0, 0.0), (0, 1, 0), 70.0)
    elif k == 2: sd.set_camera((float(rng.uniform(-9, 9)), float(rng.uniform(3, 9)), float(rng.uniform(8, 14))), (0.0, 1.0, 0.0), (0, 1, 0), float(rng.uniform(30, 60)))
    sd.lights = []
    for _ in range(int(rng.integers(1, 5))):
        sd.add_light((float(rng.uniform(-30, 30)), float(rng.uniform(5, 60)), float(rng.uniform(-10, 60))), tuple(float(x) for x in rng.uniform(20, 900, 3)),
                     rad=float(rng.uniform(15, 60)) if rng.uniform() < 0.3 else 1000000.0, shadow=bool(rng.uniform() < 0.8))
    return sd
