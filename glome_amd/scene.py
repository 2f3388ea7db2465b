"""SceneDesc: a scene written once in glome's constructor vocabulary (TestScene.hs style), replayable into any
backend exposing the same constructors -- the product's `api.Builder`, or a checker.  Constants are rounded to
fp32 on entry so every backend receives bit-identical inputs.
"""
import numpy as np


def r32(x):
    """Round scene constants to fp32, return python floats / nested lists."""
    a = np.asarray(x, dtype=np.float64)
    return a.astype(np.float32).astype(np.float64).tolist()


NODE_OPS = {"sphere", "triangle", "trianglenorm", "box", "plane", "plane_offset", "disc", "cylinder", "cone", "group",
            "transform", "difference", "difference_retexture", "intersection", "bih", "mesh", "tex", "tag", "noshadow", "onlyshadow", "bound_object",
            "innerbound", "flatten_transform", "tolist", "triangles_bulk"}
MAT_OPS = {"material_surface", "material_reflect", "material_refract", "material_layers", "material_blend", "material_blend_fn", "material_warp"}


class SceneDesc:
    def __init__(self, round32=True):
        self.ops = []  # (kind, name, args) ; kind 'n' node, 'm' material
        self.n_nodes = 0
        self.n_mats = 0
        self.round32 = round32
        self.root = None
        self.lights = []  # (pos, color, rad, shadow)
        self.cam = None   # (pos, at, up, angle)

    def _q(self, x):
        return r32(x) if self.round32 else np.asarray(x, dtype=np.float64).tolist()

    def _node(self, name, *args):
        self.ops.append(("n", name, args))
        self.n_nodes += 1
        return self.n_nodes - 1

    def _mat(self, name, *args):
        self.ops.append(("m", name, args))
        self.n_mats += 1
        return self.n_mats - 1

    # --- constructors (ids are SceneDesc-local; replay() maps them) ---
    def sphere(self, c, r): return self._node("sphere", self._q(c), self._q(r))
    def triangle(self, p1, p2, p3): return self._node("triangle", self._q(p1), self._q(p2), self._q(p3))
    def trianglenorm(self, p1, p2, p3, n1, n2, n3): return self._node("trianglenorm", *[self._q(v) for v in (p1, p2, p3, n1, n2, n3)])
    def box(self, a, b): return self._node("box", self._q(a), self._q(b))
    def plane(self, pt, n): return self._node("plane", self._q(pt), self._q(n))
    def plane_offset(self, n, off): return self._node("plane_offset", self._q(n), self._q(off))
    def disc(self, pos, n, r): return self._node("disc", self._q(pos), self._q(n), self._q(r))
    def cylinder(self, p1, p2, r): return self._node("cylinder", self._q(p1), self._q(p2), self._q(r))
    def cone(self, p1, r1, p2, r2): return self._node("cone", self._q(p1), self._q(r1), self._q(p2), self._q(r2))
    def group(self, ids): return self._node("group", list(ids))
    def transform(self, node, xfms): return self._node("transform", node, [np.asarray(x, dtype=np.float64) for x in xfms])
    def difference(self, a, b): return self._node("difference", a, b)
    def difference_retexture(self, a, b): return self._node("difference_retexture", a, b)  # Csg.hs:29-30
    def intersection(self, ids): return self._node("intersection", list(ids))
    def bih(self, ids): return self._node("bih", list(ids))
    def mesh(self, verts, norms, tris, mats):
        return self._node("mesh", np.asarray(self._q(verts)).reshape(-1, 3), np.asarray(self._q(norms)).reshape(-1, 3) if len(norms) else np.zeros((0, 3)),
                          np.asarray(tris, dtype=np.int32).reshape(-1, 8), list(mats))
    def tex(self, node, mat): return self._node("tex", node, mat)
    def tag(self, node, name=None): return self._node("tag", node)
    def noshadow(self, node): return self._node("noshadow", node)
    def onlyshadow(self, node): return self._node("onlyshadow", node)
    def bound_object(self, a, b): return self._node("bound_object", a, b)
    def innerbound(self, a, b): return self._node("innerbound", a, b)
    def flatten_transform(self, node): return self._node("flatten_transform", node)
    def tolist(self, node): return self._node("tolist", node)
    def bih_tolist(self, node): return self._node("bih_tolist", node)  # `bih (tolist node)`: a bih of the list's items, TestScene.hs:109

    def triangles_bulk(self, pts9):
        """Many triangles at once (n x 9).  Returns the list of node ids."""
        pts9 = np.asarray(self._q(pts9), dtype=np.float64).reshape(-1, 9)
        self.ops.append(("N", "triangles_bulk", (pts9,)))
        first = self.n_nodes
        self.n_nodes += pts9.shape[0]
        return list(range(first, self.n_nodes))

    def material_surface(self, color, alpha, amb, kd, ks, shine):
        return self._mat("material_surface", self._q(color), self._q(alpha), self._q(amb), self._q(kd), self._q(ks), self._q(shine))
    def material_reflect(self, refl): return self._mat("material_reflect", self._q(refl))
    def material_refract(self, refl, refr, ior): return self._mat("material_refract", self._q(refl), self._q(refr), self._q(ior))
    def material_layers(self, mats): return self._mat("material_layers", list(mats))
    def material_blend(self, a, b, w): return self._mat("material_blend", a, b, self._q(w))
    def material_blend_fn(self, a, b, fn, params): return self._mat("material_blend_fn", a, b, int(fn), self._q(list(params)))

    def material_warp(self, frame, scene, lights, xfm):
        """Warp frame scene' lights' xfm (Shader.hs:47-50).  scene: a node or None = the scene's own root; lights: [(pos, color)]
        or [(pos, color, rad, shadow)]; xfm: the 24 doubles of the closure's transform (api.compose of api.rotate / translate ...)."""
        ls = [(self._q(l[0]), self._q(l[1]), float(l[2]) if len(l) > 2 else 1000000.0, bool(l[3]) if len(l) > 3 else True) for l in lights]
        return self._mat("material_warp", frame, scene, ls, np.asarray(xfm, dtype=np.float64).ravel().copy())

    def set_root(self, node): self.root = node
    def add_light(self, pos, color, rad=1000000.0, shadow=True): self.lights.append((self._q(pos), self._q(color), float(rad), bool(shadow)))
    def set_camera(self, pos, at, up, angle): self.cam = (self._q(pos), self._q(at), self._q(up), float(angle))

    @staticmethod
    def _warp_lights(backend, ls):
        """the product builder takes glome_light structs (fp32 fields, like the render call's lights); a checker takes the tuples"""
        if hasattr(backend, "lib") and hasattr(backend.lib, "glome_sb_material_warp"):
            from . import api
            return [api.light(p, c, r, s) for (p, c, r, s) in ls]
        return ls

    # --- replay ---
    def replay(self, backend):
        """Replay into `backend`; returns (node id map, material id map) from SceneDesc ids to backend ids."""
        nmap, mmap = [], []
        N = lambda i: nmap[i]
        for kind, name, args in self.ops:
            if kind == "m":
                if name == "material_layers":
                    mmap.append(backend.material_layers([mmap[m] for m in args[0]]))
                elif name == "material_blend":
                    mmap.append(backend.material_blend(mmap[args[0]], mmap[args[1]], args[2]))
                elif name == "material_blend_fn":
                    mmap.append(backend.material_blend_fn(mmap[args[0]], mmap[args[1]], args[2], args[3]))
                elif name == "material_warp":
                    mmap.append(backend.material_warp(N(args[0]), None if args[1] is None else N(args[1]), self._warp_lights(backend, args[2]), args[3]))
                else:
                    mmap.append(getattr(backend, name)(*args))
            elif kind == "N":
                nmap.extend(backend.triangles_bulk(args[0]))
            elif name in ("group", "intersection", "bih"):
                nmap.append(getattr(backend, name)([N(i) for i in args[0]]))
            elif name == "transform":
                nmap.append(backend.transform(N(args[0]), args[1]))
            elif name in ("difference", "difference_retexture", "bound_object", "innerbound"):
                nmap.append(getattr(backend, name)(N(args[0]), N(args[1])))
            elif name == "tex":
                nmap.append(backend.tex(N(args[0]), mmap[args[1]]))
            elif name in ("tag", "noshadow", "onlyshadow", "flatten_transform", "tolist", "bih_tolist"):
                nmap.append(getattr(backend, name)(N(args[0])))
            elif name == "mesh":
                nmap.append(backend.mesh(args[0], args[1], args[2], [mmap[m] for m in args[3]]))
            else:
                nmap.append(getattr(backend, name)(*args))
        return nmap, mmap
