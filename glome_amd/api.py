"""Host-side mirror of glome's scene vocabulary over the C ABI (include/glome_hip.h).

Names follow the reference: `sphere`, `triangle`, `box`, `cone`, `cylinder`, `disc`, `plane`, `plane_offset`,
`difference`, `intersection`, `bih`, `mesh`, `group`, `transform`, `tex`, `tag`, `noshadow`, `onlyshadow`,
`bound_object`, `innerbound` (GlomeTrace/Data/Glome/Scene.hs:1-27 re-exports them), `camera` (Scene.hs:48-57),
`light` (Shader.hs:22-23), transforms `translate` / `scale` / `rotate` / `xyz_to_uvw` / `compose` (Vec.hs:461-629).
Errors surface as exceptions the way the reference's constructors `error` out.

All compute goes through the HIP library; nothing here traces rays.
"""
import ctypes as C

import numpy as np

from . import _lib as L


class GlomeError(RuntimeError):
    pass


# ---------------------------------------------------------------- transforms (24 doubles: forward 3x4 + inverse 3x4)
def _xf(fn, *args):
    lib = L.load()
    out = np.zeros(24, dtype=np.float64)
    keep = []
    cargs = []
    for a in args:
        if isinstance(a, (float, int)):
            cargs.append(C.c_double(float(a)))
        else:
            arr, p = L.dvec(a)
            keep.append(arr)
            cargs.append(p)
    rc = getattr(lib, fn)(*cargs, out.ctypes.data_as(L.c_dp))
    if rc != 0:
        raise GlomeError(f"{fn}: invalid transform (reference would `error`, Vec.hs:466-477, 577-622)")
    return out


def translate(v):
    return _xf("glome_xfm_translate", v)


def scale(v):
    return _xf("glome_xfm_scale", v)


def rotate(axis, angle_rad):
    return _xf("glome_xfm_rotate", axis, float(angle_rad))


def xyz_to_uvw(u, v, w):
    return _xf("glome_xfm_xyz_to_uvw", u, v, w)


def compose(xfms):
    lib = L.load()
    arr = np.ascontiguousarray(np.asarray(xfms, dtype=np.float64).reshape(-1, 24))
    out = np.zeros(24, dtype=np.float64)
    if lib.glome_xfm_compose(arr.ctypes.data_as(L.c_dp), arr.shape[0], out.ctypes.data_as(L.c_dp)) != 0:
        raise GlomeError("compose: corrupt matrix")
    return out


def deg(x):  # Vec.hs:17-18 (note the truncated pi, as written in the reference)
    return (x * 3.1415926535897) / 180


def camera(pos, at, up, angle_deg):
    """camera pos at up angle (Scene.hs:48-57) -> L.Camera (fp32 fields)."""
    lib = L.load()
    cam = L.Camera()
    a, pa = L.dvec(pos)
    b, pb = L.dvec(at)
    c, pc = L.dvec(up)
    lib.glome_camera_lookat(pa, pb, pc, float(angle_deg), C.byref(cam))
    return cam


def camera_from_vectors(pos, fwd, up, right):
    cam = L.Camera()
    for name, v in (("pos", pos), ("fwd", fwd), ("up", up), ("right", right)):
        getattr(cam, name)[:] = [float(x) for x in v]
    return cam


WEIGHT_PERLIN, WEIGHT_STRIPE_SQUARE, WEIGHT_STRIPE_TRIANGLE, WEIGHT_STRIPE_SINE = 1, 2, 3, 4  # GLOME_WEIGHT_*


def light(pos, color, rad=1000000.0, shadow=True):
    """light pos clr = Light pos clr (\\x -> 1/(x*x)) infinity True (Shader.hs:22-23)."""
    li = L.Light()
    li.pos[:] = [float(x) for x in pos]
    li.color[:] = [float(x) for x in color]
    li.rad = float(rad)
    li.shadow = 1 if shadow else 0
    return li


def render_params(width=720, height=480, mode=0, blocksize=65, maxdepth=3, fog=0, tile_first=0, tile_stride=1,
                  faithful=0, count_work=0, thresholds=None, rank0_share_pct=0):
    lib = L.load()
    p = L.RenderParams()
    lib.glome_render_params_default(C.byref(p))
    p.width, p.height, p.mode, p.blocksize, p.maxdepth, p.fog = width, height, mode, blocksize, maxdepth, fog
    p.tile_first, p.tile_stride, p.faithful, p.count_work = tile_first, tile_stride, faithful, count_work
    p.rank0_share_pct = rank0_share_pct
    if thresholds is not None:
        p.thresholds[:] = [float(t) for t in thresholds]
    return p


# ---------------------------------------------------------------- scene builder
class Builder:
    """One method per reference constructor; returns integer node / material ids."""

    def __init__(self):
        self.lib = L.load()
        self.h = self.lib.glome_sb_new()

    def __del__(self):
        try:
            if self.h:
                self.lib.glome_sb_free(self.h)
                self.h = None
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            raise GlomeError(f"{what}: {self.lib.glome_sb_last_error(self.h).decode()} (status {rc})")
        return rc

    def _call(self, name, *args):
        keep, cargs = [], []
        for a in args:
            if isinstance(a, float):
                cargs.append(C.c_double(a))
            elif isinstance(a, (int, np.integer)):
                cargs.append(int(a))
            else:
                arr, p = L.dvec(a)
                keep.append(arr)
                cargs.append(p)
        return self._chk(getattr(self.lib, name)(self.h, *cargs), name)

    def sphere(self, c, r): return self._call("glome_sb_sphere", c, float(r))
    def triangle(self, p1, p2, p3): return self._call("glome_sb_triangle", list(p1) + list(p2) + list(p3))
    def trianglenorm(self, p1, p2, p3, n1, n2, n3): return self._call("glome_sb_trianglenorm", list(p1) + list(p2) + list(p3), list(n1) + list(n2) + list(n3))
    def box(self, a, b): return self._call("glome_sb_box", a, b)
    def plane(self, pt, n): return self._call("glome_sb_plane", pt, n)
    def plane_offset(self, n, off): return self._call("glome_sb_plane_offset", n, float(off))
    def disc(self, pos, n, r): return self._call("glome_sb_disc", pos, n, float(r))
    def cylinder(self, p1, p2, r): return self._call("glome_sb_cylinder", p1, p2, float(r))
    def cone(self, p1, r1, p2, r2): return self._call("glome_sb_cone", p1, float(r1), p2, float(r2))

    def _ids(self, name, ids):
        arr, p = L.ivec(ids)
        return self._chk(getattr(self.lib, name)(self.h, p, len(arr)), name)

    def group(self, ids): return self._ids("glome_sb_group", ids)
    def intersection(self, ids): return self._ids("glome_sb_intersection", ids)
    def bih(self, ids): return self._ids("glome_sb_bih", ids)

    def triangles_bulk(self, pts9):
        """n x 9 array -> list of triangle ids (one glome_sb_triangle call each)."""
        pts9 = np.ascontiguousarray(pts9, dtype=np.float64).reshape(-1, 9)
        f = self.lib.glome_sb_triangle
        base = pts9.ctypes.data
        ids = [f(self.h, C.cast(base + 72 * k, L.c_dp)) for k in range(pts9.shape[0])]
        if ids and min(ids) < 0:
            self._chk(min(ids), "glome_sb_triangle")
        return ids

    def transform(self, node, xfms):
        arr = np.ascontiguousarray(np.asarray(xfms, dtype=np.float64).reshape(-1, 24))
        return self._chk(self.lib.glome_sb_transform(self.h, int(node), arr.ctypes.data_as(L.c_dp), arr.shape[0]), "glome_sb_transform")

    def difference(self, a, b): return self._chk(self.lib.glome_sb_difference(self.h, int(a), int(b)), "glome_sb_difference")
    def difference_retexture(self, a, b): return self._chk(self.lib.glome_sb_difference_retexture(self.h, int(a), int(b)), "glome_sb_difference_retexture")

    def mesh(self, verts, norms, tris, mats):
        v = np.ascontiguousarray(np.asarray(verts, dtype=np.float64).reshape(-1, 3))
        n = np.ascontiguousarray(np.asarray(norms, dtype=np.float64).reshape(-1, 3))
        t = np.ascontiguousarray(np.asarray(tris, dtype=np.int32).reshape(-1, 8))
        m = np.ascontiguousarray(np.asarray(mats, dtype=np.int32).ravel())
        return self._chk(self.lib.glome_sb_mesh(self.h, v.ctypes.data_as(L.c_dp), v.shape[0], n.ctypes.data_as(L.c_dp), n.shape[0],
                                                t.ctypes.data_as(L.c_ip), t.shape[0], m.ctypes.data_as(L.c_ip), m.shape[0]), "glome_sb_mesh")

    def tex(self, node, material): return self._chk(self.lib.glome_sb_tex(self.h, int(node), int(material)), "glome_sb_tex")
    def tag(self, node, _tag=None): return self._chk(self.lib.glome_sb_tag(self.h, int(node)), "glome_sb_tag")
    def noshadow(self, node): return self._chk(self.lib.glome_sb_noshadow(self.h, int(node)), "glome_sb_noshadow")
    def onlyshadow(self, node): return self._chk(self.lib.glome_sb_onlyshadow(self.h, int(node)), "glome_sb_onlyshadow")
    def bound_object(self, a, b): return self._chk(self.lib.glome_sb_bound_object(self.h, int(a), int(b)), "glome_sb_bound_object")
    def innerbound(self, a, b): return self._chk(self.lib.glome_sb_innerbound(self.h, int(a), int(b)), "glome_sb_innerbound")
    def flatten_transform(self, node): return self._chk(self.lib.glome_sb_flatten_transform(self.h, int(node)), "glome_sb_flatten_transform")
    def tolist(self, node): return self._chk(self.lib.glome_sb_tolist(self.h, int(node)), "glome_sb_tolist")

    def list_items(self, node):
        """the [SolidItem] that `tolist node` yields, as node ids"""
        n = self._chk(self.lib.glome_sb_list_items(self.h, int(node), None, 0), "glome_sb_list_items")
        out = np.zeros(max(1, n), np.int32)
        self._chk(self.lib.glome_sb_list_items(self.h, int(node), out.ctypes.data_as(L.c_ip), n), "glome_sb_list_items")
        return [int(x) for x in out[:n]]

    def bih_tolist(self, node): return self.bih(self.list_items(node))  # `bih (tolist node)`, TestScene.hs:109

    def material_surface(self, color, alpha, amb, kd, ks, shine):
        return self._call("glome_sb_material_surface", color, float(alpha), float(amb), float(kd), float(ks), float(shine))
    def material_reflect(self, refl): return self._call("glome_sb_material_reflect", float(refl))
    def material_refract(self, refl, refr, ior): return self._call("glome_sb_material_refract", float(refl), float(refr), float(ior))
    def material_layers(self, mats): return self._ids("glome_sb_material_layers", mats)
    def material_blend(self, a, b, w): return self._chk(self.lib.glome_sb_material_blend(self.h, int(a), int(b), float(w)), "glome_sb_material_blend")

    def load_nff(self, text, max_lights=16):
        """A scene in NFF (Spd.hs:89-254): returns (root node, (from, at, up, angle), [(pos, rgb), ...], background rgb)."""
        cam = (C.c_double * 10)(); lp = (C.c_double * (6 * max_lights))(); nl = C.c_int32(0); bg = (C.c_double * 3)()
        root = self._chk(self.lib.glome_sb_load_nff(self.h, text.encode() if isinstance(text, str) else text, cam, lp, max_lights, C.byref(nl), bg), "glome_sb_load_nff")
        c = list(cam)
        lights = [(tuple(lp[6 * k:6 * k + 3]), tuple(lp[6 * k + 3:6 * k + 6])) for k in range(min(nl.value, max_lights))]
        return root, (tuple(c[0:3]), tuple(c[3:6]), tuple(c[6:9]), c[9]), lights, tuple(bg)

    def show(self, node):
        """Node `node` as the text GlomeView's `show geom` prints (Glome.hs:431; derived Show + Solid.hs:277, Tex.hs:50, Mesh.hs:44)."""
        n = self.lib.glome_sb_show(self.h, int(node), None, 0)
        self._chk(int(n), "glome_sb_show")
        buf = C.create_string_buffer(n + 1)
        self._chk(int(self.lib.glome_sb_show(self.h, int(node), buf, n + 1)), "glome_sb_show")
        return buf.value.decode()

    def show_tex_materials(self, node):
        """Material ids of the Tex constructors of show(node), in reading order (the part the text cannot carry)."""
        n = int(self.lib.glome_sb_show_tex_materials(self.h, int(node), None, 0))
        self._chk(n, "glome_sb_show_tex_materials")
        out = (C.c_int32 * max(1, n))()
        self._chk(int(self.lib.glome_sb_show_tex_materials(self.h, int(node), out, n)), "glome_sb_show_tex_materials")
        return list(out)[:n]

    def load_show(self, text, tex_materials=(), default_material=-1):
        """Read a `show geom` text: returns (root node, number of Tex constructors).  The k-th Tex gets tex_materials[k],
        later ones default_material (textures are closures in the reference and print as "Texture")."""
        mats = (C.c_int32 * max(1, len(tex_materials)))(*[int(m) for m in tex_materials])
        nt = C.c_int32(0)
        root = self._chk(self.lib.glome_sb_load_show(self.h, text.encode() if isinstance(text, str) else text, mats, len(tex_materials), int(default_material), C.byref(nt)),
                         "glome_sb_load_show")
        return root, nt.value

    def material_blend_fn(self, a, b, fn, params):
        """Blend a b (f pos): fn = WEIGHT_PERLIN (params = [scale]) or WEIGHT_STRIPE_* (params = axis), TestScene.hs:214-234."""
        wp = (C.c_double * 4)(*([float(x) for x in params] + [0.0] * (4 - len(params))))
        return self._chk(self.lib.glome_sb_material_blend_fn(self.h, int(a), int(b), int(fn), wp), "glome_sb_material_blend_fn")

    def material_warp(self, frame, scene, lights, xfm):
        """Warp frame scene' lights' xfm (Shader.hs:47-50): scene = a node or None (the root the scene is committed with);
        lights = glome_light structs (api.light); xfm = the 24 doubles of the transform M of the closure
        \\ray hit -> xfm_ray M (Ray (pos hit) (vnorm (dir ray))) (the portal, TestScene.hs:166-172)."""
        la = (L.Light * max(1, len(lights)))(*lights)
        x = np.ascontiguousarray(np.asarray(xfm, dtype=np.float64).ravel())
        return self._chk(self.lib.glome_sb_material_warp(self.h, int(frame), -1 if scene is None else int(scene), la, len(lights), x.ctypes.data_as(L.c_dp)), "glome_sb_material_warp")

    # host-side inspection
    def primcount(self, node):
        out = (C.c_long * 3)()
        self._chk(self.lib.glome_sb_primcount(self.h, int(node), out), "glome_sb_primcount")
        return tuple(out)

    def bound(self, node):
        out = np.zeros(6)
        self._chk(self.lib.glome_sb_bound(self.h, int(node), out.ctypes.data_as(L.c_dp)), "glome_sb_bound")
        return out

    def bih_dump(self, node):
        n = self.lib.glome_sb_bih_dump(self.h, int(node), 0, None, None, None, None, None, 0)
        self._chk(int(n), "glome_sb_bih_dump")
        ls, rs = np.zeros(n), np.zeros(n)
        ax, nl = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        cap = 1 << 24
        lp = np.zeros(cap, dtype=np.int32)
        ip = C.POINTER(C.c_int)
        self.lib.glome_sb_bih_dump(self.h, int(node), n, ls.ctypes.data_as(L.c_dp), rs.ctypes.data_as(L.c_dp), ax.ctypes.data_as(ip),
                                   nl.ctypes.data_as(ip), lp.ctypes.data_as(L.c_ip), cap)
        return ls, rs, ax, nl, lp[:int(nl.sum())]


# ---------------------------------------------------------------- device objects
class Context:
    def __init__(self, device=0):
        self.lib = L.load()
        self.h = self.lib.glome_ctx_create(int(device))
        if not self.h:
            raise GlomeError("glome_ctx_create failed: " + self.lib.glome_global_error().decode() + " (no CPU fallback exists)")

    def close(self):
        if self.h:
            self.lib.glome_ctx_destroy(self.h)
            self.h = None

    def err(self):
        return self.lib.glome_last_error(self.h).decode()

    def synchronize(self):
        if self.lib.glome_ctx_synchronize(self.h) != 0:
            raise GlomeError(self.err())

    def device_info(self):
        name = C.create_string_buffer(256)
        cu, ws = C.c_int(), C.c_int()
        self.lib.glome_ctx_device_info(self.h, name, 256, C.byref(cu), C.byref(ws))
        return name.value.decode(), cu.value, ws.value

    def bih(self, builder, ids):
        """`bih ids` built on this context's GPU (glome_sb_bih_dev): the node glome_sb_bih makes, tree bit for bit.
        Returns (node, device milliseconds)."""
        arr = (C.c_int32 * max(1, len(ids)))(*[int(i) for i in ids])
        ms = C.c_float(0)
        rc = self.lib.glome_sb_bih_dev(self.h, builder.h, arr, len(ids), C.byref(ms))
        if rc < 0:
            raise GlomeError(f"glome_sb_bih_dev: {self.err()} (status {rc})")
        return rc, ms.value

    def mesh(self, builder, verts, norms, tris, mats):
        """`mesh` with its BVH built on this context's GPU (glome_sb_mesh_dev).  Returns (node, device milliseconds)."""
        v = np.ascontiguousarray(np.asarray(verts, dtype=np.float64).reshape(-1, 3))
        n = np.ascontiguousarray(np.asarray(norms, dtype=np.float64).reshape(-1, 3))
        t = np.ascontiguousarray(np.asarray(tris, dtype=np.int32).reshape(-1, 8))
        m = np.ascontiguousarray(np.asarray(mats, dtype=np.int32).ravel())
        ms = C.c_float(0)
        rc = self.lib.glome_sb_mesh_dev(self.h, builder.h, v.ctypes.data_as(L.c_dp), v.shape[0], n.ctypes.data_as(L.c_dp), n.shape[0],
                                        t.ctypes.data_as(L.c_ip), t.shape[0], m.ctypes.data_as(L.c_ip), m.shape[0], C.byref(ms))
        if rc < 0:
            raise GlomeError(f"glome_sb_mesh_dev: {self.err()} (status {rc})")
        return rc, ms.value

    def commit(self, builder, root):
        s = self.lib.glome_scene_commit(self.h, builder.h, int(root))
        if not s:
            raise GlomeError("glome_scene_commit: " + self.err())
        return Scene(self, s)


def _stats_dict(st):
    return {k: getattr(st, k) for k, _ in L.Stats._fields_}


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32).ravel())


class Scene:
    def __init__(self, ctx, h):
        self.ctx, self.lib, self.h = ctx, ctx.lib, h

    def release(self):
        if self.h:
            self.lib.glome_scene_release(self.h)
            self.h = None

    def info(self):
        i = L.SceneInfo()
        self.lib.glome_scene_get_info(self.h, C.byref(i))
        return {k: getattr(i, k) for k, _ in L.SceneInfo._fields_}

    def _chk(self, rc, what):
        if rc != 0:
            raise GlomeError(f"{what}: {self.ctx.err()} (status {rc})")

    def _rays(self, o, d, tmax):
        o = np.asarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.asarray(d, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        tm = np.broadcast_to(np.asarray(tmax, dtype=np.float32), (n,))
        cols = [_f32(o[:, 0]), _f32(o[:, 1]), _f32(o[:, 2]), _f32(d[:, 0]), _f32(d[:, 1]), _f32(d[:, 2]), _f32(tm)]
        return n, cols

    def rayint(self, o, d, tmax=1000000.0):
        """rayint over a batch (Solid.hs:146-151): returns dict t (-1 = miss), prim, n (nx3), tex (nx8: the stack innermost first, -1 padded)."""
        n, cols = self._rays(o, d, tmax)
        t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32)
        nx = np.zeros(n, np.float32); ny = np.zeros(n, np.float32); nz = np.zeros(n, np.float32)
        tex = np.zeros((n, self.lib.glome_tex_words()), np.int32)  # (8: GLOME_TEX_WORDS of the loaded library)
        self._chk(self.lib.glome_rayint_batch(self.h, n, *[c.ctypes.data_as(L.c_fp) for c in cols], t.ctypes.data_as(L.c_fp),
                                              prim.ctypes.data_as(L.c_ip), nx.ctypes.data_as(L.c_fp), ny.ctypes.data_as(L.c_fp),
                                              nz.ctypes.data_as(L.c_fp), tex.ctypes.data_as(L.c_ip)), "glome_rayint_batch")
        return {"t": t, "prim": prim, "n": np.stack([nx, ny, nz], 1), "tex": tex}

    def shadow(self, o, d, tmax):
        n, cols = self._rays(o, d, tmax)
        occ = np.zeros(n, np.uint8)
        self._chk(self.lib.glome_shadow_batch(self.h, n, *[c.ctypes.data_as(L.c_fp) for c in cols], occ.ctypes.data_as(L.c_bp)), "glome_shadow_batch")
        return occ.astype(bool)

    def inside(self, p):
        p = np.asarray(p, dtype=np.float32).reshape(-1, 3)
        n = p.shape[0]
        ins = np.zeros(n, np.uint8)
        cols = [_f32(p[:, 0]), _f32(p[:, 1]), _f32(p[:, 2])]
        self._chk(self.lib.glome_inside_batch(self.h, n, *[c.ctypes.data_as(L.c_fp) for c in cols], ins.ctypes.data_as(L.c_bp)), "glome_inside_batch")
        return ins.astype(bool)

    def render(self, cam, lights, params, want_packed=True, init=None):
        """renderTiles (Glome.hs:379-386): returns (rgbad[h,w,5] float32, packed[h,w] uint32 or None, stats dict)."""
        w, h = params.width, params.height
        img = np.zeros((h, w, 5), np.float32) if init is None else np.ascontiguousarray(init, dtype=np.float32)
        packed = np.zeros((h, w), np.uint32) if want_packed else None
        la = (L.Light * max(1, len(lights)))(*lights)
        st = L.Stats()
        self._chk(self.lib.glome_render(self.h, C.byref(cam), la, len(lights), C.byref(params), img.ctypes.data_as(L.c_fp),
                                        packed.ctypes.data_as(L.c_up) if want_packed else None, C.byref(st)), "glome_render")
        return img, packed, _stats_dict(st)

    def render_dev(self, cam, lights, params, rgbad_ptr, packed_ptr=None, want_stats=True):
        """Device-pointer render (e.g. a torch tensor's data_ptr()); asynchronous unless want_stats."""
        la = (L.Light * max(1, len(lights)))(*lights)
        st = L.Stats()
        self._chk(self.lib.glome_render_dev(self.h, C.byref(cam), la, len(lights), C.byref(params), C.c_void_p(rgbad_ptr) if rgbad_ptr else None,
                                            C.c_void_p(packed_ptr) if packed_ptr else None, C.byref(st) if want_stats else None), "glome_render_dev")
        return _stats_dict(st) if want_stats else None


class Multi:
    """Several GPUs driven by this one process (glome_multi_*): scenes[i] is the scene committed on context i; frames land
    on scenes[0]'s device.  The one-process counterpart of dist.ShardedFrame (one process per GPU)."""

    def __init__(self, scenes, params, use_rccl=True, transport=None):
        """transport: "direct" (ranks store straight into rank 0's framebuffer), "rccl", "peer-copy"; default: rccl when use_rccl else peer-copy"""
        self.lib = scenes[0].lib
        self.scenes = list(scenes)
        arr = (C.c_void_p * len(scenes))(*[s.h for s in scenes])
        code = {"direct": 2, "rccl": 1, "peer-copy": 0}[transport] if transport is not None else (1 if use_rccl else 0)
        self.h = self.lib.glome_multi_create(arr, len(scenes), C.byref(params), code)
        if not self.h:
            raise GlomeError("glome_multi_create: " + self.lib.glome_global_error().decode())
        self.params = params

    def transport(self):
        return self.lib.glome_multi_transport(self.h).decode()

    def render(self, cams, lights, packed_ptr):
        """cams: one glome_camera or a list of up to 32; packed_ptr: device pointer on rank 0's GPU (frames back to back)"""
        cams = list(cams) if isinstance(cams, (list, tuple)) else [cams]
        ca = (L.Camera * len(cams))(*cams)
        la = (L.Light * max(1, len(lights)))(*lights)
        rc = self.lib.glome_multi_render(self.h, ca, len(cams), la, len(lights), C.c_void_p(packed_ptr))
        if rc != 0:
            raise GlomeError(f"glome_multi_render: {self.lib.glome_multi_last_error(self.h).decode()} (status {rc})")

    def synchronize(self):
        rc = self.lib.glome_multi_synchronize(self.h)
        if rc != 0:
            raise GlomeError(f"glome_multi_synchronize: {self.lib.glome_multi_last_error(self.h).decode()} (status {rc})")

    def close(self):
        if self.h:
            self.lib.glome_multi_destroy(self.h)
            self.h = None
