"""ctypes binding of oracle/liboracle.so -- the CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  It offers the same
constructor names as glome_amd.api.Builder so a glome_amd.scene.SceneDesc can be replayed into it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


def build(verbose=False):
    subprocess.check_call(["make", "-C", HERE] + ([] if verbose else ["-s"]))
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.glo_new.restype = C.c_void_p
        _lib.glo_last_error.restype = C.c_char_p
        _lib.glo_bih_dump.restype = C.c_long
        _lib.glo_rgbf.restype = C.c_uint32
    return _lib


def _d(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel())


def _dp(a):
    return a.ctypes.data_as(c_dp)


class OracleError(RuntimeError):
    pass


class Oracle:
    """A scene builder + renderer in double (default) or float arithmetic."""

    def __init__(self, use_float=False):
        self.L = lib()
        self.h = C.c_void_p(self.L.glo_new(1 if use_float else 0))

    def __del__(self):
        try:
            self.L.glo_free(self.h)
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            raise OracleError(f"{what}: {self.L.glo_last_error(self.h).decode()}")
        return rc

    def _call(self, name, *args):
        keep, cargs = [], []
        for a in args:
            if isinstance(a, float):
                cargs.append(C.c_double(a))
            elif isinstance(a, (int, np.integer)):
                cargs.append(C.c_int(int(a)))
            else:
                arr = _d(a)
                keep.append(arr)
                cargs.append(_dp(arr))
        return self._chk(getattr(self.L, name)(self.h, *cargs), name)

    def _ids(self, name, ids):
        arr = np.ascontiguousarray(np.asarray(ids, dtype=np.int32))
        return self._chk(getattr(self.L, name)(self.h, arr.ctypes.data_as(c_ip), C.c_int(len(arr))), name)

    def sphere(self, c, r): return self._call("glo_sphere", c, float(r))
    def triangle(self, p1, p2, p3): return self._call("glo_triangle", list(p1) + list(p2) + list(p3))
    def trianglenorm(self, p1, p2, p3, n1, n2, n3): return self._call("glo_trianglenorm", list(p1) + list(p2) + list(p3), list(n1) + list(n2) + list(n3))
    def box(self, a, b): return self._call("glo_box", a, b)
    def plane(self, pt, n): return self._call("glo_plane", pt, n)
    def plane_offset(self, n, off): return self._call("glo_plane_offset", n, float(off))
    def disc(self, pos, n, r): return self._call("glo_disc", pos, n, float(r))
    def cylinder(self, p1, p2, r): return self._call("glo_cylinder", p1, p2, float(r))
    def cone(self, p1, r1, p2, r2): return self._call("glo_cone", p1, float(r1), p2, float(r2))
    def group(self, ids): return self._ids("glo_group", ids)
    def intersection(self, ids): return self._ids("glo_intersection", ids)
    def bih(self, ids): return self._ids("glo_bih", ids)

    def triangles_bulk(self, pts9):
        pts9 = np.ascontiguousarray(pts9, dtype=np.float64).reshape(-1, 9)
        f = self.L.glo_triangle
        base = pts9.ctypes.data
        return [f(self.h, C.cast(base + 72 * k, c_dp)) for k in range(pts9.shape[0])]

    def transform(self, node, xfms):
        arr = np.ascontiguousarray(np.asarray(xfms, dtype=np.float64).reshape(-1, 24))
        return self._chk(self.L.glo_transform(self.h, C.c_int(int(node)), _dp(arr), C.c_int(arr.shape[0])), "glo_transform")

    def difference(self, a, b): return self._chk(self.L.glo_difference(self.h, C.c_int(int(a)), C.c_int(int(b))), "glo_difference")
    def difference_retexture(self, a, b): return self._chk(self.L.glo_difference_retexture(self.h, C.c_int(int(a)), C.c_int(int(b))), "glo_difference_retexture")

    def mesh(self, verts, norms, tris, mats):
        v = np.ascontiguousarray(np.asarray(verts, dtype=np.float64).reshape(-1, 3))
        n = np.ascontiguousarray(np.asarray(norms, dtype=np.float64).reshape(-1, 3))
        t = np.ascontiguousarray(np.asarray(tris, dtype=np.int32).reshape(-1, 8))
        m = np.ascontiguousarray(np.asarray(mats, dtype=np.int32).ravel())
        return self._chk(self.L.glo_mesh(self.h, _dp(v), C.c_int(v.shape[0]), _dp(n), C.c_int(n.shape[0]), t.ctypes.data_as(c_ip), C.c_int(t.shape[0]),
                                         m.ctypes.data_as(c_ip), C.c_int(m.shape[0])), "glo_mesh")

    def tex(self, node, mat): return self._chk(self.L.glo_tex(self.h, C.c_int(int(node)), C.c_int(int(mat))), "glo_tex")
    def tag(self, node, _t=None): return self._chk(self.L.glo_tag(self.h, C.c_int(int(node))), "glo_tag")
    def noshadow(self, node): return self._chk(self.L.glo_noshadow(self.h, C.c_int(int(node))), "glo_noshadow")
    def onlyshadow(self, node): return self._chk(self.L.glo_onlyshadow(self.h, C.c_int(int(node))), "glo_onlyshadow")
    def bound_object(self, a, b): return self._chk(self.L.glo_bound_object(self.h, C.c_int(int(a)), C.c_int(int(b))), "glo_bound_object")
    def innerbound(self, a, b): return self._chk(self.L.glo_innerbound(self.h, C.c_int(int(a)), C.c_int(int(b))), "glo_innerbound")
    def flatten_transform(self, node): return self._chk(self.L.glo_flatten_transform(self.h, C.c_int(int(node))), "glo_flatten_transform")
    def tolist(self, node): return self._chk(self.L.glo_tolist(self.h, C.c_int(int(node))), "glo_tolist")
    def bih_tolist(self, node): return self._chk(self.L.glo_bih_tolist(self.h, C.c_int(int(node))), "glo_bih_tolist")
    def material_surface(self, color, alpha, amb, kd, ks, shine): return self._call("glo_material_surface", color, float(alpha), float(amb), float(kd), float(ks), float(shine))
    def material_reflect(self, refl): return self._call("glo_material_reflect", float(refl))
    def material_refract(self, refl, refr, ior): return self._call("glo_material_refract", float(refl), float(refr), float(ior))
    def material_layers(self, mats): return self._ids("glo_material_layers", mats)
    def material_blend(self, a, b, w): return self._chk(self.L.glo_material_blend(self.h, C.c_int(int(a)), C.c_int(int(b)), C.c_double(float(w))), "glo_material_blend")

    def material_blend_fn(self, a, b, fn, params):
        wp = (C.c_double * 4)(*([float(x) for x in params] + [0.0] * (4 - len(params))))
        return self._chk(self.L.glo_material_blend_fn(self.h, C.c_int(int(a)), C.c_int(int(b)), C.c_int(int(fn)), wp), "glo_material_blend_fn")

    def material_warp(self, frame, scene, lights, xfm):
        """lights: [(pos, color, rad, shadow)] or glome_light structs"""
        rows = []
        for l in lights:
            if isinstance(l, (tuple, list)):
                pos, col, rad, sh = l
            else:
                pos, col, rad, sh = list(l.pos), list(l.color), l.rad, l.shadow
            rows += [float(x) for x in pos] + [float(x) for x in col] + [float(rad), 1.0 if sh else 0.0]
        l8 = _d(rows if rows else [0.0] * 8)
        x = _d(np.asarray(xfm, dtype=np.float64).ravel())
        return self._chk(self.L.glo_material_warp(self.h, C.c_int(int(frame)), C.c_int(-1 if scene is None else int(scene)), _dp(l8), C.c_int(len(lights)), _dp(x)), "glo_material_warp")

    # ---- scene state ----
    def set_root(self, node): self._chk(self.L.glo_set_root(self.h, C.c_int(int(node))), "glo_set_root")
    def set_camera_vectors(self, pos, fwd, up, right):
        a = _d(list(pos) + list(fwd) + list(up) + list(right))
        self._chk(self.L.glo_set_camera(self.h, _dp(a)), "glo_set_camera")
    def clear_lights(self): self.L.glo_clear_lights(self.h)
    def add_light(self, pos, col, rad=1000000.0, shadow=True):
        a, b = _d(pos), _d(col)
        self._chk(self.L.glo_add_light(self.h, _dp(a), _dp(b), C.c_double(float(rad)), C.c_int(1 if shadow else 0)), "glo_add_light")

    # ---- queries ----
    def rayint(self, root, o, d, tmax=1000000.0):
        o = np.asarray(o, dtype=np.float64).reshape(-1, 3)
        d = np.asarray(d, dtype=np.float64).reshape(-1, 3)
        n = o.shape[0]
        tm = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, dtype=np.float64), (n,)))
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        t = np.zeros(n); prim = np.zeros(n, np.int32); pos = np.zeros((n, 3)); nrm = np.zeros((n, 3))
        tex = np.zeros((n, 8), np.int32); ntex = np.zeros(n, np.int32)
        self._chk(self.L.glo_rayint_batch(self.h, C.c_int(int(root)), C.c_size_t(n), *[_dp(c) for c in cols], _dp(tm), _dp(t), prim.ctypes.data_as(c_ip),
                                          _dp(pos), _dp(nrm), tex.ctypes.data_as(c_ip), ntex.ctypes.data_as(c_ip)), "glo_rayint_batch")
        return {"t": t, "prim": prim, "pos": pos, "n": nrm, "tex": tex, "ntex": ntex}

    def shadow(self, root, o, d, tmax):
        o = np.asarray(o, dtype=np.float64).reshape(-1, 3)
        d = np.asarray(d, dtype=np.float64).reshape(-1, 3)
        n = o.shape[0]
        tm = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, dtype=np.float64), (n,)))
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        occ = np.zeros(n, np.uint8)
        self._chk(self.L.glo_shadow_batch(self.h, C.c_int(int(root)), C.c_size_t(n), *[_dp(c) for c in cols], _dp(tm), occ.ctypes.data_as(C.POINTER(C.c_uint8))), "glo_shadow_batch")
        return occ.astype(bool)

    def inside(self, root, p):
        p = np.asarray(p, dtype=np.float64).reshape(-1, 3)
        n = p.shape[0]
        cols = [np.ascontiguousarray(p[:, k]) for k in range(3)]
        ins = np.zeros(n, np.uint8)
        self._chk(self.L.glo_inside_batch(self.h, C.c_int(int(root)), C.c_size_t(n), *[_dp(c) for c in cols], ins.ctypes.data_as(C.POINTER(C.c_uint8))), "glo_inside_batch")
        return ins.astype(bool)

    def render(self, width, height, mode=0, blocksize=65, maxdepth=3, fog=0, tile_first=0, tile_stride=1, thresholds=(0.14, 0.15, 0.16, 0.18),
               nthreads=1, max_tiles=0, want_packed=True):
        ip = np.array([width, height, mode, blocksize, maxdepth, fog, tile_first, tile_stride], dtype=np.int32)
        th = _d(thresholds)
        out = np.zeros((height, width, 5))
        packed = np.zeros((height, width), np.uint32) if want_packed else None
        cnt = np.zeros(6, np.uint64)
        self._chk(self.L.glo_render(self.h, ip.ctypes.data_as(c_ip), _dp(th), _dp(out), packed.ctypes.data_as(C.POINTER(C.c_uint32)) if want_packed else None,
                                    C.c_int(nthreads), C.c_int(max_tiles), cnt.ctypes.data_as(C.POINTER(C.c_uint64))), "glo_render")
        names = ["bih_nodes", "mesh_nodes", "prim_tests", "rays_primary", "rays_shadow", "rays_secondary"]
        return out, packed, dict(zip(names, [int(x) for x in cnt]))

    def primcount(self, node):
        out = (C.c_long * 3)()
        self._chk(self.L.glo_primcount(self.h, C.c_int(int(node)), out), "glo_primcount")
        return tuple(out)

    def bound(self, node):
        out = np.zeros(6)
        self._chk(self.L.glo_bound(self.h, C.c_int(int(node)), _dp(out)), "glo_bound")
        return out

    def kind(self, node):
        buf = C.create_string_buffer(64)
        self._chk(self.L.glo_kind_name(self.h, C.c_int(int(node)), buf, C.c_int(64)), "glo_kind_name")
        return buf.value.decode()

    def bih_dump(self, node):
        n = self.L.glo_bih_dump(self.h, C.c_int(int(node)), C.c_long(0), None, None, None, None, None, C.c_long(0))
        if n < 0:
            raise OracleError(self.L.glo_last_error(self.h).decode())
        ls, rs = np.zeros(n), np.zeros(n)
        ax, nl = np.zeros(n, np.int32), np.zeros(n, np.int32)
        cap = 1 << 24
        lp = np.zeros(cap, np.int32)
        self.L.glo_bih_dump(self.h, C.c_int(int(node)), C.c_long(n), _dp(ls), _dp(rs), ax.ctypes.data_as(c_ip), nl.ctypes.data_as(c_ip), lp.ctypes.data_as(c_ip), C.c_long(cap))
        return ls, rs, ax, nl, lp[:int(nl.sum())]


def camera_vectors(pos, at, up, angle):
    """camera pos at up angle (Scene.hs:48-57) in double: returns (pos, fwd, up, right)."""
    out = np.zeros(12)
    a, b, c = _d(pos), _d(at), _d(up)
    lib().glo_camera(_dp(a), _dp(b), _dp(c), C.c_double(float(angle)), _dp(out))
    return out.reshape(4, 3)


def load_scene(sd, use_float=False, camera_fp32=True):
    """Replay a glome_amd.scene.SceneDesc into a fresh Oracle; returns (oracle, node map, material map).
    The camera basis is rounded to fp32 (as the device receives it) unless camera_fp32=False."""
    o = Oracle(use_float)
    nmap, mmap = sd.replay(o)
    if sd.root is not None:
        o.set_root(nmap[sd.root])
    for pos, col, rad, sh in sd.lights:
        o.add_light(pos, col, rad, sh)
    if sd.cam is not None:
        cv = camera_vectors(*sd.cam)
        if camera_fp32:
            cv = cv.astype(np.float32).astype(np.float64)
        o.set_camera_vectors(*cv)
    return o, nmap, mmap
