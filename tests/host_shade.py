"""The closure fallback of the drop-in boundary, exercised -- test infrastructure.

glome's textures are closures `Ray -> Rayint -> Material` (Solid.hs:97).  The C ABI takes them defunctionalised; a host whose
textures are general closures keeps `trace` / `mpreshade` / `mpostshade` on ITS side and asks the library only for what it
cannot do fast itself: `rayint` and `shadow` over batches (glome_rayint_batch / glome_shadow_batch).  This file is that host:
Trace.trace (Trace.hs:59-82) with the Shader of Shader.hs:65-118 (Surface and Reflect), batch by batch in numpy, the texture of
a hit chosen by a Python closure.  `backend` is anything with rayint(o, d, tmax) -> {t, n, tex} and shadow(o, d, tmax) -> bool
arrays: glome_amd.api.Scene (the C ABI on the GPU) or helpers.HostSim (the same device headers compiled for the host).
With closures that are `t_uniform m` the image must be the one glome_render makes of the same scene."""
import numpy as np

DELTA = np.float32(0.0001)
KINF = np.float32(1000000.0)


def primary_rays(cam, w, h):
    """get_rayint / getCoordsf (Glome.hs:27-33, 119-140) for every pixel, in fp32 like the kernels (exact quotients)."""
    f = np.float32
    px, py = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    xc = ((f(px / w) * f(2) - f(1)) * f(np.float64(w) / np.float64(h))).astype(np.float32)
    yc = (-(f(py / h) * f(2) - f(1))).astype(np.float32)
    fwd, up, right, pos = (np.array(list(v), np.float32) for v in (cam.fwd, cam.up, cam.right, cam.pos))
    d = fwd[None, None, :] + right[None, None, :] * (-xc)[..., None] + up[None, None, :] * yc[..., None]
    d = d.astype(np.float32)
    inv = (f(1) / np.sqrt((d * d).sum(-1, dtype=np.float32))).astype(np.float32)
    d = d * inv[..., None]
    o = np.broadcast_to(pos, d.shape)
    return np.ascontiguousarray(o.reshape(-1, 3)), np.ascontiguousarray(d.reshape(-1, 3).astype(np.float32))


class HostShader:
    """textures: dict tex id (as the backend reports it in a hit's stack) -> closure (o, d, hit) -> material tuple, where a material
    is ("surface", color, alpha, amb, kd, ks, shine) or ("reflect", refl).  lights: [(pos, color, rad, shadow)]."""

    def __init__(self, backend, textures, lights):
        self.b, self.textures, self.lights = backend, textures, lights
        self.rays = [0, 0, 0]  # primary, shadow, secondary -- counted like the kernels do

    def _preshade(self, hit, idx):
        """mpreshade (Shader.hs:65-80) for the hits `idx`: per light (colour * falloff, ldir, lit mask)"""
        p, n = hit["p"][idx], hit["n"][idx]
        out = []
        for (lpos, lcol, rad, do_shadow) in self.lights:
            lvec = np.asarray(lpos, np.float64)[None, :] - p
            facing = (lvec * n).sum(1) >= 0
            llen = np.sqrt((lvec * lvec).sum(1))
            ldir = lvec / llen[:, None]
            lit = facing & ~(llen > rad)
            if do_shadow and lit.any():
                k = np.nonzero(lit)[0]
                so = (p[k] + n[k] * np.float64(DELTA)).astype(np.float32)
                occ = self.b.shadow(so, ldir[k].astype(np.float32), (llen[k] - 2 * np.float64(DELTA)).astype(np.float32))
                self.rays[1] += len(k)
                lit[k[occ]] = False
            out.append((np.asarray(lcol, np.float64)[None, :] / (llen * llen)[:, None], ldir, lit))
        return out

    def _postshade(self, mat, o, d, hit, idx, lights, recurs):
        """mpostshade (Shader.hs:82-118) of one material for the hits `idx` -> ColorA [len(idx), 4]"""
        n = hit["n"][idx]
        if mat[0] == "surface":
            _, color, alpha, amb, kd, ks, shine = mat
            eyedir = -d[idx].astype(np.float64)
            c = np.tile(np.asarray(color, np.float64) * amb, (len(idx), 1))
            for (lc, ldir, lit) in lights():
                half = ldir + eyedir
                half = half / np.sqrt((half * half).sum(1))[:, None]  # bisect, Vec.hs:331-332
                ldotn = np.maximum(0, (ldir * n).sum(1))
                if ks <= DELTA:
                    blinn = np.zeros(len(idx))
                else:
                    with np.errstate(invalid="ignore"):
                        bl = np.maximum(0, np.power((half * n).sum(1), shine) * ldotn)
                    blinn = np.where(np.isnan(bl), 0.0, bl)
                diffuse = (ldir * n).sum(1)
                c = c + np.where(lit[:, None], lc * (blinn * ks + diffuse * kd)[:, None], 0.0)
            return np.concatenate([c, np.full((len(idx), 1), float(alpha))], 1)
        if mat[0] == "reflect":
            refl = mat[1]
            if refl > 0 and recurs > 0:
                dd = d[idx].astype(np.float64)
                outdir = dd - n * (2 * (dd * n).sum(1))[:, None]  # reflect, Vec.hs:335-337
                ro = (hit["p"][idx] + outdir * np.float64(DELTA)).astype(np.float32)
                if recurs - 1 > 0:
                    self.rays[2] += len(idx)
                c = self.trace(ro, outdir.astype(np.float32), recurs - 1)
                c[:, 3] *= refl
                return c
            return np.tile(np.array([0.0, 0.0, 0.0, 1.0]), (len(idx), 1))
        raise ValueError("host_shade: material kind " + mat[0])

    def trace(self, o, d, recurs):
        """Trace.trace (Trace.hs:59-82) over a batch -> ColorA [n, 4]"""
        n = len(o)
        acc = np.zeros((n, 4))
        if recurs == 0 or n == 0:
            return acc
        r = self.b.rayint(o, d, KINF)
        t = r["t"].astype(np.float64)
        ishit = t >= 0
        hit = {"t": t, "n": r["n"].astype(np.float64), "p": (o + d * r["t"][:, None]).astype(np.float32).astype(np.float64), "hit": ishit}
        cache = {}

        def lights_for(idx):  # the lazily evaluated ctxb (Trace.hs:63): computed when the first Surface of a hit asks for it
            def get():
                key = idx.tobytes()
                if key not in cache:
                    cache[key] = self._preshade(hit, idx)
                return cache[key]
            return get
        tex = r["tex"]
        for k in range(tex.shape[1]):
            want = ishit & (tex[:, k] >= 0) & ~(acc[:, 3] + np.float64(DELTA) >= 1)  # opaque, Trace.hs:50-51
            if k > 0:
                want &= (tex[:, :k] >= 0).all(1)
            for tid in np.unique(tex[want, k]):
                idx = np.nonzero(want & (tex[:, k] == tid))[0]
                mat = self.textures[int(tid)](o[idx], d[idx], {key: v[idx] for key, v in hit.items()})
                if mat[0] == "split":  # the closure chose per hit: (mask, material where true, material where false)
                    c2 = np.zeros((len(idx), 4))
                    for sel, m in ((mat[1], mat[2]), (~mat[1], mat[3])):
                        if sel.any():
                            c2[sel] = self._postshade(m, o, d, hit, idx[sel], lights_for(idx[sel]), recurs)
                else:
                    c2 = self._postshade(mat, o, d, hit, idx, lights_for(idx), recurs)
                c1 = acc[idx]
                trans = 1 - c1[:, 3]
                acc[idx] = np.concatenate([c1[:, :3] + c2[:, :3] * (trans * c2[:, 3])[:, None], (c1[:, 3] + c2[:, 3] * trans)[:, None]], 1)  # cafold, Clr.hs:106-113
        return acc


def render_with_host_shading(backend, cam, lights, textures, w, h, maxdepth):
    hs = HostShader(backend, textures, lights)
    o, d = primary_rays(cam, w, h)
    hs.rays[0] += len(o)
    # the depth channel: the primary batch's own Rayint (asked for once more here; a host that cares keeps it from trace)
    r0 = backend.rayint(o, d, KINF)
    depth = np.where(r0["t"] >= 0, r0["t"], KINF).astype(np.float32)
    c = hs.trace(o, d, maxdepth)
    img = np.concatenate([c, depth[:, None].astype(np.float64)], 1).reshape(h, w, 5)
    return img, hs.rays


def uniform_textures(sd, mmap):
    """`t_uniform m` (Solid.hs / TestScene.hs:201-245) for every Surface / Reflect material of a SceneDesc: tex id -> closure"""
    out, k = {}, 0
    for kind, name, args in sd.ops:
        if kind != "m":
            continue
        if name == "material_surface":
            color, alpha, amb, kd, ks, shine = args[:6]
            mat = ("surface", tuple(float(x) for x in color), float(alpha), float(amb), float(kd), float(ks), float(shine))
        elif name == "material_reflect":
            mat = ("reflect", float(args[0]))
        else:
            mat = None
        if mat is not None:
            out[int(mmap[k])] = (lambda m: (lambda o, d, hit: m))(mat)
        k += 1
    return out

