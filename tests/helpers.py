"""Shared test helpers: the oracle binding, the host-compiled mirror of the device headers (tests/hostsim),
ray generators and comparison metrics."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle_py as O  # noqa: E402  (test infrastructure)

c_fp = C.POINTER(C.c_float)


def hostsim_lib():
    d = os.path.join(ROOT, "tests", "hostsim")
    subprocess.check_call(["make", "-s", "-C", d])
    lib = C.CDLL(os.path.join(d, "libhostsim.so"))
    lib.hostsim_commit.restype = C.c_void_p
    return lib


class HostSim:
    """The device per-ray code compiled for the host, over the product's own flattened scene (no GPU)."""

    def __init__(self, builder, root):
        self.lib = hostsim_lib()
        err = C.create_string_buffer(512)
        self.h = C.c_void_p(self.lib.hostsim_commit(C.c_void_p(builder.h), C.c_int(int(root)), err, 512))
        if not self.h:
            raise RuntimeError("hostsim_commit: " + err.value.decode())

    def info(self):
        out = (C.c_int * 6)()
        self.lib.hostsim_info(self.h, out)
        return dict(zip(["tier", "nesting", "max_bih_depth", "max_mesh_depth", "n_entries", "n_recs"], list(out)))

    @staticmethod
    def _cols(o, d, tmax):
        o = np.asarray(o, np.float32).reshape(-1, 3); d = np.asarray(d, np.float32).reshape(-1, 3)
        n = o.shape[0]
        tm = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, np.float32), (n,)))
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)] + [tm]
        return n, cols

    def rayint(self, o, d, tmax=1e6, tier=-1, analysis=0):
        n, cols = self._cols(o, d, tmax)
        t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); nrm = np.zeros((n, 3), np.float32); tex = np.zeros((n, 8), np.int32)
        cnt = np.zeros(3, np.uint64)
        rc = self.lib.hostsim_rayint(self.h, tier, analysis, C.c_size_t(n), *[c.ctypes.data_as(c_fp) for c in cols], t.ctypes.data_as(c_fp),
                                     prim.ctypes.data_as(C.POINTER(C.c_int)), nrm.ctypes.data_as(c_fp), tex.ctypes.data_as(C.POINTER(C.c_int)),
                                     cnt.ctypes.data_as(C.POINTER(C.c_ulonglong)))
        if rc != 0:
            raise RuntimeError(f"hostsim_rayint rc={rc}")
        return {"t": t, "prim": prim, "n": nrm, "tex": tex, "counters": cnt}

    def shadow(self, o, d, tmax, tier=-1):
        n, cols = self._cols(o, d, tmax)
        occ = np.zeros(n, np.uint8)
        rc = self.lib.hostsim_shadow(self.h, tier, C.c_size_t(n), *[c.ctypes.data_as(c_fp) for c in cols], occ.ctypes.data_as(C.POINTER(C.c_ubyte)))
        if rc != 0:
            raise RuntimeError(f"hostsim_shadow rc={rc}")
        return occ.astype(bool)

    def inside(self, p):
        p = np.asarray(p, np.float32).reshape(-1, 3)
        n = p.shape[0]
        cols = [np.ascontiguousarray(p[:, k]) for k in range(3)]
        ins = np.zeros(n, np.uint8)
        rc = self.lib.hostsim_inside(self.h, C.c_size_t(n), *[c.ctypes.data_as(c_fp) for c in cols], ins.ctypes.data_as(C.POINTER(C.c_ubyte)))
        if rc != 0:
            raise RuntimeError(f"hostsim_inside rc={rc}")
        return ins.astype(bool)

    def render(self, cam, lights, width, height, maxdepth, tier=-1):
        camv = np.array(list(cam.pos) + list(cam.fwd) + list(cam.up) + list(cam.right), np.float32)
        lv = np.array([list(l.pos) + list(l.color) + [l.rad, float(l.shadow)] for l in lights], np.float32).reshape(-1, 8)
        out = np.zeros((height, width, 5), np.float32)
        cnt = np.zeros(3, np.uint64)
        rc = self.lib.hostsim_render(self.h, tier, camv.ctypes.data_as(c_fp), lv.ctypes.data_as(c_fp), len(lights), width, height, maxdepth,
                                     out.ctypes.data_as(c_fp), cnt.ctypes.data_as(C.POINTER(C.c_ulonglong)))
        if rc != 0:
            raise RuntimeError(f"hostsim_render rc={rc}")
        return out, cnt


def _hs_render_subsample(self, cam, lights, width, height, maxdepth, blocksize=65, thresholds=(0.14, 0.15, 0.16, 0.18), tier=-1):
    camv = np.array(list(cam.pos) + list(cam.fwd) + list(cam.up) + list(cam.right), np.float32)
    lv = np.array([list(l.pos) + list(l.color) + [l.rad, float(l.shadow)] for l in lights], np.float32).reshape(-1, 8)
    th = np.array(thresholds, np.float32)
    out = np.zeros((height, width, 5), np.float32)
    cnt = np.zeros(3, np.uint64)
    rc = self.lib.hostsim_render_subsample(self.h, tier, camv.ctypes.data_as(c_fp), lv.ctypes.data_as(c_fp), len(lights), width, height, maxdepth, blocksize,
                                           th.ctypes.data_as(c_fp), out.ctypes.data_as(c_fp), cnt.ctypes.data_as(C.POINTER(C.c_ulonglong)))
    if rc != 0:
        raise RuntimeError(f"hostsim_render_subsample rc={rc}")
    return out, cnt


HostSim.render_subsample = _hs_render_subsample


def product_camera_lights(sd):
    """The glome_camera / glome_light structs the product gets for a SceneDesc (fp32 fields)."""
    from glome_amd import api
    cam = api.camera(*sd.cam)
    lights = [api.light(p, c, r, s) for (p, c, r, s) in sd.lights]
    return cam, lights


def oracle_for(sd, use_float=False):
    """Oracle loaded with the same scene; its camera basis is the product's fp32 camera (bit-identical inputs, Q3)."""
    o, nmap, mmap = O.load_scene(sd, use_float=use_float)
    cam, _ = product_camera_lights(sd)
    o.set_camera_vectors(list(cam.pos), list(cam.fwd), list(cam.up), list(cam.right))
    return o, nmap, mmap


def random_rays(n, seed, center=(0, 1, 0), radius=14.0, spread=6.0):
    """Rays from a shell around `center` aimed at a jittered point near it; off-axis, unit directions (fp32)."""
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    o = np.asarray(center) + v * radius * rng.uniform(0.6, 1.4, size=(n, 1))
    tgt = np.asarray(center) + rng.uniform(-spread, spread, size=(n, 3))
    d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = o.astype(np.float32); d = d.astype(np.float32)
    d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    return o, d


def compare_hits(got_t, ref_t, rtol=1e-4):
    """Return (mismatch fraction of hit/miss, max relative t error over rays both hit, errors)."""
    gh, rh = got_t >= 0, ref_t >= 0
    mism = float(np.mean(gh != rh))
    both = gh & rh
    err = np.abs(got_t[both].astype(np.float64) - ref_t[both]) / np.maximum(1.0, np.abs(ref_t[both])) if both.any() else np.zeros(0)
    return mism, (float(err.max()) if err.size else 0.0), err


REL_FLOOR = 1e-3  # the floor of the true relative error below: a colour channel darker than this is compared absolutely (at 1e-7)


def compare_images(got, ref, tol=1e-4):
    """Per-pixel RGBA error vs the fp64 oracle, two ways:
      max / p999 / frac_over / mean -- |got-ref| / max(1, |ref|): colours live in [0, 1], so this is an ABSOLUTE error;
      rel_*                         -- |got-ref| / max(|ref|, REL_FLOOR): the north star's "1e-4 relative", true for every
                                       channel brighter than 0.001 (1e-4 of a 0.01-bright pixel is 1e-6, near fp32's ulp there)."""
    g = got[..., :4].astype(np.float64); r = ref[..., :4]
    d = np.abs(g - r)
    e = (d / np.maximum(1.0, np.abs(r))).max(axis=-1)
    rel = (d / np.maximum(REL_FLOOR, np.abs(r))).max(axis=-1)
    return {"max": float(e.max()), "p999": float(np.quantile(e, 0.999)), "frac_over": float(np.mean(e > tol)), "mean": float(e.mean()),
            "rel_max": float(rel.max()), "rel_p999": float(np.quantile(rel, 0.999)), "rel_frac_over": float(np.mean(rel > tol)), "rel_mean": float(rel.mean())}
