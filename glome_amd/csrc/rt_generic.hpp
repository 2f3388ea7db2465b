// rt_generic.hpp -- the generic tier's point queries: `inside` and `get_metainfo` of glome's `Solid` class
// (Solid.hs:138-254) over the flattened record table, for scenes the flat tier cannot express: Instance, CSG
// Difference / Intersection, Bound / InnerBound, BIHs whose items are composites, and any nesting of those.  Both
// are pure functions of a point, asked only by the CSG nodes and by Bound; the recursion is unrolled at compile
// time -- f<D> only ever calls f<D-1> -- and the host validates the nesting depth at commit (flatten.hpp).
// rayint and shadow, the methods every ray runs, are the explicit-frame loop of rt_generic_vm.hpp.
#pragma once
#include "rt_device.hpp"

namespace glome {

template <bool COUNT> struct GCtx {
  const DScene& S;
  Cnt& cnt;
  unsigned int& err;
};

template <int D, bool C> struct G;  // the four class methods at nesting budget D

template <int D, bool C> GD bool inside_g(GCtx<C>& g, U4 rec, V3 p) {
  while ((rec.x & RF_KINDMASK) == R_TEX) rec = ldu4(g.S.recs, rec.y);
  const uint32_t kind = rec.x & RF_KINDMASK;
  if (kind >= R_SPHERE && kind <= R_CONE) return prim_inside(g.S, kind, rec.y, p);
  return G<D, C>::inside(g, rec, p);
}
template <int D, bool C> GD TexStack meta_g(GCtx<C>& g, U4 rec, V3 p) { return G<D, C>::meta(g, rec, p); }


template <int D, bool C> struct G {
  using Ctx = GCtx<C>;
  static constexpr bool COMPOSITES = D > 0;

  // ------------------------------------------------------------------ inside
  static GDN bool inside(Ctx& g, U4 rec, V3 p) {
    const DScene& S = g.S;
    rec = skip_tex(S, rec);
    uint32_t kind = rec.x & RF_KINDMASK;
    if (kind >= R_SPHERE && kind <= R_CONE) return prim_inside(S, kind, rec.y, p);
    if constexpr (COMPOSITES) {
      switch (kind) {
        case R_LIST:
          for (uint32_t k = 0; k < rec.z; k++) if (inside_g<D - 1>(g, ldu4(S.recs, rec.y + k), p)) return true;
          return false;
        case R_INSTANCE: { Xf6 x = load_xf(S, rec.z); return inside_g<D - 1>(g, ldu4(S.recs, rec.y), mat_point(x.i0, x.i1, x.i2, p)); }  // Solid.hs:473-475
        case R_DIFF: return inside_g<D - 1>(g, ldu4(S.recs, rec.y), p) && !inside_g<D - 1>(g, ldu4(S.recs, rec.z), p);                  // Csg.hs:92-94
        case R_ISECT: { bool acc = true; for (uint32_t k = 0; k < rec.z; k++) acc = acc && inside_g<D - 1>(g, ldu4(S.recs, rec.y + k), p); return acc; }  // Csg.hs:96-101
        case R_BOUND: return inside_g<D - 1>(g, ldu4(S.recs, rec.y), p) && inside_g<D - 1>(g, ldu4(S.recs, rec.z), p);  // Bound.hs:51-52
        case R_INNERBOUND: return inside_g<D - 1>(g, ldu4(S.recs, rec.y), p) || inside_g<D - 1>(g, ldu4(S.recs, rec.z), p);
        case R_BIH: {  // inside_bih, Bih.hs:550-565: strict box test, then both sides may be descended
          F4 h0 = ld4(S.bihhdr, 3 * rec.y), h1 = ld4(S.bihhdr, 3 * rec.y + 1);
          if (!(p.x > h0.x && p.x < h1.x && p.y > h0.y && p.y < h1.y && p.z > h0.z && p.z < h1.z)) return false;
          uint32_t st[kGenericStack];
          int sp = 0;
          uint32_t ref = as_u(h0.w);
          for (;;) {
            bool popit = true;
            if (ref & BREF_LEAF) {
              uint32_t count = (ref >> 26) & 7u, first = ref & BREF_FIRST;
              if (count == 7u) { F4 n = ld4(S.bihnodes, first); count = as_u(n.z); first = as_u(n.w); }
              for (uint32_t k = 0; k < count; k++) if (inside_g<D - 1>(g, ldu4(S.recs, first + k), p)) return true;
            } else {
              F4 n = ld4(S.bihnodes, ref);
              uint32_t w0 = as_u(n.z), w1 = as_u(n.w), axis = w0 & 3u;
              float o = vcomp(p, axis);
              bool gl = o < n.x, gr = o > n.y;
              if (gl) { if (gr && sp < kGenericStack) st[sp++] = w1; ref = w0 >> 2; popit = false; }
              else if (gr) { ref = w1; popit = false; }
            }
            if (popit) { if (sp == 0) return false; ref = st[--sp]; }
          }
        }
        default: return false;  // Mesh (Mesh.hs:211), Void
      }
    } else {
      if (kind != R_VOID) g.err = 1;
      return false;
    }
  }

  // ------------------------------------------------------------------ get_metainfo (textures only)
  static GDN TexStack meta(Ctx& g, U4 rec, V3 p) {
    const DScene& S = g.S;
    TexStack pre = 0;  // Tex records passed on the way down: tex : texs (Tex.hs:73-74), outermost first
    while ((rec.x & RF_KINDMASK) == R_TEX) { pre = tex_cat(pre, (TexStack)(rec.z + 1)); rec = ldu4(S.recs, rec.y); }
    uint32_t kind = rec.x & RF_KINDMASK;
    if (kind >= R_SPHERE && kind <= R_CONE) return tex_cat(pre, own_stack_meta(rec.z));  // primitives: ([],[]) plus folded Tex wrappers
    if constexpr (COMPOSITES) {
      TexStack res = 0;
      switch (kind) {
        case R_LIST:  // Solid.hs:337-339: later containing items are prepended
          for (uint32_t k = 0; k < rec.z; k++) {
            U4 c = ldu4(S.recs, rec.y + k);
            if (inside_g<D - 1>(g, c, p)) res = tex_cat(meta_g<D - 1>(g, c, p), res);
          }
          break;
        case R_INSTANCE: { Xf6 x = load_xf(S, rec.z); res = meta_g<D - 1>(g, ldu4(S.recs, rec.y), mat_point(x.i0, x.i1, x.i2, p)); break; }  // Solid.hs:517-519
        case R_DIFF: {  // Csg.hs:103-106
          U4 a = ldu4(S.recs, rec.y);
          if (inside_g<D - 1>(g, a, p) && !inside_g<D - 1>(g, ldu4(S.recs, rec.z), p)) res = meta_g<D - 1>(g, a, p);
          break;
        }
        case R_ISECT: {  // Csg.hs:108-111
          bool all = true;
          for (uint32_t k = 0; k < rec.z; k++) all = all && inside_g<D - 1>(g, ldu4(S.recs, rec.y + k), p);
          if (all) for (uint32_t k = 0; k < rec.z; k++) res = tex_cat(res, meta_g<D - 1>(g, ldu4(S.recs, rec.y + k), p));
          break;
        }
        case R_BOUND: if (inside_g<D - 1>(g, ldu4(S.recs, rec.y), p)) res = meta_g<D - 1>(g, ldu4(S.recs, rec.z), p); break;  // Bound.hs:54-58
        case R_INNERBOUND: res = meta_g<D - 1>(g, ldu4(S.recs, rec.z), p); break;
        case R_BIH: {  // get_metainfo_bih, Bih.hs:567-585: left result ++ right result, leaves like lists
          F4 h0 = ld4(S.bihhdr, 3 * rec.y), h1 = ld4(S.bihhdr, 3 * rec.y + 1);
          if (!(p.x > h0.x && p.x < h1.x && p.y > h0.y && p.y < h1.y && p.z > h0.z && p.z < h1.z)) break;
          uint32_t st[kGenericStack];
          int sp = 0;
          uint32_t ref = as_u(h0.w);
          for (;;) {
            bool popit = true;
            if (ref & BREF_LEAF) {
              uint32_t count = (ref >> 26) & 7u, first = ref & BREF_FIRST;
              if (count == 7u) { F4 n = ld4(S.bihnodes, first); count = as_u(n.z); first = as_u(n.w); }
              TexStack leaf = 0;
              for (uint32_t k = 0; k < count; k++) {
                U4 c = ldu4(S.recs, first + k);
                if (inside_g<D - 1>(g, c, p)) leaf = tex_cat(meta_g<D - 1>(g, c, p), leaf);
              }
              res = tex_cat(res, leaf);
            } else {
              F4 n = ld4(S.bihnodes, ref);
              uint32_t w0 = as_u(n.z), w1 = as_u(n.w), axis = w0 & 3u;
              float o = vcomp(p, axis);
              bool gl = o < n.x, gr = o > n.y;
              if (gl) { if (gr && sp < kGenericStack) st[sp++] = w1; ref = w0 >> 2; popit = false; }
              else if (gr) { ref = w1; popit = false; }
            }
            if (popit) { if (sp == 0) break; ref = st[--sp]; }
          }
          break;
        }
        default: break;  // Mesh: the class default ([],[])
      }
      return tex_cat(pre, res);
    } else {
      if (kind != R_VOID) g.err = 1;
      return pre;
    }
  }
};

}  // namespace glome
