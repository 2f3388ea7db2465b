// rt_generic.hpp -- the generic tier: glome's full `Solid` class (rayint / shadow / inside / get_metainfo,
// Solid.hs:138-254) over the flattened record table, for scenes the flat tier cannot express: Instance,
// CSG Difference / Intersection, Bound / InnerBound, BIHs whose items are composites, and any nesting of
// those.  The reference recurses through type-class dictionaries; here the recursion is unrolled at compile
// time -- f<D> only ever calls f<D-1> -- so the call graph is static, the stacks are fixed-size scratch
// arrays, and the host validates the nesting depth at commit (flatten.hpp).  Loops that the reference
// writes as self-recursion on one node (CSG ray advancing, Csg.hs / Solid.hs:85-91) are iterative here.
#pragma once
#include "rt_device.hpp"

namespace glome {

// The interpreter's variable-length scratch lives in ONE pool per ray instead of one worst-case array per nesting level:
// nested invocations stack their frames / advances on the same arrays (fr_top, adv_top), so the kernel's scratch frame
// is sized by what a ray can have live at once, not by depth x worst case.  The pools are bounded (kIsectFrames,
// kCsgMaxAdvance per ray); running out raises the context's error flag like every other device limit.
struct GPool {
  IFrame fr[kIsectFrames];
  float adv[kCsgMaxAdvance];
};
template <bool COUNT> struct GCtx {
  const DScene& S;
  Cnt& cnt;
  unsigned int err;
  GPool& pool;
  int fr_top = 0, adv_top = 0;
  // the ray of this call is not unit length (Refract's transmitted ray, Shader.hs:141): BIHs are then walked exactly as the
  // reference walks them (no ordered early-out) -- rayint_sphere's formula (Sphere.hs:20-41) reports hits for such a ray that
  // lie outside the sphere's box, so `nearer than the best so far` no longer follows from a node's interval
  bool exact_bih = false;
};

template <int D, bool C> struct G;  // the four class methods at nesting budget D

// The class-method calls.  A child that is a primitive (under any Tex wrappers) is answered here, inline at the call
// site: most calls of a CSG scene are such leaves, and an out-of-line call costs a frame of spills each.
template <int D, bool C> GD HitG rayint_g(GCtx<C>& g, U4 rec, const Ray& r, float d, TexStack tex) {
  for (;;) {  // Tex s tex: rayint s r d (tex:texs) tags, Tex.hs:66
    if (rec.x & RF_NOVIS) return hit_miss();
    if ((rec.x & RF_KINDMASK) != R_TEX) break;
    tex = tex_push(tex, rec.z);
    rec = ldu4(g.S.recs, rec.y);
  }
  const uint32_t kind = rec.x & RF_KINDMASK;
  if (kind >= R_SPHERE && kind <= R_CONE) {
    HitG h = hit_miss();
    if (C) g.cnt.prim++;
    float t; V3 n;
    if (!prim_test<true>(g.S, kind, rec.y, r, d, t, n)) return h;
    h.hit = true; h.t = t; h.n = n; h.p = vscaleadd(r.o, r.d, t); h.lo = r.o; h.ld = r.d;
    h.tex = tex_cat(own_stack_rayint(rec.z), tex); h.uid = rec.w;
    return h;
  }
  return G<D, C>::rayint(g, rec, r, d, tex);
}
template <int D, bool C> GD bool shadow_g(GCtx<C>& g, U4 rec, const Ray& r, float d) {
  for (;;) {  // shadow (Tex s _) = shadow s; NoShadow -> False (Tex.hs:69, 81)
    if (rec.x & RF_NOSHADOW) return false;
    if ((rec.x & RF_KINDMASK) != R_TEX) break;
    rec = ldu4(g.S.recs, rec.y);
  }
  const uint32_t kind = rec.x & RF_KINDMASK;
  if (kind >= R_SPHERE && kind <= R_CONE) { if (C) g.cnt.prim++; return prim_shadow(g.S, kind, rec.y, r, d); }
  return G<D, C>::shadow(g, rec, r, d);
}
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

  // ------------------------------------------------------------------ rayint
  static GDN HitG rayint(Ctx& g, U4 rec, const Ray& r, float d, TexStack tex) {
    const DScene& S = g.S;
    for (;;) {  // Tex s tex: rayint s r d (tex:texs) tags, Tex.hs:66
      if (rec.x & RF_NOVIS) return hit_miss();
      if ((rec.x & RF_KINDMASK) != R_TEX) break;
      tex = tex_push(tex, rec.z);
      rec = ldu4(S.recs, rec.y);
    }
    uint32_t kind = rec.x & RF_KINDMASK;
    if (kind >= R_SPHERE && kind <= R_CONE) {
      HitG h = hit_miss();
      if (C) g.cnt.prim++;
      float t; V3 n;
      if (!prim_test<true>(S, kind, rec.y, r, d, t, n)) return h;
      h.hit = true; h.t = t; h.n = n; h.p = vscaleadd(r.o, r.d, t); h.lo = r.o; h.ld = r.d;
      h.tex = tex_cat(own_stack_rayint(rec.z), tex); h.uid = rec.w;
      return h;
    }
    if constexpr (COMPOSITES) {
      switch (kind) {
        case R_LIST: {  // foldl' nearest RayMiss, every item with the same d (Solid.hs:327, Q9)
          HitG best = hit_miss();
          for (uint32_t k = 0; k < rec.z; k++) best = nearest_hit(best, rayint_g<D - 1>(g, ldu4(S.recs, rec.y + k), r, d, tex));
          return best;
        }
        case R_INSTANCE: {  // rayint_instance, Solid.hs:388-403 (Q8)
          Xf6 x = load_xf(S, rec.z);
          V3 newdir = mat_vec(x.i0, x.i1, x.i2, r.d), neworig = mat_point(x.i0, x.i1, x.i2, r.o);
          float lenscale = sqrtf(vdot(newdir, newdir)), invlenscale = 1.0f / lenscale;
          Ray lr; lr.o = neworig; lr.d = newdir * invlenscale;
          const bool exact_outside = g.exact_bih;
          g.exact_bih = false;  // (the local ray is unit length)
          HitG h = rayint_g<D - 1>(g, ldu4(S.recs, rec.y), lr, d * lenscale, tex);
          g.exact_bih = exact_outside;
          if (!h.hit) return h;
          h.t = h.t * invlenscale;
          h.p = mat_point(x.f0, x.f1, x.f2, h.p);
          h.n = vnorm(mat_tvec(x.i0, x.i1, x.i2, h.n));
          return h;
        }
        case R_DIFF: return diff_rayint(g, rec, r, d, tex);
        case R_ISECT: return isect_rayint(g, rec, r, d, tex);
        case R_BOUND: {  // rayint_bound, Bound.hs:30-35
          U4 sa = ldu4(S.recs, rec.y);
          if (inside_g<D - 1>(g, sa, r.o) || shadow_g<D - 1>(g, sa, r, d)) return rayint_g<D - 1>(g, ldu4(S.recs, rec.z), r, d, tex);
          return hit_miss();
        }
        case R_INNERBOUND: {  // rayint_innerbound, Bound.hs:97-99
          HitG ha = rayint_g<D - 1>(g, ldu4(S.recs, rec.y), r, d, (TexStack)0);
          return rayint_g<D - 1>(g, ldu4(S.recs, rec.z), r, ha.hit ? ha.t : kInf, tex);
        }
        case R_BIH: return bih_rayint(g, rec, r, d, tex);
        case R_MESH: return mesh_rayint(g, rec, r, d, tex);
        default: return hit_miss();
      }
    } else {
      if (kind != R_VOID) g.err = 1;  // a composite below the instantiated nesting budget (commit validates this)
      return hit_miss();
    }
  }

  // rayint_difference, Csg.hs:33-54 (Q13); the self-recursion through rayint_advance (Solid.hs:85-91) is a loop
  static GD HitG diff_rayint(Ctx& g, U4 rec, const Ray& r0, float d0, TexStack tex) {
    const DScene& S = g.S;
    U4 ra = ldu4(S.recs, rec.y), rb = ldu4(S.recs, rec.z);
    float* adds = g.pool.adv + g.adv_top;  // this invocation's advances; nested invocations stack above them
    const int adv_base = g.adv_top;
    int na = 0;
    Ray r = r0;
    float d = d0;
    HitG res = hit_miss();
    for (;;) {
      float adv;
      if (inside_g<D - 1>(g, rb, r.o)) {
        HitG hb = rayint_g<D - 1>(g, rb, r, d, tex);
        if (!hb.hit) break;
        if (inside_g<D - 1>(g, ra, hb.p) && !inside_g<D - 1>(g, rb, vscaleadd(hb.p, r.d, kDel))) {
          hb.n = vneg(hb.n);
          hb.tex = meta_g<D - 1>(g, ra, hb.p);  // `difference` = Difference a b True: textures of A at the carved point
          res = hb;
          break;
        }
        adv = hb.t;
      } else {
        HitG ha = rayint_g<D - 1>(g, ra, r, d, tex);
        if (!ha.hit) break;
        HitG hb = rayint_g<D - 1>(g, rb, r, d, tex);
        if (!hb.hit) { res = ha; break; }
        if (ha.t < hb.t) { res = ha; break; }
        adv = hb.t;
      }
      if (adv_base + na >= kCsgMaxAdvance) { g.err = 1; break; }
      float a = adv + kDel;
      adds[na++] = a;
      g.adv_top = adv_base + na;
      r.o = vscaleadd(r.o, r.d, a);  // ray_move
      d = d - a;
    }
    if (res.hit) for (int k = na - 1; k >= 0; k--) res.t = res.t + adds[k];  // RayHit (depth+a) ..., innermost first
    g.adv_top = adv_base;
    return res;
  }

  // rayint_intersection, Csg.hs:68-90 (Q14).  The reference recurses on the list tail (non-tail position) and on
  // itself with an advanced ray; both become explicit frames.
  static constexpr uint32_t kFrState1 = 1u << 30, kFrState2 = 2u << 30, kFrFrom = (1u << 30) - 1u;
  static GD bool inside_rest(Ctx& g, U4 rec, uint32_t from, V3 p) {  // inside (Intersection ss) sp: foldl' (&&) True
    bool acc = true;
    for (uint32_t k = from; k < rec.z; k++) acc = acc && inside_g<D - 1>(g, ldu4(g.S.recs, rec.y + k), p);
    return acc;
  }
  static GD HitG isect_rayint(Ctx& g, U4 rec, const Ray& r0, float d0, TexStack tex) {
    const DScene& S = g.S;
    uint32_t n = rec.z;
    const int base = g.fr_top;  // this invocation's frames start here; nested invocations stack above fr[sp]
    IFrame* fr = g.pool.fr + base;
    const int room = kIsectFrames - base;
    if (room < 1) { g.err = 1; return hit_miss(); }
    int sp = 0;
    auto push = [&](uint32_t from, V3 o, float d) {
      IFrame& c = fr[sp];
      c.from = from; c.ox = o.x; c.oy = o.y; c.oz = o.z; c.d = d; c.aux = 0;
      g.fr_top = base + sp + 1;
    };
    push(0, r0.o, d0);
    HitG ret = hit_miss();
    bool returning = false;  // true: frame fr[sp] has completed with `ret`
    for (;;) {
      if (!returning) {
        IFrame& f = fr[sp];
        const uint32_t from = f.from & kFrFrom;
        Ray r; r.o = v3(f.ox, f.oy, f.oz); r.d = r0.d;
        if (from >= n || f.d < 0) { ret = hit_miss(); returning = true; continue; }  // null slds || d < 0
        U4 s = ldu4(S.recs, rec.y + from);
        HitG hs = rayint_g<D - 1>(g, s, r, f.d, tex);
        if (from + 1 == n) { ret = hs; returning = true; continue; }  // [] -> rayint s r d t tags
        if (inside_g<D - 1>(g, s, r.o)) {
          if (!hs.hit) { f.from = from + 1; continue; }  // RayMiss -> rayint (Intersection ss) r d: a tail call
          if (sp + 1 >= room) { g.err = 1; g.fr_top = base; return hit_miss(); }
          f.aux = hs.t; f.from = from | kFrState1;  // rest = rayint (Intersection ss) r sd
          sp++; push(from + 1, r.o, hs.t);
          continue;
        }
        if (!hs.hit) { ret = hit_miss(); returning = true; continue; }
        if (inside_rest(g, rec, from + 1, hs.p)) { ret = hs; returning = true; continue; }  // RayHit sd sp sn r vzero st stags
        if (sp + 1 >= room) { g.err = 1; g.fr_top = base; return hit_miss(); }
        float a = hs.t + kDel;  // rayint_advance (Intersection slds) r d t tags sd
        f.from = from | kFrState2; f.aux = a;
        sp++; push(from, vscaleadd(r.o, r.d, a), f.d - a);
        continue;
      }
      if (sp == 0) { g.fr_top = base; return ret; }
      sp--;
      g.fr_top = base + sp + 1;
      IFrame& p = fr[sp];
      if ((p.from & ~kFrFrom) == kFrState1) {
        if (ret.hit) continue;  // hit -> hit
        if (sp + 1 >= room) { g.err = 1; g.fr_top = base; return hit_miss(); }
        float a = p.aux + kDel;
        const uint32_t pf = p.from & kFrFrom;
        p.from = pf | kFrState2; p.aux = a;
        V3 po = v3(p.ox, p.oy, p.oz);
        float pd = p.d;
        sp++; push(pf, vscaleadd(po, r0.d, a), pd - a);
        returning = false;
        continue;
      }
      if (ret.hit) ret.t = ret.t + p.aux;  // state 2: RayHit (depth+a) ...
    }
  }

  // rayint_bih, Bih.hs:332-368, over records (any leaf class); ordered early-out (see rt_device.hpp)
  static GD HitG bih_rayint(Ctx& g, U4 rec, const Ray& r, float d, TexStack tex) {
    const DScene& S = g.S;
    PrivStack stk;
    HitG best = hit_miss();
    if (g.exact_bih) {  // a ray that is not unit length: the reference's own visits, every item with tmax = far (Bih.hs:332-368)
      bih_traverse<0, C>(S, rec.y, r, d, stk, kGenericStack, g.cnt,
        [&](uint32_t frec, uint32_t, uint32_t count, float tmax) {
          for (uint32_t k = 0; k < count; k++) best = nearest_hit(best, rayint_g<D - 1>(g, ldu4(S.recs, frec + k), r, tmax, tex));
          return false;
        },
        [&]() { return kInf * 4.0f; });
      return best;
    }
    // ordered early-out: the best hit so far decides which nodes are still worth entering, but every item is tested with its
    // node's own `far` (`rayint s r far`, Bih.hs:339) -- an item may be a cylinder or a cone, whose answer depends on tmax
    // beyond the hit (rt_device.hpp bih_traverse, CLAMP)
    bih_traverse<1, C, false>(S, rec.y, r, d, stk, kGenericStack, g.cnt,
      [&](uint32_t frec, uint32_t, uint32_t count, float tmax) {
        for (uint32_t k = 0; k < count; k++) {
          const U4 it = ldu4(S.recs, frec + k);
          // a plain primitive other than a quadric (under any Tex wrappers) answers the same for every tmax beyond its hit:
          // it may be tested against the best so far (the lattice of GlomeView's default scene is 9261 such spheres)
          const uint32_t ik = skip_tex(S, it).x & RF_KINDMASK;
          const bool clampable = best.hit && ik >= R_SPHERE && ik <= R_CONE && ik != R_CYL && ik != R_CONE;
          best = nearest_hit(best, rayint_g<D - 1>(g, it, r, clampable ? gminf(tmax, best.t) : tmax, tex));
        }
        return false;
      },
      [&]() { return best.hit ? best.t : kInf * 4.0f; });
    return best;
  }
  // rayint_mesh, Mesh.hs:136-198
  static GD HitG mesh_rayint(Ctx& g, U4 rec, const Ray& r, float d, TexStack tex) {
    const DScene& S = g.S;
    PrivStack stk;
    float mt; uint32_t ti;
    mesh_closest<C>(S, rec.y, r, d, stk, kGenericStack, g.cnt, mt, ti);
    HitG h = hit_miss();
    if (ti == 0xffffffffu) return h;
    h.hit = true; h.t = mt; h.p = vscaleadd(r.o, r.d, mt); h.uid = rec.w; h.lo = r.o; h.ld = r.d;
    U4 meta = ldu4(S.mtrimeta, ti);
    F4 q0 = ld4(S.mtris, 3 * ti), q1 = ld4(S.mtris, 3 * ti + 1), q2 = ld4(S.mtris, 3 * ti + 2);
    if (meta.x == 0) h.n = v3(q0.w, q1.w, q2.w);
    else {
      float t, b1, b2;
      tri_test(q0, q1, q2, r, kInf * 8.0f, t, b1, b2);
      uint32_t nb = meta.x - 1;
      V3 n1 = v3(ld4(S.trinorms, nb)), n2 = v3(ld4(S.trinorms, nb + 1)), n3 = v3(ld4(S.trinorms, nb + 2));
      V3 a1 = n1 * (1 - (b1 + b2)), a2 = n2 * b1, a3 = n3 * b2;
      h.n = vnorm(v3(a1.x + a2.x + a3.x, a1.y + a2.y + a3.y, a1.z + a2.z + a3.z));
    }
    h.tex = meta.y ? tex_cat((TexStack)meta.y, tex) : tex;
    return h;
  }

  // ------------------------------------------------------------------ shadow
  static GDN bool shadow(Ctx& g, U4 rec, const Ray& r, float d) {
    const DScene& S = g.S;
    for (;;) {  // shadow (Tex s _) = shadow s; NoShadow -> False (Tex.hs:69, 81)
      if (rec.x & RF_NOSHADOW) return false;
      if ((rec.x & RF_KINDMASK) != R_TEX) break;
      rec = ldu4(S.recs, rec.y);
    }
    uint32_t kind = rec.x & RF_KINDMASK;
    if (kind >= R_SPHERE && kind <= R_CONE) { if (C) g.cnt.prim++; return prim_shadow(S, kind, rec.y, r, d); }
    if constexpr (COMPOSITES) {
      switch (kind) {
        case R_LIST:  // foldl' (||) False (Solid.hs:330)
          for (uint32_t k = 0; k < rec.z; k++) if (shadow_g<D - 1>(g, ldu4(S.recs, rec.y + k), r, d)) return true;
          return false;
        case R_INSTANCE: {  // shadow_instance, Solid.hs:464-471
          Xf6 x = load_xf(S, rec.z);
          V3 newdir = mat_vec(x.i0, x.i1, x.i2, r.d), neworig = mat_point(x.i0, x.i1, x.i2, r.o);
          float lenscale = sqrtf(vdot(newdir, newdir)), invlenscale = 1.0f / lenscale;
          Ray lr; lr.o = neworig; lr.d = newdir * invlenscale;
          return shadow_g<D - 1>(g, ldu4(S.recs, rec.y), lr, d * lenscale);
        }
        // Difference / Intersection have no shadow method: the class default runs rayint (Solid.hs:218-221, Q15).
        // The default sees the node itself, so an OnlyShadow flag on it does not hide it here.
        case R_DIFF: { U4 v = rec; v.x &= ~RF_NOVIS; return diff_rayint(g, v, r, d, (TexStack)0).hit; }
        case R_ISECT: { U4 v = rec; v.x &= ~RF_NOVIS; return isect_rayint(g, v, r, d, (TexStack)0).hit; }
        case R_BOUND: {  // shadow_bound, Bound.hs:44-49
          U4 sa = ldu4(S.recs, rec.y);
          if (inside_g<D - 1>(g, sa, r.o) || shadow_g<D - 1>(g, sa, r, d)) return shadow_g<D - 1>(g, ldu4(S.recs, rec.z), r, d);
          return false;
        }
        case R_INNERBOUND: return shadow_g<D - 1>(g, ldu4(S.recs, rec.y), r, d) || shadow_g<D - 1>(g, ldu4(S.recs, rec.z), r, d);  // Bound.hs:101-103
        case R_BIH: return bih_shadow(g, rec, r, d);
        default: return false;  // Mesh: `shadow s r d = False` (Mesh.hs:210); Void
      }
    } else {
      if (kind != R_VOID) g.err = 1;
      return false;
    }
  }
  static GD bool bih_shadow(Ctx& g, U4 rec, const Ray& r, float d) {  // shadow_bih, Bih.hs:510-544
    const DScene& S = g.S;
    PrivStack stk;
    bool occ = false;
    bih_traverse<2, C>(S, rec.y, r, d, stk, kGenericStack, g.cnt,
      [&](uint32_t frec, uint32_t, uint32_t count, float tmax) {
        float dd = gminf(d, tmax);
        for (uint32_t k = 0; k < count; k++) if (shadow_g<D - 1>(g, ldu4(S.recs, frec + k), r, dd)) { occ = true; return true; }
        return false;
      },
      [&]() { return 0.0f; });
    return occ;
  }

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
