// rt_generic.hpp -- the generic tier: glome's full `Solid` class (rayint / shadow / inside / get_metainfo,
// Solid.hs:138-254) over the flattened record table, for scenes the flat tier cannot express: Instance, CSG Difference /
// Intersection, Bound / InnerBound, BIHs whose items are composites, and any nesting of those.  The reference recurses
// through type-class dictionaries; here every method is a loop over explicit frames in one word stack per ray -- no
// recursion, no nesting budget but the stack's size.  rayint and shadow are the CALL / RETURN steps of one state machine
// (vm_run); inside and get_metainfo, pure functions of a point asked by the CSG nodes and Bound in the middle of a rayint
// step, are run-to-completion loops over the same stack (vm_inside, vm_meta).  A wave's lanes are usually at different
// depths of different composites; a loop whose body is "one step of whatever this lane is doing" lets all the lanes that
// are testing a primitive, or stepping through a BIH, or returning to a list, do so together whatever their depth -- round
// 1 instantiated every method once per nesting level (f<D> calls f<D-1>), and lanes at different depths executed different
// copies of the same code one after the other.
//
// Conventions between caller and callee (what a frame must keep to go on after a call):
//   r    the callee restores it: only Instance (local ray) and the CSG nodes (advanced origin) change it, and they put it back
//   d    the caller keeps it (a BIH hands its items the node's `far`, not its own d)
//   tex  the caller keeps it (Tex wrappers push on it on the way down)
// Results: `rh` (rayint) or `rb` (shadow).
#pragma once
#include "rt_device.hpp"

namespace glome {

constexpr int kHitWords = kVmHitWords;  // (kVmWords frame words per ray, scratch: rt_types.h; running out raises the context's error flag)
#ifndef GLOME_VM_BIH_STEPS
#define GLOME_VM_BIH_STEPS 32
#endif
#ifndef GLOME_PK_MIN_LANES
#define GLOME_PK_MIN_LANES 64
#endif
constexpr int kPkMinLanes = GLOME_PK_MIN_LANES;  // lanes that must wait for a packet walk before one is made while other lanes are still on their way
constexpr int kBihStepsPerPass = GLOME_VM_BIH_STEPS;  // BIH steps a lane may take in one pass of the loop

enum : uint32_t {
  VT_DONE = 0, VT_LIST_R, VT_LIST_S, VT_INST_R, VT_INST_S, VT_BOUND_R, VT_BOUND_S, VT_IB_R, VT_IB_S,
  VT_DIFF_B, VT_DIFF_A, VT_DIFF_AB, VT_ISECT_HS, VT_ISECT_S1, VT_ISECT_S2, VT_BIH_R, VT_BIH_S, VT_S_OF_R
};
enum : int { ST_CALL_R = 0, ST_CALL_S, ST_RET, ST_BIH, ST_BIH_ITEM, ST_DIFF, ST_ISECT, ST_ENTER_CSG, ST_LIST_R, ST_LIST_S, ST_PK_R, ST_PK_S };

GD void vm_st_hit(uint32_t* m, int i, const HitG& h) {
  m[i] = h.hit ? 1u : 0u; m[i + 1] = as_u(h.t);
  m[i + 2] = as_u(h.p.x); m[i + 3] = as_u(h.p.y); m[i + 4] = as_u(h.p.z);
  m[i + 5] = as_u(h.n.x); m[i + 6] = as_u(h.n.y); m[i + 7] = as_u(h.n.z);
  m[i + 8] = (uint32_t)h.tex; m[i + 9] = (uint32_t)(h.tex >> 32); m[i + 10] = h.uid;
  m[i + 11] = as_u(h.lo.x); m[i + 12] = as_u(h.lo.y); m[i + 13] = as_u(h.lo.z);
  m[i + 14] = as_u(h.ld.x); m[i + 15] = as_u(h.ld.y); m[i + 16] = as_u(h.ld.z);
}
GD HitG vm_ld_hit(const uint32_t* m, int i) {
  HitG h;
  h.hit = m[i] != 0; h.t = as_f(m[i + 1]);
  h.p = v3(as_f(m[i + 2]), as_f(m[i + 3]), as_f(m[i + 4]));
  h.n = v3(as_f(m[i + 5]), as_f(m[i + 6]), as_f(m[i + 7]));
  h.tex = (TexStack)m[i + 8] | ((TexStack)m[i + 9] << 32); h.uid = m[i + 10];
  h.lo = v3(as_f(m[i + 11]), as_f(m[i + 12]), as_f(m[i + 13]));
  h.ld = v3(as_f(m[i + 14]), as_f(m[i + 15]), as_f(m[i + 16]));
  return h;
}

// rayint_mesh, Mesh.hs:136-198: a leaf of the interpreter (no calls below it)
template <bool C> GD HitG vm_mesh_rayint(const DScene& S, Cnt& cnt, U4 rec, const Ray& r, float d, TexStack tex) {
  PrivStack stk;
  float mt; uint32_t ti;
  mesh_closest<C>(S, rec.y, r, d, stk, kGenericStack, cnt, mt, ti);
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
  h.tex = meta.y ? tex_cat((TexStack)meta.y, tex, (int)S.tex_bits) : tex;
  return h;
}

// A call whose callee is a primitive under Tex wrappers is answered in place (no frame, no pass through the loop): the
// children of most lists and the items of most BIH leaves are such.  vm_resolve_r / _s strip the wrappers of a rayint /
// shadow call: 0 = a primitive (rec is its record, tex has the wrappers' textures), 1 = answered by a flag (OnlyShadow
// misses, NoShadow casts none: Tex.hs:81-89), 2 = a composite (rec and tex as they were: the call goes through the loop).
GD int vm_resolve_r(const DScene& S, U4& rec, TexStack& tex) {
  U4 c = rec; TexStack t = tex;
  for (;;) {
    if (c.x & RF_NOVIS) return 1;
    if ((c.x & RF_KINDMASK) != R_TEX) break;
    t = tex_push(t, c.z, (int)S.tex_bits);
    c = ldu4(S.recs, c.y);
  }
  const uint32_t kind = c.x & RF_KINDMASK;
  if (kind >= R_SPHERE && kind <= R_CONE) { rec = c; tex = t; return 0; }
  return 2;
}
GD int vm_resolve_s(const DScene& S, U4& rec) {
  U4 c = rec;
  for (;;) {
    if (c.x & RF_NOSHADOW) return 1;
    if ((c.x & RF_KINDMASK) != R_TEX) break;
    c = ldu4(S.recs, c.y);
  }
  const uint32_t kind = c.x & RF_KINDMASK;
  if (kind >= R_SPHERE && kind <= R_CONE) { rec = c; return 0; }
  return 2;
}
template <bool C> GD HitG vm_prim_hit(const DScene& S, Cnt& cnt, const U4& rec, const Ray& r, float d, TexStack tex) {
  HitG h = hit_miss();
  if (C) cnt.prim++;
  float t; V3 n;
  if (prim_test<true>(S, rec.x & RF_KINDMASK, rec.y, r, d, t, n)) {
    h.hit = true; h.t = t; h.n = n; h.p = vscaleadd(r.o, r.d, t); h.lo = r.o; h.ld = r.d;
    h.tex = tex_cat(own_stack_rayint(rec.z, (int)S.tex_bits), tex, (int)S.tex_bits); h.uid = rec.w;
  }
  return h;
}

// A callee that is an Instance of a primitive (under Tex wrappers on either side of it) is answered in place too: rayint_instance /
// shadow_instance (Solid.hs:388-403, 464-471) around the primitive's test -- the same arithmetic in the same order as the
// VT_INST_R / VT_INST_S frames, without the frame and the passes through the loop.  The oak of GlomeView's default scene is a bih
// of 2,047 such items.  false: the callee is something else (nothing was computed).
template <bool C> GD bool vm_inst_prim_hit(const DScene& S, Cnt& cnt, U4 rec, TexStack tex, const Ray& ray, float d, HitG& h) {
  for (;;) {
    if (rec.x & RF_NOVIS) { h = hit_miss(); return true; }
    if ((rec.x & RF_KINDMASK) != R_TEX) break;
    tex = tex_push(tex, rec.z, (int)S.tex_bits);
    rec = ldu4(S.recs, rec.y);
  }
  if ((rec.x & RF_KINDMASK) != R_INSTANCE) return false;
  U4 c = ldu4(S.recs, rec.y);
  bool novis = false;
  for (;;) {
    if (c.x & RF_NOVIS) { novis = true; break; }
    if ((c.x & RF_KINDMASK) != R_TEX) break;
    tex = tex_push(tex, c.z, (int)S.tex_bits);
    c = ldu4(S.recs, c.y);
  }
  const uint32_t kind = c.x & RF_KINDMASK;
  const bool primlist = kind == R_LIST && (c.x & RF_PRIMLIST);  // a group of primitives (the chessboard of GlomeView's default scene: 64 boxes)
  if (!novis && !primlist && !(kind >= R_SPHERE && kind <= R_CONE)) return false;
  h = hit_miss();
  if (novis) return true;
  const Xf6 x = load_xf(S, rec.z);
  const V3 newdir = mat_vec(x.i0, x.i1, x.i2, ray.d), neworig = mat_point(x.i0, x.i1, x.i2, ray.o);
  const float lenscale = sqrtf(vdot(newdir, newdir)), invlenscale = 1.0f / lenscale;
  Ray r; r.o = neworig; r.d = newdir * invlenscale;
  if (primlist) {  // foldl' nearest RayMiss, every item with the same d (Solid.hs:327, Q9: ties -> the later item)
    const float dl = d * lenscale;
    for (uint32_t k = 0; k < c.z; k++) {
      const U4 cc = ldu4(S.recs, c.y + k);
      if (cc.x & RF_NOVIS) continue;
      const HitG hk = vm_prim_hit<C>(S, cnt, cc, r, dl, tex);
      if (hk.hit && (!h.hit || !(h.t < hk.t))) h = hk;
    }
  } else h = vm_prim_hit<C>(S, cnt, c, r, d * lenscale, tex);
  if (h.hit) {
    h.t = h.t * invlenscale;
    h.p = mat_point(x.f0, x.f1, x.f2, h.p);
    h.n = vnorm(mat_tvec(x.i0, x.i1, x.i2, h.n));
  }
  return true;
}
// 0 = not an Instance of a primitive, 1 = it casts no shadow on this ray, 2 = it does
template <bool C> GD int vm_inst_prim_shadow(const DScene& S, Cnt& cnt, U4 rec, const Ray& ray, float d) {
  for (;;) {
    if (rec.x & RF_NOSHADOW) return 1;
    if ((rec.x & RF_KINDMASK) != R_TEX) break;
    rec = ldu4(S.recs, rec.y);
  }
  if ((rec.x & RF_KINDMASK) != R_INSTANCE) return 0;
  U4 c = ldu4(S.recs, rec.y);
  for (;;) {
    if (c.x & RF_NOSHADOW) return 1;
    if ((c.x & RF_KINDMASK) != R_TEX) break;
    c = ldu4(S.recs, c.y);
  }
  const uint32_t kind = c.x & RF_KINDMASK;
  const bool primlist = kind == R_LIST && (c.x & RF_PRIMLIST);
  if (!primlist && !(kind >= R_SPHERE && kind <= R_CONE)) return 0;
  const Xf6 x = load_xf(S, rec.z);
  const V3 newdir = mat_vec(x.i0, x.i1, x.i2, ray.d), neworig = mat_point(x.i0, x.i1, x.i2, ray.o);
  const float lenscale = sqrtf(vdot(newdir, newdir)), invlenscale = 1.0f / lenscale;
  Ray r; r.o = neworig; r.d = newdir * invlenscale;
  if (primlist) {  // foldl' (||) False (Solid.hs:330)
    const float dl = d * lenscale;
    for (uint32_t k = 0; k < c.z; k++) {
      const U4 cc = ldu4(S.recs, c.y + k);
      if (cc.x & RF_NOSHADOW) continue;
      if (C) cnt.prim++;
      if (prim_shadow(S, cc.x & RF_KINDMASK, cc.y, r, dl)) return 2;
    }
    return 1;
  }
  if (C) cnt.prim++;
  return prim_shadow(S, kind, c.y, r, d * lenscale) ? 2 : 1;
}

// ------------------------------------------------------------------ a BIH of items answered in place, walked by the wave as ONE packet
// The oak of GlomeView's default scene is a bih of 2,047 Instances of cones and spheres; walked lane by lane over frames (ST_BIH /
// ST_BIH_ITEM below) every lane stands on its own node and its own item, of its own kind.  Here the lanes that wait for the same tree
// (vm_run's packet service) go down it together: node and item are the wave's, only the rays differ.  The walk is the per-lane
// ordered walk's in mask form, decision for decision -- same plane arithmetic, same go / push rules, the best hit so far only decides
// which nodes and leaves a lane still enters, an item sees the node's own `far` (a quadric's answer depends on it: rt_device.hpp
// bih_traverse, CLAMP) or, a plain primitive that is not a quadric, min(far, best) exactly as ST_BIH_ITEM tests it -- so frames are
// bit-identical with the service off (GLOME_DEBUG_NO_GENERIC_PACKETS; tests).  The nearest hit so far is kept where ST_BIH_ITEM keeps it,
// in the lane's frame memory (`m[slot ..]`, kHitWords above the lane's top frame: the caller has checked the room): the hit that comes
// back is the one the item's in-place evaluation made, not a second evaluation of it.  (The first version returned WHICH item was
// nearest and finished it by an ordinary call, ST_CALL_R: on the host build the same bits, on the GPU an ulp of the depth apart on one
// fuzz scene in 680 -- the compiler contracts the Instance frame's arithmetic and the in-place copy's differently.)
// MODE 1: rayint (Bih.hs:332-368, ordered).  MODE 2: shadow_bih (Bih.hs:510-544): true for a lane with an occluder.
// All lanes of the wave call together; `valid`: the lane takes part.  The root is a branch and depth <= stk.total_cap() (callers).
struct ItemPick { float t; uint32_t item; };  // the nearest hit's distance and its item's record (CAND_NONE: none yet)
template <int MODE, bool C, class STK>
GD bool bih_items_wave(const DScene& S, Cnt& cnt, uint32_t hdr, const Ray& r, float d, TexStack tex, bool valid, STK& stk, ItemPick& pick, uint32_t* m, int slot) {
  hdr = uni(hdr);
  const F4 h0 = ld4u(S.bihhdr, 3 * hdr), h1 = ld4u(S.bihhdr, 3 * hdr + 1);
  const uint32_t root = uni(as_u(h0.w));
  const V3 rcp = v3(dir_rcp(r.d.x), dir_rcp(r.d.y), dir_rcp(r.d.z));
  float near0, far0;
  bbclip_ub(r, v3(h0), v3(h1), near0, far0);
  far0 = gminf(d, far0);  // `traverse root near (fmin d far)`, Bih.hs:368 / 515
  pick.t = kInf * 4.0f; pick.item = CAND_NONE;
  const uint32_t oct = (rcp.x > 0 ? 1u : 0u) | (rcp.y > 0 ? 2u : 0u) | (rcp.z > 0 ? 4u : 0u);
  if (C) { if (valid && near0 > far0) cnt.bih++; }  // the root taken up with an empty interval: counted and left
  LaneMask todo = wave_ballot(valid && !(near0 > far0));
  bool occ = false;
  while (todo != 0) {  // one walk per sign pattern of the directions among the lanes (near child first is then the same child for all)
    const uint32_t fwdbits = uni(first_lane_value(todo, oct));
    LaneMask am = todo & wave_ballot(oct == fwdbits);
    todo &= ~am;
    uint32_t ref = root;
    float nearv = near0, farv = far0;  // this lane's interval in the current node, as the planes cut it
    int sp = 0;
    LaneMask occm = 0;
    for (;;) {
      while (!(ref & BREF_LEAF)) {
        ref = uni(ref); am = uni(am); sp = (int)uni((uint32_t)sp);
        if (C) cnt.bih += lane_of(am) ? 1u : 0u;
        const float cf = MODE == 1 ? gminf(farv, pick.item != CAND_NONE ? pick.t : kInf * 4.0f) : farv;  // `far` cut by the best so far: decides, is not handed down
        am &= wave_ballot(!(nearv > cf));
        if (am == 0) break;
        const F4 n = ld4u(S.bihnodes, ref);
        const uint32_t w0 = uni(as_u(n.z)), right = uni(as_u(n.w));
        const uint32_t axis = w0 & 3u, left = w0 >> 2;
        float dl, dr;
        if (axis == 0) { dl = (n.x - r.o.x) * rcp.x; dr = (n.y - r.o.x) * rcp.x; }
        else if (axis == 1) { dl = (n.x - r.o.y) * rcp.y; dr = (n.y - r.o.y) * rcp.y; }
        else { dl = (n.x - r.o.z) * rcp.z; dr = (n.y - r.o.z) * rcp.z; }
        const bool fwd = (fwdbits >> axis) & 1u;
        const uint32_t c1 = fwd ? left : right, c2 = fwd ? right : left;
        const float t1 = fwd ? dl : dr, t2 = fwd ? dr : dl;
        // (an empty leaf child has its plane at -+inf, flatten.hpp: both tests fail for it)
        const LaneMask m1 = wave_ballot(nearv < t1) & am;
        const LaneMask m2 = wave_ballot(t2 < cf) & am;
        const float f1 = gminf(t1, farv), n2 = gmaxf(t2, nearv);
        if ((m1 != 0) & (m2 != 0)) { stk.push_wave(sp, c2, m2, n2, farv); sp++; }
        const bool g1 = m1 != 0;
        ref = g1 ? c1 : c2;
        am = g1 ? m1 : m2;
        farv = g1 ? f1 : farv;
        nearv = g1 ? nearv : n2;
        if (am == 0) break;
      }
      if (am != 0) {  // a leaf
        ref = uni(ref);
        uint32_t count = (ref >> 26) & 7u, first = ref & BREF_FIRST;
        if (count == 7u) { const F4 ln = ld4u(S.bihnodes, first); count = uni(as_u(ln.z)); first = uni(as_u(ln.w)); }
        const float cf = MODE == 1 ? gminf(farv, pick.item != CAND_NONE ? pick.t : kInf * 4.0f) : farv;
        bool in = lane_of(am) && !(nearv > cf);
        if (MODE == 2) {
          const float dd = gminf(d, farv);
          for (uint32_t k = 0; k < count; k++) {
            bool hit = false;
            if (in) {
              U4 it = ldu4(S.recs, first + k);
              const int what = vm_resolve_s(S, it);
              if (what == 2) hit = vm_inst_prim_shadow<C>(S, cnt, it, r, dd) == 2;
              else if (what == 0) { if (C) cnt.prim++; hit = prim_shadow(S, it.x & RF_KINDMASK, it.y, r, dd); }
            }
            const LaneMask hm = wave_ballot(hit);
            occm |= hm; am &= ~hm;
            in = in && !hit;
            if (am == 0) break;
          }
        } else {
          for (uint32_t k = 0; k < count; k++) {
            if (in) {
              U4 it = ldu4(S.recs, first + k);
              TexStack t = tex;
              const int what = vm_resolve_r(S, it, t);
              HitG h = hit_miss();
              float tm = farv;
              if (what == 2) (void)vm_inst_prim_hit<C>(S, cnt, it, tex, r, farv, h);
              else if (what == 0) {
                const uint32_t ik = it.x & RF_KINDMASK;
                if (pick.item != CAND_NONE && ik != R_CYL && ik != R_CONE) tm = gminf(farv, pick.t);
                h = vm_prim_hit<C>(S, cnt, it, r, tm, t);
              }
              if (h.hit && (pick.item == CAND_NONE || !(pick.t < h.t))) { vm_st_hit(m, slot, h); pick.t = h.t; pick.item = first + k; }  // nearest: ties -> the later item
            }
          }
        }
      }
      am = 0;
      while (sp > 0 && am == 0) {
        sp--;
        stk.pop_wave(sp, ref, am, nearv, farv);
        if (MODE == 2) am &= ~occm;
      }
      if (am == 0) break;
    }
    if (MODE == 2) occ = occ || lane_of(occm);
  }
  return occ;
}

// `inside s p` (Solid.hs:138-254) over the same word stack, from word `base` up: a run-to-completion loop (the callers are
// the CSG nodes and Bound, in the middle of a rayint step).  A composite is an OR (List, InnerBound, the leaves of a BIH), an
// AND (Intersection, Bound, Difference = a && not b) or an Instance (the point moves into its frame).
//   frames: word 0 = tag | previous frame << 8
//     IN_OR / IN_AND   1 next record, 2 records left          IN_NOT   -          IN_INST  1-3 the outer point
//     IN_THEN (a && b, a || b after a)  1 record b, 2 = 1 for ||                    IN_BIH   1 entries, then node references
enum : uint32_t { IN_DONE = 0, IN_OR, IN_AND, IN_NOT, IN_INST, IN_THEN, IN_BIH };
// (out of line: six call sites in vm_run, nine calls in ten are answered by the primitive test in front of it, vm_inside)
GDN bool vm_inside_composite(const DScene& S, unsigned int& err, uint32_t* m, int base, U4 rec, V3 p) {
  if (base + 1 > kVmWords) { err = 1; return false; }
  int sp = base + 1, fb = base;
  m[base] = IN_DONE;
  bool val = false, ret = false;
#define IN_PUSH(tag, n) { if (sp + (n) > kVmWords) { err = 1; return false; } m[sp] = (uint32_t)(tag) | ((uint32_t)fb << 8); fb = sp; sp += (n); }
#define IN_POP() { sp = fb; fb = (int)(m[fb] >> 8); }
  for (;;) {
    if (!ret) {  // evaluate `inside rec p`
      rec = skip_tex(S, rec);
      const uint32_t kind = rec.x & RF_KINDMASK;
      ret = true;
      if (kind >= R_SPHERE && kind <= R_CONE) { val = prim_inside(S, kind, rec.y, p); continue; }
      switch (kind) {
        case R_LIST:  // foldl' (||) False
          val = false;
          if (rec.z != 0) { IN_PUSH(IN_OR, 3); m[fb + 1] = rec.y; m[fb + 2] = rec.z; }
          break;
        case R_ISECT:  // foldl' (&&) True, Csg.hs:96-101
          val = true;
          if (rec.z != 0) { IN_PUSH(IN_AND, 3); m[fb + 1] = rec.y; m[fb + 2] = rec.z; }
          break;
        case R_INSTANCE: {  // Solid.hs:473-475
          Xf6 x = load_xf(S, rec.z);
          IN_PUSH(IN_INST, 4);
          m[fb + 1] = as_u(p.x); m[fb + 2] = as_u(p.y); m[fb + 3] = as_u(p.z);
          p = mat_point(x.i0, x.i1, x.i2, p);
          rec = ldu4(S.recs, rec.y); ret = false;
          break;
        }
        case R_DIFF: case R_BOUND: case R_INNERBOUND: {  // a && not b (Csg.hs:92-94), a && b (Bound.hs:51-52), a || b
          IN_PUSH(IN_THEN, 3);
          m[fb + 1] = rec.z; m[fb + 2] = kind;
          rec = ldu4(S.recs, rec.y); ret = false;
          break;
        }
        case R_BIH: {  // inside_bih, Bih.hs:550-565: strict box test, then both sides may be descended
          F4 h0 = ld4(S.bihhdr, 3 * rec.y), h1 = ld4(S.bihhdr, 3 * rec.y + 1);
          val = false;
          if (!(p.x > h0.x && p.x < h1.x && p.y > h0.y && p.y < h1.y && p.z > h0.z && p.z < h1.z)) break;
          IN_PUSH(IN_BIH, 3);
          m[fb + 1] = as_u(h0.w); m[fb + 2] = 0;  // (word 1: the reference to look at next; 0xffffffff = take an entry)
          break;
        }
        default: val = false; break;  // Mesh (Mesh.hs:211), Void
      }
      continue;
    }
    // a value has come back to the frame on top
    const uint32_t tag = m[fb] & 0xffu;
    if (tag == IN_DONE) return val;
    if (tag == IN_OR || tag == IN_AND) {
      const bool is_or = tag == IN_OR;
      uint32_t left = m[fb + 2], cur = m[fb + 1];
      bool called = false;
      while (val != is_or && left != 0) {  // (an OR goes on while false, an AND while true)
        const U4 c = skip_tex(S, ldu4(S.recs, cur)); cur++; left--;
        const uint32_t ck = c.x & RF_KINDMASK;
        if (ck >= R_SPHERE && ck <= R_CONE) { val = prim_inside(S, ck, c.y, p); continue; }
        m[fb + 1] = cur; m[fb + 2] = left; rec = c; ret = false; called = true;
        break;
      }
      if (!called) IN_POP();
      continue;
    }
    if (tag == IN_NOT) { val = !val; IN_POP(); continue; }
    if (tag == IN_INST) { p = v3(as_f(m[fb + 1]), as_f(m[fb + 2]), as_f(m[fb + 3])); IN_POP(); continue; }
    if (tag == IN_THEN) {
      const uint32_t kind = m[fb + 2], b = m[fb + 1];
      const bool go_on = kind == R_INNERBOUND ? !val : val;
      if (go_on) {
        rec = ldu4(S.recs, b); ret = false;
        if (kind == R_DIFF) m[fb] = (m[fb] & ~0xffu) | IN_NOT; else IN_POP();
      } else IN_POP();
      continue;
    }
    {  // IN_BIH: a leaf's items came back (OR frame above has been popped) or the walk goes on
      if (val) { IN_POP(); continue; }
      uint32_t ref = m[fb + 1];
      bool called = false;
      for (;;) {
        if (ref == 0xffffffffu) {
          const uint32_t ne = m[fb + 2];
          if (ne == 0) break;
          ref = m[fb + 3 + ne - 1]; m[fb + 2] = ne - 1; sp = fb + 3 + (int)ne - 1;
        }
        if (ref & BREF_LEAF) {
          uint32_t count = (ref >> 26) & 7u, first = ref & BREF_FIRST;
          if (count == 7u) { F4 n = ld4(S.bihnodes, first); count = as_u(n.z); first = as_u(n.w); }
          ref = 0xffffffffu;
          if (count != 0) {
            m[fb + 1] = ref;
            IN_PUSH(IN_OR, 3); m[fb + 1] = first; m[fb + 2] = count;
            val = false; called = true;
            break;
          }
        } else {
          F4 n = ld4(S.bihnodes, ref);
          uint32_t w0 = as_u(n.z), w1 = as_u(n.w), axis = w0 & 3u;
          float o = vcomp(p, axis);
          bool gl = o < n.x, gr = o > n.y;
          if (gl) {
            if (gr) { if (sp + 1 > kVmWords) { err = 1; return false; } m[sp++] = w1; m[fb + 2]++; }
            ref = w0 >> 2;
          } else if (gr) ref = w1;
          else ref = 0xffffffffu;
        }
      }
      if (!called) { val = false; IN_POP(); }
      continue;
    }
  }
#undef IN_PUSH
#undef IN_POP
}
GD bool vm_inside(const DScene& S, unsigned int& err, uint32_t* m, int base, U4 rec, V3 p) {
  rec = skip_tex(S, rec);
  if ((rec.x & RF_KINDMASK) >= R_SPHERE && (rec.x & RF_KINDMASK) <= R_CONE) return prim_inside(S, rec.x & RF_KINDMASK, rec.y, p);  // (most operands)
  return vm_inside_composite(S, err, m, base, rec, p);
}

// `get_metainfo s p` (textures only; Solid.hs:138-254, Tex.hs:73-74) over the same word stack: asked of the solid a
// Difference carves, at the carved point (Csg.hs:103-106).  Out of line like vm_inside_composite (rare, and large).
//   frames: word 0 = tag | previous frame << 8, 1-2 pre (the Tex wrappers passed on the way down, outermost first),
//   3-4 res, then   MT_LIST / MT_ISECT  5 next record, 6 records left      MT_INST  5-7 the outer point      MT_PRE  -
//                   MT_BIH  5 next leaf record, 6 leaf records left, 7-8 the leaf's stack so far, 9 entries, 10 reference to look at, 11.. entries
enum : uint32_t { MT_DONE = 0, MT_LIST, MT_ISECT, MT_INST, MT_PRE, MT_BIH };
GD TexStack vm_meta_prim(const DScene& S, U4 rec, bool& is_prim) {  // a primitive under Tex wrappers: ([],[]) plus the folded wrappers
  TexStack pre = 0;
  while ((rec.x & RF_KINDMASK) == R_TEX) { pre = tex_cat(pre, (TexStack)(rec.z + 1), (int)S.tex_bits); rec = ldu4(S.recs, rec.y); }
  const uint32_t kind = rec.x & RF_KINDMASK;
  is_prim = kind >= R_SPHERE && kind <= R_CONE;
  return is_prim ? tex_cat(pre, own_stack_meta(rec.z, (int)S.tex_bits), (int)S.tex_bits) : 0;
}
GDN TexStack vm_meta(const DScene& S, unsigned int& err, uint32_t* m, int base, U4 rec, V3 p) {
  if (base + 1 > kVmWords) { err = 1; return 0; }
  int sp = base + 1, fb = base;
  m[base] = MT_DONE;
  TexStack val = 0;
  bool ret = false;
#define MT_PUSH(tag, n) { if (sp + (n) > kVmWords) { err = 1; return 0; } m[sp] = (uint32_t)(tag) | ((uint32_t)fb << 8); fb = sp; sp += (n); }
#define MT_POP() { sp = fb; fb = (int)(m[fb] >> 8); }
#define MT_GET(i) ((TexStack)m[fb + (i)] | ((TexStack)m[fb + (i) + 1] << 32))
#define MT_SET(i, t) { m[fb + (i)] = (uint32_t)(t); m[fb + (i) + 1] = (uint32_t)((t) >> 32); }
  for (;;) {
    if (!ret) {  // evaluate `get_metainfo rec p`
      TexStack pre = 0;  // Tex records passed on the way down: tex : texs (Tex.hs:73-74), outermost first
      while ((rec.x & RF_KINDMASK) == R_TEX) { pre = tex_cat(pre, (TexStack)(rec.z + 1), (int)S.tex_bits); rec = ldu4(S.recs, rec.y); }
      const uint32_t kind = rec.x & RF_KINDMASK;
      ret = true;
      if (kind >= R_SPHERE && kind <= R_CONE) { val = tex_cat(pre, own_stack_meta(rec.z, (int)S.tex_bits), (int)S.tex_bits); continue; }
      val = pre;  // (what a composite without textures at p answers: pre ++ [])
      switch (kind) {
        case R_LIST:  // Solid.hs:337-339: later containing items are prepended
          if (rec.z != 0) { MT_PUSH(MT_LIST, 7); MT_SET(1, pre); MT_SET(3, (TexStack)0); m[fb + 5] = rec.y; m[fb + 6] = rec.z; val = 0; }
          break;
        case R_ISECT:  // Csg.hs:108-111: all inside -> the items' answers in order
          if (rec.z != 0 && vm_inside(S, err, m, sp, rec, p)) { MT_PUSH(MT_ISECT, 7); MT_SET(1, pre); MT_SET(3, (TexStack)0); m[fb + 5] = rec.y; m[fb + 6] = rec.z; val = 0; }
          break;
        case R_INSTANCE: {  // Solid.hs:517-519
          Xf6 x = load_xf(S, rec.z);
          MT_PUSH(MT_INST, 8); MT_SET(1, pre);
          m[fb + 5] = as_u(p.x); m[fb + 6] = as_u(p.y); m[fb + 7] = as_u(p.z);
          p = mat_point(x.i0, x.i1, x.i2, p);
          rec = ldu4(S.recs, rec.y); ret = false;
          break;
        }
        case R_DIFF: {  // Csg.hs:103-106
          const U4 a = ldu4(S.recs, rec.y);
          if (vm_inside(S, err, m, sp, a, p) && !vm_inside(S, err, m, sp, ldu4(S.recs, rec.z), p)) { MT_PUSH(MT_PRE, 3); MT_SET(1, pre); rec = a; ret = false; }
          break;
        }
        case R_BOUND:  // Bound.hs:54-58
          if (vm_inside(S, err, m, sp, ldu4(S.recs, rec.y), p)) { MT_PUSH(MT_PRE, 3); MT_SET(1, pre); rec = ldu4(S.recs, rec.z); ret = false; }
          break;
        case R_INNERBOUND: { MT_PUSH(MT_PRE, 3); MT_SET(1, pre); rec = ldu4(S.recs, rec.z); ret = false; break; }
        case R_BIH: {  // get_metainfo_bih, Bih.hs:567-585: left result ++ right result, leaves like lists
          F4 h0 = ld4(S.bihhdr, 3 * rec.y), h1 = ld4(S.bihhdr, 3 * rec.y + 1);
          if (!(p.x > h0.x && p.x < h1.x && p.y > h0.y && p.y < h1.y && p.z > h0.z && p.z < h1.z)) break;
          MT_PUSH(MT_BIH, 11); MT_SET(1, pre); MT_SET(3, (TexStack)0);
          m[fb + 6] = 0; MT_SET(7, (TexStack)0); m[fb + 9] = 0; m[fb + 10] = as_u(h0.w);
          val = 0;
          break;
        }
        default: break;  // Mesh: the class default ([],[]); Void
      }
      continue;
    }
    // a value has come back to the frame on top
    const uint32_t tag = m[fb] & 0xffu;
    if (tag == MT_DONE) return val;
    if (tag == MT_PRE) { val = tex_cat(MT_GET(1), val, (int)S.tex_bits); MT_POP(); continue; }
    if (tag == MT_INST) { p = v3(as_f(m[fb + 5]), as_f(m[fb + 6]), as_f(m[fb + 7])); val = tex_cat(MT_GET(1), val, (int)S.tex_bits); MT_POP(); continue; }
    if (tag == MT_LIST || tag == MT_ISECT) {
      // (the frame is entered with val = 0 and nothing pending; later with a child's answer)
      const bool is_list = tag == MT_LIST;
      TexStack res = MT_GET(3);
      uint32_t cur = m[fb + 5], left = m[fb + 6];
      if (m[fb] & 0x80000000u) { res = is_list ? tex_cat(val, res, (int)S.tex_bits) : tex_cat(res, val, (int)S.tex_bits); m[fb] &= 0x7fffffffu; }
      bool called = false;
      while (left != 0) {
        const U4 c = ldu4(S.recs, cur); cur++; left--;
        if (is_list && !vm_inside(S, err, m, sp, c, p)) continue;
        bool is_prim;
        const TexStack v = vm_meta_prim(S, c, is_prim);
        if (is_prim) { res = is_list ? tex_cat(v, res, (int)S.tex_bits) : tex_cat(res, v, (int)S.tex_bits); continue; }
        MT_SET(3, res); m[fb + 5] = cur; m[fb + 6] = left; m[fb] |= 0x80000000u;  // (bit 31: a child's answer is pending)
        rec = c; ret = false; called = true;
        break;
      }
      if (!called) { val = tex_cat(MT_GET(1), res, (int)S.tex_bits); MT_POP(); }
      continue;
    }
    {  // MT_BIH
      TexStack res = MT_GET(3), leaf = MT_GET(7);
      uint32_t cur = m[fb + 5], left = m[fb + 6], ref = m[fb + 10];
      if (m[fb] & 0x80000000u) { leaf = tex_cat(val, leaf, (int)S.tex_bits); m[fb] &= 0x7fffffffu; }
      bool called = false, done = false;
      for (;;) {
        while (left != 0) {  // the leaf the walk stands on, like a list
          const U4 c = ldu4(S.recs, cur); cur++; left--;
          if (!vm_inside(S, err, m, sp, c, p)) continue;
          bool is_prim;
          const TexStack v = vm_meta_prim(S, c, is_prim);
          if (is_prim) { leaf = tex_cat(v, leaf, (int)S.tex_bits); continue; }
          MT_SET(3, res); MT_SET(7, leaf); m[fb + 5] = cur; m[fb + 6] = left; m[fb + 10] = ref; m[fb] |= 0x80000000u;
          rec = c; ret = false; called = true;
          break;
        }
        if (called) break;
        res = tex_cat(res, leaf, (int)S.tex_bits); leaf = 0;
        if (ref == 0xffffffffu) {
          const uint32_t ne = m[fb + 9];
          if (ne == 0) { done = true; break; }
          ref = m[fb + 11 + ne - 1]; m[fb + 9] = ne - 1; sp = fb + 11 + (int)ne - 1;
        }
        if (ref & BREF_LEAF) {
          uint32_t count = (ref >> 26) & 7u, first = ref & BREF_FIRST;
          if (count == 7u) { F4 n = ld4(S.bihnodes, first); count = as_u(n.z); first = as_u(n.w); }
          cur = first; left = count; ref = 0xffffffffu;
        } else {
          F4 n = ld4(S.bihnodes, ref);
          uint32_t w0 = as_u(n.z), w1 = as_u(n.w), axis = w0 & 3u;
          float o = vcomp(p, axis);
          bool gl = o < n.x, gr = o > n.y;
          if (gl) {
            if (gr) { if (sp + 1 > kVmWords) { err = 1; return 0; } m[sp++] = w1; m[fb + 9]++; }
            ref = w0 >> 2;
          } else if (gr) ref = w1;
          else ref = 0xffffffffu;
        }
      }
      if (done) { val = tex_cat(MT_GET(1), res, (int)S.tex_bits); MT_POP(); }
      continue;
    }
  }
#undef MT_PUSH
#undef MT_POP
#undef MT_GET
#undef MT_SET
}

// Frame layouts (word offsets from the frame base fb; word 0 = tag | previous fb << 8):
//   LIST_R   1 first record, 2 n, 3 k, 4 d, 5-6 tex, 7.. best hit            LIST_S  1 first, 2 n, 3 k, 4 d
//   INST_R   1-6 outer ray, 7 1/lenscale, 8 exact, 9 transform               INST_S  1-6 outer ray
//   BOUND_R  1 record b, 2 d, 3-4 tex                                         BOUND_S 1 record b, 2 d
//   IB_R     1 record b, 2-3 tex                                              IB_S    1 record b, 2 d
//   DIFF     1 record a, 2 record b, 3-4 tex, 5-7 entry origin, 8 d (current), 9 advances, 10.. hit of a, 27.. the advances
//   ISECT    1 first record, 2 n, 3 from, 4-5 tex, 6-8 origin, 9 d, 10 aux    (one frame per IFrame of the recursive form)
//   BIH      1-2 tex, 3 flags (1 root is a leaf, 2 exact walk, 4 shadow), 4 d, 5-7 1/direction, 8 leaf cursor, 9 items left,
//            10 the leaf's tmax, 11 traversal entries, [12.. best hit (rayint only)], then the entries (node, near, far)
constexpr int kBihFixedS = kVmBihFixedS, kBihFixedR = kVmBihFixedR, kDiffFixed = kVmDiffFixed, kIsectWords = kVmIsectWords;

// `pk`: the wave's packet stack (LDS rows, rt_device.hpp LaneStack) or null.  With one, a rayint / shadow call whose callee is a
// BIH of plain spheres or plain triangles (BC_SPHERE, BC_TRI: GlomeView's default scene spends three fifths of its frame in the 9,261-sphere lattice,
// which a carving Difference walks again after every advance) is not walked lane by lane over frames: the lane waits in ST_PK_R /
// ST_PK_S, and once per pass all lanes that wait for the same tree are walked as ONE packet by the flat tier's wave-wide walk
// (bih_tri_wave: wave-uniform node references, scalar loads, per-lane intervals) -- same hit and same tie order as the per-lane
// walk: frames are bit-identical with the service switched off (GLOME_DEBUG_NO_GENERIC_PACKETS, GPU test), node counters 2 % apart.  A ray that is not unit length keeps the per-lane walk (the
// ordered early-out is exact for unit rays only).
template <bool C, int PKMIN = kPkMinLanes, class PK>
GD void vm_run(const DScene& S, Cnt& cnt, unsigned int& err, uint32_t* m, PK* pk, int st, U4 rec, Ray r, float d, bool exact, HitG& rh, bool& rb) {
  int sp = 1, fb = 0;
  m[0] = VT_DONE;
  TexStack tex = 0;
  rh = hit_miss(); rb = false;
  uint32_t ref = 0; float nearv = 0, farv = 0, bt = 0;  // the BIH walk's registers (live between ST_BIH steps only)
  uint32_t pk_hdr = 0;  // the tree a lane in ST_PK_R / ST_PK_S waits for
#define VM_NEED(n) if (sp + (n) > kVmWords) { err = 1; rh = hit_miss(); rb = false; return; }
#define VM_PUSH(tag, n) { VM_NEED(n); m[sp] = (uint32_t)(tag) | ((uint32_t)fb << 8); fb = sp; sp += (n); }
#define VM_POP() { sp = fb; fb = (int)(m[fb] >> 8); }
#define VM_TAG(tag) m[fb] = (m[fb] & ~0xffu) | (uint32_t)(tag)
#define VM_TEX(i) ((TexStack)m[fb + (i)] | ((TexStack)m[fb + (i) + 1] << 32))
// a rayint call issued before ST_RET in the pass: a primitive callee answers at once and the caller's frame goes on in this pass
#define VM_CALL_R_INLINE() { const int what_ = vm_resolve_r(S, rec, tex); \
    if (what_ == 2) { if (vm_inst_prim_hit<C>(S, cnt, rec, tex, r, d, rh)) st = ST_RET; else st = ST_CALL_R; } \
    else { rh = what_ == 0 ? vm_prim_hit<C>(S, cnt, rec, r, d, tex) : hit_miss(); st = ST_RET; } }
#define VM_SET_TEX(i, t) { m[fb + (i)] = (uint32_t)(t); m[fb + (i) + 1] = (uint32_t)((t) >> 32); }
  for (;;) {
    // One pass runs the states in an order that lets the common chains finish inside it: leaf item -> call of a primitive ->
    // return to the BIH frame; return to a list -> call of the next child -> its return.
    if (st == ST_BIH_ITEM) do {  // the items of the leaf the walk stands on: primitives in place, a composite through a call
      const uint32_t flags = m[fb + 3];
      const float tmax = as_f(m[fb + 10]);
      uint32_t left = m[fb + 9], cur = m[fb + 8];
      bool called = false;
      if (flags & 4u) {  // shadow_bih: any item (Bih.hs:510-544)
        const float dd = gminf(as_f(m[fb + 4]), tmax);
        while (left != 0) {
          U4 it = ldu4(S.recs, cur); cur++; left--;
          const int what = vm_resolve_s(S, it);
          if (what == 1) continue;
          if (what == 2) {
            const int ip = vm_inst_prim_shadow<C>(S, cnt, it, r, dd);  // an Instance of a primitive: in place
            if (ip == 2) { rb = true; VM_POP(); st = ST_RET; called = true; break; }
            if (ip == 1) continue;
            m[fb + 8] = cur; m[fb + 9] = left; rec = it; d = dd; st = ST_CALL_S; called = true; break;
          }
          if (C) cnt.prim++;
          if (prim_shadow(S, it.x & RF_KINDMASK, it.y, r, dd)) { rb = true; VM_POP(); st = ST_RET; called = true; break; }
        }
      } else {
        const bool ordered = !(flags & 2u);
        const TexStack ftex = VM_TEX(1);
        bool besthit = m[fb + 12] != 0;
        float bestt = besthit ? as_f(m[fb + 13]) : 0.0f;
        while (left != 0) {
          U4 it = ldu4(S.recs, cur); cur++; left--;
          TexStack t = ftex;
          const int what = vm_resolve_r(S, it, t);
          if (what == 1) continue;
          if (what == 2) {
            HitG h;
            if (vm_inst_prim_hit<C>(S, cnt, it, ftex, r, tmax, h)) {  // an Instance of a primitive: in place, `rayint s r far` (the node's own far)
              if (h.hit && (!besthit || !(bestt < h.t))) { vm_st_hit(m, fb + 12, h); besthit = true; bestt = h.t; }
              continue;
            }
            m[fb + 8] = cur; m[fb + 9] = left; rec = it; tex = ftex; d = tmax; st = ST_CALL_R; called = true; break;
          }
          // a plain primitive other than a quadric answers the same for every tmax beyond its hit: it may be tested against
          // the best so far (the lattice of GlomeView's default scene is 9261 such spheres); a cylinder or a cone sees the
          // node's own `far` (rt_device.hpp bih_traverse, CLAMP)
          const uint32_t ik = it.x & RF_KINDMASK;
          const float dc = (ordered && besthit && ik != R_CYL && ik != R_CONE) ? gminf(tmax, bestt) : tmax;
          const HitG h = vm_prim_hit<C>(S, cnt, it, r, dc, t);
          if (h.hit && (!besthit || !(bestt < h.t))) { vm_st_hit(m, fb + 12, h); besthit = true; bestt = h.t; }  // nearest: ties -> the later item
        }
        if (!called) bt = (ordered && besthit) ? bestt : kInf * 4.0f;
      }
      if (!called) { ref = 0xffffffffu; st = ST_BIH; }
    } while (0);
    if (st == ST_ENTER_CSG) do {
      if ((rec.x & RF_KINDMASK) == R_DIFF && (rec.x & RF_PRIMLIST)) {  // a Difference of two primitives: the flat tier's loop (csg_diff), in place
        rh = csg_diff<C>(S, cnt, err, rec, r, d, tex);
        st = ST_RET;
      } else if ((rec.x & RF_KINDMASK) == R_DIFF) {  // rayint_difference, Csg.hs:33-54 (Q13)
        VM_PUSH(VT_DIFF_B, kDiffFixed);
        m[fb + 1] = rec.y; m[fb + 2] = rec.z; VM_SET_TEX(3, tex);
        m[fb + 5] = as_u(r.o.x); m[fb + 6] = as_u(r.o.y); m[fb + 7] = as_u(r.o.z);
        m[fb + 8] = as_u(d); m[fb + 9] = (rec.x & RF_RETEX) ? 0x80000000u : 0u;  // advances so far; bit 31: Difference a b False (keep B's textures)
        st = ST_DIFF;
      } else if (rec.x & RF_PRIMLIST) {
        // an Intersection of primitives (the dodecahedron and the icosahedron of GlomeView's default scene: a sphere and 12 / 20
        // planes): rayint_intersection as the flat tier's CSG class runs it (csg_isect: the same loop over its own small frames)
        rh = csg_isect<C>(S, cnt, err, rec, r, d, tex);
        st = ST_RET;
      } else {  // rayint_intersection, Csg.hs:68-90 (Q14)
        VM_PUSH(VT_ISECT_HS, kIsectWords);
        m[fb + 1] = rec.y; m[fb + 2] = rec.z; m[fb + 3] = 0; VM_SET_TEX(4, tex);
        m[fb + 6] = as_u(r.o.x); m[fb + 7] = as_u(r.o.y); m[fb + 8] = as_u(r.o.z);
        m[fb + 9] = as_u(d); m[fb + 10] = 0;
        st = ST_ISECT;
      }
    } while (0);
    if (st == ST_DIFF) do {  // one round of the advance loop (the self-recursion through rayint_advance, Solid.hs:85-91)
      const U4 rbrec = ldu4(S.recs, m[fb + 2]);
      tex = VM_TEX(3); d = as_f(m[fb + 8]);
      if (vm_inside(S, err, m, sp, rbrec, r.o)) { VM_TAG(VT_DIFF_B); rec = rbrec; }
      else { VM_TAG(VT_DIFF_A); rec = ldu4(S.recs, m[fb + 1]); }
      VM_CALL_R_INLINE();
    } while (0);
    if (st == ST_ISECT) do {  // a fresh frame: `rayint (Intersection slds) r d` at list position `from`
      const uint32_t from = m[fb + 3], n = m[fb + 2];
      const float fd = as_f(m[fb + 9]);
      r.o = v3(as_f(m[fb + 6]), as_f(m[fb + 7]), as_f(m[fb + 8]));
      if (from >= n || fd < 0) { rh = hit_miss(); VM_POP(); st = ST_RET; break; }  // null slds || d < 0
      rec = ldu4(S.recs, m[fb + 1] + from); tex = VM_TEX(4); d = fd;
      VM_TAG(VT_ISECT_HS);
      VM_CALL_R_INLINE();
    } while (0);
    if (st == ST_CALL_R) do {
      bool novis = false;
      for (;;) {  // Tex s tex: rayint s r d (tex:texs) tags, Tex.hs:66; OnlyShadow misses (Tex.hs:89)
        if (rec.x & RF_NOVIS) { novis = true; break; }
        if ((rec.x & RF_KINDMASK) != R_TEX) break;
        tex = tex_push(tex, rec.z, (int)S.tex_bits);
        rec = ldu4(S.recs, rec.y);
      }
      st = ST_RET;
      if (novis) { rh = hit_miss(); break; }
      const uint32_t kind = rec.x & RF_KINDMASK;
      if (kind >= R_SPHERE && kind <= R_CONE) {
        rh = hit_miss();
        if (C) cnt.prim++;
        float t; V3 n;
        if (prim_test<true>(S, kind, rec.y, r, d, t, n)) {
          rh.hit = true; rh.t = t; rh.n = n; rh.p = vscaleadd(r.o, r.d, t); rh.lo = r.o; rh.ld = r.d;
          rh.tex = tex_cat(own_stack_rayint(rec.z, (int)S.tex_bits), tex, (int)S.tex_bits); rh.uid = rec.w;
        }
        break;
      }
      switch (kind) {
        case R_LIST: {  // foldl' nearest RayMiss, every item with the same d (Solid.hs:327, Q9)
          rh = hit_miss();
          if (rec.z == 0) break;
          VM_PUSH(VT_LIST_R, kVmListR);
          m[fb + 1] = rec.y; m[fb + 2] = rec.z; m[fb + 3] = 0; m[fb + 4] = as_u(d); VM_SET_TEX(5, tex); m[fb + 7] = 0;
          st = ST_LIST_R;
          break;
        }
        case R_INSTANCE: {  // rayint_instance, Solid.hs:388-403 (Q8)
          Xf6 x = load_xf(S, rec.z);
          V3 newdir = mat_vec(x.i0, x.i1, x.i2, r.d), neworig = mat_point(x.i0, x.i1, x.i2, r.o);
          float lenscale = sqrtf(vdot(newdir, newdir)), invlenscale = 1.0f / lenscale;
          VM_PUSH(VT_INST_R, kVmInstR);
          m[fb + 1] = as_u(r.o.x); m[fb + 2] = as_u(r.o.y); m[fb + 3] = as_u(r.o.z);
          m[fb + 4] = as_u(r.d.x); m[fb + 5] = as_u(r.d.y); m[fb + 6] = as_u(r.d.z);
          m[fb + 7] = as_u(invlenscale); m[fb + 8] = exact ? 1u : 0u; m[fb + 9] = rec.z;
          r.o = neworig; r.d = newdir * invlenscale; d = d * lenscale;
          exact = false;  // (the local ray is unit length)
          rec = ldu4(S.recs, rec.y); st = ST_CALL_R;
          break;
        }
        case R_DIFF: case R_ISECT: st = ST_ENTER_CSG; break;
        case R_BOUND: {  // rayint_bound, Bound.hs:30-35
          U4 sa = ldu4(S.recs, rec.y);
          if (vm_inside(S, err, m, sp, sa, r.o)) { rec = ldu4(S.recs, rec.z); st = ST_CALL_R; break; }
          VM_PUSH(VT_BOUND_R, kVmBoundR);
          m[fb + 1] = rec.z; m[fb + 2] = as_u(d); VM_SET_TEX(3, tex);
          rec = sa; st = ST_CALL_S;
          break;
        }
        case R_INNERBOUND: {  // rayint_innerbound, Bound.hs:97-99
          VM_PUSH(VT_IB_R, kVmIbR);
          m[fb + 1] = rec.z; VM_SET_TEX(2, tex);
          rec = ldu4(S.recs, rec.y); tex = 0; st = ST_CALL_R;
          break;
        }
        case R_BIH: {  // rayint_bih, Bih.hs:332-368
          F4 h0 = ld4(S.bihhdr, 3 * rec.y), h1 = ld4(S.bihhdr, 3 * rec.y + 1);
          if (pk != nullptr && !exact) {  // a tree the service walks as a packet: spheres, triangles, or items answered in place -- if the wave's stack holds its depth
            const uint32_t dw = as_u(ld4(S.bihhdr, 3 * rec.y + 2).w);
            if ((as_u(h1.w) == BC_SPHERE || as_u(h1.w) == BC_TRI || ((dw & kBihItemsInPlace) && sp + kHitWords <= kVmWords)) && (int)(dw & ~kBihItemsInPlace) <= pk->total_cap()) { pk_hdr = rec.y; st = ST_PK_R; break; }  // (kHitWords: where bih_items_wave keeps the nearest hit)
          }
          bbclip_ub(r, v3(h0), v3(h1), nearv, farv);
          farv = gminf(d, farv);  // `traverse root near (fmin d far)`, Bih.hs:368
          ref = as_u(h0.w);
          VM_PUSH(VT_BIH_R, kBihFixedR);
          VM_SET_TEX(1, tex);
          m[fb + 3] = ((ref & BREF_LEAF) ? 1u : 0u) | (exact ? 2u : 0u); m[fb + 4] = as_u(d);
          m[fb + 5] = as_u(dir_rcp(r.d.x)); m[fb + 6] = as_u(dir_rcp(r.d.y)); m[fb + 7] = as_u(dir_rcp(r.d.z));
          m[fb + 9] = 0; m[fb + 11] = 0; m[fb + 12] = 0;
          bt = kInf * 4.0f;
          st = ST_BIH;
          break;
        }
        case R_MESH: rh = vm_mesh_rayint<C>(S, cnt, rec, r, d, tex); break;
        default: rh = hit_miss(); break;
      }
    } while (0);
    if (st == ST_CALL_S) do {
      bool noshadow = false;
      for (;;) {  // shadow (Tex s _) = shadow s; NoShadow -> False (Tex.hs:69, 81)
        if (rec.x & RF_NOSHADOW) { noshadow = true; break; }
        if ((rec.x & RF_KINDMASK) != R_TEX) break;
        rec = ldu4(S.recs, rec.y);
      }
      st = ST_RET;
      rb = false;
      if (noshadow) break;
      const uint32_t kind = rec.x & RF_KINDMASK;
      if (kind >= R_SPHERE && kind <= R_CONE) { if (C) cnt.prim++; rb = prim_shadow(S, kind, rec.y, r, d); break; }
      switch (kind) {
        case R_LIST: {  // foldl' (||) False (Solid.hs:330)
          if (rec.z == 0) break;
          VM_PUSH(VT_LIST_S, 5);
          m[fb + 1] = rec.y; m[fb + 2] = rec.z; m[fb + 3] = 0; m[fb + 4] = as_u(d);
          st = ST_LIST_S;
          break;
        }
        case R_INSTANCE: {  // shadow_instance, Solid.hs:464-471
          Xf6 x = load_xf(S, rec.z);
          V3 newdir = mat_vec(x.i0, x.i1, x.i2, r.d), neworig = mat_point(x.i0, x.i1, x.i2, r.o);
          float lenscale = sqrtf(vdot(newdir, newdir)), invlenscale = 1.0f / lenscale;
          VM_PUSH(VT_INST_S, 7);
          m[fb + 1] = as_u(r.o.x); m[fb + 2] = as_u(r.o.y); m[fb + 3] = as_u(r.o.z);
          m[fb + 4] = as_u(r.d.x); m[fb + 5] = as_u(r.d.y); m[fb + 6] = as_u(r.d.z);
          r.o = neworig; r.d = newdir * invlenscale; d = d * lenscale;
          rec = ldu4(S.recs, rec.y); st = ST_CALL_S;
          break;
        }
        // Difference / Intersection have no shadow method: the class default runs rayint (Solid.hs:218-221, Q15).
        // The default sees the node itself, so an OnlyShadow flag on it does not hide it here.
        case R_DIFF: case R_ISECT: {
          VM_PUSH(VT_S_OF_R, 1);
          tex = 0; st = ST_ENTER_CSG;
          break;
        }
        case R_BOUND: {  // shadow_bound, Bound.hs:44-49
          U4 sa = ldu4(S.recs, rec.y);
          if (vm_inside(S, err, m, sp, sa, r.o)) { rec = ldu4(S.recs, rec.z); st = ST_CALL_S; break; }
          VM_PUSH(VT_BOUND_S, 3);
          m[fb + 1] = rec.z; m[fb + 2] = as_u(d);
          rec = sa; st = ST_CALL_S;
          break;
        }
        case R_INNERBOUND: {  // Bound.hs:101-103
          VM_PUSH(VT_IB_S, 3);
          m[fb + 1] = rec.z; m[fb + 2] = as_u(d);
          rec = ldu4(S.recs, rec.y); st = ST_CALL_S;
          break;
        }
        case R_BIH: {  // shadow_bih, Bih.hs:510-544
          F4 h0 = ld4(S.bihhdr, 3 * rec.y), h1 = ld4(S.bihhdr, 3 * rec.y + 1);
          if (pk != nullptr) {
            const uint32_t dw = as_u(ld4(S.bihhdr, 3 * rec.y + 2).w);
            if ((as_u(h1.w) == BC_SPHERE || as_u(h1.w) == BC_TRI || (dw & kBihItemsInPlace)) && (int)(dw & ~kBihItemsInPlace) <= pk->total_cap()) { pk_hdr = rec.y; st = ST_PK_S; break; }
          }
          bbclip_ub(r, v3(h0), v3(h1), nearv, farv);
          farv = gminf(d, farv);
          ref = as_u(h0.w);
          VM_PUSH(VT_BIH_S, kBihFixedS);
          m[fb + 3] = ((ref & BREF_LEAF) ? 1u : 0u) | 4u; m[fb + 4] = as_u(d);
          m[fb + 5] = as_u(dir_rcp(r.d.x)); m[fb + 6] = as_u(dir_rcp(r.d.y)); m[fb + 7] = as_u(dir_rcp(r.d.z));
          m[fb + 9] = 0; m[fb + 11] = 0;
          bt = 0;
          st = ST_BIH;
          break;
        }
        default: break;  // Mesh: `shadow s r d = False` (Mesh.hs:210); Void
      }
    } while (0);
    // the packet service: every lane of the wave that is still in this loop comes by here once per pass
    if (pk != nullptr) {
      const bool want = st == ST_PK_R || st == ST_PK_S;
      LaneMask todo = wave_ballot(want);
      // a walk costs the same for one lane as for sixty-four: while fewer than kPkMinLanes wait and other lanes can still go on
      // (and may join them), the waiting ones wait.  (Measured on GlomeView's default scene, 1 / 8 / 16 / 32 / 64: 4.10 / 3.98 / 3.96
      // / 3.94 / 3.91 ms per frame, a frame alone 15.7 / 14.2 / 13.5 / 13.7 / 13.6: profiles/r03_probes/generic_tier_packet_service_ab.txt)
      // Round 4, with trees of items answered in place in the service too: a renderTile frame is better off NOT waiting (PKMIN 1: the test below
      // is gone from its kernel; GlomeView's default scene 2.51 -> 2.44 ms per frame; 16 and 32 change nothing), the adaptive sampler's kernel
      // is not (4.68 against 4.75): profiles/r04_probes/packet_service_min_lanes_ab.txt.  (A threshold read at run time, per launch, measured
      // like 64 in both: what the renderTile kernel gains is the code it no longer carries.)
      if (PKMIN > 1 && wave_count(want) < PKMIN && wave_any(!want)) todo = 0;
      while (todo != 0) {  // one walk per (tree, kind of call) among the waiting lanes
        const uint32_t h = uni(first_lane_value(todo, pk_hdr));
        const int kind = (int)uni(first_lane_value(todo, (uint32_t)st));
        const bool mine = want && pk_hdr == h && st == kind;
        todo &= ~wave_ballot(mine);
        float pbt = kNoBest;
        uint32_t prec = CAND_NONE;
        const uint32_t hcls = uni(as_u(ld4u(S.bihhdr, 3 * h + 1).w));  // the tree's leaf class: triangles, spheres, or items answered in place
        const bool tris = hcls == BC_TRI;
        if (hcls != BC_TRI && hcls != BC_SPHERE) {
          ItemPick pick;
          if (kind == ST_PK_R) {
            bih_items_wave<1, C>(S, cnt, h, r, d, tex, mine, *pk, pick, m, sp);
            if (mine) { rh = pick.item != CAND_NONE ? vm_ld_hit(m, sp) : hit_miss(); st = ST_RET; }
          } else {
            const bool occ = bih_items_wave<2, C>(S, cnt, h, r, d, tex, mine, *pk, pick, m, sp);
            if (mine) { rb = occ; st = ST_RET; }
          }
        } else if (kind == ST_PK_R) {
          if (tris) bih_tri_wave<1, C, 0>(S, h, r, d, mine, *pk, cnt, pbt, prec);
          else bih_tri_wave<1, C, 1>(S, h, r, d, mine, *pk, cnt, pbt, prec);
          if (mine) { rh = prec != CAND_NONE ? vm_prim_hit<false>(S, cnt, ldu4(S.recs, prec), r, kInf * 8.0f, tex) : hit_miss(); st = ST_RET; }
        } else {
          const bool occ = tris ? bih_tri_wave<2, C, 0>(S, h, r, d, mine, *pk, cnt, pbt, prec) : bih_tri_wave<2, C, 1>(S, h, r, d, mine, *pk, cnt, pbt, prec);
          if (mine) { rb = occ; st = ST_RET; }
        }
      }
    }
    if (st == ST_RET) do {
      switch (m[fb] & 0xffu) {
        case VT_DONE: return;
        case VT_LIST_R: {  // a composite child has answered
          if (rh.hit && (m[fb + 7] == 0 || !(as_f(m[fb + 8]) < rh.t))) vm_st_hit(m, fb + 7, rh);  // nearest: ties -> the later item
          st = ST_LIST_R;
          break;
        }
        case VT_LIST_S: {
          if (rb) VM_POP() else st = ST_LIST_S;
          break;
        }
        case VT_INST_R: {
          r.o = v3(as_f(m[fb + 1]), as_f(m[fb + 2]), as_f(m[fb + 3]));
          r.d = v3(as_f(m[fb + 4]), as_f(m[fb + 5]), as_f(m[fb + 6]));
          exact = m[fb + 8] != 0;
          if (rh.hit) {
            Xf6 x = load_xf(S, m[fb + 9]);
            rh.t = rh.t * as_f(m[fb + 7]);
            rh.p = mat_point(x.f0, x.f1, x.f2, rh.p);
            rh.n = vnorm(mat_tvec(x.i0, x.i1, x.i2, rh.n));
          }
          VM_POP();
          break;
        }
        case VT_INST_S: {
          r.o = v3(as_f(m[fb + 1]), as_f(m[fb + 2]), as_f(m[fb + 3]));
          r.d = v3(as_f(m[fb + 4]), as_f(m[fb + 5]), as_f(m[fb + 6]));
          VM_POP();
          break;
        }
        case VT_BOUND_R: {
          if (rb) { rec = ldu4(S.recs, m[fb + 1]); d = as_f(m[fb + 2]); tex = VM_TEX(3); st = ST_CALL_R; }
          else rh = hit_miss();
          VM_POP();
          break;
        }
        case VT_BOUND_S: {
          if (rb) { rec = ldu4(S.recs, m[fb + 1]); d = as_f(m[fb + 2]); st = ST_CALL_S; }
          VM_POP();
          break;
        }
        case VT_IB_R: {
          d = rh.hit ? rh.t : kInf;
          rec = ldu4(S.recs, m[fb + 1]); tex = VM_TEX(2); st = ST_CALL_R;
          VM_POP();
          break;
        }
        case VT_IB_S: {
          if (!rb) { rec = ldu4(S.recs, m[fb + 1]); d = as_f(m[fb + 2]); st = ST_CALL_S; }
          VM_POP();
          break;
        }
        case VT_S_OF_R: rb = rh.hit; VM_POP(); break;
        case VT_DIFF_B: case VT_DIFF_AB: case VT_DIFF_A: {
          const uint32_t tag = m[fb] & 0xffu;
          HitG res = hit_miss();
          bool finish = true;
          float adv = 0;
          if (tag == VT_DIFF_B) {
            if (rh.hit) {
              const U4 ra = ldu4(S.recs, m[fb + 1]), rbrec = ldu4(S.recs, m[fb + 2]);
              if (vm_inside(S, err, m, sp, ra, rh.p) && !vm_inside(S, err, m, sp, rbrec, vscaleadd(rh.p, r.d, kDel))) {
                res = rh;
                res.n = vneg(rh.n);
                if (!(m[fb + 9] >> 31)) res.tex = vm_meta(S, err, m, sp, ra, rh.p);  // `difference` = Difference a b True: textures of A at the carved point (difference_retexture: B's, Csg.hs:42-43)
              } else { finish = false; adv = rh.t; }
            }
          } else {
            HitG ha;
            bool have_b = tag == VT_DIFF_AB;
            if (have_b) ha = vm_ld_hit(m, fb + 10);
            else if (rh.hit) {  // the hit of a is in; now b, with the same ray
              ha = rh;
              rec = ldu4(S.recs, m[fb + 2]); tex = VM_TEX(3); d = as_f(m[fb + 8]);
              const int what = vm_resolve_r(S, rec, tex);
              if (what == 2) { vm_st_hit(m, fb + 10, ha); VM_TAG(VT_DIFF_AB); st = ST_CALL_R; break; }
              rh = what == 0 ? vm_prim_hit<C>(S, cnt, rec, r, d, tex) : hit_miss();  // (a primitive b answers in place)
              have_b = true;
            }
            if (have_b) {
              if (!rh.hit || ha.t < rh.t) res = ha;
              else { finish = false; adv = rh.t; }
            }
          }
          const uint32_t na = m[fb + 9] & 0x7fffffffu;
          if (!finish) {
            // (the reference advances as often as it takes; here as long as the frame memory lasts -- the commit-time estimate
            // allows kCsgMaxAdvance per Difference, the flat tier's fixed cap)
            VM_NEED(1);
            const float a = adv + kDel;
            m[sp++] = as_u(a); m[fb + 9] = m[fb + 9] + 1u;
            r.o = vscaleadd(r.o, r.d, a);  // ray_move
            m[fb + 8] = as_u(as_f(m[fb + 8]) - a);
            st = ST_DIFF;
            break;
          }
          if (res.hit) for (int k = (int)na - 1; k >= 0; k--) res.t = res.t + as_f(m[fb + kDiffFixed + k]);  // RayHit (depth+a) ..., innermost first
          r.o = v3(as_f(m[fb + 5]), as_f(m[fb + 6]), as_f(m[fb + 7]));
          rh = res;
          VM_POP();
          break;
        }
        case VT_ISECT_HS: {
          const uint32_t from = m[fb + 3], n = m[fb + 2];
          const V3 o = v3(as_f(m[fb + 6]), as_f(m[fb + 7]), as_f(m[fb + 8]));
          const float fd = as_f(m[fb + 9]);
          if (from + 1 == n) { VM_POP(); break; }  // [] -> rayint s r d t tags
          const U4 s = ldu4(S.recs, m[fb + 1] + from);
          uint32_t nfrom; V3 no; float nd;
          if (vm_inside(S, err, m, sp, s, o)) {
            if (!rh.hit) { m[fb + 3] = from + 1; st = ST_ISECT; break; }  // RayMiss -> rayint (Intersection ss) r d: a tail call
            m[fb + 10] = as_u(rh.t); VM_TAG(VT_ISECT_S1);  // rest = rayint (Intersection ss) r sd
            nfrom = from + 1; no = o; nd = rh.t;
          } else {
            if (!rh.hit) { rh = hit_miss(); VM_POP(); break; }
            bool rest = true;  // inside (Intersection ss) sp: foldl' (&&) True
            for (uint32_t k = from + 1; k < n; k++) rest = rest && vm_inside(S, err, m, sp, ldu4(S.recs, m[fb + 1] + k), rh.p);
            if (rest) { VM_POP(); break; }  // RayHit sd sp sn r vzero st stags
            const float a = rh.t + kDel;  // rayint_advance (Intersection slds) r d t tags sd
            m[fb + 10] = as_u(a); VM_TAG(VT_ISECT_S2);
            nfrom = from; no = vscaleadd(o, r.d, a); nd = fd - a;
          }
          const int pf = fb;
          VM_PUSH(VT_ISECT_HS, kIsectWords);
          m[fb + 1] = m[pf + 1]; m[fb + 2] = n; m[fb + 3] = nfrom; m[fb + 4] = m[pf + 4]; m[fb + 5] = m[pf + 5];
          m[fb + 6] = as_u(no.x); m[fb + 7] = as_u(no.y); m[fb + 8] = as_u(no.z); m[fb + 9] = as_u(nd); m[fb + 10] = 0;
          st = ST_ISECT;
          break;
        }
        case VT_ISECT_S1: {
          const V3 o = v3(as_f(m[fb + 6]), as_f(m[fb + 7]), as_f(m[fb + 8]));
          r.o = o;
          if (rh.hit) { VM_POP(); break; }  // hit -> hit
          const float a = as_f(m[fb + 10]) + kDel;
          m[fb + 10] = as_u(a); VM_TAG(VT_ISECT_S2);
          const float nd = as_f(m[fb + 9]) - a;
          const V3 no = vscaleadd(o, r.d, a);
          const int pf = fb;
          VM_PUSH(VT_ISECT_HS, kIsectWords);
          m[fb + 1] = m[pf + 1]; m[fb + 2] = m[pf + 2]; m[fb + 3] = m[pf + 3]; m[fb + 4] = m[pf + 4]; m[fb + 5] = m[pf + 5];
          m[fb + 6] = as_u(no.x); m[fb + 7] = as_u(no.y); m[fb + 8] = as_u(no.z); m[fb + 9] = as_u(nd); m[fb + 10] = 0;
          st = ST_ISECT;
          break;
        }
        case VT_ISECT_S2: {
          r.o = v3(as_f(m[fb + 6]), as_f(m[fb + 7]), as_f(m[fb + 8]));
          if (rh.hit) rh.t = rh.t + as_f(m[fb + 10]);  // RayHit (depth+a) ...
          VM_POP();
          break;
        }
        case VT_BIH_R: {
          if (rh.hit && (m[fb + 12] == 0 || !(as_f(m[fb + 13]) < rh.t))) vm_st_hit(m, fb + 12, rh);
          bt = (!(m[fb + 3] & 2u) && m[fb + 12] != 0) ? as_f(m[fb + 13]) : kInf * 4.0f;
          st = ST_BIH_ITEM;
          break;
        }
        default: {  // VT_BIH_S
          if (rb) { VM_POP(); break; }
          st = ST_BIH_ITEM;
          break;
        }
      }
    } while (0);
    if (st == ST_LIST_R) do {  // the children from position k on: primitives in place, a composite through a call
      const uint32_t n = m[fb + 2], first = m[fb + 1];
      uint32_t k = m[fb + 3];
      const float ld = as_f(m[fb + 4]);
      const TexStack ltex = VM_TEX(5);
      HitG best = m[fb + 7] != 0 ? vm_ld_hit(m, fb + 7) : hit_miss();
      bool dirty = false, called = false;
      for (; k < n; k++) {
        U4 c = ldu4(S.recs, first + k);
        TexStack t = ltex;
        const int what = vm_resolve_r(S, c, t);
        if (what == 1) continue;
        if (what == 2) {
          HitG hi;
          if (vm_inst_prim_hit<C>(S, cnt, c, ltex, r, ld, hi)) {  // an Instance of a primitive: in place
            if (hi.hit && (!best.hit || !(best.t < hi.t))) { best = hi; dirty = true; }
            continue;
          }
          if (dirty) vm_st_hit(m, fb + 7, best);
          m[fb + 3] = k + 1; rec = c; tex = ltex; d = ld; st = ST_CALL_R; called = true;
          break;
        }
        const HitG h = vm_prim_hit<C>(S, cnt, c, r, ld, t);
        if (h.hit && (!best.hit || !(best.t < h.t))) { best = h; dirty = true; }
      }
      if (!called) { rh = best; VM_POP(); st = ST_RET; }
    } while (0);
    if (st == ST_LIST_S) do {
      const uint32_t n = m[fb + 2], first = m[fb + 1];
      uint32_t k = m[fb + 3];
      const float ld = as_f(m[fb + 4]);
      bool called = false;
      rb = false;
      for (; k < n; k++) {
        U4 c = ldu4(S.recs, first + k);
        const int what = vm_resolve_s(S, c);
        if (what == 1) continue;
        if (what == 2) {
          const int ip = vm_inst_prim_shadow<C>(S, cnt, c, r, ld);  // an Instance of a primitive: in place
          if (ip == 2) { rb = true; break; }
          if (ip == 1) continue;
          m[fb + 3] = k + 1; rec = c; d = ld; st = ST_CALL_S; called = true; break;
        }
        if (C) cnt.prim++;
        if (prim_shadow(S, c.x & RF_KINDMASK, c.y, r, ld)) { rb = true; break; }
      }
      if (!called) { VM_POP(); st = ST_RET; }
    } while (0);
    if (st == ST_BIH) do {  // up to kBihStepsPerPass steps of the walk (most of a ray's passes are spent here)
      const uint32_t flags = m[fb + 3];
      const bool root_leaf = flags & 1u, exactm = flags & 2u, shadowm = flags & 4u, ordered = !exactm && !shadowm;
      const int fixed = shadowm ? kBihFixedS : kBihFixedR;
      const V3 rcp = v3(as_f(m[fb + 5]), as_f(m[fb + 6]), as_f(m[fb + 7]));  // (dir_rcp of the ray at entry: rt_device.hpp)
      int ne = (int)m[fb + 11];
      const int bfb = fb;
      for (int rep = 0; rep < kBihStepsPerPass; rep++) {
        if (ref == 0xffffffffu) {  // take the next entry, or finish
          if (ne == 0) {
            if (!shadowm) rh = vm_ld_hit(m, fb + 12); else rb = false;
            VM_POP(); st = ST_RET;
            break;
          }
          ne--;
          const int e = fb + fixed + 3 * ne;
          ref = m[e]; nearv = as_f(m[e + 1]); farv = as_f(m[e + 2]);
          sp = e;
        }
        bool popit = true;
        const float geo_far = farv;  // the node's interval as the planes cut it (what its items are tested with)
        if (ordered) farv = gminf(farv, bt);
        if (ref & BREF_LEAF) {
          uint32_t count = (ref >> 26) & 7u, first = ref & BREF_FIRST;
          if (count == 7u) { F4 nn = ld4(S.bihnodes, first); count = as_u(nn.z); first = as_u(nn.w); }
          if (count != 0 && (exactm || root_leaf || !(nearv > farv))) {
            m[fb + 8] = first; m[fb + 9] = count; m[fb + 10] = as_u(ordered ? geo_far : farv);
            st = ST_BIH_ITEM;
            break;
          }
        } else {
          if (C) cnt.bih++;
          if (!(nearv > farv)) {
            F4 nn = ld4(S.bihnodes, ref);
            uint32_t w0 = as_u(nn.z), w1 = as_u(nn.w);
            uint32_t axis = w0 & 3u;
            float dirr = vcomp(rcp, axis), o = vcomp(r.o, axis);
            float dl = (nn.x - o) * dirr, dr = (nn.y - o) * dirr;
            uint32_t left = w0 >> 2, right = w1;
            uint32_t c1, c2; float c1far, c2near; bool go1, go2;
            if (dirr > 0) { c1 = left; go1 = nearv < dl; c1far = gminf(dl, farv); c2 = right; go2 = dr < farv; c2near = gmaxf(dr, nearv); }
            else { c1 = right; go1 = nearv < dr; c1far = gminf(dr, farv); c2 = left; go2 = dl < farv; c2near = gmaxf(dl, nearv); }
            go1 = go1 && c1 != BREF_LEAF;
            go2 = go2 && c2 != BREF_LEAF;
            if (ordered) {  // the children keep the interval the planes give them; `best` only decided go1 / go2
              c1far = gminf(dirr > 0 ? dl : dr, geo_far);
              farv = geo_far;
            }
            if (go1) {
              if (go2) {
                VM_NEED(3);
                m[sp] = c2; m[sp + 1] = as_u(c2near); m[sp + 2] = as_u(farv); sp += 3; ne++;
              }
              ref = c1; farv = c1far; popit = false;
            } else if (go2) {
              ref = c2; nearv = c2near; popit = false;
            }
          }
        }
        if (popit) ref = 0xffffffffu;
      }
      if (st != ST_RET) m[bfb + 11] = (uint32_t)ne;
    } while (0);
  }
#undef VM_NEED
#undef VM_PUSH
#undef VM_POP
#undef VM_TAG
#undef VM_TEX
#undef VM_SET_TEX
#undef VM_CALL_R_INLINE
}

template <bool C, int PKMIN = kPkMinLanes, class PK> GD HitG vm_closest(const DScene& S, Cnt& cnt, unsigned int& err, uint32_t* m, PK* pk, const Ray& r, float tmax, uint32_t root) {
  HitG h; bool b;
  // a ray that is not unit length (Refract's transmitted ray, Shader.hs:141): BIHs are walked exactly as the reference
  // walks them (rt_device.hpp bih_traverse: the ordered early-out's pruning is exact only for unit rays)
  vm_run<C, PKMIN>(S, cnt, err, m, pk, ST_CALL_R, ldu4(S.recs, root), r, tmax, !unit_length(r.d), h, b);
  return h;
}
template <bool C, int PKMIN = kPkMinLanes, class PK> GD bool vm_occluded(const DScene& S, Cnt& cnt, unsigned int& err, uint32_t* m, PK* pk, const Ray& r, float d, uint32_t root) {
  HitG h; bool b;
  vm_run<C, PKMIN>(S, cnt, err, m, pk, ST_CALL_S, ldu4(S.recs, root), r, d, false, h, b);
  return b;
}

}  // namespace glome
