// flatten.hpp -- turn the host scene graph (host_graph.hpp) into the packed pools of rt_types.h.
// This is the "Haskell host flattens the scene graph to packed SoA device buffers" step; it is pure host
// code (no HIP) so the layout can be inspected and tested without a GPU.
#pragma once
#include <limits>
#include <cstring>
#include <map>
#include <unordered_map>

#include "host_graph.hpp"
#include "rt_types.h"

namespace glome {

struct FlatScene {
  std::vector<U4> recs;
  std::vector<F4> spheres, tris, trinorms, boxes, planes, discs, quadrics, xfms, bihhdr, bihnodes, meshhdr, meshnodes, mtris, mats, wlights;
  std::vector<U4> mtrimeta, entries;
  std::vector<uint32_t> matkids;
  std::vector<float> tripairs;  // the pair records of the triangle BIHs' leaves (emit_pairs; bih_packet_asm.hpp)
  std::vector<F4> pknodes;      // the packet walk's copy of the triangle BIHs' branch nodes: same slots as bihnodes, child references in its own form
  uint32_t root_rec = 0;
  uint32_t tier = 1;
  int nesting_depth = 0, max_bih_depth = 0, max_mesh_depth = 0;
  int max_sphere_bih_depth = 0;  // deepest BIH the interpreter's packet service can walk: items all plain spheres, all plain triangles, or all answered in place (0: none)
  uint32_t tex_bits = 16;   // bits per id of a TexStack (rt_types.h): 8 when the scene has at most 254 materials
  int64_t n_other_prims = 0;
  std::string why_generic;  // why the flat tier was not chosen
  bool pk_all = true;       // every triangle BIH has the packet walk's node form (emit_bih)
};

inline float f32(double d) { return (float)d; }
inline float as_float_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline F4 mk4(double x, double y, double z, double w) { return F4{f32(x), f32(y), f32(z), f32(w)}; }
inline F4 mk4u(double x, double y, double z, uint32_t w) { return F4{f32(x), f32(y), f32(z), as_float_bits(w)}; }
// split planes are rounded outward so that an fp32 plane never cuts into the objects it bounds
inline float round_up(double d) { float f = (float)d; return ((double)f < d) ? std::nextafterf(f, INFINITY) : f; }
inline float round_down(double d) { float f = (float)d; return ((double)f > d) ? std::nextafterf(f, -INFINITY) : f; }

constexpr uint32_t BREF_LEAF_BIT = 1u << 29, BREF_FIRST_LIMIT = 1u << 26;  // child references, see rt_device.hpp

class Flattener {
 public:
  Flattener(const Graph& g, FlatScene& out) : G(g), F(out) {}

  void run(int root) {
    emit_materials();
    // generic representation: always built (the per-ray batch seams and the generic tier use it)
    U4 r = emit(root);
    F.root_rec = (uint32_t)F.recs.size();
    F.recs.push_back(r);
    F.nesting_depth = depth_of(root);
    F.tex_bits = G.mats.size() <= 254 ? 8 : 16;
    tex_cap = 64 / (int)F.tex_bits;
    if (tex_depth_of(root) > tex_cap)
      throw limit_error("a primitive lies under " + std::to_string(tex_depth_of(root)) + " nested textures; the device's texture stack holds " + std::to_string(tex_cap) +
                        (tex_cap < kMaxTexDepth ? " in a scene of more than 254 materials (" + std::to_string(kMaxTexDepth) + " otherwise)" : ""));
    const bool warps = bind_warps();  // Warp materials refer to records: the frame's, and the scene's they look into
    if (vm_words_of(root) + 2 > kVmWords)
      throw limit_error("scene nests deeper than the device interpreter's frame memory holds (" + std::to_string(vm_words_of(root) + 2) + " of " + std::to_string(kVmWords) + " words)");
    // flat tier: root program of simple entries
    F.tier = 0;
    if (warps) { F.tier = 1; F.why_generic = "a Warp material traces other roots than the scene's"; }
    collect_entries(root, 0, 0, 0);
    if (F.tier == 0 && F.max_bih_depth > kFlatStack) { F.tier = 1; F.why_generic = "BIH deeper than the LDS stack"; }
    if (F.tier == 0 && F.max_mesh_depth > kFlatStack) { F.tier = 1; F.why_generic = "Mesh BVH deeper than the LDS stack"; }
    if (F.tier != 0) F.entries.clear();
    // the generic tier walks BIHs / Mesh BVHs with a fixed scratch stack per level: a deeper tree is refused, not truncated
    if (F.tier != 0 && F.max_mesh_depth > kGenericStack)
      throw limit_error("Mesh tree deeper than the device traversal stack (" + std::to_string(F.max_mesh_depth) + " > " + std::to_string(kGenericStack) + ")");
    pad();
  }

 private:
  const Graph& G;
  FlatScene& F;
  std::unordered_map<int, U4> memo;  // host node id -> record value (shared subtrees are emitted once)

  static bool is_prim(int k) { return k >= K_SPHERE && k <= K_CONE; }

  // The packet walk tests a leaf's triangles two at a time.  A leaf of n triangles owns ceil(n / 2) consecutive PAIR RECORDS of
  // kPairWords words: every component of p1, e1, e2 of triangles 2j and 2j + 1 as an (A, B) pair -- p1x p1y p1z e1x e1y e1z e2x
  // e2y e2z, 18 floats -- then the number of the leaf's triangles from 2j on (what the walk's loop counts down) and the record
  // index of triangle 2j (what a hit reports).  An odd last triangle is paired with itself; the walk ignores that half.
  uint32_t emit_pairs(uint32_t first_prim, uint32_t count, uint32_t first_rec) {
    const uint32_t at = (uint32_t)F.tripairs.size();
    for (uint32_t j = 0; j < count; j += 2) {
      const size_t a = first_prim + j, b = j + 1 < count ? a + 1 : a;
      float r[kPairWords];
      for (int w = 0; w < 3; w++) {  // word w of a triangle record: (p1, .) (e1, .) (e2, .)
        const F4 &A = F.tris[3 * a + w], &B = F.tris[3 * b + w];
        r[6 * w + 0] = A.x; r[6 * w + 1] = B.x; r[6 * w + 2] = A.y; r[6 * w + 3] = B.y; r[6 * w + 4] = A.z; r[6 * w + 5] = B.z;
      }
      r[18] = as_float_bits(count - j);
      r[19] = as_float_bits(first_rec + j);
      F.tripairs.insert(F.tripairs.end(), r, r + kPairWords);
    }
    return at * 4u;  // byte offset of the leaf's first record
  }

  void pad() {  // never hand a null pool to a kernel
    auto p4 = [](std::vector<F4>& v) { if (v.empty()) v.push_back(F4{0, 0, 0, 0}); };
    p4(F.spheres); p4(F.tris); p4(F.trinorms); p4(F.boxes); p4(F.planes); p4(F.discs); p4(F.quadrics); p4(F.xfms);
    p4(F.bihhdr); p4(F.bihnodes); p4(F.meshhdr); p4(F.meshnodes); p4(F.mtris); p4(F.mats); p4(F.wlights);
    if (F.mtrimeta.empty()) F.mtrimeta.push_back(U4{0, 0, 0, 0});
    if (F.entries.empty()) F.entries.push_back(U4{0, 0, 0, 0});
    if (F.matkids.empty()) F.matkids.push_back(0);
    if (F.tripairs.empty()) F.tripairs.assign(kPairWords, 0.0f);
    if (F.pknodes.size() < F.bihnodes.size()) F.pknodes.resize(F.bihnodes.size(), F4{0, 0, 0, 0});
  }

  void emit_materials() {
    for (const Mat& m : G.mats) {
      uint32_t a = 0, b = 0;
      F4 m1{0, 0, 0, 0}, m2{0, 0, 0, 0};
      float w = 0;
      switch (m.kind) {
        case MAT_SURFACE: m1 = mk4(m.color[0], m.color[1], m.color[2], m.alpha); m2 = mk4(m.amb, m.kd, m.ks, m.shine); break;
        case MAT_REFLECT: m1 = mk4(m.refl, 0, 0, 0); break;
        case MAT_REFRACT: m1 = mk4(m.refl, m.refr, m.ior, 0); break;
        case MAT_LAYERS: a = (uint32_t)F.matkids.size(); b = (uint32_t)m.kids.size(); for (int k : m.kids) F.matkids.push_back((uint32_t)k); break;
        case MAT_BLEND:
          a = (uint32_t)m.a; b = (uint32_t)m.b; w = f32(m.weight);
          m1 = F4{as_float_bits((uint32_t)m.wfn), f32(m.wp[0]), f32(m.wp[1]), f32(m.wp[2])};  // weight function + parameters
          break;
        case MAT_WARP: break;  // filled in by bind_warps once the records exist
      }
      F.mats.push_back(F4{as_float_bits((uint32_t)m.kind), as_float_bits(a), as_float_bits(b), w});
      F.mats.push_back(m1);
      F.mats.push_back(m2);
    }
    if (mat_nest_max() > kMaxMatNest) throw limit_error("material Blend/AdditiveLayers nesting deeper than the device shader supports");
    if (G.mats.size() > 65534) throw limit_error("more than 65534 materials");
  }
  // Warp materials: (kind, frame record, scene record, transform) (first light, light count, -, -)
  bool bind_warps() {
    bool any = false;
    for (size_t k = 0; k < G.mats.size(); k++) {
      const Mat& m = G.mats[k];
      if (m.kind != MAT_WARP) continue;
      any = true;
      const uint32_t frame = slot(emit(m.wframe)), scene = m.wscene < 0 ? F.root_rec : slot(emit(m.wscene));
      F.nesting_depth = std::max(F.nesting_depth, std::max(depth_of(m.wframe), m.wscene < 0 ? 0 : depth_of(m.wscene)));
      if (std::max(tex_depth_of(m.wframe), m.wscene < 0 ? 0 : tex_depth_of(m.wscene)) > tex_cap) throw limit_error("a Warp material's frame / scene has more nested textures than the device's texture stack holds");
      if (std::max(vm_words_of(m.wframe), m.wscene < 0 ? 0 : vm_words_of(m.wscene)) + 2 > kVmWords)
        throw limit_error("a Warp material's frame / scene nests deeper than the device interpreter's frame memory holds");
      const uint32_t xf = (uint32_t)(F.xfms.size() / 6);
      for (int q = 0; q < 3; q++) F.xfms.push_back(mk4(m.wxf.f.m[4 * q], m.wxf.f.m[4 * q + 1], m.wxf.f.m[4 * q + 2], m.wxf.f.m[4 * q + 3]));
      for (int q = 0; q < 3; q++) F.xfms.push_back(mk4(m.wxf.i.m[4 * q], m.wxf.i.m[4 * q + 1], m.wxf.i.m[4 * q + 2], m.wxf.i.m[4 * q + 3]));
      const uint32_t first = (uint32_t)(F.wlights.size() / 2);
      for (const WarpLight& L : m.wlights) {  // the layout of DLight: pos, color, rad, shadow
        F.wlights.push_back(mk4(L.pos[0], L.pos[1], L.pos[2], L.color[0]));
        F.wlights.push_back(F4{f32(L.color[1]), f32(L.color[2]), f32(L.rad), as_float_bits(L.shadow ? 1u : 0u)});
      }
      F.mats[3 * k] = F4{as_float_bits((uint32_t)MAT_WARP), as_float_bits(frame), as_float_bits(scene), as_float_bits(xf)};
      F.mats[3 * k + 1] = F4{as_float_bits(first), as_float_bits((uint32_t)m.wlights.size()), 0, 0};
    }
    return any;
  }
  int mat_nest(int m, int guard) const {
    if (guard > 64) throw scene_error("material graph is cyclic");
    const Mat& M = G.mats[m];
    if (M.kind == MAT_LAYERS) { int d = 0; for (int k : M.kids) d = std::max(d, mat_nest(k, guard + 1)); return d + 1; }
    if (M.kind == MAT_BLEND) return 1 + std::max(mat_nest(M.a, guard + 1), mat_nest(M.b, guard + 1));
    return 0;
  }
  int mat_nest_max() const { int d = 0; for (size_t m = 0; m < G.mats.size(); m++) d = std::max(d, mat_nest((int)m, 0)); return d; }

  // composite nesting depth (Tex / Tag / NoShadow / OnlyShadow wrappers are free: the interpreter loops over them); reported
  // in glome_scene_info.  Nothing on the device is unrolled by it any more: all four class methods run over explicit frames
  // (rt_generic.hpp), and what bounds the nesting is the frame memory (vm_words_of below).
  mutable std::unordered_map<int, int> depth_memo;
  int depth_of(int id) const {
    auto it = depth_memo.find(id);
    if (it != depth_memo.end()) return it->second;
    const Node& n = G.at(id);
    int d = 0;
    switch (n.kind) {
      case K_LIST: case K_ISECT: { for (int k : n.kids) d = std::max(d, depth_of(k)); d++; break; }
      case K_INSTANCE: d = depth_of(n.a) + 1; break;
      case K_DIFF: case K_BOUND: case K_INNERBOUND: d = std::max(depth_of(n.a), depth_of(n.b)) + 1; break;
      case K_BIH: { for (auto& bn : n.bih->nodes) for (int k : bn.items) d = std::max(d, depth_of(k)); d++; break; }
      case K_MESH: d = 1; break;
      case K_TEX: case K_TAG: case K_NOSHADOW: case K_ONLYSHADOW: d = depth_of(n.a); break;
      default: break;
    }
    depth_memo[id] = d;
    return d;
  }
  // Frame words a rayint / shadow / inside call on `id` can have live at once in the generic tier's loop (rt_generic.hpp's
  // frame layouts): what the interpreter's nesting limit is now -- memory, checked here so that a scene is refused at commit
  // rather than stopped in the middle of a frame.  An Intersection's chain of advance frames depends on the geometry; the
  // estimate allows kVmIsectChain of them and the run-time check (GLOME_E_LIMIT) remains behind it.
  mutable std::unordered_map<int, int> words_memo, iwords_memo, mwords_memo;
  int bih_items_max(const Node& n, bool inside) const {
    int d = 0;
    for (auto& bn : n.bih->nodes) for (int k : bn.items) d = std::max(d, inside ? inside_words_of(k) : vm_words_of(k));
    return d;
  }
  int inside_words_of(int id) const {
    auto it = iwords_memo.find(id);
    if (it != iwords_memo.end()) return it->second;
    const Node& n = G.at(id);
    int d = 0;
    switch (n.kind) {
      case K_LIST: case K_ISECT: { for (int k : n.kids) d = std::max(d, inside_words_of(k)); d += 3; break; }
      case K_INSTANCE: d = 4 + inside_words_of(n.a); break;
      case K_DIFF: case K_BOUND: case K_INNERBOUND: d = 3 + std::max(inside_words_of(n.a), inside_words_of(n.b)); break;
      case K_BIH: d = 3 + n.bih->depth + 3 + bih_items_max(n, true); break;
      case K_TEX: case K_TAG: case K_NOSHADOW: case K_ONLYSHADOW: d = inside_words_of(n.a); break;
      default: break;
    }
    iwords_memo[id] = d;
    return d;
  }
  int meta_words_of(int id) const {  // get_metainfo's frames (vm_meta), with the inside calls it makes above them
    auto it = mwords_memo.find(id);
    if (it != mwords_memo.end()) return it->second;
    const Node& n = G.at(id);
    int d = 0;
    switch (n.kind) {
      case K_LIST: case K_ISECT: { for (int k : n.kids) d = std::max(d, std::max(meta_words_of(k), 1 + inside_words_of(k))); d = std::max(d, 1 + inside_words_of(id)) + 7; break; }
      case K_INSTANCE: d = 8 + meta_words_of(n.a); break;
      case K_DIFF: case K_BOUND: case K_INNERBOUND: d = 3 + std::max(std::max(meta_words_of(n.a), meta_words_of(n.b)), 1 + std::max(inside_words_of(n.a), inside_words_of(n.b))); break;
      case K_BIH: { for (auto& bn : n.bih->nodes) for (int k : bn.items) d = std::max(d, std::max(meta_words_of(k), 1 + inside_words_of(k))); d += 11 + n.bih->depth; break; }
      case K_TEX: case K_TAG: case K_NOSHADOW: case K_ONLYSHADOW: d = meta_words_of(n.a); break;
      default: break;
    }
    mwords_memo[id] = d;
    return d;
  }
  int vm_words_of(int id) const {
    auto it = words_memo.find(id);
    if (it != words_memo.end()) return it->second;
    const Node& n = G.at(id);
    int d = 0;
    switch (n.kind) {
      case K_LIST: { for (int k : n.kids) d = std::max(d, vm_words_of(k)); d += kVmListR; break; }
      case K_ISECT: { for (int k : n.kids) d = std::max(d, std::max(vm_words_of(k), 1 + inside_words_of(k))); d += 1 + kVmIsectChain * kVmIsectWords; break; }
      case K_INSTANCE: d = kVmInstR + vm_words_of(n.a); break;
      case K_DIFF: d = 1 + kVmDiffFixed + kCsgMaxAdvance + std::max(std::max(vm_words_of(n.a), vm_words_of(n.b)), 1 + std::max(std::max(inside_words_of(n.a), inside_words_of(n.b)), meta_words_of(n.a))); break;
      case K_BOUND: d = kVmBoundR + std::max(std::max(vm_words_of(n.a), vm_words_of(n.b)), 1 + inside_words_of(n.a)); break;
      case K_INNERBOUND: d = kVmIbR + std::max(vm_words_of(n.a), vm_words_of(n.b)); break;
      case K_BIH: d = kVmBihFixedR + 3 * n.bih->depth + bih_items_max(n, false); break;
      case K_TEX: case K_TAG: case K_NOSHADOW: case K_ONLYSHADOW: d = vm_words_of(n.a); break;
      default: break;  // primitives and a Mesh (its walk has a stack of its own) need no frame
    }
    words_memo[id] = d;
    return d;
  }

  // longest texture stack a hit can carry: the Tex wrappers on a path from `id` down to a primitive (a Mesh adds its
  // per-triangle texture, Mesh.hs:148-150).  The device stack holds kMaxTexDepth materials; a deeper one is refused at
  // commit -- the traversal would silently drop the outermost textures.
  int tex_cap = kMaxTexDepth;  // ids a TexStack holds in this scene
  mutable std::unordered_map<int, int> tex_memo;
  int tex_depth_of(int id) const {
    auto it = tex_memo.find(id);
    if (it != tex_memo.end()) return it->second;
    const Node& n = G.at(id);
    int d = 0;
    switch (n.kind) {
      case K_LIST: case K_ISECT: for (int k : n.kids) d = std::max(d, tex_depth_of(k)); break;
      case K_INSTANCE: case K_TAG: case K_NOSHADOW: case K_ONLYSHADOW: d = tex_depth_of(n.a); break;
      case K_DIFF: case K_BOUND: case K_INNERBOUND: d = std::max(tex_depth_of(n.a), tex_depth_of(n.b)); break;
      case K_BIH: for (auto& bn : n.bih->nodes) for (int k : bn.items) d = std::max(d, tex_depth_of(k)); break;
      case K_MESH: for (const MeshTri& t : n.mesh->tris) if (t.tex >= 0) { d = 1; break; } break;
      case K_TEX: d = tex_depth_of(n.a) + 1; break;
      default: break;
    }
    tex_memo[id] = d;
    return d;
  }

  // ---- primitive pools ----
  uint32_t emit_tri(const double* p) {  // (p1, e1, e2, n) evaluated in double, rounded once
    D3 p1{p[0], p[1], p[2]}, p2{p[3], p[4], p[5]}, p3{p[6], p[7], p[8]};
    D3 e1 = p2 - p1, e2 = p3 - p1;
    D3 n = normalize(cross(e1, e2));  // vnorm $ vcross e1 e2, Triangle.hs:73
    uint32_t idx = (uint32_t)(F.tris.size() / 3);
    F.tris.push_back(mk4(p1.x, p1.y, p1.z, n.x));
    F.tris.push_back(mk4(e1.x, e1.y, e1.z, n.y));
    F.tris.push_back(mk4(e2.x, e2.y, e2.z, n.z));
    return idx;
  }
  U4 emit_prim(const Node& n) {
    const double* p = n.p;
    U4 r{0, 0, 0, (uint32_t)n.uid};
    switch (n.kind) {
      case K_SPHERE: r.x = R_SPHERE; r.y = (uint32_t)F.spheres.size(); F.spheres.push_back(mk4(p[0], p[1], p[2], p[3])); break;
      case K_TRI: r.x = R_TRI; r.y = emit_tri(p); break;
      case K_TRIN: {  // self-contained 6-word block in the trinorms heap
        D3 p1{p[0], p[1], p[2]}, p2{p[3], p[4], p[5]}, p3{p[6], p[7], p[8]};
        D3 e1 = p2 - p1, e2 = p3 - p1;
        r.x = R_TRIN; r.y = (uint32_t)F.trinorms.size();
        F.trinorms.push_back(mk4(p1.x, p1.y, p1.z, 0)); F.trinorms.push_back(mk4(e1.x, e1.y, e1.z, 0)); F.trinorms.push_back(mk4(e2.x, e2.y, e2.z, 0));
        for (int k = 0; k < 3; k++) F.trinorms.push_back(mk4(p[9 + 3 * k], p[10 + 3 * k], p[11 + 3 * k], 0));
        F.n_other_prims++;
        break;
      }
      case K_BOX: r.x = R_BOX; r.y = (uint32_t)(F.boxes.size() / 2); F.boxes.push_back(mk4(p[0], p[1], p[2], 0)); F.boxes.push_back(mk4(p[3], p[4], p[5], 0)); F.n_other_prims++; break;
      case K_PLANE: r.x = R_PLANE; r.y = (uint32_t)F.planes.size(); F.planes.push_back(mk4(p[0], p[1], p[2], p[3])); F.n_other_prims++; break;
      case K_DISC: r.x = R_DISC; r.y = (uint32_t)(F.discs.size() / 2); F.discs.push_back(mk4(p[0], p[1], p[2], p[6])); F.discs.push_back(mk4(p[3], p[4], p[5], 0)); F.n_other_prims++; break;
      case K_CYL: r.x = R_CYL; r.y = (uint32_t)F.quadrics.size(); F.quadrics.push_back(mk4(p[0], p[1], p[2], 0)); F.n_other_prims++; break;
      case K_CONE: r.x = R_CONE; r.y = (uint32_t)F.quadrics.size(); F.quadrics.push_back(mk4(p[0], p[1], p[2], p[3])); F.n_other_prims++; break;
      default: break;
    }
    return r;
  }

  // Peel Tex / Tag / NoShadow / OnlyShadow wrappers off a node.  Up to two Tex levels fold into a primitive's
  // own stack (innermost first, id+1, 16 bits each); flags fold into the record.
  struct Peeled { int id; uint32_t flags; uint32_t own; int ntex; };
  Peeled peel(int id) const {
    Peeled p{id, 0, 0, 0};
    std::vector<int> texs;  // outermost first
    for (;;) {
      const Node& n = G.at(p.id);
      if (n.kind == K_TEX) { texs.push_back(n.mat); p.id = n.a; }
      else if (n.kind == K_TAG) p.id = n.a;
      else if (n.kind == K_NOSHADOW) { p.flags |= RF_NOSHADOW; p.id = n.a; }
      else if (n.kind == K_ONLYSHADOW) { p.flags |= RF_NOVIS; p.id = n.a; }
      else break;
    }
    p.ntex = (int)texs.size();
    // head of the stack = innermost Tex = last in `texs`
    if (p.ntex <= 2) for (int k = 0; k < p.ntex; k++) p.own |= (uint32_t)(texs[p.ntex - 1 - k] + 1) << (16 * k);
    return p;
  }

  // "CSG over primitives": what the flat tier evaluates without recursion (rt_device.hpp csg_item_rayint) -- a primitive, a
  // Difference or Intersection whose operands are primitives, or an Instance of one of those (how `cylinder` / `cone` and
  // TestScene.hs's transformed CSG shapes arrive); Tex / Tag / shadow-flag wrappers anywhere in between.
  bool prim_under_wrappers(int id) const { return is_prim(G.at(peel(id).id).kind); }
  bool csg_core(int id) const {
    const Node& c = G.at(peel(id).id);
    if (c.kind == K_DIFF) return prim_under_wrappers(c.a) && prim_under_wrappers(c.b);
    if (c.kind == K_ISECT) {
      // (csg_isect's frames: one per list position it may nest through, and a few for advances before they fold in place; a longer
      // Intersection is the generic tier's, whose frames are sized per scene)
      if ((int)c.kids.size() + 8 > kIsectFrames) return false;
      for (int k : c.kids) if (!prim_under_wrappers(k)) return false;
      return true;
    }
    return false;
  }
  bool csg_simple(int id) const {
    const Node& c = G.at(peel(id).id);
    if (is_prim(c.kind)) return true;
    if (c.kind == K_INSTANCE) return prim_under_wrappers(c.a) || csg_core(c.a);
    return csg_core(id);
  }

  // emit(id) -> the record VALUE for a reference to node id (children are written into F.recs / pools)
  U4 emit(int id) {
    auto it = memo.find(id);
    if (it != memo.end()) return it->second;
    const Node& n = G.at(id);
    U4 r{R_VOID, 0, 0, (uint32_t)n.uid};
    switch (n.kind) {
      case K_VOID: break;
      case K_SPHERE: case K_TRI: case K_TRIN: case K_BOX: case K_PLANE: case K_DISC: case K_CYL: case K_CONE: r = emit_prim(n); break;
      case K_TEX: case K_TAG: case K_NOSHADOW: case K_ONLYSHADOW: {
        Peeled p = peel(id);
        const Node& c = G.at(p.id);
        if (is_prim(c.kind) && p.ntex <= 2) {  // fold into the primitive's record
          r = emit(p.id);
          r.x |= p.flags; r.z = p.own;
        } else if (n.kind == K_TEX) {  // explicit Tex record over the child
          U4 child = emit(n.a);
          r.x = R_TEX; r.y = slot(child); r.z = (uint32_t)n.mat;
        } else {  // Tag / NoShadow / OnlyShadow over a composite: flags on a copy of the child's record
          r = emit(n.a);
          if (n.kind == K_NOSHADOW) r.x |= RF_NOSHADOW;
          if (n.kind == K_ONLYSHADOW) r.x |= RF_NOVIS;
        }
        break;
      }
      case K_LIST: case K_ISECT: {
        std::vector<U4> kids;
        for (int k : n.kids) kids.push_back(emit(k));
        r.x = (n.kind == K_LIST) ? R_LIST : R_ISECT;
        if (!kids.empty()) {  // (a list or an Intersection of primitives: the generic tier answers those in place)
          bool prims = true;
          for (const U4& kr : kids) { const uint32_t kk = kr.x & RF_KINDMASK; prims = prims && kk >= R_SPHERE && kk <= R_CONE; }
          if (prims && (n.kind == K_LIST || (int)kids.size() + 8 <= kIsectFrames)) r.x |= RF_PRIMLIST;  // (csg_isect's frames: see csg_core)
        }
        r.y = (uint32_t)F.recs.size(); r.z = (uint32_t)kids.size();
        F.recs.insert(F.recs.end(), kids.begin(), kids.end());
        break;
      }
      case K_INSTANCE: {
        U4 child = emit(n.a);
        r.x = R_INSTANCE; r.y = slot(child); r.z = (uint32_t)(F.xfms.size() / 6);
        for (int k = 0; k < 3; k++) F.xfms.push_back(mk4(n.xf.f.m[4 * k], n.xf.f.m[4 * k + 1], n.xf.f.m[4 * k + 2], n.xf.f.m[4 * k + 3]));
        for (int k = 0; k < 3; k++) F.xfms.push_back(mk4(n.xf.i.m[4 * k], n.xf.i.m[4 * k + 1], n.xf.i.m[4 * k + 2], n.xf.i.m[4 * k + 3]));
        break;
      }
      case K_DIFF: case K_BOUND: case K_INNERBOUND: {
        U4 a = emit(n.a), b = emit(n.b);
        r.x = n.kind == K_DIFF ? R_DIFF : (n.kind == K_BOUND ? R_BOUND : R_INNERBOUND);
        if (n.kind == K_DIFF) {  // (a Difference of two primitives: answered in place by the generic tier)
          const uint32_t ka = a.x & RF_KINDMASK, kb = b.x & RF_KINDMASK;
          if (ka >= R_SPHERE && ka <= R_CONE && kb >= R_SPHERE && kb <= R_CONE) r.x |= RF_PRIMLIST;
          if (n.retex) r.x |= RF_RETEX;
        }
        r.y = slot(a); r.z = slot(b);
        break;
      }
      case K_BIH: r = emit_bih(n); break;
      case K_MESH: r = emit_mesh(n); break;
    }
    memo[id] = r;
    return r;
  }
  uint32_t slot(const U4& v) { F.recs.push_back(v); return (uint32_t)F.recs.size() - 1; }

  // An item the interpreter answers in place whatever the ray (rt_generic.hpp: vm_resolve_r / _s return 0 or 1, or vm_inst_prim_hit /
  // vm_inst_prim_shadow take it): a primitive, or an Instance of a primitive or of a list of primitives, under Tex wrappers on either
  // side.  A BIH made of such items alone is walked as a packet by the interpreter's service (bih_items_wave).
  bool item_in_place(U4 c) const {
    while ((c.x & RF_KINDMASK) == R_TEX) c = F.recs[c.y];
    uint32_t k = c.x & RF_KINDMASK;
    if (k >= R_SPHERE && k <= R_CONE) return true;
    if (k != R_INSTANCE) return false;
    c = F.recs[c.y];
    while ((c.x & RF_KINDMASK) == R_TEX) c = F.recs[c.y];
    k = c.x & RF_KINDMASK;
    return (k >= R_SPHERE && k <= R_CONE) || (k == R_LIST && (c.x & RF_PRIMLIST) != 0);
  }

  // BIH: nodes in preorder; leaf items become consecutive records, and (for homogeneous leaves) consecutive pool
  // entries, so a leaf is one contiguous run of 48-byte triangles / 16-byte spheres.
  U4 emit_bih(const Node& n) {
    const BihTree& T = *n.bih;
    F.max_bih_depth = std::max(F.max_bih_depth, T.depth);
    // classify
    bool all_tri = true, all_sph = true, all_simple = true, all_csg = true;
    for (auto& bn : T.nodes) for (int it : bn.items) {
      Peeled p = peel(it);
      const Node& c = G.at(p.id);
      bool simple = is_prim(c.kind) && p.ntex <= 2;
      all_simple &= simple;
      all_tri &= simple && c.kind == K_TRI && p.flags == 0;
      all_sph &= simple && c.kind == K_SPHERE && p.flags == 0;
      all_csg &= csg_simple(it);
    }
    uint32_t cls = all_tri ? BC_TRI : (all_sph ? BC_SPHERE : (all_simple ? BC_SIMPLE : (all_csg ? BC_CSG : BC_GENERIC)));
    uint32_t hdr = (uint32_t)(F.bihhdr.size() / 3);
    F.bihhdr.resize(F.bihhdr.size() + 3);  // reserved now: items of a generic BIH may emit nested BIHs before we fill it
    // Node slots: only branches (and leaves of 7+ items, whose extent lives in a slot) get one.  Branches are laid out
    // in small treelets -- a node, then its branch children, then a grandchild, four 16-byte nodes to a 64-byte cache
    // line -- and the treelets depth first, so the step after a fetch usually finds its node in the line just fetched
    // and a subtree is contiguous.  References are explicit, so the traversal does not care about the order.
    std::vector<uint32_t> slot(T.nodes.size(), 0xffffffffu);
    uint32_t base = (uint32_t)F.bihnodes.size();
    base = (base + 3u) & ~3u;  // line-align this tree's first treelet
    uint32_t nslots = 0;
    {
      std::vector<int> roots;  // treelet roots still to place (a stack: depth first)
      if (!T.nodes.empty() && !T.nodes[0].leaf) roots.push_back(0);
      while (!roots.empty()) {
        int x = roots.back(); roots.pop_back();
        int group[4]; int ng = 0;
        group[ng++] = x;
        for (int q = 0; q < ng && ng < 4; q++) {  // breadth first inside the treelet
          const BihTree::Node& b = T.nodes[group[q]];
          if (!T.nodes[b.left].leaf && ng < 4) group[ng++] = b.left;
          if (!T.nodes[b.right].leaf && ng < 4) group[ng++] = b.right;
        }
        // a treelet never straddles a cache line: one that would is started on the next line (the skipped slots stay unused)
        if ((nslots & 3u) + (uint32_t)ng > 4u) nslots = (nslots + 3u) & ~3u;
        for (int q = 0; q < ng; q++) slot[group[q]] = nslots++;
        // branch children of the group's members that did not fit start treelets of their own; right pushed first so the
        // left subtree is laid out next
        for (int q = ng - 1; q >= 0; q--) {
          const BihTree::Node& b = T.nodes[group[q]];
          if (!T.nodes[b.right].leaf && slot[b.right] == 0xffffffffu) roots.push_back(b.right);
          if (!T.nodes[b.left].leaf && slot[b.left] == 0xffffffffu) roots.push_back(b.left);
        }
      }
      for (size_t k = 0; k < T.nodes.size(); k++) if (T.nodes[k].leaf && T.nodes[k].items.size() > 6) slot[k] = nslots++;
    }
    F.bihnodes.resize(base + std::max<uint32_t>(nslots, 1u));
    if (base + nslots >= BREF_FIRST_LIMIT) throw limit_error("too many BIH nodes");
    bool pk = cls == BC_TRI && base + nslots < (1u << 27);  // (a node's byte offset is a reference: 31 bits)
    F.pknodes.resize(F.bihnodes.size(), F4{0, 0, 0, 0});
    // pass 1: leaves -- emit the items fresh (no memo) so records and pool entries are consecutive, and build the
    // child reference that describes each leaf (rt_device.hpp: BREF_*)
    std::vector<uint32_t> ref(T.nodes.size(), 0);
    std::vector<uint32_t> pkleaf(T.nodes.size(), 3u);  // a leaf as the packet walk refers to it: byte offset of its first pair record | 3
    uint32_t delta = 0;
    bool have_delta = false;
    bool in_place = cls != BC_TRI && cls != BC_SPHERE;  // (those two have their own packet walk)
    for (size_t k = 0; k < T.nodes.size(); k++) {
      const BihTree::Node& bn = T.nodes[k];
      if (!bn.leaf) { ref[k] = base + slot[k]; continue; }
      std::vector<U4> items;
      uint32_t first_prim = 0;
      for (size_t q = 0; q < bn.items.size(); q++) {
        int it = bn.items[q];
        U4 rec;
        if (cls == BC_GENERIC || cls == BC_CSG) rec = emit(it);
        else {
          Peeled p = peel(it);
          rec = emit_prim(G.at(p.id));
          rec.x |= p.flags; rec.z = p.own;
        }
        if (q == 0) first_prim = rec.y;
        in_place = in_place && item_in_place(rec);
        items.push_back(rec);
      }
      uint32_t first_rec = (uint32_t)F.recs.size();
      F.recs.insert(F.recs.end(), items.begin(), items.end());
      if (first_rec + items.size() >= BREF_FIRST_LIMIT) throw limit_error("too many records for the BIH leaf references");
      uint32_t count = (uint32_t)items.size();
      if (count && (cls == BC_TRI || cls == BC_SPHERE)) {
        uint32_t dl = first_prim - first_rec;  // both are emitted in leaf order, so this is one constant per BIH
        if (have_delta && dl != delta) throw scene_error("internal: BIH leaf pools are not contiguous");
        delta = dl; have_delta = true;
      }
      if (cls == BC_TRI && count && pk) {
        if (((uint64_t)F.tripairs.size() + (uint64_t)count * kPairWords) * 4 >= (1ull << 31)) pk = false;  // (a pair record's byte offset is a reference)
        else pkleaf[k] = emit_pairs(first_prim, count, first_rec) | 3u;
      }
      if (count == 0) ref[k] = BREF_LEAF_BIT;
      else if (count <= 6) ref[k] = BREF_LEAF_BIT | (count << 26) | first_rec;
      else {
        F.bihnodes[base + slot[k]] = F4{0.0f, 0.0f, as_float_bits(count), as_float_bits(first_rec)}; ref[k] = BREF_LEAF_BIT | (7u << 26) | (base + slot[k]);
      }
    }
    // pass 2: branches
    for (size_t k = 0; k < T.nodes.size(); k++) {
      const BihTree::Node& bn = T.nodes[k];
      if (bn.leaf) continue;
      // A child that is an empty leaf (a quarter of the leaves the reference builder makes) gets a plane at -inf / +inf:
      // the interval tests of the traversal (`near < t1`, `t2 < far`) then fail for it whatever the ray, so the device
      // never enters it and needs no test for it.  The other child's interval does not depend on this plane.
      const float inf = std::numeric_limits<float>::infinity();
      float ls = ref[bn.left] == BREF_LEAF_BIT ? -inf : round_up(bn.lsplit), rs = ref[bn.right] == BREF_LEAF_BIT ? inf : round_down(bn.rsplit);
      F.bihnodes[base + slot[k]] = F4{ls, rs, as_float_bits((uint32_t)bn.axis | (ref[bn.left] << 2)), as_float_bits(ref[bn.right])};
      // The hand-written packet walk (bih_packet_asm.hpp) reads its own copy: a branch child is referred to by its BYTE OFFSET
      // in this pool with the child's axis in the two low bits (scalar loads ignore them), a leaf child by the byte offset of
      // its first pair record (tripairs) with both low bits set -- the step's code is picked, the near child's node asked for
      // and a leaf's triangles fetched without a shift, a mask or a multiplication.  (An empty leaf is 3: never entered.)
      if (pk) {
        auto pkref = [&](int c) { return T.nodes[c].leaf ? pkleaf[c] : (((base + slot[c]) << 4) | (uint32_t)T.nodes[c].axis); };
        F.pknodes[base + slot[k]] = F4{ls, rs, as_float_bits(pkref(bn.left)), as_float_bits(pkref(bn.right))};
      }
    }
    F.bihhdr[3 * hdr] = mk4u(round_down(T.bb.lo.x), round_down(T.bb.lo.y), round_down(T.bb.lo.z), ref[0]);
    F.bihhdr[3 * hdr + 1] = mk4u(round_up(T.bb.hi.x), round_up(T.bb.hi.y), round_up(T.bb.hi.z), cls);
    if (cls == BC_TRI) F.pk_all = F.pk_all && pk;
    // (y: the root in the packet walk's form, z: 1 when this tree has that form -- triangle leaves, byte offsets that fit a reference)
    const uint32_t pkroot = (pk && !T.nodes.empty() && !T.nodes[0].leaf) ? (((base + slot[0]) << 4) | (uint32_t)T.nodes[0].axis) : 0u;
    // (w: the tree's depth -- the generic tier walks a tree as a packet when its per-wave stack holds it -- and, in bit 31, whether
    // every item is answered in place: kBihItemsInPlace)
    in_place = in_place && !T.nodes.empty() && !T.nodes[0].leaf;
    if (getenv("GLOME_DEBUG_NO_ITEM_PACKETS")) in_place = false;  // (A/B switch: such trees walked lane by lane over frames, as until round 4)
    F.bihhdr[3 * hdr + 2] = F4{as_float_bits(delta), as_float_bits(pkroot), as_float_bits(pk ? 1u : 0u), as_float_bits((uint32_t)T.depth | (in_place ? kBihItemsInPlace : 0u))};
    if (cls == BC_SPHERE || cls == BC_TRI || in_place) F.max_sphere_bih_depth = std::max(F.max_sphere_bih_depth, T.depth);
    return U4{R_BIH, hdr, 0, (uint32_t)n.uid};
  }

  U4 emit_mesh(const Node& n) {
    const MeshData& M = *n.mesh;
    F.max_mesh_depth = std::max(F.max_mesh_depth, M.depth);
    uint32_t hdr = (uint32_t)(F.meshhdr.size() / 2);
    uint32_t nbase = (uint32_t)(F.meshnodes.size() / 4);
    // node index remap: branches only (leaves are encoded in the parent's ref)
    std::vector<uint32_t> bidx(M.nodes.size(), 0);
    uint32_t nb = 0;
    for (size_t k = 0; k < M.nodes.size(); k++) if (!M.nodes[k].leaf) bidx[k] = nbase + nb++;
    F.meshnodes.resize((size_t)4 * (nbase + nb));
    std::vector<uint32_t> ref(M.nodes.size(), 0);
    // leaves first (preorder), so triangle runs follow traversal order
    for (size_t k = 0; k < M.nodes.size(); k++) {
      const MeshData::Node& mn = M.nodes[k];
      if (!mn.leaf) { ref[k] = bidx[k]; continue; }
      uint32_t first = (uint32_t)(F.mtris.size() / 3), count = (uint32_t)mn.tris.size();
      if (first + count >= (1u << 27)) throw limit_error("mesh has too many triangles");
      for (size_t q = 0; q < mn.tris.size(); q++) {
        const MeshTri& t = M.tris[mn.tris[q]];
        const D3 &a = M.verts[t.a], &b = M.verts[t.b], &c = M.verts[t.c];
        D3 e1 = b - a, e2 = c - a, nn = normalize(cross(e1, e2));
        F.mtris.push_back(mk4(a.x, a.y, a.z, nn.x));
        F.mtris.push_back(mk4(e1.x, e1.y, e1.z, nn.y));
        F.mtris.push_back(mk4(e2.x, e2.y, e2.z, nn.z));
        U4 meta{0, 0, q == 0 ? count : 0u, 0};
        if (t.na != -1) {
          meta.x = (uint32_t)F.trinorms.size() + 1;
          for (int ni : {t.na, t.nb, t.nc}) F.trinorms.push_back(mk4(M.norms[ni].x, M.norms[ni].y, M.norms[ni].z, 0));
        }
        if (t.tex != -1) meta.y = (uint32_t)M.mats[t.tex] + 1;
        F.mtrimeta.push_back(meta);
      }
      if (count == 0) { F.mtrimeta.push_back(U4{0, 0, 0, 0}); for (int q = 0; q < 3; q++) F.mtris.push_back(F4{0, 0, 0, 0}); }  // keep `first` addressable
      ref[k] = 0x80000000u | ((count >= 15 ? 15u : count) << 27) | first;
    }
    for (size_t k = 0; k < M.nodes.size(); k++) {
      const MeshData::Node& mn = M.nodes[k];
      if (mn.leaf) continue;
      F4* o = &F.meshnodes[(size_t)4 * bidx[k]];
      // boxes rounded outward
      o[0] = mk4u(round_down(mn.lbb.lo.x), round_down(mn.lbb.lo.y), round_down(mn.lbb.lo.z), ref[mn.left]);
      o[1] = mk4(round_up(mn.lbb.hi.x), round_up(mn.lbb.hi.y), round_up(mn.lbb.hi.z), 0);
      o[2] = mk4u(round_down(mn.rbb.lo.x), round_down(mn.rbb.lo.y), round_down(mn.rbb.lo.z), ref[mn.right]);
      o[3] = mk4(round_up(mn.rbb.hi.x), round_up(mn.rbb.hi.y), round_up(mn.rbb.hi.z), 0);
    }
    F.meshhdr.push_back(mk4u(round_down(M.bb.lo.x), round_down(M.bb.lo.y), round_down(M.bb.lo.z), ref[0]));
    F.meshhdr.push_back(mk4(round_up(M.bb.hi.x), round_up(M.bb.hi.y), round_up(M.bb.hi.z), 0));
    return U4{R_MESH, hdr, 0, (uint32_t)n.uid};
  }

  // ---- flat tier root program ----
  // Walk Tex / Tag / flag wrappers and lists from the root; every leaf of that walk must be a simple primitive,
  // a homogeneous BIH or a mesh.  `incoming` = texture stack pushed by the wrappers above the entry.
  void collect_entries(int id, uint32_t incoming, int nin, uint32_t flags) {
    if (F.tier != 0) return;
    const Node& n = G.at(id);
    switch (n.kind) {
      case K_VOID: return;
      case K_TAG: collect_entries(n.a, incoming, nin, flags); return;
      case K_NOSHADOW: collect_entries(n.a, incoming, nin, flags | RF_NOSHADOW); return;
      case K_ONLYSHADOW: collect_entries(n.a, incoming, nin, flags | RF_NOVIS); return;
      case K_TEX:
        if (nin >= 2) { F.tier = 1; F.why_generic = "more than two Tex levels above a root entry"; return; }
        // `tex:texs`: the inner Tex is pushed later, so it sits in front
        collect_entries(n.a, (incoming << 16) | (uint32_t)(n.mat + 1), nin + 1, flags);
        return;
      case K_LIST: for (int k : n.kids) collect_entries(k, incoming, nin, flags); return;
      case K_BIH: {
        U4 r = emit(id);
        uint32_t cls;
        std::memcpy(&cls, &F.bihhdr[3 * r.y + 1].w, 4);
        if (cls == BC_GENERIC) { F.tier = 1; F.why_generic = "BIH with composite items"; return; }
        F.entries.push_back(U4{slot(r), incoming, flags, 0});
        return;
      }
      case K_MESH: F.entries.push_back(U4{slot(emit(id)), incoming, flags, 0}); return;
      default:
        if (is_prim(n.kind) && n.kind != K_CYL && n.kind != K_CONE) { F.entries.push_back(U4{slot(emit(id)), incoming, flags, 0}); return; }
        if (csg_simple(id)) { F.entries.push_back(U4{slot(emit(id)), incoming, flags, 0}); return; }  // a Difference / Intersection / Instance over primitives
        F.tier = 1; F.why_generic = std::string("root reaches a ") + kind_name(n.kind);
        return;
    }
  }
};

}  // namespace glome
