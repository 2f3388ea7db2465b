// host_graph.hpp -- host side of the drop-in boundary: glome's scene constructors as an arena of
// typed nodes (double precision, like glome's `Flt = Double`), plus the scene-build-time pieces of
// the hot path: bounds, group/transform/flatten rewrites, the BIH builder and the Mesh BVH builder.
// Nothing here intersects rays: that is the job of the HIP kernels (rt_device.hpp).
// Reference files are cited as file:line relative to the reference tree; Qn = SURVEY.md Appendix A.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace glome {

struct scene_error : std::runtime_error { using std::runtime_error::runtime_error; };
struct limit_error : std::runtime_error { using std::runtime_error::runtime_error; };

constexpr double kInfinity = 1000000.0;  // Vec.hs:14
constexpr double kDelta = 0.0001;        // Vec.hs:40

struct D3 { double x = 0, y = 0, z = 0; };
inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 operator*(D3 a, double f) { return {a.x * f, a.y * f, a.z * f}; }
inline double dot(D3 a, D3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
inline D3 cross(D3 a, D3 b) { return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)}; }
inline double comp(D3 a, int ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }
inline void setcomp(D3& a, int ax, double v) { (ax == 0 ? a.x : (ax == 1 ? a.y : a.z)) = v; }
inline D3 normalize(D3 a) {  // vnorm, Vec.hs:314-317
  double inv = 1.0 / std::sqrt((a.x * a.x) + (a.y * a.y) + (a.z * a.z));
  return {a.x * inv, a.y * inv, a.z * inv};
}
inline double gmin(double a, double b) { return a > b ? b : a; }  // fmin, Vec.hs:44-45
inline double gmax(double a, double b) { return a > b ? a : b; }  // fmax, Vec.hs:48-49
inline bool about_equal(double a, double b) {                     // Vec.hs:96-102
  if (a > 1) return std::fabs(1 - (a / b)) < (kDelta * 10);
  return std::fabs(a - b) < (kDelta * 10);
}

// ---- transforms (Vec.hs:407-629) ----
struct Mat34 { double m[12]; };
struct Xf { Mat34 f, i; };
inline Mat34 mmul(const Mat34& A, const Mat34& B) {  // mat_mult, Vec.hs:426-443
  Mat34 o;
  for (int r = 0; r < 3; r++) {
    const double* a = A.m + 4 * r;
    o.m[4 * r + 0] = a[0] * B.m[0] + a[1] * B.m[4] + a[2] * B.m[8];
    o.m[4 * r + 1] = a[0] * B.m[1] + a[1] * B.m[5] + a[2] * B.m[9];
    o.m[4 * r + 2] = a[0] * B.m[2] + a[1] * B.m[6] + a[2] * B.m[10];
    o.m[4 * r + 3] = a[0] * B.m[3] + a[1] * B.m[7] + a[2] * B.m[11] + a[3];
  }
  return o;
}
inline Xf xf_ident() { return {{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}}, {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}}}; }
inline Xf xf_check(const Xf& x) {  // check_xfm, Vec.hs:466-477
  Mat34 p = mmul(x.f, x.i);
  static const double id[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int k = 0; k < 12; k++)
    if (!about_equal(p.m[k], id[k])) throw scene_error("corrupt matrix (forward * inverse is not the identity)");
  return x;
}
inline Xf xf_compose(const std::vector<Xf>& xs) {  // compose, Vec.hs:461-462: the first transform applies first
  Xf acc = xf_ident();
  for (const Xf& x : xs) acc = Xf{mmul(x.f, acc.f), mmul(acc.i, x.i)};  // xfm_mult x acc, Vec.hs:447-449
  return xf_check(acc);
}
inline D3 xf_point(const Xf& x, D3 v) {  // xfm_point, Vec.hs:502-509
  const double* m = x.f.m;
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3], m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7], m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11]};
}
inline D3 xf_vec(const Xf& x, D3 v) {  // xfm_vec, Vec.hs:522-529
  const double* m = x.f.m;
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
inline Xf xf_translate(D3 v) {  // Vec.hs:564-567
  return xf_check(Xf{{{1, 0, 0, v.x, 0, 1, 0, v.y, 0, 0, 1, v.z}}, {{1, 0, 0, -v.x, 0, 1, 0, -v.y, 0, 0, 1, -v.z}}});
}
inline Xf xf_scale(D3 v) {  // Vec.hs:571-574
  return xf_check(Xf{{{v.x, 0, 0, 0, 0, v.y, 0, 0, 0, 0, v.z, 0}}, {{1 / v.x, 0, 0, 0, 0, 1 / v.y, 0, 0, 0, 0, 1 / v.z, 0}}});
}
inline Xf xf_rotate(D3 v, double angle) {  // Vec.hs:577-598
  if (!about_equal(std::sqrt(dot(v, v)), 1)) throw scene_error("please use a normalized vector for rotation");
  double x = v.x, y = v.y, z = v.z, s = std::sin(angle), c = std::cos(angle);
  double m00 = ((x * x) + ((1 - (x * x)) * c)), m01 = (((x * y) * (1 - c)) - (z * s)), m02 = ((x * z * (1 - c)) + (y * s));
  double m10 = (((x * y) * (1 - c)) + (z * s)), m11 = ((y * y) + ((1 - (y * y)) * c)), m12 = ((y * z * (1 - c)) - (x * s));
  double m20 = ((x * z * (1 - c)) - (y * s)), m21 = ((y * z * (1 - c)) + (x * s)), m22 = ((z * z) + ((1 - (z * z)) * c));
  return xf_check(Xf{{{m00, m01, m02, 0, m10, m11, m12, 0, m20, m21, m22, 0}}, {{m00, m10, m20, 0, m01, m11, m21, 0, m02, m12, m22, 0}}});
}
inline Xf xf_xyz_to_uvw(D3 u, D3 v, D3 w) {  // Vec.hs:602-622
  if (!about_equal(dot(u, u), 1)) throw scene_error("unnormalized u");
  if (!about_equal(dot(v, v), 1)) throw scene_error("unnormalized v");
  if (!about_equal(dot(w, w), 1)) throw scene_error("unnormalized w");
  if (!(about_equal(dot(u, v), 0) && about_equal(dot(u, w), 0) && about_equal(dot(v, w), 0))) throw scene_error("vectors aren't orthogonal");
  return xf_check(Xf{{{u.x, v.x, w.x, 0, u.y, v.y, w.y, 0, u.z, v.z, w.z, 0}}, {{u.x, u.y, u.z, 0, v.x, v.y, v.z, 0, w.x, w.y, w.z, 0}}});
}
inline void orth(D3 v1, D3& v2, D3& v3) {  // Vec.hs:366-378
  if (!about_equal(dot(v1, v1), 1)) throw scene_error("orth: unnormalized vector");
  double dvx = dot(v1, D3{1, 0, 0});
  v2 = (dvx < 0.8 && dvx > -0.8) ? normalize(cross(v1, D3{1, 0, 0})) : normalize(cross(v1, D3{0, 1, 0}));
  v3 = cross(v1, v2);
}

// ---- boxes (Vec.hs:646-715) ----
struct Box3 { D3 lo, hi; };
inline Box3 box_empty() { return {{kInfinity, kInfinity, kInfinity}, {-kInfinity, -kInfinity, -kInfinity}}; }
inline Box3 box_everything() { return {{-kInfinity, -kInfinity, -kInfinity}, {kInfinity, kInfinity, kInfinity}}; }
inline Box3 box_join(const Box3& a, const Box3& b) {
  return {{gmin(a.lo.x, b.lo.x), gmin(a.lo.y, b.lo.y), gmin(a.lo.z, b.lo.z)}, {gmax(a.hi.x, b.hi.x), gmax(a.hi.y, b.hi.y), gmax(a.hi.z, b.hi.z)}};
}
inline Box3 box_overlap(const Box3& a, const Box3& b) {
  return {{gmax(a.lo.x, b.lo.x), gmax(a.lo.y, b.lo.y), gmax(a.lo.z, b.lo.z)}, {gmin(a.hi.x, b.hi.x), gmin(a.hi.y, b.hi.y), gmin(a.hi.z, b.hi.z)}};
}
inline Box3 box_of_points(const D3* pts, size_t n) {  // bbpts, Vec.hs:676-690: right fold with a +-delta pad
  if (n == 0) return box_empty();
  const D3& l = pts[n - 1];
  Box3 b{{l.x - kDelta, l.y - kDelta, l.z - kDelta}, {l.x + kDelta, l.y + kDelta, l.z + kDelta}};
  for (size_t k = n - 1; k-- > 0;) {
    const D3& p = pts[k];
    b = {{gmin(p.x - kDelta, b.lo.x), gmin(p.y - kDelta, b.lo.y), gmin(p.z - kDelta, b.lo.z)},
         {gmax(p.x + kDelta, b.hi.x), gmax(p.y + kDelta, b.hi.y), gmax(p.z + kDelta, b.hi.z)}};
  }
  return b;
}
inline double box_area(const Box3& b) {  // bbsa, Vec.hs:694-697 (Prelude max: `max 0 v` = if 0 <= v then v else 0)
  D3 d = b.hi - b.lo;
  double v = 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
  return (0 <= v) ? v : 0;
}
inline D3 box_mid(const Box3& b) { return (b.lo + b.hi) * 0.5; }  // bbmid, Bih.hs:162

// ---- scene graph ----
enum Kind : int {
  K_VOID = 0, K_SPHERE, K_TRI, K_TRIN, K_BOX, K_PLANE, K_DISC, K_CYL, K_CONE,  // primitives
  K_LIST, K_INSTANCE, K_DIFF, K_ISECT, K_BOUND, K_INNERBOUND, K_BIH, K_MESH,  // composites
  K_TEX, K_TAG, K_NOSHADOW, K_ONLYSHADOW                                      // wrappers (Tex.hs)
};
inline const char* kind_name(int k) {
  static const char* n[] = {"Void", "Sphere", "Triangle", "TriangleNorm", "Box", "Plane", "Disc", "Cylinder", "Cone", "List", "Instance",
                            "Difference", "Intersection", "Bound", "InnerBound", "Bih", "Mesh", "Tex", "Tag", "NoShadow", "OnlyShadow"};
  return (k >= 0 && k <= K_ONLYSHADOW) ? n[k] : "?";
}

struct BihTree {  // Bih bb root, Bih.hs:51-57, as preorder arrays
  struct Node { bool leaf; double lsplit, rsplit; int axis; int left, right; std::vector<int> items; };
  Box3 bb;
  std::vector<Node> nodes;  // nodes[0] = root
  int depth = 0;
};
struct MeshTri { int a, b, c, na, nb, nc, tex, tag; };  // Tri, Mesh.hs:29
struct MeshData {                                       // Mesh, Mesh.hs:42
  std::vector<D3> verts, norms;
  std::vector<MeshTri> tris;
  std::vector<int> mats;
  Box3 bb;
  struct Node { bool leaf; Box3 lbb, rbb; int left, right; std::vector<int> tris; };
  std::vector<Node> nodes;  // nodes[0] = root
  int depth = 0;
};

struct Node {
  int kind = K_VOID;
  int uid = -1;       // id of the constructor call that made the primitive (reported as `prim` by rayint_batch)
  double p[18] = {0}; // primitive parameters
  std::vector<int> kids;
  int a = -1, b = -1; // children of binary composites / wrappers
  int mat = -1;       // K_TEX
  bool retex = false; // K_DIFF: Difference a b False (difference_retexture, Csg.hs:29-30)
  Xf xf = xf_ident(); // K_INSTANCE
  std::shared_ptr<BihTree> bih;
  std::shared_ptr<MeshData> mesh;
};

enum MatKind : int { MAT_SURFACE = 0, MAT_REFLECT, MAT_REFRACT, MAT_LAYERS, MAT_BLEND, MAT_WARP };
struct WarpLight { double pos[3], color[3], rad; bool shadow; };
struct Mat {  // Material, Shader.hs:43-52
  int kind = MAT_SURFACE;
  double color[3] = {0, 0, 0}, alpha = 1, amb = 0, kd = 0, ks = 0, shine = 0, refl = 0, refr = 0, ior = 1, weight = 0;
  std::vector<int> kids;
  int a = -1, b = -1;
  int wfn = 0;                 // Blend weight: 0 constant, else a solid texture function of the hit position (GLOME_WEIGHT_*)
  double wp[4] = {0, 0, 0, 0};
  // Warp frame scene' lights' xfm (Shader.hs:47-50): nodes of the frame and of the scene looked into (-1: the root the
  // scene is committed with -- the portal of TestScene.hs:152-181 looks into the scene it stands in), that scene's lights,
  // and the matrix of the closure's one shape in the reference, \ray hit -> xfm_ray M (Ray (pos hit) (vnorm (dir ray)))
  int wframe = -1, wscene = -1;
  std::vector<WarpLight> wlights;
  Xf wxf;
};

struct Graph {
  std::vector<Node> nodes;
  std::vector<Mat> mats;

  int add(Node n) {
    int id = (int)nodes.size();
    if (n.uid < 0) n.uid = id;
    nodes.push_back(std::move(n));
    return id;
  }
  const Node& at(int id) const {
    if (id < 0 || id >= (int)nodes.size()) throw std::invalid_argument("bad node id");
    return nodes[id];
  }
  int next_id() const { return (int)nodes.size(); }

  // ---------------- primitive constructors ----------------
  int sphere(D3 c, double r) {  // Sphere.hs:15-17
    Node n; n.kind = K_SPHERE; n.p[0] = c.x; n.p[1] = c.y; n.p[2] = c.z; n.p[3] = r;
    return add(n);
  }
  static Node tri_node(D3 a, D3 b, D3 c) {
    Node n; n.kind = K_TRI;
    double v[9] = {a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z};
    std::copy(v, v + 9, n.p);
    return n;
  }
  int triangle(D3 a, D3 b, D3 c) { return add(tri_node(a, b, c)); }  // Triangle.hs:18-20
  int trianglenorm(D3 a, D3 b, D3 c, D3 na, D3 nb, D3 nc) {          // Triangle.hs:34-35
    Node n = tri_node(a, b, c); n.kind = K_TRIN;
    double v[9] = {na.x, na.y, na.z, nb.x, nb.y, nb.z, nc.x, nc.y, nc.z};
    std::copy(v, v + 9, n.p + 9);
    return add(n);
  }
  int box(D3 a, D3 b) {  // Box.hs:12-15
    Node n; n.kind = K_BOX;
    n.p[0] = gmin(a.x, b.x); n.p[1] = gmin(a.y, b.y); n.p[2] = gmin(a.z, b.z);
    n.p[3] = gmax(a.x, b.x); n.p[4] = gmax(a.y, b.y); n.p[5] = gmax(a.z, b.z);
    return add(n);
  }
  int plane(D3 pt, D3 nrm) {  // Plane.hs:17-20
    D3 nn = normalize(nrm);
    return plane_offset(nn, dot(pt, nn));
  }
  int plane_offset(D3 nn, double off) {  // Plane.hs:24-25
    Node n; n.kind = K_PLANE; n.p[0] = nn.x; n.p[1] = nn.y; n.p[2] = nn.z; n.p[3] = off;
    return add(n);
  }
  int disc(D3 pos, D3 nrm, double r) {  // Cone.hs:29-31
    Node n; n.kind = K_DISC; n.p[0] = pos.x; n.p[1] = pos.y; n.p[2] = pos.z; n.p[3] = nrm.x; n.p[4] = nrm.y; n.p[5] = nrm.z; n.p[6] = r * r;
    return add(n);
  }
  // cylinder / cone: a z-axis canonical primitive wrapped in an Instance, Cone.hs:40-67 (Q7)
  int axis_instance(Node prim, D3 p1, D3 p2) {
    D3 axis = p2 - p1;
    double len = std::sqrt(dot(axis, axis));
    D3 ax1 = axis * (1 / len), ax2, ax3;
    orth(ax1, ax2, ax3);
    int uid = next_id() + 1;  // the Instance made below is what the caller sees
    prim.uid = uid;
    int pid = add(prim);
    Node inst; inst.kind = K_INSTANCE; inst.a = pid; inst.uid = uid;
    inst.xf = xf_compose({xf_xyz_to_uvw(ax2, ax3, ax1), xf_translate(p1)});
    return add(inst);
  }
  int cylinder(D3 p1, D3 p2, double r) {
    D3 axis = p2 - p1;
    Node n; n.kind = K_CYL; n.p[0] = r; n.p[1] = 0; n.p[2] = std::sqrt(dot(axis, axis));
    return axis_instance(n, p1, p2);
  }
  int cone(D3 p1, double r1, D3 p2, double r2) {
    if (r1 < r2) { std::swap(p1, p2); std::swap(r1, r2); }
    if (r1 - r2 < kDelta) return cylinder(p1, p2, r2);
    D3 axis = p2 - p1;
    double len = std::sqrt(dot(axis, axis));
    Node n; n.kind = K_CONE; n.p[0] = r1; n.p[1] = 0; n.p[2] = len; n.p[3] = (r1 * len) / (r1 - r2);
    return axis_instance(n, p1, p2);
  }

  // ---------------- composites ----------------
  void tolist(int id, std::vector<int>& out) const {  // Solid.hs:230, 333, 359 (Q22)
    const Node& n = at(id);
    if (n.kind == K_VOID) return;
    if (n.kind == K_LIST) { for (int k : n.kids) tolist(k, out); return; }
    out.push_back(id);
  }
  int make_list(const std::vector<int>& kids) { Node n; n.kind = K_LIST; n.kids = kids; return add(n); }
  int group(const std::vector<int>& ids) {  // Solid.hs:293-302
    for (int i : ids) at(i);
    if (ids.empty()) { Node n; n.kind = K_VOID; return add(n); }
    if (ids.size() == 1) return ids[0];
    std::vector<int> flat;
    for (int i : ids) tolist(i, flat);
    return make_list(flat);
  }
  int instance_of(int child, const Xf& x, int uid) { Node n; n.kind = K_INSTANCE; n.a = child; n.xf = x; n.uid = uid; return add(n); }
  // transform, Solid.hs:184,235 with the overrides for Void (:360), Instance (:494-496), Triangle (Triangle.hs:164-177)
  int transform(int id, const std::vector<Xf>& xs) {
    const Node n = at(id);
    switch (n.kind) {
      case K_VOID: return id;
      case K_INSTANCE: {
        std::vector<Xf> l{n.xf};
        l.insert(l.end(), xs.begin(), xs.end());
        return transform(n.a, {xf_compose(l)});
      }
      case K_TRI: case K_TRIN: {
        Xf x = xf_compose(xs);
        auto P = [&](int o) { return xf_point(x, D3{n.p[o], n.p[o + 1], n.p[o + 2]}); };
        Node t = tri_node(P(0), P(3), P(6));
        if (n.kind == K_TRIN) {
          t.kind = K_TRIN;
          for (int k = 0; k < 3; k++) {
            D3 v = normalize(xf_vec(x, D3{n.p[9 + 3 * k], n.p[10 + 3 * k], n.p[11 + 3 * k]}));
            t.p[9 + 3 * k] = v.x; t.p[10 + 3 * k] = v.y; t.p[11 + 3 * k] = v.z;
          }
        }
        return add(t);
      }
      default: return instance_of(id, xf_compose(xs), next_id());
    }
  }
  // transform_leaf, Solid.hs:240 with overrides: list (:334), Instance (:498-500), Bound / InnerBound (Bound.hs:69-71, 112)
  int transform_leaf(int id, const std::vector<Xf>& xs) {
    const Node n = at(id);
    switch (n.kind) {
      case K_LIST: {
        std::vector<int> flat, out;
        tolist(id, flat);
        for (int k : flat) out.push_back(transform_leaf(k, xs));
        return make_list(out);
      }
      case K_INSTANCE: {
        std::vector<Xf> l{n.xf};
        l.insert(l.end(), xs.begin(), xs.end());
        return transform_leaf(n.a, {xf_compose(l)});
      }
      case K_BOUND: case K_INNERBOUND: return transform_leaf(n.b, xs);
      default: return transform(id, xs);
    }
  }
  // flatten_transform, Solid.hs:246 with overrides: list (:335), Instance (:509-511), Bound (Bound.hs:73-74, 111)
  std::vector<int> flatten_transform(int id) {
    const Node n = at(id);
    switch (n.kind) {
      case K_LIST: {
        std::vector<int> out;
        for (int k : n.kids) out.push_back(flatten_transform_item(k));
        return out;
      }
      case K_INSTANCE: return {transform_leaf(n.a, {n.xf})};
      case K_BOUND: case K_INNERBOUND: return {flatten_transform_item(n.b)};
      default: { std::vector<int> out; tolist(id, out); return out; }
    }
  }
  int flatten_transform_item(int id) { return make_list(flatten_transform(id)); }  // Solid.hs:273
  int tolist_node(int id) { std::vector<int> out; tolist(id, out); return make_list(out); }

  int difference(int a, int b, bool retexture = false) { at(a); at(b); Node n; n.kind = K_DIFF; n.a = a; n.b = b; n.retex = retexture; return add(n); }  // Csg.hs:26-30
  int intersection(const std::vector<int>& ids) { for (int i : ids) at(i); Node n; n.kind = K_ISECT; n.kids = ids; return add(n); }  // Csg.hs:64-65
  int wrap(int kind, int id, int mat = -1) {
    at(id);
    if (kind == K_TEX && (mat < 0 || mat >= (int)mats.size())) throw std::invalid_argument("bad material id");
    Node n; n.kind = kind; n.a = id; n.mat = mat;
    return add(n);
  }
  int bound_object(int sa, int sb, bool inner) { at(sa); at(sb); Node n; n.kind = inner ? K_INNERBOUND : K_BOUND; n.a = sa; n.b = sb; return add(n); }

  // ---------------- bound, Solid.hs:171 ----------------
  Box3 bound(int id) const {
    const Node& n = at(id);
    const double* p = n.p;
    switch (n.kind) {
      case K_VOID: return box_empty();                                                               // Solid.hs:358
      case K_SPHERE: return {{p[0] - p[3], p[1] - p[3], p[2] - p[3]}, {p[0] + p[3], p[1] + p[3], p[2] + p[3]}};  // Sphere.hs:78-81
      case K_TRI: case K_TRIN:                                                                        // Triangle.hs:147-158
        return {{gmin(gmin(p[0], p[3]), p[6]) - kDelta, gmin(gmin(p[1], p[4]), p[7]) - kDelta, gmin(gmin(p[2], p[5]), p[8]) - kDelta},
                {gmax(gmax(p[0], p[3]), p[6]) + kDelta, gmax(gmax(p[1], p[4]), p[7]) + kDelta, gmax(gmax(p[2], p[5]), p[8]) + kDelta}};
      case K_BOX: return {{p[0], p[1], p[2]}, {p[3], p[4], p[5]}};
      case K_PLANE: return box_everything();                                                          // Plane.hs:40-41
      case K_DISC: { double r = std::sqrt(p[6]); return {{p[0] - r, p[1] - r, p[2] - r}, {p[0] + r, p[1] + r, p[2] + r}}; }  // Cone.hs:93-95
      case K_CYL: case K_CONE: return {{-p[0], -p[0], p[1]}, {p[0], p[0], p[2]}};                      // Cone.hs:145-147, 253-255
      case K_LIST: { Box3 b = box_empty(); for (int k : n.kids) b = box_join(b, bound(k)); return b; }  // Solid.hs:332
      case K_INSTANCE: {                                                                              // Solid.hs:477-484
        Box3 b = bound(n.a);
        D3 pts[8];
        int q = 0;
        for (double x : {b.lo.x, b.hi.x}) for (double y : {b.lo.y, b.hi.y}) for (double z : {b.lo.z, b.hi.z}) pts[q++] = xf_point(n.xf, D3{x, y, z});
        return box_of_points(pts, 8);
      }
      case K_DIFF: return bound(n.a);                                                                 // Csg.hs:113-114
      case K_ISECT: {                                                                                 // Csg.hs:116-120
        if (n.kids.empty()) return box_empty();
        Box3 b = box_everything();
        for (int k : n.kids) b = box_overlap(b, bound(k));
        return b;
      }
      case K_BOUND: return box_overlap(bound(n.a), bound(n.b));  // Bound.hs:61-62
      case K_INNERBOUND: return bound(n.b);                      // Bound.hs:110
      case K_BIH: return n.bih->bb;
      case K_MESH: return n.mesh->bb;
      default: return bound(n.a);  // Tex / Tag / NoShadow / OnlyShadow
    }
  }
  // primcount, Solid.hs:197,251 and overrides
  void primcount(int id, long o[3]) const {
    const Node& n = at(id);
    auto addk = [&](int k, bool asbound) {
      long t[3] = {0, 0, 0};
      primcount(k, t);
      if (asbound) { o[1] += t[1]; o[2] += t[0] + t[2]; } else { o[0] += t[0]; o[1] += t[1]; o[2] += t[2]; }
    };
    switch (n.kind) {
      case K_VOID: case K_SPHERE: case K_TRI: case K_TRIN: case K_BOX: case K_PLANE: case K_DISC: case K_CYL: case K_CONE: o[0] += 1; break;
      case K_LIST: case K_ISECT: for (int k : n.kids) addk(k, false); break;
      case K_INSTANCE: addk(n.a, false); o[1] += 1; break;
      case K_DIFF: addk(n.a, false); addk(n.b, false); break;
      case K_BOUND: case K_INNERBOUND: addk(n.a, true); addk(n.b, false); break;
      case K_BIH: for (auto& bn : n.bih->nodes) { if (bn.leaf) for (int k : bn.items) addk(k, false); else o[2] += 1; } o[2] += 1; break;  // Bih.hs:591-595
      case K_MESH: for (auto& mn : n.mesh->nodes) { if (mn.leaf) o[0] += (long)mn.tris.size(); else o[2] += 1; } break;                   // Mesh.hs:201-205
      default: addk(n.a, false);
    }
  }

  // ---------------- BIH builder, Bih.hs:211-285 (Q11) ----------------
  struct BihBuild {
    const std::vector<Box3>& boxes;  // per input object
    const std::vector<int>& ids;     // node ids of the objects
    BihTree& T;
    int rec(const std::vector<int>& objs, const Box3& bb, D3 mid, int depth) {
      int me = (int)T.nodes.size();
      T.nodes.push_back({});
      T.depth = std::max(T.depth, depth + 1);
      size_t objcount = objs.size();
      auto leaf = [&]() {
        BihTree::Node& n = T.nodes[me];
        n.leaf = true; n.lsplit = n.rsplit = 0; n.axis = -1; n.left = n.right = -1;
        for (int o : objs) n.items.push_back(ids[o]);
        return me;
      };
      if (objcount <= 3) return leaf();
      double sa = box_area(bb);
      // four candidate partitions: bbox-centre below the midpoint on x, y, z; and big-vs-small by surface area
      std::vector<int> L[4], R[4];
      for (int o : objs) {
        D3 m = box_mid(boxes[o]);
        (m.x < mid.x ? L[0] : R[0]).push_back(o);
        (m.y < mid.y ? L[1] : R[1]).push_back(o);
        (m.z < mid.z ? L[2] : R[2]).push_back(o);
        (box_area(boxes[o]) > sa * 0.4 ? L[3] : R[3]).push_back(o);
      }
      static const int AX[4] = {0, 1, 2, 0};  // the big/small split is stored as an axis-0 node (Bih.hs:231-232, 285)
      double lmax[4], rmin[4], cost[4];
      Box3 lbb[4], rbb[4];
      for (int k = 0; k < 4; k++) {
        lmax[k] = -kInfinity; rmin[k] = kInfinity;
        for (int o : L[k]) lmax[k] = gmax(lmax[k], comp(boxes[o].hi, AX[k]));
        for (int o : R[k]) rmin[k] = gmin(rmin[k], comp(boxes[o].lo, AX[k]));
        lbb[k] = bb; setcomp(lbb[k].hi, AX[k], lmax[k]);  // child boxes shrink along the split axis only (Bih.hs:243-250)
        rbb[k] = bb; setcomp(rbb[k].lo, AX[k], rmin[k]);
        cost[k] = ((box_area(lbb[k]) * double(L[k].size())) + (box_area(rbb[k]) * double(R[k].size()))) * (k < 3 ? 1.1 : 1.2);
      }
      double costorig = sa * double(objcount);
      if (costorig < cost[0] && costorig < cost[1] && costorig < cost[2] && costorig < cost[3]) return leaf();
      int k;
      if (cost[0] < cost[1] && cost[0] < cost[2] && cost[0] < cost[3]) k = 0;
      else if (cost[1] < cost[2] && cost[1] < cost[3]) k = 1;
      else if (cost[1] < cost[3]) k = 2;  // as written in the reference (`costy < costb`, Bih.hs:283)
      else k = 3;
      int l = rec(L[k], lbb[k], box_mid(lbb[k]), depth + 1);
      int r = rec(R[k], rbb[k], box_mid(rbb[k]), depth + 1);
      BihTree::Node& n = T.nodes[me];
      n.leaf = false; n.lsplit = lmax[k] + kDelta; n.rsplit = rmin[k] - kDelta; n.axis = AX[k]; n.left = l; n.right = r;
      return me;
    }
  };
  int bih(const std::vector<int>& ids) {  // Bih.hs:309-324
    if (ids.empty()) { Node n; n.kind = K_VOID; return add(n); }
    std::vector<Box3> boxes;
    Box3 bb = box_empty();
    for (int i : ids) boxes.push_back(bound(i));
    for (auto& b : boxes) bb = box_join(bb, b);
    if (bb.lo.x == -kInfinity || bb.lo.y == -kInfinity || bb.lo.z == -kInfinity || bb.hi.x == kInfinity || bb.hi.y == kInfinity || bb.hi.z == kInfinity)
      throw scene_error("bih: infinite bounding box");
    auto T = std::make_shared<BihTree>();
    T->bb = bb;
    std::vector<int> all(ids.size());
    for (size_t k = 0; k < all.size(); k++) all[k] = (int)k;
    BihBuild B{boxes, ids, *T};
    B.rec(all, bb, box_mid(bb), 0);
    Node n; n.kind = K_BIH; n.bih = T;
    return add(n);
  }

  // ---------------- Mesh builder, Mesh.hs:50-134 (Q12) ----------------
  struct MeshBuild {
    MeshData& M;
    std::vector<Box3> tbb;
    Box3 join(const std::vector<int>& ts) const { Box3 b = box_empty(); for (int t : ts) b = box_join(b, tbb[t]); return b; }
    int rec(const std::vector<int>& ts, const Box3& box, int depth) {
      int me = (int)M.nodes.size();
      M.nodes.push_back({});
      M.depth = std::max(M.depth, depth + 1);
      auto leaf = [&]() { MeshData::Node& n = M.nodes[me]; n.leaf = true; n.left = n.right = -1; n.tris = ts; return me; };
      size_t cnt = ts.size();
      if (cnt < 3) return leaf();
      D3 mid = box_mid(box);
      double sa = box_area(box);
      std::vector<int> L[4], R[4];
      for (int t : ts) {
        D3 m = box_mid(tbb[t]);
        (m.x < mid.x ? L[0] : R[0]).push_back(t);
        (m.y < mid.y ? L[1] : R[1]).push_back(t);
        (m.z < mid.z ? L[2] : R[2]).push_back(t);
        (box_area(tbb[t]) > sa * 0.4 ? L[3] : R[3]).push_back(t);
      }
      Box3 lbb[4], rbb[4];
      double cost[4];
      for (int k = 0; k < 4; k++) {
        lbb[k] = join(L[k]); rbb[k] = join(R[k]);  // true unions of triangle boxes (Mesh.hs:89-96)
        cost[k] = (box_area(lbb[k]) * double(L[k].size()) + box_area(rbb[k]) * double(R[k].size())) * 1.1;
      }
      double lcost = box_area(box) * double(cnt);
      if (lcost < cost[0] && lcost < cost[1] && lcost < cost[2] && lcost < cost[3]) return leaf();
      int k;
      if (cost[0] < cost[1] && cost[0] < cost[2] && cost[0] < cost[3]) k = 0;
      else if (cost[1] < cost[2] && cost[1] < cost[3]) k = 1;
      else if (cost[2] < cost[3]) k = 2;
      else k = 3;
      int l = rec(L[k], lbb[k], depth + 1);
      int r = rec(R[k], rbb[k], depth + 1);
      MeshData::Node& n = M.nodes[me];
      n.leaf = false; n.lbb = lbb[k]; n.rbb = rbb[k]; n.left = l; n.right = r;
      return me;
    }
  };
  // the validated arrays, the mesh's box and the triangles' boxes -- everything of `mesh` but the tree
  std::shared_ptr<MeshData> mesh_data(std::vector<D3> verts, std::vector<D3> norms, std::vector<MeshTri> tris, std::vector<int> mesh_mats, std::vector<Box3>& tbb) const {
    auto M = std::make_shared<MeshData>();
    int nv = (int)verts.size(), nn = (int)norms.size(), nm = (int)mesh_mats.size();
    for (auto& t : tris) {
      auto in = [](int i, int lim) { return i >= 0 && i < lim; };
      if (!in(t.a, nv) || !in(t.b, nv) || !in(t.c, nv)) throw std::invalid_argument("mesh: vertex index out of range");
      if (t.na != -1 && (!in(t.na, nn) || !in(t.nb, nn) || !in(t.nc, nn))) throw std::invalid_argument("mesh: normal index out of range");
      if (t.tex != -1 && !in(t.tex, nm)) throw std::invalid_argument("mesh: texture index out of range");
    }
    for (int m : mesh_mats) if (m < 0 || m >= (int)mats.size()) throw std::invalid_argument("mesh: bad material id");
    M->verts = std::move(verts); M->norms = std::move(norms); M->tris = std::move(tris); M->mats = std::move(mesh_mats);
    M->bb = box_of_points(M->verts.data(), M->verts.size());  // Mesh.hs:55
    tbb.clear();
    for (auto& t : M->tris) { D3 pts[3] = {M->verts[t.a], M->verts[t.b], M->verts[t.c]}; tbb.push_back(box_of_points(pts, 3)); }  // Mesh.hs:119-121
    return M;
  }
  int mesh_node(std::shared_ptr<MeshData> M) { Node n; n.kind = K_MESH; n.mesh = std::move(M); return add(n); }
  int mesh(std::vector<D3> verts, std::vector<D3> norms, std::vector<MeshTri> tris, std::vector<int> mesh_mats) {
    std::vector<Box3> tbb;
    auto M = mesh_data(std::move(verts), std::move(norms), std::move(tris), std::move(mesh_mats), tbb);
    MeshBuild B2{*M, std::move(tbb)};
    std::vector<int> all(M->tris.size());
    for (size_t k = 0; k < all.size(); k++) all[k] = (int)k;
    B2.rec(all, M->bb, 0);
    return mesh_node(M);
  }

  // ---------------- materials ----------------
  int add_mat(Mat m) { mats.push_back(std::move(m)); return (int)mats.size() - 1; }
  void check_mat(int m) const { if (m < 0 || m >= (int)mats.size()) throw std::invalid_argument("bad material id"); }
};

}  // namespace glome
