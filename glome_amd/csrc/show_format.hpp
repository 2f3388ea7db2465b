// show_format.hpp -- the text GlomeView prints for a scene (`show geom`, the SDLK_s key, Glome.hs:431) as an interchange
// format: write a builder graph in it, read one back.  Host only.
//
// The text is what Haskell's derived `Show` makes of the solids' constructors (Haskell 2010 report, section 11.4:
// constructor application at precedence 10, arguments shown at precedence 11 -- negative numbers and constructor
// applications in parentheses --, record syntax for Bbox / Bih, list syntax for lists), with the reference's hand-written
// instances in between:
//   show (SolidItem s) = "SI " ++ show s            Solid.hs:277-278   (`show` only: the default showsPrec ignores the
//                                                    precedence, so an item is NEVER parenthesised, whatever holds it)
//   show (Texture)     = "Texture"                  Solid.hs:101-102   (a closure: the material does not survive)
//   show (Tag s t)     = "<Tag " ++ show s ++ ">"   Tex.hs:50-51       (the tag value does not survive)
//   show (Mesh ...)    = "Mesh " verts " " tris " " bbox " " bvh       Mesh.hs:44-46 (normals, textures, tags do not survive)
// Doubles are shown as GHC's `show :: Double -> String` does (GHC.Float.showFloat: shortest digits that read back to the
// same value; fixed notation for 0.1 <= |x| < 10^7, otherwise d.ddde<n>).  Because every geometric field is a Double
// shown losslessly and the Bih / BVH trees are spelled out node by node, a dump read back here is the reference's own
// scene, trees included -- which is the point: a `show geom` from a real GHC build pins this builder's trees, and a scene
// that only exists as Haskell source (TestScene.hs) can cross the boundary as text.  What cannot cross: materials (the
// loader takes them from the caller, one per `Tex` in reading order), tags, mesh vertex normals.
//
// No GHC in this image: the format is restated from the report's rules and the instances cited above, not checked
// against a real dump ("parity unpinned" for this file; tests pin writer and reader against an independent Python
// restatement of the same rules and against hand-written literals).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "host_graph.hpp"

namespace glome {

// GHC.Float.showFloat (formatRealFloat FFGeneric Nothing): digits ds and exponent e with x = 0.ds * 10^e
inline std::string show_double(double x) {
  if (std::isnan(x)) return "NaN";
  if (std::isinf(x)) return x < 0 ? "-Infinity" : "Infinity";
  std::string out;
  if (std::signbit(x)) { out = "-"; x = -x; }
  if (x == 0) return out + "0.0";
  char buf[40];
  int p = 0;
  for (p = 0; p < 17; p++) {  // shortest mantissa that reads back to x (floatToDigits)
    snprintf(buf, sizeof buf, "%.*e", p, x);
    if (strtod(buf, nullptr) == x) break;
  }
  std::string ds;
  const char* ep = strchr(buf, 'e');
  for (const char* c = buf; c < ep; c++) if (*c != '.') ds.push_back(*c);
  while (ds.size() > 1 && ds.back() == '0') ds.pop_back();
  int e = atoi(ep + 1) + 1;
  if (e < 0 || e > 7) {  // FFExponent
    out += ds[0];
    out += '.';
    out += ds.size() > 1 ? ds.substr(1) : std::string("0");
    out += 'e';
    out += std::to_string(e - 1);
    return out;
  }
  if (e <= 0) return out + "0." + std::string((size_t)(-e), '0') + ds;  // FFFixed, 0.1 <= x < 1
  std::string ip = ds.substr(0, std::min<size_t>(ds.size(), (size_t)e));
  ip.append((size_t)e - ip.size(), '0');
  std::string fp = ds.size() > (size_t)e ? ds.substr((size_t)e) : std::string("0");
  return out + ip + "." + fp;
}

struct ShowWriter {
  const Graph& G;
  std::string& o;
  std::vector<int>* tex_mats = nullptr;  // material ids of the Tex constructors, in the order the text holds them
  void num(double x) { o += show_double(x); }                          // showsPrec 0
  void arg(double x) { if (std::signbit(x) && !std::isnan(x)) { o += '('; num(x); o += ')'; } else num(x); }  // showsPrec 11
  void iarg(long v) { if (v < 0) { o += '('; o += std::to_string(v); o += ')'; } else o += std::to_string(v); }
  void vec(D3 v) { o += "Vec "; arg(v.x); o += ' '; arg(v.y); o += ' '; arg(v.z); }
  void vecarg(D3 v) { o += '('; vec(v); o += ')'; }
  void bbox(const Box3& b) { o += "Bbox {p1 = "; vec(b.lo); o += ", p2 = "; vec(b.hi); o += '}'; }
  void bboxarg(const Box3& b) { o += '('; bbox(b); o += ')'; }
  void matrix(const Mat34& m) { o += "(Matrix"; for (int k = 0; k < 12; k++) { o += ' '; arg(m.m[k]); } o += ')'; }
  void items(const std::vector<int>& ids) {
    o += '[';
    for (size_t k = 0; k < ids.size(); k++) { if (k) o += ','; item(ids[k]); }
    o += ']';
  }
  void item(int id) { o += "SI "; solid(id); }
  void bihnode(const BihTree& T, int k, bool paren) {
    const BihTree::Node& n = T.nodes[(size_t)k];
    if (paren) o += '(';
    if (n.leaf) { o += "BihLeaf "; items(n.items); }
    else {
      o += "BihBranch "; arg(n.lsplit); o += ' '; arg(n.rsplit); o += ' '; iarg(n.axis); o += ' ';
      bihnode(T, n.left, true); o += ' '; bihnode(T, n.right, true);
    }
    if (paren) o += ')';
  }
  void bvh(const MeshData& M, int k, bool paren) {
    const MeshData::Node& n = M.nodes[(size_t)k];
    if (paren) o += '(';
    if (n.leaf) {
      o += "Leaf [";
      for (size_t q = 0; q < n.tris.size(); q++) { if (q) o += ','; o += std::to_string(n.tris[q]); }
      o += ']';
    } else {
      o += "Branch "; bboxarg(n.lbb); o += ' '; bboxarg(n.rbb); o += ' ';
      bvh(M, n.left, true); o += ' '; bvh(M, n.right, true);
    }
    if (paren) o += ')';
  }
  void solid(int id) {
    const Node& n = G.at(id);
    const double* p = n.p;
    auto v = [&](int k) { return D3{p[k], p[k + 1], p[k + 2]}; };
    switch (n.kind) {
      case K_VOID: o += "Void"; break;
      case K_SPHERE: o += "Sphere "; vecarg(v(0)); o += ' '; arg(p[3]); o += ' '; arg(1.0 / p[3]); break;  // Sphere c r (1/r), Sphere.hs:15-17
      case K_TRI: o += "Triangle "; vecarg(v(0)); o += ' '; vecarg(v(3)); o += ' '; vecarg(v(6)); break;
      case K_TRIN:
        o += "TriangleNorm";
        for (int k = 0; k < 6; k++) { o += ' '; vecarg(v(3 * k)); }
        break;
      case K_BOX: o += "Box "; bboxarg(Box3{v(0), v(3)}); break;
      case K_PLANE: o += "Plane "; vecarg(v(0)); o += ' '; arg(p[3]); break;
      case K_DISC: o += "Disc "; vecarg(v(0)); o += ' '; vecarg(v(3)); o += ' '; arg(p[6]); break;
      case K_CYL: o += "Cylinder "; arg(p[0]); o += ' '; arg(p[1]); o += ' '; arg(p[2]); break;
      case K_CONE: o += "Cone "; arg(p[0]); o += ' '; arg(p[1]); o += ' '; arg(p[2]); o += ' '; arg(p[3]); break;
      case K_LIST: items(n.kids); break;  // instance Solid [SolidItem t m]: the list's own Show
      case K_INSTANCE: o += "Instance "; item(n.a); o += " (Xfm "; matrix(n.xf.f); o += ' '; matrix(n.xf.i); o += ')'; break;
      case K_DIFF: o += "Difference "; item(n.a); o += ' '; item(n.b); o += n.retex ? " False" : " True"; break;  // `difference` / `difference_retexture`, Csg.hs:26-30
      case K_ISECT: o += "Intersection "; items(n.kids); break;
      case K_BOUND: o += "Bound "; item(n.a); o += ' '; item(n.b); break;
      case K_INNERBOUND: o += "InnerBound "; item(n.a); o += ' '; item(n.b); break;
      case K_BIH: o += "Bih {bihbb = "; bbox(n.bih->bb); o += ", bihroot = "; bihnode(*n.bih, 0, false); o += '}'; break;
      case K_MESH: {
        const MeshData& M = *n.mesh;
        o += "Mesh [";
        for (size_t k = 0; k < M.verts.size(); k++) { if (k) o += ','; vec(M.verts[k]); }
        o += "] [";
        for (size_t k = 0; k < M.tris.size(); k++) {
          const MeshTri& t = M.tris[k];
          if (k) o += ',';
          o += "Tri";
          for (int f : {t.a, t.b, t.c, t.na, t.nb, t.nc, t.tex, t.tag}) { o += ' '; iarg(f); }
        }
        o += "] "; bbox(M.bb); o += ' '; bvh(M, 0, false);
        break;
      }
      case K_TEX: if (tex_mats) tex_mats->push_back(n.mat); o += "Tex "; item(n.a); o += " Texture"; break;
      case K_TAG: o += "<Tag "; item(n.a); o += '>'; break;
      case K_NOSHADOW: o += "NoShadow "; item(n.a); break;
      case K_ONLYSHADOW: o += "OnlyShadow "; item(n.a); break;
      default: throw std::invalid_argument("show: unknown node kind");
    }
  }
};

struct ShowReader {
  Graph& G;
  const char* s;
  const char* end;
  const int* tex_mats;
  int n_tex_mats, default_mat;
  int n_tex = 0;

  [[noreturn]] void fail(const std::string& what) const {
    size_t off = (size_t)(s - (end - total));
    throw scene_error("show text, offset " + std::to_string(off) + ": " + what);
  }
  size_t total = 0;
  void ws() { while (s < end && (*s == ' ' || *s == '\n' || *s == '\r' || *s == '\t')) s++; }
  bool peek(char c) { ws(); return s < end && *s == c; }
  bool eat(char c) { if (peek(c)) { s++; return true; } return false; }
  void need(char c) { if (!eat(c)) fail(std::string("expected '") + c + "'"); }
  static bool idc(char c) { return (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_' || c == '\''; }
  std::string name() {
    ws();
    const char* b = s;
    while (s < end && idc(*s)) s++;
    if (b == s) fail("expected a constructor");
    return std::string(b, s);
  }
  bool peek_name(const char* w) {
    ws();
    size_t n = strlen(w);
    return (size_t)(end - s) >= n && !strncmp(s, w, n) && (s + n == end || !idc(s[n]));
  }
  void need_name(const char* w) { if (!peek_name(w)) fail(std::string("expected ") + w); s += strlen(w); }
  double bare_num() {
    ws();
    bool neg = false;
    if (s < end && *s == '-') { neg = true; s++; }
    double v;
    if (peek_name("Infinity")) { s += 8; v = HUGE_VAL; }
    else if (peek_name("NaN")) { s += 3; v = NAN; }
    else {
      char* e = nullptr;
      v = strtod(s, &e);
      if (e == s || e > end) fail("expected a number");
      s = e;
    }
    return neg ? -v : v;
  }
  double num() {  // a numeric constructor argument: negative values come parenthesised
    if (eat('(')) { double v = bare_num(); need(')'); return v; }
    return bare_num();
  }
  long inum() {
    double v = num();
    if (v != std::floor(v) || std::fabs(v) > 2e9) fail("expected an integer");
    return (long)v;
  }
  D3 vec_bare() { need_name("Vec"); D3 v; v.x = num(); v.y = num(); v.z = num(); return v; }
  D3 vec() { need('('); D3 v = vec_bare(); need(')'); return v; }
  Box3 bbox_bare() {
    need_name("Bbox"); need('{'); need_name("p1"); need('='); Box3 b; b.lo = vec_bare(); need(','); need_name("p2"); need('='); b.hi = vec_bare(); need('}');
    return b;
  }
  Box3 bbox() { need('('); Box3 b = bbox_bare(); need(')'); return b; }
  Mat34 matrix() { need('('); need_name("Matrix"); Mat34 m; for (int k = 0; k < 12; k++) m.m[k] = num(); need(')'); return m; }
  std::vector<int> items() {
    std::vector<int> ids;
    need('[');
    if (eat(']')) return ids;
    do ids.push_back(item()); while (eat(','));
    need(']');
    return ids;
  }
  int item() { need_name("SI"); return solid(); }
  int bihnode(BihTree& T, int depth) {
    bool paren = eat('(');
    int me = (int)T.nodes.size();
    T.nodes.push_back({});
    T.depth = std::max(T.depth, depth + 1);
    std::string c = name();
    if (c == "BihLeaf") {
      std::vector<int> ids = items();
      BihTree::Node& n = T.nodes[(size_t)me];
      n.leaf = true; n.lsplit = n.rsplit = 0; n.axis = -1; n.left = n.right = -1; n.items = std::move(ids);
    } else if (c == "BihBranch") {
      double l = num(), r = num();
      long axis = inum();
      if (axis < 0 || axis > 2) fail("BihBranch axis must be 0, 1 or 2");
      int lc = bihnode(T, depth + 1), rc = bihnode(T, depth + 1);
      BihTree::Node& n = T.nodes[(size_t)me];
      n.leaf = false; n.lsplit = l; n.rsplit = r; n.axis = (int)axis; n.left = lc; n.right = rc;
    } else fail("expected BihLeaf or BihBranch, got " + c);
    if (paren) need(')');
    return me;
  }
  int bvh(MeshData& M, int depth) {
    bool paren = eat('(');
    int me = (int)M.nodes.size();
    M.nodes.push_back({});
    M.depth = std::max(M.depth, depth + 1);
    std::string c = name();
    if (c == "Leaf") {
      std::vector<int> ts;
      if (peek_name("fromList")) s += 8;  // older `vector` releases print `fromList [..]`
      need('[');
      if (!eat(']')) { do { long t = inum(); if (t < 0 || t >= (long)M.tris.size()) fail("BVH leaf: triangle index out of range"); ts.push_back((int)t); } while (eat(',')); need(']'); }
      MeshData::Node& n = M.nodes[(size_t)me];
      n.leaf = true; n.left = n.right = -1; n.tris = std::move(ts);
    } else if (c == "Branch") {
      Box3 lb = bbox(), rb = bbox();
      int lc = bvh(M, depth + 1), rc = bvh(M, depth + 1);
      MeshData::Node& n = M.nodes[(size_t)me];
      n.leaf = false; n.lbb = lb; n.rbb = rb; n.left = lc; n.right = rc;
    } else fail("expected Leaf or Branch, got " + c);
    if (paren) need(')');
    return me;
  }
  int prim(int kind, std::initializer_list<double> p) { Node n; n.kind = kind; int k = 0; for (double v : p) n.p[k++] = v; return G.add(n); }
  int solid() {
    if (peek('[')) return G.make_list(items());
    if (eat('<')) { need_name("Tag"); int c = item(); need('>'); return G.wrap(K_TAG, c); }
    std::string c = name();
    if (c == "Void") return prim(K_VOID, {});
    if (c == "Sphere") { D3 o = vec(); double r = num(); num(); return prim(K_SPHERE, {o.x, o.y, o.z, r}); }
    if (c == "Triangle") { D3 a = vec(), b = vec(), d = vec(); return G.add(Graph::tri_node(a, b, d)); }
    if (c == "TriangleNorm") {
      D3 a = vec(), b = vec(), d = vec(), na = vec(), nb = vec(), nc = vec();
      Node n = Graph::tri_node(a, b, d);
      n.kind = K_TRIN;
      const double v[9] = {na.x, na.y, na.z, nb.x, nb.y, nb.z, nc.x, nc.y, nc.z};
      std::copy(v, v + 9, n.p + 9);
      return G.add(n);
    }
    if (c == "Box") { Box3 b = bbox(); return prim(K_BOX, {b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z}); }
    if (c == "Plane") { D3 n = vec(); double off = num(); return prim(K_PLANE, {n.x, n.y, n.z, off}); }
    if (c == "Disc") { D3 pos = vec(), n = vec(); double r2 = num(); return prim(K_DISC, {pos.x, pos.y, pos.z, n.x, n.y, n.z, r2}); }
    if (c == "Cylinder") { double r = num(), h1 = num(), h2 = num(); return prim(K_CYL, {r, h1, h2}); }
    if (c == "Cone") { double r = num(), c1 = num(), c2 = num(), h = num(); return prim(K_CONE, {r, c1, c2, h}); }
    if (c == "Instance") {
      int child = item();
      need('('); need_name("Xfm"); Xf x; x.f = matrix(); x.i = matrix(); need(')');
      return G.instance_of(child, x, G.next_id());
    }
    if (c == "Difference") {
      int a = item(), b = item();
      std::string flag = name();
      if (flag != "True" && flag != "False") fail("expected True or False");
      return G.difference(a, b, flag == "False");  // False: difference_retexture (Csg.hs:29-30)
    }
    if (c == "Intersection") return G.intersection(items());
    if (c == "Bound") { int a = item(), b = item(); return G.bound_object(a, b, false); }
    if (c == "InnerBound") { int a = item(), b = item(); return G.bound_object(a, b, true); }
    if (c == "Tex") {
      int k = n_tex++;  // reading order = the order of the `Tex` words: an outer Tex comes before the ones inside it
      int child = item();
      need_name("Texture");
      int m = k < n_tex_mats ? tex_mats[k] : default_mat;
      if (m < 0) fail("no material for Tex number " + std::to_string(k) + " (textures are closures: `show` prints only \"Texture\")");
      return G.wrap(K_TEX, child, m);
    }
    if (c == "NoShadow") return G.wrap(K_NOSHADOW, item());
    if (c == "OnlyShadow") return G.wrap(K_ONLYSHADOW, item());
    if (c == "Bih") {
      auto T = std::make_shared<BihTree>();
      need('{'); need_name("bihbb"); need('='); T->bb = bbox_bare(); need(','); need_name("bihroot"); need('=');
      bihnode(*T, 0);
      need('}');
      Node n; n.kind = K_BIH; n.bih = T;
      return G.add(n);
    }
    if (c == "Mesh") {
      auto M = std::make_shared<MeshData>();
      if (peek_name("fromList")) s += 8;
      need('[');
      if (!eat(']')) { do M->verts.push_back(vec_bare()); while (eat(',')); need(']'); }
      if (peek_name("fromList")) s += 8;
      need('[');
      int max_tex = -1;
      if (!eat(']')) {
        do {
          need_name("Tri");
          long f[8];
          for (long& q : f) q = inum();
          for (int k = 0; k < 3; k++) if (f[k] < 0 || f[k] >= (long)M->verts.size()) fail("Tri: vertex index out of range");
          if (f[3] != -1 || f[4] != -1 || f[5] != -1) fail("mesh with vertex normals: `show` does not print the normals (Mesh.hs:44-46), the mesh cannot be restored");
          if (f[6] < -1) fail("Tri: bad texture index");
          max_tex = std::max(max_tex, (int)f[6]);
          M->tris.push_back(MeshTri{(int)f[0], (int)f[1], (int)f[2], -1, -1, -1, (int)f[6], (int)f[7]});
        } while (eat(','));
        need(']');
      }
      if (max_tex >= 0) {  // the mesh's texture vector is not printed either: every index gets the default material
        if (default_mat < 0) fail("mesh with per-triangle textures needs a default material");
        M->mats.assign((size_t)max_tex + 1, default_mat);
      }
      M->bb = bbox_bare();
      bvh(*M, 0);
      Node n; n.kind = K_MESH; n.mesh = M;
      return G.add(n);
    }
    fail("unknown solid constructor " + c);
  }
};

inline int load_show(Graph& G, const char* text, size_t len, const int* tex_mats, int n_tex_mats, int default_mat, int* n_tex) {
  for (int k = 0; k < n_tex_mats; k++) G.check_mat(tex_mats[k]);
  if (default_mat >= 0) G.check_mat(default_mat);
  ShowReader R{G, text, text + len, tex_mats, n_tex_mats, default_mat};
  R.total = len;
  int root = R.item();
  R.ws();
  if (R.s != R.end) R.fail("text after the scene");
  if (n_tex) *n_tex = R.n_tex;
  return root;
}

}  // namespace glome
