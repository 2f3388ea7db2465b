// glome_oracle.hpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// A C++17 restatement of the per-ray hot path of jimsnow/glome (Haskell), templated on the
// real type: Real=double is the oracle the HIP path is checked against; Real=float shows what
// plain fp32 arithmetic does to the same formulas.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may use anything in oracle/.  The product library
// (glome_amd/csrc) never includes, links or calls this file.
//
// PARITY STATUS: *parity unpinned by the reference*.  The reference ships no tests, golden
// vectors or fixtures and is Haskell, for which this image has no
// toolchain, so the reference itself cannot be run.  This restatement is pinned instead by
// (1) the hand-derived known-answer tests of SURVEY.md Appendix D (tests/test_oracle_kat.py),
// (2) an independent NumPy restatement of the primitive formulas (oracle/np_oracle.py),
// (3) the reference's own constructor-time invariants (check_xfm, orth, xyz_to_uvw ...).
//
// Every function cites the reference file:line it follows (paths relative to /root/reference).
// Quirk numbers Qn refer to SURVEY.md Appendix A.  No FMA contraction: build with
// -ffp-contract=off (GHC emits none).
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace glo {

// ---------------------------------------------------------------------------------------
// counters (the reference's rayint_debug convention, Bih.hs:378-412: +1 per BIH branch
// entered; we add +1 per leaf primitive tested and per ray class traced)
// ---------------------------------------------------------------------------------------
struct Counters {
  uint64_t bih_nodes = 0;     // BihBranch entered (Bih.hs:389-410 debug_wrap ... 1)
  uint64_t mesh_nodes = 0;    // Mesh Branch entered
  uint64_t prim_tests = 0;    // primitive rayint/shadow evaluations
  uint64_t rays_primary = 0;  // trace calls from the pixel loop
  uint64_t rays_shadow = 0;   // shadow calls from mpreshade
  uint64_t rays_secondary = 0;// trace calls from Reflect/Refract with recurs>0
  void add(const Counters& o) {
    bih_nodes += o.bih_nodes; mesh_nodes += o.mesh_nodes; prim_tests += o.prim_tests;
    rays_primary += o.rays_primary; rays_shadow += o.rays_shadow; rays_secondary += o.rays_secondary;
  }
};
inline Counters& tls_counters() { static thread_local Counters c; return c; }

// ---------------------------------------------------------------------------------------
// Vec.hs
// ---------------------------------------------------------------------------------------
template <class R> struct Math {
  static constexpr R infinity() { return R(1000000.0); }  // Vec.hs:12-14 (finite sentinel, Q0)
  static constexpr R delta() { return R(0.0001); }        // Vec.hs:40
  // Vec.hs:44-49: compare-selects, NOT fmin/fmax (NaN picks a specific operand, Q1)
  static R fmin(R a, R b) { return a > b ? b : a; }
  static R fmax(R a, R b) { return a > b ? a : b; }
  static R fmin3(R a, R b, R c) { return a > b ? (b > c ? c : b) : (a > c ? c : a); }  // Vec.hs:52-59
  static R fmax3(R a, R b, R c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }  // Vec.hs:62-69
  static R fabs_(R a) { return a < 0 ? -a : a; }                                        // Vec.hs:80-82
  // Prelude max/min on Double (used by Mesh.hs and bbsa): max x y = if x <= y then y else x
  static R pmax(R x, R y) { return x <= y ? y : x; }
  static R pmin(R x, R y) { return x <= y ? x : y; }
  static bool about_equal(R a, R b) {  // Vec.hs:96-102
    if (a > 1) return fabs_(1 - (a / b)) < (delta() * 10);
    return fabs_(a - b) < (delta() * 10);
  }
};

template <class R> struct Vec {
  R x, y, z;
  R operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }  // va, Vec.hs:167-172
};
template <class R> Vec<R> vset(Vec<R> v, int i, R f) {  // Vec.hs:176-181
  if (i == 0) v.x = f; else if (i == 1) v.y = f; else v.z = f;
  return v;
}
template <class R> R vdot(Vec<R> a, Vec<R> b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }  // Vec.hs:185-187
template <class R> Vec<R> vcross(Vec<R> a, Vec<R> b) {  // Vec.hs:193-198
  return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)};
}
template <class R> Vec<R> vinvert(Vec<R> a) { return {-a.x, -a.y, -a.z}; }
template <class R> R vlen(Vec<R> a) { return std::sqrt(vdot(a, a)); }  // Vec.hs:222-223
template <class R> Vec<R> vadd(Vec<R> a, Vec<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> Vec<R> vadd3(Vec<R> a, Vec<R> b, Vec<R> c) {  // Vec.hs:233-237
  return {a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z};
}
template <class R> Vec<R> vsub(Vec<R> a, Vec<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> Vec<R> vscale(Vec<R> a, R f) { return {a.x * f, a.y * f, a.z * f}; }
template <class R> Vec<R> vscaleadd(Vec<R> a, Vec<R> b, R f) {  // Vec.hs:302-306
  return {a.x + (b.x * f), a.y + (b.y * f), a.z + (b.z * f)};
}
template <class R> Vec<R> vnorm(Vec<R> a) {  // Vec.hs:314-317: reciprocal of the length, then 3 muls
  R invlen = R(1.0) / std::sqrt((a.x * a.x) + (a.y * a.y) + (a.z * a.z));
  return {a.x * invlen, a.y * invlen, a.z * invlen};
}
template <class R> Vec<R> bisect(Vec<R> a, Vec<R> b) { return vnorm(vadd(a, b)); }  // Vec.hs:331-332
template <class R> Vec<R> reflect(Vec<R> v, Vec<R> n) {  // Vec.hs:340-342
  return vscaleadd(v, n, R(-2) * vdot(v, n));
}
template <class R> Vec<R> vrcp(Vec<R> a) { return {1 / a.x, 1 / a.y, 1 / a.z}; }
template <class R> Vec<R> vmin(Vec<R> a, Vec<R> b) {
  return {Math<R>::fmin(a.x, b.x), Math<R>::fmin(a.y, b.y), Math<R>::fmin(a.z, b.z)};
}
template <class R> Vec<R> vmax(Vec<R> a, Vec<R> b) {
  return {Math<R>::fmax(a.x, b.x), Math<R>::fmax(a.y, b.y), Math<R>::fmax(a.z, b.z)};
}

template <class R> struct Ray { Vec<R> o, d; };
template <class R> Ray<R> ray_move(const Ray<R>& r, R d) { return {vscaleadd(r.o, r.d, d), r.d}; }  // Vec.hs:361-363

// orth, Vec.hs:366-378
template <class R> void orth(Vec<R> v1, Vec<R>& v2, Vec<R>& v3) {
  if (!Math<R>::about_equal(vdot(v1, v1), 1)) throw std::runtime_error("orth: unnormalized vector");
  Vec<R> X{1, 0, 0}, Y{0, 1, 0};
  R dvx = vdot(v1, X);
  v2 = (dvx < R(0.8) && dvx > R(-0.8)) ? vnorm(vcross(v1, X)) : vnorm(vcross(v1, Y));
  v3 = vcross(v1, v2);
}
// plane_int_dist, Vec.hs:391-394
template <class R> R plane_int_dist(const Ray<R>& r, Vec<R> p, Vec<R> n) {
  Vec<R> newo = vsub(r.o, p);
  return -(vdot(n, newo)) / (vdot(n, r.d));
}

// --- matrices / transforms, Vec.hs:407-629 ---
template <class R> struct Matrix { R m[12]; };
template <class R> struct Xfm { Matrix<R> f, i; };
template <class R> Matrix<R> ident_matrix() { return {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}}; }
template <class R> Xfm<R> ident_xfm() { return {ident_matrix<R>(), ident_matrix<R>()}; }
template <class R> Matrix<R> mat_mult(const Matrix<R>& A, const Matrix<R>& B) {  // Vec.hs:426-443
  const R* a = A.m; const R* b = B.m;
  Matrix<R> o;
  for (int r = 0; r < 3; r++) {
    o.m[r * 4 + 0] = a[r * 4 + 0] * b[0] + a[r * 4 + 1] * b[4] + a[r * 4 + 2] * b[8];
    o.m[r * 4 + 1] = a[r * 4 + 0] * b[1] + a[r * 4 + 1] * b[5] + a[r * 4 + 2] * b[9];
    o.m[r * 4 + 2] = a[r * 4 + 0] * b[2] + a[r * 4 + 1] * b[6] + a[r * 4 + 2] * b[10];
    o.m[r * 4 + 3] = a[r * 4 + 0] * b[3] + a[r * 4 + 1] * b[7] + a[r * 4 + 2] * b[11] + a[r * 4 + 3];
  }
  return o;
}
template <class R> Xfm<R> xfm_mult(const Xfm<R>& a, const Xfm<R>& b) {  // Vec.hs:447-449
  return {mat_mult(a.f, b.f), mat_mult(b.i, a.i)};
}
template <class R> Xfm<R> check_xfm(const Xfm<R>& x) {  // Vec.hs:466-477
  Matrix<R> p = mat_mult(x.f, x.i);
  static const R want[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int k = 0; k < 12; k++)
    if (!Math<R>::about_equal(p.m[k], want[k])) throw std::runtime_error("corrupt matrix");
  return x;
}
template <class R> Xfm<R> compose(const std::vector<Xfm<R>>& xs) {  // Vec.hs:461-462
  // foldr xfm_mult ident (reverse xs): x_n * (x_{n-1} * (... (x_1 * ident)))
  Xfm<R> acc = ident_xfm<R>();
  for (size_t k = 0; k < xs.size(); k++) acc = xfm_mult(xs[k], acc);
  return check_xfm(acc);
}
template <class R> Vec<R> xfm_point(const Xfm<R>& x, Vec<R> v) {  // Vec.hs:502-509
  const R* m = x.f.m;
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3], m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7],
          m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11]};
}
template <class R> Vec<R> invxfm_point(const Xfm<R>& x, Vec<R> v) {  // Vec.hs:512-519
  const R* m = x.i.m;
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3], m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7],
          m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11]};
}
template <class R> Vec<R> xfm_vec(const Xfm<R>& x, Vec<R> v) {  // Vec.hs:522-529
  const R* m = x.f.m;
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
          m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
template <class R> Vec<R> invxfm_vec(const Xfm<R>& x, Vec<R> v);
template <class R> Vec<R> xfm_vec(const Xfm<R>& x, Vec<R> v);
template <class R> Vec<R> xfm_point(const Xfm<R>& x, Vec<R> v);
template <class R> Vec<R> vnorm(Vec<R> v);
template <class R> Ray<R> xfm_ray(const Xfm<R>& x, const Ray<R>& r) { return Ray<R>{xfm_point(x, r.o), vnorm(xfm_vec(x, r.d))}; }  // Vec.hs:553-555
template <class R> Vec<R> invxfm_vec(const Xfm<R>& x, Vec<R> v) {  // Vec.hs:532-539
  const R* m = x.i.m;
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
          m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
template <class R> Vec<R> invxfm_norm(const Xfm<R>& x, Vec<R> v) {  // Vec.hs:543-550 (inverse transpose)
  const R* m = x.i.m;
  return {m[0] * v.x + m[4] * v.y + m[8] * v.z, m[1] * v.x + m[5] * v.y + m[9] * v.z,
          m[2] * v.x + m[6] * v.y + m[10] * v.z};
}
template <class R> Xfm<R> translate(Vec<R> v) {  // Vec.hs:564-567
  return check_xfm(Xfm<R>{{{1, 0, 0, v.x, 0, 1, 0, v.y, 0, 0, 1, v.z}}, {{1, 0, 0, -v.x, 0, 1, 0, -v.y, 0, 0, 1, -v.z}}});
}
template <class R> Xfm<R> scale(Vec<R> v) {  // Vec.hs:571-574
  return check_xfm(Xfm<R>{{{v.x, 0, 0, 0, 0, v.y, 0, 0, 0, 0, v.z, 0}},
                          {{1 / v.x, 0, 0, 0, 0, 1 / v.y, 0, 0, 0, 0, 1 / v.z, 0}}});
}
template <class R> Xfm<R> rotate(Vec<R> v, R angle) {  // Vec.hs:577-598
  if (!Math<R>::about_equal(vlen(v), 1)) throw std::runtime_error("please use a normalized vector for rotation");
  R x = v.x, y = v.y, z = v.z, s = std::sin(angle), c = std::cos(angle);
  R m00 = ((x * x) + ((1 - (x * x)) * c)), m01 = (((x * y) * (1 - c)) - (z * s)), m02 = ((x * z * (1 - c)) + (y * s));
  R m10 = (((x * y) * (1 - c)) + (z * s)), m11 = ((y * y) + ((1 - (y * y)) * c)), m12 = ((y * z * (1 - c)) - (x * s));
  R m20 = ((x * z * (1 - c)) - (y * s)), m21 = ((y * z * (1 - c)) + (x * s)), m22 = ((z * z) + ((1 - (z * z)) * c));
  return check_xfm(Xfm<R>{{{m00, m01, m02, 0, m10, m11, m12, 0, m20, m21, m22, 0}},
                          {{m00, m10, m20, 0, m01, m11, m21, 0, m02, m12, m22, 0}}});
}
template <class R> Xfm<R> xyz_to_uvw(Vec<R> u, Vec<R> v, Vec<R> w) {  // Vec.hs:602-622
  using M = Math<R>;
  if (!M::about_equal(vdot(u, u), 1)) throw std::runtime_error("unnormalized u");
  if (!M::about_equal(vdot(v, v), 1)) throw std::runtime_error("unnormalized v");
  if (!M::about_equal(vdot(w, w), 1)) throw std::runtime_error("unnormalized w");
  if (!(M::about_equal(vdot(u, v), 0) && M::about_equal(vdot(u, w), 0) && M::about_equal(vdot(v, w), 0)))
    throw std::runtime_error("vectors aren't orthogonal");
  return check_xfm(Xfm<R>{{{u.x, v.x, w.x, 0, u.y, v.y, w.y, 0, u.z, v.z, w.z, 0}},
                          {{u.x, u.y, u.z, 0, v.x, v.y, v.z, 0, w.x, w.y, w.z, 0}}});
}

// --- bounding boxes, Vec.hs:646-762 ---
template <class R> struct Bbox { Vec<R> p1, p2; };
template <class R> Bbox<R> empty_bbox() {  // Vec.hs:706-709
  R i = Math<R>::infinity();
  return {{i, i, i}, {-i, -i, -i}};
}
template <class R> Bbox<R> everything_bbox() {  // Vec.hs:712-715
  R i = Math<R>::infinity();
  return {{-i, -i, -i}, {i, i, i}};
}
template <class R> Bbox<R> bbjoin(const Bbox<R>& a, const Bbox<R>& b) { return {vmin(a.p1, b.p1), vmax(a.p2, b.p2)}; }     // Vec.hs:652-654
template <class R> Bbox<R> bboverlap(const Bbox<R>& a, const Bbox<R>& b) { return {vmax(a.p1, b.p1), vmin(a.p2, b.p2)}; }  // Vec.hs:657-659
template <class R> Bbox<R> bbpts(const std::vector<Vec<R>>& pts) {  // Vec.hs:676-690 (right fold, +-delta pad)
  using M = Math<R>;
  R d = M::delta();
  if (pts.empty()) return empty_bbox<R>();
  const Vec<R>& l = pts.back();
  Bbox<R> b{{l.x - d, l.y - d, l.z - d}, {l.x + d, l.y + d, l.z + d}};
  for (size_t k = pts.size() - 1; k-- > 0;) {
    const Vec<R>& p = pts[k];
    b = {{M::fmin(p.x - d, b.p1.x), M::fmin(p.y - d, b.p1.y), M::fmin(p.z - d, b.p1.z)},
         {M::fmax(p.x + d, b.p2.x), M::fmax(p.y + d, b.p2.y), M::fmax(p.z + d, b.p2.z)}};
  }
  return b;
}
template <class R> R bbsa(const Bbox<R>& b) {  // Vec.hs:694-697 (Prelude max)
  Vec<R> d = vsub(b.p2, b.p1);
  return Math<R>::pmax(0, 2 * (d.x * d.y + d.x * d.z + d.y * d.z));
}
template <class R> Vec<R> bbmid(const Bbox<R>& b) { return vscale(vadd(b.p1, b.p2), R(0.5)); }  // Bih.hs:162
// bbclip_ub_rcp, Vec.hs:725-741.  The ray's d holds reciprocals.
template <class R> void bbclip_ub_rcp(const Ray<R>& r, const Bbox<R>& b, R& nearv, R& farv) {
  R inx, outx, iny, outy, inz, outz;
  if (r.d.x > 0) { inx = (b.p1.x - r.o.x) * r.d.x; outx = (b.p2.x - r.o.x) * r.d.x; } else { inx = (b.p2.x - r.o.x) * r.d.x; outx = (b.p1.x - r.o.x) * r.d.x; }
  if (r.d.y > 0) { iny = (b.p1.y - r.o.y) * r.d.y; outy = (b.p2.y - r.o.y) * r.d.y; } else { iny = (b.p2.y - r.o.y) * r.d.y; outy = (b.p1.y - r.o.y) * r.d.y; }
  if (r.d.z > 0) { inz = (b.p1.z - r.o.z) * r.d.z; outz = (b.p2.z - r.o.z) * r.d.z; } else { inz = (b.p2.z - r.o.z) * r.d.z; outz = (b.p1.z - r.o.z) * r.d.z; }
  nearv = Math<R>::fmax3(inx, iny, inz);
  farv = Math<R>::fmin3(outx, outy, outz);
}
// bbclip_ub, Vec.hs:743-762: branches on d (not on 1/d), no zero guard (Q1)
template <class R> void bbclip_ub(const Ray<R>& r, const Bbox<R>& b, R& nearv, R& farv) {
  R dxr = 1 / r.d.x, dyr = 1 / r.d.y, dzr = 1 / r.d.z;
  R inx, outx, iny, outy, inz, outz;
  if (r.d.x > 0) { inx = (b.p1.x - r.o.x) * dxr; outx = (b.p2.x - r.o.x) * dxr; } else { inx = (b.p2.x - r.o.x) * dxr; outx = (b.p1.x - r.o.x) * dxr; }
  if (r.d.y > 0) { iny = (b.p1.y - r.o.y) * dyr; outy = (b.p2.y - r.o.y) * dyr; } else { iny = (b.p2.y - r.o.y) * dyr; outy = (b.p1.y - r.o.y) * dyr; }
  if (r.d.z > 0) { inz = (b.p1.z - r.o.z) * dzr; outz = (b.p2.z - r.o.z) * dzr; } else { inz = (b.p2.z - r.o.z) * dzr; outz = (b.p1.z - r.o.z) * dzr; }
  nearv = Math<R>::fmax3(inx, iny, inz);
  farv = Math<R>::fmin3(outx, outy, outz);
}

// ---------------------------------------------------------------------------------------
// Clr.hs
// ---------------------------------------------------------------------------------------
template <class R> struct Color { R r, g, b; };
template <class R> struct ColorA { R r, g, b, a; };
template <class R> Color<R> cadd(Color<R> a, Color<R> b) { return {a.r + b.r, a.g + b.g, a.b + b.b}; }  // Clr.hs:23-25
template <class R> Color<R> cscale(Color<R> c, R m) { return {c.r * m, c.g * m, c.b * m}; }              // Clr.hs:44-46
template <class R> ColorA<R> cafold(ColorA<R> c1, ColorA<R> c2) {                                        // Clr.hs:106-113
  R trans = 1 - c1.a;
  return {c1.r + (c2.r * trans * c2.a), c1.g + (c2.g * trans * c2.a), c1.b + (c2.b * trans * c2.a), c1.a + (c2.a * trans)};
}
template <class R> ColorA<R> caweight(ColorA<R> c1, ColorA<R> c2, R w) {  // Clr.hs:87-91
  auto f = [w](R a, R b) { return (a * w) + (b * (1 - w)); };
  return {f(c1.r, c2.r), f(c1.g, c2.g), f(c1.b, c2.b), f(c1.a, c2.a)};
}
template <class R> R aclamp(R x) { return x > 1 ? R(1) : (x < 0 ? R(0) : x); }  // Clr.hs:75-79
template <class R> ColorA<R> casum(const std::vector<ColorA<R>>& cs) {          // Clr.hs:93-103 (alphas :82-85)
  R r = 0, g = 0, b = 0;
  for (auto& c : cs) { r = r + c.r * c.a; g = g + c.g * c.a; b = b + c.b * c.a; }
  R prod = 1;  // Haskell `product` = foldl (*) 1
  for (auto& c : cs) prod = prod * (1 - aclamp(c.a));
  return {r, g, b, 1 - prod};
}

// ---------------------------------------------------------------------------------------
// Solid.hs: Rayint, nearest, the Solid class
// ---------------------------------------------------------------------------------------
// A texture is an opaque closure in the reference (Solid.hs:97).  Here it is defunctionalised to
// an integer material id (uniform textures, Shader.hs:55-56 / TestScene.hs:201-245); tags are
// dropped (picking only).  The stack is head-first like the Haskell list.
struct TexList {
  int n = 0;
  int v[8];
  void push_front(int t) {
    if (n >= 8) throw std::runtime_error("texture stack deeper than 8");
    for (int k = n; k > 0; k--) v[k] = v[k - 1];
    v[0] = t; n++;
  }
  static TexList cat(const TexList& a, const TexList& b) {  // a ++ b
    TexList o = a;
    for (int k = 0; k < b.n; k++) { if (o.n >= 8) throw std::runtime_error("texture stack deeper than 8"); o.v[o.n++] = b.v[k]; }
    return o;
  }
};

template <class R> struct Rayint {  // Solid.hs:20-28 (riuvw is never read by anything; omitted)
  bool hit = false;
  R depth = 0;
  Vec<R> pos{0, 0, 0}, norm{0, 0, 0};
  Ray<R> riray{{0, 0, 0}, {0, 0, 0}};  // the ray as the primitive that was hit saw it (local inside Instances, advanced inside CSG); Warp reads it
  TexList tex;
  int prim = -1;  // id of the constructor call that made the primitive (not in the reference)
};
template <class R> R ridepth(const Rayint<R>& r) { return r.hit ? r.depth : Math<R>::infinity(); }  // Solid.hs:33-34
// nearest, Solid.hs:37-44: ties -> second argument (Q9)
template <class R> const Rayint<R>& nearest(const Rayint<R>& a, const Rayint<R>& b) {
  if (!b.hit) return a;
  if (!a.hit) return b;
  return (a.depth < b.depth) ? a : b;
}

template <class R> struct Solid;
template <class R> using SP = std::shared_ptr<const Solid<R>>;
struct Pcount { long prims = 0, xfms = 0, bounds = 0; };  // Solid.hs:105-123
inline Pcount pcadd(Pcount a, Pcount b) { return {a.prims + b.prims, a.xfms + b.xfms, a.bounds + b.bounds}; }
inline Pcount asbound(Pcount a) { return {0, a.xfms, a.prims + a.bounds}; }

template <class R> SP<R> make_instance(SP<R> s, const Xfm<R>& x, int uid);
template <class R> SP<R> make_list(std::vector<SP<R>> v);

template <class R> struct Solid : std::enable_shared_from_this<Solid<R>> {  // Solid.hs:138-254
  int uid = -1;
  virtual ~Solid() {}
  virtual const char* name() const = 0;
  virtual Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const = 0;
  // default shadow falls back on rayint, Solid.hs:218-221 (Q15)
  virtual bool shadow(const Ray<R>& r, R d) const { return rayint(r, d, TexList()).hit; }
  virtual bool inside(Vec<R>) const = 0;
  virtual Bbox<R> bound() const = 0;
  virtual std::vector<SP<R>> tolist() const { return {this->shared_from_this()}; }  // Solid.hs:230
  virtual SP<R> transform(const std::vector<Xfm<R>>& xs, int newuid) const {       // Solid.hs:235
    return make_instance<R>(this->shared_from_this(), compose(xs), newuid);
  }
  virtual SP<R> transform_leaf(const std::vector<Xfm<R>>& xs, int newuid) const { return transform(xs, newuid); }  // Solid.hs:240
  virtual std::vector<SP<R>> flatten_transform() const { return tolist(); }                                         // Solid.hs:246
  virtual Pcount primcount() const { return {1, 0, 0}; }                                                            // Solid.hs:251
  virtual TexList get_metainfo(Vec<R>) const { return TexList(); }                                                  // Solid.hs:254
};
// `flatten_transform (SolidItem s) = [SolidItem (flatten_transform s)]`, Solid.hs:273
template <class R> std::vector<SP<R>> flatten_transform_item(const SP<R>& s) { return {make_list<R>(s->flatten_transform())}; }

// ---- Void, Solid.hs:349-360 ----
template <class R> struct Void : Solid<R> {
  const char* name() const override { return "Void"; }
  Rayint<R> rayint(const Ray<R>&, R, const TexList&) const override { return {}; }
  bool shadow(const Ray<R>&, R) const override { return false; }
  bool inside(Vec<R>) const override { return false; }
  Bbox<R> bound() const override { return empty_bbox<R>(); }
  std::vector<SP<R>> tolist() const override { return {}; }
  SP<R> transform(const std::vector<Xfm<R>>&, int) const override { return this->shared_from_this(); }
};

// ---- list instance (group), Solid.hs:326-339 ----
template <class R> struct ListSolid : Solid<R> {
  std::vector<SP<R>> xs;
  const char* name() const override { return "List"; }
  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override {  // foldl' nearest RayMiss; same d for all (Q9)
    Rayint<R> acc;
    for (auto& s : xs) { Rayint<R> h = s->rayint(r, d, t); acc = nearest(acc, h); }
    return acc;
  }
  bool shadow(const Ray<R>& r, R d) const override {  // foldl' (||) False (lazy: later shadows not evaluated once True)
    for (auto& s : xs) if (s->shadow(r, d)) return true;
    return false;
  }
  bool inside(Vec<R> p) const override {
    for (auto& s : xs) if (s->inside(p)) return true;
    return false;
  }
  Bbox<R> bound() const override {
    Bbox<R> b = empty_bbox<R>();
    for (auto& s : xs) b = bbjoin(b, s->bound());
    return b;
  }
  std::vector<SP<R>> tolist() const override {
    std::vector<SP<R>> o;
    for (auto& s : xs) { auto l = s->tolist(); o.insert(o.end(), l.begin(), l.end()); }
    return o;
  }
  SP<R> transform_leaf(const std::vector<Xfm<R>>& xf, int newuid) const override {  // Solid.hs:334
    std::vector<SP<R>> o;
    for (auto& s : tolist()) o.push_back(s->transform_leaf(xf, newuid));
    return make_list<R>(o);
  }
  std::vector<SP<R>> flatten_transform() const override {  // Solid.hs:335
    std::vector<SP<R>> o;
    for (auto& s : xs) { auto l = flatten_transform_item<R>(s); o.insert(o.end(), l.begin(), l.end()); }
    return o;
  }
  Pcount primcount() const override {
    Pcount p;
    for (auto& s : xs) p = pcadd(p, s->primcount());
    return p;
  }
  TexList get_metainfo(Vec<R> v) const override {  // Solid.hs:337-339: later containing items are prepended
    TexList acc;
    for (auto& s : xs) if (s->inside(v)) acc = TexList::cat(s->get_metainfo(v), acc);
    return acc;
  }
};
template <class R> SP<R> make_list(std::vector<SP<R>> v) {
  auto l = std::make_shared<ListSolid<R>>();
  l->xs = std::move(v);
  return l;
}
// group, Solid.hs:293-302 (Q22)
template <class R> SP<R> group(const std::vector<SP<R>>& slds) {
  if (slds.empty()) return std::make_shared<Void<R>>();
  if (slds.size() == 1) return slds[0];
  std::vector<SP<R>> o;
  for (auto& s : slds) { auto l = s->tolist(); o.insert(o.end(), l.begin(), l.end()); }
  return make_list<R>(o);
}

// ---- Instance, Solid.hs:386-532 (Q8) ----
template <class R> struct Instance : Solid<R> {
  SP<R> s;
  Xfm<R> x;
  const char* name() const override { return "Instance"; }
  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override {  // Solid.hs:388-403
    Vec<R> newdir = invxfm_vec(x, r.d), neworig = invxfm_point(x, r.o);
    R lenscale = vlen(newdir), invlenscale = 1 / lenscale;
    Rayint<R> h = s->rayint(Ray<R>{neworig, vscale(newdir, invlenscale)}, d * lenscale, t);
    if (!h.hit) return h;
    h.depth = h.depth * invlenscale;
    h.pos = xfm_point(x, h.pos);
    h.norm = vnorm(invxfm_norm(x, h.norm));
    return h;
  }
  bool shadow(const Ray<R>& r, R d) const override {  // Solid.hs:464-471
    Vec<R> newdir = invxfm_vec(x, r.d), neworig = invxfm_point(x, r.o);
    R lenscale = vlen(newdir), invlenscale = 1 / lenscale;
    return s->shadow(Ray<R>{neworig, vscale(newdir, invlenscale)}, d * lenscale);
  }
  bool inside(Vec<R> p) const override { return s->inside(invxfm_point(x, p)); }  // Solid.hs:473-475
  Bbox<R> bound() const override {                                                // Solid.hs:477-484
    Bbox<R> b = s->bound();
    std::vector<Vec<R>> pts;
    for (R px : {b.p1.x, b.p2.x}) for (R py : {b.p1.y, b.p2.y}) for (R pz : {b.p1.z, b.p2.z}) pts.push_back(xfm_point(x, Vec<R>{px, py, pz}));
    return bbpts(pts);
  }
  SP<R> transform(const std::vector<Xfm<R>>& x1, int newuid) const override {  // Solid.hs:494-496
    std::vector<Xfm<R>> l{x};
    l.insert(l.end(), x1.begin(), x1.end());
    return s->transform({compose(l)}, newuid);
  }
  SP<R> transform_leaf(const std::vector<Xfm<R>>& x1, int newuid) const override {  // Solid.hs:498-500
    std::vector<Xfm<R>> l{x};
    l.insert(l.end(), x1.begin(), x1.end());
    return s->transform_leaf({compose(l)}, newuid);
  }
  std::vector<SP<R>> flatten_transform() const override { return {s->transform_leaf({x}, this->uid)}; }  // Solid.hs:509-511
  Pcount primcount() const override { return pcadd(s->primcount(), Pcount{0, 1, 0}); }
  TexList get_metainfo(Vec<R> v) const override { return s->get_metainfo(invxfm_point(x, v)); }  // Solid.hs:517-519
};
template <class R> SP<R> make_instance(SP<R> s, const Xfm<R>& x, int uid) {
  auto i = std::make_shared<Instance<R>>();
  i->s = s; i->x = x; i->uid = uid;
  return i;
}

// rayint_advance, Solid.hs:85-91
template <class R> Rayint<R> rayint_advance(const Solid<R>& s, const Ray<R>& r, R d, const TexList& t, R adv) {
  R a = adv + Math<R>::delta();
  Rayint<R> h = s.rayint(ray_move(r, a), d - a, t);
  if (!h.hit) return h;
  h.depth = h.depth + a;
  return h;
}

// ---- Sphere.hs ----
template <class R> struct Sphere : Solid<R> {
  Vec<R> c; R r, invr;
  const char* name() const override { return "Sphere"; }
  Rayint<R> rayint(const Ray<R>& ray, R dist, const TexList& t) const override {  // Sphere.hs:20-41 (Q4)
    tls_counters().prim_tests++;
    Vec<R> eo = vsub(c, ray.o);
    R v = vdot(eo, ray.d), vsqr = v * v, csqr = vdot(eo, eo), rsqr = r * r;
    R disc = rsqr - (csqr - vsqr);
    if (disc < R(0.0)) return {};
    R d = std::sqrt(disc);
    R hitdist = ((v - d) > 0) ? (v - d) : (v + d);
    if ((hitdist < 0) || (hitdist > dist)) return {};
    Rayint<R> h;
    h.hit = true; h.riray = ray; h.depth = hitdist;
    h.pos = vscaleadd(ray.o, ray.d, hitdist);
    h.norm = vnorm(vsub(h.pos, c));
    h.tex = t; h.prim = this->uid;
    return h;
  }
  bool shadow(const Ray<R>& ray, R dist) const override {  // Sphere.hs:51-71
    tls_counters().prim_tests++;
    Vec<R> eo = vsub(c, ray.o);
    R v = vdot(eo, ray.d);
    if ((dist >= (v - r)) && (v > R(0.0))) {
      R vsqr = v * v, csqr = vdot(eo, eo), rsqr = r * r;
      R disc = rsqr - (csqr - vsqr);
      if (disc < R(0.0)) return false;
      R d = std::sqrt(disc);
      R hitdist = ((v - d) > 0) ? (v - d) : (v + d);
      return !((hitdist < 0) || (hitdist > dist));
    }
    return false;
  }
  bool inside(Vec<R> p) const override {  // Sphere.hs:73-76
    Vec<R> off = vsub(c, p);
    return vdot(off, off) < r * r;
  }
  Bbox<R> bound() const override {  // Sphere.hs:78-81
    Vec<R> off{r, r, r};
    return {vsub(c, off), vadd(c, off)};
  }
};

// ---- Triangle.hs ----
// shared Moller-Trumbore core (Triangle.hs:45-73 / 82-107 / 109-141 are three copies of it)
template <class R> bool mt_core(Vec<R> p1, Vec<R> p2, Vec<R> p3, const Ray<R>& ray, R dist, R& t, R& b1, R& b2) {
  Vec<R> e1 = vsub(p2, p1), e2 = vsub(p3, p1);
  Vec<R> s1 = vcross(ray.d, e2);
  R divisor = vdot(s1, e1);
  if (divisor == 0) return false;
  R invdivisor = R(1.0) / divisor;
  Vec<R> d = vsub(ray.o, p1);
  b1 = vdot(d, s1) * invdivisor;
  if (b1 < 0 || b1 > 1) return false;
  Vec<R> s2 = vcross(d, e1);
  b2 = vdot(ray.d, s2) * invdivisor;
  if (b2 < 0 || b1 + b2 > 1) return false;
  t = vdot(e2, s2) * invdivisor;
  if (t < 0 || t > dist) return false;
  return true;
}
template <class R> Bbox<R> bound_triangle(Vec<R> a, Vec<R> b, Vec<R> c) {  // Triangle.hs:147-158
  using M = Math<R>;
  R d = M::delta();
  return {{M::fmin(M::fmin(a.x, b.x), c.x) - d, M::fmin(M::fmin(a.y, b.y), c.y) - d, M::fmin(M::fmin(a.z, b.z), c.z) - d},
          {M::fmax(M::fmax(a.x, b.x), c.x) + d, M::fmax(M::fmax(a.y, b.y), c.y) + d, M::fmax(M::fmax(a.z, b.z), c.z) + d}};
}
template <class R> struct Triangle : Solid<R> {
  Vec<R> p1, p2, p3;
  const char* name() const override { return "Triangle"; }
  Rayint<R> rayint(const Ray<R>& ray, R dist, const TexList& tex) const override {  // Triangle.hs:45-73 (Q5)
    tls_counters().prim_tests++;
    R t, b1, b2;
    if (!mt_core(p1, p2, p3, ray, dist, t, b1, b2)) return {};
    Rayint<R> h;
    h.hit = true; h.riray = ray; h.depth = t;
    h.pos = vscaleadd(ray.o, ray.d, t);
    h.norm = vnorm(vcross(vsub(p2, p1), vsub(p3, p1)));  // not flipped toward the viewer
    h.tex = tex; h.prim = this->uid;
    return h;
  }
  bool shadow(const Ray<R>& ray, R dist) const override {  // Triangle.hs:82-107
    tls_counters().prim_tests++;
    R t, b1, b2;
    return mt_core(p1, p2, p3, ray, dist, t, b1, b2);
  }
  bool inside(Vec<R>) const override { return false; }
  Bbox<R> bound() const override { return bound_triangle(p1, p2, p3); }
  SP<R> transform(const std::vector<Xfm<R>>& xs, int newuid) const override {  // Triangle.hs:164-168: bakes
    Xfm<R> x = compose(xs);
    auto t = std::make_shared<Triangle<R>>();
    t->p1 = xfm_point(x, p1); t->p2 = xfm_point(x, p2); t->p3 = xfm_point(x, p3); t->uid = newuid;
    return t;
  }
};
template <class R> struct TriangleNorm : Solid<R> {
  Vec<R> p1, p2, p3, n1, n2, n3;
  const char* name() const override { return "TriangleNorm"; }
  Rayint<R> rayint(const Ray<R>& ray, R dist, const TexList& tex) const override {  // Triangle.hs:109-141
    tls_counters().prim_tests++;
    R t, b1, b2;
    if (!mt_core(p1, p2, p3, ray, dist, t, b1, b2)) return {};
    Rayint<R> h;
    h.hit = true; h.riray = ray; h.depth = t;
    h.pos = vscaleadd(ray.o, ray.d, t);
    h.norm = vnorm(vadd3(vscale(n1, 1 - (b1 + b2)), vscale(n2, b1), vscale(n3, b2)));
    h.tex = tex; h.prim = this->uid;
    return h;
  }
  bool shadow(const Ray<R>& ray, R dist) const override {  // Triangle.hs:143-145
    tls_counters().prim_tests++;
    R t, b1, b2;
    return mt_core(p1, p2, p3, ray, dist, t, b1, b2);
  }
  bool inside(Vec<R>) const override { return false; }
  Bbox<R> bound() const override { return bound_triangle(p1, p2, p3); }
  SP<R> transform(const std::vector<Xfm<R>>& xs, int newuid) const override {  // Triangle.hs:170-177
    Xfm<R> x = compose(xs);
    auto t = std::make_shared<TriangleNorm<R>>();
    t->p1 = xfm_point(x, p1); t->p2 = xfm_point(x, p2); t->p3 = xfm_point(x, p3);
    t->n1 = vnorm(xfm_vec(x, n1)); t->n2 = vnorm(xfm_vec(x, n2)); t->n3 = vnorm(xfm_vec(x, n3));
    t->uid = newuid;
    return t;
  }
};

// ---- Box.hs ----
template <class R> struct Box : Solid<R> {
  Bbox<R> bb;
  const char* name() const override { return "Box"; }
  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override {  // Box.hs:18-54 (Q1, Q6)
    tls_counters().prim_tests++;
    using M = Math<R>;
    R ox = r.o.x, oy = r.o.y, oz = r.o.z, dx = r.d.x, dy = r.d.y, dz = r.d.z;
    R dxrcp = 1 / dx, dyrcp = 1 / dy, dzrcp = 1 / dz;
    R inx, outx, iny, outy, inz, outz;
    if (dx > 0) { inx = (bb.p1.x - ox) * dxrcp; outx = (bb.p2.x - ox) * dxrcp; } else { inx = (bb.p2.x - ox) * dxrcp; outx = (bb.p1.x - ox) * dxrcp; }
    if (dy > 0) { iny = (bb.p1.y - oy) * dyrcp; outy = (bb.p2.y - oy) * dyrcp; } else { iny = (bb.p2.y - oy) * dyrcp; outy = (bb.p1.y - oy) * dyrcp; }
    if (dz > 0) { inz = (bb.p1.z - oz) * dzrcp; outz = (bb.p2.z - oz) * dzrcp; } else { inz = (bb.p2.z - oz) * dzrcp; outz = (bb.p1.z - oz) * dzrcp; }
    R lastin = M::fmax3(inx, iny, inz), firstout = M::fmin3(outx, outy, outz);
    if (lastin > firstout || firstout < 0 || lastin > d) return {};
    Rayint<R> h;
    h.hit = true; h.riray = r; h.tex = t; h.prim = this->uid;
    if (lastin < 0) {  // origin is inside
      Vec<R> n;
      if (outx == firstout) n = (dx > 0) ? Vec<R>{1, 0, 0} : Vec<R>{-1, 0, 0};
      else if (outy == firstout) n = (dy > 0) ? Vec<R>{0, 1, 0} : Vec<R>{0, -1, 0};
      else n = (dz > 0) ? Vec<R>{0, 0, 1} : Vec<R>{0, 0, -1};
      h.depth = firstout; h.pos = vscaleadd(r.o, r.d, firstout); h.norm = n;
    } else {
      Vec<R> n;
      if (inx == lastin) n = (dx > 0) ? Vec<R>{-1, 0, 0} : Vec<R>{1, 0, 0};
      else if (iny == lastin) n = (dy > 0) ? Vec<R>{0, -1, 0} : Vec<R>{0, 1, 0};
      else n = (dz > 0) ? Vec<R>{0, 0, -1} : Vec<R>{0, 0, 1};
      h.depth = lastin; h.pos = vscaleadd(r.o, r.d, lastin); h.norm = n;
    }
    return h;
  }
  bool shadow(const Ray<R>& r, R d) const override {  // Box.hs:56-62
    tls_counters().prim_tests++;
    R nearv, farv;
    bbclip_ub(r, bb, nearv, farv);
    return !((nearv > farv) || farv <= 0 || farv > d);
  }
  bool inside(Vec<R> p) const override {  // Box.hs:64-68 (strict)
    return p.x > bb.p1.x && p.x < bb.p2.x && p.y > bb.p1.y && p.y < bb.p2.y && p.z > bb.p1.z && p.z < bb.p2.z;
  }
  Bbox<R> bound() const override { return bb; }
};

// ---- Plane.hs ----
template <class R> struct Plane : Solid<R> {
  Vec<R> n; R off;
  const char* name() const override { return "Plane"; }
  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override {  // Plane.hs:27-32 (Q2: NaN passes)
    tls_counters().prim_tests++;
    R hit = -((vdot(n, r.o) - off) / vdot(n, r.d));
    if (hit < 0 || hit > d) return {};
    Rayint<R> h;
    h.hit = true; h.riray = r; h.depth = hit; h.pos = vscaleadd(r.o, r.d, hit); h.norm = n; h.tex = t; h.prim = this->uid;
    return h;
  }
  bool inside(Vec<R> p) const override {  // Plane.hs:34-38
    Vec<R> onplane = vscale(n, off);
    return vdot(vsub(onplane, p), n) > 0;
  }
  Bbox<R> bound() const override { return everything_bbox<R>(); }  // Plane.hs:40-41
};

// ---- Cone.hs: Disc, Cylinder (z axis), Cone (z axis) ----
template <class R> bool disc_hit(Vec<R> point, Vec<R> norm, R radius_sqr, const Ray<R>& r, R d, R& dist, Vec<R>& pos) {  // Cone.hs:69-79
  dist = plane_int_dist(r, point, norm);
  if (dist < 0 || dist > d) return false;
  pos = vscaleadd(r.o, r.d, dist);
  Vec<R> off = vsub(pos, point);
  return !(vdot(off, off) > radius_sqr);
}
template <class R> Rayint<R> rayint_disc(Vec<R> point, Vec<R> norm, R radius_sqr, const Ray<R>& r, R d, const TexList& t, int uid) {
  R dist; Vec<R> pos;
  if (!disc_hit(point, norm, radius_sqr, r, d, dist, pos)) return {};
  Rayint<R> h;
  h.hit = true; h.riray = r; h.depth = dist; h.pos = pos; h.norm = norm; h.tex = t; h.prim = uid;
  return h;
}
template <class R> struct Disc : Solid<R> {
  Vec<R> p, n; R r2;
  const char* name() const override { return "Disc"; }
  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override { tls_counters().prim_tests++; return rayint_disc(p, n, r2, r, d, t, this->uid); }
  bool shadow(const Ray<R>& r, R d) const override {  // Cone.hs:81-91
    tls_counters().prim_tests++;
    R dist; Vec<R> pos;
    return disc_hit(p, n, r2, r, d, dist, pos);
  }
  bool inside(Vec<R>) const override { return false; }
  Bbox<R> bound() const override {  // Cone.hs:93-95
    R r = std::sqrt(r2);
    Vec<R> off{r, r, r};
    return {vsub(p, off), vadd(p, off)};
  }
};
template <class R> struct Cylinder : Solid<R> {  // radius height1 height2, Cone.hs:22
  R r, h1, h2;
  const char* name() const override { return "Cylinder"; }
  Rayint<R> rayint(const Ray<R>& ray, R d, const TexList& t) const override {  // Cone.hs:104-139 (Q7)
    tls_counters().prim_tests++;
    using M = Math<R>;
    R ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
    R a = dx * dx + dy * dy, b = 2 * (dx * ox + dy * oy), c = ox * ox + oy * oy - r * r;
    R disc = b * b - 4 * a * c;
    if (disc < 0) return {};
    R discsqrt = std::sqrt(disc);
    R q = (b < 0) ? (b - discsqrt) * R(-0.5) : (b + discsqrt) * R(-0.5);
    R t0p = q / a, t1p = c / q;
    R t0 = M::fmin(t0p, t1p), t1 = M::fmax(t0p, t1p);
    if (t1 < 0 || t0 > d) return {};
    R dist = (t0 < 0) ? t1 : t0;
    if (dist < 0 || dist > d) return {};
    Vec<R> pos = vscaleadd(ray.o, ray.d, dist);
    if (pos.z > h1 && pos.z < h2) {
      Rayint<R> h;
      h.hit = true; h.riray = ray; h.depth = dist; h.pos = pos; h.norm = Vec<R>{pos.x / r, pos.y / r, 0}; h.tex = t; h.prim = this->uid;
      return h;
    }
    if (dz > 0) {  // ray pointing up from bottom
      if (oz < h1) return rayint_disc(Vec<R>{0, 0, h1}, Vec<R>{0, 0, -1}, r * r, ray, d, t, this->uid);
      return {};
    }
    if (oz > h2) return rayint_disc(Vec<R>{0, 0, h2}, Vec<R>{0, 0, 1}, r * r, ray, d, t, this->uid);
    return {};
  }
  // no shadow override: falls back on rayint (Cone.hs:149-152)
  bool inside(Vec<R> p) const override { return p.z > h1 && p.z < h2 && p.x * p.x + p.y * p.y < r * r; }  // Cone.hs:141-143
  Bbox<R> bound() const override { return {{-r, -r, h1}, {r, r, h2}}; }                                   // Cone.hs:145-147
};
template <class R> struct Cone : Solid<R> {  // r clip1 clip2 height, Cone.hs:23
  R r, clip1, clip2, height;
  const char* name() const override { return "Cone"; }
  // shared quadratic, Cone.hs:155-184 / 206-235; returns 0 miss, 1 side hit, 2 "try caps"
  int solve(const Ray<R>& ray, R d, R& dist, Vec<R>& pos) const {
    using M = Math<R>;
    R ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
    R kp = (r / height), k = kp * kp;
    R a = dx * dx + dy * dy - k * dz * dz;
    R b = 2 * (dx * ox + dy * oy - k * dz * (oz - height));
    R c = ox * ox + oy * oy - k * (oz - height) * (oz - height);
    R disc = b * b - 4 * a * c;
    if (disc < 0) return 0;
    R discsqrt = std::sqrt(disc);
    R q = (b < 0) ? (b - discsqrt) * R(-0.5) : (b + discsqrt) * R(-0.5);
    R t0p = q / a, t1p = c / q;
    R t0 = M::fmin(t0p, t1p), t1 = M::fmax(t0p, t1p);
    if (t1 < 0 || t0 > d) return 0;
    dist = (t0 < 0) ? t1 : t0;
    if (dist < 0 || dist > d) return 0;
    pos = vscaleadd(ray.o, ray.d, dist);
    if (pos.z > clip1 && pos.z < clip2) return 1;
    return 2;
  }
  Rayint<R> rayint(const Ray<R>& ray, R d, const TexList& t) const override {  // Cone.hs:155-200
    tls_counters().prim_tests++;
    R dist; Vec<R> pos;
    int k = solve(ray, d, dist, pos);
    if (k == 0) return {};
    if (k == 1) {
      R invhyp = 1 / std::sqrt(height * height + r * r);
      R up = r * invhyp, out = height * invhyp;
      R r_ = std::sqrt(pos.x * pos.x + pos.y * pos.y);
      R correction = out / r_;
      Rayint<R> h;
      h.hit = true; h.riray = ray; h.depth = dist; h.pos = pos; h.norm = Vec<R>{pos.x * correction, pos.y * correction, up}; h.tex = t; h.prim = this->uid;
      return h;
    }
    if (ray.d.z > 0) {
      if (ray.o.z < clip1) return rayint_disc(Vec<R>{0, 0, clip1}, Vec<R>{0, 0, -1}, r * r, ray, d, t, this->uid);
      return {};
    }
    if (ray.o.z > clip2) {
      R r2 = r * (1 - ((clip2 - clip1) / height));
      return rayint_disc(Vec<R>{0, 0, clip2}, Vec<R>{0, 0, 1}, r2 * r2, ray, d, t, this->uid);
    }
    return {};
  }
  bool shadow(const Ray<R>& ray, R d) const override {  // Cone.hs:206-245
    tls_counters().prim_tests++;
    R dist; Vec<R> pos;
    int k = solve(ray, d, dist, pos);
    if (k == 0) return false;
    if (k == 1) return true;
    R dd; Vec<R> pp;
    if (ray.d.z > 0) {
      if (ray.o.z < clip1) return disc_hit(Vec<R>{0, 0, clip1}, Vec<R>{0, 0, -1}, r * r, ray, d, dd, pp);
      return false;
    }
    if (ray.o.z > clip2) {
      R r2 = r * (1 - ((clip2 - clip1) / height));
      return disc_hit(Vec<R>{0, 0, clip2}, Vec<R>{0, 0, 1}, r2 * r2, ray, d, dd, pp);
    }
    return false;
  }
  bool inside(Vec<R> p) const override {  // Cone.hs:248-251
    R rr = r * (1 - ((p.z - clip1) / height));
    return p.z > clip1 && p.z < clip2 && p.x * p.x + p.y * p.y < rr * rr;
  }
  Bbox<R> bound() const override { return {{-r, -r, clip1}, {r, r, clip2}}; }  // Cone.hs:253-255
};

// ---- Csg.hs ----
template <class R> struct Difference : Solid<R> {
  SP<R> sa, sb; bool useatex = true;
  const char* name() const override { return "Difference"; }
  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override {  // Csg.hs:33-54 (Q13)
    if (sb->inside(r.o)) {
      Rayint<R> rib = sb->rayint(r, d, t);
      if (!rib.hit) return rib;
      if (sa->inside(rib.pos) && !sb->inside(vscaleadd(rib.pos, r.d, Math<R>::delta()))) {
        Rayint<R> h = rib;
        h.norm = vinvert(rib.norm);
        if (useatex) h.tex = sa->get_metainfo(rib.pos);
        return h;
      }
      return rayint_advance<R>(*this, r, d, t, rib.depth);
    }
    Rayint<R> ria = sa->rayint(r, d, t);
    if (!ria.hit) return ria;
    Rayint<R> rib = sb->rayint(r, d, t);
    if (!rib.hit) return ria;
    if (ria.depth < rib.depth) return ria;
    return rayint_advance<R>(*this, r, d, t, rib.depth);
  }
  bool inside(Vec<R> p) const override { return sa->inside(p) && !sb->inside(p); }  // Csg.hs:92-94
  Bbox<R> bound() const override { return sa->bound(); }                             // Csg.hs:113-114
  Pcount primcount() const override { return pcadd(sa->primcount(), sb->primcount()); }
  TexList get_metainfo(Vec<R> p) const override {  // Csg.hs:103-106
    if (sa->inside(p) && !sb->inside(p)) return sa->get_metainfo(p);
    return TexList();
  }
};
template <class R> struct Intersection : Solid<R> {
  std::vector<SP<R>> slds;
  const char* name() const override { return "Intersection"; }
  // rayint_intersection works on list tails; `from` is the index of the current head
  Rayint<R> rayint_from(size_t from, const Ray<R>& r, R d, const TexList& t) const {  // Csg.hs:68-90 (Q14)
    if (from >= slds.size() || d < 0) return {};
    const SP<R>& s = slds[from];
    if (from + 1 == slds.size()) return s->rayint(r, d, t);
    if (s->inside(r.o)) {
      Rayint<R> hs = s->rayint(r, d, t);
      if (!hs.hit) return rayint_from(from + 1, r, d, t);
      Rayint<R> rest = rayint_from(from + 1, r, hs.depth, t);
      if (!rest.hit) return advance_from(from, r, d, t, hs.depth);
      return rest;
    }
    Rayint<R> hs = s->rayint(r, d, t);
    if (!hs.hit) return hs;
    if (inside_from(from + 1, hs.pos)) return hs;  // RayHit sd sp sn r vzero st stags
    return advance_from(from, r, d, t, hs.depth);
  }
  Rayint<R> advance_from(size_t from, const Ray<R>& r, R d, const TexList& t, R adv) const {  // Solid.hs:85-91 on (Intersection slds)
    R a = adv + Math<R>::delta();
    Rayint<R> h = rayint_from(from, ray_move(r, a), d - a, t);
    if (!h.hit) return h;
    h.depth = h.depth + a;
    return h;
  }
  bool inside_from(size_t from, Vec<R> p) const {  // Csg.hs:96-101 (True for empty)
    bool acc = true;
    for (size_t k = from; k < slds.size(); k++) acc = acc && slds[k]->inside(p);
    return acc;
  }
  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override { return rayint_from(0, r, d, t); }
  bool inside(Vec<R> p) const override { return inside_from(0, p); }
  Bbox<R> bound() const override {  // Csg.hs:116-120
    if (slds.empty()) return empty_bbox<R>();
    Bbox<R> b = everything_bbox<R>();
    for (auto& s : slds) b = bboverlap(b, s->bound());
    return b;
  }
  Pcount primcount() const override {
    Pcount p;
    for (auto& s : slds) p = pcadd(p, s->primcount());
    return p;
  }
  TexList get_metainfo(Vec<R> p) const override {  // Csg.hs:108-111
    if (!inside_from(0, p)) return TexList();
    TexList acc;
    for (auto& s : slds) acc = TexList::cat(acc, s->get_metainfo(p));
    return acc;
  }
};

// ---- Bound.hs ----
template <class R> struct Bound : Solid<R> {
  SP<R> sa, sb;
  const char* name() const override { return "Bound"; }
  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override {  // Bound.hs:30-35 (Q15)
    if (sa->inside(r.o) || sa->shadow(r, d)) return sb->rayint(r, d, t);
    return {};
  }
  bool shadow(const Ray<R>& r, R d) const override {  // Bound.hs:44-49
    if (sa->inside(r.o) || sa->shadow(r, d)) return sb->shadow(r, d);
    return false;
  }
  bool inside(Vec<R> p) const override { return sa->inside(p) && sb->inside(p); }  // Bound.hs:51-52
  Bbox<R> bound() const override { return bboverlap(sa->bound(), sb->bound()); }   // Bound.hs:61-62
  SP<R> transform_leaf(const std::vector<Xfm<R>>& xs, int newuid) const override { return sb->transform_leaf(xs, newuid); }  // Bound.hs:69-71
  std::vector<SP<R>> flatten_transform() const override { return flatten_transform_item<R>(sb); }                              // Bound.hs:73-74
  Pcount primcount() const override { return pcadd(asbound(sa->primcount()), sb->primcount()); }
  TexList get_metainfo(Vec<R> v) const override { return sa->inside(v) ? sb->get_metainfo(v) : TexList(); }  // Bound.hs:54-58
};
template <class R> struct InnerBound : Solid<R> {
  SP<R> sa, sb;
  const char* name() const override { return "InnerBound"; }
  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override {  // Bound.hs:97-99
    return sb->rayint(r, ridepth(sa->rayint(r, d, TexList())), t);
  }
  bool shadow(const Ray<R>& r, R d) const override { return sa->shadow(r, d) || sb->shadow(r, d); }  // Bound.hs:101-103
  bool inside(Vec<R> p) const override { return sa->inside(p) || sb->inside(p); }
  Bbox<R> bound() const override { return sb->bound(); }
  SP<R> transform_leaf(const std::vector<Xfm<R>>& xs, int newuid) const override { return sb->transform_leaf(xs, newuid); }
  std::vector<SP<R>> flatten_transform() const override { return flatten_transform_item<R>(sb); }
  Pcount primcount() const override { return pcadd(asbound(sa->primcount()), sb->primcount()); }
  TexList get_metainfo(Vec<R> v) const override { return sb->get_metainfo(v); }
};

// ---- Tex.hs (Tag is a pass-through here: tags only feed picking) ----
template <class R> struct Tex : Solid<R> {
  SP<R> s; int tex;
  const char* name() const override { return "Tex"; }
  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override {  // Tex.hs:66
    TexList t2 = t;
    t2.push_front(tex);
    return s->rayint(r, d, t2);
  }
  bool shadow(const Ray<R>& r, R d) const override { return s->shadow(r, d); }
  bool inside(Vec<R> p) const override { return s->inside(p); }
  Bbox<R> bound() const override { return s->bound(); }
  Pcount primcount() const override { return s->primcount(); }
  TexList get_metainfo(Vec<R> v) const override {  // Tex.hs:73-74
    TexList t = s->get_metainfo(v);
    t.push_front(tex);
    return t;
  }
};
template <class R> struct Passthru : Solid<R> {  // Tag (Tex.hs:53-62), NoShadow (:77-85), OnlyShadow (:88-96)
  SP<R> s; int mode;                               // 0 = Tag, 1 = NoShadow, 2 = OnlyShadow
  const char* name() const override { return mode == 0 ? "Tag" : (mode == 1 ? "NoShadow" : "OnlyShadow"); }
  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override { return mode == 2 ? Rayint<R>() : s->rayint(r, d, t); }
  bool shadow(const Ray<R>& r, R d) const override { return mode == 1 ? false : s->shadow(r, d); }
  bool inside(Vec<R> p) const override { return s->inside(p); }
  Bbox<R> bound() const override { return s->bound(); }
  Pcount primcount() const override { return s->primcount(); }
  TexList get_metainfo(Vec<R> v) const override { return s->get_metainfo(v); }
};

// ---- Bih.hs ----
template <class R> struct BihNode {  // Bih.hs:55-57
  bool leaf = true;
  std::vector<SP<R>> objs;  // BihLeaf [s]
  R lsplit = 0, rsplit = 0; int axis = 0;
  std::unique_ptr<BihNode<R>> l, r;
};
template <class R> struct Bih : Solid<R> {
  Bbox<R> bb;
  std::unique_ptr<BihNode<R>> root;
  const char* name() const override { return "Bih"; }

  using Obj = std::pair<Bbox<R>, SP<R>>;
  // build_rec, Bih.hs:211-285 (Q11) -- restated as written, including the `costy < costb` typo at :283
  static std::unique_ptr<BihNode<R>> build_rec(const std::vector<Obj>& objs, const Bbox<R>& bb, Vec<R> mid, int depth, size_t objcount) {
    using M = Math<R>;
    auto node = std::make_unique<BihNode<R>>();
    auto leaf = [&]() { node->leaf = true; for (auto& o : objs) node->objs.push_back(o.second); };
    if (objcount <= 3) { leaf(); return node; }
    R sa = M::pmax(0, bbsa(bb));
    std::vector<Obj> l[4], r[4];  // x, y, z, big/small
    for (auto& o : objs) {
      Vec<R> m = bbmid(o.first);
      (m.x < mid.x ? l[0] : r[0]).push_back(o);
      (m.y < mid.y ? l[1] : r[1]).push_back(o);
      (m.z < mid.z ? l[2] : r[2]).push_back(o);
      (M::pmax(0, bbsa(o.first)) > sa * R(0.4) ? l[3] : r[3]).push_back(o);
    }
    static const int ax[4] = {0, 1, 2, 0};
    R lmax[4], rmin[4], cost[4];
    Bbox<R> lbb[4], rbb[4];
    for (int k = 0; k < 4; k++) {
      lmax[k] = -M::infinity(); rmin[k] = M::infinity();
      for (auto& o : l[k]) lmax[k] = M::fmax(lmax[k], o.first.p2[ax[k]]);
      for (auto& o : r[k]) rmin[k] = M::fmin(rmin[k], o.first.p1[ax[k]]);
      lbb[k] = Bbox<R>{bb.p1, vset(bb.p2, ax[k], lmax[k])};
      rbb[k] = Bbox<R>{vset(bb.p1, ax[k], rmin[k]), bb.p2};
      cost[k] = ((M::pmax(0, bbsa(lbb[k])) * R(l[k].size())) + (M::pmax(0, bbsa(rbb[k])) * R(r[k].size()))) * (k < 3 ? R(1.1) : R(1.2));
    }
    R costx = cost[0], costy = cost[1], costz = cost[2], costb = cost[3];
    R costorig = sa * R(objcount);
    if (costorig < costx && costorig < costy && costorig < costz && costorig < costb) { leaf(); return node; }
    int k;
    if (costx < costy && costx < costz && costx < costb) k = 0;
    else if (costy < costz && costy < costb) k = 1;
    else if (costy < costb) k = 2;  // sic (Bih.hs:283)
    else k = 3;
    node->leaf = false;
    node->lsplit = lmax[k] + M::delta();
    node->rsplit = rmin[k] - M::delta();
    node->axis = ax[k];
    node->l = build_rec(l[k], lbb[k], bbmid(lbb[k]), depth + 1, l[k].size());
    node->r = build_rec(r[k], rbb[k], bbmid(rbb[k]), depth + 1, r[k].size());
    return node;
  }

  Rayint<R> rayint(const Ray<R>& r, R d, const TexList& t) const override {  // Bih.hs:332-368 (Q10)
    R nearv, farv;
    bbclip_ub(r, bb, nearv, farv);
    R dirr[3] = {1 / r.d.x, 1 / r.d.y, 1 / r.d.z};
    return traverse(*root, r, dirr, nearv, Math<R>::fmin(d, farv), t);
  }
  Rayint<R> traverse(const BihNode<R>& n, const Ray<R>& r, const R* dirrs, R nearv, R farv, const TexList& t) const {
    using M = Math<R>;
    if (n.leaf) {  // rayint [s] r far: list instance, same tmax for every item
      Rayint<R> acc;
      for (auto& s : n.objs) { Rayint<R> h = s->rayint(r, farv, t); acc = nearest(acc, h); }
      return acc;
    }
    tls_counters().bih_nodes++;
    R dirr = dirrs[n.axis], o = r.o[n.axis];
    R dl = (n.lsplit - o) * dirr, dr = (n.rsplit - o) * dirr;
    if (nearv > farv) return {};
    if (dirr > 0) {
      Rayint<R> a = (nearv < dl) ? traverse(*n.l, r, dirrs, nearv, M::fmin(dl, farv), t) : Rayint<R>();
      Rayint<R> b = (dr < farv) ? traverse(*n.r, r, dirrs, M::fmax(dr, nearv), farv, t) : Rayint<R>();
      return nearest(a, b);
    }
    Rayint<R> a = (nearv < dr) ? traverse(*n.r, r, dirrs, nearv, M::fmin(dr, farv), t) : Rayint<R>();
    Rayint<R> b = (dl < farv) ? traverse(*n.l, r, dirrs, M::fmax(dl, nearv), farv, t) : Rayint<R>();
    return nearest(a, b);
  }
  bool shadow(const Ray<R>& r, R d) const override {  // Bih.hs:510-544
    R nearv, farp;
    bbclip_ub(r, bb, nearv, farp);
    return shadow_traverse(*root, r, d, nearv, Math<R>::fmin(d, farp));
  }
  bool shadow_traverse(const BihNode<R>& n, const Ray<R>& r, R d, R nearv, R farv) const {
    using M = Math<R>;
    if (n.leaf) {
      R dd = M::fmin(d, farv);
      for (auto& s : n.objs) if (s->shadow(r, dd)) return true;
      return false;
    }
    tls_counters().bih_nodes++;
    R dirr = 1 / r.d[n.axis], o = r.o[n.axis];
    R dl = (n.lsplit - o) * dirr, dr = (n.rsplit - o) * dirr;
    if (nearv > farv) return false;
    if (dirr > 0)
      return ((nearv < dl) ? shadow_traverse(*n.l, r, d, nearv, M::fmin(dl, farv)) : false) ||
             ((dr < farv) ? shadow_traverse(*n.r, r, d, M::fmax(dr, nearv), farv) : false);
    return ((nearv < dr) ? shadow_traverse(*n.r, r, d, nearv, M::fmin(dr, farv)) : false) ||
           ((dl < farv) ? shadow_traverse(*n.l, r, d, M::fmax(dl, nearv), farv) : false);
  }
  bool inside_traverse(const BihNode<R>& n, Vec<R> p) const {  // Bih.hs:552-561
    if (n.leaf) { for (auto& s : n.objs) if (s->inside(p)) return true; return false; }
    R o = p[n.axis];
    return ((o < n.lsplit) ? inside_traverse(*n.l, p) : false) || ((o > n.rsplit) ? inside_traverse(*n.r, p) : false);
  }
  bool inbox(Vec<R> p) const { return p.x > bb.p1.x && p.x < bb.p2.x && p.y > bb.p1.y && p.y < bb.p2.y && p.z > bb.p1.z && p.z < bb.p2.z; }
  bool inside(Vec<R> p) const override { return inbox(p) && inside_traverse(*root, p); }  // Bih.hs:550-565
  TexList meta_traverse(const BihNode<R>& n, Vec<R> p) const {                             // Bih.hs:568-577
    if (n.leaf) {  // get_metainfo on the leaf list (Solid.hs:337-339)
      TexList acc;
      for (auto& s : n.objs) if (s->inside(p)) acc = TexList::cat(s->get_metainfo(p), acc);
      return acc;
    }
    R o = p[n.axis];
    TexList a = (o < n.lsplit) ? meta_traverse(*n.l, p) : TexList();
    TexList b = (o > n.rsplit) ? meta_traverse(*n.r, p) : TexList();
    return TexList::cat(a, b);
  }
  TexList get_metainfo(Vec<R> p) const override { return inbox(p) ? meta_traverse(*root, p) : TexList(); }
  Bbox<R> bound() const override { return bb; }
  static Pcount bihcount(const BihNode<R>& n) {  // Bih.hs:591-595
    if (n.leaf) { Pcount p; for (auto& s : n.objs) p = pcadd(p, s->primcount()); return p; }
    return pcadd(pcadd(bihcount(*n.l), bihcount(*n.r)), Pcount{0, 0, 1});
  }
  Pcount primcount() const override { return pcadd(bihcount(*root), Pcount{0, 0, 1}); }
};
// bih, Bih.hs:309-324
template <class R> SP<R> bih(const std::vector<SP<R>>& slds, int uid) {
  if (slds.empty()) return std::make_shared<Void<R>>();
  std::vector<typename Bih<R>::Obj> objs;
  Bbox<R> bb = empty_bbox<R>();
  for (auto& s : slds) { objs.push_back({s->bound(), s}); }
  for (auto& o : objs) bb = bbjoin(bb, o.first);
  R inf = Math<R>::infinity();
  if (bb.p1.x == -inf || bb.p1.y == -inf || bb.p1.z == -inf || bb.p2.x == inf || bb.p2.y == inf || bb.p2.z == inf)
    throw std::runtime_error("bih: infinite bounding box");
  auto b = std::make_shared<Bih<R>>();
  b->bb = bb; b->uid = uid;
  b->root = Bih<R>::build_rec(objs, bb, bbmid(bb), 0, slds.size());
  return b;
}

// ---- Mesh.hs ----
struct Tri { int a, b, c, na, nb, nc, tex, tag; };  // Mesh.hs:29
template <class R> struct MeshBVH {                 // Mesh.hs:36
  bool leaf = true;
  std::vector<int> tris;
  Bbox<R> lbb, rbb;
  std::unique_ptr<MeshBVH<R>> l, r;
};
template <class R> struct Mesh : Solid<R> {
  std::vector<Vec<R>> verts, norms;
  std::vector<Tri> tris;
  std::vector<int> texs;  // material ids
  Bbox<R> bb;
  std::unique_ptr<MeshBVH<R>> bvh;
  std::vector<Bbox<R>> alltribbs;
  const char* name() const override { return "Mesh"; }

  Bbox<R> trisbb(const std::vector<int>& idx) const {  // Mesh.hs:124-125
    Bbox<R> b = empty_bbox<R>();
    for (int i : idx) b = bbjoin(b, alltribbs[i]);
    return b;
  }
  std::unique_ptr<MeshBVH<R>> build_tree(const std::vector<int>& ts, const Bbox<R>& box) const {  // Mesh.hs:69-113 (Q12)
    auto node = std::make_unique<MeshBVH<R>>();
    size_t n = ts.size();
    if (n < 3) { node->tris = ts; return node; }
    Vec<R> mid = bbmid(box);
    R sa = bbsa(box);
    std::vector<int> l[4], r[4];
    for (int t : ts) {
      Vec<R> m = bbmid(alltribbs[t]);
      (m.x < mid.x ? l[0] : r[0]).push_back(t);
      (m.y < mid.y ? l[1] : r[1]).push_back(t);
      (m.z < mid.z ? l[2] : r[2]).push_back(t);
      (bbsa(alltribbs[t]) > sa * R(0.4) ? l[3] : r[3]).push_back(t);
    }
    Bbox<R> lbb[4], rbb[4];
    R cost[4];
    for (int k = 0; k < 4; k++) {
      lbb[k] = trisbb(l[k]); rbb[k] = trisbb(r[k]);
      cost[k] = (bbsa(lbb[k]) * R(l[k].size()) + bbsa(rbb[k]) * R(r[k].size())) * R(1.1);
    }
    R xcost = cost[0], ycost = cost[1], zcost = cost[2], bcost = cost[3];
    R lcost = bbsa(box) * R(n);
    if (lcost < xcost && lcost < ycost && lcost < zcost && lcost < bcost) { node->tris = ts; return node; }
    int k;
    if (xcost < ycost && xcost < zcost && xcost < bcost) k = 0;
    else if (ycost < zcost && ycost < bcost) k = 1;
    else if (zcost < bcost) k = 2;
    else k = 3;
    node->leaf = false;
    node->lbb = lbb[k]; node->rbb = rbb[k];
    node->l = build_tree(l[k], lbb[k]);
    node->r = build_tree(r[k], rbb[k]);
    return node;
  }
  void build() {  // mesh, Mesh.hs:50-55, 119-121
    bb = bbpts(verts);
    alltribbs.clear();
    for (auto& t : tris) alltribbs.push_back(bbpts(std::vector<Vec<R>>{verts[t.a], verts[t.b], verts[t.c]}));
    std::vector<int> all(tris.size());
    for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
    bvh = build_tree(all, bb);
  }
  Rayint<R> rayint_tri(int i, const Ray<R>& ray, R farv, const TexList& texs_in) const {  // Mesh.hs:143-161
    tls_counters().prim_tests++;
    const Tri& T = tris[i];
    TexList tex = texs_in;
    if (T.tex != -1) tex.push_front(texs[T.tex]);
    Vec<R> a = verts[T.a], b = verts[T.b], c = verts[T.c];
    R t, b1, b2;
    if (!mt_core(a, b, c, ray, farv, t, b1, b2)) return {};
    Rayint<R> h;
    h.hit = true; h.riray = ray; h.depth = t; h.pos = vscaleadd(ray.o, ray.d, t); h.tex = tex; h.prim = this->uid;
    if (T.na == -1) h.norm = vnorm(vcross(vsub(b, a), vsub(c, a)));
    else h.norm = vnorm(vadd3(vscale(norms[T.na], 1 - (b1 + b2)), vscale(norms[T.nb], b1), vscale(norms[T.nc], b2)));
    return h;
  }
  Rayint<R> traverse(const MeshBVH<R>& n, const Ray<R>& ray, const Ray<R>& ray_rcp, R depth, R nearv, R farv, const TexList& t) const {  // Mesh.hs:163-196
    using M = Math<R>;
    if (n.leaf) {
      Rayint<R> acc;
      for (int i : n.tris) { Rayint<R> h = rayint_tri(i, ray, farv, t); acc = nearest(acc, h); }
      return acc;
    }
    tls_counters().mesh_nodes++;
    R lnp, lfp, rnp, rfp;
    bbclip_ub_rcp(ray_rcp, n.lbb, lnp, lfp);
    bbclip_ub_rcp(ray_rcp, n.rbb, rnp, rfp);
    R lnear = M::pmax(nearv, lnp), lfar = M::pmin(farv, lfp), rnear = M::pmax(nearv, rnp), rfar = M::pmin(farv, rfp);
    if (lnear < rnear) {
      Rayint<R> lres = (lnear > lfar || lnear > depth || lfar < 0) ? Rayint<R>() : traverse(*n.l, ray, ray_rcp, depth, lnear, lfar, t);
      R rfar2 = M::pmin(rfar, ridepth(lres));
      Rayint<R> rres = (rnear > rfar2 || rnear > depth || rfar2 < 0) ? Rayint<R>() : traverse(*n.r, ray, ray_rcp, depth, rnear, rfar, t);  // unshrunk rfar (Q12)
      return nearest(lres, rres);
    }
    Rayint<R> rres = (rnear > rfar || rnear > depth || rfar < 0) ? Rayint<R>() : traverse(*n.r, ray, ray_rcp, depth, rnear, rfar, t);
    R lfar2 = M::pmin(lfar, ridepth(rres));
    Rayint<R> lres = (lnear > lfar2 || lnear > depth || lfar2 < 0) ? Rayint<R>() : traverse(*n.l, ray, ray_rcp, depth, lnear, lfar, t);
    return nearest(rres, lres);
  }
  Rayint<R> rayint(const Ray<R>& ray, R depth, const TexList& t) const override {  // Mesh.hs:136-198
    Ray<R> ray_rcp{ray.o, vrcp(ray.d)};
    R nearv, farv;
    bbclip_ub_rcp(ray_rcp, bb, nearv, farv);
    if (nearv > farv || nearv > depth || farv < 0) return {};
    return traverse(*bvh, ray, ray_rcp, depth, nearv, farv, t);  // leaves use the box far, not depth (as written)
  }
  bool shadow(const Ray<R>&, R) const override { return false; }  // Mesh.hs:210
  bool inside(Vec<R>) const override { return false; }            // Mesh.hs:211
  Bbox<R> bound() const override { return bb; }
  static Pcount pcount(const MeshBVH<R>& n) {  // Mesh.hs:201-205
    if (n.leaf) return Pcount{(long)n.tris.size(), 0, 0};
    return pcadd(pcadd(pcount(*n.l), pcount(*n.r)), Pcount{0, 0, 1});
  }
  Pcount primcount() const override { return pcount(*bvh); }
};

// ---------------------------------------------------------------------------------------
// Shader.hs / Trace.hs
// ---------------------------------------------------------------------------------------
template <class R> struct Light {  // Shader.hs:13-23; falloff fixed to \x -> 1/(x*x) as `light` builds it
  Vec<R> pos; Color<R> col; R rad = Math<R>::infinity(); bool shadow = true;
};
// ---- solid texture functions (GlomeVec/Data/Glome/Texture.hs): scalar fields 0..1 used as Blend weights by
// TestScene.hs's t_stripe / t_mottled (TestScene.hs:214-234)
enum WeightFn { W_CONST = 0, W_PERLIN = 1, W_STRIPE_SQUARE = 2, W_STRIPE_TRIANGLE = 3, W_STRIPE_SINE = 4 };
template <class R> R tx_omega(R t_) {  // Texture.hs:48-53
  R t = std::fabs(t_), tsqr = t * t, tcube = tsqr * t;
  return R(-6) * tcube * tsqr + R(15) * tcube * t - R(10) * tcube + R(1);
}
inline int tx_phi(int i) { static const int phi[12] = {3, 0, 2, 7, 4, 1, 5, 11, 8, 10, 9, 6}; return phi[i]; }  // Texture.hs:56-57
template <class R> Vec<R> tx_grad(int i) {  // Texture.hs:59-64: the 12 edge directions, x outermost in the comprehension
  static const int g[12][3] = {{-1, -1, 0}, {-1, 0, -1}, {-1, 0, 1}, {-1, 1, 0}, {0, -1, -1}, {0, -1, 1}, {0, 1, -1}, {0, 1, 1}, {1, -1, 0}, {1, 0, -1}, {1, 0, 1}, {1, 1, 0}};
  return Vec<R>{R(g[i][0]), R(g[i][1]), R(g[i][2])};
}
template <class R> Vec<R> tx_gamma(long i, long j, long k) {  // Texture.hs:66-71
  int a = tx_phi((int)(std::labs(k) % 12));
  int b = tx_phi((int)(std::labs(j + a) % 12));
  int c = tx_phi((int)(std::labs(i + b) % 12));
  return tx_grad<R>(c);
}
template <class R> R tx_knot(long i, long j, long k, Vec<R> v) {  // Texture.hs:73-76
  return tx_omega(v.x) * tx_omega(v.y) * tx_omega(v.z) * vdot(tx_gamma<R>(i, j, k), v);
}
template <class R> R tx_noise(Vec<R> p) {  // Texture.hs:92-107
  long i = (long)std::floor(p.x), j = (long)std::floor(p.y), k = (long)std::floor(p.z);
  R u = p.x - R(i), v = p.y - R(j), w = p.z - R(k);
  return tx_knot(i, j, k, Vec<R>{u, v, w}) + tx_knot(i + 1, j, k, Vec<R>{u - 1, v, w}) + tx_knot(i, j + 1, k, Vec<R>{u, v - 1, w}) +
         tx_knot(i, j, k + 1, Vec<R>{u, v, w - 1}) + tx_knot(i + 1, j + 1, k, Vec<R>{u - 1, v - 1, w}) + tx_knot(i + 1, j, k + 1, Vec<R>{u - 1, v, w - 1}) +
         tx_knot(i, j + 1, k + 1, Vec<R>{u, v - 1, w - 1}) + tx_knot(i + 1, j + 1, k + 1, Vec<R>{u - 1, v - 1, w - 1});
}
template <class R> R tx_perlin(Vec<R> v) { return (tx_noise(v) + 1) * R(0.5); }  // Texture.hs:109-117 (the range errors cannot fire: |noise| <= 1)
template <class R> R tx_wave(int fn, R x) {  // Texture.hs:11-25
  R offset = x - std::floor(x);
  if (fn == W_STRIPE_SQUARE) return offset < R(0.5) ? R(0) : R(1);
  if (fn == W_STRIPE_TRIANGLE) return offset < R(0.5) ? offset * 2 : 2 - offset * 2;
  return std::sin(x * 2 * R(M_PI)) * R(0.5) + R(0.5);
}
// the weight of a Blend whose closure was `\_ hit -> Blend a b (f (pos hit))`: perlin (vscale pos s) (TestScene.hs:216)
// or (stripe axis wave) pos = wave (vdot pos axis) (Texture.hs:35-41, TestScene.hs:227)
template <class R> R tx_weight(int fn, const R* wp, R constant, Vec<R> pos) {
  if (fn == W_CONST) return constant;
  if (fn == W_PERLIN) return tx_perlin(vscale(pos, wp[0]));
  return tx_wave(fn, vdot(pos, Vec<R>{wp[0], wp[1], wp[2]}));
}

enum MatKind { M_SURFACE = 0, M_REFLECT = 1, M_REFRACT = 2, M_LAYERS = 3, M_BLEND = 4, M_WARP = 5 };
template <class R> struct Material {  // Shader.hs:43-52
  int kind = M_SURFACE;
  Color<R> color{0, 0, 0};
  R alpha = 1, amb = 0, kd = 0, ks = 0, shine = 0;  // Surface
  R refl = 0, refr = 0, ior = 1;                    // Reflect / Refract
  std::vector<int> kids;                            // AdditiveLayers
  int ma = -1, mb = -1; R weight = 0;               // Blend
  int wfn = W_CONST; R wp[4] = {0, 0, 0, 0};        // Blend weight as a solid texture function of the hit position
  // Warp frame scene' lights' xfm (Shader.hs:47-50).  The closure `xfm :: Ray -> Rayint -> Ray` in its one shape in the
  // reference (the portal, TestScene.hs:166-172): \ray hit -> xfm_ray M (Ray (pos hit) (vnorm (dir ray))).
  // wscene == nullptr: the scene the material is used in (the portal looks into geom'' itself, TestScene.hs:181).
  SP<R> wframe, wscene;
  std::vector<Light<R>> wlights;
  Xfm<R> wxfm;
};
template <class R> struct Camera { Vec<R> pos, fwd, up, right; };  // Scene.hs:35
template <class R> Camera<R> camera(Vec<R> pos, Vec<R> at, Vec<R> up, R angle) {  // Scene.hs:48-57
  Vec<R> fwd = vnorm(vsub(at, pos));
  Vec<R> right = vnorm(vcross(up, fwd));
  Vec<R> up_ = vnorm(vcross(fwd, right));
  R cam_scale = std::tan((R(M_PI) / 180) * (angle / 2));
  return {pos, fwd, vscale(up_, cam_scale), vscale(right, cam_scale)};
}

template <class R> struct Scene {
  SP<R> root;
  std::vector<Light<R>> lights;
  std::vector<Material<R>> mats;
  Camera<R> cam;
};
template <class R> struct LightSample { Color<R> c; Vec<R> dir; };

template <class R> struct Tracer {
  const Scene<R>& S;
  explicit Tracer(const Scene<R>& s) : S(s) {}
  using M = Math<R>;
  typedef std::vector<Light<R>> Lights;

  // mpreshade, Shader.hs:65-80 (Q18): `ls` and `sld` are the lights and the solid of the trace in progress
  std::vector<LightSample<R>> preshade(const Lights& ls, const Solid<R>& sld, const Rayint<R>& ri) const {
    std::vector<LightSample<R>> out;
    if (!ri.hit) return out;
    for (auto& L : ls) {
      Vec<R> lvec = vsub(L.pos, ri.pos);
      if (vdot(lvec, ri.norm) < 0) continue;
      R llen = vlen(lvec);
      Vec<R> ldir = vscale(lvec, 1 / llen);
      if (llen > L.rad) continue;
      if (L.shadow) {
        tls_counters().rays_shadow++;
        if (sld.shadow(Ray<R>{vscaleadd(ri.pos, ri.norm, M::delta()), ldir}, llen - (2 * M::delta()))) continue;
      }
      out.push_back({cscale(L.col, 1 / (llen * llen)), ldir});
    }
    return out;
  }
  // mpostshade, Shader.hs:82-184 (Q17).  `lights` is evaluated lazily like ctxb in Trace.hs:63.
  struct Lazy { bool done = false; std::vector<LightSample<R>> v; };
  ColorA<R> postshade(const Lights& ls, const Solid<R>& sld, Lazy& lz, int mat, const Ray<R>& ray, const Rayint<R>& ri, int recurs) const {
    if (!ri.hit) return {0, 0, 0, 0};
    const Material<R>& m = S.mats.at(mat);
    Vec<R> dir = ray.d, n = ri.norm, p = ri.pos;
    Vec<R> eyedir = vinvert(dir);
    switch (m.kind) {
      case M_SURFACE: {
        if (!lz.done) { lz.v = preshade(ls, sld, ri); lz.done = true; }
        Color<R> ambient = cscale(m.color, m.amb);
        Color<R> direct{0, 0, 0};
        for (auto& l : lz.v) {
          Vec<R> halfangle = bisect(l.dir, eyedir);
          R ldotn = M::fmax(0, vdot(l.dir, n));
          R blinn;
          if (m.ks <= M::delta()) blinn = 0;
          else {
            R b = M::fmax(0, std::pow(vdot(halfangle, n), m.shine) * ldotn);
            blinn = std::isnan(b) ? R(0) : b;
          }
          R diffuse = vdot(l.dir, n);
          direct = cadd(direct, cscale(l.c, (blinn * m.ks) + (diffuse * m.kd)));
        }
        Color<R> c = cadd(ambient, direct);
        return {c.r, c.g, c.b, m.alpha};
      }
      case M_REFLECT: {
        if ((m.refl > 0) && (recurs > 0)) {
          Vec<R> outdir = reflect(dir, n);
          tls_counters().rays_secondary += (recurs - 1 > 0);
          ColorA<R> c = trace(ls, sld, Ray<R>{vscaleadd(p, outdir, M::delta()), outdir}, M::infinity(), recurs - 1);
          return {c.r, c.g, c.b, c.a * m.refl};
        }
        return {0, 0, 0, 1};
      }
      case M_REFRACT: {
        if ((m.refl > 0 || m.refr > 0) && (recurs > 0)) {
          Vec<R> outdir = reflect(dir, n);
          tls_counters().rays_secondary += (recurs - 1 > 0);
          ColorA<R> cr = trace(ls, sld, Ray<R>{vscaleadd(p, outdir, M::delta()), outdir}, M::infinity(), recurs - 1);
          R eta = (vdot(n, eyedir) > 0) ? m.ior : 1 / m.ior;
          R c1 = vdot(dir, n);
          R cs2 = 1 - (eta * eta) * (1 - (c1 * c1));
          ColorA<R> ct{0, 0, 0, 1};  // ca_black on total internal reflection
          if (!(cs2 < 0)) {
            Vec<R> t = vadd(vscale(dir, eta), vscale(n, eta * c1 - std::sqrt(cs2)));
            tls_counters().rays_secondary += (recurs - 1 > 0);
            ct = trace(ls, sld, Ray<R>{vscaleadd(p, t, M::delta()), t}, M::infinity(), recurs - 1);
          }
          return {cr.r * m.refl + ct.r * m.refr, cr.g * m.refl + ct.g * m.refr, cr.b * m.refl + ct.b * m.refr, cr.a * m.refl + ct.a * m.refr};
        }
        return {0, 0, 0, 0};
      }
      case M_WARP: {  // Shader.hs:157-175: the frame through the hit's own (local) ray, then the other scene up to the frame's depth
        Rayint<R> fint, wint;
        tls_counters().rays_secondary += (recurs - 1 > 0);
        ColorA<R> fcolor = trace(ls, *m.wframe, ri.riray, M::infinity(), recurs - 1, &fint);
        Ray<R> wray = xfm_ray(m.wxfm, Ray<R>{ri.pos, vnorm(ray.d)});  // the portal's closure (TestScene.hs:166-172)
        tls_counters().rays_secondary += (recurs - 1 > 0);
        ColorA<R> wcolor = trace(m.wlights, m.wscene ? *m.wscene : *S.root, wray, ridepth(fint), recurs - 1, &wint);
        return (ridepth(fint) < ridepth(wint)) ? fcolor : wcolor;
      }
      case M_LAYERS: {
        std::vector<ColorA<R>> cs;
        for (int k : m.kids) cs.push_back(postshade(ls, sld, lz, k, ray, ri, recurs));
        return casum(cs);
      }
      case M_BLEND: {
        ColorA<R> ca = postshade(ls, sld, lz, m.ma, ray, ri, recurs);
        ColorA<R> cb = postshade(ls, sld, lz, m.mb, ray, ri, recurs);
        return caweight(ca, cb, tx_weight<R>(m.wfn, m.wp, m.weight, ri.pos));
      }
    }
    return {0, 0, 0, 0};
  }
  // trace, Trace.hs:59-82 (Q16); *ri_out receives the trace's Rayint
  ColorA<R> trace(const Lights& ls, const Solid<R>& sld, const Ray<R>& ray, R depth, int recurs, Rayint<R>* ri_out = nullptr) const {
    if (recurs == 0) { if (ri_out) *ri_out = Rayint<R>(); return {0, 0, 0, 0}; }
    Rayint<R> ri = sld.rayint(ray, depth, TexList());
    if (ri_out) *ri_out = ri;
    if (!ri.hit) return {0, 0, 0, 0};  // mmissshade, Shader.hs:186-187
    Lazy lz;
    ColorA<R> acc{0, 0, 0, 0};
    for (int k = 0; k < ri.tex.n; k++) {
      if (acc.a + M::delta() >= 1) break;  // opaque, Trace.hs:50-51
      acc = cafold(acc, postshade(ls, sld, lz, ri.tex.v[k], ray, ri, recurs));
    }
    return acc;
  }
  // `Trace.trace lights shader sld ray depth maxdepth` over the scene's own root and lights (Glome.hs:33)
  ColorA<R> trace(const Ray<R>& ray, R depth, int recurs, Rayint<R>* ri_out = nullptr) const { return trace(S.lights, *S.root, ray, depth, recurs, ri_out); }
};

// ---------------------------------------------------------------------------------------
// GlomeView/Glome.hs: pixel loops
// ---------------------------------------------------------------------------------------
template <class R> void getCoordsf(int width, int height, R xf, R yf, R& xc, R& yc) {  // Glome.hs:119-140 (Q19)
  R widthf = R(width), heightf = R(height);
  xc = (((xf / widthf) * 2) - 1) * (widthf / heightf);
  yc = -(((yf / heightf) * 2) - 1);
}
template <class R> struct TColor { R r, g, b, a, d; };  // Glome.hs:153

struct RenderParams {
  int width = 720, height = 480;
  int mode = 0;            // 0 = renderTile (1 ray/px, Glome.hs:162-176), 1 = renderTileSubsample (:226-323)
  int blocksize = 65;      // Glome.hs:116
  int maxdepth = 3;        // Glome.hs:25
  int fog = 0;             // 1: renderTile stores r + d/400 (Glome.hs:174, Q20); 0: the get_color tuple
  double thresholds[4] = {0.14, 0.15, 0.16, 0.18};  // Glome.hs:221-224
  int tile_first = 0, tile_stride = 1;              // multi-rank sharding: render tiles t = first, first+stride, ...
};

inline std::vector<std::pair<int, int>> chunk(int size, int blocksize) {  // Glome.hs:371-377
  std::vector<std::pair<int, int>> o;
  int pos = 0;
  for (;;) {
    if (pos + blocksize >= size) { o.push_back({pos, size - pos}); break; }
    o.push_back({pos, blocksize});
    pos += blocksize;
  }
  return o;
}
inline double cap1(double x) { return x >= 1 ? 1 - 0.0001 : x; }  // Glome.hs:98-101
inline uint32_t rgbf(double r, double g, double b) {               // Glome.hs:107-110 (no clamp below 0)
  long long v = (long long)std::floor(cap1(r) * 256) * (256 * 256) + (long long)std::floor(cap1(g) * 256) * 256 + (long long)std::floor(cap1(b) * 256);
  return (uint32_t)v;
}

template <class R> struct Renderer {
  const Scene<R>& S;
  Tracer<R> T;
  RenderParams P;
  Renderer(const Scene<R>& s, const RenderParams& p) : S(s), T(s), P(p) {}

  TColor<R> get_color(R x, R y) const {  // get_rayint + get_color_normal, Glome.hs:27-33, 53-55
    const Camera<R>& c = S.cam;
    Vec<R> dir = vnorm(vadd3(c.fwd, vscale(c.right, -x), vscale(c.up, y)));
    tls_counters().rays_primary++;
    Rayint<R> ri;
    ColorA<R> col = T.trace(Ray<R>{c.pos, dir}, Math<R>::infinity(), P.maxdepth, &ri);
    return {col.r, col.g, col.b, col.a, ridepth(ri)};
  }
  static R cCmp(const TColor<R>& a, const TColor<R>& b) {  // Glome.hs:179-189
    auto diff = [](R x, R y) { return Math<R>::fabs_(y - x); };
    auto muldiff = [](R x, R y) -> R { if (x == 0 && y == 0) return 0; return (x > y) ? (x / y) - 1 : (y / x) - 1; };
    return diff(a.r, b.r) + diff(a.g, b.g) + diff(a.b, b.b) + diff(a.a, b.a) + muldiff(a.d, b.d);
  }
  static TColor<R> cAvg(const TColor<R>& a, const TColor<R>& b, const TColor<R>& c, const TColor<R>& d) {  // Glome.hs:191-197
    return {(a.r + b.r + c.r + d.r) * R(0.25), (a.g + b.g + c.g + d.g) * R(0.25), (a.b + b.b + c.b + d.b) * R(0.25),
            (a.a + b.a + c.a + d.a) * R(0.25), (a.d + b.d + c.d + d.d) * R(0.25)};
  }
  static TColor<R> cAvg2(const TColor<R>& a, const TColor<R>& b) {  // Glome.hs:199-205
    return {(a.r + b.r) * R(0.5), (a.g + b.g) * R(0.5), (a.b + b.b) * R(0.5), (a.a + b.a) * R(0.5), (a.d + b.d) * R(0.5)};
  }
  TColor<R> decide(R threshold, R xf, R yf, const TColor<R>& a, const TColor<R>& b, const TColor<R>& c, const TColor<R>& d) const {  // Glome.hs:213-219
    R variance = Math<R>::fmax(cCmp(a, c), cCmp(b, d));
    if (variance > threshold) return get_color(xf, yf);
    return cAvg(a, b, c, d);
  }
  TColor<R> decide_int(R threshold, int x, int y, const TColor<R>& a, const TColor<R>& b, const TColor<R>& c, const TColor<R>& d) const {  // Glome.hs:207-210
    R xf, yf;
    getCoordsf<R>(P.width, P.height, R(x), R(y), xf, yf);
    return decide(threshold, xf, yf, a, b, c, d);
  }
  // renderTile, Glome.hs:162-176
  void renderTile(int xtmin, int ytmin, int tw, int th, std::vector<TColor<R>>& v) const {
    v.resize((size_t)tw * th);
    for (int i = 0; i < tw * th; i++) {
      R xc, yc;
      getCoordsf<R>(P.width, P.height, R(xtmin + (i % tw)), R(ytmin + (i / tw)), xc, yc);
      TColor<R> c = get_color(xc, yc);
      if (P.fog) c.r = c.r + (c.d / 400);
      v[i] = c;
    }
  }
  // renderTileSubsample, Glome.hs:226-323 (Q21)
  void renderTileSubsample(int xtmin, int ytmin, int tw, int th, std::vector<TColor<R>>& out) const {
    const TColor<R> blank{0, 0, 0, 0, Math<R>::infinity()};
    std::vector<TColor<R>> v((size_t)tw * th, blank);
    auto getc = [&](int x, int y) -> TColor<R> {
      if (x >= xtmin && x < xtmin + tw && y >= ytmin && y < ytmin + th) return v[(x - xtmin) + (y - ytmin) * tw];
      return blank;
    };
    auto putc = [&](std::vector<TColor<R>>& vv, int x, int y, const TColor<R>& c) { vv[(x - xtmin) + (y - ytmin) * tw] = c; };
    R t1 = R(P.thresholds[0]), t2 = R(P.thresholds[1]), t3 = R(P.thresholds[2]), t4 = R(P.thresholds[3]);
    for (int x = xtmin; x <= xtmin + tw - 1; x += 2)
      for (int y = ytmin; y <= ytmin + th - 1; y += 2)
        if (((x - xtmin) + (y - ytmin)) % 4 == 0) {
          R xf, yf;
          getCoordsf<R>(P.width, P.height, R(x), R(y), xf, yf);
          putc(v, x, y, get_color(xf, yf));
        }
    for (int x = xtmin; x <= xtmin + tw - 1; x += 2)
      for (int y = ytmin; y <= ytmin + th - 1; y += 2)
        if (((x - xtmin) + (y - ytmin)) % 4 == 2) {
          TColor<R> a = getc(x - 2, y), b = getc(x, y + 2), c = getc(x + 2, y), d = getc(x, y - 2);
          putc(v, x, y, decide_int(t1, x, y, a, b, c, d));
        }
    for (int x = xtmin + 1; x <= xtmin + tw - 1; x += 2)
      for (int y = ytmin + 1; y <= ytmin + th - 1; y += 2) {
        TColor<R> a = getc(x - 1, y - 1), b = getc(x + 1, y - 1), c = getc(x + 1, y + 1), d = getc(x - 1, y + 1);
        putc(v, x, y, decide_int(t2, x, y, a, b, c, d));
      }
    for (int x = xtmin; x <= xtmin + tw - 1; x++)
      for (int y = ytmin; y <= ytmin + th - 1; y++)
        if (((x - xtmin) + (y - ytmin)) % 2 == 1) {
          TColor<R> a = getc(x - 1, y), b = getc(x, y + 1), c = getc(x + 1, y), d = getc(x, y - 1);
          putc(v, x, y, decide_int(t3, x, y, a, b, c, d));
        }
    std::vector<TColor<R>> v2((size_t)tw * th, blank);
    for (int x = xtmin; x <= xtmin + tw - 1; x++)
      for (int y = ytmin; y <= ytmin + th - 1; y++) {
        TColor<R> a = getc(x, y), b = getc(x, y + 1), c = getc(x + 1, y + 1), d = getc(x + 1, y);
        R xf, yf;
        getCoordsf<R>(P.width, P.height, R(x) + R(0.5), R(y) + R(0.5), xf, yf);
        TColor<R> color = decide(t4, xf, yf, a, b, c, d);
        if (x == xtmin + tw - 1) {
          if (y == ytmin + th - 1) putc(v2, x, y, color);
          else putc(v2, x, y, cAvg2(color, cAvg2(a, b)));
        } else {
          if (y == ytmin + th - 1) putc(v2, x, y, cAvg2(color, cAvg2(a, d)));
          else putc(v2, x, y, cAvg2(color, cAvg(a, b, c, d)));
        }
      }
    out.swap(v2);
  }

  struct TileRect { int x, y, w, h; };
  std::vector<TileRect> tiles() const {  // renderTiles, Glome.hs:379-384: x-major order
    std::vector<TileRect> t;
    for (auto& xc : chunk(P.width, P.blocksize))
      for (auto& yc : chunk(P.height, P.blocksize)) t.push_back({xc.first, yc.first, xc.second, yc.second});
    return t;
  }
  // renderTiles + blitTile, Glome.hs:379-386, 353-358.  out5: width*height*5 doubles (r,g,b,a,d per pixel, row-major);
  // packed: width*height (or null).  nthreads tiles in parallel like parMap.  Tiles not owned (sharding) are left untouched.
  // max_tiles > 0 renders only the first max_tiles owned tiles (bounded CPU-baseline sample).
  Counters render(double* out5, uint32_t* packed, int nthreads, int max_tiles = 0) const {
    std::vector<TileRect> ts = tiles();
    std::vector<int> owned;
    for (int k = P.tile_first; k < (int)ts.size(); k += P.tile_stride) owned.push_back(k);
    if (max_tiles > 0 && (int)owned.size() > max_tiles) owned.resize(max_tiles);
    std::atomic<size_t> next{0};
    std::vector<Counters> cs((size_t)std::max(1, nthreads));
    auto work = [&](int tid) {
      Counters before = tls_counters();
      tls_counters() = Counters();
      std::vector<TColor<R>> v;
      for (;;) {
        size_t k = next.fetch_add(1);
        if (k >= owned.size()) break;
        const TileRect& t = ts[owned[k]];
        if (P.mode == 0) renderTile(t.x, t.y, t.w, t.h, v); else renderTileSubsample(t.x, t.y, t.w, t.h, v);
        for (int i = 0; i < t.w * t.h; i++) {
          int px = t.x + (i % t.w), py = t.y + (i / t.w);
          size_t o = (size_t)py * P.width + px;
          const TColor<R>& c = v[i];
          if (out5) { out5[o * 5 + 0] = c.r; out5[o * 5 + 1] = c.g; out5[o * 5 + 2] = c.b; out5[o * 5 + 3] = c.a; out5[o * 5 + 4] = c.d; }
          if (packed) packed[o] = rgbf(double(c.r) * double(c.a), double(c.g) * double(c.a), double(c.b) * double(c.a));
        }
      }
      cs[tid] = tls_counters();
      tls_counters() = before;
    };
    if (nthreads <= 1) work(0);
    else {
      std::vector<std::thread> th;
      for (int i = 0; i < nthreads; i++) th.emplace_back(work, i);
      for (auto& t : th) t.join();
    }
    Counters tot;
    for (auto& c : cs) tot.add(c);
    return tot;
  }
};

}  // namespace glo
