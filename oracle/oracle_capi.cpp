// oracle_capi.cpp -- C ABI over the CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
// Loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
// The builder calls mirror the reference's constructor functions one for one (file:line cited
// per call) so a test can replay one scene description into both this oracle and the product
// library (include/glome_hip.h).  See glome_oracle.hpp for the parity status ("parity unpinned").
#include "glome_oracle.hpp"

#include <cstring>

using namespace glo;

namespace {

struct IOracle {
  std::string err;
  virtual ~IOracle() {}
  virtual int sphere(const double* c, double r) = 0;
  virtual int triangle(const double* p) = 0;
  virtual int trianglenorm(const double* p, const double* n) = 0;
  virtual int box(const double* a, const double* b) = 0;
  virtual int plane(const double* pt, const double* n) = 0;
  virtual int plane_offset(const double* n, double off) = 0;
  virtual int disc(const double* p, const double* n, double r) = 0;
  virtual int cylinder(const double* p1, const double* p2, double r) = 0;
  virtual int cone(const double* p1, double r1, const double* p2, double r2) = 0;
  virtual int group(const int* ids, int n) = 0;
  virtual int transform(int id, const double* xfms, int n) = 0;
  virtual int difference(int a, int b, bool useatex = true) = 0;
  virtual int intersection(const int* ids, int n) = 0;
  virtual int bih(const int* ids, int n) = 0;
  virtual int mesh(const double* verts, int nv, const double* norms, int nn, const int* tris, int nt, const int* mats, int nm) = 0;
  virtual int tex(int id, int mat) = 0;
  virtual int wrap(int id, int mode) = 0;  // 0 tag, 1 noshadow, 2 onlyshadow
  virtual int bound_object(int a, int b, int inner) = 0;
  virtual int flatten_transform(int id) = 0;
  virtual int tolist_group(int id) = 0;
  virtual int bih_tolist(int id) = 0;
  virtual int mat_surface(const double* color, double alpha, double amb, double kd, double ks, double shine) = 0;
  virtual int mat_reflect(double refl) = 0;
  virtual int mat_refract(double refl, double refr, double ior) = 0;
  virtual int mat_layers(const int* ids, int n) = 0;
  virtual int mat_blend(int a, int b, double w) = 0;
  virtual int mat_blend_fn(int a, int b, int fn, const double* wp) = 0;
  virtual int mat_warp(int frame, int scene, const double* lights8, int nl, const double* xfm24) = 0;
  virtual void set_root(int id) = 0;
  virtual void set_camera(const double* c12) = 0;
  virtual void clear_lights() = 0;
  virtual void add_light(const double* pos, const double* col, double rad, int shadow) = 0;
  virtual int rayint_batch(int root, size_t n, const double* const* od, const double* tmax, double* t, int* prim, double* pos, double* nrm, int* tex0, int* ntex) = 0;
  virtual int shadow_batch(int root, size_t n, const double* const* od, const double* tmax, uint8_t* occ) = 0;
  virtual int inside_batch(int root, size_t n, const double* const* p, uint8_t* in) = 0;
  virtual int render(const RenderParams& P, double* out5, uint32_t* packed, int nthreads, int max_tiles, uint64_t* counters) = 0;
  virtual int primcount(int id, long* out3) = 0;
  virtual int bound(int id, double* out6) = 0;
  virtual long bih_dump(int id, long cap, double* lsplit, double* rsplit, int* axis, int* nleaf, int* leaf_prims, long cap_prims) = 0;
  virtual int kind_name(int id, char* buf, int cap) = 0;
};

template <class R> struct Impl : IOracle {
  std::vector<SP<R>> nodes;
  Scene<R> scene;
  static Vec<R> V(const double* p) { return {R(p[0]), R(p[1]), R(p[2])}; }
  int add(SP<R> s, bool setuid = true) {
    int id = (int)nodes.size();
    if (setuid && s->uid < 0) const_cast<Solid<R>*>(s.get())->uid = id;
    nodes.push_back(s);
    return id;
  }
  int nextid() const { return (int)nodes.size(); }
  SP<R> get(int id) const {
    if (id < 0 || id >= (int)nodes.size()) throw std::runtime_error("bad node id");
    return nodes[id];
  }
  static Xfm<R> X(const double* m) {
    Xfm<R> x;
    for (int k = 0; k < 12; k++) { x.f.m[k] = R(m[k]); x.i.m[k] = R(m[12 + k]); }
    return x;
  }
  int sphere(const double* c, double r) override {  // Sphere.hs:15-17
    auto s = std::make_shared<Sphere<R>>();
    s->c = V(c); s->r = R(r); s->invr = R(1.0) / R(r);
    return add(s);
  }
  int triangle(const double* p) override {  // Triangle.hs:18-20
    auto t = std::make_shared<Triangle<R>>();
    t->p1 = V(p); t->p2 = V(p + 3); t->p3 = V(p + 6);
    return add(t);
  }
  int trianglenorm(const double* p, const double* n) override {  // Triangle.hs:34-35
    auto t = std::make_shared<TriangleNorm<R>>();
    t->p1 = V(p); t->p2 = V(p + 3); t->p3 = V(p + 6); t->n1 = V(n); t->n2 = V(n + 3); t->n3 = V(n + 6);
    return add(t);
  }
  int box(const double* a, const double* b) override {  // Box.hs:12-15
    using M = Math<R>;
    auto x = std::make_shared<Box<R>>();
    Vec<R> A = V(a), B = V(b);
    x->bb = {{M::fmin(A.x, B.x), M::fmin(A.y, B.y), M::fmin(A.z, B.z)}, {M::fmax(A.x, B.x), M::fmax(A.y, B.y), M::fmax(A.z, B.z)}};
    return add(x);
  }
  int plane(const double* pt, const double* n) override {  // Plane.hs:17-20
    auto p = std::make_shared<Plane<R>>();
    p->n = vnorm(V(n)); p->off = vdot(V(pt), p->n);
    return add(p);
  }
  int plane_offset(const double* n, double off) override {  // Plane.hs:24-25
    auto p = std::make_shared<Plane<R>>();
    p->n = V(n); p->off = R(off);
    return add(p);
  }
  int disc(const double* pos, const double* n, double r) override {  // Cone.hs:29-31
    auto d = std::make_shared<Disc<R>>();
    d->p = V(pos); d->n = V(n); d->r2 = R(r) * R(r);
    return add(d);
  }
  SP<R> mk_cylinder(Vec<R> p1, Vec<R> p2, R r, int uid) {  // Cone.hs:40-48
    Vec<R> axis = vsub(p2, p1);
    R len = vlen(axis);
    Vec<R> ax1 = vscale(axis, 1 / len), ax2, ax3;
    orth(ax1, ax2, ax3);
    auto c = std::make_shared<Cylinder<R>>();
    c->r = r; c->h1 = 0; c->h2 = len; c->uid = uid;
    return c->transform({xyz_to_uvw(ax2, ax3, ax1), translate(p1)}, uid);
  }
  int cylinder(const double* p1, const double* p2, double r) override { return add(mk_cylinder(V(p1), V(p2), R(r), nextid())); }
  int cone(const double* p1d, double r1d, const double* p2d, double r2d) override {  // Cone.hs:52-67 (Q7)
    Vec<R> p1 = V(p1d), p2 = V(p2d);
    R r1 = R(r1d), r2 = R(r2d);
    if (r1 < r2) { std::swap(p1, p2); std::swap(r1, r2); }
    int uid = nextid();
    if (r1 - r2 < Math<R>::delta()) return add(mk_cylinder(p1, p2, r2, uid));
    Vec<R> axis = vsub(p2, p1);
    R len = vlen(axis);
    Vec<R> ax1 = vscale(axis, 1 / len), ax2, ax3;
    orth(ax1, ax2, ax3);
    R height = (r1 * len) / (r1 - r2);
    auto c = std::make_shared<Cone<R>>();
    c->r = r1; c->clip1 = 0; c->clip2 = len; c->height = height; c->uid = uid;
    return add(c->transform({xyz_to_uvw(ax2, ax3, ax1), translate(p1)}, uid));
  }
  int group(const int* ids, int n) override {  // Solid.hs:293-296
    std::vector<SP<R>> v;
    for (int k = 0; k < n; k++) v.push_back(get(ids[k]));
    SP<R> g = glo::group<R>(v);
    return add(g, n != 1);
  }
  int transform(int id, const double* xfms, int n) override {  // Solid.hs:184,235
    std::vector<Xfm<R>> xs;
    for (int k = 0; k < n; k++) xs.push_back(X(xfms + 24 * k));
    SP<R> s = get(id)->transform(xs, nextid());
    return add(s, false);
  }
  int difference(int a, int b, bool useatex) override {  // Csg.hs:26-30
    auto d = std::make_shared<Difference<R>>();
    d->sa = get(a); d->sb = get(b); d->useatex = useatex;
    return add(d);
  }
  int intersection(const int* ids, int n) override {  // Csg.hs:64-65
    auto x = std::make_shared<Intersection<R>>();
    for (int k = 0; k < n; k++) x->slds.push_back(get(ids[k]));
    return add(x);
  }
  int bih(const int* ids, int n) override {  // Bih.hs:309-324
    std::vector<SP<R>> v;
    for (int k = 0; k < n; k++) v.push_back(get(ids[k]));
    return add(glo::bih<R>(v, nextid()));
  }
  int mesh(const double* verts, int nv, const double* norms, int nn, const int* tris, int nt, const int* mats, int nm) override {  // Mesh.hs:50-55
    auto m = std::make_shared<Mesh<R>>();
    for (int k = 0; k < nv; k++) m->verts.push_back(V(verts + 3 * k));
    for (int k = 0; k < nn; k++) m->norms.push_back(V(norms + 3 * k));
    for (int k = 0; k < nt; k++) {
      const int* t = tris + 8 * k;
      Tri T{t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]};
      auto chk = [&](int i, int lim, bool opt) { if (!((opt && i == -1) || (i >= 0 && i < lim))) throw std::runtime_error("mesh: index out of range"); };
      chk(T.a, nv, false); chk(T.b, nv, false); chk(T.c, nv, false);
      chk(T.na, nn, true); if (T.na != -1) { chk(T.nb, nn, false); chk(T.nc, nn, false); }
      chk(T.tex, nm, true);
      m->tris.push_back(T);
    }
    for (int k = 0; k < nm; k++) m->texs.push_back(mats[k]);
    m->build();
    return add(m);
  }
  int tex(int id, int mat) override {  // Tex.hs:33-34
    if (mat < 0 || mat >= (int)scene.mats.size()) throw std::runtime_error("bad material id");
    auto t = std::make_shared<Tex<R>>();
    t->s = get(id); t->tex = mat;
    return add(t);
  }
  int wrap(int id, int mode) override {  // Tex.hs:38-48
    auto p = std::make_shared<Passthru<R>>();
    p->s = get(id); p->mode = mode;
    return add(p);
  }
  int bound_object(int a, int b, int inner) override {  // Bound.hs:27-28, 116
    if (inner) { auto x = std::make_shared<InnerBound<R>>(); x->sa = get(a); x->sb = get(b); return add(x); }
    auto x = std::make_shared<Bound<R>>(); x->sa = get(a); x->sb = get(b);
    return add(x);
  }
  int flatten_transform(int id) override {  // `SolidItem (flatten_transform s)`, Solid.hs:273
    return add(make_list<R>(get(id)->flatten_transform()));
  }
  int tolist_group(int id) override {  // `tolist`, Solid.hs:177,230 -> a list solid of the flattened items
    return add(make_list<R>(get(id)->tolist()));
  }
  int bih_tolist(int id) override {  // `bih (tolist s)`, TestScene.hs:109
    return add(glo::bih<R>(get(id)->tolist(), nextid()));
  }
  int addmat(const Material<R>& m) { scene.mats.push_back(m); return (int)scene.mats.size() - 1; }
  int mat_surface(const double* c, double alpha, double amb, double kd, double ks, double shine) override {  // Shader.hs:44
    Material<R> m; m.kind = M_SURFACE; m.color = {R(c[0]), R(c[1]), R(c[2])};
    m.alpha = R(alpha); m.amb = R(amb); m.kd = R(kd); m.ks = R(ks); m.shine = R(shine);
    return addmat(m);
  }
  int mat_reflect(double refl) override { Material<R> m; m.kind = M_REFLECT; m.refl = R(refl); return addmat(m); }  // Shader.hs:45
  int mat_refract(double refl, double refr, double ior) override {                                                 // Shader.hs:46
    Material<R> m; m.kind = M_REFRACT; m.refl = R(refl); m.refr = R(refr); m.ior = R(ior);
    return addmat(m);
  }
  int mat_layers(const int* ids, int n) override {  // Shader.hs:51
    Material<R> m; m.kind = M_LAYERS;
    for (int k = 0; k < n; k++) { if (ids[k] < 0 || ids[k] >= (int)scene.mats.size()) throw std::runtime_error("bad material id"); m.kids.push_back(ids[k]); }
    return addmat(m);
  }
  int mat_blend(int a, int b, double w) override {  // Shader.hs:52
    int nm = (int)scene.mats.size();
    if (a < 0 || a >= nm || b < 0 || b >= nm) throw std::runtime_error("bad material id");
    Material<R> m; m.kind = M_BLEND; m.ma = a; m.mb = b; m.weight = R(w);
    return addmat(m);
  }
  int mat_blend_fn(int a, int b, int fn, const double* wp) override {  // TestScene.hs:214-234 (t_mottled / t_stripe)
    int nm = (int)scene.mats.size();
    if (a < 0 || a >= nm || b < 0 || b >= nm) throw std::runtime_error("bad material id");
    if (fn < W_PERLIN || fn > W_STRIPE_SINE) throw std::runtime_error("bad weight function");
    Material<R> m; m.kind = M_BLEND; m.ma = a; m.mb = b; m.wfn = fn;
    for (int k = 0; k < 4; k++) m.wp[k] = R(wp[k]);
    return addmat(m);
  }
  int mat_warp(int frame, int scn, const double* l8, int nl, const double* xfm24) override {  // Shader.hs:47-50; scn < 0: the scene the material is used in
    Material<R> m; m.kind = M_WARP; m.wframe = get(frame); m.wscene = scn < 0 ? nullptr : get(scn); m.wxfm = X(xfm24);
    for (int k = 0; k < nl; k++) { Light<R> L; L.pos = V(l8 + 8 * k); L.col = {R(l8[8 * k + 3]), R(l8[8 * k + 4]), R(l8[8 * k + 5])}; L.rad = R(l8[8 * k + 6]); L.shadow = l8[8 * k + 7] != 0; m.wlights.push_back(L); }
    return addmat(m);
  }
  void set_root(int id) override { scene.root = get(id); }
  void set_camera(const double* c) override { scene.cam = {V(c), V(c + 3), V(c + 6), V(c + 9)}; }
  void clear_lights() override { scene.lights.clear(); }
  void add_light(const double* pos, const double* col, double rad, int shadow) override {  // Shader.hs:22-23
    Light<R> L; L.pos = V(pos); L.col = {R(col[0]), R(col[1]), R(col[2])}; L.rad = R(rad); L.shadow = shadow != 0;
    scene.lights.push_back(L);
  }
  int rayint_batch(int root, size_t n, const double* const* od, const double* tmax, double* t, int* prim, double* pos, double* nrm, int* tex0, int* ntex) override {
    SP<R> s = get(root);
    for (size_t i = 0; i < n; i++) {
      Ray<R> r{{R(od[0][i]), R(od[1][i]), R(od[2][i])}, {R(od[3][i]), R(od[4][i]), R(od[5][i])}};
      Rayint<R> h = s->rayint(r, R(tmax[i]), TexList());
      t[i] = h.hit ? double(h.depth) : -1.0;
      if (prim) prim[i] = h.hit ? h.prim : -1;
      if (pos) { pos[3 * i] = h.pos.x; pos[3 * i + 1] = h.pos.y; pos[3 * i + 2] = h.pos.z; }
      if (nrm) { nrm[3 * i] = h.norm.x; nrm[3 * i + 1] = h.norm.y; nrm[3 * i + 2] = h.norm.z; }
      if (ntex) ntex[i] = h.hit ? h.tex.n : 0;
      if (tex0) for (int k = 0; k < 8; k++) tex0[8 * i + k] = (h.hit && k < h.tex.n) ? h.tex.v[k] : -1;
    }
    return 0;
  }
  int shadow_batch(int root, size_t n, const double* const* od, const double* tmax, uint8_t* occ) override {
    SP<R> s = get(root);
    for (size_t i = 0; i < n; i++) {
      Ray<R> r{{R(od[0][i]), R(od[1][i]), R(od[2][i])}, {R(od[3][i]), R(od[4][i]), R(od[5][i])}};
      occ[i] = s->shadow(r, R(tmax[i])) ? 1 : 0;
    }
    return 0;
  }
  int inside_batch(int root, size_t n, const double* const* p, uint8_t* in) override {
    SP<R> s = get(root);
    for (size_t i = 0; i < n; i++) in[i] = s->inside(Vec<R>{R(p[0][i]), R(p[1][i]), R(p[2][i])}) ? 1 : 0;
    return 0;
  }
  int render(const RenderParams& P, double* out5, uint32_t* packed, int nthreads, int max_tiles, uint64_t* counters) override {
    if (!scene.root) throw std::runtime_error("no root set");
    Renderer<R> rd(scene, P);
    Counters c = rd.render(out5, packed, nthreads, max_tiles);
    if (counters) {
      counters[0] = c.bih_nodes; counters[1] = c.mesh_nodes; counters[2] = c.prim_tests;
      counters[3] = c.rays_primary; counters[4] = c.rays_shadow; counters[5] = c.rays_secondary;
    }
    return 0;
  }
  int primcount(int id, long* o) override { Pcount p = get(id)->primcount(); o[0] = p.prims; o[1] = p.xfms; o[2] = p.bounds; return 0; }
  int bound(int id, double* o) override {
    Bbox<R> b = get(id)->bound();
    o[0] = b.p1.x; o[1] = b.p1.y; o[2] = b.p1.z; o[3] = b.p2.x; o[4] = b.p2.y; o[5] = b.p2.z;
    return 0;
  }
  // preorder dump of a Bih: per node lsplit/rsplit/axis (axis = -1 for leaves) and leaf sizes; leaf_prims = uids in order
  long bih_dump(int id, long cap, double* lsplit, double* rsplit, int* axis, int* nleaf, int* leaf_prims, long cap_prims) override {
    auto b = std::dynamic_pointer_cast<const Bih<R>>(get(id));
    if (!b) throw std::runtime_error("not a Bih");
    long n = 0, np = 0;
    std::vector<const BihNode<R>*> st{b->root.get()};
    while (!st.empty()) {
      const BihNode<R>* nd = st.back(); st.pop_back();
      if (n < cap) {
        if (nd->leaf) { axis[n] = -1; nleaf[n] = (int)nd->objs.size(); lsplit[n] = rsplit[n] = 0; }
        else { axis[n] = nd->axis; nleaf[n] = 0; lsplit[n] = nd->lsplit; rsplit[n] = nd->rsplit; }
      }
      n++;
      if (nd->leaf) { for (auto& o : nd->objs) { if (np < cap_prims) leaf_prims[np] = o->uid; np++; } }
      else { st.push_back(nd->r.get()); st.push_back(nd->l.get()); }
    }
    return n;
  }
  int kind_name(int id, char* buf, int cap) override { snprintf(buf, cap, "%s", get(id)->name()); return 0; }
};

template <class F> int guard(void* h, F f) {
  IOracle* o = (IOracle*)h;
  try { return f(o); } catch (std::exception& e) { o->err = e.what(); return -1; }
}

}  // namespace

extern "C" {
void* glo_new(int use_float) { return use_float ? (IOracle*)new Impl<float>() : (IOracle*)new Impl<double>(); }
void glo_free(void* h) { delete (IOracle*)h; }
const char* glo_last_error(void* h) { return ((IOracle*)h)->err.c_str(); }
int glo_sphere(void* h, const double* c, double r) { return guard(h, [&](IOracle* o) { return o->sphere(c, r); }); }
int glo_triangle(void* h, const double* p) { return guard(h, [&](IOracle* o) { return o->triangle(p); }); }
int glo_trianglenorm(void* h, const double* p, const double* n) { return guard(h, [&](IOracle* o) { return o->trianglenorm(p, n); }); }
int glo_box(void* h, const double* a, const double* b) { return guard(h, [&](IOracle* o) { return o->box(a, b); }); }
int glo_plane(void* h, const double* pt, const double* n) { return guard(h, [&](IOracle* o) { return o->plane(pt, n); }); }
int glo_plane_offset(void* h, const double* n, double off) { return guard(h, [&](IOracle* o) { return o->plane_offset(n, off); }); }
int glo_disc(void* h, const double* p, const double* n, double r) { return guard(h, [&](IOracle* o) { return o->disc(p, n, r); }); }
int glo_cylinder(void* h, const double* p1, const double* p2, double r) { return guard(h, [&](IOracle* o) { return o->cylinder(p1, p2, r); }); }
int glo_cone(void* h, const double* p1, double r1, const double* p2, double r2) { return guard(h, [&](IOracle* o) { return o->cone(p1, r1, p2, r2); }); }
int glo_group(void* h, const int* ids, int n) { return guard(h, [&](IOracle* o) { return o->group(ids, n); }); }
int glo_transform(void* h, int id, const double* xfms, int n) { return guard(h, [&](IOracle* o) { return o->transform(id, xfms, n); }); }
int glo_difference(void* h, int a, int b) { return guard(h, [&](IOracle* o) { return o->difference(a, b); }); }
int glo_difference_retexture(void* h, int a, int b) { return guard(h, [&](IOracle* o) { return o->difference(a, b, false); }); }  // Csg.hs:29-30
int glo_intersection(void* h, const int* ids, int n) { return guard(h, [&](IOracle* o) { return o->intersection(ids, n); }); }
int glo_bih(void* h, const int* ids, int n) { return guard(h, [&](IOracle* o) { return o->bih(ids, n); }); }
int glo_mesh(void* h, const double* verts, int nv, const double* norms, int nn, const int* tris, int nt, const int* mats, int nm) {
  return guard(h, [&](IOracle* o) { return o->mesh(verts, nv, norms, nn, tris, nt, mats, nm); });
}
int glo_tex(void* h, int id, int mat) { return guard(h, [&](IOracle* o) { return o->tex(id, mat); }); }
int glo_tag(void* h, int id) { return guard(h, [&](IOracle* o) { return o->wrap(id, 0); }); }
int glo_noshadow(void* h, int id) { return guard(h, [&](IOracle* o) { return o->wrap(id, 1); }); }
int glo_onlyshadow(void* h, int id) { return guard(h, [&](IOracle* o) { return o->wrap(id, 2); }); }
int glo_bound_object(void* h, int a, int b) { return guard(h, [&](IOracle* o) { return o->bound_object(a, b, 0); }); }
int glo_innerbound(void* h, int a, int b) { return guard(h, [&](IOracle* o) { return o->bound_object(a, b, 1); }); }
int glo_flatten_transform(void* h, int id) { return guard(h, [&](IOracle* o) { return o->flatten_transform(id); }); }
int glo_tolist(void* h, int id) { return guard(h, [&](IOracle* o) { return o->tolist_group(id); }); }
int glo_bih_tolist(void* h, int id) { return guard(h, [&](IOracle* o) { return o->bih_tolist(id); }); }
int glo_material_surface(void* h, const double* c, double alpha, double amb, double kd, double ks, double shine) {
  return guard(h, [&](IOracle* o) { return o->mat_surface(c, alpha, amb, kd, ks, shine); });
}
int glo_material_reflect(void* h, double refl) { return guard(h, [&](IOracle* o) { return o->mat_reflect(refl); }); }
int glo_material_refract(void* h, double refl, double refr, double ior) { return guard(h, [&](IOracle* o) { return o->mat_refract(refl, refr, ior); }); }
int glo_material_layers(void* h, const int* ids, int n) { return guard(h, [&](IOracle* o) { return o->mat_layers(ids, n); }); }
int glo_material_blend(void* h, int a, int b, double w) { return guard(h, [&](IOracle* o) { return o->mat_blend(a, b, w); }); }
int glo_material_blend_fn(void* h, int a, int b, int fn, const double* wp) { return guard(h, [&](IOracle* o) { return o->mat_blend_fn(a, b, fn, wp); }); }
int glo_material_warp(void* h, int frame, int scene, const double* lights8, int nl, const double* xfm24) { return guard(h, [&](IOracle* o) { return o->mat_warp(frame, scene, lights8, nl, xfm24); }); }
// the scalar field alone (tests): fn as in WeightFn, wp[4], n points
int glo_weight_fn(int fn, const double* wp, int n, const double* xyz, double* out) {
  for (int i = 0; i < n; i++) out[i] = tx_weight<double>(fn, wp, 0.0, Vec<double>{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
  return 0;
}
int glo_set_root(void* h, int id) { return guard(h, [&](IOracle* o) { o->set_root(id); return 0; }); }
int glo_set_camera(void* h, const double* c12) { return guard(h, [&](IOracle* o) { o->set_camera(c12); return 0; }); }
int glo_clear_lights(void* h) { return guard(h, [&](IOracle* o) { o->clear_lights(); return 0; }); }
int glo_add_light(void* h, const double* pos, const double* col, double rad, int shadow) {
  return guard(h, [&](IOracle* o) { o->add_light(pos, col, rad, shadow); return 0; });
}
int glo_rayint_batch(void* h, int root, size_t n, const double* ox, const double* oy, const double* oz, const double* dx, const double* dy,
                     const double* dz, const double* tmax, double* t, int* prim, double* pos, double* nrm, int* tex8, int* ntex) {
  const double* od[6] = {ox, oy, oz, dx, dy, dz};
  return guard(h, [&](IOracle* o) { return o->rayint_batch(root, n, od, tmax, t, prim, pos, nrm, tex8, ntex); });
}
int glo_shadow_batch(void* h, int root, size_t n, const double* ox, const double* oy, const double* oz, const double* dx, const double* dy,
                     const double* dz, const double* tmax, uint8_t* occ) {
  const double* od[6] = {ox, oy, oz, dx, dy, dz};
  return guard(h, [&](IOracle* o) { return o->shadow_batch(root, n, od, tmax, occ); });
}
int glo_inside_batch(void* h, int root, size_t n, const double* px, const double* py, const double* pz, uint8_t* in) {
  const double* p[3] = {px, py, pz};
  return guard(h, [&](IOracle* o) { return o->inside_batch(root, n, p, in); });
}
// params: width,height,mode,blocksize,maxdepth,fog,tile_first,tile_stride ; thresholds[4]
int glo_render(void* h, const int* iparams, const double* thresholds, double* out5, uint32_t* packed, int nthreads, int max_tiles, uint64_t* counters) {
  RenderParams P;
  P.width = iparams[0]; P.height = iparams[1]; P.mode = iparams[2]; P.blocksize = iparams[3]; P.maxdepth = iparams[4];
  P.fog = iparams[5]; P.tile_first = iparams[6]; P.tile_stride = iparams[7];
  if (thresholds) for (int k = 0; k < 4; k++) P.thresholds[k] = thresholds[k];
  return guard(h, [&](IOracle* o) { return o->render(P, out5, packed, nthreads, max_tiles, counters); });
}
int glo_primcount(void* h, int id, long* out3) { return guard(h, [&](IOracle* o) { return o->primcount(id, out3); }); }
int glo_bound(void* h, int id, double* out6) { return guard(h, [&](IOracle* o) { return o->bound(id, out6); }); }
long glo_bih_dump(void* h, int id, long cap, double* lsplit, double* rsplit, int* axis, int* nleaf, int* leaf_prims, long cap_prims) {
  IOracle* o = (IOracle*)h;
  try { return o->bih_dump(id, cap, lsplit, rsplit, axis, nleaf, leaf_prims, cap_prims); } catch (std::exception& e) { o->err = e.what(); return -1; }
}
int glo_kind_name(void* h, int id, char* buf, int cap) { return guard(h, [&](IOracle* o) { return o->kind_name(id, buf, cap); }); }

// ---- stateless helpers (double), for the known-answer tests ----
static void putx(const Xfm<double>& x, double* out24) { for (int k = 0; k < 12; k++) { out24[k] = x.f.m[k]; out24[12 + k] = x.i.m[k]; } }
static Xfm<double> getx(const double* m) { Xfm<double> x; for (int k = 0; k < 12; k++) { x.f.m[k] = m[k]; x.i.m[k] = m[12 + k]; } return x; }
int glo_xfm_translate(const double* v, double* out24) { try { putx(translate(Vec<double>{v[0], v[1], v[2]}), out24); return 0; } catch (...) { return -1; } }
int glo_xfm_scale(const double* v, double* out24) { try { putx(scale(Vec<double>{v[0], v[1], v[2]}), out24); return 0; } catch (...) { return -1; } }
int glo_xfm_rotate(const double* axis, double angle, double* out24) { try { putx(rotate(Vec<double>{axis[0], axis[1], axis[2]}, angle), out24); return 0; } catch (...) { return -1; } }
int glo_xfm_xyz_to_uvw(const double* u, const double* v, const double* w, double* out24) {
  try { putx(xyz_to_uvw(Vec<double>{u[0], u[1], u[2]}, Vec<double>{v[0], v[1], v[2]}, Vec<double>{w[0], w[1], w[2]}), out24); return 0; } catch (...) { return -1; }
}
int glo_xfm_compose(const double* xfms, int n, double* out24) {
  try { std::vector<Xfm<double>> xs; for (int k = 0; k < n; k++) xs.push_back(getx(xfms + 24 * k)); putx(compose(xs), out24); return 0; } catch (...) { return -1; }
}
void glo_xfm_point(const double* x24, const double* p, double* out3) { Vec<double> r = xfm_point(getx(x24), Vec<double>{p[0], p[1], p[2]}); out3[0] = r.x; out3[1] = r.y; out3[2] = r.z; }
void glo_camera(const double* pos, const double* at, const double* up, double angle, double* out12) {  // Scene.hs:48-57
  Camera<double> c = camera(Vec<double>{pos[0], pos[1], pos[2]}, Vec<double>{at[0], at[1], at[2]}, Vec<double>{up[0], up[1], up[2]}, angle);
  const Vec<double> v[4] = {c.pos, c.fwd, c.up, c.right};
  for (int k = 0; k < 4; k++) { out12[3 * k] = v[k].x; out12[3 * k + 1] = v[k].y; out12[3 * k + 2] = v[k].z; }
}
void glo_getcoords(int w, int h, double xf, double yf, double* out2) { getCoordsf<double>(w, h, xf, yf, out2[0], out2[1]); }
void glo_cafold(const double* a, const double* b, double* out4) {
  ColorA<double> r = cafold(ColorA<double>{a[0], a[1], a[2], a[3]}, ColorA<double>{b[0], b[1], b[2], b[3]});
  out4[0] = r.r; out4[1] = r.g; out4[2] = r.b; out4[3] = r.a;
}
void glo_caweight(const double* a, const double* b, double w, double* out4) {
  ColorA<double> r = caweight(ColorA<double>{a[0], a[1], a[2], a[3]}, ColorA<double>{b[0], b[1], b[2], b[3]}, w);
  out4[0] = r.r; out4[1] = r.g; out4[2] = r.b; out4[3] = r.a;
}
void glo_casum(const double* cs, int n, double* out4) {
  std::vector<ColorA<double>> v;
  for (int k = 0; k < n; k++) v.push_back({cs[4 * k], cs[4 * k + 1], cs[4 * k + 2], cs[4 * k + 3]});
  ColorA<double> r = casum(v);
  out4[0] = r.r; out4[1] = r.g; out4[2] = r.b; out4[3] = r.a;
}
uint32_t glo_rgbf(double r, double g, double b) { return rgbf(r, g, b); }
int glo_chunk(int size, int blocksize, int* out_pairs, int cap) {
  auto c = chunk(size, blocksize);
  for (size_t k = 0; k < c.size() && (int)k < cap; k++) { out_pairs[2 * k] = c[k].first; out_pairs[2 * k + 1] = c[k].second; }
  return (int)c.size();
}
void glo_reflect(const double* v, const double* n, double* out3) {
  Vec<double> r = reflect(Vec<double>{v[0], v[1], v[2]}, Vec<double>{n[0], n[1], n[2]});
  out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
int glo_hw_threads() { return (int)std::thread::hardware_concurrency(); }
}
