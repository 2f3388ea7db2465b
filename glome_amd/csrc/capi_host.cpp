// capi_host.cpp -- host half of the C ABI (include/glome_hip.h): the scene builder (one call per glome
// constructor), the transform helpers and the host-side inspection calls.  No HIP here.
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/glome_hip.h"
#include "capi_shared.hpp"
#include "show_format.hpp"
#include "tiles.hpp"

using namespace glome;

namespace {
D3 d3(const double* p) { return D3{p[0], p[1], p[2]}; }
Xf xf_from(const double* m) { Xf x; for (int k = 0; k < 12; k++) { x.f.m[k] = m[k]; x.i.m[k] = m[12 + k]; } return x; }
void xf_to(const Xf& x, double* out) { for (int k = 0; k < 12; k++) { out[k] = x.f.m[k]; out[12 + k] = x.i.m[k]; } }

template <class F> int32_t guard(glome_sb* sb, F f) {
  if (!sb) return GLOME_E_INVALID;
  try { return (int32_t)f(); }
  catch (const scene_error& e) { sb->err = e.what(); return GLOME_E_SCENE; }
  catch (const limit_error& e) { sb->err = e.what(); return GLOME_E_LIMIT; }
  catch (const std::invalid_argument& e) { sb->err = e.what(); return GLOME_E_INVALID; }
  catch (const std::exception& e) { sb->err = e.what(); return GLOME_E_INVALID; }
}
template <class F> int xguard(F f) {
  try { f(); return 0; } catch (...) { return GLOME_E_SCENE; }
}
std::vector<int> ids_of(const int32_t* ids, int n) {
  if (n < 0 || (n > 0 && !ids)) throw std::invalid_argument("bad id list");
  return std::vector<int>(ids, ids + n);
}
}  // namespace

extern "C" {

void glome_render_params_default(glome_render_params* p) {
  if (!p) return;
  memset(p, 0, sizeof(*p));
  p->width = 720; p->height = 480;         // Glome.hs:112-113
  p->mode = GLOME_MODE_TILE;
  p->blocksize = 65;                       // Glome.hs:116
  p->maxdepth = 3;                         // Glome.hs:25
  p->thresholds[0] = 0.14f; p->thresholds[1] = 0.15f; p->thresholds[2] = 0.16f; p->thresholds[3] = 0.18f;  // Glome.hs:221-224
  p->tile_first = 0; p->tile_stride = 1; p->rank0_share_pct = 0;
}

int glome_xfm_translate(const double v[3], double out[24]) { return xguard([&] { xf_to(xf_translate(d3(v)), out); }); }
int glome_xfm_scale(const double v[3], double out[24]) { return xguard([&] { xf_to(xf_scale(d3(v)), out); }); }
int glome_xfm_rotate(const double axis[3], double angle, double out[24]) { return xguard([&] { xf_to(xf_rotate(d3(axis), angle), out); }); }
int glome_xfm_xyz_to_uvw(const double u[3], const double v[3], const double w[3], double out[24]) { return xguard([&] { xf_to(xf_xyz_to_uvw(d3(u), d3(v), d3(w)), out); }); }
int glome_xfm_compose(const double* xfms, int n, double out[24]) {
  return xguard([&] { std::vector<Xf> xs; for (int k = 0; k < n; k++) xs.push_back(xf_from(xfms + 24 * k)); xf_to(xf_compose(xs), out); });
}
int glome_camera_lookat(const double pos[3], const double at[3], const double up[3], double angle_deg, glome_camera* out) {  // Scene.hs:48-57
  if (!out) return GLOME_E_INVALID;
  D3 p = d3(pos);
  D3 fwd = normalize(d3(at) - p);
  D3 right = normalize(cross(d3(up), fwd));
  D3 up_ = normalize(cross(fwd, right));
  double cam_scale = std::tan((M_PI / 180) * (angle_deg / 2));
  D3 u = up_ * cam_scale, r = right * cam_scale;
  const D3 v[4] = {p, fwd, u, r};
  float* o = out->pos;
  for (int k = 0; k < 4; k++) { o[3 * k] = (float)v[k].x; o[3 * k + 1] = (float)v[k].y; o[3 * k + 2] = (float)v[k].z; }
  return 0;
}

int glome_tex_words(void) { return GLOME_TEX_WORDS; }  // int32 words per ray of glome_rayint_batch's texture-stack output

glome_sb* glome_sb_new(void) { return new glome_sb(); }
void glome_sb_free(glome_sb* sb) { delete sb; }
const char* glome_sb_last_error(const glome_sb* sb) { return sb ? sb->err.c_str() : "null builder"; }

int32_t glome_sb_sphere(glome_sb* sb, const double c[3], double r) { return guard(sb, [&] { return sb->graph.sphere(d3(c), r); }); }
int32_t glome_sb_triangle(glome_sb* sb, const double p[9]) { return guard(sb, [&] { return sb->graph.triangle(d3(p), d3(p + 3), d3(p + 6)); }); }
int32_t glome_sb_trianglenorm(glome_sb* sb, const double p[9], const double n[9]) {
  return guard(sb, [&] { return sb->graph.trianglenorm(d3(p), d3(p + 3), d3(p + 6), d3(n), d3(n + 3), d3(n + 6)); });
}
int32_t glome_sb_box(glome_sb* sb, const double a[3], const double b[3]) { return guard(sb, [&] { return sb->graph.box(d3(a), d3(b)); }); }
int32_t glome_sb_plane(glome_sb* sb, const double pt[3], const double n[3]) { return guard(sb, [&] { return sb->graph.plane(d3(pt), d3(n)); }); }
int32_t glome_sb_plane_offset(glome_sb* sb, const double n[3], double off) { return guard(sb, [&] { return sb->graph.plane_offset(d3(n), off); }); }
int32_t glome_sb_disc(glome_sb* sb, const double pos[3], const double n[3], double r) { return guard(sb, [&] { return sb->graph.disc(d3(pos), d3(n), r); }); }
int32_t glome_sb_cylinder(glome_sb* sb, const double p1[3], const double p2[3], double r) { return guard(sb, [&] { return sb->graph.cylinder(d3(p1), d3(p2), r); }); }
int32_t glome_sb_cone(glome_sb* sb, const double p1[3], double r1, const double p2[3], double r2) { return guard(sb, [&] { return sb->graph.cone(d3(p1), r1, d3(p2), r2); }); }
int32_t glome_sb_group(glome_sb* sb, const int32_t* ids, int n) { return guard(sb, [&] { return sb->graph.group(ids_of(ids, n)); }); }
int32_t glome_sb_transform(glome_sb* sb, int32_t id, const double* xfms, int n) {
  return guard(sb, [&] {
    if (n < 0 || (n > 0 && !xfms)) throw std::invalid_argument("bad transform list");
    std::vector<Xf> xs;
    for (int k = 0; k < n; k++) xs.push_back(xf_from(xfms + 24 * k));
    return sb->graph.transform(id, xs);
  });
}
int32_t glome_sb_difference(glome_sb* sb, int32_t a, int32_t b) { return guard(sb, [&] { return sb->graph.difference(a, b); }); }
int32_t glome_sb_difference_retexture(glome_sb* sb, int32_t a, int32_t b) { return guard(sb, [&] { return sb->graph.difference(a, b, true); }); }
int32_t glome_sb_intersection(glome_sb* sb, const int32_t* ids, int n) { return guard(sb, [&] { return sb->graph.intersection(ids_of(ids, n)); }); }
int32_t glome_sb_bih(glome_sb* sb, const int32_t* ids, int n) { return guard(sb, [&] { return sb->graph.bih(ids_of(ids, n)); }); }
int32_t glome_sb_mesh(glome_sb* sb, const double* verts, int nv, const double* norms, int nn, const int32_t* tris, int nt, const int32_t* mats, int nm) {
  return guard(sb, [&] {
    if (nv < 0 || nn < 0 || nt < 0 || nm < 0 || (nv && !verts) || (nn && !norms) || (nt && !tris) || (nm && !mats)) throw std::invalid_argument("bad mesh arrays");
    std::vector<D3> V, N;
    for (int k = 0; k < nv; k++) V.push_back(d3(verts + 3 * k));
    for (int k = 0; k < nn; k++) N.push_back(d3(norms + 3 * k));
    std::vector<MeshTri> T;
    for (int k = 0; k < nt; k++) { const int32_t* t = tris + 8 * k; T.push_back(MeshTri{t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]}); }
    return sb->graph.mesh(std::move(V), std::move(N), std::move(T), std::vector<int>(mats, mats + nm));
  });
}
int32_t glome_sb_tex(glome_sb* sb, int32_t id, int32_t material) { return guard(sb, [&] { return sb->graph.wrap(K_TEX, id, material); }); }
int32_t glome_sb_tag(glome_sb* sb, int32_t id) { return guard(sb, [&] { return sb->graph.wrap(K_TAG, id); }); }
int32_t glome_sb_noshadow(glome_sb* sb, int32_t id) { return guard(sb, [&] { return sb->graph.wrap(K_NOSHADOW, id); }); }
int32_t glome_sb_onlyshadow(glome_sb* sb, int32_t id) { return guard(sb, [&] { return sb->graph.wrap(K_ONLYSHADOW, id); }); }
int32_t glome_sb_bound_object(glome_sb* sb, int32_t a, int32_t b) { return guard(sb, [&] { return sb->graph.bound_object(a, b, false); }); }
int32_t glome_sb_innerbound(glome_sb* sb, int32_t a, int32_t b) { return guard(sb, [&] { return sb->graph.bound_object(a, b, true); }); }
int32_t glome_sb_flatten_transform(glome_sb* sb, int32_t id) { return guard(sb, [&] { return sb->graph.flatten_transform_item(id); }); }
int32_t glome_sb_tolist(glome_sb* sb, int32_t id) { return guard(sb, [&] { return sb->graph.tolist_node(id); }); }
int32_t glome_sb_list_items(glome_sb* sb, int32_t id, int32_t* out, int32_t cap) {
  return guard(sb, [&] {
    std::vector<int> items;
    sb->graph.tolist(id, items);
    for (size_t k = 0; k < items.size() && (int32_t)k < cap && out; k++) out[k] = items[k];
    return (int)items.size();
  });
}

int32_t glome_sb_material_surface(glome_sb* sb, const double color[3], double alpha, double amb, double kd, double ks, double shine) {
  return guard(sb, [&] { Mat m; m.kind = MAT_SURFACE; m.color[0] = color[0]; m.color[1] = color[1]; m.color[2] = color[2]; m.alpha = alpha; m.amb = amb; m.kd = kd; m.ks = ks; m.shine = shine; return sb->graph.add_mat(m); });
}
int32_t glome_sb_material_reflect(glome_sb* sb, double refl) { return guard(sb, [&] { Mat m; m.kind = MAT_REFLECT; m.refl = refl; return sb->graph.add_mat(m); }); }
int32_t glome_sb_material_refract(glome_sb* sb, double refl, double refr, double ior) {
  return guard(sb, [&] { Mat m; m.kind = MAT_REFRACT; m.refl = refl; m.refr = refr; m.ior = ior; return sb->graph.add_mat(m); });
}
int32_t glome_sb_material_layers(glome_sb* sb, const int32_t* mats, int n) {
  return guard(sb, [&] { Mat m; m.kind = MAT_LAYERS; m.kids = ids_of(mats, n); for (int k : m.kids) sb->graph.check_mat(k); return sb->graph.add_mat(m); });
}
int32_t glome_sb_material_blend(glome_sb* sb, int32_t a, int32_t b, double weight) {
  return guard(sb, [&] { sb->graph.check_mat(a); sb->graph.check_mat(b); Mat m; m.kind = MAT_BLEND; m.a = a; m.b = b; m.weight = weight; return sb->graph.add_mat(m); });
}
int32_t glome_sb_material_blend_fn(glome_sb* sb, int32_t a, int32_t b, int32_t weight_fn, const double* params4) {
  return guard(sb, [&] {
    sb->graph.check_mat(a); sb->graph.check_mat(b);
    if (weight_fn < GLOME_WEIGHT_PERLIN || weight_fn > GLOME_WEIGHT_STRIPE_SINE || !params4) throw std::invalid_argument("bad weight function");
    Mat m; m.kind = MAT_BLEND; m.a = a; m.b = b; m.wfn = weight_fn;
    for (int k = 0; k < 4; k++) m.wp[k] = params4[k];
    return sb->graph.add_mat(m);
  });
}

int32_t glome_sb_material_warp(glome_sb* sb, int32_t frame, int32_t scene, const glome_light* lights, int nlights, const double xfm[24]) {
  return guard(sb, [&] {
    if (nlights < 0 || nlights > kMaxLights || (nlights > 0 && !lights) || !xfm) throw std::invalid_argument("bad Warp lights / transform");
    sb->graph.at(frame);
    if (scene >= 0) sb->graph.at(scene);
    Mat m; m.kind = MAT_WARP; m.wframe = frame; m.wscene = scene < 0 ? -1 : scene; m.wxf = xf_from(xfm);
    for (int k = 0; k < nlights; k++) {
      WarpLight L;
      for (int q = 0; q < 3; q++) { L.pos[q] = lights[k].pos[q]; L.color[q] = lights[k].color[q]; }
      L.rad = lights[k].rad; L.shadow = lights[k].shadow != 0;
      m.wlights.push_back(L);
    }
    return sb->graph.add_mat(m);
  });
}

int glome_sb_primcount(glome_sb* sb, int32_t id, long out3[3]) {
  return guard(sb, [&] { out3[0] = out3[1] = out3[2] = 0; sb->graph.primcount(id, out3); return 0; });
}
int glome_sb_bound(glome_sb* sb, int32_t id, double out6[6]) {
  return guard(sb, [&] { Box3 b = sb->graph.bound(id); out6[0] = b.lo.x; out6[1] = b.lo.y; out6[2] = b.lo.z; out6[3] = b.hi.x; out6[4] = b.hi.y; out6[5] = b.hi.z; return 0; });
}
// ---- NFF / SPD scenes (GlomeTrace/Data/Glome/Spd.hs:89-254) ----
// A restatement of the reference's Read instances: whitespace-separated tokens, `#` starts a comment that runs to the
// end of the line (lexcr / lexignore, Spd.hs:13-30); statements v / l / b / f / s / c / p / pp; everything else ends
// the parse like accum_rss's fall-through (Spd.hs:212-242).
namespace {
struct NffLex {
  const char* p;
  explicit NffLex(const char* t) : p(t) {}
  void skip() {
    for (;;) {
      while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') p++;
      if (*p == '#') { while (*p && *p != '\n') p++; continue; }
      return;
    }
  }
  bool word(std::string& w) {  // the next token, not consumed unless accepted by the caller
    skip();
    const char* q = p;
    while (*q && !(*q == ' ' || *q == '\t' || *q == '\r' || *q == '\n')) q++;
    w.assign(p, q);
    return !w.empty();
  }
  void take(const std::string& w) { p += w.size(); }
  bool number(double& v) {  // reads :: Flt -- a token that parses completely as a number
    std::string w;
    if (!word(w)) return false;
    char* end = nullptr;
    double x = std::strtod(w.c_str(), &end);
    if (end == w.c_str() || *end != 0) return false;
    if (!((w[0] >= '0' && w[0] <= '9') || ((w[0] == '-') && w.size() > 1 && w[1] >= '0' && w[1] <= '9'))) return false;  // Haskell wants a leading digit
    take(w); v = x;
    return true;
  }
  bool vec(D3& v) { const char* save = p; if (number(v.x) && number(v.y) && number(v.z)) return true; p = save; return false; }
  bool keyword(const char* k) { std::string w; if (word(w) && w == k) { take(w); return true; } return false; }
};
}  // namespace

// `show geom` text (show_format.hpp)
long glome_sb_show(glome_sb* sb, int32_t id, char* buf, long cap) {
  if (!sb) return GLOME_E_INVALID;
  try {
    std::string o;
    ShowWriter W{sb->graph, o};
    W.item(id);
    if (buf && cap > 0) { size_t n = std::min<size_t>(o.size(), (size_t)cap - 1); memcpy(buf, o.data(), n); buf[n] = 0; }
    return (long)o.size();
  } catch (const std::exception& e) { sb->err = e.what(); return GLOME_E_INVALID; }
}
long glome_sb_show_tex_materials(glome_sb* sb, int32_t id, int32_t* mats, long cap) {
  if (!sb) return GLOME_E_INVALID;
  try {
    std::string o;
    std::vector<int> m;
    ShowWriter W{sb->graph, o, &m};
    W.item(id);
    for (long k = 0; mats && k < cap && k < (long)m.size(); k++) mats[k] = m[(size_t)k];
    return (long)m.size();
  } catch (const std::exception& e) { sb->err = e.what(); return GLOME_E_INVALID; }
}
int32_t glome_sb_load_show(glome_sb* sb, const char* text, const int32_t* tex_materials, int32_t n_tex_materials, int32_t default_material, int32_t* n_tex) {
  return guard(sb, [&] {
    if (!text) throw std::invalid_argument("null show text");
    if (n_tex_materials < 0 || (n_tex_materials > 0 && !tex_materials)) throw std::invalid_argument("bad material list");
    std::vector<int> mats(tex_materials, tex_materials + n_tex_materials);
    int nt = 0;
    int root = load_show(sb->graph, text, strlen(text), mats.data(), (int)mats.size(), default_material, &nt);
    if (n_tex) *n_tex = nt;
    return root;
  });
}

int32_t glome_sb_load_nff(glome_sb* sb, const char* text, double cam_from_at_up_angle[10], double* light_pos_rgb, int32_t max_lights, int32_t* n_lights,
                          double bg_rgb[3]) {
  return guard(sb, [&] {
    if (!text) throw std::invalid_argument("null NFF text");
    Graph& G = sb->graph;
    NffLex L(text);
    std::vector<int> groups;               // `tex (bih prims) fill`, in order of appearance
    std::vector<std::array<double, 6>> lights;
    bool have_cam = false, have_bg = false;
    double cam[10] = {0}, bg[3] = {0, 0, 0};
    int fill = -1;
    std::vector<int> prims;
    auto close_group = [&] { if (fill >= 0) groups.push_back(G.wrap(K_TEX, G.bih(prims), fill)); prims.clear(); };
    for (;;) {
      std::string w;
      if (!L.word(w)) break;
      const char* save = L.p;
      if (w == "v") {  // Spd.hs:89-107
        L.take(w);
        D3 from, at, up; double angle, skipn;
        std::string tok;
        bool ok = L.keyword("from") && L.vec(from) && L.keyword("at") && L.vec(at) && L.keyword("up") && L.vec(up) && L.keyword("angle") && L.number(angle) &&
                  L.keyword("hither") && L.word(tok);
        if (ok) { L.take(tok); ok = L.keyword("resolution") && L.word(tok); }
        if (ok) { L.take(tok); ok = L.word(tok); }
        if (ok) L.take(tok);
        (void)skipn;
        if (!ok) { L.p = save; break; }
        double c[10] = {from.x, from.y, from.z, at.x, at.y, at.z, up.x, up.y, up.z, angle};
        memcpy(cam, c, sizeof(c)); have_cam = true;  // the last camera of the file wins (accum_rss conses)
      } else if (w == "l") {  // Spd.hs:131-139: colour optional, default white
        L.take(w);
        D3 pos, c{1, 1, 1};
        if (!L.vec(pos)) { L.p = save; break; }
        D3 cc; if (L.vec(cc)) c = cc;
        lights.push_back({pos.x, pos.y, pos.z, c.x, c.y, c.z});
      } else if (w == "b") {  // Spd.hs:123-128
        L.take(w);
        D3 c; if (!L.vec(c)) { L.p = save; break; }
        bg[0] = c.x; bg[1] = c.y; bg[2] = c.z; have_bg = true;
      } else if (w == "f") {  // Spd.hs:141-153: Surface clr (1 - T) 0 kd ks shine
        L.take(w);
        D3 c; double kd, ks, shine, trans, ior;
        if (!(L.vec(c) && L.number(kd) && L.number(ks) && L.number(shine) && L.number(trans) && L.number(ior))) { L.p = save; break; }
        close_group();
        Mat m; m.kind = MAT_SURFACE; m.color[0] = c.x; m.color[1] = c.y; m.color[2] = c.z; m.alpha = 1 - trans; m.amb = 0; m.kd = kd; m.ks = ks; m.shine = shine;
        fill = G.add_mat(m);
      } else if (w == "s" || w == "c" || w == "p" || w == "pp") {  // Spd.hs:165-182
        if (fill < 0) break;  // a primitive needs a fill before it (readsSpdTextureGroup reads the texture first)
        L.take(w);
        if (w == "s") { D3 c; double r; if (!(L.vec(c) && L.number(r))) { L.p = save; break; } prims.push_back(G.sphere(c, r)); }
        else if (w == "c") { D3 a, b; double ra, rb; if (!(L.vec(a) && L.number(ra) && L.vec(b) && L.number(rb))) { L.p = save; break; } prims.push_back(G.cone(a, ra, b, rb)); }
        else {
          double n;
          if (!L.number(n)) { L.p = save; break; }  // the count is read and ignored: vertices are taken while they parse
          std::vector<D3> vs, ns;
          for (;;) {
            D3 v, nn;
            if (!L.vec(v)) break;
            if (w == "pp") { if (!L.vec(nn)) break; ns.push_back(nn); }
            vs.push_back(v);
          }
          std::vector<int> fan;  // triangles / trianglesnorms: a fan around the first vertex (Triangle.hs:28-42)
          for (size_t k = 1; k + 1 < vs.size(); k++)
            fan.push_back(w == "p" ? G.triangle(vs[0], vs[k], vs[k + 1]) : G.trianglenorm(vs[0], vs[k], vs[k + 1], ns[0], ns[k], ns[k + 1]));
          prims.push_back(G.group(fan));
        }
      } else break;
    }
    close_group();
    if (!have_cam) throw scene_error("NFF: no camera (v) statement");  // readsSpdScene's pattern needs one (Spd.hs:251)
    if (!have_bg) throw scene_error("NFF: no background (b) statement");
    std::reverse(groups.begin(), groups.end());  // accum_rss conses: the scene's list is in reverse order of appearance
    std::reverse(lights.begin(), lights.end());
    if (cam_from_at_up_angle) memcpy(cam_from_at_up_angle, cam, sizeof(cam));
    if (bg_rgb) memcpy(bg_rgb, bg, sizeof(bg));
    if (n_lights) *n_lights = (int32_t)lights.size();
    for (int k = 0; k < (int)lights.size() && k < max_lights && light_pos_rgb; k++) memcpy(light_pos_rgb + 6 * k, lights[k].data(), 6 * sizeof(double));
    return G.bih(groups);  // SPD (bih prims) lights cam bgc, Spd.hs:252
  });
}

long glome_sb_bih_dump(glome_sb* sb, int32_t id, long cap, double* lsplit, double* rsplit, int* axis, int* nleaf, int32_t* leaf_prims, long cap_prims) {
  if (!sb) return GLOME_E_INVALID;
  try {
    const Node& n = sb->graph.at(id);
    if (n.kind != K_BIH) throw std::invalid_argument("not a Bih");
    const BihTree& T = *n.bih;
    long cnt = 0, np = 0;
    std::vector<int> st{0};
    while (!st.empty()) {  // preorder, left before right
      const BihTree::Node& bn = T.nodes[st.back()];
      st.pop_back();
      if (cnt < cap) {
        axis[cnt] = bn.leaf ? -1 : bn.axis; nleaf[cnt] = bn.leaf ? (int)bn.items.size() : 0;
        lsplit[cnt] = bn.leaf ? 0 : bn.lsplit; rsplit[cnt] = bn.leaf ? 0 : bn.rsplit;
      }
      cnt++;
      if (bn.leaf) { for (int it : bn.items) { if (np < cap_prims) leaf_prims[np] = it; np++; } }
      else { st.push_back(bn.right); st.push_back(bn.left); }
    }
    return cnt;
  } catch (const std::exception& e) { sb->err = e.what(); return GLOME_E_INVALID; }
}

int glome_tiles_layout(const glome_render_params* P, int tile_first, int tile_stride, int32_t* xywh_base, int cap) {
  if (!P || P->width <= 0 || P->height <= 0 || P->blocksize <= 0 || tile_stride <= 0 || tile_first < 0) return GLOME_E_INVALID;
  std::vector<DTile> t; uint32_t w; int64_t px;
  owned_tiles(P->width, P->height, P->blocksize, tile_first, tile_stride, P->rank0_share_pct, t, w, px);
  for (size_t k = 0; k < t.size() && (int)k < cap; k++) {
    xywh_base[5 * k] = t[k].x; xywh_base[5 * k + 1] = t[k].y; xywh_base[5 * k + 2] = t[k].w; xywh_base[5 * k + 3] = t[k].h; xywh_base[5 * k + 4] = (int32_t)t[k].pix_base;
  }
  return (int)t.size();
}

}  // extern "C"
