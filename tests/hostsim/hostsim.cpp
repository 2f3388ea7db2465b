// hostsim.cpp -- TEST INFRASTRUCTURE ONLY.  Compiles the per-ray device headers (rt_device.hpp / rt_generic.hpp)
// as plain host C++ so their logic -- and the flattened layout they index -- can be checked against the oracle on a
// machine without a GPU, before a kernel is ever launched.  It is not part of libglome_hip.so, nothing in glome_amd
// imports it, and bench.py never times it.  GPU parity tests (-m gpu) go through the real C ABI.
#include <cstring>
#include <set>

#include <cstdlib>
#include "host_shim.hpp"  // first: GD / GHD, the one-lane wave, LaneStack -- then the device headers compile for the host
#include "../../glome_amd/csrc/capi_shared.hpp"
#include "../../glome_amd/csrc/flatten.hpp"
#include "../../glome_amd/csrc/rt_device.hpp"
#include "../../glome_amd/csrc/rt_generic.hpp"

using namespace glome;

struct SimScene {
  FlatScene F;
  DScene D;
  std::string err;
};

template <bool FAITHFUL, bool COUNT, bool FULL_> struct HostFlatTier {
  static constexpr bool FULL = FULL_;
  static constexpr bool WARP = false;
  const DScene& S;
  const DLight* lights;
  int nlights;
  LaneStack stk;
  Cnt cnt;
  unsigned int err = 0;
  HitG closest(const Ray& r, float tmax) { HitG ch; Cand c = closest_flat<FAITHFUL, COUNT, CLS_EVERY>(S, r, tmax, stk, cnt, true, &ch, &err); return finalize_flat<CLS_EVERY>(S, r, c, &ch); }
  bool occluded(const Ray& r, float d, uint32_t = 0) { return occluded_flat<COUNT, CLS_EVERY>(S, r, d, stk, cnt, true, &err); }
  // on the host a wave is one lane: the packet code runs as a single-ray traversal
  HitG closest_wave(const Ray& r, float tmax, bool valid, uint32_t = 0) { HitG ch; Cand c = closest_flat<FAITHFUL, COUNT, CLS_EVERY, true>(S, r, tmax, stk, cnt, valid, &ch, &err); return valid ? finalize_flat<CLS_EVERY>(S, r, c, &ch) : hit_miss(); }
  bool occluded_wave(const Ray& r, float d, bool valid) { return occluded_flat<COUNT, CLS_EVERY, true>(S, r, d, stk, cnt, valid, &err); }
};
struct HostStack {
  // a deliberately small "LDS" part so the overflow columns are exercised by the CPU suite too
  static constexpr int LDS_PART = 6;
  uint32_t node[kFlatStack]; float nearv[kFlatStack], farv[kFlatStack];
  uint32_t ovf[3 * kFlatStack];
  LaneStack lane(int cap) {
    LaneStack s; s.node = node; s.nearv = nearv; s.farv = farv; s.cap = LDS_PART < cap ? LDS_PART : cap;
    s.ovf = ovf; s.ovf_cap = cap - s.cap;
    return s;
  }
};
struct HostGenericTier {
  static constexpr bool FULL = true;
  static constexpr bool WARP = true;
  const DScene& S;
  const DLight* lights;
  int nlights;
  Cnt cnt;
  unsigned int err = 0;
  uint32_t vm[kVmWords];
  HostStack pkmem;
  LaneStack pk;  // the packet stack of vm_run's packet service (cap 0: none); packets(): after construction
  void packets() { pk = pkmem.lane((int)S.pk_generic_cap); if (S.pk_generic_cap == 0) { pk.cap = 0; pk.ovf_cap = 0; } }
  HitG closest(const Ray& r, float tmax, uint32_t root) { return vm_closest<true>(S, cnt, err, vm, pk.cap > 0 ? &pk : (LaneStack*)nullptr, r, tmax, root); }
  bool occluded(const Ray& r, float d, uint32_t root) { return vm_occluded<true>(S, cnt, err, vm, pk.cap > 0 ? &pk : (LaneStack*)nullptr, r, d, root); }
  HitG closest(const Ray& r, float tmax) { return closest(r, tmax, S.root_rec); }
  bool occluded(const Ray& r, float d) { return occluded(r, d, S.root_rec); }
  HitG closest_wave(const Ray& r, float tmax, bool valid, uint32_t root) { return valid ? closest(r, tmax, root) : hit_miss(); }
  bool occluded_wave(const Ray& r, float d, bool valid) { return valid && occluded(r, d); }
};


// the product's launch rule (glome_device.hip launch_render): a flat-tier frame of a scene with a Refract material, traced
// deeper than the primary ray, is traversed as the reference traverses (its transmitted rays are not unit length)
static bool refract_scene(const SimScene* s) {
  for (uint32_t k = 0; k < s->D.n_mats; k++) { uint32_t kind; memcpy(&kind, &s->D.mats[3 * k].x, 4); if (kind == DM_REFRACT) return true; }
  return false;
}

extern "C" {
void* hostsim_commit(glome_sb* sb, int root, char* errbuf, int cap) {
  SimScene* s = new SimScene();
  try {
    Flattener fl(sb_graph(sb), s->F);
    fl.run(root);
  } catch (std::exception& e) { snprintf(errbuf, cap, "%s", e.what()); delete s; return nullptr; }
  FlatScene& F = s->F;
  DScene& D = s->D;
  D.recs = F.recs.data(); D.spheres = F.spheres.data(); D.tris = F.tris.data(); D.tripairs = F.tripairs.data(); D.trinorms = F.trinorms.data(); D.boxes = F.boxes.data();
  D.planes = F.planes.data(); D.discs = F.discs.data(); D.quadrics = F.quadrics.data(); D.xfms = F.xfms.data(); D.bihhdr = F.bihhdr.data();
  D.bihnodes = F.bihnodes.data(); D.pknodes = F.pknodes.data(); D.pknodes_bytes = (uint32_t)(F.pknodes.size() * sizeof(F4)); D.meshhdr = F.meshhdr.data(); D.meshnodes = F.meshnodes.data(); D.mtris = F.mtris.data();
  D.mtrimeta = F.mtrimeta.data(); D.mats = F.mats.data(); D.wlights = F.wlights.data(); D.matkids = F.matkids.data(); D.entries = F.entries.data();
  D.n_entries = F.tier == 0 ? (uint32_t)F.entries.size() : 0; D.root_rec = F.root_rec; D.tier = F.tier; D.n_mats = (uint32_t)sb_graph(sb).mats.size(); D.tex_bits = F.tex_bits;
  D.pk_generic_cap = (F.tier != 0 && F.max_sphere_bih_depth > 0) ? (uint32_t)std::min(kGenericPacketStack, std::max(4, F.max_sphere_bih_depth)) : 0u;
  if (getenv("GLOME_DEBUG_NO_GENERIC_PACKETS")) D.pk_generic_cap = 0;  // (the product's switch, glome_device.hip glome_scene_commit)
  return s;
}
void hostsim_free(void* s) { delete (SimScene*)s; }
int hostsim_info(void* sv, int* out) {  // tier, nesting, max_bih_depth, max_mesh_depth, n_entries, n_recs
  SimScene* s = (SimScene*)sv;
  out[0] = (int)s->F.tier; out[1] = s->F.nesting_depth; out[2] = s->F.max_bih_depth; out[3] = s->F.max_mesh_depth;
  out[4] = (int)s->D.n_entries; out[5] = (int)s->F.recs.size();
  return 0;
}
// tier: -1 = the scene's own tier, 0 = flat (must be legal), 1 = generic; analysis: faithful traversal + counters
int hostsim_rayint(void* sv, int tier, int analysis, size_t n, const float* ox, const float* oy, const float* oz, const float* dx, const float* dy,
                   const float* dz, const float* tmax, float* t, int* prim, float* nrm, int* tex8, unsigned long long* counters) {
  SimScene* s = (SimScene*)sv;
  if (tier < 0) tier = (int)s->D.tier;
  if (tier == 0 && s->D.tier != 0) return -1;
  HostStack hs;
  Cnt total;
  unsigned int err = 0;
  for (size_t i = 0; i < n; i++) {
    Ray r; r.o = v3(ox[i], oy[i], oz[i]); r.d = v3(dx[i], dy[i], dz[i]);
    HitG h;
    if (tier == 0) {
      if (analysis & 1) { HostFlatTier<true, true, false> T{s->D, nullptr, 0, hs.lane(kFlatStack), Cnt()}; h = (analysis & 2) ? T.closest_wave(r, tmax[i], true) : T.closest(r, tmax[i]); total.bih += T.cnt.bih; total.prim += T.cnt.prim; total.mesh += T.cnt.mesh; }
      else if (!unit_length(r.d)) { HostFlatTier<true, false, false> T{s->D, nullptr, 0, hs.lane(kFlatStack), Cnt()}; h = T.closest(r, tmax[i]); }  // (k_rayint_batch_flat's rule for a caller's non-unit ray)
      else { HostFlatTier<false, false, false> T{s->D, nullptr, 0, hs.lane(kFlatStack), Cnt()}; h = (analysis & 2) ? T.closest_wave(r, tmax[i], true) : T.closest(r, tmax[i]); }
    } else {
      HostGenericTier T{s->D, nullptr, 0, Cnt()}; T.packets();
      h = T.closest(r, tmax[i]);
      err |= T.err; total.bih += T.cnt.bih; total.prim += T.cnt.prim; total.mesh += T.cnt.mesh;
    }
    t[i] = h.hit ? h.t : -1.0f;
    if (prim) prim[i] = h.hit ? (int)h.uid : -1;
    if (nrm) { nrm[3 * i] = h.n.x; nrm[3 * i + 1] = h.n.y; nrm[3 * i + 2] = h.n.z; }
    if (tex8) for (int k = 0; k < 8; k++) tex8[8 * i + k] = (h.hit && k * (int)s->D.tex_bits < 64) ? (int)((h.tex >> (s->D.tex_bits * k)) & ((1ull << s->D.tex_bits) - 1)) - 1 : -1;
  }
  if (counters) { counters[0] = total.bih; counters[1] = total.mesh; counters[2] = total.prim; }
  return err ? -2 : 0;
}
int hostsim_shadow(void* sv, int tier, size_t n, const float* ox, const float* oy, const float* oz, const float* dx, const float* dy, const float* dz,
                   const float* tmax, unsigned char* occ) {
  SimScene* s = (SimScene*)sv;
  if (tier < 0) tier = (int)s->D.tier;
  if (tier == 0 && s->D.tier != 0) return -1;
  HostStack hs;
  unsigned int err = 0;
  for (size_t i = 0; i < n; i++) {
    Ray r; r.o = v3(ox[i], oy[i], oz[i]); r.d = v3(dx[i], dy[i], dz[i]);
    if (tier == 0) { HostFlatTier<false, false, false> T{s->D, nullptr, 0, hs.lane(kFlatStack), Cnt()}; occ[i] = T.occluded(r, tmax[i]); }
    else { HostGenericTier T{s->D, nullptr, 0, Cnt()}; T.packets(); occ[i] = T.occluded(r, tmax[i]); err |= T.err; }
  }
  return err ? -2 : 0;
}
int hostsim_inside(void* sv, size_t n, const float* px, const float* py, const float* pz, unsigned char* in) {
  SimScene* s = (SimScene*)sv;
  unsigned int err = 0;
  uint32_t vm[kVmWords];
  for (size_t i = 0; i < n; i++) in[i] = vm_inside(s->D, err, vm, 0, s->D.recs[s->D.root_rec], v3(px[i], py[i], pz[i]));
  return err ? -2 : 0;
}
// whole-frame, 1 ray per pixel (renderTile); cam = 12 floats, lights = nl x 8 floats (pos3 col3 rad shadow)
int hostsim_render(void* sv, int tier, const float* cam, const float* lights, int nl, int width, int height, int maxdepth, float* out5, unsigned long long* counters) {
  SimScene* s = (SimScene*)sv;
  if (tier < 0) tier = (int)s->D.tier;
  if (tier == 0 && s->D.tier != 0) return -1;
  DCamera C; memcpy(&C, cam, sizeof(C));
  DLight L[kMaxLights];
  for (int i = 0; i < nl; i++) { memcpy(L[i].pos, lights + 8 * i, 12); memcpy(L[i].color, lights + 8 * i + 3, 12); L[i].rad = lights[8 * i + 6]; L[i].shadow = lights[8 * i + 7] != 0; }
  HostStack hs;
  Cnt total;
  unsigned int err = 0;
  const bool exact = refract_scene(s) && maxdepth > 1;
  for (int py = 0; py < height; py++)
    for (int px = 0; px < width; px++) {
      float xc, yc;
      get_coordsf(width, height, (float)px, (float)py, xc, yc);
      Ray ray = primary_ray(C, xc, yc);
      HitG h; CA c;
      if (tier == 0 && exact) { HostFlatTier<true, false, true> T{s->D, L, nl, hs.lane(kFlatStack), Cnt()}; c = trace_primary(T, ray, kInf, maxdepth, true, &h); err |= T.err; total.shadow += T.cnt.shadow; total.secondary += T.cnt.secondary; }
      else if (tier == 0) { HostFlatTier<false, false, true> T{s->D, L, nl, hs.lane(kFlatStack), Cnt()}; c = trace_primary(T, ray, kInf, maxdepth, true, &h); err |= T.err; total.shadow += T.cnt.shadow; total.secondary += T.cnt.secondary; }
      else { HostGenericTier T{s->D, L, nl, Cnt()}; T.packets(); c = trace_primary(T, ray, kInf, maxdepth, true, &h); err |= T.err; total.shadow += T.cnt.shadow; total.secondary += T.cnt.secondary; }
      float* o = out5 + ((size_t)py * width + px) * 5;
      o[0] = c.r; o[1] = c.g; o[2] = c.b; o[3] = c.a; o[4] = h.hit ? h.t : kInf;
    }
  if (counters) { counters[0] = (unsigned long long)width * height; counters[1] = total.shadow; counters[2] = total.secondary; }
  return err ? -2 : 0;
}

// Diagnostic: per-lane traversal work inside each 8x8 pixel block (what one wave executes), for primary rays and for
// the shadow rays they spawn.  out[block][lane][k]: k = 0 nodes, 1 prim tests (primary); 2 nodes, 3 prims (shadow; -1 = no shadow ray)
int hostsim_block_work(void* sv, const float* cam, const float* light_pos, int width, int height, int* out) {
  SimScene* s = (SimScene*)sv;
  if (s->D.tier != 0) return -1;
  DCamera C; memcpy(&C, cam, sizeof(C));
  HostStack hs;
  int bw = width / 8, bh = height / 8;
  for (int by = 0; by < bh; by++) for (int bx = 0; bx < bw; bx++) for (int lane = 0; lane < 64; lane++) {
    int px = bx * 8 + (lane & 7), py = by * 8 + (lane >> 3);
    float xc, yc; get_coordsf(width, height, (float)px, (float)py, xc, yc);
    Ray ray = primary_ray(C, xc, yc);
    HostFlatTier<false, true, false> T{s->D, nullptr, 0, hs.lane(kFlatStack), Cnt()};
    HitG h = T.closest(ray, kInf);
    int* o = out + ((size_t)(by * bw + bx) * 64 + lane) * 4;
    o[0] = (int)T.cnt.bih; o[1] = (int)T.cnt.prim; o[2] = -1; o[3] = -1;
    if (h.hit) {
      V3 lvec = v3(light_pos[0], light_pos[1], light_pos[2]) - h.p;
      if (!(vdot(lvec, h.n) < 0)) {
        float llen = sqrtf(vdot(lvec, lvec));
        Ray sr; sr.o = vscaleadd(h.p, h.n, kDel); sr.d = lvec * (1.0f / llen);
        HostFlatTier<false, true, false> T2{s->D, nullptr, 0, hs.lane(kFlatStack), Cnt()};
        T2.occluded(sr, llen - 2 * kDel);
        o[2] = (int)T2.cnt.bih; o[3] = (int)T2.cnt.prim;
      }
    }
  }
  return 0;
}

// renderTileSubsample through the device headers' pass helpers (ss_candidate / ss_neighbours / ccmp / cavg / blend),
// executed sequentially tile by tile.  Mirrors subsample_tile() in glome_device.hip.
int hostsim_render_subsample(void* sv, int tier, const float* cam, const float* lights, int nl, int width, int height, int maxdepth, int blocksize,
                             const float* thresholds, float* out5, unsigned long long* counters) {
  SimScene* s = (SimScene*)sv;
  if (tier < 0) tier = (int)s->D.tier;
  if (tier == 0 && s->D.tier != 0) return -1;
  DCamera C; memcpy(&C, cam, sizeof(C));
  DLight L[kMaxLights];
  for (int i = 0; i < nl; i++) { memcpy(L[i].pos, lights + 8 * i, 12); memcpy(L[i].color, lights + 8 * i + 3, 12); L[i].rad = lights[8 * i + 6]; L[i].shadow = lights[8 * i + 7] != 0; }
  HostStack hs;
  unsigned long long nprim = 0, nshadow = 0, nsec = 0;
  unsigned int err = 0;
  const bool exact = refract_scene(s) && maxdepth > 1;
  auto sample = [&](float xp, float yp) {
    float xc, yc; get_coordsf(width, height, xp, yp, xc, yc);
    Ray ray = primary_ray(C, xc, yc);
    HitG h; CA c;
    nprim++;
    if (tier == 0 && exact) { HostFlatTier<true, false, true> T{s->D, L, nl, hs.lane(kFlatStack), Cnt()}; c = trace_primary(T, ray, kInf, maxdepth, true, &h); nshadow += T.cnt.shadow; nsec += T.cnt.secondary; }
    else if (tier == 0) { HostFlatTier<false, false, true> T{s->D, L, nl, hs.lane(kFlatStack), Cnt()}; c = trace_primary(T, ray, kInf, maxdepth, true, &h); nshadow += T.cnt.shadow; nsec += T.cnt.secondary; }
    else { HostGenericTier T{s->D, L, nl, Cnt()}; T.packets(); c = trace_primary(T, ray, kInf, maxdepth, true, &h); err |= T.err; nshadow += T.cnt.shadow; nsec += T.cnt.secondary; }
    return tc(c.r, c.g, c.b, c.a, h.hit ? h.t : kInf);
  };
  for (int xt = 0; xt < width;) {
    int tw = (xt + blocksize >= width) ? width - xt : blocksize;
    for (int yt = 0; yt < height;) {
      int th = (yt + blocksize >= height) ? height - yt : blocksize;
      std::vector<TC> v((size_t)tw * th, tc_blank());
      auto getc = [&](int dx, int dy) { return (dx >= 0 && dx < tw && dy >= 0 && dy < th) ? v[(size_t)dy * tw + dx] : tc_blank(); };
      for (int pass = 1; pass <= 5; pass++) {
        float thr = pass >= 2 ? thresholds[pass - 2] : 0.0f;
        int ox[4], oy[4];
        ss_neighbours(pass, ox, oy);
        std::vector<TC> nv = v;  // passes never read what they write, so a copy is equivalent and keeps this mirror simple
        for (int i = 0; i < tw * th; i++) {
          int dx = i % tw, dy = i / tw;
          if (!ss_candidate(pass, dx, dy)) continue;
          TC a = tc_blank(), b = a, c = a, d = a, col;
          bool need = true;
          if (pass >= 2) {
            a = getc(dx + ox[0], dy + oy[0]); b = getc(dx + ox[1], dy + oy[1]); c = getc(dx + ox[2], dy + oy[2]); d = getc(dx + ox[3], dy + oy[3]);
            need = gmaxf(ccmp(a, c), ccmp(b, d)) > thr;
          }
          float off = pass == 5 ? 0.5f : 0.0f;
          col = need ? sample((float)(xt + dx) + off, (float)(yt + dy) + off) : cavg4(a, b, c, d);
          if (pass < 5) nv[i] = col;
          else {
            TC o = ss_pass5_blend(col, a, b, c, d, dx == tw - 1, dy == th - 1);
            float* q = out5 + ((size_t)(yt + dy) * width + (xt + dx)) * 5;
            q[0] = o.r; q[1] = o.g; q[2] = o.b; q[3] = o.a; q[4] = o.d;
          }
        }
        v.swap(nv);
      }
      yt += th;
    }
    xt += tw;
  }
  if (counters) { counters[0] = nprim; counters[1] = nshadow; counters[2] = nsec; }
  return err ? -2 : 0;
}
}
