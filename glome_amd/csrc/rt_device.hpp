// rt_device.hpp -- per-ray device code for gfx950: primitive tests, BIH / Mesh traversal, the
// trace / shade recursion.  fp32 restatement of glome's formulas (reference file:line cited per
// function; Qn = SURVEY.md Appendix A).  Compare-select min/max and the exact comparison forms
// of the reference are kept, because they decide NaN / +-0 behaviour (Q1, Q2).
//
// Two tiers share the primitive and shading code:
//   flat tier    -- root = list of simple primitives / homogeneous BIHs / meshes / CSG over primitives.  Fully
//                   inlined, deferred normals.  A wave walks a triangle / sphere BIH once for its 64 rays
//                   (bih_tri_wave: wave-uniform node reference, scalar loads, per-lane intervals; the production
//                   triangle walk is hand-written, bih_packet_asm.hpp); everything else traverses per lane with its
//                   stack in LDS (one column per lane, conflict-free).
//   generic tier -- arbitrary nesting (Instance, CSG, Bound, nested BIH): rt_generic.hpp, the four class methods as loops
//                   over explicit frames in a word stack per ray (no recursion, no nesting limit but the frame memory).
#pragma once
#include "rt_types.h"

#include <type_traits>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GD __device__ __forceinline__
#define GDN __device__ __noinline__
#define GHD __host__ __device__ inline
#elif !defined(GLOME_DEVICE_HEADERS_ON_HOST)
// This is device code.  (The CPU test suite compiles these headers with g++ to check the traversal / shading logic without a
// GPU: its translation unit defines GLOME_DEVICE_HEADERS_ON_HOST after supplying GD / GHD, the bit casts, a one-lane "wave"
// (wave_ballot, uni, ld4u, ...), dir_rcp and a LaneStack of its own.  The product includes nothing of that.)
#error "rt_device.hpp is device code: compile it with hipcc"
#endif

namespace glome {

#if defined(__HIPCC__)
// ------------------------------------------------------------------ small helpers
GD float as_f(uint32_t u) { return __uint_as_float(u); }
GD uint32_t as_u(float f) { return __float_as_uint(f); }
#endif
constexpr float kInf = 1000000.0f;  // Vec.hs:14 (Q0)
constexpr float kDel = 0.0001f;     // Vec.hs:40

GD float gminf(float a, float b) { return a > b ? b : a; }  // fmin, Vec.hs:44-45 (compare-select)
GD float gmaxf(float a, float b) { return a > b ? a : b; }  // fmax, Vec.hs:48-49
GD float gmin3f(float a, float b, float c) { return a > b ? (b > c ? c : b) : (a > c ? c : a); }  // Vec.hs:52-59
GD float gmax3f(float a, float b, float c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }  // Vec.hs:62-69
GD float pmaxf(float x, float y) { return x <= y ? y : x; }  // Prelude max (Mesh.hs:167-170)
GD float pminf(float x, float y) { return x <= y ? x : y; }  // Prelude min

struct V3 { float x, y, z; };
GD V3 v3(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
GD V3 v3(const F4& f) { return v3(f.x, f.y, f.z); }
GD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
GD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
GD V3 operator*(V3 a, float f) { return v3(a.x * f, a.y * f, a.z * f); }
GD V3 vneg(V3 a) { return v3(-a.x, -a.y, -a.z); }
GD float vdot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }  // Vec.hs:185-187
GD V3 vcross(V3 a, V3 b) { return v3((a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)); }  // Vec.hs:193-198
GD V3 vscaleadd(V3 a, V3 b, float f) { return v3(a.x + (b.x * f), a.y + (b.y * f), a.z + (b.z * f)); }  // Vec.hs:302-306
GD V3 vnorm(V3 a) {  // Vec.hs:314-317
  float inv = 1.0f / sqrtf((a.x * a.x) + (a.y * a.y) + (a.z * a.z));
  return v3(a.x * inv, a.y * inv, a.z * inv);
}
GD float vcomp(V3 a, uint32_t ax) { return ax == 0 ? a.x : (ax == 1 ? a.y : a.z); }  // va, Vec.hs:167-172

struct Ray { V3 o, d; };

// A load from one of the scene's pools.  On the device the pointer is named as global memory: where the compiler cannot see that
// itself (the generic tier's kernels keep DScene in memory for their out-of-line interpreter calls, so a pool's base comes back
// from a load) it would emit flat_load -- an aperture check per access, both wait counters, no reordering against the frame stack.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> GD const T __attribute__((address_space(1)))* as_global(const T* p) { return (const T __attribute__((address_space(1)))*)(uintptr_t)p; }
GD F4 ld4(const F4* p, uint32_t i) { return as_global(p)[i]; }
GD U4 ldu4(const U4* p, uint32_t i) { return as_global(p)[i]; }
#else
GD F4 ld4(const F4* p, uint32_t i) { return p[i]; }
GD U4 ldu4(const U4* p, uint32_t i) { return p[i]; }
#endif

// ------------------------------------------------------------------ wave-level helpers
// The packet traversal below keeps its control flow uniform across the 64 lanes of a wave: votes decide where the wave
// goes, every lane follows.
#if defined(__HIPCC__)
typedef unsigned long long LaneMask;  // one bit per lane of the wave
// min / max of values that are known not to be NaN where it matters: one instruction, without the quieting moves the
// compiler puts in front of fminf / fmaxf on values it cannot see the origin of
GD float min_nn(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
GD float max_nn(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
GD LaneMask wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
GD bool wave_any(bool p) { return __ballot(p) != 0ull; }
GD bool lane_of(LaneMask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }  // this lane's bit (m is wave-uniform)
GD uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }  // a value every lane agrees on -> SGPR
GD LaneMask uni(LaneMask m) { return (LaneMask)uni((uint32_t)m) | ((LaneMask)uni((uint32_t)(m >> 32)) << 32); }
// v of the lowest lane set in m (m != 0, wave-uniform)
GD uint32_t first_lane_value(LaneMask m, uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_ctzll(m)); }
// loads at a wave-uniform index through the constant address space with a 32-bit byte offset, so they become scalar
// loads (s_load_dwordx4 sdst, sbase, soffset); `i` counts 16-byte words and the pools are far below 4 GiB
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(4))) cf32x4;
GD F4 ld4u(const F4* p, uint32_t i) {
  const char __attribute__((address_space(4)))* b = (const char __attribute__((address_space(4)))*)(uintptr_t)p;
  f32x4 v = *(cf32x4*)(b + (uint32_t)(uni(i) << 4));
  F4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r;
}
// the 48-byte triangle record at a uniform index: one address, three loads at immediate offsets
GD void ld_tri_u(const F4* p, uint32_t tri, F4& q0, F4& q1, F4& q2) {
  const char __attribute__((address_space(4)))* b = (const char __attribute__((address_space(4)))*)(uintptr_t)p;
  cf32x4* q = (cf32x4*)(b + (uint32_t)(uni(tri) * 48u));
  f32x4 a = q[0], c = q[1], e = q[2];
  q0.x = a.x; q0.y = a.y; q0.z = a.z; q0.w = a.w; q1.x = c.x; q1.y = c.y; q1.z = c.z; q1.w = c.w; q2.x = e.x; q2.y = e.y; q2.z = e.z; q2.w = e.w;
}
#endif

// texture stacks: ids of B = DScene::tex_bits bits, innermost first, id+1 (rt_types.h)
GD TexStack tex_push(TexStack s, uint32_t mat, int B) { return (s << B) | (TexStack)(mat + 1); }  // tex:texs, Tex.hs:66
GD int tex_len(TexStack s, int B) {  // ids in s
  return s ? ((63 - __builtin_clzll(s)) >> (B == 8 ? 3 : 4)) + 1 : 0;  // (B is 8 or 16: no division)
}
GD TexStack tex_cat(TexStack a, TexStack b, int B) { const int n = tex_len(a, B); return n * B >= 64 ? a : a | (b << (n * B)); }  // a ++ b (overflow truncates; commit validates depth)
GD TexStack tex_from16(uint32_t two, int B) { return (TexStack)(two & 0xffffu) | ((TexStack)(two >> 16) << B); }  // a record's / entry's own one or two ids (16 bits each)
GD uint32_t tex_head(TexStack s, int B) { return (uint32_t)(s & ((1ull << B) - 1ull)); }

struct Cnt {  // per-lane work counters (only live when COUNT)
  uint32_t bih = 0, mesh = 0, prim = 0, shadow = 0, secondary = 0, primary = 0;
  uint32_t w_primary = 0, w_shadow = 0;  // rays counted once per WAVE where the wave acts together (scalar registers: a per-lane counter is a vector register
                                         // alive through the whole kernel); flush_counters adds both kinds
};
// `p` rays of a wave-wide step: one scalar add on the device (the host build's "wave" is one lane: its per-lane counter)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GLOME_RB_OLD_COUNT)
GD void count_wave(uint32_t&, uint32_t& per_wave, bool p) { per_wave += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); }
#else
GD void count_wave(uint32_t& per_lane, uint32_t&, bool p) { if (p) per_lane++; }
#endif

// 1 / (a ray direction component), the ONE way everywhere a slab or a split plane is clipped: a box that ends on a BIH split
// plane (or on its tree's bounds) must leave at bit-identical distances on both paths -- box_shadow's `far > d` (Box.hs:56-62)
// and every `near > far` depend on it.  Written as `1.0f / x` the compiler lowers most sites to v_rcp_f32 (the build allows
// 2.5 ulp) but loses that licence on some after hoisting them, and those come out correctly rounded: a last-bit disagreement
// between a leaf's interval and its item's slab that the generic tier's loop showed on 1 shadow ray in 10,000 (found by the
// GPU fuzz soak; the host build divides exactly everywhere and never saw it).
#if defined(__HIP_DEVICE_COMPILE__)
GD float dir_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#elif defined(__HIPCC__)
GD float dir_rcp(float x) { return 1.0f / x; }  // (hipcc's host pass: never called there)
#endif

// ------------------------------------------------------------------ slab tests (Vec.hs:725-762)
// bbclip_ub: divides by the direction itself and branches on d > 0 (not on 1/d) -- Q1: d = +0 gives
// (in, out) = (+inf, -inf) or NaN where the origin coordinate lies inside the slab.
GD void bbclip_ub(const Ray& r, V3 lo, V3 hi, float& nearv, float& farv) {
  float dxr = dir_rcp(r.d.x), dyr = dir_rcp(r.d.y), dzr = dir_rcp(r.d.z);
  float inx, outx, iny, outy, inz, outz;
  if (r.d.x > 0) { inx = (lo.x - r.o.x) * dxr; outx = (hi.x - r.o.x) * dxr; } else { inx = (hi.x - r.o.x) * dxr; outx = (lo.x - r.o.x) * dxr; }
  if (r.d.y > 0) { iny = (lo.y - r.o.y) * dyr; outy = (hi.y - r.o.y) * dyr; } else { iny = (hi.y - r.o.y) * dyr; outy = (lo.y - r.o.y) * dyr; }
  if (r.d.z > 0) { inz = (lo.z - r.o.z) * dzr; outz = (hi.z - r.o.z) * dzr; } else { inz = (hi.z - r.o.z) * dzr; outz = (lo.z - r.o.z) * dzr; }
  nearv = gmax3f(inx, iny, inz);
  farv = gmin3f(outx, outy, outz);
}
// bbclip_ub_rcp: takes reciprocals and branches on the reciprocal's sign (Vec.hs:725-741)
GD void bbclip_ub_rcp(V3 o, V3 rcp, V3 lo, V3 hi, float& nearv, float& farv) {
  float inx, outx, iny, outy, inz, outz;
  if (rcp.x > 0) { inx = (lo.x - o.x) * rcp.x; outx = (hi.x - o.x) * rcp.x; } else { inx = (hi.x - o.x) * rcp.x; outx = (lo.x - o.x) * rcp.x; }
  if (rcp.y > 0) { iny = (lo.y - o.y) * rcp.y; outy = (hi.y - o.y) * rcp.y; } else { iny = (hi.y - o.y) * rcp.y; outy = (lo.y - o.y) * rcp.y; }
  if (rcp.z > 0) { inz = (lo.z - o.z) * rcp.z; outz = (hi.z - o.z) * rcp.z; } else { inz = (hi.z - o.z) * rcp.z; outz = (lo.z - o.z) * rcp.z; }
  nearv = gmax3f(inx, iny, inz);
  farv = gmin3f(outx, outy, outz);
}

// ------------------------------------------------------------------ primitives
// Sphere.hs:20-41 (Q4).  Returns the hit distance; the normal is vnorm(p - c), computed by the caller.
// The discriminant rsqr - (csqr - vsqr) cancels catastrophically in fp32 for a small sphere far from the origin
// (csqr ~ vsqr ~ 1e2, difference ~ 1e-2).  For a unit direction csqr - vsqr = |eo - v*dir|^2 exactly, and that form
// keeps full precision, so the fp32 result lands on the reference's fp64 value.  glome's rays are unit length
// except Refract's transmitted direction (Shader.hs:141, unnormalised as written); those keep the reference form,
// which is what the fp64 path computes for them.
GD bool unit_length(V3 dir) { const float dd = vdot(dir, dir); return dd > 0.99999f && dd < 1.00001f; }
GD float sphere_disc(V3 eo, V3 dir, float v, float r) {
  if (unit_length(dir)) {
    V3 perp = eo - dir * v;
    return r * r - vdot(perp, perp);
  }
  return r * r - (vdot(eo, eo) - v * v);
}
GD bool sphere_test(const F4& s, const Ray& ray, float dist, float& t) {
  V3 eo = v3(s) - ray.o;
  float v = vdot(eo, ray.d);
  float disc = sphere_disc(eo, ray.d, v, s.w);
  if (disc < 0.0f) return false;
  float d = sqrtf(disc);
  float hitdist = ((v - d) > 0) ? (v - d) : (v + d);
  if ((hitdist < 0) || (hitdist > dist)) return false;
  t = hitdist;
  return true;
}
// Sphere.hs:51-71: the shadow form rejects early on (dist >= v - r) && (v > 0)
GD bool sphere_shadow(const F4& s, const Ray& ray, float dist) {
  V3 eo = v3(s) - ray.o;
  float v = vdot(eo, ray.d);
  if (!((dist >= (v - s.w)) && (v > 0.0f))) return false;
  float disc = sphere_disc(eo, ray.d, v, s.w);
  if (disc < 0.0f) return false;
  float d = sqrtf(disc);
  float hitdist = ((v - d) > 0) ? (v - d) : (v + d);
  return !((hitdist < 0) || (hitdist > dist));
}
// Moeller-Trumbore, Triangle.hs:45-73 / 82-107 (Q5): two sided, divisor == 0 exact, the reference's
// comparison forms.  q0 = (p1, .), q1 = (e1, .), q2 = (e2, .) from the 48-byte triangle record.
GD bool tri_test(const F4& q0, const F4& q1, const F4& q2, const Ray& ray, float dist, float& t, float& b1, float& b2) {
  V3 p1 = v3(q0), e1 = v3(q1), e2 = v3(q2);
  V3 s1 = vcross(ray.d, e2);
  float divisor = vdot(s1, e1);
  float invdivisor = 1.0f / divisor;
  V3 d = ray.o - p1;
  b1 = vdot(d, s1) * invdivisor;
  V3 s2 = vcross(d, e1);
  b2 = vdot(ray.d, s2) * invdivisor;
  t = vdot(e2, s2) * invdivisor;
  // the reference's chain `divisor == 0 || b1 < 0 || b1 > 1 || b2 < 0 || b1 + b2 > 1 || t < 0 || t > dist`, folded with
  // IEEE min/max (which return the other operand for a NaN, so every NaN case decides as the chain does)
  float lo = fminf(fminf(b1, b2), t), hi = fmaxf(b1, b1 + b2);
  return (divisor != 0) & !(lo < 0) & !(hi > 1) & !(t > dist);
}
// Box.hs:18-54 (Q1, Q6)
GD bool box_test(const F4& lo, const F4& hi, const Ray& r, float d, float& t, V3& n) {
  float dx = r.d.x, dy = r.d.y, dz = r.d.z;
  float dxr = 1.0f / dx, dyr = 1.0f / dy, dzr = 1.0f / dz;
  float inx, outx, iny, outy, inz, outz;
  if (dx > 0) { inx = (lo.x - r.o.x) * dxr; outx = (hi.x - r.o.x) * dxr; } else { inx = (hi.x - r.o.x) * dxr; outx = (lo.x - r.o.x) * dxr; }
  if (dy > 0) { iny = (lo.y - r.o.y) * dyr; outy = (hi.y - r.o.y) * dyr; } else { iny = (hi.y - r.o.y) * dyr; outy = (lo.y - r.o.y) * dyr; }
  if (dz > 0) { inz = (lo.z - r.o.z) * dzr; outz = (hi.z - r.o.z) * dzr; } else { inz = (hi.z - r.o.z) * dzr; outz = (lo.z - r.o.z) * dzr; }
  float lastin = gmax3f(inx, iny, inz), firstout = gmin3f(outx, outy, outz);
  if (lastin > firstout || firstout < 0 || lastin > d) return false;
  if (lastin < 0) {  // origin is inside: exit hit
    if (outx == firstout) n = (dx > 0) ? v3(1, 0, 0) : v3(-1, 0, 0);
    else if (outy == firstout) n = (dy > 0) ? v3(0, 1, 0) : v3(0, -1, 0);
    else n = (dz > 0) ? v3(0, 0, 1) : v3(0, 0, -1);
    t = firstout;
  } else {
    if (inx == lastin) n = (dx > 0) ? v3(-1, 0, 0) : v3(1, 0, 0);
    else if (iny == lastin) n = (dy > 0) ? v3(0, -1, 0) : v3(0, 1, 0);
    else n = (dz > 0) ? v3(0, 0, -1) : v3(0, 0, 1);
    t = lastin;
  }
  return true;
}
GD bool box_shadow(const F4& lo, const F4& hi, const Ray& r, float d) {  // Box.hs:56-62
  float nearv, farv;
  bbclip_ub(r, v3(lo), v3(hi), nearv, farv);
  return !((nearv > farv) || farv <= 0 || farv > d);
}
// Plane.hs:27-32 (Q2: a NaN passes both comparisons and is a hit)
GD bool plane_test(const F4& pl, const Ray& r, float d, float& t) {
  V3 n = v3(pl);
  float hit = -((vdot(n, r.o) - pl.w) / vdot(n, r.d));
  if (hit < 0 || hit > d) return false;
  t = hit;
  return true;
}
// Cone.hs:69-79 with plane_int_dist (Vec.hs:391-394)
GD bool disc_test(V3 point, V3 norm, float r2, const Ray& r, float d, float& t) {
  V3 newo = r.o - point;
  float dist = -(vdot(norm, newo)) / (vdot(norm, r.d));
  if (dist < 0 || dist > d) return false;
  V3 pos = vscaleadd(r.o, r.d, dist);
  V3 off = pos - point;
  if (vdot(off, off) > r2) return false;
  t = dist;
  return true;
}
// Cone.hs:104-139 (Q7): z-axis cylinder (r, h1, h2)
GD bool cyl_test(const F4& q, const Ray& ray, float d, float& t, V3& n) {
  float r = q.x, h1 = q.y, h2 = q.z;
  float ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
  float a = dx * dx + dy * dy, b = 2 * (dx * ox + dy * oy), c = ox * ox + oy * oy - r * r;
  float disc = b * b - 4 * a * c;
  if (disc < 0) return false;
  float ds = sqrtf(disc);
  float qq = (b < 0) ? (b - ds) * (-0.5f) : (b + ds) * (-0.5f);
  float t0p = qq / a, t1p = c / qq;
  float t0 = gminf(t0p, t1p), t1 = gmaxf(t0p, t1p);
  if (t1 < 0 || t0 > d) return false;
  float dist = (t0 < 0) ? t1 : t0;
  if (dist < 0 || dist > d) return false;
  V3 pos = vscaleadd(ray.o, ray.d, dist);
  if (pos.z > h1 && pos.z < h2) { t = dist; n = v3(pos.x / r, pos.y / r, 0); return true; }
  if (dz > 0) {
    if (oz < h1) { n = v3(0, 0, -1); return disc_test(v3(0, 0, h1), n, r * r, ray, d, t); }
    return false;
  }
  if (oz > h2) { n = v3(0, 0, 1); return disc_test(v3(0, 0, h2), n, r * r, ray, d, t); }
  return false;
}
// Cone.hs:155-200 / 206-245: z-axis cone (r, clip1, clip2, height)
GD bool cone_test(const F4& q, const Ray& ray, float d, float& t, V3& n) {
  float r = q.x, clip1 = q.y, clip2 = q.z, height = q.w;
  float ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
  float kp = r / height, k = kp * kp;
  float a = dx * dx + dy * dy - k * dz * dz;
  float b = 2 * (dx * ox + dy * oy - k * dz * (oz - height));
  float c = ox * ox + oy * oy - k * (oz - height) * (oz - height);
  float disc = b * b - 4 * a * c;
  if (disc < 0) return false;
  float ds = sqrtf(disc);
  float qq = (b < 0) ? (b - ds) * (-0.5f) : (b + ds) * (-0.5f);
  float t0p = qq / a, t1p = c / qq;
  float t0 = gminf(t0p, t1p), t1 = gmaxf(t0p, t1p);
  if (t1 < 0 || t0 > d) return false;
  float dist = (t0 < 0) ? t1 : t0;
  if (dist < 0 || dist > d) return false;
  V3 pos = vscaleadd(ray.o, ray.d, dist);
  if (pos.z > clip1 && pos.z < clip2) {
    float invhyp = 1.0f / sqrtf(height * height + r * r);
    float up = r * invhyp, out = height * invhyp;
    float r_ = sqrtf(pos.x * pos.x + pos.y * pos.y);
    float corr = out / r_;
    t = dist; n = v3(pos.x * corr, pos.y * corr, up);
    return true;
  }
  if (dz > 0) {
    if (oz < clip1) { n = v3(0, 0, -1); return disc_test(v3(0, 0, clip1), n, r * r, ray, d, t); }
    return false;
  }
  if (oz > clip2) {
    float r2 = r * (1 - ((clip2 - clip1) / height));
    n = v3(0, 0, 1);
    return disc_test(v3(0, 0, clip2), n, r2 * r2, ray, d, t);
  }
  return false;
}

// One simple primitive (record kinds R_SPHERE..R_CONE): closest-hit test; the normal is produced only
// when WANT_N (the flat tier defers normals to finalize and lets the compiler drop this code).
template <bool WANT_N>
GD bool prim_test(const DScene& S, uint32_t kind, uint32_t a, const Ray& r, float d, float& t, V3& n) {
  switch (kind) {
    case R_SPHERE: {
      F4 s = ld4(S.spheres, a);
      if (!sphere_test(s, r, d, t)) return false;
      if (WANT_N) n = vnorm(vscaleadd(r.o, r.d, t) - v3(s));
      return true;
    }
    case R_TRI: {
      F4 q0 = ld4(S.tris, 3 * a), q1 = ld4(S.tris, 3 * a + 1), q2 = ld4(S.tris, 3 * a + 2);
      float b1, b2;
      if (!tri_test(q0, q1, q2, r, d, t, b1, b2)) return false;
      if (WANT_N) n = v3(q0.w, q1.w, q2.w);  // vnorm(e1 x e2), Triangle.hs:73, precomputed on the host
      return true;
    }
    case R_TRIN: {  // a = base of a 6-word block in trinorms: (p1,.) (e1,.) (e2,.) n1 n2 n3
      F4 q0 = ld4(S.trinorms, a), q1 = ld4(S.trinorms, a + 1), q2 = ld4(S.trinorms, a + 2);
      float b1, b2;
      if (!tri_test(q0, q1, q2, r, d, t, b1, b2)) return false;
      if (WANT_N) {  // Triangle.hs:137-141
        V3 n1 = v3(ld4(S.trinorms, a + 3)), n2 = v3(ld4(S.trinorms, a + 4)), n3 = v3(ld4(S.trinorms, a + 5));
        V3 a1 = n1 * (1 - (b1 + b2)), a2 = n2 * b1, a3 = n3 * b2;
        n = vnorm(v3(a1.x + a2.x + a3.x, a1.y + a2.y + a3.y, a1.z + a2.z + a3.z));
      }
      return true;
    }
    case R_BOX: return box_test(ld4(S.boxes, 2 * a), ld4(S.boxes, 2 * a + 1), r, d, t, n);
    case R_PLANE: {
      F4 pl = ld4(S.planes, a);
      if (!plane_test(pl, r, d, t)) return false;
      if (WANT_N) n = v3(pl);
      return true;
    }
    case R_DISC: {
      F4 d0 = ld4(S.discs, 2 * a), d1 = ld4(S.discs, 2 * a + 1);
      if (!disc_test(v3(d0), v3(d1), d0.w, r, d, t)) return false;
      if (WANT_N) n = v3(d1);
      return true;
    }
    case R_CYL: return cyl_test(ld4(S.quadrics, a), r, d, t, n);
    case R_CONE: return cone_test(ld4(S.quadrics, a), r, d, t, n);
    default: return false;
  }
}
// shadow of one simple primitive.  Sphere / Triangle / Box / Disc / Cone override `shadow`; Plane and
// Cylinder fall back on rayint (Solid.hs:218-221, Q15).
GD bool prim_shadow(const DScene& S, uint32_t kind, uint32_t a, const Ray& r, float d) {
  float t; V3 n;
  switch (kind) {
    case R_SPHERE: return sphere_shadow(ld4(S.spheres, a), r, d);
    case R_BOX: return box_shadow(ld4(S.boxes, 2 * a), ld4(S.boxes, 2 * a + 1), r, d);
    default: return prim_test<false>(S, kind, a, r, d, t, n);
  }
}
// inside, Solid.hs:166: Sphere.hs:73-76, Box.hs:64-68, Plane.hs:34-38, Cone.hs:141-143, 248-251; others False
GD bool prim_inside(const DScene& S, uint32_t kind, uint32_t a, V3 p) {
  switch (kind) {
    case R_SPHERE: { F4 s = ld4(S.spheres, a); V3 off = v3(s) - p; return vdot(off, off) < s.w * s.w; }
    case R_BOX: { F4 lo = ld4(S.boxes, 2 * a), hi = ld4(S.boxes, 2 * a + 1); return p.x > lo.x && p.x < hi.x && p.y > lo.y && p.y < hi.y && p.z > lo.z && p.z < hi.z; }
    case R_PLANE: { F4 pl = ld4(S.planes, a); V3 n = v3(pl); V3 onplane = n * pl.w; return vdot(onplane - p, n) > 0; }
    case R_CYL: { F4 q = ld4(S.quadrics, a); return p.z > q.y && p.z < q.z && p.x * p.x + p.y * p.y < q.x * q.x; }
    case R_CONE: { F4 q = ld4(S.quadrics, a); float rr = q.x * (1 - ((p.z - q.y) / q.w)); return p.z > q.y && p.z < q.z && p.x * p.x + p.y * p.y < rr * rr; }
    default: return false;
  }
}

// wave-wide vote: how many lanes of this wave hold `pred` (host build: one "lane", so all or nothing)
#if defined(__HIPCC__)
GD int wave_count(bool pred) { return __popcll(__ballot(pred)); }
#endif

// ------------------------------------------------------------------ traversal stacks
// Flat tier: one LDS column per lane -- entry e of lane l lives at base[e * 64 + l], so a wave's push or
// pop touches 64 consecutive dwords: conflict-free for ds_read/write_b32 whatever the lanes' depths.
// The LDS part holds `cap` entries (kept small: LDS per wave is what bounds occupancy); deeper pushes, which are
// rare, spill to a per-lane column in global memory (`ovf`, same [entry][lane] layout), so any tree depth up to
// kFlatStack is traversed correctly.
#if defined(__HIPCC__)
struct LaneStack {
  // Everything here is WAVE-UNIFORM (scalar registers); a lane's column -- base + lane -- is formed where it is used.  (Until round 3
  // the struct held five per-lane pointers: ten vector registers alive through the whole kernel for the sake of the rare C++ steps,
  // in a kernel whose register budget the hand-written walk already strains -- DESIGN.md 4.1c.)
  uint32_t* lds;   // the wave's LDS rows: [reference row][near row][far row], cap * 64 words each (no reference row when !has_ref_row)
  uint32_t* ovfb;  // the wave's overflow block in global memory -- ovf_cap entries of [3][64] words, then the dump block [3][64] -- or null
  int cap;         // entries held in LDS
  int ovf_cap;     // entries available in the overflow block
  static constexpr int STRIDE = 64;
  GD int total_cap() const { return cap + ovf_cap; }
  // Packet entries (bih_tri_wave): the node reference and the mask of lanes that want the entry are wave-uniform.
  //   bih_tri_packet (C++) keeps them per lane in the reference row: every lane stores the reference, with bit 31 set
  //   when the lane is in the entry's mask; a pop reads the row back and votes.  No value ever sits in "lane k of a
  //   register": a vector register whose inactive lanes matter is not something the compiler knows about -- a spill or a
  //   copy it places under a partial EXEC mask (a divergent triangle test, say) silently drops those lanes.
  //   bih_walk_asm (hand-written) does hold entry k in lane k of three registers, but only inside its one asm block;
  //   when it hands a step back to C++ it writes them to the dump block (one word per lane and register, global memory) and
  //   reads them back on re-entry, so no C++ variable ever carries them.  push_dump / pop_dump are the C++ steps' view.
  bool has_ref_row;  // false in kernels with two LDS rows per entry (lane_stack<true>): only bih_walk_asm runs there
  unsigned long long* dbg = nullptr;  // DCounters::dbg in the render kernels (read by the GLOME_PKW_STAMPS measurement build only)
  // the lane's number, computed afresh at every use (volatile: neither hoisted nor kept in a register across the walk)
  GD static uint32_t lane() { uint32_t l; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l)); return l; }
  GD uint32_t* node_col() const { return lds + lane(); }
  GD float* near_col() const { return (float*)(lds + (has_ref_row ? cap * STRIDE : 0)) + lane(); }
  GD float* far_col() const { return (float*)(lds + (has_ref_row ? 2 : 1) * cap * STRIDE) + lane(); }
  GD uint32_t* ovf_col() const { return ovfb + lane(); }
  GD uint32_t* dump_base() const { return ovfb ? ovfb + (size_t)ovf_cap * 3 * STRIDE : nullptr; }  // uniform; lane l's three words: [l], [64 + l], [128 + l]
  GD uint32_t* dump_col() const { return dump_base() + lane(); }
  GD uint32_t lds_near_row() const { return (uint32_t)(uintptr_t)near_col(); }  // this lane's LDS byte address in the near row (bih_walk_asm)
  GD void push2(int sp, float a, float b) {
    if (__builtin_expect(sp < cap, 1)) { near_col()[sp * STRIDE] = a; far_col()[sp * STRIDE] = b; }
    else { uint32_t* o = ovf_col() + (size_t)(sp - cap) * 3 * STRIDE; __builtin_nontemporal_store(as_u(a), o + STRIDE); __builtin_nontemporal_store(as_u(b), o + 2 * STRIDE); }
  }
  GD void pop2(int sp, float& a, float& b) const {
    if (__builtin_expect(sp < cap, 1)) { a = near_col()[sp * STRIDE]; b = far_col()[sp * STRIDE]; }
    else {
      const uint32_t* o = ovf_col() + (size_t)(sp - cap) * 3 * STRIDE;
      a = as_f(__builtin_nontemporal_load(o + STRIDE)); b = as_f(__builtin_nontemporal_load(o + 2 * STRIDE));
      asm volatile("" : "+v"(a), "+v"(b));
    }
  }
  GD void push_wave(int sp, uint32_t ref, LaneMask m, float a, float b) { push(sp, ref | (lane_of(m) ? 0x80000000u : 0u), a, b); }
  GD void pop_wave(int sp, uint32_t& ref, LaneMask& m, float& a, float& b) const {
    uint32_t w;
    pop(sp, w, a, b);
    ref = uni(w & 0x7fffffffu);
    m = wave_ballot((w >> 31) != 0);
  }
  // entry `sp` of bih_walk_asm's stack, as it lies in the dump block: lane sp's three words
  GD void push_dump(int sp, uint32_t ref, LaneMask m, float a, float b) {
    if ((int)lane() == sp) { uint32_t* d = dump_col(); d[0] = ref; d[STRIDE] = (uint32_t)m; d[2 * STRIDE] = (uint32_t)(m >> 32); }
    push2(sp, a, b);
  }
  GD void pop_dump(int sp, uint32_t& ref, LaneMask& m, float& a, float& b) const {
    pop2(sp, a, b);
    const uint32_t* d = dump_col();
    const uint32_t w0 = d[0], w1 = d[STRIDE], w2 = d[2 * STRIDE];
    ref = (uint32_t)__builtin_amdgcn_readlane((int)w0, sp);
    m = (LaneMask)(uint32_t)__builtin_amdgcn_readlane((int)w1, sp) | ((LaneMask)(uint32_t)__builtin_amdgcn_readlane((int)w2, sp) << 32);
  }
  // The overflow column is spill traffic (non-temporal).  The empty asm pins the overflow loads inside their branch:
  // without it the compiler sinks both branches' loads into one access through a generic (flat) pointer, and every
  // pop becomes three flat loads that wait on the LDS and the vector-memory counters.
  GD void push(int sp, uint32_t n, float a, float b) {
    if (__builtin_expect(sp < cap, 1)) { node_col()[sp * STRIDE] = n; near_col()[sp * STRIDE] = a; far_col()[sp * STRIDE] = b; }
    else {
      uint32_t* o = ovf_col() + (size_t)(sp - cap) * 3 * STRIDE;
      __builtin_nontemporal_store(n, o); __builtin_nontemporal_store(as_u(a), o + STRIDE); __builtin_nontemporal_store(as_u(b), o + 2 * STRIDE);
    }
  }
  GD void pop(int sp, uint32_t& n, float& a, float& b) const {
    if (__builtin_expect(sp < cap, 1)) { n = node_col()[sp * STRIDE]; a = near_col()[sp * STRIDE]; b = far_col()[sp * STRIDE]; }
    else {
      const uint32_t* o = ovf_col() + (size_t)(sp - cap) * 3 * STRIDE;
      n = __builtin_nontemporal_load(o); a = as_f(__builtin_nontemporal_load(o + STRIDE)); b = as_f(__builtin_nontemporal_load(o + 2 * STRIDE));
      asm volatile("" : "+v"(n), "+v"(a), "+v"(b));
    }
  }
};
// A LaneStack with this lane's columns formed ONCE: what the Mesh walks, which push and pop at every step, work on.  LaneStack itself
// forms a column at each use, which is right for the kernel whose steps are the hand-written walk's and costs these 2.5 % (the
// 1M-triangle Mesh 0.898 -> 0.873 ms).  The columns live in vector registers only while such a walk runs.  (The BIH walks
// keep the plain LaneStack: in the per-lane ones the hoist moved the flagship kernel's register allocation -- 8 -> 13 spilled vector
// registers -- for walks that kernel never runs; in the C++ packet walk it measured neutral on the sphere scenes and +0.7 % on the
// interpreter's service: profiles/r04_probes/stack_cols_ab.txt.)
struct LaneStackCols {
  uint32_t* node; float* nearc; float* farc; uint32_t* ovf; int cap, ovf_cap;
  static constexpr int STRIDE = LaneStack::STRIDE;
  GD explicit LaneStackCols(const LaneStack& s) : cap(s.cap), ovf_cap(s.ovf_cap) {
    const uint32_t l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    node = s.lds + l; nearc = (float*)(s.lds + (s.has_ref_row ? s.cap * STRIDE : 0)) + l; farc = (float*)(s.lds + (s.has_ref_row ? 2 : 1) * s.cap * STRIDE) + l;
    ovf = s.ovfb + l;
  }
  GD int total_cap() const { return cap + ovf_cap; }
  GD void push(int sp, uint32_t n, float a, float b) {
    if (__builtin_expect(sp < cap, 1)) { node[sp * STRIDE] = n; nearc[sp * STRIDE] = a; farc[sp * STRIDE] = b; }
    else {
      uint32_t* o = ovf + (size_t)(sp - cap) * 3 * STRIDE;
      __builtin_nontemporal_store(n, o); __builtin_nontemporal_store(as_u(a), o + STRIDE); __builtin_nontemporal_store(as_u(b), o + 2 * STRIDE);
    }
  }
  GD void pop(int sp, uint32_t& n, float& a, float& b) const {
    if (__builtin_expect(sp < cap, 1)) { n = node[sp * STRIDE]; a = nearc[sp * STRIDE]; b = farc[sp * STRIDE]; }
    else {
      const uint32_t* o = ovf + (size_t)(sp - cap) * 3 * STRIDE;
      n = __builtin_nontemporal_load(o); a = as_f(__builtin_nontemporal_load(o + STRIDE)); b = as_f(__builtin_nontemporal_load(o + 2 * STRIDE));
      asm volatile("" : "+v"(n), "+v"(a), "+v"(b));  // (as in LaneStack::pop: keeps the overflow loads inside their branch)
    }
  }
  GD void push_wave(int sp, uint32_t ref, LaneMask m, float a, float b) { push(sp, ref | (lane_of(m) ? 0x80000000u : 0u), a, b); }
  GD void pop_wave(int sp, uint32_t& ref, LaneMask& m, float& a, float& b) const {
    uint32_t w;
    pop(sp, w, a, b);
    ref = uni(w & 0x7fffffffu);
    m = wave_ballot((w >> 31) != 0);
  }
};
#if !defined(GLOME_NO_STACK_COLS)  // (A/B build: the walks on the plain LaneStack, as in round 4's first measurement set)
GD LaneStackCols stack_cols(LaneStack& s) { return LaneStackCols(s); }
#endif
#endif
template <class STK> GD STK& stack_cols(STK& s) { return s; }  // every other stack already is its own columns
// Generic tier: a fixed private array (scratch).
struct PrivStack {
  uint32_t node[kGenericStack]; float nearv[kGenericStack]; float farv[kGenericStack];
  GD int total_cap() const { return kGenericStack; }
  GD void push(int sp, uint32_t n, float a, float b) { node[sp] = n; nearv[sp] = a; farv[sp] = b; }
  GD void pop(int sp, uint32_t& n, float& a, float& b) const { n = node[sp]; a = nearv[sp]; b = farv[sp]; }
};


// ------------------------------------------------------------------ BIH traversal (Bih.hs:332-368, 510-544; Q10)
// Interval traversal, near child first, far child pushed.
//   MODE 0 (faithful closest): every node the reference visits is visited; `far` is never shrunk.
//   MODE 1 (closest, ordered early-out): `far` is clamped to the best t so far.  The same nearest hit comes
//          out (ties -> later item, like `nearest`, Solid.hs:37-44), in fewer steps.
//   MODE 2 (any hit, shadow_bih): stops at the first occluder.
// leaf(first_rec, first_prim, count, tmax) tests a leaf's items; it returns true to stop the traversal.
// best_t() is the running best distance (MODE 1).
// Child references: bit 29 = leaf, then bits 28..26 = item count (7 = "read it from the leaf node at index
// bits 25..0"), bits 25..0 = first record of the leaf.  A leaf with up to 6 items is therefore described entirely by
// the reference held in its parent: reaching it costs no memory fetch, and an empty leaf (a quarter of the leaves the
// reference builder makes) is never visited at all.  Branch references are plain node indices.
constexpr uint32_t BREF_LEAF = 1u << 29, BREF_MASK = (1u << 30) - 1u, BREF_FIRST = (1u << 26) - 1u, BREF_CONT = 1u << 30;
// CLAMP (MODE 1): whether a leaf's items may be tested with tmax = min(far, best so far).  That is exact for items whose
// rayint reports the same hit for every tmax beyond it (spheres, triangles, boxes, planes, discs).  The reference's cylinder
// and cone are not like that: with the origin inside the infinite quadric, `dist = t1` (the far root) is compared with tmax
// BEFORE the end discs are looked at (Cone.hs:122-139, 176-191), so a tighter tmax turns a cap hit into a miss.  With CLAMP
// false the running best only decides which nodes are still worth entering; the items see the node's own `far`, as in the
// reference (`rayint s r far`, Bih.hs:339).
template <int MODE, bool COUNT, bool CLAMP = true, class STK, class LEAF, class BEST>
GD void bih_traverse(const DScene& S, uint32_t hdr, const Ray& r, float d, STK& stk, int stack_cap, Cnt& cnt, LEAF&& leaf, BEST&& best_t) {
  F4 h0 = ld4(S.bihhdr, 3 * hdr), h1 = ld4(S.bihhdr, 3 * hdr + 1);
  const uint32_t delta = as_u(ld4(S.bihhdr, 3 * hdr + 2).x);  // first_prim - first_rec: constant per homogeneous BIH
  float nearv, farv;
  bbclip_ub(r, v3(h0), v3(h1), nearv, farv);
  farv = gminf(d, farv);  // `traverse root near (fmin d far)`, Bih.hs:368
  V3 rcp = v3(dir_rcp(r.d.x), dir_rcp(r.d.y), dir_rcp(r.d.z));
  uint32_t ref = as_u(h0.w);
  // A tree that is one leaf: the reference tests it whatever the root interval (`traverse (BihLeaf s) near far = rayint s r
  // far`, Bih.hs:339) -- and something CAN be hit when the ray misses the box: Refract's transmitted direction is not unit
  // length (Shader.hs:141), and rayint_sphere's formula (Sphere.hs:20-41) reports hits for such rays where the line misses
  // the sphere.  The reference shows them; so do we.
  const bool root_leaf = (ref & BREF_LEAF) != 0;
  int sp = 0;
  for (;;) {
    bool popit = true;
    float geo_far = farv;  // the node's interval as the planes cut it (what its items are tested with when !CLAMP)
    if (MODE == 1) farv = gminf(farv, best_t());
    if (ref & BREF_LEAF) {
      // BihLeaf: `rayint s r far` -- the reference tests a leaf it has reached without looking at near > far
      // (Bih.hs:339); below a branch the interval was non-empty when the leaf was chosen, so with early-out an empty one
      // means `far` has shrunk to the best hit since: nothing nearer can be in it, and it is skipped.
      uint32_t count = (ref >> 26) & 7u, first = ref & BREF_FIRST;
      if (count == 7u) { F4 n = ld4(S.bihnodes, first); count = as_u(n.z); first = as_u(n.w); }
      if (count != 0 && (MODE == 0 || root_leaf || !(nearv > farv))) {
        if (leaf(first, first + delta, count, (MODE == 1 && !CLAMP) ? geo_far : farv)) return;
      }
    } else {
      if (COUNT) cnt.bih++;  // rayint_debug_bih counts every branch entered, before the near > far test (Bih.hs:389-410)
      if (!(nearv > farv)) {
        F4 n = ld4(S.bihnodes, ref);
        uint32_t w0 = as_u(n.z), w1 = as_u(n.w);
        uint32_t axis = w0 & 3u;
        float dirr = vcomp(rcp, axis), o = vcomp(r.o, axis);
        float dl = (n.x - o) * dirr, dr = (n.y - o) * dirr;
        uint32_t left = w0 >> 2, right = w1;
        uint32_t c1, c2; float c1far, c2near; bool go1, go2;
        if (dirr > 0) { c1 = left; go1 = nearv < dl; c1far = gminf(dl, farv); c2 = right; go2 = dr < farv; c2near = gmaxf(dr, nearv); }
        else { c1 = right; go1 = nearv < dr; c1far = gminf(dr, farv); c2 = left; go2 = dl < farv; c2near = gmaxf(dl, nearv); }
        // an empty leaf holds nothing to test: never visit it (no effect on results or on the branch-visit counts)
        go1 = go1 && c1 != BREF_LEAF;
        go2 = go2 && c2 != BREF_LEAF;
        if (MODE == 1 && !CLAMP) {  // the children keep the interval the planes give them; `best` only decided go1 / go2
          c1far = gminf(dirr > 0 ? dl : dr, geo_far);
          farv = geo_far;
        }
        if (go1) {
          if (go2 && sp < stack_cap) { stk.push(sp, c2, c2near, farv); sp++; }
          ref = c1; farv = c1far; popit = false;
        } else if (go2) {
          ref = c2; nearv = c2near; popit = false;
        }
      }
    }
    if (popit) {
      if (sp == 0) return;
      sp--;
      stk.pop(sp, ref, nearv, farv);
    }
  }
}

// ------------------------------------------------------------------ BIH of triangles (the `Bih Triangle` SPECIALIZE, Bih.hs:370-374)
// Same traversal as bih_traverse, shaped for a 64-lane wave: every loop iteration a lane takes ONE step -- a branch
// step or one triangle of the leaf it stands on (the leaf reference itself is the cursor: first record + remaining
// count) -- so lanes walking the tree are not held up while a neighbour works through a leaf.  An interval is known to be
// non-empty when a child is chosen; only an entry popped after `far` has shrunk (MODE 1) can be stale, and that is checked
// at the pop.  best_t / best_rec: running nearest hit (best_t = kNoBest when none); MODE 2 returns true at the first occluder.
constexpr float kNoBest = 3.0e38f;
constexpr uint32_t kNoRec = 0xffffffffu;  // == CAND_NONE below
// LEAFK: what the leaves hold -- 0 triangles (48-byte records), 1 spheres (16-byte records).
GD bool leaf_item_test(const DScene& S, int leafk, bool shadow, uint32_t prim, const Ray& r, float tmax, float& t) {
  if (leafk == 0) {
    float b1, b2;
    return tri_test(ld4(S.tris, 3 * prim), ld4(S.tris, 3 * prim + 1), ld4(S.tris, 3 * prim + 2), r, tmax, t, b1, b2);
  }
  return shadow ? sphere_shadow(ld4(S.spheres, prim), r, tmax) : sphere_test(ld4(S.spheres, prim), r, tmax, t);
}
template <int MODE, bool COUNT, int LEAFK = 0, class STK>
GD bool bih_tri(const DScene& S, uint32_t hdr, const Ray& r, float d, STK& stk, Cnt& cnt, float& best_t, uint32_t& best_rec) {
  F4 h0 = ld4(S.bihhdr, 3 * hdr), h1 = ld4(S.bihhdr, 3 * hdr + 1);
  const uint32_t delta = as_u(ld4(S.bihhdr, 3 * hdr + 2).x);
  float nearv, farv;
  bbclip_ub(r, v3(h0), v3(h1), nearv, farv);
  farv = gminf(d, farv);  // `traverse root near (fmin d far)`, Bih.hs:368
  const V3 rcp = v3(dir_rcp(r.d.x), dir_rcp(r.d.y), dir_rcp(r.d.z));
  uint32_t ref = as_u(h0.w);
  const int cap = stk.total_cap();
  int sp = 0;
  bool occ = false;
  if (nearv > farv) {  // the root interval is empty
    if (!(ref & BREF_LEAF)) { if (COUNT) cnt.bih++; return false; }  // a branch is entered, counted and left (Bih.hs:343)
    // a root leaf is tested regardless, with tmax = far (Bih.hs:339).  With a sound bound nothing can be hit then -- but
    // Refract's transmitted direction is not unit length (Shader.hs:141), and rayint_sphere's formula (Sphere.hs:20-41)
    // reports hits for such rays where the line misses the sphere and its box: the reference shows them, so do we.
  }
  for (;;) {
    bool popit;
    if (!(ref & BREF_LEAF)) {
      if (COUNT) cnt.bih++;
      F4 n = ld4(S.bihnodes, ref);
      uint32_t w0 = as_u(n.z), right = as_u(n.w);
      uint32_t axis = w0 & 3u, left = w0 >> 2;
      float dirr = vcomp(rcp, axis), o = vcomp(r.o, axis);
      float dl = (n.x - o) * dirr, dr = (n.y - o) * dirr;
      bool fwd = dirr > 0;
      uint32_t c1 = fwd ? left : right, c2 = fwd ? right : left;
      float t1 = fwd ? dl : dr, t2 = fwd ? dr : dl;        // near child ends at t1, far child starts at t2
      // an empty leaf holds nothing to test: it is never visited (no effect on results or on the branch counts)
      bool go1 = (nearv < t1) && (c1 != BREF_LEAF), go2 = (t2 < farv) && (c2 != BREF_LEAF);
      float c1far = gminf(t1, farv), c2near = gmaxf(t2, nearv);
      if (go1 && go2 && sp < cap) { stk.push(sp, c2, c2near, farv); sp++; }
      ref = go1 ? c1 : c2;
      nearv = go1 ? nearv : c2near;
      farv = go1 ? c1far : farv;
      popit = !(go1 || go2);
    } else {
      uint32_t count = (ref >> 26) & 7u, first = ref & BREF_FIRST;
      if (__builtin_expect(count == 7u, 0)) {
        // a leaf of more than 6 triangles: its extent is in the leaf node.  It is worked through six at a time; the
        // rest waits on the stack as a continuation (bit 30; the `near` slot carries the offset reached).
        F4 ln = ld4(S.bihnodes, first);
        uint32_t k0 = (ref & BREF_CONT) ? as_u(nearv) : 0u, left_over = as_u(ln.z) - k0;
        if (left_over > 6u && sp < cap) { stk.push(sp, ref | BREF_CONT, as_f(k0 + 6u), farv); sp++; }
        ref = BREF_LEAF | ((left_over < 6u ? left_over : 6u) << 26) | (as_u(ln.w) + k0);
        popit = false;
      } else if (count == 0u) {
        popit = true;  // an empty root leaf, or a stale entry retired at the pop below
      } else {
        uint32_t a = first + delta;
        float t;
        if (COUNT) cnt.prim++;
        // tmax = far (Bih.hs:339; shadow: `fmin d far`, Bih.hs:515 -- far <= d already); MODE 1: far <= best_t
        bool hit = leaf_item_test(S, LEAFK, MODE == 2, a, r, farv, t);
        ref += 1u - (1u << 26);  // the next triangle of this leaf
        popit = count == 1u;
        if (MODE == 2) { if (hit) { occ = true; sp = 0; popit = true; } }  // first occluder: leave through the one exit
        else if (hit && !(best_t < t)) { best_t = t; best_rec = first; if (MODE == 1) farv = gminf(farv, t); }
      }
    }
    if (popit) {
      if (sp == 0) break;
      sp--;
      stk.pop(sp, ref, nearv, farv);
      if (MODE == 1 && !(ref & BREF_CONT)) {
        farv = gminf(farv, best_t);
        if (nearv > farv) ref = BREF_LEAF;  // nothing nearer can be in there: retire it (one idle step, as before)
      }
    }
  }
  return occ;
}

// ------------------------------------------------------------------ BIH of triangles, one wave = one packet
// The 64 rays of a work item (an 8x8 pixel block, or the shadow rays leaving it) visit almost the same nodes: on the
// 100k-triangle scene the union over a wave is 1.2x one ray's own list.  So the wave walks the tree ONCE: the node
// reference is wave-uniform (scalar loads, scalar branches, no per-lane stack pointer), each lane carries only its own
// [near, far] interval and is simply inactive (near = +inf, far = -inf) where its ray would not go; a child is entered
// when any lane's interval reaches it.  Per lane the sequence of nodes entered and triangles tested is exactly the one
// bih_tri walks -- children are taken near-first by the direction signs, which must therefore agree across the wave;
// when they do not (a block straddling an axis plane through the eye), or the root is a leaf, the lanes fall back to
// bih_tri.  Results, tie order and work counters are identical to the per-lane traversal.
// `valid`: the lane holds a ray.  All lanes of the wave must make this call together.
// The packet loop proper.  Every branch in it is wave-uniform (the per-lane decisions are selects), and it takes and
// returns everything by value, so it can be compiled as a function of its own with plain scalar control flow.
#ifndef GLOME_LDS_STACK
#define GLOME_LDS_STACK 12
#endif
constexpr int kAsmLdsCap = GLOME_LDS_STACK;  // entries of the LDS part of the flat tier's stack: the hand-written walk is instantiated for it
struct PacketResult { float best_t; uint32_t best_rec; uint32_t occ_lo, occ_hi, n_bih, n_prim; };
template <int MODE, bool COUNT, int LEAFK, class STK>
GD PacketResult bih_tri_packet(const F4* nodes, const F4* tris, uint32_t ref, uint32_t delta, uint32_t fwdbits, uint32_t am_lo, uint32_t am_hi,
                                float nearv, float farv, V3 ro, V3 rd, V3 rcp, float best_t, STK stk) {
  ref = uni(ref); delta = uni(delta); fwdbits = uni(fwdbits);
  LaneMask am = uni((LaneMask)am_lo | ((LaneMask)am_hi << 32));
  Ray r; r.o = ro; r.d = rd;
  PacketResult R; R.best_t = best_t; R.best_rec = kNoRec; R.n_bih = 0; R.n_prim = 0;
  int sp = 0;
  LaneMask occm = 0;  // MODE 2: lanes that found an occluder
  for (;;) {
    // ---- branch steps: walk down while the reference is a branch
    while (!(ref & BREF_LEAF)) {
      ref = uni(ref); am = uni(am); sp = (int)uni((uint32_t)sp);  // wave-uniform by construction: keep them in SGPRs
      F4 n = ld4u(nodes, ref);
      const uint32_t w0 = uni(as_u(n.z)), right = uni(as_u(n.w));
      const uint32_t axis = w0 & 3u, left = w0 >> 2;
      float dl, dr;  // distances to the two planes along the ray; the axis is wave-uniform: a scalar branch, no selects
      if (axis == 0) { dl = (n.x - r.o.x) * rcp.x; dr = (n.y - r.o.x) * rcp.x; }
      else if (axis == 1) { dl = (n.x - r.o.y) * rcp.y; dr = (n.y - r.o.y) * rcp.y; }
      else { dl = (n.x - r.o.z) * rcp.z; dr = (n.y - r.o.z) * rcp.z; }
      const bool fwd = (fwdbits >> axis) & 1u;
      const uint32_t c1 = fwd ? left : right, c2 = fwd ? right : left;
      const float t1 = fwd ? dl : dr, t2 = fwd ? dr : dl;  // near child ends at t1, far child starts at t2
      if (COUNT) R.n_bih += lane_of(am) ? 1u : 0u;
      // (an empty leaf child has its plane at -+inf, flatten.hpp: both tests fail for it, it is never entered)
      const LaneMask m1 = wave_ballot(nearv < t1) & am;
      const LaneMask m2 = wave_ballot(t2 < farv) & am;
      const float f1 = min_nn(t1, farv), n2 = max_nn(t2, nearv);
      if ((m1 != 0) & (m2 != 0)) { stk.push_wave(sp, c2, m2, n2, farv); sp++; }  // depth <= capacity (validated at commit)
      const bool g1 = m1 != 0;
      ref = g1 ? c1 : c2;
      am = g1 ? m1 : m2;
      farv = g1 ? f1 : farv;
      nearv = g1 ? nearv : n2;
      if (am == 0) break;
    }
    // ---- a leaf (entered with am != 0), or nothing left below the last branch (am == 0)
    if (am != 0) {
      ref = uni(ref);
      uint32_t count = (ref >> 26) & 7u, first = ref & BREF_FIRST;
      if (count == 7u) { F4 ln = ld4u(nodes, first); count = uni(as_u(ln.z)); first = uni(as_u(ln.w)); }
      for (uint32_t k = 0; k < count; k++) {
        float t;
        bool hit;
        if (COUNT) R.n_prim += lane_of(am) ? 1u : 0u;
        // tmax = far (Bih.hs:339; shadow: `fmin d far`, Bih.hs:515 -- far <= d already); MODE 1: far <= best_t
        if (LEAFK == 0) {
          F4 p0, p1, p2;
          float b1, b2;
          ld_tri_u(tris, first + delta + k, p0, p1, p2);
          hit = tri_test(p0, p1, p2, r, farv, t, b1, b2);
        } else {
          F4 sp4 = ld4u(tris, first + delta + k);  // `tris` is the sphere pool here
          hit = MODE == 2 ? sphere_shadow(sp4, r, farv) : sphere_test(sp4, r, farv, t);
        }
        hit = hit && lane_of(am);
        if (MODE == 2) { const LaneMask hm = wave_ballot(hit); occm |= hm; am &= ~hm; }
        else {
          const bool acc = hit && !(R.best_t < t);  // selects, not a branch
          R.best_t = acc ? t : R.best_t;
          R.best_rec = acc ? first + k : R.best_rec;
          if (MODE == 1) farv = acc ? gminf(farv, t) : farv;
        }
      }
    }
    // ---- pop until an entry some lane still wants (MODE 1: `far` may have shrunk since the push; MODE 2: lanes retire)
    am = 0;
    while (sp > 0 && am == 0) {
      sp--;
      stk.pop_wave(sp, ref, am, nearv, farv);
      if (MODE == 1) { farv = gminf(farv, R.best_t); am &= wave_ballot(!(nearv > farv)); }
      if (MODE == 2) am &= ~occm;
    }
    if (am == 0) break;
  }
  R.occ_lo = (uint32_t)occm; R.occ_hi = (uint32_t)(occm >> 32);
  return R;
}

#if defined(__HIPCC__)
}  // namespace glome
#include "bih_packet_asm.hpp"
namespace glome {
// The production packet walk: bih_walk_asm (hand-written, one instance per octant) with the C++ steps of bih_tri_packet for
// what it declines -- a push or pop beyond the LDS part of the stack.  Same visits, same results, same tie order as
// bih_tri_packet<MODE, false, 0>.  `nodes` is the walk's own pool (DScene::pknodes) and `ref`, like every stack entry of this
// walk, in its form: a branch = byte offset | axis, a leaf = byte offset of its first pair record | 3 (flatten.hpp emit_bih).
template <int MODE>
GD PacketResult bih_tri_packet_hw(const F4* nodes, uint32_t nbytes, const float* pairs, uint32_t ref, uint32_t delta, uint32_t fwdbits, LaneMask am, float nearv, float farv,
                                  V3 ro, V3 rd, V3 rcp, float best_t, LaneStack& stk) {
  constexpr int CAP = kAsmLdsCap;
  ref = uni(ref); delta = uni(delta); fwdbits = uni(fwdbits); am = uni(am);
  Ray r; r.o = ro; r.d = rd;
  PacketResult R; R.best_t = best_t; R.best_rec = kNoRec; R.n_bih = 0; R.n_prim = 0;
  int sp = 0, phase = 0;
  LaneMask occm = 0;
  const uint32_t lds_row = stk.lds_near_row();
  for (;;) {
    int st;
#define GLOME_WALK(XF, YF, ZF) st = bih_walk_asm<MODE, XF, YF, ZF, CAP>(nodes, nbytes, pairs, phase, ref, am, sp, nearv, farv, R.best_t, R.best_rec, occm, r.o, rcp, r.d, lds_row, stk.dump_base())
    switch (fwdbits) {  // wave-uniform: one scalar jump per walk
      case 7: GLOME_WALK(true, true, true); break;
      case 6: GLOME_WALK(false, true, true); break;
      case 5: GLOME_WALK(true, false, true); break;
      case 4: GLOME_WALK(false, false, true); break;
      case 3: GLOME_WALK(true, true, false); break;
      case 2: GLOME_WALK(false, true, false); break;
      case 1: GLOME_WALK(true, false, false); break;
      default: GLOME_WALK(false, false, false); break;
    }
#undef GLOME_WALK
    st = (int)uni((uint32_t)st);
    if (st == PKW_DONE) break;
#ifdef GLOME_PROBE
    R.n_bih++;  // (probe builds: steps handed back to C++, render_loop sums them into DCounters::dbg[14])
#endif
    ref = uni(ref); am = uni(am); sp = (int)uni((uint32_t)sp);
    if (st == PKW_PUSH_OVERFLOW) {  // one branch step of bih_tri_packet; its push goes to the overflow columns
      F4 n = ld4u(nodes, ref >> 4);
      const uint32_t left = uni(as_u(n.z)), right = uni(as_u(n.w));
      const uint32_t axis = ref & 3u;
      float dl, dr;
      if (axis == 0) { dl = (n.x - r.o.x) * rcp.x; dr = (n.y - r.o.x) * rcp.x; }
      else if (axis == 1) { dl = (n.x - r.o.y) * rcp.y; dr = (n.y - r.o.y) * rcp.y; }
      else { dl = (n.x - r.o.z) * rcp.z; dr = (n.y - r.o.z) * rcp.z; }
      const bool fwd = (fwdbits >> axis) & 1u;
      const uint32_t c1 = fwd ? left : right, c2 = fwd ? right : left;
      const float t1 = fwd ? dl : dr, t2 = fwd ? dr : dl;
      const LaneMask m1 = wave_ballot(nearv < t1) & am, m2 = wave_ballot(t2 < farv) & am;
      const float f1 = min_nn(t1, farv), n2 = max_nn(t2, nearv);
      if ((m1 != 0) & (m2 != 0)) { stk.push_dump(sp, c2, m2, n2, farv); sp++; }  // depth <= capacity (validated at commit)
      const bool g1 = m1 != 0;
      ref = g1 ? c1 : c2;
      am = g1 ? m1 : m2;
      farv = g1 ? f1 : farv;  // (the walk only ever reads the lanes of its current mask)
      nearv = g1 ? nearv : n2;
      phase = am != 0 ? 0 : 1;
    } else {  // PKW_POP_OVERFLOW: the top entry sits in the overflow columns
      sp--;
      stk.pop_dump(sp, ref, am, nearv, farv);
      if (MODE == 1) { farv = gminf(farv, R.best_t); am &= wave_ballot(!(nearv > farv)); }
      if (MODE == 2) am &= ~occm;
      phase = am != 0 ? 0 : 1;
    }
  }
  R.occ_lo = (uint32_t)occm; R.occ_hi = (uint32_t)(occm >> 32);
  return R;
}
#endif

template <int MODE, bool COUNT, int LEAFK = 0, class STK>
GD bool bih_tri_wave(const DScene& S, uint32_t hdr, const Ray& r, float d, bool valid, STK& stk, Cnt& cnt, float& best_t, uint32_t& best_rec) {
  hdr = uni(hdr);
  F4 h0 = ld4u(S.bihhdr, 3 * hdr), h1 = ld4u(S.bihhdr, 3 * hdr + 1);
  const F4 h2 = ld4u(S.bihhdr, 3 * hdr + 2);
  const uint32_t delta = uni(as_u(h2.x));
  const uint32_t pkroot = uni(as_u(h2.y)), has_pk = uni(as_u(h2.z));  // the root as the hand-written walk refers to it (flatten.hpp emit_bih)
  (void)pkroot; (void)has_pk;  // (the host build has no hand-written walk)
  const uint32_t ref = uni(as_u(h0.w));
  const V3 rcp = v3(dir_rcp(r.d.x), dir_rcp(r.d.y), dir_rcp(r.d.z));
  float nearv, farv;
  bbclip_ub(r, v3(h0), v3(h1), nearv, farv);
  farv = gminf(d, farv);  // `traverse root near (fmin d far)`, Bih.hs:368
  if (ref & BREF_LEAF) {
    // A one-leaf tree: `rayint [s] r far` over its items (Bih.hs:339), per lane.  Nothing is pushed -- the kernels that
    // keep only two stack rows per entry (lane_stack<TWO_ROWS>) rely on that; bih_tri's continuation entries for leaves
    // of more than six items would land on a row they do not have.  The count is wave-uniform.
    uint32_t count = (ref >> 26) & 7u, first = ref & BREF_FIRST;
    if (count == 7u) { F4 ln = ld4u(S.bihnodes, first); count = uni(as_u(ln.z)); first = uni(as_u(ln.w)); }
    if (!valid) return false;  // (a root leaf is tested regardless of its interval, Bih.hs:339: see bih_traverse)
    bool occ1 = false;
    for (uint32_t k = 0; k < count; k++) {
      float t;
      if (COUNT) cnt.prim++;
      bool hit = leaf_item_test(S, LEAFK, MODE == 2, first + delta + k, r, farv, t);
      if (MODE == 2) { if (hit) { occ1 = true; break; } }
      else if (hit && !(best_t < t)) { best_t = t; best_rec = first + k; if (MODE == 1) farv = gminf(farv, t); }
    }
    return occ1;
  }
  if (COUNT) { if (valid && nearv > farv) cnt.bih++; }  // a root branch entered with an empty interval is counted and left (Bih.hs:343)
  // Children are taken near-first by the signs of the ray direction, so one packet needs one sign pattern.  Almost
  // every wave has a single pattern; a block straddling an axis plane through the eye has two or four, and is walked
  // once per pattern with the other lanes switched off (a lane takes part in exactly one walk).
  const uint32_t oct = (rcp.x > 0 ? 1u : 0u) | (rcp.y > 0 ? 2u : 0u) | (rcp.z > 0 ? 4u : 0u);
  // `am`: the lanes whose ray has a non-empty interval in the current node.  Their (near, far) are live; the other
  // lanes' are don't-cares, so no sentinel values are needed and plain min / max serve (a NaN plane distance only
  // arises on a lane that fails the activity test of that child).
  LaneMask todo = wave_ballot(valid && !(nearv > farv));
  bool occ = false;
  while (todo != 0) {
    const uint32_t fwdbits = uni(first_lane_value(todo, oct));  // per axis: do the rays of this walk run towards +axis
    const LaneMask am = todo & wave_ballot(oct == fwdbits);
    todo &= ~am;
    PacketResult R;
    bool walked = false;
#if defined(__HIPCC__)
    if constexpr (MODE != 0 && !COUNT && LEAFK == 0 && std::is_same<STK, LaneStack>::value) {
      // The hand-written walk keeps entry k of its stack in lane k of three registers and hands steps to C++ through the stack's dump
      // block: it needs EVERY lane of the wave at this call and a stack that has a dump block.  Both hold by construction in the flat
      // tier's kernels (one wave per workgroup, wave-uniform callers); the generic tier's packet service calls from inside a loop whose
      // lanes diverge and carries a stack without one (generic_packet_stack) -- checked here, at run time, rather than trusted to the
      // caller's template arguments: anything else takes the C++ packet walk below.
      if (stk.cap == kAsmLdsCap && has_pk && stk.dump_base() != nullptr && __builtin_amdgcn_read_exec() == ~0ull) { R = bih_tri_packet_hw<MODE>(S.pknodes, S.pknodes_bytes, S.tripairs, pkroot, delta, fwdbits, am, nearv, farv, r.o, r.d, rcp, best_t, stk); walked = true; }
    }
#endif
    if (!walked) {
      if (stk.has_ref_row) R = bih_tri_packet<MODE, COUNT, LEAFK, STK>(S.bihnodes, LEAFK == 0 ? S.tris : S.spheres, ref, delta, fwdbits, (uint32_t)am, (uint32_t)(am >> 32),
                                                                     nearv, farv, r.o, r.d, rcp, best_t, stk);
      else { R.best_t = best_t; R.best_rec = kNoRec; R.occ_lo = R.occ_hi = R.n_bih = R.n_prim = 0; }  // (a kernel with two stack rows is only launched for what bih_walk_asm walks)
    }
#ifdef GLOME_PROBE
    if (!COUNT) cnt.bih += R.n_bih;  // (every lane: the wave's count of C++ steps)
#endif
    if (lane_of(am)) {
      if (COUNT) { cnt.bih += R.n_bih; cnt.prim += R.n_prim; }
      if (MODE != 2 && R.best_rec != kNoRec) { best_t = R.best_t; best_rec = R.best_rec; }
      if (MODE == 2) occ = lane_of((LaneMask)R.occ_lo | ((LaneMask)R.occ_hi << 32));
    }
  }
  return occ;
}

// ------------------------------------------------------------------ Mesh 2-box BVH (Mesh.hs:136-198; Q12)
// Child refs are one word: bit 31 = leaf, bits 30..27 = triangle count (15 = read it from mtrimeta[first].z),
// bits 26..0 = first triangle (leaf order); a branch ref is a node index.  Ordered traversal: the child
// entered first is followed, the other is pushed and re-tested on pop against the best hit of this mesh,
// which is <= the depth of the first child's result that the reference clips with (Mesh.hs:178, 190).
// Leaves use the box interval `far` as tmax, not `depth` -- as written in the reference (Mesh.hs:163, 198).
constexpr uint32_t MREF_LEAF = 0x80000000u;
template <bool COUNT, class STK>
GD void mesh_closest(const DScene& S, uint32_t mh, const Ray& ray, float depth, STK& stk_, int stack_cap, Cnt& cnt, float& best_t, uint32_t& best_tri) {
  auto&& stk = stack_cols(stk_);
  F4 h0 = ld4(S.meshhdr, 2 * mh), h1 = ld4(S.meshhdr, 2 * mh + 1);
  V3 rcp = v3(dir_rcp(ray.d.x), dir_rcp(ray.d.y), dir_rcp(ray.d.z));
  float nearv, farv;
  bbclip_ub_rcp(ray.o, rcp, v3(h0), v3(h1), nearv, farv);
  best_t = kInf; best_tri = 0xffffffffu;  // ridepth RayMiss = infinity
  if (nearv > farv || nearv > depth || farv < 0) return;
  uint32_t ref = as_u(h0.w);
  int sp = 0;
  for (;;) {
    bool popit = true;
    if (ref & MREF_LEAF) {  // Leaf: foldl' nearest over the triangles, each tested with tmax = far
      uint32_t first = ref & 0x07ffffffu, count = (ref >> 27) & 15u;
      if (count == 15u) count = ldu4(S.mtrimeta, first).z;
      float tmax = pminf(farv, best_t);
      for (uint32_t k = 0; k < count; k++) {
        uint32_t ti = first + k;
        F4 q0 = ld4(S.mtris, 3 * ti), q1 = ld4(S.mtris, 3 * ti + 1), q2 = ld4(S.mtris, 3 * ti + 2);
        float t, b1, b2;
        if (COUNT) cnt.prim++;
        if (tri_test(q0, q1, q2, ray, tmax, t, b1, b2)) { best_t = t; best_tri = ti; tmax = t; }
      }
    } else {
      if (COUNT) cnt.mesh++;
      F4 a0 = ld4(S.meshnodes, 4 * ref), a1 = ld4(S.meshnodes, 4 * ref + 1), b0 = ld4(S.meshnodes, 4 * ref + 2), b1 = ld4(S.meshnodes, 4 * ref + 3);
      float lnp, lfp, rnp, rfp;
      bbclip_ub_rcp(ray.o, rcp, v3(a0), v3(a1), lnp, lfp);
      bbclip_ub_rcp(ray.o, rcp, v3(b0), v3(b1), rnp, rfp);
      float lnear = pmaxf(nearv, lnp), lfar = pminf(farv, lfp), rnear = pmaxf(nearv, rnp), rfar = pminf(farv, rfp);
      bool lfirst = lnear < rnear;
      uint32_t fref = lfirst ? as_u(a0.w) : as_u(b0.w), sref = lfirst ? as_u(b0.w) : as_u(a0.w);
      float fnear = lfirst ? lnear : rnear, ffar = lfirst ? lfar : rfar;
      float snear = lfirst ? rnear : lnear, sfar = lfirst ? rfar : lfar;
      bool gof = !(fnear > ffar || fnear > depth || ffar < 0);
      float sfar2 = pminf(sfar, best_t);
      bool gos = !(snear > sfar2 || snear > depth || sfar2 < 0);
      if (gof) {
        if (gos && sp < stack_cap) { stk.push(sp, sref, snear, sfar); sp++; }
        ref = fref; nearv = fnear; farv = ffar; popit = false;
      } else if (gos) {
        ref = sref; nearv = snear; farv = sfar; popit = false;
      }
    }
    while (popit) {
      if (sp == 0) return;
      sp--;
      stk.pop(sp, ref, nearv, farv);
      float f2 = pminf(farv, best_t);  // rfar' = min rfar (ridepth lresult)
      if (!(nearv > f2 || nearv > depth || f2 < 0)) popit = false;
    }
  }
}


#if defined(__HIPCC__)
// rayint_mesh for the 64 rays of a wave at once (round 3).  A node is fetched once, by scalar loads; each lane clips its ray
// against the two child boxes as mesh_closest does.  Unlike the BIH, a mesh ray takes the child whose box it enters first
// (`lnear < rnear`, Mesh.hs:178), so the lanes of one packet can want opposite orders -- and the order decides which of two
// equal hits on a shared edge is kept (nearest: ties -> later).  The packet therefore visits a node's children in up to three
// passes, so that every lane sees its own order: the left subtree for the lanes that take it first, the right subtree for the
// lanes that take it first or second, the left one again for the lanes that take it second.  With coherent rays one of the
// groups is empty and it is two passes, like the per-lane walk.  A stack entry is (reference, per-lane interval); a lane that
// is not in the entry stores an empty interval, and the pop's re-test against the best hit so far (Mesh.hs:190) is what tells
// the lanes in from the lanes out.  Hits, ties and ray counts are the per-lane walk's; node visit counts are not kept.
template <class STK>
GD void mesh_closest_wave(const DScene& S, uint32_t mh, const Ray& ray, float depth, bool valid, STK& stk_, float& best_t, uint32_t& best_tri, unsigned int* err) {
  auto&& stk = stack_cols(stk_);
  mh = uni(mh);
  const F4 h0 = ld4u(S.meshhdr, 2 * mh), h1 = ld4u(S.meshhdr, 2 * mh + 1);
  const V3 rcp = v3(dir_rcp(ray.d.x), dir_rcp(ray.d.y), dir_rcp(ray.d.z));
  float nearv, farv;
  bbclip_ub_rcp(ray.o, rcp, v3(h0), v3(h1), nearv, farv);
  best_t = kInf; best_tri = 0xffffffffu;  // ridepth RayMiss = infinity
  LaneMask am = wave_ballot(valid && !(nearv > farv || nearv > depth || farv < 0));
  if (am == 0) return;
  uint32_t ref = uni(as_u(h0.w));
  const int cap = stk.total_cap();
  const float kOut = 3.0e38f;  // a lane outside an entry: near = +kOut, far = -kOut fails every re-test
  int sp = 0;
  for (;;) {
    const bool in = lane_of(am);
    if (ref & MREF_LEAF) {  // Leaf: foldl' nearest over the triangles, each tested with tmax = far
      uint32_t first = ref & 0x07ffffffu, count = (ref >> 27) & 15u;
      if (count == 15u) count = uni(ldu4(S.mtrimeta, first).z);
      float tmax = pminf(farv, best_t);
      for (uint32_t k = 0; k < count; k++) {
        const uint32_t ti = first + k;
        F4 q0, q1, q2;
        ld_tri_u(S.mtris, ti, q0, q1, q2);
        float t, b1, b2;
        if (tri_test(q0, q1, q2, ray, tmax, t, b1, b2) && in) { best_t = t; best_tri = ti; tmax = t; }
      }
      am = 0;
    } else {
      const F4 a0 = ld4u(S.meshnodes, 4 * ref), a1 = ld4u(S.meshnodes, 4 * ref + 1), b0 = ld4u(S.meshnodes, 4 * ref + 2), b1 = ld4u(S.meshnodes, 4 * ref + 3);
      float lnp, lfp, rnp, rfp;
      bbclip_ub_rcp(ray.o, rcp, v3(a0), v3(a1), lnp, lfp);
      bbclip_ub_rcp(ray.o, rcp, v3(b0), v3(b1), rnp, rfp);
      const float lnear = pmaxf(nearv, lnp), lfar = pminf(farv, lfp), rnear = pmaxf(nearv, rnp), rfar = pminf(farv, rfp);
      const bool lfirst = lnear < rnear;
      const float fnear = lfirst ? lnear : rnear, ffar = lfirst ? lfar : rfar;
      const float snear = lfirst ? rnear : lnear, sfar = lfirst ? rfar : lfar;
      const bool gof = in && !(fnear > ffar || fnear > depth || ffar < 0);
      const float sfar2 = pminf(sfar, best_t);
      const bool gos = in && !(snear > sfar2 || snear > depth || sfar2 < 0);
      // what this lane does: its first child now (or, if that is a miss, its second at once), its second later
      const bool now_l = gof ? lfirst : (gos && !lfirst), now_r = gof ? !lfirst : (gos && lfirst);
      const bool later_l = gof && gos && !lfirst, later_r = gof && gos && lfirst;
      const LaneMask mNL = wave_ballot(now_l), mNR = wave_ballot(now_r), mLL = wave_ballot(later_l), mLR = wave_ballot(later_r);
      const uint32_t lref = uni(as_u(a0.w)), rref = uni(as_u(b0.w));
      // passes in order: left (mNL), right (mNR | mLR), left again (mLL) -- or, with nobody going left now, right (mNR) then left (mLL)
      const bool in_r = now_r || later_r;
      const LaneMask mR = mNR | mLR;
      if (mNL != 0) {
        if (sp + 2 > cap) { if (err) *err = 1; return; }  // (commit sizes the stack for two entries per level: never, short of a limit)
        if (mLL != 0) { stk.push(sp, lref, later_l ? lnear : kOut, later_l ? lfar : -kOut); sp++; }
        if (mR != 0) { stk.push(sp, rref, in_r ? rnear : kOut, in_r ? rfar : -kOut); sp++; }
        ref = lref; am = mNL; nearv = lnear; farv = lfar;
      } else if (mR != 0) {
        if (sp + 1 > cap) { if (err) *err = 1; return; }
        if (mLL != 0) { stk.push(sp, lref, later_l ? lnear : kOut, later_l ? lfar : -kOut); sp++; }
        ref = rref; am = mR; nearv = rnear; farv = rfar;
      } else am = 0;
    }
    while (am == 0) {  // pop until an entry some lane still wants: rfar' = min rfar (ridepth lresult), Mesh.hs:190
      if (sp == 0) return;
      sp--;
      uint32_t w;
      stk.pop(sp, w, nearv, farv);
      ref = uni(w);
      const float f2 = pminf(farv, best_t);
      am = wave_ballot(!(nearv > f2 || nearv > depth || f2 < 0));
    }
  }
}
#endif

struct HitCore {  // a Rayint (Solid.hs:20-28) as the shader reads it
  bool hit;
  float t;
  V3 p, n;
  TexStack tex;
  uint32_t uid;
};
struct HitG : HitCore {  // ... plus riray: the ray as the primitive that was hit saw it (local inside Instances, advanced inside CSG).
  V3 lo, ld;            // Only a Warp material reads it, so only the generic tier (where Warp scenes render) fills it in.
};
GD HitG hit_miss() { HitG h; h.hit = false; h.t = kInf; h.p = v3(0, 0, 0); h.n = v3(0, 0, 0); h.tex = 0; h.uid = 0xffffffffu; h.lo = v3(0, 0, 0); h.ld = v3(0, 0, 0); return h; }


// ------------------------------------------------------------------ CSG over primitives, without recursion (flat tier)
// Difference / Intersection whose operands are primitives (under any Tex wrappers), and an Instance of a primitive or of
// such a node: TestScene.hs's carved spheres and boxes, `sphereint`, plane-cut polyhedra, and every `cylinder` / `cone`
// (Cone.hs:40-67 wraps the canonical quadric in an Instance).  The reference's class methods recurse through
// dictionaries; with primitive operands each call bottoms out at once, so the methods are plain loops here, inlined into
// the flat tier's kernels.  (Composites below composites go to the generic tier's interpreter, rt_generic.hpp.)
// One explicit frame of rayint_intersection's list recursion (isect_rayint).  `from` holds the list position and, in
// its top two bits, the frame's state; `aux` is the state's one live distance (state 1: the inside hit's depth, state 2:
// the advance added back on return).
struct IFrame { uint32_t from; float ox, oy, oz, d, aux, acc; };  // acc: advances folded into this frame when the frames had run out (added back when it completes)
GD HitG nearest_hit(const HitG& a, const HitG& b) {  // nearest, Solid.hs:37-44: ties -> b
  if (!b.hit) return a;
  if (!a.hit) return b;
  return (a.t < b.t) ? a : b;
}
GD TexStack own_stack_rayint(uint32_t own, int B) { return tex_from16(own, B); }  // innermost Tex first (Tex.hs:66)
GD TexStack own_stack_meta(uint32_t own, int B) {                      // get_metainfo: outermost Tex first (Tex.hs:73-74)
  if (own >> 16) return (TexStack)(own >> 16) | ((TexStack)(own & 0xffffu) << B);
  return (TexStack)own;
}

struct Xf6 { F4 f0, f1, f2, i0, i1, i2; };
GD Xf6 load_xf(const DScene& S, uint32_t x) {
  Xf6 m;
  m.f0 = ld4(S.xfms, 6 * x); m.f1 = ld4(S.xfms, 6 * x + 1); m.f2 = ld4(S.xfms, 6 * x + 2);
  m.i0 = ld4(S.xfms, 6 * x + 3); m.i1 = ld4(S.xfms, 6 * x + 4); m.i2 = ld4(S.xfms, 6 * x + 5);
  return m;
}
GD V3 mat_point(const F4& r0, const F4& r1, const F4& r2, V3 v) {  // xfm_point / invxfm_point, Vec.hs:502-519
  return v3(r0.x * v.x + r0.y * v.y + r0.z * v.z + r0.w, r1.x * v.x + r1.y * v.y + r1.z * v.z + r1.w, r2.x * v.x + r2.y * v.y + r2.z * v.z + r2.w);
}
GD V3 mat_vec(const F4& r0, const F4& r1, const F4& r2, V3 v) {  // xfm_vec / invxfm_vec, Vec.hs:522-539
  return v3(r0.x * v.x + r0.y * v.y + r0.z * v.z, r1.x * v.x + r1.y * v.y + r1.z * v.z, r2.x * v.x + r2.y * v.y + r2.z * v.z);
}
GD V3 mat_tvec(const F4& r0, const F4& r1, const F4& r2, V3 v) {  // invxfm_norm: transpose, Vec.hs:543-550
  return v3(r0.x * v.x + r1.x * v.y + r2.x * v.z, r0.y * v.x + r1.y * v.y + r2.y * v.z, r0.z * v.x + r1.z * v.y + r2.z * v.z);
}


template <bool C> GD HitG leaf_rayint(const DScene& S, Cnt& cnt, U4 rec, const Ray& r, float d, TexStack tex) {  // a primitive under Tex records
  for (;;) {  // Tex s tex: rayint s r d (tex:texs) tags, Tex.hs:66
    if (rec.x & RF_NOVIS) return hit_miss();
    if ((rec.x & RF_KINDMASK) != R_TEX) break;
    tex = tex_push(tex, rec.z, (int)S.tex_bits);
    rec = ldu4(S.recs, rec.y);
  }
  HitG h = hit_miss();
  if (C) cnt.prim++;
  float t; V3 n;
  if (!prim_test<true>(S, rec.x & RF_KINDMASK, rec.y, r, d, t, n)) return h;
  h.hit = true; h.t = t; h.n = n; h.p = vscaleadd(r.o, r.d, t); h.lo = r.o; h.ld = r.d;  // (lo / ld: riray, read by Warp materials -- generic tier only)
  h.tex = tex_cat(own_stack_rayint(rec.z, (int)S.tex_bits), tex, (int)S.tex_bits); h.uid = rec.w;
  return h;
}
GD U4 skip_tex(const DScene& S, U4 rec) {  // strip Tex records (and stop at the first non-Tex record)
  while ((rec.x & RF_KINDMASK) == R_TEX) rec = ldu4(S.recs, rec.y);
  return rec;
}
GD bool leaf_inside(const DScene& S, U4 rec, V3 p) { rec = skip_tex(S, rec); return prim_inside(S, rec.x & RF_KINDMASK, rec.y, p); }
GD TexStack leaf_meta(const DScene& S, U4 rec) {  // get_metainfo of a primitive under Tex records: tex : texs, outermost first (Tex.hs:73-74)
  TexStack pre = 0;
  while ((rec.x & RF_KINDMASK) == R_TEX) { pre = tex_cat(pre, (TexStack)(rec.z + 1), (int)S.tex_bits); rec = ldu4(S.recs, rec.y); }
  return tex_cat(pre, own_stack_meta(rec.z, (int)S.tex_bits), (int)S.tex_bits);
}
// rayint_difference, Csg.hs:33-54 (Q13); the self-recursion through rayint_advance (Solid.hs:85-91) is a loop
template <bool C> GD HitG csg_diff(const DScene& S, Cnt& cnt, unsigned int& err, U4 rec, const Ray& r0, float d0, TexStack tex) {
  const U4 ra = ldu4(S.recs, rec.y), rb = ldu4(S.recs, rec.z);
  float adds[kCsgFlatAdvance];
  int na = 0;
  Ray r = r0;
  float d = d0;
  HitG res = hit_miss();
  for (;;) {
    const bool inb = leaf_inside(S, rb, r.o);
    HitG ha = hit_miss();
    if (!inb) { ha = leaf_rayint<C>(S, cnt, ra, r, d, tex); if (!ha.hit) break; }
    HitG hb = leaf_rayint<C>(S, cnt, rb, r, d, tex);
    if (inb) {
      if (!hb.hit) break;
      if (leaf_inside(S, ra, hb.p) && !leaf_inside(S, rb, vscaleadd(hb.p, r.d, kDel))) {
        hb.n = vneg(hb.n);
        if (!(rec.x & RF_RETEX)) hb.tex = leaf_meta(S, ra);  // `difference` = Difference a b True: textures of A at the carved point; difference_retexture keeps B's (Csg.hs:42-43)
        res = hb;
        break;
      }
    } else {
      if (!hb.hit) { res = ha; break; }
      if (ha.t < hb.t) { res = ha; break; }
    }
    const float a = hb.t + kDel;
    if (na < kCsgFlatAdvance) adds[na] = a;
    else {  // beyond the list: the advance joins the last slot (rt_types.h kCsgFlatAdvance); a runaway ray is reported, not followed for ever
      adds[kCsgFlatAdvance - 1] = adds[kCsgFlatAdvance - 1] + a;
      if (na >= kCsgRunaway) { err = 1; break; }
    }
    na++;
    r.o = vscaleadd(r.o, r.d, a);  // ray_move
    d = d - a;
  }
  if (res.hit) for (int k = (na < kCsgFlatAdvance ? na : kCsgFlatAdvance) - 1; k >= 0; k--) res.t = res.t + adds[k];  // RayHit (depth+a) ..., innermost first
  return res;
}
// rayint_intersection, Csg.hs:68-90 (Q14): the recursion on the list tail and on the advanced ray as explicit frames
template <bool C> GD HitG csg_isect(const DScene& S, Cnt& cnt, unsigned int& err, U4 rec, const Ray& r0, float d0, TexStack tex) {
  constexpr uint32_t kSt1 = 1u << 30, kSt2 = 2u << 30, kFrom = (1u << 30) - 1u;
  const uint32_t n = rec.z;
  IFrame fr[kIsectFrames];
  int sp = 0;
  auto push = [&](uint32_t from, V3 o, float d) { IFrame& c = fr[sp]; c.from = from; c.ox = o.x; c.oy = o.y; c.oz = o.z; c.d = d; c.aux = 0; c.acc = 0; };
  // an advance when no frame is left (the reference recurses without a bound, Solid.hs:85-91): the frame goes on in place from the
  // advanced origin and remembers the distance -- the same sum as the nested `RayHit (depth + a)`s in another order (an ulp of the depth)
  auto advance_in_place = [&](IFrame& f, uint32_t from, V3 o, float a) { const V3 q = vscaleadd(o, r0.d, a); f.from = from; f.ox = q.x; f.oy = q.y; f.oz = q.z; f.d = f.d - a; f.aux = 0; f.acc = f.acc + a; };
  push(0, r0.o, d0);
  HitG ret = hit_miss();
  int nadv = 0;
  bool returning = false;  // true: frame fr[sp] has completed with `ret`
  for (;;) {
    if (!returning) {
      IFrame& f = fr[sp];
      const uint32_t from = f.from & kFrom;
      Ray r; r.o = v3(f.ox, f.oy, f.oz); r.d = r0.d;
      if (from >= n || f.d < 0) { ret = hit_miss(); returning = true; continue; }  // null slds || d < 0
      const U4 s = ldu4(S.recs, rec.y + from);
      HitG hs = leaf_rayint<C>(S, cnt, s, r, f.d, tex);
      if (from + 1 == n) { ret = hs; returning = true; continue; }  // [] -> rayint s r d t tags
      if (leaf_inside(S, s, r.o)) {
        if (!hs.hit) { f.from = from + 1; continue; }  // RayMiss -> rayint (Intersection ss) r d: a tail call
        if (sp + 1 >= kIsectFrames) { err = 1; return hit_miss(); }
        f.aux = hs.t; f.from = from | kSt1;  // rest = rayint (Intersection ss) r sd
        sp++; push(from + 1, r.o, hs.t);
        continue;
      }
      if (!hs.hit) { ret = hit_miss(); returning = true; continue; }
      bool all = true;  // inside (Intersection ss) sp: foldl' (&&) True
      for (uint32_t k = from + 1; k < n; k++) all = all && leaf_inside(S, ldu4(S.recs, rec.y + k), hs.p);
      if (all) { ret = hs; returning = true; continue; }  // RayHit sd sp sn r vzero st stags
      const float a = hs.t + kDel;  // rayint_advance (Intersection slds) r d t tags sd
      if (++nadv > kCsgRunaway) { err = 1; return hit_miss(); }
      // (the list positions still ahead may each need a frame of their own: an advance takes one only while those stay free)
      if (sp + 1 + (int)(n - from) >= kIsectFrames) { advance_in_place(f, from, r.o, a); continue; }
      f.from = from | kSt2; f.aux = a;
      sp++; push(from, vscaleadd(r.o, r.d, a), f.d - a);
      continue;
    }
    if (ret.hit && fr[sp].acc != 0) ret.t = ret.t + fr[sp].acc;  // (advances this frame took in place)
    if (sp == 0) return ret;
    sp--;
    IFrame& p = fr[sp];
    if ((p.from & ~kFrom) == kSt1) {
      if (ret.hit) continue;  // hit -> hit
      const float a = p.aux + kDel;
      const uint32_t pf = p.from & kFrom;
      if (++nadv > kCsgRunaway) { err = 1; return hit_miss(); }
      if (sp + 1 + (int)(n - pf) >= kIsectFrames) { advance_in_place(p, pf, v3(p.ox, p.oy, p.oz), a); returning = false; continue; }
      p.from = pf | kSt2; p.aux = a;
      const V3 po = v3(p.ox, p.oy, p.oz);
      const float pd = p.d;
      sp++; push(pf, vscaleadd(po, r0.d, a), pd - a);
      returning = false;
      continue;
    }
    if (ret.hit) ret.t = ret.t + p.aux;  // state 2: RayHit (depth+a) ...
  }
}
// rayint of a flat-tier CSG item: primitive | Difference | Intersection, optionally inside one Instance (Solid.hs:388-403, Q8)
template <bool C> GD HitG csg_item_rayint(const DScene& S, Cnt& cnt, unsigned int& err, U4 rec, const Ray& ray, float d, TexStack tex) {
  for (;;) {  // Tex records over a composite
    if (rec.x & RF_NOVIS) return hit_miss();
    if ((rec.x & RF_KINDMASK) != R_TEX) break;
    tex = tex_push(tex, rec.z, (int)S.tex_bits);
    rec = ldu4(S.recs, rec.y);
  }
  Ray r = ray;
  float dd = d, invlenscale = 1.0f;
  Xf6 x{};
  const bool inst = (rec.x & RF_KINDMASK) == R_INSTANCE;
  if (inst) {
    x = load_xf(S, rec.z);
    const V3 newdir = mat_vec(x.i0, x.i1, x.i2, ray.d), neworig = mat_point(x.i0, x.i1, x.i2, ray.o);
    const float lenscale = sqrtf(vdot(newdir, newdir));
    invlenscale = 1.0f / lenscale;
    r.o = neworig; r.d = newdir * invlenscale;
    dd = d * lenscale;
    rec = ldu4(S.recs, rec.y);
    for (;;) {  // Tex records between the Instance and a composite child
      if (rec.x & RF_NOVIS) return hit_miss();
      if ((rec.x & RF_KINDMASK) != R_TEX) break;
      tex = tex_push(tex, rec.z, (int)S.tex_bits);
      rec = ldu4(S.recs, rec.y);
    }
  }
  const uint32_t kind = rec.x & RF_KINDMASK;
  HitG h;
  if (kind == R_DIFF) h = csg_diff<C>(S, cnt, err, rec, r, dd, tex);
  else if (kind == R_ISECT) h = csg_isect<C>(S, cnt, err, rec, r, dd, tex);
  else h = leaf_rayint<C>(S, cnt, rec, r, dd, tex);
  if (inst && h.hit) {
    h.t = h.t * invlenscale;
    h.p = mat_point(x.f0, x.f1, x.f2, h.p);
    h.n = vnorm(mat_tvec(x.i0, x.i1, x.i2, h.n));
  }
  return h;
}
// shadow of a flat-tier CSG item: shadow_instance (Solid.hs:464-471), the primitives' own methods, and for Difference /
// Intersection -- which have no shadow method -- the class default `rayint` (Solid.hs:218-221, Q15)
template <bool C> GD bool csg_item_shadow(const DScene& S, Cnt& cnt, unsigned int& err, U4 rec, const Ray& ray, float d) {
  for (;;) {  // shadow (Tex s _) = shadow s; NoShadow -> False (Tex.hs:69, 81)
    if (rec.x & RF_NOSHADOW) return false;
    if ((rec.x & RF_KINDMASK) != R_TEX) break;
    rec = ldu4(S.recs, rec.y);
  }
  Ray r = ray;
  float dd = d;
  if ((rec.x & RF_KINDMASK) == R_INSTANCE) {
    const Xf6 x = load_xf(S, rec.z);
    const V3 newdir = mat_vec(x.i0, x.i1, x.i2, ray.d), neworig = mat_point(x.i0, x.i1, x.i2, ray.o);
    const float lenscale = sqrtf(vdot(newdir, newdir)), invlenscale = 1.0f / lenscale;
    r.o = neworig; r.d = newdir * invlenscale;
    dd = d * lenscale;
    rec = ldu4(S.recs, rec.y);
    for (;;) {
      if (rec.x & RF_NOSHADOW) return false;
      if ((rec.x & RF_KINDMASK) != R_TEX) break;
      rec = ldu4(S.recs, rec.y);
    }
  }
  const uint32_t kind = rec.x & RF_KINDMASK;
  // (the class default sees the node itself, so an OnlyShadow flag on it does not hide it here)
  if (kind == R_DIFF) { U4 v = rec; v.x &= ~RF_NOVIS; return csg_diff<C>(S, cnt, err, v, r, dd, (TexStack)0).hit; }
  if (kind == R_ISECT) { U4 v = rec; v.x &= ~RF_NOVIS; return csg_isect<C>(S, cnt, err, v, r, dd, (TexStack)0).hit; }
  if (C) cnt.prim++;
  return prim_shadow(S, kind, rec.y, r, dd);
}

// ------------------------------------------------------------------ flat tier
// A candidate is (t, id, aux): id = record index of the primitive (or mesh-triangle index when aux has
// CAND_MESH); aux = entry index.  Normals are derived after traversal (finalize_flat).
struct Cand { float t; uint32_t id; uint32_t aux; };
constexpr uint32_t CAND_NONE = 0xffffffffu;
constexpr uint32_t CAND_MESH = 0x80000000u;

// closest hit over the flat root program = the list instance's `foldl' nearest RayMiss` (Solid.hs:327) over
// simple primitives, homogeneous BIHs and meshes.  Every entry is tested with the same d (Q9).
// CLS is the set of entry classes the kernel instance is compiled for (the device analogue of the reference's
// SPECIALIZE pragmas for Bih Triangle / Bih Sphere, Bih.hs:370-374): an all-triangle scene runs a kernel that contains
// only the triangle loops, which keeps it small enough to stay in registers and in the instruction cache.
constexpr int CLS_BIH_TRI = 1, CLS_BIH_SPHERE = 2, CLS_BIH_SIMPLE = 4, CLS_MESH = 8, CLS_PRIMS = 16, CLS_ALL = 31, CLS_CSG = 32, CLS_EVERY = 63;
constexpr uint32_t CAND_CSG = 0x40000000u;  // aux flag: the candidate is a CSG item's hit, complete in the side record
// WAVE: the call is made by all lanes of a wave together (`valid` = this lane holds a ray); triangle BIHs are then
// walked as one packet (bih_tri_wave), everything else per lane as before.
// CSGH: where the full hit of a CSG item goes (its normal, position and textures come out of the evaluation and cannot be
// re-derived from a record like a primitive's); null in kernels without CLS_CSG.
template <bool FAITHFUL, bool COUNT, int CLS, bool WAVE = false, class STK>
GD Cand closest_flat(const DScene& S, const Ray& r, float d, STK& stk, Cnt& cnt, bool valid = true, HitG* csgh = nullptr, unsigned int* err = nullptr) {
  Cand best; best.t = kInf; best.id = CAND_NONE; best.aux = 0;
  for (uint32_t e = 0; e < S.n_entries; e++) {
    U4 ent = ldu4(S.entries, e);
    if (WAVE) { ent.x = uni(ent.x); ent.z = uni(ent.z); }
    if (ent.z & RF_NOVIS) continue;
    U4 rec = ldu4(S.recs, ent.x);
    if (WAVE) { rec.x = uni(rec.x); rec.y = uni(rec.y); }
    uint32_t kind = rec.x & RF_KINDMASK;
    uint32_t cls = 0;
    if (kind == R_BIH) { cls = as_u(ld4(S.bihhdr, 3 * rec.y + 1).w); if (WAVE) cls = uni(cls); }
    const bool tri_bih = kind == R_BIH && (CLS & CLS_BIH_TRI) && (CLS == CLS_BIH_TRI || cls == BC_TRI);
    const bool sph_bih = kind == R_BIH && (CLS & CLS_BIH_SPHERE) && cls == BC_SPHERE;
    const bool mesh_pk = WAVE && !FAITHFUL && !COUNT && (CLS & CLS_MESH) && kind == R_MESH;
    if (WAVE && !valid && !tri_bih && !sph_bih && !mesh_pk) continue;  // only the packet walks need the lanes without a ray
    // tmax for this entry; a hit replaces the running best when !(best.t < t)  (nearest: ties -> later)
    float dd = (FAITHFUL || best.id == CAND_NONE) ? d : gminf(d, best.t);
    if (kind == R_BIH) {
      auto bestt = [&]() { return best.id == CAND_NONE ? kInf * 4.0f : best.t; };
      auto accept = [&](float t, uint32_t id) { if (best.id == CAND_NONE || !(best.t < t)) { best.t = t; best.id = id; best.aux = e; } };
      if (tri_bih) {
        float bt = best.id == CAND_NONE ? kNoBest : best.t;
        uint32_t brec = CAND_NONE;
        if (WAVE) bih_tri_wave<FAITHFUL ? 0 : 1, COUNT>(S, rec.y, r, dd, valid, stk, cnt, bt, brec);
        else bih_tri<FAITHFUL ? 0 : 1, COUNT>(S, rec.y, r, dd, stk, cnt, bt, brec);
        if (brec != CAND_NONE) { best.t = bt; best.id = brec; best.aux = e; }
      } else if (sph_bih) {
        float bt = best.id == CAND_NONE ? kNoBest : best.t;
        uint32_t brec = CAND_NONE;
        if (WAVE) bih_tri_wave<FAITHFUL ? 0 : 1, COUNT, 1>(S, rec.y, r, dd, valid, stk, cnt, bt, brec);
        else bih_tri<FAITHFUL ? 0 : 1, COUNT, 1>(S, rec.y, r, dd, stk, cnt, bt, brec);
        if (brec != CAND_NONE) { best.t = bt; best.id = brec; best.aux = e; }
      } else if ((CLS & CLS_CSG) && cls == BC_CSG) {  // primitives and CSG over primitives: every item evaluated in full
        // (items here may be cylinders / cones, whose rayint depends on tmax beyond the hit: no clamping, see bih_traverse)
        bih_traverse<FAITHFUL ? 0 : 1, COUNT, false>(S, rec.y, r, d, stk, stk.total_cap(), cnt,
          [&](uint32_t frec, uint32_t, uint32_t count, float tmax) {
            for (uint32_t k = 0; k < count; k++) {
              const HitG h = csg_item_rayint<COUNT>(S, cnt, *err, ldu4(S.recs, frec + k), r, tmax, tex_from16(ent.y, (int)S.tex_bits));  // `rayint s r far`
              if (h.hit && (best.id == CAND_NONE || !(best.t < h.t))) { best.t = h.t; best.id = frec + k; best.aux = e | CAND_CSG; *csgh = h; }
            }
            return false;
          }, bestt);
      } else if (CLS & CLS_BIH_SIMPLE) {  // BC_SIMPLE: mixed simple primitives, possibly with NoShadow / OnlyShadow flags
        bih_traverse<FAITHFUL ? 0 : 1, COUNT>(S, rec.y, r, dd, stk, stk.total_cap(), cnt,
          [&](uint32_t frec, uint32_t, uint32_t count, float tmax) {
            for (uint32_t k = 0; k < count; k++) {
              U4 it = ldu4(S.recs, frec + k);
              if (it.x & RF_NOVIS) continue;
              float t; V3 n;
              if (COUNT) cnt.prim++;
              if (prim_test<false>(S, it.x & RF_KINDMASK, it.y, r, tmax, t, n)) { accept(t, frec + k); if (!FAITHFUL) tmax = gminf(tmax, best.t); }
            }
            return false;
          }, bestt);
      }
    } else if ((CLS & CLS_MESH) && kind == R_MESH) {
      float mt; uint32_t mtri;
#if defined(__HIPCC__)
      if constexpr (WAVE && !FAITHFUL && !COUNT) mesh_closest_wave(S, rec.y, r, d, valid, stk, mt, mtri, err);  // the wave's 64 rays at once
      else
#endif
      mesh_closest<COUNT>(S, rec.y, r, d, stk, stk.total_cap(), cnt, mt, mtri);  // depth = the list's d (Q12)
      if (mtri != 0xffffffffu && (best.id == CAND_NONE || !(best.t < mt))) { best.t = mt; best.id = mtri; best.aux = e | CAND_MESH; }
    } else if ((CLS & CLS_CSG) && (kind == R_DIFF || kind == R_ISECT || kind == R_INSTANCE || kind == R_TEX)) {  // a CSG item in the root list
      const HitG h = csg_item_rayint<COUNT>(S, cnt, *err, rec, r, d, tex_from16(ent.y, (int)S.tex_bits));  // (every list item with the same d, Solid.hs:327: a cone's answer depends on it)
      if (h.hit && (best.id == CAND_NONE || !(best.t < h.t))) { best.t = h.t; best.id = ent.x; best.aux = e | CAND_CSG; *csgh = h; }
    } else if ((CLS & CLS_PRIMS) && kind != R_VOID && kind != R_MESH && kind <= R_CONE) {
      float t; V3 n;
      if (COUNT) cnt.prim++;
      if (prim_test<false>(S, kind, rec.y, r, dd, t, n) && (best.id == CAND_NONE || !(best.t < t))) { best.t = t; best.id = ent.x; best.aux = e; }
    }
  }
  return best;
}

// Turn a candidate into a full hit: position (vscaleadd o dir t), normal, texture stack, primitive id.
template <int CLS>
GD HitG finalize_flat(const DScene& S, const Ray& r, const Cand& c, const HitG* csgh = nullptr) {
  HitG h = hit_miss();
  if (c.id == CAND_NONE) return h;
  if ((CLS & CLS_CSG) && (c.aux & CAND_CSG)) return *csgh;
  h.hit = true; h.t = c.t;
  h.p = vscaleadd(r.o, r.d, c.t);
  U4 ent = ldu4(S.entries, c.aux & ~(CAND_MESH | CAND_CSG));
  if ((CLS & CLS_MESH) && (c.aux & CAND_MESH)) {
    U4 rec = ldu4(S.recs, ent.x);
    uint32_t ti = c.id;
    U4 meta = ldu4(S.mtrimeta, ti);
    F4 q0 = ld4(S.mtris, 3 * ti), q1 = ld4(S.mtris, 3 * ti + 1), q2 = ld4(S.mtris, 3 * ti + 2);
    if (meta.x == 0) h.n = v3(q0.w, q1.w, q2.w);
    else {  // rayint_trianglenorm with the mesh's normals (Mesh.hs:158-161)
      float t, b1, b2;
      tri_test(q0, q1, q2, r, kInf * 8.0f, t, b1, b2);
      uint32_t nb = meta.x - 1;
      V3 n1 = v3(ld4(S.trinorms, nb)), n2 = v3(ld4(S.trinorms, nb + 1)), n3 = v3(ld4(S.trinorms, nb + 2));
      V3 a1 = n1 * (1 - (b1 + b2)), a2 = n2 * b1, a3 = n3 * b2;
      h.n = vnorm(v3(a1.x + a2.x + a3.x, a1.y + a2.y + a3.y, a1.z + a2.z + a3.z));
    }
    TexStack own = meta.y ? (TexStack)meta.y : 0;  // (texv ! texi) : texs, Mesh.hs:148-150 (already id+1)
    h.tex = tex_cat(own, tex_from16(ent.y, (int)S.tex_bits), (int)S.tex_bits);
    h.uid = rec.w;
    return h;
  }
  U4 rec = ldu4(S.recs, c.id);
  if (CLS == CLS_BIH_TRI) {  // every primitive is a flat triangle: the normal is in the record
    h.n = v3(ld4(S.tris, 3 * rec.y).w, ld4(S.tris, 3 * rec.y + 1).w, ld4(S.tris, 3 * rec.y + 2).w);
  } else {
    float t;
    prim_test<true>(S, rec.x & RF_KINDMASK, rec.y, r, kInf * 8.0f, t, h.n);
  }
  h.tex = tex_cat(own_stack_rayint(rec.z, (int)S.tex_bits), tex_from16(ent.y, (int)S.tex_bits), (int)S.tex_bits);
  h.uid = rec.w;
  return h;
}

// shadow over the flat root program: `foldl' (||) False (map shadow xs)` (Solid.hs:330); Mesh casts none (Mesh.hs:210)
template <bool COUNT, int CLS, bool WAVE = false, class STK>
GD bool occluded_flat(const DScene& S, const Ray& r, float d, STK& stk, Cnt& cnt, bool valid = true, unsigned int* err = nullptr) {
  bool result = false;  // WAVE: an occluded lane stays in the entry loop without a ray until the wave is through
  for (uint32_t e = 0; e < S.n_entries; e++) {
    U4 ent = ldu4(S.entries, e);
    if (WAVE) { ent.x = uni(ent.x); ent.z = uni(ent.z); }
    if (ent.z & RF_NOSHADOW) continue;
    U4 rec = ldu4(S.recs, ent.x);
    if (WAVE) { rec.x = uni(rec.x); rec.y = uni(rec.y); }
    uint32_t kind = rec.x & RF_KINDMASK;
    uint32_t cls = 0;
    if (kind == R_BIH) { cls = as_u(ld4(S.bihhdr, 3 * rec.y + 1).w); if (WAVE) cls = uni(cls); }
    const bool tri_bih = kind == R_BIH && (CLS & CLS_BIH_TRI) && (CLS == CLS_BIH_TRI || cls == BC_TRI);
    const bool sph_bih = kind == R_BIH && (CLS & CLS_BIH_SPHERE) && cls == BC_SPHERE;
    if (WAVE) { if (!wave_any(valid)) break; if (!valid && !tri_bih && !sph_bih) continue; }
    if (kind == R_BIH) {
      bool occ = false;
      auto nobest = [&]() { return 0.0f; };
      if (tri_bih) {
        float bt = kNoBest; uint32_t brec = CAND_NONE;
        if (WAVE) occ = bih_tri_wave<2, COUNT>(S, rec.y, r, d, valid, stk, cnt, bt, brec);
        else occ = bih_tri<2, COUNT>(S, rec.y, r, d, stk, cnt, bt, brec);
      } else if (sph_bih) {
        float bt = kNoBest; uint32_t brec = CAND_NONE;
        if (WAVE) occ = bih_tri_wave<2, COUNT, 1>(S, rec.y, r, d, valid, stk, cnt, bt, brec);
        else occ = bih_tri<2, COUNT, 1>(S, rec.y, r, d, stk, cnt, bt, brec);
      } else if ((CLS & CLS_CSG) && cls == BC_CSG) {
        bih_traverse<2, COUNT>(S, rec.y, r, d, stk, stk.total_cap(), cnt,
          [&](uint32_t frec, uint32_t, uint32_t count, float tmax) {
            float dd = gminf(d, tmax);
            for (uint32_t k = 0; k < count; k++) if (csg_item_shadow<COUNT>(S, cnt, *err, ldu4(S.recs, frec + k), r, dd)) { occ = true; return true; }
            return false;
          }, nobest);
      } else if (CLS & CLS_BIH_SIMPLE) {
        bih_traverse<2, COUNT>(S, rec.y, r, d, stk, stk.total_cap(), cnt,
          [&](uint32_t frec, uint32_t, uint32_t count, float tmax) {
            float dd = gminf(d, tmax);
            for (uint32_t k = 0; k < count; k++) {
              U4 it = ldu4(S.recs, frec + k);
              if (it.x & RF_NOSHADOW) continue;
              if (COUNT) cnt.prim++;
              if (prim_shadow(S, it.x & RF_KINDMASK, it.y, r, dd)) { occ = true; return true; }
            }
            return false;
          }, nobest);
      }
      if (occ) { if (!WAVE) return true; result = true; valid = false; }
    } else if ((CLS & CLS_CSG) && (kind == R_DIFF || kind == R_ISECT || kind == R_INSTANCE || kind == R_TEX)) {
      if (csg_item_shadow<COUNT>(S, cnt, *err, rec, r, d)) { if (!WAVE) return true; result = true; valid = false; }
    } else if ((CLS & CLS_PRIMS) && kind != R_MESH && kind != R_VOID && kind <= R_CONE) {
      if (COUNT) cnt.prim++;
      if (prim_shadow(S, kind, rec.y, r, d)) { if (!WAVE) return true; result = true; valid = false; }
    }
  }
  return result;
}

// ------------------------------------------------------------------ colour algebra (Clr.hs)
struct CA { float r, g, b, a; };
GD CA ca(float r, float g, float b, float a) { CA c; c.r = r; c.g = g; c.b = b; c.a = a; return c; }
GD CA cafold(CA c1, CA c2) {  // Clr.hs:106-113
  float trans = 1 - c1.a;
  return ca(c1.r + (c2.r * trans * c2.a), c1.g + (c2.g * trans * c2.a), c1.b + (c2.b * trans * c2.a), c1.a + (c2.a * trans));
}
GD CA caweight(CA c1, CA c2, float w) {  // Clr.hs:87-91
  return ca((c1.r * w) + (c2.r * (1 - w)), (c1.g * w) + (c2.g * (1 - w)), (c1.b * w) + (c2.b * (1 - w)), (c1.a * w) + (c2.a * (1 - w)));
}
GD float aclamp(float x) { return x > 1 ? 1.0f : (x < 0 ? 0.0f : x); }  // Clr.hs:75-79

// ------------------------------------------------------------------ solid texture functions (GlomeVec Texture.hs)
// Scalar fields 0..1 over the hit position, used as Blend weights by TestScene.hs's t_mottled / t_stripe closures
// (TestScene.hs:214-234): Perlin noise with the reference's omega / phi / gamma tables (Texture.hs:48-117) and
// stripe = wave . vdot axis (Texture.hs:11-41).
GD float tx_omega(float t_) {  // Texture.hs:48-53
  float t = fabsf(t_), tsqr = t * t, tcube = tsqr * t;
  return (-6.0f) * tcube * tsqr + 15.0f * tcube * t - 10.0f * tcube + 1.0f;
}
GD int tx_phi(int i) {  // Texture.hs:56-57 -- [3,0,2,7,4,1,5,11,8,10,9,6] packed four bits apiece
  return (int)((0x69A8B5147203ull >> (4 * i)) & 15ull);
}
GD V3 tx_grad(int c) {  // Texture.hs:59-64: the 12 edge directions in the comprehension's order (x outermost)
  // x: -1 for c < 4, 0 for 4..7, +1 for 8..11; within a group of four: (y, z) = (-1,0) (0,-1) (0,1) (1,0) when x != 0,
  // (-1,-1) (-1,1) (1,-1) (1,1) when x == 0
  int g = c >> 2, k = c & 3;
  float x = (float)(g - 1);
  float y, z;
  if (g == 1) { y = (k & 2) ? 1.0f : -1.0f; z = (k & 1) ? 1.0f : -1.0f; }
  else { y = k == 0 ? -1.0f : (k == 3 ? 1.0f : 0.0f); z = k == 1 ? -1.0f : (k == 2 ? 1.0f : 0.0f); }
  return v3(x, y, z);
}
GD int tx_iabs(int v) { return v < 0 ? -v : v; }
GD float tx_knot(int i, int j, int k, V3 v) {  // Texture.hs:66-76
  int a = tx_phi(tx_iabs(k) % 12);
  int b = tx_phi(tx_iabs(j + a) % 12);
  int c = tx_phi(tx_iabs(i + b) % 12);
  return tx_omega(v.x) * tx_omega(v.y) * tx_omega(v.z) * vdot(tx_grad(c), v);
}
GD float tx_noise(V3 p) {  // Texture.hs:92-107
  float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
  int i = (int)fx, j = (int)fy, k = (int)fz;
  float u = p.x - fx, v = p.y - fy, w = p.z - fz;
  return tx_knot(i, j, k, v3(u, v, w)) + tx_knot(i + 1, j, k, v3(u - 1, v, w)) + tx_knot(i, j + 1, k, v3(u, v - 1, w)) +
         tx_knot(i, j, k + 1, v3(u, v, w - 1)) + tx_knot(i + 1, j + 1, k, v3(u - 1, v - 1, w)) + tx_knot(i + 1, j, k + 1, v3(u - 1, v, w - 1)) +
         tx_knot(i, j + 1, k + 1, v3(u, v - 1, w - 1)) + tx_knot(i + 1, j + 1, k + 1, v3(u - 1, v - 1, w - 1));
}
// weight of a Blend: fn 0 constant, 1 perlin (vscale pos p0), 2/3/4 square / triangle / sine wave of (vdot pos (p0,p1,p2))
GD float tx_weight(uint32_t fn, float constant, float p0, float p1, float p2, V3 pos) {
  if (fn == 0u) return constant;
  if (fn == 1u) return (tx_noise(pos * p0) + 1.0f) * 0.5f;  // perlin, Texture.hs:109-117
  float x = vdot(pos, v3(p0, p1, p2));
  float offset = x - floorf(x);
  if (fn == 2u) return offset < 0.5f ? 0.0f : 1.0f;                        // square_wave, Texture.hs:11-14
  if (fn == 3u) return offset < 0.5f ? offset * 2 : 2 - offset * 2;        // triangle_wave, Texture.hs:16-21
  return sinf(x * 2 * 3.14159265358979323846f) * 0.5f + 0.5f;              // sine_wave, Texture.hs:23-24
}

// ------------------------------------------------------------------ trace / shade (Trace.hs:59-82, Shader.hs:65-184)
// TIER supplies closest / occluded (per lane) and closest_wave / occluded_wave (all lanes of the wave together, `valid` =
// the lane holds a ray), and carries the lights + counters.
//
// glome's trace <-> mpostshade recursion is bounded by `recurs` (maxdepth, Glome.hs:25) and by the nesting of Blend /
// AdditiveLayers materials.  On the device it is not a recursion at all: shade_vm below is the same evaluation as an
// explicit state machine per lane (a frame per trace level, a frame per material being evaluated) whose two expensive
// requests -- a closest hit, a light list -- are served for the whole wave at ONE place in the code.  So secondary rays
// re-enter the same wave-wide traversal as primary rays (the lanes that hold one are the packet), nothing is called out
// of line, and the kernel's registers are those of one traversal plus one shading step, not a nest of call frames.
struct LightCache { bool done; uint32_t mask; };  // the lazily evaluated ctxb of Trace.hs:63: visibility per light

struct LightSet { const DLight* p; int n; };  // the [Light] of the trace in progress: the call's, or a Warp material's own
template <class TIER> GD LightSet light_set(const TIER& T, uint32_t warp_mat) {
  if (warp_mat == 0xffffffffu) return LightSet{T.lights, T.nlights};
  const F4 m1 = ld4(T.S.mats, 3 * warp_mat + 1);
  return LightSet{(const DLight*)(T.S.wlights + 2 * as_u(m1.x)), (int)as_u(m1.y)};
}
template <class TIER>
GD uint32_t preshade(TIER& T, const HitCore& h, LightSet ls = LightSet{nullptr, -1}, uint32_t root = 0) {  // mpreshade, Shader.hs:65-80 (Q18)
  if (ls.n < 0) ls = LightSet{T.lights, T.nlights};
  uint32_t mask = 0;
  for (int i = 0; i < ls.n; i++) {
    const DLight& L = ls.p[i];
    V3 lvec = v3(L.pos[0], L.pos[1], L.pos[2]) - h.p;
    if (vdot(lvec, h.n) < 0) continue;
    float llen = sqrtf(vdot(lvec, lvec));
    V3 ldir = lvec * (1.0f / llen);
    if (llen > L.rad) continue;
    if (L.shadow) {
      T.cnt.shadow++;
      Ray sr; sr.o = vscaleadd(h.p, h.n, kDel); sr.d = ldir;
      if (T.occluded(sr, llen - (2 * kDel), root)) continue;
    }
    mask |= 1u << i;
  }
  return mask;
}

// mpreshade for the lanes of a wave at once (`want` = this lane's hit needs its light list): the light loop is uniform,
// so the shadow rays of one light leave together and can be walked as a packet (TIER::occluded_wave).
template <class TIER>
GD uint32_t preshade_wave(TIER& T, const HitCore& h, bool want) {
  uint32_t mask = 0;
  for (int i = 0; i < T.nlights; i++) {
    const DLight& L = T.lights[i];
    V3 lvec = v3(L.pos[0], L.pos[1], L.pos[2]) - h.p;
    float llen = sqrtf(vdot(lvec, lvec));
    V3 ldir = lvec * (1.0f / llen);
    bool lit = want && !(vdot(lvec, h.n) < 0) && !(llen > L.rad);
    if (L.shadow) {
      count_wave(T.cnt.shadow, T.cnt.w_shadow, lit);
      Ray sr; sr.o = vscaleadd(h.p, h.n, kDel); sr.d = ldir;
      bool occ = T.occluded_wave(sr, llen - (2 * kDel), lit);
      lit = lit && !occ;
    }
    if (lit) mask |= 1u << i;
  }
  return mask;
}

// Surface, Shader.hs:90-105 (Q17): ambient + sum over the visible lights of lcolor * (blinn * ks + (l . n) * kd)
template <class TIER>
GD CA surface_shade(TIER& T, const F4& m1, const F4& m2, uint32_t lightmask, const Ray& ray, const HitCore& h, LightSet ls = LightSet{nullptr, -1}) {
  if (ls.n < 0) ls = LightSet{T.lights, T.nlights};
  const float amb = m2.x, kd = m2.y, ks = m2.z, shine = m2.w;
  const V3 eyedir = vneg(ray.d), n = h.n, p = h.p;
  float ar = m1.x * amb, ag = m1.y * amb, ab = m1.z * amb;  // cscale color amb
  float dr = 0, dg = 0, db = 0;                             // foldl' cadd c_black
  for (int i = 0; i < ls.n; i++) {
    if (!((lightmask >> i) & 1u)) continue;
    const DLight& L = ls.p[i];
    V3 lvec = v3(L.pos[0], L.pos[1], L.pos[2]) - p;
    float llen = sqrtf(vdot(lvec, lvec));
    V3 ldir = lvec * (1.0f / llen);
    float fall = 1.0f / (llen * llen);  // falloff, Shader.hs:23
    V3 half = vnorm(ldir + eyedir);     // bisect, Vec.hs:331-332
    float ldotn = gmaxf(0, vdot(ldir, n));
    float blinn = 0;
    if (!(ks <= kDel)) {
      float b = gmaxf(0, powf(vdot(half, n), shine) * ldotn);
      blinn = (b != b) ? 0.0f : b;  // isNaN b
    }
    float diffuse = vdot(ldir, n);
    float w = (blinn * ks) + (diffuse * kd);
    dr = dr + (L.color[0] * fall) * w; dg = dg + (L.color[1] * fall) * w; db = db + (L.color[2] * fall) * w;
  }
  return ca(ar + dr, ag + dg, ab + db, m1.w);
}
GD Ray reflect_ray(const Ray& ray, const HitCore& h) {  // Shader.hs:111-114: reflect (Vec.hs:340-342), origin p + out * delta
  V3 outdir = vscaleadd(ray.d, h.n, (-2.0f) * vdot(ray.d, h.n));
  Ray rr; rr.o = vscaleadd(h.p, outdir, kDel); rr.d = outdir;
  return rr;
}
GD float refract_cs2(float ior, const Ray& ray, const HitCore& h, float& eta, float& c1) {  // Shader.hs:128-136
  eta = (vdot(h.n, vneg(ray.d)) > 0) ? ior : 1.0f / ior;
  c1 = vdot(ray.d, h.n);
  return 1 - (eta * eta) * (1 - (c1 * c1));
}

// mpostshade of a lean kernel (TIER::FULL == false): launched only when no secondary trace can do work (maxdepth == 1 or no
// Reflect / Refract material) and no material nests, so every child trace is `trace ... 0` = traceMiss (Trace.hs:60)
template <class TIER>
GD CA postshade_lean(TIER& T, LightCache& lc, uint32_t mat, const Ray& ray, const HitCore& h, int recurs) {
  const DScene& S = T.S;
  F4 m0 = ld4(S.mats, 3 * mat), m1 = ld4(S.mats, 3 * mat + 1);
  uint32_t kind = as_u(m0.x);
  if (kind == DM_SURFACE) {
    if (!lc.done) { lc.mask = preshade(T, h); lc.done = true; }
    return surface_shade(T, m1, ld4(S.mats, 3 * mat + 2), lc.mask, ray, h);
  }
  if (kind == DM_REFLECT) {  // Shader.hs:107-118
    float refl = m1.x;
    return ((refl > 0) && (recurs > 0)) ? ca(0, 0, 0, 0 * refl) : ca(0, 0, 0, 1);
  }
  if (kind == DM_REFRACT) {  // Shader.hs:120-155: both children are traceMiss, except total internal reflection -> ca_black
    float refl = m1.x, refr = m1.y, eta, c1;
    if ((refl > 0 || refr > 0) && (recurs > 0)) return ca(0, 0, 0, (refract_cs2(m1.z, ray, h, eta, c1) < 0) ? (0 * refl + 1 * refr) : 0.0f);
    return ca(0, 0, 0, 0);
  }
  return ca(0, 0, 0, 0);
}
// trace's fold over the hit's texture stack until opaque (Trace.hs:67-80, Q16), lean kernels
template <class TIER>
GD CA shade_hit_lean(TIER& T, const Ray& ray, const HitCore& h, LightCache& lc, int recurs) {
  CA acc = ca(0, 0, 0, 0);
  TexStack ts = h.tex;
  for (int k = 0; k < kMaxTexDepth; k++) {
    uint32_t id = tex_head(ts, (int)T.S.tex_bits);
    if (id == 0) break;
    if (acc.a + kDel >= 1) break;  // opaque, Trace.hs:50-51
    acc = cafold(acc, postshade_lean(T, lc, id - 1, ray, h, recurs));
    ts >>= T.S.tex_bits;
  }
  return acc;
}

// ---- the general evaluation (TIER::FULL): trace (Trace.hs:59-82) and mpostshade (Shader.hs:82-184) as a state machine
struct VMTrace {  // one `trace` in progress: over `root` (a record; the scene's unless a Warp material chose another) with the lights `lset`
  Ray ray; float tmax; int recurs; HitCore h; LightCache lc; CA acc; TexStack ts; int k; int mbase; uint32_t root, lset;
  V3 lo, ld;  // the hit's riray (written in Warp-capable kernels only)
};
struct VMMat {    // one material being evaluated: k = children done so far, tmp (and aux) = what they have contributed
  uint32_t mat; uint32_t k; CA tmp; float aux;
};
template <class TIER>
GD CA shade_vm(TIER& T, const Ray& ray0, float tmax, int maxdepth, bool valid, HitG* hout) {
  enum : int { S_NEED_HIT, S_HIT, S_TEX, S_MAT_NEW, S_NEED_LIGHTS, S_MAT_CHILD, S_MAT_TRACED, S_MAT_RET, S_TRACE_RET, S_DONE };
  VMTrace tr[kMaxTraceDepth];
  VMMat ms[kMaxTraceDepth * (kMaxMatNest + 1)];
  const DScene& S = T.S;
  int tl = 0, mi = 0, st = S_DONE;
  CA ret = ca(0, 0, 0, 0);
  float vm_depth = kInf;
  *hout = hit_miss();
  if (valid && maxdepth > 0) { tr[0].ray = ray0; tr[0].tmax = tmax; tr[0].recurs = maxdepth; tr[0].mbase = 0; tr[0].root = S.root_rec; tr[0].lset = 0xffffffffu; st = S_NEED_HIT; }
  for (;;) {
    // ---- the wave's requests: every lane that waits for a closest hit is traced now, together (so are the light lists)
    const bool wh = st == S_NEED_HIT;
    if (wave_any(wh)) {
      const Ray r = wh ? tr[tl].ray : ray0;
      const HitG h = T.closest_wave(r, wh ? tr[tl].tmax : tmax, wh, wh ? tr[tl].root : S.root_rec);
      if (wh) { tr[tl].h = h; if constexpr (TIER::WARP) { tr[tl].lo = h.lo; tr[tl].ld = h.ld; } if (tl == 0) *hout = h; st = S_HIT; }
    }
    const bool wl = st == S_NEED_LIGHTS;
    if (wave_any(wl)) {
      uint32_t m;
      if constexpr (TIER::WARP) m = wl ? preshade(T, tr[tl].h, light_set(T, tr[tl].lset), tr[tl].root) : 0u;  // lights and root may differ lane by lane
      else m = preshade_wave(T, tr[wl ? tl : 0].h, wl);
      if (wl) { tr[tl].lc.mask = m; tr[tl].lc.done = true; st = S_MAT_NEW; }
    }
    if (!wave_any(st != S_DONE)) break;
    // ---- this lane's own steps, until it needs the wave again
    while (st != S_DONE && st != S_NEED_HIT && st != S_NEED_LIGHTS) {
      VMTrace& t = tr[tl];
      switch (st) {
        case S_HIT:  // trace, after rayint (Trace.hs:62-66)
          if (!t.h.hit) { ret = ca(0, 0, 0, 0); st = S_TRACE_RET; break; }  // mmissshade: transparent (Shader.hs:186-187)
          t.lc.done = false; t.lc.mask = 0; t.acc = ca(0, 0, 0, 0); t.ts = t.h.tex; t.k = 0;
          st = S_TEX;
          break;
        case S_TEX: {  // fold the texture stack head first until opaque (Trace.hs:67-80, Q16)
          const uint32_t id = tex_head(t.ts, (int)S.tex_bits);
          if (t.k >= kMaxTexDepth || id == 0 || t.acc.a + kDel >= 1) { ret = t.acc; st = S_TRACE_RET; break; }
          ms[mi].mat = id - 1; ms[mi].k = 0; mi++;
          st = S_MAT_NEW;
          break;
        }
        case S_MAT_NEW: {  // mpostshade of ms[mi - 1] (Shader.hs:82-184)
          VMMat& m = ms[mi - 1];
          const F4 m0 = ld4(S.mats, 3 * m.mat), m1 = ld4(S.mats, 3 * m.mat + 1);
          const uint32_t kind = as_u(m0.x);
          if (kind == DM_SURFACE) {
            if (!t.lc.done) { st = S_NEED_LIGHTS; break; }  // the lazily evaluated light list (Trace.hs:63), forced here
            if constexpr (TIER::WARP) ret = surface_shade(T, m1, ld4(S.mats, 3 * m.mat + 2), t.lc.mask, t.ray, t.h, light_set(T, t.lset));
            else ret = surface_shade(T, m1, ld4(S.mats, 3 * m.mat + 2), t.lc.mask, t.ray, t.h);
            st = S_MAT_RET;
          } else if (TIER::WARP && kind == DM_WARP) {  // Shader.hs:157-175: the frame through the hit's own ray first
            m.k = 1;
            if (t.recurs - 1 <= 0) { ret = ca(0, 0, 0, 0); vm_depth = kInf; st = S_MAT_TRACED; break; }  // traceMiss: RayMiss, depth = infinity
            VMTrace& c = tr[tl + 1];
            c.ray.o = t.lo; c.ray.d = t.ld; c.tmax = kInf; c.recurs = t.recurs - 1; c.mbase = mi; c.root = as_u(m0.y); c.lset = t.lset;
            T.cnt.secondary++;
            tl++;
            st = S_NEED_HIT;
          } else if (kind == DM_REFLECT || kind == DM_REFRACT) {  // Shader.hs:107-118, 120-155: the reflected ray comes first
            const bool go = kind == DM_REFLECT ? (m1.x > 0) : (m1.x > 0 || m1.y > 0);
            if (!(go && t.recurs > 0)) { ret = kind == DM_REFLECT ? ca(0, 0, 0, 1) : ca(0, 0, 0, 0); st = S_MAT_RET; break; }
            m.k = 1;
            if (t.recurs - 1 <= 0) { ret = ca(0, 0, 0, 0); st = S_MAT_TRACED; break; }  // `trace _ _ _ _ _ 0 = traceMiss` (Trace.hs:60)
            VMTrace& c = tr[tl + 1];
            c.ray = reflect_ray(t.ray, t.h); c.tmax = kInf; c.recurs = t.recurs - 1; c.mbase = mi; c.root = t.root; c.lset = t.lset;
            T.cnt.secondary++;
            tl++;
            st = S_NEED_HIT;
          } else if ((kind == DM_LAYERS || kind == DM_BLEND) && mi - t.mbase <= kMaxMatNest) {
            m.k = 0; m.tmp = ca(0, 0, 0, 1);  // AdditiveLayers: (r, g, b) sums and the running product of (1 - alpha)
            if (kind == DM_LAYERS && as_u(m0.z) == 0) { ret = ca(0, 0, 0, 1 - m.tmp.a); st = S_MAT_RET; break; }
            ms[mi].mat = kind == DM_LAYERS ? S.matkids[as_u(m0.y)] : as_u(m0.y); ms[mi].k = 0; mi++;
            st = S_MAT_NEW;
          } else { ret = ca(0, 0, 0, 0); st = S_MAT_RET; }
          break;
        }
        case S_MAT_CHILD: {  // a child material of ms[mi - 1] has been evaluated: `ret`
          VMMat& m = ms[mi - 1];
          const F4 m0 = ld4(S.mats, 3 * m.mat);
          if (as_u(m0.x) == DM_LAYERS) {  // casum, Clr.hs:93-103 (alphas :82-85)
            m.tmp.r = m.tmp.r + ret.r * ret.a; m.tmp.g = m.tmp.g + ret.g * ret.a; m.tmp.b = m.tmp.b + ret.b * ret.a;
            m.tmp.a = m.tmp.a * (1 - aclamp(ret.a));
            m.k++;
            if (m.k < as_u(m0.z)) { ms[mi].mat = S.matkids[as_u(m0.y) + m.k]; ms[mi].k = 0; mi++; st = S_MAT_NEW; }
            else { ret = ca(m.tmp.r, m.tmp.g, m.tmp.b, 1 - m.tmp.a); st = S_MAT_RET; }
          } else {  // Blend, Shader.hs:181-184
            if (m.k == 0) { m.tmp = ret; m.k = 1; ms[mi].mat = as_u(m0.z); ms[mi].k = 0; mi++; st = S_MAT_NEW; }
            else {
              const F4 m1 = ld4(S.mats, 3 * m.mat + 1);
              ret = caweight(m.tmp, ret, tx_weight(as_u(m1.x), m0.w, m1.y, m1.z, m1.w, t.h.p));  // constant or a solid texture function of the hit point
              st = S_MAT_RET;
            }
          }
          break;
        }
        case S_MAT_TRACED: {  // a child trace of ms[mi - 1] (Reflect / Refract) has returned: `ret`
          VMMat& m = ms[mi - 1];
          const F4 m0 = ld4(S.mats, 3 * m.mat), m1 = ld4(S.mats, 3 * m.mat + 1);
          if (as_u(m0.x) == DM_REFLECT) { ret = ca(ret.r, ret.g, ret.b, ret.a * m1.x); st = S_MAT_RET; break; }
          if (TIER::WARP && as_u(m0.x) == DM_WARP) {
            if (m.k == 1) {  // the frame is in (colour `ret`, depth `vm_depth`); now the other scene, up to that depth, through the warped ray
              m.tmp = ret; m.aux = vm_depth; m.k = 2;
              if (t.recurs - 1 <= 0) { ret = ca(0, 0, 0, 0); vm_depth = kInf; break; }
              const Xf6 x = load_xf(S, as_u(m0.w));
              VMTrace& c = tr[tl + 1];
              c.ray.o = mat_point(x.f0, x.f1, x.f2, t.h.p); c.ray.d = vnorm(mat_vec(x.f0, x.f1, x.f2, vnorm(t.ray.d)));  // xfm_ray M (Ray pos (vnorm dir)), Vec.hs:553-555
              c.tmax = m.aux; c.recurs = t.recurs - 1; c.mbase = mi; c.root = as_u(m0.z); c.lset = m.mat;
              T.cnt.secondary++;
              tl++;
              st = S_NEED_HIT;
            } else { ret = (m.aux < vm_depth) ? m.tmp : ret; st = S_MAT_RET; }  // if ridepth fint < ridepth wint then fcolor else wcolor
            break;
          }
          const float refl = m1.x, refr = m1.y;
          if (m.k == 1) {  // the reflected part is in; now the transmitted ray (unnormalised, as written, Shader.hs:141)
            m.tmp = ret; m.k = 2;
            float eta, c1;
            const float cs2 = refract_cs2(m1.z, t.ray, t.h, eta, c1);
            if (cs2 < 0) { ret = ca(0, 0, 0, 1); break; }  // total internal reflection: ca_black (state stays S_MAT_TRACED, k == 2)
            if (t.recurs - 1 <= 0) { ret = ca(0, 0, 0, 0); break; }
            const V3 tv = (t.ray.d * eta) + (t.h.n * (eta * c1 - sqrtf(cs2)));
            VMTrace& c = tr[tl + 1];
            c.ray.o = vscaleadd(t.h.p, tv, kDel); c.ray.d = tv; c.tmax = kInf; c.recurs = t.recurs - 1; c.mbase = mi; c.root = t.root; c.lset = t.lset;
            T.cnt.secondary++;
            tl++;
            st = S_NEED_HIT;
          } else {
            const CA cr = m.tmp, ct = ret;
            ret = ca(cr.r * refl + ct.r * refr, cr.g * refl + ct.g * refr, cr.b * refl + ct.b * refr, cr.a * refl + ct.a * refr);
            st = S_MAT_RET;
          }
          break;
        }
        case S_MAT_RET:  // ms[mi - 1] is evaluated: `ret`
          mi--;
          if (mi == t.mbase) { t.acc = cafold(t.acc, ret); t.ts >>= S.tex_bits; t.k++; st = S_TEX; }
          else st = S_MAT_CHILD;
          break;
        case S_TRACE_RET:  // the trace of level tl is evaluated: `ret`; vm_depth = ridepth of its Rayint (Warp compares those)
          vm_depth = (t.recurs > 0 && t.h.hit) ? t.h.t : kInf;
          if (tl == 0) st = S_DONE;
          else { tl--; st = S_MAT_TRACED; }
          break;
        default: st = S_DONE; break;
      }
    }
  }
  return ret;
}

// the pixel loop's entry: `Trace.trace lights shader sld ray infinity maxdepth` (Glome.hs:33), maxdepth <= kMaxTraceDepth.
// Called by all lanes of a wave together (`valid` = the lane has a pixel).  Lean kernels: the primary rays and, where the
// first material of the hit is a Surface -- which always forces the light list (Trace.hs:63, Shader.hs:96) -- the shadow
// rays are traced wave-wide.  Full kernels: shade_vm.
template <class TIER> GD CA trace_primary(TIER& T, const Ray& ray, float tmax, int maxdepth, bool valid, HitG* hout) {
  if constexpr (TIER::FULL) return shade_vm(T, ray, tmax, maxdepth, valid, hout);
  else {
    if (maxdepth <= 0) { *hout = hit_miss(); return ca(0, 0, 0, 0); }
    HitG h = T.closest_wave(ray, tmax, valid);
    *hout = h;
    LightCache lc; lc.done = false; lc.mask = 0;
    bool eager = false;
    if (valid && h.hit) {
      uint32_t id = tex_head(h.tex, (int)T.S.tex_bits);
      if (id != 0) eager = as_u(ld4(T.S.mats, 3 * (id - 1)).x) == DM_SURFACE;
    }
    if (wave_any(eager)) {
      uint32_t m = preshade_wave(T, h, eager);
      if (eager) { lc.mask = m; lc.done = true; }
    }
    if (!(valid && h.hit)) return ca(0, 0, 0, 0);  // mmissshade: transparent (Shader.hs:186-187)
    return shade_hit_lean(T, ray, h, lc, maxdepth);
  }
}

// ------------------------------------------------------------------ pixel mapping (Glome.hs:27-33, 119-140; Q19)
// The three quotients are IEEE divisions also on the device (the kernels are built with the fast fp32 division, whose
// reciprocal is an ulp off): 96 / 192 must be exactly 0.5.  The centre column of an even-width frame then gets xc = 0 and,
// with an axis-aligned camera, rays with an exactly zero x component -- for which the reference's slab test misses every box
// and every bih (Q1: (hi - o) / 0 = +inf on the entry side).  The reference renders that column empty; an ulp of error in xc
// would fill it in.
// (through fp64: the quotient of two floats taken in double and rounded once more is the correctly rounded fp32 quotient
// -- 53 >= 2 * 24 + 2 bits -- whatever the fp32 division of the build is; __fdiv_rn follows the build's setting)
GD float div_ieee(float a, float b) { return (float)((double)a / (double)b); }
GD void get_coordsf(int width, int height, float xf, float yf, float& xc, float& yc) {
  float widthf = (float)width, heightf = (float)height;
  xc = (((div_ieee(xf, widthf)) * 2) - 1) * div_ieee(widthf, heightf);
  yc = -(((div_ieee(yf, heightf)) * 2) - 1);
}
GD Ray primary_ray(const DCamera& c, float xc, float yc) {  // get_rayint, Glome.hs:27-33
  V3 fwd = v3(c.fwd[0], c.fwd[1], c.fwd[2]), right = v3(c.right[0], c.right[1], c.right[2]), up = v3(c.up[0], c.up[1], c.up[2]);
  V3 a = right * (-xc), b = up * yc;
  Ray r;
  r.o = v3(c.pos[0], c.pos[1], c.pos[2]);
  r.d = vnorm(v3(fwd.x + a.x + b.x, fwd.y + a.y + b.y, fwd.z + a.z + b.z));  // vadd3
  return r;
}
GD float cap1(float x) { return x >= 1 ? 1 - kDel : x; }  // Glome.hs:98-101
GD uint32_t rgbf(float r, float g, float b) {            // Glome.hs:107-110 (wraps like Word32 arithmetic)
  int ri = (int)floorf(cap1(r) * 256), gi = (int)floorf(cap1(g) * 256), bi = (int)floorf(cap1(b) * 256);
  return (uint32_t)ri * 65536u + (uint32_t)gi * 256u + (uint32_t)bi;
}


// ------------------------------------------------------------------ adaptive sampler (renderTileSubsample, Glome.hs:179-323; Q21)
struct TC { float r, g, b, a, d; };  // TColor, Glome.hs:153
GD TC tc(float r, float g, float b, float a, float d) { TC c; c.r = r; c.g = g; c.b = b; c.a = a; c.d = d; return c; }
GD TC tc_blank() { return tc(0, 0, 0, 0, kInf); }  // out-of-tile neighbours and the initial buffer, Glome.hs:231-235
GD float gabsf(float a) { return a < 0 ? -a : a; }  // fabs, Vec.hs:80-82
GD float ccmp(const TC& p, const TC& q) {            // cCmp, Glome.hs:179-189
  float md;
  if (p.d == 0 && q.d == 0) md = 0;                  // muldiff 0 0 = 0
  else md = (p.d > q.d) ? (p.d / q.d) - 1 : (q.d / p.d) - 1;
  return gabsf(q.r - p.r) + gabsf(q.g - p.g) + gabsf(q.b - p.b) + gabsf(q.a - p.a) + md;
}
GD TC cavg4(const TC& a, const TC& b, const TC& c, const TC& d) {  // cAvg, Glome.hs:191-197
  return tc((a.r + b.r + c.r + d.r) * 0.25f, (a.g + b.g + c.g + d.g) * 0.25f, (a.b + b.b + c.b + d.b) * 0.25f,
            (a.a + b.a + c.a + d.a) * 0.25f, (a.d + b.d + c.d + d.d) * 0.25f);
}
GD TC cavg2(const TC& a, const TC& b) {  // cAvg2, Glome.hs:199-205
  return tc((a.r + b.r) * 0.5f, (a.g + b.g) * 0.5f, (a.b + b.b) * 0.5f, (a.a + b.a) * 0.5f, (a.d + b.d) * 0.5f);
}
// Pass p (1..5) touches tile-local pixel (dx, dy)?  Passes 1-2: the even lattice, split by (dx+dy) mod 4; pass 3: the
// odd-odd lattice; pass 4: the remaining pixels; pass 5: every pixel (Glome.hs:241-319).
GD bool ss_candidate(int p, int dx, int dy) {
  switch (p) {
    case 1: return !(dx & 1) && !(dy & 1) && ((dx + dy) & 3) == 0;
    case 2: return !(dx & 1) && !(dy & 1) && ((dx + dy) & 3) == 2;
    case 3: return (dx & 1) && (dy & 1);
    case 4: return ((dx + dy) & 1) == 1;
    default: return true;
  }
}
// neighbour offsets (a, b, c, d) of pass p, in the reference's order (decide compares a-c and b-d, Glome.hs:213-219)
GD void ss_neighbours(int p, int* ox, int* oy) {
  switch (p) {
    case 2: ox[0] = -2; oy[0] = 0; ox[1] = 0; oy[1] = 2; ox[2] = 2; oy[2] = 0; ox[3] = 0; oy[3] = -2; break;    // :255-258
    case 3: ox[0] = -1; oy[0] = -1; ox[1] = 1; oy[1] = -1; ox[2] = 1; oy[2] = 1; ox[3] = -1; oy[3] = 1; break;  // :274-277
    case 4: ox[0] = -1; oy[0] = 0; ox[1] = 0; oy[1] = 1; ox[2] = 1; oy[2] = 0; ox[3] = 0; oy[3] = -1; break;    // :287-290
    default: ox[0] = 0; oy[0] = 0; ox[1] = 0; oy[1] = 1; ox[2] = 1; oy[2] = 1; ox[3] = 1; oy[3] = 0; break;     // pass 5, :303-306
  }
}
// One wave = one block of a tile's candidate lattice.  Block shapes are chosen so a block holds 64 candidates of its pass:
//   passes 1, 2   32x16 pixels: the even lattice (2i, 2j), i + j even (pass 1) or odd (pass 2)
//   pass 3        16x16 pixels: the odd-odd lattice
//   pass 4        16x8 pixels: dx + dy odd
//   pass 5        8x8 pixels: every pixel
GHD void ss_block_shape(int p, int& bw, int& bh) {
  switch (p) { case 1: case 2: bw = 32; bh = 16; break; case 3: bw = 16; bh = 16; break; case 4: bw = 16; bh = 8; break; default: bw = 8; bh = 8; break; }
}
// A work item (region) is a rectangle of rw x rh blocks.  The candidates of a region that need a sample are compacted
// over the WHOLE region as they are found and traced 64 at a time, so a region costs one partly filled packet at most.
// What a tile costs is packet walks (each a chain of dependent fetches), not candidates: large regions mean fewer
// walks, small regions mean more of them side by side -- and a tile's five passes are a chain, which only other tiles
// and other frames hide.  `size` 0: a frame or two per launch (1 block per region in passes 1-2, 2x2 after); 1: three to
// seven frames; 2: a batch of eight (pass 1 -- every candidate is sampled: 545 of a 65x65 tile -- in columns of blocks,
// the later passes, where a few percent of the candidates need a sample, the whole tile).  Measured on S3, ms per frame
// with 1 / 4 / 8 frames per launch (profiles/r02_f_ss_regions.log): size 0: 0.89 / 0.54 / 0.53, size 1: 1.25 / 0.49 /
// 0.45, size 2: 1.98 / 0.67 / 0.43.
// Blocks outside a clipped tile are skipped.
GHD void ss_region_shape(int p, int size, int& rw, int& rh) {
  if (size >= 2) { rw = p == 1 ? 1 : (p == 2 ? 3 : 5); rh = 5; }
  else if (size == 1) { rw = p <= 2 ? 1 : 3; rh = p == 1 ? 2 : (p == 2 || p == 4 ? 5 : 3); }
  else { rw = rh = p <= 2 ? 1 : 2; }
}
GHD int ss_regions_per_tile(int p, int tile_size, int rw, int rh, int& nrx) {
  int bw, bh; ss_block_shape(p, bw, bh);
  nrx = (tile_size + bw * rw - 1) / (bw * rw);
  return nrx * ((tile_size + bh * rh - 1) / (bh * rh));
}
// candidate pixel (tile-local) of `lane` in block (bx, by) of pass p; every lane of a block maps to a distinct candidate
GD void ss_block_pixel(int p, int bx, int by, int lane, int& dx, int& dy) {
  const int k = lane & 7, ly = lane >> 3;
  switch (p) {
    case 1: case 2: { int j = by * 8 + ly; int i = bx * 16 + 2 * k + ((j + (p - 1)) & 1); dx = 2 * i; dy = 2 * j; break; }
    case 3: dx = bx * 16 + 2 * k + 1; dy = by * 16 + 2 * ly + 1; break;
    case 4: dy = by * 8 + ly; dx = bx * 16 + 2 * k + (1 - (dy & 1)); break;
    default: dx = bx * 8 + k; dy = by * 8 + ly; break;
  }
}
// pass 5's write: the new sample blended with the pixel's neighbourhood; edge pixels use two-sample averages (:309-316)
GD TC ss_pass5_blend(const TC& color, const TC& a, const TC& b, const TC& c, const TC& d, bool last_col, bool last_row) {
  if (last_col) return last_row ? color : cavg2(color, cavg2(a, b));
  return last_row ? cavg2(color, cavg2(a, d)) : cavg2(color, cavg4(a, b, c, d));
}

}  // namespace glome
