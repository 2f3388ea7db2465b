// host_shim.hpp -- TEST INFRASTRUCTURE: what glome_amd/csrc/rt_device.hpp needs to compile with g++ for the CPU suite
// (tests/hostsim): the function-attribute macros, the bit casts, a one-lane "wave" and a host LaneStack.  With one lane a
// packet walk is a single-ray traversal, so the same traversal / shading logic the kernels run is checked against the
// oracle on a GPU-less machine; the wave intrinsics, the scalar loads and the hand-written assembly are device-only and are
// covered by the -m gpu suite.  Nothing in the product includes this file: the test build includes it FIRST, and the device
// headers then see GLOME_DEVICE_HEADERS_ON_HOST.
#pragma once
#define GLOME_DEVICE_HEADERS_ON_HOST 1
#include <cmath>
#include <cstdint>
#include <cstring>
#define GD inline
#define GHD inline
#define GDN
#include "../../glome_amd/csrc/rt_types.h"

namespace glome {
GD float as_f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
GD uint32_t as_u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
typedef unsigned long long LaneMask;
GD float min_nn(float a, float b) { return fminf(a, b); }
GD float max_nn(float a, float b) { return fmaxf(a, b); }
GD LaneMask wave_ballot(bool p) { return p ? 1ull : 0ull; }
GD bool wave_any(bool p) { return p; }
GD bool lane_of(LaneMask m) { return (m & 1ull) != 0; }
GD uint32_t uni(uint32_t v) { return v; }
GD LaneMask uni(LaneMask m) { return m; }
GD uint32_t first_lane_value(LaneMask, uint32_t v) { return v; }
GD F4 ld4u(const F4* p, uint32_t i) { return p[i]; }
GD void ld_tri_u(const F4* p, uint32_t tri, F4& q0, F4& q1, F4& q2) { q0 = p[3 * tri]; q1 = p[3 * tri + 1]; q2 = p[3 * tri + 2]; }
GD int wave_count(bool pred) { return pred ? 64 : 0;  }
// 1 / (a direction component): the device takes v_rcp_f32 (1 ulp) everywhere; the host build divides exactly.  The one place
// the last bit of it decides something -- a box ending on a split plane -- is held against the GPU by
// test_gpu_equals_the_host_build_where_boxes_end_on_split_planes on many seeds.
GD float dir_rcp(float x) { return 1.0f / x; }

// one column per "lane" with stride 1; a packet entry keeps its one-bit lane mask in bit 31 of the reference
struct LaneStack {
  uint32_t* node; float* nearv; float* farv;
  int cap;
  uint32_t* ovf;
  int ovf_cap;
  static constexpr int STRIDE = 1;
  static constexpr bool has_ref_row = true;
  GD int total_cap() const { return cap + ovf_cap; }
  GD void push(int sp, uint32_t n, float a, float b) {
    if (sp < cap) { node[sp * STRIDE] = n; nearv[sp * STRIDE] = a; farv[sp * STRIDE] = b; }
    else { uint32_t* o = ovf + (size_t)(sp - cap) * 3 * STRIDE; o[0] = n; o[STRIDE] = as_u(a); o[2 * STRIDE] = as_u(b); }
  }
  GD void pop(int sp, uint32_t& n, float& a, float& b) const {
    if (sp < cap) { n = node[sp * STRIDE]; a = nearv[sp * STRIDE]; b = farv[sp * STRIDE]; }
    else { const uint32_t* o = ovf + (size_t)(sp - cap) * 3 * STRIDE; n = o[0]; a = as_f(o[STRIDE]); b = as_f(o[2 * STRIDE]); }
  }
  GD void push_wave(int sp, uint32_t ref, LaneMask m, float a, float b) { push(sp, ref | ((uint32_t)(m & 1ull) << 31), a, b); }
  GD void pop_wave(int sp, uint32_t& ref, LaneMask& m, float& a, float& b) const { uint32_t w; pop(sp, w, a, b); ref = w & 0x7fffffffu; m = w >> 31; }
};
}  // namespace glome
