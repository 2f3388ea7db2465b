// bih_packet_asm.hpp -- the packet walk of a triangle BIH, hand-written for gfx950 (device code only).
//
// rt_device.hpp's bih_tri_packet is the reference implementation of one wave walking `rayint_bih` / `shadow_bih`
// (Bih.hs:332-368, 510-544) once for its 64 rays.  The kernel built from it is bound by the CU's scalar unit
// (profiles/r02_pmc_S3_mode0.json: 0.67 scalar instructions per cycle per CU, vector issue at 0.28 of its peak): the compiler
// spends ~25 scalar instructions on a branch step, ~30 on a triangle (packing operands for v_pk_* arithmetic, lane masks
// combined with s_and after every compare) and ~20 on a pop.  This file is the same walk written out:
//
//   branch step  ~17 scalar: the three axes and the two directions are separate straight-line pieces (the direction signs
//                of a packet are fixed for the whole walk, so the walk is instantiated per octant: no direction test, no
//                operand swap); s_and_b64 sets SCC, so a vote needs no compare; the stack pointer lives in m0 for the whole
//                walk (v_writelane / v_readlane take it from there) and the lane's LDS address is kept incrementally.
//   triangle     ~8 scalar: two scalar loads (32 + 16 bytes; a leaf's triangles are fetched two at a time, one memory
//                round trip for both), the Moeller-Trumbore arithmetic with the scalar registers as
//                direct operands (the same IEEE operations in the same order as the compiler's tri_test, so results are
//                bit-identical to every other kernel instance), and the four rejection tests as a chain of v_cmpx, which
//                narrows EXEC instead of building masks: the survivors' updates are plain moves.
//   pop          ~7 scalar.
//
// What it declines goes back to the C++ loop for one step (status codes below): pushes and pops beyond the LDS part of the
// stack (global overflow columns) and leaves of more than six items.  The wave-uniform part of the stack (reference and lane
// mask of entry k in lane k of three vector registers) never leaves the asm block in registers: before a C++ step it is
// written to the wave's dump block in global memory, on re-entry (sp > 0) read back -- a register whose inactive lanes carry
// data must not be visible to the compiler, which may spill or copy it under a partial EXEC mask.  MODE 1 (ordered early-out closest hit) and MODE 2
// (any hit) only; the faithful / counting variants stay in C++.
//
// Invariant used (MODE 1): far <= best_t on the current path at all times (the root interval is clipped with the running
// best, children only shrink it, a pop clips with best_t, an accepted hit sets both), so a triangle hit within [.., far]
// always replaces the running best: `!(best_t < t)` (nearest: ties -> later item, Solid.hs:37-44) needs no test.
//
// Scalar registers K0..K35 (below), vcc, scc and m0 (saved and restored) are scratch, named in the clobber list.
#pragma once
#if defined(__HIPCC__)

namespace glome {

// The block's scratch scalar registers: 36 consecutive ones from GLOME_PKW_BASE (a multiple of 4: the wide loads want aligned
// destinations), K0 .. K35 below.  They sit LOW in the register file on purpose: a wave's scalar registers are allocated in
// blocks of 16 and a kernel whose highest one is above 80 loses a wave per SIMD, above 96 two (MI355X_MICROARCH.md,
// "Residency"), so the walk must not be what pushes the kernel's count up.
//   K0..K3 the node | K4..K15 triangle record A | K16 K17 near-child mask | K18 K19 far-child mask (K16..K19 double as
//   record B's e2 words) | K20 K21 a popped entry's mask, temporaries | K22 K23 EXEC at entry | K24 the caller's m0 |
//   K25 the leaf's remaining-triangle bits | K26 byte offset of the next record | K27 its record index | K28..K35 record B
#ifndef GLOME_PKW_BASE
#define GLOME_PKW_BASE 36
#endif
#if GLOME_PKW_BASE == 36
#define K0 "36"
#define K1 "37"
#define K2 "38"
#define K3 "39"
#define K4 "40"
#define K5 "41"
#define K6 "42"
#define K7 "43"
#define K8 "44"
#define K9 "45"
#define K10 "46"
#define K11 "47"
#define K12 "48"
#define K13 "49"
#define K14 "50"
#define K15 "51"
#define K16 "52"
#define K17 "53"
#define K18 "54"
#define K19 "55"
#define K20 "56"
#define K21 "57"
#define K22 "58"
#define K23 "59"
#define K24 "60"
#define K25 "61"
#define K26 "62"
#define K27 "63"
#define K28 "64"
#define K29 "65"
#define K30 "66"
#define K31 "67"
#define K32 "68"
#define K33 "69"
#define K34 "70"
#define K35 "71"
#elif GLOME_PKW_BASE == 40
#define K0 "40"
#define K1 "41"
#define K2 "42"
#define K3 "43"
#define K4 "44"
#define K5 "45"
#define K6 "46"
#define K7 "47"
#define K8 "48"
#define K9 "49"
#define K10 "50"
#define K11 "51"
#define K12 "52"
#define K13 "53"
#define K14 "54"
#define K15 "55"
#define K16 "56"
#define K17 "57"
#define K18 "58"
#define K19 "59"
#define K20 "60"
#define K21 "61"
#define K22 "62"
#define K23 "63"
#define K24 "64"
#define K25 "65"
#define K26 "66"
#define K27 "67"
#define K28 "68"
#define K29 "69"
#define K30 "70"
#define K31 "71"
#define K32 "72"
#define K33 "73"
#define K34 "74"
#define K35 "75"
#elif GLOME_PKW_BASE == 44
#define K0 "44"
#define K1 "45"
#define K2 "46"
#define K3 "47"
#define K4 "48"
#define K5 "49"
#define K6 "50"
#define K7 "51"
#define K8 "52"
#define K9 "53"
#define K10 "54"
#define K11 "55"
#define K12 "56"
#define K13 "57"
#define K14 "58"
#define K15 "59"
#define K16 "60"
#define K17 "61"
#define K18 "62"
#define K19 "63"
#define K20 "64"
#define K21 "65"
#define K22 "66"
#define K23 "67"
#define K24 "68"
#define K25 "69"
#define K26 "70"
#define K27 "71"
#define K28 "72"
#define K29 "73"
#define K30 "74"
#define K31 "75"
#define K32 "76"
#define K33 "77"
#define K34 "78"
#define K35 "79"
#elif GLOME_PKW_BASE == 48
#define K0 "48"
#define K1 "49"
#define K2 "50"
#define K3 "51"
#define K4 "52"
#define K5 "53"
#define K6 "54"
#define K7 "55"
#define K8 "56"
#define K9 "57"
#define K10 "58"
#define K11 "59"
#define K12 "60"
#define K13 "61"
#define K14 "62"
#define K15 "63"
#define K16 "64"
#define K17 "65"
#define K18 "66"
#define K19 "67"
#define K20 "68"
#define K21 "69"
#define K22 "70"
#define K23 "71"
#define K24 "72"
#define K25 "73"
#define K26 "74"
#define K27 "75"
#define K28 "76"
#define K29 "77"
#define K30 "78"
#define K31 "79"
#define K32 "80"
#define K33 "81"
#define K34 "82"
#define K35 "83"
#elif GLOME_PKW_BASE == 60
#define K0 "60"
#define K1 "61"
#define K2 "62"
#define K3 "63"
#define K4 "64"
#define K5 "65"
#define K6 "66"
#define K7 "67"
#define K8 "68"
#define K9 "69"
#define K10 "70"
#define K11 "71"
#define K12 "72"
#define K13 "73"
#define K14 "74"
#define K15 "75"
#define K16 "76"
#define K17 "77"
#define K18 "78"
#define K19 "79"
#define K20 "80"
#define K21 "81"
#define K22 "82"
#define K23 "83"
#define K24 "84"
#define K25 "85"
#define K26 "86"
#define K27 "87"
#define K28 "88"
#define K29 "89"
#define K30 "90"
#define K31 "91"
#define K32 "92"
#define K33 "93"
#define K34 "94"
#define K35 "95"
#else
#error "GLOME_PKW_BASE: 36, 40, 44, 48 or 60"
#endif

enum : int { PKW_DONE = 0, PKW_PUSH_OVERFLOW = 1, PKW_BIG_LEAF = 2, PKW_POP_OVERFLOW = 3 };

// one axis piece of the branch step.  O / R: this lane's origin / reciprocal direction on the axis; NP / FP: the planes
// that end the near child's interval and start the far child's; NC / FC: the child references (s62 is shifted into place
// at the top of the piece).  s[60:63] = the node.
#define GLOME_PKW_AXIS(TAG, O, R, NP, FP, NC, FC)                                                    \
  "L_ax" TAG "_%=:\n"                                                                                 \
  "  s_lshr_b32 s" K2 ", s" K2 ", 2\n"                                                                        \
  "  v_sub_f32 %[t1], " NP ", %[" O "]\n"                                                             \
  "  v_sub_f32 %[t2], " FP ", %[" O "]\n"                                                             \
  "  v_mul_f32 %[t1], %[t1], %[" R "]\n"                                                              \
  "  v_mul_f32 %[t2], %[t2], %[" R "]\n"                                                              \
  "  v_cmp_lt_f32 vcc, %[t2], %[far]\n"                                                               \
  "  s_and_b64 s[" K18 ":" K19 "], vcc, %[am]\n"      /* lanes that reach the far child */                      \
  "  v_cmp_lt_f32 vcc, %[near], %[t1]\n"                                                              \
  "  s_and_b64 s[" K16 ":" K17 "], vcc, %[am]\n"      /* lanes that reach the near child; SCC = any */          \
  "  s_cbranch_scc0 L_no1" TAG "_%=\n"                                                                \
  "  s_cmp_lg_u64 s[" K18 ":" K19 "], 0\n"                                                                      \
  "  s_cbranch_scc0 L_nopush" TAG "_%=\n"                                                             \
  "  s_cmp_ge_u32 m0, %[cap]\n"                                                                       \
  "  s_cbranch_scc1 L_slow_%=\n"                                                                      \
  "  v_max_f32 %[t2], %[t2], %[near]\n"     /* the far child's interval starts here */                \
  "  v_writelane_b32 %[ur], " FC ", m0\n"   /* the uniform part of the entry: lane `sp` of ur / ulo / uhi */ \
  "  v_writelane_b32 %[ulo], s" K18 ", m0\n"                                                               \
  "  v_writelane_b32 %[uhi], s" K19 ", m0\n"                                                               \
  "  ds_write_b32 %[av], %[t2]\n"           /* this lane's (near, far) of the far child */            \
  "  ds_write_b32 %[av], %[far] offset:%[row1]\n"                                                     \
  "  v_add_u32 %[av], 0x100, %[av]\n"                                                                 \
  "  s_add_u32 m0, m0, 1\n"                                                                           \
  "L_nopush" TAG "_%=:\n"                                                                             \
  "  v_min_f32 %[far], %[t1], %[far]\n"                                                               \
  "  s_mov_b32 %[ref], " NC "\n"                                                                      \
  "  s_mov_b64 %[am], s[" K16 ":" K17 "]\n"                                                                     \
  "  s_branch L_node_%=\n"                                                                            \
  "L_no1" TAG "_%=:\n"                                                                                \
  "  s_cmp_lg_u64 s[" K18 ":" K19 "], 0\n"                                                                      \
  "  s_cbranch_scc0 L_pop_%=\n"             /* nobody goes on below this node */                      \
  "  v_max_f32 %[near], %[t2], %[near]\n"                                                             \
  "  s_mov_b32 %[ref], " FC "\n"                                                                      \
  "  s_mov_b64 %[am], s[" K18 ":" K19 "]\n"                                                                     \
  "  s_branch L_node_%=\n"

// rays running towards +axis take the left child (s62, ends at plane s60) first, the others the right one (s63, plane s61)
#define GLOME_PKW_AXIS_FWD(TAG, O, R) GLOME_PKW_AXIS(TAG, O, R, "s" K0, "s" K1, "s" K2, "s" K3)
#define GLOME_PKW_AXIS_BWD(TAG, O, R) GLOME_PKW_AXIS(TAG, O, R, "s" K1, "s" K0, "s" K3, "s" K2)

// One triangle test.  P1 / E1 / E2: the scalar registers that hold the record's (p1, .) (e1, .) (e2, .) words (rt_types.h).
// The arithmetic is tri_test's (Triangle.hs:45-73) operation for operation as hipcc contracts it under -ffp-contract=on:
// cross(a, b).y = fma(a.z, b.x, -(a.x * b.z)) and cyclic; dot(a, b) = fma(a.z, b.z, fma(a.x, b.x, a.y * b.y)).
// Entered with EXEC = the lanes whose interval reaches the leaf; leaves EXEC = the lanes that hit.
#define GLOME_PKW_TRI_TEST(P1X, P1Y, P1Z, E1X, E1Y, E1Z, E2X, E2Y, E2Z)                               \
  "  v_subrev_f32 %[Dx], " P1X ", %[ox]\n"  /* D = o - p1 */                                          \
  "  v_subrev_f32 %[Dy], " P1Y ", %[oy]\n"                                                            \
  "  v_subrev_f32 %[Dz], " P1Z ", %[oz]\n"                                                            \
  "  v_mul_f32_e64 %[s2y], -%[Dx], " E1Z "\n"  /* s2 = D x e1 */                                      \
  "  v_mul_f32_e64 %[s2x], -%[Dz], " E1Y "\n"                                                         \
  "  v_mul_f32_e64 %[s2z], -%[Dy], " E1X "\n"                                                         \
  "  v_fma_f32 %[s2y], %[Dz], " E1X ", %[s2y]\n"                                                      \
  "  v_fma_f32 %[s2x], %[Dy], " E1Z ", %[s2x]\n"                                                      \
  "  v_fma_f32 %[s2z], %[Dx], " E1Y ", %[s2z]\n"                                                      \
  "  v_mul_f32_e64 %[s1y], -%[dx], " E2Z "\n"  /* s1 = dir x e2 */                                    \
  "  v_mul_f32_e64 %[s1x], -%[dz], " E2Y "\n"                                                         \
  "  v_mul_f32_e64 %[s1z], -%[dy], " E2X "\n"                                                         \
  "  v_fma_f32 %[s1y], %[dz], " E2X ", %[s1y]\n"                                                      \
  "  v_fma_f32 %[s1x], %[dy], " E2Z ", %[s1x]\n"                                                      \
  "  v_fma_f32 %[s1z], %[dx], " E2Y ", %[s1z]\n"                                                      \
  "  v_mul_f32 %[div], " E1Y ", %[s1y]\n"   /* divisor = s1 . e1 */                                   \
  "  v_fmac_f32 %[div], " E1X ", %[s1x]\n"                                                            \
  "  v_fmac_f32 %[div], " E1Z ", %[s1z]\n"                                                            \
  "  v_mul_f32 %[b2], %[dy], %[s2y]\n"      /* dir . s2 */                                            \
  "  v_mul_f32 %[b1], %[Dy], %[s1y]\n"      /* D . s1 */                                              \
  "  v_mul_f32 %[t], " E2Y ", %[s2y]\n"     /* e2 . s2 */                                             \
  "  v_rcp_f32 %[inv], %[div]\n"                                                                      \
  "  v_fmac_f32 %[b2], %[dx], %[s2x]\n"                                                               \
  "  v_fmac_f32 %[b1], %[Dx], %[s1x]\n"                                                               \
  "  v_fmac_f32 %[t], " E2X ", %[s2x]\n"                                                              \
  "  v_fmac_f32 %[b2], %[dz], %[s2z]\n"                                                               \
  "  v_fmac_f32 %[b1], %[Dz], %[s1z]\n"                                                               \
  "  v_fmac_f32 %[t], " E2Z ", %[s2z]\n"                                                              \
  "  v_mul_f32 %[b2], %[b2], %[inv]\n"                                                                \
  "  v_mul_f32 %[b1], %[b1], %[inv]\n"                                                                \
  "  v_mul_f32 %[t], %[t], %[inv]\n"                                                                  \
  "  v_min3_f32 %[s2x], %[b1], %[b2], %[t]\n"  /* lo = min(b1, b2, t) */                              \
  "  v_add_f32 %[s2y], %[b1], %[b2]\n"                                                                \
  "  v_max_f32 %[s2y], %[b1], %[s2y]\n"        /* hi = max(b1, b1 + b2) */                            \
  /* divisor == 0 || b1 < 0 || b2 < 0 || t < 0 || b1 > 1 || b1 + b2 > 1 || t > far  ->  miss */       \
  "  v_cmpx_neq_f32 vcc, 0, %[div]\n"                                                                 \
  "  v_cmpx_ngt_f32 vcc, 0, %[s2x]\n"                                                                 \
  "  v_cmpx_nlt_f32 vcc, 1.0, %[s2y]\n"                                                               \
  "  v_cmpx_ngt_f32 vcc, %[t], %[far]\n"    /* EXEC = the lanes that hit */
// the triangle pool at byte offset s86: record k in s[64:75], record k + 1 (when the leaf has one) in s[88:95] + s[76:79]
#define GLOME_PKW_TRI_A GLOME_PKW_TRI_TEST("s" K4, "s" K5, "s" K6, "s" K8, "s" K9, "s" K10, "s" K12, "s" K13, "s" K14)
#define GLOME_PKW_TRI_B GLOME_PKW_TRI_TEST("s" K28, "s" K29, "s" K30, "s" K32, "s" K33, "s" K34, "s" K16, "s" K17, "s" K18)

// what the lanes that hit do (EXEC = those lanes), and what a popped entry's lane mask s[80:81] is filtered with (SCC = any left)
#define GLOME_PKW_UPDATE_1(REC)                                                                       \
  "  v_mov_b32 %[best_t], %[t]\n"                                                                     \
  "  v_mov_b32 %[far], %[t]\n"              /* far = min(far, t) = t: the test just passed t <= far */ \
  "  v_mov_b32 %[best_rec], " REC "\n"
#define GLOME_PKW_UPDATE_2(REC)                                                                       \
  "  s_or_b64 %[occ], %[occ], exec\n"                                                                 \
  "  s_andn2_b64 %[am], %[am], exec\n"      /* an occluded ray is finished; SCC = rays left in this walk's current entry */ \
  "  s_cbranch_scc0 L_leafdone_%=\n"
#define GLOME_PKW_FILTER_1                                                                            \
  "  v_min_f32 %[far], %[far], %[best_t]\n" /* `far` may have shrunk since the push */                \
  "  v_cmp_ngt_f32 vcc, %[near], %[far]\n"                                                            \
  "  s_and_b64 %[am], s[" K20 ":" K21 "], vcc\n"
#define GLOME_PKW_FILTER_2 "  s_andn2_b64 %[am], s[" K20 ":" K21 "], %[occ]\n"

// the whole walk as one statement.  AX / AY / AZ: FWD or BWD per axis; M: 1 or 2.
#define GLOME_PKW_ASM(AX, AY, AZ, M)                                                                                            \
  asm volatile(                                                                                                                 \
      "  s_mov_b32 s" K24 ", m0\n"               /* m0 holds the stack pointer for the whole walk (restored at the end) */           \
      "  s_mov_b32 m0, %[sp]\n"                                                                                                 \
      "  s_mov_b64 s[" K22 ":" K23 "], exec\n"                                                                                            \
      "  s_cmp_eq_u32 %[sp], 0\n"           /* re-entered after a C++ step: the entries come back from the dump block */         \
      "  s_cbranch_scc1 L_fresh_%=\n"                                                                                           \
      "  s_waitcnt vmcnt(0)\n"                                                                                                  \
      "  global_load_dword %[ur], %[dump], off\n"                                                                               \
      "  global_load_dword %[ulo], %[dump], off offset:256\n"                                                                   \
      "  global_load_dword %[uhi], %[dump], off offset:512\n"                                                                   \
      "  s_waitcnt vmcnt(0)\n"                                                                                                  \
      "L_fresh_%=:\n"                                                                                                           \
      "  s_lshl_b32 s" K20 ", %[sp], 8\n"                                                                                            \
      "  v_add_u32 %[av], s" K20 ", %[lds]\n"     /* this lane's slot of the next free entry */                                      \
      "  s_cmp_lg_u32 %[phase], 0\n"                                                                                            \
      "  s_cbranch_scc1 L_pop_%=\n"                                                                                             \
      /* ------------------------------------------------------------ branch steps */                                          \
      "L_node_%=:\n"                                                                                                            \
      "  s_bitcmp1_b32 %[ref], 29\n"                                                                                            \
      "  s_cbranch_scc1 L_leaf_%=\n"                                                                                            \
      "  s_lshl_b32 s" K20 ", %[ref], 4\n"                                                                                           \
      "  s_load_dwordx4 s[" K0 ":" K3 "], %[nodes], s" K20 "\n"                                                                              \
      "  s_waitcnt lgkmcnt(0)\n"                                                                                                \
      "  s_and_b32 s" K21 ", s" K2 ", 3\n"           /* axis; SCC = (axis != 0) */                                                       \
      "  s_cbranch_scc0 L_axX_%=\n"                                                                                             \
      "  s_bitcmp1_b32 s" K21 ", 1\n"                                                                                                \
      "  s_cbranch_scc1 L_axZ_%=\n"                                                                                             \
      GLOME_PKW_AXIS_##AY("Y", "oy", "ry") GLOME_PKW_AXIS_##AX("X", "ox", "rx") GLOME_PKW_AXIS_##AZ("Z", "oz", "rz")            \
      /* ------------------------------------------------------------ a leaf: up to six triangles */                           \
      "L_leaf_%=:\n"                                                                                                            \
      "  s_bfe_u32 s" K25 ", %[ref], 0x3001a\n"  /* item count (bits 28..26) */                                                      \
      "  s_and_b32 s" K27 ", %[ref], 0x3ffffff\n" /* first record */                                                                 \
      "  s_cmp_eq_u32 s" K25 ", 7\n"                                                                                                 \
      "  s_cbranch_scc1 L_big_%=\n"                                                                                             \
      "  s_cmp_eq_u32 s" K25 ", 0\n"                                                                                                 \
      "  s_cbranch_scc1 L_pop_%=\n"                                                                                             \
      "  s_bfm_b32 s" K25 ", s" K25 ", 0\n"           /* `count` ones: shifted out one per triangle */                                    \
      "  s_add_u32 s" K26 ", s" K27 ", %[delta]\n"     /* first primitive */                                                              \
      "  s_mul_i32 s" K26 ", s" K26 ", 48\n"                                                                                              \
      /* two triangles per memory round trip: scalar loads return out of order, so a wait is a wait for all of them --   */       \
      /* the second record's loads go out with the first's                                                                */       \
      "L_tri_%=:\n"                                                                                                             \
      "  s_load_dwordx8 s[" K4 ":" K11 "], %[tris], s" K26 "\n"                                                                               \
      "  s_load_dwordx4 s[" K12 ":" K15 "], %[tris], s" K26 " offset:0x20\n"                                                                   \
      "  s_bitcmp1_b32 s" K25 ", 1\n"            /* a second triangle in this leaf? */                                               \
      "  s_cbranch_scc0 L_one_%=\n"                                                                                             \
      "  s_load_dwordx8 s[" K28 ":" K35 "], %[tris], s" K26 " offset:0x30\n"                                                                   \
      "  s_load_dwordx4 s[" K16 ":" K19 "], %[tris], s" K26 " offset:0x50\n"                                                                   \
      "L_one_%=:\n"                                                                                                             \
      "  s_mov_b64 exec, %[am]\n"           /* only the lanes whose interval reaches this leaf */                               \
      "  s_waitcnt lgkmcnt(0)\n"                                                                                                \
      GLOME_PKW_TRI_A                                                                                                           \
      GLOME_PKW_UPDATE_##M("s" K27)                                                                                               \
      "  s_bitcmp1_b32 s" K25 ", 1\n"                                                                                                \
      "  s_cbranch_scc0 L_leafdone_%=\n"    /* that was the leaf's last */                                                      \
      "  s_add_u32 s" K27 ", s" K27 ", 1\n"                                                                                               \
      "  s_mov_b64 exec, %[am]\n"                                                                                               \
      GLOME_PKW_TRI_B                                                                                                           \
      GLOME_PKW_UPDATE_##M("s" K27)                                                                                               \
      "  s_add_u32 s" K26 ", s" K26 ", 96\n"                                                                                              \
      "  s_add_u32 s" K27 ", s" K27 ", 1\n"                                                                                               \
      "  s_lshr_b32 s" K25 ", s" K25 ", 2\n"          /* SCC = triangles left */                                                          \
      "  s_cbranch_scc1 L_tri_%=\n"                                                                                             \
      "L_leafdone_%=:\n"                                                                                                        \
      "  s_mov_b64 exec, s[" K22 ":" K23 "]\n"                                                                                            \
      /* ------------------------------------------------------------ pop until an entry some lane still wants */              \
      "L_pop_%=:\n"                                                                                                             \
      "  s_cmp_eq_u32 m0, 0\n"                                                                                                  \
      "  s_cbranch_scc1 L_empty_%=\n"                                                                                           \
      "  s_cmp_gt_u32 m0, %[cap]\n"                                                                                             \
      "  s_cbranch_scc1 L_popslow_%=\n"                                                                                         \
      "  s_sub_u32 m0, m0, 1\n"                                                                                                 \
      "  v_add_u32 %[av], 0xffffff00, %[av]\n"                                                                                  \
      "  ds_read_b32 %[near], %[av]\n"                                                                                          \
      "  ds_read_b32 %[far], %[av] offset:%[row1]\n"                                                                            \
      "  v_readlane_b32 %[ref], %[ur], m0\n"                                                                                    \
      "  v_readlane_b32 s" K20 ", %[ulo], m0\n"                                                                                      \
      "  v_readlane_b32 s" K21 ", %[uhi], m0\n"                                                                                      \
      "  s_waitcnt lgkmcnt(0)\n"                                                                                                \
      GLOME_PKW_FILTER_##M                                                                                                      \
      "  s_cbranch_scc0 L_pop_%=\n"                                                                                             \
      "  s_branch L_node_%=\n"                                                                                                  \
      /* ------------------------------------------------------------ exits */                                                 \
      "L_empty_%=:\n"                                                                                                           \
      "  s_mov_b64 %[am], 0\n"                                                                                                  \
      "  s_mov_b32 %[status], 0\n"                                                                                              \
      "  s_branch L_end_%=\n"                                                                                                   \
      "L_slow_%=:\n"                          /* a push that does not fit the LDS part: the C++ step takes this node */         \
      "  s_mov_b32 %[status], 1\n"                                                                                              \
      "  s_branch L_dump_%=\n"                                                                                                  \
      "L_big_%=:\n"                           /* a leaf of more than six items */                                               \
      "  s_mov_b32 %[status], 2\n"                                                                                              \
      "  s_branch L_dump_%=\n"                                                                                                  \
      "L_popslow_%=:\n"                       /* the top entry lives in the overflow columns */                                 \
      "  s_mov_b32 %[status], 3\n"                                                                                              \
      "L_dump_%=:\n"                          /* a C++ step follows: the entries (lane k = entry k) leave the registers */       \
      "  s_mov_b64 exec, s[" K22 ":" K23 "]\n"                                                                                            \
      "  global_store_dword %[dump], %[ur], off\n"                                                                              \
      "  global_store_dword %[dump], %[ulo], off offset:256\n"                                                                  \
      "  global_store_dword %[dump], %[uhi], off offset:512\n"                                                                  \
      "  s_waitcnt vmcnt(0)\n"                                                                                                  \
      "L_end_%=:\n"                                                                                                             \
      "  s_mov_b32 %[sp], m0\n"                                                                                                 \
      "  s_mov_b32 m0, s" K24 "\n"                                                                                                   \
      : [ref] "+s"(ref), [am] "+s"(am), [sp] "+s"(sp), [near] "+v"(nearv), [far] "+v"(farv), [best_t] "+v"(best_t), [best_rec] "+v"(best_rec), [ur] "=&v"(ur),    \
        [ulo] "=&v"(ulo), [uhi] "=&v"(uhi), [occ] "+s"(occm), [status] "=s"(status), [av] "=&v"(av), [t1] "=&v"(t1), [t2] "=&v"(t2), [Dx] "=&v"(Dx), [Dy] "=&v"(Dy),  \
        [Dz] "=&v"(Dz), [s2x] "=&v"(s2x), [s2y] "=&v"(s2y), [s2z] "=&v"(s2z), [s1x] "=&v"(s1x), [s1y] "=&v"(s1y), [s1z] "=&v"(s1z), [div] "=&v"(dv), [inv] "=&v"(inv),   \
        [b1] "=&v"(b1), [b2] "=&v"(b2), [t] "=&v"(tt)                                                                                                                 \
      : [nodes] "s"(nodes), [tris] "s"(tris), [delta] "s"(delta), [cap] "s"((uint32_t)CAP), [phase] "s"(phase), [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z),         \
        [rx] "v"(rcp.x), [ry] "v"(rcp.y), [rz] "v"(rcp.z), [dx] "v"(d.x), [dy] "v"(d.y), [dz] "v"(d.z), [lds] "v"(lds_row), [row1] "n"(CAP * 256), [dump] "v"(dump)                   \
      : "s" K0, "s" K1, "s" K2, "s" K3, "s" K4, "s" K5, "s" K6, "s" K7, "s" K8, "s" K9, "s" K10, "s" K11, "s" K12, "s" K13, "s" K14, "s" K15, "s" K16, "s" K17, "s" K18, "s" K19, "s" K20, "s" K21, "s" K22, \
        "s" K23, "s" K24, "s" K25, "s" K26, "s" K27, "s" K28, "s" K29, "s" K30, "s" K31, "s" K32, "s" K33, "s" K34, "s" K35, "vcc", "scc", "memory")

// MODE 1 = closest hit with ordered early-out, MODE 2 = any hit.  XF / YF / ZF: the rays of the packet run towards +x / +y / +z.
// CAP: entries of the LDS part of the stack (the far row lies CAP * 256 bytes after the near row).
// phase 0: go on from `ref` with the lanes `am`; phase 1: pop first.  Returns a PKW_* status.
template <int MODE, bool XF, bool YF, bool ZF, int CAP>
__device__ __forceinline__ int bih_walk_asm(const F4* nodes, const F4* tris, uint32_t delta, int phase, uint32_t& ref, LaneMask& am, int& sp, float& nearv, float& farv,
                                            float& best_t, uint32_t& best_rec, LaneMask& occm, V3 o, V3 rcp, V3 d, uint32_t lds_row, uint32_t* dump) {
  static_assert(MODE == 1 || MODE == 2, "the hand-written walk covers the production traversals only");
  int status;
  uint32_t av, ur, ulo, uhi;  // ur / ulo / uhi: the wave-uniform part of the stack, entry k in lane k -- alive inside the asm block only
  float t1, t2, Dx, Dy, Dz, s2x, s2y, s2z, s1x, s1y, s1z, dv, inv, b1, b2, tt;
  // wave-uniform by construction; readfirstlane pins them to scalar registers where the compiler cannot see that
  ref = uni(ref); am = uni(am); sp = (int)uni((uint32_t)sp); occm = uni(occm); phase = (int)uni((uint32_t)phase); delta = uni(delta);
  nodes = (const F4*)(uintptr_t)uni((LaneMask)(uintptr_t)nodes); tris = (const F4*)(uintptr_t)uni((LaneMask)(uintptr_t)tris);
#define GLOME_PKW_BY_OCTANT(M)                                                    \
  if constexpr (XF && YF && ZF) GLOME_PKW_ASM(FWD, FWD, FWD, M);                  \
  else if constexpr (!XF && YF && ZF) GLOME_PKW_ASM(BWD, FWD, FWD, M);            \
  else if constexpr (XF && !YF && ZF) GLOME_PKW_ASM(FWD, BWD, FWD, M);            \
  else if constexpr (!XF && !YF && ZF) GLOME_PKW_ASM(BWD, BWD, FWD, M);           \
  else if constexpr (XF && YF && !ZF) GLOME_PKW_ASM(FWD, FWD, BWD, M);            \
  else if constexpr (!XF && YF && !ZF) GLOME_PKW_ASM(BWD, FWD, BWD, M);           \
  else if constexpr (XF && !YF && !ZF) GLOME_PKW_ASM(FWD, BWD, BWD, M);           \
  else GLOME_PKW_ASM(BWD, BWD, BWD, M)
  if constexpr (MODE == 1) { GLOME_PKW_BY_OCTANT(1); } else { GLOME_PKW_BY_OCTANT(2); }
#undef GLOME_PKW_BY_OCTANT
  return status;
}

}  // namespace glome
#endif
