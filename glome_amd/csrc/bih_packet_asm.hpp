// bih_packet_asm.hpp -- the packet walk of a triangle BIH, hand-written for gfx950 (device code only).
//
// rt_device.hpp's bih_tri_packet is the reference implementation of one wave walking `rayint_bih` / `shadow_bih`
// (Bih.hs:332-368, 510-544) once for its 64 rays.  This file is the same walk written out, one instance per octant (the
// direction signs of a packet are fixed for the whole walk: no direction test, no operand swap) and per mode.
//
// What bounds it (round 3; DESIGN.md section 4.1a has the measurements): a walk of the flagship frame is 92 branch steps, 17 leaf
// visits and 25 pops, about 4,700 instructions, and the frame time follows the INSTRUCTION COUNT one to one -- 736 more per work
// item cost 6.5 % whether they are scalar, branch or vector instructions, while taking the next node's fetch out of the dependent
// chain is worth 1.5 % at six waves per SIMD (9 % at one).  A SIMD issues 0.24 scalar instructions per cycle, 0.24 vector
// instructions with a scalar operand (what a node's planes and a triangle's words are), 0.19-0.23 packed ones
// (tools/probe/valu_rate.hip); the kernel keeps the scalar port 0.57 busy (0.70 on the 1M-triangle tree), the vector port less, and
// its waves wait for 57 % of their cycles: six waves per SIMD that each execute their own stream serially.  So this walk is
// written for the FEWEST INSTRUCTIONS per step, scalar-type ones first (the busier port):
//
//   lanes        the set of lanes whose interval reaches the current node lives in EXEC, not in a scalar pair: a vote is a
//                v_cmp into VCC or a v_cmpx (which narrows EXEC itself) followed by s_cbranch_vccz / s_cbranch_execz -- no
//                s_and, no s_cmp, no mask moves.  Lanes outside EXEC keep stale intervals that nobody reads.
//   branch step  9-12 scalar-type instructions (round 2: 21).  The walk reads its own copy of the nodes (DScene::pknodes,
//                flatten.hpp): a branch child's reference is its byte offset with the child's AXIS in the two low bits, a leaf
//                child's is the byte offset of its first pair record with both low bits set, so two bit tests pick the next
//                piece of code and nothing is shifted or masked.  As soon as a node has arrived the NEAR child's node -- the
//                next step seven times in ten -- is asked for into the other of two register sets (A / B), through a buffer
//                descriptor over the pool: a leaf reference reads something harmless or, out of range, nothing, so the
//                prefetch needs no test.  The two plane distances are one packed pair: (lsplit, rsplit) sit in an aligned
//                scalar pair, v_pk_add_f32 / v_pk_mul_f32 give ((l - o) * r, (r - o) * r) -- the same IEEE operations per
//                element.  The far child is pushed as soon as some lane reaches it (before the near vote narrows EXEC) and
//                taken straight back in the rare case that no lane enters the near child.  The stack pointer lives in m0
//                for the whole walk (v_writelane / v_readlane take it from there; v_lshl_add_u32 makes the LDS address from
//                it, so no per-lane address register can drift under a partial EXEC).
//                At most one prefetch is ever in flight, and a fetch that was not prefetched (a far child entered alone, a
//                popped entry, the root) is only issued behind an s_waitcnt, so two loads never race for one register set.
//   triangles    TWO per test, any number per leaf: a leaf's triangles come as 80-byte pair records (DScene::tripairs,
//                flatten.hpp emit_pairs: every component of p1, e1, e2 of two triangles as (A, B) in an aligned scalar pair,
//                then the number of triangles left in the leaf; five of six leaves of the flagship tree hold exactly two)
//                and the Moeller-Trumbore arithmetic runs on (A, B) register pairs with v_pk_add / v_pk_mul / v_pk_fma_f32
//                -- per element the same IEEE operations in the same order as the compiler's tri_test, so results stay
//                bit-identical to every other kernel instance.  The ray's origin and direction are read from three aligned
//                register pairs ((ox, oy) (oz, dx) (dy, dz)) through op_sel, which broadcasts either half.  The rejection
//                tests are two chains of v_cmpx, A's first (its update moves `far`, which B's last test reads: `nearest`,
//                ties -> later item); a leaf's odd last triangle is a pair whose B half is computed and ignored.  A hit
//                records the triangle's record index, which the pair record carries.
//   pop          7 scalar-type.
//
// What it declines goes back to the C++ loop for one step (status codes below): pushes and pops beyond the LDS part of the
// stack (global overflow columns).  The wave-uniform part of the stack (reference and lane mask of entry k in lane k of three
// vector registers) never leaves the asm block in registers: before a C++ step it is written to the wave's dump block in
// global memory, on re-entry (sp > 0) read back -- a register whose inactive lanes carry data must not be visible to the
// compiler, which may spill or copy it under a partial EXEC mask.  MODE 1 (ordered early-out closest hit) and MODE 2 (any
// hit) only; the faithful / counting variants stay in C++.
//
// Invariant used (MODE 1): far <= best_t on the current path at all times (the root interval is clipped with the running
// best, children only shrink it, a pop clips with best_t, an accepted hit sets both), so a triangle hit within [.., far]
// always replaces the running best: `!(best_t < t)` (nearest: ties -> later item, Solid.hs:37-44) needs no test.
//
// EXEC must be all ones at entry (every caller is wave-uniform: one wave per workgroup, 64 threads): the dump block's loads
// and stores and lane k's stack entry rely on it; the block forces it around the dump traffic and restores the entry mask.
// Scalar registers K0..K39 and vector registers T0..T15 (tables below), vcc, scc and m0 (saved and restored) are scratch,
// named in the clobber list.  No scalar load is in flight when the block ends.  `status` is an EARLY-CLOBBER output: the block writes
// it before its last reads of the scalar inputs (the dump block's base, since round 4 a scalar pair), and without the `&` the compiler
// may give it an input's register -- it did: the dump stores of a step handed to C++ went to (base & ~0xffffffff) | status.
#pragma once
#if defined(__HIPCC__)

namespace glome {

// The block's scratch scalar registers: 40 consecutive ones from GLOME_PKW_BASE (a multiple of 4: the wide loads want aligned
// destinations), K0 .. K39 below.
//   K0..K3 node set A, K4..K7 node set B (lsplit, rsplit, left, right) | K8..K25 a pair record (p1x p1y p1z e1x e1y e1z e2x e2y
//   e2z, each as (A, B)), K26 the triangles left in its leaf, K27 the first one's record index | K28 K29 a popped entry's mask, v_cmpx's other
//   destination | K30 K31 EXEC at entry | K32 the caller's m0 | K33 the second triangle's record | K34 K35 a leaf's lanes |
//   K36..K39 buffer descriptor over the node pool
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
#define K36 "72"
#define K37 "73"
#define K38 "74"
#define K39 "75"
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
#define K36 "76"
#define K37 "77"
#define K38 "78"
#define K39 "79"
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
#define K36 "80"
#define K37 "81"
#define K38 "82"
#define K39 "83"
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
#define K36 "84"
#define K37 "85"
#define K38 "86"
#define K39 "87"
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
#define K36 "96"
#define K37 "97"
#define K38 "98"
#define K39 "99"
#else
#error "GLOME_PKW_BASE: 36, 40, 44, 48 or 60"
#endif

// The block's scratch vector registers: 16 consecutive ones from GLOME_PKW_VBASE (even: they are used as aligned pairs),
// T0 .. T15.  Named, not allocated by the compiler, because the packed arithmetic needs both the pair and its halves.
//   branch step: (T0, T1) the two plane distances | leaf: (T0 T1) (T2 T3) (T4 T5) D = o - p1, then b2, t, 1 / divisor;
//   (T6 T7) (T8 T9) (T10 T11) s1 = dir x e2, then s2 = D x e1, then min / max of the barycentrics; (T12 T13) divisor; (T14 T15) b1
#ifndef GLOME_PKW_VBASE
#define GLOME_PKW_VBASE 64
#endif
#if GLOME_PKW_VBASE == 40
#define T0 "40"
#define T1 "41"
#define T2 "42"
#define T3 "43"
#define T4 "44"
#define T5 "45"
#define T6 "46"
#define T7 "47"
#define T8 "48"
#define T9 "49"
#define T10 "50"
#define T11 "51"
#define T12 "52"
#define T13 "53"
#define T14 "54"
#define T15 "55"
#elif GLOME_PKW_VBASE == 48
#define T0 "48"
#define T1 "49"
#define T2 "50"
#define T3 "51"
#define T4 "52"
#define T5 "53"
#define T6 "54"
#define T7 "55"
#define T8 "56"
#define T9 "57"
#define T10 "58"
#define T11 "59"
#define T12 "60"
#define T13 "61"
#define T14 "62"
#define T15 "63"
#elif GLOME_PKW_VBASE == 56
#define T0 "56"
#define T1 "57"
#define T2 "58"
#define T3 "59"
#define T4 "60"
#define T5 "61"
#define T6 "62"
#define T7 "63"
#define T8 "64"
#define T9 "65"
#define T10 "66"
#define T11 "67"
#define T12 "68"
#define T13 "69"
#define T14 "70"
#define T15 "71"
#elif GLOME_PKW_VBASE == 64
#define T0 "64"
#define T1 "65"
#define T2 "66"
#define T3 "67"
#define T4 "68"
#define T5 "69"
#define T6 "70"
#define T7 "71"
#define T8 "72"
#define T9 "73"
#define T10 "74"
#define T11 "75"
#define T12 "76"
#define T13 "77"
#define T14 "78"
#define T15 "79"
#else
#error "GLOME_PKW_VBASE: 40, 48, 56 or 64"
#endif


enum : int { PKW_DONE = 0, PKW_PUSH_OVERFLOW = 1, PKW_POP_OVERFLOW = 3 };
constexpr uint32_t PKREF_LEAF = 3u;  // low bits of a reference in the walk's own form: 0 / 1 / 2 = a branch splitting x / y / z, 3 = a leaf

// where to go with the reference in %[ref]: 00 / 01 / 10 = a branch along x / y / z whose node is arriving in set S, 11 = a leaf
#define GLOME_PKW_DISPATCH(S)                                                                           \
  "  s_bitcmp1_b32 %[ref], 1\n"                                                                         \
  "  s_cbranch_scc1 L_zl" S "_%=\n"                                                                     \
  "  s_bitcmp1_b32 %[ref], 0\n"                                                                         \
  "  s_cbranch_scc1 L_st" S "Y_%=\n"                                                                    \
  "  s_branch L_st" S "X_%=\n"
#define GLOME_PKW_DISPATCH_FALL(S) /* ... and the X piece follows */                                     \
  "  s_bitcmp1_b32 %[ref], 1\n"                                                                         \
  "  s_cbranch_scc1 L_zl" S "_%=\n"                                                                     \
  "  s_bitcmp1_b32 %[ref], 0\n"                                                                         \
  "  s_cbranch_scc1 L_st" S "Y_%=\n"
#define GLOME_PKW_DISPATCH_TAIL(S)                                                                      \
  "L_zl" S "_%=:\n"                                                                                     \
  "  s_bitcmp1_b32 %[ref], 0\n"                                                                         \
  "  s_cbranch_scc1 L_leaf_%=\n"                                                                        \
  "  s_branch L_st" S "Z_%=\n"


// Sensitivity experiments (measurement builds only, tools/build_variants.py): extra work of one kind per branch step or per
// pair test, results unchanged -- what the frame time answers to is what bounds the walk.
#if defined(GLOME_PKW_EXP_SALU)
#define PKW_EXP_STEP "  s_mov_b32 s" K27 ", s" K27 "\n  s_mov_b32 s" K27 ", s" K27 "\n  s_mov_b32 s" K27 ", s" K27 "\n  s_mov_b32 s" K27 ", s" K27 "\n  s_mov_b32 s" K27 ", s" K27 "\n  s_mov_b32 s" K27 ", s" K27 "\n  s_mov_b32 s" K27 ", s" K27 "\n  s_mov_b32 s" K27 ", s" K27 "\n"
#elif defined(GLOME_PKW_EXP_BRANCH)
#define PKW_EXP_STEP_(TG) "  s_branch L_x1" TG "_%=\nL_x1" TG "_%=:\n  s_branch L_x2" TG "_%=\nL_x2" TG "_%=:\n  s_branch L_x3" TG "_%=\nL_x3" TG "_%=:\n  s_branch L_x4" TG "_%=\nL_x4" TG "_%=:\n"
#elif defined(GLOME_PKW_EXP_VALU)
#define PKW_EXP_STEP "  v_mov_b32 v" T3 ", v" T3 "\n  v_mov_b32 v" T3 ", v" T3 "\n  v_mov_b32 v" T3 ", v" T3 "\n  v_mov_b32 v" T3 ", v" T3 "\n  v_mov_b32 v" T3 ", v" T3 "\n  v_mov_b32 v" T3 ", v" T3 "\n  v_mov_b32 v" T3 ", v" T3 "\n  v_mov_b32 v" T3 ", v" T3 "\n"
#else
#define PKW_EXP_STEP ""
#endif
#ifndef PKW_EXP_STEP_
#define PKW_EXP_STEP_(TG) PKW_EXP_STEP
#endif

// One branch step, for a node that is in flight into (or already in) register set SET ("A" = s[K0:K3], "B" = s[K4:K7]) and
// splits along TAG's axis; EXEC = the lanes whose interval reaches it.  PL: the set's plane pair; NC / FC: its near and far
// child references (by the packet's direction on this axis); OTH: the other set, where the near child's node is prefetched; ON:
// the other set's name.  OP / RP / H: how the packed subtract / multiply pick this axis' origin and reciprocal out of their
// pairs (P0 = (ox, oy), P1 = (oz, dx); R0 = (rx, ry), R1 = (rz, .)): the pair operand and the op_sel bit that broadcasts its
// low or high half.  TN / TF: which half of (T0, T1) = ((lsplit - o) * r, (rsplit - o) * r) ends the near child's interval
// and which starts the far child's.
#define GLOME_PKW_STEP(SET, TAG, PL, NC, FC, OTH, ON, OP, RP, H, TN, TF)                                 \
  "L_st" SET TAG "_%=:\n"                                                                               \
  "  s_waitcnt lgkmcnt(0)\n"                /* the node is here */                                      \
  PKW_PREFETCH(OTH, NC)                                                                                 \
  PKW_EXP_STEP_(SET TAG)                                                                                \
  "  v_pk_add_f32 v[" T0 ":" T1 "], " PL ", %[" OP "] op_sel:[0," H "] op_sel_hi:[1," H "] neg_lo:[0,1] neg_hi:[0,1]\n" \
  "  v_pk_mul_f32 v[" T0 ":" T1 "], v[" T0 ":" T1 "], %[" RP "] op_sel:[0," H "] op_sel_hi:[1," H "]\n" \
  "  v_cmp_lt_f32 vcc, " TF ", %[far]\n"    /* lanes that reach the far child */                        \
  "  s_cbranch_vccz L_nf" SET TAG "_%=\n"                                                               \
  "  s_cmp_ge_u32 m0, %[cap]\n"                                                                         \
  "  s_cbranch_scc1 L_slow_%=\n"                                                                        \
  "  v_max_f32 " TF ", " TF ", %[near]\n"   /* the far child's interval starts here */                  \
  "  v_writelane_b32 %[ur], " FC ", m0\n"   /* the uniform part of the entry: lane `sp` of ur / ulo / uhi */ \
  "  v_writelane_b32 %[ulo], vcc_lo, m0\n"                                                              \
  "  v_writelane_b32 %[uhi], vcc_hi, m0\n"                                                              \
  "  v_lshl_add_u32 v" T2 ", m0, 8, %[lds]\n"                                                           \
  "  ds_write_b32 v" T2 ", " TF "\n"        /* this lane's (near, far) of the far child */              \
  "  ds_write_b32 v" T2 ", %[far] offset:%[row1]\n"                                                     \
  "  s_add_u32 m0, m0, 1\n"                                                                             \
  "L_nf" SET TAG "_%=:\n"                                                                               \
  "  v_cmpx_lt_f32_e64 s[" K28 ":" K29 "], %[near], " TN "\n"  /* EXEC = the lanes that reach the near child */ \
  "  s_cbranch_execz L_n1" SET TAG "_%=\n"                                                              \
  "  v_min_f32 %[far], " TN ", %[far]\n"                                                                \
  "  s_mov_b32 %[ref], " NC "\n"                                                                        \
  PKW_LATE_FETCH(OTH, NC)                                                                               \
  GLOME_PKW_DISPATCH(ON)                                                                                \
  "L_n1" SET TAG "_%=:\n"                   /* nobody enters the near child */                          \
  "  s_cbranch_vccz L_pop_%=\n"             /* nor the far one */                                       \
  "  s_sub_u32 m0, m0, 1\n"                 /* the far child alone: it was pushed a moment ago, take it back */ \
  "  s_mov_b64 exec, vcc\n"                                                                             \
  "  v_mov_b32 %[near], " TF "\n"                                                                       \
  "  s_mov_b32 %[ref], " FC "\n"                                                                        \
  "  s_branch L_node_%=\n"

// rays running towards +axis take the left child first (it ends at lsplit -> T0), the others the right one (rsplit -> T1)
#define GLOME_PKW_STEP_FWD(SET, TAG, PL, L, R, OTH, ON, OP, RP, H) GLOME_PKW_STEP(SET, TAG, PL, L, R, OTH, ON, OP, RP, H, "v" T0, "v" T1)
#define GLOME_PKW_STEP_BWD(SET, TAG, PL, L, R, OTH, ON, OP, RP, H) GLOME_PKW_STEP(SET, TAG, PL, R, L, OTH, ON, OP, RP, H, "v" T1, "v" T0)
#define GLOME_PKW_STEPS(SET, PL, L, R, OTH, ON, AX, AY, AZ)                                             \
  GLOME_PKW_STEP_##AX(SET, "X", PL, L, R, OTH, ON, "P0", "R0", "0")                                     \
  GLOME_PKW_STEP_##AY(SET, "Y", PL, L, R, OTH, ON, "P0", "R0", "1")                                     \
  GLOME_PKW_STEP_##AZ(SET, "Z", PL, L, R, OTH, ON, "P1", "R1", "0")                                     \
  GLOME_PKW_DISPATCH_TAIL(SET)
#if defined(GLOME_PKW_EXP_NOPF)  // the near child's fetch at the END of the step: the same instructions, nothing overlapped
#define PKW_PREFETCH(OTH, NC) ""
#define PKW_LATE_FETCH(OTH, NC) "  s_buffer_load_dwordx4 " OTH ", s[" K36 ":" K39 "], " NC "\n"
#else
#define PKW_PREFETCH(OTH, NC) "  s_buffer_load_dwordx4 " OTH ", s[" K36 ":" K39 "], " NC "\n"  /* the near child's node, under this step's work */
#define PKW_LATE_FETCH(OTH, NC) ""
#endif
#define PKW_SETA "s[" K0 ":" K3 "]"
#define PKW_SETB "s[" K4 ":" K7 "]"

// The arithmetic of two triangle tests on (A, B) pairs.  tri_test's (Triangle.hs:45-73) operation for operation as hipcc
// contracts it under -ffp-contract=on: cross(a, b).y = fma(a.z, b.x, -(a.x * b.z)) and cyclic; dot(a, b) = fma(a.z, b.z,
// fma(a.x, b.x, a.y * b.y)).  Scalar pairs: p1 = K8..K13, e1 = K14..K19, e2 = K20..K25 (x, y, z; each (A, B)).  Ray: ox = P0.lo,
// oy = P0.hi, oz = P1.lo, dx = P1.hi, dy = P2.lo, dz = P2.hi.  BL / BH: a pair operand's low / high half for both elements.
// Leaves (A, B) of: divisor T12 T13, min(b1, b2, t) T6 T7, max(b1, b1 + b2) T8 T9, t T2 T3.
#define PKW_BL2 " op_sel:[0,0] op_sel_hi:[0,1]"
#define PKW_BH2 " op_sel:[1,0] op_sel_hi:[1,1]"
#define PKW_BL3 " op_sel:[0,0,0] op_sel_hi:[0,1,1]"
#define PKW_BH3 " op_sel:[1,0,0] op_sel_hi:[1,1,1]"
#define PKW_NEG0 " neg_lo:[1,0] neg_hi:[1,0]"
#define PKW_NEG1 " neg_lo:[0,1] neg_hi:[0,1]"
#define PKW_DX "v[" T0 ":" T1 "]"
#define PKW_DY "v[" T2 ":" T3 "]"
#define PKW_DZ "v[" T4 ":" T5 "]"
#define PKW_AX "v[" T6 ":" T7 "]"
#define PKW_AY "v[" T8 ":" T9 "]"
#define PKW_AZ "v[" T10 ":" T11 "]"
#define PKW_DIV "v[" T12 ":" T13 "]"
#define PKW_B1 "v[" T14 ":" T15 "]"
#define PKW_B2 PKW_DX
#define PKW_TT PKW_DY
#define PKW_INV PKW_DZ
#define PKW_P1X "s[" K8 ":" K9 "]"
#define PKW_P1Y "s[" K10 ":" K11 "]"
#define PKW_P1Z "s[" K12 ":" K13 "]"
#define PKW_E1X "s[" K14 ":" K15 "]"
#define PKW_E1Y "s[" K16 ":" K17 "]"
#define PKW_E1Z "s[" K18 ":" K19 "]"
#define PKW_E2X "s[" K20 ":" K21 "]"
#define PKW_E2Y "s[" K22 ":" K23 "]"
#define PKW_E2Z "s[" K24 ":" K25 "]"
#define GLOME_PKW_PAIR_ARITH                                                                                          \
  "  v_pk_add_f32 " PKW_DX ", %[P0], " PKW_P1X PKW_BL2 PKW_NEG1 "\n"   /* D = o - p1 */                                \
  "  v_pk_add_f32 " PKW_DY ", %[P0], " PKW_P1Y PKW_BH2 PKW_NEG1 "\n"                                                   \
  "  v_pk_add_f32 " PKW_DZ ", %[P1], " PKW_P1Z PKW_BL2 PKW_NEG1 "\n"                                                   \
  "  v_pk_mul_f32 " PKW_AY ", %[P1], " PKW_E2Z PKW_BH2 PKW_NEG0 "\n"   /* s1 = dir x e2: s1y = fma(dz, e2x, -(dx * e2z)) */ \
  "  v_pk_mul_f32 " PKW_AX ", %[P2], " PKW_E2Y PKW_BH2 PKW_NEG0 "\n"   /* s1x = fma(dy, e2z, -(dz * e2y)) */            \
  "  v_pk_mul_f32 " PKW_AZ ", %[P2], " PKW_E2X PKW_BL2 PKW_NEG0 "\n"   /* s1z = fma(dx, e2y, -(dy * e2x)) */            \
  "  v_pk_fma_f32 " PKW_AY ", %[P2], " PKW_E2X ", " PKW_AY PKW_BH3 "\n"                                                \
  "  v_pk_fma_f32 " PKW_AX ", %[P2], " PKW_E2Z ", " PKW_AX PKW_BL3 "\n"                                                \
  "  v_pk_fma_f32 " PKW_AZ ", %[P1], " PKW_E2Y ", " PKW_AZ PKW_BH3 "\n"                                                \
  "  v_pk_mul_f32 " PKW_DIV ", " PKW_AY ", " PKW_E1Y "\n"              /* divisor = s1 . e1 */                          \
  "  v_pk_fma_f32 " PKW_DIV ", " PKW_AX ", " PKW_E1X ", " PKW_DIV "\n"                                                 \
  "  v_pk_fma_f32 " PKW_DIV ", " PKW_AZ ", " PKW_E1Z ", " PKW_DIV "\n"                                                 \
  "  v_pk_mul_f32 " PKW_B1 ", " PKW_DY ", " PKW_AY "\n"                /* D . s1 */                                     \
  "  v_pk_fma_f32 " PKW_B1 ", " PKW_DX ", " PKW_AX ", " PKW_B1 "\n"                                                    \
  "  v_pk_fma_f32 " PKW_B1 ", " PKW_DZ ", " PKW_AZ ", " PKW_B1 "\n"                                                    \
  "  v_pk_mul_f32 " PKW_AY ", " PKW_DX ", " PKW_E1Z PKW_NEG0 "\n"      /* s2 = D x e1 (over s1) */                      \
  "  v_pk_mul_f32 " PKW_AX ", " PKW_DZ ", " PKW_E1Y PKW_NEG0 "\n"                                                      \
  "  v_pk_mul_f32 " PKW_AZ ", " PKW_DY ", " PKW_E1X PKW_NEG0 "\n"                                                      \
  "  v_pk_fma_f32 " PKW_AY ", " PKW_DZ ", " PKW_E1X ", " PKW_AY "\n"                                                   \
  "  v_pk_fma_f32 " PKW_AX ", " PKW_DY ", " PKW_E1Z ", " PKW_AX "\n"                                                   \
  "  v_pk_fma_f32 " PKW_AZ ", " PKW_DX ", " PKW_E1Y ", " PKW_AZ "\n"                                                   \
  "  v_rcp_f32 v" T4 ", v" T12 "\n"                                    /* 1 / divisor (over D.z, dead now) */           \
  "  v_rcp_f32 v" T5 ", v" T13 "\n"                                                                                    \
  "  v_pk_mul_f32 " PKW_B2 ", %[P2], " PKW_AY PKW_BL2 "\n"             /* dir . s2 (over D.x) */                        \
  "  v_pk_mul_f32 " PKW_TT ", " PKW_AY ", " PKW_E2Y "\n"               /* e2 . s2 (over D.y) */                         \
  "  v_pk_fma_f32 " PKW_B2 ", %[P1], " PKW_AX ", " PKW_B2 PKW_BH3 "\n"                                                 \
  "  v_pk_fma_f32 " PKW_TT ", " PKW_AX ", " PKW_E2X ", " PKW_TT "\n"                                                   \
  "  v_pk_fma_f32 " PKW_B2 ", %[P2], " PKW_AZ ", " PKW_B2 PKW_BH3 "\n"                                                 \
  "  v_pk_fma_f32 " PKW_TT ", " PKW_AZ ", " PKW_E2Z ", " PKW_TT "\n"                                                   \
  "  v_pk_mul_f32 " PKW_B2 ", " PKW_B2 ", " PKW_INV "\n"                                                               \
  "  v_pk_mul_f32 " PKW_B1 ", " PKW_B1 ", " PKW_INV "\n"                                                               \
  "  v_pk_mul_f32 " PKW_TT ", " PKW_TT ", " PKW_INV "\n"                                                               \
  "  v_min3_f32 v" T6 ", v" T14 ", v" T0 ", v" T2 "\n"                 /* lo = min(b1, b2, t) */                        \
  "  v_min3_f32 v" T7 ", v" T15 ", v" T1 ", v" T3 "\n"                                                                 \
  "  v_pk_add_f32 " PKW_AY ", " PKW_B1 ", " PKW_B2 "\n"                                                                \
  "  v_max_f32 v" T8 ", v" T14 ", v" T8 "\n"                           /* hi = max(b1, b1 + b2) */                      \
  "  v_max_f32 v" T9 ", v" T15 ", v" T9 "\n"
// divisor == 0 || b1 < 0 || b2 < 0 || t < 0 || b1 > 1 || b1 + b2 > 1 || t > far  ->  miss.  Entered with EXEC = the lanes whose
// interval reaches the leaf; leaves EXEC = the lanes that hit.
#define GLOME_PKW_CHAIN(DIV, LO, HI, TT)                                                              \
  "  v_cmpx_neq_f32 vcc, 0, v" DIV "\n"                                                               \
  "  v_cmpx_ngt_f32 vcc, 0, v" LO "\n"                                                                \
  "  v_cmpx_nlt_f32 vcc, 1.0, v" HI "\n"                                                              \
  "  v_cmpx_ngt_f32 vcc, v" TT ", %[far]\n"
#define GLOME_PKW_CHAIN_A GLOME_PKW_CHAIN(T12, T6, T8, T2)
#define GLOME_PKW_CHAIN_B GLOME_PKW_CHAIN(T13, T7, T9, T3)

// what the lanes that hit do (EXEC = those lanes; TT = the half that holds their t, REC = the scalar register that names the
// triangle: its pair record's byte offset).  MODE 2 goes on to the pops when no lane of the leaf is left.
#define GLOME_PKW_UPDATE_1(TT, REC)                                                                   \
  "  v_mov_b32 %[best_t], v" TT "\n"                                                                  \
  "  v_mov_b32 %[far], v" TT "\n"           /* far = min(far, t) = t: the test just passed t <= far */ \
  "  v_mov_b32 %[best_rec], " REC "\n"
#define GLOME_PKW_UPDATE_2(TT, REC)                                                                   \
  "  s_or_b64 %[occ], %[occ], exec\n"                                                                 \
  "  s_andn2_b64 s[" K34 ":" K35 "], s[" K34 ":" K35 "], exec\n"   /* an occluded ray is finished; SCC = rays left in this leaf */ \
  "  s_cbranch_scc0 L_pop_%=\n"
// a popped entry: EXEC = its lanes (s[K28:K29]) that still want it -- MODE 1: whose interval has not been closed by a nearer
// hit since the push (after the intervals are read back), MODE 2: that are not occluded yet (before)
#define GLOME_PKW_POPMASK_1 "  s_mov_b64 exec, s[" K28 ":" K29 "]\n"
#define GLOME_PKW_POPMASK_2 "  s_andn2_b64 exec, s[" K28 ":" K29 "], %[occ]\n  s_cbranch_execz L_pop_%=\n"
#define GLOME_PKW_FILTER_1                                                                            \
  "  v_min_f32 %[far], %[far], %[best_t]\n" /* `far` may have shrunk since the push */                \
  "  v_cmpx_ngt_f32_e64 s[" K28 ":" K29 "], %[near], %[far]\n"                                        \
  "  s_cbranch_execz L_pop_%=\n"
#define GLOME_PKW_FILTER_2 ""

// the whole walk as one statement.  AX / AY / AZ: FWD or BWD per axis; M: 1 or 2.
#define GLOME_PKW_ASM(AX, AY, AZ, M)                                                                                            \
  asm volatile(                                                                                                                 \
      "  s_mov_b32 s" K32 ", m0\n"          /* m0 holds the stack pointer for the whole walk (restored at the end) */           \
      "  s_mov_b32 m0, %[sp]\n"                                                                                                 \
      "  s_mov_b64 s[" K30 ":" K31 "], exec\n"                                                                                  \
      "  s_mov_b32 s" K36 ", %[nodes_lo]\n" /* a raw buffer over the node pool: a fetch beyond it returns zeros */              \
      "  s_and_b32 s" K37 ", %[nodes_hi], 0xffff\n"                                                                             \
      "  s_mov_b32 s" K38 ", %[nbytes]\n"                                                                                       \
      "  s_mov_b32 s" K39 ", 0x20000\n"                                                                                          \
      "  s_cmp_eq_u32 %[sp], 0\n"           /* re-entered after a C++ step: the entries come back from the dump block */         \
      "  s_cbranch_scc1 L_fresh_%=\n"                                                                                           \
      "  s_mov_b64 exec, -1\n"              /* entry k lives in lane k: every lane takes part, whatever the caller's mask */     \
      "  s_waitcnt vmcnt(0)\n"                                                                                                  \
      "  v_mbcnt_lo_u32_b32 v" T2 ", -1, 0\n" /* the dump block is wave-uniform: lane l's words at 4 l, 256 + 4 l, 512 + 4 l */   \
      "  v_mbcnt_hi_u32_b32 v" T2 ", -1, v" T2 "\n"                                                                             \
      "  v_lshlrev_b32 v" T2 ", 2, v" T2 "\n"                                                                                   \
      "  global_load_dword %[ur], v" T2 ", %[dump]\n"                                                                           \
      "  global_load_dword %[ulo], v" T2 ", %[dump] offset:256\n"                                                               \
      "  global_load_dword %[uhi], v" T2 ", %[dump] offset:512\n"                                                               \
      "  s_waitcnt vmcnt(0)\n"                                                                                                  \
      "L_fresh_%=:\n"                                                                                                           \
      "  s_cmp_lg_u32 %[phase], 0\n"                                                                                            \
      "  s_cbranch_scc1 L_pop_%=\n"                                                                                             \
      "  s_mov_b64 exec, %[am]\n"           /* from here on EXEC is the set of lanes in the current node */                     \
      /* ------------------------------------------------------------ a node that nobody has asked for yet: into set A */       \
      "L_node_%=:\n"                                                                                                            \
      "  s_waitcnt lgkmcnt(0)\n"            /* a prefetch that was not used has landed: the set is free */                      \
      "L_node2_%=:\n"                                                                                                           \
      "  s_buffer_load_dwordx4 " PKW_SETA ", s[" K36 ":" K39 "], %[ref]\n"   /* (the reference is the byte offset; its two low bits do not address) */ \
      GLOME_PKW_DISPATCH_FALL("A")                                                                                                \
      /* ------------------------------------------------------------ branch steps */                                          \
      GLOME_PKW_STEPS("A", "s[" K0 ":" K1 "]", "s" K2, "s" K3, PKW_SETB, "B", AX, AY, AZ)                                        \
      GLOME_PKW_STEPS("B", "s[" K4 ":" K5 "]", "s" K6, "s" K7, PKW_SETA, "A", AX, AY, AZ)                                        \
      /* ------------------------------------------------------------ a leaf: two triangles per test */                        \
      "L_leaf_%=:\n"                        /* %[ref] = byte offset of the first pair record | 3 */                             \
      "  s_mov_b64 s[" K34 ":" K35 "], exec\n"   /* the leaf's lanes */                                                         \
      "L_tri_%=:\n"                                                                                                             \
      "  s_load_dwordx16 s[" K8 ":" K23 "], %[pairs], %[ref]\n"                                                                 \
      "  s_load_dwordx4 s[" K24 ":" K27 "], %[pairs], %[ref] offset:0x40\n"                                                     \
      "  s_waitcnt lgkmcnt(0)\n"                                                                                                \
      "  s_add_u32 s" K33 ", s" K27 ", 1\n" /* the second triangle's record index */                                           \
      GLOME_PKW_PAIR_ARITH                                                                                                      \
      GLOME_PKW_CHAIN_A                                                                                                         \
      GLOME_PKW_UPDATE_##M(T2, "s" K27)                                                                                        \
      "  s_cmp_eq_u32 s" K26 ", 1\n"                                                                                            \
      "  s_cbranch_scc1 L_pop_%=\n"         /* that was the leaf's last */                                                      \
      "  s_mov_b64 exec, s[" K34 ":" K35 "]\n"                                                                                  \
      GLOME_PKW_CHAIN_B                                                                                                         \
      GLOME_PKW_UPDATE_##M(T3, "s" K33)                                                                                         \
      "  s_cmp_eq_u32 s" K26 ", 2\n"                                                                                            \
      "  s_cbranch_scc1 L_pop_%=\n"                                                                                             \
      "  s_add_u32 %[ref], %[ref], 80\n"                                                                                        \
      "  s_mov_b64 exec, s[" K34 ":" K35 "]\n"                                                                                  \
      "  s_branch L_tri_%=\n"                                                                                                   \
      /* ------------------------------------------------------------ pop until an entry some lane still wants */              \
      "L_pop_%=:\n"                                                                                                             \
      "  s_sub_u32 m0, m0, 1\n"             /* SCC = the stack was empty */                                                     \
      "  s_cbranch_scc1 L_empty_%=\n"                                                                                           \
      "  s_cmp_ge_u32 m0, %[cap]\n"                                                                                             \
      "  s_cbranch_scc1 L_popslow_%=\n"                                                                                         \
      "  v_readlane_b32 %[ref], %[ur], m0\n"                                                                                    \
      "  v_readlane_b32 s" K28 ", %[ulo], m0\n"                                                                                 \
      "  v_readlane_b32 s" K29 ", %[uhi], m0\n"                                                                                 \
      GLOME_PKW_POPMASK_##M                                                                                                     \
      "  v_lshl_add_u32 v" T2 ", m0, 8, %[lds]\n"                                                                               \
      "  ds_read_b32 %[near], v" T2 "\n"                                                                                        \
      "  ds_read_b32 %[far], v" T2 " offset:%[row1]\n"                                                                          \
      "  s_waitcnt lgkmcnt(0)\n"                                                                                                \
      GLOME_PKW_FILTER_##M                                                                                                      \
      "  s_buffer_load_dwordx4 " PKW_SETA ", s[" K36 ":" K39 "], %[ref]\n"   /* (as at L_node2: nothing is in flight behind that wait) */ \
      GLOME_PKW_DISPATCH("A")                                                                                                   \
      /* ------------------------------------------------------------ exits */                                                 \
      "L_empty_%=:\n"                                                                                                           \
      "  s_mov_b32 m0, 0\n"                                                                                                     \
      "  s_mov_b64 %[am], 0\n"                                                                                                  \
      "  s_mov_b32 %[status], 0\n"                                                                                              \
      "  s_branch L_end_%=\n"                                                                                                   \
      "L_slow_%=:\n"                          /* a push that does not fit the LDS part: the C++ step takes this node */         \
      "  s_mov_b64 %[am], exec\n"                                                                                               \
      "  s_mov_b32 %[status], 1\n"                                                                                              \
      "  s_branch L_dump_%=\n"                                                                                                  \
      "L_popslow_%=:\n"                       /* the top entry lives in the overflow columns */                                 \
      "  s_add_u32 m0, m0, 1\n"                                                                                                 \
      "  s_mov_b32 %[status], 3\n"                                                                                              \
      "L_dump_%=:\n"                          /* a C++ step follows: the entries (lane k = entry k) leave the registers */       \
      "  s_mov_b64 exec, -1\n"                                                                                                  \
      "  v_mbcnt_lo_u32_b32 v" T2 ", -1, 0\n"                                                                                   \
      "  v_mbcnt_hi_u32_b32 v" T2 ", -1, v" T2 "\n"                                                                             \
      "  v_lshlrev_b32 v" T2 ", 2, v" T2 "\n"                                                                                   \
      "  global_store_dword v" T2 ", %[ur], %[dump]\n"                                                                          \
      "  global_store_dword v" T2 ", %[ulo], %[dump] offset:256\n"                                                              \
      "  global_store_dword v" T2 ", %[uhi], %[dump] offset:512\n"                                                              \
      "  s_waitcnt vmcnt(0)\n"                                                                                                  \
      "L_end_%=:\n"                                                                                                             \
      "  s_mov_b64 exec, s[" K30 ":" K31 "]\n"                                                                                  \
      "  s_waitcnt lgkmcnt(0)\n"              /* no prefetch outlives the block: its registers go back to the compiler */       \
      "  s_mov_b32 %[sp], m0\n"                                                                                                 \
      "  s_mov_b32 m0, s" K32 "\n"                                                                                              \
      : [ref] "+s"(ref), [am] "+s"(am), [sp] "+s"(sp), [near] "+v"(nearv), [far] "+v"(farv), [best_t] "+v"(best_t), [best_rec] "+v"(best_rec), [ur] "=&v"(ur),    \
        [ulo] "=&v"(ulo), [uhi] "=&v"(uhi), [occ] "+s"(occm), [status] "=&s"(status)                                                                                  \
      : [nodes_lo] "s"(nodes_lo), [nodes_hi] "s"(nodes_hi), [nbytes] "s"(nbytes), [pairs] "s"(pairs), [cap] "n"(CAP), [phase] "s"(phase), [P0] "v"(P0), [P1] "v"(P1),  \
        [P2] "v"(P2), [R0] "v"(R0), [R1] "v"(R1), [lds] "v"(lds_row), [row1] "n"(CAP * 256), [dump] "s"(dump)                                                         \
      : "s" K0, "s" K1, "s" K2, "s" K3, "s" K4, "s" K5, "s" K6, "s" K7, "s" K8, "s" K9, "s" K10, "s" K11, "s" K12, "s" K13, "s" K14, "s" K15, "s" K16, "s" K17, "s" K18, \
        "s" K19, "s" K20, "s" K21, "s" K22, "s" K23, "s" K24, "s" K25, "s" K26, "s" K27, "s" K28, "s" K29, "s" K30, "s" K31, "s" K32, "s" K33, "s" K34, "s" K35,          \
        "s" K36, "s" K37, "s" K38, "s" K39,                                                                                                                          \
        "v" T0, "v" T1, "v" T2, "v" T3, "v" T4, "v" T5, "v" T6, "v" T7, "v" T8, "v" T9, "v" T10, "v" T11, "v" T12, "v" T13, "v" T14, "v" T15, "vcc", "scc", "memory")

// MODE 1 = closest hit with ordered early-out, MODE 2 = any hit.  XF / YF / ZF: the rays of the packet run towards +x / +y / +z.
// CAP: entries of the LDS part of the stack (the far row lies CAP * 256 bytes after the near row).
// phase 0: go on from `ref` with the lanes `am`; phase 1: pop first.  Returns a PKW_* status.
// `nodes` / `nbytes`: the walk's own node pool (DScene::pknodes) and its size; `ref`, like every stack entry, in its form (a
// branch: byte offset | axis, a leaf: byte offset of its first pair record | 3); `pairs`: the pair records (DScene::tripairs).
// best_rec receives the record index of the triangle hit (the pair record carries it).
typedef float pkw_f2 __attribute__((ext_vector_type(2)));
template <int MODE, bool XF, bool YF, bool ZF, int CAP>
__device__ __forceinline__ int bih_walk_asm(const F4* nodes, uint32_t nbytes, const float* pairs, int phase, uint32_t& ref, LaneMask& am, int& sp, float& nearv, float& farv,
                                            float& best_t, uint32_t& best_rec, LaneMask& occm, V3 o, V3 rcp, V3 d, uint32_t lds_row, uint32_t* dump) {
  static_assert(MODE == 1 || MODE == 2, "the hand-written walk covers the production traversals only");
  int status;
  uint32_t ur, ulo, uhi;  // the wave-uniform part of the stack, entry k in lane k -- alive inside the asm block only
  // the ray as aligned register pairs (op_sel picks a half): origin and direction for the triangle tests, origin and
  // reciprocal direction for the plane distances
  const pkw_f2 P0 = {o.x, o.y}, P1 = {o.z, d.x}, P2 = {d.y, d.z}, R0 = {rcp.x, rcp.y}, R1 = {rcp.z, rcp.z};
  // wave-uniform by construction; readfirstlane pins them to scalar registers where the compiler cannot see that
  ref = uni(ref); am = uni(am); sp = (int)uni((uint32_t)sp); occm = uni(occm); phase = (int)uni((uint32_t)phase); nbytes = uni(nbytes);
  const uint32_t nodes_lo = uni((uint32_t)(uintptr_t)nodes), nodes_hi = uni((uint32_t)((uintptr_t)nodes >> 32));
  pairs = (const float*)(uintptr_t)uni((LaneMask)(uintptr_t)pairs);
  dump = (uint32_t*)(uintptr_t)uni((LaneMask)(uintptr_t)dump);  // the wave's dump block (LaneStack::dump_base): a scalar pair, the lane's offset is made where it is used
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
