// glome_device.hip -- HIP kernels for gfx950 and the device half of the C ABI (include/glome_hip.h):
// context, scene commit/upload, the per-ray batch seams and the whole-frame render.
//
// Kernel catalogue
//   k_render_flat<FAITHFUL,COUNT,FULL,CLS,LB,TWO_ROWS>
//                                       persistent: one wave pulls 64-pixel work items (8x8 blocks of a 65x65
//                                       reference tile, Glome.hs:371-386; up to 32 frames per launch) from a ticket
//                                       queue of eight heads; the wave walks a triangle / sphere BIH once for its 64 rays
//                                       (packet: rt_device.hpp bih_tri_wave; for triangles the hand-written walk of
//                                       bih_packet_asm.hpp) -> shadow rays -> shade; secondary rays re-enter the same walk
//                                       through the shading state machine (shade_vm).  No ray streams in HBM at all.
//   k_render_generic                    same loop over the generic interpreter (rt_generic.hpp: rayint / shadow / inside / get_metainfo of any
//                                       nesting of composites as one loop over explicit frames)
//   k_ss_frame_flat / k_ss_frame_generic  the adaptive sampler (renderTileSubsample, Glome.hs:226-323): five passes per
//                                       tile, one launch per frame or batch of frames
//   k_rayint_batch / k_shadow_batch / k_inside_batch   the `Solid` method seams on SoA ray streams
//   k_tiles_pack / k_tiles_blit / k_tiles_blit_packed  Tile payload <-> frame (blitTile, Glome.hs:353-358)
//   k_bb_* / k_mb_* (bih_build_device.hpp)             `bih` and the Mesh BVH built level by level (Bih.hs:211-285, Mesh.hs:69-113)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/glome_hip.h"
#include "capi_shared.hpp"
#include "flatten.hpp"
#include "tiles.hpp"
#include "rt_device.hpp"
#include "rt_generic.hpp"
#ifndef GLOME_GENERIC_LB
#define GLOME_GENERIC_LB 2  // waves per SIMD the generic-tier kernels are compiled for (256 VGPRs)
#endif
// This file is compiled several times (glome_amd/build.py, in parallel): -DGLOME_PART=k keeps the kernel instances listed for
// part k further down (and their launchers); part 0 is the host runtime, the light kernels and the tree builders.  Without the
// define the whole file is one translation unit, as it used to be (4.5 minutes of hipcc).
#ifndef GLOME_PART
#define GLOME_PART -1
#endif
#define GLOME_IN_PART(k) (GLOME_PART == -1 || GLOME_PART == (k))
#if GLOME_IN_PART(0)
#include "bih_build_device.hpp"
#endif

using namespace glome;

// ------------------------------------------------------------------------------------------------ tiers
// FAITHFUL = the reference's exact node-visit order (no ordered early-out, Bih.hs:332-368); COUNT = node / primitive
// work counters.  They back the `faithful` / `count_work` render params (byte-model measurement, parity tests).
// The production variant traverses with early-out and counts rays only.
template <bool FAITHFUL, bool COUNT, bool FULL_, int CLS = CLS_ALL>
struct FlatTier {
  static constexpr bool FULL = FULL_;  // false: lean kernel (no secondary rays, no Blend / Layers)
  static constexpr bool WARP = false;  // scenes with a Warp material (traces over other roots, Shader.hs:157-175) render on the generic tier
  const DScene& S;
  const DLight* lights;
  int nlights;
  LaneStack stk;
  Cnt cnt;
  unsigned int err = 0;  // a CSG item ran into the advance / frame cap (kernels with CLS_CSG)
  // the same tier over another copy of the launch's arguments (render_loop: the kernarg segment, re-read per work item), and back
  __device__ __forceinline__ FlatTier rebound(const DRenderArgs& A) const { return FlatTier{A.S, A.lights, A.nlights, stk, cnt, err}; }
  __device__ __forceinline__ void absorb(const FlatTier& t) { cnt = t.cnt; err = t.err; }
  __device__ __forceinline__ HitG closest(const Ray& r, float tmax) {
    HitG ch;
    Cand c = closest_flat<FAITHFUL, COUNT, CLS>(S, r, tmax, stk, cnt, true, &ch, &err);
    return finalize_flat<CLS>(S, r, c, &ch);
  }
  __device__ __forceinline__ bool occluded(const Ray& r, float d, uint32_t = 0) { return occluded_flat<COUNT, CLS>(S, r, d, stk, cnt, true, &err); }
  // wave-wide calls (every lane of the wave makes them together; `valid` = the lane holds a ray): triangle and sphere
  // BIHs are walked as packets, by primary, shadow and secondary rays alike
  static constexpr bool PACKETS = (CLS & (CLS_BIH_TRI | CLS_BIH_SPHERE | CLS_MESH)) != 0;
  __device__ __forceinline__ HitG closest_wave(const Ray& r, float tmax, bool valid, uint32_t = 0) {
    if constexpr (PACKETS) {
      HitG ch;
      Cand c = closest_flat<FAITHFUL, COUNT, CLS, true>(S, r, tmax, stk, cnt, valid, &ch, &err);
      return valid ? finalize_flat<CLS>(S, r, c, &ch) : hit_miss();
    } else {
      return valid ? closest(r, tmax) : hit_miss();
    }
  }
  __device__ __forceinline__ bool occluded_wave(const Ray& r, float d, bool valid) {
    if constexpr (PACKETS) return occluded_flat<COUNT, CLS, true>(S, r, d, stk, cnt, valid, &err);
    else return valid && occluded(r, d);
  }
};
// the kernel's own arguments where the dispatch put them (constant memory; the first explicit argument is at offset 0)
// (relies on the code-object ABI placing the first explicit by-value argument at offset 0 of the kernarg segment in host layout:
// checked once per process by k_kernarg_selftest, glome_ctx_create)
template <class ARGS> __device__ __forceinline__ const ARGS& kernel_args() {
  static_assert(std::is_trivially_copyable<ARGS>::value && alignof(ARGS) <= 16, "kernel_args: a by-value kernel argument in host layout");
  return *(const ARGS*)(const ARGS __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
}
#if GLOME_IN_PART(0)
// what kernel_args assumes, asked of the device: the unnamed first argument read through the kernarg pointer equals the bytes the
// host passed (a second, named copy of them travels as a pointer)
__global__ void k_kernarg_selftest(DRenderArgs, const DRenderArgs* expect, unsigned int* ok) {
  const unsigned char* a = (const unsigned char*)&kernel_args<DRenderArgs>();
  const unsigned char* b = (const unsigned char*)expect;
  unsigned int same = 1;
  for (size_t i = threadIdx.x; i < sizeof(DRenderArgs); i += blockDim.x) same &= a[i] == b[i] ? 1u : 0u;
  if (!same) atomicAnd(ok, 0u);
}
#endif
// PKMIN: lanes that must wait before the packet service walks (rt_generic.hpp vm_run).  COUNT: bih_nodes / prim_tests are counted -- asked for by
// glome_render_params.count_work; the instances that do not count are 4 % (renderTile) and 2 % (sampler) faster on GlomeView's default scene
// (profiles/r04_probes/generic_tier_no_count_ab.txt), like the flat tier's lean instances.
template <int PKMIN = kPkMinLanes, bool COUNT = true>
struct GenericTierT {
  static constexpr bool FULL = true;
  static constexpr bool WARP = true;
  // What the out-of-line interpreter calls take the address of -- counters, error flag, frame memory -- are locals of the kernel,
  // referred to from here, and S refers to the kernel-argument segment itself (kernel_args): this struct then never needs an
  // address, lives in registers, and a pool's base is one scalar load from the argument segment.  (With the members inside the
  // struct and S a reference to the by-value argument, both were kept in scratch: every pool access began with a per-lane flat load
  // of the pool's base pointer from the scratch copy of DScene -- two dependent round trips per primitive test.)
  const DScene& S;
  const DLight* lights;
  int nlights;
  Cnt& cnt;
  unsigned int& err;
  uint32_t* vm;  // the interpreter's frames: one word stack of kVmWords per lane for the whole kernel (scratch)
  LaneStack pk;  // the wave's LDS stack for packet walks of sphere BIHs inside the interpreter (cap 0: the scene has none)
  __device__ __forceinline__ GenericTierT rebound(const DRenderArgs&) const { return *this; }  // (already reads the kernarg segment: kernel_args<>())
  __device__ __forceinline__ void absorb(const GenericTierT&) {}
  // `root`: the record the trace runs over -- the scene's, or the frame / scene of a Warp material
  __device__ __forceinline__ HitG closest(const Ray& r, float tmax, uint32_t root) { return vm_closest<COUNT, PKMIN>(S, cnt, err, vm, pk.cap > 0 ? &pk : (LaneStack*)nullptr, r, tmax, root); }
  __device__ __forceinline__ bool occluded(const Ray& r, float d, uint32_t root) { return vm_occluded<COUNT, PKMIN>(S, cnt, err, vm, pk.cap > 0 ? &pk : (LaneStack*)nullptr, r, d, root); }
  __device__ __forceinline__ HitG closest(const Ray& r, float tmax) { return closest(r, tmax, S.root_rec); }
  __device__ __forceinline__ bool occluded(const Ray& r, float d) { return occluded(r, d, S.root_rec); }
  __device__ __forceinline__ HitG closest_wave(const Ray& r, float tmax, bool valid, uint32_t root) { return valid ? closest(r, tmax, root) : hit_miss(); }
  __device__ __forceinline__ bool occluded_wave(const Ray& r, float d, bool valid) { return valid && occluded(r, d); }
};
using GenericTier = GenericTierT<>;

// LDS carve per wave: three stack rows of cap * 64 words (reference, near, far -- the per-lane traversal's entries), or
// two (near, far) in kernels that only ever run the hand-written packet walk, which keeps its references in registers
template <bool TWO_ROWS = false>
__device__ __forceinline__ LaneStack lane_stack(uint32_t* lds, int cap, uint32_t* ovf_base, int ovf_cap) {
  const int wave = threadIdx.x >> 6;  // (uniform per wave; the flat kernels run one wave per workgroup)
  LaneStack s;
  s.lds = lds + (size_t)wave * cap * 64 * (TWO_ROWS ? 2 : 3);
  s.cap = cap;
  s.has_ref_row = !TWO_ROWS;
  // overflow: one [entry * 3][64] block per wave slot (blockIdx.x * waves_per_block + wave); the block after the last
  // entry is the dump block
  s.ovf_cap = ovf_cap;
  s.ovfb = ovf_base ? ovf_base + ((size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * (ovf_cap + 1) * 3) * 64 : nullptr;
  return s;
}
static size_t flat_lds_bytes(int cap, bool two_rows = false) { return (size_t)cap * 64 * (two_rows ? 8 : 12); }
// the generic tier's packet stack: three rows in LDS, no overflow columns (a tree deeper than `cap` keeps the per-lane walk)
__device__ __forceinline__ LaneStack generic_packet_stack(uint32_t* lds, int cap) {
  return lane_stack<false>(lds, cap, nullptr, 0);  // (ovfb null: no overflow columns and no dump block -- bih_tri_wave then never takes the hand-written walk)
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned int v) {
  unsigned long long s = v;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  return s;
}
__device__ __forceinline__ void flush_counters(DCounters* c, const Cnt& cnt, unsigned int err) {
  unsigned long long a = wave_sum(cnt.primary) + cnt.w_primary, b = wave_sum(cnt.shadow) + cnt.w_shadow, s = wave_sum(cnt.secondary);
  unsigned long long n = wave_sum(cnt.bih), m = wave_sum(cnt.mesh), p = wave_sum(cnt.prim);
  unsigned long long e = wave_sum(err);
  if ((threadIdx.x & 63) == 0) {
    if (a) atomicAdd(&c->rays_primary, a);
    if (b) atomicAdd(&c->rays_shadow, b);
    if (s) atomicAdd(&c->rays_secondary, s);
    if (n) atomicAdd(&c->bih_nodes, n);
    if (m) atomicAdd(&c->mesh_nodes, m);
    if (p) atomicAdd(&c->prim_tests, p);
    if (e) atomicOr(&c->error, 1u);
  }
}

// work item w -> tile + 64 pixels.  A tile is cut into 8x8 blocks (coherent rays per wave); the pixels left over on
// the right and bottom edges (65 = 8*8 + 1) are packed 64 at a time, so lanes are not wasted on partial blocks.
__device__ __forceinline__ bool work_to_pixel(const DRenderArgs& A, uint32_t w, int lane, int& px, int& py, size_t& dense_off) {
  int lo = (int)A.tile_lut[w >> 6];  // the tile of item (w & ~63); w's own is that one or one of the next few
  while (lo + 1 < A.ntiles && A.tiles[lo + 1].wave_base <= w) lo++;
  DTile t = A.tiles[lo];
  uint32_t j = w - t.wave_base;
  uint32_t nbx = t.w / kBlockW, nby = t.h / kBlockH, nblk = nbx * nby;
  int lx, ly;
  if (j < nblk) {
    lx = (j % nbx) * kBlockW + (lane % kBlockW);
    ly = (j / nbx) * kBlockH + (lane / kBlockW);
  } else {
    uint32_t i = (j - nblk) * 64 + lane;
    uint32_t rw = t.w - kBlockW * nbx, rcount = rw * t.h;
    if (i < rcount) { lx = kBlockW * nbx + i % rw; ly = i / rw; }
    else {
      uint32_t i2 = i - rcount, bw = kBlockW * nbx, bh = t.h - kBlockH * nby;
      if (i2 >= bw * bh) return false;
      lx = i2 % bw; ly = kBlockH * nby + i2 / bw;
    }
  }
  px = t.x + lx; py = t.y + ly;
  dense_off = (size_t)t.pix_base + (size_t)ly * t.w + lx;
  return true;
}

// The work queue of a render launch.  One ticket counter cannot feed the GPU: a returning atomic on one word completes
// about every 11 ns (MI355X_MICROARCH.md, "dequeue": ~88 per microsecond), a frame of the flagship scene is 32,400 items
// and the 6,144 resident waves get through ~150 of them per microsecond -- the waves queue up behind the counter.
// (Measured with one counter: a launch running alone took 0.39 ms per frame whatever its grid, four launches on four slots
// -- four counters -- 0.226.)  So the queue has kQueueShards heads, each on a cache line of its own; ticket chunk c
// (kQueueChunk consecutive items: one 64x64 work tile) belongs to head c mod kQueueShards, so the order in which the image
// is worked through stays what it was.  A wave starts at the head of its XCD (blocks are dealt round-robin over the XCDs:
// speed only, never correctness) and moves on when a head runs dry.  Heads found dry are published in a mask word that is
// written a handful of times per launch and therefore cheap to read (a load of a head itself would wait behind the
// atomics queued on its line: measured 4x slower), so a wave rarely pays for more than one failed take.  The last wave to
// leave puts everything back to zero: the next launch on the slot needs no reset packet on the stream.
constexpr uint32_t kNoTicket = 0xffffffffu;
// A wave may take several tickets per atomic while the head is far from empty and single ones towards the end (guided
// self-scheduling).  Measured in round 3 and left OFF (largest batch 1): in-kernel stamps put a wave's wait for a ticket at
// 2,500-3,100 cycles, 3-4 % of its lifetime on the flagship frame (21 % on a frame of empty sky, where the 46 us that 32,400
// serialised atomics take on 8 heads are most of the frame); batches of 4 or 8 won 0-7 % pipelined and lost 15-30 % on a launch
// alone, whose last items then run on too few waves (tools/probe/empty_frame.py; fixed and guided batches alike).
#ifndef GLOME_TICKET_BATCH
#define GLOME_TICKET_BATCH 1  // the largest batch
#endif
struct TicketQueue {
  uint32_t shard, dry;
  uint32_t inext = 0, left = 0, cur = 0;  // a batch in hand: its next queue index, tickets left in it, the head it came from (lane 0's)
  uint32_t batch;
  __device__ __forceinline__ uint32_t batch_for(const DRenderArgs& A, uint32_t remaining) const {  // ~half a fair share of what is left, 1..GLOME_TICKET_BATCH
    const uint32_t waves_per_head = (gridDim.x + kQueueShards - 1) / kQueueShards;
    const uint32_t b = remaining / (2u * waves_per_head);
    return b < 1u ? 1u : (b > (uint32_t)GLOME_TICKET_BATCH ? (uint32_t)GLOME_TICKET_BATCH : b);
  }
  __device__ __forceinline__ TicketQueue(const DRenderArgs& A) : shard(blockIdx.x % kQueueShards), dry(0) { batch = batch_for(A, A.shard_cap); }
  // Every lane of the wave makes the call; the state is wave-uniform (scalar registers) and only the atomics themselves are lane 0's.
  // (Until round 3 the whole take ran on lane 0 under a branch: its six state words then lived in vector registers for the kernel's lifetime.)
  __device__ __forceinline__ uint32_t take(const DRenderArgs& A) {
    constexpr uint32_t kAll = (1u << kQueueShards) - 1u;
    for (;;) {
      if (left) {
        left--;
        const uint32_t i = inext++;
        if (i < A.shard_cap) return ((i / kQueueChunk) * kQueueShards + cur) * kQueueChunk + (i % kQueueChunk);
        left = 0;  // the batch reached past the head's last ticket
      }
      if (dry == kAll) return kNoTicket;
      if (!((dry >> shard) & 1u)) {
        uint32_t i = 0;
        if (LaneStack::lane() == 0) i = atomicAdd(&A.counters->heads[shard * kQueueHeadStride], batch);
        i = uni(i);
        if (i < A.shard_cap) { inext = i; left = batch; cur = shard; batch = batch_for(A, A.shard_cap - i); continue; }
        uint32_t d = 0;
        if (LaneStack::lane() == 0) { atomicOr(&A.counters->dry, 1u << shard); d = __hip_atomic_load(&A.counters->dry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        dry |= (1u << shard) | uni(d);
        batch = 1;  // what other heads have left is shared by everybody who comes by
      }
      shard = (shard + 1) % kQueueShards;
    }
  }
  __device__ __forceinline__ void leave(const DRenderArgs& A) {  // after the wave's last take (every lane calls; lane 0 acts)
    if (LaneStack::lane() != 0) return;
    if (atomicAdd(&A.counters->done, 1u) == gridDim.x - 1u) {  // every other wave has taken its last ticket
      for (uint32_t h = 0; h < kQueueShards; h++) __hip_atomic_store(&A.counters->heads[h * kQueueHeadStride], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&A.counters->dry, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&A.counters->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
};

// GLOME_PROBE (a measurement build, tools/probe/empty_frame.py; never the product): GLOME_DEBUG_FLAGS leaves parts of a work item
// out (1 no pixel store, 2 no trace, 4 no item -> pixel lookup, 8 static items instead of tickets) or (16) stamps its sections
// with s_memtime into DCounters::dbg.  Compiled out of the product: the stamps alone cost the flagship kernel 5 %.
#ifdef GLOME_PROBE
#define GLOME_PROBE_FLAG(A, bit) ((A).debug_flags & (bit))
#else
#define GLOME_PROBE_FLAG(A, bit) false
#endif
// LEAN (the flagship instance: every step the hand-written walk's, six waves per SIMD, 80 vector registers): the three measures of
// DESIGN.md 4.1c that take registers out of the walks' way -- arguments re-read per item through an opaque pointer, the ticket taken as
// a scalar, the pixel made a second time after the trace.  They are worth 8-10 % there and COST the other instances, whose C++ walks
// then re-read table pointers inside their loops: the 1M-triangle Mesh 0.797 -> 0.872 ms with all three, 0.84 with any one of them off
// (profiles/r04_probes/mesh_regress_ab.txt); so the other flat-tier instances keep round 3's loop.
template <bool LEAN, class TIER>
__device__ __forceinline__ void render_loop(const DRenderArgs& A_, TIER& Tk) {
#ifdef GLOME_PROBE
  int lane = threadIdx.x & 63;
#endif
  TicketQueue Q(A_);
  // The launch's arguments are read where the dispatch put them (the kernarg segment: scalar loads), through a pointer the compiler
  // cannot see through from one work item to the next: what an item derives from them -- (float)width, the reciprocals of the
  // item -> pixel divisions, the table pointers -- is then made afresh per item (tens of instructions in eleven thousand) instead of
  // being hoisted out of the loop and carried, spilled, through both walks.
  const DRenderArgs __attribute__((address_space(4)))* ap_ = (const DRenderArgs __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
#ifdef GLOME_PROBE
  uint32_t stat = blockIdx.x;
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, t_s[3] = {0, 0, 0};  // item -> pixel lookup, ray generation, trace
  unsigned long long t_take = 0, n_take = 0, t_begin = (A.debug_flags & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();  // the 100 MHz clock every CU shares
  unsigned long long rt_last_item = rt_begin, worst = 0;
  uint32_t worst_steps = 0, steps_before = 0;
#endif
  for (;;) {
    if constexpr (LEAN) asm volatile("" : "+s"(ap_));
    const DRenderArgs& A = LEAN ? *(const DRenderArgs*)ap_ : A_;
    TIER T = Tk.rebound(A);
    uint32_t w = kNoTicket;
#ifdef GLOME_PROBE
    if (A.debug_flags & 8) { w = stat < A.total_waves * (uint32_t)A.nframes ? stat : kNoTicket; stat += gridDim.x; }
    else {
      const unsigned long long t0 = (A.debug_flags & 16) ? __builtin_amdgcn_s_memtime() : 0ull;
      w = Q.take(A);
      if (A.debug_flags & 16) { ts0 = __builtin_amdgcn_s_memtime(); t_take += ts0 - t0; n_take++; }
    }
#else
    if constexpr (LEAN) w = Q.take(A);  // a SCALAR: the frame, the tile and the camera the ticket names are then scalar loads, not a lane's
    else { if (LaneStack::lane() == 0) w = Q.take(A); w = __shfl(w, 0, 64); }
#endif
    if (w == kNoTicket) break;
#ifdef GLOME_PROBE
    rt_last_item = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t frame;  // wave-uniform
    if (A.chunks_per_frame) {
      // chunk by chunk through all frames: the same 64x64 work tile of every view one after the other (neighbouring views walk the
      // same part of the tree), and what a launch ends with is the last chunks of ALL its frames, not the whole of its last frame
      const uint32_t g = w / kQueueChunk, nf = (uint32_t)A.nframes;
      if (g >= A.chunks_per_frame * nf) continue;  // padding of the last round of chunks
      frame = g % nf;
      w = (g / nf) * kQueueChunk + (w % kQueueChunk);
      if (w >= A.total_waves) continue;            // padding of a frame's last chunk
    } else {
      if (w >= A.total_waves * (uint32_t)A.nframes) continue;  // padding of the last round of chunks
      frame = w / A.total_waves;
      w -= frame * A.total_waves;
    }
    int px = 0, py = 0;
    size_t dense_off = 0;
    bool valid;
    if (GLOME_PROBE_FLAG(A, 4)) { const uint32_t l_ = LaneStack::lane(); px = (int)((w * 64u + l_) % (uint32_t)A.width); py = (int)((w * 64u + l_) / (uint32_t)A.width); valid = py < A.height; }
    else valid = work_to_pixel(A, w, (int)LaneStack::lane(), px, py, dense_off);  // lanes past the end of a leftover strip idle along
#ifdef GLOME_PROBE
    if (A.debug_flags & 16) { asm volatile("" :: "v"(px), "v"(py)); ts1 = __builtin_amdgcn_s_memtime(); }
#endif
    float xc, yc;
    get_coordsf(A.width, A.height, (float)px, (float)py, xc, yc);
    Ray ray = primary_ray(frame == 0 ? A.cam : A.more_cams[frame - 1], xc, yc);
    count_wave(T.cnt.primary, T.cnt.w_primary, valid);
#ifdef GLOME_PROBE
    if (A.debug_flags & 16) { asm volatile("" :: "v"(ray.d.x), "v"(ray.d.y), "v"(ray.d.z)); ts2 = __builtin_amdgcn_s_memtime(); }
#endif
    HitG h;
    CA c;
    if (GLOME_PROBE_FLAG(A, 2)) { c = ca(ray.d.x, ray.d.y, ray.d.z, 1.0f); h = hit_miss(); }
    else c = trace_primary(T, ray, kInf, A.maxdepth, valid, &h);  // Trace.trace lights shader sld ray infinity maxdepth (Glome.hs:33)
    Tk.absorb(T);  // (counters and the error flag back into the kernel's tier)
#ifdef GLOME_PROBE
    if (A.debug_flags & 16) { asm volatile("" :: "v"(c.r), "v"(c.g), "v"(c.b)); ts3 = __builtin_amdgcn_s_memtime(); t_s[0] += ts1 - ts0; t_s[1] += ts2 - ts1; t_s[2] += ts3 - ts2; }
    if (A.debug_flags & 32) {  // the longest item, and the C++ steps its walks needed
      const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - rt_last_item;
      if (lane == 0) { if (dt > worst) { worst = dt; worst_steps = w; } }  // (which item)
      steps_before = T.cnt.bih;
    }
#endif
    if (!valid) continue;
    // the pixel once more (rather than three registers carried, spilled, through both walks): the item is a scalar, the lane a v_mbcnt
    if constexpr (LEAN) { if (!GLOME_PROBE_FLAG(A, 4)) { px = 0; py = 0; dense_off = 0; (void)work_to_pixel(A, w, (int)LaneStack::lane(), px, py, dense_off); } }
    float depth = h.hit ? h.t : kInf;      // ridepth
    float r = c.r;
    if (A.fog) r = r + (depth / 400);      // renderTile's debug fog (Glome.hs:174, Q20)
    size_t o = (A.dense ? dense_off : (size_t)py * A.width + px) + (size_t)frame * A.frame_stride;
    if (GLOME_PROBE_FLAG(A, 1)) { if (r == 12345.678f) A.packed[o] = 1u; continue; }
    if (A.out5) {
      float* out = A.out5 + o * 5;
      out[0] = r; out[1] = c.g; out[2] = c.b; out[3] = c.a; out[4] = depth;
    }
#ifdef GLOME_PROBE
    if (A.debug_flags & 64) { A.packed[o] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - rt_last_item); continue; }  // a cost image: the item's duration (10 ns units) in its pixels
#endif
    if (A.packed) A.packed[o] = rgbf(r * c.a, c.g * c.a, c.b * c.a);  // blitTile (Glome.hs:353-358)
  }
#ifdef GLOME_PROBE
  if ((A_.debug_flags & 16) && lane == 0) {  // cycles waiting for tickets, tickets asked for, the wave's lifetime, waves, cycles per section
    atomicAdd(&A_.counters->dbg[0], t_take); atomicAdd(&A_.counters->dbg[1], n_take);
    atomicAdd(&A_.counters->dbg[2], __builtin_amdgcn_s_memtime() - t_begin); atomicAdd(&A_.counters->dbg[3], 1ull);
    atomicAdd(&A_.counters->dbg[4], t_s[0]); atomicAdd(&A_.counters->dbg[5], t_s[1]); atomicAdd(&A_.counters->dbg[6], t_s[2]);
  }
  if ((A_.debug_flags & 32) && lane == 0) {  // the launch's timeline on the shared clock: first / last wave start, first / last wave's last ticket, first / last wave end
    const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
    atomicMin(&A_.counters->dbg[8], rt_begin); atomicMax(&A_.counters->dbg[9], rt_begin);
    atomicMin(&A_.counters->dbg[10], rt_last_item); atomicMax(&A_.counters->dbg[11], rt_last_item);
    atomicMin(&A_.counters->dbg[12], rt_end); atomicMax(&A_.counters->dbg[13], rt_end);
    atomicAdd(&A_.counters->dbg[14], (unsigned long long)Tk.cnt.bih);                        // C++ steps of all walks
    atomicMax(&A_.counters->dbg[15], (worst << 20) | (unsigned long long)worst_steps);     // the longest item (10 ns units) and its C++ steps
  }
  if (!(A_.debug_flags & 8)) Q.leave(A_);
#else
  Q.leave(A_);
#endif
}

// TWO_ROWS: the wave's LDS holds two stack rows per entry instead of three (lane_stack); legal when no lane ever pushes on
// its own -- a lean kernel of a triangle / sphere class over a scene whose materials are all Surface, where every ray of
// the frame goes through the packet walk.  With LB waves per SIMD asked of the register allocator that is 24 waves per
// CU instead of 16.
template <bool FAITHFUL, bool COUNT, bool FULL, int CLS, int LB = 1, bool TWO_ROWS = false>
__global__ void __launch_bounds__(64, LB) k_render_flat(DRenderArgs A, int stack_cap, uint32_t* ovf, int ovf_cap) {
  extern __shared__ uint32_t lds[];
  FlatTier<FAITHFUL, COUNT, FULL, CLS> T{A.S, A.lights, A.nlights, lane_stack<TWO_ROWS>(lds, stack_cap, ovf, ovf_cap), Cnt()};
  T.stk.dbg = A.counters->dbg;
  render_loop<TWO_ROWS>(A, T);
  if (A.want_counters) flush_counters(A.counters, T.cnt, T.err);
  else if ((CLS & (CLS_CSG | CLS_MESH)) && __builtin_amdgcn_ballot_w64(T.err != 0) && (threadIdx.x & 63) == 0) atomicOr(&A.counters->error, 1u);
}
#if GLOME_IN_PART(6) || GLOME_IN_PART(10)
template <bool COUNT>
__global__ void __launch_bounds__(64, GLOME_GENERIC_LB) k_render_generic(DRenderArgs) {
  const DRenderArgs& A = kernel_args<DRenderArgs>();
  extern __shared__ uint32_t lds[];
  Cnt cnt; unsigned int err = 0; uint32_t vm[kVmWords];
  GenericTierT<1, COUNT> T{A.S, A.lights, A.nlights, cnt, err, vm, generic_packet_stack(lds, (int)A.S.pk_generic_cap)};  // (<1>: this kernel's packet service never waits)
  // (Tried in round 3 and dropped: refilling a lane with the next pixel as soon as its trace is through, with shade_vm as a
  // resumable object.  The object form alone cost S4 0.39 -> 0.50 ms and this tier 4.3 -> 4.85 ms (its state no longer stays in
  // registers), and with refilling the lanes fall out of step, every closest-hit call then runs for a part of the wave, and the frame took 5.8 ms
  // against 4.3: what keeps the lanes idle -- 28 % of the vector lane slots are used -- is the interpreter's own divergence
  // inside a call, not pixels of unequal cost.)
  render_loop<true>(A, T);  // (the interpreter, short of registers like the flagship, measures better with the lean loop: TS 2.70 against 2.74 ms)
  if (A.want_counters) flush_counters(A.counters, T.cnt, T.err);
  else if (__builtin_amdgcn_ballot_w64(T.err != 0) && (threadIdx.x & 63) == 0) atomicOr(&A.counters->error, 1u);
}
#endif


// ------------------------------------------------------------------------------------------------ adaptive sampler
// renderTileSubsample (Glome.hs:226-323).  The reference runs five passes over each 65x65 tile; a pass looks at
// neighbour contrast (`decide`, Glome.hs:213-219) and either averages or traces a fresh sample.  Here persistent waves
// pull regions of a tile's candidate lattice (ss_block_pixel: blocks of 64 candidates of a pass in a compact pixel area,
// one per lane; a region = a rectangle of blocks, ss_region_shape); a lane takes the contrast test and writes the average when
// that settles it; the candidates that need a sample are compacted over the region (ballot + LDS ring) and traced 64 at a
// time -- neighbours in the image, so the rays are walked as a packet.  A region in which nobody needs a sample traces nothing.
// The working buffer `v` is a dense per-tile array in global memory (tile order, row major inside a tile), so all
// neighbour reads stay inside the tile like the reference's getc (Glome.hs:233-235); `v2` is the output.  A pass reads
// what the previous passes wrote anywhere in the tile: ss_frame_loop below orders the passes per tile.
// channel planes, not 5-float structs: the lanes of a block read neighbouring pixels, so a plane read is (nearly) contiguous
struct SSBuf { float* v; size_t plane; };
__device__ __forceinline__ void out5_store(float* v, size_t i, const TC& c) { float* p = v + i * 5; p[0] = c.r; p[1] = c.g; p[2] = c.b; p[3] = c.a; p[4] = c.d; }
__device__ __forceinline__ size_t ss_out_index(const DRenderArgs& A, const DTile& t, int dx, int dy) {
  return A.dense ? (size_t)t.pix_base + (size_t)dy * t.w + dx : (size_t)(t.y + dy) * A.width + (t.x + dx);
}
__device__ __forceinline__ void ss_write_out(const DRenderArgs& A, size_t frame_off, const DTile& t, int dx, int dy, const TC& c) {
  size_t o = ss_out_index(A, t, dx, dy) + frame_off;
  if (A.out5) out5_store(A.out5, o, c);
  if (A.packed) A.packed[o] = rgbf(c.r * c.a, c.g * c.a, c.b * c.a);
}

// One launch renders the frame: its work items are (pass, tile, region) in pass-major order, and an item of pass p waits
// for the tile's pass p - 1 (a counter per tile and pass) instead of the whole frame's -- no launch boundary between the
// passes, no tail of a short launch five times per frame, and the tiles that are early go on with their next pass while
// the late ones finish the last.
//   Order and progress: the items are dealt to kSSHeads queue heads by tile (tile mod kSSHeads); each head hands its items
//   out in order, so whatever an item waits for (earlier passes of the SAME tile) was handed out before it, to a wave that
//   is running: the oldest unfinished item of a head never waits.
//   Visibility: the working buffer `v` is written by one wave and read by others, on other CUs and XCDs, inside one
//   launch.  Every store to it is an agent-scope (sc1, write-through) store, every load an agent-scope (sc1) load that
//   bypasses the CU's L1; a wave drains its stores (s_waitcnt vmcnt(0)) before it counts its region as done, and polls the
//   counter with an agent-scope load before its first read (cdna_hip_programming.md, Guideline 16: payload and flag
//   both sc1, producer drained).  The pixels of the frame are write-only.
constexpr uint32_t kSSHeads = 8, kSSHeadStride = 32;
__device__ __forceinline__ float ss_ld(const float* p) { return as_f(__hip_atomic_load((const unsigned int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ void ss_st(float* p, float x) { __hip_atomic_store((unsigned int*)p, as_u(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ TC ss_load(const SSBuf& b, size_t i) { const float* p = b.v + i; return tc(ss_ld(p), ss_ld(p + b.plane), ss_ld(p + 2 * b.plane), ss_ld(p + 3 * b.plane), ss_ld(p + 4 * b.plane)); }
__device__ __forceinline__ void ss_store(const SSBuf& b, size_t i, const TC& c) { float* p = b.v + i; ss_st(p, c.r); ss_st(p + b.plane, c.g); ss_st(p + 2 * b.plane, c.b); ss_st(p + 3 * b.plane, c.a); ss_st(p + 4 * b.plane, c.d); }
__device__ __forceinline__ TC ss_getc(const SSBuf& v, const DTile& t, int dx, int dy) {  // getc: outside the tile reads blank
  // (the load is unconditional, from a clamped address, so the twenty loads of a contrast test go out back to back)
  const bool in = dx >= 0 && dx < t.w && dy >= 0 && dy < t.h;
  const TC c = ss_load(v, (size_t)t.pix_base + (in ? (size_t)dy * t.w + dx : (size_t)0));
  return in ? c : tc_blank();
}
struct SSPlan {  // per pass: regions per tile, regions per tile row, first item of the pass in a head's sequence
  uint32_t per_tile[6], nrx[6], first[7];
  uint32_t tiles_per_head;
};
__host__ __device__ inline SSPlan ss_plan(const DRenderArgs& A) {
  SSPlan P;
  P.tiles_per_head = ((uint32_t)A.ntiles * (uint32_t)A.nframes + kSSHeads - 1) / kSSHeads;  // (a tile of every frame of the launch)
  P.first[1] = 0; P.per_tile[0] = 0; P.nrx[0] = 1; P.first[0] = 0;
  for (int p = 1; p <= 5; p++) {
    int nrx;
    P.per_tile[p] = (uint32_t)ss_regions_per_tile(p, A.blocksize, A.ss_rw[p], A.ss_rh[p], nrx);  // laid out for full tiles; edge tiles leave regions empty
    P.nrx[p] = (uint32_t)nrx;
    P.first[p + 1] = P.first[p] + P.per_tile[p] * P.tiles_per_head;
  }
  return P;
}

template <class TIER>
__device__ __forceinline__ void ss_frame_loop(const DRenderArgs& A, TIER& T) {
  __shared__ uint32_t need_list[256];  // ring of the region's candidates that need a sample: dx | dy << 8 (one wave per block)
  const int lane = threadIdx.x & 63;
  const SSPlan PL = ss_plan(A);
  const uint32_t vtiles = (uint32_t)A.ntiles * (uint32_t)A.nframes;
  uint32_t shard = blockIdx.x % kSSHeads, dry = 0;  // (lane 0's)
  for (;;) {
    // ---- take the next item of a queue head (TicketQueue's scheme, over the frame's own heads)
    uint32_t w = kNoTicket, h = 0;
    if (lane == 0) {
      while (dry != (1u << kSSHeads) - 1u) {
        if (!((dry >> shard) & 1u)) {
          const uint32_t i = atomicAdd(&A.ss_cnt[shard * kSSHeadStride], 1u);
          if (i < PL.first[6]) { w = i; h = shard; break; }
          atomicOr(&A.ss_cnt[kSSHeads * kSSHeadStride], 1u << shard);
          dry |= (1u << shard) | __hip_atomic_load(&A.ss_cnt[kSSHeads * kSSHeadStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        shard = (shard + 1) % kSSHeads;
      }
    }
    w = __shfl(w, 0, 64); h = __shfl(h, 0, 64);
    if (w == kNoTicket) break;
    int pass = 1;
    while (w >= PL.first[pass + 1]) pass++;
    const uint32_t j = w - PL.first[pass];
    const uint32_t ti = (j / PL.per_tile[pass]) * kSSHeads + h;  // tile of a frame (ti mod kSSHeads == h): frame-major
    if (ti >= vtiles) continue;                                   // padding of the last round of tiles
    const int r = (int)(j % PL.per_tile[pass]), rx = r % (int)PL.nrx[pass], ry = r / (int)PL.nrx[pass];
    const uint32_t frame = ti / (uint32_t)A.ntiles;
    const DTile t = A.tiles[ti - frame * (uint32_t)A.ntiles];
    const SSBuf v{A.scratch + (size_t)frame * 5 * A.ss_plane, (size_t)A.ss_plane};  // every frame has its own working buffer
    const size_t frame_off = (size_t)frame * A.frame_stride;
    const DCamera& cam = frame == 0 ? A.cam : A.more_cams[frame - 1];
    unsigned int* done = A.ss_done + (size_t)ti * 8;
    if (pass >= 2) {  // the tile's previous pass must be complete (its regions read each other's pixels)
      if (lane == 0) while (__hip_atomic_load(&done[pass - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < PL.per_tile[pass - 1]) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_wave_barrier();
    }
    const float thr = pass >= 2 ? A.thresholds[pass - 2] : 0.0f;
    int ox[4], oy[4];
    ss_neighbours(pass, ox, oy);
    int bw, bh;
    ss_block_shape(pass, bw, bh);
    // the region's blocks inside the (possibly clipped) tile: [bx0, bx1) x [by0, by1)
    const int rw = A.ss_rw[pass], rh = A.ss_rh[pass];
    const int bx0 = rx * rw, by0 = ry * rh;
    const int bx1 = min(bx0 + rw, (t.w + bw - 1) / bw), by1 = min(by0 + rh, (t.h + bh - 1) / bh);
    const int nbx = bx1 - bx0, nb = nbx > 0 && by1 > by0 ? nbx * (by1 - by0) : 0;
    // ---- decide block after block; whenever 64 candidates wait for a sample (and at the region's end) they are traced as
    // one packet -- neighbours in the image.  Every pixel of the tile is written by exactly one of the passes 1-4 before a
    // later pass reads it (Glome.hs:241-297), so the blank initial value (:231) is only ever seen outside the tile (getc).
    uint32_t n = 0, hd = 0;  // wave-uniform: candidates listed / traced so far (ring positions)
    int b = 0;
    for (;;) {
      for (; b < nb && n - hd < 64u; b++) {
        const int bx = bx0 + b % nbx, by = by0 + b / nbx;
        int dx, dy;
        ss_block_pixel(pass, bx, by, lane, dx, dy);
        bool need = dx < t.w && dy < t.h;
        if (need && pass >= 2) {
          TC a = ss_getc(v, t, dx + ox[0], dy + oy[0]), bb = ss_getc(v, t, dx + ox[1], dy + oy[1]);
          TC c = ss_getc(v, t, dx + ox[2], dy + oy[2]), d = ss_getc(v, t, dx + ox[3], dy + oy[3]);
          need = gmaxf(ccmp(a, c), ccmp(bb, d)) > thr;  // decide, Glome.hs:215-216
          if (!need) {
            TC avg = cavg4(a, bb, c, d);
            if (pass < 5) ss_store(v, (size_t)t.pix_base + (size_t)dy * t.w + dx, avg);
            else ss_write_out(A, frame_off, t, dx, dy, ss_pass5_blend(avg, a, bb, c, d, dx == t.w - 1, dy == t.h - 1));
          }
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(need);
        if (need) need_list[(n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))) & 255u] = (uint32_t)dx | ((uint32_t)dy << 8);
        n += (uint32_t)__popcll(m);
      }
      if (n == hd) break;  // (b == nb: the region is through)
      __syncthreads();     // one wave per block: makes the list visible across lanes
      const uint32_t cnt = min(64u, n - hd);
      const bool valid = (uint32_t)lane < cnt;
      const uint32_t e = need_list[(hd + (valid ? (uint32_t)lane : 0u)) & 255u];
      hd += cnt;
      const int dx = (int)(e & 255u), dy = (int)(e >> 8);
      const float off = pass == 5 ? 0.5f : 0.0f;  // pass 5 samples between pixels (getCoordsf (x+.5) (y+.5), Glome.hs:307)
      float xc, yc;
      get_coordsf(A.width, A.height, (float)(t.x + dx) + off, (float)(t.y + dy) + off, xc, yc);
      Ray ray = primary_ray(cam, xc, yc);
      if (valid) T.cnt.primary++;
      HitG hh;
      CA col = trace_primary(T, ray, kInf, A.maxdepth, valid, &hh);
      if (valid) {
        TC smp = tc(col.r, col.g, col.b, col.a, hh.hit ? hh.t : kInf);
        if (pass < 5) ss_store(v, (size_t)t.pix_base + (size_t)dy * t.w + dx, smp);
        else {
          TC a = ss_getc(v, t, dx + ox[0], dy + oy[0]), bb = ss_getc(v, t, dx + ox[1], dy + oy[1]);
          TC c = ss_getc(v, t, dx + ox[2], dy + oy[2]), d = ss_getc(v, t, dx + ox[3], dy + oy[3]);
          ss_write_out(A, frame_off, t, dx, dy, ss_pass5_blend(smp, a, bb, c, d, dx == t.w - 1, dy == t.h - 1));
        }
      }
      __syncthreads();  // the entries just read may be overwritten by the blocks that follow
    }
    if (pass < 5) {   // the region's pixels are in memory before it counts as done
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) atomicAdd(&done[pass], 1u);
    }
  }
}
template <bool FULL, int CLS, int LB = 1, bool TWO_ROWS = false, bool FAITHFUL = false>
__global__ void __launch_bounds__(64, LB) k_ss_frame_flat(DRenderArgs A, int stack_cap, uint32_t* ovf, int ovf_cap) {
  extern __shared__ uint32_t lds[];
  FlatTier<FAITHFUL, false, FULL, CLS> T{A.S, A.lights, A.nlights, lane_stack<TWO_ROWS>(lds, stack_cap, ovf, ovf_cap), Cnt()};
  ss_frame_loop(A, T);
  if (A.want_counters) flush_counters(A.counters, T.cnt, T.err);
  else if ((CLS & (CLS_CSG | CLS_MESH)) && __builtin_amdgcn_ballot_w64(T.err != 0) && (threadIdx.x & 63) == 0) atomicOr(&A.counters->error, 1u);
}
#if GLOME_IN_PART(7) || GLOME_IN_PART(11)
template <bool COUNT>
__global__ void __launch_bounds__(64, GLOME_GENERIC_LB) k_ss_frame_generic(DRenderArgs) {
  const DRenderArgs& A = kernel_args<DRenderArgs>();
  extern __shared__ uint32_t lds[];
  Cnt cnt; unsigned int err = 0; uint32_t vm[kVmWords];
  GenericTierT<kPkMinLanes, COUNT> T{A.S, A.lights, A.nlights, cnt, err, vm, generic_packet_stack(lds, (int)A.S.pk_generic_cap)};
  ss_frame_loop(A, T);
  if (A.want_counters) flush_counters(A.counters, T.cnt, T.err);
  else if (__builtin_amdgcn_ballot_w64(T.err != 0) && (threadIdx.x & 63) == 0) atomicOr(&A.counters->error, 1u);
}
#endif

// ------------------------------------------------------------------------------------------------ batch seams
struct RayStream { const float *ox, *oy, *oz, *dx, *dy, *dz, *tmax; };
struct HitStream { float* t; int32_t* prim; float *nx, *ny, *nz; int32_t* tex8; };

__device__ __forceinline__ void store_hit(const HitStream& H, size_t i, const HitG& h, int B) {
  if (H.t) H.t[i] = h.hit ? h.t : -1.0f;
  if (H.prim) H.prim[i] = h.hit ? (int32_t)h.uid : -1;
  if (H.nx) H.nx[i] = h.n.x;
  if (H.ny) H.ny[i] = h.n.y;
  if (H.nz) H.nz[i] = h.n.z;
  if (H.tex8) {
    TexStack ts = h.hit ? h.tex : 0;
    for (int k = 0; k < 8; k++) { H.tex8[8 * i + k] = (int32_t)tex_head(ts, B) - 1; ts = k * B + B < 64 ? ts >> B : 0; }
  }
}
__device__ __forceinline__ Ray load_ray(const RayStream& R, size_t i) {
  Ray r;
  r.o = v3(R.ox[i], R.oy[i], R.oz[i]);
  r.d = v3(R.dx[i], R.dy[i], R.dz[i]);
  return r;
}
template <bool FAITHFUL>
__global__ void __launch_bounds__(64) k_rayint_batch_flat(DScene S, size_t n, RayStream R, HitStream H, int stack_cap, uint32_t* ovf, int ovf_cap, DCounters* c) {
  extern __shared__ uint32_t lds[];
  FlatTier<FAITHFUL, false, false, CLS_EVERY> T{S, nullptr, 0, lane_stack(lds, stack_cap, ovf, ovf_cap), Cnt()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const Ray r = load_ray(R, i);
    if (FAITHFUL || unit_length(r.d)) { store_hit(H, i, T.closest(r, R.tmax[i]), (int)S.tex_bits); continue; }
    // a caller's ray that is not unit length: the reference's own traversal (rayint_sphere reports hits for such rays that lie
    // outside the sphere's box, so the ordered early-out's pruning is not exact for them)
    HitG ch;
    Cand c = closest_flat<true, false, CLS_EVERY>(S, r, R.tmax[i], T.stk, T.cnt, true, &ch, &T.err);
    store_hit(H, i, finalize_flat<CLS_EVERY>(S, r, c, &ch), (int)S.tex_bits);
  }
  if (T.err) atomicOr(&c->error, 1u);
}
template <int DUMMY = 0>
__global__ void __launch_bounds__(64) k_shadow_batch_flat(DScene S, size_t n, RayStream R, uint8_t* occ, int stack_cap, uint32_t* ovf, int ovf_cap, DCounters* c) {
  extern __shared__ uint32_t lds[];
  FlatTier<false, false, false, CLS_EVERY> T{S, nullptr, 0, lane_stack(lds, stack_cap, ovf, ovf_cap), Cnt()};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    occ[i] = T.occluded(load_ray(R, i), R.tmax[i]) ? 1 : 0;
  if (T.err) atomicOr(&c->error, 1u);
}
#if GLOME_IN_PART(8)
__global__ void __launch_bounds__(64, GLOME_GENERIC_LB) k_rayint_batch_generic(DScene, size_t n, RayStream R, HitStream H, DCounters* c) {
  const DScene& S = kernel_args<DScene>();
  Cnt cnt; unsigned int err = 0; uint32_t vm[kVmWords];
  LaneStack nopk{}; nopk.cap = 0; nopk.ovf_cap = 0;  // (the ray-batch seams walk lane by lane)
  GenericTier T{S, nullptr, 0, cnt, err, vm, nopk};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    store_hit(H, i, T.closest(load_ray(R, i), R.tmax[i]), (int)S.tex_bits);
  if (T.err) atomicOr(&c->error, 1u);
}
__global__ void __launch_bounds__(64, GLOME_GENERIC_LB) k_shadow_batch_generic(DScene, size_t n, RayStream R, uint8_t* occ, DCounters* c) {
  const DScene& S = kernel_args<DScene>();
  Cnt cnt; unsigned int err = 0; uint32_t vm[kVmWords];
  LaneStack nopk{}; nopk.cap = 0; nopk.ovf_cap = 0;  // (the ray-batch seams walk lane by lane)
  GenericTier T{S, nullptr, 0, cnt, err, vm, nopk};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    occ[i] = T.occluded(load_ray(R, i), R.tmax[i]) ? 1 : 0;
  if (T.err) atomicOr(&c->error, 1u);
}
__global__ void __launch_bounds__(64, GLOME_GENERIC_LB) k_inside_batch(DScene, size_t n, const float* px, const float* py, const float* pz, uint8_t* in, DCounters* c) {
  const DScene& S = kernel_args<DScene>();
  unsigned int err = 0;
  uint32_t vm[kVmWords];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    in[i] = vm_inside(S, err, vm, 0, ldu4(S.recs, S.root_rec), v3(px[i], py[i], pz[i])) ? 1 : 0;
  if (err) atomicOr(&c->error, 1u);
}
#endif

// ------------------------------------------------------------------------------------------------ tile transport
#if GLOME_IN_PART(0)
__global__ void k_tiles_pack(const DTile* tiles, int ntiles, int width, const float* frame, float* payload) {
  for (int t = blockIdx.y; t < ntiles; t += gridDim.y) {
    DTile T = tiles[t];
    int np = T.w * T.h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < np * 5; i += gridDim.x * blockDim.x) {
      int p = i / 5, k = i - p * 5;
      size_t src = ((size_t)(T.y + p / T.w) * width + (T.x + p % T.w)) * 5 + k;
      payload[(size_t)T.pix_base * 5 + i] = frame[src];
    }
  }
}
__global__ void k_tiles_blit(const DTile* tiles, int ntiles, int width, const float* payload, float* frame, uint32_t* packed) {
  for (int t = blockIdx.y; t < ntiles; t += gridDim.y) {
    DTile T = tiles[t];
    int np = T.w * T.h;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < np; p += gridDim.x * blockDim.x) {
      const float* s = payload + ((size_t)T.pix_base + p) * 5;
      size_t o = (size_t)(T.y + p / T.w) * width + (T.x + p % T.w);
      float r = s[0], g = s[1], b = s[2], a = s[3], d = s[4];
      float* dst = frame + o * 5;
      dst[0] = r; dst[1] = g; dst[2] = b; dst[3] = a; dst[4] = d;
      if (packed) packed[o] = rgbf(r * a, g * a, b * a);
    }
  }
}
// blockIdx.z = frame of a group: frame f's payload sits f * payload_frame_stride words into every rank's slab, its
// framebuffer f * out_frame_stride words after the first
__global__ void k_tiles_blit_packed(const DTile* tiles, int ntiles, int width, const uint32_t* payload, uint32_t* packed, size_t payload_frame_stride, size_t out_frame_stride) {
  payload += (size_t)blockIdx.z * payload_frame_stride;
  packed += (size_t)blockIdx.z * out_frame_stride;
  for (int t = blockIdx.y; t < ntiles; t += gridDim.y) {
    DTile T = tiles[t];
    int np = T.w * T.h;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < np; p += gridDim.x * blockDim.x)
      packed[(size_t)(T.y + p / T.w) * width + (T.x + p % T.w)] = payload[(size_t)T.pix_base + p];
  }
}

#endif

// ------------------------------------------------------------------------------------------------ kernel instances by part
// Every instance the host runtime can ask for, listed once; the part that holds an instance defines the launcher that knows it.
#ifndef GLOME_CSG_LB
#define GLOME_CSG_LB 2  // waves per SIMD of the (CSG | primitives) instances
#endif
#ifndef GLOME_FLAG_LB
#define GLOME_FLAG_LB 6  // waves per SIMD of the flagship instance (two stack rows, every ray a packet)
#endif
struct FlatLaunch { int grid; size_t lds; hipStream_t st; int stack_cap; uint32_t* ovf; int ovf_cap; };
constexpr int render_flat_key(bool F, bool C, bool U, int CLS, int LB, bool TWO) { return (F ? 1 : 0) | (C ? 2 : 0) | (U ? 4 : 0) | (TWO ? 8 : 0) | (LB << 4) | (CLS << 8); }
constexpr int ss_flat_key(bool U, int CLS, int LB, bool TWO, bool F) { return (F ? 1 : 0) | (U ? 4 : 0) | (TWO ? 8 : 0) | (LB << 4) | (CLS << 8); }
// k_render_flat<FAITHFUL, COUNT, FULL, CLS, LB, TWO_ROWS>
#define GLOME_RENDER_FLAT_P1(X) /* production, lean */                                                                      \
  X(false, false, false, CLS_BIH_TRI, 1, false) X(false, false, false, (CLS_BIH_SPHERE | CLS_PRIMS), 1, false) X(false, false, false, CLS_MESH, 1, false) \
  X(false, false, false, CLS_ALL, 1, false) X(false, false, false, CLS_BIH_TRI, GLOME_FLAG_LB, true)
#define GLOME_RENDER_FLAT_P2(X) /* production, full (secondary rays, nested materials) */                                   \
  X(false, false, true, CLS_BIH_TRI, 1, false) X(false, false, true, (CLS_BIH_SPHERE | CLS_PRIMS), 1, false) X(false, false, true, CLS_MESH, 1, false) \
  X(false, false, true, CLS_ALL, 1, false)
#define GLOME_RENDER_FLAT_P3(X) /* the CSG class (two waves per SIMD: S4 0.72 -> 0.52 ms; three spill).  Round 3: an instance of its own for scenes of CSG items and plain primitives only -- 223 / 256 registers with 0 / 17 spills where the every-class one spills 22 / 39: S4 0.455 -> 0.393 ms */                    \
  X(false, false, false, CLS_EVERY, 2, false) X(false, false, true, CLS_EVERY, 2, false)                                     \
  X(false, false, false, (CLS_CSG | CLS_PRIMS), GLOME_CSG_LB, false) X(false, false, true, (CLS_CSG | CLS_PRIMS), GLOME_CSG_LB, false) /* CSG items and plain primitives only (S4) */
#define GLOME_RENDER_FLAT_P4(X) /* faithful / counting */                                                                   \
  X(true, true, true, CLS_EVERY, 1, false) X(true, true, false, CLS_EVERY, 1, false) X(false, true, true, CLS_EVERY, 1, false) X(false, true, false, CLS_EVERY, 1, false)
// k_ss_frame_flat<FULL, CLS, LB, TWO_ROWS, FAITHFUL>
#define GLOME_SS_FLAT_P5(X) X(false, CLS_BIH_TRI, 5, true, false) X(false, CLS_BIH_TRI, 4, true, false) X(false, CLS_BIH_TRI, 1, false, false) X(true, CLS_BIH_TRI, 1, false, false)
#define GLOME_SS_FLAT_P9(X) X(true, CLS_EVERY, 1, false, true) X(false, CLS_EVERY, 2, false, false) X(true, CLS_EVERY, 2, false, false) \
  X(false, (CLS_CSG | CLS_PRIMS), 2, false, false) X(true, (CLS_CSG | CLS_PRIMS), 2, false, false)
constexpr int kParts = 12;

#define GLOME_TRY_RENDER_FLAT(F, C, U, K, B, T)                                                                                                   \
  if (key == render_flat_key(F, C, U, K, B, T)) {                                                                                                 \
    hipLaunchKernelGGL((k_render_flat<F, C, U, K, B, T>), dim3(L.grid), dim3(64), L.lds, L.st, A, L.stack_cap, L.ovf, L.ovf_cap);                 \
    return true;                                                                                                                                  \
  }
#define GLOME_TRY_SS_FLAT(U, K, B, T, F)                                                                                                          \
  if (key == ss_flat_key(U, K, B, T, F)) {                                                                                                        \
    hipLaunchKernelGGL((k_ss_frame_flat<U, K, B, T, F>), dim3(L.grid), dim3(64), L.lds, L.st, A, L.stack_cap, L.ovf, L.ovf_cap);                  \
    return true;                                                                                                                                  \
  }
bool launch_flat_p1(int key, const FlatLaunch& L, const DRenderArgs& A);
bool launch_flat_p2(int key, const FlatLaunch& L, const DRenderArgs& A);
bool launch_flat_p3(int key, const FlatLaunch& L, const DRenderArgs& A);
bool launch_flat_p4(int key, const FlatLaunch& L, const DRenderArgs& A);
bool launch_ss_flat_p5(int key, const FlatLaunch& L, const DRenderArgs& A);
bool launch_ss_flat_p9(int key, const FlatLaunch& L, const DRenderArgs& A);
void launch_render_generic(int grid, hipStream_t st, const DRenderArgs& A);        // counts bih_nodes / prim_tests (part 6)
void launch_ss_generic(int grid, hipStream_t st, const DRenderArgs& A);            // (part 7)
void launch_render_generic_lean(int grid, hipStream_t st, const DRenderArgs& A);   // does not (part 10)
void launch_ss_generic_lean(int grid, hipStream_t st, const DRenderArgs& A);       // (part 11)
void launch_rayint_batch_flat(const FlatLaunch& L, DScene S, size_t n, RayStream R, HitStream H, DCounters* c);
void launch_shadow_batch_flat(const FlatLaunch& L, DScene S, size_t n, RayStream R, uint8_t* occ, DCounters* c);
void launch_rayint_batch_generic(int grid, hipStream_t st, DScene S, size_t n, RayStream R, HitStream H, DCounters* c);
void launch_shadow_batch_generic(int grid, hipStream_t st, DScene S, size_t n, RayStream R, uint8_t* occ, DCounters* c);
void launch_inside_batch(int grid, hipStream_t st, DScene S, size_t n, const float* px, const float* py, const float* pz, uint8_t* in, DCounters* c);
#if GLOME_IN_PART(1)
bool launch_flat_p1(int key, const FlatLaunch& L, const DRenderArgs& A) { GLOME_RENDER_FLAT_P1(GLOME_TRY_RENDER_FLAT) return false; }
#endif
#if GLOME_IN_PART(2)
bool launch_flat_p2(int key, const FlatLaunch& L, const DRenderArgs& A) { GLOME_RENDER_FLAT_P2(GLOME_TRY_RENDER_FLAT) return false; }
#endif
#if GLOME_IN_PART(3)
bool launch_flat_p3(int key, const FlatLaunch& L, const DRenderArgs& A) { GLOME_RENDER_FLAT_P3(GLOME_TRY_RENDER_FLAT) return false; }
#endif
#if GLOME_IN_PART(4)
bool launch_flat_p4(int key, const FlatLaunch& L, const DRenderArgs& A) { GLOME_RENDER_FLAT_P4(GLOME_TRY_RENDER_FLAT) return false; }
#endif
#if GLOME_IN_PART(5)
bool launch_ss_flat_p5(int key, const FlatLaunch& L, const DRenderArgs& A) { GLOME_SS_FLAT_P5(GLOME_TRY_SS_FLAT) return false; }
void launch_rayint_batch_flat(const FlatLaunch& L, DScene S, size_t n, RayStream R, HitStream H, DCounters* c) {
  hipLaunchKernelGGL((k_rayint_batch_flat<false>), dim3(L.grid), dim3(64), L.lds, L.st, S, n, R, H, L.stack_cap, L.ovf, L.ovf_cap, c);
}
void launch_shadow_batch_flat(const FlatLaunch& L, DScene S, size_t n, RayStream R, uint8_t* occ, DCounters* c) {
  hipLaunchKernelGGL((k_shadow_batch_flat<0>), dim3(L.grid), dim3(64), L.lds, L.st, S, n, R, occ, L.stack_cap, L.ovf, L.ovf_cap, c);
}
#endif
#if GLOME_IN_PART(9)
bool launch_ss_flat_p9(int key, const FlatLaunch& L, const DRenderArgs& A) { GLOME_SS_FLAT_P9(GLOME_TRY_SS_FLAT) return false; }
#endif
#if GLOME_IN_PART(6)
void launch_render_generic(int grid, hipStream_t st, const DRenderArgs& A) { hipLaunchKernelGGL(k_render_generic<true>, dim3(grid), dim3(64), flat_lds_bytes((int)A.S.pk_generic_cap), st, A); }
#endif
#if GLOME_IN_PART(7)
void launch_ss_generic(int grid, hipStream_t st, const DRenderArgs& A) { hipLaunchKernelGGL(k_ss_frame_generic<true>, dim3(grid), dim3(64), flat_lds_bytes((int)A.S.pk_generic_cap), st, A); }
#endif
#if GLOME_IN_PART(10)
void launch_render_generic_lean(int grid, hipStream_t st, const DRenderArgs& A) { hipLaunchKernelGGL(k_render_generic<false>, dim3(grid), dim3(64), flat_lds_bytes((int)A.S.pk_generic_cap), st, A); }
#endif
#if GLOME_IN_PART(11)
void launch_ss_generic_lean(int grid, hipStream_t st, const DRenderArgs& A) { hipLaunchKernelGGL(k_ss_frame_generic<false>, dim3(grid), dim3(64), flat_lds_bytes((int)A.S.pk_generic_cap), st, A); }
#endif
#if GLOME_IN_PART(8)
void launch_rayint_batch_generic(int grid, hipStream_t st, DScene S, size_t n, RayStream R, HitStream H, DCounters* c) { hipLaunchKernelGGL(k_rayint_batch_generic, dim3(grid), dim3(64), 0, st, S, n, R, H, c); }
void launch_shadow_batch_generic(int grid, hipStream_t st, DScene S, size_t n, RayStream R, uint8_t* occ, DCounters* c) { hipLaunchKernelGGL(k_shadow_batch_generic, dim3(grid), dim3(64), 0, st, S, n, R, occ, c); }
void launch_inside_batch(int grid, hipStream_t st, DScene S, size_t n, const float* px, const float* py, const float* pz, uint8_t* in, DCounters* c) {
  hipLaunchKernelGGL(k_inside_batch, dim3(grid), dim3(64), 0, st, S, n, px, py, pz, in, c);
}
#endif

#if GLOME_IN_PART(0)
// ================================================================================================ host runtime
static bool launch_render_flat(int key, const FlatLaunch& L, const DRenderArgs& A) {
  return launch_flat_p1(key, L, A) || launch_flat_p2(key, L, A) || launch_flat_p3(key, L, A) || launch_flat_p4(key, L, A);
}
static bool launch_ss_flat(int key, const FlatLaunch& L, const DRenderArgs& A) { return launch_ss_flat_p5(key, L, A) || launch_ss_flat_p9(key, L, A); }
struct glome_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  // event pool: while timing is on, every render launch records its own (start, stop) pair
  std::vector<hipEvent_t> pool;
  int pool_used = 0;
  bool timing = false;
  int timing_stride = 1, timing_seen = 0;  // every timing_stride-th launch is timed
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipDeviceProp_t prop;
  // Per-slot launch state, so several frames can be in flight on different streams (their work queues, counters and
  // workspaces must not be shared): slot 0 is the default.
  struct Slot {
    DCounters* d_counters = nullptr;
    uint32_t* d_ovf = nullptr;   // traversal-stack overflow workspace (grown on demand)
    size_t ovf_bytes = 0;
    float* d_scratch = nullptr;  // adaptive sampler working buffer
    size_t scratch_bytes = 0;
    bool launched = false;  // a launch went out on this slot since its error word was last polled
    hipStream_t launched_on = nullptr;  // ... on this stream (a caller's own stream is the caller's to synchronise)
  };
  static constexpr int kSlots = 8;
  Slot slots[kSlots];
  int cur = 0;
  Slot& slot() { return slots[cur]; }
  std::string err;
  int grid_per_cu = 0;  // 0: persistent grids sized by work (tuned for several launches in flight); > 0: this many waves per CU, resources permitting
  // tile tables cached per (w, h, blocksize, first, stride)
  // lut[w >> 6] = the tile that holds work item (w & ~63): the kernel's item -> tile lookup is one table read and a step or two
  struct TileTable { std::vector<DTile> host; DTile* dev = nullptr; uint32_t* lut = nullptr; uint32_t total_waves = 0; int64_t pixels = 0; };
  std::map<std::vector<int>, TileTable> tile_cache;
};
struct glome_scene {
  glome_ctx* ctx = nullptr;
  int ovf_cap = 0;  // stack entries per lane beyond the LDS part
  DScene dev{};
  std::vector<void*> allocs;
  glome_scene_info info{};
  int stack_cap = 8;
  bool has_secondary_mats = false, has_nested_mats = false;
  bool has_refract = false;  // a Refract material: its transmitted rays are not unit length (Shader.hs:141) -- see launch_render
  int cls_mask = CLS_ALL;  // which entry classes the flat root program contains
  bool pk_all = false;     // every triangle BIH of the scene has the hand-written walk's node form (flatten.hpp emit_bih)
};

static std::string g_global_error;
#define HIPCHK(ctx, call)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
      return GLOME_E_HIP;                                                                          \
    }                                                                                              \
  } while (0)

static int check_params(glome_ctx* ctx, const glome_render_params* P) {
  if (!P || P->width <= 0 || P->height <= 0 || P->blocksize <= 0 || P->tile_stride <= 0 || P->tile_first < 0 || P->rank0_share_pct < 0 || P->rank0_share_pct > 100) { ctx->err = "bad render params"; return GLOME_E_INVALID; }
  if ((int64_t)P->width * P->height > (1ll << 30)) { ctx->err = "frame too large"; return GLOME_E_INVALID; }
  if (P->maxdepth < 1 || P->maxdepth > kMaxTraceDepth) { ctx->err = "maxdepth must be in 1.." + std::to_string(kMaxTraceDepth); return GLOME_E_LIMIT; }
  return 0;
}
static int get_tiles(glome_ctx* ctx, const glome_render_params* P, int first, int stride, glome_ctx::TileTable** out, int blocksize = 0) {
  if (!blocksize) blocksize = P->blocksize;
  std::vector<int> key{P->width, P->height, blocksize, first, stride, P->rank0_share_pct};
  auto it = ctx->tile_cache.find(key);
  if (it == ctx->tile_cache.end()) {
    glome_ctx::TileTable tt;
    owned_tiles(P->width, P->height, blocksize, first, stride, P->rank0_share_pct, tt.host, tt.total_waves, tt.pixels);
    if (const char* e = getenv("GLOME_DEBUG_TILE_ORDER")) {  // (experiment: the order in which a launch works through its tiles; pixels do not move)
      const std::string how = e;
      if (how == "bottomup") std::stable_sort(tt.host.begin(), tt.host.end(), [](const DTile& a, const DTile& b) { return a.y > b.y; });
      else if (how == "rowmajor") std::stable_sort(tt.host.begin(), tt.host.end(), [](const DTile& a, const DTile& b) { return a.y < b.y; });
      else if (how == "reverse") std::reverse(tt.host.begin(), tt.host.end());
      uint32_t wb = 0;
      for (auto& t : tt.host) { t.wave_base = wb; wb += tile_waves(t.w, t.h); }
    }
    size_t bytes = std::max<size_t>(1, tt.host.size()) * sizeof(DTile);
    HIPCHK(ctx, hipMalloc((void**)&tt.dev, bytes));
    if (!tt.host.empty()) HIPCHK(ctx, hipMemcpy(tt.dev, tt.host.data(), tt.host.size() * sizeof(DTile), hipMemcpyHostToDevice));
    std::vector<uint32_t> lut((tt.total_waves >> 6) + 1, 0);
    for (size_t k = 0, t = 0; k < lut.size(); k++) {
      while (t + 1 < tt.host.size() && tt.host[t + 1].wave_base <= (uint32_t)(k << 6)) t++;
      lut[k] = (uint32_t)t;
    }
    HIPCHK(ctx, hipMalloc((void**)&tt.lut, lut.size() * sizeof(uint32_t)));
    HIPCHK(ctx, hipMemcpy(tt.lut, lut.data(), lut.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    it = ctx->tile_cache.emplace(key, std::move(tt)).first;
  }
  *out = &it->second;
  return 0;
}

template <class T> static int upload(glome_scene* s, const std::vector<T>& v, const T** out) {
  glome_ctx* ctx = s->ctx;
  void* d = nullptr;
  size_t bytes = v.size() * sizeof(T);
  HIPCHK(ctx, hipMalloc(&d, bytes));
  s->allocs.push_back(d);
  HIPCHK(ctx, hipMemcpy(d, v.data(), bytes, hipMemcpyHostToDevice));
  s->info.device_bytes += (int64_t)bytes;
  *out = (const T*)d;
  return 0;
}

const char* glome_global_error(void) { return g_global_error.c_str(); }

glome_ctx* glome_ctx_create(int device_ordinal) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) { g_global_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"); return nullptr; }
  if (device_ordinal < 0 || device_ordinal >= n) { g_global_error = "device ordinal out of range"; return nullptr; }
  glome_ctx* c = new glome_ctx();
  c->device = device_ordinal;
  auto fail = [&](const char* what, hipError_t err) { g_global_error = std::string(what) + ": " + hipGetErrorString(err); delete c; return (glome_ctx*)nullptr; };
  if ((e = hipSetDevice(device_ordinal)) != hipSuccess) return fail("hipSetDevice", e);
  if ((e = hipGetDeviceProperties(&c->prop, device_ordinal)) != hipSuccess) return fail("hipGetDeviceProperties", e);
  if (std::string(c->prop.gcnArchName).rfind("gfx950", 0) != 0) {
    g_global_error = std::string("device is ") + c->prop.gcnArchName + ", this library is built for gfx950 only";
    delete c;
    return nullptr;
  }
  // The context's own stream is a BLOCKING stream: work a caller put on the device's default stream (filling or allocating
  // the very buffers it hands to a *_dev entry point) is ordered before this context's launches and after them, as with any
  // HIP code that never names a stream.  A caller that wants overlap brings its own streams (glome_ctx_use_slot).
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamDefault)) != hipSuccess) return fail("hipStreamCreate", e);
  c->own_stream = c->stream;
  if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return fail("hipEventCreate", e);
  for (int k = 0; k < glome_ctx::kSlots; k++)
    if ((e = hipMalloc((void**)&c->slots[k].d_counters, sizeof(DCounters))) != hipSuccess || (e = hipMemset(c->slots[k].d_counters, 0, sizeof(DCounters))) != hipSuccess) return fail("hipMalloc", e);
#ifdef GLOME_PROBE
  for (int k = 0; k < glome_ctx::kSlots; k++)
    for (int q : {8, 10, 12}) (void)hipMemset(&c->slots[k].d_counters->dbg[q], 0xff, sizeof(unsigned long long));  // (the timeline's minima, render_loop)
#endif
  {  // kernel_args<>()'s assumption about the kernarg segment, checked once per process on the first context
    static std::once_flag once;
    static bool good = true;
    std::call_once(once, [&] {
      DRenderArgs* h = new DRenderArgs;
      unsigned char* hb = (unsigned char*)h;
      for (size_t i = 0; i < sizeof(DRenderArgs); i++) hb[i] = (unsigned char)(i * 131u + 7u);
      DRenderArgs* d = nullptr; unsigned int* ok = nullptr; unsigned int one = 1, got = 0;
      if (hipMalloc((void**)&d, sizeof(DRenderArgs)) == hipSuccess && hipMalloc((void**)&ok, sizeof(unsigned int)) == hipSuccess &&
          hipMemcpy(d, h, sizeof(DRenderArgs), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(ok, &one, sizeof(one), hipMemcpyHostToDevice) == hipSuccess) {
        hipLaunchKernelGGL(k_kernarg_selftest, dim3(1), dim3(256), 0, c->stream, *h, d, ok);
        if (hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(&got, ok, sizeof(got), hipMemcpyDeviceToHost) == hipSuccess) good = got == 1u;
        else good = false;
      } else good = false;
      if (d) (void)hipFree(d);
      if (ok) (void)hipFree(ok);
      delete h;
    });
    if (!good) { g_global_error = "kernel-argument self-test failed: the first by-value kernel argument is not at offset 0 of the kernarg segment in host layout (kernel_args<>)"; glome_ctx_destroy(c); return nullptr; }
  }
  return c;
}
void glome_ctx_destroy(glome_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (auto& kv : c->tile_cache) { if (kv.second.dev) (void)hipFree(kv.second.dev); if (kv.second.lut) (void)hipFree(kv.second.lut); }
  for (auto& sl : c->slots) {
    if (sl.d_counters) (void)hipFree(sl.d_counters);
    if (sl.d_ovf) (void)hipFree(sl.d_ovf);
    if (sl.d_scratch) (void)hipFree(sl.d_scratch);
  }
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (hipEvent_t ev : c->pool) (void)hipEventDestroy(ev);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}
const char* glome_last_error(const glome_ctx* c) { return c ? c->err.c_str() : g_global_error.c_str(); }
void* glome_ctx_stream(glome_ctx* c) { return c ? (void*)c->stream : nullptr; }
int glome_ctx_use_stream(glome_ctx* c, void* stream) {
  if (!c) return GLOME_E_INVALID;
  c->stream = stream ? (hipStream_t)stream : c->own_stream;
  return 0;
}
int glome_ctx_use_slot(glome_ctx* c, void* stream, int slot) {
  if (!c || slot < 0 || slot >= glome_ctx::kSlots) return GLOME_E_INVALID;
  c->stream = stream ? (hipStream_t)stream : c->own_stream;
  c->cur = slot;
  // a slot rebound to another stream forgets the one its last launch went to: that handle is the caller's and may be destroyed by
  // now (glome_ctx_synchronize must not query it); the slot's error word is then read without asking whether the old stream is idle
  // -- a word a running kernel ORs into later is reported by the next synchronize
  glome_ctx::Slot& sl = c->slots[slot];
  if (sl.launched && sl.launched_on != c->stream) sl.launched_on = nullptr;
  return 0;
}
int glome_ctx_set_grid_per_cu(glome_ctx* c, int waves_per_cu) {
  if (!c || waves_per_cu < 0 || waves_per_cu > 32) return GLOME_E_INVALID;
  c->grid_per_cu = waves_per_cu;
  return 0;
}
int glome_ctx_timing_begin(glome_ctx* c, int max_launches) {
  if (!c || max_launches <= 0) return GLOME_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  while ((int)c->pool.size() < 2 * max_launches) {
    hipEvent_t ev;
    HIPCHK(c, hipEventCreate(&ev));
    c->pool.push_back(ev);
  }
  c->pool_used = 0;
  c->timing = true;
  c->timing_stride = 1; c->timing_seen = 0;
  return 0;
}
int glome_ctx_timing_begin_sampled(glome_ctx* c, int max_launches, int stride) {
  int rc = glome_ctx_timing_begin(c, max_launches);
  if (rc == 0) c->timing_stride = stride < 1 ? 1 : stride;
  return rc;
}
int glome_ctx_timing_end(glome_ctx* c, float* ms_out, int cap) {
  if (!c) return GLOME_E_INVALID;
  c->timing = false;
  int n = c->pool_used / 2;
  // the pairs were recorded on whichever lane stream launched (up to kSlots of them): wait on each stop event itself
  for (int i = 0; i < n && i < cap; i++) {
    HIPCHK(c, hipEventSynchronize(c->pool[2 * i + 1]));
    HIPCHK(c, hipEventElapsedTime(&ms_out[i], c->pool[2 * i], c->pool[2 * i + 1]));
  }
  return n;
}
static int poll_device_error(glome_ctx* ctx, glome_ctx::Slot& sl);
int glome_ctx_synchronize(glome_ctx* c) {
  if (!c) return GLOME_E_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // limits hit by launches nobody asked statistics of; lanes on the caller's own streams are the caller's to synchronise
  // first (an error raised by a launch still in flight is reported by the next call)
  int rc = 0;
  for (auto& sl : c->slots) {
    if (!sl.launched) continue;
    // a slot bound to a caller's stream (glome_ctx_use_slot) may still be running: its word is read once that stream is idle,
    // by this call or a later one -- never while a kernel could still OR into it
    if (sl.launched_on && sl.launched_on != c->stream && hipStreamQuery(sl.launched_on) == hipErrorNotReady) continue;
    sl.launched = false;
    int r = poll_device_error(c, sl);
    if (r) rc = r;
  }
  return rc;
}
int glome_ctx_debug_words(glome_ctx* c, uint64_t* out16) {  // DCounters::dbg of the current slot (measurement builds write them)
  if (!c || !out16) return GLOME_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(out16, c->slot().d_counters->dbg, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return 0;
}
// ---- a framebuffer several PROCESSES render into (one process per GPU: glome_amd/dist.py) ----
// rank 0 allocates the frames and exports a handle; the other ranks open it and hand the pointer to their render calls, whose kernels
// then store their tiles' pixels straight into rank 0's memory over xGMI (hipIpc*: dmabuf handles on this driver).
int glome_ipc_alloc(glome_ctx* c, size_t bytes, void** dev_ptr, unsigned char* handle64) {
  if (!c || !dev_ptr || !handle64 || bytes == 0) return GLOME_E_INVALID;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "a HIP IPC memory handle is 64 bytes");
  HIPCHK(c, hipSetDevice(c->device));
  void* p = nullptr;
  HIPCHK(c, hipMalloc(&p, bytes));
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, p);
  if (e != hipSuccess) { (void)hipFree(p); c->err = std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e); (void)hipGetLastError(); return GLOME_E_HIP; }
  memcpy(handle64, &h, 64);
  HIPCHK(c, hipMemset(p, 0, bytes));
  *dev_ptr = p;
  return 0;
}
int glome_ipc_open(glome_ctx* c, const unsigned char* handle64, void** dev_ptr) {
  if (!c || !dev_ptr || !handle64) return GLOME_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  hipError_t e = hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { c->err = std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e); (void)hipGetLastError(); return GLOME_E_HIP; }
  return 0;
}
int glome_ipc_close(glome_ctx* c, void* dev_ptr, int owner) {  // owner: the process that allocated frees, the others close their mapping
  if (!c || !dev_ptr) return GLOME_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  if (owner) HIPCHK(c, hipFree(dev_ptr)); else HIPCHK(c, hipIpcCloseMemHandle(dev_ptr));
  return 0;
}
int glome_ctx_debug_reset(glome_ctx* c) {  // (measurement builds: the timeline words of the current slot back to "nothing seen")
  if (!c) return GLOME_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemset(c->slot().d_counters->dbg, 0, 16 * sizeof(uint64_t)));
  for (int q : {8, 10, 12}) HIPCHK(c, hipMemset(&c->slot().d_counters->dbg[q], 0xff, sizeof(unsigned long long)));
  return 0;
}
int glome_ctx_device_info(glome_ctx* c, char* name, int cap, int* cu_count, int* warp_size) {
  if (!c) return GLOME_E_INVALID;
  if (name && cap > 0) snprintf(name, cap, "%s (%s)", c->prop.name, c->prop.gcnArchName);
  if (cu_count) *cu_count = c->prop.multiProcessorCount;
  if (warp_size) *warp_size = c->prop.warpSize;
  return 0;
}

// ---- bih, built on the device (bih_build_device.hpp) ----
int32_t glome_sb_bih_dev(glome_ctx* ctx, glome_sb* sb, const int32_t* ids, int32_t n, float* gpu_ms) {
  if (!ctx || !sb) { g_global_error = "null ctx or builder"; return GLOME_E_INVALID; }
  if (gpu_ms) *gpu_ms = 0;
  try {
    if (n < 0 || (n > 0 && !ids)) throw std::invalid_argument("bad id list");
    Graph& G = sb->graph;
    std::vector<int> v(ids, ids + n);
    if (v.empty()) return G.bih(v);  // bih [] = Void, Bih.hs:309-311
    std::vector<Box3> boxes;
    Box3 bb = box_empty();
    for (int i : v) boxes.push_back(G.bound(i));
    for (auto& b : boxes) bb = box_join(bb, b);
    if (bb.lo.x == -kInfinity || bb.lo.y == -kInfinity || bb.lo.z == -kInfinity || bb.hi.x == kInfinity || bb.hi.y == kInfinity || bb.hi.z == kInfinity)
      throw scene_error("bih: infinite bounding box");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    auto T = std::make_shared<BihTree>();
    std::string err;
    if (!bihdev::build(boxes, v, bb, *T, ctx->stream, err, gpu_ms)) { ctx->err = sb->err = err; return GLOME_E_LIMIT; }
    Node nd; nd.kind = K_BIH; nd.bih = T;
    return G.add(nd);
  } catch (const scene_error& e) { ctx->err = sb->err = e.what(); return GLOME_E_SCENE; }
  catch (const std::exception& e) { ctx->err = sb->err = e.what(); return GLOME_E_INVALID; }
}

int32_t glome_sb_mesh_dev(glome_ctx* ctx, glome_sb* sb, const double* verts, int nv, const double* norms, int nn, const int32_t* tris, int nt, const int32_t* mats, int nm,
                          float* gpu_ms) {
  if (!ctx || !sb) { g_global_error = "null ctx or builder"; return GLOME_E_INVALID; }
  if (gpu_ms) *gpu_ms = 0;
  try {
    if (nv < 0 || nn < 0 || nt < 0 || nm < 0 || (nv && !verts) || (nn && !norms) || (nt && !tris) || (nm && !mats)) throw std::invalid_argument("bad mesh arrays");
    Graph& G = sb->graph;
    std::vector<D3> V, N;
    for (int k = 0; k < nv; k++) V.push_back(D3{verts[3 * k], verts[3 * k + 1], verts[3 * k + 2]});
    for (int k = 0; k < nn; k++) N.push_back(D3{norms[3 * k], norms[3 * k + 1], norms[3 * k + 2]});
    std::vector<MeshTri> T;
    for (int k = 0; k < nt; k++) { const int32_t* t = tris + 8 * k; T.push_back(MeshTri{t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]}); }
    std::vector<int> M(mats, mats + nm);
    if (nt < 3) return G.mesh(std::move(V), std::move(N), std::move(T), std::move(M));  // a single leaf (Mesh.hs:70)
    std::vector<Box3> tbb;
    auto D = G.mesh_data(std::move(V), std::move(N), std::move(T), std::move(M), tbb);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string err;
    if (!bihdev::build_mesh(tbb, *D, ctx->stream, err, gpu_ms)) { ctx->err = sb->err = err; return GLOME_E_LIMIT; }
    return G.mesh_node(D);
  } catch (const scene_error& e) { ctx->err = sb->err = e.what(); return GLOME_E_SCENE; }
  catch (const std::exception& e) { ctx->err = sb->err = e.what(); return GLOME_E_INVALID; }
}

// ---- commit ----
glome_scene* glome_scene_commit(glome_ctx* ctx, glome_sb* sb, int32_t root) {
  if (!ctx || !sb) { g_global_error = "null ctx or builder"; return nullptr; }
  FlatScene F;
  try {
    Flattener fl(sb_graph(sb), F);
    fl.run(root);
  } catch (std::exception& e) { ctx->err = e.what(); return nullptr; }
  (void)hipSetDevice(ctx->device);
  glome_scene* s = new glome_scene();
  s->ctx = ctx;
  int rc = 0;
  DScene& D = s->dev;
  rc |= upload(s, F.recs, &D.recs);
  rc |= upload(s, F.spheres, &D.spheres); rc |= upload(s, F.tris, &D.tris); rc |= upload(s, F.tripairs, &D.tripairs); rc |= upload(s, F.trinorms, &D.trinorms);
  rc |= upload(s, F.boxes, &D.boxes); rc |= upload(s, F.planes, &D.planes); rc |= upload(s, F.discs, &D.discs);
  rc |= upload(s, F.quadrics, &D.quadrics); rc |= upload(s, F.xfms, &D.xfms);
  rc |= upload(s, F.bihhdr, &D.bihhdr); rc |= upload(s, F.bihnodes, &D.bihnodes); rc |= upload(s, F.pknodes, &D.pknodes); D.pknodes_bytes = (uint32_t)(F.pknodes.size() * sizeof(F4));
  rc |= upload(s, F.meshhdr, &D.meshhdr); rc |= upload(s, F.meshnodes, &D.meshnodes); rc |= upload(s, F.mtris, &D.mtris);
  rc |= upload(s, F.mtrimeta, &D.mtrimeta); rc |= upload(s, F.mats, &D.mats); rc |= upload(s, F.wlights, &D.wlights); rc |= upload(s, F.matkids, &D.matkids);
  rc |= upload(s, F.entries, &D.entries);
  if (rc) { glome_scene_release(s); return nullptr; }
  D.n_entries = F.tier == 0 ? (uint32_t)F.entries.size() : 0;
  D.root_rec = F.root_rec; D.tier = F.tier; D.n_mats = (uint32_t)sb_graph(sb).mats.size(); D.tex_bits = F.tex_bits;
  // the generic tier's kernels carry an LDS stack of up to kGenericPacketStack entries per lane for the packet walks of its service
  // (sphere / triangle trees, trees of items answered in place: 18 KB a wave at 24, eight waves per CU fit); a deeper tree keeps the
  // per-lane walk
  D.pk_generic_cap = (F.tier != 0 && F.max_sphere_bih_depth > 0) ? (uint32_t)std::min(kGenericPacketStack, std::max(4, F.max_sphere_bih_depth)) : 0u;
  if (getenv("GLOME_DEBUG_NO_GENERIC_PACKETS")) D.pk_generic_cap = 0;  // (debug switch: every BIH inside the interpreter walked lane by lane -- the test that holds the packet service against it)
  glome_scene_info& I = s->info;
  I.tier = (int32_t)F.tier; I.nesting_depth = F.nesting_depth;
  I.n_records = (int64_t)F.recs.size(); I.n_bih_nodes = (int64_t)F.bihnodes.size(); I.n_mesh_nodes = (int64_t)F.meshnodes.size() / 4;
  I.n_triangles = (int64_t)(F.tris.size() + F.mtris.size()) / 3; I.n_spheres = (int64_t)F.spheres.size();
  I.n_other_prims = F.n_other_prims; I.n_xfms = (int64_t)F.xfms.size() / 6; I.n_materials = D.n_mats;
  I.max_bih_depth = F.max_bih_depth; I.max_mesh_depth = F.max_mesh_depth;
  for (const Mat& m : sb_graph(sb).mats) {
    if (m.kind == MAT_REFLECT || m.kind == MAT_REFRACT || m.kind == MAT_WARP) s->has_secondary_mats = true;
    if (m.kind == MAT_REFRACT) s->has_refract = true;
    if (m.kind == MAT_LAYERS || m.kind == MAT_BLEND) s->has_nested_mats = true;
  }
  if (F.tier == 0) {
    int m = 0;
    for (const U4& e : F.entries) {
      const U4& r = F.recs[e.x];
      uint32_t k = r.x & RF_KINDMASK;
      if (k == R_BIH) { uint32_t c; memcpy(&c, &F.bihhdr[3 * r.y + 1].w, 4); m |= c == BC_TRI ? CLS_BIH_TRI : (c == BC_SPHERE ? CLS_BIH_SPHERE : (c == BC_CSG ? CLS_CSG : CLS_BIH_SIMPLE)); }
      else if (k == R_MESH) m |= CLS_MESH;
      else if (k > R_CONE) m |= CLS_CSG;  // a Difference / Intersection / Instance over primitives in the root list
      else if (k != R_VOID) m |= CLS_PRIMS;
    }
    s->cls_mask = m;
  }
  s->pk_all = F.pk_all;
  int need = std::max(F.max_bih_depth, F.max_mesh_depth);
  // LDS holds up to kLdsStack entries per lane (LDS per wave bounds occupancy); a deeper tree keeps its correctness
  // through the global overflow columns.
  constexpr int kLdsStack = kAsmLdsCap;
  int lds_cap = getenv("GLOME_DEBUG_STACK_CAP") ? atoi(getenv("GLOME_DEBUG_STACK_CAP")) : kLdsStack;
  int total = std::min(kFlatStack, std::max(4, need));
  if (F.tier == 0 && F.max_mesh_depth > 0) total = std::max(total, std::min(kFlatStackMesh, 2 * F.max_mesh_depth));  // (the Mesh packet walk: up to two entries per level)
  s->stack_cap = std::max(4, std::min(lds_cap, total));
  s->ovf_cap = std::max(0, total - s->stack_cap);
  return s;
}
void glome_scene_release(glome_scene* s) {
  if (!s) return;
  (void)hipSetDevice(s->ctx->device);
  for (void* p : s->allocs) (void)hipFree(p);
  delete s;
}
int glome_scene_get_info(const glome_scene* s, glome_scene_info* out) {
  if (!s || !out) return GLOME_E_INVALID;
  *out = s->info;
  return 0;
}

// ---- launch helpers ----
// Waves of a persistent launch.  At most what the CU can hold (LDS, register budget); fewer when the launch is small: a wave
// should get ~64 work items, so that its fixed costs (set-up, the counter flush) amortise and several launches in flight
// share the CUs side by side instead of one after the other (measured on the flagship frame, 4 launches of 4 frames in
// flight: 24 waves per CU 0.272 ms, 16: 0.249, 8: 0.239; the 4K / 1M-triangle frame, 4x the items, is best at 24).
// Is anything still running on the streams the context's OTHER slots last launched on?  A launch sized by its work (below) leaves room
// for the launches beside it; one that has the GPU to itself -- a short multi-GPU run is ONE launch of a rank's shard, nothing beside it --
// wants every wave slot: twenty frames of an eighth of the flagship's tiles 0.90 -> 0.66 ms (profiles/r04_probes/short_run_shards.txt).
static bool other_slots_busy(glome_ctx* ctx) {
  for (int k = 0; k < glome_ctx::kSlots; k++) {
    const glome_ctx::Slot& sl = ctx->slots[k];
    if (k == ctx->cur || !sl.launched_on || sl.launched_on == ctx->stream) continue;
    if (hipStreamQuery(sl.launched_on) == hipErrorNotReady) return true;
  }
  (void)hipGetLastError();  // (a query of a finished stream leaves nothing behind; one of a stream its owner has destroyed must not be this call's error)
  return false;
}
static int persistent_grid(glome_ctx* ctx, size_t lds_per_block, uint32_t total_work, int max_per_cu = 32, int min_per_cu = 0) {
  int cus = ctx->prop.multiProcessorCount;
  int per_cu = max_per_cu;  // wave slots per CU the kernel's register budget allows
  if (lds_per_block) per_cu = std::min<int>(per_cu, (int)(160 * 1024 / lds_per_block));
  per_cu = std::max(per_cu, 1);
  if (ctx->grid_per_cu > 0) per_cu = std::min(per_cu, ctx->grid_per_cu);  // glome_ctx_set_grid_per_cu: the caller knows what else runs
  else if (min_per_cu > 0) {  // sized by work: ~64 items per wave (~16 when nothing runs beside the launch), not below min_per_cu waves per CU
    const bool alone = !other_slots_busy(ctx);
    const long per_wave = alone ? 16L : 64L;
    // ... and a large scene's launch that is alone not below 12: one flagship frame takes 0.45 ms with 8 waves per CU, 0.40 with 12, 0.41 with
    // 16, 0.51 with 24; two frames want 16, four and more all 24 (profiles/r04_probes/lone_launch_grid.txt)
    if (alone && min_per_cu >= 8) min_per_cu = 12;
    long want = ((long)total_work + per_wave * cus - 1) / (per_wave * cus);
    per_cu = (int)std::min<long>(per_cu, std::max<long>(min_per_cu, want));
  }
  long g = (long)cus * per_cu;
  return (int)std::max<long>(1, std::min<long>(g, total_work));
}
// The floor of a work-sized grid: a scene with a small tree has cheap work items (tens of traversal steps), and a wave's
// fixed costs then dominate a short launch -- 3 waves per CU (S2, four 720x480 frames per launch: 0.024 ms per frame
// against 0.036 with 8); items of a large scene keep a wave busy for ~0.1 ms each and a short launch wants 8 (a rank's
// shard of the flagship frame: 0.039 ms per frame with 8, 0.048 with 4).
static int grid_floor(const glome_scene* s) { return s->info.n_bih_nodes + s->info.n_mesh_nodes < 4096 ? 3 : 8; }
// per wave slot: ovf_cap overflow entries + one more block of [3][64] words, the dump block of bih_walk_asm (LaneStack::dump)
static int ensure_overflow(glome_ctx* ctx, int grid, int waves_per_block, int ovf_cap) {
  size_t need = (size_t)grid * waves_per_block * (ovf_cap + 1) * 3 * 64 * sizeof(uint32_t);
  glome_ctx::Slot& sl = ctx->slot();
  if (need <= sl.ovf_bytes) return 0;
  if (sl.d_ovf) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(sl.d_ovf)); sl.d_ovf = nullptr; sl.ovf_bytes = 0; }
  HIPCHK(ctx, hipMalloc((void**)&sl.d_ovf, need));
  sl.ovf_bytes = need;
  return 0;
}
static int ensure_scratch(glome_ctx* ctx, size_t need) {
  glome_ctx::Slot& sl = ctx->slot();
  if (need <= sl.scratch_bytes) return 0;
  if (sl.d_scratch) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, hipFree(sl.d_scratch)); sl.d_scratch = nullptr; sl.scratch_bytes = 0; }
  HIPCHK(ctx, hipMalloc((void**)&sl.d_scratch, need));
  sl.scratch_bytes = need;
  return 0;
}
// The device-side error word is sticky: kernels only ever OR into it, the counter reset leaves it alone, and it is read --
// and cleared -- where the host waits anyway (statistics, the host-buffer seams, glome_ctx_synchronize).  So a launch
// that nobody asked statistics of (the pipelined frame path) still reports a CSG-advance or frame-pool limit, at the
// next synchronize.  The caller has synchronised the slot's stream.
static int poll_device_error(glome_ctx* ctx, glome_ctx::Slot& sl) {
  unsigned int e = 0;
  unsigned int* d = &sl.d_counters->error;
  HIPCHK(ctx, hipMemcpy(&e, d, sizeof(e), hipMemcpyDeviceToHost));
  if (!e) return 0;
  HIPCHK(ctx, hipMemset(d, 0, sizeof(e)));
  ctx->err = "device-side limit hit (traversal stack or CSG advance cap)";
  return GLOME_E_LIMIT;
}
static int check_device_error(glome_ctx* ctx) { return poll_device_error(ctx, ctx->slot()); }

static int reset_counters(glome_ctx* ctx) {
  HIPCHK(ctx, hipMemsetAsync(ctx->slot().d_counters, 0, offsetof(DCounters, error), ctx->stream));  // not the sticky error word
  return 0;
}

static int scene_class(const glome_scene* s) {  // scene class -> the smallest kernel instance that covers it (SPECIALIZE analogue, Bih.hs:370-374)
  int m = s->cls_mask;
  if (m & CLS_CSG) return (m & ~(CLS_CSG | CLS_PRIMS)) == 0 ? (CLS_CSG | CLS_PRIMS) : CLS_EVERY;
  return (m & ~CLS_BIH_TRI) == 0 ? CLS_BIH_TRI : ((m & ~(CLS_BIH_SPHERE | CLS_PRIMS)) == 0 ? (CLS_BIH_SPHERE | CLS_PRIMS) : ((m & ~CLS_MESH) == 0 ? CLS_MESH : CLS_ALL));
}
// every ray of the frame is walked as a packet (see k_render_flat): two stack rows per entry, six waves per SIMD
static bool use_two_rows(const glome_scene* s, const glome_render_params* P, uint32_t items = 0xffffffffu) {
  // a small launch of a rank's shard shares the GPU with the collective's and the blit's kernels and is better off with the
  // 16-wave instance (measured at 8 ranks: 0.040 against 0.047 ms per frame); from ~48k work items on the 24-wave one wins
  if (P->tile_stride != 1 && items < 48000u) return false;
  if (s->dev.tier != 0 || P->faithful || P->count_work) return false;
  if (s->has_secondary_mats || s->has_nested_mats || s->stack_cap != kAsmLdsCap || !s->pk_all) return false;
  return scene_class(s) == CLS_BIH_TRI;  // (what bih_walk_asm walks: a two-row kernel has no row for bih_tri_packet's references)
}

static bool launch_render(glome_scene* s, const DRenderArgs& A, const glome_render_params* P, int grid, size_t lds) {
  hipStream_t st = s->ctx->stream;
  bool faithful = P->faithful != 0, count = P->count_work != 0 || faithful;
  // A scene with a Refract material, traced deeper than the primary ray: the transmitted rays are not unit length
  // (Shader.hs:141), and for those rayint_sphere (Sphere.hs:20-41) reports hits outside the sphere's box -- the ordered
  // early-out's pruning is exact only for unit rays, so such a frame is traversed as the reference traverses (the flat
  // tier's faithful instance; the generic tier switches per ray, rt_generic.hpp).
  if (s->dev.tier == 0 && s->has_refract && P->maxdepth > 1) faithful = count = true;
  if (s->dev.tier != 0) { if (P->count_work) launch_render_generic(grid, st, A); else launch_render_generic_lean(grid, st, A); return true; }
  // lean kernel: legal when no secondary trace can do work and no material nests (Blend / AdditiveLayers)
  const bool full = s->has_nested_mats || (s->has_secondary_mats && P->maxdepth > 1);
  const FlatLaunch L{grid, lds, st, s->stack_cap, s->ctx->slot().d_ovf, s->ovf_cap};
  int key;
  if (use_two_rows(s, P, A.total_waves * (uint32_t)A.nframes)) key = render_flat_key(false, false, false, CLS_BIH_TRI, GLOME_FLAG_LB, true);  // (lds sized by the caller for two rows)
  else if (faithful) key = render_flat_key(true, true, full, CLS_EVERY, 1, false);
  else if (count) key = render_flat_key(false, true, full, CLS_EVERY, 1, false);
  else {
    const int cls = scene_class(s);
    key = render_flat_key(false, false, full, cls, cls == CLS_EVERY ? 2 : (cls == (CLS_CSG | CLS_PRIMS) ? GLOME_CSG_LB : 1), false);
  }
  return launch_render_flat(key, L, A);
}

static int render_impl(glome_scene* s, const glome_camera* cam, const glome_light* lights, int nlights, const glome_render_params* P,
                       float* rgbad_dev, uint32_t* packed_dev, glome_stats* stats, int dense, int nframes = 1, int64_t frame_stride = 0) {
  if (!s) return GLOME_E_INVALID;
  glome_ctx* ctx = s->ctx;
  if (!cam || (!rgbad_dev && !(dense != 1 && packed_dev)) || nlights < 0 || (nlights > 0 && !lights)) { ctx->err = "bad argument"; return GLOME_E_INVALID; }
  if (nlights > kMaxLights) { ctx->err = "too many lights"; return GLOME_E_LIMIT; }
  int rc = check_params(ctx, P);
  if (rc) return rc;
  if (P->mode != GLOME_MODE_TILE && P->mode != GLOME_MODE_SUBSAMPLE) { ctx->err = "unknown render mode"; return GLOME_E_INVALID; }
  if (P->mode == GLOME_MODE_SUBSAMPLE && P->blocksize > 65) { ctx->err = "GLOME_MODE_SUBSAMPLE supports tiles up to 65x65"; return GLOME_E_LIMIT; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  glome_ctx::TileTable* tt;
  // a whole renderTile frame in image layout: the pixels are independent and every tile is owned, so the tile size only
  // decides how the work is cut.  64x64 tiles are all 8x8 blocks -- no thin leftover strips (a 65x65 tile has 129
  // pixels in a column and a row, whose 64-pixel items are the least coherent and slowest of the frame)
  const bool whole = P->mode == GLOME_MODE_TILE && dense == 0 && P->tile_first == 0 && P->tile_stride == 1;
  if ((rc = get_tiles(ctx, P, P->tile_first, P->tile_stride, &tt, whole ? 64 : 0))) return rc;
  DRenderArgs A;
  memset(&A, 0, sizeof(A));
  A.S = s->dev;
  memcpy(&A.cam, cam, sizeof(DCamera));
  if (nframes < 1 || nframes > kMaxBatchFrames) { ctx->err = "a launch carries 1..32 frames"; return GLOME_E_LIMIT; }
  if (nframes > 1 && (frame_stride <= 0 || frame_stride > 0xffffffffll)) { ctx->err = "frame batches: positive frame stride"; return GLOME_E_INVALID; }
  for (int f = 1; f < nframes; f++) memcpy(&A.more_cams[f - 1], cam + f, sizeof(DCamera));
  A.nframes = nframes; A.frame_stride = nframes > 1 ? (uint32_t)frame_stride : 0u;
  for (int i = 0; i < nlights; i++) {
    memcpy(A.lights[i].pos, lights[i].pos, 12); memcpy(A.lights[i].color, lights[i].color, 12);
    A.lights[i].rad = lights[i].rad; A.lights[i].shadow = lights[i].shadow;
  }
  A.nlights = nlights; A.width = P->width; A.height = P->height; A.fog = P->fog; A.maxdepth = P->maxdepth;
  memcpy(A.thresholds, P->thresholds, 16);
  A.tiles = tt->dev; A.tile_lut = tt->lut; A.ntiles = (int)tt->host.size(); A.total_waves = tt->total_waves;
  {  // tickets per queue head: the launch's chunks dealt round-robin over the heads, the last round padded
    static const bool no_interleave = getenv("GLOME_DEBUG_NO_INTERLEAVE") != nullptr;  // (A/B: frame after frame, as until round 3)
    A.chunks_per_frame = (nframes > 1 && P->mode == GLOME_MODE_TILE && !no_interleave) ? (A.total_waves + kQueueChunk - 1) / kQueueChunk : 0u;
    const uint32_t tickets = A.chunks_per_frame ? A.chunks_per_frame * kQueueChunk * (uint32_t)nframes : A.total_waves * (uint32_t)nframes, round = kQueueChunk * kQueueShards;
    A.shard_cap = ((tickets + round - 1) / round) * kQueueChunk;
  }
  // dense 0: full frame (rgbad and / or packed); 1: dense rgbad tile payload; 2: dense packed-pixel tile payload only
  A.out5 = dense == 2 ? nullptr : rgbad_dev; A.packed = dense == 1 ? nullptr : packed_dev; A.counters = ctx->slot().d_counters; A.dense = dense != 0;
  // a plain frame (no statistics wanted, renderTile mode) does not reset the counters -- the queue heads are put back by
  // the last wave of the launch before -- so the frame is a single packet on the stream
  const bool bare = !stats && P->mode == GLOME_MODE_TILE && !P->faithful && !P->count_work;
  if (!bare && (rc = reset_counters(ctx))) return rc;
  A.want_counters = (bare || (P->mode == GLOME_MODE_SUBSAMPLE && !stats)) ? 0 : 1;  // nobody reads them without `stats`
#ifdef GLOME_PROBE
  if (const char* e = getenv("GLOME_DEBUG_FLAGS")) A.debug_flags = atoi(e);  // (render_loop)
#endif
  hipEvent_t ev_start = ctx->ev0, ev_stop = ctx->ev1;
  if (A.ntiles > 0 && P->mode == GLOME_MODE_SUBSAMPLE) {
    // scratch: v (5 float planes over the owned pixels) | queue heads, one per 128-byte line, then the dry mask | one
    // line of pass counters per tile
    size_t npx = (size_t)tt->pixels;
    const size_t ctl_words = (size_t)(kSSHeads + 1) * kSSHeadStride + (size_t)A.ntiles * nframes * 8;
    if ((rc = ensure_scratch(ctx, npx * 5 * nframes * sizeof(float) + ctl_words * sizeof(unsigned int)))) return rc;
    A.scratch = ctx->slot().d_scratch;
    A.ss_cnt = (unsigned int*)(A.scratch + npx * 5 * nframes);
    A.ss_done = A.ss_cnt + (size_t)(kSSHeads + 1) * kSSHeadStride;
    A.ss_plane = (uint32_t)npx;
    A.blocksize = P->blocksize;
    // Region size by the frames of the launch: what hides a tile's chain of five dependent passes is other frames' tiles
    // (tools/probe/ss_tune.py, profiles/r02_f_ss_regions.log: one frame alone wants the small regions whatever its tile count)
    for (int pass = 1; pass <= 5; pass++) {
      int rw, rh;
      ss_region_shape(pass, nframes >= 8 ? 2 : (nframes >= 3 ? 1 : 0), rw, rh);
      // The generic tier's rays cost fifty times a flat-tier ray and its kernel runs eight waves per CU: parallelism is worth
      // more than shared compaction.  One block per region for a frame alone (GlomeView's default scene: 68.5 -> 37.0 ms), two
      // blocks in passes 3-5 of a batch (12.1 -> 9.1 ms per frame; tools/probe/ts_variants.sh with GLOME_DEBUG_SS_REGIONS)
      if (s->dev.tier != 0) { rw = (nframes >= 3 && pass >= 3) ? 2 : 1; rh = 1; }
      // ... and once the interpreter's rays had become three times cheaper (round 3) a launch of twelve or more frames wants larger
      // regions in the later passes: 5.8 -> 5.1 ms per frame with 1x1, 2x1, 2x2, 2x2, 3x3 blocks (a launch of eight alone: 7.1 -> 10.2,
      // so those keep the rule above; profiles/r03_probes/generic_tier_sampler_regions.txt)
      if (s->dev.tier != 0 && nframes >= 12) { static const int8_t W[6] = {0, 1, 2, 2, 2, 3}, H[6] = {0, 1, 1, 2, 2, 3}; rw = W[pass]; rh = H[pass]; }
      A.ss_rw[pass] = (int8_t)rw; A.ss_rh[pass] = (int8_t)rh;
    }
    if (const char* e = getenv("GLOME_DEBUG_SS_REGIONS")) {  // "1x5,3x5,5x5,5x5,5x5"
      int q[10];
      if (sscanf(e, "%dx%d,%dx%d,%dx%d,%dx%d,%dx%d", q, q + 1, q + 2, q + 3, q + 4, q + 5, q + 6, q + 7, q + 8, q + 9) == 10)
        for (int pass = 1; pass <= 5; pass++) { A.ss_rw[pass] = (int8_t)q[2 * pass - 2]; A.ss_rh[pass] = (int8_t)q[2 * pass - 1]; }
    }
    HIPCHK(ctx, hipMemsetAsync(A.ss_cnt, 0, ctl_words * sizeof(unsigned int), ctx->stream));
    bool pooled = ctx->timing && (ctx->timing_seen++ % ctx->timing_stride) == 0 && ctx->pool_used + 2 <= (int)ctx->pool.size();
    hipEvent_t e0 = pooled ? ctx->pool[ctx->pool_used] : ctx->ev0, e1 = pooled ? ctx->pool[ctx->pool_used + 1] : ctx->ev1;
    if (pooled) ctx->pool_used += 2;
    ev_start = e0; ev_stop = e1;
    HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    const bool two_rows = use_two_rows(s, P) && scene_class(s) == CLS_BIH_TRI;  // (the sampler has a triangle-class instance only)
    size_t lds = s->dev.tier == 0 ? flat_lds_bytes(s->stack_cap, two_rows) : 0;
    bool full = s->has_nested_mats || (s->has_secondary_mats && P->maxdepth > 1);
    bool tri = s->dev.tier == 0 && (s->cls_mask & ~CLS_BIH_TRI) == 0;
    const uint32_t items = ss_plan(A).first[6] * kSSHeads;
    const bool big_tree = s->info.n_bih_nodes > 500000;
    // (a wave that is not resident yet holds no item, so the items a running wave waits for are always with running waves)
    // (the sampler's items are long -- a region's contrast tests and one or more packet walks -- so the grid is sized for ~4 per wave)
    int tgrid = persistent_grid(ctx, lds, (uint32_t)std::min<uint64_t>((uint64_t)items * 16, 0x7fffffffu), two_rows ? (big_tree ? 20 : 16) : 32, grid_floor(s));
    tgrid = (int)std::min<uint32_t>((uint32_t)tgrid, items);
    if (s->dev.tier == 0 && (rc = ensure_overflow(ctx, tgrid, 1, s->ovf_cap))) return rc;
    uint32_t* ov = ctx->slot().d_ovf;
    const FlatLaunch L{tgrid, lds, ctx->stream, s->stack_cap, ov, s->ovf_cap};
    const bool refr = s->has_refract && P->maxdepth > 1;
    int key;
    // (four waves per SIMD: with 80 registers the sampler's own state spills, and every reload waits for the loads in flight;
    // a tree of a million nodes misses the caches often enough that a fifth wave pays for the spills of 96 registers:
    // S5 2.28 -> 2.12 ms per frame, S3 0.294 -> 0.310)
    if (two_rows) key = ss_flat_key(false, CLS_BIH_TRI, big_tree ? 5 : 4, true, false);
    else if (tri && !full) key = ss_flat_key(false, CLS_BIH_TRI, 1, false, false);
    else if (tri && !refr) key = ss_flat_key(true, CLS_BIH_TRI, 1, false, false);
    // (a Refract material traced deeper than the primary ray: the reference's own traversal, see launch_render)
    else if (full && refr) key = ss_flat_key(true, CLS_EVERY, 1, false, true);
    else key = ss_flat_key(full, scene_class(s) == (CLS_CSG | CLS_PRIMS) ? (CLS_CSG | CLS_PRIMS) : CLS_EVERY, 2, false, false);
    if (s->dev.tier != 0) { if (P->count_work) launch_ss_generic(tgrid, ctx->stream, A); else launch_ss_generic_lean(tgrid, ctx->stream, A); }
    else if (!launch_ss_flat(key, L, A)) { ctx->err = "no sampler kernel instance for this scene class (build error)"; return GLOME_E_INVALID; }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
  } else if (A.ntiles > 0) {
    const bool two_rows = use_two_rows(s, P, A.total_waves * (uint32_t)nframes);
    size_t lds = s->dev.tier == 0 ? flat_lds_bytes(s->stack_cap, two_rows) : 0;
    int grid = persistent_grid(ctx, lds, A.total_waves * (uint32_t)nframes, two_rows ? 4 * GLOME_FLAG_LB : 32, grid_floor(s));
    if (s->dev.tier == 0 && (rc = ensure_overflow(ctx, grid, 1, s->ovf_cap))) return rc;
    bool pooled = ctx->timing && (ctx->timing_seen++ % ctx->timing_stride) == 0 && ctx->pool_used + 2 <= (int)ctx->pool.size();
    hipEvent_t e0 = pooled ? ctx->pool[ctx->pool_used] : ctx->ev0, e1 = pooled ? ctx->pool[ctx->pool_used + 1] : ctx->ev1;
    if (pooled) ctx->pool_used += 2;
    ev_start = e0; ev_stop = e1;
    const bool timed = stats || pooled;
    if (timed) HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
    if (!launch_render(s, A, P, grid, lds)) { ctx->err = "no kernel instance for this scene class (build error)"; return GLOME_E_INVALID; }
    HIPCHK(ctx, hipGetLastError());
    if (timed) HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    DCounters c;
    HIPCHK(ctx, hipMemcpy(&c, ctx->slot().d_counters, sizeof(c), hipMemcpyDeviceToHost));
    stats->rays_primary = c.rays_primary; stats->rays_shadow = c.rays_shadow; stats->rays_secondary = c.rays_secondary;
    stats->bih_nodes = c.bih_nodes; stats->mesh_nodes = c.mesh_nodes; stats->prim_tests = c.prim_tests;
    if (A.ntiles > 0) HIPCHK(ctx, hipEventElapsedTime(&stats->kernel_ms, ev_start, ev_stop));
    stats->n_tiles = A.ntiles; stats->n_pixels = (int32_t)tt->pixels;
    ctx->slot().launched = false;
    if (c.error) {
      HIPCHK(ctx, hipMemset(&ctx->slot().d_counters->error, 0, sizeof(unsigned int)));
      ctx->err = "device-side limit hit (traversal stack or CSG advance cap)"; return GLOME_E_LIMIT;
    }
  } else if (A.ntiles > 0) { ctx->slot().launched = true; ctx->slot().launched_on = ctx->stream; }
  return 0;
}

int glome_render_dev(glome_scene* s, const glome_camera* cam, const glome_light* lights, int nlights, const glome_render_params* P,
                     float* rgbad_dev, uint32_t* packed_dev, glome_stats* stats) {
  return render_impl(s, cam, lights, nlights, P, rgbad_dev, packed_dev, stats, 0);
}
int glome_render_tiles_dev(glome_scene* s, const glome_camera* cam, const glome_light* lights, int nlights, const glome_render_params* P,
                           float* payload_dev, glome_stats* stats) {
  return render_impl(s, cam, lights, nlights, P, payload_dev, nullptr, stats, 1);
}
int glome_render_tiles_packed_dev(glome_scene* s, const glome_camera* cam, const glome_light* lights, int nlights, const glome_render_params* P,
                                  uint32_t* payload_dev, glome_stats* stats) {
  return render_impl(s, cam, lights, nlights, P, nullptr, payload_dev, stats, 2);
}

int glome_render_tiles_packed_batch_dev(glome_scene* s, const glome_camera* cams, int nframes, const glome_light* lights, int nlights,
                                        const glome_render_params* P, uint32_t* payload_dev, int64_t frame_stride_pixels, glome_stats* stats) {
  return render_impl(s, cams, lights, nlights, P, nullptr, payload_dev, stats, 2, nframes, frame_stride_pixels);
}
int glome_render_packed_batch_dev(glome_scene* s, const glome_camera* cams, int nframes, const glome_light* lights, int nlights,
                                  const glome_render_params* P, uint32_t* packed_dev, int64_t frame_stride_pixels, glome_stats* stats) {
  return render_impl(s, cams, lights, nlights, P, nullptr, packed_dev, stats, 0, nframes, frame_stride_pixels);
}

int glome_render(glome_scene* s, const glome_camera* cam, const glome_light* lights, int nlights, const glome_render_params* P, float* rgbad,
                 uint32_t* packed, glome_stats* stats) {
  if (!s) return GLOME_E_INVALID;
  glome_ctx* ctx = s->ctx;
  int rc = check_params(ctx, P);
  if (rc) return rc;
  if (!rgbad) { ctx->err = "null framebuffer"; return GLOME_E_INVALID; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  size_t np = (size_t)P->width * P->height;
  float* d5 = nullptr; uint32_t* dp = nullptr;
  HIPCHK(ctx, hipMalloc((void**)&d5, np * 5 * sizeof(float)));
  // tiles this call does not own keep the caller's values
  hipError_t e = hipMemcpy(d5, rgbad, np * 5 * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess && packed) { e = hipMalloc((void**)&dp, np * 4); if (e == hipSuccess) e = hipMemcpy(dp, packed, np * 4, hipMemcpyHostToDevice); }
  if (e != hipSuccess) { ctx->err = hipGetErrorString(e); (void)hipFree(d5); if (dp) (void)hipFree(dp); return GLOME_E_HIP; }
  glome_stats local;
  rc = glome_render_dev(s, cam, lights, nlights, P, d5, dp, stats ? stats : &local);
  if (rc == 0) {
    e = hipMemcpy(rgbad, d5, np * 5 * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess && packed) e = hipMemcpy(packed, dp, np * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); rc = GLOME_E_HIP; }
  }
  (void)hipFree(d5);
  if (dp) (void)hipFree(dp);
  return rc;
}

// ---- per-ray seams ----
static int batch_grid(glome_ctx* ctx, size_t n, size_t lds) {
  size_t blocks = (n + 63) / 64;
  return (int)std::max<size_t>(1, std::min<size_t>(blocks, (size_t)persistent_grid(ctx, lds, 0x7fffffff) * 4));
}
int glome_rayint_batch_dev(glome_scene* s, size_t n, const float* ox, const float* oy, const float* oz, const float* dx, const float* dy,
                           const float* dz, const float* tmax, float* t, int32_t* prim, float* nx, float* ny, float* nz, int32_t* tex8) {
  if (!s) return GLOME_E_INVALID;
  glome_ctx* ctx = s->ctx;
  if (n == 0) return 0;
  if (!ox || !oy || !oz || !dx || !dy || !dz || !tmax) { ctx->err = "null ray stream"; return GLOME_E_INVALID; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  RayStream R{ox, oy, oz, dx, dy, dz, tmax};
  HitStream H{t, prim, nx, ny, nz, tex8};
  if (s->dev.tier == 0) {
    size_t lds = flat_lds_bytes(s->stack_cap);
    int grid = batch_grid(ctx, n, lds), rc;
    if ((rc = ensure_overflow(ctx, grid, 1, s->ovf_cap))) return rc;
    launch_rayint_batch_flat(FlatLaunch{grid, lds, ctx->stream, s->stack_cap, ctx->slot().d_ovf, s->ovf_cap}, s->dev, n, R, H, ctx->slot().d_counters);
  } else {
    if (int rcc = reset_counters(ctx)) return rcc;
    launch_rayint_batch_generic(batch_grid(ctx, n, 0), ctx->stream, s->dev, n, R, H, ctx->slot().d_counters);
  }
  ctx->slot().launched = true; ctx->slot().launched_on = ctx->stream;  // (a limit hit here is this call's to report, at the next synchronize)
  HIPCHK(ctx, hipGetLastError());
  return 0;
}
int glome_shadow_batch_dev(glome_scene* s, size_t n, const float* ox, const float* oy, const float* oz, const float* dx, const float* dy,
                           const float* dz, const float* tmax, uint8_t* occluded) {
  if (!s) return GLOME_E_INVALID;
  glome_ctx* ctx = s->ctx;
  if (n == 0) return 0;
  if (!ox || !oy || !oz || !dx || !dy || !dz || !tmax || !occluded) { ctx->err = "null stream"; return GLOME_E_INVALID; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  RayStream R{ox, oy, oz, dx, dy, dz, tmax};
  if (s->dev.tier == 0) {
    size_t lds = flat_lds_bytes(s->stack_cap);
    int grid = batch_grid(ctx, n, lds), rc;
    if ((rc = ensure_overflow(ctx, grid, 1, s->ovf_cap))) return rc;
    launch_shadow_batch_flat(FlatLaunch{grid, lds, ctx->stream, s->stack_cap, ctx->slot().d_ovf, s->ovf_cap}, s->dev, n, R, occluded, ctx->slot().d_counters);
  } else {
    if (int rcc = reset_counters(ctx)) return rcc;
    launch_shadow_batch_generic(batch_grid(ctx, n, 0), ctx->stream, s->dev, n, R, occluded, ctx->slot().d_counters);
  }
  ctx->slot().launched = true; ctx->slot().launched_on = ctx->stream;
  HIPCHK(ctx, hipGetLastError());
  return 0;
}

// host-buffer wrappers: stage the SoA streams through HBM
struct Staging {
  glome_ctx* ctx;
  std::vector<void*> bufs;
  ~Staging() { for (void* p : bufs) (void)hipFree(p); }
  template <class T> T* in(const T* h, size_t n) {
    T* d = nullptr;
    if (hipMalloc((void**)&d, std::max<size_t>(1, n) * sizeof(T)) != hipSuccess) return nullptr;
    bufs.push_back(d);
    if (h && hipMemcpy(d, h, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
  }
};
int glome_rayint_batch(glome_scene* s, size_t n, const float* ox, const float* oy, const float* oz, const float* dx, const float* dy,
                       const float* dz, const float* tmax, float* t, int32_t* prim, float* nx, float* ny, float* nz, int32_t* tex8) {
  if (!s) return GLOME_E_INVALID;
  glome_ctx* ctx = s->ctx;
  if (n == 0) return 0;
  if (!ox || !oy || !oz || !dx || !dy || !dz || !tmax) { ctx->err = "null ray stream"; return GLOME_E_INVALID; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staging st{ctx, {}};
  const float* in[7] = {ox, oy, oz, dx, dy, dz, tmax};
  float* din[7];
  for (int k = 0; k < 7; k++) if (!(din[k] = st.in(in[k], n))) { ctx->err = "staging allocation failed"; return GLOME_E_HIP; }
  float* dt = t ? st.in<float>(nullptr, n) : nullptr;
  int32_t* dprim = prim ? st.in<int32_t>(nullptr, n) : nullptr;
  float* dnx = nx ? st.in<float>(nullptr, n) : nullptr;
  float* dny = ny ? st.in<float>(nullptr, n) : nullptr;
  float* dnz = nz ? st.in<float>(nullptr, n) : nullptr;
  int32_t* dtex = tex8 ? st.in<int32_t>(nullptr, 8 * n) : nullptr;
  int rc = glome_rayint_batch_dev(s, n, din[0], din[1], din[2], din[3], din[4], din[5], din[6], dt, dprim, dnx, dny, dnz, dtex);
  if (rc) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if ((rc = check_device_error(ctx))) return rc;
  if (t) HIPCHK(ctx, hipMemcpy(t, dt, n * 4, hipMemcpyDeviceToHost));
  if (prim) HIPCHK(ctx, hipMemcpy(prim, dprim, n * 4, hipMemcpyDeviceToHost));
  if (nx) HIPCHK(ctx, hipMemcpy(nx, dnx, n * 4, hipMemcpyDeviceToHost));
  if (ny) HIPCHK(ctx, hipMemcpy(ny, dny, n * 4, hipMemcpyDeviceToHost));
  if (nz) HIPCHK(ctx, hipMemcpy(nz, dnz, n * 4, hipMemcpyDeviceToHost));
  if (tex8) HIPCHK(ctx, hipMemcpy(tex8, dtex, n * 32, hipMemcpyDeviceToHost));
  return 0;
}
int glome_shadow_batch(glome_scene* s, size_t n, const float* ox, const float* oy, const float* oz, const float* dx, const float* dy,
                       const float* dz, const float* tmax, uint8_t* occluded) {
  if (!s) return GLOME_E_INVALID;
  glome_ctx* ctx = s->ctx;
  if (n == 0) return 0;
  if (!ox || !oy || !oz || !dx || !dy || !dz || !tmax || !occluded) { ctx->err = "null stream"; return GLOME_E_INVALID; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staging st{ctx, {}};
  const float* in[7] = {ox, oy, oz, dx, dy, dz, tmax};
  float* din[7];
  for (int k = 0; k < 7; k++) if (!(din[k] = st.in(in[k], n))) { ctx->err = "staging allocation failed"; return GLOME_E_HIP; }
  uint8_t* docc = st.in<uint8_t>(nullptr, n);
  int rc = glome_shadow_batch_dev(s, n, din[0], din[1], din[2], din[3], din[4], din[5], din[6], docc);
  if (rc) return rc;
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  if ((rc = check_device_error(ctx))) return rc;
  HIPCHK(ctx, hipMemcpy(occluded, docc, n, hipMemcpyDeviceToHost));
  return 0;
}
int glome_inside_batch(glome_scene* s, size_t n, const float* px, const float* py, const float* pz, uint8_t* inside) {
  if (!s) return GLOME_E_INVALID;
  glome_ctx* ctx = s->ctx;
  if (n == 0) return 0;
  if (!px || !py || !pz || !inside) { ctx->err = "null stream"; return GLOME_E_INVALID; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Staging st{ctx, {}};
  float *dx = st.in(px, n), *dy = st.in(py, n), *dz = st.in(pz, n);
  uint8_t* din = st.in<uint8_t>(nullptr, n);
  if (!dx || !dy || !dz || !din) { ctx->err = "staging allocation failed"; return GLOME_E_HIP; }
  if (int rcc = reset_counters(ctx)) return rcc;
  launch_inside_batch(batch_grid(ctx, n, 0), ctx->stream, s->dev, n, dx, dy, dz, din, ctx->slot().d_counters);
  HIPCHK(ctx, hipGetLastError());
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  int rc = check_device_error(ctx);
  if (rc) return rc;
  HIPCHK(ctx, hipMemcpy(inside, din, n, hipMemcpyDeviceToHost));
  return 0;
}

// ---- tile payload transport ----
int64_t glome_tiles_payload_floats(const glome_render_params* P, int tile_first, int tile_stride) {
  if (!P || P->width <= 0 || P->height <= 0 || P->blocksize <= 0 || tile_stride <= 0 || tile_first < 0) return -1;
  std::vector<DTile> t; uint32_t w; int64_t px;
  owned_tiles(P->width, P->height, P->blocksize, tile_first, tile_stride, P->rank0_share_pct, t, w, px);
  return px * 5;
}
int glome_tiles_pack_dev(glome_ctx* ctx, const glome_render_params* P, const float* rgbad_dev, float* payload_dev) {
  if (!ctx) return GLOME_E_INVALID;
  int rc = check_params(ctx, P);
  if (rc) return rc;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  glome_ctx::TileTable* tt;
  if ((rc = get_tiles(ctx, P, P->tile_first, P->tile_stride, &tt))) return rc;
  if (tt->host.empty()) return 0;
  hipLaunchKernelGGL(k_tiles_pack, dim3(21, std::min<int>((int)tt->host.size(), 1024)), dim3(256), 0, ctx->stream, tt->dev, (int)tt->host.size(), P->width, rgbad_dev, payload_dev);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}
// one table over every tile of the frame; pix_base = owner rank's slab offset + the tile's offset inside that rank's payload
static int gathered_table(glome_ctx* ctx, const glome_render_params* P, int world, int64_t stride_pixels, glome_ctx::TileTable** out) {
  std::vector<int> key{P->width, P->height, P->blocksize, -world, (int)stride_pixels, P->rank0_share_pct};
  auto it = ctx->tile_cache.find(key);
  if (it == ctx->tile_cache.end()) {
    glome_ctx::TileTable tt;
    for (int r = 0; r < world; r++) {
      std::vector<DTile> t; uint32_t w; int64_t px;
      owned_tiles(P->width, P->height, P->blocksize, r, world, P->rank0_share_pct, t, w, px);
      if ((int64_t)r * stride_pixels + px > 0xffffffffll) { ctx->err = "gathered payload too large"; return GLOME_E_LIMIT; }
      for (DTile& d : t) { d.pix_base += (uint32_t)(r * stride_pixels); tt.host.push_back(d); }
      tt.pixels += px;
    }
    size_t bytes = std::max<size_t>(1, tt.host.size()) * sizeof(DTile);
    HIPCHK(ctx, hipMalloc((void**)&tt.dev, bytes));
    if (!tt.host.empty()) HIPCHK(ctx, hipMemcpy(tt.dev, tt.host.data(), tt.host.size() * sizeof(DTile), hipMemcpyHostToDevice));
    it = ctx->tile_cache.emplace(key, std::move(tt)).first;
  }
  *out = &it->second;
  return 0;
}
int glome_tiles_blit_all_dev(glome_ctx* ctx, const glome_render_params* P, int world, const float* gathered_dev, int64_t stride_floats,
                             float* rgbad_dev, uint32_t* packed_dev) {
  if (!ctx) return GLOME_E_INVALID;
  int rc = check_params(ctx, P);
  if (rc) return rc;
  if (world <= 0 || stride_floats < 0 || stride_floats % 5 != 0) { ctx->err = "bad world / stride"; return GLOME_E_INVALID; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  glome_ctx::TileTable* tt;
  if ((rc = gathered_table(ctx, P, world, stride_floats / 5, &tt))) return rc;
  if (tt->host.empty()) return 0;
  hipLaunchKernelGGL(k_tiles_blit, dim3(17, std::min<int>((int)tt->host.size(), 1024)), dim3(256), 0, ctx->stream, tt->dev, (int)tt->host.size(), P->width, gathered_dev, rgbad_dev, packed_dev);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}
int glome_tiles_blit_all_packed_dev(glome_ctx* ctx, const glome_render_params* P, int world, const uint32_t* gathered_dev, int64_t stride_pixels,
                                    uint32_t* packed_dev) {
  return glome_tiles_blit_all_packed_batch_dev(ctx, P, world, gathered_dev, stride_pixels, 1, 0, packed_dev, 0);
}
int glome_tiles_blit_all_packed_batch_dev(glome_ctx* ctx, const glome_render_params* P, int world, const uint32_t* gathered_dev, int64_t stride_pixels,
                                          int nframes, int64_t payload_frame_stride, uint32_t* packed_dev, int64_t out_frame_stride) {
  if (!ctx) return GLOME_E_INVALID;
  int rc = check_params(ctx, P);
  if (rc) return rc;
  if (world <= 0 || stride_pixels < 0 || !gathered_dev || !packed_dev || nframes < 1 || nframes > kMaxBatchFrames || payload_frame_stride < 0 || out_frame_stride < 0) {
    ctx->err = "bad world / stride / buffer / frame count"; return GLOME_E_INVALID;
  }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  glome_ctx::TileTable* tt;
  if ((rc = gathered_table(ctx, P, world, stride_pixels, &tt))) return rc;
  if (tt->host.empty()) return 0;
  hipLaunchKernelGGL(k_tiles_blit_packed, dim3(17, std::min<int>((int)tt->host.size(), 1024), nframes), dim3(256), 0, ctx->stream, tt->dev, (int)tt->host.size(), P->width, gathered_dev,
                     packed_dev, (size_t)payload_frame_stride, (size_t)out_frame_stride);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}
int glome_tiles_blit_dev(glome_ctx* ctx, const glome_render_params* P, int tile_first, int tile_stride, const float* payload_dev, float* rgbad_dev, uint32_t* packed_dev) {
  if (!ctx) return GLOME_E_INVALID;
  int rc = check_params(ctx, P);
  if (rc) return rc;
  if (tile_stride <= 0 || tile_first < 0) { ctx->err = "bad tile shard"; return GLOME_E_INVALID; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  glome_ctx::TileTable* tt;
  if ((rc = get_tiles(ctx, P, tile_first, tile_stride, &tt))) return rc;
  if (tt->host.empty()) return 0;
  hipLaunchKernelGGL(k_tiles_blit, dim3(17, std::min<int>((int)tt->host.size(), 1024)), dim3(256), 0, ctx->stream, tt->dev, (int)tt->host.size(), P->width, payload_dev, rgbad_dev, packed_dev);
  HIPCHK(ctx, hipGetLastError());
  return 0;
}


// ================================================================================================ several GPUs, one process
// renderTiles' `runPar $ parMap` over tiles followed by `forM_ tiles (blitTile surf)` (Glome.hs:379-386) across the GPUs of a
// node, for a host that drives all of them from one process (the Haskell host of INTEGRATION.md): scenes[i] is the scene
// committed on context i, tile k of the frame belongs to rank k mod n (or to the rank rank0_share_pct's pattern gives it), a rank renders its tiles of up to 32 frames in one
// launch straight into a packed payload, the payloads travel to rank 0's GPU over xGMI, one launch there blits the frames.
// The exchange is the path's only communication step.  Transport: RCCL send / recv in one group (librccl is opened at run
// time, so the library carries no link-time dependency on it) when the ranks sit on distinct devices; peer copies on rank
// 0's stream (hipMemcpyPeerAsync) when librccl cannot be loaded or two ranks share a device (how the path is tested on a
// one-GPU box).  glome_amd/dist.py is the same data path with one process per GPU and torch.distributed's gather.
#include <dlfcn.h>
namespace {
struct Rccl {  // the five entry points the gather needs, resolved from librccl.so on first use
  typedef void* comm_t;
  int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  bool debug_lib = false;  // GLOME_DEBUG_RCCL_LIB named the library: the test stub, not RCCL
  static Rccl& get() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
      void* h = nullptr;
      // GLOME_DEBUG_RCCL_LIB: another library with the same entry points (tests/rcclstub: the RCCL branch exercised on one GPU)
      if (const char* dbg = getenv("GLOME_DEBUG_RCCL_LIB")) { h = dlopen(dbg, RTLD_NOW | RTLD_GLOBAL); r.debug_lib = true; }
      else for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
      if (!h) return;
      r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
      r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
      r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
      r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
      r.Send = (decltype(r.Send))dlsym(h, "ncclSend");
      r.Recv = (decltype(r.Recv))dlsym(h, "ncclRecv");
      r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
      r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
    });
    return r;
  }
};
constexpr int kNcclUint32 = 3;  // ncclUint32 (nccl.h: ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3)
}  // namespace

struct glome_multi {
  int n = 0;
  std::vector<glome_scene*> scenes;
  glome_render_params P{}, P0{};           // the frame (work tiles in renderTile mode); P0 = P with rank 0's shard
  std::vector<glome_render_params> Pl;     // rank i's shard: tiles i, i + n, ...
  int64_t maxp = 0;                        // pixels of the largest shard: a frame's slot in every payload
  std::vector<uint32_t*> payload;          // rank i's payload on device i: kMaxBatchFrames * maxp words (rank 0: its slab of `gathered`)
  uint32_t* gathered = nullptr;            // device 0: n slabs of kMaxBatchFrames * maxp words
  std::vector<hipEvent_t> rendered;        // rank i's launch is done
  hipEvent_t consumed = nullptr;           // rank 0 has taken the payloads of the last call
  bool used = false;
  bool rccl = false;
  bool direct = false;                     // every rank's render kernel stores its pixels straight into packed_dev on rank 0's GPU (peer access)
  std::vector<Rccl::comm_t> comms;
  std::string err;
};
#define MHIP(m, call)                                                                 \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) { (m)->err = std::string(#call) + ": " + hipGetErrorString(e_); return GLOME_E_HIP; } \
  } while (0)

glome_multi* glome_multi_create(glome_scene* const* scenes, int n, const glome_render_params* P, int transport) {
  if (!scenes || n < 1 || n > 64 || !P) { g_global_error = "glome_multi_create: bad argument"; return nullptr; }
  for (int i = 0; i < n; i++) {
    if (!scenes[i]) { g_global_error = "glome_multi_create: null scene"; return nullptr; }
    for (int j = 0; j < i; j++) if (scenes[j]->ctx == scenes[i]->ctx) { g_global_error = "glome_multi_create: every rank needs a context (and scene) of its own"; return nullptr; }
  }
  if (check_params(scenes[0]->ctx, P)) { g_global_error = scenes[0]->ctx->err; return nullptr; }
  glome_multi* m = new glome_multi();
  m->n = n;
  m->scenes.assign(scenes, scenes + n);
  m->P = *P;
  // renderTile's pixels do not depend on the tile map (the adaptive sampler's do, Q21): shard 64x64 work tiles then
  if (P->mode == GLOME_MODE_TILE) m->P.blocksize = 64;
  m->P.tile_first = 0; m->P.tile_stride = 1;
  for (int i = 0; i < n; i++) {
    glome_render_params q = m->P;
    q.tile_first = i; q.tile_stride = n;
    m->Pl.push_back(q);
    m->maxp = std::max<int64_t>(m->maxp, glome_tiles_payload_floats(&q, i, n) / 5);
  }
  m->maxp = std::max<int64_t>(m->maxp, 1);
  auto fail = [&](const std::string& what) { g_global_error = "glome_multi_create: " + what; glome_multi_destroy(m); return (glome_multi*)nullptr; };
  const size_t slab = (size_t)kMaxBatchFrames * (size_t)m->maxp;
  // transport 2 ("direct"): no payloads, no exchange, no blit -- every rank's render kernel writes its tiles' packed pixels where they
  // belong in the caller's framebuffer on rank 0's GPU (4 bytes per pixel over xGMI, a 256-byte store per work item), and rank 0's
  // stream waits for the other ranks' launches.  It needs every rank's device to reach rank 0's memory: the same device, or peer access.
  if (transport == 2 && n > 1) {
    bool can_all = true;
    for (int i = 1; i < n && can_all; i++) {
      if (scenes[i]->ctx->device == scenes[0]->ctx->device) continue;
      int can = 0;
      (void)hipDeviceCanAccessPeer(&can, scenes[i]->ctx->device, scenes[0]->ctx->device);
      if (!can) { can_all = false; break; }
      if (hipSetDevice(scenes[i]->ctx->device) != hipSuccess) { can_all = false; break; }
      hipError_t e = hipDeviceEnablePeerAccess(scenes[0]->ctx->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); can_all = false; }
    }
    m->direct = can_all;
    // nobody receives or blits any more: every rank owns a fair share of the tiles, whatever weight the caller gave rank 0
    if (m->direct) { m->P.rank0_share_pct = 0; for (auto& q : m->Pl) q.rank0_share_pct = 0; }
  }
  const bool use_rccl = transport == 1 || (transport == 2 && !m->direct);
  m->payload.assign(n, nullptr);
  m->rendered.assign(n, nullptr);
  if (!m->direct) {
    if (hipSetDevice(scenes[0]->ctx->device) != hipSuccess || hipMalloc((void**)&m->gathered, slab * n * sizeof(uint32_t)) != hipSuccess) return fail("device allocation failed");
    m->payload[0] = m->gathered;  // rank 0 renders into its own slab
  }
  for (int i = 0; i < n; i++) {
    if (hipSetDevice(scenes[i]->ctx->device) != hipSuccess) return fail("hipSetDevice failed");
    if (!m->direct && i > 0 && hipMalloc((void**)&m->payload[i], slab * sizeof(uint32_t)) != hipSuccess) return fail("device allocation failed");
    if (hipEventCreateWithFlags(&m->rendered[i], hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
  }
  (void)hipSetDevice(scenes[0]->ctx->device);
  if (hipEventCreateWithFlags(&m->consumed, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
  // transport
  bool distinct = true;
  for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) distinct &= scenes[i]->ctx->device != scenes[j]->ctx->device;
  // (GLOME_DEBUG_RCCL_SAME_DEVICE lifts the distinct-device condition -- real RCCL refuses two ranks on one device -- for the
  // stubbed transport of the one-GPU test, and ONLY for it: with the real library loaded the variable is ignored)
  if (m->direct) return m;
  if (use_rccl && n > 1 && (distinct || (Rccl::get().debug_lib && getenv("GLOME_DEBUG_RCCL_SAME_DEVICE"))) && Rccl::get().ok) {
    std::vector<int> devs;
    for (int i = 0; i < n; i++) devs.push_back(scenes[i]->ctx->device);
    m->comms.assign(n, nullptr);
    int rc = Rccl::get().CommInitAll(m->comms.data(), n, devs.data());
    if (rc != 0) return fail(std::string("ncclCommInitAll: ") + (Rccl::get().GetErrorString ? Rccl::get().GetErrorString(rc) : "error"));
    m->rccl = true;
  } else if (n > 1) {
    for (int i = 1; i < n; i++) {  // peer copies into rank 0's GPU: let it see the others' memory where they differ
      if (scenes[i]->ctx->device == scenes[0]->ctx->device) continue;
      int can = 0;
      (void)hipDeviceCanAccessPeer(&can, scenes[0]->ctx->device, scenes[i]->ctx->device);
      if (can) { (void)hipSetDevice(scenes[0]->ctx->device); hipError_t e = hipDeviceEnablePeerAccess(scenes[i]->ctx->device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError(); }
    }
  }
  return m;
}
void glome_multi_destroy(glome_multi* m) {
  if (!m) return;
  for (int i = 0; i < m->n; i++) {
    (void)hipSetDevice(m->scenes[i]->ctx->device);
    (void)hipStreamSynchronize(m->scenes[i]->ctx->stream);
    if (i > 0 && i < (int)m->payload.size() && m->payload[i]) (void)hipFree(m->payload[i]);
    if (i < (int)m->rendered.size() && m->rendered[i]) (void)hipEventDestroy(m->rendered[i]);
    if (m->rccl && i < (int)m->comms.size() && m->comms[i]) (void)Rccl::get().CommDestroy(m->comms[i]);
  }
  (void)hipSetDevice(m->scenes[0]->ctx->device);
  if (m->gathered) (void)hipFree(m->gathered);
  if (m->consumed) (void)hipEventDestroy(m->consumed);
  delete m;
}
const char* glome_multi_last_error(const glome_multi* m) { return m ? m->err.c_str() : g_global_error.c_str(); }
const char* glome_multi_transport(const glome_multi* m) { return !m ? "" : (m->n == 1 ? "none" : (m->direct ? "direct" : (m->rccl ? "rccl" : "peer-copy"))); }

int glome_multi_render(glome_multi* m, const glome_camera* cams, int nframes, const glome_light* lights, int nlights, uint32_t* packed_dev) {
  if (!m || !cams || !packed_dev) return GLOME_E_INVALID;
  if (nframes < 1 || nframes > kMaxBatchFrames || (m->P.mode != GLOME_MODE_TILE && nframes != 1)) { m->err = "a call carries 1..32 frames (one in adaptive mode)"; return GLOME_E_LIMIT; }
  const int n = m->n;
  glome_ctx* c0 = m->scenes[0]->ctx;
  const size_t slab = (size_t)kMaxBatchFrames * (size_t)m->maxp;
  if (m->direct) {
    // every rank renders its tiles of the nframes views into the frames themselves (frame layout, its own tiles only); rank 0's stream
    // -- the one the caller orders its reads of packed_dev on -- waits for the others
    const int64_t fs = (int64_t)m->P.width * m->P.height;
    for (int i = n - 1; i >= 0; i--) {  // (rank 0 last: its launch is queued behind nothing of the others)
      glome_ctx* c = m->scenes[i]->ctx;
      MHIP(m, hipSetDevice(c->device));
      int rc = nframes > 1 ? render_impl(m->scenes[i], cams, lights, nlights, &m->Pl[i], nullptr, packed_dev, nullptr, 0, nframes, fs)
                           : render_impl(m->scenes[i], cams, lights, nlights, &m->Pl[i], nullptr, packed_dev, nullptr, 0);
      if (rc) { m->err = "rank " + std::to_string(i) + ": " + c->err; return rc; }
      if (i > 0) MHIP(m, hipEventRecord(m->rendered[i], c->stream));
    }
    MHIP(m, hipSetDevice(c0->device));
    for (int i = 1; i < n; i++) MHIP(m, hipStreamWaitEvent(c0->stream, m->rendered[i], 0));
    m->used = true;
    return 0;
  }
  // every rank renders its tiles of the nframes views into its payload (frame f at f * maxp)
  for (int i = 0; i < n; i++) {
    glome_ctx* c = m->scenes[i]->ctx;
    MHIP(m, hipSetDevice(c->device));
    if (m->used && i > 0 && !m->rccl) MHIP(m, hipStreamWaitEvent(c->stream, m->consumed, 0));  // rank 0 has read this payload's last contents
    int rc = nframes > 1 ? render_impl(m->scenes[i], cams, lights, nlights, &m->Pl[i], nullptr, m->payload[i], nullptr, 2, nframes, m->maxp)
                         : render_impl(m->scenes[i], cams, lights, nlights, &m->Pl[i], nullptr, m->payload[i], nullptr, 2);
    if (rc) { m->err = "rank " + std::to_string(i) + ": " + c->err; return rc; }
    if (i > 0) MHIP(m, hipEventRecord(m->rendered[i], c->stream));
  }
  // the exchange: rank i's payload -> slab i on rank 0's GPU
  const size_t words = (size_t)nframes * (size_t)m->maxp;
  if (n > 1 && m->rccl) {
    Rccl& R = Rccl::get();
    int rc = R.GroupStart();
    // rank i's Send is ordered on ITS stream behind its render and in front of its next one, so a payload is never rewritten
    // before it has left; rank 0's Recvs are ordered on its stream in front of the blit.  (The device of a call's communicator
    // is made current first: RCCL releases before 2.18 want that.)
    hipError_t he = hipSuccess;  // (no early return inside the group: GroupEnd is called whatever happens)
    for (int i = 1; i < n && rc == 0 && he == hipSuccess; i++) {
      if ((he = hipSetDevice(m->scenes[i]->ctx->device)) != hipSuccess) break;
      rc = R.Send(m->payload[i], words, kNcclUint32, 0, m->comms[i], m->scenes[i]->ctx->stream);
      if ((he = hipSetDevice(c0->device)) != hipSuccess) break;
      if (rc == 0) rc = R.Recv(m->gathered + (size_t)i * slab, words, kNcclUint32, i, m->comms[0], c0->stream);
    }
    int rc2 = R.GroupEnd();
    if (he != hipSuccess) { m->err = std::string("hipSetDevice inside the RCCL group: ") + hipGetErrorString(he); return GLOME_E_HIP; }
    if (rc || rc2) { m->err = std::string("RCCL send / recv: ") + (R.GetErrorString ? R.GetErrorString(rc ? rc : rc2) : "error"); return GLOME_E_HIP; }
  } else if (n > 1) {
    MHIP(m, hipSetDevice(c0->device));
    for (int i = 1; i < n; i++) {
      MHIP(m, hipStreamWaitEvent(c0->stream, m->rendered[i], 0));
      MHIP(m, hipMemcpyPeerAsync(m->gathered + (size_t)i * slab, c0->device, m->payload[i], m->scenes[i]->ctx->device, words * sizeof(uint32_t), c0->stream));
    }
    MHIP(m, hipEventRecord(m->consumed, c0->stream));
  }
  m->used = true;
  // blitTile for every tile of every frame, one launch on rank 0
  MHIP(m, hipSetDevice(c0->device));
  int rc = glome_tiles_blit_all_packed_batch_dev(c0, &m->P, n, m->gathered, (int64_t)slab, nframes, m->maxp, packed_dev, (int64_t)m->P.width * m->P.height);
  if (rc) { m->err = c0->err; return rc; }
  return 0;
}
int glome_multi_synchronize(glome_multi* m) {
  if (!m) return GLOME_E_INVALID;
  int rc = 0;
  for (int i = m->n - 1; i >= 0; i--) {
    MHIP(m, hipSetDevice(m->scenes[i]->ctx->device));
    int r = glome_ctx_synchronize(m->scenes[i]->ctx);
    if (r) { m->err = m->scenes[i]->ctx->err; rc = r; }
  }
  return rc;
}
// the one-call form SURVEY.md Appendix B names: host framebuffer out
int glome_render_multi(glome_scene* const* scenes, int n, const glome_camera* cam, const glome_light* lights, int nlights, const glome_render_params* P, uint32_t* packed) {
  if (!packed) { g_global_error = "glome_render_multi: null framebuffer"; return GLOME_E_INVALID; }
  glome_multi* m = glome_multi_create(scenes, n, P, 2);  // direct stores where the GPUs reach each other's memory, else RCCL, else peer copies
  if (!m) return GLOME_E_INVALID;
  uint32_t* d = nullptr;
  const size_t np = (size_t)P->width * P->height;
  int rc = 0;
  if (hipSetDevice(scenes[0]->ctx->device) != hipSuccess || hipMalloc((void**)&d, np * 4) != hipSuccess) rc = GLOME_E_HIP;
  if (!rc) rc = glome_multi_render(m, cam, 1, lights, nlights, d);
  if (!rc) rc = glome_multi_synchronize(m);
  if (!rc && hipMemcpy(packed, d, np * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = GLOME_E_HIP;
  if (rc) g_global_error = std::string("glome_render_multi: ") + m->err;
  if (d) (void)hipFree(d);
  glome_multi_destroy(m);
  return rc;
}
#endif  // GLOME_IN_PART(0)
