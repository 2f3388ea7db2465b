// rt_types.h -- the packed scene layout in HBM, shared by the host flattener and the HIP kernels.
//
// Everything is a pool of 16-byte words (float4 / uint4) so every fetch is one dwordx4:
//
//   recs      uint4   one record per scene-graph reference: {kind|flags, a, b, c}
//   spheres   float4  (cx, cy, cz, r)                                               16 B / sphere
//   tris      float4  x3: (p1.xyz, n.x) (e1.xyz, n.y) (e2.xyz, n.z)                 48 B / triangle
//                     e1 = p2-p1, e2 = p3-p1, n = normalize(e1 x e2), all evaluated in double on
//                     the host (Triangle.hs:47-48,73) and rounded once
//   trinorms  float4  heap: TriangleNorm = 6 words (p1,e1,e2,n1,n2,n3); mesh vertex normals = 3 words per triangle
//   boxes     float4  x2: (min, -) (max, -)
//   planes    float4  (n.xyz, offset)
//   discs     float4  x2: (p.xyz, r^2) (n.xyz, -)
//   quadrics  float4  cylinder (r, h1, h2, -) / cone (r, clip1, clip2, height)
//   xfms      float4  x6: forward rows 0..2, inverse rows 0..2
//   bihhdr    float4  x3: (bb.min, root ref) (bb.max, leaf class) (first_prim - first_rec, -, -, -)
//   bihnodes  float4  branch (lsplit, rsplit, axis | left_ref<<2, right_ref); child refs: bit 29 = leaf, bits 28..26 =
//                     item count (7 = read the leaf slot), bits 25..0 = first record -- a leaf of <= 6 items costs no fetch;
//                     leaf slot (-, -, count, first rec)
//   meshhdr   float4  x2: (bb.min, root ref) (bb.max, root count)
//   meshnodes float4  x4: (lbb.min, lref) (lbb.max, lcnt) (rbb.min, rref) (rbb.max, rcnt)
//   mtris     float4  x3 per mesh triangle in leaf order, same layout as `tris`
//   mtrimeta  uint4   per mesh triangle: (normal base+1 or 0, material+1 or 0, -, -)
//   mats      float4  x3: (kind, a, b, w) (r, g, b, alpha | refl, refr, ior) (amb, kd, ks, shine);
//                     Warp: (kind, frame record, scene record, transform) (first light, light count, -, -)
//   wlights   float4  x2 per light of a Warp material, laid out as DLight (pos, color, rad, shadow)
//   matkids   uint    AdditiveLayers children
//   entries   uint4   flat-tier root program: (rec, incoming tex stack, flags, -)
#pragma once
#include <stdint.h>

namespace glome {

enum RecKind : uint32_t {
  R_VOID = 0, R_SPHERE = 1, R_TRI = 2, R_TRIN = 3, R_BOX = 4, R_PLANE = 5, R_DISC = 6, R_CYL = 7, R_CONE = 8,
  R_LIST = 9, R_INSTANCE = 10, R_DIFF = 11, R_ISECT = 12, R_BOUND = 13, R_INNERBOUND = 14, R_BIH = 15, R_MESH = 16, R_TEX = 17
};
constexpr uint32_t RF_KINDMASK = 0xffu;
constexpr uint32_t RF_NOVIS = 1u << 8;     // OnlyShadow: rayint misses (Tex.hs:89)
constexpr uint32_t RF_NOSHADOW = 1u << 9;  // NoShadow: shadow is False (Tex.hs:81)
constexpr uint32_t RF_PRIMLIST = 1u << 10; // on a list / Intersection record: every child's record is a primitive's (the generic tier answers an Instance
                                           // of such a list, and such an Intersection, in place)

constexpr uint32_t RF_RETEX = 1u << 11;    // on a Difference record: `Difference a b False` (difference_retexture, Csg.hs:29-30, 42-43): a carved surface keeps
                                           // the textures B's hit came with instead of taking A's at the point

constexpr uint32_t kBihItemsInPlace = 1u << 31;  // bihhdr[3h + 2].w: the tree's depth, and this bit when every item is a primitive or an Instance of primitives (flatten.hpp item_in_place)
enum BihLeafClass : uint32_t { BC_GENERIC = 0, BC_TRI = 1, BC_SPHERE = 2, BC_SIMPLE = 3, BC_CSG = 4 /* primitives and CSG over primitives */ };
constexpr uint32_t MESH_BRANCH = 0xffffffffu;  // count value marking "ref is a branch node"

enum DMatKind : uint32_t { DM_SURFACE = 0, DM_REFLECT = 1, DM_REFRACT = 2, DM_LAYERS = 3, DM_BLEND = 4, DM_WARP = 5 };

// A texture stack (a ray's accumulated `texs`, Tex.hs:53-74): material ids, innermost first, stored as id+1 (0 = end), in one
// 64-bit word -- 8 of them at 8 bits each in a scene of at most 254 materials (DScene::tex_bits = 8), 4 at 16 bits otherwise.
// (Records and root entries keep their one or two own Tex ids at 16 bits each: tex_from16 widens them.)
typedef uint64_t TexStack;
constexpr int kMaxTexDepth = 8;   // with 8-bit ids; 64 / tex_bits in general
constexpr int kMaxLights = 16;
constexpr int kFlatStack = 32;     // deepest BIH / Mesh tree the flat tier traverses (stack entries per lane: LDS part + global overflow columns)
constexpr int kFlatStackMesh = 64; // ... the Mesh PACKET walk may hold two entries per tree level (rt_device.hpp mesh_closest_wave)
constexpr int kGenericPacketStack = 24;  // LDS entries per lane of the generic tier's packet stack (rt_generic.hpp, vm_run's packet service): 18 KB a wave, eight waves per CU fit; the oak of GlomeView's default scene is 21 levels deep
constexpr int kGenericStack = 32;  // scratch traversal-stack entries per BIH/Mesh level in the generic tier
// the generic tier's frame stack (rt_generic.hpp): words per ray, and the frame sizes the host's commit-time estimate shares
constexpr int kVmWords = 2048 /* 768 until round 4; measured on GlomeView's default scene: +0.2 %, adaptive +1.4 % (profiles/r04_probes/vm_words_ab.txt) */, kVmHitWords = 17, kVmListR = 7 + kVmHitWords, kVmInstR = 10, kVmBoundR = 5, kVmIbR = 4, kVmDiffFixed = 10 + kVmHitWords,
              kVmIsectWords = 11, kVmBihFixedR = 12 + kVmHitWords, kVmBihFixedS = 12;
constexpr int kVmIsectChain = 8;   // Intersection frames the commit-time estimate allows for (a chain longer than the memory is caught at run time)
constexpr int kCsgMaxAdvance = 32; // ray-advance steps per Difference the generic tier's commit-time frame estimate allows for (at run time: its frame memory)
constexpr int kCsgFlatAdvance = 256; // ray-advance steps per CSG item of the flat tier whose distances are added back in the reference's own (nested) order; a ray that
                                     // advances more often keeps going (reference: unbounded recursion) with the further advances added to the last slot -- the same
                                     // sum in another order, an ulp of the depth
constexpr int kIsectFrames = 40;   // explicit frames for rayint_intersection's list recursion (flat tier's CSG items); when they run out an advance re-uses its frame
constexpr int kCsgRunaway = 1 << 20; // advances of ONE ray through ONE CSG item after which the launch reports GLOME_E_LIMIT instead of going on (every advance moves the
                                     // ray by at least delta = 1e-4, so a million of them is a ray crawling through 100 units of surfaces a hair apart: the reference
                                     // would still be recursing; a GPU wave must come back)
constexpr int kMaxTraceDepth = 8;  // maxdepth values the shading state machine has trace frames for (reference: any)
constexpr int kMaxMatNest = 4;     // Blend / AdditiveLayers nesting it has material frames for (reference: any)
constexpr int kMaxBatchFrames = 32;  // frames one render launch can carry (a launch costs ~0.3 ms besides its frames -- it ends with its slowest work items -- so the more the better: DESIGN.md 4.1b)

constexpr int kPairWords = 20;  // a pair record: 18 floats, the leaf's remaining count, the first triangle's record index (80 bytes)

struct F4 { float x, y, z, w; };
struct U4 { uint32_t x, y, z, w; };

struct DScene {
  const U4* recs;
  const F4* spheres;
  const F4* tris;
  const float* tripairs;  // the triangle BIHs' leaves as pair records (flatten.hpp emit_pairs): two triangles with every component of
                          // p1, e1, e2 as (A, B), the leaf's remaining count, the record index -- the packet walk's pair tests
  const F4* trinorms;
  const F4* boxes;
  const F4* planes;
  const F4* discs;
  const F4* quadrics;
  const F4* xfms;
  const F4* bihhdr;
  const F4* bihnodes;
  const F4* pknodes;  // the hand-written packet walk's copy of the triangle BIHs' nodes (flatten.hpp emit_bih): a branch child = byte
                      // offset | axis, a leaf child = byte offset of its first pair record | 3
  uint32_t pknodes_bytes;
  const F4* meshhdr;
  const F4* meshnodes;
  const F4* mtris;
  const U4* mtrimeta;
  const F4* mats;
  const F4* wlights;
  const uint32_t* matkids;
  const U4* entries;
  uint32_t n_entries;
  uint32_t root_rec;
  uint32_t tier;
  uint32_t n_mats;
  uint32_t tex_bits;  // 8 or 16: bits per id of a TexStack in this scene
  uint32_t pk_generic_cap;  // generic tier: entries of the per-wave LDS stack its kernels carry for packet walks of sphere BIHs
                            // (rt_generic.hpp, the packet service of vm_run); 0 = the scene has none
};

struct DCamera { float pos[3], fwd[3], up[3], right[3]; };
struct DLight { float pos[3], color[3], rad; int32_t shadow; };

struct DTile { int32_t x, y, w, h; uint32_t wave_base; uint32_t pix_base; };  // pix_base: offset of the tile in a dense payload

// a work item's 64 pixels: a block of kBlockW x kBlockH (8 x 8: the most coherent packet; 16 x 4 writes 64-byte row segments)
#ifndef GLOME_BLOCK_W
#define GLOME_BLOCK_W 8
#endif
constexpr int kBlockW = GLOME_BLOCK_W, kBlockH = 64 / GLOME_BLOCK_W;
constexpr uint32_t kQueueShards = 8;       // heads of a render launch's work queue (one per XCD)
constexpr uint32_t kQueueHeadStride = 32;  // words between heads: every head on its own 128-byte line
constexpr uint32_t kQueueChunk = 64;       // consecutive tickets that belong to one head (one 64x64 work tile of 8x8 blocks)
struct DCounters {  // device-side atomics, one block per launch slot
  unsigned long long rays_primary, rays_shadow, rays_secondary, bih_nodes, mesh_nodes, prim_tests;
  unsigned int heads[kQueueShards * kQueueHeadStride];  // persistent-kernel work queue heads (glome_device.hip TicketQueue)
  unsigned int dry_pad[31];
  unsigned int dry;        // mask of heads found empty (own line: written a few times per launch, read whenever a wave changes heads)
  unsigned int done_pad[31];
  unsigned int done;       // waves of the running launch that have left the queue (the last one resets the heads)
  unsigned int error;      // sticky: set when a device-side limit was hit (stack overflow guard, CSG cap); reset_counters stops short of it
  unsigned int dbg_pad;
  unsigned long long dbg[16];  // measurement builds only (GLOME_PKW_STAMPS: the packet walk's wait cycles by kind); glome_ctx_debug_words reads them
};

struct DRenderArgs {
  DScene S;
  DCamera cam;
  DLight lights[kMaxLights];
  int32_t nlights;
  int32_t width, height;
  int32_t fog;
  int32_t maxdepth;
  float thresholds[4];
  const DTile* tiles;  // owned tiles
  const uint32_t* tile_lut;  // tile of every 64th work item
  int32_t ntiles;
  uint32_t total_waves;
  uint32_t shard_cap;   // tickets per queue head of this launch (whole chunks; the last round of chunks may be padding)
  int32_t dense;       // 1: out5 is a dense tile payload (tile order, row major inside a tile) instead of a frame
  float* scratch;      // adaptive sampler: dense per-tile working buffer `v` (owned pixels * 5 floats)
  unsigned int* ss_cnt;  // adaptive sampler: queue heads (one per 128-byte line), then the mask of dry heads
  unsigned int* ss_done; // adaptive sampler: [tile * 8 + pass] regions of the tile's pass that are complete
  uint32_t ss_plane;     // adaptive sampler: pixels per channel plane of the working buffer `scratch` (r | g | b | a | depth planes)
  int32_t blocksize;     // adaptive sampler: tile edge (<= 65); work items are laid out for full-size tiles
  int8_t ss_rw[8], ss_rh[8];  // adaptive sampler: a work item of pass p covers ss_rw[p] x ss_rh[p] blocks of its tile
  float* out5;         // width*height*5
  uint32_t* packed;    // width*height or null
  DCounters* counters;
  // several independent frames in one launch (same scene and lights, one camera each): frame f's work items follow
  // frame f-1's in the queue, its pixels go frame_stride pixels further into out5 / packed
  int32_t nframes;
  uint32_t frame_stride;
  uint32_t chunks_per_frame;  // > 0: the frames of the launch are interleaved in the queue chunk by chunk (chunk c of every frame, then chunk
                              // c + 1 of every frame ...; a chunk = kQueueChunk items of ONE frame), chunks_per_frame = ceil(total_waves / kQueueChunk);
                              // 0: frame after frame
  int32_t want_counters;  // 0: nobody will read the ray / work counters of this launch -- the waves skip the flush
  int32_t debug_flags;    // GLOME_PROBE builds only (glome_device.hip render_loop)
  DCamera more_cams[kMaxBatchFrames - 1];
};

}  // namespace glome
