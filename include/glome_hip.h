/* glome_hip.h -- C ABI of libglome_hip.so: the MI355X (gfx950) ray-tracing core that replaces
 * glome's per-ray hot path behind glome's own constructor vocabulary.
 *
 * The reference (jimsnow/glome, Haskell) has no FFI boundary of its own; the seams this library
 * sits behind are (paths relative to the reference tree):
 *   - the `Solid` class methods  rayint / shadow / inside   GlomeTrace/Data/Glome/Solid.hs:146-166
 *   - the tile map               renderTiles                 GlomeView/Glome.hs:379-386
 *   - the scene constructors     sphere, triangle, box, ...  (cited per function below)
 * Each entry point cites the reference interface it replaces.  INTEGRATION.md shows the Haskell
 * `foreign import ccall` stubs a maintainer would add.
 *
 * Conventions
 *   - Every function returning `int` returns 0 on success, <0 (a glome_status) on error; builder
 *     functions returning int32_t return a node/material id >= 0, or <0 on error.  The message is
 *     read with glome_sb_last_error / glome_last_error.  No C++ exception crosses the boundary.
 *   - Scene constants cross as `double` (glome's `Flt = Double`, Vec.hs:9); the device computes in
 *     fp32.  Ray / hit / framebuffer arrays are caller-owned fp32 SoA buffers, valid for the call.
 *   - `_dev` variants take DEVICE pointers and run asynchronously on the context's HIP stream.
 *   - A glome_ctx owns one device + one stream and is single-threaded; distinct contexts may be
 *     driven from distinct threads.  A glome_scene is immutable after commit.
 *   - There is NO CPU fallback: every compute entry point fails with GLOME_E_NO_DEVICE when no
 *     gfx950 device is usable.  Builder and flatten-inspection calls are host-only.
 */
#ifndef GLOME_HIP_H
#define GLOME_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct glome_ctx glome_ctx;     /* one GPU + one HIP stream */
typedef struct glome_sb glome_sb;       /* scene builder: the host-side scene graph */
typedef struct glome_scene glome_scene; /* flattened scene resident in HBM */

enum glome_status {
  GLOME_OK = 0,
  GLOME_E_INVALID = -1,   /* bad argument / bad id */
  GLOME_E_SCENE = -2,     /* scene validation failed (infinite bound in bih, corrupt matrix, ...) */
  GLOME_E_NO_DEVICE = -3, /* no usable gfx950 device */
  GLOME_E_HIP = -4,       /* HIP runtime error */
  GLOME_E_LIMIT = -5      /* scene exceeds a device-side limit (texture stack, frame memory, ...) */
};

/* ---- context ---- */
/* The context launches on a stream of its own, created as a blocking stream: it is ordered against the device's default
 * (null) stream in both directions, so buffers a caller prepared there may be handed to the *_dev entry points directly.
 * Work on the caller's own non-blocking streams is the caller's to order (or the stream is given to glome_ctx_use_slot). */
glome_ctx* glome_ctx_create(int device_ordinal); /* NULL on failure; see glome_global_error() */
void glome_ctx_destroy(glome_ctx*);
const char* glome_last_error(const glome_ctx*);
const char* glome_global_error(void);  /* error text when no ctx/sb exists yet */
void* glome_ctx_stream(glome_ctx*);    /* the hipStream_t, for interop */
int glome_ctx_synchronize(glome_ctx*);
/* Run on an external HIP stream (e.g. torch's current stream) instead of the context's own; NULL restores it. */
int glome_ctx_use_stream(glome_ctx*, void* hip_stream);
/* Several frames in flight: a context has 8 launch slots, each with its own work queue, counters and workspaces.  Select
 * (stream, slot) before a launch; launches that may overlap in time must use different slots (and streams). */
int glome_ctx_use_slot(glome_ctx*, void* hip_stream, int slot);
/* Per-launch kernel timing without extra synchronisation: between begin and end every render launch records its own
 * HIP-event pair on the context's stream; end synchronises once and returns the number of launches, writing their
 * durations (ms) to ms_out[0..cap). */
int glome_ctx_timing_begin(glome_ctx*, int max_launches);
/* The same, but only every `stride`-th render launch carries an event pair: event records are extra packets on the
 * launch stream and cost frame rate when a frame is a fraction of a millisecond. */
int glome_ctx_timing_begin_sampled(glome_ctx*, int max_launches, int stride);
int glome_ctx_timing_end(glome_ctx*, float* ms_out, int cap);
/* Waves per CU of the persistent render launches.  0 (default): sized by the launch's work, for several launches in flight
 * on separate slots (each takes a share of the CUs' wave slots); n > 0: n per CU, resources permitting -- what a launch
 * that runs ALONE wants (24 for the packet instances).  The reference has no such knob (GHC's +RTS -N is the nearest). */
int glome_ctx_set_grid_per_cu(glome_ctx*, int waves_per_cu);
int glome_ctx_device_info(glome_ctx*, char* name, int cap, int* cu_count, int* warp_size);
/* debugging aid, no reference counterpart: 16 words a measurement build of the library accumulates on the device (the stock
   build leaves them zero); synchronises the context's stream */
int glome_ctx_debug_words(glome_ctx*, uint64_t* out16);
int glome_ctx_debug_reset(glome_ctx*);  /* measurement builds (-DGLOME_PROBE): the current slot's debug words back to their start values */

/* ---- transforms: Xfm = forward 3x4 (12 doubles, row major) + inverse 3x4 (12 doubles) ---- */
int glome_xfm_translate(const double v[3], double out[24]);                          /* Vec.hs:564-567 */
int glome_xfm_scale(const double v[3], double out[24]);                              /* Vec.hs:571-574 */
int glome_xfm_rotate(const double axis[3], double angle_rad, double out[24]);        /* Vec.hs:577-598 */
int glome_xfm_xyz_to_uvw(const double u[3], const double v[3], const double w[3], double out[24]); /* Vec.hs:602-622 */
int glome_xfm_compose(const double* xfms /* n*24 */, int n, double out[24]);         /* Vec.hs:461-462 */

/* ---- scene builder: one call per reference constructor ---- */
glome_sb* glome_sb_new(void);
void glome_sb_free(glome_sb*);
const char* glome_sb_last_error(const glome_sb*);
int32_t glome_sb_sphere(glome_sb*, const double c[3], double r);                                 /* Sphere.hs:15-17 */
int32_t glome_sb_triangle(glome_sb*, const double p[9]);                                         /* Triangle.hs:18-20 */
int32_t glome_sb_trianglenorm(glome_sb*, const double p[9], const double n[9]);                  /* Triangle.hs:34-35 */
int32_t glome_sb_box(glome_sb*, const double a[3], const double b[3]);                           /* Box.hs:12-15 */
int32_t glome_sb_plane(glome_sb*, const double pt[3], const double n[3]);                        /* Plane.hs:17-20 */
int32_t glome_sb_plane_offset(glome_sb*, const double n[3], double off);                         /* Plane.hs:24-25 */
int32_t glome_sb_disc(glome_sb*, const double pos[3], const double n[3], double r);              /* Cone.hs:29-31 */
int32_t glome_sb_cylinder(glome_sb*, const double p1[3], const double p2[3], double r);          /* Cone.hs:40-48 */
int32_t glome_sb_cone(glome_sb*, const double p1[3], double r1, const double p2[3], double r2);  /* Cone.hs:52-67 */
int32_t glome_sb_group(glome_sb*, const int32_t* ids, int n);                                    /* Solid.hs:293-296 */
int32_t glome_sb_transform(glome_sb*, int32_t id, const double* xfms /* n*24 */, int n);         /* Solid.hs:184,235 */
int32_t glome_sb_difference(glome_sb*, int32_t a, int32_t b);                                    /* Csg.hs:26-27 */
int32_t glome_sb_difference_retexture(glome_sb*, int32_t a, int32_t b);                          /* Csg.hs:29-30: `Difference a b False` -- the surface hollowed out by b keeps b's textures (Csg.hs:42-43) */
int32_t glome_sb_intersection(glome_sb*, const int32_t* ids, int n);                             /* Csg.hs:64-65 */
int32_t glome_sb_bih(glome_sb*, const int32_t* ids, int n);                                      /* Bih.hs:309-324 */
/* tris: 8 ints per triangle = a b c na nb nc tex tag (-1 = none), Mesh.hs:27-29; mats = the mesh's texture vector */
int32_t glome_sb_mesh(glome_sb*, const double* verts, int nv, const double* norms, int nn,
                      const int32_t* tris, int nt, const int32_t* mats, int nm);                 /* Mesh.hs:50-55 */
int32_t glome_sb_tex(glome_sb*, int32_t id, int32_t material);                                   /* Tex.hs:33-34 */
int32_t glome_sb_tag(glome_sb*, int32_t id);                                                     /* Tex.hs:38-39 (tags feed picking only) */
int32_t glome_sb_noshadow(glome_sb*, int32_t id);                                                /* Tex.hs:43 */
int32_t glome_sb_onlyshadow(glome_sb*, int32_t id);                                              /* Tex.hs:48 */
int32_t glome_sb_bound_object(glome_sb*, int32_t bounding, int32_t bounded);                     /* Bound.hs:27-28 */
int32_t glome_sb_innerbound(glome_sb*, int32_t inner, int32_t outer);                            /* Bound.hs:116 */
int32_t glome_sb_flatten_transform(glome_sb*, int32_t id);  /* `SolidItem (flatten_transform s)`, Solid.hs:192,273 */
int32_t glome_sb_tolist(glome_sb*, int32_t id);             /* `tolist`, Solid.hs:177,230: a list node of the flattened items */
/* the [SolidItem] that `tolist id` yields, as node ids: writes up to cap of them and returns their number -- what a host passes
   on to a constructor that takes a list, e.g. TestScene.hs:109's `bih (tolist (SolidItem (flatten_transform tree)))` */
int32_t glome_sb_list_items(glome_sb*, int32_t id, int32_t* out, int32_t cap);
/* A whole scene in the Neutral File Format of Eric Haines' SPD (GlomeTrace/Data/Glome/Spd.hs:89-254): statements v (camera),
 * l (light, colour optional), b (background), f (fill -> Surface clr (1-T) 0 kd ks shine), s (sphere), c (cone), p / pp
 * (polygon / polygon with normals -> a triangle fan); `#` comments.  Returns the root node -- `bih` of one
 * `tex (bih prims) fill` per fill, in the reference's (reversed) order -- and fills camera (from, at, up, angle), up to
 * max_lights lights (position + rgb each; *n_lights = how many the file holds) and the background colour. */
int32_t glome_sb_load_nff(glome_sb*, const char* text, double cam_from_at_up_angle[10], double* light_pos_rgb, int32_t max_lights, int32_t* n_lights,
                          double bg_rgb[3]);

/* The text GlomeView prints for a scene (`show geom`, SDLK_s, Glome.hs:431) as an interchange format: derived `Show` of
 * the solids' constructors (Sphere.hs:11, Triangle.hs:13-14, Box.hs:10, Cone.hs:21-23, Plane.hs:11, Csg.hs:14-15,
 * Bound.hs:20,95, Tex.hs:27-29, Solid.hs:386, Bih.hs:51-57, Vec.hs:105,407-414,646) with the hand-written instances for
 * SolidItem ("SI ...", Solid.hs:277), Texture ("Texture", Solid.hs:101), Tag (Tex.hs:50) and Mesh (Mesh.hs:44).
 * glome_sb_show writes node `id` in that text (returns its length; at most cap-1 characters + NUL go to buf, buf may
 * be NULL to ask for the length).  glome_sb_load_show reads such a text -- e.g. the dump of a real GHC build of
 * TestScene.hs -- into the builder, Bih and Mesh trees exactly as printed, and returns the root.  Materials are closures
 * on the Haskell side and print as "Texture": the k-th `Tex` of the text (reading order) gets tex_materials[k], the
 * rest default_material (-1: fail); *n_tex = how many the text holds.  Tags and mesh vertex normals are not printed by
 * the reference and do not survive (a mesh with normals is refused); `Difference _ _ False` reads as glome_sb_difference_retexture. */
long glome_sb_show(glome_sb*, int32_t id, char* buf, long cap);
/* the material ids of the `Tex` constructors of that text, in reading order (what the text itself cannot carry): returns
 * their number, writes at most cap of them */
long glome_sb_show_tex_materials(glome_sb*, int32_t id, int32_t* mats, long cap);
int32_t glome_sb_load_show(glome_sb*, const char* text, const int32_t* tex_materials, int32_t n_tex_materials, int32_t default_material, int32_t* n_tex);

/* materials (the defunctionalised `Material`, Shader.hs:43-52; a texture is a material id = t_uniform, Shader.hs:55-56) */
int32_t glome_sb_material_surface(glome_sb*, const double color[3], double alpha, double amb, double kd, double ks, double shine);
int32_t glome_sb_material_reflect(glome_sb*, double refl);
int32_t glome_sb_material_refract(glome_sb*, double refl, double refr, double ior);
int32_t glome_sb_material_layers(glome_sb*, const int32_t* mats, int n);
int32_t glome_sb_material_blend(glome_sb*, int32_t a, int32_t b, double weight);
/* A Blend whose weight is a solid texture function of the hit position (GlomeVec/Data/Glome/Texture.hs) -- the
 * defunctionalised form of the closures t_mottled / t_stripe (TestScene.hs:214-234):
 *   GLOME_WEIGHT_PERLIN            weight = perlin (vscale pos params[0])                      (Texture.hs:109-117)
 *   GLOME_WEIGHT_STRIPE_SQUARE / _TRIANGLE / _SINE   weight = wave (vdot pos params[0..2])     (Texture.hs:11-41) */
#define GLOME_WEIGHT_PERLIN 1
#define GLOME_WEIGHT_STRIPE_SQUARE 2
#define GLOME_WEIGHT_STRIPE_TRIANGLE 3
#define GLOME_WEIGHT_STRIPE_SINE 4
int32_t glome_sb_material_blend_fn(glome_sb*, int32_t a, int32_t b, int32_t weight_fn, const double* params4);
/* Warp frame scene' lights' xfm (Shader.hs:47-50, 157-175): a hit on this material traces `frame` with the hit's own ray (the
 * ray as the primitive saw it -- local coordinates inside a `transform`, Solid.hs:388-403) and the other scene with the
 * warped ray, up to the frame's depth, and shows whichever is nearer.  `scene` is a node, or -1 for the root the scene
 * is committed with (the portal of TestScene.hs:152-181 looks into the scene it stands in: in Haskell laziness ties that
 * knot).  `lights` are that scene's lights.  The closure `Ray -> Rayint -> Ray` crosses the ABI as its one shape in the
 * reference, \ray hit -> xfm_ray M (Ray (pos hit) (vnorm (dir ray))) (TestScene.hs:166-172): xfm = M, 24 doubles like
 * every transform here.  Scenes with a Warp material render on the generic tier. */
struct glome_light;
int32_t glome_sb_material_warp(glome_sb*, int32_t frame, int32_t scene, const struct glome_light* lights, int nlights, const double xfm[24]);
/* host-side inspection (no GPU needed) */
int glome_sb_primcount(glome_sb*, int32_t id, long out3[3]);  /* primcount, Solid.hs:197,251 */
int glome_sb_bound(glome_sb*, int32_t id, double out6[6]);    /* bound, Solid.hs:171 */
/* Preorder dump of the BIH built for node `id` (axis = -1 marks a leaf; nleaf = its item count;
 * leaf_prims = builder ids of leaf items in order).  Returns the node count, <0 on error. */
long glome_sb_bih_dump(glome_sb*, int32_t id, long cap, double* lsplit, double* rsplit, int* axis, int* nleaf,
                       int32_t* leaf_prims, long cap_prims);

/* `bih` (Bih.hs:309-324) built on the GPU of `ctx`: the same node as glome_sb_bih over the same ids -- the tree of the
 * reference's build_rec (Bih.hs:211-285), node for node and bit for bit -- made level by level by four kernels per tree
 * level instead of by recursion on the host (100k triangles: see DESIGN.md).  *gpu_ms (may be NULL) = device time of the
 * build.  Needs the HIP half of the library (a context); the host builder stays the default and the checker. */
int32_t glome_sb_bih_dev(glome_ctx*, glome_sb*, const int32_t* ids, int32_t n, float* gpu_ms);

/* `mesh` (Mesh.hs:50-134) with its two-box BVH (build_tree, Mesh.hs:69-113) built on the GPU of `ctx`: arguments and
 * result as glome_sb_mesh, the tree the host builder makes (boxes, leaf order). */
int32_t glome_sb_mesh_dev(glome_ctx*, glome_sb*, const double* verts, int nv, const double* norms, int nn, const int32_t* tris, int nt, const int32_t* mats, int nm,
                          float* gpu_ms);

/* ---- commit: validate + flatten to packed SoA pools + upload to HBM ---- */
glome_scene* glome_scene_commit(glome_ctx*, glome_sb*, int32_t root);
void glome_scene_release(glome_scene*);
typedef struct glome_scene_info {
  int32_t tier;            /* 0 = flat fast path (LDS-stack kernels), 1 = generic interpreter */
  int32_t nesting_depth;   /* composite nesting depth of the generic graph */
  int64_t n_records, n_bih_nodes, n_mesh_nodes, n_triangles, n_spheres, n_other_prims, n_xfms, n_materials;
  int32_t max_bih_depth, max_mesh_depth;
  int64_t device_bytes;
} glome_scene_info;
int glome_scene_get_info(const glome_scene*, glome_scene_info* out);

/* ---- per-ray seams (Solid.hs:146-166), host buffers ---- */
/* closest hit: t < 0 marks a miss (RayMiss); prim = builder id of the primitive hit; tex8 = the hit's
 * texture stack (the ids of glome_sb_material), innermost first, -1 padded: GLOME_TEX_WORDS (8) int32 PER RAY -- the buffer is
 * n * GLOME_TEX_WORDS words (until round 3 it was 4 per ray: a caller built against that header must be rebuilt; glome_tex_words()
 * returns what the loaded library writes, for a binding that wants to check at run time).  Any output pointer may be NULL. */
#define GLOME_TEX_WORDS 8
int glome_tex_words(void);
int glome_rayint_batch(glome_scene*, size_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                       const float* dy, const float* dz, const float* tmax, float* t, int32_t* prim, float* nx,
                       float* ny, float* nz, int32_t* tex8);
int glome_shadow_batch(glome_scene*, size_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                       const float* dy, const float* dz, const float* tmax, uint8_t* occluded);
int glome_inside_batch(glome_scene*, size_t n, const float* px, const float* py, const float* pz, uint8_t* inside);
/* the same on device pointers, asynchronous on the ctx stream */
int glome_rayint_batch_dev(glome_scene*, size_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                           const float* dy, const float* dz, const float* tmax, float* t, int32_t* prim, float* nx,
                           float* ny, float* nz, int32_t* tex8);
int glome_shadow_batch_dev(glome_scene*, size_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                           const float* dy, const float* dz, const float* tmax, uint8_t* occluded);

/* ---- whole-frame seam (renderTiles, Glome.hs:379-386; Scene tuple, TestScene.hs:15) ---- */
typedef struct glome_camera { float pos[3], fwd[3], up[3], right[3]; } glome_camera; /* Scene.hs:35 */
int glome_camera_lookat(const double pos[3], const double at[3], const double up[3], double angle_deg,
                        glome_camera* out); /* camera, Scene.hs:48-57 */
typedef struct glome_light {               /* Light, Shader.hs:13-23; falloff is fixed to 1/d^2 as `light` builds it */
  float pos[3], color[3], rad;
  int32_t shadow;
} glome_light;
enum { GLOME_MODE_TILE = 0 /* renderTile, Glome.hs:162-176 */, GLOME_MODE_SUBSAMPLE = 1 /* renderTileSubsample, :226-323 */ };
typedef struct glome_render_params {
  int32_t width, height;
  int32_t mode;          /* GLOME_MODE_* */
  int32_t blocksize;     /* tile edge, Glome.hs:116 (65) */
  int32_t maxdepth;      /* Glome.hs:25 (3); 1..8 supported */
  int32_t fog;           /* 1: TILE mode stores (r + depth/400, g, b, a, depth) exactly as renderTile does (Glome.hs:174,
                            Q20: a miss stores r = 2500); 0 -- the DEFAULT, a deliberate deviation from renderTile -- stores
                            the tuple get_color returns (Glome.hs:53-55) before that debug term */
  float thresholds[4];   /* Glome.hs:221-224 */
  int32_t tile_first, tile_stride; /* shard: render tiles tile_first, tile_first+tile_stride, ... (x-major order) */
  int32_t faithful;      /* 1: BIH traversal without ordered early-out, exactly as Bih.hs:332-368 visits nodes */
  int32_t count_work;    /* 1: count node visits / primitive tests (slower -- counting kernel instances; implied by faithful).
                            The generic tier always traverses with early-out (a ray that is not unit length excepted). */
  int32_t rank0_share_pct; /* shards of tile_stride ranks: 0 (or 100) = tile k belongs to rank k mod tile_stride; 1..99 = the
                            weight of rank 0, which also receives and blits every frame, in percent of one other rank's: rank
                            0 owns pct / (pct + 100 (tile_stride - 1)) of the tiles, the others split the rest evenly (an
                            evenly interleaved repeating pattern every rank derives from (tile_stride, percentage), every
                            percent a different one; parMap over tiles, Glome.hs:385, has no such notion -- a tile is a tile
                            whoever renders it) */
} glome_render_params;
void glome_render_params_default(glome_render_params*);
typedef struct glome_stats {
  uint64_t rays_primary, rays_shadow, rays_secondary; /* rays actually traversed */
  uint64_t bih_nodes, mesh_nodes, prim_tests;         /* only when count_work */
  float kernel_ms;                                    /* HIP-event time of the render kernel(s) on the ctx stream */
  int32_t n_tiles, n_pixels;
} glome_stats;
/* rgbad: width*height*5 floats (r,g,b,a,depth per pixel, row major; pixels of tiles this call does not own are
 * left untouched); packed: width*height 0x00RRGGBB words as blitTile/rgbf produce (Glome.hs:353-358, 107-110) or NULL. */
int glome_render(glome_scene*, const glome_camera*, const glome_light* lights, int nlights,
                 const glome_render_params*, float* rgbad, uint32_t* packed, glome_stats*);
/* Device-pointer variant.  Asynchronous on the ctx stream unless stats != NULL (then it synchronizes to read
 * the counters and the event timer).  rgbad_dev may be NULL when packed_dev is given: only the displayable pixels are
 * then written (trace and blitTile fused; the float tuple never leaves registers). */
int glome_render_dev(glome_scene*, const glome_camera*, const glome_light* lights, int nlights,
                     const glome_render_params*, float* rgbad_dev, uint32_t* packed_dev, glome_stats*);
/* Render the tiles owned by (params->tile_first, params->tile_stride) straight into a dense tile payload (what a
 * rank sends to the gather): tiles in owned order, row major inside a tile, 5 floats per pixel = the reference's
 * `Tile Rect (UV.Vector TColor)` (Glome.hs:153-154). */
int glome_render_tiles_dev(glome_scene*, const glome_camera*, const glome_light* lights, int nlights,
                           const glome_render_params*, float* payload_dev, glome_stats*);
/* The same render, but only the displayable pixel leaves the kernel: payload_dev receives one packed 0x00RRGGBB word per
 * owned pixel (rgbf of the premultiplied colour -- what blitTile, Glome.hs:353-358, pokes into GlomeView's framebuffer),
 * tiles in owned order, row major inside a tile.  This is the payload of the multi-GPU framebuffer gather (4 bytes per
 * pixel instead of 20).  GLOME_MODE_TILE and GLOME_MODE_SUBSAMPLE alike. */
int glome_render_tiles_packed_dev(glome_scene*, const glome_camera*, const glome_light* lights, int nlights,
                                  const glome_render_params*, uint32_t* payload_dev, glome_stats*);
/* Several independent frames in ONE launch (an animation's next views: same scene and lights, cams[0..nframes), at
 * most 8).  Frame f's pixels land frame_stride_pixels words after frame f-1's: rows of a dense tile payload
 * (..._tiles_packed_batch_dev; stride >= this rank's payload size) or whole packed framebuffers (..._packed_batch_dev;
 * stride >= width*height).  A rank's share of one frame is a few thousand work items -- too little to fill the GPU
 * beyond its slowest item; a batch restores long launches.  Both render modes (the adaptive sampler of a batch of 4 or
 * more frames works in larger regions per work item: fewer, fuller sample packets; the frames are unchanged). */
int glome_render_tiles_packed_batch_dev(glome_scene*, const glome_camera* cams, int nframes, const glome_light* lights, int nlights,
                                        const glome_render_params*, uint32_t* payload_dev, int64_t frame_stride_pixels, glome_stats*);
int glome_render_packed_batch_dev(glome_scene*, const glome_camera* cams, int nframes, const glome_light* lights, int nlights,
                                  const glome_render_params*, uint32_t* packed_dev, int64_t frame_stride_pixels, glome_stats*);
/* ---- the whole-frame seam on several GPUs driven by ONE process (renderTiles' parMap over tiles + blitTile, Glome.hs:379-386) ----
 * scenes[i] = the same scene committed on context i (a context per GPU; rank 0's GPU receives the frame).  Tile k of the
 * frame -- 64x64 work tiles in renderTile mode, the 65x65 reference tiles in adaptive mode (whose pixels depend on the tile
 * map, Q21) -- belongs to rank k mod n, or to the rank the weighted pattern of glome_render_params.rank0_share_pct gives it.
 * A call renders nframes <= 32 views (one in adaptive mode), every rank its tiles of all of them in one launch.  How the pixels
 * reach packed_dev (frame f at f * width * height words) is the TRANSPORT (glome_multi_transport says which was taken):
 *   2 "direct"     every rank's render kernel stores its tiles' packed 0x00RRGGBB pixels straight into packed_dev on rank 0's GPU
 *                  (4 bytes per pixel over xGMI); rank 0's stream waits for the others' launches.  No payload, no exchange, no blit,
 *                  and every rank owns a fair share of the tiles (rank0_share_pct is ignored).  Needs every rank's device to reach rank
 *                  0's memory (the same device, or peer access); asked for and not possible -> "rccl" / "peer-copy" as below.
 *   1 "rccl"       ranks render into packed payloads, which move to rank 0's GPU with RCCL send / recv in one group (librccl.so is
 *                  dlopen'ed; the ranks must sit on distinct devices), and one launch there blits the frames into packed_dev.
 *   0 "peer-copy"  the same with peer copies on rank 0's stream.
 * Asynchronous; glome_multi_synchronize waits for all ranks and reports device-side limits.
 * The RCCL branch needs distinct devices and has not run on real RCCL with more than one rank on this pool (one-GPU boxes): it
 * is exercised against a stand-in transport whose send / recv pairs are stream-ordered device copies (tests/rcclstub). */
/* ---- a framebuffer several processes render into (one process per GPU, glome_amd/dist.py) ----
 * glome_ipc_alloc: device memory on this context's GPU (zeroed) and a 64-byte handle another process passes to glome_ipc_open to
 * map it; a rank then renders its tiles of a frame with glome_render_packed_batch_dev (tile_first / tile_stride set, frame
 * layout) straight into the mapping -- its kernel's stores cross xGMI, nothing is gathered or blitted.  glome_ipc_close: the
 * owner frees, the others unmap.  (HIP IPC; on this driver dmabuf handles: HSA_ENABLE_IPC_MODE_LEGACY=0.) */
int glome_ipc_alloc(glome_ctx*, size_t bytes, void** dev_ptr, unsigned char* handle64);
int glome_ipc_open(glome_ctx*, const unsigned char* handle64, void** dev_ptr);
int glome_ipc_close(glome_ctx*, void* dev_ptr, int owner);
typedef struct glome_multi glome_multi;
glome_multi* glome_multi_create(glome_scene* const* scenes, int n, const glome_render_params*, int transport); /* 0 / 1 / 2 as above; NULL: glome_global_error() */
void glome_multi_destroy(glome_multi*);
int glome_multi_render(glome_multi*, const glome_camera* cams, int nframes, const glome_light* lights, int nlights, uint32_t* packed_dev);
int glome_multi_synchronize(glome_multi*);
const char* glome_multi_transport(const glome_multi*); /* "direct", "rccl", "peer-copy" or "none" (one rank) */
const char* glome_multi_last_error(const glome_multi*);
/* One frame into a host framebuffer (width * height words): create, render, synchronize, copy, destroy. */
int glome_render_multi(glome_scene* const* scenes, int n, const glome_camera*, const glome_light* lights, int nlights,
                       const glome_render_params*, uint32_t* packed);
/* Tile payload transport for multi-GPU sharding (Tile = Rect + pixel vector, Glome.hs:153-154).
 * pack: copy this rank's owned tiles from a full frame into a dense payload (tiles in owned order, row major
 * inside a tile, 5 floats per pixel).  blit: scatter a payload of the tiles owned by (tile_first, tile_stride)
 * back into a full frame (blitTile, Glome.hs:353-358).  glome_tiles_payload_floats gives the payload size. */
int64_t glome_tiles_payload_floats(const glome_render_params*, int tile_first, int tile_stride);
/* Host-only: the tiles owned by (tile_first, tile_stride) in renderTiles' order (Glome.hs:382-384).  Writes 5 ints per
 * tile (x, y, w, h, pixel offset of the tile inside the dense payload) and returns the tile count. */
int glome_tiles_layout(const glome_render_params*, int tile_first, int tile_stride, int32_t* xywh_base, int cap);
int glome_tiles_pack_dev(glome_ctx*, const glome_render_params*, const float* rgbad_dev, float* payload_dev);
int glome_tiles_blit_dev(glome_ctx*, const glome_render_params*, int tile_first, int tile_stride,
                         const float* payload_dev, float* rgbad_dev, uint32_t* packed_dev);
/* After the gather: `gathered_dev` holds `world` payload slabs of `stride_floats` floats each (rank r's payload at
 * gathered_dev + r * stride_floats).  One launch blits every rank's tiles into the frame. */
int glome_tiles_blit_all_dev(glome_ctx*, const glome_render_params*, int world, const float* gathered_dev, int64_t stride_floats,
                             float* rgbad_dev, uint32_t* packed_dev);
/* The packed-pixel form: `gathered_dev` holds `world` slabs of `stride_pixels` words (glome_render_tiles_packed_dev
 * payloads); one launch writes every rank's tiles into the packed framebuffer (width*height words). */
int glome_tiles_blit_all_packed_dev(glome_ctx*, const glome_render_params*, int world, const uint32_t* gathered_dev, int64_t stride_pixels,
                                    uint32_t* packed_dev);
/* The same for the nframes frames of a batch in one launch: frame f's payload starts f * payload_frame_stride words into
 * every rank's slab, its framebuffer f * out_frame_stride words after packed_dev. */
int glome_tiles_blit_all_packed_batch_dev(glome_ctx*, const glome_render_params*, int world, const uint32_t* gathered_dev, int64_t stride_pixels,
                                          int nframes, int64_t payload_frame_stride, uint32_t* packed_dev, int64_t out_frame_stride);

#ifdef __cplusplus
}
#endif
#endif /* GLOME_HIP_H */
