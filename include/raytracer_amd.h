/*
 * raytracer_amd.h -- flat C ABI of the MI355X (gfx950) render path.
 *
 * Everything here is plain C: POD structs, pointers and sizes.  The library
 * behind it (libraytracer_amd.so) is hand-written HIP; it needs a GPU and
 * returns RT_ERR_NO_DEVICE when there is none (there is no CPU fallback).
 *
 * The interface replaces, one for one, what the reference's only device entry
 * point does for its caller:
 *
 *     extern "C" void launch_kernel(TKSceneData *data, ShadingType stype,
 *                                   int width, int height, uchar4 *output,
 *                                   int sampleCount);       (kernel.h:30-31)
 *
 * launch_kernel itself (same symbol, same argument order) is exported too; it
 * is declared in raytracer_tokens.h because its first argument is a C++ object.
 *
 * Conventions shared with the reference:
 *   - matrices are 16 floats, column-major (GLM layout): m[col*4 + row];
 *   - the frame is row-major, index y*width + x, row 0 is the BOTTOM row,
 *     4 bytes per pixel {r, g, b, 255}                      (kernel.cu:545-570);
 *   - primitive ids number the BVH primitives in the order the reference
 *     creates them: spheres, triangles, boxes, smooth triangles, each in
 *     input order (kernel.cu:419-456); planes follow, id = n_bvh_prims + k
 *     (planes are not in the BVH: kernel.cu:411-416, bvh.cpp:220-222).
 */
#ifndef RAYTRACER_AMD_H
#define RAYTRACER_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------ */
enum {
   RT_OK = 0,
   RT_ERR_BAD_ARGUMENT = 1,   /* null pointer, non-positive size, ...            */
   RT_ERR_BAD_SAMPLE_COUNT = 2, /* sampleCount is not a perfect square (kernel.cu:707-712) */
   RT_ERR_NO_DEVICE = 3,      /* no HIP device / HIP runtime failure at init      */
   RT_ERR_HIP = 4,            /* a HIP call failed; rt_last_error() has the text  */
   RT_ERR_UNSUPPORTED = 5,    /* e.g. a .bmp that is not 24-bit uncompressed       */
   RT_ERR_PARSE = 6,          /* scene file missing or malformed                  */
   RT_ERR_IO = 7,
   RT_ERR_NO_CAMERA = 8,      /* the reference dereferences data->camera unchecked (kernel.cu:716) */
   RT_ERR_TOO_LARGE = 9
};

/* ShadingType of kernel.h:22 */
enum { RT_PHONG = 0, RT_COOK_TORRANCE = 1 };

/* How the two uniform draws of generateCameraRays (kernel.cu:488-489) are made.
 * The reference draws them from cuRAND XORWOW states, which cannot be
 * reproduced without cuRAND; RT_JITTER_HASH is this library's own
 * counter-based generator (a 32-bit integer hash of the sample index and seed), RT_JITTER_NONE pins both
 * draws to 0.5 so that the sample sits in the centre of its stratum. */
enum { RT_JITTER_NONE = 0, RT_JITTER_HASH = 1 };

/* Output layouts of rt_render_device */
enum {
   RT_OUT_FRAME = 0,   /* full W*H frame, row-major; only this call's tiles are written */
   RT_OUT_TILES = 1    /* compact: this call's tiles back to back, each RT_TILE x RT_TILE pixels row-major */
};

#define RT_TILE 32          /* pixels per tile edge (tiles are the multi-GPU sharding unit) */
#define RT_MAX_RECURSE 6    /* kMaxRecurse, kernel.h:23 */

/* ---- scene description (host, POD) ------------------------------------- */

/* Material.h:8-20 (field order follows the Material constructor call sites,
 * kernel.cu:414-450): pigment colour, pigment filter, finish terms. */
typedef struct rt_material {
   float clr[3];
   float alpha;      /* pigment filter (rgbf 4th component) */
   float amb, dif, spec, rough, refl, refr, ior;
   int32_t tex_id;   /* -1 = untextured (NO_TEXTURE, Material.h); else an index into rt_scene_desc.textures */
} rt_material;

/* TKModifier (TokenData.h:46-51) with pigment+finish folded into rt_material. */
typedef struct rt_modifier {
   rt_material mat;
   float trans[16];      /* object -> world */
   float inv_trans[16];  /* world -> object */
} rt_modifier;

typedef struct rt_sphere   { float c[3]; float r; rt_modifier mod; } rt_sphere;              /* TKSphere   TokenData.h:65-69 */
typedef struct rt_triangle {                                                                /* TKTriangle TokenData.h:83-87 */
   float p1[3], p2[3], p3[3];
   float vt1[2], vt2[2], vt3[2];   /* texture coordinates per vertex (zero when the scene gives none) */
   rt_modifier mod;
} rt_triangle;
typedef struct rt_box      { float p1[3], p2[3]; rt_modifier mod; } rt_box;                 /* TKBox      TokenData.h:60-63 */
typedef struct rt_smooth_triangle {                                                         /* TKSmoothTriangle TokenData.h:89-94 */
   float p1[3], p2[3], p3[3];
   float n1[3], n2[3], n3[3];
   float vt1[2], vt2[2], vt3[2];
   rt_modifier mod;
} rt_smooth_triangle;
typedef struct rt_plane    { float n[3]; float d; rt_modifier mod; } rt_plane;              /* TKPlane    TokenData.h:71-75 */
typedef struct rt_light    { float pos[3]; float clr[3]; } rt_light;                        /* TKPointLight TokenData.h:53-58 */
typedef struct rt_camera   { float pos[3], up[3], right[3], look_at[3]; } rt_camera;        /* TKCamera   TokenData.h:11-17 */

/* An image for pigment { image_map "file" } (POVRayParser.cpp:491-503, kernel.cu:32-64, 579-598).
 * rgba: width*height texels of 4 bytes (r, g, b, 0), row 0 = the first row stored in the .bmp, i.e. the
 * bottom row of the picture.  A description returned by rt_parse_pov carries the names only (rgba NULL):
 * attach the pixels with rt_load_bmp before rt_scene_create.  The reference itself ignores the name and
 * binds "blitz.bmp" from the working directory to every textured primitive. */
typedef struct rt_texture {
   int32_t width, height;
   uint8_t *rgba;
   char name[128];
} rt_texture;

typedef struct rt_scene_desc {
   rt_camera camera;
   int32_t has_camera;
   int32_t n_spheres;          const rt_sphere *spheres;
   int32_t n_triangles;        const rt_triangle *triangles;
   int32_t n_boxes;            const rt_box *boxes;
   int32_t n_smooth_triangles; const rt_smooth_triangle *smooth_triangles;
   int32_t n_planes;           const rt_plane *planes;
   int32_t n_lights;           const rt_light *lights;
   int32_t n_textures;         rt_texture *textures;
} rt_scene_desc;

/* ---- render parameters / statistics ------------------------------------ */
typedef struct rt_render_params {
   int32_t width, height;
   int32_t sample_count;   /* must be a perfect square; s = sqrt(sample_count) strata per pixel edge */
   int32_t shading;        /* RT_PHONG / RT_COOK_TORRANCE */
   int32_t jitter;         /* RT_JITTER_* */
   uint32_t seed;          /* RT_JITTER_HASH seed */
   int32_t max_recurse;    /* 1..RT_MAX_RECURSE; 0 means RT_MAX_RECURSE */
   /* image-tile sharding: this call renders tiles tile_first, tile_first+tile_stride, ...
    * (tile index = ty * tiles_x + tx, tiles_x = ceil(width/RT_TILE)).  0/0 or 0/1 = whole frame. */
   int32_t tile_first, tile_stride;
   int32_t out_layout;     /* RT_OUT_FRAME / RT_OUT_TILES */
} rt_render_params;

typedef struct rt_stats {
   /* every closest-hit / any-hit query that was traversed, by kind */
   uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract;
   uint64_t rays_total;
   uint64_t shadow_skipped;     /* shadow feelers whose self-test missed: the reference still walks the tree for them, the outcome is always "lit" */
   uint64_t shade_nodes;        /* shadeObject invocations */
   uint64_t samples;
   /* device time, ms (HIP events on the stream the kernels run on) */
   float ms_total;              /* the whole frame, bracketed on the caller's stream */
   float ms_trace;              /* the persistent traversal launches (one per chunk of tiles) */
   float ms_light, ms_combine, ms_resolve;   /* the kernels behind the traversal, timed one by one only for a scene made with
                                              * RT_SCENE_COUNTERS (the events between them cost a few microseconds each);
                                              * 0 otherwise: ms_total - ms_trace is what they took together */
   /* RT_SCENE_COUNTERS, single-chunk frames: when the waves of the traversal launch finished, measured from the
    * first wave's start (mean over waves, and the last one).  last >> mean = the launch is mostly its tail:
    * a few waves finishing long dependent bounce chains */
   float wave_exit_mean_ms, wave_exit_last_ms;
   int32_t launches_trace;      /* = chunks the frame was cut into */
   int32_t attempts;            /* > 1: the ray queue overflowed, was grown, and the frame was rendered again */
   /* traversal work counters (only filled when the scene was created with RT_SCENE_COUNTERS):
    * 64-byte node records fetched, primitive records tested, near-tie events -- for the closest-hit rays
    * (primary, reflection, refraction) and for the shadow feelers */
   uint64_t node_visits, prim_tests, tie_events;
   uint64_t node_visits_shadow, prim_tests_shadow, tie_events_shadow;
   /* wave-level step counts (one per wave per executed node step / leaf step): lane utilisation of
    * the traversal loop = (node_visits + node_visits_shadow) / (64 * wave_node_steps) */
   uint64_t wave_node_steps, wave_leaf_steps;
   /* where the other lanes of a node step were: waiting at a leaf for the wave's next leaf step, or without a ray
    * (waiting for the wave's next refill); and how many refills the waves made */
   uint64_t lane_steps_at_leaf, lane_steps_no_ray, wave_refills;
   /* shares of the waves' lifetime spent in refills (spawning bounce rays, starting new rays), in node steps and
    * in leaf steps */
   float wave_time_refill, wave_time_node, wave_time_leaf;
   int32_t second_tree;         /* which rays of this frame started in the second tree: bit 0 closest-hit rays, bit 1 shadow feelers */
   /* the longest single ray, in wave steps it was in flight */
   uint64_t max_ray_steps;
   /* most spawned rays / shadow items any wave held back at once (always filled); a frame that overflows the
    * lists is rendered again with longer ones, see attempts */
   uint64_t max_spawn_list;
   /* RT_SCENE_COUNTERS, scenes with the second tree: closest-hit rays whose nearest primitive had no rival within the
    * tie window and whose leaf the reference's walk provably reaches (answered without that walk); rays with a rival
    * (the reference's walk was made); rays for which the reach of the leaf could not be shown (likewise) */
   uint64_t hits_certified, hits_contested, hits_unreached;
   /* likewise: feelers reported "shadowed" from a primitive met in the second tree, clear of the target by more than
    * EQUAL_EPSILON, whose leaf the reference's walk provably reaches (the other shadowed feelers made that walk) */
   uint64_t shadows_certified;
   /* primary rays that were answered as misses without being traced, a 32x32-pixel tile at a time: every ray through
    * the tile passes the scene's bounding box by a margin (checked on the host, in double, per camera and frame
    * size; only for scenes without planes).  They are counted in rays_primary like the rest: the reference traces
    * each of them (kernel.cu:519-522) to the same result */
   uint64_t primaries_culled;
   /* shadow feelers that stopped after their own-surface test (kernel.cu:351) because isInShadow's answer cannot change
    * their point's colour: Phong frames (PhongShader.h:7-25), a light behind the surface whose mirror direction points
    * away from the eye.  Counted in rays_shadow like the rest (the reference walks its tree for each of them) */
   uint64_t shadows_unneeded;
   /* RT_SCENE_COUNTERS: of hits_contested, the rays whose walk of the reference's tree left out every subtree without one
    * of the few primitives that can change its outcome (the others made the walk in full) */
   uint64_t hits_guided;
   /* RT_SCENE_COUNTERS: of wave_node_steps, those in which the wave held rays in both trees, and those in which every
    * ray it moved was in the second tree (the rest: rays in the reference's tree only) */
   uint64_t wave_steps_both_trees, wave_steps_second_tree_only;
} rt_stats;

typedef struct rt_scene_info {
   int32_t n_bvh_prims, n_planes, n_lights, n_nodes, n_materials;
   int32_t tree_depth;       /* edges on the longest root->leaf path */
   int32_t root;             /* node index, or ~prim when the tree is a single leaf, or INT32_MIN when empty */
   uint64_t device_bytes;
   float ms_build_host, ms_upload;
   float ms_build_device;    /* RT_SCENE_GPU_BVH: the device kernels of the tree build (ms_build_host then covers the rest) */
   int32_t filter_depth;     /* levels of the miss filter's tree, 0 when the scene has none */
} rt_scene_info;

/* Flattened BVH node, 64 bytes: both children's boxes live in the parent so
 * one aligned fetch serves both slab tests.  child >= 0: internal node index;
 * child < 0: leaf, primitive id = ~child.  Topology and boxes are those of
 * formBVH (bvh.cpp:30-223) -- see rt_scene_create. */
typedef struct rt_bvh_node {
   float lmin[3], lmax[3];
   float rmin[3], rmax[3];
   int32_t left, right;
   int32_t pad[2];
} rt_bvh_node;

typedef struct rt_scene rt_scene;   /* opaque: host copy + device-resident flat scene */

/* scene creation flags */
enum {
   RT_SCENE_DEFAULT = 0,
   RT_SCENE_COUNTERS = 1,    /* traversal statistics in rt_stats (a slower variant of the kernel) */
   RT_SCENE_FAST_BVH = 2,    /* NOT the reference's tree: a surface-area-heuristic BVH over the same primitives.
                              * Far fewer box tests per ray, but near-ties between primitives (kernel.cu:108-122)
                              * are resolved by visiting order, so a few silhouette samples differ from the
                              * reference; never part of a parity claim */
   RT_SCENE_NO_FILTER = 8,   /* do not build the miss filter (a second, widened tree that only serves to prove that a ray
                              * meets no primitive, so that the reference's tree need not be walked for it); results are
                              * the same with and without it */
   RT_SCENE_FILTER_ALWAYS = 16,  /* closest-hit rays and shadow feelers always start in the second tree (default: the first
                              * frames of a size try out who should, and the quickest combination stays) */
   RT_SCENE_GPU_BVH = 4      /* NOT the reference's tree either: a linear BVH (Morton codes + radix tree) built on the
                              * device in milliseconds; same caveat as RT_SCENE_FAST_BVH, lower tree quality, for scenes
                              * that change every frame.  Ignored by rt_scene_create_host */
};

/* ---- entry points ------------------------------------------------------ */

/* Number of HIP devices visible, or 0.  Never fails. */
int rt_device_count(void);

/* Text of the last error on the calling thread ("" if none). */
const char *rt_last_error(void);

/* Build the device scene for `device`: objects in the reference's creation
 * order (initScene, kernel.cu:384-464), the reference's BVH topology and
 * boxes (formBVH, bvh.cpp:30-103, built here on the host), flattened into
 * 64-byte records and uploaded.  Replaces allocateGPUScene + formBVH
 * (kernel.cu:572-687, 732-733). */
int rt_scene_create(const rt_scene_desc *desc, int device, int flags, rt_scene **out);
/* A replica of `scene` on `device` (the same device is allowed): the host side -- flattened records, both trees -- is
 * shared with the original and built only once, everything on the device is made anew.  The replicas of the device-list
 * and frames-in-flight drivers are made this way.  A replica is an independent scene object: it renders the same bytes,
 * keeps its own camera, tuning state and statistics, and either may be destroyed first.  No counterpart in the reference,
 * which builds its one scene per call (kernel.cu:732-733). */
int rt_scene_clone(const rt_scene *scene, int device, rt_scene **out);
/* The host half of rt_scene_create alone (no device is touched): the returned
 * scene answers rt_scene_get_info / rt_scene_get_bvh and nothing else; every
 * render entry point rejects it with RT_ERR_NO_DEVICE. */
int rt_scene_create_host(const rt_scene_desc *desc, rt_scene **out);
void rt_scene_destroy(rt_scene *scene);
int rt_scene_get_info(const rt_scene *scene, rt_scene_info *info);

/* Host-side views for tests: the flattened node array (n_nodes records) and
 * the BVH leaf order (n_bvh_prims primitive ids, the sorted geomList of
 * bvh.cpp:49-72).  The pointers stay valid until rt_scene_destroy. */
int rt_scene_get_bvh(const rt_scene *scene, const rt_bvh_node **nodes, const int32_t **leaf_order);

/* Replace the camera of an existing scene (no rebuild). */
int rt_scene_set_camera(rt_scene *scene, const rt_camera *camera);

/* Whole frame into a caller-owned HOST buffer of width*height*4 bytes:
 * generate -> extend -> shadow -> shade (x depth) -> combine -> resolve -> D2H.
 * This is the body of launch_kernel after scene set-up (kernel.cu:735-760).
 * stats may be NULL. */
int rt_render(rt_scene *scene, const rt_render_params *params,
              uint8_t *out_rgba_host, rt_stats *stats);

/* Same, output left in DEVICE memory (`out_rgba_dev`, layout per
 * params->out_layout), work enqueued on `stream` (a hipStream_t, may be NULL
 * for the default stream).  Returns after the frame is complete: its statistics,
 * and the per-tile costs that order the next frame's tiles, come back with it.
 * A scene owns its working buffers (ray queue, spawn lists, tile order) and the rendering
 * calls change them -- which is why they take a non-const scene --, so calls on one scene
 * must not overlap; callers that render concurrently use one scene each.
 * Bytes needed: RT_OUT_FRAME width*height*4; RT_OUT_TILES rt_tile_count()*RT_TILE*RT_TILE*4. */
int rt_render_device(rt_scene *scene, const rt_render_params *params,
                     void *out_rgba_dev, void *stream, rt_stats *stats);

/* Number of tiles a (first, stride) selection covers for a width x height frame. */
int rt_tile_count(int width, int height, int tile_first, int tile_stride);

/* Which of the selected tiles a frame of these parameters would trace: live[k] = 1 for selection entry k
 * (k < rt_tile_count()), 0 for a tile that is left out of the launch because no primary ray through it can reach the
 * box that holds every primitive's bounding box (scenes without planes; see rt_stats.primaries_culled).  No
 * counterpart in the reference, which traces every sample (kernel.cu:519-522); a host computation, for callers and
 * tests that want to know what a frame skips.  Works on scenes made with rt_scene_create_host too. */
int rt_tiles_live(const rt_scene *scene, const rt_render_params *params, uint8_t *live, int n);

/* Scatter `n_parts` compact tile buffers (RT_OUT_TILES, part p made with
 * tile_first = p, tile_stride = n_parts), laid back to back in `tiles_dev`
 * with `part_stride_bytes` between parts, into a row-major device frame. */
int rt_untile_device(const void *tiles_dev, size_t part_stride_bytes, int n_parts,
                     int width, int height, void *frame_dev, void *stream);

/* ---- one frame on several devices --------------------------------------- */
/* The reference renders on one device (kernel.cu:689-761); its samples are independent (kernel.cu:484-543), so a
 * frame shards by tiles.  An rt_multi holds one replica of the scene per entry of `devices` (NULL: devices 0 ..
 * n_devices-1; a device may be listed more than once, its parts then share it).  rt_multi_render renders tile k of
 * the RT_TILE grid on part k mod n_devices, gathers the parts' compact tile buffers on the first device with direct
 * peer copies, un-tiles them there and copies the frame to the host: the same bytes rt_render produces.
 * params->tile_first / tile_stride / out_layout are ignored.  stats (may be NULL): counts summed over the parts,
 * times the maximum over the parts.  One frame at a time per rt_multi. */
typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_scene_desc *desc, const int *devices, int n_devices, int flags, rt_multi **out);
void rt_multi_destroy(rt_multi *multi);
int rt_multi_device_count(const rt_multi *multi);
int rt_multi_set_camera(rt_multi *multi, const rt_camera *camera);
int rt_multi_render(rt_multi *multi, const rt_render_params *params, uint8_t *out_rgba_host, rt_stats *stats);
/* the same, the frame left in memory of the first device (out_rgba_dev: width*height*4 bytes there) */
int rt_multi_render_device(rt_multi *multi, const rt_render_params *params, void *out_rgba_dev, rt_stats *stats);
/* Several frames in flight on the device list (no counterpart in the reference; see rt_sequence_* below for one device):
 * a driver made with rt_multi_create_pipelined keeps `frames_in_flight` (1 to 4) complete sets of replicas, streams and
 * host threads.  rt_multi_submit hands the next frame (params and camera copied; camera NULL = the camera of the frame
 * before it, or the scene's; out_rgba_dev: width*height*4 bytes on the first device, valid until the frame has been
 * waited for) to the next set and returns; with frames_in_flight frames submitted and not waited for it fails with
 * RT_ERR_BAD_ARGUMENT.  rt_multi_wait returns the oldest frame's status and statistics; frames complete in the order
 * submitted.  The host's work per frame (waking the part threads, waiting for them, un-tiling, synchronising) then
 * runs beside the next frame's launches.  rt_multi_render / _render_device / _set_camera need nothing in flight. */
int rt_multi_create_pipelined(const rt_scene_desc *desc, const int *devices, int n_devices, int flags, int frames_in_flight, rt_multi **out);
int rt_multi_frames_in_flight(const rt_multi *multi);
int rt_multi_submit(rt_multi *multi, const rt_render_params *params, const rt_camera *camera, void *out_rgba_dev);
int rt_multi_wait(rt_multi *multi, rt_stats *stats);

/* ---- a frame sequence with several frames in flight --------------------- */
/* The reference renders one frame per process (raytrace.cpp:18-56) and has nothing like this.  A caller that renders
 * a sequence of frames of one scene (a camera path) keeps `frames_in_flight` replicas of the scene on `device`, each
 * with a stream and a host thread of its own; frames are dealt to the replicas in turn, so that the end of one frame
 * (the last waves of its traversal launch, its resolve pass, the host's wake-up) overlaps the start of the next.
 * Every frame is a whole rt_render_device frame: the same bytes and statistics.
 *   rt_sequence_submit  queues the next frame and returns: `params` and `camera` (NULL = the camera of the frame submitted
 *                       before it, or the scene's if none has been given yet) are copied; `out_rgba_dev` (device memory,
 *                       layout per params->out_layout) must stay valid until the frame has been waited for.  With
 *                       frames_in_flight frames queued or running it fails with RT_ERR_BAD_ARGUMENT: wait for one first.
 *                       The first 6 x frames_in_flight frames of a frame size are rendered one at a time (every replica
 *                       times its first frames to choose how it uses the second tree and to order its tiles, and a
 *                       frame running beside them would spoil that): those submissions block until the frames before
 *                       them have finished.
 *   rt_sequence_wait    waits for the oldest frame not yet waited for and returns its status and statistics
 *                       (RT_ERR_BAD_ARGUMENT when no frame is in flight).  Frames complete in the order submitted.
 * Submit and wait are for one calling thread at a time. */
typedef struct rt_sequence rt_sequence;
int rt_sequence_create(const rt_scene_desc *desc, int device, int flags, int frames_in_flight, rt_sequence **out);
void rt_sequence_destroy(rt_sequence *seq);
int rt_sequence_frames_in_flight(const rt_sequence *seq);
int rt_sequence_submit(rt_sequence *seq, const rt_render_params *params, const rt_camera *camera, void *out_rgba_dev);
int rt_sequence_wait(rt_sequence *seq, rt_stats *stats);

/* Parity/debug views of the same kernels.
 * Per-sample closest hit of the primary rays (the first getClosestIntersection
 * of rayTrace, kernel.cu:531): prim_ids[i] (-1 = miss) and t[i] for sample
 * i = y*(width*s) + x; both host arrays of width*s*height*s entries. */
int rt_trace_primary(rt_scene *scene, const rt_render_params *params,
                     int32_t *prim_ids_host, float *t_host);

/* The same view of the frame the scene rendered last (rt_render / rt_render_device, whole launch): the closest hits its
 * primaries really got, as the frame's own launch stored them -- with the tile start order, the second tree and the
 * tiles left out of the launch all as that frame had them (a sample of a tile that was left out reads prim -1,
 * t FLT_MAX, which is what the frame took it for).  n = width*s*height*s of that frame.  Valid until the scene's next
 * call; RT_ERR_BAD_ARGUMENT if the last call was not a frame rendered in one launch.  Like rt_trace_primary this is
 * the first getClosestIntersection of rayTrace (kernel.cu:531) per sample. */
int rt_frame_primary_hits(const rt_scene *scene, int32_t *prim_ids_host, float *t_host, int64_t n);

/* Closest hit of arbitrary rays: rays_host = n x {ox,oy,oz,dx,dy,dz};
 * to_beat_host may be NULL (= -FLT_MAX for every ray, kernel.cu:84). */
int rt_trace_rays(rt_scene *scene, int64_t n, const float *rays_host,
                  const float *to_beat_host, int32_t *prim_ids_host, float *t_host);

/* rt_trace_rays for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc).
 * No counterpart in the reference.  Same arguments and results as rt_trace_rays -- host arrays in, host arrays out,
 * the call returns when the results are there -- but it makes no hipMalloc / hipFree: its working memory (counters, the
 * staged rays, the queue entries, the hits) is an array inside the library, RT_TRACE_RAYS_STATIC_MAX rays' worth
 * (36 MB), which every device the library runs on holds until the process ends.  More rays than that are rejected with
 * RT_ERR_TOO_LARGE.  All scenes on one device share that device's array: calls from several threads on one device take
 * turns (the library serialises them), calls on different devices run side by side.  It takes no stream and
 * synchronises, so it is not for use inside a HIP graph capture. */
#define RT_TRACE_RAYS_STATIC_MAX 524288
int rt_trace_rays_static(rt_scene *scene, int64_t n, const float *rays_host,
                         const float *to_beat_host, int32_t *prim_ids_host, float *t_host);

/* Per-sample clamped float RGB, the reference's antiAliasBuffer
 * (kernel.cu:537-541): n_samples x 3 floats, host. */
int rt_render_samples(rt_scene *scene, const rt_render_params *params,
                      float *rgb_host, rt_stats *stats);

/* ---- host-side pieces either side of the path -------------------------- */

/* POV-Ray-subset parser (POVRayParser::parseFile, POVRayParser.cpp:14-62).
 * The returned description owns its arrays; free with rt_scene_desc_free. */
int rt_parse_pov(const char *path, rt_scene_desc **out);
void rt_scene_desc_free(rt_scene_desc *desc);

/* readBMP (kernel.cu:35-61): a 24-bit uncompressed .bmp into tex->width/height/rgba.  rgba is malloc'ed and
 * overwrites the field without looking at its old value (so *tex need not be initialised, and a buffer it pointed to
 * stays the caller's); a description from rt_parse_pov frees it in rt_scene_desc_free, any other owner with free(). */
int rt_load_bmp(const char *path, rt_texture *tex);

/* 24-bit uncompressed TGA exactly as the reference's driver + Image::WriteTga
 * produce it from the uchar4 frame (raytrace.cpp:44-55, Image.cpp:38-101). */
int rt_write_tga(const char *path, int width, int height, const uint8_t *rgba);

#ifdef __cplusplus
}
#endif
#endif /* RAYTRACER_AMD_H */
