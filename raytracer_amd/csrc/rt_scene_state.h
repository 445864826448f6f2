/*
 * rt_scene_state.h -- the scene object behind the C ABI (host side), and what the files that implement it share.
 *
 * rt_scene.hip builds and destroys it, rt_render.hip renders frames with it, rt_rays.hip traces loose rays with it.
 * A scene owns its working buffers (ray queue, spawn lists, counters, tile order): one call at a time per scene.
 */
#ifndef RT_SCENE_STATE_H
#define RT_SCENE_STATE_H

#include <hip/hip_runtime.h>

#include <memory>
#include <vector>

#include "rt_kernel_types.h"

#define HIP_TRY(expr)                                                                     \
   do {                                                                                   \
      hipError_t e_ = (expr);                                                             \
      if (e_ != hipSuccess) {                                                             \
         rt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
         return RT_ERR_HIP;                                                               \
      }                                                                                   \
   } while (0)

struct Arena {
   char *base = nullptr;
   size_t cap = 0, used = 0;
};

/* owner of a flattened scene's host arrays */
struct HostScene {
   rt_flat_scene flat;
   HostScene() { memset(&flat, 0, sizeof flat); }
   ~HostScene() { rt_flat_scene_free(&flat); }
   HostScene(const HostScene &) = delete;
   HostScene &operator=(const HostScene &) = delete;
};

struct rt_scene {
   rt_flat_scene host;      /* the flattened scene and its trees on the host; the arrays belong to host_owner */
   std::shared_ptr<HostScene> host_owner;   /* shared by the replicas of a scene (rt_scene_clone): built once, freed with the last of them */
   int device;
   int flags;
   DScene d;
   rt_camera camera;
   int has_camera;
   void *dev_blob;          /* one allocation holding every scene array */
   size_t dev_bytes;
   float ms_build_host, ms_upload, ms_build_device;
   Arena arena;             /* per-frame queues; grown on demand, reused across frames */
   unsigned long long *d_ctr;    /* work counters of the chunk in flight: RT_CTR_WORDS words (see ChunkCtr) */
   unsigned long long *d_stats;
   unsigned long long *h_stats;  /* pinned; room for RT_PINNED_TILES tiles' costs and order behind it */
   float node_factor;       /* queue entries per primary ray the chain kernel gets room for; grows when a frame overflows */
   StackEntry *d_spill;     /* trees deeper than RT_LDS_STACK: the stack entries beyond LDS, (depth - RT_LDS_STACK) x lanes of a full launch */
   void *d_frame; size_t frame_cap;   /* rt_render's device copy of the frame, kept from call to call */
   /* rt_render's way back to the caller's (pageable) frame, from a scene's second such call on: through a pinned buffer
    * of the scene's own, in parts, each part copied on to the caller while the next one crosses the bus */
   void *h_frame = nullptr; size_t h_frame_cap = 0; unsigned host_frames = 0;
   hipEvent_t copy_ev[4] = {nullptr, nullptr, nullptr, nullptr}; bool copy_ev_ok = false;
   uint2 *d_lists;          /* the waves' lists of spawned rays (k_trace): list_waves rings of list_cap entries */
   unsigned list_cap, list_waves;
   /* Tiles are started costliest first, by the previous frame's depth of the bounce trees per tile: the waves that
    * pick up the last primaries of a launch then hold short chains, and the launch does not end on a few long ones. */
   unsigned *d_tile_cost, *h_tile_cost;   /* h_: pinned */
   int *d_tile_order, *h_tile_order;      /* h_: pinned */
   long long order_cap;                   /* tiles the four arrays have room for */
   bool tiles_in_block = false;           /* h_tile_cost and h_tile_order lie behind h_stats in its allocation (frames of up to RT_PINNED_TILES tiles: one pinned
                                           * allocation per scene instead of three -- freeing one costs up to 0.2 ms) */
   long long order_key[6];                /* width, height, samples, tile_first, tile_stride, tiles of the frame the order is for */
   bool can_bounce;                       /* some material reflects or refracts: without one every ray tree is a single node */
   bool order_valid, order_dirty;         /* dirty: h_tile_order differs from what d_tile_order holds */
   /* The frame's work runs chunk after chunk on the caller's stream, or (RT_OWN_STREAM=1) on this one with the caller's
    * stream only bracketing it. */
   hipStream_t chain;
   hipEvent_t ev_begin, ev_end;
   bool streams_ok;
   std::vector<hipEvent_t> timing_events;   /* spare events of the per-frame timers */
   std::vector<int32_t> prim_ids_host;      /* host copy of DScene::prim_ids */
   std::vector<unsigned> last_cost;         /* the tile costs the current start order was made from */
   /* Tiles no ray can meet anything through (see cull_tiles): recomputed when the camera or the frame geometry changes */
   int cull_enabled;                        /* RT_CULL_TILES, default on */
   bool bounds_ok; double bounds_lo[3], bounds_hi[3];   /* the box every primitive's bounding box lies in */
   int cull_stage;          /* 1: the cached tile flags come from the quick test only; 2: from the finer one too */
   int cull_fine;           /* most boxes of the tree the tile test looks at per tile before it gives up and traces the tile (0: only the scene's box is tested) */
   bool guide_ok;           /* the reference's tree on the device carries each node's range of leaves (TraceArgs::guide) */
   float tree_lo[3], tree_hi[3];   /* the box that holds both boxes of that tree's root */
   std::vector<unsigned char> tile_live;    /* per selected tile: 1 = has to be traced */
   long long n_live;
   float cull_cam[12]; long long cull_key[6];
   /* launch shape and tuning of the traversal kernel, read once per scene from its device and the environment */
   int trace_blocks;        /* workgroups of a full traversal launch: what the device holds at 4 waves per SIMD */
   int refill_idle, node_burst, leaf_trigger, use_tile_order, order_classes;
   double order_mean_weight;
   unsigned reserve;        /* queue entries a wave reserves at a time for its bounce rays */
   int fuse_resolve;        /* tiles without bounce rays are shaded by k_resolve itself */
   int skip_unlit;          /* RT_SKIP_UNLIT: most lights a scene may have for TraceArgs::skip_unlit to be used (default 16; 0 = never) */
   bool lights_finite;      /* every light's colour is a finite number */
   int own_stream;          /* RT_OWN_STREAM=1: the frame's kernels run on a stream of the scene's own, bracketed by events on the caller's;
                             * default: on the caller's stream itself (the hop costs two event waits, ~0.03 ms a frame) */
   /* Which rays start in the second tree is tried out per scene and frame size: frames are the same bytes either way,
    * so the first frames of a size go through the four combinations (closest-hit rays x feelers) and the quickest stays.
    * Scenes whose rays mostly hit something they cannot be certified for (rays inside glass) do better without. */
   int filter_auto;         /* 1: try out; 0: always as filter_choice says (RT_MISS_FILTER=1) */
   int filter_choice;       /* bits as DScene::filter_kinds */
   int probe_step;          /* frames of the current size rendered so far, up to the end of the try-out */
   long long probe_key[4];
   float probe_ms[4];
   /* the one-launch frame rendered last, for rt_frame_primary_hits: where its primaries' hits lie in the arena (they
    * stay there until the scene's next call) */
   struct LastFrame { bool valid; DGrid g; long long np; long long hits_offset; bool culled; int n_sel; } last_frame;
};

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline unsigned long long sum_stat(const unsigned long long *h, int which)
{
   unsigned long long t = 0;
   for (int r = 0; r < RT_STAT_REPL; r++) t += h[which * RT_STAT_REPL + r];
   return t;
}

static inline unsigned grid_for(long long n) { return (unsigned)((n + RT_BLOCK - 1) / RT_BLOCK); }

/* shared between the files above */
int rt_env_int(const char *name, int fallback, bool allow_zero);
int rt_stack_depth(const rt_scene *s);                 /* entries the deepest walk of the scene's trees can defer */
int rt_trace_lds_bytes(const rt_scene *s, size_t *out);
int rt_scratch_address(int device, char **out);        /* with `device` current: that device's static working memory (rt_rays.hip) */
/* one persistent traversal launch over the queue `a` describes (rt_render.hip, where the kernel is instantiated) */
void rt_launch_trace(const rt_scene *s, int out_mode, bool count, TraceArgs a, unsigned long long n_hint, size_t lds_bytes,
                     hipStream_t stream, int filter_kinds = 3);

#endif
