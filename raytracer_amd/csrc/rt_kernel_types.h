/*
 * rt_kernel_types.h -- what the kernels of the render path and the host code that launches them agree on: the device
 * views of a scene, the sample grid, the records of the ray queue, the arguments of the traversal launch and its
 * tuning constants.
 */
#ifndef RT_KERNEL_TYPES_H
#define RT_KERNEL_TYPES_H

#include <hip/hip_runtime.h>

#include <cfloat>

#include "rt_common.h"
#include "rt_device.h"

#define RT_BLOCK 256

/* ======================================================================= */
/* device-side views                                                        */
/* ======================================================================= */
struct DScene {
   const rt_bvh_node *nodes;
   const rt_dev_prim *prims;
   const rt_dev_xform *xforms;
   const rt_dev_shade *shades;
   const rt_dev_plane *planes;
   const rt_material *materials;
   const rt_dev_light *lights;
   const rt_dev_texinfo *tex_info;
   const uchar4 *texels;
   const int *prim_ids;     /* primitive id (the caller's numbering) of device primitive k; see rt_scene_create */
   const int *leaf_parent;  /* scenes with a miss filter: the node of the reference's tree whose child device primitive k is */
   int n_prims, n_planes, n_lights, root;
   int filter_root;  /* node index of the second tree's root, or RT_ROOT_EMPTY when the scene has none (or the launch uses none) */
   int filter_kinds; /* which rays of this launch start in the second tree: bit 0 closest-hit rays, bit 1 shadow feelers */
   int fast_ok;      /* every BVH bound is finite and below 2^40 in magnitude */
   int any_xform;    /* some BVH primitive carries a transform */
   /* for the margins a ray gives the second tree (make_ray_ctx): the middle of the box that holds the scene's spheres,
    * half its diagonal, and rt_flat_scene::sphere_coeff */
   float mid_x, mid_y, mid_z, mid_rad, sphere_coeff;
   float origin_reach;   /* a ray that starts further than this from the origin of the coordinates, on some axis, does not use the second tree (whose boxes are widened for the others, see the host's tree build) */
   int verify_jump;      /* every box of the reference's tree lies inside its parent's (checked when the scene is made): a ray
                          * on its way down to one leaf may go from the root straight to the leaf's parent, see the node step */
};

/* Division of a 32-bit unsigned by a divisor fixed for the launch, without the 30-odd instructions of the general
 * sequence (Granlund & Montgomery's round-up method): with l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1,
 * x / d = (mulhi(m, x) + ((x - mulhi(m, x)) >> min(l, 1))) >> max(l - 1, 0) for every x below 2^32 and d >= 1. */
struct FastDiv { unsigned m, s1, s2; };
static inline FastDiv make_fastdiv(unsigned d)
{
   unsigned l = 0;
   while ((1ULL << l) < d) l++;
   FastDiv f;
   f.m = (unsigned)((((1ULL << l) - d) << 32) / d + 1ULL);
   f.s1 = l < 1 ? l : 1; f.s2 = l > 0 ? l - 1 : 0;
   return f;
}
__device__ __forceinline__ unsigned fd_div(unsigned x, FastDiv f)
{
   unsigned t1 = __umulhi(f.m, x);
   return (t1 + ((x - t1) >> f.s1)) >> f.s2;
}

struct DCamera {
   F3 pos, up, right, look_dir;   /* look_dir = normalize(lookAt - pos), kernel.cu:717-718 */
};

/* geometry of the sample grid of one batch of image tiles */
struct DGrid {
   int width, height;       /* pixels */
   int sroot;               /* strata per pixel edge */
   int res_w, res_h;        /* samples: width*sroot, height*sroot */
   int tiles_x;             /* tiles per frame row */
   int tile_first, tile_stride;   /* tile id of selection entry k = tile_first + k*tile_stride */
   int k0;                  /* first selection entry of this batch */
   int ts;                  /* samples per tile edge = RT_TILE*sroot */
   int patches_per_row;     /* ts / 8 */
   unsigned per_tile;       /* samples of one tile = ts*ts */
   FastDiv per_tile_div, patches_per_row_div, tiles_x_div;
   float inv_res_w, inv_res_h;   /* RN(1/res_w), RN(1/res_h): for the exact divisions of the camera rays */
   int jitter; unsigned seed;
};

struct Hit { int prim; float t; };

/* queue entry, 32 bytes: origin, early-out threshold (toBeat, kernel.cu:84), direction, valid flag */
struct QRay { float ox, oy, oz, to_beat, dx, dy, dz; int valid; };

/* what a shade node hands to the combine step besides its light sum: the weights and the
 * children of kernel.cu:375-376 */
struct SpawnRec {
   float refl, alpha;
   int child_refl, child_refr;   /* index into the next level, -1 = no ray */
};

enum { ST_SHADOW_SKIPPED = 0, ST_REFLECT, ST_REFRACT, ST_SHADE_NODES,
       ST_NODE_VISITS, ST_PRIM_TESTS, ST_TIES, ST_WAVE_NODE_STEPS, ST_WAVE_LEAF_STEPS,          /* extend launches */
       ST_NODE_VISITS_SHADOW, ST_PRIM_TESTS_SHADOW, ST_TIES_SHADOW, ST_CERT_OK, ST_CERT_RIVAL, ST_CERT_UNREACHED, ST_GUIDED, ST_STEPS_BOTH, ST_STEPS_SECOND_ONLY, ST_SHADOW_CERT, ST_SHADOW_UNNEEDED, ST_LEAF_WAIT, ST_NO_RAY, ST_REFILLS, ST_CLK_REFILL, ST_CLK_NODE, ST_CLK_LEAF, ST_CLK_WAVE,
       ST_ABORTED, ST_MAX_RAY_STEPS, ST_MAX_LIST, ST_OVERFLOW, ST_LIST_OVERFLOW,
       ST_CLOCK_FIRST_START, ST_CLOCK_LAST_EXIT, ST_CLOCK_EXIT_SUM, ST_WAVES,
#ifdef RT_DEBUG_EXIT_HIST
       ST_HIST, ST_COUNT = ST_HIST + 48 };    /* when each wave of the traversal launch was done, in 0.25 ms bins */
#else
       ST_COUNT };
#endif

/* Statistics counters are replicated RT_STAT_REPL times (blocks spread over the replicas, the host
 * sums them) and bumped once per wave, so that counting does not serialise the kernels on one word. */
#define RT_STAT_REPL 32
/* the host's pinned copy of the counters, rounded up to whole kilobytes; behind it, in the same allocation, the
 * costs (two words) and the start order (one word) of up to RT_PINNED_TILES tiles (a 2560x1440 frame has 3 600) */
#define RT_STATS_BYTES_PADDED ((((size_t)ST_COUNT * RT_STAT_REPL * sizeof(unsigned long long)) + 1023) & ~(size_t)1023)
#define RT_PINNED_TILES 4096
__device__ __forceinline__ void stat_add(unsigned long long *stats, int which, unsigned long long v)
{
   atomicAdd(&stats[which * RT_STAT_REPL + (blockIdx.x & (RT_STAT_REPL - 1))], v);
}
/* counts the lanes of the wave for which pred holds; call from converged code */
__device__ __forceinline__ void stat_count(unsigned long long *stats, int which, bool pred)
{
   unsigned long long m = __ballot(pred);
   unsigned long long all = __ballot(1);
   if (m && __lane_id() == (unsigned)__ffsll((long long)all) - 1) stat_add(stats, which, (unsigned long long)__popcll(m));
}

struct StackEntry { int child; float dist; };

enum { OUT_HIT = 0, OUT_CHAIN = 2 };
enum { MODE_EXTEND = 0, MODE_SHADOW = 1, MODE_SETUP = 2, MODE_KIND = 3,
       MODE_FILTER = 4,     /* set while the ray is still in the second tree (see enter_tree below) */
       MODE_VERIFY = 8,     /* set while the ray walks the reference's tree from the root down to one leaf */
       /* a feeler in the second tree, about the primitives it has met in front of its target (see the leaf step) */
       MODE_BAND = 16,      /* one within EPSILON of the target: the reference's walk would stop at it and report "lit" */
       /* a feeler being set up (MODE_SETUP) whose answer cannot change its point's colour: see TraceArgs::skip_unlit */
       MODE_UNNEEDED = 32,
       /* a closest-hit ray on the reference's walk that knows which few leaves can change that walk's outcome and leaves
        * out every subtree that holds none of them (see finish() and the node step) */
       MODE_GUIDED = 64 };
/* The leaves such a ray keeps: at most RT_GUIDE_K, per lane in LDS (a count, the leaves, the largest parameter among
 * them).  They are all the primitives the ray met in the second tree; the walk is guided only if every one of them lies
 * within RT_GUIDE_NEAR of the nearest.  The second tree was searched RT_LOOK_ABS behind the nearest hit (less the
 * RT_SLACK_MAX a sphere's parameter may be off by), so every other primitive the ray meets is more than EQUAL_EPSILON
 * behind every kept one: it ties with none of them and is never preferred to one. */
#define RT_GUIDE_K 4
#define RT_GUIDE_WORDS (RT_GUIDE_K + 2)
#define RT_GUIDE_NEAR 0.03f
/* How far beyond the nearest hit so far a closest-hit ray still looks in the second tree: every box that starts
 * within RT_LOOK_ABS + t * RT_LOOK_REL behind it.  It has to know of every primitive within EQUAL_EPSILON of its
 * nearest one; the rest of the margin is allowance for a primitive's computed parameter lying in front of where its
 * (widened) box starts, which rounding in the primitive test could cause. */
/* How far beyond its target a feeler still looks in the second tree: boxes that start within toBeat * RT_FEELER_REL
 * behind it.  A primitive counts against a feeler only from EPSILON in front of the target, and stops the reference's
 * walk from the target's own parameter on; its box starts no later than its true parameter, so this margin (with the
 * EPSILON in the first case) is what the primitive tests' rounding may use up: 6e-5 of the light's distance, about a
 * thousand units in the last place.  (It was 2^-10 at first: behind a target 173 away from its light that is 0.17, five
 * triangles deep into the bunny, and a fifth of its feelers' fetches.) */
#ifndef RT_FEELER_REL
#define RT_FEELER_REL (1.0f / 16384.0f)
#endif
#ifndef RT_LOOK_ABS
#define RT_LOOK_ABS 0.05f
#endif
#ifndef RT_LOOK_REL
#define RT_LOOK_REL 0.0078125f
#endif
#define RT_SHADOW_ITEM 0x80000000u

#ifndef RT_STEP_CAP        /* (a test build may lower it, to turn a wave that would spin into an error) */
#define RT_STEP_CAP (1ULL << 28)
#endif
#define RT_REFILL_IDLE 24   /* pull new rays once this many lanes of the wave are idle */
#define RT_CUR_IDLE ((int)0x80000000)
#define RT_LIST_CAP_FIRST 1024u   /* entries of a wave's list to begin with; doubled when a frame overflows it */
#define RT_LIST_CAP_MAX 65536u
#define RT_NODE_BURST 12    /* most node steps between two leaf steps / refill checks */
#define RT_LEAF_TRIGGER 32  /* lanes waiting at a leaf that end a burst of node steps early */
#define RT_FETCH_STREAMS 64u /* counters the primaries are handed out through */
#define RT_CTR_WORDS (8 * ((int)RT_FETCH_STREAMS + 1))   /* 64-bit words of a launch's counters: the streams' (64 bytes apart), then the queue tail */
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 4  /* resident waves per SIMD the traversal kernel is compiled for (128 registers per lane) */
#endif
#define RT_OLD_RAY 150u      /* a ray that has cost more than this raises its wave's priority (k_trace) */
#define RT_LONG_FEELER 128u  /* a feeler of a primary that cost more than this raises its tile's cost (k_trace) */
#define RT_ORDER_CLASSES 1024 /* classes of tile cost that decide the order in which a frame's tiles are started */
#define RT_ORDER_MEAN_WEIGHT 0  /* what a tile's mean chain adds to its longest one in that cost (it was 4 while tiles nothing can be seen through still
                                 * filled the end of a launch; since they are left out, the longest chain alone orders better: dragon -6 %, balls2 -2.5 %) */
#define RT_LDS_STACK 16     /* stack entries per lane kept in LDS; deeper trees spill the rest to global memory */
#define RT_RESERVE 256      /* queue entries a wave reserves at a time for the bounce rays it spawns */

struct TraceArgs {
   QRay *rays;                  /* the queue; OUT_CHAIN appends spawned rays behind the first n */
   unsigned long long n;        /* rays to fetch (OUT_CHAIN: the primaries) */
   Hit *hits;
   unsigned *flag_words;        /* OUT_CHAIN: isInShadow of (queue entry i, light k) is bit k%32 of flag_words[i*flag_stride + k/32]; cleared when entry i is shaded */
   unsigned flag_stride;
   unsigned long long *next;    /* this launch's zeroed fetch counters: RT_FETCH_STREAMS of them, 64 bytes apart */
   unsigned long long *stats;
   int refill_idle, node_burst; /* tuning: idle lanes that trigger a refill, most node steps between two leaf steps */
   int leaf_trigger;            /* tuning: lanes waiting at a leaf that end a burst of node steps early */
   unsigned reserve;            /* OUT_CHAIN: queue entries a wave reserves per atomic for the bounce rays it spawns */
   /* OUT_CHAIN */
   DCamera cam;                 /* the first n entries are the primaries of grid `g`: made here, on fetch */
   DGrid g;
   SpawnRec *recs;
   float4 *points;              /* per queue entry that hit something: the hit point and the primitive, for its shadow feelers */
   unsigned *tail;              /* next queue entry no wave has reserved; starts at n */
   unsigned cap;                /* entries the queue has room for */
   int max_rec;
   uint2 *lists;                /* one ring of list_cap (a power of two) entries per wave of the launch: {queue entry | RT_SHADOW_ITEM, light} */
   unsigned list_cap;
   const int *tile_order;       /* when non-null: the order in which the launch's tiles are started (costliest first) */
   /* OUT_CHAIN, Phong frames: a feeler whose answer cannot change the colour of its point is not traced.  PhongShader.h:7-25 adds, for a light that is not shadowed, (dif * clamp(n.l)) * colour and
    * (spec * pow(clamp(r.e), 1/rough)) * colour to what a shadowed light gives; where both factors are exactly zero --
    * the light is behind the surface and its mirror direction points away from the eye -- the two cases are the same
    * floats, and isInShadow's answer is never looked at.  The spawn stage works the two factors out with the shader's
    * own expressions and sets the shadow bit of every light that cannot matter (whatever the bit says, the colour is the
    * same); the feeler of such a light finds its bit set, still makes its own-surface test (kernel.cu:351, which the
    * ray counts depend on) and stops there. */
   int skip_unlit;
   unsigned long long n_fetch;  /* OUT_CHAIN: the first n_fetch slots (in start order) are handed out; the tiles behind them are known to be
                                 * empty (no ray through them can meet anything) and nothing of them is made, traced or stored.  Otherwise = n */
   StackEntry *spill;           /* SPILL kernels: stack entries beyond the RT_LDS_STACK kept in LDS, [entry - RT_LDS_STACK][lane of the launch] */
   unsigned *dur;               /* when non-null: per queue entry, what the closest-hit ray cost: its own node steps + 4 x its leaf steps */
   unsigned *tile_bounce;       /* when non-null: per tile of the launch (zeroed), set when a primary of the tile spawns a bounce ray */
   unsigned *tile_hit;          /* when non-null: per tile of the launch (zeroed), set when a primary of the tile hits something */
   unsigned *tile_cost;         /* when non-null: two words per tile of the launch (zeroed); the first is raised to the cost of a long feeler of one of the tile's primaries */
   int guide;                   /* the reference's tree carries each node's range of leaves and its boxes are nested: contested closest-hit rays may walk it guided */
   float tree_lo[3], tree_hi[3];   /* the box that holds both boxes of that tree's root: a ray that starts outside it starts inside no box of the tree */
};

__device__ __forceinline__ QRay load_qray_rec(const QRay *q, unsigned long long i)
{
   const float4 *p = reinterpret_cast<const float4 *>(q + i);
   float4 a = p[0], b = p[1];
   QRay r;
   r.ox = a.x; r.oy = a.y; r.oz = a.z; r.to_beat = a.w; r.dx = b.x; r.dy = b.y; r.dz = b.z; r.valid = __float_as_int(b.w);
   return r;
}
/* the same entry read at agent scope, past the CU's vector cache: for entries this kernel wrote itself */
__device__ __forceinline__ QRay load_qray_written(QRay *q, unsigned long long i)
{
   unsigned *p = reinterpret_cast<unsigned *>(q + i);
   unsigned w[8];
#pragma unroll
   for (int k = 0; k < 8; k++) w[k] = __hip_atomic_load(p + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
   QRay r;
   r.ox = __uint_as_float(w[0]); r.oy = __uint_as_float(w[1]); r.oz = __uint_as_float(w[2]); r.to_beat = __uint_as_float(w[3]);
   r.dx = __uint_as_float(w[4]); r.dy = __uint_as_float(w[5]); r.dz = __uint_as_float(w[6]); r.valid = (int)w[7];
   return r;
}
__device__ __forceinline__ void store_qray(QRay *q, unsigned long long i, F3 o, F3 d, float to_beat, int valid)
{
   float4 *p = reinterpret_cast<float4 *>(q + i);
   p[0] = make_float4(o.x, o.y, o.z, to_beat);
   p[1] = make_float4(d.x, d.y, d.z, __int_as_float(valid));
}
__device__ __forceinline__ Ray load_qray(const QRay *q, long long i)
{
   QRay e = load_qray_rec(q, (unsigned long long)i);
   Ray r; r.o = mk3(e.ox, e.oy, e.oz); r.d = mk3(e.dx, e.dy, e.dz);
   return r;
}

struct TraceKernArgs { DScene sc; TraceArgs a; };

#endif
