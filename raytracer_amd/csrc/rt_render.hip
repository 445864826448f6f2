/*
 * rt_render.hip -- the per-pixel render path on MI355X (gfx950, wave64).
 *
 * Replaces the reference's rayTrace megakernel and its helpers (kernel.cu:67-382, 466-570).  The reference
 * gives every sample a thread that recurses to depth 6 (shadeObject<6> -> getReflection<5> -> ...); here a
 * frame is four kinds of launches:
 *
 *   k_trace         every ray of the frame in one persistent launch: primary rays, made from the sample slot
 *                   when a lane fetches it (generateCameraRays, kernel.cu:477-498); their closest hits
 *                   (getClosestIntersection, kernel.cu:83-200); the reflection / refraction rays those spawn
 *                   (getReflection / getRefraction, kernel.cu:209-276), to depth 6; and one shadow feeler per
 *                   light per hit (isInShadow, kernel.cu:202-207).  Follow-up work of a hit goes onto a list
 *                   private to the wave that found it, so waves never wait for each other.
 *   k_light         the light sum of every hit with the shadow flags (the loop of shadeObject, kernel.cu:347-358;
 *                   PhongShader.h, CookTorranceShader.h; getColor with textures, kernel.cu:323-332)
 *   k_combine_tree  colour = total*(1-refl-alpha) + refl*reflected + alpha*refracted (kernel.cu:375-376) over
 *                   each primary's bounce tree, children first: the reference's expression tree with the same
 *                   operands in the same order, no float atomics; also notes how deep each tile's trees were,
 *                   which orders the tiles of the next frame's k_trace (costliest first)
 *   k_resolve       s x s box filter to 8 bits (averageBufferColors, kernel.cu:545-570)
 *
 * Each ray walks the tree in the reference's visiting order (near child first with its `<=` rule, a deferred
 * sibling taken only if still closer than the hit so far), because the reference resolves near-ties between
 * primitives by visiting order.  The per-lane stack of deferred siblings lives in LDS, one column per lane.
 */
#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

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
   float origin_coeff;   /* RayCtx::wide as a fraction of the origin's largest coordinate (2^-20) */
   int verify_jump;      /* every box of the reference's tree lies inside its parent's (checked when the scene is made): a ray
                          * on its way down to one leaf may go from the root straight to the leaf's parent, see the node step */
};

/* Division of a 32-bit unsigned by a divisor fixed for the launch, without the 30-odd instructions of the general
 * sequence (Granlund & Montgomery's round-up method): with l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1,
 * x / d = (mulhi(m, x) + ((x - mulhi(m, x)) >> min(l, 1))) >> max(l - 1, 0) for every x below 2^32 and d >= 1. */
struct FastDiv { unsigned m, s1, s2; };
static FastDiv make_fastdiv(unsigned d)
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
       ST_NODE_VISITS_SHADOW, ST_PRIM_TESTS_SHADOW, ST_TIES_SHADOW, ST_CERT_OK, ST_CERT_RIVAL, ST_CERT_UNREACHED, ST_SHADOW_CERT, ST_SHADOW_UNNEEDED, ST_LEAF_WAIT, ST_NO_RAY, ST_REFILLS, ST_CLK_REFILL, ST_CLK_NODE, ST_CLK_LEAF, ST_CLK_WAVE,
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

/* ======================================================================= */
/* record loads                                                             */
/* ======================================================================= */
__device__ __forceinline__ void load_node(const rt_bvh_node *nodes, int idx, float4 &a, float4 &b, float4 &c, float4 &d)
{
   const float4 *p = reinterpret_cast<const float4 *>(nodes + idx);
   a = p[0]; b = p[1]; c = p[2]; d = p[3];
}

struct PrimRec { float4 a, b, c; uint4 m; };
__device__ __forceinline__ PrimRec load_prim(const rt_dev_prim *prims, int idx)
{
   const float4 *p = reinterpret_cast<const float4 *>(prims + idx);
   PrimRec r;
   r.a = p[0]; r.b = p[1]; r.c = p[2];
   r.m = *reinterpret_cast<const uint4 *>(p + 3);
   return r;
}
__device__ __forceinline__ rt_dev_xform load_xform(const rt_dev_xform *x, int idx)
{
   const float4 *p = reinterpret_cast<const float4 *>(x + idx);
   float4 a = p[0], b = p[1], c = p[2];
   rt_dev_xform m;
   m.r[0][0] = a.x; m.r[0][1] = a.y; m.r[0][2] = a.z; m.r[0][3] = a.w;
   m.r[1][0] = b.x; m.r[1][1] = b.y; m.r[1][2] = b.z; m.r[1][3] = b.w;
   m.r[2][0] = c.x; m.r[2][1] = c.y; m.r[2][2] = c.z; m.r[2][3] = c.w;
   return m;
}

/* Geometry::getIntersection (Geometry.h:16-18) for a BVH primitive */
__device__ __forceinline__ float prim_intersect(const DScene &sc, int prim, const Ray &wray)
{
   PrimRec p = load_prim(sc.prims, prim);
   Ray r = wray;
   if (sc.any_xform) {
      /* the transform is fetched beside the record, not after the record's flag has arrived: two dependent
       * fetches would double the latency of every leaf step of a scene of transformed spheres */
      rt_dev_xform x = load_xform(sc.xforms, prim);
      asm volatile("" : "+v"(x.r[0][0]), "+v"(x.r[0][1]), "+v"(x.r[0][2]), "+v"(x.r[0][3]), "+v"(x.r[1][0]), "+v"(x.r[1][1]),
                        "+v"(x.r[1][2]), "+v"(x.r[1][3]), "+v"(x.r[2][0]), "+v"(x.r[2][1]), "+v"(x.r[2][2]), "+v"(x.r[2][3]));
      if (p.m.y & RT_FLAG_XFORM) r = ray_transform(wray, x);
   }
   if (p.m.x == RT_PRIM_SPHERE) {
      return sphere_hit(r, mk3(p.a.x, p.a.y, p.a.z), p.a.w);
   } else if (p.m.x == RT_PRIM_BOX) {
      return slab_literal(r, p.a.x, p.a.y, p.a.z, p.a.w, p.b.x, p.b.y);
   } else {
      return triangle_hit(r, mk3(p.a.x, p.a.y, p.a.z), mk3(p.a.w, p.b.x, p.b.y), mk3(p.b.z, p.b.w, p.c.x),
                          mk3(p.c.y, p.c.z, p.c.w));
   }
}

__device__ __forceinline__ float plane_intersect(const DScene &sc, int k, const Ray &wray)
{
   const rt_dev_plane &pl = sc.planes[k];
   Ray r = wray;
   if (pl.flags & RT_FLAG_XFORM) r = ray_transform(wray, pl.inv);
   return plane_hit(r, mk3(pl.n[0], pl.n[1], pl.n[2]), pl.d);
}

/* id >= n_prims addresses a plane */
__device__ __forceinline__ float any_intersect(const DScene &sc, int id, const Ray &r)
{
   return id < sc.n_prims ? prim_intersect(sc, id, r) : plane_intersect(sc, id - sc.n_prims, r);
}

/* getNormalAt of a BVH primitive: Sphere.h:19-24, Triangle.h:24-26, SmoothTriangle.h:31-38, Box.h:42-60 */
__device__ __forceinline__ F3 prim_normal_at(const rt_dev_prim *prims, const rt_dev_xform *xforms, const rt_dev_shade *shades,
                                             int id, const Ray &ray, float t)
{
   PrimRec p = load_prim(prims, id);
   const float4 *sp = reinterpret_cast<const float4 *>(shades + id);
   float4 s0 = sp[0];
   if (p.m.x == RT_PRIM_SPHERE) {
      F3 pos = ray_point(ray, t);
      return normalize3(pos - mk3(s0.x, s0.y, s0.z));
   } else if (p.m.x == RT_PRIM_TRIANGLE) {
      return mk3(s0.x, s0.y, s0.z);
   } else if (p.m.x == RT_PRIM_SMOOTH) {
      float4 s1 = sp[1], s2 = sp[2];
      F3 p1 = mk3(p.a.x, p.a.y, p.a.z), p2 = mk3(p.a.w, p.b.x, p.b.y), p3 = mk3(p.b.z, p.b.w, p.c.x);
      F3 n = mk3(p.c.y, p.c.z, p.c.w);
      F3 q = ray_point(ray, t);
      float area = dot3(cross3(p2 - p1, p3 - p1), n);
      float beta = dot3(cross3(p1 - p3, q - p3), n) / area;
      float gamma = dot3(cross3(p2 - p1, q - p1), n) / area;
      float alpha = (float)(1.0 - (double)beta - (double)gamma);   /* the reference's "1.0" is a double */
      F3 n1 = mk3(s0.x, s0.y, s0.z), n2 = mk3(s0.w, s1.x, s1.y), n3 = mk3(s1.z, s1.w, s2.x);
      return normalize3(alpha * n1 + beta * n2 + gamma * n3);
   } else {
      Ray r = ray_transform(ray, load_xform(xforms, id));   /* Box.h:43 transforms unconditionally */
      F3 ip = ray_point(r, t);
      F3 mn = mk3(p.a.x, p.a.y, p.a.z), mx = mk3(p.a.w, p.b.x, p.b.y);
      F3 nm;
      if (is_equal(ip.x, mn.x)) nm = mk3(-1.0f, 0.0f, 0.0f);
      else if (is_equal(ip.x, mx.x)) nm = mk3(1.0f, 0.0f, 0.0f);
      else if (is_equal(ip.y, mn.y)) nm = mk3(0.0f, -1.0f, 0.0f);
      else if (is_equal(ip.y, mx.y)) nm = mk3(0.0f, 1.0f, 0.0f);
      else if (is_equal(ip.z, mn.z)) nm = mk3(0.0f, 0.0f, -1.0f);
      else nm = mk3(0.0f, 0.0f, 1.0f);
      float4 s1 = sp[1], s2 = sp[2];
      /* trans * (nm, 0): s = columns 0..2 of trans (rows 0..2); the 4th term is m3*0 */
      float c0x = s0.x, c0y = s0.y, c0z = s0.z, c1x = s0.w, c1y = s1.x, c1z = s1.y, c2x = s1.z, c2y = s1.w, c2z = s2.x;
      return mk3(c0x * nm.x + c1x * nm.y + c2x * nm.z + 0.0f,
                 c0y * nm.x + c1y * nm.y + c2y * nm.z + 0.0f,
                 c0z * nm.x + c1z * nm.y + c2z * nm.z + 0.0f);
   }
}

/* getNormalAt of any primitive id (planes follow the BVH primitives): Plane.h:20-22 */
__device__ __forceinline__ F3 normal_at(const DScene &sc, int id, const Ray &ray, float t)
{
   if (id >= sc.n_prims) {
      const rt_dev_plane &pl = sc.planes[id - sc.n_prims];
      return mk3(pl.wsN[0], pl.wsN[1], pl.wsN[2]);
   }
   return prim_normal_at(sc.prims, sc.xforms, sc.shades, id, ray, t);
}

/* The overlap rule of kernel.cu:108-122: of two hits closer than EQUAL_EPSILON the one whose normal
 * faces the eye more wins.  Rare (a few rays in a thousand): the traversal loop resolves it in a phase of its
 * own behind the leaf step, so that its registers do not add to the leaf step's. */
__device__ __forceinline__ bool tie_prefers_new(const rt_dev_prim *prims, const rt_dev_xform *xforms, const rt_dev_shade *shades,
                                             float ox, float oy, float oz, float dx, float dy, float dz,
                                             int old_prim, float old_t, int new_prim, float new_t)
{
   Ray ray; ray.o = mk3(ox, oy, oz); ray.d = mk3(dx, dy, dz);
   F3 oldNorm = prim_normal_at(prims, xforms, shades, old_prim, ray, old_t);
   F3 newNorm = prim_normal_at(prims, xforms, shades, new_prim, ray, new_t);
   F3 eye = normalize3(-ray.d);
   float newDot = dot3(eye, newNorm);
   float oldDot = dot3(eye, oldNorm);
   return newDot > oldDot;
}

/* ======================================================================= */
/* traversal: getClosestIntersection, kernel.cu:83-200                      */
/* ======================================================================= */
/* The per-ray constants of the box test, with the margins this ray gives the second tree.
 *
 * The reference's walk tests a primitive only after the ray has passed the box test of that primitive's own leaf box,
 * and the second tree's leaf box for the primitive contains that box; so the second tree reaches every primitive the
 * reference could test as long as its box test never turns away a box the reference's would let the ray into.  The
 * two tests round differently (their boxes differ), by a few units in the last place of the differences they take
 * from the ray's origin -- 1e-4 per subtraction for a light a thousand units from a scene of size one.  Beside the
 * widening its boxes carry for the size of their own coordinates, the second tree's boxes are therefore widened, for
 * this ray, by 2^-20 of the origin's largest coordinate (RayCtx::wide; eight times the box test's rounding).
 *
 * What remains is where along the ray a primitive reports its hit.  The second tree does not enter boxes that start
 * behind the stretch a ray cares about, on the grounds that a primitive is met no earlier than its box.  Triangles
 * and boxes keep to that within the margins used there (RT_LOOK_*, RT_FEELER_REL).  Sphere::intersects (Sphere.h:42-66)
 * does not: it takes the difference of two products of the size (distance)^2 and reports its hit off the sphere by
 * up to about 2^-22 distance^2 / radius (measured: 2^-22.6 at worst in 40 000 rays of balls2 from one, ten and a
 * hundred times its lights' distances).  `slack` is that, for this ray, as a parameter along it; it is added where
 * the second tree cuts a feeler's stretch short, it rides on RT_LOOK_ABS for the other rays, and a ray for which it
 * exceeds RT_SLACK_MAX does not use the second tree at all. */
__device__ __forceinline__ RayCtx make_ray_ctx(const DScene &sc, const Ray &r, float &slack)
{
   RayCtx c = make_ctx(r);
   c.wide = sc.origin_coeff * fmaxf(fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fabsf(r.o.z));
   slack = 0.0f;
   if (sc.sphere_coeff > 0.0f) {
      const F3 e = r.o - mk3(sc.mid_x, sc.mid_y, sc.mid_z);
      const float far = sqrtf(dot3(e, e)) + sc.mid_rad;
      slack = (2.384185791015625e-7f * sc.sphere_coeff) * (far * far) / sqrtf(dot3(r.d, r.d));
      if (!(slack <= RT_SLACK_MAX)) c.par |= RT_PAR_NO_FILTER;
   }
   if (!sc.fast_ok) c.fast = false;
   return c;
}

struct StackEntry { int child; float dist; };
extern __shared__ StackEntry lds_stack[];

/* ---- primary rays ------------------------------------------------------ */
__device__ __forceinline__ unsigned hash32(unsigned x)
{
   x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
   return x;
}

/* sample slot -> sample coordinates on the (W*s) x (H*s) grid.  A tile's samples are
 * numbered in 8x8 patches so that 64 consecutive slots cover a compact square of the image. */
__device__ __forceinline__ bool slot_to_sample(const DGrid &g, unsigned slot, int &X, int &Y)
{
   unsigned k = fd_div(slot, g.per_tile_div);
   unsigned r = slot - k * g.per_tile;
   unsigned patch = r >> 6, lane = r & 63;
   unsigned py = fd_div(patch, g.patches_per_row_div), px = patch - py * (unsigned)g.patches_per_row;
   int sx = (int)(px * 8 + (lane & 7)), sy = (int)(py * 8 + (lane >> 3));
   unsigned tile = (unsigned)g.tile_first + ((unsigned)g.k0 + k) * (unsigned)g.tile_stride;
   unsigned ty = fd_div(tile, g.tiles_x_div), tx = tile - ty * (unsigned)g.tiles_x;
   X = (int)tx * g.ts + sx; Y = (int)ty * g.ts + sy;
   return X < g.res_w && Y < g.res_h;
}

/* generateCameraRays, kernel.cu:477-498 */
__device__ __forceinline__ Ray camera_ray(const DCamera &cam, const DGrid &g, int x, int y)
{
   float r1 = 0.5f, r2 = 0.5f;
   if (g.jitter == RT_JITTER_HASH) {
      unsigned index = (unsigned)(y * g.res_w + x);
      unsigned base = hash32(index ^ (g.seed * 0x9e3779b9U));
      unsigned h1 = hash32(base ^ 0x68bc21ebU), h2 = hash32(base ^ 0x02e5be93U);
      r1 = (float)((h1 >> 8) + 1U) * 5.9604644775390625e-8f;   /* (0, 1], like curand_uniform */
      r2 = (float)((h2 >> 8) + 1U) * 5.9604644775390625e-8f;
   }
   /* the four divisions by the resolution, correctly rounded (rt_div) with the reciprocals the host passes */
   const float fw = (float)g.res_w, fh = (float)g.res_h;
   float uJitter = rt_div(r1 - .5f, fw, g.inv_res_w);
   float vJitter = rt_div(r2 - .5f, fh, g.inv_res_h);
   float u = 2.0f * rt_div((float)x, fw, g.inv_res_w) - 1.0f + uJitter;
   float v = 2.0f * rt_div((float)y, fh, g.inv_res_h) - 1.0f + vJitter;
   F3 rPos = (u * .5f) * cam.right + (v * .5f) * cam.up + cam.pos;
   F3 rDir = rPos - cam.pos + cam.look_dir;
   Ray r; r.o = rPos; r.d = rDir;
   return r;
}


/* k_trace: persistent-thread closest-hit / any-hit over a ray queue.
 *
 * Lane state is one ray: (ctx, t, closest, cur, sp).  cur >= 0 is an internal node, cur < 0 a leaf
 * (~cur = primitive).  The wave alternates between bursts of
 *   node steps  -- every lane at an internal node tests the two child boxes of its node
 *                  (kernel.cu:136-158) and descends / defers / pops,
 * and one
 *   leaf step   -- every lane at a leaf intersects its primitive (kernel.cu:104-134) and pops,
 * and refills idle lanes from the queue when RT_REFILL_IDLE of them are free.  A ray's own sequence
 * of operations is exactly that of the reference loop; only the interleaving between rays differs.
 *
 * Three variants of the node step, chosen per wave (uniform branch): the short box test for waves
 * whose rays are all ordinary, the short test with parallel-axis handling when some lane holds a ray
 * with an axis-parallel direction component (typically from the one pixel column or row through the
 * view axis), and the reference's literal arithmetic when some lane holds a ray outside the exact
 * division's range (RayCtx::fast false; no ordinary scene has one).
 *
 * OUT_HIT    a plain queue of rays: writes the closest hit of ray i to hits[i] (toBeat early-out honoured).
 * OUT_CHAIN  the whole shade forest of the first n rays (the primaries) in one launch.  When a ray has found its
 *            hit, its follow-up work is started at the wave's next refill:
 *              - its reflection and refraction rays (getReflection / getRefraction, kernel.cu:209-276, guarded as in
 *                kernel.cu:361-371), appended to the same queue behind the primaries (entries reserved a range at a
 *                time; valid = depth level + 1).  The lane goes on with one of them itself, at once; a second one
 *                goes to the front of a per-wave list (a ring in global memory);
 *              - one shadow item (isInShadow, kernel.cu:202-207, 347-352), to the back of that list: a lane that takes
 *                it traces the feeler of the node's next light and, while lights remain, puts the item back for
 *                another lane.  The feeler's own-surface test (kernel.cu:351) is an ordinary leaf step: the lane
 *                starts at the leaf of the shaded primitive in MODE_SETUP and moves to the root with the result as
 *                its toBeat.
 *            Idle lanes drain the list, front first, before they fetch new primaries.  Bounce rays never queue behind
 *            shadow items: they are the launch's longest dependent chains.  The list grows with the breadth of the
 *            forest, which is why it lives in global memory and a frame that overflows it is rendered again with
 *            longer lists.  Primaries are fetched tile by tile in the order the host passes (tiles whose chains of
 *            dependent rays took longest in the previous frame first), so that the last primaries of a launch start
 *            short chains.  No data is handed between waves, so there is nothing to wait for and every wave's exit is
 *            its own: primaries dry, list empty, lanes idle. */
enum { OUT_HIT = 0, OUT_CHAIN = 2 };
enum { MODE_EXTEND = 0, MODE_SHADOW = 1, MODE_SETUP = 2, MODE_KIND = 3,
       MODE_FILTER = 4,     /* set while the ray is still in the second tree (see enter_tree below) */
       MODE_VERIFY = 8,     /* set while the ray walks the reference's tree from the root down to one leaf */
       /* a feeler in the second tree, about the primitives it has met in front of its target (see the leaf step) */
       MODE_BAND = 16,      /* one within EPSILON of the target: the reference's walk would stop at it and report "lit" */
       /* a feeler being set up (MODE_SETUP) whose answer cannot change its point's colour: see TraceArgs::skip_unlit */
       MODE_UNNEEDED = 32 };
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
#define RT_LONG_FEELER 128u  /* a feeler of a primary that cost more than this raises its tile's cost (k_trace) */
#define RT_ORDER_CLASSES 1024 /* classes of tile cost that decide the order in which a frame's tiles are started */
#define RT_ORDER_MEAN_WEIGHT 0  /* what a tile's mean chain adds to its longest one in that cost (it was 4 while tiles nothing can be seen through still
                                 * filled the end of a launch; since they are left out, the longest chain alone orders better: dragon -6 %, balls2 -2.5 %) */
#define RT_LDS_STACK 16     /* stack entries per lane kept in LDS; deeper trees spill the rest to global memory */
#define RT_RESERVE 256      /* queue entries a wave reserves at a time for the bounce rays it spawns */

#ifdef RT_DEBUG_WAVES
#define RT_DBG_RAY_START dbg_ray_start = wall_clock64()
__device__ unsigned long long dbg_waves[4096 * 16];
extern "C" int rt_debug_read_waves(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(dbg_waves), sizeof(unsigned long long) * 4096 * 16) == hipSuccess ? 0 : 1; }
#else
#define RT_DBG_RAY_START (void)0
#endif
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

/* planes after the tree: kernel.cu:189-196 (no tie rule, no toBeat exit) */
__device__ __forceinline__ void plane_scan(const DScene &sc, const Ray &ray, float &t, int &closest)
{
   for (int k = 0; k < sc.n_planes; k++) {
      float dist = plane_intersect(sc, k, ray);
      if (less_than(dist, t) && above_zero(dist)) { closest = sc.n_prims + k; t = dist; }
   }
}

/* The reflection and refraction rays of a hit (getReflection kernel.cu:214-216, getRefraction
 * kernel.cu:239-257). */
struct BounceRays { float ro[3], rd[3], to[3], td[3]; int want_refl, want_refr; };
__device__ __forceinline__ BounceRays bounce_rays(const rt_dev_prim *prims, const rt_dev_xform *xforms, const rt_dev_shade *shades,
                                               const rt_dev_plane *planes, int n_prims,
                                               float ox, float oy, float oz, float dx, float dy, float dz, int prim, float t,
                                               float m_refl, float m_refr, float m_ior)
{
   BounceRays b;
   b.want_refl = b.want_refr = 0;
   for (int k = 0; k < 3; k++) b.ro[k] = b.rd[k] = b.to[k] = b.td[k] = 0.0f;
   Ray ray; ray.o = mk3(ox, oy, oz); ray.d = mk3(dx, dy, dz);
   F3 P = ray_point(ray, t);                                           /* kernel.cu:340 */
   F3 normal;
   if (prim >= n_prims) { const rt_dev_plane &pl = planes[prim - n_prims]; normal = mk3(pl.wsN[0], pl.wsN[1], pl.wsN[2]); }
   else normal = prim_normal_at(prims, xforms, shades, prim, ray, t);   /* kernel.cu:342 */
   F3 eye = normalize3(-ray.d);                                        /* kernel.cu:344 */
   if (m_refl > 0.0f) {
      F3 d = (2.0f * dot3(normal, eye)) * normal - eye;
      F3 o = P + RT_BIG_EPS * d;
      b.ro[0] = o.x; b.ro[1] = o.y; b.ro[2] = o.z; b.rd[0] = d.x; b.rd[1] = d.y; b.rd[2] = d.z;
      b.want_refl = 1;
   }
   if (m_refr > 0.0f) {
      float n1, n2; F3 refrNorm; F3 dd = -eye;
      if (less_than(dot3(eye, normal), 0.0f)) { n1 = m_ior; n2 = RT_AIR_IOR; refrNorm = -normal; }
      else { n1 = RT_AIR_IOR; n2 = m_ior; refrNorm = normal; }
      float dDotN = dot3(dd, refrNorm);
      float nr = n1 / n2;
      float disc = 1.0f - nr * nr * (1.0f - dDotN * dDotN);
      if (disc > 0.0f) {
         F3 d = nr * (dd - refrNorm * dDotN) - refrNorm * sqrtf(disc);
         F3 o = P + RT_BIG_EPS * d;
         b.to[0] = o.x; b.to[1] = o.y; b.to[2] = o.z; b.td[0] = d.x; b.td[1] = d.y; b.td[2] = d.z;
         b.want_refr = 1;
      }
   }
   return b;
}

struct TraceKernArgs { DScene sc; TraceArgs a; };

/* TWO: the launch uses the scene's second tree (sc.filter_root is one); without it none of the code that proves answers
 * from that tree is there */
template <int OUT, bool COUNT, bool SPILL, bool TWO>
__global__ __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_WAVES_PER_SIMD, RT_WAVES_PER_SIMD))) void k_trace(TraceKernArgs ka)
{
   /* The scene's arrays are used in every step and stay in scalar registers.  The launch's other arguments are mostly
    * wanted once per refill, spawn or finished ray: each use reads its field from the kernel-argument segment afresh
    * (a scalar load, served by the constant cache), so that a hundred values do not sit in scalar registers -- and,
    * spilled from there, in vector registers -- for the whole launch. */
   const DScene &sc = ka.sc;
   auto A = [&]() {
      unsigned zero = 0;
      asm volatile("" : "+s"(zero));      /* (keeps the compiler from hoisting these loads to the top and holding the values) */
      return (const TraceArgs *)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TraceKernArgs, a) + zero);
   };
   /* The per-lane stack of deferred siblings: one column per lane in LDS.  A tree deeper than RT_LDS_STACK levels
    * (SPILL) keeps the entries beyond that in global memory, a column per lane of the launch, so that deep trees do
    * not cost the four workgroups per CU their LDS; a ray rarely defers that many siblings at once. */
   StackEntry *stack = lds_stack + threadIdx.x;
   const int stride = RT_BLOCK;
   StackEntry *const spill = SPILL ? A()->spill + ((size_t)blockIdx.x * RT_BLOCK + threadIdx.x) : nullptr;
   const size_t spill_stride = SPILL ? (size_t)gridDim.x * RT_BLOCK : 0;
   /* (the empty asm statements keep the two loads apart: merged into one load through a generic pointer they cost more,
    * and the compiler's check of which address space that pointer is in does not assemble for gfx950) */
   auto stack_get = [&](int i) {
      StackEntry e;
      if (SPILL && i >= RT_LDS_STACK) { e = spill[(size_t)(i - RT_LDS_STACK) * spill_stride]; asm volatile("" : "+v"(e.child), "+v"(e.dist)); }
      else { e = stack[i * stride]; asm volatile("" : "+v"(e.child), "+v"(e.dist)); }
      return e;
   };
   auto stack_put = [&](int i, StackEntry e) {
      if (SPILL && i >= RT_LDS_STACK) { asm volatile("" : "+v"(e.child), "+v"(e.dist)); spill[(size_t)(i - RT_LDS_STACK) * spill_stride] = e; }
      else stack[i * stride] = e;
   };
   const unsigned lane = __lane_id();
   const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;
   /* OUT_CHAIN: this wave's list of spawned rays and shadow items still to be started, a ring of A()->list_cap entries in
    * global memory: entries head .. head+count-1 (wave-uniform).  Written with plain stores, read back past the
    * CU's vector cache (the reader is another lane of the same wave, some refills later). */
   uint2 *const list = A()->lists ? A()->lists + (size_t)(blockIdx.x * (RT_BLOCK / 64) + (threadIdx.x >> 6)) * A()->list_cap : nullptr;
   const int list_mask = (int)A()->list_cap - 1;
   int list_peak = 0;
   int list_count = 0, list_head = 0;
   auto list_put = [&](int k, uint2 v) { *reinterpret_cast<unsigned long long *>(list + ((list_head + k) & list_mask)) = (unsigned long long)v.x | ((unsigned long long)v.y << 32); };
   auto list_get = [&](int k) {
      unsigned long long w = __hip_atomic_load(reinterpret_cast<unsigned long long *>(list + ((list_head + k) & list_mask)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return make_uint2((unsigned)w, (unsigned)(w >> 32));
   };
   const unsigned long long n = A()->n_fetch;       /* slots that are handed out */
   unsigned long long *const stats = A()->stats;
   /* (what every turn of the loop or every finished ray wants stays in registers) */
   const int refill_idle = A()->refill_idle, node_burst = A()->node_burst, leaf_trigger = A()->leaf_trigger;
   Hit *const hits = A()->hits;
   const unsigned long long clock_start = COUNT ? wall_clock64() : 0ULL;     /* 100 MHz */
#ifdef RT_DEBUG_EXIT_HIST
   const unsigned long long dbg_start = wall_clock64();
#endif
#ifdef RT_DEBUG_WAVES
   const unsigned long long dbg_t0 = wall_clock64();
   unsigned long long dbg_last_fetch = 0, dbg_last_spawn = 0, dbg_list_at_dry = ~0ULL; unsigned dbg_fetched = 0, dbg_refills = 0;
   /* per lane: when its current ray started; of the last ray it finished: when, when that one started, and mode | tag << 8 | born << 16 */
   unsigned long long dbg_ray_start = 0, dbg_fin_clock = 0, dbg_fin_start = 0, dbg_fin_info = 0;
#endif

   unsigned long long guard = 0;       /* wave-uniform: turns of the loop below */
   unsigned born = 0;                  /* node and leaf steps this lane's ray has taken: what it cost, whatever else the wave was doing */
   bool active = false;
   bool pend = false;                  /* OUT_CHAIN: finished, its bounce rays not spawned yet */
   bool exhausted = n == 0;            /* wave-uniform: the queue has no more rays to fetch */
   bool wave_par = false;              /* wave-uniform: some lane may hold a ray with a parallel axis */
   bool wave_lit = false;              /* wave-uniform: some lane holds a ray the short box test is not valid for */
   bool burst_filter = false;          /* wave-uniform: some lane of the current burst of node steps is in the second tree */
   RayCtx c;
   c.r.o = mk3(0, 0, 0); c.r.d = mk3(1, 1, 1); c.yx = c.yy = c.yz = 1.0f; c.par = 0; c.fast = true; c.wide = 0.0f;
   float t = FLT_MAX, toBeat = -FLT_MAX;
   int closest = -1, cur = RT_CUR_IDLE, sp = 0, tag = 0;   /* cur: node index, ~primitive, or RT_CUR_IDLE while the lane has no ray */
   int mode = MODE_EXTEND;             /* OUT_CHAIN: what this lane's ray is; tag: depth level + 1 (closest-hit rays) or the light (feelers) */
   unsigned idx = 0;
   /* Every atomic on a word that all waves of the launch share costs the chip about 12 ns (they are served one
    * after the other, wherever the waves run): 100 000 of them are a millisecond.  So the primaries are handed out
    * through RT_FETCH_STREAMS counters instead of one, queue entries for bounce rays are reserved a range at a time
    * ([res_lo, res_hi)), and a wave sums its statistics in registers and adds them once, when it is done. */
   const unsigned long long n_patches = (n + 63ULL) >> 6;
   unsigned stream = (blockIdx.x * (RT_BLOCK / 64) + (threadIdx.x >> 6)) % RT_FETCH_STREAMS;   /* wave-uniform: the stream this wave fetches from */
   unsigned res_lo = 0, res_hi = 0;            /* wave-uniform: OUT_CHAIN, queue entries reserved, not yet used */
   bool res_full = false;                      /* wave-uniform: the queue had no entries left to reserve */
   unsigned n_skipped = 0;                     /* OUT_CHAIN: shadow feelers this lane did not have to trace */
   unsigned w_shaded = 0, w_refl = 0, w_refr = 0;   /* wave-uniform: OUT_CHAIN, shade nodes and bounce rays of this wave */
   unsigned nv = 0, np = 0, nt = 0, wn = 0, wl = 0;
   unsigned nvs = 0, nps = 0, nts = 0;                       /* COUNT only: the same for shadow feelers */
   unsigned ray_steps = 0, max_ray_steps = 0;   /* COUNT only: wave steps this ray has been in flight */
   unsigned long long clk_refill = 0, clk_node = 0, clk_leaf = 0;   /* COUNT only: 100 MHz ticks this wave spent in each part of its loop */
#ifdef RT_DEBUG_REFILL_CLK
   unsigned long long dbg_wait = 0, dbg_pre = 0, dbg_c = 0, dbg_d = 0;
#endif
#ifdef RT_PROBE_FEELERS
   bool probe_ref = false;
#endif
   unsigned n_unneeded = 0;                             /* OUT_CHAIN: feelers that stopped after their own-surface test because their answer cannot matter */
   unsigned n_shadow_cert = 0;                          /* COUNT only: feelers reported "shadowed" from the second tree and one path of the reference's */
   unsigned n_cert = 0, n_rival = 0, n_unreached = 0;   /* COUNT only: how this lane's closest-hit rays left the second tree with a hit */
   unsigned n_leafwait = 0, n_noray = 0, n_refills = 0;   /* COUNT only: node steps this lane sat out at a leaf / without a ray; refills of the wave */

   /* a ray is finished: scan the planes unless it left early, write its result */
   auto finish = [&](bool early) {
      if (TWO && OUT == OUT_CHAIN && mode == (MODE_EXTEND | MODE_FILTER) && closest >= 0) {
         /* The second tree has been searched: `closest` at t is the nearest primitive the ray meets, toBeat the nearest
          * of all the others.  If those are more than EQUAL_EPSILON apart (and t is clear of the parameters of
          * primitives behind the origin), the reference's walk reports this primitive whatever its order -- nothing else
          * can tie with it or beat it -- provided it reaches the leaf at all; that is checked on the reference's tree,
          * along the one path from its root to that leaf.  Otherwise the reference's walk is made in full. */
         const bool alone = t > 2.0f * RT_EQ_EPS && !(t + RT_EQ_EPS > toBeat);
         if (COUNT) { if (alone) n_cert++; else n_rival++; }
         if (alone) { mode = MODE_EXTEND | MODE_VERIFY; toBeat = __int_as_float(sc.leaf_parent[closest]); cur = sc.root; }
         else { mode = MODE_EXTEND; toBeat = -FLT_MAX; t = FLT_MAX; closest = -1; cur = sc.root; sp = 0; }
         return;
      }
      if (TWO && OUT == OUT_CHAIN && (mode & (MODE_KIND | MODE_FILTER)) == (MODE_SHADOW | MODE_FILTER) && closest >= 0) {
         /* A feeler that has searched the second tree and met something in front of its target.  The reference's walk
          * stops at the first primitive it accepts with a parameter below the target's (kernel.cu:117-121, 128-132) and
          * reports "shadowed" if that parameter is below it by more than EPSILON (kernel.cu:206).  `closest` is in front
          * by more than EQUAL_EPSILON, so no tie with a hit at or beyond the target can keep the walk from accepting it;
          * nothing the ray meets lies in the last EPSILON before the target (MODE_BAND is clear), so whatever the walk
          * accepts first, it reports "shadowed" -- provided it reaches the leaf of `closest` if it accepts nothing else
          * before, and that is checked on the reference's tree, along the one path from its root to that leaf.
          * (With anything less clear-cut the feeler has left the second tree at once: see the leaf step.) */
         if (COUNT) n_shadow_cert++;
         mode = MODE_SHADOW | MODE_VERIFY; t = __int_as_float(sc.leaf_parent[closest]); cur = sc.root;
         return;
      }
      if (COUNT) { max_ray_steps = ray_steps > max_ray_steps ? ray_steps : max_ray_steps; }
      if (!early) plane_scan(sc, c.r, t, closest);
      if (TWO && OUT == OUT_CHAIN && (mode & (MODE_KIND | MODE_FILTER | MODE_VERIFY)) == (MODE_SHADOW | MODE_FILTER) && t < toBeat) {
         /* A feeler the filter found nothing for, but with a plane in front of its target: whether that plane counts
          * depends, within EPSILON, on what the reference's walk leaves in t, so that walk is made after all. */
         mode = MODE_SHADOW; cur = sc.root; sp = 0; t = FLT_MAX; closest = -1;
         return;
      }
      active = false; cur = RT_CUR_IDLE;
#ifdef RT_DEBUG_WAVES
      dbg_fin_clock = wall_clock64(); dbg_fin_start = dbg_ray_start;
      dbg_fin_info = (unsigned long long)(unsigned)mode | ((unsigned long long)(unsigned)(tag & 0xff) << 8) | ((unsigned long long)born << 16);
#endif
#ifdef RT_PROBE_FEELERS
      if (COUNT && OUT == OUT_CHAIN && (mode & MODE_KIND) != MODE_EXTEND) {
         if (probe_ref) { n_rival++; if (less_than(t, toBeat)) n_cert++; }
         if (less_than(t, toBeat)) n_unreached++;
         probe_ref = false;
      }
#endif
      if (OUT == OUT_CHAIN && (mode & MODE_KIND) != MODE_EXTEND) {
         /* kernel.cu:206; a feeler that is not shadowed leaves its bit as the launch found it: clear */
         if (less_than(t, toBeat)) atomicOr(A()->flag_words + (size_t)idx * A()->flag_stride + ((unsigned)tag >> 5), 1u << ((unsigned)tag & 31u));
         /* A feeler that took long (the reference's walk, for one the second tree could not answer) belongs to a tile
          * that should start early next frame: the tile's cost is at least this feeler's.  (Only the feelers of
          * primaries say which tile they belong to; they are few.) */
         if (A()->tile_cost && born > RT_LONG_FEELER && idx < (unsigned)A()->n) atomicMax(A()->tile_cost + 2 * fd_div(idx, A()->g.per_tile_div), born);
      } else {
         Hit h; h.prim = closest; h.t = t; hits[idx] = h;
         if (OUT == OUT_CHAIN) {
            if (closest >= 0) {
               if (A()->dur) A()->dur[idx] = born;
               pend = true;
               if (tag == 1) {
                  store_qray(A()->rays, idx, c.r.o, c.r.d, -FLT_MAX, 1);      /* a primary: made on fetch, not stored then (bounce rays are, when they are spawned) */
                  if (A()->tile_hit) A()->tile_hit[fd_div(idx, A()->g.per_tile_div)] = 1u;
               }
            } else if (tag != 1) {
               /* a bounce ray that hit nothing: nothing to spawn.  (Of a primary that hit nothing only the hit is stored:
                * the kernels behind this one look at it first and want nothing else.) */
               if (A()->dur) A()->dur[idx] = born;
               *reinterpret_cast<float4 *>(A()->recs + idx) = make_float4(0.0f, 0.0f, __int_as_float(-1), __int_as_float(-1));
            }
         }
      }
   };
   /* The entry a pop would take first, read ahead of the work that decides whether a pop happens: a step either
    * pushes or pops, never both, so the top of the stack at the start of a step is the entry its pop starts with,
    * and the LDS round trip overlaps the node or primitive fetch instead of following it. */
   auto peek = [&]() {
      StackEntry e = stack_get(sp > 0 ? sp - 1 : 0);
      asm volatile("" : "+v"(e.child), "+v"(e.dist));      /* one 64-bit read, issued here */
      return e;
   };
   /* Boxes at or beyond this are not entered.  The reference's walk: t.  A closest-hit ray in the second tree looks a
    * little further than its nearest hit so far (RT_LOOK_*); a feeler there carries its limit in t already. */
   auto look_cut = [&]() {
      const bool widen = TWO && OUT == OUT_CHAIN && mode == (MODE_EXTEND | MODE_FILTER);
      return widen ? __builtin_fmaf(t, 1.0f + RT_LOOK_REL, RT_LOOK_ABS) : t;
   };
   /* pop deferred siblings until one is still closer than t (kernel.cu:160-186); none left = done.  `cut` is t, or
    * what stands in for it while the ray is in the second tree (see look_cut). */
   auto pop = [&](StackEntry top, float cut) {
      /* the first try uses the entry read ahead (nearly always enough: an entry goes stale only when t has shrunk
       * since it was pushed); the loop behind it walks past stale entries */
      bool found = false;
      if (sp > 0) {
         sp--;
         found = top.dist < cut;
         if (found) cur = top.child;
      }
      if (!found) {
         while (sp > 0) {
            sp--;
            StackEntry e = stack_get(sp);
            if (e.dist < cut) { cur = e.child; found = true; break; }
         }
         if (!found) finish(false);
      }
   };
   /* one node step of this lane (kernel.cu:136-158) */
   auto node_step = [&](auto variant_tag) {
      constexpr int VARIANT = decltype(variant_tag)::value;
      /* the record: two boxes and the two child references.  The last fetch is kept whole until the box tests are
       * done: the compiler fetches 16 bytes for the 8 that are used, and would otherwise hand the two spare
       * registers out at once -- to the stack read below, which then had to wait for the fetch it should overlap */
      const char *rec = reinterpret_cast<const char *>(sc.nodes) + ((unsigned)cur << 6);
      float4 na = *reinterpret_cast<const float4 *>(rec);
      float4 nb = *reinterpret_cast<const float4 *>(rec + 16);
      float4 nc = *reinterpret_cast<const float4 *>(rec + 32);
      int4 tail = *reinterpret_cast<const int4 *>(rec + 48);
      int2 ch; ch.x = tail.x; ch.y = tail.y;
      StackEntry top = peek();
      if (COUNT) { if (OUT == OUT_CHAIN && (mode & MODE_KIND) != MODE_EXTEND) nvs++; else nv++; }
#ifdef RT_PROBE_FEELERS
      if (COUNT && OUT == OUT_CHAIN && (mode & MODE_KIND) != MODE_EXTEND && !(mode & (MODE_FILTER | MODE_VERIFY))) { nts++; probe_ref = true; }
#endif
      born++;
      float left, right;
      if (VARIANT == 2) {
         left = slab_literal(c.r, na.x, na.y, na.z, na.w, nb.x, nb.y);
         right = slab_literal(c.r, nb.z, nb.w, nc.x, nc.y, nc.z, nc.w);
      } else {
         /* in the second tree the boxes are wider by the ray's own margin (RayCtx::wide); in the reference's, as they are */
         const float w = (TWO && (mode & MODE_FILTER)) ? c.wide : 0.0f;
         const F3 up = mk3(c.r.o.x + w, c.r.o.y + w, c.r.o.z + w), dn = mk3(c.r.o.x - w, c.r.o.y - w, c.r.o.z - w);
         SlabAB sl = slab_ab<VARIANT == 1>(c, up, dn, na.x, na.y, na.z, na.w, nb.x, nb.y);
         SlabAB sr = slab_ab<VARIANT == 1>(c, up, dn, nb.z, nb.w, nc.x, nc.y, nc.z, nc.w);
         left = slab_value(sl); right = slab_value(sr);
         if (TWO && burst_filter) {
            /* a ray still in the miss filter's tree asks the filter's question of the same two numbers */
            const bool filtering = (mode & MODE_FILTER) != 0;
            float fl = slab_filter_value(sl), fr = slab_filter_value(sr);
            left = filtering ? fl : left; right = filtering ? fr : right;
         }
      }
      /* kernel.cu:148-158 as predicates.  A child that is not ahead (box value not above zero) gets distance
       * +inf; then "right first on `<=`" is r <= l, "closer than the hit so far" is < t (t is finite or
       * FLT_MAX), and the other child is deferred exactly when its distance is not +inf. */
      asm volatile("" : : "v"(tail.z), "v"(tail.w));
      const float inf = __builtin_inff();
      float l = above_zero(left) ? left : inf, r = above_zero(right) ? right : inf;
      const float cut = look_cut();
      bool goR = r <= l && r < cut;
      bool goL = !goR && l < cut;
      float other = goR ? l : r;
      if (TWO && OUT == OUT_CHAIN) {
         /* A ray on its way down to the leaf of `closest` (the leaf hangs below node toBeat; nodes are numbered parent,
          * left subtree, right subtree) takes the turn towards it and defers nothing.  The reference's walk takes the
          * same turn here -- at once, or later from its stack -- if the child's box is ahead and starts in front of every
          * t the walk can hold before it reaches that leaf, and the least of those is beyond t + EQUAL_EPSILON
          * (kernel.cu:139-175).  If not, nothing is known: the reference's walk is made in full. */
         if (mode & MODE_VERIFY) {
            /* (a feeler on that way keeps the node number in t and its target's parameter in toBeat: the reference's walk
             * holds no t below the target's parameter before it stops) */
            const bool feeler = (mode & MODE_KIND) != MODE_EXTEND;
            const int hangs_at = __float_as_int(feeler ? t : toBeat);
            const float limit = feeler ? toBeat : t + RT_EQ_EPS;
            const bool right_turn = ch.y >= 0 ? hangs_at >= ch.y : ~ch.y == closest;
            if ((right_turn ? r : l) < limit) {
               int next = right_turn ? ch.y : ch.x;      /* at the leaf: the leaf step ends the ray */
               /* From the root straight to the leaf's parent, for a ray that does not start inside the box it turns into
                * here.  Every box further down the path lies inside this one, coordinate by coordinate, and the box test
                * is a chain of correctly rounded operations, each monotone in the bound it takes: so the ray starts
                * inside none of them (GeometryUtil.h:41-45 compares the same coordinates), their values are where the ray
                * enters them, and those are no smaller than this box's (which is ahead: it has just been found so) and no
                * larger than the leaf's own box's -- which the leaf's parent, one step from now, compares with the same
                * limit.  If that one passes, every box in between would have passed. */
               if (sc.verify_jump && cur == sc.root && next >= 0 && next != hangs_at) {
                  const float bx0 = right_turn ? nb.z : na.x, by0 = right_turn ? nb.w : na.y, bz0 = right_turn ? nc.x : na.z;
                  const float bx1 = right_turn ? nc.y : na.w, by1 = right_turn ? nc.z : nb.x, bz1 = right_turn ? nc.w : nb.y;
                  const bool inside = c.r.o.x > bx0 && c.r.o.x < bx1 && c.r.o.y > by0 && c.r.o.y < by1 && c.r.o.z > bz0 && c.r.o.z < bz1;
                  if (!inside) next = hangs_at;
               }
               cur = next;
            } else {
               if (COUNT) { if (feeler) { n_shadow_cert--; } else { n_cert--; n_unreached++; } }
               mode &= MODE_KIND; t = FLT_MAX; closest = -1; cur = sc.root; sp = 0;
               if (!feeler) toBeat = -FLT_MAX;
            }
            return;
         }
      }
      if ((goR || goL) && other < inf) {
         StackEntry e; e.child = goR ? ch.x : ch.y; e.dist = other;
         stack_put(sp, e); sp++;
      }
      if (goR || goL) cur = goR ? ch.y : ch.x;
      else pop(top, cut);
   };
   /* A ray starts its walk.  Beside the reference's tree the scene may hold a second one: a surface-area tree over the
    * same primitive boxes, every box widened.  Nothing is answered from it; it is used to prove what the reference's
    * walk returns, and that walk is made whenever the proof does not go through.
    *  - A ray that meets no primitive there at a parameter the reference would accept (a feeler: none in front of its
    *    target by more than EPSILON) is finished as if the reference's walk had been made: that walk would end as it
    *    began, whatever its order.
    *  - A closest-hit ray keeps the nearest primitive it meets there and the nearest of the others; finish() decides
    *    from those whether the reference's walk can be replaced by the path to that primitive's leaf.
    *  - A feeler that meets a primitive in front of its target leaves the second tree and makes the reference's walk
    *    from its root, unchanged.
    * Rays outside the short box test's range do not use the second tree. */
   auto enter_tree = [&](float slack) {
      const int kind_bit = (mode & MODE_KIND) == MODE_EXTEND ? 1 : 2;
      if (TWO && (sc.filter_kinds & kind_bit) && c.fast && !wave_lit && !(c.par & RT_PAR_NO_FILTER)) {
         cur = sc.filter_root; mode |= MODE_FILTER;
         if (OUT == OUT_CHAIN) {
            if ((mode & MODE_KIND) == MODE_EXTEND) toBeat = FLT_MAX;      /* the nearest of the other primitives, so far */
            else t = toBeat * (1.0f + RT_FEELER_REL) + slack;             /* nothing that starts behind the target can shadow it */
         }
      } else cur = sc.root;
   };
   /* out of the second tree with nothing learnt: the reference's walk from its root */
   auto leave_filter = [&]() {
      if (OUT == OUT_CHAIN && (mode & MODE_KIND) == MODE_EXTEND) toBeat = -FLT_MAX;
      mode &= MODE_KIND; t = FLT_MAX; closest = -1; cur = sc.root; sp = 0;
   };

   /* OUT_CHAIN: every lane that finished since the last call writes its spawn record and appends its bounce rays */
   auto spawn_stage = [&]() {
      bool want_refl = false, want_refr = false, shaded = false;
      float m_dif = 0.0f, m_spec = 0.0f, m_rough = 0.0f;
      bool cont_refl = false, cont_refr = false;      /* this lane goes on with its reflection / refraction ray itself */
      bool front_pushed = false;                      /* wave-uniform: entries were put at the front of the list */
      BounceRays b;
      SpawnRec out;
      out.refl = out.alpha = 0.0f; out.child_refl = out.child_refr = -1;
      if (pend) {
         shaded = true;
         unsigned mat_id = closest < sc.n_prims ? sc.prims[closest].material : sc.planes[closest - sc.n_prims].material;
         const rt_material &m = sc.materials[mat_id];
         float m_refl = m.refl, m_refr = m.refr, m_ior = m.ior;
         out.refl = m_refl; out.alpha = m.alpha;
         m_dif = m.dif; m_spec = m.spec; m_rough = m.rough;
         if (tag < A()->max_rec && (m_refl > 0.0f || m_refr > 0.0f)) {          /* tag = level + 1 */
            b = bounce_rays(sc.prims, sc.xforms, sc.shades, sc.planes, sc.n_prims, c.r.o.x, c.r.o.y, c.r.o.z,
                            c.r.d.x, c.r.d.y, c.r.d.z, closest, t, m_refl, m_refr, m_ior);
            want_refl = b.want_refl != 0; want_refr = b.want_refr != 0;
         }
      }
      w_shaded += (unsigned)__popcll(__ballot(shaded));
      bool want_sh = shaded && sc.n_lights > 0;
      unsigned long long mR = __ballot(want_refl), mT = __ballot(want_refr), mS = __ballot(want_sh);
      unsigned cR = __popcll(mR), cT = __popcll(mT), cS = __popcll(mS);
      if (cR + cT + cS) {
         const bool room = list != nullptr && list_count + (int)(cR + cT + cS) <= (int)A()->list_cap;
         if (!room) {
            /* The list is full: nothing of this stage is started (no queue entry is taken, the nodes keep "no child"),
             * the frame is flagged and the host renders it again with longer lists. */
            if (lane == 0) stat_add(stats, ST_LIST_OVERFLOW, 1ULL);
         } else {
            if (want_sh) {
               /* the shaded point (kernel.cu:340) and its primitive, once, for the feelers of all its lights */
               F3 P = ray_point(c.r, t);
               A()->points[idx] = make_float4(P.x, P.y, P.z, __int_as_float(closest));
               /* ... and its shadow bits start clear (only entries that are shaded have bits anyone reads, so they are
                * cleared here rather than all at once before the launch; the feelers that set them start in later
                * refills, behind the fence of a refill that takes list entries) */
               if (!A()->skip_unlit) {
                  for (unsigned w = 0; w < A()->flag_stride; w++) A()->flag_words[(size_t)idx * A()->flag_stride + w] = 0u;
               } else {
                  /* ... except those of the lights whose feeler cannot matter, with the expressions of PhongShader.h:12-22 on
                   * the values the light sums will use (same ray, same t: same floats) */
                  unsigned word = 0u;
                  const F3 nrm = normal_at(sc, closest, c.r, t);
                  const F3 eye = normalize3(-c.r.d);
                  for (int l = 0; l < sc.n_lights; l++) {
                     const rt_dev_light &Lt = sc.lights[l];
                     const F3 lightDir = normalize3(mk3(Lt.pos[0], Lt.pos[1], Lt.pos[2]) - P);
                     const float d1 = clamp01(dot3(nrm, lightDir));
                     const F3 reflect = (2.0f * dot3(lightDir, nrm)) * nrm - lightDir;
                     const float sp1 = clamp01(dot3(reflect, eye));
                     /* (dif * d1) * colour is zero when dif * d1 is; spec * pow(s, 1/rough) is zero when spec is (the power
                      * of a number in [0, 1] with a positive exponent is finite) or when s is (a positive power of zero) */
                     const bool no_diffuse = (m_dif * d1) == 0.0f;
                     const bool no_specular = m_rough > 0.0f && (m_spec == 0.0f ? sp1 == sp1 : sp1 == 0.0f);
                     if (no_diffuse && no_specular) word |= 1u << (l & 31);
                     if ((l & 31) == 31 || l + 1 == sc.n_lights) { A()->flag_words[(size_t)idx * A()->flag_stride + ((unsigned)l >> 5)] = word; word = 0u; }
                  }
               }
            }
            const unsigned cnt = cR + cT;
            if (cnt > res_hi - res_lo && !res_full) {
               /* a new range of queue entries; what is left of the old one is never used and reads as a miss */
               for (unsigned u = res_lo + lane; u < res_hi; u += 64) { Hit h; h.prim = -1; h.t = FLT_MAX; hits[u] = h; }
               unsigned base = 0;
               if (lane == 0) base = atomicAdd(A()->tail, A()->reserve);
               base = __shfl(base, 0);
               if (base >= A()->cap) { res_lo = res_hi = 0; res_full = true; }
               else { res_lo = base; res_hi = A()->cap - base < A()->reserve ? A()->cap : base + A()->reserve; }
            }
            unsigned slotR = res_lo + __popcll(mR & below), slotT = res_lo + cR + __popcll(mT & below);
            bool okR = want_refl && slotR < res_hi, okT = want_refr && slotT < res_hi;
            unsigned long long oR = __ballot(okR), oT = __ballot(okT);
            unsigned nR = __popcll(oR), nT = __popcll(oT);
            res_lo += cnt < res_hi - res_lo ? cnt : res_hi - res_lo;
            if (okT) {
               store_qray(A()->rays, slotT, mk3(b.to[0], b.to[1], b.to[2]), mk3(b.td[0], b.td[1], b.td[2]), -FLT_MAX, tag + 1);
               out.child_refr = (int)slotT;
            }
            if (okR) {
               store_qray(A()->rays, slotR, mk3(b.ro[0], b.ro[1], b.ro[2]), mk3(b.rd[0], b.rd[1], b.rd[2]), -FLT_MAX, tag + 1);
               out.child_refl = (int)slotR;
            }
            /* The shadow items go to the back of the list.  A lane with a bounce ray traces it itself, at once: the
             * bounce rays are the launch's longest dependent chains (up to six levels, each waiting for the one
             * before), and a chain that queued behind the list's backlog at every level would set the length of the
             * launch.  A lane with two bounce rays keeps the reflection and puts the refraction ray at the front of the
             * list, ahead of every shadow item, for the same reason. */
            if (want_sh) list_put(list_count + __popcll(mS & below), make_uint2(RT_SHADOW_ITEM | idx, 0u));   /* light 0 of entry idx */
            list_count += (int)cS;
            const bool second = okR && okT;
            const unsigned long long m2 = __ballot(second);
            if (m2) {
               const int n2 = (int)__popcll(m2);
               list_head = (list_head - n2) & list_mask;
               if (second) list_put((int)__popcll(m2 & below), make_uint2(slotT, 0u));
               list_count += n2;
               front_pushed = true;
            }
            list_peak = list_count > list_peak ? list_count : list_peak;
            w_refl += cR; w_refr += cT;
            if (nR + nT != cnt && lane == 0) stat_add(stats, ST_OVERFLOW, 1ULL);   /* the queue is full: the host retries with a larger one */
            cont_refl = okR; cont_refr = okT && !okR;
            /* a tile none of whose primaries spawns anything is shaded and resolved in one pass (k_resolve) */
            if (A()->tile_bounce && tag == 1 && (okR || okT)) A()->tile_bounce[fd_div(idx, A()->g.per_tile_div)] = 1u;
         }
      }
      if (pend) *reinterpret_cast<float4 *>(A()->recs + idx) = make_float4(out.refl, out.alpha, __int_as_float(out.child_refl), __int_as_float(out.child_refr));
      pend = false;
      if (cont_refl || cont_refr) {
         Ray r;
         r.o = cont_refl ? mk3(b.ro[0], b.ro[1], b.ro[2]) : mk3(b.to[0], b.to[1], b.to[2]);
         r.d = cont_refl ? mk3(b.rd[0], b.rd[1], b.rd[2]) : mk3(b.td[0], b.td[1], b.td[2]);
         float slack;
         c = make_ray_ctx(sc, r, slack);
         toBeat = -FLT_MAX; idx = (unsigned)(cont_refl ? out.child_refl : out.child_refr); tag = tag + 1; mode = MODE_EXTEND;
         t = FLT_MAX; closest = -1; sp = 0; enter_tree(slack);
         active = true; ray_steps = 0; born = 0; RT_DBG_RAY_START;
         if (cur == RT_ROOT_EMPTY) finish(false);
      }
      return front_pushed;
   };

   /* Every wave reaches its exit in a bounded number of steps: a ray visits each node at most once per
    * stack entry, so RT_STEP_CAP is far beyond any real traversal; hitting it means a corrupted scene and
    * is reported through stats[ST_ABORTED] instead of spinning. */
   for (;;) {
      if (++guard > RT_STEP_CAP) { if (lane == 0) stat_add(stats, ST_ABORTED, 1ULL); break; }
      unsigned long long m_active = __ballot(active);
      int n_idle = 64 - __popcll(m_active);
      bool any_pend = OUT == OUT_CHAIN && __ballot(pend) != 0;
      if ((n_idle >= refill_idle || m_active == 0) && (!exhausted || list_count > 0 || any_pend)) {
         const unsigned long long clk0 = COUNT ? wall_clock64() : 0ULL;
         if (COUNT) n_refills++;
         /* what the lanes that finished since the last refill spawn: a lane with a bounce ray goes on with it, the rest
          * goes onto the list */
         if (any_pend) {
            (void)spawn_stage();
            m_active = __ballot(active);
            n_idle = 64 - __popcll(m_active);
         }
         unsigned long long m_idle = ~m_active;
         int rank = __popcll(m_idle & below);
         int take = 0;
         if (OUT == OUT_CHAIN) take = n_idle < list_count ? n_idle : list_count;
         int need = exhausted ? 0 : n_idle - take;
         /* The fetches that say what the idle lanes get: the list entries (from the front: refraction rays first, then
          * the shadow items, oldest first), and the counter for the lanes that get no list entry.
          * The sample slots are dealt out in patches of 64 to RT_FETCH_STREAMS streams (patch p belongs to stream
          * p mod RT_FETCH_STREAMS), each with a counter of its own: the waves of a launch spread over the streams, so that
          * no counter is asked more often than it can answer, and a wave whose stream has run dry moves on to the next one
          * until it has seen them all dry.  The streams advance together, so the slots are still started in (nearly)
          * their order. */
         unsigned long long q_raw = 0;
         if (need > 0 && lane == 0) q_raw = atomicAdd(A()->next + (size_t)stream * 8, (unsigned long long)need);
         /* What this wave stored earlier (list entries, queue entries, shaded points) is read back below, by other lanes
          * of the wave: those stores are complete first.  Only a refill that takes list entries reads anything back. */
         if (take > 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
         uint2 item = make_uint2(0u, 0u);
         const bool own = !active && rank < take;
         if (own) item = list_get(rank);
         list_head = (list_head + take) & list_mask;
         list_count -= take;
         unsigned long long q_base = 0, q_len = 0;
         const unsigned my_stream = stream;
#ifdef RT_DEBUG_REFILL_CLK
         const unsigned long long clkA = wall_clock64();
#endif
#ifdef RT_DEBUG_WAVES
         dbg_refills++;
         if (any_pend) dbg_last_spawn = wall_clock64();
         if (exhausted && dbg_list_at_dry == ~0ULL) dbg_list_at_dry = (unsigned long long)list_count;
         if (need > 0) { dbg_last_fetch = wall_clock64(); dbg_fetched += (unsigned)need; }
#endif
         if (need > 0) {
            q_base = __shfl(q_raw, 0);
#ifdef RT_DEBUG_REFILL_CLK
            if (COUNT) { asm volatile("" : "+v"(q_base)); dbg_wait += wall_clock64() - clkA; dbg_pre += clkA - clk0; }
#endif
            q_len = n_patches > stream ? ((n_patches - stream + RT_FETCH_STREAMS - 1) / RT_FETCH_STREAMS) * 64ULL : 0ULL;
            if (q_base + (unsigned long long)need >= q_len) {
               /* This stream has run dry.  One look at all the counters (lane i reads stream i's) says which streams
                * still have slots: the wave moves to the next of those, or is done fetching.  A counter only grows, so
                * a stream seen dry stays dry; one seen live may be dry by the time the wave asks it, and then the wave
                * looks again. */
               static_assert(RT_FETCH_STREAMS == 64, "one lane per stream");
               const unsigned long long seen = __hip_atomic_load(A()->next + (size_t)lane * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               const unsigned long long len_i = n_patches > lane ? ((n_patches - lane + RT_FETCH_STREAMS - 1) / RT_FETCH_STREAMS) * 64ULL : 0ULL;
               unsigned long long live = __ballot(seen < len_i) & ~(1ULL << stream);
               if (live == 0) exhausted = true;
               else {
                  /* the waves that leave a dry stream spread over the live ones (all on the nearest one, they would
                   * queue at its counter) */
                  const unsigned wave_id = blockIdx.x * (RT_BLOCK / 64) + (threadIdx.x >> 6);
                  const unsigned pick = wave_id % (unsigned)__popcll(live);
                  const unsigned long long sel = __ballot(((live >> lane) & 1ULL) != 0 && (unsigned)__popcll(live & below) == pick);
                  stream = (unsigned)__builtin_ctzll(sel);
               }
            }
         }
         bool put_back = false;
         if (!active) {
            if (OUT == OUT_CHAIN && own && (item.x & RT_SHADOW_ITEM)) {
               /* the feeler of light k for queue entry `node` (PointLight.h:18-20): from the light to the shaded point */
               const unsigned node = item.x & ~RT_SHADOW_ITEM, k = item.y;
               unsigned long long *pp = reinterpret_cast<unsigned long long *>(A()->points + node);
               unsigned long long p01 = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               unsigned long long p23 = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
               F3 P = mk3(__uint_as_float((unsigned)p01), __uint_as_float((unsigned)(p01 >> 32)), __uint_as_float((unsigned)p23));
               int hprim = (int)(unsigned)(p23 >> 32);
               const rt_dev_light &Lt = sc.lights[k];
               Ray feeler; feeler.o = mk3(Lt.pos[0], Lt.pos[1], Lt.pos[2]); feeler.d = P - feeler.o;
               if ((int)k + 1 < sc.n_lights) { item.y = k + 1; put_back = true; }     /* same entry, next light */
               bool unneeded = false;
               if (A()->skip_unlit) {
                  const unsigned fw = __hip_atomic_load(A()->flag_words + (size_t)node * A()->flag_stride + (k >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  unneeded = ((fw >> (k & 31u)) & 1u) != 0u;
               }
               float slack;
               c = make_ray_ctx(sc, feeler, slack);
               tag = (int)k; idx = node;
               t = FLT_MAX; closest = -1; sp = 0; ray_steps = 0; born = 0; RT_DBG_RAY_START;
               if (hprim >= sc.n_prims) {
                  /* planes are not in the tree: their own-surface test is done here */
                  float param = plane_intersect(sc, hprim - sc.n_prims, feeler);   /* kernel.cu:351 */
                  if (param == -1.0f) n_skipped++;          /* lit, whatever the walk would find: its bit stays clear */
                  else if (unneeded) n_unneeded++;          /* nobody looks at its bit */
                  else {
                     toBeat = param; mode = MODE_SHADOW; enter_tree(slack); active = true;
                     if (cur == RT_ROOT_EMPTY) finish(false);
                  }
               } else {
                  /* (until its own-surface test is done the feeler keeps its slack where its t will be) */
                  mode = unneeded ? (MODE_SETUP | MODE_UNNEEDED) : MODE_SETUP; cur = ~hprim; toBeat = -FLT_MAX; t = slack; active = true;
               }
            } else {
               bool have = false;
               unsigned long long my = 0;
               if (own) { my = item.x; have = true; }
               else if (need > 0) {
                  const unsigned long long q = q_base + (unsigned long long)(rank - take);
                  my = ((q >> 6) * RT_FETCH_STREAMS + my_stream) * 64ULL + (q & 63ULL);
                  have = q < q_len && my < n;
                  if (have && OUT == OUT_CHAIN && A()->tile_order) {
                     /* the k-th tile to be started is tile_order[k] of this launch's tiles; where a sample's results
                      * are stored does not change */
                     unsigned f = (unsigned)my, k = fd_div(f, A()->g.per_tile_div);
                     my = (unsigned long long)A()->tile_order[k] * A()->g.per_tile + (f - k * A()->g.per_tile);
                  }
               }
#ifdef RT_DEBUG_REFILL_CLK
               if (COUNT) { asm volatile("" : "+v"(my)); dbg_c += wall_clock64() - clk0; }
#endif
               if (have) {
                  QRay q;
                  if (own) q = load_qray_written(A()->rays, my);
                  else if (OUT == OUT_CHAIN) {
                     /* generateCameraRays (kernel.cu:477-498) for sample slot `my`; the ray goes into the queue only if it
                      * hits something (the light sums need it then), see finish */
                     int X = 0, Y = 0;
                     bool inside = slot_to_sample(A()->g, (unsigned)my, X, Y);
                     Ray pr; pr.o = mk3(0, 0, 0); pr.d = mk3(0, 0, 1);
                     if (inside) pr = camera_ray(A()->cam, A()->g, X, Y);
                     q.ox = pr.o.x; q.oy = pr.o.y; q.oz = pr.o.z; q.to_beat = -FLT_MAX;
                     q.dx = pr.d.x; q.dy = pr.d.y; q.dz = pr.d.z; q.valid = inside ? 1 : 0;
                     if (!inside) {
                        /* a slot outside the frame: the result of a miss, no follow-up work */
                        Hit h; h.prim = -1; h.t = FLT_MAX; hits[my] = h;
                     }
                  } else q = load_qray_rec(A()->rays, my);
#ifdef RT_DEBUG_REFILL_CLK
                  if (COUNT) { asm volatile("" : "+v"(q.dx), "+v"(q.ox)); dbg_d += wall_clock64() - clk0; }
#endif
                  if (q.valid) {
                     Ray r; r.o = mk3(q.ox, q.oy, q.oz); r.d = mk3(q.dx, q.dy, q.dz);
                     float slack;
                     c = make_ray_ctx(sc, r, slack);
                     toBeat = q.to_beat; idx = (unsigned)my; tag = q.valid; mode = MODE_EXTEND;
                     t = FLT_MAX; closest = -1; sp = 0; enter_tree(slack);
                     active = true; ray_steps = 0; born = 0; RT_DBG_RAY_START;
                     if (cur == RT_ROOT_EMPTY) finish(false);
                  }
               }
            }
         }
         if (OUT == OUT_CHAIN) {
            unsigned long long mP = __ballot(put_back);
            if (mP) {
               /* at most as many go back as were taken, so there is room */
               if (put_back) list_put(list_count + __popcll(mP & below), item);
               list_count += __popcll(mP);
            }
         }
         m_active = __ballot(active);
         wave_par = __ballot(active && (c.par & RT_PAR_AXES) != 0) != 0;
         wave_lit = __ballot(active && !c.fast) != 0;
         /* the literal box test has no reading for the second tree's question: a wave that has to use it keeps no ray there */
         if (TWO && wave_lit && active && (mode & MODE_FILTER)) leave_filter();
         if (COUNT) clk_refill += wall_clock64() - clk0;
      }
      if (m_active == 0) {
         if (exhausted && list_count == 0 && !(OUT == OUT_CHAIN && __ballot(pend) != 0)) break;
         continue;
      }
      /* ---- node steps ---- */
      /* the box-test variant is the same for a whole burst (the flags only change in a refill): one loop each */
      auto burst_loop = [&](auto variant_tag) {
         /* no ray enters the second tree during a burst (that happens in refills and leaf steps) */
         burst_filter = TWO && __ballot(active && (mode & MODE_FILTER)) != 0;
#pragma unroll 1
         for (int burst = 0; burst < node_burst; burst++) {
            bool in_node = cur >= 0;                 /* idle lanes hold RT_CUR_IDLE */
            if (__ballot(in_node) == 0) break;
            /* enough lanes wait at a leaf: their step is worth its price now */
            if ((int)__popcll(__ballot(cur < 0 && cur != RT_CUR_IDLE)) >= leaf_trigger) break;
            if (COUNT) {
               wn++; if (active) ray_steps++;
               if (cur < 0) { if (cur == RT_CUR_IDLE) n_noray++; else n_leafwait++; }
            }
            if (in_node) node_step(variant_tag);
         }
      };
      const unsigned long long clk1 = COUNT ? wall_clock64() : 0ULL;
      if (wave_lit) burst_loop(std::integral_constant<int, 2>());
      else if (wave_par) burst_loop(std::integral_constant<int, 1>());
      else burst_loop(std::integral_constant<int, 0>());
      const unsigned long long clk2 = COUNT ? wall_clock64() : 0ULL;
      if (COUNT) clk_node += clk2 - clk1;
      /* ---- leaf step ---- */
      bool at_leaf = active && cur < 0;
      if (__ballot(at_leaf) != 0) {
         if (COUNT) { wl++; if (active) ray_steps++; }
         bool tie = false, take = false, tested = false;
         float leaf_dist = 0.0f;
         StackEntry top; top.child = 0; top.dist = 0.0f;
         if (at_leaf) {
            born += 4;                   /* a leaf step weighs about as much as a few node steps */
            top = peek();
            float dist = 0.0f;
            if (!(TWO && OUT == OUT_CHAIN && (mode & MODE_VERIFY))) dist = prim_intersect(sc, ~cur, c.r);
            if (TWO && OUT == OUT_CHAIN && (mode & MODE_VERIFY)) {
               /* the way down from the root has arrived: the reference's walk tests this leaf too.  A closest-hit ray's
                * answer is `closest` at t; on to the planes (the stack is empty, the pop below ends the ray).  A feeler's
                * walk stops here at the latest, with a parameter that reports "shadowed". */
               const bool feeler = (mode & MODE_KIND) != MODE_EXTEND;
               mode &= MODE_KIND; sp = 0;
               if (!feeler) toBeat = -FLT_MAX;
               if (~cur == closest) { tested = true; if (feeler) { take = true; leaf_dist = -FLT_MAX; } }
               else {
                  /* not the leaf it set out for: the reference's walk in full */
                  t = FLT_MAX; closest = -1; cur = sc.root;
                  if (COUNT) { if (feeler) { n_shadow_cert--; } else { n_cert--; n_unreached++; } }
               }
            } else if (TWO && (mode & MODE_FILTER)) {
               /* still in the filter's tree: does the ray meet this primitive where it could matter? */
               bool shadow = OUT == OUT_CHAIN && (mode & MODE_KIND) != MODE_EXTEND;
               if (COUNT) { if (shadow) nps++; else np++; }
               if (OUT == OUT_CHAIN && !shadow) {
                  /* a closest-hit ray keeps the nearest primitive it meets (t, closest) and the parameter of the
                   * nearest of the others (toBeat) */
                  if (above_zero(dist)) {
                     if (dist < t) { toBeat = t; t = dist; closest = ~cur; }
                     else toBeat = fminf(toBeat, dist);
                  }
                  tested = true;
               } else if (shadow) {
                  /* A feeler: a primitive the reference's walk would accept and stop at (in front of the target, not
                   * behind the origin) is sorted by how far in front it is -- within EPSILON the walk reports "lit",
                   * within EQUAL_EPSILON a tie with a hit at the target's distance may make it pass the primitive by,
                   * further in front it takes it and reports "shadowed".  The first of the last kind is kept, and the
                   * search goes on for one of the first kind (take stays false: on to the next deferred box); finish()
                   * sees to the rest when the tree has been searched.  With one that says "lit" and one that says
                   * "shadowed", or one that depends on a tie and none that is clear of it yet, the order of the
                   * reference's walk decides: that walk is made, from its root. */
                  bool order_decides = false;
                  if (above_zero(dist) && dist < toBeat) {
                     if (!less_than(dist, toBeat)) { mode |= MODE_BAND; order_decides = closest >= 0; }
                     else if (dist + RT_EQ_EPS > toBeat) order_decides = closest < 0;
                     else { if (closest < 0) closest = ~cur; order_decides = (mode & MODE_BAND) != 0; }
                  }
                  if (order_decides) leave_filter();
                  else tested = true;
               } else {
                  if (above_zero(dist)) leave_filter();     /* the reference's walk, from its root */
                  else tested = true;                       /* take stays false: on to the next deferred box */
               }
            } else if (OUT == OUT_CHAIN && (mode & ~MODE_UNNEEDED) == MODE_SETUP) {
               /* the feeler against its own surface (kernel.cu:351).  A feeler that misses it (-1) can never be
                * reported shadowed: isFloatLessThan(closest, -1) is false for every accepted t.  The reference
                * still walks the tree for it; that walk is skipped here and counted separately. */
               if (dist == -1.0f) { n_skipped++; active = false; cur = RT_CUR_IDLE; }
               else if (mode & MODE_UNNEEDED) { n_unneeded++; active = false; cur = RT_CUR_IDLE; }      /* its bit stays clear; nobody looks at it */
               else {
                  const float slack = t;
                  toBeat = dist; mode = MODE_SHADOW; t = FLT_MAX; enter_tree(slack);
                  if (cur == RT_ROOT_EMPTY) finish(false);
               }
            } else {
               /* what the test does to the traversal state, kernel.cu:104-134; a near-tie with the hit so far is
                * only marked here and decided below */
               bool shadow = OUT == OUT_CHAIN && (mode & MODE_KIND) != MODE_EXTEND;
               if (COUNT) { if (shadow) nps++; else np++; }
               if (is_equal(t, dist)) {
                  if (COUNT) { if (shadow) nts++; else nt++; }
                  tie = closest >= 0;
               } else {
                  take = dist < t && above_zero(dist);
               }
               leaf_dist = dist;
               tested = true;
            }
         }
         if (__ballot(tie) != 0) {
            if (tie) {
               take = tie_prefers_new(sc.prims, sc.xforms, sc.shades, c.r.o.x, c.r.o.y, c.r.o.z, c.r.d.x, c.r.d.y, c.r.d.z,
                                      closest, t, ~cur, leaf_dist);
               tie = false;
            }
         }
         if (tested) {
            bool early = false;
            if (take) { t = leaf_dist; closest = ~cur; early = t < toBeat; }      /* kernel.cu:117-121, 128-132 */
            if (early) finish(true);
            else pop(top, look_cut());
         }
      }
      if (COUNT) clk_leaf += wall_clock64() - clk2;
   }
#ifdef RT_DEBUG_WAVES
   if (lane == 0) {
      unsigned long long *w = dbg_waves + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
      w[0] = dbg_t0; w[1] = dbg_last_fetch; w[2] = wall_clock64(); w[3] = (unsigned long long)list_peak; w[4] = dbg_fetched; w[5] = dbg_refills;
      w[6] = dbg_last_spawn; w[7] = dbg_list_at_dry;
   }
   {
      /* the ray the wave finished last */
      unsigned long long latest = dbg_fin_clock;
      for (int o = 32; o > 0; o >>= 1) { unsigned long long other = __shfl_xor(latest, o); latest = other > latest ? other : latest; }
      const unsigned long long mine = __ballot(dbg_fin_clock == latest);
      if (mine && lane == (unsigned)__builtin_ctzll(mine)) {
         unsigned long long *w = dbg_waves + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
         w[8] = dbg_fin_clock; w[9] = dbg_fin_start; w[10] = dbg_fin_info;
      }
   }
#endif
#ifdef RT_DEBUG_EXIT_HIST
   if (lane == 0) {
      unsigned long long dt = wall_clock64() - dbg_start;
      unsigned bin = (unsigned)(dt / 25000ULL); if (bin > 47) bin = 47;
      stat_add(stats, ST_HIST + (int)bin, 1ULL);
   }
#endif
   if (OUT == OUT_CHAIN) {
      /* what is left of the reserved queue entries reads as a miss (the light sums run over every entry below the tail) */
      for (unsigned u = res_lo + lane; u < res_hi; u += 64) { Hit h; h.prim = -1; h.t = FLT_MAX; hits[u] = h; }
      for (int o = 32; o > 0; o >>= 1) n_skipped += __shfl_xor(n_skipped, o);
      for (int o = 32; o > 0; o >>= 1) n_unneeded += __shfl_xor(n_unneeded, o);
      if (lane == 0) {
         if (n_unneeded) stat_add(stats, ST_SHADOW_UNNEEDED, (unsigned long long)n_unneeded);
         if (list_peak) atomicMax(&stats[ST_MAX_LIST * RT_STAT_REPL], (unsigned long long)list_peak);
         if (n_skipped) stat_add(stats, ST_SHADOW_SKIPPED, (unsigned long long)n_skipped);
         if (w_shaded) stat_add(stats, ST_SHADE_NODES, (unsigned long long)w_shaded);
         if (w_refl) stat_add(stats, ST_REFLECT, (unsigned long long)w_refl);
         if (w_refr) stat_add(stats, ST_REFRACT, (unsigned long long)w_refr);
      }
   }
   if (COUNT) {
      stat_add(stats, ST_NODE_VISITS, (unsigned long long)nv);
      stat_add(stats, ST_PRIM_TESTS, (unsigned long long)np);
      stat_add(stats, ST_TIES, (unsigned long long)nt);
      if (OUT == OUT_CHAIN) {
         stat_add(stats, ST_NODE_VISITS_SHADOW, (unsigned long long)nvs);
         stat_add(stats, ST_PRIM_TESTS_SHADOW, (unsigned long long)nps);
         stat_add(stats, ST_TIES_SHADOW, (unsigned long long)nts);
      }
#ifdef RT_DEBUG_REFILL_CLK
      if (lane == 0) printf("");
      if (lane == 0 && blockIdx.x == 7 && threadIdx.x == 0) printf("refill ticks: total %llu, before the counter %llu, waiting for it %llu, slot known at %llu, ray made at %llu, refills %u\n", clk_refill, dbg_pre, dbg_wait, dbg_c, dbg_d, n_refills);
#endif
      stat_add(stats, ST_CERT_OK, (unsigned long long)n_cert); stat_add(stats, ST_CERT_RIVAL, (unsigned long long)n_rival);
      stat_add(stats, ST_CERT_UNREACHED, (unsigned long long)n_unreached);
      stat_add(stats, ST_SHADOW_CERT, (unsigned long long)n_shadow_cert);
      stat_add(stats, ST_LEAF_WAIT, (unsigned long long)n_leafwait);
      stat_add(stats, ST_NO_RAY, (unsigned long long)n_noray);
      if (lane == 0) {
         stat_add(stats, ST_REFILLS, (unsigned long long)n_refills);
         stat_add(stats, ST_CLK_REFILL, clk_refill); stat_add(stats, ST_CLK_NODE, clk_node); stat_add(stats, ST_CLK_LEAF, clk_leaf);
         stat_add(stats, ST_CLK_WAVE, wall_clock64() - clock_start);
         stat_add(stats, ST_WAVE_NODE_STEPS, (unsigned long long)wn);
         stat_add(stats, ST_WAVE_LEAF_STEPS, (unsigned long long)wl);
      }
      atomicMax(&stats[ST_MAX_RAY_STEPS * RT_STAT_REPL], (unsigned long long)max_ray_steps);
      if (lane == 0) {
         /* when the waves of the launch were done: the mean against the last one shows how much of the launch
          * is its tail (a few waves finishing long dependent chains) */
         unsigned long long now = wall_clock64();
         atomicMax(&stats[ST_CLOCK_FIRST_START * RT_STAT_REPL], ~clock_start);
         atomicMax(&stats[ST_CLOCK_LAST_EXIT * RT_STAT_REPL], now);
         stat_add(stats, ST_CLOCK_EXIT_SUM, now);
         stat_add(stats, ST_WAVES, 1ULL);
      }
   }
}

/* ======================================================================= */
/* ray sources                                                              */
/* ======================================================================= */
/* ======================================================================= */
/* kernels                                                                  */
/* ======================================================================= */

/* tex2D on the texture the reference binds (kernel.cu:32, 590-593): uchar4 texels read as normalised floats, wrap
 * addressing, linear filtering, normalised coordinates -- x = frac(u)*W - 0.5, the four texels around it, weights
 * in 9-bit fixed point with 8 fractional bits, as CUDA documents its linear filter. */
__device__ __forceinline__ F3 texture_sample(const uchar4 *texels, rt_dev_texinfo ti, float u, float v)
{
   int W = ti.width, H = ti.height;
   float x = (u - floorf(u)) * (float)W - 0.5f;
   float y = (v - floorf(v)) * (float)H - 0.5f;
   float fx = floorf(x), fy = floorf(y);
   float a = floorf((x - fx) * 256.0f + 0.5f) * 0.00390625f;
   float b = floorf((y - fy) * 256.0f + 0.5f) * 0.00390625f;
   int i0 = (int)fx, j0 = (int)fy;
   i0 = ((i0 % W) + W) % W; j0 = ((j0 % H) + H) % H;
   int i1 = (i0 + 1) % W, j1 = (j0 + 1) % H;
   const uchar4 *base = texels + ti.offset;
   uchar4 t00 = base[(size_t)j0 * W + i0], t10 = base[(size_t)j0 * W + i1];
   uchar4 t01 = base[(size_t)j1 * W + i0], t11 = base[(size_t)j1 * W + i1];
   float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
   F3 c;
   c.x = w00 * ((float)t00.x / 255.0f) + w10 * ((float)t10.x / 255.0f) + w01 * ((float)t01.x / 255.0f) + w11 * ((float)t11.x / 255.0f);
   c.y = w00 * ((float)t00.y / 255.0f) + w10 * ((float)t10.y / 255.0f) + w01 * ((float)t01.y / 255.0f) + w11 * ((float)t11.y / 255.0f);
   c.z = w00 * ((float)t00.z / 255.0f) + w10 * ((float)t10.z / 255.0f) + w01 * ((float)t01.z / 255.0f) + w11 * ((float)t11.z / 255.0f);
   return c;
}

/* getColor, kernel.cu:323-332: the pigment colour, or the texture at UVAt (Triangle.h:44-52; the other classes
 * return vec2(0): Sphere.h:34-37, Plane.h:32-35, Box.h:25-28) */
__device__ __forceinline__ F3 material_color(const DScene &sc, const rt_material &m, int prim, const Ray &ray, float t)
{
   if (m.tex_id == -1) return mk3(m.clr[0], m.clr[1], m.clr[2]);
   float u = 0.0f, v = 0.0f;
   if (prim < sc.n_prims) {
      PrimRec p = load_prim(sc.prims, prim);
      if (p.m.x == RT_PRIM_TRIANGLE || p.m.x == RT_PRIM_SMOOTH) {
         const float4 *sp = reinterpret_cast<const float4 *>(sc.shades + prim);
         float4 s2 = sp[2], s3 = sp[3];
         F3 p1 = mk3(p.a.x, p.a.y, p.a.z), p2 = mk3(p.a.w, p.b.x, p.b.y), p3 = mk3(p.b.z, p.b.w, p.c.x);
         F3 n = mk3(p.c.y, p.c.z, p.c.w);
         F3 q = ray_point(ray, t);
         float area = dot3(cross3(p2 - p1, p3 - p1), n);
         float beta = dot3(cross3(p1 - p3, q - p3), n) / area;
         float gamma = dot3(cross3(p2 - p1, q - p1), n) / area;
         float alpha = (float)(1.0 - (double)beta - (double)gamma);   /* the reference's "1.0" is a double */
         u = alpha * s2.y + beta * s2.w + gamma * s3.y;                /* vt1 = s[9..10], vt2 = s[11], uv[0], vt3 = uv[1..2] */
         v = alpha * s2.z + beta * s3.x + gamma * s3.z;
      }
   }
   return texture_sample(sc.texels, sc.tex_info[m.tex_id], u, v);
}

/* PhongShader::shade, PhongShader.h:7-25 */
__device__ __forceinline__ F3 phong(F3 matColor, float amb, float dif, float spec, float rough, F3 eye, F3 lightDir,
                                    F3 lightColor, F3 normal, bool inShadow)
{
   F3 light = mk3(0.0f, 0.0f, 0.0f);
   light = light + amb * lightColor;
   if (inShadow) return light * matColor;
   light = light + (dif * clamp01(dot3(normal, lightDir))) * lightColor;
   F3 reflect = (2.0f * dot3(lightDir, normal)) * normal - lightDir;
   light = light + (spec * powf(clamp01(dot3(reflect, eye)), 1.0f / rough)) * lightColor;
   return light * matColor;
}

/* CookTorranceShader::shade, CookTorranceShader.h:7-46 */
__device__ __forceinline__ F3 cook_torrance(F3 matColor, float amb, float dif, float spec, float rough, F3 eye,
                                            F3 lightDir, F3 lightColor, F3 normal, bool inShadow)
{
   F3 light = mk3(0.0f, 0.0f, 0.0f);
   light = light + amb * lightColor;
   if (inShadow) return light * matColor;
   light = light + (dif * clamp01(dot3(normal, lightDir))) * lightColor;
   F3 h = normalize3(lightDir + eye);
   float ior = 2.0f;
   float r0 = (1.0f - ior) * (1.0f - ior) / ((1.0f + ior) * (1.0f + ior));
   float f = r0 + (1.0f - r0) * powf(1.0f - dot3(lightDir, normal), 5.0f);
   float nDotH = dot3(normal, h);
   float mSqr = rough * rough;
   float d = expf((nDotH * nDotH - 1.0f) / (mSqr * nDotH * nDotH)) / (mSqr * nDotH * nDotH * nDotH * nDotH);
   float gStart = 2.0f * nDotH / dot3(eye, h);
   float g1 = gStart * dot3(eye, normal);
   float g2 = gStart * dot3(lightDir, normal);
   float gg = g1 < g2 ? g1 : g2;
   gg = gg < 1.0f ? gg : 1.0f;
   float kSpec = d * f * gg / (4.0f * dot3(eye, normal) * dot3(normal, lightDir));
   light = light + (spec * kSpec) * lightColor;
   return light * matColor;
}

/* what both halves of shadeObject (kernel.cu:335-382) need about a hit */
struct ShadePoint { Ray ray; F3 P, normal, eye; rt_material m; };

__device__ __forceinline__ ShadePoint shade_point(const DScene &sc, const QRay *queue, long long i, Hit h)
{
   ShadePoint sp;
   sp.ray = load_qray(queue, i);
   sp.P = ray_point(sp.ray, h.t);                               /* kernel.cu:340 */
   unsigned mat_id = h.prim < sc.n_prims ? sc.prims[h.prim].material : sc.planes[h.prim - sc.n_prims].material;
   sp.m = sc.materials[mat_id];
   sp.normal = normal_at(sc, h.prim, sp.ray, h.t);              /* kernel.cu:342 */
   sp.eye = normalize3(-sp.ray.d);                              /* kernel.cu:344 */
   return sp;
}

/* start of a chunk: its counters to zero, the queue's tail behind the primaries, and (first chunk) the statistics;
 * one launch instead of a memset each */
__global__ __launch_bounds__(RT_BLOCK) void k_begin_chunk(unsigned long long *stats, int n_stats, unsigned long long *ctr, unsigned *tail,
                                                          unsigned n_primary, unsigned *tile_cost, int n_cost, unsigned *tile_bounce, int n_bounce)
{
   int i = (int)(blockIdx.x * RT_BLOCK + threadIdx.x);
   if (stats && i < n_stats) stats[i] = 0ULL;
   if (tile_cost && i < n_cost) tile_cost[i] = 0u;
   if (tile_bounce && i < n_bounce) tile_bounce[i] = 0u;
   if (i < 8 * (int)RT_FETCH_STREAMS) ctr[i] = 0ULL;       /* the fetch counters; the tail's words follow them */
   if (i == 0) *tail = n_primary;
}

/* light: totalLight of every node, the loop of kernel.cu:347-358 with the shadow flags, over every queue entry
 * some wave reserved (entries reserved and not used read as a miss) */
__device__ __forceinline__ F3 light_sum(const DScene &sc, const QRay *queue, long long i, Hit h, const unsigned *flag_words, unsigned flag_stride,
                                        int shading)
{
   F3 total = mk3(0.0f, 0.0f, 0.0f);
   if (h.prim >= 0) {
      ShadePoint sp = shade_point(sc, queue, i, h);
      F3 matClr = material_color(sc, sp.m, h.prim, sp.ray, h.t);
      const unsigned *fw = flag_words + (size_t)i * flag_stride;
      unsigned bits = 0;
      for (int l = 0; l < sc.n_lights; l++) {
         if ((l & 31) == 0) bits = fw[l >> 5];
         const rt_dev_light &L = sc.lights[l];
         F3 lc = mk3(L.clr[0], L.clr[1], L.clr[2]);
         F3 lightDir = normalize3(mk3(L.pos[0], L.pos[1], L.pos[2]) - sp.P);    /* PointLight.h:14-16 */
         bool inShadow = ((bits >> (l & 31)) & 1u) != 0;
         F3 c = shading == RT_COOK_TORRANCE
                   ? cook_torrance(matClr, sp.m.amb, sp.m.dif, sp.m.spec, sp.m.rough, sp.eye, lightDir, lc, sp.normal, inShadow)
                   : phong(matClr, sp.m.amb, sp.m.dif, sp.m.spec, sp.m.rough, sp.eye, lightDir, lc, sp.normal, inShadow);
         total = total + c;
      }
   }
   return total;
}

/* `tile_bounce` (may be null): primaries of a tile whose flag is clear are left to k_resolve, which shades them itself
 * (a wave's 64 entries lie in one tile: per_tile is a multiple of 64 whenever the flags are in use) */
__global__ __launch_bounds__(RT_BLOCK) void k_light(DScene sc, const QRay *queue, const Hit *hits, const unsigned *flag_words, unsigned flag_stride,
                                                    const unsigned *n_dev, unsigned cap, int shading, float *totals,
                                                    const unsigned *tile_bounce, unsigned n_primary, unsigned per_tile)
{
   /* one workgroup per tile for the primaries (a tile left to k_resolve costs one look at its flag), then one per 256
    * of the entries behind them */
   const unsigned n_tiles = n_primary / per_tile;
   unsigned reserved = *n_dev;
   unsigned n = reserved < cap ? reserved : cap;
   long long first, last, step = RT_BLOCK;
   if (blockIdx.x < n_tiles) {
      if (tile_bounce && tile_bounce[blockIdx.x] == 0u) return;
      first = (long long)blockIdx.x * per_tile + threadIdx.x; last = (long long)(blockIdx.x + 1) * per_tile;
   } else {
      first = (long long)n_primary + (long long)(blockIdx.x - n_tiles) * RT_BLOCK + threadIdx.x; last = first + 1;
   }
   if (last > (long long)n) last = (long long)n;
   for (long long i = first; i < last; i += step) {
      F3 total = light_sum(sc, queue, i, hits[i], flag_words, flag_stride, shading);
      totals[3 * i] = total.x; totals[3 * i + 1] = total.y; totals[3 * i + 2] = total.z;
   }
}

/* combine: the colour of primary i from the light sums of its bounce tree, children before parents:
 *   colour(node) = total*(1 - refl - alpha) + refl*colour(reflected) + alpha*colour(refracted)      (kernel.cu:375-376)
 * with vec3(0) for a ray that was not spawned or hit nothing.  One thread walks one tree (at most
 * 2^RT_MAX_RECURSE - 1 nodes, usually a short chain) with an explicit stack of one frame per level, so every
 * node is evaluated once, by one thread, in the reference's operand order: no atomics, no races. */
__global__ __launch_bounds__(RT_BLOCK) void k_combine_tree(const Hit *hits, const SpawnRec *recs, const float *totals, const unsigned *dur, long long n_primary,
                                                           float *rgb, unsigned *tile_cost, unsigned per_tile, const unsigned *tile_bounce)
{
   /* one workgroup per tile: a tile without a bounce ray is left to k_resolve and costs one look at its flag */
   if (tile_bounce && tile_bounce[blockIdx.x] == 0u) return;
   for (long long i = (long long)blockIdx.x * per_tile + threadIdx.x; i < (long long)(blockIdx.x + 1) * per_tile; i += RT_BLOCK) {
      const bool live = i < n_primary;
      int node[RT_MAX_RECURSE], stage[RT_MAX_RECURSE];
      float rx[RT_MAX_RECURSE], ry[RT_MAX_RECURSE], rz[RT_MAX_RECURSE];
      /* a primary that hit nothing has no spawn record and no cost: its colour is the +0 the expression below gives for
       * a light sum of 0 and weights 0, 0 */
      const bool hit = live && hits[i].prim >= 0;
      int sp = hit ? 0 : -1;
      unsigned chain_refl[RT_MAX_RECURSE];
      unsigned chain = 0;          /* the longest chain of dependent rays below (and with) the node just evaluated, in traversal steps of its rays */
      node[0] = (int)i; stage[0] = 0;
      F3 ret = mk3(0.0f, 0.0f, 0.0f);
      if (hit) {
         /* most trees are a single node: the same expression with both children absent, without the stack */
         float4 rr = *reinterpret_cast<const float4 *>(recs + i);
         if (__float_as_int(rr.z) < 0 && __float_as_int(rr.w) < 0) {
            F3 total = mk3(totals[3 * i], totals[3 * i + 1], totals[3 * i + 2]);
            const F3 none = mk3(0.0f, 0.0f, 0.0f);
            ret = total * (1.0f - rr.x - rr.y) + rr.x * none + rr.y * none;
            if (dur) chain = dur[i];
            sp = -1;
         }
      }
      while (sp >= 0) {
         float4 rr = *reinterpret_cast<const float4 *>(recs + node[sp]);
         int child_refl = __float_as_int(rr.z), child_refr = __float_as_int(rr.w);
         if (stage[sp] == 0) {
            stage[sp] = 1;
            if (child_refl >= 0 && sp + 1 < RT_MAX_RECURSE) { sp++; node[sp] = child_refl; stage[sp] = 0; continue; }
            ret = mk3(0.0f, 0.0f, 0.0f); chain = 0;
         }
         if (stage[sp] == 1) {
            rx[sp] = ret.x; ry[sp] = ret.y; rz[sp] = ret.z; chain_refl[sp] = chain;
            stage[sp] = 2;
            if (child_refr >= 0 && sp + 1 < RT_MAX_RECURSE) { sp++; node[sp] = child_refr; stage[sp] = 0; continue; }
            ret = mk3(0.0f, 0.0f, 0.0f); chain = 0;
         }
         long long k = node[sp];
         F3 total = mk3(totals[3 * k], totals[3 * k + 1], totals[3 * k + 2]);
         F3 refl = mk3(rx[sp], ry[sp], rz[sp]);
         ret = total * (1.0f - rr.x - rr.y) + rr.x * refl + rr.y * ret;
         if (dur) chain = dur[k] + (chain_refl[sp] > chain ? chain_refl[sp] : chain);
         sp--;
      }
      /* rayTrace's clamp (kernel.cu:537-539); NaN passes through clamp unchanged (Util.h:28-30) */
      if (live) { rgb[3 * i] = clamp01(ret.x); rgb[3 * i + 1] = clamp01(ret.y); rgb[3 * i + 2] = clamp01(ret.z); }
      if (tile_cost) {
         /* What this tile costs, for the order in which the next frame starts its tiles: how long its longest chain of
          * dependent rays was (in traversal steps: a figure that does not depend on how the waves were scheduled, so a
          * still camera gets the same order frame after frame).  A tile started late must be one whose chains are short, or the launch ends on a
          * few waves finishing them.  A wave's 64 primaries lie in one tile (a tile's sample count is a multiple of 64). */
         unsigned cost = live ? chain : 0u;
         unsigned sum = cost;
         for (int o = 32; o > 0; o >>= 1) { unsigned other = __shfl_xor(cost, o); cost = other > cost ? other : cost; sum += __shfl_xor(sum, o); }
         if ((threadIdx.x & 63) == 0 && live && cost) {
            atomicMax(&tile_cost[2 * (unsigned)(i / per_tile)], cost);
            atomicAdd(&tile_cost[2 * (unsigned)(i / per_tile) + 1], sum);
         }
      }
   }
}

__device__ __forceinline__ unsigned char to_byte(float v)
{
   /* float -> unsigned char truncates (kernel.cu:568); NaN becomes 0 */
   if (!(v == v)) return 0;
   int k = (int)v;
   return (unsigned char)(k < 0 ? 0 : k > 255 ? 255 : k);
}

/* resolve: averageBufferColors, kernel.cu:545-570.  Thread = pixel of a batch tile. */
/* What k_light and k_combine_tree leave to this kernel: the tiles none of whose primaries spawned a bounce ray
 * (`tile_bounce` clear).  Every tree there is a single node, so a sample's colour is its own light sum times
 * (1 - refl - alpha) -- the same functions and the same expression as in those kernels, evaluated here, where the
 * samples are averaged anyway: such a tile's hits and spawn records are read once and nothing is written in between. */
struct ResolveShade {
   DScene sc;
   const QRay *queue; const Hit *hits; const SpawnRec *recs; const unsigned *flag_words; unsigned flag_stride; int shading;
   const unsigned *tile_bounce;     /* null: every tile went through k_light and k_combine_tree */
   const unsigned *tile_hit;        /* with tile_bounce: clear for a tile none of whose primaries hit anything (nothing of it is read) */
   const unsigned *dur; unsigned *tile_cost;
};
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(DGrid g, const float *rgb, int n_batch_tiles, int out_layout,
                                                      uchar4 *out, ResolveShade rs)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   const int T = RT_TILE;
   if (i >= (long long)n_batch_tiles * T * T) return;
   int k = (int)(i / (T * T));
   int r = (int)(i % (T * T));
   int lx = r % T, ly = r / T;
   int tile = g.tile_first + (g.k0 + k) * g.tile_stride;
   int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
   int x = tx * T + lx, y = ty * T + ly;
   const bool here = rs.tile_bounce && rs.tile_bounce[k] == 0u;     /* uniform over the workgroup: 256 threads are a quarter of a tile */
   const bool inside = x < g.width && y < g.height;
   if (!inside && !here) return;
   const bool empty = here && rs.tile_hit && rs.tile_hit[k] == 0u;       /* every sample of the tile is a miss: colour +0, cost 0 */
   F3 end = mk3(0.0f, 0.0f, 0.0f);
   long long tile_base = (long long)k * g.ts * g.ts;
   unsigned cost = 0, sum = 0;
   for (int xo = 0; xo < g.sroot && !empty; xo++) {
      for (int yo = 0; yo < g.sroot; yo++) {
         int sx = lx * g.sroot + xo, sy = ly * g.sroot + yo;
         long long slot = tile_base + ((long long)((sy >> 3) * g.patches_per_row + (sx >> 3)) << 6) + ((sy & 7) << 3) + (sx & 7);
         if (here) {
            if (inside) {
               const Hit h = rs.hits[slot];
               if (h.prim >= 0) {
                  /* (a sample that hit nothing has light sum 0 and weights 0, 0: its colour is +0 and adds nothing) */
                  float4 rr = *reinterpret_cast<const float4 *>(rs.recs + slot);
                  F3 total = light_sum(rs.sc, rs.queue, slot, h, rs.flag_words, rs.flag_stride, rs.shading);
                  const F3 none = mk3(0.0f, 0.0f, 0.0f);
                  F3 ret = total * (1.0f - rr.x - rr.y) + rr.x * none + rr.y * none;       /* kernel.cu:375-376 without children */
                  end = end + mk3(clamp01(ret.x), clamp01(ret.y), clamp01(ret.z));          /* kernel.cu:537-539 */
                  if (rs.dur) { unsigned d = rs.dur[slot]; cost = d > cost ? d : cost; sum += d; }
               }
            }
         } else end = end + mk3(rgb[3 * slot], rgb[3 * slot + 1], rgb[3 * slot + 2]);
      }
   }
   if (here && rs.tile_cost) {
      /* the tile's cost for the next frame's start order, as k_combine_tree records it for the other tiles */
      for (int o = 32; o > 0; o >>= 1) { unsigned other = __shfl_xor(cost, o); cost = other > cost ? other : cost; sum += __shfl_xor(sum, o); }
      if ((threadIdx.x & 63) == 0 && cost) {
         atomicMax(&rs.tile_cost[2 * (unsigned)k], cost);
         atomicAdd(&rs.tile_cost[2 * (unsigned)k + 1], sum);
      }
   }
   if (!inside) return;
   float div = (float)(g.sroot * g.sroot);
   end = mk3(end.x / div, end.y / div, end.z / div);
   end = end * 255.0f;
   uchar4 px; px.x = to_byte(end.x); px.y = to_byte(end.y); px.z = to_byte(end.z); px.w = 255;
   if (out_layout == RT_OUT_TILES) out[(long long)(g.k0 + k) * T * T + ly * T + lx] = px;
   else out[(long long)y * g.width + x] = px;
}

/* un-tile: parts made with tile_first = p, tile_stride = n_parts -> row-major frame */
__global__ __launch_bounds__(RT_BLOCK) void k_untile(const uchar4 *tiles, long long part_stride_px, int n_parts,
                                                     int width, int height, int tiles_x, uchar4 *frame)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= (long long)width * height) return;
   int x = (int)(i % width), y = (int)(i / width);
   int tile = (y / RT_TILE) * tiles_x + (x / RT_TILE);
   int part = tile % n_parts, k = tile / n_parts;
   frame[i] = tiles[part * part_stride_px + (long long)k * RT_TILE * RT_TILE + (y % RT_TILE) * RT_TILE + (x % RT_TILE)];
}

/* scatter per-slot values to the row-major sample grid (parity views) */
__global__ __launch_bounds__(RT_BLOCK) void k_unslot_hits(DGrid g, const Hit *hits, long long n, const int *prim_ids, int n_prims,
                                                          int *prim, float *t)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= n) return;
   int X, Y;
   if (!slot_to_sample(g, i, X, Y)) return;
   long long idx = (long long)Y * g.res_w + X;
   int p = hits[i].prim;
   prim[idx] = (p >= 0 && p < n_prims) ? prim_ids[p] : p;      /* the caller's numbering; planes follow the BVH primitives */
   t[idx] = hits[i].t;
}
__global__ __launch_bounds__(RT_BLOCK) void k_unslot_rgb(DGrid g, const float *rgb, long long n, float *out)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= n) return;
   int X, Y;
   if (!slot_to_sample(g, i, X, Y)) return;
   long long idx = (long long)Y * g.res_w + X;
   out[3 * idx] = rgb[3 * i]; out[3 * idx + 1] = rgb[3 * i + 1]; out[3 * idx + 2] = rgb[3 * i + 2];
}

/* ======================================================================= */
/* host side                                                                */
/* ======================================================================= */
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

struct rt_scene {
   rt_flat_scene host;
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
   unsigned long long *h_stats;  /* pinned */
   float node_factor;       /* queue entries per primary ray the chain kernel gets room for; grows when a frame overflows */
   StackEntry *d_spill;     /* trees deeper than RT_LDS_STACK: the stack entries beyond LDS, (depth - RT_LDS_STACK) x lanes of a full launch */
   void *d_frame; size_t frame_cap;   /* rt_render's device copy of the frame, kept from call to call */
   uint2 *d_lists;          /* the waves' lists of spawned rays (k_trace): list_waves rings of list_cap entries */
   unsigned list_cap, list_waves;
   /* Tiles are started costliest first, by the previous frame's depth of the bounce trees per tile: the waves that
    * pick up the last primaries of a launch then hold short chains, and the launch does not end on a few long ones. */
   unsigned *d_tile_cost, *h_tile_cost;   /* h_: pinned */
   int *d_tile_order, *h_tile_order;      /* h_: pinned */
   long long order_cap;                   /* tiles the four arrays have room for */
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

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static unsigned long long sum_stat(const unsigned long long *h, int which)
{
   unsigned long long t = 0;
   for (int r = 0; r < RT_STAT_REPL; r++) t += h[which * RT_STAT_REPL + r];
   return t;
}

/* entries of a wave's spawn list to begin with; RT_LIST_CAP overrides it (rounded up to a power of two, at least 64):
 * small values make frames overflow and exercise the retry */
static unsigned first_list_cap()
{
   const char *e = getenv("RT_LIST_CAP");
   long v = e ? atol(e) : 0;
   if (v <= 0) return RT_LIST_CAP_FIRST;
   unsigned c = 64;
   while (c < (unsigned long)v && c < RT_LIST_CAP_MAX) c *= 2;
   return c;
}

extern "C" int rt_device_count(void)
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return 0;
   return n;
}

extern "C" void rt_scene_destroy(rt_scene *s)
{
   if (!s) return;
   if (s->device >= 0) (void)hipSetDevice(s->device);
   if (s->dev_blob) (void)hipFree(s->dev_blob);
   if (s->arena.base) (void)hipFree(s->arena.base);
   if (s->d_ctr) (void)hipFree(s->d_ctr);
   if (s->d_stats) (void)hipFree(s->d_stats);
   if (s->h_stats) (void)hipHostFree(s->h_stats);
   if (s->d_lists) (void)hipFree(s->d_lists);
   if (s->d_frame) (void)hipFree(s->d_frame);
   if (s->d_spill) (void)hipFree(s->d_spill);
   if (s->d_tile_cost) (void)hipFree(s->d_tile_cost);
   if (s->d_tile_order) (void)hipFree(s->d_tile_order);
   if (s->h_tile_cost) (void)hipHostFree(s->h_tile_cost);
   if (s->h_tile_order) (void)hipHostFree(s->h_tile_order);
   for (hipEvent_t e : s->timing_events) (void)hipEventDestroy(e);
   if (s->streams_ok) {
      (void)hipStreamDestroy(s->chain);
      (void)hipEventDestroy(s->ev_begin); (void)hipEventDestroy(s->ev_end);
   }
   rt_flat_scene_free(&s->host);
   delete s;
}

static void scene_tuning(rt_scene *s);
static int env_int(const char *name, int fallback, bool allow_zero);
static int stack_depth(const rt_scene *s);
static int scratch_address(int device, char **out);

/* the box all primitives' bounding boxes lie in (for leaving whole tiles of primaries out of a launch, see cull_tiles) */
static void scene_bounds(rt_scene *s)
{
   const rt_flat_scene &h = s->host;
   s->bounds_ok = h.n_prims > 0 && h.prim_boxes != nullptr && h.bounds_in_range;
   s->n_live = -1; memset(s->cull_key, 0, sizeof s->cull_key); memset(s->cull_cam, 0, sizeof s->cull_cam);
   for (int j = 0; j < 3; j++) { s->bounds_lo[j] = 1e300; s->bounds_hi[j] = -1e300; }
   for (int k = 0; k < h.n_prims && s->bounds_ok; k++)
      for (int j = 0; j < 3; j++) {
         const double lo = h.prim_boxes[6 * (size_t)k + j], hi = h.prim_boxes[6 * (size_t)k + 3 + j];
         if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi)) { s->bounds_ok = false; break; }
         if (lo < s->bounds_lo[j]) s->bounds_lo[j] = lo;
         if (hi > s->bounds_hi[j]) s->bounds_hi[j] = hi;
      }
}

extern "C" int rt_scene_create_host(const rt_scene_desc *desc, rt_scene **out)
{
   if (!desc || !out) { rt_set_error("rt_scene_create_host: null argument"); return RT_ERR_BAD_ARGUMENT; }
   *out = nullptr;
   rt_scene *s = new (std::nothrow) rt_scene();
   if (!s) return RT_ERR_BAD_ARGUMENT;
   memset(&s->host, 0, sizeof s->host);
   memset(&s->d, 0, sizeof s->d);
   s->device = -1; s->flags = 0; s->streams_ok = false; s->ms_build_device = 0.0f;
   s->dev_blob = nullptr; s->dev_bytes = 0; s->d_ctr = nullptr; s->d_stats = nullptr; s->h_stats = nullptr; s->node_factor = 2.0f;
   s->d_tile_cost = s->h_tile_cost = nullptr; s->d_tile_order = s->h_tile_order = nullptr; s->order_cap = 0; s->order_valid = false; s->order_dirty = true;
   s->d_lists = nullptr; s->list_cap = first_list_cap(); s->list_waves = 0; s->d_spill = nullptr; s->d_frame = nullptr; s->frame_cap = 0;
   s->camera = desc->camera; s->has_camera = desc->has_camera; s->ms_upload = 0;
   auto t0 = std::chrono::steady_clock::now();
   int st = rt_flatten_scene(desc, &s->host);
   if (st != RT_OK) { delete s; return st; }
   s->ms_build_host = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
   s->cull_enabled = 1;
   scene_bounds(s);
   *out = s;
   return RT_OK;
}

extern "C" int rt_scene_create(const rt_scene_desc *desc, int device, int flags, rt_scene **out)
{
   if (!desc || !out) { rt_set_error("rt_scene_create: null argument"); return RT_ERR_BAD_ARGUMENT; }
   *out = nullptr;
   int ndev = rt_device_count();
   if (ndev <= 0) { rt_set_error("no HIP device: the render path has no CPU fallback"); return RT_ERR_NO_DEVICE; }
   if (device < 0 || device >= ndev) { rt_set_error("device %d out of range (%d devices)", device, ndev); return RT_ERR_BAD_ARGUMENT; }
   HIP_TRY(hipSetDevice(device));

   rt_scene *s = new (std::nothrow) rt_scene();
   if (!s) return RT_ERR_BAD_ARGUMENT;
   memset(&s->host, 0, sizeof s->host);
   s->device = device; s->flags = flags; s->streams_ok = false; s->ms_build_device = 0.0f;
   s->dev_blob = nullptr; s->d_ctr = nullptr; s->d_stats = nullptr; s->h_stats = nullptr; s->node_factor = 2.0f;
   s->d_tile_cost = s->h_tile_cost = nullptr; s->d_tile_order = s->h_tile_order = nullptr; s->order_cap = 0; s->order_valid = false; s->order_dirty = true;
   s->d_lists = nullptr; s->list_cap = first_list_cap(); s->list_waves = 0; s->d_spill = nullptr; s->d_frame = nullptr; s->frame_cap = 0;
   s->camera = desc->camera; s->has_camera = desc->has_camera;
   scene_tuning(s);
   { char *unused = nullptr; (void)scratch_address(device, &unused); }   /* this device's static working memory (rt_trace_rays_static), looked up while the device is current; a failure is reported by that call */

   if (!env_int("RT_MISS_FILTER", 1, true)) flags |= RT_SCENE_NO_FILTER;      /* for A/B runs; frames are the same either way */
   auto t0 = std::chrono::steady_clock::now();
   int st = rt_flatten_scene_ex(desc, flags, &s->host);
   if (st != RT_OK) { delete s; return st; }
   if ((flags & RT_SCENE_GPU_BVH) && s->host.n_prims > 1) {
      rt_bvh_node *nodes = nullptr; int32_t *order = nullptr; int depth = 0; float ms = 0.0f;
      st = rt_lbvh_build(s->host.prim_boxes, s->host.n_prims, &nodes, &order, &depth, &ms);
      if (st != RT_OK) { rt_flat_scene_free(&s->host); delete s; return st; }
      free(s->host.nodes); free(s->host.leaf_order);
      s->host.nodes = nodes; s->host.leaf_order = order;
      s->host.n_nodes = s->host.n_prims - 1; s->host.tree_depth = depth; s->host.root = 0;
      s->ms_build_device = ms;
      for (int k = 0; k < s->host.n_nodes; k++)
         for (int j = 0; j < 12; j++)
            if (!(std::fabs(s->host.nodes[k].lmin[j]) <= 1.099511627776e12f)) s->host.bounds_in_range = 0;
   }
   auto t1 = std::chrono::steady_clock::now();
   s->ms_build_host = std::chrono::duration<float, std::milli>(t1 - t0).count();

   const rt_flat_scene &h = s->host;
   /* On the device the primitives are numbered by their place among the tree's leaves (formBVH's sorted order), not
    * by the caller's order: a leaf reference is then its position, which a tree with computed child slots needs, and
    * neighbouring leaves have neighbouring records.  prim_ids maps back wherever an id leaves the device. */
   const size_t np_ = (size_t)(h.n_prims > 0 ? h.n_prims : 0);
   std::vector<int32_t> pos_of(np_, -1), ids(np_);
   bool perm_ok = h.leaf_order != nullptr;
   for (size_t k = 0; k < np_ && perm_ok; k++) {
      int32_t pr = h.leaf_order[k];
      if (pr < 0 || (size_t)pr >= np_ || pos_of[(size_t)pr] >= 0) perm_ok = false; else pos_of[(size_t)pr] = (int32_t)k;
   }
   for (size_t k = 0; k < np_; k++) { if (!perm_ok) pos_of[k] = (int32_t)k; ids[k] = perm_ok ? h.leaf_order[k] : (int32_t)k; }
   std::vector<rt_dev_prim> d_prims(np_); std::vector<rt_dev_xform> d_xforms(np_); std::vector<rt_dev_shade> d_shades(np_);
   for (size_t k = 0; k < np_; k++) { d_prims[k] = h.prims[ids[k]]; d_xforms[k] = h.xforms[ids[k]]; d_shades[k] = h.shades[ids[k]]; }
   /* one node array: the tree the rays are answered from, then (if the scene has one) the miss filter's tree, its
    * internal child references moved behind the first tree's records */
   const size_t n_tree = (size_t)(h.n_nodes > 0 ? h.n_nodes : 0), n_filter = (size_t)(h.n_filter_nodes > 0 ? h.n_filter_nodes : 0);
   std::vector<rt_bvh_node> d_nodes(n_tree + n_filter);
   for (size_t k = 0; k < d_nodes.size(); k++) {
      const bool in_filter = k >= n_tree;
      d_nodes[k] = in_filter ? h.filter_nodes[k - n_tree] : h.nodes[k];
      int32_t *child[2] = {&d_nodes[k].left, &d_nodes[k].right};
      for (int c = 0; c < 2; c++) {
         if (*child[c] < 0) *child[c] = ~pos_of[(size_t)~*child[c]];
         else if (in_filter) *child[c] += (int32_t)n_tree;
      }
   }
   s->prim_ids_host = ids;
   /* where each leaf hangs in the reference's tree, for walking the path from the root down to one leaf */
   std::vector<int32_t> leaf_parent(n_filter > 0 ? np_ : 0, 0);
   if (n_filter > 0)
      for (size_t k = 0; k < n_tree; k++) {
         if (d_nodes[k].left < 0) leaf_parent[(size_t)~d_nodes[k].left] = (int32_t)k;
         if (d_nodes[k].right < 0) leaf_parent[(size_t)~d_nodes[k].right] = (int32_t)k;
      }
   enum { N_ARR = 11 };
   size_t off[N_ARR + 1], sizes[N_ARR] = {
      sizeof(rt_bvh_node) * d_nodes.size(), sizeof(rt_dev_prim) * (size_t)h.n_prims,
      sizeof(rt_dev_xform) * (size_t)h.n_prims, sizeof(rt_dev_shade) * (size_t)h.n_prims,
      sizeof(rt_dev_plane) * (size_t)h.n_planes, sizeof(rt_material) * (size_t)h.n_materials,
      sizeof(rt_dev_light) * (size_t)h.n_lights, sizeof(rt_dev_texinfo) * (size_t)h.n_textures, 4 * h.n_texels,
      sizeof(int32_t) * np_, sizeof(int32_t) * leaf_parent.size()};
   off[0] = 0;
   for (int k = 0; k < N_ARR; k++) off[k + 1] = off[k] + align_up(sizes[k] ? sizes[k] : 64, 256);
   s->dev_bytes = off[N_ARR];
   hipError_t e = hipMalloc(&s->dev_blob, s->dev_bytes);
   if (e != hipSuccess) { rt_set_error("hipMalloc(%zu) failed: %s", s->dev_bytes, hipGetErrorString(e)); rt_scene_destroy(s); return RT_ERR_HIP; }
   const void *src[N_ARR] = {d_nodes.data(), d_prims.data(), d_xforms.data(), d_shades.data(), h.planes, h.materials, h.lights, h.tex_info,
                             h.texels, ids.data(), leaf_parent.data()};
   for (int k = 0; k < N_ARR; k++) {
      if (!sizes[k]) continue;
      e = hipMemcpy((char *)s->dev_blob + off[k], src[k], sizes[k], hipMemcpyHostToDevice);
      if (e != hipSuccess) { rt_set_error("scene upload failed: %s", hipGetErrorString(e)); rt_scene_destroy(s); return RT_ERR_HIP; }
   }
   char *b = (char *)s->dev_blob;
   s->d.nodes = (const rt_bvh_node *)(b + off[0]); s->d.prims = (const rt_dev_prim *)(b + off[1]);
   s->d.xforms = (const rt_dev_xform *)(b + off[2]); s->d.shades = (const rt_dev_shade *)(b + off[3]);
   s->d.planes = (const rt_dev_plane *)(b + off[4]); s->d.materials = (const rt_material *)(b + off[5]);
   s->d.lights = (const rt_dev_light *)(b + off[6]);
   s->d.tex_info = (const rt_dev_texinfo *)(b + off[7]); s->d.texels = (const uchar4 *)(b + off[8]);
   s->d.prim_ids = (const int *)(b + off[9]); s->d.leaf_parent = (const int *)(b + off[10]);
   s->d.n_prims = h.n_prims; s->d.n_planes = h.n_planes; s->d.n_lights = h.n_lights;
   s->d.root = (h.root < 0 && h.root != RT_ROOT_EMPTY) ? ~pos_of[(size_t)~h.root] : h.root;
   s->d.fast_ok = h.bounds_in_range;
   s->d.filter_root = (n_filter > 0 && h.root >= 0) ? (int)n_tree : RT_ROOT_EMPTY;
   s->d.filter_kinds = 3;
   {
      /* the first guess, for the frames before the try-out has spoken (a cold launch_kernel call renders only one):
       * a scene made mostly of glass does better without the second tree */
      long long glass = 0;
      for (int k = 0; k < h.n_prims; k++) if (h.materials[h.prims[k].material].refr > 0.0f) glass++;
      if (flags & RT_SCENE_FILTER_ALWAYS) { s->filter_auto = 0; s->filter_choice = 3; }
      if (s->filter_auto && h.n_prims > 0 && glass * 4 > (long long)h.n_prims) s->filter_choice = 0;
   }
   s->d.any_xform = h.any_xform;
   {
      /* every box of the reference's tree inside its parent's?  (formBVH makes them so: combineBoundingBox,
       * BoundingBox.h:48-52, takes the smaller and the larger bound and pads; a scene with a NaN bound fails here) */
      bool nested = n_tree > 0 && n_filter > 0;
      for (size_t k = 0; k < n_tree && nested; k++) {
         const rt_bvh_node &n = d_nodes[k];
         const int32_t child[2] = {n.left, n.right};
         const float *box[2] = {n.lmin, n.rmin};       /* lmin, lmax / rmin, rmax: six consecutive floats each */
         for (int c = 0; c < 2 && nested; c++) {
            if (child[c] < 0) continue;
            const rt_bvh_node &m = d_nodes[(size_t)child[c]];
            const float *sub[2] = {m.lmin, m.rmin};
            for (int b = 0; b < 2; b++)
               for (int j = 0; j < 3; j++)
                  if (!(sub[b][j] >= box[c][j]) || !(sub[b][3 + j] <= box[c][3 + j])) nested = false;
         }
      }
      s->d.verify_jump = nested && env_int("RT_VERIFY_JUMP", 1, true) ? 1 : 0;
   }
   s->d.sphere_coeff = h.sphere_coeff; s->d.origin_coeff = 9.5367431640625e-7f;
   /* (for measurements: what the two margins cost) */
   if (const char *e = getenv("RT_SPHERE_MARGIN")) s->d.sphere_coeff *= (float)atof(e);
   if (const char *e = getenv("RT_ORIGIN_MARGIN")) s->d.origin_coeff *= (float)atof(e);
   /* a bounce ray needs a material with reflection or refraction above 0 (kernel.cu:360-374) */
   s->can_bounce = false;
   for (int k = 0; k < h.n_materials; k++) if (h.materials[k].refl > 0.0f || h.materials[k].refr > 0.0f) s->can_bounce = true;
   s->lights_finite = true;
   for (int k = 0; k < h.n_lights; k++) for (int j = 0; j < 3; j++) if (!std::isfinite(h.lights[k].clr[j])) s->lights_finite = false;
   scene_bounds(s);
   if (h.sphere_coeff > 0.0f) {
      /* the box that holds the spheres (other primitives do not count: a floor of two large triangles would put every
       * ray far from "the scene") */
      double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, r2 = 0.0;
      bool ok = s->bounds_ok && h.prim_boxes != nullptr;
      for (int k = 0; k < h.n_prims && ok; k++) {
         if (h.prims[k].type != RT_PRIM_SPHERE) continue;
         for (int j = 0; j < 3; j++) {
            lo[j] = std::min(lo[j], (double)h.prim_boxes[6 * (size_t)k + j]); hi[j] = std::max(hi[j], (double)h.prim_boxes[6 * (size_t)k + 3 + j]);
         }
      }
      for (int j = 0; j < 3 && ok; j++) { ok = lo[j] <= hi[j]; const double half = 0.5 * (hi[j] - lo[j]); r2 += half * half; }
      if (ok) {
         s->d.mid_x = (float)(0.5 * (lo[0] + hi[0])); s->d.mid_y = (float)(0.5 * (lo[1] + hi[1])); s->d.mid_z = (float)(0.5 * (lo[2] + hi[2]));
         s->d.mid_rad = (float)std::sqrt(r2);
      } else s->d.filter_root = RT_ROOT_EMPTY;      /* no usable bounds to measure a ray's distance from the spheres by: such a scene goes without the second tree */
   }

   if (hipMalloc((void **)&s->d_ctr, RT_CTR_WORDS * sizeof(unsigned long long)) != hipSuccess ||
       hipMalloc((void **)&s->d_stats, ST_COUNT * RT_STAT_REPL * sizeof(unsigned long long)) != hipSuccess ||
       hipHostMalloc((void **)&s->h_stats, ST_COUNT * RT_STAT_REPL * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
      rt_set_error("allocation of queue counters failed"); rt_scene_destroy(s); return RT_ERR_HIP;
   }
   if (stack_depth(s) > RT_LDS_STACK) {
      const size_t bytes = sizeof(StackEntry) * (size_t)(stack_depth(s) - RT_LDS_STACK) * (size_t)s->trace_blocks * RT_BLOCK;
      if (hipMalloc((void **)&s->d_spill, bytes) != hipSuccess) { rt_set_error("allocation of %zu bytes of traversal stack failed", bytes); rt_scene_destroy(s); return RT_ERR_HIP; }
   }
   {
      bool ok = hipStreamCreateWithFlags(&s->chain, hipStreamNonBlocking) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&s->ev_begin, hipEventDisableTiming) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&s->ev_end, hipEventDisableTiming) == hipSuccess;
      if (!ok) { rt_set_error("stream / event creation failed"); rt_scene_destroy(s); return RT_ERR_HIP; }
      s->streams_ok = true;
   }
   e = hipDeviceSynchronize();
   if (e != hipSuccess) { rt_set_error("scene upload failed: %s", hipGetErrorString(e)); rt_scene_destroy(s); return RT_ERR_HIP; }
   auto t2 = std::chrono::steady_clock::now();
   s->ms_upload = std::chrono::duration<float, std::milli>(t2 - t1).count();
   *out = s;
   return RT_OK;
}

extern "C" int rt_scene_get_info(const rt_scene *s, rt_scene_info *info)
{
   if (!s || !info) { rt_set_error("rt_scene_get_info: null argument"); return RT_ERR_BAD_ARGUMENT; }
   info->n_bvh_prims = s->host.n_prims; info->n_planes = s->host.n_planes; info->n_lights = s->host.n_lights;
   info->n_nodes = s->host.n_nodes; info->n_materials = s->host.n_materials; info->tree_depth = s->host.tree_depth;
   info->root = s->host.root; info->device_bytes = s->dev_bytes;
   info->ms_build_host = s->ms_build_host; info->ms_upload = s->ms_upload; info->ms_build_device = s->ms_build_device;
   info->filter_depth = s->host.n_filter_nodes > 0 ? s->host.filter_depth : 0;
   return RT_OK;
}

extern "C" int rt_scene_get_bvh(const rt_scene *s, const rt_bvh_node **nodes, const int32_t **leaf_order)
{
   if (!s) { rt_set_error("rt_scene_get_bvh: null scene"); return RT_ERR_BAD_ARGUMENT; }
   if (nodes) *nodes = s->host.nodes;
   if (leaf_order) *leaf_order = s->host.leaf_order;
   return RT_OK;
}

extern "C" int rt_scene_set_camera(rt_scene *s, const rt_camera *cam)
{
   if (!s || !cam) { rt_set_error("rt_scene_set_camera: null argument"); return RT_ERR_BAD_ARGUMENT; }
   s->camera = *cam; s->has_camera = 1;
   return RT_OK;
}

static int isqrt_trunc(int n) { return (int)std::sqrt((double)n); }   /* kernel.cu:707 */

extern "C" int rt_tile_count(int width, int height, int tile_first, int tile_stride)
{
   if (width <= 0 || height <= 0) return 0;
   if (tile_stride <= 0) tile_stride = 1;
   int tiles = ((width + RT_TILE - 1) / RT_TILE) * ((height + RT_TILE - 1) / RT_TILE);
   if (tile_first < 0 || tile_first >= tiles) return 0;
   return (tiles - tile_first + tile_stride - 1) / tile_stride;
}

static DCamera make_camera(const rt_camera &c)
{
   /* normalize(lookAt - pos) on the host, in the same order the device uses (kernel.cu:717-718) */
   DCamera d;
   d.pos.x = c.pos[0]; d.pos.y = c.pos[1]; d.pos.z = c.pos[2];
   d.up.x = c.up[0]; d.up.y = c.up[1]; d.up.z = c.up[2];
   d.right.x = c.right[0]; d.right.y = c.right[1]; d.right.z = c.right[2];
   float lx = c.look_at[0] - c.pos[0], ly = c.look_at[1] - c.pos[1], lz = c.look_at[2] - c.pos[2];
   float sqr = lx * lx + ly * ly + lz * lz;
   float inv = 1.0f / std::sqrt(sqr);
   d.look_dir.x = lx * inv; d.look_dir.y = ly * inv; d.look_dir.z = lz * inv;
   return d;
}

static int arena_reserve(Arena &a, size_t bytes)
{
   if (bytes <= a.cap) return RT_OK;
   if (a.base) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(a.base)); a.base = nullptr; a.cap = 0; }
   size_t want = align_up(bytes + bytes / 4, 1 << 20);
   hipError_t e = hipMalloc((void **)&a.base, want);
   if (e != hipSuccess) { rt_set_error("hipMalloc(%zu) for ray queues failed: %s", want, hipGetErrorString(e)); return RT_ERR_HIP; }
   a.cap = want;
   return RT_OK;
}

struct Timer {
   hipEvent_t a, b;
   bool ok = false, ended = false;
   float *acc = nullptr;
};

/* timing events of one frame; the events themselves are kept by the scene from frame to frame */
struct EventPool {
   std::vector<hipEvent_t> &spare;
   std::deque<Timer> timers;   /* a deque: begin() hands out references that must survive later begin()s */
   explicit EventPool(std::vector<hipEvent_t> &cache) : spare(cache) {}
   ~EventPool()
   {
      for (auto &t : timers) if (t.ok) { spare.push_back(t.a); spare.push_back(t.b); }
   }
   bool take(hipEvent_t *e)
   {
      if (!spare.empty()) { *e = spare.back(); spare.pop_back(); return true; }
      return hipEventCreate(e) == hipSuccess;
   }
   /* `wanted` false: a timer that records nothing (an event between two kernels of a stream costs a few microseconds of
    * the stream's time: only the runs that ask for the detail pay for it) */
   Timer &begin(hipStream_t st, float *acc, bool wanted = true)
   {
      Timer t; t.acc = acc;
      if (!wanted) { timers.push_back(t); return timers.back(); }
      t.ok = take(&t.a);
      if (t.ok && !take(&t.b)) { spare.push_back(t.a); t.ok = false; }
      if (t.ok) (void)hipEventRecord(t.a, st);
      timers.push_back(t);
      return timers.back();
   }
   void end(Timer &t, hipStream_t st) { if (t.ok) { (void)hipEventRecord(t.b, st); t.ended = true; } }
   void collect()
   {
      for (auto &t : timers) {
         float ms = 0;
         if (t.ok && t.ended && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess && t.acc) *t.acc += ms;
      }
   }
};

enum { VIEW_FRAME = 0, VIEW_PRIMARY = 1, VIEW_SAMPLES = 2 };

static unsigned grid_for(long long n) { return (unsigned)((n + RT_BLOCK - 1) / RT_BLOCK); }

static long long max_batch_slots()
{
   const char *e = getenv("RT_MAX_BATCH_SLOTS");
   long long v = e ? atoll(e) : 0;
   return v > 0 ? v : (long long)(1 << 26);
}

/* entries the deepest walk can defer: one per internal level of the deeper of the scene's trees */
static int stack_depth(const rt_scene *s)
{
   int depth = s->host.tree_depth > s->host.filter_depth ? s->host.tree_depth : s->host.filter_depth;
   return depth > 0 ? depth : 1;
}

static int trace_lds_bytes(const rt_scene *s, size_t *out)
{
   /* the stack holds at most one deferred sibling per internal level; RT_LDS_STACK of them live in LDS */
   int depth = stack_depth(s);
   if (depth > RT_LDS_STACK) depth = RT_LDS_STACK;
   *out = sizeof(StackEntry) * (size_t)RT_BLOCK * (size_t)depth;
   return RT_OK;
}

static int env_int(const char *name, int fallback, bool allow_zero)
{
   const char *e = getenv(name);
   if (!e) return fallback;
   int v = atoi(e);
   return (v > 0 || (allow_zero && v == 0)) ? v : fallback;
}

/* The persistent traversal launch fills the device once: 4 workgroups of 256 per CU are what 128 registers per
 * lane allow, and every wave pulls rays until the queue is dry. */
static void scene_tuning(rt_scene *s)
{
   int cus = 256;
   hipDeviceProp_t prop;
   if (hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
   s->trace_blocks = env_int("RT_TRACE_BLOCKS", cus * 4, false);
   s->refill_idle = env_int("RT_REFILL_IDLE", RT_REFILL_IDLE, false);
   s->node_burst = env_int("RT_NODE_BURST", RT_NODE_BURST, false);
   s->leaf_trigger = env_int("RT_LEAF_TRIGGER", RT_LEAF_TRIGGER, false);
   s->use_tile_order = env_int("RT_TILE_ORDER", 1, true);
   s->order_classes = env_int("RT_ORDER_CLASSES", RT_ORDER_CLASSES, false);
   s->order_mean_weight = (double)env_int("RT_ORDER_MEAN_WEIGHT", RT_ORDER_MEAN_WEIGHT, true);
   s->fuse_resolve = env_int("RT_FUSE_RESOLVE", 1, true);
   s->cull_enabled = env_int("RT_CULL_TILES", 1, true);
   s->skip_unlit = env_int("RT_SKIP_UNLIT", 16, true);
   s->own_stream = env_int("RT_OWN_STREAM", 0, true);
   {
      const char *e = getenv("RT_MISS_FILTER");
      s->filter_auto = !(e && *e && atoi(e) == 1 && e[0] == '1');      /* "1": always; unset / "auto": try out ("0" builds no second tree at all) */
      s->filter_choice = 3; s->probe_step = 0;
      memset(s->probe_key, 0, sizeof s->probe_key);
   }
   int r = env_int("RT_RESERVE", RT_RESERVE, false);
   s->reserve = (unsigned)(r < 128 ? 128 : r);      /* one spawn stage can ask for 128 entries */
}

/* room for the lists of every wave a launch can have */
static int ensure_lists(rt_scene *s)
{
   const unsigned waves = (unsigned)s->trace_blocks * (RT_BLOCK / 64);
   if (s->d_lists && s->list_waves == waves) return RT_OK;
   if (s->d_lists) { (void)hipFree(s->d_lists); s->d_lists = nullptr; }
   hipError_t e = hipMalloc((void **)&s->d_lists, sizeof(uint2) * (size_t)waves * s->list_cap);
   if (e != hipSuccess) { rt_set_error("hipMalloc of the spawn lists (%u x %u entries) failed: %s", waves, s->list_cap, hipGetErrorString(e)); return RT_ERR_HIP; }
   s->list_waves = waves;
   return RT_OK;
}

static void launch_trace(const rt_scene *s, int out_mode, bool count, TraceArgs a, unsigned long long n_hint, size_t lds_bytes,
                         hipStream_t stream, int filter_kinds = 3)
{
   DScene d = s->d;
   d.filter_kinds = filter_kinds;
   if (!filter_kinds) d.filter_root = RT_ROOT_EMPTY;
   unsigned long long want = (n_hint + RT_BLOCK - 1) / RT_BLOCK;
   unsigned blocks = (unsigned)(want < (unsigned long long)s->trace_blocks ? want : (unsigned long long)s->trace_blocks);
   if (blocks == 0) return;
   a.refill_idle = s->refill_idle; a.node_burst = s->node_burst; a.leaf_trigger = s->leaf_trigger;
   a.reserve = s->reserve;
   if (!a.stats) a.stats = s->d_stats;
   a.spill = s->d_spill;
   const bool spill = stack_depth(s) > RT_LDS_STACK;
   const bool two = d.filter_root != RT_ROOT_EMPTY;
   TraceKernArgs ka; ka.sc = d; ka.a = a;
#define RT_LAUNCH_TRACE2(O, C, S) do { if (two) hipLaunchKernelGGL((k_trace<O, C, S, true>), dim3(blocks), dim3(RT_BLOCK), lds_bytes, stream, ka); \
                                       else hipLaunchKernelGGL((k_trace<O, C, S, false>), dim3(blocks), dim3(RT_BLOCK), lds_bytes, stream, ka); } while (0)
#define RT_LAUNCH_TRACE(O, C) do { if (spill) RT_LAUNCH_TRACE2(O, C, true); else RT_LAUNCH_TRACE2(O, C, false); } while (0)
   if (out_mode == OUT_CHAIN) { if (count) RT_LAUNCH_TRACE(OUT_CHAIN, true); else RT_LAUNCH_TRACE(OUT_CHAIN, false); }
   else { if (count) RT_LAUNCH_TRACE(OUT_HIT, true); else RT_LAUNCH_TRACE(OUT_HIT, false); }
#undef RT_LAUNCH_TRACE
#undef RT_LAUNCH_TRACE2
}

/* device counters of the chunk in flight (rt_scene::d_ctr) */
struct ChunkCtr {
   unsigned long long *next_chain;
   unsigned *tail;
};
static ChunkCtr chunk_ctr(const rt_scene *s, int ring)
{
   (void)ring;
   ChunkCtr c;
   c.next_chain = s->d_ctr;
   c.tail = reinterpret_cast<unsigned *>(s->d_ctr + 8 * RT_FETCH_STREAMS);
   return c;
}

/* Which of the selected tiles have to be traced at all.
 *
 * Every primary ray lies on a line through E = pos - look_dir: sample (u, v) starts at pos + u/2 right + v/2 up and
 * runs along u/2 right + v/2 up + look_dir (generateCameraRays, kernel.cu:477-498).  The rays of a 32x32-pixel tile
 * therefore lie inside a pyramid with apex E, spanned by the corner directions of the tile's (u, v) rectangle.  If the
 * box that holds every primitive's bounding box lies wholly beyond one of the pyramid's four side planes (or wholly
 * behind its apex), no ray of the tile enters any box of the tree, the reference's walk (kernel.cu:136-158) ends at
 * the root and the sample is the background: nothing of such a tile is made, traced, stored or read.
 *
 * The margins are large against the arithmetic they stand in for: the tile's rectangle is widened by a whole sample
 * spacing and by a thousandth of the image on every side (jitter moves a sample by half a spacing; the float
 * arithmetic of the ray set-up and of the box tests is good to ~1e-6 of the distances involved), the box by a
 * thousandth of its size and of its distance from E plus the padding the tree's levels add to their boxes
 * (BoundingBox.h: 1e-4 per construction).  Computed in double.  Scenes with planes are not culled (planes are
 * tested for every ray, kernel.cu:189-196). */
static void cull_tiles(const rt_scene *s, const DCamera &cam, const DGrid &g, int n_sel, std::vector<unsigned char> &live, long long *n_live)
{
   live.assign((size_t)n_sel, 1);
   *n_live = n_sel;
   if (!s->bounds_ok) return;
   const double E[3] = {(double)cam.pos.x - cam.look_dir.x, (double)cam.pos.y - cam.look_dir.y, (double)cam.pos.z - cam.look_dir.z};
   const double R[3] = {cam.right.x, cam.right.y, cam.right.z}, U[3] = {cam.up.x, cam.up.y, cam.up.z};
   const double Ld[3] = {cam.look_dir.x, cam.look_dir.y, cam.look_dir.z};
   double lo[3], hi[3], ext = 0.0, dist = 0.0;
   for (int j = 0; j < 3; j++) {
      ext = std::max(ext, s->bounds_hi[j] - s->bounds_lo[j]);
      const double c = 0.5 * (s->bounds_lo[j] + s->bounds_hi[j]) - E[j];
      dist += c * c;
   }
   dist = std::sqrt(dist);
   /* the float arithmetic of the ray set-up (kernel.cu:493-497) rounds a ray's origin and direction by a few units in
    * the last place of the camera position's coordinates: a camera far from the origin of the coordinates gets margins
    * that grow with that */
   const double pos_ulp = std::max(std::max(std::fabs((double)cam.pos.x), std::fabs((double)cam.pos.y)), std::fabs((double)cam.pos.z)) * (1.0 / 4194304.0);
   const double pad = 1e-3 * (ext + dist) + 2e-4 * (double)(s->host.tree_depth + s->host.filter_depth + 4) + 1e-3 + 4.0 * pos_ulp * (2.0 + dist);
   if (!std::isfinite(pad)) return;
   for (int j = 0; j < 3; j++) { lo[j] = s->bounds_lo[j] - pad; hi[j] = s->bounds_hi[j] + pad; }
   double corner[8][3];
   for (int c = 0; c < 8; c++)
      for (int j = 0; j < 3; j++) corner[c][j] = ((c >> j) & 1 ? hi[j] : lo[j]) - E[j];
   const double len_r = std::sqrt(R[0] * R[0] + R[1] * R[1] + R[2] * R[2]), len_u = std::sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
   const double pad_u = 2.0 / (double)g.res_w + 2e-3 + (len_r > 0.0 ? 4.0 * pos_ulp / len_r : 0.0);
   const double pad_v = 2.0 / (double)g.res_h + 2e-3 + (len_u > 0.0 ? 4.0 * pos_ulp / len_u : 0.0);
   if (!std::isfinite(pad_u) || !std::isfinite(pad_v)) return;
   auto dir = [&](double u, double v, double *d) { for (int j = 0; j < 3; j++) d[j] = 0.5 * u * R[j] + 0.5 * v * U[j] + Ld[j]; };
   auto cross = [](const double *a, const double *b, double *c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; };
   auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
   long long kept = 0;
   for (int k = 0; k < n_sel; k++) {
      const long long tile = (long long)g.tile_first + (long long)k * g.tile_stride;
      const long long tx = tile % g.tiles_x, ty = tile / g.tiles_x;
      /* the (u, v) rectangle of the tile's samples: sample X has u = 2 X / res_w - 1 + jitter, |jitter| <= 1/(2 res_w) */
      const double u0 = 2.0 * (double)(tx * g.ts) / (double)g.res_w - 1.0 - pad_u;
      const double u1 = 2.0 * (double)((tx + 1) * g.ts) / (double)g.res_w - 1.0 + pad_u;
      const double v0 = 2.0 * (double)(ty * g.ts) / (double)g.res_h - 1.0 - pad_v;
      const double v1 = 2.0 * (double)((ty + 1) * g.ts) / (double)g.res_h - 1.0 + pad_v;
      double d00[3], d10[3], d01[3], d11[3], mid[3];
      dir(u0, v0, d00); dir(u1, v0, d10); dir(u0, v1, d01); dir(u1, v1, d11); dir(0.5 * (u0 + u1), 0.5 * (v0 + v1), mid);
      /* the four side planes through E, each through two neighbouring corner directions; inside = the side `mid` is on */
      const double *edge[4][2] = {{d00, d10}, {d10, d11}, {d11, d01}, {d01, d00}};
      bool outside = false;
      for (int e = 0; e < 4 && !outside; e++) {
         double nrm[3]; cross(edge[e][0], edge[e][1], nrm);
         const double len = std::sqrt(dot(nrm, nrm)), m = dot(nrm, mid);
         if (!(len > 0.0) || !std::isfinite(len) || !(std::fabs(m) > 1e-9 * len)) continue;      /* a degenerate camera culls nothing */
         const double sign = m > 0.0 ? 1.0 : -1.0;
         bool all_out = true;
         for (int c = 0; c < 8 && all_out; c++) all_out = sign * dot(nrm, corner[c]) < 0.0;
         outside = all_out;
      }
      if (!outside) {
         /* ... or wholly behind the apex: beyond the plane through E across the pyramid's middle direction, which every
          * direction of the pyramid points away from */
         bool ahead = dot(mid, mid) > 0.0;
         const double *cd[4] = {d00, d10, d11, d01};
         for (int e = 0; e < 4 && ahead; e++) ahead = dot(mid, cd[e]) > 0.0;
         if (ahead) {
            bool all_behind = true;
            for (int c = 0; c < 8 && all_behind; c++) all_behind = dot(mid, corner[c]) < 0.0;
            outside = all_behind;
         }
      }
      if (outside) live[(size_t)k] = 0; else kept++;
   }
   *n_live = kept;
}

/* A start order for a frame that has no costs of an earlier one to go by: the selected tiles by their distance, in the
 * picture, from where the middle of the scene's box appears (the middle of the frame if that is behind the camera). */
static void first_order_guess(const rt_scene *s, const DCamera &cam, const DGrid &g, int n_sel, int *order)
{
   double cx = 0.5 * g.width, cy = 0.5 * g.height;
   {
      /* the middle of the box, seen from E = pos - look_dir, as a multiple of u/2 right + v/2 up + look_dir */
      const double E[3] = {(double)cam.pos.x - cam.look_dir.x, (double)cam.pos.y - cam.look_dir.y, (double)cam.pos.z - cam.look_dir.z};
      const double A[3][3] = {{0.5 * cam.right.x, 0.5 * cam.up.x, cam.look_dir.x}, {0.5 * cam.right.y, 0.5 * cam.up.y, cam.look_dir.y},
                              {0.5 * cam.right.z, 0.5 * cam.up.z, cam.look_dir.z}};
      double d[3];
      for (int j = 0; j < 3; j++) d[j] = 0.5 * (s->bounds_lo[j] + s->bounds_hi[j]) - E[j];
      const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                         A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
      if (std::isfinite(det) && std::fabs(det) > 1e-30) {
         auto det_with = [&](int col) {
            double M[3][3];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[r][c] = c == col ? d[r] : A[r][c];
            return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                   M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
         };
         const double a = det_with(0) / det, b = det_with(1) / det, c = det_with(2) / det;
         if (std::isfinite(a) && std::isfinite(b) && c > 1e-9) {
            const double u = a / c, v = b / c;
            if (std::fabs(u) < 4.0 && std::fabs(v) < 4.0) { cx = 0.5 * (u + 1.0) * g.width; cy = 0.5 * (v + 1.0) * g.height; }
         }
      }
   }
   std::vector<std::pair<double, int>> key((size_t)n_sel);
   for (int k = 0; k < n_sel; k++) {
      const long long tile = (long long)g.tile_first + (long long)k * g.tile_stride;
      const double x = ((double)(tile % g.tiles_x) + 0.5) * RT_TILE - cx, y = ((double)(tile / g.tiles_x) + 0.5) * RT_TILE - cy;
      key[(size_t)k] = std::make_pair(x * x + y * y, k);
   }
   std::sort(key.begin(), key.end());
   for (int k = 0; k < n_sel; k++) order[k] = key[(size_t)k].second;
}

extern "C" int rt_tiles_live(const rt_scene *s, const rt_render_params *p, uint8_t *live, int n)
{
   if (!s || !p || !live) { rt_set_error("rt_tiles_live: null argument"); return RT_ERR_BAD_ARGUMENT; }
   if (p->width <= 0 || p->height <= 0 || p->sample_count <= 0) { rt_set_error("rt_tiles_live: non-positive size"); return RT_ERR_BAD_ARGUMENT; }
   const int sroot = isqrt_trunc(p->sample_count);
   if (sroot * sroot != p->sample_count) { rt_set_error("Invalid sample count: %d. Sample count for anti aliasing must have an integer square root", p->sample_count); return RT_ERR_BAD_SAMPLE_COUNT; }
   if (!s->has_camera) { rt_set_error("scene has no camera"); return RT_ERR_NO_CAMERA; }
   const int tile_stride = p->tile_stride > 0 ? p->tile_stride : 1;
   const int n_sel = rt_tile_count(p->width, p->height, p->tile_first, tile_stride);
   if (n < n_sel) { rt_set_error("rt_tiles_live: room for %d tiles, the selection has %d", n, n_sel); return RT_ERR_BAD_ARGUMENT; }
   DGrid g; memset(&g, 0, sizeof g);
   g.width = p->width; g.height = p->height; g.sroot = sroot;
   g.res_w = p->width * sroot; g.res_h = p->height * sroot;
   g.tiles_x = (p->width + RT_TILE - 1) / RT_TILE;
   g.tile_first = p->tile_first; g.tile_stride = tile_stride; g.ts = RT_TILE * sroot;
   std::vector<unsigned char> v((size_t)(n_sel > 0 ? n_sel : 0), 1);
   long long kept = n_sel;
   if (n_sel > 0 && s->cull_enabled && s->host.n_planes == 0 && s->bounds_ok) cull_tiles(s, make_camera(s->camera), g, n_sel, v, &kept);
   for (int k = 0; k < n_sel; k++) live[k] = v[(size_t)k];
   return RT_OK;
}

/* The frame driver: body of launch_kernel after scene set-up (kernel.cu:735-760), over the tiles this call
 * selects.  The tiles are cut into chunks (one, whenever the frame fits); a chunk is four launches on the scene's
 * stream: k_trace (every ray of the chunk), k_light, k_combine_tree, k_resolve.  Nothing is read back while the
 * frame is in flight; the queue capacity per primary ray and the length of the waves' spawn lists are guesses
 * that the device checks, and a frame that overflowed one is rendered again with a larger one. */
static int render_impl(const rt_scene *cs, const rt_render_params *p, int view, void *out_dev, int *prim_dev,
                       float *t_dev, float *rgb_dev, hipStream_t stream, rt_stats *stats_out)
{
   rt_scene *s = const_cast<rt_scene *>(cs);   /* the arena and counters are scratch state of the scene */
#ifdef RT_DEBUG_HOST
   static double dbg_acc[4]; static int dbg_n; static std::chrono::steady_clock::time_point dbg_prev_exit;
   auto dbg_t0 = std::chrono::steady_clock::now(); std::chrono::steady_clock::time_point dbg_t1, dbg_t2;
#endif
   if (!s || !p) { rt_set_error("render: null argument"); return RT_ERR_BAD_ARGUMENT; }
   if (s->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   if (p->width <= 0 || p->height <= 0 || p->sample_count <= 0) { rt_set_error("render: non-positive size"); return RT_ERR_BAD_ARGUMENT; }
   int sroot = isqrt_trunc(p->sample_count);
   if (sroot * sroot != p->sample_count) {
      rt_set_error("Invalid sample count: %d. Sample count for anti aliasing must have an integer square root", p->sample_count);
      return RT_ERR_BAD_SAMPLE_COUNT;
   }
   if (!s->has_camera) { rt_set_error("scene has no camera"); return RT_ERR_NO_CAMERA; }
   if (p->shading != RT_PHONG && p->shading != RT_COOK_TORRANCE) { rt_set_error("Improper shading type specified"); return RT_ERR_BAD_ARGUMENT; }
   int max_rec = p->max_recurse > 0 ? p->max_recurse : RT_MAX_RECURSE;
   if (max_rec > RT_MAX_RECURSE) { rt_set_error("max_recurse > %d", RT_MAX_RECURSE); return RT_ERR_BAD_ARGUMENT; }
   if ((long long)p->width * sroot > 0x3fffffff / (long long)(p->height * (long long)sroot)) { rt_set_error("frame too large"); return RT_ERR_TOO_LARGE; }
   HIP_TRY(hipSetDevice(s->device));
   if (view == VIEW_PRIMARY) max_rec = 1;
   s->last_frame.valid = false;

   int tile_first = p->tile_first, tile_stride = p->tile_stride > 0 ? p->tile_stride : 1;
   int n_sel = rt_tile_count(p->width, p->height, tile_first, tile_stride);
   rt_stats st; memset(&st, 0, sizeof st);
   if (n_sel <= 0) { if (stats_out) *stats_out = st; return RT_OK; }

   DGrid g;
   g.width = p->width; g.height = p->height; g.sroot = sroot;
   g.res_w = p->width * sroot; g.res_h = p->height * sroot;
   g.tiles_x = (p->width + RT_TILE - 1) / RT_TILE;
   g.tile_first = tile_first; g.tile_stride = tile_stride; g.k0 = 0;
   g.ts = RT_TILE * sroot; g.patches_per_row = g.ts / 8;
   g.per_tile = (unsigned)(g.ts * g.ts);
   g.per_tile_div = make_fastdiv(g.per_tile); g.patches_per_row_div = make_fastdiv((unsigned)g.patches_per_row);
   g.tiles_x_div = make_fastdiv((unsigned)g.tiles_x);
   g.inv_res_w = 1.0f / (float)g.res_w; g.inv_res_h = 1.0f / (float)g.res_h;
   g.jitter = p->jitter; g.seed = p->seed;
   DCamera cam = make_camera(s->camera);
   const DScene &sc = s->d;
   const int L = sc.n_lights;
   const bool count = (s->flags & RT_SCENE_COUNTERS) != 0;
   const long long per_tile = (long long)g.ts * g.ts;
   size_t lds_bytes = 0;
   { int rc = trace_lds_bytes(s, &lds_bytes); if (rc != RT_OK) return rc; }
   hipStream_t M = s->own_stream ? s->chain : stream;

   /* which rays start in the second tree this frame: the scene's choice, or the next combination of the try-out */
   int filter_kinds = s->filter_choice;
   const long long probe_key[4] = {p->width, p->height, p->sample_count, (long long)tile_first * 65536 + tile_stride};
   const bool probing = s->filter_auto && view == VIEW_FRAME && sc.filter_root != RT_ROOT_EMPTY;
   if (probing) {
      if (memcmp(probe_key, s->probe_key, sizeof probe_key) != 0) { memcpy(s->probe_key, probe_key, sizeof probe_key); s->probe_step = 0; }
      static const int order[5] = {-1, 3, 1, 2, 0};       /* the first frame of a size is not timed (it has no tile order yet): it runs on the current choice */
      filter_kinds = s->probe_step < 5 && order[s->probe_step] >= 0 ? order[s->probe_step] : s->filter_choice;
   }

   int attempts = 0;
   bool culled_frame = false;      /* the frame left tiles out (s->tile_live says which) */
   long long last_hits_offset = -1;   /* a one-launch frame: where its hits lie in the arena */
   for (int attempt = 0;; attempt++) {
      memset(&st, 0, sizeof st);
      attempts = attempt + 1;
      /* chunk size: a chunk is bounded by its queue entries; a frame is one chunk whenever that allows */
      double F = max_rec > 1 ? (double)s->node_factor : 1.0;
      long long slots_cap = max_batch_slots();
      long long tiles_cap = slots_cap / per_tile; if (tiles_cap < 1) tiles_cap = 1;
      long long tiles_min = (128 * 1024) / per_tile; if (tiles_min < 1) tiles_min = 1;
      long long chunk_tiles = n_sel;
      { const char *e = getenv("RT_CHUNK_TILES"); if (e && atoll(e) > 0) chunk_tiles = atoll(e); else if (chunk_tiles < tiles_min) chunk_tiles = tiles_min; }
      if (chunk_tiles > tiles_cap) chunk_tiles = tiles_cap;
      if (chunk_tiles > n_sel) chunk_tiles = n_sel;
      const long long n_chunks = (n_sel + chunk_tiles - 1) / chunk_tiles;
      const int ring = 1;        /* the chunks of a frame run one after the other on one stream: one slice of the arena serves them all */

      /* one arena slice per chunk in flight, sized for the largest chunk */
      const long long np_max = chunk_tiles * per_tile;
      /* beside the entries the bounce rays need, every wave of the launch may leave one reserved range partly unused */
      const long long launch_waves = (np_max + RT_BLOCK - 1) / RT_BLOCK < s->trace_blocks ? (np_max + RT_BLOCK - 1) / RT_BLOCK * (RT_BLOCK / 64)
                                                                                           : (long long)s->trace_blocks * (RT_BLOCK / 64);
      long long cap_ll = (long long)((double)np_max * F) + (max_rec > 1 ? launch_waves * (long long)s->reserve : 0) + 64;
      if (cap_ll >= 0x7fffffffLL) { rt_set_error("ray queue too large"); return RT_ERR_TOO_LARGE; }       /* entry numbers keep their top bit free */
      const unsigned cap = (unsigned)cap_ll;
      size_t used = 0;
      auto carve = [&](size_t bytes) { size_t o = used; used += align_up(bytes ? bytes : 1, 256); return o; };
      const size_t o_queue = carve(sizeof(QRay) * (size_t)cap), o_hits = carve(sizeof(Hit) * (size_t)cap);
      const size_t o_recs = carve(sizeof(SpawnRec) * (size_t)cap), o_totals = carve(sizeof(float) * 3 * (size_t)cap);
      const size_t o_points = carve(sizeof(float4) * (size_t)cap);
      const size_t o_rgb = carve(sizeof(float) * 3 * (size_t)np_max);
      const size_t o_dur = carve(sizeof(unsigned) * (size_t)cap);
      /* one bit per (queue entry, light): isInShadow, set by the feelers that found an occluder */
      const unsigned flag_stride = (unsigned)((L > 0 ? L : 1) + 31) / 32u;
      const size_t flag_bytes = sizeof(unsigned) * (size_t)flag_stride * (size_t)cap;
      const size_t o_flags = carve(flag_bytes);
      /* per tile of a chunk: did a primary of the tile spawn a bounce ray?  Tiles where none did are shaded in k_resolve */
      const bool fuse = view == VIEW_FRAME && s->fuse_resolve;
      const size_t o_bounce = carve(sizeof(unsigned) * 2 * (size_t)chunk_tiles);      /* and, behind those, the same per tile for "hit anything" */
      const size_t slice = used;
      last_hits_offset = n_chunks == 1 ? (long long)o_hits : -1;
      if (slice * (size_t)ring > s->arena.cap) { int rc = arena_reserve(s->arena, slice * (size_t)ring); if (rc != RT_OK) return rc; }
      { int rc = ensure_lists(s); if (rc != RT_OK) return rc; }
#define AT(T, r, off) ((T *)(s->arena.base + (size_t)(r) * slice + (off)))

      /* tile start order: only for a frame that is one launch, and from a frame of the same geometry */
      const long long key[6] = {p->width, p->height, p->sample_count, tile_first, tile_stride, n_sel};
      const bool track_cost = s->use_tile_order && n_chunks == 1 && view != VIEW_PRIMARY && per_tile % 64 == 0;
      if (track_cost && n_sel > s->order_cap) {
         if (s->d_tile_cost) (void)hipFree(s->d_tile_cost);
         if (s->d_tile_order) (void)hipFree(s->d_tile_order);
         if (s->h_tile_cost) (void)hipHostFree(s->h_tile_cost);
         if (s->h_tile_order) (void)hipHostFree(s->h_tile_order);
         s->d_tile_cost = s->h_tile_cost = nullptr; s->d_tile_order = s->h_tile_order = nullptr; s->order_cap = 0; s->order_valid = false; s->order_dirty = true;
         const size_t bytes = sizeof(unsigned) * (size_t)n_sel;
         /* two cost words per tile: its longest chain and the sum of its chains */
         HIP_TRY(hipMalloc((void **)&s->d_tile_cost, 2 * bytes)); HIP_TRY(hipMalloc((void **)&s->d_tile_order, bytes));
         HIP_TRY(hipHostMalloc((void **)&s->h_tile_cost, 2 * bytes, hipHostMallocDefault));
         HIP_TRY(hipHostMalloc((void **)&s->h_tile_order, bytes, hipHostMallocDefault));
         s->order_cap = n_sel;
      }
      bool have_order = track_cost && s->order_valid && memcmp(key, s->order_key, sizeof key) == 0;
      /* Tiles no ray can meet anything through are left out of the launch: the start order lists the others first, and
       * only those are handed out.  (Needs what the start order needs, and the per-tile "hit anything" flags that keep
       * the kernels behind the traversal from reading such a tile.) */
      bool cull = false;
      if (track_cost && fuse && s->cull_enabled && sc.n_planes == 0 && s->bounds_ok) {
         const float camf[12] = {cam.pos.x, cam.pos.y, cam.pos.z, cam.up.x, cam.up.y, cam.up.z, cam.right.x, cam.right.y, cam.right.z,
                                 cam.look_dir.x, cam.look_dir.y, cam.look_dir.z};
         if (s->n_live < 0 || memcmp(key, s->cull_key, sizeof key) != 0 || memcmp(camf, s->cull_cam, sizeof camf) != 0) {
            cull_tiles(s, cam, g, n_sel, s->tile_live, &s->n_live);
            memcpy(s->cull_key, key, sizeof key); memcpy(s->cull_cam, camf, sizeof camf);
         }
         cull = s->n_live < n_sel;
      }
      culled_frame = cull;
      /* A frame without a start order yet (the first of its geometry) starts its tiles from the middle of the scene's
       * picture outwards: the tiles the scene's box projects onto are the likely costly ones, and a launch that keeps
       * them for its end finishes on their long chains (bunny headline, first frame: k_trace 1.5 -> 1.2 ms). */
      bool first_order = false;
      if (track_cost && !have_order && s->bounds_ok) {
         first_order_guess(s, cam, g, n_sel, s->h_tile_order);
         s->order_valid = false; s->order_dirty = true; first_order = true;
      }
      if (cull) {
         /* live tiles first, in the order they have */
         if (!have_order && !first_order) { for (long long k = 0; k < n_sel; k++) s->h_tile_order[k] = (int)k; s->order_valid = false; s->order_dirty = true; }
         std::vector<int> parted((size_t)n_sel);
         size_t at = 0;
         for (long long k = 0; k < n_sel; k++) if (s->tile_live[(size_t)s->h_tile_order[k]]) parted[at++] = s->h_tile_order[k];
         for (long long k = 0; k < n_sel; k++) if (!s->tile_live[(size_t)s->h_tile_order[k]]) parted[at++] = s->h_tile_order[k];
         if (memcmp(parted.data(), s->h_tile_order, sizeof(int) * (size_t)n_sel) != 0) {
            memcpy(s->h_tile_order, parted.data(), sizeof(int) * (size_t)n_sel);
            s->order_dirty = true;
         }
      }

      /* the frame is timed on the caller's stream; the work runs there too unless the scene uses a stream of its own */
      EventPool ev(s->timing_events);
      Timer &t_total = ev.begin(stream, &st.ms_total);
      if (M != stream) {
         HIP_TRY(hipEventRecord(s->ev_begin, stream));
         HIP_TRY(hipStreamWaitEvent(M, s->ev_begin, 0));
      }

      for (long long c = 0; c < n_chunks; c++) {
         const int r = (int)(c % ring);
         hipStream_t S = M;
         const long long k0 = c * chunk_tiles;
         const long long nb = n_sel - k0 < chunk_tiles ? n_sel - k0 : chunk_tiles;
         const long long np = nb * per_tile;
         g.k0 = (int)k0;
         ChunkCtr ctr = chunk_ctr(s, r);
         QRay *queue = AT(QRay, r, o_queue);
         Hit *hits = AT(Hit, r, o_hits);
         SpawnRec *recs = AT(SpawnRec, r, o_recs);

         {
            /* counters to zero, spawned rays appended behind the primaries, tile costs and (first chunk) statistics cleared */
            const int n_stats = c == 0 ? ST_COUNT * RT_STAT_REPL : 0, n_cost = track_cost ? 2 * (int)n_sel : 0;
            int n_init = n_stats > n_cost ? (n_stats > RT_CTR_WORDS ? n_stats : RT_CTR_WORDS) : (n_cost > RT_CTR_WORDS ? n_cost : RT_CTR_WORDS);
            if (fuse && 2 * (int)nb > n_init) n_init = 2 * (int)nb;
            hipLaunchKernelGGL(k_begin_chunk, dim3(grid_for(n_init)), dim3(RT_BLOCK), 0, S, c == 0 ? s->d_stats : (unsigned long long *)nullptr,
                               n_stats, s->d_ctr, ctr.tail, (unsigned)np, track_cost ? s->d_tile_cost : (unsigned *)nullptr, n_cost,
                               fuse ? AT(unsigned, r, o_bounce) : (unsigned *)nullptr, 2 * (int)nb);
            TraceArgs a; memset(&a, 0, sizeof a);
            a.rays = queue; a.n = a.n_fetch = (unsigned long long)np; a.hits = hits; a.next = ctr.next_chain;
            a.flag_words = AT(unsigned, r, o_flags); a.flag_stride = flag_stride;
            a.recs = recs; a.tail = ctr.tail; a.cap = cap; a.max_rec = max_rec; a.cam = cam; a.g = g;
            a.points = AT(float4, r, o_points);
            a.lists = s->d_lists; a.list_cap = s->list_cap;
            if (!a.lists || a.list_cap == 0 || (a.list_cap & (a.list_cap - 1)) != 0) { rt_set_error("internal: spawn lists not set up"); return RT_ERR_HIP; }
            if (cull) a.n_fetch = (unsigned long long)(s->n_live * per_tile);
            /* (worked out per shaded point and light by the few lanes of a spawn stage: with 256 lights that costs more than the
             * 6 % of the 1M-triangle scene's feelers it stops are worth, 247 against 230 ms at 1080p 4 spp) */
            a.skip_unlit = (s->skip_unlit && p->shading == RT_PHONG && L >= 1 && L <= s->skip_unlit && s->lights_finite) ? 1 : 0;
            if (have_order || cull || first_order) {
               if (s->order_dirty) {
                  HIP_TRY(hipMemcpyAsync(s->d_tile_order, s->h_tile_order, sizeof(int) * (size_t)n_sel, hipMemcpyHostToDevice, S));
                  s->order_dirty = false;
               }
               a.tile_order = s->d_tile_order;
            }
            if (track_cost) { a.dur = AT(unsigned, r, o_dur); a.tile_cost = s->d_tile_cost; }
            if (fuse) { a.tile_bounce = AT(unsigned, r, o_bounce); a.tile_hit = a.tile_bounce + nb; }
            Timer &tm = ev.begin(S, &st.ms_trace);
            launch_trace(s, OUT_CHAIN, count, a, a.n_fetch, lds_bytes, S, filter_kinds);
            ev.end(tm, S); st.launches_trace++;
         }
         if (view == VIEW_PRIMARY) {
            hipLaunchKernelGGL(k_unslot_hits, dim3(grid_for(np)), dim3(RT_BLOCK), 0, S, g, (const Hit *)hits, np, sc.prim_ids, sc.n_prims, prim_dev, t_dev);
            continue;
         }
         /* A scene without a reflecting or refracting material spawns no bounce ray: every tile is k_resolve's then, and
          * the two kernels before it have nothing to do. */
         if (!(fuse && !s->can_bounce)) {
            Timer &tm = ev.begin(S, &st.ms_light, count);
            hipLaunchKernelGGL(k_light, dim3((unsigned)nb + grid_for((long long)cap - np)), dim3(RT_BLOCK), 0, S, sc, (const QRay *)queue, (const Hit *)hits,
                               (const unsigned *)AT(unsigned, r, o_flags), flag_stride, (const unsigned *)ctr.tail, cap,
                               p->shading, AT(float, r, o_totals), fuse ? (const unsigned *)AT(unsigned, r, o_bounce) : (const unsigned *)nullptr,
                               (unsigned)np, (unsigned)per_tile);
            ev.end(tm, S);
            Timer &tc = ev.begin(S, &st.ms_combine, count);
            hipLaunchKernelGGL(k_combine_tree, dim3((unsigned)nb), dim3(RT_BLOCK), 0, S, (const Hit *)hits, (const SpawnRec *)recs,
                               (const float *)AT(float, r, o_totals), track_cost ? (const unsigned *)AT(unsigned, r, o_dur) : (const unsigned *)nullptr, np, AT(float, r, o_rgb),
                               track_cost ? s->d_tile_cost : (unsigned *)nullptr, (unsigned)per_tile,
                               fuse ? (const unsigned *)AT(unsigned, r, o_bounce) : (const unsigned *)nullptr);
            ev.end(tc, S);
         }
         if (view == VIEW_FRAME) {
            Timer &tr = ev.begin(S, &st.ms_resolve, count);
            long long npx = nb * RT_TILE * RT_TILE;
            ResolveShade rs; memset(&rs, 0, sizeof rs);
            rs.sc = sc; rs.queue = queue; rs.hits = hits; rs.recs = recs; rs.flag_words = AT(unsigned, r, o_flags); rs.flag_stride = flag_stride;
            rs.shading = p->shading; rs.tile_bounce = fuse ? AT(unsigned, r, o_bounce) : nullptr; rs.tile_hit = fuse ? AT(unsigned, r, o_bounce) + nb : nullptr;
            rs.dur = track_cost ? AT(unsigned, r, o_dur) : nullptr; rs.tile_cost = track_cost ? s->d_tile_cost : nullptr;
            hipLaunchKernelGGL(k_resolve, dim3(grid_for(npx)), dim3(RT_BLOCK), 0, S, g, (const float *)AT(float, r, o_rgb), (int)nb,
                               p->out_layout, (uchar4 *)out_dev, rs);
            ev.end(tr, S);
         } else {
            hipLaunchKernelGGL(k_unslot_rgb, dim3(grid_for(np)), dim3(RT_BLOCK), 0, S, g, (const float *)AT(float, r, o_rgb), np, rgb_dev);
         }
         /* the tile costs are complete once the last kernel that adds to them has run */
         if (track_cost && view != VIEW_PRIMARY) HIP_TRY(hipMemcpyAsync(s->h_tile_cost, s->d_tile_cost, 2 * sizeof(unsigned) * (size_t)n_sel, hipMemcpyDeviceToHost, S));
         HIP_TRY(hipGetLastError());
      }
#undef AT
      HIP_TRY(hipMemcpyAsync(s->h_stats, s->d_stats, ST_COUNT * RT_STAT_REPL * sizeof(unsigned long long), hipMemcpyDeviceToHost, M));
      if (M != stream) {
         HIP_TRY(hipEventRecord(s->ev_end, M));
         HIP_TRY(hipStreamWaitEvent(stream, s->ev_end, 0));
      }
      ev.end(t_total, stream);
#ifdef RT_DEBUG_HOST
      dbg_t1 = std::chrono::steady_clock::now();
#endif
      HIP_TRY(hipStreamSynchronize(stream));
      if (M != stream) HIP_TRY(hipStreamSynchronize(M));
#ifdef RT_DEBUG_HOST
      dbg_t2 = std::chrono::steady_clock::now();
#endif
      HIP_TRY(hipGetLastError());
      ev.collect();
      if (sum_stat(s->h_stats, ST_ABORTED) != 0) { rt_set_error("traversal aborted: step cap reached (corrupt scene?)"); return RT_ERR_HIP; }
      /* queue entries per primary the frame wanted (its chunks taken together) */
      const double all_slots = (double)n_sel * (double)per_tile;
      double wanted = (all_slots + (double)sum_stat(s->h_stats, ST_REFLECT) + (double)sum_stat(s->h_stats, ST_REFRACT)) / all_slots;
      if (sum_stat(s->h_stats, ST_LIST_OVERFLOW) != 0) {
         /* a wave's list of spawned rays was full and some were dropped: longer lists, and the frame again */
         if (attempt >= 8 || s->list_cap >= RT_LIST_CAP_MAX) { rt_set_error("spawn list overflow persists at %u entries per wave", s->list_cap); return RT_ERR_TOO_LARGE; }
         s->list_cap *= 2; s->list_waves = 0;
         continue;
      }
      if (sum_stat(s->h_stats, ST_OVERFLOW) != 0) {
         if (attempt >= 8) { rt_set_error("ray queue overflow persists"); return RT_ERR_TOO_LARGE; }
         double grown = wanted * 1.25 > 2.0 * F ? wanted * 1.25 : 2.0 * F;
         s->node_factor = (float)(grown < 64.0 ? grown : 64.0);
         continue;                              /* some bounce rays were dropped: render the frame again */
      }
      if (max_rec > 1 && wanted * 1.1 > F) s->node_factor = (float)(wanted * 1.25 < 64.0 ? wanted * 1.25 : 64.0);
      /* (the costs are the rays' own step counts: a still camera brings the same ones back every frame, and then the
       * order stands as it is) */
      const bool same_costs = track_cost && s->order_valid && memcmp(key, s->order_key, sizeof key) == 0 && !s->last_cost.empty() &&
                              s->last_cost.size() == 2 * (size_t)n_sel &&
                              memcmp(s->last_cost.data(), s->h_tile_cost, 2 * sizeof(unsigned) * (size_t)n_sel) == 0;
      if (track_cost && !same_costs) {
         s->last_cost.assign(s->h_tile_cost, s->h_tile_cost + 2 * (size_t)n_sel);
         /* Costliest first, in a few classes of equal size (by rank of the tile's longest chain); tiles of a class stay
          * in image order, so neighbouring waves still work on neighbouring tiles: a scattered order costs more in the
          * caches than the shorter tail wins. */
         const int classes = s->order_classes;
         /* a tile's cost: its longest chain, plus order_mean_weight times its mean chain (a tile full of long chains
          * keeps its waves busy longer than one with a single long chain) */
         std::vector<unsigned> costs((size_t)n_sel);
         for (long long k = 0; k < n_sel; k++)
            costs[(size_t)k] = s->h_tile_cost[2 * k] + (unsigned)((double)s->h_tile_cost[2 * k + 1] * s->order_mean_weight / (double)per_tile);
         std::vector<unsigned> sorted(costs);
         std::sort(sorted.begin(), sorted.end(), std::greater<unsigned>());
         std::vector<unsigned> bound((size_t)classes);        /* class c holds costs above bound[c] (and not in an earlier class) */
         for (int c = 0; c < classes; c++) { long long at = (long long)(c + 1) * n_sel / classes; bound[(size_t)c] = at < n_sel ? sorted[(size_t)at] : 0u; }
         /* bound[] falls with the class number: the class of a cost is the number of bounds it does not exceed */
         auto cls = [&](unsigned cost) {
            int lo_c = 0, hi_c = classes - 1;           /* the first c in [0, classes - 1) with cost > bound[c], else classes - 1 */
            while (lo_c < hi_c) { int mid = (lo_c + hi_c) / 2; if (cost > bound[(size_t)mid]) hi_c = mid; else lo_c = mid + 1; }
            return lo_c;
         };
         std::vector<long long> start((size_t)classes + 1, 0);
         for (long long k = 0; k < n_sel; k++) start[(size_t)cls(costs[(size_t)k]) + 1]++;
         for (int c = 0; c < classes; c++) start[(size_t)c + 1] += start[(size_t)c];
         std::vector<int> order((size_t)n_sel);
         for (long long k = 0; k < n_sel; k++) order[(size_t)start[(size_t)cls(costs[(size_t)k])]++] = (int)k;
         /* tiles that are left out of the launch come last (they cost nothing, but so do traced tiles of background) */
         if (culled_frame) std::stable_partition(order.begin(), order.end(), [&](int k) { return s->tile_live[(size_t)k] != 0; });
         const bool same = s->order_valid && memcmp(key, s->order_key, sizeof key) == 0 &&
                           memcmp(order.data(), s->h_tile_order, sizeof(int) * (size_t)n_sel) == 0;
         if (!same) {         /* an unchanged order (a still camera) is not uploaded again */
            memcpy(s->h_tile_order, order.data(), sizeof(int) * (size_t)n_sel);
            s->order_dirty = true;
         }
         memcpy(s->order_key, key, sizeof key);
         s->order_valid = true;
      }
      break;
   }
   /* The primaries are the selected tiles' samples that lie inside the frame; every shaded point issues one feeler per
    * light, traced unless its own-surface test missed: neither needs a device counter. */
   {
      unsigned long long inside = 0;
      const int tiles_x = (p->width + RT_TILE - 1) / RT_TILE;
      for (long long k = 0; k < n_sel; k++) {
         long long tile = tile_first + k * tile_stride;
         long long x0 = (tile % tiles_x) * RT_TILE, y0 = (tile / tiles_x) * RT_TILE;
         long long w = p->width - x0 < RT_TILE ? p->width - x0 : RT_TILE, h = p->height - y0 < RT_TILE ? p->height - y0 : RT_TILE;
         if (w > 0 && h > 0) {
            inside += (unsigned long long)(w * h) * (unsigned long long)p->sample_count;
            if (culled_frame && !s->tile_live[(size_t)k]) st.primaries_culled += (unsigned long long)(w * h) * (unsigned long long)p->sample_count;
         }
      }
      st.rays_primary = inside;
   }
   st.shadow_skipped = sum_stat(s->h_stats, ST_SHADOW_SKIPPED);
   st.rays_reflect = sum_stat(s->h_stats, ST_REFLECT);
   st.rays_refract = sum_stat(s->h_stats, ST_REFRACT);
   st.shade_nodes = sum_stat(s->h_stats, ST_SHADE_NODES);
   st.rays_shadow = st.shade_nodes * (unsigned long long)(L > 0 ? L : 0) - st.shadow_skipped;
   st.rays_total = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract;
   st.samples = st.rays_primary;
   st.node_visits = sum_stat(s->h_stats, ST_NODE_VISITS); st.prim_tests = sum_stat(s->h_stats, ST_PRIM_TESTS); st.tie_events = sum_stat(s->h_stats, ST_TIES);
   st.node_visits_shadow = sum_stat(s->h_stats, ST_NODE_VISITS_SHADOW); st.prim_tests_shadow = sum_stat(s->h_stats, ST_PRIM_TESTS_SHADOW);
   st.tie_events_shadow = sum_stat(s->h_stats, ST_TIES_SHADOW);
   st.wave_node_steps = sum_stat(s->h_stats, ST_WAVE_NODE_STEPS); st.wave_leaf_steps = sum_stat(s->h_stats, ST_WAVE_LEAF_STEPS);
   st.lane_steps_at_leaf = sum_stat(s->h_stats, ST_LEAF_WAIT); st.lane_steps_no_ray = sum_stat(s->h_stats, ST_NO_RAY);
   st.wave_refills = sum_stat(s->h_stats, ST_REFILLS);
   st.hits_certified = sum_stat(s->h_stats, ST_CERT_OK); st.hits_contested = sum_stat(s->h_stats, ST_CERT_RIVAL);
   st.hits_unreached = sum_stat(s->h_stats, ST_CERT_UNREACHED);
   st.shadows_certified = sum_stat(s->h_stats, ST_SHADOW_CERT);
   st.shadows_unneeded = sum_stat(s->h_stats, ST_SHADOW_UNNEEDED);
   {
      double w = (double)sum_stat(s->h_stats, ST_CLK_WAVE);
      st.wave_time_refill = w > 0 ? (float)((double)sum_stat(s->h_stats, ST_CLK_REFILL) / w) : 0.0f;
      st.wave_time_node = w > 0 ? (float)((double)sum_stat(s->h_stats, ST_CLK_NODE) / w) : 0.0f;
      st.wave_time_leaf = w > 0 ? (float)((double)sum_stat(s->h_stats, ST_CLK_LEAF) / w) : 0.0f;
   }
   if (probing && s->probe_step < 5) {
      if (s->probe_step >= 1) s->probe_ms[filter_kinds] = st.ms_trace;
      if (++s->probe_step == 5) {
         /* the quickest stays; anything but "both" has to win by more than the frame-to-frame noise */
         int best = 3;
         for (int k = 0; k < 3; k++) if (s->probe_ms[k] < 0.96f * s->probe_ms[3] && s->probe_ms[k] < s->probe_ms[best]) best = k;
         s->filter_choice = best;
      }
   }
   if (view == VIEW_FRAME && last_hits_offset >= 0) {
      g.k0 = 0;
      s->last_frame.g = g; s->last_frame.np = (long long)n_sel * per_tile; s->last_frame.hits_offset = last_hits_offset;
      s->last_frame.culled = culled_frame; s->last_frame.n_sel = n_sel; s->last_frame.valid = true;
   }
   st.second_tree = filter_kinds;
   st.max_ray_steps = s->h_stats[ST_MAX_RAY_STEPS * RT_STAT_REPL];
   st.max_spawn_list = s->h_stats[ST_MAX_LIST * RT_STAT_REPL];
   st.attempts = attempts;
#ifdef RT_DEBUG_EXIT_HIST
   fprintf(stderr, "wave exits per 0.25 ms:");
   for (int b = 0; b < 48; b++) fprintf(stderr, " %llu", (unsigned long long)sum_stat(s->h_stats, ST_HIST + b));
   fprintf(stderr, "\n");
#endif
   if (count && sum_stat(s->h_stats, ST_WAVES) > 0 && st.launches_trace == 1) {
      double first = (double)(~s->h_stats[ST_CLOCK_FIRST_START * RT_STAT_REPL]);
      double waves = (double)sum_stat(s->h_stats, ST_WAVES);
      st.wave_exit_mean_ms = (float)(((double)sum_stat(s->h_stats, ST_CLOCK_EXIT_SUM) / waves - first) * 1e-5);
      st.wave_exit_last_ms = (float)(((double)s->h_stats[ST_CLOCK_LAST_EXIT * RT_STAT_REPL] - first) * 1e-5);
   }
   if (stats_out) *stats_out = st;
#ifdef RT_DEBUG_HOST
   {
      auto t3 = std::chrono::steady_clock::now();
      auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      if (dbg_n > 0) dbg_acc[3] += us(dbg_prev_exit, dbg_t0);
      dbg_acc[0] += us(dbg_t0, dbg_t1); dbg_acc[1] += us(dbg_t1, dbg_t2); dbg_acc[2] += us(dbg_t2, t3); dbg_n++;
      dbg_prev_exit = t3;
      if (dbg_n % 10 == 0) { fprintf(stderr, "host us per frame: enqueue %.1f, wait %.1f, after %.1f, outside the call %.1f\n", dbg_acc[0] / 10, dbg_acc[1] / 10, dbg_acc[2] / 10, dbg_acc[3] / 10); dbg_acc[0] = dbg_acc[1] = dbg_acc[2] = dbg_acc[3] = 0; }
   }
#endif
   return RT_OK;
}

extern "C" int rt_render_device(const rt_scene *scene, const rt_render_params *params, void *out_rgba_dev, void *stream,
                                rt_stats *stats)
{
   if (!out_rgba_dev) { rt_set_error("rt_render_device: null output"); return RT_ERR_BAD_ARGUMENT; }
   return render_impl(scene, params, VIEW_FRAME, out_rgba_dev, nullptr, nullptr, nullptr, (hipStream_t)stream, stats);
}

extern "C" int rt_render(const rt_scene *scene, const rt_render_params *params, uint8_t *out_host, rt_stats *stats)
{
   if (!scene || !params || !out_host) { rt_set_error("rt_render: null argument"); return RT_ERR_BAD_ARGUMENT; }
   if (params->width <= 0 || params->height <= 0) { rt_set_error("rt_render: non-positive size"); return RT_ERR_BAD_ARGUMENT; }
   if (scene->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   HIP_TRY(hipSetDevice(scene->device));
   rt_render_params p = *params;
   p.out_layout = RT_OUT_FRAME;
   size_t bytes = (size_t)p.width * (size_t)p.height * 4;
   rt_scene *s = const_cast<rt_scene *>(scene);      /* the frame buffer is scratch state of the scene */
   if (bytes > s->frame_cap) {
      if (s->d_frame) { (void)hipFree(s->d_frame); s->d_frame = nullptr; s->frame_cap = 0; }
      HIP_TRY(hipMalloc(&s->d_frame, bytes));
      s->frame_cap = bytes;
   }
   void *d_out = s->d_frame;
   /* pixels of tiles this call does not select read as 0 */
   const bool all_tiles = p.tile_first == 0 && p.tile_stride <= 1;
   hipError_t e = all_tiles ? hipSuccess : hipMemset(d_out, 0, bytes);
   int rc = e == hipSuccess ? render_impl(scene, &p, VIEW_FRAME, d_out, nullptr, nullptr, nullptr, nullptr, stats) : RT_ERR_HIP;
   if (rc == RT_OK) {
      e = hipMemcpy(out_host, d_out, bytes, hipMemcpyDeviceToHost);   /* kernel.cu:759-760 */
      if (e != hipSuccess) { rt_set_error("frame copy failed: %s", hipGetErrorString(e)); rc = RT_ERR_HIP; }
   }
   return rc;
}

extern "C" int rt_untile_device(const void *tiles_dev, size_t part_stride_bytes, int n_parts, int width, int height,
                                void *frame_dev, void *stream)
{
   if (!tiles_dev || !frame_dev || n_parts <= 0 || width <= 0 || height <= 0) { rt_set_error("rt_untile_device: bad argument"); return RT_ERR_BAD_ARGUMENT; }
   long long n = (long long)width * height;
   hipLaunchKernelGGL(k_untile, dim3(grid_for(n)), dim3(RT_BLOCK), 0, (hipStream_t)stream, (const uchar4 *)tiles_dev,
                      (long long)(part_stride_bytes / 4), n_parts, width, height, (width + RT_TILE - 1) / RT_TILE, (uchar4 *)frame_dev);
   HIP_TRY(hipGetLastError());
   return RT_OK;
}

extern "C" int rt_trace_primary(const rt_scene *scene, const rt_render_params *params, int32_t *prim_host, float *t_host)
{
   if (!scene || !params || !prim_host || !t_host) { rt_set_error("rt_trace_primary: null argument"); return RT_ERR_BAD_ARGUMENT; }
   int sroot = isqrt_trunc(params->sample_count);
   if (sroot <= 0 || sroot * sroot != params->sample_count) { rt_set_error("Invalid sample count: %d", params->sample_count); return RT_ERR_BAD_SAMPLE_COUNT; }
   if (scene->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   HIP_TRY(hipSetDevice(scene->device));
   size_t n = (size_t)params->width * sroot * (size_t)params->height * sroot;
   int *d_prim = nullptr; float *d_t = nullptr;
   HIP_TRY(hipMalloc((void **)&d_prim, n * 4));
   if (hipMalloc((void **)&d_t, n * 4) != hipSuccess) { (void)hipFree(d_prim); rt_set_error("hipMalloc failed"); return RT_ERR_HIP; }
   (void)hipMemset(d_prim, 0xff, n * 4); (void)hipMemset(d_t, 0, n * 4);
   rt_render_params p = *params; p.tile_first = 0; p.tile_stride = 1;
   int rc = render_impl(scene, &p, VIEW_PRIMARY, nullptr, d_prim, d_t, nullptr, nullptr, nullptr);
   if (rc == RT_OK) {
      if (hipMemcpy(prim_host, d_prim, n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(t_host, d_t, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { rt_set_error("copy back failed"); rc = RT_ERR_HIP; }
   }
   (void)hipFree(d_prim); (void)hipFree(d_t);
   return rc;
}

extern "C" int rt_frame_primary_hits(const rt_scene *scene, int32_t *prim_host, float *t_host, int64_t n)
{
   if (!scene || !prim_host || !t_host) { rt_set_error("rt_frame_primary_hits: null argument"); return RT_ERR_BAD_ARGUMENT; }
   if (scene->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   const rt_scene::LastFrame &f = scene->last_frame;
   if (!f.valid) { rt_set_error("rt_frame_primary_hits: the scene's last call was not a frame rendered in one launch"); return RT_ERR_BAD_ARGUMENT; }
   const DGrid &g = f.g;
   const size_t ns = (size_t)g.res_w * (size_t)g.res_h;
   if (n != (int64_t)ns) { rt_set_error("rt_frame_primary_hits: room for %lld samples, the frame has %zu", (long long)n, ns); return RT_ERR_BAD_ARGUMENT; }
   HIP_TRY(hipSetDevice(scene->device));
   int *d_prim = nullptr; float *d_t = nullptr;
   HIP_TRY(hipMalloc((void **)&d_prim, ns * 4));
   if (hipMalloc((void **)&d_t, ns * 4) != hipSuccess) { (void)hipFree(d_prim); rt_set_error("hipMalloc failed"); return RT_ERR_HIP; }
   (void)hipMemset(d_prim, 0xff, ns * 4); (void)hipMemset(d_t, 0, ns * 4);
   const Hit *hits = (const Hit *)(scene->arena.base + f.hits_offset);
   hipLaunchKernelGGL(k_unslot_hits, dim3(grid_for(f.np)), dim3(RT_BLOCK), 0, nullptr, g, hits, f.np, scene->d.prim_ids, scene->d.n_prims, d_prim, d_t);
   int rc = RT_OK;
   if (hipMemcpy(prim_host, d_prim, ns * 4, hipMemcpyDeviceToHost) != hipSuccess ||
       hipMemcpy(t_host, d_t, ns * 4, hipMemcpyDeviceToHost) != hipSuccess) { rt_set_error("copy back failed"); rc = RT_ERR_HIP; }
   (void)hipFree(d_prim); (void)hipFree(d_t);
   if (rc == RT_OK && f.culled) {
      /* a tile that was left out of the launch has no stored hits: the frame took every sample of it for a miss */
      for (int k = 0; k < f.n_sel; k++) {
         if (scene->tile_live[(size_t)k]) continue;
         const long long tile = (long long)g.tile_first + (long long)k * g.tile_stride;
         const long long x0 = (tile % g.tiles_x) * g.ts, y0 = (tile / g.tiles_x) * g.ts;
         for (long long y = y0; y < y0 + g.ts && y < g.res_h; y++)
            for (long long x = x0; x < x0 + g.ts && x < g.res_w; x++) { prim_host[y * g.res_w + x] = -1; t_host[y * g.res_w + x] = FLT_MAX; }
      }
   }
   return rc;
}

extern "C" int rt_render_samples(const rt_scene *scene, const rt_render_params *params, float *rgb_host, rt_stats *stats)
{
   if (!scene || !params || !rgb_host) { rt_set_error("rt_render_samples: null argument"); return RT_ERR_BAD_ARGUMENT; }
   int sroot = isqrt_trunc(params->sample_count);
   if (sroot <= 0 || sroot * sroot != params->sample_count) { rt_set_error("Invalid sample count: %d", params->sample_count); return RT_ERR_BAD_SAMPLE_COUNT; }
   if (scene->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   HIP_TRY(hipSetDevice(scene->device));
   size_t n = (size_t)params->width * sroot * (size_t)params->height * sroot;
   float *d_rgb = nullptr;
   HIP_TRY(hipMalloc((void **)&d_rgb, n * 12));
   (void)hipMemset(d_rgb, 0, n * 12);
   rt_render_params p = *params; p.tile_first = 0; p.tile_stride = 1;
   int rc = render_impl(scene, &p, VIEW_SAMPLES, nullptr, nullptr, nullptr, d_rgb, nullptr, stats);
   if (rc == RT_OK && hipMemcpy(rgb_host, d_rgb, n * 12, hipMemcpyDeviceToHost) != hipSuccess) { rt_set_error("copy back failed"); rc = RT_ERR_HIP; }
   (void)hipFree(d_rgb);
   return rc;
}

/* ---- arbitrary rays: rt_trace_rays and its variant that allocates nothing ------------------------------------ */

/* queue entries from the caller's ray array (n x {o, d}) and optional toBeat array */
__global__ __launch_bounds__(RT_BLOCK) void k_pack_rays(const float *raw, const float *to_beat, long long n, QRay *q)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= n) return;
   const float *r = raw + 6 * i;
   store_qray(q, (unsigned long long)i, mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), to_beat ? to_beat[i] : -FLT_MAX, 1);
}
/* hits back in the caller's form: primitive ids in the caller's numbering, and t */
__global__ __launch_bounds__(RT_BLOCK) void k_unpack_hits(const Hit *hits, long long n, const int *prim_ids, int n_prims, int *prim, float *t)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= n) return;
   int p = hits[i].prim;
   prim[i] = (p >= 0 && p < n_prims) ? prim_ids[p] : p;
   t[i] = hits[i].t;
}

/* Working memory of one rt_trace_rays call, wherever it lives: the launch's counters and statistics, the caller's rays
 * as they arrive (24 bytes each, plus 4 for toBeat) -- a region that later takes the results (8 bytes each) --, the
 * queue entries and the hits. */
struct RayWork {
   unsigned long long *ctr, *stats;
   float *raw, *beat;
   int *prim_out; float *t_out;
   QRay *queue; Hit *hits;
};
#define RT_RAYWORK_HEADER 65536u
static size_t raywork_bytes(size_t n) { return RT_RAYWORK_HEADER + align_up(28 * n, 256) + align_up(sizeof(QRay) * n, 256) + align_up(sizeof(Hit) * n, 256); }
static RayWork raywork_carve(char *base, size_t n)
{
   static_assert(sizeof(unsigned long long) * (RT_CTR_WORDS + ST_COUNT * RT_STAT_REPL) <= RT_RAYWORK_HEADER, "counters and statistics must fit the header");
   RayWork w;
   w.ctr = (unsigned long long *)base; w.stats = w.ctr + RT_CTR_WORDS;
   char *p = base + RT_RAYWORK_HEADER;
   w.raw = (float *)p; w.beat = (float *)(p + 24 * n);
   w.prim_out = (int *)p; w.t_out = (float *)(p + 4 * n);        /* the results reuse the region of the raw rays */
   p += align_up(28 * n, 256);
   w.queue = (QRay *)p; p += align_up(sizeof(QRay) * n, 256);
   w.hits = (Hit *)p;
   return w;
}

/* The body both entry points share: nothing here allocates. */
static int trace_rays_in(rt_scene *s, const RayWork &w, int64_t n, const float *rays_host, const float *to_beat_host,
                         int32_t *prim_host, float *t_host)
{
   size_t lds_bytes = 0;
   { int rc = trace_lds_bytes(s, &lds_bytes); if (rc != RT_OK) return rc; }
   hipStream_t S = nullptr;
   HIP_TRY(hipMemsetAsync(w.ctr, 0, sizeof(unsigned long long) * (RT_CTR_WORDS + ST_COUNT * RT_STAT_REPL), S));
   HIP_TRY(hipMemcpyAsync(w.raw, rays_host, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice, S));
   if (to_beat_host) HIP_TRY(hipMemcpyAsync(w.beat, to_beat_host, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, S));
   hipLaunchKernelGGL(k_pack_rays, dim3(grid_for(n)), dim3(RT_BLOCK), 0, S, (const float *)w.raw, to_beat_host ? (const float *)w.beat : (const float *)nullptr,
                      (long long)n, w.queue);
   TraceArgs a; memset(&a, 0, sizeof a);
   a.rays = w.queue; a.n = a.n_fetch = (unsigned long long)n; a.hits = w.hits; a.next = w.ctr; a.stats = w.stats;
   launch_trace(s, OUT_HIT, false, a, (unsigned long long)n, lds_bytes, S);
   hipLaunchKernelGGL(k_unpack_hits, dim3(grid_for(n)), dim3(RT_BLOCK), 0, S, (const Hit *)w.hits, (long long)n, s->d.prim_ids, s->d.n_prims, w.prim_out, w.t_out);
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipMemcpyAsync(prim_host, w.prim_out, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, S));
   HIP_TRY(hipMemcpyAsync(t_host, w.t_out, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, S));
   unsigned long long aborted[RT_STAT_REPL];
   HIP_TRY(hipMemcpyAsync(aborted, w.stats + ST_ABORTED * RT_STAT_REPL, sizeof aborted, hipMemcpyDeviceToHost, S));
   HIP_TRY(hipStreamSynchronize(S));
   for (int r = 0; r < RT_STAT_REPL; r++)
      if (aborted[r]) { rt_set_error("traversal aborted: step cap reached (corrupt scene?)"); return RT_ERR_HIP; }
   return RT_OK;
}

static int trace_rays_check(const rt_scene *s, int64_t n, const float *rays_host, int32_t *prim_host, float *t_host, const char *who)
{
   if (!s || !rays_host || !prim_host || !t_host || n < 0) { rt_set_error("%s: bad argument", who); return RT_ERR_BAD_ARGUMENT; }
   if (s->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   return RT_OK;
}

extern "C" int rt_trace_rays(const rt_scene *cs, int64_t n, const float *rays_host, const float *to_beat_host,
                             int32_t *prim_host, float *t_host)
{
   rt_scene *s = const_cast<rt_scene *>(cs);
   { int rc = trace_rays_check(s, n, rays_host, prim_host, t_host, "rt_trace_rays"); if (rc != RT_OK) return rc; }
   if (n == 0) return RT_OK;
   if (n > 0x7fffffffLL) { rt_set_error("rt_trace_rays: too many rays"); return RT_ERR_TOO_LARGE; }
   HIP_TRY(hipSetDevice(s->device));
   char *base = nullptr;
   hipError_t e = hipMalloc((void **)&base, raywork_bytes((size_t)n));
   if (e != hipSuccess) { rt_set_error("rt_trace_rays: hipMalloc(%zu) failed: %s", raywork_bytes((size_t)n), hipGetErrorString(e)); return RT_ERR_HIP; }
   int rc = trace_rays_in(s, raywork_carve(base, (size_t)n), n, rays_host, to_beat_host, prim_host, t_host);
   (void)hipFree(base);
   return rc;
}

/* The working memory of rt_trace_rays_static: one array per device in the code object, held until the process ends.
 * In host code the array's name is only a handle for hipGetSymbolAddress; kernels and copies get the device address
 * that call returns for the device that is current. */
static __device__ __align__(256) unsigned char scratch[RT_RAYWORK_HEADER + (28 + sizeof(QRay) + sizeof(Hit)) * (size_t)RT_TRACE_RAYS_STATIC_MAX + 3 * 256];
#define RT_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static char *g_scratch_ptr[RT_MAX_DEVICES];      /* per device: the address of `scratch` there, once looked up */

/* with `device` current: the device address of that device's copy of `scratch` */
static int scratch_address(int device, char **out)
{
   if (device < 0 || device >= RT_MAX_DEVICES) { rt_set_error("device %d is beyond the %d this library keeps working memory for", device, RT_MAX_DEVICES); return RT_ERR_BAD_ARGUMENT; }
   std::lock_guard<std::mutex> lock(g_scratch_mutex);
   if (!g_scratch_ptr[device]) {
      void *p = nullptr;
      hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
      if (e != hipSuccess || !p) { rt_set_error("hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e)); return RT_ERR_HIP; }
      g_scratch_ptr[device] = (char *)p;
   }
   *out = g_scratch_ptr[device];
   return RT_OK;
}

extern "C" int rt_trace_rays_static(const rt_scene *cs, int64_t n, const float *rays_host, const float *to_beat_host,
                                    int32_t *prim_host, float *t_host)
{
   rt_scene *s = const_cast<rt_scene *>(cs);
   { int rc = trace_rays_check(s, n, rays_host, prim_host, t_host, "rt_trace_rays_static"); if (rc != RT_OK) return rc; }
   if (n == 0) return RT_OK;
   if (n > (int64_t)RT_TRACE_RAYS_STATIC_MAX) {
      rt_set_error("rt_trace_rays_static: %lld rays exceed the %d the static working memory holds; use rt_trace_rays", (long long)n, RT_TRACE_RAYS_STATIC_MAX);
      return RT_ERR_TOO_LARGE;
   }
   HIP_TRY(hipSetDevice(s->device));
   char *base = nullptr;
   { int rc = scratch_address(s->device, &base); if (rc != RT_OK) return rc; }
   if (raywork_bytes((size_t)n) > sizeof scratch) { rt_set_error("internal: static working memory too small"); return RT_ERR_TOO_LARGE; }
   return trace_rays_in(s, raywork_carve(base, (size_t)n), n, rays_host, to_beat_host, prim_host, t_host);
}
