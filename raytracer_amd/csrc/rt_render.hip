/*
 * rt_render.hip -- the per-pixel render path on MI355X (gfx950, wave64).
 *
 * Replaces the reference's rayTrace megakernel and its helpers
 * (kernel.cu:67-382, 466-570) with a wavefront pipeline over ray queues:
 *
 *   generate primary rays of a batch of tiles into a ray queue    (generateCameraRays, kernel.cu:477-498)
 *   extend   closest hit of every ray of a depth level          (getClosestIntersection, kernel.cu:83-200)
 *   shadow   one any-hit query per (light, shade node)          (isInShadow, kernel.cu:202-207)
 *   shade    material + light sum per node, spawns the          (shadeObject / getReflection / getRefraction,
 *            reflection / refraction rays of the next level      kernel.cu:209-276, 335-382)
 *   combine  bottom-up: colour = local*(1-refl-alpha)            (kernel.cu:375-376)
 *                     + refl*reflected + alpha*refracted
 *   resolve  s x s box filter to 8-bit                           (averageBufferColors, kernel.cu:545-570)
 *
 * The reference recurses to depth 6 inside one thread; here each depth level
 * is one batch of launches and a node's children are indices into the next
 * level, so the final combine evaluates exactly the reference's expression
 * tree (same operands, same order) without recursion or float atomics.
 *
 * extend and shadow are the same persistent-thread traversal kernel (k_trace)
 * over a ray queue: a wave keeps 64 rays in flight, and as soon as enough of its
 * lanes have finished it pulls the next rays of the queue into the idle lanes
 * (one atomic per refill), so long and short rays do not stall each other.
 * Each ray still walks the tree in the reference's visiting order (near child
 * first with its `<=` tie rule, deferred sibling re-checked against the current
 * t) because near-ties between primitives are resolved by visiting order.  The
 * per-lane stack of deferred siblings lives in LDS, one column per lane.
 */
#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
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
   int n_prims, n_planes, n_lights, root;
   int fast_ok;      /* every BVH bound is finite and below 2^40 in magnitude */
};

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

enum { ST_SHADOW_TRACED = 0, ST_SHADOW_SKIPPED, ST_REFLECT, ST_REFRACT, ST_SHADE_NODES, ST_PRIMARY,
       ST_NODE_VISITS, ST_PRIM_TESTS, ST_TIES, ST_WAVE_NODE_STEPS, ST_WAVE_LEAF_STEPS,          /* extend launches */
       ST_NODE_VISITS_SHADOW, ST_PRIM_TESTS_SHADOW, ST_TIES_SHADOW, ST_WAVE_NODE_STEPS_SHADOW, ST_WAVE_LEAF_STEPS_SHADOW,
       ST_ABORTED, ST_MAX_RAY_STEPS, ST_MAX_RAY_STEPS_SHADOW, ST_COUNT };

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
   if (p.m.y & RT_FLAG_XFORM) r = ray_transform(wray, load_xform(sc.xforms, prim));
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
 * faces the eye more wins.  Rare (a few percent of rays), so kept out of line; it takes the arrays by
 * value so that the caller's scene view stays in registers. */
__device__ __noinline__ bool tie_prefers_new(const rt_dev_prim *prims, const rt_dev_xform *xforms, const rt_dev_shade *shades,
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
struct StackEntry { int child; float dist; };
extern __shared__ StackEntry lds_stack[];

enum { OUT_HIT = 0, OUT_FLAG = 1 };

#define RT_STEP_CAP (1ULL << 28)
#define RT_REFILL_IDLE 20   /* pull new rays once this many lanes of the wave are idle */
#define RT_NODE_BURST 6     /* node steps between two leaf steps / refill checks */

__device__ __forceinline__ QRay load_qray_rec(const QRay *q, unsigned long long i)
{
   const float4 *p = reinterpret_cast<const float4 *>(q + i);
   float4 a = p[0], b = p[1];
   QRay r;
   r.ox = a.x; r.oy = a.y; r.oz = a.z; r.to_beat = a.w; r.dx = b.x; r.dy = b.y; r.dz = b.z; r.valid = __float_as_int(b.w);
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

/* one primitive test of the traversal (kernel.cu:104-134): returns true when the ray is done early */
__device__ __forceinline__ bool leaf_test(const DScene &sc, const Ray &ray, int prim, float toBeat, float &t, int &closest,
                                          unsigned &n_ties)
{
   float dist = prim_intersect(sc, prim, ray);
   bool take = false;
   if (is_equal(t, dist)) {
      n_ties++;
      if (closest >= 0)
         take = tie_prefers_new(sc.prims, sc.xforms, sc.shades, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z,
                                closest, t, prim, dist);
   } else {
      take = dist < t && above_zero(dist);
   }
   if (take) {
      t = dist; closest = prim;
      return t < toBeat;                                 /* kernel.cu:117-121, 128-132 */
   }
   return false;
}

/* planes after the tree: kernel.cu:189-196 (no tie rule, no toBeat exit) */
__device__ __forceinline__ void plane_scan(const DScene &sc, const Ray &ray, float &t, int &closest)
{
   for (int k = 0; k < sc.n_planes; k++) {
      float dist = plane_intersect(sc, k, ray);
      if (less_than(dist, t) && above_zero(dist)) { closest = sc.n_prims + k; t = dist; }
   }
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
 * Only rays the short box test is exact for run here (values inside the exact-division range:
 * RayCtx::fast); that is every ray of an ordinary scene.  A wave that holds a ray with an axis-parallel
 * direction component -- typically from the one pixel column or row through the view axis -- runs the
 * variant of the node step that handles such axes, all other waves the shorter one.  Out-of-range rays
 * are appended to `slow_list` and traced by k_trace_slow with the literal arithmetic. */
template <int OUT, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void k_trace(DScene sc, const QRay *rays, unsigned long long n, Hit *hits,
                                                    unsigned char *flags, unsigned long long *next,
                                                    unsigned *slow_list, unsigned *slow_count,
                                                    unsigned long long *stats, int stat_base)
{
   StackEntry *stack = lds_stack + threadIdx.x;
   const int stride = RT_BLOCK;
   const unsigned lane = __lane_id();
   const unsigned long long below = lane ? (~0ULL >> (64 - lane)) : 0ULL;

   bool active = false;
   bool exhausted = false;             /* wave-uniform: the queue has no more rays */
   bool wave_par = false;              /* wave-uniform: some lane may hold a ray with a parallel axis */
   RayCtx c;
   c.r.o = mk3(0, 0, 0); c.r.d = mk3(1, 1, 1); c.yx = c.yy = c.yz = 1.0f; c.par = 0; c.fast = true;
   float t = FLT_MAX, toBeat = -FLT_MAX;
   int closest = -1, cur = 0, sp = 0;
   unsigned idx = 0;
   unsigned nv = 0, np = 0, nt = 0, wn = 0, wl = 0;
   unsigned ray_steps = 0, max_ray_steps = 0;   /* COUNT only: wave steps this ray has been in flight */

   /* a ray is finished: scan the planes unless it left early, write its result */
   auto finish = [&](bool early) {
      if (COUNT) { max_ray_steps = ray_steps > max_ray_steps ? ray_steps : max_ray_steps; }
      if (!early) plane_scan(sc, c.r, t, closest);
      if (OUT == OUT_HIT) { Hit h; h.prim = closest; h.t = t; hits[idx] = h; }
      else flags[idx] = less_than(t, toBeat) ? 1 : 0;            /* kernel.cu:206 */
      active = false;
   };
   /* pop deferred siblings until one is still closer than t (kernel.cu:160-186); none left = done */
   auto pop = [&]() {
      bool found = false;
      while (sp > 0) {
         sp--;
         StackEntry e = stack[sp * stride];
         if (e.dist < t) { cur = e.child; found = true; break; }
      }
      if (!found) finish(false);
   };

   /* one node step of this lane (kernel.cu:136-158); PAR: some lane of the wave has a parallel axis */
   auto node_step = [&](auto par_tag) {
      constexpr bool PAR = decltype(par_tag)::value;
      /* 56 of the record's 64 bytes: two boxes and the two child references */
      const char *rec = reinterpret_cast<const char *>(sc.nodes) + ((unsigned)cur << 6);
      float4 a = *reinterpret_cast<const float4 *>(rec);
      float4 b = *reinterpret_cast<const float4 *>(rec + 16);
      float4 cc = *reinterpret_cast<const float4 *>(rec + 32);
      int2 ch = *reinterpret_cast<const int2 *>(rec + 48);
      if (COUNT) nv++;
      float left = slab_fast<PAR>(c, a.x, a.y, a.z, a.w, b.x, b.y);
      float right = slab_fast<PAR>(c, b.z, b.w, cc.x, cc.y, cc.z, cc.w);
      /* kernel.cu:148-158 as predicates: right first on `<=`, the other child deferred if it is ahead too */
      bool aL = above_zero(left), aR = above_zero(right);
      bool goR = aR && (right <= left || !aL) && right < t;
      bool goL = !goR && aL && (left <= right || !aR) && left < t;
      if ((goR && aL) || (goL && aR)) {
         StackEntry e; e.child = goR ? ch.x : ch.y; e.dist = goR ? left : right;
         stack[sp * stride] = e; sp++;
      }
      if (goR || goL) cur = goR ? ch.y : ch.x;
      else pop();
   };

   /* Every wave reaches its exit in a bounded number of steps: a ray visits each node at most once per
    * stack entry, so RT_STEP_CAP is far beyond any real traversal; hitting it means a corrupted scene and
    * is reported through stats[ST_ABORTED] instead of spinning. */
   unsigned long long guard = 0;
   for (;;) {
      if (++guard > RT_STEP_CAP) { if (lane == 0) stat_add(stats, ST_ABORTED, 1ULL); break; }
      unsigned long long m_active = __ballot(active);
      int n_idle = 64 - __popcll(m_active);
      if (!exhausted && n_idle >= RT_REFILL_IDLE) {
         unsigned long long m_idle = ~m_active;
         int leader = __ffsll((long long)m_idle) - 1;
         unsigned long long base = 0;
         if ((int)lane == leader) base = atomicAdd(next, (unsigned long long)n_idle);
         base = __shfl(base, leader);
         if (base + (unsigned long long)n_idle >= n) exhausted = true;
         if (!active) {
            unsigned long long my = base + (unsigned long long)__popcll(m_idle & below);
            if (my < n) {
               QRay q = load_qray_rec(rays, my);
               if (q.valid) {
                  Ray r; r.o = mk3(q.ox, q.oy, q.oz); r.d = mk3(q.dx, q.dy, q.dz);
                  c = make_ctx(r);
                  if (c.fast && sc.fast_ok) {
                     toBeat = q.to_beat; idx = (unsigned)my;
                     t = FLT_MAX; closest = -1; sp = 0; cur = sc.root;
                     active = true; ray_steps = 0;
                     if (cur == RT_ROOT_EMPTY) finish(false);
                  } else {
                     slow_list[atomicAdd(slow_count, 1u)] = (unsigned)my;
                  }
               }
            }
         }
         m_active = __ballot(active);
         wave_par = __ballot(active && c.par != 0) != 0;
      }
      if (m_active == 0) {
         if (exhausted) break;
         continue;
      }
      /* ---- node steps ---- */
#pragma unroll 1
      for (int burst = 0; burst < RT_NODE_BURST; burst++) {
         bool in_node = active && cur >= 0;
         if (__ballot(in_node) == 0) break;
         if (COUNT) { wn++; if (active) ray_steps++; }
         if (in_node) {
            if (wave_par) node_step(std::true_type());
            else node_step(std::false_type());
         }
      }
      /* ---- leaf step ---- */
      bool at_leaf = active && cur < 0;
      if (__ballot(at_leaf) != 0) {
         if (COUNT) { wl++; if (active) ray_steps++; }
         if (at_leaf) {
            if (COUNT) np++;
            if (leaf_test(sc, c.r, ~cur, toBeat, t, closest, nt)) finish(true);
            else pop();
         }
      }
   }
   if (COUNT) {
      stat_add(stats, stat_base + 0, (unsigned long long)nv);
      stat_add(stats, stat_base + 1, (unsigned long long)np);
      stat_add(stats, stat_base + 2, (unsigned long long)nt);
      if (lane == 0) {
         stat_add(stats, stat_base + 3, (unsigned long long)wn);
         stat_add(stats, stat_base + 4, (unsigned long long)wl);
      }
      atomicMax(&stats[(stat_base == ST_NODE_VISITS ? ST_MAX_RAY_STEPS : ST_MAX_RAY_STEPS_SHADOW) * RT_STAT_REPL],
                (unsigned long long)max_ray_steps);
   }
}

/* k_trace_slow: the same query with the reference's literal box test (true divisions, parallel-axis
 * branches), one ray per thread, for the rays k_trace set aside.  Grid-stride over the device-side count. */
template <int OUT, bool COUNT>
__global__ __launch_bounds__(RT_BLOCK) void k_trace_slow(DScene sc, const QRay *rays, Hit *hits, unsigned char *flags,
                                                         const unsigned *slow_list, const unsigned *slow_count,
                                                         unsigned long long *stats, int stat_base)
{
   StackEntry *stack = lds_stack + threadIdx.x;
   const int stride = RT_BLOCK;
   unsigned total = *slow_count;
   unsigned nv = 0, np = 0, nt = 0;
   for (unsigned k = blockIdx.x * RT_BLOCK + threadIdx.x; k < total; k += gridDim.x * RT_BLOCK) {
      unsigned idx = slow_list[k];
      QRay q = load_qray_rec(rays, idx);
      Ray ray; ray.o = mk3(q.ox, q.oy, q.oz); ray.d = mk3(q.dx, q.dy, q.dz);
      float toBeat = q.to_beat, t = FLT_MAX;
      int closest = -1, cur = sc.root, sp = 0;
      bool done = (cur == RT_ROOT_EMPTY), early = false;
      unsigned long long guard = 0;
      while (!done) {
         if (++guard > RT_STEP_CAP) { stat_add(stats, ST_ABORTED, 1ULL); break; }
         bool need_pop = false;
         if (cur >= 0) {
            float4 a, b, cc, d;
            load_node(sc.nodes, cur, a, b, cc, d);
            if (COUNT) nv++;
            float left = slab_literal(ray, a.x, a.y, a.z, a.w, b.x, b.y);
            float right = slab_literal(ray, b.z, b.w, cc.x, cc.y, cc.z, cc.w);
            int li = __float_as_int(d.x), ri = __float_as_int(d.y);
            bool aL = above_zero(left), aR = above_zero(right);
            if (aR && (right <= left || !aL) && right < t) {
               if (aL) { StackEntry e; e.child = li; e.dist = left; stack[sp * stride] = e; sp++; }
               cur = ri;
            } else if (aL && (left <= right || !aR) && left < t) {
               if (aR) { StackEntry e; e.child = ri; e.dist = right; stack[sp * stride] = e; sp++; }
               cur = li;
            } else need_pop = true;
         } else {
            if (COUNT) np++;
            if (leaf_test(sc, ray, ~cur, toBeat, t, closest, nt)) { early = true; done = true; }
            else need_pop = true;
         }
         if (need_pop) {
            bool found = false;
            while (sp > 0) {
               sp--;
               StackEntry e = stack[sp * stride];
               if (e.dist < t) { cur = e.child; found = true; break; }
            }
            if (!found) done = true;
         }
      }
      if (!early) plane_scan(sc, ray, t, closest);
      if (OUT == OUT_HIT) { Hit h; h.prim = closest; h.t = t; hits[idx] = h; }
      else flags[idx] = less_than(t, toBeat) ? 1 : 0;
   }
   if (COUNT) {
      stat_add(stats, stat_base + 0, (unsigned long long)nv);
      stat_add(stats, stat_base + 1, (unsigned long long)np);
      stat_add(stats, stat_base + 2, (unsigned long long)nt);
   }
}

/* ======================================================================= */
/* ray sources                                                              */
/* ======================================================================= */
__device__ __forceinline__ unsigned hash32(unsigned x)
{
   x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
   return x;
}

/* sample slot -> sample coordinates on the (W*s) x (H*s) grid.  A tile's samples are
 * numbered in 8x8 patches so that 64 consecutive slots cover a compact square of the image. */
__device__ __forceinline__ bool slot_to_sample(const DGrid &g, long long slot, int &X, int &Y)
{
   int per_tile = g.ts * g.ts;
   int k = (int)(slot / per_tile);
   int r = (int)(slot % per_tile);
   int patch = r >> 6, lane = r & 63;
   int px = patch % g.patches_per_row, py = patch / g.patches_per_row;
   int sx = px * 8 + (lane & 7), sy = py * 8 + (lane >> 3);
   int tile = g.tile_first + (g.k0 + k) * g.tile_stride;
   int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
   X = tx * g.ts + sx; Y = ty * g.ts + sy;
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
   float uJitter = (r1 - .5f) / (float)g.res_w;
   float vJitter = (r2 - .5f) / (float)g.res_h;
   float u = 2.0f * ((float)x / (float)g.res_w) - 1.0f + uJitter;
   float v = 2.0f * ((float)y / (float)g.res_h) - 1.0f + vJitter;
   F3 rPos = (u * .5f) * cam.right + (v * .5f) * cam.up + cam.pos;
   F3 rDir = rPos - cam.pos + cam.look_dir;
   Ray r; r.o = rPos; r.d = rDir;
   return r;
}

/* ======================================================================= */
/* kernels                                                                  */
/* ======================================================================= */

/* generate: the primary ray of every sample slot of the batch (slots outside the frame are invalid) */
__global__ __launch_bounds__(RT_BLOCK) void k_generate(DCamera cam, DGrid g, long long n, QRay *queue, Hit *hits,
                                                       unsigned long long *stats)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   int X = 0, Y = 0;
   bool valid = i < n && slot_to_sample(g, i, X, Y);
   if (i < n) {
      Ray r; r.o = mk3(0, 0, 0); r.d = mk3(0, 0, 1);
      if (valid) r = camera_ray(cam, g, X, Y);
      store_qray(queue, (unsigned long long)i, r.o, r.d, -FLT_MAX, valid ? 1 : 0);
      if (!valid) { Hit h; h.prim = -1; h.t = FLT_MAX; hits[i] = h; }
   }
   stat_count(stats, ST_PRIMARY, valid);
}

/* shadow set-up: the feeler of (light = blockIdx.y, node = x) into the shadow queue at light*n + node */
__global__ __launch_bounds__(RT_BLOCK) void k_shadow_setup(DScene sc, const QRay *queue, const Hit *hits, long long n,
                                                           QRay *shadow_queue, unsigned char *flags,
                                                           unsigned long long *stats)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   int light = blockIdx.y;
   bool valid = false, skipped = false;
   if (i < n) {
      unsigned long long out = (unsigned long long)light * (unsigned long long)n + (unsigned long long)i;
      Hit h = hits[i];
      F3 lp = mk3(0, 0, 0), fd = mk3(0, 0, 1);
      float param = -1.0f;
      if (h.prim >= 0) {
         Ray ray = load_qray(queue, i);
         F3 P = ray_point(ray, h.t);                                   /* kernel.cu:340 */
         const rt_dev_light &L = sc.lights[light];
         lp = mk3(L.pos[0], L.pos[1], L.pos[2]);
         fd = P - lp;                                                  /* PointLight.h:18-20 */
         Ray feeler; feeler.o = lp; feeler.d = fd;
         param = any_intersect(sc, h.prim, feeler);                    /* kernel.cu:351 */
         /* A feeler that misses its own surface (-1) can never be reported shadowed:
          * isFloatLessThan(closest, -1) is false for every accepted t.  The reference still
          * walks the tree for it; that walk is skipped here and counted separately. */
         valid = param != -1.0f;
         skipped = !valid;
      }
      store_qray(shadow_queue, out, lp, fd, param, valid ? 1 : 0);
      if (!valid) flags[out] = 0;
   }
   stat_count(stats, ST_SHADOW_TRACED, valid);
   stat_count(stats, ST_SHADOW_SKIPPED, skipped);
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

/* spawn: the reflection / refraction rays of every hit of a level (getReflection / getRefraction,
 * kernel.cu:209-276, guarded as in kernel.cu:361-371) into the next level's queue.  It needs no shadow
 * result, so it runs right behind the level's extend while the shadow rays are traced elsewhere. */
__global__ __launch_bounds__(RT_BLOCK) void k_spawn(DScene sc, const QRay *queue, const Hit *hits, long long n, int spawn,
                                                    SpawnRec *recs, QRay *next_queue, unsigned *next_count,
                                                    unsigned next_cap, unsigned long long *stats)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   bool want_refl = false, want_refr = false, shaded = false;
   Ray reflRay, refrRay;
   reflRay.o = reflRay.d = refrRay.o = refrRay.d = mk3(0, 0, 0);
   SpawnRec out;
   out.refl = out.alpha = 0.0f; out.child_refl = out.child_refr = -1;
   if (i < n) {
      Hit h = hits[i];
      if (h.prim >= 0) {
         shaded = true;
         unsigned mat_id = h.prim < sc.n_prims ? sc.prims[h.prim].material : sc.planes[h.prim - sc.n_prims].material;
         const rt_material &m = sc.materials[mat_id];
         out.refl = m.refl; out.alpha = m.alpha;
         if (spawn && (m.refl > 0.0f || m.refr > 0.0f)) {
            ShadePoint sp = shade_point(sc, queue, i, h);
            if (sp.m.refl > 0.0f) {
               /* getReflection, kernel.cu:214-216 */
               reflRay.d = (2.0f * dot3(sp.normal, sp.eye)) * sp.normal - sp.eye;
               reflRay.o = sp.P + RT_BIG_EPS * reflRay.d;
               want_refl = true;
            }
            if (sp.m.refr > 0.0f) {
               /* getRefraction, kernel.cu:239-257 */
               float n1, n2; F3 refrNorm; F3 dd = -sp.eye;
               if (less_than(dot3(sp.eye, sp.normal), 0.0f)) { n1 = sp.m.ior; n2 = RT_AIR_IOR; refrNorm = -sp.normal; }
               else { n1 = RT_AIR_IOR; n2 = sp.m.ior; refrNorm = sp.normal; }
               float dDotN = dot3(dd, refrNorm);
               float nr = n1 / n2;
               float disc = 1.0f - nr * nr * (1.0f - dDotN * dDotN);
               if (disc > 0.0f) {
                  refrRay.d = nr * (dd - refrNorm * dDotN) - refrNorm * sqrtf(disc);
                  refrRay.o = sp.P + RT_BIG_EPS * refrRay.d;
                  want_refr = true;
               }
            }
         }
      }
   }
   stat_count(stats, ST_SHADE_NODES, shaded);
   if (spawn) {
      /* wave-aggregated queue allocation: the wave's reflection rays land contiguously, then its refraction rays */
      unsigned long long mR = __ballot(want_refl), mT = __ballot(want_refr);
      unsigned cR = __popcll(mR), cT = __popcll(mT);
      if (cR + cT) {
         unsigned lane = __lane_id();
         unsigned leader = (unsigned)__ffsll((long long)(mR | mT)) - 1;
         unsigned base = 0;
         if (lane == leader) base = atomicAdd(next_count, cR + cT);
         base = __shfl(base, leader);
         unsigned long long below = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
         if (want_refl) {
            unsigned slot = base + __popcll(mR & below);
            if (slot < next_cap) { store_qray(next_queue, slot, reflRay.o, reflRay.d, -FLT_MAX, 1); out.child_refl = (int)slot; }
         }
         if (want_refr) {
            unsigned slot = base + cR + __popcll(mT & below);
            if (slot < next_cap) { store_qray(next_queue, slot, refrRay.o, refrRay.d, -FLT_MAX, 1); out.child_refr = (int)slot; }
         }
         if (lane == leader) {
            stat_add(stats, ST_REFLECT, (unsigned long long)cR);
            stat_add(stats, ST_REFRACT, (unsigned long long)cT);
         }
      }
   }
   if (i < n) *reinterpret_cast<float4 *>(recs + i) = make_float4(out.refl, out.alpha, __int_as_float(out.child_refl), __int_as_float(out.child_refr));
}

/* light: totalLight of every hit of a level, the loop of kernel.cu:347-358 with the shadow flags of the level */
__global__ __launch_bounds__(RT_BLOCK) void k_light(DScene sc, const QRay *queue, const Hit *hits, const unsigned char *flags,
                                                    long long n, int shading, float *totals)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= n) return;
   Hit h = hits[i];
   F3 total = mk3(0.0f, 0.0f, 0.0f);
   if (h.prim >= 0) {
      ShadePoint sp = shade_point(sc, queue, i, h);
      F3 matClr = mk3(sp.m.clr[0], sp.m.clr[1], sp.m.clr[2]);      /* getColor, untextured (kernel.cu:323-326) */
      for (int l = 0; l < sc.n_lights; l++) {
         const rt_dev_light &L = sc.lights[l];
         F3 lc = mk3(L.clr[0], L.clr[1], L.clr[2]);
         F3 lightDir = normalize3(mk3(L.pos[0], L.pos[1], L.pos[2]) - sp.P);    /* PointLight.h:14-16 */
         bool inShadow = flags[(long long)l * n + i] != 0;
         F3 c = shading == RT_COOK_TORRANCE
                   ? cook_torrance(matClr, sp.m.amb, sp.m.dif, sp.m.spec, sp.m.rough, sp.eye, lightDir, lc, sp.normal, inShadow)
                   : phong(matClr, sp.m.amb, sp.m.dif, sp.m.spec, sp.m.rough, sp.eye, lightDir, lc, sp.normal, inShadow);
         total = total + c;
      }
   }
   totals[3 * i] = total.x; totals[3 * i + 1] = total.y; totals[3 * i + 2] = total.z;
}

/* combine: colour of node i of a level from its own light sum and its children's colours
 * (kernel.cu:375-376).  child colours of a missing ray are the reference's vec3(0). */
__global__ __launch_bounds__(RT_BLOCK) void k_combine(const SpawnRec *recs, const float *totals, const float *child_rgb,
                                                      long long n, float *rgb, int clamp_out)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= n) return;
   float4 rr = *reinterpret_cast<const float4 *>(recs + i);
   float w_refl = rr.x, w_alpha = rr.y;
   int child_refl = __float_as_int(rr.z), child_refr = __float_as_int(rr.w);
   F3 refl = mk3(0.0f, 0.0f, 0.0f), refr = mk3(0.0f, 0.0f, 0.0f);
   if (child_refl >= 0) refl = mk3(child_rgb[3LL * child_refl], child_rgb[3LL * child_refl + 1], child_rgb[3LL * child_refl + 2]);
   if (child_refr >= 0) refr = mk3(child_rgb[3LL * child_refr], child_rgb[3LL * child_refr + 1], child_rgb[3LL * child_refr + 2]);
   F3 total = mk3(totals[3 * i], totals[3 * i + 1], totals[3 * i + 2]);
   F3 c = total * (1.0f - w_refl - w_alpha) + w_refl * refl + w_alpha * refr;
   if (clamp_out) {
      /* rayTrace's clamp (kernel.cu:537-539); NaN passes through clamp unchanged (Util.h:28-30) */
      c = mk3(clamp01(c.x), clamp01(c.y), clamp01(c.z));
   }
   rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
}

__device__ __forceinline__ unsigned char to_byte(float v)
{
   /* float -> unsigned char truncates (kernel.cu:568); NaN becomes 0 */
   if (!(v == v)) return 0;
   int k = (int)v;
   return (unsigned char)(k < 0 ? 0 : k > 255 ? 255 : k);
}

/* resolve: averageBufferColors, kernel.cu:545-570.  Thread = pixel of a batch tile. */
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(DGrid g, const float *rgb, int n_batch_tiles, int out_layout,
                                                      uchar4 *out)
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
   if (x >= g.width || y >= g.height) return;
   F3 end = mk3(0.0f, 0.0f, 0.0f);
   long long tile_base = (long long)k * g.ts * g.ts;
   for (int xo = 0; xo < g.sroot; xo++) {
      for (int yo = 0; yo < g.sroot; yo++) {
         int sx = lx * g.sroot + xo, sy = ly * g.sroot + yo;
         long long slot = tile_base + ((long long)((sy >> 3) * g.patches_per_row + (sx >> 3)) << 6) + ((sy & 7) << 3) + (sx & 7);
         end = end + mk3(rgb[3 * slot], rgb[3 * slot + 1], rgb[3 * slot + 2]);
      }
   }
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
__global__ __launch_bounds__(RT_BLOCK) void k_unslot_hits(DGrid g, const Hit *hits, long long n, int *prim, float *t)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= n) return;
   int X, Y;
   if (!slot_to_sample(g, i, X, Y)) return;
   long long idx = (long long)Y * g.res_w + X;
   prim[idx] = hits[i].prim; t[idx] = hits[i].t;
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
   float ms_build_host, ms_upload;
   Arena arena;             /* per-frame queues; grown on demand, reused across frames */
   unsigned *d_counts;      /* queue counters, one per depth level */
   unsigned long long *d_next;   /* work counters of the persistent traversal launches (one per launch of a batch) */
   unsigned long long *d_stats;
   unsigned *h_counts;      /* pinned */
   unsigned long long *h_stats;
   /* Two internal streams: the bounce chain (extend -> spawn -> extend ...) is a sequence of launches each
    * bounded below by its longest ray, so it runs on `chain` while the shadow rays and
    * light sums of the levels already extended fill the machine from `side`. */
   hipStream_t chain, side;
   hipEvent_t ev_level[RT_MAX_RECURSE], ev_begin, ev_side_done, ev_end;
   bool streams_ok;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static unsigned long long sum_stat(const unsigned long long *h, int which)
{
   unsigned long long t = 0;
   for (int r = 0; r < RT_STAT_REPL; r++) t += h[which * RT_STAT_REPL + r];
   return t;
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
   if (s->d_counts) (void)hipFree(s->d_counts);
   if (s->d_next) (void)hipFree(s->d_next);
   if (s->d_stats) (void)hipFree(s->d_stats);
   if (s->h_counts) (void)hipHostFree(s->h_counts);
   if (s->h_stats) (void)hipHostFree(s->h_stats);
   if (s->streams_ok) {
      (void)hipStreamDestroy(s->chain); (void)hipStreamDestroy(s->side);
      for (int l = 0; l < RT_MAX_RECURSE; l++) (void)hipEventDestroy(s->ev_level[l]);
      (void)hipEventDestroy(s->ev_begin); (void)hipEventDestroy(s->ev_side_done); (void)hipEventDestroy(s->ev_end);
   }
   rt_flat_scene_free(&s->host);
   delete s;
}

extern "C" int rt_scene_create_host(const rt_scene_desc *desc, rt_scene **out)
{
   if (!desc || !out) { rt_set_error("rt_scene_create_host: null argument"); return RT_ERR_BAD_ARGUMENT; }
   *out = nullptr;
   rt_scene *s = new (std::nothrow) rt_scene();
   if (!s) return RT_ERR_BAD_ARGUMENT;
   memset(&s->host, 0, sizeof s->host);
   memset(&s->d, 0, sizeof s->d);
   s->device = -1; s->flags = 0; s->streams_ok = false;
   s->dev_blob = nullptr; s->dev_bytes = 0; s->d_counts = nullptr; s->d_next = nullptr; s->d_stats = nullptr; s->h_counts = nullptr; s->h_stats = nullptr;
   s->camera = desc->camera; s->has_camera = desc->has_camera; s->ms_upload = 0;
   auto t0 = std::chrono::steady_clock::now();
   int st = rt_flatten_scene(desc, &s->host);
   if (st != RT_OK) { delete s; return st; }
   s->ms_build_host = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
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
   s->device = device; s->flags = flags; s->streams_ok = false;
   s->dev_blob = nullptr; s->d_counts = nullptr; s->d_next = nullptr; s->d_stats = nullptr; s->h_counts = nullptr; s->h_stats = nullptr;
   s->camera = desc->camera; s->has_camera = desc->has_camera;

   auto t0 = std::chrono::steady_clock::now();
   int st = rt_flatten_scene(desc, &s->host);
   if (st != RT_OK) { delete s; return st; }
   auto t1 = std::chrono::steady_clock::now();
   s->ms_build_host = std::chrono::duration<float, std::milli>(t1 - t0).count();

   const rt_flat_scene &h = s->host;
   size_t off[8], sizes[7] = {
      sizeof(rt_bvh_node) * (size_t)h.n_nodes, sizeof(rt_dev_prim) * (size_t)h.n_prims,
      sizeof(rt_dev_xform) * (size_t)h.n_prims, sizeof(rt_dev_shade) * (size_t)h.n_prims,
      sizeof(rt_dev_plane) * (size_t)h.n_planes, sizeof(rt_material) * (size_t)h.n_materials,
      sizeof(rt_dev_light) * (size_t)h.n_lights};
   off[0] = 0;
   for (int k = 0; k < 7; k++) off[k + 1] = off[k] + align_up(sizes[k] ? sizes[k] : 64, 256);
   s->dev_bytes = off[7];
   hipError_t e = hipMalloc(&s->dev_blob, s->dev_bytes);
   if (e != hipSuccess) { rt_set_error("hipMalloc(%zu) failed: %s", s->dev_bytes, hipGetErrorString(e)); rt_scene_destroy(s); return RT_ERR_HIP; }
   const void *src[7] = {h.nodes, h.prims, h.xforms, h.shades, h.planes, h.materials, h.lights};
   for (int k = 0; k < 7; k++) {
      if (!sizes[k]) continue;
      e = hipMemcpy((char *)s->dev_blob + off[k], src[k], sizes[k], hipMemcpyHostToDevice);
      if (e != hipSuccess) { rt_set_error("scene upload failed: %s", hipGetErrorString(e)); rt_scene_destroy(s); return RT_ERR_HIP; }
   }
   char *b = (char *)s->dev_blob;
   s->d.nodes = (const rt_bvh_node *)(b + off[0]); s->d.prims = (const rt_dev_prim *)(b + off[1]);
   s->d.xforms = (const rt_dev_xform *)(b + off[2]); s->d.shades = (const rt_dev_shade *)(b + off[3]);
   s->d.planes = (const rt_dev_plane *)(b + off[4]); s->d.materials = (const rt_material *)(b + off[5]);
   s->d.lights = (const rt_dev_light *)(b + off[6]);
   s->d.n_prims = h.n_prims; s->d.n_planes = h.n_planes; s->d.n_lights = h.n_lights; s->d.root = h.root;
   s->d.fast_ok = h.bounds_in_range;

   if (hipMalloc((void **)&s->d_counts, 64 * sizeof(unsigned)) != hipSuccess ||
       hipMalloc((void **)&s->d_next, 64 * sizeof(unsigned long long)) != hipSuccess ||
       hipMalloc((void **)&s->d_stats, ST_COUNT * RT_STAT_REPL * sizeof(unsigned long long)) != hipSuccess ||
       hipHostMalloc((void **)&s->h_counts, 64 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess ||
       hipHostMalloc((void **)&s->h_stats, ST_COUNT * RT_STAT_REPL * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
      rt_set_error("allocation of queue counters failed"); rt_scene_destroy(s); return RT_ERR_HIP;
   }
   {
      bool ok = hipStreamCreateWithFlags(&s->chain, hipStreamNonBlocking) == hipSuccess;
      ok = ok && hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) == hipSuccess;
      for (int l = 0; l < RT_MAX_RECURSE && ok; l++) ok = hipEventCreateWithFlags(&s->ev_level[l], hipEventDisableTiming) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&s->ev_begin, hipEventDisableTiming) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&s->ev_side_done, hipEventDisableTiming) == hipSuccess;
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
   info->ms_build_host = s->ms_build_host; info->ms_upload = s->ms_upload;
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

/* offsets into the arena (the arena may be re-allocated between levels) */
struct LevelBufs {
   long long n = 0;
   size_t queue = 0;      /* QRay[n], level >= 1 */
   size_t hits = 0;       /* Hit[n] */
   size_t recs = 0;       /* SpawnRec[n] */
   size_t totals = 0;     /* float[3n]: light sums */
   size_t rgb = 0;        /* float[3n] */
   size_t slow = 0;       /* unsigned[n]: rays set aside for the literal-arithmetic kernel */
   bool has_queue = false;
};

struct Timer {
   hipEvent_t a, b;
   bool ok = false, ended = false;
   float *acc = nullptr;
};

struct EventPool {
   std::deque<Timer> timers;   /* a deque: begin() hands out references that must survive later begin()s */
   ~EventPool()
   {
      for (auto &t : timers) if (t.ok) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
   }
   Timer &begin(hipStream_t st, float *acc)
   {
      Timer t; t.acc = acc;
      t.ok = hipEventCreate(&t.a) == hipSuccess && hipEventCreate(&t.b) == hipSuccess;
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
   return v > 0 ? v : (long long)(1 << 24);
}

static int trace_lds_bytes(const rt_scene *s, size_t *out)
{
   /* the stack holds at most one deferred sibling per internal level */
   size_t lds_bytes = sizeof(StackEntry) * (size_t)RT_BLOCK * (size_t)(s->host.tree_depth > 0 ? s->host.tree_depth : 1);
   if (lds_bytes > 160 * 1024) { rt_set_error("BVH depth %d needs %zu bytes of LDS", s->host.tree_depth, lds_bytes); return RT_ERR_TOO_LARGE; }
   if (lds_bytes > 64 * 1024) {
      HIP_TRY(hipFuncSetAttribute((const void *)k_trace<OUT_HIT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      HIP_TRY(hipFuncSetAttribute((const void *)k_trace<OUT_HIT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      HIP_TRY(hipFuncSetAttribute((const void *)k_trace<OUT_FLAG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      HIP_TRY(hipFuncSetAttribute((const void *)k_trace<OUT_FLAG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      HIP_TRY(hipFuncSetAttribute((const void *)k_trace_slow<OUT_HIT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      HIP_TRY(hipFuncSetAttribute((const void *)k_trace_slow<OUT_HIT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      HIP_TRY(hipFuncSetAttribute((const void *)k_trace_slow<OUT_FLAG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      HIP_TRY(hipFuncSetAttribute((const void *)k_trace_slow<OUT_FLAG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
   }
   *out = lds_bytes;
   return RT_OK;
}

static int g_trace_blocks = 0;

/* persistent traversal launch: enough blocks to fill the chip a few times over; every wave pulls rays until
 * the queue is dry.  `next` is this launch's zeroed work counter, next[1] (as unsigned) the slow-ray count;
 * slow_list has room for n indices. */
static void launch_trace(const rt_scene *s, bool flag_out, bool count, const QRay *rays, unsigned long long n, Hit *hits,
                         unsigned char *flags, unsigned long long *next, unsigned *slow_list, size_t lds_bytes,
                         hipStream_t stream)
{
   if (g_trace_blocks == 0) {
      const char *e = getenv("RT_TRACE_BLOCKS");
      int cus = 256;
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
      g_trace_blocks = e && atoi(e) > 0 ? atoi(e) : cus * 8;
   }
   unsigned long long want = (n + RT_BLOCK - 1) / RT_BLOCK;
   unsigned blocks = (unsigned)(want < (unsigned long long)g_trace_blocks ? want : (unsigned long long)g_trace_blocks);
   if (blocks == 0) return;
   unsigned slow_blocks = blocks < 128 ? blocks : 128;
   int base = flag_out ? ST_NODE_VISITS_SHADOW : ST_NODE_VISITS;
   unsigned *slow_count = reinterpret_cast<unsigned *>(next + 1);
#define RT_LAUNCH_TRACE(O, C)                                                                                                      \
   do {                                                                                                                            \
      hipLaunchKernelGGL((k_trace<O, C>), dim3(blocks), dim3(RT_BLOCK), lds_bytes, stream, s->d, rays, n, hits, flags, next,       \
                         slow_list, slow_count, s->d_stats, base);                                                                 \
      hipLaunchKernelGGL((k_trace_slow<O, C>), dim3(slow_blocks), dim3(RT_BLOCK), lds_bytes, stream, s->d, rays, hits, flags,      \
                         (const unsigned *)slow_list, (const unsigned *)slow_count, s->d_stats, base);                             \
   } while (0)
   if (flag_out) { if (count) RT_LAUNCH_TRACE(OUT_FLAG, true); else RT_LAUNCH_TRACE(OUT_FLAG, false); }
   else { if (count) RT_LAUNCH_TRACE(OUT_HIT, true); else RT_LAUNCH_TRACE(OUT_HIT, false); }
#undef RT_LAUNCH_TRACE
}

/* The frame driver: body of launch_kernel after scene set-up (kernel.cu:735-760), over the
 * tiles this call selects, in batches of tiles that bound the queue memory. */
static int render_impl(const rt_scene *cs, const rt_render_params *p, int view, void *out_dev, int *prim_dev,
                       float *t_dev, float *rgb_dev, hipStream_t stream, rt_stats *stats_out)
{
   rt_scene *s = const_cast<rt_scene *>(cs);   /* the arena and counters are scratch state of the scene */
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
   g.jitter = p->jitter; g.seed = p->seed;
   DCamera cam = make_camera(s->camera);
   const DScene &sc = s->d;
   const int L = sc.n_lights;
   const bool count = (s->flags & RT_SCENE_COUNTERS) != 0;

   long long per_tile = (long long)g.ts * g.ts;
   /* a batch is bounded by its sample slots and by its shadow rays (one 33-byte queue entry per light per node) */
   long long batch_slots = max_batch_slots();
   if (L > 0 && batch_slots * L > (1LL << 28)) batch_slots = (1LL << 28) / L;
   long long tiles_per_batch = batch_slots / per_tile;
   if (tiles_per_batch < 1) tiles_per_batch = 1;
   size_t lds_bytes = 0;
   { int rc = trace_lds_bytes(s, &lds_bytes); if (rc != RT_OK) return rc; }

   /* the caller's stream brackets the frame; the work itself runs on the two internal streams */
   hipStream_t M = s->chain, H = s->side;
   { const char *e = getenv("RT_SINGLE_STREAM"); if (e && atoi(e)) H = M; }
   EventPool ev;
   Timer &t_total = ev.begin(stream, &st.ms_total);
   HIP_TRY(hipEventRecord(s->ev_begin, stream));
   HIP_TRY(hipStreamWaitEvent(M, s->ev_begin, 0));
   HIP_TRY(hipMemsetAsync(s->d_stats, 0, ST_COUNT * RT_STAT_REPL * sizeof(unsigned long long), M));

   for (long long k0 = 0; k0 < n_sel; k0 += tiles_per_batch) {
      long long nb = n_sel - k0 < tiles_per_batch ? n_sel - k0 : tiles_per_batch;
      g.k0 = (int)k0;
      LevelBufs lv[RT_MAX_RECURSE];
      lv[0].n = nb * per_tile;

      /* Arena: a bump allocator re-used by every frame.  Level l+1's queue is carved for the worst case
       * (two rays per node of level l) and trimmed to the real count once that is known; everything else
       * of a batch stays allocated until the batch is done, because the side stream works behind the chain. */
      size_t used = 0;
      auto carve = [&](size_t bytes) { size_t o = used; used += align_up(bytes ? bytes : 1, 256); return o; };
      auto grow = [&](size_t keep) -> int {
         if (used <= s->arena.cap) return RT_OK;
         HIP_TRY(hipDeviceSynchronize());
         Arena bigger;
         int rc = arena_reserve(bigger, used * 2);
         if (rc != RT_OK) return rc;
         if (keep && s->arena.base) HIP_TRY(hipMemcpy(bigger.base, s->arena.base, keep, hipMemcpyDeviceToDevice));
         if (s->arena.base) HIP_TRY(hipFree(s->arena.base));
         s->arena = bigger;
         return RT_OK;
      };
      auto carve_level = [&](LevelBufs &b) {
         b.hits = carve(sizeof(Hit) * (size_t)b.n);
         b.recs = carve(sizeof(SpawnRec) * (size_t)b.n);
         b.totals = carve(sizeof(float) * 3 * (size_t)b.n);
         b.rgb = carve(sizeof(float) * 3 * (size_t)b.n);
         b.slow = carve(sizeof(unsigned) * (size_t)b.n);
      };
      if (k0 > 0) HIP_TRY(hipDeviceSynchronize());   /* the previous batch still owns the arena */
      lv[0].queue = carve(sizeof(QRay) * (size_t)lv[0].n); lv[0].has_queue = true;
      carve_level(lv[0]);
      { int rc = grow(0); if (rc != RT_OK) return rc; }
#define AT(T, off) ((T *)(s->arena.base + (off)))

      HIP_TRY(hipMemsetAsync(s->d_counts, 0, 64 * sizeof(unsigned), M));
      HIP_TRY(hipMemsetAsync(s->d_next, 0, 64 * sizeof(unsigned long long), M));
      int n_trace = 0;   /* index of the next unused work counter */

      {  /* level 0: generate the primary rays, then their closest hits */
         Timer &tg = ev.begin(M, &st.ms_generate);
         hipLaunchKernelGGL(k_generate, dim3(grid_for(lv[0].n)), dim3(RT_BLOCK), 0, M, cam, g, lv[0].n, AT(QRay, lv[0].queue),
                            AT(Hit, lv[0].hits), s->d_stats);
         ev.end(tg, M);
         Timer &tm = ev.begin(M, &st.ms_extend);
         launch_trace(s, false, count, AT(QRay, lv[0].queue), (unsigned long long)lv[0].n, AT(Hit, lv[0].hits), nullptr,
                      s->d_next + 2 * n_trace++, AT(unsigned, lv[0].slow), lds_bytes, M);
         ev.end(tm, M); st.launches_extend++;
         HIP_TRY(hipEventRecord(s->ev_level[0], M));
      }

      int last = 0;
      if (view != VIEW_PRIMARY) {
         for (int l = 0; l < max_rec; l++) {
            last = l;
            long long n = lv[l].n;
            int spawn = (l + 1 < max_rec) ? 1 : 0;
            size_t live = used;           /* everything below is still needed */
            size_t nextq_off = 0;
            unsigned next_cap = 0;
            if (spawn) {
               long long cap = 2 * n;
               if (cap > 0x7fffffffLL) { rt_set_error("ray queue too large"); return RT_ERR_TOO_LARGE; }
               next_cap = (unsigned)cap;
               nextq_off = carve(sizeof(QRay) * (size_t)next_cap);
            }
            { int rc = grow(live); if (rc != RT_OK) return rc; }
            const QRay *queue = AT(QRay, lv[l].queue);
            {  /* chain: the next level's rays */
               Timer &tm = ev.begin(M, &st.ms_shade);
               hipLaunchKernelGGL(k_spawn, dim3(grid_for(n)), dim3(RT_BLOCK), 0, M, sc, queue, (const Hit *)AT(Hit, lv[l].hits), n,
                                  spawn, AT(SpawnRec, lv[l].recs), spawn ? AT(QRay, nextq_off) : (QRay *)nullptr,
                                  s->d_counts + (l + 1), next_cap, s->d_stats);
               ev.end(tm, M);
            }
            long long n_next = 0;
            if (spawn) {
               HIP_TRY(hipMemcpyAsync(s->h_counts, s->d_counts + (l + 1), sizeof(unsigned), hipMemcpyDeviceToHost, M));
               HIP_TRY(hipStreamSynchronize(M));
               n_next = s->h_counts[0];
               if (n_next > next_cap) n_next = next_cap;
               /* keep the spawned rays, release the unused queue tail */
               used = nextq_off + align_up(sizeof(QRay) * (size_t)(n_next ? n_next : 1), 256);
            }
            live = used;
            /* side: shadow rays and light sums of this level, behind its extend */
            unsigned long long n_shadow = (unsigned long long)(L > 0 ? L : 0) * (unsigned long long)n;
            size_t flags_off = carve((size_t)(n_shadow ? n_shadow : 1));
            size_t shq_off = carve(sizeof(QRay) * (size_t)(n_shadow ? n_shadow : 1));
            size_t shslow_off = carve(sizeof(unsigned) * (size_t)(n_shadow ? n_shadow : 1));
            if (n_next > 0) {
               lv[l + 1].n = n_next;
               lv[l + 1].queue = nextq_off; lv[l + 1].has_queue = true;
               carve_level(lv[l + 1]);
            }
            { int rc = grow(live); if (rc != RT_OK) return rc; }
            queue = AT(QRay, lv[l].queue);
            if (n_next > 0) {
               /* chain first: the next extend is on the critical path */
               Timer &tm = ev.begin(M, &st.ms_extend);
               launch_trace(s, false, count, AT(QRay, lv[l + 1].queue), (unsigned long long)n_next, AT(Hit, lv[l + 1].hits), nullptr,
                            s->d_next + 2 * n_trace++, AT(unsigned, lv[l + 1].slow), lds_bytes, M);
               ev.end(tm, M); st.launches_extend++;
               HIP_TRY(hipEventRecord(s->ev_level[l + 1], M));
            }
            HIP_TRY(hipStreamWaitEvent(H, s->ev_level[l], 0));
            if (L > 0 && n > 0) {
               Timer &ts = ev.begin(H, &st.ms_shadow_setup);
               dim3 grid(grid_for(n), (unsigned)L);
               hipLaunchKernelGGL(k_shadow_setup, grid, dim3(RT_BLOCK), 0, H, sc, queue, (const Hit *)AT(Hit, lv[l].hits), n,
                                  AT(QRay, shq_off), AT(unsigned char, flags_off), s->d_stats);
               ev.end(ts, H);
               Timer &tm = ev.begin(H, &st.ms_shadow);
               launch_trace(s, true, count, AT(QRay, shq_off), n_shadow, nullptr, AT(unsigned char, flags_off),
                            s->d_next + 2 * n_trace++, AT(unsigned, shslow_off), lds_bytes, H);
               ev.end(tm, H); st.launches_shadow++;
            }
            {
               Timer &tm = ev.begin(H, &st.ms_light);
               hipLaunchKernelGGL(k_light, dim3(grid_for(n)), dim3(RT_BLOCK), 0, H, sc, queue, (const Hit *)AT(Hit, lv[l].hits),
                                  (const unsigned char *)AT(unsigned char, flags_off), n, p->shading, AT(float, lv[l].totals));
               ev.end(tm, H);
            }
            if (n_next == 0) break;
         }
         /* bottom-up combine once both streams are through */
         HIP_TRY(hipEventRecord(s->ev_side_done, H));
         HIP_TRY(hipStreamWaitEvent(M, s->ev_side_done, 0));
         Timer &tc = ev.begin(M, &st.ms_combine);
         for (int l = last; l >= 0; l--) {
            const float *child = (l < last) ? AT(float, lv[l + 1].rgb) : nullptr;
            /* the deepest level has no children: every child index is -1 there */
            hipLaunchKernelGGL(k_combine, dim3(grid_for(lv[l].n)), dim3(RT_BLOCK), 0, M, (const SpawnRec *)AT(SpawnRec, lv[l].recs),
                               (const float *)AT(float, lv[l].totals), child, lv[l].n, AT(float, lv[l].rgb), l == 0 ? 1 : 0);
         }
         ev.end(tc, M);
      }

      if (view == VIEW_FRAME) {
         Timer &tr = ev.begin(M, &st.ms_resolve);
         long long npx = nb * RT_TILE * RT_TILE;
         hipLaunchKernelGGL(k_resolve, dim3(grid_for(npx)), dim3(RT_BLOCK), 0, M, g, (const float *)AT(float, lv[0].rgb), (int)nb,
                            p->out_layout, (uchar4 *)out_dev);
         ev.end(tr, M);
      } else if (view == VIEW_PRIMARY) {
         hipLaunchKernelGGL(k_unslot_hits, dim3(grid_for(lv[0].n)), dim3(RT_BLOCK), 0, M, g, (const Hit *)AT(Hit, lv[0].hits), lv[0].n,
                            prim_dev, t_dev);
      } else {
         hipLaunchKernelGGL(k_unslot_rgb, dim3(grid_for(lv[0].n)), dim3(RT_BLOCK), 0, M, g, (const float *)AT(float, lv[0].rgb), lv[0].n,
                            rgb_dev);
      }
      HIP_TRY(hipGetLastError());
   }
   HIP_TRY(hipMemcpyAsync(s->h_stats, s->d_stats, ST_COUNT * RT_STAT_REPL * sizeof(unsigned long long), hipMemcpyDeviceToHost, M));
   HIP_TRY(hipEventRecord(s->ev_end, M));
   HIP_TRY(hipStreamWaitEvent(stream, s->ev_end, 0));
   ev.end(t_total, stream);
   HIP_TRY(hipStreamSynchronize(stream));
   HIP_TRY(hipStreamSynchronize(M));
   HIP_TRY(hipStreamSynchronize(H));
   HIP_TRY(hipGetLastError());
   ev.collect();
   if (sum_stat(s->h_stats, ST_ABORTED) != 0) { rt_set_error("traversal aborted: step cap reached (corrupt scene?)"); return RT_ERR_HIP; }
   st.rays_primary = sum_stat(s->h_stats, ST_PRIMARY);
   st.rays_shadow = sum_stat(s->h_stats, ST_SHADOW_TRACED);
   st.shadow_skipped = sum_stat(s->h_stats, ST_SHADOW_SKIPPED);
   st.rays_reflect = sum_stat(s->h_stats, ST_REFLECT);
   st.rays_refract = sum_stat(s->h_stats, ST_REFRACT);
   st.shade_nodes = sum_stat(s->h_stats, ST_SHADE_NODES);
   st.rays_total = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract;
   st.samples = st.rays_primary;
   st.node_visits = sum_stat(s->h_stats, ST_NODE_VISITS); st.prim_tests = sum_stat(s->h_stats, ST_PRIM_TESTS); st.tie_events = sum_stat(s->h_stats, ST_TIES);
   st.node_visits_shadow = sum_stat(s->h_stats, ST_NODE_VISITS_SHADOW); st.prim_tests_shadow = sum_stat(s->h_stats, ST_PRIM_TESTS_SHADOW);
   st.tie_events_shadow = sum_stat(s->h_stats, ST_TIES_SHADOW);
   st.wave_node_steps = sum_stat(s->h_stats, ST_WAVE_NODE_STEPS); st.wave_leaf_steps = sum_stat(s->h_stats, ST_WAVE_LEAF_STEPS);
   st.max_ray_steps = s->h_stats[ST_MAX_RAY_STEPS * RT_STAT_REPL]; st.max_ray_steps_shadow = s->h_stats[ST_MAX_RAY_STEPS_SHADOW * RT_STAT_REPL];
   st.wave_node_steps_shadow = sum_stat(s->h_stats, ST_WAVE_NODE_STEPS_SHADOW); st.wave_leaf_steps_shadow = sum_stat(s->h_stats, ST_WAVE_LEAF_STEPS_SHADOW);
   if (stats_out) *stats_out = st;
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
   void *d_out = nullptr;
   HIP_TRY(hipMalloc(&d_out, bytes));
   hipError_t e = hipMemset(d_out, 0, bytes);
   int rc = e == hipSuccess ? render_impl(scene, &p, VIEW_FRAME, d_out, nullptr, nullptr, nullptr, nullptr, stats) : RT_ERR_HIP;
   if (rc == RT_OK) {
      e = hipMemcpy(out_host, d_out, bytes, hipMemcpyDeviceToHost);   /* kernel.cu:759-760 */
      if (e != hipSuccess) { rt_set_error("frame copy failed: %s", hipGetErrorString(e)); rc = RT_ERR_HIP; }
   }
   (void)hipFree(d_out);
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

extern "C" int rt_trace_rays(const rt_scene *cs, int64_t n, const float *rays_host, const float *to_beat_host,
                             int32_t *prim_host, float *t_host)
{
   rt_scene *s = const_cast<rt_scene *>(cs);
   if (!s || !rays_host || !prim_host || !t_host || n < 0) { rt_set_error("rt_trace_rays: bad argument"); return RT_ERR_BAD_ARGUMENT; }
   if (s->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   if (n == 0) return RT_OK;
   if (n > 0x7fffffffLL) { rt_set_error("rt_trace_rays: too many rays"); return RT_ERR_TOO_LARGE; }
   HIP_TRY(hipSetDevice(s->device));
   std::vector<QRay> q((size_t)n);
   for (int64_t i = 0; i < n; i++) {
      const float *r = rays_host + 6 * i;
      QRay e = {r[0], r[1], r[2], to_beat_host ? to_beat_host[i] : -FLT_MAX, r[3], r[4], r[5], 1};
      q[(size_t)i] = e;
   }
   QRay *d_q = nullptr; Hit *d_h = nullptr; unsigned *d_slow = nullptr;
   int rc = RT_OK;
   size_t lds_bytes = 0;
   rc = trace_lds_bytes(s, &lds_bytes);
   if (rc != RT_OK) return rc;
   if (hipMalloc((void **)&d_q, sizeof(QRay) * (size_t)n) != hipSuccess || hipMalloc((void **)&d_h, sizeof(Hit) * (size_t)n) != hipSuccess ||
       hipMalloc((void **)&d_slow, sizeof(unsigned) * (size_t)n) != hipSuccess) {
      rt_set_error("rt_trace_rays: hipMalloc failed"); rc = RT_ERR_HIP;
   }
   if (rc == RT_OK) {
      (void)hipMemcpy(d_q, q.data(), sizeof(QRay) * (size_t)n, hipMemcpyHostToDevice);
      (void)hipMemset(s->d_stats, 0, ST_COUNT * RT_STAT_REPL * sizeof(unsigned long long));
      (void)hipMemset(s->d_next, 0, 2 * sizeof(unsigned long long));
      launch_trace(s, false, false, d_q, (unsigned long long)n, d_h, nullptr, s->d_next, d_slow, lds_bytes, 0);
      std::vector<Hit> h((size_t)n);
      hipError_t e = hipMemcpy(h.data(), d_h, sizeof(Hit) * (size_t)n, hipMemcpyDeviceToHost);
      if (e != hipSuccess) { rt_set_error("rt_trace_rays: %s", hipGetErrorString(e)); rc = RT_ERR_HIP; }
      else for (int64_t i = 0; i < n; i++) { prim_host[i] = h[(size_t)i].prim; t_host[i] = h[(size_t)i].t; }
   }
   if (d_q) (void)hipFree(d_q);
   if (d_h) (void)hipFree(d_h);
   if (d_slow) (void)hipFree(d_slow);
   return rc;
}
