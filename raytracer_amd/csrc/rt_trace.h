/*
 * rt_trace.h -- the traversal kernel of the render path (device code; included by rt_render.hip alone).
 *
 * getClosestIntersection (kernel.cu:83-200), isInShadow (kernel.cu:202-207), the bounce rays of shadeObject
 * (kernel.cu:209-276, 360-374) and generateCameraRays (kernel.cu:477-498) as one persistent wave64 launch: k_trace.
 */
#ifndef RT_TRACE_H
#define RT_TRACE_H

#include "rt_kernel_types.h"

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
 * widening its boxes carry for the size of their own coordinates, the second tree's boxes are therefore widened, when
 * the tree is built, by 2^-20 of the scene's `origin_reach` (eight times the box test's rounding for an origin that far
 * out), and a ray that starts further out than that does not use the tree.  (Each ray widening the boxes by its own
 * origin's share, in the node step, was tighter and cost 2.5 % of the frame.)
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
   if (!(fmaxf(fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fabsf(r.o.z)) <= sc.origin_reach)) c.par |= RT_PAR_NO_FILTER;
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
#ifdef RT_DEBUG_WAVES
#define RT_DBG_RAY_START dbg_ray_start = wall_clock64()
__device__ unsigned long long dbg_waves[4096 * 16];
extern "C" int rt_debug_read_waves(unsigned long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(dbg_waves), sizeof(unsigned long long) * 4096 * 16) == hipSuccess ? 0 : 1; }
#else
#define RT_DBG_RAY_START (void)0
#endif
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
   /* in front of the stacks: RT_GUIDE_WORDS words per lane for the leaves a guided walk keeps (word w of a lane is
    * guide[w * RT_BLOCK]): their number, the leaves, the largest parameter among them */
   unsigned *const guide = reinterpret_cast<unsigned *>(lds_stack) + threadIdx.x;
   StackEntry *stack = lds_stack + (RT_GUIDE_WORDS * RT_BLOCK * sizeof(unsigned)) / sizeof(StackEntry) + threadIdx.x;
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
   bool burst_guided = false;          /* wave-uniform: some lane of the current burst makes a guided walk */
   RayCtx c;
   c.r.o = mk3(0, 0, 0); c.r.d = mk3(1, 1, 1); c.yx = c.yy = c.yz = 1.0f; c.par = 0; c.fast = true;
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
   unsigned n_steps_both = 0, n_steps_second = 0;        /* COUNT only, the same in every lane: node steps with rays in both trees / in the second tree only */
   unsigned n_guided = 0;                               /* COUNT only: of n_rival, those that went on guided */
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
         else {
            /* With a rival the order of the reference's walk decides, and that walk is made.  If the second tree has shown
             * the ray every primitive that can take part in the decision -- a few, all within RT_GUIDE_NEAR of the
             * nearest, while everything else the ray meets is at least RT_LOOK_ABS - RT_SLACK_MAX behind the nearest --
             * the walk may leave out the subtrees that hold none of them.  What it would do in there changes nothing:
             * a primitive behind the origin is never accepted (t stays above 2 EQUAL_EPSILON); one of the far ones is
             * accepted only while the walk holds nothing or another far one, ties with none of the kept ones and loses to
             * each of them; and while the walk holds a far one it enters the same boxes with kept leaves as while it
             * holds nothing, since those boxes start in front of the kept leaves' parameters (the ray starts outside
             * every box of the tree: it starts outside the root's two, and the boxes are nested). */
            bool guided = false;
            if (A()->guide && t > 2.0f * RT_EQ_EPS) {
               const unsigned kept = guide[0];
               const float farthest = __uint_as_float(guide[(RT_GUIDE_K + 1) * RT_BLOCK]);
               const bool outside = !(c.r.o.x > A()->tree_lo[0] && c.r.o.x < A()->tree_hi[0] && c.r.o.y > A()->tree_lo[1] && c.r.o.y < A()->tree_hi[1] &&
                                      c.r.o.z > A()->tree_lo[2] && c.r.o.z < A()->tree_hi[2]);
               guided = kept <= RT_GUIDE_K && farthest <= t + RT_GUIDE_NEAR && outside;
            }
            if (COUNT && guided) n_guided++;
            mode = guided ? (MODE_EXTEND | MODE_GUIDED) : MODE_EXTEND; toBeat = -FLT_MAX; t = FLT_MAX; closest = -1; cur = sc.root; sp = 0;
         }
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
         SlabAB sl = slab_ab<VARIANT == 1>(c, na.x, na.y, na.z, na.w, nb.x, nb.y);
         SlabAB sr = slab_ab<VARIANT == 1>(c, nb.z, nb.w, nc.x, nc.y, nc.z, nc.w);
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
      if (TWO && OUT == OUT_CHAIN && burst_guided) {
         /* A guided walk: a child whose subtree holds none of the kept leaves is treated as not ahead -- never entered,
          * never deferred.  The node's record carries the range of leaves below it (the leaves are numbered in the
          * tree's order); the left child's part of it ends after the one leaf, or before the right child's one leaf, or
          * after one leaf more than the left subtree has nodes. */
         if (mode & MODE_GUIDED) {
            const int lo = tail.z, hi = tail.w;
            const int split = ch.x < 0 ? lo + 1 : (ch.y < 0 ? hi - 1 : lo + (ch.y - ch.x) + 1);
            const int kept = (int)guide[0];
            bool in_l = false, in_r = false;
#pragma unroll
            for (int k = 0; k < RT_GUIDE_K; k++) {
               const int leaf = (int)guide[(1 + k) * RT_BLOCK];
               const bool have = k < kept && leaf >= lo && leaf < hi;
               in_l = in_l || (have && leaf < split); in_r = in_r || (have && leaf >= split);
            }
            l = in_l ? l : inf; r = in_r ? r : inf;
         }
      }
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
            if ((mode & MODE_KIND) == MODE_EXTEND) {
               toBeat = FLT_MAX;      /* the nearest of the other primitives, so far */
               guide[0] = 0u; guide[(RT_GUIDE_K + 1) * RT_BLOCK] = __float_as_uint(-FLT_MAX);
            }
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
      /* A wave that holds a ray long under way is issued ahead of the other waves of its SIMD.  The launch ends with the
       * longest chains of dependent steps (a contested primary, then a feeler that walks the reference's tree in full);
       * each of their steps waits for its wave to come round again, and while the chip is full that takes several times
       * the memory latency.  Scheduling only: no result depends on it. */
      if (__ballot(active && born > RT_OLD_RAY) != 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
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
         /* (a ray that turns to a guided walk during the burst walks unguided until the next one: the same walk, with
          * visits that change nothing) */
         burst_guided = TWO && OUT == OUT_CHAIN && __ballot(active && (mode & MODE_GUIDED)) != 0;
#pragma unroll 1
         for (int burst = 0; burst < node_burst; burst++) {
            bool in_node = cur >= 0;                 /* idle lanes hold RT_CUR_IDLE */
            if (__ballot(in_node) == 0) break;
            /* enough lanes wait at a leaf: their step is worth its price now */
            if ((int)__popcll(__ballot(cur < 0 && cur != RT_CUR_IDLE)) >= leaf_trigger) break;
            if (COUNT) {
               wn++; if (active) ray_steps++;
               if (cur < 0) { if (cur == RT_CUR_IDLE) n_noray++; else n_leafwait++; }
               if (TWO) {
                  const unsigned long long in_second = __ballot(in_node && (mode & MODE_FILTER) != 0), in_first = __ballot(in_node && (mode & MODE_FILTER) == 0);
                  if (in_second && in_first) n_steps_both++; else if (in_second) n_steps_second++;
               }
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
                     /* (what a guided walk would keep: everything met within the range that is searched completely) */
                     if (dist < look_cut()) {
                        const unsigned kept = guide[0];
                        if (kept < RT_GUIDE_K) guide[(1 + kept) * RT_BLOCK] = (unsigned)~cur;
                        guide[0] = kept + 1u;
                        guide[(RT_GUIDE_K + 1) * RT_BLOCK] = __float_as_uint(fmaxf(__uint_as_float(guide[(RT_GUIDE_K + 1) * RT_BLOCK]), dist));
                     }
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
      stat_add(stats, ST_CERT_UNREACHED, (unsigned long long)n_unreached); stat_add(stats, ST_GUIDED, (unsigned long long)n_guided);
      stat_add(stats, ST_SHADOW_CERT, (unsigned long long)n_shadow_cert);
      stat_add(stats, ST_LEAF_WAIT, (unsigned long long)n_leafwait);
      stat_add(stats, ST_NO_RAY, (unsigned long long)n_noray);
      if (lane == 0) {
         stat_add(stats, ST_REFILLS, (unsigned long long)n_refills);
         stat_add(stats, ST_STEPS_BOTH, (unsigned long long)n_steps_both); stat_add(stats, ST_STEPS_SECOND_ONLY, (unsigned long long)n_steps_second);
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

#endif
