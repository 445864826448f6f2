/*
 * rt_device.h -- device-side arithmetic of the render path (gfx950, wave64).
 *
 * Everything here is IEEE binary32 with the operation order of the reference
 * (GLM 0.9.3 semantics: dot and cross left to right, normalize = v * (1/sqrt),
 * mat*vec row sums left to right) and must be compiled with -ffp-contract=off:
 * which primitive a ray reports depends on comparisons of these values, so a
 * fused multiply-add in the wrong place changes hit ids.  The only fused
 * operations are the explicit __builtin_fmaf calls in rt_div(), which are an
 * exact-division algorithm, not a contraction.
 */
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>
#include <float.h>
#include "rt_common.h"

#define RT_EPS 0.0001f          /* EPSILON, Util.h:4 */
#define RT_BIG_EPS 0.001f       /* BIG_EPSILON, Util.h:5 */
#define RT_EQ_EPS 0.01f         /* EQUAL_EPSILON, Util.h:6 */
#define RT_AIR_IOR 1.0f         /* kAirIOR, kernel.h:24 */

struct F3 { float x, y, z; };

__device__ __forceinline__ F3 mk3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ F3 operator-(F3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 x, F3 y)
{
   return mk3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
__device__ __forceinline__ F3 normalize3(F3 v)
{
   float sqr = v.x * v.x + v.y * v.y + v.z * v.z;
   float inv = 1.0f / sqrtf(sqr);     /* correctly rounded sqrt and divide (hipcc default) */
   return mk3(v.x * inv, v.y * inv, v.z * inv);
}

/* Util.h:8-42 */
__device__ __forceinline__ bool is_zero(float n) { return n < RT_EPS && n > -RT_EPS; }
__device__ __forceinline__ bool is_equal(float a, float b) { return a + RT_EQ_EPS > b && b + RT_EQ_EPS > a; }
/* n + EPSILON > 0 (Util.h:16-18).  The sum of two floats keeps the sign of the exact sum and is zero only
 * when that is zero, so this is the same predicate as n > -EPSILON, one instruction shorter. */
__device__ __forceinline__ bool above_zero(float n) { return n > -RT_EPS; }
__device__ __forceinline__ bool less_than(float l, float r) { return l + RT_EPS < r; }
__device__ __forceinline__ float clamp01(float x) { return x > 1.0f ? 1.0f : x < 0.0f ? 0.0f : x; }
__device__ __forceinline__ float my_min(float x, float y) { return x < y ? x : y; }
__device__ __forceinline__ float my_max(float x, float y) { return x > y ? x : y; }

struct Ray { F3 o, d; };
__device__ __forceinline__ F3 ray_point(const Ray &r, float t) { return r.o + r.d * t; }   /* Ray.h:7 */

/* Ray::transform (Ray.h:8-12) with the three kept rows of inv_trans.
 * Origin: m0*x + m1*y + m2*z + m3*1; direction: ... + m3*0. */
__device__ __forceinline__ Ray ray_transform(const Ray &r, const rt_dev_xform &m)
{
   Ray o;
   o.o.x = m.r[0][0] * r.o.x + m.r[0][1] * r.o.y + m.r[0][2] * r.o.z + m.r[0][3] * 1.0f;
   o.o.y = m.r[1][0] * r.o.x + m.r[1][1] * r.o.y + m.r[1][2] * r.o.z + m.r[1][3] * 1.0f;
   o.o.z = m.r[2][0] * r.o.x + m.r[2][1] * r.o.y + m.r[2][2] * r.o.z + m.r[2][3] * 1.0f;
   o.d.x = m.r[0][0] * r.d.x + m.r[0][1] * r.d.y + m.r[0][2] * r.d.z + m.r[0][3] * 0.0f;
   o.d.y = m.r[1][0] * r.d.x + m.r[1][1] * r.d.y + m.r[1][2] * r.d.z + m.r[1][3] * 0.0f;
   o.d.z = m.r[2][0] * r.d.x + m.r[2][1] * r.d.y + m.r[2][2] * r.d.z + m.r[2][3] * 0.0f;
   return o;
}

/* ---- exact division by a per-ray constant -------------------------------
 * The slab test divides (bound - origin) by the ray direction six times per
 * box (GeometryUtil.h:17-32) and the quotients are compared with each other
 * and with t, so they must be the correctly rounded a/d.  hipcc's own a/d is
 * ~10 instructions; with y = RN(1/d) computed once per ray (a true division)
 * one residual correction gives RN(a/d) in 3:
 *      q0 = RN(a*y);  r = a - d*q0 (one fma);  q = RN(q0 + r*y) (one fma).
 * q0 is within 1.5 ulp of a/d, so q0 + r*y differs from a/d by less than
 * 2^-22 ulp and the last rounding can only go wrong when a/d lies that close
 * to the midpoint of two floats.  Those operand pairs are few enough to list
 * (about 5e7 significand pairs) and every one of them is checked against a
 * true division by tests/native/exact_div_hard_cases.c.
 * Valid while a, d, the quotient and the residual stay normal; a ray whose
 * origin or direction is outside [2^-40, 2^40] (zero allowed for the origin)
 * takes the literal path below instead. */
__device__ __forceinline__ float rt_div(float a, float d, float y)
{
   float q0 = a * y;
   float r0 = __builtin_fmaf(-d, q0, a);
   return __builtin_fmaf(r0, y, q0);
}

__device__ __forceinline__ bool in_div_range(float v)
{
   float a = fabsf(v);
   return a >= 9.094947017729282e-13f && a <= 1.099511627776e12f;   /* 2^-40 .. 2^40 */
}

/* per-ray constants of the slab test */
struct RayCtx {
   Ray r;
   float yx, yy, yz;      /* RN(1/d) per axis (unused on a parallel axis) */
   unsigned par;          /* bit k < 3 set: axis k is "parallel", |d_k| < EPSILON (GeometryUtil.h:16,23,30); RT_PAR_NO_FILTER: see there */
   bool fast;             /* every value the short box test divides is inside rt_div's safe range */
};
#define RT_PAR_AXES 7u
#define RT_PAR_NO_FILTER 8u    /* this ray does not use the second tree (its margins would not cover what the primitive tests may be off by) */

__device__ __forceinline__ RayCtx make_ctx(const Ray &r)
{
   RayCtx c;
   c.r = r;
   c.yx = 1.0f / r.d.x; c.yy = 1.0f / r.d.y; c.yz = 1.0f / r.d.z;
   bool px = is_zero(r.d.x), py = is_zero(r.d.y), pz = is_zero(r.d.z);
   c.par = (px ? 1u : 0u) | (py ? 2u : 0u) | (pz ? 4u : 0u);
   /* a parallel axis is never divided by, so only the other axes' directions have to be in range */
   c.fast = (r.o.x == 0.0f || in_div_range(r.o.x)) && (r.o.y == 0.0f || in_div_range(r.o.y)) &&
            (r.o.z == 0.0f || in_div_range(r.o.z)) && (px || in_div_range(r.d.x)) && (py || in_div_range(r.d.y)) &&
            (pz || in_div_range(r.d.z));
   return c;
}

/* boxIntersect, GeometryUtil.h:9-51, literally (any ray, true divisions) */
__device__ __forceinline__ float slab_literal(const Ray &r, float mnx, float mny, float mnz,
                                              float mxx, float mxy, float mxz)
{
   bool miss = false;
   float tXmin = -FLT_MAX, tYmin = -FLT_MAX, tZmin = -FLT_MAX;
   float tXmax = FLT_MAX, tYmax = FLT_MAX, tZmax = FLT_MAX;
   if (!is_zero(r.d.x)) {
      tXmin = (mnx - r.o.x) / r.d.x;
      tXmax = (mxx - r.o.x) / r.d.x;
   } else if (r.o.x < mnx || r.o.x > mxx) miss = true;
   if (!is_zero(r.d.y)) {
      tYmin = (mny - r.o.y) / r.d.y;
      tYmax = (mxy - r.o.y) / r.d.y;
   } else if (r.o.y < mny || r.o.y > mxy) miss = true;
   if (!is_zero(r.d.z)) {
      tZmin = (mnz - r.o.z) / r.d.z;
      tZmax = (mxz - r.o.z) / r.d.z;
   } else if (r.o.z < mnz || r.o.z > mxz) miss = true;

   float xl = tXmin > tXmax ? tXmax : tXmin, xh = tXmin > tXmax ? tXmin : tXmax;
   float yl = tYmin > tYmax ? tYmax : tYmin, yh = tYmin > tYmax ? tYmin : tYmax;
   float zl = tZmin > tZmax ? tZmax : tZmin, zh = tZmin > tZmax ? tZmin : tZmax;
   float tBiggestMin = my_max(my_max(xl, yl), zl);
   float tSmallestMax = my_min(my_min(xh, yh), zh);
   bool inside = r.o.x > mnx && r.o.x < mxx && r.o.y > mny && r.o.y < mxy && r.o.z > mnz && r.o.z < mxz;
   float res = inside ? tSmallestMax : (tBiggestMin < tSmallestMax ? tBiggestMin : -1.0f);
   return miss ? -1.0f : res;
}

/* boxIntersect for a ray with in-range values: same bits, fewer instructions.
 *  - the quotient pairs of the non-parallel axes are never NaN, so min/max order them exactly as the
 *    reference's swaps do and max3/min3 equal its mymax/mymin chains;
 *  - the origin is strictly inside the box on such an axis exactly when lo < 0 < hi (a difference of
 *    two floats keeps its sign, and the quotient of an in-range pair cannot underflow), so "inside"
 *    is tBiggestMin < 0 < tSmallestMax instead of six coordinate comparisons;
 *  - PAR = true also takes rays with parallel axes: such an axis contributes the reference's
 *    -FLT_MAX / FLT_MAX pair (neutral in max3/min3), a miss when the origin is outside the slab, and
 *    its strict coordinate test to "inside".  Whatever rt_div made of the tiny or zero direction is
 *    discarded by the selects.  PAR = false is the same function for a wave that holds no such ray. */
/* the two numbers the box test turns on: where the ray enters the last of the three slabs (A) and where it leaves the
 * first (B), with the reference's treatment of parallel axes; miss / strict as described above */
struct SlabAB { float A, B; bool miss, strict; };
template <bool PAR>
__device__ __forceinline__ SlabAB slab_ab(const RayCtx &c, float mnx, float mny, float mnz, float mxx, float mxy, float mxz)
{
   const Ray &r = c.r;
   float x1 = rt_div(mnx - r.o.x, r.d.x, c.yx), x2 = rt_div(mxx - r.o.x, r.d.x, c.yx);
   float y1 = rt_div(mny - r.o.y, r.d.y, c.yy), y2 = rt_div(mxy - r.o.y, r.d.y, c.yy);
   float z1 = rt_div(mnz - r.o.z, r.d.z, c.yz), z2 = rt_div(mxz - r.o.z, r.d.z, c.yz);
   float xl = fminf(x1, x2), xh = fmaxf(x1, x2);
   float yl = fminf(y1, y2), yh = fmaxf(y1, y2);
   float zl = fminf(z1, z2), zh = fmaxf(z1, z2);
   SlabAB s; s.miss = false; s.strict = true;
   if (PAR) {
      bool px = (c.par & 1u) != 0, py = (c.par & 2u) != 0, pz = (c.par & 4u) != 0;
      xl = px ? -FLT_MAX : xl; xh = px ? FLT_MAX : xh;
      yl = py ? -FLT_MAX : yl; yh = py ? FLT_MAX : yh;
      zl = pz ? -FLT_MAX : zl; zh = pz ? FLT_MAX : zh;
      s.miss = (px && (r.o.x < mnx || r.o.x > mxx)) || (py && (r.o.y < mny || r.o.y > mxy)) ||
               (pz && (r.o.z < mnz || r.o.z > mxz));
      s.strict = (!px || (r.o.x > mnx && r.o.x < mxx)) && (!py || (r.o.y > mny && r.o.y < mxy)) &&
                 (!pz || (r.o.z > mnz && r.o.z < mxz));
   }
   s.A = fmaxf(fmaxf(xl, yl), zl);
   s.B = fminf(fminf(xh, yh), zh);
   return s;
}
/* the reference's answer from those two numbers */
__device__ __forceinline__ float slab_value(const SlabAB &s)
{
   /* (selects, not branches: `&` keeps the compiler from turning the conditions into jumps) */
   const bool inside = (s.A < 0.0f) & (s.B > 0.0f) & s.strict;
   float res = s.A < s.B ? s.A : -1.0f;
   res = inside ? s.B : res;
   return s.miss ? -1.0f : res;
}
template <bool PAR>
__device__ __forceinline__ float slab_fast(const RayCtx &c, float mnx, float mny, float mnz,
                                           float mxx, float mxy, float mxz)
{
   return slab_value(slab_ab<PAR>(c, mnx, mny, mnz, mxx, mxy, mxz));
}
/* How far behind a ray's origin a box of the second tree may end and still be entered: a primitive is accepted from
 * EPSILON behind the origin on, and what a sphere's test reports may lie RT_SLACK_MAX in front of the sphere's box. */
#define RT_SLACK_MAX 0.005f
#define RT_NEAR_SLACK (2.0f * RT_EPS + RT_SLACK_MAX)
/* The miss filter's reading of the same two numbers: could the ray meet anything inside this (widened) box at a
 * parameter the reference would accept (above -EPSILON)?  Yes unless the ray's line misses the box or the box lies
 * wholly behind that; the answer is where the ray enters it (0 from inside), for near-first order, or -1.  Unlike the
 * reference's value this never turns away a box the origin sits in or just in front of. */
__device__ __forceinline__ float slab_filter_value(const SlabAB &s)
{
   return (!s.miss && s.A <= s.B && s.B > -RT_NEAR_SLACK) ? fmaxf(s.A, 0.0f) : -1.0f;
}

/* Sphere::intersects, Sphere.h:42-66 (ray already in object space) */
__device__ __forceinline__ float sphere_hit(const Ray &ray, F3 c, float rad)
{
   F3 e = ray.o - c;
   float dDotD = dot3(ray.d, ray.d);
   float de = dot3(ray.d, e);
   float disc = de * de - dDotD * (dot3(e, e) - rad * rad);
   if (disc < 0.0f) return -1.0f;
   float nde = dot3(-ray.d, e);
   float sq = sqrtf(disc);
   float first = (nde - sq) / dDotD;
   if (less_than(0.0f, first)) return first;
   return (nde + sq) / dDotD;
}

/* Plane::intersects, Plane.h:39-50 */
__device__ __forceinline__ float plane_hit(const Ray &r, F3 n, float d)
{
   F3 c = n * d;
   float numer = dot3(-n, r.o - c);
   float denom = dot3(n, r.d);
   if (is_zero(numer) || is_zero(denom)) return -1.0f;
   float t = numer / denom;
   if (less_than(t, 0.0f)) return -1.0f;
   return t;
}

/* Triangle::intersects, Triangle.h:55-103 */
__device__ __forceinline__ float triangle_hit(const Ray &r, F3 A, F3 B, F3 C, F3 n)
{
   float numer = dot3(-n, r.o - A);
   float denom = dot3(n, r.d);
   if (is_zero(numer) || is_zero(denom)) return -1.0f;
   float t = numer / denom;
   if (less_than(t, 0.0f)) return -1.0f;
   F3 P = ray_point(r, t);
   F3 AB = B - A, AC = C - A;
   F3 N = cross3(AB, AC);
   F3 AP = P - A;
   if (dot3(N, cross3(AB, AP)) < 0.0f) return -1.0f;
   F3 BP = P - B, BC = C - B;
   if (dot3(N, cross3(BC, BP)) < 0.0f) return -1.0f;
   F3 CP = P - C;
   if (dot3(N, cross3(CP, AC)) < 0.0f) return -1.0f;
   return t;
}

#endif
