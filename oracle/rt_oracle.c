/*
 * rt_oracle.c -- CPU restatement of the reference's per-pixel render path.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Plain C99, IEEE binary32, no
 * FMA contraction (build with -ffp-contract=off and no -march flags), libm
 * sqrtf/powf/expf.  Each function names the reference lines it restates; the
 * floating-point operation ORDER is part of the restatement (GLM 0.9.3:
 * dot = x*x' + y*y' + z*z' left to right, func_geometric.inl:158-170;
 * cross, func_geometric.inl:199-211; normalize = v * (1/sqrt(dot)),
 * func_geometric.inl:239-248 + func_exponential.inl:145-153;
 * mat4*vec4 and vec4*mat4, type_mat4x4.inl:689-714).
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants (kernel.h:14-27, Util.h:4-6) ----------------------------- */
#define K_MAX_STACK 100
#define K_MAX_RECURSE 6
#define K_AIR_IOR 1.0f
#define EPSILON 0.0001f
#define BIG_EPSILON 0.001f
static const float EQUAL_EPSILON = 0.01f;

/* ---- Util.h:8-42 -------------------------------------------------------- */
static inline int isFloatZero(float n) { return n < EPSILON && n > -EPSILON; }
static inline int isFloatEqual(float a, float b) { return a + EQUAL_EPSILON > b && b + EQUAL_EPSILON > a; }
static inline int isFloatAboveZero(float n) { return n + EPSILON > 0.0f; }
static inline int isFloatLessThan(float l, float r) { return l + EPSILON < r; }
static inline float clampf(float x, float lo, float hi) { return x > hi ? hi : x < lo ? lo : x; }
static inline float mymin(float x, float y) { return x < y ? x : y; }
static inline float mymax(float x, float y) { return x > y ? x : y; }

/* ---- vectors ------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vload(const float *p) { return V(p[0], p[1], p[2]); }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }   /* vec3 * float and float * vec3 */
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }   /* type_vec3.inl:765-775 */
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 x, v3 y)
{
   return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline v3 vnormalize(v3 x)
{
   float sqr = x.x * x.x + x.y * x.y + x.z * x.z;
   float inv = 1.0f / sqrtf(sqr);
   return V(x.x * inv, x.y * inv, x.z * inv);
}
static inline float vget(v3 a, int axis) { return axis == 0 ? a.x : axis == 1 ? a.y : a.z; }

/* column-major 4x4: m[c*4 + r] */
typedef struct { float x, y, z, w; } v4;
static inline v4 mat_mul_vec(const float *m, float x, float y, float z, float w)
{
   v4 r;
   r.x = m[0] * x + m[4] * y + m[8] * z + m[12] * w;
   r.y = m[1] * x + m[5] * y + m[9] * z + m[13] * w;
   r.z = m[2] * x + m[6] * y + m[10] * z + m[14] * w;
   r.w = m[3] * x + m[7] * y + m[11] * z + m[15] * w;
   return r;
}
static inline v4 vec_mul_mat(float x, float y, float z, float w, const float *m)
{
   v4 r;
   r.x = m[0] * x + m[1] * y + m[2] * z + m[3] * w;
   r.y = m[4] * x + m[5] * y + m[6] * z + m[7] * w;
   r.z = m[8] * x + m[9] * y + m[10] * z + m[11] * w;
   r.w = m[12] * x + m[13] * y + m[14] * z + m[15] * w;
   return r;
}

/* ---- Ray.h:5-16 --------------------------------------------------------- */
typedef struct { v3 o, d; } ray_t;
static inline v3 ray_point(const ray_t *r, float param) { return vadd(r->o, vscale(r->d, param)); }
static inline ray_t ray_transform(const ray_t *r, const float *m)
{
   v4 no = mat_mul_vec(m, r->o.x, r->o.y, r->o.z, 1.0f);
   v4 nd = mat_mul_vec(m, r->d.x, r->d.y, r->d.z, 0.0f);
   ray_t out;
   out.o = V(no.x, no.y, no.z);
   out.d = V(nd.x, nd.y, nd.z);
   return out;
}

/* ---- GeometryUtil.h:9-51 ------------------------------------------------ */
static float boxIntersect(const ray_t *ray, v3 minP, v3 maxP)
{
   float tXmin, tYmin, tZmin;
   float tXmax, tYmax, tZmax;
   tXmin = tYmin = tZmin = -FLT_MAX;
   tXmax = tYmax = tZmax = FLT_MAX;

   if (!isFloatZero(ray->d.x)) {
      tXmin = (minP.x - ray->o.x) / ray->d.x;
      tXmax = (maxP.x - ray->o.x) / ray->d.x;
   } else if (ray->o.x < minP.x || ray->o.x > maxP.x) {
      return -1.0f;
   }
   if (!isFloatZero(ray->d.y)) {
      tYmin = (minP.y - ray->o.y) / ray->d.y;
      tYmax = (maxP.y - ray->o.y) / ray->d.y;
   } else if (ray->o.y < minP.y || ray->o.y > maxP.y) {
      return -1.0f;
   }
   if (!isFloatZero(ray->d.z)) {
      tZmin = (minP.z - ray->o.z) / ray->d.z;
      tZmax = (maxP.z - ray->o.z) / ray->d.z;
   } else if (ray->o.z < minP.z || ray->o.z > maxP.z) {
      return -1.0f;
   }
   if (tXmin > tXmax) { float s = tXmin; tXmin = tXmax; tXmax = s; }
   if (tYmin > tYmax) { float s = tYmin; tYmin = tYmax; tYmax = s; }
   if (tZmin > tZmax) { float s = tZmin; tZmin = tZmax; tZmax = s; }

   float tBiggestMin = mymax(mymax(tXmin, tYmin), tZmin);
   float tSmallestMax = mymin(mymin(tXmax, tYmax), tZmax);

   if (ray->o.x > minP.x && ray->o.x < maxP.x && ray->o.y > minP.y
       && ray->o.y < maxP.y && ray->o.z > minP.z && ray->o.z < maxP.z) {
      return tSmallestMax;
   } else {
      return tBiggestMin < tSmallestMax ? tBiggestMin : -1.0f;
   }
}
static inline v3 smallestCorner(v3 a, v3 b) { return V(mymin(a.x, b.x), mymin(a.y, b.y), mymin(a.z, b.z)); }
static inline v3 biggestCorner(v3 a, v3 b) { return V(mymax(a.x, b.x), mymax(a.y, b.y), mymax(a.z, b.z)); }

/* ---- BoundingBox.h:10-52 ------------------------------------------------ */
typedef struct { v3 min, max; } bbox_t;
static bbox_t bbox_make(v3 nMin, v3 nMax)   /* the constructor pads by EPSILON every time it runs */
{
   bbox_t b;
   b.min = V(nMin.x - EPSILON, nMin.y - EPSILON, nMin.z - EPSILON);
   b.max = V(nMax.x + EPSILON, nMax.y + EPSILON, nMax.z + EPSILON);
   return b;
}
static bbox_t generateBoundingBox(v3 mn, v3 mx, const float *trans)
{
   float pts[8][3] = {
      { mn.x, mn.y, mn.z }, { mx.x, mn.y, mn.z }, { mn.x, mx.y, mn.z }, { mx.x, mx.y, mn.z },
      { mn.x, mn.y, mx.z }, { mx.x, mn.y, mx.z }, { mn.x, mx.y, mx.z }, { mx.x, mx.y, mx.z } };
   float tmin[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
   float tmax[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
   for (int i = 0; i < 8; i++) {
      v4 p = mat_mul_vec(trans, pts[i][0], pts[i][1], pts[i][2], 1.0f);
      float q[3] = { p.x, p.y, p.z };
      for (int axis = 0; axis < 3; axis++) {
         if (q[axis] < tmin[axis]) tmin[axis] = q[axis];
         if (q[axis] > tmax[axis]) tmax[axis] = q[axis];
      }
   }
   return bbox_make(vload(tmin), vload(tmax));
}
static bbox_t combineBoundingBox(const bbox_t *a, const bbox_t *b)
{
   return bbox_make(smallestCorner(a->min, b->min), biggestCorner(a->max, b->max));
}

/* ---- geometry objects (Geometry.h, Sphere.h, Plane.h, Box.h, Triangle.h,
 *      SmoothTriangle.h) -------------------------------------------------- */
enum { G_SPHERE = 0, G_TRIANGLE = 1, G_BOX = 2, G_SMOOTH = 3, G_PLANE = 4 };

typedef struct geom {
   int type;
   rt_material mat;
   float trans[16], inv[16];
   /* Sphere */
   v3 c; float r; v3 wsCenter;
   /* Plane */
   v3 pn; float pd; v3 wsN;
   /* Triangle / SmoothTriangle */
   v3 p1, p2, p3, tn, tc, wsNormal;
   v3 wsN1, wsN2, wsN3;
   float vt1[2], vt2[2], vt3[2];
   /* Box */
   v3 minP, maxP;
} geom;

static void geom_init_common(geom *g, int type, const rt_modifier *mod)
{
   memset(g, 0, sizeof *g);
   g->type = type;
   g->mat = mod->mat;
   memcpy(g->trans, mod->trans, sizeof g->trans);
   memcpy(g->inv, mod->inv_trans, sizeof g->inv);
}
/* Sphere.h:10-15 */
static void sphere_init(geom *g, const rt_sphere *s)
{
   geom_init_common(g, G_SPHERE, &s->mod);
   g->c = vload(s->c);
   g->r = s->r;
   v4 wc = mat_mul_vec(g->trans, g->c.x, g->c.y, g->c.z, 1.0f);
   g->wsCenter = V(wc.x, wc.y, wc.z);
}
/* Plane.h:11-17 */
static void plane_init(geom *g, const rt_plane *p)
{
   geom_init_common(g, G_PLANE, &p->mod);
   g->pn = vload(p->n);
   g->pd = p->d;
   v4 wn = mat_mul_vec(g->trans, g->pn.x, g->pn.y, g->pn.z, 0.0f);
   g->wsN = V(wn.x, wn.y, wn.z);
}
/* Triangle.h:11-22 */
static void triangle_init(geom *g, int type, const float *p1, const float *p2, const float *p3, const rt_modifier *mod)
{
   geom_init_common(g, type, mod);
   g->p1 = vload(p1); g->p2 = vload(p2); g->p3 = vload(p3);
   g->tn = vnormalize(vcross(vsub(g->p2, g->p1), vsub(g->p3, g->p1)));
   g->tc = g->p1;
   v4 wsn = vec_mul_mat(g->tn.x, g->tn.y, g->tn.z, 0.0f, g->inv);
   g->wsNormal = V(wsn.x, wsn.y, wsn.z);
}
/* SmoothTriangle.h:12-28 */
static void smooth_init(geom *g, const rt_smooth_triangle *t)
{
   triangle_init(g, G_SMOOTH, t->p1, t->p2, t->p3, &t->mod);
   v4 a = vec_mul_mat(t->n1[0], t->n1[1], t->n1[2], 0.0f, g->inv);
   v4 b = vec_mul_mat(t->n2[0], t->n2[1], t->n2[2], 0.0f, g->inv);
   v4 c = vec_mul_mat(t->n3[0], t->n3[1], t->n3[2], 0.0f, g->inv);
   g->wsN1 = V(a.x, a.y, a.z);
   g->wsN2 = V(b.x, b.y, b.z);
   g->wsN3 = V(c.x, c.y, c.z);
}
/* Box.h:13-19 */
static void box_init(geom *g, const rt_box *b)
{
   geom_init_common(g, G_BOX, &b->mod);
   g->minP = smallestCorner(vload(b->p1), vload(b->p2));
   g->maxP = biggestCorner(vload(b->p1), vload(b->p2));
}

/* Sphere.h:42-66.  The unqualified sqrt(discriminant) at Sphere.h:55,63 has a float argument.  nvcc's device headers
 * give it the float overload, so the whole expression is binary32: that is what this restatement (and the HIP path)
 * computes.  A host C++ compile that includes <cmath> without <math.h> finds only ::sqrt(double): the square root,
 * the subtraction and the division are then done in double and rounded once.  orc_set_sphere_sqrt_double(1) selects
 * that reading; it exists only to explain ray counts taken from such a host build (a ray in a few hundred thousand
 * gets a t that differs in the last bit, and a bounce or a shadow decision with it). */
static int g_sphere_sqrt_double = 0;
void orc_set_sphere_sqrt_double(int on) { g_sphere_sqrt_double = on; }
static float sphere_intersects(const geom *g, const ray_t *ray)
{
   v3 eMinusC = vsub(ray->o, g->c);
   float dDotD = vdot(ray->d, ray->d);
   float discriminant = vdot(ray->d, eMinusC) * vdot(ray->d, eMinusC)
      - dDotD * (vdot(eMinusC, eMinusC) - g->r * g->r);
   if (discriminant < 0.0f)
      return -1.0f;
   if (g_sphere_sqrt_double) {
      float firstD = (float)(((double)vdot(vneg(ray->d), eMinusC) - sqrt((double)discriminant)) / (double)dDotD);
      if (isFloatLessThan(0.0f, firstD)) return firstD;
      return (float)(((double)vdot(vneg(ray->d), eMinusC) + sqrt((double)discriminant)) / (double)dDotD);
   }
   float firstIntersect = (vdot(vneg(ray->d), eMinusC) - sqrtf(discriminant)) / dDotD;
   if (isFloatLessThan(0.0f, firstIntersect)) {
      return firstIntersect;
   } else {
      return (vdot(vneg(ray->d), eMinusC) + sqrtf(discriminant)) / dDotD;
   }
}
/* Plane.h:39-50 */
static float plane_intersects(const geom *g, const ray_t *r)
{
   v3 c = vscale(g->pn, g->pd);
   float numer = vdot(vneg(g->pn), vsub(r->o, c));
   float denom = vdot(g->pn, r->d);
   float t;
   if (isFloatZero(numer) || isFloatZero(denom) || isFloatLessThan(t = numer / denom, 0.0f))
      return -1.0f;
   else
      return t;
}
/* Triangle.h:55-103 */
static float triangle_intersects(const geom *g, const ray_t *r)
{
   float numer = vdot(vneg(g->tn), vsub(r->o, g->tc));
   float denom = vdot(g->tn, r->d);
   float t;
   if (isFloatZero(numer) || isFloatZero(denom) || isFloatLessThan(t = numer / denom, 0.0f))
      return -1.0f;

   v3 P = ray_point(r, t);
   v3 A = g->p1, B = g->p2, C = g->p3;
   v3 AB = vsub(B, A);
   v3 AC = vsub(C, A);
   v3 N = vcross(AB, AC);

   v3 AP = vsub(P, A);
   v3 ABxAP = vcross(AB, AP);
   float v_num = vdot(N, ABxAP);
   if (v_num < 0.0f) return -1.0f;

   v3 BP = vsub(P, B);
   v3 BC = vsub(C, B);
   v3 BCxBP = vcross(BC, BP);
   if (vdot(N, BCxBP) < 0.0f)
      return -1.0f;

   v3 CP = vsub(P, C);
   v3 CAxCP = vcross(CP, AC);
   float u_num = vdot(N, CAxCP);
   if (u_num < 0.0f)
      return -1.0f;
   return t;
}
/* Box.h:37-39 */
static float box_intersects(const geom *g, const ray_t *r) { return boxIntersect(r, g->minP, g->maxP); }

/* Geometry.h:16-18 */
static float geom_getIntersection(const geom *g, const ray_t *r)
{
   ray_t tr = ray_transform(r, g->inv);
   switch (g->type) {
   case G_SPHERE: return sphere_intersects(g, &tr);
   case G_PLANE: return plane_intersects(g, &tr);
   case G_BOX: return box_intersects(g, &tr);
   default: return triangle_intersects(g, &tr);
   }
}

/* getNormalAt: Sphere.h:19-24, Plane.h:20-22, Triangle.h:24-26,
 * SmoothTriangle.h:31-38, Box.h:42-60 */
static v3 geom_getNormalAt(const geom *g, const ray_t *ray, float param)
{
   switch (g->type) {
   case G_SPHERE: {
      v3 pos = ray_point(ray, param);
      return vnormalize(vsub(pos, g->wsCenter));
   }
   case G_PLANE:
      return g->wsN;
   case G_TRIANGLE:
      return g->wsNormal;
   case G_SMOOTH: {
      v3 q = ray_point(ray, param);
      float area = vdot(vcross(vsub(g->p2, g->p1), vsub(g->p3, g->p1)), g->tn);
      float beta = vdot(vcross(vsub(g->p1, g->p3), vsub(q, g->p3)), g->tn) / area;
      float gamma = vdot(vcross(vsub(g->p2, g->p1), vsub(q, g->p1)), g->tn) / area;
      float alpha = (float)(1.0 - (double)beta - (double)gamma);   /* "1.0" is a double literal */
      v3 s = vadd(vadd(vscale(g->wsN1, alpha), vscale(g->wsN2, beta)), vscale(g->wsN3, gamma));
      return vnormalize(s);
   }
   default: { /* G_BOX */
      ray_t tr = ray_transform(ray, g->inv);
      v3 ip = ray_point(&tr, param);
      v3 norm;
      if (isFloatEqual(ip.x, g->minP.x)) norm = V(-1.0f, 0.0f, 0.0f);
      else if (isFloatEqual(ip.x, g->maxP.x)) norm = V(1.0f, 0.0f, 0.0f);
      else if (isFloatEqual(ip.y, g->minP.y)) norm = V(0.0f, -1.0f, 0.0f);
      else if (isFloatEqual(ip.y, g->maxP.y)) norm = V(0.0f, 1.0f, 0.0f);
      else if (isFloatEqual(ip.z, g->minP.z)) norm = V(0.0f, 0.0f, -1.0f);
      else norm = V(0.0f, 0.0f, 1.0f);
      v4 w = mat_mul_vec(g->trans, norm.x, norm.y, norm.z, 0.0f);
      return V(w.x, w.y, w.z);
   }
   }
}

/* getCenter: Sphere.h:26-28, Box.h:30-34, Triangle.h:28-31, Plane.h:24-26 */
static v3 geom_getCenter(const geom *g)
{
   switch (g->type) {
   case G_SPHERE: return g->wsCenter;
   case G_PLANE: return vscale(g->wsN, g->pd);
   case G_BOX: {
      v3 c = vdivs(vadd(g->minP, g->maxP), 2.0f);
      v4 w = mat_mul_vec(g->trans, c.x, c.y, c.z, 1.0f);
      return V(w.x, w.y, w.z);
   }
   default: {
      v3 c = vdivs(vadd(vadd(g->p1, g->p2), g->p3), 3.0f);
      v4 w = vec_mul_mat(c.x, c.y, c.z, 1.0f, g->trans);   /* row vector times matrix: translation is dropped */
      return V(w.x, w.y, w.z);
   }
   }
}

/* getBoundingBox: Sphere.h:30-32, Box.h:21-23, Triangle.h:33-42, Plane.h:28-30 */
static bbox_t geom_getBoundingBox(const geom *g)
{
   switch (g->type) {
   case G_SPHERE:
      return generateBoundingBox(V(g->c.x - g->r, g->c.y - g->r, g->c.z - g->r),
                                 V(g->c.x + g->r, g->c.y + g->r, g->c.z + g->r), g->trans);
   case G_BOX:
      return generateBoundingBox(g->minP, g->maxP, g->trans);
   case G_PLANE: {
      float f = FLT_MAX / 5;
      return bbox_make(V(-f, -f, -f), V(f, f, f));
   }
   default: {
      v4 a = mat_mul_vec(g->trans, g->p1.x, g->p1.y, g->p1.z, 1.0f);
      v4 b = mat_mul_vec(g->trans, g->p2.x, g->p2.y, g->p2.z, 1.0f);
      v4 c = mat_mul_vec(g->trans, g->p3.x, g->p3.y, g->p3.z, 1.0f);
      v3 t1 = V(a.x, a.y, a.z), t2 = V(b.x, b.y, b.z), t3 = V(c.x, c.y, c.z);
      v3 mn = smallestCorner(smallestCorner(t1, t2), t3);
      v3 mx = biggestCorner(biggestCorner(t1, t2), t3);
      return bbox_make(mn, mx);
   }
   }
}

/* ---- BVH (bvh.h:10-25, bvh.cpp:30-223) ---------------------------------- */
typedef struct bvh_node {
   int left, right;   /* node indices, -1 = NULL */
   int geom;          /* primitive id, -1 = NULL */
   bbox_t bb;
} bvh_node;

struct orc_scene {
   rt_camera cam;
   v3 lookAtDir;
   int n_geom, n_planes, n_lights;
   geom *geoms;        /* geomList in creation order: id = index */
   geom *planes;
   rt_light *lights;
   int n_textures;
   rt_texture *textures;   /* copies (pixels included) */
   int32_t *order;     /* geomList after the sort rounds */
   bvh_node *nodes;
   int n_nodes;
   int root;           /* -1 when the tree is empty */
   char *trace;
   size_t trace_len, trace_cap;
};

static void trace_add(orc_scene *s, const char *line)
{
   size_t n = strlen(line);
   if (s->trace_len + n + 1 > s->trace_cap) {
      s->trace_cap = (s->trace_cap + n + 1) * 2;
      s->trace = (char *)realloc(s->trace, s->trace_cap);
   }
   memcpy(s->trace + s->trace_len, line, n + 1);
   s->trace_len += n;
}

/* merge kernel, bvh.cpp:153-171: one "thread" idx merges two runs of `width` */
static void merge_pass(const orc_scene *s, const int32_t *oldBuf, int32_t *newBuf, int width, int size, int axis)
{
   int divs = (size - 1) / (width * 2) + 1;
   for (int idx = 0; idx < divs; idx++) {
      int i1 = idx * width * 2;
      int i2 = idx * width * 2 + width;
      int div1End = i2 < size ? i2 : size;
      int div2End = (i2 + width) < size ? (i2 + width) : size;
      for (int i = i1; i < div2End; i++) {
         if (i1 < div1End && (i2 >= div2End ||
             vget(geom_getCenter(&s->geoms[oldBuf[i1]]), axis) < vget(geom_getCenter(&s->geoms[oldBuf[i2]]), axis))) {
            newBuf[i] = oldBuf[i1++];
         } else {
            newBuf[i] = oldBuf[i2++];
         }
      }
   }
}

typedef struct { int first, second; } sort_range;

/* multiKernelSort, bvh.cpp:106-146 (ping-pong buffers, copyOver when the last
 * merge of a range landed in the scratch buffer) */
static void multiKernelSort(const orc_scene *s, int32_t *buffer1, int32_t *buffer2,
                            const sort_range *idxs, int n_idxs, int axis)
{
   int size = 0;
   int swaps = 1;
   for (int i = 0; i < n_idxs; i++) {
      int sortSize = idxs[i].second - idxs[i].first;
      if (sortSize > size) size = sortSize;
   }
   for (int width = 1; width < size + 1; width = 2 * width) {
      for (int i = 0; i < n_idxs; i++) {
         int start = idxs[i].first;
         int end = idxs[i].second;
         int sortSize = end - start;
         if (width >= sortSize) continue;
         int divs = (sortSize - 1) / (width * 2) + 1;
         merge_pass(s, buffer1 + start, buffer2 + start, width, sortSize, axis);
         if (divs == 1 && swaps % 2 == 1) {
            for (int k = 0; k < sortSize; k++) buffer1[start + k] = buffer2[start + k];
         }
      }
      swaps++;
      { int32_t *tmp = buffer1; buffer1 = buffer2; buffer2 = tmp; }
   }
}

static int new_node(orc_scene *s)
{
   bvh_node *n = &s->nodes[s->n_nodes];
   n->left = n->right = n->geom = -1;
   n->bb = bbox_make(V(0, 0, 0), V(0, 0, 0));   /* BoundingBox default arguments, BoundingBox.h:11 */
   return s->n_nodes++;
}
static int new_leaf(orc_scene *s, int prim)   /* BVHNode(Geometry*), bvh.h:16-18 */
{
   int i = new_node(s);
   s->nodes[i].geom = prim;
   s->nodes[i].bb = geom_getBoundingBox(&s->geoms[prim]);
   return i;
}

static void formBVH(orc_scene *s)
{
   int geomCount = s->n_geom;
   char line[96];
   s->order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(geomCount > 0 ? geomCount : 1));
   for (int i = 0; i < geomCount; i++) s->order[i] = i;
   s->root = -1;
   s->n_nodes = 0;
   s->nodes = NULL;
   if (geomCount <= 0) return;   /* the reference leaves tree->root uninitialised here */

   int32_t *scratch = (int32_t *)malloc(sizeof(int32_t) * (size_t)geomCount);
   sort_range *oldQ = (sort_range *)malloc(sizeof(sort_range) * (size_t)(2 * geomCount + 4));
   sort_range *newQ = (sort_range *)malloc(sizeof(sort_range) * (size_t)(2 * geomCount + 4));
   int nOld = 0, nNew = 0;
   int axis = 0;
   oldQ[nOld].first = 0; oldQ[nOld].second = geomCount; nOld++;
   while (nOld > 0) {
      multiKernelSort(s, s->order, scratch, oldQ, nOld, axis);
      while (nOld > 0) {
         int start = oldQ[nOld - 1].first;
         int end = oldQ[nOld - 1].second;
         nOld--;
         if (end - start > 2) {
            snprintf(line, sizeof line, "sorting from %d to %d\n", start, end);
            trace_add(s, line);
            int closestPow2 = 2;
            while (closestPow2 * 2 < end - start) closestPow2 *= 2;
            /* bvh.cpp:65: the left range is pushed as (start, closestPow2), not
             * (start, start + closestPow2) -- empty whenever start > 0 */
            newQ[nNew].first = start; newQ[nNew].second = closestPow2; nNew++;
            newQ[nNew].first = start + closestPow2; newQ[nNew].second = end; nNew++;
         }
      }
      { sort_range *t = oldQ; oldQ = newQ; newQ = t; nOld = nNew; nNew = 0; }
      axis = (axis + 1) % 3;
      snprintf(line, sizeof line, "using axis %d\n", axis);
      trace_add(s, line);
   }
   free(scratch); free(oldQ); free(newQ);

   /* setupBVHNodes (bvh.cpp:179-192) then convergeBVHNodes per level (194-214) */
   int bufferSize = (geomCount - 1) / 2 + 1;
   s->nodes = (bvh_node *)malloc(sizeof(bvh_node) * (size_t)(2 * geomCount + 2));
   int *buf1 = (int *)malloc(sizeof(int) * (size_t)bufferSize);
   int *buf2 = (int *)malloc(sizeof(int) * (size_t)bufferSize);
   for (int idx = 0; idx < bufferSize; idx++) {
      if (idx * 2 >= geomCount) continue;
      if (idx * 2 + 1 < geomCount) {
         int n = new_node(s);
         int l = new_leaf(s, s->order[idx * 2]);
         int r = new_leaf(s, s->order[idx * 2 + 1]);
         s->nodes[n].left = l;
         s->nodes[n].right = r;
         s->nodes[n].bb = combineBoundingBox(&s->nodes[l].bb, &s->nodes[r].bb);
         buf1[idx] = n;
      } else {
         buf1[idx] = new_leaf(s, s->order[idx * 2]);
      }
   }
   for (int nodesLeft = bufferSize; nodesLeft > 1; nodesLeft = (nodesLeft - 1) / 2 + 1) {
      for (int idx = 0; idx < bufferSize; idx++) {
         buf2[idx] = -1;
         if (idx * 2 >= bufferSize) continue;
         if (idx * 2 + 1 < nodesLeft) {
            int n = new_node(s);
            s->nodes[n].left = buf1[idx * 2];
            s->nodes[n].right = buf1[idx * 2 + 1];
            s->nodes[n].bb = combineBoundingBox(&s->nodes[s->nodes[n].left].bb, &s->nodes[s->nodes[n].right].bb);
            buf2[idx] = n;
         } else if (idx * 2 < nodesLeft) {
            buf2[idx] = buf1[idx * 2];
         }
      }
      { int *t = buf1; buf1 = buf2; buf2 = t; }
   }
   s->root = buf1[0];
   free(buf1); free(buf2);
}

orc_scene *orc_scene_create(const rt_scene_desc *desc)
{
   orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
   s->cam = desc->camera;
   s->lookAtDir = vnormalize(vsub(vload(desc->camera.look_at), vload(desc->camera.pos)));   /* kernel.cu:717-718 */
   s->n_geom = desc->n_spheres + desc->n_triangles + desc->n_boxes + desc->n_smooth_triangles;
   s->n_planes = desc->n_planes;
   s->n_lights = desc->n_lights;
   s->geoms = (geom *)malloc(sizeof(geom) * (size_t)(s->n_geom > 0 ? s->n_geom : 1));
   s->planes = (geom *)malloc(sizeof(geom) * (size_t)(s->n_planes > 0 ? s->n_planes : 1));
   s->lights = (rt_light *)malloc(sizeof(rt_light) * (size_t)(s->n_lights > 0 ? s->n_lights : 1));
   /* creation order of initScene, kernel.cu:411-456 */
   for (int i = 0; i < desc->n_planes; i++) plane_init(&s->planes[i], &desc->planes[i]);
   int k = 0;
   for (int i = 0; i < desc->n_spheres; i++) sphere_init(&s->geoms[k++], &desc->spheres[i]);
   for (int i = 0; i < desc->n_triangles; i++) {
      const rt_triangle *t = &desc->triangles[i];
      triangle_init(&s->geoms[k], G_TRIANGLE, t->p1, t->p2, t->p3, &t->mod);
      memcpy(s->geoms[k].vt1, t->vt1, 8); memcpy(s->geoms[k].vt2, t->vt2, 8); memcpy(s->geoms[k].vt3, t->vt3, 8);
      k++;
   }
   for (int i = 0; i < desc->n_boxes; i++) box_init(&s->geoms[k++], &desc->boxes[i]);
   for (int i = 0; i < desc->n_smooth_triangles; i++) {
      const rt_smooth_triangle *t = &desc->smooth_triangles[i];
      smooth_init(&s->geoms[k], t);
      memcpy(s->geoms[k].vt1, t->vt1, 8); memcpy(s->geoms[k].vt2, t->vt2, 8); memcpy(s->geoms[k].vt3, t->vt3, 8);
      k++;
   }
   s->n_textures = desc->n_textures;
   s->textures = (rt_texture *)calloc((size_t)(desc->n_textures > 0 ? desc->n_textures : 1), sizeof(rt_texture));
   for (int i = 0; i < desc->n_textures; i++) {
      s->textures[i] = desc->textures[i];
      s->textures[i].rgba = NULL;
      if (desc->textures[i].rgba) {
         size_t bytes = (size_t)desc->textures[i].width * (size_t)desc->textures[i].height * 4;
         s->textures[i].rgba = (uint8_t *)malloc(bytes);
         memcpy(s->textures[i].rgba, desc->textures[i].rgba, bytes);
      }
   }
   for (int i = 0; i < desc->n_lights; i++) s->lights[i] = desc->lights[i];
   formBVH(s);
   return s;
}

void orc_scene_destroy(orc_scene *s)
{
   if (!s) return;
   free(s->geoms); free(s->planes); free(s->lights); free(s->order); free(s->nodes); free(s->trace);
   if (s->textures) { for (int i = 0; i < s->n_textures; i++) free(s->textures[i].rgba); free(s->textures); }
   free(s);
}

int orc_n_bvh_prims(const orc_scene *s) { return s->n_geom; }
int orc_n_planes(const orc_scene *s) { return s->n_planes; }
const int32_t *orc_leaf_order(const orc_scene *s) { return s->order; }

size_t orc_bvh_trace(const orc_scene *s, char *buf, size_t cap)
{
   size_t n = s->trace_len;
   if (buf && cap > 0) {
      size_t m = n < cap - 1 ? n : cap - 1;
      if (s->trace) memcpy(buf, s->trace, m);
      buf[m] = 0;
   }
   return n;
}

static void store3(float *dst, v3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

static int flatten_rec(const orc_scene *s, int node, rt_bvh_node *out, int *count, int cap, int depth, int *maxdepth)
{
   /* node is internal; returns its flat index */
   int me = (*count)++;
   if (depth + 1 > *maxdepth) *maxdepth = depth + 1;
   const bvh_node *n = &s->nodes[node];
   const bvh_node *l = &s->nodes[n->left];
   const bvh_node *r = &s->nodes[n->right];
   int li, ri;
   if (l->geom >= 0) li = ~l->geom; else li = flatten_rec(s, n->left, out, count, cap, depth + 1, maxdepth);
   if (r->geom >= 0) ri = ~r->geom; else ri = flatten_rec(s, n->right, out, count, cap, depth + 1, maxdepth);
   if (me < cap) {
      rt_bvh_node *o = &out[me];
      store3(o->lmin, l->bb.min); store3(o->lmax, l->bb.max);
      store3(o->rmin, r->bb.min); store3(o->rmax, r->bb.max);
      o->left = li; o->right = ri; o->pad[0] = o->pad[1] = 0;
   }
   return me;
}

int orc_flatten_bvh(const orc_scene *s, rt_bvh_node *nodes, int cap, int32_t *root, int32_t *depth)
{
   int count = 0, maxdepth = 0;
   if (s->root < 0) { *root = INT32_MIN; if (depth) *depth = 0; return 0; }
   if (s->nodes[s->root].geom >= 0) { *root = ~s->nodes[s->root].geom; if (depth) *depth = 0; return 0; }
   *root = flatten_rec(s, s->root, nodes, &count, cap, 0, &maxdepth);
   if (depth) *depth = maxdepth;
   return count;
}

/* ---- getClosestIntersection, kernel.cu:70-200 --------------------------- */
typedef struct { int nextDir; int node; } stack_entry;
enum { kLeft = 0, kRight = 1 };

static void getClosestIntersection(const orc_scene *s, const ray_t *ray, int *retObj, float *retParam,
                                   float toBeat, orc_counters *cn)
{
   float t = FLT_MAX;
   int closestGeom = -1;
   stack_entry stack[K_MAX_STACK];
   int stackSize = 0;
   int justPoppedStack = 0;
   int cursor = s->root;
   int nextDir = 0;

   if (cursor >= 0) {
      do {
         if (stackSize >= K_MAX_STACK) {
            printf("Stack full, aborting!\n");
            return;
         }
         const bvh_node *cur = &s->nodes[cursor];
         if (cur->geom >= 0) {
            if (cn) { cn->prim_tests++; if (stackSize > cn->max_stack) cn->max_stack = stackSize; }
            float dist = geom_getIntersection(&s->geoms[cur->geom], ray);
            if (isFloatEqual(t, dist)) {
               if (cn) cn->tie_events++;
               if (closestGeom >= 0) {   /* the reference dereferences closestGeom unchecked; it is never NULL here for finite dist */
                  v3 oldNorm = geom_getNormalAt(&s->geoms[closestGeom], ray, t);
                  v3 newNorm = geom_getNormalAt(&s->geoms[cur->geom], ray, dist);
                  v3 eye = vnormalize(vneg(ray->d));
                  float newDot = vdot(eye, newNorm);
                  float oldDot = vdot(eye, oldNorm);
                  if (newDot > oldDot) {
                     closestGeom = cur->geom;
                     t = dist;
                     if (t < toBeat) { *retObj = closestGeom; *retParam = t; return; }
                  }
               }
            } else {
               if (dist < t && isFloatAboveZero(dist)) {
                  t = dist;
                  closestGeom = cur->geom;
                  if (t < toBeat) { *retObj = closestGeom; *retParam = t; return; }
               }
            }
         } else if (!justPoppedStack) {
            if (cn) { cn->node_first_visits++; cn->box_tests += 2; }
            float left = boxIntersect(ray, s->nodes[cur->left].bb.min, s->nodes[cur->left].bb.max);
            /* the "!cursor->right" branch (kernel.cu:139-143) is unreachable: internal nodes have two children */
            float right = boxIntersect(ray, s->nodes[cur->right].bb.min, s->nodes[cur->right].bb.max);
            if (isFloatAboveZero(right) && (right <= left || !isFloatAboveZero(left)) && right < t) {
               if (isFloatAboveZero(left)) { stack[stackSize].node = cursor; stack[stackSize].nextDir = kLeft; stackSize++; }
               cursor = cur->right;
               justPoppedStack = 0;
               continue;
            } else if (isFloatAboveZero(left) && (left <= right || !isFloatAboveZero(right)) && left < t) {
               if (isFloatAboveZero(right)) { stack[stackSize].node = cursor; stack[stackSize].nextDir = kRight; stackSize++; }
               cursor = cur->left;
               justPoppedStack = 0;
               continue;
            }
         } else {
            if (cn) { cn->node_revisits++; cn->box_tests += 1; }
            if (nextDir == kRight) {
               float right = boxIntersect(ray, s->nodes[cur->right].bb.min, s->nodes[cur->right].bb.max);
               if (right < t) { cursor = cur->right; justPoppedStack = 0; continue; }
            } else {
               float left = boxIntersect(ray, s->nodes[cur->left].bb.min, s->nodes[cur->left].bb.max);
               if (left < t) { cursor = cur->left; justPoppedStack = 0; continue; }
            }
         }
         if (stackSize == 0) break;
         cursor = stack[stackSize - 1].node;
         nextDir = stack[stackSize - 1].nextDir;
         justPoppedStack = 1;
         stackSize--;
      } while (1);
   }

   for (int planeIdx = 0; planeIdx < s->n_planes; planeIdx++) {
      if (cn) cn->plane_tests++;
      float dist = geom_getIntersection(&s->planes[planeIdx], ray);
      if (isFloatLessThan(dist, t) && isFloatAboveZero(dist)) {
         closestGeom = s->n_geom + planeIdx;
         t = dist;
      }
   }
   *retObj = closestGeom;
   *retParam = t;
}

static const geom *geom_by_id(const orc_scene *s, int id)
{
   return id < s->n_geom ? &s->geoms[id] : &s->planes[id - s->n_geom];
}

void orc_closest(const orc_scene *s, const float o[3], const float d[3], float to_beat,
                 int32_t *prim, float *t, orc_counters *c)
{
   ray_t r; r.o = vload(o); r.d = vload(d);
   int obj = -1; float tt = FLT_MAX;   /* on stack overflow the reference leaves its outputs unwritten */
   getClosestIntersection(s, &r, &obj, &tt, to_beat, c);
   *prim = obj; *t = tt;
}

/* ---- shaders ------------------------------------------------------------ */
/* PhongShader.h:7-25 */
static v3 phong_shade(v3 matColor, float amb, float dif, float spec, float roughness,
                      v3 eyeVec, v3 lightDir, v3 lightColor, v3 normal, int inShadow)
{
   v3 light = V(0.0f, 0.0f, 0.0f);
   light = vadd(light, vscale(lightColor, amb));
   if (inShadow) return vmul(light, matColor);
   light = vadd(light, vscale(lightColor, dif * clampf(vdot(normal, lightDir), 0.0f, 1.0f)));
   v3 reflect = vsub(vscale(normal, 2.0f * vdot(lightDir, normal)), lightDir);
   light = vadd(light, vscale(lightColor, spec * powf(clampf(vdot(reflect, eyeVec), 0.0f, 1.0f), 1.0f / roughness)));
   return vmul(light, matColor);
}
/* CookTorranceShader.h:7-46 */
static v3 cook_torrance_shade(v3 matColor, float amb, float dif, float spec, float roughness,
                              v3 eyeVec, v3 lightDir, v3 lightColor, v3 normal, int inShadow)
{
   v3 light = V(0.0f, 0.0f, 0.0f);
   light = vadd(light, vscale(lightColor, amb));
   if (inShadow) return vmul(light, matColor);
   light = vadd(light, vscale(lightColor, dif * clampf(vdot(normal, lightDir), 0.0f, 1.0f)));

   v3 h = vnormalize(vadd(lightDir, eyeVec));
   float ior = 2.0f;
   float r0 = (1.0f - ior) * (1.0f - ior) / ((1.0f + ior) * (1.0f + ior));
   float f = r0 + (1.0f - r0) * powf(1.0f - vdot(lightDir, normal), 5.0f);

   float nDotH = vdot(normal, h);
   float mSqr = roughness * roughness;
   float d = expf((nDotH * nDotH - 1.0f) / (mSqr * nDotH * nDotH)) /
             (mSqr * nDotH * nDotH * nDotH * nDotH);

   float gStart = 2.0f * nDotH / (vdot(eyeVec, h));
   float g1 = gStart * vdot(eyeVec, normal);
   float g2 = gStart * vdot(lightDir, normal);
   float g = g1 < g2 ? g1 : g2;
   g = g < 1.0f ? g : 1.0f;

   float kSpec = d * f * g / (4.0f * vdot(eyeVec, normal) * vdot(normal, lightDir));
   light = vadd(light, vscale(lightColor, spec * kSpec));
   return vmul(light, matColor);
}
static v3 shader_shade(int shading, v3 matColor, float amb, float dif, float spec, float roughness,
                       v3 eyeVec, v3 lightDir, v3 lightColor, v3 normal, int inShadow)
{
   if (shading == RT_COOK_TORRANCE)
      return cook_torrance_shade(matColor, amb, dif, spec, roughness, eyeVec, lightDir, lightColor, normal, inShadow);
   return phong_shade(matColor, amb, dif, spec, roughness, eyeVec, lightDir, lightColor, normal, inShadow);
}

/* ---- shadeObject / getReflection / getRefraction, kernel.cu:202-276, 335-382 */
typedef struct {
   const orc_scene *s; int shading; orc_counters *cn;
   uint64_t *sig;     /* when set: a running hash of every decision of the shade tree (which primitive each ray hit or that
                         it missed, which feelers were shadowed), for orc_interior_mask */
} shade_ctx;
static inline void sig_mix(const shade_ctx *cx, uint64_t v)
{
   if (cx->sig) { uint64_t h = *cx->sig ^ (v + 0x9e3779b97f4a7c15ULL); h *= 0xbf58476d1ce4e5b9ULL; h ^= h >> 29; *cx->sig = h; }
}

static v3 shadeObject(const shade_ctx *cx, int invRecLevel, int geomId, float intParam, ray_t ray);

static int isInShadow(const shade_ctx *cx, const ray_t *shadow, float intersectParam)
{
   float closestIntersect = FLT_MAX;
   int closestObj = -1;
   if (cx->cn) { cx->cn->rays_shadow++; if (intersectParam == -1.0f) cx->cn->shadow_selfmiss++; }
   getClosestIntersection(cx->s, shadow, &closestObj, &closestIntersect, intersectParam, cx->cn);
   sig_mix(cx, isFloatLessThan(closestIntersect, intersectParam) ? 0x51ULL : 0x52ULL);
   return isFloatLessThan(closestIntersect, intersectParam);
}

static v3 getReflection(const shade_ctx *cx, int invRecLevel, v3 point, v3 normal, v3 eyeVec)
{
   if (invRecLevel == 0) return V(0.0f, 0.0f, 0.0f);
   ray_t reflectRay;
   reflectRay.o = point;
   reflectRay.d = vsub(vscale(normal, 2.0f * vdot(normal, eyeVec)), eyeVec);
   reflectRay.o = vadd(reflectRay.o, vscale(reflectRay.d, BIG_EPSILON));
   int closestGeom = -1;
   float t = FLT_MAX;
   if (cx->cn) cx->cn->rays_reflect++;
   getClosestIntersection(cx->s, &reflectRay, &closestGeom, &t, -FLT_MAX, cx->cn);
   sig_mix(cx, 0x1000000ULL + (uint64_t)(int64_t)closestGeom);
   if (closestGeom != -1)
      return shadeObject(cx, invRecLevel, closestGeom, t, reflectRay);
   return V(0.0f, 0.0f, 0.0f);
}

static v3 getRefraction(const shade_ctx *cx, int invRecLevel, v3 point, v3 normal, float ior, v3 eyeVec)
{
   if (invRecLevel == 0) return V(0.0f, 0.0f, 0.0f);
   float n1, n2;
   v3 refrNorm;
   v3 d = vneg(eyeVec);
   if (isFloatLessThan(vdot(eyeVec, normal), 0.0f)) {
      n1 = ior; n2 = K_AIR_IOR;
      refrNorm = vneg(normal);
   } else {
      n1 = K_AIR_IOR; n2 = ior;
      refrNorm = normal;
   }
   float dDotN = vdot(d, refrNorm);
   float nr = n1 / n2;
   float discriminant = 1.0f - nr * nr * (1.0f - dDotN * dDotN);
   if (discriminant > 0.0f) {
      v3 refracDir = vsub(vscale(vsub(d, vscale(refrNorm, dDotN)), nr), vscale(refrNorm, sqrtf(discriminant)));
      ray_t refracRay;
      refracRay.o = point;
      refracRay.d = refracDir;
      refracRay.o = vadd(refracRay.o, vscale(refracRay.d, BIG_EPSILON));
      int closestGeom = -1;
      float t = FLT_MAX;
      if (cx->cn) cx->cn->rays_refract++;
      getClosestIntersection(cx->s, &refracRay, &closestGeom, &t, -FLT_MAX, cx->cn);
      sig_mix(cx, 0x2000000ULL + (uint64_t)(int64_t)closestGeom);
      if (closestGeom != -1)
         return shadeObject(cx, invRecLevel, closestGeom, t, refracRay);
   }
   return V(0.0f, 0.0f, 0.0f);
}

/* Triangle::UVAt, Triangle.h:44-52 (SmoothTriangle inherits it); the other classes return vec2(0)
 * (Sphere.h:34-37, Plane.h:32-35, Box.h:25-28). */
static void orc_uv_at(const geom *g, const ray_t *r, float param, float *uv)
{
   uv[0] = uv[1] = 0.0f;
   if (g->type != G_TRIANGLE && g->type != G_SMOOTH) return;
   v3 q = ray_point(r, param);
   float area = vdot(vcross(vsub(g->p2, g->p1), vsub(g->p3, g->p1)), g->tn);
   float beta = vdot(vcross(vsub(g->p1, g->p3), vsub(q, g->p3)), g->tn) / area;
   float gamma = vdot(vcross(vsub(g->p2, g->p1), vsub(q, g->p1)), g->tn) / area;
   float alpha = (float)(1.0 - (double)beta - (double)gamma);   /* "1.0" is a double literal */
   uv[0] = alpha * g->vt1[0] + beta * g->vt2[0] + gamma * g->vt3[0];
   uv[1] = alpha * g->vt1[1] + beta * g->vt2[1] + gamma * g->vt3[1];
}

/* tex2D(mytex, u, v) for the texture the reference binds (kernel.cu:32, 590-593): uchar4 texels read as
 * normalised floats, wrap addressing, linear filtering, normalised coordinates.  Restated from the
 * published description of CUDA's linear filter: x = frac(u)*W - 0.5, the four texels around it, weights
 * in 9-bit fixed point with 8 fractional bits.  There is no CUDA texture unit here to check the last
 * bit against: PARITY UNPINNED for this function (the reference's blitzcrank.tga is the only witness). */
static void orc_tex2d(const rt_texture *tex, float u, float v, float *rgb)
{
   int W = tex->width, H = tex->height;
   float x = (u - floorf(u)) * (float)W - 0.5f;
   float y = (v - floorf(v)) * (float)H - 0.5f;
   float fx = floorf(x), fy = floorf(y);
   float a = floorf((x - fx) * 256.0f + 0.5f) * 0.00390625f;
   float b = floorf((y - fy) * 256.0f + 0.5f) * 0.00390625f;
   int i0 = (int)fx, j0 = (int)fy;
   i0 = ((i0 % W) + W) % W; j0 = ((j0 % H) + H) % H;
   int i1 = (i0 + 1) % W, j1 = (j0 + 1) % H;
   const uint8_t *t00 = tex->rgba + 4 * ((size_t)j0 * W + i0), *t10 = tex->rgba + 4 * ((size_t)j0 * W + i1);
   const uint8_t *t01 = tex->rgba + 4 * ((size_t)j1 * W + i0), *t11 = tex->rgba + 4 * ((size_t)j1 * W + i1);
   float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
   for (int c = 0; c < 3; c++)
      rgb[c] = w00 * ((float)t00[c] / 255.0f) + w10 * ((float)t10[c] / 255.0f) + w01 * ((float)t01[c] / 255.0f) +
               w11 * ((float)t11[c] / 255.0f);
}

/* getColor, kernel.cu:323-332 */
static v3 getColor(const orc_scene *s, const geom *g, const ray_t *ray, float param)
{
   if (g->mat.tex_id == -1 || g->mat.tex_id >= s->n_textures || !s->textures[g->mat.tex_id].rgba) return vload(g->mat.clr);
   float uv[2], rgb[3];
   orc_uv_at(g, ray, param, uv);
   orc_tex2d(&s->textures[g->mat.tex_id], uv[0], uv[1], rgb);
   return V(rgb[0], rgb[1], rgb[2]);
}

static v3 shadeObject(const shade_ctx *cx, int invRecLevel, int geomId, float intParam, ray_t ray)
{
   if (invRecLevel == 0) return V(0.0f, 0.0f, 0.0f);
   const orc_scene *s = cx->s;
   const geom *g = geom_by_id(s, geomId);
   if (cx->cn) cx->cn->shade_nodes++;

   v3 intersectPoint = ray_point(&ray, intParam);
   rt_material m = g->mat;
   v3 normal = geom_getNormalAt(g, &ray, intParam);
   v3 matClr = getColor(s, g, &ray, intParam);
   v3 eyeVec = vnormalize(vneg(ray.d));
   v3 totalLight = V(0.0f, 0.0f, 0.0f);

   for (int lightIdx = 0; lightIdx < s->n_lights; lightIdx++) {
      v3 lp = vload(s->lights[lightIdx].pos);
      v3 light = vload(s->lights[lightIdx].clr);            /* PointLight.h:10-12 */
      v3 lightDir = vnormalize(vsub(lp, intersectPoint));   /* PointLight.h:14-16 */
      ray_t shadow; shadow.o = lp; shadow.d = vsub(intersectPoint, lp);   /* PointLight.h:18-20 */
      float intersectParam = geom_getIntersection(g, &shadow);
      int inShadow = isInShadow(cx, &shadow, intersectParam);
      totalLight = vadd(totalLight, shader_shade(cx->shading, matClr, m.amb, m.dif, m.spec, m.rough,
                                                 eyeVec, lightDir, light, normal, inShadow));
   }

   v3 reflectedLight = V(0.0f, 0.0f, 0.0f);
   if (m.refl > 0.0f && invRecLevel - 1 > 0)
      reflectedLight = getReflection(cx, invRecLevel - 1, intersectPoint, normal, eyeVec);

   v3 refractedLight = V(0.0f, 0.0f, 0.0f);
   if (m.refr > 0.0f && invRecLevel - 1 > 0)
      refractedLight = getRefraction(cx, invRecLevel - 1, intersectPoint, normal, m.ior, eyeVec);

   return vadd(vadd(vscale(totalLight, 1.0f - m.refl - m.alpha), vscale(reflectedLight, m.refl)),
               vscale(refractedLight, m.alpha));
}

/* ---- ray generation, kernel.cu:477-498 ---------------------------------- */
static inline uint32_t orc_hash32(uint32_t x)
{
   x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
   return x;
}
void orc_jitter_uniforms(uint32_t seed, uint32_t index, float *r1, float *r2)
{
   /* counter-based stand-in for the two curand_uniform draws: range (0, 1] like cuRAND's */
   uint32_t base = orc_hash32(index ^ (seed * 0x9e3779b9U));
   uint32_t h1 = orc_hash32(base ^ 0x68bc21ebU);
   uint32_t h2 = orc_hash32(base ^ 0x02e5be93U);
   *r1 = (float)((h1 >> 8) + 1U) * 5.9604644775390625e-8f;
   *r2 = (float)((h2 >> 8) + 1U) * 5.9604644775390625e-8f;
}

static ray_t camera_ray_at(const orc_scene *s, const orc_params *p, int sroot, int x, int y, float r1, float r2);
static ray_t camera_ray(const orc_scene *s, const orc_params *p, int sroot, int x, int y)
{
   int resWidth = p->width * sroot;
   int index = y * resWidth + x;
   float r1 = 0.5f, r2 = 0.5f;
   if (p->jitter == RT_JITTER_HASH) orc_jitter_uniforms(p->seed, (uint32_t)index, &r1, &r2);
   return camera_ray_at(s, p, sroot, x, y, r1, r2);
}
/* the ray of sample (x, y) for the two uniform draws r1, r2 of kernel.cu:488-489 */
static ray_t camera_ray_at(const orc_scene *s, const orc_params *p, int sroot, int x, int y, float r1, float r2)
{
   int resWidth = p->width * sroot, resHeight = p->height * sroot;
   float uJitter = (r1 - .5f) / (float)resWidth;
   float vJitter = (r2 - .5f) / (float)resHeight;
   float u = 2.0f * (x / (float)resWidth) - 1.0f + uJitter;
   float v = 2.0f * (y / (float)resHeight) - 1.0f + vJitter;
   v3 right = vload(s->cam.right), up = vload(s->cam.up), pos = vload(s->cam.pos);
   v3 rPos = vadd(vadd(vscale(right, u * .5f), vscale(up, v * .5f)), pos);
   v3 rDir = vadd(vsub(rPos, pos), s->lookAtDir);
   ray_t r; r.o = rPos; r.d = rDir;
   return r;
}

static int isqrt_exact(int n)
{
   int r = (int)sqrt((double)n);
   return r;
}

void orc_camera_ray(const orc_scene *s, const orc_params *p, int x, int y, float o[3], float d[3])
{
   ray_t r = camera_ray(s, p, isqrt_exact(p->sample_count), x, y);
   store3(o, r.o); store3(d, r.d);
}

static void counters_add(orc_counters *a, const orc_counters *b)
{
   a->rays_primary += b->rays_primary; a->rays_shadow += b->rays_shadow;
   a->rays_reflect += b->rays_reflect; a->rays_refract += b->rays_refract;
   a->shadow_selfmiss += b->shadow_selfmiss; a->shade_nodes += b->shade_nodes;
   a->box_tests += b->box_tests; a->node_first_visits += b->node_first_visits;
   a->node_revisits += b->node_revisits; a->prim_tests += b->prim_tests;
   a->plane_tests += b->plane_tests; a->tie_events += b->tie_events;
   if (b->max_stack > a->max_stack) a->max_stack = b->max_stack;
}

/* rayTrace (kernel.cu:515-543) + averageBufferColors (kernel.cu:545-570) */
void orc_render(const orc_scene *s, const orc_params *p, float *rgb_samples, uint8_t *rgba,
                int32_t *prim, float *tout, orc_counters *c)
{
   int sroot = isqrt_exact(p->sample_count);
   if (sroot * sroot != p->sample_count) return;   /* kernel.cu:707-712 */
   int resWidth = p->width * sroot, resHeight = p->height * sroot;
   int maxRec = p->max_recurse > 0 ? p->max_recurse : K_MAX_RECURSE;
   size_t nsamp = (size_t)resWidth * (size_t)resHeight;
   float *buf = rgb_samples ? rgb_samples : (float *)malloc(nsamp * 3 * sizeof(float));
   if (c) memset(c, 0, sizeof *c);
#ifdef _OPENMP
   if (p->threads > 0) omp_set_num_threads(p->threads);
#endif
#pragma omp parallel
   {
      orc_counters local; memset(&local, 0, sizeof local);
      shade_ctx cx; cx.s = s; cx.shading = p->shading; cx.cn = c ? &local : NULL; cx.sig = NULL;
#pragma omp for schedule(dynamic, 4)
      for (int y = 0; y < resHeight; y++) {
         for (int x = 0; x < resWidth; x++) {
            size_t index = (size_t)y * resWidth + x;
            ray_t ray = camera_ray(s, p, sroot, x, y);
            float t = FLT_MAX; int closestGeom = -1;
            if (cx.cn) cx.cn->rays_primary++;
            getClosestIntersection(s, &ray, &closestGeom, &t, -FLT_MAX, cx.cn);
            if (prim) prim[index] = closestGeom;
            if (tout) tout[index] = t;
            v3 out = V(0.0f, 0.0f, 0.0f);
            if (closestGeom != -1) {
               v3 totalColor = shadeObject(&cx, maxRec, closestGeom, t, ray);
               out = V(clampf(totalColor.x, 0, 1), clampf(totalColor.y, 0, 1), clampf(totalColor.z, 0, 1));
            }
            buf[index * 3 + 0] = out.x; buf[index * 3 + 1] = out.y; buf[index * 3 + 2] = out.z;
         }
      }
      if (c) {
#pragma omp critical
         counters_add(c, &local);
      }
   }
   if (rgba) {
      for (int y = 0; y < p->height; y++) {
         for (int x = 0; x < p->width; x++) {
            v3 endColor = V(0.0f, 0.0f, 0.0f);
            for (int xOffset = 0; xOffset < sroot; xOffset++) {
               for (int yOffset = 0; yOffset < sroot; yOffset++) {
                  size_t bufferIndex = (size_t)(x * sroot + xOffset) + (size_t)(y * sroot + yOffset) * resWidth;
                  endColor = vadd(endColor, vload(&buf[bufferIndex * 3]));
               }
            }
            float div = (float)(sroot * sroot);
            endColor = V(endColor.x / div, endColor.y / div, endColor.z / div);
            endColor = vscale(endColor, 255.0f);
            uint8_t *px = &rgba[((size_t)y * p->width + x) * 4];
            /* float -> unsigned char truncates; NaN is mapped to 0 (what cvttss2si & 0xff gives) */
            px[0] = endColor.x == endColor.x ? (uint8_t)endColor.x : 0;
            px[1] = endColor.y == endColor.y ? (uint8_t)endColor.y : 0;
            px[2] = endColor.z == endColor.z ? (uint8_t)endColor.z : 0;
            px[3] = 255;
         }
      }
   }
   if (!rgb_samples) free(buf);
}

/* Which pixels of a frame are "interior": in every stratum of the pixel the whole shade tree makes the same decisions
 * (same primitive or miss for the primary and every bounce ray, same shadowed / lit answer of every feeler) for the
 * unjittered sample and for eight draws on the boundary of the jitter square (r1, r2 in {0, 0.5, 1}), i.e. (up to features
 * thinner than half that square) anywhere the reference's random sub-pixel offset can put that sample.  On such pixels a render with another jitter sequence differs only through
 * smooth shading, so the reference's own images can be compared there with a tight tolerance.
 * mask: width*height bytes, 1 = interior. */
void orc_interior_mask(const orc_scene *s, const orc_params *p, uint8_t *mask)
{
   int sroot = isqrt_exact(p->sample_count);
   if (sroot * sroot != p->sample_count) return;
   int maxRec = p->max_recurse > 0 ? p->max_recurse : K_MAX_RECURSE;
   /* the centre first, then the corners, then the edge midpoints of the square the two draws span */
   static const float draws[9][2] = {{0.5f, 0.5f}, {0.0f, 0.0f}, {0.0f, 1.0f}, {1.0f, 0.0f}, {1.0f, 1.0f},
                                     {0.5f, 0.0f}, {0.5f, 1.0f}, {0.0f, 0.5f}, {1.0f, 0.5f}};
#ifdef _OPENMP
   if (p->threads > 0) omp_set_num_threads(p->threads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
   for (int y = 0; y < p->height; y++) {
      for (int x = 0; x < p->width; x++) {
         int same = 1;
         for (int st = 0; st < sroot * sroot && same; st++) {
            int sx = x * sroot + st % sroot, sy = y * sroot + st / sroot;
            uint64_t first = 0;
            for (int k = 0; k < 9 && same; k++) {
               uint64_t sig = 0;
               shade_ctx cx; cx.s = s; cx.shading = p->shading; cx.cn = NULL; cx.sig = &sig;
               ray_t ray = camera_ray_at(s, p, sroot, sx, sy, draws[k][0], draws[k][1]);
               float t = FLT_MAX; int closestGeom = -1;
               getClosestIntersection(s, &ray, &closestGeom, &t, -FLT_MAX, NULL);
               sig_mix(&cx, (uint64_t)(int64_t)closestGeom);
               if (closestGeom != -1) (void)shadeObject(&cx, maxRec, closestGeom, t, ray);
               if (k == 0) first = sig; else if (sig != first) same = 0;
            }
         }
         mask[(size_t)y * p->width + x] = (uint8_t)same;
      }
   }
}

/* The nearest intersection of every primary ray with no tree at all: every BVH primitive and every plane is tested and
 * the smallest accepted parameter kept (accepted as kernel.cu:124-126 accepts: above -EPSILON).  No tie rule and no
 * visiting order, so this is not what the reference reports where hits lie within EQUAL_EPSILON of each other; it is
 * the ground truth a tree of any shape is checked against: a tree that lost a primitive or fitted a box too small
 * reports a larger t (or a miss) somewhere.  prim / t: one entry per sample, row-major. */
void orc_brute_primary(const orc_scene *s, const orc_params *p, int32_t *prim, float *tout)
{
   int sroot = isqrt_exact(p->sample_count);
   if (sroot * sroot != p->sample_count) return;
   int resWidth = p->width * sroot, resHeight = p->height * sroot;
#ifdef _OPENMP
   if (p->threads > 0) omp_set_num_threads(p->threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
   for (int y = 0; y < resHeight; y++) {
      for (int x = 0; x < resWidth; x++) {
         ray_t ray = camera_ray(s, p, sroot, x, y);
         float t = FLT_MAX; int best = -1;
         for (int g = 0; g < s->n_geom + s->n_planes; g++) {
            float dist = geom_getIntersection(geom_by_id(s, g), &ray);
            if (dist < t && isFloatAboveZero(dist)) { t = dist; best = g; }
         }
         size_t index = (size_t)y * resWidth + x;
         prim[index] = best; tout[index] = t;
      }
   }
}

/* ---- per-primitive views ------------------------------------------------ */
float orc_prim_intersect(const orc_scene *s, int32_t prim, const float o[3], const float d[3])
{
   ray_t r; r.o = vload(o); r.d = vload(d);
   return geom_getIntersection(geom_by_id(s, prim), &r);
}
void orc_prim_normal(const orc_scene *s, int32_t prim, const float o[3], const float d[3], float t, float n[3])
{
   ray_t r; r.o = vload(o); r.d = vload(d);
   store3(n, geom_getNormalAt(geom_by_id(s, prim), &r, t));
}
void orc_prim_uv(const orc_scene *s, int32_t prim, const float o[3], const float d[3], float t, float uv[2])
{
   ray_t r; r.o = vload(o); r.d = vload(d);
   orc_uv_at(geom_by_id(s, prim), &r, t, uv);
}
void orc_texture_sample(const rt_texture *tex, float u, float v, float rgb[3]) { orc_tex2d(tex, u, v, rgb); }
void orc_prim_bbox(const orc_scene *s, int32_t prim, float mn[3], float mx[3])
{
   bbox_t b = geom_getBoundingBox(geom_by_id(s, prim));
   store3(mn, b.min); store3(mx, b.max);
}
void orc_prim_center(const orc_scene *s, int32_t prim, float c[3])
{
   store3(c, geom_getCenter(geom_by_id(s, prim)));
}
float orc_box_intersect(const float o[3], const float d[3], const float mn[3], const float mx[3])
{
   ray_t r; r.o = vload(o); r.d = vload(d);
   return boxIntersect(&r, vload(mn), vload(mx));
}
void orc_shade(int shading, const float mat_clr[3], float amb, float dif, float spec, float rough,
               const float eye[3], const float light_dir[3], const float light_clr[3],
               const float normal[3], int in_shadow, float out[3])
{
   store3(out, shader_shade(shading, vload(mat_clr), amb, dif, spec, rough, vload(eye), vload(light_dir),
                            vload(light_clr), vload(normal), in_shadow));
}
