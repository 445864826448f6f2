/*
 * rt_oracle.h -- CPU restatement of the reference's per-pixel render path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under raytracer_amd/ may include, link or
 * call this; it exists so that tests, the smoke check and the benchmark's
 * cpu_baseline leg can check / time the reference algorithm on host cores.
 *
 * Every function follows the reference file:line named beside it.  Parity of
 * this file with the reference is pinned by (see tests/):
 *   - the reference's own primitive / bounding-box / shader classes compiled
 *     from /root/reference with g++ (oracle/_ref/libref.so), bit-exact;
 *   - the BVH sort-schedule trace the reference keeps in its `output` file;
 *   - the reference's .tga renders (loose: they were made with nvcc fast-math
 *     and cuRAND jitter).
 * The traversal / shading recursion of kernel.cu itself cannot be compiled
 * here (it needs nvcc, curand_kernel.h and texture references).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "raytracer_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

typedef struct orc_counters {
   uint64_t rays_primary, rays_shadow, rays_reflect, rays_refract;
   uint64_t shadow_selfmiss;     /* shadow traversals issued although the self test missed */
   uint64_t shade_nodes;
   uint64_t box_tests;           /* boxIntersect calls made by the traversal */
   uint64_t node_first_visits;   /* internal node, neither child explored yet */
   uint64_t node_revisits;       /* internal node after a pop */
   uint64_t prim_tests;          /* leaf getIntersection calls (planes not counted) */
   uint64_t plane_tests;
   uint64_t tie_events;          /* isFloatEqual(t, dist) branch taken */
   int32_t  max_stack;
} orc_counters;

typedef struct orc_params {
   int32_t width, height, sample_count, shading, jitter;
   uint32_t seed;
   int32_t max_recurse;   /* 0 = 6 */
   int32_t threads;       /* OpenMP threads, 0 = default */
} orc_params;

/* initScene (kernel.cu:384-464) + formBVH (bvh.cpp:30-103) on the host. */
orc_scene *orc_scene_create(const rt_scene_desc *desc);
void orc_scene_destroy(orc_scene *s);

int orc_n_bvh_prims(const orc_scene *s);
int orc_n_planes(const orc_scene *s);

/* The sorted geomList after formBVH's sort rounds (primitive ids). */
const int32_t *orc_leaf_order(const orc_scene *s);

/* Trace of the sort rounds in the format of the reference's `output` file:
 * "sorting from A to B" per split range, "using axis K" per round.
 * Returns bytes written (excluding NUL). */
size_t orc_bvh_trace(const orc_scene *s, char *buf, size_t cap);

/* The tree flattened the way include/raytracer_amd.h documents rt_bvh_node
 * (pre-order, children's boxes in the parent).  Returns n_nodes; *root gets
 * the root reference (node 0, ~prim for a single leaf, INT32_MIN if empty). */
int orc_flatten_bvh(const orc_scene *s, rt_bvh_node *nodes, int cap, int32_t *root, int32_t *depth);

/* getClosestIntersection (kernel.cu:83-200).  prim = -1 on miss. */
void orc_closest(const orc_scene *s, const float o[3], const float d[3], float to_beat,
                 int32_t *prim, float *t, orc_counters *c);

/* Per-primitive pieces (used to pin this file against the real classes). */
float orc_prim_intersect(const orc_scene *s, int32_t prim, const float o[3], const float d[3]);  /* Geometry::getIntersection, Geometry.h:16-18 */
void  orc_prim_normal(const orc_scene *s, int32_t prim, const float o[3], const float d[3], float t, float n[3]); /* getNormalAt */
void  orc_prim_uv(const orc_scene *s, int32_t prim, const float o[3], const float d[3], float t, float uv[2]);    /* UVAt, Triangle.h:44-52 */
void  orc_texture_sample(const rt_texture *tex, float u, float v, float rgb[3]);               /* tex2D as bound at kernel.cu:590-593 */
void  orc_prim_bbox(const orc_scene *s, int32_t prim, float mn[3], float mx[3]);                 /* getBoundingBox */
void  orc_prim_center(const orc_scene *s, int32_t prim, float c[3]);                             /* getCenter */
float orc_box_intersect(const float o[3], const float d[3], const float mn[3], const float mx[3]); /* boxIntersect, GeometryUtil.h:9-51 */
void  orc_shade(int shading, const float mat_clr[3], float amb, float dif, float spec, float rough,
                const float eye[3], const float light_dir[3], const float light_clr[3],
                const float normal[3], int in_shadow, float out[3]);                             /* PhongShader.h:7-25 / CookTorranceShader.h:7-46 */

/* Camera ray of sample (x, y) on the (W*s) x (H*s) grid, kernel.cu:477-498. */
void orc_camera_ray(const orc_scene *s, const orc_params *p, int x, int y, float o[3], float d[3]);

/* rayTrace + averageBufferColors (kernel.cu:515-570).  Any output may be NULL.
 *   rgb_samples : (W*s)*(H*s)*3 floats, clamped sample colours
 *   rgba        : W*H*4 bytes
 *   prim, t     : (W*s)*(H*s) primary closest hits  */
void orc_render(const orc_scene *s, const orc_params *p, float *rgb_samples, uint8_t *rgba,
                int32_t *prim, float *t, orc_counters *c);

/* The two uniform draws of a sample, RT_JITTER_HASH flavour, in (0, 1]. */
void orc_jitter_uniforms(uint32_t seed, uint32_t index, float *r1, float *r2);
/* 1: Sphere.h:55,63 read as a host compile without the float sqrt overload does (in double); 0 (default): binary32 */
void orc_set_sphere_sqrt_double(int on);
/* 1 per pixel of a 1-sample frame whose shade tree makes the same hit / miss / shadow decisions wherever the jitter puts the sample */
void orc_interior_mask(const orc_scene *s, const orc_params *p, uint8_t *mask);
/* nearest intersection of every primary ray by testing every primitive and plane, no tree */
void orc_brute_primary(const orc_scene *s, const orc_params *p, int32_t *prim, float *t);

#ifdef __cplusplus
}
#endif
#endif
