/*
 * rt_common.h -- records of the flat device scene, shared by the host builder
 * (rt_scene_build.cpp) and the HIP kernels (rt_render.hip).
 *
 * Every record is a multiple of 16 bytes and the hot ones (BVH node, primitive
 * intersection record) are exactly one 64-byte line, so a lane fetches a record
 * with four 16-byte loads from one aligned line.
 */
#ifndef RT_COMMON_H
#define RT_COMMON_H

#include <stdint.h>
#include "raytracer_amd.h"

/* primitive classes, numbered in the reference's geomList order (kernel.cu:419-456) */
enum { RT_PRIM_SPHERE = 0, RT_PRIM_TRIANGLE = 1, RT_PRIM_BOX = 2, RT_PRIM_SMOOTH = 3 };

#define RT_FLAG_XFORM 1u      /* inv_trans is not the identity: the ray must be transformed (Geometry.h:16-18) */

#define RT_ROOT_EMPTY INT32_MIN
#define RT_FILTER_DEPTH 16       /* levels the miss filter's tree may have before the primitive count forces more */
#define RT_FILTER_MIN_PRIMS 64   /* scenes smaller than this get no miss filter: their reference tree is a handful of boxes */

/* Intersection record.
 *   sphere   g[0..2] centre, g[3] radius                          (Sphere.h:42-66)
 *   triangle g[0..8] p1 p2 p3, g[9..11] normalised face normal    (Triangle.h:11-22, 55-103)
 *   box      g[0..2] minP, g[3..5] maxP                           (Box.h:13-19, 37-39)          */
typedef struct rt_dev_prim {
   float g[12];
   uint32_t type;
   uint32_t flags;
   uint32_t material;
   uint32_t pad;
} rt_dev_prim;

/* Rows 0..2 of inv_trans (the 4th row never reaches the .xyz the reference keeps, Ray.h:8-12):
 * r[k][c] = inv_trans[c][k], i.e. r[k] = {m[0][k], m[1][k], m[2][k], m[3][k]}. */
typedef struct rt_dev_xform {
   float r[3][4];
} rt_dev_xform;

/* Shading record: what getNormalAt needs beyond the intersection record.
 *   sphere   s[0..2] world-space centre                           (Sphere.h:13-14, 19-24)
 *   triangle s[0..2] world-space normal = (n,0) * inv_trans       (Triangle.h:20-26)
 *   smooth   s[0..8] wsN1 wsN2 wsN3                               (SmoothTriangle.h:21-38)
 *   both triangle kinds keep their six texture coordinates in s[9..11], uv[0..2]
 *   box      s[0..8] columns 0..2 of trans, rows 0..2             (Box.h:59)                    */
typedef struct rt_dev_shade {
   float s[12];
   float uv[3];                  /* triangles: s[9..11], uv[0..2] = vt1, vt2, vt3 (Triangle.h:44-52) */
   uint32_t pad;
} rt_dev_shade;

/* One texture of the scene: its texels start at `offset` (in texels) of the texel array. */
typedef struct rt_dev_texinfo { uint32_t offset; int32_t width, height; uint32_t pad; } rt_dev_texinfo;

/* Plane (not in the BVH; scanned linearly after it, kernel.cu:189-196). */
typedef struct rt_dev_plane {
   float n[3]; float d;          /* Plane.h:39-50 */
   rt_dev_xform inv;
   float wsN[3];                 /* Plane.h:15-16 */
   uint32_t flags;
   uint32_t material;
   uint32_t pad[3];
} rt_dev_plane;

typedef struct rt_dev_light { float pos[3]; float pad0; float clr[3]; float pad1; } rt_dev_light;

/* The host-side result of flattening a scene description. */
typedef struct rt_flat_scene {
   int32_t n_prims, n_planes, n_lights, n_nodes, n_materials;
   int32_t root, tree_depth;
   rt_bvh_node *nodes;         /* n_nodes */
   rt_bvh_node *filter_nodes;  /* n_filter_nodes: the miss filter's tree (root 0), or none; leaves refer to primitive ids like `nodes` */
   int32_t n_filter_nodes, filter_depth;
   rt_dev_prim *prims;         /* n_prims, indexed by primitive id */
   rt_dev_xform *xforms;       /* n_prims */
   rt_dev_shade *shades;       /* n_prims */
   rt_dev_plane *planes;       /* n_planes */
   rt_material *materials;     /* n_materials (deduplicated) */
   rt_dev_light *lights;       /* n_lights */
   int32_t *leaf_order;        /* n_prims: geomList after formBVH's sort rounds */
   float *prim_boxes;          /* n_prims x {min xyz, max xyz}: getBoundingBox() of every primitive (the device builder's input) */
   int32_t n_textures;
   rt_dev_texinfo *tex_info;   /* n_textures */
   uint8_t *texels;            /* 4 bytes per texel, all textures back to back */
   size_t n_texels;
   int32_t any_xform;
   int32_t bounds_in_range;    /* every node bound is finite with magnitude <= 2^40 (or zero) */
   /* Sphere::intersects (Sphere.h:42-66) takes the difference of two products of the size (distance to the sphere)^2:
    * what it reports as the hit lies off the sphere by up to about 2^-22 * distance^2 / radius (more for a sphere
    * squeezed by its transform).  The largest of 1 / radius over the scene's spheres, in world units, with that
    * squeeze folded in; 0 for a scene without spheres. */
   float sphere_coeff;
   float origin_reach;      /* rays that start within this many units of the origin of the coordinates, on every axis, may use the second tree (its boxes are widened for them) */
   void *pending_filter;    /* RT_FLATTEN_DEFER_FILTER: the second tree's thread is still at work (filter_nodes is empty until rt_flat_scene_finish); NULL otherwise */
} rt_flat_scene;

#ifdef __cplusplus
extern "C" {
#endif
/* Host: objects in initScene order + reference-topology BVH + flattening.
 * Returns RT_OK or an RT_ERR_* code; on success the arrays are malloc'ed. */
int rt_flatten_scene(const rt_scene_desc *desc, rt_flat_scene *out);
/* flags: RT_SCENE_FAST_BVH builds the opt-in surface-area-heuristic tree instead of the reference's */
int rt_flatten_scene_ex(const rt_scene_desc *desc, int flags, rt_flat_scene *out);
void rt_flat_scene_free(rt_flat_scene *fs);
/* (internal flag of rt_flatten_scene_ex, beside the RT_SCENE_* ones) the second tree's thread runs on after the call
 * returns; rt_flat_scene_finish waits for it and fills in filter_nodes, n_filter_nodes and filter_depth.  Every copy
 * of the struct made before that call must be made again after it. */
#define RT_FLATTEN_DEFER_FILTER (1 << 30)
int rt_flat_scene_finish(rt_flat_scene *fs);
void rt_set_error(const char *fmt, ...);
/* device-side linear BVH over n >= 2 boxes (rt_lbvh.hip) */
int rt_lbvh_build(const float *boxes_host, int n, rt_bvh_node **nodes_out, int32_t **order_out, int *depth_out, float *ms_out);
#ifdef __cplusplus
}
#endif

#endif
