/*
 * ref_driver.cpp -- C entry points over the REAL reference classes.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is compiled, in the build container
 * only, together with the reference's own sources where they lie
 * (/root/reference: POVRayParser.cpp, Image.cpp and the header-only
 * Sphere/Plane/Box/Triangle/SmoothTriangle/BoundingBox/GeometryUtil/
 * PhongShader/CookTorranceShader/PointLight classes) into
 * oracle/_ref/libref.so.  No reference source is copied: the headers are
 * found through -I/root/reference, __device__/__host__/__global__ are
 * defined empty on the command line, and <cuda.h> is the CUDA toolkit header
 * that ships inside the image's triton package.
 *
 * kernel.cu and bvh.cpp are NOT part of this build (they need nvcc,
 * curand_kernel.h and texture references); the traversal and shading
 * recursion are covered by the restatement in rt_oracle.c.
 *
 * The only logic restated here is launch glue: object construction in the
 * order and with the arguments of initScene (kernel.cu:411-461) and the
 * uchar4 -> Image conversion of main (raytrace.cpp:44-55).
 */
/* <math.h> first: in C++ it puts the float overloads of sqrt/pow/exp into the
 * global namespace, which is what nvcc gives device code -- without it the
 * unqualified sqrt(float) in Sphere.h:55,63 would silently bind to the C
 * double version and the quotient would be rounded from double. */
#include <math.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <map>
#include <string>

#include "POVRayParser.h"
#include "Sphere.h"
#include "Box.h"
#include "Plane.h"
#include "Triangle.h"
#include "SmoothTriangle.h"
#include "PointLight.h"
#include "PhongShader.h"
#include "CookTorranceShader.h"
#include "Image.h"

#include "raytracer_amd.h"

namespace {

void put3(float *dst, const glm::vec3 &v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
void put2(float *dst, const glm::vec2 &v) { dst[0] = v.x; dst[1] = v.y; }
glm::vec3 get3(const float *p) { return glm::vec3(p[0], p[1], p[2]); }

void put_mat(float *dst, const glm::mat4 &m)
{
   for (int c = 0; c < 4; c++)
      for (int r = 0; r < 4; r++) dst[c * 4 + r] = m[c][r];
}
glm::mat4 get_mat(const float *p)
{
   glm::mat4 m;
   for (int c = 0; c < 4; c++)
      for (int r = 0; r < 4; r++) m[c][r] = p[c * 4 + r];
   return m;
}

void put_mod(rt_modifier *dst, const TKModifier &mod)
{
   put3(dst->mat.clr, mod.pig.clr);
   dst->mat.alpha = mod.pig.f;
   dst->mat.amb = mod.fin.amb; dst->mat.dif = mod.fin.dif; dst->mat.spec = mod.fin.spec;
   dst->mat.rough = mod.fin.rough; dst->mat.refl = mod.fin.refl; dst->mat.refr = mod.fin.refr;
   dst->mat.ior = mod.fin.ior;
   dst->mat.tex_id = mod.pig.texId;
   put_mat(dst->trans, mod.trans);
   put_mat(dst->inv_trans, mod.invTrans);
}

Material make_material(const rt_material &m)
{
   return Material(get3(m.clr), m.alpha, m.amb, m.dif, m.spec, m.rough, m.refl, m.refr, m.ior, m.tex_id);
}

struct RefScene {
   std::vector<Geometry *> geomList;   /* spheres, triangles, boxes, smooth triangles */
   std::vector<Plane *> planeList;
   std::vector<PointLight *> lights;
};

Geometry *by_id(RefScene *s, int id)
{
   return id < (int)s->geomList.size() ? s->geomList[id] : (Geometry *)s->planeList[id - (int)s->geomList.size()];
}

} // namespace

extern "C" {

/* POVRayParser::parseFile on the real parser; result flattened into the POD description. */
int ref_parse_pov(const char *path, rt_scene_desc **out)
{
   TKSceneData data;
   int status = POVRayParser::parseFile(path, &data);
   if (status != POVRayParser::kSuccess) return status;
   rt_scene_desc *d = (rt_scene_desc *)calloc(1, sizeof(rt_scene_desc));
   if (data.camera) {
      d->has_camera = 1;
      put3(d->camera.pos, data.camera->pos); put3(d->camera.up, data.camera->up);
      put3(d->camera.right, data.camera->right); put3(d->camera.look_at, data.camera->lookAt);
   }
   d->n_spheres = (int)data.spheres.size();
   rt_sphere *sp = (rt_sphere *)calloc(d->n_spheres + 1, sizeof(rt_sphere));
   for (int i = 0; i < d->n_spheres; i++) { put3(sp[i].c, data.spheres[i].p); sp[i].r = data.spheres[i].r; put_mod(&sp[i].mod, data.spheres[i].mod); }
   d->spheres = sp;
   d->n_triangles = (int)data.triangles.size();
   rt_triangle *tr = (rt_triangle *)calloc(d->n_triangles + 1, sizeof(rt_triangle));
   for (int i = 0; i < d->n_triangles; i++) {
      put3(tr[i].p1, data.triangles[i].p1); put3(tr[i].p2, data.triangles[i].p2); put3(tr[i].p3, data.triangles[i].p3);
      put2(tr[i].vt1, data.triangles[i].vt1); put2(tr[i].vt2, data.triangles[i].vt2); put2(tr[i].vt3, data.triangles[i].vt3);
      put_mod(&tr[i].mod, data.triangles[i].mod);
   }
   d->triangles = tr;
   d->n_boxes = (int)data.boxes.size();
   rt_box *bx = (rt_box *)calloc(d->n_boxes + 1, sizeof(rt_box));
   for (int i = 0; i < d->n_boxes; i++) { put3(bx[i].p1, data.boxes[i].p1); put3(bx[i].p2, data.boxes[i].p2); put_mod(&bx[i].mod, data.boxes[i].mod); }
   d->boxes = bx;
   d->n_smooth_triangles = (int)data.smoothTriangles.size();
   rt_smooth_triangle *st = (rt_smooth_triangle *)calloc(d->n_smooth_triangles + 1, sizeof(rt_smooth_triangle));
   for (int i = 0; i < d->n_smooth_triangles; i++) {
      const TKSmoothTriangle &t = data.smoothTriangles[i];
      put3(st[i].p1, t.p1); put3(st[i].p2, t.p2); put3(st[i].p3, t.p3);
      put3(st[i].n1, t.n1); put3(st[i].n2, t.n2); put3(st[i].n3, t.n3);
      put2(st[i].vt1, t.vt1); put2(st[i].vt2, t.vt2); put2(st[i].vt3, t.vt3);
      put_mod(&st[i].mod, t.mod);
   }
   d->smooth_triangles = st;
   d->n_planes = (int)data.planes.size();
   rt_plane *pl = (rt_plane *)calloc(d->n_planes + 1, sizeof(rt_plane));
   for (int i = 0; i < d->n_planes; i++) { put3(pl[i].n, data.planes[i].n); pl[i].d = data.planes[i].d; put_mod(&pl[i].mod, data.planes[i].mod); }
   d->planes = pl;
   d->n_lights = (int)data.pointLights.size();
   rt_light *li = (rt_light *)calloc(d->n_lights + 1, sizeof(rt_light));
   for (int i = 0; i < d->n_lights; i++) { put3(li[i].pos, data.pointLights[i].pos); put3(li[i].clr, data.pointLights[i].clr); }
   d->lights = li;
   d->n_textures = (int)data.textureMap.size();
   d->textures = (rt_texture *)calloc(d->n_textures + 1, sizeof(rt_texture));
   for (std::map<std::string, int>::const_iterator it = data.textureMap.begin(); it != data.textureMap.end(); ++it) {
      std::string name = it->first;          /* the parser's "%s" keeps the closing quote */
      while (!name.empty() && (name[name.size() - 1] == '"' || name[name.size() - 1] == '}')) name.erase(name.size() - 1);
      snprintf(d->textures[it->second].name, sizeof d->textures[it->second].name, "%s", name.c_str());
   }
   *out = d;
   return 0;
}

void ref_scene_desc_free(rt_scene_desc *d)
{
   if (!d) return;
   free((void *)d->spheres); free((void *)d->triangles); free((void *)d->boxes);
   free((void *)d->smooth_triangles); free((void *)d->planes); free((void *)d->lights);
   free(d->textures);
   free(d);
}

/* Objects built by the real constructors, in initScene order (kernel.cu:411-461). */
void *ref_scene_create(const rt_scene_desc *d)
{
   RefScene *s = new RefScene();
   for (int i = 0; i < d->n_planes; i++) {
      const rt_plane &p = d->planes[i];
      s->planeList.push_back(new Plane(p.d, get3(p.n), make_material(p.mod.mat), get_mat(p.mod.trans), get_mat(p.mod.inv_trans)));
   }
   for (int i = 0; i < d->n_spheres; i++) {
      const rt_sphere &q = d->spheres[i];
      s->geomList.push_back(new Sphere(get3(q.c), q.r, make_material(q.mod.mat), get_mat(q.mod.trans), get_mat(q.mod.inv_trans)));
   }
   for (int i = 0; i < d->n_triangles; i++) {
      const rt_triangle &t = d->triangles[i];
      s->geomList.push_back(new Triangle(get3(t.p1), get3(t.p2), get3(t.p3), make_material(t.mod.mat),
                                         get_mat(t.mod.trans), get_mat(t.mod.inv_trans),
                                         glm::vec2(t.vt1[0], t.vt1[1]), glm::vec2(t.vt2[0], t.vt2[1]), glm::vec2(t.vt3[0], t.vt3[1])));
   }
   for (int i = 0; i < d->n_boxes; i++) {
      const rt_box &b = d->boxes[i];
      s->geomList.push_back(new Box(get3(b.p1), get3(b.p2), make_material(b.mod.mat), get_mat(b.mod.trans), get_mat(b.mod.inv_trans)));
   }
   for (int i = 0; i < d->n_smooth_triangles; i++) {
      const rt_smooth_triangle &t = d->smooth_triangles[i];
      s->geomList.push_back(new SmoothTriangle(get3(t.p1), get3(t.p2), get3(t.p3), get3(t.n1), get3(t.n2), get3(t.n3),
                                               make_material(t.mod.mat), get_mat(t.mod.trans), get_mat(t.mod.inv_trans),
                                               glm::vec2(t.vt1[0], t.vt1[1]), glm::vec2(t.vt2[0], t.vt2[1]), glm::vec2(t.vt3[0], t.vt3[1])));
   }
   for (int i = 0; i < d->n_lights; i++)
      s->lights.push_back(new PointLight(get3(d->lights[i].pos), get3(d->lights[i].clr)));
   return s;
}

void ref_scene_destroy(void *h)
{
   RefScene *s = (RefScene *)h;
   if (!s) return;
   for (size_t i = 0; i < s->geomList.size(); i++) delete s->geomList[i];
   for (size_t i = 0; i < s->planeList.size(); i++) delete s->planeList[i];
   for (size_t i = 0; i < s->lights.size(); i++) delete s->lights[i];
   delete s;
}

float ref_prim_intersect(void *h, int prim, const float *o, const float *d)
{
   Ray r(get3(o), get3(d));
   return by_id((RefScene *)h, prim)->getIntersection(r);
}
void ref_prim_normal(void *h, int prim, const float *o, const float *d, float t, float *n)
{
   Ray r(get3(o), get3(d));
   put3(n, by_id((RefScene *)h, prim)->getNormalAt(r, t));
}
void ref_prim_uv(void *h, int prim, const float *o, const float *d, float t, float *uv)
{
   Ray r(get3(o), get3(d));
   glm::vec2 v = by_id((RefScene *)h, prim)->UVAt(r, t);
   uv[0] = v.x; uv[1] = v.y;
}
void ref_prim_bbox(void *h, int prim, float *mn, float *mx)
{
   BoundingBox b = by_id((RefScene *)h, prim)->getBoundingBox();
   put3(mn, b.min); put3(mx, b.max);
}
void ref_prim_center(void *h, int prim, float *c)
{
   put3(c, by_id((RefScene *)h, prim)->getCenter());
}
float ref_box_intersect(const float *o, const float *d, const float *mn, const float *mx)
{
   Ray r(get3(o), get3(d));
   return boxIntersect(r, get3(mn), get3(mx));
}
/* combineBoundingBox (BoundingBox.h:48-52) of two boxes given by their stored corners */
void ref_combine_bbox(const float *amin, const float *amax, const float *bmin, const float *bmax, float *mn, float *mx)
{
   BoundingBox a, b;
   a.min = get3(amin); a.max = get3(amax); b.min = get3(bmin); b.max = get3(bmax);
   BoundingBox c = combineBoundingBox(a, b);
   put3(mn, c.min); put3(mx, c.max);
}
void ref_shade(int stype, const float *matClr, float amb, float dif, float spec, float rough,
               const float *eye, const float *lightDir, const float *lightClr, const float *normal,
               int inShadow, float *out)
{
   PhongShader ph; CookTorranceShader ct;
   Shader *sh = stype == RT_COOK_TORRANCE ? (Shader *)&ct : (Shader *)&ph;
   put3(out, sh->shade(get3(matClr), amb, dif, spec, rough, get3(eye), get3(lightDir), get3(lightClr), get3(normal), inShadow != 0));
}
/* PointLight.h:14-20 */
void ref_light_dir(void *h, int light, const float *point, float *dir)
{
   put3(dir, ((RefScene *)h)->lights[light]->getLightDir(get3(point)));
}
void ref_light_feeler(void *h, int light, const float *point, float *o, float *d)
{
   Ray r = ((RefScene *)h)->lights[light]->getShadowFeeler(get3(point));
   put3(o, r.o); put3(d, r.d);
}

/* main's frame conversion (raytrace.cpp:44-55) feeding the real Image::WriteTga */
int ref_write_tga(const char *path, int w, int h, const unsigned char *rgba)
{
   Image img(w, h);
   for (int x = 0; x < w; x++) {
      for (int y = 0; y < h; y++) {
         color_t clr;
         int idx = y * w + x;
         clr.r = rgba[idx * 4 + 0] / 255.0; clr.g = rgba[idx * 4 + 1] / 255.0;
         clr.b = rgba[idx * 4 + 2] / 255.0; clr.f = 1.0;
         img.pixel(x, y, clr);
      }
   }
   img.WriteTga((char *)path);
   return 0;
}

} // extern "C"
