/*
 * rt_launch_kernel.cpp -- launch_kernel, the reference's device entry point
 * (kernel.h:30-31, kernel.cu:689-761), on top of the flat C ABI: convert the
 * token vectors to POD arrays, build the scene on the device, render, copy the
 * frame into the caller's buffer, free everything (the reference leaks its
 * device allocations and relies on process exit).
 */
#include "raytracer_amd.h"
#include "raytracer_tokens.h"
#include "rt_common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

static_assert(sizeof(rt_vec3) == 12 && sizeof(rt_vec2) == 8 && sizeof(rt_mat4) == 64, "GLM-compatible layout");
static_assert(sizeof(TKModifier) == 20 + 28 + 128, "TKModifier layout (TokenData.h:46-51)");
static_assert(sizeof(TKSphere) == 16 + sizeof(TKModifier), "TKSphere layout");
static_assert(sizeof(TKTriangle) == 36 + 24 + sizeof(TKModifier), "TKTriangle layout");
static_assert(sizeof(TKSmoothTriangle) == 72 + 24 + sizeof(TKModifier), "TKSmoothTriangle layout");

namespace {
void put3(float *d, const rt_vec3 &v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
void put2(float *d, const rt_vec2 &v) { d[0] = v.x; d[1] = v.y; }
void put_mod(rt_modifier *d, const TKModifier &m)
{
   /* Material(clr, pig.f, amb, dif, spec, rough, refl, refr, ior[, texId]): kernel.cu:414, 422, 431, 441, 450 */
   put3(d->mat.clr, m.pig.clr);
   d->mat.alpha = m.pig.f;
   d->mat.amb = m.fin.amb; d->mat.dif = m.fin.dif; d->mat.spec = m.fin.spec; d->mat.rough = m.fin.rough;
   d->mat.refl = m.fin.refl; d->mat.refr = m.fin.refr; d->mat.ior = m.fin.ior;
   d->mat.tex_id = m.pig.texId;
   memcpy(d->trans, m.trans.m, 64);
   memcpy(d->inv_trans, m.invTrans.m, 64);
}
[[noreturn]] void die(const char *what)
{
   /* the reference's HANDLE_ERROR / checkCUDAError print and exit (cudaError.h:6-22) */
   fprintf(stderr, "raytracer_amd error: %s: %s.\n", what, rt_last_error());
   exit(EXIT_FAILURE);
}
}

extern "C" void launch_kernel(TKSceneData *data, ShadingType stype, int width, int height, RT_UCHAR4 *output, int sampleCount)
{
   int sqrSampleCount = (int)sqrt((double)sampleCount);
   if (sqrSampleCount * sqrSampleCount != sampleCount) {
      printf("Invalid sample count: %d. "
             "Sample count for anti aliasing must have an integer square root", sampleCount);
      return;
   }
   if (!data || !data->camera || !output) { rt_set_error("launch_kernel: scene, camera and output must be non-null"); die("launch_kernel"); }

   std::vector<rt_sphere> spheres(data->spheres.size());
   std::vector<rt_triangle> tris(data->triangles.size());
   std::vector<rt_box> boxes(data->boxes.size());
   std::vector<rt_smooth_triangle> smooth(data->smoothTriangles.size());
   std::vector<rt_plane> planes(data->planes.size());
   std::vector<rt_light> lights(data->pointLights.size());
   for (size_t i = 0; i < spheres.size(); i++) { put3(spheres[i].c, data->spheres[i].p); spheres[i].r = data->spheres[i].r; put_mod(&spheres[i].mod, data->spheres[i].mod); }
   for (size_t i = 0; i < tris.size(); i++) {
      const TKTriangle &t = data->triangles[i];
      put3(tris[i].p1, t.p1); put3(tris[i].p2, t.p2); put3(tris[i].p3, t.p3); put_mod(&tris[i].mod, t.mod);
      put2(tris[i].vt1, t.vt1); put2(tris[i].vt2, t.vt2); put2(tris[i].vt3, t.vt3);
   }
   for (size_t i = 0; i < boxes.size(); i++) { put3(boxes[i].p1, data->boxes[i].p1); put3(boxes[i].p2, data->boxes[i].p2); put_mod(&boxes[i].mod, data->boxes[i].mod); }
   for (size_t i = 0; i < smooth.size(); i++) {
      const TKSmoothTriangle &t = data->smoothTriangles[i];
      put3(smooth[i].p1, t.p1); put3(smooth[i].p2, t.p2); put3(smooth[i].p3, t.p3);
      put3(smooth[i].n1, t.n1); put3(smooth[i].n2, t.n2); put3(smooth[i].n3, t.n3);
      put_mod(&smooth[i].mod, t.mod);
      put2(smooth[i].vt1, t.vt1); put2(smooth[i].vt2, t.vt2); put2(smooth[i].vt3, t.vt3);
   }
   for (size_t i = 0; i < planes.size(); i++) { put3(planes[i].n, data->planes[i].n); planes[i].d = data->planes[i].d; put_mod(&planes[i].mod, data->planes[i].mod); }
   for (size_t i = 0; i < lights.size(); i++) { put3(lights[i].pos, data->pointLights[i].pos); put3(lights[i].clr, data->pointLights[i].clr); }

   rt_scene_desc d; memset(&d, 0, sizeof d);
   put3(d.camera.pos, data->camera->pos); put3(d.camera.up, data->camera->up);
   put3(d.camera.right, data->camera->right); put3(d.camera.look_at, data->camera->lookAt);
   d.has_camera = 1;
   d.n_spheres = (int32_t)spheres.size(); d.spheres = spheres.data();
   d.n_triangles = (int32_t)tris.size(); d.triangles = tris.data();
   d.n_boxes = (int32_t)boxes.size(); d.boxes = boxes.data();
   d.n_smooth_triangles = (int32_t)smooth.size(); d.smooth_triangles = smooth.data();
   d.n_planes = (int32_t)planes.size(); d.planes = planes.data();
   d.n_lights = (int32_t)lights.size(); d.lights = lights.data();

   /* Textures: the reference reads "blitz.bmp" from the working directory and binds it to every textured
    * primitive, whatever name the scene gave (kernel.cu:580-598).  Same here, but only when a texture is used. */
   std::vector<rt_texture> textures(data->textureMap.size());
   rt_texture blitz; memset(&blitz, 0, sizeof blitz);
   if (!textures.empty()) {
      if (rt_load_bmp("blitz.bmp", &blitz) != RT_OK) die("texture load failed");
      for (size_t i = 0; i < textures.size(); i++) { textures[i] = blitz; snprintf(textures[i].name, sizeof textures[i].name, "blitz.bmp"); }
   }
   d.n_textures = (int32_t)textures.size(); d.textures = textures.empty() ? NULL : textures.data();

   const char *dev_env = getenv("RT_DEVICE");
   const char *jit_env = getenv("RT_JITTER");
   /* RT_DEVICES = "0,1,2,3": the frame is split over those devices (rt_multi_render) */
   std::vector<int> devices;
   if (const char *list = getenv("RT_DEVICES")) {
      for (const char *c = list; *c;) {
         char *end = NULL;
         long v = strtol(c, &end, 10);
         if (end == c) break;
         devices.push_back((int)v);
         c = *end == ',' ? end + 1 : end;
      }
   }
   rt_render_params p; memset(&p, 0, sizeof p);
   p.width = width; p.height = height; p.sample_count = sampleCount;
   p.shading = stype == COOK_TORRANCE ? RT_COOK_TORRANCE : RT_PHONG;
   p.jitter = (jit_env && atoi(jit_env) == 0) ? RT_JITTER_NONE : RT_JITTER_HASH;
   p.seed = 0; p.max_recurse = RT_MAX_RECURSE; p.tile_first = 0; p.tile_stride = 1; p.out_layout = RT_OUT_FRAME;
   int rc;
   if (devices.size() > 1) {
      rt_multi *multi = NULL;
      if (rt_multi_create(&d, devices.data(), (int)devices.size(), RT_SCENE_DEFAULT, &multi) != RT_OK) { free(blitz.rgba); die("scene set-up failed"); }
      rc = rt_multi_render(multi, &p, (uint8_t *)output, NULL);
      rt_multi_destroy(multi);
   } else {
      rt_scene *scene = NULL;
      const int device = !devices.empty() ? devices[0] : dev_env ? atoi(dev_env) : 0;
      if (rt_scene_create(&d, device, RT_SCENE_DEFAULT, &scene) != RT_OK) { free(blitz.rgba); die("scene set-up failed"); }
      rc = rt_render(scene, &p, (uint8_t *)output, NULL);
      rt_scene_destroy(scene);
   }
   free(blitz.rgba);
   if (rc != RT_OK) die("RayTrace kernel failed");
}
