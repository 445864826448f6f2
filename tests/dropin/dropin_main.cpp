/*
 * A caller in the shape of the reference's driver (raytrace.cpp:26-58): fill a TKSceneData, allocate
 * the uchar4 frame, call launch_kernel, keep the bytes.  It includes only the drop-in header and links
 * libraytracer_amd.so; the GPU test runs it and compares the frame with the one the C ABI renders from
 * the same scene.
 *
 *   dropin_main <out.rgba> <width> <height> <sampleCount> <phong|cook>
 */
#include "raytracer_tokens.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

static TKModifier modifier(rt_vec3 clr, float filter, TKFinish fin)
{
   TKModifier m;
   m.pig = TKPigment(clr, filter);
   m.fin = fin;
   return m;
}

int main(int argc, char **argv)
{
   if (argc < 6) { fprintf(stderr, "usage: %s out.rgba width height samples phong|cook\n", argv[0]); return 2; }
   int width = atoi(argv[2]), height = atoi(argv[3]), samples = atoi(argv[4]);
   ShadingType stype = strcmp(argv[5], "cook") == 0 ? COOK_TORRANCE : PHONG;

   TKSceneData data;
   data.camera = new TKCamera(rt_vec3(0, 1, 9), rt_vec3(0, 1, 0), rt_vec3(1.33333f, 0, 0), rt_vec3(0, 0.5f, 0));
   data.pointLights.push_back(TKPointLight(rt_vec3(-6, 8, 10), rt_vec3(1.2f, 1.2f, 1.2f)));
   data.pointLights.push_back(TKPointLight(rt_vec3(7, 5, 4), rt_vec3(0.4f, 0.3f, 0.3f)));

   TKSphere s;
   s.p = rt_vec3(-1.5f, 0.5f, 0); s.r = 1.5f;
   s.mod = modifier(rt_vec3(0.9f, 0.2f, 0.2f), 0.0f, TKFinish(0.2f, 0.5f, 0.6f, 0.05f, 0.4f, 0.0f, 1.0f));
   data.spheres.push_back(s);
   s.p = rt_vec3(2.0f, 0.2f, 1.5f); s.r = 1.0f;
   s.mod = modifier(rt_vec3(0.9f, 0.9f, 1.0f), 0.7f, TKFinish(0.1f, 0.3f, 0.5f, 0.02f, 0.0f, 1.0f, 1.33f));
   data.spheres.push_back(s);

   TKPlane p;
   p.n = rt_vec3(0, 1, 0); p.d = -1.0f;
   p.mod = modifier(rt_vec3(0.3f, 0.5f, 0.8f), 0.0f, TKFinish(0.3f, 0.6f, 0.0f, 0.05f, 0.2f, 0.0f, 1.0f));
   data.planes.push_back(p);

   TKTriangle t;
   t.p1 = rt_vec3(-4, -1, -3); t.p2 = rt_vec3(4, -1, -3); t.p3 = rt_vec3(0, 5, -4);
   t.mod = modifier(rt_vec3(0.2f, 0.8f, 0.3f), 0.0f, TKFinish());
   data.triangles.push_back(t);

   TKBox b;
   b.p1 = rt_vec3(0.2f, -1.0f, -1.5f); b.p2 = rt_vec3(1.4f, 0.4f, -0.4f);
   b.mod = modifier(rt_vec3(0.8f, 0.7f, 0.1f), 0.0f, TKFinish(0.2f, 0.7f, 0.3f, 0.1f, 0.0f, 0.0f, 1.0f));
   data.boxes.push_back(b);

   RT_UCHAR4 *output = (RT_UCHAR4 *)malloc((size_t)width * height * sizeof(RT_UCHAR4));     /* raytrace.cpp:41 */
   memset(output, 7, (size_t)width * height * sizeof(RT_UCHAR4));
   launch_kernel(&data, stype, width, height, output, samples);                            /* raytrace.cpp:42 */

   FILE *fp = fopen(argv[1], "wb");
   if (!fp) { perror(argv[1]); return 1; }
   fwrite(output, sizeof(RT_UCHAR4), (size_t)width * height, fp);
   fclose(fp);
   free(output);
   delete data.camera;
   return 0;
}
