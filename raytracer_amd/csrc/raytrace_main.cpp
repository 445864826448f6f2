/*
 * raytrace -- command-line driver with the reference's flags (raytrace.cpp:18-120,
 * README:16-27):  raytrace [width] [height] -I scene.pov [-O out.tga] [-p|-t] [-s samples]
 * Parses the scene, renders it on the GPU through the C ABI and writes a TGA.
 * Additions: -j0 / -j1 turn the sub-pixel jitter off / on (default on, as in
 * the reference), -d N picks the HIP device, -g N renders on N devices (N copies
 * of the scene, image tiles dealt round-robin, rt_multi_render; with fewer
 * devices in the machine the parts share them), -v prints ray and timing
 * statistics.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "raytracer_amd.h"

static void print_input_error()
{
   printf("raytrace: must specify an input file\n");
   printf("Try `raytrace --help` for more information\n");
}

int main(int argc, char *argv[])
{
   int width = 800, height = 600, samples = 1;      /* raytrace.cpp:10-12 */
   const char *in_file = NULL, *out_file = "sample.tga";
   int shading = RT_PHONG, jitter = RT_JITTER_HASH, device = 0, verbose = 0, n_gpus = 0;
   bool have_w = false, have_h = false;
   if (argc < 2) { print_input_error(); return EXIT_FAILURE; }
   if (!strcmp(argv[1], "--help")) {
      printf("raytrace options are:\n");
      printf("\timageWidth\n\timageHeight\n\t-I sample.pov\n\t-O output.tga\n\t-p (phong) | -t (cook-torrance)\n\t-s samples\n");
      printf("An example command to generate a 1024x800 image named \"image.tga\""
             " using \"input.pov\" with phong shading is :\n\n");
      printf("$raytrace 1024 800 -p -I input.pov -O image.tga\n");
      return EXIT_SUCCESS;
   }
   for (int i = 1; i < argc; i++) {
      const char *a = argv[i];
      if (a[0] == '-' && a[1] == 'O') { if (strlen(a) == 2) { if (++i < argc) out_file = argv[i]; } else out_file = a + 2; }
      else if (a[0] == '-' && a[1] == 'I') { if (strlen(a) == 2) { if (++i < argc) in_file = argv[i]; } else in_file = a + 2; }
      else if (a[0] == '-' && a[1] == 's') { if (++i < argc) samples = atoi(argv[i]); }
      else if (a[0] == '-' && a[1] == 'p') shading = RT_PHONG;
      else if (a[0] == '-' && a[1] == 't') shading = RT_COOK_TORRANCE;
      else if (a[0] == '-' && a[1] == 'j') jitter = (a[2] == '0') ? RT_JITTER_NONE : RT_JITTER_HASH;
      else if (a[0] == '-' && a[1] == 'd') { if (++i < argc) device = atoi(argv[i]); }
      else if (a[0] == '-' && a[1] == 'g') { if (++i < argc) n_gpus = atoi(argv[i]); }
      else if (a[0] == '-' && a[1] == 'v') verbose = 1;
      else if (!have_w) { width = atoi(a); have_w = true; }
      else if (!have_h) { height = atoi(a); have_h = true; }
   }
   if (!in_file) { print_input_error(); return EXIT_FAILURE; }

   rt_scene_desc *desc = NULL;
   if (rt_parse_pov(in_file, &desc) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); printf("Error parsing file\n"); return EXIT_FAILURE; }
   /* image_map files: looked up in the working directory (where the reference reads its one texture,
    * kernel.cu:580), then next to the scene file */
   for (int i = 0; i < desc->n_textures; i++) {
      rt_texture *t = &desc->textures[i];
      if (rt_load_bmp(t->name, t) == RT_OK) continue;
      std::string dir(in_file);
      size_t slash = dir.find_last_of('/');
      std::string alt = (slash == std::string::npos ? std::string() : dir.substr(0, slash + 1)) + t->name;
      if (rt_load_bmp(alt.c_str(), t) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return EXIT_FAILURE; }
   }
   if (width <= 0 || height <= 0) { fprintf(stderr, "bad image size\n"); return EXIT_FAILURE; }
   std::vector<uint8_t> frame((size_t)width * height * 4);
   rt_render_params p; memset(&p, 0, sizeof p);
   p.width = width; p.height = height; p.sample_count = samples; p.shading = shading; p.jitter = jitter;
   p.tile_stride = 1;
   rt_stats st;
   rt_scene *scene = NULL;
   rt_multi *multi = NULL;
   int rc;
   if (n_gpus > 0) {
      /* part k on device (device + k) mod the number of devices */
      const int visible = rt_device_count();
      std::vector<int> devices((size_t)n_gpus);
      for (int k = 0; k < n_gpus; k++) devices[(size_t)k] = visible > 0 ? (device + k) % visible : 0;
      if (rt_multi_create(desc, devices.data(), n_gpus, RT_SCENE_DEFAULT, &multi) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return EXIT_FAILURE; }
      rc = rt_multi_render(multi, &p, frame.data(), &st);
   } else {
      if (rt_scene_create(desc, device, RT_SCENE_DEFAULT, &scene) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return EXIT_FAILURE; }
      rc = rt_render(scene, &p, frame.data(), &st);
   }
   if (rc == RT_ERR_BAD_SAMPLE_COUNT) { printf("%s", rt_last_error()); return EXIT_SUCCESS; }   /* kernel.cu:707-712: message, no frame */
   if (rc != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return EXIT_FAILURE; }
   if (verbose && scene) {
      rt_scene_info info; rt_scene_get_info(scene, &info);
      fprintf(stderr, "prims %d planes %d lights %d nodes %d depth %d | build %.2f ms upload %.2f ms\n", info.n_bvh_prims, info.n_planes,
              info.n_lights, info.n_nodes, info.tree_depth, info.ms_build_host, info.ms_upload);
   }
   if (verbose) {
      fprintf(stderr, "rays %llu (primary %llu shadow %llu reflect %llu refract %llu) in %.3f ms = %.1f Mrays/s\n",
              (unsigned long long)st.rays_total, (unsigned long long)st.rays_primary, (unsigned long long)st.rays_shadow,
              (unsigned long long)st.rays_reflect, (unsigned long long)st.rays_refract, st.ms_total,
              st.ms_total > 0 ? st.rays_total / (st.ms_total * 1e3) : 0.0);
   }
   rc = rt_write_tga(out_file, width, height, frame.data());
   if (rc != RT_OK) { fprintf(stderr, "%s\n", rt_last_error()); return EXIT_FAILURE; }
   rt_scene_destroy(scene);
   rt_multi_destroy(multi);
   rt_scene_desc_free(desc);
   return EXIT_SUCCESS;
}
