/* The host side of scene construction (parser, flattening from several threads, the reference's tree with its radix-sorted
 * rounds, the surface-area builder) under AddressSanitizer and UndefinedBehaviorSanitizer: every scene file named on the
 * command line with each of the tree options, then a synthetic scene large enough for the threaded paths.  Built and run
 * by tests/test_native_checks.py; a sanitizer report makes the run fail. */
#include "rt_common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
/* the second tree left to its thread past the call: waited for and compared with the one built within the call, and
 * once more freed without anybody waiting */
static bool deferred_tree_matches(const rt_scene_desc *d, const rt_flat_scene &fs, const char *name)
{
   rt_flat_scene late, dropped;
   if (rt_flatten_scene_ex(d, RT_FLATTEN_DEFER_FILTER, &late) != RT_OK) return false;
   if (rt_flatten_scene_ex(d, RT_FLATTEN_DEFER_FILTER, &dropped) != RT_OK) { rt_flat_scene_free(&late); return false; }
   rt_flat_scene_free(&dropped);
   const bool was_pending = late.pending_filter != nullptr;
   (void)rt_flat_scene_finish(&late); (void)rt_flat_scene_finish(&late);
   const bool same = late.pending_filter == nullptr && late.n_filter_nodes == fs.n_filter_nodes && late.filter_depth == fs.filter_depth &&
                     (fs.n_filter_nodes == 0 || memcmp(late.filter_nodes, fs.filter_nodes, sizeof(rt_bvh_node) * (size_t)fs.n_filter_nodes) == 0);
   printf("%s deferred: pending %d same %d\n", name, (int)was_pending, (int)same);
   rt_flat_scene_free(&late);
   return same;
}

int main(int argc, char **argv)
{
   for (int a = 1; a < argc; a++) {
      rt_scene_desc *d = nullptr;
      if (rt_parse_pov(argv[a], &d) != RT_OK) { printf("%s: parse failed: %s\n", argv[a], rt_last_error()); continue; }
      for (int flags : {0, (int)RT_SCENE_FAST_BVH, (int)RT_SCENE_NO_FILTER, (int)RT_SCENE_GPU_BVH}) {
         rt_flat_scene fs;
         int rc = rt_flatten_scene_ex(d, flags, &fs);
         printf("%s flags %d: rc %d prims %d nodes %d filter %d depth %d reach %g\n", argv[a], flags, rc, fs.n_prims, fs.n_nodes, fs.n_filter_nodes, fs.tree_depth, fs.origin_reach);
         if (rc == RT_OK && flags == 0 && !deferred_tree_matches(d, fs, argv[a])) { printf("DEFERRED TREE DIFFERS\n"); return 1; }
         if (rc == RT_OK) rt_flat_scene_free(&fs);
      }
      rt_scene_desc_free(d);
   }
   /* a large synthetic scene: the parallel fill and the material table */
   {
      const int n = 70000;
      std::vector<rt_triangle> tr((size_t)n);
      memset(tr.data(), 0, sizeof(rt_triangle) * (size_t)n);
      for (int i = 0; i < n; i++) {
         float x = (float)(i % 1000) * 0.02f, z = (float)(i / 1000) * 0.02f;
         float p1[3] = {x, sinf(x), z}, p2[3] = {x + 0.02f, sinf(x), z}, p3[3] = {x, sinf(x), z + 0.02f};
         memcpy(tr[(size_t)i].p1, p1, 12); memcpy(tr[(size_t)i].p2, p2, 12); memcpy(tr[(size_t)i].p3, p3, 12);
         for (int k = 0; k < 4; k++) { tr[(size_t)i].mod.trans[k * 4 + k] = 1.0f; tr[(size_t)i].mod.inv_trans[k * 4 + k] = 1.0f; }
         tr[(size_t)i].mod.mat.clr[0] = (float)(i % 977) / 977.0f; tr[(size_t)i].mod.mat.tex_id = -1; tr[(size_t)i].mod.mat.dif = 0.6f;
      }
      rt_scene_desc d; memset(&d, 0, sizeof d);
      d.triangles = tr.data(); d.n_triangles = n;
      for (int flags : {(int)RT_SCENE_FAST_BVH, 0}) {
         rt_flat_scene fs;
         int rc = rt_flatten_scene_ex(&d, flags, &fs);
         printf("synthetic flags %d: rc %d prims %d nodes %d filter %d materials %d\n", flags, rc, fs.n_prims, fs.n_nodes, fs.n_filter_nodes, fs.n_materials);
         if (rc == RT_OK && flags == 0 && !deferred_tree_matches(&d, fs, "synthetic")) { printf("DEFERRED TREE DIFFERS\n"); return 1; }
         if (rc == RT_OK) rt_flat_scene_free(&fs);
      }
   }
   return 0;
}
