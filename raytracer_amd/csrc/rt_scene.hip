/*
 * rt_scene.hip -- making and destroying the device scene: allocateGPUScene + formBVH (kernel.cu:572-687, 732-733;
 * bvh.cpp:30-103) as a host build (rt_scene_build.cpp), a renumbering of the primitives by their place among the tree's
 * leaves, and one upload.  Host code only; the kernels are in rt_render.hip.
 */
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>

#include "rt_scene_state.h"

/* entries of a wave's spawn list to begin with; RT_LIST_CAP overrides it (rounded up to a power of two, at least 64):
 * small values make frames overflow and exercise the retry */
static unsigned first_list_cap()
{
   const char *e = getenv("RT_LIST_CAP");
   long v = e ? atol(e) : 0;
   if (v <= 0) return RT_LIST_CAP_FIRST;
   unsigned c = 64;
   while (c < (unsigned long)v && c < RT_LIST_CAP_MAX) c *= 2;
   return c;
}

extern "C" int rt_device_count(void)
{
   int n = 0;
   if (hipGetDeviceCount(&n) != hipSuccess) return 0;
   return n;
}

extern "C" void rt_scene_destroy(rt_scene *s)
{
   if (!s) return;
   if (s->device >= 0) (void)hipSetDevice(s->device);
   if (s->dev_blob) (void)hipFree(s->dev_blob);
   if (s->arena.base) (void)hipFree(s->arena.base);
   if (s->d_ctr) (void)hipFree(s->d_ctr);
   if (s->d_stats) (void)hipFree(s->d_stats);
   if (s->h_stats) (void)hipHostFree(s->h_stats);
   if (s->d_lists) (void)hipFree(s->d_lists);
   if (s->d_frame) (void)hipFree(s->d_frame);
   if (s->h_frame) (void)hipHostFree(s->h_frame);
   if (s->copy_ev_ok) for (hipEvent_t e : s->copy_ev) (void)hipEventDestroy(e);
   if (s->d_spill) (void)hipFree(s->d_spill);
   if (s->d_tile_cost) (void)hipFree(s->d_tile_cost);
   if (s->d_tile_order) (void)hipFree(s->d_tile_order);
   if (s->h_tile_cost && !s->tiles_in_block) (void)hipHostFree(s->h_tile_cost);
   if (s->h_tile_order && !s->tiles_in_block) (void)hipHostFree(s->h_tile_order);
   for (hipEvent_t e : s->timing_events) (void)hipEventDestroy(e);
   if (s->streams_ok) {
      (void)hipStreamDestroy(s->chain);
      (void)hipEventDestroy(s->ev_begin); (void)hipEventDestroy(s->ev_end);
   }
   /* (the host arrays go with their last owner: a scene and its clones share them) */
   if (!s->host_owner) rt_flat_scene_free(&s->host);
   delete s;
}

static void scene_tuning(rt_scene *s);

/* the box all primitives' bounding boxes lie in (for leaving whole tiles of primaries out of a launch, see cull_tiles) */
static void scene_bounds(rt_scene *s)
{
   const rt_flat_scene &h = s->host;
   s->bounds_ok = h.n_prims > 0 && h.prim_boxes != nullptr && h.bounds_in_range;
   s->n_live = -1; s->cull_stage = 0; memset(s->cull_key, 0, sizeof s->cull_key); memset(s->cull_cam, 0, sizeof s->cull_cam);
   for (int j = 0; j < 3; j++) { s->bounds_lo[j] = 1e300; s->bounds_hi[j] = -1e300; }
   for (int k = 0; k < h.n_prims && s->bounds_ok; k++)
      for (int j = 0; j < 3; j++) {
         const double lo = h.prim_boxes[6 * (size_t)k + j], hi = h.prim_boxes[6 * (size_t)k + 3 + j];
         if (!(lo <= hi) || !std::isfinite(lo) || !std::isfinite(hi)) { s->bounds_ok = false; break; }
         if (lo < s->bounds_lo[j]) s->bounds_lo[j] = lo;
         if (hi > s->bounds_hi[j]) s->bounds_hi[j] = hi;
      }
}

extern "C" int rt_scene_create_host(const rt_scene_desc *desc, rt_scene **out)
{
   if (!desc || !out) { rt_set_error("rt_scene_create_host: null argument"); return RT_ERR_BAD_ARGUMENT; }
   *out = nullptr;
   rt_scene *s = new (std::nothrow) rt_scene();
   if (!s) return RT_ERR_BAD_ARGUMENT;
   memset(&s->host, 0, sizeof s->host);
   memset(&s->d, 0, sizeof s->d);
   s->device = -1; s->flags = 0; s->streams_ok = false; s->ms_build_device = 0.0f;
   s->dev_blob = nullptr; s->dev_bytes = 0; s->d_ctr = nullptr; s->d_stats = nullptr; s->h_stats = nullptr; s->node_factor = 2.0f;
   s->d_tile_cost = s->h_tile_cost = nullptr; s->d_tile_order = s->h_tile_order = nullptr; s->order_cap = 0; s->order_valid = false; s->order_dirty = true;
   s->d_lists = nullptr; s->list_cap = first_list_cap(); s->list_waves = 0; s->d_spill = nullptr; s->d_frame = nullptr; s->frame_cap = 0;
   s->camera = desc->camera; s->has_camera = desc->has_camera; s->ms_upload = 0;
   auto t0 = std::chrono::steady_clock::now();
   int st = rt_flatten_scene(desc, &s->host);
   if (st != RT_OK) { delete s; return st; }
   s->host_owner = std::make_shared<HostScene>(); s->host_owner->flat = s->host;
   s->ms_build_host = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
   s->cull_enabled = 1; s->cull_fine = rt_env_int("RT_CULL_FINE", 256, true);
   scene_bounds(s);
   *out = s;
   return RT_OK;
}

/* a scene object for `device` with nothing in it yet */
static int scene_begin(int device, int flags, const rt_camera &camera, int has_camera, const char *who, rt_scene **out)
{
   *out = nullptr;
   int ndev = rt_device_count();
   if (ndev <= 0) { rt_set_error("no HIP device: the render path has no CPU fallback"); return RT_ERR_NO_DEVICE; }
   if (device < 0 || device >= ndev) { rt_set_error("%s: device %d out of range (%d devices)", who, device, ndev); return RT_ERR_BAD_ARGUMENT; }
   HIP_TRY(hipSetDevice(device));

   rt_scene *s = new (std::nothrow) rt_scene();
   if (!s) return RT_ERR_BAD_ARGUMENT;
   memset(&s->host, 0, sizeof s->host);
   s->device = device; s->flags = flags; s->streams_ok = false; s->ms_build_device = 0.0f; s->ms_build_host = 0.0f;
   s->dev_blob = nullptr; s->d_ctr = nullptr; s->d_stats = nullptr; s->h_stats = nullptr; s->node_factor = 2.0f;
   s->d_tile_cost = s->h_tile_cost = nullptr; s->d_tile_order = s->h_tile_order = nullptr; s->order_cap = 0; s->order_valid = false; s->order_dirty = true;
   s->d_lists = nullptr; s->list_cap = first_list_cap(); s->list_waves = 0; s->d_spill = nullptr; s->d_frame = nullptr; s->frame_cap = 0;
   s->camera = camera; s->has_camera = has_camera;
   scene_tuning(s);
   { char *unused = nullptr; (void)rt_scratch_address(device, &unused); }   /* this device's static working memory (rt_trace_rays_static), looked up while the device is current; a failure is reported by that call */
   *out = s;
   return RT_OK;
}

static int scene_upload(rt_scene *s, int flags);

extern "C" int rt_scene_create(const rt_scene_desc *desc, int device, int flags, rt_scene **out)
{
   if (!desc || !out) { rt_set_error("rt_scene_create: null argument"); return RT_ERR_BAD_ARGUMENT; }
   rt_scene *s = nullptr;
   { int rc = scene_begin(device, flags, desc->camera, desc->has_camera, "rt_scene_create", &s); if (rc != RT_OK) { *out = nullptr; return rc; } }
   *out = nullptr;

   if (!rt_env_int("RT_MISS_FILTER", 1, true)) flags |= RT_SCENE_NO_FILTER;      /* for A/B runs; frames are the same either way */
   auto t0 = std::chrono::steady_clock::now();
   /* (the second tree's thread runs on while the rest of the scene is prepared and copied: scene_upload waits for it) */
   int st = rt_flatten_scene_ex(desc, flags | RT_FLATTEN_DEFER_FILTER, &s->host);
   if (st != RT_OK) { delete s; return st; }
   if ((flags & RT_SCENE_GPU_BVH) && s->host.n_prims > 1) {
      rt_bvh_node *nodes = nullptr; int32_t *order = nullptr; int depth = 0; float ms = 0.0f;
      st = rt_lbvh_build(s->host.prim_boxes, s->host.n_prims, &nodes, &order, &depth, &ms);
      if (st != RT_OK) { rt_flat_scene_free(&s->host); delete s; return st; }
      free(s->host.nodes); free(s->host.leaf_order);
      s->host.nodes = nodes; s->host.leaf_order = order;
      s->host.n_nodes = s->host.n_prims - 1; s->host.tree_depth = depth; s->host.root = 0;
      s->ms_build_device = ms;
      for (int k = 0; k < s->host.n_nodes; k++)
         for (int j = 0; j < 12; j++)
            if (!(std::fabs(s->host.nodes[k].lmin[j]) <= 1.099511627776e12f)) s->host.bounds_in_range = 0;
   }
   s->host_owner = std::make_shared<HostScene>(); s->host_owner->flat = s->host;      /* from here on the arrays are the owner's */
   s->ms_build_host = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
   st = scene_upload(s, flags);
   if (st != RT_OK) return st;      /* (scene_upload has destroyed the scene) */
   *out = s;
   return RT_OK;
}

/* A replica of a scene on another device (or the same one): the flattened scene and its trees, built once on the host,
 * are shared; what is made anew is everything on the device. */
extern "C" int rt_scene_clone(const rt_scene *src, int device, rt_scene **out)
{
   if (!src || !out) { rt_set_error("rt_scene_clone: null argument"); return RT_ERR_BAD_ARGUMENT; }
   *out = nullptr;
   if (src->device < 0 || !src->host_owner) { rt_set_error("rt_scene_clone: the scene has no device side to copy (made with rt_scene_create_host)"); return RT_ERR_BAD_ARGUMENT; }
   rt_scene *s = nullptr;
   { int rc = scene_begin(device, src->flags, src->camera, src->has_camera, "rt_scene_clone", &s); if (rc != RT_OK) return rc; }
   s->host = src->host; s->host_owner = src->host_owner;
   s->ms_build_device = 0.0f;
   int flags = src->flags;
   if (!rt_env_int("RT_MISS_FILTER", 1, true)) flags |= RT_SCENE_NO_FILTER;
   int st = scene_upload(s, flags);
   if (st != RT_OK) return st;
   *out = s;
   return RT_OK;
}

/* the device side of a scene whose host side is complete; on failure the scene is destroyed */
static int scene_upload(rt_scene *s, int flags)
{
   auto t1 = std::chrono::steady_clock::now();
   hipError_t e_dev = hipSetDevice(s->device);
   if (e_dev != hipSuccess) { rt_set_error("hipSetDevice(%d) failed: %s", s->device, hipGetErrorString(e_dev)); rt_scene_destroy(s); return RT_ERR_HIP; }
   const rt_flat_scene &h = s->host;
   /* On the device the primitives are numbered by their place among the tree's leaves (formBVH's sorted order), not
    * by the caller's order: a leaf reference is then its position, which a tree with computed child slots needs, and
    * neighbouring leaves have neighbouring records.  prim_ids maps back wherever an id leaves the device. */
   const size_t np_ = (size_t)(h.n_prims > 0 ? h.n_prims : 0);
   std::vector<int32_t> pos_of(np_, -1), ids(np_);
   bool perm_ok = h.leaf_order != nullptr;
   for (size_t k = 0; k < np_ && perm_ok; k++) {
      int32_t pr = h.leaf_order[k];
      if (pr < 0 || (size_t)pr >= np_ || pos_of[(size_t)pr] >= 0) perm_ok = false; else pos_of[(size_t)pr] = (int32_t)k;
   }
   for (size_t k = 0; k < np_; k++) { if (!perm_ok) pos_of[k] = (int32_t)k; ids[k] = perm_ok ? h.leaf_order[k] : (int32_t)k; }
   std::vector<rt_dev_prim> d_prims(np_); std::vector<rt_dev_xform> d_xforms(np_); std::vector<rt_dev_shade> d_shades(np_);
   for (size_t k = 0; k < np_; k++) { d_prims[k] = h.prims[ids[k]]; d_xforms[k] = h.xforms[ids[k]]; d_shades[k] = h.shades[ids[k]]; }
   /* one node array: the tree the rays are answered from, then (if the scene has one) the miss filter's tree, its
    * internal child references moved behind the first tree's records */
   /* (a scene fresh from rt_scene_create may still be building its second tree on a thread of its own: that tree has one
    * node less than the scene has primitives -- or does not come at all -- and its records are filled in last, below,
    * after everything else has been prepared and copied) */
   const bool filter_pending = h.pending_filter != nullptr;
   const size_t n_tree = (size_t)(h.n_nodes > 0 ? h.n_nodes : 0);
   size_t n_filter = filter_pending ? (np_ > 1 ? np_ - 1 : 0) : (size_t)(h.n_filter_nodes > 0 ? h.n_filter_nodes : 0);
   std::vector<rt_bvh_node> d_nodes(n_tree + n_filter);
   auto place_nodes = [&](size_t first, size_t last) {
      for (size_t k = first; k < last; k++) {
         const bool in_filter = k >= n_tree;
         d_nodes[k] = in_filter ? s->host.filter_nodes[k - n_tree] : s->host.nodes[k];
         int32_t *child[2] = {&d_nodes[k].left, &d_nodes[k].right};
         for (int c = 0; c < 2; c++) {
            if (*child[c] < 0) *child[c] = ~pos_of[(size_t)~*child[c]];
            else if (in_filter) *child[c] += (int32_t)n_tree;
         }
      }
   };
   place_nodes(0, filter_pending ? n_tree : d_nodes.size());
   s->prim_ids_host = ids;
   /* where each leaf hangs in the reference's tree, for walking the path from the root down to one leaf */
   std::vector<int32_t> leaf_parent(n_filter > 0 ? np_ : 0, 0);
   if (n_filter > 0)
      for (size_t k = 0; k < n_tree; k++) {
         if (d_nodes[k].left < 0) leaf_parent[(size_t)~d_nodes[k].left] = (int32_t)k;
         if (d_nodes[k].right < 0) leaf_parent[(size_t)~d_nodes[k].right] = (int32_t)k;
      }
   /* Each node of the reference's tree is given the range of leaves below it, in the two spare words of its record.  The
    * primitives are numbered in the order of that tree's leaves and the nodes parent, left subtree, right subtree, every
    * node with two children: a subtree with m nodes holds m + 1 consecutive leaves.  A guided walk (k_trace) relies on
    * exactly that and is only allowed when it has been checked here. */
   s->guide_ok = false;
   if (n_filter > 0 && n_tree > 0 && h.root >= 0) {
      bool ok = true;
      auto lo_of = [&](int32_t c) { return c < 0 ? ~c : d_nodes[(size_t)c].pad[0]; };
      auto hi_of = [&](int32_t c) { return c < 0 ? ~c + 1 : d_nodes[(size_t)c].pad[1]; };
      for (size_t k = n_tree; k-- > 0 && ok;) {
         rt_bvh_node &n = d_nodes[k];
         if ((n.left >= 0 && (size_t)n.left <= k) || (n.right >= 0 && (size_t)n.right <= k) || (n.left >= 0 && (size_t)n.left != k + 1)) { ok = false; break; }
         n.pad[0] = lo_of(n.left); n.pad[1] = hi_of(n.right);
         ok = hi_of(n.left) == lo_of(n.right) && n.pad[0] < n.pad[1];
         /* the left subtree's nodes are k+1 .. right-1 when the right child is a node: one leaf more than nodes */
         if (ok && n.left >= 0 && n.right >= 0) ok = hi_of(n.left) - lo_of(n.left) == (n.right - n.left) + 1;
      }
      if (ok) ok = d_nodes[(size_t)h.root].pad[0] == 0 && d_nodes[(size_t)h.root].pad[1] == (int32_t)np_ && h.root == 0;
      if (!ok) for (size_t k = 0; k < n_tree; k++) d_nodes[k].pad[0] = d_nodes[k].pad[1] = 0;
      else {
         const rt_bvh_node &r = d_nodes[(size_t)h.root];
         for (int j = 0; j < 3; j++) { s->tree_lo[j] = std::min(r.lmin[j], r.rmin[j]); s->tree_hi[j] = std::max(r.lmax[j], r.rmax[j]); }
         for (int j = 0; j < 3; j++) if (!(s->tree_lo[j] <= s->tree_hi[j])) ok = false;
      }
      s->guide_ok = ok;
   }
   enum { N_ARR = 11 };
   size_t off[N_ARR + 1], sizes[N_ARR] = {
      sizeof(rt_bvh_node) * d_nodes.size(), sizeof(rt_dev_prim) * (size_t)h.n_prims,
      sizeof(rt_dev_xform) * (size_t)h.n_prims, sizeof(rt_dev_shade) * (size_t)h.n_prims,
      sizeof(rt_dev_plane) * (size_t)h.n_planes, sizeof(rt_material) * (size_t)h.n_materials,
      sizeof(rt_dev_light) * (size_t)h.n_lights, sizeof(rt_dev_texinfo) * (size_t)h.n_textures, 4 * h.n_texels,
      sizeof(int32_t) * np_, sizeof(int32_t) * leaf_parent.size()};
   off[0] = 0;
   for (int k = 0; k < N_ARR; k++) off[k + 1] = off[k] + align_up(sizes[k] ? sizes[k] : 64, 256);
   s->dev_bytes = off[N_ARR];
   hipError_t e = hipMalloc(&s->dev_blob, s->dev_bytes);
   if (e != hipSuccess) { rt_set_error("hipMalloc(%zu) failed: %s", s->dev_bytes, hipGetErrorString(e)); rt_scene_destroy(s); return RT_ERR_HIP; }
   const void *src[N_ARR] = {d_nodes.data(), d_prims.data(), d_xforms.data(), d_shades.data(), h.planes, h.materials, h.lights, h.tex_info,
                             h.texels, ids.data(), leaf_parent.data()};
   {
      /* one copy for all of it: a scene of a few megabytes spends more on the eleven calls than on the bytes (a scene
       * of hundreds of megabytes is not staged a second time on the host: its arrays go one by one) */
      const bool staged = s->dev_bytes <= ((size_t)64 << 20);
      std::vector<char> stage(staged ? s->dev_bytes : 0);
      /* (of the node array only the first tree's records, while the second tree is still being built) */
      const size_t nodes_now = filter_pending ? sizeof(rt_bvh_node) * n_tree : sizes[0];
      for (int k = 0; k < N_ARR; k++) {
         const size_t len = k == 0 ? nodes_now : sizes[k];
         if (!len) continue;
         if (staged) { memcpy(stage.data() + off[k], src[k], len); continue; }
         e = hipMemcpy((char *)s->dev_blob + off[k], src[k], len, hipMemcpyHostToDevice);
         if (e != hipSuccess) { rt_set_error("scene upload failed: %s", hipGetErrorString(e)); rt_scene_destroy(s); return RT_ERR_HIP; }
      }
      if (staged) {
         if (filter_pending) {
            e = nodes_now ? hipMemcpy(s->dev_blob, stage.data(), nodes_now, hipMemcpyHostToDevice) : hipSuccess;
            if (e == hipSuccess && s->dev_bytes > off[1]) e = hipMemcpy((char *)s->dev_blob + off[1], stage.data() + off[1], s->dev_bytes - off[1], hipMemcpyHostToDevice);
         } else e = hipMemcpy(s->dev_blob, stage.data(), s->dev_bytes, hipMemcpyHostToDevice);
         if (e != hipSuccess) { rt_set_error("scene upload failed: %s", hipGetErrorString(e)); rt_scene_destroy(s); return RT_ERR_HIP; }
      }
   }
   char *b = (char *)s->dev_blob;
   s->d.nodes = (const rt_bvh_node *)(b + off[0]); s->d.prims = (const rt_dev_prim *)(b + off[1]);
   s->d.xforms = (const rt_dev_xform *)(b + off[2]); s->d.shades = (const rt_dev_shade *)(b + off[3]);
   s->d.planes = (const rt_dev_plane *)(b + off[4]); s->d.materials = (const rt_material *)(b + off[5]);
   s->d.lights = (const rt_dev_light *)(b + off[6]);
   s->d.tex_info = (const rt_dev_texinfo *)(b + off[7]); s->d.texels = (const uchar4 *)(b + off[8]);
   s->d.prim_ids = (const int *)(b + off[9]); s->d.leaf_parent = (const int *)(b + off[10]);
   s->d.n_prims = h.n_prims; s->d.n_planes = h.n_planes; s->d.n_lights = h.n_lights;
   s->d.root = (h.root < 0 && h.root != RT_ROOT_EMPTY) ? ~pos_of[(size_t)~h.root] : h.root;
   s->d.fast_ok = h.bounds_in_range;
   s->d.filter_root = (n_filter > 0 && h.root >= 0) ? (int)n_tree : RT_ROOT_EMPTY;
   s->d.filter_kinds = 3;
   {
      /* the first guess, for the frames before the try-out has spoken (a cold launch_kernel call renders only one):
       * a scene made mostly of glass does better without the second tree */
      long long glass = 0;
      for (int k = 0; k < h.n_prims; k++) if (h.materials[h.prims[k].material].refr > 0.0f) glass++;
      if (flags & RT_SCENE_FILTER_ALWAYS) { s->filter_auto = 0; s->filter_choice = 3; }
      if (s->filter_auto && h.n_prims > 0 && glass * 4 > (long long)h.n_prims) s->filter_choice = 0;
   }
   s->d.any_xform = h.any_xform;
   {
      /* every box of the reference's tree inside its parent's?  (formBVH makes them so: combineBoundingBox,
       * BoundingBox.h:48-52, takes the smaller and the larger bound and pads; a scene with a NaN bound fails here) */
      bool nested = n_tree > 0 && n_filter > 0;
      for (size_t k = 0; k < n_tree && nested; k++) {
         const rt_bvh_node &n = d_nodes[k];
         const int32_t child[2] = {n.left, n.right};
         const float *box[2] = {n.lmin, n.rmin};       /* lmin, lmax / rmin, rmax: six consecutive floats each */
         for (int c = 0; c < 2 && nested; c++) {
            if (child[c] < 0) continue;
            const rt_bvh_node &m = d_nodes[(size_t)child[c]];
            const float *sub[2] = {m.lmin, m.rmin};
            for (int b = 0; b < 2; b++)
               for (int j = 0; j < 3; j++)
                  if (!(sub[b][j] >= box[c][j]) || !(sub[b][3 + j] <= box[c][3 + j])) nested = false;
         }
      }
      s->d.verify_jump = nested && rt_env_int("RT_VERIFY_JUMP", 1, true) ? 1 : 0;
      s->guide_ok = s->guide_ok && nested && rt_env_int("RT_GUIDE", 1, true) != 0;
   }
   s->d.sphere_coeff = h.sphere_coeff; s->d.origin_reach = h.origin_reach;
   /* (for measurements: what the two margins cost) */
   if (const char *e = getenv("RT_SPHERE_MARGIN")) s->d.sphere_coeff *= (float)atof(e);
   /* a bounce ray needs a material with reflection or refraction above 0 (kernel.cu:360-374) */
   s->can_bounce = false;
   for (int k = 0; k < h.n_materials; k++) if (h.materials[k].refl > 0.0f || h.materials[k].refr > 0.0f) s->can_bounce = true;
   s->lights_finite = true;
   for (int k = 0; k < h.n_lights; k++) for (int j = 0; j < 3; j++) if (!std::isfinite(h.lights[k].clr[j])) s->lights_finite = false;
   scene_bounds(s);
   if (h.sphere_coeff > 0.0f) {
      /* the box that holds the spheres (other primitives do not count: a floor of two large triangles would put every
       * ray far from "the scene") */
      double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, r2 = 0.0;
      bool ok = s->bounds_ok && h.prim_boxes != nullptr;
      for (int k = 0; k < h.n_prims && ok; k++) {
         if (h.prims[k].type != RT_PRIM_SPHERE) continue;
         for (int j = 0; j < 3; j++) {
            lo[j] = std::min(lo[j], (double)h.prim_boxes[6 * (size_t)k + j]); hi[j] = std::max(hi[j], (double)h.prim_boxes[6 * (size_t)k + 3 + j]);
         }
      }
      for (int j = 0; j < 3 && ok; j++) { ok = lo[j] <= hi[j]; const double half = 0.5 * (hi[j] - lo[j]); r2 += half * half; }
      if (ok) {
         s->d.mid_x = (float)(0.5 * (lo[0] + hi[0])); s->d.mid_y = (float)(0.5 * (lo[1] + hi[1])); s->d.mid_z = (float)(0.5 * (lo[2] + hi[2]));
         s->d.mid_rad = (float)std::sqrt(r2);
      } else s->d.filter_root = RT_ROOT_EMPTY;      /* no usable bounds to measure a ray's distance from the spheres by: such a scene goes without the second tree */
   }

   if (hipMalloc((void **)&s->d_ctr, RT_CTR_WORDS * sizeof(unsigned long long)) != hipSuccess ||
       hipMalloc((void **)&s->d_stats, ST_COUNT * RT_STAT_REPL * sizeof(unsigned long long)) != hipSuccess ||
       hipHostMalloc((void **)&s->h_stats, RT_STATS_BYTES_PADDED + 3 * sizeof(unsigned) * (size_t)RT_PINNED_TILES, hipHostMallocDefault) != hipSuccess) {
      rt_set_error("allocation of queue counters failed"); rt_scene_destroy(s); return RT_ERR_HIP;
   }
   if (rt_stack_depth(s) > RT_LDS_STACK) {
      const size_t bytes = sizeof(StackEntry) * (size_t)(rt_stack_depth(s) - RT_LDS_STACK) * (size_t)s->trace_blocks * RT_BLOCK;
      if (hipMalloc((void **)&s->d_spill, bytes) != hipSuccess) { rt_set_error("allocation of %zu bytes of traversal stack failed", bytes); rt_scene_destroy(s); return RT_ERR_HIP; }
   }
   if (s->own_stream) {
      /* (only a scene that runs its frames on a stream of its own needs one: the first streams a process creates cost
       * milliseconds each, more than a small scene's whole first frame) */
      bool ok = hipStreamCreateWithFlags(&s->chain, hipStreamNonBlocking) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&s->ev_begin, hipEventDisableTiming) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&s->ev_end, hipEventDisableTiming) == hipSuccess;
      if (!ok) { rt_set_error("stream / event creation failed"); rt_scene_destroy(s); return RT_ERR_HIP; }
      s->streams_ok = true;
   }
   if (filter_pending) {
      /* the second tree, now: its records behind the first tree's -- or, if it did not come, a scene without one */
      (void)rt_flat_scene_finish(&s->host);
      if (s->host_owner) s->host_owner->flat = s->host;
      const size_t got = (size_t)(s->host.n_filter_nodes > 0 ? s->host.n_filter_nodes : 0);
      if (got == n_filter && got > 0) {
         place_nodes(n_tree, n_tree + n_filter);
         e = hipMemcpy((char *)s->dev_blob + sizeof(rt_bvh_node) * n_tree, d_nodes.data() + n_tree, sizeof(rt_bvh_node) * n_filter, hipMemcpyHostToDevice);
         if (e != hipSuccess) { rt_set_error("scene upload failed: %s", hipGetErrorString(e)); rt_scene_destroy(s); return RT_ERR_HIP; }
      } else {
         n_filter = 0;
         s->d.filter_root = RT_ROOT_EMPTY; s->d.verify_jump = 0; s->guide_ok = false;
      }
   }
   e = hipDeviceSynchronize();
   if (e != hipSuccess) { rt_set_error("scene upload failed: %s", hipGetErrorString(e)); rt_scene_destroy(s); return RT_ERR_HIP; }
   auto t2 = std::chrono::steady_clock::now();
   s->ms_upload = std::chrono::duration<float, std::milli>(t2 - t1).count();
   return RT_OK;
}

extern "C" int rt_scene_get_info(const rt_scene *s, rt_scene_info *info)
{
   if (!s || !info) { rt_set_error("rt_scene_get_info: null argument"); return RT_ERR_BAD_ARGUMENT; }
   info->n_bvh_prims = s->host.n_prims; info->n_planes = s->host.n_planes; info->n_lights = s->host.n_lights;
   info->n_nodes = s->host.n_nodes; info->n_materials = s->host.n_materials; info->tree_depth = s->host.tree_depth;
   info->root = s->host.root; info->device_bytes = s->dev_bytes;
   info->ms_build_host = s->ms_build_host; info->ms_upload = s->ms_upload; info->ms_build_device = s->ms_build_device;
   info->filter_depth = s->host.n_filter_nodes > 0 ? s->host.filter_depth : 0;
   return RT_OK;
}

extern "C" int rt_scene_get_bvh(const rt_scene *s, const rt_bvh_node **nodes, const int32_t **leaf_order)
{
   if (!s) { rt_set_error("rt_scene_get_bvh: null scene"); return RT_ERR_BAD_ARGUMENT; }
   if (nodes) *nodes = s->host.nodes;
   if (leaf_order) *leaf_order = s->host.leaf_order;
   return RT_OK;
}

extern "C" int rt_scene_set_camera(rt_scene *s, const rt_camera *cam)
{
   if (!s || !cam) { rt_set_error("rt_scene_set_camera: null argument"); return RT_ERR_BAD_ARGUMENT; }
   s->camera = *cam; s->has_camera = 1;
   return RT_OK;
}

/* entries the deepest walk can defer: one per internal level of the deeper of the scene's trees */
int rt_stack_depth(const rt_scene *s)
{
   int depth = s->host.tree_depth > s->host.filter_depth ? s->host.tree_depth : s->host.filter_depth;
   return depth > 0 ? depth : 1;
}

int rt_env_int(const char *name, int fallback, bool allow_zero)
{
   const char *e = getenv(name);
   if (!e) return fallback;
   int v = atoi(e);
   return (v > 0 || (allow_zero && v == 0)) ? v : fallback;
}

/* The persistent traversal launch fills the device once: 4 workgroups of 256 per CU are what 128 registers per
 * lane allow, and every wave pulls rays until the queue is dry. */
static void scene_tuning(rt_scene *s)
{
   int cus = 256;
   hipDeviceProp_t prop;
   if (hipGetDeviceProperties(&prop, s->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
   s->trace_blocks = rt_env_int("RT_TRACE_BLOCKS", cus * 4, false);
   s->refill_idle = rt_env_int("RT_REFILL_IDLE", RT_REFILL_IDLE, false);
   s->node_burst = rt_env_int("RT_NODE_BURST", RT_NODE_BURST, false);
   s->leaf_trigger = rt_env_int("RT_LEAF_TRIGGER", RT_LEAF_TRIGGER, false);
   s->use_tile_order = rt_env_int("RT_TILE_ORDER", 1, true);
   s->order_classes = rt_env_int("RT_ORDER_CLASSES", RT_ORDER_CLASSES, false);
   s->order_mean_weight = (double)rt_env_int("RT_ORDER_MEAN_WEIGHT", RT_ORDER_MEAN_WEIGHT, true);
   s->fuse_resolve = rt_env_int("RT_FUSE_RESOLVE", 1, true);
   s->cull_enabled = rt_env_int("RT_CULL_TILES", 1, true);
   s->cull_fine = rt_env_int("RT_CULL_FINE", 256, true);
   s->skip_unlit = rt_env_int("RT_SKIP_UNLIT", 16, true);
   s->own_stream = rt_env_int("RT_OWN_STREAM", 0, true);
   {
      const char *e = getenv("RT_MISS_FILTER");
      s->filter_auto = !(e && *e && atoi(e) == 1 && e[0] == '1');      /* "1": always; unset / "auto": try out ("0" builds no second tree at all) */
      s->filter_choice = 3; s->probe_step = 0;
      memset(s->probe_key, 0, sizeof s->probe_key);
   }
   int r = rt_env_int("RT_RESERVE", RT_RESERVE, false);
   s->reserve = (unsigned)(r < 128 ? 128 : r);      /* one spawn stage can ask for 128 entries */
}

