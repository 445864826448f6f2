/*
 * rt_multi.cpp -- the multi-GPU frame driver: one process, N devices, one frame.
 *
 * The reference renders on one device (kernel.cu:689-761).  Its samples are independent (kernel.cu:484-543 is one
 * thread per sample, no communication) and read the scene only, so a frame shards by image tiles: the scene and its
 * BVH are replicated on every device (built once on the host, uploaded N times), tile k of the 32x32 tile grid goes
 * to part k mod N, every part renders its tiles into a compact tile buffer on its own device, and the one exchange
 * is the gather of those buffers to the first device: N - 1 direct peer copies over xGMI, each on its source
 * device's stream (the devices of a node are fully connected, so the copies run side by side; no ring).  The first
 * device then un-tiles the parts into the row-major frame and one copy brings it to the host.  No reduction: tiles
 * are disjoint, so the frame is byte for byte the one-device frame.
 *
 * A part has its own scene object (queues, counters, stream), so parts never share mutable state, and one host
 * thread per part drives it; the threads persist across frames.  The same device may appear more than once in the
 * device list: its parts then take turns on it, which is how the driver is tested on a one-GPU machine.
 *
 * Why the gather is N - 1 peer copies and not an RCCL collective: the parts are 1-4 MB, every part has exactly one
 * destination, and the copies are already what a grouped send / receive would become on a fully connected node --
 * one transfer per link, side by side -- without a communicator to set up per device list or a collective launch
 * per frame (tens of microseconds each, against a frame of about a millisecond).  One process per GPU (bench.py under
 * torch.distributed) has no peer pointers to copy through and gathers over RCCL.
 *
 * Frames in flight: a driver made with rt_multi_create_pipelined keeps that many complete sets of the above (a "core":
 * replicas, streams, part threads, gather buffers) and a thread per core that runs whole frames; rt_multi_submit hands
 * a frame to the next core and returns, rt_multi_wait returns the oldest.  The host's round trip per frame (wake the
 * part threads, wait for them, un-tile, synchronise) is then off the critical path, and the end of one frame's
 * launches overlaps the start of the next, as with rt_sequence on one device.
 */
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rt_common.h"

struct Part {
   int device = -1;
   rt_scene *scene = nullptr;
   void *d_tiles = nullptr;          /* this part's compact tile buffer, on its device */
   size_t tiles_cap = 0;
   hipStream_t stream = nullptr;
   rt_stats stats;
   int status = RT_OK;
   std::string error;
   std::thread worker;
};

struct Core {
   std::vector<Part> parts;
   void *d_parts = nullptr;          /* on parts[0].device: the gathered buffers, back to back */
   void *d_frame = nullptr;          /* on parts[0].device: the row-major frame */
   size_t parts_cap = 0, frame_cap = 0;
   /* one frame at a time: the main thread posts the parameters, every worker renders its part, the main thread waits */
   std::mutex mu;
   std::condition_variable cv_go, cv_done;
   unsigned long long generation = 0;
   int pending = 0;
   bool quit = false;
   rt_render_params params;
   size_t part_bytes = 0;
   /* the frame this core has been handed by rt_multi_submit, and the thread that runs it */
   std::thread runner;
   int job_state = 0;                /* 0 free, 1 queued or running, 2 done */
   rt_render_params job_params;
   rt_camera job_camera; bool job_has_camera = false;
   void *job_out = nullptr;
   rt_stats job_stats;
   int job_status = RT_OK;
   std::string job_error;
};

struct rt_multi {
   std::vector<Core *> cores;
   std::mutex mu;                    /* guards the cores' job_state */
   std::condition_variable cv_job, cv_job_done;
   bool quit = false;
   unsigned long long submitted = 0, waited = 0;      /* frame k runs on core k mod cores.size() */
   rt_camera camera; bool has_camera = false;         /* the camera a frame submitted without one is rendered with */
};

namespace {

#define MULTI_TRY(expr, part)                                                                 \
   do {                                                                                       \
      hipError_t e_ = (expr);                                                                 \
      if (e_ != hipSuccess) {                                                                 \
         (part).status = RT_ERR_HIP;                                                          \
         (part).error = std::string(#expr) + " failed: " + hipGetErrorString(e_);             \
         return;                                                                              \
      }                                                                                       \
   } while (0)

size_t part_tile_bytes(int width, int height, int n_parts)
{
   long long tiles = (long long)((width + RT_TILE - 1) / RT_TILE) * ((height + RT_TILE - 1) / RT_TILE);
   return (size_t)((tiles + n_parts - 1) / n_parts) * RT_TILE * RT_TILE * 4;
}

/* one frame of part k: render its tiles, send them to the first device */
void render_part(Core *m, int k)
{
   Part &p = m->parts[(size_t)k];
   p.status = RT_OK; p.error.clear();
   memset(&p.stats, 0, sizeof p.stats);
   MULTI_TRY(hipSetDevice(p.device), p);
   if (m->part_bytes > p.tiles_cap) {
      if (p.d_tiles) (void)hipFree(p.d_tiles);
      p.d_tiles = nullptr; p.tiles_cap = 0;
      MULTI_TRY(hipMalloc(&p.d_tiles, m->part_bytes), p);
      p.tiles_cap = m->part_bytes;
   }
   rt_render_params rp = m->params;
   rp.tile_first = k; rp.tile_stride = (int)m->parts.size(); rp.out_layout = RT_OUT_TILES;
   int rc = rt_render_device(p.scene, &rp, p.d_tiles, p.stream, &p.stats);
   if (rc != RT_OK) { p.status = rc; p.error = rt_last_error(); return; }
   const int n_tiles = rt_tile_count(rp.width, rp.height, rp.tile_first, rp.tile_stride);
   if (n_tiles > 0) {
      char *dst = (char *)m->d_parts + (size_t)k * m->part_bytes;
      const size_t bytes = (size_t)n_tiles * RT_TILE * RT_TILE * 4;
      if (p.device == m->parts[0].device) MULTI_TRY(hipMemcpyAsync(dst, p.d_tiles, bytes, hipMemcpyDeviceToDevice, p.stream), p);
      else MULTI_TRY(hipMemcpyPeerAsync(dst, m->parts[0].device, p.d_tiles, p.device, bytes, p.stream), p);
   }
   MULTI_TRY(hipStreamSynchronize(p.stream), p);
}

void worker_loop(Core *m, int k)
{
   unsigned long long seen = 0;
   for (;;) {
      {
         std::unique_lock<std::mutex> lock(m->mu);
         m->cv_go.wait(lock, [&] { return m->quit || m->generation != seen; });
         if (m->quit) return;
         seen = m->generation;
      }
      render_part(m, k);
      {
         std::lock_guard<std::mutex> lock(m->mu);
         if (--m->pending == 0) m->cv_done.notify_all();
      }
   }
}

int fail(rt_multi *m, int code, const char *fmt, const char *detail)
{
   rt_set_error(fmt, detail);
   rt_multi_destroy(m);
   return code;
}

void core_destroy(Core *m)
{
   if (!m) return;
   {
      std::lock_guard<std::mutex> lock(m->mu);
      m->quit = true;
   }
   m->cv_go.notify_all();
   for (Part &p : m->parts) if (p.worker.joinable()) p.worker.join();
   for (Part &p : m->parts) {
      if (p.device >= 0) (void)hipSetDevice(p.device);
      if (p.d_tiles) (void)hipFree(p.d_tiles);
      if (p.stream) (void)hipStreamDestroy(p.stream);
      if (p.scene) rt_scene_destroy(p.scene);
   }
   if (!m->parts.empty() && m->parts[0].device >= 0) (void)hipSetDevice(m->parts[0].device);
   if (m->d_parts) (void)hipFree(m->d_parts);
   if (m->d_frame) (void)hipFree(m->d_frame);
   delete m;
}

/* one complete set of replicas, streams, part threads for the device list; returns null with the error text set */
Core *core_create(const rt_scene_desc *desc, const rt_scene *first, const int *devices, int n_devices, int flags, int visible, int *status)
{
   Core *m = new (std::nothrow) Core();
   if (!m) { rt_set_error("rt_multi_create: out of memory"); *status = RT_ERR_BAD_ARGUMENT; return nullptr; }
   auto give_up = [&](int code, const std::string &why) { core_destroy(m); rt_set_error("rt_multi_create: %s", why.c_str()); *status = code; return (Core *)nullptr; };
   m->parts.resize((size_t)n_devices);
   for (int k = 0; k < n_devices; k++) {
      const int dev = devices ? devices[k] : k;
      if (dev < 0 || dev >= visible) {
         char text[96]; snprintf(text, sizeof text, "device %d (of %d visible) out of range", dev, visible);
         return give_up(RT_ERR_BAD_ARGUMENT, text);
      }
      Part &p = m->parts[(size_t)k];
      p.device = dev;
      /* the scene is replicated: one copy of the flattened scene and its trees per part, in that part's device memory;
       * flattened and built once, on the host, by the first part of all */
      const rt_scene *built = first ? first : (k > 0 ? m->parts[0].scene : nullptr);
      int rc = built ? rt_scene_clone(built, dev, &p.scene) : rt_scene_create(desc, dev, flags, &p.scene);
      if (rc != RT_OK) { std::string why = rt_last_error(); return give_up(rc, why); }
      if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) != hipSuccess)
         return give_up(RT_ERR_HIP, "stream creation failed");
   }
   /* direct copies into the first device's memory: let every other device write there */
   for (int k = 1; k < n_devices; k++) {
      const int src = m->parts[(size_t)k].device, dst = m->parts[0].device;
      int can = 0;
      if (src != dst && hipDeviceCanAccessPeer(&can, src, dst) == hipSuccess && can) {
         (void)hipSetDevice(src);
         hipError_t e = hipDeviceEnablePeerAccess(dst, 0);
         if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   /* the copy then goes through the host */
      }
   }
   try {
      for (int k = 0; k < n_devices; k++) m->parts[(size_t)k].worker = std::thread(worker_loop, m, k);
   } catch (...) {
      return give_up(RT_ERR_BAD_ARGUMENT, "could not start the host threads");
   }
   return m;
}

int core_render(Core *m, const rt_render_params *params, uint8_t *out_host, void *out_dev, rt_stats *stats);

/* the thread of a core: whole frames handed over by rt_multi_submit */
void runner_loop(rt_multi *mm, Core *c)
{
   for (;;) {
      {
         std::unique_lock<std::mutex> lock(mm->mu);
         mm->cv_job.wait(lock, [&] { return mm->quit || c->job_state == 1; });
         if (mm->quit) return;
      }
      int rc = RT_OK;
      if (c->job_has_camera) for (Part &p : c->parts) { rc = rt_scene_set_camera(p.scene, &c->job_camera); if (rc != RT_OK) break; }
      memset(&c->job_stats, 0, sizeof c->job_stats);
      if (rc == RT_OK) rc = core_render(c, &c->job_params, nullptr, c->job_out, &c->job_stats);
      c->job_status = rc;
      c->job_error = rc == RT_OK ? "" : rt_last_error();
      {
         std::lock_guard<std::mutex> lock(mm->mu);
         c->job_state = 2;
      }
      mm->cv_job_done.notify_all();
   }
}

}  // namespace

extern "C" void rt_multi_destroy(rt_multi *m)
{
   if (!m) return;
   {
      std::lock_guard<std::mutex> lock(m->mu);
      m->quit = true;
   }
   m->cv_job.notify_all();
   for (Core *c : m->cores) if (c && c->runner.joinable()) c->runner.join();
   for (Core *c : m->cores) core_destroy(c);
   delete m;
}

extern "C" int rt_multi_create_pipelined(const rt_scene_desc *desc, const int *devices, int n_devices, int flags, int frames_in_flight, rt_multi **out)
{
   if (!desc || !out || n_devices <= 0 || n_devices > 64) { rt_set_error("rt_multi_create: bad argument"); return RT_ERR_BAD_ARGUMENT; }
   *out = nullptr;
   if (frames_in_flight < 1 || frames_in_flight > 4) { rt_set_error("rt_multi_create_pipelined: 1 to 4 frames in flight, not %d", frames_in_flight); return RT_ERR_BAD_ARGUMENT; }
   const int visible = rt_device_count();
   if (visible <= 0) { rt_set_error("no HIP device: the render path has no CPU fallback"); return RT_ERR_NO_DEVICE; }
   rt_multi *m = new (std::nothrow) rt_multi();
   if (!m) { rt_set_error("rt_multi_create: out of memory"); return RT_ERR_BAD_ARGUMENT; }
   m->camera = desc->camera; m->has_camera = desc->has_camera != 0;
   for (int f = 0; f < frames_in_flight; f++) {
      int status = RT_OK;
      Core *c = core_create(desc, m->cores.empty() ? nullptr : m->cores[0]->parts[0].scene, devices, n_devices, flags, visible, &status);
      if (!c) { std::string why = rt_last_error(); return fail(m, status, "%s", why.c_str()); }
      m->cores.push_back(c);
   }
   try {
      for (Core *c : m->cores) c->runner = std::thread(runner_loop, m, c);
   } catch (...) {
      return fail(m, RT_ERR_BAD_ARGUMENT, "rt_multi_create: %s", "could not start the host threads");
   }
   *out = m;
   return RT_OK;
}

extern "C" int rt_multi_create(const rt_scene_desc *desc, const int *devices, int n_devices, int flags, rt_multi **out)
{
   return rt_multi_create_pipelined(desc, devices, n_devices, flags, 1, out);
}

extern "C" int rt_multi_device_count(const rt_multi *m) { return m && !m->cores.empty() ? (int)m->cores[0]->parts.size() : 0; }
extern "C" int rt_multi_frames_in_flight(const rt_multi *m) { return m ? (int)m->cores.size() : 0; }

extern "C" int rt_multi_set_camera(rt_multi *m, const rt_camera *camera)
{
   if (!m || !camera) { rt_set_error("rt_multi_set_camera: null argument"); return RT_ERR_BAD_ARGUMENT; }
   {
      std::lock_guard<std::mutex> lock(m->mu);
      if (m->submitted != m->waited) { rt_set_error("rt_multi_set_camera: frames are in flight; pass the camera to rt_multi_submit instead"); return RT_ERR_BAD_ARGUMENT; }
      m->camera = *camera; m->has_camera = true;
   }
   for (Core *c : m->cores) for (Part &p : c->parts) { int rc = rt_scene_set_camera(p.scene, camera); if (rc != RT_OK) return rc; }
   return RT_OK;
}

extern "C" int rt_multi_submit(rt_multi *m, const rt_render_params *params, const rt_camera *camera, void *out_rgba_dev)
{
   if (!m || !params || !out_rgba_dev) { rt_set_error("rt_multi_submit: null argument"); return RT_ERR_BAD_ARGUMENT; }
   {
      std::lock_guard<std::mutex> lock(m->mu);
      if (m->submitted - m->waited >= m->cores.size()) {
         rt_set_error("rt_multi_submit: %d frames are in flight already: wait for one first", (int)m->cores.size());
         return RT_ERR_BAD_ARGUMENT;
      }
      Core *c = m->cores[(size_t)(m->submitted % m->cores.size())];
      /* a frame without a camera of its own is rendered with the driver's current one, whichever core it runs on */
      if (camera) { m->camera = *camera; m->has_camera = true; }
      c->job_params = *params; c->job_out = out_rgba_dev;
      c->job_has_camera = m->has_camera; c->job_camera = m->camera;
      c->job_state = 1;
      m->submitted++;
   }
   m->cv_job.notify_all();
   return RT_OK;
}

extern "C" int rt_multi_wait(rt_multi *m, rt_stats *stats)
{
   if (!m) { rt_set_error("rt_multi_wait: null argument"); return RT_ERR_BAD_ARGUMENT; }
   std::unique_lock<std::mutex> lock(m->mu);
   if (m->submitted == m->waited) { rt_set_error("rt_multi_wait: no frame is in flight"); return RT_ERR_BAD_ARGUMENT; }
   Core *c = m->cores[(size_t)(m->waited % m->cores.size())];
   m->cv_job_done.wait(lock, [&] { return c->job_state == 2; });
   if (stats) *stats = c->job_stats;
   const int rc = c->job_status;
   if (rc != RT_OK) rt_set_error("%s", c->job_error.c_str());
   c->job_state = 0;
   m->waited++;
   return rc;
}

namespace {
int core_render(Core *m, const rt_render_params *params, uint8_t *out_host, void *out_dev, rt_stats *stats)
{
   if (!m || !params || (!out_host && !out_dev)) { rt_set_error("rt_multi_render: null argument"); return RT_ERR_BAD_ARGUMENT; }
   if (params->width <= 0 || params->height <= 0) { rt_set_error("rt_multi_render: non-positive size"); return RT_ERR_BAD_ARGUMENT; }
   {
      /* kernel.cu:707-712, said once for the frame rather than once per part */
      int root = 0;
      while ((root + 1) * (root + 1) <= params->sample_count) root++;
      if (params->sample_count <= 0 || root * root != params->sample_count) {
         rt_set_error("Invalid sample count: %d. Sample count for anti aliasing must have an integer square root", params->sample_count);
         return RT_ERR_BAD_SAMPLE_COUNT;
      }
   }
   const int n = (int)m->parts.size();
   const size_t frame_bytes = (size_t)params->width * (size_t)params->height * 4;
   const size_t part_bytes = part_tile_bytes(params->width, params->height, n);
   Part &first = m->parts[0];
   hipError_t e = hipSetDevice(first.device);
   if (e == hipSuccess && part_bytes * (size_t)n > m->parts_cap) {
      if (m->d_parts) (void)hipFree(m->d_parts);
      m->d_parts = nullptr; m->parts_cap = 0;
      e = hipMalloc(&m->d_parts, part_bytes * (size_t)n);
      if (e == hipSuccess) m->parts_cap = part_bytes * (size_t)n;
   }
   if (e == hipSuccess && !out_dev && frame_bytes > m->frame_cap) {
      if (m->d_frame) (void)hipFree(m->d_frame);
      m->d_frame = nullptr; m->frame_cap = 0;
      e = hipMalloc(&m->d_frame, frame_bytes);
      if (e == hipSuccess) m->frame_cap = frame_bytes;
   }
   if (e != hipSuccess) { rt_set_error("rt_multi_render: %s", hipGetErrorString(e)); return RT_ERR_HIP; }
   {
      std::unique_lock<std::mutex> lock(m->mu);
      m->params = *params; m->part_bytes = part_bytes;
      m->pending = n; m->generation++;
      m->cv_go.notify_all();
      m->cv_done.wait(lock, [&] { return m->pending == 0; });
   }
   rt_stats total; memset(&total, 0, sizeof total);
   for (Part &p : m->parts) {
      if (p.status != RT_OK) { rt_set_error("device %d: %s", p.device, p.error.c_str()); return p.status; }
      const rt_stats &s = p.stats;
      total.rays_primary += s.rays_primary; total.rays_shadow += s.rays_shadow; total.rays_reflect += s.rays_reflect;
      total.rays_refract += s.rays_refract; total.rays_total += s.rays_total; total.shadow_skipped += s.shadow_skipped;
      total.shade_nodes += s.shade_nodes; total.samples += s.samples;
      total.node_visits += s.node_visits; total.prim_tests += s.prim_tests; total.tie_events += s.tie_events;
      total.node_visits_shadow += s.node_visits_shadow; total.prim_tests_shadow += s.prim_tests_shadow;
      total.tie_events_shadow += s.tie_events_shadow;
      total.wave_node_steps += s.wave_node_steps; total.wave_leaf_steps += s.wave_leaf_steps;
      total.hits_certified += s.hits_certified; total.hits_contested += s.hits_contested; total.hits_unreached += s.hits_unreached; total.hits_guided += s.hits_guided; total.wave_steps_both_trees += s.wave_steps_both_trees; total.wave_steps_second_tree_only += s.wave_steps_second_tree_only;
      total.shadows_certified += s.shadows_certified;
      total.primaries_culled += s.primaries_culled;
      total.shadows_unneeded += s.shadows_unneeded;
      /* the parts run side by side: a frame takes as long as its slowest part */
      if (s.ms_total > total.ms_total) total.ms_total = s.ms_total;
      if (s.ms_trace > total.ms_trace) total.ms_trace = s.ms_trace;
      if (s.ms_light > total.ms_light) total.ms_light = s.ms_light;
      if (s.ms_combine > total.ms_combine) total.ms_combine = s.ms_combine;
      if (s.ms_resolve > total.ms_resolve) total.ms_resolve = s.ms_resolve;
      if (s.launches_trace > total.launches_trace) total.launches_trace = s.launches_trace;
      if (s.attempts > total.attempts) total.attempts = s.attempts;
      if (s.max_ray_steps > total.max_ray_steps) total.max_ray_steps = s.max_ray_steps;
      if (s.max_spawn_list > total.max_spawn_list) total.max_spawn_list = s.max_spawn_list;
   }
   /* every part's copy has landed in the first device's memory (each worker waited for its own); un-tile there */
   if (hipSetDevice(first.device) != hipSuccess) { rt_set_error("rt_multi_render: hipSetDevice failed"); return RT_ERR_HIP; }
   void *frame = out_dev ? out_dev : m->d_frame;
   int rc = rt_untile_device(m->d_parts, part_bytes, n, params->width, params->height, frame, first.stream);
   if (rc != RT_OK) return rc;
   if (out_host) e = hipMemcpyAsync(out_host, frame, frame_bytes, hipMemcpyDeviceToHost, first.stream);    /* kernel.cu:759-760 */
   if (e == hipSuccess) e = hipStreamSynchronize(first.stream);
   if (e != hipSuccess) { rt_set_error("rt_multi_render: %s", hipGetErrorString(e)); return RT_ERR_HIP; }
   if (stats) *stats = total;
   return RT_OK;
}
}  // namespace

/* the one-call forms run on the first core, with nothing else in flight */
static int multi_render(rt_multi *m, const rt_render_params *params, uint8_t *out_host, void *out_dev, rt_stats *stats)
{
   if (!m || m->cores.empty()) { rt_set_error("rt_multi_render: null argument"); return RT_ERR_BAD_ARGUMENT; }
   {
      std::lock_guard<std::mutex> lock(m->mu);
      if (m->submitted != m->waited) { rt_set_error("rt_multi_render: frames submitted with rt_multi_submit are still in flight"); return RT_ERR_BAD_ARGUMENT; }
   }
   /* (frames that went through rt_multi_submit may have left the first core with an older camera than the driver's) */
   if (m->has_camera) for (Part &p : m->cores[0]->parts) { int rc = rt_scene_set_camera(p.scene, &m->camera); if (rc != RT_OK) return rc; }
   return core_render(m->cores[0], params, out_host, out_dev, stats);
}

extern "C" int rt_multi_render(rt_multi *m, const rt_render_params *params, uint8_t *out_rgba_host, rt_stats *stats)
{
   return multi_render(m, params, out_rgba_host, nullptr, stats);
}

extern "C" int rt_multi_render_device(rt_multi *m, const rt_render_params *params, void *out_rgba_dev, rt_stats *stats)
{
   return multi_render(m, params, nullptr, out_rgba_dev, stats);
}
