/*
 * rt_sequence.cpp -- a frame sequence with several frames in flight on one device.
 *
 * The reference renders one frame per process (raytrace.cpp:18-56).  A caller that renders many -- a camera path, a
 * turntable -- loses the end of every frame: the traversal launch ends on the few waves that hold its longest chains
 * of dependent rays, then come the resolve pass, the copies of the statistics and the host's wake-up, and only then
 * does the next frame start.  Frames are independent, so the next one can fill that time.  A scene object owns its
 * working buffers (ray queue, counters, tile order), which is why a sequence keeps one replica of the scene per frame
 * in flight (2-3 MB each for the reference's scenes), with a stream and a host thread of its own; frames are dealt to
 * the replicas in turn, and the waves of frame k + 1 move onto the compute units as those of frame k leave them.
 * Every frame is a whole frame of rt_render_device: same bytes, same statistics.
 */
#include <hip/hip_runtime_api.h>

#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rt_common.h"

namespace {
enum { SLOT_FREE = 0, SLOT_QUEUED, SLOT_DONE };
struct Slot {
   rt_scene *scene = nullptr;
   hipStream_t stream = nullptr;
   std::thread worker;
   int state = SLOT_FREE;
   rt_render_params params;
   rt_camera camera;
   bool has_camera = false;
   void *out = nullptr;
   rt_stats stats;
   int status = RT_OK;
   std::string error;
};
}  // namespace

struct rt_sequence {
   int device = -1;
   std::vector<Slot> slots;
   std::mutex mu;
   std::condition_variable cv_work, cv_done;
   bool quit = false;
   unsigned long long submitted = 0, waited = 0;      /* frame k lives in slot k mod slots.size() */
   /* A replica times its first frames of a frame size to choose how it uses the second tree (frames 2-5) and to order
    * its tiles; a frame of another replica running beside them would spoil those timings.  So the first frames of a
    * size are rendered one at a time: this many more submissions wait until nothing else is running. */
   int serial_left = 0;
   long long size_key[5] = {0, 0, 0, 0, 0};
   /* the camera a frame submitted without one is rendered with: the last one submitted, whichever replica got that frame */
   rt_camera camera; bool has_camera = false;
};

namespace {
void worker_loop(rt_sequence *q, int k)
{
   Slot &s = q->slots[(size_t)k];
   (void)hipSetDevice(q->device);
   for (;;) {
      {
         std::unique_lock<std::mutex> lock(q->mu);
         q->cv_work.wait(lock, [&] { return q->quit || s.state == SLOT_QUEUED; });
         if (q->quit) return;
      }
      /* (the slot is this thread's until it says DONE: the submitting thread does not touch a queued slot) */
      int rc = RT_OK;
      if (s.has_camera) rc = rt_scene_set_camera(s.scene, &s.camera);
      memset(&s.stats, 0, sizeof s.stats);
      if (rc == RT_OK) rc = rt_render_device(s.scene, &s.params, s.out, s.stream, &s.stats);
      s.status = rc;
      s.error = rc == RT_OK ? "" : rt_last_error();
      {
         std::lock_guard<std::mutex> lock(q->mu);
         s.state = SLOT_DONE;
      }
      q->cv_done.notify_all();
   }
}
}  // namespace

extern "C" void rt_sequence_destroy(rt_sequence *q)
{
   if (!q) return;
   {
      std::lock_guard<std::mutex> lock(q->mu);
      q->quit = true;
   }
   q->cv_work.notify_all();
   for (Slot &s : q->slots) if (s.worker.joinable()) s.worker.join();
   if (q->device >= 0) (void)hipSetDevice(q->device);
   for (Slot &s : q->slots) {
      if (s.stream) (void)hipStreamDestroy(s.stream);
      if (s.scene) rt_scene_destroy(s.scene);
   }
   delete q;
}

extern "C" int rt_sequence_create(const rt_scene_desc *desc, int device, int flags, int frames_in_flight, rt_sequence **out)
{
   if (!desc || !out) { rt_set_error("rt_sequence_create: null argument"); return RT_ERR_BAD_ARGUMENT; }
   *out = nullptr;
   if (frames_in_flight < 1 || frames_in_flight > 8) { rt_set_error("rt_sequence_create: 1 to 8 frames in flight, not %d", frames_in_flight); return RT_ERR_BAD_ARGUMENT; }
   rt_sequence *q = new (std::nothrow) rt_sequence();
   if (!q) { rt_set_error("rt_sequence_create: out of memory"); return RT_ERR_BAD_ARGUMENT; }
   q->device = device;
   q->slots.resize((size_t)frames_in_flight);
   for (Slot &s : q->slots) {
      /* (flattened and built once; the other replicas share the first one's host side) */
      int rc = q->slots[0].scene && &s != &q->slots[0] ? rt_scene_clone(q->slots[0].scene, device, &s.scene) : rt_scene_create(desc, device, flags, &s.scene);
      if (rc != RT_OK) { std::string why = rt_last_error(); rt_sequence_destroy(q); rt_set_error("%s", why.c_str()); return rc; }
   }
   /* The streams are made one after the other, after everything else: the runtime deals its few hardware queues out to
    * streams in the order they are created, and two replicas whose streams shared a queue would take turns instead of
    * overlapping. */
   for (Slot &s : q->slots) {
      if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) {
         rt_sequence_destroy(q);
         rt_set_error("rt_sequence_create: stream creation failed");
         return RT_ERR_HIP;
      }
   }
   try {
      for (size_t k = 0; k < q->slots.size(); k++) q->slots[k].worker = std::thread(worker_loop, q, (int)k);
   } catch (...) {
      rt_sequence_destroy(q);
      rt_set_error("rt_sequence_create: could not start the host threads");
      return RT_ERR_BAD_ARGUMENT;
   }
   *out = q;
   return RT_OK;
}

extern "C" int rt_sequence_frames_in_flight(const rt_sequence *q) { return q ? (int)q->slots.size() : 0; }

extern "C" int rt_sequence_submit(rt_sequence *q, const rt_render_params *params, const rt_camera *camera, void *out_rgba_dev)
{
   if (!q || !params || !out_rgba_dev) { rt_set_error("rt_sequence_submit: null argument"); return RT_ERR_BAD_ARGUMENT; }
   {
      std::unique_lock<std::mutex> lock(q->mu);
      if (q->submitted - q->waited >= q->slots.size()) {
         rt_set_error("rt_sequence_submit: %d frames are in flight already: wait for one first", (int)q->slots.size());
         return RT_ERR_BAD_ARGUMENT;
      }
      const long long key[5] = {params->width, params->height, params->sample_count, params->tile_first, params->tile_stride};
      if (memcmp(key, q->size_key, sizeof key) != 0) { memcpy(q->size_key, key, sizeof key); q->serial_left = 6 * (int)q->slots.size(); }
      if (q->serial_left > 0) {
         q->serial_left--;
         q->cv_done.wait(lock, [&] { for (const Slot &o : q->slots) if (o.state == SLOT_QUEUED) return false; return true; });
      }
      Slot &s = q->slots[(size_t)(q->submitted % q->slots.size())];
      s.params = *params;
      if (camera) { q->camera = *camera; q->has_camera = true; }
      s.has_camera = q->has_camera;
      if (q->has_camera) s.camera = q->camera;
      s.out = out_rgba_dev;
      s.state = SLOT_QUEUED;
      q->submitted++;
   }
   q->cv_work.notify_all();
   return RT_OK;
}

extern "C" int rt_sequence_wait(rt_sequence *q, rt_stats *stats)
{
   if (!q) { rt_set_error("rt_sequence_wait: null argument"); return RT_ERR_BAD_ARGUMENT; }
   std::unique_lock<std::mutex> lock(q->mu);
   if (q->submitted == q->waited) { rt_set_error("rt_sequence_wait: no frame is in flight"); return RT_ERR_BAD_ARGUMENT; }
   Slot &s = q->slots[(size_t)(q->waited % q->slots.size())];
   q->cv_done.wait(lock, [&] { return s.state == SLOT_DONE; });
   if (stats) *stats = s.stats;
   const int rc = s.status;
   if (rc != RT_OK) rt_set_error("%s", s.error.c_str());
   s.state = SLOT_FREE;
   q->waited++;
   return rc;
}
