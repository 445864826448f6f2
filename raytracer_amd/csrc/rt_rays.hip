/*
 * rt_rays.hip -- closest hits of arbitrary rays through the same traversal kernel (a parity / debug view of
 * getClosestIntersection, kernel.cu:83-200): rt_trace_rays, and its variant that allocates nothing.
 */
#include <cstdio>
#include <cstring>
#include <mutex>

#include "rt_scene_state.h"

/* ---- arbitrary rays: rt_trace_rays and its variant that allocates nothing ------------------------------------ */

/* queue entries from the caller's ray array (n x {o, d}) and optional toBeat array */
__global__ __launch_bounds__(RT_BLOCK) void k_pack_rays(const float *raw, const float *to_beat, long long n, QRay *q)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= n) return;
   const float *r = raw + 6 * i;
   store_qray(q, (unsigned long long)i, mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), to_beat ? to_beat[i] : -FLT_MAX, 1);
}
/* hits back in the caller's form: primitive ids in the caller's numbering, and t */
__global__ __launch_bounds__(RT_BLOCK) void k_unpack_hits(const Hit *hits, long long n, const int *prim_ids, int n_prims, int *prim, float *t)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= n) return;
   int p = hits[i].prim;
   prim[i] = (p >= 0 && p < n_prims) ? prim_ids[p] : p;
   t[i] = hits[i].t;
}

/* Working memory of one rt_trace_rays call, wherever it lives: the launch's counters and statistics, the caller's rays
 * as they arrive (24 bytes each, plus 4 for toBeat) -- a region that later takes the results (8 bytes each) --, the
 * queue entries and the hits. */
struct RayWork {
   unsigned long long *ctr, *stats;
   float *raw, *beat;
   int *prim_out; float *t_out;
   QRay *queue; Hit *hits;
};
#define RT_RAYWORK_HEADER 65536u
static size_t raywork_bytes(size_t n) { return RT_RAYWORK_HEADER + align_up(28 * n, 256) + align_up(sizeof(QRay) * n, 256) + align_up(sizeof(Hit) * n, 256); }
static RayWork raywork_carve(char *base, size_t n)
{
   static_assert(sizeof(unsigned long long) * (RT_CTR_WORDS + ST_COUNT * RT_STAT_REPL) <= RT_RAYWORK_HEADER, "counters and statistics must fit the header");
   RayWork w;
   w.ctr = (unsigned long long *)base; w.stats = w.ctr + RT_CTR_WORDS;
   char *p = base + RT_RAYWORK_HEADER;
   w.raw = (float *)p; w.beat = (float *)(p + 24 * n);
   w.prim_out = (int *)p; w.t_out = (float *)(p + 4 * n);        /* the results reuse the region of the raw rays */
   p += align_up(28 * n, 256);
   w.queue = (QRay *)p; p += align_up(sizeof(QRay) * n, 256);
   w.hits = (Hit *)p;
   return w;
}

/* The body both entry points share: nothing here allocates. */
static int trace_rays_queued(rt_scene *s, const RayWork &w, int64_t n, const float *rays_host, const float *to_beat_host,
                             int32_t *prim_host, float *t_host, hipStream_t S, unsigned long long *aborted);

static int trace_rays_in(rt_scene *s, const RayWork &w, int64_t n, const float *rays_host, const float *to_beat_host,
                         int32_t *prim_host, float *t_host)
{
   hipStream_t S = nullptr;
   unsigned long long aborted[RT_STAT_REPL];
   memset(aborted, 0, sizeof aborted);
   int rc = trace_rays_queued(s, w, n, rays_host, to_beat_host, prim_host, t_host, S, aborted);
   /* whatever was queued before a failure may still be writing into the caller's arrays (or `aborted`): it ends first */
   hipError_t e = hipStreamSynchronize(S);
   if (rc != RT_OK) return rc;
   if (e != hipSuccess) { rt_set_error("hipStreamSynchronize failed: %s", hipGetErrorString(e)); return RT_ERR_HIP; }
   for (int r = 0; r < RT_STAT_REPL; r++)
      if (aborted[r]) { rt_set_error("traversal aborted: step cap reached (corrupt scene?)"); return RT_ERR_HIP; }
   return RT_OK;
}

static int trace_rays_queued(rt_scene *s, const RayWork &w, int64_t n, const float *rays_host, const float *to_beat_host,
                             int32_t *prim_host, float *t_host, hipStream_t S, unsigned long long *aborted)
{
   size_t lds_bytes = 0;
   { int rc = rt_trace_lds_bytes(s, &lds_bytes); if (rc != RT_OK) return rc; }
   HIP_TRY(hipMemsetAsync(w.ctr, 0, sizeof(unsigned long long) * (RT_CTR_WORDS + ST_COUNT * RT_STAT_REPL), S));
   HIP_TRY(hipMemcpyAsync(w.raw, rays_host, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice, S));
   if (to_beat_host) HIP_TRY(hipMemcpyAsync(w.beat, to_beat_host, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, S));
   hipLaunchKernelGGL(k_pack_rays, dim3(grid_for(n)), dim3(RT_BLOCK), 0, S, (const float *)w.raw, to_beat_host ? (const float *)w.beat : (const float *)nullptr,
                      (long long)n, w.queue);
   TraceArgs a; memset(&a, 0, sizeof a);
   a.rays = w.queue; a.n = a.n_fetch = (unsigned long long)n; a.hits = w.hits; a.next = w.ctr; a.stats = w.stats;
   rt_launch_trace(s, OUT_HIT, false, a, (unsigned long long)n, lds_bytes, S);
   hipLaunchKernelGGL(k_unpack_hits, dim3(grid_for(n)), dim3(RT_BLOCK), 0, S, (const Hit *)w.hits, (long long)n, s->d.prim_ids, s->d.n_prims, w.prim_out, w.t_out);
   HIP_TRY(hipGetLastError());
   HIP_TRY(hipMemcpyAsync(prim_host, w.prim_out, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, S));
   HIP_TRY(hipMemcpyAsync(t_host, w.t_out, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, S));
   HIP_TRY(hipMemcpyAsync(aborted, w.stats + ST_ABORTED * RT_STAT_REPL, sizeof(unsigned long long) * RT_STAT_REPL, hipMemcpyDeviceToHost, S));
   return RT_OK;
}

static int trace_rays_check(const rt_scene *s, int64_t n, const float *rays_host, int32_t *prim_host, float *t_host, const char *who)
{
   if (!s || !rays_host || !prim_host || !t_host || n < 0) { rt_set_error("%s: bad argument", who); return RT_ERR_BAD_ARGUMENT; }
   if (s->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   return RT_OK;
}

extern "C" int rt_trace_rays(rt_scene *s, int64_t n, const float *rays_host, const float *to_beat_host,
                             int32_t *prim_host, float *t_host)
{
   { int rc = trace_rays_check(s, n, rays_host, prim_host, t_host, "rt_trace_rays"); if (rc != RT_OK) return rc; }
   if (n == 0) return RT_OK;
   if (n > 0x7fffffffLL) { rt_set_error("rt_trace_rays: too many rays"); return RT_ERR_TOO_LARGE; }
   HIP_TRY(hipSetDevice(s->device));
   char *base = nullptr;
   hipError_t e = hipMalloc((void **)&base, raywork_bytes((size_t)n));
   if (e != hipSuccess) { rt_set_error("rt_trace_rays: hipMalloc(%zu) failed: %s", raywork_bytes((size_t)n), hipGetErrorString(e)); return RT_ERR_HIP; }
   int rc = trace_rays_in(s, raywork_carve(base, (size_t)n), n, rays_host, to_beat_host, prim_host, t_host);
   (void)hipFree(base);
   return rc;
}

/* The working memory of rt_trace_rays_static: one array per device in the code object, held until the process ends.
 * In host code the array's name is only a handle for hipGetSymbolAddress; kernels and copies get the device address
 * that call returns for the device that is current. */
static __device__ __align__(256) unsigned char scratch[RT_RAYWORK_HEADER + (28 + sizeof(QRay) + sizeof(Hit)) * (size_t)RT_TRACE_RAYS_STATIC_MAX + 3 * 256];
#define RT_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static char *g_scratch_ptr[RT_MAX_DEVICES];      /* per device: the address of `scratch` there, once looked up */
static std::mutex g_scratch_in_use[RT_MAX_DEVICES];   /* per device: held for the length of a call that works in `scratch` */

/* with `device` current: the device address of that device's copy of `scratch` */
int rt_scratch_address(int device, char **out)
{
   if (device < 0 || device >= RT_MAX_DEVICES) { rt_set_error("device %d is beyond the %d this library keeps working memory for", device, RT_MAX_DEVICES); return RT_ERR_BAD_ARGUMENT; }
   std::lock_guard<std::mutex> lock(g_scratch_mutex);
   if (!g_scratch_ptr[device]) {
      void *p = nullptr;
      hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
      if (e != hipSuccess || !p) { rt_set_error("hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(e)); return RT_ERR_HIP; }
      g_scratch_ptr[device] = (char *)p;
   }
   *out = g_scratch_ptr[device];
   return RT_OK;
}

extern "C" int rt_trace_rays_static(rt_scene *s, int64_t n, const float *rays_host, const float *to_beat_host,
                                    int32_t *prim_host, float *t_host)
{
   { int rc = trace_rays_check(s, n, rays_host, prim_host, t_host, "rt_trace_rays_static"); if (rc != RT_OK) return rc; }
   if (n == 0) return RT_OK;
   if (n > (int64_t)RT_TRACE_RAYS_STATIC_MAX) {
      rt_set_error("rt_trace_rays_static: %lld rays exceed the %d the static working memory holds; use rt_trace_rays", (long long)n, RT_TRACE_RAYS_STATIC_MAX);
      return RT_ERR_TOO_LARGE;
   }
   HIP_TRY(hipSetDevice(s->device));
   char *base = nullptr;
   { int rc = rt_scratch_address(s->device, &base); if (rc != RT_OK) return rc; }
   if (raywork_bytes((size_t)n) > sizeof scratch) { rt_set_error("internal: static working memory too small"); return RT_ERR_TOO_LARGE; }
   /* every scene on a device shares that device's array: calls from several threads take turns */
   std::lock_guard<std::mutex> in_use(g_scratch_in_use[s->device]);
   return trace_rays_in(s, raywork_carve(base, (size_t)n), n, rays_host, to_beat_host, prim_host, t_host);
}

