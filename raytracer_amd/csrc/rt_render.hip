/*
 * rt_render.hip -- the frame driver of the per-pixel render path on MI355X (gfx950, wave64): the body of
 * launch_kernel after scene set-up (kernel.cu:735-760).  The kernels it launches are in rt_trace.h (k_trace) and
 * rt_shade.h (the rest); the scene object is made in rt_scene.hip; loose rays are traced in rt_rays.hip.
 *
 * Replaces the reference's rayTrace megakernel and its helpers (kernel.cu:67-382, 466-570).  The reference
 * gives every sample a thread that recurses to depth 6 (shadeObject<6> -> getReflection<5> -> ...); here a
 * frame is four kinds of launches:
 *
 *   k_trace         every ray of the frame in one persistent launch: primary rays, made from the sample slot
 *                   when a lane fetches it (generateCameraRays, kernel.cu:477-498); their closest hits
 *                   (getClosestIntersection, kernel.cu:83-200); the reflection / refraction rays those spawn
 *                   (getReflection / getRefraction, kernel.cu:209-276), to depth 6; and one shadow feeler per
 *                   light per hit (isInShadow, kernel.cu:202-207).  Follow-up work of a hit goes onto a list
 *                   private to the wave that found it, so waves never wait for each other.
 *   k_light         the light sum of every hit with the shadow flags (the loop of shadeObject, kernel.cu:347-358;
 *                   PhongShader.h, CookTorranceShader.h; getColor with textures, kernel.cu:323-332)
 *   k_combine_tree  colour = total*(1-refl-alpha) + refl*reflected + alpha*refracted (kernel.cu:375-376) over
 *                   each primary's bounce tree, children first: the reference's expression tree with the same
 *                   operands in the same order, no float atomics; also notes how deep each tile's trees were,
 *                   which orders the tiles of the next frame's k_trace (costliest first)
 *   k_resolve       s x s box filter to 8 bits (averageBufferColors, kernel.cu:545-570)
 *
 * Each ray walks the tree in the reference's visiting order (near child first with its `<=` rule, a deferred
 * sibling taken only if still closer than the hit so far), because the reference resolves near-ties between
 * primitives by visiting order.  The per-lane stack of deferred siblings lives in LDS, one column per lane.
 */
#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <type_traits>
#include <vector>

#include "rt_scene_state.h"
#include "rt_shade.h"

static int isqrt_trunc(int n) { return (int)std::sqrt((double)n); }   /* kernel.cu:707 */

extern "C" int rt_tile_count(int width, int height, int tile_first, int tile_stride)
{
   if (width <= 0 || height <= 0) return 0;
   if (tile_stride <= 0) tile_stride = 1;
   int tiles = ((width + RT_TILE - 1) / RT_TILE) * ((height + RT_TILE - 1) / RT_TILE);
   if (tile_first < 0 || tile_first >= tiles) return 0;
   return (tiles - tile_first + tile_stride - 1) / tile_stride;
}

static DCamera make_camera(const rt_camera &c)
{
   /* normalize(lookAt - pos) on the host, in the same order the device uses (kernel.cu:717-718) */
   DCamera d;
   d.pos.x = c.pos[0]; d.pos.y = c.pos[1]; d.pos.z = c.pos[2];
   d.up.x = c.up[0]; d.up.y = c.up[1]; d.up.z = c.up[2];
   d.right.x = c.right[0]; d.right.y = c.right[1]; d.right.z = c.right[2];
   float lx = c.look_at[0] - c.pos[0], ly = c.look_at[1] - c.pos[1], lz = c.look_at[2] - c.pos[2];
   float sqr = lx * lx + ly * ly + lz * lz;
   float inv = 1.0f / std::sqrt(sqr);
   d.look_dir.x = lx * inv; d.look_dir.y = ly * inv; d.look_dir.z = lz * inv;
   return d;
}

static int arena_reserve(Arena &a, size_t bytes)
{
   if (bytes <= a.cap) return RT_OK;
   if (a.base) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(a.base)); a.base = nullptr; a.cap = 0; }
   size_t want = align_up(bytes + bytes / 4, 1 << 20);
   hipError_t e = hipMalloc((void **)&a.base, want);
   if (e != hipSuccess) { rt_set_error("hipMalloc(%zu) for ray queues failed: %s", want, hipGetErrorString(e)); return RT_ERR_HIP; }
   a.cap = want;
   return RT_OK;
}

struct Timer {
   hipEvent_t a, b;
   bool ok = false, ended = false;
   float *acc = nullptr;
};

/* timing events of one frame; the events themselves are kept by the scene from frame to frame */
struct EventPool {
   std::vector<hipEvent_t> &spare;
   std::deque<Timer> timers;   /* a deque: begin() hands out references that must survive later begin()s */
   explicit EventPool(std::vector<hipEvent_t> &cache) : spare(cache) {}
   ~EventPool()
   {
      for (auto &t : timers) if (t.ok) { spare.push_back(t.a); spare.push_back(t.b); }
   }
   bool take(hipEvent_t *e)
   {
      if (!spare.empty()) { *e = spare.back(); spare.pop_back(); return true; }
      return hipEventCreate(e) == hipSuccess;
   }
   /* `wanted` false: a timer that records nothing (an event between two kernels of a stream costs a few microseconds of
    * the stream's time: only the runs that ask for the detail pay for it) */
   Timer &begin(hipStream_t st, float *acc, bool wanted = true)
   {
      Timer t; t.acc = acc;
      if (!wanted) { timers.push_back(t); return timers.back(); }
      t.ok = take(&t.a);
      if (t.ok && !take(&t.b)) { spare.push_back(t.a); t.ok = false; }
      if (t.ok) (void)hipEventRecord(t.a, st);
      timers.push_back(t);
      return timers.back();
   }
   void end(Timer &t, hipStream_t st) { if (t.ok) { (void)hipEventRecord(t.b, st); t.ended = true; } }
   void collect()
   {
      for (auto &t : timers) {
         float ms = 0;
         if (t.ok && t.ended && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess && t.acc) *t.acc += ms;
      }
   }
};

enum { VIEW_FRAME = 0, VIEW_PRIMARY = 1, VIEW_SAMPLES = 2 };


static long long max_batch_slots()
{
   const char *e = getenv("RT_MAX_BATCH_SLOTS");
   long long v = e ? atoll(e) : 0;
   return v > 0 ? v : (long long)(1 << 26);
}

int rt_trace_lds_bytes(const rt_scene *s, size_t *out)
{
   /* the stack holds at most one deferred sibling per internal level; RT_LDS_STACK of them live in LDS */
   int depth = rt_stack_depth(s);
   if (depth > RT_LDS_STACK) depth = RT_LDS_STACK;
   /* in front of the stacks: the leaves a guided walk keeps, RT_GUIDE_WORDS words per lane */
   *out = sizeof(StackEntry) * (size_t)RT_BLOCK * (size_t)depth + sizeof(unsigned) * (size_t)RT_BLOCK * RT_GUIDE_WORDS;
   return RT_OK;
}

/* room for the lists of every wave a launch can have */
static int ensure_lists(rt_scene *s)
{
   const unsigned waves = (unsigned)s->trace_blocks * (RT_BLOCK / 64);
   if (s->d_lists && s->list_waves == waves) return RT_OK;
   if (s->d_lists) { (void)hipFree(s->d_lists); s->d_lists = nullptr; }
   hipError_t e = hipMalloc((void **)&s->d_lists, sizeof(uint2) * (size_t)waves * s->list_cap);
   if (e != hipSuccess) { rt_set_error("hipMalloc of the spawn lists (%u x %u entries) failed: %s", waves, s->list_cap, hipGetErrorString(e)); return RT_ERR_HIP; }
   s->list_waves = waves;
   return RT_OK;
}

void rt_launch_trace(const rt_scene *s, int out_mode, bool count, TraceArgs a, unsigned long long n_hint, size_t lds_bytes,
                         hipStream_t stream, int filter_kinds)
{
   DScene d = s->d;
   d.filter_kinds = filter_kinds;
   if (!filter_kinds) d.filter_root = RT_ROOT_EMPTY;
   unsigned long long want = (n_hint + RT_BLOCK - 1) / RT_BLOCK;
   unsigned blocks = (unsigned)(want < (unsigned long long)s->trace_blocks ? want : (unsigned long long)s->trace_blocks);
   if (blocks == 0) return;
   a.refill_idle = s->refill_idle; a.node_burst = s->node_burst; a.leaf_trigger = s->leaf_trigger;
   a.reserve = s->reserve;
   if (!a.stats) a.stats = s->d_stats;
   a.spill = s->d_spill;
   a.guide = s->guide_ok ? 1 : 0;
   for (int j = 0; j < 3; j++) { a.tree_lo[j] = s->tree_lo[j]; a.tree_hi[j] = s->tree_hi[j]; }
   const bool spill = rt_stack_depth(s) > RT_LDS_STACK;
   const bool two = d.filter_root != RT_ROOT_EMPTY;
   TraceKernArgs ka; ka.sc = d; ka.a = a;
#define RT_LAUNCH_TRACE2(O, C, S) do { if (two) hipLaunchKernelGGL((k_trace<O, C, S, true>), dim3(blocks), dim3(RT_BLOCK), lds_bytes, stream, ka); \
                                       else hipLaunchKernelGGL((k_trace<O, C, S, false>), dim3(blocks), dim3(RT_BLOCK), lds_bytes, stream, ka); } while (0)
#define RT_LAUNCH_TRACE(O, C) do { if (spill) RT_LAUNCH_TRACE2(O, C, true); else RT_LAUNCH_TRACE2(O, C, false); } while (0)
   if (out_mode == OUT_CHAIN) { if (count) RT_LAUNCH_TRACE(OUT_CHAIN, true); else RT_LAUNCH_TRACE(OUT_CHAIN, false); }
   else { if (count) RT_LAUNCH_TRACE(OUT_HIT, true); else RT_LAUNCH_TRACE(OUT_HIT, false); }
#undef RT_LAUNCH_TRACE
#undef RT_LAUNCH_TRACE2
}

/* device counters of the chunk in flight (rt_scene::d_ctr) */
struct ChunkCtr {
   unsigned long long *next_chain;
   unsigned *tail;
};
static ChunkCtr chunk_ctr(const rt_scene *s, int ring)
{
   (void)ring;
   ChunkCtr c;
   c.next_chain = s->d_ctr;
   c.tail = reinterpret_cast<unsigned *>(s->d_ctr + 8 * RT_FETCH_STREAMS);
   return c;
}

/* Which of the selected tiles have to be traced at all.
 *
 * Every primary ray lies on a line through E = pos - look_dir: sample (u, v) starts at pos + u/2 right + v/2 up and
 * runs along u/2 right + v/2 up + look_dir (generateCameraRays, kernel.cu:477-498).  The rays of a 32x32-pixel tile
 * therefore lie inside a pyramid with apex E, spanned by the corner directions of the tile's (u, v) rectangle.  If the
 * box that holds every primitive's bounding box lies wholly beyond one of the pyramid's four side planes (or wholly
 * behind its apex), no ray of the tile enters any box of the tree, the reference's walk (kernel.cu:136-158) ends at
 * the root and the sample is the background: nothing of such a tile is made, traced, stored or read.
 *
 * The margins are large against the arithmetic they stand in for: the tile's rectangle is widened by a whole sample
 * spacing and by a thousandth of the image on every side (jitter moves a sample by half a spacing; the float
 * arithmetic of the ray set-up and of the box tests is good to ~1e-6 of the distances involved), the box by a
 * thousandth of its size and of its distance from E plus the padding the tree's levels add to their boxes
 * (BoundingBox.h: 1e-4 per construction).  Computed in double.  Scenes with planes are not culled (planes are
 * tested for every ray, kernel.cu:189-196). */
static void cull_tiles(const rt_scene *s, const DCamera &cam, const DGrid &g, int n_sel, bool finer, std::vector<unsigned char> &live, long long *n_live)
{
   live.assign((size_t)n_sel, 1);
   *n_live = n_sel;
   if (!s->bounds_ok) return;
   const double E[3] = {(double)cam.pos.x - cam.look_dir.x, (double)cam.pos.y - cam.look_dir.y, (double)cam.pos.z - cam.look_dir.z};
   const double R[3] = {cam.right.x, cam.right.y, cam.right.z}, U[3] = {cam.up.x, cam.up.y, cam.up.z};
   const double Ld[3] = {cam.look_dir.x, cam.look_dir.y, cam.look_dir.z};
   double lo[3], hi[3], ext = 0.0, dist = 0.0;
   for (int j = 0; j < 3; j++) {
      ext = std::max(ext, s->bounds_hi[j] - s->bounds_lo[j]);
      const double c = 0.5 * (s->bounds_lo[j] + s->bounds_hi[j]) - E[j];
      dist += c * c;
   }
   dist = std::sqrt(dist);
   /* the float arithmetic of the ray set-up (kernel.cu:493-497) rounds a ray's origin and direction by a few units in
    * the last place of the camera position's coordinates: a camera far from the origin of the coordinates gets margins
    * that grow with that */
   const double pos_ulp = std::max(std::max(std::fabs((double)cam.pos.x), std::fabs((double)cam.pos.y)), std::fabs((double)cam.pos.z)) * (1.0 / 4194304.0);
   /* (a sphere seen from afar reports hits off its surface by up to 2^-22 distance^2 / radius, Sphere.h:42-66: that is
    * allowed for too, twice over) */
   const double pad = 1e-3 * (ext + dist) + 2e-4 * (double)(s->host.tree_depth + s->host.filter_depth + 4) + 1e-3 + 4.0 * pos_ulp * (2.0 + dist) +
                      4.8e-7 * (double)s->host.sphere_coeff * (dist + ext) * (dist + ext);
   if (!std::isfinite(pad)) return;
   /* the tree the finer test walks: the second tree if the scene has one (its boxes are the tighter ones), else the tree
    * the rays are answered from; both hold every primitive's box in their leaves' places */
   const rt_bvh_node *fine = nullptr;
   if (finer && s->cull_fine > 0) {
      if (s->host.n_filter_nodes > 0 && s->host.filter_nodes) fine = s->host.filter_nodes;
      else if (s->host.n_nodes > 0 && s->host.nodes && s->host.root == 0) fine = s->host.nodes;
   }
   const int fine_budget = s->cull_fine;
   /* (the box that showed the tile before this one to be in sight is looked at first: a floor seen through every tile
    * answers at once) */
   float seen_mn[3] = {0, 0, 0}, seen_mx[3] = {0, 0, 0}; bool have_seen = false;
   for (int j = 0; j < 3; j++) { lo[j] = s->bounds_lo[j] - pad; hi[j] = s->bounds_hi[j] + pad; }
   double corner[8][3];
   for (int c = 0; c < 8; c++)
      for (int j = 0; j < 3; j++) corner[c][j] = ((c >> j) & 1 ? hi[j] : lo[j]) - E[j];
   const double len_r = std::sqrt(R[0] * R[0] + R[1] * R[1] + R[2] * R[2]), len_u = std::sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
   const double pad_u = 2.0 / (double)g.res_w + 2e-3 + (len_r > 0.0 ? 4.0 * pos_ulp / len_r : 0.0);
   const double pad_v = 2.0 / (double)g.res_h + 2e-3 + (len_u > 0.0 ? 4.0 * pos_ulp / len_u : 0.0);
   if (!std::isfinite(pad_u) || !std::isfinite(pad_v)) return;
   auto dir = [&](double u, double v, double *d) { for (int j = 0; j < 3; j++) d[j] = 0.5 * u * R[j] + 0.5 * v * U[j] + Ld[j]; };
   auto cross = [](const double *a, const double *b, double *c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; };
   auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
   long long kept = 0;
   for (int k = 0; k < n_sel; k++) {
      const long long tile = (long long)g.tile_first + (long long)k * g.tile_stride;
      const long long tx = tile % g.tiles_x, ty = tile / g.tiles_x;
      /* the (u, v) rectangle of the tile's samples: sample X has u = 2 X / res_w - 1 + jitter, |jitter| <= 1/(2 res_w) */
      const double u0 = 2.0 * (double)(tx * g.ts) / (double)g.res_w - 1.0 - pad_u;
      const double u1 = 2.0 * (double)((tx + 1) * g.ts) / (double)g.res_w - 1.0 + pad_u;
      const double v0 = 2.0 * (double)(ty * g.ts) / (double)g.res_h - 1.0 - pad_v;
      const double v1 = 2.0 * (double)((ty + 1) * g.ts) / (double)g.res_h - 1.0 + pad_v;
      double d00[3], d10[3], d01[3], d11[3], mid[3];
      dir(u0, v0, d00); dir(u1, v0, d10); dir(u0, v1, d01); dir(u1, v1, d11); dir(0.5 * (u0 + u1), 0.5 * (v0 + v1), mid);
      /* the four side planes through E, each through two neighbouring corner directions; inside = the side `mid` is on */
      const double *edge[4][2] = {{d00, d10}, {d10, d11}, {d11, d01}, {d01, d00}};
      bool outside = false;
      for (int e = 0; e < 4 && !outside; e++) {
         double nrm[3]; cross(edge[e][0], edge[e][1], nrm);
         const double len = std::sqrt(dot(nrm, nrm)), m = dot(nrm, mid);
         if (!(len > 0.0) || !std::isfinite(len) || !(std::fabs(m) > 1e-9 * len)) continue;      /* a degenerate camera culls nothing */
         const double sign = m > 0.0 ? 1.0 : -1.0;
         bool all_out = true;
         for (int c = 0; c < 8 && all_out; c++) all_out = sign * dot(nrm, corner[c]) < 0.0;
         outside = all_out;
      }
      if (!outside) {
         /* ... or wholly behind the apex: beyond the plane through E across the pyramid's middle direction, which every
          * direction of the pyramid points away from */
         bool ahead = dot(mid, mid) > 0.0;
         const double *cd[4] = {d00, d10, d11, d01};
         for (int e = 0; e < 4 && ahead; e++) ahead = dot(mid, cd[e]) > 0.0;
         if (ahead) {
            bool all_behind = true;
            for (int c = 0; c < 8 && all_behind; c++) all_behind = dot(mid, corner[c]) < 0.0;
            outside = all_behind;
         }
      }
      if (!outside && fine) {
         /* The scene's box is in sight of the tile; is any primitive's?  Down the tree from its root, leaving out every
          * box that lies wholly beyond one of the four side planes (its corner furthest on the inner side is still
          * outside), until a leaf's box is found in sight -- or the budget of boxes is spent, which counts as found. */
         double pn[4][3]; int np_ = 0;
         for (int e = 0; e < 4; e++) {
            double nrm[3]; cross(edge[e][0], edge[e][1], nrm);
            const double len = std::sqrt(dot(nrm, nrm)), m = dot(nrm, mid);
            if (!(len > 0.0) || !std::isfinite(len) || !(std::fabs(m) > 1e-9 * len)) continue;
            const double sign = m > 0.0 ? 1.0 : -1.0;
            for (int j = 0; j < 3; j++) pn[np_][j] = sign * nrm[j];
            np_++;
         }
         auto in_sight = [&](const float *mn, const float *mx) {
            for (int e = 0; e < np_; e++) {
               double most = 0.0;
               for (int j = 0; j < 3; j++) most += pn[e][j] * ((pn[e][j] > 0.0 ? (double)mx[j] + pad : (double)mn[j] - pad) - E[j]);
               if (most < 0.0) return false;
            }
            return true;
         };
         bool found = np_ < 4;                  /* a degenerate pyramid: no finer answer */
         if (!found && have_seen && in_sight(seen_mn, seen_mx)) found = true;
         int spent = 0, sp = 0, stack[128];
         stack[sp++] = 0;
         while (sp > 0 && !found) {
            const rt_bvh_node &nd = fine[stack[--sp]];
            if (++spent > fine_budget) { found = true; break; }
            const int32_t child[2] = {nd.right, nd.left};
            const float *mn[2] = {nd.rmin, nd.lmin}, *mx[2] = {nd.rmax, nd.lmax};
            for (int c = 0; c < 2 && !found; c++) {
               if (!in_sight(mn[c], mx[c])) continue;
               if (child[c] < 0) { found = true; have_seen = true; for (int j = 0; j < 3; j++) { seen_mn[j] = mn[c][j]; seen_mx[j] = mx[c][j]; } }
               else if (sp < 128) stack[sp++] = child[c];
               else found = true;
            }
         }
         outside = !found;
      }
      if (outside) live[(size_t)k] = 0; else kept++;
   }
   *n_live = kept;
}

/* A start order for a frame that has no costs of an earlier one to go by: the selected tiles by their distance, in the
 * picture, from where the middle of the scene's box appears (the middle of the frame if that is behind the camera). */
static void first_order_guess(const rt_scene *s, const DCamera &cam, const DGrid &g, int n_sel, int *order)
{
   double cx = 0.5 * g.width, cy = 0.5 * g.height;
   {
      /* the middle of the box, seen from E = pos - look_dir, as a multiple of u/2 right + v/2 up + look_dir */
      const double E[3] = {(double)cam.pos.x - cam.look_dir.x, (double)cam.pos.y - cam.look_dir.y, (double)cam.pos.z - cam.look_dir.z};
      const double A[3][3] = {{0.5 * cam.right.x, 0.5 * cam.up.x, cam.look_dir.x}, {0.5 * cam.right.y, 0.5 * cam.up.y, cam.look_dir.y},
                              {0.5 * cam.right.z, 0.5 * cam.up.z, cam.look_dir.z}};
      double d[3];
      for (int j = 0; j < 3; j++) d[j] = 0.5 * (s->bounds_lo[j] + s->bounds_hi[j]) - E[j];
      const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                         A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
      if (std::isfinite(det) && std::fabs(det) > 1e-30) {
         auto det_with = [&](int col) {
            double M[3][3];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[r][c] = c == col ? d[r] : A[r][c];
            return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                   M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
         };
         const double a = det_with(0) / det, b = det_with(1) / det, c = det_with(2) / det;
         if (std::isfinite(a) && std::isfinite(b) && c > 1e-9) {
            const double u = a / c, v = b / c;
            if (std::fabs(u) < 4.0 && std::fabs(v) < 4.0) { cx = 0.5 * (u + 1.0) * g.width; cy = 0.5 * (v + 1.0) * g.height; }
         }
      }
   }
   std::vector<std::pair<double, int>> key((size_t)n_sel);
   for (int k = 0; k < n_sel; k++) {
      const long long tile = (long long)g.tile_first + (long long)k * g.tile_stride;
      const double x = ((double)(tile % g.tiles_x) + 0.5) * RT_TILE - cx, y = ((double)(tile / g.tiles_x) + 0.5) * RT_TILE - cy;
      key[(size_t)k] = std::make_pair(x * x + y * y, k);
   }
   std::sort(key.begin(), key.end());
   for (int k = 0; k < n_sel; k++) order[k] = key[(size_t)k].second;
}

extern "C" int rt_tiles_live(const rt_scene *s, const rt_render_params *p, uint8_t *live, int n)
{
   if (!s || !p || !live) { rt_set_error("rt_tiles_live: null argument"); return RT_ERR_BAD_ARGUMENT; }
   if (p->width <= 0 || p->height <= 0 || p->sample_count <= 0) { rt_set_error("rt_tiles_live: non-positive size"); return RT_ERR_BAD_ARGUMENT; }
   const int sroot = isqrt_trunc(p->sample_count);
   if (sroot * sroot != p->sample_count) { rt_set_error("Invalid sample count: %d. Sample count for anti aliasing must have an integer square root", p->sample_count); return RT_ERR_BAD_SAMPLE_COUNT; }
   if (!s->has_camera) { rt_set_error("scene has no camera"); return RT_ERR_NO_CAMERA; }
   const int tile_stride = p->tile_stride > 0 ? p->tile_stride : 1;
   const int n_sel = rt_tile_count(p->width, p->height, p->tile_first, tile_stride);
   if (n < n_sel) { rt_set_error("rt_tiles_live: room for %d tiles, the selection has %d", n, n_sel); return RT_ERR_BAD_ARGUMENT; }
   DGrid g; memset(&g, 0, sizeof g);
   g.width = p->width; g.height = p->height; g.sroot = sroot;
   g.res_w = p->width * sroot; g.res_h = p->height * sroot;
   g.tiles_x = (p->width + RT_TILE - 1) / RT_TILE;
   g.tile_first = p->tile_first; g.tile_stride = tile_stride; g.ts = RT_TILE * sroot;
   std::vector<unsigned char> v((size_t)(n_sel > 0 ? n_sel : 0), 1);
   long long kept = n_sel;
   if (n_sel > 0 && s->cull_enabled && s->host.n_planes == 0 && s->bounds_ok) cull_tiles(s, make_camera(s->camera), g, n_sel, true, v, &kept);
   for (int k = 0; k < n_sel; k++) live[k] = v[(size_t)k];
   return RT_OK;
}

/* The frame driver: body of launch_kernel after scene set-up (kernel.cu:735-760), over the tiles this call
 * selects.  The tiles are cut into chunks (one, whenever the frame fits); a chunk is four launches on the scene's
 * stream: k_trace (every ray of the chunk), k_light, k_combine_tree, k_resolve.  Nothing is read back while the
 * frame is in flight; the queue capacity per primary ray and the length of the waves' spawn lists are guesses
 * that the device checks, and a frame that overflowed one is rendered again with a larger one. */
static int render_impl(rt_scene *s, const rt_render_params *p, int view, void *out_dev, int *prim_dev,
                       float *t_dev, float *rgb_dev, hipStream_t stream, rt_stats *stats_out)
{
#ifdef RT_DEBUG_HOST
   static double dbg_acc[4]; static int dbg_n; static std::chrono::steady_clock::time_point dbg_prev_exit;
   auto dbg_t0 = std::chrono::steady_clock::now(); std::chrono::steady_clock::time_point dbg_t1, dbg_t2;
#endif
   if (!s || !p) { rt_set_error("render: null argument"); return RT_ERR_BAD_ARGUMENT; }
   if (s->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   if (p->width <= 0 || p->height <= 0 || p->sample_count <= 0) { rt_set_error("render: non-positive size"); return RT_ERR_BAD_ARGUMENT; }
   int sroot = isqrt_trunc(p->sample_count);
   if (sroot * sroot != p->sample_count) {
      rt_set_error("Invalid sample count: %d. Sample count for anti aliasing must have an integer square root", p->sample_count);
      return RT_ERR_BAD_SAMPLE_COUNT;
   }
   if (!s->has_camera) { rt_set_error("scene has no camera"); return RT_ERR_NO_CAMERA; }
   if (p->shading != RT_PHONG && p->shading != RT_COOK_TORRANCE) { rt_set_error("Improper shading type specified"); return RT_ERR_BAD_ARGUMENT; }
   int max_rec = p->max_recurse > 0 ? p->max_recurse : RT_MAX_RECURSE;
   if (max_rec > RT_MAX_RECURSE) { rt_set_error("max_recurse > %d", RT_MAX_RECURSE); return RT_ERR_BAD_ARGUMENT; }
   if ((long long)p->width * sroot > 0x3fffffff / (long long)(p->height * (long long)sroot)) { rt_set_error("frame too large"); return RT_ERR_TOO_LARGE; }
   HIP_TRY(hipSetDevice(s->device));
   if (view == VIEW_PRIMARY) max_rec = 1;
   s->last_frame.valid = false;

   int tile_first = p->tile_first, tile_stride = p->tile_stride > 0 ? p->tile_stride : 1;
   int n_sel = rt_tile_count(p->width, p->height, tile_first, tile_stride);
   rt_stats st; memset(&st, 0, sizeof st);
   if (n_sel <= 0) { if (stats_out) *stats_out = st; return RT_OK; }

   DGrid g;
   g.width = p->width; g.height = p->height; g.sroot = sroot;
   g.res_w = p->width * sroot; g.res_h = p->height * sroot;
   g.tiles_x = (p->width + RT_TILE - 1) / RT_TILE;
   g.tile_first = tile_first; g.tile_stride = tile_stride; g.k0 = 0;
   g.ts = RT_TILE * sroot; g.patches_per_row = g.ts / 8;
   g.per_tile = (unsigned)(g.ts * g.ts);
   g.per_tile_div = make_fastdiv(g.per_tile); g.patches_per_row_div = make_fastdiv((unsigned)g.patches_per_row);
   g.tiles_x_div = make_fastdiv((unsigned)g.tiles_x);
   g.inv_res_w = 1.0f / (float)g.res_w; g.inv_res_h = 1.0f / (float)g.res_h;
   g.jitter = p->jitter; g.seed = p->seed;
   DCamera cam = make_camera(s->camera);
   const DScene &sc = s->d;
   const int L = sc.n_lights;
   const bool count = (s->flags & RT_SCENE_COUNTERS) != 0;
   const long long per_tile = (long long)g.ts * g.ts;
   size_t lds_bytes = 0;
   { int rc = rt_trace_lds_bytes(s, &lds_bytes); if (rc != RT_OK) return rc; }
   hipStream_t M = s->own_stream ? s->chain : stream;

   /* which rays start in the second tree this frame: the scene's choice, or the next combination of the try-out */
   int filter_kinds = s->filter_choice;
   const long long probe_key[4] = {p->width, p->height, p->sample_count, (long long)tile_first * 65536 + tile_stride};
   const bool probing = s->filter_auto && view == VIEW_FRAME && sc.filter_root != RT_ROOT_EMPTY;
   if (probing) {
      if (memcmp(probe_key, s->probe_key, sizeof probe_key) != 0) { memcpy(s->probe_key, probe_key, sizeof probe_key); s->probe_step = 0; }
      static const int order[5] = {-1, 3, 1, 2, 0};       /* the first frame of a size is not timed (it has no tile order yet): it runs on the current choice */
      filter_kinds = s->probe_step < 5 && order[s->probe_step] >= 0 ? order[s->probe_step] : s->filter_choice;
   }

   int attempts = 0;
   bool culled_frame = false;      /* the frame left tiles out (s->tile_live says which) */
   long long last_hits_offset = -1;   /* a one-launch frame: where its hits lie in the arena */
   for (int attempt = 0;; attempt++) {
      memset(&st, 0, sizeof st);
      attempts = attempt + 1;
      /* chunk size: a chunk is bounded by its queue entries; a frame is one chunk whenever that allows */
      double F = max_rec > 1 ? (double)s->node_factor : 1.0;
      long long slots_cap = max_batch_slots();
      long long tiles_cap = slots_cap / per_tile; if (tiles_cap < 1) tiles_cap = 1;
      long long tiles_min = (128 * 1024) / per_tile; if (tiles_min < 1) tiles_min = 1;
      long long chunk_tiles = n_sel;
      { const char *e = getenv("RT_CHUNK_TILES"); if (e && atoll(e) > 0) chunk_tiles = atoll(e); else if (chunk_tiles < tiles_min) chunk_tiles = tiles_min; }
      if (chunk_tiles > tiles_cap) chunk_tiles = tiles_cap;
      if (chunk_tiles > n_sel) chunk_tiles = n_sel;
      const long long n_chunks = (n_sel + chunk_tiles - 1) / chunk_tiles;
      const int ring = 1;        /* the chunks of a frame run one after the other on one stream: one slice of the arena serves them all */

      /* one arena slice per chunk in flight, sized for the largest chunk */
      const long long np_max = chunk_tiles * per_tile;
      /* beside the entries the bounce rays need, every wave of the launch may leave one reserved range partly unused */
      const long long launch_waves = (np_max + RT_BLOCK - 1) / RT_BLOCK < s->trace_blocks ? (np_max + RT_BLOCK - 1) / RT_BLOCK * (RT_BLOCK / 64)
                                                                                           : (long long)s->trace_blocks * (RT_BLOCK / 64);
      long long cap_ll = (long long)((double)np_max * F) + (max_rec > 1 ? launch_waves * (long long)s->reserve : 0) + 64;
      if (cap_ll >= 0x7fffffffLL) { rt_set_error("ray queue too large"); return RT_ERR_TOO_LARGE; }       /* entry numbers keep their top bit free */
      const unsigned cap = (unsigned)cap_ll;
      size_t used = 0;
      auto carve = [&](size_t bytes) { size_t o = used; used += align_up(bytes ? bytes : 1, 256); return o; };
      const size_t o_queue = carve(sizeof(QRay) * (size_t)cap), o_hits = carve(sizeof(Hit) * (size_t)cap);
      const size_t o_recs = carve(sizeof(SpawnRec) * (size_t)cap), o_totals = carve(sizeof(float) * 3 * (size_t)cap);
      const size_t o_points = carve(sizeof(float4) * (size_t)cap);
      const size_t o_rgb = carve(sizeof(float) * 3 * (size_t)np_max);
      const size_t o_dur = carve(sizeof(unsigned) * (size_t)cap);
      /* one bit per (queue entry, light): isInShadow, set by the feelers that found an occluder */
      const unsigned flag_stride = (unsigned)((L > 0 ? L : 1) + 31) / 32u;
      const size_t flag_bytes = sizeof(unsigned) * (size_t)flag_stride * (size_t)cap;
      const size_t o_flags = carve(flag_bytes);
      /* per tile of a chunk: did a primary of the tile spawn a bounce ray?  Tiles where none did are shaded in k_resolve */
      const bool fuse = view == VIEW_FRAME && s->fuse_resolve;
      const size_t o_bounce = carve(sizeof(unsigned) * 2 * (size_t)chunk_tiles);      /* and, behind those, the same per tile for "hit anything" */
      const size_t slice = used;
      last_hits_offset = n_chunks == 1 ? (long long)o_hits : -1;
      if (slice * (size_t)ring > s->arena.cap) { int rc = arena_reserve(s->arena, slice * (size_t)ring); if (rc != RT_OK) return rc; }
      { int rc = ensure_lists(s); if (rc != RT_OK) return rc; }
#define AT(T, r, off) ((T *)(s->arena.base + (size_t)(r) * slice + (off)))

      /* tile start order: only for a frame that is one launch, and from a frame of the same geometry */
      const long long key[6] = {p->width, p->height, p->sample_count, tile_first, tile_stride, n_sel};
      const bool track_cost = s->use_tile_order && n_chunks == 1 && view != VIEW_PRIMARY && per_tile % 64 == 0;
      if (track_cost && n_sel > s->order_cap) {
         if (s->d_tile_cost) (void)hipFree(s->d_tile_cost);
         if (s->d_tile_order) (void)hipFree(s->d_tile_order);
         if (s->h_tile_cost && !s->tiles_in_block) (void)hipHostFree(s->h_tile_cost);
         if (s->h_tile_order && !s->tiles_in_block) (void)hipHostFree(s->h_tile_order);
         s->d_tile_cost = s->h_tile_cost = nullptr; s->d_tile_order = s->h_tile_order = nullptr; s->order_cap = 0; s->order_valid = false; s->order_dirty = true;
         s->tiles_in_block = false;
         const long long room = n_sel <= RT_PINNED_TILES ? RT_PINNED_TILES : n_sel;
         const size_t bytes = sizeof(unsigned) * (size_t)room;
         /* two cost words per tile: its longest chain and the sum of its chains */
         HIP_TRY(hipMalloc((void **)&s->d_tile_cost, 2 * bytes)); HIP_TRY(hipMalloc((void **)&s->d_tile_order, bytes));
         if (room == RT_PINNED_TILES) {
            /* (the usual frame: the room behind the statistics, pinned with them when the scene was made) */
            s->h_tile_cost = (unsigned *)((char *)s->h_stats + RT_STATS_BYTES_PADDED);
            s->h_tile_order = (int *)(s->h_tile_cost + 2 * (size_t)RT_PINNED_TILES);
            s->tiles_in_block = true;
         } else {
            HIP_TRY(hipHostMalloc((void **)&s->h_tile_cost, 2 * bytes, hipHostMallocDefault));
            HIP_TRY(hipHostMalloc((void **)&s->h_tile_order, bytes, hipHostMallocDefault));
         }
         s->order_cap = room;
      }
      bool have_order = track_cost && s->order_valid && memcmp(key, s->order_key, sizeof key) == 0;
      /* Tiles no ray can meet anything through are left out of the launch: the start order lists the others first, and
       * only those are handed out.  (Needs what the start order needs, and the per-tile "hit anything" flags that keep
       * the kernels behind the traversal from reading such a tile.) */
      bool cull = false;
      if (track_cost && fuse && s->cull_enabled && sc.n_planes == 0 && s->bounds_ok) {
         const float camf[12] = {cam.pos.x, cam.pos.y, cam.pos.z, cam.up.x, cam.up.y, cam.up.z, cam.right.x, cam.right.y, cam.right.z,
                                 cam.look_dir.x, cam.look_dir.y, cam.look_dir.z};
         /* A new camera or frame geometry: the quick test against the scene's box now; the finer one, down the tree to
          * the primitives' boxes (a few tenths of a millisecond of host time for a thousand tiles), when the next frame
          * comes with the same camera -- a camera that moves every frame never waits for it. */
         if (s->n_live < 0 || memcmp(key, s->cull_key, sizeof key) != 0 || memcmp(camf, s->cull_cam, sizeof camf) != 0) {
            cull_tiles(s, cam, g, n_sel, false, s->tile_live, &s->n_live);
            memcpy(s->cull_key, key, sizeof key); memcpy(s->cull_cam, camf, sizeof camf);
            s->cull_stage = 1;
         } else if (s->cull_stage == 1) {
            if (s->cull_fine > 0 && s->n_live > 0) cull_tiles(s, cam, g, n_sel, true, s->tile_live, &s->n_live);
            s->cull_stage = 2;
         }
         cull = s->n_live < n_sel;
      }
      culled_frame = cull;
      /* A frame without a start order yet (the first of its geometry) starts its tiles from the middle of the scene's
       * picture outwards: the tiles the scene's box projects onto are the likely costly ones, and a launch that keeps
       * them for its end finishes on their long chains (bunny headline, first frame: k_trace 1.5 -> 1.2 ms). */
      bool first_order = false;
      if (track_cost && !have_order && s->bounds_ok) {
         first_order_guess(s, cam, g, n_sel, s->h_tile_order);
         s->order_valid = false; s->order_dirty = true; first_order = true;
      }
      if (cull) {
         /* live tiles first, in the order they have */
         if (!have_order && !first_order) { for (long long k = 0; k < n_sel; k++) s->h_tile_order[k] = (int)k; s->order_valid = false; s->order_dirty = true; }
         std::vector<int> parted((size_t)n_sel);
         size_t at = 0;
         for (long long k = 0; k < n_sel; k++) if (s->tile_live[(size_t)s->h_tile_order[k]]) parted[at++] = s->h_tile_order[k];
         for (long long k = 0; k < n_sel; k++) if (!s->tile_live[(size_t)s->h_tile_order[k]]) parted[at++] = s->h_tile_order[k];
         if (memcmp(parted.data(), s->h_tile_order, sizeof(int) * (size_t)n_sel) != 0) {
            memcpy(s->h_tile_order, parted.data(), sizeof(int) * (size_t)n_sel);
            s->order_dirty = true;
         }
      }

      /* the frame is timed on the caller's stream; the work runs there too unless the scene uses a stream of its own */
      EventPool ev(s->timing_events);
      Timer &t_total = ev.begin(stream, &st.ms_total);
      if (M != stream) {
         HIP_TRY(hipEventRecord(s->ev_begin, stream));
         HIP_TRY(hipStreamWaitEvent(M, s->ev_begin, 0));
      }

      for (long long c = 0; c < n_chunks; c++) {
         const int r = (int)(c % ring);
         hipStream_t S = M;
         const long long k0 = c * chunk_tiles;
         const long long nb = n_sel - k0 < chunk_tiles ? n_sel - k0 : chunk_tiles;
         const long long np = nb * per_tile;
         g.k0 = (int)k0;
         ChunkCtr ctr = chunk_ctr(s, r);
         QRay *queue = AT(QRay, r, o_queue);
         Hit *hits = AT(Hit, r, o_hits);
         SpawnRec *recs = AT(SpawnRec, r, o_recs);

         {
            /* counters to zero, spawned rays appended behind the primaries, tile costs and (first chunk) statistics cleared */
            const int n_stats = c == 0 ? ST_COUNT * RT_STAT_REPL : 0, n_cost = track_cost ? 2 * (int)n_sel : 0;
            int n_init = n_stats > n_cost ? (n_stats > RT_CTR_WORDS ? n_stats : RT_CTR_WORDS) : (n_cost > RT_CTR_WORDS ? n_cost : RT_CTR_WORDS);
            if (fuse && 2 * (int)nb > n_init) n_init = 2 * (int)nb;
            hipLaunchKernelGGL(k_begin_chunk, dim3(grid_for(n_init)), dim3(RT_BLOCK), 0, S, c == 0 ? s->d_stats : (unsigned long long *)nullptr,
                               n_stats, s->d_ctr, ctr.tail, (unsigned)np, track_cost ? s->d_tile_cost : (unsigned *)nullptr, n_cost,
                               fuse ? AT(unsigned, r, o_bounce) : (unsigned *)nullptr, 2 * (int)nb);
            TraceArgs a; memset(&a, 0, sizeof a);
            a.rays = queue; a.n = a.n_fetch = (unsigned long long)np; a.hits = hits; a.next = ctr.next_chain;
            a.flag_words = AT(unsigned, r, o_flags); a.flag_stride = flag_stride;
            a.recs = recs; a.tail = ctr.tail; a.cap = cap; a.max_rec = max_rec; a.cam = cam; a.g = g;
            a.points = AT(float4, r, o_points);
            a.lists = s->d_lists; a.list_cap = s->list_cap;
            if (!a.lists || a.list_cap == 0 || (a.list_cap & (a.list_cap - 1)) != 0) { rt_set_error("internal: spawn lists not set up"); return RT_ERR_HIP; }
            if (cull) a.n_fetch = (unsigned long long)(s->n_live * per_tile);
            /* (worked out per shaded point and light by the few lanes of a spawn stage: with 256 lights that costs more than the
             * 6 % of the 1M-triangle scene's feelers it stops are worth, 247 against 230 ms at 1080p 4 spp) */
            a.skip_unlit = (s->skip_unlit && p->shading == RT_PHONG && L >= 1 && L <= s->skip_unlit && s->lights_finite) ? 1 : 0;
            if (have_order || cull || first_order) {
               if (s->order_dirty) {
                  HIP_TRY(hipMemcpyAsync(s->d_tile_order, s->h_tile_order, sizeof(int) * (size_t)n_sel, hipMemcpyHostToDevice, S));
                  s->order_dirty = false;
               }
               a.tile_order = s->d_tile_order;
            }
            if (track_cost) { a.dur = AT(unsigned, r, o_dur); a.tile_cost = s->d_tile_cost; }
            if (fuse) { a.tile_bounce = AT(unsigned, r, o_bounce); a.tile_hit = a.tile_bounce + nb; }
            Timer &tm = ev.begin(S, &st.ms_trace);
            rt_launch_trace(s, OUT_CHAIN, count, a, a.n_fetch, lds_bytes, S, filter_kinds);
            ev.end(tm, S); st.launches_trace++;
         }
         if (view == VIEW_PRIMARY) {
            hipLaunchKernelGGL(k_unslot_hits, dim3(grid_for(np)), dim3(RT_BLOCK), 0, S, g, (const Hit *)hits, np, sc.prim_ids, sc.n_prims, prim_dev, t_dev);
            continue;
         }
         /* A scene without a reflecting or refracting material spawns no bounce ray: every tile is k_resolve's then, and
          * the two kernels before it have nothing to do. */
         if (!(fuse && !s->can_bounce)) {
            Timer &tm = ev.begin(S, &st.ms_light, count);
            hipLaunchKernelGGL(k_light, dim3((unsigned)nb + grid_for((long long)cap - np)), dim3(RT_BLOCK), 0, S, sc, (const QRay *)queue, (const Hit *)hits,
                               (const unsigned *)AT(unsigned, r, o_flags), flag_stride, (const unsigned *)ctr.tail, cap,
                               p->shading, AT(float, r, o_totals), fuse ? (const unsigned *)AT(unsigned, r, o_bounce) : (const unsigned *)nullptr,
                               (unsigned)np, (unsigned)per_tile);
            ev.end(tm, S);
            Timer &tc = ev.begin(S, &st.ms_combine, count);
            hipLaunchKernelGGL(k_combine_tree, dim3((unsigned)nb), dim3(RT_BLOCK), 0, S, (const Hit *)hits, (const SpawnRec *)recs,
                               (const float *)AT(float, r, o_totals), track_cost ? (const unsigned *)AT(unsigned, r, o_dur) : (const unsigned *)nullptr, np, AT(float, r, o_rgb),
                               track_cost ? s->d_tile_cost : (unsigned *)nullptr, (unsigned)per_tile,
                               fuse ? (const unsigned *)AT(unsigned, r, o_bounce) : (const unsigned *)nullptr);
            ev.end(tc, S);
         }
         if (view == VIEW_FRAME) {
            Timer &tr = ev.begin(S, &st.ms_resolve, count);
            long long npx = nb * RT_TILE * RT_TILE;
            ResolveShade rs; memset(&rs, 0, sizeof rs);
            rs.sc = sc; rs.queue = queue; rs.hits = hits; rs.recs = recs; rs.flag_words = AT(unsigned, r, o_flags); rs.flag_stride = flag_stride;
            rs.shading = p->shading; rs.tile_bounce = fuse ? AT(unsigned, r, o_bounce) : nullptr; rs.tile_hit = fuse ? AT(unsigned, r, o_bounce) + nb : nullptr;
            rs.dur = track_cost ? AT(unsigned, r, o_dur) : nullptr; rs.tile_cost = track_cost ? s->d_tile_cost : nullptr;
            hipLaunchKernelGGL(k_resolve, dim3(grid_for(npx)), dim3(RT_BLOCK), 0, S, g, (const float *)AT(float, r, o_rgb), (int)nb,
                               p->out_layout, (uchar4 *)out_dev, rs);
            ev.end(tr, S);
         } else {
            hipLaunchKernelGGL(k_unslot_rgb, dim3(grid_for(np)), dim3(RT_BLOCK), 0, S, g, (const float *)AT(float, r, o_rgb), np, rgb_dev);
         }
         /* the tile costs are complete once the last kernel that adds to them has run */
         if (track_cost && view != VIEW_PRIMARY) HIP_TRY(hipMemcpyAsync(s->h_tile_cost, s->d_tile_cost, 2 * sizeof(unsigned) * (size_t)n_sel, hipMemcpyDeviceToHost, S));
         HIP_TRY(hipGetLastError());
      }
#undef AT
      HIP_TRY(hipMemcpyAsync(s->h_stats, s->d_stats, ST_COUNT * RT_STAT_REPL * sizeof(unsigned long long), hipMemcpyDeviceToHost, M));
      if (M != stream) {
         HIP_TRY(hipEventRecord(s->ev_end, M));
         HIP_TRY(hipStreamWaitEvent(stream, s->ev_end, 0));
      }
      ev.end(t_total, stream);
#ifdef RT_DEBUG_HOST
      dbg_t1 = std::chrono::steady_clock::now();
#endif
      HIP_TRY(hipStreamSynchronize(stream));
      if (M != stream) HIP_TRY(hipStreamSynchronize(M));
#ifdef RT_DEBUG_HOST
      dbg_t2 = std::chrono::steady_clock::now();
#endif
      HIP_TRY(hipGetLastError());
      ev.collect();
      if (sum_stat(s->h_stats, ST_ABORTED) != 0) { rt_set_error("traversal aborted: step cap reached (corrupt scene?)"); return RT_ERR_HIP; }
      /* queue entries per primary the frame wanted (its chunks taken together) */
      const double all_slots = (double)n_sel * (double)per_tile;
      double wanted = (all_slots + (double)sum_stat(s->h_stats, ST_REFLECT) + (double)sum_stat(s->h_stats, ST_REFRACT)) / all_slots;
      if (sum_stat(s->h_stats, ST_LIST_OVERFLOW) != 0) {
         /* a wave's list of spawned rays was full and some were dropped: longer lists, and the frame again */
         if (attempt >= 8 || s->list_cap >= RT_LIST_CAP_MAX) { rt_set_error("spawn list overflow persists at %u entries per wave", s->list_cap); return RT_ERR_TOO_LARGE; }
         s->list_cap *= 2; s->list_waves = 0;
         continue;
      }
      if (sum_stat(s->h_stats, ST_OVERFLOW) != 0) {
         if (attempt >= 8) { rt_set_error("ray queue overflow persists"); return RT_ERR_TOO_LARGE; }
         double grown = wanted * 1.25 > 2.0 * F ? wanted * 1.25 : 2.0 * F;
         s->node_factor = (float)(grown < 64.0 ? grown : 64.0);
         continue;                              /* some bounce rays were dropped: render the frame again */
      }
      if (max_rec > 1 && wanted * 1.1 > F) s->node_factor = (float)(wanted * 1.25 < 64.0 ? wanted * 1.25 : 64.0);
      /* (the costs are the rays' own step counts: a still camera brings the same ones back every frame, and then the
       * order stands as it is) */
      /* (nearly the same ones: a guided walk's first steps depend on where in its wave's burst it began, so a few costs
       * move by a few steps from frame to frame.  The order stands while no tile's cost has moved by more than a
       * sixteenth, or eight steps, from the costs the order was made from; ranking 2 000 tiles anew costs the host 50 us, a
       * twentieth of the headline frame.) */
      bool same_costs = track_cost && s->order_valid && memcmp(key, s->order_key, sizeof key) == 0 && !s->last_cost.empty() &&
                        s->last_cost.size() == 2 * (size_t)n_sel;
      for (long long k = 0; k < n_sel && same_costs; k++)
         for (int word = 0; word < (s->order_mean_weight > 0.0 ? 2 : 1); word++) {
            const unsigned was = s->last_cost[2 * (size_t)k + word], is = s->h_tile_cost[2 * k + word];
            const unsigned moved = was > is ? was - is : is - was;
            if (moved > 8u && moved > was / 16u) same_costs = false;
         }
      if (track_cost && !same_costs) {
         s->last_cost.assign(s->h_tile_cost, s->h_tile_cost + 2 * (size_t)n_sel);
         /* Costliest first, in a few classes of equal size (by rank of the tile's longest chain); tiles of a class stay
          * in image order, so neighbouring waves still work on neighbouring tiles: a scattered order costs more in the
          * caches than the shorter tail wins. */
         const int classes = s->order_classes;
         /* a tile's cost: its longest chain, plus order_mean_weight times its mean chain (a tile full of long chains
          * keeps its waves busy longer than one with a single long chain) */
         std::vector<unsigned> costs((size_t)n_sel);
         for (long long k = 0; k < n_sel; k++)
            costs[(size_t)k] = s->h_tile_cost[2 * k] + (unsigned)((double)s->h_tile_cost[2 * k + 1] * s->order_mean_weight / (double)per_tile);
         std::vector<unsigned> sorted(costs);
         std::sort(sorted.begin(), sorted.end(), std::greater<unsigned>());
         std::vector<unsigned> bound((size_t)classes);        /* class c holds costs above bound[c] (and not in an earlier class) */
         for (int c = 0; c < classes; c++) { long long at = (long long)(c + 1) * n_sel / classes; bound[(size_t)c] = at < n_sel ? sorted[(size_t)at] : 0u; }
         /* bound[] falls with the class number: the class of a cost is the number of bounds it does not exceed */
         auto cls = [&](unsigned cost) {
            int lo_c = 0, hi_c = classes - 1;           /* the first c in [0, classes - 1) with cost > bound[c], else classes - 1 */
            while (lo_c < hi_c) { int mid = (lo_c + hi_c) / 2; if (cost > bound[(size_t)mid]) hi_c = mid; else lo_c = mid + 1; }
            return lo_c;
         };
         std::vector<long long> start((size_t)classes + 1, 0);
         for (long long k = 0; k < n_sel; k++) start[(size_t)cls(costs[(size_t)k]) + 1]++;
         for (int c = 0; c < classes; c++) start[(size_t)c + 1] += start[(size_t)c];
         std::vector<int> order((size_t)n_sel);
         for (long long k = 0; k < n_sel; k++) order[(size_t)start[(size_t)cls(costs[(size_t)k])]++] = (int)k;
         /* tiles that are left out of the launch come last (they cost nothing, but so do traced tiles of background) */
         if (culled_frame) std::stable_partition(order.begin(), order.end(), [&](int k) { return s->tile_live[(size_t)k] != 0; });
         const bool same = s->order_valid && memcmp(key, s->order_key, sizeof key) == 0 &&
                           memcmp(order.data(), s->h_tile_order, sizeof(int) * (size_t)n_sel) == 0;
         if (!same) {         /* an unchanged order (a still camera) is not uploaded again */
            memcpy(s->h_tile_order, order.data(), sizeof(int) * (size_t)n_sel);
            s->order_dirty = true;
         }
         memcpy(s->order_key, key, sizeof key);
         s->order_valid = true;
      }
      break;
   }
   /* The primaries are the selected tiles' samples that lie inside the frame; every shaded point issues one feeler per
    * light, traced unless its own-surface test missed: neither needs a device counter. */
   {
      unsigned long long inside = 0;
      const int tiles_x = (p->width + RT_TILE - 1) / RT_TILE;
      for (long long k = 0; k < n_sel; k++) {
         long long tile = tile_first + k * tile_stride;
         long long x0 = (tile % tiles_x) * RT_TILE, y0 = (tile / tiles_x) * RT_TILE;
         long long w = p->width - x0 < RT_TILE ? p->width - x0 : RT_TILE, h = p->height - y0 < RT_TILE ? p->height - y0 : RT_TILE;
         if (w > 0 && h > 0) {
            inside += (unsigned long long)(w * h) * (unsigned long long)p->sample_count;
            if (culled_frame && !s->tile_live[(size_t)k]) st.primaries_culled += (unsigned long long)(w * h) * (unsigned long long)p->sample_count;
         }
      }
      st.rays_primary = inside;
   }
   st.shadow_skipped = sum_stat(s->h_stats, ST_SHADOW_SKIPPED);
   st.rays_reflect = sum_stat(s->h_stats, ST_REFLECT);
   st.rays_refract = sum_stat(s->h_stats, ST_REFRACT);
   st.shade_nodes = sum_stat(s->h_stats, ST_SHADE_NODES);
   st.rays_shadow = st.shade_nodes * (unsigned long long)(L > 0 ? L : 0) - st.shadow_skipped;
   st.rays_total = st.rays_primary + st.rays_shadow + st.rays_reflect + st.rays_refract;
   st.samples = st.rays_primary;
   st.node_visits = sum_stat(s->h_stats, ST_NODE_VISITS); st.prim_tests = sum_stat(s->h_stats, ST_PRIM_TESTS); st.tie_events = sum_stat(s->h_stats, ST_TIES);
   st.node_visits_shadow = sum_stat(s->h_stats, ST_NODE_VISITS_SHADOW); st.prim_tests_shadow = sum_stat(s->h_stats, ST_PRIM_TESTS_SHADOW);
   st.tie_events_shadow = sum_stat(s->h_stats, ST_TIES_SHADOW);
   st.wave_node_steps = sum_stat(s->h_stats, ST_WAVE_NODE_STEPS); st.wave_leaf_steps = sum_stat(s->h_stats, ST_WAVE_LEAF_STEPS);
   st.lane_steps_at_leaf = sum_stat(s->h_stats, ST_LEAF_WAIT); st.lane_steps_no_ray = sum_stat(s->h_stats, ST_NO_RAY);
   st.wave_refills = sum_stat(s->h_stats, ST_REFILLS);
   st.hits_certified = sum_stat(s->h_stats, ST_CERT_OK); st.hits_contested = sum_stat(s->h_stats, ST_CERT_RIVAL); st.hits_guided = sum_stat(s->h_stats, ST_GUIDED);
   st.wave_steps_both_trees = sum_stat(s->h_stats, ST_STEPS_BOTH); st.wave_steps_second_tree_only = sum_stat(s->h_stats, ST_STEPS_SECOND_ONLY);
   st.hits_unreached = sum_stat(s->h_stats, ST_CERT_UNREACHED);
   st.shadows_certified = sum_stat(s->h_stats, ST_SHADOW_CERT);
   st.shadows_unneeded = sum_stat(s->h_stats, ST_SHADOW_UNNEEDED);
   {
      double w = (double)sum_stat(s->h_stats, ST_CLK_WAVE);
      st.wave_time_refill = w > 0 ? (float)((double)sum_stat(s->h_stats, ST_CLK_REFILL) / w) : 0.0f;
      st.wave_time_node = w > 0 ? (float)((double)sum_stat(s->h_stats, ST_CLK_NODE) / w) : 0.0f;
      st.wave_time_leaf = w > 0 ? (float)((double)sum_stat(s->h_stats, ST_CLK_LEAF) / w) : 0.0f;
   }
   if (probing && s->probe_step < 5) {
      if (s->probe_step >= 1) s->probe_ms[filter_kinds] = st.ms_trace;
      if (++s->probe_step == 5) {
         /* the quickest stays; anything but "both" has to win by more than the frame-to-frame noise */
         int best = 3;
         for (int k = 0; k < 3; k++) if (s->probe_ms[k] < 0.96f * s->probe_ms[3] && s->probe_ms[k] < s->probe_ms[best]) best = k;
         s->filter_choice = best;
      }
   }
   if (view == VIEW_FRAME && last_hits_offset >= 0) {
      g.k0 = 0;
      s->last_frame.g = g; s->last_frame.np = (long long)n_sel * per_tile; s->last_frame.hits_offset = last_hits_offset;
      s->last_frame.culled = culled_frame; s->last_frame.n_sel = n_sel; s->last_frame.valid = true;
   }
   st.second_tree = s->d.filter_root != RT_ROOT_EMPTY ? filter_kinds : 0;      /* (a scene without the tree: none, whatever the choice would be) */
   st.max_ray_steps = s->h_stats[ST_MAX_RAY_STEPS * RT_STAT_REPL];
   st.max_spawn_list = s->h_stats[ST_MAX_LIST * RT_STAT_REPL];
   st.attempts = attempts;
#ifdef RT_DEBUG_EXIT_HIST
   fprintf(stderr, "wave exits per 0.25 ms:");
   for (int b = 0; b < 48; b++) fprintf(stderr, " %llu", (unsigned long long)sum_stat(s->h_stats, ST_HIST + b));
   fprintf(stderr, "\n");
#endif
   if (count && sum_stat(s->h_stats, ST_WAVES) > 0 && st.launches_trace == 1) {
      double first = (double)(~s->h_stats[ST_CLOCK_FIRST_START * RT_STAT_REPL]);
      double waves = (double)sum_stat(s->h_stats, ST_WAVES);
      st.wave_exit_mean_ms = (float)(((double)sum_stat(s->h_stats, ST_CLOCK_EXIT_SUM) / waves - first) * 1e-5);
      st.wave_exit_last_ms = (float)(((double)s->h_stats[ST_CLOCK_LAST_EXIT * RT_STAT_REPL] - first) * 1e-5);
   }
   if (stats_out) *stats_out = st;
#ifdef RT_DEBUG_HOST
   {
      auto t3 = std::chrono::steady_clock::now();
      auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      if (dbg_n > 0) dbg_acc[3] += us(dbg_prev_exit, dbg_t0);
      dbg_acc[0] += us(dbg_t0, dbg_t1); dbg_acc[1] += us(dbg_t1, dbg_t2); dbg_acc[2] += us(dbg_t2, t3); dbg_n++;
      dbg_prev_exit = t3;
      if (dbg_n % 10 == 0) { fprintf(stderr, "host us per frame: enqueue %.1f, wait %.1f, after %.1f, outside the call %.1f\n", dbg_acc[0] / 10, dbg_acc[1] / 10, dbg_acc[2] / 10, dbg_acc[3] / 10); dbg_acc[0] = dbg_acc[1] = dbg_acc[2] = dbg_acc[3] = 0; }
   }
#endif
   return RT_OK;
}

extern "C" int rt_render_device(rt_scene *scene, const rt_render_params *params, void *out_rgba_dev, void *stream,
                                rt_stats *stats)
{
   if (!out_rgba_dev) { rt_set_error("rt_render_device: null output"); return RT_ERR_BAD_ARGUMENT; }
   return render_impl(scene, params, VIEW_FRAME, out_rgba_dev, nullptr, nullptr, nullptr, (hipStream_t)stream, stats);
}

extern "C" int rt_render(rt_scene *scene, const rt_render_params *params, uint8_t *out_host, rt_stats *stats)
{
   if (!scene || !params || !out_host) { rt_set_error("rt_render: null argument"); return RT_ERR_BAD_ARGUMENT; }
   if (params->width <= 0 || params->height <= 0) { rt_set_error("rt_render: non-positive size"); return RT_ERR_BAD_ARGUMENT; }
   if (scene->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   HIP_TRY(hipSetDevice(scene->device));
   rt_render_params p = *params;
   p.out_layout = RT_OUT_FRAME;
   size_t bytes = (size_t)p.width * (size_t)p.height * 4;
   rt_scene *s = scene;
   if (bytes > s->frame_cap) {
      if (s->d_frame) { (void)hipFree(s->d_frame); s->d_frame = nullptr; s->frame_cap = 0; }
      HIP_TRY(hipMalloc(&s->d_frame, bytes));
      s->frame_cap = bytes;
   }
   void *d_out = s->d_frame;
   /* pixels of tiles this call does not select read as 0 */
   const bool all_tiles = p.tile_first == 0 && p.tile_stride <= 1;
   hipError_t e = all_tiles ? hipSuccess : hipMemset(d_out, 0, bytes);
   int rc = e == hipSuccess ? render_impl(scene, &p, VIEW_FRAME, d_out, nullptr, nullptr, nullptr, nullptr, stats) : RT_ERR_HIP;
   if (rc == RT_OK) {
      /* kernel.cu:759-760.  A copy into pageable memory goes through the runtime's own staging at about half the
       * bus's rate (a 1080p frame: 0.45 ms); a scene that is asked for a second frame gets a pinned buffer for it
       * (a one-frame scene would pay more for the buffer than it wins). */
      s->host_frames++;
      if (s->host_frames >= 2 && bytes > s->h_frame_cap && bytes <= ((size_t)256 << 20) && rt_env_int("RT_PINNED_FRAME", 1, true)) {
         if (s->h_frame) { (void)hipHostFree(s->h_frame); s->h_frame = nullptr; s->h_frame_cap = 0; }
         if (hipHostMalloc(&s->h_frame, bytes, hipHostMallocDefault) == hipSuccess) s->h_frame_cap = bytes;
         else { s->h_frame = nullptr; (void)hipGetLastError(); }      /* (no pinned memory to be had: the plain copy, then) */
         if (s->h_frame && !s->copy_ev_ok) {
            bool ok = true;
            for (int k = 0; k < 4 && ok; k++) {
               ok = hipEventCreateWithFlags(&s->copy_ev[k], hipEventDisableTiming) == hipSuccess;
               if (!ok) for (int j = 0; j < k; j++) (void)hipEventDestroy(s->copy_ev[j]);
            }
            s->copy_ev_ok = ok;
            if (!ok) { (void)hipGetLastError(); (void)hipHostFree(s->h_frame); s->h_frame = nullptr; s->h_frame_cap = 0; }
         }
      }
      if (s->h_frame && s->h_frame_cap >= bytes && s->copy_ev_ok) {
         const int parts = bytes >= ((size_t)1 << 20) ? 4 : 1;
         const size_t part = ((bytes + (size_t)parts - 1) / (size_t)parts + 4095) & ~(size_t)4095;
         int issued = 0;
         for (int k = 0; k < parts && e == hipSuccess; k++) {
            const size_t off = (size_t)k * part;
            if (off >= bytes) break;
            const size_t len = bytes - off < part ? bytes - off : part;
            e = hipMemcpyAsync((char *)s->h_frame + off, (const char *)d_out + off, len, hipMemcpyDeviceToHost, nullptr);
            if (e == hipSuccess) e = hipEventRecord(s->copy_ev[k], nullptr);
            if (e == hipSuccess) issued++;
         }
         for (int k = 0; k < issued; k++) {
            const size_t off = (size_t)k * part;
            const size_t len = bytes - off < part ? bytes - off : part;
            hipError_t w = hipEventSynchronize(s->copy_ev[k]);
            if (w != hipSuccess) { if (e == hipSuccess) e = w; continue; }      /* (every part issued is waited for, whatever happens) */
            if (e == hipSuccess) memcpy(out_host + off, (const char *)s->h_frame + off, len);
         }
      } else e = hipMemcpy(out_host, d_out, bytes, hipMemcpyDeviceToHost);
      if (e != hipSuccess) { rt_set_error("frame copy failed: %s", hipGetErrorString(e)); rc = RT_ERR_HIP; }
   }
   return rc;
}

extern "C" int rt_untile_device(const void *tiles_dev, size_t part_stride_bytes, int n_parts, int width, int height,
                                void *frame_dev, void *stream)
{
   if (!tiles_dev || !frame_dev || n_parts <= 0 || width <= 0 || height <= 0) { rt_set_error("rt_untile_device: bad argument"); return RT_ERR_BAD_ARGUMENT; }
   long long n = (long long)width * height;
   hipLaunchKernelGGL(k_untile, dim3(grid_for(n)), dim3(RT_BLOCK), 0, (hipStream_t)stream, (const uchar4 *)tiles_dev,
                      (long long)(part_stride_bytes / 4), n_parts, width, height, (width + RT_TILE - 1) / RT_TILE, (uchar4 *)frame_dev);
   HIP_TRY(hipGetLastError());
   return RT_OK;
}

extern "C" int rt_trace_primary(rt_scene *scene, const rt_render_params *params, int32_t *prim_host, float *t_host)
{
   if (!scene || !params || !prim_host || !t_host) { rt_set_error("rt_trace_primary: null argument"); return RT_ERR_BAD_ARGUMENT; }
   int sroot = isqrt_trunc(params->sample_count);
   if (sroot <= 0 || sroot * sroot != params->sample_count) { rt_set_error("Invalid sample count: %d", params->sample_count); return RT_ERR_BAD_SAMPLE_COUNT; }
   if (scene->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   HIP_TRY(hipSetDevice(scene->device));
   size_t n = (size_t)params->width * sroot * (size_t)params->height * sroot;
   int *d_prim = nullptr; float *d_t = nullptr;
   HIP_TRY(hipMalloc((void **)&d_prim, n * 4));
   if (hipMalloc((void **)&d_t, n * 4) != hipSuccess) { (void)hipFree(d_prim); rt_set_error("hipMalloc failed"); return RT_ERR_HIP; }
   (void)hipMemset(d_prim, 0xff, n * 4); (void)hipMemset(d_t, 0, n * 4);
   rt_render_params p = *params; p.tile_first = 0; p.tile_stride = 1;
   int rc = render_impl(scene, &p, VIEW_PRIMARY, nullptr, d_prim, d_t, nullptr, nullptr, nullptr);
   if (rc == RT_OK) {
      if (hipMemcpy(prim_host, d_prim, n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(t_host, d_t, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { rt_set_error("copy back failed"); rc = RT_ERR_HIP; }
   }
   (void)hipFree(d_prim); (void)hipFree(d_t);
   return rc;
}

extern "C" int rt_frame_primary_hits(const rt_scene *scene, int32_t *prim_host, float *t_host, int64_t n)
{
   if (!scene || !prim_host || !t_host) { rt_set_error("rt_frame_primary_hits: null argument"); return RT_ERR_BAD_ARGUMENT; }
   if (scene->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   const rt_scene::LastFrame &f = scene->last_frame;
   if (!f.valid) { rt_set_error("rt_frame_primary_hits: the scene's last call was not a frame rendered in one launch"); return RT_ERR_BAD_ARGUMENT; }
   const DGrid &g = f.g;
   const size_t ns = (size_t)g.res_w * (size_t)g.res_h;
   if (n != (int64_t)ns) { rt_set_error("rt_frame_primary_hits: room for %lld samples, the frame has %zu", (long long)n, ns); return RT_ERR_BAD_ARGUMENT; }
   HIP_TRY(hipSetDevice(scene->device));
   int *d_prim = nullptr; float *d_t = nullptr;
   HIP_TRY(hipMalloc((void **)&d_prim, ns * 4));
   if (hipMalloc((void **)&d_t, ns * 4) != hipSuccess) { (void)hipFree(d_prim); rt_set_error("hipMalloc failed"); return RT_ERR_HIP; }
   (void)hipMemset(d_prim, 0xff, ns * 4); (void)hipMemset(d_t, 0, ns * 4);
   const Hit *hits = (const Hit *)(scene->arena.base + f.hits_offset);
   hipLaunchKernelGGL(k_unslot_hits, dim3(grid_for(f.np)), dim3(RT_BLOCK), 0, nullptr, g, hits, f.np, scene->d.prim_ids, scene->d.n_prims, d_prim, d_t);
   int rc = RT_OK;
   if (hipMemcpy(prim_host, d_prim, ns * 4, hipMemcpyDeviceToHost) != hipSuccess ||
       hipMemcpy(t_host, d_t, ns * 4, hipMemcpyDeviceToHost) != hipSuccess) { rt_set_error("copy back failed"); rc = RT_ERR_HIP; }
   (void)hipFree(d_prim); (void)hipFree(d_t);
   if (rc == RT_OK && f.culled) {
      /* a tile that was left out of the launch has no stored hits: the frame took every sample of it for a miss */
      for (int k = 0; k < f.n_sel; k++) {
         if (scene->tile_live[(size_t)k]) continue;
         const long long tile = (long long)g.tile_first + (long long)k * g.tile_stride;
         const long long x0 = (tile % g.tiles_x) * g.ts, y0 = (tile / g.tiles_x) * g.ts;
         for (long long y = y0; y < y0 + g.ts && y < g.res_h; y++)
            for (long long x = x0; x < x0 + g.ts && x < g.res_w; x++) { prim_host[y * g.res_w + x] = -1; t_host[y * g.res_w + x] = FLT_MAX; }
      }
   }
   return rc;
}

extern "C" int rt_render_samples(rt_scene *scene, const rt_render_params *params, float *rgb_host, rt_stats *stats)
{
   if (!scene || !params || !rgb_host) { rt_set_error("rt_render_samples: null argument"); return RT_ERR_BAD_ARGUMENT; }
   int sroot = isqrt_trunc(params->sample_count);
   if (sroot <= 0 || sroot * sroot != params->sample_count) { rt_set_error("Invalid sample count: %d", params->sample_count); return RT_ERR_BAD_SAMPLE_COUNT; }
   if (scene->device < 0) { rt_set_error("scene was created without a device"); return RT_ERR_NO_DEVICE; }
   HIP_TRY(hipSetDevice(scene->device));
   size_t n = (size_t)params->width * sroot * (size_t)params->height * sroot;
   float *d_rgb = nullptr;
   HIP_TRY(hipMalloc((void **)&d_rgb, n * 12));
   (void)hipMemset(d_rgb, 0, n * 12);
   rt_render_params p = *params; p.tile_first = 0; p.tile_stride = 1;
   int rc = render_impl(scene, &p, VIEW_SAMPLES, nullptr, nullptr, nullptr, d_rgb, nullptr, stats);
   if (rc == RT_OK && hipMemcpy(rgb_host, d_rgb, n * 12, hipMemcpyDeviceToHost) != hipSuccess) { rt_set_error("copy back failed"); rc = RT_ERR_HIP; }
   (void)hipFree(d_rgb);
   return rc;
}

