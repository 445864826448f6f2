/*
 * rt_shade.h -- the kernels behind the traversal (device code; included by rt_render.hip alone): light sums
 * (PhongShader.h, CookTorranceShader.h, getColor kernel.cu:323-332), the combination of a bounce tree
 * (kernel.cu:375-376), the box filter (averageBufferColors, kernel.cu:545-570), un-tiling and the parity views.
 */
#ifndef RT_SHADE_H
#define RT_SHADE_H

#include "rt_trace.h"

/* tex2D on the texture the reference binds (kernel.cu:32, 590-593): uchar4 texels read as normalised floats, wrap
 * addressing, linear filtering, normalised coordinates -- x = frac(u)*W - 0.5, the four texels around it, weights
 * in 9-bit fixed point with 8 fractional bits, as CUDA documents its linear filter. */
__device__ __forceinline__ F3 texture_sample(const uchar4 *texels, rt_dev_texinfo ti, float u, float v)
{
   int W = ti.width, H = ti.height;
   float x = (u - floorf(u)) * (float)W - 0.5f;
   float y = (v - floorf(v)) * (float)H - 0.5f;
   float fx = floorf(x), fy = floorf(y);
   float a = floorf((x - fx) * 256.0f + 0.5f) * 0.00390625f;
   float b = floorf((y - fy) * 256.0f + 0.5f) * 0.00390625f;
   int i0 = (int)fx, j0 = (int)fy;
   i0 = ((i0 % W) + W) % W; j0 = ((j0 % H) + H) % H;
   int i1 = (i0 + 1) % W, j1 = (j0 + 1) % H;
   const uchar4 *base = texels + ti.offset;
   uchar4 t00 = base[(size_t)j0 * W + i0], t10 = base[(size_t)j0 * W + i1];
   uchar4 t01 = base[(size_t)j1 * W + i0], t11 = base[(size_t)j1 * W + i1];
   float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
   F3 c;
   c.x = w00 * ((float)t00.x / 255.0f) + w10 * ((float)t10.x / 255.0f) + w01 * ((float)t01.x / 255.0f) + w11 * ((float)t11.x / 255.0f);
   c.y = w00 * ((float)t00.y / 255.0f) + w10 * ((float)t10.y / 255.0f) + w01 * ((float)t01.y / 255.0f) + w11 * ((float)t11.y / 255.0f);
   c.z = w00 * ((float)t00.z / 255.0f) + w10 * ((float)t10.z / 255.0f) + w01 * ((float)t01.z / 255.0f) + w11 * ((float)t11.z / 255.0f);
   return c;
}

/* getColor, kernel.cu:323-332: the pigment colour, or the texture at UVAt (Triangle.h:44-52; the other classes
 * return vec2(0): Sphere.h:34-37, Plane.h:32-35, Box.h:25-28) */
__device__ __forceinline__ F3 material_color(const DScene &sc, const rt_material &m, int prim, const Ray &ray, float t)
{
   if (m.tex_id == -1) return mk3(m.clr[0], m.clr[1], m.clr[2]);
   float u = 0.0f, v = 0.0f;
   if (prim < sc.n_prims) {
      PrimRec p = load_prim(sc.prims, prim);
      if (p.m.x == RT_PRIM_TRIANGLE || p.m.x == RT_PRIM_SMOOTH) {
         const float4 *sp = reinterpret_cast<const float4 *>(sc.shades + prim);
         float4 s2 = sp[2], s3 = sp[3];
         F3 p1 = mk3(p.a.x, p.a.y, p.a.z), p2 = mk3(p.a.w, p.b.x, p.b.y), p3 = mk3(p.b.z, p.b.w, p.c.x);
         F3 n = mk3(p.c.y, p.c.z, p.c.w);
         F3 q = ray_point(ray, t);
         float area = dot3(cross3(p2 - p1, p3 - p1), n);
         float beta = dot3(cross3(p1 - p3, q - p3), n) / area;
         float gamma = dot3(cross3(p2 - p1, q - p1), n) / area;
         float alpha = (float)(1.0 - (double)beta - (double)gamma);   /* the reference's "1.0" is a double */
         u = alpha * s2.y + beta * s2.w + gamma * s3.y;                /* vt1 = s[9..10], vt2 = s[11], uv[0], vt3 = uv[1..2] */
         v = alpha * s2.z + beta * s3.x + gamma * s3.z;
      }
   }
   return texture_sample(sc.texels, sc.tex_info[m.tex_id], u, v);
}

/* PhongShader::shade, PhongShader.h:7-25 */
__device__ __forceinline__ F3 phong(F3 matColor, float amb, float dif, float spec, float rough, F3 eye, F3 lightDir,
                                    F3 lightColor, F3 normal, bool inShadow)
{
   F3 light = mk3(0.0f, 0.0f, 0.0f);
   light = light + amb * lightColor;
   if (inShadow) return light * matColor;
   light = light + (dif * clamp01(dot3(normal, lightDir))) * lightColor;
   F3 reflect = (2.0f * dot3(lightDir, normal)) * normal - lightDir;
   light = light + (spec * powf(clamp01(dot3(reflect, eye)), 1.0f / rough)) * lightColor;
   return light * matColor;
}

/* CookTorranceShader::shade, CookTorranceShader.h:7-46 */
__device__ __forceinline__ F3 cook_torrance(F3 matColor, float amb, float dif, float spec, float rough, F3 eye,
                                            F3 lightDir, F3 lightColor, F3 normal, bool inShadow)
{
   F3 light = mk3(0.0f, 0.0f, 0.0f);
   light = light + amb * lightColor;
   if (inShadow) return light * matColor;
   light = light + (dif * clamp01(dot3(normal, lightDir))) * lightColor;
   F3 h = normalize3(lightDir + eye);
   float ior = 2.0f;
   float r0 = (1.0f - ior) * (1.0f - ior) / ((1.0f + ior) * (1.0f + ior));
   float f = r0 + (1.0f - r0) * powf(1.0f - dot3(lightDir, normal), 5.0f);
   float nDotH = dot3(normal, h);
   float mSqr = rough * rough;
   float d = expf((nDotH * nDotH - 1.0f) / (mSqr * nDotH * nDotH)) / (mSqr * nDotH * nDotH * nDotH * nDotH);
   float gStart = 2.0f * nDotH / dot3(eye, h);
   float g1 = gStart * dot3(eye, normal);
   float g2 = gStart * dot3(lightDir, normal);
   float gg = g1 < g2 ? g1 : g2;
   gg = gg < 1.0f ? gg : 1.0f;
   float kSpec = d * f * gg / (4.0f * dot3(eye, normal) * dot3(normal, lightDir));
   light = light + (spec * kSpec) * lightColor;
   return light * matColor;
}

/* what both halves of shadeObject (kernel.cu:335-382) need about a hit */
struct ShadePoint { Ray ray; F3 P, normal, eye; rt_material m; };

__device__ __forceinline__ ShadePoint shade_point(const DScene &sc, const QRay *queue, long long i, Hit h)
{
   ShadePoint sp;
   sp.ray = load_qray(queue, i);
   sp.P = ray_point(sp.ray, h.t);                               /* kernel.cu:340 */
   unsigned mat_id = h.prim < sc.n_prims ? sc.prims[h.prim].material : sc.planes[h.prim - sc.n_prims].material;
   sp.m = sc.materials[mat_id];
   sp.normal = normal_at(sc, h.prim, sp.ray, h.t);              /* kernel.cu:342 */
   sp.eye = normalize3(-sp.ray.d);                              /* kernel.cu:344 */
   return sp;
}

/* start of a chunk: its counters to zero, the queue's tail behind the primaries, and (first chunk) the statistics;
 * one launch instead of a memset each */
__global__ __launch_bounds__(RT_BLOCK) void k_begin_chunk(unsigned long long *stats, int n_stats, unsigned long long *ctr, unsigned *tail,
                                                          unsigned n_primary, unsigned *tile_cost, int n_cost, unsigned *tile_bounce, int n_bounce)
{
   int i = (int)(blockIdx.x * RT_BLOCK + threadIdx.x);
   if (stats && i < n_stats) stats[i] = 0ULL;
   if (tile_cost && i < n_cost) tile_cost[i] = 0u;
   if (tile_bounce && i < n_bounce) tile_bounce[i] = 0u;
   if (i < 8 * (int)RT_FETCH_STREAMS) ctr[i] = 0ULL;       /* the fetch counters; the tail's words follow them */
   if (i == 0) *tail = n_primary;
}

/* light: totalLight of every node, the loop of kernel.cu:347-358 with the shadow flags, over every queue entry
 * some wave reserved (entries reserved and not used read as a miss) */
__device__ __forceinline__ F3 light_sum(const DScene &sc, const QRay *queue, long long i, Hit h, const unsigned *flag_words, unsigned flag_stride,
                                        int shading)
{
   F3 total = mk3(0.0f, 0.0f, 0.0f);
   if (h.prim >= 0) {
      ShadePoint sp = shade_point(sc, queue, i, h);
      F3 matClr = material_color(sc, sp.m, h.prim, sp.ray, h.t);
      const unsigned *fw = flag_words + (size_t)i * flag_stride;
      unsigned bits = 0;
      for (int l = 0; l < sc.n_lights; l++) {
         if ((l & 31) == 0) bits = fw[l >> 5];
         const rt_dev_light &L = sc.lights[l];
         F3 lc = mk3(L.clr[0], L.clr[1], L.clr[2]);
         F3 lightDir = normalize3(mk3(L.pos[0], L.pos[1], L.pos[2]) - sp.P);    /* PointLight.h:14-16 */
         bool inShadow = ((bits >> (l & 31)) & 1u) != 0;
         F3 c = shading == RT_COOK_TORRANCE
                   ? cook_torrance(matClr, sp.m.amb, sp.m.dif, sp.m.spec, sp.m.rough, sp.eye, lightDir, lc, sp.normal, inShadow)
                   : phong(matClr, sp.m.amb, sp.m.dif, sp.m.spec, sp.m.rough, sp.eye, lightDir, lc, sp.normal, inShadow);
         total = total + c;
      }
   }
   return total;
}

/* `tile_bounce` (may be null): primaries of a tile whose flag is clear are left to k_resolve, which shades them itself
 * (a wave's 64 entries lie in one tile: per_tile is a multiple of 64 whenever the flags are in use) */
__global__ __launch_bounds__(RT_BLOCK) void k_light(DScene sc, const QRay *queue, const Hit *hits, const unsigned *flag_words, unsigned flag_stride,
                                                    const unsigned *n_dev, unsigned cap, int shading, float *totals,
                                                    const unsigned *tile_bounce, unsigned n_primary, unsigned per_tile)
{
   /* one workgroup per tile for the primaries (a tile left to k_resolve costs one look at its flag), then one per 256
    * of the entries behind them */
   const unsigned n_tiles = n_primary / per_tile;
   unsigned reserved = *n_dev;
   unsigned n = reserved < cap ? reserved : cap;
   long long first, last, step = RT_BLOCK;
   if (blockIdx.x < n_tiles) {
      if (tile_bounce && tile_bounce[blockIdx.x] == 0u) return;
      first = (long long)blockIdx.x * per_tile + threadIdx.x; last = (long long)(blockIdx.x + 1) * per_tile;
   } else {
      first = (long long)n_primary + (long long)(blockIdx.x - n_tiles) * RT_BLOCK + threadIdx.x; last = first + 1;
   }
   if (last > (long long)n) last = (long long)n;
   for (long long i = first; i < last; i += step) {
      F3 total = light_sum(sc, queue, i, hits[i], flag_words, flag_stride, shading);
      totals[3 * i] = total.x; totals[3 * i + 1] = total.y; totals[3 * i + 2] = total.z;
   }
}

/* combine: the colour of primary i from the light sums of its bounce tree, children before parents:
 *   colour(node) = total*(1 - refl - alpha) + refl*colour(reflected) + alpha*colour(refracted)      (kernel.cu:375-376)
 * with vec3(0) for a ray that was not spawned or hit nothing.  One thread walks one tree (at most
 * 2^RT_MAX_RECURSE - 1 nodes, usually a short chain) with an explicit stack of one frame per level, so every
 * node is evaluated once, by one thread, in the reference's operand order: no atomics, no races. */
__global__ __launch_bounds__(RT_BLOCK) void k_combine_tree(const Hit *hits, const SpawnRec *recs, const float *totals, const unsigned *dur, long long n_primary,
                                                           float *rgb, unsigned *tile_cost, unsigned per_tile, const unsigned *tile_bounce)
{
   /* one workgroup per tile: a tile without a bounce ray is left to k_resolve and costs one look at its flag */
   if (tile_bounce && tile_bounce[blockIdx.x] == 0u) return;
   for (long long i = (long long)blockIdx.x * per_tile + threadIdx.x; i < (long long)(blockIdx.x + 1) * per_tile; i += RT_BLOCK) {
      const bool live = i < n_primary;
      int node[RT_MAX_RECURSE], stage[RT_MAX_RECURSE];
      float rx[RT_MAX_RECURSE], ry[RT_MAX_RECURSE], rz[RT_MAX_RECURSE];
      /* a primary that hit nothing has no spawn record and no cost: its colour is the +0 the expression below gives for
       * a light sum of 0 and weights 0, 0 */
      const bool hit = live && hits[i].prim >= 0;
      int sp = hit ? 0 : -1;
      unsigned chain_refl[RT_MAX_RECURSE];
      unsigned chain = 0;          /* the longest chain of dependent rays below (and with) the node just evaluated, in traversal steps of its rays */
      node[0] = (int)i; stage[0] = 0;
      F3 ret = mk3(0.0f, 0.0f, 0.0f);
      if (hit) {
         /* most trees are a single node: the same expression with both children absent, without the stack */
         float4 rr = *reinterpret_cast<const float4 *>(recs + i);
         if (__float_as_int(rr.z) < 0 && __float_as_int(rr.w) < 0) {
            F3 total = mk3(totals[3 * i], totals[3 * i + 1], totals[3 * i + 2]);
            const F3 none = mk3(0.0f, 0.0f, 0.0f);
            ret = total * (1.0f - rr.x - rr.y) + rr.x * none + rr.y * none;
            if (dur) chain = dur[i];
            sp = -1;
         }
      }
      while (sp >= 0) {
         float4 rr = *reinterpret_cast<const float4 *>(recs + node[sp]);
         int child_refl = __float_as_int(rr.z), child_refr = __float_as_int(rr.w);
         if (stage[sp] == 0) {
            stage[sp] = 1;
            if (child_refl >= 0 && sp + 1 < RT_MAX_RECURSE) { sp++; node[sp] = child_refl; stage[sp] = 0; continue; }
            ret = mk3(0.0f, 0.0f, 0.0f); chain = 0;
         }
         if (stage[sp] == 1) {
            rx[sp] = ret.x; ry[sp] = ret.y; rz[sp] = ret.z; chain_refl[sp] = chain;
            stage[sp] = 2;
            if (child_refr >= 0 && sp + 1 < RT_MAX_RECURSE) { sp++; node[sp] = child_refr; stage[sp] = 0; continue; }
            ret = mk3(0.0f, 0.0f, 0.0f); chain = 0;
         }
         long long k = node[sp];
         F3 total = mk3(totals[3 * k], totals[3 * k + 1], totals[3 * k + 2]);
         F3 refl = mk3(rx[sp], ry[sp], rz[sp]);
         ret = total * (1.0f - rr.x - rr.y) + rr.x * refl + rr.y * ret;
         if (dur) chain = dur[k] + (chain_refl[sp] > chain ? chain_refl[sp] : chain);
         sp--;
      }
      /* rayTrace's clamp (kernel.cu:537-539); NaN passes through clamp unchanged (Util.h:28-30) */
      if (live) { rgb[3 * i] = clamp01(ret.x); rgb[3 * i + 1] = clamp01(ret.y); rgb[3 * i + 2] = clamp01(ret.z); }
      if (tile_cost) {
         /* What this tile costs, for the order in which the next frame starts its tiles: how long its longest chain of
          * dependent rays was (in traversal steps: a figure that does not depend on how the waves were scheduled, so a
          * still camera gets the same order frame after frame).  A tile started late must be one whose chains are short, or the launch ends on a
          * few waves finishing them.  A wave's 64 primaries lie in one tile (a tile's sample count is a multiple of 64). */
         unsigned cost = live ? chain : 0u;
         unsigned sum = cost;
         for (int o = 32; o > 0; o >>= 1) { unsigned other = __shfl_xor(cost, o); cost = other > cost ? other : cost; sum += __shfl_xor(sum, o); }
         if ((threadIdx.x & 63) == 0 && live && cost) {
            atomicMax(&tile_cost[2 * (unsigned)(i / per_tile)], cost);
            atomicAdd(&tile_cost[2 * (unsigned)(i / per_tile) + 1], sum);
         }
      }
   }
}

__device__ __forceinline__ unsigned char to_byte(float v)
{
   /* float -> unsigned char truncates (kernel.cu:568); NaN becomes 0 */
   if (!(v == v)) return 0;
   int k = (int)v;
   return (unsigned char)(k < 0 ? 0 : k > 255 ? 255 : k);
}

/* resolve: averageBufferColors, kernel.cu:545-570.  Thread = pixel of a batch tile. */
/* What k_light and k_combine_tree leave to this kernel: the tiles none of whose primaries spawned a bounce ray
 * (`tile_bounce` clear).  Every tree there is a single node, so a sample's colour is its own light sum times
 * (1 - refl - alpha) -- the same functions and the same expression as in those kernels, evaluated here, where the
 * samples are averaged anyway: such a tile's hits and spawn records are read once and nothing is written in between. */
struct ResolveShade {
   DScene sc;
   const QRay *queue; const Hit *hits; const SpawnRec *recs; const unsigned *flag_words; unsigned flag_stride; int shading;
   const unsigned *tile_bounce;     /* null: every tile went through k_light and k_combine_tree */
   const unsigned *tile_hit;        /* with tile_bounce: clear for a tile none of whose primaries hit anything (nothing of it is read) */
   const unsigned *dur; unsigned *tile_cost;
};
__global__ __launch_bounds__(RT_BLOCK) void k_resolve(DGrid g, const float *rgb, int n_batch_tiles, int out_layout,
                                                      uchar4 *out, ResolveShade rs)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   const int T = RT_TILE;
   if (i >= (long long)n_batch_tiles * T * T) return;
   int k = (int)(i / (T * T));
   int r = (int)(i % (T * T));
   int lx = r % T, ly = r / T;
   int tile = g.tile_first + (g.k0 + k) * g.tile_stride;
   int tx = tile % g.tiles_x, ty = tile / g.tiles_x;
   int x = tx * T + lx, y = ty * T + ly;
   const bool here = rs.tile_bounce && rs.tile_bounce[k] == 0u;     /* uniform over the workgroup: 256 threads are a quarter of a tile */
   const bool inside = x < g.width && y < g.height;
   if (!inside && !here) return;
   const bool empty = here && rs.tile_hit && rs.tile_hit[k] == 0u;       /* every sample of the tile is a miss: colour +0, cost 0 */
   F3 end = mk3(0.0f, 0.0f, 0.0f);
   long long tile_base = (long long)k * g.ts * g.ts;
   unsigned cost = 0, sum = 0;
   for (int xo = 0; xo < g.sroot && !empty; xo++) {
      for (int yo = 0; yo < g.sroot; yo++) {
         int sx = lx * g.sroot + xo, sy = ly * g.sroot + yo;
         long long slot = tile_base + ((long long)((sy >> 3) * g.patches_per_row + (sx >> 3)) << 6) + ((sy & 7) << 3) + (sx & 7);
         if (here) {
            if (inside) {
               const Hit h = rs.hits[slot];
               if (h.prim >= 0) {
                  /* (a sample that hit nothing has light sum 0 and weights 0, 0: its colour is +0 and adds nothing) */
                  float4 rr = *reinterpret_cast<const float4 *>(rs.recs + slot);
                  F3 total = light_sum(rs.sc, rs.queue, slot, h, rs.flag_words, rs.flag_stride, rs.shading);
                  const F3 none = mk3(0.0f, 0.0f, 0.0f);
                  F3 ret = total * (1.0f - rr.x - rr.y) + rr.x * none + rr.y * none;       /* kernel.cu:375-376 without children */
                  end = end + mk3(clamp01(ret.x), clamp01(ret.y), clamp01(ret.z));          /* kernel.cu:537-539 */
                  if (rs.dur) { unsigned d = rs.dur[slot]; cost = d > cost ? d : cost; sum += d; }
               }
            }
         } else end = end + mk3(rgb[3 * slot], rgb[3 * slot + 1], rgb[3 * slot + 2]);
      }
   }
   if (here && rs.tile_cost) {
      /* the tile's cost for the next frame's start order, as k_combine_tree records it for the other tiles */
      for (int o = 32; o > 0; o >>= 1) { unsigned other = __shfl_xor(cost, o); cost = other > cost ? other : cost; sum += __shfl_xor(sum, o); }
      if ((threadIdx.x & 63) == 0 && cost) {
         atomicMax(&rs.tile_cost[2 * (unsigned)k], cost);
         atomicAdd(&rs.tile_cost[2 * (unsigned)k + 1], sum);
      }
   }
   if (!inside) return;
   float div = (float)(g.sroot * g.sroot);
   end = mk3(end.x / div, end.y / div, end.z / div);
   end = end * 255.0f;
   uchar4 px; px.x = to_byte(end.x); px.y = to_byte(end.y); px.z = to_byte(end.z); px.w = 255;
   if (out_layout == RT_OUT_TILES) out[(long long)(g.k0 + k) * T * T + ly * T + lx] = px;
   else out[(long long)y * g.width + x] = px;
}

/* un-tile: parts made with tile_first = p, tile_stride = n_parts -> row-major frame */
__global__ __launch_bounds__(RT_BLOCK) void k_untile(const uchar4 *tiles, long long part_stride_px, int n_parts,
                                                     int width, int height, int tiles_x, uchar4 *frame)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= (long long)width * height) return;
   int x = (int)(i % width), y = (int)(i / width);
   int tile = (y / RT_TILE) * tiles_x + (x / RT_TILE);
   int part = tile % n_parts, k = tile / n_parts;
   frame[i] = tiles[part * part_stride_px + (long long)k * RT_TILE * RT_TILE + (y % RT_TILE) * RT_TILE + (x % RT_TILE)];
}

/* scatter per-slot values to the row-major sample grid (parity views) */
__global__ __launch_bounds__(RT_BLOCK) void k_unslot_hits(DGrid g, const Hit *hits, long long n, const int *prim_ids, int n_prims,
                                                          int *prim, float *t)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= n) return;
   int X, Y;
   if (!slot_to_sample(g, i, X, Y)) return;
   long long idx = (long long)Y * g.res_w + X;
   int p = hits[i].prim;
   prim[idx] = (p >= 0 && p < n_prims) ? prim_ids[p] : p;      /* the caller's numbering; planes follow the BVH primitives */
   t[idx] = hits[i].t;
}
__global__ __launch_bounds__(RT_BLOCK) void k_unslot_rgb(DGrid g, const float *rgb, long long n, float *out)
{
   long long i = (long long)blockIdx.x * RT_BLOCK + threadIdx.x;
   if (i >= n) return;
   int X, Y;
   if (!slot_to_sample(g, i, X, Y)) return;
   long long idx = (long long)Y * g.res_w + X;
   out[3 * idx] = rgb[3 * i]; out[3 * idx + 1] = rgb[3 * i + 1]; out[3 * idx + 2] = rgb[3 * i + 2];
}

#endif
