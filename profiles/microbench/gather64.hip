// Divergent 64-byte record gather, as the BVH node fetch does it: what does the vector memory path charge for
//   A: each lane fetches its own record with four 16-byte loads (64 distinct lines per instruction)
//   B: the four lanes of a quad fetch one record together, four records in four instructions (16 lines per instruction)
// 7/8 of the fetches go to a hot set of 256 records, the rest anywhere in the table (about what the traversal's
// 89 % vector-cache hit rate looks like).  Next index depends on the fetched data, so steps are serialised per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ unsigned pick(unsigned h, unsigned n, unsigned cold) { return (h & 1023u) >= cold ? ((h >> 10) & 255u) : ((h >> 10) % n); }
#define LDSREC 96u
template <int MODE, int CHUNKS>
__global__ __launch_bounds__(256) void k(const uint4 *tab, unsigned n, int iters, unsigned *out, unsigned cold)
{
   __shared__ uint4 lds[LDSREC * 4];
   if (MODE == 2) { for (unsigned i = threadIdx.x; i < LDSREC * 4; i += 256) lds[i] = tab[i]; __syncthreads(); }
   unsigned lane = threadIdx.x & 63u, c = lane & 3u;
   unsigned idx = pick(mix(blockIdx.x * 256u + threadIdx.x), n, cold);
   unsigned acc = 0;
   for (int it = 0; it < iters; it++) {
      uint4 v0 = make_uint4(1, 2, 3, 4), v1 = v0, v2 = v0, v3 = v0;
      if (MODE == 0) {
         const uint4 *p = tab + (size_t)idx * 4;
         v0 = p[0]; if (CHUNKS > 1) v1 = p[1]; if (CHUNKS > 2) v2 = p[2]; if (CHUNKS > 3) v3 = p[3];
      } else if (MODE == 3) {
         /* D: 56 of the 64 bytes: three 16-byte loads and one 8-byte load (hand-placed: the compiler widens the last) */
         const uint4 *p = tab + (size_t)idx * 4;
         uint2 w;
         asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                      "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx2 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
                      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(w) : "v"(p) : "memory");
         v3.x = w.x; v3.y = w.y; v3.z = 0; v3.w = w.y;
      } else if (MODE == 4) {
         /* E: the same 64 bytes as eight 8-byte loads (twice the requests) */
         const uint4 *p = tab + (size_t)idx * 4;
         uint2 a0, a1, a2, a3, a4, a5, a6, a7;
         asm volatile("global_load_dwordx2 %0, %8, off\n\tglobal_load_dwordx2 %1, %8, off offset:8\n\t"
                      "global_load_dwordx2 %2, %8, off offset:16\n\tglobal_load_dwordx2 %3, %8, off offset:24\n\t"
                      "global_load_dwordx2 %4, %8, off offset:32\n\tglobal_load_dwordx2 %5, %8, off offset:40\n\t"
                      "global_load_dwordx2 %6, %8, off offset:48\n\tglobal_load_dwordx2 %7, %8, off offset:56\n\ts_waitcnt vmcnt(0)"
                      : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(p) : "memory");
         v0 = make_uint4(a0.x, a0.y, a1.x, a1.y); v1 = make_uint4(a2.x, a2.y, a3.x, a3.y);
         v2 = make_uint4(a4.x, a4.y, a5.x, a5.y); v3 = make_uint4(a6.x, a6.y, a7.x, a7.y);
      } else if (MODE == 5) {
         /* F: all 64 bytes with hand-placed loads, the reference point for D and E */
         const uint4 *p = tab + (size_t)idx * 4;
         asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                      "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
                      : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p) : "memory");
      } else if (MODE == 2) {
         /* C: the first LDSREC records are also held in LDS; a fetch of one of them goes there */
         if (idx < LDSREC) {
            const uint4 *p = lds + (size_t)idx * 4;
            v0 = p[0]; v1 = p[1]; v2 = p[2]; v3 = p[3];
         } else {
            const uint4 *p = tab + (size_t)idx * 4;
            v0 = p[0]; v1 = p[1]; v2 = p[2]; v3 = p[3];
         }
      } else {
         unsigned t0 = __shfl(idx, (lane & ~3u) | 0u), t1 = __shfl(idx, (lane & ~3u) | 1u);
         unsigned t2 = __shfl(idx, (lane & ~3u) | 2u), t3 = __shfl(idx, (lane & ~3u) | 3u);
         v0 = tab[(size_t)t0 * 4 + c]; v1 = tab[(size_t)t1 * 4 + c]; v2 = tab[(size_t)t2 * 4 + c]; v3 = tab[(size_t)t3 * 4 + c];
      }
      unsigned h = v0.x ^ v1.y ^ v2.z ^ v3.w ^ (unsigned)it;
      acc += h;
      idx = pick(mix(h + lane), n, cold);
   }
   if (acc == 0x12345678u) out[0] = acc;
}
template <int MODE, int CHUNKS>
static int run(const uint4 *tab, unsigned n, int iters, unsigned *out, int blocks, unsigned cold)
{
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      k<MODE, CHUNKS><<<blocks, 256>>>(tab, n, iters, out, cold);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) printf("mode %d chunks %d blocks %4d cold %4u/1024: %7.3f ms, %7.1f ns per wave-step, %6.1f B/clk/CU\n", MODE, CHUNKS, blocks, cold, ms,
                      ms * 1e6 / iters, (double)blocks * 4 * iters * 1024.0 * CHUNKS / 256.0 / (ms * 1e-3 * 2.4e9));
   }
   return 0;
}
int main()
{
   const unsigned n = 7382; const int iters = 4000;
   std::vector<unsigned> h((size_t)n * 16);
   for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)rand() * 2654435761u;
   uint4 *tab; unsigned *out;
   CK(hipMalloc(&tab, h.size() * 4)); CK(hipMalloc(&out, 4));
   CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
   for (int blocks = 256; blocks <= 2048; blocks *= 2) run<0, 4>(tab, n, iters, out, blocks, 128);
   run<0, 1>(tab, n, iters, out, 1024, 128); run<0, 2>(tab, n, iters, out, 1024, 128); run<0, 3>(tab, n, iters, out, 1024, 128);
   run<0, 4>(tab, n, iters, out, 1024, 0); run<0, 4>(tab, n, iters, out, 1024, 32); run<0, 4>(tab, n, iters, out, 1024, 512); run<0, 4>(tab, n, iters, out, 1024, 1024);
   run<1, 4>(tab, n, iters, out, 1024, 128); run<1, 4>(tab, n, iters, out, 1024, 0);
   run<2, 4>(tab, n, iters, out, 1024, 128); run<2, 4>(tab, n, iters, out, 1024, 0);
   run<5, 4>(tab, n, iters, out, 1024, 128); run<3, 4>(tab, n, iters, out, 1024, 128); run<4, 4>(tab, n, iters, out, 1024, 128);
   return 0;
}
