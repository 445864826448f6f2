// Returning atomicAdd from one lane per wave, 4 096 resident waves: one shared word against one word per wave,
// one word per workgroup, and one word per 32 workgroups.  ns per atomic as seen by the whole chip and by one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k(unsigned long long *ctr, int stride_words, int share_blocks, int iters, unsigned long long *sink)
{
   const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
   unsigned long long *p = ctr;
   if (share_blocks > 0) p = ctr + (size_t)(blockIdx.x / share_blocks) * stride_words;
   else if (share_blocks < 0) p = ctr + (size_t)wave * stride_words;
   unsigned long long acc = 0;
   for (int i = 0; i < iters; i++) {
      unsigned long long v = 0;
      if ((threadIdx.x & 63) == 0) v = atomicAdd(p, 64ULL);
      v = __shfl(v, 0);
      acc += v;       // the next iteration depends on this one's return, as a refill does
      if (acc == 0x123456789ULL) p++;
   }
   if (acc == 42) sink[0] = acc;
}

int main()
{
   unsigned long long *d, *sink;
   (void)hipMalloc(&d, 4096 * 64 * 8); (void)hipMalloc(&sink, 8);
   hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
   struct { const char *name; int share; } cases[] = {
      {"one word for the chip", 0}, {"one word per 256 workgroups", 256}, {"one word per 32 workgroups", 32},
      {"one word per workgroup", 1}, {"one word per wave", -1}};
   const int iters = 200;
   for (auto &c : cases)
      for (int blocks : {256, 1024}) {
         float best = 1e9f;
         for (int rep = 0; rep < 3; rep++) {
            (void)hipMemset(d, 0, 4096 * 64 * 8);
            (void)hipEventRecord(e0);
            k<<<blocks, 256>>>(d, 8, c.share, iters, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
         }
         double total = (double)blocks * 4 * iters;
         printf("%-32s %4d workgroups: %8.3f ms, %7.1f ns per atomic chip-wide, %8.1f ns per atomic per wave\n", c.name, blocks, best,
                best * 1e6 / total, best * 1e6 / iters);
      }
   return 0;
}
