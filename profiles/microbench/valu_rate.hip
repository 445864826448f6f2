// VALU issue-rate probe: independent v_fma_f32 vs v_pk_fma_f32 streams at 1..8 waves/SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int PK>
__global__ void k(float *out, int iters, float a, float b)
{
   float x[16]; v2f y[8];
   for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;
   for (int i = 0; i < 8; i++) { y[i].x = threadIdx.x + i; y[i].y = threadIdx.x - i; }
   v2f aa = {a, a}, bb = {b, b};
   for (int it = 0; it < iters; it++) {
      if (PK) {
#pragma unroll
         for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y[i]) : "v"(aa), "v"(bb));
#pragma unroll
         for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y[i]) : "v"(aa), "v"(bb));
      } else {
#pragma unroll
         for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
      }
   }
   float s = 0; for (int i = 0; i < 16; i++) s += x[i]; for (int i = 0; i < 8; i++) s += y[i].x + y[i].y;
   if (s == 123.456f) out[0] = s;
}
int main()
{
   float *d; (void)hipMalloc(&d, 4);
   hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
   int cus = p.multiProcessorCount; double ghz = p.clockRate * 1e-6;
   printf("CUs %d clock %.3f GHz\n", cus, ghz);
   hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
   const int iters = 200000;
   for (int pk = 0; pk < 2; pk++)
      for (int wps = 1; wps <= 8; wps *= 2) {
         int block = 256, blocks = cus * wps;   // 4 waves per block = 1 per SIMD per block
         for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            if (pk) k<1><<<blocks, block>>>(d, iters, 1.0001f, 0.5f); else k<0><<<blocks, block>>>(d, iters, 1.0001f, 0.5f);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) {
               double instr_per_simd = (double)iters * 16 * wps;
               double cyc = ms * 1e-3 * ghz * 1e9;
               printf("pk=%d waves/SIMD=%d  %.3f ms  cycles/instr/SIMD = %.2f (at %.2f GHz)\n", pk, wps, ms, cyc / instr_per_simd, ghz);
            }
         }
      }
   return 0;
}
