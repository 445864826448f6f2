// Exhaustive check of the 3-operation division q1 = fma(fma(-d,q0,a), y, q0), q0 = a*y, y = RN(1/d)
// on the HARD cases: every (a, d) significand pair whose exact quotient lies within 8/(2B) ulp of a
// rounding boundary (midpoint between two floats).  Elsewhere the final fma cannot misround (the total
// perturbation is < 2^-22 ulp), so passing here proves q1 == RN(a/d) for all normal-range operands.
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <omp.h>
static inline float mk(uint32_t A, int e){ return ldexpf((float)A, e); }
static inline float div3(float a,float d,float y){ float q0=a*y; float r0=fmaf(-d,q0,a); return fmaf(r0,y,q0);}
static uint64_t inv_mod_pow2(uint64_t b, int bits){ // b odd
  uint64_t x=b; for(int i=0;i<6;i++) x*=2-b*x; return x & ((bits==64)?~0ULL:((1ULL<<bits)-1)); }
int main(){
  long long tested=0,bad=0;
  #pragma omp parallel for schedule(dynamic,4096) reduction(+:tested,bad)
  for(uint32_t B=1u<<23; B<(1u<<24); B++){
    for(int casei=0; casei<2; casei++){
      // case 0: A<B, quotient in (1/2,1): midpoints M*2^-25, M odd in [2^24,2^25): R = A*2^25 - B*M
      // case 1: A>=B, quotient in [1,2):  midpoints M*2^-24:                      R = A*2^24 - B*M
      int sh = casei==0 ? 25 : 24;
      int k = __builtin_ctz(B); uint32_t Bo = B>>k; int bits = sh-k; // B*M = -R (mod 2^sh) -> need 2^k | R
      uint64_t inv = inv_mod_pow2(Bo, bits);
      for(int R=-8; R<=8; R++){ if(R==0) continue; if(R % (1<<(k>4?4:k)) && k<=4) continue; if(k>4 && (R & ((1<<k)-1))) continue;
        int64_t Rk = R / (1<<k); if ((int64_t)Rk*(1<<k)!=R) continue;
        // M = (-Rk * inv) mod 2^bits, plus multiples of 2^bits up to 2^sh
        uint64_t mask=(1ULL<<bits)-1; uint64_t M0 = ((uint64_t)(-(int64_t)Rk) * inv) & mask;
        for(uint64_t M=M0; M<(1ULL<<25); M+=(1ULL<<bits)){
          if(!(M&1)) continue; if(M < (1ULL<<24)) continue;
          __int128 num = (__int128)B*M + R; if(num<=0) continue; if(num & (((__int128)1<<sh)-1)) continue;
          uint64_t A = (uint64_t)(num>>sh); if(A<(1u<<23)||A>=(1u<<24)) continue;
          if(casei==0 && !(A<B)) continue; if(casei==1 && !(A>=B)) continue;
          for(int s=0;s<4;s++){ float a=mk((uint32_t)A,-23)*((s&1)?-1.f:1.f), d=mk(B,-23+((s&2)?5:-9))*((s&2)?-1.f:1.f);
            float y=1.0f/d; float q=a/d, q3=div3(a,d,y); tested++;
            if(memcmp(&q,&q3,4)){ bad++; if(bad<10) printf("BAD A=%llu B=%u R=%d q=%a q3=%a\n",(unsigned long long)A,B,R,q,q3);} }
        }
      }
    }
  }
  printf("hard cases tested=%lld bad=%lld\n",tested,bad);
  return 0;
}
