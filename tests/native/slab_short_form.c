// Brute-force equivalence of the reference's literal box test (GeometryUtil.h:9-51) and the short form
// the device traversal uses (rt_device.h slab_fast<PAR>), bit for bit, on randomised boxes and rays.
// Covers ordinary rays (PAR=0 form) and rays with one, two or three "parallel" axes (|d| < EPSILON,
// including exact zeros and origins exactly on a slab face), for which both PAR=1 and, when no axis is
// parallel, PAR=0 must agree with the literal result.
#include <stdio.h>
#include <stdint.h>
#include <math.h>
#include <float.h>
#include <string.h>
#include <omp.h>
#ifndef TRIALS
#define TRIALS 200000000LL
#endif
#define EPS 0.0001f
static inline int isz(float n){return n<EPS && n>-EPS;}
static inline float mymin(float x,float y){return x<y?x:y;} static inline float mymax(float x,float y){return x>y?x:y;}
static float lit(const float o[3],const float d[3],const float mn[3],const float mx[3]){
  float tmin[3],tmax[3]; int miss=0;
  for(int a=0;a<3;a++){tmin[a]=-FLT_MAX;tmax[a]=FLT_MAX;
    if(!isz(d[a])){tmin[a]=(mn[a]-o[a])/d[a];tmax[a]=(mx[a]-o[a])/d[a];} else if(o[a]<mn[a]||o[a]>mx[a]) miss=1;}
  if(miss) return -1.0f;
  for(int a=0;a<3;a++) if(tmin[a]>tmax[a]){float s=tmin[a];tmin[a]=tmax[a];tmax[a]=s;}
  float A=mymax(mymax(tmin[0],tmin[1]),tmin[2]); float B=mymin(mymin(tmax[0],tmax[1]),tmax[2]);
  if(o[0]>mn[0]&&o[0]<mx[0]&&o[1]>mn[1]&&o[1]<mx[1]&&o[2]>mn[2]&&o[2]<mx[2]) return B;
  return A<B?A:-1.0f;}
static inline float fdiv(float a,float d,float y){ float q0=a*y; float r0=fmaf(-d,q0,a); return fmaf(r0,y,q0);}
static float fast(int PAR,const float o[3],const float d[3],const float y[3],const float mn[3],const float mx[3]){
  float lo[3],hi[3]; int miss=0,strict=1;
  for(int a=0;a<3;a++){float q1=fdiv(mn[a]-o[a],d[a],y[a]),q2=fdiv(mx[a]-o[a],d[a],y[a]); lo[a]=fminf(q1,q2); hi[a]=fmaxf(q1,q2);
    if(PAR && isz(d[a])){ lo[a]=-FLT_MAX; hi[a]=FLT_MAX; if(o[a]<mn[a]||o[a]>mx[a]) miss=1; if(!(o[a]>mn[a]&&o[a]<mx[a])) strict=0; } }
  float A=fmaxf(fmaxf(lo[0],lo[1]),lo[2]),B=fminf(fminf(hi[0],hi[1]),hi[2]);
  float res=(A<0.0f&&B>0.0f&&strict)?B:(A<B?A:-1.0f);
  return miss?-1.0f:res;}
static inline uint64_t rng(uint64_t *s){ uint64_t x=*s; x^=x<<13; x^=x>>7; x^=x<<17; *s=x; return x*0x2545F4914F6CDD1DULL; }
static inline float uf(uint64_t *s){return (float)((rng(s)>>40)/16777216.0);} // [0,1)
int main(){ long long bad=0,tot=0,zs=0,npar=0;
 #pragma omp parallel reduction(+:bad,tot,zs,npar)
 { uint64_t s=0x9E3779B97F4A7C15ULL*(omp_get_thread_num()+7);
  for(long long i=0;i<TRIALS;i++){
    float o[3],d[3],y[3],mn[3],mx[3]; int mode=rng(&s)&15; int anypar=0;
    for(int a=0;a<3;a++){ mn[a]=(uf(&s)-0.5f)*8; mx[a]=mn[a]+uf(&s)*4+ (mode==3?0:0.0002f);
      o[a]=(uf(&s)-0.5f)*12; if(mode==1||mode==9) o[a]=mn[a]+(mx[a]-mn[a])*uf(&s); if((mode==2||mode==10)&&a==(int)(rng(&s)%3)) o[a]=(rng(&s)&1)?mn[a]:mx[a];
      if(mode==4) o[a]= roundf(o[a]); if(mode==5){mn[a]=roundf(mn[a]); mx[a]=mn[a]+1+roundf(uf(&s)*3);}
      do{ d[a]=(uf(&s)-0.5f)*2; if(mode==6) d[a]*=0.001f; }while(isz(d[a]));
      if(mode>=8 && (rng(&s)&3)==0){ // a parallel axis: tiny, zero or negative zero
        int k=rng(&s)&3; d[a]= k==0?0.0f: k==1?-0.0f: (uf(&s)-0.5f)*1.9e-4f; if(!isz(d[a])) d[a]=5e-5f;
        if(rng(&s)&1) o[a]=mn[a]+(mx[a]-mn[a])*uf(&s); if((rng(&s)&7)==0) o[a]=(rng(&s)&1)?mn[a]:mx[a]; }
      anypar|=isz(d[a]); y[a]=1.0f/d[a]; }
    float r1=lit(o,d,mn,mx), r2=fast(1,o,d,y,mn,mx);
    if(r1!=r2 && !(r1==0&&r2==0)) {bad++; if(bad<5) printf("bad %a %a mode %d par %d\n",r1,r2,mode,anypar);}
    if(memcmp(&r1,&r2,4)) zs++;
    if(!anypar){ float r3=fast(0,o,d,y,mn,mx); if(memcmp(&r1,&r3,4)) {zs++; if(r1!=r3) bad++;} }
    npar+=anypar; tot++; } }
 printf("tot=%lld bad=%lld bitdiff(incl zero sign)=%lld with_parallel_axis=%lld\n",tot,bad,zs,npar);}
