// Exhaustive check of the restated glibc sinf/cosf against the libm of the machine it runs on, all floats |x| < 120.
// gcc -O2 -fopenmp -ffp-contract=off -fno-builtin -mfma -DUSEFMA sincosf_exhaustive.c -lm   (FMA build of glibc: 0 mismatches)
// gcc -O2 -fopenmp -ffp-contract=off -fno-builtin sincosf_exhaustive.c -lm                  (without fused steps: 12 + 22)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>
typedef struct { double sign[4]; double hpi_inv, hpi, c0,c1,c2,c3,c4,s1,s2,s3; } sincos_t;
static const sincos_t T[2] = {
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
#ifdef USEFMA
#define MA(a,b,c) fma(a,b,c)
#else
#define MA(a,b,c) ((a)*(b)+(c))
#endif
static inline uint32_t top12(float y){uint32_t u; memcpy(&u,&y,4); return (u>>20)&0x7ff;}
static inline float poly(double x, double x2, const sincos_t*p, int n){
  if((n&1)==0){ double x3=x*x2; double s1=MA(x2,p->s3,p->s2); double x7=x3*x2; double s=MA(x3,p->s1,x); return (float)MA(x7,s1,s);}
  else { double x4=x2*x2; double c2=MA(x2,p->c4,p->c3); double c1=MA(x2,p->c2,p->c1); double x6=x4*x2; double c=MA(x2,c1,p->c0); return (float)MA(x6,c2,c);}
}
static inline double reduce_fast(double x,const sincos_t*p,int*np){ double r=x*p->hpi_inv; int n=((int32_t)r+0x800000)>>24; *np=n; return MA(-(double)n,p->hpi,x);}
float my_sinf(float y){ double x=y; int n; const sincos_t*p=&T[0];
  if(top12(y)<0x3f4){ double s=x*x; if(top12(y)<0x398) return y; return poly(x,s,p,0);} 
  else if(top12(y)<0x42f){ x=reduce_fast(x,p,&n); double s=p->sign[n&3]; if(n&2)p=&T[1]; return poly(x*s,x*x,p,n);} 
  return sinf(y);}
float my_cosf(float y){ double x=y; int n; const sincos_t*p=&T[0];
  if(top12(y)<0x3f4){ double s=x*x; if(top12(y)<0x398) return 1.0f; return poly(x,s,p,1);} 
  else if(top12(y)<0x42f){ x=reduce_fast(x,p,&n); double s=p->sign[n&3]; if(n&2)p=&T[1]; return poly(x*s,x*x,p,n^1);} 
  return cosf(y);}
int main(){ long bs=0,bc=0,tot=0; uint32_t fs=0,fc=0;
 #pragma omp parallel for reduction(+:bs,bc,tot) schedule(static)
 for(int64_t i=0;i<(1ll<<32);i++){ uint32_t u=(uint32_t)i; float y; memcpy(&y,&u,4); if(!(fabsf(y)<120.f)) continue; tot++;
   volatile float yy=y; float a=sinf(yy), b=my_sinf(y); if(memcmp(&a,&b,4)){bs++; fs=u;} a=cosf(yy); b=my_cosf(y); if(memcmp(&a,&b,4)){bc++; fc=u;} }
 printf("tested %ld sin mismatches %ld (e.g. %08x) cos mismatches %ld (e.g. %08x)\n",tot,bs,fs,bc,fc); return 0;}
