// Exhaustive check of the restated glibc logf against the libm of the machine it runs on (all positive finite floats).
// gcc -O2 -fopenmp -ffp-contract=off -fno-builtin [-mfma -DUSEFMA] logf_exhaustive.c -lm      (glibc 2.35: 0 mismatches either way)
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static const double T[16][2]={
 {0x1.661ec79f8f3bep+0,-0x1.57bf7808caadep-2},{0x1.571ed4aaf883dp+0,-0x1.2bef0a7c06ddbp-2},{0x1.49539f0f010bp+0,-0x1.01eae7f513a67p-2},
 {0x1.3c995b0b80385p+0,-0x1.b31d8a68224e9p-3},{0x1.30d190c8864a5p+0,-0x1.6574f0ac07758p-3},{0x1.25e227b0b8eap+0,-0x1.1aa2bc79c81p-3},
 {0x1.1bb4a4a1a343fp+0,-0x1.a4e76ce8c0e5ep-4},{0x1.12358f08ae5bap+0,-0x1.1973c5a611cccp-4},{0x1.0953f419900a7p+0,-0x1.252f438e10c1ep-5},
 {0x1p+0,0x0p+0},{0x1.e608cfd9a47acp-1,0x1.aa5aa5df25984p-5},{0x1.ca4b31f026aap-1,0x1.c5e53aa362eb4p-4},
 {0x1.b2036576afce6p-1,0x1.526e57720db08p-3},{0x1.9c2d163a1aa2dp-1,0x1.bc2860d22477p-3},{0x1.886e6037841edp-1,0x1.1058bc8a07ee1p-2},
 {0x1.767dcf5534862p-1,0x1.4043057b6ee09p-2}};
static const double Ln2=0x1.62e42fefa39efp-1, A0=-0x1.00ea348b88334p-2, A1=0x1.5575b0be00b6ap-2, A2=-0x1.ffffef20a4123p-2;
#ifdef USEFMA
#define MA(a,b,c) fma(a,b,c)
#else
#define MA(a,b,c) ((a)*(b)+(c))
#endif
float my_logf(float x){ uint32_t ix; memcpy(&ix,&x,4); if(ix==0x3f800000) return 0;
 if(ix-0x00800000>=0x7f800000-0x00800000){ if(ix*2==0) return -INFINITY; if(ix==0x7f800000) return x; if((ix&0x80000000)||ix*2>=0xff000000) return (x-x)/(x-x);
   float xs=x*0x1p23f; memcpy(&ix,&xs,4); ix-=23<<23; }
 uint32_t tmp=ix-0x3f330000; int i=(tmp>>19)%16; int k=(int32_t)tmp>>23; uint32_t iz=ix-(tmp&0x1ff<<23);
 double invc=T[i][0], logc=T[i][1]; float zf; memcpy(&zf,&iz,4); double z=zf;
 double r=MA(z,invc,-1.0); double y0=MA((double)k,Ln2,logc); double r2=r*r; double y=MA(A1,r,A2); y=MA(A0,r2,y); y=MA(y,r2,y0+r); return (float)y; }
int main(){ long b=0,n=0; uint32_t e=0;
 #pragma omp parallel for reduction(+:b,n)
 for(int64_t i=1;i<0x7f800000ll;i++){ uint32_t u=(uint32_t)i; float x; memcpy(&x,&u,4); volatile float xx=x; float a=logf(xx), c=my_logf(x); n++; if(memcmp(&a,&c,4)){b++; e=u;} }
 printf("logf: %ld tested, %ld mismatches (e.g. %08x)\n",n,b,e); return 0;}
