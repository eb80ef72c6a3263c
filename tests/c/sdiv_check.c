/* The arithmetic behind div3_shared (jxl-oxide_amd/csrc/post_pk.inc): a correctly rounded f32 quotient
 * n / d is what the chain  r = rcp(d); e = fma(-d, r, 1); r = fma(e, r, r); q0 = n * r;
 * t1 = fma(d, q0, -n); q1 = fma(-t1, r, q0); t2 = fma(d, q1, -n); q = fma(-t2, r, q1)  returns for ANY
 * reciprocal estimate within 1 ulp of 1/d (the accuracy of v_rcp_f32), as long as 1 <= d <= 2^20 and
 * 2^-100 <= |n| <= 2^20 (no step leaves the normal range; this is what LLVM's own division expansion
 * computes between v_div_scale and v_div_fixup).  The program checks the chain against the host's
 * IEEE division on random (n, d) with every reciprocal in {RN(1/d) - 1 ulp, RN(1/d), RN(1/d) + 1 ulp},
 * half of the denominators drawn from [1, 8) where the kernel's sum_w lives.
 * argv[1] = number of random (n, d) pairs. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static float u2f(uint32_t u){float f;memcpy(&f,&u,4);return f;}
static uint32_t f2u(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static uint64_t s=88172645463325252ull;
static uint64_t rnd(){s^=s<<13;s^=s>>7;s^=s<<17;return s;}
static float shared_div(float n,float d,float r0){
  float e=fmaf(-d,r0,1.0f); float r=fmaf(e,r0,r0);
  float q0=n*r; float t1=fmaf(d,q0,-n); float q1=fmaf(-t1,r,q0); float t2=fmaf(d,q1,-n); return fmaf(-t2,r,q1);
}
int main(int argc, char** argv){
  long bad=0,tot=0;
  long N = argc > 1 ? atol(argv[1]) : 4000000;
  for(long it=0;it<N;++it){
    // d in [1, 2^20], n magnitude in [2^-100, 2^20], random mantissas and signs
    uint32_t ed=127+((it&1)?(rnd()%3):(rnd()%21)); uint32_t md=rnd()&0x7fffff; float d=u2f((ed<<23)|md);
    uint32_t en=27+(rnd()%(147-27+1)); uint32_t mn=rnd()&0x7fffff; float n=u2f(((rnd()&1)<<31)|(en<<23)|mn);
    if (fabsf(n)>0x1p20f||fabsf(n)<0x1p-100f) continue;
    float want=n/d;
    float rc=1.0f/d;
    for(int k=-1;k<=1;++k){ // rcp within 1 ulp of the correctly rounded reciprocal
      float r0=u2f(f2u(rc)+k);
      float got=shared_div(n,d,r0);
      ++tot;
      if (f2u(got)!=f2u(want)){ if(bad<10) printf("n=%a d=%a r0=%a got=%a want=%a\n",n,d,r0,got,want); ++bad; }
    }
  }
  /* exact zeros of either sign keep their sign (the residual is formed as -(d q - n)) */
  for(int sgn=0;sgn<2;++sgn) for(int it=0;it<2000;++it){
    uint32_t ed=127+(rnd()%21); float d=u2f((ed<<23)|(rnd()&0x7fffff)); float n=u2f((uint32_t)sgn<<31);
    float rc=1.0f/d;
    for(int k=-1;k<=1;++k){ float got=shared_div(n,d,u2f(f2u(rc)+k)); ++tot; if(f2u(got)!=f2u(n/d)){ if(bad<10) printf("zero: n=%a d=%a got=%a\n",n,d,got); ++bad; } }
  }
  printf("tested %ld, mismatches %ld\n",tot,bad);
  return bad!=0;
}
