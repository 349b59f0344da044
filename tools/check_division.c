// Validates the shared-reciprocal division of vfa_kernels.hip (box_mean) against IEEE division on 4e8 random and
// adversarial operand pairs.  gcc -O2 -mfma -ffp-contract=off tools/check_division.c -lm && ./a.out
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
static inline float asf(uint32_t u){float f; memcpy(&f,&u,4); return f;}
static inline uint32_t asu(float f){uint32_t u; memcpy(&u,&f,4); return u;}
static uint64_t s=88172645463325252ULL;
static inline uint64_t rnd(){ s^=s<<13; s^=s>>7; s^=s<<17; return s; }
static inline float div5(float a, float b, float r){
    float q0 = a*r; float e0 = fmaf(-b,q0,a); float q1 = fmaf(e0,r,q0); float e1 = fmaf(-b,q1,a); return fmaf(e1,r,q1);
}
static inline float div3(float a, float b, float r){
    float q0 = a*r; float e0 = fmaf(-b,q0,a); return fmaf(e0,r,q0);
}
int main(){
    long bad5=0,bad3=0,n=0;
    for(long i=0;i<400000000L;i++){
        uint64_t x=rnd();
        uint32_t ma = x & 0x7fffff, mb = (x>>23)&0x7fffff;
        int ea = 127 + (int)((x>>46)%60) - 40;   // a in 2^-40 .. 2^19
        int eb = 127 + (int)((x>>52)%40) - 20;   // b in 2^-20 .. 2^19
        if ((i&7)==0) mb = 0x7fffff - (mb & 0xff);      // significand near 2
        if ((i&7)==1) mb = (mb & 0xff);                 // near 1
        if ((i&7)==2) ma = 0x7fffff - (ma & 0xfff);
        float a = asf(((x>>63)<<31) | ((uint32_t)ea<<23) | ma), b = asf(((uint32_t)eb<<23)|mb);
        float r = 1.0f/b; float q = a/b;
        if (asu(div5(a,b,r))!=asu(q)) bad5++;
        if (asu(div3(a,b,r))!=asu(q)) bad3++;
        n++;
    }
    printf("n=%ld bad5=%ld bad3=%ld\n",n,bad5,bad3);
    return 0;
}
