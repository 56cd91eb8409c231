#include <immintrin.h>
#include <dlfcn.h>
#include <stdio.h>
typedef __m512d (*fn_t)(__m512d);
static void* h;
int init(const char* so){ h=dlopen(so,RTLD_NOW|RTLD_GLOBAL); return h==0; }
int call(const char* sym,const double*x,double*o,long n){ fn_t f=(fn_t)dlsym(h,sym); if(!f) return 1; for(long i=0;i<n;i+=8){ __m512d v=_mm512_loadu_pd(x+i); v=f(v); _mm512_storeu_pd(o+i,v);} return 0; }
