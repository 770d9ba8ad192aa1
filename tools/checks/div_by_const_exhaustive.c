/* Exhaustive check behind greb_device.h:div_by_const: for c = 20 and c = 3 and ALL 2^32 fp32 operands, compare IEEE x/c
 * with q = x*r, e = fma(-q,c,x), fma(e,r,q).  gcc -O2 -fopenmp -ffp-contract=off -mfma div_by_const_exhaustive.c -lm; ~40 s on 8 cores.
 * Result (this container): c=20: mismatches only for |x| < 4.8e-38; c=3: only x = -0. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <omp.h>
static inline float div_c(float x, float c, float r) {
  float q = x * r;
  float e = fmaf(-q, c, x);
  return fmaf(e, r, q);
}
int main(int argc, char** argv) {
  const float cs[2] = {20.f, 3.f};
  for (int ci = 0; ci < 2; ++ci) {
    const float c = cs[ci], r = 1.0f / c;
    uint64_t bad = 0, bad_normal = 0; uint32_t first = 0; int have = 0;
    float minbad = INFINITY, maxbad = 0;
#pragma omp parallel for reduction(+:bad,bad_normal) schedule(static)
    for (int64_t i = 0; i < (1LL << 32); ++i) {
      uint32_t u = (uint32_t)i; float x; memcpy(&x, &u, 4);
      if (isnan(x) || isinf(x)) continue;
      volatile float a = x / c;
      float b = div_c(x, c, r);
      uint32_t ua, ub; float aa = a; memcpy(&ua, &aa, 4); memcpy(&ub, &b, 4);
      if (ua != ub) {
        bad++;
        if (fabsf(x) >= 1.17549435e-38f * 64.f && fabsf(aa) >= 1.17549435e-38f) bad_normal++;
#pragma omp critical
        { if (!have) { first = u; have = 1; } if (fabsf(x) < minbad) minbad = fabsf(x); if (fabsf(x) > maxbad) maxbad = fabsf(x); }
      }
    }
    printf("c=%g: mismatches %llu (with normal x and quotient: %llu) first=0x%08x |x| range of mismatches [%g, %g]\n", c,
           (unsigned long long)bad, (unsigned long long)bad_normal, first, minbad, maxbad);
  }
  return 0;
}
