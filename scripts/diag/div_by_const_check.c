// Exhaustive check: q0 = a*r; e = fma(-d, q0, a); q = fma(e, r, q0)  ==  RN32(a/d) ?  (r = RN32(1/d))
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
int main(int argc, char** argv) {
  const float d = (float)atof(argv[1]);
  const float lo = (float)atof(argv[2]), hi = (float)atof(argv[3]);
  const float r = 1.0f / d;
  const double rd = 1.0 / (double)d;
  uint64_t bad = 0, bad2 = 0, n = 0;
  for (uint64_t bits = 0; bits < (1ull << 32); ++bits) {
    uint32_t u = (uint32_t)bits; float a; memcpy(&a, &u, 4);
    if (!(a >= lo && a <= hi)) continue;
    ++n;
    const float q0 = a * r;
    const float e = fmaf(-d, q0, a);
    const float q = fmaf(e, r, q0);
    const float ref = (float)((double)a / (double)d);
    const float ref2 = (float)((double)a * rd);
    if (q != ref && !(q == 0 && ref == 0)) { if (bad < 5) printf("bad a=%a q=%a ref=%a\n", a, q, ref); ++bad; }
    if (ref2 != ref) ++bad2;
  }
  printf("d=%g range [%g,%g]: %llu values, markstein mismatches %llu, recip-product mismatches %llu\n", d, lo, hi,
         (unsigned long long)n, (unsigned long long)bad, (unsigned long long)bad2);
  return 0;
}
