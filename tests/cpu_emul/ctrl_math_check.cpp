// Test infrastructure (never part of the product path): csrc/voice_ctrl_math.h compiled for the host and compared with
// libm -- the functions the oracle's "cr" arithmetic is defined by (evaluate in fp64, round once to fp32).
//   ctrl_math_check(kind, n, seed, out[4]) -> out = {evaluations, fp32 mismatches, max |fast - libm| / |libm| in units of
//   2^-53, evaluations outside the fast domain}
#include <cstdint>
#include <cmath>
#include "voice_ctrl_math.h"

static const double g_tab[IAS_CTL_TAB_DOUBLES] = IAS_CTL_TAB_INIT;

static inline uint64_t splitmix(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static inline float u01(uint64_t& s) { return (float)(splitmix(s) >> 40) * (1.0f / 16777216.0f); }

extern "C" int ctrl_math_check(int kind, long long n, unsigned long long seed, double* out) {
  uint64_t s = seed;
  long long bad = 0, outside = 0;
  double worst = 0.0;
  for (long long it = 0; it < n; ++it) {
    double fast, ref;
    if (kind == 0) {            // pow: ramp positions as the ADSR produces them (uniform, near 0, near 1), alpha in [0.1, 6]
      float x = u01(s);
      const unsigned mode = (unsigned)(splitmix(s) & 3);
      if (mode == 1) x = ldexpf(x, -(int)(splitmix(s) % 20));           // small ramps
      if (mode == 2) x = 1.0f - ldexpf(x, -(int)(splitmix(s) % 24));    // 1 - small
      const float a = 0.1f + 5.9f * u01(s);
      if (!ias_ctl_pow_in_domain(x, a)) { ++outside; continue; }
      fast = ias_ctl_pow(x, a, g_tab);
      ref = pow((double)x, (double)a);
    } else if (kind == 1) {     // pow over the whole guarded domain: any normal x < 1, a in [2^-6, 64]
      union { float f; uint32_t u; } b;
      b.u = (uint32_t)(splitmix(s) % (0x3f800000u - 0x00800000u)) + 0x00800000u;
      const float x = b.f;
      const float a = ldexpf(1.0f + u01(s), (int)(splitmix(s) % 12) - 6);
      if (!ias_ctl_pow_in_domain(x, a)) { ++outside; continue; }
      fast = ias_ctl_pow(x, a, g_tab);
      ref = pow((double)x, (double)a);
    } else if (kind == 2) {     // cos: LFO arguments (phase + pi), up to ~2^15
      const unsigned mode = (unsigned)(splitmix(s) & 3);
      float x = u01(s) * (mode == 0 ? 8.0f : (mode == 1 ? 1100.0f : 32000.0f));
      if (mode == 3) x = (float)((double)(splitmix(s) % 20000) * 1.5707963267948966);   // next to the zeros / extrema
      if (splitmix(s) & 1) x = -x;
      if (!ias_ctl_cos_in_domain(x)) { ++outside; continue; }
      fast = ias_ctl_cos(x);
      ref = cos((double)x);
    } else if (kind == 4) {     // log2 as one value: uniform, next to 1 (both sides), tiny
      const unsigned mode = (unsigned)(splitmix(s) & 3);
      float x = u01(s);
      if (mode == 1) x = 1.0f - ldexpf(x, -(int)(splitmix(s) % 24));
      if (mode == 2) x = 1.0f + ldexpf(x, -(int)(splitmix(s) % 23));
      if (mode == 3) x = ldexpf(x + 0.5f, (int)(splitmix(s) % 250) - 140);
      if (!(x > 0.0f)) { ++outside; continue; }
      fast = ias_ctl_log2_value(x, g_tab);
      ref = log2((double)x);
      if (x == 1.0f) { if (fast != 0.0) ++bad; continue; }
    } else if (kind == 5) {     // log10 of a frequency in Hz
      const float x = 8.0f + 26000.0f * u01(s);
      fast = ias_ctl_log10_value(x, g_tab);
      ref = log10((double)x);
    } else if (kind == 6) {     // 2^v for an fp32 v (parameter curves: v = log2(u) / curve <= 0; some positive too)
      const float v = (u01(s) - 0.9f) * ldexpf(1.0f, (int)(splitmix(s) % 10) - 2);
      fast = ias_ctl_exp2((double)v, 0.0, g_tab);
      ref = exp2((double)v);
    } else if (kind == 7) {     // pow outside (0, 1): denormal x, x > 1, tiny and large exponents
      union { float f; uint32_t u; } b;
      b.u = (uint32_t)(splitmix(s) % 0x7f000000u) + 1u;
      const float x = b.f;
      const float a = ldexpf(1.0f + u01(s), (int)(splitmix(s) % 16) - 9) * ((splitmix(s) & 1) ? 1.0f : -1.0f);
      fast = ias_ctl_pow(x, a, g_tab);
      ref = pow((double)x, (double)a);
      if (!(fabs(ref) < 1.0e290) || fabs(ref) < 1.0e-290) { if ((float)fast != (float)ref) ++bad; continue; }
    } else {                    // fmod by fl32(2 pi)
      const float b = 6.2831854820251465f;
      const unsigned mode = (unsigned)(splitmix(s) & 3);
      float a = u01(s) * (mode == 0 ? 7.0f : (mode == 1 ? 1100.0f : 1.0e6f));
      if (mode == 3) a = (float)((double)(splitmix(s) % 4000) * (double)b);             // next to the multiples
      if ((splitmix(s) & 7) == 0) a = -a;
      if (!ias_ctl_fmod_in_domain(a, b)) { ++outside; continue; }
      fast = (double)ias_ctl_fmod(a, b, 1.0 / (double)b);
      ref = (double)fmodf(a, b);
      if (fast != ref) ++bad;
      continue;
    }
    if ((float)fast != (float)ref) ++bad;
    if (fabs(ref) > 1.0e-290) {                  // (relative error is meaningless next to fp64 underflow: fp32 result 0 either way)
      const double rel = fabs(fast - ref) / fabs(ref) * 9007199254740992.0;
      if (rel > worst) worst = rel;
    }
  }
  out[0] = (double)n; out[1] = (double)bad; out[2] = worst; out[3] = (double)outside;
  return 0;
}

// the fp64-argument forms (control-rate backward): max relative error against libm in units of 2^-53; out[1] counts
// results that are off by more than 2^-44 relative (sincos: absolute)
extern "C" int ctrl_math_check_d(int kind, long long n, unsigned long long seed, double* out) {
  uint64_t s = seed;
  long long bad = 0;
  double worst = 0.0;
  for (long long it = 0; it < n; ++it) {
    const double u = (double)(splitmix(s) >> 11) * (1.0 / 9007199254740992.0);
    double err;
    if (kind == 0) {            // pow (and ln) on (0, 1] x [0.1, 6]
      const unsigned mode = (unsigned)(splitmix(s) & 3);
      double x = u;
      if (mode == 1) x = ldexp(u, -(int)(splitmix(s) % 60));
      if (mode == 2) x = 1.0 - ldexp(u, -(int)(splitmix(s) % 50));
      if (!(x >= 1e-300)) x = 1e-300;
      const double a = 0.1 + 5.9 * (double)u01(s);
      double ln_x;
      const double fast = ias_ctl_pow_d(x, a, g_tab, &ln_x), ref = pow(x, a);
      err = ref > 1e-290 ? fabs(fast - ref) / ref : 0.0;
      const double lref = log(x);
      if (x != 1.0) err = fmax(err, fmin(fabs(ln_x - lref) / fabs(lref), fabs(ln_x - lref) * 1e3));
    } else if (kind == 1) {     // sin / cos up to 2^20
      const double x = (u - 0.5) * ldexp(1.0, (int)(splitmix(s) % 21));
      double sn, cs;
      ias_ctl_sincos_d(x, sn, cs);
      err = fmax(fabs(sn - sin(x)), fabs(cs - cos(x)));
    } else {                    // mod 2 pi
      const double b = 6.283185307179586, x = (u - 0.3) * ldexp(1.0, (int)(splitmix(s) % 21));
      double ref = fmod(x, b);
      if (ref < 0.0) ref += b;
      const double m = ias_ctl_mod_d(x, b, 1.0 / b);
      err = fmin(fabs(m - ref), b - fabs(m - ref)) / b;
      if (!(m >= 0.0 && m < b)) err = 1.0;
    }
    const double rel = err * 9007199254740992.0;
    if (rel > worst) worst = rel;
    if (rel > 512.0) ++bad;
  }
  out[0] = (double)n; out[1] = (double)bad; out[2] = worst; out[3] = 0.0;
  return 0;
}
