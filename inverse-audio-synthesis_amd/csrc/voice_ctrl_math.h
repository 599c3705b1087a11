// fp64 pow / cos / fmod of the control-rate pass, written out (MI355X: a wave64 fp64 FMA costs the SIMD 4 clocks).
//
// The control-rate arithmetic contract (voice_math.h, oracle/synth_oracle.py math "cr") defines the transcendentals of
// the ADSR ramps and the LFO shapes as "evaluated in fp64, rounded once to fp32".  Rounds 1-4 called the device math
// library for them -- pow() alone is ~300 fp64-rate instructions (it delivers < 1 ulp of fp64 for EVERY double), and the
// control pass of a B = 128 batch evaluates 1.3 M of them: 15 us of the whole chip per step (round 5 measurement: the
// headline step with and without the control pass, scripts/diag/run_noctrl_ab.sh).  What the contract needs is much less:
// fp32 inputs in a narrow domain, and an fp64 result good to ~2^-52 relative -- the accuracy class of the library calls
// themselves (the value rounded to fp32 then equals the rounding of the exact result unless the exact result lies within
// ~2^-52 of an fp32 rounding boundary: probability ~2^-28 per evaluation, for these functions as for the library's).
//
//   ias_ctl_pow(x, a)   0 < x < 1 (normal fp32), 2^-6 <= a <= 64:  x^a in ~40 fp64-rate instructions
//       log2 x = e + thi_i + (tlo_i + log2(1 + r)),  r = m c_i - 1 EXACT (c_i: 29 bits), |r| < 2^-8, degree-7 polynomial;
//       e + thi_i has <= 29 bits, so y_hi = a (e + thi_i) is EXACT; y_lo = a (tlo_i + P) carries an error < 2^-57;
//       2^y = 2^n E_j (1 + q(g)), y_hi + y_lo = n + j/128 + g by a two-sum (no bits of y_lo lost), degree-6 polynomial.
//   ias_ctl_cos(x)      |x| < 2^15: three-part Cody-Waite reduction by pi/2 (k P1, k P2 exact), Taylor kernels on
//                       [-pi/4, pi/4]; relative error ~2^-52 including at the zeros of the cosine.
//   ias_ctl_fmod(a, b)  |a| < 2^28 b, 1e-3 <= b <= 1e3: EXACT, like fmodf (quotient by reciprocal, exact fp64 remainder, one fix-up).
// Outside those domains the callers take the library function (a wave-uniform branch that is never taken for parameters
// in range).  The same code compiles for the host: tests/test_voice_math_cpu.py compares it with libm on 10^7 inputs per
// function (identical fp32 results required), and -- through the whole control pass -- with the oracle, bit for bit.
#pragma once
#include "ias_common.h"
#include "voice_ctrl_tables.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

// rows of 4 doubles: c_i, thi_i, tlo_i, E_i
#define IAS_CTL_TAB_DOUBLES (4 * IAS_CTL_N)

IAS_HD bool ias_ctl_pow_in_domain(float x, float a) {
  return x >= 1.17549435e-38f && x < 1.0f && a >= 0.015625f && a <= 64.0f;
}

// x^a for (x, a) in the domain above; tab = the IAS_CTL_TAB_INIT table (LDS on the device)
IAS_HD double ias_ctl_pow(float x, float a, const double* tab) {
  union { float f; uint32_t u; } b;
  b.f = x;
  const int e = (int)(b.u >> 23) - 127;
  const int i = (int)((b.u >> 16) & 127u);
  b.u = (b.u & 0x007fffffu) | 0x3f800000u;
  const double m = (double)b.f;
  const double* row = tab + 4 * i;
  const double r = fma(m, row[0], -1.0);                       // exact
  double p = IAS_CTL_L7;
  p = fma(p, r, IAS_CTL_L6);
  p = fma(p, r, IAS_CTL_L5);
  p = fma(p, r, IAS_CTL_L4);
  p = fma(p, r, IAS_CTL_L3);
  p = fma(p, r, IAS_CTL_L2);
  p = fma(p, r, IAS_CTL_L1);
  const double s_lo = fma(p, r, row[2]);                       // tlo + log2(1 + r)
  const double s_hi = (double)e + row[1];                      // exact (<= 29 bits)
  const double ad = (double)a;
  const double y_hi = ad * s_hi;                               // exact (24 + 29 bits)
  const double y_lo = ad * s_lo;
  const double yh = y_hi + y_lo;                               // two-sum: yh + yl == y_hi + y_lo
  const double yl = y_lo - (yh - y_hi);
  const double kd = rint(yh * 128.0);
  const double g = fma(kd, -0.0078125, yh) + yl;               // (first term exact) |g| <= 2^-8
  const int k = (int)kd;
  double q = IAS_CTL_X6;
  q = fma(q, g, IAS_CTL_X5);
  q = fma(q, g, IAS_CTL_X4);
  q = fma(q, g, IAS_CTL_X3);
  q = fma(q, g, IAS_CTL_X2);
  q = fma(q, g, IAS_CTL_X1);
  q = q * g;
  const double ej = tab[4 * (k & 127) + 3];
  return ldexp(fma(ej, q, ej), k >> 7);
}

IAS_HD bool ias_ctl_cos_in_domain(float x) { return fabsf(x) < 32768.0f; }

// cos(x), |x| < 2^15
IAS_HD double ias_ctl_cos(float x) {
  const double xd = (double)x;
  const double kd = rint(xd * IAS_CTL_2OPI);
  double r = fma(-kd, IAS_CTL_PIO2_1, xd);                     // exact
  r = fma(-kd, IAS_CTL_PIO2_2, r);
  r = fma(-kd, IAS_CTL_PIO2_3, r);
  const int k = (int)kd;
  const double z = r * r;
  double s = IAS_CTL_S7;
  s = fma(s, z, IAS_CTL_S6);
  s = fma(s, z, IAS_CTL_S5);
  s = fma(s, z, IAS_CTL_S4);
  s = fma(s, z, IAS_CTL_S3);
  s = fma(s, z, IAS_CTL_S2);
  s = fma(s, z, IAS_CTL_S1);
  s = fma(s * z, r, r);                                        // sin r = r + r^3 S(r^2)
  double c = IAS_CTL_C8;
  c = fma(c, z, IAS_CTL_C7);
  c = fma(c, z, IAS_CTL_C6);
  c = fma(c, z, IAS_CTL_C5);
  c = fma(c, z, IAS_CTL_C4);
  c = fma(c, z, IAS_CTL_C3);
  c = fma(c, z, IAS_CTL_C2);
  c = fma(c * z, z, fma(z, -0.5, 1.0));                        // cos r = 1 - r^2/2 + r^4 C(r^2)
  // cos(r + k pi/2): k mod 4 = 0: cos r, 1: -sin r, 2: -cos r, 3: sin r
  const double v = (k & 1) ? s : c;
  return (((k + 1) >> 1) & 1) ? -v : v;
}

IAS_HD bool ias_ctl_fmod_in_domain(float a, float b) { return b >= 1.0e-3f && b <= 1.0e3f && fabsf(a) < 268435456.0f * b; }

// fmodf(a, b) exactly (sign of a), for the domain above; binv = 1 / (double)b
IAS_HD float ias_ctl_fmod(float a, float b, double binv) {
  const double am = fabs((double)a), bd = (double)b;
  double q = trunc(am * binv);                                 // floor(am / bd) or one off
  double rem = fma(-q, bd, am);                                // exact: q < 2^28, bd 24 bits, the difference fits 53 bits
  if (rem < 0.0) rem += bd;                                    // exact
  else if (rem >= bd) rem -= bd;
  const float m = (float)rem;                                  // exact: a remainder of two floats is a float
  return a < 0.0f ? -m : m;
}
