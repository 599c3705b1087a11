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
//   ias_ctl_pow(x, a)   x > 0 finite fp32, a finite fp32:  x^a in ~40 fp64-rate instructions
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

// log2 x = hi + lo for a positive finite fp32 x (denormals included): hi = e + thi_i EXACT (<= 29 bits, a multiple of
// 2^-22), lo = tlo_i + log2(1 + r) with an absolute error < 2^-60.  tab = the IAS_CTL_TAB_INIT table (LDS or global).
IAS_HD void ias_ctl_log2(float x, const double* tab, double& hi, double& lo) {
  union { float f; uint32_t u; } b;
  b.f = x;
  int e = -127;
  if (b.u < 0x00800000u) { b.f = x * 18446744073709551616.0f; e = -127 - 64; }   // denormal: x 2^64 is exact
  e += (int)(b.u >> 23);
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
  lo = fma(p, r, row[2]);                                      // tlo + log2(1 + r)
  hi = (double)e + row[1];                                     // exact
}

// 2^(y_hi + y_lo), |y_lo| <= 2^-3 |y_hi| or so (a two-sum sorts it out); saturates to 0 / inf far outside fp32's range
IAS_HD double ias_ctl_exp2(double y_hi, double y_lo, const double* tab) {
  const double yh = y_hi + y_lo;                               // two-sum: yh + yl == y_hi + y_lo
  const double yl = y_lo - (yh - y_hi);
  if (!(yh > -2000.0)) return yh != yh ? yh : 0.0;
  if (yh > 2000.0) return INFINITY;
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

// x^a for a positive finite fp32 x and any finite fp32 a (a (e + thi) is exact for every 24-bit a)
IAS_HD double ias_ctl_pow(float x, float a, const double* tab) {
  double hi, lo;
  ias_ctl_log2(x, tab, hi, lo);
  const double ad = (double)a;
  return ias_ctl_exp2(ad * hi, ad * lo, tab);                  // ad * hi exact (24 + 29 bits)
}

// log2 x as ONE double with a RELATIVE error ~2^-52 for every positive finite fp32 x: next to x = 1, where hi + lo would
// cancel, the series in d = x - 1 (exact) instead.  x == 1 gives exactly 0.
IAS_HD double ias_ctl_log2_value(float x, const double* tab) {
  const float d = x - 1.0f;                                    // exact for x in [0.5, 2]
  if (fabsf(d) < 0.0078125f) {
    const double dd = (double)d;
    double p = IAS_CTL_L9;
    p = fma(p, dd, IAS_CTL_L8);
    p = fma(p, dd, IAS_CTL_L7);
    p = fma(p, dd, IAS_CTL_L6);
    p = fma(p, dd, IAS_CTL_L5);
    p = fma(p, dd, IAS_CTL_L4);
    p = fma(p, dd, IAS_CTL_L3);
    p = fma(p, dd, IAS_CTL_L2);
    p = fma(p, dd, IAS_CTL_L1);
    return p * dd;
  }
  double hi, lo;
  ias_ctl_log2(x, tab, hi, lo);
  return hi + lo;
}
// log10 x = log2 x * log10(2), same accuracy class (x not next to 1: the render's only use is a frequency in Hz)
IAS_HD double ias_ctl_log10_value(float x, const double* tab) {
  double hi, lo;
  ias_ctl_log2(x, tab, hi, lo);
  return fma(hi, IAS_CTL_LOG10_2_HI, fma(lo, IAS_CTL_LOG10_2_HI, hi * IAS_CTL_LOG10_2_LO));
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
  // cos(r + k pi/2): k mod 4 = 0: cos r, 1: -sin r, 2: -cos r, 3: sin r.  ONE Horner chain with the coefficient set picked
  // per lane (sin r = r + (r z) S(z), cos r = (1 - z/2) + (z z) C(z)): half the live registers of evaluating both.
  const bool odd = (k & 1) != 0;
  double p = odd ? IAS_CTL_S7 : IAS_CTL_C8;
  p = fma(p, z, odd ? IAS_CTL_S6 : IAS_CTL_C7);
  p = fma(p, z, odd ? IAS_CTL_S5 : IAS_CTL_C6);
  p = fma(p, z, odd ? IAS_CTL_S4 : IAS_CTL_C5);
  p = fma(p, z, odd ? IAS_CTL_S3 : IAS_CTL_C4);
  p = fma(p, z, odd ? IAS_CTL_S2 : IAS_CTL_C3);
  p = fma(p, z, odd ? IAS_CTL_S1 : IAS_CTL_C2);
  const double lead = odd ? r : fma(z, -0.5, 1.0);
  const double v = fma(p * z, odd ? r : z, lead);
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

// ---- the same kernels for fp64 ARGUMENTS (the control-rate backward, csrc/voice_ctrl_grad_kernels.hip, is fp64 throughout:
// gradients are checked to 1e-5, not bit for bit).  r = m c_i - 1 and a (e + thi) are rounded here, not exact: relative
// error ~2^-50 instead of ~2^-52.
// log2 x = hi + lo for a positive NORMAL double
IAS_HD void ias_ctl_log2_d(double x, const double* tab, double& hi, double& lo) {
  union { double f; uint64_t u; } b;
  b.f = x;
  const int e = (int)(b.u >> 52) - 1023;
  const int i = (int)((b.u >> 45) & 127u);
  b.u = (b.u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
  const double* row = tab + 4 * i;
  const double r = fma(b.f, row[0], -1.0);
  double p = IAS_CTL_L7;
  p = fma(p, r, IAS_CTL_L6);
  p = fma(p, r, IAS_CTL_L5);
  p = fma(p, r, IAS_CTL_L4);
  p = fma(p, r, IAS_CTL_L3);
  p = fma(p, r, IAS_CTL_L2);
  p = fma(p, r, IAS_CTL_L1);
  lo = fma(p, r, row[2]);
  hi = (double)e + row[1];
}
// x^a, x a positive normal double, a finite; ln_x (optional) receives ln x from the same logarithm
IAS_HD double ias_ctl_pow_d(double x, double a, const double* tab, double* ln_x = nullptr) {
  double hi, lo;
  ias_ctl_log2_d(x, tab, hi, lo);
  if (ln_x != nullptr) *ln_x = (hi + lo) * 0.6931471805599453;
  const double y_hi = a * hi;
  const double y_lo = fma(a, hi, -y_hi) + a * lo;
  return ias_ctl_exp2(y_hi, y_lo, tab);
}
IAS_HD double ias_ctl_log_d(double x, const double* tab) {
  double hi, lo;
  ias_ctl_log2_d(x, tab, hi, lo);
  return (hi + lo) * 0.6931471805599453;
}
// sin x and cos x for |x| < 2^20 (absolute error ~2^-52 max(1, |x| 2^-20))
IAS_HD void ias_ctl_sincos_d(double x, double& sn, double& cs) {
  const double kd = rint(x * IAS_CTL_2OPI);
  double r = fma(-kd, IAS_CTL_PIO2_1, x);
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
  s = fma(s * z, r, r);
  double c = IAS_CTL_C8;
  c = fma(c, z, IAS_CTL_C7);
  c = fma(c, z, IAS_CTL_C6);
  c = fma(c, z, IAS_CTL_C5);
  c = fma(c, z, IAS_CTL_C4);
  c = fma(c, z, IAS_CTL_C3);
  c = fma(c, z, IAS_CTL_C2);
  c = fma(c * z, z, fma(z, -0.5, 1.0));
  // (sin, cos)(r + k pi/2)
  const double s0 = (k & 1) ? c : s, c0 = (k & 1) ? s : c;
  sn = (k & 2) ? -s0 : s0;
  cs = ((k + 1) & 2) ? -c0 : c0;
}
// x mod b in [0, b) for b > 0, |x| < 2^30 b (not exact: one rounding of x - q b, as fmod's result would be after the += b)
IAS_HD double ias_ctl_mod_d(double x, double b, double binv) {
  const double q = floor(x * binv);
  double m = fma(-q, b, x);
  if (m < 0.0) m += b;
  else if (m >= b) m -= b;
  return m;
}
