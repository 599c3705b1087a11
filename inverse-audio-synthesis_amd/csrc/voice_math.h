// Per-sample arithmetic of the Voice render, shared by the HIP kernels (device)
// and by a host-only test build (tests/ compile this header with g++ to check
// the arithmetic against oracle/synth_oracle.py without a GPU).
//
// Semantics: the restatement of torchsynth's Voice.output() documented in
// oracle/synth_oracle.py, math mode "cr": every fp32 elementary operation is
// correctly rounded (transcendentals are evaluated in fp64 and rounded once),
// the small dot products / sums on the oscillator-phase path (LFO shape mix, LFO mode
// normalisation, mod matrix) are fp64-accumulated and rounded once, the linear
// upsample is fl(fl(w0*a)+fl(w1*b)), the final mixer a left-to-right mul/add chain.
// Reference call sites: /root/reference/vicreg_audio_params.py:86-94,114,
// /root/reference/audio_to_params.py:215,240-257.
#pragma once
#include "ias_common.h"
#include "voice_exp2_table.h"
#include "voice_ctrl_math.h"
#if !defined(__HIPCC__)
#include <algorithm>
using std::min;
using std::max;
#endif

#define IAS_PI_D 3.141592653589793
#define IAS_TWO_PI_D 6.283185307179586

// ---- exactly-rounded fp32 primitives (no contraction, no reassociation) ----
// Plain operators under "fp contract(off)": every * and + below is one correctly rounded IEEE operation
// and can never be fused into an FMA, on the host and on the device alike.  (The HIP __fmul_rn-style
// intrinsics give the same values but are opaque calls that keep the compiler from packing two
// samples into v_pk_mul_f32 / v_pk_add_f32.)  The translation units that include this header are also
// compiled with -ffp-contract=off.
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
IAS_HD float ias_mul(float a, float b) { return a * b; }
IAS_HD float ias_add(float a, float b) { return a + b; }
IAS_HD float ias_sub(float a, float b) { return a - b; }
IAS_HD float ias_div(float a, float b) { return a / b; }
IAS_HD float ias_fma(float a, float b, float c) { return fmaf(a, b, c); }

// ---- correctly rounded fp32 transcendentals (fp64 evaluate, round once) ----
IAS_HD float ias_pow_cr(float x, float a) { return (float)pow((double)x, (double)a); }
IAS_HD float ias_log2_cr(float x) { return (float)log2((double)x); }
IAS_HD float ias_log10_cr(float x) { return (float)log10((double)x); }
IAS_HD float ias_cos_cr(float x) { return (float)cos((double)x); }
IAS_HD float ias_exp2_slow_cr(float x) { return (float)exp2((double)x); }
// The same values from the written-out fp64 kernels of voice_ctrl_math.h (ctl: the IAS_CTL_TAB_INIT table, or NULL for
// the library calls above): ~40 fp64-rate instructions instead of ~300 for pow, ~30 instead of ~120 for cos.  Same
// accuracy class as the library calls (voice_ctrl_math.h), so the same fp32 values.
// A translation unit compiled with IAS_CTL_NO_LIBM (csrc/voice_ctrl_kernels.hip: the control pass that has to fit beside
// the render's waves, <= 56 VGPRs) never reaches the library: ctl must be given, pow / log2 / exp2 are total on their own
// (x <= 0 as IEEE pow / log2 define it), and the launcher only takes those kernels where every LFO phase is far below the
// 2^20 rad up to which cos / fmod reduce exactly.
#ifdef IAS_CTL_NO_LIBM
#define IAS_CTL_LIBM_IF(cond) if (false)
#else
#define IAS_CTL_LIBM_IF(cond) if (cond)
#endif
IAS_HD float ias_pow_ctl(float x, float a, const double* ctl) {
  IAS_CTL_LIBM_IF(ctl == nullptr || !ias_ctl_pow_in_domain(x, a)) return ias_pow_cr(x, a);
  if (!(x > 0.0f)) return x == 0.0f ? (a > 0.0f ? 0.0f : (a == 0.0f ? 1.0f : INFINITY)) : NAN;
  return (float)ias_ctl_pow(x, a, ctl);
}
IAS_HD float ias_cos_ctl(float x, const double* ctl) {
  IAS_CTL_LIBM_IF(ctl == nullptr || !ias_ctl_cos_in_domain(x)) return ias_cos_cr(x);
  return (float)ias_ctl_cos(x);
}
IAS_HD float ias_log2_ctl(float x, const double* ctl) {
  IAS_CTL_LIBM_IF(ctl == nullptr || !(x > 0.0f) || !(x < INFINITY)) return ias_log2_cr(x);
  if (!(x > 0.0f)) return x == 0.0f ? -INFINITY : NAN;
  return (float)ias_ctl_log2_value(x, ctl);
}
IAS_HD float ias_log10_ctl(float x, const double* ctl) {
  IAS_CTL_LIBM_IF(ctl == nullptr || !(x > 0.0f) || !(x < INFINITY) || fabsf(x - 1.0f) < 0.25f) return ias_log10_cr(x);
  return (float)ias_ctl_log10_value(x, ctl);
}
IAS_HD float ias_exp2_ctl(float v, const double* ctl) {
  IAS_CTL_LIBM_IF(ctl == nullptr) return ias_exp2_slow_cr(v);
  return (float)ias_ctl_exp2((double)v, 0.0, ctl);
}

// 2^t for the audio-rate pitch path.  n = rint(t), f = t - n is exact in fp32;
// 2^f by a degree-13 Taylor polynomial in fp64 (|f| <= 0.5: truncation < 2e-17
// relative), scaled by 2^n, rounded once to fp32.
IAS_HD float ias_exp2_cr(float t) {
  if (!(t > -160.0f)) return (t != t) ? t : 0.0f;
  if (t > 130.0f) return INFINITY;
  const float n = rintf(t);
  const double z = (double)(t - n) * 0.6931471805599453;
  double p = 1.6059043836821613e-10;           // 1/13!
  p = fma(p, z, 2.08767569878681e-09);         // 1/12!
  p = fma(p, z, 2.505210838544172e-08);        // 1/11!
  p = fma(p, z, 2.755731922398589e-07);        // 1/10!
  p = fma(p, z, 2.7557319223985893e-06);       // 1/9!
  p = fma(p, z, 2.48015873015873e-05);         // 1/8!
  p = fma(p, z, 0.0001984126984126984);        // 1/7!
  p = fma(p, z, 0.001388888888888889);         // 1/6!
  p = fma(p, z, 0.008333333333333333);         // 1/5!
  p = fma(p, z, 0.041666666666666664);         // 1/4!
  p = fma(p, z, 0.16666666666666666);          // 1/3!
  p = fma(p, z, 0.5);
  p = fma(p, z, 1.0);
  p = fma(p, z, 1.0);
  return (float)ldexp(p, (int)n);
}

// Same value as ias_exp2_cr for |t| <= 64 (the pitch path has t in [-5.75, 4.84]): degree-10
// near-minimax polynomial for 2^f on [-0.5, 0.5] (max relative error 3.2e-16 before the final
// rounding; coefficients from a Chebyshev fit in 80-bit arithmetic, scripts in DESIGN.md).
IAS_HD float ias_exp2_cr_fast(float t) {
  const float n = rintf(t);
  const double f = (double)(t - n);
  double p = 7.072587226569927e-09;
  p = fma(p, f, 1.0208690586037056e-07);
  p = fma(p, f, 1.3215442576793237e-06);
  p = fma(p, f, 1.5252657258547322e-05);
  p = fma(p, f, 0.00015403530441763784);
  p = fma(p, f, 0.001333355823016814);
  p = fma(p, f, 0.0096181291076068692);
  p = fma(p, f, 0.055504108664447695);
  p = fma(p, f, 0.24022650695910097);
  p = fma(p, f, 0.69314718055994995);
  p = fma(p, f, 1.0);
  return (float)ldexp(p, (int)n);
}

// Same value as ias_exp2_cr for t in [(IAS_EXP2_TAB_MIN + 1) / 256, (IAS_EXP2_TAB_MIN + LEN - 2) / 256] (the pitch path:
// t = (midi - 69) / 12, midi in [0, 127]): m = rint(256 t) by the 1.5 * 2^23 trick (the fma is exact up to the one
// rounding to an integer, ties to even as rintf), r = t - m / 256 exact in fp32 with |r| <= 1/512,
// 2^t = T[m] * (1 + r (c1 + r (c2 + r (c3 + r c4)))) in fp64 (Taylor truncation 4e-17, total error <= ~3.5e-16 before
// the single rounding to fp32), T = IAS_EXP2_TAB (fp64, correctly rounded; in LDS on the device).
// 6 fp64 operations instead of the 11 + ldexp + int conversion of ias_exp2_cr_fast.  (The render kernel runs the same
// arithmetic with the table in LDS: voice_exp2_cr_lds in voice_kernels.hip.)
IAS_HD float ias_exp2_cr_tab(float t, const double* tab) {
  const float u = fmaf(t, 256.0f, 12582912.0f);
  const float mf = u - 12582912.0f;
  const float r = fmaf(mf, -0.00390625f, t);
  union { float f; int32_t i; } b;
  b.f = u;
  const double tv = tab[b.i - (0x4B400000 + IAS_EXP2_TAB_MIN)];
  const double rd = (double)r;
  double p = IAS_EXP2_C4;
  p = fma(p, rd, IAS_EXP2_C3);
  p = fma(p, rd, IAS_EXP2_C2);
  p = fma(p, rd, IAS_EXP2_C1);
  p = fma(p, rd, 1.0);
  return (float)(p * tv);
}

// fl32(a / d) for a divisor d whose odd part is small (12, audio sample rates): the fp64 product
// with the rounded reciprocal, rounded once to fp32, equals the IEEE fp32 quotient because a
// quotient by such a d can never lie within 2^-53 (relative) of a rounding midpoint.
IAS_HD float ias_div_by_recip(float a, double recip) { return (float)((double)a * recip); }

// fl32(a / d) from three fp32 operations: q0 = fl(a r), e = a - d q0 (exact in the FMA), q = fl(q0 + e r), with
// r = fl32(1/d).  Correctly rounded for every input the render can produce, checked EXHAUSTIVELY on the host
// (scripts/diag/div_by_const_check.c: all fp32 a with 1e-30 <= |a| <= 128 for d = 12; all 1 <= a <= 1e6 for
// d in {16000, 22050, 32000, 44100, 48000, 96000}; a = 0 gives 0); it only fails where e underflows (|a/d| < 2^-126),
// far from the pitches (|c - 69| is 0 or >= 2^-18) and angular frequencies (>= 51 rad/s) of this path.
// 6 SIMD clocks instead of the 12 of convert / fp64 multiply / convert.
IAS_HD float ias_div_fma(float a, float d, float r) {
  const float q0 = a * r;
  const float e = fmaf(-d, q0, a);
  return fmaf(e, r, q0);
}
// sample rates for which ias_div_fma has been verified (others use ias_div_by_recip)
IAS_HD int ias_div_fma_rate_ok(int sample_rate) {
  return sample_rate == 16000 || sample_rate == 22050 || sample_rate == 32000 || sample_rate == 44100 ||
         sample_rate == 48000 || sample_rate == 96000;
}

// ---- parameter range mapping (torchsynth ModuleParameterRange.from_0to1) ----
// lo = fl32(minimum); span = fl32(maximum - minimum) (non-symmetric) or
// fl32((maximum - minimum) / 2) (symmetric), both rounded from the double table.
IAS_HD float ias_map_param(float u, float lo, float span, float curve, int symmetric, const double* ctl = nullptr) {
  if (!symmetric) {
    if (curve != 1.0f) u = ias_exp2_ctl(ias_div(ias_log2_ctl(u, ctl), curve), ctl);
    return ias_add(lo, ias_mul(span, u));
  }
  const float dist = ias_sub(ias_mul(2.0f, u), 1.0f);
  float v = dist;
  if (curve != 1.0f && dist != 0.0f) {
    const float mag = ias_exp2_ctl(ias_div(ias_log2_ctl(fabsf(dist), ctl), curve), ctl);
    v = dist < 0.0f ? -mag : mag;
  }
  return ias_add(lo, ias_mul(span, ias_add(v, 1.0f)));
}

// ---- ADSR (control rate) ----
// ramp(t) of torchsynth ADSR.ramp: t = control-sample index, durations in seconds.
IAS_HD float ias_ramp(int t, float duration, float start, int has_start, int inverse,
                      float alpha, float control_rate, float eps, const double* ctl = nullptr) {
  const float dur = ias_mul(duration, control_rate);
  float r = (float)t;
  if (has_start) r = ias_sub(r, ias_mul(start, control_rate));
  r = fmaxf(r, 0.0f);
  r = ias_add(ias_div(ias_add(r, eps), dur), eps);
  r = fminf(r, 1.0f);
  if (inverse && dur > 0.0f) r = ias_sub(1.0f, r);
  // saturated ramps: pow(1, a) = 1 and pow(0, a) = 0 exactly (alpha is in [0.1, 6])
  if (r == 1.0f || (r == 0.0f && alpha > 0.0f)) return r;
  return ias_pow_ctl(r, alpha, ctl);
}

struct IasAdsr { float attack, decay, sustain, release, alpha; };

// ias_ramp with the value of its flat head supplied: for t <= start*control_rate the clamped ramp
// position is 0, so the ramp (and its pow) is the same number for all those t -- `head` = ias_ramp at
// t = 0.  Bit-identical to ias_ramp; at most one of an ADSR's three ramps is outside its head /
// saturated region at any t, which cuts the fp64 pow() count by ~3x.
IAS_HD float ias_ramp_headed(int t, float duration, float start, int inverse, float alpha, float control_rate,
                             float eps, float head, const double* ctl = nullptr) {
  if (ias_sub((float)t, ias_mul(start, control_rate)) <= 0.0f) return head;
  return ias_ramp(t, duration, start, 1, inverse, alpha, control_rate, eps, ctl);
}

struct IasAdsrHeads { float decay_head, release_head; };
IAS_HD IasAdsrHeads ias_adsr_heads(const IasAdsr& e, float note_on, float control_rate, float eps,
                                   const double* ctl = nullptr) {
  const float new_attack = fminf(e.attack, note_on);
  const float new_decay = fminf(fmaxf(ias_sub(note_on, e.attack), 0.0f), e.decay);
  IasAdsrHeads h;
  h.decay_head = ias_ramp(0, new_decay, new_attack, 1, 1, e.alpha, control_rate, eps, ctl);
  h.release_head = ias_ramp(0, e.release, note_on, 1, 1, e.alpha, control_rate, eps, ctl);
  return h;
}
IAS_HD float ias_adsr_headed(int t, const IasAdsr& e, float note_on, float control_rate, float eps,
                             const IasAdsrHeads& h, const double* ctl = nullptr) {
  const float new_attack = fminf(e.attack, note_on);
  const float new_decay = fminf(fmaxf(ias_sub(note_on, e.attack), 0.0f), e.decay);
  const float a = ias_ramp(t, new_attack, 0.0f, 0, 0, e.alpha, control_rate, eps, ctl);
  const float dr = ias_ramp_headed(t, new_decay, new_attack, 1, e.alpha, control_rate, eps, h.decay_head, ctl);
  const float d = ias_add(ias_mul(ias_sub(1.0f, e.sustain), dr), e.sustain);
  const float r = ias_ramp_headed(t, e.release, note_on, 1, e.alpha, control_rate, eps, h.release_head, ctl);
  return ias_mul(ias_mul(a, d), r);
}

IAS_HD float ias_adsr(int t, const IasAdsr& e, float note_on, float control_rate, float eps, const double* ctl = nullptr) {
  const float new_attack = fminf(e.attack, note_on);
  const float new_decay = fminf(fmaxf(ias_sub(note_on, e.attack), 0.0f), e.decay);
  const float a = ias_ramp(t, new_attack, 0.0f, 0, 0, e.alpha, control_rate, eps, ctl);
  const float dr = ias_ramp(t, new_decay, new_attack, 1, 1, e.alpha, control_rate, eps, ctl);
  const float d = ias_add(ias_mul(ias_sub(1.0f, e.sustain), dr), e.sustain);
  const float r = ias_ramp(t, e.release, note_on, 1, 1, e.alpha, control_rate, eps, ctl);
  return ias_mul(ias_mul(a, d), r);
}

// ---- LFO (control rate) ----
IAS_HD float ias_lfo_inc(float freq, float depth, float rate_env, float control_rate) {
  const float fr = fmaxf(ias_add(freq, ias_mul(depth, rate_env)), 0.0f);
  return ias_div(ias_mul((float)IAS_TWO_PI_D, fr), control_rate);
}

// torch.remainder(a, b) for b > 0
IAS_HD float ias_remainder(float a, float b) {
  float m = fmodf(a, b);
  if (m != 0.0f && m < 0.0f) m = ias_add(m, b);
  return m;
}

// the same by voice_ctrl_math.h's exact fp64 remainder (b = fl32(2 pi) only; other arguments: fmodf)
IAS_HD float ias_remainder_2pi_ctl(float a, const double* ctl) {
  const float b = (float)IAS_TWO_PI_D;
  float m;
  m = ias_ctl_fmod(a, b, 1.0 / (double)(float)IAS_TWO_PI_D);
  IAS_CTL_LIBM_IF(ctl == nullptr || !ias_ctl_fmod_in_domain(a, b)) m = fmodf(a, b);
  if (m != 0.0f && m < 0.0f) m = ias_add(m, b);
  return m;
}

// arg = fl(fl32(cumsum_double(inc)) + phi0); mode[5] already normalised.
IAS_HD float ias_lfo_shape_mix(float arg, const float* mode, const double* ctl = nullptr) {
  const float two_pi = (float)IAS_TWO_PI_D;
  float c = ias_cos_ctl(ias_add(arg, (float)IAS_PI_D), ctl);
  float sq = (c > 0.0f) ? 1.0f : ((c < 0.0f) ? -1.0f : 0.0f);
  c = ias_div(ias_add(c, 1.0f), 2.0f);
  sq = ias_div(ias_add(sq, 1.0f), 2.0f);
  const float saw = ias_div(ias_remainder_2pi_ctl(arg, ctl), two_pi);
  const float rsaw = ias_sub(1.0f, saw);
  float tri = ias_mul(2.0f, saw);
  if (tri > 1.0f) tri = ias_sub(2.0f, tri);
  double o = (double)mode[0] * (double)c;   // products of two floats are exact in fp64
  o += (double)mode[1] * (double)tri;
  o += (double)mode[2] * (double)saw;
  o += (double)mode[3] * (double)rsaw;
  o += (double)mode[4] * (double)sq;
  return (float)o;
}

// mod-matrix row: sum_k w[k]*s[k], fp64 accumulate, rounded once
IAS_HD float ias_dot4_cr(float w0, float w1, float w2, float w3, float s0, float s1, float s2, float s3) {
  double o = (double)w0 * (double)s0;
  o += (double)w1 * (double)s1;
  o += (double)w2 * (double)s2;
  o += (double)w3 * (double)s3;
  return (float)o;
}

// torchsynth LFO: mode = pow(p, exponent) / sum with the LFO.__init__ default exponent (IAS_LFO_EXPONENT_F)
IAS_HD void ias_lfo_mode(const float* p5, float* mode, const double* ctl = nullptr) {
  float m[5];
  for (int k = 0; k < 5; ++k) m[k] = ias_pow_ctl(p5[k], IAS_LFO_EXPONENT_F, ctl);
  const float s = (float)((double)m[0] + (double)m[1] + (double)m[2] + (double)m[3] + (double)m[4]);
  for (int k = 0; k < 5; ++k) mode[k] = ias_div(m[k], s);
}

// ---- audio rate ----
// linear upsample with align_corners=True (torch CPU kernel arithmetic)
IAS_HD void ias_interp_pos(int j, float scale, int Tc, int& i0, int& i1, float& w0, float& w1) {
  const float real = ias_mul(scale, (float)j);
  int k = (int)floorf(real);
  if (k > Tc - 1) k = Tc - 1;
  i0 = k;
  i1 = k + (k < Tc - 1 ? 1 : 0);
  w1 = fminf(fmaxf(ias_sub(real, (float)k), 0.0f), 1.0f);
  w0 = ias_sub(1.0f, w1);
}
// Same values as ias_interp_pos without the clamps, which are provably inactive: 0 <= scale*j < Tc for
// every sample index j < T (scale = fl((Tc-1)/(T-1))), so floor = trunc, the index never exceeds Tc-1
// and real - floor(real) is already in [0, 1).
IAS_HD void ias_interp_pos_fast(int j, float scale, int Tc, int& i0, int& i1, float& w0, float& w1) {
  const float real = ias_mul(scale, (float)j);
  const int k = (int)real;
  i0 = k;
  i1 = min(k + 1, Tc - 1);
  w1 = ias_sub(real, (float)k);
  w0 = ias_sub(1.0f, w1);
}
// fl(w0 a + fl(w1 b)): ONE fused multiply-add on the rounded second product -- what torch's CPU kernel of
// nn.Upsample(mode="linear", align_corners=True) evaluates (ATen cpu_upsample_linear: `x0 * w0 + x1 * w1`, contracted
// by the compiler into fma(x0, w0, x1 * w1)); bit-equal to the torch op (tests/test_voice_math_cpu.py).  Rounds 1-3
// had the three-rounding form fl(fl(w0 a) + fl(w1 b)), which differs from the op in 24 % of the samples by one ulp.
IAS_HD float ias_lerp(float a, float b, float w0, float w1) { return fmaf(w0, a, ias_mul(w1, b)); }

IAS_HD float ias_midi_to_hz(float midi) {
  return ias_mul(440.0f, ias_exp2_cr(ias_div(ias_sub(midi, 69.0f), 12.0f)));
}

// phase increment of one VCO sample: fl(fl(2pi*hz)/sr)
IAS_HD float ias_vco_inc(float f0, float depth, float pitch_mod, float sample_rate) {
  float c = ias_add(f0, ias_mul(depth, pitch_mod));
  c = fminf(fmaxf(c, 0.0f), 127.0f);
  return ias_div(ias_mul((float)IAS_TWO_PI_D, ias_midi_to_hz(c)), sample_rate);
}

// bit-identical to ias_vco_inc, cheaper: no IEEE fp32 divisions, short exp2 polynomial.
IAS_HD float ias_vco_inc_fast(float f0, float depth, float pitch_mod, double inv_sample_rate) {
  float c = ias_add(f0, ias_mul(depth, pitch_mod));
  c = fminf(fmaxf(c, 0.0f), 127.0f);
  const float t = ias_div_by_recip(ias_sub(c, 69.0f), 1.0 / 12.0);
  const float hz = ias_mul(440.0f, ias_exp2_cr_fast(t));
  return ias_div_by_recip(ias_mul((float)IAS_TWO_PI_D, hz), inv_sample_rate);
}

IAS_HD float ias_partials_k(float midi_f0, float depth_2, const double* ctl = nullptr) {
  const float max_pitch = ias_add(midi_f0, fmaxf(depth_2, 0.0f));
  const float max_f0 = ias_midi_to_hz(max_pitch);
  const float partials = ias_div(12000.0f, ias_mul(max_f0, ias_log10_ctl(max_f0, ctl)));
  return ias_mul((float)IAS_PI_D, partials);
}

#if defined(__HIPCC__)
// sin/cos of an fp32 angle up to ~1e6 rad: revolutions in fp64 (error ~1e-11 rev), fractional part,
// gfx950 v_sin_f32 / v_cos_f32 (inputs in revolutions; measured max abs error 1.3e-7).
__device__ __forceinline__ void ias_sincos_dev(float a, float& s, float& c) {
  // The square-wave shaper multiplies sin by up to ~2500 before tanh, so sin needs RELATIVE accuracy at
  // its zero crossings.  Reduce (still in fp64) to a quarter revolution r2 in [-0.25, 0.25] around the
  // nearest zero crossing: sin(2 pi r) = (-1)^q sin(2 pi r2), r2 = r - q/2, q = rint(2 r); the fp32 r2
  // then has full relative precision where sin is small.
  const double v = (double)a * 0.15915494309189535;
  const double r = v - rint(v);
  const double q = rint(r + r);
  const float r2 = (float)fma(q, -0.5, r);
  const float sg = ((int)q & 1) ? -1.0f : 1.0f;
  s = sg * __builtin_amdgcn_sinf(r2);
  c = sg * __builtin_amdgcn_cosf(r2);
}
__device__ __forceinline__ float ias_cos_dev(float a) {
  const double v = (double)a * 0.15915494309189535;
  return __builtin_amdgcn_cosf((float)(v - rint(v)));
}
// tanh(z) = sign(z) (1 - e^{-2|z|}) / (1 + e^{-2|z|})   (v_exp_f32 + v_rcp_f32; max abs error 1.3e-7)
__device__ __forceinline__ float ias_tanh_dev(float z) {
  const float t = __builtin_amdgcn_exp2f(-2.885390081777927f * fabsf(z));
  const float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
  return z < 0.0f ? -r : r;
}
// ias_mix_sample with the device transcendentals (not amplified by the phase: ~1e-7 absolute).
__device__ __forceinline__ float ias_mix_sample_dev(float arg1, float arg2, float amp1, float amp2, float ampn,
                                                    float noise, const IasVoiceConst& vc) {
  float s2, c2;
  ias_sincos_dev(arg2, s2, c2);
  const float v1 = ias_mul(ias_cos_dev(arg1), amp1);
  const float sq = ias_tanh_dev(ias_mul(ias_mul(vc.kpart, s2), 0.5f));
  const float v2 = ias_mul(ias_mul(ias_mul(vc.shape_gain, sq), ias_add(1.0f, ias_mul(vc.shape, c2))), amp2);
  const float nz = ias_mul(noise, ampn);
  float o = ias_mul(vc.lvl0, v1);
  o = ias_add(o, ias_mul(vc.lvl1, v2));
  o = ias_add(o, ias_mul(vc.lvl2, nz));
  return o;
}

#endif

// unnormalised mixer output for one sample, given both phases (fp32, phi added).
IAS_HD float ias_mix_sample(float arg1, float arg2, float amp1, float amp2, float ampn,
                            float noise, const IasVoiceConst& vc) {
  const float v1 = ias_mul(cosf(arg1), amp1);
  const float sq = tanhf(ias_mul(ias_mul(vc.kpart, sinf(arg2)), 0.5f));
  const float v2 = ias_mul(ias_mul(ias_mul(vc.shape_gain, sq),
                                   ias_add(1.0f, ias_mul(vc.shape, cosf(arg2)))), amp2);
  const float nz = ias_mul(noise, ampn);
  float o = ias_mul(vc.lvl0, v1);
  o = ias_add(o, ias_mul(vc.lvl1, v2));
  o = ias_add(o, ias_mul(vc.lvl2, nz));
  return o;
}
