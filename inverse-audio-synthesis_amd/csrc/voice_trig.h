// Device transcendentals of the audio-rate Voice kernels (render: voice_kernels.hip, its adjoint: voice_grad_kernels.hip):
// exact fp32 split of a phase into revolutions + v_sin_f32 / v_cos_f32 / v_exp_f32 / v_rcp_f32.  gfx950 only.
#pragma once

// a / (2 pi) in revolutions as fp + t: fp = fract(fl(a * c_hi)) exactly, t = the exact rounding residual of that
// product + a * c_lo (c_hi + c_lo = 1 / (2 pi) to 2^-52 relative) -- the fp64 product of the first version in five
// fp32 instructions; |error| < 1e-9 revolutions for |a| <= 1e6 rad.
__device__ __forceinline__ void voice_rev_split(float a, float& fp, float& t) {
  const float c_hi = 0.15915493667125702f, c_lo = 6.4206382432985265e-09f;
  const float p = a * c_hi;
  const float e = fmaf(a, c_hi, -p);
  fp = __builtin_amdgcn_fractf(p);
  t = fmaf(a, c_lo, e);
}
__device__ __forceinline__ float voice_cos(float a) {
  float fp, t;
  voice_rev_split(a, fp, t);
  return __builtin_amdgcn_cosf(fp + t);
}
// sin and cos of the square-saw VCO's phase.  The shaper multiplies sin by up to ~2500 before tanh, so sin needs
// RELATIVE accuracy at its zero crossings: reduce to r2 in [-1/4, 1/4] around the nearest crossing (fp - q/2 is exact,
// the residual t is added last), sin(2 pi r) = (-1)^q sin(2 pi r2), q = rint(2 r) in {0, 1, 2}.
// -> sin(2 pi r2), cos(2 pi r2) and flip = (q == 1): the true values are both negated when flip is set (the caller
// folds the sign into its products: two selects instead of a sign factor and two multiplies).
__device__ __forceinline__ void voice_sincos(float a, float& s, float& c, bool& flip) {
  float fp, t;
  voice_rev_split(a, fp, t);
  const float r = fp + t;
  const float q = __builtin_rintf(r + r);
  const float r2 = fmaf(q, -0.5f, fp) + t;
  flip = (q == 1.0f);
  s = __builtin_amdgcn_sinf(r2);
  c = __builtin_amdgcn_cosf(r2);
}
// The render's form (round 5): the sign of both values as a FACTOR sgn = (-1)^q = 1 - 2 q (2 - q), q in {0, 1, 2} -- two
// multiply-adds -- instead of a compare whose flag the caller turns into two selects (12 clocks of quarter-rate
// instructions).  sin = sgn s, cos = sgn c.
__device__ __forceinline__ void voice_sincos_sgn(float a, float& s, float& c, float& sgn) {
  float fp, t;
  voice_rev_split(a, fp, t);
  const float r = fp + t;
  const float q = __builtin_rintf(r + r);
  const float r2 = fmaf(q, -0.5f, fp) + t;
  sgn = fmaf(q, fmaf(q, 2.0f, -4.0f), 1.0f);
  s = __builtin_amdgcn_sinf(r2);
  c = __builtin_amdgcn_cosf(r2);
}
// tanh(z / 2) = 2 / (1 + e^{-z}) - 1 for SIGNED z, given x = -z log2(e): one multiply-add behind v_exp_f32 + v_rcp_f32, no
// |z| / copysign pair around them (e^{-z} = inf gives 2 * 0 - 1 = -1, e^{-z} = 0 gives 1: no special cases).  Absolute
// error 1.2e-7 (the form below keeps RELATIVE accuracy near 0, which the mix -- held to 1e-4 absolute -- does not need).
__device__ __forceinline__ float voice_tanh_half_of_exp2arg(float x) {
  return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x)), -1.0f);
}
// |tanh(z)| = (1 - e^{-2|z|}) / (1 + e^{-2|z|})   (v_exp_f32 + v_rcp_f32; max abs error 1.3e-7)
__device__ __forceinline__ float voice_tanh_abs(float z) {
  const float t = __builtin_amdgcn_exp2f(-2.885390081777927f * fabsf(z));
  return (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
}

// |tanh(z / 2)|: the halving folded into the exponent's constant (fl(c fl(z / 2)) = fl((c / 2) z): scaling by a power
// of two commutes with rounding), one multiply less per sample than voice_tanh_abs(z * 0.5f), the same bits.
__device__ __forceinline__ float voice_tanh_abs_half(float z) {
  const float t = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(z));
  return (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
}
