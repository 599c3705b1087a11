// Backward of the control-rate pass of the Voice render for MI355X (gfx950):
//   (d loss / d ctrl [B,5,Tc], d loss / d constants [B,12])  ->  d loss / d params01 [B,78].
//
// Adjoint of csrc/voice_kernels.hip's voice_env / voice_lfo / voice_modmix kernels (torchsynth-style ADSR, LFO,
// ModulationMixer, parameter ranges), with torch.autograd's conventions where the functions have kinks
// (clamp passes the gradient on the closed side, minimum splits ties, sign() has zero gradient) and 0 where autograd
// would produce nan (0^alpha ramps, zero-length segments).  Same function as voice_grad.py:control_graph + autograd,
// which remains the definition and the test reference (tests/test_voice_grad_gpu.py); this kernel replaces ~600
// small torch kernels (2.1 ms at B=128) by one launch.  The reference has no counterpart: its loop through the
// synth is commented out (/root/reference/audio_to_params.py:56-172).
//
// One workgroup per voice; everything in fp64 (a few thousand points per voice: the cost is nowhere).  A thread owns
// a contiguous run of control points, so the two cumulative sums (LFO phase forward, its gradient backward) are a
// local loop plus one workgroup scan.  Values needed twice are recomputed rather than stored (an ADSR value is three
// pow() calls); LDS holds the LFO phases, their gradients and the six envelope gradients.
//
// Round 3 (ias_voice_control_backward_ws): one workgroup per voice is 64 workgroups at configs[4]'s share of a GPU, and
// two thirds of the kernel's time are fp64 pow() / log() of the six envelopes, which are independent of each other.  The
// split form runs the envelope VALUES (voice_env_value_kernel) and the envelope BACKWARD (voice_env_grad_kernel) on
// 6 x B workgroups each, around the per-voice kernel that keeps the scans and the modulation matrix, and a small finish:
// 245 -> 115 us at B = 64, bit-identical output.
#include "ias_common.h"
#include "voice_table.h"
#include "voice_ctrl_math.h"

// Round 5: the fp64 pow / log / sin / cos / mod of this file by csrc/voice_ctrl_math.h's written-out forms (relative error
// ~2^-50 against the device math library's < 1 ulp: the gradients are checked to 1e-5) -- pow alone was ~300 fp64-rate
// instructions per call, three of them per envelope point.  Arguments outside the forms' domains (x not a positive normal
// double) take the library function.
__device__ const double g_cg_tab[IAS_CTL_TAB_DOUBLES] = IAS_CTL_TAB_INIT;
__device__ __forceinline__ double cg_pow(double x, double a) {
  if (x >= 2.3e-308 && x < 1.0e300) return ias_ctl_pow_d(x, a, g_cg_tab);
  return pow(x, a);
}
__device__ __forceinline__ double cg_log(double x) {
  if (x >= 2.3e-308 && x < 1.0e300) return ias_ctl_log_d(x, g_cg_tab);
  return log(x);
}

#ifndef CG_THREADS
#define CG_THREADS 512    // 8 waves per voice (2 per SIMD; 16 would cap the kernel at 128 VGPRs and spill: 467 us instead of 240)
#endif
#define CG_NSCAL 12

__constant__ IasParamRange c_cg_table[78] = IAS_PARAM_TABLE_INIT;

__device__ __forceinline__ double cg_wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
// sum over the workgroup, result in every thread; s_red: CG_THREADS/64 doubles
__device__ __forceinline__ double cg_block_sum(double v, double* s_red, int tid) {
  v = cg_wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < CG_THREADS / 64; ++w) t += s_red[w];
  return t;
}
// N workgroup sums at once, each in the order of cg_block_sum (butterfly inside a wave, wave order across): two barriers
// for all of them instead of two per value -- the 45 sums of voice_ctrl_grad_kernel were 90 barriers and 45 dependent
// shuffle chains, 40 of the kernel's 60 us at configs[4]'s share (round 5).  s_buf: [N][CG_THREADS / 64] doubles.
template <int N>
__device__ __forceinline__ void cg_block_sum_n(double (&v)[N], double (*s_buf)[CG_THREADS / 64], int tid) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = cg_wave_sum(v[k]);
  __syncthreads();
  if ((tid & 63) == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) s_buf[k][tid >> 6] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double t = 0.0;
    for (int w = 0; w < CG_THREADS / 64; ++w) t += s_buf[k][w];
    v[k] = t;
  }
}
// exclusive prefix of one value per thread (thread order); REVERSE: suffix instead.  s_scan: CG_THREADS/64 doubles
template <bool REVERSE>
__device__ __forceinline__ double cg_block_excl_scan(double v, double* s_scan, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  double incl = v;                       // inclusive scan inside the wave (towards higher lanes; REVERSE: lower)
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double o = REVERSE ? __shfl_down(incl, d, 64) : __shfl_up(incl, d, 64);
    if (REVERSE ? (lane + d < 64) : (lane >= d)) incl += o;
  }
  __syncthreads();
  if (lane == (REVERSE ? 0 : 63)) s_scan[wave] = incl;
  __syncthreads();
  double t = incl - v;
  if (!REVERSE) { for (int w = 0; w < wave; ++w) t += s_scan[w]; }
  else { for (int w = CG_THREADS / 64 - 1; w > wave; --w) t += s_scan[w]; }
  return t;
}

struct CgAdsr { double attack, decay, sustain, release, alpha; };
struct CgRampGrad { double duration, start, alpha; };

// ramp value (after the pow) at control index i; the pieces the backward needs come back through the pointers
__device__ __forceinline__ double cg_ramp(int i, double duration, double alpha, double start, bool has_start,
                                          bool inverse, double cr, double* y_out, double* q_out, double* t_out,
                                          bool* t_live, double head = -1.0) {
  const double dur = duration * cr;
  double t = (double)i;
  bool live = true;
  if (has_start) { t -= start * cr; live = t >= 0.0; if (t < 0.0) t = 0.0; }
  double q = 1.0, y;
  if (dur > 0.0) { q = (t + IAS_EPS) / dur + IAS_EPS; y = q <= 1.0 ? q : 1.0; if (inverse) y = 1.0 - y; }
  else y = 1.0;
  *y_out = y; *q_out = q; *t_out = t; *t_live = live;
  if (y == 1.0) return 1.0;                      // saturated attack ramps, zero-length segments: no pow()
  if (has_start && !live && head >= 0.0) return head;   // flat head before the segment starts: same value for all i
  return cg_pow(y >= 1e-300 ? y : 1e-300, alpha);
}
// accumulate d/d(duration, start, alpha) of g * ramp
__device__ __forceinline__ void cg_ramp_back(double g, double val, double y, double q, double t, bool t_live,
                                             double duration, double alpha, bool has_start, bool inverse, double cr,
                                             CgRampGrad& acc) {
  if (g == 0.0) return;
  const double yc = y >= 1e-300 ? y : 1e-300;
  acc.alpha += g * val * cg_log(yc);
  if (y < 1e-300) return;                       // clamp_min: no gradient below the floor (0^alpha ramps)
  const double dur = duration * cr;
  if (!(dur > 0.0) || q > 1.0) return;          // constant ramp / saturated at 1
  double gq = g * alpha * val / yc;             // alpha * y^(alpha-1)
  if (inverse) gq = -gq;
  acc.duration += -gq * (t + IAS_EPS) / (dur * dur) * cr;
  if (has_start && t_live) acc.start += -(gq / dur) * cr;
}

// values of the decay and release ramps on their flat heads (index 0), computed once per envelope
struct CgHeads { double d, r; };
__device__ __forceinline__ CgHeads cg_heads(const CgAdsr& e, double note_on, double cr) {
  const double na = fmin(e.attack, note_on);
  const double nd = fmin(fmax(note_on - e.attack, 0.0), e.decay);
  double y, q, t; bool l;
  CgHeads h;
  h.d = cg_ramp(-1, nd, e.alpha, na, true, true, cr, &y, &q, &t, &l);        // t clamps to 0 for any i < start
  h.r = cg_ramp(-1, e.release, e.alpha, note_on, true, true, cr, &y, &q, &t, &l);
  return h;
}
__device__ __forceinline__ double cg_adsr(int i, const CgAdsr& e, double note_on, double cr, const CgHeads& h) {
  const double na = fmin(e.attack, note_on);
  const double nd = fmin(fmax(note_on - e.attack, 0.0), e.decay);
  double y, q, t; bool l;
  const double a = cg_ramp(i, na, e.alpha, 0.0, false, false, cr, &y, &q, &t, &l);
  const double d = (1.0 - e.sustain) * cg_ramp(i, nd, e.alpha, na, true, true, cr, &y, &q, &t, &l, h.d) + e.sustain;
  const double r = cg_ramp(i, e.release, e.alpha, note_on, true, true, cr, &y, &q, &t, &l, h.r);
  return a * d * r;
}

// the five LFO shapes at phase arg and their derivatives d shape / d arg
__device__ __forceinline__ void cg_lfo_shapes(double arg, double* sh, double* dsh) {
  const double two_pi = 6.283185307179586, pi = 3.141592653589793;
  double c, sn;
  if (fabs(arg) < 1.0e6) ias_ctl_sincos_d(arg + pi, sn, c); else { c = cos(arg + pi); sn = sin(arg + pi); }
  sh[0] = (c + 1.0) * 0.5;                       dsh[0] = -sn * 0.5;
  double m;
  if (fabs(arg) < 1.0e6) m = ias_ctl_mod_d(arg, two_pi, 0.15915494309189535);
  else { m = fmod(arg, two_pi); if (m < 0.0) m += two_pi; }
  const double saw = m / two_pi;
  const double tri2 = 2.0 * saw;
  sh[1] = tri2 > 1.0 ? 2.0 - tri2 : tri2;        dsh[1] = tri2 > 1.0 ? -1.0 / pi : 1.0 / pi;
  sh[2] = saw;                                   dsh[2] = 1.0 / two_pi;
  sh[3] = 1.0 - saw;                             dsh[3] = -1.0 / two_pi;
  sh[4] = ((c > 0.0 ? 1.0 : (c < 0.0 ? -1.0 : 0.0)) + 1.0) * 0.5;   dsh[4] = 0.0;
}

// value of parameter idx at its normalised setting u, and d value / d u (ModuleParameterRange.from_0to1)
__device__ __forceinline__ void cg_param_value(int idx, double u, double* v_out, double* dv_out) {
  const IasParamRange r = c_cg_table[idx];
  const double ic = 1.0 / r.curve;
  double v, dv;
  if (!r.symmetric) {
    const double uc = u >= 1e-300 ? u : 1e-300;
    v = r.lo + r.span * cg_pow(uc, ic);
    dv = u >= 1e-300 ? r.span * ic * cg_pow(uc, ic - 1.0) : 0.0;
  } else {
    const double dist = 2.0 * u - 1.0, ad = fabs(dist);
    const double ac = ad >= 1e-300 ? ad : 1e-300;
    const double sg = dist > 0.0 ? 1.0 : (dist < 0.0 ? -1.0 : 0.0);
    v = r.lo + r.span * (sg * cg_pow(ac, ic) + 1.0);
    dv = ad >= 1e-300 ? r.span * ic * cg_pow(ac, ic - 1.0) * 2.0 : 0.0;   // sign(d)^2 = 1
  }
  *v_out = v; *dv_out = dv;
}

// SPLIT (round 3): the six-envelope phase -- 60 % of the kernel's fp64 pow / log work, and independent per envelope once
// the envelope cotangents exist -- leaves this kernel: it writes the cotangents (genv [B][6][Tc] fp32), the parameter
// values and d value / d params01 ([B][2][78] fp64) and everything it knows of the gradient; voice_env_grad_kernel (one
// workgroup per (envelope, voice): 6 x the workgroups) and voice_ctrl_finish_kernel complete it.  One workgroup per voice
// is 64 workgroups of 8 waves at configs[4]'s share: a quarter of the CUs at 2 waves per SIMD.
template <bool SPLIT>
__global__ __launch_bounds__(CG_THREADS) void voice_ctrl_grad_kernel(
    const float* __restrict__ params01, const float* __restrict__ g_ctrl, const double* __restrict__ g_scal,
    float* __restrict__ g_params01, int Tc, int ppt /* points per thread */, double cr,
    float* __restrict__ ws_genv /* SPLIT: [B][6][Tc] */, double* __restrict__ ws_vdv /* SPLIT: [B][2][78] */,
    double* __restrict__ ws_part /* SPLIT: [B][40]: [36] = this kernel's own share of d loss / d note_on */,
    const double* __restrict__ ws_env /* SPLIT: [B][6][Tc] envelope values */) {
  extern __shared__ __attribute__((aligned(16))) double cg_smem[];
  double* s_arg = cg_smem;                 // [2][Tc] LFO phases
  double* s_garg = s_arg + 2 * Tc;         // [2][Tc] d loss / d phase
  float* s_genv = reinterpret_cast<float*>(s_garg + 2 * Tc);   // [6][Tc] d loss / d envelope
  __shared__ double s_v[78], s_dv[78], s_gv[78];
  __shared__ double s_red[CG_THREADS / 64], s_scan[CG_THREADS / 64];
  __shared__ double s_many[36][CG_THREADS / 64];

  const int tid = threadIdx.x, b = blockIdx.x;
  const int i_lo = min(tid * ppt, Tc), i_hi = min(i_lo + ppt, Tc);
  const double two_pi = 6.283185307179586;

  // ---- parameter values and d value / d params01 (ModuleParameterRange.from_0to1)
  if (tid < 78) {
    double v, dv;
    cg_param_value(tid, (double)params01[(size_t)b * 78 + tid], &v, &dv);
    s_v[tid] = v; s_dv[tid] = dv; s_gv[tid] = 0.0;
  }
  __syncthreads();
  const double note_on = s_v[IAS_P_KEYBOARD_DURATION];
  const int adsr_base[6] = {IAS_P_ADSR_1_ATTACK, IAS_P_ADSR_2_ATTACK, IAS_P_LFO_1_AMP_ADSR_ATTACK,
                            IAS_P_LFO_2_AMP_ADSR_ATTACK, IAS_P_LFO_1_RATE_ADSR_ATTACK, IAS_P_LFO_2_RATE_ADSR_ATTACK};
  const int lfo_base[2] = {IAS_P_LFO_1_FREQUENCY, IAS_P_LFO_2_FREQUENCY};
  CgAdsr env[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) {
    const int o = adsr_base[e];
    env[e].attack = s_v[o]; env[e].decay = s_v[o + 1]; env[e].sustain = s_v[o + 2];
    env[e].release = s_v[o + 3]; env[e].alpha = s_v[o + 4];
  }
  CgHeads heads[6];
#pragma unroll
  for (int e = 0; e < 6; ++e) { heads[e].d = 0.0; heads[e].r = 0.0; if (!SPLIT) heads[e] = cg_heads(env[e], note_on, cr); }
  auto adsr_val = [&](int e, int i) {
    return SPLIT ? ws_env[((size_t)b * 6 + e) * Tc + i] : cg_adsr(i, env[e], note_on, cr, heads[e]);
  };

  // ---- LFO phases: arg[i] = cumsum(2 pi max(f + depth * rate_env, 0) / cr) + phi
  for (int m = 0; m < 2; ++m) {
    const double f = s_v[lfo_base[m]], dep = s_v[lfo_base[m] + 1], phi = s_v[lfo_base[m] + 2];
    double run = 0.0;
    for (int i = i_lo; i < i_hi; ++i) {
      const double fr = fmax(f + dep * adsr_val(4 + m, i), 0.0);
      run += two_pi * fr / cr;
      s_arg[m * Tc + i] = run;
    }
    const double before = cg_block_excl_scan<false>(run, s_scan, tid);
    for (int i = i_lo; i < i_hi; ++i) s_arg[m * Tc + i] += before + phi;
  }
  __syncthreads();

  // ---- mod matrix, LFO output: per point sources and their gradients
  double mode[2][5], msum[2];
  for (int m = 0; m < 2; ++m) {
    msum[m] = 0.0;
    for (int s = 0; s < 5; ++s) { mode[m][s] = cg_pow(s_v[lfo_base[m] + 3 + s], (double)IAS_LFO_EXPONENT_F); msum[m] += mode[m][s]; }
    for (int s = 0; s < 5; ++s) mode[m][s] /= msum[m];
  }
  double w[5][4];
  for (int k = 0; k < 4; ++k)
    for (int o = 0; o < 5; ++o) w[o][k] = s_v[IAS_P_MOD_MATRIX_ADSR_1_TO_VCO_1_PITCH + k * 5 + o];
  const float* gc = g_ctrl + (size_t)b * 5 * Tc;
  double gw[5][4], gmode[2][5];
  for (int o = 0; o < 5; ++o) for (int k = 0; k < 4; ++k) gw[o][k] = 0.0;
  for (int m = 0; m < 2; ++m) for (int s = 0; s < 5; ++s) gmode[m][s] = 0.0;
  for (int i = i_lo; i < i_hi; ++i) {
    double src[4], mix[2], sh[2][5], dsh[2][5], amp[2];
    src[0] = adsr_val(0, i);
    src[1] = adsr_val(1, i);
    for (int m = 0; m < 2; ++m) {
      cg_lfo_shapes(s_arg[m * Tc + i], sh[m], dsh[m]);
      mix[m] = 0.0;
      for (int s = 0; s < 5; ++s) mix[m] += mode[m][s] * sh[m][s];
      amp[m] = adsr_val(2 + m, i);
      src[2 + m] = mix[m] * amp[m];
    }
    double go[5], gsrc[4];
    for (int o = 0; o < 5; ++o) go[o] = (double)gc[o * Tc + i];
    for (int k = 0; k < 4; ++k) {
      gsrc[k] = 0.0;
      for (int o = 0; o < 5; ++o) { gsrc[k] += w[o][k] * go[o]; gw[o][k] += go[o] * src[k]; }
    }
    s_genv[0 * Tc + i] = (float)gsrc[0];
    s_genv[1 * Tc + i] = (float)gsrc[1];
    for (int m = 0; m < 2; ++m) {
      s_genv[(2 + m) * Tc + i] = (float)(gsrc[2 + m] * mix[m]);
      const double gmix = gsrc[2 + m] * amp[m];
      double ga = 0.0;
      for (int s = 0; s < 5; ++s) { gmode[m][s] += gmix * sh[m][s]; ga += mode[m][s] * dsh[m][s]; }
      s_garg[m * Tc + i] = gmix * ga;
    }
  }
  {
    double all[30];
#pragma unroll
    for (int o = 0; o < 5; ++o)
#pragma unroll
      for (int k = 0; k < 4; ++k) all[o * 4 + k] = gw[o][k];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int s = 0; s < 5; ++s) all[20 + m * 5 + s] = gmode[m][s];
    cg_block_sum_n<30>(all, s_many, tid);
#pragma unroll
    for (int o = 0; o < 5; ++o)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (tid == 0) s_gv[IAS_P_MOD_MATRIX_ADSR_1_TO_VCO_1_PITCH + k * 5 + o] += all[o * 4 + k];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int s = 0; s < 5; ++s) gmode[m][s] = all[20 + m * 5 + s];
  }
  for (int m = 0; m < 2; ++m) {
    double gm[5], dot = 0.0;
    for (int s = 0; s < 5; ++s) { gm[s] = gmode[m][s]; dot += gm[s] * mode[m][s]; }
    if (tid == 0)
      for (int s = 0; s < 5; ++s) {   // mode = p^e / sum p^e;  d p^e / dp = e p^(e-1)
        const double p = s_v[lfo_base[m] + 3 + s], ex = (double)IAS_LFO_EXPONENT_F;
        s_gv[lfo_base[m] + 3 + s] += (gm[s] - dot) / msum[m] * ex * cg_pow(p, ex - 1.0);
      }
  }

  // ---- LFO phase gradient: reverse cumulative sum, then frequency / depth / rate envelope
  for (int m = 0; m < 2; ++m) {
    const double f = s_v[lfo_base[m]], dep = s_v[lfo_base[m] + 1];
    double run = 0.0;
    for (int i = i_hi - 1; i >= i_lo; --i) { run += s_garg[m * Tc + i]; s_garg[m * Tc + i] = run; }
    const double after = cg_block_excl_scan<true>(run, s_scan, tid);
    double gphi = (i_hi > i_lo) ? run : 0.0, gf = 0.0, gdep = 0.0;   // sum of g_arg over the thread's points
    for (int i = i_lo; i < i_hi; ++i) {
      const double ginc = s_garg[m * Tc + i] + after;       // sum_{i' >= i} g_arg[i']
      const double renv = adsr_val(4 + m, i);
      const double gfr = (f + dep * renv >= 0.0) ? ginc * two_pi / cr : 0.0;   // clamp_min passes at >= 0
      gf += gfr; gdep += gfr * renv;
      s_genv[(4 + m) * Tc + i] = (float)(gfr * dep);
    }
    double three[3] = {gphi, gf, gdep};
    cg_block_sum_n<3>(three, s_many, tid);
    if (tid == 0) { s_gv[lfo_base[m]] += three[1]; s_gv[lfo_base[m] + 1] += three[2]; s_gv[lfo_base[m] + 2] += three[0]; }
  }
  __syncthreads();

  // ---- the six envelopes: env = A * D * R
  double g_note_on = 0.0;
  if (SPLIT) {
    float* ge = ws_genv + (size_t)b * 6 * Tc;
    for (int e = 0; e < 6; ++e)
      for (int i = i_lo; i < i_hi; ++i) ge[(size_t)e * Tc + i] = s_genv[e * Tc + i];
    if (tid < 78) { ws_vdv[((size_t)b * 2) * 78 + tid] = s_v[tid]; ws_vdv[((size_t)b * 2 + 1) * 78 + tid] = s_dv[tid]; }
  }
  for (int e = 0; e < (SPLIT ? 0 : 6); ++e) {
    const CgAdsr p = env[e];
    const double na = fmin(p.attack, note_on);
    const double nd0 = fmax(note_on - p.attack, 0.0);
    const double nd = fmin(nd0, p.decay);
    CgRampGrad ga = {0, 0, 0}, gd = {0, 0, 0}, gr = {0, 0, 0};
    double gsus = 0.0;
    for (int i = i_lo; i < i_hi; ++i) {
      const double g = (double)s_genv[e * Tc + i];
      double ya, qa, ta, yd, qd, td, yr, qr, tr; bool la, ld, lr;
      const double a = cg_ramp(i, na, p.alpha, 0.0, false, false, cr, &ya, &qa, &ta, &la);
      const double dr = cg_ramp(i, nd, p.alpha, na, true, true, cr, &yd, &qd, &td, &ld, heads[e].d);
      const double r = cg_ramp(i, p.release, p.alpha, note_on, true, true, cr, &yr, &qr, &tr, &lr, heads[e].r);
      const double d = (1.0 - p.sustain) * dr + p.sustain;
      cg_ramp_back(g * d * r, a, ya, qa, ta, la, na, p.alpha, false, false, cr, ga);
      cg_ramp_back(g * a * r * (1.0 - p.sustain), dr, yd, qd, td, ld, nd, p.alpha, true, true, cr, gd);
      cg_ramp_back(g * a * d, r, yr, qr, tr, lr, p.release, p.alpha, true, true, cr, gr);
      gsus += g * a * r * (1.0 - dr);
    }
    // durations / starts back to attack, decay, release, note_on
    //   new_attack = min(attack, note_on)           (ramp A duration, ramp D start)
    //   new_decay  = min(max(note_on - attack, 0), decay)   (ramp D duration)
    //   release                                      (ramp R duration), note_on (ramp R start)
    double six[6] = {ga.duration + gd.start, gd.duration, gr.duration, gr.start, ga.alpha + gd.alpha + gr.alpha, gsus};
    cg_block_sum_n<6>(six, s_many, tid);
    const double g_na = six[0], g_nd = six[1], g_rel = six[2], g_no_r = six[3], g_alpha = six[4], g_sus = six[5];
    if (tid == 0) {
      double g_att = 0.0, g_dec = 0.0, g_no = g_no_r;
      // torch.minimum: the smaller argument takes the gradient, a tie splits it
      if (p.attack < note_on) g_att += g_na; else if (p.attack > note_on) g_no += g_na; else { g_att += 0.5 * g_na; g_no += 0.5 * g_na; }
      double g_nd0 = 0.0;
      if (nd0 < p.decay) g_nd0 = g_nd; else if (nd0 > p.decay) g_dec += g_nd; else { g_nd0 = 0.5 * g_nd; g_dec += 0.5 * g_nd; }
      if (note_on - p.attack >= 0.0) { g_no += g_nd0; g_att -= g_nd0; }      // clamp_min passes at >= 0
      const int o = adsr_base[e];
      s_gv[o] += g_att; s_gv[o + 1] += g_dec; s_gv[o + 2] += g_sus; s_gv[o + 3] += g_rel; s_gv[o + 4] += g_alpha;
      g_note_on += g_no;
    }
  }

  // ---- per-voice constants and the chain to params01
  if (tid == 0) {
    const double* gs = g_scal + (size_t)b * CG_NSCAL;
    s_gv[IAS_P_KEYBOARD_DURATION] += g_note_on;
    if (SPLIT) ws_part[(size_t)b * 40 + 36] = s_gv[IAS_P_KEYBOARD_DURATION];
    const double midi = s_v[IAS_P_KEYBOARD_MIDI_F0], dep2 = s_v[IAS_P_VCO_2_MOD_DEPTH];
    const double F = 440.0 * exp2((midi + fmax(dep2, 0.0) - 69.0) / 12.0);
    const double lg = log10(F);
    // kpart = pi * 12000 / (F log10 F)
    const double dk_dF = -3.141592653589793 * 12000.0 * (lg + 0.4342944819032518) / (F * lg * F * lg);
    const double dk_dP = dk_dF * F * 0.6931471805599453 / 12.0;
    s_gv[IAS_P_KEYBOARD_MIDI_F0] += gs[0] + gs[3] + gs[6] * dk_dP;
    s_gv[IAS_P_VCO_1_TUNING] += gs[0];
    s_gv[IAS_P_VCO_1_MOD_DEPTH] += gs[1];
    s_gv[IAS_P_VCO_1_INITIAL_PHASE] += gs[2];
    s_gv[IAS_P_VCO_2_TUNING] += gs[3];
    s_gv[IAS_P_VCO_2_MOD_DEPTH] += gs[4] + (dep2 >= 0.0 ? gs[6] * dk_dP : 0.0);
    s_gv[IAS_P_VCO_2_INITIAL_PHASE] += gs[5];
    s_gv[IAS_P_VCO_2_SHAPE] += gs[7] - 0.5 * gs[8];
    s_gv[IAS_P_MIXER_VCO_1] += gs[9];
    s_gv[IAS_P_MIXER_VCO_2] += gs[10];
    s_gv[IAS_P_MIXER_NOISE] += gs[11];
  }
  __syncthreads();
  if (tid < 78) g_params01[(size_t)b * 78 + tid] = (float)(s_gv[tid] * s_dv[tid]);
}

// Envelope values of the split form: ws_env[b][e][i] = ADSR envelope e of voice b at control point i (fp64), one
// workgroup per (envelope, voice).
#define EG_THREADS 512
__global__ __launch_bounds__(EG_THREADS) void voice_env_value_kernel(const float* __restrict__ params01,
                                                                     double* __restrict__ ws_env, int Tc, int ppt, double cr) {
  __shared__ double s_p[6];
  const int tid = threadIdx.x, e = blockIdx.x, b = blockIdx.y;
  const int adsr_base[6] = {IAS_P_ADSR_1_ATTACK, IAS_P_ADSR_2_ATTACK, IAS_P_LFO_1_AMP_ADSR_ATTACK,
                            IAS_P_LFO_2_AMP_ADSR_ATTACK, IAS_P_LFO_1_RATE_ADSR_ATTACK, IAS_P_LFO_2_RATE_ADSR_ATTACK};
  if (tid < 6) {
    const int idx = tid < 5 ? adsr_base[e] + tid : IAS_P_KEYBOARD_DURATION;
    double v, dv;
    cg_param_value(idx, (double)params01[(size_t)b * 78 + idx], &v, &dv);
    s_p[tid] = v;
  }
  __syncthreads();
  CgAdsr p;
  p.attack = s_p[0]; p.decay = s_p[1]; p.sustain = s_p[2]; p.release = s_p[3]; p.alpha = s_p[4];
  const double note_on = s_p[5];
  const CgHeads heads = cg_heads(p, note_on, cr);
  const int i_lo = min(tid * ppt, Tc), i_hi = min(i_lo + ppt, Tc);
  double* out = ws_env + ((size_t)b * 6 + e) * Tc;
  for (int i = i_lo; i < i_hi; ++i) out[i] = cg_adsr(i, p, note_on, cr, heads);
}

// The six-envelope phase of the split form: one workgroup per (envelope, voice).  ws_part [B][6][6] fp64 receives the
// envelope's d loss / d (attack, decay, sustain, release, alpha) and its share of d loss / d note_on.
__global__ __launch_bounds__(EG_THREADS) void voice_env_grad_kernel(const float* __restrict__ ws_genv,
                                                                    const double* __restrict__ ws_vdv,
                                                                    double* __restrict__ ws_part, int Tc, int ppt, double cr) {
  __shared__ double s_red[6][EG_THREADS / 64];
  const int tid = threadIdx.x, e = blockIdx.x, b = blockIdx.y;
  const int adsr_base[6] = {IAS_P_ADSR_1_ATTACK, IAS_P_ADSR_2_ATTACK, IAS_P_LFO_1_AMP_ADSR_ATTACK,
                            IAS_P_LFO_2_AMP_ADSR_ATTACK, IAS_P_LFO_1_RATE_ADSR_ATTACK, IAS_P_LFO_2_RATE_ADSR_ATTACK};
  const double* v = ws_vdv + (size_t)b * 2 * 78;
  const int o = adsr_base[e];
  CgAdsr p;
  p.attack = v[o]; p.decay = v[o + 1]; p.sustain = v[o + 2]; p.release = v[o + 3]; p.alpha = v[o + 4];
  const double note_on = v[IAS_P_KEYBOARD_DURATION];
  const CgHeads heads = cg_heads(p, note_on, cr);
  const double na = fmin(p.attack, note_on);
  const double nd0 = fmax(note_on - p.attack, 0.0);
  const double nd = fmin(nd0, p.decay);
  const int i_lo = min(tid * ppt, Tc), i_hi = min(i_lo + ppt, Tc);
  const float* ge = ws_genv + ((size_t)b * 6 + e) * Tc;
  CgRampGrad ga = {0, 0, 0}, gd = {0, 0, 0}, gr = {0, 0, 0};
  double gsus = 0.0;
  for (int i = i_lo; i < i_hi; ++i) {
    const double g = (double)ge[i];
    double ya, qa, ta, yd, qd, td, yr, qr, tr; bool la, ld, lr;
    const double a = cg_ramp(i, na, p.alpha, 0.0, false, false, cr, &ya, &qa, &ta, &la);
    const double dr = cg_ramp(i, nd, p.alpha, na, true, true, cr, &yd, &qd, &td, &ld, heads.d);
    const double r = cg_ramp(i, p.release, p.alpha, note_on, true, true, cr, &yr, &qr, &tr, &lr, heads.r);
    const double d = (1.0 - p.sustain) * dr + p.sustain;
    cg_ramp_back(g * d * r, a, ya, qa, ta, la, na, p.alpha, false, false, cr, ga);
    cg_ramp_back(g * a * r * (1.0 - p.sustain), dr, yd, qd, td, ld, nd, p.alpha, true, true, cr, gd);
    cg_ramp_back(g * a * d, r, yr, qr, tr, lr, p.release, p.alpha, true, true, cr, gr);
    gsus += g * a * r * (1.0 - dr);
  }
  // six workgroup sums at once (thread order within a wave, wave order across: the order of cg_block_sum)
  double vals[6] = {ga.duration + gd.start, gd.duration, gr.duration, gr.start, ga.alpha + gd.alpha + gr.alpha, gsus};
#pragma unroll
  for (int k = 0; k < 6; ++k) vals[k] = cg_wave_sum(vals[k]);
  if ((tid & 63) == 0)
#pragma unroll
    for (int k = 0; k < 6; ++k) s_red[k][tid >> 6] = vals[k];
  __syncthreads();
  if (tid == 0) {
    double t[6];
    for (int k = 0; k < 6; ++k) {
      t[k] = 0.0;
      for (int w = 0; w < EG_THREADS / 64; ++w) t[k] += s_red[k][w];
    }
    const double g_na = t[0], g_nd = t[1], g_rel = t[2], g_no_r = t[3], g_alpha = t[4], g_sus = t[5];
    double g_att = 0.0, g_dec = 0.0, g_no = g_no_r;
    // torch.minimum: the smaller argument takes the gradient, a tie splits it
    if (p.attack < note_on) g_att += g_na; else if (p.attack > note_on) g_no += g_na; else { g_att += 0.5 * g_na; g_no += 0.5 * g_na; }
    double g_nd0 = 0.0;
    if (nd0 < p.decay) g_nd0 = g_nd; else if (nd0 > p.decay) g_dec += g_nd; else { g_nd0 = 0.5 * g_nd; g_dec += 0.5 * g_nd; }
    if (note_on - p.attack >= 0.0) { g_no += g_nd0; g_att -= g_nd0; }      // clamp_min passes at >= 0
    double* out = ws_part + (size_t)b * 40 + e * 6;
    out[0] = g_att; out[1] = g_dec; out[2] = g_sus; out[3] = g_rel; out[4] = g_alpha; out[5] = g_no;
  }
}

// g_params01 of the 30 envelope parameters, and the envelopes' share of d loss / d note_on added (in envelope order) to
// what voice_ctrl_grad_kernel<true> left there.
__global__ __launch_bounds__(64) void voice_ctrl_finish_kernel(const double* __restrict__ ws_vdv,
                                                               const double* __restrict__ ws_part,
                                                               float* __restrict__ g_params01) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int adsr_base[6] = {IAS_P_ADSR_1_ATTACK, IAS_P_ADSR_2_ATTACK, IAS_P_LFO_1_AMP_ADSR_ATTACK,
                            IAS_P_LFO_2_AMP_ADSR_ATTACK, IAS_P_LFO_1_RATE_ADSR_ATTACK, IAS_P_LFO_2_RATE_ADSR_ATTACK};
  const double* dv = ws_vdv + ((size_t)b * 2 + 1) * 78;
  const double* part = ws_part + (size_t)b * 40;
  if (tid < 30) {
    const int e = tid / 5, k = tid - 5 * e, pidx = adsr_base[e] + k;
    g_params01[(size_t)b * 78 + pidx] = (float)(part[e * 6 + k] * dv[pidx]);
  } else if (tid == 32) {
    double g_no = 0.0;
    for (int e = 0; e < 6; ++e) g_no += part[e * 6 + 5];
    // voice_ctrl_grad_kernel<true> wrote (float)(gv dv) for note_on without this share; its gv is recovered in fp64 from
    // the parts it saved next to dv (ws_vdv[b][0][..] holds the values, that partial gradient is kept in ws_part[b][36])
    g_params01[(size_t)b * 78 + IAS_P_KEYBOARD_DURATION] =
        (float)((part[36] + g_no) * dv[IAS_P_KEYBOARD_DURATION]);
  }
}

// params01 [B,78]; g_ctrl [B,5,Tc] fp32 and g_scal [B,12] fp64 (ias_voice_backward's g_ctrl and the tile sum of its
// partials); g_params01 [B,78] out.  IAS_ERR_UNSUPPORTED when the control buffer does not fit LDS (Tc > ~3000):
// the caller then differentiates voice_grad.control_graph with torch.
extern "C" int ias_voice_control_backward(const float* params01, const float* g_ctrl, const double* g_scal,
                                          float* g_params01, int B, int Tc, int control_rate, void* stream_) {
  if (!params01 || !g_ctrl || !g_scal || !g_params01 || B <= 0 || Tc <= 1 || control_rate <= 0) return IAS_ERR_ARG;
  if (control_rate != IAS_CONTROL_RATE) return IAS_ERR_UNSUPPORTED;
  const size_t lds = sizeof(double) * 4 * (size_t)Tc + sizeof(float) * 6 * (size_t)Tc;
  if (lds > 150 * 1024) return IAS_ERR_UNSUPPORTED;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)voice_ctrl_grad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ppt = (Tc + CG_THREADS - 1) / CG_THREADS;
  hipLaunchKernelGGL(voice_ctrl_grad_kernel<false>, dim3(B), dim3(CG_THREADS), lds, (hipStream_t)stream_, params01, g_ctrl,
                     g_scal, g_params01, Tc, ppt, (double)control_rate, (float*)nullptr, (double*)nullptr, (double*)nullptr,
                     (const double*)nullptr);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// The same in four launches (round 3): envelope values and the six-envelope phase on 6 x B workgroups each.  workspace:
// ias_voice_control_backward_ws_bytes(B, Tc) bytes of device memory (16-byte aligned), contents undefined before and after.
extern "C" long long ias_voice_control_backward_ws_bytes(int B, int Tc) {
  if (B <= 0 || Tc <= 1) return IAS_ERR_ARG;
  return (long long)sizeof(float) * B * 6 * Tc + (long long)sizeof(double) * B * (2 * 78 + 40 + 6 * (long long)Tc) + 64;
}
static int voice_control_backward_stage(int stage, const float* params01, const float* g_ctrl, const double* g_scal,
                                        float* g_params01, void* workspace, long long workspace_bytes, int B, int Tc,
                                        int control_rate, void* stream_);
extern "C" int ias_voice_control_backward_ws(const float* params01, const float* g_ctrl, const double* g_scal,
                                             float* g_params01, void* workspace, long long workspace_bytes, int B, int Tc,
                                             int control_rate, void* stream_) {
  return voice_control_backward_stage(-1, params01, g_ctrl, g_scal, g_params01, workspace, workspace_bytes, B, Tc, control_rate,
                                      stream_);
}
// In two stages on the same workspace: stage 0 = the envelope VALUES (voice_env_value_kernel: parameters only, no
// cotangent; g_ctrl, g_scal, g_params01 may be NULL), stage 1 = the rest.
extern "C" int ias_voice_control_backward_ws_stage(int stage, const float* params01, const float* g_ctrl, const double* g_scal,
                                                   float* g_params01, void* workspace, long long workspace_bytes, int B,
                                                   int Tc, int control_rate, void* stream_) {
  if (stage != 0 && stage != 1) return IAS_ERR_ARG;
  return voice_control_backward_stage(stage, params01, g_ctrl, g_scal, g_params01, workspace, workspace_bytes, B, Tc,
                                      control_rate, stream_);
}
static int voice_control_backward_stage(int stage, const float* params01, const float* g_ctrl, const double* g_scal,
                                        float* g_params01, void* workspace, long long workspace_bytes, int B, int Tc,
                                        int control_rate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!params01 || !workspace || B <= 0 || B > 65535 || Tc <= 1 || control_rate <= 0) return IAS_ERR_ARG;
  if (stage != 0 && (!g_ctrl || !g_scal || !g_params01)) return IAS_ERR_ARG;
  if (control_rate != IAS_CONTROL_RATE) return IAS_ERR_UNSUPPORTED;
  if (workspace_bytes < ias_voice_control_backward_ws_bytes(B, Tc) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return IAS_ERR_WORKSPACE;
  const size_t lds = sizeof(double) * 4 * (size_t)Tc + sizeof(float) * 6 * (size_t)Tc;
  if (lds > 150 * 1024) return IAS_ERR_UNSUPPORTED;
  // doubles first (alignment), then the envelope cotangents
  double* ws_vdv = (double*)workspace;
  double* ws_part = ws_vdv + (size_t)B * 2 * 78;
  double* ws_env = ws_part + (size_t)B * 40;
  float* ws_genv = (float*)(ws_env + (size_t)B * 6 * Tc);
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute((const void*)voice_ctrl_grad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ppt = (Tc + CG_THREADS - 1) / CG_THREADS;
  const int eppt = (Tc + EG_THREADS - 1) / EG_THREADS;
  if (stage <= 0)
    hipLaunchKernelGGL(voice_env_value_kernel, dim3(6, B), dim3(EG_THREADS), 0, stream, params01, ws_env, Tc, eppt,
                       (double)control_rate);
  if (stage == 0) return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  hipLaunchKernelGGL(voice_ctrl_grad_kernel<true>, dim3(B), dim3(CG_THREADS), lds, stream, params01, g_ctrl, g_scal,
                     g_params01, Tc, ppt, (double)control_rate, ws_genv, ws_vdv, ws_part, (const double*)ws_env);
  hipLaunchKernelGGL(voice_env_grad_kernel, dim3(6, B), dim3(EG_THREADS), 0, stream, ws_genv, ws_vdv, ws_part, Tc, eppt,
                     (double)control_rate);
  hipLaunchKernelGGL(voice_ctrl_finish_kernel, dim3(B), dim3(64), 0, stream, ws_vdv, ws_part, g_params01);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
