// Host-only emulation of the Voice kernels' arithmetic (TEST AID, not product):
// compiles csrc/voice_math.h with g++ and runs the same per-sample functions in
// plain sequential loops, so `-m "not gpu"` tests can check the device math
// against oracle/synth_oracle.py (math mode "cr") without a GPU.
// Build: g++ -O2 -ffp-contract=off -shared -fPIC (see tests/test_voice_math_cpu.py).
#include <vector>
#include <cstring>
#include "voice_math.h"
#include "voice_table.h"

// emul_set_ctl(1): the control-rate pow / cos / fmod by csrc/voice_ctrl_math.h (what the HIP control kernel runs);
// 0 (default): the libm calls that DEFINE the "cr" arithmetic.  Both must give the oracle's bits.
static const double g_ctl_tab[IAS_CTL_TAB_DOUBLES] = IAS_CTL_TAB_INIT;
static const double* g_ctl = nullptr;
extern "C" void emul_set_ctl(int on) { g_ctl = on ? g_ctl_tab : nullptr; }

static void adsr_from(const float* p, int base, IasAdsr& e) {
  e.attack = p[base + 0]; e.decay = p[base + 1]; e.sustain = p[base + 2];
  e.release = p[base + 3]; e.alpha = p[base + 4];
}

extern "C" int emul_voice_render(const float* params01, const float* noise, float* audio,
                                 float* ctrl_out /* [B,5,Tc] or null */,
                                 float* mixed_out /* [B,T] or null */,
                                 int B, int T, int Tc, int sample_rate, int control_rate) {
  const float cr = (float)control_rate, sr = (float)sample_rate, eps = (float)IAS_EPS;
  const float scale = (T > 1) ? (float)(Tc - 1) / (float)(T - 1) : 0.0f;
  std::vector<float> env(6 * Tc), lfo(2 * Tc), ctrl(5 * Tc), mixed(T);
  for (int b = 0; b < B; ++b) {
    float p[IAS_NPARAMS];
    for (int i = 0; i < IAS_NPARAMS; ++i) {
      const IasParamRange& r = IAS_PARAM_TABLE[i];
      p[i] = ias_map_param(params01[b * IAS_NPARAMS + i], (float)r.lo, (float)r.span, (float)r.curve, r.symmetric, g_ctl);
    }
    const float midi_f0 = p[IAS_P_KEYBOARD_MIDI_F0], note_on = p[IAS_P_KEYBOARD_DURATION];
    // env order: adsr_1, adsr_2, lfo_1_amp, lfo_2_amp, lfo_1_rate, lfo_2_rate
    const int bases[6] = {IAS_P_ADSR_1_ATTACK, IAS_P_ADSR_2_ATTACK, IAS_P_LFO_1_AMP_ADSR_ATTACK,
                          IAS_P_LFO_2_AMP_ADSR_ATTACK, IAS_P_LFO_1_RATE_ADSR_ATTACK, IAS_P_LFO_2_RATE_ADSR_ATTACK};
    for (int a = 0; a < 6; ++a) {
      IasAdsr e; adsr_from(p, bases[a], e);
      for (int t = 0; t < Tc; ++t) env[a * Tc + t] = ias_adsr(t, e, note_on, cr, eps, g_ctl);
    }
    const int lbase[2] = {IAS_P_LFO_1_FREQUENCY, IAS_P_LFO_2_FREQUENCY};
    for (int l = 0; l < 2; ++l) {
      const float* q = p + lbase[l];
      float mode[5]; ias_lfo_mode(q + 3, mode, g_ctl);
      double acc = 0.0;
      for (int t = 0; t < Tc; ++t) {
        acc += (double)ias_lfo_inc(q[0], q[1], env[(4 + l) * Tc + t], cr);
        const float arg = ias_add((float)acc, q[2]);
        lfo[l * Tc + t] = ias_mul(ias_lfo_shape_mix(arg, mode, g_ctl), env[(2 + l) * Tc + t]);
      }
    }
    const float* w = p + IAS_P_MOD_MATRIX_ADSR_1_TO_VCO_1_PITCH;  // [input k][output j]
    for (int j = 0; j < 5; ++j)
      for (int t = 0; t < Tc; ++t) {
        const float o = ias_dot4_cr(w[0 * 5 + j], w[1 * 5 + j], w[2 * 5 + j], w[3 * 5 + j],
                                    env[0 * Tc + t], env[1 * Tc + t], lfo[0 * Tc + t], lfo[1 * Tc + t]);
        ctrl[j * Tc + t] = o;
      }
    if (ctrl_out) memcpy(ctrl_out + (size_t)b * 5 * Tc, ctrl.data(), sizeof(float) * 5 * Tc);

    IasVoiceConst vc;
    vc.f0_1 = ias_add(midi_f0, p[IAS_P_VCO_1_TUNING]); vc.depth_1 = p[IAS_P_VCO_1_MOD_DEPTH]; vc.phi_1 = p[IAS_P_VCO_1_INITIAL_PHASE];
    vc.f0_2 = ias_add(midi_f0, p[IAS_P_VCO_2_TUNING]); vc.depth_2 = p[IAS_P_VCO_2_MOD_DEPTH]; vc.phi_2 = p[IAS_P_VCO_2_INITIAL_PHASE];
    vc.kpart = ias_partials_k(midi_f0, vc.depth_2, g_ctl);
    vc.shape = p[IAS_P_VCO_2_SHAPE];
    vc.shape_gain = ias_sub(1.0f, ias_div(vc.shape, 2.0f));
    vc.lvl0 = p[IAS_P_MIXER_VCO_1]; vc.lvl1 = p[IAS_P_MIXER_VCO_2]; vc.lvl2 = p[IAS_P_MIXER_NOISE];

    double ph1 = 0.0, ph2 = 0.0;
    float peak = 0.0f;
    for (int j = 0; j < T; ++j) {
      int i0, i1; float w0, w1;
      ias_interp_pos(j, scale, Tc, i0, i1, w0, w1);
      float m[5];
      for (int k = 0; k < 5; ++k) m[k] = ias_lerp(ctrl[k * Tc + i0], ctrl[k * Tc + i1], w0, w1);
      ph1 += (double)ias_vco_inc(vc.f0_1, vc.depth_1, m[0], sr);
      ph2 += (double)ias_vco_inc(vc.f0_2, vc.depth_2, m[2], sr);
      const float a1 = ias_add((float)ph1, vc.phi_1), a2 = ias_add((float)ph2, vc.phi_2);
      const float o = ias_mix_sample(a1, a2, m[1], m[3], m[4], noise[(size_t)b * T + j], vc);
      mixed[j] = o;
      peak = fmaxf(peak, fabsf(o));
    }
    if (mixed_out) memcpy(mixed_out + (size_t)b * T, mixed.data(), sizeof(float) * T);
    for (int j = 0; j < T; ++j)
      audio[(size_t)b * T + j] = (peak > 1.0f) ? ias_div(mixed[j], peak) : mixed[j];
  }
  return 0;
}

// Counts inputs on which the cheaper device formulations differ from the specification ones.
extern "C" long long emul_check_fast_paths(const float* pm, long long n, float f0, float depth, int sample_rate) {
  long long bad = 0;
  const float sr = (float)sample_rate;
  const double inv_sr = 1.0 / (double)sample_rate;
  for (long long i = 0; i < n; ++i) {
    const float a = ias_vco_inc(f0, depth, pm[i], sr), b = ias_vco_inc_fast(f0, depth, pm[i], inv_sr);
    if (a != b) ++bad;
    const float t = pm[i] * 10.0f - 5.0f;
    if (ias_exp2_cr(t) != ias_exp2_cr_fast(t)) ++bad;
    // table form of the render kernel, over the whole pitch range t = (c - 69) / 12, c in [0, 127]
    static const double tab[IAS_EXP2_TAB_LEN] = IAS_EXP2_TAB_INIT;
    const float tp = ias_div(ias_sub(pm[i] * 127.0f, 69.0f), 12.0f);
    if (ias_exp2_cr(tp) != ias_exp2_cr_tab(tp, tab)) ++bad;
    if (i < 128) {   // the range ends exactly
      const float te = ias_div(ias_sub(i < 64 ? 0.0f : 127.0f, 69.0f), 12.0f);
      if (ias_exp2_cr(te) != ias_exp2_cr_tab(te, tab)) ++bad;
    }
    const float w = pm[i] * 7000.0f;
    if (ias_div(w, sr) != ias_div_by_recip(w, inv_sr)) ++bad;
    if (ias_div(t, 12.0f) != ias_div_by_recip(t, 1.0 / 12.0)) ++bad;
    // the three-FMA quotient of the render kernel (exhaustive proof: scripts/diag/div_by_const_check.c)
    const float s69 = pm[i] * 127.0f - 69.0f;
    if (ias_div(s69, 12.0f) != ias_div_fma(s69, 12.0f, 1.0f / 12.0f)) ++bad;
    if (ias_div_fma_rate_ok(sample_rate)) {
      const float wf = 51.0f + pm[i] * 80000.0f;
      if (ias_div(wf, sr) != ias_div_fma(wf, sr, 1.0f / sr)) ++bad;
    }
  }
  return bad;
}

// Counts control samples where the "headed" ADSR (cached flat ramp heads) differs from ias_adsr.
extern "C" long long emul_check_adsr_headed(const float* p5n /* [n][6]: attack, decay, sustain, release, alpha, note_on */,
                                            long long n, int Tc, int control_rate) {
  long long bad = 0;
  const float cr = (float)control_rate, eps = (float)IAS_EPS;
  for (long long i = 0; i < n; ++i) {
    IasAdsr e;
    e.attack = p5n[6 * i]; e.decay = p5n[6 * i + 1]; e.sustain = p5n[6 * i + 2];
    e.release = p5n[6 * i + 3]; e.alpha = p5n[6 * i + 4];
    const float note_on = p5n[6 * i + 5];
    const IasAdsrHeads h = ias_adsr_heads(e, note_on, cr, eps);
    for (int t = 0; t < Tc; ++t) {
      const float a = ias_adsr(t, e, note_on, cr, eps), b = ias_adsr_headed(t, e, note_on, cr, eps, h);
      if (!(a == b) && !(a != a && b != b)) ++bad;
    }
  }
  return bad;
}
