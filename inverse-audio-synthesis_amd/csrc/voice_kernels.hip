// Voice render for MI355X (gfx950): control-rate pass + audio-rate pass.
//
// Replaces torchsynth's Voice.output() as the reference drives it
// (/root/reference/vicreg_audio_params.py:86-94,114; audio_to_params.py:215,240-257).
// Arithmetic contract: csrc/voice_math.h (== oracle/synth_oracle.py, math "cr").
//
// Kernels
//   voice_env_kernel       one workgroup per (envelope, voice): the six ADSR envelopes [B][6][Tc].
//   voice_lfo_kernel       one workgroup per (LFO, voice): fp64 phase scan, five shapes, amplitude.
//   voice_modmix_kernel    4x5 mod matrix -> ctrl[B][5][Tc] + IasVoiceConst[B].
//   voice_audio_kernel     per (tile, voice), ONE pass: phase increments of both VCOs, fp64 scan
//                          chained across the tiles of a row (ticketed look-back; fp64 sums of
//                          fp32 increments below 2^19 are exact, so any summation order gives the
//                          same bits), oscillators, VCAs, mixer -> unnormalised audio, row peak.
//   voice_normalize_kernel audio = peak > 1 ? x / peak : x   (torchsynth normalize_if_clipping)
//
// HBM traffic per audio sample: noise 4 B read + 4 B write (pass 1), 4 B read + 4 B write
// (normalise).  Control-rate traffic is < 0.1 %.
#include <cstdio>
#include <cstdlib>
#include "voice_math.h"
#include "wave_ops.h"
#include "voice_trig.h"
#include "voice_table.h"

#define VOICE_THREADS 256                     // control-rate kernels
#define VOICE_WAVES (VOICE_THREADS / 64)
#define AUDIO_THREADS 256                     // audio-rate kernel (512-thread tiles measured no faster)
#define AUDIO_WAVES (AUDIO_THREADS / 64)
#define VOICE_SPT 16                          // samples per thread (8 and 12 measured slower: per-tile latencies)
#define VOICE_TILE (AUDIO_THREADS * VOICE_SPT)  // 4096 samples per workgroup

__constant__ IasParamRange c_param_table[IAS_NPARAMS] = IAS_PARAM_TABLE_INIT;

// wave helpers (DPP scans, reductions): wave_ops.h

// ------------------------------------------------------------------ control rate
__device__ __forceinline__ float mapped_param(const float* __restrict__ params01, int b, int idx) {
  const IasParamRange r = c_param_table[idx];
  return ias_map_param(params01[(size_t)b * IAS_NPARAMS + idx], (float)r.lo, (float)r.span, (float)r.curve,
                       r.symmetric);
}

__device__ __forceinline__ int adsr_base(int a) {
  // env order: adsr_1, adsr_2, lfo_1_amp, lfo_2_amp, lfo_1_rate, lfo_2_rate
  switch (a) {
    case 0: return IAS_P_ADSR_1_ATTACK;
    case 1: return IAS_P_ADSR_2_ATTACK;
    case 2: return IAS_P_LFO_1_AMP_ADSR_ATTACK;
    case 3: return IAS_P_LFO_2_AMP_ADSR_ATTACK;
    case 4: return IAS_P_LFO_1_RATE_ADSR_ATTACK;
    default: return IAS_P_LFO_2_RATE_ADSR_ATTACK;
  }
}

// One workgroup per (envelope, voice): env[b][a][t] for all control samples t.  768 workgroups at
// B = 128 fill the chip; the three fp64 pow() per sample dominate, so saturated ramps (base exactly
// 0 or 1, where pow is exact) skip it.
__global__ __launch_bounds__(VOICE_THREADS) void voice_env_kernel(const float* __restrict__ params01,
                                                                  float* __restrict__ env, int Tc,
                                                                  float control_rate) {
  __shared__ float s_p[8];
  const int a = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  if (tid < 5) s_p[tid] = mapped_param(params01, b, adsr_base(a) + tid);
  if (tid == 5) s_p[5] = mapped_param(params01, b, IAS_P_KEYBOARD_DURATION);
  __syncthreads();
  IasAdsr e;
  e.attack = s_p[0]; e.decay = s_p[1]; e.sustain = s_p[2]; e.release = s_p[3]; e.alpha = s_p[4];
  const float note_on = s_p[5], eps = (float)IAS_EPS;
  // flat heads of the decay / release ramps: the same for every t before the ramp starts
  if (tid == 0) s_p[6] = ias_adsr_heads(e, note_on, control_rate, eps).decay_head;
  if (tid == 64) s_p[7] = ias_adsr_heads(e, note_on, control_rate, eps).release_head;
  __syncthreads();
  IasAdsrHeads heads;
  heads.decay_head = s_p[6]; heads.release_head = s_p[7];
  float* out = env + ((size_t)b * 8 + a) * Tc;   // sig rows 0-5
  for (int t = tid; t < Tc; t += VOICE_THREADS) out[t] = ias_adsr_headed(t, e, note_on, control_rate, eps, heads);
}

// One workgroup per (LFO, voice): phase scan (fp64 accumulate, fp32 per-sample round), the five LFO
// shapes, amplitude envelope -> sig[b][6 + l][t].   sig rows: 0-5 envelopes (voice_env_kernel), 6-7 LFOs.
__global__ __launch_bounds__(VOICE_THREADS) void voice_lfo_kernel(
    const float* __restrict__ params01, float* __restrict__ sig,
    float* __restrict__ dbg /* optional [B][10][Tc]: receives LFO phases (rows 6,7) and outputs (rows 8,9) */,
    int Tc, float control_rate) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  double* s_wtot = reinterpret_cast<double*>(smem);   // VOICE_WAVES wave totals
  double* s_sum = s_wtot + VOICE_WAVES;               // Tc wave-local inclusive sums
  __shared__ float s_q[8];
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qbase = (l == 0) ? IAS_P_LFO_1_FREQUENCY : IAS_P_LFO_2_FREQUENCY;
  if (tid < 8) s_q[tid] = mapped_param(params01, b, qbase + tid);
  __syncthreads();
  const float freq = s_q[0], depth = s_q[1], phi = s_q[2];
  const float* rate_env = sig + ((size_t)b * 8 + 4 + l) * Tc;
  const float* amp_env = sig + ((size_t)b * 8 + 2 + l) * Tc;

  // each wave scans a contiguous quarter of the control buffer in chunks of 64, then the quarters are
  // chained through LDS (fp64; the order differs from a sequential loop only below 1e-16 relative,
  // which the per-sample rounding to fp32 absorbs)
  const int per_wave = ((Tc + VOICE_WAVES - 1) / VOICE_WAVES + 63) / 64 * 64;
  const int t_begin = wave * per_wave, t_end = min(t_begin + per_wave, Tc);
  double carry = 0.0;
  for (int t0 = t_begin; t0 < t_end; t0 += 64) {
    const int t = t0 + lane;
    double inc = 0.0;
    if (t < t_end) inc = (double)ias_lfo_inc(freq, depth, rate_env[t], control_rate);
    const double sc = wave_incl_scan(inc, lane) + carry;
    carry = __shfl(sc, 63, 64);
    if (t < t_end) s_sum[t] = sc;   // wave-local inclusive sum; the preceding waves' totals are added below
  }
  if (lane == 0) s_wtot[wave] = carry;
  __syncthreads();
  double base = 0.0;
  for (int w = 0; w < wave; ++w) base += s_wtot[w];
  float mode[5];
  ias_lfo_mode(s_q + 3, mode);
  float* out = sig + ((size_t)b * 8 + 6 + l) * Tc;
  for (int t = t_begin + lane; t < t_end; t += 64) {
    const double ph = base + s_sum[t];
    const float arg = ias_add((float)ph, phi);
    const float o = ias_mul(ias_lfo_shape_mix(arg, mode), amp_env[t]);
    out[t] = o;
    if (dbg != nullptr) {
      dbg[((size_t)b * 10 + 6 + l) * Tc + t] = arg;
      dbg[((size_t)b * 10 + 8 + l) * Tc + t] = o;
    }
  }
}

// ctrl[b][j][t] = sum_k w[k][j] * sig_k[t] (4x5 mod matrix, fp64-accumulated dot) and IasVoiceConst[b].
__global__ __launch_bounds__(VOICE_THREADS) void voice_modmix_kernel(
    const float* __restrict__ params01, const float* __restrict__ sig, float* __restrict__ ctrl,
    IasVoiceConst* __restrict__ vconst, float* __restrict__ dbg, int Tc) {
  __shared__ float s_w[20];
  __shared__ float s_p[IAS_NPARAMS];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (tid < 20) s_w[tid] = mapped_param(params01, b, IAS_P_MOD_MATRIX_ADSR_1_TO_VCO_1_PITCH + tid);
  if (blockIdx.x == 0 && tid >= 64 && tid < 64 + IAS_NPARAMS) s_p[tid - 64] = mapped_param(params01, b, tid - 64);
  __syncthreads();
  const float* sb = sig + (size_t)b * 8 * Tc;
  const int t = blockIdx.x * VOICE_THREADS + tid;
  if (t < Tc) {
    const float e0 = sb[t], e1 = sb[Tc + t], l0 = sb[6 * Tc + t], l1 = sb[7 * Tc + t];
    float* out = ctrl + (size_t)b * IAS_NCTRL * Tc;
#pragma unroll
    for (int j = 0; j < IAS_NCTRL; ++j)
      out[j * Tc + t] = ias_dot4_cr(s_w[j], s_w[5 + j], s_w[10 + j], s_w[15 + j], e0, e1, l0, l1);
    if (dbg != nullptr) {
#pragma unroll
      for (int r = 0; r < 6; ++r) dbg[((size_t)b * 10 + r) * Tc + t] = sb[r * Tc + t];
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    const float* p = s_p;
    const float midi_f0 = p[IAS_P_KEYBOARD_MIDI_F0];
    IasVoiceConst vc;
    vc.f0_1 = ias_add(midi_f0, p[IAS_P_VCO_1_TUNING]);
    vc.depth_1 = p[IAS_P_VCO_1_MOD_DEPTH];
    vc.phi_1 = p[IAS_P_VCO_1_INITIAL_PHASE];
    vc.f0_2 = ias_add(midi_f0, p[IAS_P_VCO_2_TUNING]);
    vc.depth_2 = p[IAS_P_VCO_2_MOD_DEPTH];
    vc.phi_2 = p[IAS_P_VCO_2_INITIAL_PHASE];
    vc.kpart = ias_partials_k(midi_f0, vc.depth_2);
    vc.shape = p[IAS_P_VCO_2_SHAPE];
    vc.shape_gain = ias_sub(1.0f, ias_div(vc.shape, 2.0f));
    vc.lvl0 = p[IAS_P_MIXER_VCO_1];
    vc.lvl1 = p[IAS_P_MIXER_VCO_2];
    vc.lvl2 = p[IAS_P_MIXER_NOISE];
    vc.pad[0] = vc.pad[1] = vc.pad[2] = vc.pad[3] = 0.0f;
    vconst[b] = vc;
  }
}

#ifdef IAS_DIAG   // measured slower in the step than the three slim kernels: diagnostic library only
// ---- the whole control pass of a voice in ONE workgroup (round 5) -------------------------------------------------
// voice_env_kernel -> voice_lfo_kernel -> voice_modmix_kernel are three dependent launches whose intermediate rows go through
// HBM, and their pow / cos / fmodf were the device math library's (pow: ~300 fp64-rate instructions): 33 us alone, and
// 15 us of every headline step (0.175 -> 0.159 ms without it, same box, scripts/diag/run_noctrl_ab.sh).  Here one
// workgroup of 1024 threads owns a voice: mapped parameters -> the six envelopes -> both LFOs -> mod matrix, the rows in
// LDS between the phases, the transcendentals by voice_ctrl_math.h (table in LDS; same fp32 values, checked against libm
// on the host and against the oracle bit for bit).  Global outputs are the same as the three kernels': sig [B][8][Tc],
// ctrl [B][5][Tc], vconst [B], and the optional dbg rows.
#define CTLF_THREADS 1024
#define CTLF_WAVES (CTLF_THREADS / 64)
#define CTLF_LFO_WAVES (CTLF_WAVES / 2)      // waves per LFO in the scan phase

__device__ const double g_ctl_tab[IAS_CTL_TAB_DOUBLES] = IAS_CTL_TAB_INIT;

__global__ __launch_bounds__(CTLF_THREADS) void voice_control_fused_kernel(
    const float* __restrict__ params01, float* __restrict__ sig, float* __restrict__ ctrl,
    IasVoiceConst* __restrict__ vconst, float* __restrict__ dbg, int Tc, float control_rate) {
  // dynamic LDS: ctl table | rows [8][Tc] floats (envelopes 0-5, LFO outputs 6-7) | scan sums [2][Tc] doubles
  extern __shared__ __attribute__((aligned(16))) double ctlf_smem[];
  double* s_tab = ctlf_smem;
  double* s_sum = s_tab + IAS_CTL_TAB_DOUBLES;               // [2][Tc]
  float* s_row = reinterpret_cast<float*>(s_sum + 2 * (size_t)Tc);
  __shared__ float s_p[IAS_NPARAMS + 2];
  __shared__ float s_head[12];
  __shared__ float s_mode[2][8];
  __shared__ double s_wtot[2][CTLF_LFO_WAVES];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float eps = (float)IAS_EPS;

  for (int i = tid; i < IAS_CTL_TAB_DOUBLES; i += CTLF_THREADS) s_tab[i] = g_ctl_tab[i];
  if (tid < IAS_NPARAMS) s_p[tid] = mapped_param(params01, b, tid);
  __syncthreads();
  const float note_on = s_p[IAS_P_KEYBOARD_DURATION];
  // per-voice scalars, each on a wave of its own beside the envelope work: flat ramp heads (12 lanes), LFO shape weights
  // (2 x 5 lanes), the audio-rate constants (one lane)
  if (wave == CTLF_WAVES - 1) {
    if (lane < 12) {
      const int a = lane >> 1;
      const float* q = s_p + adsr_base(a);
      IasAdsr e; e.attack = q[0]; e.decay = q[1]; e.sustain = q[2]; e.release = q[3]; e.alpha = q[4];
      const IasAdsrHeads h = ias_adsr_heads(e, note_on, control_rate, eps, s_tab);
      s_head[lane] = (lane & 1) ? h.release_head : h.decay_head;
    }
  } else if (wave == CTLF_WAVES - 2) {
    if (lane < 16 && (lane & 7) < 5) {
      const int l = lane >> 3, k = lane & 7;
      const int qbase = (l == 0) ? IAS_P_LFO_1_FREQUENCY : IAS_P_LFO_2_FREQUENCY;
      s_mode[l][k] = ias_pow_ctl(s_p[qbase + 3 + k], IAS_LFO_EXPONENT_F, s_tab);    // ias_lfo_mode's five powers
    }
  } else if (wave == CTLF_WAVES - 3) {
    if (lane == 0) {
      const float* p = s_p;
      const float midi_f0 = p[IAS_P_KEYBOARD_MIDI_F0];
      IasVoiceConst vc;
      vc.f0_1 = ias_add(midi_f0, p[IAS_P_VCO_1_TUNING]);
      vc.depth_1 = p[IAS_P_VCO_1_MOD_DEPTH];
      vc.phi_1 = p[IAS_P_VCO_1_INITIAL_PHASE];
      vc.f0_2 = ias_add(midi_f0, p[IAS_P_VCO_2_TUNING]);
      vc.depth_2 = p[IAS_P_VCO_2_MOD_DEPTH];
      vc.phi_2 = p[IAS_P_VCO_2_INITIAL_PHASE];
      vc.kpart = ias_partials_k(midi_f0, vc.depth_2);
      vc.shape = p[IAS_P_VCO_2_SHAPE];
      vc.shape_gain = ias_sub(1.0f, ias_div(vc.shape, 2.0f));
      vc.lvl0 = p[IAS_P_MIXER_VCO_1];
      vc.lvl1 = p[IAS_P_MIXER_VCO_2];
      vc.lvl2 = p[IAS_P_MIXER_NOISE];
      vc.pad[0] = vc.pad[1] = vc.pad[2] = vc.pad[3] = 0.0f;
      vconst[b] = vc;
    }
  }
  __syncthreads();

  // ---- phase 1: the six envelopes -> s_row[a][t] (and sig rows 0-5).  The rows the LFOs read (2-5) first.
  float* gsig = sig + (size_t)b * 8 * Tc;
  for (int idx = tid; idx < 6 * Tc; idx += CTLF_THREADS) {
    const int slot = idx / Tc, t = idx - slot * Tc;
    const int a = slot < 4 ? slot + 2 : slot - 4;             // order 2, 3, 4, 5, 0, 1
    const float* q = s_p + adsr_base(a);
    IasAdsr e; e.attack = q[0]; e.decay = q[1]; e.sustain = q[2]; e.release = q[3]; e.alpha = q[4];
    IasAdsrHeads heads; heads.decay_head = s_head[2 * a]; heads.release_head = s_head[2 * a + 1];
    const float v = ias_adsr_headed(t, e, note_on, control_rate, eps, heads, s_tab);
    s_row[a * Tc + t] = v;
    gsig[a * Tc + t] = v;
  }
  // LFO shape weights: m / sum (ias_lfo_mode's normalisation)
  if (tid < 2) {
    float* m = s_mode[tid];
    const float sm = (float)((double)m[0] + (double)m[1] + (double)m[2] + (double)m[3] + (double)m[4]);
    for (int k = 0; k < 5; ++k) m[k] = ias_div(m[k], sm);
  }
  __syncthreads();

  // ---- phase 2: the LFOs.  Waves 0-7: LFO 1, waves 8-15: LFO 2; each wave scans a contiguous stretch in chunks of 64
  // (fp64; the order differs from a sequential loop only below 1e-16 relative, which the rounding to fp32 absorbs).
  {
    const int l = wave / CTLF_LFO_WAVES, w = wave - l * CTLF_LFO_WAVES;
    const int qbase = (l == 0) ? IAS_P_LFO_1_FREQUENCY : IAS_P_LFO_2_FREQUENCY;
    const float freq = s_p[qbase], depth = s_p[qbase + 1], phi = s_p[qbase + 2];
    const float* rate_env = s_row + (4 + l) * Tc;
    const float* amp_env = s_row + (2 + l) * Tc;
    double* sums = s_sum + (size_t)l * Tc;
    const int per_wave = ((Tc + CTLF_LFO_WAVES - 1) / CTLF_LFO_WAVES + 63) / 64 * 64;
    const int t_begin = min(w * per_wave, Tc), t_end = min(t_begin + per_wave, Tc);
    double carry = 0.0;
    for (int t0 = t_begin; t0 < t_end; t0 += 64) {
      const int t = t0 + lane;
      double inc = 0.0;
      if (t < t_end) inc = (double)ias_lfo_inc(freq, depth, rate_env[t], control_rate);
      const double sc = wave_incl_scan(inc, lane) + carry;
      carry = __shfl(sc, 63, 64);
      if (t < t_end) sums[t] = sc;
    }
    if (lane == 0) s_wtot[l][w] = carry;
    __syncthreads();
    double base = 0.0;
    for (int k = 0; k < w; ++k) base += s_wtot[l][k];
    float mode[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) mode[k] = s_mode[l][k];
    for (int t = t_begin + lane; t < t_end; t += 64) {
      const double ph = base + sums[t];
      const float arg = ias_add((float)ph, phi);
      const float o = ias_mul(ias_lfo_shape_mix(arg, mode, s_tab), amp_env[t]);
      s_row[(6 + l) * Tc + t] = o;
      gsig[(6 + l) * Tc + t] = o;
      if (dbg != nullptr) {
        dbg[((size_t)b * 10 + 6 + l) * Tc + t] = arg;
        dbg[((size_t)b * 10 + 8 + l) * Tc + t] = o;
      }
    }
  }
  __syncthreads();

  // ---- phase 3: the 4 x 5 mod matrix -> ctrl[b][j][t]
  float* out = ctrl + (size_t)b * IAS_NCTRL * Tc;
  const float* wm = s_p + IAS_P_MOD_MATRIX_ADSR_1_TO_VCO_1_PITCH;
  for (int t = tid; t < Tc; t += CTLF_THREADS) {
    const float e0 = s_row[t], e1 = s_row[Tc + t], l0 = s_row[6 * Tc + t], l1 = s_row[7 * Tc + t];
#pragma unroll
    for (int j = 0; j < IAS_NCTRL; ++j)
      out[j * Tc + t] = ias_dot4_cr(wm[j], wm[5 + j], wm[10 + j], wm[15 + j], e0, e1, l0, l1);
    if (dbg != nullptr) {
#pragma unroll
      for (int r = 0; r < 6; ++r) dbg[((size_t)b * 10 + r) * Tc + t] = s_row[r * Tc + t];
    }
  }
}
static size_t voice_control_fused_lds(int Tc) {
  return sizeof(double) * (IAS_CTL_TAB_DOUBLES + 2 * (size_t)Tc) + sizeof(float) * 8 * (size_t)Tc;
}

#endif

// -------------------------------------------------------------------- audio rate
#define VOICE_MAXCTRL 320  // control points staged per tile (covers sample rates down to ~6 kHz)
#define VOICE_SPIN_LIMIT (1u << 24)
#ifndef VOICE_SPIN_SLEEP
#define VOICE_SPIN_SLEEP 8   // s_sleep units (64 clocks) between two polls of the look-back (2: 98-100 us, 8: 96-97.5, 32: 96-99; same box)
#endif
#define VOICE_READY_BIT 0x8000000000000000ull
#ifndef VOICE_MIN_WAVES
#define VOICE_MIN_WAVES 3    // waves per SIMD the audio kernel is compiled for (<= 168 VGPRs: two tiles' increments)
#endif
#ifndef VOICE_GROUP
#define VOICE_GROUP 4        // samples the compiler may interleave (a scheduling barrier after each group)
#endif
#ifndef VOICE_NCOUNTERS
#define VOICE_NCOUNTERS 8   // ticket counters (one 128-byte line each): counter c hands out the tiles of the voices b % 8 == c
#endif
#define VOICE_SYNC_HEAD_BYTES ((VOICE_NCOUNTERS + 1) * 128)   // counters, then the line of the status word
#define VOICE_MATH_CR 0     // pitch exp2 correctly rounded (oracle math "cr"): the default and the tested contract
#define VOICE_TAB_DOUBLES ((IAS_EXP2_TAB_LEN + 1) / 2 * 2)   // table padded to 16 bytes
#define VOICE_MATH_HW 1     // pitch exp2 = v_exp_f32 (the reference's own precision: fp32 exp2, <= 1 ulp); A/B only

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ const double g_exp2_tab[IAS_EXP2_TAB_LEN] = IAS_EXP2_TAB_INIT;

#ifdef VOICE_STAMPS
// Diagnostic build only (scripts/diag): per-tile s_memtime stamps of wave 0 / wave 3, written to a buffer of their own.
__device__ unsigned long long* g_voice_stamps = nullptr;
extern "C" int ias_voice_debug_set_stamps(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_voice_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#define VSTAMP(slot) do { if (g_voice_stamps && have_cur && lane == 0) { unsigned long long* sp_ = g_voice_stamps + ((size_t)(cur.b * ntiles + cur.tile) * 4 + wave) * 12; sp_[slot] = __builtin_amdgcn_s_memtime(); if ((slot) == 0) sp_[10] = __builtin_amdgcn_s_memrealtime(); if ((slot) == 6) sp_[11] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define VSTAMP(slot) do {} while (0)
#endif

typedef __attribute__((address_space(3))) void voice_lds_void;
typedef const __attribute__((address_space(3))) char voice_lds_cchar;       // 32-bit LDS addresses: one v_mad / shift-add
typedef float voice_f2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) voice_f2 voice_lds_cfloat2;
typedef const __attribute__((address_space(3))) double voice_lds_cdouble;

// ias_exp2_cr_tab (voice_math.h) with the table addressed as LDS bytes: tab_biased = LDS address of IAS_EXP2_TAB[0]
// minus 8 * (0x4B400000 + IAS_EXP2_TAB_MIN), so that the entry of u = 1.5 * 2^23 + m is at tab_biased + 8 * bits(u).
__device__ __forceinline__ float voice_exp2_cr_lds(float t, unsigned tab_biased) {
  const float u = fmaf(t, 256.0f, 12582912.0f);
  const float mf = u - 12582912.0f;
  const float r = fmaf(mf, -0.00390625f, t);
  const double tv = *(voice_lds_cdouble*)(uintptr_t)(tab_biased + ((unsigned)__float_as_uint(u) << 3));
  const double rd = (double)r;
  double p = IAS_EXP2_C4;
  p = fma(p, rd, IAS_EXP2_C3);
  p = fma(p, rd, IAS_EXP2_C2);
  p = fma(p, rd, IAS_EXP2_C1);
  p = fma(p, rd, 1.0);
  return (float)(p * tv);
}
typedef const __attribute__((address_space(1))) void voice_glb_void;
// LDS address of control row k: k * VOICE_CTRL_ROW + biased base, ONE v_mad_u32_u24 (left to itself the compiler turns
// the same expression into v_and + v_subrev + a 64-bit v_mad_u64_u32 per sample and phase).
__device__ __forceinline__ voice_lds_cchar* voice_ctrl_row(float real, unsigned ctrl_biased) {
  unsigned addr;
  const int k = (int)real;                       // real >= 0: trunc = floor
  asm("v_mad_u32_u24 %0, %1, 40, %2" : "=v"(addr) : "v"(k), "s"(ctrl_biased));
  return (voice_lds_cchar*)(uintptr_t)addr;
}
static_assert(IAS_NCTRL * 8 == 40, "voice_ctrl_row: the row stride is an immediate");
// Linear upsample of one control signal: fl(w0 a + fl(w1 b)), the ONE fused multiply-add torch's CPU kernel
// (nn.Upsample(mode="linear", align_corners=True) -> ATen cpu_upsample_linear, x0 * w0 + x1 * w1 contracted by the
// compiler) evaluates: bit-equal to the op on this build's torch (tests/test_voice_math_cpu.py).
__device__ __forceinline__ float voice_lerp(float w0, float w1, voice_f2 q) { return fmaf(w0, q.x, w1 * q.y); }
// Phase increment of one VCO sample, fl(fl(2 pi hz) / sr) with hz = fl(440 fl(2^fl((c - 69) / 12))): bit-identical to
// ias_vco_inc (MATH_CR).  FMA_DIV: the sample rate is one ias_div_fma is verified for (sr_f = the rate, sr_r = fl32 of
// its reciprocal), otherwise the fp64 reciprocal product is used.
template <int MATH, bool FMA_DIV>
__device__ __forceinline__ float voice_inc(float f0, float depth, float pm, unsigned tab_biased, double inv_sample_rate,
                                           float sr_f, float sr_r) {
  float c = f0 + depth * pm;
  c = fminf(fmaxf(c, 0.0f), 127.0f);
  const float t = ias_div_fma(c - 69.0f, 12.0f, 1.0f / 12.0f);
  const float e = (MATH == VOICE_MATH_CR) ? voice_exp2_cr_lds(t, tab_biased) : __builtin_amdgcn_exp2f(t);
  const float w = (float)IAS_TWO_PI_D * (440.0f * e);
  return FMA_DIV ? ias_div_fma(w, sr_f, sr_r) : ias_div_by_recip(w, inv_sample_rate);
}

// Single pass over the row with a chained scan across tiles ("decoupled look-back"), persistent workgroups:
//   ticket  -> (tile, voice), tile-major; a workgroup keeps taking tickets until they run out.  Tickets are handed
//              out in order, so every predecessor tile of the same voice has been taken by a running workgroup
//              when a workgroup starts on its own (no dependence on dispatch order or placement: a workgroup only
//              ever waits for smaller tickets, and those publish before they wait).
//   layout  -> a lane owns 16 CONSECUTIVE samples of the tile (lane order = time order), so the tile-local prefix
//              of a sample is (wave scan of the lane totals) + a running sum inside the lane: two wave scans per
//              tile and VCO instead of one per 4-sample chunk, and no array of partial sums.
//   phase A -> the lane's 2 x 16 phase increments (registers) and their fp64 totals; wave scan; wave totals to LDS.
//              Sums of fp32 increments below 2^19 are exact in fp64, so the order of summation is irrelevant and
//              the result is bit-identical to the sequential double accumulation of the oracle.
//   publish -> one 8-byte write-through store per VCO: the sum's bits with the sign bit as READY flag
//              (sums are >= 0).  The datum is its own flag (MI355X guide, Guideline 16 form R2).
//   wait    -> wave 0 polls the predecessors' words with relaxed agent-scope loads (L1 bypass),
//              bounded spins, and adds them up: the tile's carry-in.  An expired wait turns the carry into NaN.
//   phase B -> running fp64 phase per lane, round to fp32, + phi, oscillators, VCAs, mixer, row peak.
// The pitch exp2 reads a 2^(i/256) table (fp64, 21.7 KB) that each workgroup copies to LDS once.
#define VOICE_CTRL_ROW (IAS_NCTRL * 8)    // bytes of one control point in LDS: five (c[i], c[i+1]) pairs, interleaved

// Tickets.  One counter word serves ~88 returning atomics per microsecond (MI355X guide, "dequeue"): the 5632 tiles of a
// B = 128 render would keep a single counter busy for 64 us, and a 1024-workgroup start for 12 us.  The tiles are
// therefore handed out by VOICE_NCOUNTERS counters: counter c owns the voices b with b % NC == c, its n-th ticket is
// tile n / nv_c of voice (n % nv_c) * NC + c -- tile-major per counter, so a tile's predecessors (same voice, smaller
// tiles) are always smaller tickets OF THE SAME COUNTER, i.e. already taken by a running workgroup when it is taken.
// A workgroup starts at counter blockIdx % NC and moves on when a counter is exhausted.  -> (c << 24) | n, or -1.
__device__ __forceinline__ int voice_take_ticket(unsigned int* counters, int& c, int& tried, int ntiles, int nvoices) {
  while (tried < VOICE_NCOUNTERS) {
    const int nv_c = (nvoices - c + VOICE_NCOUNTERS - 1) / VOICE_NCOUNTERS;
    const unsigned n = __hip_atomic_fetch_add((gu32*)(counters + c * 32), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)n < nv_c * ntiles) return (c << 24) | (int)n;
    c = (c + 1) % VOICE_NCOUNTERS;
    ++tried;
  }
  return -1;
}
// One tile = 4096 consecutive samples of one voice; all fields are workgroup-uniform.
struct VoiceTile {
  int tile, b, j_tile, c_lo, ncp;
  bool fast;   // a full tile of a row whose length is a multiple of 4 samples (LDS-DMA / whole-KB path)
};
__device__ __forceinline__ VoiceTile voice_tile_of(int ticket, int nvoices, int T, int Tc, float scale) {
  VoiceTile t;
  const int c = ticket >> 24, n = ticket & 0xffffff;
  const int nv_c = (nvoices - c + VOICE_NCOUNTERS - 1) / VOICE_NCOUNTERS;
  t.tile = n / nv_c;
  t.b = (n - t.tile * nv_c) * VOICE_NCOUNTERS + c;
  t.j_tile = t.tile * VOICE_TILE;
  const int j_last = min(t.j_tile + VOICE_TILE, T) - 1;
  int i0, i1; float w0, w1;
  ias_interp_pos(t.j_tile, scale, Tc, i0, i1, w0, w1);
  t.c_lo = i0;
  ias_interp_pos(j_last, scale, Tc, i0, i1, w0, w1);
  t.ncp = i1 - t.c_lo + 1;                       // host guarantees ncp <= maxctrl
  t.fast = (t.j_tile + VOICE_TILE <= T) && (T & 3) == 0;
  return t;
}
// The control points a tile interpolates between -> LDS rows [ncp][IAS_NCTRL] of (c[i], c[min(i + 1, Tc - 1)]) pairs:
// one ds_read_b64 at an immediate offset from the point's address fetches both ends of a lerp.
__device__ __forceinline__ void voice_stage_ctrl(char* dst, const float* __restrict__ ctrl, const VoiceTile& t, int Tc) {
  const float* cb = ctrl + (size_t)t.b * IAS_NCTRL * Tc;
  for (int i = threadIdx.x; i < IAS_NCTRL * t.ncp; i += AUDIO_THREADS) {
    const int k = i / t.ncp, c = i - k * t.ncp;
    const float* crow = cb + k * Tc;
    reinterpret_cast<float2*>(dst)[c * IAS_NCTRL + k] = make_float2(crow[t.c_lo + c], crow[min(t.c_lo + c + 1, Tc - 1)]);
  }
}

// ---- phase A of a tile: the lane's 2 x 16 phase increments (registers), their fp64 totals, the wave scan of the
// totals (exclusive lane prefix ex1 / ex2) and the wave totals -> wsum[vco][wave].  A lane owns 16 CONSECUTIVE
// samples (lane order = time order): two wave scans per tile and VCO, no array of partial sums.  Sums of fp32
// increments below 2^19 are exact in fp64, so any summation order gives the bits of the oracle's sequential
// double accumulation.
template <int MATH, bool FMA_DIV, bool FAST>
__device__ __forceinline__ void voice_phase_a(const char* s_ctrl, const double* s_tab_, const IasVoiceConst& vc,
                                              const VoiceTile& t, int T, double inv_sample_rate, float sr_f, float sr_r,
                                              float scale, float (&inc1)[VOICE_SPT], float (&inc2)[VOICE_SPT],
                                              double& ex1, double& ex2, double (*wsum)[AUDIO_WAVES]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = t.j_tile + tid * VOICE_SPT;             // the lane's first sample
  const float jf0 = (float)j0;
  // 32-bit LDS addresses: control rows indexed by the absolute control index, exp2 table by the bits of 1.5 * 2^23 + m
  const unsigned ctrl_biased = (unsigned)(uintptr_t)(voice_lds_cchar*)s_ctrl - (unsigned)(t.c_lo * VOICE_CTRL_ROW);
  const unsigned s_tab = (unsigned)(uintptr_t)(voice_lds_cchar*)s_tab_ - (0x4B400000u + (unsigned)IAS_EXP2_TAB_MIN) * 8u;
  double tot1 = 0.0, tot2 = 0.0;
#pragma unroll
  for (int e = 0; e < VOICE_SPT; ++e) {
    // interpolation position (ias_interp_pos_fast): k = trunc(real) (real >= 0), w1 = real - k = fract(real), exactly
    const float real = scale * (jf0 + (float)e);
    const float w1 = __builtin_amdgcn_fractf(real), w0 = 1.0f - w1;
    voice_lds_cchar* cp = voice_ctrl_row(real, ctrl_biased);
    const voice_f2 q1 = *(voice_lds_cfloat2*)cp, q2 = *(voice_lds_cfloat2*)(cp + 16);
    const float pm1 = voice_lerp(w0, w1, q1), pm2 = voice_lerp(w0, w1, q2);
    float a = voice_inc<MATH, FMA_DIV>(vc.f0_1, vc.depth_1, pm1, s_tab, inv_sample_rate, sr_f, sr_r);
    float d = voice_inc<MATH, FMA_DIV>(vc.f0_2, vc.depth_2, pm2, s_tab, inv_sample_rate, sr_f, sr_r);
    if (!FAST && j0 + e >= T) { a = 0.0f; d = 0.0f; }
    inc1[e] = a; inc2[e] = d;
    tot1 += (double)a; tot2 += (double)d;
    // bounds the scheduler's interleaving of the 16 samples (it otherwise keeps all of them in flight)
    if ((e & (VOICE_GROUP - 1)) == VOICE_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
  }
  const double in1 = wave_incl_scan(tot1, lane), in2 = wave_incl_scan(tot2, lane);
  if (lane == 63) { wsum[0][wave] = in1; wsum[1][wave] = in2; }
  ex1 = in1 - tot1; ex2 = in2 - tot2;   // exact
}

// ---- phase B of a tile: running fp64 phase per lane (carry-in + earlier waves + ex + the lane's own increments), round
// to fp32, + phi, oscillators, VCAs, mixer, row peak.  FAST tiles: the noise was brought into the wave's 4 KB LDS block
// by LDS-DMA and the audio leaves through the same block, so every global access is a whole 1 KB per wave-instruction
// although a lane owns 16 consecutive samples.  Block layout: the lane's 16 samples are its own 64 bytes, the four
// 16-byte granules XOR-permuted by (lane >> 2) & 3 -- conflict-free for the lane-consecutive ds_read/write_b128 and
// lane-linear (the DMA's destination order) once the SOURCE granule of DMA lane i is i ^ (i >> 4).
template <bool FAST>
__device__ __forceinline__ void voice_phase_b(const char* s_ctrl, float* s_stage, const IasVoiceConst& vc, const VoiceTile& t,
                                              const float* __restrict__ nrow, float* __restrict__ arow, int T, float scale,
                                              const float (&inc1)[VOICE_SPT], const float (&inc2)[VOICE_SPT], double ex1,
                                              double ex2, const double* s_carry, const double (*wsum)[AUDIO_WAVES],
                                              float& pk) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j0 = t.j_tile + tid * VOICE_SPT;
  float jf0 = (float)j0;
  asm volatile("" : "+v"(jf0));   // the interpolation positions are recomputed: kept from phase A they cost 32+ VGPRs
  float* stage = s_stage + wave_u * (64 * VOICE_SPT);   // this wave's 4 KB block
  const int j_wave = t.j_tile + wave_u * (64 * VOICE_SPT);
  const int perm = (lane >> 2) & 3;
  const unsigned ctrl_biased = (unsigned)(uintptr_t)(voice_lds_cchar*)s_ctrl - (unsigned)(t.c_lo * VOICE_CTRL_ROW);

  const float kexp = -1.4426950408889634f * vc.kpart;
  double run1 = s_carry[0] + ex1, run2 = s_carry[1] + ex2;   // all exact
  for (int w = 0; w < wave_u; ++w) { run1 += wsum[0][w]; run2 += wsum[1][w]; }
  float o[4], nz[4];
  float4 nz_next = make_float4(0.f, 0.f, 0.f, 0.f);
  auto load_noise4 = [&](int jq) -> float4 {   // plain path only
    return make_float4(jq < T ? nrow[jq] : 0.0f, jq + 1 < T ? nrow[jq + 1] : 0.0f, jq + 2 < T ? nrow[jq + 2] : 0.0f,
                       jq + 3 < T ? nrow[jq + 3] : 0.0f);
  };
  if (!FAST) nz_next = load_noise4(j0);
#pragma unroll
  for (int e = 0; e < VOICE_SPT; ++e) {
    if ((e & 3) == 0) {
      if (FAST) {
        const float4 v = *reinterpret_cast<const float4*>(stage + lane * VOICE_SPT + 4 * ((e >> 2) ^ perm));
        nz[0] = v.x; nz[1] = v.y; nz[2] = v.z; nz[3] = v.w;
      } else {
        nz[0] = nz_next.x; nz[1] = nz_next.y; nz[2] = nz_next.z; nz[3] = nz_next.w;
        if (e + 4 < VOICE_SPT) nz_next = load_noise4(j0 + e + 4);
      }
    }
    const float real = scale * (jf0 + (float)e);
    const float w1 = __builtin_amdgcn_fractf(real);
    voice_lds_cchar* cp = voice_ctrl_row(real, ctrl_biased);
    // the amplitude rows hold (level x c[i], level x (c[i + 1] - c[i])): one multiply-add each (voice_ctrl_put)
    const voice_f2 qa = *(voice_lds_cfloat2*)(cp + 8), qb = *(voice_lds_cfloat2*)(cp + 24), qn = *(voice_lds_cfloat2*)(cp + 32);
    const float amp1 = fmaf(w1, qa.y, qa.x), amp2 = fmaf(w1, qb.y, qb.x), ampn = fmaf(w1, qn.y, qn.x);
    run1 += (double)inc1[e]; run2 += (double)inc2[e];
    const float arg1 = (float)run1 + vc.phi_1, arg2 = (float)run2 + vc.phi_2;
    float s2, c2, sgn;
    voice_sincos_sgn(arg2, s2, c2, sgn);   // sin = sgn s2, cos = sgn c2, sgn = +-1
    // square = tanh(k sin / 2) is odd in sin: sgn tanh(k s2 / 2), the tanh of the SIGNED argument (kexp = -k log2 e)
    const float th = voice_tanh_half_of_exp2arg(kexp * s2);
    // vco_2 = gain x square x (1 + shape cos) x amplitude (gain and mixer level are inside amp2):
    // sgn th (1 + shape sgn c2) = th (sgn + shape c2) -- the sign costs nothing beyond its two multiply-adds;
    // vco_1 = cos x amplitude, mix = vco_1 + vco_2 + noise x amplitude: fused multiply-adds (the amplitude path is held
    // to 1e-4, not to the bit)
    const float v2 = th * fmaf(vc.shape, c2, sgn);
    float om = voice_cos(arg1) * amp1;
    om = fmaf(v2, amp2, om);
    om = fmaf(nz[e & 3], ampn, om);
    if (FAST || j0 + e < T) pk = fmaxf(pk, fabsf(om));
    o[e & 3] = om;
    if ((e & 3) == 3) {
      if (FAST) {
        *reinterpret_cast<float4*>(stage + lane * VOICE_SPT + 4 * ((e >> 2) ^ perm)) = make_float4(o[0], o[1], o[2], o[3]);
      } else {
        const int jq = j0 + e - 3;
#pragma unroll
        for (int x = 0; x < 4; ++x) if (jq + x < T) arow[jq + x] = o[x];
      }
      if ((e & (VOICE_GROUP - 1)) == VOICE_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (FAST) {
    // the wave's 4 KB leave as four whole 1 KB stores (LDS operations of one wave execute in order: no barrier)
    const int dst = 4 * (lane ^ (lane >> 4));
#pragma unroll
    for (int q = 0; q < VOICE_SPT / 4; ++q)
      *reinterpret_cast<float4*>(arow + j_wave + q * 256 + dst) = *reinterpret_cast<const float4*>(stage + q * 256 + lane * 4);
  }
}

// element i (< IAS_NCTRL * ncp) of a tile's control stage: signal k = i / ncp, point c = i % ncp
__device__ __forceinline__ float2 voice_ctrl_elem(const float* __restrict__ ctrl, const VoiceTile& t, int Tc, int i) {
  const int k = i / t.ncp, c = i - k * t.ncp;
  const float* crow = ctrl + ((size_t)t.b * IAS_NCTRL + k) * Tc;
  return make_float2(crow[t.c_lo + c], crow[min(t.c_lo + c + 1, Tc - 1)]);
}
// Pitch signals (k = 0, 2) are staged as the pair (c[i], c[i + 1]): phase A evaluates torch's upsample form on them, bit for
// bit.  The three AMPLITUDE signals (k = 1 vco_1, 3 vco_2, 4 noise) do not feed a phase -- an error there is not amplified,
// and the audio is held to 1e-4, not to the bit -- so what is staged for them is what phase B can use in ONE multiply-add
// per sample: (g c[i], g c[i + 1] - g c[i]) with the voice's mixer level g folded in (vco_2: level x the shape gain).
// amplitude(t) = fma(w1, difference, value) then already carries the level: five multiplies and three adds per sample less
// than the left-to-right chain of the "cr" statement, a few ulp away from it (tests: audio within 1e-4 of the oracle;
// measured max |delta| in DESIGN.md).
__device__ __forceinline__ void voice_ctrl_put(char* dst, const VoiceTile& t, int i, float2 v, const IasVoiceConst& vc) {
  const int k = i / t.ncp, c = i - k * t.ncp;
  if (k == 1 || k == 3 || k == 4) {
    const float g = k == 1 ? vc.lvl0 : (k == 3 ? vc.lvl1 * vc.shape_gain : vc.lvl2);
    const float a = g * v.x;
    v = make_float2(a, g * v.y - a);
  }
  reinterpret_cast<float2*>(dst)[c * IAS_NCTRL + k] = v;
}

// Single pass over every row with a chained scan across tiles ("decoupled look-back"), persistent workgroups, three
// tiles in flight per workgroup (AFTER: ticket + control points being fetched; NEXT: phase A; CURRENT: phase B):
//   ticket  -> (tile, voice), tile-major per counter; a workgroup keeps taking tickets until they run out.  Tickets are
//              handed out in order and a workgroup only ever WAITS for smaller tickets, whose holders have either
//              published or are computing their phase A without waiting for anybody -- no dependence on dispatch order
//              or placement.
//   loop    -> phase A of the NEXT tile (increments, sums; needs nothing from other workgroups), publish its sums, then
//              look-back + phase B of the CURRENT tile.  A tile's sums are thus published one phase B (~10 us of wave
//              time) before its own audio is due, and the words a tile polls were requested before phase A of the next
//              one: the look-back round trip and the predecessors' publish delay are off the critical path (with one
//              tile in flight the waves spent 18 % of a tile waiting for them).  The ticket after next is requested
//              before phase A and read after it; its control points are fetched during phase B.
//   publish -> one 8-byte write-through store per VCO: the sum's bits with the sign bit as READY flag (sums are >= 0).
//              The datum is its own flag (MI355X guide, Guideline 16 form R2).
//   wait    -> one wave polls the predecessors' words with relaxed agent-scope loads (L1 bypass), bounded spins, and
//              adds them up: the tile's carry-in.  An expired wait turns the carry (hence the tile's audio) into NaN.
//   roles   -> wave 0 takes tickets, wave 3 looks back, wave 2 publishes: no wave carries all the serial work.
// The pitch exp2 reads a 2^(i/256) table (fp64, 21.7 KB) that each workgroup copies to LDS once.
#define VOICE_WAVE_LOOKBACK (AUDIO_WAVES - 1)
#define VOICE_WAVE_PUBLISH (AUDIO_WAVES - 2)
template <int MATH, bool FMA_DIV>
__global__ __launch_bounds__(AUDIO_THREADS, VOICE_MIN_WAVES) __attribute__((amdgpu_num_vgpr(76))) void voice_audio_kernel(
    const float* __restrict__ ctrl, const IasVoiceConst* __restrict__ vconst,
    const float* __restrict__ noise, float* __restrict__ audio, unsigned long long* agg /* [B][ntiles][2] */,
    unsigned int* ticket_status /* VOICE_NCOUNTERS ticket counters (32 words apart), then the spin-timeout flag */,
    unsigned* __restrict__ rowpeak, int T, int Tc, int ntiles, int nvoices, double inv_sample_rate, float sr_f,
    float sr_r, float scale, int maxctrl) {
  // dynamic LDS: exp2 table | per-wave noise / audio blocks [AUDIO_WAVES][1024] floats | two control stages
  extern __shared__ __attribute__((aligned(16))) double dyn_smem_d[];
  double* s_tab = dyn_smem_d;
  float* s_stage = reinterpret_cast<float*>(dyn_smem_d + VOICE_TAB_DOUBLES);
  char* s_ctrl0 = reinterpret_cast<char*>(s_stage + VOICE_TILE);
  const int ctrl_bytes = maxctrl * VOICE_CTRL_ROW;
  __shared__ double s_wsum[2][2][AUDIO_WAVES];   // [slot][vco][wave]
  __shared__ double s_carry[2];
  __shared__ float s_max[AUDIO_WAVES];
  __shared__ int s_ticket;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int my_counter = blockIdx.x % VOICE_NCOUNTERS, counters_tried = 0;   // used by thread 0 only
  if (tid == 0) s_ticket = voice_take_ticket(ticket_status, my_counter, counters_tried, ntiles, nvoices);
  if (MATH == VOICE_MATH_CR)
    for (int i = tid; i < IAS_EXP2_TAB_LEN; i += AUDIO_THREADS) s_tab[i] = g_exp2_tab[i];
  __syncthreads();

  VoiceTile cur = {}, next = {};
  bool have_cur = false;
  // the first tile: ticket and control points fetched here, synchronously
  bool have_next = __builtin_amdgcn_readfirstlane(s_ticket) >= 0;
  float2 pre = make_float2(0.f, 0.f);           // this thread's element of the control stage of `next`
  if (have_next) {
    next = voice_tile_of(__builtin_amdgcn_readfirstlane(s_ticket), nvoices, T, Tc, scale);
    if (tid < IAS_NCTRL * next.ncp) pre = voice_ctrl_elem(ctrl, next, Tc, tid);
  }
  int slot = 0;                                 // s_wsum / control stage of the CURRENT tile
  float incC1[VOICE_SPT], incC2[VOICE_SPT];     // the current tile's increments (phase B), the next tile's (phase A)
  float incN1[VOICE_SPT], incN2[VOICE_SPT];
  double exC1 = 0.0, exC2 = 0.0, exN1 = 0.0, exN2 = 0.0;
#pragma unroll
  for (int e = 0; e < VOICE_SPT; ++e) { incC1[e] = incC2[e] = 0.0f; }

  while (have_cur || have_next) {
    gu64* row = (gu64*)(agg + ((size_t)cur.b * ntiles) * 2);
    VSTAMP(0);
    // (the put comes first: its wait for the control points' loads would otherwise cover the requests issued below)
    if (have_next) {
      // the next tile's control points (fetched during the previous phase B) -> LDS
      char* dst = s_ctrl0 + (slot ^ 1) * ctrl_bytes;
      const IasVoiceConst vcp = vconst[next.b];
      if (tid < IAS_NCTRL * next.ncp) voice_ctrl_put(dst, next, tid, pre, vcp);
      for (int i = tid + AUDIO_THREADS; i < IAS_NCTRL * next.ncp; i += AUDIO_THREADS)   // windows wider than 51 points
        voice_ctrl_put(dst, next, i, voice_ctrl_elem(ctrl, next, Tc, i), vcp);
    }
    unsigned long long early1 = VOICE_READY_BIT, early2 = VOICE_READY_BIT;
    if (have_cur) {
      // the current tile's look-back words and noise are requested now and examined after phase A of the next tile
      if (wave == VOICE_WAVE_LOOKBACK && lane < cur.tile) {
        early1 = __hip_atomic_load(row + lane * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        early2 = __hip_atomic_load(row + lane * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (cur.fast) {
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const float* nsrc = noise + (size_t)cur.b * T + cur.j_tile + wave_u * (64 * VOICE_SPT) + 4 * (lane ^ (lane >> 4));
#pragma unroll
        for (int q = 0; q < VOICE_SPT / 4; ++q)
          __builtin_amdgcn_global_load_lds((voice_glb_void*)(nsrc + q * 256),
                                           (voice_lds_void*)(s_stage + wave_u * (64 * VOICE_SPT) + q * 256), 16, 0, 0);
      }
    }
    __syncthreads();   // (1) the next tile's control points are staged
    VSTAMP(1);
    unsigned after_raw = 0;
    int after_c = 0;
    if (have_next) {
      // the ticket after next: requested now, resolved after phase A
      if (tid == 0) {
        after_c = my_counter;
        after_raw = __hip_atomic_fetch_add((gu32*)(ticket_status + after_c * 32), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const IasVoiceConst vcn = vconst[next.b];
#ifdef VOICE_SKIP_A
      if (false)
#else
      if (next.fast)
#endif
        voice_phase_a<MATH, FMA_DIV, true>(s_ctrl0 + (slot ^ 1) * ctrl_bytes, s_tab, vcn, next, T, inv_sample_rate, sr_f, sr_r,
                                           scale, incN1, incN2, exN1, exN2, s_wsum[slot ^ 1]);
#ifndef VOICE_ANALYZE_FULL_ONLY
      else
        voice_phase_a<MATH, FMA_DIV, false>(s_ctrl0 + (slot ^ 1) * ctrl_bytes, s_tab, vcn, next, T, inv_sample_rate, sr_f, sr_r,
                                            scale, incN1, incN2, exN1, exN2, s_wsum[slot ^ 1]);
#endif
      if (tid == 0) {
        const int nv_c = (nvoices - after_c + VOICE_NCOUNTERS - 1) / VOICE_NCOUNTERS;
        int tk = (after_c << 24) | (int)after_raw;
        if ((int)after_raw >= nv_c * ntiles) {      // that counter is exhausted: try the others (end of the launch only)
          my_counter = (my_counter + 1) % VOICE_NCOUNTERS;
          ++counters_tried;
          tk = voice_take_ticket(ticket_status, my_counter, counters_tried, ntiles, nvoices);
        }
        s_ticket = tk;
      }
    } else if (tid == 0) {
      s_ticket = -1;
    }
    VSTAMP(2);
    // ---- the current tile's carry-in: its predecessors' sums
    if (have_cur && wave == VOICE_WAVE_LOOKBACK) {
      double a1 = 0.0, a2 = 0.0;
      bool timeout = false;
      for (int t0 = 0; t0 < cur.tile; t0 += 64) {
        const int t = t0 + lane;
        unsigned long long x1 = VOICE_READY_BIT, x2 = VOICE_READY_BIT;   // lanes without a predecessor: ready
        if (t0 == 0) { x1 = early1; x2 = early2; }
        else if (t < cur.tile) { x1 = 0; x2 = 0; }
        unsigned spins = 0;
        bool ok = (x1 & x2 & VOICE_READY_BIT) != 0;
        while (!__all(ok)) {                        // wave-uniform loop condition
          if (++spins > VOICE_SPIN_LIMIT) { timeout = true; break; }
          if (!ok) {
            x1 = __hip_atomic_load(row + t * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x2 = __hip_atomic_load(row + t * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (x1 & x2 & VOICE_READY_BIT) != 0;
          }
          if (!__all(ok)) __builtin_amdgcn_s_sleep(VOICE_SPIN_SLEEP);
        }
        if (t < cur.tile) {
          a1 += __longlong_as_double((long long)(x1 & ~VOICE_READY_BIT));
          a2 += __longlong_as_double((long long)(x2 & ~VOICE_READY_BIT));
        }
      }
      a1 = wave_sum(a1); a2 = wave_sum(a2);
      if (lane == 0) {
        // an expired wait never continues with partial carries: the tile's phases (hence its audio) become NaN,
        // and the status word says why
        if (timeout) {
          a1 = a2 = __longlong_as_double(0x7ff8000000000000ll);
          __hip_atomic_store((gu32*)ticket_status + VOICE_NCOUNTERS * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store((gu32*)ticket_status - 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sticky (VoiceWs::off_sticky)
        }
        s_carry[0] = a1; s_carry[1] = a2;
      }
    }
    VSTAMP(3);
    __syncthreads();   // (2) the next tile's wave sums, the current tile's carry, the noise blocks, the ticket after next
    VSTAMP(4);
    if (have_next && wave == VOICE_WAVE_PUBLISH && lane < 2) {
      // publish the next tile's sums: from here on nobody waits for this workgroup on its account
      double a = 0.0;
      for (int w = 0; w < AUDIO_WAVES; ++w) a += s_wsum[slot ^ 1][lane][w];
      gu64* nrow_agg = (gu64*)(agg + ((size_t)next.b * ntiles) * 2);
      __hip_atomic_store(nrow_agg + next.tile * 2 + lane, (unsigned long long)__double_as_longlong(a) | VOICE_READY_BIT,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the tile after next: its control points are requested now and land during phase B
    const int t_after = __builtin_amdgcn_readfirstlane(s_ticket);
    const bool have_after = t_after >= 0;
    VoiceTile after = {};
    if (have_after) {
      after = voice_tile_of(t_after, nvoices, T, Tc, scale);
      if (tid < IAS_NCTRL * after.ncp) pre = voice_ctrl_elem(ctrl, after, Tc, tid);
    }
    if (have_cur) {
      const IasVoiceConst vc = vconst[cur.b];
      float pk = 0.0f;
#ifdef VOICE_SKIP_B
      if (false)
#else
      if (cur.fast)
#endif
        voice_phase_b<true>(s_ctrl0 + slot * ctrl_bytes, s_stage, vc, cur, noise + (size_t)cur.b * T, audio + (size_t)cur.b * T,
                            T, scale, incC1, incC2, exC1, exC2, s_carry, s_wsum[slot], pk);
#ifndef VOICE_ANALYZE_FULL_ONLY
      else
        voice_phase_b<false>(s_ctrl0 + slot * ctrl_bytes, s_stage, vc, cur, noise + (size_t)cur.b * T, audio + (size_t)cur.b * T,
                             T, scale, incC1, incC2, exC1, exC2, s_carry, s_wsum[slot], pk);
#endif
      pk = wave_max(pk);
      if (lane == 0) s_max[wave] = pk;
    }
    VSTAMP(5);
    __syncthreads();   // (3) the current tile is done with its control stage, wave sums and carry; s_max is set
    VSTAMP(6);
    if (have_cur && tid == 0) {
      float m = s_max[0];
      for (int w = 1; w < AUDIO_WAVES; ++w) m = fmaxf(m, s_max[w]);
      atomicMax(rowpeak + cur.b, __float_as_uint(m));  // m >= 0: uint order == float order
    }
    // rotate: next -> current, after -> next
    cur = next;
    have_cur = have_next;
    next = after;
    have_next = have_after;
    slot ^= 1;
#pragma unroll
    for (int e = 0; e < VOICE_SPT; ++e) { incC1[e] = incN1[e]; incC2[e] = incN2[e]; }
    exC1 = exN1; exC2 = exN2;
  }
}

__global__ __launch_bounds__(256) void voice_normalize_kernel(float* __restrict__ audio,
                                                              const unsigned* __restrict__ rowpeak,
                                                              int T, int nvec_per_row) {
  const int b = blockIdx.y;
  const float peak = __uint_as_float(rowpeak[b]);
  if (!(peak > 1.0f)) return;
  float* row = audio + (size_t)b * T;
  if ((T & 3) == 0) {
    float4* r4 = reinterpret_cast<float4*>(row);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nvec_per_row; i += gridDim.x * blockDim.x) {
      float4 v = r4[i];
      v.x = ias_div(v.x, peak); v.y = ias_div(v.y, peak);
      v.z = ias_div(v.z, peak); v.w = ias_div(v.w, peak);
      r4[i] = v;
    }
  } else {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < T; i += gridDim.x * blockDim.x)
      row[i] = ias_div(row[i], peak);
  }
}

// ------------------------------------------------------------------------ C ABI
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct VoiceWs {
  size_t off_ctrl, off_vconst, off_env, off_sticky, off_sync, sync_bytes, off_agg, off_peak, total;
  int ntiles;
};
static VoiceWs voice_ws_layout(int B, int T, int Tc) {
  VoiceWs w;
  w.ntiles = (T + VOICE_TILE - 1) / VOICE_TILE;
  size_t o = 0;
  w.off_ctrl = o;    o = align_up(o + sizeof(float) * (size_t)B * IAS_NCTRL * Tc, 256);
  w.off_vconst = o;  o = align_up(o + sizeof(IasVoiceConst) * (size_t)B, 256);
  w.off_env = o;     o = align_up(o + sizeof(float) * (size_t)B * 8 * Tc, 256);
  // the STICKY status word: one 128-byte line directly in front of the zeroed block, set together with the per-launch
  // status word and never cleared by a render (ias_voice_read_status_sticky clears it on request): a training loop that
  // only looks every N steps still learns that SOME render since its last look lost a tile.  The caller zeroes the
  // workspace once after allocating it.
  w.off_sticky = o;  o += 128;
  // words zeroed before every launch, in one block of their own (multiple of 16 bytes):
  // [ticket counters: one 128-byte line each][status word line][agg: B*ntiles*2 u64][row peaks: B u32]
  w.off_sync = o;
  w.off_agg = o + VOICE_SYNC_HEAD_BYTES;
  w.off_peak = w.off_agg + sizeof(unsigned long long) * (size_t)B * 2 * w.ntiles;
  w.sync_bytes = align_up(w.off_peak + sizeof(unsigned) * (size_t)B - w.off_sync, 16);
  o = align_up(w.off_sync + w.sync_bytes, 256);
  w.total = o;
  return w;
}

extern "C" long long ias_voice_workspace_bytes(int B, int T, int Tc) {
  if (B <= 0 || T <= 0 || Tc <= 1) return IAS_ERR_ARG;
  return (long long)voice_ws_layout(B, T, Tc).total;
}

// control points one tile can touch (+ the pair look-ahead and rounding slack)
static int voice_maxctrl(int T, int Tc) {
  return (int)((double)VOICE_TILE * (double)(Tc - 1) / (double)(T - 1)) + 6;
}

static int voice_check_dims(int B, int T, int Tc) {
  if (B <= 0 || T <= 1 || Tc <= 1 || B > 65535) return IAS_ERR_ARG;
  // control points touched by one tile must fit the LDS stage
  const double span = (double)VOICE_TILE * (double)(Tc - 1) / (double)(T - 1);
  if (span + 6.0 > (double)VOICE_MAXCTRL) return IAS_ERR_UNSUPPORTED;   // keeps the LDS image under ~35 KB
  return IAS_OK;
}

// csrc/voice_ctrl_kernels.hip: the control pass as kernels that fit beside the render's waves (<= 56 VGPRs, no library math)
bool ias_voice_control_slim_ok(int Tc, int control_rate);
int ias_voice_control_slim_launch(const float* params01, float* ctrl, void* vconst, float* sig, float* dbg, int B, int Tc,
                                  int control_rate, hipStream_t stream);

static int voice_control_launch(const float* params01, float* ctrl, void* vconst, float* sig, float* dbg, int B,
                                int Tc, int control_rate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!params01 || !ctrl || !vconst || !sig || B <= 0 || B > 65535 || Tc <= 1 || control_rate <= 0) return IAS_ERR_ARG;
  // (diagnostic library: IAS_VOICE_CTRL=fused / libm pick the one-workgroup-per-voice kernel / the round-1 kernels with the
  // device math library's pow / cos / fmodf at any size)
  const char* form = ias_diag_env("IAS_VOICE_CTRL");
  if (form == nullptr && ias_voice_control_slim_ok(Tc, control_rate))
    return ias_voice_control_slim_launch(params01, ctrl, vconst, sig, dbg, B, Tc, control_rate, stream);
#ifdef IAS_DIAG
  // (one workgroup per voice while its rows fit the LDS of a CU: 48 bytes per control sample + the 4 KB table, Tc <= ~3200)
  const size_t flds = voice_control_fused_lds(Tc);
  if (flds <= 156 * 1024 && form != nullptr && form[0] == 'f') {
    static bool attr_set = false;                          // (idempotent: a race sets it twice)
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)voice_control_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
        return IAS_ERR_LAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL(voice_control_fused_kernel, dim3(B), dim3(CTLF_THREADS), flds, stream, params01, sig, ctrl,
                       (IasVoiceConst*)vconst, dbg, Tc, (float)control_rate);
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }
#endif
  // control buffers longer than the slim kernels take: the round-1 kernels (device math library, rows through HBM)
  const size_t lds = sizeof(double) * (VOICE_WAVES + (size_t)Tc);
  if (lds > 160 * 1024) return IAS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(voice_env_kernel, dim3(6, B), dim3(VOICE_THREADS), 0, stream, params01, sig, Tc,
                     (float)control_rate);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)voice_lfo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(voice_lfo_kernel, dim3(2, B), dim3(VOICE_THREADS), lds, stream, params01, sig, dbg, Tc,
                     (float)control_rate);
  hipLaunchKernelGGL(voice_modmix_kernel, dim3((Tc + VOICE_THREADS - 1) / VOICE_THREADS, B), dim3(VOICE_THREADS), 0,
                     stream, params01, sig, ctrl, (IasVoiceConst*)vconst, dbg, Tc);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// env: scratch [B][8][Tc] floats (six envelopes + two LFO outputs), also an output for diagnostics.
extern "C" int ias_voice_control(const float* params01, float* ctrl, void* vconst, float* env, int B, int Tc,
                                 int control_rate, void* stream_) {
  return voice_control_launch(params01, ctrl, vconst, env, nullptr, B, Tc, control_rate, stream_);
}

// The same pass into the ctrl / vconst / env regions of an ias_voice_render workspace (the layout stays private to
// this file: callers that split the render into control + stages never compute offsets themselves).
extern "C" int ias_voice_control_ws(const float* params01, void* workspace, long long workspace_bytes, int B, int T,
                                    int Tc, int control_rate, void* stream_) {
  if (!workspace) return IAS_ERR_ARG;
  int rc = voice_check_dims(B, T, Tc);
  if (rc) return rc;
  const VoiceWs w = voice_ws_layout(B, T, Tc);
  if ((size_t)workspace_bytes < w.total) return IAS_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  return voice_control_launch(params01, (float*)(ws + w.off_ctrl), ws + w.off_vconst, (float*)(ws + w.off_env), nullptr,
                              B, Tc, control_rate, stream_);
}

// Same, plus the control-rate intermediates dbg [B][10][Tc] (6 envelopes, 2 LFO phases, 2 LFO outputs).
extern "C" int ias_voice_control_debug(const float* params01, float* ctrl, void* vconst, float* env, float* dbg,
                                       int B, int Tc, int control_rate, void* stream_) {
  if (!dbg) return IAS_ERR_ARG;
  return voice_control_launch(params01, ctrl, vconst, env, dbg, B, Tc, control_rate, stream_);
}

// Persistent grid of the audio-rate kernel: resident workgroups per CU (occupancy query, cached) x CUs.
template <int MATH, bool FMA_DIV>
static int voice_audio_grid(size_t lds, int total_tiles) {
  static int cached_lds = -1, cached_grid = 0;
  if (cached_lds != (int)lds) {
    int dev = 0, ncu = 256, per_cu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, voice_audio_kernel<MATH, FMA_DIV>, AUDIO_THREADS, lds) != hipSuccess ||
        per_cu < 1)
      per_cu = 2;
    if (const char* e = ias_diag_env("IAS_VOICE_PERCU")) {   // diagnostics: fewer resident workgroups (room for neighbours)
      const int v = atoi(e);
      if (v >= 1 && v < per_cu) per_cu = v;
    }
    cached_grid = per_cu * ncu;
    cached_lds = (int)lds;
    if (ias_diag_env("IAS_DEBUG")) fprintf(stderr, "[ias] voice_audio_kernel: %d workgroups/CU x %d CUs, %zu B LDS\n", per_cu, ncu, lds);
  }
  return cached_grid < total_tiles ? cached_grid : total_tiles;
}

template <int MATH, bool FMA_DIV>
static void voice_audio_launch(hipStream_t stream, const VoiceWs& w, char* ws, const float* noise, float* audio, int B,
                               int T, int Tc, int sample_rate) {
  const float scale = (float)(Tc - 1) / (float)(T - 1);
  const float sr_f = (float)sample_rate;
  const int maxctrl = voice_maxctrl(T, Tc);
  const size_t lds = sizeof(double) * VOICE_TAB_DOUBLES + sizeof(float) * VOICE_TILE + 2 * sizeof(float2) * IAS_NCTRL * (size_t)maxctrl;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)voice_audio_kernel<MATH, FMA_DIV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int ntiles = w.ntiles;
  const int grid = voice_audio_grid<MATH, FMA_DIV>(lds, ntiles * B);
  hipLaunchKernelGGL((voice_audio_kernel<MATH, FMA_DIV>), dim3(grid), dim3(AUDIO_THREADS), lds, stream,
                     (const float*)(ws + w.off_ctrl), (const IasVoiceConst*)(ws + w.off_vconst), noise, audio,
                     (unsigned long long*)(ws + w.off_agg), (unsigned int*)(ws + w.off_sync),
                     (unsigned*)(ws + w.off_peak), T, Tc, ntiles, B, 1.0 / (double)sample_rate, sr_f,
                     1.0f / sr_f, scale, maxctrl);
}

// One stage of the render on an already-filled workspace (ias_voice_control_ws must have run into it):
//   stage 0: single-pass audio-rate kernel: phase increments, chained fp64 scan across tiles,
//            oscillators + mixer -> unnormalised audio, row peaks (voice_audio_kernel)
//   stage 1: normalize_if_clipping in place (voice_normalize_kernel)
//   stage 2: only the re-zeroing of the polled words (ticket, timeout flag, tile aggregates, row peaks) that stage 0
//            starts with -- for pipelines that issue it ahead, on another stream, once the previous readers of the
//            workspace (the render AND the consumers of its row peaks) are done
//   stage 3: stage 0 without that re-zeroing (a stage 2 on this workspace must have run since its last stage 0 / 3)
// math_mode: 0 = the tested contract (oracle math "cr"); 1 = hardware fp32 exp2 on the pitch path (A/B only).
extern "C" int ias_voice_stage(int stage, int math_mode, const float* noise, float* audio, void* workspace,
                               long long workspace_bytes, int B, int T, int Tc, int sample_rate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!workspace || sample_rate <= 0 || stage < 0 || stage > 3 || math_mode < 0 || math_mode > 1) return IAS_ERR_ARG;
  if (stage != 2 && (!noise || !audio)) return IAS_ERR_ARG;
  int rc = voice_check_dims(B, T, Tc);
  if (rc) return rc;
  const VoiceWs w = voice_ws_layout(B, T, Tc);
  if ((size_t)workspace_bytes < w.total) return IAS_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  if (stage == 2) {
    if (hipMemsetAsync(ws + w.off_sync, 0, w.sync_bytes, stream) != hipSuccess) return IAS_ERR_LAUNCH;
  } else if (stage == 0 || stage == 3) {
    // ticket, timeout flag, tile aggregates and row peaks are re-zeroed on every call
    if (stage == 0 && hipMemsetAsync(ws + w.off_sync, 0, w.sync_bytes, stream) != hipSuccess) return IAS_ERR_LAUNCH;
    const bool fma_div = ias_div_fma_rate_ok(sample_rate) != 0;   // else: fp64 reciprocal path of the increment
    if (math_mode == VOICE_MATH_CR) {
      if (fma_div) voice_audio_launch<VOICE_MATH_CR, true>(stream, w, ws, noise, audio, B, T, Tc, sample_rate);
      else voice_audio_launch<VOICE_MATH_CR, false>(stream, w, ws, noise, audio, B, T, Tc, sample_rate);
    } else {
      if (fma_div) voice_audio_launch<VOICE_MATH_HW, true>(stream, w, ws, noise, audio, B, T, Tc, sample_rate);
      else voice_audio_launch<VOICE_MATH_HW, false>(stream, w, ws, noise, audio, B, T, Tc, sample_rate);
    }
  } else {
    const int nvec = T / 4;
    int gx = (nvec + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(voice_normalize_kernel, dim3(gx, B), dim3(256), 0, stream, audio, (unsigned*)(ws + w.off_peak), T,
                       nvec);
  }
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

extern "C" int ias_voice_render(const float* params01, const float* noise, float* audio, void* workspace,
                                long long workspace_bytes, int B, int T, int Tc, int sample_rate,
                                int control_rate, int normalize, int math_mode, void* stream_) {
  if (!params01 || !noise || !audio || !workspace || sample_rate <= 0 || control_rate <= 0) return IAS_ERR_ARG;
  int rc = voice_check_dims(B, T, Tc);
  if (rc) return rc;
  const VoiceWs w = voice_ws_layout(B, T, Tc);
  if ((size_t)workspace_bytes < w.total) return IAS_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  rc = ias_voice_control(params01, (float*)(ws + w.off_ctrl), ws + w.off_vconst, (float*)(ws + w.off_env), B, Tc,
                         control_rate, stream_);
  for (int stage = 0; stage < (normalize ? 2 : 1) && rc == IAS_OK; ++stage)
    rc = ias_voice_stage(stage, math_mode, noise, audio, workspace, workspace_bytes, B, T, Tc, sample_rate, stream_);
  return rc;
}

// 0 if the last render's tile chain completed, 1 if a workgroup gave up waiting (output invalid).
extern "C" int ias_voice_read_status(const void* workspace, unsigned* status_dev, int B, int T, int Tc, void* stream_) {
  if (!workspace || !status_dev) return IAS_ERR_ARG;
  const VoiceWs w = voice_ws_layout(B, T, Tc);
  if (hipMemcpyAsync(status_dev, (const char*)workspace + w.off_sync + VOICE_NCOUNTERS * 128, sizeof(unsigned), hipMemcpyDeviceToDevice,
                     (hipStream_t)stream_) != hipSuccess)
    return IAS_ERR_LAUNCH;
  return IAS_OK;
}

// The sticky status word -> status_dev: non-zero if ANY render into this workspace since the word was last cleared lost a
// tile (see VoiceWs::off_sticky); clear != 0 re-zeroes it behind the read, in stream order.  The workspace must have been
// zeroed once after allocation.
extern "C" int ias_voice_read_status_sticky(void* workspace, unsigned* status_dev, int B, int T, int Tc, int clear,
                                            void* stream_) {
  if (!workspace || !status_dev) return IAS_ERR_ARG;
  const VoiceWs w = voice_ws_layout(B, T, Tc);
  char* word = (char*)workspace + w.off_sticky;
  if (hipMemcpyAsync(status_dev, word, sizeof(unsigned), hipMemcpyDeviceToDevice, (hipStream_t)stream_) != hipSuccess)
    return IAS_ERR_LAUNCH;
  if (clear && hipMemsetAsync(word, 0, sizeof(unsigned), (hipStream_t)stream_) != hipSuccess) return IAS_ERR_LAUNCH;
  return IAS_OK;
}

// Byte offset of the row peaks [B] (fp32 bit patterns of max |x| before normalisation) inside the workspace: consumers
// that fold the normalisation in (ias_pqmf_analysis / ias_stft `rowpeak`) read them in place.
// Byte offsets of the control signals [B,5,Tc] fp32 and of the per-voice constants [B] (64 B each) the last render (or
// ias_voice_control_ws) left in the workspace: what ias_voice_backward takes as ctrl / vconst, without a second control pass.
extern "C" long long ias_voice_ctrl_offset(int B, int T, int Tc) {
  if (B <= 0 || T <= 0 || Tc <= 1) return IAS_ERR_ARG;
  return (long long)voice_ws_layout(B, T, Tc).off_ctrl;
}
extern "C" long long ias_voice_vconst_offset(int B, int T, int Tc) {
  if (B <= 0 || T <= 0 || Tc <= 1) return IAS_ERR_ARG;
  return (long long)voice_ws_layout(B, T, Tc).off_vconst;
}

extern "C" long long ias_voice_peaks_offset(int B, int T, int Tc) {
  if (B <= 0 || T <= 0 || Tc <= 1) return IAS_ERR_ARG;
  return (long long)voice_ws_layout(B, T, Tc).off_peak;
}

// What a render's backward needs of its workspace, copied out in ONE launch (three device-to-device copies were three
// memcpy nodes of a captured step: 30 us between the render and its first consumer): ctrl [B,5,Tc], vconst [B,16] floats,
// peaks [B] (NULL: not wanted).
__global__ __launch_bounds__(256) void voice_save_kernel(const float* __restrict__ ctrl, const float* __restrict__ vconst,
                                                         const float* __restrict__ peak, float* __restrict__ ctrl_out,
                                                         float* __restrict__ vconst_out, float* __restrict__ peaks_out,
                                                         long long nctrl, int nvc, int npk) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < nctrl) ctrl_out[i] = ctrl[i];
  if (i < nvc) vconst_out[i] = vconst[i];
  if (peaks_out && i < npk) peaks_out[i] = peak[i];
}
extern "C" int ias_voice_save_for_backward(const void* workspace, float* ctrl_out, float* vconst_out, float* peaks_out, int B,
                                           int T, int Tc, void* stream_) {
  if (!workspace || !ctrl_out || !vconst_out || B <= 0 || T <= 0 || Tc <= 1) return IAS_ERR_ARG;
  const VoiceWs w = voice_ws_layout(B, T, Tc);
  const char* ws = (const char*)workspace;
  const long long nctrl = (long long)B * IAS_NCTRL * Tc;
  static_assert(sizeof(IasVoiceConst) == 64, "vconst rows are 16 floats");
  const long long nthreads = nctrl > 16LL * B ? nctrl : 16LL * B;      // (Tc of 2 or 3: the vconst copy is the longest)
  hipLaunchKernelGGL(voice_save_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                     (const float*)(ws + w.off_ctrl), (const float*)(ws + w.off_vconst), (const float*)(ws + w.off_peak),
                     ctrl_out, vconst_out, peaks_out, nctrl, B * 16, B);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// Row peaks (|x| max before normalisation) of the last render, for tests/diagnostics.
extern "C" int ias_voice_read_peaks(const void* workspace, float* peaks_dev, int B, int T, int Tc, void* stream_) {
  if (!workspace || !peaks_dev) return IAS_ERR_ARG;
  const VoiceWs w = voice_ws_layout(B, T, Tc);
  if (hipMemcpyAsync(peaks_dev, (const char*)workspace + w.off_peak, sizeof(float) * (size_t)B,
                     hipMemcpyDeviceToDevice, (hipStream_t)stream_) != hipSuccess)
    return IAS_ERR_LAUNCH;
  return IAS_OK;
}
