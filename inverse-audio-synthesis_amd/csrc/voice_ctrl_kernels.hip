// The control-rate pass of the Voice render as kernels that fit BESIDE the audio-rate render (MI355X / gfx950, round 5).
//
// Replaces, at every realistic size, the control half of torchsynth's Voice.output() as the reference drives it
// (/root/reference/vicreg_audio_params.py:86-94,114; audio_to_params.py:215,240-257): 78 normalised parameters -> six ADSR
// envelopes, two LFOs, the 4 x 5 modulation matrix -> ctrl [B][5][Tc] + per-voice constants.  Arithmetic contract:
// csrc/voice_math.h (== oracle/synth_oracle.py, math "cr"); every control-rate value is bit-equal to the oracle's.
//
// Why these kernels exist.  Measured (scripts/diag/run_noctrl_ab.sh, same box): the headline step costs 0.175 ms with the
// control pass and 0.159 ms without it -- 33 us of small kernels cost 15 us of every step, although they are issued a whole
// step ahead on a stream of their own.  The reason is residency, not latency: the persistent render keeps three workgroups
// on every CU (3 x 152 VGPRs of a SIMD's 512, 125 KB of the 160 KB of LDS), the old control kernels needed 76-106 VGPRs,
// so a control workgroup could only start on a CU that the render had LEFT, and then held it.  What a SIMD has free beside
// three render waves is 56 VGPRs and wave slots; the render itself issues on ~75 % of the SIMD's cycles.  Kernels that
// need <= 56 VGPRs, a few KB of LDS and no scratch run in those gaps.  That rules out the device math library (fp64 pow is
// ~300 instructions and > 90 VGPRs): this translation unit is compiled with IAS_CTL_NO_LIBM, i.e. every transcendental is
// csrc/voice_ctrl_math.h's written-out fp64 form (same fp32 values as libm: tests/test_voice_math_cpu.py).
//
//   voice_env_slim_kernel     one workgroup per (envelope, voice)
//   voice_lfo_slim_kernel     one workgroup per (LFO, voice): fp64 phase scan in LDS, five shapes, amplitude envelope
//   voice_modmix_slim_kernel  4 x 5 mod matrix -> ctrl, and the per-voice constants
// The launcher (voice_kernels.hip: voice_control_launch) takes them while Tc <= 4096 (the scan's LDS) and the buffer is
// shorter than 600 s (LFO phases far below the 2^20 rad up to which cos / fmod reduce exactly); longer buffers take the
// round-1 kernels with the library's functions.
#define IAS_CTL_NO_LIBM
#include "voice_math.h"
#include "wave_ops.h"
#include "voice_table.h"

#define CS_THREADS 256
#define CS_WAVES (CS_THREADS / 64)
// 56 registers per lane: on gfx90a+ (unified VGPR / AGPR file) the backend doubles the attribute's value, so 28 asks for a
// budget of 56 (checked with -Rpass-analysis=kernel-resource-usage: 52 / 56 / 55 VGPRs, no scratch)
#define CS_KERNEL __global__ __launch_bounds__(CS_THREADS) __attribute__((amdgpu_num_vgpr(28)))

__constant__ IasParamRange c_cs_table[IAS_NPARAMS] = IAS_PARAM_TABLE_INIT;
__device__ const double g_cs_tab[IAS_CTL_TAB_DOUBLES] = IAS_CTL_TAB_INIT;

__device__ __forceinline__ void cs_stage_table(double* s_tab) {
  for (int i = threadIdx.x; i < IAS_CTL_TAB_DOUBLES; i += CS_THREADS) s_tab[i] = g_cs_tab[i];
}
__device__ __forceinline__ float cs_mapped(const float* __restrict__ params01, int b, int idx, const double* s_tab) {
  const IasParamRange r = c_cs_table[idx];
  return ias_map_param(params01[(size_t)b * IAS_NPARAMS + idx], (float)r.lo, (float)r.span, (float)r.curve, r.symmetric, s_tab);
}
__device__ __forceinline__ int cs_adsr_base(int a) {
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

// sig rows 0-5: env[b][a][t]
CS_KERNEL void voice_env_slim_kernel(const float* __restrict__ params01, float* __restrict__ sig, int Tc, float control_rate) {
  __shared__ __attribute__((aligned(16))) double s_tab[IAS_CTL_TAB_DOUBLES];
  __shared__ float s_p[8];
  const int a = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  cs_stage_table(s_tab);
  __syncthreads();
  if (tid < 5) s_p[tid] = cs_mapped(params01, b, cs_adsr_base(a) + tid, s_tab);
  if (tid == 5) s_p[5] = cs_mapped(params01, b, IAS_P_KEYBOARD_DURATION, s_tab);
  __syncthreads();
  IasAdsr e;
  e.attack = s_p[0]; e.decay = s_p[1]; e.sustain = s_p[2]; e.release = s_p[3]; e.alpha = s_p[4];
  const float note_on = s_p[5], eps = (float)IAS_EPS;
  // flat heads of the decay / release ramps: the same for every t before the ramp starts
  if (tid == 0) s_p[6] = ias_adsr_heads(e, note_on, control_rate, eps, s_tab).decay_head;
  if (tid == 64) s_p[7] = ias_adsr_heads(e, note_on, control_rate, eps, s_tab).release_head;
  __syncthreads();
  IasAdsrHeads heads;
  heads.decay_head = s_p[6]; heads.release_head = s_p[7];
  float* out = sig + ((size_t)b * 8 + a) * Tc;
  for (int t = tid; t < Tc; t += CS_THREADS) out[t] = ias_adsr_headed(t, e, note_on, control_rate, eps, heads, s_tab);
}

// sig rows 6-7: phase scan (fp64 accumulate, fp32 per-sample round), the five LFO shapes, amplitude envelope
CS_KERNEL void voice_lfo_slim_kernel(const float* __restrict__ params01, float* __restrict__ sig,
                                     float* __restrict__ dbg /* optional [B][10][Tc]: LFO phases (rows 6, 7) and outputs (8, 9) */,
                                     int Tc, float control_rate) {
  extern __shared__ __attribute__((aligned(16))) double cs_smem[];
  double* s_tab = cs_smem;                                   // IAS_CTL_TAB_DOUBLES
  double* s_sum = s_tab + IAS_CTL_TAB_DOUBLES;               // Tc wave-local inclusive sums
  __shared__ double s_wtot[CS_WAVES];
  __shared__ float s_q[8];
  __shared__ float s_mode[8];
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qbase = (l == 0) ? IAS_P_LFO_1_FREQUENCY : IAS_P_LFO_2_FREQUENCY;
  cs_stage_table(s_tab);
  __syncthreads();
  if (tid < 8) s_q[tid] = cs_mapped(params01, b, qbase + tid, s_tab);
  __syncthreads();
  if (tid < 5) s_mode[tid] = ias_pow_ctl(s_q[3 + tid], IAS_LFO_EXPONENT_F, s_tab);     // ias_lfo_mode's five powers ...
  const float freq = s_q[0], depth = s_q[1], phi = s_q[2];
  const float* rate_env = sig + ((size_t)b * 8 + 4 + l) * Tc;
  const float* amp_env = sig + ((size_t)b * 8 + 2 + l) * Tc;

  // each wave scans a contiguous quarter of the control buffer in chunks of 64, then the quarters are chained through LDS
  // (fp64; the order differs from a sequential loop only below 1e-16 relative, which the per-sample rounding to fp32 absorbs)
  const int per_wave = ((Tc + CS_WAVES - 1) / CS_WAVES + 63) / 64 * 64;
  const int t_begin = min(wave * per_wave, Tc), t_end = min(t_begin + per_wave, Tc);
  double carry = 0.0;
  for (int t0 = t_begin; t0 < t_end; t0 += 64) {
    const int t = t0 + lane;
    double inc = 0.0;
    if (t < t_end) inc = (double)ias_lfo_inc(freq, depth, rate_env[t], control_rate);
    const double sc = wave_incl_scan(inc, lane) + carry;
    carry = __shfl(sc, 63, 64);
    if (t < t_end) s_sum[t] = sc;
  }
  if (lane == 0) s_wtot[wave] = carry;
  __syncthreads();
  double base = 0.0;
  for (int w = 0; w < wave; ++w) base += s_wtot[w];
  float mode[5];
  {                                                          // ... and their normalisation (every thread: five LDS reads)
    const float sm = (float)((double)s_mode[0] + (double)s_mode[1] + (double)s_mode[2] + (double)s_mode[3] + (double)s_mode[4]);
#pragma unroll
    for (int k = 0; k < 5; ++k) mode[k] = ias_div(s_mode[k], sm);
  }
  float* out = sig + ((size_t)b * 8 + 6 + l) * Tc;
  for (int t = t_begin + lane; t < t_end; t += 64) {
    const double ph = base + s_sum[t];
    const float arg = ias_add((float)ph, phi);
    const float o = ias_mul(ias_lfo_shape_mix(arg, mode, s_tab), amp_env[t]);
    out[t] = o;
    if (dbg != nullptr) {
      dbg[((size_t)b * 10 + 6 + l) * Tc + t] = arg;
      dbg[((size_t)b * 10 + 8 + l) * Tc + t] = o;
    }
  }
}

// ctrl[b][j][t] = sum_k w[k][j] * sig_k[t] (4 x 5 mod matrix, fp64-accumulated dot) and IasVoiceConst[b]
CS_KERNEL void voice_modmix_slim_kernel(const float* __restrict__ params01, const float* __restrict__ sig,
                                        float* __restrict__ ctrl, IasVoiceConst* __restrict__ vconst,
                                        float* __restrict__ dbg, int Tc) {
  __shared__ __attribute__((aligned(16))) double s_tab[IAS_CTL_TAB_DOUBLES];
  __shared__ float s_w[20];
  __shared__ float s_p[IAS_NPARAMS];
  const int b = blockIdx.y, tid = threadIdx.x;
  cs_stage_table(s_tab);
  __syncthreads();
  if (tid < 20) s_w[tid] = cs_mapped(params01, b, IAS_P_MOD_MATRIX_ADSR_1_TO_VCO_1_PITCH + tid, s_tab);
  if (blockIdx.x == 0 && tid >= 64 && tid < 64 + IAS_NPARAMS) s_p[tid - 64] = cs_mapped(params01, b, tid - 64, s_tab);
  __syncthreads();
  const float* sb = sig + (size_t)b * 8 * Tc;
  const int t = blockIdx.x * CS_THREADS + tid;
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
    vc.kpart = ias_partials_k(midi_f0, vc.depth_2, s_tab);
    vc.shape = p[IAS_P_VCO_2_SHAPE];
    vc.shape_gain = ias_sub(1.0f, ias_div(vc.shape, 2.0f));
    vc.lvl0 = p[IAS_P_MIXER_VCO_1];
    vc.lvl1 = p[IAS_P_MIXER_VCO_2];
    vc.lvl2 = p[IAS_P_MIXER_NOISE];
    vc.pad[0] = vc.pad[1] = vc.pad[2] = vc.pad[3] = 0.0f;
    vconst[b] = vc;
  }
}

// Whether these kernels take a control pass of this shape (see the head of the file), and the launch.  -> IAS_* status.
bool ias_voice_control_slim_ok(int Tc, int control_rate) {
  return Tc <= 4096 && (double)Tc / (double)control_rate <= 600.0;
}
int ias_voice_control_slim_launch(const float* params01, float* ctrl, void* vconst, float* sig, float* dbg, int B, int Tc,
                                  int control_rate, hipStream_t stream) {
  const size_t lds = sizeof(double) * (IAS_CTL_TAB_DOUBLES + (size_t)Tc);
  hipLaunchKernelGGL(voice_env_slim_kernel, dim3(6, B), dim3(CS_THREADS), 0, stream, params01, sig, Tc, (float)control_rate);
  hipLaunchKernelGGL(voice_lfo_slim_kernel, dim3(2, B), dim3(CS_THREADS), lds, stream, params01, sig, dbg, Tc, (float)control_rate);
  hipLaunchKernelGGL(voice_modmix_slim_kernel, dim3((Tc + CS_THREADS - 1) / CS_THREADS, B), dim3(CS_THREADS), 0, stream, params01,
                     (const float*)sig, ctrl, (IasVoiceConst*)vconst, dbg, Tc);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
