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
#include "voice_math.h"
#include "wave_ops.h"
#include "voice_table.h"

#define VOICE_THREADS 256                     // control-rate kernels
#define VOICE_WAVES (VOICE_THREADS / 64)
#define AUDIO_THREADS 256                     // audio-rate kernel (512-thread tiles measured no faster)
#define AUDIO_WAVES (AUDIO_THREADS / 64)
#define VOICE_SPT 16                          // samples per thread (8 and 12 measured slower: per-tile latencies)
#define VOICE_TILE (AUDIO_THREADS * VOICE_SPT)  // 4096 samples per workgroup
#ifndef VOICE_RUN
#define VOICE_RUN 4                            // consecutive samples of a lane within one chunk (one wave scan per chunk);
                                              // 8 halves the scans but costs a wave of occupancy (110 VGPRs): same time
#endif
#define VOICE_CHUNKS (VOICE_SPT / VOICE_RUN)

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

// -------------------------------------------------------------------- audio rate
#define VOICE_MAXCTRL 320  // control points staged per tile (covers sample rates down to ~6 kHz)
#define VOICE_SPIN_LIMIT (1u << 24)
#define VOICE_READY_BIT 0x8000000000000000ull

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

// Single pass over the row with a chained scan across tiles ("decoupled look-back"):
//   ticket  -> (tile, voice), tile-major; tickets are handed out in launch order, so every predecessor tile of
//              the same voice has already started when a workgroup begins (no dependence on dispatch
//              order or placement: a workgroup only ever waits for workgroups with smaller tickets,
//              and those publish before they wait).
//   phase A -> the tile's 2 x 4096 phase increments (kept in registers) and their fp64 sums.  Sums of
//              fp32 increments below 2^19 are exact in fp64, so the order of summation is irrelevant
//              and the result is bit-identical to the sequential double accumulation of the oracle.
//   publish -> one 8-byte write-through store per VCO: the sum's bits with the sign bit as READY flag
//              (sums are >= 0).  The datum is its own flag (MI355X guide, Guideline 16 form R2).
//   wait    -> wave 0 polls the predecessors' words with relaxed agent-scope loads (L1 bypass),
//              bounded spins, and adds them up: the tile's carry-in.
//   phase B -> fp64 scan with the carry, round to fp32, + phi, oscillators, VCAs, mixer, row peak.
__global__ __launch_bounds__(AUDIO_THREADS) void voice_audio_kernel(
    const float* __restrict__ ctrl, const IasVoiceConst* __restrict__ vconst,
    const float* __restrict__ noise, float* __restrict__ audio, unsigned long long* agg /* [B][ntiles][2] */,
    unsigned int* ticket_status /* [0] ticket counter, [1] spin-timeout flag */,
    unsigned* __restrict__ rowpeak, int T, int Tc, int ntiles, double inv_sample_rate, float sr_f, float sr_r,
    float scale, int maxctrl) {
  // control points as (c[i], c[i+1]) pairs: one ds_read_b64 fetches both ends of a lerp, and the clamp of
  // the upper index at the end of the buffer is folded into the table
  // dynamic LDS: s_inc [2][VOICE_SPT][AUDIO_THREADS] floats (the tile's phase increments wait here between
  // phase A and phase B instead of in 32 VGPRs) | s_ctrl [IAS_NCTRL][maxctrl] float2
  extern __shared__ __attribute__((aligned(16))) float dyn_smem[];
  float* s_inc = dyn_smem;
  float2* s_ctrl = reinterpret_cast<float2*>(dyn_smem + 2 * VOICE_SPT * AUDIO_THREADS);
  __shared__ double s_wsum[2][AUDIO_WAVES];
  __shared__ double s_carry[2];
  __shared__ float s_max[AUDIO_WAVES];
  __shared__ unsigned s_ticket;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_ticket = __hip_atomic_fetch_add((gu32*)ticket_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int ticket = (int)s_ticket;
  // tile-major order: the predecessors of (tile, b) are the tickets (tile', b), tile' < tile, i.e. at least
  // B tickets older -- they have normally published long before this workgroup polls.  (Row-major order,
  // where the predecessor is the previous ticket, spent 42 % of the wave time in the look-back wait.)
  const int nvoices = gridDim.x / ntiles;
  const int tile = ticket / nvoices, b = ticket - tile * nvoices;
  const int j_tile = tile * VOICE_TILE;
  const int j_last = min(j_tile + VOICE_TILE, T) - 1;

  // stage the control points this tile interpolates between
  int c_lo, c_hi;
  {
    int i0, i1; float w0, w1;
    ias_interp_pos(j_tile, scale, Tc, i0, i1, w0, w1);
    c_lo = i0;
    ias_interp_pos(j_last, scale, Tc, i0, i1, w0, w1);
    c_hi = i1;
  }
  const int ncp = c_hi - c_lo + 1;  // host guarantees ncp <= VOICE_MAXCTRL
  const float* cb = ctrl + (size_t)b * IAS_NCTRL * Tc;
  for (int i = tid; i < IAS_NCTRL * ncp; i += AUDIO_THREADS) {
    const int k = i / ncp, c = i - k * ncp;
    const float* row = cb + k * Tc;
    s_ctrl[k * maxctrl + c] = make_float2(row[c_lo + c], row[min(c_lo + c + 1, Tc - 1)]);
  }
  const IasVoiceConst vc = vconst[b];
  __syncthreads();
  // early look-back: the predecessors' words are requested here, after the staging barrier (tile-major
  // tickets make the predecessors ~3 us older: by now they have normally published) and examined only
  // after phase A, which hides the round trip
  gu64* row = (gu64*)(agg + ((size_t)b * ntiles) * 2);
  unsigned long long early1 = VOICE_READY_BIT, early2 = VOICE_READY_BIT;
  if (wave == 0 && lane < tile) {
    early1 = __hip_atomic_load(row + lane * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    early2 = __hip_atomic_load(row + lane * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }


  const int j_wave = j_tile + wave * (64 * VOICE_SPT);
  const float* nrow = noise + (size_t)b * T;
  const bool vec_ok = (T & 3) == 0;

  // phase A: increments (kept in registers) and per-wave totals
  double tot1 = 0.0, tot2 = 0.0;
  // (branch-free: samples past the end of the row are computed for the clamped index and masked,
  // so the four samples of a lane form straight-line code the compiler can pack two by two)
#pragma unroll
  for (int c = 0; c < VOICE_CHUNKS; ++c) {
#pragma unroll
    for (int h = 0; h < VOICE_RUN / 2; ++h) {   // two samples at a time (packed fp32)
      const int j = j_wave + c * (64 * VOICE_RUN) + lane * VOICE_RUN + 2 * h;
      int k[2]; ias_f2 w0, w1;
      ias_interp_pair(min(j, T - 1), min(j + 1, T - 1), scale, k, w0, w1);
      const int i0 = k[0] - c_lo, i1 = k[1] - c_lo;
      const ias_f2 pm1 = ias_lerp_pair(s_ctrl[i0], s_ctrl[i1], w0, w1);
      const ias_f2 pm2 = ias_lerp_pair(s_ctrl[2 * maxctrl + i0], s_ctrl[2 * maxctrl + i1], w0, w1);
      ias_f2 a = ias_vco_inc_pair(vc.f0_1, vc.depth_1, pm1, inv_sample_rate, sr_f, sr_r);
      ias_f2 d = ias_vco_inc_pair(vc.f0_2, vc.depth_2, pm2, inv_sample_rate, sr_f, sr_r);
      if (j >= T) { a.x = 0.0f; d.x = 0.0f; }
      if (j + 1 >= T) { a.y = 0.0f; d.y = 0.0f; }
      const int e0 = c * VOICE_RUN + 2 * h;
      s_inc[e0 * AUDIO_THREADS + tid] = a.x;
      s_inc[(e0 + 1) * AUDIO_THREADS + tid] = a.y;
      s_inc[(VOICE_SPT + e0) * AUDIO_THREADS + tid] = d.x;
      s_inc[(VOICE_SPT + e0 + 1) * AUDIO_THREADS + tid] = d.y;
      tot1 += (double)a.x; tot1 += (double)a.y;
      tot2 += (double)d.x; tot2 += (double)d.y;
    }
  }
  tot1 = wave_sum(tot1); tot2 = wave_sum(tot2);
  if (lane == 0) { s_wsum[0][wave] = tot1; s_wsum[1][wave] = tot2; }
  __syncthreads();

  // publish this tile's sums, then collect the predecessors' (wave 0)
  if (wave == 0) {
    if (lane < 2) {
      double a = 0.0;
      for (int w = 0; w < AUDIO_WAVES; ++w) a += s_wsum[lane][w];
      __hip_atomic_store(row + tile * 2 + lane, (unsigned long long)__double_as_longlong(a) | VOICE_READY_BIT,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double a1 = 0.0, a2 = 0.0;
    bool timeout = false;
    for (int t0 = 0; t0 < tile; t0 += 64) {
      const int t = t0 + lane;
      unsigned long long x1 = VOICE_READY_BIT, x2 = VOICE_READY_BIT;   // lanes without a predecessor: ready
      if (t0 == 0) { x1 = early1; x2 = early2; }
      else if (t < tile) { x1 = 0; x2 = 0; }
      unsigned spins = 0;
      bool ok = (x1 & x2 & VOICE_READY_BIT) != 0;
      while (!__all(ok)) {                        // wave-uniform loop condition
        if (++spins > VOICE_SPIN_LIMIT) { timeout = true; break; }
        if (!ok) {
          x1 = __hip_atomic_load(row + t * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          x2 = __hip_atomic_load(row + t * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = (x1 & x2 & VOICE_READY_BIT) != 0;
        }
        if (!__all(ok)) __builtin_amdgcn_s_sleep(2);
      }
      if (t < tile) {
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
        __hip_atomic_store((gu32*)ticket_status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      s_carry[0] = a1; s_carry[1] = a2;
    }
  }
  __syncthreads();

  // phase B: scan + oscillators
  double run1 = s_carry[0], run2 = s_carry[1];
  for (int w = 0; w < wave; ++w) { run1 += s_wsum[0][w]; run2 += s_wsum[1][w]; }
  float pk = 0.0f;
  float* arow = audio + (size_t)b * T;
#pragma unroll
  for (int c = 0; c < VOICE_CHUNKS; ++c) {
    constexpr int RN = VOICE_RUN;
    const int j0 = j_wave + c * (64 * RN) + lane * RN;
    double l1[RN], l2[RN];
    l1[0] = (double)s_inc[(c * RN) * AUDIO_THREADS + tid];
    l2[0] = (double)s_inc[(VOICE_SPT + c * RN) * AUDIO_THREADS + tid];
#pragma unroll
    for (int e = 1; e < RN; ++e) {
      l1[e] = l1[e - 1] + (double)s_inc[(c * RN + e) * AUDIO_THREADS + tid];
      l2[e] = l2[e - 1] + (double)s_inc[(VOICE_SPT + c * RN + e) * AUDIO_THREADS + tid];
    }
    const double in1 = wave_incl_scan(l1[RN - 1], lane), in2 = wave_incl_scan(l2[RN - 1], lane);
    const double base1 = run1 + (in1 - l1[RN - 1]), base2 = run2 + (in2 - l2[RN - 1]);
    run1 += __shfl(in1, 63, 64); run2 += __shfl(in2, 63, 64);

    float nz[RN];
#pragma unroll
    for (int e = 0; e < RN; ++e) nz[e] = 0.f;
    if (vec_ok && j0 + RN - 1 < T) {
#pragma unroll
      for (int q = 0; q < RN / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(nrow + j0 + 4 * q);
        nz[4 * q] = v.x; nz[4 * q + 1] = v.y; nz[4 * q + 2] = v.z; nz[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int e = 0; e < RN; ++e) if (j0 + e < T) nz[e] = nrow[j0 + e];
    }
    float o[RN];
#pragma unroll
    for (int h = 0; h < RN / 2; ++h) {   // two samples at a time (packed fp32)
      const int j = j0 + 2 * h;
      int k[2]; ias_f2 w0, w1;
      ias_interp_pair(min(j, T - 1), min(j + 1, T - 1), scale, k, w0, w1);
      const int i0 = k[0] - c_lo, i1 = k[1] - c_lo;
      const ias_f2 amp1 = ias_lerp_pair(s_ctrl[maxctrl + i0], s_ctrl[maxctrl + i1], w0, w1);
      const ias_f2 amp2 = ias_lerp_pair(s_ctrl[3 * maxctrl + i0], s_ctrl[3 * maxctrl + i1], w0, w1);
      const ias_f2 ampn = ias_lerp_pair(s_ctrl[4 * maxctrl + i0], s_ctrl[4 * maxctrl + i1], w0, w1);
      const ias_f2 a1 = (ias_f2){(float)(base1 + l1[2 * h]), (float)(base1 + l1[2 * h + 1])} + vc.phi_1;
      const ias_f2 a2 = (ias_f2){(float)(base2 + l2[2 * h]), (float)(base2 + l2[2 * h + 1])} + vc.phi_2;
      const ias_f2 om = ias_mix_pair_dev(a1, a2, amp1, amp2, ampn, (ias_f2){nz[2 * h], nz[2 * h + 1]}, vc);
      o[2 * h] = om.x; o[2 * h + 1] = om.y;
      if (j < T) pk = fmaxf(pk, fabsf(om.x));
      if (j + 1 < T) pk = fmaxf(pk, fabsf(om.y));
    }
    if (vec_ok && j0 + RN - 1 < T) {
#pragma unroll
      for (int q = 0; q < RN / 4; ++q)
        *reinterpret_cast<float4*>(arow + j0 + 4 * q) = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    } else {
#pragma unroll
      for (int e = 0; e < RN; ++e) if (j0 + e < T) arow[j0 + e] = o[e];
    }
  }
  pk = wave_max(pk);
  if (lane == 0) s_max[wave] = pk;
  __syncthreads();
  if (tid == 0) {
    float m = s_max[0];
    for (int w = 1; w < AUDIO_WAVES; ++w) m = fmaxf(m, s_max[w]);
    atomicMax(rowpeak + b, __float_as_uint(m));  // m >= 0: uint order == float order
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
  size_t off_ctrl, off_vconst, off_env, off_sync, sync_bytes, off_agg, off_peak, total;
  int ntiles;
};
static VoiceWs voice_ws_layout(int B, int T, int Tc) {
  VoiceWs w;
  w.ntiles = (T + VOICE_TILE - 1) / VOICE_TILE;
  size_t o = 0;
  w.off_ctrl = o;    o = align_up(o + sizeof(float) * (size_t)B * IAS_NCTRL * Tc, 256);
  w.off_vconst = o;  o = align_up(o + sizeof(IasVoiceConst) * (size_t)B, 256);
  w.off_env = o;     o = align_up(o + sizeof(float) * (size_t)B * 8 * Tc, 256);
  // words zeroed before every launch, in one block of their own (multiple of 16 bytes):
  // [ticket, timeout flag, pad, pad][agg: B*ntiles*2 u64][row peaks: B u32]
  w.off_sync = o;
  w.off_agg = o + 16;
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
  if (span + 6.0 > (double)VOICE_MAXCTRL) return IAS_ERR_UNSUPPORTED;   // keeps the LDS image under ~45 KB
  return IAS_OK;
}

static int voice_control_launch(const float* params01, float* ctrl, void* vconst, float* sig, float* dbg, int B,
                                int Tc, int control_rate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!params01 || !ctrl || !vconst || !sig || B <= 0 || B > 65535 || Tc <= 1 || control_rate <= 0) return IAS_ERR_ARG;
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

// One stage of the render on an already-filled workspace (ias_voice_control must have run into it):
//   stage 0: single-pass audio-rate kernel: phase increments, chained fp64 scan across tiles,
//            oscillators + mixer -> unnormalised audio, row peaks (voice_audio_kernel)
//   stage 1: normalize_if_clipping in place (voice_normalize_kernel)
extern "C" int ias_voice_stage(int stage, const float* noise, float* audio, void* workspace,
                               long long workspace_bytes, int B, int T, int Tc, int sample_rate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!noise || !audio || !workspace || sample_rate <= 0 || stage < 0 || stage > 1) return IAS_ERR_ARG;
  int rc = voice_check_dims(B, T, Tc);
  if (rc) return rc;
  const VoiceWs w = voice_ws_layout(B, T, Tc);
  if ((size_t)workspace_bytes < w.total) return IAS_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  const float* ctrl = (const float*)(ws + w.off_ctrl);
  const IasVoiceConst* vconst = (const IasVoiceConst*)(ws + w.off_vconst);
  unsigned* peak = (unsigned*)(ws + w.off_peak);
  if (stage == 0) {
    // ticket, timeout flag, tile aggregates and row peaks are re-zeroed on every call
    if (hipMemsetAsync(ws + w.off_sync, 0, w.sync_bytes, stream) != hipSuccess) return IAS_ERR_LAUNCH;
    const float scale = (float)(Tc - 1) / (float)(T - 1);
    const float sr_f = ias_div_fma_rate_ok(sample_rate) ? (float)sample_rate : 0.0f;   // 0: fp64 reciprocal path
    const int maxctrl = voice_maxctrl(T, Tc);
    const size_t lds = sizeof(float) * 2 * VOICE_SPT * AUDIO_THREADS + sizeof(float2) * IAS_NCTRL * (size_t)maxctrl;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)voice_audio_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(voice_audio_kernel, dim3(w.ntiles * B), dim3(AUDIO_THREADS), lds, stream, ctrl, vconst, noise,
                       audio, (unsigned long long*)(ws + w.off_agg), (unsigned int*)(ws + w.off_sync), peak, T, Tc,
                       w.ntiles, 1.0 / (double)sample_rate, sr_f, sr_f > 0.0f ? 1.0f / sr_f : 0.0f, scale, maxctrl);
  } else {
    const int nvec = T / 4;
    int gx = (nvec + 255) / 256;
    if (gx > 64) gx = 64;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(voice_normalize_kernel, dim3(gx, B), dim3(256), 0, stream, audio, peak, T, nvec);
  }
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

extern "C" int ias_voice_render(const float* params01, const float* noise, float* audio, void* workspace,
                                long long workspace_bytes, int B, int T, int Tc, int sample_rate,
                                int control_rate, int normalize, void* stream_) {
  if (!params01 || !noise || !audio || !workspace || sample_rate <= 0 || control_rate <= 0) return IAS_ERR_ARG;
  int rc = voice_check_dims(B, T, Tc);
  if (rc) return rc;
  const VoiceWs w = voice_ws_layout(B, T, Tc);
  if ((size_t)workspace_bytes < w.total) return IAS_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  rc = ias_voice_control(params01, (float*)(ws + w.off_ctrl), ws + w.off_vconst, (float*)(ws + w.off_env), B, Tc,
                         control_rate, stream_);
  for (int stage = 0; stage < (normalize ? 2 : 1) && rc == IAS_OK; ++stage)
    rc = ias_voice_stage(stage, noise, audio, workspace, workspace_bytes, B, T, Tc, sample_rate, stream_);
  return rc;
}

// 0 if the last render's tile chain completed, 1 if a workgroup gave up waiting (output invalid).
extern "C" int ias_voice_read_status(const void* workspace, unsigned* status_dev, int B, int T, int Tc, void* stream_) {
  if (!workspace || !status_dev) return IAS_ERR_ARG;
  const VoiceWs w = voice_ws_layout(B, T, Tc);
  if (hipMemcpyAsync(status_dev, (const char*)workspace + w.off_sync + 4, sizeof(unsigned), hipMemcpyDeviceToDevice,
                     (hipStream_t)stream_) != hipSuccess)
    return IAS_ERR_LAUNCH;
  return IAS_OK;
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
