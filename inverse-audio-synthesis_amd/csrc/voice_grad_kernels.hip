// Backward of the audio-rate Voice render for MI355X (gfx950): d loss / d (control-rate signals, per-voice
// constants) given d loss / d (un-normalised mix).
//
// The reference never differentiates through its synth -- the audio -> params -> synth -> mel-L1 loop is the
// abandoned block /root/reference/audio_to_params.py:56-172.  This file is the
// adjoint of csrc/voice_kernels.hip:voice_audio_kernel, i.e. of
//   upc   = Upsample(linear, align_corners)(ctrl)                                   [B,5,T]
//   arg_v = cumsum(2 pi * 440 * 2^((clamp(f0_v + depth_v * upc_pitch_v, 0, 127) - 69) / 12) / sr) + phi_v
//   mixed = lvl0 * cos(arg_1) * upc_amp1
//         + lvl1 * gain * tanh(kpart * sin(arg_2) / 2) * (1 + shape * cos(arg_2)) * upc_amp2
//         + lvl2 * noise * upc_ampn
// with torch.autograd's conventions (clamp passes the gradient on the closed interval).  The control-rate part
// (78 parameters -> ctrl, constants) is differentiated on the host side by torch (voice_grad.py).
//
// Four passes over the [B,T] row, all with workgroup = one tile of GRAD_TILE consecutive samples:
//   K0 increments  : the forward's phase increments (same correctly-rounded arithmetic, voice_math.h) -> planes
//                    inc_1, inc_2 (sign bit set where the pitch clamp is active) + their fp64 tile sums
//   K1 sample grads: phases = exact fp64 prefix of the increments (tile sums of the earlier tiles + in-tile
//                    scan), oscillators, then g_amp1, g_amp2, g_ampn, g_arg1, g_arg2 per sample + per-tile
//                    partial sums for the constants (lvl*, kpart, shape, gain, phi*)
//   K2 pitch grads : g_cumsum -> suffix sums of g_arg (fp64, tile totals of the later tiles + in-tile reverse
//                    scan) * d inc / d pitch -> g_pitchmod per sample, partial sums for f0 and depth
//   K3 upsample^T  : g_ctrl[b,row,i] = sum_t W[t,i] * g_upc[b,row,t] as per-interval sums (every sample read once,
//                    staged in LDS, summed in a fixed order, no atomics -> bit-reproducible) + a combine pass
// Planes are caller-owned scratch [B,8,T] fp32; partials [B,ntiles,IAS_GRAD_NS] fp64 are summed by the caller.
#include "ias_common.h"
#include "voice_math.h"
#include "wave_ops.h"
#include "voice_trig.h"
#include <cstdint>
#include <cstdlib>

#define GRAD_THREADS 256
#define GRAD_WAVES (GRAD_THREADS / 64)
#ifndef GRAD_CHUNKS
#define GRAD_CHUNKS 16      // samples per thread and tile: 16, or 8 (the lane-consecutive kernels' blocks: 64 / 32 bytes per lane;
                            // 8 gives them 16 waves per CU instead of 8 and was measured no faster: 95 + 69 vs 94 + 64 us)
#endif
#define GRAD_TILE (GRAD_THREADS * GRAD_CHUNKS)
#define IAS_GRAD_NS 12      // f0_1 depth_1 phi_1 f0_2 depth_2 phi_2 kpart shape gain lvl0 lvl1 lvl2
#define IAS_GRAD_PLANES 8   // inc_1 inc_2 | g_amp1 g_amp2 g_ampn | g_arg1->g_pm1 g_arg2->g_pm2 | interval sums of K3

enum { GS_F0_1 = 0, GS_DEPTH_1, GS_PHI_1, GS_F0_2, GS_DEPTH_2, GS_PHI_2, GS_KPART, GS_SHAPE, GS_GAIN, GS_LVL0, GS_LVL1,
       GS_LVL2 };
enum { PL_INC1 = 0, PL_INC2, PL_GAMP1, PL_GAMP2, PL_GAMPN, PL_GARG1, PL_GARG2, PL_AB };

// (wave_incl_scan: DPP moves, wave_ops.h -- a shuffle of doubles is two ds_bpermute round trips per step, and the
// chunk loops below are chains of such scans)
__device__ __forceinline__ double wave_total(double v) { return wave_sum(v); }

// Inclusive scans over the workgroup (thread order) of one value per thread and oscillator; the scanned values come
// back in place and the workgroup totals are added to the carries (same value in every thread).
// Both scans share one pair of barriers.  s_w: 2 * GRAD_WAVES doubles of LDS.
__device__ __forceinline__ void block_incl_scan2(double& v1, double& v2, double& carry1, double& carry2, double* s_w,
                                                 int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  double s1 = wave_incl_scan(v1, lane), s2 = wave_incl_scan(v2, lane);
  __syncthreads();                       // s_w free again
  if (lane == 63) { s_w[wave] = s1; s_w[GRAD_WAVES + wave] = s2; }
  __syncthreads();
  double b1 = 0.0, t1 = 0.0, b2 = 0.0, t2 = 0.0;
#pragma unroll
  for (int w = 0; w < GRAD_WAVES; ++w) {
    const double x1 = s_w[w], x2 = s_w[GRAD_WAVES + w];
    if (w < wave) { b1 += x1; b2 += x2; }
    t1 += x1; t2 += x2;
  }
  v1 = s1 + b1 + carry1; v2 = s2 + b2 + carry2;
  carry1 += t1; carry2 += t2;
}

// Workgroup sum of NS per-thread doubles -> out[NS] (thread 0 writes); fixed order.
template <int NS>
__device__ __forceinline__ void block_sums(const double (&acc)[NS], double* out, double* s_red /* [GRAD_WAVES][NS] */, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double t = wave_total(acc[k]);
    if (lane == 0) s_red[wave * NS + k] = t;
  }
  __syncthreads();
  if (tid < NS) {
    double t = 0.0;
    for (int w = 0; w < GRAD_WAVES; ++w) t += s_red[w * NS + tid];
    out[tid] = t;
  }
}

__device__ __forceinline__ float grad_lerp(const float* __restrict__ row, int i0, int i1, float w0, float w1) {
  return ias_lerp(row[i0], row[i1], w0, w1);
}

// ------------------------------------------------------------------------------------------------ K0
__global__ __launch_bounds__(GRAD_THREADS) void voice_grad_inc_kernel(
    const float* __restrict__ ctrl, const IasVoiceConst* __restrict__ vconst, float* __restrict__ planes,
    double* __restrict__ tile_sums /* [B][ntiles][2] */, int T, int Tc, int ntiles, double inv_sample_rate,
    float scale) {
  __shared__ double s_red[GRAD_WAVES * 2];
  const int tid = threadIdx.x, tile = blockIdx.x, b = blockIdx.y;
  const IasVoiceConst vc = vconst[b];
  const float* cb = ctrl + (size_t)b * IAS_NCTRL * Tc;
  float* p1 = planes + ((size_t)b * IAS_GRAD_PLANES + PL_INC1) * T;
  float* p2 = planes + ((size_t)b * IAS_GRAD_PLANES + PL_INC2) * T;
  double acc[2] = {0.0, 0.0};
  for (int it = 0; it < GRAD_CHUNKS; ++it) {
    const int j = tile * GRAD_TILE + it * GRAD_THREADS + tid;
    if (j < T) {
      int i0, i1; float w0, w1;
      ias_interp_pos_fast(j, scale, Tc, i0, i1, w0, w1);
      const float pm1 = grad_lerp(cb, i0, i1, w0, w1);
      const float pm2 = grad_lerp(cb + 2 * Tc, i0, i1, w0, w1);
      const float a = ias_vco_inc_fast(vc.f0_1, vc.depth_1, pm1, inv_sample_rate);
      const float d = ias_vco_inc_fast(vc.f0_2, vc.depth_2, pm2, inv_sample_rate);
      // torch.clamp passes the gradient on [0, 127]; outside, the increment is stored negated as the flag
      const float c1 = ias_add(vc.f0_1, ias_mul(vc.depth_1, pm1)), c2 = ias_add(vc.f0_2, ias_mul(vc.depth_2, pm2));
      p1[j] = (c1 >= 0.0f && c1 <= 127.0f) ? a : -a;
      p2[j] = (c2 >= 0.0f && c2 <= 127.0f) ? d : -d;
      acc[0] += (double)a;
      acc[1] += (double)d;
    }
  }
  block_sums<2>(acc, tile_sums + ((size_t)b * ntiles + tile) * 2, s_red, tid);
}

// c1, c2 = sums over the tiles t of [lo, hi) -- ascending t, or descending when DESC -- of src[t * stride + off1 / off2],
// in every thread.  The values are fetched by up to GRAD_THREADS lanes at once and added from LDS in the stated order
// (the straightforward loop was a chain of up to ntiles dependent global loads at the head of every workgroup).
template <bool DESC>
__device__ __forceinline__ void tile_carry(const double* __restrict__ src, size_t stride, int off1, int off2, int lo, int hi,
                                           double* s_buf /* [2][GRAD_THREADS] */, int tid, double& c1, double& c2) {
  c1 = 0.0; c2 = 0.0;
  const int n = hi - lo;
  for (int base = 0; base < n; base += GRAD_THREADS) {
    const int m = min(GRAD_THREADS, n - base);
    if (tid < m) {
      const int t = DESC ? hi - 1 - (base + tid) : lo + base + tid;
      s_buf[tid] = src[(size_t)t * stride + off1];
      s_buf[GRAD_THREADS + tid] = src[(size_t)t * stride + off2];
    }
    __syncthreads();
    for (int i = 0; i < m; ++i) { c1 += s_buf[i]; c2 += s_buf[GRAD_THREADS + i]; }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ K1
__global__ __launch_bounds__(GRAD_THREADS) void voice_grad_sample_kernel(
    const float* __restrict__ ctrl, const IasVoiceConst* __restrict__ vconst, const float* __restrict__ noise,
    const float* __restrict__ g_mixed, float* __restrict__ planes, const double* __restrict__ tile_sums,
    double* __restrict__ partials /* [B][ntiles][IAS_GRAD_NS] */, int T, int Tc, int ntiles, float scale,
    const float* __restrict__ rownorm /* NULL, or [B][4]: peak divisor, index of the peak sample, its correction */) {
  __shared__ double s_w[2 * GRAD_WAVES];
  __shared__ double s_red[GRAD_WAVES * 8];
  const int tid = threadIdx.x, tile = blockIdx.x, b = blockIdx.y;
  // the cotangent of the un-normalised mix from that of the normalised audio (voice_norm_finish_kernel)
  const float n_div = rownorm ? rownorm[4 * b] : 1.0f, n_corr = rownorm ? rownorm[4 * b + 2] : 0.0f;
  const int n_tstar = rownorm ? __float_as_int(rownorm[4 * b + 1]) : -1;
  const IasVoiceConst vc = vconst[b];
  const float* cb = ctrl + (size_t)b * IAS_NCTRL * Tc;
  float* pl = planes + (size_t)b * IAS_GRAD_PLANES * T;
  const float* nrow = noise + (size_t)b * T;
  const float* grow = g_mixed + (size_t)b * T;

  // carry-in: the increments of the earlier tiles (fp64 sums of fp32 values below 2^19 are exact)
  double carry1, carry2;
  {
    __shared__ double s_carry[2 * GRAD_THREADS];
    tile_carry<false>(tile_sums + (size_t)b * ntiles * 2, 2, 0, 1, 0, tile, s_carry, tid, carry1, carry2);
  }
  // lvl0 lvl1 lvl2 kpart shape gain phi_1 phi_2
  double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  // The chunk loop is a chain of workgroup scans (barriers): a chunk's global inputs are requested one chunk ahead
  // so that their latency does not sit between the barriers.  (All 16 chunks up front: 194 VGPRs, 330 -> 490 us.)
  auto fetch = [&](int it, float& a1, float& a2, float& nzv, float& gv) {
    const int j = tile * GRAD_TILE + it * GRAD_THREADS + tid;
    const bool ok = it < GRAD_CHUNKS && j < T;
    a1 = ok ? fabsf(pl[(size_t)PL_INC1 * T + j]) : 0.0f;
    a2 = ok ? fabsf(pl[(size_t)PL_INC2 * T + j]) : 0.0f;
    nzv = ok ? nrow[j] : 0.0f;
    gv = ok ? grow[j] : 0.0f;
    if (rownorm) { gv = gv / n_div; if (j == n_tstar) gv = gv + n_corr; }
  };
  float n_inc1, n_inc2, n_nz, n_g;
  fetch(0, n_inc1, n_inc2, n_nz, n_g);
  for (int it = 0; it < GRAD_CHUNKS; ++it) {
    const int j = tile * GRAD_TILE + it * GRAD_THREADS + tid;
    const bool ok = j < T;
    const double inc1 = (double)n_inc1, inc2 = (double)n_inc2;
    const float nz = n_nz, g = n_g;
    fetch(it + 1, n_inc1, n_inc2, n_nz, n_g);
    double cum1 = inc1, cum2 = inc2;
    block_incl_scan2(cum1, cum2, carry1, carry2, s_w, tid);
    if (!ok) continue;
    const float arg1 = ias_add((float)cum1, vc.phi_1), arg2 = ias_add((float)cum2, vc.phi_2);
    int i0, i1; float w0, w1;
    ias_interp_pos_fast(j, scale, Tc, i0, i1, w0, w1);
    const float amp1 = grad_lerp(cb + 1 * Tc, i0, i1, w0, w1);
    const float amp2 = grad_lerp(cb + 3 * Tc, i0, i1, w0, w1);
    const float ampn = grad_lerp(cb + 4 * Tc, i0, i1, w0, w1);
    float s1, c1, s2, c2;
    ias_sincos_dev(arg1, s1, c1);
    ias_sincos_dev(arg2, s2, c2);
    const float th = ias_tanh_dev(vc.kpart * s2 * 0.5f);
    const float env2 = 1.0f + vc.shape * c2;
    const float core2 = vc.shape_gain * th * env2;
    pl[(size_t)PL_GAMP1 * T + j] = g * vc.lvl0 * c1;
    pl[(size_t)PL_GAMP2 * T + j] = g * vc.lvl1 * core2;
    pl[(size_t)PL_GAMPN * T + j] = g * vc.lvl2 * nz;
    const float g_arg1 = -g * vc.lvl0 * amp1 * s1;
    const float ga2 = g * vc.lvl1 * amp2;                      // d loss / d (gain * tanh * env2)
    const float sech2 = 1.0f - th * th;
    const float g_arg2 = ga2 * vc.shape_gain * (sech2 * (0.5f * vc.kpart) * c2 * env2 - th * vc.shape * s2);
    pl[(size_t)PL_GARG1 * T + j] = g_arg1;
    pl[(size_t)PL_GARG2 * T + j] = g_arg2;
    acc[0] += (double)(g * c1 * amp1);
    acc[1] += (double)(g * core2 * amp2);
    acc[2] += (double)(g * nz * ampn);
    acc[3] += (double)(ga2 * vc.shape_gain * sech2 * (0.5f * s2) * env2);
    acc[4] += (double)(ga2 * vc.shape_gain * th * c2);
    acc[5] += (double)(ga2 * th * env2);
    acc[6] += (double)g_arg1;
    acc[7] += (double)g_arg2;
  }
  __shared__ double s_out[8];
  block_sums<8>(acc, s_out, s_red, tid);
  __syncthreads();
  if (tid == 0) {
    double* o = partials + ((size_t)b * ntiles + tile) * IAS_GRAD_NS;
    o[GS_LVL0] = s_out[0]; o[GS_LVL1] = s_out[1]; o[GS_LVL2] = s_out[2];
    o[GS_KPART] = s_out[3]; o[GS_SHAPE] = s_out[4]; o[GS_GAIN] = s_out[5];
    o[GS_PHI_1] = s_out[6]; o[GS_PHI_2] = s_out[7];
  }
}

// ------------------------------------------------------------------------------------------------ K0 (normalisation)
// audio = mix / peak on rows with peak = max |mix| > 1 (torchsynth normalize_if_clipping), attained at t*:
//   g_mix[t] = g[t] / peak,  and the peak takes  -sign(audio[t*]) sum_t g[t] audio[t] / peak  at t*
// (autograd's max-of-abs convention: the whole gradient to the first sample that attains the maximum).  Two launches
// instead of the 14 torch ones: per-tile partial (dot, max |audio|, its first index), then per row the fixed-order finish
// -> rownorm [B][4] = {divisor (peak or 1), t* (int bits; -1: no correction), correction, 0}, applied by K1 as it reads g.
#define NORM_TILE (GRAD_THREADS * 4 * 8)
__global__ __launch_bounds__(GRAD_THREADS) void voice_norm_partial_kernel(const float* __restrict__ g,
                                                                          const float* __restrict__ audio, int T,
                                                                          int ntiles, double* __restrict__ part /* [B][ntiles][2] */) {
  __shared__ double s_dot[GRAD_WAVES];
  __shared__ float s_max[GRAD_WAVES];
  __shared__ int s_idx[GRAD_WAVES];
  const int tid = threadIdx.x, tile = blockIdx.x, b = blockIdx.y;
  const float* grow = g + (size_t)b * T;
  const float* arow = audio + (size_t)b * T;
  double dot = 0.0;
  float mx = -1.0f;
  int mi = 0x7fffffff;
  for (int it = 0; it < 8; ++it) {
    const int j0 = tile * NORM_TILE + (it * GRAD_THREADS + tid) * 4;
    for (int e = 0; e < 4; ++e) {
      const int j = j0 + e;
      if (j < T) {
        const float av = arow[j];
        dot += (double)(grow[j] * av);
        const float m = fabsf(av);
        if (m > mx) { mx = m; mi = j; }        // j ascends within the thread: the first index of its maximum
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    dot += __shfl_xor(dot, d, 64);
    const float om = __shfl_xor(mx, d, 64);
    const int oi = __shfl_xor(mi, d, 64);
    if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
  }
  if ((tid & 63) == 0) { s_dot[tid >> 6] = dot; s_max[tid >> 6] = mx; s_idx[tid >> 6] = mi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < GRAD_WAVES; ++w) {
      dot += s_dot[w];
      if (s_max[w] > mx || (s_max[w] == mx && s_idx[w] < mi)) { mx = s_max[w]; mi = s_idx[w]; }
    }
    double* o = part + ((size_t)b * ntiles + tile) * 2;
    o[0] = dot;
    o[1] = __hiloint2double(__float_as_int(mx), mi);
  }
}

__global__ void voice_norm_finish_kernel(const double* __restrict__ part, const float* __restrict__ audio,
                                         const float* __restrict__ peaks, int B, int T, int ntiles,
                                         float* __restrict__ rownorm) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double dot = 0.0;
  float mx = -1.0f;
  int mi = 0;
  for (int t = 0; t < ntiles; ++t) {
    const double* o = part + ((size_t)b * ntiles + t) * 2;
    dot += o[0];
    const float om = __int_as_float(__double2hiint(o[1]));
    const int oi = __double2loint(o[1]);
    if (om > mx) { mx = om; mi = oi; }          // tiles ascend: ties keep the earlier index
  }
  const float pk = peaks[b];
  const bool clip = pk > 1.0f;
  const float av = audio[(size_t)b * T + mi];
  const float sg = av > 0.0f ? 1.0f : (av < 0.0f ? -1.0f : 0.0f);
  rownorm[4 * b] = clip ? pk : 1.0f;
  rownorm[4 * b + 1] = __int_as_float(clip ? mi : -1);
  rownorm[4 * b + 2] = clip ? -sg * (float)dot / pk : 0.0f;
  rownorm[4 * b + 3] = 0.0f;
}

// ------------------------------------------------------------------------------------------------ K2
__global__ __launch_bounds__(GRAD_THREADS) void voice_grad_pitch_kernel(
    const float* __restrict__ ctrl, const IasVoiceConst* __restrict__ vconst, float* __restrict__ planes,
    double* __restrict__ partials, int T, int Tc, int ntiles, float scale) {
  __shared__ double s_w[2 * GRAD_WAVES];
  __shared__ double s_red[GRAD_WAVES * 4];
  const int tid = threadIdx.x, tile = blockIdx.x, b = blockIdx.y;
  const IasVoiceConst vc = vconst[b];
  const float* cb = ctrl + (size_t)b * IAS_NCTRL * Tc;
  float* pl = planes + (size_t)b * IAS_GRAD_PLANES * T;
  // carry-in of the reverse scan: g_arg totals of the later tiles (K1 left them in the phi slots)
  double carry1, carry2;
  {
    __shared__ double s_carry[2 * GRAD_THREADS];
    tile_carry<true>(partials + (size_t)b * ntiles * IAS_GRAD_NS, IAS_GRAD_NS, GS_PHI_1, GS_PHI_2, tile + 1, ntiles, s_carry, tid,
                     carry1, carry2);
  }
  const double k = 0.6931471805599453 / 12.0;   // d inc / d pitch = inc * ln2 / 12
  double acc[4] = {0.0, 0.0, 0.0, 0.0};        // f0_1 depth_1 f0_2 depth_2
  auto fetch = [&](int it, float& a1, float& a2, float& i1, float& i2) {   // one chunk ahead, see K1
    const int j = tile * GRAD_TILE + it * GRAD_THREADS + (GRAD_THREADS - 1 - tid);
    const bool ok = it >= 0 && j < T;
    a1 = ok ? pl[(size_t)PL_GARG1 * T + j] : 0.0f;
    a2 = ok ? pl[(size_t)PL_GARG2 * T + j] : 0.0f;
    i1 = ok ? pl[(size_t)PL_INC1 * T + j] : 0.0f;
    i2 = ok ? pl[(size_t)PL_INC2 * T + j] : 0.0f;
  };
  float n_ga1, n_ga2, n_inc1, n_inc2;
  fetch(GRAD_CHUNKS - 1, n_ga1, n_ga2, n_inc1, n_inc2);
  for (int it = GRAD_CHUNKS - 1; it >= 0; --it) {
    const int j = tile * GRAD_TILE + it * GRAD_THREADS + (GRAD_THREADS - 1 - tid);   // descending in tid
    const bool ok = j < T;
    const double ga1 = (double)n_ga1, ga2 = (double)n_ga2;
    const float inc1 = n_inc1, inc2 = n_inc2;
    fetch(it - 1, n_ga1, n_ga2, n_inc1, n_inc2);
    double suf1 = ga1, suf2 = ga2;
    block_incl_scan2(suf1, suf2, carry1, carry2, s_w, tid);
    if (!ok) continue;
    const double gc1 = inc1 > 0.0f ? suf1 * ((double)inc1 * k) : 0.0;
    const double gc2 = inc2 > 0.0f ? suf2 * ((double)inc2 * k) : 0.0;
    int i0, i1; float w0, w1;
    ias_interp_pos_fast(j, scale, Tc, i0, i1, w0, w1);
    const float pm1 = grad_lerp(cb, i0, i1, w0, w1);
    const float pm2 = grad_lerp(cb + 2 * Tc, i0, i1, w0, w1);
    acc[0] += gc1; acc[1] += gc1 * (double)pm1;
    acc[2] += gc2; acc[3] += gc2 * (double)pm2;
    pl[(size_t)PL_GARG1 * T + j] = (float)(gc1 * (double)vc.depth_1);
    pl[(size_t)PL_GARG2 * T + j] = (float)(gc2 * (double)vc.depth_2);
  }
  __shared__ double s_out[4];
  block_sums<4>(acc, s_out, s_red, tid);
  __syncthreads();
  if (tid == 0) {
    double* o = partials + ((size_t)b * ntiles + tile) * IAS_GRAD_NS;
    o[GS_F0_1] = s_out[0]; o[GS_DEPTH_1] = s_out[1]; o[GS_F0_2] = s_out[2]; o[GS_DEPTH_2] = s_out[3];
  }
}

// ------------------------------------------------------------------------------------------------ K3
// Transposed linear upsample.  Sample j lerps between control points i0(j) = trunc(scale * j) and i0 + 1 with
// weights (w0, w1), so g_ctrl[i] = A[i] + B[i-1] with the per-INTERVAL sums A[k] = sum_{i0(j)=k} w0 g,
// B[k] = sum_{i0(j)=k} w1 g: every sample is read once.  A workgroup takes CT_INTERVALS consecutive intervals of
// one row: their samples are one contiguous run (i0 is monotone), staged as (w0 g, w1 g) in LDS by coalesced
// loads, then each wave sums whole intervals in a fixed order (no atomics: bit-reproducible).
#define CT_INTERVALS 40
#define CT_SUB 6              // threads per interval in the reduction (CT_INTERVALS * CT_SUB <= GRAD_THREADS)
#define CT_CAP 4608          // staged samples per workgroup (the host picks nint with (nint + 1) * (1/scale + 2) <= cap)

// the voice's interval sums [5][Tc][2] fp64 live in its spare scratch plane (8-byte aligned inside it)
__device__ __forceinline__ double* ct_interval_sums(float* planes, int b, int T) {
  const uintptr_t p = (uintptr_t)(planes + ((size_t)b * IAS_GRAD_PLANES + PL_AB) * T);
  return reinterpret_cast<double*>((p + 7) & ~(uintptr_t)7);
}

// first sample index whose control index is >= k (T if there is none); exact w.r.t. the forward's fp32 arithmetic
__device__ __forceinline__ int ct_first_sample(int k, float scale, int T) {
  if (k <= 0) return 0;
  long long j = (long long)ceil((double)k / (double)scale);
  if (j > T) j = T;
  while (j > 0 && (int)ias_mul(scale, (float)(j - 1)) >= k) --j;
  while (j < T && (int)ias_mul(scale, (float)j) < k) ++j;
  return (int)j;
}

__global__ __launch_bounds__(GRAD_THREADS) void voice_grad_ctrl_kernel(
    float* __restrict__ planes, int T, int Tc, float scale, int nint /* intervals per workgroup, <= CT_INTERVALS */) {
  __shared__ float s_p0[CT_CAP], s_p1[CT_CAP];
  __shared__ int s_start[CT_INTERVALS + 1];
  __shared__ int s_range[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = blockIdx.x * nint, row = blockIdx.y, b = blockIdx.z;
  const int k1 = min(k0 + nint, Tc), nk = k1 - k0;
  const int plane_of_row[IAS_NCTRL] = {PL_GARG1, PL_GAMP1, PL_GARG2, PL_GAMP2, PL_GAMPN};
  const float* src = planes + ((size_t)b * IAS_GRAD_PLANES + plane_of_row[row]) * T;
  if (tid < 2) s_range[tid] = ct_first_sample(tid == 0 ? k0 : k1, scale, T);
  if (tid <= CT_INTERVALS) s_start[tid] = -1;
  __syncthreads();
  const int j_lo = s_range[0], n = s_range[1] - s_range[0];
  float v_g[CT_CAP / GRAD_THREADS];
#pragma unroll
  for (int q = 0; q < CT_CAP / GRAD_THREADS; ++q) {          // all loads in flight before the first use
    const int o = tid + q * GRAD_THREADS;
    v_g[q] = o < n ? src[j_lo + o] : 0.0f;
  }
#pragma unroll
  for (int q = 0; q < CT_CAP / GRAD_THREADS; ++q) {
    const int o = tid + q * GRAD_THREADS;
    if (o >= n) break;
    const int j = j_lo + o;
    int i0, i1; float w0, w1;
    ias_interp_pos_fast(j, scale, Tc, i0, i1, w0, w1);
    const float g = v_g[q];
    s_p0[o] = w0 * g;
    s_p1[o] = w1 * g;
    if (o == 0 || (int)ias_mul(scale, (float)(j - 1)) != i0) s_start[i0 - k0] = o;   // first sample of its interval
  }
  if (tid == 0) s_start[nk] = n;
  __syncthreads();
  // CT_SUB threads per interval, each a strided run of its samples; then one thread adds the CT_SUB partial sums in
  // a fixed order.  (A wave per interval with shuffle reductions was 3x slower: 12 dependent LDS-latency shuffles
  // per interval.)
  __shared__ double s_part[CT_INTERVALS][CT_SUB][2];
  {
    const int kk = tid / CT_SUB, sub = tid % CT_SUB;
    if (kk < nk) {
      const int lo = s_start[kk];
      int hi = n;                                      // next interval that has samples (every one does for Tc <= T)
      for (int q = kk + 1; q <= nk; ++q) if (s_start[q] >= 0) { hi = s_start[q]; break; }
      double a = 0.0, c = 0.0;
      if (lo >= 0)
        for (int o = lo + sub; o < hi; o += CT_SUB) { a += (double)s_p0[o]; c += (double)s_p1[o]; }
      s_part[kk][sub][0] = a; s_part[kk][sub][1] = c;
    }
  }
  __syncthreads();
  if (tid < nk) {
    double a = 0.0, c = 0.0;
#pragma unroll
    for (int sub = 0; sub < CT_SUB; ++sub) { a += s_part[tid][sub][0]; c += s_part[tid][sub][1]; }
    double* out = ct_interval_sums(planes, b, T) + ((size_t)row * Tc + k0) * 2;
    out[2 * tid] = a; out[2 * tid + 1] = c;
  }
}

// g_ctrl[i] = A[i] + B[i-1]; the last control point is its own upper neighbour (i1 = min(i0 + 1, Tc - 1))
__global__ __launch_bounds__(GRAD_THREADS) void voice_grad_ctrl_combine_kernel(float* __restrict__ planes,
                                                                               float* __restrict__ g_ctrl, int T,
                                                                               int Tc) {
  const int i = blockIdx.x * GRAD_THREADS + threadIdx.x, row = blockIdx.y, b = blockIdx.z;
  if (i >= Tc) return;
  const double* p = ct_interval_sums(planes, b, T) + (size_t)row * Tc * 2;
  double v = p[2 * i];
  if (i > 0) v += p[2 * (i - 1) + 1];
  if (i == Tc - 1) v += p[2 * i + 1];
  g_ctrl[((size_t)b * IAS_NCTRL + row) * Tc + i] = (float)v;
}


// ================================================================================================ lane-consecutive form
// K1 / K2 / K3 again for rows with at least 16 samples per control interval (scale * 16 <= 1; every shape the reference
// uses: 100 samples per interval), with the sample-to-thread mapping of the forward kernel: a thread owns G16_SPT
// CONSECUTIVE samples of the tile.
//   * the phase of a sample is (tile carry) + (threads before it: ONE wave scan of the thread totals + the earlier waves'
//     totals) + a running sum inside the thread: two barriers per tile instead of two per 256 samples (the chunk loop of
//     the first form was a chain of 16 workgroup scans), no DPP scan per sample.
//   * global memory is touched in whole 1 KB rows per wave instruction, as in the forward kernel: the wave's 1024 samples
//     of each input plane arrive in a 4 KB LDS block by LDS-DMA (16 KB per wave for the four planes), a lane reads its
//     16 samples from there, the two g_arg planes leave through the blocks of the increments.  (Lanes reading their 64
//     bytes straight from global memory -- 16-byte pieces 64 bytes apart -- made the kernel 2.4x slower than its
//     arithmetic: 142 us for the loads and stores alone.)
//   * the transposed upsample is folded in: a thread's samples lie in at most two control intervals (k_first, k_first + 1),
//     so it leaves per row the two partial interval sums (A = sum w0 g, B = sum w1 g; accumulated in fp32 over its <= 16
//     samples), and one thread per (row, interval) adds the partials of the threads that touch the interval in sample
//     order, in fp64 -> per-TILE interval sums [B][ntiles][5][kslots][2] (fixed order, no atomics).  The three amplitude
//     planes and the two pitch planes of the first form are never written, K3's pass over five planes is gone;
//     voice_grad_fold_combine_kernel adds the <= 2 tiles an interval lies in.
//   * sin / cos / tanh: the forward kernel's fp32 revolution split (voice_trig.h) instead of the fp64 reduction.
// A tile that is not whole (the row's last) or a row length that is not a multiple of 4 takes the same code with guarded
// scalar accesses instead of the LDS blocks.
#define G16_SPT GRAD_CHUNKS
#define G16_NQ (G16_SPT / 4)     // 16-byte granules per lane
#define G16_BLK (64 * G16_SPT)   // floats of one plane block of a wave
#define G16_NOK 0x3fffffff       // k_first of a thread that owns no sample
#define G16_WAVE_FLOATS (4 * G16_BLK)     // staging per wave: four plane blocks
#define G16_LDS_BYTES (GRAD_WAVES * G16_WAVE_FLOATS * 4)
static_assert(G16_SPT == 8 || G16_SPT == 16, "lane-consecutive kernels: 8 or 16 samples per thread");
typedef float g16_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void g16_lds_void;
typedef const __attribute__((address_space(1))) void g16_glb_void;

__device__ __forceinline__ double* g16_tile_sums(float* planes, int b, int T, int tile, int kslots) {
  return ct_interval_sums(planes, b, T) + (size_t)tile * IAS_NCTRL * kslots * 2;
}
__device__ __forceinline__ const double* g16_tile_sums(const float* planes, int b, int T, int tile, int kslots) {
  return ct_interval_sums(const_cast<float*>(planes), b, T) + (size_t)tile * IAS_NCTRL * kslots * 2;
}

// The wave's 64 * G16_SPT samples from j_wave on of a plane row -> its LDS block, by LDS-DMA.  Block layout (the forward
// kernel's, for 4 or 2 granules per position): position p's samples are its own 64 / 32 bytes, its 16-byte granules
// XOR-permuted by (p >> 2) & 3 / (p >> 3) & 1 -- 16 consecutive positions then cover all 64 banks with each granule index:
// conflict-free ds_read/write_b128 -- and lane-linear (the DMA's destination order) once the SOURCE granule of DMA lane i
// is i ^ ((i >> 4) & (G16_NQ - 1)).
__device__ __forceinline__ void g16_dma(const float* __restrict__ row, int j_wave, float* blk, int lane) {
  const float* src = row + j_wave + 4 * (lane ^ ((lane >> 4) & (G16_NQ - 1)));
#pragma unroll
  for (int q = 0; q < G16_NQ; ++q)
    __builtin_amdgcn_global_load_lds((g16_glb_void*)(src + q * 256), (g16_lds_void*)(blk + q * 256), 16, 0, 0);
}
__device__ __forceinline__ float* g16_slot(float* blk, int pos, int q) {
  return blk + pos * G16_SPT + 4 * (q ^ ((pos >> (G16_NQ == 4 ? 2 : 3)) & (G16_NQ - 1)));
}
// the block back to a plane row, whole 1 KB rows per instruction
__device__ __forceinline__ void g16_copy_out(float* __restrict__ row, int j_wave, const float* blk, int lane) {
  const int dst = 4 * (lane ^ ((lane >> 4) & (G16_NQ - 1)));
#pragma unroll
  for (int q = 0; q < G16_NQ; ++q)
    *reinterpret_cast<g16_f4*>(row + j_wave + q * 256 + dst) = *reinterpret_cast<const g16_f4*>(blk + q * 256 + lane * 4);
}
// four consecutive samples (group q of position pos): from the wave's block, or guarded from the row itself
template <bool FAST>
__device__ __forceinline__ void g16_get4(const float* blk, int pos, int q, const float* __restrict__ row, int j, int T,
                                         float (&v)[4]) {
  if (FAST) {
    const g16_f4 x = *reinterpret_cast<const g16_f4*>(g16_slot(const_cast<float*>(blk), pos, q));
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
  } else {
#pragma unroll
    for (int x = 0; x < 4; ++x) v[x] = j + x < T ? row[j + x] : 0.0f;
  }
}

// LDS of a lane-consecutive kernel besides the staging blocks
struct G16Shared {
  double w[2 * GRAD_WAVES];
  double red[GRAD_WAVES * 8];
  double carry[2 * GRAD_THREADS];
  int kfirst[GRAD_THREADS];
  double out[8];
};
// The fold partials {A, B of interval k_first; A, B of interval k_first + 1} of fold row r, sample-order position p of the
// tile: kept in the last TWO plane blocks of the wave that owns the position (free once its sample loop is done).
// REV: wave w owns positions 64 (3 - w) ... (K2 walks the tile backwards).
template <bool REV>
__device__ __forceinline__ g16_f4* g16_fold_slot(float* stage, int r, int p) {
  const int w = REV ? GRAD_WAVES - 1 - (p >> 6) : (p >> 6);
  return reinterpret_cast<g16_f4*>(stage + w * G16_WAVE_FLOATS + 2 * G16_BLK + r * 256 + (p & 63) * 4);
}
// s_kfirst[p] non-decreasing in p (G16_NOK: none).  Thread idx < NR * kslots adds, for interval k_lo + kk, the partials of
// the positions with k_first in {k - 1, k} in ascending p -> out[(rows[r] * kslots + kk) * 2 + {0, 1}].
// Only ~3 x 45 threads of the workgroup work here while the rest wait at the kernel's end, so the step is a latency chain:
// the first position comes from arithmetic (an under-estimate, corrected by at most a few probes; a binary search was 8
// dependent LDS reads), and the partials are read eight positions at a time (16 reads in flight) and added in order
// (95 / 61 -> 94 / 60 us; with the whole transposed upsample compiled out -- this step AND the per-sample interval partials --
// the two kernels take 79 / 46 us).
template <int NR, bool REV>
__device__ __forceinline__ void g16_fold(float* stage, const int* s_kfirst, int k_lo, int kslots, const int (&rows)[NR],
                                         double* __restrict__ out, int tid, float scale, int tile_base) {
  const float inv = (1.0f / scale) * (1.0f / (float)G16_SPT);
  for (int idx = tid; idx < NR * kslots; idx += GRAD_THREADS) {
    const int r = idx / kslots, kk = idx - r * kslots, k = k_lo + kk;
    // first p with k_first(p) >= k - 1: k_first(p) = trunc(scale (tile_base + SPT p)), so p >= ((k - 1) / scale - tile_base) / SPT
    int p = (int)floorf((float)(k - 1) * inv - (float)tile_base * (1.0f / (float)G16_SPT)) - 1;
    p = max(0, min(p, GRAD_THREADS - 1));
    // (a step back is only ever needed if the estimate was rounded up; positions without samples carry G16_NOK and end it)
    for (int g = 0; g < 4 && p > 0 && s_kfirst[p - 1] >= k - 1 && s_kfirst[p - 1] != G16_NOK; ++g) --p;
    while (p < GRAD_THREADS && s_kfirst[p] < k - 1) ++p;
    double A = 0.0, B = 0.0;
    bool done = false;
    for (int base = p; base < GRAD_THREADS && !done; base += 8) {
      int kf[8]; g16_f4 f[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pp = min(base + u, GRAD_THREADS - 1);
        kf[u] = base + u < GRAD_THREADS ? s_kfirst[pp] : G16_NOK;
        f[u] = *g16_fold_slot<REV>(stage, r, pp);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (done || kf[u] > k) { done = true; continue; }
        if (kf[u] == k) { A += (double)f[u].x; B += (double)f[u].y; } else { A += (double)f[u].z; B += (double)f[u].w; }
      }
    }
    out[((size_t)rows[r] * kslots + kk) * 2] = A;
    out[((size_t)rows[r] * kslots + kk) * 2 + 1] = B;
  }
}
// the three control points a thread's samples interpolate between (its run lies in at most two intervals)
struct G16Ctrl { float v0, v1, v2; };
__device__ __forceinline__ G16Ctrl g16_ctrl(const float* __restrict__ crow, int k_first, int Tc) {
  G16Ctrl c;
  c.v0 = crow[min(k_first, Tc - 1)]; c.v1 = crow[min(k_first + 1, Tc - 1)]; c.v2 = crow[min(k_first + 2, Tc - 1)];
  return c;
}
// lerp of a sample in interval k_first (first) or k_first + 1 -- the arithmetic of ias_lerp on the same operands
__device__ __forceinline__ float g16_lerp(const G16Ctrl& c, bool first, float w0, float w1) {
  return ias_lerp(first ? c.v0 : c.v1, first ? c.v1 : c.v2, w0, w1);
}

template <bool FAST>
__device__ __forceinline__ void g16_sample_tile(
    G16Shared& sh, float* stage, const float* __restrict__ cb, const IasVoiceConst& vc, const float* __restrict__ nrow,
    const float* __restrict__ grow, float* __restrict__ pl, const double* __restrict__ tile_sums_b,
    double* __restrict__ partials_bt, double* __restrict__ isum_bt, int T, int Tc, int tile, float scale, float n_div,
    float n_corr, int n_tstar, bool norm, int kslots) {
  double* s_w = sh.w; double* s_red = sh.red; double* s_carry = sh.carry; double* s_out = sh.out;
  int* s_kfirst = sh.kfirst;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j_wave = tile * GRAD_TILE + wave * G16_BLK, j0 = j_wave + lane * G16_SPT;
  const float* pi1 = pl + (size_t)PL_INC1 * T;
  const float* pi2 = pl + (size_t)PL_INC2 * T;
  float* blk = stage + wave * G16_WAVE_FLOATS;      // blocks: inc_1 (-> g_arg1) | inc_2 (-> g_arg2) | noise | g
  if (FAST) {
    g16_dma(pi1, j_wave, blk, lane);
    g16_dma(pi2, j_wave, blk + G16_BLK, lane);
    g16_dma(nrow, j_wave, blk + 2 * G16_BLK, lane);
    g16_dma(grow, j_wave, blk + 3 * G16_BLK, lane);
  }
  double carry1, carry2;
  tile_carry<false>(tile_sums_b, 2, 0, 1, 0, tile, s_carry, tid, carry1, carry2);
  if (FAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the wave's own DMA has landed
  double tot1 = 0.0, tot2 = 0.0;
#pragma unroll
  for (int q = 0; q < G16_NQ; ++q) {
    float a1[4], a2[4];
    g16_get4<FAST>(blk, lane, q, pi1, j0 + 4 * q, T, a1);
    g16_get4<FAST>(blk + G16_BLK, lane, q, pi2, j0 + 4 * q, T, a2);
#pragma unroll
    for (int x = 0; x < 4; ++x) { tot1 += (double)fabsf(a1[x]); tot2 += (double)fabsf(a2[x]); }   // sign = clamp flag (K0)
  }
  const double in1 = wave_incl_scan(tot1, lane), in2 = wave_incl_scan(tot2, lane);
  if (lane == 63) { s_w[wave] = in1; s_w[GRAD_WAVES + wave] = in2; }
  __syncthreads();
  double run1 = carry1 + (in1 - tot1), run2 = carry2 + (in2 - tot2);   // sums of fp32 values below 2^19: exact in any order
  for (int w = 0; w < wave; ++w) { run1 += s_w[w]; run2 += s_w[GRAD_WAVES + w]; }

  const float n_inv = 1.0f / n_div;            // (the first form divides every sample: one rounding apart)
  const int k_first = (int)ias_mul(scale, (float)j0);
  const G16Ctrl ca1 = g16_ctrl(cb + 1 * Tc, k_first, Tc), ca2 = g16_ctrl(cb + 3 * Tc, k_first, Tc),
                can = g16_ctrl(cb + 4 * Tc, k_first, Tc);
  // lvl0 lvl1 lvl2 kpart shape gain: fp32 over the thread's 16 samples, fp64 across threads and tiles; phi_1 phi_2 (the
  // g_arg totals K2's suffix sums continue from): fp64 throughout
  float accf[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  double accp[2] = {0.0, 0.0};
  float ta[3] = {0.f, 0.f, 0.f}, tw[3] = {0.f, 0.f, 0.f}, pa[3] = {0.f, 0.f, 0.f}, pw[3] = {0.f, 0.f, 0.f};
  // Four samples per trip of a loop that is NOT unrolled (fully unrolled, the 16 samples' live values spilled at every
  // register budget); the increments are read again from the block.
#pragma unroll 1
  for (int q = 0; q < G16_SPT / 4; ++q) {
    const int jq = j0 + 4 * q;
    float a1[4], a2[4], nz[4], gg[4], o1[4], o2[4];
    g16_get4<FAST>(blk, lane, q, pi1, jq, T, a1);
    g16_get4<FAST>(blk + G16_BLK, lane, q, pi2, jq, T, a2);
    g16_get4<FAST>(blk + 2 * G16_BLK, lane, q, nrow, jq, T, nz);
    g16_get4<FAST>(blk + 3 * G16_BLK, lane, q, grow, jq, T, gg);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const int j = jq + x;
      float g = gg[x];
      if (norm) { g = g * n_inv; if (j == n_tstar) g = g + n_corr; }
      run1 += (double)fabsf(a1[x]); run2 += (double)fabsf(a2[x]);
      const float arg1 = ias_add((float)run1, vc.phi_1), arg2 = ias_add((float)run2, vc.phi_2);
      const float real = ias_mul(scale, (float)j);
      const int i0 = (int)real;
      const float w1 = ias_sub(real, (float)i0), w0 = ias_sub(1.0f, w1);
      const bool first = i0 == k_first;
      const float amp1 = g16_lerp(ca1, first, w0, w1), amp2 = g16_lerp(ca2, first, w0, w1), ampn = g16_lerp(can, first, w0, w1);
      float fp, ft;
      voice_rev_split(arg1, fp, ft);
      const float s1 = __builtin_amdgcn_sinf(fp + ft), c1 = __builtin_amdgcn_cosf(fp + ft);
      float s2, c2;
      bool flip;
      voice_sincos(arg2, s2, c2, flip);
      if (flip) { s2 = -s2; c2 = -c2; }
      const float th = __builtin_copysignf(voice_tanh_abs(vc.kpart * s2 * 0.5f), s2);
      const float env2 = 1.0f + vc.shape * c2;
      const float core2 = vc.shape_gain * th * env2;
      const float x_amp1 = g * vc.lvl0 * c1, x_amp2 = g * vc.lvl1 * core2, x_ampn = g * vc.lvl2 * nz[x];
      const float g_arg1 = -g * vc.lvl0 * amp1 * s1;
      const float ga2 = g * vc.lvl1 * amp2;                      // d loss / d (gain * tanh * env2)
      const float sech2 = 1.0f - th * th;
      const float g_arg2 = ga2 * vc.shape_gain * (sech2 * (0.5f * vc.kpart) * c2 * env2 - th * vc.shape * s2);
      o1[x] = g_arg1; o2[x] = g_arg2;
      // samples beyond T carry g = 0: every term below vanishes
      accf[0] += g * c1 * amp1;
      accf[1] += g * core2 * amp2;
      accf[2] += g * nz[x] * ampn;
      accf[3] += ga2 * vc.shape_gain * sech2 * (0.5f * s2) * env2;
      accf[4] += ga2 * vc.shape_gain * th * c2;
      accf[5] += ga2 * th * env2;
      accp[0] += (double)g_arg1;
      accp[1] += (double)g_arg2;
      // the transposed upsample of the three amplitude rows: totals and the share of the thread's first interval
      const float m0 = first ? 1.0f : 0.0f;
      const float xs[3] = {x_amp1, x_amp2, x_ampn};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float xw = w1 * xs[r];
        ta[r] += xs[r]; tw[r] += xw;
        pa[r] = fmaf(m0, xs[r], pa[r]); pw[r] = fmaf(m0, xw, pw[r]);
      }
    }
    if (FAST) {      // the cotangents of the phases take the places of the increments they came from
      *reinterpret_cast<g16_f4*>(g16_slot(blk, lane, q)) = (g16_f4){o1[0], o1[1], o1[2], o1[3]};
      *reinterpret_cast<g16_f4*>(g16_slot(blk + G16_BLK, lane, q)) = (g16_f4){o2[0], o2[1], o2[2], o2[3]};
    } else {
#pragma unroll
      for (int x = 0; x < 4; ++x)
        if (jq + x < T) { pl[(size_t)PL_GARG1 * T + jq + x] = o1[x]; pl[(size_t)PL_GARG2 * T + jq + x] = o2[x]; }
    }
  }
  if (FAST) {
    g16_copy_out(pl + (size_t)PL_GARG1 * T, j_wave, blk, lane);
    g16_copy_out(pl + (size_t)PL_GARG2 * T, j_wave, blk + G16_BLK, lane);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r)
    *g16_fold_slot<false>(stage, r, tid) = (g16_f4){pa[r] - pw[r], pw[r], (ta[r] - pa[r]) - (tw[r] - pw[r]), tw[r] - pw[r]};
  s_kfirst[tid] = j0 < T ? k_first : G16_NOK;
  const double acc[8] = {(double)accf[0], (double)accf[1], (double)accf[2], (double)accf[3], (double)accf[4], (double)accf[5],
                         accp[0], accp[1]};
  block_sums<8>(acc, s_out, s_red, tid);      // its barriers also publish the fold partials and s_kfirst
  __syncthreads();
  if (tid == 0) {
    partials_bt[GS_LVL0] = s_out[0]; partials_bt[GS_LVL1] = s_out[1]; partials_bt[GS_LVL2] = s_out[2];
    partials_bt[GS_KPART] = s_out[3]; partials_bt[GS_SHAPE] = s_out[4]; partials_bt[GS_GAIN] = s_out[5];
    partials_bt[GS_PHI_1] = s_out[6]; partials_bt[GS_PHI_2] = s_out[7];
  }
  const int rows[3] = {1, 3, 4};               // amp1, amp2, ampn among the five control rows
  g16_fold<3, false>(stage, s_kfirst, (int)ias_mul(scale, (float)(tile * GRAD_TILE)), kslots, rows, isum_bt, tid, scale,
                     tile * GRAD_TILE);
}

__global__ __launch_bounds__(GRAD_THREADS, G16_SPT == 8 ? 4 : 2) void voice_grad_sample16_kernel(
    const float* __restrict__ ctrl, const IasVoiceConst* __restrict__ vconst, const float* __restrict__ noise,
    const float* __restrict__ g_mixed, float* __restrict__ planes, const double* __restrict__ tile_sums,
    double* __restrict__ partials, int T, int Tc, int ntiles, float scale, const float* __restrict__ rownorm, int kslots) {
  extern __shared__ __attribute__((aligned(16))) float g16_stage[];
  __shared__ G16Shared sh;
  const int tile = blockIdx.x, b = blockIdx.y;
  const float n_div = rownorm ? rownorm[4 * b] : 1.0f, n_corr = rownorm ? rownorm[4 * b + 2] : 0.0f;
  const int n_tstar = rownorm ? __float_as_int(rownorm[4 * b + 1]) : -1;
  const IasVoiceConst vc = vconst[b];
  const float* cb = ctrl + (size_t)b * IAS_NCTRL * Tc;
  float* pl = planes + (size_t)b * IAS_GRAD_PLANES * T;
  double* isum = g16_tile_sums(planes, b, T, tile, kslots);
  const bool fast = (T & 3) == 0 && (tile + 1) * GRAD_TILE <= T;
  if (fast)
    g16_sample_tile<true>(sh, g16_stage, cb, vc, noise + (size_t)b * T, g_mixed + (size_t)b * T, pl,
                          tile_sums + (size_t)b * ntiles * 2, partials + ((size_t)b * ntiles + tile) * IAS_GRAD_NS, isum, T, Tc,
                          tile, scale, n_div, n_corr, n_tstar, rownorm != nullptr, kslots);
  else
    g16_sample_tile<false>(sh, g16_stage, cb, vc, noise + (size_t)b * T, g_mixed + (size_t)b * T, pl,
                           tile_sums + (size_t)b * ntiles * 2, partials + ((size_t)b * ntiles + tile) * IAS_GRAD_NS, isum, T, Tc,
                           tile, scale, n_div, n_corr, n_tstar, rownorm != nullptr, kslots);
}

// K2 in the same form: suffix sums of g_arg (thread order = DESCENDING sample order: wave w owns the positions
// 64 (3 - w) ..., lane l the position 63 - l of those), the pitch-modulation cotangents folded into interval sums (rows 0
// and 2), partial sums for f0 and depth.
template <bool FAST>
__device__ __forceinline__ void g16_pitch_tile(G16Shared& sh, float* stage, const float* __restrict__ cb,
                                               const IasVoiceConst& vc, const float* __restrict__ pl,
                                               const double* __restrict__ partials_b, double* __restrict__ partials_bt,
                                               double* __restrict__ isum_bt, int T, int Tc, int ntiles, int tile, float scale,
                                               int kslots) {
  double* s_w = sh.w; double* s_red = sh.red; double* s_carry = sh.carry; double* s_out = sh.out;
  int* s_kfirst = sh.kfirst;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = GRAD_THREADS - 1 - tid, pos = 63 - lane;   // position of the thread's run in the tile / in its wave's block
  const int j_wave = tile * GRAD_TILE + (GRAD_WAVES - 1 - wave) * G16_BLK, j0 = j_wave + pos * G16_SPT;
  const float* pg1 = pl + (size_t)PL_GARG1 * T;
  const float* pg2 = pl + (size_t)PL_GARG2 * T;
  const float* pi1 = pl + (size_t)PL_INC1 * T;
  const float* pi2 = pl + (size_t)PL_INC2 * T;
  float* blk = stage + wave * G16_WAVE_FLOATS;      // blocks: g_arg1 | g_arg2 | inc_1 | inc_2
  if (FAST) {
    g16_dma(pg1, j_wave, blk, lane);
    g16_dma(pg2, j_wave, blk + G16_BLK, lane);
    g16_dma(pi1, j_wave, blk + 2 * G16_BLK, lane);
    g16_dma(pi2, j_wave, blk + 3 * G16_BLK, lane);
  }
  // carry-in of the reverse scan: g_arg totals of the later tiles (K1 left them in the phi slots)
  double carry1, carry2;
  tile_carry<true>(partials_b, IAS_GRAD_NS, GS_PHI_1, GS_PHI_2, tile + 1, ntiles, s_carry, tid, carry1, carry2);
  if (FAST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  double tot1 = 0.0, tot2 = 0.0;
#pragma unroll
  for (int q = G16_NQ - 1; q >= 0; --q) {
    float g1[4], g2[4];
    g16_get4<FAST>(blk, pos, q, pg1, j0 + 4 * q, T, g1);
    g16_get4<FAST>(blk + G16_BLK, pos, q, pg2, j0 + 4 * q, T, g2);
#pragma unroll
    for (int x = 3; x >= 0; --x) { tot1 += (double)g1[x]; tot2 += (double)g2[x]; }   // the order of the running sums below
  }
  const double in1 = wave_incl_scan(tot1, lane), in2 = wave_incl_scan(tot2, lane);
  if (lane == 63) { s_w[wave] = in1; s_w[GRAD_WAVES + wave] = in2; }
  __syncthreads();
  double run1 = carry1 + (in1 - tot1), run2 = carry2 + (in2 - tot2);
  for (int w = 0; w < wave; ++w) { run1 += s_w[w]; run2 += s_w[GRAD_WAVES + w]; }

  const int k_first = (int)ias_mul(scale, (float)j0);
  const G16Ctrl cp1 = g16_ctrl(cb, k_first, Tc), cp2 = g16_ctrl(cb + 2 * Tc, k_first, Tc);
  const float kf = (float)(0.6931471805599453 / 12.0);   // d inc / d pitch = inc * ln2 / 12
  // f0_1 depth_1 f0_2 depth_2: fp32 over the thread's samples, fp64 across threads and tiles (as K1's six); the suffix sums
  // themselves stay fp64, their products with the increments are fp32 (the cotangent they give is stored as fp32 anyway)
  float accf[4] = {0.f, 0.f, 0.f, 0.f};
  float ta[2] = {0.f, 0.f}, tw[2] = {0.f, 0.f}, pa[2] = {0.f, 0.f}, pw[2] = {0.f, 0.f};
#pragma unroll 1
  for (int q = G16_SPT / 4 - 1; q >= 0; --q) {              // see K1
    const int jq = j0 + 4 * q;
    float ga1[4], ga2[4], inc1[4], inc2[4];
    g16_get4<FAST>(blk, pos, q, pg1, jq, T, ga1);
    g16_get4<FAST>(blk + G16_BLK, pos, q, pg2, jq, T, ga2);
    g16_get4<FAST>(blk + 2 * G16_BLK, pos, q, pi1, jq, T, inc1);
    g16_get4<FAST>(blk + 3 * G16_BLK, pos, q, pi2, jq, T, inc2);
#pragma unroll
    for (int x = 3; x >= 0; --x) {
      run1 += (double)ga1[x]; run2 += (double)ga2[x];          // inclusive suffix sums
      const float gc1 = inc1[x] > 0.0f ? (float)run1 * (inc1[x] * kf) : 0.0f;     // the sign of the increment is K0's clamp flag
      const float gc2 = inc2[x] > 0.0f ? (float)run2 * (inc2[x] * kf) : 0.0f;
      const float real = ias_mul(scale, (float)(jq + x));
      const int i0 = (int)real;
      const float w1 = ias_sub(real, (float)i0), w0 = ias_sub(1.0f, w1);
      const bool first = i0 == k_first;
      const float pm1 = g16_lerp(cp1, first, w0, w1), pm2 = g16_lerp(cp2, first, w0, w1);
      accf[0] += gc1; accf[1] += gc1 * pm1;
      accf[2] += gc2; accf[3] += gc2 * pm2;
      const float m0 = first ? 1.0f : 0.0f;
      const float xs[2] = {gc1 * vc.depth_1, gc2 * vc.depth_2};
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float xw = w1 * xs[r];
        ta[r] += xs[r]; tw[r] += xw;
        pa[r] = fmaf(m0, xs[r], pa[r]); pw[r] = fmaf(m0, xw, pw[r]);
      }
    }
  }
  const double acc[4] = {(double)accf[0], (double)accf[1], (double)accf[2], (double)accf[3]};
#pragma unroll
  for (int r = 0; r < 2; ++r)
    *g16_fold_slot<true>(stage, r, rt) = (g16_f4){pa[r] - pw[r], pw[r], (ta[r] - pa[r]) - (tw[r] - pw[r]), tw[r] - pw[r]};
  s_kfirst[rt] = j0 < T ? k_first : G16_NOK;
  block_sums<4>(acc, s_out, s_red, tid);
  __syncthreads();
  if (tid == 0) {
    partials_bt[GS_F0_1] = s_out[0]; partials_bt[GS_DEPTH_1] = s_out[1];
    partials_bt[GS_F0_2] = s_out[2]; partials_bt[GS_DEPTH_2] = s_out[3];
  }
  const int rows[2] = {0, 2};
  g16_fold<2, true>(stage, s_kfirst, (int)ias_mul(scale, (float)(tile * GRAD_TILE)), kslots, rows, isum_bt, tid, scale,
                    tile * GRAD_TILE);
}

__global__ __launch_bounds__(GRAD_THREADS, G16_SPT == 8 ? 4 : 2) void voice_grad_pitch16_kernel(
    const float* __restrict__ ctrl, const IasVoiceConst* __restrict__ vconst, float* __restrict__ planes,
    double* __restrict__ partials, int T, int Tc, int ntiles, float scale, int kslots) {
  extern __shared__ __attribute__((aligned(16))) float g16_stage[];
  __shared__ G16Shared sh;
  const int tile = blockIdx.x, b = blockIdx.y;
  const IasVoiceConst vc = vconst[b];
  const float* cb = ctrl + (size_t)b * IAS_NCTRL * Tc;
  const float* pl = planes + (size_t)b * IAS_GRAD_PLANES * T;
  double* isum = g16_tile_sums(planes, b, T, tile, kslots);
  const bool fast = (T & 3) == 0 && (tile + 1) * GRAD_TILE <= T;
  if (fast)
    g16_pitch_tile<true>(sh, g16_stage, cb, vc, pl, partials + (size_t)b * ntiles * IAS_GRAD_NS,
                         partials + ((size_t)b * ntiles + tile) * IAS_GRAD_NS, isum, T, Tc, ntiles, tile, scale, kslots);
  else
    g16_pitch_tile<false>(sh, g16_stage, cb, vc, pl, partials + (size_t)b * ntiles * IAS_GRAD_NS,
                          partials + ((size_t)b * ntiles + tile) * IAS_GRAD_NS, isum, T, Tc, ntiles, tile, scale, kslots);
}

// g_ctrl[i] = A[i] + B[i - 1] (the last control point is its own upper neighbour) from the per-tile interval sums: an
// interval's samples [first(k), first(k + 1)) lie in the tiles first(k) / GRAD_TILE ... (first(k + 1) - 1) / GRAD_TILE,
// added in ascending tile order.
// One thread per control point, all five rows (the interval bounds are found once); the workgroups of the first column
// also add the voice's per-tile partial sums of the 12 constants (fixed order) -> g_scal [B][IAS_GRAD_NS] (NULL: skipped).
__global__ __launch_bounds__(GRAD_THREADS) void voice_grad_fold_combine_kernel(const float* __restrict__ planes,
                                                                               float* __restrict__ g_ctrl,
                                                                               const double* __restrict__ partials,
                                                                               double* __restrict__ g_scal, int T, int Tc,
                                                                               int ntiles, float scale, int kslots) {
  const int i = blockIdx.x * GRAD_THREADS + threadIdx.x, b = blockIdx.y;
  if (g_scal && blockIdx.x == 0) {
    // wave w adds constants w, w + 4, w + 8: lane l takes the tiles l, l + 64, ..., then one wave sum (a fixed tree: the
    // tiles' values all in flight at once -- one thread walking the 44 tiles was 44 dependent L2 round trips)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < IAS_GRAD_NS; k += GRAD_WAVES) {
      const double* p = partials + (size_t)b * ntiles * IAS_GRAD_NS + k;
      double v = 0.0;
      for (int t = lane; t < ntiles; t += 64) v += p[(size_t)t * IAS_GRAD_NS];
      v = wave_total(v);
      if (lane == 0) g_scal[(size_t)b * IAS_GRAD_NS + k] = v;
    }
  }
  if (i >= Tc) return;
  // samples of interval i - 1: [fm, f0), of interval i: [f0, f1)
  const int fm = i > 0 ? ct_first_sample(i - 1, scale, T) : 0, f0 = ct_first_sample(i, scale, T),
            f1 = ct_first_sample(i + 1, scale, T);
  double v[IAS_NCTRL] = {0.0, 0.0, 0.0, 0.0, 0.0};
  auto add = [&](int k, int lo, int hi, int which) {
    if (hi <= lo) return;
    for (int t = lo / GRAD_TILE; t <= (hi - 1) / GRAD_TILE; ++t) {
      const int kk = k - (int)ias_mul(scale, (float)(t * GRAD_TILE));
      if (kk < 0 || kk >= kslots) continue;
      const double* q = g16_tile_sums(planes, b, T, t, kslots) + (size_t)kk * 2 + which;
#pragma unroll
      for (int row = 0; row < IAS_NCTRL; ++row) v[row] += q[(size_t)row * kslots * 2];
    }
  };
  add(i, f0, f1, 0);                          // A[i]
  if (i > 0) add(i - 1, fm, f0, 1);           // B[i - 1]
  if (i == Tc - 1) add(i, f0, f1, 1);         // the last control point is its own upper neighbour
#pragma unroll
  for (int row = 0; row < IAS_NCTRL; ++row) g_ctrl[((size_t)b * IAS_NCTRL + row) * Tc + i] = (float)v[row];
}

// g_scal [B][IAS_GRAD_NS] = the per-tile partial sums added in tile order (first form: its combine kernel does not)
__global__ void voice_grad_scalars_kernel(const double* __restrict__ partials, double* __restrict__ g_scal, int B, int ntiles) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * IAS_GRAD_NS) return;
  const int b = idx / IAS_GRAD_NS, k = idx - b * IAS_GRAD_NS;
  const double* p = partials + (size_t)b * ntiles * IAS_GRAD_NS + k;
  double v = 0.0;
  for (int t = 0; t < ntiles; ++t) v += p[(size_t)t * IAS_GRAD_NS];
  g_scal[idx] = v;
}

// ------------------------------------------------------------------------------------------------ C ABI
// IAS_VOICE_GRAD_V1=1: the chunk-scan kernels + separate transposed upsample for every shape (A/B and the test that compares
// the two forms; read at every call)
static bool voice_grad_force_v1() {
  const char* e = ias_diag_env("IAS_VOICE_GRAD_V1");
  return e && e[0] && e[0] != '0';
}
extern "C" int ias_voice_grad_tiles(int T) { return T > 0 ? (T + GRAD_TILE - 1) / GRAD_TILE : IAS_ERR_ARG; }
extern "C" int ias_voice_grad_nscalars(void) { return IAS_GRAD_NS; }
extern "C" int ias_voice_grad_nplanes(void) { return IAS_GRAD_PLANES; }

// ctrl [B,5,Tc], vconst [B] (64 B each): the outputs of ias_voice_control for the same parameters;
// noise [B,T]; g_mixed [B,T] = d loss / d (un-normalised mix);
// planes [B, ias_voice_grad_nplanes(), T] fp32 scratch; tile_sums [B, ntiles, 2] fp64 scratch;
// partials [B, ntiles, ias_voice_grad_nscalars()] fp64 out (sum over tiles = gradient of the per-voice
// constants in the order f0_1 depth_1 phi_1 f0_2 depth_2 phi_2 kpart shape gain lvl0 lvl1 lvl2);
// g_ctrl [B,5,Tc] fp32 out.  ntiles = ias_voice_grad_tiles(T).
extern "C" int ias_voice_backward_norm(const float* ctrl, const void* vconst, const float* noise, const float* g_mixed,
                                       const float* rownorm, float* planes, double* tile_sums, double* partials,
                                       float* g_ctrl, int B, int T, int Tc, int sample_rate, void* stream_);
extern "C" int ias_voice_backward_sums(const float* ctrl, const void* vconst, const float* noise, const float* g_mixed,
                                       const float* rownorm, float* planes, double* tile_sums, double* partials,
                                       float* g_ctrl, double* g_scal, int B, int T, int Tc, int sample_rate, void* stream_);
extern "C" int ias_voice_backward(const float* ctrl, const void* vconst, const float* noise, const float* g_mixed,
                                  float* planes, double* tile_sums, double* partials, float* g_ctrl, int B, int T,
                                  int Tc, int sample_rate, void* stream_) {
  return ias_voice_backward_norm(ctrl, vconst, noise, g_mixed, nullptr, planes, tile_sums, partials, g_ctrl, B, T, Tc,
                                 sample_rate, stream_);
}

// rownorm [B][4] from the cotangent g [B,T] of the NORMALISED audio [B,T] and the row peaks of the un-normalised mix
// (ias_voice_read_peaks); scratch: ias_voice_norm_scratch_len(T) * B doubles.  Feed rownorm and g to
// ias_voice_backward_norm: the division by the peak and the correction at the peak sample happen as g is read.
extern "C" long long ias_voice_norm_scratch_len(int T) {
  if (T <= 0) return IAS_ERR_ARG;
  return 2LL * ((T + NORM_TILE - 1) / NORM_TILE);
}
extern "C" int ias_voice_norm_backward(const float* g_audio, const float* audio, const float* peaks, double* scratch,
                                       float* rownorm, int B, int T, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!g_audio || !audio || !peaks || !scratch || !rownorm || B <= 0 || B > 65535 || T <= 0) return IAS_ERR_ARG;
  const int ntiles = (T + NORM_TILE - 1) / NORM_TILE;
  hipLaunchKernelGGL(voice_norm_partial_kernel, dim3(ntiles, B), dim3(GRAD_THREADS), 0, stream, g_audio, audio, T,
                     ntiles, scratch);
  hipLaunchKernelGGL(voice_norm_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, scratch, audio, peaks, B, T,
                     ntiles, rownorm);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// the same as ias_voice_backward with the cotangent of the normalised audio and its rownorm (NULL: g_mixed is the
// cotangent of the mix itself)
extern "C" int ias_voice_backward_norm(const float* ctrl, const void* vconst, const float* noise, const float* g_mixed,
                                       const float* rownorm, float* planes, double* tile_sums, double* partials,
                                       float* g_ctrl, int B, int T, int Tc, int sample_rate, void* stream_) {
  return ias_voice_backward_sums(ctrl, vconst, noise, g_mixed, rownorm, planes, tile_sums, partials, g_ctrl, nullptr, B, T, Tc,
                                 sample_rate, stream_);
}
// ... and g_scal [B, ias_voice_grad_nscalars()] fp64 (NULL: not wanted) = partials summed over the tiles, in tile order, by the
// last launch
static int voice_backward_stage(int stage, const float* ctrl, const void* vconst, const float* noise, const float* g_mixed,
                                const float* rownorm, float* planes, double* tile_sums, double* partials, float* g_ctrl,
                                double* g_scal, int B, int T, int Tc, int sample_rate, void* stream_);
extern "C" int ias_voice_backward_sums(const float* ctrl, const void* vconst, const float* noise, const float* g_mixed,
                                       const float* rownorm, float* planes, double* tile_sums, double* partials,
                                       float* g_ctrl, double* g_scal, int B, int T, int Tc, int sample_rate, void* stream_) {
  return voice_backward_stage(-1, ctrl, vconst, noise, g_mixed, rownorm, planes, tile_sums, partials, g_ctrl, g_scal, B, T, Tc,
                              sample_rate, stream_);
}
// The same in two stages on the same buffers.  Stage 0 is the part that does not see the cotangent (the phase increments
// and their tile sums from the control signals: voice_grad_inc_kernel) -- a caller that knows at render time that a
// backward will follow can run it beside the loss computation, on another stream; stage 1 is everything else.
// g_mixed, rownorm, partials, g_ctrl, g_scal may be NULL in stage 0.
extern "C" int ias_voice_backward_sums_stage(int stage, const float* ctrl, const void* vconst, const float* noise,
                                             const float* g_mixed, const float* rownorm, float* planes, double* tile_sums,
                                             double* partials, float* g_ctrl, double* g_scal, int B, int T, int Tc,
                                             int sample_rate, void* stream_) {
  if (stage != 0 && stage != 1) return IAS_ERR_ARG;
  return voice_backward_stage(stage, ctrl, vconst, noise, g_mixed, rownorm, planes, tile_sums, partials, g_ctrl, g_scal, B, T, Tc,
                              sample_rate, stream_);
}
static int voice_backward_stage(int stage, const float* ctrl, const void* vconst, const float* noise, const float* g_mixed,
                                const float* rownorm, float* planes, double* tile_sums, double* partials, float* g_ctrl,
                                double* g_scal, int B, int T, int Tc, int sample_rate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!ctrl || !vconst || !planes || !tile_sums) return IAS_ERR_ARG;
  if (stage != 0 && (!noise || !g_mixed || !partials || !g_ctrl)) return IAS_ERR_ARG;
  if (B <= 0 || B > 65535 || T <= 1 || Tc <= 1 || sample_rate <= 0) return IAS_ERR_ARG;
  const int ntiles = (T + GRAD_TILE - 1) / GRAD_TILE;
  if (ntiles > 65535) return IAS_ERR_UNSUPPORTED;
  const float scale = (float)(Tc - 1) / (float)(T - 1);
  // lane-consecutive form: >= G16_SPT samples per control interval and the per-tile interval sums fit the spare plane
  const int kslots = (int)(scale * (float)GRAD_TILE) + 3;
  const bool fold = scale * (float)G16_SPT <= 1.0f && !voice_grad_force_v1() &&
                    (size_t)ntiles * IAS_NCTRL * kslots * 2 * sizeof(double) + 8 <= (size_t)T * sizeof(float);
  // first form: K3 stages nint + 1 control intervals of samples in LDS (as many as fit, at most CT_INTERVALS) and keeps
  // its interval sums in the spare plane
  const int nint = min(CT_INTERVALS, (int)((double)CT_CAP / ((double)(T - 1) / (double)(Tc - 1) + 2.0)) - 1);
  if (!fold && (nint < 1 || (size_t)IAS_NCTRL * Tc * 2 * sizeof(double) + 8 > (size_t)T * sizeof(float)))
    return IAS_ERR_UNSUPPORTED;
  const IasVoiceConst* vc = (const IasVoiceConst*)vconst;
  const dim3 grid(ntiles, B), block(GRAD_THREADS);
  if (stage <= 0)
    hipLaunchKernelGGL(voice_grad_inc_kernel, grid, block, 0, stream, ctrl, vc, planes, tile_sums, T, Tc, ntiles,
                       1.0 / (double)sample_rate, scale);
  if (stage == 0) return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  if (fold) {
    static const bool lds_ok = [] {      // 64 KB of staging blocks + the static part: above the default dynamic limit
      const bool a = hipFuncSetAttribute((const void*)voice_grad_sample16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         G16_LDS_BYTES) == hipSuccess;
      const bool c = hipFuncSetAttribute((const void*)voice_grad_pitch16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         G16_LDS_BYTES) == hipSuccess;
      return a && c;
    }();
    if (!lds_ok) return IAS_ERR_LAUNCH;
    hipLaunchKernelGGL(voice_grad_sample16_kernel, grid, block, G16_LDS_BYTES, stream, ctrl, vc, noise, g_mixed, planes,
                       tile_sums, partials, T, Tc, ntiles, scale, rownorm, kslots);
    hipLaunchKernelGGL(voice_grad_pitch16_kernel, grid, block, G16_LDS_BYTES, stream, ctrl, vc, planes, partials, T, Tc,
                       ntiles, scale, kslots);
    hipLaunchKernelGGL(voice_grad_fold_combine_kernel, dim3((Tc + GRAD_THREADS - 1) / GRAD_THREADS, B), block, 0, stream,
                       planes, g_ctrl, partials, g_scal, T, Tc, ntiles, scale, kslots);
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(voice_grad_sample_kernel, grid, block, 0, stream, ctrl, vc, noise, g_mixed, planes, tile_sums,
                     partials, T, Tc, ntiles, scale, rownorm);
  hipLaunchKernelGGL(voice_grad_pitch_kernel, grid, block, 0, stream, ctrl, vc, planes, partials, T, Tc, ntiles,
                     scale);
  hipLaunchKernelGGL(voice_grad_ctrl_kernel, dim3((Tc + nint - 1) / nint, IAS_NCTRL, B), block, 0,
                     stream, planes, T, Tc, scale, nint);
  hipLaunchKernelGGL(voice_grad_ctrl_combine_kernel, dim3((Tc + GRAD_THREADS - 1) / GRAD_THREADS, IAS_NCTRL, B), block,
                     0, stream, planes, g_ctrl, T, Tc);
  if (g_scal)
    hipLaunchKernelGGL(voice_grad_scalars_kernel, dim3((B * IAS_GRAD_NS + 255) / 256), dim3(256), 0, stream, partials, g_scal,
                       B, ntiles);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
