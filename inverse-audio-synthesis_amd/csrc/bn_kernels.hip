// Training-mode BatchNorm2d with the following activation fused in (MI355X / gfx950), NCHW fp32.
//
// Replaces nn.BatchNorm2d + nn.ReLU / nn.Hardswish of torchvision's MobileNetV3 blocks as AudioEmbedding runs them
// (/root/reference/audioembed.py:61 -> vision_model.features, /root/reference/vicreg_audio_params.py:52-54) and their
// autograd.  On MIOpen the two large-map layers alone took 0.8 ms forward and 0.77 ms backward each at batch 128
// ([128,16,120,123]); the activation and its backward were separate elementwise passes.  Here:
//   forward : per-channel shifted sums (partials), then ONE pass that finalizes them per workgroup (fixed order; running
//             statistics updated as torch does) and writes y = act(w (x - mean) invstd + b);
//   backward: ONE reduction pass over (x, dy) that recomputes the pre-activation value z, applies act'(z) and sums
//             dz and dz * xhat per channel, a finalize (dw, db), and ONE pass dx = w invstd (dz - mean(dz) - xhat mean(dz xhat)).
// Nothing but mean / invstd is saved between forward and backward.  All sums in a fixed order (deterministic).
#include "ias_common.h"
#include "wave_ops.h"
#include <cstdint>
#include <cstdlib>

#define BN_THREADS 256
#define BN_UNROLL 4      // bn_partials_kernel: loads requested before the first is consumed
#define BN_ACT_NONE 0
#define BN_ACT_RELU 1
#define BN_ACT_HARDSWISH 2

__device__ __forceinline__ float bn_act(float z, int act) {
  if (act == BN_ACT_RELU) return fmaxf(z, 0.0f);
  if (act == BN_ACT_HARDSWISH) return z * fminf(fmaxf(z + 3.0f, 0.0f), 6.0f) / 6.0f;
  return z;
}
// d act / dz with torch's conventions at the kinks (threshold_backward: z > 0; hardswish_backward: z < -3 -> 0,
// z <= 3 -> z / 3 + 0.5, else 1)
__device__ __forceinline__ float bn_act_grad(float z, int act) {
  if (act == BN_ACT_RELU) return z > 0.0f ? 1.0f : 0.0f;
  if (act == BN_ACT_HARDSWISH) return z < -3.0f ? 0.0f : (z <= 3.0f ? z / 3.0f + 0.5f : 1.0f);
  return 1.0f;
}

// sum of a, b over the workgroup, fixed order: DPP scan inside each wave (wave_ops.h), the four wave totals through LDS --
// one barrier instead of the eight of a shared-memory tree on doubles
__device__ __forceinline__ void bn_block_reduce2(double& a, double& b) {
  __shared__ double s_a[BN_THREADS / 64], s_b[BN_THREADS / 64];
  a = wave_sum(a); b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_b[threadIdx.x >> 6] = b; }
  __syncthreads();
  a = 0.0; b = 0.0;
#pragma unroll
  for (int w = 0; w < BN_THREADS / 64; ++w) { a += s_a[w]; b += s_b[w]; }
}

// Workgroup (c, s) sums over the batches b = s, s + S, ... of channel c.  MODE 0: sum (x - K), sum (x - K)^2 with the
// shift K = x[0, c, 0] (no cancellation for channels with a large mean).  MODE 1: sum dz, sum dz xhat.
//
// A plane is HW contiguous floats at an arbitrary 4-byte phase ([*, C, 30, 31] maps: 930 floats, every other plane starts
// 8 bytes into a 16-byte group).  Rounds 2-5 read such planes with 4-byte loads, one dependent load per lane and plane:
// 2.3-3.0 TB/s on the six 30 x 31 layers of the trunk (profiles/r05f_trace_pretrain_list.txt: 75 us forward, 128 us
// backward per pretraining step).  Now every plane is `head` (< 4) scalars up to the next 16-byte boundary, `body4`
// 16-byte vectors and a tail of < 4 scalars; a lane's work item is one vector or one scalar, the items of the workgroup's
// planes form ONE sequence, and BN_UNROLL of them are requested before the first is consumed (the loads of several
// planes -- or of several vectors of a large plane -- in flight per lane instead of one).  Sums in a fixed order.
struct BnItem { float4 a, g; int n; };     // n: valid elements (4: a vector, 1: a scalar in .x, 0: nothing)
template <int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_partials_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ weight,
                                                                  const float* __restrict__ bias, double* __restrict__ partials,
                                                                  int B, int C, int HW, int act) {
  const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y, tid = threadIdx.x;
  float p0 = 0.0f, p1 = 0.0f;
  float k = 0.0f, m = 0.0f, is = 1.0f, w = 1.0f, bb = 0.0f;
  if (MODE == 0) k = x[(size_t)c * HW];
  else { m = mean[c]; is = invstd[c]; w = weight ? weight[c] : 1.0f; bb = bias ? bias[c] : 0.0f; }
  auto one = [&](float xv, float gv) {
    if (MODE == 0) { const float d = xv - k; p0 += d; p1 = fmaf(d, d, p1); }
    else {
      const float xh = (xv - m) * is;
      const float dz = gv * bn_act_grad(fmaf(w, xh, bb), act);
      p0 += dz; p1 = fmaf(dz, xh, p1);
    }
  };
  // x and dy share the phase of every plane when both arrays start on a 16-byte boundary (torch allocations do)
  const bool a16 = (((uintptr_t)x | (uintptr_t)dy) & 15) == 0;
  if (a16 && (HW & 3) == 0) {
    // planes of whole, aligned vectors: a plain strided loop per plane (the compiler keeps several of its loads in flight;
    // on the [*, 16, 120, 123] and [*, 72, 60, 62] maps the item sequence below measured 30 % SLOWER than this loop,
    // gpurun_out/trace_ptl of the round's third session).  Small planes share the workgroup: lpp lanes per plane (a
    // power of two >= the plane's vector count), BN_THREADS / lpp planes per pass
    const int nv = HW >> 2;
    int lpp = BN_THREADS;
    while (lpp > 16 && (lpp >> 1) >= nv) lpp >>= 1;
    const int ppi = BN_THREADS / lpp, pl = tid / lpp, li = tid - pl * lpp;
    for (int b = s + pl * S; b < B; b += S * ppi) {
      const size_t base = ((size_t)b * C + c) * HW;
      const float4* x4 = reinterpret_cast<const float4*>(x + base);
      const float4* g4 = MODE == 1 ? reinterpret_cast<const float4*>(dy + base) : nullptr;
      int i = li;
      // large planes, forward: BN_UNROLL vectors requested at once ([*, 16, 120, 123]: 26.8 -> 21.6 us; on planes of 930
      // vectors most lanes fall through to the remainder loop -- 64.7 -> 69.9 us backward -- and (x, dy) pairs gain nothing)
      if (MODE == 0 && nv >= 8 * lpp)
      for (; i + (BN_UNROLL - 1) * lpp < nv; i += BN_UNROLL * lpp) {
        float4 a[BN_UNROLL], g[BN_UNROLL];
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) {
          a[u] = x4[i + u * lpp];
          g[u] = float4{0.f, 0.f, 0.f, 0.f};
          if (MODE == 1) g[u] = g4[i + u * lpp];
        }
#pragma unroll
        for (int u = 0; u < BN_UNROLL; ++u) { one(a[u].x, g[u].x); one(a[u].y, g[u].y); one(a[u].z, g[u].z); one(a[u].w, g[u].w); }
      }
      for (; i < nv; i += lpp) {
        const float4 a = x4[i];
        float4 g = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 1) g = g4[i];
        one(a.x, g.x); one(a.y, g.y); one(a.z, g.z); one(a.w, g.w);
      }
    }
  } else if (a16) {
    // items per plane: its vectors and, behind them, up to 6 scalar slots; small planes share the workgroup: lpp lanes per
    // plane (a power of two >= the item count, at most the workgroup), BN_THREADS / lpp planes side by side
    const int nitems = (HW >> 2) + 6;
    int lpp = BN_THREADS;
    while (lpp > 16 && (lpp >> 1) >= nitems) lpp >>= 1;
    const int ppi = BN_THREADS / lpp, pl = tid / lpp, li = tid - pl * lpp;
    const int ipl = (nitems + lpp - 1) / lpp;                       // items per lane and plane
    const int nplanes = (B - s + S - 1) / S;                        // planes of this workgroup: b = s + j S
    const int jn = pl < nplanes ? (nplanes - pl + ppi - 1) / ppi : 0;   // ... of this lane group: j = pl + jj ppi
    const int total = jn * ipl;
    auto fetch = [&](int mi, BnItem& it) {
      it.n = 0;
      it.a = float4{0.f, 0.f, 0.f, 0.f}; it.g = float4{0.f, 0.f, 0.f, 0.f};
      if (mi >= total) return;
      const int jj = mi / ipl, i = li + (mi - jj * ipl) * lpp;
      const int b = s + (pl + jj * ppi) * S;
      const size_t base = ((size_t)b * C + c) * HW;
      int head = (4 - (int)(base & 3)) & 3;
      if (head > HW) head = HW;
      const int body4 = (HW - head) >> 2;
      if (i < body4) {
        it.a = *reinterpret_cast<const float4*>(x + base + head + 4 * (size_t)i);
        if (MODE == 1) it.g = *reinterpret_cast<const float4*>(dy + base + head + 4 * (size_t)i);
        it.n = 4;
      } else {
        const int e = i - body4, nscal = HW - 4 * body4;             // head + tail scalars
        if (e < nscal) {
          const size_t idx = base + (e < head ? e : 4 * body4 + e);
          it.a.x = x[idx];
          if (MODE == 1) it.g.x = dy[idx];
          it.n = 1;
        }
      }
    };
    for (int m0 = 0; m0 < total; m0 += BN_UNROLL) {
      BnItem it[BN_UNROLL];
#pragma unroll
      for (int u = 0; u < BN_UNROLL; ++u) fetch(m0 + u, it[u]);
#pragma unroll
      for (int u = 0; u < BN_UNROLL; ++u) {
        if (it[u].n > 0) one(it[u].a.x, it[u].g.x);
        if (it[u].n == 4) { one(it[u].a.y, it[u].g.y); one(it[u].a.z, it[u].g.z); one(it[u].a.w, it[u].g.w); }
      }
    }
  } else {
    for (int b = s; b < B; b += S) {
      const size_t base = ((size_t)b * C + c) * HW;
      for (int i = tid; i < HW; i += BN_THREADS) one(x[base + i], MODE == 1 ? dy[base + i] : 0.0f);
    }
  }
  double d0 = (double)p0, d1 = (double)p1;
  bn_block_reduce2(d0, d1);
  if (tid == 0) { partials[((size_t)c * S + s) * 2] = d0; partials[((size_t)c * S + s) * 2 + 1] = d1; }
}

// mean, biased variance -> save_mean, save_invstd, running statistics (momentum m: r = (1 - m) r + m stat, unbiased
// variance in the running one, as torch.nn.BatchNorm2d)
// The finalize launches sit between the partial-sum pass and the apply pass of every large-map layer, 20 per pretraining
// step, and are pure latency.  Rounds 2-4: a thread per channel walking its S <= 64 partial pairs one dependent 16-byte
// load after the other, ONE workgroup for the layer (5-10 us per launch).  Round 5: a WAVE per channel -- lane s loads
// pair s (one request per lane, all in flight together) and the pairs meet in the DPP scan of wave_ops.h (fixed order:
// deterministic, fp64 as before).
#define BN_FIN_WAVES 4
__global__ __launch_bounds__(64 * BN_FIN_WAVES) void bn_finalize_stats_kernel(
    const float* __restrict__ x, const double* __restrict__ partials, float* __restrict__ save_mean,
    float* __restrict__ save_invstd, float* __restrict__ running_mean, float* __restrict__ running_var, int C, int S, int HW,
    double n, float eps, float momentum) {
  const int lane = threadIdx.x & 63, c = blockIdx.x * BN_FIN_WAVES + (threadIdx.x >> 6);
  if (c >= C) return;                                  // (whole waves leave: c is wave-uniform)
  double s1 = 0.0, s2 = 0.0;
  if (lane < S) {
    const double2 v = reinterpret_cast<const double2*>(partials)[(size_t)c * S + lane];
    s1 = v.x; s2 = v.y;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane != 0) return;
  const double k = (double)x[(size_t)c * HW];
  const double dm = s1 / n;
  double var = s2 / n - dm * dm;
  if (var < 0.0) var = 0.0;
  const double mean = k + dm;
  save_mean[c] = (float)mean;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
  if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (n > 1.0 ? n / (n - 1.0) : 1.0));
}

__global__ __launch_bounds__(64 * BN_FIN_WAVES) void bn_finalize_grads_kernel(const double* __restrict__ partials,
                                                                              float* __restrict__ gw, float* __restrict__ gb,
                                                                              float* __restrict__ sums, int C, int S) {
  const int lane = threadIdx.x & 63, c = blockIdx.x * BN_FIN_WAVES + (threadIdx.x >> 6);
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  if (lane < S) {
    const double2 v = reinterpret_cast<const double2*>(partials)[(size_t)c * S + lane];
    s1 = v.x; s2 = v.y;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane != 0) return;
  if (gb) gb[c] = (float)s1;
  if (gw) gw[c] = (float)s2;
  sums[2 * c] = (float)s1; sums[2 * c + 1] = (float)s2;
}

// One elementwise pass over planes (b, c): MODE 0 y = act(w xhat + b); MODE 1 dx = w invstd (dz - sum_dz / n - xhat sum_dzxh / n)
template <int MODE, int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ weight, const float* __restrict__ bias,
                                                               const float* __restrict__ sums, float* __restrict__ out, int C,
                                                               int HW, long long total_vec, float inv_n, int act) {
  typedef float vt __attribute__((ext_vector_type(VEC)));
  const int hwv = HW / VEC;
  for (long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x; i < total_vec; i += (long long)gridDim.x * BN_THREADS) {
    const int c = (int)((i / hwv) % C);
    const float m = mean[c], is = invstd[c], w = weight ? weight[c] : 1.0f, bb = bias ? bias[c] : 0.0f;
    const vt xv = reinterpret_cast<const vt*>(x)[i];
    vt o;
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = bn_act(fmaf(w, (xv[e] - m) * is, bb), act);
      if (dy != nullptr) {                                   // forward: `dy` carries the residual (or null)
        const vt rv = reinterpret_cast<const vt*>(dy)[i];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] += rv[e];
      }
    } else {
      const vt gv = reinterpret_cast<const vt*>(dy)[i];
      const float a = sums[2 * c] * inv_n, b2 = sums[2 * c + 1] * inv_n, ws = w * is;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float xh = (xv[e] - m) * is;
        const float dz = gv[e] * bn_act_grad(fmaf(w, xh, bb), act);
        o[e] = ws * (dz - a - xh * b2);
      }
    }
    reinterpret_cast<vt*>(out)[i] = o;
  }
}
// (float1 as a 1-wide ext vector is not subscriptable on every compiler version: scalar specialisation)
template <int MODE>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_scalar_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                      const float* __restrict__ weight, const float* __restrict__ bias,
                                                                      const float* __restrict__ sums, float* __restrict__ out, int C,
                                                                      int HW, long long total, float inv_n, int act) {
  for (long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * BN_THREADS) {
    const int c = (int)((i / HW) % C);
    const float m = mean[c], is = invstd[c], w = weight ? weight[c] : 1.0f, bb = bias ? bias[c] : 0.0f;
    const float xh = (x[i] - m) * is;
    if (MODE == 0) out[i] = bn_act(fmaf(w, xh, bb), act) + (dy != nullptr ? dy[i] : 0.0f);
    else {
      const float dz = dy[i] * bn_act_grad(fmaf(w, xh, bb), act);
      out[i] = w * is * (dz - sums[2 * c] * inv_n - xh * sums[2 * c + 1] * inv_n);
    }
  }
}

// The pass behind the partial sums of a layer below BN_FINAPPLY_MAX_BYTES: finalize + apply in ONE launch.
// The finalize launch above sits between the partial-sum pass and the elementwise pass (mean / invstd / running statistics
// forward, dw / db / the two means backward): ~5 us of pure latency on the step's dependency chain, 20 of them per
// pretraining step.  Here every workgroup of the elementwise pass belongs to ONE
// channel (grid: channel x group of samples x segment of the plane) and its first wave finalizes that channel itself:
// lane s loads partial pair s (S <= 64: one request per lane, L2 hits), the pairs meet in the DPP scan of wave_ops.h, lane 0
// does the fp64 arithmetic the finalize kernels did -- the same operations in the same order, so every value is
// bit-identical to the three-launch form -- and the workgroup of (sample group 0, segment 0) writes save_mean / save_invstd
// / the running statistics (forward: momentum m: r = (1 - m) r + m stat, unbiased variance in the running one, as
// torch.nn.BatchNorm2d) or dw / db (backward).  Nothing is atomic and nothing polls: the partial sums are complete when
// this kernel starts.  Measured per layer against finalize + flat apply (profiles/r05g_trace_pretrain_list.txt and the
// round's earlier lists): 4-6 us less on the 30 x 31 and [*, 16, 60, 62] maps in either direction; on the two 120+ MB maps
// ([*, 16, 120, 123], [*, 72, 60, 62]) a workgroup's share is long, the launch is 1.1-2.3 rounds of resident workgroups
// and its last round runs latency-bound (53 against 44 us forward, 76 against 68 backward): those keep the three launches.
//   MODE 0: y = act(w xhat + b) (+ res: `dy` carries the residual or null);
//   MODE 1: dx = w invstd (dz - mean(dz) - xhat mean(dz xhat)).
template <int MODE, int VEC>
__global__ __launch_bounds__(BN_THREADS) void bn_finapply_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const double* __restrict__ partials,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ running_mean,
    float* __restrict__ running_var, const float* __restrict__ weight, const float* __restrict__ bias, float* __restrict__ gw,
    float* __restrict__ gb, float* __restrict__ out, int B, int C, int HW, int S, int PB, int segv, double n, float eps,
    float momentum, float inv_n, int act) {
  typedef float vt __attribute__((ext_vector_type(VEC)));
  __shared__ float s_stat[4];
  const int c = blockIdx.x, tid = threadIdx.x;
  const bool writer = blockIdx.y == 0 && blockIdx.z == 0;
  if (tid < 64) {
    double s1 = 0.0, s2 = 0.0;
    if (tid < S) {
      const double2 v = reinterpret_cast<const double2*>(partials)[(size_t)c * S + tid];
      s1 = v.x; s2 = v.y;
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (tid == 0) {
      if (MODE == 0) {
        const double k = (double)x[(size_t)c * HW];
        const double dm = s1 / n;
        double var = s2 / n - dm * dm;
        if (var < 0.0) var = 0.0;
        const double mean = k + dm;
        const float mf = (float)mean, isf = (float)(1.0 / sqrt(var + (double)eps));
        s_stat[0] = mf; s_stat[1] = isf;
        if (writer) {
          save_mean[c] = mf;
          save_invstd[c] = isf;
          if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
          if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (n > 1.0 ? n / (n - 1.0) : 1.0));
        }
      } else {
        s_stat[0] = save_mean[c]; s_stat[1] = save_invstd[c];
        s_stat[2] = (float)s1; s_stat[3] = (float)s2;
        if (writer) {
          if (gb) gb[c] = (float)s1;
          if (gw) gw[c] = (float)s2;
        }
      }
    }
  }
  __syncthreads();
  const float m = s_stat[0], is = s_stat[1], w = weight ? weight[c] : 1.0f, bb = bias ? bias[c] : 0.0f;
  const float a = MODE == 1 ? s_stat[2] * inv_n : 0.0f, b2 = MODE == 1 ? s_stat[3] * inv_n : 0.0f, ws = w * is;
  const int hwv = HW / VEC;
  const int v0 = blockIdx.z * segv, v1 = min(hwv, v0 + segv);
  const int b_begin = blockIdx.y * PB, b_end = min(B, b_begin + PB);
  for (int b = b_begin; b < b_end; ++b) {
    const size_t base = ((size_t)b * C + c) * HW;
    const vt* xp = reinterpret_cast<const vt*>(x + base);
    const vt* gp = dy != nullptr ? reinterpret_cast<const vt*>(dy + base) : nullptr;
    vt* op = reinterpret_cast<vt*>(out + base);
#pragma unroll 4
    for (int i = v0 + tid; i < v1; i += BN_THREADS) {
      const vt xv = xp[i];
      vt o;
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = bn_act(fmaf(w, (xv[e] - m) * is, bb), act);
        if (gp != nullptr) {
          const vt rv = gp[i];
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] += rv[e];
        }
      } else {
        const vt gv = gp[i];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float xh = (xv[e] - m) * is;
          const float dz = gv[e] * bn_act_grad(fmaf(w, xh, bb), act);
          o[e] = ws * (dz - a - xh * b2);
        }
      }
      op[i] = o;
    }
  }
}

// ---- small maps (round 4): ONE workgroup per channel, ONE launch per direction ----------------------------------------
// 24 of the trunk's 34 BatchNorm layers sit on 15 x 16 and 8 x 8 maps: a channel is 128 x 240 (or 64) values, 120 KB at
// most.  The three-launch form above spends 26 us forward and 35 us backward on such a layer, nearly all of it launch and
// latency (partials, a finalize launch over C numbers, apply).  Here a 1024-thread workgroup keeps its channel in
// registers (up to NV 16-byte vectors per thread): statistics, finalize and apply (forward), or dz / xhat sums, finalize
// and dx (backward) in one pass over HBM -- x is read once instead of twice, (x, dy) once instead of twice.
// Sums: per thread in fp32 over its <= 4 NV values (shifted by K = x[0, c, 0] in the forward), then in fp64 across the
// wave (xor butterfly: the same pairing for every lane count) and across the 16 waves in wave order: fixed order.
#define BNF_THREADS 1024
__device__ __forceinline__ void bnf_block_sum2(double& a, double& b, double (*s_red)[2]) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();                       // (s_red may still be read from a previous use)
  if (lane == 0) { s_red[wave][0] = a; s_red[wave][1] = b; }
  __syncthreads();
  double ta = 0.0, tb = 0.0;
  for (int w = 0; w < BNF_THREADS / 64; ++w) { ta += s_red[w][0]; tb += s_red[w][1]; }
  a = ta; b = tb;
}
// vector i (of B * HW / 4) of channel c: batch i / hwv, vector i % hwv of the plane
__device__ __forceinline__ size_t bnf_off(int i, int hwv, int C, int c) {
  const int b = i / hwv, j = i - b * hwv;
  return ((size_t)b * C + c) * hwv + j;
}

// POOL: also pooled[b][c] = mean over the plane of y[b][c] -- the average pool of the squeeze-excitation block behind a
// depthwise convolution's normalisation (torchvision SqueezeExcitation._scale: `self.avgpool(input)`), whose own launch
// (ias_se_plane_reduce, ~5 us of latency, nine per pretraining step) re-read the map this workgroup holds in registers: a
// thread leaves the sum of each of its vectors in LDS, thread b adds the hwv sums of plane b in order (fixed order).
template <int NV, bool POOL>
__global__ __launch_bounds__(BNF_THREADS) void bn_fused_forward_kernel(
    const float4* __restrict__ x, const float* __restrict__ weight, const float* __restrict__ bias,
    float* __restrict__ running_mean, float* __restrict__ running_var, float4* __restrict__ y, float* __restrict__ save_mean,
    float* __restrict__ save_invstd, int B, int C, int hwv, float eps, float momentum, int act,
    const float4* __restrict__ res /* or null: y = act(bn(x)) + res (the block's residual connection) */,
    float* __restrict__ pooled /* POOL: [B][C] */) {
  __shared__ double s_red[BNF_THREADS / 64][2];
  __shared__ float s_pool[POOL ? NV * BNF_THREADS : 1];
  const int c = blockIdx.x, tid = threadIdx.x, nvec = B * hwv;
  const float k = reinterpret_cast<const float*>(x)[(size_t)c * hwv * 4];
  float4 v[NV];
  float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int i = tid + e * BNF_THREADS;
    if (i < nvec) {
      v[e] = x[bnf_off(i, hwv, C, c)];
      const float d0 = v[e].x - k, d1 = v[e].y - k, d2 = v[e].z - k, d3 = v[e].w - k;
      p0 += d0; p0 += d1; p0 += d2; p0 += d3;
      p1 = fmaf(d0, d0, p1); p1 = fmaf(d1, d1, p1); p1 = fmaf(d2, d2, p1); p1 = fmaf(d3, d3, p1);
    }
  }
  double s1 = (double)p0, s2 = (double)p1;
  bnf_block_sum2(s1, s2, s_red);
  const double n = (double)nvec * 4.0, dm = s1 / n;
  double var = s2 / n - dm * dm;
  if (var < 0.0) var = 0.0;
  const double mean_d = (double)k + dm;
  const float mean = (float)mean_d, invstd = (float)(1.0 / sqrt(var + (double)eps));
  if (tid == 0) {
    save_mean[c] = mean; save_invstd[c] = invstd;
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean_d);
    if (running_var) running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var * (n > 1.0 ? n / (n - 1.0) : 1.0));
  }
  const float w = weight ? weight[c] : 1.0f, bb = bias ? bias[c] : 0.0f;
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int i = tid + e * BNF_THREADS;
    if (i < nvec) {
      float4 o;
      o.x = bn_act(fmaf(w, (v[e].x - mean) * invstd, bb), act); o.y = bn_act(fmaf(w, (v[e].y - mean) * invstd, bb), act);
      o.z = bn_act(fmaf(w, (v[e].z - mean) * invstd, bb), act); o.w = bn_act(fmaf(w, (v[e].w - mean) * invstd, bb), act);
      if (res != nullptr) { const float4 r = res[bnf_off(i, hwv, C, c)]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
      y[bnf_off(i, hwv, C, c)] = o;
      if (POOL) s_pool[i] = (o.x + o.y) + (o.z + o.w);
    }
  }
  if (POOL) {
    __syncthreads();
    const float inv_hw = 1.0f / (float)(4 * hwv);
    for (int b = tid; b < B; b += BNF_THREADS) {
      float t = 0.0f;
      for (int j = 0; j < hwv; ++j) t += s_pool[b * hwv + j];
      pooled[(size_t)b * C + c] = t * inv_hw;
    }
  }
}

template <int NV>
__global__ __launch_bounds__(BNF_THREADS) void bn_fused_backward_kernel(
    const float4* __restrict__ x, const float4* __restrict__ dy, const float* __restrict__ weight,
    const float* __restrict__ bias, const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
    float4* __restrict__ dx, float* __restrict__ gw, float* __restrict__ gb, int B, int C, int hwv, int act) {
  __shared__ double s_red[BNF_THREADS / 64][2];
  const int c = blockIdx.x, tid = threadIdx.x, nvec = B * hwv;
  const float m = save_mean[c], is = save_invstd[c], w = weight ? weight[c] : 1.0f, bb = bias ? bias[c] : 0.0f;
  float4 xh[NV], dz[NV];
  float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int i = tid + e * BNF_THREADS;
    if (i < nvec) {
      const size_t o = bnf_off(i, hwv, C, c);
      const float4 xv = x[o], gv = dy[o];
      xh[e].x = (xv.x - m) * is; xh[e].y = (xv.y - m) * is; xh[e].z = (xv.z - m) * is; xh[e].w = (xv.w - m) * is;
      dz[e].x = gv.x * bn_act_grad(fmaf(w, xh[e].x, bb), act); dz[e].y = gv.y * bn_act_grad(fmaf(w, xh[e].y, bb), act);
      dz[e].z = gv.z * bn_act_grad(fmaf(w, xh[e].z, bb), act); dz[e].w = gv.w * bn_act_grad(fmaf(w, xh[e].w, bb), act);
      p0 += dz[e].x; p0 += dz[e].y; p0 += dz[e].z; p0 += dz[e].w;
      p1 = fmaf(dz[e].x, xh[e].x, p1); p1 = fmaf(dz[e].y, xh[e].y, p1); p1 = fmaf(dz[e].z, xh[e].z, p1); p1 = fmaf(dz[e].w, xh[e].w, p1);
    }
  }
  double s1 = (double)p0, s2 = (double)p1;
  bnf_block_sum2(s1, s2, s_red);
  if (tid == 0) { if (gb) gb[c] = (float)s1; if (gw) gw[c] = (float)s2; }
  const float inv_n = 1.0f / (float)((long long)nvec * 4);
  const float a = (float)s1 * inv_n, b2 = (float)s2 * inv_n, ws = w * is;
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int i = tid + e * BNF_THREADS;
    if (i < nvec) {
      float4 o;
      o.x = ws * (dz[e].x - a - xh[e].x * b2); o.y = ws * (dz[e].y - a - xh[e].y * b2);
      o.z = ws * (dz[e].z - a - xh[e].z * b2); o.w = ws * (dz[e].w - a - xh[e].w * b2);
      dx[bnf_off(i, hwv, C, c)] = o;
    }
  }
}
// vectors per thread the one-workgroup form needs for this layer, or 0 if it does not apply (IAS_BN_UNFUSED=1: never)
static int bn_fused_nv(const void* p0, const void* p1, const void* p2, int B, int C, int HW) {
  static const bool off = ias_diag_env("IAS_BN_UNFUSED") && atoi(ias_diag_env("IAS_BN_UNFUSED")) != 0;
  if (off || (HW & 3) || ((((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15) != 0) || C < 32) return 0;
  const long long nvec = (long long)B * (HW >> 2);
  if (nvec <= 2 * BNF_THREADS) return 2;
  if (nvec <= 8 * BNF_THREADS) return 8;
  return 0;
}

// ------------------------------------------------------------------------ C ABI
static int bn_split(int B, int C) {
  static const int target = ias_diag_env("IAS_BN_WGS") ? atoi(ias_diag_env("IAS_BN_WGS")) : 2048;   // (diagnostics knob)
  int s = (target + C - 1) / C;
  if (s < 1) s = 1;
  if (s > B) s = B;
  if (s > 64) s = 64;     // (bn_finalize_*, bn_finapply_kernel: one partial pair per lane of a wave)
  return s;
}
// doubles of scratch the forward / backward need (partials [C][split][2])
extern "C" long long ias_bn_scratch_doubles(int B, int C) {
  if (B <= 0 || C <= 0) return IAS_ERR_ARG;
  return (long long)C * bn_split(B, C) * 2;
}

template <int MODE>
static void bn_launch_apply(hipStream_t stream, const float* x, const float* dy, const float* mean, const float* invstd,
                            const float* weight, const float* bias, const float* sums, float* out, int B, int C, int HW,
                            int act) {
  const long long total = (long long)B * C * HW;
  const float inv_n = 1.0f / (float)((long long)B * HW);
  const bool a16 = ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy) & 15) == 0);
  const int vec = (a16 && (HW & 3) == 0) ? 4 : ((a16 && (HW & 1) == 0) ? 2 : 1);
  const long long tv = total / vec;
  long long blocks = (tv + BN_THREADS - 1) / BN_THREADS;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (vec == 4)
    hipLaunchKernelGGL((bn_apply_kernel<MODE, 4>), dim3((unsigned)blocks), dim3(BN_THREADS), 0, stream, x, dy, mean, invstd,
                       weight, bias, sums, out, C, HW, tv, inv_n, act);
  else if (vec == 2)
    hipLaunchKernelGGL((bn_apply_kernel<MODE, 2>), dim3((unsigned)blocks), dim3(BN_THREADS), 0, stream, x, dy, mean, invstd,
                       weight, bias, sums, out, C, HW, tv, inv_n, act);
  else
    hipLaunchKernelGGL((bn_apply_scalar_kernel<MODE>), dim3((unsigned)blocks), dim3(BN_THREADS), 0, stream, x, dy, mean,
                       invstd, weight, bias, sums, out, C, HW, tv, inv_n, act);
}

// layers whose map is below 64 MB take the two-launch form (see bn_finapply_kernel)
#define BN_FINAPPLY_MAX_BYTES (64ll << 20)
static bool bn_finapply_takes(int B, int C, int HW) { return (long long)B * C * HW * 4 < BN_FINAPPLY_MAX_BYTES; }

// finalize + apply (bn_finapply_kernel): grid (C, groups of PB samples, segments of segv vectors).  A workgroup should
// see >= ~4 vectors per thread behind its finalize prologue, and the launch ~4096 workgroups.
template <int MODE>
static void bn_launch_finapply(hipStream_t stream, const float* x, const float* dy, const double* partials, float* save_mean,
                               float* save_invstd, float* running_mean, float* running_var, const float* weight,
                               const float* bias, float* gw, float* gb, float* out, int B, int C, int HW, int S, float eps,
                               float momentum, int act) {
  const float inv_n = 1.0f / (float)((long long)B * HW);
  const double n = (double)B * HW;
  const bool a16 = ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)dy) & 15) == 0);
  const int vec = (a16 && (HW & 3) == 0) ? 4 : ((a16 && (HW & 1) == 0) ? 2 : 1);
  const int hwv = HW / vec;
  const int segv = hwv <= 8 * BN_THREADS ? hwv : 4 * BN_THREADS;
  const int nseg = (hwv + segv - 1) / segv;
  // samples per workgroup: doubled while the launch keeps >= 2048 workgroups (1024 while a workgroup has fewer than four
  // vectors per thread) and a workgroup stays at <= 16 vectors per thread
  const long long per_plane = segv < hwv ? segv : hwv;
  int PB = 1;
  while (PB < B && 2 * PB * per_plane <= 16 * BN_THREADS &&
         (long long)C * ((B + 2 * PB - 1) / (2 * PB)) * nseg >= (PB * per_plane < 4 * BN_THREADS ? 1024 : 2048))
    PB *= 2;
  const dim3 grid(C, (B + PB - 1) / PB, nseg);
#define BN_FINAPPLY(V)                                                                                                      \
  hipLaunchKernelGGL((bn_finapply_kernel<MODE, V>), grid, dim3(BN_THREADS), 0, stream, x, dy, partials, save_mean, save_invstd, \
                     running_mean, running_var, weight, bias, gw, gb, out, B, C, HW, S, PB, segv, n, eps, momentum, inv_n, act)
  if (vec == 4) BN_FINAPPLY(4);
  else if (vec == 2) BN_FINAPPLY(2);
  else BN_FINAPPLY(1);
#undef BN_FINAPPLY
}

// y = act(BatchNorm_train(x)); x, y [B,C,HW] fp32 contiguous; weight / bias [C] or NULL; running_mean / running_var [C]
// or NULL (updated in place with `momentum`); save_mean / save_invstd [C] out; act 0 none, 1 ReLU, 2 Hardswish.
static int bn_act_forward_impl(const float* x, const float* res, const float* weight, const float* bias, float* running_mean,
                               float* running_var, float* y, float* save_mean, float* save_invstd, double* scratch, int B, int C,
                               int HW, float eps, float momentum, int act, void* stream_, float* pooled = nullptr);
extern "C" int ias_bn_act_forward(const float* x, const float* weight, const float* bias, float* running_mean,
                                  float* running_var, float* y, float* save_mean, float* save_invstd, double* scratch,
                                  int B, int C, int HW, float eps, float momentum, int act, void* stream_) {
  return bn_act_forward_impl(x, nullptr, weight, bias, running_mean, running_var, y, save_mean, save_invstd, scratch, B, C, HW,
                             eps, momentum, act, stream_);
}
// y = act(BatchNorm_train(x)) + res: the residual connection of an inverted-residual block (torchvision InvertedResidual:
// `result += input` behind the block's last ConvNormActivation) taken in the same pass -- one elementwise launch and one
// read + write of the map less per block; the same bits as the separate addition.  res [B,C,HW] like x, 16-byte aligned
// where x is.
extern "C" int ias_bn_act_forward_res(const float* x, const float* res, const float* weight, const float* bias,
                                      float* running_mean, float* running_var, float* y, float* save_mean, float* save_invstd,
                                      double* scratch, int B, int C, int HW, float eps, float momentum, int act, void* stream_) {
  if (!res) return IAS_ERR_ARG;
  return bn_act_forward_impl(x, res, weight, bias, running_mean, running_var, y, save_mean, save_invstd, scratch, B, C, HW, eps,
                             momentum, act, stream_);
}
extern "C" int ias_se_plane_reduce(const float* x, const float* m, float* out, long long planes, int HW, float scale, void* stream_);   // se_kernels.hip
static int bn_act_forward_impl(const float* x, const float* res, const float* weight, const float* bias, float* running_mean,
                               float* running_var, float* y, float* save_mean, float* save_invstd, double* scratch, int B, int C,
                               int HW, float eps, float momentum, int act, void* stream_, float* pooled) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !y || !save_mean || !save_invstd || !scratch || B <= 0 || C <= 0 || C > 65535 || HW <= 0 || act < 0 || act > 2)
    return IAS_ERR_ARG;
  if (const int nv = bn_fused_nv(x, y, res, B, C, HW)) {
#define BNF_FWD(NV, POOL)                                                                                                   \
  hipLaunchKernelGGL((bn_fused_forward_kernel<NV, POOL>), dim3(C), dim3(BNF_THREADS), 0, stream, (const float4*)x, weight, bias, \
                     running_mean, running_var, (float4*)y, save_mean, save_invstd, B, C, HW >> 2, eps, momentum, act,       \
                     (const float4*)res, pooled)
    if (nv == 2) { if (pooled) BNF_FWD(2, true); else BNF_FWD(2, false); }
    else { if (pooled) BNF_FWD(8, true); else BNF_FWD(8, false); }
#undef BNF_FWD
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }
  const int S = bn_split(B, C);
  hipLaunchKernelGGL((bn_partials_kernel<0>), dim3(C, S), dim3(BN_THREADS), 0, stream, x, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, scratch, B, C,
                     HW, act);
  if (bn_finapply_takes(B, C, HW)) {
    bn_launch_finapply<0>(stream, x, res, scratch, save_mean, save_invstd, running_mean, running_var, weight, bias, nullptr,
                          nullptr, y, B, C, HW, S, eps, momentum, act);
  } else {
    hipLaunchKernelGGL(bn_finalize_stats_kernel, dim3((C + BN_FIN_WAVES - 1) / BN_FIN_WAVES), dim3(64 * BN_FIN_WAVES), 0, stream, x, scratch, save_mean, save_invstd,
                       running_mean, running_var, C, S, HW, (double)B * HW, eps, momentum);
    bn_launch_apply<0>(stream, x, res, save_mean, save_invstd, weight, bias, nullptr, y, B, C, HW, act);
  }
  if (hipGetLastError() != hipSuccess) return IAS_ERR_LAUNCH;
  // (maps too large for the one-workgroup-per-channel kernel: the pool as its own pass)
  if (pooled) return ias_se_plane_reduce(y, nullptr, pooled, (long long)B * C, HW, 1.0f / (float)HW, stream_);
  return IAS_OK;
}
// ias_bn_act_forward that also leaves pooled[b][c] = mean_hw y[b][c] behind ([B,C]): the squeeze-excitation block's average
// pool, taken by the normalisation's own launch on maps up to 15 x 16
extern "C" int ias_bn_act_forward_pool(const float* x, const float* weight, const float* bias, float* running_mean,
                                       float* running_var, float* y, float* pooled, float* save_mean, float* save_invstd,
                                       double* scratch, int B, int C, int HW, float eps, float momentum, int act, void* stream_) {
  if (!pooled) return IAS_ERR_ARG;
  return bn_act_forward_impl(x, nullptr, weight, bias, running_mean, running_var, y, save_mean, save_invstd, scratch, B, C, HW,
                             eps, momentum, act, stream_, pooled);
}

// Backward of ias_bn_act_forward: dy [B,C,HW] -> dx [B,C,HW], gw / gb [C] (either may be NULL); sums [C][2] floats scratch.
extern "C" int ias_bn_act_backward(const float* x, const float* dy, const float* weight, const float* bias,
                                   const float* save_mean, const float* save_invstd, float* dx, float* gw, float* gb,
                                   double* scratch, float* sums, int B, int C, int HW, int act, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !dy || !dx || !save_mean || !save_invstd || !scratch || !sums || B <= 0 || C <= 0 || C > 65535 || HW <= 0 ||
      act < 0 || act > 2)
    return IAS_ERR_ARG;
  if (const int nv = bn_fused_nv(x, dy, dx, B, C, HW)) {
    if (nv == 2)
      hipLaunchKernelGGL((bn_fused_backward_kernel<2>), dim3(C), dim3(BNF_THREADS), 0, stream, (const float4*)x,
                         (const float4*)dy, weight, bias, save_mean, save_invstd, (float4*)dx, gw, gb, B, C, HW >> 2, act);
    else
      hipLaunchKernelGGL((bn_fused_backward_kernel<8>), dim3(C), dim3(BNF_THREADS), 0, stream, (const float4*)x,
                         (const float4*)dy, weight, bias, save_mean, save_invstd, (float4*)dx, gw, gb, B, C, HW >> 2, act);
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }
  const int S = bn_split(B, C);
  hipLaunchKernelGGL((bn_partials_kernel<1>), dim3(C, S), dim3(BN_THREADS), 0, stream, x, dy, save_mean, save_invstd, weight,
                     bias, scratch, B, C, HW, act);
  if (bn_finapply_takes(B, C, HW)) {
    bn_launch_finapply<1>(stream, x, dy, scratch, const_cast<float*>(save_mean), const_cast<float*>(save_invstd), nullptr, nullptr,
                          weight, bias, gw, gb, dx, B, C, HW, S, 0.0f, 0.0f, act);
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(bn_finalize_grads_kernel, dim3((C + BN_FIN_WAVES - 1) / BN_FIN_WAVES), dim3(64 * BN_FIN_WAVES), 0, stream, scratch, gw, gb, sums, C, S);
  bn_launch_apply<1>(stream, x, dy, save_mean, save_invstd, weight, bias, sums, dx, B, C, HW, act);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// Training-mode BatchNorm1d (+ ReLU) on ROW GROUPS of a [G n, F] matrix: each group of n consecutive rows is normalised
// with its own batch statistics, the groups update the running statistics one after the other (group 0 first) -- what
// the reference's shared projector does when it is applied to the audio branch and then to the parameter branch
// (/root/reference/vicreg.py:27-30: `self.projector(...)` twice; Linear -> BatchNorm1d -> ReLU per layer, vicreg.py:60-70).
// With both branches stacked (vicreg.project_pair) torch needed per layer and step: 2 x (counter add, statistics,
// finalize, transform) + cat + ReLU forward and ReLU', 2 x (zero fill, copy, reduce, elementwise), 3 adds backward --
// ~24 launches of 5-8 us each on an 8 MB activation.  Here: ONE launch per direction.  The preceding Linear's bias is an
// argument (`lin_bias`, added on the fly; its gradient, the column sums of dx, comes out of the backward launch), so the
// GEMM in front runs without a bias pass and without a column-sum launch behind it.
//
// A workgroup owns BN1_FT features and walks the groups; a thread owns one feature and every BN1_RG-th row, the rows of a
// group in registers (n <= BN1_RG * RPT; RPT = 0: re-read from memory -- the tile is 64 KB, it stays in L2).  64 features
// per workgroup: a wave's row access is 256 contiguous bytes (with 32 the kernels ran at 1.2 TB/s on 128-byte segments).
// All sums in a fixed order: deterministic.
#define BN1_FT 64
#define BN1_RG 8
#define BN1_THREADS (BN1_FT * BN1_RG)

// sums over the BN1_RG row groups of a feature, fixed order, NV values at once (one pair of barriers); every thread of
// the feature gets the results
template <int NV>
__device__ __forceinline__ void bn1_reduce(float (&v)[NV], float (*red)[BN1_RG][BN1_FT], int tx, int ty) {
  __syncthreads();              // the previous use of `red` is over
#pragma unroll
  for (int j = 0; j < NV; ++j) red[j][ty][tx] = v[j];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    float s = red[j][0][tx];
#pragma unroll
    for (int k = 1; k < BN1_RG; ++k) s += red[j][k][tx];
    v[j] = s;
  }
}

// GB: row groups handled per pass (2: both branches of the projector at once -- every load of the pass in flight together,
// half the barriers; the running statistics still chain in group order)
template <int RPT, int GB>
__global__ __launch_bounds__(BN1_THREADS) void bn1d_groups_forward_kernel(
    const float* __restrict__ z, const float* __restrict__ lin_bias, const float* __restrict__ weight,
    const float* __restrict__ bias, float* __restrict__ running_mean, float* __restrict__ running_var,
    long long* __restrict__ num_batches_tracked, float* __restrict__ y, float* __restrict__ save_mean,
    float* __restrict__ save_invstd, int G, int n, int F, float eps, float momentum, int relu) {
  __shared__ float red[2 * GB][BN1_RG][BN1_FT];
  const int tx = threadIdx.x % BN1_FT, ty = threadIdx.x / BN1_FT;
  const int f = blockIdx.x * BN1_FT + tx;
  const bool live = f < F;
  const int fc = live ? f : F - 1;                       // dead lanes read a valid column and write nothing
  const float lb = lin_bias ? lin_bias[fc] : 0.0f, w = weight ? weight[fc] : 1.0f, b = bias ? bias[fc] : 0.0f;
  float rm = running_mean ? running_mean[fc] : 0.0f, rv = running_var ? running_var[fc] : 0.0f;
  const float inv_n = 1.0f / (float)n, unbias = (float)n / (float)(n - 1);
  for (int g0 = 0; g0 < G; g0 += GB) {                   // (G % GB == 0: the launcher)
    float xv[GB][RPT > 0 ? RPT : 1];
    float s[GB], q[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      const float* zg = z + (size_t)(g0 + j) * n * F + fc;
      s[j] = 0.0f;
      if (RPT > 0) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const int r = ty + i * BN1_RG;
          xv[j][i] = r < n ? zg[(size_t)r * F] + lb : 0.0f;
          s[j] += xv[j][i];
        }
      } else {
        for (int r = ty; r < n; r += BN1_RG) s[j] += zg[(size_t)r * F] + lb;
      }
    }
    bn1_reduce<GB>(s, red, tx, ty);
    float mean[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      const float* zg = z + (size_t)(g0 + j) * n * F + fc;
      mean[j] = s[j] * inv_n;
      q[j] = 0.0f;
      if (RPT > 0) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const float d = xv[j][i] - mean[j];
          q[j] += (ty + i * BN1_RG < n) ? d * d : 0.0f;
        }
      } else {
        for (int r = ty; r < n; r += BN1_RG) { const float d = (zg[(size_t)r * F] + lb) - mean[j]; q[j] += d * d; }
      }
    }
    bn1_reduce<GB>(q, red, tx, ty);
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      const int g = g0 + j;
      const float* zg = z + (size_t)g * n * F + fc;
      float* yg = y + (size_t)g * n * F + fc;
      const float var = q[j] * inv_n;
      const float invstd = 1.0f / sqrtf(var + eps);
      const float a = invstd * w;
      if (RPT > 0) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const int r = ty + i * BN1_RG;
          const float v = (xv[j][i] - mean[j]) * a + b;
          if (live && r < n) yg[(size_t)r * F] = relu ? fmaxf(v, 0.0f) : v;
        }
      } else {
        for (int r = ty; r < n; r += BN1_RG) {
          const float v = ((zg[(size_t)r * F] + lb) - mean[j]) * a + b;
          if (live) yg[(size_t)r * F] = relu ? fmaxf(v, 0.0f) : v;
        }
      }
      if (live && ty == 0) {
        save_mean[(size_t)g * F + f] = mean[j];
        save_invstd[(size_t)g * F + f] = invstd;
      }
      rm = (1.0f - momentum) * rm + momentum * mean[j];    // as torch: one update per call, in call order
      rv = (1.0f - momentum) * rv + momentum * (var * unbias);
    }
  }
  if (live && ty == 0) {
    if (running_mean) running_mean[f] = rm;
    if (running_var) running_var[f] = rv;
  }
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += G;
}

// dy [G n, F] -> dx [G n, F] (the gradient of the Linear output in front), gw / gb [F] (BatchNorm weight and bias,
// summed over the groups in group order), g_lin_bias [F] (column sums of dx).
template <int RPT, int GB>
__global__ __launch_bounds__(BN1_THREADS) void bn1d_groups_backward_kernel(
    const float* __restrict__ z, const float* __restrict__ lin_bias, const float* __restrict__ dy,
    const float* __restrict__ weight, const float* __restrict__ bias, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, float* __restrict__ dx, float* __restrict__ gw, float* __restrict__ gb,
    float* __restrict__ g_lin_bias, int G, int n, int F, int relu) {
  __shared__ float red[2 * GB][BN1_RG][BN1_FT];
  const int tx = threadIdx.x % BN1_FT, ty = threadIdx.x / BN1_FT;
  const int f = blockIdx.x * BN1_FT + tx;
  const bool live = f < F;
  const int fc = live ? f : F - 1;
  const float lb = lin_bias ? lin_bias[fc] : 0.0f, w = weight ? weight[fc] : 1.0f, b = bias ? bias[fc] : 0.0f;
  const float inv_n = 1.0f / (float)n;
  float gw_acc = 0.0f, gb_acc = 0.0f, glb_acc = 0.0f;
  for (int g0 = 0; g0 < G; g0 += GB) {
    float xh[GB][RPT > 0 ? RPT : 1], dz[GB][RPT > 0 ? RPT : 1];
    float mean[GB], invstd[GB], a[GB], ss[2 * GB];
    // the forward's own expression for the pre-activation value: the ReLU mask is the forward's mask
    auto one = [&](int j, size_t base, int r, float& xhat, float& d) {
      const float c = (z[base + (size_t)r * F] + lb) - mean[j];
      const float v = c * a[j] + b;
      xhat = c * invstd[j];
      d = dy[base + (size_t)r * F];
      if (relu && !(v > 0.0f)) d = 0.0f;
    };
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      const size_t base = (size_t)(g0 + j) * n * F + fc;
      mean[j] = save_mean[(size_t)(g0 + j) * F + fc]; invstd[j] = save_invstd[(size_t)(g0 + j) * F + fc];
      a[j] = invstd[j] * w;
      float s1 = 0.0f, s2 = 0.0f;
      if (RPT > 0) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const int r = ty + i * BN1_RG;
          xh[j][i] = 0.0f; dz[j][i] = 0.0f;
          if (r < n) one(j, base, r, xh[j][i], dz[j][i]);
          s1 += dz[j][i]; s2 += dz[j][i] * xh[j][i];
        }
      } else {
        for (int r = ty; r < n; r += BN1_RG) { float xhat, d; one(j, base, r, xhat, d); s1 += d; s2 += d * xhat; }
      }
      ss[2 * j] = s1; ss[2 * j + 1] = s2;
    }
    bn1_reduce<2 * GB>(ss, red, tx, ty);
    float sb[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
      const size_t base = (size_t)(g0 + j) * n * F + fc;
      const float m1 = ss[2 * j] * inv_n, m2 = ss[2 * j + 1] * inv_n;
      sb[j] = 0.0f;
      if (RPT > 0) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const int r = ty + i * BN1_RG;
          const float v = a[j] * (dz[j][i] - m1 - xh[j][i] * m2);
          if (r < n) { sb[j] += v; if (live) dx[base + (size_t)r * F] = v; }
        }
      } else {
        for (int r = ty; r < n; r += BN1_RG) {
          float xhat, d; one(j, base, r, xhat, d);
          const float v = a[j] * (d - m1 - xhat * m2);
          sb[j] += v;
          if (live) dx[base + (size_t)r * F] = v;
        }
      }
    }
    bn1_reduce<GB>(sb, reinterpret_cast<float (*)[BN1_RG][BN1_FT]>(red), tx, ty);
#pragma unroll
    for (int j = 0; j < GB; ++j) { gw_acc += ss[2 * j + 1]; gb_acc += ss[2 * j]; glb_acc += sb[j]; }
  }
  if (live && ty == 0) {
    if (gw) gw[f] = gw_acc;
    if (gb) gb[f] = gb_acc;
    if (g_lin_bias) g_lin_bias[f] = glb_acc;
  }
}

static int bn1_rpt(int n) { return n <= BN1_RG * 16 ? 16 : (n <= BN1_RG * 32 ? 32 : 0); }

// z [G n, F] fp32 row-major (the Linear output WITHOUT its bias when lin_bias is given), y the same shape; lin_bias, weight,
// bias, running_mean, running_var [F] or NULL; num_batches_tracked: one int64 or NULL (+= G); save_mean / save_invstd [G, F].
extern "C" int ias_bn1d_groups_forward(const float* z, const float* lin_bias, const float* weight, const float* bias,
                                       float* running_mean, float* running_var, long long* num_batches_tracked, float* y,
                                       float* save_mean, float* save_invstd, int G, int n, int F, float eps, float momentum,
                                       int relu, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!z || !y || !save_mean || !save_invstd || G <= 0 || n < 2 || F <= 0 ||
      !(eps >= 0.0f))
    return IAS_ERR_ARG;
  const dim3 grid((F + BN1_FT - 1) / BN1_FT), block(BN1_THREADS);
#define BN1_FWD(RPT, GB) hipLaunchKernelGGL((bn1d_groups_forward_kernel<RPT, GB>), grid, block, 0, stream, z, lin_bias, weight, bias, running_mean, running_var, num_batches_tracked, y, save_mean, save_invstd, G, n, F, eps, momentum, relu)
  switch (bn1_rpt(n)) {
    case 16: if (G % 2 == 0) BN1_FWD(16, 2); else BN1_FWD(16, 1); break;
    case 32: BN1_FWD(32, 1); break;
    default: BN1_FWD(0, 1); break;
  }
#undef BN1_FWD
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

extern "C" int ias_bn1d_groups_backward(const float* z, const float* lin_bias, const float* dy, const float* weight,
                                        const float* bias, const float* save_mean, const float* save_invstd, float* dx,
                                        float* gw, float* gb, float* g_lin_bias, int G, int n, int F, int relu,
                                        void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!z || !dy || !dx || !save_mean || !save_invstd || G <= 0 || n < 2 || F <= 0) return IAS_ERR_ARG;
  const dim3 grid((F + BN1_FT - 1) / BN1_FT), block(BN1_THREADS);
#define BN1_BWD(RPT, GB) hipLaunchKernelGGL((bn1d_groups_backward_kernel<RPT, GB>), grid, block, 0, stream, z, lin_bias, dy, weight, bias, save_mean, save_invstd, dx, gw, gb, g_lin_bias, G, n, F, relu)
  switch (bn1_rpt(n)) {
    case 16: BN1_BWD(16, 1); break;      // (two groups per pass: 256 VGPRs and scratch at 512 threads, and no faster)
    case 32: BN1_BWD(32, 1); break;
    default: BN1_BWD(0, 1); break;
  }
#undef BN1_BWD
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
