// Squeeze-and-excitation blocks of the MobileNetV3 trunk (torchvision SqueezeExcitation inside
// `mobilenet_v3_small(...).features`, /root/reference/vicreg_audio_params.py:52-54): the passes over the [B,C,H,W]
// activation.  As torch ops a block cost 13 passes over that tensor per training step (mean, s*x, and in backward two
// products, a sum, the expanded pooling gradient and the accumulation of the two input gradients); here it is 7:
//   forward   pooled[b,c] = mean_hw x                       (se_plane_reduce_kernel, MODE 0)
//             y = x * s[b,c]                                (se_scale_kernel)
//   backward  gs[b,c] = sum_hw gy * x                       (se_plane_reduce_kernel, MODE 1)
//             gx = gy * s[b,c] + gpooled[b,c] / (H W)       (se_scale_kernel with the additive term)
// The two 1x1 convolutions on [B,C] between them stay GEMMs in torch (vision.py: _SEFn).  Sums are taken in a fixed
// order (deterministic).
#include "ias_common.h"

#define SE_THREADS 256

// one group of G lanes per plane (G = 16 or 64, a wave holds 64 / G planes); MODE 0: mean(a), MODE 1: sum(a * b)
template <int MODE, int G>
__global__ __launch_bounds__(SE_THREADS) void se_plane_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                     float* __restrict__ out, int planes, int hw, float scale,
                                                                     int vec) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int gid = (blockIdx.x * SE_THREADS + threadIdx.x) / G, gl = threadIdx.x % G;
  const bool live = gid < planes;
  float acc = 0.0f;
  if (live) {
    const size_t base = (size_t)gid * hw;
    if (vec) {
      const f4* a4 = reinterpret_cast<const f4*>(a + base);
      const f4* b4 = reinterpret_cast<const f4*>(MODE == 1 ? b + base : a + base);
      for (int i = gl; i < (hw >> 2); i += G) {
        const f4 va = a4[i];
        if (MODE == 1) {
          const f4 vb = b4[i];
          acc += (va[0] * vb[0] + va[1] * vb[1]) + (va[2] * vb[2] + va[3] * vb[3]);
        } else {
          acc += (va[0] + va[1]) + (va[2] + va[3]);
        }
      }
    } else {
      for (int i = gl; i < hw; i += G) acc += MODE == 1 ? a[base + i] * b[base + i] : a[base + i];
    }
  }
#pragma unroll
  for (int d = G / 2; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (live && gl == 0) out[gid] = acc * scale;
}

// y[p][i] = x[p][i] * s[p] + (add ? add[p] * add_scale : 0)
template <int VEC>
__global__ __launch_bounds__(SE_THREADS) void se_scale_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                              const float* __restrict__ add, float* __restrict__ y,
                                                              long long total, int hwv, float add_scale) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (long long i = (long long)blockIdx.x * SE_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * SE_THREADS) {
    const long long p = i / hwv;
    const float sc = s[p], ad = add ? add[p] * add_scale : 0.0f;
    if (VEC == 4) {
      const f4 v = reinterpret_cast<const f4*>(x)[i];
      f4 o;
      o[0] = fmaf(v[0], sc, ad); o[1] = fmaf(v[1], sc, ad); o[2] = fmaf(v[2], sc, ad); o[3] = fmaf(v[3], sc, ad);
      reinterpret_cast<f4*>(y)[i] = o;
    } else {
      y[i] = fmaf(x[i], sc, ad);
    }
  }
}

static bool se_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// out[p] = scale * sum_i a[p][i] (b == NULL) or scale * sum_i a[p][i] b[p][i]; a, b [planes][hw]
extern "C" int ias_se_plane_reduce(const float* a, const float* b, float* out, long long planes, int hw, float scale,
                                   void* stream_) {
  if (!a || !out || planes <= 0 || planes > 0x7fffffffLL || hw <= 0) return IAS_ERR_ARG;
  const bool vec = (hw & 3) == 0 && se_aligned16(a) && (!b || se_aligned16(b));
  const int G = hw <= 64 ? 16 : 64;
  const long long groups_per_block = SE_THREADS / G;
  const dim3 grid((unsigned)((planes + groups_per_block - 1) / groups_per_block)), block(SE_THREADS);
  hipStream_t st = (hipStream_t)stream_;
  if (b) {
    if (G == 16) hipLaunchKernelGGL((se_plane_reduce_kernel<1, 16>), grid, block, 0, st, a, b, out, (int)planes, hw, scale, vec ? 1 : 0);
    else hipLaunchKernelGGL((se_plane_reduce_kernel<1, 64>), grid, block, 0, st, a, b, out, (int)planes, hw, scale, vec ? 1 : 0);
  } else {
    if (G == 16) hipLaunchKernelGGL((se_plane_reduce_kernel<0, 16>), grid, block, 0, st, a, b, out, (int)planes, hw, scale, vec ? 1 : 0);
    else hipLaunchKernelGGL((se_plane_reduce_kernel<0, 64>), grid, block, 0, st, a, b, out, (int)planes, hw, scale, vec ? 1 : 0);
  }
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// y[p][i] = x[p][i] * s[p] + (add ? add[p] * add_scale : 0); x, y [planes][hw], s, add [planes]
extern "C" int ias_se_scale(const float* x, const float* s, const float* add, float* y, long long planes, int hw,
                            float add_scale, void* stream_) {
  if (!x || !s || !y || planes <= 0 || hw <= 0) return IAS_ERR_ARG;
  const bool vec = (hw & 3) == 0 && se_aligned16(x) && se_aligned16(y);
  const long long total = vec ? planes * (hw >> 2) : planes * hw;
  long long g = (total + SE_THREADS - 1) / SE_THREADS;
  if (g > 8192) g = 8192;
  hipStream_t st = (hipStream_t)stream_;
  if (vec) hipLaunchKernelGGL((se_scale_kernel<4>), dim3((unsigned)g), dim3(SE_THREADS), 0, st, x, s, add, y, total, hw >> 2, add_scale);
  else hipLaunchKernelGGL((se_scale_kernel<1>), dim3((unsigned)g), dim3(SE_THREADS), 0, st, x, s, add, y, total, hw, add_scale);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// ---- the block's two 1x1 convolutions on [B, C]: fc1 (C -> Cs) + ReLU, fc2 (Cs -> C) + Hardsigmoid ----------------
// At batch 128 these are GEMMs of 10 MFLOP; as rocBLAS calls they cost 15-50 us each (six per block and step, 0.8 ms
// of a training step).  Here one workgroup per sample runs both layers out of LDS (forward), or the three
// matrix-vector products of their backward; the four parameter gradients, sums over the batch of outer products, are
// small LDS-tiled products.  All sums in a fixed order.
#define SEM_THREADS 256

// acc + sum_i row[i] v[i] with four running sums (row in global memory, v in LDS), n % 4 == 0 and row 16-byte aligned
// for the vector form
__device__ __forceinline__ float sem_dot(const float* __restrict__ row, const float* __restrict__ v, int n, float acc, bool vec) {
  float a0 = acc, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  if (vec) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* r4 = reinterpret_cast<const f4*>(row);
    for (int i = 0; i < (n >> 2); ++i) {
      const f4 w = r4[i];
      a0 = fmaf(w[0], v[4 * i], a0); a1 = fmaf(w[1], v[4 * i + 1], a1);
      a2 = fmaf(w[2], v[4 * i + 2], a2); a3 = fmaf(w[3], v[4 * i + 3], a3);
    }
  } else {
    for (int i = 0; i < n; ++i) a0 = fmaf(row[i], v[i], a0);
  }
  return (a0 + a1) + (a2 + a3);
}

// Sum over the 16 lanes of a DPP row, the same value in every lane of the row (two quad permutes, two row rotations).
__device__ __forceinline__ float sem_row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124 /* row_ror:4 */, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, true));
  return v;
}

// Forward of the two layers for SEF_SB samples per workgroup (round 4; the round-2 kernel above gave every THREAD a
// weight row: 64 different cache lines per load instruction, 17.6 us per call).  Here 16 LANES take a weight row: they
// read it with consecutive 16-byte loads (256 bytes per row and round, the row once for both samples), multiply against
// the samples' input vectors in LDS, and the 16 partial sums are folded by four DPP adds -- no LDS, no readlane (a first
// version with a whole wave per row spent its time in the 64-lane folds: 28.8 us).  A wave works on 4 rows at a time,
// the 8 waves on 32.  Every workgroup still reads both weight matrices, but 64 workgroups do instead of 128.
#define SEF_SB 2
#define SEF_THREADS 512
// out[s][o] = sum_k w[o][k] v[s][k] (+ bias[o]); v [SEF_SB][K] in LDS.  A 16-lane group works on SEF_RU rows at once
// (rows g, g + 32, ... of a round of 32 SEF_RU) and takes two 64-float steps of k per iteration: 2 SEF_RU 16-byte loads in
// flight per lane (with one, the C = 576 block waited out 80 L2 latencies: 53 us).
#define SEF_RU 4
template <typename Epilogue>
__device__ __forceinline__ void sef_layer(const float* __restrict__ w, const float* __restrict__ bias, const float* s_v, int K,
                                          int O, bool vec, Epilogue&& done, int o_begin = 0) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane & 15, grp = wave * 4 + (lane >> 4);       // 32 groups of 16 lanes
  constexpr int NGRP = SEF_THREADS / 16, RPG = NGRP * SEF_RU;    // rows per workgroup and round
  for (int o0 = o_begin; o0 < O; o0 += RPG) {                    // (uniform trip count: the DPP folds need whole rows of lanes)
    const float* row[SEF_RU];
    float a0[SEF_RU], a1[SEF_RU];
#pragma unroll
    for (int u = 0; u < SEF_RU; ++u) {
      const int o = o0 + grp + NGRP * u;
      row[u] = w + (size_t)(o < O ? o : 0) * K;
      a0[u] = 0.f; a1[u] = 0.f;
    }
    if (vec) {
      int k = 4 * sub;
      for (; k + 64 < K; k += 128) {
        f4 wa[SEF_RU], wb[SEF_RU];
#pragma unroll
        for (int u = 0; u < SEF_RU; ++u) { wa[u] = *reinterpret_cast<const f4*>(row[u] + k); wb[u] = *reinterpret_cast<const f4*>(row[u] + k + 64); }
        const f4 va0 = *reinterpret_cast<const f4*>(s_v + k), va1 = *reinterpret_cast<const f4*>(s_v + K + k);
        const f4 vb0 = *reinterpret_cast<const f4*>(s_v + k + 64), vb1 = *reinterpret_cast<const f4*>(s_v + K + k + 64);
#pragma unroll
        for (int u = 0; u < SEF_RU; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            a0[u] = fmaf(wa[u][e], va0[e], a0[u]); a1[u] = fmaf(wa[u][e], va1[e], a1[u]);
            a0[u] = fmaf(wb[u][e], vb0[e], a0[u]); a1[u] = fmaf(wb[u][e], vb1[e], a1[u]);
          }
      }
      for (; k < K; k += 64) {
        f4 wa[SEF_RU];
#pragma unroll
        for (int u = 0; u < SEF_RU; ++u) wa[u] = *reinterpret_cast<const f4*>(row[u] + k);
        const f4 va0 = *reinterpret_cast<const f4*>(s_v + k), va1 = *reinterpret_cast<const f4*>(s_v + K + k);
#pragma unroll
        for (int u = 0; u < SEF_RU; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { a0[u] = fmaf(wa[u][e], va0[e], a0[u]); a1[u] = fmaf(wa[u][e], va1[e], a1[u]); }
      }
    } else {
      for (int k = sub; k < K; k += 16) {
        const float v0 = s_v[k], v1 = s_v[K + k];
#pragma unroll
        for (int u = 0; u < SEF_RU; ++u) {
          const float wv = row[u][k];
          a0[u] = fmaf(wv, v0, a0[u]); a1[u] = fmaf(wv, v1, a1[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < SEF_RU; ++u) {
      a0[u] = sem_row16_sum(a0[u]);
      a1[u] = sem_row16_sum(a1[u]);
    }
#pragma unroll
    for (int u = 0; u < SEF_RU; ++u) {                           // lane u of the group stores row u
      const int o = o0 + grp + NGRP * u;
      if (o < O && sub == u) {
        const float b0 = bias ? bias[o] : 0.0f;
        done(o, a0[u] + b0, a1[u] + b0);
      }
    }
  }
}

__global__ __launch_bounds__(SEF_THREADS) void se_mlp_forward_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                                     const float* __restrict__ b2, float* __restrict__ h,
                                                                     float* __restrict__ z, float* __restrict__ s, int B, int C,
                                                                     int Cs, int vec1, int vec2) {
  extern __shared__ __attribute__((aligned(16))) float sem_lds[];
  float* s_p = sem_lds;                    // [SEF_SB][C]
  float* s_h = sem_lds + SEF_SB * C;       // [SEF_SB][Cs]   (C % 4 == 0 whenever vec2 is set: see the launcher)
  const int bs = blockIdx.x * SEF_SB, tid = threadIdx.x;
  const bool has1 = bs + 1 < B;            // (odd batch: the last workgroup's second sample repeats the first, not stored)
  for (int i = tid; i < SEF_SB * C; i += SEF_THREADS) {
    const int sm = i / C, c = i - sm * C;
    s_p[i] = pooled[(size_t)(bs + (sm && has1 ? 1 : 0)) * C + c];
  }
  __syncthreads();
  sef_layer(w1, b1, s_p, C, Cs, vec1 != 0, [&](int o, float v0, float v1) {
    v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f);
    s_h[o] = v0; s_h[Cs + o] = v1;
    if (blockIdx.y == 0) {
      h[(size_t)bs * Cs + o] = v0;
      if (has1) h[(size_t)(bs + 1) * Cs + o] = v1;
    }
  });
  __syncthreads();
  // gridDim.y workgroups per sample pair (wide blocks: a workgroup streams its weights at ~33 GB/s, and 64 workgroups
  // leave three quarters of the chip idle): every one of them computes ALL of h (it needs it), then its own slice of the
  // second layer's outputs -- C = 576 with five slices: 663 -> 398 KB of weights per workgroup
  const int per = (C + (int)gridDim.y - 1) / (int)gridDim.y, c_lo = (int)blockIdx.y * per, c_hi = min(C, c_lo + per);
  sef_layer(w2, b2, s_h, Cs, c_hi, vec2 != 0, [&](int o, float v0, float v1) {
    z[(size_t)bs * C + o] = v0;
    s[(size_t)bs * C + o] = fminf(fmaxf(v0 + 3.0f, 0.0f), 6.0f) / 6.0f;
    if (has1) {
      z[(size_t)(bs + 1) * C + o] = v1;
      s[(size_t)(bs + 1) * C + o] = fminf(fmaxf(v1 + 3.0f, 0.0f), 6.0f) / 6.0f;
    }
  }, c_lo);
}

// per sample: gz = gs * hardsigmoid'(z); gh = relu'(h) * W2^T gz; gp = W1^T gh   (lanes along the output index: the
// rows of W2 / W1 are read coalesced, the vector operand is broadcast from LDS)
__global__ __launch_bounds__(SEM_THREADS) void se_mlp_backward_sample_kernel(const float* __restrict__ gs, const float* __restrict__ z,
                                                                             const float* __restrict__ h, const float* __restrict__ w1,
                                                                             const float* __restrict__ w2, float* __restrict__ gz,
                                                                             float* __restrict__ gh, float* __restrict__ gp, int C,
                                                                             int Cs) {
  extern __shared__ __attribute__((aligned(16))) float sem_lds[];
  float* s_gz = sem_lds;           // [C]
  float* s_gh = sem_lds + C;       // [Cs]
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += SEM_THREADS) {
    const float zz = z[(size_t)b * C + c];
    const float g = (zz > -3.0f && zz < 3.0f) ? gs[(size_t)b * C + c] * (1.0f / 6.0f) : 0.0f;
    s_gz[c] = g;
    gz[(size_t)b * C + c] = g;
  }
  __syncthreads();
  for (int j = tid; j < Cs; j += SEM_THREADS) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int c = 0;
    for (; c + 3 < C; c += 4) {
      a0 = fmaf(s_gz[c], w2[(size_t)c * Cs + j], a0); a1 = fmaf(s_gz[c + 1], w2[(size_t)(c + 1) * Cs + j], a1);
      a2 = fmaf(s_gz[c + 2], w2[(size_t)(c + 2) * Cs + j], a2); a3 = fmaf(s_gz[c + 3], w2[(size_t)(c + 3) * Cs + j], a3);
    }
    for (; c < C; ++c) a0 = fmaf(s_gz[c], w2[(size_t)c * Cs + j], a0);
    const float g = h[(size_t)b * Cs + j] > 0.0f ? (a0 + a1) + (a2 + a3) : 0.0f;
    s_gh[j] = g;
    gh[(size_t)b * Cs + j] = g;
  }
  __syncthreads();
  for (int c = tid; c < C; c += SEM_THREADS) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int j = 0;
    for (; j + 3 < Cs; j += 4) {
      a0 = fmaf(s_gh[j], w1[(size_t)j * C + c], a0); a1 = fmaf(s_gh[j + 1], w1[(size_t)(j + 1) * C + c], a1);
      a2 = fmaf(s_gh[j + 2], w1[(size_t)(j + 2) * C + c], a2); a3 = fmaf(s_gh[j + 3], w1[(size_t)(j + 3) * C + c], a3);
    }
    for (; j < Cs; ++j) a0 = fmaf(s_gh[j], w1[(size_t)j * C + c], a0);
    gp[(size_t)b * C + c] = (a0 + a1) + (a2 + a3);
  }
}

// The same three products for SEF_SB samples per workgroup with the REDUCTION split over the 8 waves (round 4).  The
// kernel above walks all C (or Cs) weight rows on one thread per output: 144-576 dependent rounds of four loads, 25.6 us
// per call, latency all the way.  Here wave w takes rows w, w + 8, ... (four rows = up to 12 loads in flight), its lanes own
// 4 consecutive outputs each (16-byte loads of the weight row, used for both samples; SEB_RU rows in flight), the 8 partial vectors meet in
// LDS and are added in wave order.  max(C, Cs) <= SEB_MAXW outputs (LDS), C % 4 == Cs % 4 == 0, 16-byte aligned weights.
#define SEB_MAXW 1024
#define SEB_RU 8          // weight rows (16-byte loads) in flight per lane
// part[wave][s][col] = sum over this wave's rows r of g[s][r] w[r][col]   (w [R][W] row-major, g [SEF_SB][R] in LDS)
__device__ __forceinline__ void seb_matvec_t(const float* __restrict__ w, int R, int W, const float* s_g, float* s_part) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NWAVE = SEF_THREADS / 64;
  for (int c0 = 4 * lane; c0 < W; c0 += 256) {
    f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
    int r = wave;
    for (; r + (SEB_RU - 1) * NWAVE < R; r += SEB_RU * NWAVE) {
      f4 wv[SEB_RU];
#pragma unroll
      for (int u = 0; u < SEB_RU; ++u) wv[u] = *reinterpret_cast<const f4*>(w + (size_t)(r + u * NWAVE) * W + c0);
#pragma unroll
      for (int u = 0; u < SEB_RU; ++u) {
        const float g0 = s_g[r + u * NWAVE], g1 = s_g[R + r + u * NWAVE];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a0[e] = fmaf(g0, wv[u][e], a0[e]); a1[e] = fmaf(g1, wv[u][e], a1[e]); }
      }
    }
    for (; r < R; r += NWAVE) {
      const f4 wv = *reinterpret_cast<const f4*>(w + (size_t)r * W + c0);
      const float g0 = s_g[r], g1 = s_g[R + r];
#pragma unroll
      for (int e = 0; e < 4; ++e) { a0[e] = fmaf(g0, wv[e], a0[e]); a1[e] = fmaf(g1, wv[e], a1[e]); }
    }
    *reinterpret_cast<f4*>(s_part + ((size_t)wave * SEF_SB + 0) * W + c0) = a0;
    *reinterpret_cast<f4*>(s_part + ((size_t)wave * SEF_SB + 1) * W + c0) = a1;
  }
}
__global__ __launch_bounds__(SEF_THREADS) void se_mlp_backward_pair_kernel(const float* __restrict__ gs, const float* __restrict__ z,
                                                                           const float* __restrict__ h, const float* __restrict__ w1,
                                                                           const float* __restrict__ w2, float* __restrict__ gz,
                                                                           float* __restrict__ gh, float* __restrict__ gp, int B,
                                                                           int C, int Cs) {
  extern __shared__ __attribute__((aligned(16))) float sem_lds[];
  constexpr int NWAVE = SEF_THREADS / 64;
  float* s_gz = sem_lds;                         // [SEF_SB][C]
  float* s_gh = s_gz + SEF_SB * C;               // [SEF_SB][Cs]
  float* s_part = s_gh + SEF_SB * Cs;            // [NWAVE][SEF_SB][max(C, Cs)]
  const int bs = blockIdx.x * SEF_SB, tid = threadIdx.x;
  const bool has1 = bs + 1 < B;
  for (int i = tid; i < SEF_SB * C; i += SEF_THREADS) {
    const int sm = i / C, c = i - sm * C;
    float g = 0.0f;
    if (sm == 0 || has1) {
      const size_t at = (size_t)(bs + sm) * C + c;
      const float zz = z[at];
      g = (zz > -3.0f && zz < 3.0f) ? gs[at] * (1.0f / 6.0f) : 0.0f;
      gz[at] = g;
    }
    s_gz[i] = g;
  }
  __syncthreads();
  seb_matvec_t(w2, C, Cs, s_gz, s_part);         // W2^T gz
  __syncthreads();
  for (int i = tid; i < SEF_SB * Cs; i += SEF_THREADS) {
    const int sm = i / Cs, j = i - sm * Cs;
    float acc = 0.0f;
#pragma unroll
    for (int wv = 0; wv < NWAVE; ++wv) acc += s_part[((size_t)wv * SEF_SB + sm) * Cs + j];
    float g = 0.0f;
    if (sm == 0 || has1) {
      const size_t at = (size_t)(bs + sm) * Cs + j;
      g = h[at] > 0.0f ? acc : 0.0f;
      gh[at] = g;
    }
    s_gh[i] = g;
  }
  __syncthreads();
  seb_matvec_t(w1, Cs, C, s_gh, s_part);         // W1^T gh
  __syncthreads();
  for (int i = tid; i < SEF_SB * C; i += SEF_THREADS) {
    const int sm = i / C, c = i - sm * C;
    if (sm && !has1) continue;
    float acc = 0.0f;
#pragma unroll
    for (int wv = 0; wv < NWAVE; ++wv) acc += s_part[((size_t)wv * SEF_SB + sm) * C + c];
    gp[(size_t)(bs + sm) * C + c] = acc;
  }
}

// out[r][q] = sum_b a[b][r] m[b][q] (a [B,R], m [B,Q]), colsum[r] = sum_b a[b][r]: 32 x 32 output tiles, the batch in
// LDS slices of SEO_SLICE = 128 rows (the whole batch of BASELINE configs[2] in ONE load phase: all 32 loads of a thread
// are in flight together, one barrier pair -- slices of 32 made it four dependent load -> barrier -> compute rounds,
// 14.5 us for a 21 MFLOP product), a thread owns a 2 x 2 block; sums in batch order.
#define SEO_SLICE 128
// Both parameter gradients of a block's MLP in ONE launch (round 5: they were two launches of ~6 us, 18 per pretraining
// step, all latency): workgroups [0, n1) take problem 1 (a1, m1 -> out1, colsum1; R1 x Q1 in tiles of 32 x 32, row tiles
// fastest), the others problem 2; the arithmetic per output is unchanged.
struct SeOuterProblem { const float* a; const float* m; float* out; float* colsum; int R, Q; };
__global__ __launch_bounds__(SEM_THREADS) void se_outer_sum_kernel(const SeOuterProblem p1, const SeOuterProblem p2, int n1, int B) {
  __shared__ float s_a[SEO_SLICE][33], s_m[SEO_SLICE][33];
  const bool second = (int)blockIdx.x >= n1;                       // (uniform)
  const float* __restrict__ a = second ? p2.a : p1.a;
  const float* __restrict__ m = second ? p2.m : p1.m;
  float* __restrict__ out = second ? p2.out : p1.out;
  float* __restrict__ colsum = second ? p2.colsum : p1.colsum;
  const int R = second ? p2.R : p1.R, Q = second ? p2.Q : p1.Q;
  const int tile = (int)blockIdx.x - (second ? n1 : 0), rt = (R + 31) / 32, tile_q = tile / rt;
  const int r0 = (tile - tile_q * rt) * 32, q0 = tile_q * 32, tid = threadIdx.x;
  const int tr = tid >> 4, tq = tid & 15;
  const int x = tid & 31, brow = tid >> 5;                       // loader: column x of rows brow + 8 i
  const bool aok = r0 + x < R, mok = q0 + x < Q;
  float acc[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, cs[2] = {0.0f, 0.0f};
  for (int b0 = 0; b0 < B; b0 += SEO_SLICE) {
    float va[SEO_SLICE / 8], vm[SEO_SLICE / 8];
#pragma unroll
    for (int i = 0; i < SEO_SLICE / 8; ++i) {
      const int bb = b0 + brow + 8 * i;
      va[i] = (bb < B && aok) ? a[(size_t)bb * R + r0 + x] : 0.0f;
      vm[i] = (bb < B && mok) ? m[(size_t)bb * Q + q0 + x] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SEO_SLICE / 8; ++i) { s_a[brow + 8 * i][x] = va[i]; s_m[brow + 8 * i][x] = vm[i]; }
    __syncthreads();
    const int nb = min(SEO_SLICE, B - b0);
#pragma unroll 8
    for (int bb = 0; bb < nb; ++bb) {
      const float a0 = s_a[bb][2 * tr], a1 = s_a[bb][2 * tr + 1], m0 = s_m[bb][2 * tq], m1 = s_m[bb][2 * tq + 1];
      acc[0][0] = fmaf(a0, m0, acc[0][0]); acc[0][1] = fmaf(a0, m1, acc[0][1]);
      acc[1][0] = fmaf(a1, m0, acc[1][0]); acc[1][1] = fmaf(a1, m1, acc[1][1]);
      cs[0] += a0; cs[1] += a1;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + 2 * tr + i;
    if (r >= R) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = q0 + 2 * tq + j;
      if (q < Q) out[(size_t)r * Q + q] = acc[i][j];
    }
    if (colsum && tile_q == 0 && tq == 0) colsum[r] = cs[i];
  }
}

// h = relu(pooled W1^T + b1) [B,Cs], z = h W2^T + b2 [B,C], s = hardsigmoid(z); pooled [B,C], w1 [Cs,C], w2 [C,Cs]
extern "C" int ias_se_mlp_forward(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2,
                                  float* h, float* z, float* s, int B, int C, int Cs, void* stream_) {
  if (!pooled || !w1 || !w2 || !h || !z || !s || B <= 0 || C <= 0 || Cs <= 0 || C + Cs > 12288) return IAS_ERR_ARG;
  // 16-byte accesses: rows of w1 (C floats) / w2 (Cs floats) and the LDS vectors [SEF_SB][C], [SEF_SB][Cs] behind them
  const int vec1 = ((C & 3) == 0 && se_aligned16(w1)) ? 1 : 0, vec2 = ((Cs & 3) == 0 && (C & 3) == 0 && se_aligned16(w2)) ? 1 : 0;
  (void)hipFuncSetAttribute((const void*)se_mlp_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)(sizeof(float) * (size_t)SEF_SB * (C + Cs)));
  const int slices = C >= 192 ? (C + 127) / 128 : 1;   // one round of 32 x SEF_RU = 128 second-layer rows per workgroup (C = 576: 5 slices of 116)
  hipLaunchKernelGGL(se_mlp_forward_kernel, dim3((B + SEF_SB - 1) / SEF_SB, slices), dim3(SEF_THREADS),
                     sizeof(float) * (size_t)SEF_SB * (C + Cs), (hipStream_t)stream_, pooled, w1, b1, w2, b2, h, z, s, B, C, Cs, vec1,
                     vec2);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// its backward from gs = d loss / d s: gp = d loss / d pooled [B,C], gw1 [Cs,C], gb1 [Cs], gw2 [C,Cs], gb2 [C];
// gz [B,C] and gh [B,Cs] are caller-owned scratch
extern "C" int ias_se_mlp_backward(const float* gs, const float* z, const float* h, const float* pooled, const float* w1,
                                   const float* w2, float* gz, float* gh, float* gp, float* gw1, float* gb1, float* gw2,
                                   float* gb2, int B, int C, int Cs, void* stream_) {
  if (!gs || !z || !h || !pooled || !w1 || !w2 || !gz || !gh || !gp || !gw1 || !gw2 || B <= 0 || C <= 0 || Cs <= 0 ||
      C + Cs > 12288)
    return IAS_ERR_ARG;
  hipStream_t st = (hipStream_t)stream_;
  if ((C & 3) == 0 && (Cs & 3) == 0 && C <= SEB_MAXW && Cs <= SEB_MAXW && se_aligned16(w1) && se_aligned16(w2)) {
    const int wmax = C > Cs ? C : Cs;
    const size_t lds = sizeof(float) * ((size_t)SEF_SB * (C + Cs) + (size_t)(SEF_THREADS / 64) * SEF_SB * wmax);
    (void)hipFuncSetAttribute((const void*)se_mlp_backward_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(se_mlp_backward_pair_kernel, dim3((B + SEF_SB - 1) / SEF_SB), dim3(SEF_THREADS), lds, st, gs, z, h, w1, w2,
                       gz, gh, gp, B, C, Cs);
  } else {
    hipLaunchKernelGGL(se_mlp_backward_sample_kernel, dim3(B), dim3(SEM_THREADS), sizeof(float) * (size_t)(C + Cs), st, gs, z, h,
                       w1, w2, gz, gh, gp, C, Cs);
  }
  const int ntiles = ((C + 31) / 32) * ((Cs + 31) / 32);
  const SeOuterProblem p2w = {gz, h, gw2, gb2, C, Cs}, p1w = {gh, pooled, gw1, gb1, Cs, C};
  hipLaunchKernelGGL(se_outer_sum_kernel, dim3(2 * ntiles), dim3(SEM_THREADS), 0, st, p2w, p1w, ntiles, B);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
