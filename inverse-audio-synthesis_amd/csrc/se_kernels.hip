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
