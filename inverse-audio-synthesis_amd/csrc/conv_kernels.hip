// Depthwise and stem convolutions of the AudioEmbedding trunk for MI355X (gfx950), NCHW fp32.
//
// The reference takes its trunk from torchvision (`mobilenet_v3_small(...).features`,
// /root/reference/vicreg_audio_params.py:52-54, used at audioembed.py:61).  On PyTorch-ROCm the eleven depthwise
// convolutions (3x3 / 5x5, stride 1 / 2) and the 3 -> 16 stem of that network run through MIOpen's fp32 fallbacks:
// naive_conv_* kernels, Winograd kernels of 0.85 ms for a 15 x 16 map, and an im2col + GEMM PER SAMPLE for the stem
// (measured 10 of the 40 ms of a batch-128 pretraining step).  They are memory-bound stencils; these kernels do them
// at stencil cost: one lane per output element, the k x k taps of the lane's channel in SGPRs (a workgroup stays
// inside one (b, c) plane), the weight gradient as per-workgroup partial sums reduced in a fixed order (deterministic).
#include "ias_common.h"

#define CV_THREADS 256

// out[b,c,ho,wo] = sum_{kh,kw} w[c,kh,kw] x[b,c,ho*S+kh-P,wo*S+kw-P]       (groups = C, zero padding P = (K-1)/2)
template <int K, int S>
__global__ __launch_bounds__(CV_THREADS) void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                float* __restrict__ out, int C, int H, int W, int Ho,
                                                                int Wo) {
  constexpr int P = (K - 1) / 2;
  const int plane = blockIdx.x, c = plane % C;
  float wt[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) wt[i] = w[c * K * K + i];
  const float* xp = x + (size_t)plane * H * W;
  float* op = out + (size_t)plane * Ho * Wo;
  for (int o = blockIdx.y * CV_THREADS + threadIdx.x; o < Ho * Wo; o += gridDim.y * CV_THREADS) {
    const int ho = o / Wo, wo = o - ho * Wo;
    float acc = 0.0f;
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      const int hi = ho * S + kh - P;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const int wi = wo * S + kw - P;
        if (wi >= 0 && wi < W) acc = fmaf(wt[kh * K + kw], xp[hi * W + wi], acc);
      }
    }
    op[o] = acc;
  }
}

// gx[b,c,hi,wi] = sum_{kh,kw : (hi+P-kh) % S == 0, ...} w[c,kh,kw] g[b,c,(hi+P-kh)/S,(wi+P-kw)/S]
template <int K, int S>
__global__ __launch_bounds__(CV_THREADS) void dwconv_bwd_data_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                                     float* __restrict__ gx, int C, int H, int W, int Ho,
                                                                     int Wo) {
  constexpr int P = (K - 1) / 2;
  const int plane = blockIdx.x, c = plane % C;
  float wt[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) wt[i] = w[c * K * K + i];
  const float* gp = g + (size_t)plane * Ho * Wo;
  float* xp = gx + (size_t)plane * H * W;
  for (int i = blockIdx.y * CV_THREADS + threadIdx.x; i < H * W; i += gridDim.y * CV_THREADS) {
    const int hi = i / W, wi = i - hi * W;
    float acc = 0.0f;
#pragma unroll
    for (int kh = 0; kh < K; ++kh) {
      const int th = hi + P - kh;
      if (th < 0 || (S > 1 && (th % S) != 0)) continue;
      const int ho = th / S;
      if (ho >= Ho) continue;
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const int tw = wi + P - kw;
        if (tw < 0 || (S > 1 && (tw % S) != 0)) continue;
        const int wo = tw / S;
        if (wo < Wo) acc = fmaf(wt[kh * K + kw], gp[ho * Wo + wo], acc);
      }
    }
    xp[i] = acc;
  }
}

// partial[chunk][c][kh*K+kw] = sum over the chunk's batch rows and all (ho,wo) of g[b,c,ho,wo] x[b,c,ho*S+kh-P,wo*S+kw-P]
// grid (C, nchunk); the chunk's planes are walked by the whole workgroup, a lane keeps its K*K sums in registers.
template <int K, int S>
__global__ __launch_bounds__(CV_THREADS) void dwconv_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                       float* __restrict__ partial, int B, int C, int H,
                                                                       int W, int Ho, int Wo, int rows_per_chunk) {
  constexpr int P = (K - 1) / 2;
  __shared__ float s_red[CV_THREADS / 64][K * K];
  const int c = blockIdx.x, chunk = blockIdx.y;
  const int b0 = chunk * rows_per_chunk, b1 = min(b0 + rows_per_chunk, B);
  float acc[K * K];
#pragma unroll
  for (int i = 0; i < K * K; ++i) acc[i] = 0.0f;
  const int n = Ho * Wo;
  for (int b = b0; b < b1; ++b) {
    const float* xp = x + ((size_t)b * C + c) * H * W;
    const float* gp = g + ((size_t)b * C + c) * n;
    for (int o = threadIdx.x; o < n; o += CV_THREADS) {
      const int ho = o / Wo, wo = o - ho * Wo;
      const float gv = gp[o];
#pragma unroll
      for (int kh = 0; kh < K; ++kh) {
        const int hi = ho * S + kh - P;
        if (hi < 0 || hi >= H) continue;
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const int wi = wo * S + kw - P;
          if (wi >= 0 && wi < W) acc[kh * K + kw] = fmaf(gv, xp[hi * W + wi], acc[kh * K + kw]);
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < K * K; ++i) {
    float v = acc[i];
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    if (lane == 0) s_red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < K * K) {
    float v = 0.0f;
    for (int wv = 0; wv < CV_THREADS / 64; ++wv) v += s_red[wv][threadIdx.x];
    partial[((size_t)chunk * C + c) * K * K + threadIdx.x] = v;
  }
}

// out[i] = sum_chunk partial[chunk][i]   (fixed order)
__global__ __launch_bounds__(CV_THREADS) void conv_reduce_partials_kernel(const float* __restrict__ partial,
                                                                          float* __restrict__ out, int n, int nchunk) {
  const int i = blockIdx.x * CV_THREADS + threadIdx.x;
  if (i >= n) return;
  float v = 0.0f;
  for (int k = 0; k < nchunk; ++k) v += partial[(size_t)k * n + i];
  out[i] = v;
}

// ---- stem: Conv2d(CIN, COUT, 3, stride 2, padding 1, bias=False), CIN = 3, COUT = 16 -----------------------------
// one lane per output position (b, ho, wo): the 27 inputs are read once, the 432 weights come from LDS (broadcast).
template <int CIN, int COUT>
__global__ __launch_bounds__(CV_THREADS) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ out, int H, int W, int Ho, int Wo) {
  __shared__ float s_w[COUT * CIN * 9];
  for (int i = threadIdx.x; i < COUT * CIN * 9; i += CV_THREADS) s_w[i] = w[i];
  __syncthreads();
  const int b = blockIdx.y;
  const float* xb = x + (size_t)b * CIN * H * W;
  float* ob = out + (size_t)b * COUT * Ho * Wo;
  for (int o = blockIdx.x * CV_THREADS + threadIdx.x; o < Ho * Wo; o += gridDim.x * CV_THREADS) {
    const int ho = o / Wo, wo = o - ho * Wo;
    float xin[CIN * 9];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int hi = ho * 2 + kh - 1, wi = wo * 2 + kw - 1;
          xin[(ci * 3 + kh) * 3 + kw] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? xb[((size_t)ci * H + hi) * W + wi] : 0.0f;
        }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < CIN * 9; ++i) acc = fmaf(s_w[co * CIN * 9 + i], xin[i], acc);
      ob[(size_t)co * Ho * Wo + o] = acc;
    }
  }
}

// partial[chunk][co][ci*9+kh*3+kw]: a workgroup stages 256 positions (27 inputs + COUT cotangents each) in LDS, then
// thread t < COUT*27 sums its weight element over them; grid (position chunks, B).
template <int CIN, int COUT>
__global__ __launch_bounds__(CV_THREADS) void stem_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                     float* __restrict__ partial, int H, int W, int Ho,
                                                                     int Wo, int chunks_x) {
  constexpr int NW = COUT * CIN * 9;     // 432
  static_assert(NW <= 2 * CV_THREADS, "two weight elements per thread at most");
  __shared__ float s_x[CIN * 9][CV_THREADS + 1];
  __shared__ float s_g[COUT][CV_THREADS + 1];
  const int b = blockIdx.y;
  const float* xb = x + (size_t)b * CIN * H * W;
  const float* gb = g + (size_t)b * COUT * Ho * Wo;
  float acc0 = 0.0f, acc1 = 0.0f;
  const int t0 = threadIdx.x, t1 = threadIdx.x + CV_THREADS;
  const int co0 = t0 / (CIN * 9), i0 = t0 - co0 * (CIN * 9);
  const int co1 = t1 / (CIN * 9), i1 = t1 - co1 * (CIN * 9);
  for (int base = blockIdx.x * CV_THREADS; base < Ho * Wo; base += chunks_x * CV_THREADS) {
    const int o = base + threadIdx.x;
    const bool ok = o < Ho * Wo;
    const int ho = ok ? o / Wo : 0, wo = ok ? o - ho * Wo : 0;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int hi = ho * 2 + kh - 1, wi = wo * 2 + kw - 1;
          s_x[(ci * 3 + kh) * 3 + kw][threadIdx.x] =
              (ok && hi >= 0 && hi < H && wi >= 0 && wi < W) ? xb[((size_t)ci * H + hi) * W + wi] : 0.0f;
        }
#pragma unroll
    for (int co = 0; co < COUT; ++co) s_g[co][threadIdx.x] = ok ? gb[(size_t)co * Ho * Wo + o] : 0.0f;
    __syncthreads();
    for (int p = 0; p < CV_THREADS; ++p) {
      acc0 = fmaf(s_g[co0][p], s_x[i0][p], acc0);
      if (t1 < NW) acc1 = fmaf(s_g[co1][p], s_x[i1][p], acc1);
    }
    __syncthreads();
  }
  float* pp = partial + ((size_t)b * chunks_x + blockIdx.x) * NW;
  pp[t0] = acc0;
  if (t1 < NW) pp[t1] = acc1;
}

// ------------------------------------------------------------------------ C ABI
static int cv_grid_x(int n) {
  int g = (n + CV_THREADS - 1) / CV_THREADS;
  return g < 1 ? 1 : (g > 64 ? 64 : g);
}

#define CV_DISPATCH(KERNEL, ...)                                                                     \
  do {                                                                                               \
    if (K == 3 && S == 1) hipLaunchKernelGGL((KERNEL<3, 1>), __VA_ARGS__);                           \
    else if (K == 3 && S == 2) hipLaunchKernelGGL((KERNEL<3, 2>), __VA_ARGS__);                      \
    else if (K == 5 && S == 1) hipLaunchKernelGGL((KERNEL<5, 1>), __VA_ARGS__);                      \
    else if (K == 5 && S == 2) hipLaunchKernelGGL((KERNEL<5, 2>), __VA_ARGS__);                      \
    else return IAS_ERR_UNSUPPORTED;                                                                 \
  } while (0)

static int cv_check(const void* a, const void* b, const void* c, int B, int C, int H, int W, int K, int S) {
  if (!a || !b || !c || B <= 0 || C <= 0 || H <= 0 || W <= 0 || (long long)B * C > 0x7fffffffLL) return IAS_ERR_ARG;
  if (!((K == 3 || K == 5) && (S == 1 || S == 2))) return IAS_ERR_UNSUPPORTED;
  return IAS_OK;
}

extern "C" int ias_conv_out_size(int n, int K, int S) { return (n + 2 * ((K - 1) / 2) - K) / S + 1; }

// Depthwise Conv2d(C, C, K, stride S, padding (K-1)/2, groups=C, bias=False) forward: x [B,C,H,W], w [C,1,K,K] -> out
extern "C" int ias_dwconv_forward(const float* x, const float* w, float* out, int B, int C, int H, int W, int K, int S,
                                  void* stream_) {
  int rc = cv_check(x, w, out, B, C, H, W, K, S);
  if (rc) return rc;
  const int Ho = ias_conv_out_size(H, K, S), Wo = ias_conv_out_size(W, K, S);
  const dim3 grid(B * C, cv_grid_x(Ho * Wo)), block(CV_THREADS);
  CV_DISPATCH(dwconv_fwd_kernel, grid, block, 0, (hipStream_t)stream_, x, w, out, C, H, W, Ho, Wo);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// its gradient w.r.t. the input: g [B,C,Ho,Wo] -> gx [B,C,H,W]
extern "C" int ias_dwconv_backward_data(const float* g, const float* w, float* gx, int B, int C, int H, int W, int K, int S,
                                        void* stream_) {
  int rc = cv_check(g, w, gx, B, C, H, W, K, S);
  if (rc) return rc;
  const int Ho = ias_conv_out_size(H, K, S), Wo = ias_conv_out_size(W, K, S);
  const dim3 grid(B * C, cv_grid_x(H * W)), block(CV_THREADS);
  CV_DISPATCH(dwconv_bwd_data_kernel, grid, block, 0, (hipStream_t)stream_, g, w, gx, C, H, W, Ho, Wo);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// floats of scratch for ias_dwconv_backward_weight
extern "C" long long ias_dwconv_weight_scratch(int B, int C, int K) {
  if (B <= 0 || C <= 0 || K <= 0) return IAS_ERR_ARG;
  int nchunk = B < 32 ? B : 32;
  return (long long)nchunk * C * K * K;
}

// its gradient w.r.t. the weights: x [B,C,H,W], g [B,C,Ho,Wo] -> gw [C,1,K,K]; scratch: ias_dwconv_weight_scratch floats
extern "C" int ias_dwconv_backward_weight(const float* x, const float* g, float* gw, float* scratch, int B, int C, int H,
                                          int W, int K, int S, void* stream_) {
  int rc = cv_check(x, g, gw, B, C, H, W, K, S);
  if (rc) return rc;
  if (!scratch || C > 65535) return IAS_ERR_ARG;
  const int Ho = ias_conv_out_size(H, K, S), Wo = ias_conv_out_size(W, K, S);
  int nchunk = B < 32 ? B : 32;
  const int rows = (B + nchunk - 1) / nchunk;
  nchunk = (B + rows - 1) / rows;
  const dim3 grid(C, nchunk), block(CV_THREADS);
  CV_DISPATCH(dwconv_bwd_weight_kernel, grid, block, 0, (hipStream_t)stream_, x, g, scratch, B, C, H, W, Ho, Wo, rows);
  const int n = C * K * K;
  hipLaunchKernelGGL(conv_reduce_partials_kernel, dim3((n + CV_THREADS - 1) / CV_THREADS), dim3(CV_THREADS), 0,
                     (hipStream_t)stream_, scratch, gw, n, nchunk);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// Stem Conv2d(3, 16, 3, stride 2, padding 1, bias=False): x [B,3,H,W], w [16,3,3,3] -> out [B,16,Ho,Wo]
extern "C" int ias_stem_forward(const float* x, const float* w, float* out, int B, int H, int W, void* stream_) {
  if (!x || !w || !out || B <= 0 || B > 65535 || H <= 0 || W <= 0) return IAS_ERR_ARG;
  const int Ho = ias_conv_out_size(H, 3, 2), Wo = ias_conv_out_size(W, 3, 2);
  hipLaunchKernelGGL((stem_fwd_kernel<3, 16>), dim3(cv_grid_x(Ho * Wo), B), dim3(CV_THREADS), 0, (hipStream_t)stream_, x, w,
                     out, H, W, Ho, Wo);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

#define STEM_CHUNKS_X 8
extern "C" long long ias_stem_weight_scratch(int B) { return B <= 0 ? IAS_ERR_ARG : (long long)B * STEM_CHUNKS_X * 432; }

// its weight gradient: x [B,3,H,W], g [B,16,Ho,Wo] -> gw [16,3,3,3]; scratch: ias_stem_weight_scratch(B) floats
extern "C" int ias_stem_backward_weight(const float* x, const float* g, float* gw, float* scratch, int B, int H, int W,
                                        void* stream_) {
  if (!x || !g || !gw || !scratch || B <= 0 || B > 65535 || H <= 0 || W <= 0) return IAS_ERR_ARG;
  const int Ho = ias_conv_out_size(H, 3, 2), Wo = ias_conv_out_size(W, 3, 2);
  hipLaunchKernelGGL((stem_bwd_weight_kernel<3, 16>), dim3(STEM_CHUNKS_X, B), dim3(CV_THREADS), 0, (hipStream_t)stream_, x, g,
                     scratch, H, W, Ho, Wo, STEM_CHUNKS_X);
  hipLaunchKernelGGL(conv_reduce_partials_kernel, dim3((432 + CV_THREADS - 1) / CV_THREADS), dim3(CV_THREADS), 0,
                     (hipStream_t)stream_, scratch, gw, 432, B * STEM_CHUNKS_X);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
