// PQMF analysis / synthesis filterbank for MI355X (gfx950).
//
// Replaces /root/reference/pqmf.py:49-55:
//   analysis : F.conv1d(x[B,1,T], H[N,1,K], padding=K//2... (taps//2), stride=N)
//   synthesis: F.conv_transpose1d(z, updown*N, stride=N) then F.conv1d(., G[1,N,K], padding=taps//2)
// and, optionally fused into the analysis epilogue, the per-band normalisation of
// /root/reference/audioembed.py:41,49 ((z - mean_c) / std_c, torchvision Normalize).
//
// Fast path (N=3, K=63 -- the reference's live configuration, vicreg_audio_params.py:40):
// a workgroup stages (FT-1)*N+K input samples in LDS once, every lane produces 4
// consecutive frames of all 3 bands from a 72-sample register window (18 ds_read_b128),
// filter taps come through the scalar cache, every store is 16 B/lane (1 KiB per wave).
// Algorithmic HBM bytes: 4 B in + 4 B out per audio sample.
#include "ias_common.h"

#define PQ_THREADS 256

template <int N, int K, int R>
__global__ __launch_bounds__(PQ_THREADS) void pqmf_analysis_fast_kernel(
    const float* __restrict__ x, const float* __restrict__ H, float* __restrict__ z,
    const float* __restrict__ mean, const float* __restrict__ stdv, int T, int L, int pad) {
  constexpr int FT = PQ_THREADS * R;            // frames per workgroup
  constexpr int SPAN = (FT - 1) * N + K;        // input samples per workgroup
  constexpr int WIN = (R - 1) * N + K;          // input samples per lane
  constexpr int WIN4 = (WIN + 3) / 4;
  constexpr int LDS_FLOATS = ((PQ_THREADS - 1) * R * N + WIN4 * 4 + 3) / 4 * 4;
  static_assert((R * N) % 4 == 0, "lane window must start 16-byte aligned");
  __shared__ __attribute__((aligned(16))) float s_x[LDS_FLOATS];

  const int b = blockIdx.y, tid = threadIdx.x;
  const int f_tile = blockIdx.x * FT;
  const long long start = (long long)f_tile * N - pad;   // sample index of s_x[0]
  const float* xr = x + (size_t)b * T;

  // stage: aligned 16-byte global loads, shifted dword LDS writes
  const long long g0 = start >= 0 ? (start & ~3LL) : -(((-start) + 3) & ~3LL);
  const int nvec = (int)((start + LDS_FLOATS - g0 + 3) / 4);
  const bool vec_ok = (T & 3) == 0;
  for (int v = tid; v < nvec; v += PQ_THREADS) {
    const long long g = g0 + 4LL * v;
    float e[4];
    if (vec_ok && g >= 0 && g + 3 < T) {
      const float4 q = *reinterpret_cast<const float4*>(xr + g);
      e[0] = q.x; e[1] = q.y; e[2] = q.z; e[3] = q.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = (g + i >= 0 && g + i < T) ? xr[g + i] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long long li = g + i - start;
      if (li >= 0 && li < LDS_FLOATS) s_x[li] = e[i];
    }
  }
  __syncthreads();

  float acc[R][N];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int k = 0; k < N; ++k) acc[r][k] = 0.0f;

  const float4* win = reinterpret_cast<const float4*>(s_x + tid * (R * N));
#pragma unroll
  for (int v = 0; v < WIN4; ++v) {
    const float4 q = win[v];
    const float xv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = v * 4 + e;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int j = i - r * N;
        if (j >= 0 && j < K) {
#pragma unroll
          for (int k = 0; k < N; ++k) acc[r][k] = fmaf(xv[e], H[k * K + j], acc[r][k]);
        }
      }
    }
  }

  const int f0 = f_tile + tid * R;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float o[R];
#pragma unroll
    for (int r = 0; r < R; ++r) o[r] = acc[r][k];
    if (mean != nullptr) {
      const float m = mean[k], s = stdv[k];
#pragma unroll
      for (int r = 0; r < R; ++r) o[r] = (o[r] - m) / s;
    }
    float* zr = z + ((size_t)b * N + k) * L;
    if (R == 4 && (L & 3) == 0 && f0 + 3 < L) {
      *reinterpret_cast<float4*>(zr + f0) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) if (f0 + r < L) zr[f0 + r] = o[r];
    }
  }
}

// Generic analysis (any N, K): one lane per output element.
__global__ __launch_bounds__(PQ_THREADS) void pqmf_analysis_generic_kernel(
    const float* __restrict__ x, const float* __restrict__ H, float* __restrict__ z,
    const float* __restrict__ mean, const float* __restrict__ stdv, int T, int L, int N, int K, int pad) {
  const int f = blockIdx.x * PQ_THREADS + threadIdx.x;
  const int k = blockIdx.y, b = blockIdx.z;
  if (f >= L) return;
  const float* xr = x + (size_t)b * T;
  const float* h = H + (size_t)k * K;
  const long long s = (long long)f * N - pad;
  float acc = 0.0f;
  for (int j = 0; j < K; ++j) {
    const long long i = s + j;
    if (i >= 0 && i < T) acc = fmaf(xr[i], h[j], acc);
  }
  if (mean != nullptr) acc = (acc - mean[k]) / stdv[k];
  z[((size_t)b * N + k) * L + f] = acc;
}

// Synthesis in polyphase form (the zero-stuffed [B,N,L*N] tensor is never built):
// out[b,t] = N * sum_k sum_{j : (t-pad+j) % N == 0} G[k,j] * z[b,k,(t-pad+j)/N],  t in [0, L*N)
__global__ __launch_bounds__(PQ_THREADS) void pqmf_synthesis_kernel(
    const float* __restrict__ z, const float* __restrict__ G, float* __restrict__ out,
    int L, int N, int K, int pad) {
  const int To = L * N;
  const int t = blockIdx.x * PQ_THREADS + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= To) return;
  // first tap j0 >= 0 with (t - pad + j0) % N == 0
  int rem = (t - pad) % N;
  if (rem < 0) rem += N;
  const int j0 = (N - rem) % N;
  float acc = 0.0f;
  for (int k = 0; k < N; ++k) {
    const float* zr = z + ((size_t)b * N + k) * L;
    const float* g = G + (size_t)k * K;
    float a = 0.0f;
    for (int j = j0; j < K; j += N) {
      const int u = t - pad + j;          // position in the zero-stuffed signal
      if (u >= 0 && u < To) a = fmaf(g[j], zr[u / N] * (float)N, a);
    }
    acc += a;
  }
  out[(size_t)b * To + t] = acc;
}

// ------------------------------------------------------------------------ C ABI
extern "C" int ias_pqmf_out_len(int T, int N, int K) {
  const int pad = (K - 1) / 2;
  if (T <= 0 || N <= 0 || K <= 0 || T + 2 * pad < K) return IAS_ERR_ARG;
  return (T + 2 * pad - K) / N + 1;
}

// x [B,T] (the reference's [B,1,T]), H [N,K] (module buffer H[N,1,K]), z [B,N,L].
// mean/stdv: optional device pointers [N] (both or neither) for the fused
// AudioEmbedding._preprocess normalisation.
extern "C" int ias_pqmf_analysis(const float* x, const float* H, float* z, const float* mean, const float* stdv,
                                 int B, int T, int N, int K, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !H || !z || B <= 0 || B > 65535 || N <= 0 || N > 65535 || K <= 0 || (K & 1) == 0) return IAS_ERR_ARG;
  if ((mean == nullptr) != (stdv == nullptr)) return IAS_ERR_ARG;
  const int pad = (K - 1) / 2;  // == taps // 2 for K = taps + 1, taps even
  const int L = ias_pqmf_out_len(T, N, K);
  if (L <= 0) return IAS_ERR_ARG;
  if (N == 3 && K == 63) {
    constexpr int FT = PQ_THREADS * 4;
    hipLaunchKernelGGL((pqmf_analysis_fast_kernel<3, 63, 4>), dim3((L + FT - 1) / FT, B), dim3(PQ_THREADS), 0,
                       stream, x, H, z, mean, stdv, T, L, pad);
  } else if (N == 4 && K == 63) {
    constexpr int FT = PQ_THREADS * 4;
    hipLaunchKernelGGL((pqmf_analysis_fast_kernel<4, 63, 4>), dim3((L + FT - 1) / FT, B), dim3(PQ_THREADS), 0,
                       stream, x, H, z, mean, stdv, T, L, pad);
  } else {
    hipLaunchKernelGGL(pqmf_analysis_generic_kernel, dim3((L + PQ_THREADS - 1) / PQ_THREADS, N, B),
                       dim3(PQ_THREADS), 0, stream, x, H, z, mean, stdv, T, L, N, K, pad);
  }
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// z [B,N,L], G [N,K] (module buffer G[1,N,K]), out [B, L*N] (the reference's [B,1,L*N]).
extern "C" int ias_pqmf_synthesis(const float* z, const float* G, float* out, int B, int L, int N, int K,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!z || !G || !out || B <= 0 || B > 65535 || L <= 0 || N <= 0 || K <= 0 || (K & 1) == 0) return IAS_ERR_ARG;
  const int pad = (K - 1) / 2;
  const long long To = (long long)L * N;
  if (To > 0x7fffffffLL) return IAS_ERR_ARG;
  hipLaunchKernelGGL(pqmf_synthesis_kernel, dim3((int)((To + PQ_THREADS - 1) / PQ_THREADS), B), dim3(PQ_THREADS), 0,
                     stream, z, G, out, L, N, K, pad);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
