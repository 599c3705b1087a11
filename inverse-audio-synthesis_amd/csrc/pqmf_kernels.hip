// PQMF analysis / synthesis filterbank for MI355X (gfx950).
//
// Replaces /root/reference/pqmf.py:49-55:
//   analysis : F.conv1d(x[B,1,T], H[N,1,K], padding=K//2... (taps//2), stride=N)
//   synthesis: F.conv_transpose1d(z, updown*N, stride=N) then F.conv1d(., G[1,N,K], padding=taps//2)
// and, optionally fused into the analysis epilogue, the per-band normalisation of
// /root/reference/audioembed.py:41,49 ((z - mean_c) / std_c, torchvision Normalize).
//
// Fast path (N=3 or 4, K=63 -- N=3 is the reference's live configuration,
// vicreg_audio_params.py:40): polyphase FIR out of LDS, see pqmf_analysis_fast_kernel.
// Algorithmic HBM bytes: 4 B in + 4 B out per audio sample.
#include "ias_common.h"

#define PQ_THREADS 256
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Register state of one lane of the polyphase FIR: accumulator pairs and the two pair windows.
template <int N, int K>
struct PqmfWindow {
  f32x2 accp[2][N];
  f32x2 wa[N][4], wb[N][4];
  const float* rows;
  int mrow, m0;

  __device__ __forceinline__ float xat(int p, int m) const { return rows[p * mrow + m + (m >> 5)]; }

  __device__ __forceinline__ void init(const float* rows_, int mrow_, int m_first) {
    rows = rows_; mrow = mrow_; m0 = m_first;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < N; ++k) accp[h][k] = (f32x2){0.0f, 0.0f};
#pragma unroll
    for (int p = 0; p < N; ++p) {
      float v[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) v[i] = xat(p, m0 + i);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wa[p][i] = (f32x2){v[2 * i], v[2 * i + 1]};
        wb[p][i] = (f32x2){v[2 * i + 1], v[2 * i + 2]};
      }
    }
  }

  // One group of 4 polyphase steps q = 4*q4 + s4.  ROT: physical slot of logical pair i is (i+ROT)&3.
  // FULL: every tap index of the group is < K (no checks); otherwise steps/taps beyond K are skipped
  // at compile time (q4 is then a compile-time-known value K / (4N)).
  template <int ROT, bool FULL>
  __device__ __forceinline__ void group(const float* __restrict__ H, int q4) {
    constexpr int QF = K / (4 * N);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
      for (int p = 0; p < N; ++p) {
        const int jc = (4 * QF + s4) * N + p;            // tap index when !FULL (compile time)
        if (!FULL && jc >= K) continue;
        const int j = FULL ? (4 * q4 + s4) * N + p : jc;  // wave-uniform -> scalar loads
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const float hk = H[k * K + j];
          const f32x2 h2 = (f32x2){hk, hk};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int slot = ((s4 >> 1) + h + ROT) & 3;
            const f32x2 xw = (s4 & 1) ? wb[p][slot] : wa[p][slot];
            accp[h][k] = __builtin_elementwise_fma(xw, h2, accp[h][k]);
          }
        }
      }
    }
    if (FULL) {
      // slide by 4 positions: logical pairs 2,3 become 0,1 (same registers, ROT advances by 2 in the
      // caller); the freed slots take positions m0+8 .. m0+11 (+12 for the odd-offset window)
      const int mn = m0 + 4 * q4 + 8;
#pragma unroll
      for (int p = 0; p < N; ++p) {
        const float n0 = xat(p, mn), n1 = xat(p, mn + 1), n2 = xat(p, mn + 2), n3 = xat(p, mn + 3),
                    n4 = xat(p, mn + 4);
        wa[p][(0 + ROT) & 3] = (f32x2){n0, n1}; wa[p][(1 + ROT) & 3] = (f32x2){n2, n3};
        wb[p][(0 + ROT) & 3] = (f32x2){n1, n2}; wb[p][(1 + ROT) & 3] = (f32x2){n3, n4};
      }
    }
  }
};

// Polyphase form: with j = N*q + p, z_k[f] = sum_p sum_q H_k[N*q+p] * x_p[f+q], x_p[m] = x[N*m + p - pad].
// The workgroup de-interleaves its input span into the N polyphase rows in LDS; a lane owns R = 4
// adjacent frames and slides a 4-value register window along each row (one new ds_read_b32 per
// (q, p) step feeds 4 frames x N bands = 12 FMAs), so registers stay ~50/lane and LDS reads are
// 1/10 of the FMA count.  Taps are wave-uniform scalar-cache loads.
template <int N, int K>
__global__ __launch_bounds__(PQ_THREADS) void pqmf_analysis_fast_kernel(
    const float* __restrict__ x, const float* __restrict__ H, float* __restrict__ z,
    const float* __restrict__ mean, const float* __restrict__ stdv, int T, int L, int pad) {
  constexpr int R = 4;
  constexpr int FT = PQ_THREADS * R;            // frames per workgroup
  constexpr int Q = (K + N - 1) / N;            // taps per polyphase branch
  constexpr int QP = (Q + 3) / 4 * 4;           // padded to whole groups of 4 steps (zero taps)
  constexpr int MLEN = FT + QP + 12;            // polyphase positions per row (incl. window look-ahead)
  // x_p[m] lives at index m + (m >> 5): lanes read m = 4*lane + c, and the extra +1 per 32 positions
  // spreads the 32 lanes of a group over all 32 banks (plain m: 4-way conflict).
  constexpr int MROW = ((MLEN + (MLEN >> 5) + 31) / 32) * 32 + 11;   // row stride: rows start 11 banks apart
  constexpr int SPAN = N * MLEN;                // input samples staged per workgroup (zeros past T)
  static_assert(N <= 4, "tap table holds up to 4 bands per 16-byte entry");
  __shared__ float s_xp[N][MROW];

  const int b = blockIdx.y, tid = threadIdx.x;
  const int f_tile = blockIdx.x * FT;
  const long long start = (long long)f_tile * N - pad;   // sample index of polyphase position (m=0, p=0)
  const float* xr = x + (size_t)b * T;

  // stage: aligned 16-byte global loads, de-interleaved dword LDS writes
  const long long g0 = start >= 0 ? (start & ~3LL) : -(((-start) + 3) & ~3LL);
  const int nvec = (int)((start + SPAN - g0 + 3) / 4);
  const bool vec_ok = (T & 3) == 0;
  for (int v = tid; v < nvec; v += PQ_THREADS) {
    const long long g = g0 + 4LL * v;
    float e[4];
    if (vec_ok && g >= 0 && g + 3 < T) {
      const float4 q4 = *reinterpret_cast<const float4*>(xr + g);
      e[0] = q4.x; e[1] = q4.y; e[2] = q4.z; e[3] = q4.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = (g + i >= 0 && g + i < T) ? xr[g + i] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int li = (int)(g + i - start);
      if (li >= 0 && li < SPAN) { const int m = li / N; s_xp[li % N][m + (m >> 5)] = e[i]; }
    }
  }
  __syncthreads();

  // gfx950 issues a plain fp32 VALU op for a wave64 in 4 cycles; only v_pk_fma_f32 (two FMAs per
  // lane) reaches the fp32 peak.  Frames are therefore processed as even-aligned register PAIRS:
  //   accp[h][k] = (acc[2h][k], acc[2h+1][k]),  h = 0, 1
  //   wa[p][i] = (x_p[m0 + 2i], x_p[m0 + 2i + 1])      pairs starting at even offsets
  //   wb[p][i] = (x_p[m0 + 2i + 1], x_p[m0 + 2i + 2])  pairs starting at odd offsets
  // with m0 = 4*tid + 4*q4 (logical index i; the physical slot is (i + ROT) & 3 so that sliding the
  // window by 4 positions moves no registers).  At step s4 (q = 4*q4 + s4) frame pair h needs
  // x_p[m0 + s4 + 2h + {0,1}]: wa[(s4 >> 1) + h] for even s4, wb[(s4 >> 1) + h] for odd s4.
  PqmfWindow<N, K> win;
  win.init(&s_xp[0][0], MROW, R * tid);
  constexpr int QF = K / (4 * N);        // groups of 4 steps in which every tap index is < K
#pragma unroll 1
  for (int q4 = 0; q4 + 1 < QF; q4 += 2) {
    win.template group<0, true>(H, q4);
    win.template group<2, true>(H, q4 + 1);
  }
  if (QF & 1) win.template group<0, true>(H, QF - 1);
  if (4 * QF < Q) {
    // remaining steps (fewer than a group, tap indices checked at compile time)
    if (QF & 1) win.template group<2, false>(H, QF); else win.template group<0, false>(H, QF);
  }
  f32x2 (&accp)[2][N] = win.accp;

  float acc[R][N];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < N; ++k) { acc[2 * h][k] = accp[h][k].x; acc[2 * h + 1][k] = accp[h][k].y; }

  const int f0 = f_tile + tid * R;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float o[R];
#pragma unroll
    for (int r = 0; r < R; ++r) o[r] = acc[r][k];
    if (mean != nullptr) {
      const float m = mean[k], sd = stdv[k];
#pragma unroll
      for (int r = 0; r < R; ++r) o[r] = (o[r] - m) / sd;
    }
    float* zr = z + ((size_t)b * N + k) * L;
    if ((L & 3) == 0 && f0 + 3 < L) {
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
    hipLaunchKernelGGL((pqmf_analysis_fast_kernel<3, 63>), dim3((L + FT - 1) / FT, B), dim3(PQ_THREADS), 0,
                       stream, x, H, z, mean, stdv, T, L, pad);
  } else if (N == 4 && K == 63) {
    constexpr int FT = PQ_THREADS * 4;
    hipLaunchKernelGGL((pqmf_analysis_fast_kernel<4, 63>), dim3((L + FT - 1) / FT, B), dim3(PQ_THREADS), 0,
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
