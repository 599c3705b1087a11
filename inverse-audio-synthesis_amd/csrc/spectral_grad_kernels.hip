// Backward of the L1 spectral losses for MI355X (gfx950): d mean|V(audio) - target| / d audio, V = (mel of) the
// power or magnitude STFT.
//
// The reference never differentiates a spectral loss -- its mel-L1 training loop is the commented-out block
// /root/reference/audio_to_params.py:150-153 over the MelSpectrogram of conf/config.yaml:51-61 (torchaudio modules,
// differentiable torch code).  This is the adjoint of csrc/spectral_kernels.hip's forward for LOSS_L1
// (the audio -> params -> synth -> mel-L1 loop).
//
//   frame   x_f[n] = w[n] * pad_reflect(a)[f*hop + n]                 X_f = DFT_N(x_f), bins 0..N/2
//   values  P = |X|^2,  V = P (power 2) or sqrt(P) (power 1),  O = melW^T V or V
//   loss    L = scale * sum |O - target|
//   adjoint gO = scale * sign(O - target);  gV = melW gO or gO;  gP = gV (power) or gV / (2 sqrt P);
//           G[k] = 2 gP[k] X[k];  g x_f[n] = Re sum_{k<=N/2} G[k] e^{+2 pi i k n / N}   (upper half zero, no Hermitian
//           doubling: each one-sided bin enters the loss once);  g a = reflect-pad^T overlap-add (w * g x_f)
//
// K_a one workgroup per PAIR of frames (one complex FFT carries two real frames in each direction): forward FFT,
//     the adjoint chain per frame, inverse FFT -> frame_grad [B,F,N] (scratch).
// K_b one lane per audio sample: gathers the frames (and reflected positions) that cover it in a fixed order --
//     no atomics across workgroups, bit-reproducible.
// The FFT is a plain Stockham radix-4 through LDS (natural order in and out, twiddles from an LDS table built with
// sincospif); the backward is a first correct path, not yet tuned like the forward.
#include "ias_common.h"
#include <cstdlib>

#define SG_THREADS 256
#define SG_PAIRS 1         // frame pairs per workgroup (twiddle table and filterbank staged once)
typedef float2 sg_cpx;

__device__ __forceinline__ sg_cpx sg_mul(sg_cpx a, sg_cpx b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// Stockham autosort FFT of N points held in `a` (scratch `b`), natural order in and out; returns the buffer that
// holds the result.  Radix-4 stages while the remaining length allows, one radix-2 stage when log2(N) is odd.
// SIGN -1: forward e^{-i..}, +1: inverse (unnormalised).  tab[j] = e^{-2 pi i j / N}, j < N/2.
template <int SIGN>
__device__ sg_cpx* sg_fft(sg_cpx* a, sg_cpx* b, const sg_cpx* tab, int N, int log2n, int tid) {
  sg_cpx* x = a;
  sg_cpx* y = b;
  int st = 0;                          // log2 of the stride s; sequence length n = N >> st
  for (; st + 2 <= log2n; st += 2) {
    const int s = 1 << st, m = N >> (st + 2);
    __syncthreads();
    for (int t = tid; t < N / 4; t += SG_THREADS) {
      const int p = t >> st, q = t & (s - 1);
      sg_cpx w1 = tab[p * s], w2 = tab[2 * p * s];
      if (SIGN > 0) { w1.y = -w1.y; w2.y = -w2.y; }
      const sg_cpx w3 = sg_mul(w1, w2);
      const sg_cpx c0 = x[q + s * p], c1 = x[q + s * (p + m)], c2 = x[q + s * (p + 2 * m)], c3 = x[q + s * (p + 3 * m)];
      const sg_cpx apc = make_float2(c0.x + c2.x, c0.y + c2.y), amc = make_float2(c0.x - c2.x, c0.y - c2.y);
      const sg_cpx bpd = make_float2(c1.x + c3.x, c1.y + c3.y), bmd = make_float2(c1.x - c3.x, c1.y - c3.y);
      // forward: -i (b - d) = (bmd.y, -bmd.x); inverse: +i (b - d) = (-bmd.y, bmd.x)
      const sg_cpx jb = SIGN < 0 ? make_float2(bmd.y, -bmd.x) : make_float2(-bmd.y, bmd.x);
      y[q + s * (4 * p)] = make_float2(apc.x + bpd.x, apc.y + bpd.y);
      y[q + s * (4 * p + 1)] = sg_mul(make_float2(amc.x + jb.x, amc.y + jb.y), w1);
      y[q + s * (4 * p + 2)] = sg_mul(make_float2(apc.x - bpd.x, apc.y - bpd.y), w2);
      y[q + s * (4 * p + 3)] = sg_mul(make_float2(amc.x - jb.x, amc.y - jb.y), w3);
    }
    sg_cpx* tmp = x; x = y; y = tmp;
  }
  if (st < log2n) {                    // last stage, n = 2: no twiddles
    const int s = 1 << st;
    __syncthreads();
    for (int t = tid; t < N / 2; t += SG_THREADS) {
      const sg_cpx u = x[t], v = x[t + s];
      y[t] = make_float2(u.x + v.x, u.y + v.y);
      y[t + s] = make_float2(u.x - v.x, u.y - v.y);
    }
    sg_cpx* tmp = x; x = y; y = tmp;
  }
  __syncthreads();
  return x;
}

struct SgArgs {
  const float* audio;     // [B,T]
  const float* window;    // [N]
  const int* mel_start;   // CSR mel filters (NULL: linear bins)
  const int* mel_count;
  const int* mel_woff;
  const float* mel_w;
  const float* target;    // [B,F,n_out]
  float* frame_grad;      // [B,F,N]
  int T, F, N, log2n, hop, n_out, power2;
  int mel_nnz;            // entries of mel_w (0 without a mel projection)
  int loss_mode;          // 1: scale * sum |V - t|;  2: MR-STFT term, V = sqrt(max(|X|^2, eps))
  const double* coef;     // loss_mode 2: device [2] = {c0, c1}: gO = c0 (V - t) + c1 sign(V - t) / V
  float scale, eps;
};

// Two frames per workgroup share each FFT: z = x_a + i x_b has Z[k] = X_a[k] + i X_b[k], so
//   X_a[k] = (Z[k] + conj Z[N-k]) / 2,   X_b[k] = (Z[k] - conj Z[N-k]) / (2i),
// and on the way back the two REAL gradient frames come out of one inverse transform of H_a + i H_b, where H_r is
// the Hermitian extension whose inverse DFT is y_r[n] = Re sum_{k<=N/2} G_r[k] e^{+i theta}:
//   H_r[k] = G_r[k] / 2 = gP_r[k] X_r[k] (0 < k < N/2),  H_r[N-k] = conj H_r[k],  H_r[0] = G_r[0],  H_r[N/2] = G_r[N/2].
__device__ __forceinline__ sg_cpx sg_frame_bin(const sg_cpx* Z, int k, int N, int which) {
  const sg_cpx z = Z[k], zc = Z[(N - k) & (N - 1)];
  // which 0: (z + conj zc)/2;  1: (z - conj zc)/(2i) = ((z.y + zc.y)/2, -(z.x - zc.x)/2)
  return which == 0 ? make_float2(0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y))
                    : make_float2(0.5f * (z.y + zc.y), -0.5f * (z.x - zc.x));
}

__global__ __launch_bounds__(SG_THREADS) void stft_grad_frames_kernel(SgArgs g) {
  extern __shared__ __attribute__((aligned(16))) float sg_smem[];
  const int N = g.N, NB = N / 2 + 1, tid = threadIdx.x, b = blockIdx.y;
  sg_cpx* bufa = reinterpret_cast<sg_cpx*>(sg_smem);
  sg_cpx* bufb = bufa + N;
  sg_cpx* tab = bufb + N;                    // N/2
  float* sP = reinterpret_cast<float*>(tab + N / 2);   // [2][NB + 3]: values V of the two frames
  float* sGV = sP + 2 * (NB + 3);            // [2][NB + 3]: d loss / d V
  float* sGO = sGV + 2 * (NB + 3);           // [2][n_out (+ pad)]: d loss / d output
  float* sGP = sGO + 2 * ((g.n_out + 3) & ~3);   // [2][NB + 3]: d loss / d |X|^2
  // the filterbank (CSR) is copied to LDS once per workgroup: the projection loops below are chains of dependent
  // reads, and out of global memory each link costs an L2 round trip
  float* sMW = sGP + 2 * (NB + 3);           // [mel_nnz]
  int* sMI = reinterpret_cast<int*>(sMW + ((g.mel_nnz + 3) & ~3));   // [3][n_out]: start, count, woff
  if (g.mel_start) {
    for (int i = tid; i < g.mel_nnz; i += SG_THREADS) sMW[i] = g.mel_w[i];
    for (int i = tid; i < g.n_out; i += SG_THREADS) {
      sMI[i] = g.mel_start[i]; sMI[g.n_out + i] = g.mel_count[i]; sMI[2 * g.n_out + i] = g.mel_woff[i];
    }
  }

  const int pad = N / 2;
  const float* arow = g.audio + (size_t)b * g.T;
  for (int j = tid; j < N / 2; j += SG_THREADS) {
    float s, c;
    sincospif(2.0f * (float)j / (float)N, &s, &c);
    tab[j] = make_float2(c, -s);
  }
  // a workgroup walks through SG_PAIRS consecutive frame pairs with the tables above in place
  for (int pp = 0; pp < SG_PAIRS; ++pp) {
  const int fa = 2 * (blockIdx.x * SG_PAIRS + pp);   // frames fa and fa + 1 (the second may not exist)
  if (fa >= g.F) break;
  const int nfr = min(2, g.F - fa);
  __syncthreads();                                   // previous pair done with the buffers
  for (int n = tid; n < N; n += SG_THREADS) {
    float v[2] = {0.0f, 0.0f};
    for (int r = 0; r < nfr; ++r) {
      int q = (fa + r) * g.hop + n - pad;
      if (q < 0) q = -q;
      if (q >= g.T) q = 2 * (g.T - 1) - q;
      v[r] = arow[q];
    }
    const float w = g.window[n];
    bufa[n] = make_float2(w * v[0], w * v[1]);
  }
  sg_cpx* Z = sg_fft<-1>(bufa, bufb, tab, N, g.log2n, tid);
  sg_cpx* other = (Z == bufa) ? bufb : bufa;

  // The value -> loss -> adjoint chain of BOTH frames at once: frame r uses row r of sP / sGV / sGO / sGP, and the
  // projection phases run one thread per (frame, output) -- all 256 threads for 2 x 128 mel bands.
  const int SB = NB + 3, SO = (g.n_out + 3) & ~3;
  for (int i = tid; i < 2 * NB; i += SG_THREADS) {
    const int r = i >= NB, k = i - r * NB;
    float v = 0.0f;
    if (r < nfr) {
      const sg_cpx x = sg_frame_bin(Z, k, N, r);
      const float p = x.x * x.x + x.y * x.y;
      v = g.power2 ? p : sqrtf(g.loss_mode == 2 ? fmaxf(p, g.eps) : p);         // V
    }
    sP[r * SB + k] = v;
    sGV[r * SB + k] = 0.0f;
  }
  __syncthreads();
  for (int i = tid; i < 2 * g.n_out; i += SG_THREADS) {
    const int r = i >= g.n_out, o = i - r * g.n_out;
    float go = 0.0f;
    if (r < nfr) {
      const float* pr = sP + r * SB;
      float v;
      if (g.mel_start) {
        const int s0 = sMI[o], cnt = sMI[g.n_out + o];
        const float* w = sMW + sMI[2 * g.n_out + o];
        v = 0.0f;
        for (int c = 0; c < cnt; ++c) v = fmaf(w[c], pr[s0 + c], v);
      } else {
        v = pr[o];
      }
      const float d = v - g.target[((size_t)b * g.F + fa + r) * g.n_out + o];
      const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
      // MR-STFT: spectral convergence ||T - V||_F / ||T||_F and mean |log V - log T| (log is monotone: sign(V - T))
      go = g.loss_mode == 2 ? (float)g.coef[0] * d + (float)g.coef[1] * sg / v : sg * g.scale;
    }
    sGO[r * SO + o] = go;
  }
  __syncthreads();
  if (g.mel_start) {
    // a bin lies under at most two triangular filters: the (commutative) sum of two terms does not depend on the
    // order of the LDS atomics
    for (int i = tid; i < 2 * g.n_out; i += SG_THREADS) {
      const int r = i >= g.n_out, o = i - r * g.n_out;
      const int s0 = sMI[o], cnt = sMI[g.n_out + o];
      const float* w = sMW + sMI[2 * g.n_out + o];
      const float go = sGO[r * SO + o];
      for (int c = 0; c < cnt; ++c) atomicAdd(&sGV[r * SB + s0 + c], w[c] * go);
    }
  } else {
    for (int i = tid; i < 2 * NB; i += SG_THREADS) { const int r = i >= NB, k = i - r * NB; sGV[r * SB + k] = sGO[r * SO + k]; }
  }
  __syncthreads();
  for (int i = tid; i < 2 * NB; i += SG_THREADS) {
    const int r = i >= NB, k = i - r * NB;
    float v = sGV[r * SB + k];
    if (!g.power2) {
      // d sqrt(P) / dP, 0 at P = 0; MR-STFT clamps P at eps first (clamp passes the gradient at P >= eps)
      const float pv = sP[r * SB + k];
      const bool live = g.loss_mode == 2 ? pv > sqrtf(g.eps) : pv > 0.0f;   // V == sqrt(eps): clamped bin
      v = live ? v / (2.0f * pv) : 0.0f;
    }
    sGP[r * SB + k] = v;
  }
  __syncthreads();
  // inverse input H_a + i H_b (see above), natural order in `other`
  for (int k = tid; k < N; k += SG_THREADS) {
    const int kk = k <= N / 2 ? k : N - k;                   // the one-sided bin this entry mirrors
    const bool edge = kk == 0 || kk == N / 2;
    sg_cpx h[2];
    for (int r = 0; r < 2; ++r) {
      const sg_cpx x = sg_frame_bin(Z, kk, N, r);
      const float gpv = sGP[r * (NB + 3) + kk] * (edge ? 2.0f : 1.0f);
      h[r] = make_float2(gpv * x.x, edge ? 0.0f : (k <= N / 2 ? gpv * x.y : -gpv * x.y));
    }
    other[k] = make_float2(h[0].x - h[1].y, h[0].y + h[1].x);   // H_a + i H_b
  }
  sg_cpx* Y = sg_fft<+1>(other, Z, tab, N, g.log2n, tid);
  for (int r = 0; r < nfr; ++r) {
    float* out = g.frame_grad + ((size_t)b * g.F + fa + r) * N;
    for (int n = tid; n < N; n += SG_THREADS) out[n] = g.window[n] * (r == 0 ? Y[n].x : Y[n].y);
  }
  }   // frame pairs
}

// g_audio[b,j] = g_loss * sum over the padded positions q that read audio[j] (itself and its reflections) of the
// frames covering q.
__global__ __launch_bounds__(SG_THREADS) void stft_grad_ola_kernel(const float* __restrict__ frame_grad,
                                                                   const float* __restrict__ g_loss,
                                                                   float* __restrict__ g_audio, int T, int F, int N,
                                                                   int hop) {
  const int j = blockIdx.x * SG_THREADS + threadIdx.x, b = blockIdx.y;
  if (j >= T) return;
  const int pad = N / 2;
  const float* fg = frame_grad + (size_t)b * F * N;
  int qs[3];
  int nq = 0;
  qs[nq++] = j + pad;
  if (j >= 1 && j <= pad) qs[nq++] = pad - j;
  if (j <= T - 2 && j >= T - 1 - pad) qs[nq++] = pad + 2 * (T - 1) - j;
  float acc = 0.0f;
  for (int i = 0; i < nq; ++i) {
    const int q = qs[i];
    int f_hi = q / hop;
    if (f_hi > F - 1) f_hi = F - 1;
    int f_lo = q - N + 1 <= 0 ? 0 : (q - N + 1 + hop - 1) / hop;
    for (int f = f_lo; f <= f_hi; ++f) acc += fg[(size_t)f * N + (q - f * hop)];
  }
  g_audio[(size_t)b * T + j] = g_loss ? acc * g_loss[0] : acc;
}

// The same from chunk spans (ias_stft_grad_spans): span c of row b covers the padded samples [c G hop, c G hop + L) with
// the sum over ITS frames; a sample lies in at most two spans (G hop >= N - hop), added lower chunk first.
__global__ __launch_bounds__(SG_THREADS) void stft_grad_combine_kernel(const float* __restrict__ spans,
                                                                       const float* __restrict__ g_loss,
                                                                       float* __restrict__ g_audio, int T, int F, int N,
                                                                       int hop, int G, int cper, int L) {
  const int j = blockIdx.x * SG_THREADS + threadIdx.x, b = blockIdx.y;
  if (j >= T) return;
  const int pad = N / 2, gh = G * hop;
  const float* sp = spans + (size_t)b * cper * L;
  int qs[3];
  int nq = 0;
  qs[nq++] = j + pad;
  if (j >= 1 && j <= pad) qs[nq++] = pad - j;
  if (j <= T - 2 && j >= T - 1 - pad) qs[nq++] = pad + 2 * (T - 1) - j;
  float acc = 0.0f;
  for (int i = 0; i < nq; ++i) {
    const int q = qs[i];
    int c = q / gh;
    if (c > cper - 1) c = cper - 1;
    float v = 0.0f;
    if (c >= 1 && q - (c - 1) * gh < L) v = sp[(size_t)(c - 1) * L + (q - (c - 1) * gh)];
    const int nf = min(F, (c + 1) * G) - c * G;             // frames of chunk c (the last one may be short)
    if (q - c * gh < (nf - 1) * hop + N) v += sp[(size_t)c * L + (q - c * gh)];
    acc += v;
  }
  g_audio[(size_t)b * T + j] = g_loss ? acc * g_loss[0] : acc;
}

// Several resolutions at once (MR-STFT): g_audio = g_loss * sum over the resolutions, in their order, of the above --
// one pass over the audio gradient instead of one per resolution plus the additions.
#define IAS_COMBINE_MAX 8
struct CombineArgs {
  const float* spans[IAS_COMBINE_MAX];
  int N[IAS_COMBINE_MAX], hop[IAS_COMBINE_MAX], G[IAS_COMBINE_MAX], cper[IAS_COMBINE_MAX], L[IAS_COMBINE_MAX], F[IAS_COMBINE_MAX];
  unsigned magic[IAS_COMBINE_MAX];      // floor(2^32 / (G hop)): q / (G hop) without an integer division (3 x nres per sample)
  int nres;
};
// one resolution's share of sample j (overlap-added chunk spans + the reflect-pad adjoint)
__device__ __forceinline__ float combine_one(const CombineArgs& a, int r, const float* __restrict__ sp, int j, int T) {
  const int N = a.N[r], hop = a.hop[r], G = a.G[r], cper = a.cper[r], L = a.L[r], F = a.F[r];
  const int pad = N / 2, gh = G * hop;
  int qs[3];
  int nq = 0;
  qs[nq++] = j + pad;
  if (j >= 1 && j <= pad) qs[nq++] = pad - j;
  if (j <= T - 2 && j >= T - 1 - pad) qs[nq++] = pad + 2 * (T - 1) - j;
  float acc = 0.0f;
  for (int i = 0; i < nq; ++i) {
    const int q = qs[i];
    int c = (int)__umulhi((unsigned)q, a.magic[r]);         // floor(q / gh) or one less
    if (q - c * gh >= gh) ++c;
    if (c > cper - 1) c = cper - 1;
    float v = 0.0f;
    if (c >= 1 && q - (c - 1) * gh < L) v = sp[(size_t)(c - 1) * L + (q - (c - 1) * gh)];
    const int nf = min(F, (c + 1) * G) - c * G;
    if (q - c * gh < (nf - 1) * hop + N) v += sp[(size_t)c * L + (q - c * gh)];
    acc += v;
  }
  return acc;
}

// A thread owns the sample pair (j, j + 1), j even.  Away from the reflected ends of a resolution the pair sits at an even
// offset of one or two chunk spans (hop, G hop, L and n_fft / 2 are all even: a pair never straddles a chunk or a span
// end): 8-byte loads, half the load instructions of the one-sample form.
__global__ __launch_bounds__(SG_THREADS) void stft_grad_combine_multi_kernel(const CombineArgs a,
                                                                             const float* __restrict__ g_loss,
                                                                             float* __restrict__ g_audio, int T) {
  const int j = 2 * (blockIdx.x * SG_THREADS + threadIdx.x), b = blockIdx.y;
  if (j >= T) return;
  const bool two = j + 1 < T;
  float t0 = 0.0f, t1 = 0.0f;
  for (int r = 0; r < a.nres; ++r) {
    const int N = a.N[r], hop = a.hop[r], G = a.G[r], cper = a.cper[r], L = a.L[r], F = a.F[r];
    const int pad = N / 2, gh = G * hop;
    const float* sp = a.spans[r] + (size_t)b * cper * L;
    float v0, v1;
    const bool even = ((hop | L) & 1) == 0 && ((reinterpret_cast<uintptr_t>(sp) & 7) == 0);
    if (two && even && j > pad && j + 1 < T - 1 - pad) {
      const int q = j + pad;
      int c = (int)__umulhi((unsigned)q, a.magic[r]);
      if (q - c * gh >= gh) ++c;
      if (c > cper - 1) c = cper - 1;
      const int off = q - c * gh;
      float2 lo = make_float2(0.0f, 0.0f), hi = make_float2(0.0f, 0.0f);
      if (c >= 1 && off + gh < L) lo = *reinterpret_cast<const float2*>(sp + (size_t)(c - 1) * L + off + gh);
      const int nf = min(F, (c + 1) * G) - c * G;
      if (off < (nf - 1) * hop + N) hi = *reinterpret_cast<const float2*>(sp + (size_t)c * L + off);
      v0 = lo.x + hi.x; v1 = lo.y + hi.y;
    } else {
      v0 = combine_one(a, r, sp, j, T);
      v1 = two ? combine_one(a, r, sp, j + 1, T) : 0.0f;
    }
    t0 = r == 0 ? v0 : t0 + v0;
    t1 = r == 0 ? v1 : t1 + v1;
  }
  const float g = g_loss ? g_loss[0] : 1.0f;
  float* o = g_audio + (size_t)b * T + j;
  if (two && (((size_t)b * T + j) & 1) == 0) *reinterpret_cast<float2*>(o) = make_float2(g_loss ? t0 * g : t0, g_loss ? t1 * g : t1);
  else { o[0] = g_loss ? t0 * g : t0; if (two) o[1] = g_loss ? t1 * g : t1; }
}

// spans_host: HOST array of nres <= 8 device pointers (ias_stft_grad_spans outputs for the same audio [B,T]);
// plans_host: HOST ints [nres][5] = {n_fft, hop, G, chunks per row, floats per span}.
extern "C" int ias_stft_grad_combine(const float* const* spans_host, const int* plans_host, int nres, const float* g_loss,
                                     float* g_audio, int B, int T, void* stream_) {
  if (!spans_host || !plans_host || !g_audio || nres < 1 || nres > IAS_COMBINE_MAX || B <= 0 || B > 65535 || T <= 0)
    return IAS_ERR_ARG;
  CombineArgs a;
  for (int r = 0; r < IAS_COMBINE_MAX; ++r) { a.spans[r] = nullptr; a.N[r] = a.hop[r] = a.G[r] = a.cper[r] = a.L[r] = a.F[r] = 1; a.magic[r] = 0; }
  for (int r = 0; r < nres; ++r) {
    const int* p = plans_host + 5 * r;
    if (!spans_host[r] || p[0] <= 0 || p[1] <= 0 || p[2] <= 0 || p[3] <= 0 || p[4] != (p[2] - 1) * p[1] + p[0]) return IAS_ERR_ARG;
    if (T <= p[0] / 2) return IAS_ERR_ARG;
    a.spans[r] = spans_host[r]; a.N[r] = p[0]; a.hop[r] = p[1]; a.G[r] = p[2]; a.cper[r] = p[3]; a.L[r] = p[4];
    a.F[r] = 1 + T / p[1];
    if ((long long)p[2] * p[1] > 0x7fffffffLL || (long long)T + p[0] > 0x7fffffffLL) return IAS_ERR_ARG;
    a.magic[r] = (long long)p[2] * p[1] == 1 ? 0xFFFFFFFFu : (unsigned)(0x100000000ULL / (unsigned long long)((long long)p[2] * p[1]));
    if (p[3] != (a.F[r] + p[2] - 1) / p[2]) return IAS_ERR_ARG;
  }
  a.nres = nres;
  hipLaunchKernelGGL(stft_grad_combine_multi_kernel, dim3(((T + 1) / 2 + SG_THREADS - 1) / SG_THREADS, B), dim3(SG_THREADS), 0,
                     (hipStream_t)stream_, a, g_loss, g_audio, T);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------ C ABI
// loss_mode 1: d (scale * sum |V(audio) - target|) / d audio, times the device scalar g_loss[0] (NULL = 1).
// loss_mode 2: one resolution of the MR-STFT loss (linear bins, power 1, V = sqrt(max(|X|^2, eps))):
//   d / d audio of  c0' ||T - V||_F^2 / 2 ... given directly through its cotangent gO = coef[0] (V - T) +
//   coef[1] sign(V - T) / V with the device doubles coef[2] (the caller derives them from the forward's sums:
//   coef[0] = g / (sqrt(sum (T-V)^2) sqrt(sum T^2)), coef[1] = g / count, both divided by the number of resolutions).
//   audio [B,T]; window [n_fft] (device); mel_* : the forward's CSR filterbank (NULL = linear bins, n_out = n_fft/2+1);
//   target [B,F,n_out] frames-major (what ias_stft wrote for the target); power: 1 (magnitude) or 2 (power);
//   frame_grad [B,F,n_fft] fp32 scratch; g_audio [B,T] out.  F = ias_stft_num_frames(T, n_fft, hop).
extern "C" int ias_stft_grad_frames(const float* audio, const float* tables, const int* mel_start, const int* mel_count,
                                    const int* mel_woff, const float* mel_w, int mel_nnz, int n_out, const float* target,
                                    const double* coef, float* frame_grad, int B, int T, int n_fft, int hop, int power,
                                    int loss_mode, float scale, float eps, void* stream);   // csrc/spectral_kernels.hip
extern "C" int ias_stft_grad_spans(const float* audio, const float* tables, const int* mel_start, const int* mel_count,
                                   const int* mel_woff, const float* mel_w, int mel_nnz, int n_out, const float* target,
                                   const double* coef, float* chunk_spans, int B, int T, int n_fft, int hop, int power,
                                   int loss_mode, float scale, float eps, int* plan_host, void* stream);

extern "C" int ias_stft_loss_backward(const float* audio, const float* window, const float* tables, const int* mel_start,
                                      const int* mel_count, const int* mel_woff, const float* mel_w, int mel_nnz,
                                      const float* target, const float* g_loss, const double* coef, float* frame_grad,
                                      float* g_audio, int B, int T, int n_fft, int hop, int n_out, int power,
                                      int loss_mode, float scale, float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!audio || !window || !target || !frame_grad || !g_audio || B <= 0 || B > 65535 || hop <= 0) return IAS_ERR_ARG;
  if (n_fft != 512 && n_fft != 1024 && n_fft != 2048) return IAS_ERR_UNSUPPORTED;
  if (power != 1 && power != 2) return IAS_ERR_UNSUPPORTED;
  if (loss_mode != 1 && loss_mode != 2) return IAS_ERR_UNSUPPORTED;
  if (loss_mode == 2 && (!coef || power != 1 || mel_start != nullptr)) return IAS_ERR_ARG;
  if (T <= n_fft / 2) return IAS_ERR_ARG;
  const bool mel = mel_start != nullptr;
  if (mel && (!mel_count || !mel_woff || !mel_w || mel_nnz <= 0)) return IAS_ERR_ARG;
  if (!mel && n_out != n_fft / 2 + 1) return IAS_ERR_ARG;
  if (n_out <= 0 || n_out > n_fft / 2 + 1) return IAS_ERR_ARG;
  const int F = 1 + T / hop;
  if (F > 2147483647 / n_fft) return IAS_ERR_UNSUPPORTED;
  if (tables != nullptr && ias_diag_env("IAS_STFT_GRAD_V1") == nullptr) {
    // the frame part on the forward's wave-per-frame FFT core; overlap-add inside the kernel where the shape allows it
    // (IAS_STFT_GRAD_NOSPAN=1: the [B,F,n_fft] tensor + stft_grad_ola_kernel of round 2)
    static const bool nospan = ias_diag_env("IAS_STFT_GRAD_NOSPAN") != nullptr && atoi(ias_diag_env("IAS_STFT_GRAD_NOSPAN")) != 0;
    if (!nospan && (reinterpret_cast<uintptr_t>(frame_grad) & 15) == 0) {
      int plan[3] = {0, 0, 0};
      const int rs = ias_stft_grad_spans(audio, tables, mel_start, mel_count, mel_woff, mel_w, mel ? mel_nnz : 0, n_out,
                                         target, coef, frame_grad, B, T, n_fft, hop, power, loss_mode, scale, eps, plan,
                                         stream_);
      if (rs == IAS_OK) {
        hipLaunchKernelGGL(stft_grad_combine_kernel, dim3((T + SG_THREADS - 1) / SG_THREADS, B), dim3(SG_THREADS), 0,
                           stream, frame_grad, g_loss, g_audio, T, F, n_fft, hop, plan[0], plan[1], plan[2]);
        return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
      }
      if (rs != IAS_ERR_UNSUPPORTED) return rs;
    }
    const int rc = ias_stft_grad_frames(audio, tables, mel_start, mel_count, mel_woff, mel_w, mel ? mel_nnz : 0, n_out,
                                        target, coef, frame_grad, B, T, n_fft, hop, power, loss_mode, scale, eps, stream_);
    if (rc != IAS_OK) return rc;
    hipLaunchKernelGGL(stft_grad_ola_kernel, dim3((T + SG_THREADS - 1) / SG_THREADS, B), dim3(SG_THREADS), 0, stream,
                       frame_grad, g_loss, g_audio, T, F, n_fft, hop);
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }
  SgArgs g;
  g.audio = audio; g.window = window; g.mel_start = mel_start; g.mel_count = mel_count; g.mel_woff = mel_woff;
  g.mel_w = mel_w; g.target = target; g.frame_grad = frame_grad;
  g.T = T; g.F = F; g.N = n_fft; g.log2n = n_fft == 512 ? 9 : (n_fft == 1024 ? 10 : 11); g.hop = hop;
  g.n_out = n_out; g.power2 = power == 2; g.scale = scale;
  g.loss_mode = loss_mode; g.coef = coef; g.eps = eps;
  g.mel_nnz = mel ? mel_nnz : 0;
  const int NB = n_fft / 2 + 1;
  const size_t lds = sizeof(sg_cpx) * (2 * (size_t)n_fft + n_fft / 2) +
                     sizeof(float) * (6 * (size_t)(NB + 3) + 2 * ((n_out + 3) & ~3) + ((g.mel_nnz + 3) & ~3)) +
                     sizeof(int) * 3 * (size_t)(mel ? n_out : 0);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)stft_grad_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int pairs = (F + 1) / 2;
  hipLaunchKernelGGL(stft_grad_frames_kernel, dim3((pairs + SG_PAIRS - 1) / SG_PAIRS, B), dim3(SG_THREADS), lds, stream,
                     g);
  hipLaunchKernelGGL(stft_grad_ola_kernel, dim3((T + SG_THREADS - 1) / SG_THREADS, B), dim3(SG_THREADS), 0, stream,
                     frame_grad, g_loss, g_audio, T, F, n_fft, hop);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------ scalar glue
// The few scalar operations between the fused reductions and the backward kernels, as one launch each instead of a
// dozen at::native elementwise launches per resolution (profiles/r03b_kstats_gradstep.csv: 6 % of the gradient step's
// kernel time and most of its launch count).  fp64, the same operations in the same order as the torch expressions
// they replace (spectral.py: MultiResolutionSTFTLoss._forward, _mrstft_plan_backward).
#define IAS_MR_MAX_RES 8
struct MrTotalArgs { const double* sums[IAS_MR_MAX_RES]; double count[IAS_MR_MAX_RES]; int nres; };

// loss = (sum_k sqrt(s_k[0]) / sqrt(s_k[1]) + s_k[2] / count_k) / nres   (auraloss MultiResolutionSTFTLoss defaults:
// spectral convergence + log-magnitude L1 per resolution, mean over resolutions)
__global__ void mrstft_total_kernel(const MrTotalArgs a, float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double total = 0.0;
  for (int k = 0; k < a.nres; ++k) {
    const double* s = a.sums[k];
    const double term = sqrt(s[0]) / sqrt(s[1]) + s[2] / a.count[k];
    total = k == 0 ? term : total + term;
  }
  loss[0] = (float)(total / (double)a.nres);
}

extern "C" int ias_mrstft_total(const double* const* sums_host, const double* counts_host, int nres, float* loss,
                                void* stream_) {
  if (!sums_host || !counts_host || !loss || nres < 1 || nres > IAS_MR_MAX_RES) return IAS_ERR_ARG;
  MrTotalArgs a;
  for (int k = 0; k < IAS_MR_MAX_RES; ++k) { a.sums[k] = nullptr; a.count[k] = 1.0; }
  for (int k = 0; k < nres; ++k) {
    if (!sums_host[k] || !(counts_host[k] > 0.0)) return IAS_ERR_ARG;
    a.sums[k] = sums_host[k]; a.count[k] = counts_host[k];
  }
  a.nres = nres;
  hipLaunchKernelGGL(mrstft_total_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, a, loss);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// coef[0] = g / (nres sqrt(s[0]) sqrt(s[1]))  (0 when the denominator is 0),  coef[1] = g / (nres count): the
// cotangent coefficients ias_stft_loss_backward (loss_mode 2) takes; g = g_loss[0] (NULL = 1).
__global__ void mrstft_coef_kernel(const double* __restrict__ s, const float* __restrict__ g_loss, double count,
                                   int nres, double* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double g = g_loss ? (double)g_loss[0] : 1.0;
  const double den = sqrt(s[0]) * sqrt(s[1]);
  coef[0] = den > 0.0 ? g / ((double)nres * den) : 0.0;
  coef[1] = g / ((double)nres * count);
}

extern "C" int ias_mrstft_coef(const double* sums, const float* g_loss, double count, int nres, double* coef,
                               void* stream_) {
  if (!sums || !coef || nres < 1 || !(count > 0.0)) return IAS_ERR_ARG;
  hipLaunchKernelGGL(mrstft_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, sums, g_loss, count, nres, coef);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------ plain L1 pair
// mean |x - y| over n floats as per-workgroup fp64 partials [grid][3] (column 0; ias_reduce_partials finishes in fixed
// order), and its gradient  gx = sign(x - y) * g_loss[0] * scale  (sign(0) = 0, torch.abs' convention): the SubbandL1
// loss (spectral.py) without the sub / abs / mean / sign / mul / expand launches.
#define L1_THREADS 256
#define L1_PER_WG (L1_THREADS * 4 * 8)
__global__ __launch_bounds__(L1_THREADS) void l1_partials_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                 long long n, double* __restrict__ partials) {
  __shared__ double s_red[L1_THREADS / 64];
  const long long base = (long long)blockIdx.x * L1_PER_WG;
  float acc = 0.0f;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const long long i = base + ((long long)it * L1_THREADS + threadIdx.x) * 4;
    if (vec && i + 3 < n) {
      const float4 a = *reinterpret_cast<const float4*>(x + i), b = *reinterpret_cast<const float4*>(y + i);
      acc += (fabsf(a.x - b.x) + fabsf(a.y - b.y)) + (fabsf(a.z - b.z) + fabsf(a.w - b.w));
    } else {
      for (int e = 0; e < 4; ++e) if (i + e < n) acc += fabsf(x[i + e] - y[i + e]);
    }
  }
  double v = (double)acc;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < L1_THREADS / 64; ++w) t += s_red[w];
    partials[(size_t)blockIdx.x * 3] = t; partials[(size_t)blockIdx.x * 3 + 1] = 0.0; partials[(size_t)blockIdx.x * 3 + 2] = 0.0;
  }
}

__global__ __launch_bounds__(L1_THREADS) void l1_grad_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ g_loss, float scale, long long n,
                                                             float* __restrict__ gx) {
  const float g = (g_loss ? g_loss[0] : 1.0f) * scale;
  const long long base = (long long)blockIdx.x * L1_PER_WG;
  const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gx)) & 15) == 0;
  auto sg = [&](float d) { return d > 0.0f ? g : (d < 0.0f ? -g : 0.0f); };
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const long long i = base + ((long long)it * L1_THREADS + threadIdx.x) * 4;
    if (vec && i + 3 < n) {
      const float4 a = *reinterpret_cast<const float4*>(x + i), b = *reinterpret_cast<const float4*>(y + i);
      *reinterpret_cast<float4*>(gx + i) = make_float4(sg(a.x - b.x), sg(a.y - b.y), sg(a.z - b.z), sg(a.w - b.w));
    } else {
      for (int e = 0; e < 4; ++e) if (i + e < n) gx[i + e] = sg(x[i + e] - y[i + e]);
    }
  }
}

extern "C" long long ias_l1_partials_count(long long n) {
  if (n <= 0) return IAS_ERR_ARG;
  return (n + L1_PER_WG - 1) / L1_PER_WG;
}

extern "C" int ias_l1_partials(const float* x, const float* y, long long n, double* partials, void* stream_) {
  if (!x || !y || !partials || n <= 0 || (n + L1_PER_WG - 1) / L1_PER_WG > 2147483647LL) return IAS_ERR_ARG;
  hipLaunchKernelGGL(l1_partials_kernel, dim3((unsigned)((n + L1_PER_WG - 1) / L1_PER_WG)), dim3(L1_THREADS), 0,
                     (hipStream_t)stream_, x, y, n, partials);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

extern "C" int ias_l1_grad(const float* x, const float* y, const float* g_loss, float scale, long long n, float* gx,
                           void* stream_) {
  if (!x || !y || !gx || n <= 0 || (n + L1_PER_WG - 1) / L1_PER_WG > 2147483647LL) return IAS_ERR_ARG;
  hipLaunchKernelGGL(l1_grad_kernel, dim3((unsigned)((n + L1_PER_WG - 1) / L1_PER_WG)), dim3(L1_THREADS), 0,
                     (hipStream_t)stream_, x, y, g_loss, scale, n, gx);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
