// Framed STFT -> power / magnitude -> (optional) mel filterbank -> spectrogram or fused loss sums on the fp32 matrix
// cores of MI355X (gfx950): the DFT as small exact-fp32 GEMMs on v_mfma_f32_16x16x4_f32, the mel projection as a
// banded fp32 GEMM over 16-frame tiles.  Same contract as stft_kernel (csrc/spectral_kernels.hip), which stays as the
// VALU form (IAS_STFT_VALU=1) and the backward's FFT core.
//
// Spec: the commented mel block /root/reference/conf/config.yaml:51-61, its use /root/reference/audio_to_params.py:150-153
// (torchaudio MelSpectrogram semantics) and the auraloss TODO audio_to_params.py:233.
//
// Why the matrix cores: the headline step is bound by the vector ALU (the render's correctly-rounded phase arithmetic),
// the fp32 MFMA pipe is idle beside it, and v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain, so nothing is given up
// numerically (no bf16 anywhere).  A radix-8 x 8 x 8 FFT on the VALU needs three lane transposes through LDS per
// frame; here the only data that ever crosses lanes outside an MFMA is the spectrum Z (one LDS round trip for the
// Hermitian unpack) and the power values (one LDS write, read back as the mel GEMM's B operand).
//
// Per frame (n_fft = 1024: 512 packed complex points = 16 x 32; index conventions in stft_mfma.h):
//   stage 1   radix 16 over n1: the frame samples ARE the A operand (lane (i, kq) loads 16 bytes of the frame for rows
//             n2 = 2i, 2i+1), the DFT matrix [[cos, -sin], [sin, cos]] is the B operand in registers: 32 MFMAs.
//             The result S[n2][k1] has n2 on (lane>>4, register) and k1 on lane&15.
//   twiddle   S' = S W_512^(n2 k1), 8 complex products per lane.
//   stage 2a  radix 8 over a = n2 >> 2: S' is the B operand AS IT STANDS in the accumulators (the contraction index is
//             on (lane>>4, register), so no lane moves: the k-order inside a step is permuted and the constant A
//             operand is permuted to match), 16 MFMAs, 4 constant registers.
//   stage 2b  twiddle W_32^(b ka), radix 4 over b = n2 & 3 in registers.
//   unpack    Z -> LDS (conflict-free 8-byte stores), read back as pairs (k, 512 - k), X[k] and X[512-k] -> power.
//   mel       power -> LDS slot of the frame; once 16 frames of the workgroup are in: banded GEMM
//             mel[16 x 16 frames] += Wmel[16 x 4] P[4 x 16 frames], about 10 MFMAs per frame, the loss on the tile.
// 58 MFMAs (1856 matrix-pipe cycles) and ~200 vector instructions per frame against 551 vector + 126 LDS instructions
// of the VALU kernel.
#include "ias_common.h"
#include "stft_mfma.h"
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

extern "C" int ias_stft_num_frames(int T, int n_fft, int hop);

// ------------------------------------------------------------------------------------------------ host: constant block
static bool sm_supported(int n_fft) { return n_fft == 512 || n_fft == 1024 || n_fft == 2048; }

struct SmTile { int kb, nblk; };
static void sm_mel_tiles(const IasSmLayout& L, const int* mel_start, const int* mel_count, int n_out,
                         std::vector<SmTile>& tiles) {
  const int nt = (n_out + 15) / 16;
  tiles.assign(nt, SmTile{0, 0});
  for (int t = 0; t < nt; ++t) {
    int lo = 1 << 30, hi = -1;
    for (int m = 16 * t; m < std::min(16 * t + 16, n_out); ++m)
      if (mel_count[m] > 0) { lo = std::min(lo, mel_start[m]); hi = std::max(hi, mel_start[m] + mel_count[m]); }
    if (hi < 0) continue;
    int kb = lo & ~3;
    const int nblk = (hi - kb + 15) / 16;
    if (kb + 16 * nblk > L.pstr) kb = L.pstr - 16 * nblk;
    tiles[t].kb = kb; tiles[t].nblk = nblk;
  }
}

extern "C" long long ias_stft_mtables_len(int n_fft, const int* mel_start_host, const int* mel_count_host, int n_out) {
  if (!sm_supported(n_fft)) return IAS_ERR_UNSUPPORTED;
  const IasSmLayout L = ias_sm_layout(n_fft);
  long long len = L.off_mela;
  if (mel_start_host != nullptr) {
    if (!mel_count_host || n_out <= 0 || n_out > 16 * IAS_SM_MAX_TILES) return IAS_ERR_UNSUPPORTED;
    std::vector<SmTile> tiles;
    sm_mel_tiles(L, mel_start_host, mel_count_host, n_out, tiles);
    for (const SmTile& t : tiles) { if (t.kb < 0) return IAS_ERR_UNSUPPORTED; len += 256LL * t.nblk; }
  }
  return len;
}

// window_host [n_fft] (zero-padded, centred); mel_* host CSR filterbank as ias_stft takes it on the device, or NULL.
extern "C" int ias_stft_build_mtables(int n_fft, const float* window_host, const int* mel_start, const int* mel_count,
                                      const int* mel_woff, const float* mel_w, int n_out, float* out_host) {
  const long long len = ias_stft_mtables_len(n_fft, mel_start, mel_count, n_out);
  if (len < 0) return (int)len;
  if (!window_host || !out_host || (mel_start && (!mel_woff || !mel_w))) return IAS_ERR_ARG;
  const IasSmLayout L = ias_sm_layout(n_fft);
  const int Q = L.Q, N2 = L.N2;
  const double tau = 6.283185307179586476925287;
  std::memset(out_host, 0, sizeof(float) * (size_t)len);
  auto E = [&](int entry, int lane) -> float& { return out_host[64 * entry + lane]; };
  for (int l = 0; l < 64; ++l) {
    const int lo = l & 15, g = l >> 4;     // as A operand: row i = lo, k-quarter kq = g; as B / C: column j = lo, group G = g
    // window, in load order
    for (int v = 0; v < L.VPL; ++v) {
      int sample;
      if (Q == 16) { const int s = v >> 1, c = v & 1; sample = 2 * Q * (4 * s + g) + 2 * lo + c; }
      else { const int nh = Q / 32, s = v / (4 * nh), h = (v / 4) % nh, e4 = v & 3; sample = 2 * Q * (4 * s + g) + 64 * h + 4 * lo + e4; }
      E(L.e_win + v, l) = window_host[sample];
    }
    // stage-1 B operand: k element n1 = 4 s + kq (kq = g), column k1 = lo
    for (int s = 0; s < 4; ++s) {
      const double th = tau * (double)(((4 * s + g) * lo) % 16) / 16.0;
      E(L.e_b1 + s, l) = (float)cos(th);
      E(L.e_b1 + 4 + s, l) = (float)sin(th);
    }
    // twiddle 1: W_N2^(n2 k1), n2 = n2_of(t, 4 G + r), k1 = lo
    for (int t = 0; t < L.NT; ++t)
      for (int r = 0; r < 4; ++r) {
        const int n2 = ias_sm_n2_of(Q, t, 4 * g + r);
        const double ph = tau * (double)((n2 * lo) % N2) / (double)N2;
        E(L.e_tw1 + 2 * (4 * t + r), l) = (float)cos(ph);
        E(L.e_tw1 + 2 * (4 * t + r) + 1, l) = (float)sin(ph);
      }
    // stage-2a A operand: row i = lo -> (ka = i >> 1, c' = i & 1); k element (c, a), a = alpha(kq, x)
    for (int c = 0; c < 2; ++c)
      for (int x = 0; x < 2; ++x) {
        const int ka = lo >> 1, cp = lo & 1, al = (Q == 64) ? 4 * x + g : 2 * g + x;
        const double ps = tau * (double)((al * ka) % 8) / 8.0;
        float v;
        if (cp == 0) v = (c == 0) ? (float)cos(ps) : (float)sin(ps);      // Tr = S'r cos + S'i sin
        else v = (c == 0) ? (float)(-sin(ps)) : (float)cos(ps);           // Ti = S'i cos - S'r sin
        E(L.e_a2 + 2 * c + x, l) = v;
      }
    // twiddle 2: W_Q^(b ka), ka = 2 G + kl
    for (int b = 1; b < L.NB; ++b)
      for (int kl = 0; kl < 2; ++kl) {
        const int ka = 2 * g + kl;
        const double om = tau * (double)((b * ka) % Q) / (double)Q;
        E(L.e_tw2 + 2 * (2 * (b - 1) + kl), l) = (float)cos(om);
        E(L.e_tw2 + 2 * (2 * (b - 1) + kl) + 1, l) = (float)sin(om);
      }
    // unpack: W_nfft^k = (cos, -sin), k = 1 + lane + 64 i
    for (int i = 0; i < L.NPAIR_IT; ++i) {
      const int k = 1 + l + 64 * i;
      const double an = tau * (double)k / (double)n_fft;
      E(L.e_unp + 2 * i, l) = (float)cos(an);
      E(L.e_unp + 2 * i + 1, l) = (float)(-sin(an));
    }
  }
  if (mel_start != nullptr) {
    std::vector<SmTile> tiles;
    sm_mel_tiles(L, mel_start, mel_count, n_out, tiles);
    const int nt = (int)tiles.size();
    // longest-processing-time assignment of the tiles to the four waves
    std::vector<int> order(nt);
    for (int t = 0; t < nt; ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tiles[a].nblk > tiles[b].nblk; });
    std::vector<int> aoff(nt);
    int off = 0;
    for (int t = 0; t < nt; ++t) { aoff[t] = off; off += tiles[t].nblk; }
    int load[IAS_SM_WAVES] = {0, 0, 0, 0};
    int* desc = reinterpret_cast<int*>(out_host + L.off_desc);
    for (int idx = 0; idx < nt; ++idx) {
      const int t = order[idx];
      if (tiles[t].nblk == 0) continue;
      int w = 0;
      for (int q = 1; q < IAS_SM_WAVES; ++q) if (load[q] < load[w]) w = q;
      int* dw = desc + w * (1 + 4 * IAS_SM_MAX_TILES);
      const int n = dw[0]++;
      dw[1 + 4 * n] = t; dw[2 + 4 * n] = tiles[t].kb; dw[3 + 4 * n] = tiles[t].nblk; dw[4 + 4 * n] = aoff[t];
      load[w] += tiles[t].nblk;
    }
    float* A = out_host + L.off_mela;
    for (int t = 0; t < nt; ++t)
      for (int q = 0; q < tiles[t].nblk; ++q)
        for (int s = 0; s < 4; ++s)
          for (int l = 0; l < 64; ++l) {
            const int m = 16 * t + (l & 15), bin = tiles[t].kb + 16 * q + 4 * (l >> 4) + s;
            float v = 0.0f;
            if (m < n_out && bin >= mel_start[m] && bin < mel_start[m] + mel_count[m]) v = mel_w[mel_woff[m] + bin - mel_start[m]];
            A[256 * (aoff[t] + q) + 64 * s + l] = v;
          }
  }
  return IAS_OK;
}

// ------------------------------------------------------------------------------------------------------- device side
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct SmArgs {
  const float* audio;      // [B,T]
  const float* mtab;       // ias_stft_build_mtables block (device)
  float* out;              // [B,F,n_out] or null
  const float* target;     // [B,F,n_out] or null
  double* partials;        // [gridDim.x][3] or null
  const float* rowpeak;    // [B] or null
  int T, F, hop, n_out;
  int nframes;             // B * F
  int value_mode;          // 1: |X|, 2: |X|^2, 3: sqrt(max(|X|^2, eps))
  int loss_mode;           // 0: none, 1: sum |v - t|, 2: MR-STFT sums {(t-v)^2, t^2, |log v - log t|}
  float eps;
};

struct cx { float re, im; };
__device__ __forceinline__ cx operator+(cx a, cx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cx operator-(cx a, cx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cx mul_negi(cx a) { return {a.im, -a.re}; }     // a * (-i)
// forward DFTs (kernel e^{-2 pi i b kb / R}), in place, natural order
__device__ __forceinline__ void sm_dft2(cx* v) { const cx a = v[0] + v[1], b = v[0] - v[1]; v[0] = a; v[1] = b; }
__device__ __forceinline__ void sm_dft4(cx& v0, cx& v1, cx& v2, cx& v3) {
  const cx t0 = v0 + v2, t1 = v0 - v2, t2 = v1 + v3, t3 = mul_negi(v1 - v3);
  v0 = t0 + t2; v1 = t1 + t3; v2 = t0 - t2; v3 = t1 - t3;
}
__device__ __forceinline__ void sm_dft8(cx* v) {
  cx e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  sm_dft4(e0, e1, e2, e3);
  sm_dft4(o0, o1, o2, o3);
  const float h = 0.70710678118654752f;
  { const cx t = o1 + mul_negi(o1); o1 = {t.re * h, t.im * h}; }    // * W8^1 = (1 - i)/sqrt2
  o2 = mul_negi(o2);                                               // * W8^2 = -i
  { const cx t = mul_negi(o3) - o3; o3 = {t.re * h, t.im * h}; }    // * W8^3 = (-1 - i)/sqrt2
  v[0] = e0 + o0; v[4] = e0 - o0; v[1] = e1 + o1; v[5] = e1 - o1;
  v[2] = e2 + o2; v[6] = e2 - o2; v[3] = e3 + o3; v[7] = e3 - o3;
}
template <int R> __device__ __forceinline__ void sm_dftR(cx* v);
template <> __device__ __forceinline__ void sm_dftR<2>(cx* v) { sm_dft2(v); }
template <> __device__ __forceinline__ void sm_dftR<4>(cx* v) { sm_dft4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void sm_dftR<8>(cx* v) { sm_dft8(v); }

__device__ __forceinline__ f32x4 sm_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// LDS hand-overs between lanes of ONE wave need program order only (the LDS executes a wave's instructions in order)
__device__ __forceinline__ void sm_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int sm_reflect(int i, int T) {
  if (i < 0) i = -i;
  if (i >= T) i = 2 * (T - 1) - i;
  return i;
}
__device__ __forceinline__ float sm_wave_sum(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// The VPL samples of frame fi this lane feeds to stage 1 (load order of stft_mfma.h): 16-byte loads of interior frames
// on aligned rows, per-sample indexing at the row ends (reflect padding) and on odd alignments.
template <int LOG2N>
__device__ __forceinline__ void sm_load_frame(const SmArgs& a, int fi, int lo, int g, float* __restrict__ x) {
  constexpr IasSmLayout L = ias_sm_layout(1 << LOG2N);
  constexpr int Q = L.Q, NH = Q >= 32 ? Q / 32 : 1;
  const int b = fi / a.F, f = fi - b * a.F;
  const int t0 = f * a.hop - L.N2;
  const float* arow = a.audio + (size_t)b * a.T;
  const bool interior = t0 >= 0 && t0 + L.n_fft <= a.T;
  if (Q >= 32) {
    const float* p = arow + t0 + 2 * Q * g + 4 * lo;
    if (interior && ((reinterpret_cast<uintptr_t>(arow + t0) & 15) == 0)) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(p + 8 * Q * s + 64 * h);
          x[(s * NH + h) * 4 + 0] = q[0]; x[(s * NH + h) * 4 + 1] = q[1];
          x[(s * NH + h) * 4 + 2] = q[2]; x[(s * NH + h) * 4 + 3] = q[3];
        }
    } else if (interior) {
#pragma unroll
      for (int v = 0; v < L.VPL; ++v) x[v] = p[8 * Q * (v / (4 * NH)) + 64 * ((v / 4) % NH) + (v & 3)];
    } else {
#pragma unroll
      for (int v = 0; v < L.VPL; ++v)
        x[v] = arow[sm_reflect(t0 + 2 * Q * g + 4 * lo + 8 * Q * (v / (4 * NH)) + 64 * ((v / 4) % NH) + (v & 3), a.T)];
    }
  } else {
    const float* p = arow + t0 + 2 * Q * g + 2 * lo;
    if (interior && ((reinterpret_cast<uintptr_t>(arow + t0) & 7) == 0)) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f32x2 q = *reinterpret_cast<const f32x2*>(p + 8 * Q * s);
        x[2 * s] = q[0]; x[2 * s + 1] = q[1];
      }
    } else if (interior) {
#pragma unroll
      for (int v = 0; v < L.VPL; ++v) x[v] = p[8 * Q * (v >> 1) + (v & 1)];
    } else {
#pragma unroll
      for (int v = 0; v < L.VPL; ++v) x[v] = arow[sm_reflect(t0 + 2 * Q * g + 2 * lo + 8 * Q * (v >> 1) + (v & 1), a.T)];
    }
  }
}

// A workgroup of four waves takes groups of 16 consecutive frames of the flattened [B*F] frame list (a group may
// straddle two rows: frames are independent); wave w transforms frames 4w .. 4w+3 of the group.  MEL: the power values
// of the 16 frames meet in LDS (slot = frame of the group) and the waves share the mel tiles of the group.
template <int LOG2N, bool MEL>
__global__ __launch_bounds__(256, 2) void stft_mfma_kernel(const SmArgs a) {
  constexpr IasSmLayout L = ias_sm_layout(1 << LOG2N);
  constexpr int N2 = L.N2, Q = L.Q, NT = L.NT, NB = L.NB, VPL = L.VPL, NPI = L.NPAIR_IT, NH = Q >= 32 ? Q / 32 : 1;
  constexpr int PSTR = L.pstr;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x2* sZall = reinterpret_cast<f32x2*>(smem);                 // [4 waves][N2] spectrum scratch
  float* sP = smem + 2 * N2 * IAS_SM_WAVES;                      // [16 slots][PSTR] power values (MEL)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 15, g = lane >> 4;
  f32x2* sZ = sZall + wave * N2;
  const float* tab = a.mtab + lane;
  auto ENT = [&](int e) { return tab[64 * e]; };

  // frame-invariant operands and twiddles: registers
  float win[VPL], b1c[4], b1s[4], b1n[4], tw1c[NT * 4], tw1s[NT * 4], a2[4], unr[NPI], uni[NPI];
  float tw2c[(NB - 1) * 2], tw2s[(NB - 1) * 2];
#pragma unroll
  for (int v = 0; v < VPL; ++v) win[v] = ENT(L.e_win + v);
#pragma unroll
  for (int s = 0; s < 4; ++s) { b1c[s] = ENT(L.e_b1 + s); b1s[s] = ENT(L.e_b1 + 4 + s); b1n[s] = -b1s[s]; }
#pragma unroll
  for (int i = 0; i < NT * 4; ++i) { tw1c[i] = ENT(L.e_tw1 + 2 * i); tw1s[i] = ENT(L.e_tw1 + 2 * i + 1); }
#pragma unroll
  for (int i = 0; i < 4; ++i) a2[i] = ENT(L.e_a2 + i);
#pragma unroll
  for (int i = 0; i < (NB - 1) * 2; ++i) { tw2c[i] = ENT(L.e_tw2 + 2 * i); tw2s[i] = ENT(L.e_tw2 + 2 * i + 1); }
#pragma unroll
  for (int i = 0; i < NPI; ++i) { unr[i] = ENT(L.e_unp + 2 * i); uni[i] = ENT(L.e_unp + 2 * i + 1); }

  if (MEL) {
    for (int i = tid; i < 16 * PSTR; i += 256) sP[i] = 0.0f;     // padding and unused slots must stay finite
    __syncthreads();
  }
  const int* desc = reinterpret_cast<const int*>(a.mtab + L.off_desc) + wave * (1 + 4 * IAS_SM_MAX_TILES);
  const float* melA = a.mtab + L.off_mela + lane;

  const int ngroups = (a.nframes + 15) >> 4;
  float l0 = 0.f, l1 = 0.f, l2 = 0.f;
  float xc[VPL], xn[VPL];
  {
    const int fi0 = blockIdx.x * 16 + 4 * wave;
    if (blockIdx.x < ngroups && fi0 < a.nframes) sm_load_frame<LOG2N>(a, fi0, lo, g, xc);
  }
  bool first = true;
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const int fi = grp * 16 + 4 * wave + q;
      const bool valid = fi < a.nframes;                     // wave-uniform
      // the next frame of this wave: prefetched while the current one is transformed
      const int fnext = q < 3 ? fi + 1 : (grp + (int)gridDim.x) * 16 + 4 * wave;
      const bool more = fnext < a.nframes && (q < 3 || grp + (int)gridDim.x < ngroups);
      if (more) sm_load_frame<LOG2N>(a, fnext, lo, g, xn);

      float pk[NPI], pn[NPI], p0 = 0.f, pN = 0.f;
      if (valid) {
        float pscale = 0.25f;
        if (a.rowpeak != nullptr) {
          const float pkv = a.rowpeak[fi / a.F];
          if (pkv > 1.0f) { const float r = 1.0f / pkv; pscale = 0.25f * (r * r); }
        }
        // ---- stage 1: S[n2][k1] on the matrix cores, frame samples as the A operand
        float xw[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) xw[v] = xc[v] * win[v];
        f32x4 acc1[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) { acc1[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const float av = Q == 16 ? xw[2 * s + c] : xw[(s * NH + (t >> 1)) * 4 + 2 * (t & 1) + c];
              acc1[t][0] = sm_mfma(av, c == 0 ? b1c[s] : b1s[s], acc1[t][0]);   // Sr += zr cos + zi sin
              acc1[t][1] = sm_mfma(av, c == 0 ? b1n[s] : b1c[s], acc1[t][1]);   // Si += zi cos - zr sin
            }
          }
        }
        // ---- twiddle 1: S' = S W_N2^(n2 k1)
        f32x4 sp[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sr = acc1[t][0][r], si = acc1[t][1][r], c = tw1c[4 * t + r], s_ = tw1s[4 * t + r];
            sp[t][0][r] = fmaf(si, s_, sr * c);
            sp[t][1][r] = fmaf(-sr, s_, si * c);
          }
        // ---- stage 2a: radix 8 over a = n2 / NB; S' is the B operand as it stands
        f32x4 acc2[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) acc2[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              const int t = Q == 16 ? 0 : (Q == 32 ? (b & 1) : 2 * x + (b & 1));
              const int r = Q == 16 ? 2 * x + b : (Q == 32 ? 2 * x + (b >> 1) : (b >> 1));
              acc2[b] = sm_mfma(a2[2 * c + x], sp[t][c][r], acc2[b]);
            }
        // ---- twiddle 2 + stage 2b (radix NB in registers), Z -> LDS in natural order
#pragma unroll
        for (int kl = 0; kl < 2; ++kl) {
          cx tv[NB];
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const float tr = acc2[b][2 * kl], ti = acc2[b][2 * kl + 1];
            if (b == 0) tv[b] = {tr, ti};
            else {
              const float c = tw2c[2 * (b - 1) + kl], s_ = tw2s[2 * (b - 1) + kl];
              tv[b] = {fmaf(ti, s_, tr * c), fmaf(-tr, s_, ti * c)};
            }
          }
          sm_dftR<NB>(tv);
#pragma unroll
          for (int kb = 0; kb < NB; ++kb) sZ[lo + 16 * ((2 * g + kl) + 8 * kb)] = (f32x2){tv[kb].re, tv[kb].im};
        }
        sm_wave_sync();
        // ---- Hermitian unpack of the packed real transform: bins k and N2 - k from Z[k], Z[N2 - k]
        const f32x2 z0 = sZ[0];
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
          const int k = 1 + lane + 64 * i;
          const f32x2 zk = sZ[k], zn = sZ[N2 - k];
          const float ea = zk[0] + zn[0], eb = zk[1] - zn[1], od = zk[0] - zn[0], os = zk[1] + zn[1];
          const float tx = fmaf(uni[i], od, unr[i] * os), ty = fmaf(-unr[i], od, uni[i] * os);
          const float xr = ea + tx, xi = eb + ty, yr = ea - tx, yi = eb - ty;
          pk[i] = fmaf(xi, xi, xr * xr) * pscale;
          pn[i] = fmaf(yi, yi, yr * yr) * pscale;
        }
        {
          const float s0 = z0[0] + z0[1], d0 = z0[0] - z0[1];
          p0 = (s0 * s0) * (4.0f * pscale);
          pN = (d0 * d0) * (4.0f * pscale);
        }
        if (a.value_mode == 1) {
#pragma unroll
          for (int i = 0; i < NPI; ++i) { pk[i] = sqrtf(pk[i]); pn[i] = sqrtf(pn[i]); }
          p0 = sqrtf(p0); pN = sqrtf(pN);
        } else if (a.value_mode == 3) {
#pragma unroll
          for (int i = 0; i < NPI; ++i) { pk[i] = sqrtf(fmaxf(pk[i], a.eps)); pn[i] = sqrtf(fmaxf(pn[i], a.eps)); }
          p0 = sqrtf(fmaxf(p0, a.eps)); pN = sqrtf(fmaxf(pN, a.eps));
        }
        sm_wave_sync();   // every Z read of this frame precedes the next frame's Z stores (program order)
      }
      if (MEL) {
        // the previous group's mel tiles are read by all four waves: nobody overwrites a slot before they are done
        if (q == 0 && !first) __syncthreads();
        if (valid) {
          float* slot = sP + (4 * wave + q) * PSTR;
#pragma unroll
          for (int i = 0; i < NPI; ++i) { const int k = 1 + lane + 64 * i; slot[k] = pk[i]; slot[N2 - k] = pn[i]; }
          if (lane == 0) { slot[0] = p0; slot[N2] = pN; }
        }
      } else if (valid) {
        // linear bins: store / fused loss sums straight from the registers
        const size_t row = (size_t)fi * (N2 + 1);
        auto emit = [&](int k, float v) {
          float t = 0.f;
          if (a.loss_mode != 0) t = a.target[row + k];
          if (a.out != nullptr) a.out[row + k] = v;
          if (a.loss_mode == 1) l0 += fabsf(v - t);
          else if (a.loss_mode == 2) {
            const float d = t - v; l0 = fmaf(d, d, l0); l1 = fmaf(t, t, l1); l2 += fabsf(logf(v) - logf(t));
          }
        };
#pragma unroll
        for (int i = 0; i < NPI; ++i) {
          const int k = 1 + lane + 64 * i;
          emit(k, pk[i]);
          if (k != N2 - k) emit(N2 - k, pn[i]);
        }
        if (lane == 0) { emit(0, p0); emit(N2, pN); }
      }
      if (more) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) xc[v] = xn[v];
      }
    }
    first = false;
    if (MEL) {
      __syncthreads();                                   // the 16 slots of the group are complete
      const int ntile = desc[0];
      const int fj = grp * 16 + lo;                      // the frame in column lo of the tile
      const bool okj = fj < a.nframes;
      for (int n = 0; n < ntile; ++n) {
        const int t = desc[1 + 4 * n], kb = desc[2 + 4 * n], nblk = desc[3 + 4 * n], aoff = desc[4 + 4 * n];
        const float* Ap = melA + 256 * aoff;
        const float* Pp = sP + lo * PSTR + kb + 4 * g;
        f32x4 acce = (f32x4){0.f, 0.f, 0.f, 0.f}, acco = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int m0 = 16 * t + 4 * g;
        f32x4 tg = (f32x4){0.f, 0.f, 0.f, 0.f};
        const size_t orow = (size_t)fj * a.n_out + m0;
        const bool full = okj && m0 + 4 <= a.n_out && (a.n_out & 3) == 0;
        if (a.loss_mode != 0 && okj) {
          if (full) tg = *reinterpret_cast<const f32x4*>(a.target + orow);
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (m0 + r < a.n_out) tg[r] = a.target[orow + r];
          }
        }
#pragma unroll 2
        for (int qb = 0; qb < nblk; ++qb) {
          const f32x4 pv = *reinterpret_cast<const f32x4*>(Pp + 16 * qb);
          const float w0 = Ap[256 * qb], w1 = Ap[256 * qb + 64], w2 = Ap[256 * qb + 128], w3 = Ap[256 * qb + 192];
          acce = sm_mfma(w0, pv[0], acce);
          acco = sm_mfma(w1, pv[1], acco);
          acce = sm_mfma(w2, pv[2], acce);
          acco = sm_mfma(w3, pv[3], acco);
        }
        const f32x4 mv = acce + acco;
        if (okj) {
          if (a.out != nullptr) {
            if (full) *reinterpret_cast<f32x4*>(a.out + orow) = mv;
            else {
#pragma unroll
              for (int r = 0; r < 4; ++r) if (m0 + r < a.n_out) a.out[orow + r] = mv[r];
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (m0 + r < a.n_out) {
              const float v = mv[r], tt = tg[r];
              if (a.loss_mode == 1) l0 += fabsf(v - tt);
              else if (a.loss_mode == 2) {
                const float d = tt - v; l0 = fmaf(d, d, l0); l1 = fmaf(tt, tt, l1); l2 += fabsf(logf(v) - logf(tt));
              }
            }
          }
        }
      }
    }
  }

  if (a.partials != nullptr) {
    __shared__ float s_red[IAS_SM_WAVES][4];
    l0 = sm_wave_sum(l0); l1 = sm_wave_sum(l1); l2 = sm_wave_sum(l2);
    if (lane == 0) { s_red[wave][0] = l0; s_red[wave][1] = l1; s_red[wave][2] = l2; }
    __syncthreads();
    if (tid < 3) {
      double sacc = 0.0;
      for (int w = 0; w < IAS_SM_WAVES; ++w) sacc += (double)s_red[w][tid];
      a.partials[(size_t)blockIdx.x * 3 + tid] = sacc;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ launch
static int sm_num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n = v;
  }
  return n;
}
// workgroups of the persistent grid (also the number of loss partial records)
int ias_sm_grid(long long nframes) {
  static const int env = getenv("IAS_STFT_MFMA_WGS") ? atoi(getenv("IAS_STFT_MFMA_WGS")) : 0;   // diagnostics
  const long long ngroups = (nframes + 15) / 16;
  const long long cap = env > 0 ? env : 2LL * sm_num_cus();
  return (int)std::min(ngroups, cap);
}
bool ias_sm_enabled(int n_fft) {
  static const int valu = getenv("IAS_STFT_VALU") ? atoi(getenv("IAS_STFT_VALU")) : 0;   // diagnostics: the VALU kernel
  return !valu && n_fft == 1024;
}

int ias_sm_launch(const float* audio, const float* mtab, bool mel, float* out, const float* target, double* partials,
                  const float* rowpeak, int B, int T, int F, int n_fft, int hop, int n_out, int value_mode, int loss_mode,
                  float eps, hipStream_t stream) {
  if ((long long)B * F > 2000000000LL) return IAS_ERR_UNSUPPORTED;
  if (mel && n_out > 16 * IAS_SM_MAX_TILES) return IAS_ERR_UNSUPPORTED;
  SmArgs a;
  a.audio = audio; a.mtab = mtab; a.out = out; a.target = target; a.partials = partials; a.rowpeak = rowpeak;
  a.T = T; a.F = F; a.hop = hop; a.n_out = n_out; a.nframes = B * F;
  a.value_mode = value_mode; a.loss_mode = loss_mode; a.eps = eps;
  const IasSmLayout L = ias_sm_layout(n_fft);
  const size_t lds = sizeof(float) * (2 * L.N2 * IAS_SM_WAVES + (mel ? 16 * L.pstr : 0));
  const dim3 grid(ias_sm_grid(a.nframes)), block(256);
#define IAS_SM_LAUNCH(LOG2N, MEL)                                                                                  \
  do {                                                                                                             \
    if (lds > 64 * 1024)                                                                                           \
      (void)hipFuncSetAttribute((const void*)stft_mfma_kernel<LOG2N, MEL>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                         \
    hipLaunchKernelGGL((stft_mfma_kernel<LOG2N, MEL>), grid, block, lds, stream, a);                               \
  } while (0)
  if (n_fft == 1024) { if (mel) IAS_SM_LAUNCH(10, true); else IAS_SM_LAUNCH(10, false); }
  else return IAS_ERR_UNSUPPORTED;
#undef IAS_SM_LAUNCH
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
