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
//   unpack    the upper half of Z -> LDS (conflict-free 8-byte stores into the frame's own power slot), each lane pairs
//             its own lower-half bins k with Z[512 - k] read back from there: X[k], X[512-k] -> power.
//   mel       power -> LDS slot of the frame; once the 16 frames of the workgroup's group are in: banded GEMM
//             mel[16 x 16 frames] += Wmel[16 x 4] P[4 x 16 frames], about 11 MFMAs per frame, the loss on the tile.
// ~60 MFMAs (1900 matrix-pipe cycles) and ~280 vector instructions per frame against 551 vector + 126 LDS
// instructions of the VALU kernel.  Workgroups of eight waves, two per CU (four waves per SIMD: <= 128 VGPRs); work is
// handed out in 16-frame groups by a ticket counter, so the CUs stay evenly loaded whatever the frame count.
#include "ias_common.h"
#include "stft_mfma.h"
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

extern "C" int ias_stft_num_frames(int T, int n_fft, int hop);

// ------------------------------------------------------------------------------------------------ host: constant block
static bool sm_supported(int n_fft) { return n_fft == 512 || n_fft == 1024 || n_fft == 2048; }

// A mel tile: rows [r0, r1) of the 16-output M tile `tile` (a tile whose band is long is cut into halves / quarters of
// its rows, each with its own, shorter band), bins [kb, kb + 16 nblk).
struct SmTile { int tile, r0, r1, kb, nblk; };
static SmTile sm_make_tile(const IasSmLayout& L, const int* mel_start, const int* mel_count, int n_out, int tile, int r0, int r1) {
  SmTile t{tile, r0, r1, 0, 0};
  int lo = 1 << 30, hi = -1;
  for (int m = 16 * tile + r0; m < std::min(16 * tile + r1, n_out); ++m)
    if (mel_count[m] > 0) { lo = std::min(lo, mel_start[m]); hi = std::max(hi, mel_start[m] + mel_count[m]); }
  if (hi < 0) return t;
  int kb = lo & ~3;
  const int nblk = (hi - kb + 15) / 16;
  if (kb + 16 * nblk > L.pstr) kb = L.pstr - 16 * nblk;
  t.kb = kb; t.nblk = nblk;
  return t;
}
// The tiles and their longest-processing-time assignment to the waves of a workgroup; while the waves are unbalanced
// (or one has more blocks than the kernel keeps in registers) the longest tile of the most loaded wave that can still
// be cut is cut in two.  Returns false if a band does not fit the power buffer or a wave ends up with too many blocks.
static bool sm_mel_plan(const IasSmLayout& L, const int* mel_start, const int* mel_count, int n_out,
                        std::vector<SmTile>& tiles, std::vector<int> (&mine)[IAS_SM_WAVES]) {
  const int nt = (n_out + 15) / 16;
  tiles.clear();
  for (int t = 0; t < nt; ++t) tiles.push_back(sm_make_tile(L, mel_start, mel_count, n_out, t, 0, 16));
  for (int iter = 0; iter < 4 * IAS_SM_MAX_TILES; ++iter) {
    std::vector<int> order(tiles.size());
    for (size_t t = 0; t < tiles.size(); ++t) order[t] = (int)t;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tiles[a].nblk > tiles[b].nblk; });
    int load[IAS_SM_WAVES] = {0, 0, 0, 0};
    for (auto& m : mine) m.clear();
    for (int t : order) {
      if (tiles[t].nblk == 0) continue;
      int w = 0;
      for (int q = 1; q < IAS_SM_WAVES; ++q) if (load[q] < load[w]) w = q;
      mine[w].push_back(t);
      load[w] += tiles[t].nblk;
    }
    int wmax = 0, wmin = 0;
    for (int q = 1; q < IAS_SM_WAVES; ++q) { if (load[q] > load[wmax]) wmax = q; if (load[q] < load[wmin]) wmin = q; }
    int cut = -1;
    if (load[wmax] - load[wmin] > 1)
      for (int t : mine[wmax])
        if (tiles[t].r1 - tiles[t].r0 >= 8 && tiles[t].nblk >= 2 && (cut < 0 || tiles[t].nblk > tiles[cut].nblk)) cut = t;
    if (cut < 0) break;
    const int tile = tiles[cut].tile, r0 = tiles[cut].r0, r1 = tiles[cut].r1, mid = (r0 + r1) / 2;
    tiles[cut] = sm_make_tile(L, mel_start, mel_count, n_out, tile, r0, mid);
    tiles.push_back(sm_make_tile(L, mel_start, mel_count, n_out, tile, mid, r1));
  }
  for (const SmTile& t : tiles) if (t.kb < 0) return false;
  for (auto& m : mine) {
    int n = 0;
    for (int t : m) n += tiles[t].nblk;
    if (n > IAS_SM_MAX_BLOCKS) return false;
  }
  return true;
}

extern "C" long long ias_stft_mtables_len(int n_fft, const int* mel_start_host, const int* mel_count_host, int n_out) {
  if (!sm_supported(n_fft)) return IAS_ERR_UNSUPPORTED;
  const IasSmLayout L = ias_sm_layout(n_fft);
  long long len = L.off_mela;
  if (mel_start_host != nullptr) {
    if (!mel_count_host || n_out <= 0 || n_out > 16 * IAS_SM_MAX_TILES) return IAS_ERR_UNSUPPORTED;
    std::vector<SmTile> tiles;
    std::vector<int> mine[IAS_SM_WAVES];
    if (!sm_mel_plan(L, mel_start_host, mel_count_host, n_out, tiles, mine)) return IAS_ERR_UNSUPPORTED;
    for (const SmTile& t : tiles) len += 256LL * t.nblk;
    len += 256LL * IAS_SM_REG_BLOCKS;   // zero blocks behind the table: a wave always loads whole passes of IAS_SM_REG_BLOCKS blocks
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
    // unpack: W_nfft^k = (cos, -sin) of the lane's own lower-half bins k = k1 + 16 (2 G + kl) + 128 kb, kb < NB/2
    for (int kl = 0; kl < 2; ++kl)
      for (int kb = 0; kb < L.NB / 2; ++kb) {
        const int k = lo + 16 * (2 * g + kl) + 128 * kb;
        const double an = tau * (double)k / (double)n_fft;
        E(L.e_unp + 2 * (kl * (L.NB / 2) + kb), l) = (float)cos(an);
        E(L.e_unp + 2 * (kl * (L.NB / 2) + kb) + 1, l) = (float)(-sin(an));
      }
  }
  if (mel_start != nullptr) {
    std::vector<SmTile> tiles;
    std::vector<int> mine[IAS_SM_WAVES];
    if (!sm_mel_plan(L, mel_start, mel_count, n_out, tiles, mine)) return IAS_ERR_UNSUPPORTED;
    int* desc = reinterpret_cast<int*>(out_host + L.off_desc);
    float* A = out_host + L.off_mela;
    int ablk = 0;
    for (int w = 0; w < IAS_SM_WAVES; ++w) {
      int* dw = desc + w * IAS_SM_DESC_W;
      int nb = 0;
      dw[1] = ablk;
      for (size_t ti = 0; ti < mine[w].size(); ++ti) {
        const SmTile& t = tiles[mine[w][ti]];
        const int next = ti + 1 < mine[w].size() ? tiles[mine[w][ti + 1]].tile : 255;
        for (int q = 0; q < t.nblk; ++q, ++nb, ++ablk) {
          dw[4 + 2 * nb] = t.kb + 16 * q;
          int quarters = 0;
          for (int qr = t.r0 / 4; qr < t.r1 / 4; ++qr) quarters |= 0x400 << qr;
          dw[5 + 2 * nb] = t.tile | (q == 0 ? 0x100 : 0) | (q == t.nblk - 1 ? 0x200 : 0) | quarters | (next << 16);
          for (int s4 = 0; s4 < 4; ++s4)
            for (int l = 0; l < 64; ++l) {
              const int row = l & 15, m = 16 * t.tile + row, bin = t.kb + 16 * q + 4 * (l >> 4) + s4;
              float v = 0.0f;
              if (row >= t.r0 && row < t.r1 && m < n_out && bin >= mel_start[m] && bin < mel_start[m] + mel_count[m])
                v = mel_w[mel_woff[m] + bin - mel_start[m]];
              A[256 * ablk + 64 * s4 + l] = v;
            }
        }
      }
      dw[0] = nb;
      dw[2] = mine[w].empty() ? 255 : tiles[mine[w][0]].tile;   // the wave's first tile (its target rows are fetched ahead)
    }
    int npass = 1;
    for (int w = 0; w < IAS_SM_WAVES; ++w)
      npass = std::max(npass, (desc[w * IAS_SM_DESC_W] + IAS_SM_REG_BLOCKS - 1) / IAS_SM_REG_BLOCKS);
    for (int w = 0; w < IAS_SM_WAVES; ++w) desc[w * IAS_SM_DESC_W + 3] = npass;
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
  int ngroups;             // 16-frame groups
  unsigned magicF;         // floor(2^32 / F)
  int* ticket;             // [2] work counter (next unit, finished workgroups / waves): zero between launches; or null
  int value_mode;          // 1: |X|, 2: |X|^2, 3: sqrt(max(|X|^2, eps))
  int loss_mode;           // 0: none, 1: sum |v - t|, 2: MR-STFT sums {(t-v)^2, t^2, |log v - log t|}
  float eps;
#ifdef IAS_SM_STAMPS
  unsigned long long* stamps;   // diagnostics build only: [workgroup][wave][256] s_memtime values
#endif
};

#ifdef IAS_DIAG   // the device code and its launcher exist in the diagnostic library only (ias_common.h): the product library's
                  // STFT is the radix-8 kernel family of spectral_kernels.hip / stft2_kernels.hip, whatever the environment says
// In-kernel stamps (diagnostic build -DIAS_SM_STAMPS only; the product build compiles none of this): where a wave's
// cycles go, phase by phase.  scripts/diag/stft_stamps.py builds and reads them.
#ifdef IAS_SM_STAMPS
#define SM_STAMP(id)                                                                                     \
  do {                                                                                                   \
    if (a.stamps != nullptr && stamp_n < 255) {                                                          \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                        \
      if (lane == 0) a.stamps[((size_t)blockIdx.x * IAS_SM_WAVES + wave) * 256 + 1 + stamp_n] = (t_ << 8) | (id); \
      ++stamp_n;                                                                                         \
    }                                                                                                    \
  } while (0)
#else
#define SM_STAMP(id) do {} while (0)
#endif

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

// The VPL samples of frame f of row b this lane feeds to stage 1 (load order of stft_mfma.h).  Interior frames: 16-byte
// loads (the hardware takes any 4-byte alignment: hops that are not multiples of four samples cost bandwidth, not a
// second code path).  Frames that touch the row ends (reflect padding: two or three per row): per-sample indexing, one
// load at a time -- rare, and written so that it adds no registers to the kernel.  b, f are wave-uniform (scalar
// registers): the address arithmetic is scalar up to the final lane offset.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
template <int LOG2N>
__device__ __forceinline__ void sm_load_frame(const SmArgs& a, int b, int f, int lo, int g, float* __restrict__ x) {
  constexpr IasSmLayout L = ias_sm_layout(1 << LOG2N);
  constexpr int Q = L.Q, NH = Q >= 32 ? Q / 32 : 1;
  const int t0 = f * a.hop - L.N2;
  const float* arow = a.audio + (size_t)b * a.T;
  const int loff = Q >= 32 ? 2 * Q * g + 4 * lo : 2 * Q * g + 2 * lo;
  if (t0 >= 0 && t0 + L.n_fft <= a.T) {
    const float* p = arow + t0 + loff;
    if (Q >= 32) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const f32x4u q = *reinterpret_cast<const f32x4u*>(p + 8 * Q * s + 64 * h);
          x[(s * NH + h) * 4 + 0] = q[0]; x[(s * NH + h) * 4 + 1] = q[1];
          x[(s * NH + h) * 4 + 2] = q[2]; x[(s * NH + h) * 4 + 3] = q[3];
        }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f32x2u q = *reinterpret_cast<const f32x2u*>(p + 8 * Q * s);
        x[2 * s] = q[0]; x[2 * s + 1] = q[1];
      }
    }
  } else {
    // a rolled loop with a select chain per sample (uniform v): slow, rare, and no more live registers than x itself
#pragma unroll 1
    for (int v = 0; v < L.VPL; ++v) {
      const int off = Q >= 32 ? 8 * Q * (v / (4 * NH)) + 64 * ((v / 4) % NH) + (v & 3) : 8 * Q * (v >> 1) + (v & 1);
      const float val = arow[sm_reflect(t0 + loff + off, a.T)];
#pragma unroll
      for (int i = 0; i < L.VPL; ++i) x[i] = v == i ? val : x[i];
    }
  }
}

// frame index -> (row, frame of the row) without an integer division: q0 = hi32(fi * floor(2^32 / F)) is the quotient or
// one below it
__device__ __forceinline__ void sm_row_of(const SmArgs& a, int fi, int& b, int& f) {
  unsigned q0 = __umulhi((unsigned)fi, a.magicF);
  int r = fi - (int)q0 * a.F;
  if (r >= a.F) { r -= a.F; ++q0; }
  b = (int)q0; f = r;
}

// A workgroup of eight waves takes groups of 16 consecutive frames of the flattened [B*F] frame list (a group may
// straddle two rows: frames are independent); wave w transforms frames 2w, 2w+1 of the group.  MEL: the power values of
// the 16 frames meet in LDS (slot = frame of the group) and the waves share the mel tiles of the group.  Groups are
// handed out by a ticket counter (a.ticket): the first gridDim.x groups by blockIdx, the rest in the order the
// workgroups get to them; without mel filters there is nothing to share and every wave draws 2-frame units of its own.
// Loss sums go out per (group, wave) in a fixed order, so the result does not depend on who processed what.
template <int LOG2N, bool MEL, int LOSS /* = a.loss_mode: the MR-STFT sums cost registers and code */,
          bool VEC4 /* MEL: n_out % 4 == 0, a lane's four outputs of a tile are one 16-byte access */>
__global__ __launch_bounds__(64 * IAS_SM_WAVES, 3 * IAS_SM_WAVES / 4) void stft_mfma_kernel(const SmArgs a) {
  constexpr IasSmLayout L = ias_sm_layout(1 << LOG2N);
  constexpr int N2 = L.N2, Q = L.Q, NT = L.NT, NB = L.NB, VPL = L.VPL, NH = Q >= 32 ? Q / 32 : 1, NBH = NB / 2;
  constexpr int PSTR = L.pstr, RB = IAS_SM_REG_BLOCKS, NTHR = 64 * IAS_SM_WAVES;
  constexpr int NLT = L.n_entries;                               // the whole constant block: one LDS copy per workgroup
  static_assert(PSTR >= N2 + 2 && IAS_SM_WAVES * IAS_SM_FPW == 16, "a power slot doubles as the frame's spectrum scratch");
  static_assert((L.e_b1 | L.e_tw1 | L.e_a2 | L.e_tw2 | L.e_unp | NLT) % 4 == 0, "16-byte rows of the LDS constant table");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  f32x4* sT = reinterpret_cast<f32x4*>(smem);                    // [NLT/4][64 lanes] x 4 consecutive entries
  float* sP = smem + 64 * NLT;                                   // MEL: [16 slots][PSTR]; else [8 waves][N2] scratch
  __shared__ int s_next;

  const int tid = threadIdx.x, lane = tid & 63, lo = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // scalar: frame bookkeeping stays off the vector ALU

  // frame-invariant operands and twiddles: lane-major rows of four entries in LDS, read (16 bytes, conflict-free)
  // where a frame needs them -- nothing frame-invariant is held in registers, which is what lets four waves share a SIMD
  for (int i = tid; i < 64 * NLT; i += NTHR) {
    const int l = i & 63, e = i >> 6;
    smem[((e >> 2) * 64 + l) * 4 + (e & 3)] = a.mtab[64 * e + l];
  }
  auto T4 = [&](int e) { return sT[(e >> 2) * 64 + lane]; };      // entries e .. e+3 of this lane (e % 4 == 0)

  if (MEL) {
    for (int i = tid; i < 16 * PSTR; i += NTHR) sP[i] = 0.0f;    // padding and unused slots must stay finite
  }
  __syncthreads();
  // this wave's mel blocks: lane l keeps the descriptor of block l (power offset, flags); the block loop reads them
  // with v_readlane: no memory access and scalar control flow
  const int* desc = reinterpret_cast<const int*>(a.mtab + L.off_desc) + wave * IAS_SM_DESC_W;
  const int mel_t0 = MEL ? __builtin_amdgcn_readfirstlane(desc[2]) : 255;
  const int mel_npass = MEL ? __builtin_amdgcn_readfirstlane(desc[3]) : 0;
  // (the lane's base pointer goes through an empty asm: knowing that its low bits are clear the compiler rewrites
  //  base + lane*4 + k*256 as an OR, loses the immediate-offset form and keeps one 64-bit address per weight load live)
  const float* melA = a.mtab + L.off_mela + (MEL ? 256 * __builtin_amdgcn_readfirstlane(desc[1]) : 0) + lane;
  asm volatile("" : "+v"(melA));
  int d_poff = 0, d_flag = 0;
  if (MEL && lane < __builtin_amdgcn_readfirstlane(desc[0])) { d_poff = desc[4 + 2 * lane]; d_flag = desc[5 + 2 * lane]; }
  const float* Pbase = sP + lo * PSTR + 4 * g;

  // work units: MEL a 16-frame group per workgroup (unit id = group), else 2 frames per wave (unit id = 8 group + wave)
  const int nunits = a.ngroups;
  const int ustep = (int)gridDim.x;
  int unit = (int)blockIdx.x;
  auto first_frame = [&](int u) { return u * 16 + IAS_SM_FPW * wave; };
#ifdef IAS_SM_STAMPS
  int stamp_n = 0;
#endif
  // the frame whose samples are in flight into xn: (row bn, frame fn of the row), flat index fin
  float xn[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) xn[v] = 0.0f;
  int bn = 0, fn = 0;
  int fin = first_frame(unit);
  bool validn = unit < nunits && fin < a.nframes;
  if (validn) { sm_row_of(a, fin, bn, fn); sm_load_frame<LOG2N>(a, bn, fn, lo, g, xn); }
  while (unit < nunits) {
    float l0 = 0.f, l1 = 0.f, l2 = 0.f;
    // the next unit: drawn now, needed when this unit's last frame prefetches its successor
    int unit_next = unit + ustep;
    if (a.ticket != nullptr && tid == 0) s_next = ustep + atomicAdd(a.ticket, 1);      // published by the barrier below
    if (MEL || a.ticket != nullptr) {
      // MEL: the spectrum scratch of a frame is its own power slot: the previous group's mel tiles must have been read by
      // all waves before anybody writes into a slot.  The barrier also publishes the next group's number.
      __syncthreads();
      if (a.ticket != nullptr) unit_next = __builtin_amdgcn_readfirstlane(s_next);
    }
#pragma unroll 1
    for (int q = 0; q < IAS_SM_FPW; ++q) {
      const int fi = fin, brow = bn;
      const bool valid = validn;                              // wave-uniform
      float pk[2][NBH], pn[2][NBH], pmid = 0.f;
      SM_STAMP(1);
      float xw[VPL];
#pragma unroll
      for (int v4 = 0; v4 < VPL / 4; ++v4) {
        const f32x4 w = T4(L.e_win + 4 * v4);
#pragma unroll
        for (int e = 0; e < 4; ++e) xw[4 * v4 + e] = xn[4 * v4 + e] * w[e];
      }
      float* slot = sP + (MEL ? (IAS_SM_FPW * wave + q) * PSTR : wave * N2);
      float pscale = 0.25f;
      f32x4 acc1[NT][2];
      if (valid) {
      if (a.rowpeak != nullptr) {
        const float pkv = a.rowpeak[brow];
        if (pkv > 1.0f) { const float r = 1.0f / pkv; pscale = 0.25f * (r * r); }
      }
      // ---- stage 1: S[n2][k1] on the matrix cores, frame samples as the A operand
      const f32x4 b1c = T4(L.e_b1), b1s = T4(L.e_b1 + 4), b1n = -b1s;
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
      SM_STAMP(2);
      }
      // the next frame of this wave: in flight while the rest of this one is transformed (issued behind stage 1, whose
      // operands then leave the registers the samples arrive in)
      if (q < IAS_SM_FPW - 1) {
        fin = fi + 1; validn = fin < a.nframes;
        if (++fn >= a.F) { fn = 0; ++bn; }
        if (validn) sm_load_frame<LOG2N>(a, bn, fn, lo, g, xn);
      } else {
        fin = first_frame(unit_next);
        validn = unit_next < nunits && fin < a.nframes;
        if (validn) { sm_row_of(a, fin, bn, fn); sm_load_frame<LOG2N>(a, bn, fn, lo, g, xn); }
      }
      if (valid) {
      // ---- twiddle 1: S' = S W_N2^(n2 k1)
      f32x4 sp[NT][2];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {
          const f32x4 tw = T4(L.e_tw1 + 4 * (2 * t + rp));               // (cos, sin) of r = 2 rp, 2 rp + 1
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int r = 2 * rp + e;
            const float sr = acc1[t][0][r], si = acc1[t][1][r], c = tw[2 * e], s_ = tw[2 * e + 1];
            sp[t][0][r] = fmaf(si, s_, sr * c);
            sp[t][1][r] = fmaf(-sr, s_, si * c);
          }
        }
      SM_STAMP(3);
      // ---- stage 2a: radix 8 over a = n2 / NB; S' is the B operand as it stands
      f32x4 acc2[NB];
      const f32x4 a2 = T4(L.e_a2);
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
      SM_STAMP(4);
      // ---- twiddle 2 + stage 2b (radix NB in registers): Z[k], k = k1 + 16 (2 G + kl) + 128 kb.  The lower half
      //      (kb < NB/2) stays in registers, the upper half goes to LDS for the partner lanes.
      cx zl[2][NBH], zu[2][NBH];
      f32x4 tw2[NB - 1];                                         // row b-1: (cos, sin) for kl = 0, (cos, sin) for kl = 1
#pragma unroll
      for (int b = 1; b < NB; ++b) tw2[b - 1] = T4(L.e_tw2 + 4 * (b - 1));
#pragma unroll
      for (int kl = 0; kl < 2; ++kl) {
        cx tv[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float tr = acc2[b][2 * kl], ti = acc2[b][2 * kl + 1];
          if (b == 0) tv[b] = {tr, ti};
          else {
            const float c = tw2[b - 1][2 * kl], s_ = tw2[b - 1][2 * kl + 1];
            tv[b] = {fmaf(ti, s_, tr * c), fmaf(-tr, s_, ti * c)};
          }
        }
        sm_dftR<NB>(tv);
#pragma unroll
        for (int kb = 0; kb < NBH; ++kb) { zl[kl][kb] = tv[kb]; zu[kl][kb] = tv[NBH + kb]; }
      }
      f32x2* sZ = reinterpret_cast<f32x2*>(slot);
#pragma unroll
      for (int kl = 0; kl < 2; ++kl)
#pragma unroll
        for (int kb = 0; kb < NBH; ++kb)
          sZ[lo + 16 * (2 * g + kl) + 128 * kb] = (f32x2){zu[kl][kb].re, zu[kl][kb].im};
      sm_wave_sync();
      SM_STAMP(5);
      // ---- Hermitian unpack of the packed real transform: bins k and N2 - k from Z[k] (own) and Z[N2 - k] (LDS)
#pragma unroll
      for (int kl = 0; kl < 2; ++kl)
#pragma unroll
        for (int kb = 0; kb < NBH; ++kb) {
          const int k = lo + 16 * (2 * g + kl) + 128 * kb;
          const cx zk = zl[kl][kb];
          f32x2 zn = sZ[N2 / 2 - k];                           // (k = 0 reads one element past the upper half)
          if (kl == 0 && kb == 0 && k == 0) zn = (f32x2){zk.re, zk.im};   // Z[N2] = Z[0]
          const int u = kl * NBH + kb;
          const f32x4 un4 = T4(L.e_unp + 4 * (u >> 1));          // (cos, -sin) of u even, (cos, -sin) of u odd
          const float unr = un4[2 * (u & 1)], uni = un4[2 * (u & 1) + 1];
          const float ea = zk.re + zn[0], eb = zk.im - zn[1], od = zk.re - zn[0], os = zk.im + zn[1];
          const float tx = fmaf(uni, od, unr * os), ty = fmaf(-unr, od, uni * os);
          const float xr = ea + tx, xi = eb + ty, yr = ea - tx, yi = eb - ty;
          pk[kl][kb] = fmaf(xi, xi, xr * xr) * pscale;
          pn[kl][kb] = fmaf(yi, yi, yr * yr) * pscale;
        }
      pmid = fmaf(zu[0][0].im, zu[0][0].im, zu[0][0].re * zu[0][0].re) * (4.0f * pscale);   // lane 0: |Z[N2/2]|^2
      if (a.value_mode == 1) {
#pragma unroll
        for (int u = 0; u < NB; ++u) { pk[u / NBH][u % NBH] = sqrtf(pk[u / NBH][u % NBH]); pn[u / NBH][u % NBH] = sqrtf(pn[u / NBH][u % NBH]); }
        pmid = sqrtf(pmid);
      } else if (a.value_mode == 3) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
          pk[u / NBH][u % NBH] = sqrtf(fmaxf(pk[u / NBH][u % NBH], a.eps));
          pn[u / NBH][u % NBH] = sqrtf(fmaxf(pn[u / NBH][u % NBH], a.eps));
        }
        pmid = sqrtf(fmaxf(pmid, a.eps));
      }
      sm_wave_sync();   // every Z read of this frame precedes the power stores / the next frame's Z stores (program order)
      SM_STAMP(6);
      }
      if (MEL) {
        if (valid) {
#pragma unroll
          for (int kl = 0; kl < 2; ++kl)
#pragma unroll
            for (int kb = 0; kb < NBH; ++kb) {
              const int k = lo + 16 * (2 * g + kl) + 128 * kb;
              slot[k] = pk[kl][kb]; slot[N2 - k] = pn[kl][kb];
            }
          if (lane == 0) slot[N2 / 2] = pmid;
        }
      } else if (valid) {
        // linear bins: store / fused loss sums straight from the registers
        const size_t row = (size_t)fi * (N2 + 1);
        float tk[2][NBH], tn[2][NBH], tmid = 0.f;
#pragma unroll
        for (int kl = 0; kl < 2; ++kl)
#pragma unroll
          for (int kb = 0; kb < NBH; ++kb) {
            const int k = lo + 16 * (2 * g + kl) + 128 * kb;
            tk[kl][kb] = LOSS != 0 ? a.target[row + k] : 0.f;
            tn[kl][kb] = LOSS != 0 ? a.target[row + N2 - k] : 0.f;
            if (a.out != nullptr) { a.out[row + k] = pk[kl][kb]; a.out[row + N2 - k] = pn[kl][kb]; }
          }
        if (lane == 0) {
          if (LOSS != 0) tmid = a.target[row + N2 / 2];
          if (a.out != nullptr) a.out[row + N2 / 2] = pmid;
        }
        if (LOSS == 1) {
#pragma unroll
          for (int kl = 0; kl < 2; ++kl)
#pragma unroll
            for (int kb = 0; kb < NBH; ++kb) l0 += fabsf(pk[kl][kb] - tk[kl][kb]) + fabsf(pn[kl][kb] - tn[kl][kb]);
          if (lane == 0) l0 += fabsf(pmid - tmid);
        } else if (LOSS == 2) {
          auto mr = [&](float v, float t) {
            const float d = t - v; l0 = fmaf(d, d, l0); l1 = fmaf(t, t, l1); l2 += fabsf(logf(v) - logf(t));
          };
#pragma unroll
          for (int kl = 0; kl < 2; ++kl)
#pragma unroll
            for (int kb = 0; kb < NBH; ++kb) { mr(pk[kl][kb], tk[kl][kb]); mr(pn[kl][kb], tn[kl][kb]); }
          if (lane == 0) mr(pmid, tmid);
        }
      }
      SM_STAMP(7);
    }
    if (MEL) {
      // mel tiles of this wave over the 16 frames of the group.  A operand: the filter weights of the wave's blocks,
      // fetched (L2) into registers that are dead between two frames, before the barrier they do not depend on;
      // B operand: power values read from the slots as 16-byte rows, one block ahead; the target rows of a tile are
      // fetched while the tile before it is computed.
      SM_STAMP(8);
      const int fj = unit * 16 + lo;                     // the frame in column lo of the tile
      const bool okj = fj < a.nframes;
      auto load_tg = [&](int tile) {
        f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int m0 = 16 * tile + 4 * g;
        if (LOSS != 0 && okj && tile != 255) {
          const size_t orow = (size_t)fj * a.n_out + m0;
          if (VEC4) { if (m0 < a.n_out) t = *reinterpret_cast<const f32x4*>(a.target + orow); }
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (m0 + r < a.n_out) t[r] = a.target[orow + r];
          }
        }
        return t;
      };
      f32x4 pv = (f32x4){0.f, 0.f, 0.f, 0.f}, acc = pv, tg = pv, tg_next = pv;
      // passes of RB blocks (one for the filterbanks this kernel is tuned for; more for wide filters, whose later
      // passes wait for their weights)
      for (int pass = 0; pass < mel_npass; ++pass) {
        float wreg[RB][4];
        // one opaque base per 4 KB of weights: every load is base + immediate (< 4096), three address registers in all
        const float* mel_chunk[(RB + 3) / 4];
#pragma unroll
        for (int c = 0; c < (RB + 3) / 4; ++c) { mel_chunk[c] = melA + 256 * RB * pass + 1024 * c; asm volatile("" : "+v"(mel_chunk[c])); }
#pragma unroll
        for (int bi = 0; bi < RB; ++bi)
#pragma unroll
          for (int s = 0; s < 4; ++s) wreg[bi][s] = mel_chunk[bi >> 2][256 * (bi & 3) + 64 * s];
        if (pass == 0) {
          tg_next = load_tg(mel_t0);
          __syncthreads();                               // the 16 slots of the group are complete
          SM_STAMP(9);
          pv = *reinterpret_cast<const f32x4*>(Pbase + __builtin_amdgcn_readlane(d_poff, 0));
        }
#pragma unroll
        for (int bi = 0; bi < RB; ++bi) {                // (blocks past the wave's last: no flags, nothing is emitted)
          const int bl = pass * RB + bi;
          const int flags = __builtin_amdgcn_readlane(d_flag, bl);
          const f32x4 pvn = *reinterpret_cast<const f32x4*>(Pbase + __builtin_amdgcn_readlane(d_poff, bl + 1));
          if (flags & 0x100) { tg = tg_next; tg_next = load_tg((flags >> 16) & 255); }
          acc = sm_mfma(wreg[bi][0], pv[0], acc);
          acc = sm_mfma(wreg[bi][1], pv[1], acc);
          acc = sm_mfma(wreg[bi][2], pv[2], acc);
          acc = sm_mfma(wreg[bi][3], pv[3], acc);
          if (flags & 0x200) {
            const int m0 = 16 * (flags & 255) + 4 * g;
            if (okj && ((flags >> (10 + g)) & 1)) {      // this lane's rows belong to the (sub)tile
              const size_t orow = (size_t)fj * a.n_out + m0;
              if (a.out != nullptr) {
                if (VEC4) { if (m0 < a.n_out) *reinterpret_cast<f32x4*>(a.out + orow) = acc; }
                else {
#pragma unroll
                  for (int r = 0; r < 4; ++r) if (m0 + r < a.n_out) a.out[orow + r] = acc[r];
                }
              }
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (m0 + r < a.n_out) {
                  const float v = acc[r], tt = tg[r];
                  if (LOSS == 1) l0 += fabsf(v - tt);
                  else if (LOSS == 2) {
                    const float d = tt - v; l0 = fmaf(d, d, l0); l1 = fmaf(tt, tt, l1); l2 += fabsf(logf(v) - logf(tt));
                  }
                }
              }
            }
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};
          }
          pv = pvn;
        }
      }
      SM_STAMP(10);
    }
    // loss sums of this (group, wave): one fixed-order record
    if (LOSS != 0 && a.partials != nullptr) {
      l0 = sm_wave_sum(l0);
      if (LOSS == 2) { l1 = sm_wave_sum(l1); l2 = sm_wave_sum(l2); }
      if (lane == 0) {
        double* rec = a.partials + (size_t)(unit * IAS_SM_WAVES + wave) * 3;
        rec[0] = (double)l0; rec[1] = (double)l1; rec[2] = (double)l2;
      }
    }
    unit = unit_next;
  }
#ifdef IAS_SM_STAMPS
  if (a.stamps != nullptr && lane == 0) a.stamps[((size_t)blockIdx.x * IAS_SM_WAVES + wave) * 256] = stamp_n;
#endif
  // the last one out re-arms the ticket counter for the next launch (everybody has drawn its final ticket by then)
  if (a.ticket != nullptr && tid == 0) {
    if (atomicAdd(a.ticket + 1, 1) == (int)gridDim.x - 1) { atomicExch(a.ticket, 0); atomicExch(a.ticket + 1, 0); }
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
// workgroups of the persistent grid: three per CU (LDS, <= 168 VGPRs), never more than there are 16-frame groups
int ias_sm_grid(long long nframes) {
  static const int env = ias_diag_env("IAS_STFT_MFMA_WGS") ? atoi(ias_diag_env("IAS_STFT_MFMA_WGS")) : 0;   // diagnostics
  const long long ngroups = (nframes + 15) / 16;
  const long long cap = env > 0 ? env : 3LL * sm_num_cus();
  return (int)std::min(ngroups, cap);
}
// loss partial records: one per (16-frame group, wave)
long long ias_sm_partials(long long nframes) { return ((nframes + 15) / 16) * IAS_SM_WAVES; }
// Opt-in (IAS_STFT_MFMA=1): measured on MI355X, v_mfma_f32_16x16x4_f32 does not run beside vector instructions -- not
// from another wave of the SIMD and not from its own wave (scripts/diag/mfma_valu_overlap.hip, mfma_valu_inwave.hip:
// an MFMA stream and an FMA stream on one SIMD take the SUM of their times) -- so the 4x larger multiply count of a
// DFT-as-GEMM is paid in full on the vector ALU and this kernel (85-90 us at the headline size) loses to the radix-8
// FFT kernel (csrc/spectral_kernels.hip).  Kept as the measured record of that experiment and for its tests.
bool ias_sm_enabled(int n_fft, bool have_mtables) {
  static const int on = ias_diag_env("IAS_STFT_MFMA") ? atoi(ias_diag_env("IAS_STFT_MFMA")) : 0;
  return on && have_mtables && n_fft == 1024;
}

#ifdef IAS_SM_STAMPS
static unsigned long long* g_sm_stamps = nullptr;
// diagnostics build only: device buffer [grid][4][256] u64 that the next launches stamp into (NULL: off)
extern "C" int ias_stft_set_stamps(void* p) { g_sm_stamps = (unsigned long long*)p; return IAS_OK; }
#endif

int ias_sm_launch(const float* audio, const float* mtab, bool mel, float* out, const float* target, double* partials,
                  const float* rowpeak, int* ticket, int B, int T, int F, int n_fft, int hop, int n_out, int value_mode,
                  int loss_mode, float eps, hipStream_t stream) {
  if ((long long)B * F > 2000000000LL) return IAS_ERR_UNSUPPORTED;
  if (mel && n_out > 16 * IAS_SM_MAX_TILES) return IAS_ERR_UNSUPPORTED;
  SmArgs a;
  a.audio = audio; a.mtab = mtab; a.out = out; a.target = target; a.partials = partials; a.rowpeak = rowpeak;
  a.T = T; a.F = F; a.hop = hop; a.n_out = n_out; a.nframes = B * F; a.ngroups = (a.nframes + 15) / 16;
  static const int noticket = ias_diag_env("IAS_STFT_NOTICKET") ? atoi(ias_diag_env("IAS_STFT_NOTICKET")) : 0;   // diagnostics
  a.ticket = noticket ? nullptr : ticket;
  a.magicF = (F == 1 ? 0xFFFFFFFFu /* 2^32 / 1 does not fit: q0 = fi - 1, which row_of's one-step correction fixes */ : (unsigned)(0x100000000ULL / (unsigned long long)F));
  a.value_mode = value_mode; a.loss_mode = loss_mode; a.eps = eps;
#ifdef IAS_SM_STAMPS
  a.stamps = g_sm_stamps;
#endif
  const IasSmLayout L = ias_sm_layout(n_fft);
  const size_t lds = sizeof(float) * (64 * L.n_entries + (mel ? 16 * L.pstr : L.N2 * IAS_SM_WAVES));
  const dim3 grid(ias_sm_grid(a.nframes)), block(64 * IAS_SM_WAVES);
#define IAS_SM_LAUNCH1(LOG2N, MEL, LOSS, VEC4)                                                                         \
  do {                                                                                                             \
    if (lds > 64 * 1024)                                                                                           \
      (void)hipFuncSetAttribute((const void*)stft_mfma_kernel<LOG2N, MEL, LOSS, VEC4>,                             \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
    hipLaunchKernelGGL((stft_mfma_kernel<LOG2N, MEL, LOSS, VEC4>), grid, block, lds, stream, a);                   \
  } while (0)
#define IAS_SM_LAUNCH(LOG2N, MEL, VEC4)                                                                            \
  do {                                                                                                             \
    if (loss_mode == 0) IAS_SM_LAUNCH1(LOG2N, MEL, 0, VEC4);                                                       \
    else if (loss_mode == 1) IAS_SM_LAUNCH1(LOG2N, MEL, 1, VEC4);                                                  \
    else IAS_SM_LAUNCH1(LOG2N, MEL, 2, VEC4);                                                                      \
  } while (0)
  const bool vec4 = mel && (n_out & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(target) & 15) == 0;
  if (n_fft == 1024) {
    if (!mel) IAS_SM_LAUNCH(10, false, false);
    else if (vec4) IAS_SM_LAUNCH(10, true, true);
    else IAS_SM_LAUNCH(10, true, false);
  }
  else return IAS_ERR_UNSUPPORTED;
#undef IAS_SM_LAUNCH1
#undef IAS_SM_LAUNCH
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
#else   // product library: the matrix-core STFT is not built
bool ias_sm_enabled(int, bool) { return false; }
long long ias_sm_partials(long long nframes) { return ((nframes + 15) / 16) * IAS_SM_WAVES; }
int ias_sm_launch(const float*, const float*, bool, float*, const float*, double*, const float*, int*, int, int, int, int, int, int,
                  int, int, float, hipStream_t) {
  return IAS_ERR_UNSUPPORTED;
}
#endif

