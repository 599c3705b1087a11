// Framed STFT -> power / magnitude -> (optional) mel filterbank -> spectrogram or fused L1 / MR-STFT
// partial sums, for MI355X (gfx950).
//
// The reference has no live spectral-loss code; the spec is the commented mel block
// /root/reference/conf/config.yaml:51-61 and its use /root/reference/audio_to_params.py:150-153
// (torchaudio MelSpectrogram semantics: hann window, center=True, reflect pad, onesided, power 2,
// slaney-normalised htk mel filterbank), plus the auraloss TODOs (audio_to_params.py:233).
//
// One wave computes one frame: the n_fft real samples are packed into n_fft/2 complex points,
// transformed by an in-register radix-R pass (R = n_fft/128) followed by two radix-8 passes that
// exchange through LDS (n_fft/2 = R*8*8), then unpacked to the n_fft/2+1 one-sided bins.  All
// arithmetic fp32 (no bf16 DFT-as-GEMM: it would not hold the 1e-3 loss tolerance).
// Frames are read straight from global memory (8 B per lane, 512 B per wave instruction); frames
// overlap, so the second touch of a sample is served by L1/L2 and every audio sample comes from HBM
// ~once: algorithmic bytes = 4 B in per sample + 4*bins/hop B out (or the same to read a cached target).
#include "ias_common.h"
#include <cstdlib>

// Waves per workgroup.  The per-lane tables (14.8 KB at n_fft 1024) are one LDS copy per workgroup and a wave's FFT
// scratch is 4.6 KB: with 4 waves a workgroup needs 36 KB and a CU holds 4 of them = 4 waves per SIMD; with 10 waves it
// needs 66 KB and a CU holds 2 = 5 waves per SIMD (the kernel is bound by the latency of its LDS round trips, five per
// frame, not by a unit: VALU and LDS are each ~50 % busy).  n_fft 1024 (<= 96 VGPRs) takes the 10-wave form.
// n_fft 2048: 9.2 KB of scratch per wave and 29 KB of tables: four-wave workgroups fit two to a CU (2 waves per SIMD
// for a kernel bound by LDS round-trip latency); one 12-wave workgroup per CU gives 3 (the 164 VGPRs allow no more).
#define SP_WAVES_MAX 12
static int stft_waves(int n_fft) {
  static const int env = ias_diag_env("IAS_STFT_WAVES") ? atoi(ias_diag_env("IAS_STFT_WAVES")) : 0;   // diagnostics: 4, 8, 10 or 12
  if (n_fft == 1024) return (env == 8 || env == 10) ? env : 4;
  if (n_fft == 2048) return env == 4 ? 4 : 12;
  return 4;
}

// Complex numbers as 2-wide vectors: complex add/sub are one packed op and a complex multiply is pk_mul +
// pk_fma (a packed fp32 op costs the SIMD 4 clocks, two plain ones 2 + 2: the same arithmetic time, fewer
// instructions to issue).  8-byte aligned: LDS accesses are ds_*_b64.
typedef float cpx __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// floats of the re-packed mel weight table in LDS (every filter padded to a multiple of four taps), rounded to 16 bytes
__host__ __device__ static inline int mel_padded_words(int mel_nnz, int n_out) {
  return n_out > 0 ? ((mel_nnz + 3 * n_out + 3) & ~3) : 0;
}
__device__ __forceinline__ cpx cmk(float x, float y) { return (cpx){x, y}; }
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return a + b; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return a - b; }
__device__ __forceinline__ cpx cmul(cpx a, cpx b) {
  return __builtin_elementwise_fma((cpx){-a.y, a.y}, (cpx){b.y, b.x}, (cpx){a.x, a.x} * b);
}
__device__ __forceinline__ cpx cmul_negi(cpx a) { return (cpx){a.y, -a.x}; }  // a * (-i)

// forward DFTs (kernel e^{-2 pi i nk/R}), in place, natural order
__device__ __forceinline__ void dft4(cpx& v0, cpx& v1, cpx& v2, cpx& v3) {
  const cpx t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3), t3 = cmul_negi(csub(v1, v3));
  v0 = cadd(t0, t2); v1 = cadd(t1, t3); v2 = csub(t0, t2); v3 = csub(t1, t3);
}
__device__ __forceinline__ void dft8(cpx* v) {
  cpx e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  cpx o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4(e0, e1, e2, e3);
  dft4(o0, o1, o2, o3);
  const float h = 0.70710678118654752f;
  o1 = (o1 + cmul_negi(o1)) * h;                      // * W8^1 = (1 - i)/sqrt2
  o2 = cmul_negi(o2);                                 // * W8^2 = -i
  o3 = (cmul_negi(o3) - o3) * h;                      // * W8^3 = (-1 - i)/sqrt2
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
  v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
  v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}
__device__ __forceinline__ void dft16(cpx* v) {
  cpx e[8], o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
  dft8(e);
  dft8(o);
  // W16^k, k = 0..7
  const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
  const cpx w[8] = {cmk(1.f, 0.f), cmk(c1, -s1), cmk(h, -h), cmk(s1, -c1),
                    cmk(0.f, -1.f), cmk(-s1, -c1), cmk(-h, -h), cmk(-c1, -s1)};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const cpx t = cmul(o[k], w[k]);
    v[k] = cadd(e[k], t);
    v[k + 8] = csub(e[k], t);
  }
}
template <int R> __device__ __forceinline__ void dftR(cpx* v);
template <> __device__ __forceinline__ void dftR<4>(cpx* v) { dft4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void dftR<8>(cpx* v) { dft8(v); }
template <> __device__ __forceinline__ void dftR<16>(cpx* v) { dft16(v); }

__device__ __forceinline__ int reflect_index(int i, int T) {
  if (i < 0) i = -i;
  if (i >= T) i = 2 * (T - 1) - i;
  return i;
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

struct SpecArgs {
  const float* audio;      // [B,T]
  const float* tables;     // lane-major window / twiddle tables (ias_stft_build_tables)
  const int* mel_start;    // [n_out] first bin of each filter (mel mode) or null (raw bins)
  const int* mel_count;    // [n_out]
  const int* mel_woff;     // [n_out] offset into mel_w
  const float* mel_w;      // packed non-zero filter weights
  float* out;              // [B,F,n_out] or null
  const float* target;     // [B,F,n_out] or null
  double* partials;        // [gridDim.x*gridDim.y][3] or null
  const float* rowpeak;    // [B] or null: row peaks; the spectrum is that of row / peak when peak > 1 (|X|^2 scales by 1/peak^2)
  int T, F, hop, n_out, mel_nnz;
  int groups;              // consecutive frames of a row each workgroup walks through
  int value_mode;          // 1: |X|, 2: |X|^2, 3: sqrt(max(|X|^2, eps))
  int loss_mode;           // 0: none, 1: sum |v - t|, 2: MR-STFT sums {(t-v)^2, t^2, |log v - log t|}
  float eps;
};

// LDS traffic inside one wave only needs ORDERING, not a workgroup barrier and not a wait: the waves of a workgroup
// work on different frames with private scratch, and the LDS instructions of one wave execute in program order
// (LLVM's AMDGPU memory model puts no s_waitcnt on a wavefront-scope fence for that reason).  So a hand-over between
// lanes of the wave is a compiler-level fence; the s_waitcnt lgkmcnt(0) that used to sit here drained the wave's LDS
// queue eight times per frame.  The compiler still waits, with exact counts, where a loaded VALUE is consumed.
__device__ __forceinline__ void wave_lds_sync() {
#ifdef IAS_STFT_DRAIN_SYNC   /* diagnostics: the round-1 form */
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// ---- register <-> lane-bits transpose (round 4): the first exchange of the radix-8 kernels without LDS.
// Pass 1 leaves element (k1, a, c) in register k1 of lane 8 a + c; pass 2 wants it in register a of lane 8 k1 + c: an
// 8 x 8 transpose between the register index and lane bits 3..5.  Bit by bit: register bit 0 <-> lane bit 3 (distance 8
// inside a row of 16: two bank-masked DPP row rotations and a copy per register pair), bit 1 <-> lane bit 4
// (v_permlane16_swap: rows 1, 3 of one register against rows 0, 2 of the other), bit 2 <-> lane bit 5
// (v_permlane32_swap).  32 + 8 + 8 = 48 VALU instructions for the 8 complex values of a lane, against 8 ds_write_b64
// (6 LDS cycles each) + 8 ds_read_b64 and two LDS round trips of latency (checked lane by lane against the LDS transpose:
// scripts/diag/permlane_swap.hip).  The second exchange transposes the register index with lane bits 0..2, for which
// gfx950 has no swap instruction (three DPP instructions per pair and level: 72): it stays in LDS.
typedef unsigned x_u2 __attribute__((ext_vector_type(2)));
template <int L> __device__ __forceinline__ void xchg_pair(float& f0, float& f1) {
  unsigned r0 = __float_as_uint(f0), r1 = __float_as_uint(f1);
  if (L == 5) { const x_u2 t = __builtin_amdgcn_permlane32_swap(r0, r1, false, false); r0 = t.x; r1 = t.y; }
  else if (L == 4) { const x_u2 t = __builtin_amdgcn_permlane16_swap(r0, r1, false, false); r0 = t.x; r1 = t.y; }
  else {
    // lanes with bit 3 set (banks 2, 3 of a row of 16): r0 <- r1 of lane ^ 8; lanes with bit 3 clear: r1 <- r0 of lane ^ 8
    const unsigned old0 = r0;
    r0 = __builtin_amdgcn_update_dpp(r0, r1, 0x128 /* row_ror:8 */, 0xf, 0xc, false);
    r1 = __builtin_amdgcn_update_dpp(r1, old0, 0x128, 0xf, 0x3, false);
  }
  f0 = __uint_as_float(r0); f1 = __uint_as_float(r1);
}
// u[k1] of lane (a, c)  ->  u[a] of lane (k1, c)
__device__ __forceinline__ void xchg_reg_lane345(cpx (&u)[8]) {
#pragma unroll
  for (int comp = 0; comp < 2; ++comp) {
    float f[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = comp ? u[q].y : u[q].x;
    xchg_pair<3>(f[0], f[1]); xchg_pair<3>(f[2], f[3]); xchg_pair<3>(f[4], f[5]); xchg_pair<3>(f[6], f[7]);
    xchg_pair<4>(f[0], f[2]); xchg_pair<4>(f[1], f[3]); xchg_pair<4>(f[4], f[6]); xchg_pair<4>(f[5], f[7]);
    xchg_pair<5>(f[0], f[4]); xchg_pair<5>(f[1], f[5]); xchg_pair<5>(f[2], f[6]); xchg_pair<5>(f[3], f[7]);
#pragma unroll
    for (int q = 0; q < 8; ++q) { if (comp) u[q].y = f[q]; else u[q].x = f[q]; }
  }
}
#ifndef IAS_S2_REGX
#define IAS_S2_REGX 1      // stft2_kernel: first exchange in registers (0: both through LDS, the round-3 form)
#endif

// Loads the 2R samples of frame f that lane `lane` owns in pass 1 (points 64*n1 + lane, n1 < R).
// Interior frames on 8-byte-aligned rows: R coalesced 8-byte loads (512 B per wave instruction);
// frames that touch the row ends (reflect padding) or odd alignments: per-sample indexing.
template <int R, int N2>
__device__ __forceinline__ void load_frame(const float* __restrict__ arow, int T, int hop, int f, int lane,
                                           float (&x)[2 * R]) {
  const int g0 = f * hop - N2;
  if (g0 >= 0 && g0 + 2 * N2 <= T && (((g0 | T) & 1) == 0)) {
    // interior frame: R 8-byte loads off ONE address register pair (immediate offsets 512 n1).  The pointer is made
    // opaque and the edge path's values pass through empty asm statements: left alone, the compiler sinks the loads of
    // both paths into one block of 2 R single-dword loads from 2 R separate 64-bit addresses (32 address VGPRs, twice
    // the memory instructions) -- for every frame, to share code with the two edge frames per row.
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(1))) f32x2_t* glb_f2p;    // stays a GLOBAL pointer through the asm
    glb_f2p p = (glb_f2p)(reinterpret_cast<const f32x2_t*>(arow + g0) + lane);
#ifndef IAS_LOADFRAME_PLAIN
    asm volatile("" : "+v"(p));
#endif
#pragma unroll
    for (int n1 = 0; n1 < R; ++n1) { const f32x2_t q = p[64 * n1]; x[2 * n1] = q[0]; x[2 * n1 + 1] = q[1]; }
  } else {
#pragma unroll
    for (int n1 = 0; n1 < R; ++n1) {
      const int m = g0 + 2 * (64 * n1 + lane);
      x[2 * n1] = arow[reflect_index(m, T)];
      x[2 * n1 + 1] = arow[reflect_index(m + 1, T)];
#ifndef IAS_LOADFRAME_PLAIN
      asm volatile("" : "+v"(x[2 * n1]), "+v"(x[2 * n1 + 1]));
#endif
    }
  }
}

// One wave per frame, no staging: every wave reads its frame straight from global memory (frames
// overlap, so the second touch of a sample is an L1/L2 hit), prefetching the next frame while it
// transforms the current one.  A workgroup (SP_WAVES waves) walks through a.groups consecutive frames
// of one row, wave w taking frames w, w+4, ...; LDS holds only the per-wave FFT scratch and the mel
// tables, and there is no workgroup barrier in the frame loop.
template <int LOG2N, int SP_WAVES, bool LOSS2 /* a.loss_mode == 2: its two extra accumulators cost registers */>
__global__ __launch_bounds__(64 * SP_WAVES, SP_WAVES == 10 ? 5 : 1) void stft_kernel(const SpecArgs a) {
  constexpr int SP_THREADS = 64 * SP_WAVES;
  constexpr int NFFT = 1 << LOG2N, N2 = NFFT / 2, R = N2 / 64, NPAIR = 8 * R, NP_IT = (NPAIR + 63) / 64;
  constexpr int SCR = NPAIR * 9;          // padded [k1*8+c][9] complex scratch, reused in place by every pass
  constexpr int NUNP = (N2 / 2) / 64 + 1; // unpack iterations per lane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  cpx* s_scr = reinterpret_cast<cpx*>(smem);                           // SP_WAVES * SCR
  // mel weights, re-packed per filter to a multiple of four taps (zero padded) at 16-byte aligned offsets, so that a
  // trip of the filter loop reads its four weights as ONE ds_read_b128 and needs no per-tap predicate
  float* s_melw = reinterpret_cast<float*>(s_scr + SP_WAVES * SCR);   // <= a.mel_nnz + 3 n_out floats (mel mode)
  const int melw_words = mel_padded_words(a.mel_nnz, a.mel_start != nullptr ? a.n_out : 0);
  int* s_meli = reinterpret_cast<int*>(s_melw + melw_words);   // [3][n_out]: start, padded count, padded offset

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const float* arow = a.audio + (size_t)b * a.T;
  const bool mel = a.mel_start != nullptr;
  // normalize_if_clipping folded in: |FFT(x / peak)|^2 = |FFT(x)|^2 / peak^2
  float pscale = 1.0f;
  if (a.rowpeak != nullptr) { const float pkv = a.rowpeak[b]; if (pkv > 1.0f) { const float r = 1.0f / pkv; pscale = r * r; } }

  if (mel) {
    for (int i = tid; i < melw_words; i += SP_THREADS) s_melw[i] = 0.0f;
    for (int i = tid; i < a.n_out; i += SP_THREADS) {
      s_meli[i] = a.mel_start[i];
      s_meli[a.n_out + i] = (a.mel_count[i] + 3) & ~3;
    }
    __syncthreads();
    for (int i = tid; i < a.n_out; i += SP_THREADS) {
      int off = 0;                                   // padded offset: sum of the padded counts before filter i
      for (int m = 0; m < i; ++m) off += s_meli[a.n_out + m];
      s_meli[2 * a.n_out + i] = off;
    }
    __syncthreads();
    for (int i = tid; i < a.n_out; i += SP_THREADS) {
      const int n = a.mel_count[i], src = a.mel_woff[i], dst = s_meli[2 * a.n_out + i];
      for (int j = 0; j < n; ++j) s_melw[dst + j] = a.mel_w[src + j];
    }
  }

  // frame-invariant per-lane tables (window of the lane's 2R samples, pass-1 twiddles W_N2^(lane*k1),
  // pass-2 twiddles W_64^(c*d), unpack twiddles W_NFFT^k): one lane-major copy per workgroup in LDS.
  // Holding them in registers cost 58 VGPRs and capped the kernel at 2-3 waves per SIMD, too few to
  // cover the ~2900-cycle global-load latency measured per frame; conflict-free ds_read_b32 instead
  // (166 -> 86 VGPRs, 120 -> 104 us).
  constexpr int NTAB = 64 * (2 * R + 2 * R + 16 * NP_IT + 2 * NUNP);
  // (the global block is [entry][lane]; the LDS copy interleaves entries 2p, 2p+1 per lane so that a window
  // pair or a (cos, -sin) twiddle is one conflict-free ds_read_b64)
  cpx* s_tab = reinterpret_cast<cpx*>(s_meli + ((3 * (mel ? a.n_out : 0) + 1) & ~1));
  for (int i = tid; i < NTAB / 2; i += SP_THREADS) {
    const int pp = i >> 6, l = i & 63;
    s_tab[i] = cmk(a.tables[64 * (2 * pp) + l], a.tables[64 * (2 * pp + 1) + l]);
  }
  const cpx* t_win = s_tab + lane;          // [n1]   -> (win[2 n1], win[2 n1 + 1])
  const cpx* t_tw1 = t_win + 64 * R;        // [k1]   -> W_N2^(lane k1)
  const cpx* t_tw2 = t_tw1 + 64 * R;        // [8i+d] -> W_64^(c d)
  const cpx* t_twu = t_tw2 + 64 * 8 * NP_IT;  // [i]  -> W_NFFT^(lane + 64 i)
  __syncthreads();   // mel tables visible; the only workgroup barrier before the final reduction

  cpx* sA = s_scr + wave * SCR;
  float l0 = 0.f, l1 = 0.f, l2 = 0.f;
  // wave-wide longest filter of each output group of the first 128 outputs (wave-uniform, frame-invariant).  The
  // per-lane filter descriptors themselves are re-read from LDS every frame: holding them cost six VGPRs, which at 10
  // waves per workgroup is the difference between 5 waves per SIMD and spilling.
  int h_nmaxa = 0, h_nmaxb = 0;
  if (mel) {
    if (lane < a.n_out) h_nmaxa = s_meli[a.n_out + lane];
    if (lane + 64 < a.n_out) h_nmaxb = s_meli[a.n_out + lane + 64];
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    h_nmaxa = max(h_nmaxa, __shfl_xor(h_nmaxa, d, 64));
    h_nmaxb = max(h_nmaxb, __shfl_xor(h_nmaxb, d, 64));
  }
  h_nmaxa = __builtin_amdgcn_readfirstlane(h_nmaxa);
  h_nmaxb = __builtin_amdgcn_readfirstlane(h_nmaxb);
  const int f_begin = blockIdx.x * a.groups;
  const int f_end = min(f_begin + a.groups, a.F);

  float xc[2 * R], xn[2 * R];
  int f = f_begin + wave;
  if (f < f_end) load_frame<R, N2>(arow, a.T, a.hop, f, lane, xc);

  for (; f < f_end; f += SP_WAVES) {
    const bool more = f + SP_WAVES < f_end;   // wave-uniform
    if (more) load_frame<R, N2>(arow, a.T, a.hop, f + SP_WAVES, lane, xn);

    // pass 1: radix-R over n1 (points 64*n1 + lane), twiddle W_N2^(lane*k1), scatter to [k1][c][a]
    cpx v[R];
#pragma unroll
    for (int n1 = 0; n1 < R; ++n1)
      v[n1] = cmk(xc[2 * n1], xc[2 * n1 + 1]) * t_win[64 * n1];
    dftR<R>(v);
    {
      const int c = lane & 7, aa = lane >> 3;
#pragma unroll
      for (int k1 = 0; k1 < R; ++k1)
        sA[(k1 * 8 + c) * 9 + aa] = cmul(v[k1], t_tw1[64 * k1]);
    }
    wave_lds_sync();
    // pass 2: radix-8 over a for each (k1, c); twiddle W_64^(c*d); scatter (in place) to [k1][d][c]
    cpx u[NP_IT][8];
#pragma unroll
    for (int i = 0; i < NP_IT; ++i) {
      const int p = lane + 64 * i;
      if (p < NPAIR) {
#pragma unroll
        for (int q = 0; q < 8; ++q) u[i][q] = sA[p * 9 + q];
      }
    }
    wave_lds_sync();   // every read of the [k1][c][a] image is in registers before it is overwritten
#pragma unroll
    for (int i = 0; i < NP_IT; ++i) {
      const int p = lane + 64 * i;
      if (p < NPAIR) {
        const int k1 = p >> 3, c = p & 7;
        dft8(u[i]);
#pragma unroll
        for (int d = 0; d < 8; ++d)
          sA[(k1 * 8 + d) * 9 + c] = cmul(u[i][d], t_tw2[64 * (8 * i + d)]);
      }
    }
    wave_lds_sync();
    // pass 3: radix-8 over c for each (k1, d) -> Z[k1 + R*d + 8R*e], natural order padded by one
    // complex per 8 (position k + k/8: the 16 lanes of a ds_write_b64 group then hit distinct banks)
#pragma unroll
    for (int i = 0; i < NP_IT; ++i) {
      const int p = lane + 64 * i;
      if (p < NPAIR) {
#pragma unroll
        for (int q = 0; q < 8; ++q) u[i][q] = sA[p * 9 + q];
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NP_IT; ++i) {
      const int p = lane + 64 * i;
      if (p < NPAIR) {
        const int k1 = p >> 3, d = p & 7;
        dft8(u[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int k = k1 + R * d + 8 * R * e; sA[k + (k >> 3)] = u[i][e]; }
      }
    }
    wave_lds_sync();
    // unpack the packed real FFT: bins k and N2-k from Z[k], Z[N2-k]; values written as floats P[0..N2]
    float pk[NUNP], pn[NUNP];
#pragma unroll
    for (int i = 0; i < NUNP; ++i) {
      const int k = lane + 64 * i;
      pk[i] = pn[i] = 0.f;
      if (k <= N2 / 2) {
        const int kn = (N2 - k) & (N2 - 1);
        const cpx zk = sA[k + (k >> 3)], zn = sA[kn + (kn >> 3)];
        const cpx ze = cmk(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const cpx zo = cmk(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        const cpx t = cmul(t_twu[64 * i], zo);
        const cpx xk = cadd(ze, t);                     // X[k]
        const cpx xq = cmk(ze.x - t.x, -(ze.y - t.y));  // X[N2-k] = conj(ze - t)
        float vk = xk.x * xk.x + xk.y * xk.y, vn = xq.x * xq.x + xq.y * xq.y;
        if (a.rowpeak != nullptr) { vk *= pscale; vn *= pscale; }
        if (a.value_mode == 1) { vk = __builtin_amdgcn_sqrtf(vk); vn = __builtin_amdgcn_sqrtf(vn); }   // v_sqrt_f32, 1 ulp
        else if (a.value_mode == 3) { vk = __builtin_amdgcn_sqrtf(fmaxf(vk, a.eps)); vn = __builtin_amdgcn_sqrtf(fmaxf(vn, a.eps)); }
        pk[i] = vk; pn[i] = vn;
      }
    }
    wave_lds_sync();   // all Z reads done before P overwrites the scratch
    float* P = reinterpret_cast<float*>(sA);
#pragma unroll
    for (int i = 0; i < NUNP; ++i) {
      const int k = lane + 64 * i;
      if (k <= N2 / 2) { P[k] = pk[i]; P[N2 - k] = pn[i]; }
    }
    wave_lds_sync();
    // epilogue: mel projection (or raw bins), store / fused loss sums
    {
      const size_t row = ((size_t)b * a.F + f) * a.n_out;
      for (int m0 = 0; m0 < a.n_out; m0 += 128) {
        // two outputs per lane (m0 + lane, m0 + 64 + lane) advance together, 4 taps per trip, all LDS
        // reads of a trip issued before the first use: the filter loop is latency-, not work-bound
        const int ma = m0 + lane, mb = m0 + 64 + lane;
        const bool oka = ma < a.n_out, okb = mb < a.n_out;
        float va = 0.f, vb = 0.f;
        if (mel) {
          const int sa = oka ? s_meli[ma] : 0, na = oka ? s_meli[a.n_out + ma] : 0, woa = oka ? s_meli[2 * a.n_out + ma] : 0;
          const int sb = okb ? s_meli[mb] : 0, nb = okb ? s_meli[a.n_out + mb] : 0, wob = okb ? s_meli[2 * a.n_out + mb] : 0;
          const float* wa = s_melw + woa;
          const float* wb = s_melw + wob;
          // the two outputs of a lane run as two loops with their own wave-wide trip counts: with mel filters the
          // lower 64 outputs are a fifth as wide as the upper 64 (5 vs 27 bins at 128 mels / 513 bins), so a
          // common loop spends most of its slots on zero weights.  The trip counts are frame-invariant: for the
          // first 128 outputs they are reduced once, before the frame loop.
          int nmaxa = h_nmaxa, nmaxb = h_nmaxb;
          if (m0 != 0) {
            nmaxa = na; nmaxb = nb;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
              nmaxa = max(nmaxa, __shfl_xor(nmaxa, d, 64));
              nmaxb = max(nmaxb, __shfl_xor(nmaxb, d, 64));
            }
          }
          // (na, nb, nmaxa, nmaxb are multiples of four; a lane whose filter has ended re-reads its first trip and
          //  keeps its sum; the zero padding multiplies power values of neighbouring bins, which are finite.)
          // Measured alternatives that were slower: one merged loop advancing both chains with the next trip's reads
          // issued ahead (94.7 vs 84.9 us: the short chain then re-reads for the long chain's trips and the LDS is the
          // co-bottleneck), 10 waves per workgroup at 5 waves per SIMD (98.9 us).
          for (int j = 0; j < nmaxa; j += 4) {
            const bool on = j < na;
            const int jj = on ? j : 0;
            const f32x4 w = *reinterpret_cast<const f32x4*>(wa + jj);
            const float* pp = P + sa + jj;
            const float p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
            const float t = fmaf(w[3], p3, fmaf(w[2], p2, fmaf(w[1], p1, fmaf(w[0], p0, va))));
            va = on ? t : va;
          }
          for (int j = 0; j < nmaxb; j += 4) {
            const bool on = j < nb;
            const int jj = on ? j : 0;
            const f32x4 w = *reinterpret_cast<const f32x4*>(wb + jj);
            const float* pp = P + sb + jj;
            const float p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
            const float t = fmaf(w[3], p3, fmaf(w[2], p2, fmaf(w[1], p1, fmaf(w[0], p0, vb))));
            vb = on ? t : vb;
          }
        } else {
          va = oka ? P[ma] : 0.f;
          vb = okb ? P[mb] : 0.f;
        }
        float ta = 0.f, tb = 0.f;
        if (a.loss_mode != 0) {
          if (oka) ta = a.target[row + ma];
          if (okb) tb = a.target[row + mb];
        }
        if (a.out != nullptr) {
          if (oka) a.out[row + ma] = va;
          if (okb) a.out[row + mb] = vb;
        }
        if (a.loss_mode == 1) {
          if (oka) l0 += fabsf(va - ta);
          if (okb) l0 += fabsf(vb - tb);
        } else if (LOSS2) {
          if (oka) { const float d = ta - va; l0 = fmaf(d, d, l0); l1 = fmaf(ta, ta, l1); l2 += fabsf(__log2f(va) - __log2f(ta)); }
          if (okb) { const float d = tb - vb; l0 = fmaf(d, d, l0); l1 = fmaf(tb, tb, l1); l2 += fabsf(__log2f(vb) - __log2f(tb)); }
        }
      }
    }
    wave_lds_sync();
    if (more) {
#pragma unroll
      for (int e = 0; e < 2 * R; ++e) xc[e] = xn[e];
    }
  }

  if (a.partials != nullptr) {
    __shared__ float s_red[SP_WAVES][4];
    l0 = wave_sum_f(l0); l1 = wave_sum_f(l1); l2 = wave_sum_f(l2) * 0.6931471805599453f;   // log-magnitude terms were taken in log2
    if (lane == 0) { s_red[wave][0] = l0; s_red[wave][1] = l1; s_red[wave][2] = l2; }
    __syncthreads();
    if (tid < 3) {
      double sacc = 0.0;
      for (int w = 0; w < SP_WAVES; ++w) sacc += (double)s_red[w][tid];
      a.partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + tid] = sacc;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// n_fft 1024, round 3 (design notes: csrc/stft2_kernels.hip, which also builds the segment-major mel tables):
// the same three radix-8 passes, then the Hermitian unpack on half the spectrum, the mel projection as conflict-free
// row reads of a segment-major power buffer, frames dealt to waves from one flat [B*F] list.
#define IAS_SEG_MAX_ROWS 17
#define IAS_SEG_HDR 16
#define IAS_SEG_STRIDE 72
// Exchange scratch of the radix-8 kernels: 64 rows of IAS_S2_ROW complex values per wave, element (row, col) at IAS_S2_AT.
// The LDS rules of MI355X_MICROARCH.md (ds_write_b64: 4 x 16 contiguous lanes over 32 dword banks; ds_read_b64: 2 x 32
// lanes over 64; ds_read_b128: 4 x 16 lanes) reproduce the counter exactly (scripts/diag/lds_exchange_model.py):
//   rows of 9 (the usual odd padding): the pass-1 scatter [(q 8 + c)][a] has 16 lanes at 9 c + a (c < 8, a in {2j, 2j+1}),
//     and 9 * 7 + 1 = 64 wraps onto 0: 2-way in every group of those eight stores (32 cycles per frame); the upper-half
//     stores of the unpack (index i + i / 8) wrap the same way (16) and its mirrored reads cost 8: 56 measured
//     (profiles/r03f_pmc_stft.txt, the kernel without the mel part);
//   rows of 10, two pad elements per eight in the upper half: the pass-1 scatter and the upper-half stores are clean, the
//     row reads become 16-byte reads (conflict-free in their lane groups, half the instructions), but the pass-2 scatter
//     [(k1 8 + d)][dd] now has its two k1 of a group 80 elements = 0 (mod 16) apart: 32 + 8 = 40 measured.
// The three patterns cannot all be conflict-free in this family (rows of 8 ... 18 elements with an offset per 8 rows,
// searched exhaustively: the 2-way conflict moves between the two scatters and the row reads), and a conflict in a STORE is
// the cheap place for it (6 issue cycles cover 4 of its 8 array cycles; a read pays all of them).  Rows of 10: 15 % fewer
// LDS cycles per frame than rows of 9, the 1024-point forward 53 -> 50.4 us.
// 640 complex values per wave also hold the mel projection's segment-major power rows at a stride of 72 floats (its scatter's
// ds_write_b32: 43 conflict cycles per frame at a stride of 65, 29 at 67, 22 at 74, 18 at 72 -- of which 2 cost time: a 2-way
// conflict of a 4-byte store hides behind its issue cycles; same model, same agreement with the counter).
#ifndef IAS_S2_ROW
#define IAS_S2_ROW 10
#endif
#define IAS_S2_AT(row, col) ((row) * IAS_S2_ROW + (col))
#define IAS_S2_UP(i) ((i) + (IAS_S2_ROW - 8) * ((i) >> 3))
static_assert(IAS_SEG_MAX_ROWS * IAS_SEG_STRIDE + 1 <= 2 * 64 * IAS_S2_ROW, "segment-major rows fit the wave's scratch");
struct Spec2Args {
  const float* audio;      // [B,T]
  const float* tables;     // ias_stft_build_tables block (n_fft 1024: with the unpack twiddles of this kernel at the end)
  const float* segtab;     // ias_stft_build_segtab block, or null: linear bins
  float* out;              // [B,F,n_out] or null
  const float* target;     // [B,F,n_out] or null
  double* partials;        // [gridDim.x][3] or null
  const float* rowpeak;    // [B] or null
  int T, F, hop, n_out, nframes;
  unsigned magicF;         // floor(2^32 / F)
  int value_mode, loss_mode;
  float eps;
#ifdef IAS_S2_STAMPS
  unsigned long long* stamps;   // diagnostics build only: [workgroup][wave][256] s_memtime values
#endif
};
// In-kernel stamps (diagnostic build -DIAS_S2_STAMPS only; the product build compiles none of this): where a wave's
// cycles go, phase by phase.  scripts/diag/stft2_stamps.py builds and reads them.
#ifdef IAS_S2_STAMPS
static unsigned long long* g_s2_stamps = nullptr;
extern "C" int ias_stft2_set_stamps(unsigned long long* p) { g_s2_stamps = p; return 0; }
#define S2_STAMP(id)                                                                                     \
  do {                                                                                                   \
    if (a.stamps != nullptr && stamp_n < 255) {                                                          \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                        \
      if (lane == 0) a.stamps[((size_t)blockIdx.x * SP_WAVES + wave) * 256 + 1 + stamp_n] = (t_ << 8) | (id); \
      ++stamp_n;                                                                                         \
    }                                                                                                    \
  } while (0)
#else
#define S2_STAMP(id) do {} while (0)
#endif
// lane l <- x of lane l-1 (lane 0 <- fill) / lane l+1 (lane 63 <- fill): one DPP move across the whole wave
__device__ __forceinline__ float wave_shr1(float x, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1(float x, float fill) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x130, 0xf, 0xf, false));
}

// NSUB = 2: n_fft 2048 on the same 8-points-per-lane core.  The 1024 packed complex points split into their even and
// odd halves (decimation in time), each a 512-point transform through the three radix-8 passes above, one after the
// other through the same scratch; Z[k] = E[k] + W_1024^k O[k] and Z[k + 512] = E[k] - W_1024^k O[k] are combined in
// the lane that holds both (same k), which is also exactly the lower / upper split the half-spectrum unpack wants.
// The round-2 kernel transforms such a frame with a radix-16 first pass at 16 points per lane: 164 VGPRs, 9.2 KB of
// scratch per wave, 2-3 waves per SIMD and 7x the time of a 1024-point frame; this form costs 2.3x.
template <int NSUB> __device__ __forceinline__ void stft2_load_frame(const float* __restrict__ arow, int T, int hop, int f,
                                                                      int lane, float (&x)[16 * NSUB]) {
  if (NSUB == 1) {
    float (&x1)[16] = reinterpret_cast<float (&)[16]>(x);
    load_frame<8, 512>(arow, T, hop, f, lane, x1);
  } else {
    // lane, n1: samples 256 n1 + 4 lane .. + 3 = (even point, odd point) of the two half-transforms; x[16 sub + 2 n1 + c]
    const int g0 = f * hop - 1024;
    if (g0 >= 0 && g0 + 2048 <= T && ((reinterpret_cast<uintptr_t>(arow + g0) & 15) == 0)) {
      // (no opaque pointer here, unlike load_frame: measured slower for the 2048-point kernels, 145 vs 123 us forward)
      const f32x4* p = reinterpret_cast<const f32x4*>(arow + g0) + lane;
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) {
        const f32x4 q = p[64 * n1];
        x[2 * n1] = q[0]; x[2 * n1 + 1] = q[1]; x[16 + 2 * n1] = q[2]; x[16 + 2 * n1 + 1] = q[3];
      }
    } else {
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          x[16 * (c >> 1) + 2 * n1 + (c & 1)] = arow[reflect_index(g0 + 256 * n1 + 4 * lane + c, T)];
        }
    }
  }
}

#ifndef IAS_STFT2_MINW
// waves per SIMD the 8-wave n_fft 1024 kernel is compiled for (register budget 512 / MINW).  Alone it does not matter
// (4: 112 VGPRs, 52.8 us; 5: 84 VGPRs, 53.1 us); in the headline step it decides how many of its waves fit on a SIMD next
// to the persistent render's (147 VGPRs each): step 0.192 / 0.173 / 0.173 / 0.179 ms at 4 / 5 / 6 / 8 (8 spills).
#define IAS_STFT2_MINW 5
#endif
// stft2_kernel's own scratch indexing.  With the first exchange in registers (IAS_S2_REGX) only the pass-2 scatter, ONE set
// of row reads and the upper-half stores / mirrored reads of the unpack touch the scratch, and the layout search of
// scripts/diag/lds_exchange_model.py (same lane groups and bank moduli) no longer has to keep the pass-1 scatter clean:
// rows of 9 complex values with two pad elements per eight in the upper half leave 8 conflict cycles per frame (the
// mirrored reads) instead of 40 -- the pass-2 scatter, 2-way at rows of 10, is clean, the row reads are eight ds_read_b64
// instead of four ds_read_b128 (the same LDS cycles).  The other radix-8 kernels keep IAS_S2_ROW = 10 (their pass-1
// scatter goes through LDS).  The wave's scratch stays 64 x IAS_S2_ROW values (the mel power rows need them).
#if IAS_S2_REGX
#define S2X_ROW 9
#define S2X_PAD 2
#else
#define S2X_ROW IAS_S2_ROW
#define S2X_PAD (IAS_S2_ROW - 8)
#endif
#define S2X_AT(row, col) ((row) * S2X_ROW + (col))
#define S2X_UP(i) ((i) + S2X_PAD * ((i) >> 3))
static_assert(64 * S2X_ROW <= 64 * IAS_S2_ROW && 256 + S2X_PAD * 32 <= 64 * IAS_S2_ROW, "stft2 scratch indexing fits the wave's scratch");
template <int SP_WAVES, bool MEL, int LOSS, int NSUB>
__global__ __launch_bounds__(64 * SP_WAVES, NSUB == 2 ? 3 * SP_WAVES / 8 : (SP_WAVES >= 8 ? (SP_WAVES == 8 ? IAS_STFT2_MINW : SP_WAVES / 2) : (3 * SP_WAVES + 3) / 4))
void stft2_kernel(const Spec2Args a) {
  static_assert(NSUB == 1 || (NSUB == 2 && !MEL), "mel filterbanks: n_fft 1024 only");
  constexpr int SP_THREADS = 64 * SP_WAVES, N2 = 512 * NSUB, R = 8, SCR = 64 * IAS_S2_ROW, NPK = 4 * NSUB, HALF = N2 / 2;
  // cpx per lane: window pairs, pass-1 twiddles, pass-2 twiddles, (NSUB = 2: combining twiddles,) unpack twiddles
  constexpr int NTAB = 8 * NSUB + 8 + 8 + (NSUB == 2 ? 8 : 0) + NPK;
  // The frame-invariant tables are LDS objects of their own, not slices of the dynamic array that holds the exchange
  // scratch: to the compiler a table read and a scratch store in ONE array may alias, so every twiddle / offset / weight
  // read placed behind a scratch store stayed behind it -- one serial LDS round trip per element of a pass.
  extern __shared__ __attribute__((aligned(16))) float smem[];
  cpx* s_scr = reinterpret_cast<cpx*>(smem);                        // SP_WAVES x SCR: FFT exchange scratch / power buffer
  __shared__ cpx s_tab[NTAB * 64];                                  // [NTAB][64]
  __shared__ cpx s_segw[MEL ? IAS_SEG_MAX_ROWS * 64 : 1];           // MEL: [rows][64] (up, down)
  __shared__ int s_sega[MEL ? 9 * 64 : 1];                          // MEL: [9][64] store offsets
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // tables.  n_fft 1024: the old block's entries [0,16) window, [16,32) pass-1, [32,48) pass-2 twiddles, and this
  // kernel's unpack twiddles W_1024^k, k = (lane >> 3) + 8 (lane & 7) + 64 e, in the 8 entries behind the old block's
  // unpack entries.  n_fft 2048: this kernel's own section behind the old block (ias_stft_build_tables), in s_tab order.
  constexpr int OLD_UNP = 10, V2_BASE_2048 = 32 + 32 + 32 + 18;
  for (int i = tid; i < NTAB * 64; i += SP_THREADS) {
    const int pp = i >> 6, l = i & 63;
    const int src = NSUB == 2 ? V2_BASE_2048 + 2 * pp : (pp < 24 ? 2 * pp : 48 + OLD_UNP + 2 * (pp - 24));
    s_tab[i] = cmk(a.tables[64 * src + l], a.tables[64 * (src + 1) + l]);
  }
  // the scratch doubles as the segment-major power buffer, some of whose words (the padding of the exchange layout,
  // positions past the end of a short segment) no pass ever writes: they meet zero weights and must be finite
  for (int i = tid; i < SP_WAVES * SCR; i += SP_THREADS) s_scr[i] = cmk(0.f, 0.f);
  int seg_rows = 0, seg_r[3] = {0, 0, 0}, seg_s0 = 0;
  if (MEL) {
    const int* hdr = reinterpret_cast<const int*>(a.segtab);
    seg_rows = hdr[1]; seg_r[0] = hdr[2]; seg_r[1] = hdr[3]; seg_r[2] = hdr[4]; seg_s0 = hdr[5];
    for (int i = tid; i < 9 * 64; i += SP_THREADS) s_sega[i] = hdr[IAS_SEG_HDR + i];
    const cpx* wsrc = reinterpret_cast<const cpx*>(a.segtab + IAS_SEG_HDR + 9 * 64);
    for (int i = tid; i < seg_rows * 64; i += SP_THREADS) s_segw[i] = wsrc[i];
  }
  const cpx* t_win = s_tab + lane;              // [8 sub + n1] -> the window at the lane's two samples of point n1
  const cpx* t_tw1 = t_win + 64 * 8 * NSUB;     // [k1]  -> W_512^(lane k1)
  const cpx* t_tw2 = t_tw1 + 64 * 8;            // [d]   -> W_64^(c d)
  const cpx* t_cmb = t_tw2 + 64 * 8;            // NSUB = 2: [e] -> W_1024^k, k = k1 + 8 d + 64 e
  const cpx* t_twu = t_cmb + (NSUB == 2 ? 64 * 8 : 0);   // [e] -> W_nfft^k, the lane's lower-half bins
  __syncthreads();                           // the only workgroup barrier before the final reduction

  cpx* sA = s_scr + wave * SCR;
  float* P2 = reinterpret_cast<float*>(sA);

  float l0 = 0.f, l1 = 0.f, l2 = 0.f;
  const int gw = blockIdx.x * SP_WAVES + wave, nw = gridDim.x * SP_WAVES;
  const int k1 = lane >> 3, dd = lane & 7;   // after the last pass the lane holds Z[k1 + 8 dd + 64 e]

  auto row_of = [&](int fi, int& b, int& f) {
    unsigned q0 = __umulhi((unsigned)fi, a.magicF);
    int r = fi - (int)q0 * a.F;
    if (r >= a.F) { r -= a.F; ++q0; }
    b = (int)q0; f = r;
  };
  float xc[16 * NSUB], xn[16 * NSUB];
#ifdef IAS_S2_STAMPS
  int stamp_n = 0;
#endif
  int fi = gw, bcur = 0, fcur = 0;
  // the row peak travels with the frame's samples (requested one frame ahead): read where it is used, its wait -- the
  // memory counter is in order -- would drain the prefetch of the next frame in every iteration
  float pk_cur = 0.0f, pk_next = 0.0f;
  if (fi < a.nframes) {
    row_of(fi, bcur, fcur);
    stft2_load_frame<NSUB>(a.audio + (size_t)bcur * a.T, a.T, a.hop, fcur, lane, xc);
    if (a.rowpeak != nullptr) pk_cur = a.rowpeak[bcur];
    // the first frame's loads are consumed HERE, outside the loop: left pending into the loop header they make the
    // compiler wait for "everything older" at the first use inside the body, which is behind the next frame's prefetch in
    // program order -- a full drain of the memory pipeline (vmcnt(0)) in every iteration
#pragma unroll
    for (int e = 0; e < 16 * NSUB; ++e) asm volatile("" : "+v"(xc[e]));
    asm volatile("" : "+v"(pk_cur));
  }
  for (; fi < a.nframes; fi += nw) {
    const bool more = fi + nw < a.nframes;   // wave-uniform
    S2_STAMP(1);
    int bnext = 0, fnext = 0;
    if (more) {
      row_of(fi + nw, bnext, fnext);
      stft2_load_frame<NSUB>(a.audio + (size_t)bnext * a.T, a.T, a.hop, fnext, lane, xn);
      if (a.rowpeak != nullptr) pk_next = a.rowpeak[bnext];
    }
    float pscale = 0.25f;
    if (pk_cur > 1.0f) { const float r = __builtin_amdgcn_rcpf(pk_cur); pscale = 0.25f * (r * r); }
    const size_t row = (size_t)fi * a.n_out;
    // MEL: the frame's target row is requested now and consumed after the transform
    float tgt_m[3] = {0.f, 0.f, 0.f};
    if (MEL && LOSS != 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) if (64 * c + lane < a.n_out) tgt_m[c] = a.target[row + 64 * c + lane];
    }

    const int kl = k1 + 8 * dd;
    // linear bins: the same for the lane's 2 NPK (+ 1) bins
    float tg_k[NPK], tg_n[NPK], tg_mid = 0.f;
    if (!MEL && LOSS != 0) {
#pragma unroll
      for (int e = 0; e < NPK; ++e) { tg_k[e] = a.target[row + kl + 64 * e]; tg_n[e] = a.target[row + N2 - kl - 64 * e]; }
      if (lane == 0) tg_mid = a.target[row + HALF];
    }
    cpx zlo[NPK], zhi[NPK];                  // Z[kl + 64 e] and Z[HALF + kl + 64 e]
    cpx ue[8];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      // pass 1: radix 8 over n1 (points 64 n1 + lane), twiddle W_512^(lane k1), scatter to [k1][c][a]
      cpx v[R];
#pragma unroll
      for (int n1 = 0; n1 < R; ++n1) v[n1] = cmk(xc[16 * sub + 2 * n1], xc[16 * sub + 2 * n1 + 1]) * t_win[64 * (8 * sub + n1)];
      dft8(v);
      cpx u[8];
#if IAS_S2_REGX
#pragma unroll
      for (int q = 0; q < R; ++q) u[q] = cmul(v[q], t_tw1[64 * q]);
      S2_STAMP(2);
      xchg_reg_lane345(u);
      S2_STAMP(3);
      S2_STAMP(4);
#else
      {
        const int c = lane & 7, aa = lane >> 3;
#pragma unroll
        for (int q = 0; q < R; ++q) sA[S2X_AT(q * 8 + c, aa)] = cmul(v[q], t_tw1[64 * q]);
      }
      S2_STAMP(2);
      wave_lds_sync();
      S2_STAMP(3);
#pragma unroll
      for (int q = 0; q < 8; ++q) u[q] = sA[S2X_AT(lane, q)];
      wave_lds_sync();
      S2_STAMP(4);
#endif
      // pass 2: radix 8 over a for each (k1, c); twiddle W_64^(c d); scatter (in place) to [k1][d][c]
      dft8(u);
#pragma unroll
      for (int d = 0; d < 8; ++d) sA[S2X_AT(k1 * 8 + d, dd)] = cmul(u[d], t_tw2[64 * d]);
      S2_STAMP(5);
      wave_lds_sync();
      S2_STAMP(6);
#pragma unroll
      for (int q = 0; q < 8; ++q) u[q] = sA[S2X_AT(lane, q)];
      wave_lds_sync();
      S2_STAMP(7);
      // pass 3: radix 8 over c for each (k1, d): u[e] = (half-)transform at k1 + 8 d + 64 e
      dft8(u);
      if (NSUB == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { zlo[e] = u[e]; zhi[e] = u[4 + e]; }
      } else if (sub == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ue[e] = u[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const cpx t = cmul(u[e], t_cmb[64 * e]);
          zlo[e % NPK] = cadd(ue[e], t); zhi[e % NPK] = csub(ue[e], t);
        }
      }
    }
    // the upper half (k >= HALF) goes to LDS at k - HALF (padded per 8 complex values, IAS_S2_UP)
#pragma unroll
    for (int e = 0; e < NPK; ++e) { const int i = kl + 64 * e; sA[S2X_UP(i)] = zhi[e]; }
    S2_STAMP(8);
    wave_lds_sync();
    S2_STAMP(9);
    // unpack: own bins k = kl + 64 e with Z[N2 - k] from the upper half
    float pk[NPK], pn[NPK];
#pragma unroll
    for (int e = 0; e < NPK; ++e) {
      const int k = kl + 64 * e;
      const int i = (HALF - k) & (HALF - 1);                 // k = 0: Z[N2] = Z[0] (own), the read is a dummy
      cpx zn = sA[S2X_UP(i)];
      const cpx zk = zlo[e];
      if (e == 0 && k == 0) zn = zk;
      const cpx w = t_twu[64 * e];
      const float ea = zk.x + zn.x, eb = zk.y - zn.y, od = zk.x - zn.x, os = zk.y + zn.y;
      const float tx = fmaf(w.y, od, w.x * os), ty = fmaf(-w.x, od, w.y * os);
      const float xr = ea + tx, xi = eb + ty, yr = ea - tx, yi = eb - ty;
      pk[e] = fmaf(xi, xi, xr * xr) * pscale;
      pn[e] = fmaf(yi, yi, yr * yr) * pscale;
    }
    float pmid = fmaf(zhi[0].y, zhi[0].y, zhi[0].x * zhi[0].x) * (4.0f * pscale);    // lane 0: |Z[HALF]|^2
    if (a.value_mode == 1) {
#pragma unroll
      for (int e = 0; e < NPK; ++e) { pk[e] = __builtin_amdgcn_sqrtf(pk[e]); pn[e] = __builtin_amdgcn_sqrtf(pn[e]); }   // v_sqrt_f32, 1 ulp
      pmid = __builtin_amdgcn_sqrtf(pmid);
    } else if (a.value_mode == 3) {
#pragma unroll
      for (int e = 0; e < NPK; ++e) { pk[e] = __builtin_amdgcn_sqrtf(fmaxf(pk[e], a.eps)); pn[e] = __builtin_amdgcn_sqrtf(fmaxf(pn[e], a.eps)); }
      pmid = __builtin_amdgcn_sqrtf(fmaxf(pmid, a.eps));
    }
    S2_STAMP(10);
    wave_lds_sync();   // every Z read is done: the power values overwrite the scratch
    S2_STAMP(11);
    auto emit_t = [&](int m, float val, float t) {
      if (a.out != nullptr) a.out[row + m] = val;
      if (LOSS == 1) l0 += fabsf(val - t);
      else if (LOSS == 2) { const float d = t - val; l0 = fmaf(d, d, l0); l1 = fmaf(t, t, l1); l2 += fabsf(__log2f(val) - __log2f(t)); }
    };

    if (MEL) {
      // segment-major store: position t of segment j at row (base + t), column j % 64
#pragma unroll
      for (int e = 0; e < 4; ++e) { P2[s_sega[64 * e + lane]] = pk[e]; P2[s_sega[64 * (4 + e) + lane]] = pn[e]; }
      if (lane == 0) P2[s_sega[64 * 8]] = pmid;
      S2_STAMP(12);
      wave_lds_sync();
      S2_STAMP(13);
      // per segment group: the rows four at a time, their eight reads in flight together (one read, one wait, one fma per
      // row -- the rolled loop -- is a serial chain of LDS round trips)
      float U[3], D[3];
      int rbase = 0;
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        float us = 0.f, ds = 0.f;
        const int n = seg_r[g];
        int t = 0;
        for (; t + 4 <= n; t += 4) {
          float pv[4]; cpx w[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { pv[u] = P2[(rbase + t + u) * IAS_SEG_STRIDE + lane]; w[u] = s_segw[(rbase + t + u) * 64 + lane]; }
#pragma unroll
          for (int u = 0; u < 4; ++u) { us = fmaf(w[u].x, pv[u], us); ds = fmaf(w[u].y, pv[u], ds); }
        }
        if (t < n) {                                       // up to three rows left: read all three slots (in bounds), use n - t
          float pv[3]; cpx w[3];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int row = min(rbase + t + u, IAS_SEG_MAX_ROWS - 1);
            pv[u] = P2[row * IAS_SEG_STRIDE + lane]; w[u] = s_segw[row * 64 + lane];
          }
#pragma unroll
          for (int u = 0; u < 3; ++u)
            if (t + u < n) { us = fmaf(w[u].x, pv[u], us); ds = fmaf(w[u].y, pv[u], ds); }
        }
        U[g] = us; D[g] = ds;
        rbase += n;
      }
      S2_STAMP(14);
      // mel m = U[segment m] + D[segment m + 1]; slot i = 64 g + lane holds segment s0 + i
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int m = 64 * c + lane;
        if (64 * c < a.n_out) {
          float val;
          if (seg_s0 == 1) {
            const float fill = c > 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(U[c > 0 ? c - 1 : 0]), 63)) : 0.f;
            val = D[c] + wave_shr1(U[c], fill);
          } else {
            const float fill = c < 2 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(D[c < 2 ? c + 1 : 2]), 0)) : 0.f;
            val = U[c] + wave_shl1(D[c], fill);
          }
          if (m < a.n_out) emit_t(m, val, tgt_m[c]);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < NPK; ++e) { const int k = kl + 64 * e; emit_t(k, pk[e], tg_k[e]); emit_t(N2 - k, pn[e], tg_n[e]); }
      if (lane == 0) emit_t(HALF, pmid, tg_mid);
    }
    S2_STAMP(15);
    wave_lds_sync();
    if (more) {
#pragma unroll
      for (int e = 0; e < 16 * NSUB; ++e) xc[e] = xn[e];
      bcur = bnext; fcur = fnext; pk_cur = pk_next;
    }
  }

#ifdef IAS_S2_STAMPS
  if (a.stamps != nullptr && lane == 0) a.stamps[((size_t)blockIdx.x * SP_WAVES + wave) * 256] = stamp_n;
#endif
  if (a.partials != nullptr) {
    __shared__ float s_red[SP_WAVES][4];
    l0 = wave_sum_f(l0); l1 = wave_sum_f(l1); l2 = wave_sum_f(l2) * 0.6931471805599453f;   // log-magnitude terms were taken in log2
    if (lane == 0) { s_red[wave][0] = l0; s_red[wave][1] = l1; s_red[wave][2] = l2; }
    __syncthreads();
    if (tid < 3) {
      double sacc = 0.0;
      for (int w = 0; w < SP_WAVES; ++w) sacc += (double)s_red[w][tid];
      a.partials[(size_t)blockIdx.x * 3 + tid] = sacc;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// n_fft 512 (linear bins: STFT-L1 and the third MR-STFT resolution) on the 8-points-per-lane core: TWO frames per wave.
// A 512-point frame is 256 packed complex points = 4 per lane: stft_kernel<9> runs its two radix-8 passes with 32 of
// the 64 lanes.  Here a lane holds four points of frame A and four of frame B; pass 1 is two radix-4 transforms (n1 of
// each frame), and what is left -- for each of the 2 x 4 (frame, k1) slots a 64-point transform over the lane index --
// is exactly passes 2 and 3 of the 1024-point kernel with the slot in place of its k1.  After the last pass lane
// (slot = lane >> 3, d = lane & 7) holds Z_frame[k1 + 4 d + 32 e] (frame = slot >> 2, k1 = slot & 3): lanes 0..31 own
// frame A, lanes 32..63 frame B, each with the 4 + 4 bins of the half-spectrum unpack (as stft2_kernel, N/2 = 256).
template <int SP_WAVES, int LOSS>
__global__ __launch_bounds__(64 * SP_WAVES, SP_WAVES >= 8 ? SP_WAVES / 2 : (3 * SP_WAVES + 3) / 4)
void stft2h_kernel(const Spec2Args a) {
  constexpr int SP_THREADS = 64 * SP_WAVES, SCR = 64 * IAS_S2_ROW, N2 = 256, HALF = 128;
  constexpr int NTAB = 4 + 4 + 8 + 4;        // cpx per lane: window pairs, pass-1 twiddles, pass-2 twiddles, unpack twiddles
  extern __shared__ __attribute__((aligned(16))) float smem[];
  cpx* s_scr = reinterpret_cast<cpx*>(smem);
  __shared__ cpx s_tab[NTAB * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the n_fft 512 block of ias_stft_build_tables: [0,8) window, [8,16) pass 1, [16,32) pass 2, [38,46) this kernel's unpack
  for (int i = tid; i < NTAB * 64; i += SP_THREADS) {
    const int pp = i >> 6, l = i & 63, src = pp < 16 ? 2 * pp : 38 + 2 * (pp - 16);
    s_tab[i] = cmk(a.tables[64 * src + l], a.tables[64 * (src + 1) + l]);
  }
  const cpx* t_win = s_tab + lane;           // [n1]  -> the window at the lane's two samples of point n1
  const cpx* t_tw1 = t_win + 64 * 4;         // [k1]  -> W_256^(lane k1)
  const cpx* t_tw2 = t_tw1 + 64 * 4;         // [d]   -> W_64^(c d), c = lane & 7
  const cpx* t_twu = t_tw2 + 64 * 8;         // [e]   -> W_512^k, k = kl + 32 e
  __syncthreads();

  cpx* sA = s_scr + wave * SCR;
  float l0 = 0.f, l1 = 0.f, l2 = 0.f;
  const int gw = blockIdx.x * SP_WAVES + wave, nw = gridDim.x * SP_WAVES;
  const int slot = lane >> 3, dd = lane & 7, fr = lane >> 5, kl = (slot & 3) + 4 * dd;
  const int npairs = (a.nframes + 1) >> 1;
  auto row_of = [&](int fi, int& b, int& f) {
    unsigned q0 = __umulhi((unsigned)fi, a.magicF);
    int r = fi - (int)q0 * a.F;
    if (r >= a.F) { r -= a.F; ++q0; }
    b = (int)q0; f = r;
  };
  // frames 2 pi and 2 pi + 1 (the last pair of an odd list repeats its first frame; the repeat is not emitted)
  auto load_pair = [&](int pi, float (&x)[16], int& bA, int& bB) {
    const int fiA = 2 * pi, fiB = min(2 * pi + 1, a.nframes - 1);
    int fA, fB;
    row_of(fiA, bA, fA); row_of(fiB, bB, fB);
    float (&xa)[8] = reinterpret_cast<float (&)[8]>(x[0]);
    float (&xb)[8] = reinterpret_cast<float (&)[8]>(x[8]);
    load_frame<4, 256>(a.audio + (size_t)bA * a.T, a.T, a.hop, fA, lane, xa);
    load_frame<4, 256>(a.audio + (size_t)bB * a.T, a.T, a.hop, fB, lane, xb);
  };
  float xc[16], xn[16];
  int pi = gw, bA = 0, bB = 0;
  if (pi < npairs) {
    load_pair(pi, xc, bA, bB);
#pragma unroll
    for (int e = 0; e < 16; ++e) asm volatile("" : "+v"(xc[e]));      // consumed outside the loop (see stft2_kernel)
  }
  for (; pi < npairs; pi += nw) {
    const bool more = pi + nw < npairs;      // wave-uniform
    int bAn = 0, bBn = 0;
    if (more) load_pair(pi + nw, xn, bAn, bBn);
    const bool own = fr == 0 || 2 * pi + 1 < a.nframes;                // this lane's frame exists
    const int fi = own ? 2 * pi + fr : 2 * pi;
    const size_t row = (size_t)fi * a.n_out;
    float pscale = 0.25f;
    if (a.rowpeak != nullptr) { const float pkv = a.rowpeak[fr ? bB : bA]; if (pkv > 1.0f) { const float r = __builtin_amdgcn_rcpf(pkv); pscale = 0.25f * (r * r); } }
    float tg_k[4], tg_n[4], tg_mid = 0.f;
    if (LOSS != 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { tg_k[e] = a.target[row + kl + 32 * e]; tg_n[e] = a.target[row + N2 - kl - 32 * e]; }
      if (kl == 0) tg_mid = a.target[row + HALF];
    }
    // pass 1: radix 4 over n1 for both frames; slot q = 4 frame + k1; twiddle W_256^(lane k1); scatter to [q][c][a]
    cpx v[8];
#pragma unroll
    for (int n1 = 0; n1 < 4; ++n1) {
      v[n1] = cmk(xc[2 * n1], xc[2 * n1 + 1]) * t_win[64 * n1];
      v[4 + n1] = cmk(xc[8 + 2 * n1], xc[8 + 2 * n1 + 1]) * t_win[64 * n1];
    }
    {
      cpx (&va)[4] = reinterpret_cast<cpx (&)[4]>(v[0]);
      cpx (&vb)[4] = reinterpret_cast<cpx (&)[4]>(v[4]);
      dftR<4>(va); dftR<4>(vb);
      const int c = lane & 7, aa = lane >> 3;
#pragma unroll
      for (int q = 0; q < 8; ++q) sA[(q * 8 + c) * IAS_S2_ROW + aa] = cmul(v[q], t_tw1[64 * (q & 3)]);
    }
    wave_lds_sync();
    cpx u[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) u[q] = sA[lane * IAS_S2_ROW + q];
    wave_lds_sync();
    // pass 2: radix 8 over a for each (slot, c); twiddle W_64^(c d); scatter (in place) to [slot][d][c]
    dft8(u);
#pragma unroll
    for (int d = 0; d < 8; ++d) sA[(slot * 8 + d) * IAS_S2_ROW + dd] = cmul(u[d], t_tw2[64 * d]);
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) u[q] = sA[lane * IAS_S2_ROW + q];
    wave_lds_sync();
    // pass 3: radix 8 over c: u[e] = Z_frame[kl + 32 e]
    dft8(u);
    // the upper halves (k >= 128) go to LDS, frame-major
#pragma unroll
    for (int e = 0; e < 4; ++e) sA[fr * HALF + kl + 32 * e] = u[4 + e];
    wave_lds_sync();
    float pk[4], pn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = kl + 32 * e;
      cpx zn = sA[fr * HALF + ((HALF - k) & (HALF - 1))];    // k = 0: Z[N2] = Z[0] (own), the read is a dummy
      const cpx zk = u[e];
      if (e == 0 && k == 0) zn = zk;
      const cpx w = t_twu[64 * e];
      const float ea = zk.x + zn.x, eb = zk.y - zn.y, od = zk.x - zn.x, os = zk.y + zn.y;
      const float tx = fmaf(w.y, od, w.x * os), ty = fmaf(-w.x, od, w.y * os);
      const float xr = ea + tx, xi = eb + ty, yr = ea - tx, yi = eb - ty;
      pk[e] = fmaf(xi, xi, xr * xr) * pscale;
      pn[e] = fmaf(yi, yi, yr * yr) * pscale;
    }
    float pmid = fmaf(u[4].y, u[4].y, u[4].x * u[4].x) * (4.0f * pscale);    // lanes with kl = 0: |Z[128]|^2
    if (a.value_mode == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { pk[e] = __builtin_amdgcn_sqrtf(pk[e]); pn[e] = __builtin_amdgcn_sqrtf(pn[e]); }
      pmid = __builtin_amdgcn_sqrtf(pmid);
    } else if (a.value_mode == 3) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { pk[e] = __builtin_amdgcn_sqrtf(fmaxf(pk[e], a.eps)); pn[e] = __builtin_amdgcn_sqrtf(fmaxf(pn[e], a.eps)); }
      pmid = __builtin_amdgcn_sqrtf(fmaxf(pmid, a.eps));
    }
    auto emit_t = [&](int m, float val, float t) {
      if (a.out != nullptr) a.out[row + m] = val;
      if (LOSS == 1) l0 += fabsf(val - t);
      else if (LOSS == 2) { const float d = t - val; l0 = fmaf(d, d, l0); l1 = fmaf(t, t, l1); l2 += fabsf(__log2f(val) - __log2f(t)); }
    };
    if (own) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const int k = kl + 32 * e; emit_t(k, pk[e], tg_k[e]); emit_t(N2 - k, pn[e], tg_n[e]); }
      if (kl == 0) emit_t(HALF, pmid, tg_mid);
    }
    wave_lds_sync();
    if (more) {
#pragma unroll
      for (int e = 0; e < 16; ++e) xc[e] = xn[e];
      bA = bAn; bB = bBn;
    }
  }
  if (a.partials != nullptr) {
    __shared__ float s_red[SP_WAVES][4];
    l0 = wave_sum_f(l0); l1 = wave_sum_f(l1); l2 = wave_sum_f(l2) * 0.6931471805599453f;   // log-magnitude terms were taken in log2
    if (lane == 0) { s_red[wave][0] = l0; s_red[wave][1] = l1; s_red[wave][2] = l2; }
    __syncthreads();
    if (tid < 3) {
      double sacc = 0.0;
      for (int w = 0; w < SP_WAVES; ++w) sacc += (double)s_red[w][tid];
      a.partials[(size_t)blockIdx.x * 3 + tid] = sacc;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the linear-bin losses, frame part, on the same wave-per-frame FFT core (round 2).
//
// d loss / d x_f for  loss_mode 1: scale * sum |V - t|  and  loss_mode 2: one MR-STFT resolution (cotangent
// gO = coef[0] (V - t) + coef[1] sign(V - t) / V), V = |X|^2 (power 2) or sqrt(|X|^2) (power 1; loss_mode 2 clamps at
// eps), linear bins (n_out = N/2 + 1).  The first version (csrc/spectral_grad_kernels.hip: one workgroup per frame
// pair, Stockham radix-4 with a workgroup barrier per stage) was 51 % of the configs[4] gradient step.  Here a wave
// owns a frame end to end:
//   forward   the stft_kernel passes (window, packed real FFT of N/2 complex points), unpack to X[k], X[N/2-k] in registers
//   adjoint   G[k] = 2 gP[k] X[k] per bin, in registers (a lane owns bins k and N/2 - k)
//   inverse   y[n] = Re sum_{k<=N/2} G[k] e^{+2 pi i k n / N} as ONE N/2-point complex inverse FFT:
//             with a[k] = G[k], a[0] = G[0] + G[N/2];  b[k] = G[k] e^{+2 pi i k / N}, b[0] = G[0] - G[N/2]:
//             y[2m] + i y[2m+1] = IDFT(Zin)[m],  Zin[k] = (a[k] + conj a[N/2-k]) / 2 + i (b[k] + conj b[N/2-k]) / 2;
//             the inverse transform is the same three passes on conj(Zin), conjugated again at the end
//   output    frame_grad[f][n] = window[n] y[n]  (the overlap-add stays in stft_grad_ola_kernel)
extern "C" int ias_stft_num_frames(int T, int n_fft, int hop);
struct SgwArgs {
  const float* audio; const float* tables; const float* target; const double* coef; float* frame_grad;
  const int* mel_start; const int* mel_count; const int* mel_woff; const float* mel_w;   // MEL: the forward's CSR filters
  int T, F, hop, groups, power2, loss_mode, n_out, mel_nnz;
  float scale, eps;
  int G, cper, L, nchunks;   // SPAN kernels: frames per chunk, chunks per row, floats per chunk span, B * cper
};

// MEL: the loss is taken on O = melW^T V (loss_mode 1 only).  V goes to LDS, a lane computes two outputs with the
// forward's padded filter loops, their cotangents are scattered back to the bins with LDS float atomics (a bin lies under
// at most two triangular filters: the sum of two terms does not depend on their order), then the bins continue as above.
//
// SPAN (round 3): overlap-add inside the kernel.  A wave owns a CHUNK of G consecutive frames of one row and walks it
// in order; the windowed frame gradients are added into a ring of n_fft floats in LDS (one wave, program order: the
// sum over frames is in frame order, deterministic), and after every frame the hop samples no later frame of the chunk
// touches leave the ring for the chunk's span  frame_grad[chunk][L], L = (G - 1) hop + n_fft.  The [B,F,n_fft] tensor
// (8.5 - 10 floats per audio sample, written here and read again by stft_grad_ola_kernel) shrinks to ~1.2 - 1.5
// floats per sample; stft_grad_combine_kernel adds the two chunks that meet at a sample, lower chunk first.
template <int LOG2N, bool MEL, bool SPAN>
__global__ __launch_bounds__(256) void stft_grad_wave_kernel(const SgwArgs a) {
  constexpr int SP_WAVES = 4, SP_THREADS = 256;
  constexpr int NFFT = 1 << LOG2N, N2 = NFFT / 2, R = N2 / 64, NPAIR = 8 * R, NP_IT = (NPAIR + 63) / 64;
  constexpr int SCR = NPAIR * IAS_S2_ROW, NUNP = (N2 / 2) / 64 + 1, NB = N2 + 1;
  constexpr int NTAB = 64 * (2 * R + 2 * R + 16 * NP_IT + 2 * NUNP);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  cpx* s_scr = reinterpret_cast<cpx*>(smem);
  float* s_ring = reinterpret_cast<float*>(s_scr + SP_WAVES * SCR);          // SPAN: [wave][NFFT]
  // the tables are an LDS object of their own: carved out of the array that holds the scratch, every table read behind a
  // scratch store was ordered after it (see stft2_kernel)
  __shared__ cpx s_tab[NTAB / 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (SPAN) for (int i = tid; i < SP_WAVES * NFFT; i += SP_THREADS) s_ring[i] = 0.0f;
  for (int i = tid; i < NTAB / 2; i += SP_THREADS) {
    const int pp = i >> 6, l = i & 63;
    s_tab[i] = cmk(a.tables[64 * (2 * pp) + l], a.tables[64 * (2 * pp + 1) + l]);
  }
  // MEL: padded weights + descriptors (as in stft_kernel), then per wave the output cotangents and the bin cotangents
  float* s_melw = reinterpret_cast<float*>(s_scr + SP_WAVES * SCR + (SPAN ? SP_WAVES * N2 : 0));
  const int melw_words = MEL ? mel_padded_words(a.mel_nnz, a.n_out) : 0;
  int* s_meli = reinterpret_cast<int*>(s_melw + melw_words);            // [3][n_out]: start, padded count, padded offset
  const int so_words = MEL ? ((a.n_out + 3) & ~3) : 0;
  float* s_go = reinterpret_cast<float*>(s_meli + (MEL ? ((3 * a.n_out + 3) & ~3) : 0)) + wave * (so_words + NB + 7);
  float* s_gv = s_go + so_words;                                        // [NB + 4] (+ slack for the zero padding)
  if (MEL) {
    for (int i = tid; i < melw_words; i += SP_THREADS) s_melw[i] = 0.0f;
    for (int i = tid; i < a.n_out; i += SP_THREADS) {
      s_meli[i] = a.mel_start[i];
      s_meli[a.n_out + i] = (a.mel_count[i] + 3) & ~3;
    }
    __syncthreads();
    for (int i = tid; i < a.n_out; i += SP_THREADS) {
      int off = 0;
      for (int m = 0; m < i; ++m) off += s_meli[a.n_out + m];
      s_meli[2 * a.n_out + i] = off;
    }
    __syncthreads();
    for (int i = tid; i < a.n_out; i += SP_THREADS) {
      const int n = a.mel_count[i], src = a.mel_woff[i], dst = s_meli[2 * a.n_out + i];
      for (int j = 0; j < n; ++j) s_melw[dst + j] = a.mel_w[src + j];
    }
  }
  const cpx* t_win = s_tab + lane;
  const cpx* t_tw1 = t_win + 64 * R;
  const cpx* t_tw2 = t_tw1 + 64 * R;
  const cpx* t_twu = t_tw2 + 64 * 8 * NP_IT;
  __syncthreads();
  cpx* sA = s_scr + wave * SCR;
  float c0 = 0.0f, c1 = 0.0f;
  if (a.loss_mode == 2) { c0 = (float)a.coef[0]; c1 = (float)a.coef[1]; }

  // the three passes of stft_kernel: v = the lane's R points (64 n1 + lane) -> DFT in natural order, sA[IAS_S2_UP(k)]
  auto fft = [&](cpx (&v)[R]) {
    dftR<R>(v);
    {
      const int c = lane & 7, aa = lane >> 3;
#pragma unroll
      for (int k1 = 0; k1 < R; ++k1) sA[(k1 * 8 + c) * IAS_S2_ROW + aa] = cmul(v[k1], t_tw1[64 * k1]);
    }
    wave_lds_sync();
    cpx u[NP_IT][8];
#pragma unroll
    for (int i = 0; i < NP_IT; ++i) {
      const int p = lane + 64 * i;
      if (p < NPAIR) {
#pragma unroll
        for (int q = 0; q < 8; ++q) u[i][q] = sA[p * IAS_S2_ROW + q];
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NP_IT; ++i) {
      const int p = lane + 64 * i;
      if (p < NPAIR) {
        const int k1 = p >> 3, c = p & 7;
        dft8(u[i]);
#pragma unroll
        for (int d = 0; d < 8; ++d) sA[(k1 * 8 + d) * IAS_S2_ROW + c] = cmul(u[i][d], t_tw2[64 * (8 * i + d)]);
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NP_IT; ++i) {
      const int p = lane + 64 * i;
      if (p < NPAIR) {
#pragma unroll
        for (int q = 0; q < 8; ++q) u[i][q] = sA[p * IAS_S2_ROW + q];
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NP_IT; ++i) {
      const int p = lane + 64 * i;
      if (p < NPAIR) {
        const int k1 = p >> 3, d = p & 7;
        dft8(u[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int k = k1 + R * d + 8 * R * e; sA[IAS_S2_UP(k)] = u[i][e]; }
      }
    }
    wave_lds_sync();
  };
  // value V of a bin from its power; d loss / d V from value and target (linear bins); d loss / d |X|^2 from d loss / d V
  auto bin_value = [&](float p) { return a.power2 ? p : __builtin_amdgcn_sqrtf(a.loss_mode == 2 ? fmaxf(p, a.eps) : p); };   // v_sqrt_f32
  auto value_grad = [&](float v, float t) {
    const float d = v - t;
    const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    return a.loss_mode == 2 ? c0 * d + c1 * sg * __builtin_amdgcn_rcpf(v) : sg * a.scale;   // v_rcp_f32 (1 ulp)
  };
  auto power_grad = [&](float gv, float v) {
    if (a.power2) return gv;
    const bool live = a.loss_mode == 2 ? v > __builtin_amdgcn_sqrtf(a.eps) : v > 0.0f;   // a clamped (or zero) bin passes nothing
    return live ? 0.5f * gv * __builtin_amdgcn_rcpf(v) : 0.0f;
  };

  float xc[2 * R];
  float* ring = s_ring + wave * NFFT;
  const int cstep = SPAN ? (int)gridDim.x * SP_WAVES : 1;
  for (int c = SPAN ? (int)blockIdx.x * SP_WAVES + wave : 0; c < (SPAN ? a.nchunks : 1); c += cstep) {
  int b, f_lo, f_hi;
  if (SPAN) { b = c / a.cper; f_lo = (c - b * a.cper) * a.G; f_hi = min(f_lo + a.G, a.F); }
  else { b = blockIdx.y; f_lo = blockIdx.x * a.groups + wave; f_hi = min((int)blockIdx.x * a.groups + a.groups, a.F); }
  const float* arow = a.audio + (size_t)b * a.T;
  float* span = a.frame_grad + (size_t)c * a.L;
  for (int f = f_lo; f < f_hi; f += SPAN ? 1 : SP_WAVES) {
    load_frame<R, N2>(arow, a.T, a.hop, f, lane, xc);
    const float* trow = a.target + ((size_t)b * a.F + f) * (MEL ? a.n_out : NB);
    // linear bins: the lane's target values are requested with the frame and consumed after the forward transform
    float tk_[NUNP], tq_[NUNP];
    if (!MEL) {
#pragma unroll
      for (int i = 0; i < NUNP; ++i) {
        const int k = lane + 64 * i;
        tk_[i] = tq_[i] = 0.0f;
        if (k <= N2 / 2) { tk_[i] = trow[k]; tq_[i] = trow[N2 - k]; }
      }
    }
    cpx v[R];
#pragma unroll
    for (int n1 = 0; n1 < R; ++n1) v[n1] = cmk(xc[2 * n1], xc[2 * n1 + 1]) * t_win[64 * n1];
    fft(v);
    // unpack per bin pair (k, N2 - k), k = lane + 64 i <= N2 / 2: X and V of both bins in registers
    cpx xk_[NUNP], xq_[NUNP];
    float vk_[NUNP], vq_[NUNP], gk_[NUNP], gq_[NUNP];
#pragma unroll
    for (int i = 0; i < NUNP; ++i) {
      const int k = lane + 64 * i;
      xk_[i] = xq_[i] = cmk(0.0f, 0.0f);
      vk_[i] = vq_[i] = gk_[i] = gq_[i] = 0.0f;
      if (k <= N2 / 2) {
        const int kn = (N2 - k) & (N2 - 1);
        const cpx zk = sA[IAS_S2_UP(k)], zn = sA[IAS_S2_UP(kn)];
        const cpx ze = cmk(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const cpx zo = cmk(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        const cpx t = cmul(t_twu[64 * i], zo);            // W_N^k Zo[k]
        xk_[i] = cadd(ze, t);                             // X[k]
        xq_[i] = cmk(ze.x - t.x, -(ze.y - t.y));          // X[N2 - k]
        vk_[i] = bin_value(xk_[i].x * xk_[i].x + xk_[i].y * xk_[i].y);
        vq_[i] = bin_value(xq_[i].x * xq_[i].x + xq_[i].y * xq_[i].y);
        if (!MEL) { gk_[i] = value_grad(vk_[i], tk_[i]); gq_[i] = value_grad(vq_[i], tq_[i]); }
      }
    }
    if (MEL) {
      wave_lds_sync();                                    // every Z read is done: V overwrites the scratch
      float* P = reinterpret_cast<float*>(sA);
#pragma unroll
      for (int i = 0; i < NUNP; ++i) {
        const int k = lane + 64 * i;
        if (k <= N2 / 2) { P[k] = vk_[i]; P[N2 - k] = vq_[i]; }
      }
      for (int i = lane; i < NB + 4; i += 64) s_gv[i] = 0.0f;
      wave_lds_sync();
      for (int m0 = 0; m0 < a.n_out; m0 += 64) {
        const int m = m0 + lane;
        if (m < a.n_out) {
          const int s0 = s_meli[m], n4 = s_meli[a.n_out + m];
          const float* w = s_melw + s_meli[2 * a.n_out + m];
          float o = 0.0f;
          for (int j = 0; j < n4; j += 4) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + j);
            const float* pp = P + s0 + j;
            o = fmaf(wv[3], pp[3], fmaf(wv[2], pp[2], fmaf(wv[1], pp[1], fmaf(wv[0], pp[0], o))));
          }
          s_go[m] = value_grad(o, trow[m]);
        }
      }
      wave_lds_sync();
      for (int m0 = 0; m0 < a.n_out; m0 += 64) {
        const int m = m0 + lane;
        if (m < a.n_out) {
          const int s0 = s_meli[m], n4 = s_meli[a.n_out + m];
          const float* w = s_melw + s_meli[2 * a.n_out + m];
          const float go = s_go[m];
          for (int j = 0; j < n4; ++j) {
            const float wj = w[j];
            if (wj != 0.0f) atomicAdd(&s_gv[s0 + j], wj * go);
          }
        }
      }
      wave_lds_sync();
#pragma unroll
      for (int i = 0; i < NUNP; ++i) {
        const int k = lane + 64 * i;
        if (k <= N2 / 2) { gk_[i] = s_gv[k]; gq_[i] = s_gv[N2 - k]; }
      }
    }
    // adjoint per bin pair; the inverse input in registers
    cpx zk_in[NUNP], zn_in[NUNP];
#pragma unroll
    for (int i = 0; i < NUNP; ++i) {
      const int k = lane + 64 * i;
      zk_in[i] = zn_in[i] = cmk(0.0f, 0.0f);
      if (k <= N2 / 2) {
        const cpx w = t_twu[64 * i];                      // W_N^k = e^{-2 pi i k / N}
        const cpx ck = xk_[i] * (2.0f * power_grad(gk_[i], vk_[i]));   // G[k]
        const cpx cq = xq_[i] * (2.0f * power_grad(gq_[i], vq_[i]));   // G[N2 - k]
        if (k == 0) {
          // edge bins (real X): a[0] = G[0] + G[N2], b[0] = G[0] - G[N2]; Zin[0] = a[0] + i b[0]
          zk_in[i] = cmk(ck.x + cq.x, ck.x - cq.x);
          zn_in[i] = zk_in[i];
        } else {
          const cpx wc = cmk(w.x, -w.y);                  // e^{+2 pi i k / N}
          const cpx bk = cmul(ck, wc);                    // b[k]
          const cpx bn = cmul(cq, cmk(-w.x, -w.y));       // b[N2 - k] = G[N2 - k] (-W^k)
          // Zin[k] = (a[k] + conj a[kn]) / 2 + i (b[k] + conj b[kn]) / 2, and the same with k <-> kn
          const cpx ak = cmk(0.5f * (ck.x + cq.x), 0.5f * (ck.y - cq.y)), an = cmk(ak.x, -ak.y);
          const cpx sk = cmk(0.5f * (bk.x + bn.x), 0.5f * (bk.y - bn.y)), sn = cmk(sk.x, -sk.y);
          zk_in[i] = cmk(ak.x - sk.y, ak.y + sk.x);
          zn_in[i] = cmk(an.x - sn.y, an.y + sn.x);
        }
      }
    }
    wave_lds_sync();                                      // every Z read is done
    // conj(Zin) in natural order (padded like Z), then the lane's R points of it
#pragma unroll
    for (int i = 0; i < NUNP; ++i) {
      const int k = lane + 64 * i;
      if (k <= N2 / 2) {
        const int kn = (N2 - k) & (N2 - 1);
        sA[IAS_S2_UP(k)] = cmk(zk_in[i].x, -zk_in[i].y);
        if (kn != k) sA[IAS_S2_UP(kn)] = cmk(zn_in[i].x, -zn_in[i].y);
      }
    }
    wave_lds_sync();
#pragma unroll
    for (int n1 = 0; n1 < R; ++n1) { const int p = 64 * n1 + lane; v[n1] = sA[IAS_S2_UP(p)]; }
    wave_lds_sync();
    fft(v);
    // z[m] = conj(out[m]) = y[2m] + i y[2m+1];  frame_grad = window * y
    if (SPAN) {
      const int rb = ((f - f_lo) * a.hop) & (NFFT - 1);     // ring position of the frame's first sample (hop is even)
#pragma unroll
      for (int n1 = 0; n1 < R; ++n1) {
        const int m = 64 * n1 + lane;
        const cpx o = sA[IAS_S2_UP(m)];
        const cpx w2 = t_win[64 * n1];
        cpx* rp = reinterpret_cast<cpx*>(ring + ((rb + 2 * m) & (NFFT - 1)));
        const cpx cur = *rp;
        *rp = cmk(cur.x + w2.x * o.x, cur.y - w2.y * o.y);
      }
      wave_lds_sync();
      float* sp = span + (size_t)(f - f_lo) * a.hop;         // the hop samples no later frame of the chunk reaches
      for (int i = lane; i < a.hop; i += 64) {
        const int idx = (rb + i) & (NFFT - 1);
        sp[i] = ring[idx];
        ring[idx] = 0.0f;
      }
      wave_lds_sync();
    } else {
      float* out = a.frame_grad + ((size_t)b * a.F + f) * NFFT;
#pragma unroll
      for (int n1 = 0; n1 < R; ++n1) {
        const int m = 64 * n1 + lane;
        const cpx o = sA[IAS_S2_UP(m)];
        const cpx w2 = t_win[64 * n1];
        *reinterpret_cast<cpx*>(out + 2 * m) = cmk(w2.x * o.x, -w2.y * o.y);
      }
      wave_lds_sync();
    }
  }
  if (SPAN && f_hi > f_lo) {                                 // the chunk's tail: what is left in the ring
    const int nf = f_hi - f_lo, rb = (nf * a.hop) & (NFFT - 1);
    float* sp = span + (size_t)nf * a.hop;
    for (int i = lane; i < NFFT - a.hop; i += 64) {
      const int idx = (rb + i) & (NFFT - 1);
      sp[i] = ring[idx];
      ring[idx] = 0.0f;
    }
    wave_lds_sync();
  }
  }
}

// Backward of the linear-bin losses for n_fft 2048 on the 8-points-per-lane core (round 3): the forward of
// stft2_kernel<.., NSUB = 2> (two 512-point half-transforms, in-lane combine, half-spectrum unpack), the cotangent per
// bin pair in registers, then the inverse 1024-point transform split the other way round (decimation in frequency):
//     A_p[k] = (Zin[k] + (-1)^p Zin[k + 512]) e^{+2 pi i k p / 1024},   z[2m + p] = IDFT_512(A_p)[m],   p = 0, 1,
// Zin[k + 512] reaches the lane that owns k through one LDS exchange (it is computed as the partner value of bin 512 - k),
// each IDFT_512 runs as the three forward passes on the conjugate, and a lane ends up with the four samples 4m .. 4m+3
// of its m = k1 + 8 d + 64 e: one 16-byte store per (lane, e).  Same arithmetic as stft_grad_wave_kernel<11, false>,
// which needs 198 VGPRs and 9.2 KB of scratch per wave (2 waves per SIMD) for its radix-16 first pass.
//
// SPAN: as in stft_grad_wave_kernel (a wave walks a chunk of G consecutive frames, overlap-add in an LDS ring, hop % 4
// == 0).  The ring is kept in 16-byte units q = sample / 4 at slot q ^ ((q >> 3) & 7): the lanes' units m = k1 + 8 d
// (+ 64 e) are 8 apart for consecutive lanes, the swizzle spreads them over the banks.
template <int SP_WAVES, bool SPAN>
__global__ __launch_bounds__(64 * SP_WAVES, SPAN ? 2 : 3 * SP_WAVES / 8) void stft_grad2k_kernel(const SgwArgs a, int nframes, unsigned magicF) {
  constexpr int SP_THREADS = 64 * SP_WAVES, N2 = 1024, HALF = 512, NFFT = 2048, SCR = 64 * IAS_S2_ROW, NB = N2 + 1;
  constexpr int NTAB = 16 + 8 + 8 + 8 + 8 + 16;   // stft2's 2048 section + the window at the lane's OUTPUT samples
  constexpr int V2_BASE_2048 = 32 + 32 + 32 + 18;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  cpx* s_scr = reinterpret_cast<cpx*>(smem);
  f32x4* s_ring = reinterpret_cast<f32x4*>(s_scr + SP_WAVES * SCR);         // SPAN: [wave][NFFT / 4]
  __shared__ cpx s_tab[NTAB * 64];                                          // an LDS object of its own (see stft2_kernel)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (SPAN) for (int i = tid; i < SP_WAVES * NFFT / 4; i += SP_THREADS) s_ring[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = tid; i < NTAB * 64; i += SP_THREADS) {
    const int pp = i >> 6, l = i & 63, src = V2_BASE_2048 + 2 * pp;
    s_tab[i] = cmk(a.tables[64 * src + l], a.tables[64 * (src + 1) + l]);
  }
  const cpx* t_win = s_tab + lane;              // [8 sub + n1]
  const cpx* t_tw1 = t_win + 64 * 16;
  const cpx* t_tw2 = t_tw1 + 64 * 8;
  const cpx* t_cmb = t_tw2 + 64 * 8;            // [e] -> W_1024^k
  const cpx* t_twu = t_cmb + 64 * 8;            // [e] -> W_2048^k
  const cpx* t_wo = t_twu + 64 * 8;             // [2 e + h] -> window at samples 4 m + 2 h + {0, 1}, m = k1 + 8 d + 64 e
  __syncthreads();
  cpx* sA = s_scr + wave * SCR;
  const int k1 = lane >> 3, dd = lane & 7, kl = k1 + 8 * dd;
  float c0 = 0.0f, c1 = 0.0f;
  if (a.loss_mode == 2) { c0 = (float)a.coef[0]; c1 = (float)a.coef[1]; }
  auto bin_value = [&](float p) { return a.power2 ? p : __builtin_amdgcn_sqrtf(a.loss_mode == 2 ? fmaxf(p, a.eps) : p); };   // v_sqrt_f32
  auto value_grad = [&](float v, float t) {
    const float d = v - t;
    const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    return a.loss_mode == 2 ? c0 * d + c1 * sg * __builtin_amdgcn_rcpf(v) : sg * a.scale;   // v_rcp_f32 (1 ulp)
  };
  auto power_grad = [&](float gv, float v) {
    if (a.power2) return gv;
    const bool live = a.loss_mode == 2 ? v > __builtin_amdgcn_sqrtf(a.eps) : v > 0.0f;
    return live ? 0.5f * gv * __builtin_amdgcn_rcpf(v) : 0.0f;
  };
  auto pad = [](int i) { return IAS_S2_UP(i); };   // i < 512: two pad elements per eight (see IAS_S2_ROW)
  // the three radix-8 passes of a 512-point transform: v = the lane's points 64 n1 + lane -> u[e] at kl + 64 e
  auto fft512 = [&](cpx (&v)[8], cpx (&u)[8]) {
    dft8(v);
    {
      const int c = lane & 7, aa = lane >> 3;
#pragma unroll
      for (int q = 0; q < 8; ++q) sA[(q * 8 + c) * IAS_S2_ROW + aa] = cmul(v[q], t_tw1[64 * q]);
    }
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) u[q] = sA[lane * IAS_S2_ROW + q];
    wave_lds_sync();
    dft8(u);
#pragma unroll
    for (int d = 0; d < 8; ++d) sA[(k1 * 8 + d) * IAS_S2_ROW + dd] = cmul(u[d], t_tw2[64 * d]);
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) u[q] = sA[lane * IAS_S2_ROW + q];
    wave_lds_sync();
    dft8(u);
  };

  const int gw = blockIdx.x * SP_WAVES + wave, nw = gridDim.x * SP_WAVES;
  f32x4* ring = s_ring + wave * (NFFT / 4);
  auto slot = [](int q) { return q ^ ((q >> 3) & 7); };
  for (int c = gw; c < (SPAN ? a.nchunks : nframes); c += nw) {
  int b, f_lo, f_hi;
  if (SPAN) { b = c / a.cper; f_lo = (c - b * a.cper) * a.G; f_hi = min(f_lo + a.G, a.F); }
  else {
    unsigned q0 = __umulhi((unsigned)c, magicF);
    int fr0 = c - (int)q0 * a.F;
    if (fr0 >= a.F) { fr0 -= a.F; ++q0; }
    b = (int)q0; f_lo = fr0; f_hi = fr0 + 1;
  }
  f32x4* span = reinterpret_cast<f32x4*>(a.frame_grad + (size_t)c * a.L);   // SPAN (L % 4 == 0)
  for (int fr = f_lo; fr < f_hi; ++fr) {
    const int fi = b * a.F + fr;
    float xc[32];
    stft2_load_frame<2>(a.audio + (size_t)b * a.T, a.T, a.hop, fr, lane, xc);
    const float* trow = a.target + (size_t)fi * NB;
    // the lane's target values are requested with the frame and consumed after the forward transform
    float tk_[8], tq_[8], th_ = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { tk_[e] = trow[kl + 64 * e]; tq_[e] = trow[N2 - kl - 64 * e]; }
    if (lane == 0) th_ = trow[HALF];
    // ---- forward: Z[k], Z[k + 512] for k = kl + 64 e
    cpx zlo[8], zhi[8];
    {
      cpx ue[8], u[8], v[8];
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) v[n1] = cmk(xc[2 * n1], xc[2 * n1 + 1]) * t_win[64 * n1];
      fft512(v, ue);
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) v[n1] = cmk(xc[16 + 2 * n1], xc[16 + 2 * n1 + 1]) * t_win[64 * (8 + n1)];
      fft512(v, u);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const cpx t = cmul(u[e], t_cmb[64 * e]); zlo[e] = cadd(ue[e], t); zhi[e] = csub(ue[e], t); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sA[pad(kl + 64 * e)] = zhi[e];
    wave_lds_sync();
    // ---- per bin pair (k, N2 - k): X, value, cotangent, the pair's two inverse inputs
    cpx zk_in[8], zn_in[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = kl + 64 * e;
      cpx zn = sA[pad((HALF - k) & (HALF - 1))];
      const cpx zk = zlo[e];
      if (e == 0 && k == 0) zn = zk;
      const cpx w = t_twu[64 * e];                              // W_N^k = e^{-2 pi i k / N}
      const cpx ze = cmk(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
      const cpx zo = cmk(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
      const cpx t = cmul(w, zo);
      const cpx xk = cadd(ze, t);                               // X[k]
      const cpx xq = cmk(ze.x - t.x, -(ze.y - t.y));            // X[N2 - k]
      const float vk = bin_value(xk.x * xk.x + xk.y * xk.y), vq = bin_value(xq.x * xq.x + xq.y * xq.y);
      const float gk = value_grad(vk, tk_[e]), gq = value_grad(vq, tq_[e]);
      const cpx ck = xk * (2.0f * power_grad(gk, vk));          // G[k]
      const cpx cq = xq * (2.0f * power_grad(gq, vq));          // G[N2 - k]
      if (e == 0 && k == 0) {
        zk_in[e] = cmk(ck.x + cq.x, ck.x - cq.x);              // edge bins (real X): a[0] + i b[0]
        zn_in[e] = zk_in[e];
      } else {
        const cpx bk = cmul(ck, cmk(w.x, -w.y));                // b[k] = G[k] e^{+2 pi i k / N}
        const cpx bn = cmul(cq, cmk(-w.x, -w.y));               // b[N2 - k] = G[N2 - k] (-W^k)
        const cpx ak = cmk(0.5f * (ck.x + cq.x), 0.5f * (ck.y - cq.y));
        const cpx sk = cmk(0.5f * (bk.x + bn.x), 0.5f * (bk.y - bn.y));
        zk_in[e] = cmk(ak.x - sk.y, ak.y + sk.x);              // Zin[k]
        zn_in[e] = cmk(ak.x + sk.y, -ak.y + sk.x);             // Zin[N2 - k] = conj(ak) + i conj(sk)
      }
    }
    // bin HALF pairs with itself: X[HALF] = conj(Z[HALF]) (lane 0 holds Z[HALF] = zhi[0])
    cpx zh_in;
    {
      const cpx xh = cmk(zhi[0].x, -zhi[0].y);
      const float vh = bin_value(xh.x * xh.x + xh.y * xh.y);
      const float gh = value_grad(vh, th_);
      const cpx ch = xh * (2.0f * power_grad(gh, vh));
      zh_in = cmk(ch.x, -ch.y);                                 // (ch + conj ch)/2 + i (i ch + conj(i ch))/2 = conj(ch)
    }
    wave_lds_sync();                                            // every Z read is done
    // ---- Zin[k + 512] to the lane that owns k: it was computed as the partner value of bin 512 - k
#pragma unroll
    for (int e = 0; e < 8; ++e) { const int k = kl + 64 * e; if (k != 0) sA[pad(HALF - k)] = zn_in[e]; }
    if (lane == 0) sA[0] = zh_in;
    wave_lds_sync();
    cpx a0[8], a1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const cpx zu = sA[pad(kl + 64 * e)];
      const cpx wc = t_cmb[64 * e];                             // W_1024^k; its conjugate is e^{+2 pi i k / 1024}
      a0[e] = cadd(zk_in[e], zu);
      a1[e] = cmul(csub(zk_in[e], zu), cmk(wc.x, -wc.y));
    }
    wave_lds_sync();
    // ---- the two inverse half-transforms, as forward passes on the conjugate
    cpx r0[8], r1[8];
    {
      cpx v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) sA[pad(kl + 64 * e)] = cmk(a0[e].x, -a0[e].y);
      wave_lds_sync();
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) v[n1] = sA[pad(64 * n1 + lane)];
      wave_lds_sync();
      fft512(v, r0);
      wave_lds_sync();
#pragma unroll
      for (int e = 0; e < 8; ++e) sA[pad(kl + 64 * e)] = cmk(a1[e].x, -a1[e].y);
      wave_lds_sync();
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) v[n1] = sA[pad(64 * n1 + lane)];
      wave_lds_sync();
      fft512(v, r1);
      wave_lds_sync();
    }
    // ---- frame_grad: samples 4 m .. 4 m + 3 of m = kl + 64 e = (y[2(2m)], y[2(2m)+1], y[2(2m+1)], y[2(2m+1)+1]) x window
    if (SPAN) {
      const int h4 = a.hop >> 2, rb = ((fr - f_lo) * h4) & (NFFT / 4 - 1);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int m = kl + 64 * e;
        const cpx w0 = t_wo[64 * (2 * e)], w1 = t_wo[64 * (2 * e + 1)];
        f32x4* rp = ring + slot((rb + m) & (NFFT / 4 - 1));
        const f32x4 cur = *rp;
        *rp = (f32x4){cur[0] + w0.x * r0[e].x, cur[1] - w0.y * r0[e].y, cur[2] + w1.x * r1[e].x, cur[3] - w1.y * r1[e].y};
      }
      wave_lds_sync();
      f32x4* sp = span + (size_t)(fr - f_lo) * h4;
      for (int i = lane; i < h4; i += 64) {
        f32x4* rp = ring + slot((rb + i) & (NFFT / 4 - 1));
        sp[i] = *rp;
        *rp = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
      }
      wave_lds_sync();
    } else {
      float* out = a.frame_grad + (size_t)fi * NFFT;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int m = kl + 64 * e;
        const cpx w0 = t_wo[64 * (2 * e)], w1 = t_wo[64 * (2 * e + 1)];
        *reinterpret_cast<f32x4*>(out + 4 * m) = (f32x4){w0.x * r0[e].x, -w0.y * r0[e].y, w1.x * r1[e].x, -w1.y * r1[e].y};
      }
    }
  }
  if (SPAN && f_hi > f_lo) {                                 // the chunk's tail: what is left in the ring
    const int h4 = a.hop >> 2, nf = f_hi - f_lo, rb = (nf * h4) & (NFFT / 4 - 1);
    f32x4* sp = span + (size_t)nf * h4;
    for (int i = lane; i < NFFT / 4 - h4; i += 64) {
      f32x4* rp = ring + slot((rb + i) & (NFFT / 4 - 1));
      sp[i] = *rp;
      *rp = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    wave_lds_sync();
  }
  }
}

// Backward of the linear-bin losses for n_fft 512, two frames per wave (the lane layout of stft2h_kernel), overlap-add
// inside the kernel (SPAN form only: ias_stft_grad_spans).  A wave walks its chunk two consecutive frames at a time:
// forward transform of both, cotangents per bin pair in the lane that owns the pair (lanes 0..31 frame A, 32..63 frame
// B), the inverse inputs Zin of both frames through the scratch in natural order, ONE more run of the three passes on
// their conjugates, then frame A's lanes add their windowed samples into the ring, after them frame B's (frame order:
// deterministic), and the 2 hop samples no later frame reaches leave the ring.
template <int SP_WAVES>
__global__ __launch_bounds__(64 * SP_WAVES, SP_WAVES / 2) void stft_grad512_kernel(const SgwArgs a) {
  constexpr int SP_THREADS = 64 * SP_WAVES, SCR = 64 * IAS_S2_ROW, NFFT = 512, N2 = 256, HALF = 128, NB = 257;
  constexpr int NTAB = 4 + 4 + 8 + 4 + 8;    // stft2h's tables + the window at the lane's OUTPUT samples
  extern __shared__ __attribute__((aligned(16))) float smem[];
  cpx* s_scr = reinterpret_cast<cpx*>(smem);
  float* s_ring = reinterpret_cast<float*>(s_scr + SP_WAVES * SCR);         // [wave][NFFT]
  __shared__ cpx s_tab[NTAB * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < SP_WAVES * NFFT; i += SP_THREADS) s_ring[i] = 0.0f;
  // n_fft 512 block of ias_stft_build_tables: [0,8) window, [8,16) pass 1, [16,32) pass 2, [38,46) unpack, [46,62) output window
  for (int i = tid; i < NTAB * 64; i += SP_THREADS) {
    const int pp = i >> 6, l = i & 63, src = pp < 16 ? 2 * pp : 38 + 2 * (pp - 16);
    s_tab[i] = cmk(a.tables[64 * src + l], a.tables[64 * (src + 1) + l]);
  }
  const cpx* t_win = s_tab + lane;
  const cpx* t_tw1 = t_win + 64 * 4;
  const cpx* t_tw2 = t_tw1 + 64 * 4;
  const cpx* t_twu = t_tw2 + 64 * 8;          // [e] -> W_512^k, k = kl + 32 e (e < 4)
  const cpx* t_wo = t_twu + 64 * 4;           // [e] -> window at samples 2 m, 2 m + 1, m = kl + 32 e (e < 8)
  __syncthreads();
  cpx* sA = s_scr + wave * SCR;
  float* ring = s_ring + wave * NFFT;
  const int slot = lane >> 3, dd = lane & 7, fr = lane >> 5, kl = (slot & 3) + 4 * dd;
  float c0 = 0.0f, c1 = 0.0f;
  if (a.loss_mode == 2) { c0 = (float)a.coef[0]; c1 = (float)a.coef[1]; }
  auto bin_value = [&](float p) { return a.power2 ? p : __builtin_amdgcn_sqrtf(a.loss_mode == 2 ? fmaxf(p, a.eps) : p); };
  auto value_grad = [&](float v, float t) {
    const float d = v - t;
    const float sg = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
    return a.loss_mode == 2 ? c0 * d + c1 * sg * __builtin_amdgcn_rcpf(v) : sg * a.scale;
  };
  auto power_grad = [&](float gv, float v) {
    if (a.power2) return gv;
    const bool live = a.loss_mode == 2 ? v > __builtin_amdgcn_sqrtf(a.eps) : v > 0.0f;
    return live ? 0.5f * gv * __builtin_amdgcn_rcpf(v) : 0.0f;
  };
  auto pad = [](int i) { return IAS_S2_UP(i); };   // i < 512: two pad elements per eight (see IAS_S2_ROW)
  // v[0..3] / v[4..7]: the lane's points 64 n1 + lane of frame A / B  ->  u[e] = transform of the lane's own frame
  // (slot >> 2) at kl + 32 e
  auto fft2 = [&](cpx (&v)[8], cpx (&u)[8]) {
    cpx (&va)[4] = reinterpret_cast<cpx (&)[4]>(v[0]);
    cpx (&vb)[4] = reinterpret_cast<cpx (&)[4]>(v[4]);
    dftR<4>(va); dftR<4>(vb);
    {
      const int c = lane & 7, aa = lane >> 3;
#pragma unroll
      for (int q = 0; q < 8; ++q) sA[(q * 8 + c) * IAS_S2_ROW + aa] = cmul(v[q], t_tw1[64 * (q & 3)]);
    }
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) u[q] = sA[lane * IAS_S2_ROW + q];
    wave_lds_sync();
    dft8(u);
#pragma unroll
    for (int d = 0; d < 8; ++d) sA[(slot * 8 + d) * IAS_S2_ROW + dd] = cmul(u[d], t_tw2[64 * d]);
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < 8; ++q) u[q] = sA[lane * IAS_S2_ROW + q];
    wave_lds_sync();
    dft8(u);
  };

  const int cstep = (int)gridDim.x * SP_WAVES;
  for (int c = (int)blockIdx.x * SP_WAVES + wave; c < a.nchunks; c += cstep) {
    const int b = c / a.cper, f_lo = (c - b * a.cper) * a.G, f_hi = min(f_lo + a.G, a.F);
    const float* arow = a.audio + (size_t)b * a.T;
    float* span = a.frame_grad + (size_t)c * a.L;
    for (int f = f_lo; f < f_hi; f += 2) {
      const bool hasB = f + 1 < f_hi;                         // wave-uniform
      const bool own = fr == 0 || hasB;
      float xc[16];
      {
        float (&xa)[8] = reinterpret_cast<float (&)[8]>(xc[0]);
        float (&xb)[8] = reinterpret_cast<float (&)[8]>(xc[8]);
        load_frame<4, 256>(arow, a.T, a.hop, f, lane, xa);
        load_frame<4, 256>(arow, a.T, a.hop, hasB ? f + 1 : f, lane, xb);
      }
      const float* trow = a.target + ((size_t)b * a.F + f + (own ? fr : 0)) * NB;
      float tk_[4], tq_[4], th_ = 0.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { tk_[e] = trow[kl + 32 * e]; tq_[e] = trow[N2 - kl - 32 * e]; }
      if (kl == 0) th_ = trow[HALF];
      // ---- forward
      cpx u[8];
      {
        cpx v[8];
#pragma unroll
        for (int n1 = 0; n1 < 4; ++n1) {
          v[n1] = cmk(xc[2 * n1], xc[2 * n1 + 1]) * t_win[64 * n1];
          v[4 + n1] = cmk(xc[8 + 2 * n1], xc[8 + 2 * n1 + 1]) * t_win[64 * n1];
        }
        fft2(v, u);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) sA[fr * HALF + kl + 32 * e] = u[4 + e];
      wave_lds_sync();
      // ---- per bin pair (k, N2 - k): X, value, cotangent, the pair's two inverse inputs
      cpx zk_in[4], zn_in[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kl + 32 * e;
        cpx zn = sA[fr * HALF + ((HALF - k) & (HALF - 1))];
        const cpx zk = u[e];
        if (e == 0 && k == 0) zn = zk;
        const cpx w = t_twu[64 * e];                              // W_N^k = e^{-2 pi i k / N}
        const cpx ze = cmk(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const cpx zo = cmk(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        const cpx t = cmul(w, zo);
        const cpx xk = cadd(ze, t);                               // X[k]
        const cpx xq = cmk(ze.x - t.x, -(ze.y - t.y));            // X[N2 - k]
        const float vk = bin_value(xk.x * xk.x + xk.y * xk.y), vq = bin_value(xq.x * xq.x + xq.y * xq.y);
        const float gk = value_grad(vk, tk_[e]), gq = value_grad(vq, tq_[e]);
        const cpx ck = xk * (2.0f * power_grad(gk, vk));          // G[k]
        const cpx cq = xq * (2.0f * power_grad(gq, vq));          // G[N2 - k]
        if (e == 0 && k == 0) {
          zk_in[e] = cmk(ck.x + cq.x, ck.x - cq.x);              // edge bins (real X): a[0] + i b[0]
          zn_in[e] = zk_in[e];
        } else {
          const cpx bk = cmul(ck, cmk(w.x, -w.y));                // b[k] = G[k] e^{+2 pi i k / N}
          const cpx bn = cmul(cq, cmk(-w.x, -w.y));               // b[N2 - k] = G[N2 - k] (-W^k)
          const cpx ak = cmk(0.5f * (ck.x + cq.x), 0.5f * (ck.y - cq.y));
          const cpx sk = cmk(0.5f * (bk.x + bn.x), 0.5f * (bk.y - bn.y));
          zk_in[e] = cmk(ak.x - sk.y, ak.y + sk.x);              // Zin[k]
          zn_in[e] = cmk(ak.x + sk.y, -ak.y + sk.x);             // Zin[N2 - k]
        }
      }
      // bin HALF pairs with itself: X[HALF] = conj(Z[HALF]) (the lanes with kl = 0 hold Z[HALF] = u[4])
      cpx zh_in;
      {
        const cpx xh = cmk(u[4].x, -u[4].y);
        const float vh = bin_value(xh.x * xh.x + xh.y * xh.y);
        const float gh = value_grad(vh, th_);
        const cpx ch = xh * (2.0f * power_grad(gh, vh));
        zh_in = cmk(ch.x, -ch.y);
      }
      wave_lds_sync();                                            // every Z read is done
      // ---- conj(Zin) of both frames in natural order (frame-major, padded), then the lane's points of it
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kl + 32 * e;
        sA[pad(fr * N2 + k)] = cmk(zk_in[e].x, -zk_in[e].y);
        if (k != 0) sA[pad(fr * N2 + N2 - k)] = cmk(zn_in[e].x, -zn_in[e].y);
      }
      if (kl == 0) sA[pad(fr * N2 + HALF)] = cmk(zh_in.x, -zh_in.y);
      wave_lds_sync();
      cpx r[8];
      {
        cpx v[8];
#pragma unroll
        for (int n1 = 0; n1 < 4; ++n1) { v[n1] = sA[pad(64 * n1 + lane)]; v[4 + n1] = sA[pad(N2 + 64 * n1 + lane)]; }
        wave_lds_sync();
        fft2(v, r);
      }
      wave_lds_sync();
      // ---- z[m] = conj(out[m]) = y[2m] + i y[2m+1], m = kl + 32 e: window, into the ring -- frame A's lanes, then B's
      // (the ring holds ONE frame length: frame A's first hop samples have to leave before frame B's last hop samples,
      // which wrap onto them, arrive -- per frame: add, then flush the hop samples no later frame reaches)
      const int rbA = ((f - f_lo) * a.hop) & (NFFT - 1);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (half == 1 && !hasB) break;                            // wave-uniform
        const int rb = (rbA + half * a.hop) & (NFFT - 1);
        if (fr == half) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int m = kl + 32 * e;
            const cpx w2 = t_wo[64 * e];
            cpx* rp = reinterpret_cast<cpx*>(ring + ((rb + 2 * m) & (NFFT - 1)));
            const cpx cur = *rp;
            *rp = cmk(cur.x + w2.x * r[e].x, cur.y - w2.y * r[e].y);
          }
        }
        wave_lds_sync();
        float* sp = span + (size_t)(f + half - f_lo) * a.hop;
        for (int i = lane; i < a.hop; i += 64) {
          const int idx = (rb + i) & (NFFT - 1);
          sp[i] = ring[idx];
          ring[idx] = 0.0f;
        }
        wave_lds_sync();
      }
    }
    if (f_hi > f_lo) {                                            // the chunk's tail: what is left in the ring
      const int nf = f_hi - f_lo, rb = (nf * a.hop) & (NFFT - 1);
      float* sp = span + (size_t)nf * a.hop;
      for (int i = lane; i < NFFT - a.hop; i += 64) {
        const int idx = (rb + i) & (NFFT - 1);
        sp[i] = ring[idx];
        ring[idx] = 0.0f;
      }
      wave_lds_sync();
    }
  }
}

// frame_grad [B,F,n_fft] <- d loss / d (windowed frames) (see stft_grad_wave_kernel); tables: ias_stft_build_tables of
// the plan's window; mel_*: the forward's CSR filterbank or NULL (linear bins, n_out = n_fft/2+1); target [B,F,n_out];
// coef: device doubles [2] (loss_mode 2, linear bins only).
// plan != NULL: the SPAN kernels (overlap-add inside the kernel); frame_grad then receives B * plan[1] chunk spans of
// plan[2] floats (plan[0] = frames per chunk), IAS_ERR_UNSUPPORTED when the shape does not allow it.
static int grad_frames_launch(const float* audio, const float* tables, const int* mel_start, const int* mel_count,
                              const int* mel_woff, const float* mel_w, int mel_nnz, int n_out, const float* target,
                              const double* coef, float* frame_grad, int B, int T, int n_fft, int hop, int power,
                              int loss_mode, float scale, float eps, int* plan, hipStream_t stream, bool dry = false) {
  if (!dry && (!audio || !tables || !target || !frame_grad)) return IAS_ERR_ARG;
  if (B <= 0 || B > 65535 || hop <= 0) return IAS_ERR_ARG;
  if (n_fft != 512 && n_fft != 1024 && n_fft != 2048) return IAS_ERR_UNSUPPORTED;
  if ((power != 1 && power != 2) || (loss_mode != 1 && loss_mode != 2) || (!dry && loss_mode == 2 && !coef)) return IAS_ERR_ARG;
  const bool mel = mel_start != nullptr || (dry && mel_nnz > 0);
  if (!dry && mel && (!mel_count || !mel_woff || !mel_w || mel_nnz <= 0 || n_out <= 0 || loss_mode != 1)) return IAS_ERR_ARG;
  if (!mel && n_out != n_fft / 2 + 1) return IAS_ERR_ARG;
  const int F = ias_stft_num_frames(T, n_fft, hop);
  if (F < 0 || F > 2147483647 / n_fft) return IAS_ERR_ARG;
  const bool span = plan != nullptr;
  SgwArgs a;
  a.audio = audio; a.tables = tables; a.target = target; a.coef = coef; a.frame_grad = frame_grad;
  a.T = T; a.F = F; a.hop = hop; a.power2 = power == 2; a.loss_mode = loss_mode; a.scale = scale; a.eps = eps;
  a.mel_start = mel_start; a.mel_count = mel_count; a.mel_woff = mel_woff; a.mel_w = mel_w;
  a.n_out = n_out; a.mel_nnz = mel ? mel_nnz : 0;
  a.G = a.cper = a.L = a.nchunks = 0;
  static const int wgs_env = ias_diag_env("IAS_STFT_GRAD_WGS") ? atoi(ias_diag_env("IAS_STFT_GRAD_WGS")) : 0;   // diagnostics
  int per_row = (wgs_env > 0 ? wgs_env : 1024) / B;   // one resident round (measured: 2.36 -> 2.29 ms for the MR-STFT loss)
  if (per_row < 1) per_row = 1;
  int g = (F + per_row - 1) / per_row;
  if (g < 8) g = 8;
  a.groups = g;
  int ncu = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  if (span && (hop > n_fft / 2 || (hop & 1) || (long long)B * F >= 2000000000LL)) return IAS_ERR_UNSUPPORTED;
  // chunks for `nwaves` resident waves: every wave one chunk of G consecutive frames (G hop >= n_fft - hop: a sample then
  // lies in at most two chunk spans), chunks never cross rows
  auto make_plan = [&](long long nwaves) -> bool {
    long long cper0 = nwaves / B;
    if (cper0 < 1) cper0 = 1;
    int G = (int)((F + cper0 - 1) / cper0);
    const int gmin = (n_fft - hop + hop - 1) / hop;
    if (G < gmin) G = gmin;
    if (G >= F) G = F;
    const int cper = (F + G - 1) / G;
    const long long L = (long long)(G - 1) * hop + n_fft;
    if ((long long)B * cper * L > (long long)B * F * n_fft || (long long)B * cper > 2000000000LL) return false;
    a.G = G; a.cper = cper; a.L = (int)L; a.nchunks = B * cper;
    plan[0] = G; plan[1] = cper; plan[2] = (int)L;
    return true;
  };
  static const int v1_2k = ias_diag_env("IAS_STFT_V1") ? atoi(ias_diag_env("IAS_STFT_V1")) : 0;   // diagnostics: round-2 kernels
  if (n_fft == 2048 && !mel && !v1_2k && (long long)B * F < 2000000000LL && (reinterpret_cast<uintptr_t>(frame_grad) & 15) == 0 &&
      (!span || (hop & 3) == 0)) {
    constexpr int W2 = 8;
    const size_t lds2 = sizeof(cpx) * (W2 * 64 * IAS_S2_ROW + (span ? W2 * 1024 : 0));       // + 32 KB of static tables
    if (span) {
      if (!make_plan((long long)ncu * W2)) return IAS_ERR_UNSUPPORTED;
      if (dry) return IAS_OK;
      const long long need = ((long long)a.nchunks + W2 - 1) / W2;
      const int grid2 = (int)(need < (long long)ncu ? need : (long long)ncu);
      (void)hipFuncSetAttribute((const void*)stft_grad2k_kernel<W2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      hipLaunchKernelGGL((stft_grad2k_kernel<W2, true>), dim3(grid2), dim3(64 * W2), lds2, stream, a, B * F, 0u);
    } else {
      const long long need = ((long long)B * F + W2 - 1) / W2;
      const int grid2 = (int)(need < 2LL * ncu ? need : 2LL * ncu);
      (void)hipFuncSetAttribute((const void*)stft_grad2k_kernel<W2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      hipLaunchKernelGGL((stft_grad2k_kernel<W2, false>), dim3(grid2), dim3(64 * W2), lds2, stream, a, B * F,
                         (F == 1 ? 0xFFFFFFFFu /* 2^32 / 1 does not fit: q0 = fi - 1, which row_of's one-step correction fixes */ : (unsigned)(0x100000000ULL / (unsigned long long)F)));
    }
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }
  static const int v1_512 = ias_diag_env("IAS_STFT_V1") ? atoi(ias_diag_env("IAS_STFT_V1")) : 0;
  if (n_fft == 512 && span && !mel && !v1_512) {
    constexpr int W5 = 8;
    const size_t lds5 = sizeof(cpx) * (W5 * 64 * IAS_S2_ROW) + sizeof(float) * W5 * 512;
    (void)hipFuncSetAttribute((const void*)stft_grad512_kernel<W5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds5);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stft_grad512_kernel<W5>, 64 * W5, lds5) != hipSuccess || nb < 1) nb = 1;
    if (!make_plan((long long)ncu * nb * W5)) return IAS_ERR_UNSUPPORTED;
    if (dry) return IAS_OK;
    const long long need = ((long long)a.nchunks + W5 - 1) / W5;
    const int grid5 = (int)(need < (long long)ncu * nb ? need : (long long)ncu * nb);
    hipLaunchKernelGGL((stft_grad512_kernel<W5>), dim3(grid5), dim3(64 * W5), lds5, stream, a);
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }
  const int R = n_fft / 128, scr = 8 * R * IAS_S2_ROW, np_it = (8 * R + 63) / 64, nunp = (n_fft / 4) / 64 + 1;
  const size_t lds_static = sizeof(float) * 64 * (2 * R + 2 * R + 16 * np_it + 2 * nunp);   // the kernel's table object
  size_t lds = sizeof(cpx) * 4 * scr + 16;
  if (span) lds += sizeof(float) * 4 * n_fft;
  if (mel)
    lds += sizeof(float) * (mel_padded_words(mel_nnz, n_out) + ((3 * n_out + 3) & ~3) +
                            4 * (((n_out + 3) & ~3) + n_fft / 2 + 1 + 7));
  if (lds + lds_static > 150 * 1024) return IAS_ERR_UNSUPPORTED;
  dim3 grid((F + g - 1) / g, B), block(256);
#define IAS_SGW_LAUNCH(LOG2N, MEL, SPAN)                                                                          \
  do {                                                                                                             \
    if (lds + lds_static > 48 * 1024)                                                                              \
      (void)hipFuncSetAttribute((const void*)stft_grad_wave_kernel<LOG2N, MEL, SPAN>,                              \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
    if (SPAN) {                                                                                                    \
      int nb = 0;                                                                                                  \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, stft_grad_wave_kernel<LOG2N, MEL, SPAN>, 256, lds) !=  \
              hipSuccess || nb < 1)                                                                                \
        nb = 1;                                                                                                    \
      if (!make_plan((long long)ncu * nb * 4)) return IAS_ERR_UNSUPPORTED;                                         \
      if (dry) return IAS_OK;                                                                                      \
      const long long need = ((long long)a.nchunks + 3) / 4;                                                       \
      grid = dim3((unsigned)(need < (long long)ncu * nb ? need : (long long)ncu * nb));                            \
    }                                                                                                              \
    hipLaunchKernelGGL((stft_grad_wave_kernel<LOG2N, MEL, SPAN>), grid, block, lds, stream, a);                    \
  } while (0)
#define IAS_SGW_PICK(MEL, SPAN)                                                                                    \
  do {                                                                                                             \
    if (n_fft == 512) IAS_SGW_LAUNCH(9, MEL, SPAN);                                                                \
    else if (n_fft == 1024) IAS_SGW_LAUNCH(10, MEL, SPAN);                                                         \
    else IAS_SGW_LAUNCH(11, MEL, SPAN);                                                                            \
  } while (0)
  if (mel) { if (span) IAS_SGW_PICK(true, true); else IAS_SGW_PICK(true, false); }
  else { if (span) IAS_SGW_PICK(false, true); else IAS_SGW_PICK(false, false); }
#undef IAS_SGW_PICK
#undef IAS_SGW_LAUNCH
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

extern "C" int ias_stft_grad_frames(const float* audio, const float* tables, const int* mel_start, const int* mel_count,
                                    const int* mel_woff, const float* mel_w, int mel_nnz, int n_out, const float* target,
                                    const double* coef, float* frame_grad, int B, int T, int n_fft, int hop, int power,
                                    int loss_mode, float scale, float eps, void* stream_) {
  return grad_frames_launch(audio, tables, mel_start, mel_count, mel_woff, mel_w, mel_nnz, n_out, target, coef, frame_grad,
                            B, T, n_fft, hop, power, loss_mode, scale, eps, nullptr, (hipStream_t)stream_);
}

// The chunk plan ias_stft_grad_spans will use for this shape (the device's occupancy enters it), without launching:
// plan_host[3] = {G, chunks per row, floats per chunk span}.  mel_nnz / n_out as for the launch (mel_nnz = 0: linear bins).
extern "C" int ias_stft_grad_span_plan(int B, int T, int n_fft, int hop, int mel_nnz, int n_out, int* plan_host) {
  if (!plan_host) return IAS_ERR_ARG;
  if (mel_nnz > 0 && (n_out <= 0 || n_out > n_fft / 2 + 1)) return IAS_ERR_ARG;
  return grad_frames_launch(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, mel_nnz, n_out, nullptr, nullptr, nullptr,
                            B, T, n_fft, hop, 1, 1, 1.0f, 0.0f, plan_host, nullptr, true);
}

// The same with the overlap-add inside the kernel: chunk_spans (B * plan[1] * plan[2] floats, 16-byte aligned) receives
// B * plan_host[1] spans of plan_host[2] floats, span c = row c / plan[1], frames [j G, j G + G) with j = c % plan[1],
// G = plan_host[0]: span[i] = sum over the chunk's frames f covering padded sample j G hop + i of window x frame
// gradient, in frame order.  stft_grad_combine_kernel (ias_stft_loss_backward) adds the (at most two) chunks per sample.
extern "C" int ias_stft_grad_spans(const float* audio, const float* tables, const int* mel_start, const int* mel_count,
                                   const int* mel_woff, const float* mel_w, int mel_nnz, int n_out, const float* target,
                                   const double* coef, float* chunk_spans, int B, int T, int n_fft, int hop, int power,
                                   int loss_mode, float scale, float eps, int* plan_host, void* stream_) {
  if (!plan_host || (reinterpret_cast<uintptr_t>(chunk_spans) & 15) != 0) return IAS_ERR_ARG;
  return grad_frames_launch(audio, tables, mel_start, mel_count, mel_woff, mel_w, mel_nnz, n_out, target, coef, chunk_spans,
                            B, T, n_fft, hop, power, loss_mode, scale, eps, plan_host, (hipStream_t)stream_);
}

// sums[0..2] = sum over n partial triples (fixed order: deterministic); optionally
// mean_out[0] = (float)(sums[0] * scale)  (the L1 mean, without further elementwise launches)
#define RP_THREADS 256    // 4 waves: finds a free CU slot beside the FFT kernels (the 16-wave form waited 20 us for one)
__global__ __launch_bounds__(RP_THREADS) void reduce_partials_kernel(const double* __restrict__ partials, long long n,
                                                                     double* __restrict__ sums, double scale,
                                                                     float* __restrict__ mean_out) {
  __shared__ double s[RP_THREADS][3];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (long long i = threadIdx.x; i < n; i += RP_THREADS) {
    a0 += partials[i * 3]; a1 += partials[i * 3 + 1]; a2 += partials[i * 3 + 2];
  }
  s[threadIdx.x][0] = a0; s[threadIdx.x][1] = a1; s[threadIdx.x][2] = a2;
  __syncthreads();
  for (int d = RP_THREADS / 2; d > 0; d >>= 1) {
    if (threadIdx.x < d)
      for (int k = 0; k < 3; ++k) s[threadIdx.x][k] += s[threadIdx.x + d][k];
    __syncthreads();
  }
  if (threadIdx.x < 3) sums[threadIdx.x] = s[0][threadIdx.x];
  if (threadIdx.x == 0 && mean_out != nullptr) mean_out[0] = (float)(s[0][0] * scale);
}

// ------------------------------------------------------------------------ C ABI
extern "C" int ias_stft_num_frames(int T, int n_fft, int hop) {
  if (T <= n_fft / 2 || hop <= 0 || n_fft <= 0) return IAS_ERR_ARG;
  return 1 + T / hop;   // center=True: 1 + (T + 2*(n_fft/2) - n_fft) / hop
}

static size_t stft_lds_bytes(int n_fft, int mel_nnz, int n_out) {
  const int R = n_fft / 128, scr = 8 * R * 9;
  const int np_it = (8 * R + 63) / 64, nunp = (n_fft / 4) / 64 + 1;
  const int ntab = 64 * (2 * R + 2 * R + 16 * np_it + 2 * nunp);
  return sizeof(cpx) * stft_waves(n_fft) * scr + sizeof(float) * mel_padded_words(mel_nnz, mel_nnz ? n_out : 0) +
         sizeof(int) * 3 * (mel_nnz ? n_out : 0) + sizeof(float) * ntab + 8;
}
// Consecutive frames of a row per workgroup (the per-lane tables and the re-packed mel weights are set up once per
// workgroup).  ONE resident round: 4-wave form ~1024 workgroups (4 per CU x 256 CUs), 10-wave form 512.  Round 1 ran two
// rounds (2048 workgroups, 3 % faster then); with the heavier per-workgroup set-up of this round one round is faster both
// alone (85.1 -> 78.9 us) and in the pipelined step (0.2085 -> 0.1983 ms, scripts/diag/run_bench_stftwgs.sh).
static int stft_groups(int B, int F, int n_fft) {
  const int waves = stft_waves(n_fft);
  static const int wgs_env = ias_diag_env("IAS_STFT_WGS") ? atoi(ias_diag_env("IAS_STFT_WGS")) : 0;   // diagnostics
  int per_row = (wgs_env > 0 ? wgs_env : (waves == 12 ? 256 : (waves >= 8 ? 512 : 1024))) / B;
  if (per_row < 1) per_row = 1;
  int g = (F + per_row - 1) / per_row;
  const int gmin = 2 * waves, gmax = 32 * waves;
  if (g < gmin) g = gmin;
  if (g > gmax) g = gmax;
  return g;
}
static int stft_grid_x(int B, int F, int n_fft, int hop) {
  (void)hop;
  const int g = stft_groups(B, F, n_fft);
  return (F + g - 1) / g;
}

// the matrix-core form (csrc/stft_mfma_kernels.hip)
bool ias_sm_enabled(int n_fft, bool have_mtables);
long long ias_sm_partials(long long nframes);
int ias_sm_launch(const float* audio, const float* mtab, bool mel, float* out, const float* target, double* partials,
                  const float* rowpeak, int* ticket, int B, int T, int F, int n_fft, int hop, int n_out, int value_mode,
                  int loss_mode, float eps, hipStream_t stream);

// have_mtables: the ias_stft call will be given an ias_stft_build_mtables block (the matrix-core kernel writes one
// record per 16-frame group and wave, the VALU kernel one per workgroup)
// the round-3 radix-8 kernel (stft2_kernel): n_fft 1024; mel plans need their segment-major tables
static int stft2_waves() {
  static const int env = ias_diag_env("IAS_STFT2_WAVES") ? atoi(ias_diag_env("IAS_STFT2_WAVES")) : 0;   // diagnostics: 4, 5, 8 or 10
  return (env == 4 || env == 5 || env == 8 || env == 10) ? env : 8;
}
static bool stft2_enabled(int n_fft, bool mel, bool have_segtab) {
  static const int v1 = ias_diag_env("IAS_STFT_V1") ? atoi(ias_diag_env("IAS_STFT_V1")) : 0;   // diagnostics: round-2 kernel
  return !v1 && ((n_fft == 1024 && (!mel || have_segtab)) || ((n_fft == 2048 || n_fft == 512) && !mel));
}
static int stft2_grid(long long nframes, int n_fft) {
  static const int env = ias_diag_env("IAS_STFT2_WGS") ? atoi(ias_diag_env("IAS_STFT2_WGS")) : 0;   // diagnostics
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    ncu = v;
  }
  const int waves = n_fft == 1024 ? stft2_waves() : 8;
  if (n_fft == 512) nframes = (nframes + 1) / 2;             // two frames per wave
  const long long need = (nframes + waves - 1) / waves;
  // workgroups per CU by LDS (scratch 4.5 KB per wave + 25 KB of tables per workgroup): 10 waves: 2, 8: 2, 5: 3, 4: 3
  const long long cap = env > 0 ? env : (waves >= 8 ? 2LL : 3LL) * ncu;
  return (int)(need < cap ? need : cap);
}

// which kernel an ias_stft call with these arguments runs decides how many records it writes:
// flags bit 0: an ias_stft_build_mtables block will be given, bit 1: mel filters, bit 2: an ias_stft_build_segtab block
extern "C" long long ias_stft_partials_count(int B, int T, int n_fft, int hop, int flags) {
  const int F = ias_stft_num_frames(T, n_fft, hop);
  if (F < 0 || B <= 0) return IAS_ERR_ARG;
  if (ias_sm_enabled(n_fft, (flags & 1) != 0)) return ias_sm_partials((long long)B * F);
  if (stft2_enabled(n_fft, (flags & 2) != 0, (flags & 4) != 0)) return stft2_grid((long long)B * F, n_fft);
  return (long long)B * stft_grid_x(B, F, n_fft, hop);
}

// Lane-major constant tables of the STFT kernel, built on the HOST from the (zero-padded, centred)
// window [n_fft]: floats = 64 * (2R + 2R + 16*NP_IT + 2*NUNP) with R = n_fft/128.
extern "C" int ias_stft_tables_len(int n_fft) {
  if (n_fft != 512 && n_fft != 1024 && n_fft != 2048) return IAS_ERR_UNSUPPORTED;
  const int N2 = n_fft / 2, R = N2 / 64, np_it = (8 * R + 63) / 64, nunp = (N2 / 2) / 64 + 1;
  // n_fft 1024: + the unpack twiddles of stft2_kernel (bins k = (lane >> 3) + 8 (lane & 7) + 64 e, e < 4);
  // n_fft 2048: + stft2_kernel<NSUB = 2>'s whole table section (window pairs of the two half-transforms, pass-1 / pass-2
  // twiddles of a 512-point transform, combining twiddles W_1024^k, unpack twiddles W_2048^k, and for the backward
  // (stft_grad2k_kernel) the window at the lane's output samples: 64 complex per lane)
  return 64 * (2 * R + 2 * R + 16 * np_it + 2 * nunp + (n_fft == 1024 ? 8 : 0) + (n_fft == 512 ? 24 : 0) + (n_fft == 2048 ? 128 : 0));
}

extern "C" int ias_stft_build_tables(int n_fft, const float* window_host, float* out_host) {
  const int len = ias_stft_tables_len(n_fft);
  if (len < 0) return len;
  if (!window_host || !out_host) return IAS_ERR_ARG;
  const int N2 = n_fft / 2, R = N2 / 64, np_it = (8 * R + 63) / 64, nunp = (N2 / 2) / 64 + 1;
  const double w0 = 6.283185307179586 / (double)n_fft;
  float* o = out_host;
  for (int e = 0; e < 2 * R; ++e)
    for (int l = 0; l < 64; ++l) o[64 * e + l] = window_host[2 * (64 * (e >> 1) + l) + (e & 1)];
  o += 64 * 2 * R;
  for (int n1 = 0; n1 < R; ++n1)
    for (int l = 0; l < 64; ++l) {
      const double ang = w0 * (double)(2 * l * n1);
      o[64 * (2 * n1) + l] = (float)cos(ang);
      o[64 * (2 * n1 + 1) + l] = (float)(-sin(ang));
    }
  o += 64 * 2 * R;
  for (int i = 0; i < np_it; ++i)
    for (int d = 0; d < 8; ++d)
      for (int l = 0; l < 64; ++l) {
        const int c = (l + 64 * i) & 7;
        const double ang = w0 * (double)((n_fft / 64) * c * d);
        o[64 * (2 * (8 * i + d)) + l] = (float)cos(ang);
        o[64 * (2 * (8 * i + d) + 1) + l] = (float)(-sin(ang));
      }
  o += 64 * 2 * 8 * np_it;
  for (int i = 0; i < nunp; ++i)
    for (int l = 0; l < 64; ++l) {
      const int k = l + 64 * i;
      const double ang = w0 * (double)(k <= N2 / 2 ? k : 0);
      o[64 * (2 * i) + l] = (float)cos(ang);
      o[64 * (2 * i + 1) + l] = (float)(-sin(ang));
    }
  o += 64 * 2 * nunp;
  if (n_fft == 1024)
    for (int e = 0; e < 4; ++e)
      for (int l = 0; l < 64; ++l) {
        const double ang = w0 * (double)((l >> 3) + 8 * (l & 7) + 64 * e);
        o[64 * (2 * e) + l] = (float)cos(ang);
        o[64 * (2 * e + 1) + l] = (float)(-sin(ang));
      }
  if (n_fft == 512)                                        // stft2h_kernel: bins k = ((l >> 3) & 3) + 4 (l & 7) + 32 e, e < 4
    for (int e = 0; e < 4; ++e)
      for (int l = 0; l < 64; ++l) {
        const double ang = w0 * (double)(((l >> 3) & 3) + 4 * (l & 7) + 32 * e);
        o[64 * (2 * e) + l] = (float)cos(ang);
        o[64 * (2 * e + 1) + l] = (float)(-sin(ang));
      }
  if (n_fft == 512)                                        // stft_grad512_kernel: the window at samples 2 m, 2 m + 1, m = kl + 32 e
    for (int e = 0; e < 8; ++e)
      for (int l = 0; l < 64; ++l) {
        const int m = ((l >> 3) & 3) + 4 * (l & 7) + 32 * e;
        o[64 * (8 + 2 * e) + l] = window_host[2 * m];
        o[64 * (8 + 2 * e + 1) + l] = window_host[2 * m + 1];
      }
  if (n_fft == 2048) {
    const double tau = 6.283185307179586;
    for (int l = 0; l < 64; ++l) {
      int e = 0;                                         // entry pairs (re, im) in s_tab order
      auto put = [&](double re, double im) { o[64 * (2 * e) + l] = (float)re; o[64 * (2 * e + 1) + l] = (float)im; ++e; };
      for (int sub = 0; sub < 2; ++sub)                  // window at samples 256 n1 + 4 l + 2 sub + {0, 1}
        for (int n1 = 0; n1 < 8; ++n1) put(window_host[256 * n1 + 4 * l + 2 * sub], window_host[256 * n1 + 4 * l + 2 * sub + 1]);
      for (int k1 = 0; k1 < 8; ++k1) put(cos(tau * (double)(l * k1) / 512.0), -sin(tau * (double)(l * k1) / 512.0));
      for (int d = 0; d < 8; ++d) put(cos(tau * (double)((l & 7) * d) / 64.0), -sin(tau * (double)((l & 7) * d) / 64.0));
      for (int ee = 0; ee < 8; ++ee) { const int k = (l >> 3) + 8 * (l & 7) + 64 * ee; put(cos(tau * k / 1024.0), -sin(tau * k / 1024.0)); }
      for (int ee = 0; ee < 8; ++ee) { const int k = (l >> 3) + 8 * (l & 7) + 64 * ee; put(cos(tau * k / 2048.0), -sin(tau * k / 2048.0)); }
      // the backward's window: samples 4 m + 2 h + {0, 1} of m = (l >> 3) + 8 (l & 7) + 64 e
      for (int ee = 0; ee < 8; ++ee)
        for (int h = 0; h < 2; ++h) {
          const int m = (l >> 3) + 8 * (l & 7) + 64 * ee;
          put(window_host[4 * m + 2 * h], window_host[4 * m + 2 * h + 1]);
        }
    }
  }
  return IAS_OK;
}

// Framed STFT of audio [B,T] (center=True, reflect pad); window and twiddles come as the device copies of
// the ias_stft_build_tables block (VALU kernel) and the ias_stft_build_mtables block (matrix-core kernel, which
// takes its mel filterbank from that block too).  Per-bin value by value_mode (1 |X|, 2 |X|^2, 3 sqrt(max(|X|^2, eps))),
// optional mel projection given as packed filters (mel_start/count/woff [n_out], mel_w [mel_nnz]);
// n_out = n_mels, or n_fft/2+1 when mel_* are NULL.
//   out      [B,F,n_out] or NULL : the spectrogram (frames-major layout)
//   target   [B,F,n_out] or NULL : with loss_mode 1 (sum |v-t|) or 2 (MR-STFT sums)
//   partials [ias_stft_partials_count][3] doubles, required when loss_mode != 0
//   rowpeak  [B] or NULL : row peaks max |audio| (ias_voice_render's workspace): the spectrum of the row normalised as
//                          torchsynth's normalize_if_clipping would, without the normalised audio ever being written
extern "C" int ias_stft(const float* audio, const float* tables, const float* mtables, const float* segtab, const int* mel_start, const int* mel_count,
                        const int* mel_woff, const float* mel_w, int mel_nnz, float* out, const float* target,
                        double* partials, const float* rowpeak, int* ticket, int B, int T, int n_fft, int hop, int n_out,
                        int value_mode, int loss_mode, float eps, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!audio || !tables || B <= 0 || B > 65535 || n_out <= 0) return IAS_ERR_ARG;
  if (value_mode < 1 || value_mode > 3 || loss_mode < 0 || loss_mode > 2) return IAS_ERR_ARG;
  if (loss_mode != 0 && (!target || !partials)) return IAS_ERR_ARG;
  if (loss_mode == 0 && !out) return IAS_ERR_ARG;
  const bool mel = mel_start != nullptr;
  if (mel && (!mel_count || !mel_woff || !mel_w || mel_nnz <= 0)) return IAS_ERR_ARG;
  if (!mel && n_out != n_fft / 2 + 1) return IAS_ERR_ARG;
  const int F = ias_stft_num_frames(T, n_fft, hop);
  if (F < 0) return IAS_ERR_ARG;
  if (n_fft != 512 && n_fft != 1024 && n_fft != 2048) return IAS_ERR_UNSUPPORTED;
  if (ias_sm_enabled(n_fft, mtables != nullptr))
    return ias_sm_launch(audio, mtables, mel, out, target, partials, rowpeak, ticket, B, T, F, n_fft, hop, n_out,
                         value_mode, loss_mode, eps, stream);

  if (stft2_enabled(n_fft, mel, segtab != nullptr)) {
    if ((long long)B * F > 2000000000LL) return IAS_ERR_UNSUPPORTED;
    Spec2Args a2;
    a2.audio = audio; a2.tables = tables; a2.segtab = mel ? segtab : nullptr; a2.out = out; a2.target = target;
    a2.partials = partials; a2.rowpeak = rowpeak; a2.T = T; a2.F = F; a2.hop = hop; a2.n_out = n_out; a2.nframes = B * F;
    a2.magicF = (F == 1 ? 0xFFFFFFFFu /* 2^32 / 1 does not fit: q0 = fi - 1, which row_of's one-step correction fixes */ : (unsigned)(0x100000000ULL / (unsigned long long)F));
    a2.value_mode = value_mode; a2.loss_mode = loss_mode; a2.eps = eps;
#ifdef IAS_S2_STAMPS
    a2.stamps = g_s2_stamps;
#endif
    const int waves2 = n_fft == 1024 ? stft2_waves() : 8;
    const size_t lds2 = sizeof(cpx) * (waves2 * 64 * IAS_S2_ROW);       // the exchange scratch; the tables are static LDS objects
    const dim3 grid2(stft2_grid(a2.nframes, n_fft)), block2(64 * waves2);
#define IAS_STFT2_LAUNCHW(W, MEL, LOSS)                                                                            \
  do {                                                                                                             \
    (void)hipFuncSetAttribute((const void*)stft2_kernel<W, MEL, LOSS, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds2);                                                                          \
    hipLaunchKernelGGL((stft2_kernel<W, MEL, LOSS, 1>), grid2, block2, lds2, stream, a2);                          \
  } while (0)
#define IAS_STFT2_LAUNCH2K(LOSS)                                                                                   \
  do {                                                                                                             \
    (void)hipFuncSetAttribute((const void*)stft2_kernel<8, false, LOSS, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds2);                                                                          \
    hipLaunchKernelGGL((stft2_kernel<8, false, LOSS, 2>), grid2, block2, lds2, stream, a2);                        \
  } while (0)
#define IAS_STFT2_LAUNCH(MEL, LOSS)                                                                                \
  do {                                                                                                             \
    if (waves2 == 4) IAS_STFT2_LAUNCHW(4, MEL, LOSS);                                                              \
    else if (waves2 == 5) IAS_STFT2_LAUNCHW(5, MEL, LOSS);                                                         \
    else if (waves2 == 8) IAS_STFT2_LAUNCHW(8, MEL, LOSS);                                                         \
    else IAS_STFT2_LAUNCHW(10, MEL, LOSS);                                                                         \
  } while (0)
#define IAS_STFT2_LAUNCHH(LOSS)                                                                                    \
  do {                                                                                                             \
    hipLaunchKernelGGL((stft2h_kernel<8, LOSS>), grid2, block2, lds2, stream, a2);                                 \
  } while (0)
    if (n_fft == 512) { if (loss_mode == 0) IAS_STFT2_LAUNCHH(0); else if (loss_mode == 1) IAS_STFT2_LAUNCHH(1); else IAS_STFT2_LAUNCHH(2); }
    else if (n_fft == 2048) { if (loss_mode == 0) IAS_STFT2_LAUNCH2K(0); else if (loss_mode == 1) IAS_STFT2_LAUNCH2K(1); else IAS_STFT2_LAUNCH2K(2); }
    else if (mel) { if (loss_mode == 0) IAS_STFT2_LAUNCH(true, 0); else if (loss_mode == 1) IAS_STFT2_LAUNCH(true, 1); else IAS_STFT2_LAUNCH(true, 2); }
    else { if (loss_mode == 0) IAS_STFT2_LAUNCH(false, 0); else if (loss_mode == 1) IAS_STFT2_LAUNCH(false, 1); else IAS_STFT2_LAUNCH(false, 2); }
#undef IAS_STFT2_LAUNCH
#undef IAS_STFT2_LAUNCH2K
#undef IAS_STFT2_LAUNCHH
#undef IAS_STFT2_LAUNCHW
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }

  SpecArgs a;
  a.audio = audio; a.tables = tables;
  a.mel_start = mel_start; a.mel_count = mel_count; a.mel_woff = mel_woff; a.mel_w = mel_w;
  a.out = out; a.target = target; a.partials = partials; a.rowpeak = rowpeak;
  a.T = T; a.F = F; a.hop = hop; a.n_out = n_out; a.mel_nnz = mel ? mel_nnz : 0;
  a.value_mode = value_mode; a.loss_mode = loss_mode; a.eps = eps;

  a.groups = stft_groups(B, F, n_fft);
  const size_t lds = stft_lds_bytes(n_fft, a.mel_nnz, n_out);
  if (lds > 150 * 1024) return IAS_ERR_UNSUPPORTED;
  const dim3 grid(stft_grid_x(B, F, n_fft, hop), B), block(64 * stft_waves(n_fft));
#define IAS_STFT_LAUNCH(LOG2N, WAVES)                                                                              \
  do {                                                                                                             \
    if (lds > 64 * 1024)                                                                                           \
      (void)hipFuncSetAttribute(loss_mode == 2 ? (const void*)stft_kernel<LOG2N, WAVES, true>                     \
                                               : (const void*)stft_kernel<LOG2N, WAVES, false>,                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
    if (loss_mode == 2) hipLaunchKernelGGL((stft_kernel<LOG2N, WAVES, true>), grid, block, lds, stream, a);        \
    else hipLaunchKernelGGL((stft_kernel<LOG2N, WAVES, false>), grid, block, lds, stream, a);                      \
  } while (0)
  if (n_fft == 512) IAS_STFT_LAUNCH(9, 4);
  else if (n_fft == 1024) {
    if (stft_waves(n_fft) == 10) IAS_STFT_LAUNCH(10, 10); else if (stft_waves(n_fft) == 8) IAS_STFT_LAUNCH(10, 8); else IAS_STFT_LAUNCH(10, 4);
  }
  else { if (stft_waves(n_fft) == 12) IAS_STFT_LAUNCH(11, 12); else IAS_STFT_LAUNCH(11, 4); }
#undef IAS_STFT_LAUNCH
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// sums[3] (doubles) = column sums of partials [n][3], in a fixed order; if mean_out != NULL also
// mean_out[0] = (float)(sums[0] * scale).
extern "C" int ias_reduce_partials(const double* partials, long long n, double* sums, double scale, float* mean_out,
                                   void* stream_) {
  if (!partials || !sums || n <= 0) return IAS_ERR_ARG;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(RP_THREADS), 0, (hipStream_t)stream_, partials, n, sums, scale,
                     mean_out);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
