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
#include <cstdint>
#include <cstdlib>

#define PQ_THREADS 256
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const float* lds_cfloat_ptr;
typedef __attribute__((address_space(3))) const volatile f32x2* lds_cvpair_ptr;

// acc += xw * (HALF ? hp.y : hp.x) on both lanes; hp is a wave-uniform pair held in SGPRs.  v_pk_fma_f32 takes
// the SGPR pair directly and op_sel / op_sel_hi pick which half feeds each lane (checked on gfx950 with
// scripts/diag/pk_opsel.hip), so a tap costs no VGPR and no v_mov splat.
template <int HALF>
__device__ __forceinline__ void pk_fma_bcast(f32x2& acc, const f32x2 xw, const f32x2 hp) {
  if (HALF == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(xw), "s"(hp));
  else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(xw), "s"(hp));
}

template <int N, int K>
struct PqmfFast {
  static constexpr int R = 4;                          // frames per lane: two adjacent pairs, 128 frames apart
  static constexpr int FT = PQ_THREADS * R;            // frames per workgroup
  static constexpr int Q = (K + N - 1) / N;            // taps per polyphase branch
  static constexpr int U = 2;                          // steps per pipeline stage (U*N*N even: pair parity static)
  static constexpr int NST = Q / U;                    // full stages; stage NST holds the Q % U tail steps
  static constexpr int COLS = FT + Q - 1;              // polyphase positions a workgroup touches
  static constexpr int ROW = (COLS + 31) / 32 * 32 + 2;   // even: pairs stay 8-byte aligned in every row
  static constexpr int TABLE = (NST + 1) * U * N * N;  // floats of the transposed tap table, zero past K (the
                                                       // pipeline always loads one whole stage ahead)
};

// One pipeline stage = U polyphase steps: their frame pairs and taps in registers.  The frame pairs
// (x_p[c+q], x_p[c+q+1]) are 8-byte aligned in the even copy of the rows when q is even and in the odd copy
// (shifted by one position) when q is odd, so each is one ds_read_b64 at an immediate offset.
template <int N, int ROW, int U>
struct PqmfStage {
  static constexpr int NH = U * N * N / 2;
  f32x2 xa[U][N], xb[U][N], h[NH];

  // volatile reads: each pair stays its own ds_read_b64 (256 B/clk; merged into ds_read2_b64 / ds_read2st64_b64
  // the same bytes move at a quarter of that rate) and they are issued here, not where the FMAs want them
  __device__ __forceinline__ void load(lds_cfloat_ptr ra, const f32x2* __restrict__ taps) {
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int off = (s & 1) ? N * ROW + s - 1 : s;    // odd copy holds x_p[c+1] at c
#pragma unroll
      for (int p = 0; p < N; ++p) {
        xa[s][p] = *(lds_cvpair_ptr)(ra + p * ROW + off);
        xb[s][p] = *(lds_cvpair_ptr)(ra + p * ROW + off + 128);
      }
    }
#pragma unroll
    for (int i = 0; i < NH; ++i) h[i] = taps[i];
  }

  // Everything this stage loaded has arrived.  LDS and scalar-cache loads share one counter and scalar loads
  // return out of order, so a wait placed at the first FMA would also drain the NEXT stage's loads issued just
  // before it; using the registers here (no instruction) puts the compiler's wait ahead of those loads.
  __device__ __forceinline__ void arrived() const {
#pragma unroll
    for (int s = 0; s < U; ++s)
#pragma unroll
      for (int p = 0; p < N; ++p) asm volatile("" ::"v"(xa[s][p]), "v"(xb[s][p]));
#pragma unroll
    for (int i = 0; i < NH; ++i) asm volatile("" ::"s"(h[i]));
  }

  template <int STEPS>
  __device__ __forceinline__ void fma(f32x2 (&acc)[N][2]) const {
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int p = 0; p < N; ++p)
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const int t = (s * N + p) * N + k;
          if (t & 1) { pk_fma_bcast<1>(acc[k][0], xa[s][p], h[t >> 1]); pk_fma_bcast<1>(acc[k][1], xb[s][p], h[t >> 1]); }
          else       { pk_fma_bcast<0>(acc[k][0], xa[s][p], h[t >> 1]); pk_fma_bcast<0>(acc[k][1], xb[s][p], h[t >> 1]); }
        }
  }
};

// Polyphase form: with j = N*q + p, z_k[f] = sum_q sum_p H_k[N*q+p] * x_p[f+q], x_p[m] = x[N*m + p - pad].
// The workgroup de-interleaves its input span into the N polyphase rows in LDS (lane c loads the N samples of
// column c and writes one dword per row: coalesced, conflict-free, no index division), twice: rows [0, N) hold
// x_p[c] at c, rows [N, 2N) hold x_p[c+1] at c.  A lane owns the adjacent frames (2*lane, 2*lane+1) of its
// wave's 256-frame span and the same pair 128 frames on; every operand pair of v_pk_fma_f32 is then one aligned
// ds_read_b64 (256 B/clk, lanes 8 B apart: no conflicts, no swizzle, no register shuffling).  Taps come from the
// transposed table Pt[j*N + k] = H[k][j] (ias_pqmf_pack_taps) through the scalar cache into SGPR pairs
// (pk_fma_bcast), so the VALU stream is the FMAs plus a handful of loop instructions.
// Workgroups are persistent: each walks tiles t = blockIdx.x, + gridDim.x, ... of the (batch row, 1024-frame tile)
// list and loads the next tile's samples into registers before it computes the current one, so the ~3000-cycle
// global-load latency overlaps the FMAs instead of preceding them.
// Row scale of torchsynth's normalize_if_clipping folded into the analysis (the filterbank is linear): rowpeak[b] =
// max |x[b, :]| as ias_voice_render leaves it in its workspace; the output is that of the row divided by its peak when
// the peak exceeds 1.  Saves the separate in-place pass over the audio (8 B per sample of HBM traffic for clipping rows).
__device__ __forceinline__ float pqmf_row_scale(const float* __restrict__ rowpeak, int b) {
  if (rowpeak == nullptr) return 1.0f;
  const float pk = rowpeak[b];
  return pk > 1.0f ? 1.0f / pk : 1.0f;
}
// The fused epilogue of every analysis kernel, spelled out so that all of them round alike whatever the compiler's
// contraction choices: band value = acc * rsc, normalised value = (acc * rsc - m) / sd with the product and the
// subtraction in one fma (rsc == 1: exactly (acc - m) / sd).
__device__ __forceinline__ float pqmf_finish(float acc, float rsc, bool norm, float m, float sd) {
  return norm ? __fdiv_rn(fmaf(acc, rsc, -m), sd) : __fmul_rn(acc, rsc);
}

template <int N, int K>
__global__ __launch_bounds__(PQ_THREADS) void pqmf_analysis_fast_kernel(
    const float* __restrict__ x, const float* __restrict__ Pt, float* __restrict__ z,
    const float* __restrict__ mean, const float* __restrict__ stdv, const float* __restrict__ rowpeak, int T, int L,
    int pad, int tiles_x, int ntiles) {
  using C = PqmfFast<N, K>;
  constexpr int FT = C::FT, Q = C::Q, U = C::U, COLS = C::COLS, ROW = C::ROW, NST = C::NST;
  constexpr int NH = U * N * N / 2;
  constexpr int NIT = (COLS + PQ_THREADS - 1) / PQ_THREADS;   // columns staged per thread
  static_assert((U * N * N) % 2 == 0 && U % 2 == 0 && NST % 2 == 0, "tap pair / row copy parity must not depend on the loop counter; stages are consumed in pairs");
  __shared__ __attribute__((aligned(16))) float s_xp[2 * N * ROW];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  float e[NIT][N];

  // Column c of tile t holds samples g .. g+N-1, g = N*(f_tile + c) - pad.  The loads are unconditional and
  // branch-free (a join would make the compiler wait for them on the spot): the address is clamped into the row
  // and the few columns that stick out of it (first / last tile of a row) are sorted out when they are consumed.
  auto load_tile = [&](int t) {
    const int b = t / tiles_x;
    const int g_tile = (t - b * tiles_x) * FT * N - pad;
    const float* xr = x + (size_t)b * T;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = min(tid + it * PQ_THREADS, COLS - 1);
      const int gc = min(max(g_tile + c * N, 0), T - N);
#pragma unroll
      for (int p = 0; p < N; ++p) e[it][p] = xr[gc + p];
    }
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  load_tile(t);
  for (; t < ntiles; t += gridDim.x) {
    const int b = t / tiles_x;
    const int f_tile = (t - b * tiles_x) * FT;
    const int g_tile = f_tile * N - pad;
    if (g_tile < 0 || g_tile + COLS * N > T) {
      // edge tile: move each sample to its place (the clamped load is shifted by g - gc) and zero what lies
      // outside the row
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = min(tid + it * PQ_THREADS, COLS - 1);
        const int g = g_tile + c * N;
        const int d = g - min(max(g, 0), T - N);
        float v[N];
#pragma unroll
        for (int p = 0; p < N; ++p) {
          float sel = 0.0f;
#pragma unroll
          for (int s2 = 0; s2 < N; ++s2) sel = (p + d == s2) ? e[it][s2] : sel;
          v[p] = (g + p >= 0 && g + p < T) ? sel : 0.0f;
        }
#pragma unroll
        for (int p = 0; p < N; ++p) e[it][p] = v[p];
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * PQ_THREADS;
      if (c < COLS) {
#pragma unroll
        for (int p = 0; p < N; ++p) s_xp[p * ROW + c] = e[it][p];
        if (c > 0) {
#pragma unroll
          for (int p = 0; p < N; ++p) s_xp[(N + p) * ROW + c - 1] = e[it][p];
        }
      }
    }
    __syncthreads();
    load_tile(min(t + (int)gridDim.x, ntiles - 1));   // the last round reloads a tile nobody will use

    lds_cfloat_ptr ra = (lds_cfloat_ptr)s_xp + wave * 256 + 2 * lane;
    const f32x2* taps = reinterpret_cast<const f32x2*>(Pt);
    f32x2 acc[N][2];
#pragma unroll
    for (int k = 0; k < N; ++k) { acc[k][0] = (f32x2){0.0f, 0.0f}; acc[k][1] = (f32x2){0.0f, 0.0f}; }

    PqmfStage<N, ROW, U> sa;
    if constexpr (4 * NH <= 40) {
      // two-stage software pipeline over the Q steps: stage i+1 is in flight while stage i feeds the FMAs
      PqmfStage<N, ROW, U> sb;
      sa.load(ra, taps);
#pragma unroll 1
      for (int i = 0; i < NST; i += 2) {
        sa.arrived();
        sb.load(ra + U, taps + NH);
        sa.template fma<U>(acc);
        ra += 2 * U;
        taps += 2 * NH;
        sb.arrived();
        sa.load(ra, taps);
        sb.template fma<U>(acc);
      }
      sa.arrived();
    } else {
      // N = 4: the taps of two stages do not fit the SGPR file next to everything else; one stage at a time
#pragma unroll 1
      for (int i = 0; i < NST; ++i) {
        sa.load(ra, taps);
        sa.template fma<U>(acc);
        ra += U;
        taps += NH;
      }
      if (Q % U) sa.load(ra, taps);
    }
    sa.template fma<Q % U>(acc);   // tail steps (the table is zero past K)

    const int f0 = f_tile + wave * 256 + 2 * lane;
    const float rsc = pqmf_row_scale(rowpeak, b);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      f32x2 o[2];
      {
        const bool nrm = mean != nullptr;
        const float m = nrm ? mean[k] : 0.0f, sd = nrm ? stdv[k] : 1.0f;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          o[r].x = pqmf_finish(acc[k][r].x, rsc, nrm, m, sd);
          o[r].y = pqmf_finish(acc[k][r].y, rsc, nrm, m, sd);
        }
      }
      float* zr = z + ((size_t)b * N + k) * L;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int f = f0 + 128 * r;
        if ((L & 1) == 0 && f + 1 < L) {
          *reinterpret_cast<f32x2*>(zr + f) = o[r];
        } else {
          if (f < L) zr[f] = o[r].x;
          if (f + 1 < L) zr[f + 1] = o[r].y;
        }
      }
    }
    __syncthreads();   // every wave is done reading the rows before the next tile overwrites them
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Analysis on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, bit-for-bit the tap-ordered fmaf chain the
// VALU kernels compute, on a pipe nothing else in the step uses).
//
// The filterbank is a GEMM z[f][k] = sum_j x[N f + j - pad] H[k][j] with only N output columns, far too few for a
// 16-wide tile when N = 3.  S consecutive frames are therefore folded into one "super-frame" row: column (k, s)
// of row F is band k of frame S F + s, its taps shifted by N s samples,
//     Hs[(k, s)][i] = H[k][i - N s]   (0 <= i - N s < K, zero elsewhere),   i < KP = K + N (S - 1),
// so that every row reads ONE contiguous run x[N S F - pad + i] and each A value feeds N S columns.  N = 3: S = 5,
// 15 of 16 columns live, KP = 75 -> 19 k-steps of 4 (78 % of the issued FMAs are real ones); N = 4: S = 4, 16 columns,
// 19 k-steps; N = 64: S = 1, four column tiles, 16 k-steps.  The zero taps add exact zeros (x finite), so the sum of
// each column is the same chain fma(x_62, h_62, ... fma(x_0, h_0, 0)) as the polyphase kernel's.
//
// Operands (guide section 3, FP32-input MFMA): lane l = (r = l & 15, kq = l >> 4) gives A[row r][k = kq] =
// x_lds[RS (16 rt + r) + 4 kk + kq] (one ds_read_b32 per k-step and row tile, immediate offsets) and B[k = kq][col r] =
// Hs[col][4 kk + kq] (KQ x CT registers loaded once per workgroup).  D: col = l & 15, row = 4 (l >> 4) + reg.
// A wave keeps RT row tiles in flight (independent accumulators cover the 40-cycle dependent-MFMA latency).  Results
// go through a wave-private LDS block so that the stores are 16-byte runs along the frame axis.
// Staging: 16-byte global loads from the 16-byte aligned address at or below the tile's first sample, one tile ahead
// in registers; the residual shift (0..3 samples) is added to the A read address.
typedef float pq_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pq_u32x4 __attribute__((ext_vector_type(4)));

template <int N, int K, int S, int RT, int PSH>
struct PqmfMfma {
  static_assert(RT % 2 == 0, "row tiles are processed in pairs");
  static constexpr int NC = N * S;                      // live columns
  static constexpr int CT = (NC + 15) / 16;             // column tiles
  static constexpr int KP = K + N * (S - 1);            // taps of a super-frame row
  static constexpr int KQ = (KP + 3) / 4;               // k-steps
  static constexpr int RS = N * S;                      // samples between super-frames
  static constexpr int WF = RT * 16 * S;                // frames per wave
  static constexpr int FT = (PQ_THREADS / 64) * WF;     // frames per workgroup tile
  static constexpr int SPAN = (N * FT + 4 * KQ - N + 3 + 3) / 4 * 4;   // staged samples (every word an A read can touch)
  static constexpr int NV4 = SPAN / 4;
  static constexpr int NIT = (NV4 + PQ_THREADS - 1) / PQ_THREADS;
  // LDS position of staged sample j: 4 pad words per 2^PSH samples when the row stride is a multiple of 16 words
  // (N S = 16: rows 4 apart share a bank; N = 64: all 16 rows do)
  __host__ __device__ static constexpr int pos(int j) { return PSH ? j + 4 * (j >> PSH) : j; }
  static constexpr int IN_WORDS = pos(SPAN) + 4;
  static constexpr int OROW = WF + 4;                   // wave-private output block [N][OROW]
  static constexpr int OUT_WORDS = (PQ_THREADS / 64) * N * OROW;
  static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(IN_WORDS + OUT_WORDS);
};

template <int N, int K, int S, int RT, int PSH>
__global__ __launch_bounds__(PQ_THREADS) void pqmf_analysis_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ H, float* __restrict__ z, const float* __restrict__ mean,
    const float* __restrict__ stdv, const float* __restrict__ rowpeak, int T, int L, int pad, int tiles_x, int ntiles,
    int zvec /* z rows are 16-byte aligned (z aligned and L % 4 == 0) */) {
  using C = PqmfMfma<N, K, S, RT, PSH>;
  constexpr int CT = C::CT, KQ = C::KQ, RS = C::RS, WF = C::WF, FT = C::FT, NV4 = C::NV4, NIT = C::NIT, OROW = C::OROW;
  extern __shared__ __attribute__((aligned(16))) float s_pm[];
  float* s_in = s_pm;
  float* s_out = s_pm + C::IN_WORDS;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;

  float bf[CT][KQ];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = 16 * ct + r, band = col / S, j0 = kq - N * (col - band * S);
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
      const int j = j0 + 4 * kk;
      bf[ct][kk] = (col < C::NC && j >= 0 && j < K) ? H[band * K + j] : 0.0f;
    }
  }

  // band constants of the fused normalisation: loaded once (a vector load in the tile loop would make its wait drain
  // the next tile's prefetch, which is queued ahead of it)
  float nm[CT], nsd[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = 16 * ct + r, band = min(col, C::NC - 1) / S;
    nm[ct] = mean != nullptr ? mean[band] : 0.0f;
    nsd[ct] = mean != nullptr ? stdv[band] : 1.0f;
  }

  pq_f32x4 e[NIT];
  auto load_tile = [&](int t) {
    const int b = t / tiles_x;
    const int a0 = ((t - b * tiles_x) * FT * N - pad) & ~3;
    const float* xr = x + (size_t)b * T;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int g = a0 + 4 * min(tid + it * PQ_THREADS, NV4 - 1);
      e[it] = *reinterpret_cast<const pq_f32x4*>(xr + min(max(g, 0), T - 4));
    }
  };

  int t = blockIdx.x;
  if (t >= ntiles) return;
  load_tile(t);
  for (; t < ntiles; t += gridDim.x) {
    const int b = t / tiles_x;
    const int f_tile = (t - b * tiles_x) * FT;
    const int g_tile = f_tile * N - pad;
    const int a0 = g_tile & ~3, sh = g_tile - a0;
    // groups of four are wholly inside or wholly outside the row (T % 4 == 0): zero padding of the convolution
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int j4 = tid + it * PQ_THREADS;
      const int g = a0 + 4 * j4;
      if (g < 0 || g >= T) e[it] = (pq_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
      if (j4 < NV4) *reinterpret_cast<pq_f32x4*>(s_in + C::pos(4 * j4)) = e[it];
    }
    __syncthreads();
    load_tile(min(t + (int)gridDim.x, ntiles - 1));   // the last round reloads a tile nobody will use

    pq_f32x4 acc[RT][CT];
    int ab[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      ab[rt] = sh + RS * (16 * (wave * RT + rt) + r) + kq;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (pq_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    // Row tiles are taken in pairs (two independent accumulation chains).  The A values of pair p + 1 are read, one
    // k-step per k-step, while pair p feeds the matrix core: every ds_read is a whole pair (2 KQ MFMAs, > 1000 cycles)
    // ahead of its use.  Left to itself the compiler places each read two or three MFMAs ahead of its use, less than
    // the LDS latency, and the MFMA stream of a wave runs at half rate (measured: 4800 instead of 2432 cycles a tile).
    float av[2][2][KQ];
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
      av[0][0][kk] = s_in[C::pos(ab[0] + 4 * kk)];
      av[0][1][kk] = s_in[C::pos(ab[1] + 4 * kk)];
    }
#pragma unroll
    for (int p = 0; p < RT / 2; ++p) {
#pragma unroll
      for (int kk = 0; kk < KQ; ++kk) {
        if (p + 1 < RT / 2) {
          av[(p + 1) & 1][0][kk] = s_in[C::pos(ab[2 * p + 2] + 4 * kk)];
          av[(p + 1) & 1][1][kk] = s_in[C::pos(ab[2 * p + 3] + 4 * kk)];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            acc[2 * p + h][ct] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(av[p & 1][h][kk], bf[ct][kk], acc[2 * p + h][ct], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float* so = s_out + wave * (N * OROW);
    const float rsc = pqmf_row_scale(rowpeak, b);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int col = 16 * ct + r;
      if (col < C::NC) {
        const int band = col / S, s = col - band * S;
        const float m = nm[ct], sd = nsd[ct];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float o = pqmf_finish(acc[rt][ct][q], rsc, mean != nullptr, m, sd);
            so[band * OROW + S * (16 * rt + 4 * kq + q) + s] = o;
          }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int W4 = WF / 4;
    const int fw = f_tile + wave * WF;
    for (int i = lane; i < N * W4; i += 64) {
      const int band = i / W4, f = fw + 4 * (i - band * W4);
      const pq_f32x4 v = *reinterpret_cast<const pq_f32x4*>(so + band * OROW + (f - fw));
      float* zr = z + ((size_t)b * N + band) * L;
      if (zvec && f + 3 < L) {
        *reinterpret_cast<pq_f32x4*>(zr + f) = v;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (f + q < L) zr[f + q] = v[q];
      }
    }
    __syncthreads();   // every wave is done with the staged samples (and its output block) before the next tile
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-pipelined form of the MFMA analysis for the folded cases (N S <= 16, one column tile: N = 3, 4).
//
// The workgroup-tiled kernel above alternates phases -- stage, barrier, MFMA, epilogue, store, barrier -- and all waves
// of a SIMD end up in the same phase (they start together and share the matrix core fairly), so the matrix core idles
// while they stage or store: 46 % busy at N = 3 (SQ_VALU_MFMA_BUSY_CYCLES), stamps: 3500 cycles of MFMA issue against
// 3000-6000 of epilogue / store / staging per tile.  Here every wave is its own pipeline and there is no workgroup
// barrier.  A wave tile is four row tiles (64 S frames) taken as two pairs; while pair p feeds the matrix core
// (2 KQ MFMAs, about 1200 cycles) the same wave's instruction stream carries, between the MFMAs,
//   * the A reads of the NEXT pair (a full pair ahead of their use),
//   * the epilogue of the PREVIOUS pair: accumulators -> wave-private LDS block -> 16-byte global stores,
//   * phase 0: staging of the next tile (registers -> the other LDS input buffer),
//     phase 1: the global loads of the tile after that (in flight for a whole tile).
// LDS traffic of one wave executes in order, so the write -> read hand-overs inside the wave need a compiler fence
// but no s_waitcnt; the only waits are on data the MFMAs or stores consume.
// s_waitcnt vmcnt(allow) that the uses of v cannot be scheduled above (allow is a constant once the caller is unrolled)
__device__ __forceinline__ void pq_vm_wait(pq_f32x4& v, int allow) {
  switch (allow) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" : "+v"(v)::"memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" : "+v"(v)::"memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" : "+v"(v)::"memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" : "+v"(v)::"memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" : "+v"(v)::"memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" : "+v"(v)::"memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" : "+v"(v)::"memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" : "+v"(v)::"memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); break;
  }
}

__device__ __forceinline__ void pq_vm_wait1(float& v, int allow) {
  switch (allow) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" : "+v"(v)::"memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" : "+v"(v)::"memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" : "+v"(v)::"memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" : "+v"(v)::"memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory"); break;
  }
}

template <int N, int K, int S, int PSH>
struct PqmfPipe {
  static constexpr int NC = N * S;
  static_assert(NC <= 16, "one column tile");
  static constexpr int KP = K + N * (S - 1), KQ = (KP + 3) / 4;
  static constexpr int RS = N * S;
  static constexpr int PF = 32 * S;                     // frames of a pair of row tiles
  static constexpr int WF = 2 * PF;                     // frames of a wave tile
  static constexpr int SPAN = (RS * 63 + 4 * KQ + 3 + 3) / 4 * 4;   // staged samples: shift + 64 rows + KQ k-steps
  static constexpr int NV4 = SPAN / 4, NIT = (NV4 + 63) / 64;
  __host__ __device__ static constexpr int pos(int j) { return PSH ? j + 4 * (j >> PSH) : j; }
  static constexpr int INW = (pos(SPAN) + 4 + 3) / 4 * 4;
  static constexpr int OROW = PF + 4;
  static constexpr int WAVEW = (2 * INW + N * OROW + 20 * S + 3) / 4 * 4;   // two input buffers, the output block and
                                                                         // the dump area of one wave
  static constexpr int W4 = PF / 4;                     // 16-byte pieces per band and pair
  static constexpr int NIO = (N * W4 + 63) / 64;
  static constexpr int NEV = 8;                         // accumulator values per lane and pair
  static constexpr size_t LDS_BYTES = sizeof(float) * (size_t)(PQ_THREADS / 64) * WAVEW;
  static_assert(10 + NIT <= KQ - 4 && NEV + 1 <= 9 && NIT - 1 + NIO + 1 <= 8 && NIO <= 4, "slot plan of the phase loop");
};

template <int N, int K, int S, int PSH, bool NORM>
__global__ __launch_bounds__(PQ_THREADS) void pqmf_analysis_pipe_kernel(
    const float* __restrict__ x, const float* __restrict__ H, float* __restrict__ z, const float* __restrict__ mean,
    const float* __restrict__ stdv, const float* __restrict__ rowpeak, int T, int L, int pad, int tiles_x, int ntiles,
    int zvec /* always 1 here: z rows 16-byte aligned, L % 4 == 0 (the host sends other shapes to the tiled kernel) */) {
  using C = PqmfPipe<N, K, S, PSH>;
  constexpr int KQ = C::KQ, RS = C::RS, PF = C::PF, WF = C::WF, NV4 = C::NV4, NIT = C::NIT, OROW = C::OROW;
  constexpr int NIO = C::NIO, W4 = C::W4, INW = C::INW;
  extern __shared__ __attribute__((aligned(16))) float s_pm[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int r = lane & 15, kq = lane >> 4;
  float* const s_in = s_pm + wave * C::WAVEW;           // [2][INW]
  float* const so = s_in + 2 * INW;                     // [N][OROW]

  const int band = min(r, C::NC - 1) / S, sfr = min(r, C::NC - 1) - band * S;
  const bool live = r < C::NC;
  float bf[KQ];
#pragma unroll
  for (int kk = 0; kk < KQ; ++kk) {
    const int j = 4 * kk + kq - N * sfr;
    bf[kk] = (live && j >= 0 && j < K) ? H[band * K + j] : 0.0f;
  }
  float nm = 0.0f, nsd = 1.0f;
  if (NORM) { nm = mean[band]; nsd = stdv[band]; }
  // where accumulator value (h, q) goes: ew + S (16 h + q); the lanes of the unused columns write a dump area instead
  // (branch-free: the epilogue stays in the MFMA stream's basic block)
  const int ew = live ? band * OROW + S * (4 * kq) + sfr : N * OROW;
  int io_lds[NIO], io_glb[NIO], io_f[NIO];   // the 16-byte pieces of the output block this lane stores
#pragma unroll
  for (int i = 0; i < NIO; ++i) {
    const int idx = lane + 64 * i, bnd = idx / W4, f4 = idx - bnd * W4;
    io_lds[i] = idx < N * W4 ? bnd * OROW + 4 * f4 : -1;
    io_glb[i] = bnd * L + 4 * f4;
    io_f[i] = 4 * f4;
  }

  // Tile bookkeeping is wave-uniform and kept in SGPRs: (row b, tile tx of the row) advance by the grid stride
  // without a division per tile.
  const int stride = gridDim.x * (PQ_THREADS / 64);
  const int stride_b = stride / tiles_x, stride_x = stride - stride_b * tiles_x;
  int t = blockIdx.x * (PQ_THREADS / 64) + wave;
  if (t >= ntiles) return;                              // no workgroup barrier anywhere: waves leave on their own
  struct Tile { int b, tx; };
  auto advance = [&](Tile c) {
    c.b += stride_b; c.tx += stride_x;
    if (c.tx >= tiles_x) { c.tx -= tiles_x; ++c.b; }
    return c;
  };
  auto tile_g0 = [&](Tile c) { return c.tx * (WF * N) - pad; };   // first sample of the tile (may be negative)
  // z as a raw buffer (byte offsets, range-checked by the hardware); the host keeps it below 2^31 - 256 bytes
  const __amdgpu_buffer_rsrc_t zres =
      __builtin_amdgcn_make_buffer_rsrc(z, 0, (int)((long long)(ntiles / tiles_x) * N * L * 4), 0x00020000);

  // Tile loads: buffer loads over the ROW (base x + b T, T * 4 bytes), so the zero padding of the convolution is the
  // hardware's range check (an offset below 0 or past the row returns 0) and there is no clamping or masking on the
  // VALU.  They are inline asm, outside the compiler's vmcnt bookkeeping, and waited for by hand (pq_vm_wait): left to
  // the compiler, the staging wait in the loop becomes vmcnt(3..0) -- at the loop header it merges the entry path with
  // the back edge and assumes nothing younger than the loads is in the queue -- and the wave waits for the stores of
  // phase 1 (issued after the loads) to COMPLETE before it stages.
  // Queue when piece `it` is staged (phase 0, slots 10..): L0 .. L(NIT-1) (phase 1 slot 0), S x NIO (phase 1), PK (the row
  // peak of this tile, phase 0 slot 0), nothing younger: the wait is vmcnt(NIT - 1 - it + NIO + 1).  The prologue issues NIO
  // dropped stores after its loads so that the first tile sees the same queue.  The peak is waited for in slot 16 behind
  // the NIO stores of phase 0: vmcnt(NIO).  Nothing in flight is carried in a register across the loop's back edge
  // except e[], which the loop only ever touches through the asm statements (no copies).
  pq_f32x4 e[NIT];
  float pk = 1.0f;
  const int lane16 = lane * 16;
  auto load_tile = [&](Tile c) {
    const float* xr = x + (size_t)c.b * T;
    const pq_u32x4 rs = {(unsigned)(uintptr_t)xr, (unsigned)((uintptr_t)xr >> 32), (unsigned)T * 4u, 0x00020000u};
    const int voff = lane16 + 4 * (tile_g0(c) & ~3);
#pragma unroll
    for (int it = 0; it < NIT; ++it)
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3"
                   : "=v"(e[it]) : "v"(voff), "s"(rs), "n"(1024 * it) : "memory");
  };
  auto stage_piece = [&](int it, float* buf, int allow) {
    pq_vm_wait(e[it], allow);
    const int j4 = lane + it * 64;
    if (NV4 % 64 == 0 || j4 < NV4) *reinterpret_cast<pq_f32x4*>(buf + C::pos(4 * j4)) = e[it];
  };

  // prologue: first tile staged, second tile in registers, A values of the first pair read
  Tile cur;
  cur.b = t / tiles_x;
  cur.tx = t - cur.b * tiles_x;
  load_tile(cur);
#pragma unroll
  for (int it = 0; it < NIT; ++it) stage_piece(it, s_in, 0);
  Tile nxt = t + stride < ntiles ? advance(cur) : cur;
  load_tile(nxt);
  // NIO stores the hardware drops (offset past the end): the queue shape stage_piece's wait counts on (see load_tile)
#pragma unroll
  for (int i = 0; i < NIO; ++i)
    __builtin_amdgcn_raw_buffer_store_b128((pq_u32x4){0u, 0u, 0u, 0u}, zres, 0x7ffffff0 - 16 * i, 0, 0);   // (distinct:
                                                                      // identical stores would be merged into one)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float av[2][2][KQ];
  const int abl = RS * r + kq;                          // lane part of an A read index
  {
    const int ab = (tile_g0(cur) & 3) + abl;
#pragma unroll
    for (int kk = 0; kk < KQ; ++kk) {
      av[0][0][kk] = s_in[C::pos(ab + 4 * kk)];
      av[0][1][kk] = s_in[C::pos(ab + RS * 16 + 4 * kk)];
    }
  }

  pq_f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (pq_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
  int cbuf = 0;
  // the pair whose epilogue phase 0 carries: pair 1 of the previous tile (none yet: pL = 0 masks its stores)
  int pb = 0, pf = 0, pL = 0;
  float prsc = 1.0f;

  for (; t < ntiles; t += stride) {
    const int b = cur.b;
    const int f_tile = cur.tx * WF;
    const int sh = tile_g0(cur) & 3, shn = tile_g0(nxt) & 3;
    const Tile nx2 = t + 2 * stride < ntiles ? advance(nxt) : nxt;
    float rsc = 1.0f;                                   // row scale of this tile: known from slot 16 of phase 0 on
    float* const bufc = s_in + cbuf * INW;
    float* const bufn = s_in + (cbuf ^ 1) * INW;

#pragma unroll
    for (int P = 0; P < 2; ++P) {
      // A values to read in this phase: pair 1 of this tile (phase 0) / pair 0 of the next tile (phase 1)
      const float* abuf = P == 0 ? bufc : bufn;
      const int ab = (P == 0 ? sh + RS * 32 : shn) + abl;
      // epilogue carried by this phase: pair 1 of the previous tile (phase 0) / pair 0 of this tile (phase 1)
      const int eb = P == 0 ? pb : b, ef = P == 0 ? pf : f_tile, eL = P == 0 ? pL : L;
      const float ersc = P == 0 ? prsc : rsc;
      pq_f32x4 ov[NIO];
#pragma unroll
      for (int kk = 0; kk < KQ; ++kk) {
        av[P ^ 1][0][kk] = abuf[C::pos(ab + 4 * kk)];
        av[P ^ 1][1][kk] = abuf[C::pos(ab + RS * 16 + 4 * kk)];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const pq_f32x4 cin = kk == 0 ? (pq_f32x4){0.0f, 0.0f, 0.0f, 0.0f} : acc[2 * P + h];
          acc[2 * P + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[P][h][kk], bf[kk], cin, 0, 0, 0);
        }
        if (kk >= 1 && kk <= 8) {                         // one accumulator value of the other pair -> LDS block
          const int j = kk - 1, h = j >> 2, q = j & 3;
          const float o = pqmf_finish(acc[2 * (P ^ 1) + h][q], ersc, NORM, nm, nsd);
          so[ew + S * (16 * h + q)] = o;
        }
        if (kk == 9) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int i = 0; i < NIO; ++i)
            ov[i] = *reinterpret_cast<const pq_f32x4*>(so + max(io_lds[i], 0));
        }
        // staging ahead of the stores: its vmcnt wait then covers the loads (issued a phase ago) and nothing younger
        if (P == 0 && kk == 0) {   // (always issued, from a dummy address without row peaks: the queue shape is fixed)
          const float* pp = rowpeak != nullptr ? rowpeak + b : x;
          asm volatile("global_load_dword %0, %1, %2" : "=v"(pk) : "v"(0), "s"(pp) : "memory");
        }
        if (P == 0 && kk >= 10 && kk < 10 + NIT) stage_piece(kk - 10, bufn, NIT - 1 - (kk - 10) + NIO + 1);
        if (kk == KQ - 4) {
          // Buffer stores: a piece outside the row (L % 4 == 0: wholly inside or wholly outside) gets an offset past
          // the end of the buffer and the hardware drops it.  No exec-masked branch: with one, the compiler cannot count
          // the stores in vmcnt and the staging wait of the next phase falls back to waiting for them to COMPLETE.
          const int zoff = (eb * N * L + ef) * 4;
#pragma unroll
          for (int i = 0; i < NIO; ++i) {
            const bool ok = io_lds[i] >= 0 && ef + io_f[i] < eL;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pq_u32x4, ov[i]), zres,
                                                   ok ? zoff + 4 * io_glb[i] : 0x7ffffff0, 0, 0);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the block is free for the next pair's values
          __builtin_amdgcn_wave_barrier();
        }
        if (P == 0 && kk == KQ - 3) {
          pq_vm_wait1(pk, NIO);
          rsc = (rowpeak != nullptr && pk > 1.0f) ? 1.0f / pk : 1.0f;
        }
        if (P == 1 && kk == 0) load_tile(nx2);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (P == 0) {   // the staged tile is visible to the A reads of phase 1
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    pb = b; pf = f_tile + PF; pL = L; prsc = rsc;
    cbuf ^= 1;
    cur = nxt; nxt = nx2;
  }

  // The loads the last phase 1 issued are still in flight and nobody consumes them.  They must have landed before the
  // compiler may reuse their registers (it does not know they are pending: a late arrival would overwrite whatever
  // lives there by then, e.g. a store address of the drain below).
#pragma unroll
  for (int it = 0; it < NIT; ++it) pq_vm_wait(e[it], 0);
  // drain: pair 1 of the last tile
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int h = j >> 2, q = j & 3;
    const float o = pqmf_finish(acc[2 + h][q], prsc, NORM, nm, nsd);
    so[ew + S * (16 * h + q)] = o;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float* zt = z + (size_t)pb * N * L + pf;
#pragma unroll
  for (int i = 0; i < NIO; ++i) {
    if (io_lds[i] < 0) continue;
    const pq_f32x4 v = *reinterpret_cast<const pq_f32x4*>(so + io_lds[i]);
    const int f = pf + io_f[i];
    if (f < pL) *reinterpret_cast<pq_f32x4*>(zt + io_glb[i]) = v;
  }
}

// Wide analysis (any N <= 64, any odd K <= 255): a small GEMM z[k][f] = sum_j H[k][j] * x[N*f + j - pad].
// The workgroup stages its input span in LDS de-interleaved by N (row = sample index mod N, one frame per column,
// odd row stride: staging writes and compute reads are conflict-free for every N, including N = 64 where frames are
// 64 samples apart).  A lane owns one frame and HB = NPAD/2 bands (the two halves of the workgroup's waves take
// the two halves of the bands): per tap one ds_read_b32 feeds HB FMAs whose taps are SGPR operands, loaded from the
// zero-padded transposed table Pt[j][NPAD] through the scalar cache.
#define PQW_FRAMES 128
template <int NPAD>
__global__ __launch_bounds__(PQ_THREADS) void pqmf_analysis_wide_kernel(
    const float* __restrict__ x, const float* __restrict__ Pt, float* __restrict__ z,
    const float* __restrict__ mean, const float* __restrict__ stdv, const float* __restrict__ rowpeak, int T, int L,
    int N, int K, int pad, int rs /* LDS row stride (odd) */) {
  constexpr int HB = NPAD / 2;
  extern __shared__ __attribute__((aligned(16))) float s_rows[];   // [N][rs]
  const int tid = threadIdx.x, b = blockIdx.y;
  const int f0 = blockIdx.x * PQW_FRAMES;
  const float* xr = x + (size_t)b * T;
  const long long g0 = (long long)f0 * N - pad;              // sample index of tile element 0
  const int span = N * (PQW_FRAMES - 1) + K;                // tile elements the frames touch
  const bool pow2 = (N & (N - 1)) == 0;
  const int sh = 31 - __builtin_clz(N);
  for (int i = tid; i < span; i += PQ_THREADS) {
    const long long g = g0 + i;
    const float v = (g >= 0 && g < T) ? xr[g] : 0.0f;
    const int col = pow2 ? (i >> sh) : (i / N);
    const int row = i - col * N;
    s_rows[row * rs + col] = v;
  }
  __syncthreads();

  const int fl = tid & (PQW_FRAMES - 1);                    // frame within the tile
  const int half = __builtin_amdgcn_readfirstlane(tid >> 7);     // wave-uniform band half
  const float* taps = Pt + half * HB;
  float acc[HB];
#pragma unroll
  for (int k = 0; k < HB; ++k) acc[k] = 0.0f;
  int row = 0, colq = 0;                                    // tap j sits in row j mod N, column fl + j div N
  for (int j = 0; j < K; ++j) {
    const float xv = s_rows[row * rs + colq + fl];
    const float* tj = taps + (size_t)j * NPAD;
#pragma unroll
    for (int k = 0; k < HB; ++k) acc[k] = fmaf(xv, tj[k], acc[k]);
    if (++row == N) { row = 0; ++colq; }
  }
  const int f = f0 + fl;
  if (f >= L) return;
  const float rsc = pqmf_row_scale(rowpeak, b);
#pragma unroll
  for (int k = 0; k < HB; ++k) {
    const int band = half * HB + k;
    if (band < N) {
      const bool nrm = mean != nullptr;
      const float o = pqmf_finish(acc[k], rsc, nrm, nrm ? mean[band] : 0.0f, nrm ? stdv[band] : 1.0f);
      z[((size_t)b * N + band) * L + f] = o;
    }
  }
}

// Pt[j*NPAD + k] = H[k][j] for k < N, zero for the padded bands
__global__ void pqmf_pack_wide_kernel(const float* __restrict__ H, float* __restrict__ Pt, int N, int K, int npad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * npad) return;
  const int j = i / npad, k = i - j * npad;
  Pt[i] = k < N ? H[k * K + j] : 0.0f;
}

static int pqmf_wide_npad(int N, int K) {
  if (N < 1 || N > 64 || K < 1 || K > 255) return 0;
  return N <= 8 ? 8 : (N <= 16 ? 16 : (N <= 32 ? 32 : 64));
}

// Generic analysis (any N, K): one lane per output element.
__global__ __launch_bounds__(PQ_THREADS) void pqmf_analysis_generic_kernel(
    const float* __restrict__ x, const float* __restrict__ H, float* __restrict__ z,
    const float* __restrict__ mean, const float* __restrict__ stdv, const float* __restrict__ rowpeak, int T, int L,
    int N, int K, int pad) {
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
  const bool nrm = mean != nullptr;
  z[((size_t)b * N + k) * L + f] =
      pqmf_finish(acc, pqmf_row_scale(rowpeak, b), nrm, nrm ? mean[k] : 0.0f, nrm ? stdv[k] : 1.0f);
}

// Synthesis in polyphase form (the zero-stuffed [B,N,L*N] tensor is never built):
// out[b,t] = N * sum_k sum_{j : (t-pad+j) % N == 0} G[k,j] * z[b,k,(t-pad+j)/N],  t in [0, L*N)
__global__ __launch_bounds__(PQ_THREADS) void pqmf_synthesis_kernel(
    const float* __restrict__ z, const float* __restrict__ G, float* __restrict__ out,
    int L, int N, int K, int pad, int Tout /* samples kept per row (<= L * N), the row stride of out */) {
  const int To = L * N;
  const int t = blockIdx.x * PQ_THREADS + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= Tout) return;
  // first tap j0 >= 0 with (t - pad + j0) % N == 0
  int rem = (t - pad) % N;
  if (rem < 0) rem += N;
  const int j0 = (N - rem) % N;
  float acc = 0.0f;
  const float* zb = z + (size_t)b * N * L;
  for (int j = j0; j < K; j += N) {
    const int u = t - pad + j;            // position in the zero-stuffed signal
    if (u < 0 || u >= To) continue;
    const int m = u / N;                  // the frame that lands there (u % N == 0 by construction of j0)
    float a = 0.0f;
    for (int k = 0; k < N; ++k) a = fmaf(G[(size_t)k * K + j], zb[(size_t)k * L + m], a);
    acc = fmaf(a, (float)N, acc);
  }
  out[(size_t)b * Tout + t] = acc;
}

// Wide synthesis (any N <= 64, odd K <= 255).  With t = N m + r:  out[N m + r] = N sum_d sum_k Gt[d][k][r] z[k][m + d],
// where phase r uses the taps j = j_r + N (d - c_r), j_r = (pad - r) mod N, c_r = (r - pad + j_r) / N, and Gt is
// zero where that j falls outside [0, K) (ias_pqmf_pack_synth_taps folds the gain N in).  The workgroup stages
// z[:, m0 + dmin .. m0 + 127 + dmax] in LDS (odd row stride); a lane owns one frame m and half of the phases: per
// (d, k) one ds_read_b32 feeds HB FMAs with SGPR taps.  The tile's outputs go back through LDS so that the store to
// out[N m0 ..] is contiguous (a lane's own outputs are N floats apart).
struct PqmfSynthGeom { int dmin, nd; };
static PqmfSynthGeom pqmf_synth_geom(int N, int K) {
  const int pad = (K - 1) / 2;
  int dmin = 1 << 30, dmax = -(1 << 30);
  for (int r = 0; r < N; ++r) {
    const int jr = ((pad - r) % N + N) % N;
    if (jr >= K) continue;
    const int cr = (r - pad + jr) / N;             // exact: r - pad + jr is a multiple of N
    const int q = (K - 1 - jr) / N;                // last tap index of the phase
    dmin = cr < dmin ? cr : dmin;
    dmax = cr + q > dmax ? cr + q : dmax;
  }
  PqmfSynthGeom g = {dmin, dmax - dmin + 1};
  return g;
}

template <int NPAD>
__global__ __launch_bounds__(PQ_THREADS) void pqmf_synthesis_wide_kernel(
    const float* __restrict__ z, const float* __restrict__ Gt, float* __restrict__ out, int L, int N, int dmin,
    int nd, int rs /* LDS row stride (odd), >= 128 + nd - 1 and >= N + 1 scaled: see host */,
    unsigned live /* bit (half * 16 + di): some phase of that half has a tap at frame offset di (K <= N: one di per half) */,
    int Tout /* samples kept per row (<= L * N), the row stride of out */) {
  constexpr int HB = NPAD / 2;
  extern __shared__ __attribute__((aligned(16))) float s_zrows[];   // [N][rs]; reused as the output tile [128][N + 1]
  const int tid = threadIdx.x, b = blockIdx.y;
  const int m0 = blockIdx.x * PQW_FRAMES;
  const int cols = PQW_FRAMES + nd - 1;
  const float* zb = z + (size_t)b * N * L;
  for (int i = tid; i < N * cols; i += PQ_THREADS) {
    const int k = i / cols, c = i - k * cols;
    const int m = m0 + dmin + c;
    s_zrows[k * rs + c] = (m >= 0 && m < L) ? zb[(size_t)k * L + m] : 0.0f;
  }
  __syncthreads();
  const int fl = tid & (PQW_FRAMES - 1);
  const int half = __builtin_amdgcn_readfirstlane(tid >> 7);
  float acc[HB];
#pragma unroll
  for (int r = 0; r < HB; ++r) acc[r] = 0.0f;
  for (int di = 0; di < nd; ++di) {
    if (di < 16 && !((live >> (half * 16 + di)) & 1u)) continue;      // all taps of this (half, offset) are zero
    for (int k = 0; k < N; ++k) {
      const float zv = s_zrows[k * rs + fl + di];
      const float* t = Gt + ((size_t)(di * N + k)) * NPAD + half * HB;
#pragma unroll
      for (int r = 0; r < HB; ++r) acc[r] = fmaf(t[r], zv, acc[r]);
    }
  }
  __syncthreads();                                   // every read of the staged rows is done
  const int os = N + 1;                              // padded row stride of the output tile
#pragma unroll
  for (int r = 0; r < HB; ++r) {
    const int ph = half * HB + r;
    if (ph < N) s_zrows[fl * os + ph] = acc[r];
  }
  __syncthreads();
  const long long t0 = (long long)m0 * N;
  float* ob = out + (size_t)b * Tout;
  for (int i = tid; i < PQW_FRAMES * N; i += PQ_THREADS) {
    const int f = i / N, ph = i - f * N;
    if (t0 + i < Tout) ob[t0 + i] = s_zrows[f * os + ph];
  }
}

// Gt[(di*N + k)*NPAD + r] = N * G[k][j_r + N (dmin + di - c_r)] or 0
__global__ void pqmf_pack_synth_kernel(const float* __restrict__ G, float* __restrict__ Gt, int N, int K, int npad,
                                       int dmin, int nd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nd * N * npad) return;
  const int r = i % npad, k = (i / npad) % N, di = i / (npad * N);
  float v = 0.0f;
  if (r < N) {
    const int pad = (K - 1) / 2;
    const int jr = ((pad - r) % N + N) % N;
    const int cr = (r - pad + jr) / N;
    const int q = dmin + di - cr, j = jr + N * q;
    if (q >= 0 && j < K && jr < K) v = (float)N * G[k * K + j];
  }
  Gt[i] = v;
}

// Pt[j*N + k] = H[k][j] for j < K, zero padding up to the table length
__global__ void pqmf_pack_taps_kernel(const float* __restrict__ H, float* __restrict__ Pt, int N, int K, int len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  const int j = i / N, k = i - j * N;
  Pt[i] = j < K ? H[k * K + j] : 0.0f;
}

// ------------------------------------------------------------------------ C ABI
// Workgroups of the persistent fast kernel: 4 per CU.  6 fit (LDS) and run the kernel alone 10 % faster (47 vs 53 us
// at B=128 x 4 s), but they hold 150 KB of each CU's LDS for the whole launch and keep the render / STFT kernels of
// the neighbouring streams off the CU: 4 per CU makes the pipelined step 1.5 % faster.
static int pqmf_resident_blocks() {
  static int blocks = 0;
  if (blocks == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    blocks = cus * 4;
  }
  return blocks;
}

// ---------------------------------------------------------------------------------------------------------------
// N = 3, K = 63 as a cosine-MODULATED filterbank (round 3).  pqmf.py:21-30 builds H[k][j] = g[j] c_k[j], g = 2 x the
// Kaiser prototype, c_k[j] = cos((2k+1) pi/6 (j - 30.5) + (-1)^k pi/4).  For N = 3 that cosine only takes the values
// 0, +-1/2, +-sqrt(3)/2, +-1, repeats with period 12 in j, changes sign every 6 and vanishes for j = 2 (mod 6):
//     v_r[f] = sum_t (-1)^t g[r + 6t] x[3f + r + 6t - 31],   r in {0, 1, 3, 4, 5}            (52 products per frame)
//     z_0 = A + C,  z_2 = A - C,  z_1 = (v_3 - v_1) - v_5,   A = sqrt(3)/2 (v_4 - v_0),  C = (v_3 - v_1)/2 + v_5
// i.e. ~60 vector instructions per frame where the three 63-tap correlations take 189 multiply-adds (243 issue slots
// in the 16x16x4 MFMA form, which -- DESIGN.md section 0 -- are vector-pipe time as well).  The sums are the same real
// numbers in another order and with the constants factored out: they agree with the tap-ordered chain to ~3e-8 of the
// output scale (not bit for bit; the reference's conv1d has no defined summation order either).
// A lane owns 4 consecutive frames: its 72-sample window comes out of the wave's LDS staging as 18 aligned 16-byte
// reads at a 48-byte lane stride (conflict-free), the 52 signed prototype taps sit in scalar registers.
#define PQD_FPL 4                       // frames per lane (the asm operand list of the stream form is written for 4)
#define PQD_WF (64 * PQD_FPL)           // frames per wave tile
#define PQD_STAGE (3 * PQD_WF + 64)     // staged samples per wave tile (3 per frame + 60 of halo, rounded up)
#define PQD_NLOAD ((PQD_STAGE + 63) / 64)
extern "C" int ias_pqmf_modtab_len() { return 64; }
// H_host [3][63] (host copy of PQMF(3).H) -> out_host [64]: a[r5][t] = (-1)^t g[r + 6t] at r5 * 11 + t for
// r = (0, 1, 3, 4, 5)[r5]; IAS_ERR_UNSUPPORTED unless H is this modulation of ONE prototype to 2e-6 of its largest tap.
extern "C" int ias_pqmf_build_modtab(const float* H_host, int N, int K, float* out_host) {
  if (!H_host || !out_host) return IAS_ERR_ARG;
  if (N != 3 || K != 63) return IAS_ERR_UNSUPPORTED;
  double c[3][12], hmax = 0.0;
  for (int k = 0; k < 3; ++k)
    for (int r = 0; r < 12; ++r)
      c[k][r] = cos((2 * k + 1) * (3.14159265358979323846 / 6.0) * ((double)r - 30.5) + ((k & 1) ? -1.0 : 1.0) * 0.78539816339744830962);
  for (int i = 0; i < 3 * 63; ++i) hmax = fmax(hmax, fabs((double)H_host[i]));
  double g[63];
  for (int j = 0; j < 63; ++j) {
    const int r = j % 12;
    // the prototype from the band whose modulation is largest at this tap
    int kb = 0;
    for (int k = 1; k < 3; ++k) if (fabs(c[k][r]) > fabs(c[kb][r])) kb = k;
    g[j] = fabs(c[kb][r]) > 0.4 ? (double)H_host[kb * 63 + j] / c[kb][r] : 0.0;
    for (int k = 0; k < 3; ++k)
      if (fabs((double)H_host[k * 63 + j] - g[j] * c[k][r]) > 2e-6 * hmax) return IAS_ERR_UNSUPPORTED;
  }
  for (int i = 0; i < 64; ++i) out_host[i] = 0.0f;
  const int rs[5] = {0, 1, 3, 4, 5};
  for (int r5 = 0; r5 < 5; ++r5)
    for (int t = 0; rs[r5] + 6 * t < 63; ++t) out_host[r5 * 11 + t] = (float)((t & 1) ? -g[rs[r5] + 6 * t] : g[rs[r5] + 6 * t]);
  return IAS_OK;
}

// Two register budgets of ONE body (same arithmetic, same order of every sum: the same bits).
//   WINDOW (rounds 3-4): the lane's 72-sample window is read into registers first, then the 20 sums run over it: 96 VGPRs.
//   STREAM (round 5): the 18 quads are consumed as they arrive -- each sample goes straight into the sums of the (up to four)
//   frames it belongs to -- so only the 20 accumulators, a few quads and the next tile's 13 prefetched values are live:
//   <= 56 VGPRs, which is what a SIMD has left beside three waves of the persistent render (csrc/voice_ctrl_kernels.hip has
//   the same story).  The step's PQMF is HBM-bound on its own (0.74 of 8 TB/s) while the render is vector-bound and uses a
//   quarter of the bandwidth: a PQMF wave that fits beside the render's runs in its gaps instead of after it.
template <bool NORM, bool STREAM>
__device__ __forceinline__ void pqmf_mod_body(const float* __restrict__ x, const float* __restrict__ modtab, float* __restrict__ z,
                                              const float* __restrict__ mean, const float* __restrict__ stdv,
                                              const float* __restrict__ rowpeak, int T, int L, int tiles_x, int ntiles, int zvec,
                                              float (*s_stage)[PQD_STAGE]) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* st = s_stage[wave];
  const int nwaves = gridDim.x * (PQ_THREADS / 64);
  int tile = blockIdx.x * (PQ_THREADS / 64) + wave;

  // the staged samples of wave tile `t`: x[b][3 f0 - 31 + q], q = lane + 64 i, zero outside the row.
  // STREAM: through a buffer descriptor of the ROW (num_records = 4 T bytes): an index beyond T - 1 is out of the buffer's
  // range and reads as 0 in hardware -- no per-load index, predicate and select (13 x 3 registers that the window form
  // keeps alive across the tile's arithmetic), one byte offset and immediate strides.  A row's first tile (indices below
  // 0) takes the predicated loads.
  float nx[PQD_NLOAD];
  auto fetch = [&](int t) {
    const int b = t / tiles_x, ft = t - b * tiles_x;
    const float* xrow = x + (size_t)b * T;
    const int base = 3 * PQD_WF * ft - 31;
    if (STREAM && ft > 0) {
      // (every tile but a row's first: base >= 0, so the byte offset is non-negative; the hardware compares
      // vgpr offset + immediate offset with num_records, which is why the stride 256 i goes into the offset and not
      // into the scalar offset operand, which the range check ignores)
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xrow), 0, 4 * T, 0x00020000);
      const int voff = 4 * (base + lane);
#pragma unroll
      for (int i = 0; i < PQD_NLOAD; ++i)
        nx[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + 256 * i, 0, 0));
    } else {
#pragma unroll
      for (int i = 0; i < PQD_NLOAD; ++i) {
        const int idx = base + lane + 64 * i;
        nx[i] = (idx >= 0 && idx < T) ? xrow[idx] : 0.0f;
      }
    }
  };
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += nwaves) {
    const int b = tile / tiles_x, ft = tile - b * tiles_x;
#pragma unroll
    for (int i = 0; i < PQD_NLOAD; ++i)
      if (lane + 64 * i < PQD_STAGE) st[lane + 64 * i] = nx[i];
    if (tile + nwaves < ntiles) fetch(tile + nwaves);     // in flight while this tile is computed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // v[i][r5] = sum_t a[r5][t] w[3 i + r + 6 t], w = samples 12 lane .. 12 lane + 71 of the stage
    float v[PQD_FPL][5];
    constexpr int rs[5] = {0, 1, 3, 4, 5};
    if (!STREAM) {
      float w[72];
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        const pq_f32x4 q = *reinterpret_cast<const pq_f32x4*>(st + 12 * lane + 4 * i);
        w[4 * i] = q[0]; w[4 * i + 1] = q[1]; w[4 * i + 2] = q[2]; w[4 * i + 3] = q[3];
      }
#pragma unroll
      for (int i = 0; i < PQD_FPL; ++i)
#pragma unroll
        for (int r5 = 0; r5 < 5; ++r5) {
          float acc = 0.0f;
#pragma unroll
          for (int t = 0; rs[r5] + 6 * t < 63; ++t) acc = fmaf(modtab[r5 * 11 + t], w[3 * i + rs[r5] + 6 * t], acc);
          v[i][r5] = acc;
        }
    } else {
#pragma unroll
      for (int i = 0; i < PQD_FPL; ++i)
#pragma unroll
        for (int r5 = 0; r5 < 5; ++r5) v[i][r5] = 0.0f;
#pragma unroll
      for (int c = 0; c < 18; ++c) {
        const pq_f32x4 q = *reinterpret_cast<const pq_f32x4*>(st + 12 * lane + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < PQD_FPL; ++i) {
            const int j = 4 * c + e - 3 * i;                 // tap of frame i that meets window sample 4 c + e
            if (j >= 0 && j < 63 && j % 6 != 2) {
              const int r = j % 6, r5 = r < 2 ? r : r - 1, t = j / 6;
              v[i][r5] = fmaf(modtab[r5 * 11 + t], q[e], v[i][r5]);   // t ascends with c for every (i, r5): the order above
            }
          }
        // at most a few quads in flight -- the point is the register count.  A sched_barrier alone orders the LDS reads but
        // lets the instruction selector sink every multiply-add below the last of them (all 72 samples live again): the
        // empty asm takes the 20 sums as in / out operands, so what feeds them is complete before it and what follows
        // starts behind it.
        if ((c & 1) == 1) {
          asm volatile("" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]), "+v"(v[0][4]),
                            "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]), "+v"(v[1][3]), "+v"(v[1][4]),
                            "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[2][2]), "+v"(v[2][3]), "+v"(v[2][4]),
                            "+v"(v[3][0]), "+v"(v[3][1]), "+v"(v[3][2]), "+v"(v[3][3]), "+v"(v[3][4]) :: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                     // every read precedes the next tile's staging stores
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float rsc = pqmf_row_scale(rowpeak, b);
    float o[3][PQD_FPL];
#pragma unroll
    for (int i = 0; i < PQD_FPL; ++i) {
      const float d31 = v[i][2] - v[i][1];                // v_3 - v_1
      const float A = 0.86602540378443865f * (v[i][3] - v[i][0]);
      const float C = fmaf(0.5f, d31, v[i][4]);
      const float z0 = A + C, z1 = d31 - v[i][4], z2 = A - C;
      o[0][i] = pqmf_finish(z0, rsc, NORM, NORM ? mean[0] : 0.0f, NORM ? stdv[0] : 1.0f);
      o[1][i] = pqmf_finish(z1, rsc, NORM, NORM ? mean[1] : 0.0f, NORM ? stdv[1] : 1.0f);
      o[2][i] = pqmf_finish(z2, rsc, NORM, NORM ? mean[2] : 0.0f, NORM ? stdv[2] : 1.0f);
    }
    const int f0 = PQD_WF * ft + PQD_FPL * lane;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float* zr = z + ((size_t)b * 3 + k) * L + f0;
      if (zvec && f0 + PQD_FPL <= L) *reinterpret_cast<pq_f32x4*>(zr) = (pq_f32x4){o[k][0], o[k][1], o[k][2], o[k][3]};
      else {
#pragma unroll
        for (int i = 0; i < PQD_FPL; ++i) if (f0 + i < L) zr[i] = o[k][i];
      }
    }
  }
}

#ifndef IAS_PQMF_MOD_MINW
#define IAS_PQMF_MOD_MINW 4   // waves per SIMD the window form is compiled for (register budget 512 / MINW)
#endif
template <bool NORM>
__global__ __launch_bounds__(PQ_THREADS, IAS_PQMF_MOD_MINW) void pqmf_analysis_mod_kernel(
    const float* __restrict__ x, const float* __restrict__ modtab, float* __restrict__ z, const float* __restrict__ mean,
    const float* __restrict__ stdv, const float* __restrict__ rowpeak, int T, int L, int tiles_x, int ntiles, int zvec) {
  __shared__ __attribute__((aligned(16))) float s_stage[PQ_THREADS / 64][PQD_STAGE];
  pqmf_mod_body<NORM, false>(x, modtab, z, mean, stdv, rowpeak, T, L, tiles_x, ntiles, zvec, s_stage);
}
// (amdgpu_num_vgpr(28): on gfx90a+ the backend doubles the value for the unified register file -> a budget of 56)
template <bool NORM>
__global__ __launch_bounds__(PQ_THREADS) __attribute__((amdgpu_num_vgpr(28))) void pqmf_analysis_mods_kernel(
    const float* __restrict__ x, const float* __restrict__ modtab, float* __restrict__ z, const float* __restrict__ mean,
    const float* __restrict__ stdv, const float* __restrict__ rowpeak, int T, int L, int tiles_x, int ntiles, int zvec) {
  __shared__ __attribute__((aligned(16))) float s_stage[PQ_THREADS / 64][PQD_STAGE];
  pqmf_mod_body<NORM, true>(x, modtab, z, mean, stdv, rowpeak, T, L, tiles_x, ntiles, zvec, s_stage);
}

// Floats of the transposed tap table (whole polyphase steps, whole pairs); 0 when (N, K) has no fast path.
extern "C" int ias_pqmf_packed_taps_len(int N, int K) {
  if (N == 3 && K == 63) return PqmfFast<3, 63>::TABLE;
  if (N == 4 && K == 63) return PqmfFast<4, 63>::TABLE;
  return K * pqmf_wide_npad(N, K);   // wide kernel: [K][NPAD]; 0 when only the generic kernel applies
}

// packed [ias_pqmf_packed_taps_len] (device, 8-byte aligned) <- H [N,K] (device).  Re-run whenever H changes.
extern "C" int ias_pqmf_pack_taps(const float* H, float* packed, int N, int K, void* stream_) {
  const int len = ias_pqmf_packed_taps_len(N, K);
  if (!H || !packed || len == 0 || ((uintptr_t)packed & 7)) return IAS_ERR_ARG;
  if ((N == 3 || N == 4) && K == 63)
    hipLaunchKernelGGL(pqmf_pack_taps_kernel, dim3((len + 255) / 256), dim3(256), 0, (hipStream_t)stream_, H,
                       packed, N, K, len);
  else
    hipLaunchKernelGGL(pqmf_pack_wide_kernel, dim3((len + 255) / 256), dim3(256), 0, (hipStream_t)stream_, H,
                       packed, N, K, pqmf_wide_npad(N, K));
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

extern "C" int ias_pqmf_out_len(int T, int N, int K) {
  const int pad = (K - 1) / 2;
  if (T <= 0 || N <= 0 || K <= 0 || T + 2 * pad < K) return IAS_ERR_ARG;
  return (T + 2 * pad - K) / N + 1;
}

// x [B,T] (the reference's [B,1,T]), H [N,K] (module buffer H[N,1,K]), z [B,N,L].
// mean/stdv: optional device pointers [N] (both or neither) for the fused
// AudioEmbedding._preprocess normalisation.
// packed: ias_pqmf_pack_taps table of H (fast kernel for N = 3, 4 with K = 63; wide kernel for other N <= 64);
// NULL, or an (N, K) for which ias_pqmf_packed_taps_len is 0, runs the generic one-lane-per-output kernel.
// rowpeak: optional [B] row peaks max |x| (ias_voice_render's workspace, ias_voice_peaks_offset): the analysis of the
// row normalised as torchsynth's normalize_if_clipping would, without the normalised audio ever being written.
extern "C" int ias_pqmf_analysis(const float* x, const float* H, const float* packed, const float* modtab, float* z,
                                 const float* mean, const float* stdv, const float* rowpeak, int B, int T, int N, int K,
                                 void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !H || !z || B <= 0 || B > 65535 || N <= 0 || N > 65535 || K <= 0 || (K & 1) == 0) return IAS_ERR_ARG;
  if ((mean == nullptr) != (stdv == nullptr)) return IAS_ERR_ARG;
  const int pad = (K - 1) / 2;  // == taps // 2 for K = taps + 1, taps even
  const int L = ias_pqmf_out_len(T, N, K);
  if (L <= 0) return IAS_ERR_ARG;
  if (packed && ((uintptr_t)packed & 7)) return IAS_ERR_ARG;
  static const bool force_valu = ias_diag_env("IAS_PQMF_VALU") != nullptr;   // diagnostics: the pre-MFMA kernels
  static const bool no_mod = ias_diag_env("IAS_PQMF_NOMOD") != nullptr;      // diagnostics: never the modulated form
  if (modtab && !no_mod && !force_valu && N == 3 && K == 63 && (long long)B * N * L < 0x7fffff00LL) {
    // the cosine-modulated form: any alignment, any T
    const int zvec = ((uintptr_t)z & 15) == 0 && (L & 3) == 0;
    const int tiles_x = (L + PQD_WF - 1) / PQD_WF;
    const long long ntiles = (long long)tiles_x * B;
    if (ntiles > 0x7fffffffLL) return IAS_ERR_ARG;
    const long long wgs = (ntiles + PQ_THREADS / 64 - 1) / (PQ_THREADS / 64);
    const long long res = pqmf_resident_blocks();
    const int grid = (int)(wgs < res ? wgs : res);
    // (diagnostic library: IAS_PQMF_MOD_WINDOW=1 takes the 96-register window form of rounds 3-4; same bits)
    const bool window = ias_diag_env("IAS_PQMF_MOD_WINDOW") != nullptr;
    if (window) {
#ifdef IAS_DIAG
      if (mean) hipLaunchKernelGGL(pqmf_analysis_mod_kernel<true>, dim3(grid), dim3(PQ_THREADS), 0, stream, x, modtab, z, mean,
                                   stdv, rowpeak, T, L, tiles_x, (int)ntiles, zvec);
      else hipLaunchKernelGGL(pqmf_analysis_mod_kernel<false>, dim3(grid), dim3(PQ_THREADS), 0, stream, x, modtab, z, mean,
                              stdv, rowpeak, T, L, tiles_x, (int)ntiles, zvec);
#endif
    } else if (mean) hipLaunchKernelGGL(pqmf_analysis_mods_kernel<true>, dim3(grid), dim3(PQ_THREADS), 0, stream, x, modtab, z, mean,
                                        stdv, rowpeak, T, L, tiles_x, (int)ntiles, zvec);
    else hipLaunchKernelGGL(pqmf_analysis_mods_kernel<false>, dim3(grid), dim3(PQ_THREADS), 0, stream, x, modtab, z, mean,
                            stdv, rowpeak, T, L, tiles_x, (int)ntiles, zvec);
  } else if (!force_valu && K == 63 && (N == 3 || N == 64) && T >= 4 && (T & 3) == 0 && ((uintptr_t)x & 15) == 0 &&
      (long long)L * N + 4 * K < 0x7fffffffLL) {
    const int zvec = ((uintptr_t)z & 15) == 0 && (L & 3) == 0;
#define IAS_PQM_LAUNCH(NN, SS, RTT, PSHH, PERCU)                                                                  \
    do {                                                                                                          \
      using C = PqmfMfma<NN, 63, SS, RTT, PSHH>;                                                                  \
      auto kern = pqmf_analysis_mfma_kernel<NN, 63, SS, RTT, PSHH>;                                               \
      static bool attr_set = false;                                                                               \
      if (!attr_set) {                                                                                            \
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,                    \
                                (int)C::LDS_BYTES) != hipSuccess) return IAS_ERR_LAUNCH;                          \
        attr_set = true;                                                                                          \
      }                                                                                                           \
      const int tiles_x = (L + C::FT - 1) / C::FT;                                                                \
      const long long ntiles = (long long)tiles_x * B;                                                            \
      if (ntiles > 0x7fffffffLL) return IAS_ERR_ARG;                                                              \
      const long long res = (long long)pqmf_resident_blocks() / 4 * PERCU;                                        \
      const int grid = (int)(ntiles < res ? ntiles : res);                                                        \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(PQ_THREADS), C::LDS_BYTES, stream, x, H, z, mean, stdv, rowpeak, T, \
                         L, pad, tiles_x, (int)ntiles, zvec);                                                     \
    } while (0)
    static const int percu = ias_diag_env("IAS_PQM_PERCU") ? atoi(ias_diag_env("IAS_PQM_PERCU")) : 0;
    static const bool tiled = ias_diag_env("IAS_PQM_TILED") != nullptr;   // diagnostics: workgroup-tiled kernel for N = 3, 4 too
#define IAS_PQP_LAUNCH(NN, SS, PSHH, NORMM, PERCU)                                                                \
    do {                                                                                                          \
      using C = PqmfPipe<NN, 63, SS, PSHH>;                                                                       \
      auto kern = pqmf_analysis_pipe_kernel<NN, 63, SS, PSHH, NORMM>;                                             \
      static bool attr_set = false;                                                                               \
      if (!attr_set) {                                                                                            \
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,                    \
                                (int)C::LDS_BYTES) != hipSuccess) return IAS_ERR_LAUNCH;                          \
        attr_set = true;                                                                                          \
      }                                                                                                           \
      const int tiles_x = (L + C::WF - 1) / C::WF;                                                                \
      const long long ntiles = (long long)tiles_x * B;              /* wave tiles */                              \
      if (ntiles > 0x7fffffffLL) return IAS_ERR_ARG;                                                              \
      const long long wgs = (ntiles + PQ_THREADS / 64 - 1) / (PQ_THREADS / 64);                                   \
      const long long res = (long long)pqmf_resident_blocks() / 4 * PERCU;                                        \
      const int grid = (int)(wgs < res ? wgs : res);                                                              \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(PQ_THREADS), C::LDS_BYTES, stream, x, H, z, mean, stdv, rowpeak, T, \
                         L, pad, tiles_x, (int)ntiles, zvec);                                                     \
    } while (0)
    if (N == 3 && !tiled && zvec && (long long)B * N * L * 4 < 0x7fffff00LL) {
      if (mean) IAS_PQP_LAUNCH(3, 5, 0, true, (percu ? percu : 3)); else IAS_PQP_LAUNCH(3, 5, 0, false, (percu ? percu : 3));
    } else if (N == 3) IAS_PQM_LAUNCH(3, 5, 4, 0, (percu ? percu : 3));
    else IAS_PQM_LAUNCH(64, 1, 2, 6, (percu ? percu : 2));
#undef IAS_PQP_LAUNCH
#undef IAS_PQM_LAUNCH
  } else if (packed && (N == 3 || N == 4) && K == 63 && T >= N && (long long)L * N + 2 * K < 0x7fffffffLL) {
    constexpr int FT = PqmfFast<3, 63>::FT;
    const int tiles_x = (L + FT - 1) / FT;
    const long long ntiles = (long long)tiles_x * B;
    if (ntiles > 0x7fffffffLL) return IAS_ERR_ARG;
    const int grid = (int)(ntiles < pqmf_resident_blocks() ? ntiles : pqmf_resident_blocks());
    if (N == 3)
      hipLaunchKernelGGL((pqmf_analysis_fast_kernel<3, 63>), dim3(grid), dim3(PQ_THREADS), 0, stream, x, packed, z,
                         mean, stdv, rowpeak, T, L, pad, tiles_x, (int)ntiles);
    else
      hipLaunchKernelGGL((pqmf_analysis_fast_kernel<4, 63>), dim3(grid), dim3(PQ_THREADS), 0, stream, x, packed, z,
                         mean, stdv, rowpeak, T, L, pad, tiles_x, (int)ntiles);
  } else if (packed && !((N == 3 || N == 4) && K == 63) /* those tables have the fast kernel's layout */ &&
             pqmf_wide_npad(N, K) && (long long)L * N + 2 * K < 0x7fffffffLL) {
    const int npad = pqmf_wide_npad(N, K);
    int rs = PQW_FRAMES + (K + N - 1) / N + 1;
    rs |= 1;
    const size_t lds = sizeof(float) * (size_t)N * rs;
    const dim3 grid((L + PQW_FRAMES - 1) / PQW_FRAMES, B), block(PQ_THREADS);
#define IAS_PQW_LAUNCH(NP)                                                                                      \
    hipLaunchKernelGGL((pqmf_analysis_wide_kernel<NP>), grid, block, lds, stream, x, packed, z, mean, stdv, rowpeak, T, \
                       L, N, K, pad, rs)
    if (npad == 8) IAS_PQW_LAUNCH(8);
    else if (npad == 16) IAS_PQW_LAUNCH(16);
    else if (npad == 32) IAS_PQW_LAUNCH(32);
    else IAS_PQW_LAUNCH(64);
#undef IAS_PQW_LAUNCH
  } else {
    hipLaunchKernelGGL(pqmf_analysis_generic_kernel, dim3((L + PQ_THREADS - 1) / PQ_THREADS, N, B),
                       dim3(PQ_THREADS), 0, stream, x, H, z, mean, stdv, rowpeak, T, L, N, K, pad);
  }
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// Floats of the synthesis tap table [nd][N][8|16|32|64] for the wide kernel (0: only the generic kernel applies).
extern "C" int ias_pqmf_synth_taps_len(int N, int K) {
  const int npad = pqmf_wide_npad(N, K);
  if (!npad || (K & 1) == 0) return 0;
  return pqmf_synth_geom(N, K).nd * N * npad;
}
// packed [ias_pqmf_synth_taps_len] (device) <- G [N,K] (device).  Re-run whenever G changes.
extern "C" int ias_pqmf_pack_synth_taps(const float* G, float* packed, int N, int K, void* stream_) {
  const int len = ias_pqmf_synth_taps_len(N, K);
  if (!G || !packed || len == 0) return IAS_ERR_ARG;
  const PqmfSynthGeom g = pqmf_synth_geom(N, K);
  hipLaunchKernelGGL(pqmf_pack_synth_kernel, dim3((len + 255) / 256), dim3(256), 0, (hipStream_t)stream_, G, packed, N,
                     K, pqmf_wide_npad(N, K), g.dmin, g.nd);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// z [B,N,L], G [N,K] (module buffer G[1,N,K]), out [B, L*N] (the reference's [B,1,L*N]).
// packed: ias_pqmf_pack_synth_taps table of G (wide kernel), or NULL (generic one-lane-per-output kernel).
// The same into out [B, T_out], T_out <= L * N: the first T_out samples of every row, contiguous (the adjoint of an
// analysis of T_out samples: no strided view of a [B, L * N] result for the next kernel to walk).
extern "C" int ias_pqmf_synthesis_t(const float* z, const float* G, const float* packed, float* out, int B, int L,
                                    int N, int K, int T_out, void* stream_);
extern "C" int ias_pqmf_synthesis(const float* z, const float* G, const float* packed, float* out, int B, int L,
                                  int N, int K, void* stream_) {
  if ((long long)L * N > 0x7fffffffLL) return IAS_ERR_ARG;
  return ias_pqmf_synthesis_t(z, G, packed, out, B, L, N, K, L * N, stream_);
}
extern "C" int ias_pqmf_synthesis_t(const float* z, const float* G, const float* packed, float* out, int B, int L,
                                    int N, int K, int T_out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!z || !G || !out || B <= 0 || B > 65535 || L <= 0 || N <= 0 || K <= 0 || (K & 1) == 0) return IAS_ERR_ARG;
  const int pad = (K - 1) / 2;
  const long long To = (long long)L * N;
  if (To > 0x7fffffffLL || T_out <= 0 || T_out > To) return IAS_ERR_ARG;
  if (packed && ias_pqmf_synth_taps_len(N, K) > 0) {
    const PqmfSynthGeom g = pqmf_synth_geom(N, K);
    const int npad = pqmf_wide_npad(N, K);
    // LDS: z rows [N][rs] with rs >= 128 + nd - 1, reused as the output tile [128][N + 1]
    int rs = PQW_FRAMES + g.nd - 1;
    const int need_out = (PQW_FRAMES * (N + 1) + N - 1) / N;
    if (rs < need_out) rs = need_out;
    rs |= 1;
    const size_t lds = sizeof(float) * (size_t)N * rs;
    if (lds <= 64 * 1024) {
      const dim3 grid((L + PQW_FRAMES - 1) / PQW_FRAMES, B), block(PQ_THREADS);
      // which (half of the phases, frame offset) pairs carry any tap: with K <= N every phase has ONE offset, and the two
      // halves of the phases one each -- the kernel then runs half its multiply-adds (N = 64, K = 63: 116 -> 60 us)
      unsigned live = 0;
      for (int r = 0; r < N; ++r) {
        const int jr = ((pad - r) % N + N) % N;
        if (jr >= K) continue;
        const int cr = (r - pad + jr) / N, q = (K - 1 - jr) / N, half = r >= npad / 2 ? 1 : 0;
        for (int d = cr; d <= cr + q; ++d)
          if (d - g.dmin < 16) live |= 1u << (half * 16 + d - g.dmin);
      }
      if (g.nd > 16) live = 0xffffffffu;
#define IAS_PQS_LAUNCH(NP)                                                                                       \
      hipLaunchKernelGGL((pqmf_synthesis_wide_kernel<NP>), grid, block, lds, stream, z, packed, out, L, N, g.dmin, \
                         g.nd, rs, live, T_out)
      if (npad == 8) IAS_PQS_LAUNCH(8);
      else if (npad == 16) IAS_PQS_LAUNCH(16);
      else if (npad == 32) IAS_PQS_LAUNCH(32);
      else IAS_PQS_LAUNCH(64);
#undef IAS_PQS_LAUNCH
      return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
    }
  }
  hipLaunchKernelGGL(pqmf_synthesis_kernel, dim3((int)((T_out + PQ_THREADS - 1) / PQ_THREADS), B), dim3(PQ_THREADS), 0,
                     stream, z, G, out, L, N, K, pad, T_out);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
