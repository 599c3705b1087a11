// Depthwise and stem convolutions of the AudioEmbedding trunk for MI355X (gfx950), NCHW fp32.
//
// The reference takes its trunk from torchvision (`mobilenet_v3_small(...).features`,
// /root/reference/vicreg_audio_params.py:52-54, used at audioembed.py:61).  On PyTorch-ROCm the eleven depthwise
// convolutions (3x3 / 5x5, stride 1 / 2) and the 3 -> 16 stem of that network run through MIOpen's fp32 fallbacks:
// naive_conv_* kernels, Winograd kernels of 0.85 ms for a 15 x 16 map, and an im2col + GEMM PER SAMPLE for the stem
// (measured 10 of the 40 ms of a batch-128 pretraining step).  They are memory-bound stencils; these kernels do them
// at stencil cost (LDS-tiled, see dw_tile_kernel), the weight gradient as per-plane partial sums reduced in a fixed
// order (deterministic).
#include "ias_common.h"

#define CV_THREADS 256

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

// ---------------------------------------------------------------------------------------------------------------
// LDS-tiled depthwise forward / stride-1 input gradient / weight gradient.  The first version of these kernels read
// every tap from global memory: 25 L1 hits per output at k = 5, and the L1 (64 B/clk/CU) bounded them (67 us for a
// [128,240,15,16] layer whose HBM traffic is 10 us; dwconv_bwd_data_kernel above still has that form).  Here a
// workgroup stages its input planes (zero-padded halo included) in LDS once; a thread owns FOUR consecutive outputs of
// a row and reads the 3 S + K inputs they share as two or three ds_read_b128 per kernel row (40 B of LDS per output
// instead of 100 B of L1), with no bounds checks in the inner loops.  Small planes share a workgroup (pp planes,
// 256 / pp threads each, pp a power of two); large planes are walked in row tiles.
// MODE 0: out = conv(x, w) (flip != 0: taps reversed -- the stride-1 input gradient is this with x = g).
// MODE 1: partial[plane][K K] = sum over the plane of g * window(x); the per-thread sums meet in LDS and are added in
//         a fixed order (deterministic); a second launch adds the planes of a channel.
#define DW_NB 8
template <int K, int S, int MODE>
__global__ __launch_bounds__(CV_THREADS) void dw_tile_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ g, float* __restrict__ out, int C,
                                                             int H, int W, int Ho, int Wo, int planes, int pp,
                                                             int rows_out, int Wp, int flip, unsigned mWp, unsigned mRows) {
  constexpr int P = (K - 1) / 2, KK = K * K, NSEG4 = (3 * S + K + 3) / 4;
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float s_dw[];
  const int rows_in = (rows_out - 1) * S + K;
  float* s_x = s_dw;                                   // [pp][rows_in][Wp]
  float* s_w = s_dw + pp * rows_in * Wp;               // MODE 0: [pp][KK] taps; MODE 1: [CV_THREADS][KK] partial sums
  const int tid = threadIdx.x;
  const int p0 = blockIdx.x * pp, npl = min(pp, planes - p0);
  const int tpp = CV_THREADS / pp, slot = tid / tpp, tl = tid - slot * tpp;   // this thread's plane and rank in it
  const int quads = (Wo + 3) >> 2;
  if (MODE == 0) {
    for (int i = tid; i < npl * KK; i += CV_THREADS) {
      const int pl = i / KK, t = i - pl * KK;
      s_w[i] = w[((p0 + pl) % C) * KK + (flip ? KK - 1 - t : t)];
    }
  }
  float aw[MODE == 1 ? KK : 1];
#pragma unroll
  for (int i = 0; i < (MODE == 1 ? KK : 1); ++i) aw[i] = 0.0f;

  // row tiles of a large plane: one per blockIdx.y (gridDim.y = number of tiles; 1: the loop walks them)
  for (int r0 = blockIdx.y * rows_out; r0 < Ho; r0 += gridDim.y * rows_out) {
    const int nr = min(rows_out, Ho - r0);
    // MODE 1: the cotangents of this thread's first two items, requested BEFORE the tile so that they arrive with it
    float gpre[2][4] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
    if (MODE == 1 && slot < npl) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int item = tl + it * tpp;
        if (item < nr * quads) {
          const int r = item / quads, q = item - r * quads;
          const float* gp = g + (size_t)(p0 + slot) * Ho * Wo + (r0 + r) * Wo + q * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) gpre[it][j] = (q * 4 + j < Wo) ? gp[j] : 0.0f;
        }
      }
    }
    __syncthreads();                                   // the previous tile has been consumed
    // staging: the tile as one flat index space, DW_NB loads in flight per thread: all of a batch's addresses first, then
    // its loads back to back, then the LDS writes (a row walk with one load per thread and pass left a 61 x 132 tile of the
    // first layer waiting out forty memory latencies: 84 us for 151 MB).  Index -> (plane, row, column) by multiply-high
    // with host-made reciprocals (exact for idx < 2^20).
    {
      const int stage_n = npl * rows_in * Wp;
      for (int base = tid; base < stage_n; base += DW_NB * CV_THREADS) {
        float v[DW_NB];
#pragma unroll
        for (int u = 0; u < DW_NB; ++u) {
          const int idx = base + u * CV_THREADS;
          const int pr = (int)__umulhi((unsigned)idx, mWp), col = idx - pr * Wp;
          const int pl = (int)__umulhi((unsigned)pr, mRows), r = pr - pl * rows_in;
          const int hi = r0 * S - P + r, wi = col - P;
          const bool ok = idx < stage_n && hi >= 0 && hi < H && wi >= 0 && wi < W;
          v[u] = ok ? x[((size_t)(p0 + pl) * H + hi) * W + wi] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < DW_NB; ++u) {
          const int idx = base + u * CV_THREADS;
          if (idx < stage_n) s_x[idx] = v[u];
        }
      }
    }
    __syncthreads();
    if (slot < npl) {
      const size_t obase = (size_t)(p0 + slot) * Ho * Wo;
      for (int it = 0, item = tl; item < nr * quads; ++it, item += tpp) {
        const int r = item / quads, q = item - r * quads;
        const float* row0 = s_x + (slot * rows_in + r * S) * Wp + q * 4 * S;
        const int o = (r0 + r) * Wo + q * 4;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (MODE == 1) {
          if (it < 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = it == 0 ? gpre[0][j] : gpre[1][j];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = (q * 4 + j < Wo) ? g[obase + o + j] : 0.0f;
          }
        }
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
          float seg[4 * NSEG4];
#pragma unroll
          for (int v = 0; v < NSEG4; ++v) {
            const f4 t = *reinterpret_cast<const f4*>(row0 + kh * Wp + 4 * v);
            seg[4 * v] = t[0]; seg[4 * v + 1] = t[1]; seg[4 * v + 2] = t[2]; seg[4 * v + 3] = t[3];
          }
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            if (MODE == 0) {
              const float wt = s_w[slot * KK + kh * K + kw];
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[j] = fmaf(wt, seg[j * S + kw], acc[j]);
            } else {
              float t = aw[kh * K + kw];
#pragma unroll
              for (int j = 0; j < 4; ++j) t = fmaf(gv[j], seg[j * S + kw], t);
              aw[kh * K + kw] = t;
            }
          }
        }
        if (MODE == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (q * 4 + j < Wo) out[obase + o + j] = acc[j];
        }
      }
    }
  }
  if (MODE == 1) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KK; ++i) s_w[tid * KK + i] = aw[i];
    __syncthreads();
    for (int i = tid; i < npl * KK; i += CV_THREADS) {
      const int pl = i / KK, t = i - pl * KK;
      float v = 0.0f;
      for (int j = 0; j < tpp; ++j) v += s_w[(pl * tpp + j) * KK + t];
      out[((size_t)blockIdx.y * planes + p0 + pl) * KK + t] = v;     // [tile][b][c][K K]
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Small planes (round 4): ONE WAVE owns whole planes and streams through them.  dw_tile_kernel gives four 15 x 16 planes
// to a workgroup that loads, waits, computes, stores and retires: 7680 short-lived workgroups at four per CU (100 VGPRs at
// k = 5), every one of them exposed to a full memory latency -- 31 us for 59 MB.  Here a workgroup IS one wave (the barrier
// is free), it walks groups of PW = 64 / tpp consecutive planes with a stride of the grid, and the loads of its next group
// (at most DWW_NL per lane, contiguous: PW planes are one run of PW H W floats) are issued before it computes the current
// one; the halo is zeroed once (the geometry never changes), every index decode happens once per lane, before the loop.
// MODE 0: out = conv(x, w) (flip: taps reversed = the stride-1 input gradient).
// MODE 1: weight gradient.  A group is then PW consecutive CHANNELS of one sample and a wave walks `bch` samples of
//         its channels before it folds its lanes' K K sums (DPP row sums + four readlanes per tap) and writes one partial
//         per channel: partial[batch chunk][c][K K], added in chunk order by conv_reduce_partials_kernel.
#define DWW_NL 16        // staged elements per lane and group: PW H W <= 1024
#define DWW_NI 4         // items (four outputs of a row) per lane: Ho quads <= 4 tpp
__device__ __forceinline__ float dww_row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /* quad_perm [2,3,0,1] */, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124 /* row_ror:4 */, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, true));
  return v;
}
struct DwWave { int PW, tpp, Wp, rows_in, nl, ni, bch, ntiles, rt; unsigned mHW, mW, mQ; size_t lds; bool ok; };
__device__ __forceinline__ int dww_div(int n, unsigned m) { return m ? (int)__umulhi((unsigned)n, m) : n; }   // (m = 0: divisor 1)
// NL / NI: compile-time bounds of geo.nl / geo.ni (register arrays): (4, 1) the 15 x 16 and 8 x 8 planes, (16, 1), (16, 4)
template <int K, int S, int MODE, int NL, int NI>
__global__ __launch_bounds__(64) void dw_wave_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ g, float* __restrict__ out, int B, int C, int H,
                                                     int W, int Ho, int Wo, int flip, const DwWave geo, int nwork) {
  constexpr int P = (K - 1) / 2, KK = K * K, NSEG4 = (3 * S + K + 3) / 4;
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float s_dw[];
  const int PW = geo.PW, tpp = geo.tpp, Wp = geo.Wp, rows_in = geo.rows_in, HW = H * W;
  float* s_x = s_dw;                                   // [PW][rows_in][Wp]
  float* s_w = s_dw + PW * rows_in * Wp;               // [PW][KK] (MODE 0)
  const int lane = threadIdx.x, slot = lane / tpp, tl = lane - slot * tpp;
  // geo.ntiles > 1: a group is ONE ROW TILE of one plane (PW = 1): geo.rt output rows, all rows_in = (rt - 1) S + K input
  // rows staged, the run of rows_in W floats starting (r0 S - P) W floats into the plane is still contiguous.
  const bool tiled = geo.ntiles > 1;
  const int ntiles = geo.ntiles, rt = geo.rt;
  const int quads = (Wo + 3) >> 2, nitems = (tiled ? rt : Ho) * quads, planes = B * C;
  for (int i = lane; i < PW * rows_in * Wp; i += 64) s_x[i] = 0.0f;          // the halo stays zero
  // Once per lane: where its staged elements come from and go to, which items it computes.  The loop below has ONE form of
  // load -- base pointer of the group (uniform) + this lane's fixed offsets -- and no predicates: lanes past the end of
  // the run re-read its last element and park it in a spare word behind the tile; a group that would hang over the end of
  // the tensor / of a sample's channels, or a tile that would hang over the plane, is moved back inside (whole planes /
  // whole rows: results of the overlap are recomputed, bit-identical), see group_of.
  const int run = tiled ? rows_in * W : PW * HW;       // staged floats per group, contiguous in memory
  const int dummy = PW * rows_in * Wp + PW * KK;
  int soff[NL];
  unsigned lofs[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int idx = lane + 64 * i;
    soff[i] = dummy;
    lofs[i] = 4u * (unsigned)min(idx, run - 1);    // bytes
    if (idx < run) {
      if (tiled) {
        const int r = dww_div(idx, geo.mW), wi = idx - r * W;
        soff[i] = r * Wp + wi + P;
      } else {
        const int pl = dww_div(idx, geo.mHW), rem = idx - pl * HW;
        const int hi = dww_div(rem, geo.mW), wi = rem - hi * W;
        soff[i] = (pl * rows_in + hi + P) * Wp + wi + P;
      }
    }
  }
  int irow[NI], iq[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int item = tl + j * tpp;
    irow[j] = -1; iq[j] = 0;
    if (j < geo.ni && item < nitems) { irow[j] = dww_div(item, geo.mQ); iq[j] = item - irow[j] * quads; }
  }
  const int cgroups = (C + PW - 1) / PW;                // MODE 1: channel groups per sample
  const int bch = geo.bch;
  float v[NL], wreg[2] = {0.0f, 0.0f}, gv[MODE == 1 ? NI : 1][4];
  float aw[MODE == 1 ? KK : 1];
  // (work item, step) -> first plane of the group (moved back so that PW planes / channels exist: the host guarantees
  // planes >= PW, C >= PW), whether there is anything to do (MODE 1: samples past the batch), first output row, and the
  // number of rows the staged window was moved DOWN (+) or UP (-) to stay inside the plane
  auto group_of = [&](int work, int step, int& p_start, bool& live, int& r0, int& drows) {
    r0 = 0; drows = 0; live = true;
    if (MODE == 0) {
      if (tiled) {
        p_start = work / ntiles;
        r0 = (work - p_start * ntiles) * rt;
      } else {
        p_start = min(work * PW, planes - PW);
      }
    } else {
      const int bc = work / cgroups, cg = work - bc * cgroups;
      int bs = step;
      if (tiled) { bs = step / ntiles; r0 = (step - bs * ntiles) * rt; }
      const int b = bc * bch + bs;
      live = b < B;
      p_start = b * C + min(cg * PW, C - PW);
    }
    if (tiled) {
      const int hi0 = r0 * S - P;                      // first input row of the tile
      const int hi1 = min(max(hi0, 0), H - rows_in);   // ... of the window that is loaded
      drows = hi1 - hi0;
    }
  };
  auto issue = [&](int work, int step) {
    int p_start, r0, drows;
    bool live;
    group_of(work, step, p_start, live, r0, drows);
    if (!live) return;
    // (the group's base as a scalar pair + the lane's byte offsets: one load instruction per element)
    const size_t goff = (size_t)p_start * HW + (size_t)(tiled ? (r0 * S - P + drows) * W : 0);
    const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)goff), ghi = __builtin_amdgcn_readfirstlane((unsigned)(goff >> 32));
    const char* src = reinterpret_cast<const char*>(x + (((size_t)ghi << 32) | glo));
#pragma unroll
    for (int i = 0; i < NL; ++i) v[i] = *reinterpret_cast<const float*>(src + lofs[i]);
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = lane + 64 * u;
        if (i < PW * KK) {
          const int pl = i / KK, t = i - pl * KK;
          wreg[u] = w[((p_start + pl) % C) * KK + (flip ? KK - 1 - t : t)];
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const bool rowok = irow[j] >= 0 && r0 + irow[j] < Ho;
        const float* gp = g + (size_t)(p_start + slot) * Ho * Wo + (r0 + irow[j]) * Wo + iq[j] * 4;
        if ((Wo & 3) == 0) {                               // (then every plane and row starts on 16 bytes)
          const f4 t = rowok ? *reinterpret_cast<const f4*>(gp) : (f4){0.0f, 0.0f, 0.0f, 0.0f};
          gv[j][0] = t[0]; gv[j][1] = t[1]; gv[j][2] = t[2]; gv[j][3] = t[3];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) gv[j][e] = (rowok && iq[j] * 4 + e < Wo) ? gp[e] : 0.0f;
        }
      }
    }
  };
  // MODE 0: a group's results are stored one iteration LATE, right before the loads of the group after next go out.
  // vmcnt counts loads and stores alike and retires them in order: stores issued after the prefetch would sit between
  // the loop top's wait and the data it waits for; stores issued BEFORE the prefetch have the arithmetic phase to drain.
  float res[MODE == 0 ? NI : 1][4];
  int pplane = -1, pr0 = 0;
  auto flush = [&]() {
    if (MODE == 0 && pplane >= 0) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (irow[j] < 0 || pr0 + irow[j] >= Ho) continue;
        float* dst = out + (size_t)pplane * Ho * Wo + (pr0 + irow[j]) * Wo + iq[j] * 4;
        if ((Wo & 3) == 0) {
          *reinterpret_cast<f4*>(dst) = (f4){res[j][0], res[j][1], res[j][2], res[j][3]};   // (Wo % 4 == 0: 16-byte aligned)
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (iq[j] * 4 + e < Wo) dst[e] = res[j][e];
        }
      }
    }
  };
  const int nsteps = MODE == 1 ? bch * ntiles : 1;
  int work = blockIdx.x, step = 0;
  if (work < nwork) issue(work, 0);
  __syncthreads();
  while (work < nwork) {
    if (MODE == 1 && step == 0) {
#pragma unroll
      for (int i = 0; i < KK; ++i) aw[i] = 0.0f;
    }
    int p_start, r0, drows;
    bool live;
    group_of(work, step, p_start, live, r0, drows);
    // registers -> LDS, then the next group's loads go out before the arithmetic
    if (live) {
      if (drows == 0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) s_x[soff[i]] = v[i];
      } else {
        // a boundary tile: the window was loaded drows rows further down (up) the plane, so everything lands drows rows
        // later (earlier) in the tile, what falls off goes to the spare word, and the rows nothing lands on -- the rows
        // outside the plane -- are zeroed
        const int lim = rows_in * Wp, sh = drows * Wp;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const int so = soff[i] + sh;
          s_x[(soff[i] != dummy && (unsigned)so < (unsigned)lim) ? so : dummy] = v[i];
        }
        const int z0 = drows > 0 ? 0 : lim + sh, zn = drows > 0 ? sh : -sh;
        for (int i = lane; i < zn; i += 64) s_x[z0 + i] = 0.0f;
      }
    }
    float gcur[MODE == 1 ? NI : 1][4];
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (lane + 64 * u < PW * KK) s_w[lane + 64 * u] = wreg[u];
    } else {
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) gcur[j][e] = gv[j][e];
    }
    flush();                                             // the previous group's outputs, ahead of the prefetch
    pplane = live ? p_start + slot : -1;
    pr0 = r0;
    int nwork_i = work, nstep = step + 1;
    if (nstep >= nsteps) { nstep = 0; nwork_i = work + gridDim.x; }
    if (nwork_i < nwork) issue(nwork_i, nstep);
    __syncthreads();
    if (live) {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        if (irow[j] < 0 || r0 + irow[j] >= Ho) continue;
        const int r = irow[j], q = iq[j];
        const float* row0 = s_x + (slot * rows_in + r * S) * Wp + q * 4 * S;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
          float seg[4 * NSEG4];
#pragma unroll
          for (int u = 0; u < NSEG4; ++u) {
            const f4 t = *reinterpret_cast<const f4*>(row0 + kh * Wp + 4 * u);
            seg[4 * u] = t[0]; seg[4 * u + 1] = t[1]; seg[4 * u + 2] = t[2]; seg[4 * u + 3] = t[3];
          }
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            if (MODE == 0) {
              const float wt = s_w[slot * KK + kh * K + kw];
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[e] = fmaf(wt, seg[e * S + kw], acc[e]);
            } else {
              float t = aw[kh * K + kw];
#pragma unroll
              for (int e = 0; e < 4; ++e) t = fmaf(gcur[j][e], seg[e * S + kw], t);
              aw[kh * K + kw] = t;
            }
          }
        }
        if (MODE == 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) res[j][e] = acc[e];
        }
      }
    }
    if (MODE == 1 && nstep == 0) {
      // fold the lanes of each plane slot: 16-lane row sums on the DPP path, rows met through readlane (fixed order)
      const int bc = work / cgroups, cg = work - bc * cgroups;
      const int rows_per_slot = tpp >> 4;                // tpp in {16, 32, 64}
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const float rs = dww_row16_sum(aw[t]);
        const float q0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs), 0));
        const float q1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs), 16));
        const float q2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs), 32));
        const float q3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs), 48));
        float tot;
        if (rows_per_slot == 4) tot = (q0 + q1) + (q2 + q3);
        else if (rows_per_slot == 2) tot = lane == 0 ? q0 + q1 : q2 + q3;
        else tot = lane == 0 ? q0 : lane == 1 ? q1 : lane == 2 ? q2 : q3;
        const int c = min(cg * PW, C - PW) + lane;       // lane s < PW writes slot s (a moved-back group rewrites the same sums)
        if (lane < PW) out[((size_t)bc * C + c) * KK + t] = tot;
      }
    }
    __syncthreads();                                     // the tile has been read: the next one may be written
    work = nwork_i; step = nstep;
  }
  flush();
}

// Stride-2 input gradient, LDS-tiled: a workgroup stages pp cotangent planes [Ho][Wo] with a one-element zero halo; a
// thread owns the 2 x 2 block of gx at (2a + i, 2b + j), whose taps (kh of the parity of i + P, likewise kw) all fall on
// the 3 x 3 cotangents around (a, b).  The taps are added in the order of dwconv_bwd_data_kernel (kh, then kw,
// ascending), so the two kernels give the same bits.
template <int K>
__global__ __launch_bounds__(CV_THREADS) void dw_tile_bwd_s2_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                                    float* __restrict__ gx, int C, int H, int W, int Ho,
                                                                    int Wo, int planes, int pp, unsigned mWp, unsigned mRows) {
  constexpr int P = (K - 1) / 2, KK = K * K;
  extern __shared__ __attribute__((aligned(16))) float s_dw[];
  const int Wp = Wo + 2, rows = Ho + 2;
  float* s_g = s_dw;                                   // [pp][Ho + 2][Wo + 2]
  float* s_w = s_dw + pp * rows * Wp;                  // [pp][KK]
  const int tid = threadIdx.x;
  const int p0 = blockIdx.x * pp, npl = min(pp, planes - p0);
  const int tpp = CV_THREADS / pp, slot = tid / tpp, tl = tid - slot * tpp;
  for (int i = tid; i < npl * KK; i += CV_THREADS) s_w[i] = w[((p0 + i / KK) % C) * KK + i % KK];
  const int stage_n = npl * rows * Wp;
  for (int base = tid; base < stage_n; base += DW_NB * CV_THREADS) {     // DW_NB loads in flight per thread (see dw_tile_kernel)
    float v[DW_NB];
#pragma unroll
    for (int u = 0; u < DW_NB; ++u) {
      const int idx = base + u * CV_THREADS;
      const int pr = (int)__umulhi((unsigned)idx, mWp), col = idx - pr * Wp;
      const int pl = (int)__umulhi((unsigned)pr, mRows), ho = pr - pl * rows - 1;
      const bool ok = idx < stage_n && ho >= 0 && ho < Ho && col >= 1 && col <= Wo;
      v[u] = ok ? g[((size_t)(p0 + pl) * Ho + ho) * Wo + col - 1] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < DW_NB; ++u) {
      const int idx = base + u * CV_THREADS;
      if (idx < stage_n) s_g[idx] = v[u];
    }
  }
  __syncthreads();
  if (slot >= npl) return;
  float* xp = gx + (size_t)(p0 + slot) * H * W;
  const float* sp = s_g + slot * rows * Wp;
  const float* wp = s_w + slot * KK;
  for (int item = tl; item < Ho * Wo; item += tpp) {
    const int a = item / Wo, b = item - a * Wo;
    float G[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) G[r][c] = sp[(a + r) * Wp + b + c];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (2 * a + i >= H) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (2 * b + j >= W) continue;
        float acc = 0.0f;
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
          if ((i + P - kh) & 1) continue;
#pragma unroll
          for (int kw = 0; kw < K; ++kw) {
            if ((j + P - kw) & 1) continue;
            acc = fmaf(wp[kh * K + kw], G[(i + P - kh) / 2 + 1][(j + P - kw) / 2 + 1], acc);
          }
        }
        xp[(size_t)(2 * a + i) * W + 2 * b + j] = acc;
      }
    }
  }
}

struct DwTile { int pp, rows_out, Wp, ntiles; size_t lds; unsigned mWp, mRows; };
#define DW_MAX_TILES 8
// floor(n / d) = umulhi(n, dw_magic(d)) for n < 2^20, d < 2^11 (the staged tile's index space)
static unsigned dw_magic(int d) { return d <= 1 ? 0u /* callers never divide by 1 this way */ : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }
static DwTile dw_tile_geometry(int H, int W, int Ho, int Wo, int K, int S, int mode) {
  (void)H;
  const int P = (K - 1) / 2, quads = (Wo + 3) / 4, nseg4 = (3 * S + K + 3) / 4;
  int Wp = W + 2 * P;
  const int need = (quads - 1) * 4 * S + 4 * nseg4;
  if (Wp < need) Wp = need;
  Wp = (Wp + 3) & ~3;
  const int budget = 8192;                              // floats of staged input per workgroup (32 KB)
  DwTile t;
  const int rows_full = (Ho - 1) * S + K;
  if (rows_full * Wp <= budget) {
    t.rows_out = Ho;
    int pp = 1;
    while (pp * 2 <= 64 && pp * 2 * rows_full * Wp <= budget && pp * 2 * Ho * quads <= CV_THREADS) pp *= 2;
    t.pp = pp;
  } else {
    // a plane too large for one workgroup: row tiles of half the budget (16 KB: ten workgroups per CU), one workgroup each
    // (the first layer's 2048 planes as 2048 workgroups walking two 32 KB tiles each ran 1.6 resident rounds at 60 us)
    int rin = budget / 2 / Wp;
    int ro = (rin - K) / S + 1;
    if (ro < 1) ro = 1;
    if ((Ho + ro - 1) / ro > DW_MAX_TILES) ro = (Ho + DW_MAX_TILES - 1) / DW_MAX_TILES;
    if (((ro - 1) * S + K) * Wp > budget) ro = ((budget / Wp) - K) / S + 1 < 1 ? 1 : ((budget / Wp) - K) / S + 1;   // (very wide rows)
    t.rows_out = ro;
    t.pp = 1;
  }
  t.ntiles = (Ho + t.rows_out - 1) / t.rows_out;
  if (t.ntiles > DW_MAX_TILES) t.ntiles = 1;            // (the kernel then walks the tiles itself)
  t.Wp = Wp;
  const int rows_in = (t.rows_out - 1) * S + K;
  t.lds = sizeof(float) * ((size_t)t.pp * rows_in * Wp + (mode == 1 ? (size_t)CV_THREADS * K * K : (size_t)t.pp * K * K));
  t.mWp = dw_magic(Wp);
  t.mRows = dw_magic(rows_in);
  return t;
}

// out[i] = sum_chunk partial[chunk][i]: one wave per output, lane l adds chunks l, l + 64, ... and the lanes meet in a
// butterfly (a fixed order: deterministic)
__global__ __launch_bounds__(CV_THREADS) void conv_reduce_partials_kernel(const float* __restrict__ partial,
                                                                          float* __restrict__ out, int n, int nchunk) {
  const int i = blockIdx.x * (CV_THREADS / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  float v = 0.0f;
  for (int k = lane; k < nchunk; k += 64) v += partial[(size_t)k * n + i];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  if (lane == 0) out[i] = v;
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

#ifdef IAS_DIAG   // the VALU / LDS form of the stem weight gradient: superseded by stem_bwd_weight_mfma_kernel, diagnostics only
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

#endif
// The same weight gradient on the matrix cores: gw[co][tap] = sum over positions of g[co][pos] * xcol[tap][pos] is a
// 16 x 27 x (B Ho Wo) GEMM.  One wave per (sample, chunk of output rows); per 64 output positions of a row it issues
// 16 k-steps of v_mfma_f32_16x16x4_f32 for each of the two tap tiles (taps 0..15, 16..26): lane (m, q) supplies
// g[co = m] and the tap-m / tap-(m+16) input of one position per k-step, each read straight from global memory (the
// four q of a load instruction read neighbouring positions, so an instruction touches ~16-32 cache lines, not 64; the
// stride-2 input reads hit the L1 4.5 times per element).  The 389 us of
// the LDS version (bound by its 4 LDS reads per 2 FMAs) become ~40 us.
template <int CIN, int COUT>
__global__ __launch_bounds__(CV_THREADS) void stem_bwd_weight_mfma_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                          float* __restrict__ partial, int B, int H, int W,
                                                                          int Ho, int Wo, int chunks, int rows_per_chunk) {
  static_assert(COUT == 16 && CIN * 9 <= 32, "one 16-row tile of output channels, two 16-column tiles of taps");
  typedef float v4f __attribute__((ext_vector_type(4)));
  constexpr int NT = CIN * 9;
  const int wave = blockIdx.x * (CV_THREADS / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wave >= B * chunks) return;
  const int b = wave / chunks, chunk = wave - b * chunks;
  const int m = lane & 15, q = lane >> 4;
  const int t1 = m + 16;
  const bool t1ok = t1 < NT;
  const int ci0 = m / 9, kh0 = (m % 9) / 3, kw0 = m % 3;
  const int ci1 = t1ok ? t1 / 9 : 0, kh1 = t1ok ? (t1 % 9) / 3 : 0, kw1 = t1ok ? t1 % 3 : 0;
  const float* xb0 = x + ((size_t)b * CIN + ci0) * H * W;
  const float* xb1 = x + ((size_t)b * CIN + ci1) * H * W;
  const float* gb = g + ((size_t)b * COUT + m) * Ho * Wo;
  v4f acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  const int ho_end = min(Ho, (chunk + 1) * rows_per_chunk);
  for (int ho = chunk * rows_per_chunk; ho < ho_end; ++ho) {
    const int hi0 = 2 * ho + kh0 - 1, hi1 = 2 * ho + kh1 - 1;
    const bool r0 = hi0 >= 0 && hi0 < H, r1 = t1ok && hi1 >= 0 && hi1 < H;
    const float* xr0 = xb0 + (size_t)(r0 ? hi0 : 0) * W;
    const float* xr1 = xb1 + (size_t)(r1 ? hi1 : 0) * W;
    const float* gr = gb + (size_t)ho * Wo;
    for (int w0 = 0; w0 < Wo; w0 += 64) {
      float a[16], b0[16], b1[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {   // k-slot (q, s) <-> position 16 (s / 4) + 4 q + s % 4: the four q of a load are neighbours
        const int wo = w0 + 16 * (s >> 2) + 4 * q + (s & 3), c0 = 2 * wo + kw0 - 1, c1 = 2 * wo + kw1 - 1;
        a[s] = wo < Wo ? gr[wo] : 0.0f;
        b0[s] = (r0 && c0 >= 0 && c0 < W) ? xr0[c0] : 0.0f;
        b1[s] = (r1 && c1 >= 0 && c1 < W) ? xr1[c1] : 0.0f;
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc1, 0, 0, 0);
      }
    }
  }
  float* pp = partial + (size_t)wave * (COUT * NT);     // D[co = 4 q + reg][tap = m (+16)]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    pp[(4 * q + r) * NT + m] = acc0[r];
    if (t1ok) pp[(4 * q + r) * NT + t1] = acc1[r];
  }
}

// The same GEMM with its operands staged through LDS (round 5).  The kernel above gathers 48 scalars per lane and 64 positions
// straight from global memory, 16-32 cache lines per load instruction: 122 us inside the training step for 211 MB (35 us of
// HBM time).  Here a wave copies what an output row needs -- the 9 input rows (3 channels x rows 2 ho - 1 .. 2 ho + 1, zero
// padded left, right and outside the image) and the 16 rows of g -- with whole 256-byte loads into an LDS block of its own
// (no workgroup barrier: LDS operations of one wave execute in order) and feeds the matrix cores from there.  Bank layout:
// input rows at a stride = 3 (mod 32) floats put tap t of position p in bank t + 2 p, g rows at a stride = 1 (mod 32) put
// channel m in bank m + p; with k-slot (q, s) <-> position w0 + 8 q + (s & 7) + 32 (s >> 3) the 64 lanes of either operand
// read fall on all 32 banks twice -- the minimum.
#define STEM_STAGE_THREADS 128
// XI, GI: 64-float pieces per staged input row / g row (compile-time: every load of a row set is issued before the first
// is waited for, and the NEXT output row's loads are in flight while the matrix cores work on the current one -- with the
// trip counts at run time the compiler waited for each load before the LDS write behind it: 265 us instead of the 122 us
// of the gather form).
template <int CIN, int COUT, int XI, int GI>
__global__ __launch_bounds__(STEM_STAGE_THREADS) void stem_bwd_weight_stage_kernel(
    const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ partial, int B, int H, int W, int Ho, int Wo,
    int chunks, int rows_per_chunk, int XS, int GS) {
  static_assert(COUT == 16 && CIN * 9 <= 32, "one 16-row tile of output channels, two 16-column tiles of taps");
  typedef float v4f __attribute__((ext_vector_type(4)));
  constexpr int NT = CIN * 9, NR = CIN * 3, GW = 64 * GI;
  extern __shared__ float s_stem[];
  const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (STEM_STAGE_THREADS / 64) + wib;
  if (wave >= B * chunks) return;                        // (no workgroup barrier in this kernel)
  float* xs = s_stem + (size_t)wib * (NR * XS + COUT * GS);
  float* gs = xs + NR * XS;
  const int b = wave / chunks, chunk = wave - b * chunks;
  const int m = lane & 15, q = lane >> 4;
  const int t1 = m + 16;
  const bool t1ok = t1 < NT;
  const int x0 = (m / 3) * XS + m % 3;                   // tap t: input row t / 3 (= 3 ci + kh), column offset kw = t % 3
  const int x1 = t1ok ? (t1 / 3) * XS + t1 % 3 : 0;
  const float* xb = x + (size_t)b * CIN * H * W;
  const float* gb = g + (size_t)b * COUT * Ho * Wo;
  v4f acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
  const int ho_end = min(Ho, (chunk + 1) * rows_per_chunk);
  float xr[NR][XI], gr[COUT][GI];
  auto fetch = [&](int ho) {                             // xs[r][i] = x[ci][2 ho + kh - 1][i - 1] or 0 ; gs[co][i] = g[co][ho][i] or 0
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int hi = 2 * ho + r % 3 - 1;
      const bool rv = hi >= 0 && hi < H;
      const float* src = xb + ((size_t)(r / 3) * H + (rv ? hi : 0)) * W;
#pragma unroll
      for (int k = 0; k < XI; ++k) {
        const int i = lane + 64 * k;
        xr[r][k] = (rv && i >= 1 && i <= W) ? src[i - 1] : 0.0f;
      }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float* src = gb + ((size_t)co * Ho + ho) * Wo;
#pragma unroll
      for (int k = 0; k < GI; ++k) {
        const int i = lane + 64 * k;
        gr[co][k] = i < Wo ? src[i] : 0.0f;
      }
    }
  };
  int ho = chunk * rows_per_chunk;
  if (ho < ho_end) fetch(ho);
  for (; ho < ho_end; ++ho) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int k = 0; k < XI; ++k)
        if (lane + 64 * k < XS) xs[r * XS + lane + 64 * k] = xr[r][k];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
      for (int k = 0; k < GI; ++k) gs[co * GS + lane + 64 * k] = gr[co][k];
    if (ho + 1 < ho_end) fetch(ho + 1);                  // in flight during the products below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int w0 = 0; w0 < GW; w0 += 64) {
      float a[16], b0[16], b1[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int pos = w0 + 8 * q + (s & 7) + 32 * (s >> 3);
        a[s] = gs[m * GS + pos];
        b0[s] = xs[x0 + 2 * pos];
        b1[s] = t1ok ? xs[x1 + 2 * pos] : 0.0f;
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc1, 0, 0, 0);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  float* pp = partial + (size_t)wave * (COUT * NT);     // D[co = 4 q + reg][tap = m (+16)]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    pp[(4 * q + r) * NT + m] = acc0[r];
    if (t1ok) pp[(4 * q + r) * NT + t1] = acc1[r];
  }
}

// The stem forward on the matrix cores: out[co][pos] = sum_tap w[co][tap] xcol[tap][pos] with M = 16 output channels,
// K = 27 taps (7 k-steps of 4, the last tap slot zero), N = 16 positions per tile.  One wave per output row (b, ho);
// lane (n, q) keeps its 7 weights w[co = n][tap = 4 ks + q] in registers and reads the tap's input of position
// wo0 + n from global memory (stride-2 reads, 108 B of L1 per position); D gives it out[co = 4 q + r][wo0 + n].
// The VALU version spent 432 broadcast LDS reads + 432 FMAs per position (144 us at B = 128; HBM floor 26 us).
template <int CIN, int COUT>
__global__ __launch_bounds__(CV_THREADS) void stem_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   float* __restrict__ out, int rows, int H, int W, int Ho,
                                                                   int Wo) {
  static_assert(COUT == 16 && CIN * 9 <= 28, "one 16-row tile of output channels, 7 k-steps of taps");
  typedef float v4f __attribute__((ext_vector_type(4)));
  constexpr int NT = CIN * 9, KS = 7;
  const int row = blockIdx.x * (CV_THREADS / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int b = row / Ho, ho = row - b * Ho;
  const int n = lane & 15, q = lane >> 4;
  float aw[KS];
  int off[KS], kwm1[KS];
  bool ok[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int tap = 4 * ks + q;
    const bool tv = tap < NT;
    const int ci = tv ? tap / 9 : 0, kh = tv ? (tap % 9) / 3 : 0, kw = tv ? tap % 3 : 0;
    const int hi = 2 * ho + kh - 1;
    aw[ks] = tv ? w[n * NT + tap] : 0.0f;
    ok[ks] = tv && hi >= 0 && hi < H;
    off[ks] = ok[ks] ? (ci * H + hi) * W : 0;
    kwm1[ks] = kw - 1;
  }
  const float* xb = x + (size_t)b * CIN * H * W;
  float* ob = out + ((size_t)b * COUT * Ho + ho) * Wo;
  for (int wo0 = 0; wo0 < Wo; wo0 += 32) {
    const int wa = wo0 + n, wb = wo0 + 16 + n;
    v4f acc_a = {0.0f, 0.0f, 0.0f, 0.0f}, acc_b = {0.0f, 0.0f, 0.0f, 0.0f};
    float va[KS], vb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ca = 2 * wa + kwm1[ks], cb = 2 * wb + kwm1[ks];
      va[ks] = (ok[ks] && ca >= 0 && ca < W) ? xb[off[ks] + ca] : 0.0f;
      vb[ks] = (ok[ks] && cb >= 0 && cb < W) ? xb[off[ks] + cb] : 0.0f;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[ks], va[ks], acc_a, 0, 0, 0);
      acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[ks], vb[ks], acc_b, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* oc = ob + (size_t)(4 * q + r) * Ho * Wo;
      if (wa < Wo) oc[wa] = acc_a[r];
      if (wb < Wo) oc[wb] = acc_b[r];
    }
  }
}

// The stem forward with its input rows staged the same way (round 5): a wave owns a run of output rows of one sample,
// copies the 9 input rows of an output row into its own LDS block with whole 256-byte loads -- the next row's loads in
// flight while the matrix cores work on the current one -- and reads the taps from there (tap t of position p in bank
// t + 2 p: lane (n, q) reads tap 4 ks + q of position wo0 + n, two lanes per bank).  The gather form above touches ~16 cache
// lines per load instruction.
template <int CIN, int COUT, int XI>
__global__ __launch_bounds__(CV_THREADS) void stem_fwd_stage_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                    float* __restrict__ out, int B, int H, int W, int Ho, int Wo,
                                                                    int chunks, int rows_per_chunk, int XS) {
  static_assert(COUT == 16 && CIN * 9 <= 28, "one 16-row tile of output channels, 7 k-steps of taps");
  typedef float v4f __attribute__((ext_vector_type(4)));
  constexpr int NT = CIN * 9, KS = 7, NR = CIN * 3;
  extern __shared__ float s_stem[];
  const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (CV_THREADS / 64) + wib;
  if (wave >= B * chunks) return;                        // (no workgroup barrier in this kernel)
  float* xs = s_stem + (size_t)wib * (NR * XS);
  const int b = wave / chunks, chunk = wave - b * chunks;
  const int n = lane & 15, q = lane >> 4;
  float aw[KS];
  int xo[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int tap = 4 * ks + q;
    const bool tv = tap < NT;
    aw[ks] = tv ? w[n * NT + tap] : 0.0f;
    xo[ks] = tv ? (tap / 3) * XS + tap % 3 : 0;          // (a tap beyond the 27 multiplies a zero weight)
  }
  const float* xb = x + (size_t)b * CIN * H * W;
  float xr[NR][XI];
  auto fetch = [&](int ho) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int hi = 2 * ho + r % 3 - 1;
      const bool rv = hi >= 0 && hi < H;
      const float* src = xb + ((size_t)(r / 3) * H + (rv ? hi : 0)) * W;
#pragma unroll
      for (int k = 0; k < XI; ++k) {
        const int i = lane + 64 * k;
        xr[r][k] = (rv && i >= 1 && i <= W) ? src[i - 1] : 0.0f;
      }
    }
  };
  const int ho_end = min(Ho, (chunk + 1) * rows_per_chunk);
  int ho = chunk * rows_per_chunk;
  if (ho < ho_end) fetch(ho);
  for (; ho < ho_end; ++ho) {
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int k = 0; k < XI; ++k)
        if (lane + 64 * k < XS) xs[r * XS + lane + 64 * k] = xr[r][k];
    if (ho + 1 < ho_end) fetch(ho + 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float* ob = out + ((size_t)b * COUT * Ho + ho) * Wo;
    for (int wo0 = 0; wo0 < Wo; wo0 += 32) {
      const int wa = wo0 + n, wb = wo0 + 16 + n;
      v4f acc_a = {0.0f, 0.0f, 0.0f, 0.0f}, acc_b = {0.0f, 0.0f, 0.0f, 0.0f};
      float va[KS], vb[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { va[ks] = xs[xo[ks] + 2 * wa]; vb[ks] = xs[xo[ks] + 2 * wb]; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        acc_a = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[ks], va[ks], acc_a, 0, 0, 0);
        acc_b = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[ks], vb[ks], acc_b, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* oc = ob + (size_t)(4 * q + r) * Ho * Wo;
        if (wa < Wo) oc[wa] = acc_a[r];
        if (wb < Wo) oc[wb] = acc_b[r];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
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

// geometry of dw_wave_kernel for x [B,C,H,W] -> [.,.,Ho,Wo]; ok = false: the plane is too large for a wave (dw_tile_kernel)
static DwWave dw_wave_geometry(int B, int H, int W, int Ho, int Wo, int K, int S) {
  DwWave g;
  g.ok = false;
  static const bool off = ias_diag_env("IAS_DW_NO_WAVE") != nullptr;
  static const bool notile = ias_diag_env("IAS_DW_NO_WAVE_TILES") != nullptr;
  const int P = (K - 1) / 2, quads = (Wo + 3) / 4, nseg4 = (3 * S + K + 3) / 4, HW = H * W;
  if (off) return g;
  int Wp = W + 2 * P;
  const int need = (quads - 1) * 4 * S + 4 * nseg4;
  if (Wp < need) Wp = need;
  g.Wp = (Wp + 3) & ~3;
  g.bch = B < 8 ? B : 8;
  g.mHW = HW > 1 ? dw_magic(HW) : 0;
  g.mW = W > 1 ? dw_magic(W) : 0;
  g.mQ = quads > 1 ? dw_magic(quads) : 0;
  g.ntiles = 1;
  g.rt = Ho;
  int nitems = Ho * quads;
  if (HW <= 64 * DWW_NL && nitems <= 64 * DWW_NI) {
    // whole planes: PW of them per wave
    int tpp = nitems <= 16 ? 16 : nitems <= 32 ? 32 : 64;
    while (tpp < 64 && (64 / tpp) * HW > 64 * DWW_NL) tpp *= 2;
    g.tpp = tpp;
    g.PW = 64 / tpp;
    g.rows_in = (Ho - 1) * S + K;
    g.nl = (g.PW * HW + 63) / 64;
  } else {
    // row tiles: the most output rows whose input rows fit the lanes' registers and whose items fit one pass of the wave
    if (notile || quads > 64) return g;
    int rt = 64 / quads;
    while (rt > 1 && ((rt - 1) * S + K) * W > 64 * DWW_NL) --rt;
    if (((rt - 1) * S + K) * W > 64 * DWW_NL) return g;
    g.rt = rt;
    g.ntiles = (Ho + rt - 1) / rt;
    if (g.ntiles < 2) return g;
    g.tpp = 64;
    g.PW = 1;
    g.rows_in = (rt - 1) * S + K;
    if (g.rows_in > H) return g;                           // (the loaded window is kept inside the plane)
    g.nl = (g.rows_in * W + 63) / 64;
    nitems = rt * quads;
  }
  g.ni = (nitems + g.tpp - 1) / g.tpp;
  g.lds = sizeof(float) * ((size_t)g.PW * g.rows_in * g.Wp + (size_t)g.PW * K * K + 4);     // (+ the spare word)
  g.ok = g.nl <= DWW_NL && g.ni <= DWW_NI && g.lds <= 10240;
  return g;
}
#define DW_WAVE_DISPATCH_KS(MODE, NL, NI, ...)                                                              \
  do {                                                                                                   \
    if (K == 3 && S == 1) hipLaunchKernelGGL((dw_wave_kernel<3, 1, MODE, NL, NI>), __VA_ARGS__);         \
    else if (K == 3 && S == 2) hipLaunchKernelGGL((dw_wave_kernel<3, 2, MODE, NL, NI>), __VA_ARGS__);    \
    else if (K == 5 && S == 1) hipLaunchKernelGGL((dw_wave_kernel<5, 1, MODE, NL, NI>), __VA_ARGS__);    \
    else if (K == 5 && S == 2) hipLaunchKernelGGL((dw_wave_kernel<5, 2, MODE, NL, NI>), __VA_ARGS__);    \
    else return IAS_ERR_UNSUPPORTED;                                                                     \
  } while (0)
#define DW_WAVE_DISPATCH(MODE, WG, ...)                                                                    \
  do {                                                                                                   \
    if ((WG).nl <= 4 && (WG).ni <= 1) DW_WAVE_DISPATCH_KS(MODE, 4, 1, __VA_ARGS__);                       \
    else if ((WG).ni <= 1) DW_WAVE_DISPATCH_KS(MODE, 16, 1, __VA_ARGS__);                                 \
    else DW_WAVE_DISPATCH_KS(MODE, 16, 4, __VA_ARGS__);                                                   \
  } while (0)
// waves in the grid: every CU full at this LDS footprint (at most 32 waves), never more than there is work
static int dw_wave_grid(const DwWave& g, int nwork) {
  int per_cu = (int)((size_t)160 * 1024 / (g.lds ? g.lds : 1));
  if (per_cu > 32) per_cu = 32;
  if (per_cu < 1) per_cu = 1;
  const long long full = 256LL * per_cu;
  return (int)(nwork < full ? nwork : full);
}

// Depthwise Conv2d(C, C, K, stride S, padding (K-1)/2, groups=C, bias=False) forward: x [B,C,H,W], w [C,1,K,K] -> out
#define DW_TILE_DISPATCH(MODE, ...)                                                                       \
  do {                                                                                                   \
    if (K == 3 && S == 1) hipLaunchKernelGGL((dw_tile_kernel<3, 1, MODE>), __VA_ARGS__);                 \
    else if (K == 3 && S == 2) hipLaunchKernelGGL((dw_tile_kernel<3, 2, MODE>), __VA_ARGS__);            \
    else if (K == 5 && S == 1) hipLaunchKernelGGL((dw_tile_kernel<5, 1, MODE>), __VA_ARGS__);            \
    else if (K == 5 && S == 2) hipLaunchKernelGGL((dw_tile_kernel<5, 2, MODE>), __VA_ARGS__);            \
    else return IAS_ERR_UNSUPPORTED;                                                                     \
  } while (0)

extern "C" int ias_dwconv_forward(const float* x, const float* w, float* out, int B, int C, int H, int W, int K, int S,
                                  void* stream_) {
  int rc = cv_check(x, w, out, B, C, H, W, K, S);
  if (rc) return rc;
  const int Ho = ias_conv_out_size(H, K, S), Wo = ias_conv_out_size(W, K, S);
  const int planes = B * C;
  const DwWave wg = dw_wave_geometry(B, H, W, Ho, Wo, K, S);
  if (wg.ok && planes >= wg.PW) {
    const int nwork = wg.ntiles > 1 ? planes * wg.ntiles : (planes + wg.PW - 1) / wg.PW;
    DW_WAVE_DISPATCH(0, wg, dim3(dw_wave_grid(wg, nwork)), dim3(64), wg.lds, (hipStream_t)stream_, x, w, (const float*)nullptr, out, B, C,
                     H, W, Ho, Wo, 0, wg, nwork);
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }
  const DwTile t = dw_tile_geometry(H, W, Ho, Wo, K, S, 0);
  DW_TILE_DISPATCH(0, dim3((planes + t.pp - 1) / t.pp, t.ntiles), dim3(CV_THREADS), t.lds, (hipStream_t)stream_, x, w,
                   (const float*)nullptr, out, C, H, W, Ho, Wo, planes, t.pp, t.rows_out, t.Wp, 0, t.mWp, t.mRows);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// its gradient w.r.t. the input: g [B,C,Ho,Wo] -> gx [B,C,H,W]
extern "C" int ias_dwconv_backward_data(const float* g, const float* w, float* gx, int B, int C, int H, int W, int K, int S,
                                        void* stream_) {
  int rc = cv_check(g, w, gx, B, C, H, W, K, S);
  if (rc) return rc;
  const int Ho = ias_conv_out_size(H, K, S), Wo = ias_conv_out_size(W, K, S);
  if (S == 1) {   // Ho = H, Wo = W: the same convolution of g with the taps reversed
    const int planes = B * C;
    const DwWave wg = dw_wave_geometry(B, Ho, Wo, H, W, K, 1);
    if (wg.ok && planes >= wg.PW) {
      const int nwork = wg.ntiles > 1 ? planes * wg.ntiles : (planes + wg.PW - 1) / wg.PW;
      DW_WAVE_DISPATCH(0, wg, dim3(dw_wave_grid(wg, nwork)), dim3(64), wg.lds, (hipStream_t)stream_, g, w, (const float*)nullptr, gx, B,
                       C, Ho, Wo, H, W, 1, wg, nwork);
      return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
    }
    const DwTile t = dw_tile_geometry(Ho, Wo, H, W, K, 1, 0);
    DW_TILE_DISPATCH(0, dim3((planes + t.pp - 1) / t.pp, t.ntiles), dim3(CV_THREADS), t.lds, (hipStream_t)stream_, g, w,
                     (const float*)nullptr, gx, C, Ho, Wo, H, W, planes, t.pp, t.rows_out, t.Wp, 1, t.mWp, t.mRows);
  } else if ((size_t)(Ho + 2) * (Wo + 2) <= 12288 && !ias_diag_env("IAS_DW_S2_DIRECT")) {
    const int planes = B * C, plane_lds = (Ho + 2) * (Wo + 2);
    int pp = 1;
    while (pp * 2 <= 64 && pp * 2 * plane_lds <= 8192 && pp * 2 * Ho * Wo <= CV_THREADS) pp *= 2;
    const size_t lds = sizeof(float) * ((size_t)pp * plane_lds + (size_t)pp * K * K);
    const dim3 grid((planes + pp - 1) / pp), block(CV_THREADS);
    if (K == 3) hipLaunchKernelGGL((dw_tile_bwd_s2_kernel<3>), grid, block, lds, (hipStream_t)stream_, g, w, gx, C, H, W, Ho, Wo, planes, pp, dw_magic(Wo + 2), dw_magic(Ho + 2));
    else hipLaunchKernelGGL((dw_tile_bwd_s2_kernel<5>), grid, block, lds, (hipStream_t)stream_, g, w, gx, C, H, W, Ho, Wo, planes, pp, dw_magic(Wo + 2), dw_magic(Ho + 2));
  } else {      // a plane too large for LDS: taps from global memory
    const dim3 grid(B * C, cv_grid_x(H * W)), block(CV_THREADS);
    CV_DISPATCH(dwconv_bwd_data_kernel, grid, block, 0, (hipStream_t)stream_, g, w, gx, C, H, W, Ho, Wo);
  }
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// floats of scratch for ias_dwconv_backward_weight: one K x K partial per (b, c) plane and row tile (large planes are cut
// into up to DW_MAX_TILES row tiles, one workgroup each); without the plane's size: the upper bound
extern "C" long long ias_dwconv_weight_scratch(int B, int C, int K) {
  if (B <= 0 || C <= 0 || K <= 0) return IAS_ERR_ARG;
  return (long long)B * C * K * K * DW_MAX_TILES;
}
extern "C" long long ias_dwconv_weight_scratch_hw(int B, int C, int H, int W, int K, int S) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !((K == 3 || K == 5) && (S == 1 || S == 2))) return IAS_ERR_ARG;
  const int Ho = ias_conv_out_size(H, K, S), Wo = ias_conv_out_size(W, K, S);
  return (long long)B * C * K * K * dw_tile_geometry(H, W, Ho, Wo, K, S, 1).ntiles;
}

// its gradient w.r.t. the weights: x [B,C,H,W], g [B,C,Ho,Wo] -> gw [C,1,K,K]; scratch: ias_dwconv_weight_scratch floats
// gw == nullptr: the partial sums only -> *nrows rows of C K K floats in `scratch` (ias_dwconv_backward_weight_partials)
static int dw_backward_weight(const float* x, const float* g, float* gw, float* scratch, int B, int C, int H, int W, int K,
                              int S, int* nrows, void* stream_) {
  int rc = cv_check(x, g, scratch, B, C, H, W, K, S);
  if (rc) return rc;
  const int Ho = ias_conv_out_size(H, K, S), Wo = ias_conv_out_size(W, K, S);
  const int planes = B * C;
  const DwWave wg0 = dw_wave_geometry(B, H, W, Ho, Wo, K, S);
  if (wg0.ok && C >= wg0.PW) {
    // partial[batch chunk][c][K K]: ceil(B / bch) chunks <= B, inside the scratch of any sizing call.  Samples per work
    // item: 8 (one fold of the lanes per 8 planes), halved until there are 2048 work items (16 channels x 16 chunks of the
    // first layer were 256 waves on 256 CUs: 320 us)
    DwWave wg = wg0;
    const int cgroups = (C + wg.PW - 1) / wg.PW;
    while (wg.bch > 1 && cgroups * ((B + wg.bch - 1) / wg.bch) < 2048) wg.bch >>= 1;
    const int nchunk = (B + wg.bch - 1) / wg.bch, nwork = cgroups * nchunk;
    DW_WAVE_DISPATCH(1, wg, dim3(dw_wave_grid(wg, nwork)), dim3(64), wg.lds, (hipStream_t)stream_, x, (const float*)nullptr, g, scratch, B,
                     C, H, W, Ho, Wo, 0, wg, nwork);
    const int n = C * K * K;
    if (nrows) *nrows = nchunk;
    if (gw)
      hipLaunchKernelGGL(conv_reduce_partials_kernel, dim3((n + 3) / 4), dim3(CV_THREADS), 0, (hipStream_t)stream_, scratch, gw, n,
                         nchunk);
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }
  const DwTile t = dw_tile_geometry(H, W, Ho, Wo, K, S, 1);
  DW_TILE_DISPATCH(1, dim3((planes + t.pp - 1) / t.pp, t.ntiles), dim3(CV_THREADS), t.lds, (hipStream_t)stream_, x,
                   (const float*)nullptr, g, scratch, C, H, W, Ho, Wo, planes, t.pp, t.rows_out, t.Wp, 0, t.mWp, t.mRows);
  const int n = C * K * K;   // scratch is [tile][b][c][K K]: the partials of a channel are n floats apart
  if (nrows) *nrows = B * t.ntiles;
  if (gw)
    hipLaunchKernelGGL(conv_reduce_partials_kernel, dim3((n + 3) / 4), dim3(CV_THREADS), 0, (hipStream_t)stream_, scratch, gw,
                       n, B * t.ntiles);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
extern "C" int ias_dwconv_backward_weight(const float* x, const float* g, float* gw, float* scratch, int B, int C, int H,
                                          int W, int K, int S, void* stream_) {
  if (!gw) return IAS_ERR_ARG;
  return dw_backward_weight(x, g, gw, scratch, B, C, H, W, K, S, nullptr, stream_);
}
// The same without the reduction launch: -> the number of partial rows (> 0; C K K floats each) for
// ias_reduce_partials_multi, or a negative status
extern "C" int ias_dwconv_backward_weight_partials(const float* x, const float* g, float* scratch, int B, int C, int H, int W,
                                                   int K, int S, void* stream_) {
  int nrows = 0;
  const int rc = dw_backward_weight(x, g, nullptr, scratch, B, C, H, W, K, S, &nrows, stream_);
  return rc != IAS_OK ? rc : nrows;
}

// Stem Conv2d(3, 16, 3, stride 2, padding 1, bias=False): x [B,3,H,W], w [16,3,3,3] -> out [B,16,Ho,Wo]
extern "C" int ias_stem_forward(const float* x, const float* w, float* out, int B, int H, int W, void* stream_) {
  if (!x || !w || !out || B <= 0 || B > 65535 || H <= 0 || W <= 0) return IAS_ERR_ARG;
  const int Ho = ias_conv_out_size(H, 3, 2), Wo = ias_conv_out_size(W, 3, 2);
  if (ias_diag_env("IAS_STEM_VALU") || (long long)3 * H * W > 0x7fffffffLL) {
    hipLaunchKernelGGL((stem_fwd_kernel<3, 16>), dim3(cv_grid_x(Ho * Wo), B), dim3(CV_THREADS), 0, (hipStream_t)stream_, x, w,
                       out, H, W, Ho, Wo);
  } else {
    const int rows = B * Ho;
    // LDS-staged form: input rows at a stride = 3 (mod 32) floats that covers the last position a 32-wide tile reads
    const int GW = (Wo + 31) / 32 * 32;
    int XS = 2 * GW + 2 > W + 2 ? 2 * GW + 2 : W + 2;
    while (XS % 32 != 3) ++XS;
    if (XS <= 320 && !ias_diag_env("IAS_STEM_FWD_GATHER")) {
      // two output rows per wave: in the step 15 / 30 / 60 / 120 chunks of the 120 rows measured 59 / 53 / 50 / 50 us
      int chunks = Ho >= 60 ? 60 : 1;
      if (const char* e = ias_diag_env("IAS_STEM_FWD_CHUNKS")) chunks = atoi(e);
      const int rpc = (Ho + chunks - 1) / chunks, waves = B * chunks;
      const size_t lds = (size_t)(CV_THREADS / 64) * 9 * XS * sizeof(float);
      hipLaunchKernelGGL((stem_fwd_stage_kernel<3, 16, 5>), dim3((waves + CV_THREADS / 64 - 1) / (CV_THREADS / 64)),
                         dim3(CV_THREADS), lds, (hipStream_t)stream_, x, w, out, B, H, W, Ho, Wo, chunks, rpc, XS);
    } else
    hipLaunchKernelGGL((stem_fwd_mfma_kernel<3, 16>), dim3((rows + CV_THREADS / 64 - 1) / (CV_THREADS / 64)),
                       dim3(CV_THREADS), 0, (hipStream_t)stream_, x, w, out, rows, H, W, Ho, Wo);
  }
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

#define STEM_CHUNKS_X 32
extern "C" long long ias_stem_weight_scratch(int B) { return B <= 0 ? IAS_ERR_ARG : (long long)B * STEM_CHUNKS_X * 432; }

// its weight gradient: x [B,3,H,W], g [B,16,Ho,Wo] -> gw [16,3,3,3]; scratch: ias_stem_weight_scratch(B) floats
// gw == nullptr: the partial sums only -> *nrows rows of 432 floats in `scratch` (ias_stem_backward_weight_partials)
static int stem_backward_weight(const float* x, const float* g, float* gw, float* scratch, int B, int H, int W, int* nrows,
                                void* stream_) {
  if (!x || !g || !scratch || B <= 0 || B > 65535 || H <= 0 || W <= 0) return IAS_ERR_ARG;
  const int Ho = ias_conv_out_size(H, 3, 2), Wo = ias_conv_out_size(W, 3, 2);
#ifdef IAS_DIAG
  if (ias_diag_env("IAS_STEM_GW_LDS")) {   // the VALU / LDS form (diagnostics)
    hipLaunchKernelGGL((stem_bwd_weight_kernel<3, 16>), dim3(STEM_CHUNKS_X, B), dim3(CV_THREADS), 0, (hipStream_t)stream_, x,
                       g, scratch, H, W, Ho, Wo, STEM_CHUNKS_X);
  } else
#endif
  {
    const int waves = B * STEM_CHUNKS_X, rows_per_chunk = (Ho + STEM_CHUNKS_X - 1) / STEM_CHUNKS_X;
    // LDS-staged form: strides = 3 and = 1 (mod 32) floats; the input rows reach past the last position a tile reads
    const int GW = (Wo + 63) / 64 * 64;
    int XS = 2 * GW + 2 > W + 2 ? 2 * GW + 2 : W + 2, GS = GW;
    while (XS % 32 != 3) ++XS;
    while (GS % 32 != 1) ++GS;
    const size_t lds = (size_t)(STEM_STAGE_THREADS / 64) * (9 * XS + 16 * GS) * sizeof(float);
    if (GW == 128 && XS <= 320 && lds <= 65536 && !ias_diag_env("IAS_STEM_GW_GATHER")) {
      // half as many chunks as the gather form: 8 resident waves per CU (LDS) x 256 CUs hold B = 128 samples in one round,
      // and only the first of a wave's 8 rows is fetched with nothing to hide behind
      const int sc = STEM_CHUNKS_X / 2, swaves = B * sc, srows = (Ho + sc - 1) / sc;
      hipLaunchKernelGGL((stem_bwd_weight_stage_kernel<3, 16, 5, 2>),
                         dim3((swaves + STEM_STAGE_THREADS / 64 - 1) / (STEM_STAGE_THREADS / 64)), dim3(STEM_STAGE_THREADS), lds,
                         (hipStream_t)stream_, x, g, scratch, B, H, W, Ho, Wo, sc, srows, XS, GS);
      if (nrows) *nrows = swaves;
      if (gw)
        hipLaunchKernelGGL(conv_reduce_partials_kernel, dim3((432 + 3) / 4), dim3(CV_THREADS), 0, (hipStream_t)stream_, scratch,
                           gw, 432, swaves);
      return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
    } else
    hipLaunchKernelGGL((stem_bwd_weight_mfma_kernel<3, 16>), dim3((waves + CV_THREADS / 64 - 1) / (CV_THREADS / 64)),
                       dim3(CV_THREADS), 0, (hipStream_t)stream_, x, g, scratch, B, H, W, Ho, Wo, STEM_CHUNKS_X,
                       rows_per_chunk);
  }
  if (nrows) *nrows = B * STEM_CHUNKS_X;
  if (gw)
    hipLaunchKernelGGL(conv_reduce_partials_kernel, dim3((432 + 3) / 4), dim3(CV_THREADS), 0, (hipStream_t)stream_, scratch, gw,
                       432, B * STEM_CHUNKS_X);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
extern "C" int ias_stem_backward_weight(const float* x, const float* g, float* gw, float* scratch, int B, int H, int W,
                                        void* stream_) {
  if (!gw) return IAS_ERR_ARG;
  return stem_backward_weight(x, g, gw, scratch, B, H, W, nullptr, stream_);
}
// The same without the reduction launch: -> the number of partial rows (> 0; 432 floats each), or a negative status
extern "C" int ias_stem_backward_weight_partials(const float* x, const float* g, float* scratch, int B, int H, int W,
                                                 void* stream_) {
  int nrows = 0;
  const int rc = stem_backward_weight(x, g, nullptr, scratch, B, H, W, &nrows, stream_);
  return rc != IAS_OK ? rc : nrows;
}

// ---- head: Conv2d(C, Cout, kernel 2) on channels-last maps as one GEMM (audioembed.py: conv7 .. conv1) -------------
// patches[(b,i,j)][(c,di,dj)] = x[b, i+di, j+dj, c], x [B,H,W,C] channels-last: the column order is the weight's own
// [Cout][(c,di,dj)] layout, so the GEMMs use weight.view(Cout, 4 C) and give the weight gradient in place (no permuted
// copies of a 16 MB weight per layer and step).  A thread reads its channel at the four pixels (lanes along c:
// coalesced) and writes one 16-byte group.
__global__ __launch_bounds__(CV_THREADS) void conv2x2_patches_kernel(const float* __restrict__ x, float* __restrict__ patches,
                                                                     int H, int W, int C, long long total) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int Ho = H - 1, Wo = W - 1;
  for (long long i = (long long)blockIdx.x * CV_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * CV_THREADS) {
    const int c = (int)(i % C);
    const long long r = i / C;
    const int j = (int)(r % Wo), ii = (int)((r / Wo) % Ho);
    const long long b = r / ((long long)Wo * Ho);
    const float* xp = x + ((b * H + ii) * W + j) * C + c;
    f4 v;
    v[0] = xp[0]; v[1] = xp[C]; v[2] = xp[(size_t)W * C]; v[3] = xp[(size_t)W * C + C];
    *reinterpret_cast<f4*>(patches + i * 4) = v;
  }
}

// its adjoint: gx[b,h,w,c] = sum over the (up to four) patches that contain the pixel
__global__ __launch_bounds__(CV_THREADS) void conv2x2_patches_bwd_kernel(const float* __restrict__ gp, float* __restrict__ gx,
                                                                         int H, int W, int C, long long total) {
  const int Ho = H - 1, Wo = W - 1;
  for (long long i = (long long)blockIdx.x * CV_THREADS + threadIdx.x; i < total; i += (long long)gridDim.x * CV_THREADS) {
    const int c = (int)(i % C);
    const long long px = i / C;
    const int w = (int)(px % W), h = (int)((px / W) % H);
    const long long b = px / ((long long)W * H);
    float acc = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        const int pi = h - di, pj = w - dj;
        if (pi >= 0 && pi < Ho && pj >= 0 && pj < Wo)
          acc += gp[(((b * Ho + pi) * Wo + pj) * C + c) * 4 + 2 * di + dj];
      }
    gx[i] = acc;
  }
}

// The first head layer reads the trunk's output, which is NCHW [B,C,H,W]: the same patches straight from that layout
// (a permute + copy to channels-last in front cost 29 us forward and 18 us backward at [128,576,8,8]).  A workgroup owns
// (sample, 64 channels): the 64 planes are read as they lie (HW contiguous floats each), transposed through LDS (row
// stride HW + 1: conflict-free both ways) and written as patch rows, 16 bytes per (position, channel).
#define C22_CT 64
__global__ __launch_bounds__(CV_THREADS) void conv2x2_patches_nchw_kernel(const float* __restrict__ x, float* __restrict__ patches,
                                                                          int H, int W, int C) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  extern __shared__ float s_tile[];                       // [C22_CT][HW + 1]
  const int HW = H * W, Ho = H - 1, Wo = W - 1, ld = HW + 1;
  const int c0 = blockIdx.x * C22_CT, b = blockIdx.y;
  const int nc = min(C22_CT, C - c0);
  const float* xb = x + ((size_t)b * C + c0) * HW;
  for (int i = threadIdx.x; i < nc * HW; i += CV_THREADS) s_tile[(i / HW) * ld + i % HW] = xb[i];
  __syncthreads();
  for (int i = threadIdx.x; i < Ho * Wo * nc; i += CV_THREADS) {
    const int c = i % nc, pos = i / nc;
    const int pi = pos / Wo, pj = pos - pi * Wo;
    const float* t = s_tile + c * ld + pi * W + pj;
    f4 v;
    v[0] = t[0]; v[1] = t[1]; v[2] = t[W]; v[3] = t[W + 1];
    *reinterpret_cast<f4*>(patches + (((size_t)b * Ho * Wo + pos) * C + c0 + c) * 4) = v;
  }
}
// its adjoint into NCHW: gx[b,c,h,w] = sum over the (up to four) patches that contain the pixel
__global__ __launch_bounds__(CV_THREADS) void conv2x2_patches_bwd_nchw_kernel(const float* __restrict__ gp, float* __restrict__ gx,
                                                                              int H, int W, int C) {
  extern __shared__ float s_tile[];
  const int HW = H * W, Ho = H - 1, Wo = W - 1, ld = HW + 1;
  const int c0 = blockIdx.x * C22_CT, b = blockIdx.y;
  const int nc = min(C22_CT, C - c0);
  for (int i = threadIdx.x; i < HW * nc; i += CV_THREADS) {
    const int c = i % nc, px = i / nc;
    const int h = px / W, w = px - h * W;
    float acc = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        const int pi = h - di, pj = w - dj;
        if (pi >= 0 && pi < Ho && pj >= 0 && pj < Wo)
          acc += gp[((((size_t)b * Ho + pi) * Wo + pj) * C + c0 + c) * 4 + 2 * di + dj];
      }
    s_tile[c * ld + px] = acc;
  }
  __syncthreads();
  float* gb = gx + ((size_t)b * C + c0) * HW;
  for (int i = threadIdx.x; i < nc * HW; i += CV_THREADS) gb[i] = s_tile[(i / HW) * ld + i % HW];
}

static int conv2x2_grid(long long total) {
  long long g = (total + CV_THREADS - 1) / CV_THREADS;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

// x [B,H,W,C] channels-last -> patches [B (H-1) (W-1)][4 C], columns (c, di, dj); H, W >= 2
extern "C" int ias_conv2x2_patches(const float* x, float* patches, int B, int H, int W, int C, void* stream_) {
  if (!x || !patches || B <= 0 || H < 2 || W < 2 || C <= 0) return IAS_ERR_ARG;
  if (((uintptr_t)patches & 15) != 0) return IAS_ERR_ARG;
  const long long total = (long long)B * (H - 1) * (W - 1) * C;
  hipLaunchKernelGGL(conv2x2_patches_kernel, dim3(conv2x2_grid(total)), dim3(CV_THREADS), 0, (hipStream_t)stream_, x, patches, H,
                     W, C, total);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// gp [B (H-1) (W-1)][4 C] -> gx [B,H,W,C]
extern "C" int ias_conv2x2_patches_backward(const float* gp, float* gx, int B, int H, int W, int C, void* stream_) {
  if (!gp || !gx || B <= 0 || H < 2 || W < 2 || C <= 0) return IAS_ERR_ARG;
  const long long total = (long long)B * H * W * C;
  hipLaunchKernelGGL(conv2x2_patches_bwd_kernel, dim3(conv2x2_grid(total)), dim3(CV_THREADS), 0, (hipStream_t)stream_, gp, gx,
                     H, W, C, total);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// the same from / into an NCHW map x, gx [B,C,H,W] (H W <= 255: the 64-channel tile is staged in LDS)
extern "C" int ias_conv2x2_patches_nchw(const float* x, float* patches, int B, int H, int W, int C, void* stream_) {
  if (!x || !patches || B <= 0 || B > 65535 || H < 2 || W < 2 || C <= 0) return IAS_ERR_ARG;
  if (((uintptr_t)patches & 15) != 0) return IAS_ERR_ARG;
  if ((long long)H * W > 255) return IAS_ERR_UNSUPPORTED;
  const size_t lds = (size_t)C22_CT * (H * W + 1) * sizeof(float);
  hipLaunchKernelGGL(conv2x2_patches_nchw_kernel, dim3((C + C22_CT - 1) / C22_CT, B), dim3(CV_THREADS), lds, (hipStream_t)stream_,
                     x, patches, H, W, C);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
extern "C" int ias_conv2x2_patches_backward_nchw(const float* gp, float* gx, int B, int H, int W, int C, void* stream_) {
  if (!gp || !gx || B <= 0 || B > 65535 || H < 2 || W < 2 || C <= 0) return IAS_ERR_ARG;
  if ((long long)H * W > 255) return IAS_ERR_UNSUPPORTED;
  const size_t lds = (size_t)C22_CT * (H * W + 1) * sizeof(float);
  hipLaunchKernelGGL(conv2x2_patches_bwd_nchw_kernel, dim3((C + C22_CT - 1) / C22_CT, B), dim3(CV_THREADS), lds,
                     (hipStream_t)stream_, gp, gx, H, W, C);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// ---- column sums of a row-major [rows, cols] matrix: the bias gradient of the head's convolutions-as-GEMMs
// (audioembed._Conv2x2Fn.backward: g2.sum(0), rows = B Ho Wo up to 6272, cols = dim).  torch's reduction zero-fills the
// result and runs a split reduction (5 + 8..21 us); here: S row slices x 32-column tiles, 8 rows in flight per thread,
// then a fixed-order sum of the S partial rows (deterministic; two short launches).
#define CS_CT 32
#define CS_RG 8
__global__ __launch_bounds__(CS_CT * CS_RG) void colsum_partial_kernel(const float* __restrict__ a, float* __restrict__ partial,
                                                                        int rows, int cols, int rows_per_slice) {
  __shared__ float red[CS_RG][CS_CT];
  const int tx = threadIdx.x % CS_CT, ty = threadIdx.x / CS_CT;
  const int c = blockIdx.x * CS_CT + tx, cc = min(c, cols - 1);
  const int r0 = blockIdx.y * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
  float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  int r = r0 + ty;
  for (; r + 7 * CS_RG < r1; r += 8 * CS_RG)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += a[(size_t)(r + k * CS_RG) * cols + cc];
  for (; r < r1; r += CS_RG) acc[0] += a[(size_t)r * cols + cc];
  red[ty][tx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (ty == 0 && c < cols) {
    float s = red[0][tx];
#pragma unroll
    for (int k = 1; k < CS_RG; ++k) s += red[k][tx];
    partial[(size_t)blockIdx.y * cols + c] = s;
  }
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int S,
                                                            int cols) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float v[16];                                   // S <= 16 (colsum_slices): every load in flight before the first add
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = k < S ? partial[(size_t)k * cols + c] : 0.0f;
  float s = v[0];
#pragma unroll
  for (int k = 1; k < 16; ++k) s += v[k];
  out[c] = s;
}
static int colsum_slices(int rows) {
  const int s = (rows + 63) / 64;               // >= 64 rows per slice (8 per thread), at most 16 slices
  return s < 1 ? 1 : (s > 16 ? 16 : s);
}
// scratch: ias_colsum_scratch_floats(rows, cols) floats
extern "C" long long ias_colsum_scratch_floats(int rows, int cols) {
  if (rows <= 0 || cols <= 0) return IAS_ERR_ARG;
  return (long long)colsum_slices(rows) * cols;
}
// out == nullptr: the slice sums only (S rows of `cols` floats in scratch)
static int colsum_launch(const float* a, float* out, float* scratch, int rows, int cols, int* nrows, void* stream_) {
  if (!a || !scratch || rows <= 0 || cols <= 0) return IAS_ERR_ARG;
  const int S = colsum_slices(rows);
  const int rps = (rows + S - 1) / S;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + CS_CT - 1) / CS_CT, S), dim3(CS_CT * CS_RG), 0, (hipStream_t)stream_, a,
                     scratch, rows, cols, rps);
  if (out) hipLaunchKernelGGL(colsum_finish_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream_, scratch, out, S, cols);
  if (nrows) *nrows = S;
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
extern "C" int ias_colsum(const float* a, float* out, float* scratch, int rows, int cols, void* stream_) {
  if (!out) return IAS_ERR_ARG;
  return colsum_launch(a, out, scratch, rows, cols, nullptr, stream_);
}
// The same without its second launch -> the number of slice rows (> 0; `cols` floats each, in slice order) left in
// `scratch` for ias_reduce_partials_multi (which adds <= 16 rows in row order, as the second launch does: the same bits),
// or a negative IAS_ERR_*
extern "C" int ias_colsum_partials(const float* a, float* scratch, int rows, int cols, void* stream_) {
  int S = 0;
  const int rc = colsum_launch(a, nullptr, scratch, rows, cols, &S, stream_);
  return rc != IAS_OK ? rc : S;
}
