// VICReg loss for MI355X (gfx950): invariance MSE + variance hinge + off-diagonal covariance^2.
//
// Replaces /root/reference/vicreg.py:35-58 (VICReg.loss) and :73-76 (off_diagonal):
//   repr = mse(x, y); x -= mean_b(x); std = sqrt(var_unbiased + 1e-4); std_loss = mean(relu(1 - std))/2 (+ y)
//   cov = x^T x / (cfg.vicreg.batch_size - 1)   <- the CONFIGURED batch size, not x.shape[0]
//   cov_loss = sum_{i != j} cov_ij^2 / embeddim (+ y);  loss = sim*repr + std*std_loss + cov*cov_loss
//
// Kernels
//   vicreg_colstats_kernel  per 64 columns: column mean (fp32), centred sum of squares, sum (x-y)^2,
//                           and the centred matrix cast to bf16 and TRANSPOSED to Xt[D][Kpad]
//                           (feature-major, batch contiguous) through an LDS tile -- centring is done
//                           in fp32 BEFORE the bf16 cast.
//   vicreg_gram_kernel      C = Xt Xt^T on the matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate),
//                           128x128 output tile per workgroup, upper-triangular tiles only (C is
//                           symmetric: off-diagonal tiles count twice); the D x D matrix is never
//                           stored: each tile is squared, summed (diagonal elements skipped) and
//                           reduced to one fp64 partial.
//   vicreg_finish_kernel    fixed-order reduction of all partials -> (loss, repr, std, cov).
// The Gram is the one dense contraction of the path: 2*B*D^2 flops nominal per branch
// (vicreg.py:47-48), of which the symmetric half is executed.
#include "ias_common.h"
#include <cstdlib>
#include <hip/hip_bf16.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float vc_f32x4;

#define VC_COLS 32        // columns per colstats workgroup: D / 32 = 256 workgroups at D = 8192, one per CU (64 columns
                          // per workgroup left half the chip idle: 60 us at B = 1024 against an HBM time of ~30)
#define VC_THREADS 1024  // 16 waves per workgroup: the column pass is latency-bound with few workgroups
#define GT 128            // gram output tile (GT x GT)
#define GK 64             // k-chunk staged per iteration
#define GLD (GK + 8)      // LDS row stride in bf16 (144 B: breaks the 128-B row bank aliasing)

__device__ __forceinline__ unsigned short f2bf(float f) {
  return __builtin_bit_cast(unsigned short, __float2bfloat16(f));
}

// x, y: [B, D] fp32.  Xt_*: [D][Kpad] bf16 (zero padded in k).  colstats: [4][D] = mean_x, mean_y, m2_x, m2_y
// msepart: [gridDim.x] fp64.
// A wave covers the workgroup's 32 columns (128-byte row segments) of TWO rows at a time: lane = (row half, column).
// NR > 0: batch <= 32 NR, every thread keeps its NR rows of x and y in registers between the two passes (x and y are
// read ONCE); NR = 0: any batch, the second pass reads them again.
template <int NR>
__global__ __launch_bounds__(VC_THREADS) void vicreg_colstats_kernel(
    const float* __restrict__ x, const float* __restrict__ y, unsigned short* __restrict__ Xt_x,
    unsigned short* __restrict__ Xt_y, float* __restrict__ colstats, double* __restrict__ msepart,
    double* __restrict__ hingepart, double* __restrict__ diagpart, unsigned short* __restrict__ Xc_x,
    unsigned short* __restrict__ Xc_y, int B, int D, int Kpad,
    size_t ld /* row stride of x and y in floats (>= D: x, y may be column blocks of a wider matrix) */) {
  constexpr int NW = VC_THREADS / 64, RPI = 2 * NW;              // rows per iteration of the workgroup
  __shared__ float s_red[6][NW][64];
  __shared__ float s_mean[2][VC_COLS];
  __shared__ unsigned short s_tile[2][VC_COLS][64 + 2];
  __shared__ double s_mse[NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & (VC_COLS - 1), rsub = lane >> 5;
  const int j0 = blockIdx.x * VC_COLS, j = j0 + col;
  const bool jok = j < D;

  // pass 1: column sums (wave w takes rows 2 w + rsub, + 32, ...), sum (x-y)^2
  float sx = 0.f, sy = 0.f, se = 0.f;
  float keepx[NR > 0 ? NR : 1], keepy[NR > 0 ? NR : 1];
  if (NR > 0) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int b = 2 * wave + rsub + RPI * i;
      keepx[i] = 0.f; keepy[i] = 0.f;
      if (jok && b < B) { keepx[i] = x[(size_t)b * ld + j]; keepy[i] = y[(size_t)b * ld + j]; }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {                               // (rows beyond B hold zeros: they add nothing)
      sx += keepx[i]; sy += keepy[i];
      const float d = keepx[i] - keepy[i];
      se = fmaf(d, d, se);
    }
  } else {
    for (int b = 2 * wave + rsub; b < B; b += RPI) {
      if (jok) {
        const float xv = x[(size_t)b * ld + j], yv = y[(size_t)b * ld + j];
        sx += xv; sy += yv;
        const float d = xv - yv;
        se = fmaf(d, d, se);
      }
    }
  }
  s_red[0][wave][lane] = sx; s_red[1][wave][lane] = sy;
  // mse: reduce within the wave, then across waves
  for (int d = 32; d > 0; d >>= 1) se += __shfl_xor(se, d, 64);
  if (lane == 0) s_mse[wave] = (double)se;
  __syncthreads();
  if (wave == 0 && lane < VC_COLS) {
    float mx = 0.f, my = 0.f;
    for (int w = 0; w < NW; ++w) {                               // fixed order: waves, row halves
      mx += s_red[0][w][lane]; mx += s_red[0][w][lane + 32];
      my += s_red[1][w][lane]; my += s_red[1][w][lane + 32];
    }
    s_mean[0][lane] = mx / (float)B;
    s_mean[1][lane] = my / (float)B;
    if (lane == 0) {
      double m = 0.0;
      for (int w = 0; w < NW; ++w) m += s_mse[w];
      msepart[blockIdx.x] = m;
    }
  }
  __syncthreads();
  const float mx = s_mean[0][col], my = s_mean[1][col];

  // pass 2: centred sum of squares + bf16 transpose, 64 rows at a time
  float qx = 0.f, qy = 0.f, rx = 0.f, ry = 0.f;                 // (r*: the same sums over the bf16-ROUNDED values)
  auto chunk = [&](const int c64) {
    const int b0 = 64 * c64;
#pragma unroll
    for (int u = 0; u < 2; ++u) {                                // rows r = 2 wave + rsub (+ 32) of this 64-row chunk
      const int r = 2 * wave + rsub + RPI * u;
      const int b = b0 + r;
      float cx = 0.f, cy = 0.f;
      if (jok && b < B) {
        if (NR > 0) {
          cx = keepx[(2 * c64 + u) < NR ? 2 * c64 + u : 0] - mx;
          cy = keepy[(2 * c64 + u) < NR ? 2 * c64 + u : 0] - my;
        } else {
          cx = x[(size_t)b * ld + j] - mx;
          cy = y[(size_t)b * ld + j] - my;
        }
        qx = fmaf(cx, cx, qx);
        qy = fmaf(cy, cy, qy);
      }
      const unsigned short bx = f2bf(cx), by = f2bf(cy);
      const float fx = __uint_as_float((unsigned)bx << 16), fy = __uint_as_float((unsigned)by << 16);
      rx = fmaf(fx, fx, rx);
      ry = fmaf(fy, fy, ry);
      s_tile[0][col][r] = bx;
      s_tile[1][col][r] = by;
      // the same values batch-major, Xc[Kpad][D] (zero rows beyond B): the backward's B x B Gram contracts over D
      if (jok) { Xc_x[(size_t)b * D + j] = bx; Xc_y[(size_t)b * D + j] = by; }
    }
    __syncthreads();
    // write out: 32 columns x 64 k as 128-byte rows; thread -> (column c = tid/4, 16 k values)
    {
      const int c = tid >> 2, part = tid & 3;
      if (tid < 4 * VC_COLS && j0 + c < D) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          unsigned short* dst = (m == 0 ? Xt_x : Xt_y) + (size_t)(j0 + c) * Kpad + b0 + part * 16;
          unsigned int pk[8];
#pragma unroll
          for (int e = 0; e < 8; ++e)
            pk[e] = (unsigned int)s_tile[m][c][part * 16 + 2 * e] | ((unsigned int)s_tile[m][c][part * 16 + 2 * e + 1] << 16);
          uint4* d4 = reinterpret_cast<uint4*>(dst);
          d4[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
          d4[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        }
      }
    }
    __syncthreads();
  };
  if constexpr (NR > 0) {
#pragma unroll
    for (int c64 = 0; c64 < (NR + 1) / 2; ++c64) {
      if (64 * c64 >= Kpad) break;
      chunk(c64);
    }
  } else {
    for (int c64 = 0; 64 * c64 < Kpad; ++c64) chunk(c64);
  }
  s_red[2][wave][lane] = qx; s_red[3][wave][lane] = qy;
  s_red[4][wave][lane] = rx; s_red[5][wave][lane] = ry;
  __syncthreads();
  const bool fin = wave == 0 && lane < VC_COLS && j0 + lane < D;
  float ax = 0.f, ay = 0.f, tx = 0.f, ty = 0.f;
  if (fin) {
    for (int w = 0; w < NW; ++w) {
      ax += s_red[2][w][lane]; ax += s_red[2][w][lane + 32];
      ay += s_red[3][w][lane]; ay += s_red[3][w][lane + 32];
      tx += s_red[4][w][lane]; tx += s_red[4][w][lane + 32];
      ty += s_red[5][w][lane]; ty += s_red[5][w][lane + 32];
    }
    colstats[0 * (size_t)D + j0 + lane] = s_mean[0][lane];
    colstats[1 * (size_t)D + j0 + lane] = s_mean[1][lane];
    colstats[2 * (size_t)D + j0 + lane] = ax;
    colstats[3 * (size_t)D + j0 + lane] = ay;
  }
  // variance hinge partial of this column block
  if (wave == 0) {
    float h = 0.f;
    if (fin) {
      const float inv_bm1 = 1.0f / (float)(B - 1);
      h = fmaxf(1.0f - sqrtf(ax * inv_bm1 + 0.0001f), 0.f) + fmaxf(1.0f - sqrtf(ay * inv_bm1 + 0.0001f), 0.f);
    }
    for (int d = 32; d > 0; d >>= 1) h += __shfl_xor(h, d, 64);
    if (lane == 0) hingepart[blockIdx.x] = (double)h;
    // the diagonal of the D x D Gram of the ROUNDED matrices, squared and summed: what the batch-side form of the
    // covariance term (vicreg_stage_ld) takes off ||Xc Xc^T||_F^2
    double dg = (double)tx * (double)tx + (double)ty * (double)ty;
    for (int d = 32; d > 0; d >>= 1) dg += __shfl_xor(dg, d, 64);
    if (lane == 0) diagpart[blockIdx.x] = dg;
  }
}

// One upper-triangular 128x128 tile of C = Xt Xt^T per workgroup and branch (blockIdx.y); partial[tile] = sum of
// squares of the tile's off-diagonal elements (x2 for tiles above the diagonal).  Any contraction depth (batch > 128:
// BASELINE configs[3], global batch 1024).  The k loop is double-buffered: the next 64-deep chunk of both panels is
// fetched into registers while the matrix cores work on the current one and written to the other LDS buffer afterwards
// (one barrier per chunk).
__global__ __launch_bounds__(256, 2) void vicreg_gram_kernel(const unsigned short* __restrict__ Xt_x,
                                                             const unsigned short* __restrict__ Xt_y,
                                                             double* __restrict__ part_x, double* __restrict__ part_y,
                                                             int D, int Kpad, int ntile) {
  __shared__ __attribute__((aligned(16))) unsigned short s_a[2][GT][GLD];
  __shared__ __attribute__((aligned(16))) unsigned short s_b[2][GT][GLD];
  __shared__ double s_part[4];
  const unsigned short* Xt = blockIdx.y ? Xt_y : Xt_x;
  double* partials = blockIdx.y ? part_y : part_x;
  // linear upper-triangular index -> (ti, tj), ti <= tj
  int t = blockIdx.x, ti = 0;
  {
    int rowlen = ntile;
    while (t >= rowlen) { t -= rowlen; --rowlen; ++ti; }
  }
  const int tj = ti + t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;      // 2 x 2 waves, 64 x 64 each
  const int r = lane & 31, h = lane >> 5;
  const int row0 = ti * GT, col0 = tj * GT;

  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

  // one 64-deep chunk of A (rows row0..+127) and B (rows col0..+127): 128 rows x 128 B each, 16 B per thread x 4
  uint4 sta[4], stb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (tid >> 3) + 32 * i, ch = tid & 7;
      sta[i] = make_uint4(0, 0, 0, 0); stb[i] = make_uint4(0, 0, 0, 0);
      if (row0 + rr < D) sta[i] = *reinterpret_cast<const uint4*>(Xt + (size_t)(row0 + rr) * Kpad + k0 + ch * 8);
      if (col0 + rr < D) stb[i] = *reinterpret_cast<const uint4*>(Xt + (size_t)(col0 + rr) * Kpad + k0 + ch * 8);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (tid >> 3) + 32 * i, ch = tid & 7;
      *reinterpret_cast<uint4*>(&s_a[buf][rr][ch * 8]) = sta[i];
      *reinterpret_cast<uint4*>(&s_b[buf][rr][ch * 8]) = stb[i];
    }
  };
  fetch(0);
  commit(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < Kpad; k0 += GK) {
    const bool more = k0 + GK < Kpad;
    if (more) fetch(k0 + GK);              // in flight while the matrix cores work on this chunk
#pragma unroll
    for (int ks = 0; ks < GK / 16; ++ks) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
        fa[m] = *reinterpret_cast<const bf16x8*>(&s_a[buf][wr * 64 + m * 32 + r][ks * 16 + h * 8]);
#pragma unroll
      for (int n = 0; n < 2; ++n)
        fb[n] = *reinterpret_cast<const bf16x8*>(&s_b[buf][wc * 64 + n * 32 + r][ks * 16 + h * 8]);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
    }
    if (more) commit(buf ^ 1);             // the other buffer: nobody reads it during this chunk
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: sum of squares; C/D layout of 32x32: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  float s = 0.f;
  const bool diag_tile = (ti == tj);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v = acc[m][n][e];
        if (diag_tile) {
          const int row = wr * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int col = wc * 64 + n * 32 + r;
          if (row != col) s = fmaf(v, v, s);
        } else {
          s = fmaf(v, v, s);
        }
      }
  for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
  if (lane == 0) s_part[wave] = (double)s;
  __syncthreads();
  if (tid == 0) {
    const double tot = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    partials[blockIdx.x] = diag_tile ? tot : 2.0 * tot;
  }
}

// ---- Gram for a contraction depth of 128 (batch <= 128, BASELINE configs[2]) -----------------------------------
// At K = 128 a 128 x 128 tile is only 32 MFMAs per wave: what bounds the kernel is the panel traffic L2 -> LDS
// (~30 B/clk per CU at best), not the matrix cores.  So a workgroup keeps a PAIR of row panels (256 rows x K = 128) in
// registers as A fragments (128 VGPRs per lane) and streams the column panels past them: one 32 KB panel fetched feeds
// 64 MFMAs per wave (two tiles), 16 B/clk at full matrix rate.  The panels come in by LDS-DMA (no VGPR staging) into a
// ring of four 32 KB slots (the pair of panels being read + the pair landing); one raw s_barrier per PAIR of steps.  LDS image of a panel:
// [128 rows][16 granules of 16 B], granule g of row r at slot g ^ (r & 15) -- conflict-free for the fragment reads
// (16 consecutive rows, same granule) and lane-linear for the DMA once the SOURCE address is permuted the same way.
// Tiles below the diagonal of a pair (2p+1, 2p) are computed and given weight 0 (1.5 % of the work).
#define GP_RING 4
#define GP_PANEL_BYTES (GT * 128 * 2)

typedef __attribute__((address_space(3))) void gram_lds_void;
typedef const __attribute__((address_space(1))) void gram_glb_void;

// A step = one column panel tj streamed past the row-panel pair p of one branch, tj = 2p .. ntile-1.  A work item is a
// run of at most GP_CH consecutive steps of ONE pair: (branch, p, chunk), enumerated branch-major, p-major.  One
// workgroup per item (235 - 260 items at D = 8192: one resident round of one 128 KB-LDS workgroup per CU).
#define GP_CH 10
__device__ __forceinline__ void gp_item(int item, int ntile, int& p, int& c) {
  p = 0;
  for (;;) {
    const int nc = (ntile - 2 * p + GP_CH - 1) / GP_CH;
    if (item < nc) break;
    item -= nc; ++p;
  }
  c = item;
}
static int gp_nitems(int ntile) {
  int n = 0;
  for (int p = 0; 2 * p < ntile; ++p) n += (ntile - 2 * p + GP_CH - 1) / GP_CH;
  return n;
}

__device__ __forceinline__ void gp_lds_read_b128(bf16x8& dst, unsigned lds_addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(lds_addr) : "memory");
}
// the wait that makes two asm-loaded fragments usable: they pass through it, so no consumer can be scheduled above it
__device__ __forceinline__ void gp_lds_wait(bf16x8& a, bf16x8& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
}

#ifdef GP_STAMPS
// Diagnostic build only (scripts/diag): s_memtime stamps of wave 0 of every workgroup, into a buffer of their own.
__device__ unsigned long long* g_gp_stamps = nullptr;
extern "C" int ias_vicreg_debug_set_stamps(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_gp_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#define GPSTAMP(i) do { if (g_gp_stamps && (threadIdx.x & 63) == 0) g_gp_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 32 + (i)] = __builtin_amdgcn_s_memtime(); if (g_gp_stamps && (threadIdx.x & 63) == 0 && (i) == 0) g_gp_stamps[((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 32 + 31] = __builtin_amdgcn_s_getreg(6 << 11 | 4 << 6 | 4); } while (0)
#else
#define GPSTAMP(i) do {} while (0)
#endif

#define GP_THREADS 512      // 8 waves, two per SIMD: one wave's square-sum epilogue runs beside its partner's MFMAs
__global__ __launch_bounds__(GP_THREADS, 2) void vicreg_gram_pair_kernel(const unsigned short* __restrict__ Xt_x,
                                                                         const unsigned short* __restrict__ Xt_y,
                                                                         double* __restrict__ part_x,
                                                                         double* __restrict__ part_y, int D, int ntile,
                                                                         int nitems) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_ring[];   // GP_RING x 32 KB
  __shared__ double s_part[GP_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;       // wave tile: rows wr*64 .. +63 of the 256-row pair, columns wc*64 .. +63
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware placement (speed only): consecutive block indices are dealt round-robin over the 8 XCDs, each with its own
  // 4 MB L2.  Blocks with (blockIdx % 8) < 4 work on branch x, the others on branch y, so that an XCD's L2 has to hold
  // only ONE of the two 2 MB matrices -- with both in every L2 a third of the panel requests missed it and the ring
  // filled at the Infinity-Cache rate (measured: 1750 of every 3650 cycles per step spent waiting for the panel).
  const int xslot = blockIdx.x & 7, branch = xslot >> 2;
  const int item = (blockIdx.x >> 3) * 4 + (xslot & 3);
  if (item >= nitems) return;
  const unsigned short* Xt = branch ? Xt_y : Xt_x;
  int p, c;
  gp_item(item, ntile, p, c);
  const int tj0 = 2 * p + c * GP_CH, nsteps = min(GP_CH, ntile - tj0);
  const bool ragged = (D % GT) != 0;

  // panel tj -> ring slot: 4 LDS-DMA instructions per wave (1 KB = 4 rows each); rows beyond D repeat row D - 1
  // (as columns they are masked in the epilogue, as rows of A they are zeroed below)
  auto request_piece = [&](int tj, int slot_i, int i) {
    unsigned char* slot = s_ring + slot_i * GP_PANEL_BYTES;
    const int row = wave * 16 + i * 4 + (lane >> 4);          // row of the panel this lane's 16 bytes belong to
    const int g = (lane & 15) ^ (row & 15);                    // source granule for LDS slot (lane & 15)
    int grow = tj * GT + row;
    grow = grow < D ? grow : D - 1;
    __builtin_amdgcn_global_load_lds((gram_glb_void*)(Xt + (size_t)grow * 128 + g * 8),
                                     (gram_lds_void*)(slot + (wave * 16 + i * 4) * 256), 16, 0, 0);
  };
  auto request = [&](int tj, int slot_i) {
#pragma unroll
    for (int i = 0; i < 4; ++i) request_piece(tj, slot_i, i);
  };

  GPSTAMP(0);
  // ---- the pair's own two panels -> ring slots 0, 1 -> A fragments in registers (they are never re-fetched per wave
  // from global memory: 64 KB per workgroup instead of 16 KB x 8 waves x 2)
  request(2 * p, 0);
  request(min(2 * p + 1, ntile - 1), 1);
  // the first two column panels go to slots 2 and 3 at once (step s uses slot (s + 2) % GP_RING): their latency runs
  // beside the A setup instead of after it
  int requested = 0;
#ifndef GP_NO_LOAD
  for (; requested < min(nsteps, 2); ++requested) request(tj0 + requested, (requested + 2) % GP_RING);
#else
  requested = min(nsteps, 2);
#endif
  if (requested == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (requested == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 fa[2][8];             // rows wr*64 + m*32 + r of the pair (panel q = wr >> 1), [m block][k step]
  {
    const int q = wr >> 1;
    const unsigned char* slot = s_ring + q * GP_PANEL_BYTES;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int prow = (wr & 1) * 64 + m * 32 + r;               // row inside the panel
      const bool ok = (2 * p + q < ntile) && ((2 * p + q) * GT + prow < D);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        bf16x8 v = *reinterpret_cast<const bf16x8*>(slot + prow * 256 + (((ks * 2 + h) ^ (prow & 15)) << 4));
        if (!ok) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0;
        }
        fa[m][ks] = v;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();        // every wave has its fragments: the slots can be reused

  GPSTAMP(1);
  // ---- stream the column panels
  const unsigned ring_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)s_ring;
#ifndef GP_NO_LOAD
  for (; requested < min(nsteps, 4); ++requested) request(tj0 + requested, (requested + 2) % GP_RING);   // slots 0, 1 are free now
#else
  requested = min(nsteps, 4);
#endif
  float tot = 0.f;
  const int ti = 2 * p + (wr >> 1);    // the row panel this wave's rows belong to
  const int rbase = (wr & 1) * 64;     // first row of this wave inside its row panel

  // The square-sum epilogue of step s runs INSIDE the MFMA stream of step s + 1 (two accumulator sets, ping-pong): eight
  // FMAs behind each group of four MFMAs, which the matrix pipe hides.  The per-step barrier starts both waves of a SIMD
  // on their MFMAs together, so without this the pipe idles while both run their epilogues.
  //   plain   : sum of squares of the previous tile's 64 values per lane (done inside the MFMA stream)
  //   finish  : what is left for the previous tile after the stream: weights, the diagonal of a diagonal tile
  //             (one element per lane and diagonal block), or the fully masked form for the last panel of a ragged D
  auto finish_prev = [&](const f32x16 (&accP)[2][2], float ssP, int tjP) {
    if (!(ti < ntile && tjP >= ti)) return;      // no such row panel / mirror image of a tile counted elsewhere
    float ss = ssP;
    if (ragged && tjP == ntile - 1) {
      ss = 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int col = wc * 64 + n * 32 + r;
          const bool col_ok = tjP * GT + col < D;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = rbase + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const float v = accP[m][n][e];
            if (col_ok && !(ti == tjP && row == col)) ss = fmaf(v, v, ss);
          }
        }
    } else if (ti == tjP) {
      // element (row, col) of block (m, n) lies on the diagonal iff rbase + m*32 + rowin == wc*64 + n*32 + r
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int rowin = wc * 64 + n * 32 + r - rbase - m * 32;   // row inside the 32 x 32 block, if in [0, 32)
          const int t = rowin - 4 * h;                                // = (e & 3) + 8 (e >> 2) for the lane's element e
          if (rowin >= 0 && rowin < 32 && t >= 0 && (t & 4) == 0) {
            const int esel = (t & 3) + 4 * (t >> 3);
            float v = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) v = (e == esel) ? accP[m][n][e] : v;
            ss = fmaf(-v, v, ss);
          }
        }
    }
    tot += (ti == tjP) ? ss : 2.0f * ss;
  };

  // Two steps per barrier (round 4).  With a barrier per step all eight waves start their 32 MFMAs together and finish
  // together, and the ~1,100 cycles between two streams (barrier, first fragment reads, the rest of the epilogue) leave
  // the matrix cores idle: in-kernel stamps had a step at 3,650 cycles for 2,050 of matrix work, with and without the
  // panel loads.  A barrier now covers a PAIR of steps: the fragments of the second panel are read while the first
  // one's MFMAs run.  Ring: the pair being read (two slots) + the pair landing (two slots); the landing pair is requested
  // in the FIRST step of the pair before (one 1 KB piece per k step), so it has the second step's duration to arrive.
  auto sync_pair = [&](int s, int npanels) {
    // this wave's loads of panels s .. s + npanels - 1 have landed once at most the loads of the later panels requested
    // so far are outstanding (vector memory operations retire in order); the barrier extends that to every wave's part
    // and frees the slots of the pair before
    const int later = requested - (s + npanels);
    if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  // one step: MFMAs of panel s into accC, plain square sums of accP (the previous step) in their shadow; `nreq` panels
  // (0, 1 or 2) are requested on the way, a piece per k step
  auto step = [&](int s, f32x16 (&accC)[2][2], const f32x16 (&accP)[2][2], bool haveP, int nreq) {
    GPSTAMP(2 + 2 * s);
    const int req_tj0 = tj0 + requested, req_slot0 = (requested + 2) % GP_RING;
    const int req_tj1 = req_tj0 + 1, req_slot1 = (requested + 3) % GP_RING;
    requested += nreq;

    // B fragments by inline-asm ds_read_b128: through the compiler's own LDS loads every step would first drain ALL
    // outstanding LDS-DMA (it cannot tell the slot being read from the slots being filled and inserts vmcnt(0)), which
    // serialises the ring.  The reads of k step ks + 1 are issued before the MFMAs of step ks and waited for after them.
    const unsigned slot_a = ring_lds + (unsigned)(((s + 2) % GP_RING) * GP_PANEL_BYTES);
    bf16x8 fb[2][2];    // [ks & 1][n]
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int brow = wc * 64 + n * 32 + r;
      gp_lds_read_b128(fb[0][n], slot_a + (unsigned)(brow * 256 + ((h ^ (brow & 15)) << 4)));
    }
    gp_lds_wait(fb[0][0], fb[0][1]);
    float ssP = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int brow = wc * 64 + n * 32 + r;
          gp_lds_read_b128(fb[(ks + 1) & 1][n], slot_a + (unsigned)(brow * 256 + ((((ks + 1) * 2 + h) ^ (brow & 15)) << 4)));
        }
      }
#ifndef GP_NO_LOAD
      if (ks < 4 ? nreq >= 1 : nreq >= 2) request_piece(ks < 4 ? req_tj0 : req_tj1, ks < 4 ? req_slot0 : req_slot1, ks & 3);
#endif
      __builtin_amdgcn_sched_barrier(0);   // keep this k step's MFMAs between the issue of the next reads and their wait
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#ifdef GP_NO_MFMA
          accC[m][n][ks] = (float)fb[ks & 1][n][0] + (float)fa[m][ks][1];
#else
          if (ks == 0) {
            f32x16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.f;
            accC[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][ks], fb[ks & 1][n], z, 0, 0, 0);
          } else {
            accC[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m][ks], fb[ks & 1][n], accC[m][n], 0, 0, 0);
          }
#endif
        }
      // eight of the previous tile's 64 squares: block (ks >> 1), elements 8 (ks & 1) .. + 7
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = accP[(ks >> 2) & 1][(ks >> 1) & 1][8 * (ks & 1) + e];
        ssP = fmaf(v, v, ssP);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 1 < 8) gp_lds_wait(fb[(ks + 1) & 1][0], fb[(ks + 1) & 1][1]);
    }
    GPSTAMP(3 + 2 * s);
    if (haveP) finish_prev(accP, ssP, tj0 + s - 1);
  };

  f32x16 acc0[2][2], acc1[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc0[m][n][e] = 0.f; acc1[m][n][e] = 0.f; }
  int s = 0;
  for (; s + 1 < nsteps; s += 2) {
    sync_pair(s, 2);
    // the slots of panels s - 2, s - 1 are free: panels s + 2, s + 3 go there (at s = 0 they were requested above)
    const int nreq = max(0, min(nsteps, s + 4) - requested);
    step(s, acc0, acc1, s > 0, nreq);
    step(s + 1, acc1, acc0, true, 0);
  }
  if (s < nsteps) {
    sync_pair(s, 1);
    step(s, acc0, acc1, s > 0, 0);
    // the last tile's squares, not hidden behind anything
    float ss = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) ss = fmaf(acc0[m][n][e], acc0[m][n][e], ss);
    finish_prev(acc0, ss, tj0 + nsteps - 1);
  } else {
    float ss = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 16; ++e) ss = fmaf(acc1[m][n][e], acc1[m][n][e], ss);
    finish_prev(acc1, ss, tj0 + nsteps - 1);
  }
  GPSTAMP(30);
  for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d, 64);
  if (lane == 0) s_part[wave] = (double)tot;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int w = 0; w < GP_THREADS / 64; ++w) t += s_part[w];
    (branch ? part_y : part_x)[item] = t;
  }
}

// ---- Gram for deep contractions (batch > 128: BASELINE configs[3], global batch 1024), 256 x 256 tiles --------------
// vicreg_gram_kernel above (128 x 128 tiles, operands staged through registers, ds_write_b128) is bound by the LDS store
// path: 8 ds_write_b128 per thread and 64-deep chunk = 832 LDS cycles per CU against 1024 MFMA cycles, on top of 512
// cycles of fragment reads (profiles/r03b_kstats_vicreg1024.csv: 0.29 of the bf16 peak).  Here:
//   * 256 x 256 tile per workgroup, 8 waves as 2 (rows) x 4 (columns), 128 x 64 per wave on v_mfma_f32_16x16x32_bf16
//     (32 accumulator tiles = 128 VGPRs): 24 fragment reads feed 64 MFMAs per wave and 64-deep K-tile (reads 37 % of the
//     MFMA time instead of 50 %, and half the staged bytes per flop);
//   * both operands' K-tiles (256 rows x 128 B each) come in by LDS-DMA (global_load_lds, 16 B per lane: no VGPRs, no
//     ds_write) into two 64 KB buffers, K-tile kt + 1 requested in pieces between the MFMA groups of K-tile kt;
//   * LDS image: row r at r * 128 B, its 16-byte granule g at slot g ^ ((r >> 1) & 7) -- lane-linear for the DMA (the
//     SOURCE address is permuted), conflict-free for the 16 x 32 fragment reads (ds_read_b128 lane groups);
//   * fragments by inline-asm ds_read_b128, double-buffered in registers: the reads of the next (k-step, row half) are
//     issued before the 16 MFMAs of the current one and waited for with a counted lgkmcnt; one raw barrier per K-tile.
// Tile (ti, tj), ti <= tj, of branch b per workgroup; partial = sum of squares of its off-diagonal elements (x 2 above
// the diagonal).  Rows beyond D repeat row D - 1 on the way in and are masked in the epilogue.
#define G2_T 256
#define G2_BK 64
#define G2_THREADS 512
#define G2_TILE_BYTES (G2_T * G2_BK * 2)
#define G2_BUF_BYTES (2 * G2_TILE_BYTES)

#define G2_WAIT(N, X, Y) do { asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]) : : "memory"); } while (0)

// acc[8][4] (16 x 16 tiles: rows wr*128 + 16 m + 4 (lane >> 4) + reg, columns wc*64 + 16 n + (lane & 15)) +=
// A[row0 .. +255][k] B[col0 .. +255][k]^T over k in [k0, k0 + 64 nkt), A / B bf16 row-major with row strides lda / ldb
// (elements; every offset < 2^31).  Rows >= arows / brows repeat the last valid row: mask them in the epilogue.
// s_g2: the workgroup's 128 KB staging area; on return every wave has passed the barrier behind the last K-tile.
__device__ __forceinline__ void g2_product(const unsigned short* __restrict__ A, unsigned lda, int arows, int row0,
                                           const unsigned short* __restrict__ Bm, unsigned ldb, int brows, int col0,
                                           int k0, int nkt, unsigned char* s_g2, vc_f32x4 (&acc)[8][4]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  // ---- LDS-DMA pieces: a K-tile is 64 pieces of 1 KB (8 rows x 128 B), piece = wave + 8 i; pieces 0..31 operand A
  unsigned src_off[8];            // element offset of this lane's 16 bytes of piece i at K-tile 0
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int piece = wave + 8 * i, pp = piece & 31;
    const int row = 8 * pp + (lane >> 3);
    const int g = (lane & 7) ^ ((row >> 1) & 7);
    int grow = (i >= 4 ? col0 : row0) + row;
    const int lim = i >= 4 ? brows : arows;
    grow = grow < lim ? grow : lim - 1;
    src_off[i] = (unsigned)grow * (i >= 4 ? ldb : lda) + (unsigned)(k0 + g * 8);
  }
  auto request = [&](int kt, int i) {
    const int piece = wave + 8 * i, pp = piece & 31;
    unsigned char* dst = s_g2 + (kt & 1) * G2_BUF_BYTES + (i >= 4 ? G2_TILE_BYTES : 0) + pp * 1024;
    __builtin_amdgcn_global_load_lds((gram_glb_void*)((i >= 4 ? Bm : A) + (size_t)src_off[i] + (size_t)kt * G2_BK),
                                     (gram_lds_void*)dst, 16, 0, 0);
  };
  // ---- fragment addresses
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)s_g2;
  const unsigned sw0 = (unsigned)((q ^ ((r >> 1) & 7)) << 4), sw1 = (unsigned)(((4 + q) ^ ((r >> 1) & 7)) << 4);
  const unsigned a_lane = (unsigned)((wr * 128 + r) * 128), b_lane = (unsigned)(G2_TILE_BYTES + (wc * 64 + r) * 128);

  // K-tile 0
#pragma unroll
  for (int i = 0; i < 8; ++i) request(0, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 fa[2][4], fb[2][4];
  for (int kt = 0; kt < nkt; ++kt) {
    const unsigned buf = lds0 + (unsigned)((kt & 1) * G2_BUF_BYTES);
    const bool more = kt + 1 < nkt;
    auto read_a = [&](bf16x8 (&dst)[4], int ks, int mq) {
#pragma unroll
      for (int m = 0; m < 4; ++m) gp_lds_read_b128(dst[m], buf + a_lane + (ks ? sw1 : sw0) + (unsigned)(mq * 8192 + m * 2048));
    };
    auto read_b = [&](bf16x8 (&dst)[4], int ks) {
#pragma unroll
      for (int n = 0; n < 4; ++n) gp_lds_read_b128(dst[n], buf + b_lane + (ks ? sw1 : sw0) + (unsigned)(n * 2048));
    };
    auto mfma16 = [&](const bf16x8 (&Af)[4], const bf16x8 (&Bf)[4], int mq) {
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[mq * 4 + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[m], Bf[n], acc[mq * 4 + m][n], 0, 0, 0);
    };
    read_a(fa[0], 0, 0);
    read_b(fb[0], 0);
    // unit 0: (k-step 0, rows 0..63 of the wave)
    read_a(fa[1], 0, 1);
    if (more) { request(kt + 1, 0); request(kt + 1, 1); request(kt + 1, 2); }
    G2_WAIT(4, fa[0], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma16(fa[0], fb[0], 0);
    __builtin_amdgcn_sched_barrier(0);
    // unit 1: (k-step 0, rows 64..127)
    read_a(fa[0], 1, 0);
    read_b(fb[1], 1);
    if (more) { request(kt + 1, 3); request(kt + 1, 4); request(kt + 1, 5); }
    G2_WAIT(8, fa[1], fb[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfma16(fa[1], fb[0], 1);
    __builtin_amdgcn_sched_barrier(0);
    // unit 2: (k-step 1, rows 0..63)
    read_a(fa[1], 1, 1);
    if (more) { request(kt + 1, 6); request(kt + 1, 7); }
    G2_WAIT(4, fa[0], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma16(fa[0], fb[1], 0);
    __builtin_amdgcn_sched_barrier(0);
    // unit 3: (k-step 1, rows 64..127)
    G2_WAIT(0, fa[1], fb[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfma16(fa[1], fb[1], 1);
    __builtin_amdgcn_sched_barrier(0);
    // K-tile kt + 1 has landed (this wave's pieces), every wave is done reading K-tile kt
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

__device__ __forceinline__ void g2_tri_item(int item, int ntile, int& ti, int& tj) {
  int t = item;
  ti = 0;
  int rowlen = ntile;
  while (t >= rowlen) { t -= rowlen; --rowlen; ++ti; }
  tj = ti + t;
}

__global__ __launch_bounds__(G2_THREADS, 2) void vicreg_gram256_kernel(const unsigned short* __restrict__ Xt_x,
                                                                       const unsigned short* __restrict__ Xt_y,
                                                                       double* __restrict__ part_x,
                                                                       double* __restrict__ part_y, int D, int Kpad,
                                                                       int ntile, int ntri) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_g2[];   // 2 buffers x (A 32 KB + B 32 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  // XCD placement as in vicreg_gram_pair_kernel: an XCD's L2 serves one branch
  const int xslot = blockIdx.x & 7, branch = xslot >> 2;
#ifndef G2_NO_STRIPS
  // Round 4: which tiles an XCD works on at the same time.  Workgroups go to the XCDs round-robin by block index; dealt
  // tile by tile along the rows of the triangle (rounds 2-3), the 32 tiles resident on an XCD were one row panel against
  // 32 column panels -- 33 panels of 512 KB through a 4 MB L2, 12.7 x the operands fetched from beyond it
  // (profiles/r03g_traffic.json).  Now an XCD takes CONTIGUOUS runs of 32 tiles of an enumeration that walks strips of
  // four tile rows column by column: its resident tiles are 4 row panels x 8 column panels (6 MB).
  const int q8 = blockIdx.x >> 3;                       // this XCD's q8-th workgroup
  const int item = (q8 >> 5) * 128 + (xslot & 3) * 32 + (q8 & 31);
  if (item >= ntri) return;
  int ti, tj;
  {
    int t = item, I = 0;
    for (;;) {                                          // strip I: tile rows 4 I .. 4 I + h - 1, columns 4 I .. ntile - 1
      const int h = min(4, ntile - 4 * I), w = ntile - 4 * I;
      const int cnt = h * w - h * (h - 1) / 2;          // the strip's tiles on or above the diagonal
      if (t < cnt) {
        // column-major inside the strip: column c (from the strip's first) has min(c + 1, h) tiles
        int c = 0;
        while (c < h - 1 && t >= c + 1) { t -= c + 1; ++c; }
        if (c == h - 1) { c += t / h; t -= (t / h) * h; }
        ti = 4 * I + t; tj = 4 * I + c;
        break;
      }
      t -= cnt; ++I;
    }
  }
  const unsigned short* Xt = branch ? Xt_y : Xt_x;
#else
  const int item = (blockIdx.x >> 3) * 4 + (xslot & 3);
  if (item >= ntri) return;
  const unsigned short* Xt = branch ? Xt_y : Xt_x;
  int ti, tj;
  g2_tri_item(item, ntile, ti, tj);
#endif
  vc_f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (vc_f32x4){0.f, 0.f, 0.f, 0.f};
  g2_product(Xt, (unsigned)Kpad, D, ti * G2_T, Xt, (unsigned)Kpad, D, tj * G2_T, 0, Kpad / G2_BK, s_g2, acc);

  // ---- epilogue: C/D layout of a 16 x 16 tile: col = lane & 15, row = 4 (lane >> 4) + reg
  float ss = 0.f;
  const bool diag = ti == tj, ragged = (D % G2_T) != 0 && (tj == ntile - 1);
  if (diag || ragged) {
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = wr * 128 + 16 * m + 4 * q + e, col = wc * 64 + 16 * n + r;
          const bool ok = ti * G2_T + row < D && tj * G2_T + col < D && !(diag && row == col);
          const float v = acc[m][n][e];
          if (ok) ss = fmaf(v, v, ss);
        }
  } else {
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) ss = fmaf(acc[m][n][e], acc[m][n][e], ss);
  }
  for (int d = 32; d > 0; d >>= 1) ss += __shfl_xor(ss, d, 64);
  double* s_part = reinterpret_cast<double*>(s_g2);       // the staging buffers are idle now (barrier above)
  if (lane == 0) s_part[wave] = (double)ss;
  __syncthreads();
  if (tid == 0) {
    double tot = 0.0;
    for (int w = 0; w < G2_THREADS / 64; ++w) tot += s_part[w];
    (branch ? part_y : part_x)[item] = diag ? tot : 2.0 * tot;
  }
}

// out[0..3] = loss, repr_loss, std_loss, cov_loss (fp32): fixed-order reduction of every partial, by 256 threads.
// gram_x (+ gram_y): partial sums of squares; diagpart (batch-side form, else null): [nmse], taken off.
struct VcFinish {
  const double *hingepart, *msepart, *gram_x, *gram_y, *diagpart;
  int nmse, ngram, B, D, cfg_batch;
  float sim_coeff, std_coeff, cov_coeff;
  float* out;
};
// (A variant in which the fold's last-arriving workgroup did this reduction -- ticket counter, device-scope loads -- cost
// more than the launch it saved: +0 us at B = 128, +6 us at B = 1024, HISTORY.md round 4.)
__global__ __launch_bounds__(256) void vicreg_finish_kernel(const VcFinish f) {
  __shared__ double s[256][4];
  double mse = 0.0, hinge = 0.0, gx = 0.0, gy = 0.0;
  for (int i = threadIdx.x; i < f.nmse; i += 256) mse += f.msepart[i];
  for (int i = threadIdx.x; i < f.ngram; i += 256) { gx += f.gram_x[i]; if (f.gram_y) gy += f.gram_y[i]; }
  if (f.diagpart)                                                // gram_x holds sum G_ab^2 = sum over ALL of C: take C's diagonal off
    for (int i = threadIdx.x; i < f.nmse; i += 256) gy -= f.diagpart[i];
  for (int i = threadIdx.x; i < f.nmse; i += 256) hinge += f.hingepart[i];
  s[threadIdx.x][0] = mse; s[threadIdx.x][1] = hinge; s[threadIdx.x][2] = gx; s[threadIdx.x][3] = gy;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d)
      for (int k = 0; k < 4; ++k) s[threadIdx.x][k] += s[threadIdx.x + d][k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double repr = s[0][0] / ((double)f.B * (double)f.D);
    const double stdl = s[0][1] / (2.0 * (double)f.D);
    const double den = (double)(f.cfg_batch - 1);
    const double cov = (s[0][2] + s[0][3]) / (den * den) / (double)f.D;
    f.out[0] = (float)((double)f.sim_coeff * repr + (double)f.std_coeff * stdl + (double)f.cov_coeff * cov);
    f.out[1] = (float)repr;
    f.out[2] = (float)stdl;
    f.out[3] = (float)cov;
  }
}
// ------------------------------------------------------------------------ C ABI
static inline size_t vc_align(size_t x) { return (x + 255) / 256 * 256; }
struct VicregWs { size_t xt_x, xt_y, colstats, mse, hinge, diag, gram_x, gram_y, xc_x, xc_y, bgram, bgram16, gdiag, g2part, total; int Kpad, ntile, ngram, nmse, nsplit, ksplit, ng2; bool t256, b256, dual; };
// IAS_VICREG_GRAM128=1 (diagnostics): the 128 x 128 register-staged kernels of round 2 for batch > 128 as well
static bool vicreg_force128() {
  static const bool v = ias_diag_env("IAS_VICREG_GRAM128") != nullptr && atoi(ias_diag_env("IAS_VICREG_GRAM128")) != 0;
  return v;
}
// The covariance term from the batch side.  sum_{i != j} C_ij^2 with C = Xc^T Xc (D x D) equals ||G||_F^2 - sum_j C_jj^2
// with G = Xc Xc^T (B x B): the same number from 2 B^2 D flops instead of 2 B D^2, and G is what the backward needs
// anyway (G vc).  Taken whenever the padded batch is no larger than the embedding (B = 128 / 1024 against D = 8192: 64x /
// 8x fewer flops); the feature-side kernels stay for batch > D.  No cancellation on this side of the switch: C has rank
// < B, so ||C||_F^2 >= (tr C)^2 / (B - 1) >= D / (B - 1) times its diagonal's share.
// Which side is taken is a pure function of the shape: the library keeps no state.  (Diagnostic library only:
// ias_vicreg_set_form(0 / 1 / -1) = feature side always / batch side where it applies / default, and the environment
// switch IAS_VICREG_DXD=1 -> 0: bench.py times the D x D kernels of rounds 1-3 through THAT library for `roofline.dxd`.)
#ifdef IAS_DIAG
static int g_vicreg_form = -1;
extern "C" int ias_vicreg_set_form(int form) {
  if (form < -1 || form > 1) return IAS_ERR_ARG;
  g_vicreg_form = form;
  return IAS_OK;
}
static bool vicreg_batch_side(int Kpad, int D) {
  static const bool env_dxd = ias_diag_env("IAS_VICREG_DXD") != nullptr && atoi(ias_diag_env("IAS_VICREG_DXD")) != 0;
  const bool want = g_vicreg_form < 0 ? !env_dxd : g_vicreg_form == 1;
  return want && D >= 8 && (D & 7) == 0 && Kpad <= D;
}
#else
static bool vicreg_batch_side(int Kpad, int D) { return D >= 8 && (D & 7) == 0 && Kpad <= D; }
#endif

static VicregWs vicreg_ws(int B, int D) {
  VicregWs w;
  w.Kpad = (B + GT - 1) / GT * GT;   // multiple of 128: the pair kernel's depth, the backward's batch tiles (zero padded)
  w.ntile = (D + GT - 1) / GT;
  w.ngram = w.ntile * (w.ntile + 1) / 2;
  w.nmse = (D + VC_COLS - 1) / VC_COLS;
  size_t o = 0;
  w.xt_x = o;     o = vc_align(o + sizeof(unsigned short) * (size_t)D * w.Kpad);
  w.xt_y = o;     o = vc_align(o + sizeof(unsigned short) * (size_t)D * w.Kpad);
  w.colstats = o; o = vc_align(o + sizeof(float) * 4 * (size_t)D);
  w.mse = o;      o = vc_align(o + sizeof(double) * w.nmse);
  w.hinge = o;    o = vc_align(o + sizeof(double) * w.nmse);
  w.diag = o;     o = vc_align(o + sizeof(double) * w.nmse);
  w.gram_x = o;   o = vc_align(o + sizeof(double) * w.ngram);
  w.gram_y = o;   o = vc_align(o + sizeof(double) * w.ngram);
  // backward: centred bf16 copies batch-major [Kpad][D], and the two B x B Grams [2][Kpad][Kpad] fp32
  w.xc_x = o;     o = vc_align(o + sizeof(unsigned short) * (size_t)D * w.Kpad);
  w.xc_y = o;     o = vc_align(o + sizeof(unsigned short) * (size_t)D * w.Kpad);
  // the B x B Gram is contracted over D in `nsplit` slices (enough workgroups for the chip at any batch size), each
  // slice writing its own partial Gram [nsplit][2][Kpad][Kpad]; vicreg_gconv_kernel adds them in slice order
  // batch > 128: the 256 x 256 LDS-DMA kernels (forward Gram; backward when D is a multiple of the 64-deep K-tile)
  w.t256 = w.Kpad > 128 && !vicreg_force128() && D >= 256 && (unsigned long long)D * (unsigned long long)w.Kpad < (1ull << 31);
  w.dual = vicreg_batch_side(w.Kpad, D);
  {
    const bool b256 = w.t256 && D % 64 == 0;
    w.b256 = b256;
    const int bt = b256 ? (w.Kpad + 255) / 256 : w.Kpad / GT, npair = bt * (bt + 1) / 2;
    int nsplit = (b256 ? 256 : 512) / (2 * npair);      // one resident round: 1 (256 tiles) / 2 (128 tiles) workgroups per CU
    if (nsplit < 1) nsplit = 1;
    if (nsplit > D / 256) nsplit = D / 256 > 0 ? D / 256 : 1;
    w.ksplit = ((D + nsplit - 1) / nsplit + GK - 1) / GK * GK;
    w.nsplit = (D + w.ksplit - 1) / w.ksplit;
  }
  w.bgram = o;    o = vc_align(o + sizeof(float) * 2 * (size_t)w.Kpad * w.Kpad * w.nsplit);
  w.bgram16 = o;  o = vc_align(o + sizeof(unsigned short) * 2 * (size_t)w.Kpad * w.Kpad);
  w.gdiag = o;    o = vc_align(o + sizeof(float) * 2 * (size_t)w.Kpad);
  {                                                              // partial sums of G_ab^2, one or two per fold workgroup
    const int bt2 = (w.Kpad + 255) / 256;
    size_t cgrid = (2 * (size_t)w.Kpad * w.Kpad + 255) / 256;
    if (cgrid > 2048) cgrid = 2048;
    w.ng2 = w.b256 ? 2 * 16 * (bt2 * (bt2 + 1) / 2) : 2 * (int)cgrid;
  }
  w.g2part = o;   o = vc_align(o + sizeof(double) * (size_t)w.ng2);
  w.total = o;
  return w;
}

extern "C" long long ias_vicreg_workspace_bytes(int B, int D) {
  if (B < 2 || D < 1) return IAS_ERR_ARG;
  return (long long)vicreg_ws(B, D).total;
}

// x, y [B,D] fp32 -> out[4] = (loss, repr_loss, std_loss, cov_loss).  cfg_batch = the configured batch
// size whose (cfg_batch - 1) divides the covariance (vicreg.py:47-48).  After the call the workspace
// holds colstats [4][D] (mean_x, mean_y, centred sum of squares x / y) at ias_vicreg_colstats_offset().
// stage < 0: the whole loss; 0: column pass (means, centred bf16 transposes, MSE / hinge partials); 1: the Gram
// kernel(s) on the matrix cores; 2: the final reduction.  Stages run on a workspace the earlier stages have filled
// (bench.py times stage 1 alone with HIP events for the MFMA roofline).
static void vicreg_launch_batch_gram(const VicregWs& w, char* ws, int D, hipStream_t stream, bool squares);

static int vicreg_stage_ld(int stage, const float* x, const float* y, long long ld, float* out, void* workspace,
                           long long workspace_bytes, int B, int D, int cfg_batch, float sim_coeff, float std_coeff,
                           float cov_coeff, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !y || !out || !workspace || B < 2 || D < 1 || cfg_batch < 2 || stage > 2 || ld < D) return IAS_ERR_ARG;
  const VicregWs w = vicreg_ws(B, D);
  if ((size_t)workspace_bytes < w.total) return IAS_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  unsigned short* xt_x = (unsigned short*)(ws + w.xt_x);
  unsigned short* xt_y = (unsigned short*)(ws + w.xt_y);
  float* colstats = (float*)(ws + w.colstats);
  double* mse = (double*)(ws + w.mse);
  double* hinge = (double*)(ws + w.hinge);
  double* gram_x = (double*)(ws + w.gram_x);
  double* gram_y = (double*)(ws + w.gram_y);
  if (stage < 0 || stage == 0)
  {
    auto launch = [&](auto kernel) {
      hipLaunchKernelGGL(kernel, dim3(w.nmse), dim3(VC_THREADS), 0, stream, x, y, xt_x, xt_y, colstats, mse, hinge,
                         (double*)(ws + w.diag), (unsigned short*)(ws + w.xc_x), (unsigned short*)(ws + w.xc_y), B, D, w.Kpad,
                         (size_t)ld);
    };
    static const bool reread = ias_diag_env("IAS_VICREG_COLSTATS_REREAD") != nullptr;   // (diagnostics: the two-read form at any batch)
    if (reread || w.Kpad > 1024) launch(vicreg_colstats_kernel<0>);
    else if (w.Kpad <= 128) launch(vicreg_colstats_kernel<4>);
    else if (w.Kpad <= 256) launch(vicreg_colstats_kernel<8>);
    else if (w.Kpad <= 512) launch(vicreg_colstats_kernel<16>);
    else launch(vicreg_colstats_kernel<32>);
  }
  VcFinish fin;
  fin.hingepart = hinge; fin.msepart = mse; fin.nmse = w.nmse;
  fin.B = B; fin.D = D; fin.cfg_batch = cfg_batch;
  fin.sim_coeff = sim_coeff; fin.std_coeff = std_coeff; fin.cov_coeff = cov_coeff;
  fin.out = out;
  if (w.dual) {
    // batch side: G = Xc Xc^T in D-slices, folded (and squared, summed) in slice order; the backward finds G in place
    if (w.Kpad % GT) return IAS_ERR_UNSUPPORTED;
    fin.gram_x = (const double*)(ws + w.g2part); fin.gram_y = nullptr; fin.ngram = w.ng2;
    fin.diagpart = (const double*)(ws + w.diag);
    if (stage < 0 || stage == 1) vicreg_launch_batch_gram(w, ws, D, stream, true);
    if (stage < 0 || stage == 2) hipLaunchKernelGGL(vicreg_finish_kernel, dim3(1), dim3(256), 0, stream, fin);
    return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
  }
  int ngram = w.ngram;
  int nitems = 0;
  // deep contractions: 256 x 256 tiles
  const int nt256 = (D + G2_T - 1) / G2_T;
  const bool use256 = w.t256;
  if (w.Kpad == 128) {
    nitems = gp_nitems(w.ntile);       // pair kernel: one workgroup per (branch, row-panel pair, run of column panels)
    ngram = nitems;
  } else if (use256) {
    ngram = nt256 * (nt256 + 1) / 2;
  }
  if (stage < 0 || stage == 1) {
    if (w.Kpad == 128) {
      const size_t lds = (size_t)GP_RING * GP_PANEL_BYTES;
      (void)hipFuncSetAttribute((const void*)vicreg_gram_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(vicreg_gram_pair_kernel, dim3(8 * ((nitems + 3) / 4)), dim3(GP_THREADS), lds, stream, xt_x, xt_y,
                         gram_x, gram_y, D, w.ntile, nitems);
    } else if (use256) {
      const size_t lds = 2 * (size_t)G2_BUF_BYTES;
      (void)hipFuncSetAttribute((const void*)vicreg_gram256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      // (q8 = block >> 3 runs over 32 per 128 items: see the kernel's enumeration)
      hipLaunchKernelGGL(vicreg_gram256_kernel, dim3(8 * 32 * ((ngram + 127) / 128)), dim3(G2_THREADS), lds, stream, xt_x, xt_y,
                         gram_x, gram_y, D, w.Kpad, nt256, ngram);
    } else {
      hipLaunchKernelGGL(vicreg_gram_kernel, dim3(w.ngram, 2), dim3(256), 0, stream, xt_x, xt_y, gram_x, gram_y, D, w.Kpad,
                         w.ntile);
    }
  }
  if (stage < 0 || stage == 2) {
    fin.gram_x = gram_x; fin.gram_y = gram_y; fin.ngram = ngram; fin.diagpart = nullptr;
    hipLaunchKernelGGL(vicreg_finish_kernel, dim3(1), dim3(256), 0, stream, fin);
  }
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

extern "C" int ias_vicreg_stage(int stage, const float* x, const float* y, float* out, void* workspace,
                                long long workspace_bytes, int B, int D, int cfg_batch, float sim_coeff, float std_coeff,
                                float cov_coeff, void* stream_) {
  return vicreg_stage_ld(stage, x, y, D, out, workspace, workspace_bytes, B, D, cfg_batch, sim_coeff, std_coeff, cov_coeff,
                         stream_);
}

extern "C" int ias_vicreg_loss(const float* x, const float* y, float* out, void* workspace, long long workspace_bytes,
                               int B, int D, int cfg_batch, float sim_coeff, float std_coeff, float cov_coeff,
                               void* stream_) {
  return vicreg_stage_ld(-1, x, y, D, out, workspace, workspace_bytes, B, D, cfg_batch, sim_coeff, std_coeff, cov_coeff,
                         stream_);
}

// The same loss on x, y that are column blocks of wider row-major matrices (row stride ld >= D floats): the gathered
// [W B_l, 2 D] buffer of FullGatherLayer(cat(x, y, dim=1)) is consumed in place, x = buf[:, :D], y = buf[:, D:].
extern "C" int ias_vicreg_loss_ld(const float* x, const float* y, long long ld, float* out, void* workspace,
                                  long long workspace_bytes, int B, int D, int cfg_batch, float sim_coeff, float std_coeff,
                                  float cov_coeff, void* stream_) {
  return vicreg_stage_ld(-1, x, y, ld, out, workspace, workspace_bytes, B, D, cfg_batch, sim_coeff, std_coeff, cov_coeff,
                         stream_);
}


// ------------------------------------------------------------------------ backward
// d loss / d x, d loss / d y of vicreg.py:35-58 in closed form, never forming a D x D matrix.  With vc = v - mean_b(v)
// (v = x or y), m2_j = sum_b vc_bj^2, s_j = sqrt(m2_j / (B - 1) + 1e-4), G = vc vc^T (B x B), and the incoming
// cotangents folded into a = g_loss sim + g_repr, b = g_loss std + g_std, c = g_loss cov + g_cov:
//   grad_v = +-a 2 (x - y) / (B D)  -  b [s_j < 1] vc / (2 D (B - 1) s_j)  +  c kappa (G vc - vc m2_j),
//   kappa = 4 / ((cfg_batch - 1)^2 D)          (the centring adjoint vanishes: every column of the sum is zero-mean)
// Kernels: vicreg_bgram_kernel  G = Xc Xc^T on the matrix cores (bf16 in, fp32 accumulate), split over D into slices that
//                               each write their own partial Gram (summed in slice order: deterministic);
//          vicreg_grad_kernel   G vc as a second bf16 MFMA product (A = G cast to bf16, B = Xt) for x and y at once,
//                               epilogue adds the elementwise terms from x, y and the column statistics, writes gx, gy.

// The cotangents of (loss, repr_loss, std_loss, cov_loss): one device float each, null = zero (autograd hands the unused
// outputs over as None; packing them into one buffer cost a concatenation kernel per backward).
struct VcCot { const float* g[4]; };
__device__ __forceinline__ void vc_cot_coefs(const VcCot& k, float sim_coeff, float std_coeff, float cov_coeff, float& ca,
                                             float& cb, float& cc) {
  const float gl = k.g[0] ? *k.g[0] : 0.0f;
  ca = gl * sim_coeff + (k.g[1] ? *k.g[1] : 0.0f);
  cb = gl * std_coeff + (k.g[2] ? *k.g[2] : 0.0f);
  cc = gl * cov_coeff + (k.g[3] ? *k.g[3] : 0.0f);
}

// acc += A[row0 .. row0+127][k] B[col0 .. col0+127][k]^T over k in [kbeg, kend) (kbeg a multiple of GK, kend of 8), A / B bf16 row-major
// with row strides lda / ldb; rows >= arows / brows and k >= kend read as zero.  The k loop is double-buffered as in vicreg_gram_kernel:
// the next 64-deep chunk of both panels is fetched into registers while the matrix cores work on the current one and
// written to the other LDS buffer afterwards (one barrier per chunk).
__device__ __forceinline__ void vc_tile_product(const unsigned short* __restrict__ A, size_t lda, int arows,
                                                const unsigned short* __restrict__ Bm, size_t ldb, int brows, int row0,
                                                int col0, int kbeg, int kend, unsigned short (*s_a)[GT][GLD],
                                                unsigned short (*s_b)[GT][GLD], f32x16 (&acc)[2][2]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  uint4 sta[4], stb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (tid >> 3) + 32 * i, ch = tid & 7;
      sta[i] = make_uint4(0, 0, 0, 0); stb[i] = make_uint4(0, 0, 0, 0);
      const bool kin = k0 + ch * 8 + 8 <= kend;            // (kend % 8 == 0; a last chunk shorter than GK reads zeros)
      if (kin && row0 + rr < arows) sta[i] = *reinterpret_cast<const uint4*>(A + (size_t)(row0 + rr) * lda + k0 + ch * 8);
      if (kin && col0 + rr < brows) stb[i] = *reinterpret_cast<const uint4*>(Bm + (size_t)(col0 + rr) * ldb + k0 + ch * 8);
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = (tid >> 3) + 32 * i, ch = tid & 7;
      *reinterpret_cast<uint4*>(&s_a[buf][rr][ch * 8]) = sta[i];
      *reinterpret_cast<uint4*>(&s_b[buf][rr][ch * 8]) = stb[i];
    }
  };
  __syncthreads();                         // the buffers may still be read by a previous product of this workgroup
  fetch(kbeg);
  commit(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += GK) {
    const bool more = k0 + GK < kend;
    if (more) fetch(k0 + GK);
#pragma unroll
    for (int ks = 0; ks < GK / 16; ++ks) {
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
        fa[m] = *reinterpret_cast<const bf16x8*>(&s_a[buf][wr * 64 + m * 32 + r][ks * 16 + h * 8]);
#pragma unroll
      for (int n = 0; n < 2; ++n)
        fb[n] = *reinterpret_cast<const bf16x8*>(&s_b[buf][wc * 64 + n * 32 + r][ks * 16 + h * 8]);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], fb[n], acc[m][n], 0, 0, 0);
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
}

// One upper-triangular 128 x 128 tile of one D-slice of G = Xc Xc^T per workgroup (grid: tiles x slices x branches).
// The slice's partial tile (and its mirror image) is WRITTEN to that slice's own Gram, once: no atomics, nothing to
// zero, and the sum over slices is taken in slice order by vicreg_gconv_kernel -- the backward is bit-reproducible
// (the reference trains with deterministic=True, /root/reference/pretrain.py:100).
__global__ __launch_bounds__(256, 2) void vicreg_bgram_kernel(const unsigned short* __restrict__ Xc_x,
                                                              const unsigned short* __restrict__ Xc_y, float* __restrict__ Gp,
                                                              int D, int Kpad, int ntile, int ksplit /* features per slice */) {
  __shared__ __attribute__((aligned(16))) unsigned short s_a[2][GT][GLD];
  __shared__ __attribute__((aligned(16))) unsigned short s_b[2][GT][GLD];
  int t = blockIdx.x, ti = 0;
  {
    int rowlen = ntile;
    while (t >= rowlen) { t -= rowlen; --rowlen; ++ti; }
  }
  const int tj = ti + t;
  const int branch = blockIdx.z;
  const unsigned short* Xc = branch ? Xc_y : Xc_x;
  float* Gb = Gp + ((size_t)blockIdx.y * 2 + branch) * Kpad * Kpad;
  const int kbeg = blockIdx.y * ksplit, kend = min(kbeg + ksplit, D);   // D % 8 == 0 is checked on the host
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
  const int row0 = ti * GT, col0 = tj * GT;
  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
  vc_tile_product(Xc, (size_t)D, Kpad, Xc, (size_t)D, Kpad, row0, col0, kbeg, kend, s_a, s_b, acc);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row0 + wr * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int col = col0 + wc * 64 + n * 32 + r;
        Gb[(size_t)row * Kpad + col] = acc[m][n][e];
        if (ti != tj) Gb[(size_t)col * Kpad + row] = acc[m][n][e];
      }
}

// G = sum over the D-slices of their partial Grams, in slice order (fp32, both branches) -> bf16 with the diagonal taken
// out (it stays in fp32: gdiag[branch][b] = G_bb).  The diagonal of G is 10-100x its off-diagonal entries and would
// carry its bf16 rounding (2^-9) straight into the dominant term G_bb vc_bj of the gradient; the epilogue of
// vicreg_grad_kernel adds that term in fp32 instead.
__global__ __launch_bounds__(256) void vicreg_gconv_kernel(const float* __restrict__ Gp, unsigned short* __restrict__ Gb,
                                                           float* __restrict__ gdiag, int Kpad, int nsplit,
                                                           double* __restrict__ g2part /* [gridDim.x][2] or null */) {
  __shared__ double s_sq[4][2];
  const size_t n = (size_t)2 * Kpad * Kpad;
  double sq[2] = {0.0, 0.0};                                      // sum of G_ab^2 per branch (the diagonal included)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t q = i / ((size_t)Kpad * Kpad), rem = i - q * (size_t)Kpad * Kpad;
    const int row = (int)(rem / Kpad), col = (int)(rem - (size_t)row * Kpad);
    float v = 0.0f;
    for (int s0 = 0; s0 < nsplit; s0 += 8) {                     // eight slices in flight, added in slice order
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = s0 + u < nsplit ? Gp[(size_t)(s0 + u) * n + i] : 0.0f;
#pragma unroll
      for (int u = 0; u < 8; ++u) v += a[u];
    }
    sq[q] += (double)v * (double)v;
    if (row == col) gdiag[q * Kpad + row] = v;
    Gb[i] = f2bf(row == col ? 0.0f : v);
  }
  if (g2part) {                                                  // fixed order: lanes (butterfly), then waves
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      for (int d = 32; d > 0; d >>= 1) sq[k] += __shfl_xor(sq[k], d, 64);
      if ((threadIdx.x & 63) == 0) s_sq[threadIdx.x >> 6][k] = sq[k];
    }
    __syncthreads();
    if (threadIdx.x < 2)
      g2part[2 * (size_t)blockIdx.x + threadIdx.x] =
          ((s_sq[0][threadIdx.x] + s_sq[1][threadIdx.x]) + s_sq[2][threadIdx.x]) + s_sq[3][threadIdx.x];
  }
}

// One 128 (batch rows) x 128 (features) tile of gx (branch 0) or gy (branch 1) per workgroup: the product (G - diag) vc
// through the double-buffered LDS panels, then the elementwise terms.  (Both branches in one workgroup, round 2, were 64
// workgroups at B = 128, D = 8192: a quarter of the chip.)
__global__ __launch_bounds__(256, 2) void vicreg_grad_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const unsigned short* __restrict__ Xt_x,
    const unsigned short* __restrict__ Xt_y, const unsigned short* __restrict__ Gb, const float* __restrict__ gdiag,
    const float* __restrict__ colstats,
    const VcCot gcoef /* g_loss, g_repr, g_std, g_cov */, float* __restrict__ gx, float* __restrict__ gy,
    int B, int D, int Kpad, int cfg_batch, float sim_coeff, float std_coeff, float cov_coeff, size_t ld, size_t ldg) {
  __shared__ __attribute__((aligned(16))) unsigned short s_a[2][GT][GLD];   // G rows (bf16)
  __shared__ __attribute__((aligned(16))) unsigned short s_b[2][GT][GLD];   // Xt rows (features)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int col0 = blockIdx.x * GT, row0 = blockIdx.y * GT, branch = blockIdx.z;   // one branch per workgroup (grid.z = 2)
  float ca, cb, cc;
  vc_cot_coefs(gcoef, sim_coeff, std_coeff, cov_coeff, ca, cb, cc);
  const float kappa = 4.0f / ((float)(cfg_batch - 1) * (float)(cfg_batch - 1) * (float)D);

  f32x16 acc[2][2];   // [m][n]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
  vc_tile_product(Gb + (size_t)branch * Kpad * Kpad, (size_t)Kpad, Kpad, branch ? Xt_y : Xt_x, (size_t)Kpad, D, row0, col0, 0,
                  Kpad, s_a, s_b, acc);

  // epilogue: elementwise terms.  C layout of a 32 x 32 block: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  const float inv_bm1 = 1.0f / (float)(B - 1);
  const float repr_k = (branch ? -ca : ca) * 2.0f / ((float)B * (float)D);
  float* gout = branch ? gy : gx;
  const float* own = branch ? y : x;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int j = col0 + wc * 64 + n * 32 + r;
    if (j >= D) continue;
    const float mu = colstats[(size_t)branch * D + j], m2 = colstats[(size_t)(2 + branch) * D + j];
    const float sd = sqrtf(m2 * inv_bm1 + 0.0001f);
    // coefficient of vc: variance hinge (active where s < 1) and the diagonal part of the covariance term
    const float av = (sd < 1.0f ? -cb / (2.0f * (float)D * (float)(B - 1) * sd) : 0.0f) - cc * kappa * m2;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int b = row0 + wr * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (b >= B) continue;
        const size_t idx = (size_t)b * ld + j;
        const float xv = x[idx], yv = y[idx];
        const float v = own[idx] - mu;
        gout[(size_t)b * ldg + j] = repr_k * (xv - yv) + av * v + cc * kappa * (acc[m][n][e] + gdiag[(size_t)branch * Kpad + b] * v);
      }
  }
}

// ---- batch <= 128 (BASELINE configs[2]): both branches of a 128 (batch) x 64 (features) tile in one workgroup ---------
// The whole contraction (K = 128) fits in LDS at once: G_x, G_y (2 x 32 KB bf16) and the two 64 x 128 slabs of Xt are
// loaded with 16-byte accesses, one barrier, 16 MFMAs per wave (8 waves: 4 row blocks x 2 column blocks, both branches),
// then the two accumulator tiles go through LDS so that the elementwise epilogue reads x and y ONCE for both branches
// and moves 16 bytes per access (256-byte row segments).  vicreg_grad_kernel (one branch per workgroup, 4-byte accesses
// to x, y read twice) took 27 us at B = 128, D = 8192; it stays as the fallback for unaligned views.
#define G1_COLS 64
#define G1_LDK 136        // bf16 row stride of the staged operands (272 B)
#define G1_LDC 72         // fp32 row stride of the staged result (4 rows apart = 32 banks apart: the two lane halves)
#define G1_LDS_BYTES ((2 * 128 + 2 * G1_COLS) * G1_LDK * 2)
static_assert(2 * 128 * G1_LDC * 4 <= G1_LDS_BYTES, "the result tiles overlay the operands");
__global__ __launch_bounds__(512) void vicreg_grad128_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const unsigned short* __restrict__ Xt_x,
    const unsigned short* __restrict__ Xt_y, const unsigned short* __restrict__ Gb, const float* __restrict__ gdiag,
    const float* __restrict__ colstats, const VcCot gcoef, float* __restrict__ gx, float* __restrict__ gy,
    int B, int D, int cfg_batch, float sim_coeff, float std_coeff, float cov_coeff, size_t ld, size_t ldg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_g1[];
  unsigned short (*s_a)[128][G1_LDK] = reinterpret_cast<unsigned short (*)[128][G1_LDK]>(s_g1);
  unsigned short (*s_b)[G1_COLS][G1_LDK] =
      reinterpret_cast<unsigned short (*)[G1_COLS][G1_LDK]>(s_g1 + 2 * 128 * G1_LDK * 2);
  float (*s_c)[128][G1_LDC] = reinterpret_cast<float (*)[128][G1_LDC]>(s_g1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * G1_COLS;
  // ---- operands -> LDS (rows of 128 bf16 = 16 chunks of 16 bytes)
  {
    const int ch = tid & 15, r0 = tid >> 4;                      // 32 rows per pass
#pragma unroll
    for (int br = 0; br < 2; ++br) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + 32 * i;
        *reinterpret_cast<uint4*>(&s_a[br][row][ch * 8]) =
            *reinterpret_cast<const uint4*>(Gb + ((size_t)br * 128 + row) * 128 + ch * 8);
      }
      const unsigned short* Xt = br ? Xt_y : Xt_x;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = r0 + 32 * i;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (j0 + row < D) v = *reinterpret_cast<const uint4*>(Xt + (size_t)(j0 + row) * 128 + ch * 8);
        *reinterpret_cast<uint4*>(&s_b[br][row][ch * 8]) = v;
      }
    }
  }
  __syncthreads();
  // ---- (G - diag) vc, both branches: wave -> rows 32 (wave & 3), columns 32 (wave >> 2)
  const int wr = wave & 3, wc = wave >> 2, r = lane & 31, h = lane >> 5;
  f32x16 acc[2];
#pragma unroll
  for (int br = 0; br < 2; ++br)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[br][e] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int br = 0; br < 2; ++br) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(&s_a[br][wr * 32 + r][ks * 16 + h * 8]);
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(&s_b[br][wc * 32 + r][ks * 16 + h * 8]);
      acc[br] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[br], 0, 0, 0);
    }
  __syncthreads();                                               // every wave is done with the operands
#pragma unroll
  for (int br = 0; br < 2; ++br)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      s_c[br][wr * 32 + (e & 3) + 8 * (e >> 2) + 4 * h][wc * 32 + r] = acc[br][e];
  __syncthreads();
  // ---- elementwise terms: thread -> 4 fixed columns, rows tid / 16 + 32 i
  float ca, cb, cc;
  vc_cot_coefs(gcoef, sim_coeff, std_coeff, cov_coeff, ca, cb, cc);
  const float kappa = 4.0f / ((float)(cfg_batch - 1) * (float)(cfg_batch - 1) * (float)D);
  const float inv_bm1 = 1.0f / (float)(B - 1);
  const float repr_k = ca * 2.0f / ((float)B * (float)D), cck = cc * kappa;
  const int c4 = tid & 15, jc = j0 + 4 * c4;
  if (jc >= D) return;                                           // D % 4 == 0: the four columns are in or out together
  float mu[2][4], av[2][4];
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    const vc_f32x4 m = *reinterpret_cast<const vc_f32x4*>(colstats + (size_t)br * D + jc);
    const vc_f32x4 m2 = *reinterpret_cast<const vc_f32x4*>(colstats + (size_t)(2 + br) * D + jc);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mu[br][c] = m[c];
      const float sd = sqrtf(m2[c] * inv_bm1 + 0.0001f);
      // coefficient of vc: variance hinge (active where s < 1) and the diagonal part of the covariance term
      av[br][c] = (sd < 1.0f ? -cb / (2.0f * (float)D * (float)(B - 1) * sd) : 0.0f) - cck * m2[c];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = (tid >> 4) + 32 * i;
    if (b >= B) continue;
    const vc_f32x4 xv = *reinterpret_cast<const vc_f32x4*>(x + (size_t)b * ld + jc);
    const vc_f32x4 yv = *reinterpret_cast<const vc_f32x4*>(y + (size_t)b * ld + jc);
    const vc_f32x4 cx = *reinterpret_cast<const vc_f32x4*>(&s_c[0][b][4 * c4]);
    const vc_f32x4 cy = *reinterpret_cast<const vc_f32x4*>(&s_c[1][b][4 * c4]);
    const float gdx = gdiag[b], gdy = gdiag[128 + b];
    vc_f32x4 ox, oy;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = repr_k * (xv[c] - yv[c]);
      const float vx = xv[c] - mu[0][c], vy = yv[c] - mu[1][c];
      ox[c] = d + av[0][c] * vx + cck * (cx[c] + gdx * vx);
      oy[c] = -d + av[1][c] * vy + cck * (cy[c] + gdy * vy);
    }
    *reinterpret_cast<vc_f32x4*>(gx + (size_t)b * ldg + jc) = ox;
    *reinterpret_cast<vc_f32x4*>(gy + (size_t)b * ldg + jc) = oy;
  }
}

// ---- the same two products for batch > 128 on the 256 x 256 LDS-DMA tile product (g2_product) -------------------------
// One upper-triangular 256 x 256 tile of one D-slice of G per workgroup (grid: tiles x slices x branches), written once
// to the slice's own Gram (fixed-order sum over the slices in vicreg_gconv256_kernel: deterministic).
__global__ __launch_bounds__(G2_THREADS, 2) void vicreg_bgram256_kernel(const unsigned short* __restrict__ Xc_x,
                                                                        const unsigned short* __restrict__ Xc_y,
                                                                        float* __restrict__ Gp, int D, int Kpad, int nt,
                                                                        int ksplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_g2[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3, r = lane & 15, q = lane >> 4;
  int ti, tj;
  g2_tri_item(blockIdx.x, nt, ti, tj);
  const int branch = blockIdx.z;
  const unsigned short* Xc = branch ? Xc_y : Xc_x;
  float* Gb = Gp + ((size_t)blockIdx.y * 2 + branch) * Kpad * Kpad;
  const int kbeg = blockIdx.y * ksplit, kend = min(kbeg + ksplit, D);     // D % 64 == 0, ksplit % 64 == 0 (host)
  vc_f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (vc_f32x4){0.f, 0.f, 0.f, 0.f};
  g2_product(Xc, (unsigned)D, Kpad, ti * G2_T, Xc, (unsigned)D, Kpad, tj * G2_T, kbeg, (kend - kbeg) / G2_BK, s_g2, acc);
  // rows of the tile through LDS (see vicreg_grad256_kernel): 16-byte stores, 1 KB per wave and row.  Only the upper
  // triangle of tiles is written; vicreg_gconv256_kernel mirrors it when it folds the slices.
  float* s_c = reinterpret_cast<float*>(s_g2);
  const int tid = threadIdx.x, c4 = tid & 63, j0 = tj * G2_T + 4 * c4;
  for (int half = 0; half < 2; ++half) {
    if (wr == half) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            s_c[(16 * m + 4 * q + e) * 256 + ((wc * 64 + 16 * n + r) ^ (q << 4))] = acc[m][n][e];
    }
    __syncthreads();
    for (int i = 0; i < 16; ++i) {
      const int rl = 8 * i + (tid >> 6);
      const int row = ti * G2_T + 128 * half + rl;
      if (row < Kpad && j0 < Kpad)                               // Kpad % 4 == 0: the four columns are in or out together
        *reinterpret_cast<vc_f32x4*>(Gb + (size_t)row * Kpad + j0) =
            *reinterpret_cast<const vc_f32x4*>(s_c + rl * 256 + ((4 * c4) ^ (((rl >> 2) & 3) << 4)));
    }
    __syncthreads();
  }
}

// G = sum over the D-slices of their partial Grams, in slice order, for the upper-triangular 256 x 256 tiles
// vicreg_bgram256_kernel wrote; one 64 x 64 block per workgroup: bf16 with the diagonal taken out (gdiag, see
// vicreg_gconv_kernel), written in place and -- off the diagonal tiles -- mirrored through an LDS transpose.
__global__ __launch_bounds__(256) void vicreg_gconv256_kernel(const float* __restrict__ Gp, unsigned short* __restrict__ Gb,
                                                              float* __restrict__ gdiag, int Kpad, int nsplit, int nt,
                                                              double* __restrict__ g2part /* [2][gridDim.x] or null */) {
  __shared__ float s_t[64][65];
  __shared__ double s_sq[4];
  int ti, tj;
  g2_tri_item(blockIdx.x >> 4, nt, ti, tj);
  const int sub = blockIdx.x & 15, branch = blockIdx.y;
  const int r0 = ti * G2_T + (sub >> 2) * 64, c0 = tj * G2_T + (sub & 3) * 64;
  const bool inside = r0 < Kpad && c0 < Kpad;                    // Kpad % 64 == 0: a block is inside or outside
  const size_t plane = (size_t)Kpad * Kpad, n2 = 2 * plane;
  const int tid = threadIdx.x, cc = tid & 63;
  double sq = 0.0;                                               // sum of G_ab^2 over the block (the diagonal included)
  if (inside)
    for (int rr = tid >> 6; rr < 64; rr += 4) {
      const int row = r0 + rr, col = c0 + cc;
      const size_t i = (size_t)branch * plane + (size_t)row * Kpad + col;
      float v = 0.0f;
      for (int s0 = 0; s0 < nsplit; s0 += 8) {                   // eight slices in flight, added in slice order
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = s0 + u < nsplit ? Gp[(size_t)(s0 + u) * n2 + i] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) v += a[u];
      }
      sq += (double)v * (double)v;
      if (row == col) { gdiag[(size_t)branch * Kpad + row] = v; v = 0.0f; }
      Gb[i] = f2bf(v);
      s_t[rr][cc] = v;
    }
  if (g2part) {
    for (int d = 32; d > 0; d >>= 1) sq += __shfl_xor(sq, d, 64);
    if ((tid & 63) == 0) s_sq[tid >> 6] = sq;
  }
  __syncthreads();
  if (g2part && tid == 0)                                        // tiles above the diagonal stand for their mirror image too
    g2part[(size_t)branch * gridDim.x + blockIdx.x] = (ti == tj ? 1.0 : 2.0) * (((s_sq[0] + s_sq[1]) + s_sq[2]) + s_sq[3]);
  if (inside && ti != tj)
    for (int rr = tid >> 6; rr < 64; rr += 4)                    // row c0 + rr of the mirror image, columns r0 + cc
      Gb[(size_t)branch * plane + (size_t)(c0 + rr) * Kpad + r0 + cc] = f2bf(s_t[cc][rr]);
}

// One 256 (batch rows) x 256 (features) tile of gx (branch 0) or gy (branch 1) per workgroup: (G - diag) vc on the
// matrix cores, then the elementwise terms (the epilogue of vicreg_grad_kernel, one branch).
__global__ __launch_bounds__(G2_THREADS, 2) void vicreg_grad256_kernel(
    const float* __restrict__ x, const float* __restrict__ y, const unsigned short* __restrict__ Xt_x,
    const unsigned short* __restrict__ Xt_y, const unsigned short* __restrict__ Gb, const float* __restrict__ gdiag,
    const float* __restrict__ colstats, const VcCot gcoef, float* __restrict__ gx, float* __restrict__ gy,
    int B, int D, int Kpad, int cfg_batch, float sim_coeff, float std_coeff, float cov_coeff, size_t ld, size_t ldg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_g2[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3, r = lane & 15, q = lane >> 4;
  const int col0 = blockIdx.x * G2_T, row0 = blockIdx.y * G2_T, branch = blockIdx.z;
  float ca, cb, cc;
  vc_cot_coefs(gcoef, sim_coeff, std_coeff, cov_coeff, ca, cb, cc);
  const float kappa = 4.0f / ((float)(cfg_batch - 1) * (float)(cfg_batch - 1) * (float)D);
  vc_f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = (vc_f32x4){0.f, 0.f, 0.f, 0.f};
  g2_product(Gb + (size_t)branch * Kpad * Kpad, (unsigned)Kpad, Kpad, row0, branch ? Xt_y : Xt_x, (unsigned)Kpad, D, col0, 0,
             Kpad / G2_BK, s_g2, acc);
  // ---- epilogue through LDS (the staging area is idle: every wave has passed the product's last barrier): the tile goes
  // to LDS half by half (128 rows x 256 fp32 = 128 KB; 16-column groups XOR-swizzled by the lane's row quarter, so the
  // accumulator stores are conflict-free), then every thread owns 4 fixed columns and walks the rows with 16-byte
  // accesses to x, y and the gradient (1 KB per wave and row) instead of 64-byte fragments of four rows each.
  const float inv_bm1 = 1.0f / (float)(B - 1);
  const float repr_k = (branch ? -ca : ca) * 2.0f / ((float)B * (float)D);
  float* gout = branch ? gy : gx;
  const float* own = branch ? y : x;
  float* s_c = reinterpret_cast<float*>(s_g2);
  const int tid = threadIdx.x, c4 = tid & 63, j0 = col0 + 4 * c4;
  float mu[4], av[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = j0 + c;
    mu[c] = 0.f; av[c] = 0.f;
    if (j < D) {
      mu[c] = colstats[(size_t)branch * D + j];
      const float m2 = colstats[(size_t)(2 + branch) * D + j];
      const float sd = sqrtf(m2 * inv_bm1 + 0.0001f);
      // coefficient of vc: variance hinge (active where s < 1) and the diagonal part of the covariance term
      av[c] = (sd < 1.0f ? -cb / (2.0f * (float)D * (float)(B - 1) * sd) : 0.0f) - cc * kappa * m2;
    }
  }
  const bool vec = (D & 3) == 0 && ((ld | ldg) & 3) == 0 && j0 + 3 < D;   // (16-byte aligned bases: checked by the caller)
  for (int half = 0; half < 2; ++half) {
    if (wr == half) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            s_c[(16 * m + 4 * q + e) * 256 + ((wc * 64 + 16 * n + r) ^ (q << 4))] = acc[m][n][e];
    }
    __syncthreads();
    for (int i = 0; i < 16; ++i) {
      const int rl = 8 * i + (tid >> 6);                       // row inside the half
      const int b = row0 + 128 * half + rl;
      if (b >= B) continue;
      const vc_f32x4 cv = *reinterpret_cast<const vc_f32x4*>(s_c + rl * 256 + ((4 * c4) ^ (((rl >> 2) & 3) << 4)));
      const float gd = gdiag[(size_t)branch * Kpad + b];
      const size_t idx = (size_t)b * ld + j0, odx = (size_t)b * ldg + j0;
      if (vec) {
        const vc_f32x4 xv = *reinterpret_cast<const vc_f32x4*>(x + idx), yv = *reinterpret_cast<const vc_f32x4*>(y + idx);
        vc_f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float v = (branch ? yv[c] : xv[c]) - mu[c];
          o[c] = repr_k * (xv[c] - yv[c]) + av[c] * v + cc * kappa * (cv[c] + gd * v);
        }
        *reinterpret_cast<vc_f32x4*>(gout + odx) = o;
      } else {
        for (int c = 0; c < 4; ++c) {
          if (j0 + c >= D) break;
          const float v = own[idx + c] - mu[c];
          gout[odx + c] = repr_k * (x[idx + c] - y[idx + c]) + av[c] * v + cc * kappa * (cv[c] + gd * v);
        }
      }
    }
    __syncthreads();
  }
}

// G = Xc Xc^T for both branches: partial Grams per D-slice [nsplit][2][Kpad][Kpad] (every word written), folded in slice
// order to bf16 (diagonal apart, fp32); `squares` (forward): the fold also leaves the partial sums of G_ab^2 at w.g2part.
static void vicreg_launch_batch_gram(const VicregWs& w, char* ws, int D, hipStream_t stream, bool squares) {
  float* G = (float*)(ws + w.bgram);
  const int bt = w.Kpad / GT, npair = bt * (bt + 1) / 2;
  const size_t lds256 = 2 * (size_t)G2_BUF_BYTES;
  const int bt2 = (w.Kpad + G2_T - 1) / G2_T;
  unsigned short* Gb = (unsigned short*)(ws + w.bgram16);
  float* gdiag = (float*)(ws + w.gdiag);
  double* g2part = squares ? (double*)(ws + w.g2part) : nullptr;
  if (w.b256) {
    (void)hipFuncSetAttribute((const void*)vicreg_bgram256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
    hipLaunchKernelGGL(vicreg_bgram256_kernel, dim3(bt2 * (bt2 + 1) / 2, w.nsplit, 2), dim3(G2_THREADS), lds256, stream,
                       (const unsigned short*)(ws + w.xc_x), (const unsigned short*)(ws + w.xc_y), G, D, w.Kpad, bt2, w.ksplit);
    hipLaunchKernelGGL(vicreg_gconv256_kernel, dim3(16 * (bt2 * (bt2 + 1) / 2), 2), dim3(256), 0, stream, G, Gb, gdiag, w.Kpad,
                       w.nsplit, bt2, g2part);
  } else {
    hipLaunchKernelGGL(vicreg_bgram_kernel, dim3(npair, w.nsplit, 2), dim3(256), 0, stream, (const unsigned short*)(ws + w.xc_x),
                       (const unsigned short*)(ws + w.xc_y), G, D, w.Kpad, bt, w.ksplit);
    hipLaunchKernelGGL(vicreg_gconv_kernel, dim3(w.ng2 / 2), dim3(256), 0, stream, G, Gb, gdiag, w.Kpad, w.nsplit, g2part);
  }
}

// Backward of ias_vicreg_loss on the SAME workspace (it must still hold the forward's column statistics and centred
// bf16 copies -- and, after a batch-side forward, G): gcoef [4] device floats = the cotangents of (loss, repr_loss, std_loss, cov_loss) -> gx, gy [B,D] fp32.
static int vicreg_backward_ld(const float* x, const float* y, long long ld, const VcCot gcoef, float* gx, float* gy,
                              long long ldg, void* workspace, long long workspace_bytes, int B, int D, int cfg_batch,
                              float sim_coeff, float std_coeff, float cov_coeff, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !y || !gx || !gy || !workspace || B < 2 || D < 8 || (D & 7) || cfg_batch < 2 || ld < D || ldg < D)
    return IAS_ERR_ARG;
  // the 256-tile epilogue moves 16 bytes per access: strided views must keep that alignment
  if ((ld != D || ldg != D) && ((((size_t)x | (size_t)y | (size_t)gx | (size_t)gy) & 15) || ((ld | ldg) & 3))) return IAS_ERR_ARG;
  const VicregWs w = vicreg_ws(B, D);
  if ((size_t)workspace_bytes < w.total) return IAS_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  const int bt = w.Kpad / GT;
  if (w.Kpad % GT) return IAS_ERR_UNSUPPORTED;
  const bool b256 = w.b256;
  const size_t lds256 = 2 * (size_t)G2_BUF_BYTES;
  const int bt2 = (w.Kpad + G2_T - 1) / G2_T;
  if (!w.dual) vicreg_launch_batch_gram(w, ws, D, stream, false);   // (batch-side forward: G and its diagonal are in place)
  unsigned short* Gb = (unsigned short*)(ws + w.bgram16);
  float* gdiag = (float*)(ws + w.gdiag);
  if (b256) {
    (void)hipFuncSetAttribute((const void*)vicreg_grad256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256);
    hipLaunchKernelGGL(vicreg_grad256_kernel, dim3((D + G2_T - 1) / G2_T, bt2, 2), dim3(G2_THREADS), lds256, stream, x, y,
                       (const unsigned short*)(ws + w.xt_x), (const unsigned short*)(ws + w.xt_y), Gb, gdiag,
                       (const float*)(ws + w.colstats), gcoef, gx, gy, B, D, w.Kpad, cfg_batch, sim_coeff, std_coeff, cov_coeff,
                       (size_t)ld, (size_t)ldg);
  } else if (w.Kpad == 128 && !((((size_t)x | (size_t)y | (size_t)gx | (size_t)gy) & 15) || ((ld | ldg) & 3))) {
    (void)hipFuncSetAttribute((const void*)vicreg_grad128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G1_LDS_BYTES);
    hipLaunchKernelGGL(vicreg_grad128_kernel, dim3((D + G1_COLS - 1) / G1_COLS), dim3(512), G1_LDS_BYTES, stream, x, y,
                       (const unsigned short*)(ws + w.xt_x), (const unsigned short*)(ws + w.xt_y), Gb, gdiag,
                       (const float*)(ws + w.colstats), gcoef, gx, gy, B, D, cfg_batch, sim_coeff, std_coeff, cov_coeff,
                       (size_t)ld, (size_t)ldg);
  } else {
    hipLaunchKernelGGL(vicreg_grad_kernel, dim3((D + GT - 1) / GT, bt, 2), dim3(256), 0, stream, x, y,
                       (const unsigned short*)(ws + w.xt_x), (const unsigned short*)(ws + w.xt_y), Gb, gdiag,
                       (const float*)(ws + w.colstats), gcoef, gx, gy, B, D, w.Kpad, cfg_batch, sim_coeff, std_coeff, cov_coeff,
                       (size_t)ld, (size_t)ldg);
  }
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

extern "C" int ias_vicreg_backward(const float* x, const float* y, const float* gcoef, float* gx, float* gy, void* workspace,
                                   long long workspace_bytes, int B, int D, int cfg_batch, float sim_coeff, float std_coeff,
                                   float cov_coeff, void* stream_) {
  if (!gcoef) return IAS_ERR_ARG;
  return vicreg_backward_ld(x, y, D, VcCot{{gcoef, gcoef + 1, gcoef + 2, gcoef + 3}}, gx, gy, D, workspace, workspace_bytes, B, D, cfg_batch, sim_coeff, std_coeff,
                            cov_coeff, stream_);
}

// Backward of ias_vicreg_loss_ld: x, y with row stride ld, gx, gy WRITTEN with row stride ldg (the two column blocks of
// the [W B_l, 2 D] cotangent that FullGatherLayer's backward reduce-scatters: no split / cat copies either way).
extern "C" int ias_vicreg_backward_ld(const float* x, const float* y, long long ld, const float* gcoef, float* gx, float* gy,
                                      long long ldg, void* workspace, long long workspace_bytes, int B, int D, int cfg_batch,
                                      float sim_coeff, float std_coeff, float cov_coeff, void* stream_) {
  if (!gcoef) return IAS_ERR_ARG;
  return vicreg_backward_ld(x, y, ld, VcCot{{gcoef, gcoef + 1, gcoef + 2, gcoef + 3}}, gx, gy, ldg, workspace, workspace_bytes, B, D, cfg_batch, sim_coeff, std_coeff,
                            cov_coeff, stream_);
}

// The same backward with the four cotangents as separate device floats, any of them null (= zero): what autograd hands
// over when only some of the four outputs were differentiated -- no packing kernel in front.
extern "C" int ias_vicreg_backward4_ld(const float* x, const float* y, long long ld, const float* g_loss, const float* g_repr,
                                       const float* g_std, const float* g_cov, float* gx, float* gy, long long ldg,
                                       void* workspace, long long workspace_bytes, int B, int D, int cfg_batch,
                                       float sim_coeff, float std_coeff, float cov_coeff, void* stream_) {
  return vicreg_backward_ld(x, y, ld, VcCot{{g_loss, g_repr, g_std, g_cov}}, gx, gy, ldg, workspace, workspace_bytes, B, D,
                            cfg_batch, sim_coeff, std_coeff, cov_coeff, stream_);
}

extern "C" long long ias_vicreg_colstats_offset(int B, int D) {
  if (B < 2 || D < 1) return IAS_ERR_ARG;
  return (long long)vicreg_ws(B, D).colstats;
}
