// 1x1 convolutions of the MobileNetV3 trunk on fp32 MFMA (MI355X / gfx950), NCHW.
//
// The reference takes these layers from torchvision (`mobilenet_v3_small(...).features`,
// /root/reference/vicreg_audio_params.py:52-54, run at audioembed.py:61).  Per sample they are GEMMs of a 16..288-row
// weight with a [C, H W] activation: as 128-batch rocBLAS / hipBLASLt `bmm`s (vision.PointwiseConv2d's other form)
// the sixteen thin ones cost 1.8 ms of a batch-128 training step, 5-10x their HBM traffic.  Here the weight lives in
// LDS in MFMA operand order, the activation goes from global memory straight into the B operand (a lane reads VEC
// consecutive positions of its input row), and the output leaves as 16-byte stores:
//   forward / input gradient : y[b] = A x[b],  A = W [M,K] or W^T          (pw_apply_kernel)
//   weight gradient          : gW = sum_b g[b] x[b]^T as per-(sample, split) partials + a fixed-order reduction
//                              (pw_wgrad_kernel, conv_reduce-style second launch; deterministic)
// v_mfma_f32_16x16x4_f32: A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15],
// D[row = 4 (lane >> 4) + r][col = lane & 15], r = 0..3.
#include "ias_common.h"

#define PW_THREADS 256
#define PW_CG 4          // output-row tiles (of 16) a wave accumulates at once: 4 x 4 x 4 = 64 accumulator registers
#define PW_NTI 15        // input-row tiles of the weight gradient: K <= 240
#define PW_LDS_FLOATS 16384

typedef float pw_v4f __attribute__((ext_vector_type(4)));

// In-kernel stamps (diagnostic build -DIAS_PW_STAMPS only, scripts/diag/pw_stamps.py; the product build compiles none of
// this): s_memrealtime (100 MHz) at the phase boundaries of pw_apply_kernel, per wave: [wave][8].
#ifdef IAS_PW_STAMPS
static __device__ unsigned long long* g_pw_stamps = nullptr;
extern "C" int ias_pw_set_stamps(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pw_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define PW_STAMP(id)                                                                                                    \
  do {                                                                                                                  \
    if (g_pw_stamps != nullptr && (threadIdx.x & 63) == 0)                                                              \
      g_pw_stamps[((size_t)blockIdx.x * (PW_THREADS / 64) + (threadIdx.x >> 6)) * 8 + (id)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define PW_STAMP(id) do {} while (0)
#endif

// Row stride of the staged weight image [T KS][PW_WROW]: 64 operand floats + 1.  With rows of exactly 64 floats the
// staging scatter -- consecutive lanes hold consecutive 16-byte groups of a weight row, i.e. consecutive k-steps of the
// same output row -- put all 64 lanes of a store on ONE bank: 6.8 us of a 23 us wave on the 240 -> 40 layer
// (scripts/diag/pw_stamps.py).  One float of padding moves consecutive k-steps to consecutive banks; the operand reads
// (64 consecutive floats of a row) are conflict-free either way.
#define PW_WROW 65

// positions of a 64-position block as (N-tile j, column n): VEC 4: 4 n + j; VEC 2: 32 (j >> 1) + 2 n + (j & 1);
// VEC 1: 16 j + n -- so that a lane's VEC tiles are consecutive positions (one load / store).
template <int VEC>
__device__ __forceinline__ void pw_load_row64(const float* __restrict__ row, int p0, int n, int HW, float (&v)[4]) {
  if (VEC == 4) {
    const int p = p0 + 4 * n;
    pw_v4f t = {0.0f, 0.0f, 0.0f, 0.0f};
    if (p < HW) t = *reinterpret_cast<const pw_v4f*>(row + p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else if (VEC == 2) {
    typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = p0 + 32 * h + 2 * n;
      v2f t = {0.0f, 0.0f};
      if (p < HW) t = *reinterpret_cast<const v2f*>(row + p);
      v[2 * h] = t[0]; v[2 * h + 1] = t[1];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = p0 + 16 * j + n;
      v[j] = p < HW ? row[p] : 0.0f;
    }
  }
}

template <int VEC>
__device__ __forceinline__ void pw_store_row64(float* __restrict__ row, int p0, int n, int HW, const float (&v)[4]) {
  if (VEC == 4) {
    const int p = p0 + 4 * n;
    if (p < HW) { pw_v4f t = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<pw_v4f*>(row + p) = t; }
  } else if (VEC == 2) {
    typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = p0 + 32 * h + 2 * n;
      if (p < HW) { v2f t = {v[2 * h], v[2 * h + 1]}; *reinterpret_cast<v2f*>(row + p) = t; }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = p0 + 16 * j + n;
      if (p < HW) row[p] = v[j];
    }
  }
}

// y[b][m][p] = sum_k A[m][k] x[b][k][p];  A[m][k] = transpose ? w[k M + m] : w[m K + k];  K % 4 == 0.
// A wave takes items (sample, 64-position block, group of <= PW_CG row tiles) in a grid-stride loop; the workgroup's
// four waves share the weight, staged once.  The input rows of PW_PF k-steps are requested ahead of the MFMAs that
// use them (a k-step is only 4 x tiles MFMAs, ~0.2 us: a distance of one step left the wave waiting on every load).
// SCALED: y[b] = A (x[b] * xs[b]) with a scale per (sample, input row) -- the squeeze-excitation gate of an
// inverted-residual block taken on load by the projection behind it (one multiply per loaded value beside 4 x tiles
// MFMAs per k-step) instead of a pass of its own over the expanded map; the scales travel with the rows' prefetch.
#define PW_PF 8
template <int VEC, bool SCALED>
__global__ __launch_bounds__(PW_THREADS) void pw_apply_kernel(const float* __restrict__ x, const float* __restrict__ xs,
                                                              const float* __restrict__ w,
                                                              float* __restrict__ y, int B, int M, int K, int HW,
                                                              int transpose, int blocks, int ngroups, int cg, long long items) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];          // [T][KS][PW_WROW]: the A operand of (tile, k-step)
  const int T = (M + 15) >> 4, KS = K >> 2, tid = threadIdx.x;
  PW_STAMP(0);
  // the weight, read linearly in 16-byte groups and scattered into operand order: element (m, k) goes to
  // [(m >> 4) KS + (k >> 2)][16 (k & 3) + (m & 15)]; the rows past M of the last tile are zero
  if (M & 15) {
    const int t = T - 1, pad0 = M & 15;
    for (int i = tid; i < KS * 64; i += PW_THREADS)
      if ((i & 15) >= pad0) s_w[(t * KS + (i >> 6)) * PW_WROW + (i & 63)] = 0.0f;
  }
  const pw_v4f* w4 = reinterpret_cast<const pw_v4f*>(w);
#pragma unroll 4
  for (int i = tid; i < (M * K) >> 2; i += PW_THREADS) {
    const pw_v4f v = w4[i];
    if (transpose) {                       // w[k][m .. m+3]: four neighbours in operand order
      const int k = (4 * i) / M, m = 4 * i - k * M;
      float* d = s_w + ((m >> 4) * KS + (k >> 2)) * PW_WROW + 16 * (k & 3) + (m & 15);
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    } else {                               // w[m][k .. k+3]: the four k-slots of one k-step
      const int m = (4 * i) / K, k = 4 * i - m * K;
      float* d = s_w + ((m >> 4) * KS + (k >> 2)) * PW_WROW + (m & 15);
      d[0] = v[0]; d[16] = v[1]; d[32] = v[2]; d[48] = v[3];
    }
  }
  __syncthreads();
  PW_STAMP(1);
  const int lane = tid & 63, n = lane & 15, q = lane >> 4;
  for (long long id = (long long)blockIdx.x * (PW_THREADS / 64) + (tid >> 6); id < items;
       id += (long long)gridDim.x * (PW_THREADS / 64)) {
    const int g = (int)(id % ngroups);
    const long long rest = id / ngroups;
    const int pb = (int)(rest % blocks);
    const long long b = rest / blocks;
    const int t0 = g * cg, tiles = min(cg, T - t0), p0 = pb * 64;
    pw_v4f acc[PW_CG][4];
#pragma unroll
    for (int i = 0; i < PW_CG; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = pw_v4f{0.0f, 0.0f, 0.0f, 0.0f};
    const float* xb = x + ((size_t)b * K + q) * HW;                     // row 4 ks + q: advance by 4 HW per k-step
    const float* sa = s_w + (size_t)t0 * KS * PW_WROW + lane;
    const float* sb = SCALED ? xs + (size_t)b * K + q : nullptr;        // scale of row 4 ks + q
    float cur[PW_PF][4], nxt[PW_PF][4];
    float cur_s[PW_PF], nxt_s[PW_PF];
#pragma unroll
    for (int d = 0; d < PW_PF; ++d) {
#pragma unroll
      for (int j = 0; j < 4; ++j) cur[d][j] = nxt[d][j] = 0.0f;
      cur_s[d] = nxt_s[d] = 1.0f;
      if (d < KS) {
        pw_load_row64<VEC>(xb + (size_t)d * 4 * HW, p0, n, HW, cur[d]);
        if (SCALED) cur_s[d] = sb[4 * d];
      }
    }
    PW_STAMP(2);
    for (int ks0 = 0; ks0 < KS; ks0 += PW_PF) {
#pragma unroll
      for (int d = 0; d < PW_PF; ++d)
        if (ks0 + PW_PF + d < KS) {
          pw_load_row64<VEC>(xb + (size_t)(ks0 + PW_PF + d) * 4 * HW, p0, n, HW, nxt[d]);
          if (SCALED) nxt_s[d] = sb[4 * (ks0 + PW_PF + d)];
        }
#pragma unroll
      for (int d = 0; d < PW_PF; ++d) {
        if (ks0 + d < KS) {
          if (SCALED) {
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[d][j] *= cur_s[d];
          }
#pragma unroll
          for (int i = 0; i < PW_CG; ++i) {
            if (i < tiles) {
              const float a = sa[(i * KS + ks0 + d) * PW_WROW];
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, cur[d][j], acc[i][j], 0, 0, 0);
            }
          }
        }
      }
#pragma unroll
      for (int d = 0; d < PW_PF; ++d) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[d][j] = nxt[d][j];
        if (SCALED) cur_s[d] = nxt_s[d];
      }
    }
    PW_STAMP(3);
#pragma unroll
    for (int i = 0; i < PW_CG; ++i) {
      if (i < tiles) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * (t0 + i) + 4 * q + r;
          if (m < M) {
            const float v[4] = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
            pw_store_row64<VEC>(y + ((size_t)b * M + m) * HW, p0, n, HW, v);
          }
        }
      }
    }
    PW_STAMP(4);
  }
}

// k-slot (q, s) of a 64-position chunk <-> position: VEC 4: 16 (s >> 2) + 4 q + (s & 3); VEC 2: 8 (s >> 1) + 2 q + (s & 1);
// VEC 1: 4 s + q -- a lane's VEC consecutive slots are consecutive positions, the four q of a load are neighbours.
template <int VEC>
__device__ __forceinline__ void pw_load_slots(const float* __restrict__ row, bool row_ok, int p0, int q, int HW, float (&v)[16]) {
  if (VEC == 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = p0 + 16 * j + 4 * q;
      pw_v4f t = {0.0f, 0.0f, 0.0f, 0.0f};
      if (row_ok && p < HW) t = *reinterpret_cast<const pw_v4f*>(row + p);
      v[4 * j] = t[0]; v[4 * j + 1] = t[1]; v[4 * j + 2] = t[2]; v[4 * j + 3] = t[3];
    }
  } else if (VEC == 2) {
    typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = p0 + 8 * j + 2 * q;
      v2f t = {0.0f, 0.0f};
      if (row_ok && p < HW) t = *reinterpret_cast<const v2f*>(row + p);
      v[2 * j] = t[0]; v[2 * j + 1] = t[1];
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int p = p0 + 4 * s + q;
      v[s] = (row_ok && p < HW) ? row[p] : 0.0f;
    }
  }
}

// partial[(b, split)][m][k] = sum over the split's positions of g[b][m][p] x[b][k][p]; one wave per (b, split, 16-row
// tile of g), all <= PW_NTI input tiles accumulated at once.  The input rows of a (chunk, tile) step are requested two
// steps (32 MFMAs) ahead, the g rows of the next chunk one chunk ahead.
// SCALED: the gradient of W in y[b] = W (x[b] * xs[b]): column k of a (sample, split) unit's sums times xs[b][k], applied
// once where the unit's sums leave the accumulators.
template <int VEC, bool SCALED>
__global__ __launch_bounds__(PW_THREADS) void pw_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                              const float* __restrict__ xs,
                                                              float* __restrict__ partial, int B, int M, int K, int HW,
                                                              int splits, int chunks_per_split) {
  extern __shared__ __attribute__((aligned(16))) float s_acc[];      // [4 waves][NTI][4][64] (SCALED: + [4 waves][256])
  const int T = (M + 15) >> 4, NTI = (K + 15) >> 4, wave = threadIdx.x >> 6;
  const int t = (int)(blockIdx.x % T);
  const long long unit = (long long)(blockIdx.x / T) * (PW_THREADS / 64) + wave;   // (b, split), four per workgroup
  const int sp = (int)(unit % splits);
  const long long b_raw = unit / splits;
  const bool live = b_raw < B;
  const long long b = live ? b_raw : 0;
  const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
  pw_v4f acc[PW_NTI];
#pragma unroll
  for (int u = 0; u < PW_NTI; ++u) acc[u] = pw_v4f{0.0f, 0.0f, 0.0f, 0.0f};
  // (SCALED) the unit's K column scales go to a wave-private LDS row here: fetched where the sums leave the accumulators
  // they were up to 15 loads, one exposed latency after the other (the scaled launches measured 12 % over the plain
  // ones), and 15 more registers would cost the kernel its third wave per SIMD (160 -> 176 VGPRs)
  float* s_sc = s_acc + (size_t)(PW_THREADS / 64) * NTI * 256 + wave * 256;
  if (SCALED)
    for (int i = lane; i < 16 * NTI; i += 64) s_sc[i] = i < K ? xs[(size_t)b * K + i] : 0.0f;
  const int gm = 16 * t + n;
  const bool gok = gm < M;
  const float* grow = g + ((size_t)b * M + (gok ? gm : 0)) * HW;
  const float* xbase = x + (size_t)b * K * HW;
  const int nchunks = (HW + 63) >> 6;
  const int c_begin = sp * chunks_per_split, c_end = live ? min(nchunks, (sp + 1) * chunks_per_split) : c_begin;
  const int nsteps = (c_end - c_begin) * NTI;
  auto load_step = [&](int step, float (&dst)[16]) {        // step -> (chunk, input tile)
    if (step < nsteps) {
      const int c = c_begin + step / NTI, u = step % NTI;
      const int xk = 16 * u + n;
      const bool xok = xk < K;
      pw_load_slots<VEC>(xbase + (size_t)(xok ? xk : 0) * HW, xok, c * 64, q, HW, dst);
    }
  };
  float a[16], a_next[16], cur[16], nx1[16], nx2[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) a[s] = a_next[s] = cur[s] = nx1[s] = nx2[s] = 0.0f;
  if (c_begin < c_end) pw_load_slots<VEC>(grow, gok, c_begin * 64, q, HW, a);
  load_step(0, cur);
  load_step(1, nx1);
  int step = 0;
  for (int c = c_begin; c < c_end; ++c) {
    if (c + 1 < c_end) pw_load_slots<VEC>(grow, gok, (c + 1) * 64, q, HW, a_next);
#pragma unroll
    for (int u = 0; u < PW_NTI; ++u) {
      if (u < NTI) {
        load_step(step + 2, nx2);
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], cur[s], acc[u], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) { cur[s] = nx1[s]; nx1[s] = nx2[s]; }
        ++step;
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) a[s] = a_next[s];
  }
  // the four waves of the workgroup (same row tile, four (sample, split) units) meet in LDS, added in wave order
#pragma unroll
  for (int u = 0; u < PW_NTI; ++u)
    if (u < NTI) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s_acc[((wave * NTI + u) * 4 + r) * 64 + lane] = SCALED ? acc[u][r] * s_sc[16 * u + n] : acc[u][r];
    }
  __syncthreads();
  float* pp = partial + (size_t)(blockIdx.x / T) * M * K;
  for (int e = threadIdx.x; e < NTI * 256; e += PW_THREADS) {          // e = (u, r, lane)
    const int l = e & 63, r = (e >> 6) & 3, u = e >> 8;
    float v = 0.0f;
#pragma unroll
    for (int wv = 0; wv < PW_THREADS / 64; ++wv) v += s_acc[((wv * NTI + u) * 4 + r) * 64 + l];
    const int m = 16 * t + 4 * (l >> 4) + r, k = 16 * u + (l & 15);
    if (m < M && k < K) pp[(size_t)m * K + k] = v;
  }
}

// out[i] = sum_chunk partial[chunk][i]: a workgroup takes 16 outputs (lanes along i) in 16 groups of chunks, each
// thread adding its chunks in order with four running sums; the groups meet in LDS (a fixed order)
__global__ __launch_bounds__(PW_THREADS) void pw_reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                                        int n, int nchunk) {
  __shared__ float s_v[16][16];
  const int il = threadIdx.x & 15, kg = threadIdx.x >> 4, i = blockIdx.x * 16 + il;
  float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
  if (i < n) {
    int k = kg;
    for (; k + 48 < nchunk; k += 64) {
      v0 += partial[(size_t)k * n + i];
      v1 += partial[(size_t)(k + 16) * n + i];
      v2 += partial[(size_t)(k + 32) * n + i];
      v3 += partial[(size_t)(k + 48) * n + i];
    }
    for (; k < nchunk; k += 16) v0 += partial[(size_t)k * n + i];
  }
  s_v[kg][il] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (threadIdx.x < 16 && i < n) {
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += s_v[j][il];
    out[i] = v;
  }
}

static int pw_vec(const void* a, const void* b, int HW) {
  const uintptr_t bits = (uintptr_t)a | (uintptr_t)b;
  if ((HW & 3) == 0 && (bits & 15) == 0) return 4;
  if ((HW & 1) == 0 && (bits & 7) == 0) return 2;
  return 1;
}

// 1 when (Cin, Cout) is a shape these kernels take (both orientations of the weight fit LDS, channel counts are
// multiples of 4, the weight gradient's input tiles fit the accumulators); the caller keeps its GEMM form otherwise
extern "C" int ias_pwconv_supported(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || (Cin & 3) || (Cout & 3)) return 0;
  const long long pad_out = (Cout + 15) & ~15, pad_in = (Cin + 15) & ~15;
  if (pad_out * Cin > PW_LDS_FLOATS || pad_in * Cout > PW_LDS_FLOATS) return 0;
  return Cin <= 16 * PW_NTI ? 1 : 0;
}

template <bool SCALED>
static void pw_apply_launch(int vec, long long grid, size_t lds, hipStream_t st, const float* x, const float* xs, const float* w,
                            float* y, int B, int M, int K, int HW, int transpose, int blocks, int ngroups, int cg, long long items) {
  if (lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)pw_apply_kernel<4, SCALED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)pw_apply_kernel<2, SCALED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)pw_apply_kernel<1, SCALED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  switch (vec) {
    case 4: hipLaunchKernelGGL((pw_apply_kernel<4, SCALED>), dim3((unsigned)grid), dim3(PW_THREADS), lds, st, x, xs, w, y, B, M, K, HW, transpose, blocks, ngroups, cg, items); break;
    case 2: hipLaunchKernelGGL((pw_apply_kernel<2, SCALED>), dim3((unsigned)grid), dim3(PW_THREADS), lds, st, x, xs, w, y, B, M, K, HW, transpose, blocks, ngroups, cg, items); break;
    default: hipLaunchKernelGGL((pw_apply_kernel<1, SCALED>), dim3((unsigned)grid), dim3(PW_THREADS), lds, st, x, xs, w, y, B, M, K, HW, transpose, blocks, ngroups, cg, items); break;
  }
}
// xs: NULL, or a scale per (sample, input row) [B][K] applied to x on load
static int pw_apply(const float* x, const float* xs, const float* w, float* y, int B, int M, int K, int HW, int transpose, hipStream_t st) {
  if (((uintptr_t)w & 15) != 0) return IAS_ERR_ARG;
  const int T = (M + 15) / 16, blocks = (HW + 63) / 64;
  int cg = PW_CG;                                   // fewer tiles per wave (the input re-read from L2) when that is what
  while (cg > 1 && (long long)B * blocks * ((T + cg - 1) / cg) < 2048) --cg;   // it takes to put work on every SIMD
  const int ngroups = (T + cg - 1) / cg;
  const long long items = (long long)B * blocks * ngroups;
  long long grid = (items + PW_THREADS / 64 - 1) / (PW_THREADS / 64);
  if (grid > 1024) grid = 1024;                     // the weight is staged once per workgroup
  const size_t lds = sizeof(float) * (size_t)T * (K / 4) * PW_WROW;
  if (xs) pw_apply_launch<true>(pw_vec(x, y, HW), grid, lds, st, x, xs, w, y, B, M, K, HW, transpose, blocks, ngroups, cg, items);
  else pw_apply_launch<false>(pw_vec(x, y, HW), grid, lds, st, x, xs, w, y, B, M, K, HW, transpose, blocks, ngroups, cg, items);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// Conv2d(Cin, Cout, 1, bias=False) forward: x [B,Cin,HW], w [Cout,Cin] -> y [B,Cout,HW]
extern "C" int ias_pwconv_forward(const float* x, const float* w, float* y, int B, int Cin, int Cout, int HW, void* stream_) {
  if (!x || !w || !y || B <= 0 || HW <= 0) return IAS_ERR_ARG;
  if (!ias_pwconv_supported(Cin, Cout)) return IAS_ERR_UNSUPPORTED;
  return pw_apply(x, nullptr, w, y, B, Cout, Cin, HW, 0, (hipStream_t)stream_);
}
// The same on a gated input: y[b] = W (x[b] * scale[b]), scale [B,Cin] -- the projection behind a squeeze-excitation block
// (torchvision InvertedResidual: SqueezeExcitation -> Conv2dNormActivation(cexp, cout, 1); reference: the trunk of
// /root/reference/audioembed.py:61) without the gate's own pass over the expanded map.
extern "C" int ias_pwconv_forward_scaled(const float* x, const float* scale, const float* w, float* y, int B, int Cin, int Cout,
                                         int HW, void* stream_) {
  if (!x || !scale || !w || !y || B <= 0 || HW <= 0) return IAS_ERR_ARG;
  if (!ias_pwconv_supported(Cin, Cout)) return IAS_ERR_UNSUPPORTED;
  return pw_apply(x, scale, w, y, B, Cout, Cin, HW, 0, (hipStream_t)stream_);
}

// its input gradient: g [B,Cout,HW], w [Cout,Cin] -> gx [B,Cin,HW]
extern "C" int ias_pwconv_backward_data(const float* g, const float* w, float* gx, int B, int Cin, int Cout, int HW,
                                        void* stream_) {
  if (!g || !w || !gx || B <= 0 || HW <= 0) return IAS_ERR_ARG;
  if (!ias_pwconv_supported(Cin, Cout)) return IAS_ERR_UNSUPPORTED;
  return pw_apply(g, nullptr, w, gx, B, Cin, Cout, HW, 1, (hipStream_t)stream_);
}

static int pw_wgrad_splits(int B, int Cout, int HW) {
  const int T = (Cout + 15) / 16, chunks = (HW + 63) / 64;
  int splits = (2048 + B * T - 1) / (B * T);
  if (splits < 1) splits = 1;
  if (splits > chunks) splits = chunks;
  return splits;
}

// floats of scratch for ias_pwconv_backward_weight
extern "C" long long ias_pwconv_weight_scratch(int B, int Cin, int Cout, int HW) {
  if (B <= 0 || Cin <= 0 || Cout <= 0 || HW <= 0) return IAS_ERR_ARG;
  return (long long)B * pw_wgrad_splits(B, Cout, HW) * Cin * Cout;
}

// its weight gradient: g [B,Cout,HW], x [B,Cin,HW] -> gw [Cout,Cin]; scratch: ias_pwconv_weight_scratch floats.
// gw == nullptr: the partial sums only -> *nchunk rows of Cout Cin floats in `scratch` (ias_pwconv_backward_weight_partials)
template <bool SCALED>
static void pw_wgrad_launch(int vec, long long grid, size_t lds, hipStream_t st, const float* g, const float* x, const float* xs,
                            float* scratch, int B, int Cout, int Cin, int HW, int splits, int cps) {
  switch (vec) {
    case 4: hipLaunchKernelGGL((pw_wgrad_kernel<4, SCALED>), dim3((unsigned)grid), dim3(PW_THREADS), lds, st, g, x, xs, scratch, B, Cout, Cin, HW, splits, cps); break;
    case 2: hipLaunchKernelGGL((pw_wgrad_kernel<2, SCALED>), dim3((unsigned)grid), dim3(PW_THREADS), lds, st, g, x, xs, scratch, B, Cout, Cin, HW, splits, cps); break;
    default: hipLaunchKernelGGL((pw_wgrad_kernel<1, SCALED>), dim3((unsigned)grid), dim3(PW_THREADS), lds, st, g, x, xs, scratch, B, Cout, Cin, HW, splits, cps); break;
  }
}
// xs: NULL, or the [B][Cin] scale of the forward's gated input (ias_pwconv_forward_scaled)
static int pw_backward_weight(const float* g, const float* x, const float* xs, float* gw, float* scratch, int B, int Cin, int Cout, int HW,
                              int* nchunk, void* stream_) {
  if (!g || !x || !scratch || B <= 0 || HW <= 0) return IAS_ERR_ARG;
  if (!ias_pwconv_supported(Cin, Cout)) return IAS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream_;
  const int T = (Cout + 15) / 16, NTI = (Cin + 15) / 16, chunks = (HW + 63) / 64, splits = pw_wgrad_splits(B, Cout, HW);
  const int cps = (chunks + splits - 1) / splits;
  const long long units = (long long)B * splits, ugroups = (units + PW_THREADS / 64 - 1) / (PW_THREADS / 64);
  const long long grid = ugroups * T;
  if (grid > 0x7fffffffLL) return IAS_ERR_ARG;
  const size_t lds = sizeof(float) * (size_t)(PW_THREADS / 64) * (NTI * 256 + (xs ? 256 : 0));   // (+ the waves' scale rows)
  if (xs) pw_wgrad_launch<true>(pw_vec(g, x, HW), grid, lds, st, g, x, xs, scratch, B, Cout, Cin, HW, splits, cps);
  else pw_wgrad_launch<false>(pw_vec(g, x, HW), grid, lds, st, g, x, xs, scratch, B, Cout, Cin, HW, splits, cps);
  if (nchunk) *nchunk = (int)ugroups;
  if (gw) {
    const int n = Cout * Cin;
    hipLaunchKernelGGL(pw_reduce_partials_kernel, dim3((n + 15) / 16), dim3(PW_THREADS), 0, st, scratch, gw, n, (int)ugroups);
  }
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
extern "C" int ias_pwconv_backward_weight(const float* g, const float* x, float* gw, float* scratch, int B, int Cin, int Cout,
                                          int HW, void* stream_) {
  if (!gw) return IAS_ERR_ARG;
  return pw_backward_weight(g, x, nullptr, gw, scratch, B, Cin, Cout, HW, nullptr, stream_);
}
// the weight gradient of ias_pwconv_forward_scaled: gw = sum_b g[b] (x[b] * scale[b])^T
extern "C" int ias_pwconv_backward_weight_scaled(const float* g, const float* x, const float* scale, float* gw, float* scratch,
                                                 int B, int Cin, int Cout, int HW, void* stream_) {
  if (!gw || !scale) return IAS_ERR_ARG;
  return pw_backward_weight(g, x, scale, gw, scratch, B, Cin, Cout, HW, nullptr, stream_);
}
// The same without the reduction launch: -> the number of partial rows (> 0; row r is scratch[r Cout Cin ...]) for
// ias_reduce_partials_multi, or a negative status
extern "C" int ias_pwconv_backward_weight_partials(const float* g, const float* x, float* scratch, int B, int Cin, int Cout,
                                                   int HW, void* stream_) {
  int nchunk = 0;
  const int rc = pw_backward_weight(g, x, nullptr, nullptr, scratch, B, Cin, Cout, HW, &nchunk, stream_);
  return rc != IAS_OK ? rc : nchunk;
}
extern "C" int ias_pwconv_backward_weight_partials_scaled(const float* g, const float* x, const float* scale, float* scratch,
                                                          int B, int Cin, int Cout, int HW, void* stream_) {
  if (!scale) return IAS_ERR_ARG;
  int nchunk = 0;
  const int rc = pw_backward_weight(g, x, scale, nullptr, scratch, B, Cin, Cout, HW, &nchunk, stream_);
  return rc != IAS_OK ? rc : nchunk;
}

// ---- the weight-gradient reductions of a whole backward pass in ONE launch ------------------------------------------
// Every weight gradient of the trunk (thin 1x1, depthwise, stem) ends in out[i] = sum_r partial[r n + i], a launch of
// 5-6 us that is all latency, 27 of them on the backward's dependency chain per pretraining step.  The *_partials entry
// points leave the partial rows behind; this kernel folds all of them: a workgroup per 16 outputs of one item, the
// arithmetic of pw_reduce_partials_kernel (fixed order: deterministic).  The table travels BY VALUE in the kernel
// arguments (2.7 KB of the 4 KB a launch may carry): no device table, no staging copy, nothing to keep alive for a
// captured graph, and nothing allocated while a capture is open.
#define IAS_REDUCE_MAX_ITEMS 96
struct IasReduceTable {
  IasReduceItem it[IAS_REDUCE_MAX_ITEMS];
  int first[IAS_REDUCE_MAX_ITEMS + 1];       // running sums of ceil(n / 16)
};
__global__ __launch_bounds__(PW_THREADS) void reduce_partials_multi_kernel(const IasReduceTable tab, int count) {
  __shared__ float s_v[16][16];
  int j = 0;
  while (j + 1 < count && (int)blockIdx.x >= tab.first[j + 1]) ++j;     // uniform: scalar loads from the argument segment
  const float* __restrict__ partial = tab.it[j].partial;
  float* __restrict__ out = tab.it[j].out;
  const int n = tab.it[j].n, nchunk = tab.it[j].rows;
  const int il = threadIdx.x & 15, kg = threadIdx.x >> 4, i = ((int)blockIdx.x - tab.first[j]) * 16 + il;
  float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
  if (i < n) {
    int k = kg;
    for (; k + 48 < nchunk; k += 64) {
      v0 += partial[(size_t)k * n + i];
      v1 += partial[(size_t)(k + 16) * n + i];
      v2 += partial[(size_t)(k + 32) * n + i];
      v3 += partial[(size_t)(k + 48) * n + i];
    }
    for (; k < nchunk; k += 16) v0 += partial[(size_t)k * n + i];
  }
  s_v[kg][il] = (v0 + v1) + (v2 + v3);
  __syncthreads();
  if (threadIdx.x < 16 && i < n) {
    float v = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) v += s_v[q][il];
    out[i] = v;
  }
}
// items: HOST array of `count` entries (device pointers inside); more than IAS_REDUCE_MAX_ITEMS entries: several launches
extern "C" int ias_reduce_partials_multi(const IasReduceItem* items, int count, void* stream_) {
  if (!items || count <= 0) return IAS_ERR_ARG;
  for (int i = 0; i < count; ++i)
    if (!items[i].partial || !items[i].out || items[i].n <= 0 || items[i].rows <= 0) return IAS_ERR_ARG;
  for (int c0 = 0; c0 < count; c0 += IAS_REDUCE_MAX_ITEMS) {
    IasReduceTable tab;
    const int c = count - c0 < IAS_REDUCE_MAX_ITEMS ? count - c0 : IAS_REDUCE_MAX_ITEMS;
    long long blocks = 0;
    for (int i = 0; i < c; ++i) {
      tab.it[i] = items[c0 + i];
      tab.first[i] = (int)blocks;
      blocks += (items[c0 + i].n + 15) / 16;
    }
    if (blocks > 0x7fffffffLL) return IAS_ERR_ARG;
    tab.first[c] = (int)blocks;
    hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3((unsigned)blocks), dim3(PW_THREADS), 0, (hipStream_t)stream_, tab, c);
  }
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
