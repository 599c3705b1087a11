// Do the fp32 MFMA pipe and the vector ALU of a SIMD overlap when DIFFERENT waves feed them (the STFT kernel's premise)?
// 512-thread workgroups, one per CU: waves 0-3 and 4-7 land pairwise on the four SIMDs.  Roles by wave number:
//   M: v_mfma_f32_16x16x4_f32 stream, 4 independent accumulators     V: v_fma_f32 stream, 8 independent chains
//   L: ds_read_b128 stream (conflict-free)
// Reported: kernel time of each role alone (the partner half exits at once) and of the pair together.
// build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 scripts/diag/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int ROLE> __device__ __forceinline__ float work(int rep, float a, float b, const f4* lds) {
  float s = 0.f;
  if (ROLE == 0) {
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f4){0, 0, 0, 0};
#pragma unroll 1
    for (int r = 0; r < rep; ++r) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if (ROLE == 1) {
    float c[8];
    for (int i = 0; i < 8; ++i) c[i] = a + i;
#pragma unroll 1
    for (int r = 0; r < rep; ++r) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_fmaf(c[i], b, a);
    }
    for (int i = 0; i < 8; ++i) s += c[i];
  } else if (ROLE == 3) {                       // bf16 MFMA 32x32x16, 2 accumulators
    typedef short bf8 __attribute__((ext_vector_type(8)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v acc[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    bf8 av, bv;
    for (int j = 0; j < 8; ++j) { av[j] = (short)(0x3f80 + j); bv[j] = (short)(0x3f00 + (threadIdx.x & 7)); }
#pragma unroll 1
    for (int r = 0; r < rep; ++r) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  } else if (ROLE == 4) {                       // integer VALU stream
    int c[8];
    for (int i = 0; i < 8; ++i) c[i] = (int)a + i;
#pragma unroll 1
    for (int r = 0; r < rep; ++r) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = (c[i] ^ (int)b) + i;
    }
    for (int i = 0; i < 8; ++i) s += (float)c[i];
  } else if (ROLE == 2) {
    f4 c = (f4){0, 0, 0, 0};
#pragma unroll 1
    for (int r = 0; r < rep; ++r) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { const f4 v = lds[(u * 64 + (threadIdx.x & 63)) ]; c += v; }
    }
    s = c[0] + c[1] + c[2] + c[3];
  }
  return s;
}
// waves 0-3: role RA (repA iterations), waves 4-7: role RB (repB iterations); rep 0 = that half exits at once
template <int RA, int RB> __global__ __launch_bounds__(512) void pair(float* out, int repA, int repB, float a, float b) {
  __shared__ f4 lds[16 * 64];
  for (int i = threadIdx.x; i < 16 * 64; i += 512) lds[i] = (f4){1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const int half = threadIdx.x >> 8;
  float s = half == 0 ? work<RA>(repA, a + threadIdx.x, b, lds) : work<RB>(repB, a + threadIdx.x, b, lds);
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <typename K> static float timeit(K kern, float* out, int ra, int rb) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int w = 0; w < 3; ++w) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, ra, rb, 1.0f, 0.999f);
    hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms * 1e3f;
}
int main() {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 512);
  const int RM = 2000, RV = 2000, RL = 2000;   // M: 64000 MFMAs (2.05 M cycles at 32), V: 256000 fmas, L: 32000 b128 reads
  const char* names[3] = {"MFMA", "VALU", "LDS "};
  printf("role alone (us):  M %.1f   V %.1f   L %.1f\n", timeit(pair<0, 1>, out, RM, 0), timeit(pair<0, 1>, out, 0, RV), timeit(pair<0, 2>, out, 0, RL));
  printf("M + V on one SIMD: %.1f us\n", timeit(pair<0, 1>, out, RM, RV));
  printf("M + M on one SIMD: %.1f us\n", timeit(pair<0, 0>, out, RM, RM));
  printf("V + V on one SIMD: %.1f us\n", timeit(pair<1, 1>, out, RV, RV));
  printf("M + L on one SIMD: %.1f us\n", timeit(pair<0, 2>, out, RM, RL));
  printf("V + L on one SIMD: %.1f us\n", timeit(pair<1, 2>, out, RV, RL));
  printf("bf16 MFMA alone: %.1f us;  bf16 M + V: %.1f us;  bf16 M + f32 M: %.1f us\n", timeit(pair<3, 1>, out, RM, 0), timeit(pair<3, 1>, out, RM, RV), timeit(pair<3, 0>, out, RM, RM));
  printf("int VALU alone: %.1f us;  f32 M + int V: %.1f us\n", timeit(pair<0, 4>, out, 0, RV), timeit(pair<0, 4>, out, RM, RV));
  (void)names;
  return 0;
}
