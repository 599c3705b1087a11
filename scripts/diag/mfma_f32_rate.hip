// Issue rate of the f32-input MFMAs on one wave per SIMD: cycles per instruction, measured with s_memtime.
// build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 scripts/diag/mfma_f32_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int CH, int REP>
__global__ void k16(float* out, unsigned long long* cyc, float a0, float b0) {
  f4 acc[CH];
  for (int i = 0; i < CH; ++i) acc[i] = (f4){0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH, int REP>
__global__ void k32(float* out, unsigned long long* cyc, float a0, float b0) {
  f16v acc[CH];
  for (int i = 0; i < CH; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  float a = a0 + threadIdx.x, b = b0;
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < CH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < CH; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K>
static void run(const char* name, K kern, int nmfma, int blocks, int threads) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s blocks %4d x %3d thr: %7.1f memtime ticks/MFMA (wave 0), kernel %.1f us -> %.1f ns per MFMA per wave\n", name, blocks,
         threads, (double)c / nmfma, ms * 1e3, ms * 1e6 / nmfma);
  hipFree(out); hipFree(cyc);
}

int main() {
  constexpr int REP = 2000;
  run("16x16x4 f32, 1 chain", k16<1, REP>, REP * 8 * 1, 256, 256);
  run("16x16x4 f32, 2 chains", k16<2, REP>, REP * 8 * 2, 256, 256);
  run("16x16x4 f32, 4 chains", k16<4, REP>, REP * 8 * 4, 256, 256);
  run("16x16x4 f32, 4 chains, 2 waves/SIMD", k16<4, REP>, REP * 8 * 4, 512, 256);
  run("16x16x4 f32, 4 chains, 1 WG only", k16<4, REP>, REP * 8 * 4, 1, 256);
  run("32x32x2 f32, 1 chain", k32<1, REP>, REP * 8 * 1, 256, 256);
  run("32x32x2 f32, 2 chains", k32<2, REP>, REP * 8 * 2, 256, 256);
  run("32x32x2 f32, 2 chains, 1 WG only", k32<2, REP>, REP * 8 * 2, 1, 256);
  return 0;
}
