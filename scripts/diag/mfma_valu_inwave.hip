// How many independent vector instructions hide behind one v_mfma_f32_16x16x4_f32 of the SAME wave (one wave per SIMD),
// and does a higher priority let a VALU wave overlap an MFMA wave of the same SIMD?
// build+run on the GPU box: hipcc -O3 --offload-arch=gfx950 scripts/diag/mfma_valu_inwave.hip -o /tmp/iw && /tmp/iw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K> __global__ __launch_bounds__(256) void inwave(float* out, int rep, float a, float b) {
  f4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f4){0, 0, 0, 0};
  float c[8];
  for (int i = 0; i < 8; ++i) c[i] = a + i + threadIdx.x;
  const float av = a + threadIdx.x;
#pragma unroll 1
  for (int r = 0; r < rep; ++r) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < K; ++k) c[(k + i) & 7] = __builtin_fmaf(c[(k + i) & 7], b, a);
        __builtin_amdgcn_sched_barrier(0);
      }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += c[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// cross-wave with priority: waves 0-3 MFMA, waves 4-7 VALU at priority PRIO
template <int PRIO> __global__ __launch_bounds__(512) void cross(float* out, int repM, int repV, float a, float b) {
  float s = 0;
  if (threadIdx.x < 256) {
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f4){0, 0, 0, 0};
    const float av = a + threadIdx.x;
#pragma unroll 1
    for (int r = 0; r < repM; ++r)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    __builtin_amdgcn_s_setprio(PRIO);
    float c[8];
    for (int i = 0; i < 8; ++i) c[i] = a + i + threadIdx.x;
#pragma unroll 1
    for (int r = 0; r < repV; ++r)
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_fmaf(c[i], b, a);
    for (int i = 0; i < 8; ++i) s += c[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <typename F> static float timeit(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int w = 0; w < 3; ++w) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1); }
  return ms * 1e3f;
}
int main() {
  float* out; hipMalloc(&out, sizeof(float) * 256 * 512);
  const int R = 2000;   // 64000 MFMAs per wave
#define IW(K) printf("in-wave: 1 MFMA + %d fma per slot: %.1f us (MFMA alone = K 0)\n", K, timeit([&] { hipLaunchKernelGGL(inwave<K>, dim3(256), dim3(256), 0, 0, out, R, 1.0f, 0.999f); }))
  IW(0); IW(2); IW(4); IW(6); IW(8); IW(12); IW(16);
  printf("cross-wave M + V, V at prio 0: %.1f us;  prio 3: %.1f us\n",
         timeit([&] { hipLaunchKernelGGL(cross<0>, dim3(256), dim3(512), 0, 0, out, R, R, 1.0f, 0.999f); }),
         timeit([&] { hipLaunchKernelGGL(cross<3>, dim3(256), dim3(512), 0, 0, out, R, R, 1.0f, 0.999f); }));
  return 0;
}
