// Diagnostic: device vs host results of the voice_math.h primitives on random inputs.
#include <cstdio>
#include <vector>
#include <random>
#include "voice_math.h"

#define NFN 10
__host__ __device__ inline float evalfn(int fn, float a, float b) {
  switch (fn) {
    case 0: return ias_cos_cr(a);
    case 1: return ias_remainder(a, (float)IAS_TWO_PI_D);
    case 2: return ias_div(a, b);
    case 3: return ias_pow_cr(fabsf(a) * 0.01f, fabsf(b));
    case 4: return ias_exp2_cr(a * 0.05f);
    case 5: { float mode[5] = {0.1f, 0.3f, 0.2f, 0.25f, 0.15f}; return ias_lfo_shape_mix(a, mode); }
    case 6: return ias_log2_cr(fabsf(a));
    case 7: return ias_exp2_slow_cr(a * 0.05f);
    case 8: return ias_fma(a, b, 0.37f);
    case 9: return ias_log10_cr(fabsf(a) + 1.0f);
  }
  return 0.f;
}
__global__ void k(int fn, const float* a, const float* b, float* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = evalfn(fn, a[i], b[i]);
}
int main() {
  const int n = 1 << 20;
  std::mt19937 g(1);
  std::uniform_real_distribution<float> u(-100.f, 100.f);
  std::vector<float> a(n), b(n), o(n);
  for (int i = 0; i < n; ++i) { a[i] = u(g); b[i] = u(g) * 0.06f; }
  float *da, *db, *dout;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
  const char* names[NFN] = {"cos_cr", "remainder", "div", "pow_cr", "exp2_cr", "lfo_shape_mix", "log2_cr", "exp2_slow_cr", "fma", "log10_cr"};
  for (int fn = 0; fn < NFN; ++fn) {
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, fn, da, db, dout, n);
    hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0, first = -1;
    for (int i = 0; i < n; ++i) {
      float h = evalfn(fn, a[i], b[i]);
      if (!(h == o[i]) && !(h != h && o[i] != o[i])) { if (first < 0) first = i; ++bad; }
    }
    printf("%-14s mismatches %d / %d", names[fn], bad, n);
    if (first >= 0) printf("   e.g. a=%.9g b=%.9g host=%.9g dev=%.9g", a[first], b[first], evalfn(fn, a[first], b[first]), o[first]);
    printf("\n");
  }
  return 0;
}
