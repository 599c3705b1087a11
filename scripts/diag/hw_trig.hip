// Diagnostic: accuracy of gfx950 hardware sin/cos (revolutions input) and exp/rcp based tanh.
#include <cstdio>
#include <cmath>
#include <vector>
#include <hip/hip_runtime.h>
__global__ void k(const float* fr, float* s, float* c, float* th, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s[i] = __builtin_amdgcn_sinf(fr[i]);
  c[i] = __builtin_amdgcn_cosf(fr[i]);
  float z = fr[i] * 40.0f;   // tanh argument in [-20, 20]
  float az = fabsf(z);
  float t = __builtin_amdgcn_exp2f(-2.885390081777927f * az);   // exp(-2|z|)
  float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
  th[i] = z < 0 ? -r : r;
}
int main() {
  const int n = 1 << 22;
  std::vector<float> fr(n), s(n), c(n), th(n);
  for (int i = 0; i < n; ++i) fr[i] = -0.5f + (float)i / (float)(n - 1);
  float *d0, *d1, *d2, *d3;
  hipMalloc(&d0, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&d3, n * 4);
  hipMemcpy(d0, fr.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d0, d1, d2, d3, n);
  hipMemcpy(s.data(), d1, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(th.data(), d3, n * 4, hipMemcpyDeviceToHost);
  double es = 0, ec = 0, et = 0;
  for (int i = 0; i < n; ++i) {
    double x = 2 * M_PI * (double)fr[i];
    es = fmax(es, fabs(s[i] - sin(x)));
    ec = fmax(ec, fabs(c[i] - cos(x)));
    et = fmax(et, fabs(th[i] - tanh((double)fr[i] * 40.0)));
  }
  printf("max abs err: v_sin %.3e  v_cos %.3e  tanh(exp2,rcp) %.3e\n", es, ec, et);
  return 0;
}
