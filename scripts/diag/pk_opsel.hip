// Does v_pk_fma_f32 honour op_sel / op_sel_hi on an SGPR-pair source on gfx950?  (x = (1, 10), s = (2, 3))
//   plain                          -> expect (2, 30)
//   op_sel_hi:[1,0,1]              -> (2, 20)   low half broadcast
//   op_sel:[0,1,0] op_sel_hi:[1,1,1] -> (3, 30) high half broadcast
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* __restrict__ s, float* out) {
  f32x2 sp = *reinterpret_cast<const f32x2*>(s);   // uniform -> SGPR pair
  f32x2 x = {1.0f, 10.0f};
  f32x2 z = {0.0f, 0.0f};
  f32x2 a, b, c;
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a) : "v"(x), "s"(sp), "v"(z));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(b) : "v"(x), "s"(sp), "v"(z));
  asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(c) : "v"(x), "s"(sp), "v"(z));
  if (threadIdx.x == 0) {
    out[0] = a.x; out[1] = a.y; out[2] = b.x; out[3] = b.y; out[4] = c.x; out[5] = c.y;
  }
}
int main() {
  float hs[2] = {2.0f, 3.0f}, ho[6];
  float *s, *o;
  hipMalloc(&s, 8); hipMalloc(&o, 24);
  hipMemcpy(s, hs, 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, o);
  hipMemcpy(ho, o, 24, hipMemcpyDeviceToHost);
  printf("plain (%g,%g)  lo-bcast (%g,%g)  hi-bcast (%g,%g)\n", ho[0], ho[1], ho[2], ho[3], ho[4], ho[5]);
  return 0;
}
