// Diagnostic: SIMD time per wave64 VALU instruction as a function of waves per SIMD (1, 2, 4, 8), for plain fp32
// (independent accumulators), fp64, and a plain/fp64/convert mix -- does the 2-clock plain-fp32 rate need 8 waves?
#include <cstdio>
#include <hip/hip_runtime.h>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void k(float* out, int iters) {
  float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f;
  float f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
  double d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3, db = 1.0000001;
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP8(REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\nv_fma_f32 %1, %8, %9, %1\nv_fma_f32 %2, %8, %9, %2\nv_fma_f32 %3, %8, %9, %3\nv_fma_f32 %4, %8, %9, %4\nv_fma_f32 %5, %8, %9, %5\nv_fma_f32 %6, %8, %9, %6\nv_fma_f32 %7, %8, %9, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(a));)) }
    if (OP == 1) { REP8(REP8(asm volatile("v_fma_f64 %0, %4, %4, %0\nv_fma_f64 %1, %4, %4, %1\nv_fma_f64 %2, %4, %4, %2\nv_fma_f64 %3, %4, %4, %3\nv_fma_f64 %0, %4, %4, %0\nv_fma_f64 %1, %4, %4, %1\nv_fma_f64 %2, %4, %4, %2\nv_fma_f64 %3, %4, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));)) }
    // mix: 4 plain fp32, 2 fp64, 1 convert, 1 floor  (8 instructions)
    if (OP == 2) { REP8(REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\nv_mul_f32 %1, %8, %1\nv_fma_f64 %4, %10, %10, %4\nv_add_f32 %2, %9, %2\nv_cvt_f64_f32 %5, %0\nv_fma_f32 %3, %8, %9, %3\nv_add_f64 %6, %10, %6\nv_floor_f32 %7, %1" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(f7) : "v"(b), "v"(a), "v"(db));)) }
    // dependent plain chain
    if (OP == 3) { REP8(REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(b), "v"(a));)) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + (float)(d0 + d1 + d2 + d3);
}
template <int OP> void sweep(const char* name) {
  float* out; hipMalloc(&out, 4096 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 8; wps *= 2) {      // waves per SIMD: blocks of 256 threads, wps blocks per CU
    const int iters = 2000, blocks = 256 * wps;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %d wave(s)/SIMD: %6.2f ns of SIMD time per wave64 instruction\n", name, wps, ms * 1e6 / ((double)iters * 512.0 * wps));
  }
  hipFree(out);
}
int main() {
  sweep<0>("v_fma_f32 x8 indep"); sweep<3>("v_fma_f32 dependent"); sweep<1>("v_fma_f64 x4 indep"); sweep<2>("mix 4 f32/2 f64/cvt/floor");
  return 0;
}
