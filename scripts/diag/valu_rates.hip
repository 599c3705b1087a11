// Diagnostic: issue cost (cycles per wave64 instruction, one wave per SIMD and 4 waves per SIMD) of the
// VALU instructions the Voice kernel is made of.
#include <cstdio>
#include <hip/hip_runtime.h>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  float a = threadIdx.x * 1e-3f + 1.0f, b = 1.0001f, c = 0.5f;
  double da = a, db = 1.0000001, dc = 0.25;
  float a2 = a + 1.f, b2 = b, c2 = c;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
    if (OP == 1) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(da) : "v"(db), "v"(dc));) }
    if (OP == 2) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(da) : "v"(db), "v"(dc));) }
    if (OP == 3) { REP64(asm volatile("v_add_f64 %0, %0, %1" : "+v"(da) : "v"(db));) }
    if (OP == 4) { REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(da) : "v"(db));) }
    if (OP == 5) { REP64(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(da) : "v"(a));) }
    if (OP == 6) { REP64(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a) : "v"(da));) }
    if (OP == 7) { REP64(asm volatile("v_rndne_f64 %0, %1" : "=v"(db) : "v"(da));) }
    if (OP == 8) { REP64(asm volatile("v_ldexp_f64 %0, %1, 3" : "=v"(db) : "v"(da));) }
    if (OP == 9) { REP64(asm volatile("v_sin_f32 %0, %1" : "=v"(b) : "v"(a));) }
    if (OP == 10) { REP64(asm volatile("v_exp_f32 %0, %1" : "=v"(b) : "v"(a));) }
    if (OP == 11) { REP64(asm volatile("v_rcp_f32 %0, %1" : "=v"(b) : "v"(a));) }
    if (OP == 12) { REP64(asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b) : "v"(a), "v"(c));) }
    if (OP == 13) { REP64(asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(db) : "v"(da), "v"(dc));) }
    if (OP == 14) { REP64(asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(b) : "v"(a));) }
    if (OP == 15) { REP64(asm volatile("v_mov_b32 %0, %1" : "=v"(b) : "v"(a));) }
    if (OP == 16) { REP64(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(a));) }
    if (OP == 17) { REP64(asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(b) : "v"(a), "v"(c), "v"(c2));) }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + (float)(da + db + dc) + a2 + b2 + c2;
}
template <int OP> void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4 * 4); hipMalloc(&cyc, 2048 * 8);
  for (int waves = 1; waves <= 4; waves *= 4) {   // waves per SIMD
    const int threads = 256 * waves / 1 > 1024 ? 1024 : 256 * waves, iters = 200;
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
    printf("%-16s %d wave(s)/SIMD: %6.2f cycles per instruction per wave; %6.2f cycles of SIMD time per instruction\n", name, waves,
           avg / (iters * 64.0), avg / (iters * 64.0) / waves);
  }
}
int main() {
  run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<12>("v_mul_f32"); run<13>("v_pk_mul_f32"); run<2>("v_fma_f64"); run<3>("v_add_f64"); run<4>("v_mul_f64");
  run<5>("v_cvt_f64_f32"); run<6>("v_cvt_f32_f64"); run<7>("v_rndne_f64"); run<8>("v_ldexp_f64");
  run<9>("v_sin_f32"); run<10>("v_exp_f32"); run<11>("v_rcp_f32"); run<14>("v_cvt_i32_f32"); run<15>("v_mov_b32"); run<16>("v_mov_b32_dpp"); run<17>("v_med3_f32");
  return 0;
}
