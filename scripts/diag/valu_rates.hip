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
  double sp = __builtin_amdgcn_readfirstlane(iters) * 1e-9 + 1.0;   // uniform -> SGPR pair
  float sf = __builtin_amdgcn_readfirstlane(iters) * 1e-9f + 1.0f;
  sp = __longlong_as_double((((long long)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(sp) >> 32))) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)__double_as_longlong(sp)));
  double q0 = da, q1 = da + 1, q2 = da + 2, q3 = da + 3, q4 = da + 4, q5 = da + 5, q6 = da + 6, q7 = da + 7;
  float f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4 = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
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
    if (OP == 18) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(da) : "v"(db), "s"(sp));) }
    if (OP == 19) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(da) : "v"(db), "s"(sp));) }
    if (OP == 20) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(da) : "v"(db), "s"(sp));) }
    if (OP == 21) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(da) : "v"(db), "v"(dc));) }
    if (OP == 22) { REP8(asm volatile("v_pk_fma_f32 %0, %8, %9, %0\nv_pk_fma_f32 %1, %8, %9, %1\nv_pk_fma_f32 %2, %8, %9, %2\nv_pk_fma_f32 %3, %8, %9, %3\nv_pk_fma_f32 %4, %8, %9, %4\nv_pk_fma_f32 %5, %8, %9, %5\nv_pk_fma_f32 %6, %8, %9, %6\nv_pk_fma_f32 %7, %8, %9, %7" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(db), "v"(dc));) }
    if (OP == 23) { REP8(asm volatile("v_pk_fma_f32 %0, %8, %9, %0 op_sel_hi:[1,0,1]\nv_pk_fma_f32 %1, %8, %9, %1 op_sel_hi:[1,0,1]\nv_pk_fma_f32 %2, %8, %9, %2 op_sel_hi:[1,0,1]\nv_pk_fma_f32 %3, %8, %9, %3 op_sel_hi:[1,0,1]\nv_pk_fma_f32 %4, %8, %9, %4 op_sel_hi:[1,0,1]\nv_pk_fma_f32 %5, %8, %9, %5 op_sel_hi:[1,0,1]\nv_pk_fma_f32 %6, %8, %9, %6 op_sel_hi:[1,0,1]\nv_pk_fma_f32 %7, %8, %9, %7 op_sel_hi:[1,0,1]" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7) : "v"(db), "s"(sp));) }
    if (OP == 24) { REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\nv_fma_f32 %1, %8, %9, %1\nv_fma_f32 %2, %8, %9, %2\nv_fma_f32 %3, %8, %9, %3\nv_fma_f32 %4, %8, %9, %4\nv_fma_f32 %5, %8, %9, %5\nv_fma_f32 %6, %8, %9, %6\nv_fma_f32 %7, %8, %9, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "s"(sf));) }
    if (OP == 25) { REP64(asm volatile("v_add_f32 %0, %1, %2" : "=v"(b) : "v"(a), "v"(c));) }
    if (OP == 26) { REP64(asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(db) : "v"(da), "v"(dc));) }
    if (OP == 27) { REP64(asm volatile("v_add_u32 %0, %1, %2" : "=v"(b) : "v"(a), "v"(c));) }
    if (OP == 28) { REP64(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(b) : "v"(a), "v"(c));) }
    if (OP == 29) { REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(c) : "vcc");) }
    if (OP == 30) { REP64(asm volatile("v_fract_f32 %0, %1" : "=v"(b) : "v"(a));) }
    if (OP == 31) { REP64(asm volatile("v_floor_f32 %0, %1" : "=v"(b) : "v"(a));) }
    if (OP == 32) { REP64(asm volatile("v_cos_f32 %0, %1" : "=v"(b) : "v"(a));) }
    if (OP == 33) { REP64(asm volatile("v_log_f32 %0, %1" : "=v"(b) : "v"(a));) }
    if (OP == 34) { REP64(asm volatile("v_sqrt_f32 %0, %1" : "=v"(b) : "v"(a));) }
    if (OP == 35) { REP64(asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(da) : "v"(a));) }
    if (OP == 36) { REP64(asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a) : "v"(da));) }
    if (OP == 37) { REP64(asm volatile("v_fract_f64 %0, %1" : "=v"(db) : "v"(da));) }
    if (OP == 38) { REP64(asm volatile("v_floor_f64 %0, %1" : "=v"(db) : "v"(da));) }
    if (OP == 39) { REP64(asm volatile("v_lshlrev_b64 %0, 3, %1" : "=v"(db) : "v"(da));) }
    if (OP == 40) { REP64(asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(b) : "v"(a), "v"(c));) }
    if (OP == 41) { REP64(asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(b) : "v"(a), "v"(c));) }
    if (OP == 42) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(da) : "v"(a), "v"(c) : "vcc");) }
    if (OP == 43) { REP64(asm volatile("v_mul_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(a), "v"(c));) }
    if (OP == 44) { REP64(asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a) : "s20");) }
    if (OP == 45) { REP64(asm volatile("v_mov_b64 %0, %1" : "=v"(db) : "v"(da));) }
    if (OP == 46) { REP64(asm volatile("v_pk_mov_b32 %0, %1, %2" : "=v"(db) : "v"(da), "v"(dc));) }
    if (OP == 47) { REP64(asm volatile("v_add_f64 %0, %1, %2" : "=v"(q0) : "v"(da), "v"(dc));) }
    if (OP == 48) { REP64(asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(q0) : "v"(da), "v"(dc), "v"(db));) }
    if (OP == 49) { REP64(asm volatile("v_mul_f64 %0, %1, %2" : "=v"(q0) : "v"(da), "v"(dc));) }
    if (OP == 50) { REP64(asm volatile("v_bfe_u32 %0, %1, 3, 5" : "=v"(b) : "v"(a));) }
    if (OP == 51) { REP64(asm volatile("v_and_b32 %0, %1, %2" : "=v"(b) : "v"(a), "v"(c));) }
    if (OP == 52) { REP64(asm volatile("v_rndne_f32 %0, %1" : "=v"(b) : "v"(a));) }
    if (OP == 53) { REP64(asm volatile("v_ldexp_f32 %0, %1, 3" : "=v"(b) : "v"(a));) }
    if (OP == 54) { REP64(asm volatile("v_max_f32 %0, %1, %2" : "=v"(b) : "v"(a), "v"(c));) }
    if (OP == 17) { REP64(asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(b) : "v"(a), "v"(c), "v"(c2));) }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + (float)(da + db + dc) + a2 + b2 + c2 + (float)(q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7) + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
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
// saturation: B blocks of 1024 threads per CU; wall time (HIP events) gives the tick length and the SIMD's true rate
// Saturated rate: 2 blocks of 1024 threads per CU (8 waves per SIMD), wall time by HIP events.
static double g_fma_ns = 0;
template <int OP> void sat(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4096 * 1024 * 4); hipMalloc(&cyc, 4096 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int per_cu = 2, iters = 1000, blocks = 256 * per_cu;
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, out, cyc, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, out, cyc, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double ns = ms * 1e6 / ((double)iters * 64.0 * 4 * per_cu);
  if (g_fma_ns == 0) g_fma_ns = ns;
  printf("%-18s %6.2f ns of SIMD time per wave64 instruction = %5.2f x v_fma_f32\n", name, ns, ns / g_fma_ns);
  hipFree(out); hipFree(cyc);
}
int main() {
  sat<0>("v_fma_f32"); sat<24>("v_fma_f32 sgpr x8"); sat<12>("v_mul_f32"); sat<25>("v_add_f32"); sat<54>("v_max_f32");
  sat<1>("v_pk_fma_f32"); sat<22>("v_pk_fma_f32 x8"); sat<23>("v_pk_fma sgpr opsel"); sat<13>("v_pk_mul_f32"); sat<26>("v_pk_add_f32");
  sat<15>("v_mov_b32"); sat<45>("v_mov_b64"); sat<46>("v_pk_mov_b32"); sat<16>("v_mov_b32_dpp"); sat<43>("v_mul_f32_dpp"); sat<44>("v_readlane_b32");
  sat<27>("v_add_u32"); sat<51>("v_and_b32"); sat<50>("v_bfe_u32"); sat<28>("v_cndmask_b32"); sat<29>("v_cmp_lt_f32"); sat<17>("v_med3_f32");
  sat<40>("v_mul_lo_u32"); sat<41>("v_mul_hi_u32"); sat<42>("v_mad_u64_u32"); sat<39>("v_lshlrev_b64");
  sat<30>("v_fract_f32"); sat<31>("v_floor_f32"); sat<52>("v_rndne_f32"); sat<53>("v_ldexp_f32"); sat<14>("v_cvt_i32_f32");
  sat<9>("v_sin_f32"); sat<32>("v_cos_f32"); sat<10>("v_exp_f32"); sat<33>("v_log_f32"); sat<11>("v_rcp_f32"); sat<34>("v_sqrt_f32");
  sat<48>("v_fma_f64"); sat<47>("v_add_f64"); sat<49>("v_mul_f64"); sat<5>("v_cvt_f64_f32"); sat<6>("v_cvt_f32_f64"); sat<35>("v_cvt_f64_i32"); sat<36>("v_cvt_i32_f64");
  sat<7>("v_rndne_f64"); sat<37>("v_fract_f64"); sat<38>("v_floor_f64"); sat<8>("v_ldexp_f64");
  return 0;
}
