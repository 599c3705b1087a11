// What v_permlane32_swap / v_permlane16_swap and the bank-masked DPP row moves do on gfx950, lane by lane, and the
// register <-> lane-bits transpose built from them (the FFT exchange of csrc/spectral_kernels.hip) against the same
// transpose through LDS.  hipcc --offload-arch=gfx950 -O3 permlane_swap.hip -o permlane_swap
#include <cstdio>
#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));

// transpose register bit b (pairs r, r | 1 << b) with lane bit L in {3, 4, 5} of 8 registers
template <int L> __device__ __forceinline__ void swap_pair(unsigned& r0, unsigned& r1) {
  if (L == 5) { const u2 t = __builtin_amdgcn_permlane32_swap(r0, r1, false, false); r0 = t.x; r1 = t.y; }
  else if (L == 4) { const u2 t = __builtin_amdgcn_permlane16_swap(r0, r1, false, false); r0 = t.x; r1 = t.y; }
  else {
    // lanes with bit 3 set (banks 2, 3 of a row of 16): r0 <- r1 of lane ^ 8; lanes with bit 3 clear: r1 <- r0 of lane ^ 8
    const unsigned old0 = r0;
    r0 = __builtin_amdgcn_update_dpp(r0, r1, 0x128 /* row_ror:8 */, 0xf, 0xc, false);
    r1 = __builtin_amdgcn_update_dpp(r1, old0, 0x128, 0xf, 0x3, false);
  }
}
__global__ void k(unsigned* out) {
  const int lane = threadIdx.x;
  __shared__ unsigned s[64 * 9];
  unsigned v[8], ref[8];
  for (int q = 0; q < 8; ++q) v[q] = 1000 * lane + q;           // element (lane = 8 a + c, reg = k1): value 1000 lane + k1
  // reference through LDS: new lane (k1, c) reg a  <-  old lane (a, c) reg k1
  for (int q = 0; q < 8; ++q) s[((q * 8 + (lane & 7))) * 9 + (lane >> 3)] = v[q];
  __syncthreads();
  for (int q = 0; q < 8; ++q) ref[q] = s[lane * 9 + q];
  // registers: reg bit 0 <-> lane bit 3, reg bit 1 <-> lane bit 4, reg bit 2 <-> lane bit 5
  swap_pair<3>(v[0], v[1]); swap_pair<3>(v[2], v[3]); swap_pair<3>(v[4], v[5]); swap_pair<3>(v[6], v[7]);
  swap_pair<4>(v[0], v[2]); swap_pair<4>(v[1], v[3]); swap_pair<4>(v[4], v[6]); swap_pair<4>(v[5], v[7]);
  swap_pair<5>(v[0], v[4]); swap_pair<5>(v[1], v[5]); swap_pair<5>(v[2], v[6]); swap_pair<5>(v[3], v[7]);
  for (int q = 0; q < 8; ++q) { out[lane * 8 + q] = v[q]; out[512 + lane * 8 + q] = ref[q]; }
}
int main() {
  unsigned* d; hipMalloc(&d, 1024 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 512; ++i) bad += h[i] != h[512 + i];
  printf("register transpose vs LDS transpose: %d mismatches of 512\n", bad);
  for (int l : {0, 1, 8, 9, 16, 32, 63}) { printf("lane %2d:", l); for (int q = 0; q < 8; ++q) printf(" %5u/%5u", h[l * 8 + q], h[512 + l * 8 + q]); printf("\n"); }
  return bad != 0;
}
