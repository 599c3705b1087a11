// Wave64 scan / reduction helpers shared by the Voice forward and backward kernels (device only).
#pragma once
#include <hip/hip_runtime.h>

// Inclusive wave64 scan of doubles with DPP moves (no LDS crossbar, no selects): Hillis-Steele inside
// each row of 16 lanes (row_shr:d shifts zeros in), then row_bcast:15 / row_bcast:31 carry the row totals
// across rows.  Every step adds +0.0 where nothing arrives, which is exact.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
  (void)lane;
  v += dpp_move<0x111, 0xf>(v);   // row_shr:1
  v += dpp_move<0x112, 0xf>(v);   // row_shr:2
  v += dpp_move<0x114, 0xf>(v);   // row_shr:4
  v += dpp_move<0x118, 0xf>(v);   // row_shr:8
  v += dpp_move<0x142, 0xa>(v);   // row_bcast:15 -> rows 1 and 3
  v += dpp_move<0x143, 0xc>(v);   // row_bcast:31 -> rows 2 and 3
  return v;
}
// wave64 sum, the same value in every lane: the DPP scan's last lane, read back with v_readlane (a butterfly of
// __shfl_xor on doubles is 12 dependent ds_bpermute round trips)
__device__ __forceinline__ double wave_sum(double v) {
  v = wave_incl_scan(v, 0);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}

