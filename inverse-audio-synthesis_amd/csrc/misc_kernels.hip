// Library version + a float4 streaming-copy kernel used by bench.py to calibrate the
// achievable HBM rate on the box it runs on (MI355X_MICROARCH: 6.29 TB/s float4 copy).
#include "ias_common.h"

#define IAS_VERSION 100

extern "C" int ias_version(void) { return IAS_VERSION; }

__global__ __launch_bounds__(256) void copy_f4_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                      long long n4) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

extern "C" int ias_stream_copy(const float* src, float* dst, long long n, void* stream_) {
  if (!src || !dst || n <= 0 || (n & 3)) return IAS_ERR_ARG;
  hipLaunchKernelGGL(copy_f4_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream_, (const float4*)src,
                     (float4*)dst, n / 4);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

// One device timestamp (s_memrealtime: the constant 100 MHz counter, 10 ns per tick) into dst[0], as a stream-ordered
// launch: bench.py brackets the stages of the step with these INSIDE a captured graph (HIP events cannot be read back
// from a replayed graph), so the in-step durations it reports belong to the replayed schedule it times.
__global__ void stamp_kernel(unsigned long long* dst) { dst[0] = __builtin_amdgcn_s_memrealtime(); }
extern "C" int ias_stamp(unsigned long long* dst, void* stream_) {
  if (!dst) return IAS_ERR_ARG;
  hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_, dst);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}
