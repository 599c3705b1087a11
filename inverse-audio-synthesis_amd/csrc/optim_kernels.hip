// LARS update as three multi-tensor launches (MI355X / gfx950).
//
// Replaces the optimizer step of /root/reference/vicreg_audio_params.py:134-151 (flash.core.optimizers.LARS with
// momentum 0: trust ratio ||p|| / (||g|| + wd ||p|| + eps) * trust_coefficient on tensors whose two norms are non-zero,
// weight decay inside the ratio, then p -= lr * update).  As torch._foreach ops the step costs two norm passes, four
// read-modify-write passes over all parameters and -- because the per-tensor ratios are device scalars -- one small
// multiply launch per tensor (369 launches, 2.4 ms of a 36 ms pretraining step at B = 128).  Here: one pass for the
// norms, one tiny pass for the coefficients, one read-modify-write pass for the update; every sum in a fixed order.
//
// Tables (device, built once per parameter set by the caller):
//   tensors     [n][3] int64 : parameter pointer, gradient pointer, element count
//   chunks      [nchunks][2] int32 : tensor index, chunk index inside the tensor (chunks of IAS_LARS_CHUNK elements)
//   first_chunk [n + 1] int32 : first chunk of each tensor (prefix sums)
#include "ias_common.h"
#include <cstdint>

#define IAS_LARS_CHUNK 65536
#define LARS_THREADS 256

// GRAD_ONLY (round 5): the parameters' partial sums are carried over from the previous step's update pass (lars_update_kernel
// writes sum p_new^2 per chunk, in THIS kernel's order of additions, into partials[2 c]); only the gradient is read here.
template <bool GRAD_ONLY>
__global__ __launch_bounds__(LARS_THREADS) void lars_norm_partials_kernel(const long long* __restrict__ tensors,
                                                                           const int* __restrict__ chunks,
                                                                           double* __restrict__ partials) {
  const int c = blockIdx.x, t = chunks[2 * c], ci = chunks[2 * c + 1];
  const float* p = reinterpret_cast<const float*>(tensors[3 * t]);
  const float* g = reinterpret_cast<const float*>(tensors[3 * t + 1]);
  const long long n = tensors[3 * t + 2];
  const long long off = (long long)ci * IAS_LARS_CHUNK;
  const int len = (int)(n - off < IAS_LARS_CHUNK ? n - off : IAS_LARS_CHUNK);
  p += off; g += off;
  float sp = 0.0f, sg = 0.0f;
  const int tid = threadIdx.x;
  if ((((uintptr_t)p | (uintptr_t)g) & 15) == 0) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const int n4 = len >> 2;
    // four 16-byte loads of each operand in flight per thread (a full chunk is 16 rounds of 4); the sums keep the order
    // of the one-load-per-round loop
    int i = tid;
    for (; i + 3 * LARS_THREADS < n4; i += 4 * LARS_THREADS) {
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { if (!GRAD_ONLY) a[u] = p4[i + u * LARS_THREADS]; b[u] = g4[i + u * LARS_THREADS]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!GRAD_ONLY) { sp = fmaf(a[u].x, a[u].x, sp); sp = fmaf(a[u].y, a[u].y, sp); sp = fmaf(a[u].z, a[u].z, sp); sp = fmaf(a[u].w, a[u].w, sp); }
        sg = fmaf(b[u].x, b[u].x, sg); sg = fmaf(b[u].y, b[u].y, sg); sg = fmaf(b[u].z, b[u].z, sg); sg = fmaf(b[u].w, b[u].w, sg);
      }
    }
    for (; i < n4; i += LARS_THREADS) {
      const float4 b = g4[i];
      if (!GRAD_ONLY) { const float4 a = p4[i]; sp = fmaf(a.x, a.x, sp); sp = fmaf(a.y, a.y, sp); sp = fmaf(a.z, a.z, sp); sp = fmaf(a.w, a.w, sp); }
      sg = fmaf(b.x, b.x, sg); sg = fmaf(b.y, b.y, sg); sg = fmaf(b.z, b.z, sg); sg = fmaf(b.w, b.w, sg);
    }
    for (int i = (n4 << 2) + tid; i < len; i += LARS_THREADS) { if (!GRAD_ONLY) sp = fmaf(p[i], p[i], sp); sg = fmaf(g[i], g[i], sg); }
  } else {
    for (int i = tid; i < len; i += LARS_THREADS) { if (!GRAD_ONLY) sp = fmaf(p[i], p[i], sp); sg = fmaf(g[i], g[i], sg); }
  }
  __shared__ double s_p[LARS_THREADS], s_g[LARS_THREADS];
  s_p[tid] = (double)sp; s_g[tid] = (double)sg;
  __syncthreads();
#pragma unroll
  for (int d = LARS_THREADS / 2; d > 0; d >>= 1) {
    if (tid < d) { s_p[tid] += s_p[tid + d]; s_g[tid] += s_g[tid + d]; }
    __syncthreads();
  }
  if (tid == 0) { if (!GRAD_ONLY) partials[2 * c] = s_p[0]; partials[2 * c + 1] = s_g[0]; }
}

// coef[t] = (ratio, ratio * wd) where both norms are non-zero, (1, 0) elsewhere   (hyper = lr, wd, trust, eps)
// One wave per tensor: lane l adds the partials of chunks l, l + 64, ... in order, then a butterfly -- fixed order, and the
// 1024 chunks of an 8192 x 8192 weight are 16 rounds of loads instead of 1024 dependent ones on one thread (142 -> ~5 us).
__global__ __launch_bounds__(64) void lars_coef_kernel(const int* __restrict__ first_chunk, const double* __restrict__ partials,
                                                       const float* __restrict__ hyper, float* __restrict__ coef, int ntensors) {
  const int t = blockIdx.x, lane = threadIdx.x;
  if (t >= ntensors) return;
  double sp = 0.0, sg = 0.0;
  const int c1 = first_chunk[t + 1];
  for (int c = first_chunk[t] + lane; c < c1; c += 64) { sp += partials[2 * c]; sg += partials[2 * c + 1]; }
  for (int d = 32; d > 0; d >>= 1) { sp += __shfl_xor(sp, d, 64); sg += __shfl_xor(sg, d, 64); }
  if (lane != 0) return;
  const float pn = (float)sqrt(sp), gn = (float)sqrt(sg);
  const float wd = hyper[1], trust = hyper[2], eps = hyper[3];
  float ratio = 1.0f, decay = 0.0f;
  if (pn != 0.0f && gn != 0.0f) {
    ratio = pn / (gn + pn * wd + eps) * trust;
    decay = ratio * wd;
  }
  coef[2 * t] = ratio; coef[2 * t + 1] = decay;
}

// CARRY (round 5): the pass also sums the squares of the values it writes -- element by element in the order in which
// lars_norm_partials_kernel adds them, through the same workgroup reduction -- into partials[2 c]: the next step's norm pass
// then reads the gradient only (570 MB less per step of the 142 M-parameter model), and gets the bits it would have computed.
template <bool CARRY>
__global__ __launch_bounds__(LARS_THREADS) void lars_update_kernel(const long long* __restrict__ tensors,
                                                                    const int* __restrict__ chunks,
                                                                    const float* __restrict__ coef,
                                                                    const float* __restrict__ hyper,
                                                                    double* __restrict__ partials) {
  // chunks in DESCENDING order: the norm pass in front read the gradients in ascending order, so the last ~200 MB it
  // touched are what the 256 MB memory-side cache still holds -- the update starts with those
  const int c = gridDim.x - 1 - blockIdx.x, t = chunks[2 * c], ci = chunks[2 * c + 1];
  float* p = reinterpret_cast<float*>(tensors[3 * t]);
  const float* g = reinterpret_cast<const float*>(tensors[3 * t + 1]);
  const long long n = tensors[3 * t + 2];
  const long long off = (long long)ci * IAS_LARS_CHUNK;
  const int len = (int)(n - off < IAS_LARS_CHUNK ? n - off : IAS_LARS_CHUNK);
  p += off; g += off;
  const float ratio = coef[2 * t], decay = coef[2 * t + 1], nlr = -hyper[0];
  const int tid = threadIdx.x;
  float sp = 0.0f;
  // update = ratio * g + decay * p ;  p += (-lr) * update     (the order of torch's foreach formulation)
  if ((((uintptr_t)p | (uintptr_t)g) & 15) == 0) {
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const int n4 = len >> 2;
    auto upd = [&](float4 a, const float4 b) {
      a.x = fmaf(nlr, fmaf(decay, a.x, ratio * b.x), a.x);
      a.y = fmaf(nlr, fmaf(decay, a.y, ratio * b.y), a.y);
      a.z = fmaf(nlr, fmaf(decay, a.z, ratio * b.z), a.z);
      a.w = fmaf(nlr, fmaf(decay, a.w, ratio * b.w), a.w);
      return a;
    };
    int i = tid;
    for (; i + 3 * LARS_THREADS < n4; i += 4 * LARS_THREADS) {    // four loads of each operand in flight per thread
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = p4[i + u * LARS_THREADS]; b[u] = g4[i + u * LARS_THREADS]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 n = upd(a[u], b[u]);
        p4[i + u * LARS_THREADS] = n;
        if (CARRY) { sp = fmaf(n.x, n.x, sp); sp = fmaf(n.y, n.y, sp); sp = fmaf(n.z, n.z, sp); sp = fmaf(n.w, n.w, sp); }
      }
    }
    for (; i < n4; i += LARS_THREADS) {
      const float4 n = upd(p4[i], g4[i]);
      p4[i] = n;
      if (CARRY) { sp = fmaf(n.x, n.x, sp); sp = fmaf(n.y, n.y, sp); sp = fmaf(n.z, n.z, sp); sp = fmaf(n.w, n.w, sp); }
    }
    for (int i = (n4 << 2) + tid; i < len; i += LARS_THREADS) {
      const float n = fmaf(nlr, fmaf(decay, p[i], ratio * g[i]), p[i]);
      p[i] = n;
      if (CARRY) sp = fmaf(n, n, sp);
    }
  } else {
    for (int i = tid; i < len; i += LARS_THREADS) {
      const float n = fmaf(nlr, fmaf(decay, p[i], ratio * g[i]), p[i]);
      p[i] = n;
      if (CARRY) sp = fmaf(n, n, sp);
    }
  }
  if (CARRY) {
    __shared__ double s_p[LARS_THREADS];
    s_p[tid] = (double)sp;
    __syncthreads();
#pragma unroll
    for (int d = LARS_THREADS / 2; d > 0; d >>= 1) {
      if (tid < d) s_p[tid] += s_p[tid + d];
      __syncthreads();
    }
    if (tid == 0) partials[2 * c] = s_p[0];
  }
}

// ------------------------------------------------------------------------ C ABI
extern "C" int ias_lars_chunk_elems(void) { return IAS_LARS_CHUNK; }

// One LARS step (momentum 0) over ntensors parameter tensors.  hyper [4] (device floats): lr, weight_decay,
// trust_coefficient, eps -- on the device so that a captured graph picks up the scheduler's learning rate.
// partials [nchunks][2] doubles and coef [ntensors][2] floats are caller-owned scratch.  weight_decay 0 is the caller's
// business (plain p -= lr g: set coef to (1, 0) and pass skip_norms != 0).
static int lars_step_impl(const long long* tensors, const int* chunks, const int* first_chunk, double* partials,
                          float* coef, const float* hyper, int ntensors, int nchunks, int skip_norms, int carry,
                          void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!tensors || !chunks || !first_chunk || !partials || !coef || !hyper || ntensors <= 0 || nchunks <= 0 || carry < 0 || carry > 2)
    return IAS_ERR_ARG;
  if (!skip_norms) {
    if (carry == 2)
      hipLaunchKernelGGL(lars_norm_partials_kernel<true>, dim3(nchunks), dim3(LARS_THREADS), 0, stream, tensors, chunks, partials);
    else
      hipLaunchKernelGGL(lars_norm_partials_kernel<false>, dim3(nchunks), dim3(LARS_THREADS), 0, stream, tensors, chunks, partials);
    hipLaunchKernelGGL(lars_coef_kernel, dim3(ntensors), dim3(64), 0, stream, first_chunk, partials,
                       hyper, coef, ntensors);
  }
  if (carry != 0 && !skip_norms)
    hipLaunchKernelGGL(lars_update_kernel<true>, dim3(nchunks), dim3(LARS_THREADS), 0, stream, tensors, chunks, coef, hyper, partials);
  else
    hipLaunchKernelGGL(lars_update_kernel<false>, dim3(nchunks), dim3(LARS_THREADS), 0, stream, tensors, chunks, coef, hyper, partials);
  return hipGetLastError() == hipSuccess ? IAS_OK : IAS_ERR_LAUNCH;
}

extern "C" int ias_lars_step(const long long* tensors, const int* chunks, const int* first_chunk, double* partials,
                             float* coef, const float* hyper, int ntensors, int nchunks, int skip_norms,
                             void* stream_) {
  return lars_step_impl(tensors, chunks, first_chunk, partials, coef, hyper, ntensors, nchunks, skip_norms, 0, stream_);
}

// The same step with the parameters' norms carried from update to update.  carry = 1: norms of p and g as in ias_lars_step,
// and the update pass leaves sum p_new^2 per chunk in partials[2 c]; carry = 2: the norm pass reads the gradient only and
// takes the parameters' sums that the previous call (carry 1 or 2, same tensors, same partials buffer, parameters not
// modified in between) left there.  Bit-identical coefficients and parameters to ias_lars_step.  The caller keeps
// `partials` alive and unmodified between the calls (inside a captured step: outside the graph's memory pool).
extern "C" int ias_lars_step_carry(const long long* tensors, const int* chunks, const int* first_chunk, double* partials,
                                   float* coef, const float* hyper, int ntensors, int nchunks, int carry, void* stream_) {
  if (carry != 1 && carry != 2) return IAS_ERR_ARG;
  return lars_step_impl(tensors, chunks, first_chunk, partials, coef, hyper, ntensors, nchunks, 0, carry, stream_);
}
