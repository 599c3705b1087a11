// Common definitions for the inverse-audio-synthesis MI355X (gfx950) kernels.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define IAS_HD __host__ __device__ __forceinline__
#else
// Host-only build of the per-sample arithmetic (used by tests/ to check the
// device math against the oracle without a GPU; never part of the product path).
#define IAS_HD inline
#endif

// Diagnostic switches.  The product library (libias_hip.so) has one kernel per operation and shape and reads NOTHING from
// the environment: there ias_diag_env() is the constant NULL, so every alternative behind it -- and the kernels only such
// an alternative reaches, which are additionally fenced by #ifdef IAS_DIAG -- is compiled out, and the library keeps no
// mutable global state.  `make diag` builds libias_hip_diag.so from the same sources with -DIAS_DIAG, where
// the switches are live: scripts/diag, bench.py's `dxd` comparison figure, and the tests that compare a superseded kernel
// with its replacement load THAT library (inverse-audio-synthesis_amd/_lib.py: load_diag / use_library).
#ifdef IAS_DIAG
#include <stdlib.h>
static inline const char* ias_diag_env(const char* name) { return getenv(name); }
#else
#define ias_diag_env(name) ((const char*)0)
#endif

// Error codes returned by every C-ABI entry point.
#define IAS_OK 0
#define IAS_ERR_ARG (-1)        // bad pointer / dimension
#define IAS_ERR_UNSUPPORTED (-2)
#define IAS_ERR_LAUNCH (-3)     // HIP launch failure (hipGetLastError != success)
#define IAS_ERR_WORKSPACE (-4)  // workspace too small

#define IAS_NPARAMS 78
// torchsynth LFO shape-mix exponent: LFO.__init__'s default `exponent = tensor(e)`, i.e. the fp32 0x402DF854
#define IAS_LFO_EXPONENT_F 2.7182817459106445f
#define IAS_NCTRL 5             // mod-matrix outputs: vco1 pitch, vco1 amp, vco2 pitch, vco2 amp, noise amp

// One entry of ias_reduce_partials_multi's table (include/ias_hip.h): out[i] = sum_r partial[r n + i], i < n.
struct IasReduceItem {
  const float* partial;
  float* out;
  int n, rows;
};

// Per-voice scalars produced by the control-rate kernel, consumed at audio rate.
struct IasVoiceConst {
  float f0_1, depth_1, phi_1;   // vco_1: fl(midi_f0 + tuning), mod_depth, initial_phase
  float f0_2, depth_2, phi_2;   // vco_2
  float kpart;                  // fl(pi_f32 * partials_constant)
  float shape;                  // vco_2 shape
  float shape_gain;             // fl(1 - shape/2)
  float lvl0, lvl1, lvl2;       // mixer levels: vco_1, vco_2, noise
  float pad[4];
};
