// Framed STFT (n_fft 1024) -> power / magnitude -> (optional) mel filterbank -> spectrogram or fused loss sums: the
// round-3 form of the wave-per-frame radix-8 FFT kernel (csrc/spectral_kernels.hip: stft_kernel, which keeps serving
// n_fft 512 / 2048 and the backward's FFT core).  Same contract and arithmetic.
//
// Spec: the commented mel block /root/reference/conf/config.yaml:51-61, its use /root/reference/audio_to_params.py:150-153
// (torchaudio MelSpectrogram semantics) and the auraloss TODO audio_to_params.py:233.
//
// What changed against stft_kernel, and why (profiles/r02f_pmc_stft_lds.txt: 38 % of the kernel's LDS time were bank
// conflicts, the mel projection 42 % of the LDS time for 4 % of the flops; csrc/stft_mfma_kernels.hip: the fp32 matrix
// cores are not a second pipe beside the vector ALU, so the work has to shrink, not move):
//   * mel projection without gathers at per-filter offsets.  A triangular filterbank puts every bin under at most two
//     ADJACENT filters, so the bins split into contiguous segments (between two filter centres) and
//         mel[m] = U[m] + D[m+1],   U[j] = sum over segment j of up_k P_k,   D[j] = sum over segment j of down_k P_k.
//     The power values are stored segment-major ([position in segment][segment], row stride 72 floats): a lane that owns
//     segment j then reads position t of all segments as one conflict-free row, with the two weights of that bin as one
//     8-byte read from a table of the same shape.  Each bin is read once (513 reads per frame instead of ~1100 at
//     per-lane offsets), U/D meet through one DPP wave shift.
//   * Hermitian unpack on half the spectrum: after the last radix-8 pass a lane holds Z[k], k = k1 + 8 d + 64 e; the
//     e < 4 values (k < 256) stay in registers, only the e >= 4 half goes to LDS, and the lane pairs its own four bins
//     with Z[512 - k] from there (4 stores + 4 loads instead of 8 + 10).
//   * frames are dealt to waves from one flat [B*F] list (wave w of all: frames w, w + W, ...): no per-row grid, no
//     ragged last workgroup per row, any workgroup size (ten waves: the frame-invariant tables are shared by more waves).
#include "ias_common.h"
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

// ------------------------------------------------------------------------------------- host: segment-major mel tables
// Block layout (32-bit words): [0..15] header {magic, rows, rA, rB, rC, s0, nseg, n_out, bins},
// [16 .. 16 + 9*64) the lanes' store offsets (floats into the segment-major buffer) of their nine bins: entries e < 4 the
// own bins k = (lane >> 3) + 8 (lane & 7) + 64 e, entries 4 + e the partners 512 - k, entry 8 bin 256 (lane 0),
// then rows x 64 x (up, down) weights.  Slot i = segment s0 + i lives in group i / 64 (A, B, C) on lane i % 64; row
// r of group g holds position r - base(g) of the group's segments.
#define IAS_SEG_MAGIC 0x5e67ab
#define IAS_SEG_MAX_ROWS 17          // rows x 72 floats fit the wave's 5120-byte FFT scratch (spectral_kernels.hip: IAS_S2_ROW)
#define IAS_SEG_HDR 16
#define IAS_SEG_STRIDE 72

struct SegPlan {
  int rows, r[3], s0, nseg;
  std::vector<int> seg_of_bin, pos_of_bin;       // slot and position per bin (-1: bin carries no weight)
  std::vector<float> up, down;                   // per bin
};

static bool seg_plan(int n_fft, const int* mel_start, const int* mel_count, const int* mel_woff, const float* mel_w,
                     int n_out, SegPlan& P) {
  if (n_fft != 1024 || n_out <= 0 || n_out > 192) return false;
  const int bins = n_fft / 2 + 1;
  // per bin: the (at most two, adjacent) filters with a non-zero weight
  std::vector<int> first(bins, -1), cnt(bins, 0);
  std::vector<int> peak(n_out, 0);
  auto W = [&](int m, int k) -> float {
    return (k >= mel_start[m] && k < mel_start[m] + mel_count[m]) ? mel_w[mel_woff[m] + k - mel_start[m]] : 0.0f;
  };
  for (int m = 0; m < n_out; ++m) {
    float best = -1.0f;
    for (int k = mel_start[m]; k < mel_start[m] + mel_count[m]; ++k) {
      const float w = W(m, k);
      if (w < 0.0f) return false;
      if (w > best) { best = w; peak[m] = k; }
      if (w != 0.0f) {
        if (cnt[k] == 0) first[k] = m;
        else if (m != first[k] + cnt[k]) return false;          // not adjacent
        if (++cnt[k] > 2) return false;
      }
    }
  }
  // segment (global numbering j = 0 .. n_out) of every weighted bin
  std::vector<int> jk(bins, -1);
  for (int k = 0; k < bins; ++k) {
    if (cnt[k] == 2) jk[k] = first[k] + 1;
    else if (cnt[k] == 1) jk[k] = k <= peak[first[k]] ? first[k] : first[k] + 1;
  }
  int prev = -1, jmin = 1 << 30, jmax = -1;
  for (int k = 0; k < bins; ++k) {
    if (jk[k] < 0) continue;
    if (jk[k] < prev) return false;                             // not monotone: not a triangular filterbank
    prev = jk[k];
    jmin = std::min(jmin, jk[k]); jmax = std::max(jmax, jk[k]);
  }
  if (jmax < 0) return false;
  P.s0 = jmin >= 1 ? 1 : 0;
  P.nseg = jmax - P.s0 + 1;
  if (P.nseg > 192) return false;
  P.seg_of_bin.assign(bins, -1); P.pos_of_bin.assign(bins, 0);
  P.up.assign(bins, 0.0f); P.down.assign(bins, 0.0f);
  std::vector<int> len(P.nseg, 0);
  for (int k = 0; k < bins; ++k) {
    if (jk[k] < 0) continue;
    const int j = jk[k], i = j - P.s0;
    P.seg_of_bin[k] = i; P.pos_of_bin[k] = len[i]++;
    P.up[k] = j < n_out ? W(j, k) : 0.0f;
    P.down[k] = j >= 1 ? W(j - 1, k) : 0.0f;
    // every weight of the bin is accounted for
    for (int m = std::max(0, j - 2); m < std::min(n_out, j + 2); ++m)
      if (m != j && m != j - 1 && W(m, k) != 0.0f) return false;
  }
  P.rows = 0;
  for (int g = 0; g < 3; ++g) {
    P.r[g] = 0;
    for (int i = 64 * g; i < std::min(P.nseg, 64 * g + 64); ++i) P.r[g] = std::max(P.r[g], len[i]);
    P.rows += P.r[g];
  }
  return P.rows >= 1 && P.rows <= IAS_SEG_MAX_ROWS;
}

extern "C" long long ias_stft_segtab_len(int n_fft, const int* mel_start_host, const int* mel_count_host,
                                         const int* mel_woff_host, const float* mel_w_host, int n_out) {
  if (!mel_start_host || !mel_count_host || !mel_woff_host || !mel_w_host) return IAS_ERR_ARG;
  SegPlan P;
  if (!seg_plan(n_fft, mel_start_host, mel_count_host, mel_woff_host, mel_w_host, n_out, P)) return IAS_ERR_UNSUPPORTED;
  return IAS_SEG_HDR + 9 * 64 + (long long)P.rows * 128;
}

extern "C" int ias_stft_build_segtab(int n_fft, const int* mel_start_host, const int* mel_count_host,
                                     const int* mel_woff_host, const float* mel_w_host, int n_out, float* out_host) {
  const long long len = ias_stft_segtab_len(n_fft, mel_start_host, mel_count_host, mel_woff_host, mel_w_host, n_out);
  if (len < 0) return (int)len;
  if (!out_host) return IAS_ERR_ARG;
  SegPlan P;
  seg_plan(n_fft, mel_start_host, mel_count_host, mel_woff_host, mel_w_host, n_out, P);
  std::memset(out_host, 0, sizeof(float) * (size_t)len);
  int* hdr = reinterpret_cast<int*>(out_host);
  hdr[0] = IAS_SEG_MAGIC; hdr[1] = P.rows; hdr[2] = P.r[0]; hdr[3] = P.r[1]; hdr[4] = P.r[2]; hdr[5] = P.s0;
  hdr[6] = P.nseg; hdr[7] = n_out; hdr[8] = n_fft / 2 + 1;
  const int base[3] = {0, P.r[0], P.r[0] + P.r[1]};
  auto addr_of = [&](int k) {
    const int i = P.seg_of_bin[k];
    if (i < 0) return IAS_SEG_MAX_ROWS * IAS_SEG_STRIDE + 0;    // unweighted bins: a dump word behind the rows
    return (base[i >> 6] + P.pos_of_bin[k]) * IAS_SEG_STRIDE + (i & 63);
  };
  int* addr = hdr + IAS_SEG_HDR;
  float* wt = out_host + IAS_SEG_HDR + 9 * 64;
  for (int l = 0; l < 64; ++l) {
    for (int e = 0; e < 4; ++e) {
      const int k = (l >> 3) + 8 * (l & 7) + 64 * e;
      addr[64 * e + l] = addr_of(k);
      addr[64 * (4 + e) + l] = addr_of(512 - k);
    }
    addr[64 * 8 + l] = addr_of(256);
  }
  for (int k = 0; k < n_fft / 2 + 1; ++k) {
    const int i = P.seg_of_bin[k];
    if (i < 0) continue;
    const int r = base[i >> 6] + P.pos_of_bin[k];
    wt[(r * 64 + (i & 63)) * 2] = P.up[k];
    wt[(r * 64 + (i & 63)) * 2 + 1] = P.down[k];
  }
  return IAS_OK;
}
