// Layout of the constant block ("mtables") of the matrix-core STFT kernel (csrc/stft_mfma_kernels.hip), shared by the
// host-side builder and the kernel.  Everything here is a function of n_fft (and n_out for the mel part) alone, so
// the launcher never has to read the block back.
//
// Index conventions (N2 = n_fft/2 packed complex points z[m] = x[2m] + i x[2m+1], Q = N2/16):
//   m = Q n1 + n2  (n1 < 16, n2 < Q)          k = k1 + 16 k2   (k1 < 16, k2 < Q)
//   stage 1   S[n2][k1]  = sum_n1 z[Q n1 + n2] W_16^(n1 k1)            frame samples are the A operand, rows = n2
//   twiddle   S'[n2][k1] = S[n2][k1] W_N2^(n2 k1)
//   stage 2a  T[b][ka]   = sum_a S'[NB a + b][k1] W_8^(a ka)            (n2 = NB a + b, NB = Q/8) S' is the B operand
//   twiddle   T'[b][ka]  = T[b][ka] W_Q^(b ka)
//   stage 2b  Z[k1 + 16 (ka + 8 kb)] = sum_b T'[b][ka] W_NB^(b kb)      in-lane radix NB on the VALU
// v_mfma_f32_16x16x4_f32 lane maps: A[l&15][l>>4], B[l>>4][l&15], C/D col = l&15, row = 4 (l>>4) + reg.
#pragma once

#define IAS_SM_MAX_TILES 16   // mel tiles of 16 outputs: n_out <= 256
#define IAS_SM_REG_BLOCKS 12  // mel blocks whose weights a wave holds in registers at a time (one pass of its mel loop)
#define IAS_SM_MAX_BLOCKS 60  // 16-bin k-blocks of the mel tiles one wave may be given
#define IAS_SM_DESC_W (4 + 2 * IAS_SM_MAX_BLOCKS)   // descriptor ints per wave: nblk, first A block, first tile, passes (max over the waves of ceil(nblk / IAS_SM_REG_BLOCKS)), then per block (power offset, tile | first << 8 | last << 9 | row quarters (rows 4q .. 4q+3) << 10 | next tile << 16)
#define IAS_SM_WAVES 4        // waves per workgroup; a group of 16 frames = IAS_SM_FPW frames per wave
#define IAS_SM_FPW (16 / IAS_SM_WAVES)

struct IasSmLayout {
  int n_fft, N2, Q, NT, NB, VPL, NPAIR_IT;
  int e_win, e_b1, e_tw1, e_a2, e_tw2, e_unp, n_entries;   // entry offsets; one entry = 64 floats (one per lane)
  int pstr;                                                // floats per frame slot of the power buffer (mel mode)
  int off_desc, off_mela;                                  // float offsets of the mel descriptors / A-operand table
};

#if defined(__HIPCC__)
#define IAS_SM_HD __host__ __device__ constexpr inline
#else
#define IAS_SM_HD constexpr inline
#endif

IAS_SM_HD IasSmLayout ias_sm_layout(int n_fft) {
  IasSmLayout L{};
  L.n_fft = n_fft; L.N2 = n_fft / 2; L.Q = L.N2 / 16; L.NT = L.Q / 16; L.NB = L.Q / 8;
  L.VPL = L.Q / 2;                 // samples per lane and frame
  L.NPAIR_IT = L.N2 / 128;         // = NB: bin pairs (k, N2 - k) per lane, k = the lane's own lower-half bins
  int e = 0;
  L.e_win = e; e += L.VPL;         // window of the lane's samples, in load order
  L.e_b1 = e;  e += 8;             // stage-1 B operand: cos[4], sin[4]
  L.e_tw1 = e; e += L.NT * 4 * 2;  // [t][r] (cos, sin) of W_N2^(n2 k1)
  L.e_a2 = e;  e += 4;             // stage-2a A operand [c][x]
  L.e_tw2 = e; e += (L.NB - 1) * 2 * 2;   // [b-1][kl] (cos, sin) of W_Q^(b ka)
  L.e_unp = e; e += L.NPAIR_IT * 2;       // [i] (cos, -sin) of W_nfft^k
  L.n_entries = e;
  // power buffer stride: >= bins, = 8 mod 64 (the 16-byte B-operand reads of 16 frame slots are then conflict-free)
  const int bins = L.N2 + 1;
  L.pstr = ((bins - 8 + 63) / 64) * 64 + 8;
  L.off_desc = 64 * L.n_entries;
  L.off_mela = L.off_desc + IAS_SM_WAVES * IAS_SM_DESC_W;
  return L;
}

// row i of stage-1 M-tile t  ->  n2
IAS_SM_HD int ias_sm_n2_of(int Q, int t, int row) {
  if (Q == 16) return row;
  const int h = t >> 1, e = t & 1;
  return 2 * (row + 16 * h) + e;
}
