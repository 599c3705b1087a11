/* C ABI of libias_hip.so -- the MI355X (gfx950) kernels of the inverse-audio-synthesis inner loop.
 *
 * The reference (turian/inverse-audio-synthesis) is pure Python and has no FFI layer; its boundary
 * for this path is a set of Python classes.  Each entry point below names the
 * reference interface it stands behind.  All pointers are DEVICE pointers unless said otherwise,
 * tensors are contiguous fp32, `stream` is a hipStream_t (NULL = default stream).  Outputs and
 * workspaces are caller-allocated; nothing is allocated, freed or synchronised inside a call, so
 * every call can be captured into a hipGraph.  Return value: 0 (IAS_OK) or a negative IAS_ERR_*.
 */
#ifndef IAS_HIP_H
#define IAS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define IAS_OK 0
#define IAS_ERR_ARG (-1)
#define IAS_ERR_UNSUPPORTED (-2)
#define IAS_ERR_LAUNCH (-3)
#define IAS_ERR_WORKSPACE (-4)

/* Library version (major*100 + minor). */
int ias_version(void);

/* dst[0..n) = src[0..n) with 16-byte accesses; n % 4 == 0.  Bench calibration of the HBM rate. */
int ias_stream_copy(const float* src, float* dst, long long n, void* stream);

/* dst[0] = the device's constant 100 MHz counter (10 ns ticks) at the point of the stream where the call is placed;
 * capturable into a hipGraph (measurement plumbing of bench.py; no reference counterpart). */
int ias_stamp(unsigned long long* dst, void* stream);

/* ---- Voice render: torchsynth.synth.Voice as called at
 * reference vicreg_audio_params.py:86-94,114; audio_to_params.py:196-203,215,240-257; pretrain.py:75.
 * B voices, T = buffer_size samples, Tc = control buffer size (buffer_size_seconds * 441). */

/* Bytes of workspace ias_voice_render needs (ctrl signals, per-voice constants, tile sums, peaks). */
long long ias_voice_workspace_bytes(int B, int T, int Tc);

/* Control-rate pass only: params01 [B,78] in [0,1] (registration order, voice_spec.py) ->
 * ctrl [B,5,Tc] (mod-matrix outputs) and vconst [B] x 64 bytes (IasVoiceConst).  env [B,8,Tc] is
 * scratch that receives the six envelopes (adsr_1, adsr_2, lfo_1_amp, lfo_2_amp, lfo_1_rate, lfo_2_rate)
 * and the two LFO outputs. */
int ias_voice_control(const float* params01, float* ctrl, void* vconst, float* env, int B, int Tc,
                      int control_rate, void* stream);

/* ias_voice_control into the ctrl / vconst / env regions of an ias_voice_render workspace (workspace_bytes >=
 * ias_voice_workspace_bytes(B, T, Tc)); followed by ias_voice_stage calls on the same workspace it is the render
 * split into separately launchable pieces (pipelined schedules).  The workspace layout is private to the library. */
int ias_voice_control_ws(const float* params01, void* workspace, long long workspace_bytes, int B, int T, int Tc,
                         int control_rate, void* stream);

/* Same as ias_voice_control plus the intermediates dbg [B,10,Tc]: rows 0-5 the envelopes,
 * 6-7 the LFO phases, 8-9 the LFO outputs. */
int ias_voice_control_debug(const float* params01, float* ctrl, void* vconst, float* env, float* dbg, int B,
                            int Tc, int control_rate, void* stream);

/* Voice.output(): params01 [B,78], noise [B,T] (the fixed Noise(seed=13) buffer) -> audio [B,T].
 * normalize != 0 applies torchsynth's normalize_if_clipping (row / max(|row|) when the max > 1).
 * math_mode 0: the tested arithmetic contract (oracle math "cr": every fp32 operation of the phase path correctly
 * rounded); 1: the pitch exp2 is the hardware's fp32 v_exp_f32 (<= 1 ulp, the reference's own precision) -- kept
 * for the A/B measurement in DESIGN.md, not covered by the bit-exactness tests. */
int ias_voice_render(const float* params01, const float* noise, float* audio, void* workspace,
                     long long workspace_bytes, int B, int T, int Tc, int sample_rate, int control_rate,
                     int normalize, int math_mode, void* stream);

/* One stage of ias_voice_render on a workspace ias_voice_control_ws has filled (ias_voice_render = control +
 * stage 0 [+ stage 1]): 0 the single-pass audio-rate kernel (phase increments, chained fp64 scan across tiles,
 * oscillators, mixer -> unnormalised audio and row peaks), 1 normalize_if_clipping in place, 2 only the re-zeroing of
 * the kernel's polled words that stage 0 starts with (pipelines issue it ahead, on another stream, once the render and
 * the readers of its row peaks are done with the workspace; noise / audio may be NULL), 3 stage 0 without the re-zeroing. */
int ias_voice_stage(int stage, int math_mode, const float* noise, float* audio, void* workspace,
                    long long workspace_bytes, int B, int T, int Tc, int sample_rate, void* stream);

/* status[0] (device) = 0 if the last render's tile chain completed, 1 if a workgroup's bounded wait for
 * its predecessors expired (that tile's audio is then NaN: an expired wait never continues with partial sums). */
int ias_voice_read_status(const void* workspace, unsigned* status, int B, int T, int Tc, void* stream);

/* The same flag, STICKY: status[0] (device) != 0 if ANY render into this workspace lost a tile since the word was last
 * cleared (clear != 0 re-zeroes it behind the read, in stream order).  ias_voice_read_status only sees the last render --
 * every render re-arms its own word --; a training loop that looks every N steps reads this one (the reference runs its
 * Trainer with detect_anomaly=True, pretrain.py:96: a NaN loss must come with its reason).  Lives in the workspace in
 * front of the words a render re-zeroes; the caller zeroes the workspace ONCE after allocating it. */
int ias_voice_read_status_sticky(void* workspace, unsigned* status, int B, int T, int Tc, int clear, void* stream);

/* Byte offset inside the workspace of the B row peaks (fp32, max |x| of the un-normalised mix) of the last render.
 * ias_pqmf_analysis / ias_stft take that address as `rowpeak` to fold normalize_if_clipping into their own pass
 * (render with normalize = 0): the normalised audio is then never written or re-read. */
/* Byte offsets of the control signals [B,5,Tc] fp32 / the per-voice constants [B] x 64 B that the last render (or
 * ias_voice_control_ws) left in the workspace: ias_voice_backward's ctrl / vconst without a second control pass. */
long long ias_voice_ctrl_offset(int B, int T, int Tc);
long long ias_voice_vconst_offset(int B, int T, int Tc);
long long ias_voice_peaks_offset(int B, int T, int Tc);

/* ctrl [B,5,Tc], vconst [B,16] and (peaks_out non-NULL) the row peaks [B] of the last render, copied out of its workspace in
 * one launch: what Voice's autograd node keeps for the backward (torchsynth has no counterpart: the reference never
 * differentiates through its synth, /root/reference/audio_to_params.py:56-172). */
int ias_voice_save_for_backward(const void* workspace, float* ctrl_out, float* vconst_out, float* peaks_out, int B, int T,
                                int Tc, void* stream);

/* Copy the B row peaks (max |x| before normalisation) of the last render out of the workspace. */
int ias_voice_read_peaks(const void* workspace, float* peaks, int B, int T, int Tc, void* stream);

/* ---- PQMF: reference pqmf.py:46-55 (PQMF.forward/analysis/synthesis), K = taps + 1 (odd). */

/* Output frames of analysis: floor((T + 2*(K-1)/2 - K) / N) + 1, or a negative error. */
int ias_pqmf_out_len(int T, int N, int K);

/* ---- backward of the audio-rate render (the audio -> params -> synth -> loss loop the
 * reference left commented out, audio_to_params.py:56-172; torchsynth's modules are differentiable torch code) ----
 * ctrl [B,5,Tc] and vconst [B] (64 B each) are ias_voice_control's outputs for the same parameters; noise [B,T];
 * g_mixed [B,T] = d loss / d (un-normalised mix).  Scratch: planes [B, ias_voice_grad_nplanes(), T] fp32,
 * tile_sums [B, ias_voice_grad_tiles(T), 2] fp64.  Outputs: g_ctrl [B,5,Tc] fp32 = d loss / d ctrl;
 * partials [B, ias_voice_grad_tiles(T), ias_voice_grad_nscalars()] fp64, whose sum over tiles is d loss / d of the
 * per-voice constants f0_1 depth_1 phi_1 f0_2 depth_2 phi_2 kpart shape gain lvl0 lvl1 lvl2. */
int ias_voice_grad_tiles(int T);
int ias_voice_grad_nscalars(void);
int ias_voice_grad_nplanes(void);
int ias_voice_backward(const float* ctrl, const void* vconst, const float* noise, const float* g_mixed,
                       float* planes, double* tile_sums, double* partials, float* g_ctrl, int B, int T, int Tc,
                       int sample_rate, void* stream);
/* The same behind torchsynth's normalize_if_clipping (audio = mix / peak on rows with peak = max |mix| > 1): g_audio is
 * the cotangent of the NORMALISED audio [B,T]; ias_voice_norm_backward turns it, the audio and the row peaks
 * (ias_voice_read_peaks) into rownorm [B][4] = {divisor, index t* of the peak sample as int bits (-1: none), correction
 * -sign(audio[t*]) sum_t g[t] audio[t] / peak, 0} with two launches (scratch: B * ias_voice_norm_scratch_len(T)
 * doubles); ias_voice_backward_norm applies g / divisor (+ correction at t*) as it reads g_audio.  rownorm NULL =
 * ias_voice_backward.  Replaces the 14 elementwise / reduce / gather / scatter launches of the torch expression. */
long long ias_voice_norm_scratch_len(int T);
int ias_voice_norm_backward(const float* g_audio, const float* audio, const float* peaks, double* scratch,
                            float* rownorm, int B, int T, void* stream);
int ias_voice_backward_norm(const float* ctrl, const void* vconst, const float* noise, const float* g_audio,
                            const float* rownorm, float* planes, double* tile_sums, double* partials, float* g_ctrl,
                            int B, int T, int Tc, int sample_rate, void* stream);
/* ias_voice_backward_norm that also leaves g_scal [B, ias_voice_grad_nscalars()] fp64 = partials summed over the tiles (in
 * tile order) -- the gradient of the per-voice constants -- instead of leaving that sum to the caller. */
int ias_voice_backward_sums(const float* ctrl, const void* vconst, const float* noise, const float* g_mixed,
                            const float* rownorm, float* planes, double* tile_sums, double* partials, float* g_ctrl,
                            double* g_scal, int B, int T, int Tc, int sample_rate, void* stream);
/* In two stages on the same buffers: stage 0 = the phase increments and their tile sums from the control signals (no
 * cotangent: g_mixed, rownorm, noise, partials, g_ctrl, g_scal may be NULL), stage 1 = everything else. */
int ias_voice_backward_sums_stage(int stage, const float* ctrl, const void* vconst, const float* noise, const float* g_mixed,
                                  const float* rownorm, float* planes, double* tile_sums, double* partials, float* g_ctrl,
                                  double* g_scal, int B, int T, int Tc, int sample_rate, void* stream);

/* Control-rate half of the same backward: params01 [B,78], g_ctrl [B,5,Tc] fp32 and g_scal [B,12] fp64 (g_ctrl of
 * ias_voice_backward and the sum over tiles of its partials) -> g_params01 [B,78] fp32.  One launch instead of the
 * several hundred small torch kernels of voice_grad.control_graph + autograd (which remains the definition and the
 * test reference).  IAS_ERR_UNSUPPORTED when Tc does not fit LDS (> ~3000 points) or control_rate != 441. */
int ias_voice_control_backward(const float* params01, const float* g_ctrl, const double* g_scal, float* g_params01,
                               int B, int Tc, int control_rate, void* stream);
/* The same in three launches (round 3): the six-envelope phase -- 60 % of the fp64 pow / log work -- on 6 x B workgroups
 * instead of B.  workspace: ias_voice_control_backward_ws_bytes(B, Tc) bytes of device memory, 16-byte aligned. */
long long ias_voice_control_backward_ws_bytes(int B, int Tc);
int ias_voice_control_backward_ws(const float* params01, const float* g_ctrl, const double* g_scal, float* g_params01,
                                  void* workspace, long long workspace_bytes, int B, int Tc, int control_rate,
                                  void* stream);
/* The same in two stages on the same workspace: stage 0 = the part that does not see the cotangent (the envelope values,
 * parameters only; g_ctrl, g_scal, g_params01 may be NULL) -- a caller that knows at render time that a backward will
 * follow can run it beside the loss computation on another stream; stage 1 = the rest. */
int ias_voice_control_backward_ws_stage(int stage, const float* params01, const float* g_ctrl, const double* g_scal,
                                        float* g_params01, void* workspace, long long workspace_bytes, int B, int Tc,
                                        int control_rate, void* stream);

/* Transposed, zero-padded tap table: ias_pqmf_packed_taps_len(N, K) floats -- the fast kernel's layout for N = 3, 4
 * with K = 63, the wide kernel's [K][8|16|32|64] layout for other N <= 64 with K <= 255, 0 otherwise (generic
 * kernel only); ias_pqmf_pack_taps fills packed (device, 8-byte aligned) from H [N,K] (device); re-run when H
 * changes. */
int ias_pqmf_packed_taps_len(int N, int K);
int ias_pqmf_pack_taps(const float* H, float* packed, int N, int K, void* stream);

/* HOST helper: N = 3, K = 63 filters that are the cosine modulation pqmf.py:21-30 builds (H[k][j] = g[j] c_k[j]) ->
 * the 52 signed prototype taps of the modulated-form kernel, out_host [ias_pqmf_modtab_len() = 64] (copy to the
 * device once); IAS_ERR_UNSUPPORTED for any other H (then pass modtab = NULL). */
int ias_pqmf_modtab_len(void);
int ias_pqmf_build_modtab(const float* H_host, int N, int K, float* out_host);

/* analysis: x [B,T] (= [B,1,T]), H [N,K] (= buffer H[N,1,K]) -> z [B,N,L]   (pqmf.py:49-50).
 * modtab != NULL (N = 3, K = 63, the device copy of ias_pqmf_build_modtab(H)): the filterbank is evaluated in its
 * cosine-modulated form (five alternating-sign prototype sums + a 3-point modulation per frame: ~60 instead of 189
 * multiply-adds; agrees with the tap-ordered chain to ~3e-8 of the output scale, not bit for bit), any alignment.
 * modtab == NULL:
 * K = 63 with N = 3 or 64, x 16-byte aligned and T % 4 == 0: the filterbank runs on the fp32 matrix cores
 * (v_mfma_f32_16x16x4_f32, exact fp32; N = 3 with z 16-byte aligned and L % 4 == 0: the wave-pipelined kernel) and
 * `packed` is not read.  Otherwise: packed = the ias_pqmf_pack_taps table of H (fast / wide VALU kernels), or NULL
 * (generic one-lane-per-output kernel, 10-100x slower).  Every modtab == NULL path evaluates the same tap-ordered fmaf
 * chain per output: results are bit-identical across them.
 * mean/stdv [N] (both or neither, may be NULL): fused (z - mean[k]) / stdv[k] of
 * AudioEmbedding._preprocess (reference audioembed.py:41,49).
 * rowpeak [B] (may be NULL): row peaks of x; the result is the analysis of x[b] / rowpeak[b] where rowpeak[b] > 1
 * (torchsynth normalize_if_clipping folded in, see ias_voice_peaks_offset). */
int ias_pqmf_analysis(const float* x, const float* H, const float* packed, const float* modtab, float* z, const float* mean,
                      const float* stdv, const float* rowpeak, int B, int T, int N, int K, void* stream);

/* synthesis: z [B,N,L], G [N,K] (= buffer G[1,N,K]) -> out [B, L*N] (= [B,1,L*N])   (pqmf.py:52-55). */
/* packed: ias_pqmf_pack_synth_taps table of G (ias_pqmf_synth_taps_len floats; wide kernel for N <= 64, K <= 255),
 * or NULL (generic kernel, several times slower). */
int ias_pqmf_synth_taps_len(int N, int K);
int ias_pqmf_pack_synth_taps(const float* G, float* packed, int N, int K, void* stream);
int ias_pqmf_synthesis(const float* z, const float* G, const float* packed, float* out, int B, int L, int N, int K,
                       void* stream);
/* The same into out [B, T_out], T_out <= L * N: the first T_out samples of every row, contiguous. */
int ias_pqmf_synthesis_t(const float* z, const float* G, const float* packed, float* out, int B, int L, int N, int K,
                         int T_out, void* stream);

/* ---- STFT / mel spectral losses.  No live reference code: spec = the commented mel block at
 * reference conf/config.yaml:51-61 and its use at audio_to_params.py:150-153 (torchaudio
 * MelSpectrogram semantics), plus the auraloss TODO at audio_to_params.py:233. */

/* Frames of a center=True STFT: 1 + T / hop (needs T > n_fft/2 for reflect padding). */
int ias_stft_num_frames(int T, int n_fft, int hop);

/* Number of [3]-double partial records ias_stft writes when loss_mode != 0.  It depends on the kernel the call will
 * run, hence on its arguments: flags bit 0: an ias_stft_build_mtables block is given, bit 1: mel filters are given,
 * bit 2: an ias_stft_build_segtab block is given. */
long long ias_stft_partials_count(int B, int T, int n_fft, int hop, int flags);

/* HOST helpers: length (floats) and contents of the lane-major window/twiddle table block the kernel
 * keeps in registers.  window_host [n_fft] = the analysis window zero-padded and centred to n_fft
 * (host memory), out_host [ias_stft_tables_len(n_fft)] (host memory; copy it to the device once). */
int ias_stft_tables_len(int n_fft);
int ias_stft_build_tables(int n_fft, const float* window_host, float* out_host);

/* HOST helpers of the matrix-core STFT kernel (v_mfma_f32_16x16x4_f32; csrc/stft_mfma_kernels.hip): length (floats)
 * and contents of its constant block: per-lane window, DFT-matrix operands and twiddles, and -- when the host copies
 * of the packed mel filterbank (mel_start / mel_count / mel_woff [n_out], mel_w) are given -- the filterbank re-cut
 * into banded 16-output tiles in A-operand order with a per-wave tile list.  NULL mel_* = linear bins. */
long long ias_stft_mtables_len(int n_fft, const int* mel_start_host, const int* mel_count_host, int n_out);
int ias_stft_build_mtables(int n_fft, const float* window_host, const int* mel_start_host, const int* mel_count_host,
                           const int* mel_woff_host, const float* mel_w_host, int n_out, float* out_host);

/* HOST helpers of the n_fft 1024 kernel's mel projection (csrc/stft2_kernels.hip): the packed triangular filterbank
 * (host copies of mel_start / mel_count / mel_woff / mel_w) re-cut into contiguous bin segments between filter centres,
 * with the store offset of every bin in a segment-major buffer and the two weights (rising into filter j, falling out of
 * filter j-1) per bin.  IAS_ERR_UNSUPPORTED: not a triangular filterbank, more than 192 filters, or segments too long
 * for the kernel's buffer -- ias_stft then takes the packed filters (segtab = NULL). */
long long ias_stft_segtab_len(int n_fft, const int* mel_start_host, const int* mel_count_host, const int* mel_woff_host,
                              const float* mel_w_host, int n_out);
int ias_stft_build_segtab(int n_fft, const int* mel_start_host, const int* mel_count_host, const int* mel_woff_host,
                          const float* mel_w_host, int n_out, float* out_host);

/* Framed STFT of audio [B,T] (center=True, reflect padding, one-sided); tables = device copy of the
 * ias_stft_build_tables block, mtables = device copy of the ias_stft_build_mtables block (built with the same
 * filterbank as mel_*; NULL or IAS_STFT_MFMA unset: the radix-8 kernels), segtab = device copy of the
 * ias_stft_build_segtab block of the same filterbank or NULL.  ticket [2] ints or NULL: the
 * matrix-core kernel's work counter, zero before the first launch (the kernel re-arms it on exit); launches that may
 * run concurrently need counters of their own; NULL = static round-robin assignment of the frame groups.  Per-bin value by value_mode: 1 |X|, 2 |X|^2, 3 sqrt(max(|X|^2, eps)).
 * Optional mel projection as packed triangular filters (mel_start/mel_count/mel_woff [n_out],
 * mel_w [mel_nnz]); with NULL mel_* n_out must be n_fft/2+1.
 * out [B,F,n_out] (frames-major) or NULL; target [B,F,n_out] + partials required when
 * loss_mode is 1 (sum |v-t|) or 2 (MR-STFT sums {(t-v)^2, t^2, |log v - log t|}).
 * rowpeak [B] or NULL: row peaks of audio; the spectrum is that of audio[b] / rowpeak[b] where rowpeak[b] > 1
 * (normalize_if_clipping folded in: |X|^2 scales by 1 / peak^2).
 * n_fft in {512, 1024, 2048}. */
int ias_stft(const float* audio, const float* tables, const float* mtables, const float* segtab, const int* mel_start,
             const int* mel_count, const int* mel_woff, const float* mel_w, int mel_nnz, float* out, const float* target,
             double* partials, const float* rowpeak, int* ticket, int B, int T, int n_fft, int hop, int n_out,
             int value_mode, int loss_mode, float eps, void* stream);

/* Backward of the spectral losses w.r.t. the audio (the reference's mel-L1 loop
 * audio_to_params.py:150-153 is commented out and would have used torchaudio's differentiable modules, its MR-STFT
 * is the auraloss TODO at audio_to_params.py:233):
 *   loss_mode 1: g_audio [B,T] = g_loss[0] * d (scale * sum |V(audio) - target|) / d audio, V = (mel of) |STFT|^power;
 *   loss_mode 2: one resolution of the MR-STFT loss (linear bins, power 1, V = sqrt(max(|X|^2, eps))), cotangent
 *                coef[0] (V - target) + coef[1] sign(V - target) / V with coef a device double[2].
 * window [n_fft] on the device; mel_* as for ias_stft (NULL = linear bins, n_out = n_fft/2+1); target [B,F,n_out]
 * frames-major; power 1 or 2; g_loss a device scalar (NULL = 1); frame_grad [B,F,n_fft] fp32 scratch.
 * tables: the plan's ias_stft_build_tables block (device) or NULL.  With tables the frame part runs on the forward's
 * wave-per-frame FFT core (ias_stft_grad_frames); with NULL the workgroup-per-frame-pair kernel of round 1. */
int ias_stft_loss_backward(const float* audio, const float* window, const float* tables, const int* mel_start,
                           const int* mel_count, const int* mel_woff, const float* mel_w, int mel_nnz,
                           const float* target, const float* g_loss, const double* coef, float* frame_grad,
                           float* g_audio, int B, int T, int n_fft, int hop, int n_out, int power, int loss_mode,
                           float scale, float eps, void* stream);
/* The frame part alone: frame_grad [B,F,n_fft] = window * d loss / d frame (mel_* NULL: linear bins). */
int ias_stft_grad_frames(const float* audio, const float* tables, const int* mel_start, const int* mel_count,
                         const int* mel_woff, const float* mel_w, int mel_nnz, int n_out, const float* target,
                         const double* coef, float* frame_grad, int B, int T, int n_fft, int hop, int power,
                         int loss_mode, float scale, float eps, void* stream);
/* The frame part with the overlap-add inside the kernel (round 3; what ias_stft_loss_backward uses when the shape allows:
 * hop even (n_fft 2048: hop % 4 == 0), hop <= n_fft / 2): a wave walks a chunk of plan_host[0] = G consecutive frames of
 * one row and adds the windowed frame gradients in frame order in an LDS ring; chunk_spans (B * plan[1] * plan[2] floats
 * <= B * F * n_fft, 16-byte aligned) receives B * plan_host[1] spans of plan_host[2] = (G - 1) hop + n_fft floats: span c, entry i = the
 * sum over the frames of chunk c (row c / plan[1], frames [j G, j G + G), j = c % plan[1]) at padded sample j G hop + i.
 * plan_host: int[3] on the HOST, written before return.  IAS_ERR_UNSUPPORTED: shape not served, nothing launched. */
int ias_stft_grad_spans(const float* audio, const float* tables, const int* mel_start, const int* mel_count,
                        const int* mel_woff, const float* mel_w, int mel_nnz, int n_out, const float* target,
                        const double* coef, float* chunk_spans, int B, int T, int n_fft, int hop, int power,
                        int loss_mode, float scale, float eps, int* plan_host, void* stream);
/* The plan ias_stft_grad_spans will use for a shape, without launching (size the span buffer with it: B * plan[1] *
 * plan[2] floats; mel_nnz = 0: linear bins, n_out = n_fft/2+1), and the finish for several resolutions at once:
 * g_audio [B,T] = g_loss[0] (device fp32, NULL = 1) * sum over the resolutions, in their order, of the overlap-added and
 * reflect-folded chunk spans.  spans_host: HOST array of nres <= 8 device pointers; plans_host: HOST ints [nres][5] =
 * {n_fft, hop, plan[0], plan[1], plan[2]}. */
int ias_stft_grad_span_plan(int B, int T, int n_fft, int hop, int mel_nnz, int n_out, int* plan_host);
int ias_stft_grad_combine(const float* const* spans_host, const int* plans_host, int nres, const float* g_loss,
                          float* g_audio, int B, int T, void* stream);

/* MultiResolutionSTFTLoss scalar glue (auraloss defaults; spectral.py), one launch each, fp64 inside:
 * ias_mrstft_total: loss[0] (device fp32) = (sum_k sqrt(s_k[0]) / sqrt(s_k[1]) + s_k[2] / counts_host[k]) / nres with
 *   sums_host a HOST array of nres <= 8 device pointers to the resolutions' ias_reduce_partials sums;
 * ias_mrstft_coef: coef[2] (device doubles) for ias_stft_loss_backward(loss_mode 2) = {g / (nres sqrt(s[0]) sqrt(s[1]))
 *   (0 when the denominator is 0), g / (nres count)}, g = g_loss[0] (device fp32; NULL = 1). */
int ias_mrstft_total(const double* const* sums_host, const double* counts_host, int nres, float* loss, void* stream);
int ias_mrstft_coef(const double* sums, const float* g_loss, double count, int nres, double* coef, void* stream);

/* Plain L1 between two fp32 arrays of n elements (the sub-band L1 of configs[4]): partials [ias_l1_partials_count(n)][3]
 * doubles (column 0 = sum |x - y| of a workgroup; finish with ias_reduce_partials(scale = 1/n) for the mean), and
 * gx = sign(x - y) * g_loss[0] * scale (sign(0) = 0; g_loss device fp32, NULL = 1). */
long long ias_l1_partials_count(long long n);
int ias_l1_partials(const float* x, const float* y, long long n, double* partials, void* stream);
int ias_l1_grad(const float* x, const float* y, const float* g_loss, float scale, long long n, float* gx, void* stream);

/* sums[3] (doubles) = column sums of partials [n][3], fixed order (deterministic); when mean_out is not
 * NULL also mean_out[0] = (float)(sums[0] * scale). */
int ias_reduce_partials(const double* partials, long long n, double* sums, double scale, float* mean_out,
                        void* stream);

/* ---- VICReg loss: reference vicreg.py:35-58 (VICReg.loss) and :73-76 (off_diagonal). */

/* Workspace bytes for ias_vicreg_loss (bf16 transposed centred copies, column stats, partials). */
long long ias_vicreg_workspace_bytes(int B, int D);

/* Byte offset inside that workspace of colstats [4][D] fp32 (mean_x, mean_y, sum (x-mean)^2, same for y),
 * valid after ias_vicreg_loss. */
long long ias_vicreg_colstats_offset(int B, int D);

/* x, y [B,D] fp32 -> out[4] = (loss, repr_loss, std_loss, cov_loss).  cfg_batch: the CONFIGURED batch
 * size whose (cfg_batch - 1) divides the covariance (vicreg.py:47-48 reads it from cfg, not from x). */
int ias_vicreg_loss(const float* x, const float* y, float* out, void* workspace, long long workspace_bytes,
                    int B, int D, int cfg_batch, float sim_coeff, float std_coeff, float cov_coeff, void* stream);

/* Backward of ias_vicreg_loss (the reference gets it from autograd through vicreg.py:35-58): gcoef [4] device floats,
 * the cotangents of (loss, repr_loss, std_loss, cov_loss) -> gx, gy [B,D] fp32.  Same workspace as the forward call,
 * untouched in between (it holds the column statistics and the centred bf16 copies); D % 8 == 0.  Closed form with
 * the B x B Gram (never a D x D matrix); both matrix products on the bf16 matrix cores, fp32 accumulate. */
int ias_vicreg_backward(const float* x, const float* y, const float* gcoef, float* gx, float* gy, void* workspace,
                        long long workspace_bytes, int B, int D, int cfg_batch, float sim_coeff, float std_coeff,
                        float cov_coeff, void* stream);

/* The same pair for x, y (and gx, gy) that are COLUMN BLOCKS of wider row-major matrices: row strides ld / ldg in floats
 * (>= D; with ld != D or ldg != D: multiples of 4 and 16-byte aligned base pointers).  This is how the global-batch loss
 * of the gather the reference keeps commented out (vicreg.py:38-39 with FullGatherLayer :79-95) runs without copies:
 * ONE all-gather of cat(x, y, dim=1) lands in a [W B_l, 2 D] buffer, x = buf[:, :D], y = buf[:, D:] are consumed in
 * place, and the two gradient blocks are written straight into the [W B_l, 2 D] cotangent the backward reduce-scatters. */
int ias_vicreg_loss_ld(const float* x, const float* y, long long ld, float* out, void* workspace, long long workspace_bytes,
                       int B, int D, int cfg_batch, float sim_coeff, float std_coeff, float cov_coeff, void* stream);
int ias_vicreg_backward_ld(const float* x, const float* y, long long ld, const float* gcoef, float* gx, float* gy,
                           long long ldg, void* workspace, long long workspace_bytes, int B, int D, int cfg_batch,
                           float sim_coeff, float std_coeff, float cov_coeff, void* stream);

/* ias_vicreg_backward_ld with the four cotangents as separate device floats, any of them NULL (= zero): autograd hands the
 * outputs that were not differentiated over as None, and packing four scalars into gcoef costs a kernel per step. */
int ias_vicreg_backward4_ld(const float* x, const float* y, long long ld, const float* g_loss, const float* g_repr,
                            const float* g_std, const float* g_cov, float* gx, float* gy, long long ldg, void* workspace,
                            long long workspace_bytes, int B, int D, int cfg_batch, float sim_coeff, float std_coeff,
                            float cov_coeff, void* stream);

/* One stage of ias_vicreg_loss on the same workspace: 0 column pass, 1 the Gram kernel(s) on the matrix cores,
 * 2 the final reduction (stage < 0: all of them = ias_vicreg_loss).  Lets a caller time the Gram alone. */
int ias_vicreg_stage(int stage, const float* x, const float* y, float* out, void* workspace, long long workspace_bytes,
                     int B, int D, int cfg_batch, float sim_coeff, float std_coeff, float cov_coeff, void* stream);

/* Which side the covariance term (vicreg.py:47-51) is contracted on is a pure function of the shape.
 * sum_{i != j} cov_ij^2 over the D x D matrix cov = Xc^T Xc / (n - 1) equals
 * (||Xc Xc^T||_F^2 - sum_j (Xc^T Xc)_jj^2) / (n - 1)^2: the B x B matrix Xc Xc^T costs 2 B^2 D flops instead of 2 B D^2
 * (64x fewer at the reference's B = 128, D = 8192) and is the matrix the backward needs anyway.  Batch side wherever the
 * padded batch (multiple of 128) <= D and D % 8 == 0; the D x D kernels otherwise.  (The diagnostic library can force a
 * side: include/ias_hip_diag.h.) */

/* ---- AudioEmbedding trunk: the depthwise convolutions and the stem of torchvision's mobilenet_v3_small.features
 * (reference vicreg_audio_params.py:52-54, audioembed.py:61), NCHW fp32, padding (K-1)/2, no bias.
 * K in {3, 5}, stride S in {1, 2}.  ias_conv_out_size: output extent of one spatial dimension. */
int ias_conv_out_size(int n, int K, int S);
int ias_dwconv_forward(const float* x, const float* w, float* out, int B, int C, int H, int W, int K, int S, void* stream);
int ias_dwconv_backward_data(const float* g, const float* w, float* gx, int B, int C, int H, int W, int K, int S,
                             void* stream);
long long ias_dwconv_weight_scratch(int B, int C, int K);     /* floats: upper bound for any plane size */
long long ias_dwconv_weight_scratch_hw(int B, int C, int H, int W, int K, int S);   /* floats, for this plane size */
int ias_dwconv_backward_weight(const float* x, const float* g, float* gw, float* scratch, int B, int C, int H, int W,
                               int K, int S, void* stream);
/* ... without its reduction launch: the partial rows stay in `scratch` -> their number (> 0; C K K floats each) for
 * ias_reduce_partials_multi below, or a negative IAS_ERR_* */
int ias_dwconv_backward_weight_partials(const float* x, const float* g, float* scratch, int B, int C, int H, int W, int K,
                                        int S, void* stream);
/* Conv2d(3, 16, 3, stride 2, padding 1, bias=False): x [B,3,H,W], w [16,3,3,3] -> out [B,16,Ho,Wo]; weight gradient. */
/* 1x1 convolutions (nn.Conv2d(Cin, Cout, 1, bias=False)) of torchvision's mobilenet_v3_small.features as run at
 * /root/reference/audioembed.py:61, NCHW fp32, on fp32 MFMA: x [B,Cin,HW], w [Cout,Cin], y / g [B,Cout,HW].
 * ias_pwconv_supported(Cin, Cout) = 1 for the shapes taken (channel counts multiples of 4, 16-padded Cout * Cin and
 * 16-padded Cin * Cout <= 16384, Cin <= 240); the other entry points return IAS_ERR_UNSUPPORTED otherwise.
 * The weight gradient is reduced in a fixed order (deterministic); scratch: ias_pwconv_weight_scratch floats. */
int ias_pwconv_supported(int Cin, int Cout);
int ias_pwconv_forward(const float* x, const float* w, float* y, int B, int Cin, int Cout, int HW, void* stream);
int ias_pwconv_backward_data(const float* g, const float* w, float* gx, int B, int Cin, int Cout, int HW, void* stream);
long long ias_pwconv_weight_scratch(int B, int Cin, int Cout, int HW);
int ias_pwconv_backward_weight(const float* g, const float* x, float* gw, float* scratch, int B, int Cin, int Cout, int HW,
                               void* stream);
/* ... without its reduction launch -> the number of partial rows (> 0; Cout Cin floats each), or a negative IAS_ERR_* */
int ias_pwconv_backward_weight_partials(const float* g, const float* x, float* scratch, int B, int Cin, int Cout, int HW,
                                        void* stream);
/* The projection behind a squeeze-excitation block with the block's gate taken on load: y[b] = W (x[b] * scale[b]),
 * scale [B,Cin] -- torchvision InvertedResidual's `SqueezeExcitation -> Conv2dNormActivation(cexp, cout, 1)` inside
 * mobilenet_v3_small.features (run at /root/reference/audioembed.py:61) without the gate's own pass over the expanded map
 * (`scale * input` of torchvision's SqueezeExcitation.forward); same bits as ias_se_scale followed by ias_pwconv_forward.
 * The input gradient of the product x * scale is ias_pwconv_backward_data; the weight gradient is
 * gw = sum_b g[b] (x[b] * scale[b])^T (scale applied once per sample and column where the sums leave the accumulators). */
int ias_pwconv_forward_scaled(const float* x, const float* scale, const float* w, float* y, int B, int Cin, int Cout, int HW,
                              void* stream);
int ias_pwconv_backward_weight_scaled(const float* g, const float* x, const float* scale, float* gw, float* scratch, int B,
                                      int Cin, int Cout, int HW, void* stream);
int ias_pwconv_backward_weight_partials_scaled(const float* g, const float* x, const float* scale, float* scratch, int B,
                                               int Cin, int Cout, int HW, void* stream);

/* The weight-gradient reductions of a whole backward pass of the trunk (torchvision mobilenet_v3_small.features, run at
 * /root/reference/audioembed.py:61; autograd of /root/reference/vicreg_audio_params.py:96-122) in ONE launch:
 * out[i] = sum_r partial[r n + i] (i < n, r < rows; fixed order) for every entry of a HOST array of `count` entries
 * holding device pointers -- the table travels in the kernel arguments (nothing is copied to the device, nothing has to
 * outlive the call; capturable).  The *_backward_weight_partials entry points leave such rows behind. */
typedef struct IasReduceItem {
  const float* partial;
  float* out;
  int n, rows;
} IasReduceItem;
int ias_reduce_partials_multi(const IasReduceItem* items, int count, void* stream);

/* squeeze-and-excitation blocks of the trunk (torchvision SqueezeExcitation; /root/reference/vicreg_audio_params.py:52-54):
 * per-plane reductions and the per-plane scale over a [planes][hw] activation (planes = B C).
 *   ias_se_plane_reduce: out[p] = scale * sum_i a[p][i]          (b == NULL: the average pool with scale = 1/hw)
 *                        out[p] = scale * sum_i a[p][i] b[p][i]   (the gradient of the per-plane scale)
 *   ias_se_scale:        y[p][i] = x[p][i] * s[p] + (add ? add[p] * add_scale : 0) */
int ias_se_plane_reduce(const float* a, const float* b, float* out, long long planes, int hw, float scale, void* stream);
int ias_se_scale(const float* x, const float* s, const float* add, float* y, long long planes, int hw, float add_scale,
                 void* stream);
/* the block's two 1x1 convolutions on the pooled [B,C]: h = relu(pooled w1^T + b1) [B,Cs], z = h w2^T + b2 [B,C],
 * s = hardsigmoid(z); w1 [Cs,C], w2 [C,Cs]; b1 / b2 may be NULL.  Backward from gs = dL/ds: gp = dL/dpooled and the four
 * parameter gradients (gb1 / gb2 may be NULL); gz [B,C], gh [B,Cs] are scratch.  Sums in a fixed order. */
int ias_se_mlp_forward(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* h,
                       float* z, float* s, int B, int C, int Cs, void* stream);
int ias_se_mlp_backward(const float* gs, const float* z, const float* h, const float* pooled, const float* w1,
                        const float* w2, float* gz, float* gh, float* gp, float* gw1, float* gb1, float* gw2, float* gb2,
                        int B, int C, int Cs, void* stream);

/* head Conv2d(C, Cout, kernel_size=2) of AudioEmbedding (/root/reference/audioembed.py:15-33, 62-68) as one GEMM on
 * channels-last maps: patches [B (H-1) (W-1)][4 C] (columns ordered (c, di, dj), i.e. weight.view(Cout, 4 C) is the
 * GEMM's other operand) from x [B,H,W,C], and the adjoint gp -> gx [B,H,W,C].  patches 16-byte aligned. */
int ias_conv2x2_patches(const float* x, float* patches, int B, int H, int W, int C, void* stream);
int ias_conv2x2_patches_backward(const float* gp, float* gx, int B, int H, int W, int C, void* stream);
/* the same patches from / gradient into an NCHW map x, gx [B,C,H,W] (the trunk's output in front of the first head layer:
 * no permuted copy); IAS_ERR_UNSUPPORTED when H W > 255 */
int ias_conv2x2_patches_nchw(const float* x, float* patches, int B, int H, int W, int C, void* stream);
int ias_conv2x2_patches_backward_nchw(const float* gp, float* gx, int B, int H, int W, int C, void* stream);
/* column sums of a row-major fp32 [rows, cols] matrix in a fixed order (the bias gradient g.sum(0) of the head
 * convolutions run as GEMMs); scratch: ias_colsum_scratch_floats(rows, cols) floats */
long long ias_colsum_scratch_floats(int rows, int cols);
int ias_colsum(const float* a, float* out, float* scratch, int rows, int cols, void* stream);
/* ... without its second launch: the slice sums stay in `scratch` -> their number (> 0, <= 16; `cols` floats each) for
 * ias_reduce_partials_multi (same bits as ias_colsum), or a negative IAS_ERR_*.  Reference: the bias gradient of the
 * head's nn.Conv2d(dim, dim, 2) layers, autograd of /root/reference/audioembed.py:62-68. */
int ias_colsum_partials(const float* a, float* scratch, int rows, int cols, void* stream);
int ias_stem_forward(const float* x, const float* w, float* out, int B, int H, int W, void* stream);
long long ias_stem_weight_scratch(int B);                     /* floats */
int ias_stem_backward_weight(const float* x, const float* g, float* gw, float* scratch, int B, int H, int W, void* stream);
/* ... without its reduction launch -> the number of partial rows (> 0; 432 floats each), or a negative IAS_ERR_* */
int ias_stem_backward_weight_partials(const float* x, const float* g, float* scratch, int B, int H, int W, void* stream);

/* ---- Training-mode BatchNorm2d with the following activation fused (act 0 none, 1 ReLU, 2 Hardswish): replaces the
 * nn.BatchNorm2d + activation pairs of torchvision's mobilenet_v3_small.features (audioembed.py:61) and their autograd.
 * x, y, dy, dx [B,C,HW] fp32 contiguous (NCHW); weight / bias / running_mean / running_var [C] or NULL; save_mean,
 * save_invstd [C]; scratch: ias_bn_scratch_doubles(B, C) doubles; sums: [C][2] floats.  Running statistics are updated
 * as torch does (momentum, unbiased variance). */
long long ias_bn_scratch_doubles(int B, int C);
int ias_bn_act_forward(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                       float* y, float* save_mean, float* save_invstd, double* scratch, int B, int C, int HW, float eps,
                       float momentum, int act, void* stream);
/* the same with the block's residual connection taken in the same pass: y = act(BatchNorm_train(x)) + res (torchvision's
 * InvertedResidual: `result += input` behind the block's last ConvNormActivation); res [B,C,HW]; the same bits as the
 * separate addition */
int ias_bn_act_forward_res(const float* x, const float* res, const float* weight, const float* bias, float* running_mean,
                           float* running_var, float* y, float* save_mean, float* save_invstd, double* scratch, int B,
                           int C, int HW, float eps, float momentum, int act, void* stream);
/* ias_bn_act_forward that also leaves pooled[b][c] = mean over the plane of y[b][c] behind ([B,C]): the average pool of the
 * squeeze-excitation block that follows a depthwise convolution's normalisation in torchvision's InvertedResidual
 * (SqueezeExcitation._scale: `self.avgpool(input)`; the trunk run at /root/reference/audioembed.py:61) -- in the
 * normalisation's own launch on maps up to 15 x 16, as ias_se_plane_reduce behind it on larger ones */
int ias_bn_act_forward_pool(const float* x, const float* weight, const float* bias, float* running_mean, float* running_var,
                            float* y, float* pooled, float* save_mean, float* save_invstd, double* scratch, int B, int C,
                            int HW, float eps, float momentum, int act, void* stream);
int ias_bn_act_backward(const float* x, const float* dy, const float* weight, const float* bias, const float* save_mean,
                        const float* save_invstd, float* dx, float* gw, float* gb, double* scratch, float* sums, int B,
                        int C, int HW, int act, void* stream);

/* ---- Training-mode BatchNorm1d (+ ReLU) on row groups: the Linear -> BatchNorm1d -> ReLU layers of the shared projector
 * (/root/reference/vicreg.py:27-30, 60-70: one projector applied to the audio branch, then to the parameter branch) with
 * both branches stacked.  z, y, dy, dx [G n, F] fp32 row-major; each group of n rows gets its own batch statistics, the
 * running statistics are updated group after group (G calls of nn.BatchNorm1d in that order), *num_batches_tracked += G.
 * lin_bias (or NULL): the bias of the Linear in front, added on the fly (z is the GEMM output without it); g_lin_bias: its
 * gradient.  save_mean / save_invstd [G, F]; weight / bias / running_* [F] or NULL; n >= 2. */
int ias_bn1d_groups_forward(const float* z, const float* lin_bias, const float* weight, const float* bias,
                            float* running_mean, float* running_var, long long* num_batches_tracked, float* y,
                            float* save_mean, float* save_invstd, int G, int n, int F, float eps, float momentum, int relu,
                            void* stream);
int ias_bn1d_groups_backward(const float* z, const float* lin_bias, const float* dy, const float* weight, const float* bias,
                             const float* save_mean, const float* save_invstd, float* dx, float* gw, float* gb,
                             float* g_lin_bias, int G, int n, int F, int relu, void* stream);

/* ---- LARS optimizer step (momentum 0) as three multi-tensor launches: replaces flash.core.optimizers.LARS.step as
 * configured at vicreg_audio_params.py:134-151.
 * tensors [n][3] int64 (device): parameter pointer, gradient pointer, element count (fp32, contiguous);
 * chunks [nchunks][2] int32 (device): tensor index, chunk index within the tensor, chunks of ias_lars_chunk_elems()
 * elements; first_chunk [n+1] int32 (device): prefix of the chunk counts; partials [nchunks][2] doubles and coef [n][2]
 * floats: scratch; hyper [4] floats (device): lr, weight_decay, trust_coefficient, eps.  skip_norms != 0: coef is taken
 * as given (weight_decay == 0: the caller sets (1, 0) per tensor). */
int ias_lars_chunk_elems(void);
int ias_lars_step(const long long* tensors, const int* chunks, const int* first_chunk, double* partials, float* coef,
                  const float* hyper, int ntensors, int nchunks, int skip_norms, void* stream);
/* The same step (weight decay on) with the parameters' norms carried from update to update: carry = 1 computes both norms
 * as ias_lars_step does and lets the update pass leave sum p_new^2 per chunk in partials[2 c] (summed in the norm pass'
 * own order); carry = 2 reads only the gradient in the norm pass and takes the parameters' sums the previous call left in
 * `partials` -- valid while the same tensors are stepped, the buffer is kept and nothing else wrote the parameters.  Same
 * bits as ias_lars_step; one read of the parameters less per step. */
int ias_lars_step_carry(const long long* tensors, const int* chunks, const int* first_chunk, double* partials, float* coef,
                        const float* hyper, int ntensors, int nchunks, int carry, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IAS_HIP_H */
