"""STFT / mel spectral losses on the HIP path (csrc/spectral_kernels.hip).

The reference has no live spectral-loss code.  This module implements the spec it left behind:
  * ``MelSpectrogram`` with the argument names/defaults of the commented config block
    /root/reference/conf/config.yaml:51-61 (torchaudio.transforms.MelSpectrogram semantics),
  * ``MelSpectrogramL1``: ``mean(|mel(audio) - mel(pred_audio)|)`` of the commented training
    code /root/reference/audio_to_params.py:150-153,
  * ``STFTL1`` (BASELINE config #1 "STFT L1 loss") and ``MultiResolutionSTFTLoss`` (the auraloss
    TODO at audio_to_params.py:233; auraloss defaults).
Filterbank / window / twiddle tables are built once on the host; every per-sample operation runs
in the HIP kernel.  The L1 losses (``MelSpectrogramL1``, ``STFTL1``) are differentiable with respect to the audio
(``csrc/spectral_grad_kernels.hip``): with ``Voice.render`` that closes the audio -> params -> synth -> mel-L1 loop
the reference left commented out (audio_to_params.py:56-172); ``MultiResolutionSTFTLoss`` is differentiable with
respect to its first argument (the prediction) through the same kernels.
"""
import ctypes
import os
import math

import torch
import torch.nn as nn

from . import _lib

VALUE_MAG, VALUE_POWER, VALUE_MAG_CLAMPED = 1, 2, 3
LOSS_NONE, LOSS_L1, LOSS_MRSTFT = 0, 1, 2


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm="slaney", mel_scale="htk"):
    """Triangular mel filterbank [n_freqs, n_mels] (torchaudio.functional.melscale_fbanks, htk scale)."""
    assert mel_scale == "htk", "only the htk mel scale of conf/config.yaml is implemented"
    to_mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    mel_pts = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
    hz_pts = 700.0 * (10.0 ** (mel_pts / 2595.0) - 1.0)
    width = hz_pts[1:] - hz_pts[:-1]
    dist = hz_pts.unsqueeze(0) - freqs.unsqueeze(1)
    falling = (-1.0 * dist[:, :-2]) / width[:-1]
    rising = dist[:, 2:] / width[1:]
    fb = torch.max(torch.zeros(1), torch.min(falling, rising))
    if norm == "slaney":
        fb = fb * (2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])).unsqueeze(0)
    else:
        assert norm is None
    return fb


class STFTPlan(nn.Module):
    """Device tables for one (n_fft, win_length, hop, optional mel filterbank) configuration."""

    def __init__(self, n_fft=1024, win_length=None, hop_length=None, n_mels=None, sample_rate=44100,
                 f_min=0.0, f_max=None, norm="slaney", mel_scale="htk"):
        super().__init__()
        assert n_fft in (512, 1024, 2048), "HIP STFT kernel supports n_fft 512/1024/2048"
        self.n_fft = n_fft
        self.win_length = n_fft if win_length is None else win_length
        self.hop_length = self.win_length // 2 if hop_length is None else hop_length
        assert self.win_length <= n_fft and self.hop_length > 0
        win = torch.hann_window(self.win_length)
        left = (n_fft - self.win_length) // 2
        window = torch.zeros(n_fft)
        window[left:left + self.win_length] = win
        self.register_buffer("window", window, persistent=False)
        # lane-major window/twiddle tables of the kernel, built on the host by the C library
        lib = _lib.load()
        n_tab = lib.ias_stft_tables_len(n_fft)
        _lib.check(min(n_tab, 0), "ias_stft_tables_len")
        tables = torch.empty(n_tab, dtype=torch.float32)
        wc = window.contiguous()
        _lib.check(lib.ias_stft_build_tables(n_fft, ctypes.c_void_p(wc.data_ptr()), ctypes.c_void_p(tables.data_ptr())),
                   "ias_stft_build_tables")
        self.register_buffer("tables", tables, persistent=False)
        self.n_mels = n_mels
        if n_mels is not None:
            f_max = float(sample_rate // 2) if f_max is None else f_max
            fb = melscale_fbanks(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate, norm, mel_scale)
            self.register_buffer("fb", fb, persistent=False)
            start, count, woff, w = [], [], [], []
            for m in range(n_mels):
                nz = torch.nonzero(fb[:, m]).flatten()
                if nz.numel() == 0:
                    start.append(0); count.append(0); woff.append(len(w))
                    continue
                s, e = int(nz[0]), int(nz[-1]) + 1
                start.append(s); count.append(e - s); woff.append(len(w))
                w.extend(fb[s:e, m].tolist())
            self.register_buffer("mel_start", torch.tensor(start, dtype=torch.int32), persistent=False)
            self.register_buffer("mel_count", torch.tensor(count, dtype=torch.int32), persistent=False)
            self.register_buffer("mel_woff", torch.tensor(woff, dtype=torch.int32), persistent=False)
            self.register_buffer("mel_w", torch.tensor(w if w else [0.0], dtype=torch.float32), persistent=False)
            self.n_out = n_mels
        else:
            self.n_out = n_fft // 2 + 1
        # constant block of the matrix-core kernel (per-lane DFT operands / twiddles, mel filterbank as banded tiles)
        if n_mels is not None:
            mel_host = [self.mel_start, self.mel_count, self.mel_woff, self.mel_w]
            mel_args = [ctypes.c_void_p(t.data_ptr()) for t in mel_host] + [n_mels]
        else:
            mel_args = [None, None, None, None, 0]
        n_mtab = lib.ias_stft_mtables_len(n_fft, mel_args[0], mel_args[1], mel_args[4])
        if n_mtab == -2:
            # a filterbank whose bands do not fit the matrix-core kernel's tiles: the VALU kernel serves the plan
            self.mtables = None
        else:
            _lib.check(min(n_mtab, 0), "ias_stft_mtables_len")
            mtables = torch.empty(n_mtab, dtype=torch.float32)
            _lib.check(lib.ias_stft_build_mtables(n_fft, ctypes.c_void_p(wc.data_ptr()), *mel_args,
                                                  ctypes.c_void_p(mtables.data_ptr())), "ias_stft_build_mtables")
            self.register_buffer("mtables", mtables, persistent=False)
        # segment-major mel tables of the n_fft 1024 kernel (triangular filterbanks with short segments; else None)
        self.register_buffer("segtab", None, persistent=False)
        if n_mels is not None and n_fft == 1024:
            n_seg = lib.ias_stft_segtab_len(n_fft, mel_args[0], mel_args[1], mel_args[2], mel_args[3], n_mels)
            if n_seg > 0:
                segtab = torch.empty(n_seg, dtype=torch.float32)
                _lib.check(lib.ias_stft_build_segtab(n_fft, mel_args[0], mel_args[1], mel_args[2], mel_args[3], n_mels,
                                                     ctypes.c_void_p(segtab.data_ptr())), "ias_stft_build_segtab")
                self.segtab = segtab
        self._tickets = {}

    def _ticket(self, device):
        """The matrix-core kernel's work counter (two ints, zero between launches: the kernel re-arms it), one per
        stream: launches on different streams may overlap."""
        key = (device, torch.cuda.current_stream().cuda_stream)
        t = self._tickets.get(key)
        if t is None:
            t = self._tickets[key] = torch.zeros(2, dtype=torch.int32, device=device)
        return t

    def num_frames(self, T):
        F = _lib.load().ias_stft_num_frames(T, self.n_fft, self.hop_length)
        _lib.check(min(F, 0), "ias_stft_num_frames (need T > n_fft/2 for reflect padding)")
        return F

    def _call(self, audio, out, target, partials, value_mode, loss_mode, eps, rowpeak=None):
        lib = _lib.load()
        B, T = audio.shape
        mel = self.n_mels is not None
        st = lib.ias_stft(_lib.ptr(audio), _lib.ptr(self.tables), _lib.ptr(self.mtables), _lib.ptr(self.segtab),
                          _lib.ptr(self.mel_start) if mel else None, _lib.ptr(self.mel_count) if mel else None,
                          _lib.ptr(self.mel_woff) if mel else None, _lib.ptr(self.mel_w) if mel else None,
                          int(self.mel_w.numel()) if mel else 0, _lib.ptr(out), _lib.ptr(target), _lib.ptr(partials),
                          _lib.ptr(rowpeak), _lib.ptr(self._ticket(audio.device)), B, T, self.n_fft, self.hop_length,
                          self.n_out, value_mode, loss_mode, float(eps), _lib.stream())
        _lib.check(st, "ias_stft")

    @staticmethod
    def _audio2d(audio):
        a = audio.reshape(audio.shape[0], -1) if audio.dim() == 3 else audio
        assert a.dim() == 2
        a = a.contiguous()
        _lib.require_f32(a)
        return a

    def values(self, audio, value_mode=VALUE_POWER, eps=0.0, rowpeak=None):
        """-> [B, frames, n_out] (frames-major) spectrogram values.  ``rowpeak`` [B]: the row peaks of an
        un-normalised render (``Voice.peaks_view``): values of audio / peak where peak > 1, audio left as it is."""
        a = self._audio2d(audio)
        out = torch.empty((a.shape[0], self.num_frames(a.shape[1]), self.n_out), dtype=torch.float32, device=a.device)
        self._call(a, out, None, None, value_mode, LOSS_NONE, eps, rowpeak)
        return out

    def loss_sums(self, audio, target_values, value_mode, loss_mode, eps=0.0, mean_scale=None, rowpeak=None,
                  reduce_stream=None):
        """Fused STFT + comparison with cached target values -> 3 fp64 sums on the device
        (with ``mean_scale``: -> the fp32 scalar sums[0] * mean_scale, computed by the reduction launch).
        ``reduce_stream``: issue the small fixed-order reduction of the per-workgroup partials on that stream (it waits
        for the STFT kernel through an event) instead of behind the STFT on the current one; the result then belongs to
        ``reduce_stream`` (pipelines whose STFT queue is the critical path: bench.py)."""
        a = self._audio2d(audio)
        lib = _lib.load()
        F = self.num_frames(a.shape[1])
        assert target_values.shape == (a.shape[0], F, self.n_out) and target_values.is_contiguous()
        n = lib.ias_stft_partials_count(a.shape[0], a.shape[1], self.n_fft, self.hop_length,
                                        (0 if self.mtables is None else 1) | (0 if self.n_mels is None else 2) |
                                        (0 if self.segtab is None else 4))
        partials = torch.empty((n, 3), dtype=torch.float64, device=a.device)
        self._call(a, None, target_values, partials, value_mode, loss_mode, eps, rowpeak)
        sums = torch.empty(3, dtype=torch.float64, device=a.device)
        mean = torch.empty((), dtype=torch.float32, device=a.device) if mean_scale is not None else None

        def reduce():
            _lib.check(lib.ias_reduce_partials(_lib.ptr(partials), n, _lib.ptr(sums),
                                               float(mean_scale) if mean_scale is not None else 0.0,
                                               _lib.ptr(mean), _lib.stream()), "ias_reduce_partials")
        if reduce_stream is None:
            reduce()
        else:
            done = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(reduce_stream):
                reduce_stream.wait_event(done)
                for t in (partials, sums, mean, target_values):
                    if t is not None:
                        t.record_stream(reduce_stream)
                reduce()
        return sums if mean is None else mean


class _L1LossFn(torch.autograd.Function):
    """mean |V(audio) - target| with the fused HIP forward and the HIP adjoint w.r.t. the audio."""

    @staticmethod
    def forward(ctx, audio, plan, target_values, value_mode):
        a = plan._audio2d(audio)
        ctx.plan, ctx.value_mode, ctx.shape = plan, value_mode, audio.shape
        ctx.save_for_backward(a, target_values)
        return plan.loss_sums(a, target_values, value_mode, LOSS_L1, mean_scale=1.0 / target_values.numel())

    @staticmethod
    def backward(ctx, g_loss):
        a, target = ctx.saved_tensors
        plan = ctx.plan
        lib = _lib.load()
        B, T = a.shape
        F = plan.num_frames(T)
        mel = plan.n_mels is not None
        frame_grad = torch.empty((B, F, plan.n_fft), dtype=torch.float32, device=a.device)
        g_audio = torch.empty_like(a)
        gl = g_loss.to(torch.float32).reshape(1).contiguous()
        st = lib.ias_stft_loss_backward(
            _lib.ptr(a), _lib.ptr(plan.window), _lib.ptr(plan.tables), _lib.ptr(plan.mel_start) if mel else None,
            _lib.ptr(plan.mel_count) if mel else None, _lib.ptr(plan.mel_woff) if mel else None,
            _lib.ptr(plan.mel_w) if mel else None, int(plan.mel_w.numel()) if mel else 0, _lib.ptr(target), _lib.ptr(gl),
            None, _lib.ptr(frame_grad),
            _lib.ptr(g_audio), B, T, plan.n_fft, plan.hop_length, plan.n_out,
            2 if ctx.value_mode == VALUE_POWER else 1, LOSS_L1, 1.0 / target.numel(), 0.0, _lib.stream())
        _lib.check(st, "ias_stft_loss_backward")
        return g_audio.reshape(ctx.shape), None, None, None


def _l1_loss(plan, audio, target_values, value_mode, rowpeak=None, reduce_stream=None):
    if torch.is_grad_enabled() and audio.requires_grad:
        assert value_mode in (VALUE_POWER, VALUE_MAG)
        assert rowpeak is None, "the folded normalisation is a forward-only path"
        return _L1LossFn.apply(audio, plan, target_values, value_mode)
    return plan.loss_sums(audio, target_values, value_mode, LOSS_L1, mean_scale=1.0 / target_values.numel(),
                          rowpeak=rowpeak, reduce_stream=reduce_stream)


class MelSpectrogram(nn.Module):
    """conf/config.yaml:51-61 block -> [B, n_mels, frames] (torchaudio layout; a transposed view of
    the kernel's frames-major output)."""

    def __init__(self, sample_rate=44100, n_fft=1024, win_length=None, hop_length=512, center=True,
                 pad_mode="reflect", power=2.0, norm="slaney", onesided=True, n_mels=128, mel_scale="htk",
                 f_min=0.0, f_max=None):
        super().__init__()
        assert center and pad_mode == "reflect" and onesided, "kernel implements center/reflect/onesided"
        assert power in (1.0, 2.0)
        self.power = power
        self.plan = STFTPlan(n_fft, win_length, hop_length, n_mels, sample_rate, f_min, f_max, norm, mel_scale)

    @property
    def value_mode(self):
        return VALUE_POWER if self.power == 2.0 else VALUE_MAG

    def frames_major(self, audio):
        return self.plan.values(audio, self.value_mode)

    def forward(self, audio):
        return self.frames_major(audio).transpose(1, 2)


class MelSpectrogramL1(nn.Module):
    """mean(|mel(audio) - mel(target)|)   (audio_to_params.py:150-153)."""

    def __init__(self, **mel_kwargs):
        super().__init__()
        self.mel = MelSpectrogram(**mel_kwargs)

    def target(self, target_audio):
        """Cacheable frames-major mel of the target audio."""
        return self.mel.frames_major(target_audio)

    def forward(self, audio, target_audio=None, target_mel=None, rowpeak=None, reduce_stream=None):
        """``rowpeak``: row peaks of an un-normalised render; the loss is that of the normalised audio
        (torchsynth normalize_if_clipping folded into the STFT pass).  ``reduce_stream``: see ``STFTPlan.loss_sums``
        (forward-only path)."""
        if target_mel is None:
            target_mel = self.target(target_audio)
        return _l1_loss(self.mel.plan, audio, target_mel.detach(), self.mel.value_mode, rowpeak, reduce_stream)


class STFTL1(nn.Module):
    """mean | |STFT(a)|^p - |STFT(b)|^p |  (BASELINE config #1 "STFT L1 loss")."""

    def __init__(self, n_fft=1024, hop_length=512, win_length=None, power=1.0):
        super().__init__()
        assert power in (1.0, 2.0)
        self.value_mode = VALUE_POWER if power == 2.0 else VALUE_MAG
        self.plan = STFTPlan(n_fft, win_length, hop_length)

    def forward(self, audio, target_audio):
        tgt = self.plan.values(target_audio.detach(), self.value_mode)
        return _l1_loss(self.plan, audio, tgt, self.value_mode)


class MultiResolutionSTFTLoss(nn.Module):
    """auraloss.freq.MultiResolutionSTFTLoss defaults: per resolution spectral convergence
    ||Y|-|X||_F/||Y||_F + L1(log|X|, log|Y|), averaged over resolutions (x = prediction, y = target)."""

    def __init__(self, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240),
                 eps=1e-8):
        super().__init__()
        self.eps = eps
        self.parallel = True      # the resolutions on side streams (False: one after the other on the caller's stream)
        self.fused_combine = True # backward: one combine launch for all resolutions (False: one per resolution + adds)
        self.plans = nn.ModuleList([STFTPlan(n, w, h) for n, h, w in zip(fft_sizes, hop_sizes, win_lengths)])

    def target(self, y):
        """Cacheable clamped magnitudes of the target at every resolution (pass as ``targets=`` to ``forward``)."""
        return [plan.values(y.detach(), VALUE_MAG_CLAMPED, self.eps) for plan in self.plans]

    def forward(self, x, y=None, targets=None):
        """x = prediction; the target either as audio ``y`` or as the cached ``targets = self.target(y)``."""
        assert (y is None) != (targets is None), "give the target audio or its cached magnitudes"
        if targets is None:
            targets = self.target(y)
        if torch.is_grad_enabled() and x.requires_grad:
            return _MRSTFTFn.apply(x, self, *targets)
        return self._forward(x, targets)[0]

    def _streams(self, device):
        """One side stream per resolution (created once per device).  The kernels of the three resolutions are latency
        bound at low occupancy and independent of each other: issued on three streams they overlap (forward and backward
        of the configs[4] loss).  The results are joined on the caller's stream in a fixed order."""
        pool = self.__dict__.setdefault("_side_streams", {})
        if device not in pool:
            pool[device] = [torch.cuda.Stream(device) for _ in self.plans]
        return pool[device]

    def _forward(self, x, targets):
        cur = torch.cuda.current_stream(x.device) if x.is_cuda else None
        streams = self._streams(x.device) if (x.is_cuda and self.parallel) else None
        sums = []
        for k, (plan, tgt) in enumerate(zip(self.plans, targets)):
            if streams is None:
                sums.append(plan.loss_sums(x, tgt, VALUE_MAG_CLAMPED, LOSS_MRSTFT, self.eps))
            else:
                streams[k].wait_stream(cur)
                with torch.cuda.stream(streams[k]):
                    s = plan.loss_sums(x, tgt, VALUE_MAG_CLAMPED, LOSS_MRSTFT, self.eps)
                    s.record_stream(cur)
                    sums.append(s)
        saved = []
        for k, (tgt, s) in enumerate(zip(targets, sums)):
            if streams is not None:
                cur.wait_stream(streams[k])
            saved.append((tgt, s))
        # (sum_k sqrt(s_k[0]) / sqrt(s_k[1]) + s_k[2] / count_k) / nres in fp64 -> fp32, one launch
        n = len(sums)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        ptrs = (ctypes.c_void_p * n)(*[s.data_ptr() for s in sums])
        counts = (ctypes.c_double * n)(*[float(t.numel()) for t in targets])
        _lib.check(_lib.load().ias_mrstft_total(ptrs, counts, n, _lib.ptr(loss), _lib.stream()), "ias_mrstft_total")
        return loss, saved


class _MRSTFTFn(torch.autograd.Function):
    """MultiResolutionSTFTLoss with the HIP adjoint w.r.t. the prediction (the target gets no gradient)."""

    @staticmethod
    def forward(ctx, x, module, *targets):
        a = STFTPlan._audio2d(x)
        loss, saved = module._forward(a, targets)
        ctx.module, ctx.shape = module, x.shape
        ctx.save_for_backward(a, *[t for pair in saved for t in pair])
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        a, *rest = ctx.saved_tensors
        module = ctx.module
        lib = _lib.load()
        B, T = a.shape
        nres = len(module.plans)
        g32 = g_loss.to(torch.float32).reshape(()).contiguous()
        cur = torch.cuda.current_stream(a.device)
        streams = module._streams(a.device) if module.parallel else None
        eps = float(module.eps)
        # chunk plans of the fused overlap-add (csrc/spectral_kernels.hip, SPAN kernels); any unsupported shape -> the
        # per-resolution path (frame tensor / spans + one combine each, summed with torch)
        plans = []
        for plan in module.plans:
            hp = (ctypes.c_int * 3)()
            if module.fused_combine and lib.ias_stft_grad_span_plan(B, T, plan.n_fft, plan.hop_length, 0, plan.n_out, hp) == 0:
                plans.append((hp[0], hp[1], hp[2]))
        if len(plans) == nres and nres <= 8:
            spans = []
            for i, plan in enumerate(module.plans):
                tgt, s = rest[2 * i], rest[2 * i + 1]
                if streams is not None:
                    streams[i].wait_stream(cur)
                with torch.cuda.stream(streams[i] if streams is not None else cur):
                    spans.append(_mrstft_plan_spans(lib, plan, a, tgt, s, g32, nres, eps, plans[i], cur))
            if streams is not None:
                for st in streams[:nres]:
                    cur.wait_stream(st)
            g_total = torch.empty_like(a)
            ptrs = (ctypes.c_void_p * nres)(*[sp.data_ptr() for sp in spans])
            flat = []
            for plan, (G, cper, L) in zip(module.plans, plans):
                flat += [plan.n_fft, plan.hop_length, G, cper, L]
            _lib.check(lib.ias_stft_grad_combine(ptrs, (ctypes.c_int * len(flat))(*flat), nres, None, _lib.ptr(g_total),
                                                 B, T, _lib.stream()), "ias_stft_grad_combine")
            return (g_total.reshape(ctx.shape), None) + (None,) * nres
        grads = []
        for i, plan in enumerate(module.plans):
            tgt, s = rest[2 * i], rest[2 * i + 1]
            if streams is not None:
                streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i] if streams is not None else cur):
                grads.append(_mrstft_plan_backward(lib, plan, a, tgt, s, g32, nres, eps, cur))
        g_total = None
        for i, g_audio in enumerate(grads):      # joined in a fixed order
            if streams is not None:
                cur.wait_stream(streams[i])
            g_total = g_audio if g_total is None else g_total + g_audio
        return (g_total.reshape(ctx.shape), None) + (None,) * len(module.plans)


def _mrstft_coef(lib, s, g32, count, nres, device):
    """[g / (nres sqrt(l0) sqrt(l1)) or 0, g / (nres count)]: the cotangent coefficients of one resolution
    (d (sqrt(l0) / sqrt(l1)) / dV = (V - T) / (sqrt(l0) sqrt(l1));  d (l2 / count) / dV = sign(V - T) / (V count))."""
    coef = torch.empty(2, dtype=torch.float64, device=device)
    _lib.check(lib.ias_mrstft_coef(_lib.ptr(s), _lib.ptr(g32), float(count), nres, _lib.ptr(coef), _lib.stream()),
               "ias_mrstft_coef")
    return coef


def _mrstft_plan_spans(lib, plan, a, tgt, s, g32, nres, eps, chunk_plan, consumer_stream):
    """One resolution's chunk spans of d loss / d (windowed frames), overlap-added inside the kernel (on the current
    stream; the result is handed to ``consumer_stream``, where ``ias_stft_grad_combine`` finishes all resolutions)."""
    B, T = a.shape
    g32.record_stream(torch.cuda.current_stream(a.device))
    coef = _mrstft_coef(lib, s, g32, tgt.numel(), nres, a.device)
    G, cper, L = chunk_plan
    spans = torch.empty(B * cper * L, dtype=torch.float32, device=a.device)
    hp = (ctypes.c_int * 3)()
    st = lib.ias_stft_grad_spans(_lib.ptr(a), _lib.ptr(plan.tables), None, None, None, None, 0, plan.n_out, _lib.ptr(tgt),
                                 _lib.ptr(coef), _lib.ptr(spans), B, T, plan.n_fft, plan.hop_length, 1, LOSS_MRSTFT, 0.0,
                                 eps, hp, _lib.stream())
    _lib.check(st, "ias_stft_grad_spans")
    assert (hp[0], hp[1], hp[2]) == chunk_plan
    spans.record_stream(consumer_stream)
    return spans


def _mrstft_plan_backward(lib, plan, a, tgt, s, g32, nres, eps, consumer_stream):
    """One resolution's d loss / d audio (on the current stream; the result is handed to ``consumer_stream``)."""
    B, T = a.shape
    g32.record_stream(torch.cuda.current_stream(a.device))
    coef = _mrstft_coef(lib, s, g32, tgt.numel(), nres, a.device)
    frame_grad = torch.empty((B, plan.num_frames(T), plan.n_fft), dtype=torch.float32, device=a.device)
    g_audio = torch.empty_like(a)
    st = lib.ias_stft_loss_backward(_lib.ptr(a), _lib.ptr(plan.window), _lib.ptr(plan.tables), None, None, None,
                                    None, 0, _lib.ptr(tgt), None, _lib.ptr(coef), _lib.ptr(frame_grad),
                                    _lib.ptr(g_audio), B, T, plan.n_fft, plan.hop_length, plan.n_out, 1, LOSS_MRSTFT,
                                    0.0, eps, _lib.stream())
    _lib.check(st, "ias_stft_loss_backward")
    g_audio.record_stream(consumer_stream)
    return g_audio


class ParallelLossSum(nn.Module):
    """sum_k loss_k(audio, **kwargs_k) with the terms issued on streams of their own: the first on the caller's stream,
    every other one on a side stream forked from it (and joined before the sum).  autograd runs a node's backward on the
    stream its forward ran on, so the terms' backward passes overlap the same way.  configs[4] pairs the 3-resolution
    STFT loss with the 64-band sub-band L1: issued one after the other, the sub-band branch (PQMF analysis, L1, its
    gradient and the PQMF adjoint: 0.17 ms) waited behind the STFT kernels, whose tails leave most of the chip idle."""

    def __init__(self, *losses):
        super().__init__()
        self.losses = nn.ModuleList(losses)
        self.parallel = True
        self.share_from = 0       # which of the first term's side streams the extra terms borrow (-1: streams of their own)

    def _streams(self, device):
        # A first term with side streams of its own (MultiResolutionSTFTLoss: one per resolution) lends them: the extra
        # terms then queue behind its SHORTER resolutions, forward and backward, instead of opening more branches than the
        # device has hardware queues (a fifth branch of the captured step was run behind the longest one).
        first = self.losses[0]
        if self.share_from >= 0 and hasattr(first, "_streams") and getattr(first, "parallel", False):
            own = list(first._streams(device))
            own = own[self.share_from:] + own[:self.share_from]
            if len(own) >= len(self.losses) - 1:
                return own[:len(self.losses) - 1]
        pool = self.__dict__.setdefault("_side_streams", {})
        if device not in pool:
            pool[device] = [torch.cuda.Stream(device) for _ in self.losses[1:]]
        return pool[device]

    def forward(self, audio, kwargs_list):
        assert len(kwargs_list) == len(self.losses)
        if not (audio.is_cuda and self.parallel) or len(self.losses) == 1:
            return sum(m(audio, **kw) for m, kw in zip(self.losses, kwargs_list))
        cur = torch.cuda.current_stream(audio.device)
        side = self._streams(audio.device)
        terms = []
        for s, m, kw in zip(side, list(self.losses)[1:], kwargs_list[1:]):      # fork: they wait for the audio only
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                terms.append(m(audio, **kw))
        total = self.losses[0](audio, **kwargs_list[0])
        for s, t in zip(side, terms):                                               # join
            cur.wait_stream(s)
            t.record_stream(cur)
            total = total + t
        return total


class SubbandL1(nn.Module):
    """mean |PQMF(a) - PQMF(target)|: an L1 loss in the sub-band domain of a ``pqmf.PQMF`` filterbank (BASELINE
    configs[4] pairs the 64-band PQMF with the multi-resolution STFT loss in one gradient step; the reference has no
    live code for either, /root/reference/conf/config.yaml:51-61, audio_to_params.py:233).  Differentiable with
    respect to the audio through the HIP adjoint of the analysis (pqmf._AnalysisFn)."""

    def __init__(self, gram):
        super().__init__()
        self.gram = gram

    def target(self, target_audio):
        a = target_audio.detach()
        return self.gram(a if a.dim() == 3 else a.unsqueeze(1))

    def forward(self, audio, target_audio=None, target_bands=None):
        if target_bands is None:
            target_bands = self.target(target_audio)
        z = self.gram(audio if audio.dim() == 3 else audio.unsqueeze(1))
        return l1_mean(z, target_bands.detach())


class _L1MeanFn(torch.autograd.Function):
    """mean |x - y| with the gradient w.r.t. x: two fused launches each way (csrc/spectral_grad_kernels.hip) instead of
    the sub / abs / mean and sign / mul / expand chains."""

    @staticmethod
    def forward(ctx, x, y):
        lib = _lib.load()
        xc, yc = x.detach().contiguous(), y.contiguous()
        _lib.require_f32(xc, yc)
        assert xc.shape == yc.shape
        n = xc.numel()
        nwg = lib.ias_l1_partials_count(n)
        partials = torch.empty((nwg, 3), dtype=torch.float64, device=xc.device)
        _lib.check(lib.ias_l1_partials(_lib.ptr(xc), _lib.ptr(yc), n, _lib.ptr(partials), _lib.stream()), "ias_l1_partials")
        sums = torch.empty(3, dtype=torch.float64, device=xc.device)
        mean = torch.empty((), dtype=torch.float32, device=xc.device)
        _lib.check(lib.ias_reduce_partials(_lib.ptr(partials), nwg, _lib.ptr(sums), 1.0 / n, _lib.ptr(mean), _lib.stream()),
                   "ias_reduce_partials")
        ctx.save_for_backward(xc, yc)
        return mean

    @staticmethod
    def backward(ctx, g):
        xc, yc = ctx.saved_tensors
        lib = _lib.load()
        g32 = g.to(torch.float32).reshape(()).contiguous()
        gx = torch.empty_like(xc)
        _lib.check(lib.ias_l1_grad(_lib.ptr(xc), _lib.ptr(yc), _lib.ptr(g32), 1.0 / xc.numel(), xc.numel(), _lib.ptr(gx),
                                   _lib.stream()), "ias_l1_grad")
        return gx, None


def l1_mean(x, y):
    """mean |x - y| (fp32 tensors of one shape on the device; differentiable w.r.t. x)."""
    return _L1MeanFn.apply(x, y)
