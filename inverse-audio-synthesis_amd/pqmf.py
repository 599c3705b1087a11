"""PQMF filterbank module -- drop-in for /root/reference/pqmf.py:9-55 on MI355X.

Same constructor, attributes (``N, taps, cutoff, beta, pad_fn``), registered buffers
(``H [N,1,taps+1]``, ``G [1,N,taps+1]``, ``updown_filter [N,N,N]`` -> identical
state_dict keys) and methods (``forward == analysis``, ``synthesis``).  The filter design
is host-side numpy/scipy at construction; analysis/synthesis run the HIP kernels of
csrc/pqmf_kernels.hip (synthesis in polyphase form -- the zero-stuffed tensor of
pqmf.py:53 is never materialised).
"""
import numpy as np
import torch
from scipy import signal as _sig

from . import _lib


def design_filters(N, taps, cutoff, beta):
    """Kaiser-windowed prototype, cosine-modulated into N analysis/synthesis filters."""
    proto = _sig.firwin(taps + 1, cutoff, window=("kaiser", beta))
    k = np.arange(N)[:, None]
    n = np.arange(taps + 1)[None, :] - ((taps - 1) / 2)
    arg = (2 * k + 1) * (np.pi / (2 * N)) * n
    phase = ((-1.0) ** k) * np.pi / 4
    H = 2 * proto[None, :] * np.cos(arg + phase)
    G = 2 * proto[None, :] * np.cos(arg - phase)
    return H, G


_PACKED_CACHE = {}


def _packed_taps(H, Hc, N, K):
    """Duplicated (h, h) tap table of H for the fast kernel, cached per (storage address, tensor version, device): an
    in-place write to H (or a new H) packs again.  (A module-level table like ``_MODTAB_CACHE``; round 2 hung the cache
    off the tensor object.)"""
    lib = _lib.load()
    n = lib.ias_pqmf_packed_taps_len(N, K)
    if n <= 0:
        return None
    key = (Hc.data_ptr(), H._version, str(Hc.device), N, K)
    hit = _PACKED_CACHE.get(key)
    if hit is None:
        packed = torch.empty(n, dtype=torch.float32, device=Hc.device)
        _lib.check(lib.ias_pqmf_pack_taps(_lib.ptr(Hc), _lib.ptr(packed), N, K, _lib.stream()), "ias_pqmf_pack_taps")
        if len(_PACKED_CACHE) > 64:
            _PACKED_CACHE.clear()
        hit = _PACKED_CACHE[key] = (packed, Hc)          # Hc kept alive: its address is part of the key
    return hit[0]


# N = 3: evaluate the filterbank in its cosine-modulated form (csrc/pqmf_kernels.hip: pqmf_analysis_mod_kernel) when H is
# that modulation of one prototype (it is for every PQMF(3) this module builds); False: the tap-ordered kernels
USE_MODULATED = True
_MODTAB_CACHE = {}


def _modulated_taps(H, Hc, N, K):
    """Device table of the modulated-form kernel for this H, or None (other N / K, H not a cosine modulation).  Built
    once per (tensor, version): the first call reads H back to the host."""
    if not USE_MODULATED or N != 3 or K != 63:
        return None
    key = (Hc.data_ptr(), H._version, str(Hc.device))
    hit = _MODTAB_CACHE.get(key)
    if hit is None:
        import ctypes
        lib = _lib.load()
        host = Hc.detach().to("cpu", torch.float32).contiguous()
        out = torch.empty(lib.ias_pqmf_modtab_len(), dtype=torch.float32)
        st = lib.ias_pqmf_build_modtab(ctypes.c_void_p(host.data_ptr()), N, K, ctypes.c_void_p(out.data_ptr()))
        if st == -2:
            hit = (None,)
        else:
            _lib.check(st, "ias_pqmf_build_modtab")
            hit = (out.to(Hc.device),)
        if len(_MODTAB_CACHE) > 64:
            _MODTAB_CACHE.clear()
        hit = hit + (Hc,)                                 # Hc kept alive: its address is part of the key
        _MODTAB_CACHE[key] = hit
    return hit[0]


class _AnalysisFn(torch.autograd.Function):
    """PQMF analysis, differentiable w.r.t. the audio.  The adjoint of the strided correlation
    z[k,f] = sum_j H[k,j] x[N f + j - pad] is the polyphase synthesis kernel run with the time-reversed filters and
    without its gain N (pad = (K-1)/2 makes the two index maps mirror images), cropped to the input length."""

    @staticmethod
    def forward(ctx, x, H, mean, std):
        ctx.shape = x.shape
        ctx.save_for_backward(H, std)
        return _analysis_nograd(x, H, mean, std)

    @staticmethod
    def backward(ctx, g_z):
        H, std = ctx.saved_tensors
        T = ctx.shape[-1]
        N = H.shape[0]
        g = g_z.to(torch.float32)
        if std is not None:
            g = g / std.reshape(1, N, 1)
        LN = g.shape[-1] * N
        g_x = pqmf_synthesis(g, _adjoint_filters(H), out_len=min(T, LN))[:, 0, :]
        if g_x.shape[-1] < T:
            g_x = torch.nn.functional.pad(g_x, (0, T - g_x.shape[-1]))
        return g_x.reshape(ctx.shape), None, None, None


def pqmf_analysis(x, H, mean=None, std=None, rowpeak=None):
    """x [B,1,T] or [B,T] fp32 on a ROCm device, H [N,1,K] -> z [B,N,L].

    ``mean``/``std`` ([N] device tensors) fuse the per-band normalisation of
    /root/reference/audioembed.py:49 into the store.  Differentiable with respect to ``x`` (not H).
    ``rowpeak`` [B]: row peaks of an un-normalised render (``Voice.peaks_view``): the analysis of x / peak where
    peak > 1 (torchsynth normalize_if_clipping folded in; forward only).
    """
    if torch.is_grad_enabled() and x.requires_grad:
        assert rowpeak is None, "the folded normalisation is a forward-only path"
        return _AnalysisFn.apply(x, H, mean, std)
    return _analysis_nograd(x, H, mean, std, rowpeak)


def _analysis_nograd(x, H, mean=None, std=None, rowpeak=None):
    lib = _lib.load()
    if x.dim() == 3:
        assert x.shape[1] == 1, "PQMF analysis takes one input channel"
    x2 = x.reshape(x.shape[0], -1).contiguous()
    Hc = H.reshape(H.shape[0], -1).contiguous()
    _lib.require_f32(x2, Hc, mean, std, rowpeak)
    B, T = x2.shape
    N, K = Hc.shape
    L = lib.ias_pqmf_out_len(T, N, K)
    _lib.check(min(L, 0), "ias_pqmf_out_len")
    z = torch.empty((B, N, L), dtype=torch.float32, device=x2.device)
    st = lib.ias_pqmf_analysis(_lib.ptr(x2), _lib.ptr(Hc), _lib.ptr(_packed_taps(H, Hc, N, K)),
                               _lib.ptr(_modulated_taps(H, Hc, N, K)), _lib.ptr(z),
                               _lib.ptr(mean), _lib.ptr(std), _lib.ptr(rowpeak), B, T, N, K, _lib.stream())
    _lib.check(st, "ias_pqmf_analysis")
    return z


_ADJOINT_CACHE = {}
_SYNTH_CACHE = {}


def _adjoint_filters(H):
    """G [1,N,K] = flip(H) / N: the synthesis filters whose polyphase synthesis is the adjoint of the analysis with H.
    Cached per (storage address, version, device) of H -- the backward built it (flip, divide, pack: three launches) on
    every call.  The cache holds H, so its address cannot be reused while the entry lives."""
    key = (H.data_ptr(), H._version, str(H.device), tuple(H.shape))
    hit = _ADJOINT_CACHE.get(key)
    if hit is None:
        N = H.shape[0]
        G = (torch.flip(H.detach().reshape(N, -1), dims=[1]).reshape(1, N, -1) / float(N)).contiguous()
        if len(_ADJOINT_CACHE) > 64:
            _ADJOINT_CACHE.clear()
        hit = _ADJOINT_CACHE[key] = (G, H)
    return hit[0]


def _packed_synth_taps(G, Gc, N, K):
    """Phase-major tap table of G for the wide synthesis kernel, cached like the analysis one."""
    lib = _lib.load()
    n = lib.ias_pqmf_synth_taps_len(N, K)
    if n <= 0:
        return None
    key = (Gc.data_ptr(), G._version, str(Gc.device), N, K)
    hit = _SYNTH_CACHE.get(key)
    if hit is None:
        packed = torch.empty(n, dtype=torch.float32, device=Gc.device)
        _lib.check(lib.ias_pqmf_pack_synth_taps(_lib.ptr(Gc), _lib.ptr(packed), N, K, _lib.stream()),
                   "ias_pqmf_pack_synth_taps")
        if len(_SYNTH_CACHE) > 64:
            _SYNTH_CACHE.clear()
        hit = _SYNTH_CACHE[key] = (packed, Gc)            # Gc kept alive: its address is part of the key
    return hit[0]


def pqmf_synthesis(z, G, out_len=None):
    """z [B,N,L], G [1,N,K] -> [B,1,L*N]  (``out_len``: only the first ``out_len`` <= L*N samples, as a contiguous
    [B,1,out_len] -- what the analysis' adjoint wants)."""
    lib = _lib.load()
    zc = z.contiguous()
    Gc = G.reshape(-1, G.shape[-1]).contiguous()
    _lib.require_f32(zc, Gc)
    B, N, L = zc.shape
    assert Gc.shape[0] == N
    To = L * N if out_len is None else int(out_len)
    assert 0 < To <= L * N
    out = torch.empty((B, 1, To), dtype=torch.float32, device=zc.device)
    st = lib.ias_pqmf_synthesis_t(_lib.ptr(zc), _lib.ptr(Gc), _lib.ptr(_packed_synth_taps(G, Gc, N, Gc.shape[1])),
                                  _lib.ptr(out), B, L, N, Gc.shape[1], To, _lib.stream())
    _lib.check(st, "ias_pqmf_synthesis_t")
    return out


class PQMF(torch.nn.Module):
    def __init__(self, N=4, taps=62, cutoff=0.15, beta=9.0):
        super().__init__()
        self.N = N
        self.taps = taps
        self.cutoff = cutoff
        self.beta = beta
        H, G = design_filters(N, taps, cutoff, beta)
        self.register_buffer("H", torch.from_numpy(H[:, None, :]).float())
        self.register_buffer("G", torch.from_numpy(G[None, :, :]).float())
        updown = torch.zeros((N, N, N)).float()
        updown[torch.arange(N), torch.arange(N), 0] = 1.0
        self.register_buffer("updown_filter", updown)
        self.pad_fn = torch.nn.ConstantPad1d(taps // 2, 0.0)

    def forward(self, x):
        return self.analysis(x)

    def analysis(self, x, rowpeak=None):
        return pqmf_analysis(x, self.H, rowpeak=rowpeak)

    def synthesis(self, x):
        return pqmf_synthesis(x, self.G)
