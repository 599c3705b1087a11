"""HIP PQMF vs the oracle and the reference's golden vectors, through the C ABI.  fp32 FIR sums in a
different order than torch's conv1d: tolerance 2e-5 absolute on O(1) data (1e-4 rel bar of the path)."""
import os

import numpy as np
import pytest
import torch

from oracle import pqmf_oracle as po
from helpers import randn, checks

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _mod(dev, N=3, **kw):
    from inverse_audio_synthesis_amd.pqmf import PQMF
    return PQMF(N=N, **kw).to(dev)


def test_analysis_golden_small_and_edges(lib, dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "pqmf_analysis.npz"))
    m3, m4 = _mod(dev, 3), _mod(dev, 4)
    z = m3(randn((4, 1, 16000), 101).to(dev))
    assert z.shape == (4, 3, 5334)
    np.testing.assert_allclose(z.cpu().numpy(), g["small_z3"], atol=TOL)
    for T in (1, 31, 62, 63, 64, 1000, 1001):
        xe = randn((2, 1, T), 200 + T).to(dev)
        np.testing.assert_allclose(m4(xe).cpu().numpy(), g[f"edge_T{T}_z4"], atol=TOL)
        np.testing.assert_allclose(m3(xe).cpu().numpy(), g[f"edge_T{T}_z3"], atol=TOL)


def test_analysis_golden_full_length(lib, dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "pqmf_analysis.npz"))
    x = randn((2, 1, 176400), 102).to(dev)
    for N, tag in ((3, "z3"), (64, "z64")):
        z = _mod(dev, N)(x).cpu()
        assert list(z.shape) == list(g[f"full_{tag}_shape"])
        np.testing.assert_allclose(z.flatten()[::97].numpy(), g[f"full_{tag}_sub"], atol=TOL)
        np.testing.assert_allclose(checks(z), g[f"full_{tag}_checks"], rtol=1e-5)


# N = 3, 4 with 62 taps: fast kernel; other N <= 64 with K <= 255: wide kernel (pow2 and non-pow2 N, K > N and
# K < N, ragged T); N > 64 or K > 255: generic kernel
@pytest.mark.parametrize("N,taps,T", [(3, 62, 176400), (4, 62, 4099), (8, 30, 5000), (64, 62, 20000), (2, 62, 77),
                                      (5, 62, 12345), (33, 20, 9999), (64, 254, 30000), (16, 62, 1),
                                      (96, 62, 20000), (4, 300, 5000)])
def test_analysis_vs_oracle(lib, dev, N, taps, T):
    kw = dict(taps=taps) if taps == 62 else dict(taps=taps, cutoff=0.07, beta=7.0)
    m = _mod(dev, N, **kw)
    x = randn((3, 1, T), 900 + N)
    ref = po.analysis(x, m.H.cpu(), N, taps)
    np.testing.assert_allclose(m(x.to(dev)).cpu().numpy(), ref.numpy(), atol=TOL)


def test_synthesis_golden(lib, dev, golden_dir):
    g = np.load(os.path.join(golden_dir, "pqmf_synthesis.npz"))
    for N in (3, 4, 64):
        m = _mod(dev, N)
        y = m.synthesis(torch.from_numpy(g[f"z{N}"]).to(dev))
        assert y.shape == g[f"y{N}"].shape
        np.testing.assert_allclose(y.cpu().numpy(), g[f"y{N}"], atol=1e-4)


def test_fused_preprocess_golden(lib, dev, golden_dir):
    from inverse_audio_synthesis_amd.pqmf import pqmf_analysis
    g = np.load(os.path.join(golden_dir, "audioembed_preprocess.npz"))
    m = _mod(dev, 3)
    mean = torch.tensor([0.485, 0.456, 0.406], device=dev)
    std = torch.tensor([0.229, 0.224, 0.225], device=dev)
    img = pqmf_analysis(randn((2, 1, 176400), 104).to(dev), m.H, mean, std).reshape(-1, 3, 240, 245).cpu()
    assert list(img.shape) == list(g["shape"])
    np.testing.assert_allclose(img.flatten()[::89].numpy(), g["sub"], atol=1e-4)
    np.testing.assert_allclose(checks(img), g["checks"], rtol=1e-5)


def test_full_size_linearity_and_shift(lib, dev):
    """BASELINE size [128,1,176400]: linearity, and a shift by N samples shifts the bands by one frame."""
    m = _mod(dev, 3)
    g = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn((128, 1, 176400), generator=g).to(dev)
    b = torch.randn((128, 1, 176400), generator=g).to(dev)
    za, zb, zs = m(a), m(b), m(2.0 * a - 0.5 * b)
    assert za.shape == (128, 3, 58800)
    assert (zs - (2.0 * za - 0.5 * zb)).abs().max().item() <= 1e-4
    sh = torch.zeros_like(a)
    sh[:, :, 3:] = a[:, :, :-3]
    zsh = m(sh)
    assert (zsh[:, :, 12:-12] - za[:, :, 11:-13]).abs().max().item() <= 1e-5


def test_packed_taps_follow_filter_updates(lib, dev):
    """The packed tap table of the fast path is rebuilt when H is written in place or replaced, and the C-ABI call
    without it (packed = NULL: generic kernel, same j-ascending FMA chain) gives the same output."""
    from inverse_audio_synthesis_amd import _lib
    from inverse_audio_synthesis_amd.pqmf import pqmf_analysis
    m = _mod(dev, 3)
    x = randn((2, 1, 9000), 77).to(dev)
    z0 = m(x).clone()
    with torch.no_grad():
        m.H.mul_(0.5)
    z1 = m(x)
    np.testing.assert_allclose(z1.cpu().numpy(), po.analysis(x.cpu(), m.H.cpu(), 3, 62).numpy(), atol=TOL)
    assert not torch.equal(z0, z1)
    for seed in range(3):                     # fresh filters, possibly at a recycled address
        H = (randn((3, 1, 63), 500 + seed) * 0.1).to(dev)
        np.testing.assert_allclose(pqmf_analysis(x, H).cpu().numpy(), po.analysis(x.cpu(), H.cpu(), 3, 62).numpy(),
                                   atol=TOL)
        del H
    z2 = torch.empty_like(z1)
    Hc = m.H.reshape(3, 63).contiguous()
    x2 = x.reshape(2, -1)
    st = lib.ias_pqmf_analysis(_lib.ptr(x2), _lib.ptr(Hc), None, None, _lib.ptr(z2), None, None, None, 2, 9000, 3, 63,
                               _lib.stream())
    assert st == 0
    np.testing.assert_allclose(z2.cpu().numpy(), z1.cpu().numpy(), atol=1e-6)


@pytest.mark.parametrize("N,taps,T", [(3, 62, 20000), (4, 62, 4099), (64, 62, 20000), (5, 30, 1234)])
def test_analysis_gradient_matches_conv1d_autograd(lib, dev, N, taps, T):
    """d/dx of PQMF.analysis (HIP: the synthesis kernel with reversed filters) against torch.autograd through the
    reference's own formulation F.conv1d(x, H, padding=taps//2, stride=N) (pqmf.py:49-50) in fp64, also with the fused
    per-band normalisation of AudioEmbedding._preprocess."""
    from inverse_audio_synthesis_amd.pqmf import pqmf_analysis
    kw = dict(taps=taps) if taps == 62 else dict(taps=taps, cutoff=0.07, beta=7.0)
    m = _mod(dev, N, **kw)
    x = randn((2, 1, T), 31)
    xd = x.double().requires_grad_(True)
    zr = torch.nn.functional.conv1d(xd, m.H.cpu().double(), padding=taps // 2, stride=N)
    w = randn(tuple(zr.shape), 32)
    (ref,) = torch.autograd.grad((zr * w.double()).sum(), xd, retain_graph=True)
    xa = x.to(dev).requires_grad_(True)
    (m(xa) * w.to(dev)).sum().backward()
    assert xa.grad.shape == x.shape
    np.testing.assert_allclose(xa.grad.cpu().numpy(), ref.float().numpy(), atol=2e-5 * float(ref.abs().max()) + 1e-6)
    if N == 3:
        mean, std = torch.tensor([0.1, 0.2, 0.3]).to(dev), torch.tensor([0.5, 2.0, 1.5]).to(dev)
        xb = x.to(dev).requires_grad_(True)
        (pqmf_analysis(xb, m.H, mean, std) * w.to(dev)).sum().backward()
        (ref2,) = torch.autograd.grad(((zr - mean.cpu().double().reshape(1, 3, 1)) / std.cpu().double().reshape(1, 3, 1)
                                       * w.double()).sum(), xd)
        np.testing.assert_allclose(xb.grad.cpu().numpy(), ref2.float().numpy(), atol=2e-5 * float(ref2.abs().max()) + 1e-6)


# wide synthesis kernel (N <= 64, K <= 255) and the generic one beyond that, against the oracle's
# conv_transpose1d + conv1d formulation (pqmf.py:52-55)
@pytest.mark.parametrize("N,taps,L", [(3, 62, 5000), (4, 62, 1), (5, 30, 777), (8, 30, 1000), (33, 20, 300),
                                      (64, 254, 400), (64, 62, 129), (96, 62, 200)])
def test_synthesis_vs_oracle(lib, dev, N, taps, L):
    kw = dict(taps=taps) if taps == 62 else dict(taps=taps, cutoff=0.07, beta=7.0)
    m = _mod(dev, N, **kw)
    z = randn((2, N, L), 700 + N)
    ref = po.synthesis(z, m.G.cpu(), m.updown_filter.cpu(), N, taps)
    got = m.synthesis(z.to(dev)).cpu()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-4 * max(1.0, float(ref.abs().max())))


# (3, *, T) with T % 4 == 0 and L % 4 == 0 (L = (T - 1) // 3 + 1): the wave-pipelined kernel, down to a single, mostly
# empty wave tile (T = 12: L = 4) and with more rows than resident waves' first tiles; other N = 3 shapes: the tiled kernel
@pytest.mark.parametrize("N,B,T", [(3, 5, 176400), (3, 2, 4100), (4, 3, 20004), (64, 2, 176400), (64, 3, 8192),
                                   (3, 3, 12), (3, 2, 1020), (3, 1, 96000), (3, 130, 3840), (3, 7, 960 * 3 + 12)])
def test_matrix_core_path_is_bit_identical_to_the_vector_kernels(lib, dev, N, B, T):
    """The fp32 MFMA analysis (16-byte aligned rows, T % 4 == 0) and the VALU kernels (taken for any other alignment)
    evaluate the same tap-ordered fmaf chain per output: the results must agree bit for bit, with and without the
    fused band normalisation and row scale.  A misaligned view of the same samples selects the VALU kernels."""
    from inverse_audio_synthesis_amd import _lib
    m = _mod(dev, N)
    Hc = m.H.reshape(N, 63).contiguous()
    packed = torch.empty(lib.ias_pqmf_packed_taps_len(N, 63), device=dev)
    assert lib.ias_pqmf_pack_taps(_lib.ptr(Hc), _lib.ptr(packed), N, 63, _lib.stream()) == 0
    x = randn((B, T), 4000 + N).to(dev)
    x[0, :40] = 0.0
    x[-1, -100:] *= 1e-30                              # subnormal products
    xa = x.contiguous()
    xm = torch.empty(B * T + 1, device=dev)[1:]
    xm.copy_(x.flatten())
    assert _lib.ptr(xa).value % 16 == 0 and _lib.ptr(xm).value % 16 != 0
    L = lib.ias_pqmf_out_len(T, N, 63)
    mean = torch.linspace(-0.1, 0.1, N, device=dev)
    std = torch.linspace(0.5, 1.5, N, device=dev)
    peak = torch.linspace(0.5, 3.0, B, device=dev)
    for fused in (False, True):
        za = torch.full((B, N, L), 7.0, device=dev)
        zm = torch.full((B, N, L), 9.0, device=dev)
        args = (_lib.ptr(mean), _lib.ptr(std), _lib.ptr(peak)) if fused else (None, None, None)
        assert lib.ias_pqmf_analysis(_lib.ptr(xa), _lib.ptr(Hc), _lib.ptr(packed), None, _lib.ptr(za), *args, B, T, N, 63,
                                     _lib.stream()) == 0
        assert lib.ias_pqmf_analysis(_lib.ptr(xm), _lib.ptr(Hc), _lib.ptr(packed), None, _lib.ptr(zm), *args, B, T, N, 63,
                                     _lib.stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(za, zm), f"max diff {(za - zm).abs().max().item():.3e}"


@pytest.mark.parametrize("B,T", [(5, 176400), (2, 4100), (3, 12), (1, 1), (2, 31), (3, 1021), (130, 3840), (2, 100003)])
def test_modulated_form_agrees_with_the_tap_ordered_kernels(lib, dev, B, T):
    """N = 3: the cosine-modulated evaluation (52 signed prototype products + a 3-point modulation per frame, the
    default of PQMF(3).analysis) against the tap-ordered fmaf chain of the other kernels, through the C ABI: the same
    real numbers in another summation order -- 2e-6 of the output scale, with and without the fused band normalisation
    and row scale, aligned and misaligned rows, ragged and one-sample inputs."""
    import ctypes
    from inverse_audio_synthesis_amd import _lib
    m = _mod(dev, 3)
    Hc = m.H.reshape(3, 63).contiguous()
    host = Hc.cpu().contiguous()
    tab = torch.empty(lib.ias_pqmf_modtab_len())
    assert lib.ias_pqmf_build_modtab(ctypes.c_void_p(host.data_ptr()), 3, 63, ctypes.c_void_p(tab.data_ptr())) == 0
    tab = tab.to(dev)
    x = randn((B, T), 4100 + B).to(dev)
    xm = torch.empty(B * T + 1, device=dev)[1:]
    xm.copy_(x.flatten())
    L = lib.ias_pqmf_out_len(T, 3, 63)
    mean = torch.linspace(-0.1, 0.1, 3, device=dev)
    std = torch.linspace(0.5, 1.5, 3, device=dev)
    peak = torch.linspace(0.5, 3.0, B, device=dev)
    for fused in (False, True):
        args = (_lib.ptr(mean), _lib.ptr(std), _lib.ptr(peak)) if fused else (None, None, None)
        ref = torch.full((B, 3, L), 7.0, device=dev)
        assert lib.ias_pqmf_analysis(_lib.ptr(x), _lib.ptr(Hc), None, None, _lib.ptr(ref), *args, B, T, 3, 63,
                                     _lib.stream()) == 0
        for src in (x, xm):
            got = torch.full((B, 3, L), 9.0, device=dev)
            assert lib.ias_pqmf_analysis(_lib.ptr(src), _lib.ptr(Hc), None, _lib.ptr(tab), _lib.ptr(got), *args, B, T, 3, 63,
                                         _lib.stream()) == 0
            torch.cuda.synchronize()
            scale = max(1.0, ref.abs().max().item())
            assert (got - ref).abs().max().item() <= 2e-6 * scale
    # and the module takes it by default
    import inverse_audio_synthesis_amd.pqmf as pq
    assert pq.USE_MODULATED and pq._modulated_taps(m.H, Hc, 3, 63) is not None


def test_modulated_table_is_refused_for_other_filters(lib, dev):
    import ctypes
    H = torch.randn(3, 63)
    tab = torch.empty(lib.ias_pqmf_modtab_len())
    assert lib.ias_pqmf_build_modtab(ctypes.c_void_p(H.data_ptr()), 3, 63, ctypes.c_void_p(tab.data_ptr())) == -2
    assert lib.ias_pqmf_build_modtab(ctypes.c_void_p(H.data_ptr()), 4, 63, ctypes.c_void_p(tab.data_ptr())) == -2
