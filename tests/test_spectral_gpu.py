"""HIP STFT / mel / spectral losses vs the oracle (torch.stft based), through the C ABI.
Tolerances: spectral loss 1e-3 relative (BASELINE.json north_star); spectrogram values 1e-4 of the
spectrogram's max (fp32 FFT vs torch's fp32 FFT)."""
import pytest
import torch

from oracle import spectral_oracle as spo
from helpers import randn

pytestmark = pytest.mark.gpu
LOSS_RTOL = 1e-3


def _close_to_scale(a, ref, tol=1e-4):
    scale = ref.abs().max().item()
    err = (a - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err} vs scale {scale}"


@pytest.mark.parametrize("B,T,sr", [(4, 16000, 16000), (2, 176400, 44100), (3, 5001, 22050)])
def test_mel_spectrogram(lib, dev, B, T, sr):
    from inverse_audio_synthesis_amd.spectral import MelSpectrogram
    x = randn((B, T), 31 + B) * 0.3
    ref = spo.mel_spectrogram(x, sample_rate=sr)
    mel = MelSpectrogram(sample_rate=sr).to(dev)
    out = mel(x.to(dev))
    assert out.shape == ref.shape
    _close_to_scale(out.cpu(), ref)
    assert torch.equal(mel.plan.fb.cpu(), spo.melscale_fbanks(513, 0.0, float(sr // 2), 128, sr))


@pytest.mark.parametrize("n_fft,hop,win,power", [(512, 50, 240, 1.0), (1024, 120, 600, 2.0), (2048, 240, 1200, 1.0),
                                                 (1024, 512, None, 2.0), (2048, 2048, None, 2.0), (512, 77, 300, 2.0)])
def test_raw_stft_all_sizes(lib, dev, n_fft, hop, win, power):
    from inverse_audio_synthesis_amd.spectral import STFTPlan, VALUE_MAG, VALUE_POWER
    x = randn((3, 20000), 7) * 0.5
    ref = spo.spectrogram(x, n_fft, win, hop, power)  # [B, bins, frames]
    plan = STFTPlan(n_fft, win, hop).to(dev)
    out = plan.values(x.to(dev), VALUE_POWER if power == 2.0 else VALUE_MAG).transpose(1, 2)
    assert out.shape == ref.shape
    _close_to_scale(out.cpu(), ref)


def test_mel_l1_loss(lib, dev):
    from inverse_audio_synthesis_amd.spectral import MelSpectrogramL1
    a, b = randn((4, 16000), 1) * 0.2, randn((4, 16000), 2) * 0.3
    loss = MelSpectrogramL1(sample_rate=16000).to(dev)
    got = loss(a.to(dev), b.to(dev)).item()
    ref = spo.mel_l1(a, b, sample_rate=16000).item()
    assert abs(got - ref) <= LOSS_RTOL * abs(ref)
    # cached-target form gives the same value
    tm = loss.target(b.to(dev))
    assert loss(a.to(dev), target_mel=tm).item() == got


def test_stft_l1_and_mrstft(lib, dev):
    from inverse_audio_synthesis_amd.spectral import STFTL1, MultiResolutionSTFTLoss
    a, b = randn((4, 16000), 3) * 0.2, randn((4, 16000), 4) * 0.25
    got = STFTL1().to(dev)(a.to(dev), b.to(dev)).item()
    ref = spo.stft_l1(a, b).item()
    assert abs(got - ref) <= LOSS_RTOL * abs(ref)
    got = MultiResolutionSTFTLoss().to(dev)(a.to(dev), b.to(dev)).item()
    ref = spo.mrstft_loss(a, b)[0].item()
    assert abs(got - ref) <= LOSS_RTOL * abs(ref)


def test_short_and_ragged_inputs(lib, dev):
    from inverse_audio_synthesis_amd.spectral import MelSpectrogram
    mel = MelSpectrogram(sample_rate=16000).to(dev)
    for T in (513, 600, 1023, 1024, 1025, 4097):
        x = randn((2, T), T)
        ref = spo.mel_spectrogram(x, sample_rate=16000)
        out = mel(x.to(dev))
        assert out.shape == ref.shape
        _close_to_scale(out.cpu(), ref)
    with pytest.raises(RuntimeError):
        mel(torch.zeros(1, 512, device=dev))  # reflect padding needs T > n_fft/2


def test_full_size_properties(lib, dev):
    """BASELINE size 128 x 176400: loss(a,a)=0, power-2 mel is 2-homogeneous, loss is symmetric."""
    from inverse_audio_synthesis_amd.spectral import MelSpectrogramL1
    g = torch.Generator(device="cpu").manual_seed(9)
    a = (torch.rand((128, 176400), generator=g) - 0.5).to(dev)
    b = (torch.rand((128, 176400), generator=g) - 0.5).to(dev)
    loss = MelSpectrogramL1().to(dev)
    ma = loss.target(a)
    assert ma.shape == (128, 345, 128)
    assert loss(a, target_mel=ma).item() == 0.0
    m2 = loss.target(2.0 * a)
    assert (m2 - 4.0 * ma).abs().max().item() <= 1e-4 * ma.abs().max().item()
    lab, lba = loss(a, b).item(), loss(b, a).item()
    assert abs(lab - lba) <= 1e-6 * abs(lab)
    # rows are independent: the first 4 rows alone give the same mel
    assert torch.equal(loss.target(a[:4]), ma[:4])
    ref = spo.mel_spectrogram(a[:2].cpu())
    _close_to_scale(ma[:2].transpose(1, 2).cpu(), ref)


@pytest.mark.parametrize("n_fft,hop,win", [(1024, 512, None), (2048, 512, 1200), (512, 128, 300)])
def test_full_size_parseval(lib, dev, n_fft, hop, win):
    """BASELINE size 128 x 176400, a check that needs no FFT library (the STFT's parity is otherwise unpinned: torch.stft
    is what the oracle calls): for every frame, |X_0|^2 + |X_{N/2}|^2 + 2 sum_{0<k<N/2} |X_k|^2 = N sum_n (w_n x_n)^2 with
    the frame cut from the reflect-padded signal (centre = True) -- time-domain side in fp64 torch ops on the device.
    Tolerance 1e-4 of the largest frame energy (fp32 transform)."""
    from inverse_audio_synthesis_amd.spectral import STFTPlan, VALUE_POWER
    g = torch.Generator(device="cpu").manual_seed(11)
    x = ((torch.rand((128, 176400), generator=g) - 0.5) * torch.linspace(0.05, 1.0, 128)[:, None]).to(dev)
    plan = STFTPlan(n_fft, win, hop).to(dev)
    P = plan.values(x, VALUE_POWER).double()                     # [B, frames, n_fft/2 + 1]
    assert P.shape == (128, 176400 // hop + 1, n_fft // 2 + 1)
    spec = P[..., 0] + P[..., -1] + 2.0 * P[..., 1:-1].sum(-1)
    xp = torch.nn.functional.pad(x.double()[:, None], (n_fft // 2, n_fft // 2), mode="reflect")[:, 0]
    frames = xp.unfold(1, n_fft, hop)                            # [B, frames, n_fft]
    energy = n_fft * (frames * plan.window.double()).square().sum(-1)
    assert spec.shape == energy.shape
    assert (spec - energy).abs().max().item() <= 1e-4 * energy.max().item()
    # and per row, so that a quiet voice is held to its own scale
    rel = (spec - energy).abs().amax(1) / energy.amax(1)
    assert rel.max().item() <= 1e-4


def test_mrstft_cached_targets_equal_recomputed(lib, dev):
    """MultiResolutionSTFTLoss.target(y) caches the target magnitudes (a fixed target is then not transformed again
    every step); the loss is bit-identical either way."""
    from inverse_audio_synthesis_amd.spectral import MultiResolutionSTFTLoss
    m = MultiResolutionSTFTLoss().to(dev)
    x, y = (randn((2, 20000), 41) * 0.1).to(dev), (randn((2, 20000), 42) * 0.1).to(dev)
    tg = m.target(y)
    assert len(tg) == 3
    assert torch.equal(m(x, y), m(x, targets=tg))
    with pytest.raises(AssertionError):
        m(x)


@pytest.mark.parametrize("env", [{"IAS_STFT_MFMA": "1"}, {"IAS_STFT_V1": "1"}])
def test_alternative_stft_kernels_in_a_child_process(lib, dev, env):
    """The DIAGNOSTIC library (csrc/libias_hip_diag.so; the product library reads nothing from the environment and does
    not contain the matrix-core kernel) reads the kernel choice once per process from the environment: the opt-in
    matrix-core kernel (IAS_STFT_MFMA=1) and the round-2 kernel (IAS_STFT_V1=1) are checked in a
    child process (a fresh interpreter, not a re-exec of this one; IAS_HIP_LIB points the package at the diagnostic
    library) against the same oracle and tolerances as the default kernel: mel spectrogram, raw power
    spectrogram with a hop that is not a multiple of four samples, and the fused mel-L1 loss."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import sys, torch
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from oracle import spectral_oracle as spo
from helpers import randn
from inverse_audio_synthesis_amd.spectral import MelSpectrogram, MelSpectrogramL1, STFTPlan, VALUE_POWER
dev = torch.device("cuda:0")
x = randn((3, 20000), 7) * 0.5
def close(a, ref, tol=1e-4):
    assert (a - ref).abs().max().item() <= tol * ref.abs().max().item()
close(MelSpectrogram(sample_rate=44100).to(dev)(x.to(dev)).cpu(), spo.mel_spectrogram(x, sample_rate=44100))
for hop in (512, 50):
    plan = STFTPlan(1024, None, hop).to(dev)
    close(plan.values(x.to(dev), VALUE_POWER).transpose(1, 2).cpu(), spo.spectrogram(x, 1024, None, hop, 2.0))
a, b = randn((4, 16000), 1) * 0.2, randn((4, 16000), 2) * 0.3
got = MelSpectrogramL1(sample_rate=16000).to(dev)(a.to(dev), b.to(dev)).item()
ref = spo.mel_l1(a, b, sample_rate=16000).item()
assert abs(got - ref) <= 1e-3 * abs(ref), (got, ref)
print("child ok")
'''.replace("ROOT", repr(ROOT))
    from inverse_audio_synthesis_amd import _lib
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, IAS_HIP_LIB=_lib.DIAG_LIB_PATH, **env),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "child ok" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("B,T", [(3, 5200), (1, 700), (2, 5000)])
def test_stft_l1_at_512_with_odd_and_even_frame_counts(lib, dev, B, T):
    """n_fft 512 runs two frames per wave (stft2h_kernel): an odd total frame count leaves the last wave one real frame and
    one repeat that must not be counted; loss and values against the oracle, and the backward (stft_grad512_kernel, which
    pairs frames inside a chunk) against autograd through it."""
    from inverse_audio_synthesis_amd.spectral import STFTL1, STFTPlan, VALUE_POWER
    a, b = randn((B, T), 60 + B) * 0.2, randn((B, T), 70 + B) * 0.25
    m = STFTL1(n_fft=512, hop_length=128, power=2.0).to(dev)
    xa = a.to(dev).requires_grad_(True)
    got = m(xa, b.to(dev))
    ad = a.double().requires_grad_(True)
    ref = spo.stft_l1(ad, b.double(), n_fft=512, hop_length=128, power=2.0)
    assert abs(got.item() - ref.item()) <= LOSS_RTOL * abs(ref.item())
    got.backward()
    (g,) = torch.autograd.grad(ref, ad)
    assert torch.linalg.norm(xa.grad.cpu().double() - g) <= 2e-3 * torch.linalg.norm(g)
    out = STFTPlan(512, None, 128).to(dev).values(a.to(dev), VALUE_POWER).transpose(1, 2)
    _close_to_scale(out.cpu(), spo.spectrogram(a, 512, None, 128, 2.0))
