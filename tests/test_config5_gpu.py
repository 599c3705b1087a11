"""BASELINE config #5 shapes on one GPU's share (global 512 / 8 GPUs = 64 voices): 64-band PQMF and the
3-resolution STFT loss at 4 s @ 44.1 kHz, against the oracle on a sub-batch and via properties at full size."""
import pytest
import torch

from oracle import pqmf_oracle as po
from oracle import spectral_oracle as spo
from helpers import randn

pytestmark = pytest.mark.gpu


def test_pqmf64_full_length(lib, dev):
    from inverse_audio_synthesis_amd.pqmf import PQMF
    m = PQMF(N=64).to(dev)
    g = torch.Generator(device="cpu").manual_seed(64)
    x = torch.randn((64, 1, 176400), generator=g)
    z = m(x.to(dev))
    assert z.shape == (64, 64, 2757)
    ref = po.analysis(x[:3], m.H.cpu(), 64, 62)
    assert (z[:3].cpu() - ref).abs().max().item() <= 2e-5
    # synthesis length and linearity at full size
    y = m.synthesis(z)
    assert y.shape == (64, 1, 2757 * 64)
    y2 = m.synthesis(2.0 * z)
    assert (y2 - 2.0 * y).abs().max().item() <= 1e-4 * y.abs().max().item()
    yr = po.synthesis(z[:2].cpu(), m.G.cpu(), m.updown_filter.cpu(), 64, 62)
    assert (y[:2].cpu() - yr).abs().max().item() <= 1e-3 * yr.abs().max().item()


def test_mrstft_full_length(lib, dev):
    from inverse_audio_synthesis_amd.spectral import MultiResolutionSTFTLoss
    loss = MultiResolutionSTFTLoss().to(dev)
    a = randn((64, 176400), 1) * 0.1
    b = randn((64, 176400), 2) * 0.1 + 0.05 * torch.sin(torch.arange(176400) * 0.01)
    got = loss(a.to(dev), b.to(dev)).item()
    sub = loss(a[:4].to(dev), b[:4].to(dev)).item()
    ref = spo.mrstft_loss(a[:4], b[:4])[0].item()
    assert abs(sub - ref) <= 1e-3 * abs(ref)
    assert got > 0 and abs(got - sub) <= 0.2 * sub   # rows are i.i.d.: the full batch is close to the sub-batch
    same = loss(a.to(dev), a.to(dev)).item()
    assert abs(same) <= 1e-6
