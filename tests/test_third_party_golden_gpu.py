"""The stages whose arithmetic lives in packages the reference only names -- torchsynth (Voice), torchaudio (mel
spectrogram), auraloss (MR-STFT loss) -- are PARITY UNPINNED while those packages are absent from the build container
(DESIGN.md section 2).  scripts/make_golden.py writes their golden vectors the moment one of them imports there; these
tests pick the files up when they exist and are skipped, saying why, while they do not."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _load(name, package):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} absent: {package} does not import in the build container -- this stage is parity-unpinned")
    return np.load(path, allow_pickle=False)


def test_voice_against_torchsynth_golden(lib, dev):
    """north_star: rendered audio within 1e-4 relative of the reference on identical seeds / params."""
    g = _load("voice_torchsynth.npz", "torchsynth")
    from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
    for tag in ("small", "head"):
        B, sr, sec, idx = g[f"{tag}_cfg"]
        v = Voice(SynthConfig(batch_size=int(B), sample_rate=int(sr), buffer_size_seconds=float(sec), reproducible=False)).to(dev)
        audio, params, is_train = v(int(idx))
        assert np.allclose(params.cpu().numpy(), g[f"{tag}_params"], atol=0, rtol=0), "parameter sampling differs"
        assert np.array_equal(is_train.cpu().numpy(), g[f"{tag}_is_train"])
        a = audio.cpu()
        ref = torch.from_numpy(g[f"{tag}_audio"]) if f"{tag}_audio" in g else None
        if ref is not None:
            rel = ((a - ref).norm(dim=-1) / ref.norm(dim=-1).clamp_min(1e-30)).max().item()
        else:
            sub, ref_sub = a.flatten()[::97], torch.from_numpy(g[f"{tag}_audio_sub"])
            rel = ((sub - ref_sub).norm() / ref_sub.norm().clamp_min(1e-30)).item()
        assert rel <= 1e-4, (tag, rel)


def test_mel_against_torchaudio_golden(lib, dev):
    g = _load("mel_torchaudio.npz", "torchaudio")
    from inverse_audio_synthesis_amd.spectral import MelSpectrogram
    x = torch.randn(tuple(g["shape"]), generator=torch.Generator().manual_seed(int(g["seed"]))) * float(g["scale"])
    got = MelSpectrogram(sample_rate=44100).to(dev)(x.to(dev)).cpu()
    ref = torch.from_numpy(g["mel"])
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


def test_mrstft_against_auraloss_golden(lib, dev):
    """north_star: spectral loss within 1e-3."""
    g = _load("mrstft_auraloss.npz", "auraloss")
    from inverse_audio_synthesis_amd.spectral import MultiResolutionSTFTLoss
    s0, s1 = (int(v) for v in g["seeds"])
    shape = tuple(g["shape"])
    a = torch.randn(shape, generator=torch.Generator().manual_seed(s0)) * float(g["scale"])
    b = torch.randn(shape, generator=torch.Generator().manual_seed(s1)) * float(g["scale"])
    got = MultiResolutionSTFTLoss().to(dev)(a.squeeze(1).to(dev), b.squeeze(1).to(dev)).item()
    assert abs(got - float(g["loss"])) <= 1e-3 * abs(float(g["loss"]))
