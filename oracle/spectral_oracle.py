"""CPU oracle for the spectral losses (TEST INFRASTRUCTURE -- never imported by the product).

The reference has NO live spectral-loss code.  What exists:
  * the commented-out mel block /root/reference/conf/config.yaml:51-61 (n_fft 1024, win_length null,
    hop 512, center, reflect, power 2.0, slaney norm, onesided, 128 mels, htk scale),
  * the commented-out use /root/reference/audio_to_params.py:150-153
    (mel_l1_error = mean(|mel(audio) - mel(predicted_audio)|)),
  * two "auraloss" TODO comments (audio_to_params.py:233, evaluate_audio_representations.py:184).
torchaudio / auraloss are absent from this machine, so this file restates
torchaudio.transforms.MelSpectrogram (Spectrogram -> MelScale with melscale_fbanks) and
auraloss.freq.MultiResolutionSTFTLoss defaults from their published definitions on top of
``torch.stft``: **PARITY UNPINNED**.
"""
import math

import torch


def hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm="slaney"):
    """torchaudio.functional.melscale_fbanks, mel_scale="htk" -> fb [n_freqs, n_mels] fp32."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min, m_max = hz_to_mel_htk(f_min), hz_to_mel_htk(f_max)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down, up))
    if norm == "slaney":
        enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
        fb = fb * enorm.unsqueeze(0)
    return fb


def spectrogram(x, n_fft=1024, win_length=None, hop_length=512, power=2.0, center=True, pad_mode="reflect"):
    """torchaudio Spectrogram (hann window, normalized=False, onesided) -> [B, n_fft/2+1, frames]."""
    win_length = n_fft if win_length is None else win_length
    window = torch.hann_window(win_length).to(x.dtype)   # fp64 input: the gradient reference of the tests
    spec = torch.stft(x, n_fft, hop_length, win_length, window, center=center, pad_mode=pad_mode,
                      normalized=False, onesided=True, return_complex=True)
    mag = spec.abs()
    return mag if power == 1.0 else mag.pow(power)


def mel_spectrogram(x, sample_rate=44100, n_fft=1024, win_length=None, hop_length=512, n_mels=128,
                    power=2.0, f_min=0.0, f_max=None, norm="slaney"):
    """-> [B, n_mels, frames] (torchaudio MelSpectrogram semantics)."""
    spec = spectrogram(x, n_fft, win_length, hop_length, power)
    f_max = float(sample_rate // 2) if f_max is None else f_max
    fb = melscale_fbanks(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate, norm)
    return torch.matmul(spec.transpose(-1, -2), fb.to(spec.dtype)).transpose(-1, -2)


def mel_l1(audio, target_audio, **kw):
    """mean(|mel(audio) - mel(target)|)   (audio_to_params.py:150-153, commented block)."""
    return torch.mean(torch.abs(mel_spectrogram(audio, **kw) - mel_spectrogram(target_audio, **kw)))


def stft_l1(audio, target_audio, n_fft=1024, hop_length=512, win_length=None, power=1.0):
    """BASELINE config #1 "STFT L1 loss": mean |  |STFT(a)|^p - |STFT(b)|^p  |."""
    return torch.mean(torch.abs(spectrogram(audio, n_fft, win_length, hop_length, power)
                                - spectrogram(target_audio, n_fft, win_length, hop_length, power)))


def mrstft_loss(x, y, fft_sizes=(1024, 2048, 512), hop_sizes=(120, 240, 50), win_lengths=(600, 1200, 240),
                eps=1e-8):
    """auraloss MultiResolutionSTFTLoss defaults: per resolution spectral convergence
    ||Y|-|X||_F / ||Y||_F plus L1 of log magnitudes, magnitudes = sqrt(clamp(re^2+im^2, eps));
    averaged over resolutions.  x = prediction, y = target."""
    total = 0.0
    parts = []
    for n_fft, hop, win in zip(fft_sizes, hop_sizes, win_lengths):
        w = torch.hann_window(win).to(x.dtype)
        X = torch.stft(x, n_fft, hop, win, w, return_complex=True)
        Y = torch.stft(y, n_fft, hop, win, w, return_complex=True)
        xm = torch.sqrt(torch.clamp(X.real ** 2 + X.imag ** 2, min=eps))
        ym = torch.sqrt(torch.clamp(Y.real ** 2 + Y.imag ** 2, min=eps))
        sc = torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
        lm = torch.nn.functional.l1_loss(torch.log(xm), torch.log(ym))
        parts.append((sc, lm))
        total = total + sc + lm
    return total / len(fft_sizes), parts
