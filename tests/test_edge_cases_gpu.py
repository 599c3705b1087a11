"""Edge cases the domain offers, HIP vs oracle: ragged lengths that defeat the 16-byte paths, other
sample rates, extreme parameter vectors (zero durations, zero LFO weights -> NaN exactly where the oracle
has NaN), random shapes for PQMF / VICReg / STFT."""
import pytest
import torch

from oracle import pqmf_oracle as po
from oracle import spectral_oracle as spo
from oracle import synth_oracle as so
from oracle import vicreg_oracle as vo
from helpers import randn

pytestmark = pytest.mark.gpu


def _voice(dev, B, sr, sec):
    from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
    return Voice(SynthConfig(batch_size=B, sample_rate=sr, buffer_size_seconds=sec, reproducible=False)).to(dev)


@pytest.mark.parametrize("sr,sec", [(22050, 0.5), (48000, 1.0), (44100, 0.9999), (16000, 0.26)])
def test_voice_other_rates_and_ragged_lengths(lib, dev, sr, sec):
    """T not a multiple of 4 (scalar load/store paths), tiles that end mid-chunk, short control buffers."""
    B = 5
    v = _voice(dev, B, sr, sec)
    cfg = so.VoiceConfig(batch_size=B, sample_rate=sr, buffer_size_seconds=sec)
    assert v.synthconfig.buffer_size == cfg.buffer_size
    audio, params, _ = v(2)
    ref, parts = so.render_from_params01(cfg, params.cpu(), so.make_noise(cfg), "cr", True)
    assert torch.equal(v.control_signals()[0].cpu(), parts["ctrl"])
    assert (audio.cpu() - ref).abs().max().item() <= 1e-4
    assert v.chain_status() == 0


def test_voice_extreme_parameters(lib, dev):
    """All-ones, all-0.5, tiny and all-zero parameter rows: zero durations, saturated ramps, zero LFO mode
    weights (0/0 -> NaN).  NaNs must appear exactly where the oracle produces them."""
    v = _voice(dev, 4, 16000, 1.0)
    p = torch.stack([torch.full((78,), 1.0), torch.full((78,), 0.5), torch.full((78,), 1e-3), torch.zeros(78)])
    cfg = so.VoiceConfig(batch_size=4, sample_rate=16000, buffer_size_seconds=1.0)
    ref, parts = so.render_from_params01(cfg, p, so.make_noise(cfg), "cr", True)
    ctrl = v.control_signals(p.to(dev))[0].cpu()
    assert torch.equal(torch.isnan(ctrl), torch.isnan(parts["ctrl"]))
    ok = ~torch.isnan(parts["ctrl"])
    assert torch.equal(ctrl[ok], parts["ctrl"][ok])
    got = v.render(p.to(dev)).cpu()
    rows_ok = ~torch.isnan(ref).any(dim=1)
    assert rows_ok.sum() >= 3
    assert (got[rows_ok] - ref[rows_ok]).abs().max().item() <= 1e-4
    assert torch.isnan(got[~rows_ok]).any(dim=1).all() if (~rows_ok).any() else True


def test_pqmf_random_shapes(lib, dev):
    from inverse_audio_synthesis_amd.pqmf import PQMF
    g = torch.Generator().manual_seed(0)
    for _ in range(12):
        N = int(torch.randint(1, 9, (1,), generator=g))
        taps = 2 * int(torch.randint(2, 40, (1,), generator=g))
        T = int(torch.randint(taps + 1, 9000, (1,), generator=g))
        B = int(torch.randint(1, 5, (1,), generator=g))
        m = PQMF(N=N, taps=taps, cutoff=0.1, beta=8.0).to(dev)
        x = randn((B, 1, T), 1000 + T)
        ref = po.analysis(x, m.H.cpu(), N, taps)
        z = m(x.to(dev))
        assert z.shape == ref.shape
        assert (z.cpu() - ref).abs().max().item() <= 2e-5
        y = m.synthesis(z)
        yr = po.synthesis(ref, m.G.cpu(), m.updown_filter.cpu(), N, taps)
        assert y.shape == yr.shape
        assert (y.cpu() - yr).abs().max().item() <= 1e-4 * max(1.0, yr.abs().max().item())


def test_vicreg_random_shapes(lib, dev):
    from inverse_audio_synthesis_amd.vicreg import vicreg_loss
    g = torch.Generator().manual_seed(1)
    for _ in range(10):
        B = int(torch.randint(2, 300, (1,), generator=g))
        D = int(torch.randint(1, 700, (1,), generator=g))
        cfgB = int(torch.randint(2, 400, (1,), generator=g))
        x, y = randn((B, D), 7 * B) * 1.5 - 0.2, randn((B, D), 11 * D) * 0.6 + 0.4
        ref = [o.item() for o in vo.loss(x, y, cfgB, D)]
        out = [o.item() for o in vicreg_loss(x.to(dev), y.to(dev), cfgB)]
        assert abs(out[1] - ref[1]) <= 1e-5 * abs(ref[1])
        assert abs(out[2] - ref[2]) <= 1e-5 * max(abs(ref[2]), 1e-3)
        assert abs(out[3] - ref[3]) <= 2e-3 * max(abs(ref[3]), 1e-6), (B, D, cfgB, out[3], ref[3])


def test_stft_random_shapes(lib, dev):
    from inverse_audio_synthesis_amd.spectral import STFTPlan, VALUE_POWER
    g = torch.Generator().manual_seed(2)
    for _ in range(10):
        n_fft = [512, 1024, 2048][int(torch.randint(0, 3, (1,), generator=g))]
        win = int(torch.randint(n_fft // 4, n_fft + 1, (1,), generator=g))
        hop = int(torch.randint(1, n_fft + 300, (1,), generator=g))
        T = int(torch.randint(n_fft // 2 + 1, 12000, (1,), generator=g))
        x = randn((2, T), T + hop) * 0.4
        ref = spo.spectrogram(x, n_fft, win, hop, 2.0)
        out = STFTPlan(n_fft, win, hop).to(dev).values(x.to(dev), VALUE_POWER).transpose(1, 2).cpu()
        assert out.shape == ref.shape, (n_fft, win, hop, T)
        assert (out - ref).abs().max().item() <= 1e-4 * ref.abs().max().item(), (n_fft, win, hop, T)


@pytest.mark.parametrize("n_fft", [512, 1024, 2048])
def test_stft_single_frame_rows(lib, dev, n_fft):
    """n_fft / 2 < T < hop: ONE frame per row (F = 1).  The flat frame list [B F] is decoded with a multiply-high by
    floor(2^32 / F), which does not fit 32 bits at F = 1; with B >= 3 a wrapped constant sends frames 2.. to the wrong row."""
    from inverse_audio_synthesis_amd.spectral import STFTL1, STFTPlan, VALUE_POWER
    B, T, hop = 5, n_fft // 2 + 90, n_fft
    assert lib.ias_stft_num_frames(T, n_fft, hop) == 1
    x, y = randn((B, T), 5) * 0.4, randn((B, T), 6) * 0.4
    ref = spo.spectrogram(x, n_fft, n_fft, hop, 2.0)
    out = STFTPlan(n_fft, n_fft, hop).to(dev).values(x.to(dev), VALUE_POWER).transpose(1, 2).cpu()
    assert out.shape == ref.shape == (B, n_fft // 2 + 1, 1)
    assert (out - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    xg = x.to(dev).requires_grad_()
    loss = STFTL1(n_fft, hop).to(dev)(xg, y.to(dev))
    xr = x.clone().requires_grad_()
    lref = spo.stft_l1(xr, y, n_fft, hop)
    assert abs(loss.item() - lref.item()) <= 1e-3 * abs(lref.item())
    loss.backward(); lref.backward()
    assert (xg.grad.cpu() - xr.grad).abs().max().item() <= 2e-3 * xr.grad.abs().max().item()


def test_c_abi_rejects_bad_arguments(lib, dev):
    """Every entry point returns a negative IAS_ERR_* for null pointers / impossible dimensions instead of launching
    (the reference's behaviour at this boundary is a bare assert / exception)."""
    from inverse_audio_synthesis_amd import _lib
    z = torch.zeros(64, device=dev)
    zd = torch.zeros(64, device=dev, dtype=torch.float64)
    p = _lib.ptr
    ARG, UNSUP = -1, -2
    assert lib.ias_voice_backward(None, p(z), p(z), p(z), p(z), p(zd), p(zd), p(z), 1, 16000, 441, 16000, None) == ARG
    assert lib.ias_voice_backward(p(z), p(z), p(z), p(z), p(z), p(zd), p(zd), p(z), 0, 16000, 441, 16000, None) == ARG
    # a control rate too close to the sample rate: the transposed upsample cannot stage its intervals
    assert lib.ias_voice_backward(p(z), p(z), p(z), p(z), p(z), p(zd), p(zd), p(z), 1, 64, 60, 16000, None) == UNSUP
    assert lib.ias_voice_control_backward(None, p(z), p(zd), p(z), 1, 441, 441, None) == ARG
    assert lib.ias_voice_control_backward(p(z), p(z), p(zd), p(z), 1, 441, 440, None) == UNSUP     # kernel is built for 441
    assert lib.ias_voice_control_backward(p(z), p(z), p(zd), p(z), 1, 100000, 441, None) == UNSUP  # does not fit LDS
    assert lib.ias_stft_loss_backward(p(z), p(z), None, None, None, None, None, 0, p(z), None, None, p(z), p(z), 1, 4000, 1000,
                                      256, 501, 2, 1, 1.0, 0.0, None) == UNSUP   # n_fft not 512 / 1024 / 2048
    assert lib.ias_stft_loss_backward(p(z), p(z), None, None, None, None, None, 0, p(z), None, None, p(z), p(z), 1, 400, 1024,
                                      256, 513, 2, 1, 1.0, 0.0, None) == ARG     # T <= n_fft / 2: reflect padding impossible
    assert lib.ias_stft_loss_backward(p(z), p(z), None, None, None, None, None, 0, p(z), None, None, p(z), p(z), 1, 4000, 1024,
                                      256, 128, 2, 1, 1.0, 0.0, None) == ARG     # linear bins need n_out = n_fft / 2 + 1
    assert lib.ias_stft_loss_backward(p(z), p(z), None, None, None, None, None, 0, p(z), None, None, p(z), p(z), 1, 4000, 1024,
                                      256, 513, 1, 2, 1.0, 1e-8, None) == ARG    # MR-STFT mode needs its coefficients
    assert lib.ias_pqmf_pack_taps(p(z), p(z), 200, 63, None) == ARG                  # no packed layout for N > 64
    assert lib.ias_pqmf_packed_taps_len(200, 63) == 0 and lib.ias_pqmf_packed_taps_len(3, 63) > 0
    assert lib.ias_pqmf_analysis(p(z), p(z), None, None, p(z), None, None, None, 1, 64, 3, 62, None) == ARG   # even tap count
    torch.cuda.synchronize()
