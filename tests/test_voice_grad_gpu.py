"""Gradient of the HIP Voice render w.r.t. the 78 normalised parameters (csrc/voice_grad_kernels.hip + the torch
control graph of voice_grad.py) against torch.autograd through the oracle evaluated in fp64 (math "f64").

The reference has no backward of its own to compare with (audio_to_params.py:56-172 is commented
out), so the oracle's autograd IS the definition.  Tolerance: the HIP forward rounds the oscillator phases to fp32
as the reference does (the fp64 oracle does not), which perturbs each per-sample gradient term by ~1e-3 relative;
sums over the row average that down.  Asserted: relative L2 error per voice <= 2e-2 and <= 5e-3 over a batch of four
(1e-2 over the two-voice 4 s batch, whose norm is one voice's: its midi_f0 gradient of ~2e7 dominates)."""
import pytest
import torch

from oracle import synth_oracle as so
from helpers import rel_l2

pytestmark = pytest.mark.gpu


def _voice(dev, B, sr, sec):
    from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
    return Voice(SynthConfig(batch_size=B, sample_rate=sr, buffer_size_seconds=sec, reproducible=False)).to(dev)


def _oracle_grad(cfg, params01, noise, w, normalize=True):
    p = params01.double().requires_grad_(True)
    if normalize:
        a = so.render_from_params01(cfg, p, noise, "f64")
    else:
        a = so.render_from_params01(cfg, p, noise, "f64", return_parts=True)[1]["mixed"]
    (g,) = torch.autograd.grad((a * w.double()).sum(), p)
    return g


def test_control_graph_matches_hip_control_kernels(lib, dev):
    from inverse_audio_synthesis_amd.voice_grad import control_graph
    v = _voice(dev, 8, 44100, 4.0)
    for seed in (0, 2):
        p = so.sample_params01(so.VoiceConfig(8, 44100, 4.0), seed).to(dev)
        ctrl, vconst = v.control_signals(p)
        ctrl_t, scal_t = control_graph(p.double(), v.synthconfig)
        # the LFO saw / square shapes jump: a control point that sits on a jump may fall on either side in fp32
        # vs fp64, so a handful of isolated points differ by up to the jump height
        err = (ctrl_t.float() - ctrl).abs().flatten()
        assert (err > 2e-5).float().mean().item() <= 1e-3 and err.mean().item() <= 1e-5
        # IasVoiceConst: f0_1 depth_1 phi_1 f0_2 depth_2 phi_2 kpart shape gain lvl0 lvl1 lvl2
        got = vconst[:, :12].double()
        assert ((scal_t - got).abs() / (1.0 + got.abs())).max().item() <= 1e-5


# 8002 samples: not a multiple of 4 -> every tile of the backward takes the guarded scalar path instead of the LDS-DMA
# blocks; 16000 x 1 s: three whole tiles + a partial one; 44100 x 4 s: 43 + 1.  (At 11025 Hz the highest pitches alias and
# the fp32 phases of the product decorrelate from the fp64 oracle's: not a usable gradient reference.)
@pytest.mark.parametrize("B,sr,sec,seed,normalize", [(4, 16000, 1.0, 0, True), (4, 16000, 1.0, 2, True),
                                                     (4, 16000, 1.0, 1, False), (2, 44100, 4.0, 3, True),
                                                     (4, 16000, 0.500125, 4, True)])
def test_gradient_matches_oracle_autograd(lib, dev, B, sr, sec, seed, normalize):
    cfg = so.VoiceConfig(B, sr, sec)
    v = _voice(dev, B, sr, sec)
    p0 = so.sample_params01(cfg, seed)
    w = torch.randn((B, cfg.buffer_size), generator=torch.Generator().manual_seed(100 + seed))
    ref = _oracle_grad(cfg, p0, so.make_noise(cfg), w, normalize)
    p = p0.to(dev).requires_grad_(True)
    audio = v.render(p, normalize=normalize)
    assert audio.requires_grad
    (audio * w.to(dev)).sum().backward()
    g = p.grad.cpu().double()
    assert g.shape == (B, 78) and torch.isfinite(g).all()
    per_voice = [rel_l2(g[b], ref[b]) for b in range(B)]
    assert max(per_voice) <= 2e-2, per_voice
    assert rel_l2(g, ref) <= (5e-3 if B >= 4 else 1e-2)
    # parameters the audio does not depend on get exactly zero, the rest the right sign
    dead = ref.abs() == 0
    assert (g[dead] == 0).all()
    big = ref.abs() >= 1e-3 * ref.abs().max(dim=1, keepdim=True)[0]
    assert (torch.sign(g[big]) == torch.sign(ref[big])).all()


@pytest.mark.parametrize("control_rate,sr,sec", [(100, 16000, 1.0), (882, 44100, 1.0)])
def test_gradient_at_other_control_rates(lib, dev, control_rate, sr, sec):
    """the backward kernels take the control rate as an argument as well (interval sums of the transposed upsample,
    ramps of the control-rate backward)"""
    from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
    B = 4
    cfg = so.VoiceConfig(B, sr, sec, control_rate=control_rate)
    v = Voice(SynthConfig(batch_size=B, sample_rate=sr, buffer_size_seconds=sec, control_rate=control_rate,
                          reproducible=False)).to(dev)
    p0 = so.sample_params01(cfg, 9)
    w = torch.randn((B, cfg.buffer_size), generator=torch.Generator().manual_seed(31))
    ref = _oracle_grad(cfg, p0, so.make_noise(cfg), w, True)
    p = p0.to(dev).requires_grad_(True)
    (v.render(p) * w.to(dev)).sum().backward()
    g = p.grad.cpu().double()
    assert torch.isfinite(g).all()
    assert max(rel_l2(g[b], ref[b]) for b in range(B)) <= 2e-2
    assert rel_l2(g, ref) <= 5e-3


def test_lane_consecutive_backward_agrees_with_the_chunk_scan_kernels(lib, dev, monkeypatch):
    """The default audio-rate backward (16 consecutive samples per thread, LDS-DMA blocks, transposed upsample folded in)
    against the first form (IAS_VOICE_GRAD_V1=1, read at every call) on the same inputs: whole tiles, a partial tile, a
    row length that is not a multiple of 4 (guarded scalar accesses), with and without the normalisation adjoint.  The two
    differ in the sin / cos range reduction (fp32 split vs fp64), in fp32 vs fp64 partial sums inside a thread and in the
    order of the interval sums: agreement to 1e-5 of each output's largest element; the default form is bit-reproducible."""
    from inverse_audio_synthesis_amd import _lib
    from inverse_audio_synthesis_amd.voice import SynthConfig, Voice
    from inverse_audio_synthesis_amd.voice_grad import audio_rate_backward, normalisation_rows
    for B, sr, sec in ((3, 44100, 0.5), (2, 16000, 0.500125), (2, 44100, 4.0)):
        v = Voice(SynthConfig(batch_size=B, sample_rate=sr, buffer_size_seconds=sec, reproducible=False)).to(dev)
        v.randomize(21)
        audio = v.render()
        g = torch.randn(audio.shape, generator=torch.Generator().manual_seed(8)).to(dev)
        ctl = v.rendered_control()
        for rn in (None, normalisation_rows(g, audio, v.read_peaks())):
            monkeypatch.delenv("IAS_VOICE_GRAD_V1", raising=False)
            c_new, s_new = audio_rate_backward(v, v.params01, g, rn, ctl)
            c_again, s_again = audio_rate_backward(v, v.params01, g, rn, ctl)
            monkeypatch.setenv("IAS_VOICE_GRAD_V1", "1")     # honoured by the diagnostic library only
            with _lib.use_library(_lib.load_diag()):
                c_old, s_old = audio_rate_backward(v, v.params01, g, rn, ctl)
            monkeypatch.delenv("IAS_VOICE_GRAD_V1", raising=False)
            assert torch.equal(c_new, c_again) and torch.equal(s_new, s_again)
            assert torch.isfinite(c_new).all() and torch.isfinite(s_new).all()
            for row in range(5):
                err = (c_new[:, row] - c_old[:, row]).abs().max().item()
                assert err <= 1e-5 * c_old[:, row].abs().max().item() + 1e-12, (B, sr, sec, row, err)
            for k in range(s_old.shape[1]):
                err = (s_new[:, k] - s_old[:, k]).abs().max().item()
                assert err <= 1e-5 * s_old[:, k].abs().max().item() + 1e-12, (B, sr, sec, k, err)


def test_gradient_is_deterministic_and_leaves_forward_untouched(lib, dev):
    v = _voice(dev, 4, 16000, 1.0)
    p0 = so.sample_params01(so.VoiceConfig(4, 16000, 1.0), 5).to(dev)
    plain = v.render(p0)
    grads = []
    for _ in range(2):
        p = p0.clone().requires_grad_(True)
        a = v.render(p)
        assert torch.equal(a.detach(), plain)
        a.square().mean().backward()
        grads.append(p.grad.clone())
    assert torch.equal(grads[0], grads[1])
    with torch.no_grad():
        assert not v.render(p0.clone().requires_grad_(True)).requires_grad


def test_gradient_descent_on_audio_loss_reduces_it(lib, dev):
    """A few steps of plain gradient descent on the mixer levels and VCO amplitudes' sources of a rendered target:
    the loss must go down (end-to-end sign / scale sanity of the whole chain)."""
    from inverse_audio_synthesis_amd import voice_spec as S
    cfg = so.VoiceConfig(4, 16000, 1.0)
    v = _voice(dev, 4, 16000, 1.0)
    target_p = so.sample_params01(cfg, 7).to(dev)
    target = v.render(target_p, normalize=False)
    cols = [S.INDEX[("mixer", n)] for n in ("vco_1", "vco_2", "noise")]
    p = target_p.clone()
    p[:, cols] = 0.5
    losses = []
    for _ in range(25):
        q = p.clone().requires_grad_(True)
        loss = (v.render(q, normalize=False) - target).square().mean()
        loss.backward()
        losses.append(loss.item())
        step = torch.zeros_like(p)
        step[:, cols] = q.grad[:, cols]
        p = (p - 0.5 * step / step.abs().max().clamp_min(1e-12) * 0.05).clamp(0.01, 0.99)
    assert losses[-1] < 0.5 * losses[0], losses


def test_full_batch_gradient_rows_match_small_batch(lib, dev):
    """Headline size (B=128, 4 s @ 44.1 kHz): finite gradients, and voices are independent -- the first 8 rows equal
    the gradient of a batch of those 8 voices alone (same kernels, different grid), bit for bit."""
    B, T = 128, 176400
    v, v8 = _voice(dev, B, 44100, 4.0), _voice(dev, 8, 44100, 4.0)
    p0 = torch.rand(B, 78, generator=torch.Generator().manual_seed(1000)).to(dev)
    w = torch.randn((B, T), generator=torch.Generator().manual_seed(5)).to(dev)
    p = p0.clone().requires_grad_(True)
    (v.render(p) * w).sum().backward()
    assert torch.isfinite(p.grad).all()
    q = p0[:8].clone().requires_grad_(True)
    (v8.render(q) * w[:8]).sum().backward()
    assert torch.equal(p.grad[:8], q.grad)


@pytest.mark.parametrize("B,sr,sec", [(8, 44100, 4.0), (5, 16000, 1.0)])
def test_hip_control_backward_matches_torch_graph(lib, dev, B, sr, sec):
    """csrc/voice_ctrl_grad_kernels.hip against torch.autograd through voice_grad.control_graph (its definition) in
    fp64, for random upstream gradients: <= 1e-5 relative per voice (fp32 output and envelope-gradient storage)."""
    from inverse_audio_synthesis_amd import voice_grad as vg
    from inverse_audio_synthesis_amd.voice import SynthConfig
    cfg = SynthConfig(batch_size=B, sample_rate=sr, buffer_size_seconds=sec, reproducible=False)
    Tc = cfg.control_buffer_size
    for seed in (0, 1, 2):
        p = so.sample_params01(so.VoiceConfig(B, sr, sec), seed).to(dev)
        if seed == 2:            # short notes, zero-length segments, params at the ends of their ranges
            p[0, 1] = 1e-3; p[1, 2] = 0.0; p[2, 3] = 0.0; p[3, :] = 0.999; p[4 % B, :] = 1e-3
        g = torch.Generator().manual_seed(10 + seed)
        g_ctrl = torch.randn((B, 5, Tc), generator=g).to(dev)
        g_scal = torch.randn((B, 12), generator=g, dtype=torch.float64).to(dev)
        ref = vg._control_backward_eager(cfg, p, g_ctrl, g_scal)
        got = vg._control_backward_hip(cfg, p, g_ctrl, g_scal)
        assert got is not None and torch.isfinite(got).all()
        for b in range(B):
            assert rel_l2(got[b].cpu(), ref[b].cpu()) <= 1e-5, (seed, b, rel_l2(got[b].cpu(), ref[b].cpu()))


def test_gradient_is_finite_at_the_ends_of_the_parameter_ranges(lib, dev):
    """Parameters at (almost) 0, 1 and mid-range: zero-length envelope segments, clamped pitches, unit ramps -- the
    places where plain autograd produces 0 * inf.  The HIP gradient stays finite and matches the fp64 oracle's
    (which uses the same 'gradient 0 at the kink' conventions) where that one is finite too."""
    cfg = so.VoiceConfig(3, 16000, 1.0)
    v = _voice(dev, 3, 16000, 1.0)
    p0 = torch.stack([torch.full((78,), 0.999), torch.full((78,), 0.5), torch.full((78,), 1e-3)])
    w = torch.randn((3, cfg.buffer_size), generator=torch.Generator().manual_seed(3))
    p = p0.to(dev).requires_grad_(True)
    (v.render(p) * w.to(dev)).sum().backward()
    g = p.grad.cpu().double()
    assert torch.isfinite(g).all()
    ref = _oracle_grad(cfg, p0, so.make_noise(cfg), w, True)
    ok = torch.isfinite(ref)
    for b in range(3):
        if ok[b].all() and ref[b].norm() > 0:
            assert rel_l2(g[b], ref[b]) <= 5e-2, (b, rel_l2(g[b], ref[b]))


def test_control_backward_in_four_launches_equals_the_single_kernel(lib, dev):
    """ias_voice_control_backward_ws (envelope values and envelope gradients on 6 x B workgroups around the per-voice
    kernel) against ias_voice_control_backward (everything in one workgroup per voice): same arithmetic per control point,
    same summation orders -> the same bits."""
    from inverse_audio_synthesis_amd import _lib
    B, Tc = 5, 441
    g = torch.Generator().manual_seed(3)
    p = torch.rand(B, 78, generator=g).to(dev)
    p[0, :] = 0.0
    p[1, :] = 1.0
    g_ctrl = torch.randn(B, 5, Tc, generator=g).to(dev)
    g_scal = torch.randn(B, 12, generator=g, dtype=torch.float64).to(dev)
    one, four = torch.empty(B, 78, device=dev), torch.empty(B, 78, device=dev)
    _lib.check(lib.ias_voice_control_backward(_lib.ptr(p), _lib.ptr(g_ctrl), _lib.ptr(g_scal), _lib.ptr(one), B, Tc, 441,
                                              _lib.stream()), "ias_voice_control_backward")
    ws = torch.empty(int(lib.ias_voice_control_backward_ws_bytes(B, Tc)), dtype=torch.uint8, device=dev)
    _lib.check(lib.ias_voice_control_backward_ws(_lib.ptr(p), _lib.ptr(g_ctrl), _lib.ptr(g_scal), _lib.ptr(four), _lib.ptr(ws),
                                                 ws.numel(), B, Tc, 441, _lib.stream()), "ias_voice_control_backward_ws")
    assert torch.isfinite(one[2:]).all()                 # rows 0/1 sit on the range ends, where the curve maps' derivatives
    assert torch.allclose(one, four, rtol=0.0, atol=0.0, equal_nan=True)      # are not finite (in torch autograd neither)
    small = torch.empty(16, dtype=torch.uint8, device=dev)
    assert lib.ias_voice_control_backward_ws(_lib.ptr(p), _lib.ptr(g_ctrl), _lib.ptr(g_scal), _lib.ptr(four), _lib.ptr(small),
                                             small.numel(), B, Tc, 441, _lib.stream()) == -4      # IAS_ERR_WORKSPACE


@pytest.mark.parametrize("B,sr,sec", [(4, 16000, 1.0), (3, 44100, 0.5)])
def test_backward_prelude_on_a_side_stream_gives_the_same_bits(lib, dev, monkeypatch, B, sr, sec):
    """The cotangent-free parts of the backward (phase increments, envelope values) start at render time on a stream of
    their own (voice_grad.BackwardPrelude); IAS_VOICE_PRELUDE=0 keeps them in the backward.  Same gradient, bit for bit,
    also through a retained graph differentiated twice and inside a captured hipGraph."""
    v = _voice(dev, B, sr, sec)
    p0 = so.sample_params01(so.VoiceConfig(B, sr, sec), 9).to(dev)
    w = torch.randn(B, v.synthconfig.buffer_size, generator=torch.Generator().manual_seed(3)).to(dev)

    def grad(twice=False):
        p = p0.clone().requires_grad_(True)
        loss = (v.render(p) * w).sum()
        if twice:
            g1 = torch.autograd.grad(loss, p, retain_graph=True)[0]
            g2 = torch.autograd.grad(loss, p)[0]
            assert torch.equal(g1, g2)
            return g1
        return torch.autograd.grad(loss, p)[0]

    monkeypatch.setenv("IAS_VOICE_PRELUDE", "0")
    ref = grad()
    monkeypatch.setenv("IAS_VOICE_PRELUDE", "1")
    assert torch.equal(grad(), ref)
    assert torch.equal(grad(twice=True), ref)
    torch.cuda.synchronize()
    out = torch.zeros_like(ref)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            grad()
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out.copy_(grad())
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
