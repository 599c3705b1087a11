"""CPU oracle for the Voice render (TEST INFRASTRUCTURE -- never imported by the product).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module, and only as the checker / the timed CPU baseline.

What it restates: ``torchsynth.synth.Voice.output()`` as the reference drives it
(/root/reference/vicreg_audio_params.py:86-94,114; audio_to_params.py:215,240-257;
pretrain.py:75).  torchsynth itself is NOT on this machine
(/root/reference/requirements.txt:1, unpinned, un-vendored) and the reference holds
no test or golden vector at that boundary, so this is a restatement of
torchsynth 1.0.x's published algorithm written from its documentation:
**PARITY UNPINNED** for this file.

Numerics.  Every step below is issued as the torch CPU op torchsynth would issue
(fp32 tensors, ``torch.cumsum`` accumulating in double on CPU, ``nn.Upsample``
linear/align_corners).  Two math modes:

* ``math="torch"``  transcendental ops (pow/exp2/log2/cos at control rate, exp2 at
  audio rate) are the fp32 torch CPU ops (SLEEF, <=1 ulp, CPU-ISA dependent).
* ``math="cr"``     a fully specified, host-independent arithmetic: transcendental
  ops are evaluated in float64 and rounded once to fp32 ("correctly rounded"),
  the small matmuls / sums that feed the oscillator phase (LFO shape mix, LFO mode
  normalisation, mod matrix) are float64-accumulated dot products rounded once
  (torch's fp32 BLAS / vectorised reductions round differently from host to host),
  the linear upsample is written out as fl(w0*a + fl(w1*b)) -- one fused multiply-add
  on the rounded second product, which is what torch's CPU kernel of nn.Upsample
  computes (bit-equal to the op: tests/test_voice_math_cpu.py) -- and the final mixer
  as a left-to-right mul/add chain.  This is the bit-reproducible definition the
  HIP kernels implement; the difference between the two modes is the irreducible
  libm-to-libm / BLAS-to-BLAS spread of the reference itself (quantified in tests
  and DESIGN.md).
"""
import math

import torch

from . import synth_spec as S

TWO_PI = 2 * math.pi


class VoiceConfig:
    def __init__(self, batch_size=128, sample_rate=44100, buffer_size_seconds=4.0,
                 control_rate=S.CONTROL_RATE, reproducible=False):
        self.batch_size = int(batch_size)
        self.sample_rate = int(sample_rate)
        self.buffer_size_seconds = float(buffer_size_seconds)
        self.control_rate = int(control_rate)
        self.reproducible = bool(reproducible)
        self.buffer_size = int(self.buffer_size_seconds * self.sample_rate)
        self.control_buffer_size = int(self.buffer_size_seconds * self.control_rate)


# --------------------------------------------------------------------------- math


class _Math:
    def __init__(self, mode):
        # "f64": the "torch" op sequence evaluated in double (the gradient reference of tests/test_voice_grad_gpu.py)
        assert mode in ("torch", "cr", "f64")
        self.cr = mode == "cr"
        self.dtype = torch.float64 if mode == "f64" else torch.float32

    def _u(self, fn, *xs):
        if self.cr:
            return fn(*[x.double() for x in xs]).float()
        return fn(*xs)

    def pow(self, x, a):
        if self.dtype == torch.float64:
            # gradient reference: d/dx x^a at x = 0 is taken as 0 (autograd gives inf * 0 = nan for a < 1)
            return torch.pow(x.clamp_min(1e-300), a)
        return self._u(torch.pow, x, a)

    def exp2(self, x):
        return self._u(torch.exp2, x)

    def log2(self, x):
        return self._u(torch.log2, x)

    def log10(self, x):
        return self._u(torch.log10, x)

    def cos(self, x):
        return self._u(torch.cos, x)

    def matmul(self, a, b):
        """Small dot products on the phase path: fp64 accumulate, round once (cr)."""
        return self._u(torch.matmul, a, b)

    def sum1(self, x):
        if self.cr:
            return torch.sum(x.double(), dim=1, keepdim=True).float()
        return torch.sum(x, dim=1, keepdim=True)

    def cumsum1(self, x):
        if self.cr:
            return torch.cumsum(x.double(), dim=1).float()
        return torch.cumsum(x, dim=1)  # torch CPU accumulates fp32 cumsum in double as well

    def upsample(self, ctrl, cfg):
        """nn.Upsample(size=T, mode="linear", align_corners=True) on [B,C,Tc]."""
        if not self.cr:
            return torch.nn.Upsample(size=cfg.buffer_size, mode="linear", align_corners=True)(ctrl)
        Tc, T = ctrl.shape[-1], cfg.buffer_size
        scale = torch.tensor(float(Tc - 1), dtype=torch.float32) / torch.tensor(float(T - 1), dtype=torch.float32)
        real = scale * torch.arange(T, dtype=torch.float32)
        i0 = torch.clamp(torch.floor(real).long(), max=Tc - 1)
        i1 = i0 + (i0 < Tc - 1).long()
        w1 = torch.clamp(real - i0.float(), 0.0, 1.0)
        w0 = 1.0 - w1
        return fma32(w0, ctrl[..., i0], w1 * ctrl[..., i1])


def fma32(a, b, c):
    """fl32(a * b + c) for fp32 tensors with ONE rounding (a hardware fma), emulated in float64: the product of two
    floats is exact in double; its sum with c is taken with round-to-odd (the double sum, nudged onto the odd neighbour
    when it was inexact), which makes the final rounding to fp32 the rounding of the exact value (no double rounding)."""
    p = a.double() * b.double()
    c = c.double().expand_as(p)
    s = p + c
    bb = s - p                                  # TwoSum: s + err = p + c exactly
    err = (p - (s - bb)) + (c - bb)
    inexact = (err != 0) & torch.isfinite(s)
    bits = s.contiguous().view(torch.int64)
    even = (bits & 1) == 0
    # move an even-mantissa inexact sum one ulp toward the exact value: for a positive s the next double up is bits + 1
    toward_up = (err > 0) == (s > 0)            # away from zero in magnitude
    step = torch.where(toward_up, torch.ones_like(bits), -torch.ones_like(bits))
    bits = torch.where(inexact & even, bits + step, bits)
    return bits.view(torch.float64).float()


# --------------------------------------------------------------------- parameters


def from_0to1(u, lo, hi, curve, symmetric, m):
    """torchsynth ModuleParameterRange.from_0to1 (skewed [0,1] -> [lo,hi])."""
    if not symmetric:
        if curve != 1.0:
            u = m.exp2(m.log2(u) / curve)
        return lo + (hi - lo) * u
    dist = 2.0 * u - 1.0
    if curve != 1.0:
        u = torch.where(dist == 0.0, dist, m.exp2(m.log2(torch.abs(dist)) / curve) * torch.sign(dist))
    else:
        u = dist
    return lo + (hi - lo) / 2.0 * (u + 1.0)


def sample_params01(cfg, batch_idx):
    """Voice.randomize(seed=batch_idx): [B,78] uniforms in registration order.

    torchsynth draws the values in sub-batches of 32 voices (seed = global
    sub-batch number) and assigns them to the parameters in *sorted name* order.
    A batch that is not a multiple of 32 (only possible with reproducible=False)
    is drawn as one block seeded with batch_idx.
    """
    names = [f"{m}.{n}" for (m, n, *_r) in S.PARAMS]
    order = sorted(range(S.NPARAMS), key=lambda i: names[i])
    B = cfg.batch_size
    g = torch.Generator(device="cpu")
    sub = S.REPRODUCIBLE_SUBBATCH
    if B % sub == 0:
        blocks = []
        for i in range(B // sub):
            g.manual_seed(int(batch_idx) * (B // sub) + i)
            blocks.append(torch.rand((sub, S.NPARAMS), generator=g))
        drawn = torch.cat(blocks, 0)
    else:
        g.manual_seed(int(batch_idx))
        drawn = torch.rand((B, S.NPARAMS), generator=g)
    out = torch.empty(B, S.NPARAMS)
    for col, i in enumerate(order):
        out[:, i] = drawn[:, col]
    return out


def is_train(cfg, batch_idx):
    B = cfg.batch_size
    idx = torch.arange(B * int(batch_idx), B * (int(batch_idx) + 1))
    return (idx // S.REPRODUCIBLE_SUBBATCH) % 10 != 9


def make_noise(cfg):
    """Noise(seed=13): a fixed [B,T] uniform(-1,1) buffer from a CPU MT19937 stream."""
    g = torch.Generator(device="cpu").manual_seed(S.NOISE_SEED)
    return torch.rand((cfg.batch_size, cfg.buffer_size), generator=g) * 2.0 - 1.0


# ------------------------------------------------------------------ control rate


class _P:
    """Range-mapped parameter lookup p(module, name) -> [B] fp32."""

    def __init__(self, params01, m):
        self.v = {}
        for i, (mod, name, lo, hi, curve, sym) in enumerate(S.PARAMS):
            self.v[(mod, name)] = from_0to1(params01[:, i].to(m.dtype), lo, hi, curve, sym, m)

    def __call__(self, mod, name):
        return self.v[(mod, name)]


def _ramp(cfg, m, duration, alpha, start=None, inverse=False):
    dur = (duration * cfg.control_rate).unsqueeze(1)
    rng = torch.arange(cfg.control_buffer_size, dtype=torch.float32)
    ramp = rng.expand(duration.shape[0], -1)
    if start is not None:
        ramp = ramp - (start * cfg.control_rate).unsqueeze(1)
    ramp = torch.maximum(ramp, torch.tensor(0.0))
    # dur == 0 gives +inf -> 1 after the minimum; written with a guarded divisor (same values) so that autograd
    # does not produce 0 * inf = nan for the duration's gradient
    safe = torch.where(dur > 0.0, dur, torch.ones_like(dur))
    ramp = torch.where(dur > 0.0, (ramp + S.EPS) / safe + S.EPS, torch.full_like(ramp, float("inf")) + 0.0 * dur)
    ramp = torch.minimum(ramp, torch.tensor(1.0, dtype=ramp.dtype))
    if inverse:
        ramp = torch.where(dur > 0.0, 1.0 - ramp, ramp)
    return m.pow(ramp, alpha)


def adsr(cfg, m, p, mod, note_on):
    attack, decay, sustain = p(mod, "attack"), p(mod, "decay"), p(mod, "sustain")
    release, alpha = p(mod, "release"), p(mod, "alpha").unsqueeze(1)
    new_attack = torch.minimum(attack, note_on)
    new_decay = torch.maximum(note_on - attack, torch.tensor(0.0))
    new_decay = torch.minimum(new_decay, decay)
    a = _ramp(cfg, m, new_attack, alpha)
    sus = sustain.unsqueeze(1)
    d = (1.0 - sus) * _ramp(cfg, m, new_decay, alpha, start=new_attack, inverse=True) + sus
    r = _ramp(cfg, m, release, alpha, start=note_on, inverse=True)
    return a * d * r


def lfo(cfg, m, p, mod, rate_env, return_phase=False):
    freq = p(mod, "frequency").unsqueeze(1)
    freq = torch.maximum(freq + p(mod, "mod_depth").unsqueeze(1) * rate_env, torch.tensor(0.0))
    arg = m.cumsum1(TWO_PI * freq / cfg.control_rate)
    arg = arg + p(mod, "initial_phase").unsqueeze(1)
    cos = m.cos(arg + math.pi)
    square = torch.sign(cos)
    cos = (cos + 1.0) / 2.0
    square = (square + 1.0) / 2.0
    saw = torch.remainder(arg, TWO_PI) / TWO_PI
    revsaw = 1.0 - saw
    tri = 2 * saw
    tri = torch.where(tri > 1.0, 2.0 - tri, tri)
    shapes = torch.stack([cos, tri, saw, revsaw, square], dim=1)
    mode = torch.stack([p(mod, s) for s in S.LFO_SHAPES], dim=1)
    mode = m.pow(mode, torch.tensor(S.LFO_EXPONENT))
    mode = mode / m.sum1(mode)
    out = m.matmul(mode.unsqueeze(1), shapes).squeeze(1)
    return (out, arg) if return_phase else out


def control_signals(cfg, params01, math_mode="torch", return_debug=False):
    """-> (ctrl [B,5,Tc] fp32 mod-matrix outputs, p) ; order = S.MOD_OUTPUTS.
    return_debug adds [B,10,Tc]: adsr_1, adsr_2, lfo_1_amp, lfo_2_amp, lfo_1_rate, lfo_2_rate,
    lfo_1 phase, lfo_2 phase, lfo_1 out, lfo_2 out."""
    m = _Math(math_mode)
    p = _P(params01, m)
    note_on = p("keyboard", "duration")
    lfo_1_rate = adsr(cfg, m, p, "lfo_1_rate_adsr", note_on)
    lfo_2_rate = adsr(cfg, m, p, "lfo_2_rate_adsr", note_on)
    lfo_1_amp = adsr(cfg, m, p, "lfo_1_amp_adsr", note_on)
    lfo_2_amp = adsr(cfg, m, p, "lfo_2_amp_adsr", note_on)
    lfo_1, ph_1 = lfo(cfg, m, p, "lfo_1", lfo_1_rate, True)
    lfo_2, ph_2 = lfo(cfg, m, p, "lfo_2", lfo_2_rate, True)
    lfo_1 = lfo_1 * lfo_1_amp
    lfo_2 = lfo_2 * lfo_2_amp
    adsr_1 = adsr(cfg, m, p, "adsr_1", note_on)
    adsr_2 = adsr(cfg, m, p, "adsr_2", note_on)
    w = torch.stack([p("mod_matrix", f"{i}->{o}") for i in S.MOD_INPUTS for o in S.MOD_OUTPUTS], dim=1)
    w = w.reshape(-1, len(S.MOD_INPUTS), len(S.MOD_OUTPUTS)).swapaxes(1, 2)
    mod = torch.stack([adsr_1, adsr_2, lfo_1, lfo_2], dim=1)
    if return_debug:
        dbg = torch.stack([adsr_1, adsr_2, lfo_1_amp, lfo_2_amp, lfo_1_rate, lfo_2_rate, ph_1, ph_2, lfo_1, lfo_2], 1)
        return m.matmul(w, mod), p, dbg
    return m.matmul(w, mod), p


# -------------------------------------------------------------------- audio rate


def midi_to_hz(midi, m):
    return 440.0 * m.exp2((midi - 69.0) / 12.0)


def vco_phase(cfg, m, p, mod, midi_f0, pitch_mod):
    f0 = (midi_f0 + p(mod, "tuning")).unsqueeze(1)
    control = torch.clamp(f0 + p(mod, "mod_depth").unsqueeze(1) * pitch_mod, 0.0, 127.0)
    hz = midi_to_hz(control, m)
    arg = m.cumsum1(TWO_PI * hz / cfg.sample_rate)
    return arg + p(mod, "initial_phase").unsqueeze(1)


def render_from_params01(cfg, params01, noise, math_mode="torch", return_parts=False):
    """Voice.output() for explicit normalised parameters -> audio [B,T] fp32."""
    ctrl, p = control_signals(cfg, params01, math_mode)
    m = _Math(math_mode)
    upc = m.upsample(ctrl, cfg)  # [B,5,T]
    midi_f0 = p("keyboard", "midi_f0")

    arg1 = vco_phase(cfg, m, p, "vco_1", midi_f0, upc[:, 0])
    vco_1 = torch.cos(arg1) * upc[:, 1]

    arg2 = vco_phase(cfg, m, p, "vco_2", midi_f0, upc[:, 2])
    max_pitch = midi_f0 + torch.maximum(p("vco_2", "mod_depth"), torch.tensor(0.0))
    max_f0 = midi_to_hz(max_pitch, m)
    partials = (12000.0 / (max_f0 * m.log10(max_f0))).unsqueeze(1)
    square = torch.tanh(math.pi * partials * torch.sin(arg2) / 2)
    shape = p("vco_2", "shape").unsqueeze(1)
    vco_2 = ((1 - shape / 2) * square * (1 + shape * torch.cos(arg2))) * upc[:, 3]

    noise_out = noise * upc[:, 4]

    lv = torch.stack([p("mixer", "vco_1"), p("mixer", "vco_2"), p("mixer", "noise")], dim=1)
    if m.cr:
        mixed = lv[:, 0:1] * vco_1 + lv[:, 1:2] * vco_2 + lv[:, 2:3] * noise_out
    else:
        sig = torch.stack([vco_1, vco_2, noise_out], dim=1)
        mixed = torch.matmul(lv.unsqueeze(1), sig).squeeze(1)
    peak = torch.max(torch.abs(mixed), dim=1, keepdim=True)[0]
    audio = torch.where(peak > 1.0, mixed / peak, mixed)
    if return_parts:
        return audio, dict(ctrl=ctrl, arg1=arg1, arg2=arg2, mixed=mixed, peak=peak)
    return audio


class OracleVoice:
    """Minimal look-alike of ``Voice(synthconfig)``: voice(batch_idx) -> (audio, params, is_train)."""

    def __init__(self, cfg, math_mode="torch"):
        self.cfg = cfg
        self.math_mode = math_mode
        self.noise = make_noise(cfg)
        self.params01 = torch.full((cfg.batch_size, S.NPARAMS), 0.5)

    def __call__(self, batch_idx=None):
        if batch_idx is not None:
            self.params01 = sample_params01(self.cfg, batch_idx)
            train = is_train(self.cfg, batch_idx)
        else:
            train = torch.ones(self.cfg.batch_size, dtype=torch.bool)
        audio = render_from_params01(self.cfg, self.params01, self.noise, self.math_mode)
        return audio, self.params01.clone(), train
