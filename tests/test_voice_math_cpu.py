"""Host build of the device arithmetic (csrc/voice_math.h) against the oracle, no GPU.

The emulation library (tests/cpu_emul/voice_emul.cpp) is test infrastructure: it runs the same
per-sample inline functions the HIP kernels call, in plain sequential loops."""
import ctypes
import os
import subprocess

import pytest
import torch

from conftest import ROOT
from oracle import synth_oracle as so


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emul") / "libvoice_emul.so")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gen_voice_table
    gen_voice_table.main()
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I",
                           os.path.join(ROOT, "inverse-audio-synthesis_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpu_emul", "voice_emul.cpp"), "-o", out])
    return ctypes.CDLL(out)


def _run(emul, cfg, p01, noise):
    B, T, Tc = cfg.batch_size, cfg.buffer_size, cfg.control_buffer_size
    audio, ctrl, mixed = torch.empty(B, T), torch.empty(B, 5, Tc), torch.empty(B, T)
    fp = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = emul.emul_voice_render(fp(p01.contiguous()), fp(noise), fp(audio), fp(ctrl), fp(mixed), B, T, Tc,
                                cfg.sample_rate, cfg.control_rate)
    assert rc == 0
    return audio, ctrl, mixed


@pytest.mark.parametrize("ctl", [0, 1])
@pytest.mark.parametrize("B,sr,sec,seed", [(4, 16000, 1.0, 0), (6, 44100, 4.0, 3)])
def test_device_math_matches_cr_oracle(emul, B, sr, sec, seed, ctl):
    """ctl = 0: the libm calls that define the "cr" arithmetic; ctl = 1: the written-out fp64 pow / cos / fmod of
    csrc/voice_ctrl_math.h that the HIP control kernel runs (round 5).  The same bits either way."""
    cfg = so.VoiceConfig(batch_size=B, sample_rate=sr, buffer_size_seconds=sec)
    noise = so.make_noise(cfg)
    p01 = so.sample_params01(cfg, seed)
    ref, parts = so.render_from_params01(cfg, p01, noise, "cr", True)
    emul.emul_set_ctl(ctl)
    try:
        audio, ctrl, mixed = _run(emul, cfg, p01, noise)
    finally:
        emul.emul_set_ctl(0)
    assert torch.equal(ctrl, parts["ctrl"]), "control-rate signals must be bit-exact"
    assert (audio - ref).abs().max().item() <= 2e-6
    assert not torch.isnan(audio).any()


def test_extreme_parameters(emul):
    """All-zeros / all-ones / mid parameters: zero durations, zero LFO weights etc. stay finite or
    fail exactly as the oracle does."""
    cfg = so.VoiceConfig(batch_size=3, sample_rate=16000, buffer_size_seconds=1.0)
    noise = so.make_noise(cfg)
    p01 = torch.stack([torch.full((78,), 1.0), torch.full((78,), 0.5), torch.full((78,), 1e-3)])
    ref, parts = so.render_from_params01(cfg, p01, noise, "cr", True)
    audio, ctrl, _ = _run(emul, cfg, p01, noise)
    assert torch.equal(torch.isnan(ctrl), torch.isnan(parts["ctrl"]))
    ok = ~torch.isnan(ref)
    assert (audio[ok] - ref[ok]).abs().max().item() <= 2e-6


def test_cr_upsample_is_the_torch_op_bit_for_bit():
    """The one audio-rate op of the "cr" arithmetic that is NOT a single IEEE operation, the linear upsample, is pinned by
    the op it restates: nn.Upsample(mode="linear", align_corners=True) on torch's CPU (what torchsynth issues) evaluates
    fl(w0 a + fl(w1 b)) -- x0 * w0 + x1 * w1 contracted into one fma -- and the oracle's statement of it (fma32, an
    exactly rounded emulation) gives the same bits on every element, at the headline and at a ragged length.  (The
    three-rounding form of rounds 1-3 differed from the op in 24 % of the elements.)"""
    # "cr" DEFINES the upsample as fl(w0 a + fl(w1 b)).  That this is also what the local torch op computes is a property of
    # the compiler and ISA torch was built with (which product of `x0 * w0 + x1 * w1` gets fused, if any), not of the op:
    # probe the local op's rounding form first, assert bit-equality where it is the fused form (this image), and a
    # one-ulp bound on any other build.
    pcfg = so.VoiceConfig(batch_size=1, sample_rate=44100, buffer_size_seconds=4096 / so.VoiceConfig().control_rate)
    probe = torch.randn(1, 1, pcfg.control_buffer_size, generator=torch.Generator().manual_seed(99)) * 3.0
    local_is_fused_form = torch.equal(so._Math("torch").upsample(probe, pcfg), so._Math("cr").upsample(probe, pcfg))
    for B, sr, sec in ((3, 44100, 4.0), (2, 16000, 0.37)):
        cfg = so.VoiceConfig(batch_size=B, sample_rate=sr, buffer_size_seconds=sec)
        ctrl = torch.randn(B, 5, cfg.control_buffer_size, generator=torch.Generator().manual_seed(B)) * 3.0
        want = so._Math("torch").upsample(ctrl, cfg)
        got = so._Math("cr").upsample(ctrl, cfg)
        if local_is_fused_form:
            assert torch.equal(got, want)
        else:
            ulp = torch.maximum(want.abs(), got.abs()) * 2.0 ** -23 + 1e-45
            assert ((got - want).abs() <= ulp).all()
    # fma32 is a single rounding: against exact rational arithmetic, including cancellation and far-apart exponents
    from fractions import Fraction
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(4000, generator=g), torch.randn(4000, generator=g)
    c = torch.randn(4000, generator=g) * torch.tensor(2.0) ** torch.randint(-40, 40, (4000,), generator=g)
    c[:500] = -(a[:500] * b[:500])                       # near-total cancellation
    got = so.fma32(a, b, c)
    import numpy as np
    for i in range(4000):
        ex = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        f = np.float32(float(ex))
        cands = [np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))]
        best = min(cands, key=lambda t: (abs(Fraction(float(t)) - ex), int(np.float32(t).view(np.uint32)) & 1))
        assert np.float32(got[i].item()) == best, (i, float(a[i]), float(b[i]), float(c[i]))


def test_torch_vs_cr_math_spread_is_documented():
    """The two oracle math modes differ by the libm-to-libm spread (DESIGN.md section on parity):
    not bit-equal, but small in relative L2 for a typical batch."""
    cfg = so.VoiceConfig(batch_size=4, sample_rate=16000, buffer_size_seconds=1.0)
    noise = so.make_noise(cfg)
    p01 = so.sample_params01(cfg, 0)
    a = so.render_from_params01(cfg, p01, noise, "torch")
    b = so.render_from_params01(cfg, p01, noise, "cr")
    rel = ((a - b).norm() / b.norm()).item()
    assert rel < 5e-2


def test_torch_vs_cr_spread_at_full_length():
    """4 s @ 44.1 kHz, 16 voices: the spread between the reference op sequence evaluated with torch's fp32 CPU ops
    and with correctly rounded ops -- the bound the GPU parity test (tests/test_voice_gpu.py) asserts for the HIP
    render against "torch" is this spread, not 1e-4."""
    cfg = so.VoiceConfig(batch_size=16)
    noise = so.make_noise(cfg)
    p01 = so.sample_params01(cfg, 0)
    a = so.render_from_params01(cfg, p01, noise, "torch")
    b = so.render_from_params01(cfg, p01, noise, "cr")
    d = (a - b).double()
    rel = d.norm(dim=1) / b.double().norm(dim=1)
    assert rel.max().item() <= 1.5e-2 and rel.median().item() <= 3e-4
    assert d.abs().max().item() <= 1e-1
    assert (d.abs() > 1e-4).double().mean().item() <= 6e-2


def test_fast_formulations_are_bit_identical(emul):
    """The cheaper device formulations (fp64-reciprocal divisions, degree-10 exp2 polynomial) give the
    same fp32 bits as the specification ones, for the sample rates the configs use."""
    import ctypes as C
    emul.emul_check_fast_paths.restype = C.c_longlong
    n = 2_000_000
    for seed, sr in enumerate((16000, 22050, 44100, 48000, 96000)):
        pm = torch.rand(n, generator=torch.Generator().manual_seed(seed))
        bad = emul.emul_check_fast_paths(C.c_void_p(pm.data_ptr()), C.c_longlong(n), C.c_float(57.3 + seed),
                                         C.c_float(31.0 - 9 * seed), C.c_int(sr))
        assert bad == 0, f"{bad} mismatches at sample rate {sr}"


def test_headed_adsr_is_bit_identical(emul):
    """The env kernel caches the flat head of the decay / release ramps; same bits as the plain ADSR,
    including zero durations and notes shorter than the attack."""
    import ctypes as C
    emul.emul_check_adsr_headed.restype = C.c_longlong
    g = torch.Generator().manual_seed(3)
    n = 400
    p = torch.rand(n, 6, generator=g)
    p[:, 0] *= 2.0; p[:, 1] *= 2.0; p[:, 3] *= 5.0; p[:, 4] = 0.1 + 5.9 * p[:, 4]; p[:, 5] = 0.01 + 3.99 * p[:, 5]
    p[:20, 0] = 0.0          # zero attack
    p[20:40, 1] = 0.0        # zero decay
    p[40:60, 3] = 0.0        # zero release
    p[60:80, 5] = 0.01       # note shorter than the attack
    p = p.contiguous()
    bad = emul.emul_check_adsr_headed(C.c_void_p(p.data_ptr()), C.c_longlong(n), C.c_int(1764), C.c_int(441))
    assert bad == 0


@pytest.fixture(scope="module")
def ctl_check(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ctl") / "libctrl_math_check.so")
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import gen_ctrl_tables
    gen_ctrl_tables.main()
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-march=native", "-shared", "-fPIC", "-I",
                           os.path.join(ROOT, "inverse-audio-synthesis_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpu_emul", "ctrl_math_check.cpp"), "-o", out])
    lib = ctypes.CDLL(out)
    lib.ctrl_math_check.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_ulonglong, ctypes.POINTER(ctypes.c_double)]
    return lib


@pytest.mark.parametrize("kind,name,max_ulp64", [(0, "pow on ADSR-like ramps", 4.0), (1, "pow: any normal x < 1, a in [2^-6, 64]", 4.0),
                                                 (2, "cos", 4.0), (3, "fmod by fl32(2 pi)", 0.0),
                                                 (4, "log2 as one value, also next to 1", 4.0), (5, "log10 of a frequency", 4.0),
                                                 (6, "exp2 of an fp32", 4.0),
                                                 (7, "pow outside the render's domain: denormal x, x > 1, |a| up to 256", 1000.0)])
def test_written_out_control_math_equals_libm_after_the_rounding_to_fp32(ctl_check, kind, name, max_ulp64):
    """csrc/voice_ctrl_math.h (the fp64 pow / cos / fmod the HIP control pass evaluates instead of calling the device math
    library) against libm, the functions the "cr" contract is defined by, on 10^7 random arguments each: the value rounded
    to fp32 is IDENTICAL everywhere, the fp64 values agree to a few ulp (fmod: exactly)."""
    out = (ctypes.c_double * 4)()
    assert ctl_check.ctrl_math_check(kind, 10_000_000, 2024 + kind, out) == 0
    n, bad, worst, outside = list(out)
    assert bad == 0, (name, bad)
    assert worst <= max_ulp64, (name, worst)
    assert outside <= 0.02 * n, (name, outside)


@pytest.mark.parametrize("kind,name,max_ulp64", [(0, "pow and ln of a double in (0, 1]", 16.0), (1, "sin / cos up to 2^20 rad", 8.0),
                                                 (2, "x mod 2 pi", 8.0)])
def test_written_out_fp64_argument_forms_against_libm(ctl_check, kind, name, max_ulp64):
    """The fp64-ARGUMENT forms of csrc/voice_ctrl_math.h (ias_ctl_pow_d / log_d / sincos_d / mod_d: the control-rate
    backward, whose gradients are checked to 1e-5) against libm on 5 * 10^6 arguments each: within a few ulp of fp64."""
    ctl_check.ctrl_math_check_d.argtypes = [ctypes.c_int, ctypes.c_longlong, ctypes.c_ulonglong, ctypes.POINTER(ctypes.c_double)]
    out = (ctypes.c_double * 4)()
    assert ctl_check.ctrl_math_check_d(kind, 5_000_000, 4242 + kind, out) == 0
    n, bad, worst, _ = list(out)
    assert bad == 0 and worst <= max_ulp64, (name, bad, worst)
