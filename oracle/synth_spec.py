"""Voice parameter table shared by the oracle and the host code (data, no arithmetic).

TEST INFRASTRUCTURE + SPEC DATA.  This file only lists names and ranges; the
arithmetic lives in ``oracle/synth_oracle.py`` (checker) and in the HIP kernels
(product).  The host package keeps its *own* copy of this table
(``inverse-audio-synthesis_amd/voice_spec.py``) so that the product never
imports ``oracle/``; a CPU test asserts the two tables are identical.

Provenance: the reference calls ``torchsynth.synth.Voice``
(/root/reference/vicreg_audio_params.py:86-94,114; audio_to_params.py:196-203,215,240-257).
torchsynth is an un-pinned, un-vendored third-party dependency
(/root/reference/requirements.txt:1) that is absent from this machine, so the
table below is a restatement of torchsynth 1.0.x's published ``Voice`` layout
(module order, parameter names, default ranges) -- **parity unpinned**.  The one
in-repo cross-check is ``nparams: 78`` (/root/reference/conf/config.yaml:27):
2 + 6*5 + 2*8 + 20 + 3 + 4 + 3 = 78.
"""
import math

# (minimum, maximum, curve, symmetric)
_ADSR = [
    ("attack", 0.0, 2.0, 0.5, False),
    ("decay", 0.0, 2.0, 0.5, False),
    ("sustain", 0.0, 1.0, 1.0, False),
    ("release", 0.0, 5.0, 0.5, False),
    ("alpha", 0.1, 6.0, 1.0, False),
]
LFO_SHAPES = ["sin", "tri", "saw", "rsaw", "sqr"]
_LFO = [
    ("frequency", 0.0, 20.0, 0.25, False),
    ("mod_depth", -10.0, 20.0, 0.5, True),
    ("initial_phase", -math.pi, math.pi, 1.0, False),
] + [(s, 0.0, 1.0, 1.0, False) for s in LFO_SHAPES]
_VCO = [
    ("tuning", -24.0, 24.0, 1.0, False),
    ("mod_depth", -96.0, 96.0, 0.2, True),
    ("initial_phase", -math.pi, math.pi, 1.0, False),
]
MOD_INPUTS = ["adsr_1", "adsr_2", "lfo_1", "lfo_2"]
MOD_OUTPUTS = ["vco_1_pitch", "vco_1_amp", "vco_2_pitch", "vco_2_amp", "noise_amp"]
_MODMATRIX = [
    (f"{i}->{o}", 0.0, 1.0, 0.5, False) for i in MOD_INPUTS for o in MOD_OUTPUTS
]
_MIXER = [
    ("vco_1", 0.0, 1.0, 1.0, False),
    ("vco_2", 0.0, 1.0, 1.0, False),
    ("noise", 0.0, 1.0, 0.025, False),
]

# Module registration order of Voice (== column order of the [B, 78] matrix).
MODULES = [
    ("keyboard", [("midi_f0", 0.0, 127.0, 1.0, False), ("duration", 0.01, 4.0, 0.5, False)]),
    ("adsr_1", _ADSR),
    ("adsr_2", _ADSR),
    ("lfo_1", _LFO),
    ("lfo_2", _LFO),
    ("lfo_1_amp_adsr", _ADSR),
    ("lfo_2_amp_adsr", _ADSR),
    ("lfo_1_rate_adsr", _ADSR),
    ("lfo_2_rate_adsr", _ADSR),
    ("mod_matrix", _MODMATRIX),
    ("vco_1", _VCO),
    ("vco_2", _VCO + [("shape", 0.0, 1.0, 1.0, False)]),
    ("mixer", _MIXER),
]

PARAMS = [
    (mod, name, lo, hi, curve, sym)
    for mod, plist in MODULES
    for (name, lo, hi, curve, sym) in plist
]
NPARAMS = len(PARAMS)
assert NPARAMS == 78
INDEX = {(m, n): i for i, (m, n, *_r) in enumerate(PARAMS)}

# Constants of the signal graph.
CONTROL_RATE = 441
EPS = 1e-6
LFO_EXPONENT = 2.7182817459106445  # torchsynth LFO default exponent tensor(e), as the fp32 it is stored in
NOISE_SEED = 13
REPRODUCIBLE_SUBBATCH = 32
