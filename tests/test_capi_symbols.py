"""The C-ABI library builds, loads, and exports every symbol include/ias_hip.h declares (no compute)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "ias_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ias_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    from inverse_audio_synthesis_amd import _lib
    names = _declared()
    assert len(names) >= 9
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/ias_hip.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype in _lib.SYMBOLS"
    for n in _lib.SYMBOLS:
        assert n in names, f"{n} bound in _lib.py but not declared in include/ias_hip.h"


def test_version_and_arg_checks(lib):
    assert lib.ias_version() >= 100
    # pure host-side argument validation (no GPU touched)
    assert lib.ias_pqmf_out_len(176400, 3, 63) == 58800
    assert lib.ias_pqmf_out_len(16000, 3, 63) == 5334
    assert lib.ias_pqmf_out_len(176400, 64, 63) == 2757
    assert lib.ias_voice_workspace_bytes(128, 176400, 1764) > 128 * 5 * 1764 * 4
    assert lib.ias_voice_workspace_bytes(0, 10, 10) < 0


def test_voice_spec_tables_agree():
    from oracle import synth_spec as a
    from inverse_audio_synthesis_amd import voice_spec as b
    assert a.PARAMS == b.PARAMS and a.NPARAMS == b.NPARAMS == 78
    assert (a.CONTROL_RATE, a.EPS, a.NOISE_SEED, a.LFO_EXPONENT) == (b.CONTROL_RATE, b.EPS, b.NOISE_SEED, b.LFO_EXPONENT)
