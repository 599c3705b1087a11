"""MI355X-native inner loop of turian/inverse-audio-synthesis.

Voice render -> PQMF filterbank -> STFT/mel spectral loss, and the VICReg loss, as
hand-written HIP kernels for gfx950 behind a C ABI (``csrc/libias_hip.so``,
``include/ias_hip.h``), wrapped in the reference's own module API.  Import as
``inverse_audio_synthesis_amd`` (the directory name carries a hyphen; the
top-level ``inverse_audio_synthesis_amd.py`` aliases it).
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
