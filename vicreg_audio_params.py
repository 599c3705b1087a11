"""Flat-layout alias of the reference's ``vicreg_audio_params`` module (see inverse-audio-synthesis_amd/harness.py)."""
from inverse_audio_synthesis_amd.harness import VicregAudioParams  # noqa: F401
