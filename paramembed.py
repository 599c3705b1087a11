"""Flat-layout alias so the reference's ``from paramembed import ...`` keeps working (see inverse-audio-synthesis_amd/paramembed.py)."""
from inverse_audio_synthesis_amd.paramembed import *  # noqa: F401,F403
