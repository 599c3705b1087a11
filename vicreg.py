"""Flat-layout alias so the reference's ``from vicreg import ...`` keeps working (see inverse-audio-synthesis_amd/vicreg.py)."""
from inverse_audio_synthesis_amd.vicreg import *  # noqa: F401,F403
