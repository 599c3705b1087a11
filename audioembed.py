"""Flat-layout alias so the reference's ``from audioembed import ...`` keeps working (see inverse-audio-synthesis_amd/audioembed.py)."""
from inverse_audio_synthesis_amd.audioembed import *  # noqa: F401,F403
