"""Flat-layout alias so the reference's ``from pqmf import ...`` keeps working (see inverse-audio-synthesis_amd/pqmf.py)."""
from inverse_audio_synthesis_amd.pqmf import *  # noqa: F401,F403
