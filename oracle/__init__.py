"""CPU oracle of the hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package,
and only as the checker / the timed CPU baseline.  The product (``inverse-audio-synthesis_amd/``) never
imports it and has no CPU fallback.

Modules: ``pqmf_oracle`` and ``vicreg_oracle`` restate /root/reference/pqmf.py and vicreg.py and are pinned
by golden vectors generated from the imported reference (tests/golden/, scripts/make_golden.py);
``synth_oracle`` (torchsynth Voice) and ``spectral_oracle`` (torchaudio / auraloss semantics) restate absent
third-party code from its published definition -- PARITY UNPINNED.
"""
