#!/usr/bin/env python3
"""Downstream stage (frozen VICReg -> parameters) as a module and an entry point.

As a module it exposes the names of /root/reference/audio_to_params.py (``AudioRepresentationToParams``
:16-53, ``AudioToParams`` :177-312).  Run as a script it does what /root/reference/downstream.py:20-70
does: load ``vicreg.ckpt`` if present (else a fresh VicregAudioParams), train ``AudioToParams``, run one
test step: ``python audio_to_params.py [key=value ...]``."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from inverse_audio_synthesis_amd.harness import AudioToParams  # noqa: E402,F401
from inverse_audio_synthesis_amd.paramembed import AudioRepresentationToParams  # noqa: E402,F401


def app(overrides=None):
    import torch
    from inverse_audio_synthesis_amd.config import load_config
    from inverse_audio_synthesis_amd.harness import VicregAudioParams
    from inverse_audio_synthesis_amd.trainer import Trainer
    cfg = load_config(os.path.join(ROOT, "conf"), "config", overrides if overrides is not None else sys.argv[1:])
    # both Voices must render the same batch size
    cfg.vicreg.batch_size = cfg.audio_to_params.batch_size
    from inverse_audio_synthesis_amd.trainer import load_reference_state_dict
    torch.manual_seed(int(cfg.seed))     # before any module is built, as runsetup.py:22
    vicreg = VicregAudioParams(cfg)
    ckpt = os.path.join(ROOT, "vicreg.ckpt")
    if os.path.exists(ckpt):
        # a checkpoint of this build, or one shaped like the reference's Lightning file (extra classifier / voice keys)
        # (tensors and plain containers only: nothing in the file is executed.  A Lightning checkpoint that carries pickled
        # hyper-parameter objects is refused by this loader -- re-save its "state_dict" entry alone)
        sd = torch.load(ckpt, map_location="cpu", weights_only=True)["state_dict"]
        missing, unexpected, dropped = load_reference_state_dict(vicreg, sd)
        if missing or unexpected:
            print({"vicreg.ckpt": {"missing": missing, "unexpected": unexpected, "dropped": len(dropped)}}, flush=True)
    model = AudioToParams(cfg, vicreg)
    trainer = Trainer(cfg, model, stage="audio_to_params")
    history = trainer.fit()
    test = trainer.evaluate("test", count=int(cfg.ntest_batches), step_fn="test_step")
    if trainer.rank == 0:
        print({"test": test}, flush=True)
    return history


if __name__ == "__main__":
    app()
