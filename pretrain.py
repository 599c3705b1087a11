#!/usr/bin/env python3
"""VICReg pre-training entry point: ``python pretrain.py [key=value ...]`` (config root ``conf/``,
name ``config``), as /root/reference/pretrain.py:51-129 -- hosted on this build's own trainer."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def app(overrides=None):
    from inverse_audio_synthesis_amd.config import load_config
    from inverse_audio_synthesis_amd.harness import VicregAudioParams
    from inverse_audio_synthesis_amd.trainer import Trainer
    cfg = load_config(os.path.join(ROOT, "conf"), "config", overrides if overrides is not None else sys.argv[1:])
    import torch
    torch.manual_seed(int(cfg.seed))     # seed_everything(42) BEFORE the model is built (runsetup.py:22 -> pretrain.py:60)
    model = VicregAudioParams(cfg)
    trainer = Trainer(cfg, model, stage="vicreg")
    if trainer.rank == 0:
        n = sum(p.numel() for p in model.parameters())
        print(f"VicregAudioParams: {n / 1e6:.1f} M parameters, batch {cfg.vicreg.batch_size} per rank, "
              f"world {trainer.world}", flush=True)
    history = trainer.fit()
    val = trainer.evaluate("val", count=min(int(cfg.vicreg.limit_val_batches or 1), 2))
    if trainer.rank == 0:
        print({"validation": val}, flush=True)
    return history


if __name__ == "__main__":
    app()
