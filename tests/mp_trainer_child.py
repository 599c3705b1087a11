"""Child process of tests/test_training_gpu.py::test_two_rank_training_keeps_replicas_identical (not a test module):
one rank of a 2-rank pretraining run over gloo, both ranks on the one GPU.  Writes what the parent asserts on to
$IAS_MP_OUT/rank<r>.json."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    from inverse_audio_synthesis_amd.config import load_config
    from inverse_audio_synthesis_amd.harness import VicregAudioParams
    from inverse_audio_synthesis_amd.trainer import Trainer
    out_dir = os.environ["IAS_MP_OUT"]
    overrides = json.loads(os.environ["IAS_MP_OVERRIDES"])
    cfg = load_config(os.path.join(ROOT, "conf"), "config", overrides)
    torch.manual_seed(int(cfg.seed) + 17 * int(os.environ["RANK"]))   # DIFFERENT init per rank: the trainer must broadcast rank 0's
    model = VicregAudioParams(cfg)
    trainer = Trainer(cfg, model, stage="vicreg")
    rec = {"rank": trainer.rank, "world": trainer.world, "steps": [], "batches": []}

    def digest():
        h = hashlib.sha256()
        # parameters only: BatchNorm running statistics are local to a replica (no SyncBN in the reference's plain DDP)
        for _, p in sorted(model.named_parameters()):
            h.update(p.detach().cpu().contiguous().numpy().tobytes())
        return h.hexdigest()

    rec["initial_digest"] = digest()
    step_fn = model.training_step

    def training_step(batch, batch_idx=None):
        rec["batches"].append(int(batch))
        return step_fn(batch, batch_idx)
    model.training_step = training_step
    log_fn = trainer._log

    def log(step, extra=None):
        local = {k: float(v) for k, v in model.logged.items()}
        log_fn(step, extra)
        rec["steps"].append({"step": step, "local": local, "reduced": {k: trainer.history[-1][k] for k in local},
                             "digest": digest()})
    trainer._log = log
    trainer.fit()
    rec["checkpoint_written_by_this_rank"] = trainer.save_checkpoint("probe.ckpt") is not None
    with open(os.path.join(out_dir, f"rank{trainer.rank}.json"), "w") as f:
        json.dump(rec, f)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
