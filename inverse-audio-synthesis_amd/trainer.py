"""A minimal trainer in place of Lightning's ``Trainer.fit/test`` (host glue).

What it keeps from the reference's run: seed 42 (``runsetup.py:22``), batch-index datasets split
90/10/ntest from ``range(num_batches)`` (:28-44, drawn lazily instead of materialising 50 M indices),
one optimizer step per batch index, scheduler stepped every optimizer step, metric names, periodic
checkpoints of the module state_dict.  Data parallelism: every rank takes its own batch indices
(rank-strided), gradients averaged by ``dist.GradBucketer`` over RCCL.
"""
import json
import os
import time

import torch

from . import dist as ias_dist


def split_indices(num_batches, ntest_batches, seed, count, part, rank=0, world=1):
    """``count`` batch indices of split ``part`` in {"train","val","test"}: a seeded draw from the
    index universe (train = [0, 0.9n), val = [0.9n, n - ntest), test = the last ntest)."""
    n = int(num_batches)
    lo_val = int(0.9 * (n - ntest_batches))
    ranges = {"train": (0, lo_val), "val": (lo_val, n - ntest_batches), "test": (n - ntest_batches, n)}
    lo, hi = ranges[part]
    g = torch.Generator().manual_seed(int(seed) + {"train": 0, "val": 1, "test": 2}[part])
    idx = torch.randint(lo, hi, (count * world,), generator=g)
    return idx[rank::world].tolist()


class Trainer:
    def __init__(self, cfg, module, stage="vicreg", device=None):
        self.cfg, self.module, self.stage = cfg, module, stage
        self.rank, self.local_rank, self.world = ias_dist.init_from_env()
        self.device = device or torch.device("cuda", self.local_rank)
        torch.manual_seed(cfg.seed)
        self.module.to(self.device)
        opt = module.configure_optimizers()
        if isinstance(opt, dict):
            self.optimizer, self.scheduler = opt["optimizer"], opt["lr_scheduler"]["scheduler"]
        else:
            self.optimizer, self.scheduler = opt, None
        t = cfg.trainer
        self.bucketer = ias_dist.GradBucketer(module, bucket_bytes=int(t.bucket_mb) << 20)
        self.out_dir = t.out_dir
        self.history = []

    def _log(self, step, extra=None):
        rec = {"step": step}
        for k, v in self.module.logged.items():
            rec[k] = float(ias_dist.all_reduce_mean(v))   # sync_dist=True
        if self.scheduler is not None:
            rec["lr"] = self.scheduler.get_last_lr()[0]
        rec.update(extra or {})
        self.history.append(rec)
        if self.rank == 0:
            print(json.dumps(rec), flush=True)

    def save_checkpoint(self, name):
        if self.rank != 0:
            return None
        os.makedirs(self.out_dir, exist_ok=True)
        path = os.path.join(self.out_dir, name)
        torch.save({"state_dict": self.module.state_dict(), "optimizer": self.optimizer.state_dict(),
                    "scheduler": self.scheduler.state_dict() if self.scheduler else None}, path)
        return path

    def load_checkpoint(self, path):
        ck = torch.load(path, map_location=self.device)
        self.module.load_state_dict(ck["state_dict"])
        return ck

    def fit(self, max_steps=None):
        cfg, st = self.cfg, self.cfg[self.stage]
        steps = max_steps or cfg.trainer.max_steps or st.limit_train_batches
        assert steps, "set trainer.max_steps or <stage>.limit_train_batches"
        idx = split_indices(cfg.num_batches, cfg.ntest_batches, cfg.seed, steps, "train", self.rank, self.world)
        self.module.train()
        t0 = time.perf_counter()
        for step, batch in enumerate(idx):
            self.bucketer.begin_step()
            loss = self.module.training_step(batch, step)
            loss.backward()
            self.bucketer.finish()
            self.optimizer.step()
            if self.scheduler is not None:
                self.scheduler.step()
            if step % int(cfg.trainer.log_every) == 0 or step == steps - 1:
                self._log(step, {"elapsed_s": round(time.perf_counter() - t0, 3)})
            every = st.get("checkpoint_every_nbatches")
            if every and (step + 1) % int(every) == 0:
                self.save_checkpoint(f"{self.stage}-step{step + 1:06d}.ckpt")
        self.save_checkpoint(f"{self.stage}-last.ckpt")
        return self.history

    @torch.no_grad()
    def evaluate(self, part="val", count=1, step_fn="validation_step"):
        cfg = self.cfg
        idx = split_indices(cfg.num_batches, cfg.ntest_batches, cfg.seed, count, part, self.rank, self.world)
        self.module.eval()
        out = []
        for i, batch in enumerate(idx):
            getattr(self.module, step_fn)(batch, i)
            out.append({k: float(ias_dist.all_reduce_mean(v)) for k, v in self.module.logged.items()})
        return out
