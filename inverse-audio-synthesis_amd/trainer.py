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


def _feistel_perm(i, m, seed):
    """Index i of a seeded permutation of range(m), in O(1) memory: a 4-round Feistel network on the next even
    number of bits, cycle-walked back into [0, m).  Draws are therefore WITHOUT replacement, like the reference's
    ``random_split`` permutation (/root/reference/runsetup.py:28-44), without materialising 50 M indices."""
    bits = max(2, (int(m - 1).bit_length() + 1) // 2 * 2)
    half, mask = bits // 2, (1 << (bits // 2)) - 1
    x = int(i)
    while True:
        l, r = x >> half, x & mask
        for rnd in range(4):
            f = ((r * 0x9E3779B1 + (int(seed) + 1) * 0x85EBCA77 + rnd * 0xC2B2AE3D) >> 7) & mask
            l, r = r, l ^ f
        x = (l << half) | r
        if x < m:
            return x


def split_sizes(num_batches, ntest_batches):
    n = int(num_batches)
    lo_val = int(0.9 * (n - ntest_batches))
    return {"train": (0, lo_val), "val": (lo_val, n - ntest_batches), "test": (n - ntest_batches, n)}


def split_indices(num_batches, ntest_batches, seed, count, part, rank=0, world=1, start=0):
    """``count`` batch indices (per rank) of split ``part`` in {"train","val","test"}: consecutive entries of a seeded
    permutation of that split's slice of the index universe (train = [0, 0.9n), val = [0.9n, n - ntest), test = the
    last ntest), rank-strided; no index repeats until the split is exhausted."""
    lo, hi = split_sizes(num_batches, ntest_batches)[part]
    m = hi - lo
    sd = int(seed) * 3 + {"train": 0, "val": 1, "test": 2}[part]
    return [lo + _feistel_perm((start + k * world + rank) % m, m, sd) for k in range(count)]


TUNING_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "tunableop_gfx950.csv")


def setup_gemm_tuning(mode):
    """The step's library GEMMs (8192-wide projector, kernel-2 conv head as GEMMs, the six wide 1x1 convolutions as
    strided-batched GEMMs: 5.6 of the 13 ms of a batch-128 step) run at 100-120 TFLOP/s with the libraries' default
    solutions.  torch's TunableOp picks the fastest rocBLAS / hipBLASLt solution per shape; the choices measured on an
    MI355X are shipped (``tuning/tunableop_gfx950.csv``: fp32 in, fp32 out -- only the kernel changes, not the arithmetic
    type) and looked up at run time: 13.0 -> 12.4 ms per step.  The file carries the library versions it was recorded
    with; TunableOp ignores it when they differ.  -> what was set up ("off", "file", "online", or "unavailable")."""
    mode = str(mode or "off").lower()
    if mode in ("off", "false", "none", "0") or not torch.cuda.is_available():
        return "off"
    try:
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(mode == "online")
        # whatever TunableOp writes goes to a scratch file of this rank's own (online: ./tunableop_online_rank<r>.csv), never
        # into the shipped table; in "file" mode nothing is recorded, so nothing is written at exit either
        import tempfile
        rank = int(os.environ.get("RANK", "0"))
        if mode == "online":
            tunable.set_filename(os.path.join(os.getcwd(), f"tunableop_online_rank{rank}.csv"), False)
        else:
            tunable.set_filename(os.path.join(tempfile.gettempdir(), f"ias_tunableop_{os.getpid()}.csv"), False)
            if hasattr(tunable, "write_file_on_exit"):
                tunable.write_file_on_exit(False)
        loaded = os.path.exists(TUNING_FILE) and bool(tunable.read_file(TUNING_FILE))
        if mode == "online":
            return "online"
        if not loaded:
            # no table, or one recorded with other library versions (TunableOp's validators reject it): the libraries'
            # default solutions -- say so instead of reporting a table that is not in use
            tunable.enable(False)
            return "unavailable"
        return "file"
    except Exception as ex:  # noqa: BLE001 -- a torch build without TunableOp: the libraries' defaults
        import warnings
        warnings.warn(f"trainer.gemm_tuning: TunableOp not available ({type(ex).__name__}: {ex})")
        return "unavailable"


class Trainer:
    def __init__(self, cfg, module, stage="vicreg", device=None):
        self.cfg, self.module, self.stage = cfg, module, stage
        self.rank, self.local_rank, self.world = ias_dist.init_from_env()
        self.device = device or torch.device("cuda", torch.cuda.current_device() if self.world > 1 else self.local_rank)
        self.gemm_tuning = setup_gemm_tuning(cfg.trainer.get("gemm_tuning", "off"))
        # (the entry points seed BEFORE building the module, as runsetup.py:22 does; GradBucketer then broadcasts
        # rank 0's parameters and buffers, as Lightning's DDP does, so replicas start identical by construction)
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

    def _deferred_reductions(self):
        """The trunk's weight-gradient reductions as ONE launch at the end of the backward pass (vision.defer_weight_reductions)
        -- on exactly when nothing reads a gradient DURING the pass: GradBucketer's hooks do, on several ranks."""
        import contextlib
        from . import vision

        @contextlib.contextmanager
        def scope():
            old = vision.defer_weight_reductions(not self.bucketer.collective)
            try:
                yield
            finally:
                vision.defer_weight_reductions(old)
        return scope()

    def _seed(self):
        """The loss' cotangent, a 1 that lives outside any graph pool (autograd otherwise creates it with a fill launch per
        step)."""
        one = getattr(self, "_one", None)
        if one is None or one.device != self.device:
            one = self._one = torch.ones((), dtype=torch.float32, device=self.device)
        return one

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

    def _check_render(self):
        """At the logging cadence (which synchronises anyway): did ANY render since the last look lose a tile?  The render
        turns such a tile into NaN audio, the loss follows, and without this the run would log NaNs without a reason (the
        reference runs with detect_anomaly=True, pretrain.py:96).  Reads the STICKY status word, so it also covers the
        steps nobody looked at, and steps replayed from the captured graph."""
        voice = getattr(self.module, "voice", None)
        if voice is not None and hasattr(voice, "chain_status_sticky") and self.device.type == "cuda" and \
                voice.chain_status_sticky() != 0:
            raise RuntimeError("voice render: a tile's bounded wait for its predecessors expired in a training step since the "
                               "last check (ias_voice_read_status_sticky != 0): that step's audio, loss and update are NaN")

    def save_checkpoint(self, name):
        if self.rank != 0:
            return None
        os.makedirs(self.out_dir, exist_ok=True)
        path = os.path.join(self.out_dir, name)
        torch.save({"state_dict": self.module.state_dict(), "optimizer": self.optimizer.state_dict(),
                    "scheduler": self.scheduler.state_dict() if self.scheduler else None,
                    "step": getattr(self, "_step", 0)}, path)
        return path

    def load_checkpoint(self, path, strict=True):
        """Resume: module, optimizer and scheduler state (so a resumed LARS run does not restart its warm-up)."""
        ck = torch.load(path, map_location=self.device, weights_only=True)   # tensors / plain containers only
        self.module.load_state_dict(ck["state_dict"], strict=strict)
        if ck.get("optimizer") is not None:
            self.optimizer.load_state_dict(ck["optimizer"])
        if self.scheduler is not None and ck.get("scheduler") is not None:
            self.scheduler.load_state_dict(ck["scheduler"])
        self.start_step = int(ck.get("step", 0))
        # the parameters changed under a captured step (its LARS launches carry the parameters' norms from update to update)
        # and under the eager warm-up state: capture again
        self._graph, self._graph_warm = None, 0
        if hasattr(self.optimizer, "invalidate_carried_norms"):
            self.optimizer.invalidate_carried_norms()
        return ck

    # ---- whole-step hipGraph (trainer.cuda_graph) --------------------------------------------------------------
    def _use_graph(self):
        """The whole step replays from one captured hipGraph when asked to (trainer.cuda_graph) -- on several ranks too: the
        bucketed gradient all-reduces (and the embedding gather of trainer.gather_embeddings) are RCCL collectives, which
        capture like any other stream work; only a gloo group (tests / rehearsals on CPU transport) has to stay eager."""
        import torch.distributed as dist
        if not (bool(self.cfg.trainer.get("cuda_graph")) and self.stage == "vicreg" and self.device.type == "cuda" and
                hasattr(self.module, "voice") and not getattr(self, "_graph_failed", False)):
            return False
        if self.bucketer.collective or self.world > 1:
            return dist.is_initialized() and dist.get_backend() == "nccl"
        return True

    def _graph_step(self, batch, step, warmup=3):
        """One training step as a replay of ONE captured hipGraph: render + PQMF + trunk + projector + loss + backward +
        LARS are ~1500 launches per step and the eager loop is bound by issuing them (36 ms wall for 22 ms of kernels at
        batch 128).  Host work stays outside the graph: the synth parameters of the batch are sampled on the CPU and
        copied into the voice's parameter buffer, the scheduler's learning rate is pushed to the device scalar the fused
        LARS step reads.  The first ``warmup`` steps run eagerly (library handles, MIOpen find, allocator), the next one
        is captured (and executed by its first replay)."""
        m, opt = self.module, self.optimizer
        m.voice.randomize(int(batch))

        def one_step():
            self.bucketer.begin_step()            # drop the gradients (a captured step's own live in the graph's pool)
            with self._deferred_reductions():
                m.training_step(None, step).backward(self._seed())   # (a cached 1: `.backward()` alone launches a fill for it)
            self.bucketer.finish()                # several ranks: the bucket all-reduces, issued during backward, joined here
            opt.step()

        g = getattr(self, "_graph", None)
        if g is None:
            n = getattr(self, "_graph_warm", 0)
            if n < warmup:
                self._graph_warm = n + 1
                one_step()
                return
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            try:
                # with collectives in the step the capture must not trip over RCCL's watchdog thread polling its events
                mode = "thread_local" if self.bucketer.collective or self.world > 1 else "global"
                with torch.cuda.graph(g, capture_error_mode=mode):
                    one_step()
            except Exception as ex:                   # something in the step cannot be captured: stay eager
                import warnings
                warnings.warn(f"trainer.cuda_graph: capture failed ({type(ex).__name__}: {ex}); running eagerly")
                self._graph_failed = True
                torch.cuda.synchronize(self.device)
                one_step()
                return
            self._graph = g
            self._graph_logged = m.logged         # the metric tensors of the captured step (graph pool memory)
        # the learning rate of THIS step -> the device scalars the captured LARS launches read (a capture never contains
        # that copy: optim.LARS._sync_group_hyper), in stream order ahead of the replay
        if hasattr(opt, "sync_hyper"):
            opt.sync_hyper()
        g.replay()
        # evaluate() rebinds module.logged to its own eager tensors; a replay re-runs no Python, so point it back at
        # the tensors the replayed kernels write
        m.logged = self._graph_logged

    def _eager_step(self, batch, step):
        """One training step launched from Python: the bucket all-reduces go out during backward (GradBucketer's hooks),
        ``finish`` joins them and averages."""
        self.bucketer.begin_step()
        with self._deferred_reductions():
            loss = self.module.training_step(batch, step)
            loss.backward(self._seed())
        self.bucketer.finish()
        self.optimizer.step()
        return loss

    def fit(self, max_steps=None):
        cfg, st = self.cfg, self.cfg[self.stage]
        lo, hi = split_sizes(cfg.num_batches, cfg.ntest_batches)["train"]
        # null / null = one epoch over the train split (what the reference's max_epochs=1 means), shared by the ranks
        # `steps` is the TOTAL length of the run: a resumed trainer (load_checkpoint -> start_step) runs the remainder, with
        # the global step number in checkpoint names and in the logging / validation cadence
        steps = max_steps or cfg.trainer.max_steps or st.limit_train_batches or (hi - lo) // self.world
        start = min(getattr(self, "start_step", 0), steps)
        todo = steps - start
        idx = split_indices(cfg.num_batches, cfg.ntest_batches, cfg.seed, todo, "train", self.rank, self.world,
                            start=start * self.world) if 0 < todo <= 1 << 20 else None
        val_every, val_count = st.get("val_check_interval"), st.get("limit_val_batches")
        self.module.train()
        t0 = time.perf_counter()
        for i in range(todo):
            step = start + i                      # global step
            batch = idx[i] if idx is not None else split_indices(
                cfg.num_batches, cfg.ntest_batches, cfg.seed, 1, "train", self.rank, self.world,
                start=step * self.world)[0]
            if self._use_graph():
                self._graph_step(batch, step)
            else:
                self._eager_step(batch, step)
            if self.scheduler is not None:
                self.scheduler.step()
            self._step = step + 1
            if step % int(cfg.trainer.log_every) == 0 or step == steps - 1:
                self._check_render()
                self._log(step, {"elapsed_s": round(time.perf_counter() - t0, 3)})
            every = st.get("checkpoint_every_nbatches")
            if every and (step + 1) % int(every) == 0:
                self.save_checkpoint(f"{self.stage}-step{step + 1:06d}.ckpt")
            if val_every and val_count and (step + 1) % int(val_every) == 0 and hasattr(self.module, "validation_step"):
                val = self.evaluate("val", count=int(val_count))   # Lightning's val_check_interval
                keys = val[0].keys() if val else []
                self.history.append({"step": step, **{k: sum(v[k] for v in val) / len(val) for k in keys}})
                if self.rank == 0:
                    print(json.dumps(self.history[-1]), flush=True)
                self.module.train()
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


def load_reference_state_dict(module, state_dict):
    """Load a state_dict shaped like the reference's Lightning ``vicreg.ckpt`` (/root/reference/downstream.py:29) into
    this build's ``VicregAudioParams``.  The reference's checkpoint also holds the torchvision classifier head
    (``*.classifier.*``; this build keeps only ``.features``, the part audioembed.py:61 uses) and torchsynth's
    per-module ``voice.*`` parameter tensors (the Voice here registers no parameters: it is re-seeded per batch);
    those are dropped.  Anything else that does not line up is reported.  -> (missing, unexpected, dropped)"""
    dropped = [k for k in state_dict if ".classifier." in k or k.startswith("voice.") or ".voice." in k]
    kept = {k: v for k, v in state_dict.items() if k not in set(dropped)}
    res = module.load_state_dict(kept, strict=False)
    return list(res.missing_keys), list(res.unexpected_keys), dropped
