"""LARS + linear-warmup/cosine schedule, as the reference configures them.

/root/reference/vicreg_audio_params.py:134-151: ``flash.core.optimizers.LARS(params, weight_decay=wd,
lr = batch_size / 256 * base_lr)`` and ``pl_bolts LinearWarmupCosineAnnealingLR(optimizer,
warmup_epochs, max_epochs, warmup_start_lr, eta_min)`` stepped every optimizer step (:154-165).
Neither package is in this image; both are restated from their published definitions (LARS: You et
al. 2017 as implemented by lightning-flash/bolts: trust ratio on parameters with weight decay, then
plain momentum-SGD; default momentum 0, trust_coefficient 1e-3, eps 1e-8).  On a ROCm device with momentum 0 (the
reference's configuration) the whole step is three HIP launches (``ias_lars_step``: norm partials, per-tensor
coefficients, update; csrc/optim_kernels.hip); otherwise multi-tensor ``torch._foreach`` ops.
"""
import math

import torch


class LARS(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False,
                 trust_coefficient=0.001, eps=1e-8):
        assert lr >= 0 and momentum >= 0 and weight_decay >= 0
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                        nesterov=nesterov, trust_coefficient=trust_coefficient, eps=eps)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            gs = [p.grad for p in ps]
            wd, lr, mom = group["weight_decay"], group["lr"], group["momentum"]
            if mom == 0 and self._hip_ok(ps, gs):
                self._hip_step(gi, group, ps, gs)
                continue
            if wd != 0:
                p_norm = torch._foreach_norm(ps)
                g_norm = torch._foreach_norm(gs)
                pn, gn = torch.stack(p_norm), torch.stack(g_norm)
                # flash's LARS applies BOTH the weight decay and the trust ratio only where p_norm != 0 and g_norm != 0
                on = (pn != 0) & (gn != 0)
                ratio = torch.where(on, pn / (gn + pn * wd + group["eps"]) * group["trust_coefficient"],
                                    torch.ones_like(pn))
                upd = torch._foreach_mul(gs, list(ratio.unbind(0)))
                torch._foreach_add_(upd, torch._foreach_mul(ps, list((ratio * wd * on).unbind(0))))
            else:
                upd = [g.clone() for g in gs]
            if mom != 0:
                bufs = []
                for p, u in zip(ps, upd):
                    st = self.state[p]
                    if "momentum_buffer" not in st:
                        st["momentum_buffer"] = u.clone()
                    else:
                        st["momentum_buffer"].mul_(mom).add_(u, alpha=1 - group["dampening"])
                    bufs.append(st["momentum_buffer"])
                upd = torch._foreach_add(upd, bufs, alpha=mom) if group["nesterov"] else bufs
            torch._foreach_add_(ps, upd, alpha=-lr)
        return loss


    # ------------------------------------------------------------------ fused HIP path (momentum 0)
    HYPER_RING = 8   # pinned staging slots of the device hyper-parameters (see _sync_group_hyper)

    def _group_hyper(self, gi, dev):
        """The device scalars (lr, weight decay, trust coefficient, eps) of group ``gi`` and their staging ring: ONE
        allocation per group for the optimizer's lifetime, made OUTSIDE any hipGraph capture.  A tensor allocated during a
        capture lives in the graph's private pool, where a replay's earlier kernels may reuse its bytes (the pool recycles
        what the captured step freed before the allocation): values written into it from outside the graph -- the
        scheduler's learning rate before each replay -- would be overwritten by the replay itself."""
        table = self.__dict__.setdefault("_hip_hyper", {})
        h = table.get(gi)
        if h is not None and h["dev"].device != dev:      # the module moved to another device after its first step
            h = None
        if h is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("LARS: the first fused step of a parameter group cannot run inside a hipGraph capture "
                                   "(its device hyper-parameters must be allocated outside the graph's pool): run one "
                                   "eager step, or call optimizer.prepare_capture(), first")
            h = table[gi] = dict(dev=torch.empty(4, dtype=torch.float32, device=dev),
                                 ring=[torch.empty(4, dtype=torch.float32).pin_memory() for _ in range(self.HYPER_RING)],
                                 events=[None] * self.HYPER_RING, pos=0, vals=None)
        return h

    def _group_carry(self, gi, dev, pkey, nchunks):
        """The per-chunk sums of squares of group ``gi``'s parameters, carried from one update to the next
        (``ias_lars_step_carry``): the update pass writes them, the next step's norm pass then reads only the gradient.
        Persistent and OUTSIDE any graph pool for the same reason as ``_group_hyper``; -> None when it cannot be set up
        here (first use inside a capture, or another parameter set inside a capture)."""
        table = self.__dict__.setdefault("_hip_carry", {})
        st = table.get(gi)
        if st is None or st["pkey"] != pkey or st["partials"].device != dev:
            if torch.cuda.is_current_stream_capturing():
                return None
            st = table[gi] = dict(pkey=pkey, partials=torch.zeros(2 * nchunks, dtype=torch.float64, device=dev),
                                  valid=False, versions=None)
        return st

    def invalidate_carried_norms(self):
        """Forget the carried parameter norms (after anything wrote the parameters behind torch's version counters)."""
        for st in self.__dict__.get("_hip_carry", {}).values():
            st["valid"] = False

    def prepare_capture(self):
        """Allocate what must not live in a graph's pool (see ``_group_hyper``) for every group whose parameters are on a
        ROCm device; for loops that capture their very first step."""
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.is_cuda]
            if ps:
                self._sync_group_hyper(group, self._group_hyper(gi, ps[0].device))

    @staticmethod
    def _sync_group_hyper(group, h):
        """Push (lr, weight decay, trust coefficient, eps) to the device scalars the fused step reads, when they changed.
        The copy is asynchronous and the host may run many steps ahead of the GPU (a replaying loop never synchronises),
        so the staging buffer a copy reads must not be rewritten before that copy has executed: the values go through a
        ring of pinned slots, each guarded by an event recorded behind its last copy (the host only waits when it is a
        whole ring ahead).  Never inside a hipGraph capture: a captured copy would re-read its slot at every replay and
        put a stale learning rate back; the replaying loop pushes the values itself (``sync_hyper``) before each replay."""
        vals = (float(group["lr"]), float(group["weight_decay"]), float(group["trust_coefficient"]), float(group["eps"]))
        if h["vals"] == vals or torch.cuda.is_current_stream_capturing():
            return
        i = h["pos"]
        h["pos"] = (i + 1) % len(h["ring"])
        if h["events"][i] is not None:
            h["events"][i].synchronize()
        host = h["ring"][i]
        host[0], host[1], host[2], host[3] = vals
        h["dev"].copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        h["events"][i] = ev
        h["vals"] = vals

    def sync_hyper(self):
        """Push lr / weight decay / trust coefficient / eps of every group to the device scalars the fused step reads.
        ``step()`` does this itself; a step replayed from a captured hipGraph cannot (the Python around the launches
        does not run again), so the replaying loop calls this after the scheduler has moved the learning rate."""
        for gi, group in enumerate(self.param_groups):
            h = self.__dict__.get("_hip_hyper", {}).get(gi)
            if h is not None:
                self._sync_group_hyper(group, h)

    @staticmethod
    def _hip_ok(ps, gs):
        return all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous() and
                   g.dtype == torch.float32 and g.device == p.device for p, g in zip(ps, gs)) and \
            len({p.device for p in ps}) == 1

    def _hip_step(self, gi, group, ps, gs):
        from . import _lib
        lib = _lib.load()
        dev = ps[0].device
        cache = self.__dict__.setdefault("_hip_tables", {})
        key = tuple((p.data_ptr(), g.data_ptr(), p.numel()) for p, g in zip(ps, gs))
        ent = cache.get(gi)
        if ent is None or ent["key"] != key:
            chunk = lib.ias_lars_chunk_elems()
            tensors, chunks, first = [], [], [0]
            for t, (pp, gp, n) in enumerate(key):
                tensors.append((pp, gp, n))
                nc = (n + chunk - 1) // chunk
                chunks.extend((t, c) for c in range(nc))
                first.append(first[-1] + nc)
            def up(values, dtype):
                # pinned staging + asynchronous copy: also legal while a hipGraph is being captured (the gradients of
                # a captured step live in the graph's private pool, so the tables are rebuilt during capture)
                host = torch.tensor(values, dtype=dtype).pin_memory()
                return host, torch.empty(host.shape, dtype=dtype, device=dev).copy_(host, non_blocking=True)
            t_host, t_dev = up(tensors, torch.int64)
            c_host, c_dev = up(chunks, torch.int32)
            f_host, f_dev = up(first, torch.int32)
            ent = dict(key=key, n=len(key), nchunks=len(chunks), hosts=(t_host, c_host, f_host),
                       tensors=t_dev, chunks=c_dev, first=f_dev,
                       partials=torch.empty(2 * len(chunks), dtype=torch.float64, device=dev),
                       coef=torch.empty(2 * len(key), dtype=torch.float32, device=dev),
                       hyper=self._group_hyper(gi, dev)["dev"])
            cache[gi] = ent
        self._sync_group_hyper(group, self._group_hyper(gi, dev))
        skip = group["weight_decay"] == 0
        if skip:
            c2 = ent["coef"].view(-1, 2)
            c2[:, 0].fill_(1.0)
            c2[:, 1].fill_(0.0)
        # the parameters' norms travel from update to update: the update pass sums the squares of what it writes (in the
        # norm pass' own order: the same bits), so the norm pass of the next step reads the gradient only.  Valid while
        # nobody else wrote the parameters -- torch's version counters say so (this optimizer's kernels do not bump them)
        st = None if skip else self._group_carry(gi, dev, tuple((p.data_ptr(), p.numel()) for p in ps), ent["nchunks"])
        if st is None:
            self.invalidate_carried_norms()
            _lib.check(lib.ias_lars_step(_lib.ptr(ent["tensors"]), _lib.ptr(ent["chunks"]), _lib.ptr(ent["first"]),
                                         _lib.ptr(ent["partials"]), _lib.ptr(ent["coef"]), _lib.ptr(ent["hyper"]),
                                         ent["n"], ent["nchunks"], int(skip), _lib.stream()), "ias_lars_step")
            return
        versions = tuple(p._version for p in ps)
        carry = 2 if (st["valid"] and st["versions"] == versions) else 1
        _lib.check(lib.ias_lars_step_carry(_lib.ptr(ent["tensors"]), _lib.ptr(ent["chunks"]), _lib.ptr(ent["first"]),
                                           _lib.ptr(st["partials"]), _lib.ptr(ent["coef"]), _lib.ptr(ent["hyper"]),
                                           ent["n"], ent["nchunks"], carry, _lib.stream()), "ias_lars_step_carry")
        st["valid"], st["versions"] = True, tuple(p._version for p in ps)


class LinearWarmupCosineAnnealingLR:
    """Closed form of pl_bolts' scheduler; ``step()`` once per optimizer step ("epochs" are steps here,
    exactly as the reference uses it with interval="step")."""

    def __init__(self, optimizer, warmup_epochs, max_epochs, warmup_start_lr=0.0, eta_min=0.0):
        self.optimizer = optimizer
        self.warmup_epochs, self.max_epochs = int(warmup_epochs), int(max_epochs)
        self.warmup_start_lr, self.eta_min = float(warmup_start_lr), float(eta_min)
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = 0
        self._apply()

    def lr_at(self, epoch, base_lr):
        if epoch < self.warmup_epochs:
            return self.warmup_start_lr + epoch * (base_lr - self.warmup_start_lr) / max(self.warmup_epochs - 1, 1)
        span = max(self.max_epochs - self.warmup_epochs, 1)
        return self.eta_min + 0.5 * (base_lr - self.eta_min) * (1 + math.cos(math.pi * (epoch - self.warmup_epochs) / span))

    def _apply(self):
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = self.lr_at(self.last_epoch, base)

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch}

    def load_state_dict(self, sd):
        self.last_epoch = sd["last_epoch"]
        self._apply()
