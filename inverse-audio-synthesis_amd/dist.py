"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The reference's only parallelism is Lightning ``strategy: "ddp"`` (/root/reference/conf/config.yaml:8 ->
pretrain.py:98): every rank renders its own batch, gradients are averaged.  Here:
  * ``init_from_env``     rendezvous from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun contract).
  * ``GradBucketer``      bucketed gradient all-reduce overlapped with backward.  A gradient is copied once into
                          its slice of a few large flat buffers and lives there from then on (the optimizer reads
                          the averaged slice); a bucket's all-reduce is issued on RCCL's stream as soon as its
                          last gradient has arrived, and averages inside the collective (ReduceOp.AVG).  xGMI is
                          point-to-point (7 links x ~153 GB/s per GPU) and ring collectives are per-link
                          bound, so buckets are few and large (default 128 MiB) rather than NVSwitch-sized.
  * ``all_reduce_mean``   the ``sync_dist=True`` metric reduction of vicreg_audio_params.py:117-120.
The embedding gather of vicreg.py:79-95 is ``vicreg.FullGatherLayer``.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group if WORLD_SIZE > 1.  -> (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # IAS_DIST_BACKEND=gloo: ranks that share one GPU (tests, rehearsals on a one-GPU box); RCCL otherwise
            backend = os.environ.get("IAS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if torch.cuda.is_available():
            # whatever the backend: the current device (and with it torch.cuda.current_stream(), which every kernel launch
            # of this package reads) must be the one the Trainer places the module on.  Ranks that share GPUs over gloo
            # (LOCAL_RANK >= device count: tests, rehearsals) wrap around
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def all_reduce_mean(t):
    """Mean of a (scalar) tensor over ranks; identity on one rank."""
    if world_size() == 1:
        return t
    t = t.detach().clone()
    dist.all_reduce(t)
    return t / world_size()


class GradBucketer:
    """Bucketed, backward-overlapped gradient averaging for ``module`` (replicas only: no SyncBN,
    matching the reference's plain DDP)."""
    SMALL = 16384          # elements: gradients below this are copied into their bucket by one multi-tensor launch

    def __init__(self, module, bucket_bytes=128 << 20, always_reduce=False, local_only=False):
        """``always_reduce``: issue the collectives on a one-rank group as well (exercises the RCCL path on a
        single GPU; tests).  ``local_only``: no collectives even on several ranks (bench.py's measurement of what the
        all-reduce adds to a step: the same step with and without it)."""
        self.world = world_size()
        self.collective = dist.is_available() and dist.is_initialized() and (self.world > 1 or always_reduce) and \
            not local_only
        if self.collective:
            # replicas start from rank 0's parameters AND buffers (BatchNorm running statistics), as torch DDP /
            # Lightning's strategy "ddp" do at construction (/root/reference/conf/config.yaml:8 -> pretrain.py:97-99)
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, 0)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.buckets = []      # (flat buffer, [params])
        self._pending = {}     # bucket index -> grads still to arrive this step
        self._handles = []     # the post-accumulate-grad hooks (close() removes them)
        self._small = {}       # bucket index -> (views, gradients) of the small tensors waiting for their joint copy
        self._work = []
        # RCCL averages inside the collective (no division pass over the buckets afterwards); gloo has no AVG.  On a one-rank
        # RCCL group (always_reduce: tests, rehearsals) the average is a real RCCL kernel per bucket, which is the point of
        # such a group -- and a necessity inside a captured step: RCCL's in-place one-rank SUM enqueues nothing, and a
        # captured step whose fork onto RCCL's stream and join back enclose no work computed WRONG losses on replay (the
        # forward's, with nothing but the empty branch changed: scripts/diag/dbg_ddp_losses.py, HISTORY.md)
        self._op = dist.ReduceOp.AVG if self.collective and dist.get_backend() == "nccl" else dist.ReduceOp.SUM
        if not self.collective:
            # one replica: nothing to reduce, so no flat buckets and no hooks -- begin_step() drops the gradients and
            # autograd hands each parameter its gradient tensor directly (no per-parameter accumulate / copy launches)
            return
        # reverse registration order ~ the order gradients become ready in backward
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype):
                self._seal(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._seal(cur)
        self._bucket_of, self._offset_of = {}, {}
        for bi, (_flat, plist) in enumerate(self.buckets):
            off = 0
            for p in plist:
                self._bucket_of[p] = bi
                self._offset_of[p] = off          # (looked up by identity: list.index would compare tensors by value)
                off += p.numel()
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.begin_step()

    def close(self):
        """Take this bucketer's hooks off the parameters (before another one is installed on the same module: its hooks
        would go on copying gradients into its buckets and counting them)."""
        for h in self._handles:
            h.remove()
        self._handles = []
        self.collective = False

    def _seal(self, plist):
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        for p in plist:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.buckets.append((flat, plist))

    def begin_step(self):
        """Drop the gradients (call instead of optimizer.zero_grad()).  Several ranks too: autograd then hands every parameter
        a fresh gradient tensor and ``_on_grad`` copies it into the parameter's slice of its bucket -- one read and one write
        per gradient byte.  (Until round 5 the flat buffers were zeroed here and autograd accumulated into the views: a
        write, then two reads and a write: four passes over the 693 MB of the pretraining step's gradients instead of two.)"""
        self._work = []
        for p in self.params:
            p.grad = None
        if not self.collective:
            return
        for bi, (_flat, plist) in enumerate(self.buckets):
            self._pending[bi] = len(plist)
            self._small[bi] = ([], [])

    def _view(self, p):
        flat = self.buckets[self._bucket_of[p]][0]
        off = self._offset_of[p]
        return flat[off:off + p.numel()].view_as(p)

    def _on_grad(self, p):
        bi = self._bucket_of[p]
        flat, plist = self.buckets[bi]
        if p.grad.data_ptr() < flat.data_ptr() or p.grad.data_ptr() >= flat.data_ptr() + flat.numel() * flat.element_size():
            # autograd's own tensor (every step, since begin_step drops the gradients): into the bucket, and re-point --
            # the optimizer then reads the averaged slice
            view = self._view(p)
            if p.numel() < self.SMALL:
                # the many small gradients (biases, normalisation scales: ~150 of the pretraining step's ~200 tensors) go
                # into their bucket together, one multi-tensor launch when the bucket is complete
                self._small[bi][0].append(view)
                self._small[bi][1].append(p.grad)
            else:
                view.copy_(p.grad)
            p.grad = view
        self._pending[bi] -= 1
        if self._pending[bi] < 0:
            raise RuntimeError("GradBucketer: a second backward() since begin_step() -- a bucket is reduced as soon as its "
                               "gradients of ONE backward are in (one backward per step, as the reference's training_step)")
        if self._pending[bi] == 0 and self.collective:
            self._reduce(bi)

    def _reduce(self, bi):
        views, grads = self._small[bi]
        if views:
            torch._foreach_copy_(views, grads)
            self._small[bi] = ([], [])
        self._work.append(dist.all_reduce(self.buckets[bi][0], op=self._op, async_op=True))

    def finish(self):
        """Wait for the outstanding all-reduces; the result is the MEAN over ranks (RCCL: the collective averages itself,
        ``ReduceOp.AVG``; other backends: sum, then one division pass)."""
        if not self.collective:
            return
        # parameters that received no gradient this step still have to take part in the collective: as zeros
        for bi, (flat, plist) in enumerate(self.buckets):
            if self._pending[bi] > 0:
                for p in plist:
                    if p.grad is None:
                        view = self._view(p)
                        view.zero_()
                        p.grad = view
                self._reduce(bi)
                self._pending[bi] = 0
        for w in self._work:
            w.wait()
        self._work = []
        if self._op != dist.ReduceOp.AVG and self.world > 1:
            for flat, _plist in self.buckets:
                flat.div_(self.world)
