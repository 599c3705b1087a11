"""The configs[2] pretraining step at B = 128 on a ONE-rank RCCL group with the bucketed gradient all-reduce switched on
(GradBucketer(always_reduce=True)): what the bucket bookkeeping costs around the collectives.  Three forms, captured step:
  none : no collectives (local_only)
  new  : the tree's GradBucketer (gradients dropped, copied into their bucket slice by the hook, small ones by one multi-tensor launch per bucket)
  newsum: the same with ReduceOp.SUM (on one rank RCCL enqueues nothing for it: timing only, the replayed losses are wrong)
  old  : the round-4 form (flat buffers zeroed, autograd accumulates into the views, sum + division pass)
usage (GPU box): python scripts/diag/time_ddp_onerank.py"""
import os, socket, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from inverse_audio_synthesis_amd import dist as ias_dist
from inverse_audio_synthesis_amd.config import load_config
from inverse_audio_synthesis_amd.harness import VicregAudioParams
from inverse_audio_synthesis_amd.trainer import Trainer


class OldBucketer(ias_dist.GradBucketer):
    def begin_step(self):
        self._work = []
        if not self.collective:
            for p in self.params:
                p.grad = None
            return
        for bi, (flat, plist) in enumerate(self.buckets):
            if any(p.grad is None for p in plist):
                for p in plist:
                    p.grad = self._view(p)
            flat.zero_()
            self._pending[bi] = len(plist)

    def _on_grad(self, p):
        bi = self._bucket_of[p]
        flat, plist = self.buckets[bi]
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and self.collective:
            self._work.append(dist.all_reduce(flat, async_op=True))

    def finish(self):
        if not self.collective:
            return
        for w in self._work:
            w.wait()
        self._work = []
        for flat, _plist in self.buckets:
            flat.div_(self.world)


def main():
    dev = torch.device("cuda:0")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    for mode in (sys.argv[1:] or ("none", "new", "old", "none", "new", "old")):
        torch.manual_seed(42)
        cfg = load_config(os.path.join(ROOT, "conf"), "config", ["vicreg.batch_size=128", "trainer.cuda_graph=true", "param_embed.dropout=0.0"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = VicregAudioParams(cfg)
        tr = Trainer(cfg, model, stage="vicreg", device=dev)
        mb = int(cfg.trainer.bucket_mb) << 20
        tr.bucketer.close()
        if mode == "none":
            tr.bucketer = ias_dist.GradBucketer(model, bucket_bytes=mb, local_only=True)
        elif mode in ("new", "newsum", "newnodefer"):
            tr.bucketer = ias_dist.GradBucketer(model, bucket_bytes=mb, always_reduce=True)
            if mode == "newnodefer":      # every gradient copied by its own launch
                tr.bucketer.SMALL = 0
            if mode == "newsum":          # RCCL's in-place one-rank sum enqueues nothing (see dist.GradBucketer: not inside a capture)
                tr.bucketer._op = dist.ReduceOp.SUM
        else:
            tr.bucketer = OldBucketer(model, bucket_bytes=mb, always_reduce=True)
        model.train()
        for i in range(5):
            tr._graph_step(i, i)
        assert getattr(tr, "_graph", None) is not None
        ts = []
        k = 100
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                tr._graph_step(k, k)
                k += 1
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 10 * 1e3)
        ts.sort()
        print(f"{mode:5s}: {ts[len(ts) // 2]:.3f} ms / step (min {ts[0]:.3f}), buckets {len(tr.bucketer.buckets)}, loss "
              f"{float(model.logged['vicreg/train/loss']):.4f}", flush=True)
        del tr, model
        torch.cuda.empty_cache()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
