"""Does the B = 128 pretraining step, replayed from its captured graph WITHOUT host synchronisation between steps, train the
same on a one-rank RCCL group with GradBucketer(always_reduce=True) as without collectives?  Final loss and a parameter
checksum after 5 + 50 steps, three loop forms: plain (replays back to back), sync (torch.cuda.synchronize() after every
step), kern (a small kernel on the stream between two replays).
usage (GPU box): python scripts/diag/dbg_ddp_losses.py"""
import os, socket, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from inverse_audio_synthesis_amd import dist as ias_dist
from inverse_audio_synthesis_amd.config import load_config
from inverse_audio_synthesis_amd.harness import VicregAudioParams
from inverse_audio_synthesis_amd.trainer import Trainer

dev = torch.device("cuda:0")
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B = int(os.environ.get("B", "128"))
for form in ("kern",):
    for mode in (sys.argv[1:] or ("none", "new", "new_sum", "new_noar")):
        torch.manual_seed(42)
        cfg = load_config(os.path.join(ROOT, "conf"), "config", [f"vicreg.batch_size={B}", "trainer.cuda_graph=true", "param_embed.dropout=0.0"])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model = VicregAudioParams(cfg)
        tr = Trainer(cfg, model, stage="vicreg", device=dev)
        mb = int(cfg.trainer.bucket_mb) << 20
        tr.bucketer.close()
        tr.bucketer = ias_dist.GradBucketer(model, bucket_bytes=mb, local_only=(mode == "none"), always_reduce=(mode != "none"))
        if mode == "new_sum":             # RCCL's in-place one-rank SUM: nothing is enqueued between the fork and the join
            tr.bucketer._op = dist.ReduceOp.SUM
        if mode == "new_noar":            # the hooks and their copies, but no collective call at all
            def _reduce_without_collective(bi, b=tr.bucketer):
                views, grads = b._small[bi]
                if views:
                    torch._foreach_copy_(views, grads)
                    b._small[bi] = ([], [])
            tr.bucketer._reduce = _reduce_without_collective
        model.train()
        buf = torch.zeros(64, device=dev)
        for i in range(5):
            tr._graph_step(i, i)
        k = 100
        for _ in range(5):
            torch.cuda.synchronize()
            for _ in range(10):
                tr._graph_step(k, k)
                if form == "sync":
                    torch.cuda.synchronize()
                elif form == "kern":
                    buf[k % 64].copy_(model.logged["vicreg/train/loss"])
                k += 1
            torch.cuda.synchronize()
        cs = sum(float(p.double().sum()) for p in model.parameters())
        cs2 = sum(float(p.double().pow(2).sum()) for p in model.parameters())
        lg = {k.split("/")[-1]: float(v) for k, v in model.logged.items()}
        comb = 25.0 * lg["repr_loss"] + 25.0 * lg["std_loss"] + lg["cov_loss"]
        pe = [float(p.double().sum()) for n, p in model.named_parameters() if "param" in n.lower()][:4]
        print(f"form={form:5s} {mode:5s} loss {lg['loss']:.6f} (25 repr + 25 std + cov = {comb:.6f}) parts {lg['repr_loss']:.6f} {lg['std_loss']:.6f} "
              f"{lg['cov_loss']:.6f} checksum {cs:.12f} sumsq {cs2:.12f} paramembed sums {pe} hist {[round(float(v), 4) for v in buf.cpu()[36:44]]} params01 is batch 149: {bool(torch.equal(model.voice.params01.cpu(), __import__('inverse_audio_synthesis_amd.voice', fromlist=['x']).sample_params01(B, 149)))}", flush=True)
        del tr, model
        torch.cuda.empty_cache()
dist.destroy_process_group()
