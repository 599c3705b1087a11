"""The RCCL branches on one GPU: a one-rank ``nccl`` process group runs FullGatherLayer's
all_gather_into_tensor / reduce_scatter_tensor (vicreg.py of this package; reference /root/reference/vicreg.py:79-95),
GradBucketer's broadcast + bucketed all-reduce (reference: Lightning strategy "ddp", conf/config.yaml:8) and the
global-batch VICReg loss through them, against the oracle.  (Two-rank semantics are covered on CPU with gloo in
tests/test_distributed_cpu.py; an 8-GPU node is only available to the driver.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


@pytest.fixture()
def nccl_group(dev):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        yield
    finally:
        dist.destroy_process_group()


def test_full_gather_layer_on_rccl(lib, dev, nccl_group):
    from inverse_audio_synthesis_amd.vicreg import FullGatherLayer
    from oracle import vicreg_oracle as vo
    assert dist.get_backend() == "nccl"
    local = torch.randn(16, 256, generator=torch.Generator().manual_seed(3))
    x = local.to(dev).requires_grad_()
    gathered = FullGatherLayer.apply(x)
    ref = vo.full_gather_forward([local])
    assert len(gathered) == 1 and torch.equal(gathered[0].cpu(), ref[0])
    w = torch.randn(16, 256, generator=torch.Generator().manual_seed(4))
    (gathered[0] * w.to(dev)).sum().backward()
    want = vo.full_gather_backward([(w,)], 0)
    assert torch.allclose(x.grad.cpu(), want, atol=1e-6)


def test_global_batch_vicreg_loss_through_the_gather(lib, dev, nccl_group):
    """BASELINE configs[3] path on one rank: all-gather of x and y, loss on the gathered batch (HIP kernels),
    reduce-scatter of the gradients."""
    import types
    from inverse_audio_synthesis_amd.vicreg import VICReg
    from oracle import vicreg_oracle as vo
    B, D = 64, 512
    cfg = types.SimpleNamespace(dim=32, embeddim=D, vicreg=types.SimpleNamespace(
        mlp="64-64-%d", batch_size=B, sim_coeff=25.0, std_coeff=25.0, cov_coeff=1.0))
    m = VICReg(cfg, torch.nn.Identity(), torch.nn.Identity(), gather_distributed="always").to(dev)
    x0 = torch.randn(B, D, generator=torch.Generator().manual_seed(0))
    y0 = torch.randn(B, D, generator=torch.Generator().manual_seed(1))
    x, y = x0.to(dev).requires_grad_(), y0.to(dev).requires_grad_()
    out = m.loss(x, y)
    out[0].backward()
    xr, yr = x0.clone().requires_grad_(), y0.clone().requires_grad_()
    ref = vo.loss(xr, yr, B, D)
    ref[0].backward()
    assert abs(out[0].item() - ref[0].item()) <= 2e-3 * abs(ref[0].item())
    assert (x.grad.cpu() - xr.grad).abs().max().item() <= 2e-3 * xr.grad.abs().max().item()
    assert (y.grad.cpu() - yr.grad).abs().max().item() <= 2e-3 * yr.grad.abs().max().item()


def test_grad_bucketer_on_rccl(lib, dev, nccl_group):
    from inverse_audio_synthesis_amd.dist import GradBucketer, all_reduce_mean
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(),
                                torch.nn.Linear(16, 4)).to(dev)
    b = GradBucketer(model, bucket_bytes=300, always_reduce=True)   # tiny buckets: several RCCL all-reduces
    assert b.collective and len(b.buckets) >= 2
    data = torch.randn(5, 6, generator=torch.Generator().manual_seed(50)).to(dev)
    b.begin_step()
    model(data).pow(2).sum().backward()
    b.finish()
    got = [p.grad.clone() for p in model.parameters()]
    for p in model.parameters():
        p.grad = None
    ref = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(),
                              torch.nn.Linear(16, 4)).to(dev)
    ref.load_state_dict({k: v for k, v in model.state_dict().items() if "running" not in k and "num_batches" not in k},
                        strict=False)
    ref(data).pow(2).sum().backward()
    for g, p in zip(got, ref.parameters()):
        assert torch.allclose(g, p.grad, atol=1e-5)
    assert all_reduce_mean(torch.tensor(3.0, device=dev)).item() == 3.0


def _spawn_two(script, extra_env, tmp_path, timeout=600):
    """Two FRESH child processes (never a re-exec of this one) as ranks 0 / 1 over gloo on the one GPU."""
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   IAS_DIST_BACKEND="gloo", IAS_MP_OUT=str(tmp_path), **extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=timeout)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    return outs


@pytest.mark.parametrize("Bl,D", [(64, 512), (96, 1024)])
def test_module_loss_with_the_gather_on_at_world_size_two(lib, dev, tmp_path, Bl, D):
    """BASELINE configs[3] through the MODULE at world size 2 (/root/reference/vicreg.py:38-39 un-commented, :47-48,
    :79-95): every rank's ``VICReg.loss`` 4-tuple equals the oracle's loss on the concatenated [2 B_l, D] batch with the
    denominator B_global - 1 (cfg.vicreg.batch_size is the per-rank batch), one collective per direction, and a rank's
    gradient is W x its rows of the oracle's autograd gradient -- FullGatherLayer's backward SUMS the ranks' cotangents
    (:92-95) and every rank holds the same loss; the gradient average over ranks (DDP) takes the W out again.
    (96 rows per rank: global batch 192 > 128 runs the 256-tile kernels through the strided entry points.)"""
    from oracle import vicreg_oracle as vo
    _spawn_two("mp_vicreg_child.py", {"IAS_MP_BL": str(Bl), "IAS_MP_D": str(D)}, tmp_path)
    W = 2
    xg = torch.randn(W * Bl, D, generator=torch.Generator().manual_seed(0)).requires_grad_()
    yg = torch.randn(W * Bl, D, generator=torch.Generator().manual_seed(1)).requires_grad_()
    ref = vo.loss(xg, yg, W * Bl, D)
    ref[0].backward()
    wrong = vo.loss(xg.detach(), yg.detach(), Bl, D)            # the per-rank denominator round 3 shipped: ~W^2 too large
    assert wrong[3].item() > 3.0 * ref[3].item()
    for r in range(W):
        rec = torch.load(tmp_path / f"vicreg_rank{r}.pt")
        for got, want, tol in zip(rec["out"], ref, (2e-3, 1e-5, 1e-5, 2e-3)):
            assert abs(got - want.item()) <= tol * abs(want.item()), (r, rec["out"], [float(v) for v in ref])
        assert rec["collectives_forward"] == 1 and rec["collectives_total"] == 2      # ONE exchange per direction
        # what vicreg.py:92-95 defines: every rank contributes the oracle's gradient rows, summed over the W ranks
        wx = vo.full_gather_backward([tuple(xg.grad[q * Bl:(q + 1) * Bl] for q in range(W))] * W, r)
        wy = vo.full_gather_backward([tuple(yg.grad[q * Bl:(q + 1) * Bl] for q in range(W))] * W, r)
        assert torch.equal(wx, W * xg.grad[r * Bl:(r + 1) * Bl])
        assert (rec["gx"] - wx).abs().max().item() <= 2e-3 * wx.abs().max().item()
        assert (rec["gy"] - wy).abs().max().item() <= 2e-3 * wy.abs().max().item()


def test_two_rank_training_with_gathered_embeddings(lib, dev, tmp_path):
    """trainer.gather_embeddings=true at world size 2: three pretraining steps with the global-batch loss; replicas stay
    bit-identical, and both ranks log the SAME loss terms (the gathered loss is a function of the global batch)."""
    import json
    from test_training_gpu import SMALL
    overrides = SMALL + ["trainer.max_steps=3", f"trainer.out_dir={tmp_path}", "trainer.cuda_graph=false",
                         "trainer.gather_embeddings=true"]
    _spawn_two("mp_trainer_child.py", {"IAS_MP_OVERRIDES": json.dumps(overrides)}, tmp_path)
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert recs[0]["initial_digest"] == recs[1]["initial_digest"] and len(recs[0]["steps"]) == 3
    for s0, s1 in zip(recs[0]["steps"], recs[1]["steps"]):
        assert s0["digest"] == s1["digest"], f"replicas diverged at step {s0['step']}"
        for k in s0["local"]:
            assert abs(s0["local"][k] - s1["local"][k]) <= 1e-5 * max(1.0, abs(s0["local"][k])), (k, s0["local"], s1["local"])
    assert recs[0]["steps"][0]["digest"] != recs[0]["steps"][2]["digest"]


def test_captured_ddp_step_equals_the_eager_loop_on_a_one_rank_rccl_group(lib, dev, tmp_path, nccl_group):
    """trainer.cuda_graph with the gradient all-reduce INSIDE the captured graph (round 5; reference: Lightning strategy
    "ddp" around the training step, pretrain.py:91-118, conf/config.yaml:6-8): on a one-rank RCCL group with
    GradBucketer(always_reduce=True) the bucket all-reduces are real RCCL launches in the step.  Six steps (three eager
    warm-up steps, the capture, replays) of the same small run, captured against eager: the logged losses, the learning
    rates and the final parameters agree bit for bit, and the captured run did go through the collective path."""
    import os
    from conftest import ROOT
    from inverse_audio_synthesis_amd import dist as ias_dist
    from inverse_audio_synthesis_amd.config import load_config
    from inverse_audio_synthesis_amd.harness import VicregAudioParams
    from inverse_audio_synthesis_amd.trainer import Trainer
    small = ["vicreg=fast", "vicreg.batch_size=4", "dim=64", "embeddim=256",
             "vicreg.mlp=128-128-%d", "trainer.log_every=1", "vicreg.checkpoint_every_nbatches=null", "param_embed.dropout=0.0",
             "trainer.bucket_mb=1"]
    results = {}
    for mode in ("false", "true"):
        cfg = load_config(os.path.join(ROOT, "conf"), "config", small + [f"trainer.cuda_graph={mode}",
                                                                       f"trainer.out_dir={tmp_path / mode}"])
        torch.manual_seed(42)
        m = VicregAudioParams(cfg)
        tr = Trainer(cfg, m, stage="vicreg", device=dev)
        tr.bucketer.close()
        tr.bucketer = ias_dist.GradBucketer(m, bucket_bytes=1 << 20, always_reduce=True)
        assert tr.bucketer.collective and len(tr.bucketer.buckets) > 1
        assert tr._use_graph() == (mode == "true")
        hist = tr.fit(max_steps=6)
        if mode == "true":
            assert getattr(tr, "_graph", None) is not None and not getattr(tr, "_graph_failed", False)
        results[mode] = (hist, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    he, hg = results["false"][0], results["true"][0]
    assert len(he) == len(hg) == 6
    for a, b in zip(he, hg):
        assert a["lr"] == b["lr"]
        for k in ("vicreg/train/loss", "vicreg/train/repr_loss", "vicreg/train/std_loss", "vicreg/train/cov_loss"):
            assert a[k] == b[k], (k, a, b)
    assert he[0]["vicreg/train/loss"] != he[-1]["vicreg/train/loss"]
    for k, v in results["false"][1].items():
        assert torch.equal(v, results["true"][1][k]), k


def test_gathered_loss_step_captures_with_its_rccl_collectives(lib, dev, nccl_group):
    """What `bench.py --gpus N`'s legs.vicreg_gather replays at N > 1: global_batch_loss forward + backward -- the RCCL
    all_gather_into_tensor and reduce_scatter_tensor included -- captured into one hipGraph (capture mode thread_local, as the
    Trainer and bench.py use it) on the one-rank RCCL group with gather="always", K steps per graph, against the eager
    steps: same four numbers, same gradients, bit for bit."""
    from inverse_audio_synthesis_amd.vicreg import global_batch_loss
    B, D, K = 128, 1024, 3
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(5)).to(dev).requires_grad_()
    y = torch.randn(B, D, generator=torch.Generator().manual_seed(6)).to(dev).requires_grad_()
    state = {}

    def step():
        out = global_batch_loss(x, y, B, 25.0, 25.0, 1.0, gather="always")
        gx, gy = torch.autograd.grad(out[0], (x, y))
        state["out"], state["g"] = [o.detach() for o in out], (gx, gy)

    step(); step()
    torch.cuda.synchronize()
    eager = ([float(o) for o in state["out"]], state["g"][0].clone(), state["g"][1].clone())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        for _ in range(K):
            step()
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert [float(o) for o in state["out"]] == eager[0]
    assert torch.equal(state["g"][0], eager[1]) and torch.equal(state["g"][1], eager[2])
