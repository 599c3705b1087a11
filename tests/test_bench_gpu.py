"""bench.py's N > 1 control flow on the one-GPU box: `python bench.py --gpus 2` spawns its two ranks itself (fresh child
processes, before anything touches the GPU), they share the GPU over gloo (IAS_BENCH_BACKEND=gloo: the timing collectives
on CPU tensors) and rank 0 prints ONE JSON line.  The only rehearsal of the driver's SCALE command this pool allows
(/root/reference/conf/config.yaml:8 strategy ddp -> pretrain.py:97-99: one process per GPU)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run_bench(extra, timeout=900, legs=False, torchrun=False):
    env = dict(os.environ, IAS_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    head = [sys.executable, os.path.join(ROOT, "bench.py")]
    if torchrun:        # the driver's launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), os.path.join(ROOT, "bench.py")]
    r = subprocess.run(head + ["--gpus", "2", "--steps", "5", "--warmup", "2",
                               "--no-cpu-baseline", "--replays", "5"] + ([] if legs else ["--no-legs"]) + extra, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 alone prints, once
    return json.loads(lines[0])


def test_bench_synth_two_ranks(lib, dev):
    j = _run_bench([])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["scaling"] == "weak" and j["unit"] == "audio-s/s"
    B = j["config"]["batch_per_gpu"]
    assert B == 128
    per_step = 2 * B * 4.0 / (j["ms_per_step"] * 1e-3)        # whole-job value: both ranks' batches over the max-over-ranks time
    assert abs(j["value"] - per_step) <= 1e-3 * per_step
    assert "cpu_baseline" not in j and "legs" not in j          # N = 1 only
    assert j["roofline"]["frac"] > 0


def test_bench_vicreg_two_ranks(lib, dev):
    j = _run_bench(["--workload", "vicreg"])
    assert j["n_gpus"] == 2 and j["config"]["gather"] is True and j["config"]["rccl_world_size"] == 2
    assert j["config"]["batch_per_gpu"] == 128 and "global batch 256" in j["config"]["workload"]
    per_step = 2 * 128 / (j["ms_per_step"] * 1e-3)
    assert abs(j["value"] - per_step) <= 1e-3 * per_step
    # the loss both ranks computed is the global-batch one: cov term with denominator 255, not 127 (~4x larger)
    import torch
    from oracle import vicreg_oracle as vo
    x = torch.cat([torch.randn(128, 8192, generator=torch.Generator().manual_seed(2 * r)) for r in range(2)])
    y = torch.cat([torch.randn(128, 8192, generator=torch.Generator().manual_seed(2 * r + 1)) for r in range(2)])
    ref = vo.loss(x, y, 256, 8192)
    assert abs(j["config"]["cov_loss"] - ref[3].item()) <= 2e-3 * ref[3].item()
    assert abs(j["config"]["loss"] - ref[0].item()) <= 2e-3 * ref[0].item()


def test_bench_two_ranks_run_the_collective_legs(lib, dev):
    """The driver's SCALE command at N = 2, rehearsed over gloo on the one GPU: after the headline regions every rank starts
    a fresh child process (never an exec of a GPU-initialised one) on a process group of its own, and rank 0's line carries
    legs.vicreg_gather (configs[3]: the all-gather / reduce-scatter of vicreg.py:79-95 around the global-batch loss) and
    legs.pretrain_ddp (conf/config.yaml:6-8 strategy ddp: the pretraining step with the bucketed gradient all-reduce, next
    to the same step without it).  The headline fields are those of the N = 2 run without legs."""
    j = _run_bench([], timeout=1500, legs=True)
    assert j["n_gpus"] == 2 and j["unit"] == "audio-s/s" and j["config"]["batch_per_gpu"] == 128
    legs = j["legs"]
    g, d = legs["vicreg_gather"], legs["pretrain_ddp"]
    assert "error" not in g, g
    assert "error" not in d, d
    assert g["rccl_world_size"] == 2 and "all_gather" in g["collective"] and g["backend"] == "gloo"
    assert g["ms_per_step"] > 0 and "global batch 256" in g["workload"]
    import torch
    from oracle import vicreg_oracle as vo
    x = torch.cat([torch.randn(128, 8192, generator=torch.Generator().manual_seed(2 * r)) for r in range(2)])
    y = torch.cat([torch.randn(128, 8192, generator=torch.Generator().manual_seed(2 * r + 1)) for r in range(2)])
    ref = vo.loss(x, y, 256, 8192)
    assert abs(g["cov_loss"] - ref[3].item()) <= 2e-3 * ref[3].item()
    assert d["rccl_world_size"] == 2 and d["collective"] is True and d["buckets"] >= 2
    assert d["gradient_bytes"] > 500e6                                    # ~143 M fp32 parameters
    assert d["ms_per_step_eager"] > 0 and d["ms_per_step_no_allreduce"] > 0
    assert d["allreduce_exposed_ms"] == pytest.approx(d["ms_per_step_eager"] - d["ms_per_step_no_allreduce"], abs=2e-3)
    import math
    assert math.isfinite(d["loss"])
    gs = legs["gradstep_all_ranks"]            # configs[4] batch-split over the ranks, no collective
    assert "error" not in gs, gs
    assert gs["global_batch"] == 128 and gs["n_gpus"] == 2 and gs["ms_per_step"] > 0
    assert abs(gs["value"] - 2 * 64 * 4.0 / (gs["ms_per_step"] * 1e-3)) <= 1e-3 * gs["value"]
