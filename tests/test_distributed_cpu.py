"""world_size-2 gloo tests (CPU) of the N>1 path: FullGatherLayer, bucketed gradient averaging,
metric reduction.  Spawned with torch.multiprocessing on 127.0.0.1."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, ret):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = globals()[fn_name](rank, world)
    finally:
        dist.destroy_process_group()


def _spawn(fn_name, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn_name, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


# ---- bodies run inside each rank -------------------------------------------------------------
def _body_full_gather(rank, world):
    from inverse_audio_synthesis_amd.vicreg import FullGatherLayer
    from oracle import vicreg_oracle as vo
    local = [torch.randn(3, 5, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
    x = local[rank].clone().requires_grad_()
    gathered = FullGatherLayer.apply(x)
    ref = vo.full_gather_forward(local)
    ok_fwd = all(torch.equal(a, b) for a, b in zip(gathered, ref))
    # a loss that weights each gathered block differently on each rank
    wts = [torch.randn(3, 5, generator=torch.Generator().manual_seed(100 * rank + r)) for r in range(world)]
    sum((g * w).sum() for g, w in zip(gathered, wts)).backward()
    per_rank = [tuple(torch.randn(3, 5, generator=torch.Generator().manual_seed(100 * rr + r)) for r in range(world))
                for rr in range(world)]
    want = vo.full_gather_backward(per_rank, rank)
    return bool(ok_fwd and torch.allclose(x.grad, want, atol=1e-6))


def _body_gather_rows(rank, world):
    """gather_rows(x) = cat(FullGatherLayer.apply(x), 0) in value and gradient (the one collective of the global-batch
    loss, inverse-audio-synthesis_amd/vicreg.py:global_batch_loss; reference /root/reference/vicreg.py:38-39,79-95)."""
    from inverse_audio_synthesis_amd.vicreg import FullGatherLayer, gather_rows
    from oracle import vicreg_oracle as vo
    local = [torch.randn(3, 6, generator=torch.Generator().manual_seed(20 + r)) for r in range(world)]
    x = local[rank].clone().requires_grad_()
    got = gather_rows(x)
    ok = torch.equal(got, torch.cat(vo.full_gather_forward(local), 0))
    wts = [torch.randn(world * 3, 6, generator=torch.Generator().manual_seed(300 + rr)) for rr in range(world)]
    (got * wts[rank]).sum().backward()
    want = vo.full_gather_backward([tuple(wts[rr][q * 3:(q + 1) * 3] for q in range(world)) for rr in range(world)], rank)
    ok = ok and torch.allclose(x.grad, want, atol=1e-6)
    x2 = local[rank].clone().requires_grad_()
    (torch.cat(FullGatherLayer.apply(x2), 0) * wts[rank]).sum().backward()
    return bool(ok and torch.allclose(x.grad, x2.grad, atol=1e-6))


def _body_grad_bucketer(rank, world):
    from inverse_audio_synthesis_amd.dist import GradBucketer, all_reduce_mean
    torch.manual_seed(0)  # same init on every rank
    H = 2800              # 6 x 2800 weights: above GradBucketer.SMALL (own copy launch); the rest go in by the joint copy
    model = torch.nn.Sequential(torch.nn.Linear(6, H), torch.nn.ReLU(), torch.nn.Linear(H, 4),
                                torch.nn.Linear(4, 3))
    bucketer = GradBucketer(model, bucket_bytes=300)  # tiny buckets: several collectives
    assert len(bucketer.buckets) >= 3
    assert model[0].weight.numel() >= GradBucketer.SMALL > model[2].weight.numel()
    data = [torch.randn(5, 6, generator=torch.Generator().manual_seed(50 + r)) for r in range(world)]
    ok = True
    for step in range(3):
        bucketer.begin_step()
        ok = ok and all(p.grad is None for p in model.parameters())
        nback = 1
        model[2](model[1](model[0](data[rank]))).pow(2).sum().backward()   # model[3] unused
        bucketer.finish()
        # reference: mean over ranks of the single-process gradients
        ref_model = torch.nn.Sequential(torch.nn.Linear(6, H), torch.nn.ReLU(), torch.nn.Linear(H, 4))
        ref_model.load_state_dict({k: v for k, v in model.state_dict().items() if not k.startswith("3.")})
        grads = None
        for r in range(world):
            ref_model.zero_grad()
            ref_model(data[r]).pow(2).sum().backward()
            g = [p.grad.clone() for p in ref_model.parameters()]
            grads = g if grads is None else [a + b for a, b in zip(grads, g)]
        for p, g in zip(list(model.parameters())[:4], grads):
            ok = ok and torch.allclose(p.grad, nback * g / world, rtol=1e-5, atol=1e-6)
        # a parameter without a gradient takes part as zeros; every gradient lives in its bucket after the step
        ok = ok and all(float(p.grad.abs().sum()) == 0.0 for p in model[3].parameters())
        for flat, plist in bucketer.buckets:
            lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
            ok = ok and all(lo <= p.grad.data_ptr() < hi for p in plist)
    # one backward per step: a bucket goes out as soon as its gradients of ONE backward are in
    bucketer.begin_step()
    model[2](model[1](model[0](data[rank]))).pow(2).sum().backward()
    try:
        model[2](model[1](model[0](data[rank]))).pow(2).sum().backward()
        ok = False
    except RuntimeError as ex:
        ok = ok and "second backward" in str(ex)
    bucketer.finish()
    m = all_reduce_mean(torch.tensor(float(rank + 1)))
    return bool(ok and abs(m.item() - (world + 1) / 2) < 1e-6)


# ---- tests ------------------------------------------------------------------------------------
def test_full_gather_layer_two_ranks():
    assert _spawn("_body_full_gather") == [True, True]


def test_gather_rows_two_ranks():
    assert _spawn("_body_gather_rows") == [True, True]


def test_grad_bucketer_two_ranks():
    assert _spawn("_body_grad_bucketer") == [True, True]


def test_single_process_is_identity():
    from inverse_audio_synthesis_amd.dist import GradBucketer, all_reduce_mean, world_size
    assert world_size() == 1
    model = torch.nn.Linear(3, 2)
    b = GradBucketer(model)
    b.begin_step()
    model(torch.ones(4, 3)).sum().backward()
    b.finish()
    assert torch.allclose(model.weight.grad, torch.full((2, 3), 4.0))
    assert all_reduce_mean(torch.tensor(3.0)).item() == 3.0
