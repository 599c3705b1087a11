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
