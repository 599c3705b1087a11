import torch


def randn(shape, seed):
    """Seeded CPU randn (same generator the golden script uses)."""
    return torch.randn(shape, generator=torch.Generator(device="cpu").manual_seed(seed))


def checks(t):
    d = t.double()
    return [d.sum().item(), d.abs().sum().item(), (d * d).sum().item()]


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
