import torch


def randn(shape, seed):
    """Seeded CPU randn (same generator the golden script uses)."""
    return torch.randn(shape, generator=torch.Generator(device="cpu").manual_seed(seed))


def checks(t):
    d = t.double()
    return [d.sum().item(), d.abs().sum().item(), (d * d).sum().item()]


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def vicreg_backward_closed_form(x, y, gcoef, cfg_batch, sim, std, cov):
    """The closed-form gradient of the VICReg loss (/root/reference/vicreg.py:35-58) as plain fp32 torch ops on the inputs'
    device: a second definition the HIP backward is tested against, besides autograd through the oracle (test
    infrastructure: the package does not use it).  gcoef: the cotangents of (loss, repr, std, cov)."""
    B, D = x.shape
    a = gcoef[0] * sim + gcoef[1]
    b = gcoef[0] * std + gcoef[2]
    c = gcoef[0] * cov + gcoef[3]
    d_repr = (x - y) * (2.0 / (B * D))

    def branch(v):
        vc = v - v.mean(dim=0)
        m2 = (vc * vc).sum(dim=0)
        s = torch.sqrt(m2 / (B - 1) + 0.0001)
        d_std = -(s < 1).to(v.dtype) / (2.0 * D * (B - 1) * s) * vc
        gram = vc @ vc.T
        d_cov = (gram @ vc - vc * m2) * (4.0 / ((cfg_batch - 1) ** 2 * D))
        return b * d_std + c * d_cov

    return a * d_repr + branch(x), -a * d_repr + branch(y)
