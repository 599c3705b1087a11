"""CPU oracle for the VICReg loss (TEST INFRASTRUCTURE -- never imported by the product).

Restates /root/reference/vicreg.py:35-58 (loss), :73-76 (off_diagonal), :79-95
(FullGatherLayer semantics: all_gather forward, summed-gradient slice backward).
Pinned against the reference by tests/golden/vicreg_loss.npz.
"""
import torch
import torch.nn.functional as F


def off_diagonal(x):
    n, m = x.shape
    assert n == m
    return x.flatten()[:-1].view(n - 1, n + 1)[:, 1:].flatten()


def loss(x, y, cfg_batch_size, embeddim, sim_coeff=25.0, std_coeff=25.0, cov_coeff=1.0):
    """-> (loss, repr_loss, std_loss, cov_loss); the covariance denominator is the
    *configured* batch size minus one (vicreg.py:47-48), not x.shape[0]."""
    repr_loss = F.mse_loss(x, y)
    x = x - x.mean(dim=0)
    y = y - y.mean(dim=0)
    std_x = torch.sqrt(x.var(dim=0) + 0.0001)
    std_y = torch.sqrt(y.var(dim=0) + 0.0001)
    std_loss = torch.mean(F.relu(1 - std_x)) / 2 + torch.mean(F.relu(1 - std_y)) / 2
    cov_x = (x.T @ x) / (cfg_batch_size - 1)
    cov_y = (y.T @ y) / (cfg_batch_size - 1)
    cov_loss = off_diagonal(cov_x).pow(2).sum().div(embeddim) + off_diagonal(cov_y).pow(2).sum().div(embeddim)
    total = sim_coeff * repr_loss + std_coeff * std_loss + cov_coeff * cov_loss
    return total, repr_loss, std_loss, cov_loss


def full_gather_forward(local_tensors):
    """What FullGatherLayer.forward returns on every rank: the tuple of all ranks' tensors."""
    return tuple(t.clone() for t in local_tensors)


def full_gather_backward(per_rank_grads, rank):
    """per_rank_grads[r] = tuple of W grads rank r received; -> grad of rank `rank`'s input
    (vicreg.py:92-95: stack, all_reduce(sum), take [rank])."""
    return sum(torch.stack(g)[rank] for g in per_rank_grads)
