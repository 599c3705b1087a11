"""ParamEmbed -- drop-in for /root/reference/paramembed.py:5-40.

Linear(nparams, dim) -> norm -> dropout -> ReLU -> Linear(dim, dim) -> norm -> dropout -> ReLU ->
Linear(dim, dim); ``hidden_norm`` is the string "nn.BatchNorm1d" or "nn.Identity" (anything else
asserts, paramembed.py:13-18).  Submodule names (lin1, norm1, do1, lin2, norm2, do2, lin3) are kept so
state_dicts interchange.  Tiny GEMMs: stays on torch.nn (rocBLAS), SURVEY.md section 8(b).
"""
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

_NORMS = {"nn.BatchNorm1d": nn.BatchNorm1d, "nn.Identity": lambda _dim: nn.Identity()}


def _hidden_norm(kind, dim):
    assert kind in _NORMS, f"hidden_norm must be one of {sorted(_NORMS)}"
    return _NORMS[kind](dim)


def linear_norm(lin, norm, x):
    """``norm(lin(x))``.  Training on the GPU with nn.BatchNorm1d: the GEMM without its bias + ONE launch that adds the bias,
    takes the batch statistics, updates the running ones and their counter and normalises (vicreg._BN1dGroupsFn with one
    row group; its backward is one launch that also returns the Linear bias' gradient) instead of torch's counter add,
    statistics, finalize and transform launches and a column-sum launch for the bias gradient."""
    from .vicreg import _BN1dGroupsFn, _bn1d_hip_ok
    if x.dim() == 2 and _bn1d_hip_ok(norm, x, x.shape[0]):
        return _BN1dGroupsFn.apply(F.linear(x, lin.weight), lin.bias, norm.weight, norm.bias, norm.running_mean,
                                   norm.running_var, norm.num_batches_tracked, norm.eps, norm.momentum, 1, False)
    return norm(lin(x))


class ParamEmbed(nn.Module):
    def __init__(self, nparams, dim, hidden_norm, dropout):
        super().__init__()
        self.nparams, self.dim = nparams, dim
        self.relu = nn.ReLU()
        self.lin1 = nn.Linear(nparams, dim)
        self.norm1 = _hidden_norm(hidden_norm, dim)
        self.do1 = nn.Dropout(dropout)
        self.lin2 = nn.Linear(dim, dim)
        self.norm2 = _hidden_norm(hidden_norm, dim)
        self.do2 = nn.Dropout(dropout)
        self.lin3 = nn.Linear(dim, dim)

    def forward(self, x: Tensor) -> Tensor:
        h = self.relu(self.do1(linear_norm(self.lin1, self.norm1, x)))
        h = self.relu(self.do2(linear_norm(self.lin2, self.norm2, h)))
        return self.lin3(h)


class AudioRepresentationToParams(nn.Module):
    """/root/reference/audio_to_params.py:16-53: the same 3-layer MLP, dim -> nparams, sigmoid output."""

    def __init__(self, nparams, dim, hidden_norm, dropout):
        super().__init__()
        self.nparams, self.dim = nparams, dim
        self.relu = nn.ReLU()
        self.lin1 = nn.Linear(dim, dim)
        self.norm1 = _hidden_norm(hidden_norm, dim)
        self.do1 = nn.Dropout(dropout)
        self.lin2 = nn.Linear(dim, dim)
        self.norm2 = _hidden_norm(hidden_norm, dim)
        self.do2 = nn.Dropout(dropout)
        self.lin3 = nn.Linear(dim, nparams)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x: Tensor) -> Tensor:
        h = self.relu(self.do1(linear_norm(self.lin1, self.norm1, x)))
        h = self.relu(self.do2(linear_norm(self.lin2, self.norm2, h)))
        return self.sigmoid(self.lin3(h))
