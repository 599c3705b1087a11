"""AudioEmbedding -- drop-in for /root/reference/audioembed.py:5-72.

audio [B,1,176400] -> PQMF(3) [B,3,58800] -> reshape [B,3,240,245] -> per-channel normalise ->
``vision_model.features`` [B,576,8,8] -> conv7..conv1 (kernel 2, 8->1 spatial) -> [B,dim].
Same constructor, submodule names (conv1..conv7) and methods (``_preprocess``, ``forward``, ``features``).
When ``gram`` is this package's PQMF and ``img_preprocess`` is a ``ChannelNormalize``, the PQMF, the
reshape (a pure view) and the normalisation run as ONE HIP kernel (the fused epilogue of
csrc/pqmf_kernels.hip) instead of a 90 MB intermediate read + write.
"""
import torch
import torch.nn as nn

from .pqmf import PQMF, pqmf_analysis
from .vision import trunk_torch


class ChannelNormalize(nn.Module):
    """(x - mean_c) / std_c on [B,C,H,W] -- what torchvision.transforms.Normalize does
    (reference vicreg_audio_params.py:60-62 with ImageNet constants)."""

    def __init__(self, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
        super().__init__()
        self.register_buffer("mean", torch.tensor(mean, dtype=torch.float32), persistent=False)
        self.register_buffer("std", torch.tensor(std, dtype=torch.float32), persistent=False)

    def forward(self, x):
        return (x - self.mean.view(1, -1, 1, 1)) / self.std.view(1, -1, 1, 1)


class _Conv2x2Fn(torch.autograd.Function):
    """out[(b,i,j), :] = patches[(b,i,j), (c,di,dj)] @ weight.view(Cout, 4 Cin)^T + bias, patches gathered by
    ``ias_conv2x2_patches`` (csrc/conv_kernels.hip); backward: two GEMMs, the patch adjoint in one kernel, the weight
    gradient directly in the weight's layout."""

    @staticmethod
    def forward(ctx, x, weight, bias, nchw=False):
        from . import _lib
        lib = _lib.load()
        x, weight = x.contiguous(), weight.contiguous()
        _lib.require_f32(x, weight)
        if nchw:
            B, C, H, W = x.shape
        else:
            B, H, W, C = x.shape
        Cout = weight.shape[0]
        patches = torch.empty((B * (H - 1) * (W - 1), 4 * C), dtype=torch.float32, device=x.device)
        if nchw:
            _lib.check(lib.ias_conv2x2_patches_nchw(_lib.ptr(x), _lib.ptr(patches), B, H, W, C, _lib.stream()),
                       "ias_conv2x2_patches_nchw")
        else:
            _lib.check(lib.ias_conv2x2_patches(_lib.ptr(x), _lib.ptr(patches), B, H, W, C, _lib.stream()), "ias_conv2x2_patches")
        w2 = weight.view(Cout, 4 * C)
        out = torch.addmm(bias, patches, w2.t()) if bias is not None else torch.mm(patches, w2.t())
        ctx.save_for_backward(patches, weight)
        ctx.shape = (B, H, W, C)
        ctx.nchw = bool(nchw)
        ctx.has_bias = bias is not None
        return out.view(B, H - 1, W - 1, Cout)

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        patches, weight = ctx.saved_tensors
        B, H, W, C = ctx.shape
        Cout = weight.shape[0]
        g2 = g.reshape(-1, Cout)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gp = torch.mm(g2, weight.view(Cout, 4 * C))
            if ctx.nchw:
                gx = torch.empty((B, C, H, W), dtype=torch.float32, device=g.device)
                _lib.check(lib.ias_conv2x2_patches_backward_nchw(_lib.ptr(gp), _lib.ptr(gx), B, H, W, C, _lib.stream()),
                           "ias_conv2x2_patches_backward_nchw")
            else:
                gx = torch.empty((B, H, W, C), dtype=torch.float32, device=g.device)
                _lib.check(lib.ias_conv2x2_patches_backward(_lib.ptr(gp), _lib.ptr(gx), B, H, W, C, _lib.stream()),
                           "ias_conv2x2_patches_backward")
        if ctx.needs_input_grad[1]:
            gw = torch.mm(g2.t(), patches).view_as(weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            g2c = g2.contiguous()
            gb = torch.empty(Cout, dtype=torch.float32, device=g.device)
            scratch = torch.empty(int(lib.ias_colsum_scratch_floats(g2c.shape[0], Cout)), dtype=torch.float32, device=g.device)
            from . import vision
            if vision._DEFER["on"]:
                # the second launch joins the backward pass' ONE reduction launch (vision.defer_weight_reductions)
                srows = lib.ias_colsum_partials(_lib.ptr(g2c), _lib.ptr(scratch), g2c.shape[0], Cout, _lib.stream())
                _lib.check(min(int(srows), 0), "ias_colsum_partials")
                vision._defer_reduction(scratch, gb, Cout, srows)
            else:
                _lib.check(lib.ias_colsum(_lib.ptr(g2c), _lib.ptr(gb), _lib.ptr(scratch), g2c.shape[0], Cout, _lib.stream()),
                           "ias_colsum")
        return gx, gw, gb, None


def conv2x2_nhwc(x, weight, bias):
    """``nn.Conv2d(kernel_size=2)`` (stride 1, no padding) on a channels-last activation x [B,H,W,Cin] with the module's
    own parameters (weight [Cout,Cin,2,2], bias [Cout]) -> [B,H-1,W-1,Cout], as ONE gather + ONE dense GEMM.  The seven
    layers of AudioEmbedding's head (/root/reference/audioembed.py:15-33,62-68) are 1024 -> 1024 convolutions on
    8x8 ... 2x2 maps: as nn.Conv2d on MIOpen they ran as naive fallbacks plus one im2col + one small GEMM PER SAMPLE
    (4096 Im2d2Col launches per step at batch 128); as a GEMM [B Ho Wo, 4 Cin] x [4 Cin, Cout] they are seven rocBLAS /
    hipBLASLt calls.  fp32 throughout.  The patch columns are ordered (c, di, dj) -- the weight's own memory order -- so
    neither the weight nor its gradient is ever permuted."""
    return _Conv2x2Fn.apply(x, weight, bias)


def conv2x2_from_nchw(x, weight, bias):
    """The same layer on an NCHW activation x [B,Cin,H,W] (H W <= 255) -> channels-last [B,H-1,W-1,Cout]: the patches are
    gathered straight from the planes (``ias_conv2x2_patches_nchw``), the input gradient comes back NCHW -- the first
    head layer needs no permuted copy of the trunk's output in either direction."""
    return _Conv2x2Fn.apply(x, weight, bias, True)


class AudioEmbedding(nn.Module):
    def __init__(self, gram, vision_model, img_preprocess, dim):
        super().__init__()
        self.gram = gram
        self.vision_model = vision_model
        self.img_preprocess = img_preprocess
        self.dim = dim
        self.conv7 = nn.Conv2d(in_channels=576, out_channels=dim, kernel_size=2)
        for i in range(6, 0, -1):
            setattr(self, f"conv{i}", nn.Conv2d(in_channels=dim, out_channels=dim, kernel_size=2))

    def _preprocess(self, audio):
        if isinstance(self.gram, PQMF) and isinstance(self.img_preprocess, ChannelNormalize) and self.gram.N == 3:
            z = pqmf_analysis(audio, self.gram.H, self.img_preprocess.mean, self.img_preprocess.std)
            return z.reshape(-1, 3, 240, 245)
        return self.img_preprocess(self.gram(audio).reshape(-1, 3, 240, 245))

    def forward(self, audio):
        t = self.vision_model.features(self._preprocess(audio))
        def plain2x2(c):
            return (c.kernel_size == (2, 2) and c.stride == (1, 1) and c.padding == (0, 0) and c.dilation == (1, 1)
                    and c.groups == 1 and c.padding_mode == "zeros")
        # vision.FORCE_TORCH_LAYERS (diagnostics): the nn.Conv2d head; the trunk layers honour the same attribute
        if t.is_cuda and t.dtype == torch.float32 and not trunk_torch() and \
                all(plain2x2(getattr(self, f"conv{i}")) for i in range(1, 8)):
            for i in range(7, 0, -1):                 # the head is channels-last from its first output on
                c = getattr(self, f"conv{i}")
                if i != 7:
                    t = conv2x2_nhwc(t, c.weight, c.bias)
                elif t.shape[2] * t.shape[3] <= 255:  # the trunk's output is NCHW
                    t = conv2x2_from_nchw(t, c.weight, c.bias)
                else:
                    t = conv2x2_nhwc(t.permute(0, 2, 3, 1), c.weight, c.bias)
            return t.reshape(-1, self.dim)            # [B,1,1,dim]
        for i in range(7, 0, -1):
            t = getattr(self, f"conv{i}")(t)
        return t.view(-1, self.dim)

    def features(self, audio):
        return self.forward(audio)
