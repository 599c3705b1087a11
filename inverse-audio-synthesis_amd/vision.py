"""MobileNetV3-small feature trunk (the ``vision_model.features`` AudioEmbedding calls).

The reference takes it from torchvision (/root/reference/vicreg_audio_params.py:52-54,
``mobilenet_v3_small(pretrained=...)``); torchvision is not in this image and there is no network, so
the architecture (Howard et al. 2019, "small" table) is defined here with torchvision's module layout
-- ``features.N``, ``.block``, ``fc1/fc2`` -- so torchvision state_dicts load by key.  Weights are
random-initialised; ``pretrained=True`` only warns.  On a ROCm device in fp32 the depthwise, stem and thin 1x1
convolutions, BatchNorm + activation and the squeeze-excitation blocks run on this repo's HIP kernels
(csrc/conv_kernels.hip, pointwise_kernels.hip, bn_kernels.hip, se_kernels.hip), the wide 1x1 convolutions as
rocBLAS / hipBLASLt GEMMs; everything else (CPU tensors, other dtypes) takes the torch.nn layers these modules subclass
(the trunk is a caller of the hot path, not part of it).
"""
import os
import ctypes
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

# Diagnostics only (scripts/diag set these attributes; nothing in the package or the environment does): every layer of the
# trunk AND the conv head of audioembed.py on their torch.nn parents (MIOpen / rocBLAS) / the thin 1x1 convolutions as
# batched GEMMs -- for A/B timing against the HIP kernels.
FORCE_TORCH_LAYERS = False
FORCE_PW_BMM = False


def trunk_torch():
    return FORCE_TORCH_LAYERS


# ---- the weight-gradient reductions of a backward pass, deferred to ONE launch at its end ---------------------------------
# Every weight gradient of the trunk (thin 1x1, depthwise, stem) ends in a reduction of per-workgroup partial rows: a launch
# of 5-6 us, all latency, 27 of them on the dependency chain of a pretraining step's backward.  With
# ``defer_weight_reductions(True)`` the backward functions below leave their partial rows behind
# (``ias_*_backward_weight_partials``), hand autograd the still unwritten gradient tensor, and ONE
# ``ias_reduce_partials_multi`` launch fills all of them from a callback the autograd engine runs when the backward pass is
# over (``queue_callback``: before ``backward()`` / ``autograd.grad()`` returns, also inside a hipGraph capture).  Nothing
# may READ a weight gradient during the backward pass, then: a post-accumulate-grad hook would (dist.GradBucketer on
# several ranks copies gradients into its buckets from one) -- so this is a switch the owner of the training loop sets
# (Trainer: on exactly when no such hooks are installed), off by default.
_DEFER = {"on": False, "items": []}


def defer_weight_reductions(on):
    """-> the previous setting."""
    old, _DEFER["on"] = _DEFER["on"], bool(on)
    return old


def _defer_reduction(scratch, gw, n, rows):
    # (the gradient's STORAGE is kept alive, not the tensor: with another reference to the tensor around, autograd's
    # AccumulateGrad would not adopt it as the parameter's .grad but clone it on the spot -- the still unwritten bytes)
    _DEFER["items"].append((scratch, gw.untyped_storage(), gw.data_ptr(), gw.device, int(n), int(rows)))
    # (one callback per item: the first one of a pass reduces everything queued so far, the others find nothing -- no flag
    # that a failed backward pass could leave set)
    torch.autograd.Variable._execution_engine.queue_callback(_flush_reductions)


class _ReduceItem(ctypes.Structure):      # IasReduceItem (include/ias_hip.h)
    _fields_ = [("partial", ctypes.c_void_p), ("out", ctypes.c_void_p), ("n", ctypes.c_int), ("rows", ctypes.c_int)]


def _flush_reductions():
    items, _DEFER["items"] = _DEFER["items"], []
    if not items:
        return
    from . import _lib
    lib = _lib.load()
    # a HOST table: the library passes it to the kernel by value (no device table, no staging copy: nothing is allocated
    # here, which matters inside a hipGraph capture -- pinned memory must not be allocated while one is open)
    table = (_ReduceItem * len(items))()
    for t, (s, _st, gp, _dev, n, r) in zip(table, items):
        t.partial, t.out, t.n, t.rows = s.data_ptr(), gp, n, r
    _lib.check(lib.ias_reduce_partials_multi(ctypes.cast(table, ctypes.c_void_p), len(items), _lib.stream()),
               "ias_reduce_partials_multi")


_PW_SUPPORTED = {}


def _pw_mfma(cin, cout):
    """Whether csrc/pointwise_kernels.hip takes this 1x1 convolution (ias_pwconv_supported)."""
    if FORCE_PW_BMM:
        return False
    key = (cin, cout)
    if key not in _PW_SUPPORTED:
        from . import _lib
        _PW_SUPPORTED[key] = bool(_lib.load().ias_pwconv_supported(cin, cout))
    return _PW_SUPPORTED[key]


class _PointwiseFn(torch.autograd.Function):
    """csrc/pointwise_kernels.hip: a thin 1x1 convolution (no bias) on fp32 MFMA -- forward, input gradient, weight
    gradient (deterministic reduction)."""

    @staticmethod
    def forward(ctx, x, w):
        from . import _lib
        lib = _lib.load()
        x, w = x.contiguous(), w.contiguous()
        _lib.require_f32(x, w)
        B, C, H, W = x.shape
        Cout = w.shape[0]
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
        _lib.check(lib.ias_pwconv_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), B, C, Cout, H * W, _lib.stream()),
                   "ias_pwconv_forward")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        x, w = ctx.saved_tensors
        g = g.contiguous()
        B, C, H, W = x.shape
        Cout = w.shape[0]
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _lib.check(lib.ias_pwconv_backward_data(_lib.ptr(g), _lib.ptr(w), _lib.ptr(gx), B, C, Cout, H * W, _lib.stream()),
                       "ias_pwconv_backward_data")
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            scratch = torch.empty(int(lib.ias_pwconv_weight_scratch(B, C, Cout, H * W)), dtype=torch.float32, device=x.device)
            if _DEFER["on"]:
                rows = lib.ias_pwconv_backward_weight_partials(_lib.ptr(g), _lib.ptr(x), _lib.ptr(scratch), B, C, Cout, H * W,
                                                               _lib.stream())
                _lib.check(min(int(rows), 0), "ias_pwconv_backward_weight_partials")
                _defer_reduction(scratch, gw, gw.numel(), rows)
            else:
                _lib.check(lib.ias_pwconv_backward_weight(_lib.ptr(g), _lib.ptr(x), _lib.ptr(gw), _lib.ptr(scratch), B, C, Cout,
                                                          H * W, _lib.stream()), "ias_pwconv_backward_weight")
        return gx, gw


class PointwiseConv2d(nn.Conv2d):
    """A 1x1 convolution (same parameters and state_dict keys as nn.Conv2d).  On a ROCm device it is what it is -- ONE
    strided-batched GEMM  y[b] = W [Cout,Cin] x[b] [Cin, H W]  on rocBLAS / hipBLASLt, layout unchanged (NCHW in, NCHW
    out) -- instead of MIOpen's fp32 fallback, which ran it as an im2col plus a small GEMM PER SAMPLE (3 x 1024 launches
    per training step at batch 128 for the MobileNet body).  Elsewhere this is the plain nn.Conv2d."""

    def forward(self, x):
        if x.is_cuda and not trunk_torch() and self.kernel_size == (1, 1) and self.stride == (1, 1) and self.groups == 1 and \
                x.dim() == 4 and x.is_contiguous():
            B, C, H, W = x.shape
            if self.bias is None and x.dtype == torch.float32 and _pw_mfma(C, self.out_channels):
                return _PointwiseFn.apply(x, self.weight)      # the thin layers: weight in LDS, fp32 MFMA
            # bmm with the weight expanded along the batch (stride 0), not torch.matmul(W, x): matmul folds the batch into
            # the rows of a transposed COPY of x (and of the result, and again in backward) -- 106 copy launches and
            # 2.3 ms of a pretraining step at batch 128.  Autograd of this form is two more bmm's and one sum over the
            # batch for the weight gradient, all on the operands where they lie.
            y = torch.bmm(self.weight.view(1, self.out_channels, C).expand(B, -1, -1), x.view(B, C, H * W))
            if self.bias is not None:
                y = y + self.bias.view(1, -1, 1)
            return y.view(B, self.out_channels, H, W)
        return super().forward(x)


class _DepthwiseFn(torch.autograd.Function):
    """csrc/conv_kernels.hip: depthwise conv forward, input gradient, weight gradient (deterministic reduction)."""

    @staticmethod
    def forward(ctx, x, w, K, S):
        from . import _lib
        lib = _lib.load()
        x, w = x.contiguous(), w.contiguous()
        _lib.require_f32(x, w)
        B, C, H, W = x.shape
        Ho, Wo = lib.ias_conv_out_size(H, K, S), lib.ias_conv_out_size(W, K, S)
        out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
        _lib.check(lib.ias_dwconv_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(out), B, C, H, W, K, S, _lib.stream()),
                   "ias_dwconv_forward")
        ctx.save_for_backward(x, w)
        ctx.ks = (K, S)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        x, w = ctx.saved_tensors
        K, S = ctx.ks
        B, C, H, W = x.shape
        g = g.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _lib.check(lib.ias_dwconv_backward_data(_lib.ptr(g), _lib.ptr(w), _lib.ptr(gx), B, C, H, W, K, S, _lib.stream()),
                       "ias_dwconv_backward_data")
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            scratch = torch.empty(int(lib.ias_dwconv_weight_scratch_hw(B, C, H, W, K, S)), dtype=torch.float32, device=x.device)
            if _DEFER["on"]:
                rows = lib.ias_dwconv_backward_weight_partials(_lib.ptr(x), _lib.ptr(g), _lib.ptr(scratch), B, C, H, W, K, S,
                                                               _lib.stream())
                _lib.check(min(int(rows), 0), "ias_dwconv_backward_weight_partials")
                _defer_reduction(scratch, gw, gw.numel(), rows)
            else:
                _lib.check(lib.ias_dwconv_backward_weight(_lib.ptr(x), _lib.ptr(g), _lib.ptr(gw), _lib.ptr(scratch), B, C, H, W,
                                                          K, S, _lib.stream()), "ias_dwconv_backward_weight")
        return gx, gw, None, None


class DepthwiseConv2d(nn.Conv2d):
    """Depthwise convolution (groups = channels; same parameters / state_dict keys as nn.Conv2d).  On a ROCm device, for
    the shapes of MobileNetV3 (3x3 / 5x5, stride 1 / 2, padding (k-1)/2, no bias), it runs the stencil kernels of
    csrc/conv_kernels.hip instead of MIOpen's fp32 fallbacks (naive_conv_*, 0.85 ms Winograd calls on 15 x 16 maps);
    elsewhere it is the plain nn.Conv2d."""

    def forward(self, x):
        k, s = self.kernel_size[0], self.stride[0]
        if x.is_cuda and not trunk_torch() and x.dtype == torch.float32 and self.groups == self.in_channels == self.out_channels and \
                self.bias is None and self.kernel_size in ((3, 3), (5, 5)) and self.stride in ((1, 1), (2, 2)) and \
                self.padding == ((k - 1) // 2, (k - 1) // 2) and self.dilation == (1, 1):
            return _DepthwiseFn.apply(x, self.weight, k, s)
        return super().forward(x)


class _StemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        from . import _lib
        lib = _lib.load()
        x, w = x.contiguous(), w.contiguous()
        _lib.require_f32(x, w)
        B, _c, H, W = x.shape
        Ho, Wo = lib.ias_conv_out_size(H, 3, 2), lib.ias_conv_out_size(W, 3, 2)
        out = torch.empty((B, 16, Ho, Wo), dtype=torch.float32, device=x.device)
        _lib.check(lib.ias_stem_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(out), B, H, W, _lib.stream()), "ias_stem_forward")
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        x, w = ctx.saved_tensors
        B, _c, H, W = x.shape
        g = g.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:   # the image never requires grad in this model; kept correct through torch
            gx = torch.nn.grad.conv2d_input(x.shape, w, g, stride=2, padding=1)
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            scratch = torch.empty(int(lib.ias_stem_weight_scratch(B)), dtype=torch.float32, device=x.device)
            if _DEFER["on"]:
                rows = lib.ias_stem_backward_weight_partials(_lib.ptr(x), _lib.ptr(g), _lib.ptr(scratch), B, H, W, _lib.stream())
                _lib.check(min(int(rows), 0), "ias_stem_backward_weight_partials")
                _defer_reduction(scratch, gw, gw.numel(), rows)
            else:
                _lib.check(lib.ias_stem_backward_weight(_lib.ptr(x), _lib.ptr(g), _lib.ptr(gw), _lib.ptr(scratch), B, H, W,
                                                        _lib.stream()), "ias_stem_backward_weight")
        return gx, gw


class StemConv2d(nn.Conv2d):
    """The 3 -> 16, 3x3, stride-2 stem (same parameters as nn.Conv2d): a direct HIP kernel on a ROCm device -- MIOpen ran
    it as an im2col plus a GEMM per sample -- the plain nn.Conv2d elsewhere."""

    def forward(self, x):
        if x.is_cuda and not trunk_torch() and x.dtype == torch.float32 and (self.in_channels, self.out_channels) == (3, 16) and \
                self.kernel_size == (3, 3) and self.stride == (2, 2) and self.padding == (1, 1) and self.bias is None and \
                self.groups == 1 and x.shape[0] <= 65535:
            return _StemFn.apply(x, self.weight)
        return super().forward(x)


class _BNActFn(torch.autograd.Function):
    """Training-mode batch norm + activation as the HIP kernels of csrc/bn_kernels.hip (ias_bn_act_forward / _backward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, act, res=None, pool=False):
        from . import _lib
        lib = _lib.load()
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty_like(mean)
        scratch = torch.empty(int(lib.ias_bn_scratch_doubles(B, C)), dtype=torch.float64, device=x.device)
        pooled = None
        if pool:
            # (y, pooled): the following squeeze-excitation block's average pool from the same launch.  pooled is NOT a
            # differentiable output: the block's backward (_SEFn / _SEProjFn) carries the pool's gradient itself
            assert res is None
            pooled = torch.empty((B, C), dtype=torch.float32, device=x.device)
            _lib.check(lib.ias_bn_act_forward_pool(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(running_mean),
                                                   _lib.ptr(running_var), _lib.ptr(y), _lib.ptr(pooled), _lib.ptr(mean),
                                                   _lib.ptr(invstd), _lib.ptr(scratch), B, C, HW, float(eps), float(momentum),
                                                   int(act), _lib.stream()), "ias_bn_act_forward_pool")
        elif res is not None:
            # the block's residual connection in the same pass: y = act(bn(x)) + res
            res = res.contiguous()
            assert res.shape == x.shape
            _lib.check(lib.ias_bn_act_forward_res(_lib.ptr(x), _lib.ptr(res), _lib.ptr(weight), _lib.ptr(bias),
                                                  _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(y), _lib.ptr(mean),
                                                  _lib.ptr(invstd), _lib.ptr(scratch), B, C, HW, float(eps), float(momentum),
                                                  int(act), _lib.stream()), "ias_bn_act_forward_res")
        else:
            _lib.check(lib.ias_bn_act_forward(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(running_mean),
                                              _lib.ptr(running_var), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(invstd),
                                              _lib.ptr(scratch), B, C, HW, float(eps), float(momentum), int(act),
                                              _lib.stream()), "ias_bn_act_forward")
        ctx.save_for_backward(x, weight, bias, mean, invstd)
        ctx.act = int(act)
        ctx.has_res = res is not None
        if pool:
            # (no zero cotangent for the pool: autograd would fill one per backward pass, a launch per block)
            ctx.mark_non_differentiable(pooled)
            ctx.set_materialize_grads(False)
            return y, pooled
        return y

    @staticmethod
    def backward(ctx, g, _g_pooled=None):
        if g is None:
            return (None,) * 10
        from . import _lib
        lib = _lib.load()
        x, weight, bias, mean, invstd = ctx.saved_tensors
        g = g.contiguous()
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        dx = torch.empty_like(x)
        gw = torch.empty_like(weight) if weight is not None else None
        gb = torch.empty_like(bias) if bias is not None else None
        scratch = torch.empty(int(lib.ias_bn_scratch_doubles(B, C)), dtype=torch.float64, device=x.device)
        sums = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        _lib.check(lib.ias_bn_act_backward(_lib.ptr(x), _lib.ptr(g), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(mean),
                                           _lib.ptr(invstd), _lib.ptr(dx), _lib.ptr(gw), _lib.ptr(gb), _lib.ptr(scratch),
                                           _lib.ptr(sums), B, C, HW, ctx.act, _lib.stream()), "ias_bn_act_backward")
        return dx, gw, gb, None, None, None, None, None, (g if ctx.has_res else None), None


_ACT_CODE = {None: 0, nn.ReLU: 1, nn.Hardswish: 2}


class BatchNormAct2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state_dict keys) that also applies the activation that follows it in
    torchvision's ConvNormActivation (``act``: None, nn.ReLU or nn.Hardswish).  Training on a ROCm device: one fused HIP
    forward and one fused HIP backward (csrc/bn_kernels.hip); evaluation, CPU tensors and exotic settings: the torch ops."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, act=None):
        super().__init__(num_features, eps=eps, momentum=momentum)
        assert act in _ACT_CODE
        self.act_code = _ACT_CODE[act]

    def forward(self, x, residual=None, pool=False):
        """``residual`` (InvertedResidual's skip connection): returns act(bn(x)) + residual, on the HIP path in one pass.
        ``pool=True``: returns (y, pooled) with pooled = y.mean((2, 3)) [B,C], detached -- the average pool of the
        squeeze-excitation block behind this layer, on the HIP path from the same launch (hand it to that block; its
        backward carries the pool's gradient)."""
        if pool:
            assert residual is None
        if self.training and x.is_cuda and not trunk_torch() and x.dtype == torch.float32 and x.dim() == 4 and self.track_running_stats and \
                self.momentum is not None and self.affine:
            if self.num_batches_tracked is not None and not getattr(self, "counter_deferred", False):
                self.num_batches_tracked.add_(1)     # (deferred: one multi-tensor add for all layers, see defer_bn_counters)
            return _BNActFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                  self.momentum, self.act_code, residual, pool)
        if pool:
            y = self.forward(x)
            return y, y.detach().mean((2, 3))
        if residual is not None:
            return residual + self.forward(x)
        y = super().forward(x)
        if self.training and getattr(self, "counter_deferred", False) and self.num_batches_tracked is not None and \
                self.track_running_stats:
            # nn.BatchNorm2d bumped the counter itself (and, with momentum=None, has just used the bumped value as its
            # averaging factor): undo it AFTER the forward, the deferred multi-tensor add counts this step
            self.num_batches_tracked.sub_(1)
        if self.act_code == 1:
            return F.relu(y)
        if self.act_code == 2:
            return F.hardswish(y)
        return y


def defer_bn_counters(module):
    """The 34 BatchNormAct2d layers of the trunk each bump their ``num_batches_tracked`` with a one-element launch per
    step.  After this call they leave it to ``bump_bn_counters(module)``, which adds 1 to all of them in ONE multi-tensor
    launch (call it once per training forward).  Nothing is captured here: the counters are looked up when they are
    bumped, so ``module.to(device)`` (which REBINDS the buffers) and ``copy.deepcopy`` keep working."""
    for m in module.modules():
        if isinstance(m, BatchNormAct2d) and m.num_batches_tracked is not None:
            m.counter_deferred = True


def bump_bn_counters(module):
    """+1 on the ``num_batches_tracked`` of every BatchNormAct2d under ``module`` that defers its counter, as they are
    bound NOW."""
    counters = [m.num_batches_tracked for m in module.modules()
                if isinstance(m, BatchNormAct2d) and getattr(m, "counter_deferred", False) and m.num_batches_tracked is not None]
    if counters:
        torch._foreach_add_(counters, 1)


def _divisible(v, d=8):
    new = max(d, int(v + d / 2) // d * d)
    return new + d if new < 0.9 * v else new


class ConvBNAct(nn.Sequential):
    def __init__(self, cin, cout, k=3, stride=1, groups=1, act=None):
        conv = PointwiseConv2d if (k == 1 and groups == 1) else (DepthwiseConv2d if groups == cin == cout else
                                                                 (StemConv2d if (cin, cout, k, stride) == (3, 16, 3, 2) else nn.Conv2d))
        # torchvision's layout is [conv, norm, activation]; the activation has no parameters, so folding it into the
        # norm module (index 1) leaves every state_dict key where it was
        layers = [conv(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
                  BatchNormAct2d(cout, eps=0.001, momentum=0.01, act=act)]
        super().__init__(*layers)


class _SEFn(torch.autograd.Function):
    """y = x * hardsigmoid(fc2(relu(fc1(mean_hw x)))), all of it in csrc/se_kernels.hip: the passes over x (pool, scale,
    and in backward the per-plane <gy, x> and gx = gy s + gpool / HW, one pass each) and the two 1x1 convolutions on
    [B, C] with their backward (one workgroup per sample; parameter gradients as small tiled products over the batch)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, pooled=None):
        from . import _lib
        lib = _lib.load()
        x = x.contiguous()
        _lib.require_f32(x, w1, b1, w2, b2)
        B, C, H, W = x.shape
        hw = H * W
        Cs = w1.shape[0]
        w1, w2 = w1.contiguous(), w2.contiguous()
        if pooled is None:          # (else: x.mean((2, 3)) as the layer in front left it behind, BatchNormAct2d.forward(pool=True))
            pooled = torch.empty((B, C), dtype=torch.float32, device=x.device)
            _lib.check(lib.ias_se_plane_reduce(_lib.ptr(x), None, _lib.ptr(pooled), B * C, hw, 1.0 / hw, _lib.stream()),
                       "ias_se_plane_reduce")
        h = torch.empty((B, Cs), dtype=torch.float32, device=x.device)
        z, s = torch.empty_like(pooled), torch.empty_like(pooled)
        _lib.check(lib.ias_se_mlp_forward(_lib.ptr(pooled), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(h),
                                          _lib.ptr(z), _lib.ptr(s), B, C, Cs, _lib.stream()), "ias_se_mlp_forward")
        y = torch.empty_like(x)
        _lib.check(lib.ias_se_scale(_lib.ptr(x), _lib.ptr(s), None, _lib.ptr(y), B * C, hw, 0.0, _lib.stream()), "ias_se_scale")
        ctx.save_for_backward(x, pooled, h, z, s, w1, w2)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import _lib
        lib = _lib.load()
        x, pooled, h, z, s, w1, w2 = ctx.saved_tensors
        gy = gy.contiguous()
        B, C, H, W = x.shape
        hw = H * W
        Cs = w1.shape[0]
        gs = torch.empty((B, C), dtype=torch.float32, device=x.device)
        _lib.check(lib.ias_se_plane_reduce(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(gs), B * C, hw, 1.0, _lib.stream()),
                   "ias_se_plane_reduce")
        gz, gp = torch.empty_like(gs), torch.empty_like(gs)
        gh = torch.empty((B, Cs), dtype=torch.float32, device=x.device)
        gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
        gb1 = torch.empty(Cs, dtype=torch.float32, device=x.device)
        gb2 = torch.empty(C, dtype=torch.float32, device=x.device)
        _lib.check(lib.ias_se_mlp_backward(_lib.ptr(gs), _lib.ptr(z), _lib.ptr(h), _lib.ptr(pooled), _lib.ptr(w1), _lib.ptr(w2),
                                           _lib.ptr(gz), _lib.ptr(gh), _lib.ptr(gp), _lib.ptr(gw1), _lib.ptr(gb1), _lib.ptr(gw2),
                                           _lib.ptr(gb2), B, C, Cs, _lib.stream()), "ias_se_mlp_backward")
        gx = torch.empty_like(x)
        _lib.check(lib.ias_se_scale(_lib.ptr(gy), _lib.ptr(s), _lib.ptr(gp), _lib.ptr(gx), B * C, hw, 1.0 / hw, _lib.stream()),
                   "ias_se_scale")
        return gx, gw1, gb1, gw2, gb2, None


class _SEProjFn(torch.autograd.Function):
    """y = W (x * s), s = hardsigmoid(fc2(relu(fc1(mean_hw x)))): a SqueezeExcitation block AND the bias-free 1x1 projection
    behind it (torchvision InvertedResidual) as one node.  The gate is applied by the projection on load
    (ias_pwconv_forward_scaled): the forward has no `scale * input` pass over the expanded map; the backward is _SEFn's
    with the projection's input gradient as its cotangent, plus the weight gradient on the gated input
    (ias_pwconv_backward_weight*_scaled).  Same values as SqueezeExcitation followed by PointwiseConv2d."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w, pooled=None):
        from . import _lib
        lib = _lib.load()
        x, w = x.contiguous(), w.contiguous()
        _lib.require_f32(x, w1, b1, w2, b2, w)
        B, C, H, W = x.shape
        hw = H * W
        Cs, Cout = w1.shape[0], w.shape[0]
        w1, w2 = w1.contiguous(), w2.contiguous()
        if pooled is None:          # (else: as the layer in front left it behind, BatchNormAct2d.forward(pool=True))
            pooled = torch.empty((B, C), dtype=torch.float32, device=x.device)
            _lib.check(lib.ias_se_plane_reduce(_lib.ptr(x), None, _lib.ptr(pooled), B * C, hw, 1.0 / hw, _lib.stream()),
                       "ias_se_plane_reduce")
        h = torch.empty((B, Cs), dtype=torch.float32, device=x.device)
        z, s = torch.empty_like(pooled), torch.empty_like(pooled)
        _lib.check(lib.ias_se_mlp_forward(_lib.ptr(pooled), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(h),
                                          _lib.ptr(z), _lib.ptr(s), B, C, Cs, _lib.stream()), "ias_se_mlp_forward")
        y = torch.empty((B, Cout, H, W), dtype=torch.float32, device=x.device)
        _lib.check(lib.ias_pwconv_forward_scaled(_lib.ptr(x), _lib.ptr(s), _lib.ptr(w), _lib.ptr(y), B, C, Cout, hw,
                                                 _lib.stream()), "ias_pwconv_forward_scaled")
        ctx.save_for_backward(x, pooled, h, z, s, w1, w2, w)
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        x, pooled, h, z, s, w1, w2, w = ctx.saved_tensors
        g = g.contiguous()
        B, C, H, W = x.shape
        hw = H * W
        Cs, Cout = w1.shape[0], w.shape[0]
        # the cotangent of the gated map x * s (_SEFn's backward runs on it below)
        gy = torch.empty_like(x)
        _lib.check(lib.ias_pwconv_backward_data(_lib.ptr(g), _lib.ptr(w), _lib.ptr(gy), B, C, Cout, hw, _lib.stream()),
                   "ias_pwconv_backward_data")
        # ... and the projection's weight gradient on the gated input (behind the input gradient, as _PointwiseFn orders them)
        gw = None
        if ctx.needs_input_grad[5]:
            gw = torch.empty_like(w)
            scratch = torch.empty(int(lib.ias_pwconv_weight_scratch(B, C, Cout, hw)), dtype=torch.float32, device=x.device)
            if _DEFER["on"]:
                rows = lib.ias_pwconv_backward_weight_partials_scaled(_lib.ptr(g), _lib.ptr(x), _lib.ptr(s), _lib.ptr(scratch),
                                                                      B, C, Cout, hw, _lib.stream())
                _lib.check(min(int(rows), 0), "ias_pwconv_backward_weight_partials_scaled")
                _defer_reduction(scratch, gw, gw.numel(), rows)
            else:
                _lib.check(lib.ias_pwconv_backward_weight_scaled(_lib.ptr(g), _lib.ptr(x), _lib.ptr(s), _lib.ptr(gw),
                                                                 _lib.ptr(scratch), B, C, Cout, hw, _lib.stream()),
                           "ias_pwconv_backward_weight_scaled")
        gs = torch.empty((B, C), dtype=torch.float32, device=x.device)
        _lib.check(lib.ias_se_plane_reduce(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(gs), B * C, hw, 1.0, _lib.stream()),
                   "ias_se_plane_reduce")
        gz, gp = torch.empty_like(gs), torch.empty_like(gs)
        gh = torch.empty((B, Cs), dtype=torch.float32, device=x.device)
        gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
        gb1 = torch.empty(Cs, dtype=torch.float32, device=x.device)
        gb2 = torch.empty(C, dtype=torch.float32, device=x.device)
        _lib.check(lib.ias_se_mlp_backward(_lib.ptr(gs), _lib.ptr(z), _lib.ptr(h), _lib.ptr(pooled), _lib.ptr(w1), _lib.ptr(w2),
                                           _lib.ptr(gz), _lib.ptr(gh), _lib.ptr(gp), _lib.ptr(gw1), _lib.ptr(gb1), _lib.ptr(gw2),
                                           _lib.ptr(gb2), B, C, Cs, _lib.stream()), "ias_se_mlp_backward")
        gx = torch.empty_like(x)
        _lib.check(lib.ias_se_scale(_lib.ptr(gy), _lib.ptr(s), _lib.ptr(gp), _lib.ptr(gx), B * C, hw, 1.0 / hw, _lib.stream()),
                   "ias_se_scale")
        return gx, gw1, gb1, gw2, gb2, gw, None


FUSE_SE_PROJECTION = True      # (tests / scripts/diag A/B runs: False keeps SqueezeExcitation and its projection apart)
SE_POOL_FROM_NORM = True       # (the same: False lets the gate pool its input itself)


def se_projection(se, conv, x, pooled=None):
    """``conv(se(x))`` for a SqueezeExcitation ``se`` and the 1x1 projection ``conv`` behind it; one fused node
    (_SEProjFn) where the projection is a shape of csrc/pointwise_kernels.hip, the two modules otherwise.  ``pooled``:
    ``x.mean((2, 3))`` where the layer in front has left it behind (BatchNormAct2d.forward(pool=True)), or None."""
    if FUSE_SE_PROJECTION and x.is_cuda and not trunk_torch() and x.dtype == torch.float32 and x.dim() == 4 and isinstance(conv, PointwiseConv2d) and \
            conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1 and conv.bias is None and \
            _pw_mfma(x.shape[1], conv.out_channels):
        return _SEProjFn.apply(x, se.fc1.weight, se.fc1.bias, se.fc2.weight, se.fc2.bias, conv.weight, pooled)
    return conv(se(x, pooled))


class SqueezeExcitation(nn.Module):
    def __init__(self, channels, squeeze):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = PointwiseConv2d(channels, squeeze, 1)
        self.fc2 = PointwiseConv2d(squeeze, channels, 1)
        self.activation = nn.ReLU()
        self.scale_activation = nn.Hardsigmoid()

    def forward(self, x, pooled=None):
        """``pooled``: ``x.mean((2, 3))`` [B,C] where the layer in front has left it behind (a hint for the HIP path: the
        gradient of the pool is taken through ``x`` either way), or None."""
        if x.is_cuda and not trunk_torch() and x.dtype == torch.float32 and x.dim() == 4:
            return _SEFn.apply(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, pooled)
        s = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
        return s * x


class InvertedResidual(nn.Module):
    def __init__(self, cin, k, cexp, cout, use_se, hs, stride):
        super().__init__()
        act = nn.Hardswish if hs else nn.ReLU
        self.use_res = stride == 1 and cin == cout
        layers = []
        if cexp != cin:
            layers.append(ConvBNAct(cin, cexp, 1, act=act))
        layers.append(ConvBNAct(cexp, cexp, k, stride, groups=cexp, act=act))
        if use_se:
            layers.append(SqueezeExcitation(cexp, _divisible(cexp // 4)))
        layers.append(ConvBNAct(cexp, cout, 1, act=None))
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        # the block's last ConvBNAct by hand: its 1x1 projection takes the squeeze-excitation gate on load (se_projection)
        # and its normalisation adds the skip connection (one pass: BatchNormAct2d.forward(residual=...))
        layers = list(self.block)
        conv, norm = layers[-1][0], layers[-1][1]
        has_se = isinstance(layers[-2], SqueezeExcitation)
        body = layers[:-2] if has_se else layers[:-1]
        h, pooled = x, None
        for i, layer in enumerate(body):
            if has_se and SE_POOL_FROM_NORM and i == len(body) - 1 and isinstance(layer, ConvBNAct) and isinstance(layer[1], BatchNormAct2d):
                # the depthwise layer in front of the squeeze-excitation block: its normalisation leaves the block's
                # average pool behind (same launch on the small maps)
                h, pooled = layer[1](layer[0](h), pool=True)
            else:
                h = layer(h)
        if has_se:
            h = se_projection(layers[-2], conv, h, pooled)
        else:
            h = conv(h)
        return norm(h, residual=x) if self.use_res else norm(h)


# (in, kernel, expanded, out, squeeze-excite, hardswish, stride)
_SMALL = [
    (16, 3, 16, 16, True, False, 2), (16, 3, 72, 24, False, False, 2), (24, 3, 88, 24, False, False, 1),
    (24, 5, 96, 40, True, True, 2), (40, 5, 240, 40, True, True, 1), (40, 5, 240, 40, True, True, 1),
    (40, 5, 120, 48, True, True, 1), (48, 5, 144, 48, True, True, 1), (48, 5, 288, 96, True, True, 2),
    (96, 5, 576, 96, True, True, 1), (96, 5, 576, 96, True, True, 1),
]


class MobileNetV3SmallFeatures(nn.Module):
    """Only ``.features`` is used by AudioEmbedding (audioembed.py:61): [B,3,240,245] -> [B,576,8,8]."""

    def __init__(self):
        super().__init__()
        layers = [ConvBNAct(3, 16, 3, 2, act=nn.Hardswish)]
        layers += [InvertedResidual(*c) for c in _SMALL]
        layers.append(ConvBNAct(96, 576, 1, act=nn.Hardswish))
        self.features = nn.Sequential(*layers)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        return self.features(x)


def mobilenet_v3_small(pretrained=False):
    if pretrained:
        warnings.warn("pretrained MobileNetV3 weights are not available offline; using random init")
    return MobileNetV3SmallFeatures()
