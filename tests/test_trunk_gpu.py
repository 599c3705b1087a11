"""The trunk's convolution kernels (csrc/conv_kernels.hip) and GEMM forms against torch's own Conv2d on the same
device (the reference takes these layers from torchvision's mobilenet_v3_small, vicreg_audio_params.py:52-54): forward,
input gradient, weight gradient; and the whole AudioEmbedding trunk against the same modules run as plain nn.Conv2d."""
import pytest
import torch
import torch.nn.functional as F

from helpers import randn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,K,S,H,W", [(16, 3, 2, 120, 123), (88, 3, 1, 30, 31), (96, 5, 2, 30, 31), (240, 5, 1, 15, 16),
                                       (576, 5, 1, 8, 8), (7, 3, 1, 5, 4), (5, 5, 2, 9, 2), (576, 5, 1, 8, 7),
                                       (4, 5, 1, 100, 90), (3, 3, 1, 130, 7), (2, 5, 2, 64, 201),    # row-tiled planes
                                       (72, 3, 2, 60, 62), (288, 5, 2, 15, 16), (3, 3, 2, 7, 1), (2, 3, 2, 300, 200)])
def test_depthwise_conv_matches_torch(lib, dev, C, K, S, H, W):
    from inverse_audio_synthesis_amd.vision import DepthwiseConv2d
    B = 128 if (H, W) == (8, 7) else 6          # 128 x 576 planes: more than one grid dimension's 65535
    m = DepthwiseConv2d(C, C, K, S, (K - 1) // 2, groups=C, bias=False).to(dev)
    x = randn((B, C, H, W), 1).to(dev).requires_grad_(True)
    y = m(x)
    ref = F.conv2d(x, m.weight, None, S, (K - 1) // 2, 1, C)
    assert y.shape == ref.shape and (y - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    g = randn(tuple(ref.shape), 2).to(dev)
    gx, gw = torch.autograd.grad(y, (x, m.weight), g)
    rx, rw = torch.autograd.grad(ref, (x, m.weight), g)
    assert (gx - rx).abs().max().item() <= 1e-5 * max(1.0, rx.abs().max().item())
    assert (gw - rw).abs().max().item() <= 2e-4 * max(1.0, rw.abs().max().item())
    gw2 = torch.autograd.grad(m(x), m.weight, g)[0]
    assert torch.equal(gw, gw2), "the weight gradient is reduced in a fixed order"
    if S == 2:      # the LDS-tiled input gradient adds its taps in the order of the direct kernel (which only the
        import os   # diagnostic library can be told to take at this shape: the product library reads no environment)
        from inverse_audio_synthesis_amd import _lib
        os.environ["IAS_DW_S2_DIRECT"] = "1"
        try:
            with _lib.use_library(_lib.load_diag()):
                gx_direct = torch.autograd.grad(m(x), x, g)[0]
        finally:
            del os.environ["IAS_DW_S2_DIRECT"]
        assert torch.equal(gx, gx_direct)


def _random_dw_shapes(n, seed=7):
    import random
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        K, S = rnd.choice([3, 5]), rnd.choice([1, 2])
        H, W = rnd.randint(1, 70), rnd.randint(1, 70)
        if rnd.random() < 0.25:
            H, W = rnd.randint(60, 140), rnd.randint(40, 130)       # row-tiled waves / workgroup tiles
        out.append((rnd.randint(1, 9), rnd.randint(1, 7), K, S, H, W))
    return out


# rows too wide for a wave (more than 64 four-output items per row): the workgroup-tiled kernels (dw_tile_kernel with its row
# tiles as grid.y, per-tile weight-gradient partials, dw_tile_bwd_s2_kernel / the direct stride-2 kernel)
_WIDE_DW_SHAPES = [(2, 3, 3, 1, 5, 1100), (2, 2, 5, 2, 40, 600), (2, 2, 3, 1, 200, 300), (3, 2, 5, 1, 70, 520), (2, 1, 3, 2, 130, 700)]


@pytest.mark.parametrize("B,C,K,S,H,W", _random_dw_shapes(48) + _WIDE_DW_SHAPES)
def test_depthwise_conv_random_shapes(lib, dev, B, C, K, S, H, W):
    """Random plane sizes through every depthwise path (wave-owned planes with 1 / 2 / 4 planes per wave, groups moved back
    at the end of the tensor / of the channels, row-tiled waves with boundary tiles, workgroup tiles, the direct kernels):
    forward, input gradient and weight gradient against torch's conv2d on the same device."""
    from inverse_audio_synthesis_amd.vision import DepthwiseConv2d
    m = DepthwiseConv2d(C, C, K, S, (K - 1) // 2, groups=C, bias=False).to(dev)
    x = randn((B, C, H, W), 31).to(dev).requires_grad_(True)
    y = m(x)
    ref = F.conv2d(x, m.weight, None, S, (K - 1) // 2, 1, C)
    assert y.shape == ref.shape and (y - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    g = randn(tuple(ref.shape), 32).to(dev)
    gx, gw = torch.autograd.grad(y, (x, m.weight), g)
    rx, rw = torch.autograd.grad(ref, (x, m.weight), g)
    assert (gx - rx).abs().max().item() <= 1e-5 * max(1.0, rx.abs().max().item())
    assert (gw - rw).abs().max().item() <= 2e-4 * max(1.0, rw.abs().max().item())


def test_stem_conv_matches_torch(lib, dev):
    from inverse_audio_synthesis_amd.vision import StemConv2d
    m = StemConv2d(3, 16, 3, 2, 1, bias=False).to(dev)
    for shape in ((4, 3, 240, 245), (3, 3, 17, 10), (128, 3, 240, 245), (2, 3, 121, 128), (1, 3, 5, 300)):   # third: configs[2]
        x = randn(shape, 3).to(dev)
        y = m(x)
        ref = F.conv2d(x, m.weight, None, 2, 1)
        assert y.shape == ref.shape and (y - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
        g = randn(tuple(ref.shape), 4).to(dev)
        gw = torch.autograd.grad(y, m.weight, g)[0]
        rw = torch.autograd.grad(ref, m.weight, g)[0]
        assert (gw - rw).abs().max().item() <= 2e-4 * max(1.0, rw.abs().max().item())


def test_trunk_matches_plain_conv2d_modules(lib, dev):
    """AudioEmbedding (fused PQMF epilogue, HIP stem / depthwise kernels, 1x1 and 2x2 convolutions as GEMMs) against the
    same parameters evaluated with torch's own conv2d everywhere: forward and parameter gradients (eval mode)."""
    from inverse_audio_synthesis_amd.audioembed import AudioEmbedding, ChannelNormalize
    from inverse_audio_synthesis_amd.pqmf import PQMF
    from inverse_audio_synthesis_amd.vision import mobilenet_v3_small
    torch.manual_seed(0)
    net = AudioEmbedding(PQMF(N=3), mobilenet_v3_small(), ChannelNormalize(), dim=32).to(dev).eval()
    audio = (randn((2, 1, 176400), 5) * 0.3).to(dev)
    out = net(audio)
    w = randn(tuple(out.shape), 6).to(dev)
    params = [p for p in net.parameters() if p.requires_grad]
    grads = torch.autograd.grad((out * w).sum(), params)

    def plain_forward(m, x):      # every conv through F.conv2d
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d):
                mod.forward = (lambda self: lambda t: F.conv2d(t, self.weight, self.bias, self.stride, self.padding,
                                                               self.dilation, self.groups))(mod)
        t = m.vision_model.features(m._preprocess(x))
        for i in range(7, 0, -1):
            t = getattr(m, f"conv{i}")(t)
        return t.view(-1, m.dim)

    ref = plain_forward(net, audio)
    assert (out - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    rgrads = torch.autograd.grad((ref * w).sum(), params)
    for a, b in zip(grads, rgrads):
        assert (a - b).abs().max().item() <= 2e-3 * max(1e-3, b.abs().max().item())


# ((128, 96, 8, 8) ... (100, 40, 15, 16) take the one-workgroup-per-channel kernels: 2 and 8 vectors per thread, a ragged
# vector count; (128, 88, 30, 31) is a 30 x 31 layer of the trunk at its full size: planes of 930 floats, every other one
# starting 8 bytes into a 16-byte group -- head / vector body / tail items of bn_partials_kernel, several planes per lane;
# (6, 3, 5, 5): planes at all four phases, shorter than a workgroup; (128, 16, 120, 123): the stem's map at its full size,
# 121 MB -- maps of 64 MB and more keep a finalize launch between the partial sums and the elementwise pass, the others
# finalize in the elementwise pass' workgroups (bn_finapply_kernel))
@pytest.mark.parametrize("shape", [(8, 16, 120, 123), (4, 24, 30, 31), (3, 5, 7, 9), (16, 576, 8, 8), (2, 3, 1, 1),
                                   (128, 96, 8, 8), (128, 240, 15, 16), (100, 40, 15, 16), (128, 88, 30, 31), (6, 3, 5, 5),
                                   (128, 16, 120, 123)])
@pytest.mark.parametrize("act", [None, torch.nn.ReLU, torch.nn.Hardswish])
def test_fused_batchnorm_activation_matches_torch(lib, dev, shape, act):
    """BatchNormAct2d (ias_bn_act_forward / _backward) against nn.BatchNorm2d + activation in fp64 on the same data:
    output, running statistics after two steps, gradients w.r.t. input, weight and bias."""
    from inverse_audio_synthesis_amd.vision import BatchNormAct2d
    C = shape[1]
    g = torch.Generator().manual_seed(7)
    fused = BatchNormAct2d(C, eps=0.001, momentum=0.01, act=act).to(dev).train()
    ref = torch.nn.BatchNorm2d(C, eps=0.001, momentum=0.01).double().train()
    with torch.no_grad():
        w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
        fused.weight.copy_(w); fused.bias.copy_(b); ref.weight.copy_(w.double()); ref.bias.copy_(b.double())
    actf = (lambda t: t) if act is None else act()
    for step in range(2):
        x = torch.randn(shape, generator=g) * 2.0 + torch.linspace(-5, 50, C).view(1, C, 1, 1)   # large channel means
        up = torch.randn(shape, generator=g)
        xf = x.to(dev).requires_grad_(True)
        xr = x.double().requires_grad_(True)
        yf = fused(xf)
        yr = actf(ref(xr))
        yf.backward(up.to(dev))
        yr.backward(up.double())
        assert (yf.detach().cpu().double() - yr.detach()).abs().max().item() <= 2e-5 * max(1.0, yr.abs().max().item())
        sc = max(1.0, xr.grad.abs().max().item())
        ddx = (xf.grad.cpu().double() - xr.grad).abs()
        if act is not None:
            # at a kink the derivative jumps (ReLU: at 0; Hardswish: at -3 and 3, by 1/2): an element whose pre-activation
            # value is within fp32 rounding of one may land on the other side in fp32 (a few among the millions of the
            # large shapes); they are left out
            xd = xr.detach()
            mu, var = xd.mean(dim=(0, 2, 3), keepdim=True), xd.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
            z = (xd - mu) / torch.sqrt(var + 0.001) * ref.weight.detach().view(1, -1, 1, 1) + ref.bias.detach().view(1, -1, 1, 1)
            away = z.abs() > 1e-4 if act is torch.nn.ReLU else ((z - 3.0).abs() > 1e-4) & ((z + 3.0).abs() > 1e-4)
            assert (~away).double().mean().item() <= 1e-3
            ddx = ddx[away]
        assert ddx.max().item() <= 5e-5 * sc, "dx"
        flip = torch.zeros(C, dtype=torch.float64)      # what the left-out kink elements could move a channel's sums by
        if act is not None:
            xhat = (xd - mu) / torch.sqrt(var + 0.001)
            flip = ((~away) * up.double().abs() * (1.0 + xhat.abs())).sum(dim=(0, 2, 3))
        for name in ("weight", "bias"):
            gf, gr = getattr(fused, name).grad.cpu().double(), getattr(ref, name).grad
            assert ((gf - gr).abs() <= 1e-4 * max(1.0, gr.abs().max().item()) + flip).all(), name
        fused.zero_grad(); ref.zero_grad()
    assert (fused.running_mean.cpu().double() - ref.running_mean).abs().max().item() <= 1e-5 * 50
    assert (fused.running_var.cpu().double() - ref.running_var).abs().max().item() <= 1e-5 * max(1.0, ref.running_var.max().item())
    assert int(fused.num_batches_tracked) == 2
    fused.eval()
    xe = torch.randn(shape, generator=g)
    ye = fused(xe.to(dev)).cpu().double()
    ref.eval()
    assert (ye - actf(ref(xe.double()))).abs().max().item() <= 1e-4


@pytest.mark.parametrize("B,H,W,C,Cout", [(3, 8, 8, 64, 32), (2, 2, 2, 8, 8), (5, 3, 6, 12, 20), (128, 8, 8, 576, 16)])
def test_head_conv2x2_matches_torch(lib, dev, B, H, W, C, Cout):
    """conv2x2_nhwc (ias_conv2x2_patches + GEMM) against F.conv2d with the same weight / bias: output, input gradient,
    weight and bias gradients."""
    from inverse_audio_synthesis_amd.audioembed import conv2x2_nhwc
    x = randn((B, H, W, C), 11).to(dev).requires_grad_(True)
    w = (randn((Cout, C, 2, 2), 12) * 0.1).to(dev).requires_grad_(True)
    b = randn((Cout,), 13).to(dev).requires_grad_(True)
    y = conv2x2_nhwc(x, w, b)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b).permute(0, 2, 3, 1)
    assert y.shape == ref.shape and (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    g = randn(tuple(ref.shape), 14).to(dev)
    got = torch.autograd.grad(y, (x, w, b), g)
    want = torch.autograd.grad(ref, (x, w, b), g)
    for a, r, name in zip(got, want, ("gx", "gw", "gb")):
        assert (a - r).abs().max().item() <= 1e-4 * max(1.0, r.abs().max().item()), name


@pytest.mark.parametrize("B,H,W,C,Cout", [(3, 8, 8, 64, 32), (2, 2, 2, 8, 8), (5, 3, 6, 12, 20), (128, 8, 8, 576, 16),
                                          (2, 15, 17, 70, 8), (1, 4, 5, 130, 12)])
def test_head_conv2x2_from_nchw_matches_torch(lib, dev, B, H, W, C, Cout):
    """conv2x2_from_nchw (ias_conv2x2_patches_nchw + GEMM: the first head layer on the trunk's NCHW output, no permuted
    copy) against F.conv2d: output, NCHW input gradient, weight and bias gradients.  Channel counts that are not a
    multiple of the 64-channel tile, the largest map the LDS tile takes (15 x 17)."""
    from inverse_audio_synthesis_amd.audioembed import conv2x2_from_nchw
    x = randn((B, C, H, W), 11).to(dev).requires_grad_(True)
    w = (randn((Cout, C, 2, 2), 12) * 0.1).to(dev).requires_grad_(True)
    b = randn((Cout,), 13).to(dev).requires_grad_(True)
    y = conv2x2_from_nchw(x, w, b)
    ref = F.conv2d(x, w, b).permute(0, 2, 3, 1)
    assert y.shape == ref.shape and (y - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    g = randn(tuple(ref.shape), 14).to(dev)
    got = torch.autograd.grad(y, (x, w, b), g)
    want = torch.autograd.grad(ref, (x, w, b), g)
    for a, r, name in zip(got, want, ("gx", "gw", "gb")):
        assert a.shape == r.shape and (a - r).abs().max().item() <= 1e-4 * max(1.0, r.abs().max().item()), name
    with pytest.raises(RuntimeError):     # maps beyond the LDS tile are refused by the C ABI (the module permutes instead)
        conv2x2_from_nchw(randn((1, 4, 16, 16), 15).to(dev), w[:, :4].contiguous(), b)


@pytest.mark.parametrize("B,C,Cs,H,W", [(4, 16, 8, 60, 62), (3, 96, 24, 15, 16), (2, 576, 144, 8, 8), (2, 8, 8, 3, 3),
                                        (5, 12, 4, 7, 10), (128, 240, 64, 15, 16), (130, 120, 32, 4, 4), (3, 10, 6, 5, 5),
                                        (2, 1028, 260, 2, 2), (1, 288, 72, 8, 8)])
def test_squeeze_excitation_matches_torch(lib, dev, B, C, Cs, H, W):
    """SqueezeExcitation on the device (csrc/se_kernels.hip + GEMMs) against the same block written with torch ops in
    fp64: output, input gradient and the four parameter gradients."""
    from inverse_audio_synthesis_amd.vision import SqueezeExcitation
    torch.manual_seed(3)
    m = SqueezeExcitation(C, Cs).to(dev)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    x = randn((B, C, H, W), 21).to(dev).requires_grad_(True)
    g = randn((B, C, H, W), 22).to(dev)
    y = m(x)
    params = [m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]
    got = torch.autograd.grad(y, [x] + params, g)

    xd = x.detach().double().requires_grad_(True)
    pd = [p.detach().double().requires_grad_(True) for p in params]
    pooled = xd.mean((2, 3), keepdim=True)
    s = F.hardsigmoid(F.conv2d(F.relu(F.conv2d(pooled, pd[0], pd[1])), pd[2], pd[3]))
    ref = s * xd
    want = torch.autograd.grad(ref, [xd] + pd, g.double())
    assert (y.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    for a, r, name in zip(got, want, ("gx", "gw1", "gb1", "gw2", "gb2")):
        assert (a.double() - r).abs().max().item() <= 1e-4 * max(1.0, r.abs().max().item()), name


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(3, 16, 16, 60, 62), (2, 16, 72, 60, 62), (4, 72, 24, 30, 31), (4, 24, 88, 30, 31),
                                            (5, 240, 40, 15, 16), (5, 40, 240, 15, 16), (3, 48, 288, 15, 16), (2, 288, 96, 8, 8),
                                            (7, 8, 12, 3, 3), (2, 96, 576, 8, 8), (130, 24, 96, 5, 7), (3, 576, 96, 8, 8),
                                            (2, 1028, 16, 4, 4), (2, 6, 10, 4, 4)])     # the last two: batched-GEMM form
def test_pointwise_conv_matches_torch(lib, dev, B, Cin, Cout, H, W):
    """PointwiseConv2d (csrc/pointwise_kernels.hip where ias_pwconv_supported, the batched GEMM otherwise) against
    F.conv2d in fp64: output, input gradient, weight gradient; the weight gradient twice (fixed-order reduction)."""
    from inverse_audio_synthesis_amd.vision import PointwiseConv2d
    m = PointwiseConv2d(Cin, Cout, 1, bias=False).to(dev)
    x = randn((B, Cin, H, W), 31).to(dev).requires_grad_(True)
    g = randn((B, Cout, H, W), 32).to(dev)
    y = m(x)
    gx, gw = torch.autograd.grad(y, (x, m.weight), g)
    xd, wd = x.detach().double().requires_grad_(True), m.weight.detach().double().requires_grad_(True)
    ref = F.conv2d(xd, wd)
    rx, rw = torch.autograd.grad(ref, (xd, wd), g.double())
    assert (y.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    assert (gx.double() - rx).abs().max().item() <= 1e-5 * max(1.0, rx.abs().max().item())
    assert (gw.double() - rw).abs().max().item() <= 1e-4 * max(1.0, rw.abs().max().item())
    gw2 = torch.autograd.grad(m(x), m.weight, g)[0]
    if lib.ias_pwconv_supported(Cin, Cout):
        assert torch.equal(gw, gw2)


@pytest.mark.parametrize("shape", [(128, 240, 15, 16), (128, 96, 8, 8), (100, 40, 15, 16), (16, 576, 8, 8), (4, 16, 60, 62),
                                   (3, 24, 30, 31)])
@pytest.mark.parametrize("act", [None, torch.nn.Hardswish])
def test_batchnorm_leaves_the_squeeze_excitation_pool_behind(lib, dev, shape, act):
    """BatchNormAct2d.forward(x, pool=True) (ias_bn_act_forward_pool: the following squeeze-excitation block's average pool
    from the normalisation's own launch on maps up to 15 x 16, ias_se_plane_reduce behind it on larger ones): the same y
    and the same gradients as the plain forward, pooled = y.mean((2, 3)), not differentiable; and an inverted-residual
    block that hands the pool on against the same block with the pool computed by the gate itself."""
    from inverse_audio_synthesis_amd.vision import BatchNormAct2d
    import copy
    torch.manual_seed(6)
    a = BatchNormAct2d(shape[1], eps=0.001, momentum=0.01, act=act).to(dev).train()
    with torch.no_grad():
        a.weight.copy_(1.0 + 0.2 * torch.randn(shape[1])); a.bias.copy_(0.1 * torch.randn(shape[1]))
    b = copy.deepcopy(a)
    x1 = randn(shape, 51).to(dev).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    y1, pooled = a(x1, pool=True)
    y2 = b(x2)
    assert torch.equal(y1, y2)
    assert not pooled.requires_grad and pooled.shape == shape[:2]
    ref = y2.detach().double().mean((2, 3))
    assert (pooled.double() - ref).abs().max().item() <= 1e-6 * max(1.0, ref.abs().max().item())
    g = randn(shape, 52).to(dev)
    got = torch.autograd.grad(y1, [x1, a.weight, a.bias], g)
    want = torch.autograd.grad(y2, [x2, b.weight, b.bias], g)
    for u, v in zip(got, want):
        assert torch.equal(u, v)
    assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)


def test_inverted_residual_hands_the_pool_to_its_gate(lib, dev):
    """vision.InvertedResidual with a squeeze-excitation block: the depthwise normalisation's pool reaches the gate
    (BatchNormAct2d.forward(pool=True) -> se_projection(..., pooled)) -- against the block's layers applied one by one
    (the gate pooling its input itself): output and every gradient to 1e-5 of the largest element."""
    from inverse_audio_synthesis_amd import vision
    import copy
    torch.manual_seed(7)
    blk = vision.InvertedResidual(40, 5, 240, 40, True, True, 1).to(dev).train()
    ref = copy.deepcopy(blk)
    x1 = randn((9, 40, 15, 16), 61).to(dev).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    g = randn((9, 40, 15, 16), 62).to(dev)
    y1 = blk(x1)
    h = x2
    for layer in ref.block:
        h = layer(h)
    y2 = h + x2
    assert (y1 - y2).abs().max().item() <= 1e-5 * max(1.0, y2.abs().max().item())
    got = torch.autograd.grad(y1, [x1] + list(blk.parameters()), g)
    want = torch.autograd.grad(y2, [x2] + list(ref.parameters()), g)
    for (n, _), u, v in zip([("x", None)] + list(blk.named_parameters()), got, want):
        assert (u - v).abs().max().item() <= 1e-4 * max(1e-3, v.abs().max().item()), n


@pytest.mark.parametrize("B,C,Cs,Cout,H,W", [(3, 16, 8, 16, 60, 62), (4, 96, 24, 40, 15, 16), (5, 240, 64, 40, 15, 16),
                                             (3, 120, 32, 48, 15, 16), (130, 144, 40, 48, 15, 16), (3, 24, 8, 8, 5, 7),
                                             (2, 288, 72, 96, 8, 8)])      # the last: the projection's batched-GEMM form
def test_squeeze_excitation_gate_taken_by_the_projection_on_load(lib, dev, B, C, Cs, Cout, H, W):
    """vision.se_projection (_SEProjFn: ias_pwconv_forward_scaled / ias_pwconv_backward_weight_scaled, no `scale * input`
    pass) against SqueezeExcitation followed by PointwiseConv2d: the same bits forward and for every gradient that does not
    pass through the weight-gradient sums, the projection's weight gradient against fp64 (and, where the fused node runs,
    the same bits twice and through the deferred joint reduction)."""
    from inverse_audio_synthesis_amd import vision
    torch.manual_seed(5)
    se = vision.SqueezeExcitation(C, Cs).to(dev)
    conv = vision.PointwiseConv2d(C, Cout, 1, bias=False).to(dev)
    with torch.no_grad():
        for p in se.parameters():
            p.copy_(torch.randn(p.shape) * 0.5)
    params = [se.fc1.weight, se.fc1.bias, se.fc2.weight, se.fc2.bias, conv.weight]
    x1 = randn((B, C, H, W), 41).to(dev).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True)
    g = randn((B, Cout, H, W), 42).to(dev)
    y1 = vision.se_projection(se, conv, x1)
    y2 = conv(se(x2))
    assert torch.equal(y1, y2)
    got = torch.autograd.grad(y1, [x1] + params, g)
    want = torch.autograd.grad(y2, [x2] + params, g)
    for a, r, name in zip(got[:5], want[:5], ("gx", "gw1", "gb1", "gw2", "gb2")):
        assert torch.equal(a, r), name
    xd = x1.detach().double().requires_grad_(True)
    pd = [p.detach().double().requires_grad_(True) for p in params]
    sd = F.hardsigmoid(F.conv2d(F.relu(F.conv2d(xd.mean((2, 3), keepdim=True), pd[0], pd[1])), pd[2], pd[3]))
    rw = torch.autograd.grad(F.conv2d(sd * xd, pd[4]), pd[4], g.double())[0]
    assert (got[5].double() - rw).abs().max().item() <= 1e-4 * max(1.0, rw.abs().max().item())
    if lib.ias_pwconv_supported(C, Cout):
        again = torch.autograd.grad(vision.se_projection(se, conv, x1), conv.weight, g)[0]
        assert torch.equal(again, got[5])
        vision.defer_weight_reductions(True)
        try:
            deferred = torch.autograd.grad(vision.se_projection(se, conv, x1), conv.weight, g)[0]
            vision._flush_reductions()
        finally:
            vision.defer_weight_reductions(False)
        assert torch.equal(deferred, got[5])


@pytest.mark.parametrize("shape", [(128, 40, 15, 16), (6, 24, 30, 31), (3, 96, 8, 8), (2, 33, 7, 9), (128, 24, 30, 31)])
def test_batchnorm_with_the_residual_in_the_same_pass(lib, dev, shape):
    """BatchNormAct2d.forward(x, residual=r) (ias_bn_act_forward_res: the skip connection of an inverted-residual block
    added by the block's last normalisation) against the separate addition: the same bits forward, the same gradients for
    x, the residual and the affine parameters -- on the one-workgroup-per-channel kernels (maps up to 15 x 16), the
    three-launch form (30 x 31) and the scalar form (odd sizes)."""
    from inverse_audio_synthesis_amd.vision import BatchNormAct2d
    import copy
    torch.manual_seed(4)
    a = BatchNormAct2d(shape[1], eps=0.001, momentum=0.01, act=None).to(dev).train()
    with torch.no_grad():
        a.weight.copy_(1.0 + 0.2 * torch.randn(shape[1])); a.bias.copy_(0.1 * torch.randn(shape[1]))
    b = copy.deepcopy(a)
    x1 = randn(shape, 1).to(dev).requires_grad_(True); r1 = randn(shape, 2).to(dev).requires_grad_(True)
    x2 = x1.detach().clone().requires_grad_(True); r2 = r1.detach().clone().requires_grad_(True)
    y1 = a(x1, residual=r1)
    y2 = r2 + b(x2)
    assert torch.equal(y1, y2)
    g = randn(shape, 3).to(dev)
    got = torch.autograd.grad(y1, [x1, r1, a.weight, a.bias], g)
    want = torch.autograd.grad(y2, [x2, r2, b.weight, b.bias], g)
    for u, v in zip(got, want):
        assert torch.equal(u, v)
    assert torch.equal(a.running_mean, b.running_mean) and torch.equal(a.running_var, b.running_var)


def test_deferred_weight_gradient_reductions_equal_the_immediate_ones(lib, dev):
    """vision.defer_weight_reductions(True): the thin 1x1 / depthwise / stem weight gradients leave their partial rows
    behind and ONE ias_reduce_partials_multi launch at the end of the backward pass fills all of them (Trainer's setting on
    one rank) -- against the same backward with a reduction launch per layer.  Train mode, B = 6, through autograd.grad
    and through .backward(); the 1x1 layers add in the same order in both forms (bit-equal), the depthwise / stem layers
    in another fixed order (1e-6 of the largest element)."""
    from inverse_audio_synthesis_amd import vision
    from inverse_audio_synthesis_amd.audioembed import AudioEmbedding, ChannelNormalize
    from inverse_audio_synthesis_amd.pqmf import PQMF
    torch.manual_seed(0)
    net = AudioEmbedding(PQMF(N=3), vision.mobilenet_v3_small(), ChannelNormalize(), dim=32).to(dev).train()
    audio = (randn((6, 1, 176400), 5) * 0.3).to(dev)
    params = [p for p in net.parameters() if p.requires_grad]
    names = [n for n, p in net.named_parameters() if p.requires_grad]
    state = {k: v.clone() for k, v in net.state_dict().items()}
    res = {}
    for on in (False, True, True):
        net.load_state_dict(state)                  # (the same BatchNorm running statistics going in)
        old = vision.defer_weight_reductions(on)
        try:
            out = net(audio)
            w = randn(tuple(out.shape), 6).to(dev)
            if len(res) < 2:
                g = torch.autograd.grad((out * w).sum(), params)
            else:
                for p in params:
                    p.grad = None
                (out * w).sum().backward()
                g = [p.grad for p in params]
        finally:
            assert vision.defer_weight_reductions(old) == on
        assert not vision._DEFER["items"]           # everything queued was reduced before the pass returned
        res[len(res)] = [t.detach().clone() for t in g]
    assert vision._DEFER["on"] is False
    nthin = 0
    for n, a, b, c in zip(names, res[0], res[1], res[2]):
        assert torch.equal(b, c), n
        scale = max(1e-6, a.abs().max().item())
        assert (a - b).abs().max().item() <= 1e-6 * scale, (n, (a - b).abs().max().item(), scale)
        if a.dim() == 4 and a.shape[2:] == (1, 1) and torch.equal(a, b):
            nthin += 1
    assert nthin >= 10


def test_reduce_partials_multi_through_the_c_abi(lib, dev):
    """ias_reduce_partials_multi: out_j[i] = sum_r partial_j[r n_j + i] for a host table of items -- ragged n (1, 15, 16, 17,
    432, 9600), 1 .. 600 rows, and more items than one launch carries in its arguments (the library then launches twice);
    bad entries are refused."""
    import ctypes
    from inverse_audio_synthesis_amd import _lib
    from inverse_audio_synthesis_amd.vision import _ReduceItem
    g = torch.Generator().manual_seed(7)
    shapes = [(1, 1), (15, 3), (16, 64), (17, 65), (432, 128), (9600, 130), (33, 600)] + [(5 + k, 1 + (k % 7)) for k in range(120)]
    parts = [torch.randn(r, n, generator=g).to(dev) for n, r in shapes]
    outs = [torch.full((n,), float("nan"), device=dev) for n, r in shapes]
    table = (_ReduceItem * len(shapes))()
    for t, p, o, (n, r) in zip(table, parts, outs, shapes):
        t.partial, t.out, t.n, t.rows = p.data_ptr(), o.data_ptr(), n, r
    assert lib.ias_reduce_partials_multi(ctypes.cast(table, ctypes.c_void_p), len(shapes), _lib.stream()) == 0
    torch.cuda.synchronize()
    for p, o in zip(parts, outs):
        ref = p.double().sum(0)
        assert (o.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    again = [torch.empty_like(o) for o in outs]
    for t, o in zip(table, again):
        t.out = o.data_ptr()
    assert lib.ias_reduce_partials_multi(ctypes.cast(table, ctypes.c_void_p), len(shapes), _lib.stream()) == 0
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(outs, again))          # fixed order: the same bits
    table[3].rows = 0
    assert lib.ias_reduce_partials_multi(ctypes.cast(table, ctypes.c_void_p), len(shapes), _lib.stream()) == -1
    assert lib.ias_reduce_partials_multi(None, 3, _lib.stream()) == -1
