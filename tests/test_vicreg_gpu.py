"""HIP VICReg loss vs the reference's golden values and the oracle, through the C ABI.

Tolerances: repr/std losses are fp32 reductions (1e-5 rel).  cov_loss comes from a bf16 MFMA Gram with
fp32 accumulation (north_star names bf16; it states no bar): measured error is ~1e-4 relative, asserted
at 2e-3.  The backward is the HIP closed form with both products on the bf16 matrix cores (fp32 accumulate): compared
with autograd through the oracle, asserted at 2e-3 of the largest gradient element (measured values are printed), and
with the same closed form evaluated in fp32 torch ops."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import vicreg_oracle as vo
from helpers import randn

pytestmark = pytest.mark.gpu
COV_RTOL = 2e-3


def _check(out, ref):
    loss, rep, std, cov = [float(o) for o in out]
    assert abs(rep - ref[1]) <= 1e-5 * abs(ref[1])
    assert abs(std - ref[2]) <= 1e-5 * abs(ref[2])
    assert abs(cov - ref[3]) <= COV_RTOL * abs(ref[3])
    assert abs(loss - ref[0]) <= COV_RTOL * abs(ref[0])


@pytest.mark.parametrize("tag", ["tiny", "b16", "b128", "denom_quirk", "b1024"])
def test_loss_golden(lib, dev, golden_dir, tag):
    from inverse_audio_synthesis_amd.vicreg import vicreg_loss
    g = np.load(os.path.join(golden_dir, "vicreg_loss.npz"))
    B, D, cfgB, s1, s2 = [int(v) for v in g[tag + "_meta"]]
    x, y = randn((B, D), s1), randn((B, D), s2) * 0.7 + 0.1
    out = vicreg_loss(x.to(dev), y.to(dev), cfgB)
    _check(out, g[tag + "_out"])


@pytest.mark.parametrize("B,D", [(2, 1), (3, 64), (17, 130), (64, 127), (200, 1000)])
def test_loss_ragged_shapes(lib, dev, B, D):
    from inverse_audio_synthesis_amd.vicreg import vicreg_loss
    x, y = randn((B, D), 1) * 1.3 + 0.2, randn((B, D), 2) * 0.5
    ref = [o.item() for o in vo.loss(x, y, B, D, 10.0, 5.0, 2.0)]
    out = [o.item() for o in vicreg_loss(x.to(dev), y.to(dev), B, 10.0, 5.0, 2.0)]
    assert abs(out[1] - ref[1]) <= 1e-5 * abs(ref[1])
    assert abs(out[2] - ref[2]) <= 1e-5 * max(abs(ref[2]), 1e-3)
    assert abs(out[3] - ref[3]) <= COV_RTOL * max(abs(ref[3]), 1e-6)
    assert abs(out[0] - ref[0]) <= COV_RTOL * abs(ref[0])


GRAD_RTOL = 2e-3   # of the largest gradient element (bf16 operands, fp32 accumulation)


# (widths that are not a multiple of 8 -- 130, 127, 12, 3: the HIP kernels on zero-padded columns, vicreg._VICRegLossFn.backward)
@pytest.mark.parametrize("B,D,cfgB", [(32, 256, 48), (128, 1024, 128), (200, 512, 200), (17, 136, 17), (17, 130, 17),
                                      (64, 127, 64), (8, 12, 8), (5, 3, 5)])
def test_backward_matches_autograd_of_oracle(lib, dev, B, D, cfgB):
    from inverse_audio_synthesis_amd.vicreg import vicreg_loss
    x0, y0 = randn((B, D), 5) * 0.8, randn((B, D), 6) * 1.1 + 0.3   # std < 1 and > 1 columns both occur
    xr, yr = x0.clone().requires_grad_(), y0.clone().requires_grad_()
    ref = vo.loss(xr, yr, cfgB, D)
    (ref[0] + 0.5 * ref[1] - 2.0 * ref[2] + 3.0 * ref[3]).backward()
    xg, yg = x0.to(dev).requires_grad_(), y0.to(dev).requires_grad_()
    out = vicreg_loss(xg, yg, cfgB)
    (out[0] + 0.5 * out[1] - 2.0 * out[2] + 3.0 * out[3]).backward()
    for got, want in ((xg.grad.cpu(), xr.grad), (yg.grad.cpu(), yr.grad)):
        err = (got - want).abs().max().item() / want.abs().max().item()
        print(f"[vicreg backward B={B} D={D}] max|err|/max|g| = {err:.2e}")
        assert err <= GRAD_RTOL


def test_backward_full_size_vs_fp32_closed_form(lib, dev):
    """BASELINE configs[2] / [3] shapes (128 and 1024 x 8192): the HIP backward against the same closed form in fp32
    device ops (the round-1 backward, itself 1e-4 from autograd through the oracle), and, for B = 128, against
    autograd through the CPU oracle."""
    from inverse_audio_synthesis_amd.vicreg import vicreg_loss
    from helpers import vicreg_backward_closed_form
    for B in (128, 1024):
        g = torch.Generator(device="cpu").manual_seed(B)
        x0 = torch.randn((B, 8192), generator=g) * 0.9
        y0 = torch.randn((B, 8192), generator=g) * 1.1 + 0.2
        xg, yg = x0.to(dev).requires_grad_(), y0.to(dev).requires_grad_()
        out = vicreg_loss(xg, yg, B)
        out[0].backward()
        gcoef = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev)
        rx, ry = vicreg_backward_closed_form(xg.detach(), yg.detach(), gcoef, B, 25.0, 25.0, 1.0)
        for got, want in ((xg.grad, rx), (yg.grad, ry)):
            err = (got - want).abs().max().item() / want.abs().max().item()
            print(f"[vicreg backward B={B} D=8192 vs fp32 closed form] max|err|/max|g| = {err:.2e}")
            assert err <= GRAD_RTOL
        if B == 128:
            xr, yr = x0.clone().requires_grad_(), y0.clone().requires_grad_()
            vo.loss(xr, yr, B, 8192)[0].backward()
            for got, want in ((xg.grad.cpu(), xr.grad), (yg.grad.cpu(), yr.grad)):
                assert (got - want).abs().max().item() <= GRAD_RTOL * want.abs().max().item()


def test_module_api_and_offdiagonal(lib, dev, golden_dir):
    from inverse_audio_synthesis_amd.vicreg import VICReg, off_diagonal, exclude_bias_and_norm
    g = np.load(os.path.join(golden_dir, "vicreg_loss.npz"))
    assert np.array_equal(off_diagonal(torch.from_numpy(g["offdiag7_in"]).to(dev)).cpu().numpy(), g["offdiag7_out"])
    cfg = types.SimpleNamespace(dim=32, embeddim=96, vicreg=types.SimpleNamespace(
        batch_size=8, mlp="64-64-%d", sim_coeff=25.0, std_coeff=25.0, cov_coeff=1.0))
    model = VICReg(cfg, torch.nn.Identity(), torch.nn.Identity()).to(dev)
    assert [type(m).__name__ for m in model.projector] == ["Linear", "BatchNorm1d", "ReLU", "Linear", "BatchNorm1d", "ReLU", "Linear"]
    assert model.projector[-1].bias is None and model.projector[-1].out_features == 96
    x, y = torch.from_numpy(g["tiny_x"]).to(dev), torch.from_numpy(g["tiny_y"]).to(dev)
    _check(model.loss(x, y), g["tiny_out"])
    assert exclude_bias_and_norm(torch.zeros(3)) and not exclude_bias_and_norm(torch.zeros(3, 3))


def test_full_size_properties(lib, dev):
    """BASELINE sizes (128 and 1024 x 8192): x == y gives repr 0; permuting the batch leaves every term
    unchanged (up to fp32 sum order); scaling x,y by c scales cov_loss by c^4."""
    from inverse_audio_synthesis_amd.vicreg import vicreg_loss
    for B in (128, 1024):
        g = torch.Generator(device="cpu").manual_seed(B)
        x = torch.randn((B, 8192), generator=g).to(dev)
        y = (torch.randn((B, 8192), generator=g) * 0.5).to(dev)
        out = [o.item() for o in vicreg_loss(x, y, B)]
        same = [o.item() for o in vicreg_loss(x, x.clone(), B)]
        assert same[1] == 0.0
        perm = torch.randperm(B, generator=g).to(dev)
        outp = [o.item() for o in vicreg_loss(x[perm], y[perm], B)]
        for a, b in zip(out, outp):
            assert abs(a - b) <= 1e-4 * abs(a)
        out2 = [o.item() for o in vicreg_loss(2.0 * x, 2.0 * y, B)]
        assert abs(out2[3] - 16.0 * out[3]) <= 1e-3 * 16.0 * out[3]
        assert abs(out2[1] - 4.0 * out[1]) <= 1e-5 * 4.0 * out[1]


@pytest.mark.parametrize("B,D", [(128, 8192), (1024, 8192), (200, 1032)])
def test_backward_is_bit_reproducible(lib, dev, B, D):
    """The reference trains with deterministic=True (pretrain.py:100): the HIP backward has no float atomics -- the B x B
    Gram's D-slices write their own partial tiles and are summed in slice order -- so repeated calls return identical bits."""
    from inverse_audio_synthesis_amd.vicreg import vicreg_loss
    x = (randn((B, D), 11) * 0.7).to(dev).requires_grad_()
    y = (randn((B, D), 12) * 0.7 + 0.1).to(dev).requires_grad_()
    grads = []
    for _ in range(3):
        out = vicreg_loss(x, y, B, 25.0, 25.0, 1.0)
        grads.append(torch.autograd.grad(out[0], (x, y)))
    for g in grads[1:]:
        assert torch.equal(g[0], grads[0][0]) and torch.equal(g[1], grads[0][1])


@pytest.fixture
def both_forms(lib):
    """ias_vicreg_set_form exists in the DIAGNOSTIC library only (include/ias_hip_diag.h; the product library picks the
    side from the shape and keeps no state) and is process-wide there: the test routes the package through that library
    for the comparison and puts the default back whatever happens."""
    from inverse_audio_synthesis_amd import _lib
    assert not hasattr(lib, "ias_vicreg_set_form"), "the product library must not export the process-wide switch"
    diag = _lib.load_diag()
    with _lib.use_library(diag):
        yield diag.ias_vicreg_set_form
    diag.ias_vicreg_set_form(-1)


def _whitened(B, D, seed):
    """Columns as decorrelated as a rank-(B-1) matrix allows: orthonormal rows scaled to unit column variance (the case
    in which the D x D matrix's diagonal is as large a share of ||C||_F^2 as it can be)."""
    g = torch.Generator().manual_seed(seed)
    q, _ = torch.linalg.qr(torch.randn(D, B, generator=g, dtype=torch.float64))
    x = q.t().contiguous()
    x = x - x.mean(0, keepdim=True)
    return (x / x.std(0, keepdim=True).clamp_min(1e-6)).float()


@pytest.mark.parametrize("B,D,cfgB,kind", [(128, 8192, 128, "randn"), (1024, 8192, 1024, "randn"), (128, 128, 128, "randn"),
                                           (120, 128, 120, "white"), (128, 8192, 128, "white"), (17, 136, 17, "randn"),
                                           (200, 512, 64, "randn"), (256, 256, 256, "white"), (300, 1032, 300, "randn")])
def test_batch_side_and_feature_side_forms_agree(lib, dev, both_forms, B, D, cfgB, kind):
    """The covariance term contracted over the batch (B x B matrix, the default where the padded batch <= D) and over
    the features (D x D, the reference's literal order, vicreg.py:47-51): same loss and same gradient, and both against
    the oracle.  Also checks that the backward after a batch-side forward (which reuses the forward's matrix) equals the
    backward after a feature-side forward (which builds it)."""
    from inverse_audio_synthesis_amd.vicreg import vicreg_loss
    if kind == "white":
        x0, y0 = _whitened(B, D, 3), _whitened(B, D, 4) * 0.9
    else:
        x0, y0 = randn((B, D), 11) * 0.9 + 0.1, randn((B, D), 12) * 1.2
    big = D > 2048                                                  # (fp64 D x D on the host only where it is cheap)
    ref = [float(o) for o in (vo.loss(x0, y0, cfgB, D) if big else vo.loss(x0.double(), y0.double(), cfgB, D))]
    res = {}
    for form in (0, 1):
        assert both_forms(form) == 0
        xd, yd = x0.to(dev).requires_grad_(), y0.to(dev).requires_grad_()
        out = vicreg_loss(xd, yd, cfgB)
        out[0].backward()
        res[form] = ([float(o.detach()) for o in out], xd.grad.cpu(), yd.grad.cpu())
        _check(res[form][0], ref)
    for k in range(4):
        assert abs(res[0][0][k] - res[1][0][k]) <= 1e-4 * abs(ref[k]) + 1e-12, (k, res[0][0], res[1][0])
    for a, b in ((res[0][1], res[1][1]), (res[0][2], res[1][2])):
        assert torch.equal(a, b)                 # the same B x B matrix either way: the gradient is bit-equal
    # ... and the product library (no switch: the side follows from the shape) gives the batch-side bits wherever that
    # side applies, the feature-side bits elsewhere
    from inverse_audio_synthesis_amd import _lib
    with _lib.use_library(lib):
        xd, yd = x0.to(dev).requires_grad_(), y0.to(dev).requires_grad_()
        out = vicreg_loss(xd, yd, cfgB)
        out[0].backward()
    kpad = (B + 127) // 128 * 128
    like = 1 if (D % 8 == 0 and kpad <= D) else 0
    assert [float(o.detach()) for o in out] == res[like][0]
    assert torch.equal(xd.grad.cpu(), res[like][1]) and torch.equal(yd.grad.cpu(), res[like][2])
    print(f"cov_loss: oracle {ref[3]:.6e}  D x D {res[0][0][3]:.6e}  B x B {res[1][0][3]:.6e}")


@pytest.mark.parametrize("B,D", [(128, 1024), (100, 136)])
def test_backward_on_views_that_are_not_16_byte_aligned(lib, dev, B, D):
    """Batch <= 128 takes vicreg_grad128_kernel (16-byte accesses) when x, y and the gradients are 16-byte aligned and
    the 4-byte kernel otherwise: a contiguous view that starts one float into its storage gives the same gradient."""
    from inverse_audio_synthesis_amd.vicreg import vicreg_loss
    x0, y0 = randn((B, D), 21) * 0.8 + 0.2, randn((B, D), 22) * 1.1
    res = []
    for off in (0, 1):
        fx, fy = torch.zeros(B * D + 4, device=dev), torch.zeros(B * D + 4, device=dev)
        xd, yd = fx[off:off + B * D].view(B, D), fy[off:off + B * D].view(B, D)
        xd.copy_(x0); yd.copy_(y0)
        assert xd.data_ptr() % 16 == 4 * off
        xd.requires_grad_(); yd.requires_grad_()
        out = vicreg_loss(xd, yd, B)
        gx, gy = torch.autograd.grad(out[0], (xd, yd))
        res.append((gx.cpu(), gy.cpu()))
    scale = float(res[0][0].abs().max())
    for a, b in zip(res[0], res[1]):
        assert float((a - b).abs().max()) <= 1e-5 * scale


@pytest.mark.parametrize("n,widths", [(128, (96, 256, 256, 64)), (20, (40, 200, 72, 24)), (200, (32, 64, 64, 32)),
                                       (300, (16, 40, 32, 8)), (128, (1024, 8192, 8192, 8192))])   # last: configs[2]'s projector
def test_project_pair_fused_batchnorm_matches_two_projector_calls(lib, dev, n, widths):
    """vicreg.project_pair on the GPU (one GEMM without bias + ONE launch per Linear -> BatchNorm1d -> ReLU layer for both
    branches: ias_bn1d_groups_forward / _backward) against projector(a), projector(b) as the reference calls it
    (vicreg.py:27-30) on torch's own kernels: outputs, running statistics, counters, input and parameter gradients.
    Row counts per branch on each of the kernel's three forms (16 / 32 rows per thread in registers, re-read), feature
    counts that are not a multiple of the 32-feature tile."""
    import copy
    from inverse_audio_synthesis_amd import vicreg
    torch.manual_seed(11)
    d0, d1, d2, d3 = widths
    if d1 >= 8192:
        # the full-size projector WITHOUT its ReLUs: of 2 x 2 M pre-activations a few hundred lie within rounding of zero and
        # take the other side of the ReLU in one of the two runs (the GEMMs in front differ in their bias pass), which moves
        # isolated gradient elements by percents; the mask is covered by the small cases, the sizes by this one
        proj = torch.nn.Sequential(torch.nn.Linear(d0, d1), torch.nn.BatchNorm1d(d1),
                                   torch.nn.Linear(d1, d2), torch.nn.BatchNorm1d(d2),
                                   torch.nn.Linear(d2, d3, bias=False)).to(dev)
    else:
        proj = torch.nn.Sequential(torch.nn.Linear(d0, d1), torch.nn.BatchNorm1d(d1), torch.nn.ReLU(True),
                                   torch.nn.Linear(d1, d2), torch.nn.BatchNorm1d(d2), torch.nn.ReLU(True),
                                   torch.nn.Linear(d2, d3, bias=False)).to(dev)
    with torch.no_grad():
        for m in proj:
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(1.0 + 0.3 * randn((m.num_features,), 3).to(dev))
                m.bias.copy_(0.2 * randn((m.num_features,), 4).to(dev))
                m.running_mean.copy_(0.1 * randn((m.num_features,), 5).to(dev))
    ref = copy.deepcopy(proj)
    a = (randn((n, d0), 41) * 2 + 0.5).to(dev).requires_grad_(True)
    b = (randn((n, d0), 42) - 0.25).to(dev).requires_grad_(True)
    wa, wb = randn((n, d3), 43).to(dev), randn((n, d3), 44).to(dev)
    for _ in range(2):                      # two steps: the running statistics chain through both
        xa, xb = vicreg.project_pair(proj, a, b)
        ra, rb = ref(a), ref(b)
    tol = 2e-5
    scale = max(1.0, ra.abs().max().item())
    assert (xa - ra).abs().max().item() <= tol * scale and (xb - rb).abs().max().item() <= tol * scale
    got = torch.autograd.grad((xa * wa).sum() + (xb * wb).sum(), [a, b] + list(proj.parameters()))
    want = torch.autograd.grad((ra * wa).sum() + (rb * wb).sum(), [a, b] + list(ref.parameters()))
    names = ["a", "b"] + [k for k, _ in proj.named_parameters()]
    for k, g, w in zip(names, got, want):
        # the Linear biases in front of a BatchNorm have a gradient that is zero in exact arithmetic: absolute bound
        bound = 2e-4 * max(1.0, w.abs().max().item())
        d = (g - w).abs()
        assert d.max().item() <= bound, (k, d.max().item(), w.abs().max().item())
    for m, r in zip(proj, ref):
        if isinstance(m, torch.nn.BatchNorm1d):
            assert torch.allclose(m.running_mean, r.running_mean, atol=1e-6, rtol=1e-5)
            assert torch.allclose(m.running_var, r.running_var, atol=1e-6, rtol=1e-5)
            assert int(m.num_batches_tracked) == int(r.num_batches_tracked) == 4
    # the same call with the switch that keeps nn.BatchNorm1d: the torch form of the pair is still there and agrees
    vicreg.PROJECT_PAIR_TORCH = True
    try:
        ta, tb = vicreg.project_pair(copy.deepcopy(ref), a, b)
    finally:
        vicreg.PROJECT_PAIR_TORCH = False
    ra2, rb2 = copy.deepcopy(ref)(a), None
    assert (ta - ra2).abs().max().item() <= tol * scale


@pytest.mark.parametrize("B,dim", [(128, 256), (7, 40)])
def test_paramembed_fused_linear_batchnorm_matches_torch(lib, dev, B, dim):
    """ParamEmbed / AudioRepresentationToParams in training mode on the GPU (Linear -> BatchNorm1d as a GEMM without bias +
    ias_bn1d_groups_* with one row group) against the same modules on nn.BatchNorm1d (reference paramembed.py:20-40,
    audio_to_params.py:16-53): outputs, parameter gradients, running statistics and counters; dropout 0 so that the two
    runs see the same mask."""
    import copy
    from inverse_audio_synthesis_amd import vicreg
    from inverse_audio_synthesis_amd.paramembed import AudioRepresentationToParams, ParamEmbed
    torch.manual_seed(2)
    for mod, x in ((ParamEmbed(78, dim, "nn.BatchNorm1d", 0.0), torch.rand(B, 78)),
                   (AudioRepresentationToParams(78, dim, "nn.BatchNorm1d", 0.0), randn((B, dim), 9))):
        mod = mod.to(dev).train()
        ref = copy.deepcopy(mod)
        x = x.to(dev)
        y = mod(x)
        vicreg.PROJECT_PAIR_TORCH = True
        try:
            r = ref(x)
        finally:
            vicreg.PROJECT_PAIR_TORCH = False
        assert (y - r).abs().max().item() <= 2e-5 * max(1.0, r.abs().max().item())
        w = randn(tuple(r.shape), 10).to(dev)
        got = torch.autograd.grad((y * w).sum(), list(mod.parameters()))
        want = torch.autograd.grad((r * w).sum(), list(ref.parameters()))
        for (k, _), g, t in zip(mod.named_parameters(), got, want):
            assert (g - t).abs().max().item() <= 2e-4 * max(1.0, t.abs().max().item()), k
        for m, t in zip(mod.modules(), ref.modules()):
            if isinstance(m, torch.nn.BatchNorm1d):
                assert torch.allclose(m.running_mean, t.running_mean, atol=1e-6, rtol=1e-5)
                assert torch.allclose(m.running_var, t.running_var, atol=1e-6, rtol=1e-5)
                assert int(m.num_batches_tracked) == int(t.num_batches_tracked) == 1
