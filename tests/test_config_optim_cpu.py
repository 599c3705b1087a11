"""Config composition, LARS and the warmup-cosine schedule (host logic, CPU)."""
import math
import os

import torch

from conftest import ROOT
from inverse_audio_synthesis_amd.config import load_config
from inverse_audio_synthesis_amd.optim import LARS, LinearWarmupCosineAnnealingLR


def test_defaults_match_reference_values():
    c = load_config(os.path.join(ROOT, "conf"))
    assert (c.dim, c.embeddim, c.nparams, c.seed, c.num_batches) == (1024, 8192, 78, 42, 50000000)
    assert (c.torchsynth.rate, c.torchsynth.buffer_size_seconds, c.torchsynth.reproducible) == (44100, 4.0, False)
    v = c.vicreg
    assert (v.batch_size, v.mlp, v.sim_coeff, v.std_coeff, v.cov_coeff) == (16, "8192-8192-%d", 25.0, 25.0, 1.0)
    assert v.optim.name == "lars" and v.optim.args.base_lr == 3.2 and v.optim.args.weight_decay == 1e-6
    assert v.scheduler.args.warmup_epochs == 1000 and v.scheduler.args.max_epochs == 22510
    assert c.audio_to_params.batch_size == 1024 and c.param_embed.hidden_norm == "nn.BatchNorm1d"
    assert (c.mel.n_fft, c.mel.hop_length, c.mel.n_mels, c.mel.power) == (1024, 512, 128, 2.0)


def test_overrides_and_groups():
    c = load_config(os.path.join(ROOT, "conf"), "config", ["vicreg=fast", "vicreg.optim.name=sgd", "dim=256",
                                                           "vicreg.optim.args.lr=0.032", "new.key=[1,2]"])
    assert c.vicreg.batch_size == 1024 and c.vicreg.mlp == "256-256-%d"
    assert c.vicreg.optim.name == "sgd" and c.vicreg.optim.args.lr == 0.032 and c.dim == 256
    assert c.new.key == [1, 2]
    assert c.to_dict()["vicreg"]["optim"]["name"] == "sgd"


def test_warmup_cosine_closed_form():
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=2.0)
    s = LinearWarmupCosineAnnealingLR(opt, warmup_epochs=5, max_epochs=25, warmup_start_lr=0.0, eta_min=0.1)
    lrs = [opt.param_groups[0]["lr"]]
    for _ in range(25):
        s.step()
        lrs.append(opt.param_groups[0]["lr"])
    assert lrs[0] == 0.0 and abs(lrs[4] - 2.0) < 1e-12 and abs(lrs[5] - 2.0) < 1e-12
    assert abs(lrs[15] - (0.1 + 0.5 * 1.9 * (1 + math.cos(math.pi * 10 / 20)))) < 1e-12
    assert abs(lrs[25] - 0.1) < 1e-12
    assert all(a >= b - 1e-12 for a, b in zip(lrs[5:], lrs[6:]))


def test_lars_matches_single_tensor_formula():
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.zeros(2))]
    ref = [p.detach().clone() for p in ps]
    opt = LARS(ps, lr=0.5, weight_decay=1e-2, momentum=0.9)
    bufs = [None] * 3
    for step in range(3):
        for i, p in enumerate(ps):
            p.grad = torch.randn_like(p, generator=torch.Generator().manual_seed(10 * step + i)) if False else \
                torch.randn(p.shape, generator=torch.Generator().manual_seed(10 * step + i))
        grads = [p.grad.clone() for p in ps]
        opt.step()
        for i, (w, g) in enumerate(zip(ref, grads)):
            pn, gn = w.norm(), g.norm()
            d = g + 1e-2 * w
            if pn != 0 and gn != 0:
                d = d * (pn / (gn + pn * 1e-2 + 1e-8) * 0.001)
            bufs[i] = d.clone() if bufs[i] is None else bufs[i] * 0.9 + d
            ref[i] = w - 0.5 * bufs[i]
        for p, w in zip(ps, ref):
            assert torch.allclose(p.detach(), w, atol=1e-6)


def test_lars_skips_weight_decay_when_a_norm_is_zero():
    """flash's LARS adds weight decay only where p_norm != 0 and g_norm != 0: a parameter whose gradient is exactly
    zero must not move (plain `g + wd p` would shrink it)."""
    p = torch.nn.Parameter(torch.ones(3))
    q = torch.nn.Parameter(torch.zeros(3))
    opt = LARS([p, q], lr=1.0, weight_decay=0.1)
    p.grad = torch.zeros(3)
    q.grad = torch.ones(3)
    opt.step()
    assert torch.equal(p.detach(), torch.ones(3))          # g_norm == 0: no decay, no update
    assert torch.allclose(q.detach(), -torch.ones(3))      # p_norm == 0: plain gradient step, ratio 1


def test_split_indices_draw_without_replacement():
    from inverse_audio_synthesis_amd.trainer import split_indices, split_sizes
    n, nt = 1000, 10
    sz = split_sizes(n, nt)
    tr = split_indices(n, nt, 42, sz["train"][1] - sz["train"][0], "train")
    assert sorted(tr) == list(range(*sz["train"])), "one epoch visits every train index exactly once"
    va = split_indices(n, nt, 42, 50, "val")
    te = split_indices(n, nt, 42, nt, "test")
    assert len(set(va)) == 50 and all(sz["val"][0] <= i < sz["val"][1] for i in va)
    assert sorted(te) == list(range(n - nt, n))
    # rank-strided: two ranks see disjoint indices, together the one-rank sequence
    r0 = split_indices(n, nt, 42, 20, "train", 0, 2)
    r1 = split_indices(n, nt, 42, 20, "train", 1, 2)
    assert not set(r0) & set(r1) and sorted(r0 + r1) == sorted(tr[:40])
    assert split_indices(n, nt, 43, 20, "train") != tr[:20]
    # the full-size universe is addressed lazily
    big = split_indices(50_000_000, 1, 42, 5, "train")
    assert len(set(big)) == 5 and all(0 <= i < 45_000_000 for i in big)


def test_reference_shaped_checkpoint_loads():
    """A state_dict shaped like the reference's Lightning vicreg.ckpt (with torchvision classifier keys and
    torchsynth voice.* parameter tensors) loads into VicregAudioParams; only the documented extras are dropped."""
    import warnings
    from inverse_audio_synthesis_amd.harness import VicregAudioParams
    from inverse_audio_synthesis_amd.trainer import load_reference_state_dict
    cfg = load_config(os.path.join(ROOT, "conf"), "config", ["vicreg=fast", "dim=32", "embeddim=64",
                                                             "vicreg.batch_size=2", "vicreg.mlp=48-48-%d"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = VicregAudioParams(cfg)
    sd = {k: torch.randn_like(v) if v.is_floating_point() else v.clone() for k, v in m.state_dict().items()}
    sd["vision_model.classifier.0.weight"] = torch.zeros(1024, 576)
    sd["audio_repr.vision_model.classifier.3.bias"] = torch.zeros(1000)
    sd["voice.adsr_1.torchparameters.attack"] = torch.zeros(2)
    sd["voice.noise.noise"] = torch.zeros(2, 8)
    missing, unexpected, dropped = load_reference_state_dict(m, sd)
    assert missing == [] and unexpected == [] and len(dropped) == 4
    k = "vicreg.projector.0.weight"
    assert torch.equal(m.state_dict()[k], sd[k])
