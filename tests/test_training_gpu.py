"""End-to-end steps of the two entry points on one GPU with a small model (same code path as the
full-size configs: Voice render -> fused PQMF/normalise -> MobileNetV3 trunk -> projector -> HIP VICReg
loss -> closed-form backward -> bucketed grads -> LARS)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = ["vicreg=fast", "dim=64", "embeddim=256", "vicreg.batch_size=4", "vicreg.mlp=128-128-%d",
         "audio_to_params.batch_size=4", "trainer.log_every=1", "vicreg.checkpoint_every_nbatches=null"]


def test_pretrain_entry_point(lib, dev, tmp_path):
    import pretrain
    hist = pretrain.app(SMALL + ["trainer.max_steps=4", f"trainer.out_dir={tmp_path}"])
    assert len(hist) == 4
    names = {"vicreg/train/loss", "vicreg/train/repr_loss", "vicreg/train/std_loss", "vicreg/train/cov_loss"}
    assert names <= set(hist[0].keys())
    assert all(math.isfinite(h["vicreg/train/loss"]) for h in hist)
    assert (tmp_path / "vicreg-last.ckpt").exists()
    ck = torch.load(tmp_path / "vicreg-last.ckpt", map_location="cpu")
    keys = ck["state_dict"].keys()
    assert "gram.H" in keys and "audio_repr.conv7.weight" in keys and "vicreg.projector.0.weight" in keys


def test_step_gradients_match_torch_reference(lib, dev):
    """One VicregAudioParams step: the loss value and parameter gradients agree with the same step
    computed with the oracle loss (autograd through plain torch ops on the same device)."""
    from inverse_audio_synthesis_amd.config import load_config
    from inverse_audio_synthesis_amd.harness import VicregAudioParams
    from oracle import vicreg_oracle as vo
    from conftest import ROOT
    import os
    cfg = load_config(os.path.join(ROOT, "conf"), "config", SMALL)
    torch.manual_seed(0)
    m = VicregAudioParams(cfg).to(dev).eval()   # eval: no dropout randomness, BN uses running stats
    loss = m.training_step(3)
    loss.backward()
    g_hip = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    m.zero_grad()
    audio, params, _ = m.voice(3)
    x, y = m.forward(audio, params)
    ref = vo.loss(x, y, cfg.vicreg.batch_size, cfg.embeddim)[0]
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 2e-3 * abs(ref.item())
    for n, p in m.named_parameters():
        if p.grad is None:
            continue
        scale = p.grad.abs().max().item()
        assert (g_hip[n] - p.grad).abs().max().item() <= 2e-3 * scale + 1e-7, n


def test_audio_to_params_entry_point(lib, dev, tmp_path):
    import audio_to_params
    hist = audio_to_params.app(SMALL + ["trainer.max_steps=2", f"trainer.out_dir={tmp_path}"])
    assert {"audio_to_params/train/loss", "audio_to_params/train/frozen_vicreg_loss"} <= set(hist[0].keys())
    assert all(math.isfinite(h["audio_to_params/train/loss"]) for h in hist)


def test_audio_to_params_through_synth_mel_l1(lib, dev, tmp_path):
    """audio_to_params.loss=mel_l1: the reference's commented-out objective (audio_to_params.py:56-172) -- the MLP's
    predicted parameters are rendered by the HIP synth and compared with the true audio by mel-L1; the gradient
    reaches the MLP through the synth's HIP backward."""
    import audio_to_params
    hist = audio_to_params.app(SMALL + ["audio_to_params.loss=mel_l1", "trainer.max_steps=3",
                                        f"trainer.out_dir={tmp_path}"])
    assert all("audio_to_params/train/mel_l1_error" in h for h in hist)
    assert all(math.isfinite(h["audio_to_params/train/mel_l1_error"]) for h in hist)

    from inverse_audio_synthesis_amd.config import load_config
    from inverse_audio_synthesis_amd.harness import AudioToParams, VicregAudioParams
    from conftest import ROOT
    import os
    cfg = load_config(os.path.join(ROOT, "conf"), "config", SMALL + ["audio_to_params.loss=mel_l1"])
    model = AudioToParams(cfg, VicregAudioParams(cfg)).to(dev).train()
    loss = model.training_step(3)
    loss.backward()
    grads = [p.grad for p in model.audio_repr_to_params.parameters() if p.requires_grad]
    assert all(g is not None and torch.isfinite(g).all() for g in grads)
    assert any(g.abs().max().item() > 0 for g in grads)
    assert all(p.grad is None for p in model.vicreg.parameters())


def test_audio_to_params_test_step_renders_prediction(lib, dev):
    from inverse_audio_synthesis_amd.config import load_config
    from inverse_audio_synthesis_amd.harness import AudioToParams, VicregAudioParams
    from conftest import ROOT
    import os
    cfg = load_config(os.path.join(ROOT, "conf"), "config", SMALL)
    model = AudioToParams(cfg, VicregAudioParams(cfg)).to(dev).eval()
    with torch.no_grad():
        model.test_step(5)
    true_audio, pred_audio = model.last_predicted_audio
    assert true_audio.shape == (4, 1, 176400) and pred_audio.shape == (4, 176400)
    assert not torch.isnan(pred_audio).any() and pred_audio.abs().max().item() <= 1.0 + 1e-6
    assert not model.voice._frozen, "unfreeze_all_parameters must run after the render"
    # vicreg stays frozen
    assert all(not p.requires_grad for p in model.vicreg.parameters())


def test_retrieval_finds_the_same_voice(lib, dev):
    """evaluate_audio_representations.py:202-231 counterpart: a bank item queried with its own audio is its
    own nearest neighbour at distance ~0."""
    from inverse_audio_synthesis_amd.config import load_config
    from inverse_audio_synthesis_amd.harness import VicregAudioParams
    from inverse_audio_synthesis_amd.retrieval import build_bank, embed_audio, nearest
    from conftest import ROOT
    import os
    cfg = load_config(os.path.join(ROOT, "conf"), "config", SMALL)
    torch.manual_seed(1)
    m = VicregAudioParams(cfg).to(dev)
    bank, params = build_bank(m, [0, 1, 2])
    assert bank.shape == (12, 64) and params.shape == (12, 78)
    audio, _, _ = m.voice(1)
    dist, idx = nearest(embed_audio(m, audio), bank, k=2)
    assert idx[:, 0].tolist() == [4, 5, 6, 7]
    assert dist[:, 0].max().item() <= 1e-3 * max(dist[:, 1].min().item(), 1e-6) + 1e-4


def test_captured_step_reproduces_the_eager_loop(lib, dev, tmp_path):
    """trainer.cuda_graph=true replays render + forward + backward + LARS of a step as one hipGraph; host work (parameter
    sampling, scheduler) stays outside.  Every reduction of the step has a fixed order (round 3: the VICReg backward's
    split-K Gram writes per-slice partials instead of fp32 atomics; the reference trains with deterministic=True,
    pretrain.py:100), so eight steps of the same small run give the eager loop's losses BIT FOR BIT."""
    import pretrain
    args = SMALL + ["trainer.max_steps=8", "param_embed.dropout=0.0"]
    h_e = pretrain.app(args + ["trainer.cuda_graph=false", f"trainer.out_dir={tmp_path / 'eager'}"])
    h_g = pretrain.app(args + ["trainer.cuda_graph=true", f"trainer.out_dir={tmp_path / 'graph'}"])
    assert len(h_e) == len(h_g) == 8
    for a, b in zip(h_e, h_g):
        assert a["lr"] == b["lr"]
        for k in ("vicreg/train/loss", "vicreg/train/repr_loss", "vicreg/train/std_loss", "vicreg/train/cov_loss"):
            assert a[k] == b[k], (k, a, b)
    assert h_e[0]["vicreg/train/loss"] != h_e[-1]["vicreg/train/loss"]          # the run did train
    se = torch.load(tmp_path / "eager" / "vicreg-last.ckpt", map_location="cpu")["state_dict"]
    sg = torch.load(tmp_path / "graph" / "vicreg-last.ckpt", map_location="cpu")["state_dict"]
    for k in ("audio_repr.conv7.weight", "vicreg.projector.0.weight", "vision_model.features.0.1.running_mean"):
        assert torch.equal(se[k], sg[k]), k


def test_pretrain_full_size_steps(lib, dev, tmp_path):
    """BASELINE configs[2] shape end to end through the entry point: batch 128 x 4 s @ 44.1 kHz, dim 1024, embeddim 8192,
    MobileNetV3 trunk, LARS, the step captured as one hipGraph after three eager steps.  Six steps: finite, changing
    losses, the learning-rate warm-up of the scheduler, a checkpoint with the reference's key layout."""
    import pretrain
    hist = pretrain.app(["vicreg.batch_size=128", "trainer.max_steps=6", "trainer.log_every=1",
                         "vicreg.checkpoint_every_nbatches=null", f"trainer.out_dir={tmp_path}"])
    assert len(hist) == 6
    losses = [h["vicreg/train/loss"] for h in hist]
    assert all(math.isfinite(v) for v in losses) and len(set(losses)) == 6
    lrs = [h["lr"] for h in hist]
    assert all(b > a for a, b in zip(lrs, lrs[1:]))                      # linear warm-up
    ck = torch.load(tmp_path / "vicreg-last.ckpt", map_location="cpu")
    sd = ck["state_dict"]
    assert sd["vicreg.projector.0.weight"].shape == (8192, 1024) and sd["audio_repr.conv1.weight"].shape == (1024, 1024, 2, 2)
    assert sd["vision_model.features.1.block.0.1.running_mean"].abs().sum() > 0   # BatchNorm statistics were updated
    assert all(torch.isfinite(v).all() for v in sd.values() if v.is_floating_point())


def test_two_rank_training_keeps_replicas_identical(lib, dev, tmp_path):
    """conf/config.yaml:8 (strategy ddp) -> pretrain.py:97-99, vicreg_audio_params.py:117-120 (sync_dist metrics): two
    freshly spawned ranks (children of this process, never a re-exec of it) share the GPU over gloo and run three
    pretraining steps.  Replicas start from rank 0's parameters although they were built from different seeds, stay
    bit-identical after every step, draw disjoint batch indices, log the mean of the ranks' metrics, and only rank 0
    writes checkpoints."""
    import json
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    overrides = SMALL + ["trainer.max_steps=3", f"trainer.out_dir={tmp_path}", "trainer.cuda_graph=false"]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   IAS_DIST_BACKEND="gloo", IAS_MP_OUT=str(tmp_path), IAS_MP_OVERRIDES=json.dumps(overrides))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_trainer_child.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-2000:] for o in outs)
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    assert [r["world"] for r in recs] == [2, 2]
    assert recs[0]["initial_digest"] == recs[1]["initial_digest"]              # broadcast at construction
    assert len(recs[0]["steps"]) == 3
    for s0, s1 in zip(recs[0]["steps"], recs[1]["steps"]):
        assert s0["digest"] == s1["digest"], f"replicas diverged at step {s0['step']}"
        for k in s0["local"]:
            mean = 0.5 * (s0["local"][k] + s1["local"][k])
            assert abs(s0["reduced"][k] - mean) <= 1e-6 * max(1.0, abs(mean))  # sync_dist=True
            assert s0["reduced"][k] == s1["reduced"][k]
    assert recs[0]["steps"][0]["digest"] != recs[0]["steps"][2]["digest"]      # the parameters did move
    assert len(set(recs[0]["batches"]) | set(recs[1]["batches"])) == 6          # rank-strided, disjoint
    assert recs[0]["checkpoint_written_by_this_rank"] and not recs[1]["checkpoint_written_by_this_rank"]
    assert (tmp_path / "vicreg-last.ckpt").exists()


def test_resume_continues_the_run(lib, dev, tmp_path):
    """Trainer.load_checkpoint restores module, optimizer, scheduler and the step counter: a run cut after two of four
    steps and resumed reproduces the uninterrupted run's remaining steps (same batches, same learning rates, same
    losses), and numbers its steps and checkpoints globally."""
    import os
    from inverse_audio_synthesis_amd.config import load_config
    from inverse_audio_synthesis_amd.harness import VicregAudioParams
    from inverse_audio_synthesis_amd.trainer import Trainer
    from conftest import ROOT
    base = SMALL + ["trainer.cuda_graph=false", "param_embed.dropout=0.0", "vicreg.checkpoint_every_nbatches=1"]

    def make(out):
        cfg = load_config(os.path.join(ROOT, "conf"), "config", base + [f"trainer.out_dir={out}"])
        torch.manual_seed(int(cfg.seed))
        return Trainer(cfg, VicregAudioParams(cfg), stage="vicreg")

    full = make(tmp_path / "full").fit(max_steps=4)
    first = make(tmp_path / "cut")
    first.fit(max_steps=2)
    resumed = make(tmp_path / "cut")
    ck = resumed.load_checkpoint(tmp_path / "cut" / "vicreg-last.ckpt")
    assert ck["step"] == 2 and resumed.start_step == 2
    hist = resumed.fit(max_steps=4)
    assert [h["step"] for h in hist] == [2, 3]
    for h, ref in zip(hist, full[2:]):
        assert h["lr"] == ref["lr"]
        assert abs(h["vicreg/train/loss"] - ref["vicreg/train/loss"]) <= 2e-3 * abs(ref["vicreg/train/loss"])
    assert (tmp_path / "cut" / "vicreg-step000004.ckpt").exists() and not (tmp_path / "cut" / "vicreg-step000005.ckpt").exists()


def test_train_metrics_survive_a_validation_inside_the_captured_loop(lib, dev, tmp_path):
    """trainer.cuda_graph with val_check_interval < max_steps: evaluate() rebinds module.logged to its own tensors; the
    replayed step must log the TRAIN metrics again afterwards (they keep changing), not the last validation batch."""
    import pretrain
    hist = pretrain.app(SMALL + ["trainer.max_steps=8", f"trainer.out_dir={tmp_path}", "trainer.cuda_graph=true",
                                 "vicreg.val_check_interval=5", "vicreg.limit_val_batches=1"])
    train = [h for h in hist if "vicreg/train/loss" in h]
    assert [h["step"] for h in train] == list(range(8))
    after = [h["vicreg/train/loss"] for h in train if h["step"] >= 5]
    assert len(set(after)) == len(after), after          # three different steps, three different losses
    assert any("vicreg/validation/loss" in h for h in hist)


@pytest.mark.parametrize("graph", ["false", "true"])
def test_bn_counters_follow_the_module_to_the_device(lib, dev, tmp_path, graph):
    """The trunk's num_batches_tracked counters are bumped by one multi-tensor add per training step (vision.defer_bn_counters).
    Trainer.__init__ moves the module AFTER construction, which rebinds every buffer: the bump must reach the tensors the
    module holds now, eager and replayed, and a checkpoint must carry the step count (torch.nn.BatchNorm2d semantics,
    /root/reference/vicreg_audio_params.py:52-54 -> torchvision BatchNorm2d)."""
    import pretrain
    steps = 6
    pretrain.app(SMALL + [f"trainer.max_steps={steps}", f"trainer.out_dir={tmp_path}", f"trainer.cuda_graph={graph}"])
    sd = torch.load(tmp_path / "vicreg-last.ckpt", map_location="cpu")["state_dict"]
    counters = {k: int(v) for k, v in sd.items() if k.endswith("num_batches_tracked") and k.startswith("vision_model.")}
    assert len(counters) == 34
    assert set(counters.values()) == {steps}, counters
    # the projector's / paramembed's BatchNorm1d layers count for themselves (two branches through the projector per step)
    other = {k: int(v) for k, v in sd.items() if k.endswith("num_batches_tracked") and not k.startswith("vision_model.")
             and not k.startswith("audio_repr.")}
    assert other and all(v >= steps for v in other.values()), other
