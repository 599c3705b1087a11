#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference's importable modules.

Run ONLY in the build container (needs /root/reference; the GPU box has no reference).
Importable there: pqmf, vicreg, paramembed, audioembed.  Everything
else on the hot path (torchsynth Voice, spectral loss) is absent from the reference tree,
so no golden vector exists for it (parity unpinned).

Only inputs/outputs are stored -- no reference source text.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, REF)

import audioembed as ref_audioembed  # noqa: E402
import paramembed as ref_paramembed  # noqa: E402
import pqmf as ref_pqmf  # noqa: E402
import vicreg as ref_vicreg  # noqa: E402


def randn(shape, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(shape, generator=g)


def checks(t):
    d = t.double()
    return np.array([d.sum().item(), d.abs().sum().item(), (d * d).sum().item()])


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)

    # (1) filter banks
    fb = {}
    for N in (3, 4, 64):
        m = ref_pqmf.PQMF(N=N)
        fb[f"H{N}"] = m.H.numpy()
        fb[f"G{N}"] = m.G.numpy()
        fb[f"updown{N}"] = m.updown_filter.numpy()
    m = ref_pqmf.PQMF(N=8, taps=30, cutoff=0.07, beta=7.0)
    fb["H8_t30"] = m.H.numpy()
    fb["G8_t30"] = m.G.numpy()
    np.savez_compressed(os.path.join(OUT, "pqmf_filters.npz"), **fb)

    # (2) analysis
    an = {}
    x = randn((4, 1, 16000), 101)
    an["small_seed"] = np.array(101)
    an["small_z3"] = ref_pqmf.PQMF(N=3).analysis(x).numpy()
    x = randn((2, 1, 176400), 102)
    z = ref_pqmf.PQMF(N=3).analysis(x)
    an["full_seed"] = np.array(102)
    an["full_z3_shape"] = np.array(z.shape)
    an["full_z3_sub"] = z.flatten()[::97].numpy()
    an["full_z3_checks"] = checks(z)
    z64 = ref_pqmf.PQMF(N=64).analysis(x)
    an["full_z64_shape"] = np.array(z64.shape)
    an["full_z64_sub"] = z64.flatten()[::97].numpy()
    an["full_z64_checks"] = checks(z64)
    # ragged / edge lengths
    for T in (1, 31, 62, 63, 64, 1000, 1001):
        xe = randn((2, 1, T), 200 + T)
        an[f"edge_T{T}_z4"] = ref_pqmf.PQMF(N=4).analysis(xe).numpy()
        an[f"edge_T{T}_z3"] = ref_pqmf.PQMF(N=3).analysis(xe).numpy()
    np.savez_compressed(os.path.join(OUT, "pqmf_analysis.npz"), **an)

    # (3) synthesis(analysis(x))
    sy = {}
    x = randn((2, 1, 4096), 103)
    for N in (3, 4, 64):
        m = ref_pqmf.PQMF(N=N)
        z = m.analysis(x)
        sy[f"z{N}"] = z.numpy()
        sy[f"y{N}"] = m.synthesis(z).numpy()
    sy["seed"] = np.array(103)
    np.savez_compressed(os.path.join(OUT, "pqmf_synthesis.npz"), **sy)

    # (4) VICReg.loss, (5) off_diagonal
    vl = {}
    for (B, D, cfgB, tag) in [(16, 8192, 16, "b16"), (128, 8192, 128, "b128"), (1024, 8192, 1024, "b1024"),
                              (48, 512, 64, "denom_quirk"), (8, 96, 8, "tiny")]:
        cfg = types.SimpleNamespace(dim=32, embeddim=D,
                                    vicreg=types.SimpleNamespace(batch_size=cfgB, mlp="64-64-%d", sim_coeff=25.0,
                                                                 std_coeff=25.0, cov_coeff=1.0))
        model = ref_vicreg.VICReg(cfg, torch.nn.Identity(), torch.nn.Identity())
        x, y = randn((B, D), 300 + B), randn((B, D), 400 + B) * 0.7 + 0.1
        out = model.loss(x, y)
        vl[f"{tag}_meta"] = np.array([B, D, cfgB, 300 + B, 400 + B])
        vl[f"{tag}_out"] = np.array([o.item() for o in out], dtype=np.float64)
        if tag == "tiny":
            vl["tiny_x"], vl["tiny_y"] = x.numpy(), y.numpy()
    vl["offdiag_in"] = np.arange(9, dtype=np.float32).reshape(3, 3)
    vl["offdiag_out"] = ref_vicreg.off_diagonal(torch.arange(9.0).reshape(3, 3)).numpy()
    a = randn((7, 7), 5)
    vl["offdiag7_in"], vl["offdiag7_out"] = a.numpy(), ref_vicreg.off_diagonal(a).numpy()
    np.savez_compressed(os.path.join(OUT, "vicreg_loss.npz"), **vl)

    # (6) small MLPs in eval(): Projector, ParamEmbed
    ml = {}
    cfg = types.SimpleNamespace(dim=32, embeddim=48, vicreg=types.SimpleNamespace(mlp="64-64-%d"))
    torch.manual_seed(7)
    proj = ref_vicreg.Projector(cfg, 32).eval()
    for k, v in proj.state_dict().items():
        ml["proj." + k] = v.numpy()
    xin = randn((5, 32), 8)
    ml["proj_in"], ml["proj_out"] = xin.numpy(), proj(xin).detach().numpy()
    for norm in ("nn.BatchNorm1d", "nn.Identity"):
        torch.manual_seed(9)
        pe = ref_paramembed.ParamEmbed(nparams=78, dim=40, hidden_norm=norm, dropout=0.1).eval()
        tag = "pe_bn" if "Batch" in norm else "pe_id"
        for k, v in pe.state_dict().items():
            ml[f"{tag}." + k] = v.numpy()
        pin = torch.rand((6, 78), generator=torch.Generator().manual_seed(10))
        ml[f"{tag}_in"], ml[f"{tag}_out"] = pin.numpy(), pe(pin).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "mlp_forward.npz"), **ml)

    # (7) AudioEmbedding._preprocess (PQMF(3) -> reshape -> per-channel normalise)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    ae = ref_audioembed.AudioEmbedding(ref_pqmf.PQMF(N=3), vision_model=None,
                                       img_preprocess=lambda t: (t - mean) / std, dim=8)
    x = randn((2, 1, 176400), 104)
    img = ae._preprocess(x)
    pp = {"seed": np.array(104), "shape": np.array(img.shape), "sub": img.flatten()[::89].numpy(),
          "checks": checks(img)}
    np.savez_compressed(os.path.join(OUT, "audioembed_preprocess.npz"), **pp)
    # (8) AudioRepresentationToParams (audio_to_params.py:16-53), eval() forward from a seeded state_dict.  The module
    # audio_to_params.py itself cannot be imported here (its top-level imports need flash / lightning / torchsynth /
    # wandb: ModuleNotFoundError), so only that class definition is taken from the reference file (parsed, compiled
    # and executed in a namespace that provides torch.nn) -- the reference's own code computes the vectors.
    import ast
    tree = ast.parse(open(os.path.join(REF, "audio_to_params.py")).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AudioRepresentationToParams"]
    ns = {"nn": torch.nn, "Tensor": torch.Tensor, "torch": torch}
    exec(compile(ast.Module(body=cls, type_ignores=[]), "audio_to_params.py", "exec"), ns)
    ar = {}
    for norm in ("nn.BatchNorm1d", "nn.Identity"):
        torch.manual_seed(11)
        m = ns["AudioRepresentationToParams"](nparams=78, dim=40, hidden_norm=norm, dropout=0.1)
        tag = "a2p_bn" if "Batch" in norm else "a2p_id"
        if "Batch" in norm:     # non-trivial running statistics
            m.train()
            for i in range(3):
                m(randn((16, 40), 20 + i) * (1.0 + 0.5 * i))
        m.eval()
        for k, v in m.state_dict().items():
            ar[f"{tag}." + k] = v.numpy()
        xin = randn((6, 40), 12)
        ar[f"{tag}_in"], ar[f"{tag}_out"] = xin.numpy(), m(xin).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "audio_repr_to_params.npz"), **ar)
    third_party_goldens()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


def third_party_goldens():
    """The arithmetic of the render and of the spectral losses lives in packages the reference only NAMES
    (requirements.txt: torchsynth, torchaudio; audio_to_params.py:233: auraloss) and this image does not have: those
    stages are "parity unpinned" (DESIGN.md section 2).  The moment one of them imports in the build container, this
    writes the vectors that pin it -- inputs and outputs only, small -- and tests/test_oracle_golden.py /
    tests/test_third_party_golden_gpu.py start using them; nothing but the .npz travels to the GPU box.
      voice_torchsynth.npz   Voice(SynthConfig(...))(batch_idx): params, audio (B = 4 x 1 s @ 16 kHz in full; B = 8 x 4 s @
                             44.1 kHz as a strided subsample + checksums), is_train
      mel_torchaudio.npz     torchaudio.transforms.MelSpectrogram with conf/config.yaml:52-61's settings on seeded noise
      mrstft_auraloss.npz    auraloss.freq.MultiResolutionSTFTLoss() on two seeded signals"""
    made = []
    try:
        from torchsynth.config import SynthConfig
        from torchsynth.synth import Voice
        vg = {}
        for tag, B, sr, sec, idx in (("small", 4, 16000, 1.0, 0), ("head", 8, 44100, 4.0, 3)):
            # batch sizes below torchsynth's reproducibility unit need reproducible=False, as the reference passes it
            # (vicreg_audio_params.py:86-91 reads cfg.torchsynth.reproducible = False)
            voice = Voice(synthconfig=SynthConfig(batch_size=B, reproducible=False, sample_rate=sr, buffer_size_seconds=sec))
            audio, params, is_train = voice(idx)
            vg[f"{tag}_cfg"] = np.array([B, sr, sec, idx], dtype=np.float64)
            vg[f"{tag}_params"] = params.detach().cpu().numpy()
            vg[f"{tag}_is_train"] = is_train.detach().cpu().numpy()
            a = audio.detach().cpu()
            if a.numel() <= 1 << 17:
                vg[f"{tag}_audio"] = a.numpy()
            else:
                vg[f"{tag}_audio_sub"] = a.flatten()[::97].numpy()
                vg[f"{tag}_audio_checks"] = checks(a)
            vg[f"{tag}_param_names"] = np.array([f"{m}.{n}" for (m, n), _ in voice.get_parameters().items()])
        np.savez_compressed(os.path.join(OUT, "voice_torchsynth.npz"), **vg)
        made.append("voice_torchsynth.npz")
    except ImportError as e:
        print(f"[make_golden] torchsynth not importable ({e}): the Voice stays parity-unpinned")
    try:
        import torchaudio
        x = randn((3, 20000), 301) * 0.5
        mel = torchaudio.transforms.MelSpectrogram(sample_rate=44100, n_fft=1024, win_length=None, hop_length=512, center=True,
                                                   pad_mode="reflect", power=2.0, norm="slaney", n_mels=128, mel_scale="htk")
        np.savez_compressed(os.path.join(OUT, "mel_torchaudio.npz"), seed=np.array(301), scale=np.array(0.5),
                            shape=np.array(x.shape), mel=mel(x).numpy(), fb=mel.mel_scale.fb.numpy())
        made.append("mel_torchaudio.npz")
    except ImportError as e:
        print(f"[make_golden] torchaudio not importable ({e}): the mel spectrogram stays parity-unpinned")
    try:
        import auraloss
        a, b = randn((2, 1, 20000), 302) * 0.1, randn((2, 1, 20000), 303) * 0.1
        loss = auraloss.freq.MultiResolutionSTFTLoss()(a, b)
        np.savez_compressed(os.path.join(OUT, "mrstft_auraloss.npz"), seeds=np.array([302, 303]), scale=np.array(0.1),
                            shape=np.array(a.shape), loss=np.array(float(loss)))
        made.append("mrstft_auraloss.npz")
    except ImportError as e:
        print(f"[make_golden] auraloss not importable ({e}): the MR-STFT loss stays parity-unpinned")
    return made


if __name__ == "__main__":
    main()
