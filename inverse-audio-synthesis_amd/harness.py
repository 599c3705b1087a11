"""Training-step modules: counterparts of the reference's two LightningModules, without Lightning.

``VicregAudioParams``  /root/reference/vicreg_audio_params.py:33-165  (voice -> audio/param backbones ->
                       shared projector -> VICReg loss; metric names vicreg/{name}/{loss,repr_loss,std_loss,cov_loss})
``AudioToParams``      /root/reference/audio_to_params.py:177-312     (frozen VICReg -> MLP -> params;
                       loss = MSE of projected param embeddings; test step re-renders predicted params).
                       ``audio_to_params.loss: mel_l1`` switches to the objective of the commented-out
                       ``train_audio_to_params_through_torchsynth`` (audio_to_params.py:56-172): predicted params ->
                       Voice render -> mel-L1 against the true audio, differentiated through the HIP synth.
Both keep the reference's attribute names (gram, vision_model, img_preprocess, paramembed, audio_repr,
vicreg, synthconfig, voice / audio_repr_to_params) so state_dict keys line up.  ``_step`` returns the
loss and fills ``self.logged`` with the metrics the reference passes to ``self.log``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .audioembed import AudioEmbedding, ChannelNormalize
from .optim import LARS, LinearWarmupCosineAnnealingLR
from .paramembed import AudioRepresentationToParams, ParamEmbed
from .pqmf import PQMF
from .vicreg import VICReg
from .vision import mobilenet_v3_small
from .spectral import MelSpectrogramL1
from .voice import SynthConfig, Voice


def _batch_num(batch):
    if torch.is_tensor(batch):
        assert batch.numel() == 1
        return int(batch.reshape(-1)[0].item())
    return int(batch)


class VicregAudioParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.gram = PQMF(N=3)  # 3 bands = RGB channels of the "image"
        self.vision_model = mobilenet_v3_small(pretrained=cfg.vicreg.pretrained_vision_model)
        self.img_preprocess = ChannelNormalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
        self.paramembed = ParamEmbed(nparams=cfg.nparams, dim=cfg.dim, hidden_norm=cfg.param_embed.hidden_norm,
                                     dropout=cfg.param_embed.dropout)
        self.audio_repr = AudioEmbedding(self.gram, self.vision_model, img_preprocess=self.img_preprocess, dim=cfg.dim)
        gather = bool(getattr(getattr(cfg, "trainer", None), "gather_embeddings", False)) if hasattr(cfg, "trainer") else False
        self.vicreg = VICReg(cfg=cfg, backbone_audio=self.audio_repr, backbone_param=self.paramembed,
                             gather_distributed=gather)
        self.synthconfig = SynthConfig(batch_size=cfg.vicreg.batch_size, reproducible=cfg.torchsynth.reproducible,
                                       sample_rate=cfg.torchsynth.rate,
                                       buffer_size_seconds=cfg.torchsynth.buffer_size_seconds)
        self.voice = Voice(synthconfig=self.synthconfig)
        self.logged = {}
        # the trunk's 34 num_batches_tracked counters are bumped by ONE multi-tensor add per training forward of THIS
        # module (installed here, before any forward, so the first step is not counted twice; a training-mode forward of
        # vision_model outside _step must call self._bump_bn() itself)
        from .vision import defer_bn_counters
        defer_bn_counters(self.vision_model)

    def _bump_bn(self):
        # the counters are resolved at call time: Trainer's module.to(device) rebinds every buffer after construction
        from .vision import bump_bn_counters
        bump_bn_counters(self.vision_model)

    def forward(self, audio, params):
        assert audio.ndim == 2 and params.ndim == 2 and audio.shape[0] == params.shape[0]
        return self.vicreg(audio=audio.unsqueeze(1), params=params)

    def _step(self, name, batch, batch_idx=None):
        with torch.no_grad():
            # batch None: the parameters already stored in the voice (Trainer's captured step samples them on the host,
            # outside the graph, with voice.randomize(batch))
            audio, params, _is_train = self.voice(None if batch is None else _batch_num(batch))
        x, y = self.forward(audio, params)
        if self.training:
            self._bump_bn()
        loss, repr_loss, std_loss, cov_loss = self.vicreg.loss(x, y)
        self.logged = {f"vicreg/{name}/loss": loss.detach(), f"vicreg/{name}/repr_loss": repr_loss.detach(),
                       f"vicreg/{name}/std_loss": std_loss.detach(), f"vicreg/{name}/cov_loss": cov_loss.detach()}
        return loss

    def training_step(self, batch, batch_idx=None):
        return self._step("train", batch, batch_idx)

    def validation_step(self, batch, batch_idx=None):
        return self._step("validation", batch, batch_idx)

    def configure_optimizers(self):
        v = self.cfg.vicreg
        params = [p for p in self.parameters() if p.requires_grad]
        if v.optim.name == "sgd":
            opt = torch.optim.SGD(params, lr=v.optim.args.lr)
        elif v.optim.name == "lars":
            opt = LARS(params, weight_decay=v.optim.args.weight_decay, lr=v.batch_size / 256 * v.optim.args.base_lr)
        else:
            assert False, v.optim.name
        assert v.scheduler.name == "LinearWarmupCosineAnnealingLR", v.scheduler.name
        sched = LinearWarmupCosineAnnealingLR(opt, **dict(v.scheduler.args))
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sched, "interval": "step", "frequency": 1}}


class AudioToParams(nn.Module):
    def __init__(self, cfg, vicreg):
        super().__init__()
        self.cfg = cfg
        self.vicreg = vicreg
        self._freeze_vicreg()
        a = cfg.audio_to_params
        self.audio_repr_to_params = AudioRepresentationToParams(nparams=cfg.nparams, dim=cfg.dim,
                                                                hidden_norm=a.hidden_norm, dropout=a.dropout)
        self.voice = Voice(synthconfig=SynthConfig(batch_size=a.batch_size, reproducible=cfg.torchsynth.reproducible,
                                                   sample_rate=cfg.torchsynth.rate,
                                                   buffer_size_seconds=cfg.torchsynth.buffer_size_seconds))
        self.loss_kind = a.get("loss", "embedding_mse")
        assert self.loss_kind in ("embedding_mse", "mel_l1"), self.loss_kind
        if self.loss_kind == "mel_l1":
            m = cfg.mel
            self.mel_l1 = MelSpectrogramL1(sample_rate=cfg.torchsynth.rate, n_fft=m.n_fft, win_length=m.win_length,
                                           hop_length=m.hop_length, center=m.center, pad_mode=m.pad_mode, power=m.power,
                                           norm=m.norm, onesided=m.onesided, n_mels=m.n_mels, mel_scale=m.mel_scale)
        self.logged = {}
        self.last_predicted_audio = None

    def _freeze_vicreg(self):
        for p in self.vicreg.parameters():
            p.requires_grad_(False)
        self.vicreg.eval()

    def train(self, mode=True):
        super().train(mode)
        self.vicreg.eval()  # the pretrained model stays in eval mode (audio_to_params.py:211-212)
        return self

    def _step(self, batch, batch_idx, name):
        self._freeze_vicreg()
        net = self.vicreg.vicreg
        with torch.no_grad():
            audio, params, _ = self.vicreg.voice(_batch_num(batch))
            audio = audio.unsqueeze(1)
            true_params_embedding = net.projector(net.backbone_param(params))
            audio_repr = net.backbone_audio(audio)
            true_audio_embedding = net.projector(audio_repr)
        predicted_params = self.audio_repr_to_params(audio_repr)
        predicted_params_embedding = net.projector(net.backbone_param(predicted_params))
        repr_loss = F.mse_loss(true_params_embedding, predicted_params_embedding)
        frozen_vicreg_loss = F.mse_loss(true_params_embedding, true_audio_embedding)
        self.logged = {f"audio_to_params/{name}/loss": repr_loss.detach(),
                       f"audio_to_params/{name}/frozen_vicreg_loss": frozen_vicreg_loss.detach()}
        loss = repr_loss
        if self.loss_kind == "mel_l1":
            # audio =(vicreg)=> repr =(MLP)=> params =(synth)=> audio, true vs predicted mel (audio_to_params.py:66-70,150-153)
            predicted_audio = self.voice.render(predicted_params)
            loss = self.mel_l1(predicted_audio, target_audio=audio.squeeze(1))
            self.logged[f"audio_to_params/{name}/mel_l1_error"] = loss.detach()
        if name == "test":
            # set -> freeze -> voice(None) -> unfreeze, exactly the sequence of audio_to_params.py:240-257
            for (mod, pname), value in zip(self.voice.get_parameters().keys(), predicted_params.T):
                getattr(self.voice, mod).set_parameter_0to1(pname, value)
            with torch.no_grad():
                self.voice.freeze_parameters(self.voice.get_parameters().keys())
                predicted_audio, _pp, _it = self.voice(None)
                self.voice.unfreeze_all_parameters()
            self.last_predicted_audio = (audio.detach(), predicted_audio.detach())
        return loss

    def training_step(self, batch, batch_idx=None):
        return self._step(batch, batch_idx, "train")

    def test_step(self, batch, batch_idx=None):
        return self._step(batch, batch_idx, "test")

    def configure_optimizers(self):
        a = self.cfg.audio_to_params
        params = [p for p in self.parameters() if p.requires_grad]
        if a.optim.name == "sgd":
            return torch.optim.SGD(params, **dict(a.optim.args))
        assert a.optim.name == "lars", a.optim.name
        return LARS(params, weight_decay=a.optim.args.weight_decay, lr=a.batch_size / 256 * a.optim.args.base_lr)
