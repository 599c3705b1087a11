"""``SynthConfig`` / ``Voice`` look-alikes of the torchsynth objects the reference uses.

Reference call sites (the classes themselves live in the absent third-party
``torchsynth``, /root/reference/requirements.txt:1):
  SynthConfig(batch_size=, reproducible=, sample_rate=, buffer_size_seconds=)
      /root/reference/vicreg_audio_params.py:86-91, audio_to_params.py:196-201
  Voice(synthconfig=), .to(device), voice(batch_idx) -> (audio, params, is_train)
      /root/reference/vicreg_audio_params.py:92-94,114; pretrain.py:75
  voice.get_parameters(), getattr(voice, module).set_parameter_0to1(name, value),
  voice.freeze_parameters(keys), voice.unfreeze_all_parameters(), voice(None)
      /root/reference/audio_to_params.py:240-257

The render itself is the HIP path (csrc/voice_kernels.hip); parameter sampling stays on
the host with torch's CPU generator so that seeds mean the same thing as in the oracle.
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from . import voice_spec as S

_CHECK_STATUS = os.environ.get("IAS_CHECK_STATUS", "0") not in ("", "0")


class SynthConfig:
    def __init__(self, batch_size=128, sample_rate=44100, buffer_size_seconds=4.0,
                 control_rate=S.CONTROL_RATE, reproducible=True, no_grad=True, debug=False, eps=S.EPS):
        self.batch_size = int(batch_size)
        self.sample_rate = int(sample_rate)
        self.buffer_size_seconds = float(buffer_size_seconds)
        self.buffer_size = int(self.buffer_size_seconds * self.sample_rate)
        self.control_rate = int(control_rate)
        self.control_buffer_size = int(self.buffer_size_seconds * self.control_rate)
        self.reproducible = bool(reproducible)
        self.no_grad = bool(no_grad)
        self.debug = bool(debug)
        self.eps = float(eps)
        # the control rate is a kernel argument (a render whose control window per tile or whose samples per control
        # interval do not fit the kernels' LDS stages is refused by the C ABI with IAS_ERR_UNSUPPORTED); eps is compiled in
        assert self.control_rate > 0 and self.control_buffer_size > 1, "control_rate * buffer_size_seconds must exceed 1"
        assert self.eps == S.EPS, "the HIP control kernels are built for eps=1e-6"


class _ModuleView:
    """What ``getattr(voice, module_name)`` returns: per-module parameter access."""

    def __init__(self, voice, name):
        self._voice = voice
        self._name = name

    def set_parameter_0to1(self, parameter_id, value):
        self._voice._set_column(S.INDEX[(self._name, parameter_id)], value)

    def get_parameter_0to1(self, parameter_id):
        return self._voice.params01[:, S.INDEX[(self._name, parameter_id)]]


def sample_params01(batch_size, batch_idx):
    """Voice.randomize(seed=batch_idx) on the host: [B,78] uniforms, registration order.

    torchsynth draws sub-batches of 32 voices (seed = global sub-batch number) and assigns
    the columns in sorted-name order; a batch that is not a multiple of 32 is one block
    seeded with batch_idx.
    """
    names = [f"{m}.{n}" for (m, n, *_r) in S.PARAMS]
    order = sorted(range(S.NPARAMS), key=lambda i: names[i])
    g = torch.Generator(device="cpu")
    sub = S.REPRODUCIBLE_SUBBATCH
    if batch_size % sub == 0:
        blocks = []
        for i in range(batch_size // sub):
            g.manual_seed(int(batch_idx) * (batch_size // sub) + i)
            blocks.append(torch.rand((sub, S.NPARAMS), generator=g))
        drawn = torch.cat(blocks, 0)
    else:
        g.manual_seed(int(batch_idx))
        drawn = torch.rand((batch_size, S.NPARAMS), generator=g)
    inv = torch.empty(S.NPARAMS, dtype=torch.long)
    inv[torch.tensor(order)] = torch.arange(S.NPARAMS)
    # index_select, not drawn[:, inv]: advanced indexing of this 128 x 78 tensor goes through the intra-op thread pool and
    # took 30-50 ms on the host (8 us this way) -- it was what bounded a replayed pretraining step
    return drawn.index_select(1, inv)


class Voice(nn.Module):
    def __init__(self, synthconfig=None, nebula="default"):
        super().__init__()
        self.synthconfig = synthconfig if synthconfig is not None else SynthConfig()
        c = self.synthconfig
        g = torch.Generator(device="cpu").manual_seed(S.NOISE_SEED)
        noise = torch.rand((c.batch_size, c.buffer_size), generator=g) * 2.0 - 1.0
        self.register_buffer("noise", noise, persistent=False)
        self.register_buffer("params01", torch.full((c.batch_size, S.NPARAMS), 0.5), persistent=False)
        self._frozen = set()
        self._workspace = None
        # 0: the tested arithmetic contract (oracle "cr"); 1: hardware fp32 exp2 on the pitch path (A/B measurement)
        self.math_mode = 0
        for mod, _plist in S.MODULES:
            object.__setattr__(self, "_view_" + mod, _ModuleView(self, mod))

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            view = self.__dict__.get("_view_" + name)
            if view is None:
                raise
            return view

    # ---- torchsynth-style parameter API ----
    @property
    def batch_size(self):
        return self.synthconfig.batch_size

    def get_parameters(self, include_frozen=True):
        out = OrderedDict()
        for i, (mod, name, *_r) in enumerate(S.PARAMS):
            if include_frozen or (mod, name) not in self._frozen:
                out[(mod, name)] = self.params01[:, i]
        return out

    def freeze_parameters(self, keys):
        for k in keys:
            self._frozen.add(tuple(k))

    def unfreeze_all_parameters(self):
        self._frozen.clear()

    def set_parameters01(self, params01):
        assert params01.shape == self.params01.shape
        self.params01.copy_(params01.detach().to(self.params01.dtype))

    def _set_column(self, idx, value):
        value = torch.as_tensor(value, dtype=torch.float32, device=self.params01.device)
        assert value.numel() in (1, self.batch_size)
        self.params01[:, idx] = value.detach().reshape(-1)

    def randomize(self, seed):
        new = sample_params01(self.batch_size, seed).to(self.params01.device)
        if self._frozen:
            keep = torch.tensor([(m, n) in self._frozen for (m, n, *_r) in S.PARAMS], device=new.device)
            new = torch.where(keep.unsqueeze(0), self.params01, new)
        self.params01.copy_(new)

    def _is_train(self, batch_idx):
        B = self.batch_size
        if batch_idx is None:
            return torch.ones(B, dtype=torch.bool, device=self.params01.device)
        idx = torch.arange(B * int(batch_idx), B * (int(batch_idx) + 1))
        return ((idx // S.REPRODUCIBLE_SUBBATCH) % 10 != 9).to(self.params01.device)

    # ---- render ----
    def render(self, params01=None, normalize=True):
        """HIP render of ``params01`` ([B,78] in 0..1, default: the stored ones) -> audio [B,T].

        Differentiable with respect to ``params01`` when it requires grad (``voice_grad.py``)."""
        p = self.params01 if params01 is None else params01
        if torch.is_grad_enabled() and p.requires_grad:
            from .voice_grad import render_with_grad
            return render_with_grad(self, p, normalize)
        return self._render_nograd(p, normalize)

    def _render_nograd(self, p, normalize=True):
        c = self.synthconfig
        p = p.detach().to(torch.float32).contiguous()
        assert p.shape == (c.batch_size, S.NPARAMS)
        lib = _lib.load()
        _lib.require_f32(p, self.noise)
        need = lib.ias_voice_workspace_bytes(c.batch_size, c.buffer_size, c.control_buffer_size)
        if need < 0:
            _lib.check(int(need), "ias_voice_workspace_bytes")
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != p.device:
            self._workspace = torch.zeros(int(need), dtype=torch.uint8, device=p.device)   # (zeroed once: sticky status word)
        audio = torch.empty((c.batch_size, c.buffer_size), dtype=torch.float32, device=p.device)
        st = lib.ias_voice_render(_lib.ptr(p), _lib.ptr(self.noise), _lib.ptr(audio), _lib.ptr(self._workspace),
                                  self._workspace.numel(), c.batch_size, c.buffer_size, c.control_buffer_size,
                                  c.sample_rate, c.control_rate, 1 if normalize else 0, self.math_mode, _lib.stream())
        _lib.check(st, "ias_voice_render")
        self._check_chain(self._workspace)
        return audio

    def new_workspace(self, device=None):
        """A private workspace for double-buffered pipelines (see ``render_control`` / ``render_audio``)."""
        c = self.synthconfig
        need = int(_lib.load().ias_voice_workspace_bytes(c.batch_size, c.buffer_size, c.control_buffer_size))
        return torch.zeros(need, dtype=torch.uint8, device=device or self.params01.device)   # (zeroed once: sticky status word)

    def render_control(self, workspace, params01=None):
        """Control-rate pass only (78 params -> control signals + per-voice constants) into ``workspace``."""
        c = self.synthconfig
        p = (self.params01 if params01 is None else params01).detach().to(torch.float32).contiguous()
        st = _lib.load().ias_voice_control_ws(_lib.ptr(p), _lib.ptr(workspace), workspace.numel(), c.batch_size,
                                              c.buffer_size, c.control_buffer_size, c.control_rate, _lib.stream())
        _lib.check(st, "ias_voice_control_ws")

    def clear_chain(self, workspace):
        """Re-zero the polled words of the audio-rate kernel in ``workspace`` on the current stream (what ``render_audio``
        starts with; pipelines issue it ahead and call ``render_audio(..., precleared=True)``).  The row peaks of the
        previous render are among those words: their readers must be done."""
        c = self.synthconfig
        st = _lib.load().ias_voice_stage(2, self.math_mode, None, None, _lib.ptr(workspace), workspace.numel(), c.batch_size,
                                         c.buffer_size, c.control_buffer_size, c.sample_rate, _lib.stream())
        _lib.check(st, "ias_voice_stage(clear)")

    def render_audio(self, workspace, out=None, on_stage=None, normalize=True, precleared=False):
        """Audio-rate pass (+ normalise) from a workspace ``render_control`` has filled -> audio [B,T]."""
        c = self.synthconfig
        lib = _lib.load()
        if out is None:
            audio = torch.empty((c.batch_size, c.buffer_size), dtype=torch.float32, device=workspace.device)
        else:
            assert out.shape == (c.batch_size, c.buffer_size) and out.dtype == torch.float32 and out.is_contiguous()
            audio = out
        hook = on_stage or (lambda name, phase: None)
        for stage, name in enumerate(("oscillators", "normalize")[: 2 if normalize else 1]):
            hook(name, "begin")
            st = lib.ias_voice_stage(3 if (stage == 0 and precleared) else stage, self.math_mode, _lib.ptr(self.noise),
                                     _lib.ptr(audio), _lib.ptr(workspace),
                                     workspace.numel(),
                                     c.batch_size, c.buffer_size, c.control_buffer_size, c.sample_rate, _lib.stream())
            _lib.check(st, f"ias_voice_stage({name})")
            hook(name, "end")
        self._check_chain(workspace)
        return audio

    def render_staged(self, params01=None, on_stage=None, out=None):
        """The same render issued stage by stage (control, oscillators, normalise);
        ``on_stage(name, phase)`` is called with phase "begin"/"end" around each (bench event hooks).
        ``out``: optional preallocated [B,T] fp32 buffer (double-buffered pipelines)."""
        p = self.params01 if params01 is None else params01
        if self._workspace is None or self._workspace.device != p.device:
            self._workspace = self.new_workspace(p.device)
        hook = on_stage or (lambda name, phase: None)
        hook("control", "begin")
        self.render_control(self._workspace, p)
        hook("control", "end")
        return self.render_audio(self._workspace, out=out, on_stage=on_stage)

    def peaks_view(self, workspace=None):
        """The row peaks [B] (fp32, max |mix| before normalisation) of the last render INSIDE its workspace, as a
        tensor view (no copy): the ``rowpeak`` argument of ``PQMF.analysis`` / ``MelSpectrogramL1`` when the render was
        issued with ``normalize=False`` and the normalisation is folded into the consumers."""
        c = self.synthconfig
        ws = self._workspace if workspace is None else workspace
        off = int(_lib.load().ias_voice_peaks_offset(c.batch_size, c.buffer_size, c.control_buffer_size))
        return ws[off:off + 4 * c.batch_size].view(torch.float32)

    def read_peaks(self):
        """Row peaks max|mix| [B] of the last render (before normalisation)."""
        c = self.synthconfig
        pk = torch.empty(c.batch_size, dtype=torch.float32, device=self._workspace.device)
        _lib.check(_lib.load().ias_voice_read_peaks(_lib.ptr(self._workspace), _lib.ptr(pk), c.batch_size, c.buffer_size,
                                                    c.control_buffer_size, _lib.stream()), "ias_voice_read_peaks")
        return pk

    def chain_status(self, workspace=None):
        """0 if the last render into ``workspace`` (default: the module's own) completed its cross-tile scan; 1 if a
        bounded wait for a predecessor tile expired.  Synchronises (reads one word back)."""
        c = self.synthconfig
        ws = self._workspace if workspace is None else workspace
        st = torch.zeros(1, dtype=torch.int32, device=ws.device)
        _lib.check(_lib.load().ias_voice_read_status(_lib.ptr(ws), _lib.ptr(st), c.batch_size,
                                                     c.buffer_size, c.control_buffer_size, _lib.stream()),
                   "ias_voice_read_status")
        return int(st.item())

    def chain_status_sticky(self, workspace=None, clear=True):
        """Non-zero if ANY render into ``workspace`` (default: the module's own) since the last cleared look lost a tile
        (``ias_voice_read_status_sticky``): what a loop that checks every N steps reads -- also behind a replayed hipGraph,
        where no Python runs per step.  Synchronises (reads one word back)."""
        c = self.synthconfig
        ws = self._workspace if workspace is None else workspace
        if ws is None:
            return 0
        st = torch.zeros(1, dtype=torch.int32, device=ws.device)
        _lib.check(_lib.load().ias_voice_read_status_sticky(_lib.ptr(ws), _lib.ptr(st), c.batch_size, c.buffer_size,
                                                            c.control_buffer_size, 1 if clear else 0, _lib.stream()),
                   "ias_voice_read_status_sticky")
        return int(st.item())

    def _check_chain(self, workspace):
        """An expired wait is never silent: the kernel turns that tile's audio into NaN (so any loss computed from it
        is NaN), and with IAS_CHECK_STATUS=1 every render also reads the status word back and raises (this
        synchronises, so it is opt-in and skipped while a hipGraph is being captured)."""
        if _CHECK_STATUS and not torch.cuda.is_current_stream_capturing() and self.chain_status(workspace) != 0:
            raise RuntimeError("voice render: a tile's bounded wait for its predecessors expired "
                               "(ias_voice_read_status != 0); the audio of that tile is NaN")

    def control_debug(self, params01=None):
        """Control-rate intermediates [B,10,Tc] (envelopes, LFO phases, LFO outputs) for tests."""
        c = self.synthconfig
        p = (self.params01 if params01 is None else params01).detach().to(torch.float32).contiguous()
        lib = _lib.load()
        ctrl = torch.empty((c.batch_size, 5, c.control_buffer_size), dtype=torch.float32, device=p.device)
        vconst = torch.empty((c.batch_size, 16), dtype=torch.float32, device=p.device)
        dbg = torch.empty((c.batch_size, 10, c.control_buffer_size), dtype=torch.float32, device=p.device)
        env = torch.empty((c.batch_size, 8, c.control_buffer_size), dtype=torch.float32, device=p.device)
        st = lib.ias_voice_control_debug(_lib.ptr(p), _lib.ptr(ctrl), _lib.ptr(vconst), _lib.ptr(env), _lib.ptr(dbg), c.batch_size,
                                         c.control_buffer_size, c.control_rate, _lib.stream())
        _lib.check(st, "ias_voice_control_debug")
        return dbg

    def rendered_control(self, workspace=None):
        """(ctrl [B,5,Tc] fp32, vconst [B,16] fp32) copied out of the workspace the last render used: the control
        signals and per-voice constants of THAT render, for its backward (no second control pass)."""
        c = self.synthconfig
        ws = self._workspace if workspace is None else workspace
        lib = _lib.load()
        oc = int(lib.ias_voice_ctrl_offset(c.batch_size, c.buffer_size, c.control_buffer_size))
        ov = int(lib.ias_voice_vconst_offset(c.batch_size, c.buffer_size, c.control_buffer_size))
        nc = c.batch_size * 5 * c.control_buffer_size * 4
        ctrl = ws[oc:oc + nc].view(torch.float32).reshape(c.batch_size, 5, c.control_buffer_size).clone()
        vconst = ws[ov:ov + c.batch_size * 64].view(torch.float32).reshape(c.batch_size, 16).clone()
        return ctrl, vconst

    def saved_for_backward(self, with_peaks=True, workspace=None):
        """(ctrl [B,5,Tc], vconst [B,16], peaks [B] or None) of the last render, out of its workspace in one launch:
        what the render's autograd node keeps (``rendered_control`` + ``read_peaks`` are three copies)."""
        c = self.synthconfig
        ws = self._workspace if workspace is None else workspace
        ctrl = torch.empty((c.batch_size, 5, c.control_buffer_size), dtype=torch.float32, device=ws.device)
        vconst = torch.empty((c.batch_size, 16), dtype=torch.float32, device=ws.device)
        peaks = torch.empty(c.batch_size, dtype=torch.float32, device=ws.device) if with_peaks else None
        _lib.check(_lib.load().ias_voice_save_for_backward(_lib.ptr(ws), _lib.ptr(ctrl), _lib.ptr(vconst), _lib.ptr(peaks),
                                                           c.batch_size, c.buffer_size, c.control_buffer_size,
                                                           _lib.stream()), "ias_voice_save_for_backward")
        return ctrl, vconst, peaks

    def control_signals(self, params01=None):
        """Mod-matrix outputs [B,5,Tc] of the control-rate kernel (diagnostics / tests)."""
        c = self.synthconfig
        p = (self.params01 if params01 is None else params01).detach().to(torch.float32).contiguous()
        lib = _lib.load()
        ctrl = torch.empty((c.batch_size, 5, c.control_buffer_size), dtype=torch.float32, device=p.device)
        vconst = torch.empty((c.batch_size, 16), dtype=torch.float32, device=p.device)
        env = torch.empty((c.batch_size, 8, c.control_buffer_size), dtype=torch.float32, device=p.device)
        st = lib.ias_voice_control(_lib.ptr(p), _lib.ptr(ctrl), _lib.ptr(vconst), _lib.ptr(env), c.batch_size,
                                   c.control_buffer_size, c.control_rate, _lib.stream())
        _lib.check(st, "ias_voice_control")
        return ctrl, vconst

    def forward(self, batch_idx=None):
        if batch_idx is not None:
            self.randomize(batch_idx)
        audio = self.render()
        return audio, self.params01.clone(), self._is_train(batch_idx)
