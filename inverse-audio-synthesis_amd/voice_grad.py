"""Gradient of the Voice render with respect to the normalised parameters.

The reference never got this far: the audio -> params -> synth -> mel-L1 training loop is the commented-out block
/root/reference/audio_to_params.py:56-172 (torchsynth's own modules are differentiable torch code, which is what
that block relied on).  Here the render is a hand-written HIP kernel, so its adjoint is one too:

* audio rate ([B,T] work): ``csrc/voice_grad_kernels.hip`` (``ias_voice_backward``) turns d loss / d mix into
  d loss / d (control-rate signals [B,5,Tc]) and d loss / d (12 per-voice constants);
* control rate ([B,Tc] work, a few thousand points per voice): ``csrc/voice_ctrl_grad_kernels.hip``
  (``ias_voice_control_backward``) carries the two gradients above back to the 78 parameters.  Its DEFINITION is
  ``control_graph`` below -- the 78 parameters -> (control signals, constants) map restated with differentiable torch
  ops in fp64 -- differentiated by torch's autograd: the kernel is tested against it, it is the fallback for control
  buffers too long for LDS, and ``tests/test_voice_grad_gpu.py`` checks that its forward values agree with the HIP
  control kernels of the render.
"""
import math

import torch

from . import _lib
from . import voice_spec as S

TWO_PI = 2.0 * math.pi
# order of the per-voice constants in ias_voice_backward's partials (csrc/voice_grad_kernels.hip GS_*)
SCALARS = ("f0_1", "depth_1", "phi_1", "f0_2", "depth_2", "phi_2", "kpart", "shape", "gain", "lvl0", "lvl1", "lvl2")


_ADSRS = ("adsr_1", "adsr_2", "lfo_1_amp_adsr", "lfo_2_amp_adsr", "lfo_1_rate_adsr", "lfo_2_rate_adsr")
_LFOS = ("lfo_1", "lfo_2")
_TABLE = {}
_INDEX = {}


def _table(device, dtype):
    """Range table of the 78 parameters as [78] tensors (lo, hi, 1/curve, symmetric)."""
    key = (str(device), dtype)
    if key not in _TABLE:
        cols = list(zip(*[(lo, hi, 1.0 / curve, 1.0 if sym else 0.0) for (_m, _n, lo, hi, curve, sym) in S.PARAMS]))
        _TABLE[key] = tuple(torch.tensor(c, dtype=dtype, device=device) for c in cols)
    return _TABLE[key]


def _from_0to1(params01):
    """torchsynth ModuleParameterRange.from_0to1 for all 78 columns at once: [B,78] in 0..1 -> parameter values.

    non-symmetric: lo + (hi-lo) u^(1/curve);  symmetric: lo + (hi-lo)/2 (sign(d) |d|^(1/curve) + 1), d = 2u-1.
    (x^(1/curve) as exp2(log2(x)/curve); the base is kept off zero so that autograd stays finite there.)"""
    lo, hi, inv_curve, sym = _table(params01.device, params01.dtype)
    tiny = 1e-300 if params01.dtype == torch.float64 else 1e-30
    un = torch.exp2(torch.log2(params01.clamp_min(tiny)) * inv_curve)
    dist = 2.0 * params01 - 1.0
    us = torch.exp2(torch.log2(dist.abs().clamp_min(tiny)) * inv_curve) * torch.sign(dist)
    return torch.where(sym > 0.5, lo + (hi - lo) * 0.5 * (us + 1.0), lo + (hi - lo) * un)


class _Params:
    """p(module, name) -> [B];  p.many(modules, name) -> [B, len(modules)]."""

    def __init__(self, params01):
        self.v = _from_0to1(params01)

    def __call__(self, mod, name):
        return self.v[:, S.INDEX[(mod, name)]]

    def many(self, mods, name):
        key = (str(self.v.device), mods, name)
        idx = _INDEX.get(key)
        if idx is None:   # device index tensors are built once (a host->device copy is not capturable)
            idx = _INDEX[key] = torch.tensor([S.INDEX[(m, name)] for m in mods], dtype=torch.long, device=self.v.device)
        return self.v.index_select(1, idx)


def _ramp(cfg, duration, alpha, start=None, inverse=False):
    """duration, start [B,E] seconds, alpha [B,E,1] -> [B,E,Tc]."""
    dur = (duration * cfg.control_rate).unsqueeze(-1)
    ramp = torch.arange(cfg.control_buffer_size, dtype=duration.dtype, device=duration.device).expand(*duration.shape, -1)
    if start is not None:
        ramp = ramp - (start * cfg.control_rate).unsqueeze(-1)
    ramp = torch.clamp_min(ramp, 0.0)
    # a zero duration makes the ramp +inf -> 1; guarded divisor so that autograd does not produce 0 * inf = nan
    safe = torch.where(dur > 0.0, dur, torch.ones_like(dur))
    ramp = torch.where(dur > 0.0, torch.clamp_max((ramp + S.EPS) / safe + S.EPS, 1.0), torch.ones_like(ramp))
    if inverse:
        ramp = torch.where(dur > 0.0, 1.0 - ramp, ramp)
    # d/dx x^alpha at x = 0 is taken as 0 (autograd would give inf * 0 = nan for alpha < 1)
    return torch.pow(ramp.clamp_min(1e-300), alpha)


def _adsrs(cfg, p, mods, note_on):
    """All envelopes of ``mods`` in one pass -> [B, len(mods), Tc]."""
    attack, decay, sustain = p.many(mods, "attack"), p.many(mods, "decay"), p.many(mods, "sustain")
    release, alpha = p.many(mods, "release"), p.many(mods, "alpha").unsqueeze(-1)
    note_on = note_on.unsqueeze(1).expand_as(attack)
    new_attack = torch.minimum(attack, note_on)
    new_decay = torch.minimum(torch.clamp_min(note_on - attack, 0.0), decay)
    a = _ramp(cfg, new_attack, alpha)
    sus = sustain.unsqueeze(-1)
    d = (1.0 - sus) * _ramp(cfg, new_decay, alpha, start=new_attack, inverse=True) + sus
    r = _ramp(cfg, release, alpha, start=note_on, inverse=True)
    return a * d * r


def _lfos(cfg, p, mods, rate_env):
    """Both LFOs in one pass: rate_env [B,2,Tc] -> [B,2,Tc]."""
    freq = p.many(mods, "frequency").unsqueeze(-1)
    freq = torch.clamp_min(freq + p.many(mods, "mod_depth").unsqueeze(-1) * rate_env, 0.0)
    arg = torch.cumsum(TWO_PI * freq / cfg.control_rate, dim=-1) + p.many(mods, "initial_phase").unsqueeze(-1)
    cos = torch.cos(arg + math.pi)
    square = (torch.sign(cos) + 1.0) / 2.0
    cos = (cos + 1.0) / 2.0
    saw = torch.remainder(arg, TWO_PI) / TWO_PI
    revsaw = 1.0 - saw
    tri = 2.0 * saw
    tri = torch.where(tri > 1.0, 2.0 - tri, tri)
    shapes = torch.stack([cos, tri, saw, revsaw, square], dim=2)              # [B,2,5,Tc]
    mode = torch.stack([p.many(mods, s) for s in S.LFO_SHAPES], dim=2)        # [B,2,5]
    mode = torch.pow(mode, S.LFO_EXPONENT)
    mode = mode / torch.sum(mode, dim=2, keepdim=True)
    return (mode.unsqueeze(-1) * shapes).sum(dim=2)


def _midi_to_hz(midi):
    return 440.0 * torch.exp2((midi - 69.0) / 12.0)


def control_graph(params01, cfg):
    """Differentiable restatement of the control-rate pass (csrc/voice_kernels.hip voice_env / voice_lfo /
    voice_modmix kernels): params01 [B,78] -> (ctrl [B,5,Tc], constants [B,12] in ``SCALARS`` order).
    The six envelopes, the two LFOs and the 78 range maps are each evaluated as one batched expression (a few
    dozen device kernels instead of a few thousand)."""
    p = _Params(params01)
    env = _adsrs(cfg, p, _ADSRS, p("keyboard", "duration"))                  # [B,6,Tc]
    lfo = _lfos(cfg, p, _LFOS, env[:, 4:6]) * env[:, 2:4]
    i0 = S.INDEX[("mod_matrix", f"{S.MOD_INPUTS[0]}->{S.MOD_OUTPUTS[0]}")]     # the 20 weights are consecutive columns
    w = p.v[:, i0:i0 + len(S.MOD_INPUTS) * len(S.MOD_OUTPUTS)]
    w = w.reshape(-1, len(S.MOD_INPUTS), len(S.MOD_OUTPUTS)).swapaxes(1, 2)
    ctrl = torch.matmul(w, torch.cat([env[:, 0:2], lfo], dim=1))

    midi_f0 = p("keyboard", "midi_f0")
    depth_2 = p("vco_2", "mod_depth")
    max_f0 = _midi_to_hz(midi_f0 + torch.clamp_min(depth_2, 0.0))
    kpart = math.pi * 12000.0 / (max_f0 * torch.log10(max_f0))
    shape = p("vco_2", "shape")
    scal = torch.stack([
        midi_f0 + p("vco_1", "tuning"), p("vco_1", "mod_depth"), p("vco_1", "initial_phase"),
        midi_f0 + p("vco_2", "tuning"), depth_2, p("vco_2", "initial_phase"),
        kpart, shape, 1.0 - shape / 2.0,
        p("mixer", "vco_1"), p("mixer", "vco_2"), p("mixer", "noise")], dim=1)
    return ctrl, scal


def normalisation_rows(g_audio, audio, peaks):
    """rownorm [B,4] for ``audio_rate_backward``: the adjoint of torchsynth's normalize_if_clipping (audio = mix / peak
    on rows with peak = max |mix| > 1) folded into two small launches: the divisor per row and the correction
    -sign(audio[t*]) * sum_t g[t] audio[t] / peak at the peak sample t* (csrc/voice_grad_kernels.hip, K0)."""
    lib = _lib.load()
    B, T = audio.shape
    _lib.require_f32(g_audio, audio, peaks)
    scratch = torch.empty((B, lib.ias_voice_norm_scratch_len(T)), dtype=torch.float64, device=audio.device)
    rownorm = torch.empty((B, 4), dtype=torch.float32, device=audio.device)
    _lib.check(lib.ias_voice_norm_backward(_lib.ptr(g_audio), _lib.ptr(audio), _lib.ptr(peaks), _lib.ptr(scratch),
                                           _lib.ptr(rownorm), B, T, _lib.stream()), "ias_voice_norm_backward")
    return rownorm


class BackwardPrelude:
    """The parts of the two backward calls that do not see a cotangent -- the phase increments + tile sums of the
    audio-rate adjoint (``ias_voice_backward_sums_stage(0)``) and the envelope values of the control-rate adjoint
    (``ias_voice_control_backward_ws_stage(0)``) -- launched at RENDER time on a stream of their own, beside whatever the
    caller computes between the render and its backward (the losses).  ``join()`` makes the current stream wait for them.
    ``IAS_VOICE_PRELUDE=0``: everything in the backward, as before."""

    _streams = {}

    def __init__(self, voice, p, ctrl, vconst):
        c = voice.synthconfig
        lib = _lib.load()
        B, T, Tc = c.batch_size, c.buffer_size, c.control_buffer_size
        dev = p.device
        self.planes = torch.empty((B, lib.ias_voice_grad_nplanes(), T), dtype=torch.float32, device=dev)
        self.tile_sums = torch.empty((B, lib.ias_voice_grad_tiles(T), 2), dtype=torch.float64, device=dev)
        nws = int(lib.ias_voice_control_backward_ws_bytes(B, Tc))
        self.cws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
        cur = torch.cuda.current_stream(dev)
        side = self._streams.get(dev)
        if side is None:
            side = self._streams[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        self.ctrl_ok = True
        self._joined = False
        # Everything the side stream touches was allocated on (or belongs to) the CURRENT stream: tell the caching allocator
        # (as spectral.py does for its side streams), or a render whose autograd node is dropped without a backward -- a
        # grad-enabled render used only for metrics, an exception, `del loss` -- would hand these blocks to current-stream
        # work while the side-stream kernels are still writing / reading them.
        for t in (self.planes, self.tile_sums, self.cws, p, ctrl, vconst):
            t.record_stream(side)
        with torch.cuda.stream(side):
            st = lib.ias_voice_backward_sums_stage(0, _lib.ptr(ctrl), _lib.ptr(vconst), None, None, None, _lib.ptr(self.planes),
                                                   _lib.ptr(self.tile_sums), None, None, None, B, T, Tc, c.sample_rate,
                                                   _lib.stream())
            _lib.check(st, "ias_voice_backward_sums_stage")
            st = lib.ias_voice_control_backward_ws_stage(0, _lib.ptr(p), None, None, None, _lib.ptr(self.cws), self.cws.numel(),
                                                         B, Tc, c.control_rate, _lib.stream())
            if st == -2:                                  # control buffer too long for the HIP form: the torch graph does it all
                self.ctrl_ok = False
            else:
                _lib.check(st, "ias_voice_control_backward_ws_stage")
        self.side = side

    def join(self):
        torch.cuda.current_stream(self.planes.device).wait_stream(self.side)
        self._joined = True

    def __del__(self):
        # a prelude that never saw its backward still has to be joined: inside a hipGraph capture an unjoined fork fails
        # the capture, outside one it would let later current-stream work overtake the side stream's kernels
        try:
            if not getattr(self, "_joined", True):
                self.join()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


def prelude_enabled():
    import os
    return os.environ.get("IAS_VOICE_PRELUDE", "1") not in ("0", "")


def audio_rate_backward(voice, params01, g_mixed, rownorm=None, control=None, prelude=None):
    """HIP adjoint of the audio-rate render: g_mixed [B,T] -> (g_ctrl [B,5,Tc] fp32, g_constants [B,12] fp64).
    ``rownorm``: ``normalisation_rows`` of a normalised render; ``g_mixed`` is then the cotangent of the normalised
    audio.  ``prelude``: a joined ``BackwardPrelude`` of the same render (its stage 0 has run)."""
    c = voice.synthconfig
    lib = _lib.load()
    B, T, Tc = c.batch_size, c.buffer_size, c.control_buffer_size
    g_mixed = g_mixed.to(torch.float32).contiguous()
    _lib.require_f32(g_mixed, voice.noise)
    # the forward's control signals when the caller kept them (voice.rendered_control()), else one more control pass
    ctrl, vconst = control if control is not None else voice.control_signals(params01)
    dev = g_mixed.device
    ntiles = lib.ias_voice_grad_tiles(T)
    if prelude is not None:
        planes, tile_sums = prelude.planes, prelude.tile_sums
    else:
        planes = torch.empty((B, lib.ias_voice_grad_nplanes(), T), dtype=torch.float32, device=dev)
        tile_sums = torch.empty((B, ntiles, 2), dtype=torch.float64, device=dev)
    partials = torch.empty((B, ntiles, lib.ias_voice_grad_nscalars()), dtype=torch.float64, device=dev)
    g_ctrl = torch.empty((B, 5, Tc), dtype=torch.float32, device=dev)
    g_scal = torch.empty((B, lib.ias_voice_grad_nscalars()), dtype=torch.float64, device=dev)
    args = (_lib.ptr(ctrl), _lib.ptr(vconst), _lib.ptr(voice.noise), _lib.ptr(g_mixed), _lib.ptr(rownorm), _lib.ptr(planes),
            _lib.ptr(tile_sums), _lib.ptr(partials), _lib.ptr(g_ctrl), _lib.ptr(g_scal), B, T, Tc, c.sample_rate, _lib.stream())
    if prelude is not None:
        _lib.check(lib.ias_voice_backward_sums_stage(1, *args), "ias_voice_backward_sums_stage")
    else:
        _lib.check(lib.ias_voice_backward_sums(*args), "ias_voice_backward_sums")
    return g_ctrl, g_scal


def _control_backward_eager(cfg, p, g_ctrl, g_scal):
    with torch.enable_grad():
        pd = p.double().requires_grad_(True)
        ctrl_t, scal_t = control_graph(pd, cfg)
        (g_p,) = torch.autograd.grad([ctrl_t, scal_t], pd, [g_ctrl.double(), g_scal])
    return g_p


class _ControlBackwardGraph:
    """The control graph's forward + backward is a few hundred tiny device kernels with static shapes: captured
    once per (batch, control length, device) into a hipGraph and replayed on static buffers (host launch time
    4.5 ms -> one graph launch)."""

    def __init__(self, cfg, B, device):
        Tc = cfg.control_buffer_size
        self.p = torch.full((B, S.NPARAMS), 0.5, dtype=torch.float32, device=device)
        self.g_ctrl = torch.zeros((B, 5, Tc), dtype=torch.float32, device=device)
        self.g_scal = torch.zeros((B, len(SCALARS)), dtype=torch.float64, device=device)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):                      # warm-up outside the capture (allocator, lazy inits)
                _control_backward_eager(cfg, self.p, self.g_ctrl, self.g_scal)
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = _control_backward_eager(cfg, self.p, self.g_ctrl, self.g_scal)

    def __call__(self, p, g_ctrl, g_scal):
        self.p.copy_(p)
        self.g_ctrl.copy_(g_ctrl)
        self.g_scal.copy_(g_scal)
        self.graph.replay()
        return self.out.clone()


_GRAPHS = {}


def _control_backward_hip(cfg, p, g_ctrl, g_scal, prelude=None):
    """The same adjoint as one HIP launch (csrc/voice_ctrl_grad_kernels.hip); None if the shape is unsupported."""
    lib = _lib.load()
    g_ctrl = g_ctrl.to(torch.float32).contiguous()
    g_scal = g_scal.to(torch.float64).contiguous()
    out = torch.empty((p.shape[0], S.NPARAMS), dtype=torch.float32, device=p.device)
    if prelude is not None and prelude.ctrl_ok:
        ws = prelude.cws                                  # stage 0 (the envelope values) ran beside the losses
        st = lib.ias_voice_control_backward_ws_stage(1, _lib.ptr(p), _lib.ptr(g_ctrl), _lib.ptr(g_scal), _lib.ptr(out),
                                                     _lib.ptr(ws), ws.numel(), p.shape[0], cfg.control_buffer_size,
                                                     cfg.control_rate, _lib.stream())
        _lib.check(st, "ias_voice_control_backward_ws_stage")
        return out
    nws = lib.ias_voice_control_backward_ws_bytes(p.shape[0], cfg.control_buffer_size)
    ws = torch.empty(max(int(nws), 16), dtype=torch.uint8, device=p.device)
    st = lib.ias_voice_control_backward_ws(_lib.ptr(p), _lib.ptr(g_ctrl), _lib.ptr(g_scal), _lib.ptr(out), _lib.ptr(ws),
                                           ws.numel(), p.shape[0], cfg.control_buffer_size, cfg.control_rate, _lib.stream())
    if st == -2:      # IAS_ERR_UNSUPPORTED: control buffer too long for LDS -> torch graph
        return None
    _lib.check(st, "ias_voice_control_backward_ws")
    return out


def _control_backward(cfg, p, g_ctrl, g_scal, use_hip=True, prelude=None):
    """d loss / d params01 [B,78] from the gradients of the control signals and per-voice constants."""
    if use_hip:
        out = _control_backward_hip(cfg, p, g_ctrl, g_scal, prelude)
        if out is not None:
            return out
    if torch.cuda.is_current_stream_capturing():
        return _control_backward_eager(cfg, p, g_ctrl, g_scal)
    key = (p.shape[0], cfg.control_buffer_size, cfg.control_rate, str(p.device))
    runner = _GRAPHS.get(key)
    if runner is None:
        try:
            runner = _ControlBackwardGraph(cfg, p.shape[0], p.device)
        except RuntimeError:                          # capture not possible here: plain launches
            runner = False
        _GRAPHS[key] = runner
    if runner is False:
        return _control_backward_eager(cfg, p, g_ctrl, g_scal)
    return runner(p, g_ctrl, g_scal)


class _RenderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params01, voice, normalize):
        p = params01.detach().to(torch.float32).contiguous()
        audio = voice._render_nograd(p, normalize)
        ctrl, vconst, peaks = voice.saved_for_backward(with_peaks=normalize)
        ctx.voice, ctx.normalize = voice, normalize
        ctx.save_for_backward(p, audio, peaks, ctrl, vconst)
        # a backward will follow (this is only reached when params01 requires grad): its cotangent-free part starts now
        ctx.prelude = BackwardPrelude(voice, p, ctrl, vconst) if ctx.needs_input_grad[0] and prelude_enabled() and p.is_cuda \
            else None
        return audio

    @staticmethod
    def backward(ctx, g_audio):
        p, audio, peaks, ctrl, vconst = ctx.saved_tensors
        voice = ctx.voice
        g = g_audio.to(torch.float32).contiguous()
        # audio = mixed / peak on rows with peak > 1 (peak = max |mixed|, attained at t*):
        #   g_mixed = g / peak, and the peak itself takes -sign(mixed[t*]) * sum_t g[t] audio[t] / peak at t*
        rownorm = normalisation_rows(g, audio, peaks) if ctx.normalize else None
        prelude, ctx.prelude = ctx.prelude, None          # (a second backward through a retained graph runs the whole thing)
        if prelude is not None:
            prelude.join()
        g_ctrl, g_scal = audio_rate_backward(voice, p, g, rownorm, (ctrl, vconst), prelude)
        g_p = _control_backward(voice.synthconfig, p, g_ctrl, g_scal, prelude=prelude)
        return g_p.to(torch.float32), None, None


def render_with_grad(voice, params01, normalize=True):
    """audio [B,T] = voice render of params01 [B,78], differentiable with respect to params01."""
    return _RenderFn.apply(params01, voice, normalize)
