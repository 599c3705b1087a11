"""ctypes binding of the C-ABI HIP library (``csrc/libias_hip.so``, declared in ``include/ias_hip.h``).

The product path has no CPU fallback: if the library is missing or a call returns a
non-zero status, an exception is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# IAS_HIP_LIB: developer override used by scripts/diag to time alternative builds of the same C ABI
LIB_PATH = os.environ.get("IAS_HIP_LIB") or os.path.join(_HERE, "csrc", "libias_hip.so")

_ERRORS = {
    -1: "IAS_ERR_ARG (bad pointer or dimension)",
    -2: "IAS_ERR_UNSUPPORTED (shape outside the kernel's limits)",
    -3: "IAS_ERR_LAUNCH (HIP launch failed)",
    -4: "IAS_ERR_WORKSPACE (workspace too small)",
}

_c = ctypes
_P, _I, _LL = _c.c_void_p, _c.c_int, _c.c_longlong
_F = _c.c_float

# name -> (restype, argtypes).  Must list every symbol include/ias_hip.h declares.
SYMBOLS = {
    "ias_version": (_I, []),
    "ias_stream_copy": (_I, [_P, _P, _LL, _P]),
    "ias_stamp": (_I, [_P, _P]),
    "ias_voice_workspace_bytes": (_LL, [_I, _I, _I]),
    "ias_voice_control": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ias_voice_control_ws": (_I, [_P, _P, _LL, _I, _I, _I, _I, _P]),
    "ias_voice_control_debug": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ias_voice_render": (_I, [_P, _P, _P, _P, _LL, _I, _I, _I, _I, _I, _I, _I, _P]),
    "ias_voice_stage": (_I, [_I, _I, _P, _P, _P, _LL, _I, _I, _I, _I, _P]),
    "ias_voice_read_status": (_I, [_P, _P, _I, _I, _I, _P]),
    "ias_voice_read_status_sticky": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ias_voice_peaks_offset": (_LL, [_I, _I, _I]),
    "ias_voice_ctrl_offset": (_LL, [_I, _I, _I]),
    "ias_voice_vconst_offset": (_LL, [_I, _I, _I]),
    "ias_voice_read_peaks": (_I, [_P, _P, _I, _I, _I, _P]),
    "ias_voice_save_for_backward": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ias_voice_grad_tiles": (_I, [_I]),
    "ias_voice_grad_nscalars": (_I, []),
    "ias_voice_grad_nplanes": (_I, []),
    "ias_voice_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_voice_backward_norm": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_voice_backward_sums": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_voice_backward_sums_stage": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_voice_norm_scratch_len": (_LL, [_I]),
    "ias_voice_norm_backward": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "ias_voice_control_backward": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ias_voice_control_backward_ws_bytes": (_LL, [_I, _I]),
    "ias_voice_control_backward_ws": (_I, [_P, _P, _P, _P, _P, _LL, _I, _I, _I, _P]),
    "ias_voice_control_backward_ws_stage": (_I, [_I, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _P]),
    "ias_pqmf_out_len": (_I, [_I, _I, _I]),
    "ias_pqmf_packed_taps_len": (_I, [_I, _I]),
    "ias_pqmf_pack_taps": (_I, [_P, _P, _I, _I, _P]),
    "ias_pqmf_modtab_len": (_I, []),
    "ias_pqmf_build_modtab": (_I, [_P, _I, _I, _P]),
    "ias_pqmf_analysis": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_pqmf_synth_taps_len": (_I, [_I, _I]),
    "ias_pqmf_pack_synth_taps": (_I, [_P, _P, _I, _I, _P]),
    "ias_pqmf_synthesis": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_pqmf_synthesis_t": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ias_stft_num_frames": (_I, [_I, _I, _I]),
    "ias_stft_partials_count": (_LL, [_I, _I, _I, _I, _I]),
    "ias_stft_tables_len": (_I, [_I]),
    "ias_stft_build_tables": (_I, [_I, _P, _P]),
    "ias_stft_segtab_len": (_LL, [_I, _P, _P, _P, _P, _I]),
    "ias_stft_build_segtab": (_I, [_I, _P, _P, _P, _P, _I, _P]),
    "ias_stft_mtables_len": (_LL, [_I, _P, _P, _I]),
    "ias_stft_build_mtables": (_I, [_I, _P, _P, _P, _P, _P, _I, _P]),
    "ias_stft": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "ias_stft_loss_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I,
                                   ctypes.c_float, ctypes.c_float, _P]),
    "ias_stft_grad_frames": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, ctypes.c_float,
                                 ctypes.c_float, _P]),
    "ias_stft_grad_spans": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, ctypes.c_float,
                                ctypes.c_float, _P, _P]),
    "ias_stft_grad_span_plan": (_I, [_I, _I, _I, _I, _I, _I, _P]),
    "ias_stft_grad_combine": (_I, [_P, _P, _I, _P, _P, _I, _I, _P]),
    "ias_reduce_partials": (_I, [_P, _LL, _P, _c.c_double, _P, _P]),
    "ias_mrstft_total": (_I, [_P, _P, _I, _P, _P]),
    "ias_mrstft_coef": (_I, [_P, _P, _c.c_double, _I, _P, _P]),
    "ias_l1_partials_count": (_LL, [_LL]),
    "ias_l1_partials": (_I, [_P, _P, _LL, _P, _P]),
    "ias_l1_grad": (_I, [_P, _P, _P, _F, _LL, _P, _P]),
    "ias_vicreg_workspace_bytes": (_LL, [_I, _I]),
    "ias_vicreg_colstats_offset": (_LL, [_I, _I]),
    "ias_vicreg_loss": (_I, [_P, _P, _P, _P, _LL, _I, _I, _I, _F, _F, _F, _P]),
    "ias_vicreg_backward": (_I, [_P, _P, _P, _P, _P, _P, _LL, _I, _I, _I, _F, _F, _F, _P]),
    "ias_vicreg_loss_ld": (_I, [_P, _P, _LL, _P, _P, _LL, _I, _I, _I, _F, _F, _F, _P]),
    "ias_vicreg_backward_ld": (_I, [_P, _P, _LL, _P, _P, _P, _LL, _P, _LL, _I, _I, _I, _F, _F, _F, _P]),
    "ias_vicreg_backward4_ld": (_I, [_P, _P, _LL, _P, _P, _P, _P, _P, _P, _LL, _P, _LL, _I, _I, _I, _F, _F, _F, _P]),
    "ias_vicreg_stage": (_I, [_I, _P, _P, _P, _P, _LL, _I, _I, _I, _F, _F, _F, _P]),
    "ias_conv_out_size": (_I, [_I, _I, _I]),
    "ias_dwconv_forward": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ias_dwconv_backward_data": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ias_dwconv_weight_scratch": (_LL, [_I, _I, _I]),
    "ias_dwconv_weight_scratch_hw": (_LL, [_I, _I, _I, _I, _I, _I]),
    "ias_dwconv_backward_weight": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ias_dwconv_backward_weight_partials": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ias_pwconv_supported": (_I, [_I, _I]),
    "ias_pwconv_forward": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_pwconv_backward_data": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_pwconv_weight_scratch": (_LL, [_I, _I, _I, _I]),
    "ias_pwconv_backward_weight": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_pwconv_backward_weight_partials": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_pwconv_forward_scaled": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_pwconv_backward_weight_scaled": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_pwconv_backward_weight_partials_scaled": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_reduce_partials_multi": (_I, [_P, _I, _P]),
    "ias_se_plane_reduce": (_I, [_P, _P, _P, _LL, _I, _F, _P]),
    "ias_se_scale": (_I, [_P, _P, _P, _P, _LL, _I, _F, _P]),
    "ias_se_mlp_forward": (_I, [_P] * 8 + [_I, _I, _I, _P]),
    "ias_se_mlp_backward": (_I, [_P] * 13 + [_I, _I, _I, _P]),
    "ias_conv2x2_patches": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ias_conv2x2_patches_backward": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ias_conv2x2_patches_backward_nchw": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ias_conv2x2_patches_nchw": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ias_colsum_scratch_floats": (_LL, [_I, _I]),
    "ias_colsum": (_I, [_P, _P, _P, _I, _I, _P]),
    "ias_colsum_partials": (_I, [_P, _P, _I, _I, _P]),
    "ias_stem_forward": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ias_stem_weight_scratch": (_LL, [_I]),
    "ias_stem_backward_weight": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ias_stem_backward_weight_partials": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "ias_bn_scratch_doubles": (_LL, [_I, _I]),
    "ias_bn_act_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P]),
    "ias_bn_act_forward_res": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P]),
    "ias_bn_act_forward_pool": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P]),
    "ias_bn_act_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_bn1d_groups_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P]),
    "ias_bn1d_groups_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ias_lars_chunk_elems": (_I, []),
    "ias_lars_step": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ias_lars_step_carry": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
}

# Symbols only the diagnostic library exports (include/ias_hip_diag.h).
DIAG_SYMBOLS = {
    "ias_vicreg_set_form": (_I, [_I]),
}
DIAG_LIB_PATH = os.path.join(_HERE, "csrc", "libias_hip_diag.so")

_lib = None
_diag = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load libias_hip.so (once).  Raises HipLibraryMissing if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C inverse-audio-synthesis_amd/csrc`.  There is no CPU fallback."
        )
    _lib = _bind(ctypes.CDLL(LIB_PATH), SYMBOLS)
    return _lib


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def load_diag():
    """The diagnostic build of the same sources (csrc/libias_hip_diag.so, -DIAS_DIAG: environment switches live,
    superseded kernels and ``ias_vicreg_set_form`` compiled in; include/ias_hip_diag.h).  A SEPARATE library instance:
    nothing the package does goes through it unless a test or a diagnostic script asks for it (``use_library``)."""
    global _diag
    if _diag is None:
        if not os.path.exists(DIAG_LIB_PATH):
            raise HipLibraryMissing(f"{DIAG_LIB_PATH} not found: `make -C inverse-audio-synthesis_amd/csrc`")
        _diag = _bind(_bind(ctypes.CDLL(DIAG_LIB_PATH), SYMBOLS), DIAG_SYMBOLS)
    return _diag


class use_library:
    """``with use_library(load_diag()): ...`` -- route every call of the package through another build of the C ABI for the
    duration of the block (tests and diagnostics that compare kernel variants; never used by the package itself)."""

    def __init__(self, lib):
        self.lib = lib

    def __enter__(self):
        global _lib
        load()
        self.saved, _lib = _lib, self.lib
        return self.lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed: {_ERRORS.get(status, status)}")


def ptr(t):
    """Device pointer of a contiguous CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("inverse-audio-synthesis_amd kernels need tensors on a ROCm device (no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"expected float32, got {t.dtype}")
