"""ctypes binding of liblyricalign_hip.so (see include/lyricalign.h).

The HIP library is the product: if it is missing or cannot be loaded this module
raises -- there is deliberately no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int32, c_int64, c_size_t, c_void_p

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# LA_LIB_PATH: another build of the same library (same-box A/B of two builds; tools/ab_bench.sh)
LIB_PATH = os.environ.get("LA_LIB_PATH") or os.path.join(HERE, "liblyricalign_hip.so")

LA_OK, LA_EINVAL, LA_EINFEASIBLE, LA_EEMPTY, LA_EHIP, LA_ETIMEOUT, LA_EUNSUPPORTED = range(7)
LA_F32, LA_BF16, LA_F16 = 0, 1, 2
LA_Q_LOG2 = 0x100        # modifier on the attention entry points' dtype / la_encoder_weights.dtype (include/lyricalign.h)
LA_VARIANT_PLAIN, LA_VARIANT_CTC = 0, 1
EPI_BIAS, EPI_GELU, EPI_RESIDUAL, EPI_OUT_F32, EPI_MISH = 1, 2, 4, 8, 16
EPI_GELU_ERF = 4096
EPI_RES_GELU_GRAD = 32        # la_gemm_f16x2: result * gelu'(residual) instead of + residual
GEMM_TRANS_A, GEMM_TRANS_W = 512, 1024        # la_gemm_ex operand layout flags (float32)

# every symbol include/lyricalign.h declares: (name, restype, argtypes)
_I32, _I64, _P, _SZ = c_int32, c_int64, c_void_p, c_size_t
SYMBOLS = {
    "la_version": (c_int32, []),
    "la_last_error": (c_char_p, []),
    "la_device_arch_ok": (c_int32, []),
    "la_timer_enable": (c_int32, [c_char_p]),
    "la_timer_disable": (c_int32, []),
    "la_timer_read": (c_int32, [POINTER(c_double), POINTER(c_int64)]),
    "la_timer_reset": (c_int32, []),
    "la_timer_sample": (c_int32, [c_int32]),
    "la_timer_read_work": (c_int32, [POINTER(c_double), POINTER(c_int64), POINTER(c_double), POINTER(c_int64)]),
    "la_split_f16x2": (c_int32, [_P, _I64, _I32, _I32, _P, _I64, _P, _P]),
    "la_split_f16x2_t": (c_int32, [_P, _I64, _I32, _I32, _P, _I64, _P, _P]),
    "la_split_f16x2_act": (c_int32, [_P, _I64, _I32, _I32, _P, _I64, _P, _I32, _P]),
    "la_split_f16x2_t_act": (c_int32, [_P, _I64, _I32, _I32, _P, _I64, _P, _I32, _P]),
    "la_split_f16x2_t_colsum": (c_int32, [_P, _I64, _I32, _I32, _P, _I64, _P, _I32, _P, _P]),
    "la_split_f16x2_max": (c_int32, [_P, _I64, _I32, _I32, _P, _I64, _P, _I32, _P, _P]),
    "la_split_f16x2_t_tmax": (c_int32, [_P, _I64, _I32, _I32, _P, _I64, _P, _I32, _P, _P, _P]),
    "la_gemm_f16x2": (c_int32, [_I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I64, _P, _P, _I64, _I32, _P]),
    "la_layernorm_f16x2": (c_int32, [_P, _I64, _I32, _I32, _P, _P, _P, _I64, _P, _P]),
    "la_fc_emissions_x2_workspace_bytes": (c_int32, [_I32, _I32, _I32, _I32, _I32, POINTER(_SZ)]),
    "la_fc_emissions_x2": (c_int32, [_P, _I64, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P, _I32, _P, _I64, _I64, _P, _SZ, _P]),
    "la_set_option": (c_int32, [c_char_p, c_int64]),
    "la_get_option": (c_int32, [c_char_p, POINTER(c_int64)]),
    "la_has_experiments": (c_int32, []),
    "la_viterbi_workspace_bytes": (c_int32, [_I32, _I32, _I32, POINTER(_SZ)]),
    "la_viterbi_batch": (c_int32, [_P, _I64, _I64, _P, _I32, _P, _P, _I32, _I32, _I32, _P, _P, _I32, _P, _P, _P, _SZ, _P]),
    "la_viterbi_core": (c_int32, [_P, _I64, _P, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "la_emissions_from_logits": (c_int32, [_P, _I64, _I64, _I32, _I32, _I32, _I32, _P, _I32, _P, _I32, _P, _I64, _I64, _P]),
    "la_logmel_workspace_bytes": (c_int32, [_I32, _I32, POINTER(_SZ)]),
    "la_logmel_f32": (c_int32, [_P, _I32, _I32, _P, _P, _P, _I64, _I64, _P, _SZ, _P]),
    "la_logmel_constants_bytes": (c_int32, [POINTER(_SZ)]),
    "la_logmel_constants": (c_int32, [_P, _P, _P, _SZ, _P]),
    "la_logmel_f32_prepared": (c_int32, [_P, _I32, _I32, _P, _P, _I64, _I64, _P, _SZ, _P]),
    "la_gemm": (c_int32, [_I32, _I32, _I32, _I32, _I32, _P, _I64, _I64, _P, _P, _I64, _I64, _P, _P, _I64, _I64, _I32, _P]),
    "la_layernorm": (c_int32, [_P, _I64, _I32, _I32, _P, _P, _P, _I64, _I32, _P]),
    "la_attention": (c_int32, [_I32, _P, _I64, _P, _I64, _I32, _I32, _I32, _P]),
    "la_attention_ex": (c_int32, [_I32, _P, _I64, _P, _P, _I64, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P]),
    "la_embed_tokens": (c_int32, [_P, _I32, _I32, _P, _I32, _P, _I32, _P, _P]),
    "la_mel_to_rows": (c_int32, [_P, _I64, _I64, _I32, _I32, _I32, _P, _I32, _I32, _P]),
    "la_gru_workspace_bytes": (c_int32, [_I32, _I32, _I32, POINTER(_SZ)]),
    "la_gru_layer": (c_int32, [_I32, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _SZ, _P, _P]),
    "la_fc_emissions_workspace_bytes": (c_int32, [_I32, _I32, _I32, _I32, _I32, _I32, POINTER(_SZ)]),
    "la_fc_emissions": (c_int32, [_I32, _P, _I64, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P, _I32, _P, _I64, _I64, _P, _SZ, _P]),
    "la_multitask_loss_workspace_bytes": (c_int32, [_I32, _I32, _I32, POINTER(_SZ)]),
    "la_multitask_loss": (c_int32, [_P, _I64, _I64, _I32, _I32, _I32, _P, _P, _I32, _P, _I32, _I32, _I32, ctypes.c_float, _P, _P, _I64, _I64, _P, _SZ, _P]),
    "la_grad_sqnorm_f32": (c_int32, [_P, _I64, _P, _P]),
    "la_adamw_step_f32": (c_int32, [_P, _P, _P, _P, _I64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _I32, _P, ctypes.c_float, ctypes.c_float, _P]),
    "la_gru_layer_train_fwd": (c_int32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _SZ, _P, _P]),
    "la_gru_layer_bwd": (c_int32, [_P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _SZ, _P, _P]),
    "la_transpose_pad_f32": (c_int32, [_P, _I64, _I32, _I32, _P, _I64, _I32, _I32, _P]),
    "la_colsum_f32": (c_int32, [_P, _I64, _I32, _I32, _P, _P]),
    "la_mish_f32": (c_int32, [_P, _P, _I64, _P]),
    "la_mish_bwd_f32": (c_int32, [_P, _P, _P, _I64, _P]),
    "la_mask_scale_f32": (c_int32, [_P, _P, ctypes.c_float, _P, _I64, _P]),
    "la_attention_cached": (c_int32, [_I32, _P, _I64, _I64, _P, _P, _I64, _I64, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P]),
    "la_topk_rows_f32": (c_int32, [_P, _I64, _I32, _I32, _I32, _P, _P, _P, _P]),
    "la_argmax_rows_f32": (c_int32, [_P, _I64, _I32, _I32, _P, _P]),
    "la_gemm_fused_ln": (c_int32, [_I32, _I32, _I32, _I32, _I32, _P, _I64, _I64, _P, _P, _I64, _I64, _P, _P, _I64, _I64, _I32, _P, _I64, _I64,
                                   _P, _P, _P, _P]),
    "la_ln_stats_finalize": (c_int32, [_P, _I32, _I32, ctypes.c_float, _P, _P]),
    "la_gemm_split": (c_int32, [_I32, _I32, _I32, _I32, _I32, _P, _I64, _I64, _P, _P, _P, _I64, _I64, _P, _P, _I64, _I64, _I32, _P, _P]),
    "la_layernorm_split": (c_int32, [_I32, _P, _P, _I64, _I32, _I32, _P, _P, _P, _I64, _I32, _P]),
    "la_row_stats16": (c_int32, [_I32, _P, _I64, _I32, _I32, ctypes.c_float, _P, _P]),
    "la_gemm_ex": (c_int32, [_I32, _I32, _I32, _I32, _I32, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I64, _P, _I32, _P]),
    "la_transpose_pad_batched_f32": (c_int32, [_P, _I64, _I64, _I32, _I32, _P, _I64, _I64, _I32, _I32, _I32, _P]),
    "la_gelu_f32": (c_int32, [_P, _P, _I64, _P]),
    "la_gelu_bwd_f32": (c_int32, [_P, _P, _P, _I64, _P]),
    "la_add_f32": (c_int32, [_P, _P, _P, _I64, _P]),
    "la_scale_f32": (c_int32, [_P, ctypes.c_float, _P, _I64, _P]),
    "la_layernorm_bwd_f32": (c_int32, [_P, _P, _P, _I32, _I32, _P, _P, _P]),
    "la_layernorm_bwd_sums_f32": (c_int32, [_P, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P]),
    "la_softmax_rows_f32": (c_int32, [_P, _I64, _I64, _I32, _I32, _P]),
    "la_attention_bwd_workspace_bytes": (c_int32, [_I32, _I32, _I32, POINTER(_SZ)]),
    "la_attention_bwd_stats_f32": (c_int32, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "la_attention_bwd_f32": (c_int32, [_P, _I64, _P, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P,
                                       _SZ, _P]),
    "la_attention_lse_f32": (c_int32, [_P, _I64, _P, _P, _I64, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "la_attention_f16x2_workspace_bytes": (c_int32, [_I32, _I32, _I32, _I32, POINTER(_SZ)]),
    "la_attention_lse_f16x2": (c_int32, [_P, _I64, _P, _P, _I64, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "la_attention_bwd_f16x2_workspace_bytes": (c_int32, [_I32, _I32, _I32, _I32, POINTER(_SZ)]),
    "la_attention_bwd_f16x2": (c_int32, [_P, _I64, _P, _P, _I64, _P, _I64, _P, _I64, _P, _I64, _P, _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P,
                                         _SZ, _P]),
    "la_embed_tokens_bwd_f32": (c_int32, [_P, _P, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "la_cross_entropy_f32": (c_int32, [_P, _I64, _I32, _I32, _P, ctypes.c_float, _P, _P, _P, _I64, _P]),
    "la_softmax_bwd_rows_f32": (c_int32, [_P, _P, _I64, _I64, _I32, _P]),
    "la_col2im3_f32": (c_int32, [_P, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "la_resample_poly_f32": (c_int32, [_P, _I64, _P, _I64, _I32, _I32, _I64, _P, _I64, _P]),
    "la_cast_f32_to_bf16": (c_int32, [_P, _P, _I64, _P]),
    "la_encoder_workspace_bytes": (c_int32, [_P, _I32, POINTER(_SZ)]),
    "la_encoder_forward": (c_int32, [_P, _P, _I64, _I64, _I32, _P, _I64, _I32, _P, _SZ, _P]),
    "la_align_head_workspace_bytes": (c_int32, [_P, _I32, _I32, _I32, POINTER(_SZ)]),
    "la_align_head_forward": (c_int32, [_P, _P, _I64, _I64, _I32, _I32, _I32, _P, _I32, _P, _I32, _P, _P, _I32, _P, _P, _P, _P, _SZ, _P, _P]),
    "la_cast_bf16_to_f32": (c_int32, [_P, _P, _I64, _P]),
}

# symbols of the EXPERIMENT build only (csrc/lab/; bound when the loaded library has them: tools/build_variant.sh lab -DLA_EXPERIMENTS)
LAB_SYMBOLS = {}     # (none at present: the experiments left in csrc/lab/ are alternative kernels behind la_gemm's own entry points)




class EncoderBlockC(ctypes.Structure):
    """la_encoder_block (include/lyricalign.h): device pointers of one residual attention block."""
    _fields_ = [(n, c_void_p) for n in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_g", "ln2_b", "w1", "b1", "w2", "b2",
                                        "wqkv_ln", "cqkv", "bqkv_ln", "w1_ln", "c1", "b1_ln",
                                        "wqkv_x2", "wqkv_x2s", "wo_x2", "wo_x2s", "w1_x2", "w1_x2s", "w2_x2", "w2_x2s")]


class EncoderWeightsC(ctypes.Structure):
    """la_encoder_weights."""
    _fields_ = [("dtype", c_int32), ("d", c_int32), ("n_head", c_int32), ("n_layer", c_int32), ("n_mels", c_int32),
                ("conv1_w", c_void_p), ("conv1_b", c_void_p), ("conv2_w", c_void_p), ("conv2_b", c_void_p), ("pos", c_void_p),
                ("lnp_g", c_void_p), ("lnp_b", c_void_p), ("blocks", POINTER(EncoderBlockC))]


class HeadWeightsC(ctypes.Structure):
    """la_head_weights."""
    _fields_ = [("dtype", c_int32), ("hidden", c_int32), ("in_dim", c_int32), ("vocab", c_int32), ("n_layers", c_int32),
                ("w_ih", c_void_p * 2), ("b_ih", c_void_p * 2), ("w_hh", c_void_p * 2), ("b_hh", c_void_p * 2),
                ("w_fc", c_void_p), ("b_fc", c_void_p),
                ("w_ih_x2", c_void_p * 2), ("w_ih_x2s", c_void_p * 2), ("w_fc_x2", c_void_p), ("w_fc_x2s", c_void_p)]


_lib = None


class LyricAlignHipError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load the HIP library (build it with `python -m lyricalignment_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LyricAlignHipError(
                f"{LIB_PATH} is missing: the HIP extension is the product path and has no fallback. "
                "Build it with `python -m lyricalignment_amd.build`.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        if L.la_has_experiments():
            for name, (res, args) in LAB_SYMBOLS.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    msg = lib().la_last_error()
    return msg.decode() if msg else ""


def check(rc: int, what: str = "") -> None:
    """Map la_status to the exception types the reference raises."""
    if rc == LA_OK:
        return
    detail = f"{what}: {last_error()}" if what else last_error()
    if rc == LA_EINVAL:
        raise ValueError(f"liblyricalign_hip invalid argument: {detail}")
    if rc == LA_EUNSUPPORTED:
        raise NotImplementedError(f"liblyricalign_hip unsupported shape: {detail}")
    if rc == LA_ETIMEOUT:
        raise TimeoutError(f"liblyricalign_hip in-kernel wait timed out: {detail}")
    raise LyricAlignHipError(f"liblyricalign_hip status {rc}: {detail}")


def set_option(name: str, value: int) -> None:
    """include/lyricalign.h la_set_option: pin one of the library's choices between shipped kernel forms (takes effect on the next launch)."""
    check(lib().la_set_option(name.encode(), int(value)), "set_option")


def get_option(name: str) -> int:
    v = c_int64(0)
    check(lib().la_get_option(name.encode(), ctypes.byref(v)), "get_option")
    return int(v.value)


class option:
    """with option("gemm_loop", 99): ...  -- the option set for the block, its previous value restored afterwards (tests)."""

    def __init__(self, name: str, value: int):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.prev)
        return False


def has_experiments() -> bool:
    """True for a library built with -DLA_EXPERIMENTS (tools/build_variant.sh): the measured-slower kernel structures and their
    per-launch developer switches are present.  The shipped library returns False."""
    return bool(lib().la_has_experiments())


def ptr(t) -> int:
    if t is None:
        return 0
    return t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return LA_F32
    if dt == torch.bfloat16:
        return LA_BF16
    if dt == torch.float16:
        return LA_F16
    raise ValueError(f"unsupported compute dtype {dt} (float32, bfloat16 or float16)")


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise LyricAlignHipError("no HIP device visible: lyricalignment_amd runs on MI355X (gfx950) only")
    if not lib().la_device_arch_ok():
        raise LyricAlignHipError("current device is not gfx950; the kernels are built for MI355X only")
