"""Host-side plumbing over the C ABI: torch tensors in, kernels enqueued on the
current HIP stream.  Every wrapper validates shapes / dtypes / contiguity on the
host before a hand-written kernel is launched.  No arithmetic happens here.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (EPI_BIAS, EPI_GELU, EPI_GELU_ERF, EPI_MISH, EPI_OUT_F32, EPI_RESIDUAL, LA_BF16, LA_F32, LA_VARIANT_CTC,
                   LA_VARIANT_PLAIN, check, dtype_code, lib, ptr, stream_ptr)


# The float32 training forward of the attention on the f16 matrix pipe at float32 accuracy (la_attention_lse_f16x2;
# csrc/la_attention_f16x2.hip): sequences of at least 128 queries and keys.  LA_ATTN_F16X2=0 keeps the float32-MFMA kernel (A/B partner).
ATTN_F16X2 = os.environ.get("LA_ATTN_F16X2", "1") != "0"


def _dev(t: torch.Tensor, name: str, dtype=None):
    if not t.is_cuda:
        raise ValueError(f"{name} must be a device tensor")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")


def _capacity(t: torch.Tensor) -> int:
    """Elements addressable from t.data_ptr() to the end of its storage."""
    return t.untyped_storage().nbytes() // t.element_size() - t.storage_offset()


# --------------------------------------------------------------------------- #
# alignment DP                                                                  #
# --------------------------------------------------------------------------- #
def viterbi_batch(em: torch.Tensor, labels: torch.Tensor, n_labels: torch.Tensor, n_frames: torch.Tensor
                  ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """em [B,T,E] f32 (E >= Lmax+1), labels [B,Lmax] i32, n_labels [B] i32, n_frames [B] i32 (device).
    -> onset [B,Lmax] i32, offset [B,Lmax] i32, final_score [B] f64, status [B] i32 (device)."""
    _dev(em, "em", torch.float32); _dev(labels, "labels", torch.int32)
    _dev(n_labels, "n_labels", torch.int32); _dev(n_frames, "n_frames", torch.int32)
    if em.dim() != 3 or labels.dim() != 2 or em.stride(2) != 1 or labels.stride(1) != 1:
        raise ValueError("viterbi_batch: em [B,T,E] / labels [B,Lmax] with unit inner stride expected")
    B, T, E = em.shape
    Lmax = labels.shape[1]
    if labels.shape[0] != B or n_labels.shape != (B,) or n_frames.shape != (B,) or E < Lmax + 1:
        raise ValueError("viterbi_batch: inconsistent shapes")
    n_labels = n_labels.contiguous(); n_frames = n_frames.contiguous()
    onset = torch.empty((B, Lmax), dtype=torch.int32, device=em.device)
    offset = torch.empty((B, Lmax), dtype=torch.int32, device=em.device)
    score = torch.empty((B,), dtype=torch.float64, device=em.device)
    status = torch.empty((B,), dtype=torch.int32, device=em.device)
    need = ctypes.c_size_t(0)
    check(lib().la_viterbi_workspace_bytes(B, T, Lmax, ctypes.byref(need)), "viterbi_workspace_bytes")
    ws = torch.empty((max(need.value, 16),), dtype=torch.uint8, device=em.device)
    check(lib().la_viterbi_batch(ptr(em), em.stride(0), em.stride(1), ptr(labels), labels.stride(0), ptr(n_labels),
                                 ptr(n_frames), B, T, Lmax, ptr(onset), ptr(offset), Lmax, ptr(score), ptr(status),
                                 ptr(ws), need.value, stream_ptr()), "viterbi_batch")
    return onset, offset, score, status


def emissions_from_logits(logits: torch.Tensor, labels: torch.Tensor, n_labels: torch.Tensor, variant: int) -> torch.Tensor:
    """logits [B,T,V] f32 device -> compact emissions [B,T,Lmax+1] f32."""
    _dev(logits, "logits", torch.float32); _dev(labels, "labels", torch.int32); _dev(n_labels, "n_labels", torch.int32)
    if logits.dim() != 3 or logits.stride(2) != 1:
        raise ValueError("emissions_from_logits: logits [B,T,V] with unit inner stride expected")
    B, T, V = logits.shape
    Lmax = labels.shape[1]
    if labels.shape[0] != B or n_labels.shape != (B,) or labels.stride(1) != 1:
        raise ValueError("emissions_from_logits: inconsistent label shapes")
    em = torch.empty((B, T, Lmax + 1), dtype=torch.float32, device=logits.device)
    check(lib().la_emissions_from_logits(ptr(logits), logits.stride(0), logits.stride(1), B, T, V, variant, ptr(labels),
                                         labels.stride(0), ptr(n_labels.contiguous()), Lmax, ptr(em), em.stride(0),
                                         em.stride(1), stream_ptr()), "emissions_from_logits")
    return em


# --------------------------------------------------------------------------- #
# encoder / head building blocks                                                #
# --------------------------------------------------------------------------- #
def gemm(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None, *, bias: Optional[torch.Tensor] = None,
         residual: Optional[torch.Tensor] = None, gelu: bool = False, mish: bool = False, out_f32: bool = False,
         M: Optional[int] = None, lda: Optional[int] = None, batch: int = 1, stride_a: int = 0, stride_c: int = 0,
         stride_r: int = 0, ldc: Optional[int] = None, ldr: Optional[int] = None, out16: Optional[torch.Tensor] = None,
         ln_stats: Optional[torch.Tensor] = None, ln_csum: Optional[torch.Tensor] = None,
         ln_part: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[z][m][n] = epi(sum_k a[z][m][k] w[n][k]).  `a` may be a flat buffer addressed through
    (M, lda, stride_a): that is how the conv-as-GEMM views (overlapping rows) are expressed.
    LayerNorm folded into the neighbouring GEMMs (la_gemm_fused_ln): `out16` = a second, 16-bit copy of the f32 result rows
    (same row pitch / batch stride as out), `ln_part` [N/64, M, 2] = also its per-segment partial row statistics
    (ln_stats_finalize turns them into [M,2]); `ln_stats` [M,2] + `ln_csum` [N] = apply rstd (acc - mean c) before the bias;
    `ln_csum` alone = the same with the row statistics of `a` taken inside the main loop (K % 128 == 0, K >= 256)."""
    _dev(a, "a"); _dev(w, "w")
    dt = dtype_code(w.dtype)
    if a.dtype != w.dtype:
        raise ValueError("gemm: a and w dtypes differ")
    if w.dim() != 2 or not w.is_contiguous():
        raise ValueError("gemm: w must be contiguous [N,K]")
    N, K = w.shape
    if M is None:
        if a.dim() != 2 or a.stride(1) != 1 or a.shape[1] != K:
            raise ValueError("gemm: a must be [M,K] with unit inner stride (or pass M/lda)")
        M, lda = a.shape[0], a.stride(0)
    if _capacity(a) < (batch - 1) * stride_a + (M - 1) * lda + K:
        raise ValueError("gemm: a buffer smaller than the addressed view")
    c_dtype = torch.float32 if (out_f32 or dt == LA_F32) else w.dtype
    if out is None:
        if batch != 1:
            raise ValueError("gemm: batched call needs an explicit out buffer")
        out = torch.empty((M, N), dtype=c_dtype, device=a.device)
    if out.dtype != c_dtype:
        raise ValueError(f"gemm: out must be {c_dtype}")
    ldc = out.stride(-2) if ldc is None else ldc
    if _capacity(out) < (batch - 1) * stride_c + (M - 1) * ldc + N:
        raise ValueError("gemm: out buffer smaller than the addressed view")
    epi = 0
    if bias is not None:
        _dev(bias, "bias", torch.float32)
        if bias.numel() < N:
            raise ValueError("gemm: bias shorter than N")
        epi |= EPI_BIAS
    if residual is not None:
        _dev(residual, "residual", torch.float32)
        ldr = residual.stride(-2) if ldr is None else ldr
        if _capacity(residual) < (batch - 1) * stride_r + (M - 1) * ldr + N:
            raise ValueError("gemm: residual buffer smaller than the addressed view")
        epi |= EPI_RESIDUAL
    if gelu:
        epi |= EPI_GELU
        if gelu == "erf":                # 16-bit results: the erfc-based form instead of the sigmoid fit (LA_EPI_GELU_ERF)
            epi |= EPI_GELU_ERF
    if mish:
        epi |= EPI_MISH
    if c_dtype == torch.float32 and dt != LA_F32:
        epi |= EPI_OUT_F32
    if out16 is not None or ln_stats is not None or ln_csum is not None:
        if out16 is not None:
            _dev(out16, "out16", w.dtype)
            if c_dtype != torch.float32 or _capacity(out16) < (batch - 1) * stride_c + (M - 1) * ldc + N:
                raise ValueError("gemm: out16 accompanies an f32 out of the same layout")
        if ln_part is not None:
            _dev(ln_part, "ln_part", torch.float32)
            if out16 is None or N % 64 or ln_part.numel() < (N // 64) * M * 2 or not ln_part.is_contiguous():
                raise ValueError("gemm: ln_part [N/64, M, 2] goes with out16, N % 64 == 0")
        if ln_stats is not None or ln_csum is not None:      # ln_csum alone: the main loop takes the row statistics itself
            _dev(ln_csum, "ln_csum", torch.float32)
            if ln_csum.numel() < N or batch != 1:
                raise ValueError("gemm: ln_csum [N] expected (batch 1)")
        if ln_stats is not None:
            _dev(ln_stats, "ln_stats", torch.float32)
            if ln_stats.numel() < 2 * M or not ln_stats.is_contiguous():
                raise ValueError("gemm: ln_stats [M,2] expected")
        check(lib().la_gemm_fused_ln(dt, M, N, K, batch, ptr(a), lda, stride_a, ptr(w), ptr(out), ldc, stride_c, ptr(bias),
                                     ptr(residual), ldr or 0, stride_r, epi, ptr(out16), ldc, stride_c, ptr(ln_stats), ptr(ln_csum),
                                     ptr(ln_part), stream_ptr()), "gemm_fused_ln")
        return out
    check(lib().la_gemm(dt, M, N, K, batch, ptr(a), lda, stride_a, ptr(w), ptr(out), ldc, stride_c, ptr(bias),
                        ptr(residual), ldr or 0, stride_r, epi, stream_ptr()), "gemm")
    return out


def ln_stats_finalize(part: torch.Tensor, out: Optional[torch.Tensor] = None, eps: float = 1e-5) -> torch.Tensor:
    """part [slots, M, 2] (mean, sum of squared deviations) of 64-column row segments -> [M,2] (mean, 1/sqrt(var + eps))."""
    _dev(part, "part", torch.float32)
    if part.dim() != 3 or part.shape[2] != 2 or not part.is_contiguous():
        raise ValueError("ln_stats_finalize: [slots, M, 2] expected")
    slots, M, _ = part.shape
    if out is None:
        out = torch.empty((M, 2), dtype=torch.float32, device=part.device)
    if out.shape != (M, 2) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("ln_stats_finalize: bad out buffer")
    check(lib().la_ln_stats_finalize(ptr(part), slots, M, float(eps), ptr(out), stream_ptr()), "ln_stats_finalize")
    return out


def row_stats16(x: torch.Tensor, out: Optional[torch.Tensor] = None, eps: float = 1e-5) -> torch.Tensor:
    """x [M,d] bf16 / f16 rows -> [M,2] f32 (mean, 1/sqrt(var + eps)) per row (two-pass, biased variance like LayerNorm)."""
    _dev(x, "x")
    if x.dim() != 2 or x.stride(1) != 1 or x.dtype not in (torch.bfloat16, torch.float16):
        raise ValueError("row_stats16: [M,d] 16-bit rows expected")
    M, d = x.shape
    if out is None:
        out = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    if out.shape != (M, 2) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("row_stats16: bad out buffer")
    check(lib().la_row_stats16(dtype_code(x.dtype), ptr(x), x.stride(0), M, d, float(eps), ptr(out), stream_ptr()), "row_stats16")
    return out


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out_dtype: torch.dtype,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(x, "x", torch.float32); _dev(gamma, "gamma", torch.float32); _dev(beta, "beta", torch.float32)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("layernorm: x must be [M,d] with unit inner stride")
    M, d = x.shape
    if gamma.numel() != d or beta.numel() != d:
        raise ValueError("layernorm: gamma/beta size mismatch")
    if out is None:
        out = torch.empty((M, d), dtype=out_dtype, device=x.device)
    if out.dtype != out_dtype or out.shape != (M, d) or out.stride(1) != 1:
        raise ValueError("layernorm: bad out buffer")
    check(lib().la_layernorm(ptr(x), x.stride(0), M, d, ptr(gamma), ptr(beta), ptr(out), out.stride(0),
                             dtype_code(out_dtype), stream_ptr()), "layernorm")
    return out


def gemm_split(a: torch.Tensor, w: torch.Tensor, hi: torch.Tensor, lo: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None, in_place: bool = False, gelu: bool = False, M: Optional[int] = None,
               lda: Optional[int] = None, batch: int = 1, stride_a: int = 0, stride_c: int = 0, ld: Optional[int] = None,
               ldr: Optional[int] = None, stride_r: int = 0, ln_part: Optional[torch.Tensor] = None) -> None:
    """(hi, lo) <- epi(a w^T) + residual on the SPLIT residual stream of the 16-bit encoder (la_gemm_split): hi = x rounded to the
    operand dtype [M, N] (the next LayerNorm-folded GEMM's raw operand), lo uint8 [M, N] = the remainder in steps of ulp(hi) / 254.
    residual: f32 rows (the stem's positional embedding), or in_place=True: the stream's own rows (x += ...)."""
    _dev(a, "a"); _dev(w, "w"); _dev(hi, "hi", w.dtype); _dev(lo, "lo", torch.uint8)
    if a.dtype != w.dtype or w.dtype not in (torch.bfloat16, torch.float16):
        raise ValueError("gemm_split: 16-bit operands of one dtype")
    if w.dim() != 2 or not w.is_contiguous():
        raise ValueError("gemm_split: w must be contiguous [N,K]")
    if residual is not None and in_place:
        raise ValueError("gemm_split: residual is either an f32 array or the stream itself")
    N, K = w.shape
    if M is None:
        if a.dim() != 2 or a.stride(1) != 1 or a.shape[1] != K:
            raise ValueError("gemm_split: a must be [M,K] with unit inner stride (or pass M/lda)")
        M, lda = a.shape[0], a.stride(0)
    if _capacity(a) < (batch - 1) * stride_a + (M - 1) * lda + K:
        raise ValueError("gemm_split: a buffer smaller than the addressed view")
    ld = hi.stride(-2) if ld is None else ld
    need = (batch - 1) * stride_c + (M - 1) * ld + N
    if _capacity(hi) < need or _capacity(lo) < need:
        raise ValueError("gemm_split: hi / lo smaller than the addressed view")
    epi = 0
    if bias is not None:
        _dev(bias, "bias", torch.float32)
        if bias.numel() < N:
            raise ValueError("gemm_split: bias shorter than N")
        epi |= EPI_BIAS
    if residual is not None:
        _dev(residual, "residual", torch.float32)
        ldr = residual.stride(-2) if ldr is None else ldr
        if _capacity(residual) < (batch - 1) * stride_r + (M - 1) * ldr + N:
            raise ValueError("gemm_split: residual buffer smaller than the addressed view")
    if residual is not None or in_place:
        epi |= EPI_RESIDUAL
    if gelu:
        epi |= EPI_GELU
    if ln_part is not None:
        _dev(ln_part, "ln_part", torch.float32)
        if N % 64 or ln_part.numel() < (N // 64) * M * 2 or not ln_part.is_contiguous():
            raise ValueError("gemm_split: ln_part [N/64, M, 2], N % 64 == 0")
    check(lib().la_gemm_split(dtype_code(w.dtype), M, N, K, batch, ptr(a), lda, stride_a, ptr(w), ptr(hi), ptr(lo), ld, stride_c, ptr(bias),
                              ptr(residual), ldr or 0, stride_r, epi, ptr(ln_part), stream_ptr()), "gemm_split")


def split_decode(hi: torch.Tensor, lo: torch.Tensor) -> torch.Tensor:
    """The f32 values a split residual stream (hi 16-bit, lo uint8) stands for -- plain torch, for tests and inspection; the
    kernels decode in registers (la_common.h SplitRes): x = hi + (lo - 128) (128 / 127) 2^(e - SH), e the frexp exponent of hi."""
    hf = hi.float()
    _, e = torch.frexp(hf)
    sh = 16 if hi.dtype == torch.bfloat16 else 19
    if hi.dtype == torch.float16:
        e = e.clamp(min=-13)                       # f16 subnormals share the ulp of the smallest normal binade
    step = torch.tensor(128.0 / 127.0, dtype=torch.float32, device=hi.device)
    return hf + torch.ldexp(torch.addcmul(-128.0 * step, lo.float(), step), e.to(torch.int32) - sh)


def layernorm_split(hi: torch.Tensor, lo: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out_dtype: torch.dtype,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """LayerNorm over rows of a split residual stream (la_layernorm_split)."""
    _dev(hi, "hi"); _dev(lo, "lo", torch.uint8); _dev(gamma, "gamma", torch.float32); _dev(beta, "beta", torch.float32)
    if hi.dtype not in (torch.bfloat16, torch.float16) or hi.dim() != 2 or hi.stride(1) != 1 or lo.shape != hi.shape or lo.stride() != hi.stride():
        raise ValueError("layernorm_split: hi [M,d] 16-bit and lo [M,d] uint8 of one layout expected")
    M, d = hi.shape
    if gamma.numel() != d or beta.numel() != d:
        raise ValueError("layernorm_split: gamma/beta size mismatch")
    if out is None:
        out = torch.empty((M, d), dtype=out_dtype, device=hi.device)
    if out.dtype != out_dtype or out.shape != (M, d) or out.stride(1) != 1:
        raise ValueError("layernorm_split: bad out buffer")
    check(lib().la_layernorm_split(dtype_code(hi.dtype), ptr(hi), ptr(lo), hi.stride(0), M, d, ptr(gamma), ptr(beta), ptr(out), out.stride(0),
                                   dtype_code(out_dtype), stream_ptr()), "layernorm_split")
    return out


def mel_to_rows(mel: torch.Tensor, c_pad: int, dtype: torch.dtype) -> torch.Tensor:
    """mel [B,n_mels,frames] f32 -> [B, frames+2, c_pad] channels-last, zero border rows / pad channels."""
    _dev(mel, "mel", torch.float32)
    if mel.dim() != 3 or mel.stride(2) != 1:
        raise ValueError("mel_to_rows: mel [B,n_mels,frames] with unit inner stride expected")
    B, n_mels, frames = mel.shape
    out = torch.empty((B, frames + 2, c_pad), dtype=dtype, device=mel.device)
    check(lib().la_mel_to_rows(ptr(mel), mel.stride(0), mel.stride(1), B, n_mels, frames, ptr(out), c_pad,
                               dtype_code(dtype), stream_ptr()), "mel_to_rows")
    return out


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    _dev(x, "x", torch.float32)
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib().la_cast_f32_to_bf16(ptr(x), ptr(y), x.numel(), stream_ptr()), "cast_f32_to_bf16")
    return y


def attention(qkv: torch.Tensor, batch: int, frames: int, n_head: int, out: Optional[torch.Tensor] = None, q_log2: bool = False) -> torch.Tensor:
    """qkv [batch*frames, 3*n_head*64] (q pre-scaled by 1/8, or by log2(e)/8 with q_log2 -- 16-bit dtypes, LA_Q_LOG2) ->
    out [batch*frames, n_head*64], same dtype."""
    _dev(qkv, "qkv")
    dt = dtype_code(qkv.dtype)
    d = n_head * 64
    if qkv.dim() != 2 or qkv.stride(1) != 1 or qkv.shape[1] != 3 * d or qkv.shape[0] < batch * frames:
        raise ValueError("attention: qkv must be [>=batch*frames, 3*n_head*64] with unit inner stride")
    if out is None:
        out = torch.empty((qkv.shape[0], d), dtype=qkv.dtype, device=qkv.device)
    if out.dtype != qkv.dtype or out.dim() != 2 or out.shape[1] != d or out.shape[0] < batch * frames or out.stride(1) != 1:
        raise ValueError("attention: bad out buffer")
    check(lib().la_attention(dt | (_lib.LA_Q_LOG2 if q_log2 else 0), ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), batch, frames, n_head,
                             stream_ptr()), "attention")
    return out


def gru_layer(gi: torch.Tensor, w_hh: torch.Tensor, b_hh: torch.Tensor, out: Optional[torch.Tensor] = None,
              want_mish: bool = False, out_mish: Optional[torch.Tensor] = None, flag: Optional[torch.Tensor] = None):
    """gi [B,T,2,3H] f32; w_hh [2,3H,H] (f32 | bf16); b_hh [2,3H] f32 -> out [B,T,2H] (w_hh dtype) [, Mish(out)].
    `out_mish`: caller-owned [B,T,2H] buffer for Mish(out) (implies want_mish); `flag`: caller-owned int32 [1] word the
    kernel sets when one of its bounded waits times out (default: a fresh zeroed one per call)."""
    _dev(gi, "gi", torch.float32); _dev(w_hh, "w_hh"); _dev(b_hh, "b_hh", torch.float32)
    dt = dtype_code(w_hh.dtype)
    if gi.dim() != 4 or gi.shape[2] != 2 or not gi.is_contiguous() or not w_hh.is_contiguous() or not b_hh.is_contiguous():
        raise ValueError("gru_layer: gi must be contiguous [B,T,2,3H]")
    B, T, _, H3 = gi.shape
    H = H3 // 3
    if w_hh.shape != (2, 3 * H, H) or b_hh.shape != (2, 3 * H):
        raise ValueError("gru_layer: weight shapes do not match gi")
    if out is None:
        out = torch.empty((B, T, 2 * H), dtype=w_hh.dtype, device=gi.device)
    if out.shape != (B, T, 2 * H) or out.dtype != w_hh.dtype or not out.is_contiguous():
        raise ValueError("gru_layer: bad out buffer")
    if out_mish is not None:
        want_mish = True
        if out_mish.shape != out.shape or out_mish.dtype != out.dtype or not out_mish.is_contiguous():
            raise ValueError("gru_layer: bad out_mish buffer")
    elif want_mish:
        out_mish = torch.empty_like(out)
    need = ctypes.c_size_t(0)
    check(lib().la_gru_workspace_bytes(B, T, H, ctypes.byref(need)), "gru_workspace_bytes")
    ws = torch.empty((need.value,), dtype=torch.uint8, device=gi.device)
    if flag is None:
        flag = torch.zeros((1,), dtype=torch.int32, device=gi.device)
    _dev(flag, "flag", torch.int32)
    check(lib().la_gru_layer(dt, ptr(gi), ptr(w_hh), ptr(b_hh), ptr(out), ptr(out_mish), B, T, H, ptr(ws), need.value,
                             ptr(flag), stream_ptr()), "gru_layer")
    return (out, out_mish, flag) if want_mish else (out, flag)


def fc_emissions(act: torch.Tensor, w_fc: torch.Tensor, b_fc: torch.Tensor, batch: int, frames: int,
                 labels: torch.Tensor, n_labels: torch.Tensor, variant: int, w_x2=None) -> torch.Tensor:
    """act [batch*frames, 2H] (Mish(GRU out)), w_fc [V,2H], b_fc [V] -> compact emissions [batch, frames, Lmax+1] f32.
    The [batch, frames, V] logits are never materialised.  w_x2 = (planes [V,2,2H] f16, inv_scale [V]) of a float32 w_fc: the normaliser
    product on the f16 matrix pipe at float32 accuracy (la_fc_emissions_x2)."""
    _dev(act, "act"); _dev(w_fc, "w_fc"); _dev(b_fc, "b_fc", torch.float32)
    _dev(labels, "labels", torch.int32); _dev(n_labels, "n_labels", torch.int32)
    dt = dtype_code(w_fc.dtype)
    if act.dtype != w_fc.dtype or act.dim() != 2 or act.stride(1) != 1 or not w_fc.is_contiguous():
        raise ValueError("fc_emissions: act [rows,2H] / w_fc [V,2H] of one dtype expected")
    V, K = w_fc.shape
    if act.shape[1] != K or act.shape[0] < batch * frames or b_fc.numel() != V:
        raise ValueError("fc_emissions: inconsistent shapes")
    Lmax = labels.shape[1]
    if labels.shape[0] != batch or n_labels.shape != (batch,) or labels.stride(1) != 1:
        raise ValueError("fc_emissions: inconsistent label shapes")
    em = torch.empty((batch, frames, Lmax + 1), dtype=torch.float32, device=act.device)
    need = ctypes.c_size_t(0)
    if w_x2 is not None:
        planes, inv = w_x2
        _dev(planes, "w_x2 planes", torch.float16); _dev(inv, "w_x2 scales", torch.float32)
        if dt != LA_F32 or tuple(planes.shape) != (V, 2, K) or not planes.is_contiguous() or inv.numel() != V:
            raise ValueError("fc_emissions: w_x2 goes with float32 operands, planes [V, 2, 2H] contiguous and [V] inverse scales")
        check(lib().la_fc_emissions_x2_workspace_bytes(batch, frames, K, V, Lmax, ctypes.byref(need)), "fc_emissions_x2_workspace_bytes")
        ws = torch.empty((need.value,), dtype=torch.uint8, device=act.device)
        check(lib().la_fc_emissions_x2(ptr(act), act.stride(0), ptr(w_fc), ptr(b_fc), ptr(planes), ptr(inv), batch, frames, K, V, variant, ptr(labels),
                                       labels.stride(0), ptr(n_labels.contiguous()), Lmax, ptr(em), em.stride(0), em.stride(1),
                                       ptr(ws), need.value, stream_ptr()), "fc_emissions_x2")
        return em
    check(lib().la_fc_emissions_workspace_bytes(dt, batch, frames, K, V, Lmax, ctypes.byref(need)), "fc_emissions_workspace_bytes")
    ws = torch.empty((need.value,), dtype=torch.uint8, device=act.device)
    check(lib().la_fc_emissions(dt, ptr(act), act.stride(0), ptr(w_fc), ptr(b_fc), batch, frames, K, V, variant, ptr(labels),
                                labels.stride(0), ptr(n_labels.contiguous()), Lmax, ptr(em), em.stride(0), em.stride(1),
                                ptr(ws), need.value, stream_ptr()), "fc_emissions")
    return em


def attention_ex(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, batch: int, q_len: int, kv_len: int, n_head: int,
                 causal: bool = False, out: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None, x2: bool = False) -> torch.Tensor:
    """General attention for the text decoder: q [batch*q_len, >=d] / k, v [batch*kv_len, >=d] row views (column slices of
    packed projections are fine: only the row pitch and the 16-byte alignment matter), q pre-scaled by 1/8."""
    for name, t in (("q", q), ("k", k), ("v", v)):
        _dev(t, name)
        if t.dim() != 2 or t.stride(1) != 1:
            raise ValueError(f"attention_ex: {name} must be a 2-D row view with unit inner stride")
    dt = dtype_code(q.dtype)
    d = n_head * 64
    if k.dtype != q.dtype or v.dtype != q.dtype or k.stride(0) != v.stride(0):
        raise ValueError("attention_ex: q/k/v dtypes differ or k and v have different row pitch")
    if q.shape[0] < batch * q_len or k.shape[0] < batch * kv_len or v.shape[0] < batch * kv_len or min(q.shape[1], k.shape[1], v.shape[1]) < d:
        raise ValueError("attention_ex: views smaller than batch*len x n_head*64")
    if out is None:
        out = torch.empty((batch * q_len, d), dtype=q.dtype, device=q.device)
    if lse is None and x2 and q.dtype == torch.float32 and q_len >= 128 and kv_len >= 128:
        # float32 inference on the f16 matrix pipe at float32 accuracy (la_attention_lse_f16x2 without the row statistic)
        need = ctypes.c_size_t(0)
        check(lib().la_attention_f16x2_workspace_bytes(batch, q_len, kv_len, n_head, ctypes.byref(need)), "attention_f16x2_workspace_bytes")
        ws = torch.empty((need.value + 256,), dtype=torch.uint8, device=q.device)
        off = (-ws.data_ptr()) % 256
        check(lib().la_attention_lse_f16x2(ptr(q), q.stride(0), ptr(k), ptr(v), k.stride(0), ptr(out), out.stride(0), batch, q_len, kv_len,
                                           n_head, 1 if causal else 0, None, ws.data_ptr() + off, need.value, stream_ptr()), "attention_lse_f16x2")
        return out
    if lse is not None:                 # float32 training forward: also the row statistic for la_attention_bwd_f32
        _dev(lse, "lse", torch.float32)
        if q.dtype != torch.float32 or lse.numel() < batch * n_head * q_len or not lse.is_contiguous():
            raise ValueError("attention_ex: lse goes with float32 operands, [batch, n_head, q_len] contiguous")
        if ATTN_F16X2 and q_len >= 128 and kv_len >= 128:
            need = ctypes.c_size_t(0)
            check(lib().la_attention_f16x2_workspace_bytes(batch, q_len, kv_len, n_head, ctypes.byref(need)), "attention_f16x2_workspace_bytes")
            ws = torch.empty((need.value + 256,), dtype=torch.uint8, device=q.device)
            off = (-ws.data_ptr()) % 256
            check(lib().la_attention_lse_f16x2(ptr(q), q.stride(0), ptr(k), ptr(v), k.stride(0), ptr(out), out.stride(0), batch, q_len, kv_len,
                                               n_head, 1 if causal else 0, ptr(lse), ws.data_ptr() + off, need.value, stream_ptr()), "attention_lse_f16x2")
            return out
        check(lib().la_attention_lse_f32(ptr(q), q.stride(0), ptr(k), ptr(v), k.stride(0), ptr(out), out.stride(0), batch, q_len, kv_len,
                                         n_head, 1 if causal else 0, ptr(lse), stream_ptr()), "attention_lse")
        return out
    check(lib().la_attention_ex(dt, ptr(q), q.stride(0), ptr(k), ptr(v), k.stride(0), ptr(out), out.stride(0), batch, q_len,
                                kv_len, n_head, 1 if causal else 0, stream_ptr()), "attention_ex")
    return out


def embed_tokens(tokens: torch.Tensor, token_embedding: torch.Tensor, positional_embedding: torch.Tensor) -> torch.Tensor:
    """tokens int64 [B,n] -> f32 [B*n, d] = token_embedding[tokens] + positional_embedding[:n]."""
    _dev(tokens, "tokens", torch.int64); _dev(token_embedding, "token_embedding", torch.float32)
    _dev(positional_embedding, "positional_embedding", torch.float32)
    B, n = tokens.shape
    V, d = token_embedding.shape
    if positional_embedding.shape[0] < n or positional_embedding.shape[1] != d:
        raise ValueError("embed_tokens: positional table too short or width mismatch")
    x = torch.empty((B * n, d), dtype=torch.float32, device=tokens.device)
    check(lib().la_embed_tokens(ptr(tokens.contiguous()), B, n, ptr(token_embedding.contiguous()), V,
                                ptr(positional_embedding.contiguous()), d, ptr(x), stream_ptr()), "embed_tokens")
    return x


def attention_cached(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, batch: int, q_len: int, kv_len: int, n_head: int,
                     q_batch_rows: int, kv_batch_rows: int, causal: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention of the last q_len positions against a key / value cache holding kv_batch_rows (>= kv_len) rows per clip:
    q [batch*q_batch_rows, >= d], k / v [batch*kv_batch_rows, >= d] row views -> out [batch*q_len, d]."""
    for name, t in (("q", q), ("k", k), ("v", v)):
        _dev(t, name)
        if t.dim() != 2 or t.stride(1) != 1:
            raise ValueError(f"attention_cached: {name} must be a 2-D row view with unit inner stride")
    dt = dtype_code(q.dtype)
    d = n_head * 64
    if k.dtype != q.dtype or v.dtype != q.dtype or k.stride(0) != v.stride(0):
        raise ValueError("attention_cached: q/k/v dtypes differ or k and v have different row pitch")
    if q.shape[0] < (batch - 1) * q_batch_rows + q_len or min(k.shape[0], v.shape[0]) < (batch - 1) * kv_batch_rows + kv_len:
        raise ValueError("attention_cached: views shorter than the addressed rows")
    if out is None:
        out = torch.empty((batch * q_len, d), dtype=q.dtype, device=q.device)
    check(lib().la_attention_cached(dt, ptr(q), q.stride(0), q_batch_rows, ptr(k), ptr(v), k.stride(0), kv_batch_rows, ptr(out),
                                    out.stride(0), batch, q_len, kv_len, n_head, 1 if causal else 0, stream_ptr()), "attention_cached")
    return out


def argmax_rows(x: torch.Tensor) -> torch.Tensor:
    """x [R, C] f32 (row view) -> int64 [R]: index of each row's first maximum."""
    _dev(x, "x", torch.float32)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("argmax_rows: x must be a 2-D row view with unit inner stride")
    out = torch.empty((x.shape[0],), dtype=torch.int64, device=x.device)
    check(lib().la_argmax_rows_f32(ptr(x), x.stride(0), x.shape[0], x.shape[1], ptr(out), stream_ptr()), "argmax_rows")
    return out


def topk_rows(x: torch.Tensor, k: int):
    """x [R, C] f32 row view -> (values [R,k] f32, indices [R,k] int64, lse [R] f32): the k largest entries of each row in
    descending order and the row's log-sum-exp (log-probability = value - lse)."""
    _dev(x, "x", torch.float32)
    if x.dim() != 2 or x.stride(1) != 1 or not 1 <= k <= min(8, x.shape[1]):
        raise ValueError("topk_rows: x must be a 2-D row view and 1 <= k <= min(8, columns)")
    R = x.shape[0]
    vals = torch.empty((R, k), dtype=torch.float32, device=x.device)
    idx = torch.empty((R, k), dtype=torch.int64, device=x.device)
    lse = torch.empty((R,), dtype=torch.float32, device=x.device)
    check(lib().la_topk_rows_f32(ptr(x), x.stride(0), R, x.shape[1], k, ptr(vals), ptr(idx), ptr(lse), stream_ptr()), "topk_rows")
    return vals, idx, lse


# --------------------------------------------------------------------------- #
# model-level entry points (csrc/la_model.cpp)                                  #
# --------------------------------------------------------------------------- #
def _workspace(nbytes: int, device, cache: Optional[dict], key: str) -> torch.Tensor:
    """A 256-byte aligned device byte buffer of at least `nbytes` (torch's caching allocator aligns to 512 B); kept in `cache`
    (one buffer per stage and engine: uses on one stream are ordered) and regrown when a larger one is asked for."""
    if cache is not None:
        t = cache.get(key)
        if t is not None and t.numel() >= nbytes and t.device == device:
            return t
    t = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=device)
    if cache is not None:
        cache[key] = t
    return t


def encoder_forward(weights_c, mel: torch.Tensor, out: torch.Tensor, ws_cache: Optional[dict] = None) -> torch.Tensor:
    """la_encoder_forward: mel [B, n_mels, 3000] f32 -> out [B*1500, d] rows; the whole stem + blocks + ln_post sequence is
    enqueued by ONE C call (whisper_model.embed_audio, module/align_model.py:91)."""
    _dev(mel, "mel", torch.float32); _dev(out, "out")
    if mel.dim() != 3 or mel.shape[2] != 3000 or mel.stride(2) != 1 or mel.shape[1] != weights_c.n_mels:
        raise ValueError("encoder_forward: mel [B, n_mels, 3000] with unit inner stride expected")
    B = mel.shape[0]
    if out.dim() != 2 or out.shape != (B * 1500, weights_c.d) or out.stride(1) != 1:
        raise ValueError("encoder_forward: out must be [B*1500, d] rows")
    need = ctypes.c_size_t(0)
    check(lib().la_encoder_workspace_bytes(ctypes.byref(weights_c), B, ctypes.byref(need)), "encoder_workspace_bytes")
    ws = _workspace(need.value, mel.device, ws_cache, "encoder")
    check(lib().la_encoder_forward(ctypes.byref(weights_c), ptr(mel), mel.stride(0), mel.stride(1), B, ptr(out), out.stride(0),
                                   dtype_code(out.dtype), ptr(ws), ws.numel(), stream_ptr()), "encoder_forward")
    return out


def align_head_forward(weights_c, feats: torch.Tensor, clip_stride_rows: int, batch: int, frames: int, labels: torch.Tensor,
                       n_labels: torch.Tensor, variant: int, flag: torch.Tensor, want_emissions: bool = False,
                       ws_cache: Optional[dict] = None):
    """la_align_head_forward: encoder rows -> BiGRU x 2 -> Mish -> fused FC + emission prep -> DP, ONE C call.
    -> (onset, offset, score, status[, emissions])."""
    _dev(feats, "feats"); _dev(labels, "labels", torch.int32); _dev(n_labels, "n_labels", torch.int32); _dev(flag, "flag", torch.int32)
    if feats.dim() != 2 or feats.stride(1) != 1 or feats.shape[1] != weights_c.in_dim:
        raise ValueError("align_head_forward: feats must be [rows, in_dim] with unit inner stride")
    if feats.shape[0] < (batch - 1) * clip_stride_rows + frames or labels.shape[0] != batch or n_labels.shape != (batch,) or labels.stride(1) != 1:
        raise ValueError("align_head_forward: inconsistent shapes")
    Lmax = labels.shape[1]
    dev = feats.device
    onset = torch.empty((batch, Lmax), dtype=torch.int32, device=dev)
    offset = torch.empty((batch, Lmax), dtype=torch.int32, device=dev)
    score = torch.empty((batch,), dtype=torch.float64, device=dev)
    status = torch.empty((batch,), dtype=torch.int32, device=dev)
    em = torch.empty((batch, frames, Lmax + 1), dtype=torch.float32, device=dev) if want_emissions else None
    need = ctypes.c_size_t(0)
    check(lib().la_align_head_workspace_bytes(ctypes.byref(weights_c), batch, frames, Lmax, ctypes.byref(need)), "align_head_workspace_bytes")
    ws = _workspace(need.value, dev, ws_cache, "head")
    check(lib().la_align_head_forward(ctypes.byref(weights_c), ptr(feats), feats.stride(0), clip_stride_rows, batch, frames, variant,
                                      ptr(labels), labels.stride(0), ptr(n_labels.contiguous()), Lmax, ptr(onset), ptr(offset), Lmax,
                                      ptr(score), ptr(status), ptr(em), ptr(ws), ws.numel(), ptr(flag), stream_ptr()), "align_head_forward")
    return (onset, offset, score, status, em) if want_emissions else (onset, offset, score, status)
