"""Training (forward + backward) of the Whisper audio encoder on the HIP kernels, float32 like the reference's training.

The reference fine-tunes the whole Whisper backbone unless --freeze-encoder is given (train_multitask.py:36-37, 168-175;
whisper/model.py AudioEncoder / ResidualAttentionBlock / MultiHeadAttention).  EncoderFunction is that encoder as one
autograd node: mel [B, n_mels, 3000] -> ln_post output [B, 1500, d], with gradients for every encoder parameter.

Forward reuses the inference kernels in float32 (conv-as-GEMM stem, LayerNorm, MFMA GEMMs, the flash attention kernel) and
keeps the per-block activations.  Backward is a composition of
  * la_gemm / la_gemm_ex (float32 MFMA)   weight / input gradients as K-contiguous "NT" products, batched over heads inside
                                          the packed [T][3d] projections for the attention gradients,
  * la_transpose_pad(_batched)_f32        zero-padded transposes feeding them,
  * la_softmax_rows_f32 / la_softmax_bwd_rows_f32 on score tiles recomputed per clip (P is not kept by the forward kernel),
  * la_layernorm_bwd_sums_f32, la_gelu_bwd_f32, la_col2im3_f32, la_colsum_f32, la_add_f32, la_scale_f32.
Host code only sequences kernels and moves / pads buffers.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import os

import weakref

import torch

from . import _lib, f32x2, ops
from ._lib import check, lib, ptr, stream_ptr
from .head_train import _rup, colsum, gemm_nn, gemm_tn, grads_to, linear_grads

N_FRAMES, N_CTX, C_PAD = 3000, 1500, 128
LA_F32 = 0
PER_BLOCK = ("attn_ln.weight", "attn_ln.bias", "attn.query.weight", "attn.query.bias", "attn.key.weight", "attn.value.weight",
             "attn.value.bias", "attn.out.weight", "attn.out.bias", "mlp_ln.weight", "mlp_ln.bias", "mlp.0.weight", "mlp.0.bias",
             "mlp.2.weight", "mlp.2.bias")


def encoder_param_names(n_layer: int) -> List[str]:
    """openai-whisper AudioEncoder parameter names in EncoderFunction's order (positional_embedding is a buffer)."""
    names = ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias"]
    for i in range(n_layer):
        names += [f"blocks.{i}.{k}" for k in PER_BLOCK]
    return names + ["ln_post.weight", "ln_post.bias"]


def _ew(fn: str, *tensors, n: int):
    check(getattr(lib(), fn)(*[ptr(t) for t in tensors], n, stream_ptr()), fn)


def gelu(x: torch.Tensor) -> torch.Tensor:
    y = torch.empty_like(x)
    _ew("la_gelu_f32", x, y, n=x.numel())
    return y


def gelu_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    dx = torch.empty_like(x)
    _ew("la_gelu_bwd_f32", x, dy, dx, n=x.numel())
    return dx


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    y = torch.empty_like(a)
    _ew("la_add_f32", a, b, y, n=a.numel())
    return y


def scale(x: torch.Tensor, alpha: float) -> torch.Tensor:
    x = x.contiguous()
    y = torch.empty_like(x)
    check(lib().la_scale_f32(ptr(x), float(alpha), ptr(y), x.numel(), stream_ptr()), "scale")
    return y


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, residual: Optional[torch.Tensor] = None):
    """-> dx [M,d] (+ residual: the gradient that bypasses the block), dgamma [d], dbeta [d]"""
    M, d = x.shape
    dx = torch.empty_like(x)
    dg, db = torch.empty((d,), dtype=torch.float32, device=x.device), torch.empty((d,), dtype=torch.float32, device=x.device)
    x, dy, gamma = x.contiguous(), dy.contiguous(), gamma.contiguous()
    residual = residual.contiguous() if residual is not None else None
    # la_layernorm_bwd_sums_f32: x and dy read once, no dy * xhat buffer (these widths, 16-byte aligned rows)
    one_pass = d % 256 == 0 and d <= 2048 and all(t_.data_ptr() % 16 == 0 for t_ in (x, dy, gamma, dx) + ((residual,) if residual is not None else ()))
    t = None if one_pass else torch.empty_like(x)
    check(lib().la_layernorm_bwd_sums_f32(ptr(x), ptr(dy), ptr(gamma), ptr(residual), M, d, ptr(dx), ptr(dg), ptr(db), ptr(t), stream_ptr()),
          "layernorm_bwd_sums")
    return dx, dg, db


def gemm_ex(M, N, K, batch, a, lda, stride_a, w, ldw, stride_w, c, ldc, stride_c, flags: int = 0):
    check(lib().la_gemm_ex(LA_F32, M, N, K, batch, ptr(a), lda, stride_a, ptr(w), ldw, stride_w, ptr(c), ldc, stride_c, None, flags,
                           stream_ptr()), "gemm_ex")


def transpose_batched(src, ld_in, bs_in, rows, cols, out, ld_out, bs_out, out_rows, out_cols, batch):
    check(lib().la_transpose_pad_batched_f32(ptr(src), ld_in, bs_in, rows, cols, ptr(out), ld_out, bs_out, out_rows, out_cols,
                                             batch, stream_ptr()), "transpose_pad_batched")


ATTN_BWD_COMPOSED = os.environ.get("LA_ATTN_BWD", "fused") == "composed"     # developer A/B: the round-1 composition of batched GEMMs


def attention_bwd_ex(q, k, v, do, dq, dk, dv, B: int, Tq: int, Tk: int, H: int, causal: bool = False, o=None, lse=None) -> None:
    """Row views q/do/dq [B*Tq, >=64H], k/v/dk/dv [B*Tk, >=64H] (column slices of packed projections are fine; q pre-scaled
    by 1/8).  With the forward output `o` given: the fused sweeps (scores recomputed per tile, nothing of size Tq x Tk materialised, all
    clips at once) -- la_attention_bwd_f16x2 (the seven products on the f16 pipe at float32 accuracy) from 128 queries and keys on,
    la_attention_bwd_f32 (float32 MFMA) below that or with LA_ATTN_F16X2=0.  Without it (or LA_ATTN_BWD=composed), per clip, for all heads
    at once: S = q k^T and P = softmax(S) are recomputed as whole [H, Tq, Tk] tiles,
    dP = dO v^T, dS = P o (dP - rowsum(dP o P)), dQ = dS k, dK = dS^T q, dV = P^T dO."""
    if o is not None and not ATTN_BWD_COMPOSED and k.stride(0) == v.stride(0) and dk.stride(0) == dv.stride(0):
        import ctypes
        need = ctypes.c_size_t(0)
        if ops.ATTN_F16X2 and Tq >= 128 and Tk >= 128:
            # the seven products on the f16 pipe at float32 accuracy (la_attention_bwd_f16x2; csrc/la_attention_f16x2.hip)
            check(lib().la_attention_bwd_f16x2_workspace_bytes(B, Tq, Tk, H, ctypes.byref(need)), "attention_bwd_f16x2_workspace_bytes")
            ws = torch.empty((need.value + 256,), dtype=torch.uint8, device=q.device)
            off = (-ws.data_ptr()) % 256
            check(lib().la_attention_bwd_f16x2(ptr(q), q.stride(0), ptr(k), ptr(v), k.stride(0), ptr(o), o.stride(0), ptr(do), do.stride(0),
                                               ptr(dq), dq.stride(0), ptr(dk), ptr(dv), dk.stride(0), B, Tq, Tk, H, 1 if causal else 0,
                                               ptr(lse) if lse is not None else None, ws.data_ptr() + off, need.value, stream_ptr()), "attention_bwd_f16x2")
            return
        check(lib().la_attention_bwd_workspace_bytes(B, Tq, H, ctypes.byref(need)), "attention_bwd_workspace_bytes")
        ws = torch.empty((need.value // 4,), dtype=torch.float32, device=q.device)
        check(lib().la_attention_bwd_f32(ptr(q), q.stride(0), ptr(k), ptr(v), k.stride(0), ptr(o), o.stride(0), ptr(do), do.stride(0),
                                         ptr(dq), dq.stride(0), ptr(dk), ptr(dv), dk.stride(0), B, Tq, Tk, H, 1 if causal else 0,
                                         ptr(lse) if lse is not None else None, ptr(ws), need.value, stream_ptr()), "attention_bwd")
        return
    Tqp, Tkp = _rup(Tq), _rup(Tk)
    f = dict(dtype=torch.float32, device=q.device)
    P = torch.empty((H, Tq, Tkp), **f)
    dS = torch.empty((H, Tq, Tkp), **f)
    lq, lk, lv, ldo = q.stride(0), k.stride(0), v.stride(0), do.stride(0)
    in_place = Tk % 4 == 0                # transposed-operand GEMMs (la_gemm_ex flags) read P, dS, q, k, dO as they lie
    if not in_place:
        Pt = torch.empty((H, Tk, Tqp), **f)
        dSt = torch.empty((H, Tk, Tqp), **f)
        Kt = torch.empty((H, 64, Tkp), **f)
        Qt = torch.empty((H, 64, Tqp), **f)
        dOt = torch.empty((H, 64, Tqp), **f)
    TA, TW = _lib.GEMM_TRANS_A, _lib.GEMM_TRANS_W
    for b in range(B):
        qb, dob, dqb = q[b * Tq:(b + 1) * Tq], do[b * Tq:(b + 1) * Tq], dq[b * Tq:(b + 1) * Tq]
        kb, vb = k[b * Tk:(b + 1) * Tk], v[b * Tk:(b + 1) * Tk]
        dkb, dvb = dk[b * Tk:(b + 1) * Tk], dv[b * Tk:(b + 1) * Tk]
        gemm_ex(Tq, Tk, 64, H, qb, lq, 64, kb, lk, 64, P, Tkp, Tq * Tkp)
        check(lib().la_softmax_rows_f32(ptr(P), Tkp, H * Tq, Tk, Tq if causal else 0, stream_ptr()), "softmax_rows")
        gemm_ex(Tq, Tk, 64, H, dob, ldo, 64, vb, lv, 64, dS, Tkp, Tq * Tkp)
        check(lib().la_softmax_bwd_rows_f32(ptr(P), ptr(dS), Tkp, H * Tq, Tk, stream_ptr()), "softmax_bwd_rows")
        if in_place:
            gemm_ex(Tq, 64, Tk, H, dS, Tkp, Tq * Tkp, kb, lk, 64, dqb, dq.stride(0), 64, TW)           # dQ = dS k      (k: [Tk][64] = [K][rows])
            gemm_ex(Tk, 64, Tq, H, dS, Tkp, Tq * Tkp, qb, lq, 64, dkb, dk.stride(0), 64, TA | TW)      # dK = dS^T q
            gemm_ex(Tk, 64, Tq, H, P, Tkp, Tq * Tkp, dob, ldo, 64, dvb, dv.stride(0), 64, TA | TW)     # dV = P^T dO
            continue
        transpose_batched(kb, lk, 64, Tk, 64, Kt, Tkp, 64 * Tkp, 64, Tkp, H)
        transpose_batched(qb, lq, 64, Tq, 64, Qt, Tqp, 64 * Tqp, 64, Tqp, H)
        transpose_batched(dob, ldo, 64, Tq, 64, dOt, Tqp, 64 * Tqp, 64, Tqp, H)
        transpose_batched(P, Tkp, Tq * Tkp, Tq, Tk, Pt, Tqp, Tk * Tqp, Tk, Tqp, H)
        transpose_batched(dS, Tkp, Tq * Tkp, Tq, Tk, dSt, Tqp, Tk * Tqp, Tk, Tqp, H)
        gemm_ex(Tq, 64, Tkp, H, dS, Tkp, Tq * Tkp, Kt, Tkp, 64 * Tkp, dqb, dq.stride(0), 64)       # dQ
        gemm_ex(Tk, 64, Tqp, H, dSt, Tqp, Tk * Tqp, Qt, Tqp, 64 * Tqp, dkb, dk.stride(0), 64)      # dK
        gemm_ex(Tk, 64, Tqp, H, Pt, Tqp, Tk * Tqp, dOt, Tqp, 64 * Tqp, dvb, dv.stride(0), 64)      # dV


def attention_bwd(qkv: torch.Tensor, datt: torch.Tensor, B: int, T: int, H: int, causal: bool = False, att=None, lse=None) -> torch.Tensor:
    """Self-attention over a packed projection: qkv [B*T, 3d] (q pre-scaled), datt [B*T, d] -> dqkv [B*T, 3d];
    att = the forward output [B*T, d] (selects the fused kernel)."""
    d = 64 * H
    dqkv = torch.empty((B * T, 3 * d), dtype=torch.float32, device=qkv.device)
    attention_bwd_ex(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], datt.contiguous(), dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, T, T, H,
                     causal, o=att, lse=lse)
    return dqkv


# Per-layer derived weights (the packed q | k | v projection with the 1/8 folded in) and the split planes of the four weight matrices, kept
# while the parameters are unchanged: key = the block's Parameter objects (weakly held), entry = (state, value, weak references) with
# state = (cache epoch, the parameters' versions, their storage addresses).
# The reference's accumulation loop (train_multitask.py:240-326) runs eight micro-steps between two optimizer steps.  What moves the state:
#   * torch optimizers update `p` in place under no_grad -> p._version;
#   * FineTuner binds parameters as `p.data = flat[...]` views: such a Parameter keeps a version counter of ITS OWN (set_data does not share the
#     flat buffer's), so a kernel that writes the flat buffer through a raw pointer (la_adamw_step_f32) moves neither -- FlatAdamW.step
#     therefore bumps the EPOCH (invalidate_weight_caches) and FineTuner.step the Parameters' own versions;
#   * re-binding p.data to other storage -> the address.
# Writes through `p.data.copy_()` / `p.data.add_()` move nothing torch can see: call invalidate_weight_caches() after them.
# Footprint: derived weights + both split-plane sets (plain, transposed) of every weight ~ 2 x the model's float32 size on the device (incl.
# 2 x 212 MB for the tied token embedding); clear_weight_cache() returns it.
_LAYER_CACHE: Dict[tuple, tuple] = {}
_EPOCH = [0]


def invalidate_weight_caches() -> None:
    """Every cached derived weight / split plane is stale from here on (they are rebuilt on their next use)."""
    _EPOCH[0] += 1


def clear_weight_cache() -> None:
    """Drop the cached derived weights and split planes (device memory ~ 2 x the model's float32 size) now."""
    _LAYER_CACHE.clear()
    _EPOCH[0] += 1


def _purge_dead() -> None:
    for k in [k for k, e in _LAYER_CACHE.items() if any(r() is None for r in e[2])]:      # entries of models that are gone: free their planes
        del _LAYER_CACHE[k]


def cached_for(sources, build):
    """build() once per state of `sources` (tensors, typically the module's Parameters): the entry is theirs -- held by weak references (a
    freed model's storage address and object ids come back with the next one), their versions, their storage addresses and the cache epoch."""
    key = tuple(id(t) for t in sources)
    state = (_EPOCH[0], tuple(t._version for t in sources), tuple(t.data_ptr() for t in sources))
    hit = _LAYER_CACHE.get(key)
    if hit is not None and hit[0] == state and all(r() is t for r, t in zip(hit[2], sources)):
        return hit[1]
    if hit is not None:
        del _LAYER_CACHE[key]                 # the stale value's planes go before the new ones are built
        hit = None
    _purge_dead()
    value = build()
    if len(_LAYER_CACHE) > 512:
        _LAYER_CACHE.clear()
    try:
        _LAYER_CACHE[key] = (state, value, [weakref.ref(t) for t in sources])
    except TypeError:
        pass
    return value


def _layer_weights(layer_params, sources):
    """One encoder block: (wqkv, bqkv, [f32x2.WeightPlanes of wqkv, wo, w1, w2]) from its float32 device tensors; `sources` = the tensors they
    came from (the module's Parameters)."""
    (g1, be1, wq, bq, wk, wv, bv, wo, bo, g2, be2, w1, b1, w2, b2) = layer_params

    def build():
        wqkv = torch.cat([scale(wq, 0.125), wk, wv], 0).contiguous()          # (head_dim^-0.25)^2 folded into q: exact
        bqkv = torch.cat([scale(bq, 0.125), torch.zeros_like(bq), bv], 0).contiguous()
        return wqkv, bqkv, [f32x2.WeightPlanes() for _ in range(4)]
    return cached_for([sources[j] for j in (2, 3, 4, 5, 6, 7, 11, 13)], build)     # wq, bq, wk, wv, bv, wo, w1, w2


class EncoderFunction(torch.autograd.Function):
    """y = AudioEncoder(mel);  params in encoder_param_names() order, `pos` the positional_embedding buffer [1500, d]."""

    @staticmethod
    def forward(ctx, mel, pos, n_head, *params):
        _lib.require_gpu()
        dev = mel.device
        if dev.type != 'cuda':
            raise _lib.LyricAlignHipError('EncoderFunction: mel must be a device tensor')
        P = [p.detach().to(device=dev, dtype=torch.float32).contiguous() for p in params]
        n_layer = (len(P) - 6) // len(PER_BLOCK)
        if len(P) != 6 + n_layer * len(PER_BLOCK):
            raise ValueError("EncoderFunction: unexpected parameter count")
        c1, b1c, c2, b2c = P[:4]
        d, n_mels, _ = c1.shape
        H = int(n_head)
        if d != 64 * H:
            raise NotImplementedError(f"encoder width {d} / heads {H}: kernels are built for head_dim 64")
        if mel.dim() != 3 or mel.shape[1] != n_mels or mel.shape[2] != N_FRAMES:
            raise AssertionError("incorrect audio shape")
        B = mel.shape[0]
        M = B * N_CTX
        f = dict(dtype=torch.float32, device=dev)
        # ---- stem (conv-as-GEMM over zero-bordered channels-last rows) ----
        c1w = torch.zeros((d, 3, C_PAD), **f)
        c1w[:, :, :n_mels] = c1.permute(0, 2, 1)
        c1w = c1w.view(d, 3 * C_PAD)
        c2w = c2.permute(0, 2, 1).reshape(d, 3 * d).contiguous()
        rows0 = ops.mel_to_rows(mel.detach().to(torch.float32).contiguous(), C_PAD, torch.float32)          # [B, 3002, 128]
        pre1 = torch.empty((B * N_FRAMES, d), **f)
        ops.gemm(rows0, c1w, pre1, bias=b1c, M=N_FRAMES, lda=C_PAD, batch=B, stride_a=(N_FRAMES + 2) * C_PAD,
                 stride_c=N_FRAMES * d, ldc=d)
        y1 = torch.zeros((B, N_FRAMES + 2, d), **f)
        y1[:, 1:-1] = gelu(pre1).view(B, N_FRAMES, d)
        pre2 = torch.empty((M, d), **f)
        ops.gemm(y1, c2w, pre2, bias=b2c, M=N_CTX, lda=2 * d, batch=B, stride_a=(N_FRAMES + 2) * d, stride_c=N_CTX * d, ldc=d)
        pos_rep = pos.detach().to(**f).view(1, N_CTX, d).expand(B, N_CTX, d).contiguous().view(M, d)
        x = add(gelu(pre2), pos_rep)
        # ---- blocks ----
        saved = []
        packed = []
        all_mx = f32x2.OperandMax.many(x.device, 8 * n_layer)
        for i in range(n_layer):
            (g1, be1, wq, bq, wk, wv, bv, wo, bo, g2, be2, w1, b1, w2, b2) = P[4 + i * 15: 4 + (i + 1) * 15]
            wqkv, bqkv, wc = _layer_weights(P[4 + i * 15: 4 + (i + 1) * 15], params[4 + i * 15: 4 + (i + 1) * 15])
            # (the operands' maxima from these plain splits scale the transposed splits of the weight gradients: f32x2.OperandMax)
            mx = all_mx[8 * i: 8 * i + 8]                            # 0-3 the activations, 4-7 the weights (for dx = dy w)
            h1 = ops.layernorm(x, g1, be1, torch.float32)
            qkv = f32x2.linear(h1, wqkv, bias=bqkv, x_max=mx[0], w_max=mx[4], w_cache=wc[0])
            lse = torch.empty((B, H, N_CTX), dtype=torch.float32, device=qkv.device)        # row statistic for the fused backward
            att = ops.attention_ex(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, N_CTX, N_CTX, H, lse=lse)
            x_mid = f32x2.linear(att, wo, bias=bo, residual=x, x_max=mx[1], w_max=mx[5], w_cache=wc[1])
            h2 = ops.layernorm(x_mid, g2, be2, torch.float32)
            u_pre = f32x2.linear(h2, w1, bias=b1, x_max=mx[2], w_max=mx[6], w_cache=wc[2])
            x_next = f32x2.linear(u_pre, w2, bias=b2, residual=x_mid, x_act="gelu", x_max=mx[3], w_max=mx[7], w_cache=wc[3])
            saved.append((x, h1, qkv, att, x_mid, h2, u_pre, lse, mx))
            packed.append((g1, wqkv, wo, g2, w1, w2, wc))
            x = x_next
        y = ops.layernorm(x, P[-2], P[-1], torch.float32)
        ctx.dims = (B, d, H, n_mels, n_layer)
        ctx.param_devices = [p.device for p in params]
        ctx.stem = (rows0, pre1, y1, pre2, c2w)
        ctx.saved, ctx.packed, ctx.x_last, ctx.lnp_g = saved, packed, x, P[-2]
        return y.view(B, N_CTX, d)

    @staticmethod
    def backward(ctx, dy):
        from . import head_train
        if head_train._deferred_flags and not head_train.CALLER_CHECKS_FLAGS:
            # a head that ran beside the decoder parked its GRU time-out flags (module/align_model.py): both branches' backward sweeps are
            # enqueued by now -- read them (synchronises) before anything is built on their gradients
            head_train.check_deferred_flags()
        B, d, H, n_mels, n_layer = ctx.dims
        M = B * N_CTX
        dyf = dy.to(torch.float32).contiguous().view(M, d)
        grads: List[Optional[torch.Tensor]] = [None] * (6 + 15 * n_layer)
        dx, grads[-2], grads[-1] = layernorm_bwd(ctx.x_last, dyf, ctx.lnp_g)
        all_my = f32x2.OperandMax.many(dyf.device, 4 * n_layer)
        for i in reversed(range(n_layer)):
            x, h1, qkv, att, x_mid, h2, u_pre, lse, mx = ctx.saved[i]
            g1, wqkv, wo, g2, w1, w2, wc = ctx.packed[i]
            G = [None] * 15
            # every incoming gradient is split plain first (its input-gradient product), which leaves its maximum for the transposed split
            # of the weight-gradient product (no pass for column maxima)
            my = all_my[4 * i: 4 * i + 4]
            # x_next = x_mid + gelu(u_pre) W2^T + b2
            du_pre = gemm_nn(dx, w2, gelu_grad_of=u_pre, a_max=my[0], w_max=mx[7], w_cache=wc[3])
            G[13], G[14] = linear_grads(dx, u_pre, x_act="gelu", dy_max=my[0], x_max=mx[3])
            dh2 = gemm_nn(du_pre, w1, a_max=my[1], w_max=mx[6], w_cache=wc[2])
            G[11], G[12] = linear_grads(du_pre, h2, dy_max=my[1], x_max=mx[2])
            dx_mid, G[9], G[10] = layernorm_bwd(x_mid, dh2, g2, residual=dx)
            # x_mid = x + att Wo^T + bo
            datt = gemm_nn(dx_mid, wo, a_max=my[2], w_max=mx[5], w_cache=wc[1])
            G[7], G[8] = linear_grads(dx_mid, att, dy_max=my[2], x_max=mx[1])
            dqkv = attention_bwd(qkv, datt, B, N_CTX, H, att=att, lse=lse)
            dh1 = gemm_nn(dqkv, wqkv, a_max=my[3], w_max=mx[4], w_cache=wc[0])
            dwqkv, dbqkv = linear_grads(dqkv, h1, dy_max=my[3], x_max=mx[0])
            G[2], G[3] = scale(dwqkv[:d], 0.125), scale(dbqkv[:d], 0.125)
            G[4], G[5], G[6] = dwqkv[d:2 * d], dwqkv[2 * d:], dbqkv[2 * d:]
            dx, G[0], G[1] = layernorm_bwd(x, dh1, g1, residual=dx_mid)
            grads[4 + 15 * i: 4 + 15 * (i + 1)] = G
            ctx.saved[i] = None
        # ---- stem ----
        rows0, pre1, y1, pre2, c2w = ctx.stem
        dpre2 = gelu_bwd(pre2, dx)
        cols2 = y1.as_strided((B, N_CTX, 3 * d), ((N_FRAMES + 2) * d, 2 * d, 1)).contiguous().view(M, 3 * d)   # im2col (copy)
        grads[2] = gemm_tn(dpre2, cols2).view(d, 3, d).permute(0, 2, 1).contiguous()
        grads[3] = colsum(dpre2)
        dcols = gemm_nn(dpre2, c2w)                                                                              # [M, 3d]
        dy1 = torch.empty((B, N_FRAMES + 2, d), dtype=torch.float32, device=dx.device)
        check(lib().la_col2im3_f32(ptr(dcols), B, N_CTX, 2, d, ptr(dy1), N_FRAMES + 2, stream_ptr()), "col2im3")
        dpre1 = gelu_bwd(pre1, dy1[:, 1:-1].contiguous().view(B * N_FRAMES, d))
        cols1 = rows0.as_strided((B, N_FRAMES, 3 * C_PAD), ((N_FRAMES + 2) * C_PAD, C_PAD, 1)).contiguous().view(B * N_FRAMES, 3 * C_PAD)
        grads[0] = gemm_tn(dpre1, cols1).view(d, 3, C_PAD)[:, :, :n_mels].permute(0, 2, 1).contiguous()
        grads[1] = colsum(dpre1)
        return (None, None, None, *grads_to(grads, ctx.param_devices))


def encoder_params(encoder_module, n_layer: Optional[int] = None) -> List[torch.nn.Parameter]:
    """Parameters of a whisper_compat.AudioEncoder (or any module with openai-whisper names) in EncoderFunction's order."""
    named = dict(encoder_module.named_parameters())
    if n_layer is None:
        n_layer = len({k.split(".")[1] for k in named if k.startswith("blocks.")})
    return [named[k] for k in encoder_param_names(n_layer)]
