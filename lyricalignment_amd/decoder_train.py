"""Training (forward + backward) of the Whisper text decoder on the HIP kernels, float32.

The reference's transcript branch (module/align_model.py:118-121; train_multitask.py:285,308): logits =
whisper_model.decoder(y_in, embed_pad) and F.cross_entropy(logits.permute(0,2,1), y_out) with -100 ignored.
DecoderFunction is whisper/model.py TextDecoder as one autograd node: (tokens [B,n], xa [B,1500,d]) -> logits [B,n,V] with
gradients for every decoder parameter and for xa (which flows on into the encoder).  Same building blocks as
encoder_train.py plus causal score tiles, cross-attention (separate q / kv lengths), the tied embedding projection and
la_embed_tokens_bwd_f32.  Host code only sequences kernels and moves / pads buffers.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _lib, f32x2, ops
from ._lib import check, lib, ptr, stream_ptr
from .encoder_train import add, attention_bwd, attention_bwd_ex, cached_for, gelu, gelu_bwd, layernorm_bwd, scale
from .head_train import colsum, gemm_nn, gemm_tn, grads_to, linear_grads

PER_BLOCK = ("attn_ln.weight", "attn_ln.bias", "attn.query.weight", "attn.query.bias", "attn.key.weight", "attn.value.weight",
             "attn.value.bias", "attn.out.weight", "attn.out.bias",
             "cross_attn_ln.weight", "cross_attn_ln.bias", "cross_attn.query.weight", "cross_attn.query.bias",
             "cross_attn.key.weight", "cross_attn.value.weight", "cross_attn.value.bias", "cross_attn.out.weight",
             "cross_attn.out.bias", "mlp_ln.weight", "mlp_ln.bias", "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias")
NB = len(PER_BLOCK)   # 24


def decoder_param_names(n_layer: int) -> List[str]:
    names = ["token_embedding.weight", "positional_embedding"]
    for i in range(n_layer):
        names += [f"blocks.{i}.{k}" for k in PER_BLOCK]
    return names + ["ln.weight", "ln.bias"]


class DecoderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, xa, n_head, *params):
        _lib.require_gpu()
        dev = xa.device
        P = [p.detach().to(device=dev, dtype=torch.float32).contiguous() for p in params]
        n_layer = (len(P) - 4) // NB
        if len(P) != 4 + n_layer * NB:
            raise ValueError("DecoderFunction: unexpected parameter count")
        tok_emb, pos = P[0], P[1]
        V, d = tok_emb.shape
        H = int(n_head)
        if d != 64 * H:
            raise NotImplementedError("decoder: kernels are built for head_dim 64")
        B, n = tokens.shape
        Ta = xa.shape[1]
        tokens = tokens.to(device=dev, dtype=torch.int64).contiguous()
        xa2 = xa.detach().to(torch.float32).contiguous().view(B * Ta, d)
        x = ops.embed_tokens(tokens, tok_emb, pos)
        saved, packed = [], []
        for i in range(n_layer):
            (g1, be1, wq, bq, wk, wv, bv, wo, bo, gc, bec, wqc, bqc, wkc, wvc, bvc, woc, boc, g2, be2, w1, b1, w2, b2) = \
                P[2 + i * NB: 2 + (i + 1) * NB]
            def build(wq=wq, bq=bq, wk=wk, wv=wv, bv=bv, wqc=wqc, bqc=bqc, wkc=wkc, wvc=wvc, bvc=bvc):
                return (torch.cat([scale(wq, 0.125), wk, wv], 0).contiguous(), torch.cat([scale(bq, 0.125), torch.zeros_like(bq), bv], 0).contiguous(),
                        scale(wqc, 0.125), scale(bqc, 0.125), torch.cat([wkc, wvc], 0).contiguous(),
                        torch.cat([torch.zeros_like(bvc), bvc], 0).contiguous(), f32x2.WeightPlanes())
            # (derived weights and the planes of the audio-side k | v projection, kept while the block's parameters are unchanged)
            src = params[2 + i * NB: 2 + (i + 1) * NB]
            wqkv, bqkv, wq_c, bq_c, wkv_c, bkv_c, wc_kv = cached_for([src[j] for j in (2, 3, 4, 5, 6, 11, 12, 13, 14, 15)], build)
            h1 = ops.layernorm(x, g1, be1, torch.float32)
            qkv = f32x2.linear(h1, wqkv, bias=bqkv)
            # (the row statistic of both attentions goes to the backward: its statistics launch then takes D alone instead of sweeping
            #  the scores once more for the log-sum-exp)
            lse_s = torch.empty((B, H, n), dtype=torch.float32, device=qkv.device)
            lse_c = torch.empty((B, H, n), dtype=torch.float32, device=qkv.device)
            a = ops.attention_ex(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, n, n, H, causal=True, lse=lse_s)
            x1 = f32x2.linear(a, wo, bias=bo, residual=x)
            hc = ops.layernorm(x1, gc, bec, torch.float32)
            qc = f32x2.linear(hc, wq_c, bias=bq_c)
            kv = f32x2.linear(xa2, wkv_c, bias=bkv_c, w_cache=wc_kv)
            ac = ops.attention_ex(qc, kv[:, :d], kv[:, d:], B, n, Ta, H, causal=False, lse=lse_c)
            x2 = f32x2.linear(ac, woc, bias=boc, residual=x1)
            h2 = ops.layernorm(x2, g2, be2, torch.float32)
            u_pre = f32x2.linear(h2, w1, bias=b1)
            x3 = f32x2.linear(u_pre, w2, bias=b2, residual=x2, x_act="gelu")
            saved.append((x, h1, qkv, a, x1, hc, qc, kv, ac, x2, h2, u_pre, lse_s, lse_c))
            packed.append((g1, wqkv, wo, gc, wq_c, wkv_c, woc, g2, w1, w2, wc_kv))
            x = x3
        hf = ops.layernorm(x, P[-2], P[-1], torch.float32)
        wc_tok = cached_for([params[0]], f32x2.WeightPlanes)                # planes of the tied token embedding (51865 x d: 212 MB a split)
        logits = f32x2.linear(hf, tok_emb, w_cache=wc_tok)
        ctx.dims = (B, n, Ta, d, H, V, n_layer, pos.shape[0])
        ctx.saved, ctx.packed = saved, packed
        ctx.tail = (tokens, xa2, x, hf, tok_emb, P[-2])
        ctx.wc_tok = wc_tok
        ctx.xa_needs_grad = xa.requires_grad
        ctx.param_devices = [p.device for p in params]
        return logits.view(B, n, V)

    @staticmethod
    def backward(ctx, dlogits):
        B, n, Ta, d, H, V, n_layer, n_pos = ctx.dims
        tokens, xa2, x_last, hf, tok_emb, ln_g = ctx.tail
        M = B * n
        dev = dlogits.device
        dl = dlogits.to(torch.float32).contiguous().view(M, V)
        grads: List[Optional[torch.Tensor]] = [None] * (4 + NB * n_layer)
        dtok = gemm_tn(dl, hf)                                               # tied projection: logits = hf tok_emb^T
        dx, grads[-2], grads[-1] = layernorm_bwd(x_last, gemm_nn(dl, tok_emb, w_cache=ctx.wc_tok), ln_g)
        dxa = None
        for i in reversed(range(n_layer)):
            x, h1, qkv, a, x1, hc, qc, kv, ac, x2, h2, u_pre, lse_s, lse_c = ctx.saved[i]
            g1, wqkv, wo, gc, wq_c, wkv_c, woc, g2, w1, w2, wc_kv = ctx.packed[i]
            G = [None] * NB
            # MLP
            G[22], G[23] = linear_grads(dx, u_pre, x_act="gelu")
            du_pre = gemm_nn(dx, w2, gelu_grad_of=u_pre)
            G[20], G[21] = linear_grads(du_pre, h2)
            dx2, G[18], G[19] = layernorm_bwd(x2, gemm_nn(du_pre, w1), g2, residual=dx)
            # cross-attention to the audio features
            G[16], G[17] = linear_grads(dx2, ac)
            dac = gemm_nn(dx2, woc)
            dqc = torch.empty((M, d), dtype=torch.float32, device=dev)
            dkv = torch.empty((B * Ta, 2 * d), dtype=torch.float32, device=dev)
            attention_bwd_ex(qc, kv[:, :d], kv[:, d:], dac, dqc, dkv[:, :d], dkv[:, d:], B, n, Ta, H, causal=False, o=ac, lse=lse_c)
            G[11], G[12] = scale(gemm_tn(dqc, hc), 0.125), scale(colsum(dqc), 0.125)
            dwkv, dbkv = linear_grads(dkv, xa2)
            G[13], G[14], G[15] = dwkv[:d], dwkv[d:], dbkv[d:]
            if ctx.xa_needs_grad:
                t = gemm_nn(dkv, wkv_c, w_cache=wc_kv)
                dxa = t if dxa is None else add(dxa, t)
            dx1, G[9], G[10] = layernorm_bwd(x1, gemm_nn(dqc, wq_c), gc, residual=dx2)
            # causal self-attention
            G[7], G[8] = linear_grads(dx1, a)
            dqkv = attention_bwd(qkv, gemm_nn(dx1, wo), B, n, H, causal=True, att=a, lse=lse_s)
            dwqkv, dbqkv = linear_grads(dqkv, h1)
            G[2], G[3] = scale(dwqkv[:d], 0.125), scale(dbqkv[:d], 0.125)
            G[4], G[5], G[6] = dwqkv[d:2 * d], dwqkv[2 * d:], dbqkv[2 * d:]
            dx, G[0], G[1] = layernorm_bwd(x, gemm_nn(dqkv, wqkv), g1, residual=dx1)
            grads[2 + NB * i: 2 + NB * (i + 1)] = G
            ctx.saved[i] = None
        # embeddings: token rows accumulate on top of the tied-projection gradient, positions are summed over the batch
        dtok = dtok.contiguous()
        dpos = torch.zeros((n_pos, d), dtype=torch.float32, device=dev)
        check(lib().la_embed_tokens_bwd_f32(ptr(dx), ptr(tokens), B, n, d, int(dtok.shape[0]), ptr(dtok), ptr(dpos), stream_ptr()), "embed_tokens_bwd")
        grads[0], grads[1] = dtok, dpos
        return (None, dxa.view(B, Ta, d) if dxa is not None else None, None, *grads_to(grads, ctx.param_devices))


def decoder_params(decoder_module, n_layer: Optional[int] = None) -> List[torch.nn.Parameter]:
    named = dict(decoder_module.named_parameters())
    if n_layer is None:
        n_layer = len({k.split(".")[1] for k in named if k.startswith("blocks.")})
    return [named[k] for k in decoder_param_names(n_layer)]


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, scale_grad: float = 1.0, want_grad: bool = True):
    """F.cross_entropy(logits.permute(0,2,1), target) of train_multitask.py:285,308 (ignore_index -100, mean over kept
    tokens): logits [B,n,V] f32 device, target [B,n] -> (loss device scalar, dlogits * scale_grad or None)."""
    _lib.require_gpu()
    if not logits.is_cuda or logits.dtype != torch.float32 or not logits.is_contiguous():
        raise ValueError("cross_entropy: logits must be a contiguous float32 device tensor")
    V = logits.shape[-1]
    R = logits.numel() // V
    tgt = target.to(device=logits.device, dtype=torch.int64).contiguous().view(-1)
    if tgt.numel() != R:
        raise ValueError("cross_entropy: target shape does not match logits")
    loss2 = torch.empty((2,), dtype=torch.float32, device=logits.device)
    ws = torch.empty((2 * R,), dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    check(lib().la_cross_entropy_f32(ptr(logits), V, R, V, ptr(tgt), float(scale_grad), ptr(loss2), ptr(ws), ptr(dl), V,
                                     stream_ptr()), "cross_entropy")
    return loss2[0], dl
