"""Training (forward + backward) of the alignment head -- nn.GRU(2 layers, bidirectional, inter-layer dropout) -> Mish ->
Linear, module/align_model.py:11-40 -- on the HIP kernels, float32 like the reference's training.

This is the part of the fine-tune step that runs with a frozen encoder (train_multitask.py --freeze-encoder): the encoder is
forward-only, the head's 18 parameter tensors receive gradients.  The backward pass is a composition of
  * la_gru_layer_bwd       persistent backward recurrence (gate pre-activation gradients dgi / dgh per step)
  * la_gemm (float32 MFMA) every weight / input gradient, expressed as K-contiguous "NT" products through
  * la_transpose_pad_f32   zero-padded transposes,
  * la_colsum_f32, la_mish_bwd_f32, la_mask_scale_f32.
Host code here only sequences kernels and moves / pads buffers (torch slicing, zeros, cat).
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib, ops
from ._lib import check, lib, ptr, stream_ptr

_PAD = 32   # float32 GEMM K granule


def _rup(x: int, m: int = _PAD) -> int:
    return (x + m - 1) // m * m


def _padK(a: torch.Tensor) -> torch.Tensor:
    """[M,K] f32 -> contiguous [M,Kp] with zero tail columns (data movement only)."""
    M, K = a.shape
    Kp = _rup(K)
    if Kp == K and a.is_contiguous():
        return a
    out = torch.zeros((M, Kp), dtype=torch.float32, device=a.device)
    out[:, :K] = a
    return out


def transpose_pad(a: torch.Tensor, rows: Optional[int] = None) -> torch.Tensor:
    """a [R,C] (row view, unit inner stride) -> [C, Rp] f32, Rp = R rounded up to 32, zero padded."""
    R, C = a.shape
    Rp = _rup(R)
    out = torch.empty((C, Rp), dtype=torch.float32, device=a.device)
    check(lib().la_transpose_pad_f32(ptr(a), a.stride(0), R, C, ptr(out), Rp, C, Rp, stream_ptr()), "transpose_pad")
    return out


def colsum(a: torch.Tensor) -> torch.Tensor:
    R, C = a.shape
    out = torch.empty((C,), dtype=torch.float32, device=a.device)
    check(lib().la_colsum_f32(ptr(a), a.stride(0), R, C, ptr(out), stream_ptr()), "colsum")
    return out


def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, a_max=None, w_max=None, w_cache=None) -> torch.Tensor:
    """a [M,K] . w [N,K]^T (+ bias), float32 in and out: the large products as three f16 products over split operands (f32x2.linear:
    float32 accuracy at 3/16 of the float32 pipe's cost), the others on the float32 MFMA kernel (K zero-padded to its granule)."""
    if a.shape[1] != w.shape[1]:
        raise ValueError("gemm_nt: K mismatch")
    from . import f32x2
    if f32x2.eligible(a.shape[0], w.shape[0], a.shape[1]):
        return f32x2.linear(a, w, bias=bias, x_max=a_max, w_max=w_max, w_cache=w_cache)
    return ops.gemm(_padK(a), _padK(w), bias=bias, out_f32=True)


def _rows_ok(t: torch.Tensor) -> bool:
    """usable as a transposed ([K][rows]) operand of la_gemm_ex: f32 row view, rows % 4 == 0, 16-byte aligned rows"""
    return (t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.shape[1] % 4 == 0 and t.stride(0) % 4 == 0
            and t.data_ptr() % 16 == 0)


def _gemm_ex_flags(Mg: int, Ng: int, Kg: int, a: torch.Tensor, w: torch.Tensor, flags: int) -> torch.Tensor:
    from ._lib import LA_F32
    out = torch.empty((Mg, Ng), dtype=torch.float32, device=a.device)
    check(lib().la_gemm_ex(LA_F32, Mg, Ng, Kg, 1, ptr(a), a.stride(0), 0, ptr(w), w.stride(0), 0, ptr(out), Ng, 0, None, flags,
                           stream_ptr()), "gemm_ex")
    return out


def linear_grads(dy: torch.Tensor, x: torch.Tensor, x_act: Optional[str] = None, dy_max=None, x_max=None):
    """(dw, db) of y = act(x) w^T + b from dy [M, N] and x [M, K]: dw = dy^T act(x), db = the column sums of dy -- on the f16x2 path from the
    one pass over dy that the operand split makes anyway (la_split_f16x2_t_colsum), otherwise la_colsum_f32.  dy_max / x_max
    (f32x2.OperandMax): the operands' maxima from their plain splits (dy: gemm_nn of the same step; x: the forward's Linear)."""
    from . import f32x2
    db = torch.empty((dy.shape[1],), dtype=torch.float32, device=dy.device)
    return f32x2.gemm_tn(dy, x, x_act=x_act, colsum=db, dy_max=dy_max, x_max=x_max), db


def gemm_tn(a: torch.Tensor, b: torch.Tensor, b_act: Optional[str] = None) -> torch.Tensor:
    """a [M,N]^T . act(b) [M,K] -> [N,K]: the weight gradient of a Linear (b_act = "gelu": of the MLP's second Linear, whose operand is
    gelu(b)).  Large products on the f16x2 path, the rest gemm_tn_f32."""
    from . import f32x2
    return f32x2.gemm_tn(a, b, x_act=b_act)


def gemm_nn(a: torch.Tensor, w: torch.Tensor, gelu_grad_of: Optional[torch.Tensor] = None, a_max=None, w_max=None, w_cache=None) -> torch.Tensor:
    """a [M,N] . w [N,K] -> [M,K]: the input gradient of a Linear (gelu_grad_of = u [M,K]: times gelu'(u), the gradient at the
    pre-activation of the MLP's hidden layer).  Large products on the f16x2 path, the rest gemm_nn_f32."""
    from . import f32x2
    return f32x2.gemm_nn(a, w, gelu_grad_of=gelu_grad_of, dy_max=a_max, w_max=w_max, w_cache=w_cache)


def gemm_tn_f32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a [M,N]^T . b [M,K] -> [N,K]: the weight-gradient shape (contraction over the rows of both).  Both operands are read
    as they lie (la_gemm_ex, LA_GEMM_TRANS_A | LA_GEMM_TRANS_W); shapes the transposed staging does not take go through
    explicit transposes."""
    if a.shape[0] != b.shape[0]:
        raise ValueError("gemm_tn: row mismatch")
    if _rows_ok(a) and _rows_ok(b):
        return _gemm_ex_flags(a.shape[1], b.shape[1], a.shape[0], a, b, _lib.GEMM_TRANS_A | _lib.GEMM_TRANS_W)
    return ops.gemm(transpose_pad(a), transpose_pad(b), out_f32=True)


def gemm_nn_f32(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """a [M,N] . w [N,K] -> [M,K]: the input-gradient shape (w read as it lies: LA_GEMM_TRANS_W)."""
    if a.shape[1] != w.shape[0]:
        raise ValueError("gemm_nn: inner size mismatch")
    if (_rows_ok(w) and a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1 and a.shape[1] % 32 == 0
            and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0):
        return _gemm_ex_flags(a.shape[0], w.shape[1], a.shape[1], a, w, _lib.GEMM_TRANS_W)
    return ops.gemm(_padK(a), transpose_pad(w), out_f32=True)


def grads_to(grads, devices):
    """Gradients back on the device their parameter lives on (a module kept on the host still trains: a copy per step)."""
    return [g if (g is None or g.device == d) else g.to(d) for g, d in zip(grads, devices)]


def _gru_ws(B: int, T: int, H: int, device):
    need = ctypes.c_size_t(0)
    check(lib().la_gru_workspace_bytes(B, T, H, ctypes.byref(need)), "gru_workspace_bytes")
    return torch.empty((need.value,), dtype=torch.uint8, device=device), need.value


# The persistent GRU kernels report a timed-out inter-workgroup wait through a device flag.  Reading it (flag.item()) synchronises the
# host with the stream, which is what keeps the head from running BESIDE the decoder branch (module/align_model.py): with
# DEFER_FLAG_CHECKS set around HeadFunction.apply the flags of that forward and of its backward are parked here and read by
# check_deferred_flags() -- FineTuner calls it after the backward of the step.
DEFER_FLAG_CHECKS = False
_deferred_flags: List = []
# Who reads the parked flags.  False (default): the backward node where the two branches' gradients meet -- EncoderFunction.backward -- reads them
# before it differentiates the encoder, so a plain `loss.backward(); optimizer.step()` loop (train_multitask.py:325-340) gets its TimeoutError out
# of loss.backward(), before any update is computed from a timed-out sweep.  True (FineTuner, around its own backward calls): the caller reads
# them after the backward (check_deferred_flags), with no host synchronisation inside the step.
CALLER_CHECKS_FLAGS = False


def _flag_check(flag: torch.Tensor, what: str, defer: bool) -> None:
    if defer:
        _deferred_flags.append((flag, what))
    elif int(flag.item()) != 0:
        raise TimeoutError(what)


def check_deferred_flags() -> None:
    pending, bad = list(_deferred_flags), None
    del _deferred_flags[:]
    for flag, what in pending:
        if int(flag.item()) != 0:
            bad = what
    if bad:
        raise TimeoutError(bad)


class HeadFunction(torch.autograd.Function):
    """logits = Linear(Mish(GRU(x))) with gradients for the head parameters (and for x when it requires grad)."""

    @staticmethod
    def forward(ctx, x, dropout_p, training, *params):
        # params: per layer l in (0, 1): w_ih, w_hh, b_ih, b_hh, w_ih_rev, w_hh_rev, b_ih_rev, b_hh_rev ; then fc.weight, fc.bias
        _lib.require_gpu()
        B, T, D = x.shape
        dev = x.device
        x0 = x.detach().to(torch.float32).contiguous().view(B * T, D)
        from . import f32x2
        from .encoder_train import cached_for

        def build_layers():
            out = []
            for l in range(2):
                w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r = [p.detach().to(device=dev, dtype=torch.float32) for p in params[8 * l: 8 * l + 8]]
                out.append(dict(w_ih=torch.cat([w_ih, w_ih_r], 0).contiguous(), b_ih=torch.cat([b_ih, b_ih_r], 0).contiguous(),
                                w_hh=torch.stack([w_hh, w_hh_r], 0).contiguous(), b_hh=torch.stack([b_hh, b_hh_r], 0).contiguous(),
                                wc=f32x2.WeightPlanes()))
            return out
        # (the stacked direction pairs and the planes of the input projections / the output Linear, kept while the parameters are unchanged)
        layers = cached_for(list(params[:16]), build_layers)
        w_fc, b_fc = [p.detach().to(device=dev, dtype=torch.float32).contiguous() for p in params[16:18]]
        wc_fc = cached_for([params[16]], f32x2.WeightPlanes)
        ctx.param_devices = [p.device for p in params]
        H = layers[0]["w_hh"].shape[2]
        flag = torch.zeros((1,), dtype=torch.int32, device=dev)
        saved = []
        inp = x0
        mask = None
        for l, lw in enumerate(layers):
            gi = gemm_nt(inp, lw["w_ih"], bias=lw["b_ih"], w_cache=lw["wc"]).view(B, T, 2, 3 * H)
            out = torch.empty((B, T, 2 * H), dtype=torch.float32, device=dev)
            gates = torch.empty((B, T, 2, 4 * H), dtype=torch.float32, device=dev)
            ws, nbytes = _gru_ws(B, T, H, dev)
            check(lib().la_gru_layer_train_fwd(ptr(gi), ptr(lw["w_hh"]), ptr(lw["b_hh"]), ptr(out), ptr(gates), B, T, H, ptr(ws),
                                               nbytes, ptr(flag), stream_ptr()), "gru_layer_train_fwd")
            saved.append((inp, out, gates))
            if l == 0:
                if training and dropout_p > 0.0:
                    mask = torch.empty((B * T, 2 * H), dtype=torch.uint8, device=dev).bernoulli_(1.0 - dropout_p)   # RNG draw only
                    nxt = torch.empty((B * T, 2 * H), dtype=torch.float32, device=dev)
                    check(lib().la_mask_scale_f32(ptr(out), ptr(mask), 1.0 / (1.0 - dropout_p), ptr(nxt), nxt.numel(), stream_ptr()), "mask_scale")
                    inp = nxt
                else:
                    inp = out.view(B * T, 2 * H)
        out1 = saved[1][1].view(B * T, 2 * H)
        act = torch.empty_like(out1)
        check(lib().la_mish_f32(ptr(out1), ptr(act), act.numel(), stream_ptr()), "mish")
        ctx.fc_max = (f32x2.OperandMax(dev), f32x2.OperandMax(dev))      # of act and of fc.weight, for the transposed splits of the backward
        ctx.wc_fc = wc_fc
        logits = gemm_nt(act, w_fc, bias=b_fc, a_max=ctx.fc_max[0], w_max=ctx.fc_max[1], w_cache=wc_fc)
        ctx.defer = bool(DEFER_FLAG_CHECKS)
        _flag_check(flag, "persistent GRU kernel: a bounded inter-workgroup wait timed out", ctx.defer)
        ctx.layers, ctx.w_fc, ctx.saved, ctx.mask, ctx.act = layers, w_fc, saved, mask, act
        ctx.dims, ctx.p, ctx.x_needs_grad = (B, T, D, H), float(dropout_p) if training else 0.0, x.requires_grad
        return logits.view(B, T, -1)

    @staticmethod
    def backward(ctx, dlogits):
        B, T, D, H = ctx.dims
        dev = dlogits.device
        M = B * T
        dl = dlogits.to(torch.float32).contiguous().view(M, -1)
        # ---- Linear ----
        from . import f32x2
        m_dl = f32x2.OperandMax(dev)
        dact = gemm_nn(dl, ctx.w_fc, a_max=m_dl, w_max=ctx.fc_max[1], w_cache=ctx.wc_fc)    # [M, 2H]
        dw_fc, db_fc = linear_grads(dl, ctx.act, dy_max=m_dl, x_max=ctx.fc_max[0])      # [V, 2H], [V]
        # ---- Mish ----
        out1 = ctx.saved[1][1].view(M, 2 * H)
        dout = torch.empty_like(out1)
        check(lib().la_mish_bwd_f32(ptr(out1), ptr(dact), ptr(dout), dout.numel(), stream_ptr()), "mish_bwd")
        grads: List[Optional[torch.Tensor]] = [None] * 18
        flag = torch.zeros((1,), dtype=torch.int32, device=dev)
        dx = None
        for l in (1, 0):
            lw = ctx.layers[l]
            inp, out, gates = ctx.saved[l]
            dgi = torch.empty((B, T, 2, 3 * H), dtype=torch.float32, device=dev)
            dgh = torch.empty((B, T, 2, 3 * H), dtype=torch.float32, device=dev)
            ws, nbytes = _gru_ws(B, T, H, dev)
            check(lib().la_gru_layer_bwd(ptr(gates), ptr(out), ptr(dout), ptr(lw["w_hh"]), ptr(dgi), ptr(dgh), B, T, H, ptr(ws),
                                         nbytes, ptr(flag), stream_ptr()), "gru_layer_bwd")
            dgi2, dgh2 = dgi.view(M, 6 * H), dgh.view(M, 6 * H)
            # layer input as the recurrence saw it
            if l == 1 and ctx.mask is not None:
                xin = torch.empty((M, 2 * H), dtype=torch.float32, device=dev)
                check(lib().la_mask_scale_f32(ptr(ctx.saved[0][1]), ptr(ctx.mask), 1.0 / (1.0 - ctx.p), ptr(xin), xin.numel(), stream_ptr()), "mask_scale")
            else:
                xin = inp if l == 0 else ctx.saved[0][1].view(M, 2 * H)
            dw_ih = gemm_tn(dgi2, xin)                                   # [6H, in]  (forward rows, then reverse rows)
            db_ih = colsum(dgi2)
            db_hh = colsum(dgh2)
            # h_{t-1} as each direction saw it: shift the output sequence by one step, zero at the start (data movement)
            o = out.view(B, T, 2 * H)
            hp_f = torch.zeros((B, T, H), dtype=torch.float32, device=dev)
            hp_r = torch.zeros((B, T, H), dtype=torch.float32, device=dev)
            if T > 1:
                hp_f[:, 1:] = o[:, :-1, :H]
                hp_r[:, :-1] = o[:, 1:, H:]
            dw_hh_f = gemm_tn(dgh2[:, : 3 * H], hp_f.view(M, H))
            dw_hh_r = gemm_tn(dgh2[:, 3 * H:], hp_r.view(M, H))
            base = 8 * l
            grads[base + 0], grads[base + 4] = dw_ih[: 3 * H], dw_ih[3 * H:]
            grads[base + 1], grads[base + 5] = dw_hh_f, dw_hh_r
            grads[base + 2], grads[base + 6] = db_ih[: 3 * H], db_ih[3 * H:]
            grads[base + 3], grads[base + 7] = db_hh[: 3 * H], db_hh[3 * H:]
            if l == 1 or ctx.x_needs_grad:
                dxin = gemm_nn(dgi2, lw["w_ih"], w_cache=lw["wc"])       # [M, in]
                if l == 1:
                    if ctx.mask is not None:
                        dout = torch.empty_like(dxin)
                        check(lib().la_mask_scale_f32(ptr(dxin), ptr(ctx.mask), 1.0 / (1.0 - ctx.p), ptr(dout), dout.numel(), stream_ptr()), "mask_scale")
                    else:
                        dout = dxin
                else:
                    dx = dxin.view(B, T, D)
        grads[16], grads[17] = dw_fc, db_fc
        _flag_check(flag, "persistent GRU backward kernel: a bounded inter-workgroup wait timed out", ctx.defer)
        return (dx, None, None, *grads_to(grads, ctx.param_devices))


def head_params(rnn_module) -> List[torch.nn.Parameter]:
    """The 18 parameter tensors of the reference's RNN module in HeadFunction's order."""
    g = rnn_module.rnn
    out = []
    for l in range(2):
        for suffix in ("", "_reverse"):
            out += [getattr(g, f"weight_ih_l{l}{suffix}"), getattr(g, f"weight_hh_l{l}{suffix}"),
                    getattr(g, f"bias_ih_l{l}{suffix}"), getattr(g, f"bias_hh_l{l}{suffix}")]
    return out + [rnn_module.fc.weight, rnn_module.fc.bias]
