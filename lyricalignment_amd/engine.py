"""AlignEngine: the MI355X hot path of AlignModel as a sequence of HIP kernels.

    mel --conv stem--> x --L x {LN, QKV, attention, out-proj(+res), LN, MLP(+res)}--> ln_post
        --BiGRU x2 (input GEMM + persistent recurrence)--> Mish --fused FC + emission prep--> em
        --batched Viterbi--> onset / offset frames

Replaces, for inference, whisper_model.embed_audio (module/align_model.py:91,101,112,137),
align_rnn (module/align_model.py:35-38) and perform_viterbi(_ctc) (utils/alignment.py) of the
reference.  This file is host plumbing only: weight packing (layout / dtype changes), buffer
management and kernel sequencing on the current HIP stream; every arithmetic step is a kernel
in liblyricalign_hip.so, called through the C ABI (lyricalignment_amd.ops).

Data layout in HBM (DESIGN.md "Layout"): activations are token-major rows [clip*1500 + frame][d];
the residual stream is f32, GEMM operands are in the compute dtype (bf16 or f32); weights keep
nn.Linear's [out][in] layout (both GEMM operands K-contiguous), q/k/v projections are fused into
one [3d][d] matrix with head_dim^-0.5 folded into the q rows; conv weights are [out][tap][in].
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import ctypes
import os

import torch

from . import _lib, ops
from ._lib import LA_VARIANT_CTC, LA_VARIANT_PLAIN

N_FRAMES = 3000   # whisper.audio.N_FRAMES
N_CTX = 1500      # encoder positions
C_PAD = 128       # mel channels padded so 3*C is a multiple of the GEMM K tile
LN_FUSION = os.environ.get("LA_LN_FUSION", "1") != "0"   # developer switch: 0 = always the separate LayerNorm pass
# Row statistics of the folded LayerNorm: "pass" (default) = row_stats16 reads the 16-bit rows back (98 MB, 21 us per LayerNorm);
# "loop" (experiment build of the library only, tools/build_variant.sh) = the CONSUMER GEMM's main loop takes them from the A fragments it multiplies (v_dot2c in MFMA gaps: no pass over the
# stream, no statistics loads in its epilogue -- but 16 more vector instructions per k-step in a loop that is issue- and
# power-bound: the QKV / MLP-up launches run ~7 % slower, 42.7 against 42.2 ms per step, profiles/r4_ab_ln_stats_in_loop.txt);
# "epilogue" = the producer GEMM takes them per 64-column segment while the rows pass through its registers + a finalize kernel
# (costs the residual GEMMs more than the pass it removes).
LN_STATS = os.environ.get("LA_LN_STATS", "pass")
LN_STATS_IN_EPILOGUE = LN_STATS == "epilogue"
# With the LayerNorm fold the residual stream is kept SPLIT (ops.gemm_split: hi 16-bit = the next GEMM's raw operand, + one lo byte
# per element) instead of f32 with a 16-bit copy beside it; 0 = the f32 stream (the A/B partner; la_model.cpp reads the same switch).
RESID_SPLIT = os.environ.get("LA_RESID_SPLIT", "1") != "0"
# The kernel sequences of encode() and of the head + DP exist twice: as ONE C call each (csrc/la_model.cpp: la_encoder_forward,
# la_align_head_forward -- the default) and spelled out below in Python over the op-level calls (LA_ENGINE_PY=1; also what the
# training path and the developer switches above use).  Same kernels, same order, same results.
ENGINE_PY = os.environ.get("LA_ENGINE_PY", "0") == "1"
HEAD_CLIPS_MAX = 256   # clips per head launch set (GRU: 16 workgroup groups of 16 clips co-resident = 192 CUs, out buffer < 2 GiB)


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


@dataclass
class BlockWeights:
    ln1_g: torch.Tensor; ln1_b: torch.Tensor
    wqkv: torch.Tensor; bqkv: torch.Tensor
    wo: torch.Tensor; bo: torch.Tensor
    ln2_g: torch.Tensor; ln2_b: torch.Tensor
    w1: torch.Tensor; b1: torch.Tensor
    w2: torch.Tensor; b2: torch.Tensor
    # LayerNorm folded into the QKV / MLP-up GEMMs (16-bit modes): W' = gamma o W, c = row sums of the ROUNDED W', b' = b + W beta
    wqkv_ln: Optional[torch.Tensor] = None; cqkv: Optional[torch.Tensor] = None; bqkv_ln: Optional[torch.Tensor] = None
    w1_ln: Optional[torch.Tensor] = None; c1: Optional[torch.Tensor] = None; b1_ln: Optional[torch.Tensor] = None
    # float32 mode: the four weight matrices also as f16x2 planes [N, 2, K] + per-row inverse scales [N] (la_split_f16x2, once at pack time):
    # with them the block runs on the f16 matrix pipe at float32 accuracy (la_encoder_forward's x2 route)
    x2: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None      # [(planes, inv_scale)] of wqkv, wo, w1, w2


def _x2_planes(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """A float32 device weight [N, K] (K a multiple of 8) -> its f16x2 planes [N, 2, K] and per-row inverse scales [N] (la_split_f16x2)."""
    N, K = w.shape
    planes = torch.empty((N, 2, K), dtype=torch.float16, device=w.device)
    inv = torch.empty((N,), dtype=torch.float32, device=w.device)
    _lib.check(_lib.lib().la_split_f16x2(_lib.ptr(w), w.stride(0), N, K, _lib.ptr(planes), K, _lib.ptr(inv), _lib.stream_ptr()), "split_f16x2")
    return planes, inv


def x2_inference_on() -> bool:
    """Library option "x2_inference" (include/lyricalign.h): float32 inference on the f16 matrix pipe at float32 accuracy."""
    return _lib.get_option("x2_inference") != 0


def _x2_domain(M: int, N: int, K: int) -> bool:
    """la_gemm_f16x2's domain without split-K slots (the weight planes are packed at their own K)."""
    return K % 128 == 0 and K >= 256 and N > 128 and -(-M // 256) * -(-N // 256) >= 192


def _x2_linear(x: torch.Tensor, w_x2: Tuple[torch.Tensor, torch.Tensor], bias: Optional[torch.Tensor], out: Optional[torch.Tensor] = None,
               residual: Optional[torch.Tensor] = None, x_planes=None) -> torch.Tensor:
    """out [M, N] f32 = x [M, K] f32 (or its ready planes) . w^T + bias (+ residual) with w as packed f16x2 planes: three f16 products at float32 accuracy."""
    from . import f32x2
    planes, inv = w_x2
    N, _, K = planes.shape
    a = x_planes if x_planes is not None else f32x2.split(x, K)
    return f32x2.gemm(a, f32x2.Planes(planes, inv, N, K, K), out=out, bias=bias, residual=residual)


# LA_X2_PACK=0: float32 engines are packed without the f16x2 planes of their weights (every float32 product stays on the float32-MFMA
# kernels; the planes double the packed weights' footprint).  The library option "x2_inference" switches the route at run time instead.
X2_PACK = os.environ.get("LA_X2_PACK", "1") != "0"


def _fold_ln(w: torch.Tensor, b: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, dtype: torch.dtype, device):
    """LN(x) W^T + b = rstd (x W'^T - mean c) + b'   with W' = gamma o W (rounded to `dtype`), c = W'.sum(1), b' = b + W beta."""
    w, b, gamma, beta = (t.detach().double().cpu() for t in (w, b, gamma, beta))
    wl = (w * gamma[None, :]).to(dtype)
    return (wl.to(device).contiguous(), wl.double().sum(dim=1).float().to(device).contiguous(),
            (b + w @ beta).float().to(device).contiguous())


@dataclass
class EncoderWeights:
    d: int
    n_head: int
    n_mels: int
    dtype: torch.dtype
    conv1_w: torch.Tensor; conv1_b: torch.Tensor
    conv2_w: torch.Tensor; conv2_b: torch.Tensor
    pos: torch.Tensor
    blocks: List[BlockWeights]
    lnp_g: torch.Tensor; lnp_b: torch.Tensor
    q_log2: bool = False        # the q rows of wqkv / bqkv carry head_dim^-0.5 log2(e) (include/lyricalign.h LA_Q_LOG2)


@dataclass
class HeadWeights:
    hidden: int
    in_dim: int
    vocab: int
    dtype: torch.dtype
    w_ih: List[torch.Tensor]   # per layer [6H, in] (forward rows then reverse rows)
    b_ih: List[torch.Tensor]   # per layer [6H] f32
    w_hh: List[torch.Tensor]   # per layer [2, 3H, H]
    b_hh: List[torch.Tensor]   # per layer [2, 3H] f32
    w_fc: torch.Tensor         # [V, 2H]
    b_fc: torch.Tensor         # [V] f32
    # float32 mode: f16x2 planes + inverse scales of the input projections and of the output Linear (la_align_head_forward's x2 route)
    w_ih_x2: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None
    w_fc_x2: Optional[Tuple[torch.Tensor, torch.Tensor]] = None


def pack_encoder(sd: Dict[str, torch.Tensor], n_head: int, dtype: torch.dtype, device, prefix: str = "encoder.") -> EncoderWeights:
    """openai-whisper AudioEncoder state_dict -> device weights in kernel layout."""
    g = lambda k: sd[prefix + k]
    d, n_mels, _ = g("conv1.weight").shape
    if d % 64 or d // n_head != 64:
        raise NotImplementedError(f"encoder width {d} / heads {n_head}: kernels are built for head_dim 64")
    c1 = torch.zeros((d, 3, C_PAD), dtype=torch.float32)
    c1[:, :, :n_mels] = g("conv1.weight").detach().float().cpu().permute(0, 2, 1)
    c2 = g("conv2.weight").detach().float().cpu().permute(0, 2, 1).reshape(d, 3 * d)
    pos = g("positional_embedding")
    blocks = []
    i = 0
    scale = 64 ** -0.5  # (head_dim^-0.25 on q) * (head_dim^-0.25 on k), folded into q: exact power of two
    # bfloat16: log2(e) too, so that the scores arrive in the exp2 domain and the attention kernel's softmax needs no multiply
    # (LA_Q_LOG2; the product is rounded to bf16 once, like the plain fold).  LA_ATTN_Q_LOG2=0: developer A/B.
    q_log2 = dtype == torch.bfloat16 and os.environ.get("LA_ATTN_Q_LOG2", "1") != "0"
    if q_log2:
        scale *= 1.4426950408889634
    while f"{prefix}blocks.{i}.attn.query.weight" in sd:
        b = f"blocks.{i}."
        wq, bq = g(b + "attn.query.weight").detach().float() * scale, g(b + "attn.query.bias").detach().float() * scale
        wk = g(b + "attn.key.weight").detach().float()
        wv, bv = g(b + "attn.value.weight").detach().float(), g(b + "attn.value.bias").detach().float()
        wqkv = torch.cat([wq, wk, wv], dim=0)
        bqkv = torch.cat([bq, torch.zeros_like(bq), bv], dim=0)
        blk = BlockWeights(
            _f32(g(b + "attn_ln.weight"), device), _f32(g(b + "attn_ln.bias"), device),
            wqkv.to(device=device, dtype=dtype).contiguous(), _f32(bqkv, device),
            g(b + "attn.out.weight").detach().to(device=device, dtype=dtype).contiguous(), _f32(g(b + "attn.out.bias"), device),
            _f32(g(b + "mlp_ln.weight"), device), _f32(g(b + "mlp_ln.bias"), device),
            g(b + "mlp.0.weight").detach().to(device=device, dtype=dtype).contiguous(), _f32(g(b + "mlp.0.bias"), device),
            g(b + "mlp.2.weight").detach().to(device=device, dtype=dtype).contiguous(), _f32(g(b + "mlp.2.bias"), device))
        if dtype == torch.float32 and X2_PACK and d % 128 == 0 and d >= 256 and torch.device(device).type == "cuda":
            blk.x2 = [_x2_planes(t) for t in (blk.wqkv, blk.wo, blk.w1, blk.w2)]
        if dtype in (torch.bfloat16, torch.float16):
            blk.wqkv_ln, blk.cqkv, blk.bqkv_ln = _fold_ln(wqkv, bqkv, g(b + "attn_ln.weight"), g(b + "attn_ln.bias"), dtype, device)
            blk.w1_ln, blk.c1, blk.b1_ln = _fold_ln(g(b + "mlp.0.weight"), g(b + "mlp.0.bias"), g(b + "mlp_ln.weight"),
                                                    g(b + "mlp_ln.bias"), dtype, device)
        blocks.append(blk)
        i += 1
    return EncoderWeights(d, n_head, n_mels, dtype,
                          c1.reshape(d, 3 * C_PAD).to(device=device, dtype=dtype).contiguous(), _f32(g("conv1.bias"), device),
                          c2.to(device=device, dtype=dtype).contiguous(), _f32(g("conv2.bias"), device),
                          _f32(pos, device), blocks, _f32(g("ln_post.weight"), device), _f32(g("ln_post.bias"), device), q_log2)


@dataclass
class DecoderBlockWeights:
    ln1_g: torch.Tensor; ln1_b: torch.Tensor
    wqkv: torch.Tensor; bqkv: torch.Tensor
    wo: torch.Tensor; bo: torch.Tensor
    lnc_g: torch.Tensor; lnc_b: torch.Tensor
    wq_c: torch.Tensor; bq_c: torch.Tensor
    wkv_c: torch.Tensor; bkv_c: torch.Tensor
    wo_c: torch.Tensor; bo_c: torch.Tensor
    ln2_g: torch.Tensor; ln2_b: torch.Tensor
    w1: torch.Tensor; b1: torch.Tensor
    w2: torch.Tensor; b2: torch.Tensor


@dataclass
class DecoderWeights:
    d: int
    n_head: int
    dtype: torch.dtype
    tok_emb_f32: torch.Tensor   # [V, d] f32 (lookup)
    tok_emb: torch.Tensor       # [V, d] compute dtype (tied output projection)
    pos: torch.Tensor           # [n_text_ctx, d] f32
    blocks: List[DecoderBlockWeights]
    ln_g: torch.Tensor; ln_b: torch.Tensor


def pack_decoder(sd: Dict[str, torch.Tensor], n_head: int, dtype: torch.dtype, device, prefix: str = "decoder.") -> DecoderWeights:
    """openai-whisper TextDecoder state_dict -> device weights (whisper/model.py TextDecoder; module/align_model.py:118-121)."""
    g = lambda k: sd[prefix + k].detach().float()
    w = lambda t: t.to(device=device, dtype=dtype).contiguous()
    d = g("token_embedding.weight").shape[1]
    if d // n_head != 64:
        raise NotImplementedError("decoder: kernels are built for head_dim 64")
    scale = 64 ** -0.5
    blocks, i = [], 0
    while f"{prefix}blocks.{i}.attn.query.weight" in sd:
        b = f"blocks.{i}."
        bq = g(b + "attn.query.bias") * scale
        wqkv = torch.cat([g(b + "attn.query.weight") * scale, g(b + "attn.key.weight"), g(b + "attn.value.weight")], 0)
        bqkv = torch.cat([bq, torch.zeros_like(bq), g(b + "attn.value.bias")], 0)
        wkv_c = torch.cat([g(b + "cross_attn.key.weight"), g(b + "cross_attn.value.weight")], 0)
        bv_c = g(b + "cross_attn.value.bias")
        bkv_c = torch.cat([torch.zeros_like(bv_c), bv_c], 0)
        blocks.append(DecoderBlockWeights(
            _f32(g(b + "attn_ln.weight"), device), _f32(g(b + "attn_ln.bias"), device), w(wqkv), _f32(bqkv, device),
            w(g(b + "attn.out.weight")), _f32(g(b + "attn.out.bias"), device),
            _f32(g(b + "cross_attn_ln.weight"), device), _f32(g(b + "cross_attn_ln.bias"), device),
            w(g(b + "cross_attn.query.weight") * scale), _f32(g(b + "cross_attn.query.bias") * scale, device),
            w(wkv_c), _f32(bkv_c, device), w(g(b + "cross_attn.out.weight")), _f32(g(b + "cross_attn.out.bias"), device),
            _f32(g(b + "mlp_ln.weight"), device), _f32(g(b + "mlp_ln.bias"), device),
            w(g(b + "mlp.0.weight")), _f32(g(b + "mlp.0.bias"), device), w(g(b + "mlp.2.weight")), _f32(g(b + "mlp.2.bias"), device)))
        i += 1
    te = g("token_embedding.weight")
    return DecoderWeights(d, n_head, dtype, _f32(te, device), w(te), _f32(g("positional_embedding"), device), blocks,
                          _f32(g("ln.weight"), device), _f32(g("ln.bias"), device))


def pack_head(sd: Dict[str, torch.Tensor], dtype: torch.dtype, device, prefix: str = "align_rnn.") -> HeadWeights:
    """RNN state_dict (nn.GRU 2 layers + nn.Linear, module/align_model.py:23-33) -> kernel layout.
    bidirectional=False (module/align_model.py:20,48; the reference's scripts always build the bidirectional head) runs on the
    same kernels with an all-zero reverse direction: zero W_ih / W_hh / biases keep that direction's state at exactly 0
    (n = tanh(0) = 0, h' = (1 - z) * 0 + z * 0), and the next layer's / the Linear's columns that would read it are zero
    columns appended to their weights -- so the forward direction's arithmetic is unchanged and the results are those of the
    unidirectional nn.GRU, at the price of the idle half of the recurrence."""
    g = lambda k: sd[prefix + k].detach()
    H = g("rnn.weight_hh_l0").shape[1]
    bidir = f"{prefix}rnn.weight_ih_l0_reverse" in sd
    w_ih, b_ih, w_hh, b_hh = [], [], [], []
    layer = 0
    while f"{prefix}rnn.weight_ih_l{layer}" in sd:
        wf, bf_, hf, cf = g(f"rnn.weight_ih_l{layer}").float(), g(f"rnn.bias_ih_l{layer}").float(), g(f"rnn.weight_hh_l{layer}").float(), g(f"rnn.bias_hh_l{layer}").float()
        if bidir:
            wr, br, hr, cr = (g(f"rnn.{n}_l{layer}_reverse").float() for n in ("weight_ih", "bias_ih", "weight_hh", "bias_hh"))
        else:
            if layer > 0:                                   # layer input = [forward | (absent) reverse]: zero columns for the latter
                wf = torch.cat([wf, torch.zeros_like(wf)], dim=1)
            wr, br, hr, cr = torch.zeros_like(wf), torch.zeros_like(bf_), torch.zeros_like(hf), torch.zeros_like(cf)
        w_ih.append(torch.cat([wf, wr], 0).to(device=device, dtype=dtype).contiguous())
        b_ih.append(_f32(torch.cat([bf_, br], 0), device))
        w_hh.append(torch.stack([hf, hr], 0).to(device=device, dtype=dtype).contiguous())
        b_hh.append(_f32(torch.stack([cf, cr], 0), device))
        layer += 1
    w_fc = g("fc.weight").float()
    if not bidir:
        w_fc = torch.cat([w_fc, torch.zeros_like(w_fc)], dim=1)
    hw = HeadWeights(H, w_ih[0].shape[1], w_fc.shape[0], dtype, w_ih, b_ih, w_hh, b_hh,
                     w_fc.to(device=device, dtype=dtype).contiguous(), _f32(g("fc.bias"), device))
    if (dtype == torch.float32 and X2_PACK and torch.device(device).type == "cuda" and len(w_ih) == 2
            and all(w.shape[1] % 8 == 0 for w in w_ih) and hw.w_fc.shape[1] % 8 == 0):
        hw.w_ih_x2 = [_x2_planes(w) for w in hw.w_ih]
        hw.w_fc_x2 = _x2_planes(hw.w_fc)
    return hw


class AlignEngine:
    """Owns packed weights + scratch buffers for one (model, compute dtype, device)."""

    def __init__(self, enc: EncoderWeights, head: Optional[HeadWeights], device="cuda", dec: Optional[DecoderWeights] = None):
        _lib.require_gpu()
        self.enc, self.head, self.dec, self.device = enc, head, dec, torch.device(device)
        self._buf: Dict[Tuple, torch.Tensor] = {}
        self._ws: Dict[str, torch.Tensor] = {}          # workspaces of the model-level C entry points
        self._enc_c = self._encoder_struct(enc)
        self._head_c = self._head_struct(head) if head is not None and len(head.w_ih) == 2 else None
        self._gru_flag: Optional[torch.Tensor] = None   # timeout word shared by this engine's GRU launches (check_gru reads it)

    # ---- C structs of the packed weights (la_encoder_weights / la_head_weights) -------------
    @staticmethod
    def _encoder_struct(e: EncoderWeights):
        P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        blocks = (_lib.EncoderBlockC * max(1, len(e.blocks)))()
        for i, b in enumerate(e.blocks):
            x2 = [P(t) for pair in (b.x2 or [(None, None)] * 4) for t in pair]
            blocks[i] = _lib.EncoderBlockC(P(b.ln1_g), P(b.ln1_b), P(b.wqkv), P(b.bqkv), P(b.wo), P(b.bo), P(b.ln2_g), P(b.ln2_b),
                                           P(b.w1), P(b.b1), P(b.w2), P(b.b2), P(b.wqkv_ln), P(b.cqkv), P(b.bqkv_ln), P(b.w1_ln), P(b.c1), P(b.b1_ln), *x2)
        c = _lib.EncoderWeightsC(_lib.dtype_code(e.dtype) | (_lib.LA_Q_LOG2 if e.q_log2 else 0), e.d, e.n_head, len(e.blocks), e.n_mels, P(e.conv1_w), P(e.conv1_b), P(e.conv2_w),
                                 P(e.conv2_b), P(e.pos), P(e.lnp_g), P(e.lnp_b), blocks)
        c._keep = blocks            # the struct points into this host array
        return c

    @staticmethod
    def _head_struct(h: HeadWeights):
        V2 = ctypes.c_void_p * 2
        P = lambda t: t.data_ptr()
        Pn = lambda t: t.data_ptr() if t is not None else None
        ih = h.w_ih_x2 or [(None, None)] * 2
        fc = h.w_fc_x2 or (None, None)
        return _lib.HeadWeightsC(_lib.dtype_code(h.dtype), h.hidden, h.in_dim, h.vocab, 2, V2(P(h.w_ih[0]), P(h.w_ih[1])), V2(P(h.b_ih[0]), P(h.b_ih[1])),
                                 V2(P(h.w_hh[0]), P(h.w_hh[1])), V2(P(h.b_hh[0]), P(h.b_hh[1])), P(h.w_fc), P(h.b_fc),
                                 V2(Pn(ih[0][0]), Pn(ih[1][0])), V2(Pn(ih[0][1]), Pn(ih[1][1])), Pn(fc[0]), Pn(fc[1]))

    # ---- scratch -----------------------------------------------------------------
    def _get(self, name: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        key = (name, tuple(shape), dtype)
        t = self._buf.get(key)
        if t is None:
            for k in [k for k in self._buf if k[0] == name]:
                del self._buf[k]
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._buf[key] = t
        return t

    # ---- encoder: Whisper.embed_audio -----------------------------------------------
    def encode(self, mel: torch.Tensor, out_dtype: Optional[torch.dtype] = None, slot: int = 0,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """mel [B, n_mels, 3000] f32 (device) -> ln_post output [B*1500, d] in `out_dtype` (default: compute dtype).
        `slot` names the output buffer (the two-stream pipeline double-buffers it); `out` is a caller-owned
        [B*1500, d] row view to write into instead."""
        e = self.enc
        if mel.dim() != 3 or mel.shape[1] != e.n_mels or mel.shape[2] != N_FRAMES:
            raise AssertionError("incorrect audio shape")  # whisper AudioEncoder asserts the same
        mel = mel.to(device=self.device, dtype=torch.float32)
        if mel.stride(2) != 1:
            mel = mel.contiguous()
        B, d, dt = mel.shape[0], e.d, e.dtype
        M = B * N_CTX
        if not ENGINE_PY and (LN_FUSION or dt == torch.float32) and not LN_STATS_IN_EPILOGUE:
            out_dtype = out_dtype or dt
            y = out if out is not None else self._get(f"enc_out{slot}", (M, d), out_dtype)
            return ops.encoder_forward(self._enc_c, mel, y, self._ws)          # the sequence below, enqueued by one C call
        rows0 = ops.mel_to_rows(mel, C_PAD, dt)                                     # [B, 3002, 128]
        y1 = self._get("y1", (B, N_FRAMES + 2, d), dt, zero=True)                    # border rows stay zero
        ops.gemm(rows0, e.conv1_w, y1.view(-1)[d:], bias=e.conv1_b, gelu=True, M=N_FRAMES, lda=C_PAD, batch=B,
                 stride_a=(N_FRAMES + 2) * C_PAD, stride_c=(N_FRAMES + 2) * d, ldc=d)
        x = self._get("x", (M, d), torch.float32)
        h = self._get("h", (M, d), dt)
        qkv = self._get("qkv", (M, 3 * d), dt)
        att = self._get("att", (M, d), dt)
        u = self._get("u", (M, 4 * d), dt)
        # LayerNorm folded into the GEMMs around it: the GEMMs that write the f32 residual stream x also store it rounded to
        # bf16 (h = raw x), a small kernel takes the row statistics of h, and the QKV / MLP-up GEMMs apply
        # rstd (acc - mean c) + b' in their epilogue on gamma-folded weights.  Only where every GEMM of a block runs on the
        # 256x256 kernel (>= 192 tiles, i.e. >= 9 clips at d = 1024) and in the 16-bit modes (float16: the raw stream must stay
        # below 65504, as it must for whisper's own fp16 inference); otherwise the separate LayerNorm pass.
        fused = (LN_FUSION and dt in (torch.bfloat16, torch.float16) and e.blocks and e.blocks[0].wqkv_ln is not None and d > 128
                 and -(-M // 256) * -(-d // 256) >= 192 and -(-N_CTX // 256) * -(-d // 256) * B >= 192)
        split = fused and RESID_SPLIT
        if split:
            lo = x.view(torch.uint8).view(-1)[: M * d].view(M, d)         # the stream's lo bytes live in the first quarter of x
            ops.gemm_split(y1, e.conv2_w, h, lo, bias=e.conv2_b, gelu=True, residual=e.pos, M=N_CTX, lda=2 * d, batch=B,
                           stride_a=(N_FRAMES + 2) * d, stride_c=N_CTX * d, ld=d, ldr=d, stride_r=0)
        else:
            ops.gemm(y1, e.conv2_w, x, bias=e.conv2_b, gelu=True, residual=e.pos, out_f32=True, M=N_CTX, lda=2 * d, batch=B,
                     stride_a=(N_FRAMES + 2) * d, stride_c=N_CTX * d, ldc=d, ldr=d, stride_r=0, out16=h if fused else None)
        if fused:
            stats = self._get("ln_stats", (M, 2), torch.float32)
            part = self._get("ln_part", (d // 64, M, 2), torch.float32) if (LN_STATS_IN_EPILOGUE and d % 64 == 0) else None

            # (experiment build only, bfloat16: the consumer GEMM's main loop takes the row statistics itself)
            in_loop = (LN_STATS == "loop" and dt == torch.bfloat16 and d % 128 == 0 and d >= 256 and _lib.has_experiments()
                       and os.environ.get("LA_PP_DBG") not in ("99", "73"))
            if in_loop:
                stats = None                                                                  # ln_csum alone: the main loop takes them

            def row_stats():
                if in_loop:
                    return
                if part is not None:
                    ops.ln_stats_finalize(part, out=stats)
                else:
                    ops.row_stats16(h, out=stats)

            if not in_loop:
                ops.row_stats16(h, out=stats)                                                 # of the stem's output (batched GEMM)
            for blk in e.blocks:
                ops.gemm(h, blk.wqkv_ln, qkv, bias=blk.bqkv_ln, ln_stats=stats, ln_csum=blk.cqkv)
                ops.attention(qkv, B, N_CTX, e.n_head, out=att, q_log2=e.q_log2)
                if split:
                    ops.gemm_split(att, blk.wo, h, lo, bias=blk.bo, in_place=True, ln_part=part)         # (h, lo) += out-proj
                else:
                    ops.gemm(att, blk.wo, x, bias=blk.bo, residual=x, out_f32=True, out16=h, ln_part=part)   # x += out-proj; h = bf16(x)
                row_stats()
                ops.gemm(h, blk.w1_ln, u, bias=blk.b1_ln, gelu=True, ln_stats=stats, ln_csum=blk.c1)
                if split:
                    ops.gemm_split(u, blk.w2, h, lo, bias=blk.b2, in_place=True, ln_part=part)           # (h, lo) += mlp
                else:
                    ops.gemm(u, blk.w2, x, bias=blk.b2, residual=x, out_f32=True, out16=h, ln_part=part)     # x += mlp; h = bf16(x)
                row_stats()
        elif (dt == torch.float32 and e.blocks and all(b_.x2 is not None for b_ in e.blocks) and x2_inference_on() and d <= 4096
              and -(-M // 256) * -(-d // 256) >= 192):
            # float32 on the f16 matrix pipe at float32 accuracy: la_encoder_forward's x2 route, spelled out over the op-level calls
            from . import f32x2
            pl = lambda t: f32x2.Planes(t[0], t[1], t[0].shape[0], t[0].shape[2], t[0].shape[2])
            for blk in e.blocks:
                wq, wo_, w1_, w2_ = (pl(t) for t in blk.x2)
                f32x2.gemm(f32x2.layernorm_split(x, blk.ln1_g, blk.ln1_b), wq, out=qkv, bias=blk.bqkv)
                ops.attention_ex(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, N_CTX, N_CTX, e.n_head, out=att, x2=True)
                f32x2.gemm(f32x2.split(att, d), wo_, out=x, bias=blk.bo, residual=x)
                f32x2.gemm(f32x2.layernorm_split(x, blk.ln2_g, blk.ln2_b), w1_, out=u, bias=blk.b1)
                f32x2.gemm(f32x2.split(u, 4 * d, act="gelu"), w2_, out=x, bias=blk.b2, residual=x)
        else:
            for blk in e.blocks:
                ops.layernorm(x, blk.ln1_g, blk.ln1_b, dt, out=h)
                ops.gemm(h, blk.wqkv, qkv, bias=blk.bqkv)
                ops.attention(qkv, B, N_CTX, e.n_head, out=att, q_log2=e.q_log2)
                ops.gemm(att, blk.wo, x, bias=blk.bo, residual=x, out_f32=True)          # x += out-proj (in place)
                ops.layernorm(x, blk.ln2_g, blk.ln2_b, dt, out=h)
                ops.gemm(h, blk.w1, u, bias=blk.b1, gelu=True)
                ops.gemm(u, blk.w2, x, bias=blk.b2, residual=x, out_f32=True)            # x += mlp (in place)
        out_dtype = out_dtype or dt
        y = out if out is not None else self._get(f"enc_out{slot}", (M, d), out_dtype)
        if split:
            ops.layernorm_split(h, lo, e.lnp_g, e.lnp_b, out_dtype, out=y)
        else:
            ops.layernorm(x, e.lnp_g, e.lnp_b, out_dtype, out=y)
        return y

    # ---- text decoder: Whisper.logits(tokens, audio_features) --------------------------------
    def decode(self, tokens: torch.Tensor, xa: torch.Tensor, n_audio: int = N_CTX) -> torch.Tensor:
        """tokens int64 [B,n]; xa = encoder output rows [B*n_audio, d] in the compute dtype -> logits [B,n,n_vocab] f32
        (whisper TextDecoder.forward: embeddings, L x {causal self-attention, cross-attention to xa, MLP}, ln, tied projection)."""
        dw = self.dec
        if dw is None:
            raise _lib.LyricAlignHipError("this engine was packed without decoder weights")
        B, n = tokens.shape
        d, dt, H = dw.d, dw.dtype, dw.n_head
        tokens = tokens.to(device=self.device, dtype=torch.int64)
        x = ops.embed_tokens(tokens, dw.tok_emb_f32, dw.pos)
        h = torch.empty((B * n, d), dtype=dt, device=self.device)
        for blk in dw.blocks:
            ops.layernorm(x, blk.ln1_g, blk.ln1_b, dt, out=h)
            qkv = ops.gemm(h, blk.wqkv, bias=blk.bqkv)
            a = ops.attention_ex(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, n, n, H, causal=True)
            ops.gemm(a, blk.wo, x, bias=blk.bo, residual=x, out_f32=True)
            ops.layernorm(x, blk.lnc_g, blk.lnc_b, dt, out=h)
            q = ops.gemm(h, blk.wq_c, bias=blk.bq_c)
            kv = ops.gemm(xa, blk.wkv_c, bias=blk.bkv_c)
            a = ops.attention_ex(q, kv[:, :d], kv[:, d:], B, n, n_audio, H, causal=False)
            ops.gemm(a, blk.wo_c, x, bias=blk.bo_c, residual=x, out_f32=True)
            ops.layernorm(x, blk.ln2_g, blk.ln2_b, dt, out=h)
            u = ops.gemm(h, blk.w1, bias=blk.b1, gelu=True)
            ops.gemm(u, blk.w2, x, bias=blk.b2, residual=x, out_f32=True)
        ops.layernorm(x, dw.ln_g, dw.ln_b, dt, out=h)
        logits = ops.gemm(h, dw.tok_emb, out_f32=True)
        return logits.view(B, n, dw.tok_emb.shape[0])

    # ---- autoregressive decoding with a key / value cache --------------------------------------
    def _token_step(self, tok: torch.Tensor, t: int, caches, kv_cross, n_audio_clips: int, rows_per_clip: int, n_max: int,
                    n_audio: int) -> torch.Tensor:
        """One decoder position for N = n_audio_clips * rows_per_clip sequences (clip-major): tok int64 [N] at position t ->
        the residual stream x [N, d] f32 after all blocks.  Self-attention keys / values of this position are appended to
        `caches`; cross-attention treats a clip's rows_per_clip sequences as the query rows of one clip (no mask), so the
        audio keys / values are projected and stored once per clip whatever the beam width."""
        dw = self.dec
        d, dt, H = dw.d, dw.dtype, dw.n_head
        N = tok.shape[0]
        x = ops.embed_tokens(tok.view(N, 1).contiguous(), dw.tok_emb_f32, dw.pos[t:])                    # [N, d] f32
        h = torch.empty((N, d), dtype=dt, device=self.device)
        for blk, kvc, kc in zip(dw.blocks, kv_cross, caches):
            ops.layernorm(x, blk.ln1_g, blk.ln1_b, dt, out=h)
            qkv = ops.gemm(h, blk.wqkv, bias=blk.bqkv)                                                  # [N, 3d]
            kc.view(N, n_max, 2 * d)[:, t] = qkv[:, d:]                                                 # cache write (copy)
            a = ops.attention_cached(qkv[:, :d], kc[:, :d], kc[:, d:], N, 1, t + 1, H, q_batch_rows=1, kv_batch_rows=n_max)
            ops.gemm(a, blk.wo, x, bias=blk.bo, residual=x, out_f32=True)
            ops.layernorm(x, blk.lnc_g, blk.lnc_b, dt, out=h)
            q = ops.gemm(h, blk.wq_c, bias=blk.bq_c)
            a = ops.attention_ex(q, kvc[:, :d], kvc[:, d:], n_audio_clips, rows_per_clip, n_audio, H, causal=False)
            ops.gemm(a, blk.wo_c, x, bias=blk.bo_c, residual=x, out_f32=True)
            ops.layernorm(x, blk.ln2_g, blk.ln2_b, dt, out=h)
            u = ops.gemm(h, blk.w1, bias=blk.b1, gelu=True)
            ops.gemm(u, blk.w2, x, bias=blk.b2, residual=x, out_f32=True)
        return x

    def _last_logits(self, x: torch.Tensor) -> torch.Tensor:
        dw = self.dec
        h = ops.layernorm(x, dw.ln_g, dw.ln_b, dw.dtype)
        return ops.gemm(h, dw.tok_emb, out_f32=True)                                                    # [N, V] f32

    def _decode_setup(self, B: int, n0: int, max_new_tokens: int, xa: torch.Tensor, rows_per_clip: int):
        dw = self.dec
        if dw is None:
            raise _lib.LyricAlignHipError("this engine was packed without decoder weights")
        n_max = n0 + max_new_tokens
        if n_max > dw.pos.shape[0]:
            raise ValueError(f"decoding: {n_max} positions exceed the decoder's context ({dw.pos.shape[0]})")
        kv_cross = [ops.gemm(xa, blk.wkv_c, bias=blk.bkv_c) for blk in dw.blocks]                        # [B*n_audio, 2d], once
        caches = [torch.empty((B * rows_per_clip * n_max, 2 * dw.d), dtype=dw.dtype, device=self.device) for _ in dw.blocks]
        return n_max, kv_cross, caches

    @torch.no_grad()
    def decode_greedy(self, prompt: torch.Tensor, xa: torch.Tensor, max_new_tokens: int, eot: int, n_audio: int = N_CTX) -> torch.Tensor:
        """Greedy decoding of the text decoder (the token loop under whisper's transcribe / DecodingTask with
        temperature 0 and no beam: argmax of Whisper.logits at the last position, fed back, until every sequence has
        produced `eot` or max_new_tokens are out).  prompt int64 [B, n0] (sot / language / task tokens), xa = encoder output
        rows [B*n_audio, d] -> tokens int64 [B, n0 + steps]; finished sequences are padded with `eot`.
        Cross-attention keys / values are projected once per layer; self-attention keys / values go into a per-layer
        cache, so a step costs one token's worth of projections plus attention over the cache.  No token suppression,
        timestamp rules or temperature fallback (those live in whisper/decoding.py, outside this path)."""
        B, n0 = prompt.shape
        n_max, kv_cross, caches = self._decode_setup(B, n0, max_new_tokens, xa, 1)
        dev = self.device
        tokens = torch.full((B, n_max), int(eot), dtype=torch.int64, device=dev)
        tokens[:, :n0] = prompt.to(device=dev, dtype=torch.int64)
        finished = torch.zeros((B,), dtype=torch.bool, device=dev)
        n_out = n0
        for t in range(n_max - 1):
            x = self._token_step(tokens[:, t].contiguous(), t, caches, kv_cross, B, 1, n_max, n_audio)
            if t + 1 < n0:
                continue                                                   # still consuming the prompt: nothing to choose
            nxt = ops.argmax_rows(self._last_logits(x))                    # [B]
            tokens[:, t + 1] = torch.where(finished, tokens[:, t + 1], nxt)   # finished rows keep their eot padding
            finished |= nxt == eot
            n_out = t + 2
            if bool(finished.all()):                                       # one host sync per step: the loop is data dependent
                break
        return tokens[:, :n_out]

    @torch.no_grad()
    def decode_beam(self, prompt: torch.Tensor, xa: torch.Tensor, beam_size: int, max_new_tokens: int, eot: int,
                    patience: float = 1.0, n_audio: int = N_CTX):
        """Beam search over the text decoder (what inference_transcript.py:88-91 asks whisper's transcribe for with
        beam_size=5): per step the beam_size + 1 best continuations of every live sequence (la_topk_rows_f32 on the device:
        values and the row log-sum-exp), candidates ranked per clip by summed log-probability, sequences ending in `eot`
        moved to the clip's finished set (at most round(beam_size * patience) kept), the key / value caches re-gathered to
        follow their source sequences, until every clip has its finished set or max_new_tokens are out; the returned
        sequence per clip is the finished one with the highest summed log-probability per generated token.
        Follows the published algorithm of whisper/decoding.py (BeamSearchDecoder + MaximumLikelihoodRanker without
        length penalty); openai-whisper is not in this image, so this is NOT pinned to it -- the tests pin the device
        path (cache gather, top-k, log-sum-exp) to a plain restatement run on the oracle's decoder.
        prompt int64 [B, n0]; xa [B*n_audio, d] -> (list of B int64 tensors (prompt + generated, without eot),
        list of B summed log-probabilities)."""
        B, n0 = prompt.shape
        beam = int(beam_size)
        if not 1 <= beam <= 7:
            raise ValueError("decode_beam: beam_size must be in 1..7")
        n_max, kv_cross, caches = self._decode_setup(B, n0, max_new_tokens, xa, beam)
        dev = self.device
        N = B * beam
        seqs = [[int(v) for v in prompt[i].tolist()] for i in range(B) for _ in range(beam)]     # clip-major, beam-minor
        sum_lp = [0.0] * N
        finished = [dict() for _ in range(B)]
        max_cand = max(1, round(beam * patience))
        cur = torch.tensor([s_[0] for s_ in seqs], dtype=torch.int64, device=dev)
        for t in range(n_max - 1):
            x = self._token_step(cur, t, caches, kv_cross, B, beam, n_max, n_audio)
            if t + 1 < n0:
                cur = torch.tensor([s_[t + 1] for s_ in seqs], dtype=torch.int64, device=dev)
                continue
            vals, idx, lse = ops.topk_rows(self._last_logits(x), beam + 1)
            lp = (vals - lse[:, None]).cpu().tolist()                                             # [N][beam+1] log-probabilities
            ix = idx.cpu().tolist()
            new_seqs, new_lp, src = [], [], []
            for i in range(B):
                scores, sources = {}, {}
                for j in range(beam):
                    r = i * beam + j
                    for c in range(beam + 1):
                        key = tuple(seqs[r] + [ix[r][c]])
                        scores[key] = sum_lp[r] + lp[r][c]
                        sources[key] = r
                saved = 0
                for key in sorted(scores, key=scores.get, reverse=True):
                    if key[-1] == eot:
                        finished[i][key] = scores[key]
                    else:
                        new_seqs.append(list(key)); new_lp.append(scores[key]); src.append(sources[key])
                        saved += 1
                        if saved == beam:
                            break
                while saved < beam:                                  # fewer live continuations than beams: repeat the best
                    new_seqs.append(list(new_seqs[-1]) if saved else seqs[i * beam] + [eot])
                    new_lp.append(new_lp[-1] if saved else -float("inf")); src.append(src[-1] if saved else i * beam)
                    saved += 1
                if len(finished[i]) > max_cand:                      # keep the best max_cand finished sequences
                    keep = sorted(finished[i], key=finished[i].get, reverse=True)[:max_cand]
                    finished[i] = {k_: finished[i][k_] for k_ in keep}
            seqs, sum_lp = new_seqs, new_lp
            src_dev = torch.tensor(src, dtype=torch.int64, device=dev)
            for li in range(len(caches)):                            # caches follow their source sequences (gather = data movement)
                kc = caches[li].view(N, n_max, -1)
                caches[li] = kc.index_select(0, src_dev).view(N * n_max, -1)
            if all(len(f) >= max_cand for f in finished) or t + 2 >= n_max:
                break
            cur = torch.tensor([s_[t + 1] for s_ in seqs], dtype=torch.int64, device=dev)
        out_tokens, out_lp = [], []
        for i in range(B):
            cands = dict(finished[i])
            if len(cands) < beam:                                    # not enough finished: the live beams count as ended here
                order = sorted(range(beam), key=lambda j: sum_lp[i * beam + j], reverse=True)
                for j in order:
                    if len(cands) >= beam:
                        break
                    cands[tuple(seqs[i * beam + j] + [eot])] = sum_lp[i * beam + j]
            best = max(cands, key=lambda k_: cands[k_] / max(1, len(k_) - n0 - 1))
            out_tokens.append(torch.tensor(list(best[:-1]), dtype=torch.int64))
            out_lp.append(float(cands[best]))
        return out_tokens, out_lp

    # ---- head: align_rnn up to Mish ------------------------------------------------------
    def head_clip_cap(self, T: int) -> int:
        """Clips one launch set of the persistent GRU recurrence can take: every workgroup of the launch must be resident
        (la_gru.hip: 2 directions x hidden/(16 * waves) workgroups per 16 clips, at most 224 per launch -> 288 clips with 4-wave
        workgroups, 144 in float32 at hidden 384), the out buffer is addressed through a 2 GiB buffer descriptor, and both
        this sequence and la_align_head_forward clamp to HEAD_CLIPS_MAX = 256."""
        hw = self.head
        if hw.hidden < 64 or hw.hidden % 64 != 0:
            raise NotImplementedError(f"liblyricalign_hip unsupported shape: GRU hidden={hw.hidden} (a multiple of 64 is required)")
        es = 4 if hw.dtype == torch.float32 else 2
        nsplit = hw.hidden // (16 * (2 if hw.dtype == torch.float32 else 4))
        by_cus = (224 // (2 * nsplit)) * 16
        by_desc = (2 ** 31 - 1) // (max(1, T) * 2 * hw.hidden * es)
        return max(1, min(HEAD_CLIPS_MAX, by_cus, by_desc))

    def head_hidden(self, feats: torch.Tensor, B: int, T: int, feat_clip_stride: int) -> torch.Tensor:
        """feats: rows [.., d] in compute dtype, clip b at rows b*feat_clip_stride .. +T.  -> Mish(GRU) [B*T, 2H].
        More clips than one launch set of the recurrence takes (head_clip_cap) run as consecutive slices of clips."""
        hw = self.head
        H, dt = hw.hidden, hw.dtype
        cap = self.head_clip_cap(T)
        act = self._get("head_act", (B * T, 2 * H), dt)
        for b0 in range(0, B, cap):
            b1 = min(B, b0 + cap)
            self._head_slice(feats[b0 * feat_clip_stride:], b1 - b0, T, feat_clip_stride, min(B, cap), act[b0 * T: b1 * T])
        return act

    def _head_slice(self, feats: torch.Tensor, B: int, T: int, feat_clip_stride: int, Bbuf: int, act_out: torch.Tensor) -> None:
        hw = self.head
        H, dt = hw.hidden, hw.dtype
        x = feats
        lda, stride_a = feats.stride(0), feat_clip_stride * feats.stride(0)
        n_layers = len(hw.w_ih)
        for layer in range(n_layers):
            gi = self._get("gi", (Bbuf, T, 2, 3 * H), torch.float32)[:B]
            K = hw.w_ih[layer].shape[1]
            if (hw.w_ih_x2 is not None and stride_a == T * lda and lda == K and _x2_domain(B * T, 6 * H, K) and x2_inference_on()):
                _x2_linear(x[: B * T] if x.dim() == 2 else x.view(-1, K)[: B * T], hw.w_ih_x2[layer], hw.b_ih[layer], out=gi.view(B * T, 6 * H))
            else:
                ops.gemm(x, hw.w_ih[layer], gi.view(B * T, 6 * H), bias=hw.b_ih[layer], out_f32=True, M=T, lda=lda, batch=B,
                         stride_a=stride_a, stride_c=T * 6 * H, ldc=6 * H)
            out = self._get(f"gru{layer}", (Bbuf, T, 2 * H), dt)[:B]
            last = layer == n_layers - 1
            if self._gru_flag is None:
                self._gru_flag = torch.zeros((1,), dtype=torch.int32, device=self.device)
            ops.gru_layer(gi, hw.w_hh[layer], hw.b_hh[layer], out=out, out_mish=act_out.view(B, T, 2 * H) if last else None,
                          flag=self._gru_flag)
            x = out.view(B * T, 2 * H)
            lda, stride_a = 2 * H, T * 2 * H

    def logits(self, feats: torch.Tensor, B: int, T: int, feat_clip_stride: int) -> torch.Tensor:
        """Materialised align logits [B, T, V] f32 (the reference's frame_manual_forward output)."""
        act = self.head_hidden(feats, B, T, feat_clip_stride)
        out = torch.empty((B * T, self.head.vocab), dtype=torch.float32, device=self.device)
        hw = self.head
        if hw.w_fc_x2 is not None and _x2_domain(B * T, hw.vocab, hw.w_fc.shape[1]) and x2_inference_on():
            _x2_linear(act, hw.w_fc_x2, hw.b_fc, out=out)
        else:
            ops.gemm(act, hw.w_fc, out, bias=hw.b_fc, out_f32=True)
        return out.view(B, T, hw.vocab)

    def emissions(self, feats: torch.Tensor, B: int, T: int, feat_clip_stride: int, labels: torch.Tensor,
                  n_labels: torch.Tensor, variant: int) -> torch.Tensor:
        act = self.head_hidden(feats, B, T, feat_clip_stride)
        return ops.fc_emissions(act, self.head.w_fc, self.head.b_fc, B, T, labels, n_labels, variant,
                                w_x2=self.head.w_fc_x2 if (self.head.w_fc_x2 is not None and x2_inference_on()) else None)

    def align_feats(self, feats: torch.Tensor, B: int, T: int, feat_clip_stride: int, labels: torch.Tensor, n_labels: torch.Tensor,
                    variant: int, flag: Optional[torch.Tensor] = None):
        """Encoder rows -> (onset, offset, final_score, status): head + emission prep + DP (one C call unless LA_ENGINE_PY=1).
        The op-by-op Python sequence below is also taken when HEAD_CLIPS_MAX was lowered (a test knob of that sequence; the C
        call slices by its own cap, LA_HEAD_CLIP_CAP) and for heads the C struct does not describe (not 2 GRU layers)."""
        if self._gru_flag is None:
            self._gru_flag = torch.zeros((1,), dtype=torch.int32, device=self.device)
        if ENGINE_PY or self._head_c is None or HEAD_CLIPS_MAX != 256:
            own, self._gru_flag = self._gru_flag, (flag if flag is not None else self._gru_flag)
            try:
                em = self.emissions(feats, B, T, feat_clip_stride, labels, n_labels, variant)
            finally:
                self._gru_flag = own
            nf = torch.full((B,), T, dtype=torch.int32, device=self.device)
            return ops.viterbi_batch(em, labels, n_labels, nf)
        return ops.align_head_forward(self._head_c, feats, feat_clip_stride, B, T, labels, n_labels, variant,
                                      flag if flag is not None else self._gru_flag, ws_cache=self._ws)

    def align_feats_checked(self, feats: torch.Tensor, B: int, T: int, feat_clip_stride: int, labels: torch.Tensor, n_labels: torch.Tensor,
                            variant: int):
        """align_feats with the persistent GRU's time-out handled (synchronises): the recurrence needs every workgroup of its launch set
        co-resident, which HIP does not promise while other streams own the CUs; a launch that waited out its bound (option
        gru_timeout_us, 3 s) leaves garbage.  The device is then idle -- the head is re-enqueued ONCE, alone, as a fresh launch -- and only a
        second time-out raises TimeoutError.  self.gru_recoveries counts the re-launches."""
        if self._gru_flag is None:
            self._gru_flag = torch.zeros((1,), dtype=torch.int32, device=self.device)
        res = self.align_feats(feats, B, T, feat_clip_stride, labels, n_labels, variant)
        if int(self._gru_flag.item()) != 0:
            self._gru_flag.zero_()
            torch.cuda.synchronize(self.device)
            self.gru_recoveries = getattr(self, "gru_recoveries", 0) + 1
            res = self.align_feats(feats, B, T, feat_clip_stride, labels, n_labels, variant)
            self.check_gru()
        return res

    def check_gru(self) -> None:
        """Host check of the persistent GRU kernel's bounded waits (synchronises): every launch since the last check."""
        if self._gru_flag is not None and int(self._gru_flag.item()) != 0:
            self._gru_flag.zero_()
            raise TimeoutError("persistent GRU kernel: a bounded inter-workgroup wait timed out")

    # ---- whole path on a ready mel batch ---------------------------------------------------
    def align_mel(self, mel: torch.Tensor, labels: torch.Tensor, n_labels: torch.Tensor, n_frames: int = N_CTX,
                  use_ctc: bool = True):
        """mel [B,80,3000] -> (onset, offset, final_score, status) device tensors; frames = first n_frames of each clip."""
        B = mel.shape[0]
        feats = self.encode(mel)
        variant = LA_VARIANT_CTC if use_ctc else LA_VARIANT_PLAIN
        return self.align_feats(feats, B, n_frames, N_CTX, labels, n_labels, variant)


class PipelinedAligner:
    """Two-stream software pipeline over consecutive batches: the encoders of the next batches (stream E, GEMM / attention
    bound, fill the chip) overlap the head of the previous ones (stream H: the persistent GRU recurrence occupies
    2 x hidden/64 CUs per 32 clips and is latency-bound, ~13 ms per layer whatever the batch; then the fused FC and the DP).

    head_group: the head runs once per `head_group` submitted batches, over all their clips at once (clips are
    independent: more clips are more workgroup groups of the GRU kernel, not more steps).  While the recurrence is resident
    the encoder's GEMMs lose CUs and with them whole rounds of 256x256 tiles (752 tiles: 3 rounds on 256 CUs, 4 on 244);
    running it half / a quarter as often takes that cost off most batches.  Results of a batch arrive when its group is
    flushed (drain() flushes a partial group).

    The only shared buffers are the encoder outputs of a group, double-buffered and handed over with events; every other
    scratch buffer belongs to exactly one stream.  This changes no result -- only which kernels are in flight together."""

    def __init__(self, engine: AlignEngine, head_group: int = 2, encoder_streams: int = 1):
        if head_group < 1:
            raise ValueError("head_group must be >= 1")
        self.eng = engine
        self.G = int(head_group)
        dev = engine.device
        # encoder_streams = 2: consecutive batches' encoders alternate between two streams (each with its own scratch
        # buffers, same packed weights), so one batch's kernel tails and launch gaps are filled by the other's kernels
        self.n_enc = int(encoder_streams)
        self._enc_engines = [engine] + [AlignEngine(engine.enc, engine.head, dev, dec=engine.dec) for _ in range(self.n_enc - 1)]
        self._enc_streams = [torch.cuda.Stream(device=dev) for _ in range(self.n_enc)]
        self._enc_events = [torch.cuda.Event() for _ in range(self.n_enc)]
        self._enc_i = 0
        self.stream_e = self._enc_streams[0]
        # the GRU's few workgroups should dispatch promptly (LA_HEAD_PRIO: developer A/B of the head stream's priority, -1 | 0)
        self.stream_h = torch.cuda.Stream(device=dev, priority=int(os.environ.get("LA_HEAD_PRIO", "-1")))
        self.enc_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.head_done = [torch.cuda.Event(), torch.cuda.Event()]
        self._head_used = [False, False]
        self.gi = 0                 # group counter (parity = buffer set)
        self._pending: List[dict] = []
        self._pending_events: List[torch.cuda.Event] = []   # one per submitted batch of the open group (its encoder is done)
        self._key = None
        self._feats = [None, None]  # per buffer set: [G*B*1500, d] encoder outputs
        # Time-out of the persistent recurrence (co-residency of its workgroups is not guaranteed while the encoders own the CUs): every group's
        # head gets a flag word of its own, copied to pinned host memory behind the head; a group is VERIFIED (lazily in submit(), at the
        # latest in drain()) once its head is done and the word is zero.  Until then its inputs are kept, and a group whose word is set is
        # recomputed -- encoder and head, alone on an idle device, a fresh launch -- into the result tensors it already handed out.
        self._unverified: List[dict] = []
        self._host_flags = torch.zeros((256,), dtype=torch.int32).pin_memory()      # one word per group in flight (ring)
        self.recovered_groups = 0

    def submit(self, mel: torch.Tensor, labels: torch.Tensor, n_labels: torch.Tensor, n_frames: int = N_CTX,
               use_ctc: bool = True, host_out=None):
        """Enqueue one batch of <= 30 s clips (mel [B,80,3000]); returns (onset, offset, score, status) device tensors that
        are valid after drain() (or once the batch's group has been flushed and its head_done event has passed).
        host_out: optional (onset, offset, status) pinned host tensors for an async D2H."""
        return self._submit(mel, mel.shape[0], int(n_frames), N_CTX, labels, n_labels, use_ctc, host_out)

    def submit_songs(self, mel: torch.Tensor, labels: torch.Tensor, n_labels: torch.Tensor, get_orig_len: bool = True,
                     use_ctc: bool = True, host_out=None):
        """Long form (BASELINE configs[4]; reference chunking module/align_model.py:94-105): mel [S,80,n] of whole songs.
        The songs' 30 s chunks go through the encoder as ONE song-major batch on the encoder stream; the recurrence over
        the T = n/2 frames of each song, the fused FC and the DP run on the head stream under the next batch's encoder
        (T sequential GRU steps per layer whatever S is: songs are the GRU's batch rows / workgroup groups)."""
        from .module.align_model import frame_plan, song_major_chunks      # host bookkeeping shared with AlignModel
        from .whisper_compat import pad_or_trim
        plan = frame_plan(mel.shape[-1], get_orig_len)
        S = mel.shape[0]
        if len(plan) == 1:
            return self._submit(pad_or_trim(mel, N_FRAMES), S, plan[0][2], N_CTX, labels, n_labels, use_ctc, host_out)
        return self._submit(song_major_chunks(mel, plan), S, sum(k for _, _, k in plan), len(plan) * N_CTX, labels, n_labels,
                            use_ctc, host_out)

    def _submit(self, mel: torch.Tensor, B: int, n_frames: int, clip_rows: int, labels, n_labels, use_ctc: bool, host_out):
        """mel: [B * clip_rows / 1500, 80, 3000] encoder clips, clip-major; clip b owns encoder rows b*clip_rows .. +n_frames."""
        eng = self.eng
        key = (B, int(labels.shape[1]), int(n_frames), bool(use_ctc), int(clip_rows))
        if self._pending and key != self._key:
            self._flush()                                            # a group holds batches of one shape
        self._key = key
        slot = self.gi & 1
        j = len(self._pending)
        d, dt = eng.enc.d, eng.enc.dtype
        cur = torch.cuda.current_stream(eng.device)
        k = self._enc_i % self.n_enc
        self._enc_i += 1
        st, enc_eng = self._enc_streams[k], self._enc_engines[k]
        st.wait_stream(cur)
        rows = B * clip_rows
        with torch.cuda.stream(st):
            if self._head_used[slot]:
                st.wait_event(self.head_done[slot])                  # the head of group g-2 has consumed this buffer set
            fb = self._feats[slot]
            if fb is None or fb.shape[0] != self.G * rows or fb.dtype != dt:
                fb = self._feats[slot] = torch.empty((self.G * rows, d), dtype=dt, device=eng.device)
                fb.record_stream(self.stream_h)
                for s2 in self._enc_streams:
                    fb.record_stream(s2)
            mel.record_stream(st)
            enc_eng.encode(mel, out=fb[j * rows:(j + 1) * rows])
            ev = torch.cuda.Event()
            ev.record(st)
        self._pending_events.append(ev)
        Lmax = labels.shape[1]
        out = (torch.empty((B, Lmax), dtype=torch.int32, device=eng.device), torch.empty((B, Lmax), dtype=torch.int32, device=eng.device),
               torch.empty((B,), dtype=torch.float64, device=eng.device), torch.empty((B,), dtype=torch.int32, device=eng.device))
        for t_ in out:
            t_.record_stream(self.stream_h)          # written on stream H: the allocator must not recycle them under it
        # read on stream H when the group is flushed, possibly long after the caller dropped its references: the caching
        # allocator must not hand these blocks back to the caller's stream before the head has consumed them
        labels.record_stream(self.stream_h)
        n_labels.record_stream(self.stream_h)
        self._pending.append(dict(labels=labels, n_labels=n_labels, out=out, host_out=host_out, mel=mel))
        if len(self._pending) == self.G:
            self._flush()
        self._verify(block=False)
        return out

    def _flush(self):
        if not self._pending:
            return
        eng, slot = self.eng, self.gi & 1
        B, _, n_frames, use_ctc, clip_rows = self._key
        n = len(self._pending)
        with torch.cuda.stream(self.stream_h):
            for ev in self._pending_events:                          # every encoder of the group, whichever stream ran it
                self.stream_h.wait_event(ev)
            self._pending_events = []
            labels = self._pending[0]["labels"] if n == 1 else torch.cat([q["labels"] for q in self._pending], dim=0)
            n_labels = self._pending[0]["n_labels"] if n == 1 else torch.cat([q["n_labels"] for q in self._pending], dim=0)
            variant = LA_VARIANT_CTC if use_ctc else LA_VARIANT_PLAIN
            feats = self._feats[slot][: n * B * clip_rows]
            flag = torch.zeros((1,), dtype=torch.int32, device=eng.device)
            flag.record_stream(self.stream_h)
            res = eng.align_feats(feats, n * B, n_frames, clip_rows, labels, n_labels, variant, flag=flag)
            for j, q in enumerate(self._pending):
                for dst, src in zip(q["out"], res):
                    dst.copy_(src[j * B:(j + 1) * B], non_blocking=True)
                if q["host_out"] is not None:
                    q["host_out"][0].copy_(q["out"][0], non_blocking=True)
                    q["host_out"][1].copy_(q["out"][1], non_blocking=True)
                    q["host_out"][2].copy_(q["out"][3], non_blocking=True)
            if len(self._unverified) >= self._host_flags.numel() - 1:
                self._verify(block=True)                                # (never in practice: submit() verifies finished groups as it goes)
            host_flag = self._host_flags[self.gi % self._host_flags.numel(): self.gi % self._host_flags.numel() + 1]
            host_flag.copy_(flag, non_blocking=True)
            self.head_done[slot].record(self.stream_h)
            done = torch.cuda.Event()
            done.record(self.stream_h)
            self._head_used[slot] = True
        self._unverified.append(dict(batches=self._pending, key=self._key, host_flag=host_flag, done=done))
        self._pending = []
        self.gi += 1

    def _verify(self, block: bool) -> None:
        """Groups whose head has finished: time-out word zero -> their inputs are released; set -> _recover.  block: wait for every group."""
        while self._unverified:
            g = self._unverified[0]
            if block:
                g["done"].synchronize()
            elif not g["done"].query():
                return
            self._unverified.pop(0)
            if int(g["host_flag"][0]) != 0:
                self._recover(g)

    def _recover(self, g: dict) -> None:
        """One re-run of a group whose recurrence timed out: everything in flight is waited for (the device is idle, so the launch set
        is co-resident), then each of its batches goes through encoder and head again on the current stream, into the result tensors
        (and pinned host copies) it had handed out.  A second time-out raises TimeoutError."""
        eng = self.eng
        torch.cuda.synchronize(eng.device)
        B, _, n_frames, use_ctc, clip_rows = g["key"]
        variant = LA_VARIANT_CTC if use_ctc else LA_VARIANT_PLAIN
        flag = torch.zeros((1,), dtype=torch.int32, device=eng.device)
        for q in g["batches"]:
            feats = eng.encode(q["mel"])
            res = eng.align_feats(feats, B, n_frames, clip_rows, q["labels"], q["n_labels"], variant, flag=flag)
            for dst, src in zip(q["out"], res):
                dst.copy_(src)
            if q["host_out"] is not None:
                q["host_out"][0].copy_(q["out"][0]); q["host_out"][1].copy_(q["out"][1]); q["host_out"][2].copy_(q["out"][3])
        torch.cuda.synchronize(eng.device)
        self.recovered_groups += 1
        if int(flag.item()) != 0:
            raise TimeoutError("persistent GRU kernel: a bounded inter-workgroup wait timed out again on the re-run of its group (alone on the device)")

    def drain(self):
        self._flush()
        for st in self._enc_streams:
            st.synchronize()
        self.stream_h.synchronize()
        self._verify(block=True)
        self.eng.check_gru()
