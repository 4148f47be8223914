"""float32 Linear products of the fine-tune step on the f16 matrix pipe at float32 accuracy ("f16x2"; csrc/la_f32x2.hip).

The reference trains in float32 (train_multitask.py:325-326: loss.backward() through nn.Linear); gfx950 multiplies float32
operands at 1/16 of its 16-bit MFMA rate.  A float32 matrix whose rows are scaled by powers of two splits exactly into two
IEEE-half planes (hi + lo = 22 bits), and one pass of the 256 x 256 f16 kernel (three products per 32-wide k chunk) accumulates
a_lo w_hi + a_hi w_lo + a_hi w_hi in float32 -- 3/16 of the float32 pipe's cost, error against a float64 product smaller than
the float32 kernel's own (profiles/r5_kbench_f32emu.txt).  This module is the host side: which products take that path (the
large ones: >= 192 tiles of 256 x 256, counting split-K slots), their operand splits, and the three product shapes of a Linear

    linear(x, w, bias, residual)   y  = x w^T (+ bias) (+ residual)          forward
    gemm_nn(dy, w)                 dx = dy w                                  input gradient
    gemm_tn(dy, x)                 dw = dy^T x                                weight gradient (contraction over the rows)

with the float32 MFMA kernel (ops.gemm / head_train.gemm_nn / gemm_tn) for every shape outside that domain.
LA_F32X2=0 keeps every product on the float32 kernel (the A/B partner; read at import).  The MLP's gelu(u) operand is split
straight from u (x_act / act = "gelu": the activation runs inside the split kernels, no float32 buffer of gelu(u) in the forward
or in the weight gradient) and gemm_nn(..., gelu_grad_of=u) multiplies the input gradient by gelu'(u) in the product's epilogue;
LA_F32X2_ACT=0 goes through the buffers (same bits).  An operand that is split both ways leaves its largest magnitude with the plain
split for the transposed one (OperandMax: no pass for column maxima; LA_F32X2_TMAX=0 keeps the pass), and gemm_tn(..., colsum=db) returns
the bias gradient from the transposed split's own tiles.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

ENABLED = os.environ.get("LA_F32X2", "1") != "0"
FUSE_ACT = os.environ.get("LA_F32X2_ACT", "1") != "0"     # 0: gelu(x) through its own float32 buffer before the split (A/B partner)
MIN_TILES = 192          # the 256 x 256 kernel's domain (la_gemm_f16x2)


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class Planes:
    """Split operand: planes [rows, 2, kp] float16 (hi plane, lo plane), inv_scale [rows] float32; k = the unpadded length."""
    planes: torch.Tensor
    inv_scale: torch.Tensor
    rows: int
    k: int
    kp: int


ACT = {None: 0, "gelu": 1}
# LA_F32X2_TMAX=0: a transposed split always makes its own pass for column maxima (A/B partner; read at import)
REUSE_MAX = os.environ.get("LA_F32X2_TMAX", "1") != "0"


class OperandMax:
    """The largest magnitude of an operand, left by its plain split (la_split_f16x2_max: 64 words, atomicMax) for a later TRANSPOSED split
    of the same operand (la_split_f16x2_t_tmax), which then takes one power-of-two scale for the whole operand and skips its pass for
    column maxima.  dy is split both ways in one backward step (dx = dy w, dw = dy^T x); an activation is split plain in the forward and
    transposed for its weight gradient."""

    WORDS = 64 * 32                          # 64 words in use, one per 128-byte line

    def __init__(self, device, words: Optional[torch.Tensor] = None):
        # `words`: a zeroed slice of a buffer the caller allocated for many operands at once (OperandMax.many)
        self.words = words if words is not None else torch.zeros((self.WORDS,), dtype=torch.int32, device=device)
        self.act: Optional[str] = None
        self.valid = False

    @classmethod
    def many(cls, device, count: int):
        """`count` of them over one zeroed buffer (one fill instead of `count`)."""
        buf = torch.zeros((count, cls.WORDS), dtype=torch.int32, device=device)
        return [cls(device, buf[i]) for i in range(count)]


def split(x: torch.Tensor, kp: Optional[int] = None, act: Optional[str] = None, omax: Optional[OperandMax] = None) -> Planes:
    """x [rows, k] float32 (row view, unit inner stride) -> its planes along k, zero-padded to kp (default: k rounded up to 128).
    act = "gelu": the planes of gelu(x) (exact erf), applied inside the split.  omax: receives the operand's largest magnitude."""
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("f32x2.split: a float32 [rows, k] row view is expected")
    rows, k = x.shape
    kp = _rup(k, 128) if kp is None else kp
    planes = torch.empty((rows, 2, kp), dtype=torch.float16, device=x.device)
    inv = torch.empty((rows,), dtype=torch.float32, device=x.device)
    if omax is not None and REUSE_MAX:
        check(lib().la_split_f16x2_max(ptr(x), x.stride(0), rows, k, ptr(planes), kp, ptr(inv), ACT[act], ptr(omax.words), stream_ptr()), "split_f16x2")
        omax.act, omax.valid = act, True
    else:
        check(lib().la_split_f16x2_act(ptr(x), x.stride(0), rows, k, ptr(planes), kp, ptr(inv), ACT[act], stream_ptr()), "split_f16x2")
    return Planes(planes, inv, rows, k, kp)


def layernorm_split(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, kp: Optional[int] = None) -> Planes:
    """LayerNorm(x) gamma + beta (eps 1e-5) straight into the planes of the next Linear's operand (la_layernorm_f16x2: the float32 rows of the
    normalised activations are never written).  x [rows, d] float32 row view, d % 4 == 0, d <= 4096."""
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("f32x2.layernorm_split: a float32 [rows, d] row view is expected")
    rows, d = x.shape
    kp = d if kp is None else kp
    planes = torch.empty((rows, 2, kp), dtype=torch.float16, device=x.device)
    inv = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(lib().la_layernorm_f16x2(ptr(x), x.stride(0), rows, d, ptr(gamma.contiguous()), ptr(beta.contiguous()), ptr(planes), kp, ptr(inv), stream_ptr()),
          "layernorm_f16x2")
    return Planes(planes, inv, rows, d, kp)


def split_t(x: torch.Tensor, mp: Optional[int] = None, act: Optional[str] = None, colsum: Optional[torch.Tensor] = None,
            omax: Optional[OperandMax] = None) -> Planes:
    """x [m, k] float32 -> the planes of x^T (of act(x)^T): rows = k, contraction length m zero-padded to mp (default: m rounded up to 128).
    colsum [k] float32 (act None): also filled with the column sums of x, from the pass that finds the column maxima.
    omax (valid, same act): the operand's maximum from its plain split -- one scale for the whole operand, no pass for column maxima."""
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("f32x2.split_t: a float32 [m, k] row view is expected")
    m, k = x.shape
    mp = _rup(m, 128) if mp is None else mp
    planes = torch.empty((k, 2, mp), dtype=torch.float16, device=x.device)
    inv = torch.empty((k,), dtype=torch.float32, device=x.device)
    words = omax.words if (omax is not None and omax.valid and omax.act == act and REUSE_MAX) else None
    check(lib().la_split_f16x2_t_tmax(ptr(x), x.stride(0), m, k, ptr(planes), mp, ptr(inv), ACT[act], ptr(colsum), ptr(words), stream_ptr()),
          "split_f16x2_t")
    return Planes(planes, inv, k, m, mp)


class WeightPlanes:
    """The split planes of ONE weight matrix, kept for as long as the weight is unchanged: the reference's accumulation loop runs eight
    micro-steps on the same weights (train_multitask.py:240-326), each of which would split every weight twice (plain for the forward,
    transposed for dx = dy w).  {("n" | "t", plane length): Planes}; the owner drops it when the weight's version moves on."""

    def __init__(self):
        self.planes = {}

    def put(self, orient: str, length: int, value: "Planes") -> None:
        """Keep the most recent plane length per orientation (a different batch size pads the contraction differently; the planes of the
        previous one are not kept beside the new ones)."""
        for k in [k for k in self.planes if k[0] == orient]:
            del self.planes[k]
        self.planes[(orient, length)] = value


def slots_for(M: int, N: int) -> int:
    """Split-K slots that bring a product with few 256 x 256 tiles up to the kernel's domain (1 = none needed)."""
    tiles = -(-M // 256) * -(-N // 256)
    return max(1, -(-MIN_TILES // tiles))


def eligible(M: int, N: int, K: int) -> bool:
    """Does C [M, N] = A [M, K] W [N, K]^T take the f16x2 path?  Large products only: N > 128, at least 256 of K per split-K slot and
    no more than 16 slots (beyond that the partial sums cost more than the float32 kernel)."""
    if not ENABLED or N <= 128 or K < 256:
        return False
    s = slots_for(M, N)
    return s <= 16 and K // s >= 256


def padded_k(M: int, N: int, K: int) -> int:
    """Plane length both operands of C [M, N] are split to: K rounded up so that every split-K slot is a multiple of 128."""
    s = slots_for(M, N)
    return s * _rup(-(-K // s), 128)


def gemm(a: Planes, w: Planes, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
         residual: Optional[torch.Tensor] = None, gelu: bool = False, gelu_grad_of: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out [a.rows, w.rows] f32 = epi(A W^T) from split operands of the same plane length.  gelu_grad_of = u [a.rows, w.rows]: the
    product is multiplied by gelu'(u) in the epilogue (LA_EPI_RES_GELU_GRAD; u takes the residual operand's place)."""
    if a.kp != w.kp:
        raise ValueError(f"f32x2.gemm: operands were split to different plane lengths ({a.kp}, {w.kp})")
    M, N = a.rows, w.rows
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.planes.device)
    epi = 0
    if bias is not None:
        epi |= _lib.EPI_BIAS
    if gelu_grad_of is not None:
        if residual is not None:
            raise ValueError("f32x2.gemm: gelu_grad_of takes the residual operand's place")
        residual = gelu_grad_of
        epi |= _lib.EPI_RES_GELU_GRAD
    if residual is not None:
        epi |= _lib.EPI_RESIDUAL
    if gelu:
        epi |= _lib.EPI_GELU
    check(lib().la_gemm_f16x2(M, N, a.kp, slots_for(M, N), ptr(a.planes), ptr(a.inv_scale), ptr(w.planes), ptr(w.inv_scale), ptr(out),
                              out.stride(0), ptr(bias), ptr(residual), residual.stride(0) if residual is not None else 0, epi,
                              stream_ptr()), "gemm_f16x2")
    return out


def _plain2d(t: torch.Tensor) -> bool:
    return t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1


def _apply(x: torch.Tensor, act: Optional[str]) -> torch.Tensor:
    if act is None:
        return x
    from .encoder_train import gelu
    return gelu(x)


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
           x_act: Optional[str] = None, x_max: Optional[OperandMax] = None, w_max: Optional[OperandMax] = None,
           w_cache: Optional[WeightPlanes] = None) -> torch.Tensor:
    """y = act(x) w^T (+ bias) (+ residual), float32 in and out (F.linear; whisper/model.py Linear; x_act = "gelu": the MLP's second
    Linear on gelu(x), the activation applied inside the operand split)."""
    from . import ops
    M, K = x.shape
    N = w.shape[0]
    if _plain2d(x) and _plain2d(w) and eligible(M, N, K) and (residual is None or _plain2d(residual)):
        kp = padded_k(M, N, K)
        if not FUSE_ACT:
            x, x_act = _apply(x, x_act), None
        wp = w_cache.planes.get(("n", kp)) if w_cache is not None else None
        if wp is None:
            wp = split(w, kp, omax=w_max)
            if w_cache is not None:
                w_cache.put("n", kp, wp)
        return gemm(split(x, kp, act=x_act, omax=x_max), wp, bias=bias, residual=residual)
    return ops.gemm(_apply(x, x_act), w, bias=bias, residual=residual)


def gemm_nn(dy: torch.Tensor, w: torch.Tensor, gelu_grad_of: Optional[torch.Tensor] = None, dy_max: Optional[OperandMax] = None,
            w_max: Optional[OperandMax] = None, w_cache: Optional[WeightPlanes] = None) -> torch.Tensor:
    """dy [M, N] . w [N, K] -> [M, K]: the input gradient of y = x w^T.  gelu_grad_of = u [M, K]: times gelu'(u) -- the gradient at the
    pre-activation u of x = gelu(u), in the product's epilogue on the f16x2 path (la_gelu_bwd_f32 on the result otherwise)."""
    from . import head_train
    M, N = dy.shape
    K = w.shape[1]
    fused = gelu_grad_of is not None and FUSE_ACT and _plain2d(gelu_grad_of)
    if _plain2d(dy) and _plain2d(w) and eligible(M, K, N):
        np_ = padded_k(M, K, N)
        wt = w_cache.planes.get(("t", np_)) if w_cache is not None else None
        if wt is None:
            wt = split_t(w, np_, omax=w_max)
            if w_cache is not None:
                w_cache.put("t", np_, wt)
        dx = gemm(split(dy, np_, omax=dy_max), wt, gelu_grad_of=gelu_grad_of if fused else None)
        if fused:
            return dx
    else:
        dx = head_train.gemm_nn_f32(dy, w)
    if gelu_grad_of is not None:
        from .encoder_train import gelu_bwd
        dx = gelu_bwd(gelu_grad_of, dx)
    return dx


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, x_act: Optional[str] = None, colsum: Optional[torch.Tensor] = None,
            dy_max: Optional[OperandMax] = None, x_max: Optional[OperandMax] = None) -> torch.Tensor:
    """dy [M, N]^T . act(x) [M, K] -> [N, K]: the weight gradient of y = act(x) w^T (contraction over the M rows of both).
    colsum [N] float32: also filled with the bias gradient sum_m dy[m, :] (from the split's pass over dy, or la_colsum_f32)."""
    from . import head_train
    M, N = dy.shape
    K = x.shape[1]
    if _plain2d(dy) and _plain2d(x) and eligible(N, K, M):
        mp = padded_k(N, K, M)
        if not FUSE_ACT:
            x, x_act = _apply(x, x_act), None
        return gemm(split_t(dy, mp, colsum=colsum, omax=dy_max), split_t(x, mp, act=x_act, omax=x_max))
    if colsum is not None:
        check(lib().la_colsum_f32(ptr(dy), dy.stride(0), M, N, ptr(colsum), stream_ptr()), "colsum")
    return head_train.gemm_tn_f32(dy, _apply(x, x_act))
