"""Flat files for consumers of the C ABI that are not Python (examples/align_capi.cpp): the packed weights of an AlignEngine in the
order la_encoder_weights / la_encoder_block / la_head_weights list them (include/lyricalign.h), and an input batch.
Data movement only: every tensor is written as `int64 nbytes` + its bytes in the layout engine.pack_encoder / pack_head produced."""
from __future__ import annotations

import struct

import torch

from . import _lib

MAGIC = 0x4C414331   # "LAC1"


def _tensor(f, t) -> None:
    if t is None:
        f.write(struct.pack("<q", 0))
        return
    b = t.detach().contiguous().cpu().view(torch.uint8).numpy().tobytes() if t.dtype != torch.bfloat16 else \
        t.detach().contiguous().cpu().view(torch.int16).numpy().tobytes()
    f.write(struct.pack("<q", len(b)))
    f.write(b)


def write_weights(engine, path: str) -> None:
    """engine: lyricalignment_amd.engine.AlignEngine (encoder + bidirectional 2-layer head packed)."""
    e, h = engine.enc, engine.head
    if h is None or len(h.w_ih) != 2:
        raise ValueError("capi_export: the engine needs the 2-layer head")
    with open(path, "wb") as f:
        f.write(struct.pack("<i", MAGIC))
        f.write(struct.pack("<5i", _lib.dtype_code(e.dtype) | (_lib.LA_Q_LOG2 if e.q_log2 else 0), e.d, e.n_head, len(e.blocks), e.n_mels))
        for t in (e.conv1_w, e.conv1_b, e.conv2_w, e.conv2_b, e.pos, e.lnp_g, e.lnp_b):
            _tensor(f, t)
        for b in e.blocks:
            for t in (b.ln1_g, b.ln1_b, b.wqkv, b.bqkv, b.wo, b.bo, b.ln2_g, b.ln2_b, b.w1, b.b1, b.w2, b.b2,
                      b.wqkv_ln, b.cqkv, b.bqkv_ln, b.w1_ln, b.c1, b.b1_ln):
                _tensor(f, t)
        f.write(struct.pack("<5i", _lib.dtype_code(h.dtype), h.hidden, h.in_dim, h.vocab, 2))
        for l in range(2):
            for t in (h.w_ih[l], h.b_ih[l], h.w_hh[l], h.b_hh[l]):
                _tensor(f, t)
        _tensor(f, h.w_fc)
        _tensor(f, h.b_fc)


def write_input(path: str, mel: torch.Tensor, labels: torch.Tensor, n_labels: torch.Tensor, frames: int, variant: int) -> None:
    """mel [B, n_mels, 3000] f32, labels [B, Lmax] i32, n_labels [B] i32."""
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", mel.shape[0], int(frames), labels.shape[1], int(variant)))
        _tensor(f, mel.float())
        _tensor(f, labels.to(torch.int32))
        _tensor(f, n_labels.to(torch.int32))
