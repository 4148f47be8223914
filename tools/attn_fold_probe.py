#!/usr/bin/env python3
"""Timing A/B of the bf16 attention kernel on the Whisper-medium layer shape (32 clips x 1500 frames x 16 heads): the optimistic
softmax with scores scaled by log2 e and a running maximum subtracted per element, against LA_Q_LOG2 (q carries log2 e: scores in
the exp2 domain, no subtraction while no query has needed a maximum).  Both runs compute the same softmax."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyricalignment_amd import ops
from tools.kbench import timeit, rnd

B, T, H = 32, 1500, 16
qkv = rnd(B * T, 3 * H * 64)
qkv[:, : H * 64] *= 0.125
qkv2 = qkv.clone()
qkv2[:, : H * 64] = (qkv[:, : H * 64].float() * 1.4426950408889634).to(torch.bfloat16)
out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
out2 = torch.empty_like(out)
fl = 4.0 * T * T * H * 64 * B
for rd in range(4):
    med, mn = timeit(lambda: ops.attention(qkv, B, T, H, out=out), 30)
    print(f"default : median {med*1e3:.1f} us  min {mn*1e3:.1f} us  {fl/med/1e9:.1f} TF/s", flush=True)
    med, mn = timeit(lambda: ops.attention(qkv2, B, T, H, out=out2, q_log2=True), 30)
    print(f"q_log2  : median {med*1e3:.1f} us  min {mn*1e3:.1f} us  {fl/med/1e9:.1f} TF/s   max abs diff to default {float((out.float()-out2.float()).abs().max()):.3e}", flush=True)
