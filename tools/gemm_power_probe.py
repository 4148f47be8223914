#!/usr/bin/env python3
"""Is the 256x256 GEMM power-limited or feed-limited?  The same four encoder shapes on (a) random operands, (b) all-zero operands
(identical instruction stream and memory traffic, no data toggling -> no dynamic power in the multipliers / LDS / register file).
    python tools/gemm_power_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lyricalignment_amd import ops

def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[len(ts) // 2]

M = 48000
for name, N, K, f32out in (("qkv", 3072, 1024, False), ("mlp_up+gelu", 4096, 1024, False), ("out_proj+res", 1024, 1024, True), ("mlp_down+res", 1024, 4096, True)):
    for kind in ("random", "zeros", "random"):
        if kind == "random":
            a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        else:
            a = torch.zeros(M, K, device="cuda", dtype=torch.bfloat16); w = torch.zeros(N, K, device="cuda", dtype=torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda") if f32out else None
        out = torch.empty(M, N, device="cuda", dtype=torch.float32 if f32out else torch.bfloat16)
        ms = timeit(lambda: ops.gemm(a, w, out, bias=bias, residual=res, gelu="gelu" in name, out_f32=f32out))
        print(f"{name:14s} {kind:7s} {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
