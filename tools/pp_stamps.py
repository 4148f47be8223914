"""Diagnostic (needs a library built with LA_EXTRA_CXXFLAGS=-DLA_PP_STAMPS): per-segment cycle counts of one K-tile of the 256x256
ping-pong GEMM main loop, median over the sampled workgroups, for wave group 0 and group 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyricalignment_amd import ops

M, N, K = 48000, 3072, 1024
a = (torch.randn(M, K, device="cuda")).bfloat16()
w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
bias = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
buf = torch.zeros(64 * 8 * 32, dtype=torch.int32, device="cuda")
os.environ["LA_STAMP_PTR"] = hex(buf.data_ptr())
for _ in range(5):
    ops.gemm(a, w, out, bias=bias)
torch.cuda.synchronize()
s = buf.cpu().numpy().reshape(64, 8, 32)[:, :, :25].astype(np.int64) & 0xFFFFFFFF
d = (s[:, :, 1:] - s[:, :, :-1]) & 0xFFFFFFFF
names = ["reads issued", "DMA issued", "barrier", "(head DMA +) lgkmcnt(0)", "16 MFMAs", "barrier (+vmcnt)"]
for g in (0, 1):
    dd = d[:, 4 * g:4 * g + 4, :].reshape(-1, 24)
    med = np.median(dd, axis=0)
    print(f"wave group {g}: K-tile {np.median((s[:, 4*g:4*g+4, 24] - s[:, 4*g:4*g+4, 0]) & 0xFFFFFFFF):.0f} cycles")
    for ph in range(4):
        print(f"   phase {ph}: " + "  ".join(f"{names[i]} {med[ph * 6 + i]:.0f}" for i in range(6)))
