"""How long one ROUND of 256x256 tiles takes as the chip fills (developer tool, DESIGN.md "GEMM, round 2"): the out-projection GEMM
(N = K = 1024) at 64 .. 752 tiles, with the f32-residual epilogue and with the plain 16-bit one.  A lone quarter of the chip runs a
tile in ~19 us (46 k cycles at 2.4 GHz); all 256 CUs together take ~37 us per round -- the clock under full MFMA load -- and the
f32 residual epilogue adds ~13 us per round of bandwidth-bound stores that every CU issues at the same moment.
    python tools/gemm_rounds.py
"""
import os, sys, torch
sys.path.insert(0, ".")
from lyricalignment_amd import ops
from tools.kbench import timeit, rnd
N, K = 1024, 1024
for M in (256 * 16, 256 * 32, 256 * 64, 256 * 128, 48000):
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    out16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    med, mn = timeit(lambda: ops.gemm(a, w, out, bias=bias, residual=res, out_f32=True), 20)
    med2, mn2 = timeit(lambda: ops.gemm(a, w, out16, bias=bias), 20)
    tiles = (M + 255) // 256 * 4
    print(f"out_proj M={M}: {tiles} tiles ({tiles/256:.2f} rounds): f32 residual epilogue {med*1e3:.1f} us, plain bf16 epilogue {med2*1e3:.1f} us", flush=True)
