"""Yardstick only (not used by the product): what the vendor library reaches on the encoder's GEMM shapes, to judge how far the
hand-written kernel is from a tuned one.  torch.matmul -> hipBLASLt / rocBLAS, bf16 inputs, f32 accumulate."""
import torch, time
shapes = [(48000, 3072, 1024), (48000, 1024, 1024), (48000, 4096, 1024), (48000, 1024, 4096)]
for M, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        c = a @ w.t()
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 10
    print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.0f} TF/s", flush=True)
