import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyricalignment_amd import _lib, ops
torch.manual_seed(0)
B, T, H = 32, 1500, 16
qkv = torch.randn(B * T, 3 * H * 64, device="cuda").bfloat16()
qkv[:, : H * 64] *= 0.125 * 1.4427
out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2] * 1e3
res = {}
outs = {}
for rd in range(3):
    for nw in ("4", "8"):
        os.environ["LA_ATTN_NW"] = nw
        _lib.set_option("attn_nw", int(nw))
        res.setdefault(nw, []).append(timeit(lambda: ops.attention(qkv, B, T, H, out=out, q_log2=True)))
        outs[nw] = out.clone()
for nw in res:
    print(f"exp2-domain attention, {nw} waves per workgroup: {sorted(res[nw])[1]:.1f} us")
print("max abs diff 8 vs 4:", float((outs["8"].float() - outs["4"].float()).abs().max()))
