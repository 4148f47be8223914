import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from lyricalignment_amd import ops, head_train as ht
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for (M,N,K) in [(1024,1024,3008),(1024,4096,3008),(3072,1024,3008),(3008,1024,1024),(3008,1024,4096),(3008,4096,1024)]:
    at=torch.randn(K,M,device="cuda"); wt=torch.randn(K,N,device="cuda")
    a=at.T.contiguous(); w=wt.T.contiguous()
    fl=2.0*M*N*K
    t_nt=timeit(lambda: ops.gemm(a,w,out_f32=True)); t_tt=timeit(lambda: ht.gemm_tn(at,wt)); t_tw=timeit(lambda: ht.gemm_nn(a,wt))
    print(f"M={M} N={N} K={K}: NT {t_nt:.0f} us ({fl/t_nt/1e6:.0f} TF/s)  TT {t_tt:.0f} us ({fl/t_tt/1e6:.0f})  TW {t_tw:.0f} us ({fl/t_tw/1e6:.0f})")
