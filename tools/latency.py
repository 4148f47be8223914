"""Single-clip latency of the alignment path (B=1, 30 s), eager launches vs the encoder captured in a HIP graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyricalignment_amd import whisper_compat as wc
from lyricalignment_amd.module.align_model import AlignModel

size = sys.argv[1] if len(sys.argv) > 1 else "medium"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wm = wc.build_model(size, seed=0)
d = wm.dims.n_audio_state
model = AlignModel(wm, embed_dim=d, hidden_dim=384, output_dim=21129, device="cuda", compute_dtype=torch.bfloat16).eval()
eng = model.engine()
rs = np.random.RandomState(0)
mel = torch.from_numpy(rs.uniform(-1, 1, size=(B, 80, 3000)).astype(np.float32)).cuda()
labels = torch.from_numpy(rs.randint(2, 402, size=(B, 26)).astype(np.int32)).cuda()
n_labels = torch.full((B,), 26, dtype=torch.int32).cuda()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    return float(np.median(t)) * 1e3


with torch.no_grad():
    full = timed(lambda: eng.align_mel(mel, labels, n_labels, n_frames=1500, use_ctc=True))
    enc = timed(lambda: eng.encode(mel))
    print(f"{size} B={B}: align_mel {full:.2f} ms, encoder alone {enc:.2f} ms (eager)")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.encode(mel)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        y = eng.encode(mel)
    encg = timed(lambda: g.replay())
    print(f"{size} B={B}: encoder in a HIP graph {encg:.2f} ms")
