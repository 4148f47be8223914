"""ms per batch of the alignment path by stage, per compute mode (float32 = the reference's own arithmetic; float16 / bfloat16 = the
throughput modes): encoder alone, head + emission prep + DP alone, whole path -- sequential launches on one stream, HIP events.
usage: mode_bench.py [size=medium] [B=32] [modes=f32,f16,bf16] [iters=5]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lyricalignment_amd import whisper_compat as wc
from lyricalignment_amd.module.align_model import AlignModel

size = sys.argv[1] if len(sys.argv) > 1 else "medium"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
modes = (sys.argv[3] if len(sys.argv) > 3 else "f32,f16,bf16").split(",")
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}

wm = wc.build_model(size, seed=0)
d = wm.dims.n_audio_state
rs = np.random.RandomState(0)
mel = torch.from_numpy(rs.uniform(-1, 1, size=(B, 80, 3000)).astype(np.float32)).cuda()
labels = torch.from_numpy(rs.randint(2, 402, size=(B, 26)).astype(np.int32)).cuda()
n_labels = torch.full((B,), 26, dtype=torch.int32).cuda()


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


out = {}
for mode in modes:
    model = AlignModel(wm, embed_dim=d, hidden_dim=384, output_dim=21129, device="cuda", compute_dtype=DT[mode]).eval()
    wc.init_align_head(model, seed=7)
    with torch.no_grad():
        eng = model.engine()
        feats = eng.encode(mel).clone()
        enc = timed(lambda: eng.encode(mel), iters)
        head = timed(lambda: eng.align_feats(feats, B, 1500, 1500, labels, n_labels, 1), iters)
        full = timed(lambda: eng.align_mel(mel, labels, n_labels, n_frames=1500, use_ctc=True), iters)
        eng.check_gru()
    out[mode] = dict(encoder_ms=enc, head_dp_ms=head, whole_ms=full, audio_s_per_s=B * 30.0 / (full * 1e-3))
    print(f"{size} B={B} {mode}: encoder {enc:.2f} ms, head+DP {head:.2f} ms, whole {full:.2f} ms -> {out[mode]['audio_s_per_s']:.0f} audio-s/s", flush=True)
    del model, eng, feats
    torch.cuda.empty_cache()
print(json.dumps(out))
