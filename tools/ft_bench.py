"""Time the whole-model fine-tune step (BASELINE configs[2]: Whisper-medium, batch 2 x 30 s, CE + BCE + CTC + decoder CE) on
one MI355X: micro-step (forward + losses + backward) and optimizer step.  Synthetic audio / labels, random-init weights."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from lyricalignment_amd import finetune as ft, whisper_compat as wc
from lyricalignment_amd.module.align_model import AlignModel

ap = argparse.ArgumentParser()
ap.add_argument("--size", default="medium")
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--accum", type=int, default=2)
ap.add_argument("--no-decoder", action="store_true")
a = ap.parse_args()
dims = wc.dims_for(a.size) if hasattr(wc, "dims_for") else None
wm = wc.build_model(a.size, seed=0, with_decoder=not a.no_decoder) if dims is None else wc.build_model(dims=dims, seed=0, with_decoder=not a.no_decoder)
d = wm.dims.n_audio_state
model = AlignModel(wm, embed_dim=d, hidden_dim=384, output_dim=21129, dropout=0.15, train_transcript=not a.no_decoder, device="cuda").to("cuda")
tuner = ft.FineTuner(model, warmup_steps=1, train_steps=100)
rs = np.random.RandomState(0)
audios = [(rs.randn(480000) * 0.1).astype(np.float32) for _ in range(a.batch)]
labels = torch.from_numpy(rs.randint(1, 402, size=(a.batch, 26)))
fl = torch.full((a.batch, 1500), -100, dtype=torch.long)
for b in range(a.batch):
    for i in range(26):
        fl[b, 40 + 50 * i: 40 + 50 * i + 30] = labels[b, i]
dec_in = torch.from_numpy(rs.randint(0, 50000, size=(a.batch, 40)))
dec_out = torch.from_numpy(rs.randint(0, 50000, size=(a.batch, 40)))
res = []
for it in range(a.steps + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.accum):
        l = tuner.micro_step(audios, labels, fl, None if a.no_decoder else dec_in, None if a.no_decoder else dec_out, accum_grad_steps=a.accum)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    tuner.step()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    res.append(((t1 - t0) / a.accum, t2 - t1, l.cpu().tolist()))
    print(f"iter {it}: micro-step {res[-1][0] * 1e3:.1f} ms, optimizer step {res[-1][1] * 1e3:.1f} ms, losses {res[-1][2]}", flush=True)
ms = float(np.mean([r[0] for r in res[1:]])) * 1e3
print(json.dumps({"workload": f"whisper-{a.size} fine-tune micro-step, batch {a.batch} x 30 s", "micro_step_ms": ms,
                  "optimizer_step_ms": float(np.mean([r[1] for r in res[1:]])) * 1e3,
                  "audio_seconds_per_second": a.batch * 30.0 / (ms / 1e3), "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}))
