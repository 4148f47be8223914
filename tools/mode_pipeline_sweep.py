"""Pipelined ms per step of one compute mode of the headline workload against the number of encoder streams / head group.
usage: mode_pipeline_sweep.py [mode=f32] [steps=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from lyricalignment_amd import whisper_compat as wc
from lyricalignment_amd.engine import PipelinedAligner
from lyricalignment_amd.module.align_model import AlignModel

mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[mode]
dims = wc.dims_for("medium")
model = AlignModel(wc.build_model("medium", seed=0), embed_dim=dims.n_audio_state, hidden_dim=bench.HIDDEN, output_dim=bench.VOCAB, device="cuda:0", compute_dtype=dt).eval()
dev = torch.device("cuda", 0)
bench.fit_head(model, dev)
with torch.no_grad():
    eng = model.engine()
mel, labels, n_labels, Ls, plans = bench.build_inputs(dev)
for streams, group in ((1, 4), (2, 4), (3, 4), (2, 2), (3, 6), (4, 4)):
    pipe = PipelinedAligner(eng, head_group=group, encoder_streams=streams)
    def run(n):
        with torch.no_grad():
            for _ in range(n):
                pipe.submit(mel, labels, n_labels, n_frames=1500, use_ctc=True)
            pipe.drain()
        torch.cuda.synchronize()
    run(group)
    t = []
    for rep in range(2):
        t0 = time.perf_counter(); run(steps); t.append((time.perf_counter() - t0) / steps * 1e3)
    print(f"{mode}: {streams} encoder stream(s), head over {group} batches: {min(t):.2f} ms per step", flush=True)
    del pipe
    torch.cuda.empty_cache()
