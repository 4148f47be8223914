"""The fine-tune step's gradient at BASELINE configs[2] size (Whisper-medium, 8 x 2 clips fused), f16x2 path against the float32-MFMA path:
one process, one model, the same data and dropout draws; the two flat gradient buckets (head, backbone) compared after ONE accumulate().

    python tools/ft_grad_compare.py > gpurun_out/r5_finetune_gradient_parity_medium.txt

Both paths approximate exact float32 arithmetic (the f16x2 products are closer to float64 than the float32 MFMA kernels:
profiles/r5_kbench_f32emu.txt); the comparison bounds what the scheme changes in the quantity the optimizer sees."""
import os, sys, time
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench                                                    # noqa: E402
from lyricalignment_amd import _lib, f32x2, finetune as ft, ops, whisper_compat as wc      # noqa: E402
from lyricalignment_amd.module.align_model import AlignModel   # noqa: E402

_lib.require_gpu()
dev = torch.device("cuda:0")
dims = wc.dims_for("medium")
wm = bench.build_weights("medium", True, 0, 1, None)
torch.manual_seed(0)
model = AlignModel(wm, embed_dim=dims.n_audio_state, hidden_dim=bench.HIDDEN, output_dim=bench.VOCAB, dropout=0.15, train_transcript=True,
                   device="cuda:0").to(dev)
tuner = ft.FineTuner(model, warmup_steps=1, train_steps=10000, allreduce_chunks=0)
B, n_tok, accum = 2, 32, 8
rs = np.random.RandomState(114514)
audios = [(rs.randn(480000) * 0.1).astype(np.float32) for _ in range(B)]
labels = torch.from_numpy(rs.randint(1, 402, size=(B, 26)))
fl = torch.full((B, bench.T_FRAMES), -100, dtype=torch.long)
for b in range(B):
    for i in range(26):
        fl[b, 40 + 50 * i: 40 + 50 * i + 30] = labels[b, i]
micro = dict(audios=audios, ctc_labels=labels, frame_labels=fl, decoder_input=torch.from_numpy(rs.randint(0, 50000, size=(B, n_tok))),
             decoder_output=torch.from_numpy(rs.randint(0, 50000, size=(B, n_tok))))


def grads(x2: bool):
    f32x2.ENABLED = x2
    ops.ATTN_F16X2 = x2
    _lib.set_option("gru_handoff", 0 if x2 else 1)             # 1 = the float32-MFMA GRU training sweeps (counter hand-off)
    for g in tuner.grad:
        g.zero_()
    torch.manual_seed(1234)                                     # the dropout draws
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = tuner.accumulate([micro] * accum, accum_grad_steps=accum, fused=True)
    torch.cuda.synchronize()
    return [g.clone() for g in tuner.grad], losses.cpu().double(), time.perf_counter() - t0


grads(True)                                                     # warm-up (allocator, lazily built constants)
ga, la, ta = grads(True)
gb, lb, tb = grads(False)
gc, lc, _ = grads(True)                                         # the f16x2 path twice: run-to-run
print(f"one accumulate() of {accum} x {B} clips, Whisper-medium: f16x2 path {ta * 1e3:.0f} ms, float32-MFMA path {tb * 1e3:.0f} ms")
print("summed losses [CE, BCE, CTC, decoder CE]:  f16x2", [f"{v:.6f}" for v in la.tolist()], " float32", [f"{v:.6f}" for v in lb.tolist()])
for name, a, b, c in zip(("head bucket", "backbone bucket"), ga, gb, gc):
    a64, b64 = a.double(), b.double()
    rel = float((a64 - b64).norm() / b64.norm())
    mx = float((a64 - b64).abs().max() / b64.abs().max())
    cos = float((a64 * b64).sum() / (a64.norm() * b64.norm()))
    print(f"{name}: {a.numel()} floats, |g| {float(b64.norm()):.4e}; f16x2 vs float32 path: relative L2 difference {rel:.2e}, "
          f"max |difference| / max |g| {mx:.2e}, 1 - cosine {1 - cos:.1e}; f16x2 run to run: max |difference| {float((a - c).abs().max()):.1e}")
