#!/usr/bin/env python3
"""Round-3 diagnosis of BENCH_r02's cpu_vs_gpu_onset_mae_s = 1.087: run bench.py's headline pipeline with the driver's
warm-up / step counts and compare ALL 32 clips' boundaries (a) between pipeline shapes / repeats on the device and
(b) against the fp32 oracle for the first clips.

    python tools/selfcheck_repro.py [--oracle-clips 2] [--repeats 2]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle-clips", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--fc-scale", type=float, default=bench.HEAD_FC_SCALE)
    ap.add_argument("--flat-head", action="store_true", help="nn.Module default initialisation of the head (the rounds 1-2 bench)")
    ap.add_argument("--uniform-mel", action="store_true", help="uniform[-1,1] noise mel (the rounds 1-2 bench input)")
    ap.add_argument("--warmups", default="5,2,1,3,4,5")
    ap.add_argument("--tag", default="")
    ap.add_argument("--unfitted-head", action="store_true")
    ap.add_argument("--random-labels", action="store_true", help="random class ids instead of the head's own top classes")
    args = ap.parse_args()
    from lyricalignment_amd import _lib, whisper_compat as wc
    from lyricalignment_amd.engine import PipelinedAligner
    from lyricalignment_amd.module.align_model import AlignModel
    _lib.require_gpu()
    device = torch.device("cuda", 0)
    dims = wc.dims_for(bench.MODEL)
    wm = wc.build_model(bench.MODEL, seed=0)
    model = AlignModel(wm, embed_dim=dims.n_audio_state, hidden_dim=bench.HIDDEN, output_dim=bench.VOCAB, device="cuda:0",
                       compute_dtype=torch.bfloat16).eval()
    if args.flat_head:
        torch.manual_seed(0)
        for m_ in (model.align_rnn.rnn, model.align_rnn.fc):
            m_.reset_parameters()                     # nn.Module's default initialisation: the rounds 1-2 bench head
    elif args.unfitted_head:
        wc.init_align_head(model, seed=7, fc_scale=args.fc_scale)     # peaked but knows nothing of the songs' timbres
    else:
        print(json.dumps({"head_fit": bench.fit_head(model, device)}), flush=True)     # bench.py's head
    with torch.no_grad():
        eng = model.engine()
    mel, labels, n_labels, Ls, plans = bench.build_inputs(device)
    if args.random_labels:
        rl = np.random.RandomState(4)
        for b, L in enumerate(Ls):
            labels[b, :L] = torch.from_numpy(rl.randint(2, 403, size=L).astype(np.int32)).to(device)
    if args.uniform_mel:
        mel = torch.from_numpy(np.random.RandomState(2).uniform(-1.0, 1.0, size=(bench.BATCH, 80, 3000)).astype(np.float32)).to(device)
    B = bench.BATCH
    out = {"runs": []}

    with torch.no_grad():
        ref = eng.align_mel(mel, labels, n_labels, n_frames=bench.T_FRAMES, use_ctc=True)
    torch.cuda.synchronize()
    ref_on, ref_off = ref[0].cpu().numpy().copy(), ref[1].cpu().numpy().copy()
    with torch.no_grad():
        ref2 = eng.align_mel(mel, labels, n_labels, n_frames=bench.T_FRAMES, use_ctc=True)
    torch.cuda.synchronize()
    out["single_stream_repeat_equal"] = bool((ref2[0].cpu().numpy() == ref_on).all() and (ref2[1].cpu().numpy() == ref_off).all())
    mask = np.arange(labels.shape[1])[None, :] < Ls[:, None]

    def diff(on, off):
        d_on = np.abs(on.astype(np.int64) - ref_on)[mask]
        d_off = np.abs(off.astype(np.int64) - ref_off)[mask]
        clips = sorted(set(np.nonzero((np.abs(on.astype(np.int64) - ref_on) * mask).sum(1))[0].tolist()))
        return {"n_diff": int((d_on != 0).sum() + (d_off != 0).sum()), "max_frames": int(max(d_on.max(), d_off.max())),
                "mae_s": float(d_on.mean() * 0.02), "clips": clips}

    for warm in [int(w) for w in args.warmups.split(',') if w]:
        for rep in range(args.repeats):
            pinned = [torch.empty((B, labels.shape[1]), dtype=torch.int32).pin_memory() for _ in range(2)]
            pst = torch.empty((B,), dtype=torch.int32).pin_memory()
            pipe = PipelinedAligner(eng, head_group=2)
            outs = []
            with torch.no_grad():
                for _ in range(warm):
                    outs.append(pipe.submit(mel, labels, n_labels, n_frames=bench.T_FRAMES, use_ctc=True, host_out=(pinned[0], pinned[1], pst)))
                pipe.drain()
                torch.cuda.synchronize()
                warm_res = [diff(o[0].cpu().numpy(), o[1].cpu().numpy()) for o in outs]
                outs = []
                for _ in range(args.steps):
                    outs.append(pipe.submit(mel, labels, n_labels, n_frames=bench.T_FRAMES, use_ctc=True, host_out=(pinned[0], pinned[1], pst)))
                pipe.drain()
                torch.cuda.synchronize()
            timed_res = [diff(o[0].cpu().numpy(), o[1].cpu().numpy()) for o in outs]
            rec = {"warmup": warm, "rep": rep, "pinned": diff(pinned[0].numpy(), pinned[1].numpy()),
                   "warm_bad": [(i, r) for i, r in enumerate(warm_res) if r["n_diff"]],
                   "timed_bad": [(i, r) for i, r in enumerate(timed_res) if r["n_diff"]]}
            out["runs"].append(rec)
            print(json.dumps(rec), flush=True)

    # oracle on the first clips
    if args.oracle_clips > 0:
        from oracle import alignment_oracle as ao
        from oracle import model_oracle as mo
        ao.build()
        torch.set_num_threads(bench.usable_cores())
        p = {"encoder." + k: v.detach().float().cpu() for k, v in model.whisper_model.encoder.state_dict().items()}
        p.update({"align_rnn." + k: v.detach().float().cpu() for k, v in model.align_rnn.state_dict().items()})
        orc = []
        for b in range(args.oracle_clips):
            L = int(Ls[b])
            with torch.no_grad():
                emb = mo.encoder_forward(p, mel[b:b + 1].cpu(), n_head=dims.n_audio_head)
                logits = mo.gru_head_forward(p, emb)
                res = ao.perform_viterbi_ctc(logits, labels[b:b + 1, :L].cpu().long())
            cpu_on = np.array([s[0] for s in res[0]])
            cpu_off = np.array([s[1] for s in res[0]])
            gpu_on, gpu_off = ref_on[b, :L] * 0.02, ref_off[b, :L] * 0.02
            lg = logits[0]
            # how decided is the lattice?  score of the oracle's best path on the ORACLE's emissions against the score of the
            # device's path on the same emissions (a gap below the 16-bit modes' accumulated emission error is a coin toss)
            lp, ls = mo.emission_prep_ctc(logits)
            em = np.concatenate([ls[0].numpy(), lp[0][:, (labels[b, :L].cpu().long() - 1)].numpy()], axis=1).astype(np.float64)

            def path_score(on_f, off_f):
                state = np.zeros(em.shape[0], dtype=np.int64)
                for i_, (a_, z_) in enumerate(zip(on_f, off_f)):
                    state[int(a_):int(z_)] = i_ + 1
                return float(em[np.arange(em.shape[0]), state].sum())

            rec = {"clip": b, "L": L, "onset_mae_s": float(np.abs(cpu_on - gpu_on).mean()), "offset_mae_s": float(np.abs(cpu_off - gpu_off).mean()),
                   "n_equal": int((np.abs(cpu_on - gpu_on) < 1e-9).sum() + (np.abs(cpu_off - gpu_off) < 1e-9).sum()), "n_bound": 2 * L,
                   "n_within2": int((np.abs(cpu_on - gpu_on) < 0.041).sum() + (np.abs(cpu_off - gpu_off) < 0.041).sum()),
                   "max_dev_s": float(max(np.abs(cpu_on - gpu_on).max(), np.abs(cpu_off - gpu_off).max())),
                   "mean_frames_per_label": float(np.mean(cpu_off - cpu_on) / 0.02),
                   "onset_vs_note_edge_mae_s": float(np.abs(cpu_on - plans[b][0][:-1] * 0.01).mean()),
                   "oracle_path_score": path_score(np.round(cpu_on / 0.02), np.round(cpu_off / 0.02)),
                   "device_path_score_on_oracle_emissions": path_score(ref_on[b, :L], ref_off[b, :L]),
                   "logit_absmax": float(lg.abs().max()), "logit_std": float(lg.std())}
            orc.append(rec)
            print(json.dumps(rec), flush=True)
        out["oracle"] = orc
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"selfcheck_repro{args.tag}.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
