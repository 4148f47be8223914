#!/usr/bin/env python3
"""Full-depth parity numbers (developer tool; run on the GPU box): Whisper-medium / tiny / large-v2 random-init weights,
one clip, HIP path in float32 / bfloat16 / float16 against the CPU oracle -- max / mean error of encoder output, logits and
CTC emissions, and the exact-match rate of onset / offset frames against the oracle's own end-to-end result on peaked
emissions.  The tests in tests/test_gpu_parity_full.py assert bounds calibrated from this table.
    python tools/depth_parity.py [medium|tiny|large-v2 ...] > gpurun_out/depth_parity.json"""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lyricalignment_amd import _lib, whisper_compat as wc
from lyricalignment_amd.module.align_model import AlignModel
from lyricalignment_amd.utils import alignment as ua
from oracle import alignment_oracle as ao, model_oracle as mo


def wave(n, seed=0):
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    return (rs.randn(n) * 0.05 + 0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t * (1 + 0.1 * t))).astype(np.float32)


def head_init(model, hidden, fc_scale, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.align_rnn.named_parameters():
            s = fc_scale if n.startswith("fc.weight") else 1.5
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (s / hidden ** 0.5))


def run(name, vocab, n_samples, L, fc_scale=12.0, dtypes=(torch.float32, torch.bfloat16, torch.float16)):
    dims = wc.dims_for(name)
    wm = wc.build_model(name, seed=3)
    d, H = dims.n_audio_state, dims.n_audio_head
    audio = wave(n_samples, 5)
    labels = torch.from_numpy(np.random.RandomState(6).randint(2, min(vocab - 2, 403), size=(1, L)))
    labels[0, 3] = labels[0, 2]
    out = {"model": name, "n_samples": n_samples, "L": L}
    ref = None
    for dt in dtypes:
        model = AlignModel(wm, embed_dim=d, hidden_dim=384, output_dim=vocab, device="cuda", compute_dtype=dt).eval()
        head_init(model, 384, fc_scale, 7)
        if ref is None:
            p = {"encoder." + k: v.detach().float().cpu() for k, v in wm.encoder.state_dict().items()}
            p.update({"align_rnn." + k: v.detach().float().cpu() for k, v in model.align_rnn.state_dict().items()})
            t0 = time.time()
            mel = mo.pad_or_trim(mo.log_mel_spectrogram(audio[None]), 3000)
            T = mo.frame_count(n_samples // 160)
            enc = mo.encoder_forward(p, mel, n_head=H)
            logits = mo.gru_head_forward(p, enc[:, :T])
            lp, ls = mo.emission_prep_ctc(logits)
            secs = ao.perform_viterbi_ctc(logits, labels)
            out["oracle_s"] = round(time.time() - t0, 2)
            ref = (enc, logits, lp, ls, secs, T)
        enc, logits, lp, ls, secs, T = ref
        with torch.no_grad():
            ours_enc = model.whisper_model.embed_audio(torch.as_tensor(mo.pad_or_trim(mo.log_mel_spectrogram(audio[None]), 3000)).cuda()).cpu() \
                if dt == torch.float32 else None
            lg, _ = model.frame_manual_forward([audio])
            got = model.align([audio], labels, use_ctc=True)
            two = ua.perform_viterbi_ctc(lg, labels)
            eng = model.engine()
            lab_dev, n_lab, lists = ua._labels_to_device(labels, 1, eng.device)
            feats, B, T2, stride = model._features(model._mel_of([audio]), True)
            em = eng.emissions(feats, B, T2, stride, lab_dev, n_lab, _lib.LA_VARIANT_CTC).cpu()
        idx = torch.tensor(lists[0]) - 1
        e_lab = (em[0, :, 1:1 + L] - lp[0][:, idx]).abs()
        e_sil = (em[0, :, 0] - ls[0, :, 0]).abs()
        e_log = (lg.cpu() - logits).abs()
        flat = lambda r: np.array([x for u in r for seg in u for x in seg])
        row = {"logits_max": float(e_log.max()), "logits_mean": float(e_log.mean()), "logits_absmax_ref": float(logits.abs().max()),
               "em_max": float(max(e_lab.max(), e_sil.max())), "em_mean": float(torch.cat([e_lab.flatten(), e_sil]).mean()),
               "fused_vs_oracle_exact": float(np.mean(flat(got) == flat(secs))), "two_step_vs_oracle_exact": float(np.mean(flat(two) == flat(secs))),
               "fused_vs_two_step_exact": float(np.mean(flat(got) == flat(two))),
               "fused_max_dev_s": float(np.abs(flat(got) - flat(secs)).max())}
        if ours_enc is not None:
            row["enc_max"] = float((ours_enc - enc).abs().max())
        out[str(dt).split(".")[-1]] = row
        del model
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "medium", "large-v2"]
    res = []
    for w in which:
        if w == "tiny":
            res.append(run("tiny", 21129, 60096, 11))
            res.append(run("tiny", 21129, 480000, 26))
        elif w == "medium":
            res.append(run("medium", 21129, 480000, 26))
        else:
            res.append(run("large-v2", 21129, 480000, 26, dtypes=(torch.float16, torch.float32)))
        print(json.dumps(res[-1]), flush=True)
