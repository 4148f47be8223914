"""BASELINE configs[4] shape on one GPU: S songs of 180 s (mel resident in HBM) -> 6 encoder chunks per song in one
encoder batch -> BiGRU over T = 9000 frames per song -> emissions -> Viterbi with L labels per song.
Prints ms per pass, audio-seconds per second and the split encoder / head+DP (events on the launch stream).
With batches > 1 the same songs also go through PipelinedAligner.submit_songs (head of batch i under the encoder of i+1).
usage: longform_bench.py [songs=5] [labels=238] [size=medium] [batches=4] [head_group=1]"""
import json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyricalignment_amd import whisper_compat as wc, _lib, ops
from lyricalignment_amd.module.align_model import AlignModel

json_out = None
if "--json" in sys.argv:
    i = sys.argv.index("--json"); json_out = sys.argv[i + 1]; del sys.argv[i:i + 2]
S = int(sys.argv[1]) if len(sys.argv) > 1 else 5
L = int(sys.argv[2]) if len(sys.argv) > 2 else 238
size = sys.argv[3] if len(sys.argv) > 3 else "medium"
NB = int(sys.argv[4]) if len(sys.argv) > 4 else 4
G = int(sys.argv[5]) if len(sys.argv) > 5 else 1
wm = wc.build_model(size, seed=0)
d = wm.dims.n_audio_state
model = AlignModel(wm, embed_dim=d, hidden_dim=384, output_dim=21129, device="cuda", compute_dtype=torch.bfloat16).eval()
eng = model.engine()
rs = np.random.RandomState(0)
mel = torch.from_numpy(rs.uniform(-1, 1, size=(S, 80, 18000)).astype(np.float32)).cuda()
labels = torch.from_numpy(rs.randint(2, 402, size=(S, L)).astype(np.int64))


def one():
    return model.align(mel=mel, labels=labels, use_ctc=True, return_frames=True)


with torch.no_grad():
    for _ in range(2):
        on, off, score, status = one()
    torch.cuda.synchronize()
    assert int(status.abs().sum()) == 0
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    # split: encoder alone on the same chunks
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    feats, B, T, stride = model._features(mel, True)
    ev[1].record()
    from lyricalignment_amd.utils.alignment import _labels_to_device
    lab_dev, n_lab, _ = _labels_to_device(labels, B, eng.device)
    em = eng.emissions(feats, B, T, stride, lab_dev, n_lab, _lib.LA_VARIANT_CTC)
    nf = torch.full((B,), T, dtype=torch.int32, device=eng.device)
    ops.viterbi_batch(em, lab_dev, n_lab, nf)
    ev[2].record()
    torch.cuda.synchronize()
    print(f"{size}: {S} songs x 180 s, {L} labels, T={T}: {ms:.1f} ms per pass = {S * 180 / ms * 1e3:.0f} audio-s/s; "
          f"encoder {ev[0].elapsed_time(ev[1]):.1f} ms, head + DP {ev[1].elapsed_time(ev[2]):.1f} ms")
    rec = {"workload": f"BASELINE configs[4]: {S} songs x 180 s (6 chunks of 30 s each, T = {T} frames per song), {L} labels per song, "
                       f"whisper-{size} bf16, one MI355X", "songs_per_batch": S, "labels": L, "frames": T,
           "single_stream": {"ms_per_batch": ms, "audio_s_per_s": S * 180 / ms * 1e3, "encoder_ms": ev[0].elapsed_time(ev[1]),
                             "head_dp_ms": ev[1].elapsed_time(ev[2])}}
    if NB > 1:
        from lyricalignment_amd.engine import PipelinedAligner
        pipe = PipelinedAligner(eng, head_group=G)
        for _ in range(2):
            outs = [pipe.submit_songs(mel, lab_dev, n_lab) for _ in range(NB)]
            pipe.drain()
        assert all(torch.equal(o[0], on) and torch.equal(o[1], off) for o in outs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            for _ in range(NB):
                pipe.submit_songs(mel, lab_dev, n_lab)
            pipe.drain()
        ms = (time.perf_counter() - t0) / (reps * NB) * 1e3
        print(f"pipelined (head_group {G}, {NB} batches): {ms:.1f} ms per batch of {S} songs = {S * 180 / ms * 1e3:.0f} audio-s/s")
        rec["pipelined"] = {"head_group": G, "batches": NB, "ms_per_batch": ms, "audio_s_per_s": S * 180 / ms * 1e3}
    rec["peak_hbm_bytes"] = int(torch.cuda.max_memory_allocated())
    if json_out:
        json.dump(rec, open(json_out, "w"), indent=1)
