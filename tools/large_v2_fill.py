"""BASELINE configs[3]: Whisper-large-v2 (d=1280, 32 layers, 20 heads), float16, growing batch of 30 s clips on one MI355X:
throughput and peak HBM per batch size (the [B, T, 21129] logits are never materialised)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyricalignment_amd import whisper_compat as wc
from lyricalignment_amd.module.align_model import AlignModel

wm = wc.build_model("large-v2", seed=0)
model = AlignModel(wm, embed_dim=1280, hidden_dim=384, output_dim=21129, device="cuda", compute_dtype=torch.float16).eval()
eng = model.engine()
del wm
rs = np.random.RandomState(0)
rows = []
json_out = None
args = sys.argv[1:]
if "--json" in args:
    i = args.index("--json"); json_out = args[i + 1]; del args[i:i + 2]
for B in [int(b) for b in (args or ["32", "128", "512", "1024"])]:
    try:
        mel = torch.from_numpy(rs.uniform(-1, 1, size=(B, 80, 3000)).astype(np.float32)).cuda()
        labels = torch.from_numpy(rs.randint(2, 402, size=(B, 26)).astype(np.int32)).cuda()
        n_labels = torch.full((B,), 26, dtype=torch.int32).cuda()
        torch.cuda.reset_peak_memory_stats()
        with torch.no_grad():
            for _ in range(2):
                out = eng.align_mel(mel, labels, n_labels, n_frames=1500, use_ctc=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                out = eng.align_mel(mel, labels, n_labels, n_frames=1500, use_ctc=True)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        ok = int((out[3] != 0).sum()) == 0
        print(f"B={B}: {dt*1e3:.1f} ms per batch (single stream), {B*30/dt:.0f} audio-s/s, peak HBM {torch.cuda.max_memory_allocated()/1e9:.1f} GB, status ok={ok}", flush=True)
        rows.append({"clips": B, "ms_per_batch": dt * 1e3, "audio_s_per_s": B * 30 / dt, "peak_hbm_bytes": int(torch.cuda.max_memory_allocated()),
                     "all_status_ok": ok})
        del mel, labels, n_labels, out
        eng._buf.clear(); eng._ws.clear(); torch.cuda.empty_cache()
    except Exception as e:
        print(f"B={B}: {type(e).__name__}: {str(e)[:200]}", flush=True)
        eng._buf.clear(); eng._ws.clear(); torch.cuda.empty_cache()
        rows.append({"clips": B, "error": f"{type(e).__name__}: {str(e)[:200]}"})
        break
if json_out:
    free, total = torch.cuda.mem_get_info()
    json.dump({"workload": "BASELINE configs[3]: Whisper-large-v2 (d=1280, 32 blocks, 20 heads) align, float16, one MI355X, single stream; "
                           "the [B, 1500, 21129] logits are never materialised", "hbm_total_bytes": int(total), "batches": rows}, open(json_out, "w"), indent=1)
