#!/usr/bin/env python3
"""bench.py -- aligned audio-seconds per wall-second of the AlignModel hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload = BASELINE.json configs[1]: Whisper-medium encoder + BiGRU/FC head + CTC forced alignment,
batch = 32 x 30 s synthetic mel, bf16 operands (f32 accumulate / residual / softmax / DP in f64), one MI355X.
A "step" is one pass of the whole hot path over one batch already resident in HBM:
mel [32,80,3000] -> conv stem -> 24 blocks -> ln_post -> 2 x BiGRU -> Mish -> fused FC + emission prep
-> batched Viterbi -> onset/offset frames (device) -> async D2H of the [32, Lmax] int32 results.
Multi-GPU: clips are independent, so every rank aligns its own batch with NO data-path collective
("weak" scaling); torch.distributed (RCCL) is only used for the barrier and the MAX over ranks of the time.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL = "medium"
BATCH, CLIP_SECONDS, T_FRAMES = 32, 30.0, 1500
HIDDEN, VOCAB = 384, 21129
HEAD_FC_SCALE = 12.0     # output Linear of an unfitted synthetic head (tests): |logit| <= ~20, posteriors peaked like a trained head's


def algorithmic_gemm_flops_per_clip(d: int, n_layer: int, H: int = HIDDEN) -> float:
    """SURVEY.md 2.1 formulae restricted to the launches of the gemm_bf16 kernel family:
    conv1 + conv2 + per layer (QKV/out 8*T*d^2 + MLP 16*T*d^2) + the GRU input projections + the gathered-column GEMM."""
    T = T_FRAMES
    conv = 2 * 3000 * 80 * 3 * d + 2 * T * d * 3 * d
    layers = n_layer * (8 * T * d * d + 16 * T * d * d)
    gru_in = 2 * (2 * T * d * 3 * H) + 2 * (2 * T * 2 * H * 3 * H)
    gather = 2 * T * 2 * H * 27
    return float(conv + layers + gru_in + gather)


def total_flops_per_clip(d: int, n_layer: int, H: int = HIDDEN, V: int = VOCAB) -> float:
    T = T_FRAMES
    enc = 2 * 3000 * 80 * 3 * d + 2 * T * d * 3 * d + n_layer * (8 * T * d * d + 4 * T * T * d + 16 * T * d * d)
    gru = 2 * (2 * T * d * 3 * H + 2 * T * H * 3 * H) + 2 * (2 * T * 2 * H * 3 * H + 2 * T * H * 3 * H)
    return float(enc + gru + 2 * T * 2 * H * V)


PMC_SUMMARY = "r6_pmc_gemm_pp.csv"   # FETCH_SIZE / WRITE_SIZE passes over bench.py itself (tools/pmc_traffic.sh r6)


def measured_gemm_traffic_per_launch():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950, WRITE_SIZE as is; both in KiB), averaged over the
    encoder's four GEMM shapes weighted by their launches per step.  None if the summary is not there."""
    path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    if not os.path.exists(path):
        return None
    per_shape = {}
    with open(path) as f:
        next(f)
        for line in f:
            grid, variant, counter, val = line.strip().split(",")
            per_shape.setdefault((grid, variant), {})[counter] = float(val)
    # grid sizes (threads) of the config-2 shapes: QKV, MLP-up, and the two f32-out shapes (out-proj == MLP-down grid)
    weights = {("1155072", "bf16out"): 24, ("1540096", "bf16out"): 24, ("385024", "f32out"): 48}
    tot_b, tot_n = 0.0, 0
    for key, n in weights.items():
        c = per_shape.get(key)
        if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            return None
        tot_b += n * (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        tot_n += n
    return tot_b / tot_n


SELFCHECK_TOL_S = 0.02   # GPU (16-bit) vs fp32 oracle onset MAE over the checked clips; one frame = 0.02 s

# The pipeline shape of the headline run.  tests/test_gpu_headline.py reads these (and the driver's --warmup / --steps) and holds
# exactly this shape to the oracle, so a change of a default here cannot escape the parity tests.
DEFAULT_HEAD_GROUP = 4        # batches whose head (GRU / FC / DP) runs as one launch set: 128 clips
DEFAULT_ENCODER_STREAMS = 2   # consecutive batches' encoders alternate between two HIP streams (profiles/r4_sweep_pipeline_shape.txt)
DEFAULT_STEPS, DEFAULT_WARMUP = 20, 2
DRIVER_STEPS, DRIVER_WARMUP = 20, 5      # the round-end run: `bench.py --gpus 1 --steps 20 --warmup 5` (BENCH_rNN.json "cmd")
ROOFLINE_PASS_STEPS = 12      # steps of the single-encoder-stream pass after the timed region that the roofline leg is measured on


N_TIMBRES = 40            # distinct "syllables" of the synthetic songs: a spectral envelope and a class id each
HEAD_DISTRACTOR_SCALE, HEAD_DISTRACTOR_BIAS, HEAD_SILENCE_BIAS, HEAD_TARGET = 3.0, -14.0, -8.0, 8.0


def timbre_bank():
    """-> (envelopes [K, 80] float64 in [0, 1]: three triangular formant bumps over the mel channels; class ids [K] int64,
    distinct, in the label columns 2 .. V-3 of the output Linear).  MT19937 and + - * / abs max only: the same bits everywhere."""
    rs = np.random.RandomState(77)
    ch = np.arange(80, dtype=np.float64)
    env = np.zeros((N_TIMBRES, 80))
    for k in range(N_TIMBRES):
        for _ in range(3):
            c, w, a = rs.uniform(2.0, 78.0), rs.uniform(3.0, 14.0), rs.uniform(0.5, 1.0)
            env[k] = np.maximum(env[k], a * np.maximum(0.0, 1.0 - np.abs(ch - c) / w))
    ids = np.sort(rs.choice(np.arange(2, VOCAB - 2), size=N_TIMBRES, replace=False)).astype(np.int64)
    return env, ids


def note_plan(n_notes, n_frames: int = 3000, seed: int = 2):
    """Per clip (edges [L+1], timbre [L]): the mel-frame boundaries of L back-to-back notes filling the clip -- durations
    proportional to MT19937 uniforms in [0.5, 1.5), one IEEE division and floor per boundary -- and the timbre sung on each
    note, never the previous note's (a repeated syllable has no boundary to find)."""
    rs = np.random.RandomState(seed + 1000)
    plans = []
    for L in n_notes:
        w = np.cumsum(rs.uniform(0.5, 1.5, size=int(L)))
        edges = np.concatenate([[0], np.floor(w / w[-1] * n_frames).astype(np.int64)])
        edges[-1] = n_frames
        timbre, prev = [], -1
        for _ in range(int(L)):
            k = int(rs.randint(N_TIMBRES)) if prev < 0 else (prev + 1 + int(rs.randint(N_TIMBRES - 1))) % N_TIMBRES
            timbre.append(k)
            prev = k
        plans.append((edges, np.array(timbre, dtype=np.int64)))
    return plans


def synthetic_mel(plans, n_frames: int = 3000, seed: int = 2) -> np.ndarray:
    """Synthetic log-mel batch [len(plans), 80, n_frames] float32 in [-1, 1] with the structure of a sung line: every clip is
    the note sequence of its plan, each note the spectral envelope of its timbre (timbre_bank) at a loudness of 0.8 .. 1.0,
    plus 10 % uniform noise.  (Uniform noise alone -- rounds 1-2 -- is featureless: the encoder's output then varies with the
    position only and the lattice has nothing to align to.)
    Only MT19937 integers / uniforms and + - * / abs max floor are used, so the batch is the same bits on every host
    (no libm / SIMD transcendental whose last place depends on the CPU)."""
    rs = np.random.RandomState(seed)
    env, _ = timbre_bank()
    mel = np.empty((len(plans), 80, n_frames), dtype=np.float32)
    for b, (edges, timbre) in enumerate(plans):
        for t0, t1, k in zip(edges[:-1], edges[1:], timbre):
            n = int(t1 - t0)
            loud = rs.uniform(0.8, 1.0)
            noise = rs.uniform(-1.0, 1.0, size=(80, n))
            mel[b, :, int(t0):int(t1)] = (-0.8 + 1.6 * loud * env[k][:, None] + 0.1 * noise).astype(np.float32)
    return np.clip(mel, -1.0, 1.0)


def synthetic_waveform(plans, device, n_frames: int = 3000, seed: int = 2) -> torch.Tensor:
    """--from-waveform: the same songs as synthetic_mel, as 16 kHz audio [len(plans), n_frames * 160] float32 on the device: every
    note is a chord of 80 partials at the mel channels' centre frequencies whose amplitudes follow the note's timbre envelope over
    40 dB (10^(2 env - 2), loudness 0.8 .. 1.0, random phases) plus -60 dB noise, so its log-mel -- computed by the device kernel
    INSIDE the timed step -- has the note structure the transcript names.  Setup only (torch on the device, outside the timing)."""
    from lyricalignment_amd.audio_frontend import _slaney_hz_to_mel, _slaney_mel_to_hz
    rs = np.random.RandomState(seed + 5000)
    env, _ = timbre_bank()
    edges_hz = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(0.0), _slaney_hz_to_mel(8000.0), 82))
    omega = torch.from_numpy(2.0 * np.pi * edges_hz[1:-1] / 16000.0).to(device=device, dtype=torch.float32)[:, None]      # [80, 1]
    wave = torch.empty((len(plans), n_frames * 160), dtype=torch.float32, device=device)
    for b, (edges, timbre) in enumerate(plans):
        for t0, t1, k in zip(edges[:-1], edges[1:], timbre):
            n = int(t1 - t0) * 160
            amp = torch.from_numpy(10.0 ** (2.0 * env[k] - 2.0) * rs.uniform(0.8, 1.0) / 8.0).to(device=device, dtype=torch.float32)[:, None]
            ph = torch.from_numpy(rs.uniform(0.0, 2.0 * np.pi, size=(80, 1))).to(device=device, dtype=torch.float32)
            t = torch.arange(n, device=device, dtype=torch.float32)[None, :]
            wave[b, int(t0) * 160: int(t1) * 160] = (amp * torch.sin(omega * t + ph)).sum(dim=0)
    g = torch.Generator(device="cpu").manual_seed(seed + 6000)
    wave += (torch.rand(wave.shape, generator=g) * 2e-3 - 1e-3).to(device)
    return wave


def build_inputs(device, seed_offset: int = 0, from_waveform: bool = False):
    """-> (mel [32,80,3000] f32, labels [32,Lmax] i32, n_labels [32] i32 on the device, Ls host, plans)
    (from_waveform: mel is the WAVEFORM [32, 480000] f32 of the same songs instead -- the step then starts at the log-mel).
    Every clip is a line of L = 5..26 sung notes (note_plan) and its transcript is the class id of each note's timbre: what
    was sung, as the reference's datasets give it.  (Rounds 1-2 used random ids unrelated to the audio: near-tied lattices
    whose boundaries no two precisions agree on -- profiles/r3_selfcheck_diagnosis.md.)"""
    Ls = np.random.RandomState(3 + seed_offset).randint(5, 27, size=BATCH)
    plans = note_plan(Ls, 3000, 2 + seed_offset)
    if from_waveform:
        mel = synthetic_waveform(plans, device, 3000, 2 + seed_offset)
    else:
        mel = torch.from_numpy(synthetic_mel(plans, 3000, 2 + seed_offset)).to(device)
    _, ids = timbre_bank()
    labels = np.zeros((BATCH, int(Ls.max())), dtype=np.int32)
    for b, (_, timbre) in enumerate(plans):
        labels[b, :len(timbre)] = ids[timbre]
    return mel, torch.from_numpy(labels).to(device), torch.from_numpy(Ls.astype(np.int32)).to(device), Ls, plans


def fit_head(model, device, fit_seed_offset: int = 100, ridge: float = 1e-3, from_waveform: bool = False) -> dict:
    """Give the synthetic AlignModel a head that has "learnt" the synthetic songs, the way a fine-tuned checkpoint has learnt
    its corpus: a linear probe.  A batch of OTHER songs (seed fit_seed_offset; same timbres, other melodies and noise) goes
    through the random-init encoder and BiGRU on the device; the output Linear's rows of the N_TIMBRES syllable classes are the
    ridge-regression solution that maps Mish(GRU) of a frame to +HEAD_TARGET for the syllable sung there and -HEAD_TARGET for
    the others (normal equations accumulated in float64 on the device, solved on the host).  The ~21 k other label columns
    keep host-independent random rows (scale HEAD_DISTRACTOR_SCALE) under a bias of HEAD_DISTRACTOR_BIAS -- the characters a
    trained model gives no mass to -- and the silence column says "voiced" (HEAD_SILENCE_BIAS; the songs have no rests).
    The posteriors that result are as decided as a trained model's on its own data, so the lattice's best path is fixed by
    margins of many nats per frame: what the bench's self-check and tests/test_gpu_headline.py compare between the 16-bit
    device path and the fp32 oracle is then the pipeline, not a lottery of near-ties.  (The frame-level shapes, the FLOPs
    and the operand statistics of every kernel are what they were: the FC is still a [48000, 768] x [768, 21129] GEMM on
    random-valued rows.)  -> {"fit_frame_accuracy": share of fit frames whose top label column is the sung one}."""
    from lyricalignment_amd import whisper_compat as wc
    wc.init_align_head(model, seed=7, fc_scale=HEAD_DISTRACTOR_SCALE)
    fc = model.align_rnn.fc
    with torch.no_grad():
        fc.bias.fill_(HEAD_DISTRACTOR_BIAS)
        fc.weight[-1].zero_()
        fc.bias[-1] = HEAD_SILENCE_BIAS
        eng = model.engine()
        mel, _, _, Ls, plans = build_inputs(device, seed_offset=fit_seed_offset, from_waveform=from_waveform)
        if from_waveform:
            from lyricalignment_amd.audio_frontend import log_mel_spectrogram
            mel = log_mel_spectrogram(mel, device=device)
        X = eng.head_hidden(eng.encode(mel), BATCH, T_FRAMES, T_FRAMES).double()          # [32*1500, 2H]
        sung = np.empty((BATCH, T_FRAMES), dtype=np.int64)
        for b, (edges, timbre) in enumerate(plans):
            sung[b] = timbre[np.searchsorted(edges, 2 * np.arange(T_FRAMES) + 1, side="right") - 1]   # encoder frame f = mel frames 2f, 2f+1
        sung_d = torch.from_numpy(sung.reshape(-1)).to(device)
        Xa = torch.cat([X, torch.ones((X.shape[0], 1), dtype=torch.float64, device=device)], dim=1)
        Y = torch.full((X.shape[0], N_TIMBRES), -HEAD_TARGET, dtype=torch.float64, device=device)
        Y[torch.arange(X.shape[0], device=device), sung_d] = HEAD_TARGET
        G = (Xa.T @ Xa).cpu().numpy()
        R = (Xa.T @ Y).cpu().numpy()
        G[np.arange(G.shape[0] - 1), np.arange(G.shape[0] - 1)] += ridge * X.shape[0]       # the bias row is not penalised
        W = torch.from_numpy(np.linalg.solve(G, R)).to(device)                              # [2H+1, K]
        acc = float(((Xa @ W).argmax(dim=1) == sung_d).double().mean())
        _, ids = timbre_bank()
        ids_t = torch.from_numpy(ids).to(fc.weight.device)
        fc.weight[ids_t] = W[:-1].T.to(fc.weight.dtype).to(fc.weight.device)
        fc.bias[ids_t] = W[-1].to(fc.bias.dtype).to(fc.bias.device)
        del X, Xa, Y, mel
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return {"fit_frame_accuracy": acc, "fit_clips": BATCH, "probe_weight_absmax": float(W[:-1].abs().max())}


class PowerSampler:
    """Socket power and shader clock of one GPU read from its hwmon nodes (power1_input in microwatts, freq1_input in hertz;
    plain sysfs reads, nothing is set) every 50 ms by a thread while the timed region runs -- the roofline's `peak` is the
    datasheet rate at the nominal 2.4 GHz, and this says what clock the package actually held under its power cap.
    Silent when the nodes are not there (-> None in the JSON)."""

    def __init__(self, device_index: int):
        self.samples, self.cap_w, self._stop, self._thread, self._dir = [], None, False, None, None
        try:
            import glob
            want = None
            try:
                pr = torch.cuda.get_device_properties(device_index)
                want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            except Exception:
                pass
            cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
            cands = [c for c in cands if os.path.exists(os.path.join(c, "power1_input")) and os.path.exists(os.path.join(c, "freq1_input"))]
            if want is not None:
                hit = [c for c in cands if want in os.path.realpath(os.path.dirname(os.path.dirname(c)))]
                cands = hit or (cands if len(cands) == 1 else [])
            if cands:
                self._dir = cands[0]
                with open(os.path.join(self._dir, "power1_cap")) as f:
                    self.cap_w = int(f.read()) / 1e6
        except Exception:
            self._dir = None

    def _read(self):
        with open(os.path.join(self._dir, "power1_input")) as f:
            w = int(f.read()) / 1e6
        with open(os.path.join(self._dir, "freq1_input")) as f:
            mhz = int(f.read()) / 1e6
        return w, mhz

    def _run(self):
        while not self._stop:
            try:
                self.samples.append(self._read())
            except Exception:
                return
            time.sleep(0.05)

    def start(self):
        if self._dir is not None:
            import threading
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)
        if not self.samples:
            return None
        w = sorted(s[0] for s in self.samples)
        f = sorted(s[1] for s in self.samples)
        return {"socket_w_median": w[len(w) // 2], "socket_w_max": w[-1], "cap_w": self.cap_w, "sclk_mhz_median": f[len(f) // 2],
                "sclk_mhz_min": f[0], "samples": len(w), "source": "hwmon power1_input / freq1_input every 50 ms through the timed region"}


def usable_cores() -> int:
    """Cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (a container on a
    256-thread host often owns only a few cores; oversubscribing torch's intra-op pool makes it crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def log(msg: str):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def _cpu_stage_times(p, n_head: int, audio: np.ndarray, label: np.ndarray) -> dict:
    """One clip through the CPU oracle stage by stage (SURVEY 8(d) / BASELINE.md section 3 item 1): waveform -> log-mel + pad_or_trim
    (module/align_model.py:84-90) -> encoder (:91) -> BiGRU / Mish / Linear head (:107) -> emission prep (utils/alignment.py:123-134)
    -> DP + backtrace (compiled oracle).  Milliseconds, one pass each (the caller has warmed the weights)."""
    from oracle import alignment_oracle as ao
    from oracle import model_oracle as mo
    out = {}
    with torch.no_grad():
        t = time.perf_counter(); mel = mo.pad_or_trim(mo.log_mel_spectrogram(audio[None]), 3000); out["mel"] = (time.perf_counter() - t) * 1e3
        T = mo.frame_count(len(audio) // 160)
        t = time.perf_counter(); emb = mo.encoder_forward(p, mel, n_head=n_head)[:, :T]; out["encoder"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter(); logits = mo.gru_head_forward(p, emb); out["head"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter(); lp, ls = mo.emission_prep_ctc(logits); out["emission_prep"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter(); rc = ao.align_frames(lp[0].numpy(), ls[0].numpy(), label)[0]; out["dp"] = (time.perf_counter() - t) * 1e3
    assert rc == 0
    out["frames"] = int(T)
    return {k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()}, lp[0].numpy(), ls[0].numpy()


def cpu_stage_baseline(model, n_head: int) -> dict:
    """cpu_baseline.stages_ms: the reference's CPU path stage by stage on this node's host cores, one clip (B = 1) each of
      * whisper-medium, 30 s          (the headline configuration's model),
      * whisper-tiny, 3.756 s and 30 s (BASELINE configs[0]: the reference's own CPU-runnable case; 3.756 s = the first Opencpop
                                        test utterance's length),
    and python_dp_s: the lattice recurrence as plain Python loops (oracle/viterbi_python.py: what the reference's run_viterbi_core
    costs without numba, utils/alignment.py:73-119) on the medium clip's [1500, 21127] emissions -- the DP's upper bound.
    Bounded: ~2 s (medium) + ~1 s (tiny) + ~1 s (Python DP) of CPU work after the weights are warm."""
    from lyricalignment_amd import whisper_compat as wc
    from oracle import model_oracle as mo
    from oracle import viterbi_python as vp
    rs = np.random.RandomState(11)
    t30 = np.arange(480000) / 16000.0
    audio30 = (rs.randn(480000) * 0.05 + 0.3 * np.sin(2 * np.pi * 220 * t30)).astype(np.float32)
    label = make_stage_labels(26)
    stages = {}
    pm = {"encoder." + k: v.detach().float().cpu() for k, v in model.whisper_model.encoder.state_dict().items()}
    pm.update({"align_rnn." + k: v.detach().float().cpu() for k, v in model.align_rnn.state_dict().items()})
    stages["medium_30s"], lp, ls = _cpu_stage_times(pm, n_head, audio30, label)
    t = time.perf_counter()
    vp.viterbi_lattice(lp, ls, label)
    python_dp_s = time.perf_counter() - t
    del pm
    dims = wc.dims_for("tiny")
    tiny = wc.build_model("tiny", seed=0)
    pt = {"encoder." + k: v.detach().float().cpu() for k, v in tiny.encoder.state_dict().items()}
    pt.update(mo.random_head_params(dims.n_audio_state, HIDDEN, VOCAB, seed=1))
    with torch.no_grad():                                            # warm-up pass (first touch of the tiny weights)
        mo.gru_head_forward(pt, mo.encoder_forward(pt, mo.pad_or_trim(mo.log_mel_spectrogram(audio30[None, :60096]), 3000), n_head=dims.n_audio_head)[:, :10])
    stages["tiny_3.756s"] = _cpu_stage_times(pt, dims.n_audio_head, audio30[:60096], make_stage_labels(11))[0]
    stages["tiny_30s"] = _cpu_stage_times(pt, dims.n_audio_head, audio30, label)[0]
    return {"stages_ms": stages, "python_dp_s": round(python_dp_s, 3),
            "stages_note": "one clip (B = 1) per entry through the fp32 CPU oracle, ms per stage: mel = log-mel + pad_or_trim from the "
                           "waveform, encoder, head = BiGRU + Mish + Linear(21129), emission_prep = log-softmax / logsigmoid over the "
                           "vocabulary, dp = compiled Viterbi + backtrace (26 / 11 labels); python_dp_s = the same lattice of the medium "
                           "clip in plain Python loops (the reference's DP without numba)"}


def make_stage_labels(n: int) -> np.ndarray:
    lab = np.random.RandomState(12).randint(2, VOCAB - 2, size=n).astype(np.int64)
    for i in range(1, n):
        if lab[i] == lab[i - 1]:
            lab[i] = 2 + (lab[i] - 1) % (VOCAB - 4)
    return lab


def cpu_baseline(model, mel_cpu: np.ndarray, labels_rows, n_head: int):
    """The oracle (CPU restatement of the reference's fp32 path) timed on this node's host cores on a bounded sample:
    30 s clips of the same workload (same weights), one clip per pass: 1 warm-up pass + 2 timed passes = the first three
    clips of the batch, so the same ~8 s of CPU work also give the self-check three clips to compare.  Plus the per-stage
    breakdown (cpu_stage_baseline).
    -> (cpu_baseline object, [per-clip oracle result])."""
    from oracle import alignment_oracle as ao
    from oracle import model_oracle as mo
    ao.build()
    torch.set_num_threads(usable_cores())
    p = {"encoder." + k: v.detach().float().cpu() for k, v in model.whisper_model.encoder.state_dict().items()}
    p.update({"align_rnn." + k: v.detach().float().cpu() for k, v in model.align_rnn.state_dict().items()})

    def one(i):
        with torch.no_grad():
            emb = mo.encoder_forward(p, torch.from_numpy(mel_cpu[i][None]), n_head=n_head)
            logits = mo.gru_head_forward(p, emb)
            return ao.perform_viterbi_ctc(logits, torch.from_numpy(labels_rows[i][None].astype(np.int64)))[0]

    t0 = time.perf_counter()
    results = [one(0)]                # warm-up (also the first-touch of the weights)
    warm = time.perf_counter() - t0
    log(f"cpu baseline warm-up pass {warm:.1f} s on {torch.get_num_threads()} threads")
    times = []
    for i in range(1, 3 if warm < 20 else 2):   # keep the CPU leg bounded
        t0 = time.perf_counter()
        results.append(one(i))
        times.append(time.perf_counter() - t0)
    sec = float(np.median(times))
    base = {"value": CLIP_SECONDS / sec, "unit": "audio-sec/sec", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{len(times)} x 30 s clips (clips 1..{len(times)} of the batch; clip 0 = warm-up), whisper-{MODEL} fp32 oracle incl. CPU "
                      f"emission prep + C Viterbi, median {sec:.2f} s per clip"}
    del p
    if warm < 20:                               # (a host this slow would not finish the stage passes in bounded time)
        t0 = time.perf_counter()
        base.update(cpu_stage_baseline(model, n_head))
        log(f"cpu stage breakdown {time.perf_counter() - t0:.1f} s")
    return base, results


def selfcheck(gpu_onset: np.ndarray, gpu_offset: np.ndarray, cpu_results, Ls, plans=None) -> dict:
    """Boundaries of the timed GPU path (last batch of the timed region, 16-bit operands) against the fp32 oracle's own
    end-to-end result on the same clips (utils/alignment.py:121-188 semantics: seconds = frame * 0.02); with the clips' note
    plans also both against the note edges the songs were synthesised with (onset of note i = edges[i] * 0.01 s)."""
    per_clip, on_err, off_err, g_true, c_true = [], [], [], [], []
    for i, res in enumerate(cpu_results):
        L = int(Ls[i])
        c_on, c_off = np.array([seg[0] for seg in res]), np.array([seg[1] for seg in res])
        g_on, g_off = gpu_onset[i, :L] * 0.02, gpu_offset[i, :L] * 0.02
        on_err.append(np.abs(g_on - c_on)); off_err.append(np.abs(g_off - c_off))
        per_clip.append({"clip": i, "labels": L, "onset_mae_s": float(on_err[-1].mean()), "offset_mae_s": float(off_err[-1].mean()),
                         "boundaries_equal": int((np.abs(g_on - c_on) < 1e-9).sum() + (np.abs(g_off - c_off) < 1e-9).sum()),
                         "boundaries": 2 * L})
        if plans is not None:
            true_on = plans[i][0][:-1] * 0.01
            g_true.append(np.abs(g_on - true_on)); c_true.append(np.abs(c_on - true_on))
    on_all, off_all = np.concatenate(on_err), np.concatenate(off_err)
    out = {"clips": len(cpu_results), "onset_mae_s": float(on_all.mean()), "offset_mae_s": float(off_all.mean()),
           "max_dev_s": float(max(on_all.max(), off_all.max())), "tol_s": SELFCHECK_TOL_S, "per_clip": per_clip}
    if plans is not None:
        out["gpu_onset_vs_note_edges_mae_s"] = float(np.concatenate(g_true).mean())
        out["cpu_onset_vs_note_edges_mae_s"] = float(np.concatenate(c_true).mean())
    return out


def run_finetune(args, rank, world, local_rank, device, dist):
    """-> the bench line of BASELINE configs[2] as a dict (None on ranks other than 0).
    BASELINE configs[2]: multitask fine-tune (CE + silence BCE + CTC on the align logits, decoder CE), float32 like the
    reference, data parallel: every rank runs `accum` micro-steps of 2 x 30 s clips (reference defaults, train_multitask.py:
    240,325), then ONE all-reduce (sum) per flat gradient bucket over RCCL / xGMI, the fused clip + AdamW (:337-340).
    A step = one optimizer step; value = audio-seconds the whole job trained on per wall-second."""
    from lyricalignment_amd import _lib, finetune as ft, whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    _lib.require_gpu()
    log(f"rank {rank}/{world}: building random-init whisper-{args.model} weights (fine-tune mode)")
    dims = wc.dims_for(args.model)
    t_start = time.perf_counter()
    wm = build_weights(args.model, True, rank, world, dist)
    log(f"rank {rank}: weights built in {time.perf_counter() - t_start:.1f} s")
    torch.manual_seed(0)                       # the head's torch-default initialisation: identical on every rank
    model = AlignModel(wm, embed_dim=dims.n_audio_state, hidden_dim=HIDDEN, output_dim=VOCAB, dropout=0.15, train_transcript=True,
                       device=f"cuda:{local_rank}").to(device)
    tuner = ft.FineTuner(model, warmup_steps=1, train_steps=10000, allreduce_chunks=args.allreduce_chunks)
    B, n_tok = 2, 32
    rs = np.random.RandomState(114514 + rank)   # reference seed (train_multitask.py:136-139) + rank: every rank its own clips
    audios = [(rs.randn(480000) * 0.1).astype(np.float32) for _ in range(B)]
    labels = torch.from_numpy(rs.randint(1, 402, size=(B, 26)))
    fl = torch.full((B, T_FRAMES), -100, dtype=torch.long)
    for b in range(B):
        for i in range(26):                     # frame labels from uniformly spaced on / offsets (SURVEY 8d cfg 3)
            fl[b, 40 + 50 * i: 40 + 50 * i + 30] = labels[b, i]
    dec_in = torch.from_numpy(rs.randint(0, 50000, size=(B, n_tok)))
    dec_out = torch.from_numpy(rs.randint(0, 50000, size=(B, n_tok)))
    ar_events = []

    micro = dict(audios=audios, ctc_labels=labels, frame_labels=fl, decoder_input=dec_in, decoder_output=dec_out)

    def step(timed):
        # the accumulation loop of train_step: fused = ONE forward / backward over the accum x 2 clips, per-micro-batch losses
        # (FineTuner.accumulate; --accum-mode loop = accum separate micro-steps, the round-1/2 form)
        # the exchange step of the data-parallel path rides on the last backward (finetune.OverlappedAllReduce: chunks of the flat
        # buckets are all-reduced as autograd completes them); step() waits for it -- what is left exposed is bracketed there
        tuner.accumulate([micro] * args.accum, accum_grad_steps=args.accum, fused=args.accum_mode == "fused")
        if tuner.overlap is None:               # --allreduce-chunks 0: one blocking all-reduce per bucket, bracketed here
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ft.allreduce_mean_(tuner.grad, tuner.world)
            e1.record()
            if timed:
                ar_events.append((e0, e1))
            tuner.step(allreduced=True)
            return
        tuner.step()
        if timed and getattr(tuner.overlap, "_events", None) is not None:
            ar_events.append(tuner.overlap._events)

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    L = _lib.lib()
    L.la_timer_reset()
    L.la_timer_sample(args.timer_period)      # (every launch bracketed = two barrier packets each: ~40 ms of a 0.8 s step)
    from lyricalignment_amd import f32x2
    x2 = f32x2.ENABLED          # the large Linear products run as three f16 products over split operands (csrc/la_f32x2.hip)
    L.la_timer_enable(os.environ.get("LA_BENCH_TIMER", "gemm_f16x2" if x2 else "gemm_f32").encode())
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    L.la_timer_disable()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    import ctypes
    total_ms, launches, work, seen = ctypes.c_double(0.0), ctypes.c_int64(0), ctypes.c_double(0.0), ctypes.c_int64(0)
    _lib.check(L.la_timer_read_work(ctypes.byref(total_ms), ctypes.byref(launches), ctypes.byref(work), ctypes.byref(seen)), "timer_read_work")
    d, nl = dims.n_audio_state, dims.n_audio_layer
    T = T_FRAMES
    # algorithmic GEMM flops of one micro-step (forward + the two backward products of every Linear = 3 x forward):
    # encoder Linears / convs, GRU input projections + FC, decoder (cross-attention K/V projections over 1500 frames dominate)
    fwd = (2 * 3000 * 80 * 3 * d + 2 * T * d * 3 * d + nl * 24 * T * d * d) + (2 * (2 * T * d * 3 * HIDDEN) + 2 * (2 * T * 2 * HIDDEN * 3 * HIDDEN)
           + 2 * T * 2 * HIDDEN * VOCAB) + dims.n_text_layer * (2 * T * d * 2 * d + n_tok * 22 * d * d) + 2 * n_tok * d * 51865
    gemm_flops_step = 3.0 * fwd * B * args.accum          # (analytic count of the Linear layers, for reference: "linear_gflop_per_step")
    # achieved = 2 M N K of the bracketed float32 GEMM launches (summed by the library) / their HIP-event durations
    achieved = work.value / (total_ms.value * 1e-3) / 1e12 if total_ms.value > 0 else 0.0
    ar_ms = float(np.mean([a.elapsed_time(b) for a, b in ar_events])) if ar_events else 0.0
    grad_bytes = int(sum(g.numel() for g in tuner.grad) * 4)
    line = None
    if rank == 0:
        audio_sec = world * B * CLIP_SECONDS * args.accum * args.steps
        line = ({
            "metric": f"fine-tuned audio-sec/sec, Whisper-{args.model} multitask (CTC + CE + decoder CE), DP={world}",
            "value": audio_sec / elapsed, "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (f16x2 split of the large Linear and attention products: 3 f16 MFMA products per float32 product, f32 accumulate)" if x2 else "f32",
            "data": "synthetic (Gaussian waveforms, random class-id / frame / token labels, random-init weights)",
            "config": {"workload": f"whisper-{args.model} multitask fine-tune step, per-GPU micro-batch {B} x 30 s x accum {args.accum} "
                                   "(BASELINE.json configs[2])", "mode": "finetune", "micro_batch": B, "accum": args.accum, "accum_mode": args.accum_mode,
                       "decoder_tokens": n_tok, "grad_bytes_per_step": grad_bytes,
                       "sharding": f"clips over ranks; the flat gradient buckets all-reduced (sum) once per optimizer step in >= {args.allreduce_chunks} "
                                   "chunks, each as soon as the last backward has completed it (overlapped with the rest of that backward)"},
            "micro_step_ms": (elapsed / args.steps * 1e3 - ar_ms) / args.accum,
            "allreduce_ms_per_step": ar_ms,
            "allreduce_exposed_ms": ar_ms,      # what the exchange costs on the compute stream after the overlap (0 at N = 1)
            # bus bandwidth of the ring all-reduce (only meaningful for the blocking form, --allreduce-chunks 0)
            "allreduce_GBps": (2.0 * (world - 1) / world * grad_bytes / (ar_ms * 1e-3) / 1e9) if (world > 1 and ar_ms > 0 and args.allreduce_chunks == 0) else None,
            "roofline": ({"bound": "mfma", "kernel": "gemm_pp_kernel<f16, segmented K> (la_gemm_f16x2: the large Linear forward + both backward "
                                                    "products as a_lo w_hi + a_hi w_lo + a_hi w_hi in v_mfma_f32_16x16x32_f16)",
                          # achieved = ALGORITHMIC float32 flops (2 M N K) of the bracketed launches / their HIP-event durations; the kernel
                          # executes three times that in f16 MFMAs (mfma_executed_tflops), priced against the dense f16 peak
                          "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0, "traffic": None,
                          "mfma_executed_tflops": 3.0 * achieved, "frac_executed": 3.0 * achieved / 2500.0,
                          "vs_f32_mfma_peak": achieved / 157.3,
                          "launches_per_step": seen.value / max(args.steps, 1), "avg_launch_ms": total_ms.value / max(launches.value, 1),
                          "timed_launches": launches.value, "linear_gflop_per_step": gemm_flops_step / 1e9} if x2 else
                         {"bound": "mfma", "kernel": "gemm_kernel<float> (v_mfma_f32_16x16x4_f32; every Linear forward + both backward products)",
                          "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s", "frac": achieved / 157.3, "traffic": None,
                          "launches_per_step": seen.value / max(args.steps, 1), "avg_launch_ms": total_ms.value / max(launches.value, 1),
                          "timed_launches": launches.value, "linear_gflop_per_step": gemm_flops_step / 1e9}),
            "cpu_baseline": None,
            "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9})
    del tuner, model
    return line


def finetune_mode(args, rank, world, local_rank, device, dist):
    line = run_finetune(args, rank, world, local_rank, device, dist)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_other_config(args, rank, world, local_rank, device, dist, model=None):
    """-> the bench line of BASELINE configs[3] / configs[4] as a dict (None on ranks other than 0); `model`: a ready AlignModel of the
    mode's architecture and dtype (the in-process legs of the default run hand over the headline's).
    BASELINE configs[3] / configs[4] through the same harness (their records under profiles/ come from tools/large_v2_fill.py and
    tools/longform_bench.py; this prints them in the bench line's format, clips / songs sharded over ranks with no collective).
    largev2: Whisper-large-v2, float16, --clips x 30 s per GPU and step, single stream (the batch is the parallelism).
    longform: --songs x 180 s per GPU and step (6 x 30 s chunks per song through the encoder as one song-major batch, GRU + DP over
    T = 9000 frames per song), consecutive steps pipelined over two streams with head_group 2."""
    from lyricalignment_amd import _lib, whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    _lib.require_gpu()
    large = args.mode == "largev2"
    name = "large-v2" if large else MODEL
    log(f"rank {rank}/{world}: building random-init whisper-{name} weights ({args.mode} mode)")
    dims = wc.dims_for(name)
    dt = torch.float16 if large else torch.bfloat16
    if model is None:
        model = AlignModel(build_weights(name, False, rank, world, dist), embed_dim=dims.n_audio_state, hidden_dim=HIDDEN, output_dim=VOCAB,
                           device=f"cuda:{local_rank}", compute_dtype=dt).eval()
    with torch.no_grad():
        eng = model.engine()
    rs = np.random.RandomState(2)
    if large:
        units, unit_s, L = args.clips, CLIP_SECONDS, 26
        mel = torch.from_numpy(rs.uniform(-1, 1, size=(units, 80, 3000)).astype(np.float32)).to(device)
    else:
        units, unit_s, L = args.songs, 180.0, 238
        mel = torch.from_numpy(rs.uniform(-1, 1, size=(units, 80, 18000)).astype(np.float32)).to(device)
    labels = torch.from_numpy(rs.randint(2, 402, size=(units, L)).astype(np.int32)).to(device)
    n_labels = torch.full((units,), L, dtype=torch.int32, device=device)
    pipe = None
    if not large:
        from lyricalignment_amd.engine import PipelinedAligner
        pipe = PipelinedAligner(eng, head_group=2)
    last = {}

    def step():
        with torch.no_grad():
            if pipe is not None:
                last["out"] = pipe.submit_songs(mel, labels, n_labels)
            else:
                last["out"] = eng.align_mel(mel, labels, n_labels, n_frames=T_FRAMES, use_ctc=True)

    def finish():
        if pipe is not None:
            pipe.drain()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    finish()
    eng.check_gru()
    L_ = _lib.lib()
    L_.la_timer_reset()
    L_.la_timer_sample(1)                      # every launch of the family bracketed: total_ms below is the family's whole time
    L_.la_timer_enable(os.environ.get("LA_BENCH_TIMER", "gemm_bf16").encode())
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    finish()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    L_.la_timer_disable()
    eng.check_gru()
    if int((last["out"][3] != 0).sum()) != 0:
        raise SystemExit("alignment reported non-OK status on the synthetic batch")
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    import ctypes
    total_ms, launches = ctypes.c_double(0.0), ctypes.c_int64(0)
    _lib.check(L_.la_timer_read(ctypes.byref(total_ms), ctypes.byref(launches)), "timer_read")
    L_.la_timer_reset()
    clips30 = units if large else units * 6
    gemm_flops_step = algorithmic_gemm_flops_per_clip(dims.n_audio_state, dims.n_audio_layer) * clips30
    achieved = gemm_flops_step * args.steps / (total_ms.value * 1e-3) / 1e12 if total_ms.value > 0 else 0.0
    line = None
    if rank == 0:
        line = ({
            "metric": f"aligned audio-sec/sec (RTF^-1), Whisper-{name} " + ("30 s clips" if large else "3-minute songs"),
            "value": world * units * unit_s * args.steps / elapsed, "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16" if large else "bf16",
            "data": "synthetic (uniform[-1,1] mel, random class-id labels, random-init weights of the architecture)",
            "config": {"workload": (f"whisper-large-v2 align, {units} x 30 s clips per GPU and step, float16, single stream (BASELINE.json configs[3])" if large else
                                    f"whisper-medium long-form align, {units} songs x 180 s per GPU and step (6 chunks each, T = 9000, {L} labels), "
                                    "two-stream pipeline (BASELINE.json configs[4])"),
                       "mode": args.mode, "units_per_gpu": units, "sharding": "clips / songs over ranks, no collective"},
            "roofline": {"bound": "mfma", "kernel": "gemm_pp_kernel / gemm_kernel <16-bit>", "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": achieved / 2500.0, "traffic": None, "launches_per_step": launches.value / max(args.steps, 1),
                         "avg_launch_ms": total_ms.value / max(launches.value, 1)},
            "cpu_baseline": None, "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9})
    return line


def other_config_mode(args, rank, world, local_rank, device, dist):
    line = run_other_config(args, rank, world, local_rank, device, dist)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ---- the default run's extra legs (N = 1): the other compute modes of the headline workload and the other BASELINE configs, measured on
# short passes AFTER the timed region and the roofline pass.  None of them touches the headline keys or the exit code: a leg that fails
# reports {"error": ...} in its place.
EXTRA_LEGS_BUDGET_S = 150.0


def mode_legs(wm, head_state, dims, device, local_rank, mel, labels, n_labels, head_group, encoder_streams, ref_frames, steps: int = 6) -> dict:
    """The headline workload (same weights, same batch, same pipeline shape) in the two other compute modes:
      f32_parity  float32 like the reference end to end (module/align_model.py:72-123) -- the one mode in which the seconds equal the
                  oracle's -- on the f16 matrix pipe at float32 accuracy (three f16 MFMA products per float32 product, csrc/la_f32x2.hip);
      f16         float16 operands (same MFMA rate as bfloat16, 3 more mantissa bits).
    Per mode: ms per step and audio-s/s over `steps` pipelined steps, the GEMM family's achieved TFLOP/s from the library's sampled
    HIP-event timer on that pass, and how many of the batch's onset / offset frames equal the float32 mode's (the precision side of the trade;
    `ref_frames` = the bf16 headline's frames, compared the same way)."""
    from lyricalignment_amd import _lib
    from lyricalignment_amd.engine import PipelinedAligner
    from lyricalignment_amd.module.align_model import AlignModel
    import ctypes
    L = _lib.lib()
    out, frames = {}, {}
    alg = algorithmic_gemm_flops_per_clip(dims.n_audio_state, dims.n_audio_layer)
    for name, dt, family, note in (("f32_parity", torch.float32, "gemm_f16x2", "float32 in / out, every large product as a_lo w_hi + a_hi w_lo + a_hi w_hi in f16 MFMAs"),
                                   ("f16", torch.float16, "gemm_bf16", "float16 operands, f32 accumulate")):
        try:
            m = AlignModel(wm, embed_dim=dims.n_audio_state, hidden_dim=HIDDEN, output_dim=VOCAB, device=f"cuda:{local_rank}", compute_dtype=dt).eval()
            m.align_rnn.load_state_dict(head_state)
            with torch.no_grad():
                eng = m.engine()
            pipe = PipelinedAligner(eng, head_group=head_group, encoder_streams=encoder_streams)
            pinned = [torch.empty((BATCH, labels.shape[1]), dtype=torch.int32).pin_memory() for _ in range(2)]
            status = torch.empty((BATCH,), dtype=torch.int32).pin_memory()

            def run(n):
                with torch.no_grad():
                    for _ in range(n):
                        pipe.submit(mel, labels, n_labels, n_frames=T_FRAMES, use_ctc=True, host_out=(pinned[0], pinned[1], status))
                    pipe.drain()
                torch.cuda.synchronize()

            run(head_group)
            t0 = time.perf_counter()
            run(steps)
            ms = (time.perf_counter() - t0) / steps * 1e3
            # the GEMM family's own time, as the headline's roofline leg takes it: a pass through a pipeline with ONE encoder stream
            # (two co-running GEMMs would each read twice their own time), every 7th launch bracketed
            timed_frames = (pinned[0].numpy().copy(), pinned[1].numpy().copy())
            pipe = PipelinedAligner(eng, head_group=head_group, encoder_streams=1)
            run(head_group)
            L.la_timer_reset(); L.la_timer_sample(7); L.la_timer_enable(family.encode())
            run(4)
            L.la_timer_disable()
            if not (np.array_equal(pinned[0].numpy(), timed_frames[0]) and np.array_equal(pinned[1].numpy(), timed_frames[1])):
                raise RuntimeError("one-encoder-stream pass and the two-stream pipeline disagree on the frames")
            total_ms, launches, work, seen = ctypes.c_double(0.0), ctypes.c_int64(0), ctypes.c_double(0.0), ctypes.c_int64(0)
            _lib.check(L.la_timer_read_work(ctypes.byref(total_ms), ctypes.byref(launches), ctypes.byref(work), ctypes.byref(seen)), "timer_read_work")
            eng.check_gru()
            tf = work.value / (total_ms.value * 1e-3) / 1e12 if total_ms.value > 0 else 0.0
            frames[name] = (pinned[0].numpy().copy(), pinned[1].numpy().copy())
            out[name] = {"ms_per_step": ms, "audio_s_per_s": BATCH * CLIP_SECONDS / (ms * 1e-3), "dtype": name.split("_")[0], "arithmetic": note,
                         "status_ok": bool(int((status != 0).sum()) == 0),
                         "gemm_family": family, "gemm_achieved_tflops": tf,
                         # f32_parity: the library counts the ALGORITHMIC float32 flops of a launch; the kernel executes three times that in f16 MFMAs
                         "gemm_mfma_executed_tflops": 3.0 * tf if name == "f32_parity" else tf,
                         "gemm_frac_of_f16_peak": (3.0 * tf if name == "f32_parity" else tf) / 2500.0,
                         "whole_path_tflops": total_flops_per_clip(dims.n_audio_state, dims.n_audio_layer) * BATCH / (ms * 1e-3) / 1e12,
                         "timing": f"{steps} pipelined steps after the timed region, the headline's pipeline shape; GEMM family: HIP events around every 7th "
                                   f"launch on a further pass of 4 steps with ONE encoder stream ({launches.value} launches bracketed; frames checked equal)"}
            del pipe, eng, m
        except Exception as e:          # noqa: BLE001 -- a leg never takes the headline down
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    # the precision side: boundaries (onset + offset frames of every label of the batch) equal to the float32 mode's
    if "f32_parity" in frames:
        ref_on, ref_off = frames["f32_parity"]
        Ls = n_labels.cpu().numpy()
        total = int(2 * Ls.sum())

        def same(on, off):
            return int(sum((on[b, :Ls[b]] == ref_on[b, :Ls[b]]).sum() + (off[b, :Ls[b]] == ref_off[b, :Ls[b]]).sum() for b in range(BATCH)))
        out["boundaries_equal_to_f32_parity"] = {"of": total, "bf16_headline": same(*ref_frames)}
        if "f16" in frames:
            out["boundaries_equal_to_f32_parity"]["f16"] = same(*frames["f16"])
    return out


def other_config_legs(headline_model, device, local_rank, deadline: float) -> dict:
    """BASELINE configs[2], [3], [4] on short passes in the same process (their full runs: --mode finetune | largev2 | longform)."""
    import types
    out = {}

    def leg(name, fn):
        if time.perf_counter() > deadline:
            out[name] = {"error": "skipped: the extra legs' time budget was used up"}
            return
        try:
            t0 = time.perf_counter()
            line = fn()
            r = line.get("roofline") or {}
            out[name] = {"ms_per_step": line["ms_per_step"], "audio_s_per_s": line["value"], "dtype": line["dtype"], "steps": line["steps"],
                         "roofline_frac": r.get("frac"), "roofline_achieved_tflops": r.get("achieved"), "workload": line["config"]["workload"],
                         "peak_mem_GB": line.get("peak_mem_GB"), "leg_wall_s": time.perf_counter() - t0}
        except Exception as e:          # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()

    base = dict(timer_period=7, allreduce_chunks=4, accum=8, accum_mode="fused", model=MODEL)
    leg("longform", lambda: run_other_config(types.SimpleNamespace(mode="longform", songs=16, clips=0, steps=6, warmup=2, **base), 0, 1, local_rank,
                                             device, None, model=headline_model))
    leg("largev2", lambda: run_other_config(types.SimpleNamespace(mode="largev2", songs=0, clips=512, steps=2, warmup=1, **base), 0, 1, local_rank,
                                            device, None))
    leg("finetune", lambda: run_finetune(types.SimpleNamespace(mode="finetune", steps=3, warmup=2, **base), 0, 1, local_rank, device, None))
    return out


def build_weights(name: str, with_decoder: bool, rank: int, world: int, dist):
    """Random-init weights of the architecture, the same bits on every rank.  With several ranks on the node each generates 1 / world of
    the tensors and they exchange the pieces through /dev/shm (whisper_compat.build_model_shared) instead of `world` full host builds
    on cores // world threads each."""
    from lyricalignment_amd import whisper_compat as wc
    if world == 1 or dist is None or os.environ.get("LA_BENCH_SHARED_BUILD", "1") == "0":
        return wc.build_model(name, seed=0, with_decoder=with_decoder)
    tag = os.environ.get("MASTER_PORT", "0") + "_" + name
    return wc.build_model_shared(name, 0, with_decoder, rank, world, dist.barrier, tag)


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside a launcher: start N ranks (one per GPU) as a CHILD `torch.distributed.run`
    with the same arguments and return its exit code.  This parent has made no HIP call (importing torch makes none) and
    makes none: the ranks' stdout is inherited, so rank 0's JSON line is this command's JSON line."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=DEFAULT_STEPS)
    ap.add_argument("--warmup", type=int, default=DEFAULT_WARMUP)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="single stream: no encoder/head overlap across batches")
    ap.add_argument("--dtype", choices=["bf16", "f16"], default="bf16",
                    help="16-bit operand type of the throughput kernels (BASELINE configs[1] is quoted on bf16)")
    ap.add_argument("--encoder-streams", type=int, default=DEFAULT_ENCODER_STREAMS,
                    help="HIP streams the encoders of consecutive batches alternate between (2: one batch's kernel tails and launch gaps "
                         "are filled by the other's kernels, ~2 %% per step; the roofline leg is then measured on a separate one-stream pass)")
    ap.add_argument("--roofline-steps", type=int, default=ROOFLINE_PASS_STEPS,
                    help="align mode: steps of the untimed single-encoder-stream pass after the timed region on which the per-launch timer runs")
    ap.add_argument("--head-group", type=int, default=DEFAULT_HEAD_GROUP,
                    help="batches whose head (GRU / FC / DP) runs as one launch set in the two-stream pipeline "
                         "(same-box sweep, profiles/r4_sweep_pipeline_shape.txt: 2 -> 42.47, 4 -> 42.27, 5 -> 42.23, 10 -> 42.69 ms per step)")
    ap.add_argument("--mode", choices=["align", "finetune", "largev2", "longform"], default="align",
                    help="align = BASELINE configs[1] (the headline metric); finetune = configs[2], the data-parallel multitask "
                         "fine-tune step (float32, per-GPU micro-batch 2 x 30 s, --accum micro-steps, ONE gradient all-reduce per step); "
                         "largev2 = configs[3] (Whisper-large-v2, float16, --clips 30 s clips per GPU and step); longform = configs[4] "
                         "(--songs 3-minute songs per GPU and step, 6 chunks each, pipelined)")
    ap.add_argument("--clips", type=int, default=512, help="largev2 mode: clips per GPU and step (3072 fills 207 GB)")
    ap.add_argument("--songs", type=int, default=16, help="longform mode: 180 s songs per GPU and step")
    ap.add_argument("--model", default=MODEL, help="finetune mode only: whisper architecture (medium = configs[2])")
    ap.add_argument("--accum", type=int, default=8, help="finetune mode: micro-steps per optimizer step (reference default 8)")
    ap.add_argument("--allreduce-chunks", type=int, default=4,
                    help="finetune mode: chunks the backbone gradient bucket is all-reduced in, each launched as the last backward completes it "
                         "(0 = one blocking all-reduce per bucket after the backward)")
    ap.add_argument("--accum-mode", choices=["fused", "loop"], default="fused",
                    help="finetune mode: the accum micro-batches as one fused forward / backward (per-micro-batch losses) or as a loop")
    ap.add_argument("--from-waveform", action="store_true",
                    help="align mode: the step starts one stage earlier, at the resident 16 kHz waveform [32, 480000] f32 -- the device "
                         "log-mel (la_logmel_f32_prepared) runs inside the timed step and feeds the encoder; adds logmel_ms_per_step")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="align mode, N = 1: skip the short passes after the timed region that fill \"modes\" (float32-parity and float16 on the "
                         "headline workload) and \"other_configs\" (BASELINE configs[2..4]) in the JSON line")
    ap.add_argument("--timer-period", type=int, default=7,
                    help="align mode: the roofline leg brackets every n-th launch of the GEMM family with HIP events (1 = every launch; odd and not a divisor of the 199 launches per pair of batches, so every shape is sampled alike)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks and never touches the GPU
        raise SystemExit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU "
                         f"(plain `python bench.py --gpus {args.gpus}` does that by itself)")
    # host threads: the box reports every core of the node, the cgroup quota is what this job may use, and with one rank
    # per GPU the ranks share it (random-init weight generation runs on the host)
    torch.set_num_threads(max(1, usable_cores() // max(world, 1)))
    # test knobs (exercise the multi-rank control flow on a one-GPU box): all ranks on device 0 over gloo
    same_device = os.environ.get("LA_BENCH_SAME_DEVICE") == "1"
    backend = os.environ.get("LA_BENCH_DIST_BACKEND", "nccl")
    if same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    # LA_BENCH_FORCE_DIST=1 (test knob): form the process group at world size 1 too, so that a one-GPU box exercises the RCCL
    # code path of the N > 1 runs (init with device_id, barrier, all_reduce MAX) -- needs RANK / WORLD_SIZE / MASTER_* from a launcher
    if world > 1 or os.environ.get("LA_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        world = dist.get_world_size()

    if args.mode == "finetune":
        return finetune_mode(args, rank, world, local_rank, device, dist)
    if args.mode in ("largev2", "longform"):
        return other_config_mode(args, rank, world, local_rank, device, dist)

    log(f"rank {rank}/{world}: building random-init whisper-{MODEL} weights")
    from lyricalignment_amd import _lib, whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    _lib.require_gpu()
    dims = wc.dims_for(MODEL)
    t_start = time.perf_counter()
    wm = build_weights(MODEL, False, rank, world, dist)
    t_built = time.perf_counter()
    model = AlignModel(wm, embed_dim=dims.n_audio_state, hidden_dim=HIDDEN, output_dim=VOCAB, device=f"cuda:{local_rank}",
                       compute_dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float16).eval()
    fit = fit_head(model, device, from_waveform=args.from_waveform)   # a head that has learnt the synthetic songs (linear probe)
    log(f"head fitted: {fit}")
    with torch.no_grad():
        eng = model.engine()
    log(f"rank {rank}: start-up {time.perf_counter() - t_start:.1f} s (weights built in {t_built - t_start:.1f} s; head fit + packing on the device the rest)")
    mel, labels, n_labels, Ls, plans = build_inputs(device, seed_offset=0, from_waveform=args.from_waveform)   # same batch on every rank
    wave = None
    if args.from_waveform:
        from lyricalignment_amd.audio_frontend import log_mel_spectrogram
        wave = mel                                                 # [32, 480000] f32, resident
        mel = log_mel_spectrogram(wave, device=device)             # (what the timed steps recompute; kept for the host self-check)
    pinned = [torch.empty((BATCH, labels.shape[1]), dtype=torch.int32).pin_memory() for _ in range(2)]
    pinned_status = torch.empty((BATCH,), dtype=torch.int32).pin_memory()

    from lyricalignment_amd.engine import PipelinedAligner
    pipe = None if args.no_overlap else PipelinedAligner(eng, head_group=args.head_group, encoder_streams=args.encoder_streams)

    def step():
        with torch.no_grad():
            mel_in = mel if wave is None else log_mel_spectrogram(wave, device=device)     # 2 launches on the encoder's stream
            if pipe is not None:   # encoder of this batch overlaps the head (GRU/FC/DP) of the previous one
                pipe.submit(mel_in, labels, n_labels, n_frames=T_FRAMES, use_ctc=True, host_out=(pinned[0], pinned[1], pinned_status))
                return
            onset, offset, score, status = eng.align_mel(mel_in, labels, n_labels, n_frames=T_FRAMES, use_ctc=True)
        pinned[0].copy_(onset, non_blocking=True)
        pinned[1].copy_(offset, non_blocking=True)
        pinned_status.copy_(status, non_blocking=True)

    def barrier():
        if dist is not None:
            dist.barrier()

    for i in range(args.warmup):
        step()
    if pipe is not None:
        pipe.drain()
    torch.cuda.synchronize()
    log(f"{args.warmup} warm-up steps done")
    eng.check_gru()

    L = _lib.lib()
    # The per-launch timer of the roofline leg reads kernel durations only while ONE encoder stream feeds the chip (two co-running
    # GEMMs share the CUs and each reads twice its own time).  With the default two encoder streams the timed region therefore runs
    # untimed per launch, and the leg is measured on a short pass of the same steps through a one-encoder-stream pipeline afterwards
    # (same engine, inputs, head group; the head stream still co-runs, as it does in the timed region).
    separate_pass = pipe is not None and args.encoder_streams > 1 and args.roofline_steps > 0
    L.la_timer_reset()
    L.la_timer_sample(args.timer_period)
    if not separate_pass:
        L.la_timer_enable(os.environ.get("LA_BENCH_TIMER", "gemm_bf16").encode())
    power = PowerSampler(local_rank) if rank == 0 else None
    barrier()
    torch.cuda.synchronize()
    if power is not None:
        power.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if pipe is not None:
        pipe.drain()                      # flushes a partial head group: every submitted batch is finished inside the timed region
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    power_reading = power.stop() if power is not None else None
    L.la_timer_disable()
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s")
    eng.check_gru()
    if int((pinned_status != 0).sum()) != 0:
        raise SystemExit("alignment reported non-OK status on the synthetic batch")
    last_onset, last_offset = pinned[0].numpy().copy(), pinned[1].numpy().copy()     # the timed region's last batch (self-check)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    roofline_pass_ms = None
    if separate_pass and rank == 0:
        timed_pipe = pipe
        pipe = PipelinedAligner(eng, head_group=args.head_group, encoder_streams=1)
        for _ in range(args.head_group):          # one untimed group: buffers of the one-stream pipeline allocated, clocks where they were
            step()
        pipe.drain()
        torch.cuda.synchronize()
        L.la_timer_reset()
        L.la_timer_enable(os.environ.get("LA_BENCH_TIMER", "gemm_bf16").encode())
        t1 = time.perf_counter()
        for _ in range(args.roofline_steps):
            step()
        pipe.drain()
        torch.cuda.synchronize()
        roofline_pass_ms = (time.perf_counter() - t1) / args.roofline_steps * 1e3
        L.la_timer_disable()
        eng.check_gru()
        if not (np.array_equal(pinned[0].numpy(), last_onset) and np.array_equal(pinned[1].numpy(), last_offset)):
            raise SystemExit("the one-encoder-stream roofline pass and the timed two-stream pipeline disagree on the frames")
        log(f"roofline pass: {args.roofline_steps} steps, one encoder stream, {roofline_pass_ms:.2f} ms per step")
        pipe = timed_pipe
    roofline_steps = args.roofline_steps if roofline_pass_ms is not None else args.steps

    import ctypes
    total_ms, launches, work, seen = ctypes.c_double(0.0), ctypes.c_int64(0), ctypes.c_double(0.0), ctypes.c_int64(0)
    _lib.check(L.la_timer_read_work(ctypes.byref(total_ms), ctypes.byref(launches), ctypes.byref(work), ctypes.byref(seen)), "timer_read_work")
    # achieved = algorithmic flops of the bracketed launches / their summed HIP-event durations.  The library sums 2 M N K of what it
    # launched; the only launch that computes padding is conv1 (80 mel channels padded to 128): scaled out analytically.
    d_ = dims.n_audio_state
    alg = algorithmic_gemm_flops_per_clip(d_, dims.n_audio_layer)
    launched = alg + 2.0 * 3000 * (128 - 80) * 3 * d_
    achieved_tf = work.value * (alg / launched) / (total_ms.value * 1e-3) / 1e12 if total_ms.value > 0 else 0.0

    logmel_ms = None
    if wave is not None:                      # the log-mel alone on the chip (it overlaps the previous batch's encoder inside the step)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a_, b_ in evs:
            a_.record(); log_mel_spectrogram(wave, device=device); b_.record()
        torch.cuda.synchronize()
        logmel_ms = sorted(a_.elapsed_time(b_) for a_, b_ in evs)[5]

    selfcheck_failed = False
    if rank == 0:
        audio_sec = world * BATCH * CLIP_SECONDS * args.steps
        out = {
            "metric": "aligned audio-sec/sec (RTF^-1), Whisper-medium 30 s clips",
            "value": audio_sec / elapsed,
            "unit": "audio-sec/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (songs of 5..26 notes out of 40 timbres as log-mel in [-1,1], transcript = the timbres' class ids; random-init "
                    "weights of the whisper-medium architecture, host-independent bits, with the head's 40 syllable rows fitted as a linear "
                    "probe on other synthetic songs)",
            "config": {"workload": "whisper-medium encoder + BiGRU/FC head + CTC forced alignment, batch 32 x 30 s mel per GPU "
                                   "(BASELINE.json configs[1])",
                       "clips_per_gpu": BATCH, "frames": T_FRAMES, "vocab": VOCAB,
                       "labels_per_clip": "5..26 (one per sung note)", "head_fit": fit,
                       "pipeline": ("single stream" if pipe is None else
                                    f"{args.encoder_streams} encoder stream(s) + 1 head stream, head over {args.head_group} batches"),
                       "sharding": "clips over ranks, no collective"},
            "whole_path_tflops": total_flops_per_clip(dims.n_audio_state, dims.n_audio_layer) * BATCH * world * args.steps / elapsed / 1e12,
            "roofline": {"bound": "mfma", "kernel": "gemm_pp_kernel / gemm_kernel <bf16> (every Linear, conv-as-GEMM and GRU input projection launch)",
                         "achieved": achieved_tf, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved_tf / 2500.0,
                         "traffic": measured_gemm_traffic_per_launch(), "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/" + PMC_SUMMARY + ")",
                         "launches_per_step": seen.value / max(roofline_steps, 1),
                         "avg_launch_ms": total_ms.value / max(launches.value, 1),
                         "timed_launches": launches.value,
                         "power": power_reading,
                         "timing": (f"HIP events around every {args.timer_period}. launch of the family on its own stream, " +
                                    (f"taken on a separate pass of {args.roofline_steps} steps AFTER the timed region through a pipeline with ONE "
                                     f"encoder stream ({roofline_pass_ms:.2f} ms per step there; same engine, inputs and head group, frames checked "
                                     "equal): the timed region runs two encoder streams, whose co-running launches would each read twice their "
                                     "own time.  value / ms_per_step / whole_path_tflops / power are the timed region's"
                                     if roofline_pass_ms is not None else
                                     "inside the timed region (an event record is a barrier packet: ~6.6 us of stream idle time each)"))},
        }
        if wave is not None:
            out["config"]["input"] = "16 kHz waveform [32, 480000] f32 resident in HBM; device log-mel inside the timed step"
            out["logmel_ms_per_step"] = logmel_ms
            out["logmel_launches_per_step"] = 2
            out["logmel_algorithmic_bytes"] = BATCH * (480000 * 4 + 80 * 3000 * 4)
        if world > 1:
            out["cpu_baseline"] = None          # the host baseline is timed on rank 0 of the N = 1 run only
        elif not args.no_cpu_baseline:
            lab_cpu = labels.cpu().numpy()
            base, cpu_res = cpu_baseline(model, mel[:3].cpu().numpy(), [lab_cpu[i, : int(Ls[i])] for i in range(3)], dims.n_audio_head)
            out["cpu_baseline"] = base
            chk = selfcheck(last_onset, last_offset, cpu_res, Ls, plans)
            out["selfcheck"] = chk
            out["cpu_vs_gpu_onset_mae_s"] = chk["onset_mae_s"]
            selfcheck_failed = max(chk["onset_mae_s"], chk["offset_mae_s"]) > SELFCHECK_TOL_S
        if world == 1 and not args.no_extra_legs and not args.no_overlap and wave is None:
            # the headline dict above is complete; what follows only ADDS keys.  The headline's pipeline buffers go first (the fine-tune
            # leg keeps ~60 GB of activations).
            t_legs = time.perf_counter()
            head_state = {k: v.detach().clone() for k, v in model.align_rnn.state_dict().items()}
            pipe = None
            torch.cuda.empty_cache()
            try:
                out["modes"] = mode_legs(wm, head_state, dims, device, local_rank, mel, labels, n_labels, args.head_group, args.encoder_streams,
                                         (last_onset, last_offset))
            except Exception as e:          # noqa: BLE001
                out["modes"] = {"error": f"{type(e).__name__}: {e}"}
            log(f"mode legs done in {time.perf_counter() - t_legs:.1f} s")
            try:
                out["other_configs"] = other_config_legs(model, device, local_rank, t_legs + EXTRA_LEGS_BUDGET_S)
            except Exception as e:          # noqa: BLE001
                out["other_configs"] = {"error": f"{type(e).__name__}: {e}"}
            out["extra_legs_wall_s"] = time.perf_counter() - t_legs
            log(f"extra legs done in {out['extra_legs_wall_s']:.1f} s")
        print(json.dumps(out), flush=True)
        if selfcheck_failed:
            log(f"SELF-CHECK FAILED: GPU vs oracle boundary MAE {chk['onset_mae_s']:.3f} / {chk['offset_mae_s']:.3f} s > {SELFCHECK_TOL_S} s")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if selfcheck_failed:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
