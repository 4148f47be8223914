#!/usr/bin/env python3
"""tests/golden/gen_golden.py -- regenerates the committed golden fixtures.

Run ONLY in the build container (needs the read-only reference checkout):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [/root/reference]

It imports the REFERENCE's own Python modules (utils/alignment.py,
module/align_model.py, train_multitask.py) and records inputs + the outputs
they produce.  Nothing of the reference's source text is stored: fixtures are
data (seeds, arrays, scalars).  The test-suite and the GPU box read only the
fixture files, never /root/reference.

Third-party modules the reference imports but this image lacks are replaced by
inert stand-ins *inside this process only* (sys.modules), exactly as
SURVEY.md Appendix E describes:
  numba     -> jit() returning the undecorated function (run_viterbi_core is
               plain Python underneath; @jit(nopython=True) has no fastmath)
  pypinyin  -> names only (never called on the hot path)
  whisper   -> class Whisper + whisper.audio built from oracle/model_oracle.py
               (openai-whisper is un-pinned and absent: the encoder / log-mel
               side of the oracle stays "parity unpinned")
  librosa   -> name only
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = next((a for a in sys.argv[1:] if not a.startswith("--")), "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import model_oracle as mo


# --------------------------------------------------------------------------- #
# stand-ins for absent third-party packages                                     #
# --------------------------------------------------------------------------- #
def _install_stubs():
    numba = types.ModuleType("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    numba.jit = jit
    sys.modules["numba"] = numba

    pyp = types.ModuleType("pypinyin")
    pyp.lazy_pinyin = lambda *a, **k: []
    pyp.Style = types.SimpleNamespace(NORMAL=0, TONE3=8)
    sys.modules["pypinyin"] = pyp

    whisper = types.ModuleType("whisper")

    class Whisper(torch.nn.Module):
        pass

    whisper.Whisper = Whisper
    whisper.log_mel_spectrogram = mo.log_mel_spectrogram
    whisper.pad_or_trim = mo.pad_or_trim
    audio = types.ModuleType("whisper.audio")
    audio.N_FRAMES = mo.N_FRAMES
    audio.pad_or_trim = mo.pad_or_trim
    audio.log_mel_spectrogram = mo.log_mel_spectrogram
    tok = types.ModuleType("whisper.tokenizer")
    tok.Tokenizer = type("Tokenizer", (), {})
    tok.get_tokenizer = lambda *a, **k: None
    whisper.audio = audio
    whisper.tokenizer = tok
    whisper.load_model = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    sys.modules["whisper"] = whisper
    sys.modules["whisper.audio"] = audio
    sys.modules["whisper.tokenizer"] = tok

    librosa = types.ModuleType("librosa")
    librosa.load = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    sys.modules["librosa"] = librosa


import transformers  # noqa: E402,F401  (must be imported BEFORE the librosa stand-in exists:
from transformers import AutoTokenizer, get_linear_schedule_with_warmup  # noqa: E402,F401  Appendix E gotcha)

_install_stubs()
sys.path.insert(0, REF)
import utils.alignment as ref_align  # noqa: E402  (the reference's module)
import module.align_model as ref_model  # noqa: E402
import train_multitask as ref_train  # noqa: E402

assert os.path.realpath(ref_align.__file__).startswith(os.path.realpath(REF))


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def make_labels(rs, L, lo, hi, repeat_at=None):
    lab = rs.randint(lo, hi, size=L).astype(np.int64)
    # no accidental repeats, then one forced repeat where asked
    for i in range(1, L):
        while lab[i] == lab[i - 1]:
            lab[i] = rs.randint(lo, hi)
    if repeat_at is not None and 0 < repeat_at < L:
        lab[repeat_at] = lab[repeat_at - 1]
    return lab


# --------------------------------------------------------------------------- #
# 1. run_viterbi_core called directly (exact, platform-stable inputs)           #
# --------------------------------------------------------------------------- #
def core_inputs(seed, T, L, Vp, scale, repeat_at):
    """Shared with tests/test_oracle_viterbi.py through the stored parameters:
    legacy RandomState streams are frozen by numpy, so these regenerate
    bit-identically anywhere."""
    rs = np.random.RandomState(seed)
    lp = (-rs.rand(T, Vp) * scale).astype(np.float32)
    ls = (-rs.rand(T, 1) * scale).astype(np.float32)
    label = make_labels(rs, L, 1, Vp + 1, repeat_at)
    return lp, ls, label


def gen_core():
    cases = []
    grid = [
        (101, 5, 4, 12, 3.0, 2), (102, 60, 4, 40, 3.0, 2), (103, 400, 19, 450, 3.0, 7),
        (104, 1500, 26, 402, 3.0, 11), (105, 1500, 26, 402, 0.01, 11), (106, 1500, 26, 402, 0.0, 5),
        (107, 3000, 120, 402, 1.0, 60), (108, 5389, 171, 402, 0.1, 33), (109, 800, 19, 21127, 0.003, 3),
        (110, 2, 1, 5, 1.0, None), (111, 1, 1, 5, 1.0, None), (112, 300, 53, 402, 0.001, 20),
    ]
    for seed, T, L, Vp, scale, rep in grid:
        lp, ls, label = core_inputs(seed, T, L, Vp, scale, rep)
        S = 2 * L + 1
        dp = np.full((T, S), -10000000.0, dtype=np.float64)
        bt = np.zeros((T, S), dtype=np.int64)
        dp[0][0] = ls[0][0]
        dp[0][1] = lp[0][label[0] - 1]
        dp, bt = ref_align.run_viterbi_core(dp, bt, lp, ls, label)
        cases.append(dict(seed=seed, T=T, L=L, Vp=Vp, scale=scale, repeat_at=rep,
                          label=label.tolist(), bt_sha256=sha(bt), dp_sha256=sha(dp),
                          dp_last=[float(v) for v in dp[-1]]))
        print("core", seed, T, L, Vp, scale, flush=True)
    with open(os.path.join(HERE, "viterbi_core.json"), "w") as f:
        json.dump(dict(generator="gen_golden.py::gen_core",
                       reference="utils/alignment.py:73-119 run_viterbi_core", cases=cases), f, indent=1)


# --------------------------------------------------------------------------- #
# 2. perform_viterbi(_ctc) end to end, recording what the DP core was fed        #
# --------------------------------------------------------------------------- #
class Recorder:
    def __init__(self):
        self.calls = []
        self.orig = ref_align.run_viterbi_core

    def __call__(self, dp, bt, lp, ls, label):
        self.calls.append((np.array(lp, copy=True), np.array(ls, copy=True), np.array(label, copy=True)))
        return self.orig(dp, bt, lp, ls, label)


def compact(lp, ls, label):
    em = np.empty((lp.shape[0], 1 + len(label)), dtype=np.float32)
    em[:, 0] = ls[:, 0]
    em[:, 1:] = lp[:, np.asarray(label) - 1]
    return em


def gen_e2e():
    out = {}
    meta = []
    grid = [
        # name, variant, seed, B, T, V, scale, Ls, repeat
        ("ctc_t5_repeat", "ctc", 201, 1, 5, 14, 3.0, [4], 2),
        ("ctc_t60", "ctc", 202, 1, 60, 42, 3.0, [4], 2),
        ("ctc_t400", "ctc", 203, 1, 400, 452, 3.0, [19], 7),
        ("ctc_t1500", "ctc", 204, 1, 1500, 404, 3.0, [26], 11),
        ("ctc_t1500_lowcontrast", "ctc", 205, 1, 1500, 404, 0.01, [26], 11),
        ("ctc_t1500_flat", "ctc", 206, 1, 1500, 404, 0.0, [26], 5),
        ("ctc_bigvocab", "ctc", 207, 1, 188, 21129, 1.0, [11], 3),
        ("plain_batch3", "plain", 208, 3, 120, 60, 2.0, [9, 4, 1], 2),
        ("ctc_batch4", "ctc", 209, 4, 250, 404, 1.0, [26, 5, 1, 13], 3),
        ("plain_t1500", "plain", 210, 1, 1500, 403, 0.02, [26], 9),
    ]
    for name, variant, seed, B, T, V, scale, Ls, rep in grid:
        rs = np.random.RandomState(seed)
        logits = torch.from_numpy((rs.randn(B, T, V) * scale).astype(np.float32))
        n_cls = V - 2 if variant == "ctc" else V - 1
        Lmax = max(Ls)
        labels = torch.full((B, Lmax), -100, dtype=torch.long)
        for b, L in enumerate(Ls):
            labels[b, :L] = torch.from_numpy(make_labels(rs, L, 1, n_cls + 1, rep if L > rep else None))
        rec = Recorder()
        ref_align.run_viterbi_core = rec
        try:
            fn = ref_align.perform_viterbi_ctc if variant == "ctc" else ref_align.perform_viterbi
            res = fn(logits, labels)
        finally:
            ref_align.run_viterbi_core = rec.orig
        for b, (lp, ls, label) in enumerate(rec.calls):
            assert lp.dtype == np.float32 and ls.dtype == np.float32 and label.dtype == np.int64
            out[f"{name}/{b}/em"] = compact(lp, ls, label)
            out[f"{name}/{b}/label"] = label
            out[f"{name}/{b}/seconds"] = np.asarray(res[b], dtype=np.float64)
        meta.append(dict(name=name, variant=variant, seed=seed, B=B, T=T, V=V, scale=scale, Ls=Ls,
                         labels=labels.tolist()))
        print("e2e", name, flush=True)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "viterbi_e2e.npz"), **out)

    # error behaviour (utils/alignment.py:152 IndexError, :183 ValueError)
    errs = []
    rs = np.random.RandomState(301)
    logits = torch.from_numpy(rs.randn(1, 4, 14).astype(np.float32))
    for labels, T in (([[3, 7, 7, 9]], 4), ([[3, 7, 7, 9]], 5), ([[-100, -100]], 4), ([[5, 6, 7, 8, 9]], 3)):
        lg = torch.from_numpy(np.random.RandomState(302).randn(1, T, 14).astype(np.float32))
        try:
            r = ref_align.perform_viterbi_ctc(lg, torch.tensor(labels))
            errs.append(dict(labels=labels, T=T, V=14, seed=302, raises=None, result=r))
        except Exception as e:  # noqa: BLE001
            errs.append(dict(labels=labels, T=T, V=14, seed=302, raises=type(e).__name__, message=str(e)))
    with open(os.path.join(HERE, "viterbi_errors.json"), "w") as f:
        json.dump(errs, f, indent=1)


# --------------------------------------------------------------------------- #
# 3. emission prep (full lp / ls as the reference computed them)                #
# --------------------------------------------------------------------------- #
def gen_emission():
    out = {}
    for variant in ("ctc", "plain"):
        rs = np.random.RandomState(401)
        logits = (rs.randn(2, 6, 11) * 4).astype(np.float32)
        logits[0, 1, -1] = 100.0   # silence saturates: log(1 - sigmoid) = -inf -> -1000
        logits[0, 2, -1] = -120.0  # voiced saturates: log(sigmoid) = -inf -> -1000
        logits[1, 3, 4] = 2000.0   # one column dominates: other log-probs clip at -1000
        logits[1, 4, 0] = 3000.0   # column 0 (CTC blank / plain silence)
        lg = torch.from_numpy(logits)
        labels = torch.tensor([[1, 2, 3], [4, 5, -100]])
        rec = Recorder()
        ref_align.run_viterbi_core = rec
        try:
            (ref_align.perform_viterbi_ctc if variant == "ctc" else ref_align.perform_viterbi)(lg, labels)
        finally:
            ref_align.run_viterbi_core = rec.orig
        out[f"{variant}/logits"] = logits
        out[f"{variant}/lp"] = np.stack([c[0] for c in rec.calls])
        out[f"{variant}/ls"] = np.stack([c[1] for c in rec.calls])
    np.savez_compressed(os.path.join(HERE, "emission_prep.npz"), **out)
    print("emission", flush=True)


# --------------------------------------------------------------------------- #
# 4. the reference's RNN head class and AlignModel control flow                 #
# --------------------------------------------------------------------------- #
def gen_head():
    torch.manual_seed(501)
    rnn = ref_model.RNN(input_size=24, hidden_size=8, output_size=13, bidirectional=True, dropout=0.15)
    rnn.eval()
    x = torch.randn(2, 17, 24)
    with torch.no_grad():
        y = rnn(x)
    out = {"x": x.numpy(), "y": y.numpy()}
    for k, v in rnn.state_dict().items():
        out["sd/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "head_rnn.npz"), **out)
    keys = {k: list(v.shape) for k, v in ref_model.RNN(1024, 384, 21129, dropout=0.15).state_dict().items()}
    with open(os.path.join(HERE, "head_state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=1)
    print("head", flush=True)


class FakeWhisper(sys.modules["whisper"].Whisper):
    """Counts encoder calls; returns a deterministic [B,1500,d] tensor."""

    def __init__(self, d=8):
        super().__init__()
        self.d = d
        self.calls = []
        self.w = torch.nn.Parameter(torch.zeros(1))

    def embed_audio(self, mel):
        assert mel.shape[-1] == 3000
        self.calls.append(tuple(mel.shape))
        t = torch.arange(1500, dtype=torch.float32)[None, :, None]
        return (t / 1500.0 + mel[:, :1, ::2].transpose(1, 2)).expand(mel.shape[0], 1500, self.d).contiguous()

    def logits(self, tokens, audio_features):
        return audio_features[:, : tokens.shape[1], :1].expand(-1, -1, 5)


def gen_frames():
    rows = []
    for n_samples in (16000, 48160, 48480, 48000, 60096, 479999, 480000, 480160, 496000, 1040000, 960000, 961760):
        for get_orig_len in (True, False):
            fw = FakeWhisper()
            m = ref_model.AlignModel(whisper_model=fw, embed_dim=8, hidden_dim=4, output_dim=7, device="cpu")
            m.eval()
            rs = np.random.RandomState(n_samples % 1000)
            audios = [(rs.randn(n_samples) * 0.01).astype(np.float32),
                      (rs.randn(max(1, n_samples - 700)) * 0.01).astype(np.float32)]
            with torch.no_grad():
                a, t = m.frame_manual_forward(audios, get_orig_len=get_orig_len)
            rows.append(dict(n_samples=n_samples, get_orig_len=get_orig_len, out_shape=list(a.shape),
                             encoder_calls=[list(c) for c in fw.calls], transcribe_is_none=t is None))
    with open(os.path.join(HERE, "frame_counts.json"), "w") as f:
        json.dump(dict(reference="module/align_model.py:72-123", rows=rows), f, indent=1)
    print("frames", flush=True)


# --------------------------------------------------------------------------- #
# 5. losses and MAE                                                             #
# --------------------------------------------------------------------------- #
def gen_losses():
    out = {}
    rs = np.random.RandomState(601)
    V = 31  # stand-in for 21128; logits carry V+1 columns like the CTC head (21129)
    logits = torch.from_numpy(rs.randn(2, 40, V + 1).astype(np.float32)).requires_grad_(True)
    frame_labels = torch.full((2, 33), -100, dtype=torch.long)
    frame_labels[0, 3:9] = 5
    frame_labels[0, 9:20] = 17
    frame_labels[1, 0:12] = 30
    frame_labels[1, 20:33] = 2
    loss_fn = {"ce_loss": torch.nn.CrossEntropyLoss(), "silence_ce_loss": torch.nn.BCEWithLogitsLoss()}
    ce = ref_train.compute_ce_loss(logits, frame_labels.clone(), loss_fn, compute_sil=True, vocab_size=V,
                                   device="cpu")
    (g_ce,) = torch.autograd.grad(ce, logits)
    labels = torch.tensor([[4, 9, 9, 21, -100], [7, 3, -100, -100, -100]])
    ctc = ref_train.compute_ctc_loss(logits[:, :, :V], labels, device="cpu")
    (g_ctc,) = torch.autograd.grad(ctc, logits)
    out.update(logits=logits.detach().numpy(), frame_labels=frame_labels.numpy(), labels=labels.numpy(),
               vocab_size=np.int64(V), ce=ce.detach().numpy(), g_ce=g_ce.numpy(), ctc=ctc.detach().numpy(),
               g_ctc=g_ctc.numpy())
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)

    gt = [[[0.10, 0.52], [0.52, 0.98]], [[1.0, 1.5]]]
    pr = [[[0.12, 0.50], [0.56, 1.00]], [[1.66, 1.6600000000000001]]]
    with open(os.path.join(HERE, "mae.json"), "w") as f:
        json.dump(dict(gt=gt, predict=pr, mae=ref_align.get_mae(gt, pr)), f, indent=1)
    print("losses", flush=True)


# --------------------------------------------------------------------------- #
# 6. evaluation glue: the reference's align_and_evaluate on fake model + loader  #
# --------------------------------------------------------------------------- #
def gen_harness():
    import contextlib
    import io
    import inference_alignment as ref_inf

    class FakeModel(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))
            self.i = 0

        def frame_manual_forward(self, audios):
            rs = np.random.RandomState(900 + self.i)
            self.i += 1
            return torch.from_numpy((rs.randn(len(audios), 90, 404) * 3).astype(np.float32)), None

    with open(os.path.join(REF, "bert_base_chinese_pronunce_table.json")) as f:
        token_pinyin, _, lookup = json.load(f)
    rs = np.random.RandomState(77)
    ok_ids = [i for i, p in enumerate(token_pinyin) if lookup[p] <= 402 and i not in (0, 102)]
    batches, raw_tokens = [], []
    for b in range(5):
        B = 1 if b % 2 == 0 else 2
        Ls = [int(rs.randint(3, 9)) for _ in range(B)]
        tok = torch.full((B, max(Ls)), -100, dtype=torch.long)
        for i, L in enumerate(Ls):
            tok[i, :L] = torch.tensor([ok_ids[j] for j in rs.randint(0, len(ok_ids), size=L)])
        raw_tokens.append(tok.clone())
        if b == 3:
            gt = (None,)
        else:
            gt = tuple([[float(0.1 * k), float(0.1 * k + 0.08)] for k in range(L)] for L in Ls)
        batches.append(((np.zeros(16000, dtype=np.float32),) * B, tok, None, gt, None, None))
    cwd = os.getcwd()
    os.chdir(REF)  # the reference opens its JSON table relative to the CWD (inference_alignment.py:136)
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            avg = ref_inf.align_and_evaluate(FakeModel(), None, batches, use_ctc_loss=True, device="cpu")
    finally:
        os.chdir(cwd)
    mapped = [b[1] for b in batches]  # the reference maps the token tensors in place (:149-152)
    pairs = sorted({(int(r), int(m)) for rt, mt in zip(raw_tokens, mapped) for r, m in zip(rt.flatten(), mt.flatten()) if r != -100})
    with open(os.path.join(HERE, "harness.json"), "w") as f:
        json.dump(dict(reference="inference_alignment.py:126-180 align_and_evaluate", avg_mae=avg,
                       logits=dict(seed_base=900, shape_tail=[90, 404], scale=3.0),
                       raw_tokens=[t.tolist() for t in raw_tokens], mapped_tokens=[t.tolist() for t in mapped],
                       token_to_class=pairs,
                       gt=[None if b[3] == (None,) else list(b[3]) for b in batches]), f, indent=1)
    print("harness", avg, flush=True)


# --------------------------------------------------------------------------- #
# 6b. the reference's own train_step / evaluate on an oracle-backed whisper        #
# --------------------------------------------------------------------------- #
TRAIN_STEP_DIMS = dict(n_audio_state=128, n_audio_head=2, n_audio_layer=1, n_text_state=128, n_text_head=2, n_text_layer=1, n_vocab=311, n_text_ctx=64)
TRAIN_STEP_CFG = dict(hidden_dim=64, lr=5e-3, backbone_lr=1e-4, weight_decay=1e-5, warmup_steps=1, train_steps=10, accum_grad_steps=2,
                      max_grad_norm=1.0, optimizer_steps=3, whisper_seed=90, whisper_std=0.05, head_seed=7, head_fc_scale=1.0, head_rnn_scale=1.0,
                      n_samples=12)


def _train_step_batches(ok_ids, n_batches, seed):
    """Batches in the layout the reference's collate_fn hands to train_step (dataset.py:212-232): (audios, align_text_tokens [B, Lmax],
    [frame_labels_i | None], [onset_offset_i | None], decoder_input [B, n], decoder_output [B, n]); items 0, 1 carry frame labels (the
    multitask sub-batch of split_batch), item 2 does not (the transcript-only one).  Token ids are raw bert-base-chinese ids (train_step maps them)."""
    rs = np.random.RandomState(seed)
    out = []
    for b in range(n_batches):
        B = 3
        ars = np.random.RandomState(seed * 100 + b)          # the clips from a stream of their own: the tests regenerate them from (seed, b, lengths)
        audios = tuple((ars.randn(int(16000 * s)) * 0.1).astype(np.float32) for s in (1.0, 0.75, 0.9))
        Ls = [int(rs.randint(3, 7)) for _ in range(B)]
        tok = torch.full((B, max(Ls)), -100, dtype=torch.long)
        for i, L in enumerate(Ls):
            tok[i, :L] = torch.tensor([ok_ids[j] for j in rs.randint(0, len(ok_ids), size=L)])
        frame_labels, onoff = [], []
        for i, L in enumerate(Ls):
            if i == 2:
                frame_labels.append(None); onoff.append(None)
                continue
            n_fr = int(rs.randint(40, 50))
            fl = torch.full((n_fr,), -100, dtype=torch.long)
            edges = np.linspace(2, n_fr - 2, L + 1).astype(int)
            for k in range(L):
                fl[edges[k]: edges[k + 1] - 1] = tok[i, k]
            frame_labels.append(fl)
            onoff.append([[float(edges[k]) * 0.02, float(edges[k + 1] - 1) * 0.02] for k in range(L)])
        n_tok = int(rs.randint(4, 7))
        dec_in = torch.from_numpy(rs.randint(1, 300, size=(B, n_tok)).astype(np.int64))
        dec_out = torch.from_numpy(rs.randint(1, 300, size=(B, n_tok)).astype(np.int64))
        dec_out[1, -1] = -100
        out.append((audios, tok, frame_labels, onoff, dec_in, dec_out))
    return out


def gen_train_step():
    """Drives train_multitask.train_step (:215-342) and evaluate (:345-458) THEMSELVES -- split_batch, the in-place label LUT, compute_ce_loss /
    compute_ctc_loss, loss.backward(), clip_grad_norm_(model.parameters()), torch.optim.AdamW with the script's two parameter groups (:683-686),
    get_linear_schedule_with_warmup -- on the reference's AlignModel around a `whisper` stand-in whose embed_audio / logits are the oracle's
    float32 restatement over openai-whisper-named Parameters (tiny dims, dropout 0).  Stores the inputs (data), the returned `losses` dicts of
    every optimizer step, evaluate()'s dict before and after, and per parameter the L2 norm and 12 sampled entries of (parameter - initial
    value) after every step.  Both use_ctc_loss values."""
    from lyricalignment_amd import whisper_compat as wc       # parameter CONTAINER + host-independent initialisation only: no arithmetic of the build runs here
    WhisperBase = sys.modules["whisper"].Whisper

    class OracleWhisper(WhisperBase):
        def __init__(self, dims):
            super().__init__()
            inner = wc.build_model(dims=dims, seed=TRAIN_STEP_CFG["whisper_seed"], std=TRAIN_STEP_CFG["whisper_std"], with_decoder=True)
            self.dims = dims
            self.encoder, self.decoder = inner.encoder, inner.decoder

        def _p(self):
            p = dict(self.named_parameters())
            p.update(dict(self.named_buffers()))
            return p

        def embed_audio(self, mel):
            return mo.encoder_forward(self._p(), mel, n_head=self.dims.n_audio_head)

        def logits(self, tokens, audio_features):
            return mo.decoder_forward(self._p(), tokens, audio_features, n_head=self.dims.n_text_head)

    with open(os.path.join(REF, "bert_base_chinese_pronunce_table.json")) as f:
        token_pinyin, _, lookup = json.load(f)
    ok_ids = [i for i, p in enumerate(token_pinyin) if 1 <= lookup[p] <= 402 and i not in (0, 102)]
    cfg = TRAIN_STEP_CFG
    fixture = dict(reference="train_multitask.py:215-342 train_step, :345-458 evaluate, :683-690 optimizer / schedule", dims=TRAIN_STEP_DIMS, cfg=cfg, runs={})
    arrays = {}
    for use_ctc in (True, False):
        tag = "ctc" if use_ctc else "noctc"
        dims = wc.ModelDimensions(**TRAIN_STEP_DIMS)
        torch.manual_seed(0)
        V = 21128 + int(use_ctc)
        model = ref_model.AlignModel(whisper_model=OracleWhisper(dims), embed_dim=dims.n_audio_state, hidden_dim=cfg["hidden_dim"], dropout=0.0,
                                     output_dim=V, train_alignment=True, train_transcript=True, device="cpu")
        wc.init_align_head(model, seed=cfg["head_seed"], fc_scale=cfg["head_fc_scale"], rnn_scale=cfg["head_rnn_scale"])
        init = {n: p_.detach().clone() for n, p_ in model.named_parameters()}
        optimizer = torch.optim.AdamW([{"params": model.align_rnn.parameters(), "lr": cfg["lr"]},
                                       {"params": model.whisper_model.parameters(), "lr": cfg["backbone_lr"]}], lr=cfg["lr"], weight_decay=cfg["weight_decay"])
        scheduler = get_linear_schedule_with_warmup(optimizer, num_warmup_steps=cfg["warmup_steps"], num_training_steps=cfg["train_steps"])
        loss_fn = {"ce_loss": torch.nn.CrossEntropyLoss(), "silence_ce_loss": torch.nn.BCEWithLogitsLoss()}
        train = _train_step_batches(ok_ids, cfg["optimizer_steps"] * cfg["accum_grad_steps"], seed=1000 + int(use_ctc))
        dev = _train_step_batches(ok_ids, 2, seed=2000 + int(use_ctc))
        it = iter(train)
        import contextlib
        import io

        def run_eval():
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                return ref_train.evaluate(model, dev, loss_fn, token_pinyin, lookup, use_ctc_loss=use_ctc, get_orig_len=False)

        rs = np.random.RandomState(5)
        sample_idx = {n: rs.randint(0, p_.numel(), size=cfg["n_samples"]) for n, p_ in init.items()}
        run = dict(eval_before=run_eval(), steps=[])
        for k in range(cfg["optimizer_steps"]):
            losses = ref_train.train_step(model, it, optimizer, scheduler, cfg["accum_grad_steps"], cfg["max_grad_norm"], loss_fn, token_pinyin, lookup,
                                          use_ctc_loss=use_ctc, get_orig_len=False)
            delta_l2, param_l2 = {}, {}
            for n, p_ in model.named_parameters():
                dlt = (p_.detach().double() - init[n].double())
                delta_l2[n] = float(dlt.norm())
                param_l2[n] = float(p_.detach().double().norm())
                arrays[f"{tag}.step{k}.delta.{n}"] = dlt.flatten()[sample_idx[n]].numpy()
            run["steps"].append(dict(losses={a: float(b) for a, b in losses.items()}, delta_l2=delta_l2, param_l2=param_l2,
                                     lr=[float(g["lr"]) for g in optimizer.param_groups]))
            print("train_step", tag, k, run["steps"][-1]["losses"], flush=True)
        run["eval_after"] = run_eval()
        fixture["runs"][tag] = run
        for n in init:
            arrays[f"{tag}.sample_idx.{n}"] = sample_idx[n]
        # the inputs: audio by seed (RandomState is stable across hosts), everything else as data
        def pack(batches, name):
            for bi, (audios, tok, fls, onoff, din, dout) in enumerate(batches):
                arrays[f"{tag}.{name}{bi}.tokens"] = tok.numpy()
                for i, fl in enumerate(fls):
                    if fl is not None:
                        arrays[f"{tag}.{name}{bi}.frame_labels{i}"] = fl.numpy()
                arrays[f"{tag}.{name}{bi}.dec_in"] = din.numpy()
                arrays[f"{tag}.{name}{bi}.dec_out"] = dout.numpy()
                arrays[f"{tag}.{name}{bi}.audio_len"] = np.array([len(a) for a in audios])
        pack(train, "train"); pack(dev, "dev")
        fixture["runs"][tag].update(train_seed=1000 + int(use_ctc), dev_seed=2000 + int(use_ctc), n_train=len(train), n_dev=len(dev), output_dim=V)
        used = sorted({int(t) for bs in (train, dev) for b in bs for t in b[1].flatten().tolist() if t != -100})
        fixture["runs"][tag]["token_to_class"] = [[t, int(lookup[token_pinyin[t]])] for t in used]
    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **arrays)
    with open(os.path.join(HERE, "train_step.json"), "w") as f:
        json.dump(fixture, f, indent=1)
    print("train_step fixture written", flush=True)


# --------------------------------------------------------------------------- #
# 7. collate-side label builders (dataset.py)                                    #
# --------------------------------------------------------------------------- #
def gen_dataset_labels():
    import dataset as ref_ds
    cases = []
    for use_ctc in (True, False):
        ds = ref_ds.AlignDataset(records=[], hf_tokenizer=None, use_ctc=use_ctc)
        for toks, oo in (([5, 17, 17, 230], [[0.11, 0.51], [0.51, 0.99], [1.01, 1.49], [1.5, 1.75]]),
                         ([401], [[0.0, 0.03]]),
                         ([7, 8, 9], [[0.25, 0.45], [0.45, 0.65], [0.65, 3.756]])):
            fl = ds._get_frame_label(torch.tensor(toks), oo)
            cases.append(dict(use_ctc=use_ctc, tokens=toks, on_offset=oo, frame_labels=fl.tolist()))
    with open(os.path.join(HERE, "frame_labels.json"), "w") as f:
        json.dump(dict(reference="dataset.py:129-145 AlignDataset._get_frame_label", cases=cases), f, indent=1)
    print("dataset labels", flush=True)


# --------------------------------------------------------------------------- #
# 7. CER scoring (utils/CER.py): error rate + operation counts on seeded pairs   #
# --------------------------------------------------------------------------- #
def gen_cer():
    from utils import CER as ref_cer
    rs = np.random.RandomState(4242)
    cases = []

    def add(hyp, ref):
        cer, nb = ref_cer.CER(hypothesis=list(hyp), reference=list(ref))
        cases.append({"hyp": list(hyp), "ref": list(ref), "cer": float(cer), "nb_map": {k: int(v) for k, v in nb.items()}})

    alphabet = [chr(0x4e00 + i) for i in range(12)]          # a small alphabet so that matches are frequent
    add([], alphabet[:5])                                      # empty hypothesis (evaluate_transcript.py's except branch)
    add(alphabet[:6], alphabet[:6])                            # identical
    add(alphabet[:1], alphabet[:1])
    add(alphabet[:3], alphabet[3:4])                           # all wrong, hypothesis longer
    for n in range(40):
        L = int(rs.randint(1, 30))
        ref = [alphabet[j] for j in rs.randint(0, len(alphabet), size=L)]
        hyp = list(ref)
        for _ in range(int(rs.randint(0, 8))):                 # random edits
            op = rs.randint(0, 3)
            pos = int(rs.randint(0, len(hyp) + 1))
            if op == 0 and hyp:
                hyp[min(pos, len(hyp) - 1)] = alphabet[int(rs.randint(0, len(alphabet)))]
            elif op == 1:
                hyp.insert(pos, alphabet[int(rs.randint(0, len(alphabet)))])
            elif hyp:
                del hyp[min(pos, len(hyp) - 1)]
        add(hyp, ref)
    with open(os.path.join(HERE, "cer.json"), "w") as f:
        json.dump(cases, f, ensure_ascii=True)
    print("cer.json", len(cases))


if __name__ == "__main__":
    if "--cer-only" in sys.argv:
        gen_cer()
        sys.exit(0)
    if "--labels-only" in sys.argv:
        gen_dataset_labels()
        sys.exit(0)
    if "--train-step-only" in sys.argv:
        gen_train_step()
        sys.exit(0)
    if "--harness-only" not in sys.argv:
        gen_core()
        gen_e2e()
        gen_emission()
        gen_head()
        gen_frames()
        gen_losses()
        gen_dataset_labels()
        gen_cer()
    gen_harness()
    gen_train_step()
    leftovers = [d for d, _, fs in os.walk(REF) if d.endswith("__pycache__")]
    assert not leftovers, f"bytecode written into the reference tree: {leftovers}"
    print("done")
