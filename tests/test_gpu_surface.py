"""GPU tests of the drop-in surface rows that had no run through the HIP path: AlignModel.forward(mel, y_in) (SURVEY 8 a6,
module/align_model.py:126-152), the evaluation harness glue with the real model (row H, inference_alignment.py:126-180,
inference_alignment_nogt.py:130-178), the INTEGRATION.md recipes executed as written, the multi-rank control flow of
bench.py, and the pipeline's label lifetime."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import load_json

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wave(n, seed=0):
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    return (rs.randn(n) * 0.05 + 0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t * (1 + 0.1 * t))).astype(np.float32)


def _model(dtype=torch.float32, seed=0, vocab=300, dropout=0.15, with_decoder=False, **kw):
    from lyricalignment_amd import whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    dims = wc.ModelDimensions(n_audio_state=128, n_audio_head=2, n_audio_layer=2, n_text_state=128, n_text_head=2, n_text_layer=1,
                              n_vocab=311, n_text_ctx=64)
    wm = wc.build_model(dims=dims, seed=seed, std=0.05, with_decoder=with_decoder)
    model = AlignModel(wm, embed_dim=128, hidden_dim=64, output_dim=vocab, dropout=dropout, device="cuda", compute_dtype=dtype, **kw)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.align_rnn.named_parameters():
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * ((6.0 if n.startswith("fc.weight") else 1.5) / 64 ** 0.5))
    return model.eval()


def _oracle_params(model, double=False, grad=False):
    p = {}
    for k, v in model.state_dict().items():
        key = k[len("whisper_model."):] if k.startswith("whisper_model.") else k
        t = v.detach().cpu().double() if double else v.detach().cpu().float()
        p[key] = t.requires_grad_(grad and "encoder.positional_embedding" not in k)
    return p


# ------------------------------------------------------------------------------------------------ a6: AlignModel.forward
def test_forward_eval_matches_oracle_with_and_without_decoder():
    """AlignModel.forward(mel, y_in) in eval mode (module/align_model.py:126-152): align logits over all 1500 frames and,
    with train_transcript, the decoder logits -- against the oracle, float32 within 1e-3."""
    from oracle import model_oracle as mo
    mel = torch.rand(2, 80, 3000, generator=torch.Generator().manual_seed(5)) * 2 - 1
    tokens = torch.randint(0, 311, (2, 9), generator=torch.Generator().manual_seed(6))
    for freeze in (False, True):
        model = _model(seed=20, with_decoder=True, train_transcript=True, freeze_encoder=freeze)
        with torch.no_grad():
            al, tr = model(mel.cuda(), tokens)
            al2, tr2 = model(mel.cuda())
        p = _oracle_params(model)
        xa = mo.encoder_forward(p, mel, n_head=2)
        np.testing.assert_allclose(al.cpu().numpy(), mo.gru_head_forward(p, xa).numpy(), rtol=0, atol=1e-3)
        np.testing.assert_allclose(tr.cpu().numpy(), mo.decoder_forward(p, tokens, xa, n_head=2).numpy(), rtol=0, atol=1e-3)
        assert tuple(al.shape) == (2, 1500, 300) and tuple(tr.shape) == (2, 9, 311) and tr2 is None and torch.equal(al, al2)
    model = _model(seed=21, train_alignment=False, with_decoder=True, train_transcript=True)
    with torch.no_grad():
        al, tr = model(mel.cuda(), tokens)
    assert al is None and tr is not None


@pytest.mark.parametrize("freeze_encoder", [True, False])
def test_forward_train_mode_freeze_encoder_on_and_off(freeze_encoder):
    """forward() in train mode: `freeze_encoder` puts the encoder under no_grad (:135-139) -- gradients then reach the head
    (and the decoder) only; without it they reach the encoder too.  Values and gradients against torch autograd through the
    float64 oracle (dropout 0 so both sides compute the same function)."""
    from oracle import model_oracle as mo
    F = torch.nn.functional
    model = _model(seed=30, dropout=0.0, with_decoder=True, train_transcript=True, freeze_encoder=freeze_encoder).to("cuda")
    model.train()
    mel = torch.rand(1, 80, 3000, generator=torch.Generator().manual_seed(31)) * 2 - 1
    tokens = torch.randint(0, 311, (1, 7), generator=torch.Generator().manual_seed(32))
    w_al = torch.randn(1, 1500, 300, generator=torch.Generator().manual_seed(33)) / 1500
    w_tr = torch.randn(1, 7, 311, generator=torch.Generator().manual_seed(34)) / 7
    al, tr = model(mel.cuda(), tokens.cuda())
    assert al.requires_grad and tr.requires_grad
    ((al * w_al.cuda()).sum() + (tr * w_tr.cuda()).sum()).backward()
    p = _oracle_params(model, double=True, grad=True)
    xa = mo.encoder_forward(p, mel.double(), n_head=2)
    if freeze_encoder:
        xa = xa.detach()
    ral, rtr = mo.gru_head_forward(p, xa), mo.decoder_forward(p, tokens, xa, n_head=2)
    ((ral * w_al.double()).sum() + (rtr * w_tr.double()).sum()).backward()
    np.testing.assert_allclose(al.detach().cpu().numpy(), ral.detach().numpy(), rtol=0, atol=1e-3)
    np.testing.assert_allclose(tr.detach().cpu().numpy(), rtr.detach().numpy(), rtol=0, atol=1e-3)
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / b.abs().max().clamp_min(1e-12))
    for name, prm in model.align_rnn.named_parameters():
        assert rel(prm.grad, p["align_rnn." + name].grad) < 2e-3, name
    for name, prm in model.whisper_model.decoder.named_parameters():
        assert rel(prm.grad, p["decoder." + name].grad) < 2e-3, name
    for name, prm in model.whisper_model.encoder.named_parameters():
        if freeze_encoder:
            assert prm.grad is None, name
        else:
            assert rel(prm.grad, p["encoder." + name].grad) < 2e-3, name


# ------------------------------------------------------------------------------------------------ H: the harness glue
def _harness_batches(fx, audios_of):
    out = []
    for i, (raw, gt) in enumerate(zip(fx["raw_tokens"], fx["gt"])):
        out.append((audios_of(i, len(raw)), torch.tensor(raw), None, (None,) if gt is None else tuple(gt), None, None))
    return out


def _lut(fx):
    from lyricalignment_amd import harness
    n_tok = max(r for r, _ in fx["token_to_class"]) + 1
    lookup = {f"p{i}": 1 for i in range(n_tok)}
    for r, m in fx["token_to_class"]:
        lookup[f"p{r}"] = m
    return harness.PinyinClassLUT([f"p{i}" for i in range(n_tok)], lookup)


def test_harness_two_step_on_reference_logits_gives_the_reference_average():
    """evaluate_batches(two_step=True) exactly as the reference's loop runs (frame_manual_forward -> perform_viterbi_ctc ->
    get_mae -> mean of per-batch means, inference_alignment.py:159-177) with a real AlignModel object whose
    frame_manual_forward hands out the logits the fixture was generated from (on the device): label LUT, emission prep and
    DP on the HIP kernels, skipping of the (None,) batch -- the average must equal the reference's value to the last bit."""
    from lyricalignment_amd import harness
    fx = load_json("harness.json")
    lg = fx["logits"]
    model = _model(seed=40, vocab=lg["shape_tail"][1])
    calls = {"i": 0}

    def frame_manual_forward(audios, y_in=None, get_orig_len=True):
        rs = np.random.RandomState(lg["seed_base"] + calls["i"])      # the reference forwards evaluated batches only: seeds advance with them
        calls["i"] += 1
        return torch.from_numpy((rs.randn(len(audios), *lg["shape_tail"]) * lg["scale"]).astype(np.float32)).cuda(), None

    model.frame_manual_forward = frame_manual_forward
    batches = _harness_batches(fx, lambda i, B: (np.zeros(16000, dtype=np.float32),) * B)
    avg, maes = harness.evaluate_batches(model, batches, _lut(fx), use_ctc_loss=True, two_step=True)
    assert maes[3] is None and sum(m is not None for m in maes) == 4 and calls["i"] == 4
    assert avg == fx["avg_mae"] == 0.540436507936508


def test_harness_and_align_records_with_the_hip_model_match_the_oracle_pipeline():
    """evaluate_batches (fused align() and two-step) and align_records with the real HIP AlignModel on real audio, against
    the same glue computed from the oracle's pipeline (oracle log-mel / encoder / head -> oracle perform_viterbi_ctc ->
    get_mae, mean of per-batch means): identical averages, identical [[onset, offset, char], ...] records."""
    from types import SimpleNamespace
    from lyricalignment_amd import harness
    from oracle import alignment_oracle as ao, model_oracle as mo
    fx = load_json("harness.json")
    model = _model(seed=41, vocab=404)
    lut = _lut(fx)
    audios_of = lambda i, B: tuple(_wave(30000 + 800 * i + 1600 * b, 50 + 3 * i + b) for b in range(B))
    batches = _harness_batches(fx, audios_of)
    p = _oracle_params(model)

    def oracle_align(audios, labels):
        n = max(map(len, audios))
        batch = np.zeros((len(audios), n), dtype=np.float32)
        for i, a in enumerate(audios):
            batch[i, : len(a)] = a
        mel = mo.pad_or_trim(mo.log_mel_spectrogram(batch), 3000)
        T = mo.frame_count(n // 160)
        return ao.perform_viterbi_ctc(mo.gru_head_forward(p, mo.encoder_forward(p, mel, n_head=2)[:, :T]), labels)

    want = []
    for audios, tokens, _, gt, _, _ in batches:
        if gt == (None,):
            want.append(None)
            continue
        res = oracle_align(audios, lut(tokens))
        # the fixture's ground truth has one [onset, offset] per label of ITS clips; reuse it as the (arbitrary) target
        want.append(mo.get_mae(gt, res))
    done = [m for m in want if m is not None]
    total = 0
    for m in done:
        total += m
    avg_fused, maes_fused = harness.evaluate_batches(model, batches, lut, use_ctc_loss=True)
    avg_two, maes_two = harness.evaluate_batches(model, batches, lut, use_ctc_loss=True, two_step=True)
    assert maes_fused == want and maes_two == want
    assert avg_fused == avg_two == total / len(done)
    # inference_alignment_nogt.py:130-178: one record at a time, BERT ids without [CLS] / [SEP], [[onset, offset, char], ...]
    ids_of = {"abcdefg": [r for r, _ in fx["token_to_class"][:7]], "xyz": [r for r, _ in fx["token_to_class"][7:10]]}
    records = [SimpleNamespace(audio=_wave(40000, 60), text="abcdefg"), SimpleNamespace(audio=_wave(25000, 61), text="xyz")]
    got = harness.align_records(model, records, lut, tokenize=lambda text: ids_of[text], use_ctc_loss=True)
    for rec, out in zip(records, got):
        ref = oracle_align([rec.audio], lut(torch.tensor([ids_of[rec.text]])))[0]
        assert out == [[ref[j][0], ref[j][1], rec.text[j]] for j in range(len(rec.text))]


# ------------------------------------------------------------------------------------------------ INTEGRATION.md as written
def _code_blocks(md_path):
    with open(md_path) as f:
        return re.findall(r"```python\n(.*?)```", f.read(), flags=re.S)


def test_integration_recipe_1_module_aliases_run_the_reference_import_lines():
    """INTEGRATION.md section 1, executed as written in a fresh interpreter: after the sys.modules aliases, the reference's
    own import lines (inference_alignment.py:21-23) resolve to the HIP-backed objects and align a clip."""
    block = next(b for b in _code_blocks(os.path.join(ROOT, "INTEGRATION.md")) if 'sys.modules["module.align_model"]' in b)
    script = f"import sys\nsys.path.insert(0, {ROOT!r})\n" + block + r'''
from module.align_model import AlignModel
from utils.alignment import perform_viterbi_ctc, perform_viterbi, get_mae
import numpy as np, torch, lyricalignment_amd
from lyricalignment_amd import whisper_compat as wc
assert AlignModel is lyricalignment_amd.module.align_model.AlignModel
dims = wc.ModelDimensions(n_audio_state=128, n_audio_head=2, n_audio_layer=1, n_text_state=128, n_text_head=2, n_text_layer=0)
model = AlignModel(wc.build_model(dims=dims, seed=0, std=0.05), embed_dim=128, hidden_dim=64, output_dim=50, device="cuda").eval()
audio = (np.random.RandomState(0).randn(32000) * 0.1).astype(np.float32)
with torch.no_grad():
    logits, _ = model.frame_manual_forward([audio])
res = perform_viterbi_ctc(logits.cpu(), torch.tensor([[3, 9, 9, 20]]))
assert len(res) == 1 and len(res[0]) == 4 and get_mae(res, res) == 0.0
print("ALIAS_OK", res[0][0])
'''
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode == 0 and "ALIAS_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_integration_recipe_2_ctypes_snippet_equals_oracle_and_golden():
    """the ctypes perform_viterbi_ctc stub a maintainer of the reference would paste, executed
    verbatim (only the library path is made absolute) on the reference's golden logits: same seconds, same exceptions."""
    from conftest import load_npz
    from lyricalignment_amd import _lib
    block = next(b for b in _code_blocks(os.path.join(ROOT, "INTEGRATION.md")) if "def perform_viterbi_ctc" in b)
    assert '"liblyricalign_hip.so"' in block
    ns = {}
    exec(block.replace('"liblyricalign_hip.so"', repr(_lib.LIB_PATH)), ns)
    z = load_npz("viterbi_e2e.npz")
    n_checked = 0
    for m in json.loads(bytes(z["meta_json"]).decode()):
        if m["variant"] != "ctc" or m["scale"] < 1.0:
            continue
        rs = np.random.RandomState(m["seed"])
        logits = torch.from_numpy((rs.randn(m["B"], m["T"], m["V"]) * m["scale"]).astype(np.float32))
        res = ns["perform_viterbi_ctc"](logits, m["labels"])
        for b in range(m["B"]):
            assert res[b] == z[f"{m['name']}/{b}/seconds"].tolist(), m["name"]
            n_checked += 1
    assert n_checked >= 3
    for e in load_json("viterbi_errors.json"):
        lg = torch.from_numpy(np.random.RandomState(e["seed"]).randn(1, e["T"], e["V"]).astype(np.float32))
        if e["raises"] is None:
            assert ns["perform_viterbi_ctc"](lg, e["labels"]) == e["result"]
        else:
            with pytest.raises({"ValueError": ValueError, "IndexError": IndexError}[e["raises"]]):
                ns["perform_viterbi_ctc"](lg, e["labels"])


# ------------------------------------------------------------------------------------------------ multi-rank control flow
def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def test_bench_two_ranks_on_one_device_runs_the_multi_rank_control_flow():
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank), as a fresh child
    process: both ranks on device 0 over gloo (LA_BENCH_SAME_DEVICE / LA_BENCH_DIST_BACKEND), so no speed-up is expected --
    the point is that init_process_group, the barriers, the MAX all-reduce of the time and the rank-0 JSON line run."""
    env = dict(os.environ, LA_BENCH_SAME_DEVICE="1", LA_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]                    # rank 0 prints, rank 1 does not
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["cpu_baseline"] is None and out["steps"] == 2 and out["scaling"] == "weak"
    assert np.isfinite(out["value"]) and out["value"] > 0 and out["roofline"]["frac"] > 0
    assert abs(out["value"] - 2 * 32 * 30.0 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]   # whole-job aggregate over both ranks
    assert "rank 1/2" in r.stderr and "rank 0/2" in r.stderr


def test_finetune_bench_two_ranks_run_the_gradient_allreduce():
    """bench.py --mode finetune with 2 ranks over gloo on one device: the data-parallel step (micro-steps, ONE all-reduce per
    bucket, clip + AdamW) runs end to end and reports the time of the collective (train_multitask.py:337-340 is where it sits)."""
    env = dict(os.environ, LA_BENCH_SAME_DEVICE="1", LA_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "finetune", "--model", "tiny",
           "--steps", "2", "--warmup", "1", "--accum", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["mode"] == "finetune" and out["value"] > 0
    assert out["allreduce_ms_per_step"] >= 0 and out["config"]["grad_bytes_per_step"] > 0
    assert out["allreduce_GBps"] is None or out["allreduce_GBps"] > 0        # bus bandwidth of the one exchange step


@pytest.mark.parametrize("extra", [[], ["--mode", "finetune", "--model", "tiny", "--accum", "2"]], ids=["align", "finetune"])
def test_plain_bench_gpus_2_starts_its_own_two_ranks(extra):
    """`python3 bench.py --gpus 2 ...` with NO launcher around it (how the driver's scaling run may call it): the parent makes
    no GPU call, starts torch.distributed.run as a child with the same arguments and exits with its code; the ranks assert
    world size == --gpus and rank 0 prints the one JSON line with n_gpus = 2."""
    env = dict(os.environ, LA_BENCH_SAME_DEVICE="1", LA_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert "starting 2 ranks" in r.stderr and "rank 1/2" in r.stderr and "rank 0/2" in r.stderr


@pytest.mark.parametrize("extra", [[], ["--mode", "finetune", "--model", "tiny", "--accum", "2"]], ids=["align", "finetune"])
def test_plain_bench_three_ranks_on_one_device_start_up_and_finish_in_bounded_time(extra):
    """Dress rehearsal of the driver's widest launch on a one-GPU box: plain `python3 bench.py --gpus 3` (the parent starts its own
    ranks), all three on device 0 over gloo.  Round-4 verdict item 4 asked for eight; this pool admits at most SIX processes on a card
    at once and kills the whole run at a seventh: it happened with six ranks + this test process (which holds the GPU open too), and
    again with five inside the full suite, where the ranks of the test before were still closing their contexts.  A killed run loses the whole GPU test
    tier, so three it is (3 + this process + up to 2 stragglers = 6) -- the control flow is the same at any N:
    one JSON line from rank 0 with n_gpus = 3, every rank's start-up logged, the synthetic weights built cooperatively (each rank
    1 / 3 of the tensors, exchanged through /dev/shm: whisper_compat.build_model_shared) rather than three full host builds, and the whole
    command inside a wall-time budget that leaves the driver's scaling run room (150 s on the GPU box's 16 cores)."""
    import time
    env = dict(os.environ, LA_BENCH_SAME_DEVICE="1", LA_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", *extra]
    time.sleep(3.0)                       # (let the ranks of the test before finish closing their GPU contexts)
    t0 = time.perf_counter()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    wall = time.perf_counter() - t0
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["steps"] == 2 and out["value"] > 0
    for rk in range(3):
        assert f"rank {rk}/3" in r.stderr
    assert r.stderr.count("weights built in") >= 3
    print(f"3 ranks on one device {extra}: wall {wall:.1f} s")
    assert wall < 150.0, f"{wall:.1f} s"
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("la_weights_")]      # the exchange files are gone


@pytest.mark.parametrize("extra", [[], ["--mode", "finetune", "--model", "tiny", "--accum", "2"]], ids=["align", "finetune"])
def test_bench_process_group_over_rccl_at_world_size_one(extra):
    """The N > 1 runs form their process group over RCCL (backend "nccl", device_id = the rank's GPU) and call barrier() and
    all_reduce(MAX) around the timed region; a one-GPU box cannot hold two RCCL ranks, so this drives exactly those calls
    with ONE rank under the launcher (LA_BENCH_FORCE_DIST=1): a wrong keyword, a CPU tensor handed to RCCL or a missing
    HSA_ENABLE_IPC_MODE_LEGACY would fail here instead of in the first 8-GPU run."""
    env = dict(os.environ, LA_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LA_BENCH_SAME_DEVICE", "LA_BENCH_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", *extra]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["value"] > 0
    if not extra:
        # the default run's extra legs (round 6): the other compute modes of the headline workload and BASELINE configs[2..4], measured in the same
        # process after the headline is complete -- present, error-free, and the float32-parity mode ran on the f16x2 GEMM family
        modes, others = out["modes"], out["other_configs"]
        assert modes["f32_parity"]["status_ok"] and modes["f32_parity"]["gemm_family"] == "gemm_f16x2" and modes["f32_parity"]["ms_per_step"] > 0
        assert modes["f16"]["status_ok"] and modes["f16"]["ms_per_step"] > 0
        eq = modes["boundaries_equal_to_f32_parity"]
        assert eq["f16"] >= eq["bf16_headline"] >= 0.9 * eq["of"]
        for name in ("longform", "largev2", "finetune"):
            if str(others[name].get("error", "")).startswith("skipped"):          # (a box slow enough to use up the legs' time budget: reported, not a failure)
                continue
            assert "error" not in others[name] and others[name]["ms_per_step"] > 0 and 0 < others[name]["roofline_frac"] < 1, (name, others[name])


@pytest.mark.parametrize("mode_args", [["--mode", "longform", "--songs", "4"], ["--mode", "largev2", "--clips", "64"]])
def test_other_config_bench_modes_two_ranks(mode_args):
    """BASELINE configs[4] (long form) and configs[3] (large-v2 float16) through bench.py with 2 ranks over gloo on one device:
    the dress rehearsal of the driver's N > 1 launch for the modes the align / finetune tests above do not cover (songs / clips
    sharded over ranks, no data-path collective, whole-job aggregate in `value`)."""
    env = dict(os.environ, LA_BENCH_SAME_DEVICE="1", LA_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", *mode_args, "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    out = json.loads(lines[0])
    units, unit_s = (4, 180.0) if mode_args[1] == "longform" else (64, 30.0)
    assert out["n_gpus"] == 2 and out["config"]["mode"] == mode_args[1] and out["scaling"] == "weak" and out["cpu_baseline"] is None
    assert abs(out["value"] - 2 * units * unit_s * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    assert out["roofline"]["frac"] > 0


# ------------------------------------------------------------------------------------------------ pipeline label lifetime
def test_pipelined_aligner_with_fresh_label_tensors_per_submit():
    """A streaming caller allocates its label tensors per batch and drops them right after submit(); the head of a group
    reads them on the head stream much later.  PipelinedAligner records them on that stream, so the caching allocator cannot
    hand their blocks to the next batch's labels first: results must equal the single-stream reference per batch."""
    from lyricalignment_amd.engine import PipelinedAligner
    model = _model(torch.bfloat16, seed=70)
    eng = model.engine()
    rs = np.random.RandomState(71)
    host = []
    for i in range(8):
        host.append((rs.uniform(-1, 1, size=(3, 80, 3000)).astype(np.float32), rs.randint(1, 299, size=(3, 9)).astype(np.int32),
                     np.array([9, 1 + i % 8, 2 + i % 5], dtype=np.int32)))
    with torch.no_grad():
        ref = []
        for mel, lab, nl in host:
            ref.append(tuple(t.clone() for t in eng.align_mel(torch.from_numpy(mel).cuda(), torch.from_numpy(lab).cuda(),
                                                              torch.from_numpy(nl).cuda(), n_frames=700)))
        torch.cuda.synchronize()
        pipe = PipelinedAligner(eng, head_group=3)
        outs = []
        for mel, lab, nl in host:
            m, l, n = torch.from_numpy(mel).cuda(), torch.from_numpy(lab).cuda(), torch.from_numpy(nl).cuda()
            outs.append(pipe.submit(m, l, n, n_frames=700))
            del m, l, n                                                    # the caller's references are gone before the head runs
            junk = torch.randint(1, 299, (3, 9), dtype=torch.int32, device="cuda")   # would reuse the freed label block
            del junk
        pipe.drain()
    for r, o in zip(ref, outs):
        for a, b in zip(r, o):
            assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------ model-level C entry points
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_model_level_c_entry_points_equal_the_python_sequencing(dtype, monkeypatch):
    """la_encoder_forward / la_align_head_forward (csrc/la_model.cpp: one C call per stage, caller workspace) enqueue the same
    kernels in the same order as the op-by-op sequencing in engine.py (LA_ENGINE_PY=1): bit-identical encoder rows, frames,
    scores; also with the head sliced over clips inside the C call (option head_clip_cap) and on the long-form clip stride."""
    from lyricalignment_amd import _lib, engine as eng_mod
    model = _model(dtype, seed=80)
    eng = model.engine()
    rs = np.random.RandomState(81)
    B = 5
    mel = torch.from_numpy(rs.uniform(-1, 1, size=(B, 80, 3000)).astype(np.float32)).cuda()
    labels = torch.from_numpy(rs.randint(1, 299, size=(B, 9)).astype(np.int32)).cuda()
    n_labels = torch.tensor([9, 5, 2, 7, 1], dtype=torch.int32).cuda()
    with torch.no_grad():
        monkeypatch.setattr(eng_mod, "ENGINE_PY", True)
        enc_py = eng.encode(mel, out_dtype=torch.float32).clone()
        res_py = [t.clone() for t in eng.align_mel(mel, labels, n_labels, n_frames=700)]
        monkeypatch.setattr(eng_mod, "ENGINE_PY", False)
        enc_c = eng.encode(mel, out_dtype=torch.float32).clone()
        res_c = [t.clone() for t in eng.align_mel(mel, labels, n_labels, n_frames=700)]
        with _lib.option("head_clip_cap", 2):
            res_sliced = [t.clone() for t in eng.align_mel(mel, labels, n_labels, n_frames=700)]
        feats = eng.encode(mel)
        two = eng.align_feats(feats, 2, 2900, 3000, labels[:2], n_labels[:2].contiguous(), 1)       # 2 "songs" of 2 chunks each
        monkeypatch.setattr(eng_mod, "ENGINE_PY", True)
        two_py = eng.align_feats(feats, 2, 2900, 3000, labels[:2], n_labels[:2].contiguous(), 1)
    eng.check_gru()
    assert torch.equal(enc_py, enc_c)
    for i, (a, b, c) in enumerate(zip(res_py, res_c, res_sliced)):
        assert torch.equal(a, b)
        if dtype == torch.bfloat16:
            assert torch.equal(a, c)
        elif i == 2:      # float32: launches of <= 4 clips run the recurrence on v_fma_f32, larger ones on f32 MFMA tiles -- same
            np.testing.assert_allclose(c.cpu().numpy(), a.cpu().numpy(), rtol=1e-6)     # sums in another order: scores agree to rounding
    for a, b in zip(two, two_py):
        assert torch.equal(a, b)
    assert int(res_c[3].abs().sum()) == 0


def test_encoder_c_entry_point_takes_the_layernorm_folded_path_at_scale(monkeypatch):
    """From 9 clips at d = 1024 the blocks' LayerNorms are folded into the GEMMs: la_encoder_forward makes that choice itself
    and must equal the Python sequencing of the same folded path bit for bit."""
    from lyricalignment_amd import engine as eng_mod, whisper_compat as wc
    dims = wc.ModelDimensions(n_audio_state=1024, n_audio_head=16, n_audio_layer=2, n_text_state=1024, n_text_head=16, n_text_layer=0)
    wm = wc.build_model(dims=dims, seed=83, std=0.02)
    sd = {"encoder." + k: v for k, v in wm.encoder.state_dict().items()}
    dev = torch.device("cuda")
    e16 = eng_mod.AlignEngine(eng_mod.pack_encoder(sd, 16, torch.bfloat16, dev), None, dev)
    mel = torch.from_numpy(np.random.RandomState(84).uniform(-1, 1, size=(9, 80, 3000)).astype(np.float32)).cuda()
    with torch.no_grad():
        monkeypatch.setattr(eng_mod, "ENGINE_PY", True)
        ref = e16.encode(mel, out_dtype=torch.float32).clone()
        monkeypatch.setattr(eng_mod, "LN_FUSION", False)
        plain = e16.encode(mel, out_dtype=torch.float32).clone()
        monkeypatch.setattr(eng_mod, "LN_FUSION", True)
        monkeypatch.setattr(eng_mod, "ENGINE_PY", False)
        got = e16.encode(mel, out_dtype=torch.float32).clone()
    assert torch.equal(got, ref) and not torch.equal(got, plain)


def test_unidirectional_head_matches_oracle():
    """AlignModel(bidirectional=False) (module/align_model.py:20,48): the head's kernels are bidirectional; the unidirectional GRU
    runs on them with an all-zero reverse direction (engine.pack_head).  Logits against the oracle's unidirectional head,
    float32 within 1e-3; fused align() seconds == the oracle's end-to-end result."""
    from lyricalignment_amd import whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    from oracle import alignment_oracle as ao, model_oracle as mo
    dims = wc.ModelDimensions(n_audio_state=128, n_audio_head=2, n_audio_layer=2, n_text_state=128, n_text_head=2, n_text_layer=0)
    wm = wc.build_model(dims=dims, seed=95, std=0.05)
    model = AlignModel(wm, embed_dim=128, hidden_dim=64, output_dim=300, bidirectional=False, device="cuda").eval()
    g = torch.Generator().manual_seed(96)
    with torch.no_grad():
        for n, p_ in model.align_rnn.named_parameters():
            p_.copy_((torch.rand(p_.shape, generator=g) * 2 - 1) * ((6.0 if n.startswith("fc.weight") else 1.5) / 64 ** 0.5))
    assert "rnn.weight_ih_l0_reverse" not in model.align_rnn.state_dict() and model.align_rnn.fc.weight.shape == (300, 64)
    audio = _wave(60096, 97)
    labels = torch.tensor([[5, 17, 17, 250, 9, 33]])
    with torch.no_grad():
        lg, _ = model.frame_manual_forward([audio])
        got = model.align([audio], labels, use_ctc=True)
    p = _oracle_params(model)
    mel = mo.pad_or_trim(mo.log_mel_spectrogram(audio[None]), 3000)
    ref = mo.gru_head_forward(p, mo.encoder_forward(p, mel, n_head=2)[:, :188], bidirectional=False)
    np.testing.assert_allclose(lg.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-3)
    assert got == ao.perform_viterbi_ctc(ref, labels)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_cpp_consumer_of_the_model_level_c_abi_matches_the_python_path(dtype, tmp_path):
    """examples/align_capi.cpp -- a C++ program with no Python and no torch in it -- drives the hot path through
    la_encoder_forward + la_align_head_forward from flat weight / input files (lyricalignment_amd/capi_export.py) and must
    produce exactly the frames, scores and status AlignEngine.align_mel does."""
    import struct
    from lyricalignment_amd import build as la_build, capi_export
    exe = la_build.build_example()
    assert exe and os.path.exists(exe)
    model = _model(dtype, seed=100)
    eng = model.engine()
    rs = np.random.RandomState(101)
    B, Lmax, frames = 3, 8, 900
    mel = torch.from_numpy(rs.uniform(-1, 1, size=(B, 80, 3000)).astype(np.float32))
    labels = torch.from_numpy(rs.randint(1, 299, size=(B, Lmax)).astype(np.int32))
    n_labels = torch.tensor([8, 3, 5], dtype=torch.int32)
    with torch.no_grad():
        on, off, score, status = [t.cpu() for t in eng.align_mel(mel.cuda(), labels.cuda(), n_labels.cuda(), n_frames=frames, use_ctc=True)]
    wpath, ipath, opath = (str(tmp_path / n) for n in ("weights.bin", "input.bin", "output.bin"))
    capi_export.write_weights(eng, wpath)
    capi_export.write_input(ipath, mel, labels, n_labels, frames, 1)
    r = subprocess.run([exe, wpath, ipath, opath], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    raw = open(opath, "rb").read()
    b2, l2 = struct.unpack_from("<2i", raw, 0)
    assert (b2, l2) == (B, Lmax)
    o = 8
    got_on = np.frombuffer(raw, np.int32, B * Lmax, o).reshape(B, Lmax); o += B * Lmax * 4
    got_off = np.frombuffer(raw, np.int32, B * Lmax, o).reshape(B, Lmax); o += B * Lmax * 4
    got_st = np.frombuffer(raw, np.int32, B, o); o += B * 4
    got_sc = np.frombuffer(raw, np.float64, B, o)
    assert got_st.tolist() == status.tolist() == [0, 0, 0]
    assert got_on.tolist() == on.tolist() and got_off.tolist() == off.tolist()
    assert got_sc.tolist() == score.tolist()
