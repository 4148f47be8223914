"""Full-depth parity of the HIP path against the CPU oracle (north_star: "encoder / CTC log-probs within 1e-3 fp32,
onset / offset frames identical"), at the BASELINE architectures themselves rather than toy widths:
  * Whisper-medium, all 24 blocks, one 30 s clip: float32 within 1e-3 (encoder output, logits, CTC emissions) and the
    same seconds as the oracle's own end-to-end run; bfloat16 / float16 throughput modes: emission error bounded and
    recorded, exact-match rate of the onset / offset boundaries on peaked emissions;
  * Whisper-tiny (BASELINE configs[0]: d = 384, 6 heads, 4 blocks), float32, the SURVEY 8(d) cfg-1 inputs (3.756 s with
    11 labels, 30 s with 26 labels): align() seconds == the oracle's perform_viterbi_ctc on the oracle's logits;
  * Whisper-large-v2 (configs[3]), all 32 blocks, one clip, float16.
The random-init encoders are the architecture's own sizes; the head's Linear is scaled so that the frame posteriors are
peaked (as a trained head's are) -- near-flat emissions would let 1e-6 rounding decide ties, which is not what the path is
for (tie-breaking itself is covered bit-exactly on identical emissions in test_gpu_viterbi.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VOCAB = 21129


def _wave(n, seed=0):
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    return (rs.randn(n) * 0.05 + 0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t * (1 + 0.1 * t))).astype(np.float32)


def _head_init(model, hidden, fc_scale, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.align_rnn.named_parameters():
            s = fc_scale if n.startswith("fc.weight") else 1.5
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (s / hidden ** 0.5))


def _build(name, dtype, wm=None, fc_scale=12.0):
    from lyricalignment_amd import whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    dims = wc.dims_for(name)
    wm = wm if wm is not None else wc.build_model(name, seed=3)
    model = AlignModel(wm, embed_dim=dims.n_audio_state, hidden_dim=384, output_dim=VOCAB, device="cuda", compute_dtype=dtype).eval()
    _head_init(model, 384, fc_scale, 7)
    return model, dims


def _oracle_run(model, dims, audio, labels):
    """The oracle's own end-to-end result on one clip: (encoder output, logits, lp, ls, seconds, T)."""
    from oracle import alignment_oracle as ao, model_oracle as mo
    p = {"encoder." + k: v.detach().float().cpu() for k, v in model.whisper_model.encoder.state_dict().items()}
    p.update({"align_rnn." + k: v.detach().float().cpu() for k, v in model.align_rnn.state_dict().items()})
    mel = mo.pad_or_trim(mo.log_mel_spectrogram(audio[None]), 3000)
    T = mo.frame_count(len(audio) // 160)
    with torch.no_grad():
        enc = mo.encoder_forward(p, mel, n_head=dims.n_audio_head)
        logits = mo.gru_head_forward(p, enc[:, :T])
        lp, ls = mo.emission_prep_ctc(logits)
    return enc, logits, lp, ls, ao.perform_viterbi_ctc(logits, labels), T


def _device_emissions(model, audio, labels):
    from lyricalignment_amd import _lib
    from lyricalignment_amd.utils import alignment as ua
    eng = model.engine()
    lab_dev, n_lab, lists = ua._labels_to_device(labels, 1, eng.device)
    feats, B, T, stride = model._features(model._mel_of([audio]), True)
    return eng.emissions(feats, B, T, stride, lab_dev, n_lab, _lib.LA_VARIANT_CTC).cpu(), lists[0]


def _labels(L, seed=6):
    lab = torch.from_numpy(np.random.RandomState(seed).randint(2, 403, size=(1, L)))
    lab[0, 3] = lab[0, 2]                                   # a repeated label: the CTC lattice must keep the blank between them
    return lab


_flat = lambda r: np.array([x for u in r for seg in u for x in seg])


class _count_launches:
    """Launches of one kernel family of the library inside the block (la_timer_*: the family's launches are counted whether or not they are
    bracketed) -- how a test knows which route a call took."""

    def __init__(self, family):
        self.family, self.n = family, 0

    def __enter__(self):
        from lyricalignment_amd import _lib
        L = _lib.lib()
        L.la_timer_reset(); L.la_timer_sample(1000003); L.la_timer_enable(self.family.encode())
        return self

    def __exit__(self, *exc):
        import ctypes
        from lyricalignment_amd import _lib
        L = _lib.lib()
        torch.cuda.synchronize()
        L.la_timer_disable()
        ms, timed, work, seen = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_int64(0)
        L.la_timer_read_work(ctypes.byref(ms), ctypes.byref(timed), ctypes.byref(work), ctypes.byref(seen))
        self.n = int(seen.value)
        L.la_timer_reset(); L.la_timer_sample(1)
        return False


# The float32 mode has two routes (include/lyricalign.h option "x2_inference"): the float32-MFMA kernels, and -- from the batch size on at
# which every Linear of a block fills the 256 x 256 kernel (9 clips at d = 1024, 17 at d = 384) -- the f16 matrix pipe at float32 accuracy
# (three f16 products per float32 product, csrc/la_f32x2.hip).  The oracle runs ONE clip; for the second route the device gets that clip
# `copies` times in one batch (clips are independent in every stage after the log-mel, whose clamp at the batch maximum is the same for
# identical clips) and the first and the last copy are held to the oracle.
ROUTES = [("f32_mfma", 1), ("f16x2", 12)]


@pytest.fixture(scope="module")
def medium():
    """Whisper-medium random-init weights + the oracle's result for one 30 s clip (computed once: ~3 s of CPU)."""
    from lyricalignment_amd import whisper_compat as wc
    wm = wc.build_model("medium", seed=3)
    model, dims = _build("medium", torch.float32, wm)
    audio, labels = _wave(480000, 5), _labels(26)
    return dict(wm=wm, dims=dims, audio=audio, labels=labels, ref=_oracle_run(model, dims, audio, labels))


@pytest.mark.parametrize("route,copies", ROUTES)
def test_medium_24_blocks_float32_within_1e3_of_oracle(medium, route, copies):
    """(a) of the full-depth list: every float the path hands on -- encoder output, align logits, CTC emissions -- within
    1e-3 of the oracle after all 24 blocks, and the seconds of the fused path and of the two-step drop-in path equal the
    oracle's own end-to-end result exactly (utils/alignment.py:121-188; module/align_model.py:72-123) -- on both routes of the float32
    mode (ROUTES above)."""
    from lyricalignment_amd import _lib
    from lyricalignment_amd.utils import alignment as ua
    from oracle import model_oracle as mo
    model, dims = _build("medium", torch.float32, medium["wm"])
    audio, labels = medium["audio"], medium["labels"]
    enc, logits, lp, ls, secs, T = medium["ref"]
    assert T == 1500
    audios, labs = [audio] * copies, labels.repeat(copies, 1)
    with torch.no_grad(), _count_launches("gemm_f16x2") as x2:
        mel1 = mo.pad_or_trim(mo.log_mel_spectrogram(audio[None]), 3000).cuda()
        ours_enc = model.whisper_model.embed_audio(mel1.repeat(copies, 1, 1)).cpu()
        lg, _ = model.frame_manual_forward(audios)
        got = model.align(audios, labs, use_ctc=True)
        two = ua.perform_viterbi_ctc(lg, labs)
        eng = model.engine()
        lab_dev, n_lab, lists = ua._labels_to_device(labs, copies, eng.device)
        feats, B, Tn, stride = model._features(model._mel_of(audios), True)
        em = eng.emissions(feats, B, Tn, stride, lab_dev, n_lab, _lib.LA_VARIANT_CTC).cpu()
        lab = lists[0]
    # the route under test is the one that ran: 4 Linear launches per block and call on the f16x2 route, none on the other
    assert (x2.n >= 4 * 24 * 4) if route == "f16x2" else (x2.n <= 4), x2.n      # (one clip: only the output Linear [1500 x 21129] fills the 256 x 256 kernel)
    idx = torch.tensor(lab) - 1
    for c in sorted({0, copies - 1}):
        np.testing.assert_allclose(ours_enc[c].numpy(), enc[0].numpy(), rtol=0, atol=1e-3)
        np.testing.assert_allclose(lg[c].cpu().numpy(), logits[0].numpy(), rtol=0, atol=1e-3)
        np.testing.assert_allclose(em[c, :, 0].numpy(), ls[0, :, 0].numpy(), rtol=0, atol=1e-3)
        np.testing.assert_allclose(em[c, :, 1:1 + len(lab)].numpy(), lp[0][:, idx].numpy(), rtol=0, atol=1e-3)
    assert got == secs * copies and two == secs * copies


@pytest.mark.parametrize("route,copies", [("f32_mfma", 1), ("f16x2", 6)])
def test_medium_24_blocks_plain_variant_and_batch_of_two_equal_oracle(medium, route, copies):
    """The non-CTC variant (perform_viterbi, utils/alignment.py:13-71: log_softmax over all V, silence = column 0) at full
    depth, float32, on a batch of two clips of different lengths (ragged T via get_orig_len): seconds of the fused path and of the
    two-step drop-in path == the oracle's perform_viterbi on the oracle's logits.  f16x2 route: the pair of clips six times in one batch."""
    from lyricalignment_amd.utils import alignment as ua
    from oracle import alignment_oracle as ao, model_oracle as mo
    model, dims = _build("medium", torch.float32, medium["wm"])
    audios = [medium["audio"][: 16000 * 7 + 333], _wave(16000 * 7 + 333 - 4000, 9)]
    labels = torch.full((2, 9), -100, dtype=torch.long)
    labels[0] = torch.from_numpy(np.random.RandomState(11).randint(1, 403, size=9))
    labels[1, :5] = torch.from_numpy(np.random.RandomState(12).randint(1, 403, size=5))
    labels[0, 5] = labels[0, 4]
    p = {"encoder." + k: v.detach().float().cpu() for k, v in model.whisper_model.encoder.state_dict().items()}
    p.update({"align_rnn." + k: v.detach().float().cpu() for k, v in model.align_rnn.state_dict().items()})
    n = len(audios[0])
    batch = np.zeros((2, n), dtype=np.float32)
    batch[0] = audios[0]; batch[1, : len(audios[1])] = audios[1]
    T = mo.frame_count(n // 160)
    with torch.no_grad():
        ref_logits = mo.gru_head_forward(p, mo.encoder_forward(p, mo.pad_or_trim(mo.log_mel_spectrogram(batch), 3000), n_head=dims.n_audio_head)[:, :T])
        want = ao.perform_viterbi(ref_logits, labels)
        with _count_launches("gemm_f16x2") as x2:
            lg, _ = model.frame_manual_forward(audios * copies)
            got = model.align(audios * copies, labels.repeat(copies, 1), use_ctc=False)
            two = ua.perform_viterbi(lg, labels.repeat(copies, 1))
    assert (x2.n >= 4 * 24 * 2) if route == "f16x2" else (x2.n <= 4), x2.n      # (one clip: only the output Linear [1500 x 21129] fills the 256 x 256 kernel)
    for c in sorted({0, copies - 1}):
        np.testing.assert_allclose(lg[2 * c: 2 * c + 2].cpu().numpy(), ref_logits.numpy(), rtol=0, atol=1e-3)
    assert got == want * copies and two == want * copies


# Bounds calibrated on MI355X with tools/depth_parity.py (profiles/r2_depth_parity.json holds the measured table):
# emission error of the 16-bit throughput modes after 24 blocks against the fp32 oracle, and the share of the 52 onset /
# offset boundaries that equal the oracle's own end-to-end result.
# Measured (r2): bfloat16 mean 0.0216 / max 0.109, float16 mean 0.0029 / max 0.0152 on logits of magnitude <= 20; all 52 boundaries
# equal the oracle's in both modes.  The bounds leave ~3x for other boxes / seeds.
@pytest.mark.parametrize("dtype,em_mean_tol,em_max_tol,min_exact", [(torch.bfloat16, 0.06, 0.35, 0.9), (torch.float16, 0.01, 0.06, 0.95)])
def test_medium_24_blocks_16bit_emission_error_and_boundary_match(medium, dtype, em_mean_tol, em_max_tol, min_exact):
    model, dims = _build("medium", dtype, medium["wm"])
    audio, labels = medium["audio"], medium["labels"]
    enc, logits, lp, ls, secs, T = medium["ref"]
    with torch.no_grad():
        got = model.align([audio], labels, use_ctc=True)
        em, lab = _device_emissions(model, audio, labels)
    idx = torch.tensor(lab) - 1
    err = torch.cat([(em[0, :, 1:1 + len(lab)] - lp[0][:, idx]).abs().flatten(), (em[0, :, 0] - ls[0, :, 0]).abs()])
    exact = float(np.mean(_flat(got) == _flat(secs)))
    dev = float(np.abs(_flat(got) - _flat(secs)).max())
    print(f"{dtype}: emission error mean {float(err.mean()):.4f} max {float(err.max()):.4f}; boundaries exact {exact:.3f}, max deviation {dev:.2f} s")
    assert float(err.mean()) < em_mean_tol and float(err.max()) < em_max_tol
    assert exact >= min_exact, (exact, dev)


@pytest.mark.parametrize("route,copies", [("f32_mfma", 1), ("f16x2", 18)])
@pytest.mark.parametrize("n_samples,L", [(60096, 11), (480000, 26)])
def test_tiny_dims_cfg1_align_equals_oracle(n_samples, L, route, copies):
    """BASELINE configs[0] on the HIP path: Whisper-tiny (d = 384, 6 heads, 4 blocks), float32, SURVEY 8(d) cfg-1 inputs.
    align() seconds must equal alignment_oracle.perform_viterbi_ctc(oracle logits); logits within 1e-3.  Both routes of the float32 mode
    (ROUTES above; d = 384 fills the 256 x 256 kernel from 17 clips on)."""
    from lyricalignment_amd.utils import alignment as ua
    model, dims = _build("tiny", torch.float32)
    assert (dims.n_audio_state, dims.n_audio_head, dims.n_audio_layer) == (384, 6, 4)
    audio, labels = _wave(n_samples, 0), _labels(L, seed=1)
    enc, logits, lp, ls, secs, T = _oracle_run(model, dims, audio, labels)
    assert T == (188 if n_samples == 60096 else 1500)
    with torch.no_grad(), _count_launches("gemm_f16x2") as x2:
        lg, _ = model.frame_manual_forward([audio] * copies)
        got = model.align([audio] * copies, labels.repeat(copies, 1), use_ctc=True)
        two = ua.perform_viterbi_ctc(lg.cpu(), labels.repeat(copies, 1))
    assert (x2.n >= 4 * 4 * 2) if route == "f16x2" else (x2.n <= 4), x2.n      # (one clip: only the output Linear [1500 x 21129] fills the 256 x 256 kernel)
    for c in sorted({0, copies - 1}):
        np.testing.assert_allclose(lg[c].cpu().numpy(), logits[0].numpy(), rtol=0, atol=1e-3)
    assert got == secs * copies and two == secs * copies


def test_large_v2_32_blocks_float16_one_clip():
    """BASELINE configs[3] at full depth: Whisper-large-v2 (d = 1280, 20 heads, 32 blocks), float16 MFMA path, one 30 s
    clip: emissions against the fp32 oracle (bounds from tools/depth_parity.py), frames bit-exact on the device's own
    emissions, boundary match against the oracle's end-to-end result."""
    from oracle import alignment_oracle as ao
    model, dims = _build("large-v2", torch.float16)
    assert (dims.n_audio_state, dims.n_audio_head, dims.n_audio_layer) == (1280, 20, 32)
    audio, labels = _wave(480000, 5), _labels(26)
    enc, logits, lp, ls, secs, T = _oracle_run(model, dims, audio, labels)
    with torch.no_grad():
        on, off, score, status = model.align([audio], labels, use_ctc=True, return_frames=True)
        got = model.align([audio], labels, use_ctc=True)
        em, lab = _device_emissions(model, audio, labels)
    assert int(status[0]) == 0
    rc, on_o, off_o, sc_o = ao.align_frames_compact(em[0].numpy(), np.array(lab))
    assert rc == 0 and on.cpu().numpy()[0, :26].tolist() == on_o.tolist() and off.cpu().numpy()[0, :26].tolist() == off_o.tolist()
    assert float(score[0]) == sc_o
    idx = torch.tensor(lab) - 1
    err = torch.cat([(em[0, :, 1:27] - lp[0][:, idx]).abs().flatten(), (em[0, :, 0] - ls[0, :, 0]).abs()])
    exact = float(np.mean(_flat(got) == _flat(secs)))
    print(f"large-v2 f16: emission error mean {float(err.mean()):.4f} max {float(err.max()):.4f}; boundaries exact {exact:.3f}")
    assert float(err.mean()) < 0.015 and float(err.max()) < 0.1
    assert exact >= 0.9


def test_large_v2_32_blocks_float32_on_the_f16x2_route_equals_oracle():
    """BASELINE configs[3]'s architecture (d = 1280, 20 heads, 32 blocks: K = 1280 and 5120) in the reference's precision on the f16 matrix pipe:
    nine copies of one 30 s clip in a batch (every Linear fills the 256 x 256 kernel), logits within 1e-3 of the fp32 oracle after 32 blocks and
    the seconds of the fused path equal to the oracle's own end-to-end result."""
    model, dims = _build("large-v2", torch.float32)
    audio, labels = _wave(480000, 5), _labels(26)
    enc, logits, lp, ls, secs, T = _oracle_run(model, dims, audio, labels)
    copies = 9
    with torch.no_grad(), _count_launches("gemm_f16x2") as x2:
        lg, _ = model.frame_manual_forward([audio] * copies)
        got = model.align([audio] * copies, labels.repeat(copies, 1), use_ctc=True)
    assert x2.n >= 4 * 32 * 2, x2.n
    for c in (0, copies - 1):
        np.testing.assert_allclose(lg[c].cpu().numpy(), logits[0].numpy(), rtol=0, atol=1e-3)
    assert got == secs * copies
