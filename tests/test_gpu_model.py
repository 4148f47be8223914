"""GPU parity of the assembled path (log-mel -> encoder -> BiGRU head -> emissions -> DP) through the
drop-in Python surface, against the CPU oracle on the same seeded inputs and weights.
Tolerances (north_star): encoder / CTC log-probs within 1e-3 in float32 mode; integer frames bit-exact
given identical emissions.  bfloat16 is the throughput mode: its tolerance is stated where used."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _wave(n, seed=0):
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    return (rs.randn(n) * 0.05 + 0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 3000 * t * (1 + 0.1 * t))).astype(np.float32)


def _small_model(dtype, seed=0, d=128, heads=2, layers=2, hidden=64, vocab=300):
    from lyricalignment_amd import whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    dims = wc.ModelDimensions(n_audio_state=d, n_audio_head=heads, n_audio_layer=layers, n_text_state=d, n_text_head=heads, n_text_layer=1)
    wm = wc.build_model(dims=dims, seed=seed, std=0.05)
    model = AlignModel(wm, embed_dim=d, hidden_dim=hidden, output_dim=vocab, device="cuda", compute_dtype=dtype)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for p in model.align_rnn.parameters():
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * (1.5 / hidden ** 0.5))
    model.eval()
    return model


def _oracle_params(model):
    p = {"encoder." + k: v.detach().float().cpu() for k, v in model.whisper_model.encoder.state_dict().items()}
    p.update({"align_rnn." + k: v.detach().float().cpu() for k, v in model.align_rnn.state_dict().items()})
    return p


def test_log_mel_matches_oracle():
    from lyricalignment_amd.audio_frontend import log_mel_spectrogram
    from oracle import model_oracle as mo
    for n in (60096, 16000, 48160):
        batch = np.stack([_wave(n, 1), _wave(n, 2) * 1e-2])
        ours = log_mel_spectrogram(batch).cpu()
        ref = mo.log_mel_spectrogram(batch)
        assert ours.shape == ref.shape == (2, 80, n // 160)
        np.testing.assert_allclose(ours.numpy(), ref.numpy(), rtol=0, atol=2e-4)  # f32 DFT-by-GEMM vs f32 FFT, log10 domain / 4
    one = log_mel_spectrogram(_wave(480000, 3)).cpu()
    np.testing.assert_allclose(one.numpy(), mo.log_mel_spectrogram(_wave(480000, 3)).numpy(), rtol=0, atol=2e-4)


def test_log_mel_ragged_lengths_and_the_one_call_entry_point():
    """The fused tile kernel at lengths that are not multiples of the 64-frame tile or of the hop (partial last tile, the reflect
    padding at both ends inside ONE tile, a 3-clip batch whose whole-tensor maximum sits in another clip's tile), and
    la_logmel_f32 (constants built inside the call, 3 launches) against la_logmel_f32_prepared (cached constants): same bits."""
    import ctypes
    from lyricalignment_amd import _lib
    from lyricalignment_amd.audio_frontend import _device_tables, log_mel_spectrogram
    from oracle import model_oracle as mo
    for n in (201, 333, 10241, 12345, 64 * 160, 65 * 160 + 159):
        batch = np.stack([_wave(n, 4) * 1e-3, _wave(n, 5), _wave(n, 6) * 0.1])
        ours = log_mel_spectrogram(batch)
        ref = mo.log_mel_spectrogram(batch)
        assert ours.shape == ref.shape == (3, 80, n // 160)
        np.testing.assert_allclose(ours.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-4, err_msg=str(n))
        a = torch.from_numpy(batch).cuda()
        filt, win = _device_tables(torch.cuda.current_device())
        mel = torch.full((3, 80, n // 160 + 5), 7.0, device="cuda")                    # row pitch > frames: the slack stays untouched
        need = ctypes.c_size_t(0)
        _lib.check(_lib.lib().la_logmel_workspace_bytes(3, n, ctypes.byref(need)), "ws")
        ws = torch.empty((need.value,), dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().la_logmel_f32(_lib.ptr(a), 3, n, _lib.ptr(filt), _lib.ptr(win), _lib.ptr(mel), mel.stride(0), mel.stride(1),
                                            _lib.ptr(ws), need.value, _lib.stream_ptr()), "logmel_f32")
        assert torch.equal(mel[:, :, : n // 160], ours) and bool((mel[:, :, n // 160:] == 7.0).all())
    with pytest.raises(ValueError):
        log_mel_spectrogram(_wave(200, 1))                                                # torch.stft refuses the reflect pad there too


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 6e-2), (torch.float16, 8e-3)])
def test_encoder_matches_oracle(dtype, tol):
    from oracle import model_oracle as mo
    model = _small_model(dtype)
    mel = torch.rand(2, 80, 3000, generator=torch.Generator().manual_seed(5)) * 2 - 1
    with torch.no_grad():
        ours = model.whisper_model.embed_audio(mel.cuda()).cpu()
    ref = mo.encoder_forward(_oracle_params(model), mel, n_head=2)
    assert ours.shape == ref.shape == (2, 1500, 128) and ours.dtype == torch.float32
    np.testing.assert_allclose(ours.numpy(), ref.numpy(), rtol=0, atol=tol)   # outputs are LayerNorm-scaled, |x| ~ 1


def test_encoder_matches_hf_transformers():
    """Second, independent reference for the un-pinned third-party encoder: HF WhisperEncoder with the same weights."""
    from transformers import WhisperConfig
    from transformers.models.whisper.modeling_whisper import WhisperEncoder
    model = _small_model(torch.float32, seed=3)
    sd = model.whisper_model.encoder.state_dict()
    cfg = WhisperConfig(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=512, num_mel_bins=80,
                        max_source_positions=1500, decoder_layers=1, decoder_attention_heads=2, decoder_ffn_dim=512)
    cfg._attn_implementation = "eager"
    enc = WhisperEncoder(cfg).eval()
    names = {"attn.query": "self_attn.q_proj", "attn.key": "self_attn.k_proj", "attn.value": "self_attn.v_proj", "attn.out": "self_attn.out_proj",
             "attn_ln": "self_attn_layer_norm", "mlp.0": "fc1", "mlp.2": "fc2", "mlp_ln": "final_layer_norm"}
    hf = {"conv1.weight": sd["conv1.weight"], "conv1.bias": sd["conv1.bias"], "conv2.weight": sd["conv2.weight"], "conv2.bias": sd["conv2.bias"],
          "embed_positions.weight": sd["positional_embedding"], "layer_norm.weight": sd["ln_post.weight"], "layer_norm.bias": sd["ln_post.bias"]}
    for i in range(2):
        for a, b in names.items():
            for wb in ("weight", "bias"):
                if f"blocks.{i}.{a}.{wb}" in sd:
                    hf[f"layers.{i}.{b}.{wb}"] = sd[f"blocks.{i}.{a}.{wb}"]
    missing, unexpected = enc.load_state_dict(hf, strict=False)
    assert not unexpected
    for i in range(2):
        if enc.layers[i].self_attn.k_proj.bias is not None:
            enc.layers[i].self_attn.k_proj.bias.data.zero_()
    mel = torch.rand(1, 80, 3000, generator=torch.Generator().manual_seed(6)) * 2 - 1
    with torch.no_grad():
        ref = enc(mel).last_hidden_state
        ours = model.whisper_model.embed_audio(mel.cuda()).cpu()
    np.testing.assert_allclose(ours.numpy(), ref.numpy(), rtol=0, atol=1e-3)


@pytest.mark.parametrize("n_samples", [60096, 48160, 480000])
def test_frame_manual_forward_matches_oracle_f32(n_samples):
    """Whole reference call chain: log-mel -> pad -> encoder -> truncate (banker's rounding) -> head logits."""
    from oracle import model_oracle as mo
    model = _small_model(torch.float32, seed=7)
    audios = (_wave(n_samples, 8), _wave(n_samples - 700, 9))      # a tuple: the reference would fail on it
    with torch.no_grad():
        logits, tr = model.frame_manual_forward(audios)
    assert tr is None
    T = mo.frame_count(n_samples // 160)
    assert tuple(logits.shape) == (2, T, 300)
    p = _oracle_params(model)
    batch = np.zeros((2, n_samples), dtype=np.float32)
    batch[0] = audios[0]; batch[1, : n_samples - 700] = audios[1]
    mel = mo.pad_or_trim(mo.log_mel_spectrogram(batch), 3000)
    emb = mo.encoder_forward(p, mel, n_head=2)[:, :T]
    ref = mo.gru_head_forward(p, emb)
    np.testing.assert_allclose(logits.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-3)
    assert len(audios[1]) == n_samples - 700                         # caller's data not mutated


def test_long_form_chunking_matches_oracle():
    from oracle import model_oracle as mo
    model = _small_model(torch.float32, seed=11)
    n = 16000 * 65
    audio = _wave(n, 12)
    with torch.no_grad():
        logits, _ = model.frame_manual_forward([audio])
    plan = mo.chunk_plan(n // 160)
    T = sum(k for _, _, k in plan)
    assert tuple(logits.shape) == (1, T, 300) and T == 3250
    p = _oracle_params(model)
    mel = mo.log_mel_spectrogram(audio[None])
    parts = [mo.encoder_forward(p, mo.pad_or_trim(mel[:, :, s:e], 3000), n_head=2)[:, :k] for s, e, k in plan]
    ref = mo.gru_head_forward(p, torch.cat(parts, dim=1))
    np.testing.assert_allclose(logits.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-3)


@pytest.mark.parametrize("use_ctc", [True, False])
def test_fused_align_vs_oracle_and_two_step_path(use_ctc):
    """align() (fused, logits never materialised) vs (a) the oracle's emissions within 1e-3 and its DP bit-exactly on
    the SAME emissions, (b) the drop-in two-step path perform_viterbi(_ctc)(frame_manual_forward(...))."""
    from lyricalignment_amd import _lib, ops
    from lyricalignment_amd.utils import alignment as ua
    from oracle import alignment_oracle as ao
    from oracle import model_oracle as mo
    from lyricalignment_amd import whisper_compat as wc
    model = _small_model(torch.float32, seed=13)
    wc.init_align_head(model, seed=17, fc_scale=12.0)   # peaked posteriors (a trained head's): boundaries are decided, not tie-broken
    n = 60096
    audios = [_wave(n, 14), _wave(n, 15)]
    rs = np.random.RandomState(16)
    ncls = 298 if use_ctc else 299
    labels = torch.full((2, 11), -100, dtype=torch.long)
    labels[0] = torch.from_numpy(rs.randint(1, ncls + 1, size=11))
    labels[1, :6] = torch.from_numpy(rs.randint(1, ncls + 1, size=6))
    labels[0, 4] = labels[0, 3]
    with torch.no_grad():
        on, off, score, status = model.align(audios, labels, use_ctc=use_ctc, return_frames=True)
        secs = model.align(audios, labels, use_ctc=use_ctc)
        logits, _ = model.frame_manual_forward(audios)
    assert (status.cpu().numpy() == 0).all()
    T = logits.shape[1]
    # (a) oracle emissions from the oracle's own fp32 pipeline
    p = _oracle_params(model)
    mel = mo.pad_or_trim(mo.log_mel_spectrogram(np.stack(audios)), 3000)
    ref_logits = mo.gru_head_forward(p, mo.encoder_forward(p, mel, n_head=2)[:, :T])
    lp, ls = (mo.emission_prep_ctc if use_ctc else mo.emission_prep_plain)(ref_logits)
    eng = model.engine()
    lab_dev, n_lab, lists = ua._labels_to_device(labels, 2, eng.device)
    feats, B, T2, stride = model._features(model._mel_of(audios), True)
    em = eng.emissions(feats, B, T2, stride, lab_dev, n_lab, _lib.LA_VARIANT_CTC if use_ctc else _lib.LA_VARIANT_PLAIN).cpu().numpy()
    for b, labs in enumerate(lists):
        idx = np.array(labs) - 1
        np.testing.assert_allclose(em[b, :, 0], ls[b, :, 0].numpy(), rtol=0, atol=1e-3)
        np.testing.assert_allclose(em[b, :, 1:1 + len(labs)], lp[b][:, idx].numpy(), rtol=0, atol=1e-3)
        rc, on_o, off_o, sc_o = ao.align_frames_compact(em[b], np.array(labs))       # DP on the SAME emissions: bit-exact
        assert rc == 0
        assert on.cpu().numpy()[b, : len(labs)].tolist() == on_o.tolist()
        assert off.cpu().numpy()[b, : len(labs)].tolist() == off_o.tolist()
        assert score.cpu().numpy()[b] == sc_o
        assert secs[b] == [[float(int(a)) * 0.02, float(int(c)) * 0.02] for a, c in zip(on_o, off_o)]
    # (b) the two-step drop-in path (materialised logits, device or host) gives the same seconds up to emission rounding
    two_dev = (ua.perform_viterbi_ctc if use_ctc else ua.perform_viterbi)(logits, labels)
    two_host = (ua.perform_viterbi_ctc if use_ctc else ua.perform_viterbi)(logits.cpu(), labels)
    assert two_dev == two_host
    assert two_dev == secs       # fused FC + emission prep vs materialised logits: 1e-6 apart, the same decisions on a peaked head
    assert ua.get_mae(two_dev, two_dev) == 0.0


def test_drop_in_alignment_module_vs_golden():
    """utils.alignment drop-in on the reference's golden vectors: seconds and exceptions."""
    import json
    from conftest import load_json, load_npz
    from lyricalignment_amd.utils import alignment as ua
    z = load_npz("viterbi_e2e.npz")
    meta = json.loads(bytes(z["meta_json"]).decode())
    for m in meta:
        if m["scale"] < 1.0:
            continue   # near-flat logits: emission-prep rounding decides ties; covered on exact emissions elsewhere
        rs = np.random.RandomState(m["seed"])
        logits = torch.from_numpy((rs.randn(m["B"], m["T"], m["V"]) * m["scale"]).astype(np.float32))
        fn = ua.perform_viterbi_ctc if m["variant"] == "ctc" else ua.perform_viterbi
        res = fn(logits, torch.tensor(m["labels"]))
        for b in range(m["B"]):
            assert res[b] == z[f"{m['name']}/{b}/seconds"].tolist(), m["name"]
    for e in load_json("viterbi_errors.json"):
        lg = torch.from_numpy(np.random.RandomState(e["seed"]).randn(1, e["T"], e["V"]).astype(np.float32))
        if e["raises"] is None:
            assert ua.perform_viterbi_ctc(lg, torch.tensor(e["labels"])) == e["result"]
        else:
            with pytest.raises({"ValueError": ValueError, "IndexError": IndexError}[e["raises"]]):
                ua.perform_viterbi_ctc(lg, torch.tensor(e["labels"]))
    g = load_json("mae.json")
    assert ua.get_mae(g["gt"], g["predict"]) == g["mae"]


def test_run_viterbi_core_drop_in_bit_exact():
    import hashlib
    from conftest import core_inputs, load_json
    from lyricalignment_amd.utils import alignment as ua
    for case in load_json("viterbi_core.json")["cases"]:
        if case["T"] > 1600 or case["Vp"] > 1000:
            continue
        lp, ls, label = core_inputs(case["seed"], case["T"], case["L"], case["Vp"], case["scale"], case["repeat_at"])
        T, S = case["T"], 2 * case["L"] + 1
        dp = np.full((T, S), -10000000.0, dtype=np.float64)
        bt = np.zeros((T, S), dtype=np.int64)
        dp[0][0] = ls[0][0]; dp[0][1] = lp[0][label[0] - 1]
        ua.run_viterbi_core(dp, bt, lp, ls, label)
        sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
        assert sha(bt) == case["bt_sha256"] and sha(dp) == case["dp_sha256"], case["seed"]


def test_state_dict_layout_and_training_mode_forward():
    from conftest import load_json
    model = _small_model(torch.float32)
    keys = load_json("head_state_dict_keys.json")
    got = {k[len("align_rnn."):] for k in model.state_dict() if k.startswith("align_rnn.")}
    assert got == set(keys)
    assert any(k.startswith("whisper_model.encoder.blocks.0.attn.query.weight") for k in model.state_dict())
    assert "whisper_model.encoder.blocks.0.attn.key.bias" not in model.state_dict()
    model.train()                                    # training mode: autograd through the HIP forward / backward kernels
    logits, _ = model.frame_manual_forward([_wave(16000)])
    assert logits.requires_grad and logits.shape[1] == 50
    logits.backward(torch.ones_like(logits))         # the module lives on the host here: gradients come back to it
    g = model.align_rnn.fc.bias.grad
    assert g is not None and g.device.type == "cpu" and torch.isfinite(g).all()
    assert model.whisper_model.encoder.conv1.weight.grad is not None
    model.eval()
    with torch.no_grad():
        ev, _ = model.frame_manual_forward([_wave(16000)])
    assert not ev.requires_grad and ev.shape == logits.shape


@pytest.mark.parametrize("head_group,encoder_streams", [(1, 1), (2, 1), (3, 1), (2, 2)])
def test_pipelined_aligner_matches_single_stream(head_group, encoder_streams):
    """Two-stream encoder/head overlap across consecutive batches, with the head run once per `head_group` batches,
    changes no result (5 batches: full groups and a partial one flushed by drain())."""
    from lyricalignment_amd.engine import PipelinedAligner
    model = _small_model(torch.bfloat16, seed=21)
    eng = model.engine()
    pipe = PipelinedAligner(eng, head_group=head_group, encoder_streams=encoder_streams)
    rs = np.random.RandomState(22)
    batches = []
    for i in range(5):
        mel = torch.from_numpy(rs.uniform(-1, 1, size=(3, 80, 3000)).astype(np.float32)).cuda()
        labels = torch.from_numpy(rs.randint(1, 299, size=(3, 9)).astype(np.int32)).cuda()
        n_labels = torch.tensor([9, 5, 2], dtype=torch.int32).cuda()
        batches.append((mel, labels, n_labels))
    with torch.no_grad():
        ref = [tuple(t.clone() for t in eng.align_mel(*b, n_frames=700)) for b in batches]
        torch.cuda.synchronize()
        outs = [pipe.submit(*b, n_frames=700) for b in batches]
        pipe.drain()
    for r, o in zip(ref, outs):
        assert torch.equal(r[0], o[0]) and torch.equal(r[1], o[1]) and torch.equal(r[3], o[3])
        assert torch.equal(r[2], o[2])


@pytest.mark.parametrize("head_group", [1, 2])
def test_pipelined_long_form_songs_match_single_stream(head_group):
    """BASELINE configs[4]: whole songs (73 s -> two full 30 s chunks + a partial one, T = 3651 frames) through
    PipelinedAligner.submit_songs -- song-major chunk batch on the encoder stream, recurrence + FC + DP of the previous
    songs on the head stream -- give exactly AlignModel.align's frames; and a batch of <= 30 s songs takes the short branch."""
    from lyricalignment_amd.engine import PipelinedAligner
    model = _small_model(torch.bfloat16, seed=23)
    eng = model.engine()
    pipe = PipelinedAligner(eng, head_group=head_group)
    rs = np.random.RandomState(24)
    batches = []
    for i in range(3):
        n_mel = 7302 if i < 2 else 2001                      # 2001 mel frames -> short branch, round(2001 / 2) = 1000 frames
        mel = torch.from_numpy(rs.uniform(-1, 1, size=(2, 80, n_mel)).astype(np.float32)).cuda()
        labels = torch.from_numpy(rs.randint(1, 299, size=(2, 40)).astype(np.int64))
        labels[1, 25:] = -100
        batches.append((mel, labels))
    from lyricalignment_amd.utils.alignment import _labels_to_device
    with torch.no_grad():
        ref = [tuple(t.clone() for t in model.align(mel=m, labels=l, return_frames=True)) for m, l in batches]
        torch.cuda.synchronize()
        outs = []
        for m, l in batches:
            lab_dev, n_lab, _ = _labels_to_device(l, 2, eng.device)
            outs.append(pipe.submit_songs(m, lab_dev, n_lab))
        pipe.drain()
    assert int(ref[0][3].abs().sum()) == 0
    for r, o in zip(ref, outs):
        for a, b in zip(r, o):
            assert torch.equal(a, b)


def test_audio_loader_resample_matches_scipy(tmp_path):
    """utils.audio.load_audio_file: WAV decode + device polyphase resampling vs scipy.signal.resample_poly
    (the reference's librosa resampler is absent / un-pinned: parity with it is unpinned)."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    from lyricalignment_amd.utils.audio import load_audio_file
    rs = np.random.RandomState(0)
    for sr, n in ((44100, 44100 * 2 + 17), (48000, 30000), (22050, 9999), (16000, 5000)):
        t = np.arange(n) / sr
        stereo = np.stack([0.4 * np.sin(2 * np.pi * 440 * t) + 0.05 * rs.randn(n), 0.3 * np.sin(2 * np.pi * 1000 * t)], axis=1)
        path = str(tmp_path / f"a{sr}.wav")
        wavfile.write(path, sr, (stereo * 32767).astype(np.int16))
        dec = (stereo * 32767).astype(np.int16).astype(np.float32) / 32768.0
        from fractions import Fraction
        fr = Fraction(16000, sr)
        want_mono = resample_poly(dec.mean(axis=1).astype(np.float64), fr.numerator, fr.denominator) if sr != 16000 else dec.mean(axis=1)
        got = load_audio_file(path, 0)
        assert got["sampling_rate"] == 16000 and got["speech"].dtype == np.float32
        np.testing.assert_allclose(got["speech"], want_mono, rtol=0, atol=2e-6)
        want_ch1 = resample_poly(dec[:, 1].astype(np.float64), fr.numerator, fr.denominator) if sr != 16000 else dec[:, 1]
        np.testing.assert_allclose(load_audio_file(path, 2)["speech"], want_ch1, rtol=0, atol=2e-6)
        want_mix = (resample_poly(dec[:, 0].astype(np.float64), fr.numerator, fr.denominator) + want_ch1) / 2 if sr != 16000 else dec.mean(axis=1)
        np.testing.assert_allclose(load_audio_file(path, 1)["speech"], want_mix, rtol=0, atol=2e-6)
    with pytest.raises(ValueError):
        load_audio_file(path, 3)


def test_large_v2_dims_one_layer():
    """BASELINE configs[3] architecture (d=1280, 20 heads): one block of it through the bf16 and f32 paths vs the oracle."""
    from lyricalignment_amd import whisper_compat as wc
    from oracle import model_oracle as mo
    dims = wc.ModelDimensions(n_audio_state=1280, n_audio_head=20, n_audio_layer=1, n_text_state=1280, n_text_head=20, n_text_layer=0)
    wm = wc.build_model(dims=dims, seed=31, std=0.02)
    mel = torch.rand(1, 80, 3000, generator=torch.Generator().manual_seed(32)) * 2 - 1
    p = {"encoder." + k: v.detach().float() for k, v in wm.encoder.state_dict().items()}
    ref = mo.encoder_forward(p, mel, n_head=20)
    with torch.no_grad():
        out = wm.embed_audio(mel.cuda()).cpu()        # bare Whisper object: float32 engine
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=0, atol=1e-3)


def test_long_form_three_minute_song_config5():
    """BASELINE configs[4] shape: a 3-minute song -> 6 encoder chunks, T = 9000 frames, 150 labels (multi-wave DP with
    backpointers in the HBM workspace, GRU recurrence over 9000 steps).  Device DP vs the oracle on the device's emissions."""
    from lyricalignment_amd import _lib
    from lyricalignment_amd.utils import alignment as ua
    from oracle import alignment_oracle as ao
    model = _small_model(torch.bfloat16, seed=41)
    n = 16000 * 180
    audio = _wave(n, 42)
    rs = np.random.RandomState(43)
    labels = torch.from_numpy(rs.randint(1, 299, size=(1, 150)))
    with torch.no_grad():
        on, off, score, status = model.align([audio], labels, use_ctc=True, return_frames=True)
        eng = model.engine()
        lab_dev, n_lab, lists = ua._labels_to_device(labels, 1, eng.device)
        feats, B, T, stride = model._features(model._mel_of([audio]), True)
        em = eng.emissions(feats, B, T, stride, lab_dev, n_lab, _lib.LA_VARIANT_CTC).cpu().numpy()
    assert T == 9000 and int(status[0]) == 0
    rc, on_o, off_o, sc_o = ao.align_frames_compact(em[0], np.array(lists[0]))
    assert rc == 0 and on.cpu().numpy()[0].tolist() == on_o.tolist() and off.cpu().numpy()[0].tolist() == off_o.tolist()
    assert float(score[0]) == sc_o


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 8e-2)])
def test_decoder_logits_match_oracle(dtype, tol):
    """Whisper.logits (text decoder: causal self-attention, cross-attention to the audio features, tied projection)
    through AlignModel.frame_manual_forward(train_transcript=True) vs the oracle's TextDecoder restatement."""
    from lyricalignment_amd import whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    from oracle import model_oracle as mo
    dims = wc.ModelDimensions(n_audio_state=128, n_audio_head=2, n_audio_layer=2, n_text_state=128, n_text_head=2, n_text_layer=2,
                              n_vocab=311, n_text_ctx=64)
    wm = wc.build_model(dims=dims, seed=51, std=0.05, with_decoder=True)
    model = AlignModel(wm, embed_dim=128, hidden_dim=64, output_dim=300, train_transcript=True, device="cuda", compute_dtype=dtype).eval()
    audios = [_wave(60096, 52), _wave(48160, 53)]
    tokens = torch.randint(0, 311, (2, 37), generator=torch.Generator().manual_seed(54))
    with torch.no_grad():
        align_logit, tr = model.frame_manual_forward(audios, y_in=tokens)
    assert tuple(tr.shape) == (2, 37, 311) and tr.dtype == torch.float32 and align_logit.shape[1] == 188
    p = {"encoder." + k: v.detach().float().cpu() for k, v in wm.encoder.state_dict().items()}
    p.update({"decoder." + k: v.detach().float().cpu() for k, v in wm.decoder.state_dict().items()})
    batch = np.zeros((2, 60096), dtype=np.float32); batch[0] = audios[0]; batch[1, :48160] = audios[1]
    mel = mo.pad_or_trim(mo.log_mel_spectrogram(batch), 3000)
    xa = mo.encoder_forward(p, mel, n_head=2)
    ref = mo.decoder_forward(p, tokens, xa, n_head=2)
    np.testing.assert_allclose(tr.cpu().numpy(), ref.numpy(), rtol=0, atol=tol)
    if dtype == torch.float32:   # bare Whisper object API: embed_audio + logits
        with torch.no_grad():
            feats = wm.embed_audio(mel.cuda())
            tr2 = wm.logits(tokens=tokens.cuda(), audio_features=feats)
        np.testing.assert_allclose(tr2.cpu().numpy(), ref.numpy(), rtol=0, atol=tol)


def test_attention_ex_causal_and_cross():
    from lyricalignment_amd import ops
    g = torch.Generator().manual_seed(60)
    B, n, m, H = 2, 70, 150, 2
    d = H * 64
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        qkv = (torch.randn(B * n, 3 * d, generator=g)).to(dtype)
        qkv[:, :d] *= 0.3
        x = qkv.cuda()
        out = ops.attention_ex(x[:, :d], x[:, d:2 * d], x[:, 2 * d:], B, n, n, H, causal=True).float().cpu()
        xd = qkv.double().reshape(B, n, 3, H, 64)
        q, k, v = xd[:, :, 0].transpose(1, 2), xd[:, :, 1].transpose(1, 2), xd[:, :, 2].transpose(1, 2)
        mask = torch.full((n, n), float("-inf"), dtype=torch.float64).triu_(1)
        ref = (torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1) @ v).transpose(1, 2).reshape(B * n, d)
        np.testing.assert_allclose(out.double().numpy(), ref.numpy(), rtol=0, atol=tol)
        kv = torch.randn(B * m, 2 * d, generator=g).to(dtype)
        y = kv.cuda()
        out = ops.attention_ex(x[:, :d], y[:, :d], y[:, d:], B, n, m, H, causal=False).float().cpu()
        kd = kv.double().reshape(B, m, 2, H, 64)
        k2, v2 = kd[:, :, 0].transpose(1, 2), kd[:, :, 1].transpose(1, 2)
        ref = (torch.softmax(q @ k2.transpose(-1, -2), dim=-1) @ v2).transpose(1, 2).reshape(B * n, d)
        np.testing.assert_allclose(out.double().numpy(), ref.numpy(), rtol=0, atol=tol)


def test_attention_ex_causal_and_cross_on_long_sequences():
    """From 1024 query positions on the 16-bit kernels run 256-query / 8-wave workgroups (la_attention.hip); the text decoder never
    gets there, but la_attention_ex is a public entry point: causal self-attention over 1100 positions (block diagonal clipping with
    256-query blocks, a partial last block) and cross-attention of 1100 queries over 1300 keys, bf16 and f16, against float64."""
    from lyricalignment_amd import ops
    g = torch.Generator().manual_seed(61)
    B, n, m, H = 1, 1100, 1300, 2
    d = H * 64
    for dtype, tol in ((torch.bfloat16, 2e-2), (torch.float16, 4e-3)):
        qkv = (torch.randn(B * n, 3 * d, generator=g)).to(dtype)
        qkv[:, :d] *= 0.3
        x = qkv.cuda()
        out = ops.attention_ex(x[:, :d], x[:, d:2 * d], x[:, 2 * d:], B, n, n, H, causal=True).float().cpu()
        xd = qkv.double().reshape(B, n, 3, H, 64)
        q, k, v = xd[:, :, 0].transpose(1, 2), xd[:, :, 1].transpose(1, 2), xd[:, :, 2].transpose(1, 2)
        mask = torch.full((n, n), float("-inf"), dtype=torch.float64).triu_(1)
        ref = (torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1) @ v).transpose(1, 2).reshape(B * n, d)
        np.testing.assert_allclose(out.double().numpy(), ref.numpy(), rtol=0, atol=tol)
        kv = torch.randn(B * m, 2 * d, generator=g).to(dtype)
        y = kv.cuda()
        out = ops.attention_ex(x[:, :d], y[:, :d], y[:, d:], B, n, m, H, causal=False).float().cpu()
        kd = kv.double().reshape(B, m, 2, H, 64)
        k2, v2 = kd[:, :, 0].transpose(1, 2), kd[:, :, 1].transpose(1, 2)
        ref = (torch.softmax(q @ k2.transpose(-1, -2), dim=-1) @ v2).transpose(1, 2).reshape(B * n, d)
        np.testing.assert_allclose(out.double().numpy(), ref.numpy(), rtol=0, atol=tol)


@pytest.mark.parametrize("B,n0,steps", [(2, 3, 12), (1, 1, 20)])
def test_greedy_decode_with_kv_cache_matches_full_recompute(B, n0, steps):
    """Whisper.decode_greedy (cross K/V projected once, self K/V cache, one-token attention, device argmax) against the
    oracle's TextDecoder run from scratch on the growing sequence at every step: same tokens, and the last-step logits of
    the cached path equal Whisper.logits on the final sequence."""
    from lyricalignment_amd import whisper_compat as wc
    from oracle import model_oracle as mo
    dims = wc.ModelDimensions(n_audio_state=128, n_audio_head=2, n_audio_layer=1, n_text_state=128, n_text_head=2, n_text_layer=2,
                              n_vocab=97, n_text_ctx=40)
    wm = wc.build_model(dims=dims, seed=71 + B, std=0.3, with_decoder=True)     # std 0.3: well separated logits
    p = {"decoder." + k: v.detach().float().cpu() for k, v in wm.decoder.state_dict().items()}
    g = torch.Generator().manual_seed(72)
    xa = torch.randn(B, 1500, 128, generator=g)
    prompt = torch.randint(1, 97, (B, n0), generator=g)
    eot = 0
    ref = prompt.clone()
    done = torch.zeros(B, dtype=torch.bool)
    for _ in range(steps):
        nxt = mo.decoder_forward(p, ref, xa, n_head=2)[:, -1].argmax(dim=-1)
        nxt = torch.where(done, torch.full_like(nxt, eot), nxt)
        done |= nxt == eot
        ref = torch.cat([ref, nxt[:, None]], dim=1)
        if bool(done.all()):
            break
    got = wm.decode_greedy(prompt.cuda(), xa.cuda(), max_new_tokens=steps, eot=eot).cpu()
    assert got.shape == ref.shape and torch.equal(got, ref), (got, ref)
    full = wm.logits(tokens=got[:, :-1].cuda(), audio_features=xa.cuda())[:, -1].cpu()
    assert torch.equal(full.argmax(dim=-1), torch.where(done & (got[:, -1] == eot), full.argmax(dim=-1), got[:, -1]))


def test_attention_cached_and_argmax_ops():
    from lyricalignment_amd import ops
    g = torch.Generator().manual_seed(73)
    B, H, n_max, n = 3, 2, 24, 17
    d = 64 * H
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2)):
        q = (torch.randn(B, d, generator=g) * 0.3).to(dtype)
        kv = torch.randn(B, n_max, 2 * d, generator=g).to(dtype)
        out = ops.attention_cached(q.cuda(), kv.cuda().view(B * n_max, 2 * d)[:, :d], kv.cuda().view(B * n_max, 2 * d)[:, d:],
                                   B, 1, n, H, q_batch_rows=1, kv_batch_rows=n_max).float().cpu()
        qd = q.double().view(B, H, 1, 64)
        kd = kv[:, :n, :d].double().view(B, n, H, 64).transpose(1, 2)
        vd = kv[:, :n, d:].double().view(B, n, H, 64).transpose(1, 2)
        ref = (torch.softmax(qd @ kd.transpose(-1, -2), dim=-1) @ vd).transpose(1, 2).reshape(B, d)
        np.testing.assert_allclose(out.double().numpy(), ref.numpy(), rtol=0, atol=tol)
    x = torch.randn(5, 3001, generator=g)
    x[2, 17] = x[2, 2900] = 9.0                                   # tie: the first maximum wins, like torch.argmax
    assert torch.equal(ops.argmax_rows(x.cuda()).cpu(), torch.tensor([int(r.argmax()) if i != 2 else 17 for i, r in enumerate(x)]))


def _reference_beam_search(p, prompt_row, xa_row, beam, max_new, eot, n_head):
    """Plain restatement of the beam search of whisper/decoding.py (BeamSearchDecoder.update / finalize + the
    maximum-likelihood ranker without length penalty) on the oracle's decoder, re-run from scratch on every hypothesis."""
    from oracle import model_oracle as mo
    n0 = len(prompt_row)
    seqs = [list(prompt_row)] * beam
    sums = [0.0] * beam
    finished = {}
    for _ in range(max_new):
        toks = torch.tensor(seqs)
        logits = mo.decoder_forward(p, toks, xa_row[None].expand(beam, -1, -1), n_head=n_head)[:, -1]
        logp = torch.log_softmax(logits.double(), dim=-1)
        scores, sources = {}, {}
        for j in range(beam):
            v, ix = logp[j].topk(beam + 1)
            for lp, tk in zip(v.tolist(), ix.tolist()):
                key = tuple(seqs[j] + [tk])
                scores[key] = sums[j] + lp
                sources[key] = j
        new_seqs, new_sums = [], []
        for key in sorted(scores, key=scores.get, reverse=True):
            if key[-1] == eot:
                finished[key] = scores[key]
            else:
                new_seqs.append(list(key)); new_sums.append(scores[key])
                if len(new_seqs) == beam:
                    break
        seqs, sums = new_seqs, new_sums
        if len(finished) > beam:
            keep = sorted(finished, key=finished.get, reverse=True)[:beam]
            finished = {k: finished[k] for k in keep}
        if len(finished) >= beam:
            break
    cands = dict(finished)
    for j in sorted(range(beam), key=lambda j: sums[j], reverse=True):
        if len(cands) >= beam:
            break
        cands[tuple(seqs[j] + [eot])] = sums[j]
    best = max(cands, key=lambda k: cands[k] / max(1, len(k) - n0 - 1))
    return list(best[:-1]), cands[best]


@pytest.mark.parametrize("beam,eot", [(3, 0), (5, 7)])
def test_beam_search_matches_plain_restatement(beam, eot):
    """AlignEngine.decode_beam (device top-k + log-sum-exp, K/V cache gathered along the surviving hypotheses,
    cross-attention shared by a clip's beams) against the plain restatement above: same winning tokens, same score."""
    from lyricalignment_amd import whisper_compat as wc
    dims = wc.ModelDimensions(n_audio_state=128, n_audio_head=2, n_audio_layer=1, n_text_state=128, n_text_head=2, n_text_layer=2,
                              n_vocab=61, n_text_ctx=32)
    wm = wc.build_model(dims=dims, seed=80 + beam, std=0.25, with_decoder=True)
    p = {"decoder." + k: v.detach().float().cpu() for k, v in wm.decoder.state_dict().items()}
    g = torch.Generator().manual_seed(81)
    B, n0, max_new = 2, 2, 9
    xa = torch.randn(B, 1500, 128, generator=g)
    prompt = torch.randint(1, 61, (B, n0), generator=g)
    toks, lps = wm.decode_beam(prompt.cuda(), xa.cuda(), beam_size=beam, max_new_tokens=max_new, eot=eot)
    for i in range(B):
        ref_t, ref_lp = _reference_beam_search(p, prompt[i].tolist(), xa[i], beam, max_new, eot, 2)
        assert toks[i].tolist() == ref_t, (i, toks[i].tolist(), ref_t)
        assert abs(lps[i] - ref_lp) < 2e-3 * max(1.0, abs(ref_lp))


def test_topk_rows_matches_torch():
    from lyricalignment_amd import ops
    g = torch.Generator().manual_seed(82)
    x = torch.randn(7, 51865, generator=g) * 3
    v, i, lse = ops.topk_rows(x.cuda(), 6)
    rv, ri = x.topk(6, dim=1)
    assert torch.equal(i.cpu(), ri) and torch.equal(v.cpu(), rv)
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(x.double(), dim=1).numpy(), rtol=2e-6)


def test_full_size_medium_batch_properties():
    """BASELINE configs[1] at full size (Whisper-medium, 32 x 30 s, bf16, T = 1500): size-independent properties of the
    whole path.  (1) every alignment is well formed: status OK, onset <= offset - 1, segments ordered and inside [0, T];
    (2) sharding invariance: aligning the two halves of the batch separately gives bit-identical frames and scores (clips
    are independent -- the multi-GPU path shards them with no collective); (3) the two-stream pipeline changes nothing;
    (4) idempotence: a second pass over the same batch reproduces the first bit for bit."""
    from lyricalignment_amd import whisper_compat as wc
    from lyricalignment_amd.engine import PipelinedAligner
    from lyricalignment_amd.module.align_model import AlignModel
    wm = wc.build_model("medium", seed=3)
    model = AlignModel(wm, embed_dim=1024, hidden_dim=384, output_dim=21129, device="cuda", compute_dtype=torch.bfloat16).eval()
    eng = model.engine()
    rs = np.random.RandomState(5)
    B, T, Lmax = 32, 1500, 26
    mel = torch.from_numpy(rs.uniform(-1, 1, size=(B, 80, 3000)).astype(np.float32)).cuda()
    Ls = rs.randint(5, Lmax + 1, size=B)
    labels = torch.zeros((B, Lmax), dtype=torch.int32)
    for b in range(B):
        labels[b, : Ls[b]] = torch.from_numpy(rs.randint(2, 403, size=Ls[b]).astype(np.int32))
    labels, n_labels = labels.cuda(), torch.from_numpy(Ls.astype(np.int32)).cuda()
    with torch.no_grad():
        on, off, score, status = [t.clone() for t in eng.align_mel(mel, labels, n_labels, n_frames=T, use_ctc=True)]
        on2, off2, score2, _ = [t.clone() for t in eng.align_mel(mel, labels, n_labels, n_frames=T, use_ctc=True)]
        halves = [[t.clone() for t in eng.align_mel(mel[s], labels[s], n_labels[s], n_frames=T, use_ctc=True)] for s in (slice(0, 16), slice(16, 32))]
        pipe = PipelinedAligner(eng, head_group=2)
        outs = [pipe.submit(mel, labels, n_labels, n_frames=T) for _ in range(3)]
        pipe.drain()
    assert int((status != 0).sum()) == 0
    onc, offc = on.cpu().numpy(), off.cpu().numpy()
    for b in range(B):
        L = int(Ls[b])
        o, f = onc[b, :L], offc[b, :L]
        assert (o >= 0).all() and (f <= T).all() and (f > o).all() and (o[1:] >= f[:-1]).all(), b
    assert torch.equal(on, on2) and torch.equal(off, off2) and torch.equal(score, score2)
    assert torch.equal(torch.cat([halves[0][0], halves[1][0]]), on) and torch.equal(torch.cat([halves[0][1], halves[1][1]]), off)
    assert torch.equal(torch.cat([halves[0][2], halves[1][2]]), score)
    for o in outs:
        assert torch.equal(o[0], on) and torch.equal(o[1], off) and torch.equal(o[2], score)


def test_large_v2_fp16_alignment_config4():
    """BASELINE configs[3]: Whisper-large-v2 shape (d = 1280, 20 heads; two blocks here), float16 MFMA path, fused
    align: emissions within the float16 tolerance of the fp32 oracle and frames bit-exact given the device emissions."""
    from lyricalignment_amd import _lib, whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    from lyricalignment_amd.utils.alignment import _labels_to_device
    from oracle import alignment_oracle as ao, model_oracle as mo
    dims = wc.ModelDimensions(n_audio_state=1280, n_audio_head=20, n_audio_layer=2, n_text_state=1280, n_text_head=20, n_text_layer=0)
    wm = wc.build_model(dims=dims, seed=131, std=0.02)
    model = AlignModel(wm, embed_dim=1280, hidden_dim=384, output_dim=420, device="cuda", compute_dtype=torch.float16).eval()
    audio = _wave(60096, 132)
    lists = [[3, 17, 17, 250, 9, 401, 44, 2, 90, 91, 300]]
    with torch.no_grad():
        on, off, score, status = model.align([audio], lists, use_ctc=True, return_frames=True)
        eng = model.engine()
        lab_dev, n_lab, _ = _labels_to_device(lists, 1, eng.device)
        feats, B, T, stride = model._features(model._mel_of([audio]), True)
        em = eng.emissions(feats, B, T, stride, lab_dev, n_lab, _lib.LA_VARIANT_CTC).cpu().numpy()
    assert int(status[0]) == 0 and T == 188
    rc, on_o, off_o, sc_o = ao.align_frames_compact(em[0], np.array(lists[0]))
    assert rc == 0 and on.cpu().numpy()[0, :11].tolist() == on_o.tolist() and off.cpu().numpy()[0, :11].tolist() == off_o.tolist()
    # emissions vs the fp32 oracle of the same weights
    p = {"encoder." + k: v.detach().float().cpu() for k, v in wm.encoder.state_dict().items()}
    p.update({"align_rnn." + k: v.detach().float().cpu() for k, v in model.align_rnn.state_dict().items()})
    mel = mo.pad_or_trim(mo.log_mel_spectrogram(audio[None]), 3000)
    logits = mo.gru_head_forward(p, mo.encoder_forward(p, mel, n_head=20)[:, :188])
    lp, ls = mo.emission_prep_ctc(logits)
    np.testing.assert_allclose(em[0, :, 0], ls[0, :, 0].numpy(), rtol=0, atol=2e-2)
    idx = torch.tensor(lists[0]) - 1
    np.testing.assert_allclose(em[0, :, 1:12], lp[0][:, idx].numpy(), rtol=0, atol=2e-2)


def test_head_chunking_for_huge_batches(monkeypatch):
    """align_mel slices the head over clips when a batch exceeds HEAD_CLIPS_MAX (BASELINE configs[3] batches): same result."""
    from lyricalignment_amd import engine as eng_mod
    model = _small_model(torch.bfloat16, seed=141)
    eng = model.engine()
    rs = np.random.RandomState(142)
    B = 5
    mel = torch.from_numpy(rs.uniform(-1, 1, size=(B, 80, 3000)).astype(np.float32)).cuda()
    labels = torch.from_numpy(rs.randint(1, 299, size=(B, 9)).astype(np.int32)).cuda()
    n_labels = torch.tensor([9, 5, 2, 7, 1], dtype=torch.int32).cuda()
    with torch.no_grad():
        ref = [t.clone() for t in eng.align_mel(mel, labels, n_labels, n_frames=700)]
        monkeypatch.setattr(eng_mod, "HEAD_CLIPS_MAX", 2)
        got = eng.align_mel(mel, labels, n_labels, n_frames=700)
    for r, g in zip(ref, got):
        assert torch.equal(r, g)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_layernorm_folded_into_gemms_matches_separate_pass(monkeypatch, dtype):
    """Encoder blocks at a size where every GEMM runs on the 256x256 kernel (d = 1024, 9 clips): the LayerNorm-folded path
    (GEMMs that write the f32 residual stream also emit its bf16 copy, row statistics, rstd (acc - mean c) + b' epilogue on
    gamma-folded weights) against the separate LayerNorm pass and against the float32 engine.  Same function, different
    rounding points: the folded path must be as close to float32 as the separate pass is."""
    from lyricalignment_amd import engine as eng_mod, whisper_compat as wc
    dims = wc.ModelDimensions(n_audio_state=1024, n_audio_head=16, n_audio_layer=2, n_text_state=1024, n_text_head=16, n_text_layer=0)
    wm = wc.build_model(dims=dims, seed=77, std=0.02)
    with torch.no_grad():                                   # a mean offset and a few large channels in the residual stream
        wm.encoder.conv2.bias.add_(0.3)
        wm.encoder.conv2.bias[::97].mul_(8.0)
    sd = {"encoder." + k: v for k, v in wm.encoder.state_dict().items()}
    dev = torch.device("cuda")
    e16 = eng_mod.AlignEngine(eng_mod.pack_encoder(sd, 16, dtype, dev), None, dev)
    e32 = eng_mod.AlignEngine(eng_mod.pack_encoder(sd, 16, torch.float32, dev), None, dev)
    mel = torch.from_numpy(np.random.RandomState(78).uniform(-1, 1, size=(9, 80, 3000)).astype(np.float32)).cuda()
    with torch.no_grad():
        ref = e32.encode(mel).float().clone()
        monkeypatch.setattr(eng_mod, "LN_FUSION", True)
        monkeypatch.setattr(eng_mod, "LN_STATS_IN_EPILOGUE", True)
        fused = e16.encode(mel, out_dtype=torch.float32).clone()
        monkeypatch.setattr(eng_mod, "LN_STATS_IN_EPILOGUE", False)          # statistics by a separate read of the bf16 copy
        fused_pass = e16.encode(mel, out_dtype=torch.float32).clone()
        assert float((fused - fused_pass).abs().max()) < 2e-2
        monkeypatch.setattr(eng_mod, "LN_STATS_IN_EPILOGUE", True)
        monkeypatch.setattr(eng_mod, "LN_FUSION", False)
        plain = e16.encode(mel, out_dtype=torch.float32).clone()
    assert not torch.equal(fused, plain)                    # the folded path did run
    err_f = (fused - ref).abs()
    err_p = (plain - ref).abs()
    assert float(err_p.max()) < 0.1 and float(err_f.max()) < 0.1
    assert float(err_f.mean()) < 1.3 * float(err_p.mean()) + 1e-4, (float(err_f.mean()), float(err_p.mean()))
    assert float(err_f.max()) < 1.5 * float(err_p.max()) + 1e-3, (float(err_f.max()), float(err_p.max()))
    # fewer than 9 clips: the GEMMs leave the 256x256 kernel, encode() takes the separate pass by itself
    monkeypatch.setattr(eng_mod, "LN_FUSION", True)
    with torch.no_grad():
        small = e16.encode(mel[:2], out_dtype=torch.float32)
    assert torch.equal(small, plain[: 2 * 1500])
