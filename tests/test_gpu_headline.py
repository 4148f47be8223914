"""The bench's OWN configuration against the oracle (BASELINE configs[1]: Whisper-medium, bfloat16, 32 x 30 s clips, the
LayerNorm-folded 256x256 GEMMs, the head over 64 clips, two streams) -- round-2 verdict item 1.

BENCH_r02's self-check read 1.087 s on the driver's box and 0.0 on the builder's.  Diagnosis (profiles/r3_selfcheck_diagnosis.md):
the bench's head came from nn.Module's default initialisation out of the UNSEEDED global generator, whose stream differs from
host to host (tools/weights_fingerprint.py), and a default-initialised head on featureless noise gives near-flat posteriors:
the best lattice path then leads the runner-up by ~1e-4 of a total of ~1100, so which boundaries the bf16 path and the fp32
oracle agree on was a draw per host.  The device side is deterministic: every pipeline shape / warm-up count / repeat gives
bit-identical frames (first test below).  Since round 3 bench.py's weights and inputs are host-independent bits and the task is
well-posed: the clips are synthetic songs (notes of 40 timbres), the transcript is what was sung, and the head's 40 syllable rows are
a linear probe fitted on other songs (bench.fit_head) -- the lattice's best path leads by nats per frame, as a trained model's does on
its own data.  These tests hold that configuration to the oracle (utils/alignment.py:121-188, inference_alignment.py:159-177) and to
the note edges the songs were synthesised with."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_ORACLE_CLIPS = 4


@pytest.fixture(scope="module")
def headline():
    import bench
    from lyricalignment_amd import whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    from oracle import alignment_oracle as ao, model_oracle as mo
    ao.build()
    dims = wc.dims_for(bench.MODEL)
    model = AlignModel(wc.build_model(bench.MODEL, seed=0), embed_dim=dims.n_audio_state, hidden_dim=bench.HIDDEN, output_dim=bench.VOCAB,
                       device="cuda:0", compute_dtype=torch.bfloat16).eval()
    device = torch.device("cuda", 0)
    fit = bench.fit_head(model, device)
    with torch.no_grad():
        eng = model.engine()
    mel, labels, n_labels, Ls, plans = bench.build_inputs(device)
    with torch.no_grad():
        ref = eng.align_mel(mel, labels, n_labels, n_frames=bench.T_FRAMES, use_ctc=True)
    torch.cuda.synchronize()
    assert int((ref[3] != 0).sum()) == 0
    # the fp32 oracle's own end-to-end result on the first clips (one ~2.5 s CPU pass each)
    torch.set_num_threads(bench.usable_cores())
    p = {"encoder." + k: v.detach().float().cpu() for k, v in model.whisper_model.encoder.state_dict().items()}
    p.update({"align_rnn." + k: v.detach().float().cpu() for k, v in model.align_rnn.state_dict().items()})
    oracle = []
    for b in range(N_ORACLE_CLIPS):
        L = int(Ls[b])
        with torch.no_grad():
            logits = mo.gru_head_forward(p, mo.encoder_forward(p, mel[b:b + 1].cpu(), n_head=dims.n_audio_head))
            lp, ls = mo.emission_prep_ctc(logits)
            secs = ao.perform_viterbi_ctc(logits, labels[b:b + 1, :L].cpu().long())[0]
        oracle.append(dict(L=L, lp=lp[0], ls=ls[0], on=np.array([s[0] for s in secs]), off=np.array([s[1] for s in secs])))
    return dict(bench=bench, eng=eng, mel=mel, labels=labels, n_labels=n_labels, Ls=Ls, oracle=oracle, plans=plans, fit=fit, model=model, dims=dims,
                ref=[t.cpu().numpy().copy() for t in ref[:2]])


def _run_pipeline(h, submits, head_group=2, encoder_streams=1):
    """submits: list of batch counts, one drain() after each -> list of (onset, offset) numpy arrays, one per submitted batch."""
    from lyricalignment_amd.engine import PipelinedAligner
    bench = h["bench"]
    pipe = PipelinedAligner(h["eng"], head_group=head_group, encoder_streams=encoder_streams)
    got = []
    with torch.no_grad():
        for n in submits:
            outs = [pipe.submit(h["mel"], h["labels"], h["n_labels"], n_frames=bench.T_FRAMES, use_ctc=True) for _ in range(n)]
            pipe.drain()
            torch.cuda.synchronize()
            got += [(o[0].cpu().numpy(), o[1].cpu().numpy(), o[3].cpu().numpy()) for o in outs]
    return got


def _bench_shapes():
    """(head_group, encoder_streams, submits) of the pipeline: bench.py's OWN defaults with the driver's --warmup / --steps (read from
    bench.py at collection time, so a change of a default there changes what is held to the oracle here), and the earlier shapes."""
    import bench
    return [pytest.param(bench.DEFAULT_HEAD_GROUP, bench.DEFAULT_ENCODER_STREAMS, [bench.DRIVER_WARMUP, bench.DRIVER_STEPS], id="bench-defaults-driver-run"),
            pytest.param(bench.DEFAULT_HEAD_GROUP, bench.DEFAULT_ENCODER_STREAMS, [bench.DEFAULT_WARMUP, 5], id="bench-defaults-partial-flush"),
            pytest.param(bench.DEFAULT_HEAD_GROUP, 1, [bench.DEFAULT_HEAD_GROUP, 6], id="roofline-pass-shape"),
            pytest.param(2, 1, [5, 4], id="head-group-2")]


@pytest.mark.parametrize("head_group,encoder_streams,submits", _bench_shapes())
def test_pipeline_with_partial_flush_is_bit_identical_to_single_stream(headline, head_group, encoder_streams, submits):
    """The pipeline shapes bench.py runs -- its defaults (head over DEFAULT_HEAD_GROUP batches = 128 clips per launch set,
    DEFAULT_ENCODER_STREAMS encoder streams) under the driver's `--warmup 5 --steps 20` (a partial flush of 1 batch at the first drain,
    five full groups after it), the one-encoder-stream pass its roofline leg is measured on, and the round-2 shape (head groups of 2:
    64, 64, 32, 64, 64 clips).  Every batch's frames == the single-stream align_mel result: the pipeline changes which kernels are in
    flight together, never a result; and the LAST batch -- what bench.py's self-check reads -- against the fp32 oracle's own
    end-to-end boundaries on the first clips, to bench.py's tolerance (inference_alignment.py:159-177)."""
    bench = headline["bench"]
    mask = np.arange(headline["labels"].shape[1])[None, :] < headline["Ls"][:, None]
    got = _run_pipeline(headline, submits, head_group=head_group, encoder_streams=encoder_streams)
    assert len(got) == sum(submits)
    for i, (on, off, st) in enumerate(got):
        assert (st == 0).all()
        assert (on[mask] == headline["ref"][0][mask]).all() and (off[mask] == headline["ref"][1][mask]).all(), f"batch {i} differs"
    on, off, _ = got[-1]
    on_err = np.concatenate([np.abs(on[b, :o["L"]] * 0.02 - o["on"]) for b, o in enumerate(headline["oracle"])])
    off_err = np.concatenate([np.abs(off[b, :o["L"]] * 0.02 - o["off"]) for b, o in enumerate(headline["oracle"])])
    assert on_err.mean() <= bench.SELFCHECK_TOL_S and off_err.mean() <= bench.SELFCHECK_TOL_S
    assert max(on_err.max(), off_err.max()) <= 0.2


def test_head_over_the_bench_group_of_128_clips_emissions_against_oracle(headline):
    """The head launch set of the bench's default pipeline -- DEFAULT_HEAD_GROUP batches = 128 clips through ONE GRU / FC launch set
    (8 workgroup groups of the persistent recurrence) -- emission log-probs of the LAST batch's first clips against the fp32 oracle
    (the 32-clip launch set is held to it below), and every batch's emissions bit-identical to the first batch's (same clips)."""
    from lyricalignment_amd import _lib
    bench, eng = headline["bench"], headline["eng"]
    G, B, T = bench.DEFAULT_HEAD_GROUP, bench.BATCH, bench.T_FRAMES
    with torch.no_grad():
        feats = torch.empty((G * B * T, eng.enc.d), dtype=eng.enc.dtype, device=eng.device)
        for j in range(G):
            eng.encode(headline["mel"], out=feats[j * B * T:(j + 1) * B * T])
        labels, n_labels = torch.cat([headline["labels"]] * G, dim=0), torch.cat([headline["n_labels"]] * G, dim=0)
        em = eng.emissions(feats, G * B, T, T, labels, n_labels, _lib.LA_VARIANT_CTC).view(G, B, T, -1)
        valid = (torch.arange(em.shape[-1], device=em.device)[None, :] <= headline["n_labels"][:, None].long())[:, None, :]   # [B, 1, Lmax+1]
        for j in range(1, G):       # (columns past a clip's own labels are not written: compared where they are)
            assert torch.equal(em[j][valid.expand_as(em[j])], em[0][valid.expand_as(em[0])]), f"batch {j} of the 128-clip head launch set differs from batch 0"
        em = em[G - 1, :N_ORACLE_CLIPS].cpu()
    errs = []
    for b, o in enumerate(headline["oracle"]):
        L = o["L"]
        idx = headline["labels"][b, :L].cpu().long() - 1
        errs.append(torch.cat([(em[b, :, 1:1 + L] - o["lp"][:, idx]).abs().flatten(), (em[b, :, 0] - o["ls"][:, 0]).abs()]))
    errs = torch.cat(errs)
    assert float(errs.mean()) < 0.06 and float(errs.max()) < 0.6
    eng.check_gru()


def test_headline_batch_boundaries_and_emissions_against_oracle(headline):
    """The first clips of the B = 32 bf16 batch against alignment_oracle.perform_viterbi_ctc(oracle logits): emission log-probs
    (range 0 .. -35 with the fitted head) within bf16 bounds (measured: mean 0.035, max 0.30), boundary MAE within bench.py's
    self-check tolerance (one frame; measured 0.0005 s, no boundary further than one frame), >= 90 % of the boundaries equal
    (measured 97.6 %; 94.5 ... 100 % per clip over 16 clips, tools/selfcheck_repro.py), and the device's onsets within 0.1 s MAE of
    the note edges the songs were synthesised with (measured 0.020 s = one frame)."""
    from lyricalignment_amd import _lib
    bench, eng = headline["bench"], headline["eng"]
    n = N_ORACLE_CLIPS
    with torch.no_grad():
        feats = eng.encode(headline["mel"])
        em = eng.emissions(feats, bench.BATCH, bench.T_FRAMES, bench.T_FRAMES, headline["labels"], headline["n_labels"], _lib.LA_VARIANT_CTC)[:n].cpu()
    on_err, off_err, em_err, true_err = [], [], [], []
    for b, o in enumerate(headline["oracle"]):
        L = o["L"]
        true_err.append(np.abs(headline["ref"][0][b, :L] * 0.02 - headline["plans"][b][0][:-1] * 0.01))
        idx = headline["labels"][b, :L].cpu().long() - 1
        em_err.append(torch.cat([(em[b, :, 1:1 + L] - o["lp"][:, idx]).abs().flatten(), (em[b, :, 0] - o["ls"][:, 0]).abs()]))
        on_err.append(np.abs(headline["ref"][0][b, :L] * 0.02 - o["on"]))
        off_err.append(np.abs(headline["ref"][1][b, :L] * 0.02 - o["off"]))
    em_err = torch.cat(em_err)
    on_err, off_err = np.concatenate(on_err), np.concatenate(off_err)
    both = np.concatenate([on_err, off_err])
    print(f"bf16 B=32: emission error mean {float(em_err.mean()):.4f} max {float(em_err.max()):.4f}; boundaries equal "
          f"{float((both < 1e-9).mean()):.3f}, onset MAE {on_err.mean():.4f} s, max deviation {both.max():.2f} s; device onsets vs the "
          f"songs' note edges: MAE {np.concatenate(true_err).mean():.4f} s; head fit {headline['fit']}")
    assert headline["fit"]["fit_frame_accuracy"] > 0.9
    assert float(em_err.mean()) < 0.06 and float(em_err.max()) < 0.6
    assert on_err.mean() <= bench.SELFCHECK_TOL_S and off_err.mean() <= bench.SELFCHECK_TOL_S
    assert float((both < 1e-9).mean()) >= 0.90 and both.max() <= 0.2
    assert np.concatenate(true_err).mean() <= 0.1


def test_headline_pipeline_repeats_beside_a_side_stream_load_are_bit_identical(headline):
    """Ten runs of the two-stream pipeline while a third stream keeps the chip busy with unrelated GEMMs (different CU
    availability, different interleaving of the persistent GRU recurrence with the encoder's tiles): frames bit-identical
    every time -- a result that moved with timing would be a race."""
    from lyricalignment_amd import ops
    mask = np.arange(headline["labels"].shape[1])[None, :] < headline["Ls"][:, None]
    side = torch.cuda.Stream()
    a = torch.randn((8192, 2048), device="cuda").to(torch.bfloat16)
    w = torch.randn((2048, 2048), device="cuda").to(torch.bfloat16)
    c = torch.empty((8192, 2048), dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    for rep in range(10):
        with torch.cuda.stream(side):
            for _ in range(40 + 10 * rep):                    # a different amount of foreign work each time
                ops.gemm(a, w, c)
        got = _run_pipeline(headline, [3], head_group=2 if rep % 2 == 0 else 1)
        side.synchronize()
        for on, off, st in got:
            assert (st == 0).all()
            assert (on[mask] == headline["ref"][0][mask]).all() and (off[mask] == headline["ref"][1][mask]).all(), f"repeat {rep} differs"
    headline["eng"].check_gru()


def test_float32_parity_mode_at_the_headline_size_equals_the_oracle(headline):
    """bench.py `modes.f32_parity`: BASELINE configs[1]'s own batch (32 x 30 s, the fitted head) in the reference's precision, on the f16 matrix
    pipe at float32 accuracy (the route every Linear takes from 9 clips on), through the bench's pipeline shape.  "Onset / offset MAE identical to
    the reference" at the headline size: EVERY boundary of the oracle's clips equals the fp32 oracle's own end-to-end result (no tolerance), their
    emission log-probs (range 0 .. -35) are within 2e-3, the pipeline's frames equal the single-stream call's for the whole batch, and the
    bfloat16 headline differs from this mode in a few per cent of the boundaries only (what `modes.boundaries_equal_to_f32_parity` reports)."""
    import ctypes
    from lyricalignment_amd import _lib
    from lyricalignment_amd.engine import PipelinedAligner
    from lyricalignment_amd.module.align_model import AlignModel
    bench, dims = headline["bench"], headline["dims"]
    m32 = AlignModel(headline["model"].whisper_model, embed_dim=dims.n_audio_state, hidden_dim=bench.HIDDEN, output_dim=bench.VOCAB, device="cuda:0",
                     compute_dtype=torch.float32).eval()
    m32.align_rnn.load_state_dict(headline["model"].align_rnn.state_dict())
    L = _lib.lib()
    with torch.no_grad():
        eng = m32.engine()
        L.la_timer_reset(); L.la_timer_sample(1000003); L.la_timer_enable(b"gemm_f16x2")
        ref = eng.align_mel(headline["mel"], headline["labels"], headline["n_labels"], n_frames=bench.T_FRAMES, use_ctc=True)
        torch.cuda.synchronize()
        L.la_timer_disable()
        ms, timed, work, seen = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_int64(0)
        L.la_timer_read_work(ctypes.byref(ms), ctypes.byref(timed), ctypes.byref(work), ctypes.byref(seen))
        L.la_timer_reset(); L.la_timer_sample(1)
        assert seen.value >= 4 * 24 + 2                                          # every block's four Linears and the head's two projections
        assert int((ref[3] != 0).sum()) == 0
        pipe = PipelinedAligner(eng, head_group=bench.DEFAULT_HEAD_GROUP, encoder_streams=bench.DEFAULT_ENCODER_STREAMS)
        outs = [pipe.submit(headline["mel"], headline["labels"], headline["n_labels"], n_frames=bench.T_FRAMES, use_ctc=True) for _ in range(5)]
        pipe.drain()
        feats = eng.encode(headline["mel"])
        em = eng.emissions(feats, bench.BATCH, bench.T_FRAMES, bench.T_FRAMES, headline["labels"], headline["n_labels"], _lib.LA_VARIANT_CTC)[:N_ORACLE_CLIPS].cpu()
    on, off = ref[0].cpu().numpy(), ref[1].cpu().numpy()
    mask = np.arange(headline["labels"].shape[1])[None, :] < headline["Ls"][:, None]
    for o in outs:
        assert (o[0].cpu().numpy()[mask] == on[mask]).all() and (o[1].cpu().numpy()[mask] == off[mask]).all() and int((o[3] != 0).sum()) == 0
    for b, o in enumerate(headline["oracle"]):
        Lb = o["L"]
        assert (on[b, :Lb] * 0.02 == o["on"]).all() and (off[b, :Lb] * 0.02 == o["off"]).all(), f"clip {b}: boundaries differ from the oracle's"
        idx = headline["labels"][b, :Lb].cpu().long() - 1
        err = torch.cat([(em[b, :, 1:1 + Lb] - o["lp"][:, idx]).abs().flatten(), (em[b, :, 0] - o["ls"][:, 0]).abs()])
        assert float(err.max()) < 2e-3, (b, float(err.max()))
    same = int((on[mask] == headline["ref"][0][mask]).sum() + (off[mask] == headline["ref"][1][mask]).sum())
    print(f"float32-parity mode at B = 32: all {sum(2 * o['L'] for o in headline['oracle'])} boundaries of the oracle's clips equal; bf16 headline vs this mode: "
          f"{same} of {2 * int(mask.sum())} boundaries equal")
    assert same >= 0.9 * 2 * int(mask.sum())
    eng.check_gru()
