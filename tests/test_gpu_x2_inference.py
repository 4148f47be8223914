"""float32 INFERENCE on the f16 matrix pipe at float32 accuracy (library option "x2_inference"; csrc/la_f32x2.hip, la_model.cpp): the reference
is float32 end to end (module/align_model.py:72-123), and "seconds equal the oracle's" holds in that mode only.  gfx950 multiplies float32
operands at 1/16 of its 16-bit MFMA rate; the route under test computes every large product as three f16 products over split operands.
Pieces (LayerNorm into planes, the normaliser product of the output Linear, the recurrence without gate stores) against float64 / the
float32-MFMA kernels, then the model-level entry points route against route and against the oracle."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_layernorm_into_planes_reconstructs_layernorm():
    """la_layernorm_f16x2: (hi + lo) * inv_scale == LayerNorm(x) gamma + beta to 2^-21 of each element / 2^-38 of the row's maximum plus the
    float32 rounding of the normalisation itself; inverse scales are powers of two with the row maximum in [2^13, 2^14); d = 384 .. 4096,
    a zero row, a constant row, a row pitch wider than d, kp > d (zero tail)."""
    from lyricalignment_amd import f32x2
    for d, pitch, kp in ((1024, 1024, 1024), (384, 400, 384), (1280, 1280, 1408), (4096, 4096, 4096)):
        x = _rand(333, pitch, seed=d) * torch.exp(_rand(333, 1, seed=d + 1) * 2.0)
        x[:, 7] *= 40.0
        x[5] = 0.0
        x[6] = 3.25
        g, b = 1.0 + 0.3 * _rand(d, seed=3), 0.2 * _rand(d, seed=4)
        P = f32x2.layernorm_split(x.cuda()[:, :d], g.cuda(), b.cuda(), kp)
        ref = torch.nn.functional.layer_norm(x[:, :d].double(), (d,), g.double(), b.double(), eps=1e-5)
        planes, inv = P.planes.cpu(), P.inv_scale.cpu()
        assert planes.shape == (333, 2, kp) and (kp == d or float(planes[:, :, d:].abs().max()) == 0.0)
        rec = (planes[:, 0, :d].double() + planes[:, 1, :d].double()) * inv.double()[:, None]
        rowmax = ref.abs().amax(dim=1, keepdim=True)
        # float32 LayerNorm arithmetic: a few ulp of the row's largest normalised value (|xhat| <= sqrt(d))
        bound = ref.abs() * 2.0 ** -21 + rowmax * 2.0 ** -38 + 8 * 2.0 ** -24 * (rowmax + 1.0)
        assert bool(((rec - ref).abs() <= bound).all()), (d, float(((rec - ref).abs() / bound).max()))
        m, _ = torch.frexp(inv)
        assert bool((m == 0.5).all())
        scaled = rec.abs().amax(dim=1) / inv.double()
        nz = scaled > 0
        assert bool(((scaled[nz] >= 2.0 ** 13 * (1 - 1e-6)) & (scaled[nz] < 2.0 ** 14)).all())


@pytest.mark.parametrize("variant", ["ctc", "plain"])
def test_fc_emissions_normaliser_on_the_f16_pipe_matches_float64(variant):
    """la_fc_emissions_x2: emissions against (float64 logits rounded to float32) -> the oracle's emission prep -- the reference's own float32
    formulas incl. the naive log(1 - sigmoid) (utils/alignment.py:125-129), whose cancellation both sides share -- and, on the plain
    variant (a pure log-softmax), not worse than the float32-MFMA kernel's."""
    from lyricalignment_amd import engine, ops
    from oracle import model_oracle as mo
    B, T, K, V = 9, 1500, 256, 2000
    act = _rand(B * T, K, seed=50)
    w = _rand(V, K, seed=51, scale=2.0 / K ** 0.5)
    bias = _rand(V, seed=52, scale=0.5)
    bias[-1] = 0.3
    rs = np.random.RandomState(53)
    Lmax = 26
    ncls = V - 2 if variant == "ctc" else V - 1
    labels = torch.from_numpy(rs.randint(1, ncls + 1, size=(B, Lmax)).astype(np.int32))
    n_labels = torch.from_numpy(rs.randint(1, Lmax + 1, size=B).astype(np.int32))
    var = 1 if variant == "ctc" else 0
    wd = w.cuda()
    em_x2 = ops.fc_emissions(act.cuda(), wd, bias.cuda(), B, T, labels.cuda(), n_labels.cuda(), var, w_x2=engine._x2_planes(wd)).cpu()
    em_32 = ops.fc_emissions(act.cuda(), wd, bias.cuda(), B, T, labels.cuda(), n_labels.cuda(), var).cpu()
    logits = (act.double() @ w.double().T + bias.double()).float().reshape(B, T, V)
    lp, ls = (mo.emission_prep_ctc if variant == "ctc" else mo.emission_prep_plain)(logits)
    e_x2 = e_32 = 0.0
    for b in range(B):
        L = int(n_labels[b])
        idx = labels[b, :L].long() - 1
        for em, name in ((em_x2, "x2"), (em_32, "f32")):
            e = max(float((em[b, :, 0].double() - ls[b, :, 0]).abs().max()), float((em[b, :, 1:1 + L].double() - lp[b][:, idx]).abs().max()))
            if name == "x2":
                e_x2 = max(e_x2, e)
            else:
                e_32 = max(e_32, e)
    print(f"fc_emissions {variant}: max |err| x2 {e_x2:.2e}, float32 kernel {e_32:.2e}")
    # (ctc: log(1 - sigmoid) amplifies one float32 ulp of the silence logit by 1 / (1 - sigmoid) -- on either kernel)
    assert e_x2 < (5e-4 if variant == "ctc" else 2e-5) and e_x2 <= 1.5 * e_32 + 2e-6


@pytest.mark.parametrize("B,T,H", [(5, 40, 64), (32, 50, 128), (40, 23, 384), (17, 300, 384)])
def test_float32_recurrence_on_the_f16_pipe_matches_the_float32_kernel_and_nn_gru(B, T, H):
    """la_gru_layer(LA_F32) from 5 clips on = gru_train_x2_kernel without the gate stores (+ Mish): against torch.nn.GRU in float64, and
    against the float32-MFMA kernel (option x2_inference = 0)."""
    from lyricalignment_amd import _lib, ops
    I = 48
    gru = torch.nn.GRU(I, H, num_layers=1, batch_first=True, bidirectional=True).double()
    g = torch.Generator().manual_seed(40 + T)
    with torch.no_grad():
        for prm in gru.parameters():
            prm.copy_((torch.rand(prm.shape, generator=g, dtype=torch.float64) * 2 - 1) * (1.5 / H ** 0.5))
        x = torch.randn(B, T, I, generator=g, dtype=torch.float64)
        ref, _ = gru(x)
        gi = torch.stack([x @ gru.weight_ih_l0.T + gru.bias_ih_l0, x @ gru.weight_ih_l0_reverse.T + gru.bias_ih_l0_reverse], dim=2).float()
        w_hh = torch.stack([gru.weight_hh_l0, gru.weight_hh_l0_reverse]).float()
        b_hh = torch.stack([gru.bias_hh_l0, gru.bias_hh_l0_reverse]).float()
    args = (gi.contiguous().cuda(), w_hh.contiguous().cuda(), b_hh.contiguous().cuda())
    out, mish, flag = ops.gru_layer(*args, want_mish=True)
    with _lib.option("x2_inference", 0):
        out0, mish0, flag0 = ops.gru_layer(*args, want_mish=True)
    assert int(flag.item()) == 0 and int(flag0.item()) == 0
    e1, e0 = float((out.cpu().double() - ref).abs().max()), float((out0.cpu().double() - ref).abs().max())
    print(f"GRU B={B} T={T} H={H}: max |err| vs float64 x2 {e1:.2e}, float32 kernel {e0:.2e}")
    assert e1 < 2e-5 and e1 <= 2.0 * e0 + 2e-6
    np.testing.assert_allclose(mish.cpu().numpy(), torch.nn.functional.mish(ref).float().numpy(), rtol=0, atol=2e-5)
    assert not torch.equal(out, out0) or H < 64          # (two different kernels ran)


def _small_model(d=256, L=2, H=4, hidden=128, V=500, seed=3):
    from lyricalignment_amd import whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    dims = wc.ModelDimensions(n_audio_state=d, n_audio_head=H, n_audio_layer=L, n_text_state=d, n_text_head=H, n_text_layer=1, n_vocab=300, n_text_ctx=32)
    wm = wc.build_model(dims=dims, seed=seed, with_decoder=False)
    model = AlignModel(wm, embed_dim=d, hidden_dim=hidden, output_dim=V, device="cuda", compute_dtype=torch.float32).eval()
    wc.init_align_head(model, seed=7)
    return model, dims


class _count:
    def __init__(self, family):
        self.family, self.n = family, 0

    def __enter__(self):
        from lyricalignment_amd import _lib
        L = _lib.lib()
        L.la_timer_reset(); L.la_timer_sample(1000003); L.la_timer_enable(self.family.encode())
        return self

    def __exit__(self, *exc):
        from lyricalignment_amd import _lib
        L = _lib.lib()
        torch.cuda.synchronize()
        L.la_timer_disable()
        ms, timed, work, seen = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_int64(0)
        L.la_timer_read_work(ctypes.byref(ms), ctypes.byref(timed), ctypes.byref(work), ctypes.byref(seen))
        self.n = int(seen.value)
        L.la_timer_reset(); L.la_timer_sample(1)
        return False


def test_model_level_routes_agree_with_each_other_and_the_oracle():
    """la_encoder_forward / la_align_head_forward with float32 weights that carry the f16x2 planes, 48 clips of a d = 256 two-block model
    (every Linear in the 256 x 256 kernel's domain): the f16x2 route is the one that runs by default, option x2_inference = 0 gives the
    float32-MFMA route; both within 2e-4 of the float32 oracle on encoder output and logits (4 clips checked), the f16x2 route not further
    from a float64 run of the oracle than the float32-MFMA route, and the same seconds from both and from the oracle's own DP.
    The op-by-op Python sequence (LA_ENGINE_PY) gives the C call's bits."""
    from lyricalignment_amd import _lib, engine as eng_mod
    from lyricalignment_amd.utils import alignment as ua
    from oracle import alignment_oracle as ao, model_oracle as mo
    model, dims = _small_model()
    B = 48
    rs = np.random.RandomState(0)
    mel = torch.from_numpy(rs.uniform(-1, 1, size=(B, 80, 3000)).astype(np.float32))
    labels = torch.from_numpy(rs.randint(1, 499, size=(B, 12)))
    p = {"encoder." + k: v.detach().float().cpu() for k, v in model.whisper_model.encoder.state_dict().items()}
    p.update({"align_rnn." + k: v.detach().float().cpu() for k, v in model.align_rnn.state_dict().items()})
    chk = [0, 1, 17, 47]
    with torch.no_grad():
        enc_ref = mo.encoder_forward(p, mel[chk], n_head=dims.n_audio_head)
        log_ref = mo.gru_head_forward(p, enc_ref)
        p64 = {k: v.double() for k, v in p.items()}
        enc64 = mo.encoder_forward(p64, mel[chk].double(), n_head=dims.n_audio_head)
        want = ao.perform_viterbi_ctc(log_ref, labels[chk])
        res = {}
        for route, opt in (("x2", 1), ("f32", 0)):
            with _lib.option("x2_inference", opt), _count("gemm_f16x2") as c:
                model._engine = None
                e = model.engine()
                enc = e.encode(mel.cuda(), out_dtype=torch.float32).view(B, 1500, -1)[chk].cpu()
                lg, _ = model.forward(mel.cuda())
                secs = model.align(mel=mel.cuda(), labels=labels, use_ctc=True, get_orig_len=False)
            res[route] = (enc, lg[chk].cpu(), [secs[i] for i in chk], c.n)
        assert res["x2"][3] >= 3 * (4 * 2) and res["f32"][3] == 0, (res["x2"][3], res["f32"][3])
        for route in ("x2", "f32"):
            np.testing.assert_allclose(res[route][0].numpy(), enc_ref.numpy(), rtol=0, atol=2e-4)
            np.testing.assert_allclose(res[route][1].numpy(), log_ref.numpy(), rtol=0, atol=2e-4)
            assert res[route][2] == want, route
        e_x2 = float((res["x2"][0].double() - enc64).abs().max())
        e_32 = float((res["f32"][0].double() - enc64).abs().max())
        print(f"encoder output max |err| vs float64 oracle: f16x2 route {e_x2:.2e}, float32-MFMA route {e_32:.2e}")
        assert e_x2 <= 1.5 * e_32 + 1e-6
        # the op-by-op sequence of the same route: identical bits
        old = eng_mod.ENGINE_PY
        eng_mod.ENGINE_PY = True
        try:
            model._engine = None
            enc_py = model.engine().encode(mel.cuda(), out_dtype=torch.float32).view(B, 1500, -1)[chk].cpu()
        finally:
            eng_mod.ENGINE_PY = old
            model._engine = None
        assert torch.equal(enc_py, res["x2"][0])


def test_documented_limits_raise_outside_and_work_at_the_edge():
    """INTEGRATION.md "Limits that differ from the reference": the alignment DP takes up to 4095 labels per utterance, the CTC loss lattice
    up to 511; one label beyond raises NotImplementedError / ValueError instead of computing garbage; a GRU launch set beyond the
    co-residency cap is refused by la_gru_layer (callers slice), and the model-level call slices transparently."""
    from lyricalignment_amd import finetune as ft, ops
    T = 8300
    for L, ok in ((4095, True), (4096, False)):
        em = torch.zeros((1, T, L + 1), dtype=torch.float32, device="cuda")
        labels = (torch.arange(L, dtype=torch.int32, device="cuda") % 7 + 1).view(1, L)
        args = (em, labels, torch.tensor([L], dtype=torch.int32, device="cuda"), torch.tensor([T], dtype=torch.int32, device="cuda"))
        if ok:
            onset, offset, score, status = ops.viterbi_batch(*args)
            assert int(status[0]) == 0 and int(onset[0, 0]) >= 0 and int(offset[0, L - 1]) == T          # (offset = last frame + 1: utils/alignment.py:183)
        else:
            with pytest.raises(NotImplementedError):
                ops.viterbi_batch(*args)
    V = 40
    logits = torch.randn(1, 1100, V + 1, generator=torch.Generator().manual_seed(1)).cuda()
    for L, ok in ((511, True), (512, False)):
        lab = (torch.arange(L) % (V - 2) + 1).view(1, L)
        if ok:
            l3, dlog = ft.multitask_loss(logits, None, lab, vocab_size=V)
            assert torch.isfinite(l3[2]) and torch.isfinite(dlog).all()
        else:
            with pytest.raises(ValueError):
                ft.multitask_loss(logits, None, lab, vocab_size=V)
    H = 384
    gi = torch.zeros((608, 2, 2, 3 * H), dtype=torch.float32, device="cuda")
    with pytest.raises(NotImplementedError):                       # 608 clips: 38 groups x 2 directions x 3 workgroups = 228 > the 224 co-resident workgroups of one launch
        ops.gru_layer(gi, torch.zeros((2, 3 * H, H), dtype=torch.bfloat16, device="cuda"), torch.zeros((2, 3 * H), device="cuda"))


def test_transposed_split_with_one_operand_scale_keeps_the_documented_precision():
    """Round-5 advisor finding, pinned as documented behaviour: la_split_f16x2_t_tmax gives the transposed planes ONE power-of-two scale (from
    the operand's maximum, left by its plain split) instead of one per column.  A weight gradient dW = dy^T x computed from such planes, dy with
    columns spanning 2^36 of dynamic range, against float64, ROW BY ROW of dW (= column of dy): rows whose dy column lies within 2^17 of the
    operand's maximum keep the full 22 bits (relative error of the row < 2^-19), rows down to 2^28 below keep at least the hi plane's 11 bits
    (< 2^-9), and below that the ABSOLUTE error stays under 2^-36 of the largest row (half's subnormal floor) -- entries that small are
    below AdamW's eps after the global-norm clip.  The per-column form (LA_F32X2_TMAX=0) keeps 22 bits in every row."""
    from lyricalignment_amd import f32x2
    M, N, K = 24000, 512, 1024
    g = torch.Generator().manual_seed(2)
    dy = torch.randn(M, N, generator=g)
    expo = torch.linspace(0, -36, N)
    dy = dy * torch.exp2(expo)[None, :]
    x = torch.randn(M, K, generator=g)
    ref = dy.double().T @ x.double()
    dyd, xd = dy.cuda(), x.cuda()
    mp = f32x2.padded_k(N, K, M)
    omax = f32x2.OperandMax(dyd.device)
    f32x2.split(dyd, omax=omax)                                                    # the plain split leaves the operand's maximum
    res = {}
    for name, om in (("one scale", omax), ("per column", None)):
        dw = f32x2.gemm(f32x2.split_t(dyd, mp, omax=om), f32x2.split_t(xd, mp)).cpu().double()
        res[name] = (dw - ref).norm(dim=1) / ref.norm(dim=1)
    top = ref.norm(dim=1).max()
    rel, per_col = res["one scale"], res["per column"]
    assert float(per_col.max()) < 2.0 ** -19
    near, mid, far = expo > -16, (expo <= -18) & (expo > -27), expo <= -30
    assert float(rel[near].max()) < 2.0 ** -19 and float(rel[mid].max()) < 2.0 ** -9
    abs_far = (rel * ref.norm(dim=1))[far]
    assert float(abs_far.max()) < float(top) * 2.0 ** -36
    print(f"dW rows by dy-column magnitude: within 2^16 of the maximum {float(rel[near].max()):.1e}, 2^18..2^27 below {float(rel[mid].max()):.1e}, "
          f"2^30+ below: absolute {float(abs_far.max() / top):.1e} of the largest row; per-column scales {float(per_col.max()):.1e}")
