"""GPU parity of the fine-tune pieces: CE / BCE / CTC losses (forward + gradient w.r.t. the logits) against the
reference's own functions (golden) and torch; fused clip + AdamW against torch.optim.AdamW + clip_grad_norm_."""
import os
import numpy as np
import pytest
import torch

from conftest import load_npz

pytestmark = pytest.mark.gpu


def test_losses_match_reference_golden():
    from lyricalignment_amd import finetune as ft
    z = load_npz("losses.npz")
    V = int(z["vocab_size"])
    logits = torch.from_numpy(z["logits"]).cuda()
    losses, g = ft.multitask_loss(logits, torch.from_numpy(z["frame_labels"]), None, vocab_size=V)
    l = losses.cpu().numpy()
    np.testing.assert_allclose(l[0] + l[1], z["ce"], rtol=2e-6)                  # compute_ce_loss = word CE + silence BCE
    np.testing.assert_allclose(g.cpu().numpy(), z["g_ce"], rtol=0, atol=2e-7)
    losses, g = ft.multitask_loss(logits, None, torch.from_numpy(z["labels"]), vocab_size=V)
    np.testing.assert_allclose(losses.cpu().numpy()[2], z["ctc"], rtol=2e-6)
    np.testing.assert_allclose(g.cpu().numpy(), z["g_ctc"], rtol=0, atol=1e-5)   # fp32 log-space alpha/beta in both; order of log-adds differs
    both, g = ft.multitask_loss(logits, torch.from_numpy(z["frame_labels"]), torch.from_numpy(z["labels"]), vocab_size=V, scale=0.125)
    np.testing.assert_allclose(g.cpu().numpy(), 0.125 * (z["g_ce"] + z["g_ctc"]), rtol=0, atol=2e-6)   # loss / accum_grad_steps


@pytest.mark.parametrize("B,T,V,Ls", [(2, 300, 500, [26, 9]), (3, 120, 2000, [40, 1, 17]), (2, 1500, 21128, [26, 11]),
                                      (3, 200, 300, [31, 1, 5]),      # 63 lattice states: the widest the one-wave lattice takes; a single label
                                      (2, 1501, 64, [3, 30])])        # a frame count that is not a multiple of the 8-step prefetch blocks
def test_losses_match_torch(B, T, V, Ls):
    from lyricalignment_amd import finetune as ft
    rs = np.random.RandomState(B * T)
    logits = torch.from_numpy((rs.randn(B, T, V + 1) * 2).astype(np.float32))
    Lmax = max(Ls)
    labels = torch.full((B, Lmax), -100, dtype=torch.long)
    for b, L in enumerate(Ls):
        lab = rs.randint(1, V, size=L)
        if L > 3:
            lab[2] = lab[1]
        labels[b, :L] = torch.from_numpy(lab)
    fl = torch.full((B, T - 7), -100, dtype=torch.long)
    for b in range(B):
        for k in range(10):
            s = rs.randint(0, T - 30)
            fl[b, s:s + 12] = int(rs.randint(1, V))
    ref = logits.double().requires_grad_(True)   # float64 torch reference: fp32 CTC recursions lose 2-3 digits at T >= 300
    flp = ft.pad_frame_labels(fl, T).clone()
    tgt = flp.clone(); tgt[tgt != -100] -= 1
    ce = torch.nn.functional.cross_entropy(ref[:, :, 1:V].transpose(1, 2), tgt)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(ref[:, :, V], (flp == -100).double())
    lsm = torch.nn.functional.log_softmax(ref[:, :, :V], dim=2).transpose(0, 1)
    ctc = torch.nn.functional.ctc_loss(lsm, labels, torch.full((B,), T, dtype=torch.long), (labels != -100).sum(1))
    (ce + bce + ctc).backward()
    losses, g = ft.multitask_loss(logits.cuda(), fl, labels, vocab_size=V)
    l = losses.cpu().numpy()
    np.testing.assert_allclose(l, [ce.item(), bce.item(), ctc.item()], rtol=2e-5)
    np.testing.assert_allclose(g.cpu().numpy(), ref.grad.numpy(), rtol=0, atol=2e-6)


def test_fused_clip_adamw_matches_torch():
    from lyricalignment_amd import finetune as ft
    g = torch.Generator().manual_seed(0)
    p1, p2 = torch.randn(100003, generator=g), torch.randn(5000, generator=g)
    ref1, ref2 = torch.nn.Parameter(p1.clone()), torch.nn.Parameter(p2.clone())
    opt = torch.optim.AdamW([{"params": [ref1], "lr": 5e-3}, {"params": [ref2], "lr": 5e-6}], weight_decay=1e-5)
    mine = ft.FlatAdamW([{"params": p1.clone().cuda(), "lr": 5e-3}, {"params": p2.clone().cuda(), "lr": 5e-6}], weight_decay=1e-5)
    for step in range(4):
        g1, g2 = torch.randn(100003, generator=g) * (3.0 if step % 2 else 0.001), torch.randn(5000, generator=g)
        ref1.grad, ref2.grad = g1.clone(), g2.clone()
        total = torch.nn.utils.clip_grad_norm_([ref1, ref2], 1.0)
        opt.step()
        ss = mine.step([g1.cuda(), g2.cuda()], max_norm=1.0)
        np.testing.assert_allclose(float(ss.item()) ** 0.5, float(total), rtol=1e-5)
        np.testing.assert_allclose(mine.groups[0]["params"].cpu().numpy(), ref1.detach().numpy(), rtol=0, atol=2e-6)
        np.testing.assert_allclose(mine.groups[1]["params"].cpu().numpy(), ref2.detach().numpy(), rtol=0, atol=2e-6)


def _ref_rnn(D, H, V, seed):
    torch.manual_seed(seed)
    gru = torch.nn.GRU(D, H, num_layers=2, batch_first=True, bidirectional=True, dropout=0.0)
    fc = torch.nn.Linear(2 * H, V)
    return gru, fc


@pytest.mark.parametrize("B,T,D,H,V", [(3, 20, 64, 64, 50), (2, 150, 128, 128, 333), (33, 9, 64, 64, 40), (4, 12, 64, 64, 30), (20, 6, 64, 64, 30),
                                         (6, 40, 128, 128, 60), (17, 25, 128, 384, 50), (7, 30, 64, 192, 40)])
def test_head_backward_matches_torch_autograd(B, T, D, H, V):
    """HeadFunction (HIP forward + backward of GRUx2 -> Mish -> Linear) vs torch autograd on nn.GRU / nn.Mish / nn.Linear."""
    from lyricalignment_amd.head_train import HeadFunction
    gru, fc = _ref_rnn(D, H, V, 70 + T)
    g = torch.Generator().manual_seed(71)
    x = torch.randn(B, T, D, generator=g)
    wgt = torch.randn(B, T, V, generator=g)
    xr = x.clone().requires_grad_(True)
    out, _ = gru(xr)
    ref = fc(torch.nn.functional.mish(out))
    (ref * wgt).sum().backward()
    names = []
    for l in range(2):
        for sfx in ("", "_reverse"):
            names += [f"weight_ih_l{l}{sfx}", f"weight_hh_l{l}{sfx}", f"bias_ih_l{l}{sfx}", f"bias_hh_l{l}{sfx}"]
    params = [getattr(gru, n).detach().clone().cuda().requires_grad_(True) for n in names]
    params += [fc.weight.detach().clone().cuda().requires_grad_(True), fc.bias.detach().clone().cuda().requires_grad_(True)]
    xd = x.clone().cuda().requires_grad_(True)
    logits = HeadFunction.apply(xd, 0.0, True, *params)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=2e-5)
    (logits * wgt.cuda()).sum().backward()
    scale = max(1.0, float(T) ** 0.5)
    for n, p_ in zip(names, params[:16]):
        want = getattr(gru, n).grad
        np.testing.assert_allclose(p_.grad.cpu().numpy(), want.numpy(), rtol=0, atol=2e-4 * scale * max(1.0, float(want.abs().max())), err_msg=n)
    np.testing.assert_allclose(params[16].grad.cpu().numpy(), fc.weight.grad.numpy(), rtol=0, atol=2e-4 * scale)
    np.testing.assert_allclose(params[17].grad.cpu().numpy(), fc.bias.grad.numpy(), rtol=0, atol=2e-4 * scale)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=0, atol=2e-4)


def test_head_only_finetune_step_through_alignmodel():
    """Frozen-encoder fine-tune step through the drop-in surface: frame_manual_forward under autograd, the reference's loss
    formulas (torch ops, as train_multitask.py computes them), backward through the HIP head, fused clip + AdamW.
    The loss must go down on a fixed tiny batch, and encoder parameters must receive no gradient."""
    from lyricalignment_amd import finetune as ft, whisper_compat as wc
    from lyricalignment_amd.head_train import head_params
    from lyricalignment_amd.module.align_model import AlignModel
    dims = wc.ModelDimensions(n_audio_state=128, n_audio_head=2, n_audio_layer=1, n_text_state=128, n_text_head=2, n_text_layer=0)
    model = AlignModel(wc.build_model(dims=dims, seed=81, std=0.05), embed_dim=128, hidden_dim=64, output_dim=41, dropout=0.15,
                       freeze_encoder=True, device="cuda").to("cuda")
    for p_ in model.whisper_model.parameters():
        p_.requires_grad_(False)
    model.train()
    rs = np.random.RandomState(82)
    audios = [(rs.randn(16000) * 0.1).astype(np.float32), (rs.randn(12000) * 0.1).astype(np.float32)]
    labels = torch.tensor([[3, 7, 7, 12], [5, 9, -100, -100]]).cuda()
    frame_labels = torch.full((2, 50), -100, dtype=torch.long); frame_labels[0, 5:20] = 3; frame_labels[1, 10:30] = 9
    V = 40
    hp = head_params(model.align_rnn)
    flat = torch.cat([p_.detach().reshape(-1) for p_ in hp]).contiguous()
    opt = ft.FlatAdamW([{"params": flat, "lr": 5e-3}], weight_decay=1e-5)
    losses = []
    for it in range(6):
        logits, _ = model.frame_manual_forward(audios, get_orig_len=False)
        assert logits.requires_grad and tuple(logits.shape) == (2, 1500, 41)
        ce = torch.nn.functional.cross_entropy(logits[:, :, 1:V].transpose(1, 2), torch.where(ft.pad_frame_labels(frame_labels, 1500) == -100, -100, ft.pad_frame_labels(frame_labels, 1500) - 1).cuda())
        lsm = torch.nn.functional.log_softmax(logits[:, :, :V], dim=2).transpose(0, 1)
        ctc = torch.nn.functional.ctc_loss(lsm, labels, torch.full((2,), 1500, dtype=torch.long), (labels != -100).sum(1))
        loss = ce + ctc
        for p_ in hp:
            p_.grad = None
        loss.backward()
        assert all(p_.grad is not None for p_ in hp) and all(p_.grad is None for p_ in model.whisper_model.parameters())
        grads = torch.cat([p_.grad.reshape(-1) for p_ in hp]).contiguous()
        opt.step([grads], max_norm=1.0)
        with torch.no_grad():                      # scatter the flat bucket back into the module's parameters
            off = 0
            for p_ in hp:
                p_.copy_(flat[off: off + p_.numel()].view_as(p_)); off += p_.numel()
        losses.append(float(loss.item()))
    assert losses[-1] < losses[0], losses


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("B,T,H", [(1, 96, 2), (2, 1500, 1), (1, 333, 3), (3, 64, 2), (2, 130, 1)])
def test_attention_backward_matches_torch_autograd(B, T, H, fused):
    """dQ/dK/dV against torch autograd: the fused kernel la_attention_bwd_f32 (forward output given: scores recomputed per 64 x 64
    tile; from 128 queries and keys on la_attention_bwd_f16x2, the same sweeps on the f16 pipe) and the round-1 composition (batched f32 MFMA
    GEMMs over whole score tiles + row softmax kernels)."""
    from lyricalignment_amd import encoder_train as et
    d = 64 * H
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(B * T, 3 * d, generator=g)
    qkv[:, :d] *= 0.35          # q pre-scaled: logits of a few units, a peaky softmax
    datt = torch.randn(B * T, d, generator=g)
    ref_in = qkv.clone().requires_grad_(True)
    q, k, v = [t.view(B, T, H, 64).permute(0, 2, 1, 3) for t in ref_in.split(d, dim=1)]
    o = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, d)
    o.backward(datt)
    lse = None
    if fused and T % 2 == 0:      # half of the shapes: the row statistic comes from the forward kernel (la_attention_lse_f32), as in training
        from lyricalignment_amd import ops
        qd = qkv.cuda()
        lse = torch.empty((B, H, T), dtype=torch.float32, device="cuda")
        o_dev = ops.attention_ex(qd[:, :d], qd[:, d:2 * d], qd[:, 2 * d:], B, T, T, H, lse=lse)
        np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(q @ k.transpose(-1, -2), dim=-1).detach().numpy(), rtol=0, atol=2e-5)
        np.testing.assert_allclose(o_dev.cpu().numpy(), o.detach().numpy(), rtol=0, atol=2e-5)
    got = et.attention_bwd(qkv.cuda(), datt.cuda(), B, T, H, att=o.detach().cuda() if fused else None, lse=lse).cpu()
    for name, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        assert _rel(got[:, sl], ref_in.grad[:, sl]) < 2e-4, name     # float32 tolerance (north_star: 1e-3)


@pytest.mark.parametrize("B,Tq,Tk,H,causal,spread", [(2, 1500, 1500, 2, False, 1.0), (1, 333, 333, 3, True, 1.0), (3, 130, 1500, 1, False, 1.0),
                                                     (1, 256, 128, 2, False, 1.0), (2, 448, 448, 8, True, 1e-4), (1, 700, 515, 2, False, 3e3)])
def test_attention_forward_f16x2_matches_float64_at_float32_accuracy(B, Tq, Tk, H, causal, spread):
    """la_attention_lse_f16x2 (operands as half planes with one power-of-two scale per clip and head, three f16 MFMAs per product)
    against a float64 attention, next to the float32-MFMA kernel la_attention_lse_f32 on the same operands: out and lse no further
    from float64 than 1.5 x the float32 kernel's error (+ 1e-7 of the largest magnitude), and inside the float32 tolerance of the
    other attention tests.  Ragged last query / key tiles, causal masks, cross shapes (q_len != kv_len), 8 heads (the XCD-aware
    block order), operands as column slices of packed projections, heads of very different magnitude (`spread`: every other head's
    k and v scaled -- the per-head scales), a key / value row of zeros."""
    from lyricalignment_amd import ops
    d = 64 * H
    g = torch.Generator().manual_seed(Tq * 3 + Tk)
    q0 = torch.randn(B * Tq, d, generator=g) * 0.35
    kv0 = torch.randn(B * Tk, 2 * d, generator=g)
    kv0[7] = 0.0
    if H > 1:
        for hh in range(1, H, 2):
            kv0[:, d + 64 * hh: d + 64 * hh + 64] *= spread            # v of the odd heads
            kv0[:, 64 * hh: 64 * hh + 64] *= min(spread, 1.0) ** 0.5   # k of the odd heads (kept from saturating the softmax)
    q = q0.double().view(B, Tq, H, 64).permute(0, 2, 1, 3)
    k, v = [t.view(B, Tk, H, 64).permute(0, 2, 1, 3) for t in kv0.double().split(d, dim=1)]
    sc = q @ k.transpose(-1, -2)
    if causal:
        sc = sc + torch.full((Tq, Tk), float("-inf"), dtype=torch.float64).triu(1)
    ref = (torch.softmax(sc, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * Tq, d)
    ref_lse = torch.logsumexp(sc, dim=-1)
    qd, kvd = q0.cuda(), kv0.cuda()
    res = {}
    for name, flag in (("f16x2", True), ("f32", False)):
        ops.ATTN_F16X2 = flag
        try:
            lse = torch.empty((B, H, Tq), dtype=torch.float32, device="cuda")
            out = ops.attention_ex(qd, kvd[:, :d], kvd[:, d:], B, Tq, Tk, H, causal=causal, lse=lse)
        finally:
            ops.ATTN_F16X2 = True
        # per head: the heads differ by orders of magnitude
        eo = ((out.cpu().double() - ref).abs().view(B * Tq, H, 64).amax(dim=(0, 2)) / ref.abs().view(B * Tq, H, 64).amax(dim=(0, 2))).max()
        res[name] = (float(eo), float((lse.cpu().double() - ref_lse).abs().max()))
    print(res)
    assert res["f16x2"][0] <= 1.5 * res["f32"][0] + 1e-7 and res["f16x2"][0] < 2e-5, res
    assert res["f16x2"][1] <= 1.5 * res["f32"][1] + 5e-7 and res["f16x2"][1] < 2e-5, res


def test_encoder_weight_cache_follows_the_parameters():
    """encoder_train._LAYER_CACHE (a block's packed q|k|v projection and the split planes of its weights, kept across the micro-steps of an
    accumulation window): the same output on a second call, a changed output after an in-place update of a weight (the version moved) equal
    to the one computed with an empty cache, no entry shared between two models whose parameters happen to sit at the same addresses, and
    FlatAdamW.step leaves every parameter view of its flat buffer with a new version."""
    from oracle import model_oracle as mo
    from lyricalignment_amd import encoder_train as et, finetune as ft
    d, L = 256, 2
    names = et.encoder_param_names(L)

    def make(seed):
        p = mo.random_encoder_params(d, L, seed=seed)
        prm = [torch.nn.Parameter(torch.as_tensor(p["encoder." + n] if ("encoder." + n) in p else p[n]).float().cuda()) for n in names]
        return prm, torch.as_tensor(p["encoder.positional_embedding"]).float().cuda()

    mel = torch.randn(1, 80, 3000, generator=torch.Generator().manual_seed(3)).cuda()
    prm, pos = make(1)
    with torch.no_grad():
        y0 = et.EncoderFunction.apply(mel, pos, d // 64, *prm).clone()
        assert torch.equal(et.EncoderFunction.apply(mel, pos, d // 64, *prm), y0)
        prm[names.index("blocks.0.attn.query.weight")].mul_(1.5)
        y1 = et.EncoderFunction.apply(mel, pos, d // 64, *prm).clone()
        et._LAYER_CACHE.clear()
        assert torch.equal(et.EncoderFunction.apply(mel, pos, d // 64, *prm), y1) and not torch.equal(y1, y0)
        ptrs = [t.data_ptr() for t in prm]
        del prm
        prm2, pos2 = make(2)                                   # (often the same addresses: the allocator hands the freed blocks out again)
        y2 = et.EncoderFunction.apply(mel, pos2, d // 64, *prm2).clone()
        et._LAYER_CACHE.clear()
        assert torch.equal(et.EncoderFunction.apply(mel, pos2, d // 64, *prm2), y2)
    flat = torch.zeros(64, device="cuda")
    views = [flat[:32], flat[32:]]
    v0 = [t._version for t in views]
    opt = ft.FlatAdamW([dict(params=flat, lr=1e-3)])
    opt.step([torch.ones(64, device="cuda")], max_norm=None)
    assert all(t._version > a for t, a in zip(views, v0))


def test_finetuner_steps_never_reuse_derived_weights_of_the_previous_step():
    """Round-5 advisor finding: FineTuner binds parameters as `p.data = flat[...]` views, which keep version counters of their own; the fused
    AdamW writes the flat buffer through a raw pointer, so nothing keyed on p._version alone saw the update and the packed q|k|v projection,
    the decoder's cross projections, the stacked GRU weights and every cached split plane stayed at their first-step values.  Three optimizer
    steps (large learning rates, dropout 0) with the caches alive against the same run with the caches emptied before every forward: the same
    losses and the same parameters; and the Parameters' own versions move with every step."""
    from lyricalignment_amd import encoder_train as et, finetune as ft
    audios, labels, frame_labels, dec_in, dec_out = _tiny_batch()
    runs = []
    for clear in (False, True):
        model = _tiny_full_model(dropout=0.0)
        torch.manual_seed(5)
        with torch.no_grad():
            for p_ in model.align_rnn.parameters():
                p_.copy_(torch.randn(p_.shape) * 0.1)
        tuner = ft.FineTuner(model, lr=2e-2, backbone_lr=2e-2, warmup_steps=0, train_steps=50, vocab_size=40, world=1)
        losses = []
        for it in range(3):
            v0 = [p_._version for g in tuner.groups for p_ in g]
            for _ in range(2):
                if clear:
                    et.clear_weight_cache()
                losses.append(tuner.micro_step(audios, labels, frame_labels, dec_in, dec_out, accum_grad_steps=2).cpu())
            tuner.step()
            assert all(p_._version > a for a, (p_) in zip(v0, [p_ for g in tuner.groups for p_ in g]))
        runs.append((torch.stack(losses), [f.clone().cpu() for f in tuner.flat]))
    assert not torch.equal(runs[0][0][0], runs[0][0][-1])                      # the steps did move the model
    np.testing.assert_allclose(runs[0][0].numpy(), runs[1][0].numpy(), rtol=1e-6, atol=1e-7)
    for a, b in zip(runs[0][1], runs[1][1]):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=1e-6 * float(b.abs().max()))


def test_attention_f16x2_on_sequences_longer_than_a_clip():
    """1700 tokens (beyond the 1536 the per-head split keeps in registers: its two-pass form; 27 key tiles, ragged last tiles of both sweeps):
    forward and backward of the f16x2 attention against the float32-MFMA kernels, self-attention and a cross shape."""
    from lyricalignment_amd import encoder_train as et, ops
    for (B, Tq, Tk, H) in ((1, 1700, 1700, 2), (1, 1600, 300, 1)):
        d = 64 * H
        g = torch.Generator().manual_seed(Tq - Tk)
        q = (torch.randn(B * Tq, d, generator=g) * 0.35).cuda()
        kv = torch.randn(B * Tk, 2 * d, generator=g).cuda()
        do = torch.randn(B * Tq, d, generator=g).cuda()
        res = []
        for flag in (True, False):
            ops.ATTN_F16X2 = flag
            try:
                lse = torch.empty((B, H, Tq), dtype=torch.float32, device="cuda")
                o = ops.attention_ex(q, kv[:, :d], kv[:, d:], B, Tq, Tk, H, lse=lse)
                dq, dkv = torch.empty_like(q), torch.empty_like(kv)
                et.attention_bwd_ex(q, kv[:, :d], kv[:, d:], do, dq, dkv[:, :d], dkv[:, d:], B, Tq, Tk, H, o=o, lse=lse)
            finally:
                ops.ATTN_F16X2 = True
            res.append((o, lse, dq, dkv))
        for a, b, name in zip(res[0], res[1], ("out", "lse", "dq", "dkv")):
            assert _rel(a, b) < 5e-6, (name, _rel(a, b))


def test_layernorm_backward_with_sums_matches_torch_autograd():
    """la_layernorm_bwd_sums_f32 (one pass: dx + the residual gradient, dgamma, dbeta) against torch autograd in float64: the register form
    (d = 1024, 768; rows not a multiple of the 128-row blocks) and the fallback widths (d = 320: la_layernorm_bwd_f32 + column sums + add)."""
    from lyricalignment_amd import encoder_train as et
    for (M, d) in ((3000, 1024), (517, 768), (300, 320)):
        g = torch.Generator().manual_seed(M + d)
        x = torch.randn(M, d, generator=g) * torch.exp(torch.randn(M, 1, generator=g))
        dy, res, gamma = torch.randn(M, d, generator=g), torch.randn(M, d, generator=g), torch.rand(d, generator=g) + 0.5
        xr, gr = x.double().requires_grad_(True), gamma.double().requires_grad_(True)
        br = torch.zeros(d, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-5).backward(dy.double())
        dx, dg, db = et.layernorm_bwd(x.cuda(), dy.cuda(), gamma.cuda(), residual=res.cuda())
        assert _rel(dx.cpu().double(), xr.grad + res.double()) < 2e-5
        assert _rel(dg.cpu().double(), gr.grad) < 2e-5 and _rel(db.cpu().double(), br.grad) < 2e-5
        dx2, _, _ = et.layernorm_bwd(x.cuda(), dy.cuda(), gamma.cuda())
        assert _rel(dx2.cpu().double(), xr.grad) < 2e-5


def test_attention_backward_f16x2_is_deterministic_and_matches_the_float32_sweeps():
    """la_attention_bwd_f16x2 twice on the same operands: the same bits (the split of P and dS runs in inline assembly whose wait states
    before the consuming MFMA are placed by hand -- without them dK differed from run to run); and within 5e-6 of the float32-MFMA sweeps
    (la_attention_bwd_f32) relative to each gradient's largest magnitude, for self-attention with a ragged last tile, heads of different
    magnitude, a causal mask, and a cross shape (q_len != kv_len)."""
    from lyricalignment_amd import encoder_train as et, ops
    for (B, Tq, Tk, H, causal) in ((2, 1500, 1500, 2, False), (1, 333, 333, 3, True), (2, 200, 1500, 1, False)):
        d = 64 * H
        g = torch.Generator().manual_seed(Tq + Tk)
        q = (torch.randn(B * Tq, d, generator=g) * 0.35).cuda()
        kv = torch.randn(B * Tk, 2 * d, generator=g)
        kv[:, d + 64 * (H - 1):] *= 1e-3
        kv = kv.cuda()
        do = (torch.randn(B * Tq, d, generator=g) * torch.exp(torch.randn(B * Tq, 1, generator=g))).cuda()
        lse = torch.empty((B, H, Tq), dtype=torch.float32, device="cuda")
        o = ops.attention_ex(q, kv[:, :d], kv[:, d:], B, Tq, Tk, H, causal=causal, lse=lse)
        outs = []
        for flag in (True, True, False):
            ops.ATTN_F16X2 = flag
            try:
                dq, dkv = torch.empty_like(q), torch.empty_like(kv)
                et.attention_bwd_ex(q, kv[:, :d], kv[:, d:], do, dq, dkv[:, :d], dkv[:, d:], B, Tq, Tk, H, causal=causal, o=o, lse=lse)
            finally:
                ops.ATTN_F16X2 = True
            outs.append((dq, dkv[:, :d].clone(), dkv[:, d:].clone()))
        for a, b, ref, name in zip(outs[0], outs[1], outs[2], ("dq", "dk", "dv")):
            assert torch.equal(a, b), name
            for hh in range(H):
                sl = slice(64 * hh, 64 * hh + 64)
                assert _rel(a[:, sl], ref[:, sl]) < 5e-6, (name, hh, _rel(a[:, sl], ref[:, sl]))


@pytest.mark.parametrize("B,Tq,Tk,H,causal", [(2, 37, 37, 2, True), (2, 5, 1500, 2, False), (1, 70, 200, 1, False), (3, 129, 129, 1, True),
                                              (2, 64, 64, 2, True), (1, 132, 132, 1, True), (2, 8, 1500, 2, False), (1, 72, 200, 3, False)])
def test_fused_attention_backward_causal_and_cross_shapes(B, Tq, Tk, H, causal):
    """la_attention_bwd_f32 on the text decoder's shapes: causal self-attention and cross-attention (q_len != kv_len, ragged last
    tiles, operands as column slices of packed projections) against torch autograd."""
    from lyricalignment_amd import encoder_train as et
    d = 64 * H
    g = torch.Generator().manual_seed(Tq * 7 + Tk)
    q0 = (torch.randn(B * Tq, d, generator=g) * 0.35).requires_grad_(True)
    kv0 = torch.randn(B * Tk, 2 * d, generator=g).requires_grad_(True)
    do = torch.randn(B * Tq, d, generator=g)
    q = q0.view(B, Tq, H, 64).permute(0, 2, 1, 3)
    k, v = [t.view(B, Tk, H, 64).permute(0, 2, 1, 3) for t in kv0.split(d, dim=1)]
    s = q @ k.transpose(-1, -2)
    if causal:
        s = s + torch.full((Tq, Tk), float("-inf")).triu(1)
    o = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * Tq, d)
    o.backward(do)
    qd, kvd = q0.detach().cuda(), kv0.detach().cuda()
    dq = torch.empty_like(qd)
    dkv = torch.empty_like(kvd)
    et.attention_bwd_ex(qd, kvd[:, :d], kvd[:, d:], do.cuda(), dq, dkv[:, :d], dkv[:, d:], B, Tq, Tk, H, causal=causal, o=o.detach().cuda())
    assert _rel(dq.cpu(), q0.grad) < 2e-4
    assert _rel(dkv.cpu()[:, :d], kv0.grad[:, :d]) < 2e-4 and _rel(dkv.cpu()[:, d:], kv0.grad[:, d:]) < 2e-4


@pytest.mark.parametrize("d,H,L,B", [(64, 1, 1, 1), (128, 2, 2, 2), (512, 8, 1, 8)])
def test_encoder_backward_matches_torch_autograd(d, H, L, B, monkeypatch):
    """EncoderFunction (HIP forward + backward) against torch autograd through the oracle's AudioEncoder restatement:
    output and the gradient of every encoder parameter.  The last case (12000 rows of width 512) is large enough for the Linear
    products to take the f16x2 path (f32x2: three f16 products over split operands, split-K slots for the weight gradients) --
    asserted -- under the same tolerance as the float32 kernels."""
    from oracle import model_oracle as mo
    from lyricalignment_amd import encoder_train as et, f32x2
    x2_calls = []
    real_gemm = f32x2.gemm
    monkeypatch.setattr(f32x2, "gemm", lambda a, w, *args, **kw: (x2_calls.append((a.rows, w.rows, a.kp)), real_gemm(a, w, *args, **kw))[1])
    p = mo.random_encoder_params(d, L, seed=d + L)
    for i in range(L):                      # sharper attention than the 0.02-std init gives, so softmax' is exercised
        p[f"encoder.blocks.{i}.attn.query.weight"] *= 12
        p[f"encoder.blocks.{i}.attn.key.weight"] *= 12
    names = et.encoder_param_names(L)
    g = torch.Generator().manual_seed(5)
    mel = torch.randn(B, 80, 3000, generator=g) * 0.5
    dy = torch.randn(B, 1500, d, generator=g)
    ref_p = {k: v.clone().requires_grad_(k != "encoder.positional_embedding") for k, v in p.items()}
    y_ref = mo.encoder_forward(ref_p, mel, H)
    y_ref.backward(dy)
    params = [p["encoder." + n].cuda().requires_grad_(True) for n in names]
    y = et.EncoderFunction.apply(mel.cuda(), p["encoder.positional_embedding"].cuda(), H, *params)
    assert _rel(y.detach().cpu(), y_ref.detach()) < 1e-4
    y.backward(dy.cuda())
    worst = {}
    for n, t in zip(names, params):
        worst[n] = _rel(t.grad.cpu(), ref_p["encoder." + n].grad)
    bad = {n: e for n, e in worst.items() if e > 1e-3}
    assert not bad, bad
    assert (len(x2_calls) >= 8) == (d >= 512), x2_calls


@pytest.mark.parametrize("B,n,Ta,L", [(2, 37, 1500, 2), (1, 5, 100, 1)])
def test_decoder_backward_matches_torch_autograd(B, n, Ta, L):
    """DecoderFunction (HIP forward + backward: causal self-attention, cross-attention, tied projection, embeddings)
    against torch autograd through the oracle's TextDecoder restatement: logits, every parameter gradient, d/d(xa)."""
    from oracle import model_oracle as mo
    from lyricalignment_amd import decoder_train as dt, whisper_compat as wc
    d, H, V = 128, 2, 311
    dims = wc.ModelDimensions(n_audio_state=d, n_audio_head=H, n_audio_layer=1, n_text_state=d, n_text_head=H, n_text_layer=L,
                              n_vocab=V, n_text_ctx=64)
    wm = wc.build_model(dims=dims, seed=60 + n, std=0.05, with_decoder=True)
    sd = {k: v.detach().float().clone() for k, v in wm.decoder.state_dict().items()}
    for i in range(L):
        for a in ("attn", "cross_attn"):
            sd[f"blocks.{i}.{a}.query.weight"] *= 6
            sd[f"blocks.{i}.{a}.key.weight"] *= 6
    names = dt.decoder_param_names(L)
    g = torch.Generator().manual_seed(61)
    tokens = torch.randint(0, V, (B, n), generator=g)
    tokens[0, -1] = tokens[0, 0]                                   # a repeated token: embedding rows accumulate
    xa = torch.randn(B, Ta, d, generator=g)
    dlog = torch.randn(B, n, V, generator=g)
    ref_p = {"decoder." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xa_ref = xa.clone().requires_grad_(True)
    ref = mo.decoder_forward(ref_p, tokens, xa_ref, H)
    ref.backward(dlog)
    params = [sd[k].cuda().requires_grad_(True) for k in names]
    xa_dev = xa.cuda().requires_grad_(True)
    out = dt.DecoderFunction.apply(tokens.cuda(), xa_dev, H, *params)
    assert _rel(out.detach().cpu(), ref.detach()) < 1e-4
    out.backward(dlog.cuda())
    bad = {}
    for k, t in zip(names, params):
        e = _rel(t.grad.cpu(), ref_p["decoder." + k].grad)
        if e > 1e-3:
            bad[k] = e
    e = _rel(xa_dev.grad.cpu(), xa_ref.grad)
    if e > 1e-3:
        bad["xa"] = e
    assert not bad, bad


def test_cross_entropy_matches_torch():
    from lyricalignment_amd import decoder_train as dt
    g = torch.Generator().manual_seed(70)
    logits = torch.randn(3, 29, 5187, generator=g) * 3
    tgt = torch.randint(0, 5187, (3, 29), generator=g)
    tgt[0, 20:] = -100; tgt[2, :4] = -100
    ref_in = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in.permute(0, 2, 1), tgt)           # train_multitask.py:285
    (ref * 0.125).backward()
    loss, dl = dt.cross_entropy(logits.cuda(), tgt.cuda(), scale_grad=0.125)
    np.testing.assert_allclose(float(loss), float(ref.detach()), rtol=2e-6)
    np.testing.assert_allclose(dl.cpu().numpy(), ref_in.grad.numpy(), rtol=0, atol=1e-8)


def _tiny_full_model(dropout=0.0, seed=90):
    from lyricalignment_amd import whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    dims = wc.ModelDimensions(n_audio_state=128, n_audio_head=2, n_audio_layer=1, n_text_state=128, n_text_head=2, n_text_layer=1,
                              n_vocab=311, n_text_ctx=64)
    wm = wc.build_model(dims=dims, seed=seed, std=0.05, with_decoder=True)
    return AlignModel(wm, embed_dim=128, hidden_dim=64, output_dim=41, dropout=dropout, train_transcript=True, device="cuda").to("cuda")


def _tiny_batch():
    rs = np.random.RandomState(91)
    audios = [(rs.randn(16000) * 0.1).astype(np.float32), (rs.randn(12000) * 0.1).astype(np.float32)]
    labels = torch.tensor([[3, 7, 7, 12], [5, 9, -100, -100]])
    frame_labels = torch.full((2, 50), -100, dtype=torch.long); frame_labels[0, 5:20] = 3; frame_labels[1, 10:30] = 9
    dec_in = torch.tensor([[1, 20, 33, 47, 200], [1, 90, 91, 2, 2]])
    dec_out = torch.tensor([[20, 33, 47, 200, 2], [90, 91, 2, -100, -100]])
    return audios, labels, frame_labels, dec_in, dec_out


def test_full_finetune_micro_step_gradients_match_torch_autograd():
    """The whole config-3 micro-step -- log-mel, encoder, BiGRU head, decoder, CE + silence BCE + CTC + decoder CE, backward
    through every kernel, flat gradient buckets -- against torch autograd through the oracle restatement of the same model
    with the reference's loss formulas (train_multitask.py:587-633, 285)."""
    from oracle import model_oracle as mo
    from lyricalignment_amd import finetune as ft
    F = torch.nn.functional
    model = _tiny_full_model(dropout=0.0)
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    audios, labels, frame_labels, dec_in, dec_out = _tiny_batch()
    tuner = ft.FineTuner(model, vocab_size=40, world=1)
    losses = tuner.micro_step(audios, labels, frame_labels, dec_in, dec_out, accum_grad_steps=2, get_orig_len=False).cpu()
    # ---- reference: the same model as torch functional code on the CPU ----
    p = {}
    for k, v in sd.items():
        key = k[len("whisper_model."):] if k.startswith("whisper_model.") else k
        p[key] = v.double().requires_grad_("encoder.positional_embedding" not in k)     # float64: the truth both sides approximate
    batch = np.zeros((2, 16000), dtype=np.float32); batch[0] = audios[0]; batch[1, :12000] = audios[1]
    mel = mo.pad_or_trim(mo.log_mel_spectrogram(batch), 3000).double()
    xa = mo.encoder_forward(p, mel, n_head=2)
    logits = mo.gru_head_forward(p, xa)
    assert logits.dtype == torch.float64
    ce = mo.ce_loss(logits, frame_labels, vocab_size=40)
    lsm = F.log_softmax(logits[:, :, :40], dim=2).transpose(0, 1)
    ctc = F.ctc_loss(lsm, labels, torch.full((2,), 1500, dtype=torch.long), (labels != -100).sum(1))
    tr = F.cross_entropy(mo.decoder_forward(p, dec_in, xa, n_head=2).permute(0, 2, 1), dec_out)
    ((ce + ctc + tr) / 2).backward()
    np.testing.assert_allclose(float(losses[0] + losses[1]), float(ce.detach()), rtol=2e-4)
    np.testing.assert_allclose(float(losses[2]), float(ctc.detach()), rtol=2e-4)
    np.testing.assert_allclose(float(losses[3]), float(tr.detach()), rtol=2e-4)
    for bucket, params, prefix in ((tuner.grad[0], model.align_rnn.named_parameters(), "align_rnn."),
                                   (tuner.grad[1], model.whisper_model.named_parameters(), "")):
        got = bucket.cpu()
        off = 0
        worst = {}
        for name, prm in params:
            if not prm.requires_grad:
                continue
            g = got[off: off + prm.numel()].view(prm.shape); off += prm.numel()
            ref = p[prefix + name].grad
            worst[name] = float((g - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
        assert off == got.numel()
        bad = {k: v for k, v in worst.items() if v > 2e-3}
        assert not bad, bad


@pytest.mark.parametrize("use_ctc", [True, False])
def test_whole_micro_step_with_both_sub_batches_matches_float64_torch(use_ctc):
    """train_step's micro-step as the reference composes it (train_multitask.py:241-326): the MULTITASK sub-batch (clips with
    frame labels: compute_ce_loss -- with compute_sil = use_ctc_loss, i.e. the plain all-column cross-entropy of :603-605 in
    the non-CTC configuration -- [+ compute_ctc_loss] + decoder CE) AND the TRANSCRIPT-ONLY sub-batch (second
    frame_manual_forward: decoder CE [+ CTC on its align logits]), summed, divided by accum_grad_steps, one backward each.
    Every loss term and every parameter gradient in the flat buckets against float64 torch autograd through the oracle."""
    from oracle import model_oracle as mo
    from lyricalignment_amd import finetune as ft
    from lyricalignment_amd import whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    F = torch.nn.functional
    V = 40
    dims = wc.ModelDimensions(n_audio_state=128, n_audio_head=2, n_audio_layer=1, n_text_state=128, n_text_head=2, n_text_layer=1,
                              n_vocab=311, n_text_ctx=64)
    wm = wc.build_model(dims=dims, seed=190, std=0.05, with_decoder=True)
    model = AlignModel(wm, embed_dim=128, hidden_dim=64, output_dim=V + int(use_ctc), dropout=0.0, train_transcript=True, device="cuda").to("cuda")
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    audios, labels, frame_labels, dec_in, dec_out = _tiny_batch()
    rs = np.random.RandomState(191)
    t_audios = [(rs.randn(14000) * 0.1).astype(np.float32)]
    t_labels = torch.tensor([[4, 4, 30]])
    t_in, t_out = torch.tensor([[1, 7, 8, 9]]), torch.tensor([[7, 8, 9, 2]])
    tuner = ft.FineTuner(model, vocab_size=V, use_ctc_loss=use_ctc, world=1)
    losses = tuner.micro_step(audios, labels, frame_labels, dec_in, dec_out, accum_grad_steps=2, get_orig_len=False,
                              transcript_batch=(t_audios, t_labels, t_in, t_out)).cpu()
    # ---- reference: the same two forward passes as float64 torch on the CPU ----
    p = {}
    for k, v in sd.items():
        key = k[len("whisper_model."):] if k.startswith("whisper_model.") else k
        p[key] = v.double().requires_grad_("encoder.positional_embedding" not in k)

    def forward(auds, y_in):
        n = max(map(len, auds))
        batch = np.zeros((len(auds), n), dtype=np.float32)
        for i, a in enumerate(auds):
            batch[i, : len(a)] = a
        xa = mo.encoder_forward(p, mo.pad_or_trim(mo.log_mel_spectrogram(batch), 3000).double(), n_head=2)
        return mo.gru_head_forward(p, xa), mo.decoder_forward(p, y_in, xa, n_head=2)

    def ctc(logits, lab):
        lsm = F.log_softmax(logits[:, :, :V], dim=2).transpose(0, 1)
        return F.ctc_loss(lsm, lab, torch.full((lab.shape[0],), logits.shape[1], dtype=torch.long), (lab != -100).sum(1))

    al, tr = forward(audios, dec_in)
    fl = ft.pad_frame_labels(frame_labels, al.shape[1])
    if use_ctc:
        ce = mo.ce_loss(al, frame_labels.clone(), vocab_size=V)
        align_ctc = ctc(al, labels)
    else:
        ce = F.cross_entropy(al.permute(0, 2, 1), fl)            # compute_sil == False (:603-605)
        align_ctc = torch.zeros((), dtype=torch.float64)
    tr_ce = F.cross_entropy(tr.permute(0, 2, 1), dec_out)
    al2, tr2 = forward(t_audios, t_in)
    t_ce = F.cross_entropy(tr2.permute(0, 2, 1), t_out)
    t_ctc = ctc(al2, t_labels) if use_ctc else torch.zeros((), dtype=torch.float64)
    ((ce + align_ctc + tr_ce + t_ce + t_ctc) / 2).backward()
    np.testing.assert_allclose(float(losses[0] + losses[1]), float(ce.detach()), rtol=2e-4)
    np.testing.assert_allclose(float(losses[2]), float((align_ctc + t_ctc).detach()), rtol=2e-4, atol=1e-12)
    np.testing.assert_allclose(float(losses[3]), float((tr_ce + t_ce).detach()), rtol=2e-4)
    for bucket, params, prefix in ((tuner.grad[0], model.align_rnn.named_parameters(), "align_rnn."),
                                   (tuner.grad[1], model.whisper_model.named_parameters(), "")):
        got = bucket.cpu()
        off = 0
        bad = {}
        for name, prm in params:
            if not prm.requires_grad:
                continue
            g = got[off: off + prm.numel()].view(prm.shape); off += prm.numel()
            ref = p[prefix + name].grad
            err = float((g - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
            if err > 2e-3:
                bad[name] = err
        assert off == got.numel() and not bad, bad


def test_medium_width_micro_step_f16x2_route_against_float32_route_and_float64():
    """tools/ft_grad_compare.py as a test (round-5 verdict): the whole micro-step at Whisper-medium WIDTH (d = 1024, 16 heads; 2 encoder blocks,
    1 decoder block, 3 clips x 1500 frames, dropout 0) on the f16x2 route (every large Linear, the attention sweeps and the GRU sweeps as three f16
    products) and on the float32-MFMA route, both against float64 torch autograd through the oracle: every parameter gradient of both routes
    within 2e-4 of float64 (relative to the parameter's largest gradient entry; measured 3e-6 / 1.5e-5), the f16x2 route's worst error not above 1.5 x the float32
    route's, the f16x2 route actually taken (launch counts of its kernel families), and two runs of it bit-identical."""
    import ctypes
    from oracle import model_oracle as mo
    from lyricalignment_amd import _lib, f32x2, finetune as ft, ops, whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    F = torch.nn.functional
    dims = wc.ModelDimensions(n_audio_state=1024, n_audio_head=16, n_audio_layer=2, n_text_state=1024, n_text_head=16, n_text_layer=1, n_vocab=1000, n_text_ctx=64)
    V = 500
    wm = wc.build_model(dims=dims, seed=21, std=0.02, with_decoder=True)
    model = AlignModel(wm, embed_dim=1024, hidden_dim=128, output_dim=V + 1, dropout=0.0, train_transcript=True, device="cuda").to("cuda")
    wc.init_align_head(model, seed=7, fc_scale=1.0, rnn_scale=1.0)
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    rs = np.random.RandomState(5)
    B = 3
    audios = [(rs.randn(24000 + 800 * i) * 0.1).astype(np.float32) for i in range(B)]
    labels = torch.from_numpy(rs.randint(1, V - 1, size=(B, 6)))
    labels[1, 4:] = -100
    fl = torch.full((B, 1500), -100, dtype=torch.long)
    for b in range(B):
        for i in range(6):
            if labels[b, i] != -100:
                fl[b, 30 + 200 * i: 150 + 200 * i] = labels[b, i]
    dec_in = torch.from_numpy(rs.randint(1, 999, size=(B, 8)))
    dec_out = torch.from_numpy(rs.randint(1, 999, size=(B, 8)))
    dec_out[2, 6:] = -100
    tuner = ft.FineTuner(model, vocab_size=V, world=1)
    L = _lib.lib()

    def run(x2: bool, count=None):
        f32x2.ENABLED, ops.ATTN_F16X2 = x2, x2
        for g in tuner.grad:
            g.zero_()
        if count:
            L.la_timer_reset(); L.la_timer_sample(1000003); L.la_timer_enable(count.encode())
        with _lib.option("gru_handoff", 0 if x2 else 1):          # 1 = the float32-MFMA GRU training sweeps
            losses = tuner.micro_step(audios, labels, fl, dec_in, dec_out, accum_grad_steps=1, get_orig_len=False).cpu().double()
        torch.cuda.synchronize()
        n = 0
        if count:
            L.la_timer_disable()
            ms, timed, work, seen = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_int64(0)
            L.la_timer_read_work(ctypes.byref(ms), ctypes.byref(timed), ctypes.byref(work), ctypes.byref(seen))
            n = int(seen.value)
            L.la_timer_reset(); L.la_timer_sample(1)
        return [g.clone() for g in tuner.grad], losses, n

    try:
        ga, la, n_gemm = run(True, "gemm_f16x2")
        gc, lc, n_attn = run(True, "attention_bwd_f16x2")
        gb, lb, n_off = run(False, "gemm_f16x2")
    finally:
        f32x2.ENABLED, ops.ATTN_F16X2 = True, True
    # forward + two backward products of 4 Linear layers per encoder block (+ the decoder's, the head's): the route was taken / not taken
    assert n_gemm >= 3 * 4 * 2 and n_attn >= 2 and n_off == 0, (n_gemm, n_attn, n_off)
    for a, c in zip(ga, gc):
        assert torch.equal(a, c)                                   # run to run: the same bits
    assert torch.equal(la, lc)
    # ---- float64 truth: the oracle restatement with the reference's loss formulas ----
    p = {}
    for k, v in sd.items():
        key = k[len("whisper_model."):] if k.startswith("whisper_model.") else k
        p[key] = v.double().requires_grad_("positional_embedding" not in k or "decoder" in k)
    n = max(len(a) for a in audios)
    batch = np.zeros((B, n), dtype=np.float32)
    for i, a in enumerate(audios):
        batch[i, : len(a)] = a
    xa = mo.encoder_forward(p, mo.pad_or_trim(mo.log_mel_spectrogram(batch), 3000).double(), n_head=16)
    logits = mo.gru_head_forward(p, xa)
    ce = mo.ce_loss(logits, fl, vocab_size=V)
    lsm = F.log_softmax(logits[:, :, :V], dim=2).transpose(0, 1)
    ctc = F.ctc_loss(lsm, labels, torch.full((B,), 1500, dtype=torch.long), (labels != -100).sum(1))
    tr = F.cross_entropy(mo.decoder_forward(p, dec_in, xa, n_head=16).permute(0, 2, 1), dec_out)
    (ce + ctc + tr).backward()
    for ls in (la, lb):
        np.testing.assert_allclose([float(ls[0] + ls[1]), float(ls[2]), float(ls[3])], [float(ce.detach()), float(ctc.detach()), float(tr.detach())], rtol=2e-4)
    worst = {}
    for route, grads in (("f16x2", ga), ("float32", gb)):
        w = {}
        for bucket, params, prefix in ((grads[0], model.align_rnn.named_parameters(), "align_rnn."), (grads[1], model.whisper_model.named_parameters(), "")):
            got, off = bucket.cpu().double(), 0
            for name, prm in params:
                if not prm.requires_grad:
                    continue
                g = got[off: off + prm.numel()].view(prm.shape); off += prm.numel()
                ref = p[prefix + name].grad
                w[name] = float((g - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
            assert off == got.numel()
        worst[route] = w
        bad = {k: v for k, v in w.items() if v > 2e-4}
        assert not bad, (route, bad)
    wa, wb = max(worst["f16x2"].values()), max(worst["float32"].values())
    print(f"medium width, {B} clips: worst parameter-gradient error against float64 -- f16x2 route {wa:.2e}, float32-MFMA route {wb:.2e}; "
          f"buckets f16x2 vs float32: " + ", ".join(f"{float((a.double() - b.double()).norm() / b.double().norm()):.2e}" for a, b in zip(ga, gb)))
    assert wa <= 1.5 * wb + 1e-5


def test_finetuner_leaves_parameters_without_gradient_alone():
    """torch.optim.AdamW skips parameters whose .grad is None -- no update and no weight decay.  A model built with a decoder
    but train_transcript=False never produces decoder gradients: its parameters stay out of the flat buckets and are
    bit-identical after optimizer steps, while head and encoder move."""
    from lyricalignment_amd import finetune as ft
    from lyricalignment_amd import whisper_compat as wc
    from lyricalignment_amd.module.align_model import AlignModel
    dims = wc.ModelDimensions(n_audio_state=128, n_audio_head=2, n_audio_layer=1, n_text_state=128, n_text_head=2, n_text_layer=1,
                              n_vocab=311, n_text_ctx=64)
    wm = wc.build_model(dims=dims, seed=195, std=0.05, with_decoder=True)
    model = AlignModel(wm, embed_dim=128, hidden_dim=64, output_dim=41, dropout=0.0, train_transcript=False, device="cuda").to("cuda")
    dec_before = {k: v.detach().clone() for k, v in model.whisper_model.decoder.state_dict().items()}
    enc_before = model.whisper_model.encoder.conv1.weight.detach().clone()
    audios, labels, frame_labels, dec_in, dec_out = _tiny_batch()
    tuner = ft.FineTuner(model, lr=5e-3, backbone_lr=1e-3, weight_decay=0.1, vocab_size=40, world=1)
    n_dec = sum(p.numel() for p in model.whisper_model.decoder.parameters())
    n_enc = sum(p.numel() for p in model.whisper_model.encoder.parameters() if p.requires_grad)
    assert tuner.flat[1].numel() == n_enc and n_dec > 0
    for _ in range(2):
        tuner.micro_step(audios, labels, frame_labels, dec_in, dec_out, accum_grad_steps=1)
        tuner.step()
    for k, v in model.whisper_model.decoder.state_dict().items():
        assert torch.equal(v, dec_before[k]), k
    assert not torch.equal(model.whisper_model.encoder.conv1.weight, enc_before)


def test_full_finetune_steps_reduce_the_loss():
    """FineTuner: a few optimizer steps (2 micro-batches each) of whole-model fine-tuning on a fixed tiny batch; the
    parameters are views of the flat buckets, so the fused AdamW updates the module in place and eval re-packs."""
    from lyricalignment_amd import finetune as ft
    model = _tiny_full_model(dropout=0.15)
    audios, labels, frame_labels, dec_in, dec_out = _tiny_batch()
    tuner = ft.FineTuner(model, lr=5e-3, backbone_lr=2e-4, warmup_steps=1, train_steps=20, vocab_size=40, world=1)
    totals = []
    for it in range(5):
        acc = torch.zeros(4)
        for _ in range(2):
            acc += tuner.micro_step(audios, labels, frame_labels, dec_in, dec_out, accum_grad_steps=2).cpu() / 2
        sumsq = tuner.step()
        assert float(sumsq) > 0 and torch.isfinite(acc).all()
        totals.append(float(acc.sum()))
    assert totals[-1] < totals[1], totals            # step 0 runs at lr 0 (warm-up), like LambdaLR
    model.eval()
    with torch.no_grad():
        logits, _ = model.frame_manual_forward(audios, get_orig_len=True)
    assert tuple(logits.shape) == (2, 50, 41) and torch.isfinite(logits).all()


def test_fused_gradient_accumulation_equals_the_loop_of_micro_steps():
    """FineTuner.accumulate(fused=True): the accumulation loop of train_step (train_multitask.py:240-326) as ONE forward /
    backward over all micro-batches' clips, every loss taken per micro-batch slice with its own means and 1 / accum factor.
    Three micro-batches of different sizes, label counts and decoder lengths (dropout 0: the masks are the only thing that
    may differ): the accumulated gradient buckets and the loss vector equal the loop of micro_step() calls."""
    from lyricalignment_amd import finetune as ft
    audios, labels, frame_labels, dec_in, dec_out = _tiny_batch()
    rs = np.random.RandomState(92)
    a3 = (rs.randn(14000) * 0.1).astype(np.float32)
    fl3 = torch.full((1, 50), -100, dtype=torch.long); fl3[0, 2:40] = 11
    mbs = [
        dict(audios=audios, ctc_labels=labels, frame_labels=frame_labels, decoder_input=dec_in, decoder_output=dec_out),
        dict(audios=[a3], ctc_labels=torch.tensor([[11, 4, 30]]), frame_labels=fl3, decoder_input=torch.tensor([[1, 55, 2]]),
             decoder_output=torch.tensor([[55, 2, -100]])),
        dict(audios=audios[::-1], ctc_labels=labels.flip(0), frame_labels=frame_labels.flip(0), decoder_input=dec_in.flip(0),
             decoder_output=dec_out.flip(0)),
    ]
    grads, losses = [], []
    for fused in (False, True):
        model = _tiny_full_model(dropout=0.0)
        torch.manual_seed(3)
        with torch.no_grad():                                   # identical head in both runs (it is drawn from the global generator)
            for p_ in model.align_rnn.parameters():
                p_.copy_(torch.randn(p_.shape) * 0.1)
        tuner = ft.FineTuner(model, vocab_size=40, world=1)
        losses.append(tuner.accumulate(mbs, fused=fused).cpu())
        grads.append([g.clone().cpu() for g in tuner.grad])
    # (float32 forward with other GEMM shapes -- other tile / split-K decisions, other summation orders: 1e-4 relative)
    np.testing.assert_allclose(losses[1].numpy(), losses[0].numpy(), rtol=3e-4, atol=1e-6)
    for g_loop, g_fused in zip(*grads):
        scale = float(g_loop.abs().max())
        assert scale > 0
        np.testing.assert_allclose(g_fused.numpy(), g_loop.numpy(), rtol=0, atol=3e-4 * scale)


def test_fused_accumulation_with_heterogeneous_micro_batches_takes_the_loop():
    """Round-3 advisor finding: accumulate(fused=True) used to drop the decoder loss for ALL micro-batches as soon as one of them came
    without a decoder pair (and ignored transcript_batch).  Heterogeneous micro-batches now take the loop of micro_step() calls: the
    decoder CE of the micro-batch that has a pair is in the loss vector and in the gradients, exactly as accumulate(fused=False)."""
    from lyricalignment_amd import finetune as ft
    audios, labels, frame_labels, dec_in, dec_out = _tiny_batch()
    mbs = [dict(audios=audios, ctc_labels=labels, frame_labels=frame_labels, decoder_input=dec_in, decoder_output=dec_out),
           dict(audios=audios[::-1], ctc_labels=labels.flip(0), frame_labels=frame_labels.flip(0))]           # no decoder pair
    out = []
    for fused in (False, True):
        model = _tiny_full_model(dropout=0.0)
        torch.manual_seed(3)
        with torch.no_grad():
            for p_ in model.align_rnn.parameters():
                p_.copy_(torch.randn(p_.shape) * 0.1)
        tuner = ft.FineTuner(model, vocab_size=40, world=1)
        l = tuner.accumulate(mbs, fused=fused).cpu()
        out.append((l, [g.clone().cpu() for g in tuner.grad]))
    assert float(out[0][0][3]) > 0                                   # the decoder CE of the first micro-batch is there
    assert torch.equal(out[0][0], out[1][0])                         # same path -> same bits
    for a, b in zip(out[0][1], out[1][1]):
        assert torch.equal(a, b)


def test_attention_backward_refuses_a_causal_mask_with_unequal_lengths():
    """Round-3 advisor finding: la_attention_bwd_f32 accepted causal != 0 with q_len != kv_len although its mask has no kv_len - q_len
    offset (the forward entry points reject that combination): now LA_EINVAL."""
    from lyricalignment_amd import encoder_train as et
    B, Tq, Tk, H = 1, 8, 12, 1
    q = torch.randn(B * Tq, 64, device="cuda"); kv = torch.randn(B * Tk, 128, device="cuda"); do = torch.randn(B * Tq, 64, device="cuda")
    dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    with pytest.raises(ValueError):
        et.attention_bwd_ex(q, kv[:, :64], kv[:, 64:], do, dq, dkv[:, :64], dkv[:, 64:], B, Tq, Tk, H, causal=True, o=torch.randn_like(q))


def test_out_of_vocabulary_token_ids_are_clamped_not_faulted():
    """A decoder prompt with ids outside [0, n_vocab) (bad user data; the reference would raise from nn.Embedding) must not become
    an out-of-bounds gather / atomic on the device: la_embed_tokens and la_embed_tokens_bwd_f32 clamp the id the same way."""
    from lyricalignment_amd import finetune as ft
    model = _tiny_full_model(dropout=0.0)
    audios, labels, frame_labels, dec_in, dec_out = _tiny_batch()
    bad = dec_in.clone(); bad[0, 2] = 10 ** 6; bad[1, 1] = -7
    tuner = ft.FineTuner(model, vocab_size=40, world=1)
    losses = tuner.micro_step(audios, labels, frame_labels, bad, dec_out, accum_grad_steps=1)
    torch.cuda.synchronize()
    assert torch.isfinite(losses).all() and all(torch.isfinite(g).all() for g in tuner.grad)


def _dp_worker(rank, world, port, out_dir, chunks=4):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)      # RCCL on a multi-GPU node; gloo here (one GPU)
    from lyricalignment_amd import finetune as ft
    torch.manual_seed(95)                                                      # the head's torch-default initialisation
    model = _tiny_full_model(dropout=0.0, seed=95)
    audios, labels, frame_labels, dec_in, dec_out = _tiny_batch()
    if rank == 1:                                                              # a replica that starts elsewhere: the constructor's
        with torch.no_grad():                                                  # broadcast must bring it to rank 0's parameters
            for p in model.align_rnn.parameters():
                p.add_(0.01)
    tuner = ft.FineTuner(model, lr=5e-3, backbone_lr=2e-4, vocab_size=40, allreduce_chunks=chunks)      # world from the process group
    assert tuner.world == world
    sl = slice(rank, rank + 1)                                                   # one clip per rank
    # two optimizer steps: the second sees the first's update (and, with chunks, hooks that must have disarmed and re-armed)
    for _ in range(2):
        tuner.micro_step([audios[rank]], labels[sl], frame_labels[sl], dec_in[sl], dec_out[sl], accum_grad_steps=1, last=True)
        if chunks:
            assert len(tuner.overlap.chunks) >= 2 and any(tuner.overlap._launched)    # chunks left DURING the backward
        tuner.step()
        assert tuner.allreduce_exposed_ms >= 0.0
    torch.save([f.cpu() for f in tuner.flat], os.path.join(out_dir, f"rank{rank}_c{chunks}.pt"))
    dist.destroy_process_group()


def test_data_parallel_step_equals_accumulated_single_process(tmp_path):
    """The data-parallel fine-tune step (one flat gradient all-reduce per bucket before the clip, mean folded into the
    update) on 2 ranks with one clip each == one process accumulating the same two clips (each scaled by 1/2): identical
    parameters on both ranks, and equal to the single-process result."""
    import torch.multiprocessing as mp
    from lyricalignment_amd import finetune as ft
    port = 29600 + (os.getpid() % 200)
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path), 4), nprocs=2, join=True)       # gradient chunks all-reduced during the backward
    mp.spawn(_dp_worker, args=(2, port + 1, str(tmp_path), 0), nprocs=2, join=True)   # one blocking all-reduce per bucket after it
    r0 = torch.load(tmp_path / "rank0_c4.pt"); r1 = torch.load(tmp_path / "rank1_c4.pt")
    b0 = torch.load(tmp_path / "rank0_c0.pt")
    for a, b, c in zip(r0, r1, b0):
        assert torch.equal(a, b)
        assert torch.equal(a, c)          # 2 ranks: a + b whatever the chunking -- bit-equal to the single-bucket path
    torch.manual_seed(95)
    model = _tiny_full_model(dropout=0.0, seed=95)
    audios, labels, frame_labels, dec_in, dec_out = _tiny_batch()
    tuner = ft.FineTuner(model, lr=5e-3, backbone_lr=2e-4, vocab_size=40, world=1)
    for _ in range(2):
        for r in range(2):
            sl = slice(r, r + 1)
            tuner.micro_step([audios[r]], labels[sl], frame_labels[sl], dec_in[sl], dec_out[sl], accum_grad_steps=2)
        tuner.step()
    for a, b in zip(r0, tuner.flat):
        np.testing.assert_allclose(a.numpy(), b.cpu().numpy(), rtol=0, atol=4e-6)
