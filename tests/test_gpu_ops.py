"""GPU numerics of the encoder / head building blocks through the C ABI, against plain
torch fp32 references of the same op (floating-point kernels: tolerance stated per test)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def attn_nw():
    """Set the library option attn_nw (attention workgroup form) inside a test; restored afterwards."""
    from lyricalignment_amd import _lib
    prev = _lib.get_option("attn_nw")
    yield lambda v: _lib.set_option("attn_nw", int(v))
    _lib.set_option("attn_nw", prev)


@pytest.fixture
def gemm_loop():
    """Set the library option gemm_loop (0 = hand-placed loop where it fits, 99 = quadrant ping-pong everywhere); restored afterwards."""
    from lyricalignment_amd import _lib
    prev = _lib.get_option("gemm_loop")
    yield lambda v: _lib.set_option("gemm_loop", int(v))
    _lib.set_option("gemm_loop", prev)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 128), (1500, 1152, 384), (257, 21129 // 8, 768), (48, 36, 1024)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_plain(M, N, K, dtype):
    from lyricalignment_amd import ops
    a = _rand(M, K, seed=1, scale=0.5); w = _rand(N, K, seed=2, scale=0.5)
    ad, wd = a.to(dtype).cuda(), w.to(dtype).cuda()
    out = ops.gemm(ad, wd, out_f32=True)
    ref = ad.float().cpu().double() @ wd.float().cpu().double().T
    tol = 2e-4 if dtype == torch.float32 else 2e-3  # f32: fmaf-chain rounding over K; bf16: exact products, f32 accumulate
    np.testing.assert_allclose(out.cpu().double().numpy(), ref.numpy(), rtol=0, atol=tol * (K / 64) ** 0.5)


def test_gemm_asymmetric_identity():
    """A = I with an asymmetric W catches a transposed C write (cdna guide, 'Always A=I-check')."""
    from lyricalignment_amd import ops
    K = 128
    a = torch.eye(K)
    w = torch.arange(200 * K, dtype=torch.float32).reshape(200, K) % 251
    out = ops.gemm(a.cuda(), w.cuda())
    assert torch.equal(out.cpu(), w.T.contiguous())
    outb = ops.gemm(a.bfloat16().cuda(), w.bfloat16().cuda(), out_f32=True)
    assert torch.equal(outb.cpu(), w.bfloat16().float().T.contiguous())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_epilogues(dtype):
    from lyricalignment_amd import ops
    M, N, K = 200, 136, 192
    a = _rand(M, K, seed=3, scale=0.3).to(dtype); w = _rand(N, K, seed=4, scale=0.3).to(dtype)
    bias = _rand(N, seed=5); res = _rand(M, N, seed=6)
    base = a.float() @ w.float().T + bias
    tol = 1e-4 if dtype == torch.float32 else 1e-3
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), gelu=True, out_f32=True).cpu()
    np.testing.assert_allclose(out.numpy(), torch.nn.functional.gelu(base).numpy(), rtol=0, atol=tol)
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), out_f32=True).cpu()
    np.testing.assert_allclose(out.numpy(), (base + res).numpy(), rtol=0, atol=tol)
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), mish=True, out_f32=True).cpu()
    np.testing.assert_allclose(out.numpy(), torch.nn.functional.mish(base).numpy(), rtol=0, atol=tol)
    if dtype != torch.float32:
        out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda()).cpu()
        assert out.dtype == dtype
        np.testing.assert_allclose(out.float().numpy(), base.numpy(), rtol=1e-2, atol=1e-2)
        out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), gelu=True).cpu()          # packed-FP32 GELU, 16-bit out
        np.testing.assert_allclose(out.float().numpy(), torch.nn.functional.gelu(base).numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("M,N,K", [(80, 1024, 4096), (80, 1000, 1024), (129, 260, 3072), (80, 1024, 1632), (3000, 1024, 1024), (80, 512, 51872)])
def test_gemm_f32_split_k_small_grids(M, N, K):
    """float32 GEMMs with few output tiles and a long K (the text decoder's 80-row GEMMs in the fine-tune step) run as S
    K-chunks + a fixed-order reduction that applies the epilogue: same results as the one-pass kernel within f32 rounding,
    run-to-run identical (no atomics).  Also: uneven chunks (K = 51 and 1621 k-steps) and the 192-tile case that is split
    4 ways to fill its last round of tiles."""
    from lyricalignment_amd import ops
    a = _rand(M, K, seed=13, scale=0.3); w = _rand(N, K, seed=14, scale=0.3)
    bias = _rand(N, seed=15); res = _rand(M, N, seed=16)
    base = (a.double() @ w.double().T + bias.double())
    tol = 2e-4 * (K / 64) ** 0.5
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), out_f32=True).cpu()
    np.testing.assert_allclose(out.double().numpy(), base.numpy(), rtol=0, atol=tol)
    again = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), out_f32=True).cpu()
    assert torch.equal(out, again)
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), gelu=True, out_f32=True).cpu()
    np.testing.assert_allclose(out.double().numpy(), torch.nn.functional.gelu(base).numpy(), rtol=0, atol=tol)
    out = ops.gemm(a.cuda(), w.cuda(), bias=bias.cuda(), residual=res.cuda(), out_f32=True).cpu()
    np.testing.assert_allclose(out.double().numpy(), (base + res.double()).numpy(), rtol=0, atol=tol)
    x = res.clone().cuda()                                   # in-place residual (C aliases the residual), as the encoder uses it
    ops.gemm(a.cuda(), w.cuda(), x, bias=bias.cuda(), residual=x, out_f32=True)
    np.testing.assert_allclose(x.cpu().double().numpy(), (base + res.double()).numpy(), rtol=0, atol=tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_as_gemm_views(dtype):
    """Conv1d(k=3, pad=1, stride s) on channels-last rows == GEMM over overlapping row views."""
    from lyricalignment_amd import ops
    B, C, T, Cout = 2, 128, 64, 96
    x = _rand(B, C, T, seed=7, scale=0.5)
    wconv = _rand(Cout, C, 3, seed=8, scale=0.2)
    bias = _rand(Cout, seed=9)
    rows = torch.zeros(B, T + 2, C)
    rows[:, 1:T + 1] = x.permute(0, 2, 1)
    wk = wconv.permute(0, 2, 1).reshape(Cout, 3 * C).contiguous()  # [out][tap][in]
    for stride in (1, 2):
        To = T // stride
        ref = torch.nn.functional.conv1d(x.to(dtype).float(), wconv.to(dtype).float(), bias, stride=stride, padding=1).permute(0, 2, 1)
        out = torch.empty(B, To, Cout, dtype=torch.float32, device="cuda")
        ops.gemm(rows.to(dtype).cuda(), wk.to(dtype).cuda(), out, bias=bias.cuda(), out_f32=True, M=To, lda=stride * C,
                 batch=B, stride_a=(T + 2) * C, stride_c=To * Cout, ldc=Cout)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-4 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("d", [384, 1024, 1280])
def test_layernorm(d):
    from lyricalignment_amd import ops
    x = _rand(77, d, seed=10, scale=3.0) + 0.5
    g = 1 + _rand(d, seed=11, scale=0.1); b = _rand(d, seed=12, scale=0.1)
    ref = torch.nn.functional.layer_norm(x, (d,), g, b, 1e-5)
    out = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), torch.float32).cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=0, atol=2e-5)
    outb = ops.layernorm(x.cuda(), g.cuda(), b.cuda(), torch.bfloat16).cpu()
    np.testing.assert_allclose(outb.float().numpy(), ref.numpy(), rtol=1e-2, atol=1e-2)


def test_mel_to_rows_and_cast():
    from lyricalignment_amd import ops
    mel = _rand(3, 80, 100, seed=13)
    out = ops.mel_to_rows(mel.cuda(), 128, torch.float32).cpu()
    assert out.shape == (3, 102, 128)
    assert torch.equal(out[:, 1:101, :80], mel.permute(0, 2, 1))
    assert out[:, 0].abs().sum() == 0 and out[:, 101].abs().sum() == 0 and out[:, :, 80:].abs().sum() == 0
    outb = ops.mel_to_rows(mel.cuda(), 128, torch.bfloat16).cpu()
    assert torch.equal(outb[:, 1:101, :80], mel.permute(0, 2, 1).bfloat16())
    x = _rand(1001, seed=14)
    assert torch.equal(ops.cast_bf16(x.cuda()).cpu(), x.bfloat16())


@pytest.mark.parametrize("B,T,H", [(1, 5, 1), (2, 64, 2), (1, 200, 3), (2, 1500, 2), (1, 129, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_attention(B, T, H, dtype):
    """softmax(q k^T) v per head, q pre-scaled; fp64 reference on the host over the FULL tensor."""
    from lyricalignment_amd import ops
    d = H * 64
    qkv = _rand(B * T, 3 * d, seed=20 + T, scale=1.0)
    qkv[:, :d] *= 0.125 * 3.0  # pre-scaled q (x3 so the softmax is peaked and the online rescale is exercised)
    x = qkv.to(dtype)
    out = ops.attention(x.cuda(), B, T, H).float().cpu()
    xd = x.double().reshape(B, T, 3, H, 64)
    q, k, v = xd[:, :, 0].transpose(1, 2), xd[:, :, 1].transpose(1, 2), xd[:, :, 2].transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(B * T, d)
    tol = 2e-5 if dtype == torch.float32 else 2e-2  # bf16: P and the output are rounded to bf16 (8-bit mantissa), |v| ~ 1
    np.testing.assert_allclose(out.double().numpy(), ref.numpy(), rtol=0, atol=tol)


@pytest.mark.parametrize("nw", ["4", "8"])
def test_attention_online_rescale_spike(nw, attn_nw):
    """Force the running-max update: one key late in the sequence dominates one query (cdna guide rule 26).  Both workgroup forms
    of the 16-bit kernels (option attn_nw: 128-query / 256-query workgroups; the second is the default from 1024 positions on)."""
    from lyricalignment_amd import ops
    attn_nw(nw)
    T, H = 300, 1
    qkv = _rand(T, 192, seed=31, scale=0.3)
    qkv[7, :64] = 0.0; qkv[7, 0] = 4.0           # query 7 looks at feature 0
    qkv[:, 64] = 0.0; qkv[250, 64] = 10.0         # key 250 spikes on feature 0 (tile 3), key 10 a smaller spike (tile 0)
    qkv[10, 64] = 5.0
    qkv[130, 64] = 7.5                             # and an intermediate one in the FIRST half of tile 2 (both redo paths of the
    qkv[9, :64] = 0.0; qkv[9, 0] = -4.0            # optimistic softmax); query 9 looks the other way: its scores only fall
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 2e-2), (torch.float16, 4e-3)):
        x = qkv.to(dtype)
        out = ops.attention(x.cuda(), 1, T, H).float().cpu()
        xd = x.double()
        q, k, v = xd[:, :64], xd[:, 64:128], xd[:, 128:]
        ref = torch.softmax(q @ k.T, dim=-1) @ v
        np.testing.assert_allclose(out.double().numpy(), ref.numpy(), rtol=0, atol=tol)


@pytest.mark.parametrize("B,T,H", [(1, 5, 1), (2, 64, 2), (1, 200, 3), (2, 1500, 2), (1, 129, 1)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_q_log2(B, T, H, dtype):
    """LA_Q_LOG2: q carries log2(e) / 8, the scores arrive in the exp2 domain (bf16: the kernel that starts its running maximum at
    0 and subtracts nothing until a query needs it; f16: the plain kernel with a scale of 1).  Same softmax as test_attention."""
    from lyricalignment_amd import ops
    d = H * 64
    qkv = _rand(B * T, 3 * d, seed=20 + T, scale=1.0)
    qkv[:, :d] *= 0.125 * 3.0
    x = qkv.to(dtype)
    x2 = x.clone()
    x2[:, :d] = (x[:, :d].double() * 1.4426950408889634).to(dtype)        # what pack_encoder's fold gives the kernel
    out = ops.attention(x2.cuda(), B, T, H, q_log2=True).float().cpu()
    xd = x2.double().reshape(B, T, 3, H, 64)
    q, k, v = xd[:, :, 0].transpose(1, 2) / 1.4426950408889634, xd[:, :, 1].transpose(1, 2), xd[:, :, 2].transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(B * T, d)
    np.testing.assert_allclose(out.double().numpy(), ref.numpy(), rtol=0, atol=2e-2)


@pytest.mark.parametrize("nw", ["4", "8"])
def test_attention_q_log2_extreme_scores(nw, attn_nw):
    """The exp2-domain kernel's two exits from its no-maximum fast path: scores above 2^30 (the redo path installs a maximum and
    the wave subtracts it from then on) and queries whose scores ALL lie below 2^-100 (the sum underflows: the block repeats its
    sweep the textbook way) -- next to ordinary queries in the same block, in the first / a middle / the last tile.  Both
    workgroup forms (LA_ATTN_NW)."""
    from lyricalignment_amd import ops
    attn_nw(nw)
    T, H = 300, 1
    qkv = _rand(T, 192, seed=33, scale=0.3)
    qkv[:, 64] = 0.0
    qkv[10, 64] = 4.0; qkv[130, 64] = 6.0; qkv[250, 64] = 8.0; qkv[299, 64] = 7.0      # keys that answer to feature 0
    qkv[5, 1:64] = 0.0                                                                # (queries 5 / 7 / 9 / 200 look at feature 0 only)
    qkv[7, :64] = 0.0; qkv[7, 0] = 12.0             # scores up to 96 (exp2 domain): above the 2^30 limit from tile 0 on, rising later
    qkv[:, 1] = 0.0; qkv[:, 65] = 5.0               # every key answers 5 on feature 1, which only query 9 looks at:
    qkv[9, :64] = 0.0; qkv[9, 1] = -30.0            # all of its scores are -150: the sum of exp2 underflows -> its block repeats the textbook way
    qkv[200, :64] = 0.0; qkv[200, 0] = 5.0          # overflows only in tile 3 (key 250): a redo in the middle of the sweep
    qkv[5, 0] = -20.0                               # underflow, but other keys (feature 0 == 0) keep the sum at ~2^0: no repeat
    x = qkv.to(torch.bfloat16)
    out = ops.attention(x.cuda(), 1, T, H, q_log2=True).float().cpu()
    xd = x.double()
    q, k, v = xd[:, :64] / 1.4426950408889634, xd[:, 64:128], xd[:, 128:]
    ref = torch.softmax(q @ k.T, dim=-1) @ v
    assert torch.isfinite(out).all()
    np.testing.assert_allclose(out.double().numpy(), ref.numpy(), rtol=0, atol=2e-2)
    # every score of every query far below zero (all blocks repeat) and far above (all redo): still the same softmax
    for shift in (-150.0, 100.0):
        y = _rand(T, 192, seed=34, scale=0.3)
        y[:, 0] = shift ** 0.5 if shift > 0 else -((-shift) ** 0.5)    # q_0 k_0 = shift on every (query, key) pair
        y[:, 64] = abs(shift) ** 0.5
        yb = y.to(torch.bfloat16)
        out = ops.attention(yb.cuda(), 1, T, H, q_log2=True).float().cpu()
        yd = yb.double()
        ref = torch.softmax((yd[:, :64] / 1.4426950408889634) @ yd[:, 64:128].T, dim=-1) @ yd[:, 128:]
        np.testing.assert_allclose(out.double().numpy(), ref.numpy(), rtol=0, atol=2e-2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("q_log2", [False, True])
def test_attention_256_query_workgroups_give_the_same_bits(dtype, q_log2, attn_nw):
    """The 8-wave / 256-query workgroup form (default from 1024 positions on: a K / V tile is staged once for twice the queries)
    against the 4-wave / 128-query form (option attn_nw = 4): every query's sweep over the keys is the same arithmetic in the same
    order, so the outputs are identical -- T = 1500 (the encoder; a partial last query block in both forms) and 1100, peaked scores."""
    from lyricalignment_amd import ops
    for B, T, H in ((2, 1500, 3), (1, 1100, 2)):
        d = H * 64
        qkv = _rand(B * T, 3 * d, seed=40 + T, scale=1.0)
        qkv[:, :d] *= 0.125 * 3.0 * (1.4426950408889634 if q_log2 else 1.0)
        x = qkv.to(dtype).cuda()
        attn_nw(4)
        ref = ops.attention(x, B, T, H, q_log2=q_log2).clone()
        attn_nw(8)
        out8 = ops.attention(x, B, T, H, q_log2=q_log2).clone()
        attn_nw(0)
        out = ops.attention(x, B, T, H, q_log2=q_log2)
        assert torch.equal(out8.view(torch.int16), ref.view(torch.int16))
        assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_hand_placed_loop_matches_ping_pong_in_every_epilogue_mode(dtype, gemm_loop):
    """The default main loop of the 256x256 kernel (hand-placed flat stream, K % 128 == 0) against the quadrant ping-pong
    (option gemm_loop = 99) -- same bits -- in the epilogue modes the encoder uses beyond the plain ones: the LayerNorm-consumer epilogue
    (row statistics + column sums), the producer epilogue (f32 result + 16-bit copy + per-segment statistics), a batched launch,
    ragged last row / column tiles, both 16-bit operand types, K = 256 (two ring turns: prologue + peeled tail only) and K = 1280."""
    from lyricalignment_amd import ops
    for M, N, K in ((256 * 24 + 72, 2048 + 128, 1280), (256 * 16, 3072, 256)):       # >= 192 tiles: the 256x256 kernel's domain
        a = _rand(M, K, seed=191).to(dtype).cuda()
        w = _rand(N, K, seed=192, scale=K ** -0.5).to(dtype).cuda()
        bias = _rand(N, seed=193).cuda()
        res = _rand(M, N, seed=194).cuda()
        stats = torch.stack([_rand(M, seed=195) * 0.1, _rand(M, seed=196).abs() + 0.5], dim=1).contiguous().cuda()
        csum = _rand(N, seed=197).cuda()

        def run_all():
            outs = [ops.gemm(a, w, bias=bias, gelu=True, ln_stats=stats, ln_csum=csum).clone()]
            o32 = torch.empty(M, N, device="cuda")
            o16 = torch.empty(M, N, device="cuda", dtype=dtype)
            part = torch.zeros(N // 64, M, 2, device="cuda")
            ops.gemm(a, w, o32, bias=bias, residual=res, out_f32=True, out16=o16, ln_part=part)
            outs += [o32.clone(), o16.clone(), part.clone()]
            ab = a[: (M // 2 // 16) * 16 * 2].reshape(2, -1, K)              # two batch slots of equal row count
            mb = ab.shape[1]
            ob = torch.empty(2, mb, N, device="cuda", dtype=dtype)
            ops.gemm(ab.reshape(-1, K), w, ob.reshape(-1, N), bias=bias, M=mb, lda=K, batch=2, stride_a=mb * K, stride_c=mb * N)
            outs.append(ob.clone())
            return outs

        gemm_loop(0)
        new = run_all()
        gemm_loop(99)
        old = run_all()
        for x, y in zip(new, old):
            assert torch.equal(x, y)
        gemm_loop(0)


@pytest.mark.parametrize("B,T,H", [(1, 1, 64), (3, 7, 64), (32, 50, 128), (40, 23, 384), (2, 300, 384), (20, 11, 128)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gru_layer(B, T, H, dtype):
    """Persistent bidirectional GRU recurrence vs torch.nn.GRU (CPU fp32) fed the same input projections."""
    from lyricalignment_amd import ops
    I = 48
    gru = torch.nn.GRU(I, H, num_layers=1, batch_first=True, bidirectional=True)
    g = torch.Generator().manual_seed(40 + T)
    with torch.no_grad():
        for prm in gru.parameters():
            prm.copy_((torch.rand(prm.shape, generator=g) * 2 - 1) * (1.0 / H ** 0.5))
        w_hh = torch.stack([gru.weight_hh_l0, gru.weight_hh_l0_reverse])
        if dtype != torch.float32:  # the reference net uses the same (rounded) recurrent weights
            gru.weight_hh_l0.copy_(w_hh[0].to(dtype).float()); gru.weight_hh_l0_reverse.copy_(w_hh[1].to(dtype).float())
        x = torch.randn(B, T, I, generator=g)
        ref, _ = gru(x)
        gi = torch.stack([x @ gru.weight_ih_l0.T + gru.bias_ih_l0, x @ gru.weight_ih_l0_reverse.T + gru.bias_ih_l0_reverse], dim=2)
        b_hh = torch.stack([gru.bias_hh_l0, gru.bias_hh_l0_reverse])
    out, out_mish, flag = ops.gru_layer(gi.contiguous().cuda(), w_hh.to(dtype).contiguous().cuda(), b_hh.contiguous().cuda(), want_mish=True)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0, "bounded wait in the persistent GRU kernel timed out"
    tol = {torch.float32: 2e-5, torch.bfloat16: 2e-2, torch.float16: 3e-3}[dtype]  # 16-bit: h is rounded before every recurrent product
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.numpy(), rtol=0, atol=tol)
    np.testing.assert_allclose(out_mish.float().cpu().numpy(), torch.nn.functional.mish(ref).numpy(), rtol=0, atol=tol)
    if dtype != torch.float32 and H % 128 == 0:
        # both hand-off forms on every batch size (the default takes the granules from 17 clips on): identical bits
        from lyricalignment_amd import _lib
        for form in (1, 2):
            with _lib.option("gru_handoff", form):
                o2, m2, f2 = ops.gru_layer(gi.contiguous().cuda(), w_hh.to(dtype).contiguous().cuda(), b_hh.contiguous().cuda(), want_mish=True)
            assert int(f2.item()) == 0
            assert torch.equal(o2.view(torch.int16), out.view(torch.int16)) and torch.equal(m2.view(torch.int16), out_mish.view(torch.int16)), form


@pytest.mark.parametrize("variant", ["ctc", "plain"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,K,V", [(2, 37, 128, 300), (3, 150, 768, 21129)])
def test_fc_emissions_fused(variant, dtype, B, T, K, V):
    """Fused FC + row normaliser + gather vs (fp64 logits of the same rounded operands) -> the oracle's emission prep."""
    from lyricalignment_amd import ops
    from oracle import model_oracle as mo
    act = _rand(B * T, K, seed=50, scale=1.0).to(dtype)
    w = _rand(V, K, seed=51, scale=2.0 / K ** 0.5).to(dtype)
    bias = _rand(V, seed=52, scale=0.5)
    bias[-1] = 0.3
    rs = np.random.RandomState(53)
    Lmax = 26
    ncls = V - 2 if variant == "ctc" else V - 1
    labels = torch.from_numpy(rs.randint(1, ncls + 1, size=(B, Lmax)).astype(np.int32))
    labels[0, 5] = labels[0, 4]
    n_labels = torch.tensor([26, 11, 1][:B], dtype=torch.int32)
    var = 1 if variant == "ctc" else 0
    em = ops.fc_emissions(act.cuda(), w.cuda(), bias.cuda(), B, T, labels.cuda(), n_labels.cuda(), var).cpu()
    logits = (act.double() @ w.double().T + bias.double()).float().reshape(B, T, V)
    lp, ls = (mo.emission_prep_ctc if variant == "ctc" else mo.emission_prep_plain)(logits)
    tol = 2e-4 if dtype == torch.float32 else 1e-3  # same rounded operands; only accumulation order / exp2 differ
    for b in range(B):
        L = int(n_labels[b])
        np.testing.assert_allclose(em[b, :, 0].numpy(), ls[b, :, 0].numpy(), rtol=0, atol=tol)
        idx = labels[b, :L].long() - 1
        np.testing.assert_allclose(em[b, :, 1:1 + L].numpy(), lp[b][:, idx].numpy(), rtol=0, atol=tol)
        assert torch.equal(em[b, :, 5], em[b, :, 6]) if b == 0 else True  # repeated label -> identical columns


_GRU_DIGESTS = []


@pytest.mark.parametrize("fence", [False, True, "counter"])
def test_gru_handoff_under_uneven_load(fence):
    """The in-launch inter-workgroup hand-off of the persistent GRU must not depend on timing or placement: 12 runs of the
    config-2 shape while another stream keeps the chip busy with GEMMs of varying size give bit-identical outputs -- the same
    bits in every hand-off form.  fence=False: the default form at 32 clips (data-tagged 8-byte granules, no counter); "counter": the
    write-through counter form (option gru_handoff = 1: sc1 stores, drained; counter; sc1 loads); fence=True: the release /
    acquire fence form, which the library selects per process (LA_GRU_FENCE=1) -- that case runs this test body in a child process."""
    if fence == "counter":
        from lyricalignment_amd import _lib
        with _lib.option("gru_handoff", 1):
            test_gru_handoff_under_uneven_load(False)
        return
    if fence:
        import os, subprocess, sys
        if os.environ.get("LA_GRU_FENCE"):
            pytest.skip("already inside the LA_GRU_FENCE child")
        env = dict(os.environ, LA_GRU_FENCE="1")
        r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-m", "gpu", "-k",
                            "test_gru_handoff_under_uneven_load and False"], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
        return
    from lyricalignment_amd import ops
    B, T, H = 32, 400, 384
    g = torch.Generator().manual_seed(77)
    gi = (torch.randn(B, T, 2, 3 * H, generator=g) * 0.5).cuda()
    w = (torch.randn(2, 3 * H, H, generator=g) * H ** -0.5).bfloat16().cuda()
    b = (torch.randn(2, 3 * H, generator=g) * 0.1).cuda()
    ref, flag = ops.gru_layer(gi, w, b)
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    ref = ref.clone()
    import hashlib
    digest = hashlib.sha256(ref.view(torch.int16).cpu().numpy().tobytes()).hexdigest()
    _GRU_DIGESTS.append(digest)
    assert len(set(_GRU_DIGESTS)) == 1, "the hand-off forms disagree"
    side = torch.cuda.Stream()
    a = torch.randn(8192, 1024, device="cuda").bfloat16(); ww = torch.randn(4096, 1024, device="cuda").bfloat16()
    for it in range(12):
        with torch.cuda.stream(side):
            for _ in range(1 + it % 4):
                ops.gemm(a[: 1024 * (1 + it % 8)], ww)
        out, flag = ops.gru_layer(gi, w, b)
        torch.cuda.synchronize()
        assert int(flag.item()) == 0
        assert torch.equal(out, ref), f"run {it}: hand-off delivered stale or torn data"


def test_gemm_fused_layernorm_pieces():
    """la_row_stats16 and the two faces of la_gemm_fused_ln against torch: (a) the 16-bit copy of an f32 result,
    (b) LN(x) W^T + b computed from raw rows, gamma-folded weights, row statistics and column sums."""
    from lyricalignment_amd import ops, _lib
    M, d, N = 256 * 48 + 40, 1024, 1024                      # 49 x 4 tiles (>= 192: the 256x256 kernel), last row of tiles ragged
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(M, d, generator=g) * 1.5 + 0.4)
    x[:, 7] *= 20.0
    xb = x.bfloat16().cuda()
    st = ops.row_stats16(xb).cpu()
    xf = xb.float().cpu().double()
    np.testing.assert_allclose(st[:, 0].numpy(), xf.mean(1).numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(st[:, 1].numpy(), (1.0 / torch.sqrt(xf.var(1, unbiased=False) + 1e-5)).numpy(), rtol=2e-5)
    gamma = 1.0 + 0.1 * torch.randn(d, generator=g); beta = 0.1 * torch.randn(d, generator=g)
    w = torch.randn(N, d, generator=g) * 0.03; b = torch.randn(N, generator=g) * 0.1
    wl = (w.double() * gamma.double()[None, :]).bfloat16()
    csum = wl.double().sum(1).float().cuda()
    bl = (b.double() + w.double() @ beta.double()).float().cuda()
    out = ops.gemm(xb, wl.cuda(), bias=bl, ln_stats=st.cuda(), ln_csum=csum, out_f32=True).cpu()
    want = torch.nn.functional.layer_norm(xf, (d,), gamma.double(), beta.double(), 1e-5) @ w.double().T + b.double()
    np.testing.assert_allclose(out.double().numpy(), want.numpy(), rtol=0, atol=3e-2)    # bf16 operands, |want| ~ 1
    assert float((out.double() - want).abs().mean()) < 3e-3
    # (a) producer face: f32 result + residual, and its bf16 copy
    a = (torch.randn(M, d, generator=g) * 0.3).bfloat16().cuda()
    res = torch.randn(M, N, generator=g).cuda()
    c = res.clone(); c16 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    ops.gemm(a, wl.cuda(), c, bias=bl, residual=c, out_f32=True, out16=c16)
    c_ref = ops.gemm(a, wl.cuda(), bias=bl, residual=res, out_f32=True)
    assert torch.equal(c, c_ref) and torch.equal(c16, c_ref.bfloat16())
    # ... and the copy's row statistics taken inside that epilogue (per 64-column segment) + finalize == a read of the copy
    c = res.clone(); part = torch.full((N // 64, M, 2), float("nan"), device="cuda")
    ops.gemm(a, wl.cuda(), c, bias=bl, residual=c, out_f32=True, out16=c16, ln_part=part)
    assert torch.equal(c, c_ref) and torch.equal(c16, c_ref.bfloat16()) and bool(torch.isfinite(part).all())
    st_e = ops.ln_stats_finalize(part).cpu(); st_p = ops.row_stats16(c16).cpu()
    np.testing.assert_allclose(st_e[:, 0].numpy(), st_p[:, 0].numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(st_e[:, 1].numpy(), st_p[:, 1].numpy(), rtol=2e-5)
    # shapes that do not run on the 256x256 kernel are refused loudly
    with pytest.raises(NotImplementedError):
        ops.gemm(xb[:300], wl.cuda(), bias=bl, ln_stats=st[:300].contiguous().cuda(), ln_csum=csum, out_f32=True)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1024, 1024, 3000), (260, 136, 77), (64, 4096, 1500), (1152, 384, 33)])
def test_gemm_f32_transposed_operands(M, N, K):
    """la_gemm_ex with LA_GEMM_TRANS_A | LA_GEMM_TRANS_W: C[m][n] = sum_k At[k][m] Wt[k][n] reads both operands as [K][rows]
    (the weight-gradient shape; any K), and LA_GEMM_TRANS_W alone: C = A Wt (the input-gradient shape) -- against float64."""
    from lyricalignment_amd import head_train as ht
    at = _rand(K, M, seed=31, scale=0.3).cuda(); wt = _rand(K, N, seed=32, scale=0.3).cuda()
    ref = at.double().cpu().T @ wt.double().cpu()
    tol = 2e-4 * (K / 64) ** 0.5
    out = ht.gemm_tn(at, wt).cpu()                            # [M, N] = at^T wt
    np.testing.assert_allclose(out.double().numpy(), ref.numpy(), rtol=0, atol=tol)
    views = ht.gemm_tn(at[:, : M - 4] if M > 8 else at, wt).cpu()          # row views with a pitch larger than the row
    np.testing.assert_allclose(views.double().numpy(), (ref[: M - 4] if M > 8 else ref).numpy(), rtol=0, atol=tol)
    if K % 32 == 0:
        a = _rand(M, K, seed=33, scale=0.3).cuda()
        out = ht.gemm_nn(a, wt).cpu()                         # [M, N] = a wt
        np.testing.assert_allclose(out.double().numpy(), (a.double().cpu() @ wt.double().cpu()).numpy(), rtol=0, atol=tol)


def test_colsum_deterministic():
    """la_colsum_f32 (bias gradients): double partials per row chunk added in a fixed order; back-to-back calls of different
    shapes share the per-stream scratch."""
    from lyricalignment_amd import head_train as ht
    for rows, cols in [(3000, 1024), (7, 5), (3000, 4096), (80, 51865), (1, 64), (3000, 1024), (40000, 130)]:
        x = _rand(rows, cols, seed=rows + cols, scale=1.0).cuda()
        got = ht.colsum(x)
        again = ht.colsum(x)
        assert torch.equal(got, again)
        np.testing.assert_allclose(got.cpu().double().numpy(), x.double().sum(0).cpu().numpy(), rtol=0, atol=1e-6 * max(1.0, rows ** 0.5) * 4)
    v = _rand(3000, 2048, seed=9).cuda()[:, 100:1124]        # row view with a pitch
    np.testing.assert_allclose(ht.colsum(v).cpu().double().numpy(), v.double().sum(0).cpu().numpy(), rtol=0, atol=3e-4)


@pytest.mark.parametrize("dtype,ulp", [(torch.bfloat16, 2.0 ** -8), (torch.float16, 2.0 ** -11)])
def test_gemm_gelu_forms_of_the_16_bit_epilogue(dtype, ulp):
    """include/lyricalign.h's contract for LA_EPI_GELU with a 16-bit result: the default sigmoid fit is within 2.5e-5 ABSOLUTE
    (+ the output rounding) of the erf GELU (whisper's F.gelu); LA_EPI_GELU_ERF (gelu="erf") within 1.3e-6 (+ rounding)."""
    from lyricalignment_amd import ops
    M, N, K = 256 * 48, 1024, 256
    a = (_rand(M, K, seed=71) * 1.5).to(dtype).cuda()
    w = _rand(N, K, seed=72, scale=K ** -0.5).to(dtype).cuda()
    bias = _rand(N, seed=73).cuda()
    pre = a.double() @ w.double().T + bias.double()
    ref = torch.nn.functional.gelu(pre)                       # erf form, float64
    fit = ops.gemm(a, w, bias=bias, gelu=True).double()
    erf = ops.gemm(a, w, bias=bias, gelu="erf").double()
    tol_round = ref.abs() * ulp + 1e-6                         # output rounding (half an ulp would do) + f32 accumulation of K = 256
    assert float(pre.abs().max()) > 4.0 and float(pre.min()) < -3.0        # the negative tail is in the sample
    assert float(((fit - ref).abs() - tol_round).max()) < 2.5e-5
    assert float(((erf - ref).abs() - tol_round).max()) < 1.3e-6
    assert not torch.equal(fit, erf)                           # the flag does select another form


def test_gemm_ping_pong_loop_is_bit_identical_to_the_hand_placed_one(gemm_loop):
    """The A/B partner of the 256x256 kernel's default hand-placed loop that ships (option gemm_loop = 99: the quadrant ping-pong, the
    loop of every K that is not a multiple of 128) walks k in the same order per accumulator and applies the same epilogue
    arithmetic: identical bits, including the ragged last row / column of tiles.  (The measured-slower structures -- one wave per
    SIMD, persistent tiles, four-wave workgroups -- live in the experiment build: tests/test_gpu_lab.py.)"""
    from lyricalignment_amd import ops
    M, N, K = 256 * 49 + 40, 1024 + 64, 1024
    a = _rand(M, K, seed=91).bfloat16().cuda()
    w = _rand(N, K, seed=92, scale=K ** -0.5).bfloat16().cuda()
    bias = _rand(N, seed=93).cuda()
    res = _rand(M, N, seed=94).cuda()
    gemm_loop(0)
    ref16 = ops.gemm(a, w, bias=bias, gelu=True).clone()
    ref32 = ops.gemm(a, w, bias=bias, residual=res, out_f32=True).clone()
    gemm_loop(99)
    for _ in range(3):
        assert torch.equal(ops.gemm(a, w, bias=bias, gelu=True), ref16)
        assert torch.equal(ops.gemm(a, w, bias=bias, residual=res, out_f32=True), ref32)


def test_library_options_round_trip_and_shipped_library_has_no_experiments():
    """la_set_option / la_get_option (include/lyricalign.h): every documented name reads back what was set, unknown names are refused,
    and the library the tests run on is the shipped one: la_has_experiments() == 0 unless LA_LIB_PATH selects another build."""
    import os
    from lyricalignment_amd import _lib
    for name in ("gemm_tile", "gemm_loop", "gemm_splitk", "attn_nw", "gru_nw", "gru_fence", "gru_handoff", "viterbi_dpp", "head_clip_cap", "ln_fusion", "resid_split", "x2_inference"):
        before = _lib.get_option(name)
        with _lib.option(name, 7):
            assert _lib.get_option(name) == 7
        assert _lib.get_option(name) == before
    with pytest.raises(ValueError):
        _lib.set_option("no_such_option", 1)
    if not os.environ.get("LA_LIB_PATH"):
        assert not _lib.has_experiments()
    with pytest.raises(NotImplementedError):       # in-loop LayerNorm statistics: experiment build only
        a = torch.zeros(256 * 48, 1024, dtype=torch.bfloat16, device="cuda")
        from lyricalignment_amd import ops
        if _lib.has_experiments():
            raise NotImplementedError
        ops.gemm(a, a[:1024].contiguous(), ln_csum=torch.zeros(1024, device="cuda"))


def test_kernel_timer_sampling_and_work_accounting():
    """la_timer_sample / la_timer_read_work (bench.py's roofline leg): with period 3, launches 0, 3, 6 of the family are bracketed
    by HIP events, their 2 M N K is summed, every launch is counted, other families are ignored."""
    import ctypes
    from lyricalignment_amd import _lib, ops
    L = _lib.lib()
    a = _rand(512, 256, seed=1).bfloat16().cuda()
    w = _rand(384, 256, seed=2, scale=0.06).bfloat16().cuda()
    x = _rand(64, 128, seed=3).cuda()
    L.la_timer_reset()
    assert L.la_timer_sample(0) != 0                  # period must be >= 1
    assert L.la_timer_sample(3) == 0
    L.la_timer_enable(b"gemm_bf16")
    try:
        for i in range(7):
            ops.gemm(a, w)
            ops.layernorm(x, torch.ones(128, device="cuda"), torch.zeros(128, device="cuda"), torch.float32)   # another family
        torch.cuda.synchronize()
    finally:
        L.la_timer_disable()
    ms, timed, work, seen = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_int64(0)
    assert L.la_timer_read_work(ctypes.byref(ms), ctypes.byref(timed), ctypes.byref(work), ctypes.byref(seen)) == 0
    assert (timed.value, seen.value) == (3, 7) and ms.value > 0
    assert work.value == 3 * 2.0 * 512 * 384 * 256
    L.la_timer_reset()
    L.la_timer_sample(1)


# ------------------------------------------------------------------------------------------------ split residual stream
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_split_stream_matches_the_f32_stream(dtype):
    """la_gemm_split (the residual GEMMs of the 16-bit encoder on the SPLIT stream: hi 16-bit + lo byte) against the f32-stream
    form of the same kernel (la_gemm_fused_ln with a 16-bit copy), same operands: hi is bit-for-bit the 16-bit copy, the decoded
    stream is within half a lo step (ulp(hi) / 508) of the f32 row, for the stem form (GELU + f32 residual, batched), the in-place
    form (x += ...), and partial last tiles (M = 6000 = 23.4 row tiles: the element-wise edge path)."""
    from lyricalignment_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = 6000, 2048, 256
    sh = 16 if dtype == torch.bfloat16 else 19

    def unit(hi):
        _, e = torch.frexp(hi.float())
        if dtype == torch.float16:
            e = e.clamp(min=-13)                   # f16 subnormals: the ulp of the smallest normal binade
        return torch.ldexp(torch.ones_like(hi, dtype=torch.float32), e.to(torch.int32) - sh)

    a = (torch.randn(M, K, device="cuda", generator=g)).to(dtype)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dtype)
    bias = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(M, N, device="cuda", generator=g) * 3.0
    res[5, 7] = 0.0
    # stem form: f32 residual in, split stream out
    x = torch.empty(M, N, device="cuda")
    h = torch.empty(M, N, device="cuda", dtype=dtype)
    ops.gemm(a, w, x, bias=bias, residual=res, gelu=True, out_f32=True, out16=h)
    hi = torch.empty(M, N, device="cuda", dtype=dtype)
    lo = torch.zeros(M, N, device="cuda", dtype=torch.uint8)
    ops.gemm_split(a, w, hi, lo, bias=bias, residual=res, gelu=True)
    assert torch.equal(hi.view(torch.int16), h.view(torch.int16))
    dec = ops.split_decode(hi, lo)
    # (half a lo step + a few f32 ulps: the two instantiations may contract GELU's last multiply with the residual add differently,
    #  and with f16 the lo step is only 32 f32 ulps)
    assert bool(((dec - x).abs() <= 0.505 * unit(hi) + 1e-6 * x.abs()).all())
    assert float((dec - x).abs().max()) < float((h.float() - x).abs().max()) / 100          # 8 more bits than the 16-bit copy alone
    # in-place form: the stream is its own residual
    a2 = (torch.randn(M, K, device="cuda", generator=g)).to(dtype)
    w2 = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dtype)
    x0 = ops.split_decode(hi, lo)
    x1 = x0.clone()
    h1 = torch.empty_like(h)
    ops.gemm(a2, w2, x1, bias=bias, residual=x1, out_f32=True, out16=h1)
    ops.gemm_split(a2, w2, hi, lo, bias=bias, in_place=True)
    assert torch.equal(hi.view(torch.int16), h1.view(torch.int16))
    # (+ an f32 ulp of the LARGER of old and new value: the kernel decodes the old stream with one fused multiply-add where torch
    #  rounds twice, and x0 + delta may cancel to something whose lo step is smaller than that ulp)
    assert bool(((ops.split_decode(hi, lo) - x1).abs() <= 0.505 * unit(hi) + 1e-6 * torch.maximum(x1.abs(), x0.abs())).all())
    # ln_part (statistics of the hi rows from inside the epilogue) and LayerNorm over the decoded rows
    part = torch.empty(N // 64, M, 2, device="cuda")
    hi2, lo2 = hi.clone(), lo.clone()
    ops.gemm_split(a, w, hi2, lo2, bias=bias, in_place=True, ln_part=part)
    st = ops.ln_stats_finalize(part)
    ref = ops.row_stats16(hi2)
    assert torch.allclose(st, ref, rtol=2e-5, atol=2e-6)
    gam, bet = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    y = ops.layernorm_split(hi2, lo2, gam, bet, torch.float32)
    y_ref = ops.layernorm(ops.split_decode(hi2, lo2), gam, bet, torch.float32)
    assert torch.allclose(y, y_ref, rtol=0, atol=2e-5)       # (the kernel's decode is one fused multiply-add, torch's rounds twice)
    with pytest.raises(NotImplementedError):
        ops.gemm_split(a[:256], w[:128], hi[:256, :128].contiguous(), lo[:256, :128].contiguous())      # not a 256x256-kernel shape


def test_gemm_split_batched_stem_layout_and_zero_rows():
    """The stem's call shape: batch of clips, overlapping-row A view, residual rows shared by every clip (stride_r = 0), hi / lo
    with a batch stride; an all-zero product row decodes to exactly the residual it was given."""
    from lyricalignment_amd import ops
    g = torch.Generator(device="cuda").manual_seed(4)
    B, T, d = 12, 1500, 1024
    y1 = torch.zeros(B, 2 * T + 2, d, device="cuda", dtype=torch.bfloat16)
    y1[1:] = (torch.randn(B - 1, 2 * T + 2, d, device="cuda", generator=g) * 0.5).to(torch.bfloat16)     # clip 0: zero activations
    w = (torch.randn(d, 3 * d, device="cuda", generator=g) * (3 * d) ** -0.5).to(torch.bfloat16)
    pos = torch.randn(T, d, device="cuda", generator=g)
    x = torch.empty(B * T, d, device="cuda")
    h = torch.empty(B * T, d, device="cuda", dtype=torch.bfloat16)
    kw = dict(M=T, lda=2 * d, batch=B, stride_a=(2 * T + 2) * d, stride_c=T * d, ldr=d, stride_r=0)
    ops.gemm(y1, w, x, residual=pos, out_f32=True, ldc=d, out16=h, **kw)
    hi = torch.empty_like(h)
    lo = torch.empty(B * T, d, device="cuda", dtype=torch.uint8)
    ops.gemm_split(y1, w, hi, lo, residual=pos, ld=d, **kw)
    assert torch.equal(hi.view(torch.int16), h.view(torch.int16))
    dec = ops.split_decode(hi, lo)
    assert float((dec - x).abs().max()) <= float(x.abs().max()) * 1.6e-5
    assert float((dec[:T] - pos).abs().max()) <= float(pos.abs().max()) * 1.6e-5


# ------------------------------------------------------------------------------------------------ float32 products as f16x2
def _heavy(rows, cols, seed, scale=1.0):
    """Gaussian with outlier columns (x 30) and a log-normal spread of row magnitudes: activation / gradient-like dynamic range."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(rows, cols, generator=g)
    x[:, torch.randperm(cols, generator=g)[: max(1, cols // 128)]] *= 30.0
    x *= torch.exp(torch.randn(rows, 1, generator=g) * 1.5)
    return x * scale


def test_split_f16x2_planes_reconstruct_the_float32_values():
    """la_split_f16x2 / la_split_f16x2_t: (hi + lo) * inv_scale gives x back to 2^-21 of each element (22 bits) or 2^-38 of its row's
    (column's) largest magnitude, whichever is larger (half's subnormal floor); inverse scales are powers of two that put the largest
    magnitude in [2^13, 2^14); zero rows, a zero matrix tail (padding) and rows of tiny / huge magnitude included; the transposed
    form equals the plain form of x^T bit for bit."""
    from lyricalignment_amd import f32x2
    x = _heavy(1000, 777, seed=1)
    x[5] = 0.0
    x[6] *= 1e-30
    x[7] *= 1e25
    x[8, 3] = 0.0
    xd = x.cuda()
    for P, ref, axis_max in ((f32x2.split(xd, 896), x, x.abs().amax(dim=1, keepdim=True)),
                             (f32x2.split_t(xd, 1024), x.t().contiguous(), x.abs().amax(dim=0)[:, None])):
        planes, inv = P.planes.cpu(), P.inv_scale.cpu()
        n = ref.shape[1]
        assert planes.shape == (ref.shape[0], 2, P.kp) and float(planes[:, :, n:].abs().max()) == 0.0
        rec = (planes[:, 0, :n].double() + planes[:, 1, :n].double()) * inv.double()[:, None]
        err = (rec - ref.double()).abs()
        bound = torch.maximum(ref.double().abs() * 2.0 ** -21, axis_max.double() * 2.0 ** -38)
        assert bool((err <= bound).all()), float((err / bound.clamp_min(1e-300)).max())
        m, e = torch.frexp(inv)
        assert bool((m == 0.5).all())                                           # powers of two
        scaled = axis_max[:, 0].double() / inv.double()
        nz = axis_max[:, 0] > 0
        assert bool(((scaled[nz] >= 2.0 ** 13) & (scaled[nz] < 2.0 ** 14)).all()) and bool((inv[~nz] == 1.0).all())
    a, b = f32x2.split_t(xd, 1024), f32x2.split(xd.t().contiguous(), 1024)
    assert torch.equal(a.planes.view(torch.int16), b.planes.view(torch.int16)) and torch.equal(a.inv_scale, b.inv_scale)
    # rows longer than the 4096 columns the split keeps in registers (its two-pass form), odd row pitch (scalar loads)
    for cols, pitch in ((4200, 4200), (4200, 4201), (1000, 1001)):
        w = _heavy(70, pitch, seed=cols + pitch)[:, :cols]
        P = f32x2.split(w.cuda(), 4224 if cols > 4096 else 1024)
        rec = (P.planes[:, 0, :cols].double() + P.planes[:, 1, :cols].double()).cpu() * P.inv_scale.cpu().double()[:, None]
        bound = torch.maximum(w.double().abs() * 2.0 ** -21, w.abs().amax(dim=1, keepdim=True).double() * 2.0 ** -38)
        assert bool(((rec - w.double()).abs() <= bound).all()), (cols, pitch)


def test_split_f16x2_with_gelu_equals_the_split_of_the_gelu_buffer():
    """act = "gelu" (la_split_f16x2_act / la_split_f16x2_t_act): the planes and scales of gelu(x) without its float32 buffer are, bit
    for bit, those of splitting la_gelu_f32's output (same erf form in both); f32x2.linear / gemm_tn with the activation inside the
    split give the bits of the two-step form; an unknown activation code is refused."""
    from lyricalignment_amd import encoder_train, f32x2
    from lyricalignment_amd._lib import lib, ptr
    x = (_heavy(3000, 1100, seed=31, scale=0.02)).cuda()
    gx = encoder_train.gelu(x)
    for fn, pad in ((f32x2.split, 1152), (f32x2.split_t, 3072)):
        a, b = fn(x, pad, act="gelu"), fn(gx, pad)
        assert torch.equal(a.planes.view(torch.int16), b.planes.view(torch.int16)) and torch.equal(a.inv_scale, b.inv_scale)
    w, dy = _rand(1536, 1100, seed=32, scale=0.03).cuda(), _heavy(3000, 1536, seed=33, scale=1e-3).cuda()
    assert f32x2.eligible(3000, 1536, 1100) and f32x2.eligible(1536, 1100, 3000)
    assert torch.equal(f32x2.linear(x, w, x_act="gelu"), f32x2.linear(gx, w))
    assert torch.equal(f32x2.gemm_tn(dy, x, x_act="gelu"), f32x2.gemm_tn(dy, gx))
    small = x[:64, :192].contiguous()                                           # outside the f16x2 domain: the float32 kernel on gelu(x)
    assert torch.equal(f32x2.linear(small, w[:, :192].contiguous(), x_act="gelu"), f32x2.linear(encoder_train.gelu(small), w[:, :192].contiguous()))
    P = f32x2.split(x, 1152)
    assert lib().la_split_f16x2_act(ptr(x), x.stride(0), 3000, 1100, ptr(P.planes), 1152, ptr(P.inv_scale), 7, None) != 0


@pytest.mark.parametrize("kind", ["gauss", "heavy"])
def test_gemm_f16x2_is_at_least_as_accurate_as_the_float32_kernel(kind):
    """la_gemm_f16x2 (three f16 products over segmented K, scale epilogue) against a float64 product, next to float32 la_gemm on the
    same operands: max |err| / max |ref| no larger than the float32 kernel's (measured 2-3 x smaller) and below 2e-6; bias + residual
    epilogue; a ragged last row / column of tiles; K not a multiple of 128 (zero-padded planes)."""
    from lyricalignment_amd import f32x2, ops
    M, N, K = 256 * 48 + 40, 1024 + 72, 1056
    mk = _heavy if kind == "heavy" else (lambda r, c, seed, scale=1.0: _rand(r, c, seed=seed, scale=scale))
    a, w = mk(M, K, seed=11), _rand(N, K, seed=12, scale=K ** -0.5)
    bias, res = _rand(N, seed=13), _rand(M, N, seed=14)
    ad, wd = a.cuda(), w.cuda()
    assert f32x2.eligible(M, N, K)
    out = f32x2.linear(ad, wd, bias=bias.cuda(), residual=res.cuda()).cpu()
    nat = ops.gemm(ad, wd, bias=bias.cuda(), residual=res.cuda()).cpu()
    rows = torch.cat([torch.arange(0, 512), torch.arange(M - 64, M)])
    ref = a[rows].double() @ w.double().t() + bias.double() + res[rows].double()
    e_x2 = float((out[rows].double() - ref).abs().max() / ref.abs().max())
    e_f32 = float((nat[rows].double() - ref).abs().max() / ref.abs().max())
    print(f"f16x2 {e_x2:.2e} vs float32 kernel {e_f32:.2e}")
    assert e_x2 <= max(e_f32, 4e-7) and e_x2 < 2e-6


def test_f16x2_gradient_products_with_split_k_slots():
    """f32x2.gemm_nn (dx = dy w) and gemm_tn (dw = dy^T x: few tiles, the 6000-row contraction cut into split-K slots summed in slot
    order) against float64 and against the float32 kernels; the slotted product is deterministic (two runs, same bits)."""
    from lyricalignment_amd import f32x2, head_train
    M, N, K = 6000, 1536, 512
    dy, x, w = _heavy(M, N, seed=21, scale=1e-4), _heavy(M, K, seed=22), _rand(N, K, seed=23, scale=K ** -0.5)
    dyd, xd, wd = dy.cuda(), x.cuda(), w.cuda()
    assert f32x2.eligible(N, K, M) and f32x2.slots_for(N, K) > 1 and f32x2.eligible(M, K, N)
    dw = f32x2.gemm_tn(dyd, xd)
    assert torch.equal(dw, f32x2.gemm_tn(dyd, xd))
    dx = f32x2.gemm_nn(dyd, wd)
    ref_dw, ref_dx = dy.double().t() @ x.double(), dy.double() @ w.double()
    for got, nat, ref in ((dw.cpu(), head_train.gemm_tn_f32(dyd, xd).cpu(), ref_dw), (dx.cpu(), head_train.gemm_nn_f32(dyd, wd).cpu(), ref_dx)):
        e_x2 = float((got.double() - ref).abs().max() / ref.abs().max())
        e_f32 = float((nat.double() - ref).abs().max() / ref.abs().max())
        assert e_x2 <= max(e_f32, 4e-7) and e_x2 < 3e-6, (e_x2, e_f32)
    # the operands' maxima from their plain splits (f32x2.OperandMax) scale the transposed splits: one power-of-two scale per operand, no pass
    # for column maxima; the weight gradient and the bias gradient (now from the split kernel itself) keep the float32-level accuracy
    m_dy, m_x = f32x2.OperandMax(dyd.device), f32x2.OperandMax(dyd.device)
    f32x2.gemm_nn(dyd, wd, dy_max=m_dy)
    f32x2.linear(xd, _rand(1536, K, seed=25, scale=K ** -0.5).cuda(), x_max=m_x)
    assert m_dy.valid and m_x.valid
    assert float(torch.from_numpy(m_dy.words.cpu().numpy().view("float32")).max()) == float(dy.abs().max())
    Pt = f32x2.split_t(dyd, 6016, omax=m_dy)
    assert int(Pt.inv_scale.unique().numel()) == 1                       # one scale for the whole operand
    rec = (Pt.planes[:, 0, :M].double() + Pt.planes[:, 1, :M].double()).cpu() * Pt.inv_scale.cpu().double()[:, None]
    assert float((rec - dy.double().t()).abs().max()) <= float(dy.abs().max()) * 2.0 ** -21
    db_m = torch.empty((N,), dtype=torch.float32, device=dyd.device)
    dw_m = f32x2.gemm_tn(dyd, xd, colsum=db_m, dy_max=m_dy, x_max=m_x)
    e_m = float((dw_m.cpu().double() - ref_dw).abs().max() / ref_dw.abs().max())
    assert e_m < 3e-6 and torch.equal(dw_m, f32x2.gemm_tn(dyd, xd, dy_max=m_dy, x_max=m_x)), e_m
    assert float((db_m.cpu().double() - dy.double().sum(0)).abs().max()) <= 1e-6 * float(dy.abs().sum(0).max())
    # dx times gelu'(u) in the product's epilogue (LA_EPI_RES_GELU_GRAD): the bits of la_gelu_bwd_f32 on the plain product
    from lyricalignment_amd import encoder_train
    u = _rand(M, K, seed=24, scale=1.5).cuda()
    assert torch.equal(f32x2.gemm_nn(dyd, wd, gelu_grad_of=u), encoder_train.gelu_bwd(u, dx))
    # the bias gradient from the same pass over dy (la_split_f16x2_t_colsum): float64 partials per row block, added in order
    dw2, db = head_train.linear_grads(dyd, xd)
    assert torch.equal(dw2, dw) and torch.equal(db, head_train.linear_grads(dyd, xd)[1])
    ref_db = dy.double().sum(0)
    assert float((db.cpu().double() - ref_db).abs().max()) <= 1e-6 * float(dy.abs().sum(0).max())
    assert float((head_train.colsum(dyd).cpu().double() - ref_db).abs().max()) <= 1e-6 * float(dy.abs().sum(0).max())
    odd = dyd[:, :1534].contiguous()                 # columns not a multiple of 4 / products outside the f16x2 domain: la_colsum_f32 behind the same call
    _, db_odd = head_train.linear_grads(odd[:200], xd[:200])
    assert float((db_odd.cpu().double() - dy[:200, :1534].double().sum(0)).abs().max()) <= 1e-6 * float(dy[:200].abs().sum(0).max())
    with pytest.raises(NotImplementedError):          # outside the 256 x 256 kernel's domain: the C entry point refuses, the wrappers never ask
        f32x2.gemm(f32x2.split(xd[:300], 512), f32x2.split(wd, 512))
