"""The EXPERIMENT build of the library (tools/build_variant.sh lab -DLA_EXPERIMENTS -> ab/lab/liblyricalign_hip.so): the GEMM structures
of rounds 2-4 that were measured slower than the shipped 256x256 kernel stay buildable and BIT-IDENTICAL to it -- persistent tiles,
four-wave workgroups two per CU, one wave per SIMD, LayerNorm statistics inside the consumer's main loop.  None of them is in the
shipped library (tests/test_gpu_ops.py::test_library_options_round_trip_and_shipped_library_has_no_experiments).

The library is chosen per process (LA_LIB_PATH), so the first test below re-runs this file in a child process on the experiment
build; the tests themselves skip unless the loaded library has the experiments."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB_LIB = os.path.join(ROOT, "ab", "lab", "liblyricalign_hip.so")


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def test_experiment_build_in_a_child_process():
    """Runs the rest of this file on ab/lab/liblyricalign_hip.so (skipped when that build is not there: the experiment build is a
    developer tool, `bash tools/build_variant.sh lab -DLA_EXPERIMENTS`)."""
    if os.environ.get("LA_LIB_PATH"):
        pytest.skip("already inside the child")
    if not os.path.exists(LAB_LIB):
        pytest.skip("experiment build not present (bash tools/build_variant.sh lab -DLA_EXPERIMENTS)")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-m", "gpu", "-k", "not child_process"],
                       env=dict(os.environ, LA_LIB_PATH=LAB_LIB), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-3000:] + r.stderr[-2000:]


@pytest.fixture(autouse=True)
def _needs_experiments(request):
    if "child_process" in request.node.name:
        return
    from lyricalignment_amd import _lib
    if not _lib.has_experiments():
        pytest.skip("the loaded library is the shipped one (no experiments)")


def test_gemm_one_wave_per_simd_kernel_is_bit_identical():
    """gemm_mono_kernel (LA_PP_DBG=73, read per launch: one wave per SIMD with 128x128 wave tiles and its own AGPR-direct epilogue)
    walks k in the same order per accumulator and applies the same epilogue arithmetic as the shipped loop: identical bits,
    including the ragged last row / column of tiles."""
    from lyricalignment_amd import ops
    M, N, K = 256 * 49 + 40, 1024 + 64, 1024
    a = _rand(M, K, seed=91).bfloat16().cuda()
    w = _rand(N, K, seed=92, scale=K ** -0.5).bfloat16().cuda()
    bias = _rand(N, seed=93).cuda()
    res = _rand(M, N, seed=94).cuda()
    os.environ.pop("LA_PP_DBG", None)
    ref16 = ops.gemm(a, w, bias=bias, gelu=True).clone()
    ref32 = ops.gemm(a, w, bias=bias, residual=res, out_f32=True).clone()
    os.environ["LA_PP_DBG"] = "73"
    try:
        for _ in range(3):
            assert torch.equal(ops.gemm(a, w, bias=bias, gelu=True), ref16)
            assert torch.equal(ops.gemm(a, w, bias=bias, residual=res, out_f32=True), ref32)
    finally:
        os.environ.pop("LA_PP_DBG", None)


@pytest.mark.parametrize("dtype", [torch.bfloat16])
def test_gemm_layernorm_consumer_takes_row_statistics_in_its_main_loop(dtype):
    """la_gemm_fused_ln with ln_csum but NO ln_stats: the hand-placed main loop sums every A row and its squares from the fragments
    it multiplies (v_dot2c in MFMA gaps) and the epilogue applies mean / rstd from LDS -- against the same launch fed with
    la_row_stats16's two-pass statistics and against float64 LayerNorm + matmul.  Rows with a large common offset (mean = 30 sigma,
    the one-pass variance's worst case), an outlier channel, an all-equal row (variance 0), a ragged last row of tiles, GELU."""
    from lyricalignment_amd import ops
    M, d, N = 256 * 48 + 40, 1024, 1024
    g = torch.Generator().manual_seed(6)
    x = torch.randn(M, d, generator=g) * 1.5 + 0.4
    x[:, 7] *= 20.0
    x[100:200] += 45.0                                        # mean = 30 sigma
    x[300] = 3.25                                             # variance exactly 0
    xb = x.to(dtype).cuda()
    xf = xb.float().cpu().double()
    gamma = 1.0 + 0.1 * torch.randn(d, generator=g); beta = 0.1 * torch.randn(d, generator=g)
    w = torch.randn(N, d, generator=g) * 0.03; b = torch.randn(N, generator=g) * 0.1
    wl = (w.double() * gamma.double()[None, :]).to(dtype)
    csum = wl.double().sum(1).float().cuda()
    bl = (b.double() + w.double() @ beta.double()).float().cuda()
    st = ops.row_stats16(xb)
    for gelu in (False, True):
        ref = ops.gemm(xb, wl.cuda(), bias=bl, gelu=gelu, ln_stats=st, ln_csum=csum, out_f32=not gelu).float().cpu()
        out = ops.gemm(xb, wl.cuda(), bias=bl, gelu=gelu, ln_csum=csum, out_f32=not gelu).float().cpu()
        want = torch.nn.functional.layer_norm(xf, (d,), gamma.double(), beta.double(), 1e-5) @ w.double().T + b.double()
        if gelu:
            want = torch.nn.functional.gelu(want)
        ok = torch.ones(M, dtype=torch.bool); ok[100:200] = False
        # ordinary rows: the in-loop statistics change the result by far less than the 16-bit operands do
        assert float((out[ok] - ref[ok]).abs().max()) < (2e-3 if not gelu else 2e-2)
        np.testing.assert_allclose(out[ok].double().numpy(), want[ok].numpy(), rtol=0, atol=4e-2)
        assert float((out[ok].double() - want[ok]).abs().mean()) < 4e-3
        # offset rows: sum x^2 - n mean^2 loses ~3 digits of the variance at mean = 30 sigma; still inside the 16-bit result's own error
        np.testing.assert_allclose(out[~ok].double().numpy(), want[~ok].numpy(), rtol=0, atol=2.5e-1 if dtype == torch.bfloat16 else 6e-2)
        assert bool(torch.isfinite(out).all())
    with pytest.raises(NotImplementedError):                      # K = 192 is not a multiple of 128: no hand-placed main loop
        ops.gemm(xb[:, :192].contiguous(), wl[:, :192].contiguous().cuda(), bias=bl, ln_csum=csum)


# ------------------------------------------------------------------------------------------------ persistent 256x256 kernel
@pytest.mark.parametrize("dtype", [torch.bfloat16])
def test_gemm_persistent_kernel_is_bit_identical_to_one_workgroup_per_tile(dtype):
    """gemm_pp_persist_kernel (opt-in, LA_GEMM_PERSIST=1 read per launch, bf16: workgroups draw tiles from per-XCD ticket counters and
    issue the next tile's first stages before the current tile's epilogue) against the one-workgroup-per-tile kernel: the same main loop, the
    same epilogue arithmetic, so the same bits -- for the plain, LayerNorm-consumer (+ GELU) and split-stream
    forms, with a partial last row of tiles (M = 48000 = 187.5 x 256: edge tiles break the prefetch chain), five launches each (a
    race in the ticket / prefetch choreography would show as a differing run), while another stream keeps CUs busy so that some
    persistent workgroups start late and the tickets have to balance."""
    from lyricalignment_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    M = 48000
    side = torch.cuda.Stream()
    sa = torch.randn(8192, 1024, device="cuda", generator=g).to(dtype)
    sw = torch.randn(4096, 1024, device="cuda", generator=g).to(dtype)

    def both(fn):
        os.environ.pop("LA_GEMM_PERSIST", None)
        ref = [t.clone() for t in fn()]
        for it in range(5):
            with torch.cuda.stream(side):
                for _ in range(1 + it % 3):
                    ops.gemm(sa[: 2048 * (1 + it % 4)], sw)
            os.environ["LA_GEMM_PERSIST"] = "1"
            try:
                out = fn()
            finally:
                os.environ.pop("LA_GEMM_PERSIST", None)
            torch.cuda.synchronize()
            for a_, b_ in zip(out, ref):
                assert torch.equal(a_.view(torch.int16) if a_.dtype != torch.float32 and a_.dtype != torch.uint8 else a_,
                                   b_.view(torch.int16) if b_.dtype != torch.float32 and b_.dtype != torch.uint8 else b_), f"run {it}"

    for N, K in ((3072, 1024), (1024, 4096)):
        a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dtype)
        bias = torch.randn(N, device="cuda", generator=g)
        csum = torch.randn(N, device="cuda", generator=g)
        stats = torch.stack([torch.randn(M, device="cuda", generator=g) * 0.1, 1.0 + 0.1 * torch.rand(M, device="cuda", generator=g)], dim=1).contiguous()
        out16 = torch.empty(M, N, device="cuda", dtype=dtype)
        both(lambda: [ops.gemm(a, w, out16, bias=bias)])                                                    # plain
        both(lambda: [ops.gemm(a, w, out16, bias=bias, gelu=True, ln_stats=stats, ln_csum=csum)])          # LayerNorm consumer + GELU
        hi0 = torch.randn(M, N, device="cuda", generator=g).to(dtype)
        lo0 = torch.randint(1, 256, (M, N), device="cuda", generator=g, dtype=torch.uint8)

        def split():
            hi, lo = hi0.clone(), lo0.clone()
            ops.gemm_split(a, w, hi, lo, bias=bias, in_place=True)
            return [hi, lo]
        both(split)                                                                                         # split stream in place


# ------------------------------------------------------------------------------------------------ four-wave workgroups, two per CU
@pytest.mark.parametrize("form", ["1", "2"])
def test_gemm_four_wave_two_workgroups_per_cu_is_bit_identical(form):
    """gemm_q4_kernel (opt-in, LA_GEMM_Q4 read per launch, bf16: 256 x 128 tiles (1) or 128 x 256 tiles (2), four waves, a ring of
    three stages, two workgroups resident per CU) against the 8-wave kernel: the same wave tiles, accumulation order and epilogues, so
    the same bits -- plain, LayerNorm consumer (+ GELU) and the split stream in place; M with a partial last row of tiles, N with a
    partial last column tile in the plain form (edge epilogue), K = 256 (loop body never runs), 1024 and 4096."""
    from lyricalignment_amd import ops
    g = torch.Generator(device="cuda").manual_seed(12)
    dtype = torch.bfloat16

    def both(fn):
        os.environ.pop("LA_GEMM_Q4", None)
        ref = [t.clone() for t in fn()]
        os.environ["LA_GEMM_Q4"] = form
        try:
            out = fn()
        finally:
            os.environ.pop("LA_GEMM_Q4", None)
        torch.cuda.synchronize()
        for a_, b_ in zip(out, ref):
            assert torch.equal(a_.view(torch.int16) if a_.dtype == dtype else a_, b_.view(torch.int16) if b_.dtype == dtype else b_)

    for M, N, K in ((12500, 1024, 256), (6100, 3072, 1024), (12500, 1024, 4096)):       # (>= 192 tiles: the 256x256 kernel's launches)
        a = torch.randn(M, K, device="cuda", generator=g).to(dtype)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(dtype)
        bias = torch.randn(N, device="cuda", generator=g)
        csum = torch.randn(N, device="cuda", generator=g)
        stats = torch.stack([torch.randn(M, device="cuda", generator=g) * 0.1, 1.0 + 0.1 * torch.rand(M, device="cuda", generator=g)], dim=1).contiguous()
        out16 = torch.empty(M, N, device="cuda", dtype=dtype)
        both(lambda: [ops.gemm(a, w, out16, bias=bias)])
        both(lambda: [ops.gemm(a, w, out16, bias=bias, gelu=True, ln_stats=stats, ln_csum=csum)])
        hi0 = torch.randn(M, N, device="cuda", generator=g).to(dtype)
        lo0 = torch.randint(1, 256, (M, N), device="cuda", generator=g, dtype=torch.uint8)

        def split():
            hi, lo = hi0.clone(), lo0.clone()
            ops.gemm_split(a, w, hi, lo, bias=bias, in_place=True)
            return [hi, lo]
        both(split)
    a = torch.randn(12500, 1024, device="cuda", generator=g).to(dtype)                # partial last column tile (N = 1000)
    w = (torch.randn(1000, 1024, device="cuda", generator=g) / 32).to(dtype)
    out16 = torch.empty(12500, 1000, device="cuda", dtype=dtype)
    both(lambda: [ops.gemm(a, w, out16)])
