#!/usr/bin/env python3
"""Kernel micro-benchmarks on the config-2 shapes (developer tool; run on the GPU box).
    python tools/kbench.py gemm|attn|gru|fc|all [--iters N]
Interleaved rounds in one process, random operands (cdna guide rules 24/25).
The A/B modes that flip developer switches per launch (gemm --variants 73, persist, q4, q4w, order, epi, attn_ko / attn variants) need
the EXPERIMENT build of the library: bash tools/build_variant.sh lab -DLA_EXPERIMENTS; LA_LIB_PATH=$PWD/ab/lab/liblyricalign_hip.so."""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lyricalignment_amd import _lib, ops


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return t[len(t) // 2], t[0]


def rnd(*shape, dtype=torch.bfloat16, scale=1.0):
    return (torch.randn(*shape, device="cuda") * scale).to(dtype)


def bench_gemm_variants(iters, variants, rounds=5):
    """A/B of main-loop variants (LA_PP_DBG, read per launch) in ONE process: interleaved rounds, random operands, and a race
    screen -- every variant's output must equal variant 0's bit for bit (same k order per accumulator) in every round."""
    M = 48000
    shapes = [("qkv", 3072, 1024, False, False), ("mlp_up+gelu", 4096, 1024, False, False), ("out_proj+res", 1024, 1024, True, True),
              ("mlp_down+res", 1024, 4096, True, True)]
    for name, N, K, f32out, res_ in shapes:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda") if res_ else None
        outs = {v: torch.empty(M, N, device="cuda", dtype=torch.float32 if f32out else torch.bfloat16) for v in variants}

        def run(v):
            if v in (1000, 1001):              # pseudo-variants: default main loop; LA_GELU_PK=1 erfc-form GELU on the packed pipe,
                os.environ["LA_GELU_PK"] = "1" if v == 1000 else "2"       # =2 the sigmoid form one value at a time
            else:
                os.environ.pop("LA_GELU_PK", None)
            os.environ["LA_PP_DBG"] = str(0 if v >= 1000 else v)      # (73 and the pseudo-variants: experiment build, LA_LIB_PATH=ab/lab/...)
            _lib.set_option("gemm_loop", 99 if v == 99 else 0)
            ops.gemm(a, w, outs[v], bias=bias, residual=res, gelu="gelu" in name, out_f32=f32out)

        times = {v: [] for v in variants}
        bad = {v: 0 for v in variants}
        for rd in range(rounds):
            for v in variants:
                med, mn = timeit(lambda: run(v), iters)
                times[v].append(med)
                if v != variants[0] and not torch.equal(outs[v], outs[variants[0]]):
                    bad[v] += 1
                    if rd == 0:
                        d = (outs[v].float() - outs[variants[0]].float()).abs()
                        print(f"   variant {v} vs {variants[0]} on {name}: max abs diff {float(d.max()):.3e}, differing elements {float((d > 0).float().mean()):.4f}", flush=True)
        fl = 2.0 * M * N * K
        for v in variants:
            t = sorted(times[v])
            print(f"gemm {name:14s} N={N} K={K} variant {v:3d}: median {t[len(t)//2]*1e3:8.1f} us  min {t[0]*1e3:8.1f} us  "
                  f"{fl/t[len(t)//2]/1e9:7.1f} TF/s  mismatching rounds vs variant {variants[0]}: {bad[v]}/{rounds}", flush=True)
    os.environ.pop("LA_PP_DBG", None)
    os.environ.pop("LA_GELU_PK", None)
    _lib.set_option("gemm_loop", 0)


def bench_gemm(iters):
    M = 48000
    shapes = [("qkv", 3072, 1024, False), ("mlp_up+gelu", 4096, 1024, False), ("mlp_up_plain", 4096, 1024, False), ("out_proj+res", 1024, 1024, True),
              ("mlp_down+res", 1024, 4096, True), ("gi0", 2304, 1024, True), ("gi1", 2304, 768, True)]
    for name, N, K, f32out in shapes:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device="cuda")
        if f32out:
            res = torch.randn(M, N, device="cuda")
            out = torch.empty(M, N, device="cuda")
            fn = lambda: ops.gemm(a, w, out, bias=bias, residual=res if "res" in name else None, out_f32=True)
        else:
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            fn = lambda: ops.gemm(a, w, out, bias=bias, gelu="gelu" in name)
        if os.environ.get("KB_NOSTORE"):
            from lyricalignment_amd._lib import lib, ptr, stream_ptr
            epi = 1 | {0: 0, 1: 256, 2: 2048}[int(os.environ["KB_NOSTORE"])] | (8 if f32out else 0) | (2 if "gelu" in name else 0)   # 1: no epilogue, 2: no stores
            fn = lambda: lib().la_gemm(1, M, N, K, 1, ptr(a), K, 0, ptr(w), ptr(out), N, 0, ptr(bias), 0, 0, 0, epi, stream_ptr())
        med, mn = timeit(fn, iters)
        fl = 2.0 * M * N * K
        print(f"gemm {name:14s} M={M} N={N} K={K}: median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us  {fl/med/1e9:7.1f} TF/s (min {fl/mn/1e9:7.1f})", flush=True)


def bench_gemm_calib(iters):
    """Calibration against the guide's verified 256x256 8-phase template (cdna_hip_programming.md section 5: ~1320-1340 TF/s at
    4096^3 and ~1470 at 8192^3 on uniform random [-1, 1) operands): the shipped kernel on exactly those problems -- square,
    plain bf16 out, no bias -- next to torch.matmul (hipBLASLt) on the same operands, interleaved rounds in one process; and
    the K = 1024 / 4096 encoder shapes with the same plain epilogue, to separate "the shape" from "the main loop"."""
    shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (48000, 3072, 1024), (48000, 4096, 1024), (48000, 1024, 1024), (48000, 1024, 4096),
              (8192, 8192, 1024)]
    for M, N, K in shapes:
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        wt = w.t()
        fl = 2.0 * M * N * K
        ours, blas = [], []
        for rd in range(3):
            ours.append(timeit(lambda: ops.gemm(a, w, out), iters)[0])
            blas.append(timeit(lambda: torch.matmul(a, wt, out=ref), iters)[0])
        err = float((out.float() - ref.float()).abs().max()) / max(1e-9, float(ref.float().abs().max()))
        o, b = sorted(ours)[1], sorted(blas)[1]
        print(f"calib M={M} N={N} K={K}: this kernel {o*1e3:8.1f} us = {fl/o/1e9:7.1f} TF/s | torch.matmul {b*1e3:8.1f} us = {fl/b/1e9:7.1f} TF/s "
              f"| max rel diff {err:.2e}", flush=True)


def bench_epi_probe(iters):
    """What bounds the residual GEMMs' epilogue (out-proj / MLP-down with the f32 residual stream and its 16-bit copy): the same
    launch with legs of the epilogue's memory traffic left out (LA_EPI_PROBE bits: 1 no residual loads, 2 no f32 store, 4 no 16-bit
    copy), interleaved rounds in one process.  Results of the probe launches are garbage."""
    M = 48000
    for name, N, K in (("out_proj", 1024, 1024), ("mlp_down", 1024, 4096)):
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device="cuda")
        x = torch.randn(M, N, device="cuda")
        h = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        plain = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        res = {}
        for rd in range(3):
            for probe in (0, 1, 2, 4, 6, 7, -1):
                if probe >= 0:
                    os.environ["LA_EPI_PROBE"] = str(probe)
                    fn = lambda: ops.gemm(a, w, x, bias=bias, residual=x, out_f32=True, out16=h)
                else:
                    os.environ.pop("LA_EPI_PROBE", None)
                    fn = lambda: ops.gemm(a, w, plain, bias=bias)
                res.setdefault(probe, []).append(timeit(fn, iters)[0])
        os.environ.pop("LA_EPI_PROBE", None)
        names = {0: "full (f32 residual in place + 16-bit copy)", 1: "no residual loads", 2: "no f32 store", 4: "no 16-bit copy",
                 6: "no stores at all", 7: "no residual loads, no stores", -1: "plain 16-bit out, bias only"}
        for probe, ts in res.items():
            t = sorted(ts)[1]
            print(f"epi-probe {name} N={N} K={K} {names[probe]:45s}: {t*1e3:7.1f} us", flush=True)


def bench_mlpup(iters):
    """Knock-out table of the in-pipeline MLP-up launch gemm_pp_kernel<false, true, bf16, 2> at 48000 x 4096 x 1024 (round-5 verdict: 352 us
    alone with a plain epilogue against 441 us inside the pipeline): the LayerNorm-consumer terms, the GELU and the 16-bit store each left out
    alone and together, interleaved rounds of `iters` back-to-back launches each (the chip stays at its power cap as inside the pipeline), A
    freshly written before every round's first launch as the producer GEMM leaves it.  Experiment build for the store knock-out (LA_EPI_PROBE=2)."""
    M, N, K = 48000, 4096, 1024
    a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
    bias, csum = torch.randn(N, device="cuda"), torch.randn(N, device="cuda") * 0.1
    stats = torch.stack([torch.randn(M, device="cuda") * 0.01, torch.rand(M, device="cuda") + 0.5], dim=1).contiguous()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    lab = _lib.has_experiments()
    forms = [("full: LN fold + bias + GELU + store", dict(ln=True, gelu=True), 0), ("no LN fold (plain bias + GELU)", dict(ln=False, gelu=True), 0),
             ("no GELU (LN fold + bias)", dict(ln=True, gelu=False), 0), ("neither (plain bias)", dict(ln=False, gelu=False), 0)]
    if lab:
        forms += [("full, no 16-bit store", dict(ln=True, gelu=True), 2), ("neither, no store", dict(ln=False, gelu=False), 2)]
    res = {n: [] for n, _, _ in forms}
    for rd in range(5):
        for name, f, probe in forms:
            if probe:
                os.environ["LA_EPI_PROBE"] = str(probe)
            else:
                os.environ.pop("LA_EPI_PROBE", None)
            fn = lambda: ops.gemm(a, w, out, bias=bias, gelu=f["gelu"], ln_stats=stats if f["ln"] else None, ln_csum=csum if f["ln"] else None)
            res[name].append(timeit(fn, iters)[0])
    os.environ.pop("LA_EPI_PROBE", None)
    fl = 2.0 * M * N * K
    for name, ts in res.items():
        t = sorted(ts)[len(ts) // 2]
        print(f"mlp-up {name:40s}: {t*1e3:7.1f} us = {fl/t/1e9:7.1f} TF/s", flush=True)
    if lab:
        for mb in ("0", "32"):
            os.environ["LA_GEMM_MBLOCK"] = mb
            t = timeit(lambda: ops.gemm(a, w, out, bias=bias, gelu=True, ln_stats=stats, ln_csum=csum), iters)[0]
            print(f"mlp-up full, LA_GEMM_MBLOCK={mb}: {t*1e3:7.1f} us", flush=True)
        os.environ.pop("LA_GEMM_MBLOCK", None)


def bench_tile_order(iters):
    """Tile order sweep of the 256x256 kernel (LA_GEMM_GROUP = column tiles per group, LA_GEMM_MBLOCK = row tiles per M block, 0 =
    groups sweep all of M), plain 16-bit epilogue, uniform random operands, interleaved rounds; torch.matmul beside it."""
    shapes = [(48000, 4096, 1024), (48000, 3072, 1024), (48000, 1024, 4096), (8192, 8192, 8192), (4096, 4096, 4096)]
    orders = [(None, None), (4, 8), (4, 4), (8, 4), (2, 16), (4, 16), (8, 8), (16, 2), (4, 32)]
    for M, N, K in shapes:
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        wt = w.t()
        fl = 2.0 * M * N * K
        res = {}
        for rd in range(3):
            for g, mb in orders:
                for k_, v_ in (("LA_GEMM_GROUP", g), ("LA_GEMM_MBLOCK", mb)):
                    if v_ is None: os.environ.pop(k_, None)
                    else: os.environ[k_] = str(v_)
                res.setdefault((g, mb), []).append(timeit(lambda: ops.gemm(a, w, out), iters)[0])
            os.environ.pop("LA_GEMM_GROUP", None); os.environ.pop("LA_GEMM_MBLOCK", None)
            res.setdefault("blas", []).append(timeit(lambda: torch.matmul(a, wt, out=ref), iters)[0])
        for key, ts in res.items():
            t = sorted(ts)[1]
            print(f"order M={M} N={N} K={K} (group, mblock)={key}: {t*1e3:8.1f} us = {fl/t/1e9:7.1f} TF/s", flush=True)


def bench_persist(iters, flag_name="LA_GEMM_PERSIST", label="persistent", on="1"):
    """The persistent 256x256 kernel (opt-in LA_GEMM_PERSIST=1: tickets + next tile's stages prefetched under the epilogue) against one
    workgroup per tile on the encoder's four GEMMs WITH their epilogues, interleaved rounds in one process.  `q4`: the same
    comparison for the four-wave / two-workgroups-per-CU form (LA_GEMM_Q4=1), with a bit-identity check of every case first."""
    M = 48000
    cases = [("qkv (LN consumer)", 3072, 1024, "ln"), ("mlp_up (LN consumer + GELU)", 4096, 1024, "ln_gelu"), ("out_proj (split stream)", 1024, 1024, "split"),
             ("mlp_down (split stream)", 1024, 4096, "split"), ("plain 16-bit", 3072, 1024, "plain")]
    for name, N, K, kind in cases:
        a, w = rnd(M, K), rnd(N, K, scale=K ** -0.5)
        bias = torch.randn(N, device="cuda")
        if kind == "split":
            hi = torch.randn(M, N, device="cuda").bfloat16()
            lo = torch.full((M, N), 128, dtype=torch.uint8, device="cuda")
            fn = lambda: ops.gemm_split(a, w, hi, lo, bias=bias, in_place=True)
        elif kind == "plain":
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            fn = lambda: ops.gemm(a, w, out, bias=bias)
        else:
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            stats = torch.stack([torch.zeros(M, device="cuda"), torch.ones(M, device="cuda")], dim=1).contiguous()
            csum = torch.randn(N, device="cuda")
            fn = lambda: ops.gemm(a, w, out, bias=bias, gelu=kind == "ln_gelu", ln_stats=stats, ln_csum=csum)
        same = None
        if kind != "split":                       # (the in-place split update is checked by tests/test_gpu_ops.py)
            outs = []
            for flag in (on, "0"):
                os.environ[flag_name] = flag
                out.zero_(); fn(); torch.cuda.synchronize()
                outs.append(out.clone())
            same = bool(torch.equal(outs[0], outs[1]))
        res = {on: [], "0": []}
        for rd in range(3):
            for flag in (on, "0"):
                os.environ[flag_name] = flag
                res[flag].append(timeit(fn, iters)[0])
        os.environ.pop(flag_name, None)
        fl = 2.0 * M * N * K
        p_, o_ = sorted(res[on])[1], sorted(res["0"])[1]
        print(f"{label} {name:30s} N={N} K={K}: {label} {p_*1e3:7.1f} us ({fl/p_/1e9:6.1f} TF/s) | one 8-wave workgroup per tile {o_*1e3:7.1f} us ({fl/o_/1e9:6.1f} TF/s) | {100*(o_/p_-1):+.1f} % | identical bits: {same}", flush=True)


def bench_f32emu(iters):
    """Round-5 verdict item 2: float32 Linear products of the fine-tune step (BASELINE configs[2]; fused accum 8 x 2 clips = 24000
    rows) on the 16-bit matrix pipe with float32-level error, PROTOTYPED WITHOUT A NEW KERNEL: each f32 operand is split into 16-bit
    terms, the terms are concatenated along K (small products first) and the existing 256x256 kernel runs one launch with K' = n K,
    f32 accumulate.  Against gemm_kernel<float> (v_mfma_f32_16x16x4_f32, 1/16 of the 16-bit rate) on the same f32 operands; error
    of both against a float64 product on a sample of rows: max |err| / max |ref| and rms(err) / rms(ref).
      bf16x3/6: a = a1 + a2 + a3 (bf16, exact to 2^-24), products (3,1) (2,2) (1,3) (2,1) (1,2) (1,1): 6/16 of the f32 MFMA cost
      f16x2/3 : rows scaled by a power of two to max 2^14, a = a1 + a2 (f16: 22 bits), products (2,1) (1,2) (1,1): 3/16
      f16x2/4 : the same with (2,2): 4/16
    Split / concatenation / row scaling are done by torch OUTSIDE the timing (a product path would fuse them into the producers);
    the time is the GEMM launch alone.  Shapes: forward (NT), input gradient dX = dY W (as NT on W^T) and weight gradient
    dW = dY^T X (as NT on the transposed operands, the 24000-long reduction cut into 4 batch slots so that 192 tiles fill the chip,
    partial sums added afterwards: counted in its time)."""
    M = 24000

    def split_bf16(x, n):
        out, r = [], x.clone()
        for _ in range(n):
            t = r.to(torch.bfloat16)
            out.append(t)
            r = r - t.float()
        return out

    def row_scale(x):                       # power of two s with max |x s| in (2^13, 2^14] per row
        mx = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
        _, e = torch.frexp(mx)              # mx = m 2^e, m in [0.5, 1)
        return torch.ldexp(torch.ones_like(mx), 14 - e)

    def split_f16(x):
        s = row_scale(x)
        xs = x * s
        a1 = xs.to(torch.float16)
        a2 = (xs - a1.float()).to(torch.float16)
        return a1, a2, (1.0 / s).squeeze(1)

    def data(rows, cols, kind, scale=1.0):
        x = torch.randn(rows, cols, device="cuda")
        if kind == "heavy":                 # outlier feature columns (x 30) and a log-normal spread of row magnitudes
            x[:, torch.randperm(cols, device="cuda")[: max(1, cols // 128)]] *= 30.0
            x *= torch.exp(torch.randn(rows, 1, device="cuda") * 1.5)
        return x * scale

    def err(c, ref):
        d = (c.double() - ref)
        return float(d.abs().max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())

    cases = [("fwd qkv", M, 3072, 1024, 1), ("fwd mlp-up", M, 4096, 1024, 1), ("fwd out-proj", M, 1024, 1024, 1), ("fwd mlp-down", M, 1024, 4096, 1),
             ("dX qkv", M, 1024, 3072, 1), ("dX mlp-up", M, 1024, 4096, 1), ("dW qkv", 3072, 1024, 24064, 4), ("dW mlp-down", 1024, 4096, 24064, 4)]
    for kind in ("gauss", "heavy"):
        for name, m, n, k, slots in cases:
            a = data(m, k, kind)
            w = data(n, k, "gauss", k ** -0.5)
            rows = torch.randperm(m, device="cuda")[:512]
            ref = a[rows].double() @ w.double().t()
            res = {}
            # native float32
            out = torch.empty(m, n, device="cuda")
            t = timeit(lambda: ops.gemm(a, w, out), iters)[0]
            res["f32 native"] = (t, *err(out[rows], ref))

            def emu(ap, wp, post=None, label=""):
                """ap / wp: lists of planes in product order (concatenated along K); `slots` > 1 = split-K over batch slots."""
                kc = k // slots
                if slots == 1:
                    A, W = torch.cat(ap, dim=1).contiguous(), torch.cat(wp, dim=1).contiguous()
                    o = torch.empty(m, n, device="cuda")
                    fn = lambda: ops.gemm(A, W, o, out_f32=True)
                else:                            # slot z holds k range z*kc..: [slots][rows][n_terms * kc]
                    A = torch.stack([torch.cat([p_[:, z * kc:(z + 1) * kc] for p_ in ap], dim=1) for z in range(slots)]).contiguous()
                    W = torch.stack([torch.cat([p_[:, z * kc:(z + 1) * kc] for p_ in wp], dim=1) for z in range(slots)]).contiguous()
                    part = torch.empty(slots, m, n, device="cuda")
                    o = torch.empty(m, n, device="cuda")
                    kk = A.shape[2]
                    def fn():
                        from lyricalignment_amd._lib import lib, ptr, stream_ptr, check, dtype_code
                        check(lib().la_gemm_ex(dtype_code(A.dtype), m, n, kk, slots, ptr(A), kk, m * kk, ptr(W), kk, n * kk, ptr(part), n, m * n, None, 8, stream_ptr()), "gemm_ex")
                        torch.sum(part, dim=0, out=o)
                t_ = timeit(fn, iters)[0]
                c = o[rows]
                if post is not None:
                    c = c * post[0][rows][:, None] * post[1][None, :]
                res[label] = (t_, *err(c, ref))

            a1, a2, a3 = split_bf16(a, 3)
            b1, b2, b3 = split_bf16(w, 3)
            emu([a3, a2, a1, a2, a1, a1], [b1, b2, b3, b1, b2, b1], label="bf16x3/6")
            del a3, b3
            fa1, fa2, sa = split_f16(a)
            fb1, fb2, sb = split_f16(w)
            emu([fa2, fa1, fa1], [fb1, fb2, fb1], post=(sa, sb), label="f16x2/3")
            emu([fa2, fa2, fa1, fa1], [fb2, fb1, fb2, fb1], post=(sa, sb), label="f16x2/4")
            base = res["f32 native"][0]
            for lab, (t_, emax, erms) in res.items():
                print(f"f32emu {kind:5s} {name:13s} M={m} N={n} K={k}: {lab:11s} {t_*1e3:8.1f} us ({base/t_:4.2f}x)  max|err|/max|ref| {emax:.2e}  rms {erms:.2e}", flush=True)
            del a, w, a1, a2, b1, b2, fa1, fa2, fb1, fb2
            torch.cuda.empty_cache()


def bench_x2(iters):
    """The shipped f16x2 path (csrc/la_f32x2.hip) on the fine-tune shapes (24000 rows): split kernels (GB/s of their algorithmic bytes:
    4 B read + 4 B written per element; the transposed form reads twice) and the three products of a Linear next to the float32 kernels."""
    from lyricalignment_amd import f32x2, head_train
    M = 24000
    for name, N, K in (("qkv", 3072, 1024), ("mlp-up", 4096, 1024), ("out-proj", 1024, 1024), ("mlp-down", 1024, 4096)):
        x, w, dy = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5, torch.randn(M, N, device="cuda") * 1e-3
        t = timeit(lambda: f32x2.split(x), iters)[0]
        print(f"x2 {name:9s} split    x [{M},{K}]: {t*1e3:7.1f} us  {M*K*8/t/1e6:7.1f} GB/s", flush=True)
        t = timeit(lambda: f32x2.split_t(dy, f32x2.padded_k(N, K, M)), iters)[0]
        print(f"x2 {name:9s} split_t dy [{M},{N}]: {t*1e3:7.1f} us  {M*N*12/t/1e6:7.1f} GB/s", flush=True)
        for lab, fx, fn, fl in (("fwd y=xW^T", lambda: f32x2.linear(x, w), lambda: ops.gemm(x, w), 2.0 * M * N * K),
                                ("dx=dy W   ", lambda: f32x2.gemm_nn(dy, w), lambda: head_train.gemm_nn_f32(dy, w), 2.0 * M * N * K),
                                ("dw=dy^T x ", lambda: f32x2.gemm_tn(dy, x), lambda: head_train.gemm_tn_f32(dy, x), 2.0 * M * N * K)):
            tx, tn = timeit(fx, iters)[0], timeit(fn, iters)[0]
            print(f"x2 {name:9s} {lab}: f16x2 incl. splits {tx*1e3:8.1f} us ({fl/tx/1e9:6.1f} TF/s algorithmic) | float32 kernel {tn*1e3:8.1f} us ({fl/tn/1e9:6.1f}) | {tn/tx:4.2f}x", flush=True)


def bench_x2gemm(iters):
    """la_gemm_f16x2 alone (operands pre-split) on the float32-inference shapes (48000 rows = 32 clips): the kernel's own time and the f16 MFMA
    work it executes (3 x the algorithmic flops), interleaved rounds."""
    from lyricalignment_amd import f32x2
    M = 48000
    shapes = (("qkv", 3072, 1024), ("mlp-up", 4096, 1024), ("out-proj", 1024, 1024), ("mlp-down", 1024, 4096))
    ops_ = []
    for name, N, K in shapes:
        x, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5
        ops_.append((name, N, K, f32x2.split(x, K), f32x2.split(w, K), torch.empty(M, N, device="cuda"), torch.randn(N, device="cuda")))
    res = {n: [] for n, _, _ in shapes}
    for rd in range(3):
        for name, N, K, a, b, out, bias in ops_:
            res[name].append(timeit(lambda: f32x2.gemm(a, b, out=out, bias=bias), iters)[0])
    for name, N, K in shapes:
        t = sorted(res[name])[1]
        fl = 2.0 * M * N * K
        print(f"x2gemm {name:9s} M={M} N={N} K={K}: {t*1e3:8.1f} us  {fl/t/1e9:7.1f} TF/s algorithmic = {3*fl/t/1e9:7.1f} TF/s of f16 MFMA work ({3*fl/t/1e9/2500:.3f} of peak)", flush=True)


def bench_x2order(iters):
    """Tile-order sweep for la_gemm_f16x2 at M = 48000 (experiment build: LA_GEMM_GROUP / LA_GEMM_MBLOCK are read per launch): the planes double
    the bytes of a row panel, so the order tuned for the 16-bit kernel need not be the best one here."""
    from lyricalignment_amd import f32x2
    M = 48000
    for name, N, K in (("qkv", 3072, 1024), ("mlp-up", 4096, 1024), ("out-proj", 1024, 1024), ("mlp-down", 1024, 4096)):
        x, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * K ** -0.5
        a, b, out = f32x2.split(x, K), f32x2.split(w, K), torch.empty(M, N, device="cuda")
        res = {}
        for rd in range(2):
            for g, mb in ((None, None), (2, 32), (4, 16), (4, 64), (8, 32), (8, 16), (16, 32), (4, 8), (2, 16), (16, 0), (4, 0)):
                for k_, v_ in (("LA_GEMM_GROUP", g), ("LA_GEMM_MBLOCK", mb)):
                    if v_ is None: os.environ.pop(k_, None)
                    else: os.environ[k_] = str(v_)
                res.setdefault((g, mb), []).append(timeit(lambda: f32x2.gemm(a, b, out=out), iters)[0])
        os.environ.pop("LA_GEMM_GROUP", None); os.environ.pop("LA_GEMM_MBLOCK", None)
        for key, ts in res.items():
            print(f"x2order {name:9s} (group, mblock)={key}: {min(ts)*1e3:8.1f} us", flush=True)


def bench_attn_bwd(iters):
    """Fused float32 attention backward: the f16x2 sweeps (la_attention_bwd_f16x2) against the float32-MFMA sweeps (la_attention_bwd_f32) --
    time per layer at the fine-tune shape (16 clips x 1500 frames x 16 heads), and max |err| / max |ref| of dq / dk / dv of both against
    torch autograd in float64 on 1 clip x 2 heads."""
    from lyricalignment_amd import encoder_train as et, ops as ops_mod
    shapes = ((1, 1500, 2, True), (16, 1500, 16, False))
    if os.environ.get("KB_BIG_ONLY"):
        shapes = shapes[1:]
    for (B, T, H, check) in shapes:
        d = 64 * H
        g = torch.Generator(device="cuda").manual_seed(3)
        qkv = torch.randn(B * T, 3 * d, device="cuda", generator=g)
        qkv[:, :d] *= 0.35
        datt = torch.randn(B * T, d, device="cuda", generator=g) * torch.exp(torch.randn(B * T, 1, device="cuda", generator=g) * 1.5) * 1e-3
        lse = torch.empty((B, H, T), dtype=torch.float32, device="cuda")
        o = ops.attention_ex(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, T, T, H, lse=lse)
        res = {}
        for name, flag in (("f16x2", True), ("f32  ", False)):
            ops_mod.ATTN_F16X2 = flag
            t = timeit(lambda: et.attention_bwd(qkv, datt, B, T, H, att=o, lse=lse), iters)[0]
            res[name] = (t, et.attention_bwd(qkv, datt, B, T, H, att=o, lse=lse))
        ops_mod.ATTN_F16X2 = True
        line = f"attn bwd B={B} T={T} H={H}: " + " | ".join(f"{n} {t * 1e3:8.1f} us" for n, (t, _) in res.items()) + f" | {res['f32  '][0] / res['f16x2'][0]:.2f}x"
        if check:
            x = qkv.double().requires_grad_(True)
            q, k, v = [t_.view(B, T, H, 64).permute(0, 2, 1, 3) for t_ in x.split(d, dim=1)]
            oo = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, d)
            oo.backward(datt.double())
            for n, (_, got) in res.items():
                errs = [float((got[:, sl].double() - x.grad[:, sl]).abs().max() / x.grad[:, sl].abs().max()) for sl in (slice(0, d), slice(d, 2 * d), slice(2 * d, 3 * d))]
                line += f" | {n} err dq {errs[0]:.2e} dk {errs[1]:.2e} dv {errs[2]:.2e}"
        print(line, flush=True)


def bench_attn_fwd(iters):
    """float32 training forward of the attention (out + lse): the f16x2 kernel (la_attention_lse_f16x2) against the float32-MFMA kernel
    (la_attention_lse_f32): time per layer at the fine-tune shape and max |err| / max |ref| of out, max |err| of lse against float64."""
    from lyricalignment_amd import ops as ops_mod
    shapes = ((1, 1500, 2, True), (16, 1500, 16, False))
    if os.environ.get("KB_BIG_ONLY"):                  # under rocprofv3: per-kernel averages of the fine-tune shape alone
        shapes = shapes[1:]
    for (B, T, H, check) in shapes:
        d = 64 * H
        g = torch.Generator(device="cuda").manual_seed(4)
        qkv = torch.randn(B * T, 3 * d, device="cuda", generator=g)
        qkv[:, :d] *= 0.35
        res = {}
        for name, flag in (("f16x2", True), ("f32  ", False)):
            ops_mod.ATTN_F16X2 = flag
            lse = torch.empty((B, H, T), dtype=torch.float32, device="cuda")
            fn = lambda: ops.attention_ex(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], B, T, T, H, lse=lse)
            t = timeit(fn, iters)[0]
            res[name] = (t, fn().clone(), lse.clone())
        ops_mod.ATTN_F16X2 = True
        line = f"attn fwd B={B} T={T} H={H}: " + " | ".join(f"{n} {t * 1e3:8.1f} us" for n, (t, _, _) in res.items()) + f" | {res['f32  '][0] / res['f16x2'][0]:.2f}x"
        if check:
            x = qkv.double()
            q, k, v = [t_.view(B, T, H, 64).permute(0, 2, 1, 3) for t_ in x.split(d, dim=1)]
            sc = q @ k.transpose(-1, -2)
            ref = (torch.softmax(sc, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, d)
            rl = torch.logsumexp(sc, dim=-1)
            for n, (_, o, l) in res.items():
                line += f" | {n} out {float((o.double() - ref).abs().max() / ref.abs().max()):.2e} lse {float((l.double() - rl).abs().max()):.2e}"
        print(line, flush=True)


def bench_attn(iters):
    B, T, H = 32, 1500, 16
    qkv = rnd(B * T, 3 * H * 64)
    qkv[:, : H * 64] *= float(os.environ.get("KB_QSCALE", "0.125"))   # q pre-scaled by 1/8: scores ~ N(0, 1) like a real layer
    out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
    fl = 4.0 * T * T * H * 64 * B
    ref = None
    for rd in range(3):
        for nw, msum, thr in (("4", False, "0"), ("4", False, "8"), ("4", False, "opt"), ("8", False, "opt"), ("8", False, "8"), ("4", True, "8")):
            os.environ["LA_ATTN_NW"] = nw
            _lib.set_option("attn_nw", int(nw))
            if thr.startswith("opt"):              # the default: optimistic softmax, no per-tile maximum
                os.environ.pop("LA_ATTN_OPT", None)
            else:
                os.environ["LA_ATTN_OPT"] = "0"
            os.environ["LA_ATTN_THR"] = thr
            if msum:
                os.environ["LA_ATTN_MSUM"] = "1"
            else:
                os.environ.pop("LA_ATTN_MSUM", None)
            med, mn = timeit(lambda: ops.attention(qkv, B, T, H, out=out), iters)
            if ref is None:
                ref = out.clone()
            dmax = float((out.float() - ref.float()).abs().max())
            print(f"attention B={B} T={T} H={H} waves/workgroup {nw} msum {int(msum)} defer-thr {thr}: median {med*1e3:.1f} us  min {mn*1e3:.1f} us  {fl/med/1e9:.1f} TF/s  "
                  f"max abs diff to the first run: {dmax:.3e}", flush=True)
    os.environ.pop("LA_ATTN_NW", None); os.environ.pop("LA_ATTN_MSUM", None); os.environ.pop("LA_ATTN_THR", None); os.environ.pop("LA_ATTN_OPT", None)


def bench_attn_knockout(iters):
    """Diagnostic build only (LA_EXTRA_CXXFLAGS=-DLA_ATTN_KNOCKOUT python -m lyricalignment_amd.build): the tile loop with parts
    left out -- 1 exponentials, 2 V^T fragment reads, 4 K fragment reads, 8 staging, 16 MFMAs (bit mask); results are garbage."""
    B, T, H = 32, 1500, 16
    qkv = rnd(B * T, 3 * H * 64)
    qkv[:, : H * 64] *= 0.125
    out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
    names = {0: "full", 1: "no exp", 2: "no V reads", 4: "no K reads", 6: "no K/V reads", 8: "no staging", 14: "no LDS traffic at all",
             16: "no MFMA", 17: "no MFMA, no exp", 7: "no exp, no K/V reads", 15: "no exp, no LDS traffic", 30: "no MFMA, no LDS traffic",
             100: "optimistic softmax, full", 101: "optimistic, no exp", 106: "optimistic, no K/V reads", 108: "optimistic, no staging",
             132: "optimistic, no tile barrier", 140: "optimistic, no tile barrier, no staging", 114: "optimistic, no LDS traffic", 116: "optimistic, no MFMA", 130: "optimistic, no MFMA, no LDS traffic",
             201: "optimistic, S phase at raised priority", 202: "optimistic, S and PV raised", 203: "optimistic, softmax raised"}
    for rd in range(2):
        for ko, nm in names.items():
            os.environ["LA_ATTN_KO"] = str(ko)
            med, mn = timeit(lambda: ops.attention(qkv, B, T, H, out=out), iters)
            print(f"attention knockout {ko:2d} ({nm}): median {med*1e3:.1f} us  min {mn*1e3:.1f} us", flush=True)
    os.environ.pop("LA_ATTN_KO", None)


def bench_attn_occupancy(iters):
    """Diagnostic build (-DLA_ATTN_KNOCKOUT): the shipped attention kernel at 4 / 3 / 2 / 1 workgroups (= waves per SIMD) per CU,
    forced by unused dynamic LDS."""
    B, T, H = 32, 1500, 16
    qkv = rnd(B * T, 3 * H * 64)
    qkv[:, : H * 64] *= 0.125
    out = torch.empty(B * T, H * 64, device="cuda", dtype=torch.bfloat16)
    os.environ["LA_ATTN_KO"] = "100"
    for pad, occ in ((0, 4), (20480, 3), (49152, 2), (102400, 1)):
        os.environ["LA_ATTN_LDSPAD"] = str(pad)
        med, mn = timeit(lambda: ops.attention(qkv, B, T, H, out=out), iters)
        print(f"attention at {occ} waves per SIMD: median {med*1e3:.1f} us  min {mn*1e3:.1f} us", flush=True)
    os.environ.pop("LA_ATTN_KO", None); os.environ.pop("LA_ATTN_LDSPAD", None)


def bench_gru(iters):
    """The persistent GRU recurrence, one layer (T = 1500, H = 384, bf16): the two hand-off forms (option gru_handoff: 0 = data-tagged
    granules, 1 = counter) interleaved in one process, 16 / 32 / 64 clips (KB_GRU_B overrides), bit-identity between the forms."""
    import os
    T, H = 1500, 384
    for B in ([int(os.environ["KB_GRU_B"])] if os.environ.get("KB_GRU_B") else [1, 16, 32, 64]):
        gi = torch.randn(B, T, 2, 3 * H, device="cuda") * 0.5
        w = rnd(2, 3 * H, H, scale=H ** -0.5)
        b = torch.randn(2, 3 * H, device="cuda") * 0.1
        outs = {}
        for rd in range(2):
            for form, name in ((2, "granules"), (1, "counter ")):
                _lib.set_option("gru_handoff", form)
                out = torch.empty(B, T, 2 * H, device="cuda", dtype=torch.bfloat16)
                med, mn = timeit(lambda: ops.gru_layer(gi, w, b, out=out, want_mish=True), max(3, iters // 4))
                outs[form] = out
                print(f"gru layer B={B} T={T} H={H} hand-off {name}: median {med:.2f} ms  ({med*1e3/T:.2f} us/step)", flush=True)
        print(f"   outputs of the two forms identical: {torch.equal(outs[2].view(torch.int16), outs[1].view(torch.int16))}", flush=True)
    _lib.set_option("gru_handoff", 0)


def bench_gru_delay(iters):
    """Sweep of option gru_poll_delay (64-clock sleeps before a step's first poll of the granule hand-off), interleaved rounds."""
    import os
    T, H = 1500, 384
    for B in ([int(os.environ["KB_GRU_B"])] if os.environ.get("KB_GRU_B") else [1, 32]):
        gi = torch.randn(B, T, 2, 3 * H, device="cuda") * 0.5
        w = rnd(2, 3 * H, H, scale=H ** -0.5)
        b = torch.randn(2, 3 * H, device="cuda") * 0.1
        out = torch.empty(B, T, 2 * H, device="cuda", dtype=torch.bfloat16)
        for rd in range(2):
            for d in (0, 2, 4, 6, 8, 10, 12, 16, 20, 28):
                _lib.set_option("gru_poll_delay", d)
                med, mn = timeit(lambda: ops.gru_layer(gi, w, b, out=out, want_mish=True), max(3, iters // 4))
                print(f"gru layer B={B} poll delay {d:2d} x 64 clk: median {med:.2f} ms  ({med*1e3/T:.2f} us/step)", flush=True)
    _lib.set_option("gru_poll_delay", 0)


def bench_fc(iters):
    B, T, K, V = 32, 1500, 768, 21129
    act, w = rnd(B * T, K), rnd(V, K, scale=K ** -0.5)
    bias = torch.randn(V, device="cuda") * 0.1
    labels = torch.randint(2, 403, (B, 26), device="cuda", dtype=torch.int32)
    nl = torch.full((B,), 26, device="cuda", dtype=torch.int32)
    med, mn = timeit(lambda: ops.fc_emissions(act, w, bias, B, T, labels, nl, 1), max(3, iters // 2))
    print(f"fc_emissions: median {med*1e3:.1f} us  {2.0*B*T*K*V/med/1e9:.1f} TF/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--variants", default="", help="gemm: comma-separated LA_PP_DBG values to A/B in one process (first = reference)")
    a = ap.parse_args()
    torch.manual_seed(0)
    if a.what == "gemm" and a.variants:
        bench_gemm_variants(a.iters, [int(v) for v in a.variants.split(",")])
        sys.exit(0)
    if a.what == "persist":
        bench_persist(a.iters)
        sys.exit(0)
    if a.what in ("q4", "q4w"):                                   # 256 x 128 tiles / 128 x 256 tiles
        bench_persist(a.iters, "LA_GEMM_Q4", a.what, "1" if a.what == "q4" else "2")
        sys.exit(0)
    if a.what == "order":
        bench_tile_order(a.iters)
        sys.exit(0)
    if a.what == "mlpup":
        bench_mlpup(a.iters)
        sys.exit(0)
    if a.what == "epi":
        bench_epi_probe(a.iters)
        sys.exit(0)
    if a.what == "attn_fwd":
        bench_attn_fwd(a.iters)
        sys.exit(0)
    if a.what == "attn_bwd":
        bench_attn_bwd(a.iters)
        sys.exit(0)
    if a.what == "x2":
        bench_x2(a.iters)
        sys.exit(0)
    if a.what == "x2gemm":
        bench_x2gemm(a.iters)
        sys.exit(0)
    if a.what == "x2order":
        bench_x2order(a.iters)
        sys.exit(0)
    if a.what == "f32emu":
        bench_f32emu(a.iters)
        sys.exit(0)
    if a.what == "calib":
        bench_gemm_calib(a.iters)
        sys.exit(0)
    if a.what in ("gemm", "all"): bench_gemm(a.iters)
    if a.what in ("attn", "all"): bench_attn(a.iters)
    if a.what == "attn_ko": bench_attn_knockout(a.iters)
    if a.what == "attn_occ": bench_attn_occupancy(a.iters)
    if a.what in ("gru", "all"): bench_gru(a.iters)
    if a.what == "gru_delay": bench_gru_delay(a.iters)
    if a.what in ("fc", "all"): bench_fc(a.iters)
