#!/usr/bin/env python3
"""Diagnostic (library built with LA_EXTRA_CXXFLAGS=-DLA_TILE_STAMPS): where a 256x256 GEMM tile's lifetime goes on its CU and how
long the CU waits for its next workgroup.  Every workgroup of one launch leaves (wall clock at entry, after the prologue, after the
main loop, at the end; HW_ID; XCC_ID); workgroups are grouped by CU and ordered in time.
    python tools/tile_timeline.py [qkv|mlp_up|out_proj|mlp_down]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lyricalignment_amd import _lib, ops

M = 48000
shapes = {"qkv": (3072, 1024, "ln"), "mlp_up": (4096, 1024, "ln_gelu"), "out_proj": (1024, 1024, "split"), "mlp_down": (1024, 4096, "split"),
          "plain": (3072, 1024, "plain")}
for name in (sys.argv[1:] or ["qkv", "mlp_up", "out_proj", "mlp_down"]):
    N, K, kind = shapes[name]
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    q4 = os.environ.get("LA_GEMM_Q4", "0")                   # four-wave workgroups, two per CU: 256 x 128 (1) or 128 x 256 (2) tiles
    tiles = -(-M // 256) * -(-N // 256) if q4 not in ("1", "2") else (-(-M // 256) * -(-N // 128) if q4 == "1" else -(-M // 128) * -(-N // 256))
    buf = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
    _lib.lib().la_debug_set_tile_stamps.argtypes = [ctypes.c_void_p]
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
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    _lib.check(_lib.lib().la_debug_set_tile_stamps(buf.data_ptr()), "set_tile_stamps")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(); fn(); ev1.record()
    torch.cuda.synchronize()
    _lib.lib().la_debug_set_tile_stamps(0)
    s = buf.cpu().numpy().reshape(tiles, 8)
    if os.environ.get("LA_GEMM_PERSIST") == "1":
        # persistent kernel: one record per TILE: iteration top, first fragments in, main loop done, epilogue end, HW_ID, XCC_ID,
        # epilogue begin, prefetched flag
        tt = s[:, [0, 1, 2, 6, 3]].astype(np.float64) * 0.01
        tt -= tt[:, 0].min()
        cu = (s[:, 5] & 0xF) * 65536 + ((s[:, 4] >> 8) & 0xFF)
        pf = s[:, 7] == 1
        gaps = []
        for c in np.unique(cu):
            idx = np.where(cu == c)[0]
            idx = idx[np.argsort(tt[idx, 0])]
            gaps += list(tt[idx[1:], 0] - tt[idx[:-1], 4])
        q = lambda x: f"median {np.median(x):6.2f}  p10 {np.percentile(x, 10):6.2f}  p90 {np.percentile(x, 90):6.2f}"
        print(f"{name} (persistent): N={N} K={K} {kind}: launch {ev0.elapsed_time(ev1) * 1e3:.1f} us, {tiles} tiles on {len(np.unique(cu))} CUs, "
              f"{int(pf.sum())} tiles with prefetched stages, last ends at {tt[:, 4].max():.1f} us")
        for sel, lab in ((pf, "prefetched"), (~pf, "not prefetched")):
            if sel.sum() == 0:
                continue
            print(f"   [{lab}] entry (top -> first fragments)      {q(tt[sel, 1] - tt[sel, 0])} us")
            print(f"   [{lab}] main loop                          {q(tt[sel, 2] - tt[sel, 1])} us")
            print(f"   [{lab}] ticket + next tile's stages issued {q(tt[sel, 3] - tt[sel, 2])} us")
            print(f"   [{lab}] epilogue                           {q(tt[sel, 4] - tt[sel, 3])} us")
        print(f"   epilogue end -> next iteration top        {q(np.array(gaps))} us (n = {len(gaps)})")
        print(f"   tile period                               {np.median(tt[:, 4] - tt[:, 0]) + np.median(gaps):6.2f} us", flush=True)
        continue
    t = s[:, :4].astype(np.float64) * 0.01                      # 100 MHz ticks -> us
    t -= t[:, 0].min()
    cu = (s[:, 5] & 0xF) * 65536 + ((s[:, 4] >> 8) & 0xFF)    # (XCC, SE / SH / CU bits of HW_ID)
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    if q4 in ("1", "2"):
        # two workgroups share a CU: per CU, the share of its busy span with 0 / 1 / 2 workgroups inside their main loops
        q = lambda x: f"median {np.median(x):6.2f}  p10 {np.percentile(x, 10):6.2f}  p90 {np.percentile(x, 90):6.2f}"
        share = np.zeros(3)
        for c in np.unique(cu):
            idx = np.where(cu == c)[0]
            ev = sorted([(t[i, 1], 1) for i in idx] + [(t[i, 2], -1) for i in idx])
            lo, hi, n, last = t[idx, 0].min(), t[idx, 3].max(), 0, None
            last = lo
            for when, d in ev:
                share[min(n, 2)] += when - last
                last, n = when, n + d
            share[0] += hi - last
        share /= share.sum()
        print(f"{name} (q4={q4}): N={N} K={K} {kind}: launch {ev0.elapsed_time(ev1) * 1e3:.1f} us, {tiles} tiles on {len(np.unique(cu))} CUs, last workgroup ends at {t[:, 3].max():.1f} us")
        print(f"   prologue (entry -> stages 0, 1 in)  {q(pro)} us")
        print(f"   main loop                           {q(loop)} us")
        print(f"   epilogue                            {q(epi)} us")
        print(f"   CU time with 0 / 1 / 2 workgroups in their main loops: {share[0]:.2f} / {share[1]:.2f} / {share[2]:.2f}", flush=True)
        continue
    gaps, first = [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(t[idx, 0])]
        first.append(t[idx[0], 0])
        gaps += list(t[idx[1:], 0] - t[idx[:-1], 3])
    gaps = np.array(gaps)
    print(f"{name}: N={N} K={K} {kind}: launch {ev0.elapsed_time(ev1) * 1e3:.1f} us, {tiles} tiles on {len(np.unique(cu))} CUs "
          f"({tiles / len(np.unique(cu)):.2f} rounds), last workgroup ends at {t[:, 3].max():.1f} us")
    q = lambda x: f"median {np.median(x):6.2f}  p10 {np.percentile(x, 10):6.2f}  p90 {np.percentile(x, 90):6.2f}"
    print(f"   first workgroup of a CU starts at   {q(np.array(first))} us")
    print(f"   prologue (entry -> stages 0, 1 in)  {q(pro)} us")
    print(f"   main loop                           {q(loop)} us")
    print(f"   epilogue                            {q(epi)} us")
    print(f"   gap: end of one -> entry of next WG {q(gaps)} us  (n = {len(gaps)})")
    print(f"   tile lifetime + gap                 {np.median(t[:, 3] - t[:, 0]) + np.median(gaps):6.2f} us", flush=True)
