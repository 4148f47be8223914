#!/usr/bin/env python3
"""Build-time check of the opt-in persistent GEMM kernel (csrc/la_gemm.hip gemm_pp_persist_kernel): its tile ticket is a returning atomic
whose result lands ASYNCHRONOUSLY in the physical register v255 while the main loop runs, so nothing else in the kernel may write v255
between the `global_atomic_add v255, ...` and the `v_mov_b32 ..., v255` that consumes it.  hipcc is not told (it cannot be): this script
compiles the file to assembly (device only, ~3 min) and verifies that every persistent kernel references v255 exactly in those two
instructions and that its SGPR spills live in another VGPR.
    python tools/check_persist_asm.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "lyricalignment_amd", "csrc", "la_gemm.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "la_gemm.s")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on", "-x", "hip", "--cuda-device-only", "-S",
                    "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*persist_kernel.*:\s*;", l)]
ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
bad = 0
for i in starts:
    body = lines[i:min(x for x in ends if x > i)]
    refs = [b.strip() for b in body if re.search(r"\bv255\b", b) or re.search(r"v\[\d+:255\]", b)]
    ok = len(refs) == 2 and refs[0].startswith("global_atomic_add v255") and re.match(r"v_mov_b32 v\d+, v255", refs[1])
    spill = sorted(set(re.findall(r"v_writelane_b32 (v\d+)", "\n".join(body))))
    print(("ok  " if ok and "v255" not in spill else "BAD ") + lines[i].split(":")[0][-60:], refs, "SGPR spills in", spill)
    bad += 0 if ok and "v255" not in spill else 1
sys.exit(1 if bad or not starts else 0)
