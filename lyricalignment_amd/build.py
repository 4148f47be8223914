"""Build liblyricalign_hip.so (gfx950) in-tree with hipcc.

    python -m lyricalignment_amd.build [--force]

hipcc cross-compiles without a GPU; the .so sits next to this file so it
travels to the GPU box with the repo snapshot (it is git-ignored).
"""
from __future__ import annotations

import concurrent.futures
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# LA_BUILD_VARIANT=<name> (tools/build_variant.sh): a second library ab/<name>/liblyricalign_hip.so with its own object directory,
# compiled with LA_EXTRA_CXXFLAGS; with -DLA_EXPERIMENTS among them the sources under csrc/lab/ (the measured-slower kernel
# structures of rounds 2-4 and their developer switches) are built in.  The default build never compiles csrc/lab/.
VARIANT = os.environ.get("LA_BUILD_VARIANT", "")
OBJ = os.path.join(CSRC, "_obj" + ("_" + VARIANT if VARIANT else ""))
LIB = (os.path.join(HERE, "..", "ab", VARIANT, "liblyricalign_hip.so") if VARIANT else os.path.join(HERE, "liblyricalign_hip.so"))
ARCH = "gfx950"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-fno-fast-math", "-ffp-contract=on"] + os.environ.get("LA_EXTRA_CXXFLAGS", "").split()
# every object is stamped with the flags it was compiled with: an A/B build (LA_EXTRA_CXXFLAGS=-D...) never reuses a stale one
FLAGS_TAG = hashlib.sha256(" ".join([HIPCC] + CXXFLAGS).encode()).hexdigest()[:12]

# Kernels whose main loop is a hand-placed `asm volatile` stream (la_gemm_pp.h): the fragment registers are written by one asm
# statement and waited for in another ~27 MFMAs later, so a register-allocator spill or copy between the two would read stale
# data without any error.  The build fails if one of them has spilled registers or scratch (round-2 advisor finding: the
# LayerNorm-consumer instantiations sat at 256 VGPRs with 107-127 spills).  The one-wave-per-SIMD experiment (LA_PP_DBG=73) is
# only reported: its spills are in the hipcc-scheduled epilogue, after the loop's last wait.
NO_SPILL_KERNELS = re.compile(r"gemm_pp_kernel|gemm_pp_persist_kernel|gemm_q4_kernel|fc_lse_pp_kernel|fc_lse_x2_kernel")
REPORT_KERNELS = re.compile(r"gemm_mono_kernel")


EXPERIMENTS = "-DLA_EXPERIMENTS" in CXXFLAGS


def _sources():
    srcs = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp"))
    if EXPERIMENTS:
        srcs += glob.glob(os.path.join(CSRC, "lab", "*.hip"))
    return sorted(srcs)


def _headers():
    return (glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "lab", "*.h")) +
            glob.glob(os.path.join(HERE, "..", "include", "*.h")))


def _parse_resource_usage(stderr: str) -> dict:
    """-Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {"VGPRs": n, "VGPRs Spill": n, ...}}."""
    usage, cur = {}, None
    for line in stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = usage.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass-analysis", line)
        if m and cur is not None:
            try:
                cur[m.group(1).strip()] = int(m.group(2))
            except ValueError:
                cur[m.group(1).strip()] = m.group(2)
    return usage


def check_spills(usage: dict, src: str) -> None:
    bad = []
    for name, u in usage.items():
        # (SGPR spills go to lanes of a reserved VGPR by v_writelane -- no memory, nothing asynchronous: not a hazard)
        spilled = u.get("VGPRs Spill", 0) or u.get("ScratchSize [bytes/lane]", 0)
        if spilled and NO_SPILL_KERNELS.search(name):
            bad.append(f"{name}: VGPRs {u.get('VGPRs')}, spilled {u.get('VGPRs Spill')}, scratch {u.get('ScratchSize [bytes/lane]')} B/lane")
        elif spilled and REPORT_KERNELS.search(name):
            sys.stderr.write(f"[build] note: {name} spills {u.get('VGPRs Spill')} VGPRs (developer variant, not gated)\n")
    if bad and os.environ.get("LA_ALLOW_SPILLS") == "1":          # developer experiments only
        sys.stderr.write("[build] SPILLS (allowed by LA_ALLOW_SPILLS=1):\n  " + "\n  ".join(bad) + "\n")
    elif bad:
        raise RuntimeError(f"{os.path.basename(src)}: hand-placed-loop kernels must not spill (stale-fragment hazard):\n  " + "\n  ".join(bad))


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    meta = obj + ".json"                                  # flags tag + per-kernel register usage of the compile that made obj
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in _headers()])
    if not force and os.path.exists(obj) and os.path.exists(meta) and os.path.getmtime(obj) >= newest:
        try:
            with open(meta) as f:
                m = json.load(f)
            if m.get("flags") == FLAGS_TAG:
                check_spills(m.get("usage", {}), src)
                return obj
        except (OSError, ValueError):
            pass
    cmd = [HIPCC, *CXXFLAGS, "-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    usage = _parse_resource_usage(r.stderr)
    rest = "\n".join(l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l and not re.match(r"^\s*(\d+ \||\|)", l))
    if rest.strip():
        sys.stderr.write(rest + "\n")
    with open(meta, "w") as f:
        json.dump({"flags": FLAGS_TAG, "usage": usage}, f)
    check_spills(usage, src)
    return obj


def build(force: bool = False, jobs: int | None = None) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(os.path.abspath(LIB)), exist_ok=True)
    srcs = _sources()
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if VARIANT:
        return os.path.abspath(LIB)
    try:                                   # the C++ consumer example is a check of the header, not a part of the library:
        build_example(force)               # tests/test_gpu_surface.py asserts that it builds, links and runs
    except RuntimeError as e:
        sys.stderr.write(f"[build] warning: {e}\n")
    return LIB


EXAMPLE_SRC = os.path.join(HERE, "..", "examples", "align_capi.cpp")
EXAMPLE_BIN = os.path.join(HERE, "..", "examples", "_bin", "align_capi")


def build_example(force: bool = False) -> str:
    """examples/align_capi.cpp: the C ABI's model-level entry points driven from a C++ program (no Python, no torch)."""
    src, out = os.path.abspath(EXAMPLE_SRC), os.path.abspath(EXAMPLE_BIN)
    if not os.path.exists(src):
        return ""
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(LIB)):
        return out
    cmd = [HIPCC, f"--offload-arch={ARCH}", "-O2", "-std=c++17", src, "-I", os.path.join(HERE, "..", "include"), "-L", HERE,
           "-llyricalign_hip", "-Wl,-rpath,$ORIGIN/../../lyricalignment_amd", "-Wl,-rpath," + HERE, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"example build failed:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
