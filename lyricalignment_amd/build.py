"""Build liblyricalign_hip.so (gfx950) in-tree with hipcc.

    python -m lyricalignment_amd.build [--force]

hipcc cross-compiles without a GPU; the .so sits next to this file so it
travels to the GPU box with the repo snapshot (it is git-ignored).
"""
from __future__ import annotations

import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "liblyricalign_hip.so")
ARCH = "gfx950"
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-fno-fast-math", "-ffp-contract=on"] + os.environ.get("LA_EXTRA_CXXFLAGS", "").split()


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    newest = max([os.path.getmtime(src)] + [os.path.getmtime(h) for h in _headers()])
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj
    cmd = [HIPCC, *CXXFLAGS, "-x", "hip", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, jobs: int | None = None) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    build_example(force)
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
