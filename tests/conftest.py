import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_path(name):
    return os.path.join(GOLDEN, name)


def load_json(name):
    with open(golden_path(name)) as f:
        return json.load(f)


def load_npz(name):
    return np.load(golden_path(name))


def make_labels(rs, L, lo, hi, repeat_at=None):
    """Same label recipe as tests/golden/gen_golden.py::make_labels."""
    lab = rs.randint(lo, hi, size=L).astype(np.int64)
    for i in range(1, L):
        while lab[i] == lab[i - 1]:
            lab[i] = rs.randint(lo, hi)
    if repeat_at is not None and 0 < repeat_at < L:
        lab[repeat_at] = lab[repeat_at - 1]
    return lab


def core_inputs(seed, T, L, Vp, scale, repeat_at):
    """Same recipe as tests/golden/gen_golden.py::core_inputs (legacy RandomState is frozen)."""
    rs = np.random.RandomState(seed)
    lp = (-rs.rand(T, Vp) * scale).astype(np.float32)
    ls = (-rs.rand(T, 1) * scale).astype(np.float32)
    label = make_labels(rs, L, 1, Vp + 1, repeat_at)
    return lp, ls, label


def e2e_cases():
    z = load_npz("viterbi_e2e.npz")
    meta = json.loads(bytes(z["meta_json"]).decode())
    for m in meta:
        for b in range(m["B"]):
            yield m, b, z[f"{m['name']}/{b}/em"], z[f"{m['name']}/{b}/label"], z[f"{m['name']}/{b}/seconds"]


@pytest.fixture(scope="session")
def has_gpu():
    import torch
    return torch.cuda.is_available()
