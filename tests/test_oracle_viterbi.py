"""Pins oracle/viterbi_oracle.c + oracle/alignment_oracle.py against golden
vectors produced by the reference's own utils/alignment.py (CPU only)."""
import hashlib
import json

import numpy as np
import pytest
import torch

from conftest import core_inputs, e2e_cases, load_json, load_npz
from oracle import alignment_oracle as ao
from oracle import model_oracle as mo


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("case", load_json("viterbi_core.json")["cases"], ids=lambda c: f"s{c['seed']}_T{c['T']}_L{c['L']}")
def test_core_matches_reference_bit_exact(case):
    lp, ls, label = core_inputs(case["seed"], case["T"], case["L"], case["Vp"], case["scale"], case["repeat_at"])
    assert label.tolist() == case["label"]
    T, S = case["T"], 2 * case["L"] + 1
    dp = np.full((T, S), -10000000.0, dtype=np.float64)
    bt = np.zeros((T, S), dtype=np.int64)
    dp[0][0] = ls[0][0]
    dp[0][1] = lp[0][label[0] - 1]
    ao.run_viterbi_core(dp, bt, lp, ls, label)
    assert sha(bt) == case["bt_sha256"]
    assert sha(dp) == case["dp_sha256"]
    assert dp[-1].tolist() == case["dp_last"]


@pytest.mark.parametrize("m,b,em,label,seconds", list(e2e_cases()), ids=lambda v: v["name"] if isinstance(v, dict) else None)
def test_backtrace_on_reference_emissions(m, b, em, label, seconds):
    """Emissions are the ones the reference fed its DP; seconds are what it returned."""
    rc, on, off, _ = ao.align_frames_compact(em, label)
    assert rc == 0
    got = np.array([[float(int(a)) * 0.02, float(int(c)) * 0.02] for a, c in zip(on, off)])
    assert got.tolist() == seconds.tolist()  # float(frame)*hop is bit-exact, e.g. 83 -> 1.6600000000000001
    # and the non-compact entry point agrees
    L = len(label)
    lp = np.ascontiguousarray(em[:, 1:1 + L])
    lab2 = np.arange(1, L + 1)
    for n in range(1, L):
        if label[n] == label[n - 1]:
            lab2[n] = lab2[n - 1]
    rc2, on2, off2, _ = ao.align_frames(lp, em[:, 0], lab2)
    assert rc2 == 0 and on2.tolist() == on.tolist() and off2.tolist() == off.tolist()


def test_e2e_from_logits_high_contrast():
    """Whole perform_viterbi(_ctc) restatement from seeded logits (peaked cases only:
    emission prep is fp32 torch on whatever CPU runs this)."""
    z = load_npz("viterbi_e2e.npz")
    meta = json.loads(bytes(z["meta_json"]).decode())
    for m in meta:
        if m["scale"] < 1.0:
            continue
        rs = np.random.RandomState(m["seed"])
        logits = torch.from_numpy((rs.randn(m["B"], m["T"], m["V"]) * m["scale"]).astype(np.float32))
        labels = torch.tensor(m["labels"])
        fn = ao.perform_viterbi_ctc if m["variant"] == "ctc" else ao.perform_viterbi
        res = fn(logits, labels)
        for b in range(m["B"]):
            assert res[b] == z[f"{m['name']}/{b}/seconds"].tolist(), m["name"]


def test_error_behaviour_matches_reference():
    for e in load_json("viterbi_errors.json"):
        lg = torch.from_numpy(np.random.RandomState(e["seed"]).randn(1, e["T"], e["V"]).astype(np.float32))
        labels = torch.tensor(e["labels"])
        if e["raises"] is None:
            assert ao.perform_viterbi_ctc(lg, labels) == e["result"]
        else:
            exc = {"ValueError": ValueError, "IndexError": IndexError}[e["raises"]]
            with pytest.raises(exc):
                ao.perform_viterbi_ctc(lg, labels)


@pytest.mark.parametrize("variant", ["ctc", "plain"])
def test_emission_prep(variant):
    z = load_npz("emission_prep.npz")
    logits = torch.from_numpy(z[f"{variant}/logits"])
    lp, ls = (mo.emission_prep_ctc if variant == "ctc" else mo.emission_prep_plain)(logits)
    np.testing.assert_allclose(lp.numpy(), z[f"{variant}/lp"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(ls.numpy(), z[f"{variant}/ls"], rtol=0, atol=1e-5)
    assert (lp.numpy() == -1000).sum() == (z[f"{variant}/lp"] == -1000).sum()  # saturation -> clip, not -inf
    assert np.isfinite(lp.numpy()).all() and np.isfinite(ls.numpy()).all()


def test_get_mae():
    g = load_json("mae.json")
    assert ao.get_mae(g["gt"], g["predict"]) == g["mae"]


def test_python_loop_dp_equals_the_c_oracle():
    """oracle/viterbi_python.py (the recurrence as plain Python loops: what the reference runs per cell without numba; bench.py times
    it as the DP's upper bound) fills dp / bt exactly as the compiled oracle does -- which is pinned to the reference's own matrices
    above -- on lattices with repeated labels, a single label, and more states than frames can reach."""
    from oracle import alignment_oracle as ao, viterbi_python as vp
    for seed, T, V, L, rep in ((0, 400, 60, 19, 5), (1, 50, 7, 1, None), (2, 30, 40, 26, 3), (3, 1500, 300, 26, 11)):
        lp, ls, lab = core_inputs(seed, T, L, V, 3.0, rep)
        dp = np.full((T, 2 * L + 1), -10000000.0)
        bt = np.zeros((T, 2 * L + 1), dtype=np.int64)
        dp[0][0] = ls[0][0]
        dp[0][1] = lp[0][lab[0] - 1]
        ao.run_viterbi_core(dp, bt, lp, ls, lab)
        d2, b2 = vp.viterbi_lattice(lp, ls, lab)
        assert np.array_equal(dp, d2) and np.array_equal(bt, b2)
