"""GPU parity of the HIP alignment DP (through the C ABI) against the oracle and the
reference-generated golden vectors.  Bit-exact: integer frames, float64 scores."""
import json

import numpy as np
import pytest
import torch

from conftest import core_inputs, e2e_cases, load_json, load_npz

pytestmark = pytest.mark.gpu


def _fix_repeats(em, lab):
    """Compact-layout contract: equal neighbouring labels read the same class column, so their
    emission columns are identical (that is what the fused head / emission prep produce)."""
    for n in range(1, len(lab)):
        if lab[n] == lab[n - 1]:
            em[:, 1 + n] = em[:, n]
    return em


def _run(ems, labels_list, T_list=None):
    from lyricalignment_amd import ops
    B = len(ems)
    Lmax = max(max(len(l) for l in labels_list), 1)
    Tmax = max(e.shape[0] for e in ems)
    em = torch.zeros((B, Tmax, Lmax + 1), dtype=torch.float32)
    labels = torch.zeros((B, Lmax), dtype=torch.int32)
    for b, (e, l) in enumerate(zip(ems, labels_list)):
        em[b, : e.shape[0], : e.shape[1]] = torch.from_numpy(np.ascontiguousarray(e))
        labels[b, : len(l)] = torch.tensor(list(l), dtype=torch.int32)
    n_labels = torch.tensor([len(l) for l in labels_list], dtype=torch.int32)
    n_frames = torch.tensor([e.shape[0] for e in ems] if T_list is None else T_list, dtype=torch.int32)
    on, off, score, status = ops.viterbi_batch(em.cuda(), labels.cuda(), n_labels.cuda(), n_frames.cuda())
    torch.cuda.synchronize()
    return on.cpu().numpy(), off.cpu().numpy(), score.cpu().numpy(), status.cpu().numpy()


def test_golden_emissions_bit_exact():
    """Emissions the reference fed its DP -> frames/seconds the reference returned."""
    cases = list(e2e_cases())
    ems = [c[2] for c in cases]
    labs = [c[3].tolist() for c in cases]
    on, off, score, status = _run(ems, labs)
    assert (status == 0).all()
    for b, (m, _, em, label, seconds) in enumerate(cases):
        L = len(label)
        got = [[float(int(a)) * 0.02, float(int(c)) * 0.02] for a, c in zip(on[b, :L], off[b, :L])]
        assert got == seconds.tolist(), m["name"]
        assert (on[b, L:] == -1).all() and (off[b, L:] == -1).all()


@pytest.mark.parametrize("case", load_json("viterbi_core.json")["cases"], ids=lambda c: f"s{c['seed']}_T{c['T']}_L{c['L']}")
def test_against_oracle_on_core_cases(case):
    """Same seeded inputs as the reference's run_viterbi_core fixtures; compares frames and the
    float64 final score with the C oracle (itself pinned to those fixtures bit for bit)."""
    from oracle import alignment_oracle as ao
    lp, ls, label = core_inputs(case["seed"], case["T"], case["L"], case["Vp"], case["scale"], case["repeat_at"])
    em = np.concatenate([ls, lp[:, label - 1]], axis=1).astype(np.float32)
    rc, on_o, off_o, score_o = ao.align_frames(lp, ls, label)
    on, off, score, status = _run([em], [label.tolist()])
    assert status[0] == rc
    if rc == 0:
        assert on[0, : len(label)].tolist() == on_o.tolist()
        assert off[0, : len(label)].tolist() == off_o.tolist()
        assert score[0] == score_o  # float64, same additions in the same order
        assert score[0] == max(case["dp_last"][-1], case["dp_last"][-2]) or case["dp_last"][-1] == case["dp_last"][-2]


def test_ragged_batch_and_errors():
    """Ragged T and L in one launch; empty labels -> LA_EEMPTY (IndexError in the reference),
    too-short utterance -> LA_EINFEASIBLE (ValueError in the reference)."""
    from oracle import alignment_oracle as ao
    rs = np.random.RandomState(7)
    specs = [(50, 3), (5, 4), (4, 4), (200, 31), (1, 1), (30, 0), (333, 17), (2, 1)]
    ems, labs = [], []
    for T, L in specs:
        ems.append((-rs.rand(T, L + 1) * 3).astype(np.float32))
        lab = rs.randint(1, 400, size=L)
        if L == 4:
            lab[2] = lab[1]  # repeat needs one extra frame: T=4 infeasible, T=5 feasible
        _fix_repeats(ems[-1], lab)
        labs.append(lab.tolist())
    on, off, score, status = _run(ems, labs)
    for b, (T, L) in enumerate(specs):
        if L == 0:
            assert status[b] == 3
            continue
        rc, on_o, off_o, score_o = ao.align_frames_compact(ems[b], np.array(labs[b]))
        assert status[b] == rc, (T, L)
        if rc == 0:
            assert on[b, :L].tolist() == on_o.tolist() and off[b, :L].tolist() == off_o.tolist()
            assert score[b] == score_o
    assert status[2] == 2 and status[1] == 0


@pytest.mark.parametrize("T,L", [(700, 40), (1500, 100), (2000, 171), (9000, 238), (600, 500)])
def test_multiwave_lattices_match_oracle(T, L):
    """S > 64 states: multi-wave workgroup with the LDS row exchange; long T spills backpointers to HBM."""
    from oracle import alignment_oracle as ao
    rs = np.random.RandomState(T + L)
    em = (-rs.rand(T, L + 1) * 0.05).astype(np.float32)
    lab = rs.randint(1, 400, size=L)
    lab[L // 2] = lab[L // 2 - 1]
    _fix_repeats(em, lab)
    rc, on_o, off_o, score_o = ao.align_frames_compact(em, lab)
    on, off, score, status = _run([em], [lab.tolist()])
    assert status[0] == rc == 0
    assert on[0, :L].tolist() == on_o.tolist() and off[0, :L].tolist() == off_o.tolist()
    assert score[0] == score_o


@pytest.mark.parametrize("T,L,scale", [(2500, 600, 0.05), (3000, 1100, 1.0), (4500, 2100, 0.0), (9000, 1024, 0.001)])
def test_strip_lattices_beyond_1024_states_match_oracle(T, L, scale):
    """S > 1024 states (whole songs with > 511 characters; the reference's run_viterbi_core has no limit): the strip kernel
    (1024 threads x R consecutive states each, R = 2 / 4 / 8) against the C oracle, bit for bit -- incl. scale 0 (every
    emission equal: pure tie-breaking) and a repeated label."""
    from oracle import alignment_oracle as ao
    rs = np.random.RandomState(T + L)
    em = (-rs.rand(T, L + 1) * scale).astype(np.float32)
    lab = rs.randint(1, 400, size=L)
    lab[L // 2] = lab[L // 2 - 1]
    lab[7] = lab[6]
    _fix_repeats(em, lab)
    rc, on_o, off_o, score_o = ao.align_frames_compact(em, lab)
    on, off, score, status = _run([em], [lab.tolist()])
    assert status[0] == rc == 0
    assert on[0, :L].tolist() == on_o.tolist() and off[0, :L].tolist() == off_o.tolist()
    assert score[0] == score_o


def test_strip_lattice_ragged_batch_and_infeasible():
    """One launch of the strip kernel over utterances of different T and L (one of them too short for its labels:
    LA_EINFEASIBLE like the reference's ValueError, one empty: LA_EEMPTY), checked against the oracle per utterance."""
    from oracle import alignment_oracle as ao
    rs = np.random.RandomState(99)
    specs = [(1800, 700), (900, 30), (650, 640), (1300, 0), (1290, 640), (2000, 513)]
    ems, labs = [], []
    for T, L in specs:
        ems.append((-rs.rand(T, L + 1) * 0.3).astype(np.float32))
        lab = rs.randint(1, 400, size=L)
        for i in range(1, L):                  # (650, 640) with 12 forced repeats needs 652 frames: infeasible
            if i % 53 == 0:
                lab[i] = lab[i - 1]
        _fix_repeats(ems[-1], lab)
        labs.append(lab.tolist())
    on, off, score, status = _run(ems, labs)
    for b, (T, L) in enumerate(specs):
        if L == 0:
            assert status[b] == 3
            continue
        rc, on_o, off_o, score_o = ao.align_frames_compact(ems[b], np.array(labs[b]))
        assert status[b] == rc, (T, L)
        if rc == 0:
            assert on[b, :L].tolist() == on_o.tolist() and off[b, :L].tolist() == off_o.tolist()
            assert score[b] == score_o
            assert (on[b, L:] == -1).all() and (off[b, L:] == -1).all()
    assert status[2] == 2 and status[4] == 0


def test_full_size_properties():
    """BASELINE config-2 size (B=32, T=1500, L<=26): monotone, in-range, contiguous coverage,
    and identical to the oracle on a sample of the batch."""
    from oracle import alignment_oracle as ao
    rs = np.random.RandomState(2)
    B, T = 32, 1500
    Ls = rs.randint(5, 27, size=B)
    ems = [(rs.randn(T, L + 1) * 2 - 3).astype(np.float32) for L in Ls]
    labs = [rs.randint(2, 403, size=L).tolist() for L in Ls]
    for e, l in zip(ems, labs):
        _fix_repeats(e, l)
    on, off, score, status = _run(ems, labs)
    assert (status == 0).all()
    for b in range(B):
        L = Ls[b]
        o, f = on[b, :L], off[b, :L]
        assert (o >= 0).all() and (f <= T).all() and (f > o).all()
        assert (o[1:] >= f[:-1]).all()  # label segments never overlap
    for b in (0, 7, 31):
        rc, on_o, off_o, score_o = ao.align_frames_compact(ems[b], np.array(labs[b]))
        assert on[b, : Ls[b]].tolist() == on_o.tolist() and off[b, : Ls[b]].tolist() == off_o.tolist() and score[b] == score_o


def test_no_dpp_variant_matches(monkeypatch):
    """The single-wave LDS-exchange build of the kernel (LA_VITERBI_NO_DPP) is checked in a child process."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from test_gpu_viterbi import _run
        from oracle import alignment_oracle as ao
        rs = np.random.RandomState(5)
        em = (-rs.rand(400, 20) * 0.01).astype(np.float32); lab = rs.randint(1, 400, size=19)
        rc, on_o, off_o, sc = ao.align_frames_compact(em, lab)
        on, off, score, status = _run([em], [lab.tolist()])
        assert status[0] == 0 and on[0, :19].tolist() == on_o.tolist() and off[0, :19].tolist() == off_o.tolist() and score[0] == sc
        print("ok")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LA_VITERBI_NO_DPP="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("variant", ["ctc", "plain"])
def test_emissions_from_logits(variant):
    """la_emissions_from_logits against the reference's own emission prep (golden) and the oracle."""
    from lyricalignment_amd import ops
    from oracle import model_oracle as mo
    z = load_npz("emission_prep.npz")
    logits = torch.from_numpy(z[f"{variant}/logits"])
    labels = torch.tensor([[1, 2, 3], [4, 5, 0]], dtype=torch.int32)
    n_labels = torch.tensor([3, 2], dtype=torch.int32)
    var = 1 if variant == "ctc" else 0
    em = ops.emissions_from_logits(logits.cuda(), labels.cuda(), n_labels.cuda(), var).cpu().numpy()
    lp, ls = z[f"{variant}/lp"], z[f"{variant}/ls"]
    for b, L in enumerate([3, 2]):
        np.testing.assert_allclose(em[b, :, 0], ls[b, :, 0], rtol=0, atol=1e-5)
        for n in range(L):
            np.testing.assert_allclose(em[b, :, 1 + n], lp[b, :, int(labels[b, n]) - 1], rtol=0, atol=1e-5)
    assert np.isfinite(em[0]).all()
    # big vocabulary, random rows, vs oracle (tolerance 1e-4: 21127-term fp32 sums in a different order)
    rs = np.random.RandomState(3)
    V = 21129
    big = torch.from_numpy((rs.randn(2, 37, V) * 2).astype(np.float32))
    lab = torch.from_numpy(rs.randint(1, 403, size=(2, 26)).astype(np.int32))
    nl = torch.tensor([26, 11], dtype=torch.int32)
    em = ops.emissions_from_logits(big.cuda(), lab.cuda(), nl.cuda(), var).cpu()
    lp_o, ls_o = (mo.emission_prep_ctc if variant == "ctc" else mo.emission_prep_plain)(big)
    for b, L in enumerate([26, 11]):
        np.testing.assert_allclose(em[b, :, 0].numpy(), ls_o[b, :, 0].numpy(), rtol=0, atol=1e-4)
        idx = (lab[b, :L].long() - 1)
        np.testing.assert_allclose(em[b, :, 1:1 + L].numpy(), lp_o[b][:, idx].numpy(), rtol=0, atol=1e-4)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_ragged_batches_bit_exact(seed):
    """Randomised sweep: 48 utterances per launch with ragged T (1..1700) and L (1..70), emission scales from 'all ties'
    (0) to well separated, quantised emissions (many exact ties), repeated labels, infeasible ones (T < L) -- frames, float64
    scores and status codes equal to the C oracle for every utterance."""
    from oracle import alignment_oracle as ao
    rs = np.random.RandomState(1000 + seed)
    ems, labs = [], []
    for _ in range(48):
        L = int(rs.randint(1, 71))
        T = int(rs.randint(1, 1701)) if rs.rand() > 0.15 else int(rs.randint(1, 2 * L + 2))
        scale = [0.0, 1e-3, 0.05, 1.0, 3.0][int(rs.randint(0, 5))]
        em = (-rs.rand(T, L + 1) * scale).astype(np.float32)
        if rs.rand() < 0.3:
            em = np.round(em * 4) / 4                       # quarter-step values: exact ties everywhere
        lab = rs.randint(1, 400, size=L)
        for _ in range(int(rs.randint(0, 4))):
            if L > 1:
                j = int(rs.randint(1, L)); lab[j] = lab[j - 1]
        _fix_repeats(em, lab)
        ems.append(em); labs.append(lab.tolist())
    on, off, score, status = _run(ems, labs)
    n_bad = 0
    for b, (em, lab) in enumerate(zip(ems, labs)):
        rc, on_o, off_o, score_o = ao.align_frames_compact(em, np.array(lab))
        assert status[b] == rc, (b, status[b], rc)
        if rc == 0:
            assert on[b, : len(lab)].tolist() == on_o.tolist() and off[b, : len(lab)].tolist() == off_o.tolist(), b
            assert score[b] == score_o
        else:
            n_bad += 1
    assert n_bad < 48
