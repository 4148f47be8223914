"""oracle/viterbi_python.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The lattice recurrence of utils/alignment.py:73-119 (run_viterbi_core) as plain Python loops over NumPy arrays: what the
reference executes per (frame, state) cell when numba is not installed and its @jit decorator is a no-op (numba is an
optional accelerator there: requirements.txt lists it, utils/alignment.py:1-10 imports it).  bench.py's cpu_baseline times
this on one 30 s clip as the UPPER bound of the reference's DP cost; oracle/viterbi_oracle.c is the
compiled restatement the parity tests use.  Pinned to that C oracle (and through it to the reference's own dp / bt hashes) by
tests/test_oracle_viterbi.py::test_python_loop_dp_equals_the_c_oracle.

States: 0 = leading blank, 2 i + 1 = label i, 2 i + 2 = the blank after label i.  A blank state is entered from itself or from the
label before it; a label state from itself, from the blank before it, or -- when the label differs from the previous one --
straight from the previous label (that jump wins ties against both others; "stay" must be strictly better than "advance").
"""
from __future__ import annotations

import numpy as np

NEG = -10000000.0      # utils/alignment.py:144


def viterbi_lattice(lp: np.ndarray, ls: np.ndarray, label) -> tuple:
    """lp [T, V'] float32 label log-probs, ls [T, 1] float32 silence log-probs, label [L] (1-based class ids)
    -> (dp [T, S] float64, bt [T, S] int64) exactly as the reference fills them (:144-151 initialisation, :73-119 recurrence)."""
    T, L = int(lp.shape[0]), int(len(label))
    S = 2 * L + 1
    dp = np.full((T, S), NEG, dtype=np.float64)
    bt = np.zeros((T, S), dtype=np.int64)
    dp[0][0] = ls[0][0]
    dp[0][1] = lp[0][label[0] - 1]
    col = [0] + [0 if s % 2 == 0 else int(label[s // 2]) - 1 for s in range(1, S)]              # emission column of a label state
    may_skip = [False] * S
    for s in range(3, S, 2):
        may_skip[s] = bool(label[s // 2] != label[s // 2 - 1])
    # Every cell reads and writes the 2-D arrays element by element (dp[t - 1][s], lp[t][c]: a NumPy row view + a NumPy scalar per
    # access), as the reference's un-jitted loop body does -- hoisting the rows into locals would time a faster program than the
    # reference runs.
    for t in range(1, T):
        bt[t][0] = 0
        dp[t][0] = dp[t - 1][0] + ls[t][0]
        for s in range(1, S):
            emit = ls[t][0] if s % 2 == 0 else lp[t][col[s]]
            if may_skip[s] and dp[t - 1][s - 2] >= dp[t - 1][s - 1] and dp[t - 1][s - 2] >= dp[t - 1][s]:
                bt[t][s] = s - 2
                dp[t][s] = dp[t - 1][s - 2] + emit
            elif dp[t - 1][s] > dp[t - 1][s - 1]:
                bt[t][s] = s
                dp[t][s] = dp[t - 1][s] + emit
            else:
                bt[t][s] = s - 1
                dp[t][s] = dp[t - 1][s - 1] + emit
    return dp, bt
