"""Drop-in for the reference's utils/CER.py (character / phoneme error rate of the transcript scripts,
evaluate_transcript.py:45-85).  Host-side text scoring, no device work: it lives here because shadowing the reference's
`utils` package shadows `utils.CER` too.

CER(hypothesis, reference) -> (error_rate, counts) with counts = {"N": len(reference), "C": matches, "W": edit distance,
"I": hypothesis symbols with no partner, "D": reference symbols with no partner, "S": substitutions}.
Behaviour kept from utils/CER.py:4-76, including what its traceback does at the borders of the table:
  * edit distance with unit costs; on a mismatch the cheapest of (substitute, drop a hypothesis symbol, drop a reference
    symbol) wins, ties in that order (:21-31);
  * the traceback starts at the bottom-right cell and reads the operation stored there; cells of row 0 / column 0 hold
    "no operation", so the walk leaves the table diagonally and every step taken outside it is booked as one missing
    symbol on the side that still has symbols (:39-65) -- counts therefore need not add up to the distance W;
  * error_rate = W / len(reference) (numpy division: an empty reference gives inf / nan with a warning, as there).
PER (:78-101) maps both strings to pinyin initials + finals with pypinyin and scores those; pypinyin is imported on use.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np

_SUB, _DROP_HYP, _DROP_REF = 1, 2, 3


def _tables(hyp: Sequence, ref: Sequence) -> Tuple[List[List[int]], List[List[int]]]:
    """cost[i][j] = distance between hyp[:i] and ref[:j]; op[i][j] = operation chosen at a mismatch (0 elsewhere)."""
    n_h, n_r = len(hyp), len(ref)
    cost = [[0] * (n_r + 1) for _ in range(n_h + 1)]
    op = [[0] * (n_r + 1) for _ in range(n_h + 1)]
    cost[0] = list(range(n_r + 1))
    for i in range(1, n_h + 1):
        row, above = cost[i], cost[i - 1]
        row[0] = i
        h = hyp[i - 1]
        for j in range(1, n_r + 1):
            if h == ref[j - 1]:
                row[j] = above[j - 1]
                continue
            best, which = above[j - 1], _SUB
            if above[j] < best:
                best, which = above[j], _DROP_HYP
            if row[j - 1] < best:
                best, which = row[j - 1], _DROP_REF
            row[j] = best + 1
            op[i][j] = which
    return cost, op


def CER(hypothesis: Sequence, reference: Sequence) -> Tuple[float, Dict[str, int]]:
    hyp, ref = list(hypothesis), list(reference)
    cost, op = _tables(hyp, ref)
    counts = {"N": len(ref), "C": 0, "W": 0, "I": 0, "D": 0, "S": 0}
    i, j = len(hyp), len(ref)
    while i >= 0 or j >= 0:
        step = op[max(i, 0)][max(j, 0)]
        if step == 0:
            if i >= 1 and j >= 1:
                counts["C"] += 1
            i, j = i - 1, j - 1
        elif step == _DROP_HYP:
            i -= 1
            counts["I"] += 1
        elif step == _DROP_REF:
            j -= 1
            counts["D"] += 1
        else:
            i, j = i - 1, j - 1
            counts["S"] += 1
        if i < 0 <= j:
            counts["D"] += 1
        elif j < 0 <= i:
            counts["I"] += 1
    wrong = np.int16(cost[len(hyp)][len(ref)])          # the reference keeps its table in int16
    counts["W"] = wrong
    return wrong / len(ref) if len(ref) else np.divide(np.float64(wrong), 0.0), counts


def _phonemes(text) -> List[str]:
    try:
        from pypinyin import Style, lazy_pinyin
    except ImportError as e:  # pragma: no cover - pypinyin is not part of this image
        raise ImportError("PER needs the `pypinyin` package (utils/CER.py:2)") from e
    ini = lazy_pinyin(text, style=Style.INITIALS, strict=False)
    fin = lazy_pinyin(text, style=Style.FINALS, strict=False)
    out: List[str] = []
    for a, b in zip(ini, fin):
        out += [a, b]
    return out


def PER(hypothesis, reference):
    return CER(hypothesis=_phonemes(hypothesis), reference=_phonemes(reference))
