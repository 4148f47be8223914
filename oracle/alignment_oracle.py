"""oracle/alignment_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python face of oracle/viterbi_oracle.c plus the emission prep, i.e. a CPU
restatement of utils/alignment.py:13-71 (perform_viterbi) and :121-188
(perform_viterbi_ctc) of the reference.  Parity status: PINNED (see the header
of viterbi_oracle.c and tests/test_oracle_viterbi.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Sequence

import numpy as np
import torch

from . import model_oracle

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libla_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """gcc-compile viterbi_oracle.c -> oracle/_build/libla_oracle.so."""
    src = os.path.join(_HERE, "viterbi_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        c_i64, c_p = ctypes.c_int64, ctypes.c_void_p
        L.la_oracle_viterbi_core.argtypes = [c_p, c_p, c_p, c_i64, c_p, c_p, c_i64, c_i64]
        L.la_oracle_viterbi_core.restype = None
        L.la_oracle_align.argtypes = [c_p, c_i64, c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_p]
        L.la_oracle_align.restype = ctypes.c_int
        L.la_oracle_align_compact.argtypes = [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p, c_p]
        L.la_oracle_align_compact.restype = ctypes.c_int
        _lib = L
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def run_viterbi_core(dp, bt, lp, ls, label):
    """Same contract as the reference's run_viterbi_core (in place + returned)."""
    assert dp.dtype == np.float64 and bt.dtype == np.int64 and dp.flags.c_contiguous and bt.flags.c_contiguous
    lp = np.ascontiguousarray(lp, dtype=np.float32)
    ls = np.ascontiguousarray(ls, dtype=np.float32).reshape(-1)
    label = np.ascontiguousarray(label, dtype=np.int64)
    lib().la_oracle_viterbi_core(_ptr(dp), _ptr(bt), _ptr(lp), lp.shape[1], _ptr(ls), _ptr(label),
                                 lp.shape[0], label.shape[0])
    return dp, bt


def align_frames(lp: np.ndarray, ls: np.ndarray, label: Sequence[int], want_path: bool = False):
    """One utterance: emissions -> (status, onset[L], offset[L], final_score[, path])."""
    lp = np.ascontiguousarray(lp, dtype=np.float32)
    ls = np.ascontiguousarray(ls, dtype=np.float32).reshape(-1)
    label = np.ascontiguousarray(label, dtype=np.int64)
    T, Vp = lp.shape
    L = label.shape[0]
    onset = np.full(max(L, 1), -1, dtype=np.int32)
    offset = np.full(max(L, 1), -1, dtype=np.int32)
    score = np.zeros(1, dtype=np.float64)
    path = np.zeros(T, dtype=np.int64) if want_path else None
    rc = lib().la_oracle_align(_ptr(lp), Vp, _ptr(ls), _ptr(label), L, T, _ptr(onset), _ptr(offset),
                               _ptr(score), _ptr(path) if want_path else None)
    out = (rc, onset[:L], offset[:L], float(score[0]))
    return out + (path,) if want_path else out


def align_frames_compact(em: np.ndarray, label: Sequence[int]):
    """One utterance on the HIP path's compact emission layout [T, >=L+1]."""
    em = np.ascontiguousarray(em, dtype=np.float32)
    label = np.ascontiguousarray(label, dtype=np.int64)
    T, stride = em.shape
    L = label.shape[0]
    onset = np.full(max(L, 1), -1, dtype=np.int32)
    offset = np.full(max(L, 1), -1, dtype=np.int32)
    score = np.zeros(1, dtype=np.float64)
    rc = lib().la_oracle_align_compact(_ptr(em), stride, L, T, _ptr(label), _ptr(onset), _ptr(offset),
                                       _ptr(score), None)
    return rc, onset[:L], offset[:L], float(score[0])


def _labels_of(labels, i) -> np.ndarray:
    row = labels[i]
    return np.array([int(row[j]) for j in range(len(row)) if int(row[j]) != -100], dtype=np.int64)


def _perform(log_prediction: torch.Tensor, log_silence: torch.Tensor, labels, hop: float):
    out: List[List[List[float]]] = []
    for i in range(log_prediction.shape[0]):
        cur_label = _labels_of(labels, i)
        if cur_label.shape[0] == 0:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")  # :152
        rc, on, off, _ = align_frames(log_prediction[i].numpy(), log_silence[i].numpy(), cur_label)
        if rc == 2:
            k = int(np.argmax(on < 0)) * 2 + 1
            raise ValueError(f"{k} is not in list")  # :183
        if rc != 0:
            raise RuntimeError(f"oracle status {rc}")
        out.append([[float(int(a)) * hop, float(int(b)) * hop] for a, b in zip(on, off)])
    return out


def perform_viterbi_ctc(prediction: torch.Tensor, labels, hop_size_second: float = 0.02):
    lp, ls = model_oracle.emission_prep_ctc(prediction)
    return _perform(lp, ls, labels, hop_size_second)


def perform_viterbi(prediction: torch.Tensor, labels, hop_size_second: float = 0.02):
    lp, ls = model_oracle.emission_prep_plain(prediction)
    return _perform(lp, ls, labels, hop_size_second)


get_mae = model_oracle.get_mae
