"""Drop-in for the reference's utils/alignment.py: same function names, arguments, return values and
exception types, with emission prep + the alignment DP + backtrace on the MI355X (la_emissions_from_logits,
la_viterbi_batch, la_viterbi_core).

Reference lines mirrored: perform_viterbi :13-71; run_viterbi_core :73-119; perform_viterbi_ctc :121-188;
get_mae :190-199.  `prediction` may live on the host (the reference hands over `.cpu()` logits) or on the
device (no copy).  Errors: ValueError("<k> is not in list") when a label state is never visited (:183),
IndexError when an utterance has no labels (:152) -- raised for the first offending utterance, like the
reference's per-utterance loop.  No CPU fallback: the HIP library is required.
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence, Tuple

import numpy as np
import torch

from .. import _lib, ops
from .._lib import LA_EEMPTY, LA_EINFEASIBLE, LA_OK, LA_VARIANT_CTC, LA_VARIANT_PLAIN, check, lib, ptr, stream_ptr


def _label_lists(labels, batch: int) -> List[List[int]]:
    """labels: LongTensor [B,Lmax] with -100 padding, ndarray, or list of lists -> per-utterance class ids (:141)."""
    out = []
    for i in range(batch):
        row = labels[i]
        out.append([int(row[j]) for j in range(len(row)) if int(row[j]) != -100])
    return out


def _labels_to_device(labels, batch: int, device) -> Tuple[torch.Tensor, torch.Tensor, List[List[int]]]:
    lists = _label_lists(labels, batch)
    Lmax = max(1, max((len(l) for l in lists), default=1))
    lab = torch.zeros((batch, Lmax), dtype=torch.int32)
    for b, l in enumerate(lists):
        if l:
            lab[b, : len(l)] = torch.tensor(l, dtype=torch.int32)
    n = torch.tensor([len(l) for l in lists], dtype=torch.int32)
    return lab.to(device), n.to(device), lists


def _seconds_from_frames(onset, offset, status, label_lists, hop_size_second):
    on, off, st = onset.cpu().numpy(), offset.cpu().numpy(), status.cpu().numpy()
    result = []
    for b, labs in enumerate(label_lists):
        if st[b] == LA_EEMPTY:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")          # (:152)
        if st[b] == LA_EINFEASIBLE:
            k = int(np.argmax(on[b, : len(labs)] < 0)) * 2 + 1
            raise ValueError(f"{k} is not in list")                                         # (:183)
        if st[b] != LA_OK:
            raise _lib.LyricAlignHipError(f"viterbi status {int(st[b])} for utterance {b}")
        result.append([[float(int(on[b, n])) * hop_size_second, float(int(off[b, n])) * hop_size_second]
                       for n in range(len(labs))])                                         # (:185) float(frame) * hop
    return result


def _device_of(prediction) -> torch.device:
    _lib.require_gpu()
    if torch.is_tensor(prediction) and prediction.is_cuda:
        return prediction.device
    return torch.device(f"cuda:{torch.cuda.current_device()}")


def _perform(prediction, labels, hop_size_second, variant):
    dev = _device_of(prediction)
    pred = torch.as_tensor(prediction).to(device=dev, dtype=torch.float32)
    if pred.dim() != 3:
        raise ValueError("prediction must be [B, T, V]")
    if pred.stride(2) != 1:
        pred = pred.contiguous()
    B, T, V = pred.shape
    lab, n_lab, lists = _labels_to_device(labels, B, dev)
    em = ops.emissions_from_logits(pred, lab, n_lab, variant)
    nf = torch.full((B,), T, dtype=torch.int32, device=dev)
    onset, offset, _, status = ops.viterbi_batch(em, lab, n_lab, nf)
    return _seconds_from_frames(onset, offset, status, lists, hop_size_second)


def perform_viterbi(prediction, labels, hop_size_second=0.02):
    return _perform(prediction, labels, hop_size_second, LA_VARIANT_PLAIN)


def perform_viterbi_ctc(prediction, labels, hop_size_second=0.02):
    return _perform(prediction, labels, hop_size_second, LA_VARIANT_CTC)


def run_viterbi_core(dp_matrix, backtrace_dp_matrix, cur_log_prediction, cur_log_silence_prediction, cur_label):
    """In place on the caller's numpy arrays and returned, like the reference (dp float64 [T,S], bt int64 [T,S],
    lp float32 [T,V'], ls float32 [T,1], label int64 [L]); row 0 of dp is taken as initialised by the caller."""
    dev = _device_of(None)
    lp = np.asarray(cur_log_prediction, dtype=np.float32)
    ls = np.asarray(cur_log_silence_prediction, dtype=np.float32).reshape(lp.shape[0], -1)[:, :1]
    label = np.asarray(cur_label, dtype=np.int64)
    T, L = lp.shape[0], label.shape[0]
    S = 2 * L + 1
    if dp_matrix.shape != (T, S) or backtrace_dp_matrix.shape != (T, S):
        raise ValueError("dp / backtrace matrices must be [T, 2L+1]")
    em = torch.from_numpy(np.ascontiguousarray(np.concatenate([ls, lp[:, label - 1]], axis=1))).to(dev)          # compact layout (gather = data movement)
    dp = torch.from_numpy(np.ascontiguousarray(dp_matrix, dtype=np.float64)).to(dev)
    bt = torch.from_numpy(np.ascontiguousarray(backtrace_dp_matrix, dtype=np.int64)).to(dev)
    lab = torch.from_numpy(label.astype(np.int32)).to(dev)
    nl = torch.tensor([L], dtype=torch.int32, device=dev)
    nf = torch.tensor([T], dtype=torch.int32, device=dev)
    scratch_i = torch.empty((2 * L + 1,), dtype=torch.int32, device=dev)
    scratch_f = torch.empty((1,), dtype=torch.float64, device=dev)
    need = ctypes.c_size_t(0)
    check(lib().la_viterbi_workspace_bytes(1, T, L, ctypes.byref(need)), "viterbi_workspace_bytes")
    ws = torch.empty((max(need.value, 16),), dtype=torch.uint8, device=dev)
    check(lib().la_viterbi_core(ptr(em), em.stride(0), ptr(lab), L, T, ptr(nl), ptr(nf), ptr(dp), ptr(bt), ptr(scratch_i),
                                ptr(scratch_f), ptr(ws), need.value, stream_ptr()), "viterbi_core")
    dp_matrix[...] = dp.cpu().numpy()
    backtrace_dp_matrix[...] = bt.cpu().numpy()
    return dp_matrix, backtrace_dp_matrix


def get_mae(gt, predict):
    """Mean absolute onset/offset error over every character of the batch (:190-199).  A handful of Python-float
    operations on host lists: kept in Python float64 so the value is bit-identical to the reference's."""
    error = 0.0
    cnt = 0
    for i in range(len(gt)):
        for j in range(len(gt[i])):
            error = error + abs(gt[i][j][0] - predict[i][j][0]) + abs(gt[i][j][1] - predict[i][j][1])
            cnt = cnt + 2.0
    error = error / cnt
    return error
