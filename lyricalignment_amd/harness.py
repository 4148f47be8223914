"""The build's counterpart of the reference's evaluation glue (inference_alignment.align_and_evaluate,
inference_alignment.py:126-180, and inference_alignment_nogt.py:130-178): per-batch label mapping, forward,
Viterbi, MAE averaging.  Reproduces, on purpose:
  (1) class = pinyin_lookup_table[token_pinyin[token_id]] for every token id != -100 (:149-152)
  (2) batches whose ground truth is (None,) are skipped (:156-157)
  (3) avg_mae = sum(per-batch get_mae) / #evaluated batches  -- a mean of per-batch means (:172-177)
so the sharded evaluation keeps batch composition and averages the gathered per-batch values.
The alignment itself goes through AlignModel.align (fused HIP path) or, with two_step=True, through the
drop-in frame_manual_forward + perform_viterbi(_ctc) pair exactly like the reference's loop.
"""
from __future__ import annotations

from typing import Any, Iterable, List, Optional, Sequence

import numpy as np
import torch

from .sharding import map_sharded
from .utils.alignment import get_mae, perform_viterbi, perform_viterbi_ctc


class PinyinClassLUT:
    """Vectorised form of the reference's Python double loop: a [n_tokens] int table token id -> class id."""

    def __init__(self, token_pinyin: Sequence[str], pinyin_lookup_table: dict):
        self.table = np.asarray([pinyin_lookup_table[p] for p in token_pinyin], dtype=np.int64)

    def __call__(self, tokens) -> torch.Tensor:
        t = torch.as_tensor(tokens).clone().long()
        keep = t != -100
        t[keep] = torch.from_numpy(self.table)[t[keep]]
        return t


def evaluate_batches(model, batches: Sequence[Any], lut: Optional[PinyinClassLUT] = None, use_ctc_loss: bool = False,
                     two_step: bool = False, rank: int = 0, world: int = 1):
    """batches: sequence of (audios, tokens, _, lyric_word_onset_offset, _, _) as the reference's DataLoader yields.
    Returns (avg_mae, per_batch_maes) with skipped batches as None; identical on every rank."""

    def run(i: int):
        audios, tokens, _, onset_offset, _, _ = batches[i]
        labels = lut(tokens) if lut is not None else torch.as_tensor(tokens)
        if onset_offset == (None,):
            return None
        if two_step:
            logits, _ = model.frame_manual_forward(audios)
            res = (perform_viterbi_ctc if use_ctc_loss else perform_viterbi)(logits, labels)
        else:
            res = model.align(audios, labels, use_ctc=use_ctc_loss)
        return get_mae(onset_offset, res)

    with torch.no_grad():
        maes = map_sharded(run, len(batches), rank, world)
    done = [m for m in maes if m is not None]
    total = 0
    for m in done:          # same accumulation order as the reference's running sum
        total += m
    avg = total / len(done) if done else float("nan")
    return avg, maes


def align_records(model, records: Iterable[Any], lut: PinyinClassLUT, tokenize, use_ctc_loss: bool = True) -> List[list]:
    """inference_alignment_nogt.py:130-178: one record at a time, returns [[onset, offset, char], ...] per record.
    `tokenize(text) -> list[int]` are the BERT ids without [CLS]/[SEP] (the reference slices [1:-1], :158-163)."""
    out = []
    with torch.no_grad():
        for rec in records:
            ids = torch.tensor([tokenize(rec.text)], dtype=torch.long)
            res = model.align([rec.audio], lut(ids), use_ctc=use_ctc_loss)[0]
            out.append([[res[j][0], res[j][1], rec.text[j]] for j in range(len(res))])
    return out
