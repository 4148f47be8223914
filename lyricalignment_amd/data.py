"""Host-side data formats either side of the hot path (SURVEY.md 8f "next" rows 2 and 3):
records JSON (data_processor/record.py:22-39), label conventions of the collate step (dataset.py:212-232),
frame-label builder (dataset.py:129-145) and checkpoint directories written by train_multitask.py
(:461-465, 640, 678-679) / read by inference_alignment.py (:86-124).  Pure data handling: no model arithmetic.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch


@dataclass
class Record:
    audio_path: str
    text: str
    lyric_onset_offset: Optional[list] = None


def read_data(data_path: str) -> List[Record]:
    """JSON list of {song_path, lyric[, on_offset]} -> records (data_processor/record.py:22-39)."""
    assert os.path.exists(data_path)
    with open(data_path, "r") as f:
        data_list = json.load(f)
    records = []
    for data in data_list:
        rec = Record(audio_path=data["song_path"], text=data["lyric"])
        if "on_offset" in data:
            rec.lyric_onset_offset = data["on_offset"]
        records.append(rec)
    return records


def mask_special_tokens(input_ids: torch.Tensor) -> torch.Tensor:
    """collate_fn's label convention (dataset.py:215-220): drop [CLS] (first column), [PAD]=0 and [SEP]=102 -> -100."""
    t = input_ids[:, 1:].clone()
    t[t == 0] = -100
    t[t == 102] = -100
    return t


def frame_labels(lyric_tokens: Sequence[int], lyric_word_onset_offset: Sequence[Sequence[float]], use_ctc: bool = False,
                 hop_size_second: float = 0.02) -> torch.Tensor:
    """dataset.py:129-145: framewise token labels at a 20 ms hop; Python round() (banker's) like the reference."""
    fill_value = -100 if use_ctc else 0
    total = int(round(lyric_word_onset_offset[-1][-1] / hop_size_second)) + 1
    out = torch.full((total,), fill_value=fill_value)
    for j in range(len(lyric_word_onset_offset)):
        onset = int(round(lyric_word_onset_offset[j][0] / hop_size_second))
        offset = int(round(lyric_word_onset_offset[j][1] / hop_size_second)) + 1
        out[onset:offset] = int(lyric_tokens[j])
    return out


# --------------------------------------------------------------------------- #
# checkpoints                                                                   #
# --------------------------------------------------------------------------- #
def _dims_from_state_dict(sd) -> "ModelDimensions":
    from .whisper_compat import ModelDimensions
    d = sd["whisper_model.encoder.conv1.weight"].shape[0]
    n_mels = sd["whisper_model.encoder.conv1.weight"].shape[1]
    n_layer = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("whisper_model.encoder.blocks."))
    dims = ModelDimensions(n_mels=n_mels, n_audio_state=d, n_audio_head=d // 64, n_audio_layer=n_layer,
                           n_text_state=d, n_text_head=d // 64, n_text_layer=0)
    dec = [k for k in sd if k.startswith("whisper_model.decoder.blocks.")]
    if dec:
        dims.n_text_layer = 1 + max(int(k.split(".")[3]) for k in dec)
        dims.n_vocab = sd["whisper_model.decoder.token_embedding.weight"].shape[0]
        dims.n_text_ctx = sd["whisper_model.decoder.positional_embedding"].shape[0]
    return dims


def load_align_model(model_dir: str, model_name: str = "best", device: str = "cuda", compute_dtype=torch.float32):
    """inference_alignment.load_align_model_and_tokenizer (:86-124) without the network: reads
    {model_name}_model.pt (+ model_args.json when present) from a directory written by train_multitask.py and returns an
    AlignModel whose state_dict was loaded STRICTLY for the head and the Whisper encoder (decoder keys are loaded when the
    architecture has one; non-persistent buffers such as the causal mask may be present or absent)."""
    from .module.align_model import AlignModel
    from .whisper_compat import Whisper
    assert os.path.exists(model_dir)
    path = os.path.join(model_dir, f"{model_name}_model.pt")
    sd = torch.load(path, map_location="cpu")
    model_args = {}
    margs = os.path.join(model_dir, "model_args.json")
    if os.path.exists(margs):
        with open(margs) as f:
            model_args = json.load(f)
    dims = _dims_from_state_dict(sd)
    hidden = model_args.get("hidden_dim", sd["align_rnn.rnn.weight_hh_l0"].shape[1])
    out_dim = model_args.get("output_dim", sd["align_rnn.fc.weight"].shape[0])
    wm = Whisper(dims, with_decoder=dims.n_text_layer > 0)
    model = AlignModel(wm, embed_dim=model_args.get("embed_dim", dims.n_audio_state), hidden_dim=hidden, output_dim=out_dim,
                       bidirectional=model_args.get("bidirectional", True), device=device, compute_dtype=compute_dtype)
    tolerated = ("mask", "alignment_heads")
    missing, unexpected = model.load_state_dict(sd, strict=False)
    bad_missing = [k for k in missing if not any(t in k for t in tolerated)]
    bad_unexpected = [k for k in unexpected if not any(t in k for t in tolerated)]
    if bad_missing or bad_unexpected:
        raise RuntimeError(f"checkpoint does not match the AlignModel layout: missing {bad_missing[:5]}, unexpected {bad_unexpected[:5]}")
    return model.eval()


def save_align_model(model, model_dir: str, model_name: str = "last", whisper_model_name: str = "medium") -> None:
    """train_multitask.save_model layout (:461-465) + args.json / model_args.json (:640, :655-679)."""
    os.makedirs(model_dir, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(model_dir, f"{model_name}_model.pt"))
    rnn = model.align_rnn
    with open(os.path.join(model_dir, "model_args.json"), "w") as f:
        json.dump({"embed_dim": rnn.rnn.input_size, "hidden_dim": rnn.rnn.hidden_size, "output_dim": rnn.fc.out_features,
                   "bidirectional": rnn.rnn.bidirectional, "freeze_encoder": model.freeze_encoder,
                   "train_alignment": model.train_alignment, "train_transcript": model.train_transcript}, f, indent=4)
    with open(os.path.join(model_dir, "args.json"), "w") as f:
        json.dump({"whisper_model": whisper_model_name}, f, indent=4)
