"""Minimal stand-in for the parts of `openai-whisper` the reference's alignment path touches
(module/align_model.py:8-9,84,89,91,100-101,109,112,120; train_multitask.py:647-648).

openai-whisper is not installed in this image (and is un-pinned in the reference's
requirements.txt:6).  This module keeps its *names*, *state_dict key layout* and call
signatures so that (a) checkpoints written by the reference load unchanged and (b) an AlignModel
can be built offline with random-init weights of the right architecture.  The compute behind
`embed_audio` is the HIP encoder of lyricalignment_amd.engine -- there is no torch fallback.
If the real `whisper` package is importable, AlignModel accepts its models just the same: only
`.encoder` parameters (by name) and `.dims` are read.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
from torch import nn

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
N_FRAMES = 3000
N_SAMPLES = 480000

_DIMS = {  # name -> (n_state, n_head, n_layer)
    "tiny": (384, 6, 4), "base": (512, 8, 6), "small": (768, 12, 12), "medium": (1024, 16, 24),
    "large": (1280, 20, 32), "large-v1": (1280, 20, 32), "large-v2": (1280, 20, 32),
}


@dataclass
class ModelDimensions:
    n_mels: int = 80
    n_audio_ctx: int = 1500
    n_audio_state: int = 384
    n_audio_head: int = 6
    n_audio_layer: int = 4
    n_vocab: int = 51865
    n_text_ctx: int = 448
    n_text_state: int = 384
    n_text_head: int = 6
    n_text_layer: int = 4


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    log_timescale_increment = np.log(max_timescale) / (channels // 2 - 1)
    inv_timescales = torch.exp(-log_timescale_increment * torch.arange(channels // 2))
    scaled_time = torch.arange(length)[:, np.newaxis] * inv_timescales[np.newaxis, :]
    return torch.cat([torch.sin(scaled_time), torch.cos(scaled_time)], dim=1)


class _Attention(nn.Module):
    def __init__(self, n_state: int, n_head: int):
        super().__init__()
        self.n_head = n_head
        self.query = nn.Linear(n_state, n_state)
        self.key = nn.Linear(n_state, n_state, bias=False)
        self.value = nn.Linear(n_state, n_state)
        self.out = nn.Linear(n_state, n_state)


class _Block(nn.Module):
    def __init__(self, n_state: int, n_head: int, cross_attention: bool = False):
        super().__init__()
        self.attn = _Attention(n_state, n_head)
        self.attn_ln = nn.LayerNorm(n_state)
        self.cross_attn = _Attention(n_state, n_head) if cross_attention else None
        self.cross_attn_ln = nn.LayerNorm(n_state) if cross_attention else None
        self.mlp = nn.Sequential(nn.Linear(n_state, 4 * n_state), nn.GELU(), nn.Linear(4 * n_state, n_state))
        self.mlp_ln = nn.LayerNorm(n_state)


class AudioEncoder(nn.Module):
    """Parameter container with openai-whisper's names; forward runs on the HIP engine."""

    def __init__(self, n_mels: int, n_ctx: int, n_state: int, n_head: int, n_layer: int):
        super().__init__()
        self.conv1 = nn.Conv1d(n_mels, n_state, kernel_size=3, padding=1)
        self.conv2 = nn.Conv1d(n_state, n_state, kernel_size=3, stride=2, padding=1)
        self.register_buffer("positional_embedding", sinusoids(n_ctx, n_state))
        self.blocks = nn.ModuleList([_Block(n_state, n_head) for _ in range(n_layer)])
        self.ln_post = nn.LayerNorm(n_state)
        self.n_head = n_head


class TextDecoder(nn.Module):
    """Parameter container with openai-whisper's names; forward runs on the HIP engine (AlignEngine.decode)."""

    def __init__(self, n_vocab: int, n_ctx: int, n_state: int, n_head: int, n_layer: int):
        super().__init__()
        self.token_embedding = nn.Embedding(n_vocab, n_state)
        self.positional_embedding = nn.Parameter(torch.empty(n_ctx, n_state).normal_(std=0.01))
        self.blocks = nn.ModuleList([_Block(n_state, n_head, cross_attention=True) for _ in range(n_layer)])
        self.ln = nn.LayerNorm(n_state)


class Whisper(nn.Module):
    def __init__(self, dims: ModelDimensions, with_decoder: bool = False):
        super().__init__()
        self.dims = dims
        self.encoder = AudioEncoder(dims.n_mels, dims.n_audio_ctx, dims.n_audio_state, dims.n_audio_head, dims.n_audio_layer)
        self.decoder = TextDecoder(dims.n_vocab, dims.n_text_ctx, dims.n_text_state, dims.n_text_head,
                                   dims.n_text_layer) if with_decoder else None
        self._engine_owner = None  # set by AlignModel so embed_audio can reach the packed HIP weights

    def embed_audio(self, mel: torch.Tensor) -> torch.Tensor:
        if self._engine_owner is None:
            from .module.align_model import encoder_only_engine
            return encoder_only_engine(self, mel)
        return self._engine_owner()._embed_audio(mel)

    def logits(self, tokens: torch.Tensor, audio_features: torch.Tensor) -> torch.Tensor:
        """TextDecoder forward on the HIP engine: tokens [B,n], audio_features [B,1500,d] -> [B,n,n_vocab] float32."""
        if self.decoder is None:
            raise RuntimeError("this Whisper object was built without a decoder")
        from .module.align_model import decoder_engine
        return decoder_engine(self, tokens, audio_features)

    def decode_greedy(self, prompt: torch.Tensor, audio_features: torch.Tensor, max_new_tokens: int, eot: int) -> torch.Tensor:
        """Greedy token loop on the HIP engine with a key / value cache: prompt int64 [B,n0] (sot / language / task tokens),
        audio_features [B,1500,d] -> tokens int64 [B, n0 + steps] (finished rows padded with `eot`).  This is the device part
        of the reference's transcript script (inference_transcript.py:72-104 calls whisper's transcribe with beam_size 5);
        text needs whisper's tiktoken vocabulary, which is not in this image, so the surface stops at token ids."""
        if self.decoder is None:
            raise RuntimeError("this Whisper object was built without a decoder")
        from .module.align_model import decoder_engine
        return decoder_engine(self, prompt, audio_features, greedy=(int(max_new_tokens), int(eot)))

    def decode_beam(self, prompt: torch.Tensor, audio_features: torch.Tensor, beam_size: int, max_new_tokens: int, eot: int):
        """Beam search on the HIP engine (AlignEngine.decode_beam): -> (list of token tensors, list of summed log-probs)."""
        if self.decoder is None:
            raise RuntimeError("this Whisper object was built without a decoder")
        from .module.align_model import decoder_engine
        return decoder_engine(self, prompt, audio_features, greedy=(int(max_new_tokens), int(eot), int(beam_size)))

    def transcribe(self, *a, **k):
        raise NotImplementedError("whisper's transcribe (temperature fallback, timestamp rules, token suppression, tokenizer) is "
                                  "not rebuilt here; decode_greedy() / decode_beam() run the decoder's token loop on the device")


def dims_for(name: str) -> ModelDimensions:
    if name not in _DIMS:
        raise KeyError(f"unknown whisper model {name!r}")
    d, h, l = _DIMS[name]
    return ModelDimensions(n_audio_state=d, n_audio_head=h, n_audio_layer=l, n_text_state=d, n_text_head=h, n_text_layer=l)


def build_model(name: str = "tiny", seed: int = 0, with_decoder: bool = False, std: float = 0.02,
                dims: Optional[ModelDimensions] = None) -> Whisper:
    """Random-init weights of the named architecture (no checkpoints are reachable offline)."""
    dims = dims or dims_for(name)
    model = Whisper(dims, with_decoder=with_decoder)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("_ln.weight") or n.endswith("ln_post.weight") or n.endswith("ln.weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif "ln" in n.split(".")[-2] and n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif n.startswith("encoder.conv1.weight"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(std * torch.randn(p.shape, generator=g))
    return model


def load_model(name_or_path: str, device: str = "cuda", **_) -> Whisper:
    """whisper.load_model look-alike for LOCAL openai-format checkpoints ({'dims', 'model_state_dict'})."""
    import os
    if not os.path.exists(name_or_path):
        raise RuntimeError(f"{name_or_path!r}: no network here -- pass a local .pt checkpoint or use build_model(name)")
    ck = torch.load(name_or_path, map_location="cpu")
    dims = ModelDimensions(**ck["dims"])
    model = Whisper(dims, with_decoder=True)
    missing, unexpected = model.load_state_dict(ck["model_state_dict"], strict=False)
    bad = [k for k in missing if "mask" not in k and "alignment_heads" not in k]
    if bad:
        raise RuntimeError(f"checkpoint misses keys {bad[:5]}...")
    return model.float().to(device)


def pad_or_trim(array, length: int = N_FRAMES, *, axis: int = -1):
    """Zero-pad / trim the last axis (whisper.audio.pad_or_trim); pure data movement."""
    if torch.is_tensor(array):
        if array.shape[axis] > length:
            array = array.index_select(dim=axis, index=torch.arange(length, device=array.device))
        if array.shape[axis] < length:
            pad = [(0, 0)] * array.ndim
            pad[axis] = (0, length - array.shape[axis])
            array = torch.nn.functional.pad(array, [p for sizes in pad[::-1] for p in sizes])
        return array
    if array.shape[axis] > length:
        array = array.take(indices=range(length), axis=axis)
    if array.shape[axis] < length:
        pad = [(0, 0)] * array.ndim
        pad[axis] = (0, length - array.shape[axis])
        array = np.pad(array, pad)
    return array
