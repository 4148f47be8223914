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
from typing import Optional, Tuple

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

    def decode(self, mel_or_features: torch.Tensor, options=None, tokenizer=None):
        """whisper.decode on 30 s windows: mel [B, 80, 3000] (or encoder outputs [B, 1500, d]) -> list of DecodingResult
        (lyricalignment_amd.transcribe: logit filters, greedy / sampling / beam search; token ids, text only with a tokenizer)."""
        from .transcribe import decode
        x = mel_or_features
        if x.dim() == 3 and x.shape[1] == self.dims.n_mels and x.shape[2] == N_FRAMES:
            x = self.embed_audio(x)
        return decode(self, x, options, tokenizer)

    def transcribe(self, audio, **kwargs):
        """whisper's transcribe (inference_transcript.py:88-91): sliding 30 s windows, temperature fallback, timestamp rules,
        no-speech skipping, conditioning on the previous text -- lyricalignment_amd.transcribe.transcribe.  result["tokens"] /
        ["segments"] are always there; result["text"] when whisper's rank file can be found (lyricalignment_amd/tokenizer.py)."""
        from .transcribe import transcribe
        return transcribe(self, audio, **kwargs)


def dims_for(name: str) -> ModelDimensions:
    if name not in _DIMS:
        raise KeyError(f"unknown whisper model {name!r}")
    d, h, l = _DIMS[name]
    return ModelDimensions(n_audio_state=d, n_audio_head=h, n_audio_layer=l, n_text_state=d, n_text_head=h, n_text_layer=l)


class HostIndependentRng:
    """Synthetic-weight generator whose values are the same BITS on every host.

    torch.randn's CPU kernel is vectorised per CPU capability (AVX2 / AVX512 Box-Muller paths whose log / cos differ in the
    last place, and whose fill order differs), and nn.Module default initialisation draws from the unseeded global
    generator: "seed 0" weights differed between the build container and the GPU boxes (tools/weights_fingerprint.py,
    profiles/r3_selfcheck_diagnosis.md).  Here every value is an integer function of the raw PCG64 stream
    (numpy BitGenerator.random_raw: a documented, platform-independent sequence) followed by ONE float32 multiply:
    normal()  = (sum of four uint16 - 131070) / sqrt((65536^2 - 1) / 3)   Irwin-Hall(4): mean 0, variance 1, |x| <= 3.46
    uniform() = (uint32 >> 8) * 2^-23 - 1                                  uniform on [-1, 1)"""

    _NORM = np.float32(1.0 / np.sqrt((65536.0 ** 2 - 1.0) / 3.0))

    def __init__(self, seed: int):
        self._bg = np.random.PCG64(int(seed))

    def normal(self, shape) -> torch.Tensor:
        n = int(np.prod(shape)) if len(shape) else 1
        x = torch.from_numpy(self._bg.random_raw(n).view(np.int64))
        m = 0x0000FFFF0000FFFF                                  # lane sums in integer arithmetic (exact, thread-count independent)
        x = (x & m) + ((x >> 16) & m)
        k = ((x & 0xFFFFFFFF) + (x >> 32) - 131070).to(torch.float32)
        return (k * float(self._NORM)).reshape(tuple(shape))

    def uniform(self, shape) -> torch.Tensor:
        n = int(np.prod(shape)) if len(shape) else 1
        raw = self._bg.random_raw((n + 1) // 2).view(np.uint32)[:n]
        return torch.from_numpy(((raw >> 8).astype(np.float32) * np.float32(2.0 ** -23) - np.float32(1.0)).reshape(tuple(shape)))

    def skip_normal(self, shape) -> None:
        """Move past the draws normal(shape) would take without making them (PCG64.advance is O(log n)): a rank that builds only its
        share of a model's tensors still gives every tensor the values the sequential build gives it."""
        self._bg.advance(int(np.prod(shape)) if len(shape) else 1)


def build_model(name: str = "tiny", seed: int = 0, with_decoder: bool = False, std: float = 0.02,
                dims: Optional[ModelDimensions] = None, part: Optional[Tuple[int, int]] = None) -> Whisper:
    """Random-init weights of the named architecture (no checkpoints are reachable offline); bit-identical on every host
    (HostIndependentRng), so a bench or test that names a seed means the same model on the build container, the
    builder's GPU box and the driver's.
    part = (r, w): generate only the parameters whose index is r modulo w (the others stay zero, their draws are skipped): the
    ranks of one node each build 1 / w of the model and exchange the pieces (build_model_shared) instead of w full host builds."""
    dims = dims or dims_for(name)
    with torch.device("meta"):                      # every parameter is assigned below: skip nn.Module's own initialisation
        model = Whisper(dims, with_decoder=with_decoder)
    model = model.to_empty(device="cpu")
    with torch.no_grad():
        # whisper's sinusoids() with its float32 rounding points (the argument t * inv_timescale is a float32 product), but exp / sin /
        # cos evaluated in float64 and rounded once: torch's float32 transcendentals differ in the last place between hosts (vendor
        # math libraries), a double-precision value rounded to float32 practically never does
        n_ctx, ch = dims.n_audio_ctx, dims.n_audio_state
        inc = np.float32(np.log(10000.0) / (ch // 2 - 1))
        inv = np.exp((-inc * np.arange(ch // 2, dtype=np.float32)).astype(np.float64)).astype(np.float32)
        st = (np.arange(n_ctx, dtype=np.float32)[:, None] * inv[None, :]).astype(np.float64)
        model.encoder.positional_embedding.copy_(torch.from_numpy(np.concatenate([np.sin(st), np.cos(st)], axis=1).astype(np.float32)))
    g = HostIndependentRng(seed)
    with torch.no_grad():
        for i, (n, p) in enumerate(model.named_parameters()):
            if part is not None and i % part[1] != part[0]:
                g.skip_normal(p.shape)
                p.zero_()
                continue
            if n.endswith("_ln.weight") or n.endswith("ln_post.weight") or n.endswith("ln.weight"):
                p.copy_(1.0 + 0.1 * g.normal(p.shape))
            elif "ln" in n.split(".")[-2] and n.endswith("bias"):
                p.copy_(0.1 * g.normal(p.shape))
            elif n.startswith("encoder.conv1.weight"):
                p.copy_(0.05 * g.normal(p.shape))
            else:
                p.copy_(std * g.normal(p.shape))
    return model


def _share_from_rank0(obj, rank: int):
    """rank 0's `obj` on every rank of the default process group."""
    import torch.distributed as dist
    box = [obj if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def build_model_shared(name: str, seed: int, with_decoder: bool, rank: int, world: int, barrier, tag: str,
                       shm_dir: str = "/dev/shm", share=None) -> Whisper:
    """build_model for the `world` ranks of ONE node: every rank generates the parameters with index = rank (mod world), writes them
    to a file in shared memory, and reads the others' after `barrier()` -- the same bits as build_model on every rank, 1 / world of the
    generator work per rank (the generator is a single-threaded stream: 8 full builds on 2 threads each took 11 s apiece).
    RANK 0 ALONE decides where the pieces are exchanged -- a directory of its own making (mkdtemp: mode 0700, unpredictable name) under
    `shm_dir` or the temporary directory, or nowhere (no room: every rank builds the whole model) -- and every rank follows that one
    decision (`share(obj)`: rank 0's object on every rank; default = broadcast_object_list over the default process group), so all ranks
    take the same path through the two barriers whatever each of them would have measured for itself.  Pieces are read with
    weights_only=True.  `tag` names this job's directory (e.g. the rendezvous port); it is removed after the second barrier."""
    if world == 1:
        return build_model(name, seed=seed, with_decoder=with_decoder)
    import os
    import shutil
    import tempfile
    if share is None:
        share = lambda obj: _share_from_rank0(obj, rank)
    dims = dims_for(name)
    exchange = None
    if rank == 0:
        # room for the whole model in the exchange directory?  (a container's /dev/shm can be 64 MB)
        need = 4 * (12 * dims.n_audio_layer * dims.n_audio_state ** 2
                    + (16 * dims.n_text_layer * dims.n_text_state ** 2 + dims.n_vocab * dims.n_text_state if with_decoder else 0)) + (64 << 20)
        for base in (shm_dir, tempfile.gettempdir()):
            try:
                if shutil.disk_usage(base).free >= 2 * need:
                    exchange = tempfile.mkdtemp(prefix=f"la_weights_{tag}_", dir=base)
                    break
            except OSError:
                continue
    exchange = share(exchange)
    if exchange is None:                                         # no room anywhere: every rank builds the whole model (no barrier on this path,
        return build_model(name, seed=seed, with_decoder=with_decoder)      # on any rank)
    model = build_model(name, seed=seed, with_decoder=with_decoder, part=(rank, world))
    params = list(model.named_parameters())
    path = lambda r: os.path.join(exchange, f"{r}.pt")
    torch.save({n: p.detach() for i, (n, p) in enumerate(params) if i % world == rank}, path(rank))
    barrier()
    with torch.no_grad():
        for r in range(world):
            if r == rank:
                continue
            piece = torch.load(path(r), map_location="cpu", weights_only=True)
            for i, (n, p) in enumerate(params):
                if i % world == r:
                    p.copy_(piece[n])
    barrier()
    for f in (lambda: os.remove(path(rank)), lambda: os.rmdir(exchange)):       # the last rank to leave takes the directory with it
        try:
            f()
        except OSError:
            pass
    return model


def init_align_head(model, seed: int = 7, fc_scale: float = 12.0, rnn_scale: float = 1.5) -> None:
    """Synthetic weights for AlignModel.align_rnn (module/align_model.py:11-40), bit-identical on every host: uniform on
    +-scale / sqrt(hidden), GRU matrices and biases with rnn_scale, the output Linear's weight with fc_scale.  fc_scale ~ 12
    gives frame posteriors as peaked as a trained head's (|logit| <= ~20); nn.Module's default initialisation
    (fc_scale ~ 0.6) leaves all 21129 classes within +-0.8 of each other, and on such near-flat emissions the best lattice
    path is decided by differences far below the 16-bit modes' rounding -- boundaries of a flat head are not comparable
    between precisions (tests/test_gpu_parity_full.py docstring; profiles/r3_selfcheck_diagnosis.md)."""
    g = HostIndependentRng(seed)
    hidden = model.align_rnn.rnn.hidden_size
    with torch.no_grad():
        for n, p in model.align_rnn.named_parameters():
            s = fc_scale if n.startswith("fc.weight") else rnn_scale
            p.copy_(g.uniform(p.shape) * (s / hidden ** 0.5))


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


# ----------------------------------------------------------------------------------------------- Hugging Face <-> upstream names
# The reference stores upstream openai-whisper key names under `whisper_model.` (train_multitask.py:461-465 saves
# AlignModel.state_dict(); inference_alignment.py:86-124 loads it back).  Whisper weights that exist on disk without the network
# are usually the Hugging Face export of the same tensors under other names.  The two tables below are the
# whole difference; tensors are identical (HF stores the sinusoid table of the encoder as an Embedding weight, upstream as a buffer).
_HF_LAYER = (("self_attn.q_proj", "attn.query"), ("self_attn.k_proj", "attn.key"), ("self_attn.v_proj", "attn.value"),
             ("self_attn.out_proj", "attn.out"), ("self_attn_layer_norm", "attn_ln"),
             ("encoder_attn.q_proj", "cross_attn.query"), ("encoder_attn.k_proj", "cross_attn.key"),
             ("encoder_attn.v_proj", "cross_attn.value"), ("encoder_attn.out_proj", "cross_attn.out"),
             ("encoder_attn_layer_norm", "cross_attn_ln"), ("fc1", "mlp.0"), ("fc2", "mlp.2"), ("final_layer_norm", "mlp_ln"))
_HF_TOP = (("encoder.embed_positions.weight", "encoder.positional_embedding"), ("encoder.layer_norm.", "encoder.ln_post."),
           ("decoder.embed_tokens.weight", "decoder.token_embedding.weight"),
           ("decoder.embed_positions.weight", "decoder.positional_embedding"), ("decoder.layer_norm.", "decoder.ln."))


def hf_to_upstream_key(key: str) -> Optional[str]:
    """`model.encoder.layers.3.self_attn.q_proj.weight` (transformers WhisperModel / WhisperForConditionalGeneration) ->
    `encoder.blocks.3.attn.query.weight` (openai-whisper); None for `proj_out.weight` (tied to the token embedding: upstream
    has no such tensor)."""
    if key == "proj_out.weight":
        return None
    k = key[len("model."):] if key.startswith("model.") else key
    for hf, up in _HF_TOP:
        if k.startswith(hf):
            return up + k[len(hf):]
    side, _, rest = k.partition(".layers.")
    if rest:
        idx, _, tail = rest.partition(".")
        for hf, up in _HF_LAYER:
            if tail.startswith(hf + "."):
                return f"{side}.blocks.{idx}.{up}{tail[len(hf):]}"
        raise KeyError(f"unknown Hugging Face whisper layer key {key!r}")
    if k.split(".")[0] in ("encoder", "decoder") and k.split(".")[1] in ("conv1", "conv2"):
        return k
    raise KeyError(f"unknown Hugging Face whisper key {key!r}")


def upstream_to_hf_key(key: str) -> str:
    """Inverse of hf_to_upstream_key (keys of WhisperForConditionalGeneration: `model.` prefix)."""
    for hf, up in _HF_TOP:
        if key.startswith(up):
            return "model." + hf + key[len(up):]
    side, _, rest = key.partition(".blocks.")
    if rest:
        idx, _, tail = rest.partition(".")
        for hf, up in sorted(_HF_LAYER, key=lambda t: -len(t[1])):       # `cross_attn_ln` before `cross_attn`, `attn_ln` before `attn`
            if tail.startswith(up + "."):
                return f"model.{side}.layers.{idx}.{hf}{tail[len(up):]}"
        raise KeyError(f"unknown whisper block key {key!r}")
    return "model." + key


def hf_state_dict_to_upstream(sd: dict, prefix: str = "") -> dict:
    """A transformers Whisper state_dict under upstream names (tensors shared, not copied); prefix='whisper_model.' gives the
    keys AlignModel.state_dict() uses for the backbone."""
    out = {}
    for k, v in sd.items():
        u = hf_to_upstream_key(k)
        if u is not None:
            out[prefix + u] = v
    return out


def upstream_key_shapes(dims: ModelDimensions, with_decoder: bool = True) -> dict:
    """{upstream key: shape} of a whisper checkpoint of these dimensions, written out from SURVEY.md Appendix B / C (NOT read off
    this package's modules: tests hold both the modules and a transformers model of the same size to this table)."""
    d, dt = dims.n_audio_state, dims.n_text_state
    s = {"encoder.conv1.weight": (d, dims.n_mels, 3), "encoder.conv1.bias": (d,), "encoder.conv2.weight": (d, d, 3),
         "encoder.conv2.bias": (d,), "encoder.positional_embedding": (dims.n_audio_ctx, d),
         "encoder.ln_post.weight": (d,), "encoder.ln_post.bias": (d,)}

    def block(p: str, w: int, cross: bool):
        for a in ("attn", "cross_attn") if cross else ("attn",):
            for n in ("query", "key", "value", "out"):
                s[f"{p}.{a}.{n}.weight"] = (w, w)
                if n != "key":                                   # whisper's key projection has no bias
                    s[f"{p}.{a}.{n}.bias"] = (w,)
            s[f"{p}.{a}_ln.weight"] = s[f"{p}.{a}_ln.bias"] = (w,)
        s[f"{p}.mlp.0.weight"], s[f"{p}.mlp.0.bias"] = (4 * w, w), (4 * w,)
        s[f"{p}.mlp.2.weight"], s[f"{p}.mlp.2.bias"] = (w, 4 * w), (w,)
        s[f"{p}.mlp_ln.weight"] = s[f"{p}.mlp_ln.bias"] = (w,)

    for i in range(dims.n_audio_layer):
        block(f"encoder.blocks.{i}", d, False)
    if with_decoder:
        s["decoder.token_embedding.weight"] = (dims.n_vocab, dt)
        s["decoder.positional_embedding"] = (dims.n_text_ctx, dt)
        s["decoder.ln.weight"] = s["decoder.ln.bias"] = (dt,)
        for i in range(dims.n_text_layer):
            block(f"decoder.blocks.{i}", dt, True)
    return s
