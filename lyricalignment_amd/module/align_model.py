"""Drop-in for the reference's module/align_model.py (same class names, constructor and method
signatures, attribute names and state_dict key layout), with the inference arithmetic on the
MI355X HIP kernels of liblyricalign_hip.so.

Reference lines mirrored:  RNN  module/align_model.py:11-40;  AlignModel.__init__ :43-70;
frame_manual_forward :72-123;  forward :126-152.

Differences that are deliberate:
  * `audios` may be a list OR a tuple and is never mutated (the reference pads the caller's list
    in place, :78-80, and fails on tuples);
  * the log-mel runs on the device (the reference computes it on the CPU, :84);
  * `compute_dtype` (extra keyword, default float32 = the reference's numerics; torch.bfloat16 / torch.float16 are
    the throughput mode) and `align(...)` (the fused, logits-free fast path) are additions.
There is no PyTorch / CPU fallback: without the HIP library and a gfx950 device these methods raise.
"""
from __future__ import annotations

import os
import weakref
from typing import List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from .. import _lib, ops
from ..audio_frontend import log_mel_spectrogram
from ..engine import N_CTX, N_FRAMES, AlignEngine, pack_decoder, pack_encoder, pack_head
from ..whisper_compat import pad_or_trim


def frame_plan(n_mel: int, get_orig_len: bool = True):
    """Host bookkeeping of frame_manual_forward (module/align_model.py:86-115): the 3000-frame mel chunks that go
    through the encoder and how many of each chunk's 1500 output frames are kept.  Python round() = banker's
    rounding, as in the reference (301 -> 150, 303 -> 152).  -> [(start, end, kept_frames)]"""
    if not get_orig_len:
        return [(0, min(n_mel, N_FRAMES), N_CTX)]
    if n_mel <= N_FRAMES:
        return [(0, n_mel, int(round(n_mel / 2.0)))]
    plan = []
    for start in range(0, n_mel, N_FRAMES):
        end = min(start + N_FRAMES, n_mel)
        plan.append((start, end, int(round((end - start) / 2.0))))
    return plan


def song_major_chunks(mel: torch.Tensor, plan) -> torch.Tensor:
    """Long form (:94-105): mel [S,80,n] -> zero-padded 3000-frame chunks ordered song-major, [S*C, 80, 3000].  Every chunk
    but a song's last keeps all of its 1500 output frames, so with this order song s's frames are the contiguous encoder
    output rows s*C*1500 .. +T: the head reads them in place (clip stride C*1500), nothing is concatenated."""
    S, C = mel.shape[0], len(plan)
    assert all(k == N_CTX for _, _, k in plan[:-1])
    out = torch.zeros((S, C, mel.shape[1], N_FRAMES), dtype=mel.dtype, device=mel.device)
    for c, (s0, e0, _) in enumerate(plan):
        out[:, c, :, : e0 - s0] = mel[:, :, s0:e0]
    return out.view(S * C, mel.shape[1], N_FRAMES)


# LA_BRANCH_STREAMS=0: alignment head and text decoder of the training forward / backward on one stream (A/B partner; read at import)
BRANCH_STREAMS = os.environ.get("LA_BRANCH_STREAMS", "1") != "0"

class RNN(nn.Module):
    """GRU(2 layers, bidirectional) -> Mish -> Linear; parameters live in nn.GRU / nn.Linear so the
    state_dict keys are the reference's (align_rnn.rnn.weight_ih_l0 ... align_rnn.fc.bias)."""

    def __init__(self, input_size, hidden_size, output_size, num_layers: int = 2, dropout: float = 0.1,
                 batch_first: bool = True, bidirectional: bool = True) -> None:
        super().__init__()
        self.rnn = nn.GRU(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, dropout=dropout,
                          batch_first=batch_first, bidirectional=bidirectional)
        self.activate = nn.Mish()
        self.fc = nn.Linear(hidden_size + (bidirectional * hidden_size), output_size)
        self._owner = None

    def forward(self, x):
        if self._owner is None or self._owner() is None:
            raise _lib.LyricAlignHipError("RNN.forward runs through its AlignModel's HIP engine; call it via AlignModel")
        return self._owner()._head_logits(x)


class AlignModel(torch.nn.Module):
    def __init__(self, whisper_model, embed_dim: int = 1280, hidden_dim: int = 384, dropout: float = 0.15,
                 output_dim: int = 10000, bidirectional: bool = True, freeze_encoder: bool = False,
                 train_alignment: bool = True, train_transcript: bool = False, device: str = 'cuda',
                 compute_dtype: torch.dtype = torch.float32) -> None:
        super().__init__()
        self.whisper_model = whisper_model
        self.align_rnn = RNN(input_size=embed_dim, hidden_size=hidden_dim, output_size=output_dim,
                             bidirectional=bidirectional, dropout=dropout)
        self.freeze_encoder = freeze_encoder
        self.train_alignment = train_alignment
        self.train_transcript = train_transcript
        self.device = device
        self.compute_dtype = compute_dtype
        self._engine: Optional[AlignEngine] = None
        self._engine_key = None
        self.align_rnn._owner = weakref.ref(self)
        if hasattr(whisper_model, "_engine_owner"):
            whisper_model._engine_owner = weakref.ref(self)

    # ------------------------------------------------------------------ engine plumbing
    def _n_head(self) -> int:
        dims = getattr(self.whisper_model, "dims", None)
        if dims is not None:
            return int(dims.n_audio_head)
        return int(self.whisper_model.encoder.conv1.weight.shape[0]) // 64

    def _weights_version(self):
        from ..encoder_train import _EPOCH
        return (tuple((p._version, p.data_ptr()) for p in self.parameters())
                + (_EPOCH[0], str(self.compute_dtype), bool(self.train_transcript)))

    def engine(self) -> AlignEngine:
        """Packed device weights; re-packed when parameters were updated (optimizer step, load_state_dict)."""
        key = self._weights_version()
        if self._engine is None or self._engine_key != key:
            dev = self._device()
            enc_sd = {"encoder." + k: v for k, v in self.whisper_model.encoder.state_dict().items()}
            head_sd = {"align_rnn." + k: v for k, v in self.align_rnn.state_dict().items()}
            enc = pack_encoder(enc_sd, self._n_head(), self.compute_dtype, dev)
            head = pack_head(head_sd, self.compute_dtype, dev)
            dec = None
            decoder = getattr(self.whisper_model, "decoder", None)
            if decoder is not None and self.train_transcript:
                dec_sd = {"decoder." + k: v for k, v in decoder.state_dict().items()}
                dec = pack_decoder(dec_sd, int(getattr(self.whisper_model.dims, "n_text_head", self._n_head())), self.compute_dtype, dev)
            self._engine = AlignEngine(enc, head, dev, dec=dec)
            self._engine_key = key
        return self._engine

    def _device(self) -> torch.device:
        _lib.require_gpu()
        dev = torch.device(self.device if self.device != 'cuda' else f'cuda:{torch.cuda.current_device()}')
        if dev.type != "cuda":
            raise _lib.LyricAlignHipError("AlignModel runs on the MI355X only (device must be a cuda/HIP device)")
        return dev

    def _wants_grad(self) -> bool:
        return torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters())

    def _encoder_frozen(self) -> bool:
        """No encoder parameter asks for a gradient.  (The reference's freeze_encoder flag only acts in forward(), :135-139;
        frame_manual_forward back-propagates into whatever still requires grad, and so does this.)"""
        return not any(p.requires_grad for p in self.whisper_model.encoder.parameters())

    def _head_train_logits(self, feats: torch.Tensor, B: int, T: int, stride: int) -> torch.Tensor:
        """Training-mode head (float32, autograd through the HIP forward/backward kernels, inter-layer dropout active)."""
        from ..head_train import HeadFunction, head_params
        d = feats.shape[1]
        x = feats.view(-1, stride, d)[:, :T] if stride != T else feats.view(B, T, d)
        return HeadFunction.apply(x.float(), float(self.align_rnn.rnn.dropout), True, *head_params(self.align_rnn))

    def _encoder_train_features(self, mel: torch.Tensor, get_orig_len: bool) -> torch.Tensor:
        """Trainable-backbone encoder (float32, autograd through the HIP forward/backward kernels): mel -> (embed [B, T, d],
        embed_pad [B, 1500, d]) with the same chunking as the forward-only path (:87-115)."""
        from ..encoder_train import EncoderFunction, encoder_params
        enc = self.whisper_model.encoder
        pos = enc.positional_embedding
        params = encoder_params(enc)
        plan = frame_plan(mel.shape[-1], get_orig_len)
        H = self._n_head()
        if len(plan) == 1:
            embed_pad = EncoderFunction.apply(pad_or_trim(mel, N_FRAMES), pos, H, *params)
            return embed_pad[:, : plan[0][2]], embed_pad
        chunks = [pad_or_trim(mel[:, :, s:e], N_FRAMES) for s, e, _ in plan]
        B = mel.shape[0]
        y = EncoderFunction.apply(torch.cat(chunks, dim=0), pos, H, *params).view(len(chunks), B, N_CTX, -1)
        embed = torch.cat([y[c, :, : plan[c][2]] for c in range(len(chunks))], dim=1)
        return embed, embed[:, :N_CTX]

    def _decoder_train_logits(self, y_in: torch.Tensor, embed_pad: torch.Tensor) -> torch.Tensor:
        """whisper_model.logits(tokens=y_in, audio_features=embed_pad) under autograd (:118-121)."""
        from ..decoder_train import DecoderFunction, decoder_params
        dec = self.whisper_model.decoder
        n_head = int(getattr(self.whisper_model.dims, "n_text_head", self._n_head()))
        return DecoderFunction.apply(y_in.to(embed_pad.device), embed_pad, n_head, *decoder_params(dec))

    def _embed_audio(self, mel: torch.Tensor) -> torch.Tensor:
        """whisper_model.embed_audio: [B,80,3000] -> [B,1500,d] float32."""
        eng = self.engine()
        B = mel.shape[0]
        y = eng.encode(mel, out_dtype=torch.float32)
        return y.view(B, N_CTX, eng.enc.d).clone()

    def _head_logits(self, embed: torch.Tensor) -> torch.Tensor:
        """align_rnn(embed): [B,T,d] -> [B,T,output_dim] float32."""
        eng = self.engine()
        B, T, d = embed.shape
        feats = embed.to(device=eng.device, dtype=self.compute_dtype).contiguous().view(B * T, d)
        out = eng.logits(feats, B, T, T)
        return out

    # ------------------------------------------------------------------ reference API
    def _mel_of(self, audios: Sequence[np.ndarray]) -> torch.Tensor:
        max_audio_len = max(map(len, audios))
        batch = np.zeros((len(audios), max_audio_len), dtype=np.float32)   # zero-pad to the batch max (:78-82), no mutation
        for i, a in enumerate(audios):
            batch[i, : len(a)] = np.asarray(a, dtype=np.float32)
        return log_mel_spectrogram(batch, device=self._device())           # (:84) on the device

    def _features(self, mel: torch.Tensor, get_orig_len: bool):
        """-> (feats rows [., d] in compute dtype, B, T, clip stride in rows, embed_pad provider)."""
        eng = self.engine()
        B = mel.shape[0]
        plan = frame_plan(mel.shape[-1], get_orig_len)
        if len(plan) == 1:                                                  # (:87-92) and (:109-115)
            feats = eng.encode(pad_or_trim(mel, N_FRAMES))
            return feats, B, plan[0][2], N_CTX
        # long form (:94-105): non-overlapping 3000-frame chunks, every chunk of every clip in ONE encoder batch
        feats = eng.encode(song_major_chunks(mel, plan))
        return feats, B, sum(k for _, _, k in plan), len(plan) * N_CTX

    def frame_manual_forward(self, audios: List[np.ndarray], y_in=None, get_orig_len: bool = True, mel: Optional[torch.Tensor] = None):
        """(:72-123).  `mel` (addition, training path only): a ready log-mel batch instead of `audios` -- FineTuner.accumulate
        computes it per micro-batch (the reference's log-mel clamps at the BATCH maximum - 8, :84, so a micro-batch's features
        depend on which clips it holds) and runs the micro-batches as one batch from there."""
        train = self._wants_grad()
        if train and not self._encoder_frozen():                            # whole-model fine-tune (train_multitask.py default)
            from ..head_train import HeadFunction, head_params
            if mel is None:
                with torch.no_grad():
                    mel = self._mel_of(audios)
            embed, embed_pad = self._encoder_train_features(mel, get_orig_len)
            align_logit = transcribe_logit = None
            both = self.train_alignment and self.train_transcript and y_in is not None
            # (the GRU time-out flags of this branch are parked and read before its gradients are used: by EncoderFunction.backward in a plain
            # `loss.backward(); optimizer.step()` loop, by FineTuner after its backward -- head_train.CALLER_CHECKS_FLAGS)
            if both and embed.is_cuda and getattr(self, "_branch_streams", True) and BRANCH_STREAMS:
                # The two branches are independent until their losses, and the head's GRU sweeps are 2 x 1500 dependent steps on 12
                # workgroups (13 ms forward, 19 ms backward with the other 244 CUs idle): the head runs on a stream of its own beside
                # the decoder.  autograd runs a node's backward on the stream of its forward and joins the streams where gradients
                # meet (the encoder output), so the backward sweeps overlap the decoder's backward the same way.
                cur = torch.cuda.current_stream(embed.device)
                side = getattr(self, "_head_stream", None)
                if side is None or side.device != embed.device:
                    side = self._head_stream = torch.cuda.Stream(device=embed.device)
                from .. import head_train
                head_train.check_deferred_flags()                           # (of an earlier forward whose backward never ran)
                side.wait_stream(cur)
                head_train.DEFER_FLAG_CHECKS = True                         # the GRU time-out flags are read after the backward: no host
                try:                                                        # synchronisation between the two branches' launches
                    with torch.cuda.stream(side):
                        align_logit = HeadFunction.apply(embed, float(self.align_rnn.rnn.dropout), True, *head_params(self.align_rnn))
                finally:
                    head_train.DEFER_FLAG_CHECKS = False
                embed.record_stream(side)
                transcribe_logit = self._decoder_train_logits(y_in, embed_pad)
                cur.wait_stream(side)
                align_logit.record_stream(cur)
                return align_logit, transcribe_logit
            if self.train_alignment:
                align_logit = HeadFunction.apply(embed, float(self.align_rnn.rnn.dropout), True, *head_params(self.align_rnn))
            if self.train_transcript and y_in is not None:
                transcribe_logit = self._decoder_train_logits(y_in, embed_pad)
            return align_logit, transcribe_logit
        with torch.no_grad():                                               # frozen encoder: forward only
            mel = self._mel_of(audios)
            eng = self.engine()
            feats, B, T, stride = self._features(mel, get_orig_len)
        align_logit = None
        if self.train_alignment:
            if train:
                align_logit = self._head_train_logits(feats, B, T, stride)
            else:
                align_logit = eng.logits(feats, B, T, stride)               # (:106-107, :114-115)
        transcribe_logit = None
        if self.train_transcript and y_in is not None:                      # (:118-121): decoder over embed_pad = first 1500 frames
            embed_pad = feats.view(B, -1, eng.enc.d)[:, :N_CTX].contiguous().view(B * N_CTX, eng.enc.d)
            if train and any(p.requires_grad for p in self.whisper_model.decoder.parameters()):
                transcribe_logit = self._decoder_train_logits(y_in, embed_pad.view(B, N_CTX, -1).float())
            else:
                transcribe_logit = eng.decode(y_in, embed_pad, N_CTX)
        return align_logit, transcribe_logit

    def forward(self, mel, y_in=None):
        if self._wants_grad():                                              # training through forward(): (:135-149)
            from ..encoder_train import EncoderFunction, encoder_params
            from ..head_train import HeadFunction, head_params
            enc = self.whisper_model.encoder
            mel = mel.to(self._device())
            if self.freeze_encoder or self._encoder_frozen():
                with torch.no_grad():
                    embed = self._embed_audio(mel)
            else:
                embed = EncoderFunction.apply(mel, enc.positional_embedding, self._n_head(), *encoder_params(enc))
            align_logit = transcribe_logit = None
            if self.train_alignment:
                align_logit = HeadFunction.apply(embed, float(self.align_rnn.rnn.dropout), True, *head_params(self.align_rnn))
            if self.train_transcript and y_in is not None:
                transcribe_logit = self._decoder_train_logits(y_in, embed)
            return align_logit, transcribe_logit
        eng = self.engine()
        feats = eng.encode(mel)                                             # (:135-139)
        B = mel.shape[0]
        align_logit = eng.logits(feats, B, N_CTX, N_CTX) if self.train_alignment else None
        transcribe_logit = None
        if self.train_transcript and y_in is not None:
            transcribe_logit = eng.decode(y_in, feats, N_CTX)
        return align_logit, transcribe_logit

    # ------------------------------------------------------------------ fused fast path (addition)
    @torch.no_grad()
    def align(self, audios: Optional[Sequence[np.ndarray]] = None, labels=None, *, mel: Optional[torch.Tensor] = None,
              use_ctc: bool = True, hop_size_second: float = 0.02, get_orig_len: bool = True, return_frames: bool = False):
        """audios (or a ready mel) + class-id labels ([B,Lmax] with -100 padding, or list of lists) ->
        list[B] of list[L] of [onset_s, offset_s], exactly what perform_viterbi(_ctc)(frame_manual_forward(...))
        returns in the reference -- but the [B,T,V] logits are never materialised and nothing leaves the GPU
        except the [L,2] integer frames."""
        from ..utils.alignment import _labels_to_device, _seconds_from_frames
        eng = self.engine()
        if mel is None:
            mel = self._mel_of(audios)
        feats, B, T, stride = self._features(mel.to(eng.device), get_orig_len)
        lab_dev, n_lab, lab_lists = _labels_to_device(labels, B, eng.device)
        onset, offset, score, status = eng.align_feats_checked(feats, B, T, stride, lab_dev, n_lab,
                                                               _lib.LA_VARIANT_CTC if use_ctc else _lib.LA_VARIANT_PLAIN)
        if return_frames:
            return onset, offset, score, status
        return _seconds_from_frames(onset, offset, status, lab_lists, hop_size_second)


def encoder_only_engine(whisper_model, mel: torch.Tensor) -> torch.Tensor:
    """embed_audio for a bare whisper_compat.Whisper that is not wrapped in an AlignModel (float32 compute)."""
    _lib.require_gpu()
    cache = getattr(whisper_model, "_la_engine", None)
    from ..encoder_train import _EPOCH
    key = (_EPOCH[0],) + tuple((p._version, p.data_ptr()) for p in whisper_model.encoder.parameters())
    if cache is None or cache[0] != key:
        dev = torch.device(f"cuda:{torch.cuda.current_device()}")
        sd = {"encoder." + k: v for k, v in whisper_model.encoder.state_dict().items()}
        n_head = int(whisper_model.dims.n_audio_head)
        cache = (key, AlignEngine(pack_encoder(sd, n_head, torch.float32, dev), None, dev))
        whisper_model._la_engine = cache
    eng = cache[1]
    return eng.encode(mel, out_dtype=torch.float32).view(mel.shape[0], N_CTX, eng.enc.d).clone()


def decoder_engine_of(whisper_model) -> AlignEngine:
    """The float32 encoder + decoder engine of a bare whisper_compat.Whisper (packed once, re-packed when parameters change)."""
    _lib.require_gpu()
    if whisper_model.decoder is None:
        raise RuntimeError("this Whisper object was built without a decoder")
    cache = getattr(whisper_model, "_la_dec_engine", None)
    from ..encoder_train import _EPOCH
    key = (_EPOCH[0],) + tuple((p._version, p.data_ptr()) for p in whisper_model.parameters())
    if cache is None or cache[0] != key:
        dev = torch.device(f"cuda:{torch.cuda.current_device()}")
        enc_sd = {"encoder." + k: v for k, v in whisper_model.encoder.state_dict().items()}
        dec_sd = {"decoder." + k: v for k, v in whisper_model.decoder.state_dict().items()}
        eng = AlignEngine(pack_encoder(enc_sd, int(whisper_model.dims.n_audio_head), torch.float32, dev), None, dev,
                          dec=pack_decoder(dec_sd, int(whisper_model.dims.n_text_head), torch.float32, dev))
        cache = (key, eng)
        whisper_model._la_dec_engine = cache
    return cache[1]


def decoder_engine(whisper_model, tokens: torch.Tensor, audio_features: torch.Tensor, greedy=None) -> torch.Tensor:
    """Whisper.logits for a bare whisper_compat.Whisper (float32 compute): packs encoder + decoder weights once.
    greedy = (max_new_tokens, eot): run the greedy token loop from the prompt `tokens` instead and return the tokens."""
    eng = decoder_engine_of(whisper_model)
    B, n_audio, d = audio_features.shape
    xa = audio_features.to(device=eng.device, dtype=torch.float32).contiguous().view(B * n_audio, d)
    if greedy is not None and len(greedy) == 3:
        return eng.decode_beam(tokens, xa, greedy[2], greedy[0], greedy[1], n_audio=n_audio)
    if greedy is not None:
        return eng.decode_greedy(tokens, xa, greedy[0], greedy[1], n_audio)
    return eng.decode(tokens, xa, n_audio)
