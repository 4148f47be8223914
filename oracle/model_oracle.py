"""oracle/model_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU fp32 (torch / numpy) restatement of the floating-point half of the
reference's AlignModel path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product package
(lyricalignment_amd/) never does.

What each function follows (paths relative to the reference checkout):

* frame_count, chunk_plan        module/align_model.py:86-105 (Python round())
* gru_head_forward               module/align_model.py:11-40  (nn.GRU -> Mish -> Linear)
* emission_prep_ctc              utils/alignment.py:123-134
* emission_prep_plain            utils/alignment.py:14-20
* get_mae                        utils/alignment.py:190-199
* ce_loss / ctc_loss             train_multitask.py:587-633
* log_mel_spectrogram, pad_or_trim, sinusoids, encoder_forward, decoder_forward
    third-party `openai-whisper` (requirements.txt:6, UN-PINNED, not vendored,
    not installed here).  Call sites: module/align_model.py:84,89,91,100-101,
    109,112,120.  Restated from its published algorithm (whisper/audio.py,
    whisper/model.py).

Parity status
-------------
* head / emission prep / losses / frame bookkeeping: PINNED against the
  reference's own classes and functions imported in the build container
  (tests/golden/gen_golden.py -> tests/golden/*.npz|json).
* log-mel + encoder + decoder: **PARITY UNPINNED** against openai-whisper
  itself (absent, no golden vectors in the reference).  Cross-checked instead
  against the independent HF `transformers` Whisper implementation that is
  installed in this image (tests/test_oracle_model.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SAMPLE_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
N_MELS = 80
N_FRAMES = 3000  # whisper.audio.N_FRAMES, used at module/align_model.py:87-105
N_CTX = 1500

WHISPER_DIMS = {  # (n_state, n_head, n_layer); train_multitask.py:145-149 lists n_state only
    "tiny": (384, 6, 4),
    "base": (512, 8, 6),
    "small": (768, 12, 12),
    "medium": (1024, 16, 24),
    "large": (1280, 20, 32),
    "large-v2": (1280, 20, 32),
}


# --------------------------------------------------------------------------- #
# log-mel front end (whisper/audio.py)                                         #
# --------------------------------------------------------------------------- #
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log_part = min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep
    return np.where(f >= min_log_hz, log_part, mels)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filters(n_mels: int = N_MELS, n_fft: int = N_FFT, sr: int = SAMPLE_RATE) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels) (Slaney scale, Slaney area norm) ->
    float32 [n_mels, n_fft//2+1]; openai-whisper ships this table as
    assets/mel_filters.npz."""
    n_freq = n_fft // 2 + 1
    fftfreqs = np.linspace(0.0, sr / 2.0, n_freq)
    mel_pts = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(sr / 2.0), n_mels + 2)
    hz_pts = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, n_freq), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2 : n_mels + 2] - hz_pts[:n_mels])
    weights *= enorm[:, None]
    return weights.astype(np.float32)


def log_mel_spectrogram(audio) -> torch.Tensor:
    """whisper.audio.log_mel_spectrogram: [.., N] float32 waveform -> [.., 80, N//160].
    NOTE the -8 floor uses the max over the WHOLE tensor (batch included)."""
    if not torch.is_tensor(audio):
        audio = torch.from_numpy(np.asarray(audio))
    audio = audio.to(torch.float32)
    window = torch.hann_window(N_FFT)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, window=window, return_complex=True)
    magnitudes = stft[..., :-1].abs() ** 2
    filters = torch.from_numpy(mel_filters())
    mel_spec = filters @ magnitudes
    log_spec = torch.clamp(mel_spec, min=1e-10).log10()
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)
    log_spec = (log_spec + 4.0) / 4.0
    return log_spec


def pad_or_trim(array: torch.Tensor, length: int = N_FRAMES, axis: int = -1) -> torch.Tensor:
    """whisper.audio.pad_or_trim (torch branch)."""
    if array.shape[axis] > length:
        array = array.index_select(dim=axis, index=torch.arange(length, device=array.device))
    if array.shape[axis] < length:
        pad_widths = [(0, 0)] * array.ndim
        pad_widths[axis] = (0, length - array.shape[axis])
        array = F.pad(array, [pad for sizes in pad_widths[::-1] for pad in sizes])
    return array


# --------------------------------------------------------------------------- #
# frame bookkeeping of AlignModel.frame_manual_forward                          #
# --------------------------------------------------------------------------- #
def frame_count(n_mel: int) -> int:
    """module/align_model.py:88,98 -- Python round() = banker's rounding."""
    return int(round(n_mel / 2.0))


def chunk_plan(n_mel: int) -> List[Tuple[int, int, int]]:
    """[(start, end, kept_frames)] for the mel of a clip (module/align_model.py:87-103)."""
    if n_mel <= N_FRAMES:
        return [(0, n_mel, frame_count(n_mel))]
    plan = []
    for start in range(0, n_mel, N_FRAMES):
        end = min(start + N_FRAMES, n_mel)
        plan.append((start, end, frame_count(end - start)))
    return plan


# --------------------------------------------------------------------------- #
# Whisper audio encoder / text decoder (whisper/model.py)                       #
# --------------------------------------------------------------------------- #
def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    assert channels % 2 == 0
    log_timescale_increment = np.log(max_timescale) / (channels // 2 - 1)
    inv_timescales = torch.exp(-log_timescale_increment * torch.arange(channels // 2))
    scaled_time = torch.arange(length)[:, np.newaxis] * inv_timescales[np.newaxis, :]
    return torch.cat([torch.sin(scaled_time), torch.cos(scaled_time)], dim=1)


def _mha(p: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, xa: Optional[torch.Tensor],
         n_head: int, mask: Optional[torch.Tensor]) -> torch.Tensor:
    q = F.linear(x, p[prefix + "query.weight"], p[prefix + "query.bias"])
    src = x if xa is None else xa
    k = F.linear(src, p[prefix + "key.weight"])  # no bias
    v = F.linear(src, p[prefix + "value.weight"], p[prefix + "value.bias"])
    n_batch, n_ctx, n_state = q.shape
    scale = (n_state // n_head) ** -0.25
    q = q.view(*q.shape[:2], n_head, -1).permute(0, 2, 1, 3) * scale
    k = k.view(*k.shape[:2], n_head, -1).permute(0, 2, 3, 1) * scale
    v = v.view(*v.shape[:2], n_head, -1).permute(0, 2, 1, 3)
    qk = q @ k
    if mask is not None:
        qk = qk + mask[:n_ctx, :n_ctx]
    # whisper: softmax(qk.float()).to(q.dtype); float64 inputs (the tests' high-precision reference runs) stay float64
    w = F.softmax(qk if qk.dtype == torch.float64 else qk.float(), dim=-1).to(q.dtype)
    out = (w @ v).permute(0, 2, 1, 3).flatten(start_dim=2)
    return F.linear(out, p[prefix + "out.weight"], p[prefix + "out.bias"])


def _block(p, prefix, x, xa, n_head, mask):
    d = x.shape[-1]
    x = x + _mha(p, prefix + "attn.", F.layer_norm(x, (d,), p[prefix + "attn_ln.weight"],
                                                  p[prefix + "attn_ln.bias"]), None, n_head, mask)
    if xa is not None:
        x = x + _mha(p, prefix + "cross_attn.",
                     F.layer_norm(x, (d,), p[prefix + "cross_attn_ln.weight"],
                                  p[prefix + "cross_attn_ln.bias"]), xa, n_head, None)
    h = F.layer_norm(x, (d,), p[prefix + "mlp_ln.weight"], p[prefix + "mlp_ln.bias"])
    h = F.linear(h, p[prefix + "mlp.0.weight"], p[prefix + "mlp.0.bias"])
    h = F.gelu(h)
    h = F.linear(h, p[prefix + "mlp.2.weight"], p[prefix + "mlp.2.bias"])
    return x + h


def encoder_forward(p: Dict[str, torch.Tensor], mel: torch.Tensor, n_head: int,
                    prefix: str = "encoder.") -> torch.Tensor:
    """whisper.model.AudioEncoder.forward == Whisper.embed_audio.
    mel [B, 80, 3000] -> [B, 1500, d].  `p` uses openai-whisper key names."""
    x = F.gelu(F.conv1d(mel, p[prefix + "conv1.weight"], p[prefix + "conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, p[prefix + "conv2.weight"], p[prefix + "conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1)
    pos = p.get(prefix + "positional_embedding")
    if pos is None:
        pos = sinusoids(x.shape[1], x.shape[2])
    assert x.shape[1:] == pos.shape, "incorrect audio shape"
    x = x + pos
    n_layer = 0
    while f"{prefix}blocks.{n_layer}.attn.query.weight" in p:
        n_layer += 1
    for i in range(n_layer):
        x = _block(p, f"{prefix}blocks.{i}.", x, None, n_head, None)
    d = x.shape[-1]
    return F.layer_norm(x, (d,), p[prefix + "ln_post.weight"], p[prefix + "ln_post.bias"])


def decoder_forward(p: Dict[str, torch.Tensor], tokens: torch.Tensor, xa: torch.Tensor,
                    n_head: int, prefix: str = "decoder.") -> torch.Tensor:
    """whisper.model.TextDecoder.forward == Whisper.logits(tokens, audio_features)."""
    n_tok = tokens.shape[-1]
    x = p[prefix + "token_embedding.weight"][tokens] + p[prefix + "positional_embedding"][:n_tok]
    mask = torch.full((n_tok, n_tok), float("-inf")).triu_(1)
    n_layer = 0
    while f"{prefix}blocks.{n_layer}.attn.query.weight" in p:
        n_layer += 1
    for i in range(n_layer):
        x = _block(p, f"{prefix}blocks.{i}.", x, xa, n_head, mask)
    d = x.shape[-1]
    x = F.layer_norm(x, (d,), p[prefix + "ln.weight"], p[prefix + "ln.bias"])
    return (x @ p[prefix + "token_embedding.weight"].T).float()


def random_encoder_params(n_state: int, n_layer: int, seed: int = 0, n_mels: int = N_MELS,
                          std: float = 0.02, prefix: str = "encoder.") -> Dict[str, torch.Tensor]:
    """Seeded random-init encoder weights in openai-whisper key layout (no
    checkpoints are reachable offline).  LN gamma=1+noise, beta=noise so LN
    parameters are exercised."""
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, s=std):
        return torch.randn(*shape, generator=g) * s

    p = {
        prefix + "conv1.weight": rn(n_state, n_mels, 3, s=0.05),
        prefix + "conv1.bias": rn(n_state),
        prefix + "conv2.weight": rn(n_state, n_state, 3),
        prefix + "conv2.bias": rn(n_state),
        prefix + "positional_embedding": sinusoids(N_CTX, n_state),
        prefix + "ln_post.weight": 1.0 + rn(n_state, s=0.1),
        prefix + "ln_post.bias": rn(n_state, s=0.1),
    }
    for i in range(n_layer):
        b = f"{prefix}blocks.{i}."
        for name in ("query", "value", "out"):
            p[b + f"attn.{name}.weight"] = rn(n_state, n_state)
            p[b + f"attn.{name}.bias"] = rn(n_state)
        p[b + "attn.key.weight"] = rn(n_state, n_state)
        p[b + "attn_ln.weight"] = 1.0 + rn(n_state, s=0.1)
        p[b + "attn_ln.bias"] = rn(n_state, s=0.1)
        p[b + "mlp.0.weight"] = rn(4 * n_state, n_state)
        p[b + "mlp.0.bias"] = rn(4 * n_state)
        p[b + "mlp.2.weight"] = rn(n_state, 4 * n_state)
        p[b + "mlp.2.bias"] = rn(n_state)
        p[b + "mlp_ln.weight"] = 1.0 + rn(n_state, s=0.1)
        p[b + "mlp_ln.bias"] = rn(n_state, s=0.1)
    return p


# --------------------------------------------------------------------------- #
# BiGRU -> Mish -> Linear head (module/align_model.py:11-40)                    #
# --------------------------------------------------------------------------- #
def _gru_direction(x: torch.Tensor, w_ih, w_hh, b_ih, b_hh, reverse: bool) -> torch.Tensor:
    """One direction of one nn.GRU layer, batch_first.  Gate order r, z, n;
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1 - z) * n + z * h."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gi = F.linear(x, w_ih, b_ih)  # [B, T, 3H]
    h = x.new_zeros(B, H)
    out = x.new_empty(B, T, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gh = F.linear(h, w_hh, b_hh)
        i_r, i_z, i_n = gi[:, t].chunk(3, dim=1)
        h_r, h_z, h_n = gh.chunk(3, dim=1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h = (1.0 - z) * n + z * h
        out[:, t] = h
    return out


def gru_head_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, num_layers: int = 2,
                     bidirectional: bool = True, prefix: str = "align_rnn.",
                     return_hidden: bool = False) -> torch.Tensor:
    """RNN.forward in eval mode (inter-layer dropout inactive): [B,T,d] -> [B,T,V]."""
    h = x
    for layer in range(num_layers):
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)):
            if rev and not bidirectional:
                continue
            outs.append(_gru_direction(
                h,
                p[f"{prefix}rnn.weight_ih_l{layer}{suffix}"], p[f"{prefix}rnn.weight_hh_l{layer}{suffix}"],
                p[f"{prefix}rnn.bias_ih_l{layer}{suffix}"], p[f"{prefix}rnn.bias_hh_l{layer}{suffix}"], rev))
        h = torch.cat(outs, dim=2)
    act = h * torch.tanh(F.softplus(h))  # nn.Mish
    if return_hidden:
        return act
    return F.linear(act, p[prefix + "fc.weight"], p[prefix + "fc.bias"])


def random_head_params(embed_dim: int, hidden: int, output_dim: int, seed: int = 1,
                       prefix: str = "align_rnn.") -> Dict[str, torch.Tensor]:
    """Seeded head weights with nn.GRU / nn.Linear default-init scale."""
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / math.sqrt(hidden)

    def un(*shape, bound):
        return (torch.rand(*shape, generator=g) * 2 - 1) * bound

    p = {}
    for layer in range(2):
        in_dim = embed_dim if layer == 0 else 2 * hidden
        for suffix in ("", "_reverse"):
            p[f"{prefix}rnn.weight_ih_l{layer}{suffix}"] = un(3 * hidden, in_dim, bound=k)
            p[f"{prefix}rnn.weight_hh_l{layer}{suffix}"] = un(3 * hidden, hidden, bound=k)
            p[f"{prefix}rnn.bias_ih_l{layer}{suffix}"] = un(3 * hidden, bound=k)
            p[f"{prefix}rnn.bias_hh_l{layer}{suffix}"] = un(3 * hidden, bound=k)
    kf = 1.0 / math.sqrt(2 * hidden)
    p[prefix + "fc.weight"] = un(output_dim, 2 * hidden, bound=kf)
    p[prefix + "fc.bias"] = un(output_dim, bound=kf)
    return p


# --------------------------------------------------------------------------- #
# emission prep (utils/alignment.py:14-20, 123-134)                             #
# --------------------------------------------------------------------------- #
def emission_prep_ctc(prediction: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (log_prediction [B,T,V-2], log_silence [B,T,1]); naive log(1-sigmoid)."""
    log_prediction = F.log_softmax(prediction[:, :, 1:-1], dim=2)
    silence_prediction = torch.sigmoid(prediction[:, :, -1:])
    voiced_prediction = 1.0 - silence_prediction
    log_silence_prediction = torch.log(silence_prediction)
    log_voiced_prediction = torch.log(voiced_prediction)
    log_prediction = torch.clip(log_prediction + log_voiced_prediction, min=-1000)
    log_silence_prediction = torch.clip(log_silence_prediction, min=-1000)
    return log_prediction, log_silence_prediction


def emission_prep_plain(prediction: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (log_prediction [B,T,V-1], log_silence [B,T,1])."""
    log_prediction = F.log_softmax(prediction, dim=2)
    silence_prediction = log_prediction[:, :, 0:1]
    log_prediction = torch.clip(log_prediction, min=-1000)[:, :, 1:]
    log_silence_prediction = torch.clip(silence_prediction, min=-1000)
    return log_prediction, log_silence_prediction


def get_mae(gt, predict) -> float:
    """utils/alignment.py:190-199."""
    error = 0.0
    cnt = 0
    for i in range(len(gt)):
        for j in range(len(gt[i])):
            error = error + abs(gt[i][j][0] - predict[i][j][0]) + abs(gt[i][j][1] - predict[i][j][1])
            cnt = cnt + 2.0
    return error / cnt


# --------------------------------------------------------------------------- #
# fine-tune losses (train_multitask.py:587-633)                                 #
# --------------------------------------------------------------------------- #
def ce_loss(logits: torch.Tensor, frame_labels: torch.Tensor, vocab_size: int = 21128) -> torch.Tensor:
    """compute_ce_loss(compute_sil=True): frame CE over columns 1..vocab_size-1 plus
    BCE-with-logits of column vocab_size against (label == -100).  Does not
    mutate its argument (the reference does, :607)."""
    T = logits.shape[1]
    fl = frame_labels[:, :T].clone()
    if fl.shape[1] < T:
        fl = torch.cat((fl, torch.full((fl.shape[0], T - fl.shape[1]), -100, dtype=fl.dtype)), dim=1)
    fl[fl != -100] -= 1
    word = F.cross_entropy(logits[:, :, 1:vocab_size].transpose(1, 2), fl)
    sil_label = torch.where(fl == -100, 1, 0).to(logits.dtype)
    sil = F.binary_cross_entropy_with_logits(logits[:, :, vocab_size], sil_label)
    return word + sil


def ctc_loss(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """compute_ctc_loss: explicit alpha recursion (blank 0, mean over batch of
    nll / target_len), equivalent to F.ctc_loss defaults used at :632."""
    lsm = F.log_softmax(logits.double(), dim=2)
    B, T, _ = lsm.shape
    total = lsm.new_zeros(())
    for b in range(B):
        y = [int(v) for v in labels[b].tolist() if v != -100]
        L = len(y)
        ext = [0]
        for v in y:
            ext += [v, 0]
        S = len(ext)
        neg = float("-inf")
        alpha = [lsm.new_tensor(neg)] * S
        alpha[0] = lsm[b, 0, 0]
        if S > 1:
            alpha[1] = lsm[b, 0, ext[1]]
        for t in range(1, T):
            new = [lsm.new_tensor(neg)] * S
            for s in range(S):
                cands = [alpha[s]]
                if s >= 1:
                    cands.append(alpha[s - 1])
                if s >= 2 and ext[s] != 0 and ext[s] != ext[s - 2]:
                    cands.append(alpha[s - 2])
                cands = [c for c in cands if c.item() != neg]  # keep autograd NaN-free
                if cands:
                    new[s] = torch.logsumexp(torch.stack(cands), 0) + lsm[b, t, ext[s]]
            alpha = new
        tail = [c for c in ([alpha[S - 1], alpha[S - 2]] if S > 1 else [alpha[S - 1]]) if c.item() != neg]
        nll = -torch.logsumexp(torch.stack(tail), 0) if tail else lsm.new_tensor(float("inf"))
        total = total + nll / max(L, 1)
    return (total / B).float()
