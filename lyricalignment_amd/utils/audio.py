"""Drop-in for the reference's utils/audio.py (load_audio_file, :3-20): decode an audio file, resample to 16 kHz,
select mono / (L+R)/2 / channel 1.  Output contract (what enters the hot path): {"speech": float32 [N], "sampling_rate": 16000}.

The reference delegates to librosa.load (absent here, un-pinned there), whose resampler is soxr_hq or kaiser_best
depending on the librosa version -- parity with IT is unpinned.  This build decodes WAV / AIFF / AU on the host (I/O) and
resamples on the device with a Kaiser-windowed-sinc polyphase FIR (la_resample_poly_f32) designed exactly like
scipy.signal.resample_poly, which is the checker the tests use.  No CPU fallback for the resampling arithmetic.
"""
from __future__ import annotations

import math
from fractions import Fraction

import numpy as np
import torch

from .. import _lib
from .._lib import check, lib, ptr, stream_ptr

TARGET_SR = 16000


def _design(up: int, down: int):
    """scipy.signal.resample_poly's default filter: firwin(2*half+1, 1/max_rate, window=('kaiser', 5.0)) * up, padded so
    the phase delay is a whole number of output samples.  Constant table construction (like the mel filter bank)."""
    max_rate = max(up, down)
    half_len = 10 * max_rate
    numtaps = 2 * half_len + 1
    m = np.arange(numtaps, dtype=np.float64) - half_len
    fc = 1.0 / max_rate
    h = fc * np.sinc(fc * m) * np.kaiser(numtaps, 5.0)
    h /= h.sum()
    h *= up
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    h = np.concatenate([np.zeros(n_pre_pad), h]).astype(np.float32)
    return h, n_pre_remove


def resample_to_16k(x: np.ndarray, sr: int, device="cuda") -> np.ndarray:
    """float32 [N] at `sr` Hz -> float32 [ceil(N*16000/sr)] at 16 kHz (device polyphase FIR)."""
    if sr == TARGET_SR:
        return np.ascontiguousarray(x, dtype=np.float32)
    _lib.require_gpu()
    fr = Fraction(TARGET_SR, int(sr))
    up, down = fr.numerator, fr.denominator
    h, skip = _design(up, down)
    n_in = x.shape[0]
    n_out = int(math.ceil(n_in * up / down))
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)
    hd = torch.from_numpy(h).to(device)
    y = torch.empty((n_out,), dtype=torch.float32, device=device)
    check(lib().la_resample_poly_f32(ptr(xd), n_in, ptr(hd), h.shape[0], up, down, skip, ptr(y), n_out, stream_ptr()), "resample_poly")
    return y.cpu().numpy()


def _pcm_to_float(raw: bytes, width: int, n_channels: int, big_endian: bool) -> np.ndarray:
    """interleaved signed PCM of `width` bytes per sample -> float32 [channels, N] in [-1, 1)"""
    if width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        hi, mid, lo = (b[:, 0], b[:, 1], b[:, 2]) if big_endian else (b[:, 2], b[:, 1], b[:, 0])
        v = (hi << 16) | (mid << 8) | lo
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
    else:
        v = np.frombuffer(raw, dtype=np.dtype({1: "i1", 2: "i2", 4: "i4"}[width]).newbyteorder(">" if big_endian else "<"))
    x = v.astype(np.float64) / float(1 << (8 * width - 1))
    return np.ascontiguousarray(x.reshape(-1, n_channels).T.astype(np.float32))


def _decode(file: str):
    """-> (float32 [channels, N] in [-1, 1], sample rate).  By the file's magic bytes: RIFF WAV (PCM 8/16/24/32-bit, float32/64)
    via scipy.io.wavfile; AIFF / AIFF-C (big-endian PCM, 'sowt', u-law / a-law) and Sun AU (PCM 8/16/24/32-bit, u-law) via the
    standard library's aifc / sunau.  Compressed formats (FLAC, MP3, Ogg) need a decoder this image does not have -- librosa
    would hand them to libsndfile / audioread (utils/audio.py:3-20)."""
    with open(file, "rb") as f:
        magic = f.read(12)
    if magic[:4] == b"FORM" and magic[8:12] in (b"AIFF", b"AIFC"):
        import aifc
        with aifc.open(file, "rb") as a:
            nch, width, sr, n = a.getnchannels(), a.getsampwidth(), a.getframerate(), a.getnframes()
            raw = a.readframes(n)                       # PCM and 'sowt' come back big-endian; u-law / a-law are expanded by
            companded = a.getcomptype() in (b"ulaw", b"ULAW", b"alaw", b"ALAW")   # audioop to NATIVE-endian 16-bit samples
        import sys
        return _pcm_to_float(raw, width, nch, big_endian=(sys.byteorder == "big") if companded else True), int(sr)
    if magic[:4] == b".snd":
        import sunau
        with sunau.open(file, "rb") as a:
            nch, width, sr, n = a.getnchannels(), a.getsampwidth(), a.getframerate(), a.getnframes()
            raw = a.readframes(n)                       # u-law arrives as 16-bit linear in native byte order, PCM as stored (big-endian)
            ulaw = a.getcomptype() == "ULAW"
        import sys
        return _pcm_to_float(raw, width, nch, big_endian=(sys.byteorder == "big") if ulaw else True), int(sr)
    if magic[:4] not in (b"RIFF", b"RIFX", b"RF64"):
        raise ValueError(f"{file}: not a WAV / AIFF / AU file (magic {magic[:4]!r}); compressed formats need a decoder that is not in this build")
    from scipy.io import wavfile
    sr, data = wavfile.read(file)
    if data.ndim == 1:
        data = data[:, None]
    if data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    elif data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    else:
        x = data.astype(np.float32)
    return np.ascontiguousarray(x.T), int(sr)


def load_audio_file(file, audio_type: int = 0):
    # audio_type: 0 => mono; 1 => mixture; 2 => mixture, but vocal only          (utils/audio.py:4)
    if audio_type not in (0, 1, 2):
        raise ValueError("audio_type must be 0, 1, or 2")
    x, sr = _decode(file)
    if audio_type == 0:
        x = x.mean(axis=0, keepdims=True)                                        # librosa.to_mono
    chans = [resample_to_16k(c, sr) for c in x]
    batch = {}
    if audio_type == 0:
        batch["speech"] = chans[0]
    elif audio_type == 1:
        batch["speech"] = (chans[0] + chans[1]) / 2
    else:
        batch["speech"] = chans[1]
    batch["sampling_rate"] = TARGET_SR
    return batch
