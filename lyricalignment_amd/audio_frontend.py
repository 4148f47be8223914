"""Host-side constants + plumbing of the log-mel front end (whisper.audio, call site
module/align_model.py:84).  The arithmetic (STFT as a DFT-GEMM, mel projection, log, global-max
floor) runs in la_logmel_f32 on the device; this file only builds the two constant tables
openai-whisper ships as assets (Hann window, Slaney mel filter bank) and moves buffers."""
from __future__ import annotations

import ctypes
import functools
from typing import Sequence, Union

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr, stream_ptr

SAMPLE_RATE, N_FFT, HOP_LENGTH, N_MELS = 16000, 400, 160, 80


def _slaney_hz_to_mel(f: np.ndarray) -> np.ndarray:
    f = np.asarray(f, dtype=np.float64)
    lin = f * 3.0 / 200.0
    log_region = 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) * (27.0 / np.log(6.4))
    return np.where(f >= 1000.0, log_region, lin)


def _slaney_mel_to_hz(m: np.ndarray) -> np.ndarray:
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((m - 15.0) * (np.log(6.4) / 27.0)), m * 200.0 / 3.0)


@functools.lru_cache(maxsize=4)
def mel_filter_table(n_mels: int = N_MELS) -> np.ndarray:
    """The table openai-whisper stores in assets/mel_filters.npz: librosa.filters.mel(sr=16000, n_fft=400,
    n_mels) = triangular filters on the Slaney mel scale with Slaney (area) normalisation, float32 [n_mels, 201]."""
    n_freq = N_FFT // 2 + 1
    freqs = np.linspace(0.0, SAMPLE_RATE / 2.0, n_freq)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(0.0), _slaney_hz_to_mel(SAMPLE_RATE / 2.0), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    fb = np.maximum(0.0, np.minimum(-ramps[:-2] / width[:-1, None], ramps[2:] / width[1:, None]))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return fb.astype(np.float32)


@functools.lru_cache(maxsize=None)
def _device_tables(device_index: int):
    dev = torch.device("cuda", device_index)
    filt = torch.from_numpy(mel_filter_table()).to(dev)
    win = torch.hann_window(N_FFT, periodic=True, dtype=torch.float32).to(dev)
    return filt, win


@functools.lru_cache(maxsize=None)
def _device_constants(device_index: int) -> torch.Tensor:
    """The log-mel kernel's constants (windowed DFT matrix in fragment order + padded filter bank), built once per device
    from the two tables above (la_logmel_constants) -- whisper caches its filter asset per process the same way."""
    dev = torch.device("cuda", device_index)
    filt, win = _device_tables(device_index)
    need = ctypes.c_size_t(0)
    check(lib().la_logmel_constants_bytes(ctypes.byref(need)), "logmel_constants_bytes")
    consts = torch.empty((need.value,), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        check(lib().la_logmel_constants(ptr(filt), ptr(win), ptr(consts), need.value, stream_ptr()), "logmel_constants")
        torch.cuda.current_stream().synchronize()      # other streams may use the buffer from now on
    return consts


def log_mel_spectrogram(audio: Union[np.ndarray, torch.Tensor], device="cuda") -> torch.Tensor:
    """[.., N] float waveform(s) -> [.., 80, N // 160] float32 on `device` (device log-mel kernel).
    Same contract as whisper.audio.log_mel_spectrogram incl. the whole-tensor max for the -8 floor."""
    _lib.require_gpu()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.LyricAlignHipError("log_mel_spectrogram runs on the MI355X only (no CPU fallback)")
    if not torch.is_tensor(audio):
        audio = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
    squeeze = audio.dim() == 1
    a = audio.reshape(-1, audio.shape[-1]).to(device=dev, dtype=torch.float32).contiguous()
    B, N = a.shape
    frames = N // HOP_LENGTH
    consts = _device_constants(dev.index if dev.index is not None else torch.cuda.current_device())
    mel = torch.empty((B, N_MELS, frames), dtype=torch.float32, device=dev)
    need = ctypes.c_size_t(0)
    check(lib().la_logmel_workspace_bytes(B, N, ctypes.byref(need)), "logmel_workspace_bytes")
    ws = torch.empty((need.value,), dtype=torch.uint8, device=dev)
    check(lib().la_logmel_f32_prepared(ptr(a), B, N, ptr(consts), ptr(mel), mel.stride(0), mel.stride(1), ptr(ws), need.value,
                                       stream_ptr()), "logmel_f32_prepared")
    return mel[0] if squeeze else mel.reshape(*audio.shape[:-1], N_MELS, frames)
