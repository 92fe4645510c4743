"""WAV ingest for the hot path: the libsndfile decode that `librosa.load(path,
sr=sr, mono=True)` performs at latent_space_exploration/map_detector_core.py:210,
for PCM16 files already at the requested rate (resampling = SURVEY §8(f) rank 4,
not built).  Mono files go through libamph's native reader; anything else
through the stdlib parser + a channel mean, as float32."""
from __future__ import annotations

import ctypes as C
import wave
from pathlib import Path
from typing import Sequence, Tuple

import numpy as np

from . import _native as N


def read_pcm16_mono(path: Path, sr: int, max_samples: int) -> Tuple[np.ndarray, int]:
    """Native reader: (int16 [max_samples] zero-filled beyond the file, n_valid)."""
    out = np.empty(max_samples, dtype=np.int16)
    n = C.c_int32(0)
    st = N.lib.amph_wav_read_pcm16(str(path).encode(), int(sr), out.ctypes.data_as(C.c_void_p), int(max_samples),
                                   C.byref(n))
    if st == N.AMPH_ERR_UNSUPPORTED:
        raise NotImplementedError(f"{path}: not mono PCM16 at {sr} Hz")
    if st != N.AMPH_OK:
        raise OSError(f"{path}: unreadable WAV ({N.STATUS_NAMES.get(st, st)})")
    return out, int(n.value)


def load_float_mono(path: Path, sr: int) -> np.ndarray:
    """float32 mono signal as librosa.load returns it (int16 / 32768, channel mean)."""
    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2 or w.getcomptype() != "NONE":
            raise RuntimeError(f"{path}: only PCM16 WAV is supported")
        if w.getframerate() != sr:
            raise RuntimeError(f"{path}: sample rate {w.getframerate()} != {sr} (resampling is not implemented)")
        ch = w.getnchannels()
        q = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, ch)
    y = q.astype(np.float32) / np.float32(32768.0)
    return np.ascontiguousarray(y.mean(axis=1, dtype=np.float32) if ch > 1 else y[:, 0])


def read_many_pcm16(paths: Sequence[Path], sr: int, max_samples: int, n_threads: int = 8, pinned: bool = False):
    """Batch ingest with libamph's host thread pool.  Returns (pcm int16 [n, max_samples],
    lengths int32 [n], status int32 [n], pinned_ptr); rows of failed files are zero and
    flagged.  With pinned=True the rows live in hipHostMalloc memory: release it with
    free_pinned(pinned_ptr) once `pcm` is no longer referenced."""
    n = len(paths)
    if pinned and n > 0:
        nbytes = n * max_samples * 2
        ptr = N.lib.amph_host_alloc(nbytes)
        if not ptr:
            raise MemoryError("hipHostMalloc failed")
        buf = (C.c_int16 * (n * max_samples)).from_address(ptr)
        pcm = np.frombuffer(buf, dtype=np.int16).reshape(n, max_samples)
    else:
        ptr = None
        pcm = np.zeros((n, max_samples), dtype=np.int16)
    lengths = np.zeros(n, dtype=np.int32)
    status = np.zeros(n, dtype=np.int32)
    if n:
        arr = (C.c_char_p * n)(*[str(p).encode() for p in paths])
        st = N.lib.amph_wav_read_many(arr, n, int(sr), pcm.ctypes.data_as(C.c_void_p), int(max_samples),
                                      lengths.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p),
                                      int(n_threads))
        if st != N.AMPH_OK:
            raise RuntimeError(f"amph_wav_read_many failed: {N.STATUS_NAMES.get(st, st)}")
    return pcm, lengths, status, ptr


def free_pinned(ptr) -> None:
    if ptr:
        N.lib.amph_host_free(ptr)
