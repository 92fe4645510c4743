"""WAV ingest for the hot path: what `librosa.load(path, sr=sr, mono=True)` does at
latent_space_exploration/map_detector_core.py:210 - libsndfile decode to float32, channel
mean, and (only when the file's rate differs from `sr`) resampy "kaiser_best" resampling.

  fast path   mono PCM16 at the requested rate: libamph's native reader (+ host thread pool)
  general     RIFF/WAVE with PCM 8/16/24/32-bit, IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE, any channel
              count: parsed here with numpy (`load_audio`), converted with libsndfile's float scaling,
              averaged over channels; a rate mismatch goes through the HIP resampler (K8,
              amph_resample_f32 - SURVEY 8(f) rank 4; parity unpinned)."""
from __future__ import annotations

import ctypes as C
import struct
import sys
from pathlib import Path
from typing import Optional, Sequence, Tuple

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


def load_audio(path: Path) -> Tuple[np.ndarray, int]:
    """(float32 mono signal, native sample rate): soundfile.read(dtype="float32") + librosa.to_mono.
    Integer PCM is scaled as libsndfile does (int16 / 2^15, int24 / 2^23, int32 / 2^31, uint8 -> (x - 128) / 128)."""
    data = Path(path).read_bytes()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise RuntimeError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, body = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        start = pos + 8
        if cid == b"fmt ":
            if size < 16:
                raise RuntimeError(f"{path}: short fmt chunk")
            tag, ch, rate, _, _, bits = struct.unpack_from("<HHIIHH", data, start)
            if tag == 0xFFFE and size >= 26:                       # WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
                tag = struct.unpack_from("<H", data, start + 24)[0]
            fmt = (tag, ch, rate, bits)
        elif cid == b"data":
            body = data[start:min(start + size, len(data))]
            break
        pos = start + size + (size & 1)
    if fmt is None or body is None:
        raise RuntimeError(f"{path}: missing fmt or data chunk")
    tag, ch, rate, bits = fmt
    if ch < 1 or rate < 1:
        raise RuntimeError(f"{path}: bad channel count / sample rate")
    nbytes = bits // 8
    n = len(body) // (nbytes * ch) * ch
    if tag == 1 and bits == 16:
        y = np.frombuffer(body, dtype="<i2", count=n).astype(np.float32) / np.float32(32768.0)
    elif tag == 1 and bits == 8:
        y = (np.frombuffer(body, dtype=np.uint8, count=n).astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif tag == 1 and bits == 24:
        b = np.frombuffer(body, dtype=np.uint8, count=n * 3).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        y = v.astype(np.float32) / np.float32(8388608.0)
    elif tag == 1 and bits == 32:
        y = (np.frombuffer(body, dtype="<i4", count=n).astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 3 and bits == 32:
        y = np.frombuffer(body, dtype="<f4", count=n).astype(np.float32)
    elif tag == 3 and bits == 64:
        y = np.frombuffer(body, dtype="<f8", count=n).astype(np.float32)
    else:
        raise RuntimeError(f"{path}: unsupported WAV encoding (format tag {tag}, {bits} bits)")
    y = y.reshape(-1, ch)
    return np.ascontiguousarray(y.mean(axis=1, dtype=np.float32) if ch > 1 else y[:, 0]), int(rate)


def load_float_mono(path: Path, sr: int, session=None) -> np.ndarray:
    """float32 mono signal at `sr` as librosa.load returns it.  A file at another rate is resampled on the GPU by
    `session` (an AmphSession); without one it raises."""
    y, rate = load_audio(path)
    if rate != sr:
        if session is None:
            raise RuntimeError(f"{path}: sample rate {rate} != {sr} and no GPU session was given to resample with")
        y = session.resample(y, rate, sr).cpu().numpy()[0]
    return y


class PinnedRows:
    """Grow-only pinned (hipHostMalloc) staging slots for the batch ingest.  Pinning ~1 GB per call cost more than
    parsing the files (0.3 s per 20 000 one-second files): slots are allocated once per process and reused."""

    def __init__(self):
        self._slots = {}

    def rows(self, slot: int, n: int, max_samples: int) -> np.ndarray:
        need = max(1, n * max_samples) * 2
        ptr, cap = self._slots.get(slot, (None, 0))
        if cap < need:
            if ptr:
                N.lib.amph_host_free(ptr)
            cap = int(need * 1.25)
            ptr = N.lib.amph_host_alloc(cap)
            if not ptr:
                self._slots.pop(slot, None)
                raise MemoryError("hipHostMalloc failed")
            self._slots[slot] = (ptr, cap)
        buf = (C.c_int16 * (n * max_samples)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.int16).reshape(n, max_samples)

    def release(self):
        for ptr, _ in self._slots.values():
            N.lib.amph_host_free(ptr)
        self._slots = {}


PINNED = PinnedRows()


class PathArray:
    """The paths of a file list as ONE C array of char* into ONE bytes blob (built once per list: per-file str(Path).encode() + a
    ctypes c_char_p each cost ~1.5 us per file - as much as the native reader needs for the file itself at 16 threads); `window(off, n)`
    is the view of one batch."""

    def __init__(self, paths: Sequence):
        import os
        strs = [p if type(p) is str else os.fsdecode(p) if isinstance(p, bytes) else os.fspath(p) for p in paths]
        self.n = len(strs)
        blob = "\0".join(strs).encode(sys.getfilesystemencoding(), "surrogateescape") + b"\0"
        if len(blob) == sum(map(len, strs)) + self.n:             # one byte per character: offsets straight from the str lengths
            lens = np.fromiter(map(len, strs), dtype=np.int64, count=self.n)
        else:
            lens = np.fromiter((len(os.fsencode(x)) for x in strs), dtype=np.int64, count=self.n)
        self._blob = C.create_string_buffer(blob, len(blob))
        offs = np.zeros(max(1, self.n), dtype=np.uint64)
        if self.n > 1:
            offs[1:self.n] = np.cumsum(lens[:-1] + 1).astype(np.uint64)
        self._ptrs = offs + np.uint64(C.addressof(self._blob))    # uint64 [n]: a char* array in all but name

    def window(self, off: int = 0, n: Optional[int] = None):
        n = self.n - off if n is None else n
        if off < 0 or n < 0 or off + n > self.n:
            raise IndexError("PathArray.window out of range")
        return C.cast(C.c_void_p(self._ptrs.ctypes.data + 8 * off), C.POINTER(C.c_char_p)), n


def file_sizes(paths, n_threads: int = 8) -> np.ndarray:
    """Sizes in bytes (int64, -1 for a missing file) with libamph's host thread pool (fstatat relative to the directory)."""
    pa = paths if isinstance(paths, PathArray) else PathArray(paths)
    out = np.full(pa.n, -1, dtype=np.int64)
    if pa.n:
        ptr, n = pa.window()
        st = N.lib.amph_wav_sizes(ptr, n, out.ctypes.data_as(C.c_void_p), int(n_threads))
        if st != N.AMPH_OK:
            raise RuntimeError(f"amph_wav_sizes failed: {N.STATUS_NAMES.get(st, st)}")
    return out


def read_many_into(pcm: np.ndarray, paths, sr: int, n_threads: int = 8, offset: int = 0):
    """Batch ingest into caller-owned rows `pcm` int16 [n, max_samples] with libamph's host thread pool (the call
    releases the GIL).  `paths`: a sequence of paths, or a PathArray whose entries offset .. offset + n - 1 are read.
    Returns (lengths int32 [n], status int32 [n]); rows of failed files are zero and flagged."""
    n, max_samples = pcm.shape
    lengths = np.zeros(n, dtype=np.int32)
    status = np.zeros(n, dtype=np.int32)
    if n:
        pa = paths if isinstance(paths, PathArray) else PathArray(paths)
        ptr, _ = pa.window(offset, n)
        st = N.lib.amph_wav_read_many(ptr, n, int(sr), pcm.ctypes.data_as(C.c_void_p), int(max_samples),
                                      lengths.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p),
                                      int(n_threads))
        if st != N.AMPH_OK:
            raise RuntimeError(f"amph_wav_read_many failed: {N.STATUS_NAMES.get(st, st)}")
    return lengths, status


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
    lengths, status = read_many_into(pcm, paths, sr, n_threads)
    return pcm, lengths, status, ptr


def free_pinned(ptr) -> None:
    if ptr:
        N.lib.amph_host_free(ptr)
