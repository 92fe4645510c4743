"""Oracle: WAV chunk -> RMS normalise -> log-mel image, on the CPU in numpy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED for the
librosa part: librosa==0.9.2 (requirements-thesis-baseline-macos-arm64.txt:33)
is absent from /root/reference and from this image, so the functions below
restate its published algorithm; they follow the reference's call sites:

  rms_normalize          latent_space_exploration/00_normalize_dataset_rms.py:29-38
  pcm16 write/read       latent_space_exploration/00_normalize_dataset_rms.py:57
                         (soundfile default WAV subtype PCM_16) and
                         latent_space_exploration/map_detector_core.py:210
  fix length             latent_space_exploration/map_detector_core.py:212-217
  melspectrogram         latent_space_exploration/map_detector_core.py:219-228
  power_to_db(ref=max)   latent_space_exploration/map_detector_core.py:229
  standardise            latent_space_exploration/map_detector_core.py:232
  crop_or_pad_time       latent_space_exploration/map_detector_core.py:185-195

numpy here is 2.x; the reference pins numpy 1.26.4
(requirements-thesis-baseline-macos-arm64.txt:46).  The only place that matters
is scalar promotion in `rms_normalize`: under 1.26 (value-based casting) an
np.float32 SCALAR combined with a Python float is computed in float64, so the
silence gate compares in float64 and the scale `target_rms / (rms + eps)` is a
float64 that the array multiply casts to float32 once; under 2.x (NEP 50) both
stay float32.  The scale then differs by 1 ulp on about a third of all chunks,
and the PCM16 re-quantisation behind it (00 writes PCM16, 07 reads it back)
turns that into whole LSBs.  `rms_normalize` therefore takes `promotion`:
"numpy1" (default: the pinned environment, emulated with explicit float64
scalars - no numpy 1.26 needed) or "numpy2" (what the reference's function
computes when run on the numpy of this image).  Both are pinned by
tests/golden/rms_normalize.npz (see tests/golden/make_golden.py for how the
reference's own function was made to run either way).  power_to_db's reference
value is an array op (np.maximum on a 0-d result of np.max) and is float32 in
both.
"""
from __future__ import annotations

import wave
from pathlib import Path

import numpy as np

# ----------------------------------------------------------------------------
# 00_normalize_dataset_rms.py
# ----------------------------------------------------------------------------


RMS_PROMOTION = "numpy1"   # default scalar arithmetic of rms_normalize: the reference's pinned numpy 1.26.4


def rms_scale(y, target_rms=0.05, rms_min=1e-4, eps=1e-8, promotion=None):
    """The scalar half of 00_normalize_dataset_rms.py:30-36: returns (scale as
    np.float32, ok).  `promotion` "numpy1": np.float32 scalar <op> Python float
    evaluated in float64 (numpy 1.26 value-based casting), the result cast to
    float32 once by the array multiply; "numpy2": float32 throughout (NEP 50)."""
    promotion = promotion or RMS_PROMOTION
    y = np.asarray(y)
    rms = np.sqrt(np.mean(y ** 2))              # np.float32 for float32 input under either numpy
    if promotion == "numpy1" and y.dtype == np.float32:
        rms64 = np.float64(rms)
        if rms64 < np.float64(rms_min):
            return np.float32(1.0), False
        return np.float32(np.float64(target_rms) / (rms64 + np.float64(eps))), True
    if promotion not in ("numpy1", "numpy2"):
        raise ValueError(promotion)
    if rms < rms_min:
        return np.float32(1.0), False
    return np.float32(target_rms / (rms + eps)), True


def rms_normalize(y, target_rms=0.05, rms_min=1e-4, eps=1e-8, promotion=None):
    """00_normalize_dataset_rms.py:29-38 (same names, same order of operations;
    the scalar arithmetic per `promotion`, see rms_scale)."""
    y = np.asarray(y)
    scale, ok = rms_scale(y, target_rms, rms_min, eps, promotion)
    if not ok:
        return y, False
    y_norm = y * scale                           # float32 array x float32 scalar
    y_norm = np.clip(y_norm, -1.0, 1.0)
    return y_norm, True


def float_to_pcm16(x):
    """libsndfile float->PCM_16 with normalisation on (what sf.write does at
    00_normalize_dataset_rms.py:57): lrintf(x * 0x7FFF), round-half-even.
    Input is already clipped to [-1, 1] by rms_normalize."""
    x = np.asarray(x, dtype=np.float32)
    q = np.rint(x * np.float32(32767.0))
    return np.clip(q, -32768, 32767).astype(np.int16)


def pcm16_to_float(q):
    """libsndfile PCM_16->float32 (librosa.load / soundfile.read): q / 0x8000."""
    return np.asarray(q, dtype=np.int16).astype(np.float32) / np.float32(32768.0)


def read_wav_pcm16(path):
    """Mono/stereo PCM16 RIFF reader (stdlib).  Returns (int16 [n, ch], sr)."""
    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2 or w.getcomptype() != "NONE":
            raise ValueError(f"only PCM16 WAV supported by the oracle: {path}")
        sr = w.getframerate()
        ch = w.getnchannels()
        raw = w.readframes(w.getnframes())
    q = np.frombuffer(raw, dtype="<i2").reshape(-1, ch)
    return q, sr


def load_wav(path, sr):
    """librosa.load(path, sr=sr, mono=True) for files already at `sr`
    (map_detector_core.py:210).  Resampling (kaiser_best) is out of scope."""
    q, file_sr = read_wav_pcm16(path)
    if file_sr != sr:
        raise ValueError(f"resampling not restated (file sr {file_sr} != {sr})")
    y = pcm16_to_float(q)
    if y.shape[1] > 1:
        y = np.mean(y, axis=1, dtype=np.float32)
    else:
        y = y[:, 0]
    return np.ascontiguousarray(y, dtype=np.float32)


def fix_length(y, sr, duration):
    """map_detector_core.py:212-217."""
    if duration > 0:
        target_len = int(sr * duration)
        if y.shape[0] < target_len:
            y = np.pad(y, (0, target_len - y.shape[0]), mode="constant")
        else:
            y = y[:target_len]
    return y


# ----------------------------------------------------------------------------
# librosa 0.9.2 restatement (Slaney mel scale, STFT, mel filterbank, dB)
# ----------------------------------------------------------------------------

_F_SP = 200.0 / 3
_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = _MIN_LOG_HZ / _F_SP  # 15.0
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(frequencies):
    """librosa.hz_to_mel(htk=False): linear below 1 kHz, log above."""
    f = np.asanyarray(frequencies, dtype=np.float64)
    mels = f / _F_SP
    if f.ndim:
        log_t = f >= _MIN_LOG_HZ
        mels[log_t] = _MIN_LOG_MEL + np.log(f[log_t] / _MIN_LOG_HZ) / _LOGSTEP
    elif f >= _MIN_LOG_HZ:
        mels = _MIN_LOG_MEL + np.log(f / _MIN_LOG_HZ) / _LOGSTEP
    return mels


def mel_to_hz(mels):
    """librosa.mel_to_hz(htk=False)."""
    m = np.asanyarray(mels, dtype=np.float64)
    freqs = _F_SP * m
    if m.ndim:
        log_t = m >= _MIN_LOG_MEL
        freqs[log_t] = _MIN_LOG_HZ * np.exp(_LOGSTEP * (m[log_t] - _MIN_LOG_MEL))
    elif m >= _MIN_LOG_MEL:
        freqs = _MIN_LOG_HZ * np.exp(_LOGSTEP * (m - _MIN_LOG_MEL))
    return freqs


def mel_frequencies(n_mels, fmin, fmax):
    return mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels))


def fft_frequencies(sr, n_fft):
    return np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(htk=False, norm='slaney', dtype=float32) -> [M, F].
    Rows whose centre frequencies lie above sr/2 come out all-zero, as in
    librosa (which only warns)."""
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=np.float32)
    fftfreqs = fft_frequencies(sr, n_fft)
    mel_f = mel_frequencies(n_mels + 2, fmin, fmax)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def hann_periodic(n_fft):
    """scipy.signal.get_window('hann', n_fft, fftbins=True), float64."""
    n = np.arange(n_fft, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)


def stft(y, n_fft, hop_length, pad_mode="reflect"):
    """librosa.stft(center=True, window='hann', win_length=n_fft) for a float32
    signal: float64 window x float32 frame, double-precision rfft, stored as
    complex64.  [F, T_raw], T_raw = 1 + len(y) // hop_length."""
    y = np.asarray(y, dtype=np.float32)
    win = hann_periodic(n_fft)
    if pad_mode == "reflect":
        yp = np.pad(y, n_fft // 2, mode="reflect")
    elif pad_mode == "constant":
        yp = np.pad(y, n_fft // 2, mode="constant")
    else:
        raise ValueError(pad_mode)
    n_frames = 1 + (yp.shape[0] - n_fft) // hop_length
    idx = np.arange(n_fft)[:, None] + hop_length * np.arange(n_frames)[None, :]
    frames = yp[idx]  # [n_fft, T] float32
    spec = np.fft.rfft(win[:, None] * frames, axis=0)  # float64 -> complex128
    return spec.astype(np.complex64)


def melspectrogram(y, sr, n_fft, hop_length, n_mels, fmin, fmax, pad_mode="reflect"):
    """librosa.feature.melspectrogram(power=2.0) -> float32 [M, T_raw]."""
    S = np.abs(stft(y, n_fft, hop_length, pad_mode)) ** 2.0  # float32
    basis = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    return np.einsum("ft,mf->mt", S, basis, optimize=True).astype(np.float32, copy=False)


def power_to_db(S, amin=1e-10, top_db=80.0):
    """librosa.power_to_db(S, ref=np.max)."""
    S = np.asarray(S)
    ref_value = np.max(S)
    log_spec = 10.0 * np.log10(np.maximum(amin, S))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value))
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def crop_or_pad_time(mel, target_frames):
    """map_detector_core.py:185-195."""
    _, T = mel.shape
    if T == target_frames:
        return mel
    if T > target_frames:
        start = (T - target_frames) // 2
        return mel[:, start:start + target_frames]
    pad_total = target_frames - T
    pad_left = pad_total // 2
    pad_right = pad_total - pad_left
    return np.pad(mel, ((0, 0), (pad_left, pad_right)), mode="constant")


def wave_to_mel(y, *, sr, duration, n_mels, fmin, fmax, hop_length, n_fft,
                target_frames, pad_mode="reflect"):
    """map_detector_core.py:212-237 starting from the decoded float32 signal.
    Returns float32 [n_mels, target_frames]."""
    y = fix_length(np.asarray(y, dtype=np.float32), sr, duration)
    S = melspectrogram(y, sr, n_fft, hop_length, n_mels, fmin, fmax, pad_mode)
    S_db = power_to_db(S)
    S_db = (S_db - S_db.mean()) / (S_db.std() + 1e-8)
    S_db = crop_or_pad_time(S_db, target_frames=target_frames)
    return np.ascontiguousarray(S_db, dtype=np.float32)


def wav_to_mel(wav_path, *, sr, duration, n_mels, fmin, fmax, hop_length, n_fft,
               target_frames, pad_mode="reflect"):
    """map_detector_core.py:198-237 (numpy array instead of a torch tensor)."""
    y = load_wav(Path(wav_path), sr)
    return wave_to_mel(y, sr=sr, duration=duration, n_mels=n_mels, fmin=fmin, fmax=fmax,
                       hop_length=hop_length, n_fft=n_fft, target_frames=target_frames,
                       pad_mode=pad_mode)


def normalize_pcm16_chunk(q, apply_rms=True, promotion=None):
    """The 00 -> disk -> 07 hand-off for one raw PCM16 chunk held in memory:
    decode (q/32768), rms_normalize, write PCM16 (x*32767, rint), decode again.
    With apply_rms=False the chunk is only decoded.  Returns (float32 y, ok)."""
    y = pcm16_to_float(q)
    if not apply_rms:
        return y, True
    y_norm, ok = rms_normalize(y, promotion=promotion)
    return pcm16_to_float(float_to_pcm16(y_norm)), ok
