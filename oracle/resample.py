"""Oracle: the sample-rate conversion inside `librosa.load(path, sr=sr, mono=True)` when the file's rate
differs from `sr` (latent_space_exploration/map_detector_core.py:210, 00_normalize_dataset_rms.py:51).
TEST INFRASTRUCTURE.

PARITY UNPINNED.  librosa 0.9.2 delegates to resampy 0.4.3 (`res_type="kaiser_best"`); neither package nor
its filter data file (`resampy/data/kaiser_best.npz`) is present in this image and the reference holds no
resampled fixture.  What follows restates the PUBLISHED algorithm:

  librosa.core.audio.resample   ratio = target_sr / orig_sr;  y_hat = resampy.resample(y, orig_sr, target_sr,
                                filter="kaiser_best");  fix_length(y_hat, ceil(n * ratio));  no rescaling; float32
  resampy.core.resample         n_out = int(n * ratio);  interp_win scaled by ratio when downsampling;
                                interp_delta = diff(interp_win) (last element 0);  t_out = arange(n_out) / ratio
  resampy.interpn.resample_f    per output sample: left wing then right wing of the windowed-sinc table with
                                linear interpolation between table entries, accumulated INTO the float32 output
                                element (`y[t] += weight * x[...]`: float64 product, float32 running sum)
  resampy.filters.sinc_window   kaiser_best = sinc_window(num_zeros=64, precision=9,
                                window=kaiser(beta=14.769656459379492), rolloff=0.9475937167399596)
                                (the parameters of resampy's documentation; the shipped table cannot be compared)

Anchors available here (tests/test_oracle_golden.py): a band-limited tone resampled 48000 -> 22050 Hz equals the
analytically resampled tone to 1e-4 away from the edges; scipy.signal.resample_poly agrees to 2e-3; identity when
the rates are equal is handled by the caller (librosa does not resample then).
"""
from __future__ import annotations

import numpy as np
from scipy.signal.windows import kaiser

KAISER_BEST = dict(num_zeros=64, precision=9, beta=14.769656459379492, rolloff=0.9475937167399596)


def sinc_window(num_zeros: int, precision: int, beta: float, rolloff: float):
    """resampy.filters.sinc_window with a Kaiser taper -> (half window float64 [num_zeros * 2^precision + 1], 2^precision)."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def filter_tables(sample_ratio: float, params=KAISER_BEST):
    """(interp_win, interp_delta, num_table) as resampy.core.resample prepares them for one ratio."""
    win, num_table = sinc_window(**params)
    if sample_ratio < 1:
        win = sample_ratio * win
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    return win, delta, num_table


def resample(y: np.ndarray, orig_sr: int, target_sr: int, params=KAISER_BEST) -> np.ndarray:
    """float32 mono signal at orig_sr -> float32 at target_sr, length ceil(n * ratio) (librosa.resample, fix=True)."""
    y = np.asarray(y, dtype=np.float32)
    ratio = float(target_sr) / orig_sr
    n_out = int(y.shape[-1] * ratio)
    if n_out < 1:
        raise ValueError("Input signal length is too small to resample")
    win, delta, num_table = filter_tables(ratio, params)
    scale = min(1.0, ratio)
    t_out = np.arange(n_out) * (1.0 / ratio)
    index_step = int(scale * num_table)
    nwin, n_orig = win.shape[0], y.shape[0]
    out = np.zeros(n_out, dtype=np.float32)
    yd = y.astype(np.float64)
    for t in range(n_out):
        tr = t_out[t]
        n = int(tr)
        frac = scale * (tr - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        acc = np.float32(0.0)
        i_max = min(n + 1, (nwin - offset) // index_step)
        idx = offset + np.arange(i_max) * index_step
        for w, x in zip(win[idx] + eta * delta[idx], yd[n - np.arange(i_max)]):
            acc = np.float32(np.float64(acc) + w * x)                 # float64 product, float32 running sum
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        idx = offset + np.arange(k_max) * index_step
        for w, x in zip(win[idx] + eta * delta[idx], yd[n + 1 + np.arange(k_max)]):
            acc = np.float32(np.float64(acc) + w * x)
        out[t] = acc
    n_fix = int(np.ceil(y.shape[-1] * ratio))                          # librosa.util.fix_length(size=ceil(n * ratio))
    if n_fix > n_out:
        out = np.pad(out, (0, n_fix - n_out))
    return out[:n_fix].astype(np.float32)
