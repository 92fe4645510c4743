"""The interpolation filter of librosa.load's default resampler (librosa 0.9.2 `res_type="kaiser_best"` = resampy's
pre-computed windowed sinc), rebuilt from resampy's published construction

    sinc_window(num_zeros=64, precision=9, window=kaiser(beta=14.769656459379492), rolloff=0.9475937167399596)

because the package and its data file are not vendored by the reference
(requirements-thesis-baseline-macos-arm64.txt:71 pins resampy==0.4.3).  The HIP kernel that applies it is K8
(`AmphSession.resample`, include/amph.h amph_resample_f32).  Parity unpinned: the shipped table could not be compared.
"""
from __future__ import annotations

from functools import lru_cache
from typing import Tuple

import numpy as np

KAISER_BEST = dict(num_zeros=64, precision=9, beta=14.769656459379492, rolloff=0.9475937167399596)


def sinc_window(num_zeros: int, precision: int, beta: float, rolloff: float) -> Tuple[np.ndarray, int]:
    """Right half of a Kaiser-windowed sinc sampled 2^precision times per zero crossing -> (float64 half window, 2^precision)."""
    if not 0 < rolloff <= 1 or num_zeros < 1 or precision < 0:
        raise ValueError("invalid filter parameters")
    try:
        from scipy.signal.windows import kaiser
    except ImportError:                                  # numpy's kaiser differs from scipy's in the last bits only
        kaiser = np.kaiser
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = kaiser(2 * n + 1, beta)[n:]
    return np.ascontiguousarray(taper * sinc_win, dtype=np.float64), num_bits


@lru_cache(maxsize=4)
def kaiser_best_window() -> Tuple[np.ndarray, int]:
    return sinc_window(**KAISER_BEST)
