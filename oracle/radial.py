"""Oracle: radial detector fit + decision, numpy.  TEST INFRASTRUCTURE.

PINNED against the reference's own functions (tests/golden/radial_*.npz):

  l2                      latent_space_exploration/09_evaluate_wav_detection.py:354-355
  PRIORITY_ORDER          latent_space_exploration/09_evaluate_wav_detection.py:61-66
  decision block          latent_space_exploration/10_benchmark_folder_detection.py:175-199
  l2_norm_rows            latent_space_exploration/08_fit_radial_detector.py:105-106
  quantile_safe           latent_space_exploration/08_fit_radial_detector.py:109-112
  fit_species_with_fp_control   latent_space_exploration/08_fit_radial_detector.py:310-333
"""
from __future__ import annotations

import numpy as np

PRIORITY_ORDER = [
    "Batrachyla_leptopus",
    "Batrachyla_taeniata",
    "Calyptocephalella_gayi",
    "Pleurodema_thaul",
]


def l2(a):
    return float(np.sqrt(np.sum(a * a)))


def decide(z, centroids: dict, thresholds: dict, priority=PRIORITY_ORDER):
    """10_benchmark_folder_detection.py:175-199 -> (detected, species|None, best_d)."""
    accepted = []
    best_d = float("inf")
    for sp, mu in centroids.items():
        if sp not in thresholds:
            continue
        rk = float(thresholds[sp])
        if mu.shape[0] != z.shape[0]:
            continue
        d = float(l2(z - mu))
        best_d = min(best_d, d)
        if d <= rk:
            accepted.append(sp)
    if not accepted:
        return False, None, best_d
    if priority:
        for sp in priority:
            if sp in accepted:
                return True, sp, best_d
    return True, sorted(accepted)[0], best_d


def l2_norm_rows(x):
    return np.sqrt(np.sum(x * x, axis=1))


QUANTILE_PROMOTION = "numpy1"    # the reference's pinned numpy 1.26.4 (requirements-thesis-baseline-macos-arm64.txt:46)


def quantile_linear(x, q, promotion=None):
    """float(np.quantile(x, q)) for a 1-D float32 array, method "linear", restated with the scalar arithmetic spelled out:
      "numpy1"  numpy 1.26: q, the virtual index (n - 1) q and the weight t stay float64; a + (b - a) t with the difference
                taken in float32 and the rest in float64 (t >= 0.5: b - (b - a)(1 - t));
      "numpy2"  numpy 2.x: a Python-float q is cast to float32 first, everything in float32.
    Pinned both ways by tests/golden/radial_fit.npz (rk1_* / rk_*), "numpy2" also against the installed numpy."""
    promotion = promotion or QUANTILE_PROMOTION
    x = np.asarray(x)
    n = x.shape[0]
    f = np.float64 if (promotion == "numpy1" or x.dtype != np.float32) else np.float32
    if promotion not in ("numpy1", "numpy2"):
        raise ValueError(promotion)
    vi = f(n - 1) * f(q)
    if vi >= n - 1:
        lo = hi = n - 1
    elif vi < 0:
        lo = hi = 0
    else:
        lo = int(np.floor(vi))
        hi = lo + 1
    s = np.sort(x)
    a, b = s[lo], s[hi]
    t = f(vi - f(np.floor(vi)))
    diff = b - a                                       # array dtype (float32)
    if t >= 0.5:
        r = f(b) - f(diff) * (f(1) - t)
    else:
        r = f(a) + f(diff) * t
    return float(r)


def quantile_safe(x, q, promotion=None):
    if x.size == 0:
        return 0.0
    return quantile_linear(x, q, promotion)


def fit_species_with_fp_control(Z_in, Z_out, q_in, q_out, promotion=None):
    """08_fit_radial_detector.py:310-333 -> (mu, rk, rk_in, rk_out)."""
    mu = np.mean(Z_in, axis=0).astype(np.float32)
    rho_in = l2_norm_rows(Z_in - mu[None, :])
    rk_in = quantile_safe(rho_in, q_in, promotion)
    if Z_out is None or Z_out.size == 0:
        rk_out = float("inf")
    else:
        rho_out = l2_norm_rows(Z_out - mu[None, :])
        rk_out = quantile_safe(rho_out, q_out, promotion)
    rk = float(min(rk_in, rk_out))
    return mu, rk, rk_in, rk_out


def pairwise_sum_f32(a):
    """Literal emulation of numpy's float32 pairwise summation,
    used to pin the summation order the HIP decision kernel reproduces."""
    a = np.asarray(a, dtype=np.float32)
    n = a.shape[0]
    f = np.float32
    if n < 8:
        # numpy starts from -0.0 so that a sum of negative zeros keeps its sign
        # (irrelevant for sums of squares).
        r = f(-0.0)
        for i in range(n):
            r = f(r + a[i])
        return r
    if n <= 128:
        r = [f(a[j]) for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = f(r[j] + a[i + j])
            i += 8
        res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
        while i < n:
            res = f(res + a[i])
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return f(pairwise_sum_f32(a[:n2]) + pairwise_sum_f32(a[n2:]))
