"""Radial detector on top of the encoded latents: host fit + on-GPU decision.

  fit      latent_space_exploration/08_fit_radial_detector.py:105-112, 310-333, 525-587
           (O(N*D), runs once per q-grid point: stays numpy on the host)
  decide   latent_space_exploration/10_benchmark_folder_detection.py:152-199 with
           l2 / get_detector_from_config / PRIORITY_ORDER from
           09_evaluate_wav_detection.py:354-355, 113-149, 61-66  -> HIP kernel K6
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

PRIORITY_ORDER: List[str] = [
    "Batrachyla_leptopus",
    "Batrachyla_taeniata",
    "Calyptocephalella_gayi",
    "Pleurodema_thaul",
]


# ---------------------------------------------------------------------------- fit (host)
def l2_norm_rows(x: np.ndarray) -> np.ndarray:
    return np.sqrt(np.sum(x * x, axis=1))


QUANTILE_NUMPY2 = False   # scalar arithmetic of np.quantile: the reference's pinned numpy 1.26.4 (float64) unless set


def lerp_quantile(a, b, n: int, q: float, numpy2: Optional[bool] = None) -> float:
    """The interpolation step of float(np.quantile(x, q)) (method "linear") given the two order statistics a = x_(lo),
    b = x_(lo + 1) of a float32 sample of size n (see order_indices).  numpy 1.26 (the reference's pin, the default here):
    q, the virtual index (n - 1) q and the weight t in float64, a + (b - a) t with the difference in float32 and the rest
    in float64; numpy 2.x casts a Python-float q to float32 first and does everything in float32 (a 1e-8 difference in
    the thresholds; both are pinned by tests/golden/radial_fit.npz)."""
    numpy2 = QUANTILE_NUMPY2 if numpy2 is None else numpy2
    f = np.float32 if numpy2 else np.float64
    vi = f(n - 1) * f(q)
    t = f(vi - f(np.floor(vi)))
    a, b = np.float32(a), np.float32(b)
    diff = b - a
    return float(f(b) - f(diff) * (f(1) - t)) if t >= 0.5 else float(f(a) + f(diff) * t)


def order_indices(n: int, q: float, numpy2: Optional[bool] = None) -> Tuple[int, int]:
    """(lo, hi): the order statistics np.quantile interpolates between."""
    numpy2 = QUANTILE_NUMPY2 if numpy2 is None else numpy2
    f = np.float32 if numpy2 else np.float64
    vi = f(n - 1) * f(q)
    if vi >= n - 1:
        return n - 1, n - 1
    if vi < 0:
        return 0, 0
    lo = int(np.floor(vi))
    return lo, lo + 1


def quantile_safe(x: np.ndarray, q: float, numpy2: Optional[bool] = None) -> float:
    """08_fit_radial_detector.py:109-112: float(np.quantile(x, q)), 0.0 for an empty sample."""
    if not x.size:
        return 0.0
    if x.dtype != np.float32:
        return float(np.quantile(x, q))
    lo, hi = order_indices(x.shape[0], q, numpy2)
    part = np.partition(x, sorted({lo, hi}))
    return lerp_quantile(part[lo], part[hi], x.shape[0], q, numpy2)


def summarize_dist(x: np.ndarray, numpy2: Optional[bool] = None) -> Dict[str, float]:
    """08_fit_radial_detector.py:115-123 (its quantiles in the same scalar arithmetic as the thresholds')."""
    if x.size == 0:
        return dict.fromkeys(("min", "p50", "p90", "max"), float("nan"))
    return {"min": float(np.min(x)), "p50": quantile_safe(x, 0.50, numpy2), "p90": quantile_safe(x, 0.90, numpy2),
            "max": float(np.max(x))}


def fit_species_with_fp_control(Z_in: np.ndarray, Z_out: Optional[np.ndarray], q_in: float, q_out: float,
                                numpy2: Optional[bool] = None):
    """centroid = mean(Z_in); rk = min(quantile_q_in(in-class radii), quantile_q_out(out-class radii))."""
    mu = np.mean(Z_in, axis=0).astype(np.float32)
    rho_in = l2_norm_rows(Z_in - mu[None, :])
    rk_in = quantile_safe(rho_in, q_in, numpy2)
    if Z_out is None or Z_out.size == 0:
        rho_out, rk_out = np.array([], dtype=np.float32), float("inf")
    else:
        rho_out = l2_norm_rows(Z_out - mu[None, :])
        rk_out = quantile_safe(rho_out, q_out, numpy2)
    rk = float(min(rk_in, rk_out))
    return mu, rk, rk_in, rk_out, {"rho_in_summary": summarize_dist(rho_in, numpy2), "rho_out_summary": summarize_dist(rho_out, numpy2)}


def fit_radial_detector(Z_by_species: Dict[str, Any], q_in: float, q_out: float,
                        failed: Optional[Dict[str, int]] = None, verbose: bool = True, session=None,
                        numpy2: Optional[bool] = None) -> Dict[str, Any]:
    """-> the `radial_detector` block the reference writes into config.json
    ({"centroids", "thresholds", "per_species"}); prints the `✅ sp: rk_in=… | rk_out=… | rk=…`
    line that 9105_make_config_snapshot_from_log.py:11-13 parses.
    With `session` (an AmphSession) the array work - centroids, all distances, the order statistics behind every quantile -
    runs on the GPU on the latents where they are (K9, amph_fit_radial; values may be device tensors): same centroids,
    same thresholds, to the last bit, as the host form."""
    numpy2 = QUANTILE_NUMPY2 if numpy2 is None else numpy2
    centroids, thresholds, meta = {}, {}, {}
    dev = None
    if session is not None:
        names = list(Z_by_species)
        parts = [Z_by_species[sp] if torch.is_tensor(Z_by_species[sp]) else torch.from_numpy(np.ascontiguousarray(Z_by_species[sp], dtype=np.float32))
                 for sp in names]
        offsets = np.concatenate([[0], np.cumsum([int(t.shape[0]) for t in parts])]).astype(np.int64)
        zcat = torch.cat([t.to(session.device, dtype=torch.float32) for t in parts], dim=0)
        cent, rk_in_a, rk_out_a, rk_a, summ = session.fit_radial(zcat, offsets, q_in, q_out, numpy2=numpy2)
        keys = ("min", "p50", "p90", "max")
        dev = {sp: (cent[i], float(rk_a[i]), float(rk_in_a[i]), float(rk_out_a[i]),
                    {"rho_in_summary": dict(zip(keys, (float(v) for v in summ[i, 0]))),
                     "rho_out_summary": dict(zip(keys, (float(v) for v in summ[i, 1])))}) for i, sp in enumerate(names)}
    for sp, Z_in in Z_by_species.items():
        others = [Z for o, Z in Z_by_species.items() if o != sp]
        if dev is not None:
            mu, rk, rk_in, rk_out, extra = dev[sp]
            Z_out = None if not others else _Rows(sum(int(z.shape[0]) for z in others))
        else:
            Z_out = np.concatenate(others, axis=0) if others else None
            mu, rk, rk_in, rk_out, extra = fit_species_with_fp_control(Z_in, Z_out, q_in, q_out, numpy2)
        centroids[sp] = mu.tolist()
        thresholds[sp] = float(rk)
        meta[sp] = {"N_in": int(Z_in.shape[0]), "N_out": int(Z_out.shape[0]) if Z_out is not None else 0,
                    "rk_in": float(rk_in), "rk_out": float(rk_out) if np.isfinite(rk_out) else None,
                    "rk_final": float(rk), "failed": int((failed or {}).get(sp, 0)), "used": int(Z_in.shape[0]),
                    **extra}
        if verbose:
            rk_out_print = rk_out if np.isfinite(rk_out) else float("nan")
            print(f"✅ {sp}: rk_in={rk_in:.6f} | rk_out={rk_out_print:.6f} | rk={rk:.6f}")
    return {"centroids": centroids, "thresholds": thresholds, "per_species": meta}


class _Rows:
    """Stand-in for the concatenated out-class matrix on the device path: only its row count is reported."""

    def __init__(self, n: int):
        self.shape = (n,)


# ---------------------------------------------------------------------------- decide (GPU)
def get_detector_from_config(cfg: Dict[str, Any]) -> Tuple[Dict[str, np.ndarray], Dict[str, float], float]:
    rd = cfg.get("radial_detector")
    if not isinstance(rd, dict):
        raise ValueError("config.json no contiene radial_detector (dict). Ejecuta antes 08_fit_radial_detector.py")
    cent, thr = rd.get("centroids"), rd.get("thresholds")
    if not isinstance(cent, dict) or not isinstance(thr, dict):
        raise ValueError("radial_detector debe contener 'centroids' y 'thresholds' como dicts.")
    centroids = {sp: np.array(v, dtype=np.float32) for sp, v in cent.items()
                 if isinstance(sp, str) and isinstance(v, list) and len(v) > 0}
    thresholds = {sp: float(v) for sp, v in thr.items() if isinstance(sp, str)}
    if not centroids or not thresholds:
        raise ValueError("centroids/thresholds vacíos o mal formateados en config.json.")
    try:
        chunk_seconds = float(cfg.get("chunk_seconds", 5.0))
    except Exception:  # noqa: BLE001
        chunk_seconds = 5.0
    return centroids, thresholds, chunk_seconds


def priority_ranks(names: Sequence[str], priority: Sequence[str] = PRIORITY_ORDER) -> np.ndarray:
    """rank such that `min rank among accepted` == "first of PRIORITY_ORDER in accepted,
    else sorted(accepted)[0]" (10_benchmark_folder_detection.py:191-199)."""
    pr = {sp: i for i, sp in enumerate(priority)}
    rest = sorted(sp for sp in names if sp not in pr)
    rr = {sp: len(priority) + i for i, sp in enumerate(rest)}
    return np.array([pr.get(sp, rr.get(sp)) for sp in names], dtype=np.int32)


class RadialDetector:
    """Centroids + thresholds held for batched on-GPU decisions."""

    def __init__(self, centroids: Dict[str, np.ndarray], thresholds: Dict[str, float], latent_dim: int,
                 priority: Sequence[str] = PRIORITY_ORDER):
        # the reference's loop skips species without a threshold or with a mismatching
        # dimension (10_benchmark_folder_detection.py:178-183); keep config order
        self.names = [sp for sp, mu in centroids.items() if sp in thresholds and mu.shape[0] == latent_dim]
        self.dim = latent_dim
        self.centroids = (np.stack([np.asarray(centroids[sp], np.float32) for sp in self.names])
                          if self.names else np.zeros((0, latent_dim), np.float32))
        self.thresholds = np.array([float(thresholds[sp]) for sp in self.names], dtype=np.float64)
        self.rank = priority_ranks(self.names, priority) if priority else priority_ranks(self.names, [])

    @classmethod
    def from_config(cls, cfg: Dict[str, Any], latent_dim: int):
        c, t, _ = get_detector_from_config(cfg)
        return cls(c, t, latent_dim)

    def decide(self, session, Z) -> Tuple[np.ndarray, np.ndarray]:
        """Z [B, D] (device tensor or array) -> (pred index int32 [B] with -1 = NO_DETECT, best_d float32 [B])."""
        pred, best = session.decide_radial(Z, self.centroids, self.thresholds, self.rank)
        return pred.cpu().numpy(), best.cpu().numpy()

    def species(self, pred: np.ndarray) -> List[Optional[str]]:
        return [self.names[i] if i >= 0 else None for i in pred.tolist()]


@torch.no_grad()
def detect_chunks(encoder, detector: RadialDetector, chunks, cfg, *, lengths=None, apply_rms: bool = False):
    """Streaming form of the folder benchmark's inner loop (10_benchmark_folder_detection.py:387-418):
    encode + decide on the GPU without the latents leaving HBM.
    -> (detected bool [B], species list, best_d float32 [B], latents device tensor)."""
    sess = encoder.session(cfg)
    mu, _, _ = sess.encode(chunks, lengths=lengths, apply_rms=apply_rms)
    pred, best = detector.decide(sess, mu)
    return pred >= 0, detector.species(pred), best, mu
