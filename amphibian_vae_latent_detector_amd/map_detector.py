"""Gaussian-MAP detector on top of the encoded latents: host fit + on-GPU scoring / decision.

  fit      latent_space_exploration/08b_fit_map_detector.py:60-81 (estimate_cov), :259-351 (priors, means,
           lda / qda covariances, true-class scores, tau, the `map_detector` config block).
           O(N*D^2) once per fit: stays numpy on the host, except the N*K score evaluations, which run
           through the same HIP kernel the decision uses (K7) when a session is given.
  decide   latent_space_exploration/10b_benchmark_folder_detection_map.py:108-169 (MapDetectorSession)
           -> HIP kernel K7 (amph_decide_map): all K scores + argmax + tau reject per chunk on the GPU.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .map_detector_core import (gaussian_logpdf_from_precision, get_chunk_seconds_for_map, get_priors_from_map_meta,
                                inv_and_logdet, read_map_detector_params, summarize_1d)


# ---------------------------------------------------------------------------- fit (host)
def estimate_cov(Z: np.ndarray, eps: float, shrink: float, cov_structure: str) -> np.ndarray:
    """Sample covariance (ddof 1), optional diagonal structure, shrinkage towards avg_var * I, + eps * I."""
    n, d = Z.shape
    cov = np.eye(d, dtype=np.float32) if n < 2 else np.cov(Z, rowvar=False, bias=False).astype(np.float32)
    if cov_structure == "diag":
        cov = np.diag(np.diag(cov)).astype(np.float32)
    if shrink > 0:
        avg_var = float(np.mean(np.diag(cov))) if d > 0 else 1.0
        cov = (1.0 - shrink) * cov + shrink * (avg_var * np.eye(d, dtype=np.float32))
    cov = cov + (eps * np.eye(d, dtype=np.float32))
    return cov.astype(np.float32)


def fit_map_detector(Z_by_species: Dict[str, np.ndarray], *, cov_type: str = "lda", cov_structure: str = "full",
                     priors: str = "empirical", eps: float = 1e-6, shrink: float = 0.0, set_tau_q: Optional[float] = None,
                     failed: Optional[Dict[str, int]] = None, session=None, meta: Optional[Dict[str, Any]] = None,
                     verbose: bool = True) -> Dict[str, Any]:
    """-> the `map_detector` block 08b writes into config.json (:322-351).

    `session` (an AmphSession): evaluate the true-class scores of the tau quantile with the HIP kernel instead of
    the reference's per-vector numpy loop (:302-306); without one the loop runs on the host."""
    if not (0.0 <= shrink <= 1.0):
        raise ValueError("shrink debe estar en [0,1].")
    if not Z_by_species:
        raise ValueError("No se codificó ninguna especie.")
    species = sorted(Z_by_species.keys())
    K = len(species)
    if priors == "uniform":
        pri = {sp: 1.0 / K for sp in species}
    else:
        total = float(sum(Z_by_species[sp].shape[0] for sp in species))
        pri = {sp: float(Z_by_species[sp].shape[0]) / total for sp in species}
    means = {sp: np.mean(Z_by_species[sp], axis=0).astype(np.float32) for sp in species}
    covs, precs, logdets = {}, {}, {}
    if cov_type == "lda":
        Zc = np.concatenate([Z_by_species[sp] - means[sp][None, :] for sp in species], axis=0)
        cov = estimate_cov(Zc, float(eps), float(shrink), cov_structure)
        prec, ld = inv_and_logdet(cov)
        for sp in species:
            covs[sp], precs[sp], logdets[sp] = cov, prec, ld
    elif cov_type == "qda":
        for sp in species:
            covs[sp] = estimate_cov(Z_by_species[sp] - means[sp][None, :], float(eps), float(shrink), cov_structure)
            precs[sp], logdets[sp] = inv_and_logdet(covs[sp])
    else:
        raise ValueError(f"cov_type inesperado: {cov_type}")

    det = MapDetector(means, precs, logdets, pri, None, int(next(iter(means.values())).shape[0]))
    scores_true: List[float] = []
    per_species: Dict[str, Any] = {}
    for sp in species:
        Z = Z_by_species[sp]
        if session is not None and Z.shape[0]:
            s = det.scores(session, Z)[:, det.names.index(sp)]
        else:
            lp = float(np.log(pri[sp] + 1e-12))
            s = np.array([gaussian_logpdf_from_precision(z, means[sp], precs[sp], logdets[sp]) + lp for z in Z], dtype=np.float64)
        scores_true.extend(list(s))
        per_species[sp] = {"N": int(Z.shape[0]), "failed": int((failed or {}).get(sp, 0)), "used": int(Z.shape[0]),
                           "prior": float(pri[sp]), "score_true_summary": summarize_1d(s.astype(np.float32))}
    arr = np.array(scores_true, dtype=np.float64)
    tau = None
    if set_tau_q is not None:
        tau = float(np.quantile(arr, float(set_tau_q)))
        if verbose:
            print(f"\n✅ tau fijado desde train: tau = quantile(score_true_class, q={float(set_tau_q)}) = {tau:.6f}")
    meta_fit = dict(meta or {})
    meta_fit.update({"eps": float(eps), "shrink": float(shrink),
                     "tau_from_train_quantile": float(set_tau_q) if set_tau_q is not None else None,
                     "score_true_global_summary": summarize_1d(arr.astype(np.float32)), "per_species": per_species})
    return {"model": "gaussian_map", "cov_type": str(cov_type), "cov_structure": str(cov_structure), "priors": str(priors),
            "means": {sp: means[sp].astype(float).tolist() for sp in species},
            "cov": {sp: covs[sp].astype(float).tolist() for sp in species},
            "precision": {sp: precs[sp].astype(float).tolist() for sp in species},
            "logdet_cov": {sp: float(logdets[sp]) for sp in species}, "tau": tau, "meta_fit": meta_fit}


# ---------------------------------------------------------------------------- decide (GPU)
class MapDetector:
    """Means / precisions / log-dets / priors held for batched on-GPU MAP decisions (K7)."""

    def __init__(self, means: Dict[str, np.ndarray], precisions: Dict[str, np.ndarray], logdets: Dict[str, float],
                 priors: Dict[str, float], tau: Optional[float], latent_dim: int):
        # sorted intersection (10b:113); species whose mean has another dimension are skipped (10b:153-154)
        names = sorted(set(means.keys()) & set(precisions.keys()) & set(logdets.keys()))
        self.names = [sp for sp in names if np.asarray(means[sp]).shape[0] == latent_dim]
        self.dim = latent_dim
        K = len(self.names)
        self.means = (np.stack([np.asarray(means[sp], np.float32) for sp in self.names]) if K else np.zeros((0, latent_dim), np.float32))
        self.precisions = (np.stack([np.asarray(precisions[sp], np.float32) for sp in self.names]) if K
                           else np.zeros((0, latent_dim, latent_dim), np.float32))
        self.logdets = np.array([float(logdets[sp]) for sp in self.names], dtype=np.float64)
        self.log_priors = np.array([float(np.log(float(priors.get(sp, 1e-12)) + 1e-12)) for sp in self.names], dtype=np.float64)
        self.tau = None if tau is None else float(tau)

    @classmethod
    def from_config(cls, cfg: Dict[str, Any], latent_dim: int):
        means, precs, logdets, tau = read_map_detector_params(cfg)
        species = sorted(set(means.keys()) & set(precs.keys()) & set(logdets.keys()))
        if not species:
            raise RuntimeError("map_detector inconsistente: no hay intersección entre means/precision/logdet_cov.")
        return cls(means, precs, logdets, get_priors_from_map_meta(cfg, species), tau, latent_dim)

    def decide(self, session, Z) -> Tuple[np.ndarray, np.ndarray]:
        """Z [B, D] -> (pred index int32 [B] with -1 = NO_DETECT, best_score float64 [B])."""
        pred, best, _ = session.decide_map(Z, self.means, self.precisions, self.logdets, self.log_priors, self.tau)
        return pred.cpu().numpy(), best.cpu().numpy()

    def scores(self, session, Z) -> np.ndarray:
        """all scores float64 [B, K] (log-likelihood + log prior per species)."""
        _, _, sc = session.decide_map(Z, self.means, self.precisions, self.logdets, self.log_priors, self.tau, want_scores=True)
        return sc.cpu().numpy()

    def species(self, pred: np.ndarray) -> List[Optional[str]]:
        return [self.names[i] if i >= 0 else None for i in np.asarray(pred).tolist()]


@torch.no_grad()
def detect_chunks_map(encoder, detector: MapDetector, chunks, cfg, *, lengths=None, apply_rms: bool = False):
    """Streaming form of 10b's inner loop (10b_benchmark_folder_detection_map.py:351-375): encode + MAP decision
    on the GPU without the latents leaving HBM.  -> (detected bool [B], species list, best_score f64 [B], latents)."""
    sess = encoder.session(cfg)
    mu, _, _ = sess.encode(chunks, lengths=lengths, apply_rms=apply_rms)
    pred, best = detector.decide(sess, mu)
    return pred >= 0, detector.species(pred), best, mu


__all__ = ["estimate_cov", "fit_map_detector", "MapDetector", "detect_chunks_map", "get_chunk_seconds_for_map"]
