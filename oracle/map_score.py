"""Oracle: Gaussian-MAP detector fit + scoring + decision, numpy.  TEST INFRASTRUCTURE.

PINNED against the reference's own functions (tests/golden/map_fit.npz, map_decide*.npz,
captured by tests/golden/make_golden_map.py):

  estimate_cov                     latent_space_exploration/08b_fit_map_detector.py:60-81
  inv_and_logdet                   latent_space_exploration/map_detector_core.py:306-317
  gaussian_logpdf_from_precision   latent_space_exploration/map_detector_core.py:319-323
                                   (= gaussian_logpdf, 08b_fit_map_detector.py:84-88)
  fit block of 08b's main          latent_space_exploration/08b_fit_map_detector.py:262-312
  get_priors_from_map_meta         latent_space_exploration/map_detector_core.py:326-356
  decision block                   latent_space_exploration/10b_benchmark_folder_detection_map.py:143-169

Numerics: `diff.T @ prec @ diff` is float32 BLAS (sgemv + sdot; the summation order is the
BLAS library's own and is not specified), everything after `float(quad)` is float64.
"""
from __future__ import annotations

import numpy as np


def estimate_cov(Z, eps, shrink, cov_structure):
    """08b_fit_map_detector.py:60-81."""
    n, d = Z.shape
    if n < 2:
        cov = np.eye(d, dtype=np.float32)
    else:
        cov = np.cov(Z, rowvar=False, bias=False).astype(np.float32)
    if cov_structure == "diag":
        cov = np.diag(np.diag(cov)).astype(np.float32)
    if shrink > 0:
        avg_var = float(np.mean(np.diag(cov))) if d > 0 else 1.0
        cov = (1.0 - shrink) * cov + shrink * (avg_var * np.eye(d, dtype=np.float32))
    cov = cov + (eps * np.eye(d, dtype=np.float32))
    return cov.astype(np.float32)


def inv_and_logdet(cov):
    """map_detector_core.py:306-317."""
    sign, ld = np.linalg.slogdet(cov)
    if sign <= 0:
        d = cov.shape[0]
        cov2 = cov + (1e-3 * np.eye(d, dtype=cov.dtype))
        sign, ld = np.linalg.slogdet(cov2)
        if sign <= 0:
            raise RuntimeError("Covarianza no PD incluso tras regularización.")
        cov = cov2
    prec = np.linalg.inv(cov).astype(np.float32)
    return prec, float(ld)


def gaussian_logpdf_from_precision(z, mu, prec, logdet_cov):
    """map_detector_core.py:319-323."""
    d = int(z.shape[0])
    diff = (z - mu).astype(np.float32)
    quad = float(diff.T @ prec @ diff)
    return -0.5 * (quad + float(logdet_cov) + d * float(np.log(2.0 * np.pi)))


def fit_map(Z_by_species, cov_type="lda", cov_structure="full", priors="empirical", eps=1e-6, shrink=0.0, tau_q=None):
    """08b_fit_map_detector.py:259-312 -> dict(species, priors, means, covs, precs, logdets, scores_true, tau)."""
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
        cov = estimate_cov(Zc, eps, shrink, cov_structure)
        prec, ld = inv_and_logdet(cov)
        for sp in species:
            covs[sp], precs[sp], logdets[sp] = cov, prec, ld
    else:
        for sp in species:
            Zc = Z_by_species[sp] - means[sp][None, :]
            covs[sp] = estimate_cov(Zc, eps, shrink, cov_structure)
            precs[sp], logdets[sp] = inv_and_logdet(covs[sp])
    scores = {}
    allscores = []
    for sp in species:
        lp = float(np.log(pri[sp] + 1e-12))
        s = np.array([gaussian_logpdf_from_precision(z, means[sp], precs[sp], logdets[sp]) + lp for z in Z_by_species[sp]],
                     dtype=np.float64)
        scores[sp] = s
        allscores.extend(list(s))
    tau = float(np.quantile(np.array(allscores, dtype=np.float64), float(tau_q))) if tau_q is not None else None
    return dict(species=species, priors=pri, means=means, covs=covs, precs=precs, logdets=logdets, scores_true=scores, tau=tau)


def get_priors(per_species_prior, species):
    """map_detector_core.py:326-356: normalised meta priors when every species has one, else uniform."""
    pri, ok = {}, True
    for sp in species:
        try:
            pri[sp] = float(per_species_prior.get(sp))
        except Exception:  # noqa: BLE001
            ok = False
            break
    if ok and pri:
        s = sum(max(0.0, v) for v in pri.values())
        if s > 0:
            pri = {k: max(0.0, v) / s for k, v in pri.items()}
        return pri
    K = len(species)
    return {sp: 1.0 / K for sp in species} if K else {}


def scores(z, species, means, precs, logdets, priors):
    out = []
    for sp in species:
        lp = float(np.log(float(priors.get(sp, 1e-12)) + 1e-12))
        out.append(gaussian_logpdf_from_precision(z, means[sp], precs[sp], logdets[sp]) + lp)
    return np.array(out, dtype=np.float64)


def decide(z, species, means, precs, logdets, priors, tau):
    """10b_benchmark_folder_detection_map.py:143-169 -> (detected, species|None, best_score)."""
    best_sp, best = None, -float("inf")
    for sp in species:
        if means[sp].shape[0] != z.shape[0]:
            continue
        lp = float(np.log(float(priors.get(sp, 1e-12)) + 1e-12))
        s = gaussian_logpdf_from_precision(z, means[sp], precs[sp], logdets[sp]) + lp
        if s > best:
            best, best_sp = s, sp
    if best_sp is None:
        return False, None, best
    if tau is not None and best < float(tau):
        return False, None, best
    return True, best_sp, best
