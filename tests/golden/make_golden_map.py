#!/usr/bin/env python3
"""Golden vectors for the Gaussian-MAP detector, captured from the REFERENCE's own functions
(build container only; same scaffolding and rules as make_golden.py: librosa / soundfile are
registered as empty placeholder modules so that the module bodies execute, the captured
functions are pure numpy and never touch them; nothing here travels to the GPU box except
the .npz files this script writes).

Captured (reference file:line -> fixture):
  estimate_cov                    08b_fit_map_detector.py:60-81          map_fit.npz
  inv_and_logdet                  map_detector_core.py:306-317           map_fit.npz
  gaussian_logpdf (fit copy)      08b_fit_map_detector.py:84-88          map_fit.npz
  the fit block of 08b's main (priors, means, lda / qda covariances, true-class scores, tau)
                                  08b_fit_map_detector.py:262-312        map_fit.npz
    (main() itself needs an encoder and files; the block is re-driven here line by line with
     the reference's own estimate_cov / inv_and_logdet / gaussian_logpdf doing the arithmetic)
  gaussian_logpdf_from_precision  map_detector_core.py:319-323           map_decide.npz
  read_map_detector_params / get_priors_from_map_meta / get_chunk_seconds_for_map
                                  map_detector_core.py:326-420           map_decide.npz
  MapDetectorSession.predict_one  10b_benchmark_folder_detection_map.py:129-169   map_decide.npz
    (the module's encode_wav_to_latent replaced by a lookup of prepared latents)
"""
import importlib.util
import json
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF_ROOT = Path("/root/reference")
REF = REF_ROOT / "latent_space_exploration"


def _load(fname, modname):
    for name in ("librosa", "soundfile"):
        sys.modules.setdefault(name, types.ModuleType(name))  # inert, attribute-free
    if str(REF_ROOT) not in sys.path:
        sys.path.insert(0, str(REF_ROOT))                      # `from latent_space_exploration.map_detector_core import …`
    spec = importlib.util.spec_from_file_location(modname, REF / fname)
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def main():
    import matplotlib
    matplotlib.use("Agg")
    core = _load("map_detector_core.py", "refcore_map")
    m08b = _load("08b_fit_map_detector.py", "ref08b")
    m10b = _load("10b_benchmark_folder_detection_map.py", "ref10b")

    rng = np.random.default_rng(11)
    species = ["Batrachyla_leptopus", "Batrachyla_taeniata", "Calyptocephalella_gayi", "Pleurodema_thaul"]
    D = 32
    centers = rng.standard_normal((4, D)).astype(np.float32) * 1.2
    mix = rng.standard_normal((D, D)).astype(np.float32) * 0.3 + np.eye(D, dtype=np.float32)
    Z = {sp: (centers[i] + rng.standard_normal((70 + 13 * i, D)).astype(np.float32) @ mix).astype(np.float32)
         for i, sp in enumerate(species)}

    # ------------------------------------------------------------------ fit
    out = {"species_json": np.array(json.dumps(species))}
    for i, sp in enumerate(species):
        out[f"Z_{i}"] = Z[sp]
    variants = [("lda", "full", 0.0, "empirical", 0.01), ("qda", "full", 0.0, "uniform", None),
                ("lda", "diag", 0.1, "empirical", 0.05), ("qda", "diag", 0.25, "empirical", 0.02)]
    for vi, (cov_type, cov_structure, shrink, priors_mode, tau_q) in enumerate(variants):
        eps = 1e-6
        species_present = sorted(Z.keys())
        K = len(species_present)
        # --- 08b_fit_map_detector.py:262-268 (priors)
        if priors_mode == "uniform":
            priors = {sp: 1.0 / K for sp in species_present}
        else:
            total = float(sum(Z[sp].shape[0] for sp in species_present))
            priors = {sp: float(Z[sp].shape[0]) / total for sp in species_present}
        # --- :271 (means)
        means = {sp: np.mean(Z[sp], axis=0).astype(np.float32) for sp in species_present}
        covs, precs, logdets = {}, {}, {}
        # --- :278-296 (covariances through the reference's own estimate_cov / inv_and_logdet)
        if cov_type == "lda":
            Zc = np.concatenate([Z[sp] - means[sp][None, :] for sp in species_present], axis=0)
            cov_shared = m08b.estimate_cov(Zc, eps=eps, shrink=shrink, cov_structure=cov_structure)
            prec_shared, logdet_shared = m08b.inv_and_logdet(cov_shared)
            for sp in species_present:
                covs[sp], precs[sp], logdets[sp] = cov_shared, prec_shared, logdet_shared
        else:
            for sp in species_present:
                Zc = Z[sp] - means[sp][None, :]
                covs[sp] = m08b.estimate_cov(Zc, eps=eps, shrink=shrink, cov_structure=cov_structure)
                precs[sp], logdets[sp] = m08b.inv_and_logdet(covs[sp])
        # --- :302-312 (true-class scores, tau)
        scores_true = []
        for sp in species_present:
            lp = float(np.log(priors[sp] + 1e-12))
            s = np.array([m08b.gaussian_logpdf(z, means[sp], precs[sp], logdets[sp]) + lp for z in Z[sp]], dtype=np.float64)
            scores_true.extend(list(s))
            out[f"v{vi}_scores_{species.index(sp)}"] = s
        tau = float(np.quantile(np.array(scores_true, dtype=np.float64), float(tau_q))) if tau_q is not None else np.nan
        out[f"v{vi}_params_json"] = np.array(json.dumps(dict(cov_type=cov_type, cov_structure=cov_structure, shrink=shrink,
                                                              priors=priors_mode, tau_q=tau_q, eps=eps)))
        out[f"v{vi}_tau"] = np.array(tau, dtype=np.float64)
        for sp in species_present:
            i = species.index(sp)
            out[f"v{vi}_prior_{i}"] = np.array(priors[sp], dtype=np.float64)
            out[f"v{vi}_mean_{i}"] = means[sp]
            out[f"v{vi}_cov_{i}"] = covs[sp]
            out[f"v{vi}_prec_{i}"] = precs[sp]
            out[f"v{vi}_logdet_{i}"] = np.array(logdets[sp], dtype=np.float64)
    # estimate_cov edge cases: n < 2 and a singular covariance that needs the 1e-3 regularisation branch
    out["edge_cov_n1"] = m08b.estimate_cov(np.zeros((1, 5), np.float32), eps=1e-6, shrink=0.0, cov_structure="full")
    sing = np.ones((4, 4), dtype=np.float32)                     # rank 1 -> slogdet sign 0 -> + 1e-3 I
    p_s, ld_s = core.inv_and_logdet(sing.copy())
    out["edge_singular_in"], out["edge_singular_prec"], out["edge_singular_logdet"] = sing, p_s, np.array(ld_s)
    out["numpy_version"] = np.array(np.__version__)
    np.savez_compressed(HERE / "map_fit.npz", **out)

    # --------------------------------------------------------------- decide
    # config exactly as 08b writes it (:322-351), read back through the reference's readers
    D = 8
    names = ["Pleurodema_thaul", "Batrachyla_taeniata", "Zeta_extra", "Alpha_extra", "Batrachyla_leptopus"]
    means = {sp: rng.standard_normal(D).astype(np.float32) * 1.5 for sp in names}
    precs, logdets = {}, {}
    for sp in names:
        A = rng.standard_normal((D, D)).astype(np.float32) * 0.4 + np.eye(D, dtype=np.float32)
        cov = (A @ A.T + 0.05 * np.eye(D)).astype(np.float32)
        precs[sp], logdets[sp] = core.inv_and_logdet(cov)
    pri = {"Pleurodema_thaul": 0.4, "Batrachyla_taeniata": 0.3, "Zeta_extra": 0.1, "Alpha_extra": 0.1, "Batrachyla_leptopus": 0.1}
    cfg = {"chunk_seconds": 5.0,
           "map_detector": {"model": "gaussian_map",
                            "means": {sp: means[sp].astype(float).tolist() for sp in names},
                            "precision": {sp: precs[sp].astype(float).tolist() for sp in names},
                            "logdet_cov": {sp: float(logdets[sp]) for sp in names},
                            "tau": -14.0,
                            "meta_fit": {"chunk_seconds": 1.0, "per_species": {sp: {"prior": pri[sp]} for sp in names}}}}
    r_means, r_precs, r_logdets, r_tau = core.read_map_detector_params(cfg)
    r_species = sorted(set(r_means) & set(r_precs) & set(r_logdets))
    r_priors = core.get_priors_from_map_meta(cfg, r_species)
    r_dur = core.get_chunk_seconds_for_map(cfg)
    zs = [(means[names[i % 5]] + rng.standard_normal(D).astype(np.float32) * rng.uniform(0.2, 2.5)).astype(np.float32)
          for i in range(400)]
    zs.append(np.full(D, 40.0, dtype=np.float32))                # far away -> below tau -> NO_DETECT
    zs = np.stack(zs)

    cur = {"z": None}
    m10b.encode_wav_to_latent = lambda **kw: cur["z"]             # the name predict_one calls (10b:132)
    sess = m10b.MapDetectorSession(project_root=Path("."), config_path=Path("."), encoder_pt=Path("."),
                                   encoder_yaml=Path("."), device="cpu")
    sess.means, sess.precisions, sess.logdets, sess.tau = r_means, r_precs, r_logdets, r_tau
    sess.species, sess.priors, sess.duration, sess.encoder = r_species, r_priors, r_dur, None
    det, pred, best = [], [], []
    for z in zs:
        cur["z"] = z
        d, sp, sc = sess.predict_one(Path("unused.wav"))
        det.append(bool(d)); pred.append(sp if sp is not None else ""); best.append(sc)
    # per-species scores from the core function, for the kernel-level check
    scores = np.array([[core.gaussian_logpdf_from_precision(z, r_means[sp], r_precs[sp], r_logdets[sp]) +
                        float(np.log(float(r_priors.get(sp, 1e-12)) + 1e-12)) for sp in r_species] for z in zs], dtype=np.float64)
    np.savez_compressed(HERE / "map_decide.npz", Z=zs, detected=np.array(det), pred=np.array(pred),
                        best_score=np.array(best, dtype=np.float64), scores=scores,
                        species_json=np.array(json.dumps(r_species)), priors_json=np.array(json.dumps(r_priors)),
                        duration=np.array(r_dur), cfg_json=np.array(json.dumps(cfg)),
                        numpy_version=np.array(np.__version__))
    # no-tau / uniform-prior fallback variant
    cfg2 = json.loads(json.dumps(cfg))
    cfg2["map_detector"]["tau"] = None
    del cfg2["map_detector"]["meta_fit"]["per_species"]
    m2, p2, l2, t2 = core.read_map_detector_params(cfg2)
    pri2 = core.get_priors_from_map_meta(cfg2, r_species)
    sess.tau, sess.priors = t2, pri2
    det2, pred2, best2 = [], [], []
    for z in zs:
        cur["z"] = z
        d, sp, sc = sess.predict_one(Path("unused.wav"))
        det2.append(bool(d)); pred2.append(sp if sp is not None else ""); best2.append(sc)
    np.savez_compressed(HERE / "map_decide_notau.npz", detected=np.array(det2), pred=np.array(pred2),
                        best_score=np.array(best2, dtype=np.float64), priors_json=np.array(json.dumps(pri2)),
                        cfg_json=np.array(json.dumps(cfg2)))
    print("MAP golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
