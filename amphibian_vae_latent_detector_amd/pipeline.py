"""The callers either side of the hot path, with the reference's on-disk formats
(SURVEY.md §8(f) rank 1-2), driven by the GPU encode / decide:

  fit_radial_from_folder   = latent_space_exploration/08_fit_radial_detector.py main()  (:365-590)
                             -> config.json:radial_detector {centroids, thresholds, meta_fit} + config.json.bak
                             -> the "✅ sp: rk_in=… | rk_out=… | rk=…" log line 9105_…:11-13 parses
  benchmark_folder         = latent_space_exploration/10_benchmark_folder_detection.py main() (:326-456)
                             -> results.csv (file,true_species,pred_species,detected,correct,best_distance[,error])
                             -> summary.txt in the layout 9100_spearman_rk_analysis.py:53-58 parses
  run_qout_grid            = latent_space_exploration/run_qout_grid.sh (08 then 10 per q_out, outputs per run):
                             the train / eval folders are encoded ONCE and the latents stay resident.
  fit_map_from_folder      = latent_space_exploration/08b_fit_map_detector.py main() (:129-360)
                             -> config.json:map_detector {model, cov_type, ..., means, cov, precision, logdet_cov, tau, meta_fit}
  benchmark_folder_map     = latent_space_exploration/10b_benchmark_folder_detection_map.py main() (:286-407)
                             -> results.csv (…, best_score[, error]) + summary.txt ("… Summary (MAP)")

Plots are not produced (reporting, out of scope).  Fit arithmetic stays numpy on the host
(O(N*D), once per grid point); encode and decide run in the HIP kernels.
"""
from __future__ import annotations

import csv
import math
import shutil
from datetime import datetime
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import folder as _folder
from . import map_detector as _map
from . import radial as _radial
from .config import FrontendConfig
from .map_detector_core import load_json, save_json

AUDIO_EXTS = (".wav", ".WAV")


# ---------------------------------------------------------------------------- 08: fit + write config
def write_radial_config(cfg_path: Path, rd: Dict[str, Any], *, chunks_dir: Path, q_in: float, q_out: float,
                        chunk_seconds: float, fe: FrontendConfig, max_per_class: int, seed: int) -> Path:
    """08_fit_radial_detector.py:561-587: merge the fit into config.json (same keys, same types), after
    copying the previous file to config.json.bak.  Returns the backup path."""
    cfg_path = Path(cfg_path)
    cfg = load_json(cfg_path)
    if not isinstance(cfg.get("radial_detector"), dict):
        cfg["radial_detector"] = {}
    cfg["radial_detector"]["centroids"] = rd["centroids"]
    cfg["radial_detector"]["thresholds"] = rd["thresholds"]
    cfg["radial_detector"]["meta_fit"] = {
        "chunks_dir": str(chunks_dir), "chunks_name": Path(chunks_dir).name,
        "q_in": float(q_in), "q_out": float(q_out), "chunk_seconds": float(chunk_seconds),
        "sr": int(fe.sr), "n_mels": int(fe.n_mels), "target_frames": int(fe.target_frames),
        "fmin": float(fe.fmin), "fmax": float(fe.fmax), "hop_length": int(fe.hop_length), "n_fft": int(fe.n_fft),
        "max_per_class": int(max_per_class), "seed": int(seed), "per_species": rd["per_species"],
    }
    backup = cfg_path.with_suffix(cfg_path.suffix + ".bak")
    shutil.copy2(cfg_path, backup)
    save_json(cfg_path, cfg)
    return backup


def _frontend_from(config: Dict[str, Any], fe: FrontendConfig) -> Tuple[FrontendConfig, float]:
    """chunk_seconds of config.json overrides the duration (08:392-396), fallback 5.0."""
    try:
        chunk_seconds = float(config.get("chunk_seconds", 5.0))
    except Exception:  # noqa: BLE001
        chunk_seconds = 5.0
    return FrontendConfig(sr=fe.sr, duration=chunk_seconds, n_fft=fe.n_fft, hop_length=fe.hop_length, n_mels=fe.n_mels,
                          fmin=fe.fmin, fmax=fe.fmax, target_frames=fe.target_frames, pad_mode=fe.pad_mode), chunk_seconds


def fit_radial_from_folder(encoder, chunks_dir: Path, config_path: Path, fe: FrontendConfig, *, q_in: float = 0.95,
                           q_out: float = 0.01, max_per_class: int = 0, seed: int = 123,
                           cache_dir: Optional[Path] = None, apply_rms: bool = False, group=None,
                           latents: Optional[Tuple[Dict[str, np.ndarray], Dict[str, int]]] = None,
                           verbose: bool = True) -> Dict[str, Any]:
    """08's main(): encode `chunks_dir/<species>/*.wav` for the species listed in config.json (sharded
    over the ranks of `group`), fit centroids + thresholds, write them into config.json.
    `latents` = (Z_by_species, failed_by_species) skips the encode (resident latents of a q_out grid)."""
    if not (0.0 < q_in < 1.0) or not (0.0 < q_out < 1.0):
        raise ValueError("q_in and q_out must be in (0, 1)")
    config_path = Path(config_path)
    config = load_json(config_path)
    species = config.get("species")
    if not isinstance(species, list) or not all(isinstance(s, str) for s in species):
        raise ValueError("config.json debe tener un campo 'species' (lista de strings).")
    fe2, chunk_seconds = _frontend_from(config, fe)
    if latents is None:
        latents = _folder.encode_folder(encoder, chunks_dir, species, fe2, max_per_class=max_per_class, seed=seed,
                                        cache_dir=cache_dir, apply_rms=apply_rms, group=group, verbose=verbose)
    Z_by, failed = latents
    if not Z_by:
        raise RuntimeError("No se codificó ninguna especie. Revisa root y/o pipeline.")
    rank = _folder._dist_info(group)[0]
    # the fit's array work (centroids, distances, order statistics) on the GPU when the encoder is the HIP one (K9); the
    # result equals the host fit bit for bit, so the other ranks of a sharded run may take either
    session = encoder.session(fe2) if hasattr(encoder, "session") else None
    rd = _radial.fit_radial_detector(Z_by, q_in, q_out, failed=failed, verbose=verbose and rank == 0, session=session)
    if rank == 0:
        write_radial_config(config_path, rd, chunks_dir=Path(chunks_dir).resolve(), q_in=q_in, q_out=q_out,
                            chunk_seconds=chunk_seconds, fe=fe2, max_per_class=max_per_class, seed=seed)
    return rd


# ---------------------------------------------------------------------------- 10: folder benchmark
def list_audio_files(root: Path) -> List[Path]:
    """10_benchmark_folder_detection.py:98-103 (recursive, .wav / .WAV, sorted)."""
    return sorted(p for p in Path(root).rglob("*") if p.is_file() and p.suffix in AUDIO_EXTS)


def _class_files(root: Path) -> List[Tuple[str, Path]]:
    dirs = sorted(d for d in Path(root).iterdir() if d.is_dir() and not d.name.startswith("."))
    if not dirs:
        raise RuntimeError(f"No encontré subcarpetas de especies en: {root}")
    return [(d.name, w) for d in dirs for w in list_audio_files(d)]


def summary_text(rows: Sequence[Dict[str, Any]], title: str = "=== Detection Benchmark Summary ===") -> str:
    """write_summary of 10_benchmark_folder_detection.py:278-301 (10b:243-267 with the "(MAP)" title) for the
    non-ERROR rows."""
    ok = [r for r in rows if r["pred_species"] != "ERROR"]
    total = len(ok)
    correct = sum(bool(r["correct"]) for r in ok)
    no_det = sum(r["pred_species"] == "NO_DETECT" for r in ok)
    lines = [title, f"Total files: {total}",
             f"Correct: {correct}  | Accuracy: {(correct / total if total else 0.0) * 100:.2f}%",
             f"NO_DETECT: {no_det} | Rate: {(no_det / total if total else 0.0) * 100:.2f}%", "", "=== Per-class ==="]
    per: Dict[str, List[Dict[str, Any]]] = {}
    for r in ok:
        per.setdefault(r["true_species"], []).append(r)
    stats = [(sp, len(rs), sum(bool(r["correct"]) for r in rs) / len(rs),
              sum(r["pred_species"] == "NO_DETECT" for r in rs) / len(rs)) for sp, rs in sorted(per.items())]
    for sp, n, acc, nd in sorted(stats, key=lambda t: -t[2]):         # groupby key order, then acc descending (stable)
        lines.append(f"- {sp:30s}  n={n:4d}  acc={acc * 100:6.2f}%  no_detect={nd * 100:6.2f}%")
    return "\n".join(lines)


def write_results(rows: Sequence[Dict[str, Any]], out_dir: Path, value_col: str = "best_distance",
                  title: str = "=== Detection Benchmark Summary ===") -> Tuple[Path, Path]:
    out_dir = Path(out_dir)
    out_dir.mkdir(parents=True, exist_ok=True)
    cols = ["file", "true_species", "pred_species", "detected", "correct", value_col]
    if any("error" in r for r in rows):
        cols.append("error")
    csv_path = out_dir / "results.csv"
    with open(csv_path, "w", newline="", encoding="utf-8") as fh:
        w = csv.writer(fh, lineterminator="\n")
        w.writerow(cols)
        for r in rows:
            bd = r[value_col]
            w.writerow([r["file"], r["true_species"], r["pred_species"], bool(r["detected"]), bool(r["correct"]),
                        "" if (isinstance(bd, float) and math.isnan(bd)) else repr(float(bd))] +
                       ([r.get("error", "")] if "error" in cols else []))
    summary_path = out_dir / "summary.txt"
    summary_path.write_text(summary_text(rows, title), encoding="utf-8")
    return csv_path, summary_path


def rows_from_decisions(files: Sequence[Tuple[str, Path]], status: np.ndarray, species: Sequence[Optional[str]],
                        best_d: np.ndarray, value_col: str = "best_distance") -> List[Dict[str, Any]]:
    """One results.csv row per file (10:401-418, 10b:354-375); files the ingest flagged become ERROR rows."""
    rows = []
    for (true_sp, wav), st, pred, bd in zip(files, status, species, best_d):
        if st != 0:
            rows.append({"file": str(wav), "true_species": true_sp, "pred_species": "ERROR", "detected": False,
                         "correct": False, value_col: float("nan"), "error": f"WAV ingest status {int(st)}"})
            continue
        pred_species = pred if pred is not None else "NO_DETECT"
        rows.append({"file": str(wav), "true_species": true_sp, "pred_species": pred_species,
                     "detected": pred is not None, "correct": pred_species == true_sp, value_col: float(bd)})
    return rows


def benchmark_folder(encoder, root: Path, config_path: Path, fe: FrontendConfig, out_dir: Optional[Path] = None, *,
                     apply_rms: bool = False, group=None,
                     latents: Optional[Tuple[List[Tuple[str, Path]], np.ndarray, np.ndarray]] = None
                     ) -> List[Dict[str, Any]]:
    """10's main() without the plots: every WAV under `root/<class>/` is encoded and decided on the GPU with
    the detector stored in config.json; returns the rows and (rank 0, when out_dir is given) writes
    results.csv + summary.txt.  `latents` = (files, Z, status) reuses resident latents."""
    config = load_json(Path(config_path))
    fe2, _ = _frontend_from(config, fe)
    if latents is None:
        files = _class_files(Path(root).expanduser().resolve())
        if not files:
            raise RuntimeError("No se procesó ningún archivo (rows vacío).")
        Z, status = _folder.encode_file_list(encoder, [w for _, w in files], fe2, apply_rms=apply_rms, group=group)
    else:
        files, Z, status = latents
    det = _radial.RadialDetector.from_config(config, latent_dim=Z.shape[1])
    pred, best = det.decide(encoder.session(fe2), Z)
    rows = rows_from_decisions(files, status, det.species(pred), best)
    if out_dir is not None and _folder._dist_info(group)[0] == 0:
        write_results(rows, out_dir)
    return rows


# ---------------------------------------------------------------------------- run_qout_grid.sh
def run_qout_grid(encoder, train_dir: Path, eval_dir: Path, config_path: Path, fe: FrontendConfig, grid_root: Path, *,
                  q_in: float = 0.95, q_out_grid: Sequence[float] = (0.10, 0.15, 0.20, 0.25), max_per_class: int = 400,
                  seed: int = 123, apply_rms: bool = False, group=None, verbose: bool = True) -> Dict[float, Dict[str, Any]]:
    """run_qout_grid.sh:13-58 with the encodes hoisted out of the loop: both folders are encoded once, then
    for every q_out the fit (08) and the folder benchmark (10) run on the resident latents and their
    outputs go to <grid_root>/qout_<q>/{summary.txt, results.csv, config_used.json, config_snapshot.json}."""
    config = load_json(Path(config_path))
    species = config.get("species")
    fe2, _ = _frontend_from(config, fe)
    train = _folder.encode_folder(encoder, train_dir, species, fe2, max_per_class=max_per_class, seed=seed,
                                  apply_rms=apply_rms, group=group, verbose=verbose)
    files = _class_files(Path(eval_dir).expanduser().resolve())
    Zev, stev = _folder.encode_file_list(encoder, [w for _, w in files], fe2, apply_rms=apply_rms, group=group)
    rank = _folder._dist_info(group)[0]
    out: Dict[float, Dict[str, Any]] = {}
    for q_out in q_out_grid:
        rd = fit_radial_from_folder(encoder, train_dir, config_path, fe, q_in=q_in, q_out=q_out, max_per_class=max_per_class,
                                    seed=seed, group=group, latents=train, verbose=verbose)
        run_dir = Path(grid_root) / f"qout_{q_out:.2f}"
        rows = benchmark_folder(encoder, eval_dir, config_path, fe, run_dir, group=group, latents=(files, Zev, stev))
        if rank == 0:
            shutil.copy2(config_path, run_dir / "config_used.json")
            m = rd["per_species"]
            six = lambda v: float(f"{v:.6f}") if v is not None else float("nan")   # what 9105 parses back from the log line
            save_json(run_dir / "config_snapshot.json", {       # 9105_make_config_snapshot_from_log.py:50-58
                "timestamp": datetime.now().isoformat(), "q_in": float(q_in), "q_out": float(q_out),
                "rk_in_per_species": {sp: six(m[sp]["rk_in"]) for sp in m},
                "rk_out_per_species": {sp: six(m[sp]["rk_out"]) for sp in m},
                "rk_per_species": {sp: six(m[sp]["rk_final"]) for sp in m},
                "source_log": "resident latents (no log file: fit and benchmark ran in one process)"})
        out[float(q_out)] = {"radial_detector": rd, "rows": rows}
    return out


# ---------------------------------------------------------------------------- 08b / 10b: Gaussian-MAP detector
def fit_map_from_folder(encoder, chunks_dir: Path, config_path: Path, fe: FrontendConfig, *, cov_type: str = "lda",
                        cov_structure: str = "full", priors: str = "empirical", eps: float = 1e-6, shrink: float = 0.0,
                        set_tau_q: Optional[float] = None, max_per_class: int = 0, seed: int = 123,
                        cache_dir: Optional[Path] = None, apply_rms: bool = False, group=None,
                        latents: Optional[Tuple[Dict[str, np.ndarray], Dict[str, int]]] = None,
                        verbose: bool = True) -> Dict[str, Any]:
    """08b's main(): encode `chunks_dir/<species>/*.wav` (same loop, sampling and cache files as 08), fit the
    Gaussian-MAP parameters on the host (the tau quantile's N x K score evaluations on the GPU), write the
    `map_detector` block into config.json after copying the previous file to config.json.bak."""
    config_path = Path(config_path)
    config = load_json(config_path)
    species = config.get("species")
    if not isinstance(species, list) or not all(isinstance(s, str) for s in species):
        raise ValueError("config.json debe tener un campo 'species' (lista de strings).")
    fe2, chunk_seconds = _frontend_from(config, fe)
    if latents is None:
        latents = _folder.encode_folder(encoder, chunks_dir, species, fe2, max_per_class=max_per_class, seed=seed,
                                        cache_dir=cache_dir, apply_rms=apply_rms, group=group, verbose=verbose)
    Z_by, failed = latents
    if not Z_by:
        raise RuntimeError("No se codificó ninguna especie. Revisa root y/o pipeline.")
    rank = _folder._dist_info(group)[0]
    meta = {"chunks_dir": str(Path(chunks_dir).resolve()), "chunks_name": Path(chunks_dir).name,
            "chunk_seconds": float(chunk_seconds), "sr": int(fe2.sr), "n_mels": int(fe2.n_mels),
            "target_frames": int(fe2.target_frames), "fmin": float(fe2.fmin), "fmax": float(fe2.fmax),
            "hop_length": int(fe2.hop_length), "n_fft": int(fe2.n_fft), "max_per_class": int(max_per_class), "seed": int(seed)}
    md = _map.fit_map_detector(Z_by, cov_type=cov_type, cov_structure=cov_structure, priors=priors, eps=eps, shrink=shrink,
                               set_tau_q=set_tau_q, failed=failed, session=encoder.session(fe2), meta=meta,
                               verbose=verbose and rank == 0)
    if rank == 0:
        config["map_detector"] = md
        shutil.copy2(config_path, config_path.with_suffix(config_path.suffix + ".bak"))
        save_json(config_path, config)
    return md


def benchmark_folder_map(encoder, root: Path, config_path: Path, fe: FrontendConfig, out_dir: Optional[Path] = None, *,
                         apply_rms: bool = False, group=None,
                         latents: Optional[Tuple[List[Tuple[str, Path]], np.ndarray, np.ndarray]] = None
                         ) -> List[Dict[str, Any]]:
    """10b's main() without the plots: encode every WAV under `root/<class>/`, score + decide on the GPU with the
    `map_detector` of config.json (duration from its meta_fit.chunk_seconds, map_detector_core.py:359-371)."""
    config = load_json(Path(config_path))
    dur = _map.get_chunk_seconds_for_map(config)
    fe2 = FrontendConfig(sr=fe.sr, duration=dur, n_fft=fe.n_fft, hop_length=fe.hop_length, n_mels=fe.n_mels, fmin=fe.fmin,
                         fmax=fe.fmax, target_frames=fe.target_frames, pad_mode=fe.pad_mode)
    if latents is None:
        files = _class_files(Path(root).expanduser().resolve())
        if not files:
            raise RuntimeError("No se procesó ningún archivo (rows vacío).")
        Z, status = _folder.encode_file_list(encoder, [w for _, w in files], fe2, apply_rms=apply_rms, group=group)
    else:
        files, Z, status = latents
    det = _map.MapDetector.from_config(config, latent_dim=Z.shape[1])
    pred, best = det.decide(encoder.session(fe2), Z)
    rows = rows_from_decisions(files, status, det.species(pred), best, value_col="best_score")
    if out_dir is not None and _folder._dist_info(group)[0] == 0:
        write_results(rows, out_dir, value_col="best_score", title="=== Detection Benchmark Summary (MAP) ===")
    return rows
