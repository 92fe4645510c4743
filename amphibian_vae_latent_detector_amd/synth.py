"""Deterministic synthetic chunk generator.

The reference ships no audio (README.md:78-81), so benchmarks and tests run on
synthetic mono PCM16 chunks at the preset sample rate: three enveloped tones +
noise with >= 40 dB of spectral contrast, a log-uniform overall gain (exercises
the RMS scaling and clipping), and every 50th chunk digital silence (exercises
the `rms < 1e-4` gate and the std == 0 path).
"""
from __future__ import annotations

import wave
from pathlib import Path
from typing import Sequence

import numpy as np

SPECIES = (  # latent_space_exploration/09_evaluate_wav_detection.py:61-66
    "Batrachyla_leptopus",
    "Batrachyla_taeniata",
    "Calyptocephalella_gayi",
    "Pleurodema_thaul",
)


def synth_chunk_pcm16(i: int, n_samples: int, sr: int) -> np.ndarray:
    """Chunk `i` as int16 [n_samples]; depends only on (i, n_samples, sr)."""
    if i % 50 == 49:
        return np.zeros(n_samples, dtype=np.int16)
    rng = np.random.default_rng(seed=1234 + i)
    t = np.arange(n_samples, dtype=np.float64) / float(sr)
    dur = n_samples / float(sr)
    y = np.zeros(n_samples, dtype=np.float64)
    for _ in range(3):
        f = rng.uniform(200.0, 8000.0)
        a = rng.uniform(0.05, 0.5)
        ph = rng.uniform(0.0, 2.0 * np.pi)
        centre = rng.uniform(0.2, 0.8) * dur
        width = rng.uniform(0.2, 0.9) * dur
        u = (t - (centre - width / 2)) / width
        env = np.where((u >= 0) & (u <= 1), 0.5 - 0.5 * np.cos(2 * np.pi * u), 0.0)
        y += a * np.sin(2 * np.pi * f * t + ph) * env
    y += 0.02 * rng.standard_normal(n_samples)
    gain = np.exp(rng.uniform(np.log(0.003), np.log(0.7)))
    y = np.clip(y * gain, -1.0, 1.0)
    return np.rint(y * 32767.0).astype(np.int16)


def synth_batch_pcm16(indices: Sequence[int], n_samples: int, sr: int, workers: int = 1) -> np.ndarray:
    """Chunks `indices` as int16 [len(indices), n_samples].  `workers` > 1 (contiguous index ranges only, e.g. the
    100 000 chunks of BASELINE configs[2]) fans the generation out over child processes that run THIS FILE as a script
    (numpy only: they import neither torch nor the HIP library and never touch the GPU, so it is safe to call after the GPU
    has been initialised)."""
    idx = np.asarray(indices)
    if workers > 1 and idx.size >= 64 * workers and np.array_equal(idx, np.arange(idx[0], idx[0] + idx.size)):
        import subprocess
        import sys
        import tempfile
        lo0, n = int(idx[0]), int(idx.size)
        step = -(-n // workers)
        out = np.empty((n, n_samples), dtype=np.int16)
        with tempfile.TemporaryDirectory(prefix="amph_synth_") as td:
            jobs = []
            for w, a in enumerate(range(0, n, step)):
                dst = Path(td) / f"part{w}.npy"
                jobs.append((a, min(n, a + step), dst, subprocess.Popen(
                    [sys.executable, str(Path(__file__).resolve()), str(lo0 + a), str(lo0 + min(n, a + step)), str(n_samples),
                     str(sr), str(dst)])))
            for a, b, dst, proc in jobs:
                if proc.wait() != 0:
                    raise RuntimeError(f"synth worker for chunks [{lo0 + a}, {lo0 + b}) failed")
                out[a:b] = np.load(dst)
        return out
    out = np.empty((len(indices), n_samples), dtype=np.int16)
    for r, i in enumerate(indices):
        out[r] = synth_chunk_pcm16(int(i), n_samples, sr)
    return out


def write_wav_pcm16(path: Path, q: np.ndarray, sr: int) -> None:
    q = np.ascontiguousarray(q, dtype="<i2")
    ch = 1 if q.ndim == 1 else q.shape[1]
    with wave.open(str(path), "wb") as w:
        w.setnchannels(ch)
        w.setsampwidth(2)
        w.setframerate(int(sr))
        w.writeframes(q.tobytes())


def write_synth_folder(root: Path, n_chunks: int, n_samples: int, sr: int,
                       species: Sequence[str] = SPECIES) -> list:
    """<root>/<species_k>/chunk_<i:06d>.wav, round-robin over species.
    Returns the list of (species, path) in creation order."""
    root = Path(root)
    made = []
    for sp in species:
        (root / sp).mkdir(parents=True, exist_ok=True)
    for i in range(n_chunks):
        sp = species[i % len(species)]
        p = root / sp / f"chunk_{i:06d}.wav"
        write_wav_pcm16(p, synth_chunk_pcm16(i, n_samples, sr), sr)
        made.append((sp, p))
    return made


def synth_folder_plan(n_chunks: int, max_per_class: int = 0, seed: int = 123, species: Sequence[str] = SPECIES):
    """The ordered chunk list 08_fit_radial_detector.py:374,461-484 would build over write_synth_folder(root, n_chunks):
    species in config order; inside a species the name-sorted files (chunk_<i:06d>.wav: numeric order); with
    max_per_class > 0 a `random.sample` per species from ONE generator seeded once.  For chunk matrices that never
    become files (bench.py --config3).  -> {species: [chunk index, ...]}"""
    import random
    random.seed(seed)
    out = {}
    for k, sp in enumerate(species):
        idx = list(range(k, n_chunks, len(species)))
        if not idx:
            continue
        if max_per_class and len(idx) > max_per_class:
            idx = random.sample(idx, max_per_class)
        out[sp] = idx
    return out


if __name__ == "__main__":      # worker of synth_batch_pcm16(workers > 1): lo hi n_samples sr out.npy
    import sys
    _lo, _hi, _n, _sr = (int(v) for v in sys.argv[1:5])
    np.save(sys.argv[5], synth_batch_pcm16(range(_lo, _hi), _n, _sr))
