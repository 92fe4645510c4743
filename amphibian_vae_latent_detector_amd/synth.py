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


def synth_batch_pcm16(indices: Sequence[int], n_samples: int, sr: int) -> np.ndarray:
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
